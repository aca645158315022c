// In-lane (register-resident) FFT building blocks for the fused STFT kernel.
//
// Everything here is straight-line code after inlining: sizes, strides and twiddle
// factors are template parameters, so arrays are scalarised into VGPRs and every twiddle
// is an instruction literal.  Radix-2 decimation in time with the trivial twiddles
// (1, -i, (1-i)/sqrt2, (-1-i)/sqrt2) specialised, which gives split-radix-like operation
// counts (N = 16 complex: 144 adds + 24 multiplies); butterflies with a general twiddle are
// written in multiply-add form (six instructions instead of eight).
//
// Compiles for host and device so the host unit test (tests/test_fft_inlane.py) can
// exercise the same templates.
#pragma once
#include <hip/hip_runtime.h>

#include <type_traits>

#include "twiddle_consts.h"

#define PDS_HD __host__ __device__ __forceinline__

// general-twiddle butterflies in multiply-add form (see CFFT::run, rdft_scaled)
#ifndef PDS_FMA_BUTTERFLY
#define PDS_FMA_BUTTERFLY 1
#endif

namespace pds {
namespace inl {

template <int I>
using Int = std::integral_constant<int, I>;

template <int B, int E, typename F>
PDS_HD void static_for(F &&f) {
  if constexpr (B < E) {
    f(Int<B>{});
    static_for<B + 1, E>(f);
  }
}

// e^{-2 pi i K / N}; N must divide 256
template <int N, int K>
struct Tw {
  static_assert(256 % N == 0, "twiddle table covers divisors of 256");
  static constexpr int idx = (((K % N) + N) % N) * (256 / N);
  static constexpr float re = (float)kCos256[idx];
  static constexpr float im = (float)(-kSin256[idx]);
};

// (yr + i yi) = (xr + i xi) * e^{-2 pi i K / N}
template <int N, int K>
PDS_HD void mul_tw(float xr, float xi, float &yr, float &yi) {
  constexpr int k = ((K % N) + N) % N;
  constexpr float h = 0.70710678118654752440f;
  if constexpr (k == 0) {
    yr = xr;
    yi = xi;
  } else if constexpr (4 * k == N) {  // -i
    yr = xi;
    yi = -xr;
  } else if constexpr (2 * k == N) {  // -1
    yr = -xr;
    yi = -xi;
  } else if constexpr (4 * k == 3 * N) {  // +i
    yr = -xi;
    yi = xr;
  } else if constexpr (8 * k == N) {  // (1 - i) / sqrt 2
    yr = (xr + xi) * h;
    yi = (xi - xr) * h;
  } else if constexpr (8 * k == 3 * N) {  // (-1 - i) / sqrt 2
    yr = (xi - xr) * h;
    yi = -(xr + xi) * h;
  } else {
    constexpr float c = Tw<N, k>::re, s = Tw<N, k>::im;
    yr = xr * c - xi * s;
    yi = xr * s + xi * c;
  }
}

// Complex FFT of N points read with stride IS from (xr, xi); natural-order output.
template <int N, int IS>
struct CFFT {
  static PDS_HD void run(const float *xr, const float *xi, float *yr, float *yi) {
    constexpr int H = N / 2;
    float er[H], ei[H], qr[H], qi[H];
    CFFT<H, 2 * IS>::run(xr, xi, er, ei);
    CFFT<H, 2 * IS>::run(xr + IS, xi + IS, qr, qi);
    static_for<0, H>([&](auto kk) {
      constexpr int k = decltype(kk)::value;
      if constexpr (k != 0 && 4 * k != N && 8 * k != N && 8 * k != 3 * N) {
#if PDS_FMA_BUTTERFLY
        // general twiddle: e + W q as two chained multiply-adds per component, e - W q as
        // 2 e - (e + W q): six instructions instead of eight
        constexpr float c = Tw<N, k>::re, s = Tw<N, k>::im;
        const float sr = fmaf(qr[k], c, fmaf(-s, qi[k], er[k]));
        const float si = fmaf(qr[k], s, fmaf(c, qi[k], ei[k]));
        yr[k] = sr;
        yi[k] = si;
        yr[k + H] = fmaf(2.0f, er[k], -sr);
        yi[k + H] = fmaf(2.0f, ei[k], -si);
        return;
#endif
      }
      float tr, ti;
      mul_tw<N, k>(qr[k], qi[k], tr, ti);
      yr[k] = er[k] + tr;
      yi[k] = ei[k] + ti;
      yr[k + H] = er[k] - tr;
      yi[k + H] = ei[k] - ti;
    });
  }
};

template <int IS>
struct CFFT<1, IS> {
  static PDS_HD void run(const float *xr, const float *xi, float *yr, float *yi) {
    yr[0] = xr[0];
    yi[0] = xi[0];
  }
};

template <int IS>
struct CFFT<2, IS> {
  static PDS_HD void run(const float *xr, const float *xi, float *yr, float *yi) {
    const float ar = xr[0], ai = xi[0], br = xr[IS], bi = xi[IS];
    yr[0] = ar + br;
    yi[0] = ai + bi;
    yr[1] = ar - br;
    yi[1] = ai - bi;
  }
};

template <int IS>
struct CFFT<4, IS> {
  static PDS_HD void run(const float *xr, const float *xi, float *yr, float *yi) {
    const float ar = xr[0] + xr[2 * IS], ai = xi[0] + xi[2 * IS];
    const float br = xr[0] - xr[2 * IS], bi = xi[0] - xi[2 * IS];
    const float cr = xr[IS] + xr[3 * IS], ci = xi[IS] + xi[3 * IS];
    const float dr = xr[IS] - xr[3 * IS], di = xi[IS] - xi[3 * IS];
    yr[0] = ar + cr;
    yi[0] = ai + ci;
    yr[2] = ar - cr;
    yi[2] = ai - ci;
    yr[1] = br + di;  // b - i d
    yi[1] = bi - dr;
    yr[3] = br - di;  // b + i d
    yi[3] = bi + dr;
  }
};

// DFT of M real points a[0..M): produces
//   even_sum = sum a[2m], odd_sum = sum a[2m+1]     (so A[0] = even + odd, A[M/2] = even - odd)
//   (Ar, Ai)[k] for k = 1 .. M/2 - 1, SCALED: 2 * A[k], except k = M/4 which is A[k] itself.
// The factor is undone by the caller's next twiddle multiply (its table is pre-scaled), which
// saves the halving of the even/odd split.  A[k] = sum_n a[n] e^{-2 pi i n k / M}.
template <int M>
PDS_HD void rdft_scaled(const float *a, float &even_sum, float &odd_sum, float *Ar, float *Ai) {
  constexpr int H = M / 2;
  float zr[H], zi[H];
  CFFT<H, 2>::run(a, a + 1, zr, zi);  // z[m] = a[2m] + i a[2m+1]
  even_sum = zr[0];
  odd_sum = zi[0];
  if constexpr (H >= 2) {
    Ar[H / 2] = zr[H / 2];  // A[M/4] = conj(Z[H/2])
    Ai[H / 2] = -zi[H / 2];
  }
  static_for<1, H / 2>([&](auto kk) {
    constexpr int k = decltype(kk)::value;
    // Fe = (Z[k] + conj Z[H-k]) / 2, Fo = (Z[k] - conj Z[H-k]) / (2i); A[k] = Fe + W_M^k Fo,
    // A[H-k] = conj(Fe - W_M^k Fo).  Everything below carries the factor 2.
    const float sr = zr[k] + zr[H - k], si = zi[k] - zi[H - k];
    const float dr = zr[k] - zr[H - k], di = zi[k] + zi[H - k];
    constexpr float wr = Tw<M, k>::re, wi = Tw<M, k>::im;
#if PDS_FMA_BUTTERFLY
    // s + W (di - i dr) as chained multiply-adds, the mirrored output from it: ten instructions
    // per pair instead of twelve
    const float ar = fmaf(wr, di, fmaf(wi, dr, sr));
    const float ai = fmaf(wi, di, fmaf(-wr, dr, si));
    Ar[k] = ar;
    Ai[k] = ai;
    Ar[H - k] = fmaf(2.0f, sr, -ar);
    Ai[H - k] = fmaf(-2.0f, si, ai);
#else
    const float tr = wr * di + wi * dr;  // W * (di - i dr)
    const float ti = wi * di - wr * dr;
    Ar[k] = sr + tr;
    Ai[k] = si + ti;
    Ar[H - k] = sr - tr;
    Ai[H - k] = ti - si;
#endif
  });
}

// DFT of M real points by decimation in time ON REAL DATA (no detour over a complex transform of half the size and
// its untangling): the transform of the even samples E and of the odd samples O -- both of real sequences, so both
// half spectra -- are combined as A[k] = E[k] + W_M^k O[k], A[M/2 - k] = conj(E[k] - W_M^k O[k]) for k = 1 ..
// M/4 - 1 (six multiply-adds per pair), A[M/4] = E[M/4] - i O[M/4] (both real: free), A[0] / A[M/2] = E[0] +- O[0].
// 6 (M/4 - 1) + 2 instructions per level: 162 for M = 32 against the 230 of rdft_scaled (a 16-point complex
// transform and 70 instructions of untangling), 418 against 580 for M = 64; literal-zero inputs (the padded tail of
// the frame) prune both alike.  Outputs UNSCALED: (Ar, Ai)[k] = A[k], k = 1 .. M/2 - 1; r0 = A[0], rh = A[M/2].
template <int M, int IS>
struct RDIT {
  static PDS_HD void run(const float *x, float &r0, float &rh, float *Ar, float *Ai) {
    constexpr int H = M / 2, Q = M / 4;
    float e0, eh, o0, oh;
    float Er[Q > 1 ? Q : 1], Ei[Q > 1 ? Q : 1], Or[Q > 1 ? Q : 1], Oi[Q > 1 ? Q : 1];
    RDIT<H, 2 * IS>::run(x, e0, eh, Er, Ei);
    RDIT<H, 2 * IS>::run(x + IS, o0, oh, Or, Oi);
    r0 = e0 + o0;
    rh = e0 - o0;
    Ar[Q] = eh;
    Ai[Q] = -oh;
    static_for<1, Q>([&](auto kk) {
      constexpr int k = decltype(kk)::value;
      constexpr float c = Tw<M, k>::re, s = Tw<M, k>::im;
      // E + W O as chained multiply-adds per component, the mirrored output conj(E - W O) = conj(2 E - (E + W O))
      const float ar = fmaf(Or[k], c, fmaf(-s, Oi[k], Er[k]));
      const float ai = fmaf(Or[k], s, fmaf(c, Oi[k], Ei[k]));
      Ar[k] = ar;
      Ai[k] = ai;
      Ar[H - k] = fmaf(2.0f, Er[k], -ar);
      Ai[H - k] = fmaf(-2.0f, Ei[k], ai);
    });
  }
};
template <int IS>
struct RDIT<2, IS> {
  static PDS_HD void run(const float *x, float &r0, float &rh, float *, float *) {
    r0 = x[0] + x[IS];
    rh = x[0] - x[IS];
  }
};
template <int IS>
struct RDIT<4, IS> {
  static PDS_HD void run(const float *x, float &r0, float &rh, float *Ar, float *Ai) {
    const float e0 = x[0] + x[2 * IS], e1 = x[0] - x[2 * IS];
    const float o0 = x[IS] + x[3 * IS], o1 = x[IS] - x[3 * IS];
    r0 = e0 + o0;
    rh = e0 - o0;
    Ar[1] = e1;
    Ai[1] = -o1;
  }
};
// ... with the interface of rdft_scaled (even_sum / odd_sum instead of A[0] / A[M/2]) and UNSCALED outputs
template <int M>
PDS_HD void rdft_dit(const float *a, float &even_sum, float &odd_sum, float *Ar, float *Ai) {
  static_assert(M >= 8 && (M & (M - 1)) == 0, "power-of-two in-lane transform");
  constexpr int H = M / 2, Q = M / 4;
  float eh, oh;
  float Er[Q], Ei[Q], Or[Q], Oi[Q];
  RDIT<H, 2>::run(a, even_sum, eh, Er, Ei);
  RDIT<H, 2>::run(a + 1, odd_sum, oh, Or, Oi);
  Ar[Q] = eh;
  Ai[Q] = -oh;
  static_for<1, Q>([&](auto kk) {
    constexpr int k = decltype(kk)::value;
    constexpr float c = Tw<M, k>::re, s = Tw<M, k>::im;
    const float ar = fmaf(Or[k], c, fmaf(-s, Oi[k], Er[k]));
    const float ai = fmaf(Or[k], s, fmaf(c, Oi[k], Ei[k]));
    Ar[k] = ar;
    Ai[k] = ai;
    Ar[H - k] = fmaf(2.0f, Er[k], -ar);
    Ai[H - k] = fmaf(-2.0f, Ei[k], ai);
  });
}

// The K inter-stage twiddles W^k, k = 1..K (K = 15 or 31), of one lane (W = e^{-2 pi i r / N}, a unit complex
// number per lane) regenerated from three seeds w1 = W, w4 = W^4, wq = W^((K + 1) / 2) instead of being held in
// 2 K registers: K - 3 complex products (four instructions each), none more than four (K = 15) or five
// (K = 31) products deep, so a twiddle carries a handful of float32 roundings (tests/test_twiddle_chain.py
// replays this routine against float64: max error 1.3e-7 / 2e-7 over all lanes of the 32 x 16 / 64 x 16
// geometries).  t[(K + 1) / 2] comes out DOUBLED: the caller feeds rdft_scaled a window times 1/2, which makes
// that routine's outputs A[k] for k != M/4 and A[M/4] / 2 (see rdft_scaled), so no twiddle needs a scale factor
// of its own.
PDS_HD void cmul(float ar, float ai, float br, float bi, float &cr, float &ci) {
  cr = fmaf(ar, br, -(ai * bi));
  ci = fmaf(ar, bi, ai * br);
}
// (DOUBLE_Q = false: every twiddle of unit size, for the unscaled outputs of rdft_dit and a whole window)
template <int K, bool DOUBLE_Q = true>
PDS_HD void twiddle_chain(float w1r, float w1i, float w4r, float w4i, float wqr, float wqi, float *tr, float *ti) {
  static_assert(K == 15 || K == 31, "twiddles of the 32- and 64-point in-lane transforms");
  constexpr int Q = (K + 1) / 2;  // 8 or 16
  tr[1] = w1r, ti[1] = w1i;
  tr[4] = w4r, ti[4] = w4i;
  cmul(w1r, w1i, w1r, w1i, tr[2], ti[2]);
  cmul(tr[2], ti[2], w1r, w1i, tr[3], ti[3]);
  cmul(w1r, w1i, w4r, w4i, tr[5], ti[5]);
  cmul(tr[2], ti[2], w4r, w4i, tr[6], ti[6]);
  cmul(tr[3], ti[3], w4r, w4i, tr[7], ti[7]);
  if constexpr (K == 31) {
    cmul(w4r, w4i, w4r, w4i, tr[8], ti[8]);
    static_for<1, 8>([&](auto kk) {
      constexpr int k = decltype(kk)::value;
      cmul(tr[k], ti[k], tr[8], ti[8], tr[8 + k], ti[8 + k]);
    });
  }
  static_for<1, Q>([&](auto kk) {
    constexpr int k = decltype(kk)::value;
    cmul(tr[k], ti[k], wqr, wqi, tr[Q + k], ti[Q + k]);
  });
  if constexpr (DOUBLE_Q) {
    tr[Q] = wqr + wqr, ti[Q] = wqi + wqi;
  } else {
    tr[Q] = wqr, ti[Q] = wqi;
  }
}
PDS_HD void twiddle_chain15(float w1r, float w1i, float w4r, float w4i, float w8r, float w8i, float *tr, float *ti) {
  twiddle_chain<15>(w1r, w1i, w4r, w4i, w8r, w8i, tr, ti);
}

// cos / -sin of 2 pi K / M for the mixed-radix sizes (M divides 600)
template <int M, int K>
struct Tw600 {
  static_assert(600 % M == 0, "second twiddle table covers divisors of 600");
  static constexpr int idx = (((K % M) + M) % M) * (600 / M);
  static constexpr float re = (float)kCos600[idx];
  static constexpr float im = (float)(-kSin600[idx]);
};

constexpr bool is_pow2(int v) { return v > 0 && (v & (v - 1)) == 0; }

// DFT of M real points for sizes that are not powers of two (M = 10, 20, 25, 30 ...), same
// interface as rdft_scaled but UNSCALED outputs (Ar, Ai)[k], k = 1 .. (M - 1) / 2:
//   even M: even_sum / odd_sum as in rdft_scaled;  odd M: even_sum = sum a[n], odd_sum = 0
// Direct evaluation on the symmetric / antisymmetric halves of the input,
//   Re A[k] = a[0] + sum_j (a[j] + a[M-j]) cos(2 pi j k / M)  (+ (-1)^k a[M/2]),
//   Im A[k] =       - sum_j (a[j] - a[M-j]) sin(2 pi j k / M),      j = 1 .. (M - 1) / 2,
// i.e. about M^2 / 2 fused multiply-adds with literal coefficients: for the M <= 30 in use this
// is on a par with a mixed-radix network (25: 288 FMA) and has no data movement at all.
// 25 real points as 5 x 5 (n = 5 n1 + n2, k = k1 + 5 k2): five real 5-point transforms over n1 (outputs k1 = 0, 1, 2;
// 3 and 4 are their conjugates), the twiddles W_25^(n2 k1), then over n2 one real 5-point transform (k1 = 0: bins 0, 5,
// 10) and two complex ones (k1 = 1: bins 1, 6, 11 and, conjugated, 9, 4; k1 = 2: bins 2, 7, 12 and 8, 3).  188
// instructions where the direct evaluation below takes 324.  c1/s1, c2/s2: cos / sin of 72 and 144 degrees.
#ifndef PDS_RDFT25_MIXED
#define PDS_RDFT25_MIXED 1
#endif
PDS_HD void rdft25(const float *a, float &even_sum, float &odd_sum, float *Ar, float *Ai) {
  constexpr float c1 = Tw600<5, 1>::re, c2 = Tw600<5, 2>::re, s1 = -Tw600<5, 1>::im, s2 = -Tw600<5, 2>::im;
  float y0[5], y1r[5], y1i[5], y2r[5], y2i[5];
  static_for<0, 5>([&](auto nn) {
    constexpr int n2 = decltype(nn)::value;
    const float x0 = a[n2], x1 = a[5 + n2], x2 = a[10 + n2], x3 = a[15 + n2], x4 = a[20 + n2];
    const float p1 = x1 + x4, p2 = x2 + x3, d1 = x1 - x4, d2 = x2 - x3;
    y0[n2] = x0 + p1 + p2;
    const float r1 = fmaf(c2, p2, fmaf(c1, p1, x0)), i1 = -fmaf(s2, d2, s1 * d1);
    const float r2 = fmaf(c1, p2, fmaf(c2, p1, x0)), i2 = fmaf(s1, d2, -(s2 * d1));
    if constexpr (n2 == 0) {
      y1r[0] = r1, y1i[0] = i1, y2r[0] = r2, y2i[0] = i2;
    } else {  // times W_25^(n2 k1)
      constexpr float w1r = Tw600<25, n2>::re, w1i = Tw600<25, n2>::im, w2r = Tw600<25, 2 * n2>::re, w2i = Tw600<25, 2 * n2>::im;
      y1r[n2] = fmaf(r1, w1r, -(i1 * w1i));
      y1i[n2] = fmaf(r1, w1i, i1 * w1r);
      y2r[n2] = fmaf(r2, w2r, -(i2 * w2i));
      y2i[n2] = fmaf(r2, w2i, i2 * w2r);
    }
  });
  {  // k1 = 0: real inputs, bins 0, 5, 10
    const float p1 = y0[1] + y0[4], p2 = y0[2] + y0[3], d1 = y0[1] - y0[4], d2 = y0[2] - y0[3];
    even_sum = y0[0] + p1 + p2;
    odd_sum = 0.0f;
    Ar[5] = fmaf(c2, p2, fmaf(c1, p1, y0[0]));
    Ai[5] = -fmaf(s2, d2, s1 * d1);
    Ar[10] = fmaf(c1, p2, fmaf(c2, p1, y0[0]));
    Ai[10] = fmaf(s1, d2, -(s2 * d1));
  }
  auto cdft5 = [&](const float *tr, const float *ti, float *zr, float *zi) {
    const float p1r = tr[1] + tr[4], p1i = ti[1] + ti[4], p2r = tr[2] + tr[3], p2i = ti[2] + ti[3];
    const float d1r = tr[1] - tr[4], d1i = ti[1] - ti[4], d2r = tr[2] - tr[3], d2i = ti[2] - ti[3];
    zr[0] = tr[0] + p1r + p2r;
    zi[0] = ti[0] + p1i + p2i;
    const float a1r = fmaf(c2, p2r, fmaf(c1, p1r, tr[0])), a1i = fmaf(c2, p2i, fmaf(c1, p1i, ti[0]));
    const float a2r = fmaf(c1, p2r, fmaf(c2, p1r, tr[0])), a2i = fmaf(c1, p2i, fmaf(c2, p1i, ti[0]));
    const float b1r = fmaf(s2, d2r, s1 * d1r), b1i = fmaf(s2, d2i, s1 * d1i);
    const float b2r = fmaf(-s1, d2r, s2 * d1r), b2i = fmaf(-s1, d2i, s2 * d1i);
    zr[1] = a1r + b1i, zi[1] = a1i - b1r;  // A1 - i B1
    zr[4] = a1r - b1i, zi[4] = a1i + b1r;  // A1 + i B1
    zr[2] = a2r + b2i, zi[2] = a2i - b2r;
    zr[3] = a2r - b2i, zi[3] = a2i + b2r;
  };
  float zr[5], zi[5];
  cdft5(y1r, y1i, zr, zi);  // bins 1, 6, 11, 16 (= conj of 9), 21 (= conj of 4)
  Ar[1] = zr[0], Ai[1] = zi[0], Ar[6] = zr[1], Ai[6] = zi[1], Ar[11] = zr[2], Ai[11] = zi[2];
  Ar[9] = zr[3], Ai[9] = -zi[3], Ar[4] = zr[4], Ai[4] = -zi[4];
  cdft5(y2r, y2i, zr, zi);  // bins 2, 7, 12, 17 (= conj of 8), 22 (= conj of 3)
  Ar[2] = zr[0], Ai[2] = zi[0], Ar[7] = zr[1], Ai[7] = zi[1], Ar[12] = zr[2], Ai[12] = zi[2];
  Ar[8] = zr[3], Ai[8] = -zi[3], Ar[3] = zr[4], Ai[3] = -zi[4];
}

// 5-point building blocks of the mixed-radix real transforms below (c1/s1, c2/s2: cos / sin of 72 and 144 degrees;
// outputs Z[k] = sum_j t[j] e^(-2 pi i j k / 5))
struct Dft5 {
  static constexpr float c1 = Tw600<5, 1>::re, c2 = Tw600<5, 2>::re, s1 = -Tw600<5, 1>::im, s2 = -Tw600<5, 2>::im;
  // real inputs: Z[0] (real), Z[1], Z[2]
  static PDS_HD void real(const float *y, float &z0, float &z1r, float &z1i, float &z2r, float &z2i) {
    const float p1 = y[1] + y[4], p2 = y[2] + y[3], d1 = y[1] - y[4], d2 = y[2] - y[3];
    z0 = y[0] + p1 + p2;
    z1r = fmaf(c2, p2, fmaf(c1, p1, y[0]));
    z1i = -fmaf(s2, d2, s1 * d1);
    z2r = fmaf(c1, p2, fmaf(c2, p1, y[0]));
    z2i = fmaf(s1, d2, -(s2 * d1));
  }
  // complex inputs, the first NOUT outputs (5: all; 2: Z[0], Z[1])
  template <int NOUT>
  static PDS_HD void cplx(const float *tr, const float *ti, float *zr, float *zi) {
    const float p1r = tr[1] + tr[4], p1i = ti[1] + ti[4], p2r = tr[2] + tr[3], p2i = ti[2] + ti[3];
    const float d1r = tr[1] - tr[4], d1i = ti[1] - ti[4], d2r = tr[2] - tr[3], d2i = ti[2] - ti[3];
    zr[0] = tr[0] + p1r + p2r;
    zi[0] = ti[0] + p1i + p2i;
    const float a1r = fmaf(c2, p2r, fmaf(c1, p1r, tr[0])), a1i = fmaf(c2, p2i, fmaf(c1, p1i, ti[0]));
    const float b1r = fmaf(s2, d2r, s1 * d1r), b1i = fmaf(s2, d2i, s1 * d1i);
    zr[1] = a1r + b1i, zi[1] = a1i - b1r;  // A1 - i B1
    if constexpr (NOUT > 2) {
      const float a2r = fmaf(c1, p2r, fmaf(c2, p1r, tr[0])), a2i = fmaf(c1, p2i, fmaf(c2, p1i, ti[0]));
      const float b2r = fmaf(-s1, d2r, s2 * d1r), b2i = fmaf(-s1, d2i, s2 * d1i);
      zr[4] = a1r - b1i, zi[4] = a1i + b1r;  // A1 + i B1
      zr[2] = a2r + b2i, zi[2] = a2i - b2r;
      zr[3] = a2r - b2i, zi[3] = a2i + b2r;
    }
  }
};

// 20 real points as 4 x 5 (n = 5 n1 + n2, k = k1 + 4 k2): five real 4-point transforms over n1 (k1 = 0, 1, 2; 3 is the
// conjugate of 1), the twiddles W_20^(n2 k1), then over n2 a real 5-point transform (k1 = 0: bins 0, 4, 8), a complex
// one (k1 = 1: bins 1, 5, 9 and, conjugated, 7, 3) and the first two outputs of another (k1 = 2: bins 2, 6; bin 10 is
// the alternating sum).  ~140 instructions where the direct evaluation takes ~210.
PDS_HD void rdft20(const float *a, float &even_sum, float &odd_sum, float *Ar, float *Ai) {
  float y0[5], y1r[5], y1i[5], y2r[5], y2i[5], alt = 0.0f;
  static_for<0, 5>([&](auto nn) {
    constexpr int n2 = decltype(nn)::value;
    const float x0 = a[n2], x1 = a[5 + n2], x2 = a[10 + n2], x3 = a[15 + n2];
    const float e0 = x0 + x2, e1 = x0 - x2, o0 = x1 + x3, o1 = x1 - x3;
    y0[n2] = e0 + o0;
    const float q = e0 - o0;           // k1 = 2 (real)
    alt = (n2 % 2 == 0) ? alt + q : alt - q;  // bin 10 = sum_n (-1)^n a[n]
    if constexpr (n2 == 0) {
      y1r[0] = e1, y1i[0] = -o1, y2r[0] = q, y2i[0] = 0.0f;
    } else {
      constexpr float w1r = Tw600<20, n2>::re, w1i = Tw600<20, n2>::im, w2r = Tw600<20, 2 * n2>::re, w2i = Tw600<20, 2 * n2>::im;
      y1r[n2] = fmaf(e1, w1r, o1 * w1i);   // (e1 - i o1) (w1r + i w1i)
      y1i[n2] = fmaf(e1, w1i, -(o1 * w1r));
      y2r[n2] = q * w2r;
      y2i[n2] = q * w2i;
    }
  });
  float z0;
  Dft5::real(y0, z0, Ar[4], Ai[4], Ar[8], Ai[8]);
  even_sum = 0.5f * (z0 + alt);
  odd_sum = 0.5f * (z0 - alt);
  float zr[5], zi[5];
  Dft5::cplx<5>(y1r, y1i, zr, zi);  // bins 1, 5, 9, 13 (= conj of 7), 17 (= conj of 3)
  Ar[1] = zr[0], Ai[1] = zi[0], Ar[5] = zr[1], Ai[5] = zi[1], Ar[9] = zr[2], Ai[9] = zi[2];
  Ar[7] = zr[3], Ai[7] = -zi[3], Ar[3] = zr[4], Ai[3] = -zi[4];
  Dft5::cplx<2>(y2r, y2i, zr, zi);  // bins 2, 6
  Ar[2] = zr[0], Ai[2] = zi[0], Ar[6] = zr[1], Ai[6] = zi[1];
}

// 30 real points as 6 x 5 (n = 5 n1 + n2, k = k1 + 6 k2): five real 6-point transforms over n1 (as 2 x 3; k1 = 0 .. 3,
// 4 and 5 are conjugates), the twiddles W_30^(n2 k1), then over n2 a real 5-point transform (k1 = 0: bins 0, 6, 12), two
// complex ones (k1 = 1: bins 1, 7, 13, conjugated 11, 5; k1 = 2: bins 2, 8, 14, conjugated 10, 4) and the first two
// outputs of a third (k1 = 3: bins 3, 9; bin 15 is the alternating sum).  ~260 instructions instead of ~460.
PDS_HD void rdft30(const float *a, float &even_sum, float &odd_sum, float *Ar, float *Ai) {
  constexpr float h3 = 0.86602540378443864676f;  // sin 60
  float y0[5], y1r[5], y1i[5], y2r[5], y2i[5], y3r[5], y3i[5], alt = 0.0f;
  static_for<0, 5>([&](auto nn) {
    constexpr int n2 = decltype(nn)::value;
    const float x0 = a[n2], x1 = a[5 + n2], x2 = a[10 + n2], x3 = a[15 + n2], x4 = a[20 + n2], x5 = a[25 + n2];
    // 3-point transforms of the even (x0, x2, x4) and the odd (x1, x3, x5) samples: T0 real, T1 = tr + i ti (T2 = conj T1)
    const float es = x2 + x4, ed = x2 - x4, os = x3 + x5, od = x3 - x5;
    const float e0 = x0 + es, e1r = fmaf(-0.5f, es, x0), e1i = -h3 * ed;
    const float o0 = x1 + os, o1r = fmaf(-0.5f, os, x1), o1i = -h3 * od;
    y0[n2] = e0 + o0;
    const float q = e0 - o0;  // k1 = 3 (real)
    alt = (n2 % 2 == 0) ? alt + q : alt - q;  // bin 15 = sum_n (-1)^n a[n]
    // k1 = 1: E1 + W6 O1, W6 = 1/2 - i h3;  k1 = 2: conj(E1) + W3 conj(O1), W3 = -1/2 - i h3
    const float r1 = fmaf(h3, o1i, fmaf(0.5f, o1r, e1r)), i1 = fmaf(-h3, o1r, fmaf(0.5f, o1i, e1i));
    const float r2 = fmaf(-h3, o1i, fmaf(-0.5f, o1r, e1r)), i2 = fmaf(-h3, o1r, fmaf(0.5f, o1i, -e1i));
    if constexpr (n2 == 0) {
      y1r[0] = r1, y1i[0] = i1, y2r[0] = r2, y2i[0] = i2, y3r[0] = q, y3i[0] = 0.0f;
    } else {
      constexpr float w1r = Tw600<30, n2>::re, w1i = Tw600<30, n2>::im, w2r = Tw600<30, 2 * n2>::re, w2i = Tw600<30, 2 * n2>::im;
      constexpr float w3r = Tw600<30, 3 * n2>::re, w3i = Tw600<30, 3 * n2>::im;
      y1r[n2] = fmaf(r1, w1r, -(i1 * w1i));
      y1i[n2] = fmaf(r1, w1i, i1 * w1r);
      y2r[n2] = fmaf(r2, w2r, -(i2 * w2i));
      y2i[n2] = fmaf(r2, w2i, i2 * w2r);
      y3r[n2] = q * w3r;
      y3i[n2] = q * w3i;
    }
  });
  float z0;
  Dft5::real(y0, z0, Ar[6], Ai[6], Ar[12], Ai[12]);
  even_sum = 0.5f * (z0 + alt);
  odd_sum = 0.5f * (z0 - alt);
  float zr[5], zi[5];
  Dft5::cplx<5>(y1r, y1i, zr, zi);  // bins 1, 7, 13, 19 (= conj of 11), 25 (= conj of 5)
  Ar[1] = zr[0], Ai[1] = zi[0], Ar[7] = zr[1], Ai[7] = zi[1], Ar[13] = zr[2], Ai[13] = zi[2];
  Ar[11] = zr[3], Ai[11] = -zi[3], Ar[5] = zr[4], Ai[5] = -zi[4];
  Dft5::cplx<5>(y2r, y2i, zr, zi);  // bins 2, 8, 14, 20 (= conj of 10), 26 (= conj of 4)
  Ar[2] = zr[0], Ai[2] = zi[0], Ar[8] = zr[1], Ai[8] = zi[1], Ar[14] = zr[2], Ai[14] = zi[2];
  Ar[10] = zr[3], Ai[10] = -zi[3], Ar[4] = zr[4], Ai[4] = -zi[4];
  Dft5::cplx<2>(y3r, y3i, zr, zi);  // bins 3, 9
  Ar[3] = zr[0], Ai[3] = zi[0], Ar[9] = zr[1], Ai[9] = zi[1];
}

template <int M>
PDS_HD void rdft_direct(const float *a, float &even_sum, float &odd_sum, float *Ar, float *Ai) {
#if PDS_RDFT25_MIXED
  if constexpr (M == 25) {
    rdft25(a, even_sum, odd_sum, Ar, Ai);
    return;
  } else if constexpr (M == 20) {
    rdft20(a, even_sum, odd_sum, Ar, Ai);
    return;
  } else if constexpr (M == 30) {
    rdft30(a, even_sum, odd_sum, Ar, Ai);
    return;
  }
#endif
  constexpr int J = (M - 1) / 2;
  constexpr bool EVEN = M % 2 == 0;
  float s[J + 1], d[J + 1];
  static_for<1, J + 1>([&](auto jj) {
    constexpr int j = decltype(jj)::value;
    s[j] = a[j] + a[M - j];
    d[j] = a[j] - a[M - j];
  });
  if constexpr (EVEN) {
    float ev = a[0], od = 0.0f;
    static_for<1, M>([&](auto nn) {
      constexpr int n = decltype(nn)::value;
      if constexpr (n % 2 == 0) ev += a[n]; else od += a[n];
    });
    even_sum = ev;
    odd_sum = od;
  } else {
    float tot = a[0];
    static_for<1, J + 1>([&](auto jj) { tot += s[decltype(jj)::value]; });
    even_sum = tot;
    odd_sum = 0.0f;
  }
  static_for<1, J + 1>([&](auto kk) {
    constexpr int k = decltype(kk)::value;
    float re = a[0], im = 0.0f;
    if constexpr (EVEN) re += (k % 2 == 0) ? a[M / 2] : -a[M / 2];
    static_for<1, J + 1>([&](auto jj) {
      constexpr int j = decltype(jj)::value;
      re += s[j] * Tw600<M, j * k>::re;
      im += d[j] * Tw600<M, j * k>::im;
    });
    Ar[k] = re;
    Ai[k] = im;
  });
}

// DFT of M real points stored as z[m] = c[2m] + i c[2m+1] ALREADY transformed: given
// Y = FFT_{M/2}(z), writes |A[m]|^2 (or |A[m]|) for m = 0 .. M/2, UNSCALED.
template <int M, typename Emit>
PDS_HD void rdft_finish_power(const float *Yr, const float *Yi, Emit &&emit) {
  constexpr int H = M / 2;
  {
    const float a0 = Yr[0] + Yi[0], aH = Yr[0] - Yi[0];
    emit(Int<0>{}, a0, 0.0f);
    emit(Int<H>{}, aH, 0.0f);
  }
  if constexpr (H >= 2) emit(Int<H / 2>{}, Yr[H / 2], -Yi[H / 2]);
  static_for<1, H / 2>([&](auto kk) {
    constexpr int k = decltype(kk)::value;
    const float sr = Yr[k] + Yr[H - k], si = Yi[k] - Yi[H - k];
    const float dr = Yr[k] - Yr[H - k], di = Yi[k] + Yi[H - k];
    constexpr float wr = Tw<M, k>::re, wi = Tw<M, k>::im;
    const float tr = wr * di + wi * dr;
    const float ti = wi * di - wr * dr;
    emit(Int<k>{}, 0.5f * (sr + tr), 0.5f * (si + ti));
    emit(Int<H - k>{}, 0.5f * (sr - tr), 0.5f * (ti - si));
  });
}

}  // namespace inl
}  // namespace pds
