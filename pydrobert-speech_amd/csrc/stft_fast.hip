// Fused STFT filter-bank kernel for power-of-two DFT sizes (float32), gfx950.
//
// One persistent 512-thread workgroup per CU walks batches of 64 consecutive frames of one
// utterance.  Per batch:
//
//  phase 1 (FFT, 16 lanes per frame, everything in registers except ONE LDS exchange)
//    N = N1 * N2.  Lane n2 of a frame's lane group loads the N1 samples x[N2*n1 + n2]
//    straight from global memory (symmetric reflection resolved in the index), windows them
//    and runs an in-lane REAL DFT of size N1 (fft_inlane.h).  Its outputs k1 = 1..N1/2-1 are
//    multiplied by the per-lane twiddles W_N^(n2*k1) and written to the wave's private LDS
//    exchange area, transposed: lane k1 then reads column k1 (N2 complex values), runs an
//    in-lane complex FFT of size N2 and holds bins k1 + N1*k2, k2 = 0..N2-1.  Bins beyond
//    N/2 are the mirror images of bins below it, and only |X|^2 is needed, so nothing is
//    wasted: the (N1/2-1)*N2 column bins plus the N2+1 bins that are multiples of N1/2 (a
//    real DFT of the per-lane even/odd sums, done by lane 0 of the group) are exactly the
//    N/2+1 half-spectrum bins.  |X|^2 goes to the batch's power buffer P[bin][frame] in LDS.
//  phase 2 (filter bank, lane = frame)
//    Each wave takes a share of the filters; a filter's weights and bin offsets are
//    wave-uniform, so they arrive through the scalar cache and the inner loop is one
//    ds_read + one v_fmac per tap.  log() and the energy column are applied here and the
//    64 x C result tile is staged in LDS.
//  phase 3: the tile is copied to global memory with coalesced stores.
//
// Reference semantics: compute_full framing (compute.py:574-607) and _compute_frame
// (compute.py:388-460), float32 arithmetic (the north star's 1e-4 tolerance).
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "fft_inlane.h"
#include "pds_internal.h"

namespace pds {

struct FastParams {
  const float *sig;
  const int64_t *offsets, *lengths, *nframes, *row_off;
  float *out;
  int64_t out_stride;
  const float *win_lane;    // [N2][N1]  window[N2*n1 + n2], zero beyond L
  const float2 *tw_lane;    // [N2][N1/2] W_N^(n2*k1), pre-scaled (see rdft_scaled)
  const int32_t *f_rowptr;  // [F + 1]
  const int32_t *f_order;   // [F] filters, longest row first
  const int32_t *t_off;     // [nnz] bin * FS
  const float *t_w;         // [nnz]
  int L, S, pad_left, F, include_energy, use_power, use_log;
  float log_floor, inv_L;
  int tiles_per_utt, n_items;
  // wave-independent variant (stft_wave_kernel): filter table in ELL form, see below
  const float *ell_w;      // per slot: [N2][len + 4] dense weight rows (row r = lane r's filter)
  const int32_t *ell_meta; // [ell_slots][N2] first bin of the row | (filter + 1) << 16
  const int32_t *ell_len;  // [ell_slots] row length in bins (multiple of 8)
  const int32_t *ell_woff; // [ell_slots] start of the slot's rows inside ell_w (floats)
  int ell_wfloats, ell_slots;
  int chunks_per_utt, num_utts;
};

template <int N1, int N2, int WAVES>
struct FastGeom {
  static constexpr int N = N1 * N2;
  static constexpr int H1 = N1 / 2;          // step-1 outputs k1 = 0..H1
  static constexpr int CPL = H1 / N2;        // step-3 columns per lane
  static constexpr int GROUPS = 64 / N2;     // frames per wave iteration
  static constexpr int GPH = 32 / N2;        // lane groups per 32-lane half
  static constexpr int FPB = 64;             // frames per batch (= lanes in phase 2)
  static constexpr int ITERS = FPB / (WAVES * GROUPS);
  static constexpr int NB = N / 2 + 1;       // half-spectrum bins
  static constexpr int NSLOT = NB + 1;       // + energy
  static constexpr int FS = FPB + 1;         // P row stride (floats): conflict-free both ways
  static constexpr int RS = N2 + 2;          // exchange row stride (float2)
  static constexpr int P_FLOATS = (NSLOT * FS + 3) / 4 * 4;
  static constexpr int EXCH_F2_PER_WAVE = GROUPS * H1 * RS;
  static constexpr int STAGE_FLOATS = WAVES * EXCH_F2_PER_WAVE * 2;
  static constexpr size_t SMEM_BYTES = (size_t)(P_FLOATS + STAGE_FLOATS) * 4;
  static_assert(H1 % N2 == 0 && CPL >= 1, "columns must split evenly over the lane group");
  static_assert(FPB % (WAVES * GROUPS) == 0 && ITERS >= 1, "batch must split evenly over waves");
  static_assert(N2 <= 32 && (RS * 8) % 16 == 0, "exchange rows must stay 16-byte aligned");
  static_assert(EXCH_F2_PER_WAVE * 2 >= 64 * N1, "edge-frame gather reuses the exchange area");
};

// v_mul_legacy_f32: IEEE multiply except that 0 * x = 0 for every x (NaN and Inf included)
__device__ __forceinline__ float mul_legacy(float x, float y) {
  float z;
  asm("v_mul_legacy_f32 %0, %1, %2" : "=v"(z) : "v"(x), "v"(y));
  return z;
}

// NROWS = rows of N2 samples a frame spans, ceil(L / N2), a compile-time constant so that the
// zero-padded tail of the DFT input is literal zeros (the in-lane FFT is pruned accordingly).
template <int N1, int N2, int WAVES, int NROWS>
__global__ __launch_bounds__(WAVES * 64) void stft_fast_kernel(const FastParams p) {
  using G = FastGeom<N1, N2, WAVES>;
  constexpr int N = G::N, H1 = G::H1, FS = G::FS, RS = G::RS, NB = G::NB;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float *P = smem;
  float *stage = smem + G::P_FLOATS;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int g = lane / N2, r = lane % N2;
  float2 *exch = reinterpret_cast<float2 *>(stage) + wave * G::EXCH_F2_PER_WAVE + g * H1 * RS;

  // per-lane constants, loop invariant: window slice and inter-stage twiddles
  float win[NROWS];
  float twr[H1], twi[H1];
#pragma unroll
  for (int n1 = 0; n1 < NROWS; ++n1) win[n1] = p.win_lane[r * N1 + n1];
#pragma unroll
  for (int k1 = 1; k1 < H1; ++k1) {
    const float2 t = p.tw_lane[r * H1 + k1];
    twr[k1] = t.x;
    twi[k1] = t.y;
  }
  const int L = p.L, S = p.S;
  const bool use_power = p.use_power != 0;

  for (int item = blockIdx.x; item < p.n_items; item += gridDim.x) {
    const int b = item / p.tiles_per_utt;
    const int64_t t0 = (int64_t)(item - b * p.tiles_per_utt) * G::FPB;
    const int64_t nfr = p.nframes[b];
    if (t0 >= nfr) continue;  // uniform
    const int n = (int)p.lengths[b];
    const float *x = p.sig + p.offsets[b];

    // ------------------------------------------------------------ phase 1: FFT ------
#pragma unroll 1
    for (int it = 0; it < G::ITERS; ++it) {
      const int fr = (g % G::GPH) * N2 + (g / G::GPH) * 32 + wave * G::ITERS + it;
      const bool valid = t0 + fr < nfr;
      if (!__any(valid)) continue;  // uniform: all of this wave's frames lie past the end
      // lanes of a frame past the end recompute the last frame; their rows are never stored
      const int64_t t = valid ? t0 + fr : nfr - 1;
      const int start = (int)(t * S) - p.pad_left;
      // 0: every row read lies inside the signal, 1: one bounce suffices, 2: general reflection
      int mode = 0;
      if (start < 0 || start + NROWS * N2 > n) mode = 1;
      if (start < -n || start + L > 2 * n) mode = 2;
      const int wmode = __builtin_amdgcn_readfirstlane(
          __any(mode == 2) ? 2 : (__any(mode == 1) ? 1 : 0));

      float a[N1];
      float energy = 0.0f;
      if (wmode == 0) {
        // interior frames: one 64-bit lane pointer + immediate offsets, no index math, no
        // predicates.  Lanes past the frame's end in the last row read samples that belong to
        // the next frame; the window (exactly 0 there, applied with the 0 * x = 0 multiply
        // below) removes them.
        const float *xp = x + (start + r);
#pragma unroll
        for (int n1 = 0; n1 < NROWS; ++n1) a[n1] = xp[n1 * N2];
      } else {
        // a frame of this wave touches a signal end: gather through LDS with a rolled loop
        float *tmp = reinterpret_cast<float *>(stage) + wave * (G::EXCH_F2_PER_WAVE * 2);
#pragma unroll 1
        for (int n1 = 0; n1 < NROWS; ++n1) {
          const int idx = n1 * N2 + r;
          float v = 0.0f;
          if (idx < L) {
            int i = start + idx;
            if (wmode == 1) {
              i = i < 0 ? -1 - i : (i >= n ? 2 * n - 1 - i : i);
            } else {
              i = (int)reflect_index((int64_t)i, (int64_t)n);
            }
            v = x[i];
          }
          tmp[n1 * 64 + lane] = v;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int n1 = 0; n1 < NROWS; ++n1) a[n1] = tmp[n1 * 64 + lane];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      }
      if (p.include_energy) {
#pragma unroll
        for (int n1 = 0; n1 < NROWS; ++n1) {
          const float v = (n1 * N2 + r < L) ? a[n1] : 0.0f;
          energy = fmaf(v, v, energy);
        }
      }
      // v_mul_legacy_f32: 0 * anything = 0, so samples under a zero of the window (the
      // padding lanes above included) never leak a NaN/Inf into the frame
#pragma unroll
      for (int n1 = 0; n1 < NROWS; ++n1) a[n1] = mul_legacy(a[n1], win[n1]);
#pragma unroll
      for (int n1 = NROWS; n1 < N1; ++n1) a[n1] = 0.0f;

      float even_sum, odd_sum, Ar[H1], Ai[H1];
      inl::rdft_scaled<N1>(a, even_sum, odd_sum, Ar, Ai);

      // transpose through LDS: row k1 of this frame's block holds column k1 for all n2
      {
        float *row0 = reinterpret_cast<float *>(exch);
        row0[r] = even_sum;        // c[n2]      = sum of even-indexed samples
        row0[N2 + r] = odd_sum;    // c[n2 + N2] = sum of odd-indexed samples
      }
#pragma unroll
      for (int k1 = 1; k1 < H1; ++k1) {
        float2 v;
        v.x = Ar[k1] * twr[k1] - Ai[k1] * twi[k1];
        v.y = Ar[k1] * twi[k1] + Ai[k1] * twr[k1];
        exch[k1 * RS + r] = v;
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

      float *Pf = P + fr;
#pragma unroll
      for (int q = 0; q < G::CPL; ++q) {
        const int kk = q * N2 + r;
        const float4 *row = reinterpret_cast<const float4 *>(exch + kk * RS);
        float zr[N2], zi[N2], Yr[N2], Yi[N2];
#pragma unroll
        for (int j = 0; j < N2 / 2; ++j) {
          const float4 v = row[j];
          zr[2 * j] = v.x;
          zi[2 * j] = v.y;
          zr[2 * j + 1] = v.z;
          zi[2 * j + 1] = v.w;
        }
        inl::CFFT<N2, 1>::run(zr, zi, Yr, Yi);
        if (q == 0 && r == 0) {
          // bins that are multiples of N1/2: real DFT of the 2*N2 even/odd sums
          float pw[N2 + 1];
          inl::rdft_finish_power<2 * N2>(Yr, Yi, [&](auto mm, float re, float im) {
            pw[decltype(mm)::value] = re * re + im * im;
          });
          if (!use_power) {
#pragma unroll
            for (int m = 0; m <= N2; ++m) pw[m] = __builtin_amdgcn_sqrtf(pw[m]);
          }
#pragma unroll
          for (int m = 0; m <= N2; ++m) Pf[(m * H1) * FS] = pw[m];
        } else {
          float pw[N2];
#pragma unroll
          for (int k2 = 0; k2 < N2; ++k2) pw[k2] = Yr[k2] * Yr[k2] + Yi[k2] * Yi[k2];
          if (!use_power) {
#pragma unroll
            for (int k2 = 0; k2 < N2; ++k2) pw[k2] = __builtin_amdgcn_sqrtf(pw[k2]);
          }
#pragma unroll
          for (int k2 = 0; k2 < N2; ++k2) {
            // bin kk + N1*k2, or its mirror image when beyond N/2
            const int bin = (k2 < N2 / 2) ? kk + N1 * k2 : N - kk - N1 * k2;
            Pf[bin * FS] = pw[k2];
          }
        }
      }
      if (p.include_energy) {
#pragma unroll
        for (int off = N2 / 2; off >= 1; off >>= 1) energy += __shfl_xor(energy, off, 64);
        if (r == 0) Pf[NB * FS] = energy;
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    __syncthreads();

    // ------------------------------------------------------------ phase 2: filters ---
    const int C = p.F + (p.include_energy ? 1 : 0);
    const int CP = C | 1;  // odd row stride: conflict-free for lane = frame
    const int col0 = p.include_energy ? 1 : 0;
    const float *Pl = P + lane;
    for (int fi = wave; fi < p.F; fi += WAVES) {
      const int f = p.f_order[fi];
      const int q1 = p.f_rowptr[f + 1];
      float acc = 0.0f;
      for (int q = p.f_rowptr[f]; q < q1; ++q) acc = fmaf(p.t_w[q], Pl[p.t_off[q]], acc);
      if (p.use_log) acc = __logf(fmaxf(acc, p.log_floor));
      stage[lane * CP + col0 + f] = acc;
    }
    if (p.include_energy && wave == WAVES - 1) {
      float e = Pl[NB * FS] * p.inv_L;
      if (!use_power) e = __builtin_amdgcn_sqrtf(e);
      if (p.use_log) e = __logf(fmaxf(e, p.log_floor));
      stage[lane * CP] = e;
    }
    __syncthreads();

    // ------------------------------------------------------------ phase 3: store -----
    {
      const int64_t left = nfr - t0;
      const int rows = left < G::FPB ? (int)left : G::FPB;
      const int total = rows * C;
      const float invC = 1.0f / (float)C;
      float *dst = p.out + (p.row_off[b] + t0) * p.out_stride;
      for (int e = threadIdx.x; e < total; e += WAVES * 64) {
        const int row = (int)(((float)e + 0.5f) * invC);
        const int c = e - row * C;
        dst[(int64_t)row * p.out_stride + c] = stage[row * CP + c];
      }
    }
    __syncthreads();
  }
}

// ====================================================================================
// Variant 2: every wavefront works alone (no workgroup barrier, no batch buffer).
//
// A wave takes 64/N2 consecutive frames of one utterance, runs the same two-step FFT as
// above, leaves |X|^2 of its frames in its own LDS area (P[frame][bin], aliased over the
// exchange area) and integrates the filters itself: lane (frame g, j) owns every N2-th filter
// of the length-sorted filter list ("ELL" layout: slot s of lane j is filter order[s*N2+j];
// all lanes walk slot s for the same number of steps, shorter rows padded with zero weights).
// 9 KB of LDS per wave and <= 168 VGPRs give 12 resident waves per CU instead of 8, and no
// wave ever waits for another.
// ====================================================================================
template <int N1, int N2, int WAVES, int NROWS>
struct WaveGeom {
  static constexpr int N = N1 * N2;
  static constexpr int H1 = N1 / 2;
  static constexpr int CPL = H1 / N2;
  static constexpr int GROUPS = 64 / N2;
  static constexpr int NB = N / 2 + 1;
  static constexpr int RS = N2 + 2;
  static constexpr int EXCH_F2 = GROUPS * H1 * RS;  // float2 per wave
  // P row stride: >= NB + 1 (energy) and = 16 (mod 32) so that the two frames sharing a
  // 32-lane half write to disjoint banks
  static constexpr int PSTR = ((NB + 1 + 15) / 32) * 32 + 16;
  static_assert(GROUPS * PSTR <= EXCH_F2 * 2, "P must fit over the exchange area");
  static_assert(EXCH_F2 * 2 >= 64 * NROWS, "edge-frame gather reuses the exchange area");
  static_assert(H1 % N2 == 0 && CPL >= 1 && N2 <= 32 && (RS * 8) % 16 == 0, "geometry");
};

// tuning knobs (workgroup size in waves, resident waves per SIMD the register budget is for)
#ifndef PDS_WAVE_WAVES
#define PDS_WAVE_WAVES 8
#endif
#ifndef PDS_WAVE_MINW
#define PDS_WAVE_MINW 4
#endif

template <int N1, int N2, int WAVES, int NROWS>
__global__ __launch_bounds__(WAVES * 64, PDS_WAVE_MINW) void stft_wave_kernel(const FastParams p) {
  using G = WaveGeom<N1, N2, WAVES, NROWS>;
  constexpr int N = G::N, H1 = G::H1, RS = G::RS, NB = G::NB, PSTR = G::PSTR;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int g = lane / N2, r = lane % N2;
  float *wbase = smem + wave * (G::EXCH_F2 * 2);
  float2 *exch = reinterpret_cast<float2 *>(wbase) + g * H1 * RS;
  float *Pg = wbase + g * PSTR;
  // filter table -> LDS once per workgroup (read by every wave, every iteration): per-slot
  // weight rows, then per (slot, lane) the row's first bin and filter index
  float *ellw = smem + WAVES * (G::EXCH_F2 * 2);
  int *ell_meta = reinterpret_cast<int *>(ellw + p.ell_wfloats);
  for (int i = threadIdx.x; i < p.ell_wfloats; i += WAVES * 64) ellw[i] = p.ell_w[i];
  for (int i = threadIdx.x; i < p.ell_slots * N2; i += WAVES * 64) ell_meta[i] = p.ell_meta[i];
  // the wave areas start out zeroed so that never-written P padding is finite
  for (int i = threadIdx.x; i < WAVES * G::EXCH_F2 * 2; i += WAVES * 64) smem[i] = 0.0f;
  __syncthreads();

  float win[NROWS];
  float twr[H1], twi[H1];
#pragma unroll
  for (int n1 = 0; n1 < NROWS; ++n1) win[n1] = p.win_lane[r * N1 + n1];
#pragma unroll
  for (int k1 = 1; k1 < H1; ++k1) {
    const float2 t = p.tw_lane[r * H1 + k1];
    twr[k1] = t.x;
    twi[k1] = t.y;
  }
  const int L = p.L, S = p.S;
  const bool use_power = p.use_power != 0;
  const int col0 = p.include_energy ? 1 : 0;

  // work items: (utterance, chunk of GROUPS consecutive frames); the waves of a workgroup
  // take neighbouring chunks so that overlapping samples are shared through the CU's L1
  const int stride = gridDim.x * WAVES;
  int b = 0;
  int chunk = blockIdx.x * WAVES + wave;
  while (chunk >= p.chunks_per_utt) {
    chunk -= p.chunks_per_utt;
    ++b;
  }
  for (; b < p.num_utts; chunk += stride) {
    while (chunk >= p.chunks_per_utt) {
      chunk -= p.chunks_per_utt;
      ++b;
    }
    if (b >= p.num_utts) break;
    const int64_t nfr = p.nframes[b];
    const int64_t tb = (int64_t)chunk * G::GROUPS;
    if (tb >= nfr) continue;  // uniform
    const int n = (int)p.lengths[b];
    const float *x = p.sig + p.offsets[b];
    const bool valid = tb + g < nfr;
    const int64_t t = valid ? tb + g : nfr - 1;
    const int start = (int)(t * S) - p.pad_left;
    int mode = 0;
    if (start < 0 || start + NROWS * N2 > n) mode = 1;
    if (start < -n || start + L > 2 * n) mode = 2;
    const int wmode = __builtin_amdgcn_readfirstlane(
        __any(mode == 2) ? 2 : (__any(mode == 1) ? 1 : 0));

    float a[N1];
    float energy = 0.0f;
    if (wmode == 0) {
      const float *xp = x + (start + r);
#pragma unroll
      for (int n1 = 0; n1 < NROWS; ++n1) a[n1] = xp[n1 * N2];
    } else {
      float *tmp = wbase;
#pragma unroll 1
      for (int n1 = 0; n1 < NROWS; ++n1) {
        const int idx = n1 * N2 + r;
        float v = 0.0f;
        if (idx < L) {
          int i = start + idx;
          if (wmode == 1) {
            i = i < 0 ? -1 - i : (i >= n ? 2 * n - 1 - i : i);
          } else {
            i = (int)reflect_index((int64_t)i, (int64_t)n);
          }
          v = x[i];
        }
        tmp[n1 * 64 + lane] = v;
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
      for (int n1 = 0; n1 < NROWS; ++n1) a[n1] = tmp[n1 * 64 + lane];
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    if (p.include_energy) {
#pragma unroll
      for (int n1 = 0; n1 < NROWS; ++n1) {
        const float v = (n1 * N2 + r < L) ? a[n1] : 0.0f;
        energy = fmaf(v, v, energy);
      }
    }
#pragma unroll
    for (int n1 = 0; n1 < NROWS; ++n1) a[n1] = mul_legacy(a[n1], win[n1]);
#pragma unroll
    for (int n1 = NROWS; n1 < N1; ++n1) a[n1] = 0.0f;

    float even_sum, odd_sum, Ar[H1], Ai[H1];
    inl::rdft_scaled<N1>(a, even_sum, odd_sum, Ar, Ai);
    {
      float *row0 = reinterpret_cast<float *>(exch);
      row0[r] = even_sum;
      row0[N2 + r] = odd_sum;
    }
#pragma unroll
    for (int k1 = 1; k1 < H1; ++k1) {
      float2 v;
      v.x = Ar[k1] * twr[k1] - Ai[k1] * twi[k1];
      v.y = Ar[k1] * twi[k1] + Ai[k1] * twr[k1];
      exch[k1 * RS + r] = v;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

    float pw[G::CPL][N2 + 1];
#pragma unroll
    for (int q = 0; q < G::CPL; ++q) {
      const int kk = q * N2 + r;
      const float4 *row = reinterpret_cast<const float4 *>(exch + kk * RS);
      float zr[N2], zi[N2], Yr[N2], Yi[N2];
#pragma unroll
      for (int j = 0; j < N2 / 2; ++j) {
        const float4 v = row[j];
        zr[2 * j] = v.x;
        zi[2 * j] = v.y;
        zr[2 * j + 1] = v.z;
        zi[2 * j + 1] = v.w;
      }
      inl::CFFT<N2, 1>::run(zr, zi, Yr, Yi);
      if (q == 0 && r == 0) {
        // lane 0 transformed the packed even/odd sums: untangle to the bins m * N1/2.
        // (Spreading this over the idle lanes through LDS or DPP was measured/estimated to
        // cost more LDS time than the ~100 VALU issue slots it saves.)
        inl::rdft_finish_power<2 * N2>(Yr, Yi, [&](auto mm, float re, float im) {
          pw[q][decltype(mm)::value] = re * re + im * im;
        });
      } else {
#pragma unroll
        for (int k2 = 0; k2 < N2; ++k2) pw[q][k2] = Yr[k2] * Yr[k2] + Yi[k2] * Yi[k2];
        pw[q][N2] = 0.0f;
      }
    }
    if (!use_power) {
#pragma unroll
      for (int q = 0; q < G::CPL; ++q)
#pragma unroll
        for (int k2 = 0; k2 <= N2; ++k2) pw[q][k2] = __builtin_amdgcn_sqrtf(pw[q][k2]);
    }
    // every lane is done with the exchange area (same wave, in order): reuse it as P
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int q = 0; q < G::CPL; ++q) {
      const int kk = q * N2 + r;
      if (q == 0 && r == 0) {
#pragma unroll
        for (int m = 0; m <= N2; ++m) Pg[m * H1] = pw[q][m];
      } else {
#pragma unroll
        for (int k2 = 0; k2 < N2; ++k2) {
          const int bin = (k2 < N2 / 2) ? kk + N1 * k2 : N - kk - N1 * k2;
          Pg[bin] = pw[q][k2];
        }
      }
    }
    // slots past the last bin are read (with weight 0) by the filter walk: keep them finite
    if (r < G::PSTR - NB - 1) Pg[NB + 1 + r] = 0.0f;
    if (p.include_energy) {
#pragma unroll
      for (int off = N2 / 2; off >= 1; off >>= 1) energy += __shfl_xor(energy, off, 64);
      if (r == 0) Pg[NB] = energy;
    } else if (r == 0) {
      Pg[NB] = 0.0f;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

    // ---- filter bank: lane (g, r) integrates one filter per slot.  Rows are dense bin ranges
    // starting on a multiple of 4 bins, so weights and powers both arrive as 16-byte reads
    float *orow = p.out + (p.row_off[b] + t) * p.out_stride;
    for (int sl = 0; sl < p.ell_slots; ++sl) {
      const int len = p.ell_len[sl];                 // steps of this slot, multiple of 8
      const int wstride = len + 4;                   // floats; conflict-free row skew
      const float4 *wrow = reinterpret_cast<const float4 *>(ellw + p.ell_woff[sl] + r * wstride);
      const int meta = ell_meta[sl * N2 + r];        // first bin (multiple of 4) | filter << 16
      const float4 *prow = reinterpret_cast<const float4 *>(Pg + (meta & 0xffff));
      float acc0 = 0.0f, acc1 = 0.0f, acc2 = 0.0f, acc3 = 0.0f;
      for (int t4 = 0; t4 < len / 4; t4 += 2) {
        const float4 w0 = wrow[t4], w1 = wrow[t4 + 1];
        const float4 p0 = prow[t4], p1 = prow[t4 + 1];
        acc0 = fmaf(w0.x, p0.x, acc0);
        acc1 = fmaf(w0.y, p0.y, acc1);
        acc2 = fmaf(w0.z, p0.z, acc2);
        acc3 = fmaf(w0.w, p0.w, acc3);
        acc0 = fmaf(w1.x, p1.x, acc0);
        acc1 = fmaf(w1.y, p1.y, acc1);
        acc2 = fmaf(w1.z, p1.z, acc2);
        acc3 = fmaf(w1.w, p1.w, acc3);
      }
      float acc = (acc0 + acc1) + (acc2 + acc3);
      if (p.use_log) acc = __logf(fmaxf(acc, p.log_floor));
      const int f = (meta >> 16) - 1;
      if (valid && f >= 0) orow[col0 + f] = acc;
    }
    if (p.include_energy && r == 0) {
      float e = Pg[NB] * p.inv_L;
      if (!use_power) e = __builtin_amdgcn_sqrtf(e);
      if (p.use_log) e = __logf(fmaxf(e, p.log_floor));
      if (valid) orow[0] = e;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
}

// ----------------------------------------------------------------------- host side ---

template <int N1, int N2, int WAVES, int NROWS>
static int32_t launch_geom(const pds_stft_plan *plan, const BatchArgs &a) {
  using G = FastGeom<N1, N2, WAVES>;
  const FastTables &ft = plan->fast;
  FastParams p;
  p.sig = (const float *)a.d_signal;
  p.offsets = a.d_offsets;
  p.lengths = a.d_lengths;
  p.nframes = a.d_nframes;
  p.row_off = a.d_row_off;
  p.out = (float *)a.d_out;
  p.out_stride = a.out_stride;
  p.win_lane = ft.d_window;
  p.tw_lane = (const float2 *)ft.d_twiddle;
  p.f_rowptr = plan->d_row_ptr;
  p.f_order = ft.d_order;
  p.t_off = ft.d_toff;
  p.t_w = ft.d_wval;
  p.L = plan->d.frame_length;
  p.S = plan->d.frame_shift;
  p.pad_left = a.pad_left;
  p.F = plan->d.num_filts;
  p.include_energy = plan->d.include_energy;
  p.use_power = plan->d.use_power;
  p.use_log = plan->d.use_log;
  p.log_floor = (float)plan->d.log_floor;
  p.inv_L = 1.0f / (float)plan->d.frame_length;
  p.tiles_per_utt = (int)((a.max_frames + G::FPB - 1) / G::FPB);
  const int64_t items = (int64_t)p.tiles_per_utt * a.B;
  if (items > 0x7fffffff) {
    set_error("stft_batch: too many frame tiles in one call");
    return PDS_ERR_INVALID;
  }
  p.n_items = (int)items;
  auto kern = stft_fast_kernel<N1, N2, WAVES, NROWS>;
  static bool attr_set = false;  // per instantiation
  if (!attr_set) {
    PDS_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)G::SMEM_BYTES));
    attr_set = true;
  }
  int grid = ft.num_cus;  // one persistent workgroup per CU
  if (grid > p.n_items) grid = p.n_items;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(WAVES * 64), G::SMEM_BYTES, a.stream, p);
  PDS_HIP(hipGetLastError());
  return PDS_OK;
}

static void fill_common(FastParams &p, const pds_stft_plan *plan, const BatchArgs &a) {
  const FastTables &ft = plan->fast;
  p.sig = (const float *)a.d_signal;
  p.offsets = a.d_offsets;
  p.lengths = a.d_lengths;
  p.nframes = a.d_nframes;
  p.row_off = a.d_row_off;
  p.out = (float *)a.d_out;
  p.out_stride = a.out_stride;
  p.win_lane = ft.d_window;
  p.tw_lane = (const float2 *)ft.d_twiddle;
  p.f_rowptr = plan->d_row_ptr;
  p.f_order = ft.d_order;
  p.t_off = ft.d_toff;
  p.t_w = ft.d_wval;
  p.L = plan->d.frame_length;
  p.S = plan->d.frame_shift;
  p.pad_left = a.pad_left;
  p.F = plan->d.num_filts;
  p.include_energy = plan->d.include_energy;
  p.use_power = plan->d.use_power;
  p.use_log = plan->d.use_log;
  p.log_floor = (float)plan->d.log_floor;
  p.inv_L = 1.0f / (float)plan->d.frame_length;
  p.tiles_per_utt = 0;
  p.n_items = 0;
  p.ell_w = ft.d_ell_w;
  p.ell_meta = ft.d_ell_meta;
  p.ell_len = ft.d_ell_len;
  p.ell_woff = ft.d_ell_woff;
  p.ell_wfloats = ft.ell_wfloats;
  p.ell_slots = ft.ell_slots;
  p.chunks_per_utt = 0;
  p.num_utts = a.B;
}

template <int N1, int N2, int WAVES, int NROWS>
static int32_t launch_wave(const pds_stft_plan *plan, const BatchArgs &a) {
  using G = WaveGeom<N1, N2, WAVES, NROWS>;
  const FastTables &ft = plan->fast;
  FastParams p;
  fill_common(p, plan, a);
  const int64_t chunks = (a.max_frames + G::GROUPS - 1) / G::GROUPS;
  if (chunks * a.B > 0x7fffffff || chunks > 0x3fffffff) {
    set_error("stft_batch: too many frame chunks in one call");
    return PDS_ERR_INVALID;
  }
  p.chunks_per_utt = (int)chunks;
  const size_t smem = (size_t)WAVES * G::EXCH_F2 * 8 + (size_t)ft.ell_wfloats * 4 +
                      (size_t)ft.ell_slots * N2 * 4;
  auto kern = stft_wave_kernel<N1, N2, WAVES, NROWS>;
  static size_t attr_smem = 0;  // per instantiation
  if (smem > attr_smem) {
    PDS_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)smem));
    attr_smem = smem;
  }
  int wgs_per_cu = (int)((160 * 1024) / smem);
  if (wgs_per_cu > 4 * PDS_WAVE_MINW / WAVES) wgs_per_cu = 4 * PDS_WAVE_MINW / WAVES;
  if (wgs_per_cu < 1) wgs_per_cu = 1;
  int64_t grid = (int64_t)ft.num_cus * wgs_per_cu;
  const int64_t need = (chunks * a.B + WAVES - 1) / WAVES;
  if (grid > need) grid = need;
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(WAVES * 64), smem, a.stream, p);
  PDS_HIP(hipGetLastError());
  return PDS_OK;
}

int32_t launch_stft_fast_f32(const pds_stft_plan *plan, const BatchArgs &a) {
  // frame starts and reflected indices are formed in 32-bit arithmetic
  // instantiated row counts: the common frame lengths exactly (25 ms @ 16 kHz = 25 rows of 16,
  // 20 ms = 20, 30 ms = 30, ...); other lengths use the next larger one
#define PDS_ROWS_CASE(N1, N2, R)                                          \
  if (rows <= R)                                                          \
    return plan->fast.variant == 1 ? launch_geom<N1, N2, 8, R>(plan, a)   \
                                   : launch_wave<N1, N2, PDS_WAVE_WAVES, R>(plan, a);
  const int rows = plan->fast.rows;
  if (plan->fast.kind == 512) {
    PDS_ROWS_CASE(32, 16, 20) PDS_ROWS_CASE(32, 16, 25) PDS_ROWS_CASE(32, 16, 28)
    PDS_ROWS_CASE(32, 16, 30) PDS_ROWS_CASE(32, 16, 32)
  } else if (plan->fast.kind == 256) {
    PDS_ROWS_CASE(32, 8, 20) PDS_ROWS_CASE(32, 8, 25) PDS_ROWS_CASE(32, 8, 28)
    PDS_ROWS_CASE(32, 8, 30) PDS_ROWS_CASE(32, 8, 32)
  }
#undef PDS_ROWS_CASE
  set_error("stft_batch: no fused kernel for this plan");
  return PDS_ERR_INVALID;
}

int32_t fast_tables_create(pds_stft_plan *plan, const double *window, const int32_t *row_ptr,
                           const int32_t *col, const double *val) {
  FastTables &ft = plan->fast;
  ft.kind = 0;
  const pds_stft_desc &d = plan->d;
  int n1 = 0, n2 = 0;
  if (d.dft_size == 512) { n1 = 32; n2 = 16; }
  else if (d.dft_size == 256) { n1 = 32; n2 = 8; }
  else return PDS_OK;  // generic kernel
  const int C = d.num_filts + (d.include_energy ? 1 : 0);
  const int stage_floats = 8 * (64 / n2) * (n1 / 2) * (n2 + 2) * 2;
  if (64 * (C | 1) > stage_floats) return PDS_OK;  // output tile would not fit the staging area
  if (d.frame_length > d.dft_size) return PDS_OK;
  const int N = d.dft_size, H1 = n1 / 2, FS = 65;
  std::vector<float> win((size_t)n1 * n2, 0.0f);
  for (int r = 0; r < n2; ++r)
    for (int k = 0; k < n1; ++k) {
      const int idx = n2 * k + r;
      if (idx < d.frame_length) win[(size_t)r * n1 + k] = (float)window[idx];
    }
  std::vector<float> tw((size_t)n2 * H1 * 2, 0.0f);
  for (int r = 0; r < n2; ++r)
    for (int k1 = 1; k1 < H1; ++k1) {
      const double ang = -2.0 * M_PI * (double)((r * k1) % N) / (double)N;
      const double scale = (2 * k1 == H1) ? 1.0 : 0.5;  // undo rdft_scaled's factor
      tw[((size_t)r * H1 + k1) * 2 + 0] = (float)(scale * std::cos(ang));
      tw[((size_t)r * H1 + k1) * 2 + 1] = (float)(scale * std::sin(ang));
    }
  std::vector<int32_t> order(d.num_filts), toff(d.nnz);
  for (int f = 0; f < d.num_filts; ++f) order[f] = f;
  std::stable_sort(order.begin(), order.end(), [&](int x, int y) {
    return row_ptr[x + 1] - row_ptr[x] > row_ptr[y + 1] - row_ptr[y];
  });
  std::vector<float> wv(d.nnz);
  for (int q = 0; q < d.nnz; ++q) {
    toff[q] = col[q] * FS;
    wv[q] = (float)val[q];
  }
  // ELL form of the filter table for the wave-independent kernel: slot s of lane j is filter
  // order[s * n2 + j].  A row is the filter's dense bin range, extended down to a multiple of
  // 4 bins and up to the slot's common length (a multiple of 8); zero weights fill the rest.
  const int slots = (d.num_filts + n2 - 1) / n2;
  const int pstr = ((N / 2 + 2 + 15) / 32) * 32 + 16;  // WaveGeom::PSTR
  std::vector<int32_t> ell_meta((size_t)slots * n2, 0), ell_len(slots, 0), ell_woff(slots, 0);
  std::vector<float> ell_w;
  bool ell_ok = true;
  for (int sl = 0; sl < slots; ++sl) {
    int longest = 8;
    std::vector<int> first(n2, 0);
    for (int j = 0; j < n2 && sl * n2 + j < d.num_filts; ++j) {
      const int f = order[sl * n2 + j];
      if (row_ptr[f + 1] == row_ptr[f]) continue;
      const int c0 = col[row_ptr[f]] & ~3, c1 = col[row_ptr[f + 1] - 1];  // cols ascend
      first[j] = c0;
      longest = std::max(longest, (c1 - c0 + 1 + 7) / 8 * 8);
    }
    ell_len[sl] = longest;
    ell_woff[sl] = (int32_t)ell_w.size();
    const int wstride = longest + 4;
    ell_w.resize(ell_w.size() + (size_t)n2 * wstride, 0.0f);
    for (int j = 0; j < n2; ++j) {
      int f = -1;
      if (sl * n2 + j < d.num_filts) f = order[sl * n2 + j];
      if (first[j] + longest > pstr) {
        // keep every 16-byte read inside the frame's P row
        const int shift = (first[j] + longest - pstr + 3) / 4 * 4;
        first[j] -= shift;
        if (first[j] < 0) ell_ok = false;
      }
      ell_meta[(size_t)sl * n2 + j] = first[j] | ((f + 1) << 16);
      if (f < 0) continue;
      for (int q = row_ptr[f]; q < row_ptr[f + 1]; ++q) {
        const int t = col[q] - first[j];
        if (t < 0 || t >= longest) { ell_ok = false; continue; }
        ell_w[(size_t)ell_woff[sl] + (size_t)j * wstride + t] = (float)val[q];
      }
    }
  }
  ft.ell_wfloats = (int)ell_w.size();
  ft.ell_slots = slots;
  const char *variant = std::getenv("PDS_STFT_VARIANT");
  ft.variant = (variant && variant[0] == '1') ? 1 : 2;
  if (!ell_ok || d.num_filts > 32767 || ell_w.size() * 4 > 48 * 1024) ft.variant = 1;
  int32_t rc = PDS_OK;
  if (rc == PDS_OK) rc = upload(&ft.d_ell_w, ell_w.data(), ell_w.size());
  if (rc == PDS_OK) rc = upload(&ft.d_ell_meta, ell_meta.data(), ell_meta.size());
  if (rc == PDS_OK) rc = upload(&ft.d_ell_len, ell_len.data(), ell_len.size());
  if (rc == PDS_OK) rc = upload(&ft.d_ell_woff, ell_woff.data(), ell_woff.size());
  if (rc == PDS_OK) rc = upload(&ft.d_window, win.data(), win.size());
  if (rc == PDS_OK) rc = upload(&ft.d_twiddle, tw.data(), tw.size());
  if (rc == PDS_OK) rc = upload(&ft.d_order, order.data(), order.size());
  if (rc == PDS_OK) rc = upload(&ft.d_toff, toff.data(), toff.size());
  if (rc == PDS_OK) rc = upload(&ft.d_wval, wv.data(), wv.size());
  if (rc != PDS_OK) return rc;
  hipDeviceProp_t prop;
  PDS_HIP(hipGetDeviceProperties(&prop, plan->device));
  ft.num_cus = prop.multiProcessorCount;
  ft.n1 = n1;
  ft.n2 = n2;
  ft.rows = (d.frame_length + n2 - 1) / n2;
  ft.kind = d.dft_size;
  return PDS_OK;
}

void fast_tables_destroy(pds_stft_plan *plan) {
  FastTables &ft = plan->fast;
  (void)hipFree(ft.d_window);
  (void)hipFree(ft.d_twiddle);
  (void)hipFree(ft.d_order);
  (void)hipFree(ft.d_toff);
  (void)hipFree(ft.d_wval);
  (void)hipFree(ft.d_ell_w);
  (void)hipFree(ft.d_ell_meta);
  (void)hipFree(ft.d_ell_len);
  (void)hipFree(ft.d_ell_woff);
  ft = FastTables();
}

}  // namespace pds
