// Generic STFT filter-bank kernels: any DFT size N, float32 or float64 -- a direct DFT for any N
// (stft_generic_kernel) and, for powers of two, a radix-2 FFT in LDS (stft_fft_kernel, further down).
//
// One wavefront per frame.  The frame is reflected/windowed into LDS, every lane
// evaluates bins k = lane, lane + 64, ... as a dot product over the samples, the
// power (or magnitude) spectrum goes back to LDS and lanes walk the CSR rows of the
// bin-weight table.  The dot product runs in blocks of 16 samples: the lane fetches the 16
// twiddles W^(j k), j < 16, and the block step W^(16 k) from the table once per bin, forms the
// block's cosine and sine sums from LDS broadcasts (32 multiply-adds, no table access) and
// rotates them by a running twiddle that advances one block step at a time -- 2.75 vector
// instructions per sample and bin instead of a table gather each (8x faster), with the running
// rotation taking only L / 16 steps, so its rounding stays at a few ulps.
// O(L * N/2) per frame -- this is the always-available path
// (non-power-of-two N, float64 parity with the reference's float64 arithmetic,
// cross-check of the fused kernel), not the fast one (stft_fast.hip).
//
// Follows _compute_frame (reference compute.py:388-460) and the framing of
// compute_full (compute.py:574-607).
#include "pds_internal.h"

namespace pds {

template <typename T>
struct Tw;
template <>
struct Tw<float> {
  using type = float2;
  typedef float quad __attribute__((ext_vector_type(4)));  // 16-byte LDS read
};
template <>
struct Tw<double> {
  using type = double2;
  typedef double quad __attribute__((ext_vector_type(2)));
};

template <typename T>
__device__ __forceinline__ T wave_sum(T v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// sum of val[q] * pw[col[q]] over one CSR row, four taps in flight (a lane walks its filter alone,
// so the loads of consecutive taps are what hides their latency)
template <typename T>
__device__ __forceinline__ T filter_sum(int q, int end, const int32_t *__restrict__ col,
                                        const T *__restrict__ val, const T *pw) {
  T a0 = 0, a1 = 0, a2 = 0, a3 = 0;
  for (; q + 4 <= end; q += 4) {
    const int c0 = col[q], c1 = col[q + 1], c2 = col[q + 2], c3 = col[q + 3];
    const T v0 = val[q], v1 = val[q + 1], v2 = val[q + 2], v3 = val[q + 3];
    a0 += v0 * pw[c0];
    a1 += v1 * pw[c1];
    a2 += v2 * pw[c2];
    a3 += v3 * pw[c3];
  }
  for (; q < end; ++q) a0 += val[q] * pw[col[q]];
  return (a0 + a1) + (a2 + a3);
}

// one wave owns its LDS area: its LDS operations execute in order, this only keeps the compiler
// from moving memory operations across the hand-off between lanes
__device__ __forceinline__ void lds_handoff() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <typename T, int FPB>
__global__ __launch_bounds__(64 * FPB) void stft_generic_kernel(
    const T *__restrict__ sig, const int64_t *__restrict__ offsets,
    const int64_t *__restrict__ lengths, const int64_t *__restrict__ nframes,
    const int64_t *__restrict__ row_off, const T *__restrict__ window,
    const typename Tw<T>::type *__restrict__ tw, const int32_t *__restrict__ row_ptr,
    const int32_t *__restrict__ col, const T *__restrict__ val, T *__restrict__ out,
    int64_t out_stride, int L, int S, int N, int num_bins, int pad_left, int F, int use_power,
    int use_log, int include_energy, T log_floor, T preemph) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int b = blockIdx.y;
  const int64_t nfr = nframes[b];
  const int64_t t = (int64_t)blockIdx.x * FPB + wave;
  if (t >= nfr) return;  // whole wave leaves; no block-level barrier is used below
  const int64_t n = lengths[b];
  const T *x = sig + offsets[b];
  constexpr int J = 16;                   // samples per block of the dot product
  const int Lp = (L + J - 1) / J * J;     // frame padded with zeros to whole blocks
  // (a wave's area is a multiple of 16 bytes long, so the frame can be read 16 bytes at a time)
  T *xw = reinterpret_cast<T *>(smem_raw) + (size_t)wave * (Lp + ((num_bins + 3) & ~3));
  T *pw = xw + Lp;

  // frame -> LDS (windowed); energy on the un-windowed samples (compute.py:392-393)
  const int64_t start = t * S - pad_left;
  T e = 0;
  for (int j = lane; j < L; j += 64) {
    const int64_t i = reflect_index(start + j, n);
    T s = x[i];
    if (preemph != (T)0 && i > 0) s = preemph_sample(s, x[i - 1], preemph);  // pre.py:146 before framing
    e += s * s;
    xw[j] = s * window[j];
  }
  for (int j = L + lane; j < Lp; j += 64) xw[j] = (T)0;
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): this wave's LDS writes have landed

  // direct DFT of the L non-zero samples (zero padding to N contributes nothing)
  for (int k = lane; k < num_bins; k += 64) {
    T cj[J], sj[J];  // (cos, sin)(2 pi j k / N), j < J
#pragma unroll
    for (int j = 0; j < J; ++j) {
      const typename Tw<T>::type w = tw[(j * k) % N];
      cj[j] = w.x;
      sj[j] = w.y;
    }
    const typename Tw<T>::type step = tw[(J * k) % N];
    T c0 = 1, s0 = 0;  // (cos, sin)(2 pi j0 k / N) of the current block's first sample
    T re = 0, im = 0;
    for (int j0 = 0; j0 < Lp; j0 += J) {
      T a = 0, bs = 0;  // sum v cos, sum v sin over the block, relative to its first sample
      using Quad = typename Tw<T>::quad;
      constexpr int QN = 16 / sizeof(T);
#pragma unroll
      for (int j = 0; j < J; j += QN) {
        const Quad v = *reinterpret_cast<const Quad *>(xw + j0 + j);
#pragma unroll
        for (int u = 0; u < QN; ++u) {
          a += v[u] * cj[j + u];
          bs += v[u] * sj[j + u];
        }
      }
      re += c0 * a - s0 * bs;
      im -= s0 * a + c0 * bs;
      const T c1 = c0 * step.x - s0 * step.y;
      s0 = s0 * step.x + c0 * step.y;
      c0 = c1;
    }
    pw[k] = use_power ? re * re + im * im : sqrt(re * re + im * im);
  }
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_s_waitcnt(0xc07f);

  T *orow = out + (row_off[b] + t) * out_stride;
  if (include_energy) {
    e = wave_sum(e) / (T)L;
    if (!use_power) e = sqrt(e);
    if (use_log) e = log(log_floor > e ? log_floor : e);  // Python max(): NaN stays NaN
    if (lane == 0) orow[0] = e;
    orow += 1;
  }
  for (int f = lane; f < F; f += 64) {
    T acc = filter_sum(row_ptr[f], row_ptr[f + 1], col, val, pw);
    if (use_log) acc = log(log_floor > acc ? log_floor : acc);
    orow[f] = acc;
  }
}

// Power-of-two transforms outside the fused kernel's reach (float64 signals; float32 frames longer
// than 4096 samples): the same frame-per-wavefront layout with the direct DFT replaced by an
// N/2-point radix-2 FFT in LDS.  The real frame is packed as z[n] = x[2n] + i x[2n+1], stored in
// bit-reversed order, transformed in place in radix-4 passes (pairs of radix-2 passes; twiddles from
// the plan's N-entry table, staged in LDS, so float64 results keep the table's accuracy) and untangled into the
// N/2 + 1 bins while the power (or magnitude) spectrum is formed.  O(N log N) per frame instead of
// O(L N): 15 x the direct kernel at N = 512.
template <typename T, int FPB>
__global__ __launch_bounds__(64 * FPB) void stft_fft_kernel(
    const T *__restrict__ sig, const int64_t *__restrict__ offsets,
    const int64_t *__restrict__ lengths, const int64_t *__restrict__ nframes,
    const int64_t *__restrict__ row_off, const T *__restrict__ window,
    const typename Tw<T>::type *__restrict__ tw, const int32_t *__restrict__ row_ptr,
    const int32_t *__restrict__ col, const T *__restrict__ val, T *__restrict__ out,
    int64_t out_stride, int L, int S, int N, int log2m, int pad_left, int F, int use_power,
    int use_log, int include_energy, T log_floor, T preemph) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  using C = typename Tw<T>::type;  // (re, im)
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int b = blockIdx.y;
  const int M = N >> 1;  // complex points
  // the first half of the twiddle table (all the passes and the untangling need) -> LDS, once per
  // workgroup: a global-memory round trip per butterfly otherwise
  C *twl = reinterpret_cast<C *>(smem_raw);
  for (int i = threadIdx.x; i < M; i += 64 * FPB) twl[i] = tw[i];
  __syncthreads();
  const int64_t nfr = nframes[b];
  const int64_t t = (int64_t)blockIdx.x * FPB + wave;
  if (t >= nfr) return;  // whole wave leaves; no block-level barrier is used below
  const int64_t n = lengths[b];
  const T *x = sig + offsets[b];
  // a wave's area: z[M] then pw[M + 1] (rounded up to keep the next wave's z 16-byte aligned)
  C *z = twl + M + (size_t)wave * (M + (M + 2) / 2);
  T *pw = reinterpret_cast<T *>(z + M);
  T *zs = reinterpret_cast<T *>(z);

  // frame -> LDS, windowed, packed and bit-reversed; energy on the un-windowed samples
  const int64_t start = t * S - pad_left;
  T e = 0;
  for (int j = lane; j < N; j += 64) {
    T v = 0;
    if (j < L) {
      const int64_t i = reflect_index(start + j, n);
      T s = x[i];
      if (preemph != (T)0 && i > 0) s = preemph_sample(s, x[i - 1], preemph);
      e += s * s;
      v = s * window[j];
    }
    const unsigned pos = log2m ? __brev((unsigned)(j >> 1)) >> (32 - log2m) : 0u;
    zs[2 * pos + (j & 1)] = v;
  }
  lds_handoff();

  // decimation in time: pass s joins blocks of half = 2^(s-1) points with W_(2 half)^k = tw[k N / (2 half)].
  // Passes are taken two at a time (a radix-4 butterfly on i, i + half, i + 2 half, i + 3 half: the
  // same arithmetic as the two radix-2 passes, half the trips through LDS and half the hand-offs);
  // an odd pass count ends with one radix-2 pass.
  auto cmul = [](const C &v, const C &w) {  // v * (w.x - i w.y)
    C r;
    r.x = v.x * w.x + v.y * w.y;
    r.y = v.y * w.x - v.x * w.y;
    return r;
  };
  int s = 1;
  for (; s + 1 <= log2m; s += 2) {
    const int half = 1 << (s - 1);
    for (int q = lane; q < (M >> 2); q += 64) {
      const int k = q & (half - 1);
      const int i = ((q - k) << 2) + k;
      const C w1 = twl[k * (N >> s)], w2 = twl[k * (N >> (s + 1))];
      const C a0 = z[i], a1 = cmul(z[i + half], w1), a2 = z[i + 2 * half], a3 = cmul(z[i + 3 * half], w1);
      C b0, b1, b2, b3;
      b0.x = a0.x + a1.x, b0.y = a0.y + a1.y;
      b1.x = a0.x - a1.x, b1.y = a0.y - a1.y;
      b2.x = a2.x + a3.x, b2.y = a2.y + a3.y;
      b3.x = a2.x - a3.x, b3.y = a2.y - a3.y;
      const C c2 = cmul(b2, w2), t3 = cmul(b3, w2);
      C c3;  // W_(4 half)^(k + half) = -i W_(4 half)^k
      c3.x = t3.y;
      c3.y = -t3.x;
      C o;
      o.x = b0.x + c2.x, o.y = b0.y + c2.y;
      z[i] = o;
      o.x = b1.x + c3.x, o.y = b1.y + c3.y;
      z[i + half] = o;
      o.x = b0.x - c2.x, o.y = b0.y - c2.y;
      z[i + 2 * half] = o;
      o.x = b1.x - c3.x, o.y = b1.y - c3.y;
      z[i + 3 * half] = o;
    }
    lds_handoff();
  }
  if (s == log2m) {
    const int half = 1 << (s - 1);
    for (int q = lane; q < (M >> 1); q += 64) {
      const int k = q & (half - 1);
      const int i = ((q - k) << 1) + k, j = i + half;
      const C a = z[i], tq = cmul(z[j], twl[k * (N >> s)]);
      C lo, hi;
      lo.x = a.x + tq.x, lo.y = a.y + tq.y;
      hi.x = a.x - tq.x, hi.y = a.y - tq.y;
      z[i] = lo;
      z[j] = hi;
    }
    lds_handoff();
  }

  // untangle: with E = (Z[k] + conj Z[M-k]) / 2 and O = (Z[k] - conj Z[M-k]) / 2i,
  // X[k] = E + W_N^k O and X[M-k] = conj(E - W_N^k O)
  for (int k = lane; k <= (M >> 1); k += 64) {
    const C zk = z[k], zm = z[(M - k) & (M - 1)];
    const T er = (T)0.5 * (zk.x + zm.x), ei = (T)0.5 * (zk.y - zm.y);
    const T orr = (T)0.5 * (zk.y + zm.y), oi = (T)0.5 * (zm.x - zk.x);
    const C w = twl[k];
    const T pr = orr * w.x + oi * w.y, pi = oi * w.x - orr * w.y;  // W_N^k O
    const T ar = er + pr, ai = ei + pi, br = er - pr, bi = ei - pi;
    const T pa = ar * ar + ai * ai, pb = br * br + bi * bi;
    pw[k] = use_power ? pa : sqrt(pa);
    pw[M - k] = use_power ? pb : sqrt(pb);
  }
  lds_handoff();

  T *orow = out + (row_off[b] + t) * out_stride;
  if (include_energy) {
    e = wave_sum(e) / (T)L;
    if (!use_power) e = sqrt(e);
    if (use_log) e = log(log_floor > e ? log_floor : e);  // Python max(): NaN stays NaN
    if (lane == 0) orow[0] = e;
    orow += 1;
  }
  for (int f = lane; f < F; f += 64) {
    T acc = filter_sum(row_ptr[f], row_ptr[f + 1], col, val, pw);
    if (use_log) acc = log(log_floor > acc ? log_floor : acc);
    orow[f] = acc;
  }
}

template <typename T, int FPB>
static int32_t launch_fft(const pds_stft_plan *p, const BatchArgs &a, const T *window,
                          const typename Tw<T>::type *tw, const T *val, size_t smem, int log2m) {
  dim3 grid((unsigned)((a.max_frames + FPB - 1) / FPB), (unsigned)a.B);
  auto kern = stft_fft_kernel<T, FPB>;
  if (smem > 64 * 1024)
    PDS_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)smem));
  hipLaunchKernelGGL(kern, grid, dim3(64 * FPB), smem, a.stream, (const T *)a.d_signal,
                     a.d_offsets, a.d_lengths, a.d_nframes, a.d_row_off, window, tw,
                     p->d_row_ptr, p->d_col, val, (T *)a.d_out, a.out_stride, p->d.frame_length,
                     p->d.frame_shift, p->d.dft_size, log2m, a.pad_left, p->d.num_filts,
                     p->d.use_power, p->d.use_log, p->d.include_energy, (T)p->d.log_floor,
                     (T)a.preemph);
  PDS_HIP(hipGetLastError());
  return PDS_OK;
}

template <typename T, int FPB>
static int32_t launch_one(const pds_stft_plan *p, const BatchArgs &a, const T *window,
                          const typename Tw<T>::type *tw, const T *val, size_t smem) {
  dim3 grid((unsigned)((a.max_frames + FPB - 1) / FPB), (unsigned)a.B);
  auto kern = stft_generic_kernel<T, FPB>;
  if (smem > 64 * 1024)
    PDS_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)smem));
  hipLaunchKernelGGL(kern, grid, dim3(64 * FPB), smem, a.stream, (const T *)a.d_signal,
                     a.d_offsets, a.d_lengths, a.d_nframes, a.d_row_off, window, tw,
                     p->d_row_ptr, p->d_col, val, (T *)a.d_out, a.out_stride, p->d.frame_length,
                     p->d.frame_shift, p->d.dft_size, p->num_bins, a.pad_left, p->d.num_filts,
                     p->d.use_power, p->d.use_log, p->d.include_energy, (T)p->d.log_floor,
                     (T)a.preemph);
  PDS_HIP(hipGetLastError());
  return PDS_OK;
}

template <typename T>
static int32_t launch_generic(const pds_stft_plan *p, const BatchArgs &a, const T *window,
                              const typename Tw<T>::type *tw, const T *val) {
  const size_t budget = 150 * 1024;
  const int N = p->d.dft_size;
  if (N >= 4 && (N & (N - 1)) == 0) {  // power of two: FFT in LDS
    int log2m = 0;
    while ((2 << log2m) < N) ++log2m;  // N / 2 = 2^log2m
    const size_t per_wave = (size_t)(N / 2 + (N / 2 + 2) / 2) * 2 * sizeof(T);
    const size_t table = (size_t)(N / 2) * 2 * sizeof(T);
    if (table + 4 * per_wave <= budget)
      return launch_fft<T, 4>(p, a, window, tw, val, table + 4 * per_wave, log2m);
    if (table + per_wave <= budget) return launch_fft<T, 1>(p, a, window, tw, val, table + per_wave, log2m);
  }
  const size_t per_frame = (size_t)((p->d.frame_length + 15) / 16 * 16 + ((p->num_bins + 3) & ~3)) * sizeof(T);
  if (4 * per_frame <= budget) return launch_one<T, 4>(p, a, window, tw, val, 4 * per_frame);
  if (per_frame <= budget) return launch_one<T, 1>(p, a, window, tw, val, per_frame);
  set_error("stft_batch: frame_length + dft_size/2 too large for LDS");
  return PDS_ERR_INVALID;
}

int32_t launch_stft_generic_f32(const pds_stft_plan *p, const BatchArgs &a) {
  return launch_generic<float>(p, a, p->d_window_f32, p->d_tw_f32, p->d_val_f32);
}

int32_t launch_stft_generic_f64(const pds_stft_plan *p, const BatchArgs &a) {
  return launch_generic<double>(p, a, p->d_window_f64, p->d_tw_f64, p->d_val_f64);
}

}  // namespace pds
