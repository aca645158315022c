// Short-integration features (reference compute.py:613-996, ShortIntegrationFrameComputer):
//
//     y_f[i] = sum_{k < M} g_f[k] * sig[i + start - k]      FIR filter bank, complex or real taps
//     z_f[i] = |y_f[i]|^2  or  |y_f[i]|
//     out[t][f] = log(max(sum_{m < 2S} window[m] * z_f[t S + m], floor))
//
// The reference evaluates this with a streaming overlap-save FFT per filter (compute.py:781-932);
// its own test restates it as the plain convolution above (tests/test_compute.py:129-171), and
// that is what this first device version computes, directly in the time domain:
//
// * a workgroup takes one utterance and JB consecutive shift-sized blocks of the integrated
//   stream; the signal stretch those need (JB S + M - 1 samples) is staged once in LDS and
//   serves every filter;
// * a thread owns R = 9 consecutive samples (an odd stride keeps the 32 lanes of an LDS access
//   on distinct banks).  For nine taps at a time it reads the 17 samples its outputs touch and
//   issues 81 (real taps) or 162 (complex taps) fused multiply-adds on them; the taps are uniform
//   across the workgroup and arrive as scalar loads, so they cost no vector registers and no LDS
//   traffic;
// * |y|^2 times the two window halves goes back to LDS, one wave per (block, half) sums it in a
//   fixed order, and frame t = first-half sum of block t + second-half sum of block t + 1 is
//   written with the log applied.  Neighbouring workgroups overlap by one block, so there is no
//   scratch array, no second kernel and no atomics: results are bitwise reproducible.
//
// Arithmetic is the signal's own precision (float or double), as for the STFT kernels.  Work is
// O(M) per sample and filter; an overlap-save FFT variant is the obvious next step for banks with
// supports in the thousands of samples.
#include <vector>

#include "pds_internal.h"

namespace pds {

static int32_t invalid_si(const char *msg) {
  set_error(msg);
  return PDS_ERR_INVALID;
}

constexpr int kSiR = 9;          // consecutive samples per thread
constexpr int kSiThreads = 256;

template <typename T>
struct SiArgs {
  const T *sig;
  const int64_t *offsets, *lengths, *nframes, *row_off;
  T *out;
  int64_t out_stride;
  const T *taps, *window;
  int64_t start;
  int S, M, mpad, C, JB, use_power, use_log;
  T log_floor;
};

template <typename T>
__device__ __forceinline__ T load_uniform(const T *ptr) {
  typedef const T __attribute__((address_space(4))) *const_ptr;
  return *(const_ptr)(uintptr_t)ptr;
}

template <typename T, bool COMPLEX>
__global__ __launch_bounds__(kSiThreads) void si_conv_kernel(const SiArgs<T> p) {
  constexpr int R = kSiR;
  extern __shared__ __attribute__((aligned(16))) unsigned char si_smem[];
  const int S = p.S, JB = p.JB, tile = JB * S;
  const int seglen = tile + p.mpad - 1 + R;  // + R: the last thread's window may overhang
  T *seg = reinterpret_cast<T *>(si_smem);
  T *zw = seg + seglen;          // [2][tile]: z times first / second window half
  T *red = zw + 2 * tile;        // [2][JB]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.y;
  const int64_t Tb = p.nframes[b];
  const int64_t j0 = (int64_t)blockIdx.x * (JB - 1);
  if (j0 >= Tb) return;
  const int64_t n = p.lengths[b];
  const T *x = p.sig + p.offsets[b];
  // stage sig[j0 S + start - (mpad - 1) ...], zeros outside the utterance
  const int64_t a = j0 * S + p.start - (p.mpad - 1);
  for (int e = tid; e < seglen; e += kSiThreads) {
    const int64_t idx = a + e;
    seg[e] = (idx >= 0 && idx < n) ? x[idx] : (T)0;
  }
  __syncthreads();
  T *obase = p.out + (p.row_off[b] + j0) * p.out_stride;
  for (int c = 0; c < p.C; ++c) {
   // (a tile longer than kSiThreads * R samples -- frame shifts beyond ~1150 samples -- takes
   // several passes of the thread block)
   for (int first = tid * R; first < tile; first += kSiThreads * R) {
    const int base = p.mpad - 1 + first;    // the thread's position in `seg` for tap 0
    T yr[R], yi[R];
#pragma unroll
    for (int r = 0; r < R; ++r) yr[r] = yi[r] = (T)0;
    {
      const T *g = p.taps + (size_t)c * p.mpad * (COMPLEX ? 2 : 1);
      for (int kb = 0; kb < p.mpad; kb += R) {
        T v[2 * R - 1];  // v[q] = sample (q - (R - 1)) positions after the one tap kb reaches
#pragma unroll
        for (int q = 0; q < 2 * R - 1; ++q) v[q] = seg[base - kb - (R - 1) + q];
#pragma unroll
        for (int u = 0; u < R; ++u) {
          T gr, gi = (T)0;
          if constexpr (COMPLEX) {  // one 8- or 16-byte scalar load per tap
            typedef T Pair __attribute__((ext_vector_type(2)));
            const Pair t = load_uniform(reinterpret_cast<const Pair *>(g) + kb + u);
            gr = t.x;
            gi = t.y;
          } else {
            gr = load_uniform(g + kb + u);
          }
#pragma unroll
          for (int r = 0; r < R; ++r) {
            yr[r] += gr * v[R - 1 + r - u];
            if constexpr (COMPLEX) yi[r] += gi * v[R - 1 + r - u];
          }
        }
      }
      int m = first % S;
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const int il = first + r;
        T z = yr[r] * yr[r] + yi[r] * yi[r];
        if (!p.use_power) z = sqrt(z);
        if (il < tile) {
          zw[il] = z * p.window[m];
          zw[tile + il] = z * p.window[S + m];
        }
        m = m + 1 == S ? 0 : m + 1;
      }
    }
   }
    __syncthreads();
    // (block jj, half h): strided partial sums, then a butterfly over the wave (fixed order)
    for (int task = wave; task < 2 * JB; task += kSiThreads / 64) {
      const int jj = task >> 1, h = task & 1;
      const T *src = zw + h * tile + jj * S;
      T s = (T)0;
      for (int m = lane; m < S; m += 64) s += src[m];
#pragma unroll
      for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off, 64);
      if (lane == 0) red[h * JB + jj] = s;
    }
    __syncthreads();
    if (tid < JB - 1 && j0 + tid < Tb) {
      T val = red[tid] + red[JB + tid + 1];
      // max(val, floor) as numpy.maximum evaluates it: NaN propagates (compute.py:989-990)
      if (p.use_log) val = log(val < p.log_floor ? p.log_floor : val);
      obase[(int64_t)tid * p.out_stride + c] = val;
    }
    // `red` is read above by threads that only reach the next reduction after the next barrier
  }
}

template <typename T>
static int32_t launch_si(const pds_si_plan *plan, const T *d_signal, const int64_t *d_offsets,
                         const int64_t *d_lengths, const int64_t *d_nframes,
                         const int64_t *d_row_off, int32_t B, int64_t max_frames, int64_t start,
                         T *d_out, int64_t out_stride, void *stream) {
  if (!plan) return invalid_si("si_batch: null plan");
  if (B < 0 || max_frames < 0) return invalid_si("si_batch: negative size");
  if (B == 0 || max_frames == 0) return PDS_OK;
  if (B > 65535) return invalid_si("si_batch: B > 65535");
  if (!d_signal || !d_offsets || !d_lengths || !d_nframes || !d_row_off || !d_out)
    return invalid_si("si_batch: null pointer");
  const pds_si_desc &d = plan->d;
  if (out_stride < d.num_coeffs) return invalid_si("si_batch: out_stride < num_coeffs");
  if (int32_t rc = check_plan_device(plan->device, "si_batch"); rc != PDS_OK) return rc;
  const int S = d.frame_shift;
  int JB = (kSiThreads * kSiR) / S;
  if (JB < 2) JB = 2;  // long shifts: two blocks per tile, several passes of the thread block
  if ((int64_t)JB - 1 > max_frames) JB = (int)max_frames + 1;
  const int tile = JB * S;
  const size_t smem = ((size_t)tile + plan->mpad - 1 + kSiR + 2 * (size_t)tile + 2 * (size_t)JB) * sizeof(T);
  if (smem > 160 * 1024) return invalid_si("si_batch: filter support too long for the LDS tile");
  SiArgs<T> p;
  p.sig = d_signal;
  p.offsets = d_offsets;
  p.lengths = d_lengths;
  p.nframes = d_nframes;
  p.row_off = d_row_off;
  p.out = d_out;
  p.out_stride = out_stride;
  if constexpr (sizeof(T) == 4) {
    p.taps = (const T *)plan->d_taps_f32;
    p.window = (const T *)plan->d_window_f32;
  } else {
    p.taps = (const T *)plan->d_taps_f64;
    p.window = (const T *)plan->d_window_f64;
  }
  p.start = start;
  p.S = S;
  p.M = d.max_support;
  p.mpad = plan->mpad;
  p.C = d.num_coeffs;
  p.JB = JB;
  p.use_power = d.use_power;
  p.use_log = d.use_log;
  p.log_floor = (T)d.log_floor;
  auto kern = d.taps_complex ? si_conv_kernel<T, true> : si_conv_kernel<T, false>;
  PDS_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
  dim3 grid((unsigned)((max_frames + JB - 2) / (JB - 1)), (unsigned)B);
  hipLaunchKernelGGL(kern, grid, dim3(kSiThreads), smem, (hipStream_t)stream, p);
  PDS_HIP(hipGetLastError());
  return PDS_OK;
}

}  // namespace pds

extern "C" {

int32_t pds_si_plan_create(const pds_si_desc *desc, const double *taps, const double *window,
                           pds_si_plan **plan_out) {
  if (!desc || !taps || !window || !plan_out) return pds::invalid_si("si_plan_create: null argument");
  const pds_si_desc &d = *desc;
  if (d.frame_shift < 1 || d.max_support < 1 || d.num_coeffs < 1 || d.reserved != 0 || d.reserved2 != 0)
    return pds::invalid_si("si_plan_create: need frame_shift, max_support, num_coeffs >= 1");
  if (!(d.log_floor > 0.0)) return pds::invalid_si("si_plan_create: log_floor must be positive");
  int device = -1;
  PDS_HIP(hipGetDevice(&device));
  pds_si_plan *plan = nullptr;
  const int32_t status = pds::no_throw("si_plan_create", [&]() -> int32_t {
    plan = new pds_si_plan();
    plan->d = d;
    plan->device = device;
    const int R = pds::kSiR, M = d.max_support, C = d.num_coeffs, w = d.taps_complex ? 2 : 1;
    plan->mpad = (M + R - 1) / R * R;
    std::vector<double> t64((size_t)C * plan->mpad * w, 0.0);
    for (int c = 0; c < C; ++c)
      for (int k = 0; k < M * w; ++k) t64[(size_t)c * plan->mpad * w + k] = taps[(size_t)c * M * w + k];
    std::vector<float> t32(t64.begin(), t64.end());
    std::vector<float> w32(window, window + 2 * (size_t)d.frame_shift);
    int32_t rc = pds::upload(&plan->d_taps_f64, t64.data(), t64.size());
    if (rc == PDS_OK) rc = pds::upload(&plan->d_taps_f32, t32.data(), t32.size());
    if (rc == PDS_OK) rc = pds::upload(&plan->d_window_f64, window, 2 * (size_t)d.frame_shift);
    if (rc == PDS_OK) rc = pds::upload(&plan->d_window_f32, w32.data(), w32.size());
    if (rc == PDS_OK) rc = pds::si_fft_tables_create(plan, taps);
    return rc;
  });
  if (status != PDS_OK) {
    pds_si_plan_destroy(plan);  // frees whatever was built (null-safe)
    return status;
  }
  *plan_out = plan;
  return PDS_OK;
}

void pds_si_plan_destroy(pds_si_plan *plan) {
  if (!plan) return;
  (void)hipFree(plan->d_taps_f32);
  (void)hipFree(plan->d_taps_f64);
  (void)hipFree(plan->d_window_f32);
  (void)hipFree(plan->d_window_f64);
  pds::si_fft_tables_destroy(plan);
  delete plan;
}

int32_t pds_si_plan_fft_size(const pds_si_plan *plan) {
  if (!plan || plan->fft.blocks == 0) return 0;
  return plan->fft.big ? 2048 : 1024;
}

int64_t pds_si_scratch_len(const pds_si_plan *plan, int32_t B, int64_t max_frames) {
  return pds::si_fft_scratch_len(plan, B, max_frames);
}

int32_t pds_si_batch_f32(const pds_si_plan *plan, const float *d_signal, const int64_t *d_offsets,
                         const int64_t *d_lengths, const int64_t *d_nframes,
                         const int64_t *d_row_off, int32_t B, int64_t max_frames, int64_t start,
                         float *d_scratch, float *d_out, int64_t out_stride, void *stream) {
  if (plan && d_scratch && plan->fft.blocks > 0 && B > 0 && B <= 65535 && max_frames > 0 && d_signal &&
      d_offsets && d_lengths && d_nframes && d_row_off && d_out && out_stride >= plan->d.num_coeffs &&
      pds::check_plan_device(plan->device, "si_batch") == PDS_OK)  // (the direct form reports a mismatch)
    return pds::launch_si_fft(plan, d_signal, d_offsets, d_lengths, d_nframes, d_row_off, B, max_frames,
                              start, d_scratch, d_out, out_stride, stream);
  return pds::launch_si<float>(plan, d_signal, d_offsets, d_lengths, d_nframes, d_row_off, B,
                               max_frames, start, d_out, out_stride, stream);
}

int32_t pds_si_batch_f64(const pds_si_plan *plan, const double *d_signal, const int64_t *d_offsets,
                         const int64_t *d_lengths, const int64_t *d_nframes,
                         const int64_t *d_row_off, int32_t B, int64_t max_frames, int64_t start,
                         double *d_out, int64_t out_stride, void *stream) {
  return pds::launch_si<double>(plan, d_signal, d_offsets, d_lengths, d_nframes, d_row_off, B,
                                max_frames, start, d_out, out_stride, stream);
}

}  // extern "C"
