// Pre-processor kernels: Preemphasize (reference pre.py:103-149) and Dither (pre.py:67-100).
// Element-wise, HBM-bound; intermediate arithmetic in float64 like the reference, result cast
// back to the signal's type.  The STFT kernels can also apply pre-emphasis while loading frames
// (pds_stft_batch_*'s `preemph` argument), which saves this pass over the signal.
#include "pds_internal.h"

namespace pds {

static int32_t invalid_pre(const char *msg) {
  set_error(msg);
  return PDS_ERR_INVALID;
}

// packed utterances: new[i] = old[i] - coeff * old[i-1] for i >= 1, new[0] = old[0], per utterance
template <typename T>
__global__ __launch_bounds__(256) void preemph_kernel(const T *__restrict__ in,
                                                      const int64_t *__restrict__ offsets,
                                                      const int64_t *__restrict__ lengths,
                                                      double coeff, T *__restrict__ out) {
  const int b = blockIdx.y;
  const int64_t n = lengths[b];
  const T *x = in + offsets[b];
  T *y = out + offsets[b];
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    double v = (double)x[i];
    if (i > 0) v = __dsub_rn(v, __dmul_rn(coeff, (double)x[i - 1]));
    y[i] = (T)v;
  }
}

// Philox4x32-10 (Salmon et al., SC'11): counter = element block, key = seed
__device__ __forceinline__ void philox4x32(uint32_t c[4], uint32_t k0, uint32_t k1) {
#pragma unroll
  for (int round = 0; round < 10; ++round) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
    const uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0;
    const uint32_t n1 = (uint32_t)p1;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
    const uint32_t n3 = (uint32_t)p0;
    c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
}

// out[i] = in[i] + coeff * N(0, 1); one thread per 2 samples (one Box-Muller pair from 53+53 bits)
template <typename T>
__global__ __launch_bounds__(256) void dither_kernel(const T *__restrict__ in, int64_t total,
                                                     double coeff, uint64_t seed,
                                                     T *__restrict__ out) {
  const int64_t pairs = (total + 1) / 2;
  for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < pairs;
       q += (int64_t)gridDim.x * blockDim.x) {
    uint32_t c[4] = {(uint32_t)q, (uint32_t)(q >> 32), 0u, 0u};
    philox4x32(c, (uint32_t)seed, (uint32_t)(seed >> 32));
    // two uniforms in (0, 1] and [0, 1)
    const double u1 = ((double)(((uint64_t)c[0] << 21) ^ (c[1] >> 11)) + 1.0) * (1.0 / 9007199254740992.0);
    const double u2 = (double)(((uint64_t)c[2] << 21) ^ (c[3] >> 11)) * (1.0 / 9007199254740992.0);
    const double rad = sqrt(-2.0 * log(u1));
    double s, co;
    sincospi(2.0 * u2, &s, &co);
    const int64_t i = 2 * q;
    out[i] = (T)((double)in[i] + coeff * rad * co);
    if (i + 1 < total) out[i + 1] = (T)((double)in[i + 1] + coeff * rad * s);
  }
}

template <typename T>
static int32_t launch_preemph(const T *d_in, const int64_t *d_off, const int64_t *d_len, int32_t B,
                              int64_t max_len, double coeff, T *d_out, void *stream) {
  if (B < 0 || max_len < 0) return invalid_pre("preemphasize: negative size");
  if (B == 0 || max_len == 0) return PDS_OK;
  if (B > 65535) return invalid_pre("preemphasize: B > 65535");
  if (!d_in || !d_off || !d_len || !d_out) return invalid_pre("preemphasize: null pointer");
  int64_t blocks = (max_len + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(preemph_kernel<T>, dim3((unsigned)blocks, (unsigned)B), dim3(256), 0,
                     (hipStream_t)stream, d_in, d_off, d_len, coeff, d_out);
  PDS_HIP(hipGetLastError());
  return PDS_OK;
}

template <typename T>
static int32_t launch_dither(const T *d_in, int64_t total, double coeff, uint64_t seed, T *d_out,
                             void *stream) {
  if (total < 0) return invalid_pre("dither: negative size");
  if (total == 0) return PDS_OK;
  if (!d_in || !d_out) return invalid_pre("dither: null pointer");
  int64_t blocks = ((total + 1) / 2 + 255) / 256;
  if (blocks > 65536 * 4) blocks = 65536 * 4;
  hipLaunchKernelGGL(dither_kernel<T>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                     d_in, total, coeff, seed, d_out);
  PDS_HIP(hipGetLastError());
  return PDS_OK;
}

}  // namespace pds

extern "C" {

int32_t pds_preemphasize_f32(const float *d_in, const int64_t *d_offsets, const int64_t *d_lengths,
                             int32_t B, int64_t max_len, double coeff, float *d_out, void *stream) {
  return pds::launch_preemph<float>(d_in, d_offsets, d_lengths, B, max_len, coeff, d_out, stream);
}
int32_t pds_preemphasize_f64(const double *d_in, const int64_t *d_offsets,
                             const int64_t *d_lengths, int32_t B, int64_t max_len, double coeff,
                             double *d_out, void *stream) {
  return pds::launch_preemph<double>(d_in, d_offsets, d_lengths, B, max_len, coeff, d_out, stream);
}
int32_t pds_dither_f32(const float *d_in, int64_t total, double coeff, uint64_t seed, float *d_out,
                       void *stream) {
  return pds::launch_dither<float>(d_in, total, coeff, seed, d_out, stream);
}
int32_t pds_dither_f64(const double *d_in, int64_t total, double coeff, uint64_t seed,
                       double *d_out, void *stream) {
  return pds::launch_dither<double>(d_in, total, coeff, seed, d_out, stream);
}

}  // extern "C"
