// Post-processor kernels: Deltas (reference post.py:462-491) and Standardize / CMVN
// (post.py:193-212, 250-295).  Both are streaming, HBM-bound element-wise/stencil work;
// accumulation is float64 as in the reference.
#include "pds_internal.h"

namespace pds {

static int32_t invalid_post(const char *msg) {
  set_error(msg);
  return PDS_ERR_INVALID;
}

// ------------------------------------------------------------------ Deltas ---------

// one thread per input element (o, t, i); writes the K + 1 outputs of that element
template <typename T>
__global__ __launch_bounds__(256) void deltas_kernel(
    const T *__restrict__ in, int64_t outer, int64_t time, int64_t inner,
    const double *__restrict__ filts, const int32_t *__restrict__ filt_off, int K,
    int edge_clamp, int max_off, T *__restrict__ out, int64_t sk, int64_t so, int64_t st,
    int64_t si) {
  const int64_t total = outer * time * inner;
  const int64_t in_time = edge_clamp ? time : time + 2 * (int64_t)max_off;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t i = e % inner;
    const int64_t t = (e / inner) % time;
    const int64_t o = e / (inner * time);
    const T *col = in + o * in_time * inner + i;
    T *dst = out + o * so + t * st + i * si;
    const int64_t tc = edge_clamp ? t : t + max_off;
    dst[0] = col[tc * inner];
    for (int k = 1; k <= K; ++k) {
      const int lo = filt_off[k - 1], len = filt_off[k] - lo;
      const int M = (len - 1) / 2;
      double acc = 0.0;
      for (int j = 0; j < len; ++j) {
        int64_t tt = tc + j - M;
        if (edge_clamp) tt = tt < 0 ? 0 : (tt >= time ? time - 1 : tt);
        // multiply and add rounded separately (no FMA): bit-identical to numpy.correlate's
        // sequential sum, which matters when the result is cast to an integer dtype
        acc = __dadd_rn(acc, __dmul_rn(filts[lo + j], (double)col[tt * inner]));
      }
      dst[k * sk] = (T)acc;
    }
  }
}

template <typename T>
static int32_t launch_deltas(const T *d_in, int64_t outer, int64_t time, int64_t inner,
                             const double *d_filts, const int32_t *d_filt_off, int32_t K,
                             int32_t edge_clamp, int32_t max_off, T *d_out, int64_t sk,
                             int64_t so, int64_t st, int64_t si, void *stream) {
  if (outer < 0 || time < 0 || inner < 0 || K < 0 || max_off < 0)
    return invalid_post("deltas: negative size");
  const int64_t total = outer * time * inner;
  if (total == 0) return PDS_OK;
  if (!d_in || !d_out || (K > 0 && (!d_filts || !d_filt_off)))
    return invalid_post("deltas: null pointer");
  const int64_t blocks = (total + 255) / 256;
  const unsigned grid = (unsigned)(blocks < 65536 * 4 ? blocks : 65536 * 4);
  hipLaunchKernelGGL(deltas_kernel<T>, dim3(grid), dim3(256), 0, (hipStream_t)stream, d_in,
                     outer, time, inner, d_filts, d_filt_off, K, edge_clamp, max_off, d_out, sk,
                     so, st, si);
  PDS_HIP(hipGetLastError());
  return PDS_OK;
}

// ragged rows: grid (row tiles, utterance).  A block stages its rows plus the stencil halo in
// LDS once (edge clamping resolved there), then every thread produces the K delta values of its
// (row, coefficient) elements from LDS: HBM sees each static once and each delta once.
__global__ __launch_bounds__(256) void deltas_rows_kernel(
    const float *__restrict__ in, int64_t in_stride, const int64_t *__restrict__ row_off,
    const int64_t *__restrict__ nrows, int inner, const double *__restrict__ filts,
    const int32_t *__restrict__ filt_off, int K, int halo, float *__restrict__ out,
    int64_t out_stride, int rows_per_block, int copy_statics) {
  extern __shared__ float tile[];  // [(rows_per_block + 2 halo)][inner]
  const int b = blockIdx.y;
  const int64_t T = nrows[b];
  const int64_t t0 = (int64_t)blockIdx.x * rows_per_block;
  if (t0 >= T) return;
  const float *src = in + row_off[b] * in_stride;
  float *dst = out + row_off[b] * out_stride;
  const float inv_inner = 1.0f / (float)inner;
  const int staged = (rows_per_block + 2 * halo) * inner;
  for (int e = threadIdx.x; e < staged; e += blockDim.x) {
    const int lr = (int)(((float)e + 0.5f) * inv_inner);
    const int i = e - lr * inner;
    int64_t t = t0 - halo + lr;
    t = t < 0 ? 0 : (t >= T ? T - 1 : t);  // "edge" padding (post.py:447)
    tile[e] = src[t * in_stride + i];
  }
  __syncthreads();
  const int64_t left = T - t0;
  const int rows = left < rows_per_block ? (int)left : rows_per_block;
  const int items = rows * inner;
  for (int e = threadIdx.x; e < items; e += blockDim.x) {
    const int lr = (int)(((float)e + 0.5f) * inv_inner);
    const int i = e - lr * inner;
    float *orow = dst + (t0 + lr) * out_stride + i;
    const float *col = tile + (lr + halo) * inner + i;
    if (copy_statics) orow[0] = col[0];
    for (int k = 1; k <= K; ++k) {
      const int lo = filt_off[k - 1], len = filt_off[k] - lo;
      const int M = (len - 1) / 2;
      double acc = 0.0;
      for (int j = 0; j < len; ++j)
        acc = __dadd_rn(acc, __dmul_rn(filts[lo + j], (double)col[(j - M) * inner]));
      orow[(int64_t)k * inner] = (float)acc;
    }
  }
}

// ragged rows, register-window form for the reference's delta filters (filter k has 2 k W + 1
// taps, post.py:441-452): a thread owns 8 consecutive rows of one coefficient, reads the
// 8 + 2 K W statics it needs once (edge clamping resolved in the row index), converts them to
// float64 once and produces all 8 x K deltas from registers with the taps held in scalar
// registers.  No LDS, no barrier: the rows a neighbouring thread re-reads come out of L1/L2, HBM
// sees each static once.  ~40 instructions per static instead of ~80 for the tiled kernel above,
// which is what bounds this kernel (28 separately rounded float64 operations per static for
// K = 2, W = 2).  Filters that do not have the expected lengths take the generic loop at the end.
#ifndef PDS_DELTA_ROWS
#define PDS_DELTA_ROWS 8  // consecutive rows per thread
#endif
template <int K, int W>
__global__ __launch_bounds__(256) void deltas_rows_win_kernel(
    const float *__restrict__ in, int64_t in_stride, const int64_t *__restrict__ row_off,
    const int64_t *__restrict__ nrows, int inner, const double *__restrict__ filts,
    const int32_t *__restrict__ filt_off, float *__restrict__ out, int64_t out_stride,
    int copy_statics) {
  constexpr int H = K * W, R = PDS_DELTA_ROWS, NV = R + 2 * H;
  constexpr int TAPS = K * (K + 1) * W + K;  // sum over k of 2 k W + 1
  const int b = blockIdx.y;
  const int64_t T = nrows[b];
  const int64_t items = (T + R - 1) / R * inner;
  // Workgroups go to the 8 XCDs round-robin by their linear index (each XCD has an L2 of its own), and
  // a row group re-reads 2 H rows of its neighbours: renumber the workgroups of an utterance so that
  // an XCD's ones cover one contiguous stretch of rows and the halo rows are fetched into one L2
  // instead of two (measured: 648 MB fetched per launch for 332 MB of statics before, PMC FETCH_SIZE)
  unsigned bx = blockIdx.x;
  {
    const unsigned main = gridDim.x & ~7u;  // (the linear index is x + y * gridDim.x: x % 8 when 8 | gridDim.x)
    if ((gridDim.x & 7u) == 0 && bx < main) bx = (bx & 7u) * (main >> 3) + (bx >> 3);
  }
  const int64_t e = (int64_t)bx * 256 + threadIdx.x;
  if ((int64_t)bx * 256 >= items) return;
  const float *src = in + row_off[b] * in_stride;
  float *dst = out + row_off[b] * out_stride;
  bool regular = true;
#pragma unroll
  for (int k = 1; k <= K; ++k) regular = regular && filt_off[k] - filt_off[k - 1] == 2 * k * W + 1;
  if (e >= items) return;
  const int64_t g = e / inner;
  const int i = (int)(e - g * inner);
  const int64_t t0 = g * R;
  if (regular) {
    double f[TAPS];
#pragma unroll
    for (int j = 0; j < TAPS; ++j) f[j] = filts[filt_off[0] + j];
    double v[NV];
    if (t0 >= H && t0 + R + H <= T) {
      const float *c = src + (t0 - H) * in_stride + i;
#pragma unroll
      for (int r = 0; r < NV; ++r) v[r] = (double)c[r * in_stride];
    } else {
#pragma unroll
      for (int r = 0; r < NV; ++r) {
        int64_t t = t0 - H + r;
        t = t < 0 ? 0 : (t >= T ? T - 1 : t);  // "edge" padding (post.py:447)
        v[r] = (double)src[t * in_stride + i];
      }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
      if (t0 + r < T) {
        float *orow = dst + (t0 + r) * out_stride + i;
        if (copy_statics) orow[0] = (float)v[H + r];
        int fo = 0;
#pragma unroll
        for (int k = 1; k <= K; ++k) {
          const int len = 2 * k * W + 1, M = k * W;
          double acc = 0.0;
#pragma unroll
          for (int j = 0; j < len; ++j)
            acc = __dadd_rn(acc, __dmul_rn(f[fo + j], v[H + r + j - M]));
          orow[(int64_t)k * inner] = (float)acc;
          fo += len;
        }
      }
    }
    return;
  }
  for (int r = 0; r < R && t0 + r < T; ++r) {
    const int64_t t = t0 + r;
    float *orow = dst + t * out_stride + i;
    if (copy_statics) orow[0] = src[t * in_stride + i];
    for (int k = 1; k <= K; ++k) {
      const int lo = filt_off[k - 1], len = filt_off[k] - lo;
      const int M = (len - 1) / 2;
      double acc = 0.0;
      for (int j = 0; j < len; ++j) {
        int64_t tt = t + j - M;
        tt = tt < 0 ? 0 : (tt >= T ? T - 1 : tt);
        acc = __dadd_rn(acc, __dmul_rn(filts[lo + j], (double)src[tt * in_stride + i]));
      }
      orow[(int64_t)k * inner] = (float)acc;
    }
  }
}

// ------------------------------------------------------------------ Stack ----------

// Stack (reference post.py:494-563) over a packed ragged batch: output row t' of utterance b is
// the concatenation of its input rows t' nv .. t' nv + nv - 1.  Rows past the end of the
// utterance (only with padding: pad 1 = zeros, 2 = repeat the last row) are synthesised.
// Pure data movement: one thread per output element, reads and writes coalesced along a row.
__global__ __launch_bounds__(256) void stack_rows_kernel(
    const float *__restrict__ in, int64_t in_stride, const int64_t *__restrict__ row_off,
    const int64_t *__restrict__ nrows, const int64_t *__restrict__ out_row_off, int F, int nv,
    int pad, float *__restrict__ out, int64_t out_stride) {
  const int b = blockIdx.y;
  const int64_t T = nrows[b];
  const int64_t Tout = pad ? (T + nv - 1) / nv : T / nv;
  const int64_t width = (int64_t)nv * F;
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= Tout * width) return;
  const int64_t to = e / width;
  const int c = (int)(e - to * width);
  const int v = c / F, i = c - v * F;
  int64_t t = to * nv + v;
  float val = 0.0f;
  if (t < T || pad == 2) {
    if (t >= T) t = T - 1;
    val = in[(row_off[b] + t) * in_stride + i];
  }
  out[(out_row_off[b] + to) * out_stride + c] = val;
}

// ------------------------------------------------------------------ CMVN -----------

constexpr int kStatSlabs = 128;  // partial-sum slabs over `outer` (deterministic two-stage sum)

// stage 1: column q = c * inner + i; 64 columns x 4 row phases per block
template <typename T>
__global__ __launch_bounds__(256) void cmvn_partial_kernel(const T *__restrict__ in,
                                                           int64_t outer, int64_t Q,
                                                           double *__restrict__ partial) {
  __shared__ double red[2][4][64];
  const int lane = threadIdx.x & 63, phase = threadIdx.x >> 6;
  const int64_t q = (int64_t)blockIdx.y * 64 + lane;
  const int64_t slab = (outer + gridDim.x - 1) / gridDim.x;
  const int64_t o0 = (int64_t)blockIdx.x * slab;
  const int64_t o1 = o0 + slab < outer ? o0 + slab : outer;
  double s1 = 0.0, s2 = 0.0;
  if (q < Q)
    for (int64_t o = o0 + phase; o < o1; o += 4) {
      const double v = (double)in[o * Q + q];
      s1 += v;
      s2 += v * v;
    }
  red[0][phase][lane] = s1;
  red[1][phase][lane] = s2;
  __syncthreads();
  if (phase == 0 && q < Q) {
    s1 = (red[0][0][lane] + red[0][1][lane]) + (red[0][2][lane] + red[0][3][lane]);
    s2 = (red[1][0][lane] + red[1][1][lane]) + (red[1][2][lane] + red[1][3][lane]);
    partial[((int64_t)blockIdx.x * Q + q) * 2 + 0] = s1;
    partial[((int64_t)blockIdx.x * Q + q) * 2 + 1] = s2;
  }
}

// stage 2: one thread per coefficient sums slabs and the `inner` columns, fixed order
__global__ void cmvn_final_kernel(const double *__restrict__ partial, int slabs, int64_t C,
                                  int64_t inner, double *__restrict__ stats) {
  const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const int64_t Q = C * inner;
  double s1 = 0.0, s2 = 0.0;
  for (int g = 0; g < slabs; ++g)
    for (int64_t i = 0; i < inner; ++i) {
      s1 += partial[((int64_t)g * Q + c * inner + i) * 2 + 0];
      s2 += partial[((int64_t)g * Q + c * inner + i) * 2 + 1];
    }
  stats[c] = s1;
  stats[C + c] = s2;
}

template <typename T>
__global__ __launch_bounds__(256) void cmvn_apply_kernel(const T *__restrict__ in,
                                                         int64_t total, int64_t C,
                                                         int64_t inner,
                                                         const double *__restrict__ scale,
                                                         const double *__restrict__ shift,
                                                         double *__restrict__ out) {
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t c = (e / inner) % C;
    // x * scale - mean * scale, two roundings as in post.py:293-294
    out[e] = __dsub_rn(__dmul_rn((double)in[e], scale[c]), shift[c]);
  }
}

static int slabs_for(int64_t outer) {
  int64_t s = (outer + 63) / 64;
  if (s < 1) s = 1;
  return (int)(s < kStatSlabs ? s : kStatSlabs);
}

template <typename T>
static int32_t launch_cmvn_stats(const T *d_in, int64_t outer, int64_t C, int64_t inner,
                                 double *d_stats, double *d_scratch, void *stream) {
  if (outer <= 0 || C <= 0 || inner <= 0) return invalid_post("cmvn_stats: empty tensor");
  if (!d_in || !d_stats || !d_scratch) return invalid_post("cmvn_stats: null pointer");
  const int64_t Q = C * inner;
  const int slabs = slabs_for(outer);
  const int64_t qblocks = (Q + 63) / 64;
  if (qblocks > 65535) return invalid_post("cmvn_stats: coeff * inner too large");
  hipLaunchKernelGGL(cmvn_partial_kernel<T>, dim3(slabs, (unsigned)qblocks), dim3(256), 0,
                     (hipStream_t)stream, d_in, outer, Q, d_scratch);
  hipLaunchKernelGGL(cmvn_final_kernel, dim3((unsigned)((C + 63) / 64)), dim3(64), 0,
                     (hipStream_t)stream, d_scratch, slabs, C, inner, d_stats);
  PDS_HIP(hipGetLastError());
  return PDS_OK;
}

template <typename T>
static int32_t launch_cmvn_apply(const T *d_in, int64_t outer, int64_t C, int64_t inner,
                                 const double *d_scale, const double *d_shift, double *d_out,
                                 void *stream) {
  const int64_t total = outer * C * inner;
  if (total <= 0) return invalid_post("cmvn_apply: empty tensor");
  if (!d_in || !d_scale || !d_shift || !d_out) return invalid_post("cmvn_apply: null pointer");
  const int64_t blocks = (total + 255) / 256;
  const unsigned grid = (unsigned)(blocks < 65536 * 4 ? blocks : 65536 * 4);
  hipLaunchKernelGGL(cmvn_apply_kernel<T>, dim3(grid), dim3(256), 0, (hipStream_t)stream, d_in,
                     total, C, inner, d_scale, d_shift, d_out);
  PDS_HIP(hipGetLastError());
  return PDS_OK;
}

// per-utterance CMVN over ragged rows: one block per (utterance, 64-column chunk), 16 row
// phases x 64 columns; pass 1 statistics with 8 independent row loads in flight per thread
// (a 10 s utterance is only 1000 rows, so the latency of a load is what has to be covered),
// pass 2 normalises (the rows are re-read from L2)
constexpr int kRowPhases = 16;
#ifndef PDS_CMVN_APPLY_DEPTH  // row loads in flight per thread in the normalising pass of cmvn_rows_partials_kernel
#define PDS_CMVN_APPLY_DEPTH 8
#endif
template <typename OutT>
__global__ __launch_bounds__(kRowPhases * 64) void cmvn_rows_kernel(
    const float *__restrict__ in, int64_t in_stride, const int64_t *__restrict__ row_off,
    const int64_t *__restrict__ nrows, int C, int norm_var, double *__restrict__ stats,
    OutT *__restrict__ out, int64_t out_stride, int32_t *__restrict__ zero_var) {
  constexpr int P = kRowPhases, U = 8;
  __shared__ double red[2][P][64];
  __shared__ double sc[64], sh[64];
  const int b = blockIdx.y;
  const int lane = threadIdx.x & 63, phase = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + lane;
  const int64_t T = nrows[b];
  if (T <= 0) return;
  const float *src = in + row_off[b] * in_stride + (c < C ? c : 0);
  double s1 = 0.0, s2 = 0.0;
  if (c < C) {
    int64_t t = phase;
    for (; t + (U - 1) * P < T; t += U * P) {
      float v[U];
#pragma unroll
      for (int u = 0; u < U; ++u) v[u] = src[(t + u * P) * in_stride];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const double d = (double)v[u];
        s1 += d;
        s2 += d * d;
      }
    }
    for (; t < T; t += P) {
      const double d = (double)src[t * in_stride];
      s1 += d;
      s2 += d * d;
    }
  }
  red[0][phase][lane] = s1;
  red[1][phase][lane] = s2;
  __syncthreads();
  if (phase == 0 && c < C) {
    s1 = 0.0;
    s2 = 0.0;
#pragma unroll
    for (int q = 0; q < P; ++q) {  // fixed order: deterministic
      s1 += red[0][q][lane];
      s2 += red[1][q][lane];
    }
    stats[((int64_t)b * 2 + 0) * C + c] = s1;
    stats[((int64_t)b * 2 + 1) * C + c] = s2;
    const double mean = s1 / (double)T;
    double scale = 1.0;
    if (norm_var) {
      double var = s2 / (double)T - mean * mean;
      if (fabs(var) <= 1e-8) {  // numpy.isclose(var, 0) (post.py:283)
        var = 1.0;
        if (zero_var) atomicAdd(zero_var, 1);
      }
      scale = 1.0 / sqrt(var);
    }
    sc[lane] = scale;
    sh[lane] = mean * scale;
  }
  __syncthreads();
  if (c < C) {
    OutT *dst = out + row_off[b] * out_stride + c;
    const double scale = sc[lane], shift = sh[lane];
    int64_t t = phase;
    for (; t + (U - 1) * P < T; t += U * P) {
      float v[U];
#pragma unroll
      for (int u = 0; u < U; ++u) v[u] = src[(t + u * P) * in_stride];
#pragma unroll
      for (int u = 0; u < U; ++u)
        dst[(t + u * P) * out_stride] = (OutT)__dsub_rn(__dmul_rn((double)v[u], scale), shift);
    }
    for (; t < T; t += P)
      dst[t * out_stride] = (OutT)__dsub_rn(__dmul_rn((double)src[t * in_stride], scale), shift);
  }
}

template <typename OutT>
static int32_t launch_cmvn_rows(const float *d_in, int64_t in_stride, const int64_t *d_row_off,
                                const int64_t *d_nrows, int32_t B, int32_t C, int32_t norm_var,
                                double *d_stats, OutT *d_out, int64_t out_stride,
                                int32_t *d_zero_var, void *stream) {
  if (B < 0 || C <= 0) return invalid_post("cmvn_rows: bad B / coeff");
  if (B == 0) return PDS_OK;
  if (B > 65535) return invalid_post("cmvn_rows: B > 65535");
  if (!d_in || !d_row_off || !d_nrows || !d_stats || !d_out)
    return invalid_post("cmvn_rows: null pointer");
  hipLaunchKernelGGL(cmvn_rows_kernel<OutT>, dim3((unsigned)((C + 63) / 64), (unsigned)B),
                     dim3(kRowPhases * 64), 0, (hipStream_t)stream, d_in, in_stride, d_row_off,
                     d_nrows, C, norm_var, d_stats, d_out, out_stride, d_zero_var);
  PDS_HIP(hipGetLastError());
  return PDS_OK;
}

// The same normalisation with the sums taken by the STFT launch (pds_stft_cmvn_batch_f32: every wave walked one
// contiguous stretch of the batch's chunks and left the float64 sums of each piece of an utterance at slot
// (global wave + utterance) of `partials`): an utterance's pieces are the waves from the one holding its first
// chunk to the one holding its last, added here in that order -- deterministic -- and the rows are read ONCE.
template <typename OutT>
__global__ __launch_bounds__(kRowPhases * 64) void cmvn_rows_partials_kernel(
    const float *__restrict__ in, int64_t in_stride, const int64_t *__restrict__ row_off,
    const int64_t *__restrict__ nrows, int C, int norm_var, const int64_t *__restrict__ prefix, int B,
    const double *__restrict__ partials, int grid_waves, double *__restrict__ stats, OutT *__restrict__ out,
    int64_t out_stride, int32_t *__restrict__ zero_var) {
  constexpr int P = kRowPhases, U = PDS_CMVN_APPLY_DEPTH;
  __shared__ double sc[64], sh[64];
  const int b = blockIdx.y;
  const int lane = threadIdx.x & 63, phase = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + lane;
  const int64_t T = nrows[b];
  if (T <= 0) return;
  if (phase == 0 && c < C) {
    // the stretch of chunks wave w walked: `per` chunks, one more for the first `rem` waves (stft_wave_kernel)
    const int64_t total = prefix[B], first = prefix[b], last = prefix[b + 1] - 1;
    const int64_t per = total / grid_waves, rem = total - per * grid_waves;
    auto wave_of = [&](int64_t chunk) {
      return chunk < rem * (per + 1) ? chunk / (per + 1) : rem + (chunk - rem * (per + 1)) / (per > 0 ? per : 1);
    };
    double s1 = 0.0, s2 = 0.0;
    for (int64_t w = wave_of(first); w <= wave_of(last); ++w) {
      const double *slot = partials + (w + b) * (2 * (int64_t)C);
      s1 += slot[c];
      s2 += slot[C + c];
    }
    stats[((int64_t)b * 2 + 0) * C + c] = s1;
    stats[((int64_t)b * 2 + 1) * C + c] = s2;
    const double mean = s1 / (double)T;
    double scale = 1.0;
    if (norm_var) {
      double var = s2 / (double)T - mean * mean;
      if (fabs(var) <= 1e-8) {  // numpy.isclose(var, 0) (post.py:283)
        var = 1.0;
        if (zero_var) atomicAdd(zero_var, 1);
      }
      scale = 1.0 / sqrt(var);
    }
    sc[lane] = scale;
    sh[lane] = mean * scale;
  }
  __syncthreads();
  if (c < C) {
    const float *src = in + row_off[b] * in_stride + c;
    OutT *dst = out + row_off[b] * out_stride + c;
    const double scale = sc[lane], shift = sh[lane];
    int64_t t = phase;
    for (; t + (U - 1) * P < T; t += U * P) {
      float v[U];
#pragma unroll
      for (int u = 0; u < U; ++u) v[u] = src[(t + u * P) * in_stride];
#pragma unroll
      for (int u = 0; u < U; ++u)
        dst[(t + u * P) * out_stride] = (OutT)__dsub_rn(__dmul_rn((double)v[u], scale), shift);
    }
    for (; t < T; t += P)
      dst[t * out_stride] = (OutT)__dsub_rn(__dmul_rn((double)src[t * in_stride], scale), shift);
  }
}

int32_t launch_cmvn_rows_partials(const float *d_in, int64_t in_stride, const int64_t *d_row_off, const int64_t *d_nrows,
                                  int32_t B, int32_t C, int32_t norm_var, const int64_t *d_chunk_prefix,
                                  const double *d_partials, int32_t grid_waves, double *d_stats, void *d_out,
                                  int32_t out_is_f64, int64_t out_stride, int32_t *d_zero_var, void *stream) {
  if (B <= 0 || C <= 0 || grid_waves <= 0) return invalid_post("cmvn_rows: bad B / coeff");
  if (B > 65535) return invalid_post("cmvn_rows: B > 65535");
  if (!d_in || !d_row_off || !d_nrows || !d_stats || !d_out || !d_chunk_prefix || !d_partials)
    return invalid_post("cmvn_rows: null pointer");
  const dim3 grid((unsigned)((C + 63) / 64), (unsigned)B), block(kRowPhases * 64);
  if (out_is_f64)
    hipLaunchKernelGGL(cmvn_rows_partials_kernel<double>, grid, block, 0, (hipStream_t)stream, d_in, in_stride, d_row_off,
                       d_nrows, C, norm_var, d_chunk_prefix, B, d_partials, grid_waves, d_stats, (double *)d_out, out_stride,
                       d_zero_var);
  else
    hipLaunchKernelGGL(cmvn_rows_partials_kernel<float>, grid, block, 0, (hipStream_t)stream, d_in, in_stride, d_row_off,
                       d_nrows, C, norm_var, d_chunk_prefix, B, d_partials, grid_waves, d_stats, (float *)d_out, out_stride,
                       d_zero_var);
  PDS_HIP(hipGetLastError());
  return PDS_OK;
}

}  // namespace pds

extern "C" {

int32_t pds_deltas_f32(const float *d_in, int64_t outer, int64_t time, int64_t inner,
                       const double *d_filts, const int32_t *d_filt_off, int32_t K,
                       int32_t edge_clamp, int32_t max_off, float *d_out, int64_t out_sk,
                       int64_t out_so, int64_t out_st, int64_t out_si, void *stream) {
  return pds::launch_deltas<float>(d_in, outer, time, inner, d_filts, d_filt_off, K, edge_clamp,
                                   max_off, d_out, out_sk, out_so, out_st, out_si, stream);
}

int32_t pds_deltas_f64(const double *d_in, int64_t outer, int64_t time, int64_t inner,
                       const double *d_filts, const int32_t *d_filt_off, int32_t K,
                       int32_t edge_clamp, int32_t max_off, double *d_out, int64_t out_sk,
                       int64_t out_so, int64_t out_st, int64_t out_si, void *stream) {
  return pds::launch_deltas<double>(d_in, outer, time, inner, d_filts, d_filt_off, K,
                                    edge_clamp, max_off, d_out, out_sk, out_so, out_st, out_si,
                                    stream);
}

int32_t pds_deltas_rows_f32(const float *d_in, int64_t in_stride, const int64_t *d_row_off,
                            const int64_t *d_nrows, int32_t B, int64_t max_rows,
                            int32_t inner, const double *d_filts, const int32_t *d_filt_off,
                            int32_t K, int32_t halo, float *d_out, int64_t out_stride,
                            void *stream) {
  if (B < 0 || max_rows < 0 || inner <= 0 || K < 0)
    return pds::invalid_post("deltas_rows: bad size");
  if (B == 0 || max_rows == 0) return PDS_OK;
  if (B > 65535) return pds::invalid_post("deltas_rows: B > 65535");
  if (!d_in || !d_row_off || !d_nrows || !d_out || (K > 0 && (!d_filts || !d_filt_off)))
    return pds::invalid_post("deltas_rows: null pointer");
  if (in_stride < inner || out_stride < (int64_t)(K + 1) * inner)
    return pds::invalid_post("deltas_rows: stride too small");
  if (halo < 0) return pds::invalid_post("deltas_rows: negative halo");
  // statics already in place when the input IS the output buffer's first columns
  const int copy_statics = !(d_in == d_out && in_stride == out_stride);
  if (K >= 1 && halo >= K && halo % K == 0) {
    // the reference's filter family: register-window kernel (checks the lengths itself)
    const int W = halo / K;
    void (*kern)(const float *, int64_t, const int64_t *, const int64_t *, int, const double *,
                 const int32_t *, float *, int64_t, int) = nullptr;
#define PDS_DELTA_CASE(KK, WW) \
  if (K == KK && W == WW) kern = pds::deltas_rows_win_kernel<KK, WW>;
    PDS_DELTA_CASE(1, 1) PDS_DELTA_CASE(1, 2) PDS_DELTA_CASE(1, 3) PDS_DELTA_CASE(1, 4)
    PDS_DELTA_CASE(2, 1) PDS_DELTA_CASE(2, 2) PDS_DELTA_CASE(2, 3)
    PDS_DELTA_CASE(3, 1) PDS_DELTA_CASE(3, 2)
#undef PDS_DELTA_CASE
    const int64_t items = (max_rows + PDS_DELTA_ROWS - 1) / PDS_DELTA_ROWS * inner;
    if (kern && items < ((int64_t)1 << 39)) {
      // (a multiple of 8 workgroups per utterance: the kernel's XCD-aware numbering; the spare ones exit)
      dim3 grid((unsigned)(((items + 255) / 256 + 7) / 8 * 8), (unsigned)B);
      hipLaunchKernelGGL(kern, grid, dim3(256), 0, (hipStream_t)stream, d_in, in_stride,
                         d_row_off, d_nrows, inner, d_filts, d_filt_off, d_out, out_stride,
                         copy_statics);
      PDS_HIP(hipGetLastError());
      return PDS_OK;
    }
  }
  int rows_per_block = (24 * 1024 / 4) / inner - 2 * halo;  // ~24 KB of LDS per block
  if (rows_per_block > 64) rows_per_block = 64;
  if (rows_per_block < 1) return pds::invalid_post("deltas_rows: rows too wide for the LDS tile");
  const size_t smem = (size_t)(rows_per_block + 2 * halo) * inner * sizeof(float);
  dim3 grid((unsigned)((max_rows + rows_per_block - 1) / rows_per_block), (unsigned)B);
  hipLaunchKernelGGL(pds::deltas_rows_kernel, grid, dim3(256), smem, (hipStream_t)stream, d_in,
                     in_stride, d_row_off, d_nrows, inner, d_filts, d_filt_off, K, halo, d_out,
                     out_stride, rows_per_block, copy_statics);
  PDS_HIP(hipGetLastError());
  return PDS_OK;
}

int32_t pds_stack_rows_f32(const float *d_in, int64_t in_stride, const int64_t *d_row_off,
                          const int64_t *d_nrows, const int64_t *d_out_row_off, int32_t B,
                          int64_t max_out_rows, int32_t coeff, int32_t num_vectors,
                          int32_t pad_mode, float *d_out, int64_t out_stride, void *stream) {
  if (B < 0 || max_out_rows < 0 || coeff <= 0 || num_vectors < 1 || pad_mode < 0 || pad_mode > 2)
    return pds::invalid_post("stack_rows: bad size or pad mode");
  if (B == 0 || max_out_rows == 0) return PDS_OK;
  if (B > 65535) return pds::invalid_post("stack_rows: B > 65535");
  if (!d_in || !d_row_off || !d_nrows || !d_out_row_off || !d_out)
    return pds::invalid_post("stack_rows: null pointer");
  if (in_stride < coeff || out_stride < (int64_t)coeff * num_vectors)
    return pds::invalid_post("stack_rows: stride too small");
  const int64_t items = max_out_rows * coeff * num_vectors;
  dim3 grid((unsigned)((items + 255) / 256), (unsigned)B);
  hipLaunchKernelGGL(pds::stack_rows_kernel, grid, dim3(256), 0, (hipStream_t)stream, d_in,
                     in_stride, d_row_off, d_nrows, d_out_row_off, coeff, num_vectors, pad_mode,
                     d_out, out_stride);
  PDS_HIP(hipGetLastError());
  return PDS_OK;
}

int64_t pds_cmvn_scratch_len(int64_t coeff, int64_t inner) {
  return (int64_t)pds::kStatSlabs * coeff * inner * 2;
}

int32_t pds_cmvn_stats_f32(const float *d_in, int64_t outer, int64_t coeff, int64_t inner,
                           double *d_stats, double *d_scratch, void *stream) {
  return pds::launch_cmvn_stats<float>(d_in, outer, coeff, inner, d_stats, d_scratch, stream);
}
int32_t pds_cmvn_stats_f64(const double *d_in, int64_t outer, int64_t coeff, int64_t inner,
                           double *d_stats, double *d_scratch, void *stream) {
  return pds::launch_cmvn_stats<double>(d_in, outer, coeff, inner, d_stats, d_scratch, stream);
}
int32_t pds_cmvn_apply_f32(const float *d_in, int64_t outer, int64_t coeff, int64_t inner,
                           const double *d_scale, const double *d_shift, double *d_out,
                           void *stream) {
  return pds::launch_cmvn_apply<float>(d_in, outer, coeff, inner, d_scale, d_shift, d_out,
                                       stream);
}
int32_t pds_cmvn_apply_f64(const double *d_in, int64_t outer, int64_t coeff, int64_t inner,
                           const double *d_scale, const double *d_shift, double *d_out,
                           void *stream) {
  return pds::launch_cmvn_apply<double>(d_in, outer, coeff, inner, d_scale, d_shift, d_out,
                                        stream);
}
int32_t pds_cmvn_rows_f32(const float *d_in, int64_t in_stride, const int64_t *d_row_off,
                          const int64_t *d_nrows, int32_t B, int32_t coeff, int32_t norm_var,
                          double *d_stats, double *d_out, int64_t out_stride,
                          int32_t *d_zero_var, void *stream) {
  return pds::launch_cmvn_rows<double>(d_in, in_stride, d_row_off, d_nrows, B, coeff, norm_var,
                                       d_stats, d_out, out_stride, d_zero_var, stream);
}
int32_t pds_cmvn_rows_f32out(const float *d_in, int64_t in_stride, const int64_t *d_row_off,
                             const int64_t *d_nrows, int32_t B, int32_t coeff,
                             int32_t norm_var, double *d_stats, float *d_out,
                             int64_t out_stride, int32_t *d_zero_var, void *stream) {
  return pds::launch_cmvn_rows<float>(d_in, in_stride, d_row_off, d_nrows, B, coeff, norm_var,
                                      d_stats, d_out, out_stride, d_zero_var, stream);
}

}  // extern "C"
