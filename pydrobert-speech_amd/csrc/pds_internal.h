// Internal declarations shared by the translation units of libpds_amd.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <exception>
#include <new>
#include <string>

#include "../../include/pds_amd.h"

namespace pds {

void set_error(const std::string &msg);
int32_t hip_fail(hipError_t err, const char *what);

#define PDS_HIP(call)                                       \
  do {                                                      \
    hipError_t _e = (call);                                 \
    if (_e != hipSuccess) return pds::hip_fail(_e, #call);  \
  } while (0)

// Tables of the fused kernel (stft_fast.hip); owned by the plan.
struct FastTables {
  int kind = 0;               // 0 = not available for this plan, else the DFT size served
  int n1 = 0, n2 = 0;         // N = n1 * n2: in-lane real DFT size x lanes per frame
  int rows = 0;               // ceil(frame_length / n2): rows of n2 samples per frame
  int num_cus = 0;            // sizes the persistent grid
  float *d_window = nullptr;  // [n2][n1] window in lane order, zero padded
  float *d_twiddle = nullptr; // [n2][n1/2] (re, im) inter-stage twiddles in lane order
  float *d_tw_special = nullptr;  // [n2] (re, im) of e^{-2 pi i r / (2 n2)}
  float *d_win_half = nullptr;    // prefetch instantiations (32 x 16): [n2][n1] window times 1/2
  float *d_tw_seed = nullptr;     // ... and [n2][3] (re, im) twiddle seeds W_N^r, W_N^4r, W_N^8r
  float *d_ell_w = nullptr;       // ELL weight rows (see stft_fast.hip)
  int32_t *d_ell_meta = nullptr;  // [ell_slots][n2] first bin | (filter + 1) << 16
  int32_t *d_ell_len = nullptr;   // [ell_slots]
  int32_t *d_ell_woff = nullptr;  // [ell_slots]
  int ell_wfloats = 0, ell_slots = 0;
  // segmented filter walk (16-lane geometries, dense banks): 0 rounds = not built for this plan
  float *d_seg_w = nullptr;        // [seg_rounds * 64][seg_len + 4]
  int32_t *d_seg_meta = nullptr;   // [seg_rounds * 64] first bins, then [num_filts] first slot | segments << 16
  int seg_rounds = 0, seg_len = 0, seg_wfloats = 0, seg_meta_ints = 0;
  // row-segment filter walk (rseg_tables.h; 16-lane geometries): 0 rounds = not built for this plan
  float *d_rs_w = nullptr;         // [rs_rounds][rs_len / 4][64][4]
  int32_t *d_rs_meta = nullptr;    // [rs_rounds * 64]
  int rs_rounds = 0, rs_len = 0, rs_wfloats = 0;
  // the same walk with the filters in numbered order along the lanes (fused statics + deltas launches)
  float *d_rsn_w = nullptr;
  int32_t *d_rsn_meta = nullptr;
  int rsn_rounds = 0, rsn_len = 0, rsn_wfloats = 0;
  int rs_eslot = -1;               // round * 64 + lane of a lane that finishes no filter (-1: every lane does)
  // matrix-pipe segment walk (mseg_tables.h; 16-lane power-of-two geometries, dense banks): 0 rounds = not built
  float *d_ms_w = nullptr;         // [ms_rounds][ms_len / 4][64][4]
  int32_t *d_ms_meta = nullptr;    // [ms_rounds * 16] first bin | flush << 15 | slot << 16, then [num_filts] first partial entry | slots << 16
  int ms_rounds = 0, ms_len = 0, ms_wfloats = 0, ms_meta_ints = 0, ms_slots = 0;
  long ms_reads = 0;
  int walk = 0;  // preferred filter walk: 0 ELL, 1 segments (dense banks), 2 row segments, 3 matrix-pipe segments
  long rs_reads = 0, rs_cost = 0, ell_reads = 0, seg_reads = 0;  // 16-byte LDS reads per lane and item of each walk
  // matrix-pipe front end (mfma_front.h; 32 x 16 geometry): device image of the tables, built for
  // the kernel instantiation of `mf_rows` rows (0 = not built: PDS_STFT_FRONT=valu, other geometry)
  float *d_mf_tab = nullptr;
  int mf_rows = 0;
};

}  // namespace pds

struct pds_stft_plan {
  pds_stft_desc d;
  int device = 0;
  int num_bins = 0;  // N/2 + 1 (even N) or (N+1)/2 (odd N): len(rfft)
  // generic kernel tables
  float *d_window_f32 = nullptr;
  double *d_window_f64 = nullptr;
  int32_t *d_row_ptr = nullptr;
  int32_t *d_col = nullptr;
  float *d_val_f32 = nullptr;
  double *d_val_f64 = nullptr;
  float2 *d_tw_f32 = nullptr;   // (cos, sin)(2 pi j / N), j in [0, N)
  double2 *d_tw_f64 = nullptr;
  pds::FastTables fast;
};

// Tables of the FFT form of the short-integration kernel (si_fft.hip); owned by the plan.
namespace pds {
struct SiFftTables {
  int blocks = 0;              // shift-sized blocks each transform yields (0: form not available)
  bool big = false;            // 2048-point transforms (one per wavefront) instead of 1024-point
  float2 *d_spectra = nullptr;  // [C][NT] filter spectra / NT
  float2 *d_twiddle = nullptr;  // [32][32] W_1024^(q * lane), row q
  float2 *d_twiddle2k = nullptr;  // [32][32] W_2048^(32 q + lane), row q
  int num_cus = 0;
};
}  // namespace pds

struct pds_si_plan {
  pds_si_desc d;
  int device = 0;
  int mpad = 0;                  // taps per filter padded to a multiple of the register block
  float *d_taps_f32 = nullptr;   // [C][mpad] (re) or [C][mpad][2] (re, im)
  double *d_taps_f64 = nullptr;
  float *d_window_f32 = nullptr;  // [2 S]
  double *d_window_f64 = nullptr;
  pds::SiFftTables fft;
};

namespace pds {

// si_fft.hip
int32_t si_fft_tables_create(pds_si_plan *plan, const double *taps);
void si_fft_tables_destroy(pds_si_plan *plan);
int64_t si_fft_scratch_len(const pds_si_plan *plan, int32_t B, int64_t max_frames);
int32_t launch_si_fft(const pds_si_plan *plan, const float *d_signal, const int64_t *d_offsets,
                      const int64_t *d_lengths, const int64_t *d_nframes, const int64_t *d_row_off,
                      int32_t B, int64_t max_frames, int64_t start, float *d_scratch, float *d_out,
                      int64_t out_stride, void *stream);

struct BatchArgs {
  const void *d_signal;
  const int64_t *d_offsets, *d_lengths, *d_nframes, *d_row_off;
  int32_t B;
  int64_t max_frames;
  int32_t pad_left;
  double preemph;
  void *d_out;
  int64_t out_stride;
  hipStream_t stream;
  // fused kernel only: float64 samples rounded at the frame load / float64 features widened at the
  // store (float32 arithmetic either way); pds_stft_batch_f64in
  bool in_f64 = false, out_f64 = false;
  // fused kernel only: int16 samples converted at the frame load (pds_stft_batch_i16in)
  bool in_i16 = false;
  // fused statics + deltas (pds_stft_deltas_batch_f32): order (0 = none) and the taps, order 1 then 2
  int dl_K = 0;
  double dl_taps[16] = {0};
  int64_t *d_chunk_prefix = nullptr;  // workspace of B + 1 entries (the utterances' chunk counts, summed)
  // ragged batches (pds_stft_batch_ragged_f32): stretch scheduling over the existing chunks, needs the workspace
  bool stretch = false;
  // CMVN sums fused with the producer (pds_stft_cmvn_batch_f32; stretch launches): the pieces' float64 sums,
  // [(grid waves + B)][2][num_coeffs], and where the launch reports its grid's waves
  double *d_stat_part = nullptr;
  int32_t *grid_waves_out = nullptr;
  // d_chunk_prefix already holds the batch's chunk prefix sums (pds_stft_prepare_chunk_prefix): no kernel in front
  bool prefix_prepared = false;
};

// stft_generic.hip
int32_t launch_stft_generic_f32(const pds_stft_plan *plan, const BatchArgs &a);
int32_t launch_stft_generic_f64(const pds_stft_plan *plan, const BatchArgs &a);
// stft_fast.hip
int32_t fast_tables_create(pds_stft_plan *plan, const double *window, const int32_t *row_ptr,
                           const int32_t *col, const double *val);
void fast_tables_destroy(pds_stft_plan *plan);
int32_t launch_stft_fast_f32(const pds_stft_plan *plan, const BatchArgs &a);
bool fast_has_f64in(const pds_stft_plan *plan);
bool fast_has_fused_deltas(const pds_stft_plan *plan);
bool fast_has_fused_cmvn(const pds_stft_plan *plan);
// chunk_prefix[b] = chunks of `groups` frames in front of utterance b, [B] = all of them (stft_fast.hip)
int32_t launch_chunk_prefix(const int64_t *d_nframes, int B, int groups, int64_t *d_prefix, hipStream_t stream);
// post.hip: per-utterance CMVN whose sums the STFT launch left as per-piece partials (pds_stft_cmvn_batch_f32)
int32_t launch_cmvn_rows_partials(const float *d_in, int64_t in_stride, const int64_t *d_row_off, const int64_t *d_nrows,
                                  int32_t B, int32_t C, int32_t norm_var, const int64_t *d_chunk_prefix,
                                  const double *d_partials, int32_t grid_waves, double *d_stats, void *d_out,
                                  int32_t out_is_f64, int64_t out_stride, int32_t *d_zero_var, void *stream);

// Runs `body` (plan construction: host allocations) so that no C++ exception crosses the C ABI.
template <typename F>
int32_t no_throw(const char *what, F &&body) {
  try {
    return body();
  } catch (const std::bad_alloc &) {
    set_error(std::string(what) + ": out of host memory");
    return PDS_ERR_NOMEM;
  } catch (const std::exception &err) {
    set_error(std::string(what) + ": " + err.what());
    return PDS_ERR_INVALID;
  }
}

// capi.hip: PDS_OK when the calling thread's current device is the one a plan's tables live on
int32_t check_plan_device(int plan_device, const char *what);

// host helper: upload a host array
template <typename T>
int32_t upload(T **dst, const T *src, size_t count) {
  *dst = nullptr;
  if (count == 0) return PDS_OK;
  PDS_HIP(hipMalloc((void **)dst, count * sizeof(T)));
  PDS_HIP(hipMemcpy(*dst, src, count * sizeof(T), hipMemcpyHostToDevice));
  return PDS_OK;
}

#ifndef PDS_PREEMPH_FMA
#define PDS_PREEMPH_FMA 1
#endif
// x[i] - c x[i-1] in the signal's own precision.  The reference's Preemphasize (pre.py:140-149) and
// pre.hip's separate pass evaluate this in float64 and cast back.  float64 signals: product rounded before
// the subtraction (the multiply goes through asm so that -ffp-contract=fast cannot fuse it), so the fused
// samples are bit-identical to theirs.  float32 signals: the float32 coefficient and ONE fused
// multiply-add (one rounding; round 1 used a rounded product and a subtraction, 25 more instructions per
// item): within about one ulp per sample of the reference's -- inside the feature tolerance (1e-4), not
// bit-identical.
__device__ __forceinline__ float preemph_sample(float cur, float prev, float c) {
#if PDS_PREEMPH_FMA
  return fmaf(-c, prev, cur);  // (one rounding: closer to the float64 value than the two of mul + sub; 25 instructions fewer per item)
#else
  float t;
  asm("v_mul_f32 %0, %1, %2" : "=v"(t) : "v"(c), "v"(prev));
  return cur - t;
#endif
}
__device__ __forceinline__ double preemph_sample(double cur, double prev, double c) {
  double t;
  asm("v_mul_f64 %0, %1, %2" : "=v"(t) : "v"(c), "v"(prev));
  return cur - t;
}

// symmetric reflection of index i into [0, n): numpy.pad(..., "symmetric") for any
// pad width (reference compute.py:600)
__device__ __forceinline__ int64_t reflect_index(int64_t i, int64_t n) {
  if (i >= 0 && i < n) return i;
  const int64_t period = 2 * n;
  int64_t m = i % period;
  if (m < 0) m += period;
  return m < n ? m : period - 1 - m;
}

}  // namespace pds
