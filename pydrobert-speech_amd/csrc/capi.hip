// C ABI layer of libpds_amd.so: argument checking, plan life cycle, kernel dispatch.
// Declarations and the reference interfaces they replace: include/pds_amd.h.
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>

#include "pds_internal.h"

namespace pds {

static thread_local std::string g_last_error;

void set_error(const std::string &msg) { g_last_error = msg; }

int32_t hip_fail(hipError_t err, const char *what) {
  g_last_error = std::string(what) + ": " + hipGetErrorString(err);
  return PDS_ERR_HIP;
}

static int32_t invalid(const std::string &msg) {
  g_last_error = msg;
  return PDS_ERR_INVALID;
}

int32_t check_plan_device(int plan_device, const char *what) {
  int current = -1;
  PDS_HIP(hipGetDevice(&current));
  if (current != plan_device)
    return invalid(std::string(what) + ": the plan was created on device " + std::to_string(plan_device) +
                   ", the current device is " + std::to_string(current) + " (one plan per device)");
  return PDS_OK;
}

}  // namespace pds

using pds::invalid;

extern "C" {

int32_t pds_version(void) { return 100; }

const char *pds_last_error(void) { return pds::g_last_error.c_str(); }

int32_t pds_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int32_t pds_stft_plan_create(const pds_stft_desc *desc, const double *window,
                             const int32_t *row_ptr, const int32_t *col, const double *val,
                             pds_stft_plan **plan_out) {
  if (!desc || !window || !row_ptr || !plan_out) return invalid("plan_create: null argument");
  const pds_stft_desc &d = *desc;
  if (d.frame_length < 1 || d.frame_shift < 1 || d.dft_size < d.frame_length)
    return invalid("plan_create: need frame_length >= 1, frame_shift >= 1, dft_size >= frame_length");
  if (d.pad_left < 0) return invalid("plan_create: pad_left < 0");
  if (d.num_filts < 0 || d.nnz < 0 || d.reserved != 0)
    return invalid("plan_create: bad num_filts / nnz / reserved");
  if (!(d.log_floor > 0.0)) return invalid("plan_create: log_floor must be positive");
  if (d.nnz > 0 && (!col || !val)) return invalid("plan_create: null col/val with nnz > 0");
  const int num_bins = d.dft_size / 2 + 1;  // len(rfft(n=N)) for even and odd N
  if (row_ptr[0] != 0 || row_ptr[d.num_filts] != d.nnz)
    return invalid("plan_create: row_ptr must start at 0 and end at nnz");
  for (int f = 0; f < d.num_filts; ++f)
    if (row_ptr[f + 1] < row_ptr[f]) return invalid("plan_create: row_ptr not monotone");
  for (int e = 0; e < d.nnz; ++e)
    if (col[e] < 0 || col[e] >= num_bins) return invalid("plan_create: col out of range");

  int device = -1;
  if (hipGetDevice(&device) != hipSuccess) return invalid("plan_create: no HIP device");
  pds_stft_plan *p = nullptr;
  const int32_t status = pds::no_throw("plan_create", [&]() -> int32_t {
    p = new pds_stft_plan();
    p->d = d;
    p->num_bins = num_bins;
    p->device = device;
    std::vector<float> wf(d.frame_length), vf(d.nnz);
    for (int i = 0; i < d.frame_length; ++i) wf[i] = (float)window[i];
    for (int i = 0; i < d.nnz; ++i) vf[i] = (float)val[i];
    std::vector<float2> twf(d.dft_size);
    std::vector<double2> twd(d.dft_size);
    for (int j = 0; j < d.dft_size; ++j) {
      // exact quadrant handling keeps cos/sin of multiples of pi/2 exact
      const double ang = 2.0 * M_PI * (double)j / (double)d.dft_size;
      twd[j] = make_double2(std::cos(ang), std::sin(ang));
      twf[j] = make_float2((float)twd[j].x, (float)twd[j].y);
    }
    int32_t rc = PDS_OK;
    if (rc == PDS_OK) rc = pds::upload(&p->d_window_f32, wf.data(), wf.size());
    if (rc == PDS_OK) rc = pds::upload(&p->d_window_f64, window, (size_t)d.frame_length);
    if (rc == PDS_OK) rc = pds::upload(&p->d_row_ptr, row_ptr, (size_t)d.num_filts + 1);
    if (rc == PDS_OK) rc = pds::upload(&p->d_col, col, (size_t)d.nnz);
    if (rc == PDS_OK) rc = pds::upload(&p->d_val_f32, vf.data(), vf.size());
    if (rc == PDS_OK) rc = pds::upload(&p->d_val_f64, val, (size_t)d.nnz);
    if (rc == PDS_OK) rc = pds::upload(&p->d_tw_f32, twf.data(), twf.size());
    if (rc == PDS_OK) rc = pds::upload(&p->d_tw_f64, twd.data(), twd.size());
    if (rc == PDS_OK) rc = pds::fast_tables_create(p, window, row_ptr, col, val);
    return rc;
  });
  if (status != PDS_OK) {
    pds_stft_plan_destroy(p);  // frees whatever was built (null-safe)
    return status;
  }
  *plan_out = p;
  return PDS_OK;
}

void pds_stft_plan_destroy(pds_stft_plan *p) {
  if (!p) return;
  pds::fast_tables_destroy(p);
  (void)hipFree(p->d_window_f32);
  (void)hipFree(p->d_window_f64);
  (void)hipFree(p->d_row_ptr);
  (void)hipFree(p->d_col);
  (void)hipFree(p->d_val_f32);
  (void)hipFree(p->d_val_f64);
  (void)hipFree(p->d_tw_f32);
  (void)hipFree(p->d_tw_f64);
  delete p;
}

int32_t pds_stft_num_coeffs(const pds_stft_plan *p) {
  return p ? p->d.num_filts + (p->d.include_energy ? 1 : 0) : 0;
}

int64_t pds_stft_num_frames(const pds_stft_plan *p, int64_t n) {
  if (!p || n < p->d.frame_length / 2 + 1) return 0;
  return (n + p->d.frame_shift / 2) / p->d.frame_shift;
}

int32_t pds_stft_plan_kernel_kind(const pds_stft_plan *p) { return p ? p->fast.kind : 0; }

static int32_t check_batch(const pds_stft_plan *plan, const void *sig, const int64_t *off,
                           const int64_t *len, const int64_t *nfr, const int64_t *row,
                           int32_t B, int64_t max_frames, int32_t pad_left, const void *out,
                           int64_t out_stride) {
  if (!plan) return invalid("stft_batch: null plan");
  if (B < 0 || max_frames < 0) return invalid("stft_batch: negative B / max_frames");
  if (B == 0 || max_frames == 0) return 1;  // nothing to do
  if (!sig || !off || !len || !nfr || !row || !out) return invalid("stft_batch: null pointer");
  if (out_stride < pds_stft_num_coeffs(plan)) return invalid("stft_batch: out_stride < num_coeffs");
  if (pad_left < -1) return invalid("stft_batch: pad_left < -1");
  if (B > 65535) return invalid("stft_batch: B > 65535 utterances per call (split the batch)");
  return PDS_OK;
}

#define PDS_BATCH_BODY(LAUNCH)                                                               \
  int32_t rc = check_batch(plan, d_signal, d_offsets, d_lengths, d_nframes, d_row_off, B,    \
                           max_frames, pad_left, d_out, out_stride);                         \
  if (rc == 1) return PDS_OK;                                                                \
  if (rc != PDS_OK) return rc;                                                               \
  rc = pds::check_plan_device(plan->device, "stft_batch");                                   \
  if (rc != PDS_OK) return rc;                                                               \
  pds::BatchArgs a{d_signal, d_offsets,  d_lengths,                                          \
                   d_nframes, d_row_off, B,                                                  \
                   max_frames, pad_left < 0 ? plan->d.pad_left : pad_left,                   \
                   preemph,  d_out,    out_stride, (hipStream_t)stream};                               \
  return LAUNCH(plan, a);

int32_t pds_stft_batch_f32(const pds_stft_plan *plan, const float *d_signal,
                           const int64_t *d_offsets, const int64_t *d_lengths,
                           const int64_t *d_nframes, const int64_t *d_row_off, int32_t B,
                           int64_t max_frames, int32_t pad_left, double preemph, float *d_out,
                           int64_t out_stride, void *stream) {
  PDS_BATCH_BODY((plan->fast.kind ? pds::launch_stft_fast_f32 : pds::launch_stft_generic_f32))
}

int32_t pds_stft_batch_f32_generic(const pds_stft_plan *plan, const float *d_signal,
                                   const int64_t *d_offsets, const int64_t *d_lengths,
                                   const int64_t *d_nframes, const int64_t *d_row_off,
                                   int32_t B, int64_t max_frames, int32_t pad_left,
                                   double preemph, float *d_out, int64_t out_stride, void *stream) {
  PDS_BATCH_BODY(pds::launch_stft_generic_f32)
}

int32_t pds_stft_batch_f64(const pds_stft_plan *plan, const double *d_signal,
                           const int64_t *d_offsets, const int64_t *d_lengths,
                           const int64_t *d_nframes, const int64_t *d_row_off, int32_t B,
                           int64_t max_frames, int32_t pad_left, double preemph, double *d_out,
                           int64_t out_stride, void *stream) {
  PDS_BATCH_BODY(pds::launch_stft_generic_f64)
}

int32_t pds_stft_plan_has_f64in(const pds_stft_plan *plan) { return plan && pds::fast_has_f64in(plan) ? 1 : 0; }

int32_t pds_stft_batch_f64in(const pds_stft_plan *plan, const double *d_signal, const int64_t *d_offsets,
                             const int64_t *d_lengths, const int64_t *d_nframes, const int64_t *d_row_off,
                             int32_t B, int64_t max_frames, int32_t pad_left, double preemph, void *d_out,
                             int32_t out_is_f64, int64_t out_stride, void *stream) {
  int32_t rc = check_batch(plan, d_signal, d_offsets, d_lengths, d_nframes, d_row_off, B, max_frames, pad_left,
                           d_out, out_stride);
  if (rc == 1) return PDS_OK;
  if (rc != PDS_OK) return rc;
  if (!pds::fast_has_f64in(plan)) return invalid("stft_batch_f64in: the plan has no fused float64-input kernel");
  rc = pds::check_plan_device(plan->device, "stft_batch");
  if (rc != PDS_OK) return rc;
  pds::BatchArgs a{d_signal,   d_offsets, d_lengths,  d_nframes, d_row_off, B, max_frames,
                   pad_left < 0 ? plan->d.pad_left : pad_left, preemph, d_out, out_stride, (hipStream_t)stream};
  a.in_f64 = true;
  a.out_f64 = out_is_f64 != 0;
  return pds::launch_stft_fast_f32(plan, a);
}

int32_t pds_stft_plan_has_i16in(const pds_stft_plan *plan) { return plan && pds::fast_has_f64in(plan) ? 1 : 0; }

int32_t pds_stft_batch_i16in(const pds_stft_plan *plan, const int16_t *d_signal, const int64_t *d_offsets,
                             const int64_t *d_lengths, const int64_t *d_nframes, const int64_t *d_row_off,
                             int32_t B, int64_t max_frames, int32_t pad_left, double preemph, float *d_out,
                             int64_t out_stride, void *stream) {
  int32_t rc = check_batch(plan, d_signal, d_offsets, d_lengths, d_nframes, d_row_off, B, max_frames, pad_left,
                           d_out, out_stride);
  if (rc == 1) return PDS_OK;
  if (rc != PDS_OK) return rc;
  if (!pds::fast_has_f64in(plan)) return invalid("stft_batch_i16in: the plan has no fused int16-input kernel");
  rc = pds::check_plan_device(plan->device, "stft_batch");
  if (rc != PDS_OK) return rc;
  pds::BatchArgs a{d_signal,   d_offsets, d_lengths,  d_nframes, d_row_off, B, max_frames,
                   pad_left < 0 ? plan->d.pad_left : pad_left, preemph, d_out, out_stride, (hipStream_t)stream};
  a.in_i16 = true;
  return pds::launch_stft_fast_f32(plan, a);
}

int32_t pds_stft_plan_has_fused_deltas(const pds_stft_plan *plan) {
  return plan && pds::fast_has_fused_deltas(plan) ? 1 : 0;
}

int32_t pds_stft_batch_ragged_f32(const pds_stft_plan *plan, const float *d_signal, const int64_t *d_offsets,
                                  const int64_t *d_lengths, const int64_t *d_nframes, const int64_t *d_row_off,
                                  int32_t B, int64_t max_frames, int32_t pad_left, double preemph, int64_t *d_workspace,
                                  float *d_out, int64_t out_stride, void *stream) {
  int32_t rc = check_batch(plan, d_signal, d_offsets, d_lengths, d_nframes, d_row_off, B, max_frames, pad_left,
                           d_out, out_stride);
  if (rc == 1) return PDS_OK;
  if (rc != PDS_OK) return rc;
  if (!d_workspace) return invalid("stft_batch_ragged: null workspace (B + 1 int64 on the device)");
  rc = pds::check_plan_device(plan->device, "stft_batch");
  if (rc != PDS_OK) return rc;
  pds::BatchArgs a{d_signal,   d_offsets, d_lengths,  d_nframes, d_row_off, B, max_frames,
                   pad_left < 0 ? plan->d.pad_left : pad_left, preemph, d_out, out_stride, (hipStream_t)stream};
  a.d_chunk_prefix = d_workspace;
  a.stretch = true;
  // (plans without a fused kernel take the generic one: its workgroups are frames, nothing to balance)
  return plan->fast.kind ? pds::launch_stft_fast_f32(plan, a) : pds::launch_stft_generic_f32(plan, a);
}

int32_t pds_stft_batch_ragged_i16in(const pds_stft_plan *plan, const int16_t *d_signal, const int64_t *d_offsets,
                                    const int64_t *d_lengths, const int64_t *d_nframes, const int64_t *d_row_off,
                                    int32_t B, int64_t max_frames, int32_t pad_left, double preemph, int64_t *d_workspace,
                                    float *d_out, int64_t out_stride, void *stream) {
  int32_t rc = check_batch(plan, d_signal, d_offsets, d_lengths, d_nframes, d_row_off, B, max_frames, pad_left,
                           d_out, out_stride);
  if (rc == 1) return PDS_OK;
  if (rc != PDS_OK) return rc;
  if (!pds::fast_has_f64in(plan)) return invalid("stft_batch_ragged_i16in: the plan has no fused int16-input kernel");
  if (!d_workspace) return invalid("stft_batch_ragged_i16in: null workspace (B + 1 int64 on the device)");
  rc = pds::check_plan_device(plan->device, "stft_batch");
  if (rc != PDS_OK) return rc;
  pds::BatchArgs a{d_signal,   d_offsets, d_lengths,  d_nframes, d_row_off, B, max_frames,
                   pad_left < 0 ? plan->d.pad_left : pad_left, preemph, d_out, out_stride, (hipStream_t)stream};
  a.in_i16 = true;
  a.d_chunk_prefix = d_workspace;
  a.stretch = true;
  return pds::launch_stft_fast_f32(plan, a);
}

int32_t pds_stft_deltas_batch(const pds_stft_plan *plan, const void *d_signal, int32_t signal_is_f64,
                              const int64_t *d_offsets, const int64_t *d_lengths, const int64_t *d_nframes,
                              const int64_t *d_row_off, int32_t B, int64_t max_frames, int32_t pad_left, double preemph,
                              int32_t num_deltas, int32_t context_window, const double *taps, int64_t *d_workspace,
                              int32_t workspace_prepared, float *d_out, int64_t out_stride, void *stream) {
  int32_t rc = check_batch(plan, d_signal, d_offsets, d_lengths, d_nframes, d_row_off, B, max_frames, pad_left,
                           d_out, out_stride);
  if (rc == 1) return PDS_OK;
  if (rc != PDS_OK) return rc;
  if (!pds::fast_has_fused_deltas(plan)) return invalid("stft_deltas_batch: the plan has no fused statics + deltas kernel");
  if (signal_is_f64 < 0 || signal_is_f64 > 2) return invalid("stft_deltas_batch: sample format must be 0 (float32), 1 (float64) or 2 (int16)");
  if (signal_is_f64 && !pds::fast_has_f64in(plan))
    return invalid("stft_deltas_batch: float64 / int16 samples are not served for this plan (pds_stft_plan_has_f64in)");
  if (num_deltas < 1 || num_deltas > 2 || context_window != 2 || !taps)
    return invalid("stft_deltas_batch: orders 1 and 2 with context_window 2 only (and their taps)");
  if (!d_workspace) return invalid("stft_deltas_batch: null workspace (B + 1 int64 on the device)");
  if (out_stride < (int64_t)(num_deltas + 1) * pds_stft_num_coeffs(plan))
    return invalid("stft_deltas_batch: out_stride < (num_deltas + 1) * num_coeffs");
  rc = pds::check_plan_device(plan->device, "stft_batch");
  if (rc != PDS_OK) return rc;
  pds::BatchArgs a{d_signal,   d_offsets, d_lengths,  d_nframes, d_row_off, B, max_frames,
                   pad_left < 0 ? plan->d.pad_left : pad_left, preemph, d_out, out_stride, (hipStream_t)stream};
  a.in_f64 = signal_is_f64 == 1;
  a.in_i16 = signal_is_f64 == 2;
  a.dl_K = num_deltas;
  a.d_chunk_prefix = d_workspace;
  a.prefix_prepared = workspace_prepared != 0;
  for (int j = 0; j < (num_deltas == 1 ? 5 : 14); ++j) a.dl_taps[j] = taps[j];
  return pds::launch_stft_fast_f32(plan, a);
}

int32_t pds_stft_deltas_batch_f32(const pds_stft_plan *plan, const float *d_signal, const int64_t *d_offsets,
                                  const int64_t *d_lengths, const int64_t *d_nframes, const int64_t *d_row_off,
                                  int32_t B, int64_t max_frames, int32_t pad_left, int32_t num_deltas,
                                  int32_t context_window, const double *taps, int64_t *d_workspace, float *d_out,
                                  int64_t out_stride, void *stream) {
  return pds_stft_deltas_batch(plan, d_signal, 0, d_offsets, d_lengths, d_nframes, d_row_off, B, max_frames, pad_left,
                               0.0, num_deltas, context_window, taps, d_workspace, 0, d_out, out_stride, stream);
}

int32_t pds_stft_plan_has_fused_cmvn(const pds_stft_plan *plan) { return plan && pds::fast_has_fused_cmvn(plan) ? 1 : 0; }

int64_t pds_stft_cmvn_partials_len(const pds_stft_plan *plan, int32_t B) {
  if (!plan || B < 0) return 0;
  // one slot of [2][num_coeffs] float64 per (wave of the grid + utterance): at most 16 waves per CU
  const int64_t waves = (int64_t)(plan->fast.num_cus > 0 ? plan->fast.num_cus : 256) * 16;
  return (waves + B) * 2 * (int64_t)pds_stft_num_coeffs(plan);
}

int32_t pds_stft_prepare_chunk_prefix(const pds_stft_plan *plan, const int64_t *d_nframes, int32_t B,
                                      int64_t *d_chunk_prefix, void *stream) {
  if (!plan || !d_nframes || !d_chunk_prefix || B < 0) return invalid("stft_prepare_chunk_prefix: bad argument");
  if (!plan->fast.kind || plan->fast.n2 <= 0) return invalid("stft_prepare_chunk_prefix: the plan has no fused kernel");
  int32_t rc = pds::check_plan_device(plan->device, "stft_prepare_chunk_prefix");
  if (rc != PDS_OK) return rc;
  return pds::launch_chunk_prefix(d_nframes, B, 64 / plan->fast.n2, d_chunk_prefix, (hipStream_t)stream);
}

int32_t pds_stft_cmvn_batch_f32(const pds_stft_plan *plan, const float *d_signal, const int64_t *d_offsets,
                                const int64_t *d_lengths, const int64_t *d_nframes, const int64_t *d_row_off, int32_t B,
                                int64_t max_frames, int32_t pad_left, int32_t norm_var, int64_t *d_chunk_prefix,
                                int32_t prefix_prepared, double *d_partials, int64_t partials_len, float *d_feats,
                                int64_t feats_stride, double *d_stats, void *d_out, int32_t out_is_f64, int64_t out_stride,
                                int32_t *d_zero_var, void *stream) {
  int32_t rc = check_batch(plan, d_signal, d_offsets, d_lengths, d_nframes, d_row_off, B, max_frames, pad_left,
                           d_feats, feats_stride);
  if (rc == 1) return PDS_OK;
  if (rc != PDS_OK) return rc;
  if (!pds::fast_has_fused_cmvn(plan)) return invalid("stft_cmvn_batch: the plan has no fused CMVN sums");
  const int32_t C = pds_stft_num_coeffs(plan);
  if (!d_chunk_prefix || !d_partials || partials_len < pds_stft_cmvn_partials_len(plan, B))
    return invalid("stft_cmvn_batch: chunk prefix or partial sums missing (pds_stft_cmvn_partials_len() float64)");
  if (!d_stats || !d_out || out_stride < C) return invalid("stft_cmvn_batch: null output or out_stride < num_coeffs");
  if (B > 65535) return invalid("stft_cmvn_batch: B > 65535");
  rc = pds::check_plan_device(plan->device, "stft_batch");
  if (rc != PDS_OK) return rc;
  int32_t grid_waves = 0;
  pds::BatchArgs a{d_signal,   d_offsets, d_lengths,  d_nframes, d_row_off, B, max_frames,
                   pad_left < 0 ? plan->d.pad_left : pad_left, 0.0, d_feats, feats_stride, (hipStream_t)stream};
  a.stretch = true;
  a.d_chunk_prefix = d_chunk_prefix;
  a.prefix_prepared = prefix_prepared != 0;
  a.d_stat_part = d_partials;
  a.grid_waves_out = &grid_waves;
  rc = pds::launch_stft_fast_f32(plan, a);
  if (rc != PDS_OK) return rc;
  return pds::launch_cmvn_rows_partials(d_feats, feats_stride, d_row_off, d_nframes, B, C, norm_var, d_chunk_prefix,
                                        d_partials, grid_waves, d_stats, d_out, out_is_f64, out_stride, d_zero_var, stream);
}

}  // extern "C"
