/*
 * pds_amd.h -- C ABI of the MI355X (gfx950) STFT filter-bank feature library.
 *
 * This is the drop-in boundary for ONE hot path of sdrobert/pydrobert-speech:
 *   ShortTimeFourierTransformFrameComputer.compute_full   (reference compute.py:574-607,
 *   per-frame worker _compute_frame compute.py:388-460) and the two post-processors
 *   Deltas.apply (post.py:462-491) and Standardize/CMVN.apply (post.py:250-305).
 *
 * The reference is pure Python; it has no FFI of its own.  The entry points below are
 * what a ctypes binding inside the reference would bind (INTEGRATION.md shows the stub).
 * Conventions:
 *   - plain C types only; every `d_*` pointer is a DEVICE pointer owned by the caller;
 *   - every call that launches work takes the hipStream_t to launch on as `void *stream`
 *     (NULL = the default stream) and never synchronises, allocates or frees;
 *   - a plan owns device copies of its tables; plans are immutable after creation and may
 *     be shared between host threads and streams;
 *   - return value 0 = success, negative = error; pds_last_error() returns a thread-local
 *     human-readable message for the last failure on the calling thread.
 */
#ifndef PDS_AMD_H
#define PDS_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PDS_OK 0
#define PDS_ERR_INVALID -1 /* bad argument / unsupported configuration */
#define PDS_ERR_HIP -2     /* a HIP runtime call failed                  */
#define PDS_ERR_NOMEM -3   /* host memory exhausted while building a plan */

/* library version, major * 10000 + minor * 100 + patch */
int32_t pds_version(void);
/* 1 when the library was built with -DPDS_EXPERIMENTS=1: measured-and-rejected kernel forms (matrix-pipe front
 * end, N = 1024 as 32 x 32, sample prefetch) are present and selectable through their environment variables;
 * 0 in the product build, where those variables are ignored */
int32_t pds_build_experiments(void);
/* message describing the last error on this thread ("" if none) */
const char *pds_last_error(void);
/* number of visible HIP devices (0 when there is none; never fails) */
int32_t pds_device_count(void);

/* ---------------------------------------------------------------------------------
 * STFT filter-bank plan: everything ShortTimeFourierTransformFrameComputer.__init__
 * (reference compute.py:290-362) derives from its configuration.
 * --------------------------------------------------------------------------------- */
typedef struct pds_stft_desc {
  int32_t frame_length;   /* L: samples per frame            (compute.py:332)          */
  int32_t frame_shift;    /* S: samples between frames       (compute.py:305)          */
  int32_t dft_size;       /* N >= L: DFT length              (compute.py:344-347)      */
  int32_t pad_left;       /* left reflection: 0 causal, L/2-S/2 kaldi, (L+1)/2-1 else
                             (compute.py:582-587)                                       */
  int32_t num_filts;      /* F: rows of the bin-weight table                           */
  int32_t nnz;            /* entries of the bin-weight table                           */
  int32_t use_power;      /* 1: sum |X|^2 * w, 0: sum |X| * w (compute.py:221-226,348) */
  int32_t use_log;        /* 1: log(max(., log_floor))       (compute.py:396,458)      */
  int32_t include_energy; /* 1: column 0 = frame energy      (compute.py:392-398)      */
  int32_t reserved;       /* must be 0                                                 */
  double log_floor;       /* config.LOG_FLOOR_VALUE snapshot (config.py:52)            */
} pds_stft_desc;

typedef struct pds_stft_plan pds_stft_plan;

/*
 * window : host, double[L]                      (compute.py:343)
 * row_ptr: host, int32[F + 1]  CSR row starts   -- the per-filter loop compute.py:416-460
 * col    : host, int32[nnz]    half-spectrum bin of each entry, 0 <= col <= N/2,
 *                              already folded the way the reference walks the spectrum
 *                              (compute.py:423-455)
 * val    : host, double[nnz]   weight = (2 if bank.is_real) * sum |H_f[k]|^p over the taps
 *                              k that land on that bin (p = 2 if use_power else 1)
 * Replaces: the tables `_window`, `_filt_start_idxs`, `_truncated_filts` (compute.py:352-359).
 * A plan's tables live on the device that is current when it is created; batch calls made while
 * another device is current return PDS_ERR_INVALID (create one plan per device -- plans are
 * immutable and may be shared by the threads and streams of their device).
 */
int32_t pds_stft_plan_create(const pds_stft_desc *desc, const double *window,
                             const int32_t *row_ptr, const int32_t *col, const double *val,
                             pds_stft_plan **plan_out);
void pds_stft_plan_destroy(pds_stft_plan *plan);

/* coefficients per frame: F + include_energy (compute.py:216-218) */
int32_t pds_stft_num_coeffs(const pds_stft_plan *plan);
/* frames compute_full yields for a signal of n samples: 0 if n < L/2 + 1, else
 * (n + S/2) / S (compute.py:580-581, 596) */
int64_t pds_stft_num_frames(const pds_stft_plan *plan, int64_t n);
/* which kernel family the plan dispatches to for float32 input: the DFT size when a fused
 * LDS/register FFT geometry serves it (128 .. 2048, or an unpadded N = L of 160 .. 960),
 * 0 = generic kernels (radix-2 FFT in LDS for powers of two, direct DFT otherwise) */
int32_t pds_stft_plan_kernel_kind(const pds_stft_plan *plan);

/*
 * Batched compute_full (compute.py:574-607) over B utterances packed in one buffer.
 *
 * d_signal   : device, T[...]       all utterances, utterance b at d_signal[d_offsets[b]]
 * d_offsets  : device, int64[B]
 * d_lengths  : device, int64[B]     samples of utterance b (reflection is about its ends)
 * d_nframes  : device, int64[B]     frames to emit for b (pds_stft_num_frames for
 *                                   compute_full; the streaming host logic passes the
 *                                   count of completed frames, compute.py:480, 552-556)
 * d_row_off  : device, int64[B]     first output row of utterance b
 * max_frames : max over b of d_nframes[b] (sizes the launch grid; host knows it)
 * pad_left   : left reflection to use for this call, or -1 for the plan's
 * preemph    : 0, or the coefficient of Preemphasize (reference pre.py:103-149) applied to
 *              every utterance before framing: x'[i] = x[i] - preemph * x[i-1], x'[0] = x[0]
 *              (what the reference's drivers do before compute_full, command_line.py:346-348)
 * d_out      : device, T[rows * out_stride]  row r, coefficient c at r * out_stride + c
 * out_stride : >= num_coeffs (lets a caller leave room for Deltas in the same rows)
 *
 * Frame t of utterance b, sample j:  i = t*S - pad_left + j, reflected symmetrically
 * into [0, n) (numpy.pad(..., "symmetric"), compute.py:600).
 */
int32_t pds_stft_batch_f32(const pds_stft_plan *plan, const float *d_signal,
                           const int64_t *d_offsets, const int64_t *d_lengths,
                           const int64_t *d_nframes, const int64_t *d_row_off, int32_t B,
                           int64_t max_frames, int32_t pad_left, double preemph, float *d_out,
                           int64_t out_stride, void *stream);
/* float64 signals -> float64 features, float64 arithmetic throughout (the reference's
 * internal precision, compute.py:403-414); always the generic kernel */
int32_t pds_stft_batch_f64(const pds_stft_plan *plan, const double *d_signal,
                           const int64_t *d_offsets, const int64_t *d_lengths,
                           const int64_t *d_nframes, const int64_t *d_row_off, int32_t B,
                           int64_t max_frames, int32_t pad_left, double preemph, double *d_out,
                           int64_t out_stride, void *stream);
/* float64 signals, float32 ARITHMETIC (the fused kernel): every sample is rounded to float32 as its
 * frame is loaded (a fused pre-emphasis is applied before that, in float64, so the samples framed are
 * bit-identical to the reference's own pass, pre.py:140-149), features are float32 (out_is_f64 = 0)
 * or widened to float64 at the store (out_is_f64 = 1: compute_full's "output dtype = input dtype",
 * compute.py:601; not together with preemph != 0).  This is the dtype flow of the reference's drivers
 * -- float64 audio in, float32 features stored (command_line.py:107-108, 345-350) -- without a
 * conversion pass over the signal.  Served for plans of the common power-of-two transform sizes
 * (256, 512, 1024, 2048) whose filter tables are LDS-resident: pds_stft_plan_has_f64in(); results
 * are within the float32 tolerance (1e-5 + 1e-4 |ref|), not the 1e-9 of pds_stft_batch_f64. */
int32_t pds_stft_plan_has_f64in(const pds_stft_plan *plan);
int32_t pds_stft_batch_f64in(const pds_stft_plan *plan, const double *d_signal,
                             const int64_t *d_offsets, const int64_t *d_lengths,
                             const int64_t *d_nframes, const int64_t *d_row_off, int32_t B,
                             int64_t max_frames, int32_t pad_left, double preemph, void *d_out,
                             int32_t out_is_f64, int64_t out_stride, void *stream);
/* int16 samples (PCM as a WAV file holds it; what scipy.io.wavfile / wave hand the reference's readers before
 * their .astype(dtype), util.py:207-235) -> float32 features: every sample is converted to float32 as its frame
 * is loaded (exact), then the arithmetic of pds_stft_batch_f32 -- a fused pre-emphasis included -- so the result
 * equals pds_stft_batch_f32 on the converted signal.  Half the bytes of float32 samples, a quarter of the
 * reference drivers' float64 ones, over PCIe and out of HBM; no conversion pass on the host or the device.
 * Served for the plans pds_stft_plan_has_f64in() serves (pds_stft_plan_has_i16in()). */
int32_t pds_stft_plan_has_i16in(const pds_stft_plan *plan);
int32_t pds_stft_batch_i16in(const pds_stft_plan *plan, const int16_t *d_signal,
                             const int64_t *d_offsets, const int64_t *d_lengths,
                             const int64_t *d_nframes, const int64_t *d_row_off, int32_t B,
                             int64_t max_frames, int32_t pad_left, double preemph, float *d_out,
                             int64_t out_stride, void *stream);
/* pds_stft_batch_f32 (reference compute.py:574-607, compute_full per utterance) for RAGGED batches: the same kernels, but every wave walks one contiguous stretch of the
 * chunks that exist (the utterances' chunk counts are summed into `d_workspace`, B + 1 int64 on the device, by
 * a small kernel in front) instead of the waves being dealt (utterance, chunk < chunks of the longest) pairs of
 * which short utterances have none: lengths uniform in 1 ... 15 s run 9 % slower per frame than equal lengths
 * with the plain call, as fast with this one.  Same values bit for bit.  With fused pre-emphasis, or a plan
 * without a fused kernel, it is the plain call. */
int32_t pds_stft_batch_ragged_f32(const pds_stft_plan *plan, const float *d_signal,
                                  const int64_t *d_offsets, const int64_t *d_lengths,
                                  const int64_t *d_nframes, const int64_t *d_row_off, int32_t B,
                                  int64_t max_frames, int32_t pad_left, double preemph,
                                  int64_t *d_workspace, float *d_out, int64_t out_stride, void *stream);
/* ... for int16 samples (pds_stft_batch_i16in's values).  Stretch scheduling serves float32 samples of every fused
 * plan whose tables are LDS-resident; with a fused pre-emphasis, and for int16 samples (with or without one), the
 * row-segment kernels of the transform sizes 512 and 1024 (mel-like banks) -- other plans and calls run the
 * round-robin order of pds_stft_batch_f32 / _i16in through these entry points, with the same values. */
int32_t pds_stft_batch_ragged_i16in(const pds_stft_plan *plan, const int16_t *d_signal,
                                    const int64_t *d_offsets, const int64_t *d_lengths,
                                    const int64_t *d_nframes, const int64_t *d_row_off, int32_t B,
                                    int64_t max_frames, int32_t pad_left, double preemph,
                                    int64_t *d_workspace, float *d_out, int64_t out_stride, void *stream);
/* Statics AND deltas in one launch (BASELINE.json configs[2]; reference post.py:462-491 applied to
 * compute_full's output along time, "edge" padding): row r of utterance b receives the num_coeffs
 * statics at columns [0, C) and the order-k deltas at [k C, (k + 1) C), k = 1 .. num_deltas.  The
 * statics make no second trip through HBM and no second kernel runs: every wave walks a contiguous
 * stretch of frames and differentiates them from the coefficients it holds in registers
 * (stft_fast.hip, DLT).  num_deltas 1 or 2, context_window 2; `taps`: HOST array of the filters
 * exactly as the caller's Deltas object holds them -- 5 taps of order 1, then (if num_deltas = 2) the
 * 9 taps of order 2 (numpy.convolve of the ramp with itself, post.py:456-460).  `d_workspace`:
 * B + 1 int64 on the device (the utterances' chunk counts are summed up there by a small kernel in
 * front of the main one, on `stream`).  The statics equal pds_stft_batch_f32's bit for bit; the
 * deltas are formed in float32 where pds_deltas_rows_f32 (and the reference) accumulate in float64
 * and round: they agree within a few float32 ulps of the statics, i.e. inside the feature tolerance.
 * Served for plans with transform sizes 512 and 1024 whose banks run the row-segment filter walk in
 * at most two rounds (mel banks of up to ~100 filters): pds_stft_plan_has_fused_deltas().  (The statics are
 * bit-identical to pds_stft_batch_f32's when that launch takes the row-segment walk too; a plan that prefers
 * another walk for the plain launch sums the same products in another order: a few float32 ulps.) */
int32_t pds_stft_plan_has_fused_deltas(const pds_stft_plan *plan);
/* ... with everything the reference's drivers put in front of compute_full (command_line.py:345-350): `d_signal`
 * float32 (signal_is_f64 = 0 = PDS_SAMPLES_F32), float64 (1: plans with pds_stft_plan_has_f64in(); rounded to float32
 * as the frame is loaded, as in pds_stft_batch_f64in) or int16 samples (2 = PDS_SAMPLES_I16: same plans; converted as
 * the frame is loaded, as in pds_stft_batch_i16in) and a fused pre-emphasis (`preemph` != 0, pre.py:140-149: in the
 * signal's own precision, before the rounding).  float64 audio -> Preemphasize -> compute_full -> Deltas in one
 * launch; features float32.  workspace_prepared != 0: d_workspace already holds the batch's chunk prefix sums
 * (pds_stft_prepare_chunk_prefix(), see pds_stft_cmvn_batch_f32) and no kernel runs in front. */
int32_t pds_stft_deltas_batch(const pds_stft_plan *plan, const void *d_signal, int32_t signal_is_f64,
                              const int64_t *d_offsets, const int64_t *d_lengths, const int64_t *d_nframes,
                              const int64_t *d_row_off, int32_t B, int64_t max_frames, int32_t pad_left,
                              double preemph, int32_t num_deltas, int32_t context_window, const double *taps,
                              int64_t *d_workspace, int32_t workspace_prepared, float *d_out, int64_t out_stride,
                              void *stream);
/* (float32 samples, no pre-emphasis: the round-2 signature) */
int32_t pds_stft_deltas_batch_f32(const pds_stft_plan *plan, const float *d_signal,
                                  const int64_t *d_offsets, const int64_t *d_lengths,
                                  const int64_t *d_nframes, const int64_t *d_row_off, int32_t B,
                                  int64_t max_frames, int32_t pad_left, int32_t num_deltas,
                                  int32_t context_window, const double *taps, int64_t *d_workspace,
                                  float *d_out, int64_t out_stride, void *stream);
/* compute_full followed by per-utterance CMVN (reference post.py:250-295, "local" statistics; BASELINE.json
 * configs[4]) with the SUMS taken by the STFT launch: every wave walks one contiguous stretch of the batch's chunks
 * (as in pds_stft_batch_ragged_f32), adds the coefficients it stores to float64 sums of its own and leaves them per
 * piece of an utterance in the workspace; the normalising kernel adds an utterance's pieces in wave order
 * (deterministic, like pds_cmvn_rows_f32's fixed-order sums, though not the same order) and reads the float32
 * features ONCE: x * scale - shift into d_out (float64 like the reference's, or float32).  d_feats receives the
 * float32 features (num_coeffs columns, row stride feats_stride), d_stats [B][2][num_coeffs] the sums.
 * d_chunk_prefix: B + 1 int64 on the device -- filled by a small kernel in front of the main one, or, with
 * prefix_prepared != 0, already holding what pds_stft_prepare_chunk_prefix() left there for this batch (the prefix
 * sums depend on d_nframes only: prepare them once with the batch's other index arrays, share them between launches).
 * d_partials: scratch of pds_stft_cmvn_partials_len(plan, B) float64, rewritten by every launch.  Served for the
 * 16-lane power-of-two geometries (N = 512, 1024) whose plan takes a segment walk with registers to spare, when the
 * waves' sums fit in LDS beside its tables (else PDS_ERR_INVALID: use pds_stft_batch_f32 + pds_cmvn_rows_f32);
 * float32 samples, no fused pre-emphasis. */
int32_t pds_stft_plan_has_fused_cmvn(const pds_stft_plan *plan);
int64_t pds_stft_cmvn_partials_len(const pds_stft_plan *plan, int32_t B);
int32_t pds_stft_prepare_chunk_prefix(const pds_stft_plan *plan, const int64_t *d_nframes, int32_t B,
                                      int64_t *d_chunk_prefix, void *stream);
int32_t pds_stft_cmvn_batch_f32(const pds_stft_plan *plan, const float *d_signal, const int64_t *d_offsets,
                                const int64_t *d_lengths, const int64_t *d_nframes, const int64_t *d_row_off,
                                int32_t B, int64_t max_frames, int32_t pad_left, int32_t norm_var,
                                int64_t *d_chunk_prefix, int32_t prefix_prepared, double *d_partials,
                                int64_t partials_len, float *d_feats, int64_t feats_stride, double *d_stats,
                                void *d_out, int32_t out_is_f64, int64_t out_stride, int32_t *d_zero_var,
                                void *stream);
/* float32 input through the generic kernels regardless of N: radix-2 FFT in LDS for powers of two,
 * direct DFT otherwise (cross-check of the fused kernel; also what sizes without a fused geometry use) */
int32_t pds_stft_batch_f32_generic(const pds_stft_plan *plan, const float *d_signal,
                                   const int64_t *d_offsets, const int64_t *d_lengths,
                                   const int64_t *d_nframes, const int64_t *d_row_off,
                                   int32_t B, int64_t max_frames, int32_t pad_left, double preemph,
                                   float *d_out, int64_t out_stride, void *stream);

/* ---------------------------------------------------------------------------------
 * Host feed: utterances in HOST memory through a plan and back (csrc/feed.hip).
 *
 * The reference's callers hold their audio on the host -- compute_full(signal) takes a numpy array
 * (compute.py:574), signals-to-torch-feat-dir reads files (command_line.py:337-607).  For them the rate of
 * the path is what it makes of PCIe: the kernel needs 0.27 ms for a batch the link needs ~12 ms to deliver.
 * A feed is a ring of `slots` staging slots on the current device -- pinned host buffers for samples and
 * features, their device twins, one stream each -- so that batch k + 1 uploads while batch k computes and
 * batch k - 1 downloads.  Samples travel as they are stored (PDS_SAMPLES_I16: 2 bytes per sample, converted as
 * a frame is loaded; PDS_SAMPLES_F64: the reference drivers' float64, rounded as a frame is loaded; a plan whose
 * launch does not serve the format -- no fused kernel for the transform size, a filter table too large for LDS --
 * gets its samples widened to float32 by a device pass in front of the float32 launch instead).
 *
 *   pds_feed_acquire  the ring's next slot (blocks until its previous batch was released) and the pinned host
 *                     buffer the caller -- its reader threads -- fills with the batch's utterances back to back
 *   pds_feed_pack     optional: fills that buffer from one host pointer per utterance with `threads` copying
 *                     threads (a single memcpy stream does not keep up with the link)
 *   pds_feed_submit   lengths of the utterances packed there; queues upload + kernel (+ download of the
 *                     features when `download` != 0) on the slot's stream and returns at once
 *   pds_feed_device_view / pds_feed_download
 *                     after a submit with download = 0: the device features, the slot's stream -- queue
 *                     post-processors (pds_deltas_rows_f32, pds_cmvn_rows_f32out, ...) on it -- then the download of
 *                     whatever device buffer holds the final rows (at most slot_rows x feature_cols float32)
 *   pds_feed_collect  blocks until the slot's download has arrived: float32 rows in pinned host memory and the
 *                     utterances' n_utts + 1 row offsets (both valid until pds_feed_release)
 *   pds_feed_unpack   optional: the collected features copied out of the pinned buffer by `threads` copying threads
 *   pds_feed_release  gives the slot back to the ring
 * Slots are acquired, and must be collected and released, in ring order.  One thread may feed (acquire / pack /
 * submit) while another drains (collect / release).
 * --------------------------------------------------------------------------------- */
#define PDS_SAMPLES_F32 0
#define PDS_SAMPLES_F64 1
#define PDS_SAMPLES_I16 2
typedef struct pds_feed pds_feed;
/* slot_samples / slot_utts: capacity of one slot; a slot holds slot_rows = slot_samples / frame_shift + slot_utts
 * rows.  feature_cols: float32 columns per row the slot's feature buffers have room for (0 or less than the
 * plan's num_coeffs: num_coeffs) -- more when a post-processor queued on the slot's stream widens the rows
 * (Deltas: (K + 1) num_coeffs) and its result is what pds_feed_download sends back */
int32_t pds_feed_create(const pds_stft_plan *plan, int32_t sample_format, int64_t slot_samples, int32_t slot_utts,
                        int32_t slots, int32_t feature_cols, pds_feed **feed_out);
void pds_feed_destroy(pds_feed *feed);
int64_t pds_feed_slot_rows(const pds_feed *feed);
/* direct != 0 (the default for float32 and int16 samples): the kernel reads the samples from the slot's pinned host buffer and writes the features
 * to its pinned host buffer itself, so upload and download run concurrently, driven by the kernel's own loads and
 * stores (with download = 0 in pds_feed_submit the features stay in device memory for the post-processors);
 * direct = 0 (the default for float64 samples): staged -- DMA upload into device memory, kernel, DMA download.
 * Measured on 1024 x 10 s at 16 kHz: int16 samples 6.4 ms direct against 8.8 ms staged, float32 13.9 against 14.5
 * (the two DMA directions take turns on the measured platform), float64 29.9 against 26.3 (the kernel's 16-byte pair
 * loads do worse over the link than the DMA engine).  Not while a batch is in flight. */
int32_t pds_feed_set_direct(pds_feed *feed, int32_t direct);
int32_t pds_feed_acquire(pds_feed *feed, int32_t *slot_out, void **h_samples_out);
int32_t pds_feed_pack(pds_feed *feed, int32_t slot, const void *const *signals, const int64_t *lengths, int32_t n_utts,
                      int32_t threads);
int32_t pds_feed_submit(pds_feed *feed, int32_t slot, const int64_t *lengths, int32_t n_utts, double preemph,
                        int32_t download);
/* ... with the frame counts and the left reflection of the call given, as pds_stft_batch_* take them (nframes NULL:
 * pds_stft_num_frames of every length; pad_left -1: the plan's): what the streaming host logic needs, whose chunks
 * yield the frames completed so far (compute.py:480, 552-556) */
int32_t pds_feed_submit_frames(pds_feed *feed, int32_t slot, const int64_t *lengths, const int64_t *nframes,
                               int32_t n_utts, int32_t pad_left, double preemph, int32_t download);
int32_t pds_feed_device_view(pds_feed *feed, int32_t slot, void **d_features, int64_t *rows, const int64_t **row_offsets,
                             void **stream);
int32_t pds_feed_download(pds_feed *feed, int32_t slot, const void *d_src, int64_t bytes);
int32_t pds_feed_collect(pds_feed *feed, int32_t slot, const float **h_features, const int64_t **row_offsets,
                         int64_t *rows);
/* optional, between collect and release: the first `bytes` of the slot's features copied to ordinary host memory
 * by `threads` copying threads */
int32_t pds_feed_unpack(pds_feed *feed, int32_t slot, void *dst, int64_t bytes, int32_t threads);
int32_t pds_feed_release(pds_feed *feed, int32_t slot);

/* ---------------------------------------------------------------------------------
 * Pre-processors as separate passes (reference pre.py:67-149); `preemph` above fuses the
 * first one into the frame load instead.
 * pds_preemphasize: per utterance of a packed buffer (offsets/lengths as in pds_stft_batch),
 *   out[i] = in[i] - coeff * in[i-1], out[0] = in[0]; float64 intermediate, d_out != d_in.
 * pds_dither: out[i] = in[i] + N(0, coeff^2) noise from a counter-based generator
 *   (Philox4x32-10 keyed by `seed`, Box-Muller); d_out may equal d_in.
 * --------------------------------------------------------------------------------- */
int32_t pds_preemphasize_f32(const float *d_in, const int64_t *d_offsets, const int64_t *d_lengths,
                             int32_t B, int64_t max_len, double coeff, float *d_out, void *stream);
int32_t pds_preemphasize_f64(const double *d_in, const int64_t *d_offsets,
                             const int64_t *d_lengths, int32_t B, int64_t max_len, double coeff,
                             double *d_out, void *stream);
int32_t pds_dither_f32(const float *d_in, int64_t total, double coeff, uint64_t seed, float *d_out,
                       void *stream);
int32_t pds_dither_f64(const double *d_in, int64_t total, double coeff, uint64_t seed,
                       double *d_out, void *stream);

/* ---------------------------------------------------------------------------------
 * Deltas.apply (reference post.py:462-491): correlation along a "time" axis of a
 * tensor viewed as [outer, time, inner], float64 accumulation, result cast back.
 *
 * d_in     : device, T[outer * time * inner]
 * d_filts  : device, double[sum(filt_len)]   the filters of post.py:455-460, order 1..K,
 *                                            concatenated
 * d_filt_off: device, int32[K + 1]           start of each filter in d_filts
 * edge_clamp: 1 = index clip(t + j, 0, time - 1) ("edge" padding, post.py:447);
 *             0 = the caller already padded `time` by max_off on both sides of every
 *                 slice (any other numpy.pad mode) and `time` is the unpadded length
 * Output element (k, o, t, i), k = 0..K (k = 0 copies the input), is written to
 *   d_out[k*out_sk + o*out_so + t*out_st + i*out_si]
 * so one call can write the concatenated or the stacked layout of post.py:488-491.
 * --------------------------------------------------------------------------------- */
int32_t pds_deltas_f32(const float *d_in, int64_t outer, int64_t time, int64_t inner,
                       const double *d_filts, const int32_t *d_filt_off, int32_t K,
                       int32_t edge_clamp, int32_t max_off, float *d_out, int64_t out_sk,
                       int64_t out_so, int64_t out_st, int64_t out_si, void *stream);
int32_t pds_deltas_f64(const double *d_in, int64_t outer, int64_t time, int64_t inner,
                       const double *d_filts, const int32_t *d_filt_off, int32_t K,
                       int32_t edge_clamp, int32_t max_off, double *d_out, int64_t out_sk,
                       int64_t out_so, int64_t out_st, int64_t out_si, void *stream);
/* ragged batch of [time_b, inner] feature matrices stored row-wise in one buffer (the
 * layout pds_stft_batch writes): utterance b occupies rows d_row_off[b] .. + d_nrows[b];
 * deltas never cross an utterance boundary.  in/out strides are per row; `halo` is the reach
 * of the longest filter, (len - 1) / 2.  d_in may be the first `inner` columns of d_out
 * (d_in == d_out, equal strides): the statics then stay where they are. */
int32_t pds_deltas_rows_f32(const float *d_in, int64_t in_stride, const int64_t *d_row_off,
                            const int64_t *d_nrows, int32_t B, int64_t max_rows,
                            int32_t inner, const double *d_filts, const int32_t *d_filt_off,
                            int32_t K, int32_t halo, float *d_out, int64_t out_stride,
                            void *stream);

/* Stack (reference post.py:494-563) on a packed ragged batch: output row t' of utterance b is
 * its input rows t' * num_vectors .. t' * num_vectors + num_vectors - 1 side by side, written at
 * row d_out_row_off[b] + t' of d_out.  pad_mode 0 drops the incomplete last group (the
 * reference's default), 1 completes it with zeros (numpy.pad "constant"), 2 repeats the last
 * row ("edge").  max_out_rows bounds the output rows of any one utterance. */
int32_t pds_stack_rows_f32(const float *d_in, int64_t in_stride, const int64_t *d_row_off,
                           const int64_t *d_nrows, const int64_t *d_out_row_off, int32_t B,
                           int64_t max_out_rows, int32_t coeff, int32_t num_vectors,
                           int32_t pad_mode, float *d_out, int64_t out_stride, void *stream);

/* ---------------------------------------------------------------------------------
 * Standardize / CMVN (reference post.py:193-212 accumulate, 250-295 apply) on a tensor
 * viewed as [outer, coeff, inner]; statistics are over outer x inner per coefficient.
 * --------------------------------------------------------------------------------- */
/* d_stats: device, double[2 * C]: sum x (first C) and sum x^2 (next C); overwritten.
 * d_scratch: device, double[pds_cmvn_scratch_len(C, inner)]; sums are formed in a fixed
 * order (two-stage), so results are bitwise reproducible */
int64_t pds_cmvn_scratch_len(int64_t coeff, int64_t inner);
int32_t pds_cmvn_stats_f32(const float *d_in, int64_t outer, int64_t coeff, int64_t inner,
                           double *d_stats, double *d_scratch, void *stream);
int32_t pds_cmvn_stats_f64(const double *d_in, int64_t outer, int64_t coeff, int64_t inner,
                           double *d_stats, double *d_scratch, void *stream);
/* out[o, c, i] = in[o, c, i] * d_scale[c] - d_shift[c]   (post.py:293-294), float64 out */
int32_t pds_cmvn_apply_f32(const float *d_in, int64_t outer, int64_t coeff, int64_t inner,
                           const double *d_scale, const double *d_shift, double *d_out,
                           void *stream);
int32_t pds_cmvn_apply_f64(const double *d_in, int64_t outer, int64_t coeff, int64_t inner,
                           const double *d_scale, const double *d_shift, double *d_out,
                           void *stream);
/* per-utterance ("local", post.py:278-281) CMVN over the ragged row layout: one pass
 * of statistics, one of normalisation; variance within 1e-8 of 0 is replaced by 1
 * (post.py:283-286) and counted in *d_zero_var (device int32, may be NULL).
 * d_out is float64 like the reference's, or float32 with the _f32out variant. */
int32_t pds_cmvn_rows_f32(const float *d_in, int64_t in_stride, const int64_t *d_row_off,
                          const int64_t *d_nrows, int32_t B, int32_t coeff, int32_t norm_var,
                          double *d_stats /* [B][2][coeff] scratch+result */,
                          double *d_out, int64_t out_stride, int32_t *d_zero_var,
                          void *stream);
int32_t pds_cmvn_rows_f32out(const float *d_in, int64_t in_stride, const int64_t *d_row_off,
                             const int64_t *d_nrows, int32_t B, int32_t coeff,
                             int32_t norm_var, double *d_stats, float *d_out,
                             int64_t out_stride, int32_t *d_zero_var, void *stream);

/* ---------------------------------------------------------------------------------
 * ShortIntegrationFrameComputer (reference compute.py:613-996):
 *   y_f[i] = sum_{k < M} taps[f][k] * sig[i + start - k]          (sig = 0 outside the utterance)
 *   out[t][f] = log(max(sum_{m < 2S} window[m] * |y_f[t S + m]|^2 (or | |), log_floor))
 * The plan holds what the reference derives in __init__ (compute.py:660-744): the translated
 * impulse responses clamped to the longest support M (the "dirac" energy filter first when the
 * computer includes energy, compute.py:701-710) and the 2S-sample integration window.
 * --------------------------------------------------------------------------------- */
typedef struct pds_si_desc {
  int32_t frame_shift;   /* S (compute.py:674)                                         */
  int32_t max_support;   /* M (compute.py:682-697)                                     */
  int32_t num_coeffs;    /* filters + energy                                           */
  int32_t taps_complex;  /* taps are (re, im) pairs (bank not real, compute.py:677)    */
  int32_t use_power;     /* |y|^2 else |y| (compute.py:909-912)                        */
  int32_t use_log;       /* compute.py:989-990                                         */
  int32_t reserved;      /* must be 0                                                  */
  int32_t reserved2;     /* must be 0                                                  */
  double log_floor;      /* config.LOG_FLOOR_VALUE snapshot                            */
} pds_si_desc;

typedef struct pds_si_plan pds_si_plan;

/* taps: host, double[num_coeffs][M] or double[num_coeffs][M][2]; window: host, double[2 S] */
int32_t pds_si_plan_create(const pds_si_desc *desc, const double *taps, const double *window,
                           pds_si_plan **plan_out);
void pds_si_plan_destroy(pds_si_plan *plan);

/* Batched compute_full (compute.py:852-855): arguments as pds_stft_batch_*.  d_nframes[b] is the
 * number of frames to emit for utterance b (the host restates the reference's block arithmetic,
 * compute.py:781-850); `start` is the stream position of the first integrated sample
 * (skipped samples minus virtual leading zeros, compute.py:859-865) -- a caller that streams
 * passes start + t0 * S to continue at frame t0. */
/* float32 has two forms: overlap-save with 1024- or 2048-point FFTs (filter supports up to 2048 - S taps)
 * when d_scratch points to pds_si_scratch_len(plan, B, max_frames) floats of device memory, and
 * direct time-domain filtering when d_scratch is NULL (or the supports are too long: the length
 * is then 0).  float64 signals always take the direct form. */
int64_t pds_si_scratch_len(const pds_si_plan *plan, int32_t B, int64_t max_frames);
/* transform size of the plan's FFT form: 1024, 2048 (supports up to 2048 - S taps; also chosen
 * for shorter supports when it wastes less of a transform on the overlap), or 0 when the supports
 * are too long for it and float32 takes the direct form as well */
int32_t pds_si_plan_fft_size(const pds_si_plan *plan);
int32_t pds_si_batch_f32(const pds_si_plan *plan, const float *d_signal, const int64_t *d_offsets,
                         const int64_t *d_lengths, const int64_t *d_nframes,
                         const int64_t *d_row_off, int32_t B, int64_t max_frames, int64_t start,
                         float *d_scratch, float *d_out, int64_t out_stride, void *stream);
int32_t pds_si_batch_f64(const pds_si_plan *plan, const double *d_signal, const int64_t *d_offsets,
                         const int64_t *d_lengths, const int64_t *d_nframes,
                         const int64_t *d_row_off, int32_t B, int64_t max_frames, int64_t start,
                         double *d_out, int64_t out_stride, void *stream);

/* ---------------------------------------------------------------------------------
 * Multi-GPU: gather of feature rows over RCCL (xGMI inside a node).
 *
 * The reference has no counterpart: its only parallelism is DataLoader worker processes
 * (command_line.py:594).  The path shards by utterance with no data dependence (SURVEY.md
 * section 8 e); what a caller that wants every feature matrix in one place needs is ONE
 * all-gather of the rows each GPU produced.  RCCL (librccl.so.1) is loaded on first use:
 * the other entry points of this library do not need it.
 *
 * Two ways to make communicators:
 *   - one process per GPU: rank 0 calls pds_comm_unique_id and hands the 128 bytes to the
 *     other ranks (any channel: a file, MPI, torch.distributed's store); every rank then calls
 *     pds_comm_init_rank with its own current device;
 *   - one process driving several GPUs: pds_comm_init_all (ncclCommInitAll), one handle per
 *     device; calls on several handles from one thread go between pds_comm_group_start / _end.
 * --------------------------------------------------------------------------------- */
typedef struct pds_comm pds_comm;

#define PDS_COMM_ID_BYTES 128
int32_t pds_comm_unique_id(void *id128);
int32_t pds_comm_init_rank(const void *id128, int32_t world, int32_t rank, pds_comm **comm_out);
/* comms_out: array of ndev handles; devices: ndev HIP device ordinals (NULL = 0 .. ndev - 1) */
int32_t pds_comm_init_all(int32_t ndev, const int32_t *devices, pds_comm **comms_out);
int32_t pds_comm_world(const pds_comm *comm);
int32_t pds_comm_rank(const pds_comm *comm);
void pds_comm_destroy(pds_comm *comm);
int32_t pds_comm_group_start(void);
int32_t pds_comm_group_end(void);

/*
 * All-gather of rows: rank r contributes rows_per_rank[r] rows of row_bytes bytes from its
 * d_local; every rank receives all rows, in rank order (= utterance order for contiguous
 * utterance blocks), in its d_out (sum(rows_per_rank) * row_bytes bytes; d_local may be the
 * rank's own slice of d_out).  rows_per_rank: HOST array of `world` entries, the same on every
 * rank (frame counts are a pure function of the utterance lengths, compute.py:596).  Equal
 * shards are one ncclAllGather; ragged shards one group of ncclBroadcast, one per rank.
 * Enqueued on `stream`; nothing here synchronises.
 */
int32_t pds_gather_rows(pds_comm *comm, const void *d_local, const int64_t *rows_per_rank,
                        int64_t row_bytes, void *d_out, void *stream);
/* sum over ranks of a small table, in place (corpus-level CMVN statistics over a sharded
 * corpus: float64[2][C + 1], post.py:193-212) */
int32_t pds_allreduce_sum_f64(pds_comm *comm, double *d_table, int64_t count, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* PDS_AMD_H */
