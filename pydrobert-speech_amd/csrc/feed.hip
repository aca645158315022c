// Host feed: utterances that live in HOST memory through a plan's fused STFT kernel and back, as a ring of
// staging slots so that one batch's upload, the kernel of the batch before and the download of the one before
// that overlap (three engines: H2D copy, compute, D2H copy).  The kernel itself takes 0.27 ms for what PCIe needs
// ~12 ms to deliver (1024 x 10 s of float32 samples), so for host-resident audio -- every caller of the
// reference's compute_full and its signals-to-torch-feat-dir tool (command_line.py:337-607) -- the rate is what
// this file makes of the link: pinned staging on both sides, no synchronisation between batches, int16 samples
// kept as int16 until a frame is loaded.  Declarations: include/pds_amd.h ("host feed").
#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstring>
#include <mutex>
#include <thread>
#include <vector>

#include "pds_internal.h"

namespace {

enum SlotState { FREE = 0, ACQUIRED = 1, SUBMITTED = 2, COLLECTED = 3 };

struct Slot {
  void *h_samples = nullptr;     // pinned, slot_samples * sample_bytes
  void *d_samples = nullptr;
  int64_t *h_index = nullptr;    // pinned, [4][slot_utts]: offsets, lengths, frames, first rows
  int64_t *d_index = nullptr;
  int64_t *h_rows = nullptr;     // host (not pinned), [slot_utts + 1] row offsets handed to the caller
  int64_t *d_work = nullptr;     // [min(slot_utts, 65535) + 1] chunk prefix sums of the ragged launch
  float *d_feats = nullptr;      // slot_rows * out_cols
  float *d_wide = nullptr;       // slot_samples float32: the samples widened on the device (plans whose launch does
                                 // not serve the slot's sample format, see pds_feed_submit); allocated on first use
  float *h_feats = nullptr;      // pinned
  void *hd_samples = nullptr;    // the pinned buffers as the device addresses them (direct mode)
  float *hd_feats = nullptr;
  hipStream_t stream = nullptr;
  hipEvent_t done = nullptr;
  int state = FREE;
  int32_t n_utts = 0;
  int64_t rows = 0, samples = 0;
};

}  // namespace

struct pds_feed {
  const pds_stft_plan *plan = nullptr;
  int device = 0;
  int32_t format = 0, sample_bytes = 4, slot_utts = 0, num_slots = 0, out_cols = 0, feature_cols = 0;
  int64_t slot_samples = 0, slot_rows = 0;
  std::vector<Slot> slots;
  std::mutex mu;
  std::condition_variable cv;
  int next_acquire = 0;
  // direct mode: the kernel reads the samples from the pinned host buffer and writes the features to the pinned host
  // buffer itself -- upload and download then run concurrently, driven by the kernel's own loads and stores, where
  // the two DMA copies of the staged mode take turns on this platform (tools/pcie_overlap.py, tools/zero_copy_probe.py:
  // int16 samples 9.0 -> 6.4 ms per 1024 x 10 s batch, float32 14.7 -> 13.8 ms)
  bool direct = true;
  // the plan's launch refused the slots' sample format (a filter table too large for LDS beside the waves' areas):
  // samples are widened to float32 by a device pass first, as compute.py::launch does for such plans
  std::atomic<bool> widen{false};
};

namespace {

template <typename T>
__global__ __launch_bounds__(256) void widen_kernel(const T *__restrict__ in, float *__restrict__ out, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) out[i] = (float)in[i];
}

int32_t fail(const std::string &msg) {
  pds::set_error(msg);
  return PDS_ERR_INVALID;
}

void release_slot(Slot &s) {
  if (s.h_samples) (void)hipHostFree(s.h_samples);
  if (s.d_samples) (void)hipFree(s.d_samples);
  if (s.h_index) (void)hipHostFree(s.h_index);
  if (s.d_index) (void)hipFree(s.d_index);
  if (s.d_work) (void)hipFree(s.d_work);
  if (s.d_feats) (void)hipFree(s.d_feats);
  if (s.d_wide) (void)hipFree(s.d_wide);
  if (s.h_feats) (void)hipHostFree(s.h_feats);
  if (s.done) (void)hipEventDestroy(s.done);
  if (s.stream) (void)hipStreamDestroy(s.stream);
  delete[] s.h_rows;
  s = Slot();
}

int32_t build_slot(pds_feed *f, Slot &s) {
  const size_t sb = (size_t)std::max<int64_t>(f->slot_samples, 1) * f->sample_bytes;
  const size_t ib = (size_t)4 * f->slot_utts * sizeof(int64_t);
  const size_t fb = (size_t)std::max<int64_t>(f->slot_rows, 1) * f->feature_cols * sizeof(float);
  PDS_HIP(hipHostMalloc(&s.h_samples, sb, hipHostMallocDefault));
  PDS_HIP(hipMalloc(&s.d_samples, sb));
  PDS_HIP(hipHostMalloc((void **)&s.h_index, ib, hipHostMallocDefault));
  PDS_HIP(hipMalloc((void **)&s.d_index, ib));
  PDS_HIP(hipMalloc((void **)&s.d_work, ((size_t)std::min<int32_t>(f->slot_utts, 65535) + 1) * sizeof(int64_t)));
  PDS_HIP(hipMalloc((void **)&s.d_feats, fb));
  PDS_HIP(hipHostMalloc((void **)&s.h_feats, fb, hipHostMallocDefault));
  PDS_HIP(hipHostGetDevicePointer(&s.hd_samples, s.h_samples, 0));
  PDS_HIP(hipHostGetDevicePointer((void **)&s.hd_feats, s.h_feats, 0));
  PDS_HIP(hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking));
  PDS_HIP(hipEventCreateWithFlags(&s.done, hipEventDisableTiming));
  s.h_rows = new int64_t[(size_t)f->slot_utts + 1];
  return PDS_OK;
}

}  // namespace

extern "C" {

int32_t pds_feed_create(const pds_stft_plan *plan, int32_t sample_format, int64_t slot_samples, int32_t slot_utts,
                        int32_t slots, int32_t feature_cols, pds_feed **feed_out) {
  if (!plan || !feed_out) return fail("feed_create: null argument");
  if (sample_format != PDS_SAMPLES_F32 && sample_format != PDS_SAMPLES_F64 && sample_format != PDS_SAMPLES_I16)
    return fail("feed_create: sample_format must be PDS_SAMPLES_F32, PDS_SAMPLES_F64 or PDS_SAMPLES_I16");
  if (slot_samples < 1 || slot_utts < 1 || slots < 1 || slots > 16 || feature_cols < 0)
    return fail("feed_create: need slot_samples >= 1, slot_utts >= 1, 1 <= slots <= 16, feature_cols >= 0");
  int current = -1;
  PDS_HIP(hipGetDevice(&current));
  if (current != plan->device) return fail("feed_create: the plan lives on another device than the current one");
  pds_feed *f = new (std::nothrow) pds_feed();
  if (!f) return PDS_ERR_NOMEM;
  f->plan = plan;
  f->device = current;
  f->format = sample_format;
  f->sample_bytes = sample_format == PDS_SAMPLES_F64 ? 8 : sample_format == PDS_SAMPLES_I16 ? 2 : 4;
  f->slot_samples = slot_samples;
  f->slot_utts = slot_utts;
  f->num_slots = slots;
  f->out_cols = pds_stft_num_coeffs(plan);
  f->feature_cols = std::max(feature_cols, f->out_cols);
  // (float64 samples: the kernel's 16-byte pair loads do worse over the link than the DMA engine -- 29.9 against
  // 26.3 ms per 1024 x 10 s batch -- so they are staged by default)
  f->direct = sample_format != PDS_SAMPLES_F64;
  // (transform sizes without a fused float64- / int16-input kernel: widened on the device from the first batch on)
  f->widen.store((sample_format == PDS_SAMPLES_F64 && !pds_stft_plan_has_f64in(plan)) ||
                 (sample_format == PDS_SAMPLES_I16 && !pds_stft_plan_has_i16in(plan)));
  // rows a slot can hold: an utterance of n samples yields at most (n + S / 2) / S <= n / S + 1 frames
  f->slot_rows = slot_samples / plan->d.frame_shift + slot_utts;
  f->slots.resize(slots);
  for (Slot &s : f->slots) {
    const int32_t rc = build_slot(f, s);
    if (rc != PDS_OK) {
      for (Slot &t : f->slots) release_slot(t);
      delete f;
      return rc;
    }
  }
  *feed_out = f;
  return PDS_OK;
}

void pds_feed_destroy(pds_feed *f) {
  if (!f) return;
  for (Slot &s : f->slots) {
    if (s.stream) (void)hipStreamSynchronize(s.stream);
    release_slot(s);
  }
  delete f;
}

int64_t pds_feed_slot_rows(const pds_feed *f) { return f ? f->slot_rows : 0; }

int32_t pds_feed_set_direct(pds_feed *f, int32_t direct) {
  if (!f) return fail("feed_set_direct: null feed");
  std::lock_guard<std::mutex> lk(f->mu);
  for (const Slot &s : f->slots)
    if (s.state == ACQUIRED || s.state == SUBMITTED) return fail("feed_set_direct: a batch is in flight");
  f->direct = direct != 0;
  return PDS_OK;
}

int32_t pds_feed_acquire(pds_feed *f, int32_t *slot_out, void **h_samples_out) {
  if (!f || !slot_out || !h_samples_out) return fail("feed_acquire: null argument");
  std::unique_lock<std::mutex> lk(f->mu);
  const int k = f->next_acquire;
  Slot &s = f->slots[k];
  if (s.state == ACQUIRED) return fail("feed_acquire: the next slot of the ring was acquired and never submitted");
  // (SUBMITTED / COLLECTED: the caller's collecting side has it; wait for its release)
  f->cv.wait(lk, [&] { return s.state == FREE; });
  s.state = ACQUIRED;
  f->next_acquire = (k + 1) % f->num_slots;
  *slot_out = k;
  *h_samples_out = s.h_samples;
  return PDS_OK;
}

int32_t pds_feed_pack(pds_feed *f, int32_t slot, const void *const *signals, const int64_t *lengths, int32_t n_utts,
                      int32_t threads) {
  if (!f || slot < 0 || slot >= f->num_slots || (n_utts > 0 && (!signals || !lengths)))
    return fail("feed_pack: bad argument");
  Slot &s = f->slots[slot];
  if (s.state != ACQUIRED) return fail("feed_pack: the slot is not acquired");
  if (n_utts > f->slot_utts) return fail("feed_pack: more utterances than the slot holds");
  std::vector<int64_t> off((size_t)n_utts + 1, 0);
  for (int32_t b = 0; b < n_utts; ++b) {
    if (lengths[b] < 0) return fail("feed_pack: negative length");
    off[b + 1] = off[b] + lengths[b];
  }
  if (off[n_utts] > f->slot_samples) return fail("feed_pack: more samples than the slot holds");
  // the copies are dealt out by bytes, not by utterances: thread t takes bytes [t, t + 1) * total / threads of the
  // packed buffer, whatever utterances they belong to
  const int64_t total = off[n_utts] * f->sample_bytes;
  const int nt = (int)std::max<int64_t>(1, std::min<int64_t>(std::min(threads, 64), total >> 20));
  char *dst = static_cast<char *>(s.h_samples);
  const int sbytes = f->sample_bytes;
  auto work = [&](int t) {
    const int64_t lo = total * t / nt, hi = total * (t + 1) / nt;
    // first utterance whose end lies beyond lo
    int32_t b = (int32_t)(std::upper_bound(off.begin(), off.end(), lo / sbytes) - off.begin()) - 1;
    if (b < 0) b = 0;
    for (; b < n_utts && off[b] * sbytes < hi; ++b) {
      const int64_t u0 = off[b] * sbytes, u1 = off[b + 1] * sbytes;
      const int64_t c0 = std::max(u0, lo), c1 = std::min(u1, hi);
      if (c1 > c0) std::memcpy(dst + c0, static_cast<const char *>(signals[b]) + (c0 - u0), (size_t)(c1 - c0));
    }
  };
  if (nt == 1) {
    work(0);
  } else {
    std::vector<std::thread> pool;
    pool.reserve(nt - 1);
    for (int t = 1; t < nt; ++t) pool.emplace_back(work, t);
    work(0);
    for (std::thread &th : pool) th.join();
  }
  return PDS_OK;
}

int32_t pds_feed_submit(pds_feed *f, int32_t slot, const int64_t *lengths, int32_t n_utts, double preemph,
                        int32_t download) {
  return pds_feed_submit_frames(f, slot, lengths, nullptr, n_utts, -1, preemph, download);
}

int32_t pds_feed_submit_frames(pds_feed *f, int32_t slot, const int64_t *lengths, const int64_t *nframes, int32_t n_utts,
                               int32_t pad_left, double preemph, int32_t download) {
  if (!f || slot < 0 || slot >= f->num_slots || n_utts < 0 || (n_utts > 0 && !lengths) || pad_left < -1)
    return fail("feed_submit: bad argument");
  Slot &s = f->slots[slot];
  if (s.state != ACQUIRED) return fail("feed_submit: the slot is not acquired");
  if (n_utts > f->slot_utts) return fail("feed_submit: more utterances than the slot holds");
  int current = -1;
  PDS_HIP(hipGetDevice(&current));
  if (current != f->device) return fail("feed_submit: the feed lives on another device than the current one");
  const int64_t B = n_utts, SU = f->slot_utts;
  int64_t *off = s.h_index, *len = s.h_index + SU, *nfr = s.h_index + 2 * SU, *row = s.h_index + 3 * SU;
  int64_t samples = 0, rows = 0, longest = 0;
  for (int64_t b = 0; b < B; ++b) {
    if (lengths[b] < 0) return fail("feed_submit: negative length");
    off[b] = samples;
    len[b] = lengths[b];
    nfr[b] = nframes ? nframes[b] : pds_stft_num_frames(f->plan, lengths[b]);
    if (nfr[b] < 0) return fail("feed_submit: negative frame count");
    row[b] = rows;
    s.h_rows[b] = rows;
    samples += lengths[b];
    rows += nfr[b];
    longest = std::max(longest, nfr[b]);
  }
  s.h_rows[B] = rows;
  if (samples > f->slot_samples || rows > f->slot_rows) return fail("feed_submit: the batch does not fit the slot");
  s.n_utts = n_utts;
  s.rows = rows;
  s.samples = samples;
  const void *sig = f->direct ? s.hd_samples : s.d_samples;
  float *feat = (f->direct && download) ? s.hd_feats : s.d_feats;
  if (!f->direct && samples > 0)
    PDS_HIP(hipMemcpyAsync(s.d_samples, s.h_samples, (size_t)samples * f->sample_bytes, hipMemcpyHostToDevice, s.stream));
  if (B > 0) PDS_HIP(hipMemcpyAsync(s.d_index, s.h_index, (size_t)4 * SU * sizeof(int64_t), hipMemcpyHostToDevice, s.stream));
  // a batch whose utterances differ in length: stretch scheduling over the chunks that exist (float32 samples)
  const bool ragged = B > 0 && longest > 0 && (double)rows < 0.9 * (double)longest * (double)B;
  for (int attempt = 0; attempt < 2; ++attempt) {
    int32_t format = f->format;
    const void *src = sig;
    if (f->widen.load() && format != PDS_SAMPLES_F32 && rows > 0) {
      if (!s.d_wide) PDS_HIP(hipMalloc((void **)&s.d_wide, (size_t)f->slot_samples * sizeof(float)));
      const unsigned blocks = (unsigned)std::min<int64_t>((samples + 255) / 256, 256 * 32);
      if (format == PDS_SAMPLES_F64)
        hipLaunchKernelGGL(widen_kernel<double>, dim3(blocks), dim3(256), 0, s.stream, (const double *)sig, s.d_wide, samples);
      else
        hipLaunchKernelGGL(widen_kernel<int16_t>, dim3(blocks), dim3(256), 0, s.stream, (const int16_t *)sig, s.d_wide, samples);
      PDS_HIP(hipGetLastError());
      src = s.d_wide;
      format = PDS_SAMPLES_F32;
    }
    int32_t rc = PDS_OK;
    for (int64_t lo = 0; lo < B && rows > 0 && rc == PDS_OK; lo += 65535) {
      const int32_t nb = (int32_t)std::min<int64_t>(65535, B - lo);
      int64_t mx = 0;
      for (int64_t b = lo; b < lo + nb; ++b) mx = std::max(mx, nfr[b]);
      const int64_t *d_off = s.d_index + lo, *d_len = s.d_index + SU + lo, *d_nfr = s.d_index + 2 * SU + lo, *d_row = s.d_index + 3 * SU + lo;
      if (format == PDS_SAMPLES_F64)
        rc = pds_stft_batch_f64in(f->plan, (const double *)src, d_off, d_len, d_nfr, d_row, nb, mx, pad_left, preemph, feat, 0,
                                  f->out_cols, s.stream);
      else if (format == PDS_SAMPLES_I16 && ragged)
        rc = pds_stft_batch_ragged_i16in(f->plan, (const int16_t *)src, d_off, d_len, d_nfr, d_row, nb, mx, pad_left, preemph,
                                         s.d_work, feat, f->out_cols, s.stream);
      else if (format == PDS_SAMPLES_I16)
        rc = pds_stft_batch_i16in(f->plan, (const int16_t *)src, d_off, d_len, d_nfr, d_row, nb, mx, pad_left, preemph, feat,
                                  f->out_cols, s.stream);
      else if (ragged)
        rc = pds_stft_batch_ragged_f32(f->plan, (const float *)src, d_off, d_len, d_nfr, d_row, nb, mx, pad_left, preemph, s.d_work,
                                       feat, f->out_cols, s.stream);
      else
        rc = pds_stft_batch_f32(f->plan, (const float *)src, d_off, d_len, d_nfr, d_row, nb, mx, pad_left, preemph, feat,
                                f->out_cols, s.stream);
      // (the first piece is refused before anything of it is queued: switch the feed to widened samples and start over)
      if (rc == PDS_ERR_INVALID && lo == 0 && format != PDS_SAMPLES_F32 && attempt == 0) break;
      if (rc != PDS_OK) return rc;
    }
    if (rc == PDS_OK) break;
    f->widen.store(true);
  }
  if (download) {
    if (rows > 0 && !f->direct)
      PDS_HIP(hipMemcpyAsync(s.h_feats, s.d_feats, (size_t)rows * f->out_cols * sizeof(float), hipMemcpyDeviceToHost, s.stream));
    PDS_HIP(hipEventRecord(s.done, s.stream));
  }
  {
    std::lock_guard<std::mutex> lk(f->mu);
    s.state = download ? SUBMITTED : ACQUIRED;
  }
  return PDS_OK;
}

int32_t pds_feed_device_view(pds_feed *f, int32_t slot, void **d_features, int64_t *rows, const int64_t **row_offsets,
                             void **stream) {
  if (!f || slot < 0 || slot >= f->num_slots) return fail("feed_device_view: bad argument");
  Slot &s = f->slots[slot];
  if (d_features) *d_features = s.d_feats;
  if (rows) *rows = s.rows;
  if (row_offsets) *row_offsets = s.h_rows;
  if (stream) *stream = s.stream;
  return PDS_OK;
}

int32_t pds_feed_download(pds_feed *f, int32_t slot, const void *d_src, int64_t bytes) {
  if (!f || slot < 0 || slot >= f->num_slots || bytes < 0 || (bytes > 0 && !d_src)) return fail("feed_download: bad argument");
  Slot &s = f->slots[slot];
  if (s.state != ACQUIRED) return fail("feed_download: the slot is not between submit(download = 0) and collect");
  if ((size_t)bytes > (size_t)std::max<int64_t>(f->slot_rows, 1) * f->feature_cols * sizeof(float))
    return fail("feed_download: more bytes than the slot's host buffer holds (slot_rows x feature_cols float32)");
  if (bytes > 0) PDS_HIP(hipMemcpyAsync(s.h_feats, d_src, (size_t)bytes, hipMemcpyDeviceToHost, s.stream));
  PDS_HIP(hipEventRecord(s.done, s.stream));
  std::lock_guard<std::mutex> lk(f->mu);
  s.state = SUBMITTED;
  return PDS_OK;
}

int32_t pds_feed_collect(pds_feed *f, int32_t slot, const float **h_features, const int64_t **row_offsets, int64_t *rows) {
  if (!f || slot < 0 || slot >= f->num_slots) return fail("feed_collect: bad argument");
  Slot &s = f->slots[slot];
  {
    std::lock_guard<std::mutex> lk(f->mu);
    if (s.state != SUBMITTED) return fail("feed_collect: the slot has no submitted batch");
  }
  PDS_HIP(hipEventSynchronize(s.done));
  if (h_features) *h_features = s.h_feats;
  if (row_offsets) *row_offsets = s.h_rows;
  if (rows) *rows = s.rows;
  std::lock_guard<std::mutex> lk(f->mu);
  s.state = COLLECTED;
  return PDS_OK;
}

int32_t pds_feed_unpack(pds_feed *f, int32_t slot, void *dst, int64_t bytes, int32_t threads) {
  if (!f || slot < 0 || slot >= f->num_slots || bytes < 0 || (bytes > 0 && !dst)) return fail("feed_unpack: bad argument");
  Slot &s = f->slots[slot];
  {
    std::lock_guard<std::mutex> lk(f->mu);
    if (s.state != COLLECTED) return fail("feed_unpack: the slot's batch was not collected");
  }
  if ((size_t)bytes > (size_t)std::max<int64_t>(f->slot_rows, 1) * f->feature_cols * sizeof(float))
    return fail("feed_unpack: more bytes than the slot's host buffer holds");
  const int nt = (int)std::max<int64_t>(1, std::min<int64_t>(std::min(threads, 64), bytes >> 20));
  const char *src = reinterpret_cast<const char *>(s.h_feats);
  char *out = static_cast<char *>(dst);
  auto work = [&](int t) {
    const int64_t lo = bytes * t / nt, hi = bytes * (t + 1) / nt;
    if (hi > lo) std::memcpy(out + lo, src + lo, (size_t)(hi - lo));
  };
  if (nt == 1) {
    work(0);
  } else {
    std::vector<std::thread> pool;
    pool.reserve(nt - 1);
    for (int t = 1; t < nt; ++t) pool.emplace_back(work, t);
    work(0);
    for (std::thread &th : pool) th.join();
  }
  return PDS_OK;
}

int32_t pds_feed_release(pds_feed *f, int32_t slot) {
  if (!f || slot < 0 || slot >= f->num_slots) return fail("feed_release: bad argument");
  Slot &s = f->slots[slot];
  {
    std::lock_guard<std::mutex> lk(f->mu);
    if (s.state != COLLECTED) return fail("feed_release: the slot's batch was not collected");
    s.state = FREE;
  }
  f->cv.notify_all();
  return PDS_OK;
}

}  // extern "C"
