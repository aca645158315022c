/* A plain-C client of the HOST FEED of the boundary (include/pds_amd.h, pds_feed_*): host buffers in, host buffers out --
 * the caller makes no HIP call and links no HIP library itself.  Reads the problem of c_abi_client.c with int16 samples,
 * sends it through a two-slot feed in TWO batches (so the ring is reused), writes the features back.
 *
 *   file in : int32 L, S, N, pad_left, F, nnz, use_power, use_log, include_energy, B, total_samples
 *             double window[L]; int32 row_ptr[F + 1]; int32 col[nnz]; double val[nnz];
 *             int64 offsets[B]; int64 lengths[B]; int16 signal[total_samples]
 *   file out: int64 total_frames, num_coeffs; float feats[total_frames * num_coeffs]
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "pds_amd.h"

#define CHECK_PDS(call)                                                        \
  do {                                                                         \
    if ((call) != PDS_OK) {                                                    \
      fprintf(stderr, "%s: %s\n", #call, pds_last_error());                    \
      return 3;                                                                \
    }                                                                          \
  } while (0)

static int read_all(FILE *fh, void *dst, size_t bytes) { return fread(dst, 1, bytes, fh) == bytes ? 0 : 1; }

int main(int argc, char **argv) {
  if (argc != 3) return 1;
  FILE *fh = fopen(argv[1], "rb");
  if (!fh) return 1;
  int32_t h[11];
  if (read_all(fh, h, sizeof h)) return 1;
  const int32_t L = h[0], S = h[1], N = h[2], pad = h[3], F = h[4], nnz = h[5], B = h[9], total = h[10];
  double *window = malloc(sizeof(double) * L), *val = malloc(sizeof(double) * (nnz ? nnz : 1));
  int32_t *row_ptr = malloc(sizeof(int32_t) * (F + 1)), *col = malloc(sizeof(int32_t) * (nnz ? nnz : 1));
  int64_t *offsets = malloc(sizeof(int64_t) * B), *lengths = malloc(sizeof(int64_t) * B);
  int16_t *signal = malloc(sizeof(int16_t) * (total ? total : 1));
  if (read_all(fh, window, sizeof(double) * L) || read_all(fh, row_ptr, sizeof(int32_t) * (F + 1)) ||
      read_all(fh, col, sizeof(int32_t) * nnz) || read_all(fh, val, sizeof(double) * nnz) ||
      read_all(fh, offsets, sizeof(int64_t) * B) || read_all(fh, lengths, sizeof(int64_t) * B) ||
      read_all(fh, signal, sizeof(int16_t) * total))
    return 1;
  fclose(fh);
  if (pds_device_count() < 1) {
    fprintf(stderr, "no HIP device\n");
    return 4;
  }
  pds_stft_desc desc = {L, S, N, pad, F, nnz, h[6], h[7], h[8], 0, 1e-5};
  pds_stft_plan *plan = NULL;
  CHECK_PDS(pds_stft_plan_create(&desc, window, row_ptr, col, val, &plan));
  const int32_t C = pds_stft_num_coeffs(plan);
  pds_feed *feed = NULL;
  CHECK_PDS(pds_feed_create(plan, PDS_SAMPLES_I16, total > 0 ? total : 1, B > 0 ? B : 1, 2, 0, &feed));
  printf("feed: slots of %d samples / %d utterances, %lld rows\n", total, B, (long long)pds_feed_slot_rows(feed));

  /* two batches: utterances [0, B / 2) and [B / 2, B); the second is submitted before the first is collected */
  const int32_t cut[3] = {0, B / 2, B};
  int32_t slot[2];
  for (int k = 0; k < 2; ++k) {
    void *staging = NULL;
    const int32_t n = cut[k + 1] - cut[k];
    const void **ptrs = malloc(sizeof(void *) * (n ? n : 1));
    for (int32_t b = 0; b < n; ++b) ptrs[b] = signal + offsets[cut[k] + b];
    CHECK_PDS(pds_feed_acquire(feed, &slot[k], &staging));
    CHECK_PDS(pds_feed_pack(feed, slot[k], ptrs, lengths + cut[k], n, 2));
    CHECK_PDS(pds_feed_submit(feed, slot[k], lengths + cut[k], n, 0.0, 1));
    free(ptrs);
  }
  int64_t total_rows = 0;
  for (int32_t b = 0; b < B; ++b) total_rows += pds_stft_num_frames(plan, lengths[b]);
  float *out = malloc(sizeof(float) * (size_t)(total_rows ? total_rows : 1) * C);
  int64_t at = 0;
  for (int k = 0; k < 2; ++k) {
    const float *feats = NULL;
    const int64_t *row_off = NULL;
    int64_t rows = 0;
    CHECK_PDS(pds_feed_collect(feed, slot[k], &feats, &row_off, &rows));
    if (row_off[cut[k + 1] - cut[k]] != rows) return 5;
    memcpy(out + at * C, feats, sizeof(float) * (size_t)rows * C);
    at += rows;
    CHECK_PDS(pds_feed_release(feed, slot[k]));
  }
  if (at != total_rows) return 6;
  pds_feed_destroy(feed);
  pds_stft_plan_destroy(plan);
  fh = fopen(argv[2], "wb");
  if (!fh) return 1;
  const int64_t head[2] = {total_rows, C};
  fwrite(head, sizeof head, 1, fh);
  fwrite(out, sizeof(float), (size_t)total_rows * C, fh);
  fclose(fh);
  printf("features: %lld rows of %d\n", (long long)total_rows, C);
  return 0;
}
