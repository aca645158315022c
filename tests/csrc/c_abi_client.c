/* A plain-C client of the boundary (include/pds_amd.h): what a cgo / JNI / ctypes binding does,
 * without Python or torch.  Reads a problem from a flat binary file written by the test, runs the
 * batched STFT through the C ABI on the GPU and writes the features back.
 *
 *   file in : int32 L, S, N, pad_left, F, nnz, use_power, use_log, include_energy, B, total_samples
 *             double window[L]; int32 row_ptr[F + 1]; int32 col[nnz]; double val[nnz];
 *             int64 offsets[B]; int64 lengths[B]; float signal[total_samples]
 *   file out: int64 total_frames, num_coeffs; float feats[total_frames * num_coeffs]
 */
#include <hip/hip_runtime_api.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "pds_amd.h"

#define CHECK_HIP(call)                                                        \
  do {                                                                         \
    hipError_t e_ = (call);                                                    \
    if (e_ != hipSuccess) {                                                    \
      fprintf(stderr, "%s: %s\n", #call, hipGetErrorString(e_));               \
      return 2;                                                                \
    }                                                                          \
  } while (0)
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
  int64_t *meta = malloc(sizeof(int64_t) * 4 * B); /* offsets | lengths | nframes | row offsets */
  float *signal = malloc(sizeof(float) * (total ? total : 1));
  if (read_all(fh, window, sizeof(double) * L) || read_all(fh, row_ptr, sizeof(int32_t) * (F + 1)) ||
      read_all(fh, col, sizeof(int32_t) * nnz) || read_all(fh, val, sizeof(double) * nnz) ||
      read_all(fh, meta, sizeof(int64_t) * 2 * B) || read_all(fh, signal, sizeof(float) * total))
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
  int64_t rows = 0, max_frames = 0;
  for (int32_t b = 0; b < B; ++b) {
    const int64_t nf = pds_stft_num_frames(plan, meta[B + b]);
    meta[2 * B + b] = nf;
    meta[3 * B + b] = rows;
    rows += nf;
    if (nf > max_frames) max_frames = nf;
  }
  float *d_sig, *d_out;
  int64_t *d_meta;
  CHECK_HIP(hipMalloc((void **)&d_sig, sizeof(float) * (total ? total : 1)));
  CHECK_HIP(hipMalloc((void **)&d_out, sizeof(float) * (rows * C ? rows * C : 1)));
  CHECK_HIP(hipMalloc((void **)&d_meta, sizeof(int64_t) * 4 * B));
  CHECK_HIP(hipMemcpy(d_sig, signal, sizeof(float) * total, hipMemcpyHostToDevice));
  CHECK_HIP(hipMemcpy(d_meta, meta, sizeof(int64_t) * 4 * B, hipMemcpyHostToDevice));
  hipStream_t stream;
  CHECK_HIP(hipStreamCreate(&stream));
  CHECK_PDS(pds_stft_batch_f32(plan, d_sig, d_meta, d_meta + B, d_meta + 2 * B, d_meta + 3 * B, B, max_frames,
                               -1, 0.0, d_out, C, stream));
  CHECK_HIP(hipStreamSynchronize(stream));
  float *feats = malloc(sizeof(float) * (rows * C ? rows * C : 1));
  CHECK_HIP(hipMemcpy(feats, d_out, sizeof(float) * rows * C, hipMemcpyDeviceToHost));
  const int32_t kind = pds_stft_plan_kernel_kind(plan);
  pds_stft_plan_destroy(plan);

  fh = fopen(argv[2], "wb");
  if (!fh) return 1;
  const int64_t dims[2] = {rows, C};
  fwrite(dims, sizeof dims, 1, fh);
  fwrite(feats, sizeof(float), (size_t)(rows * C), fh);
  fclose(fh);
  printf("kernel kind %d, %lld frames x %d coefficients\n", kind, (long long)rows, C);
  return 0;
}
