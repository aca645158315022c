// tools/mfma4_microbench.hip: issue cost of v_mfma_f32_4x4x1_16B_f32 (and 16x16x4 for scale) by the number of
// independent accumulator chains and waves per SIMD.  hipcc --offload-arch=gfx950 -O3 -o tools/mfma4_microbench ...
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int CHAINS, bool BIG>
__global__ __launch_bounds__(64) void k(float *out, int iters, float a, float b) {
  f32x4 acc[CHAINS];
  for (int c = 0; c < CHAINS; ++c) acc[c] = f32x4{0, 0, 0, 0};
  float av = a + threadIdx.x, bv = b - threadIdx.x;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int r = 0; r < 8; ++r)
#pragma unroll
      for (int c = 0; c < CHAINS; ++c) {
        if constexpr (BIG) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc[c], 0, 0, 0);
        else acc[c] = __builtin_amdgcn_mfma_f32_4x4x1f32(av, bv, acc[c], 0, 0, 0);
      }
  }
  float s = 0;
  for (int c = 0; c < CHAINS; ++c) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
  out[blockIdx.x * 64 + threadIdx.x] = s;
}

template <int CHAINS, bool BIG>
void run(const char *name, int waves_per_simd) {
  float *out;
  const int cus = 256, blocks = cus * 4 * waves_per_simd, iters = 2000;
  hipMalloc(&out, blocks * 64 * sizeof(float));
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    k<CHAINS, BIG><<<blocks, 64>>>(out, iters, 1.0f, 2.0f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
  }
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double n = (double)iters * 8 * CHAINS * waves_per_simd;  // instructions per SIMD
  printf("%-28s chains %d waves/SIMD %d: %.3f ms -> %.1f cycles per instruction per SIMD (@2.4 GHz nominal)\n", name,
         CHAINS, waves_per_simd, ms, ms * 1e-3 * 2.4e9 / n);
  hipFree(out);
}

int main() {
  for (int w : {1, 2}) {
    run<1, false>("v_mfma_f32_4x4x1_16B_f32", w);
    run<2, false>("v_mfma_f32_4x4x1_16B_f32", w);
    run<4, false>("v_mfma_f32_4x4x1_16B_f32", w);
    run<1, true>("v_mfma_f32_16x16x4_f32", w);
    run<2, true>("v_mfma_f32_16x16x4_f32", w);
  }
  return 0;
}
