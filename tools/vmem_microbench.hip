// Micro-benchmark: what one global load instruction of a wave costs the CU's vector-memory path on
// gfx950 when the data is cache resident (16 waves per CU, 8 loads in flight per wave) -- the frame
// loads of the fused STFT kernel are 25 global_load_dword per item and wave.
// Build: hipcc -O3 --offload-arch=gfx950 tools/vmem_microbench.hip -o tools/vmem_microbench
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP8(x) x x x x x x x x

// KIND 0: dword, 256 contiguous bytes per instruction   1: dword, four 64-byte runs 640 B apart (STFT pattern)
//      2: dwordx2 contiguous (512 B)                     3: dwordx4 contiguous (1 KB)
//      4: dword, scalar base + 32-bit lane offset (saddr form), 256 contiguous bytes
template <int KIND>
__global__ void bench(const float *buf, float *out, int iters) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // every wave sweeps its own 8 KB window of a buffer that stays in L1/L2
  const char *base = reinterpret_cast<const char *>(buf) + (blockIdx.x % 4) * 65536 + wave * 8192;
  float acc = 0.0f;
  const char *p;
  if (KIND == 1) p = base + (lane >> 4) * 640 + (lane & 15) * 4;
  else if (KIND == 2) p = base + lane * 8;
  else if (KIND == 3) p = base + lane * 16;
  else p = base + lane * 4;
  const unsigned off = lane * 4;
  const unsigned long long ub = (unsigned long long)base;
  const unsigned long long sbase = ((unsigned long long)__builtin_amdgcn_readfirstlane((int)(ub >> 32)) << 32) |
                                   (unsigned)__builtin_amdgcn_readfirstlane((int)ub);
  for (int i = 0; i < iters; ++i) {
    if constexpr (KIND == 0 || KIND == 1) {
      float v0, v1, v2, v3, v4, v5, v6, v7;
      asm volatile("global_load_dword %0, %8, off\n global_load_dword %1, %8, off offset:64\n global_load_dword %2, %8, off offset:128\n"
                   "global_load_dword %3, %8, off offset:192\n global_load_dword %4, %8, off offset:256\n global_load_dword %5, %8, off offset:320\n"
                   "global_load_dword %6, %8, off offset:384\n global_load_dword %7, %8, off offset:448\n s_waitcnt vmcnt(0)\n"
                   : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3), "=&v"(v4), "=&v"(v5), "=&v"(v6), "=&v"(v7) : "v"(p) : "memory");
      acc += v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7;
    } else if constexpr (KIND == 2) {
      typedef float f2 __attribute__((ext_vector_type(2)));
      f2 v0, v1, v2, v3, v4, v5, v6, v7;
      asm volatile("global_load_dwordx2 %0, %8, off\n global_load_dwordx2 %1, %8, off offset:512\n global_load_dwordx2 %2, %8, off offset:1024\n"
                   "global_load_dwordx2 %3, %8, off offset:1536\n global_load_dwordx2 %4, %8, off offset:2048\n global_load_dwordx2 %5, %8, off offset:2560\n"
                   "global_load_dwordx2 %6, %8, off offset:3072\n global_load_dwordx2 %7, %8, off offset:3584\n s_waitcnt vmcnt(0)\n"
                   : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3), "=&v"(v4), "=&v"(v5), "=&v"(v6), "=&v"(v7) : "v"(p) : "memory");
      acc += v0.x + v1.y + v2.x + v3.y + v4.x + v5.y + v6.x + v7.y;
    } else if constexpr (KIND == 3) {
      typedef float f4 __attribute__((ext_vector_type(4)));
      f4 v0, v1, v2, v3, v4, v5, v6, v7;
      asm volatile("global_load_dwordx4 %0, %8, off\n global_load_dwordx4 %1, %8, off offset:1024\n global_load_dwordx4 %2, %8, off offset:2048\n"
                   "global_load_dwordx4 %3, %8, off offset:3072\n global_load_dwordx4 %4, %8, off\n global_load_dwordx4 %5, %8, off offset:1024\n"
                   "global_load_dwordx4 %6, %8, off offset:2048\n global_load_dwordx4 %7, %8, off offset:3072\n s_waitcnt vmcnt(0)\n"
                   : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3), "=&v"(v4), "=&v"(v5), "=&v"(v6), "=&v"(v7) : "v"(p) : "memory");
      acc += v0.x + v1.y + v2.z + v3.w + v4.x + v5.y + v6.z + v7.w;
    } else {
      float v0, v1, v2, v3, v4, v5, v6, v7;
      asm volatile("global_load_dword %0, %8, %9\n global_load_dword %1, %8, %9 offset:256\n global_load_dword %2, %8, %9 offset:512\n"
                   "global_load_dword %3, %8, %9 offset:768\n global_load_dword %4, %8, %9 offset:1024\n global_load_dword %5, %8, %9 offset:1280\n"
                   "global_load_dword %6, %8, %9 offset:1536\n global_load_dword %7, %8, %9 offset:1792\n s_waitcnt vmcnt(0)\n"
                   : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3), "=&v"(v4), "=&v"(v5), "=&v"(v6), "=&v"(v7) : "v"(off), "s"(sbase) : "memory");
      acc += v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7;
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

template <int KIND>
void run(const char *name, int bytes) {
  float *buf, *out;
  hipMalloc(&buf, 4 * 65536 + 16384);
  hipMemset(buf, 0, 4 * 65536 + 16384);
  hipMalloc(&out, 256 * 8 * 256 * 4);
  const int iters = 4000;
  for (int waves_per_simd : {1, 2, 4}) {
    const int blocks = 256 * waves_per_simd;  // 256-thread blocks: one wave per SIMD each
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(bench<KIND>, dim3(blocks), dim3(256), 0, 0, buf, out, 10);
    hipEventRecord(e0);
    hipLaunchKernelGGL(bench<KIND>, dim3(blocks), dim3(256), 0, 0, buf, out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double loads_per_cu = (double)iters * 8 * 4 * waves_per_simd;  // wave-instructions per CU
    const double cyc = ms * 1e-3 * 2.4e9 / loads_per_cu;
    printf("%-44s waves/SIMD %d: %.3f ms -> %.1f cycles per wave-instruction per CU (@2.4 GHz nominal), %.0f B/clk/CU\n", name,
           waves_per_simd, ms, cyc, bytes / cyc);
  }
  hipFree(buf);
  hipFree(out);
}

int main() {
  run<0>("global_load_dword, 256 contiguous bytes", 256);
  run<1>("global_load_dword, 4 x 64 B runs 640 B apart", 256);
  run<4>("global_load_dword saddr + voffset, 256 B", 256);
  run<2>("global_load_dwordx2, 512 contiguous bytes", 512);
  run<3>("global_load_dwordx4, 1 KB contiguous", 1024);
  return 0;
}
