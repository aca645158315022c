// Micro-benchmark: issue cost of VALU/LDS instruction kinds on gfx950 at 1, 2, 4 waves per SIMD.
// Build: hipcc -O3 --offload-arch=gfx950 tools/valu_microbench.hip -o /tmp/valu_microbench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))

template <int KIND>
__global__ void bench(float *out, int iters) {
  float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  float b0 = 1.0001f, b1 = 0.9999f;
  __shared__ float lds[4096];
  lds[threadIdx.x] = a0;
  __syncthreads();
  int addr = (threadIdx.x & 63) * 4;
  for (int i = 0; i < iters; ++i) {
    if constexpr (KIND == 0) {  // v_add_f32, 8 independent chains
      REP8(asm volatile("v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n"
                        "v_add_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_add_f32 %6, %6, %8\n v_add_f32 %7, %7, %8\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0));)
    } else if constexpr (KIND == 1) {  // v_fma_f32
      REP8(asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                        "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0), "v"(b1));)
    } else if constexpr (KIND == 2) {  // v_pk_add_f32 on 4 register pairs
      typedef float float2_ __attribute__((ext_vector_type(2)));
      float2_ p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, q = {b0, b1};
      REP8(asm volatile("v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4\n"
                        "v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4\n"
                        : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(q));)
      a0 = p0.x; a1 = p0.y; a2 = p1.x; a3 = p1.y; a4 = p2.x; a5 = p2.y; a6 = p3.x; a7 = p3.y;
    } else if constexpr (KIND == 3) {  // v_pk_fma_f32
      typedef float float2_ __attribute__((ext_vector_type(2)));
      float2_ p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, q = {b0, b1};
      REP8(asm volatile("v_pk_fma_f32 %0, %0, %4, %4\n v_pk_fma_f32 %1, %1, %4, %4\n v_pk_fma_f32 %2, %2, %4, %4\n v_pk_fma_f32 %3, %3, %4, %4\n"
                        "v_pk_fma_f32 %0, %0, %4, %4\n v_pk_fma_f32 %1, %1, %4, %4\n v_pk_fma_f32 %2, %2, %4, %4\n v_pk_fma_f32 %3, %3, %4, %4\n"
                        : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(q));)
      a0 = p0.x; a1 = p0.y; a2 = p1.x; a3 = p1.y; a4 = p2.x; a5 = p2.y; a6 = p3.x; a7 = p3.y;
    } else if constexpr (KIND == 4) {  // v_mov_b32
      REP8(asm volatile("v_mov_b32 %0, %1\n v_mov_b32 %1, %2\n v_mov_b32 %2, %3\n v_mov_b32 %3, %4\n"
                        "v_mov_b32 %4, %5\n v_mov_b32 %5, %6\n v_mov_b32 %6, %7\n v_mov_b32 %7, %0\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
    } else if constexpr (KIND == 5) {  // dependent chain v_add (latency)
      REP64(asm volatile("v_add_f32 %0, %0, %1" : "+v"(a0) : "v"(b0));)
    } else if constexpr (KIND == 6) {  // ds_read_b32 x8 then wait
      REP8(asm volatile("ds_read_b32 %0, %8\n ds_read_b32 %1, %8 offset:256\n ds_read_b32 %2, %8 offset:512\n ds_read_b32 %3, %8 offset:768\n"
                        "ds_read_b32 %4, %8 offset:1024\n ds_read_b32 %5, %8 offset:1280\n ds_read_b32 %6, %8 offset:1536\n ds_read_b32 %7, %8 offset:1792\n s_waitcnt lgkmcnt(0)\n"
                        : "=v"(a0), "=v"(a1), "=v"(a2), "=v"(a3), "=v"(a4), "=v"(a5), "=v"(a6), "=v"(a7) : "v"(addr));)
    } else if constexpr (KIND == 7) {  // ds_write_b32 x8
      REP8(asm volatile("ds_write_b32 %8, %0\n ds_write_b32 %8, %1 offset:256\n ds_write_b32 %8, %2 offset:512\n ds_write_b32 %8, %3 offset:768\n"
                        "ds_write_b32 %8, %4 offset:1024\n ds_write_b32 %8, %5 offset:1280\n ds_write_b32 %8, %6 offset:1536\n ds_write_b32 %8, %7 offset:1792\n s_waitcnt lgkmcnt(0)\n"
                        :: "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(a4), "v"(a5), "v"(a6), "v"(a7), "v"(addr) : "memory");)
    } else if constexpr (KIND == 8) {  // v_mul_legacy + v_sqrt (transcendental rate)
      REP8(asm volatile("v_sqrt_f32 %0, %0\n v_sqrt_f32 %1, %1\n v_sqrt_f32 %2, %2\n v_sqrt_f32 %3, %3\n"
                        "v_sqrt_f32 %4, %4\n v_sqrt_f32 %5, %5\n v_sqrt_f32 %6, %6\n v_sqrt_f32 %7, %7\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
    } else if constexpr (KIND == 9) {  // ds_write_b64 x8
      typedef float float2_ __attribute__((ext_vector_type(2)));
      float2_ p0 = {a0, a1};
      int addr8 = (threadIdx.x & 63) * 8;
      REP8(asm volatile("ds_write_b64 %1, %0\n ds_write_b64 %1, %0 offset:512\n ds_write_b64 %1, %0 offset:1024\n ds_write_b64 %1, %0 offset:1536\n"
                        "ds_write_b64 %1, %0 offset:2048\n ds_write_b64 %1, %0 offset:2560\n ds_write_b64 %1, %0 offset:3072\n ds_write_b64 %1, %0 offset:3584\n s_waitcnt lgkmcnt(0)\n"
                        :: "v"(p0), "v"(addr8) : "memory");)
    } else if constexpr (KIND == 11) {  // v_add_f32 interleaved 1:1 with independent s_add_u32
      unsigned sa = i, sb = 3;
      REP8(asm volatile("v_add_f32 %0, %0, %9\n s_add_u32 %8, %8, %10\n v_add_f32 %1, %1, %9\n s_add_u32 %8, %8, %10\n"
                        "v_add_f32 %2, %2, %9\n s_add_u32 %8, %8, %10\n v_add_f32 %3, %3, %9\n s_add_u32 %8, %8, %10\n"
                        "v_add_f32 %4, %4, %9\n s_add_u32 %8, %8, %10\n v_add_f32 %5, %5, %9\n s_add_u32 %8, %8, %10\n"
                        "v_add_f32 %6, %6, %9\n s_add_u32 %8, %8, %10\n v_add_f32 %7, %7, %9\n s_add_u32 %8, %8, %10\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+s"(sa) : "v"(b0), "s"(sb) : "scc");)
      a0 += (float)sa;
    } else if constexpr (KIND == 12) {  // v_add_f32 interleaved 1:1 with s_nop 0
      REP8(asm volatile("v_add_f32 %0, %0, %8\n s_nop 0\n v_add_f32 %1, %1, %8\n s_nop 0\n v_add_f32 %2, %2, %8\n s_nop 0\n v_add_f32 %3, %3, %8\n s_nop 0\n"
                        "v_add_f32 %4, %4, %8\n s_nop 0\n v_add_f32 %5, %5, %8\n s_nop 0\n v_add_f32 %6, %6, %8\n s_nop 0\n v_add_f32 %7, %7, %8\n s_nop 0\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0));)
    } else if constexpr (KIND == 13) {  // v_add_f32 interleaved 1:1 with s_waitcnt (nothing outstanding)
      REP8(asm volatile("v_add_f32 %0, %0, %8\n s_waitcnt lgkmcnt(0)\n v_add_f32 %1, %1, %8\n s_waitcnt lgkmcnt(0)\n v_add_f32 %2, %2, %8\n s_waitcnt lgkmcnt(0)\n v_add_f32 %3, %3, %8\n s_waitcnt lgkmcnt(0)\n"
                        "v_add_f32 %4, %4, %8\n s_waitcnt lgkmcnt(0)\n v_add_f32 %5, %5, %8\n s_waitcnt lgkmcnt(0)\n v_add_f32 %6, %6, %8\n s_waitcnt lgkmcnt(0)\n v_add_f32 %7, %7, %8\n s_waitcnt lgkmcnt(0)\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0));)
    } else if constexpr (KIND == 10) {  // ds_read_b128 x8
      typedef float float4_ __attribute__((ext_vector_type(4)));
      float4_ r0, r1, r2, r3;
      int addr16 = (threadIdx.x & 63) * 16;
      REP8(asm volatile("ds_read_b128 %0, %4\n ds_read_b128 %1, %4 offset:1024\n ds_read_b128 %2, %4 offset:2048\n ds_read_b128 %3, %4 offset:3072\n"
                        "ds_read_b128 %0, %4 offset:4096\n ds_read_b128 %1, %4 offset:5120\n ds_read_b128 %2, %4 offset:6144\n ds_read_b128 %3, %4 offset:7168\n s_waitcnt lgkmcnt(0)\n"
                        : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3) : "v"(addr16));)
      a0 += r0.x + r1.y + r2.z + r3.w;
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

template <int KIND>
void run(const char *name, int per_iter) {
  float *out;
  hipMalloc(&out, 256 * 1024 * 4 * 8);
  const int iters = 2000;
  for (int waves_per_simd : {1, 2, 4, 8}) {
    int threads = 256;                  // 4 waves = one per SIMD
    int blocks = 256 * waves_per_simd;  // blocks per CU = waves per SIMD
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(bench<KIND>, dim3(blocks), dim3(threads), 0, 0, out, 10);
    hipEventRecord(e0);
    hipLaunchKernelGGL(bench<KIND>, dim3(blocks), dim3(threads), 0, 0, out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    // cycles per instruction per SIMD at 2.4 GHz (nominal): time * clk / (instrs per wave * waves per simd)
    double instrs_per_wave = (double)iters * per_iter;
    double cyc = ms * 1e-3 * 2.4e9 / (instrs_per_wave * waves_per_simd);
    printf("%-14s waves/SIMD %d: %.3f ms  -> %.2f cycles/instr/SIMD (@2.4GHz nominal)\n", name, waves_per_simd, ms, cyc);
  }
  hipFree(out);
}

int main() {
  run<0>("v_add_f32", 64);
  run<1>("v_fma_f32", 64);
  run<2>("v_pk_add_f32", 64);
  run<3>("v_pk_fma_f32", 64);
  run<4>("v_mov_b32", 64);
  run<5>("v_add dep", 64);
  run<8>("v_sqrt_f32", 64);
  run<6>("ds_read_b32", 64);
  run<7>("ds_write_b32", 64);
  run<9>("ds_write_b64", 64);
  run<10>("ds_read_b128", 64);
  // per VALU instruction (64 per iteration) with an equal number of scalar instructions mixed in
  run<11>("v_add+s_add", 64);
  run<12>("v_add+s_nop", 64);
  run<13>("v_add+s_waitcnt", 64);
  return 0;
}
