// Energy per instruction on gfx950: the headline kernel runs AT the socket's power cap (1394 W of 1400 W, 1936 MHz instead
// of 2400: tools/power_sample.py), so its rate is set by the joules a frame costs, not only by issue slots.  This
// program runs one instruction kind at a time on every SIMD (4 waves each, 16 independent chains per lane, operands with
// random mantissas unless the kind says "zero") for ~2 s, samples the socket power and clock from hwmon meanwhile, and
// prints time per wave-instruction, power, clock and energy per wave-instruction above the resident-but-idle baseline.
// Build: hipcc -O3 --offload-arch=gfx950 tools/energy_microbench.hip -o tools/energy_microbench -lpthread
#include <hip/hip_runtime.h>
#include <dirent.h>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#define CHECK(x)                                                                   \
  do {                                                                             \
    hipError_t e_ = (x);                                                           \
    if (e_ != hipSuccess) {                                                        \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));    \
      exit(1);                                                                     \
    }                                                                              \
  } while (0)

typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

enum Kind {
  K_IDLE = 0,   // waves resident, s_sleep
  K_SNOP,       // s_nop only
  K_FMA,        // v_fma_f32 (three VGPR operands)
  K_FMA_ZERO,   // ... on zeros
  K_FMAC,       // v_fmac_f32 (VOP2: two VGPR reads + the accumulator)
  K_FMAMK,      // v_fmamk_f32 (literal multiplier)
  K_ADD,        // v_add_f32
  K_MUL,        // v_mul_f32
  K_PK_FMA,     // v_pk_fma_f32
  K_PK_ADD,     // v_pk_add_f32
  K_PK_MUL,     // v_pk_mul_f32
  K_MOV,        // v_mov_b32
  K_DPP,        // v_mov_b32 dpp row_ror:1
  K_LDS_R128,   // ds_read_b128
  K_LDS_W64,    // ds_write_b64
  K_MFMA_F32,   // v_mfma_f32_16x16x4_f32
  K_MFMA_4X4,   // v_mfma_f32_4x4x1_16b_f32
  K_MFMA_F16,   // v_mfma_f32_16x16x32_f16
  K_CVT,        // v_cvt_pkrtz_f16_f32
  K_LOG,        // v_log_f32
  K_GLOAD_L2,   // global_load_dword, 256 B per wave-instruction, a 2 MB window per workgroup's XCD (L2 hits)
  K_GLOAD_HBM,  // ... streaming through 2 GB (HBM)
  K_GSTORE,     // global_store_dword, streaming through 1 GB
  K_COUNT
};
static const char *kind_name[K_COUNT] = {"idle(s_sleep)", "s_nop", "v_fma_f32", "v_fma_f32 zeros", "v_fmac_f32", "v_fmamk_f32",
                                         "v_add_f32", "v_mul_f32", "v_pk_fma_f32", "v_pk_add_f32", "v_pk_mul_f32", "v_mov_b32",
                                         "v_mov_b32 dpp", "ds_read_b128", "ds_write_b64", "mfma_f32_16x16x4_f32", "mfma_f32_4x4x1_f32",
                                         "mfma_f32_16x16x32_f16", "v_cvt_pkrtz_f16_f32", "v_log_f32", "global_load_dword (L2)",
                                         "global_load_dword (HBM)", "global_store_dword (HBM)"};
// wave-instructions of the measured kind per loop iteration
static const int kind_per_iter[K_COUNT] = {1, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 16, 16, 16, 16, 16, 64, 64, 16, 16, 16};

#define R4(x) x x x x
#define R16(x) R4(R4(x))

template <int KIND>
__global__ __launch_bounds__(256, 4) void bench(const float *in, float *out, int iters, float *big = nullptr) {
  const int t = threadIdx.x + blockIdx.x * blockDim.x;
  float a[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) a[i] = in[(t * 16 + i) & 0xfffff];
  // multipliers near 1 with random mantissas, small addends: the chains stay bounded and keep random mantissas
  float m0 = 0.75f + 0.25f * in[(t + 7) & 0xfffff], m1 = -(0.75f + 0.25f * in[(t + 11) & 0xfffff]);
  float c0 = in[(t + 13) & 0xfffff], c1 = -0.999f * c0;
  float i0 = 1.0f / m0, i1 = 1.0f / m1;
  if (KIND == K_FMA_ZERO) {
#pragma unroll
    for (int i = 0; i < 16; ++i) a[i] = 0.0f;
    m0 = m1 = c0 = c1 = i0 = i1 = 0.0f;
  }
  __shared__ float lds[256 * 20];
  for (int i = threadIdx.x; i < 256 * 20; i += 256) lds[i] = in[(t + i) & 0xfffff];
  __syncthreads();
  const int addr16 = threadIdx.x * 16, addr8 = threadIdx.x * 8;
  f4 acc0 = {a[0], a[1], a[2], a[3]}, acc1 = {a[4], a[5], a[6], a[7]}, acc2 = {a[8], a[9], a[10], a[11]}, acc3 = {a[12], a[13], a[14], a[15]};
  for (int it = 0; it < iters; ++it) {
    if constexpr (KIND == K_IDLE) {
      asm volatile("s_sleep 64");
    } else if constexpr (KIND == K_SNOP) {
      R16(asm volatile("s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0");)
    } else if constexpr (KIND == K_FMA || KIND == K_FMA_ZERO) {
      // x <- x * m + c with alternating multiplier / addend
#pragma unroll
      for (int r = 0; r < 4; ++r) {
#pragma unroll
        for (int i = 0; i < 16; ++i)
          asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"((r & 1) ? m1 : m0), "v"((i & 1) ? c1 : c0));
      }
    } else if constexpr (KIND == K_FMAC) {
      // x <- x + m * c, the product's sign alternating (bounded)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
#pragma unroll
        for (int i = 0; i < 16; ++i) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[i]) : "v"((r & 1) ? m1 : m0), "v"((i & 1) ? c1 : c0));
      }
    } else if constexpr (KIND == K_FMAMK) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          if (r & 1)
            asm volatile("v_fmamk_f32 %0, %0, 0xbf4f1bbd, %1" : "+v"(a[i]) : "v"((i & 1) ? c1 : c0));
          else
            asm volatile("v_fmamk_f32 %0, %0, 0x3f6c835e, %1" : "+v"(a[i]) : "v"((i & 1) ? c1 : c0));
        }
      }
    } else if constexpr (KIND == K_ADD) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
#pragma unroll
        for (int i = 0; i < 16; ++i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(((i + r) & 1) ? c1 : c0));
      }
    } else if constexpr (KIND == K_MUL) {
      // x <- x * m, then x * (1 / m): the magnitude stays, the mantissas keep moving
#pragma unroll
      for (int r = 0; r < 4; ++r) {
#pragma unroll
        for (int i = 0; i < 16; ++i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"((r & 1) ? i0 : m0));
      }
    } else if constexpr (KIND == K_PK_FMA || KIND == K_PK_ADD || KIND == K_PK_MUL) {
      f2 p[8], m = {m0, m1}, c = {c0, c1}, mm = KIND == K_PK_MUL ? f2{i0, i1} : f2{m1, m0}, cc = {c1, c0};
#pragma unroll
      for (int i = 0; i < 8; ++i) p[i] = f2{a[2 * i], a[2 * i + 1]};
#pragma unroll
      for (int r = 0; r < 8; ++r) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          if constexpr (KIND == K_PK_FMA)
            asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"((r & 1) ? mm : m), "v"((i & 1) ? cc : c));
          else if constexpr (KIND == K_PK_ADD)
            asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(((i + r) & 1) ? cc : c));
          else
            asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"((r & 1) ? mm : m));
        }
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        a[2 * i] = p[i].x;
        a[2 * i + 1] = p[i].y;
      }
    } else if constexpr (KIND == K_MOV) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
#pragma unroll
        for (int i = 0; i < 16; ++i) asm volatile("v_mov_b32 %0, %1" : "=v"(a[i]) : "v"(a[(i + 1) & 15]));
      }
    } else if constexpr (KIND == K_DPP) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
#pragma unroll
        for (int i = 0; i < 16; ++i)
          asm volatile("s_nop 1\n v_mov_b32_dpp %0, %1 row_ror:1 row_mask:0xf bank_mask:0xf" : "=v"(a[i]) : "v"(a[(i + 1) & 15]));
      }
    } else if constexpr (KIND == K_LDS_R128) {
      f4 q[4];
      R4(asm volatile("ds_read_b128 %0, %4\n ds_read_b128 %1, %4 offset:4096\n ds_read_b128 %2, %4 offset:8192\n ds_read_b128 %3, %4 offset:12288\n s_waitcnt lgkmcnt(0)"
                      : "=v"(q[0]), "=v"(q[1]), "=v"(q[2]), "=v"(q[3]) : "v"(addr16) : "memory");
         a[0] += q[0].x;)
    } else if constexpr (KIND == K_LDS_W64) {
      f2 w0 = {a[0], a[1]}, w1 = {a[2], a[3]}, w2 = {a[4], a[5]}, w3 = {a[6], a[7]};
      R4(asm volatile("ds_write_b64 %4, %0\n ds_write_b64 %4, %1 offset:2048\n ds_write_b64 %4, %2 offset:4096\n ds_write_b64 %4, %3 offset:6144\n s_waitcnt lgkmcnt(0)"
                      :: "v"(w0), "v"(w1), "v"(w2), "v"(w3), "v"(addr8) : "memory");)
    } else if constexpr (KIND == K_MFMA_F32) {
      R4(acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[4], a[5], acc0, 0, 0, 0);
         acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[6], a[7], acc1, 0, 0, 0);
         acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[8], a[9], acc2, 0, 0, 0);
         acc3 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[10], a[11], acc3, 0, 0, 0);)
      acc0 *= 0.01f; acc1 *= 0.01f; acc2 *= 0.01f; acc3 *= 0.01f;
    } else if constexpr (KIND == K_MFMA_4X4) {
      R4(acc0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a[4], a[5], acc0, 0, 0, 0);
         acc1 = __builtin_amdgcn_mfma_f32_4x4x1f32(a[6], a[7], acc1, 0, 0, 0);
         acc2 = __builtin_amdgcn_mfma_f32_4x4x1f32(a[8], a[9], acc2, 0, 0, 0);
         acc3 = __builtin_amdgcn_mfma_f32_4x4x1f32(a[10], a[11], acc3, 0, 0, 0);)
      acc0 *= 0.1f; acc1 *= 0.1f; acc2 *= 0.1f; acc3 *= 0.1f;
    } else if constexpr (KIND == K_MFMA_F16) {
      h8 x, y;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        x[i] = (_Float16)a[i];
        y[i] = (_Float16)a[8 + i];
      }
      R4(acc0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(x, y, acc0, 0, 0, 0);
         acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(y, x, acc1, 0, 0, 0);
         acc2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(x, x, acc2, 0, 0, 0);
         acc3 = __builtin_amdgcn_mfma_f32_16x16x32_f16(y, y, acc3, 0, 0, 0);)
      acc0 *= 0.01f; acc1 *= 0.01f; acc2 *= 0.01f; acc3 *= 0.01f;
    } else if constexpr (KIND == K_CVT) {
      int h[16];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
#pragma unroll
        for (int i = 0; i < 16; ++i) asm volatile("v_cvt_pkrtz_f16_f32 %0, %1, %2" : "=v"(h[i]) : "v"(a[i]), "v"(a[(i + r + 1) & 15]));
      }
#pragma unroll
      for (int i = 0; i < 16; ++i) a[i] += 1e-30f * (float)h[i];
    } else if constexpr (KIND == K_GLOAD_L2 || KIND == K_GLOAD_HBM) {
      // 16 loads of 256 contiguous bytes per wave in flight, then one wait (as the kernel's 25 per item)
      const size_t window = KIND == K_GLOAD_L2 ? (size_t)(1 << 19) : (size_t)(1 << 29);  // floats
      const size_t wave = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
      const size_t base = KIND == K_GLOAD_L2 ? ((wave * 4099 + (size_t)it * 1024) % (window - 1024 - 64))
                                              : ((wave * (size_t)iters + (size_t)it) * 1024 % (window - 1024 - 64));
      const float *src = big + base + (threadIdx.x & 63);
      float v[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) v[u] = __builtin_nontemporal_load(src + u * 64);
#pragma unroll
      for (int u = 0; u < 16; ++u) a[u] += v[u];
    } else if constexpr (KIND == K_GSTORE) {
      const size_t window = (size_t)(1 << 28);
      const size_t wave = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
      float *dst = big + ((wave * (size_t)iters + (size_t)it) * 1024 % (window - 1024 - 64)) + (threadIdx.x & 63);
#pragma unroll
      for (int u = 0; u < 16; ++u) dst[u * 64] = a[u];
    } else if constexpr (KIND == K_LOG) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
#pragma unroll
        for (int i = 0; i < 16; ++i) asm volatile("v_log_f32 %0, %0" : "+v"(a[i]));
#pragma unroll
        for (int i = 0; i < 16; ++i) a[i] = a[i] * a[i] + 1.5f;
      }
    }
  }
  float s = 0.0f;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += a[i];
  s += acc0.x + acc0.y + acc0.z + acc0.w + acc1.x + acc2.y + acc3.z + lds[(threadIdx.x * 7) & 4095];
  out[t] = s;
}

// ---- power sampling -----------------------------------------------------------------------------------------
struct Hwmon {
  std::string power, freq;
};
static long read_long(const std::string &path) {
  FILE *f = fopen(path.c_str(), "r");
  if (!f) return -1;
  long v = -1;
  if (fscanf(f, "%ld", &v) != 1) v = -1;
  fclose(f);
  return v;
}
static std::vector<Hwmon> find_hwmons() {
  std::vector<Hwmon> r;
  DIR *d = opendir("/sys/class/drm");
  if (!d) return r;
  while (dirent *e = readdir(d)) {
    if (strncmp(e->d_name, "card", 4) != 0 || strchr(e->d_name, '-')) continue;
    std::string base = std::string("/sys/class/drm/") + e->d_name + "/device/hwmon";
    DIR *h = opendir(base.c_str());
    if (!h) continue;
    while (dirent *he = readdir(h)) {
      if (strncmp(he->d_name, "hwmon", 5) != 0) continue;
      Hwmon m;
      m.power = base + "/" + he->d_name + "/power1_input";
      m.freq = base + "/" + he->d_name + "/freq1_input";
      if (read_long(m.power) >= 0) r.push_back(m);
    }
    closedir(h);
  }
  closedir(d);
  return r;
}

static float *g_big = nullptr;

template <int KIND>
static void run(const float *d_in, float *d_out, const std::vector<Hwmon> &mons, double base_w, double *out_idle_w) {
  const int blocks = 256 * 4;  // 16 waves per CU: 4 per SIMD
  int iters = KIND == K_IDLE ? 2000 : 4000;
  // calibrate to ~8 ms per launch
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  hipLaunchKernelGGL(bench<KIND>, dim3(blocks), dim3(256), 0, 0, d_in, d_out, 100, g_big);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  hipLaunchKernelGGL(bench<KIND>, dim3(blocks), dim3(256), 0, 0, d_in, d_out, iters, g_big);
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float ms;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  iters = (int)(iters * 8.0f / ms) + 1;
  std::atomic<bool> stop{false};
  std::vector<std::pair<long, long>> samples;  // (power uW, freq Hz) of the busiest card
  std::thread sampler([&] {
    while (!stop.load()) {
      long bp = -1, bf = -1;
      for (const Hwmon &m : mons) {
        const long p = read_long(m.power);
        if (p > bp) {
          bp = p;
          bf = read_long(m.freq);
        }
      }
      samples.push_back({bp, bf});
      std::this_thread::sleep_for(std::chrono::milliseconds(20));
    }
  });
  const auto t0 = std::chrono::steady_clock::now();
  int launches = 0;
  double kernel_ms = 0.0;
  while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < 2.5) {
    CHECK(hipEventRecord(e0));
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(bench<KIND>, dim3(blocks), dim3(256), 0, 0, d_in, d_out, iters, g_big);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    kernel_ms += ms;
    launches += 20;
  }
  stop.store(true);
  sampler.join();
  // the second half of the samples (the power reading is a moving average)
  double pw = 0.0, fq = 0.0;
  int n = 0;
  for (size_t i = samples.size() / 2; i < samples.size(); ++i) {
    pw += samples[i].first * 1e-6;
    fq += samples[i].second * 1e-6;
    ++n;
  }
  pw /= n > 0 ? n : 1;
  fq /= n > 0 ? n : 1;
  const double waves = (double)blocks * 4;
  const double instr = (double)launches * iters * kind_per_iter[KIND] * waves;  // wave-instructions
  const double secs = kernel_ms * 1e-3;
  const double per_simd_cycles = secs * fq * 1e6 / ((double)launches * iters * kind_per_iter[KIND] * 4);  // 4 waves per SIMD
  if (KIND == K_IDLE && out_idle_w) *out_idle_w = pw;
  printf("%-24s %7.1f W %6.0f MHz  %6.2f cycles/instr/SIMD  %8.1f pJ/lane-instr above resident-idle (%6.1f nJ per wave-instr)\n",
         kind_name[KIND], pw, fq, per_simd_cycles, (pw - base_w) * secs / instr / 64 * 1e12, (pw - base_w) * secs / instr * 1e9);
  fflush(stdout);
  CHECK(hipEventDestroy(e0));
  CHECK(hipEventDestroy(e1));
}

int main() {
  const std::vector<Hwmon> mons = find_hwmons();
  printf("hwmon power sensors found: %zu\n", mons.size());
  float *d_in, *d_out;
  std::vector<float> h(1 << 20);
  unsigned s = 12345u;
  for (float &v : h) {
    s = s * 1664525u + 1013904223u;
    v = ((s >> 8) * (1.0f / 16777216.0f)) * 2.0f - 1.0f;  // uniform (-1, 1), random mantissas
  }
  CHECK(hipMalloc(&d_in, h.size() * 4));
  CHECK(hipMalloc(&d_out, 256 * 4 * 256 * 4));
  CHECK(hipMemcpy(d_in, h.data(), h.size() * 4, hipMemcpyHostToDevice));
  CHECK(hipMalloc(&g_big, (size_t)(1 << 29) * sizeof(float)));
  CHECK(hipMemset(g_big, 0, (size_t)(1 << 29) * sizeof(float)));
  double idle_w = 0.0;
  run<K_IDLE>(d_in, d_out, mons, 0.0, &idle_w);
  printf("resident-idle baseline: %.1f W\n", idle_w);
  run<K_SNOP>(d_in, d_out, mons, idle_w, nullptr);
  run<K_FMA>(d_in, d_out, mons, idle_w, nullptr);
  run<K_FMA_ZERO>(d_in, d_out, mons, idle_w, nullptr);
  run<K_FMAC>(d_in, d_out, mons, idle_w, nullptr);
  run<K_FMAMK>(d_in, d_out, mons, idle_w, nullptr);
  run<K_ADD>(d_in, d_out, mons, idle_w, nullptr);
  run<K_MUL>(d_in, d_out, mons, idle_w, nullptr);
  run<K_PK_FMA>(d_in, d_out, mons, idle_w, nullptr);
  run<K_PK_ADD>(d_in, d_out, mons, idle_w, nullptr);
  run<K_PK_MUL>(d_in, d_out, mons, idle_w, nullptr);
  run<K_MOV>(d_in, d_out, mons, idle_w, nullptr);
  run<K_DPP>(d_in, d_out, mons, idle_w, nullptr);
  run<K_LDS_R128>(d_in, d_out, mons, idle_w, nullptr);
  run<K_LDS_W64>(d_in, d_out, mons, idle_w, nullptr);
  run<K_MFMA_F32>(d_in, d_out, mons, idle_w, nullptr);
  run<K_MFMA_4X4>(d_in, d_out, mons, idle_w, nullptr);
  run<K_MFMA_F16>(d_in, d_out, mons, idle_w, nullptr);
  run<K_CVT>(d_in, d_out, mons, idle_w, nullptr);
  run<K_LOG>(d_in, d_out, mons, idle_w, nullptr);
  run<K_GLOAD_L2>(d_in, d_out, mons, idle_w, nullptr);
  run<K_GLOAD_HBM>(d_in, d_out, mons, idle_w, nullptr);
  run<K_GSTORE>(d_in, d_out, mons, idle_w, nullptr);
  return 0;
}
