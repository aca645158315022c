// Fused STFT filter-bank kernel for power-of-two DFT sizes (float32), gfx950.
//
// One persistent 512-thread workgroup per CU walks batches of 64 consecutive frames of one
// utterance.  Per batch:
//
//  phase 1 (FFT, 16 lanes per frame, everything in registers except ONE LDS exchange)
//    N = N1 * N2.  Lane n2 of a frame's lane group loads the N1 samples x[N2*n1 + n2]
//    straight from global memory (symmetric reflection resolved in the index), windows them
//    and runs an in-lane REAL DFT of size N1 (fft_inlane.h).  Its outputs k1 = 1..N1/2-1 are
//    multiplied by the per-lane twiddles W_N^(n2*k1) and written to the wave's private LDS
//    exchange area, transposed: lane k1 then reads column k1 (N2 complex values), runs an
//    in-lane complex FFT of size N2 and holds bins k1 + N1*k2, k2 = 0..N2-1.  Bins beyond
//    N/2 are the mirror images of bins below it, and only |X|^2 is needed, so nothing is
//    wasted: the (N1/2-1)*N2 column bins plus the N2+1 bins that are multiples of N1/2 (a
//    real DFT of the per-lane even/odd sums, done by lane 0 of the group) are exactly the
//    N/2+1 half-spectrum bins.  |X|^2 goes to the batch's power buffer P[bin][frame] in LDS.
//  phase 2 (filter bank, lane = frame)
//    Each wave takes a share of the filters; a filter's weights and bin offsets are
//    wave-uniform, so they arrive through the scalar cache and the inner loop is one
//    ds_read + one v_fmac per tap.  log() and the energy column are applied here and the
//    64 x C result tile is staged in LDS.
//  phase 3: the tile is copied to global memory with coalesced stores.
//
// Reference semantics: compute_full framing (compute.py:574-607) and _compute_frame
// (compute.py:388-460), float32 arithmetic (the north star's 1e-4 tolerance).
#include <algorithm>
#include <cmath>
#include <vector>

#include "fft_inlane.h"
#include "pds_internal.h"

namespace pds {

struct FastParams {
  const float *sig;
  const int64_t *offsets, *lengths, *nframes, *row_off;
  float *out;
  int64_t out_stride;
  const float *win_lane;    // [N2][N1]  window[N2*n1 + n2], zero beyond L
  const float2 *tw_lane;    // [N2][N1/2] W_N^(n2*k1), pre-scaled (see rdft_scaled)
  const int32_t *f_rowptr;  // [F + 1]
  const int32_t *f_order;   // [F] filters, longest row first
  const int32_t *t_off;     // [nnz] bin * FS
  const float *t_w;         // [nnz]
  int L, S, pad_left, F, include_energy, use_power, use_log;
  float log_floor, inv_L;
  int tiles_per_utt, n_items;
};

template <int N1, int N2, int WAVES>
struct FastGeom {
  static constexpr int N = N1 * N2;
  static constexpr int H1 = N1 / 2;          // step-1 outputs k1 = 0..H1
  static constexpr int CPL = H1 / N2;        // step-3 columns per lane
  static constexpr int GROUPS = 64 / N2;     // frames per wave iteration
  static constexpr int GPH = 32 / N2;        // lane groups per 32-lane half
  static constexpr int FPB = 64;             // frames per batch (= lanes in phase 2)
  static constexpr int ITERS = FPB / (WAVES * GROUPS);
  static constexpr int NB = N / 2 + 1;       // half-spectrum bins
  static constexpr int NSLOT = NB + 1;       // + energy
  static constexpr int FS = FPB + 1;         // P row stride (floats): conflict-free both ways
  static constexpr int RS = N2 + 2;          // exchange row stride (float2)
  static constexpr int P_FLOATS = (NSLOT * FS + 3) / 4 * 4;
  static constexpr int EXCH_F2_PER_WAVE = GROUPS * H1 * RS;
  static constexpr int STAGE_FLOATS = WAVES * EXCH_F2_PER_WAVE * 2;
  static constexpr size_t SMEM_BYTES = (size_t)(P_FLOATS + STAGE_FLOATS) * 4;
  static_assert(H1 % N2 == 0 && CPL >= 1, "columns must split evenly over the lane group");
  static_assert(FPB % (WAVES * GROUPS) == 0 && ITERS >= 1, "batch must split evenly over waves");
  static_assert(N2 <= 32 && (RS * 8) % 16 == 0, "exchange rows must stay 16-byte aligned");
};

template <int N1, int N2, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void stft_fast_kernel(const FastParams p) {
  using G = FastGeom<N1, N2, WAVES>;
  constexpr int N = G::N, H1 = G::H1, FS = G::FS, RS = G::RS, NB = G::NB;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float *P = smem;
  float *stage = smem + G::P_FLOATS;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int g = lane / N2, r = lane % N2;
  float2 *exch = reinterpret_cast<float2 *>(stage) + wave * G::EXCH_F2_PER_WAVE + g * H1 * RS;

  // per-lane constants, loop invariant: window slice and inter-stage twiddles
  float win[N1];
  float twr[H1], twi[H1];
#pragma unroll
  for (int n1 = 0; n1 < N1; ++n1) win[n1] = p.win_lane[r * N1 + n1];
#pragma unroll
  for (int k1 = 1; k1 < H1; ++k1) {
    const float2 t = p.tw_lane[r * H1 + k1];
    twr[k1] = t.x;
    twi[k1] = t.y;
  }
  const int L = p.L, S = p.S;

  for (int item = blockIdx.x; item < p.n_items; item += gridDim.x) {
    const int b = item / p.tiles_per_utt;
    const int64_t t0 = (int64_t)(item - b * p.tiles_per_utt) * G::FPB;
    const int64_t nfr = p.nframes[b];
    if (t0 >= nfr) continue;  // uniform
    const int n = (int)p.lengths[b];
    const float *x = p.sig + p.offsets[b];

    // ------------------------------------------------------------ phase 1: FFT ------
#pragma unroll 1
    for (int it = 0; it < G::ITERS; ++it) {
      const int fr = (g % G::GPH) * N2 + (g / G::GPH) * 32 + wave * G::ITERS + it;
      const int64_t t = t0 + fr;
      const bool valid = t < nfr;
      const int start = (int)(t * S) - p.pad_left;
      // 0: every sample inside the signal, 1: one bounce suffices, 2: general reflection
      int mode = 0;
      if (valid) {
        if (start < 0 || start + L > n) mode = 1;
        if (start < -n || start + L > 2 * n) mode = 2;
      }
      const int wmode = __builtin_amdgcn_readfirstlane(
          __any(mode == 2) ? 2 : (__any(mode == 1) ? 1 : 0));

      float a[N1];
      float energy = 0.0f;
#pragma unroll
      for (int n1 = 0; n1 < N1; ++n1) {
        a[n1] = 0.0f;
        if (n1 * N2 < L) {  // uniform
          const int idx = n1 * N2 + r;
          if (valid && idx < L) {
            int i = start + idx;
            if (wmode == 1) {
              i = i < 0 ? -1 - i : (i >= n ? 2 * n - 1 - i : i);
            } else if (wmode == 2) {
              i = (int)reflect_index((int64_t)i, (int64_t)n);
            }
            a[n1] = x[i];
          }
        }
      }
      if (p.include_energy) {
#pragma unroll
        for (int n1 = 0; n1 < N1; ++n1) energy = fmaf(a[n1], a[n1], energy);
      }
#pragma unroll
      for (int n1 = 0; n1 < N1; ++n1) a[n1] *= win[n1];

      float even_sum, odd_sum, Ar[H1], Ai[H1];
      inl::rdft_scaled<N1>(a, even_sum, odd_sum, Ar, Ai);

      // transpose through LDS: row k1 of this frame's block holds column k1 for all n2
      {
        float *row0 = reinterpret_cast<float *>(exch);
        row0[r] = even_sum;        // c[n2]      = sum of even-indexed samples
        row0[N2 + r] = odd_sum;    // c[n2 + N2] = sum of odd-indexed samples
      }
#pragma unroll
      for (int k1 = 1; k1 < H1; ++k1) {
        float2 v;
        v.x = Ar[k1] * twr[k1] - Ai[k1] * twi[k1];
        v.y = Ar[k1] * twi[k1] + Ai[k1] * twr[k1];
        exch[k1 * RS + r] = v;
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

      float *Pf = P + fr;
#pragma unroll
      for (int q = 0; q < G::CPL; ++q) {
        const int kk = q * N2 + r;
        const float4 *row = reinterpret_cast<const float4 *>(exch + kk * RS);
        float zr[N2], zi[N2], Yr[N2], Yi[N2];
#pragma unroll
        for (int j = 0; j < N2 / 2; ++j) {
          const float4 v = row[j];
          zr[2 * j] = v.x;
          zi[2 * j] = v.y;
          zr[2 * j + 1] = v.z;
          zi[2 * j + 1] = v.w;
        }
        inl::CFFT<N2, 1>::run(zr, zi, Yr, Yi);
        if (q == 0 && r == 0) {
          // bins that are multiples of N1/2: real DFT of the 2*N2 even/odd sums
          inl::rdft_finish_power<2 * N2>(Yr, Yi, [&](auto mm, float re, float im) {
            constexpr int m = decltype(mm)::value;
            float pw = re * re + im * im;
            if (!p.use_power) pw = sqrtf(pw);
            Pf[(m * H1) * FS] = pw;
          });
        } else {
#pragma unroll
          for (int k2 = 0; k2 < N2; ++k2) {
            float pw = Yr[k2] * Yr[k2] + Yi[k2] * Yi[k2];
            if (!p.use_power) pw = sqrtf(pw);
            // bin kk + N1*k2, or its mirror image when beyond N/2
            const int bin = (k2 < N2 / 2) ? kk + N1 * k2 : N - kk - N1 * k2;
            Pf[bin * FS] = pw;
          }
        }
      }
      if (p.include_energy) {
#pragma unroll
        for (int off = N2 / 2; off >= 1; off >>= 1) energy += __shfl_xor(energy, off, 64);
        if (r == 0) Pf[NB * FS] = energy;
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    __syncthreads();

    // ------------------------------------------------------------ phase 2: filters ---
    const int C = p.F + (p.include_energy ? 1 : 0);
    const int CP = C | 1;  // odd row stride: conflict-free for lane = frame
    const int col0 = p.include_energy ? 1 : 0;
    const float *Pl = P + lane;
    for (int fi = wave; fi < p.F; fi += WAVES) {
      const int f = p.f_order[fi];
      const int q1 = p.f_rowptr[f + 1];
      float acc = 0.0f;
      for (int q = p.f_rowptr[f]; q < q1; ++q) acc = fmaf(p.t_w[q], Pl[p.t_off[q]], acc);
      if (p.use_log) acc = __logf(fmaxf(acc, p.log_floor));
      stage[lane * CP + col0 + f] = acc;
    }
    if (p.include_energy && wave == WAVES - 1) {
      float e = Pl[NB * FS] * p.inv_L;
      if (!p.use_power) e = sqrtf(e);
      if (p.use_log) e = __logf(fmaxf(e, p.log_floor));
      stage[lane * CP] = e;
    }
    __syncthreads();

    // ------------------------------------------------------------ phase 3: store -----
    {
      const int64_t left = nfr - t0;
      const int rows = left < G::FPB ? (int)left : G::FPB;
      const int total = rows * C;
      const float invC = 1.0f / (float)C;
      float *dst = p.out + (p.row_off[b] + t0) * p.out_stride;
      for (int e = threadIdx.x; e < total; e += WAVES * 64) {
        const int row = (int)(((float)e + 0.5f) * invC);
        const int c = e - row * C;
        dst[(int64_t)row * p.out_stride + c] = stage[row * CP + c];
      }
    }
    __syncthreads();
  }
}

// ----------------------------------------------------------------------- host side ---

template <int N1, int N2, int WAVES>
static int32_t launch_geom(const pds_stft_plan *plan, const BatchArgs &a) {
  using G = FastGeom<N1, N2, WAVES>;
  const FastTables &ft = plan->fast;
  FastParams p;
  p.sig = (const float *)a.d_signal;
  p.offsets = a.d_offsets;
  p.lengths = a.d_lengths;
  p.nframes = a.d_nframes;
  p.row_off = a.d_row_off;
  p.out = (float *)a.d_out;
  p.out_stride = a.out_stride;
  p.win_lane = ft.d_window;
  p.tw_lane = (const float2 *)ft.d_twiddle;
  p.f_rowptr = plan->d_row_ptr;
  p.f_order = ft.d_order;
  p.t_off = ft.d_toff;
  p.t_w = ft.d_wval;
  p.L = plan->d.frame_length;
  p.S = plan->d.frame_shift;
  p.pad_left = a.pad_left;
  p.F = plan->d.num_filts;
  p.include_energy = plan->d.include_energy;
  p.use_power = plan->d.use_power;
  p.use_log = plan->d.use_log;
  p.log_floor = (float)plan->d.log_floor;
  p.inv_L = 1.0f / (float)plan->d.frame_length;
  p.tiles_per_utt = (int)((a.max_frames + G::FPB - 1) / G::FPB);
  const int64_t items = (int64_t)p.tiles_per_utt * a.B;
  if (items > 0x7fffffff) {
    set_error("stft_batch: too many frame tiles in one call");
    return PDS_ERR_INVALID;
  }
  p.n_items = (int)items;
  auto kern = stft_fast_kernel<N1, N2, WAVES>;
  static bool attr_set = false;  // per instantiation
  if (!attr_set) {
    PDS_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)G::SMEM_BYTES));
    attr_set = true;
  }
  int grid = ft.num_cus;  // one persistent workgroup per CU
  if (grid > p.n_items) grid = p.n_items;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(WAVES * 64), G::SMEM_BYTES, a.stream, p);
  PDS_HIP(hipGetLastError());
  return PDS_OK;
}

int32_t launch_stft_fast_f32(const pds_stft_plan *plan, const BatchArgs &a) {
  // frame starts and reflected indices are formed in 32-bit arithmetic
  switch (plan->fast.kind) {
    case 512: return launch_geom<32, 16, 8>(plan, a);
    case 256: return launch_geom<32, 8, 8>(plan, a);
    default: break;
  }
  set_error("stft_batch: no fused kernel for this plan");
  return PDS_ERR_INVALID;
}

int32_t fast_tables_create(pds_stft_plan *plan, const double *window, const int32_t *row_ptr,
                           const int32_t *col, const double *val) {
  FastTables &ft = plan->fast;
  ft.kind = 0;
  const pds_stft_desc &d = plan->d;
  int n1 = 0, n2 = 0;
  if (d.dft_size == 512) { n1 = 32; n2 = 16; }
  else if (d.dft_size == 256) { n1 = 32; n2 = 8; }
  else return PDS_OK;  // generic kernel
  const int C = d.num_filts + (d.include_energy ? 1 : 0);
  const int stage_floats = 8 * (64 / n2) * (n1 / 2) * (n2 + 2) * 2;
  if (64 * (C | 1) > stage_floats) return PDS_OK;  // output tile would not fit the staging area
  if (d.frame_length > d.dft_size) return PDS_OK;
  const int N = d.dft_size, H1 = n1 / 2, FS = 65;
  std::vector<float> win((size_t)n1 * n2, 0.0f);
  for (int r = 0; r < n2; ++r)
    for (int k = 0; k < n1; ++k) {
      const int idx = n2 * k + r;
      if (idx < d.frame_length) win[(size_t)r * n1 + k] = (float)window[idx];
    }
  std::vector<float> tw((size_t)n2 * H1 * 2, 0.0f);
  for (int r = 0; r < n2; ++r)
    for (int k1 = 1; k1 < H1; ++k1) {
      const double ang = -2.0 * M_PI * (double)((r * k1) % N) / (double)N;
      const double scale = (2 * k1 == H1) ? 1.0 : 0.5;  // undo rdft_scaled's factor
      tw[((size_t)r * H1 + k1) * 2 + 0] = (float)(scale * std::cos(ang));
      tw[((size_t)r * H1 + k1) * 2 + 1] = (float)(scale * std::sin(ang));
    }
  std::vector<int32_t> order(d.num_filts), toff(d.nnz);
  for (int f = 0; f < d.num_filts; ++f) order[f] = f;
  std::stable_sort(order.begin(), order.end(), [&](int x, int y) {
    return row_ptr[x + 1] - row_ptr[x] > row_ptr[y + 1] - row_ptr[y];
  });
  std::vector<float> wv(d.nnz);
  for (int q = 0; q < d.nnz; ++q) {
    toff[q] = col[q] * FS;
    wv[q] = (float)val[q];
  }
  int32_t rc = PDS_OK;
  if (rc == PDS_OK) rc = upload(&ft.d_window, win.data(), win.size());
  if (rc == PDS_OK) rc = upload(&ft.d_twiddle, tw.data(), tw.size());
  if (rc == PDS_OK) rc = upload(&ft.d_order, order.data(), order.size());
  if (rc == PDS_OK) rc = upload(&ft.d_toff, toff.data(), toff.size());
  if (rc == PDS_OK) rc = upload(&ft.d_wval, wv.data(), wv.size());
  if (rc != PDS_OK) return rc;
  hipDeviceProp_t prop;
  PDS_HIP(hipGetDeviceProperties(&prop, plan->device));
  ft.num_cus = prop.multiProcessorCount;
  ft.n1 = n1;
  ft.n2 = n2;
  ft.kind = d.dft_size;
  return PDS_OK;
}

void fast_tables_destroy(pds_stft_plan *plan) {
  FastTables &ft = plan->fast;
  (void)hipFree(ft.d_window);
  (void)hipFree(ft.d_twiddle);
  (void)hipFree(ft.d_order);
  (void)hipFree(ft.d_toff);
  (void)hipFree(ft.d_wval);
  ft = FastTables();
}

}  // namespace pds
