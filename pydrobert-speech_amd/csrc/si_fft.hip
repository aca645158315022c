// Short-integration features, FFT form (float32): the filter bank of csrc/si.hip evaluated by
// overlap-save with 1024-point transforms, as the reference does it (compute.py:781-932), but with
// a block size of this kernel's own choosing: a transform yields `blocks` whole shift-sized blocks
// of filtered samples (blocks * S <= 1024 - (M - 1)), so every block's two window-weighted sums are
// complete inside one transform and no accumulator crosses a transform boundary.
//
// A 1024-point complex FFT lives in HALF a wavefront: 32 lanes x 32 registers, element n in lane
// n % 32, register n / 32.  One pass = in-lane FFT over the register index (fft_inlane.h), twiddle
// W_1024^(lane * q), transposition through the half-wave's private LDS area, in-lane FFT again;
// the result has the same layout (bin k in lane k % 32, register k / 32), so the inverse transform
// is the same routine between two conjugations, and the spectra of the filters are read in natural
// order, coalesced.  Per transform block: one forward FFT of the signal stretch (its spectrum stays in
// the lane's registers), then per filter a pointwise product, an inverse FFT, |y|^2 into LDS at the
// sample's own position, and per shift-sized block two sums over it weighted by the window's halves
// (the window repeats from block to block: ceil(S / lanes) factors per lane and half, in registers).
// The sums go to a scratch array [utterance][block][coefficient][half]; a second, trivial kernel adds
// first-half(t) + second-half(t + 1), applies the log and writes the features.  No workgroup barrier
// after the tables are staged, no atomics; results are bitwise reproducible.
//
// Two waves share a SIMD (256 registers each; 8-wave workgroups with 143-160 KB of LDS): the twiddles
// live in an LDS table, a filter's spectrum is fetched while the sums of the filter before are formed.
// Cost per filter and transform pair (one wave): ~1.2 k vector instructions for 2 x `blocks` frames,
// against 2 M S per frame for the direct form.  profiles/HISTORY.md 4.4 has the measurements.
#include <algorithm>
#include <cmath>
#include <vector>

#include "fft_inlane.h"
#include "pds_internal.h"

// Phase boundaries of a filter's pass, as scheduling barriers (at one wave per SIMD they were worth 7 %;
// at two waves -DPDS_SI_NO_PHASE measures +0.5 % on the 1024-point form and -2 % on the 2048-point one).
#ifdef PDS_SI_NO_PHASE
#define PDS_SI_PHASE()
#else
#define PDS_SI_PHASE() __builtin_amdgcn_sched_barrier(0)
#endif
// Wave priority per phase of a filter's pass (one hex digit each, from the lowest: 0 product + first in-lane
// transform, 1 twiddles + exchange, 2 second in-lane transform, 3 |y|^2 stores, 4 spectrum loads + block sums);
// negative: no hints.  The two waves of a SIMD run the same phases: see stft_switches.h on why staggering pays.
#ifndef PDS_SI_PRIO_1K
#define PDS_SI_PRIO_1K 0x32110  // 1024-point form: a pass's later phases first
#endif
#ifndef PDS_SI_PRIO_2K
#define PDS_SI_PRIO_2K 0x00012  // 2048-point form: a pass's earlier phases first
#endif
#define PDS_SI_SETPRIO(i) do { if (PRIO >= 0) __builtin_amdgcn_s_setprio((PRIO >> (4 * (i))) & 3); } while (0)
#ifndef PDS_SI_TW_BATCH
#define PDS_SI_TW_BATCH 16
#endif

namespace pds {

namespace {

constexpr int kN = 1024, kL = 32;       // transform size; lanes = registers = 32
constexpr int kRowStride = kL + 1;      // exchange row stride in float2: conflict-free both ways
constexpr int kWaves = 8;               // wavefronts per workgroup (two per SIMD), two transforms each
constexpr int kTwBatch = PDS_SI_TW_BATCH;   // twiddle rows read from LDS at a time
constexpr int kMaxBlocks = 8;           // shift-sized blocks a transform may yield (register budget)

// x + (x of the lane a DPP control word selects inside the 16-lane row)
template <int CTRL>
__device__ __forceinline__ float dpp_sum(float x) {
  return x + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xf, 0xf, false));
}

struct SiFftArgs {
  const float *sig;
  const int64_t *offsets, *lengths, *nframes;
  float *scratch;
  int64_t blocks_per_utt;   // scratch rows (shift-sized blocks) reserved per utterance
  const float2 *spectra, *twiddle;
  const float2 *twiddle2k;  // [32][32] W_2048^(32 q + l), row q (2048-point form only; staged as (1, 0) | W rows)
  const float *window;
  int64_t start;
  int S, C, blocks, use_power;
  int c_per_group;          // filters a workgroup walks (blockIdx.z picks the group): small batches are split over the filters too
};

// lower half-wave: x(lane) + x(lane + 32); upper half-wave: x(lane - 32) - x(lane)  (sgn = +1 / -1)
__device__ __forceinline__ float halves_sum(float x, float sgn) {
  const auto s = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  return fmaf(sgn, __uint_as_float(s[1]), __uint_as_float(s[0]));
}

__device__ __forceinline__ void half_wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// (zr, zi)[q] = element 32 q + l  ->  (zr, zi)[q] = bin 32 q + l of the 1024-point DFT
// tw[q * 32 + l] = W_1024^(l q), in LDS: two waves share a SIMD, so a wave has 256 registers and
// the table's reads are covered by the other wave
template <int PRIO>
__device__ __forceinline__ void fft1024(float (&zr)[kL], float (&zi)[kL], float2 *xch, const float2 *tw, int l) {
  float ar[kL], ai[kL];
  inl::CFFT<kL, 1>::run(zr, zi, ar, ai);
  PDS_SI_SETPRIO(1);
  // (twiddles in batches of kTwBatch rows: a table read behind an exchange write waits for it, one round
  // trip per row if they alternate)
#pragma unroll
  for (int q0 = 0; q0 < kL; q0 += kTwBatch) {
    float tr[kTwBatch], ti[kTwBatch];
#pragma unroll
    for (int i = 0; i < kTwBatch; ++i) {
      const float2 t = tw[(q0 + i) * kL + l];
      tr[i] = t.x;
      ti[i] = t.y;
    }
#pragma unroll
    for (int i = 0; i < kTwBatch; ++i) {
      const int q = q0 + i;
      float2 v;
      v.x = ar[q] * tr[i] - ai[q] * ti[i];
      v.y = ar[q] * ti[i] + ai[q] * tr[i];
      xch[q * kRowStride + l] = v;
    }
  }
  half_wave_sync();
#pragma unroll
  for (int q = 0; q < kL; ++q) {
    const float2 v = xch[l * kRowStride + q];
    ar[q] = v.x;
    ai[q] = v.y;
  }
  half_wave_sync();
  PDS_SI_SETPRIO(2);
  inl::CFFT<kL, 1>::run(ar, ai, zr, zi);
}

// BIG: one 2048-point transform per wavefront instead of two 1024-point ones -- radix 2 on top of
// the half-wave routine.  Time layout: sample n in half n & 1, lane (n >> 1) & 31, register n >> 6
// (each half holds a decimated sequence); frequency layout: bin k in half k >> 10, lane k & 31,
// register (k >> 5) & 31.  Forward: FFT-1024 per half, then E +- W^j O between the halves; the
// inverse runs the mirror image (sum / twiddled difference between the halves, then FFT-1024 per
// half), so both directions cost one exchange between the halves and one twiddle multiply more
// than the 1024-point form.  It serves filter supports up to 2048 - S taps and is also chosen for
// shorter ones when it wastes less of each transform on the overlap.
//
// NW: window factors a lane keeps per window half -- the lane sums samples lt, lt + LANES, ... of
// every block, and the window repeats from block to block, so ceil(S / LANES) <= NW factors per half
// weight all of them.
template <bool BIG, int NW>
__global__ __launch_bounds__(kWaves * 64, 1) void si_fft_kernel(const SiFftArgs p) {
  constexpr int NT = BIG ? 2 * kN : kN;
  constexpr int LANES = BIG ? 64 : kL;     // lanes of one transform
  constexpr int PRIO = BIG ? (PDS_SI_PRIO_2K) : (PDS_SI_PRIO_1K);
  extern __shared__ __attribute__((aligned(16))) unsigned char si_fft_smem[];
  float2 *xch_all = reinterpret_cast<float2 *>(si_fft_smem);          // [2 kWaves][32][33]
  float2 *tw = xch_all + 2 * kWaves * kL * kRowStride;                // [32][32]
  float2 *tw2 = tw + kL * kL;  // [32][64] (BIG): row q = 32 x (1, 0) for the lower half-wave, W_2048^(32 q + l) for the upper
  // (the padding slots of the exchange rows are never written and the block sums may read them,
  // weighted by zero: they must not hold a NaN another kernel left behind)
  for (int i = threadIdx.x; i < kWaves * kL * kRowStride; i += kWaves * 64)
    reinterpret_cast<float4 *>(xch_all)[i] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
  for (int i = threadIdx.x; i < kL * kL; i += kWaves * 64) tw[i] = p.twiddle[i];
  if constexpr (BIG)
    for (int i = threadIdx.x; i < 2 * kL * kL; i += kWaves * 64)
      tw2[i] = (i & 32) ? p.twiddle2k[(i >> 6) * kL + (i & 31)] : make_float2(1.0f, 0.0f);
  __syncthreads();
  const int half = threadIdx.x >> 5;       // 0 .. 2 kWaves - 1: one FFT-1024 each
  const int l = threadIdx.x & 31;
  const int hw = half & 1;                 // half of the wavefront
  const int lt = BIG ? (threadIdx.x & 63) : l;  // lane within the transform
  float2 *xch = xch_all + half * kL * kRowStride;
  const int b = blockIdx.y;
  const int V = p.blocks * p.S;
  const int64_t Tb = p.nframes[b];
  const int64_t num_dft = (Tb + 1 + p.blocks - 1) / p.blocks;  // frames need blocks 0 .. Tb
  const int64_t d = BIG ? (int64_t)blockIdx.x * kWaves + (half >> 1) : (int64_t)blockIdx.x * (2 * kWaves) + half;
  // (both halves of a wave take the same branch or wave-level code below would deadlock: the
  // half without work still walks through with a zero signal and stores nothing)
  const bool has_work = d < num_dft;
  if (__builtin_amdgcn_readfirstlane((int)__any(has_work)) == 0) return;
  const int64_t n = p.lengths[b];
  const float *x = p.sig + p.offsets[b];
  // element m of the stretch is signal sample d V + start - (NT - V) + m: the last V outputs of
  // the circular convolution are the filtered samples d V .. d V + V - 1
  const int64_t s0 = d * V + p.start - (NT - V);
  const int first_valid = NT - V;
  // sample held in register q (time layout) and bin held in register q (frequency layout)
  auto sample_of = [&](int q) { return BIG ? 64 * q + 2 * l + hw : kL * q + l; };
  auto bin_of = [&](int q) { return BIG ? kN * hw + kL * q + l : kL * q + l; };
  // window factors of the samples lt + LANES j of a block (0 past the block's end: those reads
  // land in the next block or behind the last one and are weighted away)
  float wa[NW], wb[NW];
#pragma unroll
  for (int j = 0; j < NW; ++j) {
    const int m = lt + LANES * j;
    wa[j] = m < p.S ? p.window[m] : 0.0f;
    wb[j] = m < p.S ? p.window[p.S + m] : 0.0f;
  }
  const float sgn = hw ? -1.0f : 1.0f;
  float zr[kL], zi[kL];
#pragma unroll
  for (int q = 0; q < kL; ++q) {
    const int64_t idx = s0 + sample_of(q);
    zr[q] = (has_work && idx >= 0 && idx < n) ? x[idx] : 0.0f;
    zi[q] = 0.0f;
  }
  fft1024<PRIO>(zr, zi, xch, tw, l);
  if constexpr (BIG) {
    // X[j] = E[j] + W^j O[j] (lower half), X[j + 1024] = E[j] - W^j O[j] (upper half): the upper half
    // multiplies (the lower one by 1), v_permlane32_swap hands every lane both halves' values
#pragma unroll
    for (int q = 0; q < kL; ++q) {
      const float2 t2 = tw2[q * 64 + lt];
      const float ur = zr[q] * t2.x - zi[q] * t2.y, ui = zr[q] * t2.y + zi[q] * t2.x;
      zr[q] = halves_sum(ur, sgn);
      zi[q] = halves_sum(ui, sgn);
    }
  }
  float xr[kL], xi[kL];  // the stretch's spectrum stays in the lane's registers for every filter
#pragma unroll
  for (int q = 0; q < kL; ++q) {
    xr[q] = zr[q];
    xi[q] = zi[q];
  }
  // |y|^2 of the transform's NT samples after the last transposition of a filter (2048-point form:
  // the wave's two areas, contiguous); the lane writes its samples and reads its share of the blocks
  float *zw = reinterpret_cast<float *>(BIG ? xch_all + (half & ~1) * kL * kRowStride : xch);
  float *zw_l = zw + (BIG ? 2 * l + hw : l);
  const float *zl = zw + first_valid + lt;
  float *srow = p.scratch + ((int64_t)b * p.blocks_per_utt + d * p.blocks) * p.C * 2;
  // the filter's spectrum is fetched while the block sums of the filter before are formed: by then
  // the transform's registers are free, and the loads land under the sums
  float hr[kL], hi[kL];
  // (rows 16 .. 31 from a second scalar base: their byte offsets do not fit the load's immediate field, and
  // per-row 64-bit addresses made the second half of the loads wait for the first)
  const unsigned lane_bin = BIG ? kN * hw + l : l;
  auto load_spectrum = [&](const float2 *h) {
    size_t off16 = 16 * kL;  // (opaque: otherwise the rows are re-expressed from h, address by address)
    if constexpr (!BIG || NW > 8) asm volatile("" : "+s"(off16));  // (2048-point form: sixteen 32-bit offsets fit its registers, and it measured 4 % faster that way)
    const float2 *h16 = h + off16;
#pragma unroll
    for (int q = 0; q < kL; ++q) {
      const float2 hs = q < 16 ? h[lane_bin + q * kL] : h16[lane_bin + (q - 16) * kL];
      hr[q] = hs.x;
      hi[q] = hs.y;
    }
  };
  const int c_lo = blockIdx.z * p.c_per_group, c_hi = min(p.C, c_lo + p.c_per_group);
  load_spectrum(p.spectra + (size_t)c_lo * NT);
  // a filter's sums are stored one pass later, in front of the next spectrum's loads: the wait for
  // those loads at the top of a pass then does not wait for stores issued behind them
  float tot[kMaxBlocks];
  const bool storer = has_work && lt < 32 && (lt & 15) == 0;
  float *sdst = srow + (lt >> 4);
  auto store_sums = [&](int c) {
    if (storer) {
#pragma unroll
      for (int k = 0; k < kMaxBlocks; ++k)
        if (k < p.blocks) sdst[((int64_t)k * p.C + c) * 2] = tot[k];
    }
  };
  for (int c = c_lo; c < c_hi; ++c) {
    PDS_SI_PHASE();
    PDS_SI_SETPRIO(0);
    // conj(X H): the inverse transform is conj(FFT(conj(.))) (1 / 1024 is folded into H)
#pragma unroll
    for (int q = 0; q < kL; ++q) {
      zr[q] = xr[q] * hr[q] - xi[q] * hi[q];
      zi[q] = -(xr[q] * hi[q] + xi[q] * hr[q]);
    }
    PDS_SI_PHASE();
    if constexpr (BIG) {
      // a[j] = Z[j] + Z[j + 1024] (lower half), b[j] = (Z[j] - Z[j + 1024]) W^j (upper half)
#pragma unroll
      for (int q0 = 0; q0 < kL; q0 += kTwBatch) {
        float2 t2[kTwBatch];
#pragma unroll
        for (int i = 0; i < kTwBatch; ++i) t2[i] = tw2[(q0 + i) * 64 + lt];
#pragma unroll
        for (int i = 0; i < kTwBatch; ++i) {
          const int q = q0 + i;
          const float sr = halves_sum(zr[q], sgn), si = halves_sum(zi[q], sgn);
          zr[q] = sr * t2[i].x - si * t2[i].y;
          zi[q] = sr * t2[i].y + si * t2[i].x;
        }
      }
    }
    fft1024<PRIO>(zr, zi, xch, tw, l);
    PDS_SI_PHASE();
    PDS_SI_SETPRIO(3);
    // |y|^2 (the conjugation does not matter) -> LDS, at the sample's own position in the transform
    // (one address register, the row in the instruction's offset); groups of four rows that lie
    // wholly inside the aliased part are skipped by a scalar branch
#pragma unroll
    for (int q4 = 0; q4 < kL; q4 += 4) {
      if (LANES * (q4 + 4) <= first_valid) continue;
      float z[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) z[i] = zr[q4 + i] * zr[q4 + i] + zi[q4 + i] * zi[q4 + i];
      if (!p.use_power) {
#pragma unroll
        for (int i = 0; i < 4; ++i) z[i] = __builtin_amdgcn_sqrtf(z[i]);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) zw_l[LANES * (q4 + i)] = z[i];
    }
    half_wave_sync();
    PDS_SI_PHASE();
    PDS_SI_SETPRIO(4);
    if (c > c_lo) store_sums(c - 1);
    load_spectrum(p.spectra + (size_t)(c + 1 < c_hi ? c + 1 : c) * NT);
    // every lane weights its share of every block with the two window halves (independent LDS
    // reads); four DPP steps sum inside the 16-lane rows, and one row swap between a block's two sums
    // leaves the first-half sum in the transform's lane 0 and the second-half sum in its lane 16
#pragma unroll
    for (int k = 0; k < kMaxBlocks; ++k)
      if (k < p.blocks) {
        const float *src = zl + k * p.S;
        float zs[NW];
#pragma unroll
        for (int j = 0; j < NW; ++j) zs[j] = src[j * LANES];
        float va = 0.0f, vb = 0.0f;
#pragma unroll
        for (int j = 0; j < NW; ++j) {
          va = fmaf(zs[j], wa[j], va);
          vb = fmaf(zs[j], wb[j], vb);
        }
        va = dpp_sum<0xB1>(va);   // quad_perm [1,0,3,2]
        vb = dpp_sum<0xB1>(vb);
        va = dpp_sum<0x4E>(va);   // quad_perm [2,3,0,1]
        vb = dpp_sum<0x4E>(vb);
        va = dpp_sum<0x141>(va);  // row_half_mirror
        vb = dpp_sum<0x141>(vb);
        va = dpp_sum<0x140>(va);  // row_mirror
        vb = dpp_sum<0x140>(vb);
        // rows (a0 b0 a2 b2) + (a1 b1 a3 b3)
        const auto sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(va), __float_as_uint(vb), false, false);
        float r = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
        if constexpr (BIG) {
          const auto s2 = __builtin_amdgcn_permlane32_swap(__float_as_uint(r), __float_as_uint(r), false, false);
          r = __uint_as_float(s2[0]) + __uint_as_float(s2[1]);
        }
        tot[k] = r;
      }
    half_wave_sync();
  }
  store_sums(c_hi - 1);
}

// frame t = first-half sum of block t + second-half sum of block t + 1 (compute.py:980-990)
__global__ __launch_bounds__(256) void si_combine_kernel(const float *scratch, int64_t blocks_per_utt,
                                                         const int64_t *nframes, const int64_t *row_off,
                                                         int C, int use_log, float log_floor, float *out,
                                                         int64_t out_stride) {
  const int b = blockIdx.y;
  const int64_t Tb = nframes[b];
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= Tb * C) return;
  const int64_t t = e / C;
  const int c = (int)(e - t * C);
  const float *base = scratch + ((int64_t)b * blocks_per_utt + t) * C * 2;
  float val = base[c * 2] + base[(C + c) * 2 + 1];
  if (use_log) val = logf(val < log_floor ? log_floor : val);  // NaN stays NaN
  out[(row_off[b] + t) * out_stride + c] = val;
}

}  // namespace

// blocks of S filtered samples one NT-point transform yields for supports of M taps (0: none)
constexpr int kMaxWindowRegs = 16;
static int blocks_for(int NT, int M, int S) {
  if (M > NT) return 0;
  const int blocks = std::min(kMaxBlocks, (NT - (M - 1)) / S);
  const int lanes = NT / kL;  // of one transform
  if (blocks < 1 || S > kMaxWindowRegs * lanes) return 0;
  // (the transposition area(s) of the transform -- 2.06 NT floats -- are reused for its NT squared
  // samples; the block sums read less than `lanes` floats past them, weighted by zero)
  return blocks;
}

int32_t si_fft_tables_create(pds_si_plan *plan, const double *taps) {
  SiFftTables &ft = plan->fft;
  ft.blocks = 0;
  const pds_si_desc &d = plan->d;
  const int M = d.max_support, S = d.frame_shift, C = d.num_coeffs;
  // 1024- or 2048-point transforms: whichever spends less of a transform on the overlap (the
  // larger one pays ~20 % more per point for its extra radix-2 stage)
  const int b1 = blocks_for(kN, M, S), b2 = blocks_for(2 * kN, M, S);
  const double eff1 = (double)b1 * S / kN, eff2 = (double)b2 * S / (2 * kN) / 1.2;
  if (b1 == 0 && b2 == 0) return PDS_OK;  // supports too long for this form: direct kernel
  ft.big = eff2 > eff1;
  const int NT = ft.big ? 2 * kN : kN, blocks = ft.big ? b2 : b1;
  std::vector<double> cs(NT), sn(NT);
  for (int j = 0; j < NT; ++j) {
    cs[j] = std::cos(2.0 * M_PI * j / NT);
    sn[j] = std::sin(2.0 * M_PI * j / NT);
  }
  std::vector<float2> spectra((size_t)C * NT);
  const int w = d.taps_complex ? 2 : 1;
  for (int c = 0; c < C; ++c)
    for (int k = 0; k < NT; ++k) {
      double re = 0.0, im = 0.0;
      for (int m = 0; m < M; ++m) {  // sum g[m] e^{-2 pi i k m / NT}
        const double gr = taps[((size_t)c * M + m) * w], gi = w == 2 ? taps[((size_t)c * M + m) * 2 + 1] : 0.0;
        const int j = (int)(((int64_t)k * m) % NT);
        re += gr * cs[j] + gi * sn[j];
        im += gi * cs[j] - gr * sn[j];
      }
      spectra[(size_t)c * NT + k] = make_float2((float)(re / NT), (float)(im / NT));
    }
  // W_1024^(q l) of the half-wave transform; W_2048^(32 q + l) of the radix-2 stage on top
  std::vector<float2> tw((size_t)kL * kL), tw2((size_t)kL * kL);
  for (int q = 0; q < kL; ++q)
    for (int l = 0; l < kL; ++l) {
      const double a1 = -2.0 * M_PI * (q * l) / kN, a2 = -2.0 * M_PI * (kL * q + l) / (2 * kN);
      tw[(size_t)q * kL + l] = make_float2((float)std::cos(a1), (float)std::sin(a1));
      tw2[(size_t)q * kL + l] = make_float2((float)std::cos(a2), (float)std::sin(a2));
    }
  int32_t rc = upload(&ft.d_spectra, spectra.data(), spectra.size());
  if (rc == PDS_OK) rc = upload(&ft.d_twiddle, tw.data(), tw.size());
  if (rc == PDS_OK) rc = upload(&ft.d_twiddle2k, tw2.data(), tw2.size());
  if (rc != PDS_OK) return rc;
  hipDeviceProp_t prop;
  PDS_HIP(hipGetDeviceProperties(&prop, plan->device));
  ft.num_cus = prop.multiProcessorCount;
  ft.blocks = blocks;
  return PDS_OK;
}

void si_fft_tables_destroy(pds_si_plan *plan) {
  (void)hipFree(plan->fft.d_spectra);
  (void)hipFree(plan->fft.d_twiddle);
  (void)hipFree(plan->fft.d_twiddle2k);
  plan->fft = SiFftTables();
}

static int64_t transforms_for(const pds_si_plan *plan, int64_t max_frames) {
  return (max_frames + 1 + plan->fft.blocks - 1) / plan->fft.blocks;
}

int64_t si_fft_scratch_len(const pds_si_plan *plan, int32_t B, int64_t max_frames) {
  if (!plan || plan->fft.blocks == 0 || B <= 0 || max_frames <= 0) return 0;
  return (int64_t)B * transforms_for(plan, max_frames) * plan->fft.blocks * plan->d.num_coeffs * 2;
}

int32_t launch_si_fft(const pds_si_plan *plan, const float *d_signal, const int64_t *d_offsets,
                      const int64_t *d_lengths, const int64_t *d_nframes, const int64_t *d_row_off,
                      int32_t B, int64_t max_frames, int64_t start, float *d_scratch, float *d_out,
                      int64_t out_stride, void *stream) {
  const pds_si_desc &d = plan->d;
  const int64_t transforms = transforms_for(plan, max_frames);
  SiFftArgs p;
  p.sig = d_signal;
  p.offsets = d_offsets;
  p.lengths = d_lengths;
  p.nframes = d_nframes;
  p.scratch = d_scratch;
  p.blocks_per_utt = transforms * plan->fft.blocks;
  p.spectra = plan->fft.d_spectra;
  p.twiddle = plan->fft.d_twiddle;
  p.twiddle2k = plan->fft.d_twiddle2k;
  p.window = plan->d_window_f32;
  p.start = start;
  p.S = d.frame_shift;
  p.C = d.num_coeffs;
  p.blocks = plan->fft.blocks;
  p.use_power = d.use_power;
  const size_t smem = ((size_t)2 * kWaves * kL * kRowStride + (plan->fft.big ? 3 : 1) * kL * kL) * sizeof(float2);
  const int lanes = plan->fft.big ? 64 : kL;
  // window factors per lane and half: the smallest of the built counts that covers a block
  const int nw = (p.S + lanes - 1) / lanes;
  void (*kern)(const SiFftArgs);
  if (plan->fft.big)
    kern = nw <= 3 ? si_fft_kernel<true, 3> : nw <= 5 ? si_fft_kernel<true, 5> : nw <= 8 ? si_fft_kernel<true, 8> : si_fft_kernel<true, kMaxWindowRegs>;
  else
    kern = nw <= 3 ? si_fft_kernel<false, 3> : nw <= 5 ? si_fft_kernel<false, 5> : nw <= 8 ? si_fft_kernel<false, 8> : si_fft_kernel<false, kMaxWindowRegs>;
  const int per_wg = plan->fft.big ? kWaves : 2 * kWaves;  // transforms per workgroup
  PDS_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
  // few workgroups (one utterance, a streaming call): the filters are dealt to several workgroups per stretch, each
  // repeating the stretch's forward transform -- 1 / c_per_group more work for a pass of filters in parallel
  const int64_t wgs = (transforms + per_wg - 1) / per_wg * B;
  const int want = (int)std::min<int64_t>(8, std::max<int64_t>(1, 2 * plan->fft.num_cus / std::max<int64_t>(1, wgs)));
  p.c_per_group = (p.C + want - 1) / want;
  const unsigned groups = (unsigned)((p.C + p.c_per_group - 1) / p.c_per_group);
  dim3 grid((unsigned)((transforms + per_wg - 1) / per_wg), (unsigned)B, groups);
  hipLaunchKernelGGL(kern, grid, dim3(kWaves * 64), smem, (hipStream_t)stream, p);
  PDS_HIP(hipGetLastError());
  const int64_t items = max_frames * d.num_coeffs;
  dim3 grid2((unsigned)((items + 255) / 256), (unsigned)B);
  hipLaunchKernelGGL(si_combine_kernel, grid2, dim3(256), 0, (hipStream_t)stream, d_scratch,
                     p.blocks_per_utt, d_nframes, d_row_off, d.num_coeffs, d.use_log, (float)d.log_floor,
                     d_out, out_stride);
  PDS_HIP(hipGetLastError());
  return PDS_OK;
}

}  // namespace pds
