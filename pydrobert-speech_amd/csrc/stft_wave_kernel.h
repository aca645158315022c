// The fused STFT filter-bank kernel (template).  Its compile-time switches: stft_switches.h; its launcher:
// stft_wave_launch.h (included by stft_geom.hip, compiled once per geometry of stft_geoms.def, and by stft_fast.hip).
//
// Fused STFT filter-bank kernel for power-of-two DFT sizes 128..2048 (float32), gfx950.
//
// Every wavefront works alone -- no workgroup barrier in the steady state, no batch buffer.
// A wave takes 64/N2 consecutive frames of one utterance (N2 lanes per frame) and does:
//
//  1. load: lane n2 of a frame's lane group reads the samples x[N2*n1 + n2] straight from
//     global memory into registers (one 64-bit lane pointer + immediate offsets).  Frames
//     that touch an end of the signal take a rolled gather that resolves numpy's
//     "symmetric" reflection in the index (compute.py:599-600).
//  2. window (v_mul_legacy_f32) and an in-lane REAL DFT of size N1 = N / N2 over n1
//     (fft_inlane.h; the zero-padded tail rows are literal zeros, so the FFT is pruned).
//  3. outputs k1 = 1..N1/2-1 are multiplied by the per-lane twiddles W_N^(n2*k1) and written
//     to the wave's private LDS exchange area, transposed; lane k1 reads column k1 (N2
//     complex values), runs an in-lane complex FFT of size N2 and holds the bins
//     k1 + N1*k2.  Bins beyond N/2 mirror bins below it and only |X|^2 is needed, so the
//     (N1/2 - 1) * N2 column bins plus the N2 + 1 multiples of N1/2 (a real DFT of the lanes'
//     even/odd sample sums, done by lane 0 of the group) are exactly the half spectrum.
//  4. |X|^2 (or |X|) of the wave's frames goes to its LDS area again (P[frame][bin], aliased
//     over the exchange area) and the same lanes integrate the filters: lane (frame, j) owns
//     every N2-th filter of the length-sorted list (ELL layout: slot s of lane j is filter
//     order[s*N2 + j]; rows are dense bin ranges starting on a multiple of 4 bins, so both
//     the weights and the powers arrive as 16-byte LDS reads), applies log() and stores.
//
// 9..18 KB of LDS per wave and <= 128 VGPRs (N <= 512) give 16 resident waves per CU.
// Reference semantics: compute_full framing (compute.py:574-607) and _compute_frame
// (compute.py:388-460), float32 arithmetic (the north star's 1e-4 tolerance).
//
// Measured alternatives are recorded in DESIGN.md (batch kernel with workgroup barriers and a
// lane = frame filter phase: 0.76 G frames/s; this design: 2.2 G frames/s on one MI355X).
#pragma once
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <type_traits>
#include <vector>

#include "fft_inlane.h"
#include "mfma_front.h"
#include "rseg_tables.h"
#include "mseg_tables.h"
#include "pds_internal.h"

#include "stft_switches.h"

namespace pds {

__device__ __forceinline__ void keep_alive(float v) { asm volatile("" ::"v"(v)); }

struct FastParams {
  const void *sig;  // TIN samples
  const int64_t *offsets, *lengths, *nframes, *row_off;
  void *out;  // TOUT features
  int64_t out_stride;
  const float *win_lane;    // [N2][N1]   window[N2*n1 + n2], zero beyond L
  const float2 *tw_lane;    // [N2][N1/2] W_N^(n2*k1), pre-scaled (see rdft_scaled)
  const float2 *tw_special; // [N2] e^{-2 pi i r / (2 N2)}
  const float *win_half;    // PF: [N2][N1] the window times 1/2 (see twiddle_seeds)
  const float2 *tw_seed;    // PF: [N2][3] W_N^(n2), W_N^(4 n2), W_N^(8 n2), unscaled
  const float *ell_w;       // per slot: [N2][len + 4] dense weight rows (row j = lane j's filter)
  const int32_t *ell_meta;  // [slots][N2] first bin of the row | (filter + 1) << 16
  const int32_t *ell_len;   // [slots] row length in bins (multiple of 8)
  const int32_t *ell_woff;  // [slots] start of the slot's rows inside ell_w (floats)
  int ell_wfloats, ell_slots, ell_meta_pad, ell_meta_ints;  // meta_pad: ints in LDS, multiple of 4; meta_ints: valid ones
  // segmented filter walk (dense banks, 16-lane frames; see the kernel): when seg_rounds > 0,
  // ell_w / ell_meta hold its tables instead -- weights [slot][seg_len + 4], then in ell_meta the
  // slots' first bins [seg_rounds * 64] followed by (first slot | segments << 16) per filter
  int seg_rounds, seg_len, num_filts;
  int L, S, pad_left, include_energy, use_power, use_log;
  float log_floor, inv_L, preemph;
  double preemph_d;  // the coefficient at full precision (float64 samples)
  int chunks_per_utt, num_utts;
  int waves;  // wavefronts per workgroup of this launch (<= MAXWAVES)
  int area_f; // floats of a wave's private LDS area in this launch (a multiple of 4; WaveGeom::EXCH_F2 * 2 unless the walk needs less)
  unsigned waves_rcp;  // ceil(2^32 / waves): ticket / waves as a multiply-high (tickets < 2^28)
  int lds_ticket_off;  // floats from the start of the workgroup's LDS to its ticket counter
  int dyn;             // items handed out by the workgroup's ticket counter (0: static round-robin)
  const float *mf_tab;  // matrix-pipe front end (MF instantiations): image of MfmaFrontTables
  unsigned long long *stamps;  // diagnostic builds (PDS_STAMPS): [grid waves][12] phase times and absolute times, else null
  // fused statics + deltas launches (DLT instantiations): every wave walks ONE contiguous stretch of the
  // batch's chunks and keeps the statics of the last three chunks in registers (see the kernel)
  const int64_t *chunk_prefix;  // [num_utts + 1] chunks in front of every utterance (chunk_prefix_kernel)
  // CMVN sums fused with the producer (STR launches, pds_stft_cmvn_batch_f32; reference post.py:250-295, SURVEY.md
  // section 8 d): every wave adds the coefficients it stores to float64 sums in its own LDS slots, [2][stat_cs], and
  // leaves them -- sum x, then sum x^2 -- at stat_part[(global wave + utterance) * 2 * stat_c] when its piece of an
  // utterance ends; the normalising kernel adds an utterance's pieces in wave order (deterministic)
  double *stat_part;
  int stat_c, stat_cs;          // coefficients per row, and their count rounded up to a multiple of 4
  int lds_stat_off;             // floats from the start of the workgroup's LDS to the waves' sums
  int dl_inner;                 // coefficients per frame (= num_coeffs): order k goes to columns [k C, (k + 1) C)
  int dl_order;                 // 1 or 2: orders of deltas appended (the DLT instantiation serves both)
  int dl_debug;                 // measurement switches (PDS_DL_DEBUG): 1 no delta stores, 2 no deltas at all, 4 no statics stores
  int dl_eslot;                 // round * 64 + lane of a lane without a filter: carries the energy (-1: none)
  float dl_f1[5], dl_f2[9];     // taps of order 1 and order 2 (context window 2), correlation order
  int step_utts, step_chunks;  // (grid waves) / chunks_per_utt and (grid waves) % chunks_per_utt
};

// v_mul_legacy_f32: IEEE multiply except that 0 * x = 0 for every x (NaN and Inf included)
__device__ __forceinline__ float mul_legacy(float x, float y) {
  float z;
  asm("v_mul_legacy_f32 %0, %1, %2" : "=v"(z) : "v"(x), "v"(y));
  return z;
}

// The window multiply of row n1 of a frame.  Rows below N1 / 2 lie wholly inside the frame (the kernel requires
// L > N / 2); a row above may hold lanes past the frame's end -- they read samples of the next frame, possibly Inf / NaN,
// and their window value is exactly 0 -- and takes the 0 * x = 0 multiply.  The ordinary multiplies of the rows below
// contract with the transform's first additions, x[n] +- x[n + N1/2]: one multiply and two multiply-adds per pair
// instead of two and two.
template <int N1>
__device__ __forceinline__ float window_mul(int n1, float x, float w) {
#if PDS_WINDOW_CONTRACT
  return n1 < N1 / 2 ? x * w : mul_legacy(x, w);
#else
  return mul_legacy(x, w);
#endif
}

// DPP row_shr:M of `src` into `old`: lane i of a 16-lane row takes src of lane i - M; lanes whose
// source would lie outside the row keep `old`
template <int M>
__device__ __forceinline__ float dpp_row_shr_keep(float old, float src) {
  return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(src),
                                                    0x110 + M, 0xf, 0xf, false));
}

// natural log as v_log_f32 (log2, 1 ulp) times ln 2.  The argument is never below the plan's log
// floor, which plan creation requires to be a normal float for this kernel, so the denormal
// rescue sequence of the library logf is dead weight (12 instructions per value).
__device__ __forceinline__ float fast_log(float x) {
  return __builtin_amdgcn_logf(x) * 0.69314718055994530942f;
}

// Load through the constant address space: tables and utterance records are never written by
// this kernel, and a uniform address then always becomes a scalar load (the compiler otherwise
// falls back to 64-lane vector loads wherever it cannot prove that no store precedes the load).
typedef int Int4 __attribute__((ext_vector_type(4)));
template <typename T>
__device__ __forceinline__ T load_const(const T *ptr) {
  typedef const T __attribute__((address_space(4))) *const_ptr;
  return *(const_ptr)(uintptr_t)ptr;
}

// x + (x of the lane selected by a DPP control word within the row)
template <int CTRL>
__device__ __forceinline__ float dpp_add(float x) {
  return x + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xf, 0xf, false));
}

__device__ __forceinline__ void wave_sync() {
  // LDS operations of one wave execute in order; this only stops the compiler from moving
  // memory operations across the hand-off between lanes
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// N = N1 * N2: in-lane real DFT size x lanes per frame.  NROWS = ceil(L / N2) rows of N2
// samples per frame, a compile-time constant so that the zero-padded tail is literal zeros.
// N1 is a power of two (N = 128 .. 2048, radix-2 network) or, for transforms without zero
// padding (pad_to_nearest_power_of_two = False: N = L = 160, 200, 240, 320, 400, 480, 640, 800,
// 960), one of 20, 25, 30 evaluated directly; N2 is always a power of two.
// Two-pass exchange (the 64 x 16 geometry, N = 1024): a lane owns two columns there, and the exchange of all
// 32 columns at once takes 18.4 KB of LDS per wave -- beside the 38 KB filter table of the 64 gammatone filters at
// 48 kHz (BASELINE.json configs[4]) six waves per CU.  Columns 0..15 and 16..31 go through the SAME 16 rows one after
// the other instead (the second half waits, twiddled, in 32 registers: the geometry is built for two waves per SIMD
// and has them), so a wave's area is what the power spectra and the filter walk's partial sums need: 13.5 KB and
// eight waves per CU at first; 10.5 KB since the matrix-pipe walk keeps its sums in the accumulators across rounds
// (mseg_tables.h), which with regenerated twiddles and the window slice read from LDS (149 VGPRs instead of 243) lets
// that walk's launches run THREE waves per SIMD, eleven per CU.  -DPDS_TWOPASS=0: the one-pass exchange of rounds 1-2.
#ifndef PDS_TWOPASS
#define PDS_TWOPASS 1
#endif
// Lean form of the 64 x 32 geometry (N = 2048: 25 ms frames at 48 kHz, two frames per wave).  With its window
// slice (38 registers) and inter-stage twiddles (62) in registers it needs 209-235 of them and 17.4 KB of LDS per
// wave: two waves per SIMD, 0.21 of the HBM roofline.  Lean: the twiddles are regenerated per item from three seeds
// (inl::twiddle_chain<31>), the window slice is read from an LDS table where it is applied, and the exchange runs
// in two passes -- real parts, then imaginary parts, through ONE float per element (9.2 KB per wave) -- so that
// three waves per SIMD fit the registers and the LDS.  -DPDS_LEAN_2048=0: the form of rounds 1-2 (A/B builds).
#ifndef PDS_LEAN_2048
#define PDS_LEAN_2048 1
#endif
constexpr bool lean_geometry(int n1, int n2) { return PDS_LEAN_2048 && n1 == 64 && n2 == 32; }
// floats of a wave's private LDS area (exchange, then power spectra + the walks' partial sums)
constexpr int wave_area_floats(int n1, int n2) {
  const int cols = (n1 - 1) / 2 + 1, groups = 64 / n2, one_pass = groups * cols * (n2 + 2) * 2;
  // (N = 1024: the power rows, 2112 floats, + 36 partial-sum slots of the matrix-pipe walk, 16 floats each)
  return (PDS_TWOPASS && n1 == 64 && n2 == 16) ? 2688 : lean_geometry(n1, n2) ? groups * cols * (n2 + 4) : one_pass;
}
// floats of the window table a lean geometry keeps in LDS: [N2][stride], stride = 4 (mod 8) floats for
// conflict-free 16-byte reads
constexpr int win_table_stride(int rows) { return ((rows + 3) & ~3) % 8 == 4 ? ((rows + 3) & ~3) : ((rows + 3) & ~3) + 4; }

template <int N1, int N2, int NROWS>
struct WaveGeom {
  static constexpr int N = N1 * N2;
  static constexpr int NREG = (N1 - 1) / 2;  // step-2 outputs that become regular columns: k1 = 1..NREG
  static constexpr int COLS = NREG + 1;     // + column 0: the real-valued outputs (k1 = 0, and N1/2 if even)
  static constexpr int CPL = (COLS + N2 - 1) / N2;  // step-3 columns per lane
  static constexpr bool FULL = COLS % N2 == 0;      // every lane owns CPL columns (powers of two)
  // the packed real column (multiples of N1/2) is untangled by lanes 0..N2/2, one bin pair each
  static constexpr bool DIST = N2 >= 16;
  static constexpr int GROUPS = 64 / N2;    // frames per wave iteration
  static constexpr int NB = N / 2 + 1;      // half-spectrum bins
  static constexpr int RS = N2 + 2;         // exchange row stride (float2): conflict-free
  static constexpr bool TWOPASS = PDS_TWOPASS && N1 == 64 && N2 == 16;
  static constexpr bool LEAN = lean_geometry(N1, N2);    // regenerated twiddles, LDS window, real / imaginary exchange
  static constexpr int RSF = N2 + 4;                     // LEAN: exchange row stride (floats)
  static constexpr int XROWS = TWOPASS ? N2 : COLS;      // rows of a frame's exchange block
  static constexpr int EXCH_F2 = wave_area_floats(N1, N2) / 2;  // float2 per wave (the wave's whole area)
  static_assert(LEAN ? GROUPS * XROWS * RSF <= EXCH_F2 * 2 : GROUPS * XROWS * RS <= EXCH_F2, "the exchange must fit the wave's area");
  static_assert(!LEAN || (CPL == 1 && N2 >= 32), "lean form: one column per lane");
  // P row stride: >= NB + 1 (energy) and = 16 (mod 32) so that the two frames sharing a
  // 32-lane half write to disjoint banks
  static constexpr int PSTR = ((NB + 1 + 15) / 32) * 32 + 16;
  static_assert(GROUPS * PSTR <= EXCH_F2 * 2, "P must fit over the exchange area");
  // (a launch's area may be smaller than EXCH_F2 -- FastParams::area_f -- but never smaller than the exchange)
  static constexpr int XMIN_F = LEAN ? GROUPS * XROWS * RSF : GROUPS * XROWS * RS * 2;  // floats of the exchange itself
  static constexpr int GCH = XMIN_F / 64 < NROWS ? XMIN_F / 64 : NROWS;  // rows per pass of the edge-frame gather
  static_assert(N % 2 == 0 && CPL >= 1 && N2 <= 64 && (RS * 8) % 16 == 0, "geometry");
  // (N = 4096 = 64 x 64 has 32 columns for 64 lanes: the upper half of the wave idles in step 3)
  static_assert(inl::is_pow2(N1) ? (FULL || N2 == 2 * COLS) : 600 % N1 == 0,
                "in-lane DFT sizes: 2^k, or a divisor of 600");
  static_assert(NROWS <= N1 && NROWS > 0, "rows");
};

// ELL_LDS: the filter weight rows are staged in LDS (else read from global memory through L1/L2)
// PRE: pre-emphasis x[i] - c x[i-1] (reference pre.py:146) applied while loading the frame
// MAXWAVES / MINW: launch bounds (workgroup size limit, waves per SIMD the register budget must
// allow); the actual workgroup size is chosen per launch from the LDS the filter table needs
// filter slots handled by the unrolled slot loop: their row lengths and table offsets arrive with
// two 16-byte scalar loads per item; banks with more than USLOTS * N2 filters run the remaining
// slots from memory one by one
constexpr int USLOTS = 4;

// SEG: the filter phase is the segmented walk for dense banks (its own instantiation: inside the
// ELL kernel its registers cost the headline instantiation three spills)
//
// MF > 0: steps 1-3a (loads, window, N1-point real DFT) in the matrix-pipe form of mfma_front.h with
// MF pair steps: a lane holds k-slot q = lane / 16 of every frame's 16 x 16 x 4 tiles instead of
// one frame's samples, the N1-point DFT costs 2 (MF + 1) v_mfma_f32_16x16x4_f32 per frame on the
// matrix pipe (which runs beside the vector pipe) instead of ~250 vector instructions per item, and
// the lane twiddles and stores four output rows of each frame.  From the exchange on the kernel is
// the same.  (16-lane geometries with N1 = 32.)
typedef float f32x4 __attribute__((ext_vector_type(4)));

//
// RSG: the filter phase is the row-segment walk of rseg_tables.h (power spectra bin-major in LDS,
// a lane = one segment of one filter for all four frames; 16-lane geometries, tables in LDS)
//
// TIN / TOUT: sample and feature types in memory.  The arithmetic is float32 whatever they are:
// float64 samples are rounded as the frame is loaded (pre-emphasis, if fused, before the rounding, in
// float64 like the reference's own pass), float64 features are widened at the store -- the dtype flow
// of the reference's drivers (float64 audio in, compute.py:601 output dtype = input dtype) without
// separate conversion passes over the signal and the features.
template <int N1, int N2, int NROWS, int MAXWAVES, int MINW, bool ELL_LDS, bool PRE, int SEG = 0, int MF = 0,
          bool RSG = false, typename TIN = float, typename TOUT = float, int DLT = 0, bool STR = false, bool PF = false>
__global__ __launch_bounds__(MAXWAVES * 64, MINW) void stft_wave_kernel(const FastParams p) {
  // PF: the NEXT item's samples are loaded into registers while this item's column transforms and filter
  // walk run (a wave's own frame loads -- issue and wait -- were 27 % of its item, profiles/r2b_phase_stamps.txt).
  // The registers come from the inter-stage twiddles: instead of thirty loop-invariant registers the lane keeps
  // three seeds W^r, W^4r, W^8r and regenerates W^(r k1), k1 = 1..15, per item by twelve complex products at
  // most four deep (twiddle_chain; tests/test_twiddle_chain.py replays it against float64).  The next item's
  // utterance record is fetched at the top of the item for that.
  static_assert(!PF || (DLT == 0 && !STR && MF == 0 && !PRE && std::is_same<TIN, float>::value && (N1 == 32 || N1 == 64) && N2 == 16),
                "prefetch: 16-lane power-of-two geometries, float32 samples, round-robin scheduling");
  // STR: the stretch scheduling of the DLT launches without their deltas -- every wave walks one contiguous
  // stretch of the batch's EXISTING chunks (chunk_prefix) -- for ragged batches: dealt round-robin over
  // (utterance, chunk < chunks of the longest) the waves skip the chunks short utterances do not have and end
  // up with unequal shares (lengths uniform in 1 ... 15 s: 9 % slower per frame than equal lengths).
  constexpr bool STRETCH = DLT > 0 || STR;
  static_assert(!(STR && DLT > 0), "stretch scheduling is part of the fused-deltas launches already");
  // DLT = K > 0: Deltas(K, context_window 2, edge padding) of the features appended to every row by the
  // same launch (reference post.py:462-491; BASELINE.json configs[2]).  Every wave walks ONE contiguous
  // stretch of the batch's chunks (chunk_prefix: the utterances' chunk counts summed up by a small
  // kernel in front, so ragged batches are dealt evenly too) plus one halo chunk in front of and
  // behind every piece of an utterance inside it, and keeps the logged coefficients of the last two
  // chunks in registers: in the row-segment walk a filter's four frames end up in ONE lane, so with
  // the chunk just computed a lane holds twelve consecutive frames of its filter -- the reach of the
  // order-2 taps around the middle chunk, whose deltas it then forms in float32 (the reference
  // accumulates in float64 and rounds: |difference| of a few float32 ulps of the statics, inside the
  // feature tolerance) and stores.  No static is read back from memory, no second kernel; the price is
  // three waves per SIMD instead of four (the window registers) and ~14 multiply-adds per delta.  The
  // energy column has no filter lane: a spare lane of the walk (dl_eslot) collects the four frames'
  // energies and differentiates them like a filter.
  // (DLT is 0 or 2: a launch with Deltas(1) runs the order-2 instantiation with p.dl_order = 1 -- the window
  // registers are the same, and every template axis multiplies the instantiation matrix.  Round 3: fused
  // pre-emphasis and float64 samples too -- the reference drivers' chain float64 audio -> Preemphasize ->
  // compute_full -> Deltas, command_line.py:345-350, in one launch.)
  static_assert(DLT == 0 || (DLT == 2 && N2 == 16 && RSG && std::is_same<TOUT, float>::value && MF == 0),
                "fused deltas: row-segment walk, float32 features");
  static_assert(MF == 0 || std::is_same<TIN, float>::value, "matrix-pipe front end: float32 samples");
  static_assert(std::is_same<TIN, float>::value || std::is_same<TIN, double>::value || std::is_same<TIN, int16_t>::value, "samples: float32, float64 or int16");
  using G = WaveGeom<N1, N2, NROWS>;
  constexpr int N = G::N, COLS = G::COLS, NREG = G::NREG, RS = G::RS, NB = G::NB, PSTR = G::PSTR;
  constexpr int MSLOTS = 2 * MF + 1;  // per-lane sample slots of a frame (MF)
  static_assert(MF == 0 || (N2 == 16 && N1 == 32 && MF == mfma_front_steps(NROWS)), "matrix-pipe front end: 32 x 16");
  static_assert(MF == 0 || 4 * MSLOTS * 64 <= G::EXCH_F2 * 2, "edge-frame gather reuses the exchange area");
  constexpr int NBP = (NB + 3) / 4 * 4;  // RSG: bins kept in LDS (bin NBP = dump slot), as build_rseg sizes them
  static_assert(!RSG || (G::GROUPS == 4 && ELL_LDS && !SEG && (NBP + 1) * 4 <= G::EXCH_F2 * 2),
                "row-segment walk: four frames per wave, tables in LDS");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int g = lane / N2, r = lane % N2;
  // PAIR (float64 samples, 16 lanes per frame): the frame is loaded as 16-byte pairs of samples --
  // an 8-byte load instruction of a wave costs the memory path as much as a 16-byte one
  // (tools/vmem_microbench) -- so a lane receives residues 2s and 2s + 1 of one row, lanes 0..7 of the
  // even rows, lanes 8..15 of the odd ones; lane s and lane s + 8 then swap one sample of every pair
  // (a masked row_ror:8 each way) and lane s owns residue 2s, lane s + 8 residue 2s + 1 of EVERY row.
  // `rho` is the residue (n2) whose samples, window and twiddles the lane holds until the exchange;
  // it is `r` everywhere else.
  constexpr bool F64S = std::is_same<TIN, double>::value;  // float64 samples (int16 samples convert like float32 ones)
  constexpr bool PAIR = F64S && N2 == 16 && MF == 0;
  const int rho = PAIR ? ((r & 7) * 2 + (r >> 3)) : r;
  constexpr int LOADSPAN = PAIR ? (NROWS + 1) / 2 * 32 : NROWS * N2;  // samples a frame's direct loads reach over
  [[maybe_unused]] const unsigned long long st_entry = PDS_STAMPS ? __builtin_readcyclecounter() : 0;
  float *wbase = smem + wave * p.area_f;
  // (STR launches with fused CMVN sums: the wave's float64 sums, zero at the start of every piece)
  [[maybe_unused]] double *wstat = reinterpret_cast<double *>(smem + p.lds_stat_off) + wave * (2 * p.stat_cs);
  // (compiled into the stretch-scheduled kernels of the segment walks, and of the row-segment walk where the
  // geometry has 256 registers: in the 128-register row-segment kernel the sums cost four spilled registers)
  constexpr bool STATS = STR && G::GROUPS == 4 && (SEG != 0 || (RSG && MINW <= 2));
  [[maybe_unused]] const bool stats_on = STATS && p.stat_part != nullptr;
  // the coefficient `col` of the item's four frames (the first `nf` of them exist), held by ONE lane
  [[maybe_unused]] auto stat_add4 = [&](const int col, const float (&v)[4], const int nf) {
    double t1 = 0.0, t2 = 0.0;
#pragma unroll
    for (int q = 0; q < 4; ++q)
      if (q < nf) {
        const double d = (double)v[q];
        t1 += d;
        t2 = fma(d, d, t2);
      }
    wstat[col] += t1;
    wstat[p.stat_cs + col] += t2;
  };
  float2 *exch = reinterpret_cast<float2 *>(wbase) + g * G::XROWS * RS;
  [[maybe_unused]] float *xf = wbase + g * (G::XROWS * G::RSF);  // LEAN: the frame's exchange block, one float per element
  float *Pg = wbase + g * PSTR;
  // filter table -> LDS once per workgroup (read by every wave, every iteration): per-slot
  // weight rows.  Tables too large for LDS (dense complex banks) stay in global memory and
  // come through L1/L2.
  const int nthreads = p.waves * 64;
  float2 *sw_lds = reinterpret_cast<float2 *>(smem + p.waves * p.area_f);  // [N2]
  int *meta_lds = reinterpret_cast<int *>(sw_lds + N2);
  float *ellw_lds = reinterpret_cast<float *>(meta_lds + p.ell_meta_pad);
  // twiddles regenerated per item from three seeds (inl::twiddle_chain) and a window times 1/2: the float64-sample
  // instantiations of the 16-lane geometries, whose 16-byte pair loads keep 4 registers per pair in flight (52 at 25
  // rows: with thirty registers of twiddles beside them the kernel spilled 10 ... 34 registers), and the prefetch
  // experiment
  constexpr bool TWCHAIN = (PF && PDS_PF_TW == 1) || (PAIR && (N1 == 32 || N1 == 64)) || (PDS_DLT_CHAIN && DLT > 0 && N1 == 32 && N2 == 16) || G::LEAN ||
                           (N1 == 64 && N2 == 16 && (SEG == 2 || PDS_LEAN_1024));  // (N = 1024: the matrix-pipe walk's launches, which run three waves per SIMD; PDS_LEAN_1024: every kernel of the geometry)
  constexpr bool WINLDS = PF && PDS_PF_WIN == 1;  // window slice re-read from LDS per item (in front of the item: prefetch experiment)
  constexpr bool WINUSE = G::LEAN || (N1 == 64 && N2 == 16 && (SEG == 2 || PDS_LEAN_1024));  // ... read from LDS where it is applied
  constexpr int WSTR = win_table_stride(NROWS);
  [[maybe_unused]] float *win_lds = ellw_lds + (ELL_LDS ? p.ell_wfloats : 0);  // [N2][WSTR]
  if constexpr (WINLDS || WINUSE) {
    const float *wsrc = TWCHAIN ? p.win_half : p.win_lane;
    for (int i = threadIdx.x; i < N2 * WSTR; i += p.waves * 64) {
      const int rr = i / WSTR, k = i - rr * WSTR;
      win_lds[i] = k < NROWS ? wsrc[rr * N1 + k] : 0.0f;
    }
  }
  if (threadIdx.x < N2) sw_lds[threadIdx.x] = p.tw_special[threadIdx.x];
  // DYN: the workgroup's waves draw their items from one ticket counter in LDS instead of taking every
  // (grid waves)-th item each.  The four waves of a SIMD do not run at the same pace (the issue arbiter
  // prefers the older wave, and the priorities that keep them staggered add to it): with equal shares
  // the waves of one CU finished between 430 000 and 500 000 cycles of a launch (tools/wave_spread.py),
  // i.e. the last tenth of a launch ran on a draining CU.  Ticket t of workgroup w is item
  // (t / waves) * (grid waves) + w * waves + t % waves -- the same items as before, so neighbouring
  // waves still work on neighbouring chunks at about the same time -- and a wave fetches its next
  // ticket (one ds_add_rtn) at the top of an item, a filter walk ahead of needing it.
  constexpr bool DYN = PDS_DYN && !STRETCH && !PF;
  [[maybe_unused]] int *ticket_lds = reinterpret_cast<int *>(smem + p.lds_ticket_off);
  if constexpr (DYN) {
    if (threadIdx.x == 0) *ticket_lds = p.waves;  // (tickets 0 .. waves - 1 are the waves' first items)
  }
  if constexpr (STR && G::GROUPS == 4 && (SEG != 0 || (RSG && MINW <= 2))) {
    if (p.stat_part) {
      double *z = reinterpret_cast<double *>(smem + p.lds_stat_off);
      for (int i = threadIdx.x; i < p.waves * 2 * p.stat_cs; i += p.waves * 64) z[i] = 0.0;
    }
  }
  // (slots beyond the table read as "no filter": the unrolled slot loop fetches USLOTS entries)
  for (int i = threadIdx.x; i < p.ell_meta_pad; i += nthreads)
    meta_lds[i] = i < p.ell_meta_ints ? p.ell_meta[i] : 0;
  // (16 bytes per thread and pass: every table is a multiple of four floats long and starts on 16 bytes, and
  // so do the wave areas -- a launch's prologue is ~10 us of its ~270, and a pass of this loop a trip to L2)
#if PDS_FAST_PROLOGUE
  static_assert((G::EXCH_F2 * 2) % 4 == 0, "wave areas are zeroed 16 bytes at a time");
  if constexpr (ELL_LDS) {
    const float4 *src4 = reinterpret_cast<const float4 *>(p.ell_w);
    float4 *dst4 = reinterpret_cast<float4 *>(ellw_lds);
    for (int i = threadIdx.x; i < (p.ell_wfloats >> 2); i += nthreads) dst4[i] = src4[i];
  }
  // the wave areas start out zeroed so that never-written P padding is finite
  {
    float4 *z4 = reinterpret_cast<float4 *>(smem);
    for (int i = threadIdx.x; i < p.waves * (p.area_f / 4); i += nthreads) z4[i] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
  }
#else
  if constexpr (ELL_LDS)
    for (int i = threadIdx.x; i < p.ell_wfloats; i += nthreads) ellw_lds[i] = p.ell_w[i];
  // the wave areas start out zeroed so that never-written P padding is finite
  for (int i = threadIdx.x; i < p.waves * p.area_f; i += nthreads) smem[i] = 0.0f;
#endif
  __syncthreads();

  // per-lane constants, loop invariant: window slice and inter-stage twiddles (issued in front of the LDS
  // set-up instead: measured slower, 0.0328 against 0.0308 ms at 64 utterances)
  [[maybe_unused]] float win[(MF || WINLDS || WINUSE) ? 1 : NROWS];
  [[maybe_unused]] float twr[(MF || TWCHAIN) ? 1 : COLS], twi[(MF || TWCHAIN) ? 1 : COLS];
  [[maybe_unused]] float sd[6] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};  // TWCHAIN: the seeds W^r, W^4r, W^8r
  // MF: window and byte offset of the lane's sample slots, A operands of the two chains, twiddles of
  // the lane's four output rows
  [[maybe_unused]] float mwin[MSLOTS], mare[MF + 1], maim[MF + 1], mtwr[4], mtwi[4];
  [[maybe_unused]] unsigned moff[MSLOTS];
  if constexpr (MF > 0) {
    const float *tab = p.mf_tab;
#pragma unroll
    for (int sl = 0; sl < MSLOTS; ++sl) {
      mwin[sl] = tab[sl * 64 + lane];
      moff[sl] = (unsigned)reinterpret_cast<const int *>(tab)[(MSLOTS + sl) * 64 + lane] * 4u;
    }
#pragma unroll
    for (int t = 0; t <= MF; ++t) {
      mare[t] = tab[(3 * MSLOTS + t) * 64 + lane];
      maim[t] = tab[(3 * MSLOTS + MF + 1 + t) * 64 + lane];
    }
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      const float2 t = reinterpret_cast<const float2 *>(tab + (3 * MSLOTS + 2 * (MF + 1)) * 64)[v * 64 + lane];
      mtwr[v] = t.x;
      mtwi[v] = t.y;
    }
  } else {
    if constexpr (!WINLDS && !WINUSE) {
      const float *wsrc = TWCHAIN ? p.win_half : p.win_lane;
#pragma unroll
      for (int n1 = 0; n1 < NROWS; ++n1) win[n1] = wsrc[rho * N1 + n1];
    }
    if constexpr (TWCHAIN) {
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const float2 t = p.tw_seed[rho * 3 + j];
        sd[2 * j] = t.x;
        sd[2 * j + 1] = t.y;
      }
    } else {
#pragma unroll
      for (int k1 = 1; k1 <= NREG; ++k1) {
        const float2 t = p.tw_lane[rho * COLS + k1];
        twr[k1] = t.x;
        twi[k1] = t.y;
      }
    }
  }
  // PF: the next item's samples (in flight from the middle of an item to the top of the next) and whether
  // they are the next item's at all (a regular item: all frames exist, every row inside the signal)
  [[maybe_unused]] float pfv[PF ? NROWS : 1];
  // WINLDS: the window slice lives in registers from the end of an item's filter walk to the top of the next item
  // only (read from the LDS table behind the walk's own reads, so that the round trip is not the first thing an
  // item waits for); the column transforms, where the registers are scarce, run without it
  [[maybe_unused]] float wl[WINLDS ? NROWS : 1];
  [[maybe_unused]] auto read_window = [&]() {
    if constexpr (WINLDS) {
      const float4 *w4 = reinterpret_cast<const float4 *>(win_lds + rho * WSTR);
#pragma unroll
      for (int j = 0; j < (NROWS + 3) / 4; ++j) {
        const float4 w = w4[j];
        const float wv[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
        for (int u = 0; u < 4; ++u)
          if (4 * j + u < NROWS) wl[4 * j + u] = wv[u];
      }
    }
  };
  read_window();
  [[maybe_unused]] bool pf_ready = false;
  [[maybe_unused]] const int pf_lane = __mul24(g, p.S) + r;  // the lane's first sample inside an item's stretch
  const int L = p.L, S = p.S;
  const bool use_power = p.use_power != 0;
  const int col0 = p.include_energy ? 1 : 0;

  // work items: (utterance, chunk of GROUPS consecutive frames), dealt round-robin to all waves
  // of the grid: the waves of a workgroup take neighbouring chunks, so overlapping samples are
  // shared through the CU's L1.  The step to a wave's next item is a precomputed (utterances,
  // chunks) pair -- scalar instructions compete with vector ones for issue slots, so the
  // bookkeeping per item is kept to a handful of them.
  int b = 0;
  // Workgroups are handed to the 8 XCDs round-robin (workgroup i runs on XCD i % 8, each with its
  // own L2): renumber them so that an XCD's workgroups take neighbouring chunks and the samples
  // two neighbours share (L - S per frame) are fetched into one L2 instead of two.
  int wg = blockIdx.x;
#ifndef PDS_NO_XCD_MAP
  if ((gridDim.x & 7) == 0) wg = (wg & 7) * (gridDim.x >> 3) + (wg >> 3);
#endif
  int chunk = wg * p.waves + wave;
  if constexpr (!STRETCH) {
    while (chunk >= p.chunks_per_utt && b < p.num_utts) {  // once per kernel
      chunk -= p.chunks_per_utt;
      ++b;
    }
  }
  // DYN: (rb, rc) = utterance and chunk of the workgroup's first item of round `rnd` (items
  // rnd * (grid waves) + wg * waves + 0 .. waves - 1); a wave moves it forward to its ticket's round
  [[maybe_unused]] int rnd = 0, rb = 0, rc = wg * p.waves;
  if constexpr (DYN) {
    while (rc >= p.chunks_per_utt && rb < p.num_utts) {
      rc -= p.chunks_per_utt;
      ++rb;
    }
  }
  // sets (b, chunk) to ticket t's item; b >= num_utts: no such item (and none for any later ticket)
  [[maybe_unused]] auto take_ticket = [&](const int t) {
    const int jn = (int)__umulhi((unsigned)t, p.waves_rcp), o = t - jn * p.waves;
    while (rnd < jn && rb < p.num_utts) {  // (one step per item taken in the steady state)
      rc += p.step_chunks;
      rb += p.step_utts;
      if (rc >= p.chunks_per_utt) {
        rc -= p.chunks_per_utt;
        ++rb;
      }
      ++rnd;
    }
    b = rb;
    chunk = rc + o;
    while (chunk >= p.chunks_per_utt && b < p.num_utts) {
      chunk -= p.chunks_per_utt;
      ++b;
    }
  };
  [[maybe_unused]] int ticket_v = 0;  // lane 0: the wave's next ticket, in flight from the top of an item
  [[maybe_unused]] auto fetch_ticket = [&]() {
    if (lane == 0) ticket_v = __hip_atomic_fetch_add(ticket_lds, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  };
  // the next item's (b, chunk) and its utterance record
  [[maybe_unused]] auto next_item_dyn = [&](int &nfr_, int &n_, int64_t &sig_off_, int64_t &out_row_) {
    const int last_b = b;
    take_ticket(__builtin_amdgcn_readfirstlane(ticket_v));
    const int pb_ = b < p.num_utts ? b : last_b;  // (any valid record past the end)
    nfr_ = (int)load_const(p.nframes + pb_);
    n_ = (int)load_const(p.lengths + pb_);
    sig_off_ = load_const(p.offsets + pb_);
    out_row_ = load_const(p.row_off + pb_);
  };
  // The utterance record of an item (frame count, length, signal offset, output row) is fetched
  // during the filter phase of the item before: four dependent scalar-load round trips at the
  // top of every item otherwise.
  int nfr = 0, n = 0;
  int64_t sig_off = 0, out_row = 0;
  if (!STRETCH && b < p.num_utts) {
    nfr = (int)load_const(p.nframes + b);  // (frame and sample counts fit an int: host check)
    n = (int)load_const(p.lengths + b);
    sig_off = load_const(p.offsets + b);
    out_row = load_const(p.row_off + b);
  }
  // (DLT) the wave's stretch of the batch's chunks: `left` chunks still to emit starting with chunk
  // e_lo of utterance b.  A PIECE is the part of one utterance inside the stretch, chunks [e_lo, e_hi):
  // the wave computes chunks max(e_lo - 1, 0) .. min(e_hi, last chunk) (the first and the last one only
  // for their statics: halos), stores the statics of the piece's chunks and, with chunk c computed,
  // the deltas of chunk c - 1.  Frames in front of the utterance repeat its first frame, frames
  // behind it the last one (post.py:447 "edge"); the lanes of a chunk's frames past the utterance's
  // end compute the last frame again (see the loads), so the last chunk's window is already padded.
  constexpr int DR = 2;  // row-segment rounds a fused-deltas launch may have (window registers per round)
  [[maybe_unused]] int left = 0, e_lo = 0, e_hi = 0, c_last = -1;
  [[maybe_unused]] bool piece_open = false, ran = false;
  [[maybe_unused]] float Wp[DR][4], Wc[DR][4], e_keep = 0.0f;
  if constexpr (STRETCH) {
    const int gw = wg * p.waves + wave, GW = (int)gridDim.x * p.waves;
    const int total = (int)load_const(p.chunk_prefix + p.num_utts);  // (fits an int: host check)
    const int per = total / GW, rem = total - per * GW;
    const int pos = gw * per + (gw < rem ? gw : rem);
    left = per + (gw < rem ? 1 : 0);
    // the utterance holding chunk `pos`: the last one whose prefix is <= pos (empty ones are skipped)
    int lo = 0, hi = p.num_utts;  // prefix[lo] <= pos < prefix[hi] (when left > 0)
    while (left > 0 && hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if ((int)load_const(p.chunk_prefix + mid) <= pos) lo = mid; else hi = mid;
    }
    b = lo;
    e_lo = left > 0 ? pos - (int)load_const(p.chunk_prefix + lo) : 0;
    chunk = 0;
#pragma unroll
    for (int rd = 0; rd < DR; ++rd)
#pragma unroll
      for (int q = 0; q < 4; ++q) Wp[rd][q] = Wc[rd][q] = 0.0f;
  }
  // The rows of chunk `ce` of the utterance in hand, statics AND deltas, from the window (Wp, Wc, nx) =
  // frames 4 ce - 4 .. 4 ce + 7: the lanes form the deltas of their coefficients, everything is put down
  // row-major in the wave's LDS area (free between the filter walk and the next exchange) and leaves as
  // ONE contiguous store per row, 16 bytes per lane.  (Stored straight from the lanes that hold them --
  // a dword per lane and frame, some forty lanes per instruction, 25 instructions per item -- the
  // stores alone cost a third of the launch: the vector memory path handles such an instruction lane
  // by lane.)
  [[maybe_unused]] auto emit_rows = [&](const int ce, const float (&nx)[DR][4]) {
    if (p.dl_debug & 2) return;
    const int C = p.dl_inner;
    const int W = (p.dl_order + 1) * C, WS = (W + 3) & ~3;  // row width, staged row stride (floats)
    float *stage = wbase;
    wave_sync();  // (the filter walk's reads of the area are done: same wave, in order)
#pragma unroll
    for (int rd = 0; rd < DR; ++rd) {
      if (rd >= p.seg_rounds) break;
      // the lane's column: its filter (first lane of a run), the energy (dl_eslot), or none: a dump slot
      const int f = (meta_lds[rd * 64 + lane] >> 16) - 1;
      const int col = f >= 0 ? col0 + f : (p.dl_eslot == rd * 64 + lane ? 0 : -1);
      const float v[12] = {Wp[rd][0], Wp[rd][1], Wp[rd][2], Wp[rd][3], Wc[rd][0], Wc[rd][1],
                           Wc[rd][2], Wc[rd][3], nx[rd][0], nx[rd][1], nx[rd][2], nx[rd][3]};
      float *mine = stage + (col >= 0 ? col : 4 * WS + lane);
      const int wrow = col >= 0 ? WS : 0, wk = col >= 0 ? C : 0;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        float d = p.dl_f1[0] * v[q + 2];
#pragma unroll
        for (int t = 1; t < 5; ++t) d = fmaf(p.dl_f1[t], v[q + 2 + t], d);
        mine[q * wrow] = v[4 + q];
        mine[q * wrow + wk] = d;
        if (p.dl_order == 2) {
          float dd = p.dl_f2[0] * v[q];
#pragma unroll
          for (int t = 1; t < 9; ++t) dd = fmaf(p.dl_f2[t], v[q + t], dd);
          mine[q * wrow + 2 * wk] = dd;
        }
      }
    }
    wave_sync();
    if (!(p.dl_debug & 1)) {
      float *orow = static_cast<float *>(p.out) + (out_row + (int64_t)ce * 4) * p.out_stride;
      const int fh = nfr - ce * 4;  // frames of the chunk that exist (>= 1)
      const int full = W >> 2, tail = W & 3;
      typedef float F4 __attribute__((ext_vector_type(4), aligned(4)));  // rows start on any float
      if (fh >= 4 && full <= 64) {
        // (the common case: four reads in flight, then four stores, one lane predicate)
        const int l = lane < full ? lane : full - 1;
        float4 val[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) val[q] = *reinterpret_cast<const float4 *>(stage + q * WS + 4 * l);
        if (lane < full) {
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            F4 o = {val[q].x, val[q].y, val[q].z, val[q].w};
            *reinterpret_cast<F4 *>(orow + (int64_t)q * p.out_stride + 4 * l) = o;
          }
        }
      } else {
        for (int l0 = 0; l0 < full; l0 += 64) {
          const int l = l0 + lane;
          if (l < full) {
#pragma unroll 1
            for (int q = 0; q < 4; ++q)
              if (q < fh) {
                const float4 val = *reinterpret_cast<const float4 *>(stage + q * WS + 4 * l);
                F4 o = {val.x, val.y, val.z, val.w};
                *reinterpret_cast<F4 *>(orow + (int64_t)q * p.out_stride + 4 * l) = o;
              }
          }
        }
      }
      if (tail) {  // the last W % 4 columns of the four rows: one dword store
        const int q = tail == 1 ? lane : tail == 2 ? lane >> 1 : (lane * 11) >> 5;  // lane / tail where it is < 4
        const int jj = lane - q * tail;
        if (q < fh && q < 4) orow[(int64_t)q * p.out_stride + 4 * full + jj] = stage[q * WS + 4 * full + jj];
      }
    }
    wave_sync();
  };
  [[maybe_unused]] unsigned long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  [[maybe_unused]] unsigned long long st_prev = PDS_STAMPS ? __builtin_readcyclecounter() : 0;
  [[maybe_unused]] const unsigned long long st_loop = st_prev;
  while (STRETCH || b < p.num_utts) {
    if constexpr (STRETCH) {
      if (!piece_open) {
        if (left <= 0) break;
        nfr = (int)load_const(p.nframes + b);
        const int chunks_b = (nfr + G::GROUPS - 1) / G::GROUPS;
        if (e_lo >= chunks_b) {  // an utterance without frames (only the first piece starts past chunk 0)
          ++b;
          e_lo = 0;
          continue;
        }
        n = (int)load_const(p.lengths + b);
        sig_off = load_const(p.offsets + b);
        out_row = load_const(p.row_off + b);
        e_hi = e_lo + left < chunks_b ? e_lo + left : chunks_b;
        // (DLT: one halo chunk on either side where the utterance has one; STR: the piece's own chunks)
        c_last = DLT > 0 ? (e_hi < chunks_b ? e_hi : chunks_b - 1) : e_hi - 1;
        chunk = DLT > 0 ? (e_lo > 0 ? e_lo - 1 : 0) : e_lo;
        piece_open = true;
        ran = false;
      }
      if constexpr (STR) {  // (the item of the pass before is done: on to the next chunk)
        if (ran) ++chunk;
        ran = true;
      }
      if (chunk > c_last) {
        // the piece's chunks are computed.  Where it runs to the utterance's end the last chunk's deltas
        // are still due: the frames behind it repeat the last frame, which is frame 3 of the last chunk
        // whatever the frame count (see above)
        if (DLT > 0 && c_last < e_hi) {
          float nx[DR][4];
#pragma unroll
          for (int rd = 0; rd < DR; ++rd)
#pragma unroll
            for (int q = 0; q < 4; ++q) nx[rd][q] = Wc[rd][3];
          emit_rows(c_last, nx);
        }
        if constexpr (STATS) {
          if (stats_on) {
            // the piece's sums: slot (global wave + utterance) -- pieces in chunk order differ in one of the two
            wave_sync();
            double *dst = p.stat_part + ((int64_t)(wg * p.waves + wave) + b) * (2 * p.stat_c);
            for (int c = lane; c < p.stat_c; c += 64) {
              dst[c] = wstat[c];
              dst[p.stat_c + c] = wstat[p.stat_cs + c];
              wstat[c] = 0.0;
              wstat[p.stat_cs + c] = 0.0;
            }
            wave_sync();
          }
        }
        left -= e_hi - e_lo;
        ++b;
        e_lo = 0;
        piece_open = false;
        continue;
      }
    }
    int nchunk = chunk + p.step_chunks, nb = b + p.step_utts;
    if (nchunk >= p.chunks_per_utt) {
      nchunk -= p.chunks_per_utt;
      ++nb;
    }
    const int pb = nb < p.num_utts ? nb : b;  // record to fetch (any valid one past the end)
    const int tb = chunk * G::GROUPS;  // first frame of the chunk (frames * S fits an int)
    if constexpr (DYN) {
      if (p.dyn) fetch_ticket();
    }
    if (!STRETCH && tb >= nfr) {  // uniform: utterance shorter than the longest
      if (DYN && p.dyn) {
        next_item_dyn(nfr, n, sig_off, out_row);
        continue;
      }
      b = nb;
      chunk = nchunk;
      nfr = (int)load_const(p.nframes + pb);
      n = (int)load_const(p.lengths + pb);
      sig_off = load_const(p.offsets + pb);
      out_row = load_const(p.row_off + pb);
      continue;
    }
    PDS_STAMP(7, 0);  // item bookkeeping (and whatever the previous item left undrained)
    if constexpr (PDS_STAMPS != 0) ++st_acc[6];
    PDS_PHASE(0);
    // PF: the next item's record now (it is needed in the middle of this item, for the prefetch)
    [[maybe_unused]] int nfr_nx = 0, n_nx = 0;
    [[maybe_unused]] int64_t sig_off_nx = 0, out_row_nx = 0;
    // (issued behind the window multiplies: scalar loads return out of order, so the next wait for an LDS read
    // is a wait for them too -- from there the in-lane transform covers them)
    [[maybe_unused]] auto fetch_next_record = [&]() {
      __builtin_amdgcn_sched_barrier(0);
      nfr_nx = (int)load_const(p.nframes + pb);
      n_nx = (int)load_const(p.lengths + pb);
      sig_off_nx = load_const(p.offsets + pb);
      out_row_nx = load_const(p.row_off + pb);
    };
    // the prefetch: 25 loads whatever the next item is -- an item that is not regular (or none) re-reads the
    // lane's window slice instead, so that the loads sit in straight-line code the scheduler may spread over
    // the arithmetic that follows (a branch around them would pin them in a block of their own)
    [[maybe_unused]] auto issue_prefetch = [&]() {
      const int tb2 = nchunk * G::GROUPS, s02 = tb2 * S - p.pad_left;
      pf_ready = nb < p.num_utts && tb2 + G::GROUPS <= nfr_nx && s02 >= 0 && s02 + (G::GROUPS - 1) * S + LOADSPAN <= n_nx;
      const float *base = pf_ready ? static_cast<const float *>(p.sig) + (sig_off_nx + s02) : p.win_lane;
      const float *xp2 = base + (pf_ready ? pf_lane : r);
#pragma unroll
      for (int n1 = 0; n1 < NROWS; ++n1) {
        pfv[n1] = xp2[n1 * N2];
#if PDS_PF_ILV > 0
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);          // one load
        __builtin_amdgcn_sched_group_barrier(0x002, PDS_PF_ILV, 0);  // then vector instructions of what follows
#endif
      }
    };
    const TIN *x = static_cast<const TIN *>(p.sig) + sig_off;
    const int start0 = tb * S - p.pad_left;
    bool valid = true;
    float energy = 0.0f;
    // W_(2 N2)^r, twiddle of lane r in the distributed untangling of the even/odd-sum column:
    // re-read from LDS every iteration rather than held in two registers per lane -- the kernel
    // sits exactly at the 128-VGPR occupancy step
    const float2 sw = sw_lds[r];
    [[maybe_unused]] float even_sum, odd_sum, Ar[COLS], Ai[COLS];  // (in-lane front end)
    if constexpr (MF > 0) {
      // ---- matrix-pipe front end (mfma_front.h).  Here a lane is (k-slot q = g, residue r) of
      // EVERY frame of the item; frame starts are scalars.
      int fstart[4];
#pragma unroll
      for (int fg = 0; fg < 4; ++fg) fstart[fg] = start0 + fg * S;
      int wmode = 0;
      if (!(tb + 4 <= nfr && start0 >= (PRE ? 1 : 0) && start0 + 3 * S + NROWS * 16 <= n)) {
        // frames past the utterance's last one recompute the last frame (their rows are never
        // stored); frames touching a signal end are gathered with reflected indices
        valid = tb + g < nfr;
#pragma unroll
        for (int fg = 0; fg < 4; ++fg) {
          const int st = (tb + fg < nfr ? tb + fg : nfr - 1) * S - p.pad_left;
          fstart[fg] = st;
          int mode = (st < (PRE ? 1 : 0) || st + NROWS * 16 > n) ? 1 : 0;
          if (st < -n || st + L > 2 * n) mode = 2;
          wmode = mode > wmode ? mode : wmode;
        }
      }
      float xs[4][MSLOTS];
      if (wmode == 0) {
#pragma unroll
        for (int fg = 0; fg < 4; ++fg) {
          // scalar frame base + the lane's 32-bit byte offset: no 64-bit vector arithmetic
          const char *xg = reinterpret_cast<const char *>(reinterpret_cast<const float *>(x) + fstart[fg]);
#pragma unroll
          for (int sl = 0; sl < MSLOTS; ++sl) {
            float v = (PDS_ABLATE & 1) ? (float)(lane + sl) : *reinterpret_cast<const float *>(xg + moff[sl]);
            if constexpr (PRE) v = preemph_sample(v, *reinterpret_cast<const float *>(xg + moff[sl] - 4), p.preemph);
            xs[fg][sl] = v;
          }
        }
      } else {
        float *tmp = wbase;
        const int *offs = reinterpret_cast<const int *>(p.mf_tab) + MSLOTS * 64 + lane;
#pragma unroll 1
        for (int fg = 0; fg < 4; ++fg) {
          const int st = fstart[fg];
#pragma unroll 1
          for (int sl = 0; sl < MSLOTS; ++sl) {
            int i = st + offs[sl * 64];
            if (wmode == 1) {
              i = i < 0 ? -1 - i : (i >= n ? 2 * n - 1 - i : i);
            } else {
              i = (int)reflect_index((int64_t)i, (int64_t)n);
            }
            float v = (float)x[i];
            if (PRE && i > 0) v = preemph_sample(v, (float)x[i - 1], p.preemph);
            tmp[(fg * MSLOTS + sl) * 64 + lane] = v;
          }
        }
        wave_sync();
#pragma unroll
        for (int fg = 0; fg < 4; ++fg)
#pragma unroll
          for (int sl = 0; sl < MSLOTS; ++sl) xs[fg][sl] = tmp[(fg * MSLOTS + sl) * 64 + lane];
        wave_sync();
      }
      if (p.include_energy) {
        // compute.py:392-393 on the un-windowed samples: a slot counts where it is a sample of the
        // frame of its own (mask table); lane sums, then the 16 lanes of a row, then the four rows
        const float *em = p.mf_tab + 2 * MSLOTS * 64 + lane;
        float eg[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int sl = 0; sl < MSLOTS; ++sl) {
          const float m = em[sl * 64];
#pragma unroll
          for (int fg = 0; fg < 4; ++fg) {
            const float xm = mul_legacy(xs[fg][sl], m);
            eg[fg] = fmaf(xm, xm, eg[fg]);
          }
        }
#pragma unroll
        for (int fg = 0; fg < 4; ++fg) {
          float e = eg[fg];
          e = dpp_add<0xB1>(e);   // quad_perm [1,0,3,2]
          e = dpp_add<0x4E>(e);   // quad_perm [2,3,0,1]
          e = dpp_add<0x141>(e);  // row_half_mirror
          e = dpp_add<0x140>(e);  // row_mirror: every lane of a row holds the row's sum
          const int ei = __float_as_int(e);
          const float tot = (__int_as_float(__builtin_amdgcn_readlane(ei, 0)) +
                             __int_as_float(__builtin_amdgcn_readlane(ei, 16))) +
                            (__int_as_float(__builtin_amdgcn_readlane(ei, 32)) +
                             __int_as_float(__builtin_amdgcn_readlane(ei, 48)));
          if (g == fg) energy = tot;
        }
      }
      PDS_STAMP(0, 0);  // frame loads issued
      PDS_STAMP(1, 1);  // ... and arrived
      PDS_PHASE(4);
      float sv[4][MF], dv[4][MF], cv[4], uv[4];
#pragma unroll
      for (int fg = 0; fg < 4; ++fg) {
        // (plain multiplies: a slot without a sample of its own re-reads a sample of the SAME frame
        // with weight zero, so nothing from outside the frame can reach it)
        cv[fg] = xs[fg][2 * MF] * mwin[2 * MF];
        float u = cv[fg];
#pragma unroll
        for (int t = 0; t < MF; ++t) {
          const float a = xs[fg][2 * t] * mwin[2 * t];
          sv[fg][t] = fmaf(xs[fg][2 * t + 1], mwin[2 * t + 1], a);
          dv[fg][t] = fmaf(-xs[fg][2 * t + 1], mwin[2 * t + 1], a);
          u += sv[fg][t];
        }
        uv[fg] = u;
      }
      f32x4 accR[4], accI[4];
#pragma unroll
      for (int fg = 0; fg < 4; ++fg) {
        accR[fg] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        accI[fg] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
      }
      if constexpr (!(PDS_ABLATE & 2)) {
#pragma unroll
        for (int t = 0; t < MF; ++t)
#pragma unroll
          for (int fg = 0; fg < 4; ++fg) {
            accR[fg] = __builtin_amdgcn_mfma_f32_16x16x4f32(mare[t], sv[fg][t], accR[fg], 0, 0, 0);
            accI[fg] = __builtin_amdgcn_mfma_f32_16x16x4f32(maim[t], dv[fg][t], accI[fg], 0, 0, 0);
          }
#pragma unroll
        for (int fg = 0; fg < 4; ++fg) {
          accR[fg] = __builtin_amdgcn_mfma_f32_16x16x4f32(mare[MF], cv[fg], accR[fg], 0, 0, 0);
          accI[fg] = __builtin_amdgcn_mfma_f32_16x16x4f32(maim[MF], uv[fg], accI[fg], 0, 0, 0);
        }
      } else {
#pragma unroll
        for (int fg = 0; fg < 4; ++fg) {
          accR[fg] = f32x4{sv[fg][0], sv[fg][1], sv[fg][2], cv[fg]};
          accI[fg] = f32x4{dv[fg][0], dv[fg][1], dv[fg][2], uv[fg]};
        }
      }
      // the lane's rows 4 q + v are columns k1 = 4 q + v + 1 of every frame; row 15 (q = 3, v = 3)
      // holds the even / odd row sums, which go to row 0 of the frame's exchange block
      PDS_PHASE(1);
      float2 *mine = reinterpret_cast<float2 *>(wbase) + (4 * g + 1) * RS + r;
      float *row0 = wbase + r;
#pragma unroll
      for (int fg = 0; fg < 4; ++fg) {
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          float2 val;
          val.x = accR[fg][v] * mtwr[v] - accI[fg][v] * mtwi[v];
          val.y = accR[fg][v] * mtwi[v] + accI[fg][v] * mtwr[v];
          if (v < 3 || g != 3) {
            mine[fg * COLS * RS + v * RS] = val;
          } else {
            row0[fg * COLS * RS * 2] = accR[fg][3];
            row0[fg * COLS * RS * 2 + 16] = accI[fg][3];
          }
        }
      }
    } else {
      // Common case, decided on scalars: all GROUPS frames exist and every row they read lies
      // inside the signal.  Otherwise per-lane bookkeeping: lanes of a frame past the end
      // recompute the last frame (their rows are never stored); frames touching a signal end are
      // gathered with reflected indices (1: one bounce suffices, 2: general reflection).
      int start = start0 + __mul24(g, S);  // (24-bit multiplies issue at full rate, 32-bit ones at a quarter)
      int wmode = 0;
      if (!(PF && pf_ready) && !(tb + G::GROUPS <= nfr && start0 >= (PRE ? 1 : 0) &&
            start0 + (G::GROUPS - 1) * S + LOADSPAN <= n)) {
        valid = tb + g < nfr;
        start = (valid ? tb + g : nfr - 1) * S - p.pad_left;
        int mode = 0;
        // (with fused pre-emphasis the direct loads also read x[start - 1])
        if (start < (PRE ? 1 : 0) || start + LOADSPAN > n) mode = 1;
        if (start < -n || start + L > 2 * n) mode = 2;
        wmode = __builtin_amdgcn_readfirstlane(__any(mode == 2) ? 2 : (__any(mode == 1) ? 1 : 0));
      }

      float a[N1];
      if (PF && pf_ready) {
        // (a regular item by construction: wmode = 0, valid)
#pragma unroll
        for (int n1 = 0; n1 < NROWS; ++n1) a[n1] = pfv[n1];
      } else if (wmode == 0) {
        // Lanes past the frame's end in the last row read samples of the next frame; the
        // window (exactly 0 there, applied with the 0 * x = 0 multiply) removes them.
        const TIN *xp = x + (start + r);
        if constexpr (PAIR) {
          typedef double D2 __attribute__((ext_vector_type(2), aligned(8)));
          const TIN *xq = x + (start + 2 * r);  // the lane's pair inside every block of two rows
          // (all loads first: the lane exchanges below are convergent operations, which the compiler
          // does not move loads across -- interleaved in the source, every load waited for the one before)
          constexpr int NP = (NROWS + 1) / 2;
          D2 w2[NP];
          // PRE: the sample in front of a lane's pair is the second sample of the lane before (row_shr:1; lane 8's
          // is lane 7's: the end of the even row), and lane 0's the second sample of lane 15's PREVIOUS pair (the
          // end of the odd row before: row_ror:1 of that pair, kept as `old` where row_shr has no source lane) --
          // no second load per pair (round 2 loaded the predecessors: 13 more 8-byte loads on the unit this
          // kernel is bound by, and 26 more registers in flight: 34 spilled).  Only the sample in front of the
          // frame is loaded, by lane 0.
          [[maybe_unused]] double before = 0.0;
#pragma unroll
          for (int j = 0; j < NP; ++j) w2[j] = *reinterpret_cast<const D2 *>(xq + 32 * j);
          if constexpr (PRE) {
            if (r == 0) before = xq[-1];
          }
          [[maybe_unused]] auto dpp64 = [](double old, double src, auto ctrl) {
            const long long o = __double_as_longlong(old), s_ = __double_as_longlong(src);
            const int lo = __builtin_amdgcn_update_dpp((int)o, (int)s_, decltype(ctrl)::value, 0xf, 0xf, false);
            const int hi = __builtin_amdgcn_update_dpp((int)(o >> 32), (int)(s_ >> 32), decltype(ctrl)::value, 0xf, 0xf, false);
            return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
          };
#pragma unroll
          for (int j = 0; j < NP; ++j) {
            double u = w2[j].x, v = w2[j].y;
            if constexpr (PRE) {
              // float64 pre-emphasis before the rounding (pre.py:140-149): v's predecessor is u
              const double wrap = j == 0 ? before : dpp64(0.0, (double)w2[j > 0 ? j - 1 : 0].y, inl::Int<0x121>{});  // row_ror:1
              const double pu = dpp64(wrap, (double)w2[j].y, inl::Int<0x111>{});                                 // row_shr:1
              v = preemph_sample(w2[j].y, w2[j].x, (TIN)p.preemph_d);
              u = preemph_sample(w2[j].x, pu, (TIN)p.preemph_d);
            }
            const float fu = (float)u, fv = (float)v;
            // lanes 8..15 take residue 2s + 1 of the even row from lane s; lanes 0..7 take residue 2s of
            // the odd row from lane s + 8 (row_ror:8 = lane ^ 8, bank masks pick the receiving half)
            a[2 * j] = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(fu), __float_as_int(fv), 0x128, 0xf, 0xc, false));
            if (2 * j + 1 < NROWS)
              a[2 * j + 1] = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(fv), __float_as_int(fu), 0x128, 0xf, 0x3, false));
          }
        } else if constexpr (F64S) {
          // float64 samples: 8-byte loads, pre-emphasis in float64 (bit-identical to the reference's
          // own pass, pre.py:140-149), one rounding to float32
          // (with pre-emphasis -- a 16-byte load of predecessor and sample per row -- in batches of rows, all loads of
          // a batch in front of its arithmetic: written row by row, the compiler, short of registers for ROWS x 4
          // of them in flight, waited for every load before issuing the next: 25 memory round trips per item in a
          // row.  8 kHz audio as float64 with pre-emphasis: 2.63 -> 3.83 G frames/s, N = 2048: +21 %)
          // (without pre-emphasis the compiler's own order of the 8-byte loads measures 0-5 % better than any batch)
          constexpr int FB = (PRE && PDS_F64_ROW_BATCH > 0) ? PDS_F64_ROW_BATCH / 2 : NROWS;
#pragma unroll
          for (int n0 = 0; n0 < NROWS; n0 += FB) {
            TIN cur[FB], prev[FB];
#pragma unroll
            for (int i = 0; i < FB; ++i) {
              if (n0 + i < NROWS) {
                cur[i] = xp[(n0 + i) * N2];
                if constexpr (PRE) prev[i] = xp[(n0 + i) * N2 - 1];
              }
            }
            if constexpr (PRE && PDS_F64_ROW_BATCH > 0) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < FB; ++i) {
              if (n0 + i < NROWS) {
                TIN v = cur[i];
                if constexpr (PRE) v = preemph_sample(v, prev[i], (TIN)p.preemph_d);
                a[n0 + i] = (float)v;
              }
            }
          }
        } else {
#pragma unroll
        for (int n1 = 0; n1 < NROWS; ++n1) a[n1] = (PDS_ABLATE & 1) ? (float)(lane + n1) : (float)xp[n1 * N2];
        }
        if constexpr (PRE && !F64S) {  // (int16 samples: converted above, pre-emphasised like float32 ones)
          // predecessor of lane r's sample: lane r - 1 of the same row, or (r = 0) the last lane
          // of the previous row.  With 16 lanes per frame a lane group is one DPP row and
          // row_ror:1 delivers both; other group sizes load the predecessor.
          if constexpr (N2 == 16) {
#if PDS_PREEMPH_DPP
            // Two multiply-adds per row, each taking its predecessor sample through a DPP operand: lanes 1..15 from
            // their left neighbour (row_shr:1; lane 0 has no source and adds 0), lane 0 from the last lane of the
            // row before (row_shl:15: only lane 0 has a source).  Rows in descending order, so that the row before is
            // still un-emphasised when it is read.  Same products and one rounding per sample as the mov + select +
            // multiply-add form: bit-identical, 25 vector instructions fewer per item.
            float carry = 0.0f;  // the sample in front of the frame (lane 0 of the group; 0 elsewhere)
            if (r == 0) carry = (float)xp[-1];
            const float negc = -p.preemph;
            // (hand-written: the compiler keeps a v_mov_b32_dpp per predecessor instead of folding it into the
            // multiply-add.  Lanes without a DPP source are disabled for that instruction, which is the select.  A DPP
            // operand must not be read within two instructions of a vector write to it: the rows' registers come from
            // memory loads for float32 samples; converted int16 samples get the wait states.)
            inl::static_for<0, NROWS>([&](auto k) {
              constexpr int n1 = NROWS - 1 - decltype(k)::value;
              if constexpr (n1 > 0) {
                if constexpr (std::is_same<TIN, float>::value)
                  asm volatile("v_fmac_f32_dpp %0, %0, %2 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
                               "v_fmac_f32_dpp %0, %1, %2 row_shl:15 row_mask:0xf bank_mask:0xf"
                               : "+v"(a[n1]) : "v"(a[n1 - 1]), "v"(negc));
                else
                  asm volatile("s_nop 1\n\t"
                               "v_fmac_f32_dpp %0, %0, %2 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
                               "v_fmac_f32_dpp %0, %1, %2 row_shl:15 row_mask:0xf bank_mask:0xf"
                               : "+v"(a[n1]) : "v"(a[n1 - 1]), "v"(negc));
              } else {
                if constexpr (std::is_same<TIN, float>::value)
                  asm volatile("v_fmac_f32_dpp %0, %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[0]) : "v"(negc));
                else
                  asm volatile("s_nop 1\n\tv_fmac_f32_dpp %0, %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[0]) : "v"(negc));
                a[0] = fmaf(negc, carry, a[0]);
              }
            });
#else
            float carry = 0.0f;
            if (r == 0) carry = (float)xp[-1];
#pragma unroll
            for (int n1 = 0; n1 < NROWS; ++n1) {
              const float rot = __int_as_float(__builtin_amdgcn_update_dpp(
                  0, __float_as_int(a[n1]), 0x121 /* row_ror:1 */, 0xf, 0xf, false));
              const float prev = r == 0 ? carry : rot;
              carry = rot;
              a[n1] = preemph_sample(a[n1], prev, p.preemph);
            }
#endif
          } else if constexpr (N2 == 8 && PDS_PREEMPH_DPP && PDS_PREEMPH_DPP8) {
            // Eight lanes per frame: a DPP row holds TWO frames (lanes 0-7 and 8-15), so the 16-lane form's
            // "no source lane = no write" does not separate them -- row_shr:1 hands lane 8 the other frame's lane 7,
            // and row_shl:7 (lane 0 <- 7, lane 8 <- 15: the last sample of the row before) also feeds lanes 1-7
            // from lanes 8-14.  The sources are therefore split by a select first: `b` = the row without its last
            // lane of each frame (lane 8 then adds -c * 0), `c` = the row before reduced to those last lanes (lanes
            // 1-7 add -c * 0).  Two selects + two DPP multiply-adds per row instead of a second load + a multiply
            // -add; same product and one rounding per sample: bit-identical.
            float carry = 0.0f;  // the sample in front of the frame (lane 0 of the group; 0 elsewhere)
            if (r == 0) carry = (float)xp[-1];
            const float negc = -p.preemph;
            const bool last = r == N2 - 1;
            inl::static_for<0, NROWS>([&](auto k) {
              constexpr int n1 = NROWS - 1 - decltype(k)::value;
              const float b = last ? 0.0f : a[n1];
              if constexpr (n1 > 0) {
                const float c = last ? a[n1 - 1] : 0.0f;
                asm volatile("s_nop 1\n\t"
                             "v_fmac_f32_dpp %0, %1, %3 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
                             "v_fmac_f32_dpp %0, %2, %3 row_shl:7 row_mask:0xf bank_mask:0xf"
                             : "+v"(a[n1]) : "v"(b), "v"(c), "v"(negc));
              } else {
                asm volatile("s_nop 1\n\tv_fmac_f32_dpp %0, %1, %2 row_shr:1 row_mask:0xf bank_mask:0xf"
                             : "+v"(a[0]) : "v"(b), "v"(negc));
                a[0] = fmaf(negc, carry, a[0]);
              }
            });
          } else {
#pragma unroll
            for (int n1 = 0; n1 < NROWS; ++n1) a[n1] = preemph_sample(a[n1], (float)xp[n1 * N2 - 1], p.preemph);
          }
        }
      } else {
        // (rolled: the lane writes and reads back its own slots, GCH rows per pass through the wave's area)
        float *tmp = wbase;
        inl::static_for<0, (NROWS + G::GCH - 1) / G::GCH>([&](auto cc) {
          constexpr int base = decltype(cc)::value * G::GCH;
          constexpr int top = base + G::GCH < NROWS ? base + G::GCH : NROWS;
#pragma unroll 1
          for (int n1 = base; n1 < top; ++n1) {
            const int idx = n1 * N2 + rho;
            float v = 0.0f;
            if (idx < L) {
              int i = start + idx;
              if (wmode == 1) {
                i = i < 0 ? -1 - i : (i >= n ? 2 * n - 1 - i : i);
              } else {
                i = (int)reflect_index((int64_t)i, (int64_t)n);
              }
              if constexpr (!F64S) {
                v = (float)x[i];
                if (PRE && i > 0) v = preemph_sample(v, (float)x[i - 1], p.preemph);
              } else {
                TIN w = x[i];
                if (PRE && i > 0) w = preemph_sample(w, x[i - 1], (TIN)p.preemph_d);
                v = (float)w;
              }
            }
            tmp[(n1 - base) * 64 + lane] = v;
          }
          wave_sync();
#pragma unroll
          for (int n1 = base; n1 < top; ++n1) a[n1] = tmp[(n1 - base) * 64 + lane];
          wave_sync();
        });
      }
      if (p.include_energy) {
        // compute.py:392-393, on the un-windowed samples of the frame proper
        // Rows below N1/2 lie inside the frame (this kernel requires L > N/2); the others are
        // masked against L.  The limit goes through an opaque asm so that the compares are made
        // here, per item: hoisted out of the loop they would sit in one scalar register pair per
        // row for the whole kernel.
        int lim = L - rho;
        asm volatile("" : "+v"(lim));
#pragma unroll
        for (int n1 = 0; n1 < NROWS; ++n1) {
          const float v = (n1 < N1 / 2 || n1 * N2 < lim) ? a[n1] : 0.0f;
          energy = fmaf(v, v, energy);
        }
      }
      PDS_STAMP(0, 0);  // frame loads issued
      PDS_STAMP(1, 1);  // ... and arrived
      PDS_PHASE(4);
      if constexpr (WINLDS) {
#pragma unroll
        for (int n1 = 0; n1 < NROWS; ++n1) a[n1] = window_mul<N1>(n1, a[n1], wl[n1]);
      } else if constexpr (WINUSE) {
        const float4 *w4 = reinterpret_cast<const float4 *>(win_lds + rho * WSTR);
#pragma unroll
        for (int j = 0; j < (NROWS + 3) / 4; ++j) {
          const float4 w = w4[j];
          const float wv[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
          for (int u = 0; u < 4; ++u)
            if (4 * j + u < NROWS) a[4 * j + u] = window_mul<N1>(4 * j + u, a[4 * j + u], wv[u]);
        }
      } else {
#pragma unroll
        for (int n1 = 0; n1 < NROWS; ++n1) a[n1] = window_mul<N1>(n1, a[n1], win[n1]);
      }
      if constexpr (PF) fetch_next_record();
#pragma unroll
      for (int n1 = NROWS; n1 < N1; ++n1) a[n1] = 0.0f;

      if constexpr (PDS_ABLATE & 2) {
        even_sum = a[0];
        odd_sum = a[1];
#pragma unroll
        for (int k = 0; k < COLS; ++k) {
          Ar[k] = a[2 * k];
          Ai[k] = a[2 * k + 1];
        }
      } else if constexpr (inl::is_pow2(N1)) {
#if PDS_RDFT_DIT
        inl::rdft_dit<N1>(a, even_sum, odd_sum, Ar, Ai);  // (unscaled outputs: unit twiddles, see fast_tables_create)
#else
        inl::rdft_scaled<N1>(a, even_sum, odd_sum, Ar, Ai);
#endif
      } else {
        inl::rdft_direct<N1>(a, even_sum, odd_sum, Ar, Ai);
      }

      // transpose through LDS: row k1 of this frame's block holds column k1 for all n2; row 0
      // holds c[j] = sum_q xw[j + 2 N2 q] (j < 2 N2), the input of the multiples of N1/2.  (Odd
      // N1: c[j] = sum_q xw[j + N2 q] for j < N2 and zeros above, whose 2 N2-point transform has
      // the multiples of N1 at its even outputs; the odd ones are not bins and are dropped.)
      PDS_PHASE(1);
      // LEAN: real parts now, imaginary parts (kept in Ai) behind the column reads of the real ones.  Row 0 holds the
      // packed sums z[m] = c[2 m] + i c[2 m + 1], c = (even sums, odd sums): lane 2 m supplies the real part of
      // elements m and N2 / 2 + m, lane 2 m + 1 their imaginary part; the other lanes write to the row's padding.
      if constexpr (G::LEAN) {
        const int slot = (rho & 1) ? N2 + (rho & 3) : (rho >> 1);
        xf[slot] = even_sum;
        xf[(rho & 1) ? N2 + (rho & 3) : N2 / 2 + (rho >> 1)] = odd_sum;
      } else {
        float *row0 = reinterpret_cast<float *>(exch);
        row0[rho] = even_sum;
        row0[N2 + rho] = odd_sum;
      }
      [[maybe_unused]] float ctr[TWCHAIN ? COLS : 1], cti[TWCHAIN ? COLS : 1];
      if constexpr (TWCHAIN) {
        // (the seeds pass through an opaque asm: the chain is loop invariant, and hoisted out of the item loop it
        // would sit in the thirty registers it is there to free)
#pragma unroll
        for (int j = 0; j < 6; ++j) asm volatile("" : "+v"(sd[j]));
        inl::twiddle_chain<NREG, !PDS_RDFT_DIT>(sd[0], sd[1], sd[2], sd[3], sd[4], sd[5], ctr, cti);
      }
#pragma unroll
      for (int k1 = 1; k1 <= NREG; ++k1) {
        const float wr_ = TWCHAIN ? ctr[TWCHAIN ? k1 : 0] : twr[TWCHAIN ? 0 : k1];
        const float wi_ = TWCHAIN ? cti[TWCHAIN ? k1 : 0] : twi[TWCHAIN ? 0 : k1];
        float2 v;
        v.x = Ar[k1] * wr_ - Ai[k1] * wi_;
        v.y = Ar[k1] * wi_ + Ai[k1] * wr_;
        if constexpr ((PDS_ABLATE & 4) != 0) {
          Ar[k1] = v.x;
          Ai[k1] = v.y;
        } else if (G::TWOPASS && k1 >= N2) {
          // two-pass exchange: the second half of the columns waits, twiddled, for the first half's transforms
          Ar[k1] = v.x;
          Ai[k1] = v.y;
        } else if constexpr (G::LEAN) {
          xf[k1 * G::RSF + rho] = v.x;
          Ai[k1] = v.y;
        } else {
          exch[k1 * RS + rho] = v;
        }
      }
    }
    PDS_STAMP(2, 0);  // window, N1-point transform, twiddles, exchange stores issued
    wave_sync();

    float pw[G::CPL][N2 + 1];
    float sp0 = 0.0f, sp1 = 0.0f;  // the two special bins of lane r <= N2/2
    // PSPLIT (N = 4096 = 64 x 64, one frame per wave): the 32 columns would occupy half the wave and each
    // lane a 64-point transform (256 registers of data).  Instead a column is shared by the lane PAIR
    // (c, c + 32): the lower lane transforms the even-numbered column elements, the upper lane the odd ones
    // (32 points each), `v_permlane32_swap` hands each lane one half of the other's outputs, and the lane
    // finishes the radix-2 step for 16 values of k: Y[k'] = A + W_64^k' B and Y[k' + 32] = A - W_64^k' B,
    // k' = k + 16 h.  Every lane works in this phase, on half the data.
    // (The same for 32 lanes per frame and 16 columns -- N = 1024 as 32 x 32, two frames per wave, lane pairs
    // (c, c + 16) through v_permlane16_swap: the 64 x 16 form of that size holds a 64-point in-lane real
    // transform, 256 registers and 18 KB of LDS per wave; this one 128 and 8.7.)
    constexpr bool PSPLIT = PDS_PAIR_SPLIT && N2 == 2 * COLS && (N2 == 64 || N2 == 32);
    constexpr int PH = N2 / 2, PQ = N2 / 4;  // points of a lane's transform, values of k it finishes
    [[maybe_unused]] const int ph = r / (N2 / 2), pc = r % (N2 / 2);
    if constexpr (PSPLIT) {
      static_assert(!PSPLIT || (G::CPL == 1 && COLS == PH), "pair split: the columns fill half the frame's lanes");
      const float4 *row = reinterpret_cast<const float4 *>(exch + pc * RS);
      float zr[PH], zi[PH], Hr[PH], Hi[PH];
#pragma unroll
      for (int j = 0; j < PH; ++j) {
        const float4 v = row[j];  // elements 2 j and 2 j + 1 of the column
        zr[j] = ph ? v.z : v.x;
        zi[j] = ph ? v.w : v.y;
      }
      PDS_PHASE(5);
      inl::CFFT<PH, 1>::run(zr, zi, Hr, Hi);  // A (lower lane) or B (upper lane)
      PDS_PHASE(2);
      inl::static_for<0, PQ>([&](auto k_) {
        constexpr int k = decltype(k_)::value;
        // (first' = first.lo | second.lo, second' = first.hi | second.hi, lo / hi = the halves of a frame's lane
        // group): lower lane A[k], B[k]; upper lane A[k + PQ], B[k + PQ]
        float ar, br, ai, bi;
        if constexpr (N2 == 64) {
          const auto sr_ = __builtin_amdgcn_permlane32_swap(__float_as_uint(Hr[k]), __float_as_uint(Hr[k + PQ]), false, false);
          const auto si_ = __builtin_amdgcn_permlane32_swap(__float_as_uint(Hi[k]), __float_as_uint(Hi[k + PQ]), false, false);
          ar = __uint_as_float(sr_[0]), br = __uint_as_float(sr_[1]);
          ai = __uint_as_float(si_[0]), bi = __uint_as_float(si_[1]);
        } else {
          const auto sr_ = __builtin_amdgcn_permlane16_swap(__float_as_uint(Hr[k]), __float_as_uint(Hr[k + PQ]), false, false);
          const auto si_ = __builtin_amdgcn_permlane16_swap(__float_as_uint(Hi[k]), __float_as_uint(Hi[k + PQ]), false, false);
          ar = __uint_as_float(sr_[0]), br = __uint_as_float(sr_[1]);
          ai = __uint_as_float(si_[0]), bi = __uint_as_float(si_[1]);
        }
        // t = W_N2^(k + PQ h) B = (-i)^h (W_N2^k B)
        float ur, ui;
        inl::mul_tw<N2, k>(br, bi, ur, ui);
        const float tr = ph ? ui : ur, ti = ph ? -ur : ui;
        const float yar = ar + tr, yai = ai + ti, ybr = ar - tr, ybi = ai - ti;
        pw[0][k] = yar * yar + yai * yai;            // k2 = k + PQ h
        pw[0][PQ + k] = ybr * ybr + ybi * ybi;       // k2 = k + PQ h + PH
        if (pc == 0) {
          // the packed real column (the pair's two lanes): its outputs go to row 0 of the exchange block for
          // the lanes that untangle one bin pair each (both lanes are past reading the row)
          exch[k + PQ * ph] = make_float2(yar, yai);
          exch[k + PQ * ph + PH] = make_float2(ybr, ybi);
        }
      });
#pragma unroll
      for (int k2 = PH; k2 <= N2; ++k2) pw[0][k2] = 0.0f;
      wave_sync();
      {
        const float2 ya = exch[r], yb = exch[(N2 - r) & (N2 - 1)];
        const float ar = ya.x, ai = ya.y, br = yb.x, bi = yb.y;
        const float sr = ar + br, si = ai - bi;
        const float dr = ar - br, di = ai + bi;
        const float tr = sw.x * di + sw.y * dr;
        const float ti = sw.y * di - sw.x * dr;
        const float xr = sr + tr, xi = si + ti;  // 2 X[m]
        const float yr = sr - tr, yi = ti - si;  // 2 X[N2 - m]
        sp0 = 0.25f * (xr * xr + xi * xi);
        sp1 = 0.25f * (yr * yr + yi * yi);
      }
    } else
#pragma unroll
    for (int q = 0; q < G::CPL; ++q) {
      const int kk = q * N2 + r;
      if constexpr (G::TWOPASS) {
        if (q > 0) {
          // (every lane has read its column of the pass before: same wave, in order)
          wave_sync();
#pragma unroll
          for (int k1 = q * N2; k1 < (q + 1) * N2 && k1 <= NREG; ++k1) exch[(k1 - q * N2) * RS + rho] = make_float2(Ar[k1], Ai[k1]);
          wave_sync();
        }
      }
      const float4 *row = reinterpret_cast<const float4 *>(exch + (G::TWOPASS ? r : kk) * RS);
      float zr[N2], zi[N2], Yr[N2], Yi[N2];
      if constexpr (G::LEAN) {
        // the column's real parts; then the imaginary parts take the same rows
        const float4 *rowf = reinterpret_cast<const float4 *>(xf + r * G::RSF);
#pragma unroll
        for (int j = 0; j < N2 / 4; ++j) {
          const float4 v = rowf[j];
          zr[4 * j] = v.x, zr[4 * j + 1] = v.y, zr[4 * j + 2] = v.z, zr[4 * j + 3] = v.w;
        }
        wave_sync();
        {
          const int slot = (rho & 1) ? (rho >> 1) : N2 + (rho & 3);
          xf[slot] = even_sum;
          xf[(rho & 1) ? N2 / 2 + (rho >> 1) : N2 + (rho & 3)] = odd_sum;
        }
#pragma unroll
        for (int k1 = 1; k1 <= NREG; ++k1) xf[k1 * G::RSF + rho] = Ai[k1];
        wave_sync();
#pragma unroll
        for (int j = 0; j < N2 / 4; ++j) {
          const float4 v = rowf[j];
          zi[4 * j] = v.x, zi[4 * j + 1] = v.y, zi[4 * j + 2] = v.z, zi[4 * j + 3] = v.w;
        }
      } else
#pragma unroll
      for (int j = 0; j < N2 / 2; ++j) {
        float4 v;
        if constexpr (PDS_ABLATE & 4) {
          v = make_float4(Ar[(2 * j) % COLS], Ai[(2 * j) % COLS], Ar[(2 * j + 1) % COLS], Ai[(2 * j + 1) % COLS]);
        } else {
          v = row[j];
        }
        zr[2 * j] = v.x;
        zi[2 * j] = v.y;
        zr[2 * j + 1] = v.z;
        zi[2 * j + 1] = v.w;
      }
      // PF: the next item's loads behind this item's column reads (which the transform waits for), spread over
      // the transform's arithmetic
      // (the transform's priority is set in front of them: s_setprio is a scheduling boundary)
      if constexpr (PF && PDS_PF_PLACE == 0) {
        PDS_PHASE(5);
#if PDS_PF_ILV > 0
        __builtin_amdgcn_sched_group_barrier(0x100, N2 / 2, 0);  // the column reads first
#endif
        if (q == 0) issue_prefetch();
      }
      if constexpr (PDS_ABLATE & 8) {
#pragma unroll
        for (int k2 = 0; k2 < N2; ++k2) {
          Yr[k2] = zr[k2];
          Yi[k2] = zi[k2];
        }
      } else {
        if constexpr (!(PF && PDS_PF_PLACE == 0)) PDS_PHASE(5);
        inl::CFFT<N2, 1>::run(zr, zi, Yr, Yi);
        if (q + 1 < G::CPL) {
          PDS_PHASE(1);
        } else {
          PDS_PHASE(2);
        }
      }
      // regular columns; lane 0 (q = 0) holds Y = FFT(c[2m] + i c[2m+1]) of the even/odd sums
#pragma unroll
      for (int k2 = 0; k2 < N2; ++k2) pw[q][k2] = Yr[k2] * Yr[k2] + Yi[k2] * Yi[k2];
      pw[q][N2] = 0.0f;
      if constexpr (!G::DIST) {
        // 8 lanes per frame: lane 0 untangles its short packed-sum transform itself (spreading
        // it would cost registers these geometries do not have)
        if (q == 0 && r == 0)
          inl::rdft_finish_power<2 * N2>(Yr, Yi, [&](auto mm, float re, float im) {
            pw[q][decltype(mm)::value] = re * re + im * im;
          });
      } else if (q == 0) {
        // The bins m * N1/2 need Y[m] and Y[N2 - m] of that lane: lanes 0..N2/2 untangle one bin
        // pair each -- a few moves + 20 flops for every lane instead of a block of ~10 N2
        // instructions that only lane 0 needs.
        float ar, ai, br, bi;
        if constexpr (N2 == 16 && !PDS_PACKED_LDS) {
          // a lane group is one DPP row, so "row_shr:m, keep old where the source lane is outside
          // the row", applied for m = 1, 2, ... in order, leaves lane m with lane 0's register m
          ar = Yr[0], ai = Yi[0], br = Yr[0], bi = Yi[0];
          inl::static_for<1, N2 / 2 + 1>([&](auto mm) {
            constexpr int m = decltype(mm)::value;
            ar = dpp_row_shr_keep<m>(ar, Yr[m]);
            ai = dpp_row_shr_keep<m>(ai, Yi[m]);
            br = dpp_row_shr_keep<m>(br, Yr[N2 - m]);
            bi = dpp_row_shr_keep<m>(bi, Yi[N2 - m]);
          });
        } else {
          // other group sizes: lane 0 hands its column over through row 0 of the frame's exchange
          // block (its only reader, lane 0 itself, is past it)
          // (LEAN: the frame's block is addressed in floats; its first 2 N2 of them take the column)
          float2 *hand = G::LEAN ? reinterpret_cast<float2 *>(xf) : exch;
          if (r == 0) {
            if constexpr (N2 % 2 == 0) {
              float4 *hand4 = reinterpret_cast<float4 *>(hand);
#pragma unroll
              for (int k2 = 0; k2 < N2; k2 += 2) hand4[k2 / 2] = make_float4(Yr[k2], Yi[k2], Yr[k2 + 1], Yi[k2 + 1]);
            } else {
#pragma unroll
              for (int k2 = 0; k2 < N2; ++k2) hand[k2] = make_float2(Yr[k2], Yi[k2]);
            }
          }
          wave_sync();
          const float2 ya = hand[r], yb = hand[(N2 - r) & (N2 - 1)];
          ar = ya.x, ai = ya.y, br = yb.x, bi = yb.y;
        }
        const float sr = ar + br, si = ai - bi;
        const float dr = ar - br, di = ai + bi;
        const float tr = sw.x * di + sw.y * dr;
        const float ti = sw.y * di - sw.x * dr;
        const float xr = sr + tr, xi = si + ti;  // 2 X[m]
        const float yr = sr - tr, yi = ti - si;  // 2 X[N2 - m]
        // (TWCHAIN with rdft_scaled: the window carries the factor 1/2 already)
        sp0 = (TWCHAIN && !PDS_RDFT_DIT ? 1.0f : 0.25f) * (xr * xr + xi * xi);
        sp1 = (TWCHAIN && !PDS_RDFT_DIT ? 1.0f : 0.25f) * (yr * yr + yi * yi);
      }
    }
    if (!use_power) {
#pragma unroll
      for (int q = 0; q < G::CPL; ++q)
#pragma unroll
        for (int k2 = 0; k2 <= N2; ++k2) pw[q][k2] = __builtin_amdgcn_sqrtf(pw[q][k2]);
      sp0 = __builtin_amdgcn_sqrtf(sp0);
      sp1 = __builtin_amdgcn_sqrtf(sp1);
    }
    // row lengths and table offsets of the first USLOTS slots (tables padded to USLOTS entries):
    // fetched here so that the P stores below cover the latency.  (Scalar loads return out of
    // order, so a wait for one is a wait for all: they are issued in two batches per item, this
    // one and the next item's record at the start of the filter phase.)
    Int4 lens4 = {0, 0, 0, 0}, woff4 = {0, 0, 0, 0};
    if constexpr (!RSG && !SEG) {
      lens4 = load_const(reinterpret_cast<const Int4 *>(p.ell_len));
      woff4 = load_const(reinterpret_cast<const Int4 *>(p.ell_woff));
    }
    const int slot_len[USLOTS] = {lens4.x, lens4.y, lens4.z, lens4.w};
    const int slot_woff[USLOTS] = {woff4.x, woff4.y, woff4.z, woff4.w};
    static_assert(USLOTS == 4, "slot tables are fetched as one int4 each");
    // every lane is done with the exchange area (same wave, in order): reuse it as P
    PDS_STAMP(3, 0);  // exchange reads, N2-point transforms, power spectrum
    wave_sync();
    // Stores below avoid lane predicates (each costs exec-mask bookkeeping on the scalar unit):
    // lanes without a value of their own write to a padding slot that is zeroed afterwards.
    // P[frame][bin] (row stride PSTR), or bin-major P[bin][frame] for the row-segment walk
    float *const Pw = RSG ? wbase + g : Pg;           // the lane's frame
    constexpr int PB = RSG ? 4 : 1;                   // floats per bin step
    constexpr int PDUMP = RSG ? NBP : PSTR - 1;       // bin index of the padding slot
    if constexpr (PSPLIT) {
      // the lane's bins: pc + N1 k2 for k2 = k + PQ h, and the mirror images N - pc - N1 (k2 + PH)
      // (the mirror images from their LOWEST address up: one address register and the step in the instruction's
      // offset; counted down from hi_bin the compiler kept an address per step, sixteen registers it had to spill)
      const int lo_bin = pc + N1 * PQ * ph, hi_bin = N / 2 - pc - N1 * PQ * ph;
      float *const Plo = Pw + lo_bin * PB, *const Phi = Pw + (hi_bin - N1 * (PQ - 1)) * PB;
#pragma unroll
      for (int k = 0; k < PQ; ++k) {
        Plo[N1 * k * PB] = pw[0][k];
        Phi[N1 * (PQ - 1 - k) * PB] = pw[0][PQ + k];
      }
    } else
#pragma unroll
    for (int q = 0; q < G::CPL; ++q) {
      const int kk = q * N2 + r;
      if (!G::DIST && q == 0 && r == 0) {
        // output m of the 2 N2-point transform is bin m N1 / 2 (odd N1: even m only)
#pragma unroll
        for (int m = 0; m <= N2; m += (N1 % 2 ? 2 : 1)) Pw[(m * N1 / 2) * PB] = pw[q][m];
      } else {
        // (DIST: lane 0 of q = 0 writes its meaningless column to multiples of N1, all of which
        // the special bins written next overwrite).  Lanes beyond the last column (only when the
        // columns do not fill the lanes) send theirs to the padding slot.
        const bool live = G::FULL || kk < COLS;
        if constexpr (G::FULL && PDS_MIRROR_REBASE) {
          // bins kk + N1 k2 up to N/2, then the mirror images N - kk - N1 k2 -- written from their LOWEST
          // address up, so that both runs are one address register + the step in the instruction's offset
          // (counted down, the compiler keeps one address per step: N2/2 registers)
          float *const Pa = Pw + kk * PB, *const Pm = Pw + (N1 - kk) * PB;
#pragma unroll
          for (int k2 = 0; k2 < N2 / 2; ++k2) Pa[N1 * k2 * PB] = pw[q][k2];
#pragma unroll
          for (int k2 = N2 / 2; k2 < N2; ++k2) Pm[N1 * (N2 - 1 - k2) * PB] = pw[q][k2];
        } else
#pragma unroll
        for (int k2 = 0; k2 < N2; ++k2) {
          // bin kk + N1*k2, or its mirror image when beyond N/2
          const int bin = (k2 < N2 / 2) ? kk + N1 * k2 : N - kk - N1 * k2;
          Pw[(live ? bin : PDUMP) * PB] = pw[q][k2];
        }
      }
    }
    if constexpr (G::DIST) {
      // output m of the 2 N2-point transform is bin m N1 / 2 (odd N1: even m only)
      const bool has = r <= N2 / 2 && (N1 % 2 == 0 || r % 2 == 0);
      Pw[(has ? r * N1 / 2 : PDUMP) * PB] = sp0;
      Pw[(has ? (N2 - r) * N1 / 2 : PDUMP) * PB] = sp1;
    }
    // slots past the last bin are read (with weight 0) by the filter walk: keep them finite
    if constexpr (RSG) {
      constexpr int PADF = (NBP - NB) * 4;  // 4 .. 12 floats behind the last bin
      wbase[NB * 4 + (lane < PADF ? lane : PADF - 1)] = 0.0f;
    } else {
#pragma unroll
      for (int j0 = 0; j0 < PSTR - NB; j0 += N2) {
        const int j = j0 + r;
        Pg[NB + (j < PSTR - NB ? j : PSTR - NB - 1)] = 0.0f;
      }
    }
    // ---- filter bank: lane (g, r) integrates one filter per slot
    // scalar row base + a 32-bit lane offset: no 64-bit vector arithmetic per store
    TOUT *obase = static_cast<TOUT *>(p.out) + (out_row + tb) * p.out_stride;
    const unsigned lane_off = (unsigned)(g * (int)p.out_stride + col0);
    if (p.include_energy) {
      // sum over the frame's lanes with DPP butterflies (lanes 1^, 2^, 7-, 15- within the row);
      // every lane ends up with the total, lane 0 of the frame stores it (compute.py:392-398)
      // (the matrix-pipe front end has summed it already)
      if constexpr (MF == 0) {
        energy = dpp_add<0xB1>(energy);   // quad_perm [1,0,3,2]
        energy = dpp_add<0x4E>(energy);   // quad_perm [2,3,0,1]
        energy = dpp_add<0x141>(energy);  // row_half_mirror
        if constexpr (N2 >= 16) energy = dpp_add<0x140>(energy);  // row_mirror
        if constexpr (N2 >= 32) energy += __shfl_xor(energy, 16, 64);
        if constexpr (N2 == 64) energy += __shfl_xor(energy, 32, 64);
      }
      float e = energy * p.inv_L;
      if (!use_power) e = __builtin_amdgcn_sqrtf(e);
      if (p.use_log) e = fast_log(p.log_floor > e ? p.log_floor : e);
      if (DLT == 0 && valid && r == 0) obase[lane_off - col0] = (TOUT)e;
      if constexpr (DLT > 0) e_keep = e;
      if constexpr (STATS) {
        if (stats_on) {
          const float e4[4] = {__int_as_float(__builtin_amdgcn_readlane(__float_as_int(e), 0)),
                               __int_as_float(__builtin_amdgcn_readlane(__float_as_int(e), 16)),
                               __int_as_float(__builtin_amdgcn_readlane(__float_as_int(e), 32)),
                               __int_as_float(__builtin_amdgcn_readlane(__float_as_int(e), 48))};
          if (lane == 0) stat_add4(0, e4, nfr - tb);
        }
      }
    }
    PDS_STAMP(4, 0);  // P stores, energy
    PDS_PHASE(3);
    int metas[USLOTS] = {0};  // first bin of the lane's row | (filter + 1) << 16, per slot
    if constexpr (!RSG && !SEG) {
#pragma unroll
      for (int sl = 0; sl < USLOTS; ++sl) metas[sl] = meta_lds[sl * N2 + r];
    }
    wave_sync();
    [[maybe_unused]] const int frames_here = nfr - tb;  // frames of this chunk that exist (SEG)
    // this item's record is dead from here on: fetch the next one under the filter phase
    if constexpr (STRETCH) {
      // (the piece's record stays)
    } else if constexpr (PF) {
      b = nb;
      chunk = nchunk;
      nfr = nfr_nx;
      n = n_nx;
      sig_off = sig_off_nx;
      out_row = out_row_nx;
      if constexpr (PDS_PF_PLACE == 1) issue_prefetch();
    } else if (DYN && p.dyn) {
      next_item_dyn(nfr, n, sig_off, out_row);
    } else {
      b = nb;
      chunk = nchunk;
      nfr = (int)load_const(p.nframes + pb);
      n = (int)load_const(p.lengths + pb);
      sig_off = load_const(p.offsets + pb);
      out_row = load_const(p.row_off + pb);
    }

    auto run_slot = [&](const int meta, const int len, const int slot_woff_) {
      const float4 *prow = reinterpret_cast<const float4 *>(Pg + (meta & 0xffff));
      const int woff = slot_woff_ + __mul24(r, len + 4);  // + 4: conflict-free row skew
      float acc0 = 0.0f, acc1 = 0.0f, acc2 = 0.0f, acc3 = 0.0f;
      const float4 *wrow = reinterpret_cast<const float4 *>((ELL_LDS ? ellw_lds : p.ell_w) + woff);
      // 8 bins per step, two steps per pass of the loop (8 LDS reads in flight per 16
      // multiply-adds); the trip count is a shift of the row length, which is a multiple of 8
      auto step = [&]() {
        // weights and powers of one half, then the other: the first multiply-adds wait for two
        // reads instead of five (LDS returns in order; +3 % on the dense gammatone bank)
        const float4 w0 = wrow[0];
        const float4 p0 = prow[0];
        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
        const float4 w1 = wrow[1];
        const float4 p1 = prow[1];
        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
        acc0 = fmaf(w0.x, p0.x, acc0);
        acc1 = fmaf(w0.y, p0.y, acc1);
        acc2 = fmaf(w0.z, p0.z, acc2);
        acc3 = fmaf(w0.w, p0.w, acc3);
        acc0 = fmaf(w1.x, p1.x, acc0);
        acc1 = fmaf(w1.y, p1.y, acc1);
        acc2 = fmaf(w1.z, p1.z, acc2);
        acc3 = fmaf(w1.w, p1.w, acc3);
        wrow += 2;
        prow += 2;
      };
      const unsigned steps = (PDS_ABLATE & 32) ? 1u : (unsigned)len >> 3;
      if constexpr (N >= PDS_ELL_DEEP_N) {
        // (256-register geometries: four steps per pass, 16 reads in flight)
#pragma unroll 1
        for (unsigned i = steps >> 2; i != 0; --i) {
          step();
          step();
          step();
          step();
        }
        if (steps & 2u) {
          step();
          step();
        }
      } else {
#pragma unroll 1
        for (unsigned i = steps >> 1; i != 0; --i) {
          step();
          step();
        }
      }
      if (steps & 1u) step();
      float acc = (acc0 + acc1) + (acc2 + acc3);
      // max(val, floor) as Python evaluates it: a NaN stays a NaN (compute.py:459)
      if (p.use_log) acc = fast_log(p.log_floor > acc ? p.log_floor : acc);
      if constexpr (PDS_ABLATE & 64) {
        keep_alive(acc);
      } else {
        // filter index f = (meta >> 16) - 1; lanes without a filter in this slot have meta < 2^16
        const unsigned byte_off = (lane_off + (unsigned)(meta >> 16) - 1u) * (unsigned)sizeof(TOUT);
        if (valid && meta >= 0x10000)
          *reinterpret_cast<TOUT *>(reinterpret_cast<char *>(obase) + byte_off) = (TOUT)acc;
      }
    };
    if constexpr (RSG) {
      // Row-segment walk (rseg_tables.h): lane = one segment of seg_len bins of one filter, for all
      // four frames; weights [round][seg_len / 4][lane] float4, powers bin-major.
      const float4 *P4 = reinterpret_cast<const float4 *>(wbase);
      const int t4n = p.seg_len >> 2;
      auto round_body = [&](const int rd, float (&logged)[4]) {
        const int meta = meta_lds[rd * 64 + lane];
        const float4 *prow = P4 + (meta & 0x3fff);
        const float4 *wrow = reinterpret_cast<const float4 *>(ellw_lds) + __mul24(rd, t4n) * 64 + lane;
        float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f, a3 = 0.0f;
        auto quad = [&](int t4) {
          const float4 w = wrow[t4 * 64];
          const float4 x0 = prow[4 * t4], x1 = prow[4 * t4 + 1], x2 = prow[4 * t4 + 2], x3 = prow[4 * t4 + 3];
          a0 = fmaf(w.x, x0.x, a0);
          a1 = fmaf(w.x, x0.y, a1);
          a2 = fmaf(w.x, x0.z, a2);
          a3 = fmaf(w.x, x0.w, a3);
          a0 = fmaf(w.y, x1.x, a0);
          a1 = fmaf(w.y, x1.y, a1);
          a2 = fmaf(w.y, x1.z, a2);
          a3 = fmaf(w.y, x1.w, a3);
          a0 = fmaf(w.z, x2.x, a0);
          a1 = fmaf(w.z, x2.y, a1);
          a2 = fmaf(w.z, x2.z, a2);
          a3 = fmaf(w.z, x2.w, a3);
          a0 = fmaf(w.w, x3.x, a0);
          a1 = fmaf(w.w, x3.y, a1);
          a2 = fmaf(w.w, x3.z, a2);
          a3 = fmaf(w.w, x3.w, a3);
        };
        // the common segment lengths run unrolled, every read of the segment in flight at once
        if (t4n == 3) {
          quad(0);
          quad(1);
          quad(2);
        } else if (t4n == 1) {
          quad(0);
        } else if (t4n == 2) {
          quad(0);
          quad(1);
        } else {
#pragma unroll 2
          for (int t4 = 0; t4 < t4n; ++t4) quad(t4);
        }
#if PDS_PF_WINAT == 0
        read_window();  // (WINLDS: the next item's window slice, behind this round's reads)
#endif
        // add up a filter's segments (at most four, on consecutive lanes of one DPP row): lane i takes
        // lane i + 1's sums where the table says the run continues, then lane i + 2's
        const float m1 = (meta & (1 << 14)) ? 1.0f : 0.0f, m2 = (meta & (1 << 15)) ? 1.0f : 0.0f;
        auto shl = [](float v, auto ctrl) {
          return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), decltype(ctrl)::value, 0xf, 0xf, true));
        };
#if PDS_WALK_DPP_FMAC
        // the neighbour's sums as DPP operands of the multiply-adds (the compiler keeps a v_mov_b32_dpp per operand: eight
        // instructions more per round); lanes without a source add 0 (bound_ctrl); a DPP operand must not be read within
        // two instructions of its last vector write: one s_nop in front, the rest are four instructions apart
        (void)shl;
        asm volatile("s_nop 1\n\t"
                     "v_fmac_f32_dpp %0, %0, %4 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
                     "v_fmac_f32_dpp %1, %1, %4 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
                     "v_fmac_f32_dpp %2, %2, %4 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
                     "v_fmac_f32_dpp %3, %3, %4 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
                     "v_fmac_f32_dpp %0, %0, %5 row_shl:2 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
                     "v_fmac_f32_dpp %1, %1, %5 row_shl:2 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
                     "v_fmac_f32_dpp %2, %2, %5 row_shl:2 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
                     "v_fmac_f32_dpp %3, %3, %5 row_shl:2 row_mask:0xf bank_mask:0xf bound_ctrl:0"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m1), "v"(m2));
#else
        a0 = fmaf(shl(a0, inl::Int<0x101>{}), m1, a0);
        a1 = fmaf(shl(a1, inl::Int<0x101>{}), m1, a1);
        a2 = fmaf(shl(a2, inl::Int<0x101>{}), m1, a2);
        a3 = fmaf(shl(a3, inl::Int<0x101>{}), m1, a3);
        a0 = fmaf(shl(a0, inl::Int<0x102>{}), m2, a0);
        a1 = fmaf(shl(a1, inl::Int<0x102>{}), m2, a1);
        a2 = fmaf(shl(a2, inl::Int<0x102>{}), m2, a2);
        a3 = fmaf(shl(a3, inl::Int<0x102>{}), m2, a3);
#endif
        const int f = (meta >> 16) - 1;  // the filter, on the first lane of its run; -1 elsewhere
        const float vals[4] = {a0, a1, a2, a3};
        TOUT *dst = obase + col0 + (f < 0 ? 0 : f);
#pragma unroll
        for (int gg = 0; gg < 4; ++gg) {
          float v = vals[gg];
          // max(val, floor) as Python evaluates it: a NaN stays a NaN (compute.py:459)
          if (p.use_log) v = fast_log(p.log_floor > v ? p.log_floor : v);
          if (DLT == 0 && f >= 0 && gg < frames_here) dst[(int64_t)gg * p.out_stride] = (TOUT)v;
          logged[gg] = v;
        }
        if constexpr (STATS) {
          if (stats_on && f >= 0) stat_add4(col0 + f, logged, frames_here);
        }
      };
      if constexpr (DLT == 0) {
        float unused[4];
        for (int rd = 0; rd < p.seg_rounds; ++rd) round_body(rd, unused);
#if PDS_PF_WINAT == 1
        read_window();  // (WINLDS: the next item's window slice)
#endif
      } else {
        // the chunk's logged coefficients: the lane's filter of every round, the energies on their lane
        float cur[DR][4];
#pragma unroll
        for (int rd = 0; rd < DR; ++rd) {
#pragma unroll
          for (int q = 0; q < 4; ++q) cur[rd][q] = 0.0f;
          if (rd < p.seg_rounds) round_body(rd, cur[rd]);
          if (p.dl_eslot == rd * 64 + lane) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
              cur[rd][q] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(e_keep), 16 * q));
          }
        }
        if (chunk == 0) {  // frames in front of the utterance repeat its first frame
#pragma unroll
          for (int rd = 0; rd < DR; ++rd)
#pragma unroll
            for (int q = 0; q < 4; ++q) Wc[rd][q] = cur[rd][0];
        }
        if (chunk > e_lo) emit_rows(chunk - 1, cur);
#pragma unroll
        for (int rd = 0; rd < DR; ++rd)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            Wp[rd][q] = Wc[rd][q];
            Wc[rd][q] = cur[rd][q];
          }
        ++chunk;
      }
      PDS_STAMP(5, 0);  // filter walk
      wave_sync();
      continue;
    }
    if constexpr (SEG) {
      static_assert(!SEG || ELL_LDS, "segmented walks: tables in LDS");
      static_assert(SEG != 2 || G::GROUPS == 4, "matrix-pipe segment walk: four frames per wave");
      {
        // Segmented walk for dense banks.  The rows of all filters are cut into segments of
        // seg_len bins and dealt to the 64 lanes, seg_rounds segments each; a lane reads a
        // segment's weights ONCE and applies them to all four frames of the wave (one 16-byte
        // weight read per four 16-byte power reads instead of one per one, and no lane waits for
        // a longer row than its own: every segment has the same length).  Partial sums go to the
        // part of the wave's area that P leaves free; then lane f adds up filter f's segments for
        // the four frames and stores four coefficients (64 consecutive floats per store).
        float4 *part = reinterpret_cast<float4 *>(wbase + G::GROUPS * PSTR);
        constexpr int SEG_DEPTH = N >= PDS_SEG_DEEP_N ? 4 : 2;
        const int steps = p.seg_len >> 2;
        if constexpr (SEG == 2) {
          // Matrix-pipe form (mseg_tables.h): the filters in quads of four neighbours, a quad's bin range
          // in units of seg_len bins, 16 units per round -- one per block of v_mfma_f32_4x4x1_16B_f32.
          // Lane 4 b + i supplies the weight of filter i of block b's quad, lane 4 b + j the power of
          // frame j, and lane 4 b + j receives the block's four filter sums for frame j: 256 exact
          // float32 multiply-adds per instruction on the pipe the kernel leaves idle otherwise, for one
          // 16-byte weight read and one 16-byte power read per four of them (the segmented walk below:
          // five reads and sixteen vector multiply-adds per four bins of ONE filter).  Two accumulators
          // take the bins alternately, so an instruction does not wait for the one before.
          const int blk = lane >> 2, fr = lane & 3;
          const float *Pj = wbase + fr * PSTR;
          float *pf = reinterpret_cast<float *>(part) + fr;
          // A round: ALL its operand reads (two 16-byte reads per four bins), then its matrix instructions;
          // the reads of the NEXT round are issued before this round's instructions (two operand buffers),
          // and a round's first bin is fetched two rounds ahead -- so neither an LDS round trip nor the
          // table look-up in front of it is exposed per round.  (Measured at 1.5 waves per SIMD, Gammatone-64
          // at N = 1024: reads and instructions of one round back to back take ~950 cycles per round for
          // ~260 cycles of matrix pipe.)
          const float4 *wbase4 = reinterpret_cast<const float4 *>(ellw_lds) + lane;
          const int last = p.seg_rounds - 1;
          // (a round's table entry: first bin | flush << 15 | partial slot << 16 -- the block's sums stay in the
          // instruction's accumulators across rounds and leave for their slot where the block's next unit belongs to
          // another quad of filters, or the block ends: mseg_tables.h)
          auto meta_of = [&](int rd) { return meta_lds[(rd < last ? rd : last) * 16 + blk]; };
          auto walk = [&](auto steps_c) {
            constexpr int ST = decltype(steps_c)::value;
            f32x4 acc0 = {0.0f, 0.0f, 0.0f, 0.0f}, acc1 = {0.0f, 0.0f, 0.0f, 0.0f};
            auto fetch = [&](float4 (&w)[ST], float4 (&x)[ST], const int rd, const int meta) {
              const float4 *wrow = wbase4 + __mul24(rd, ST) * 64;
              const float4 *prow = reinterpret_cast<const float4 *>(Pj + (meta & 0x7fff));
#pragma unroll
              for (int u = 0; u < ST; ++u) {
                w[u] = wrow[u * 64];
                x[u] = prow[u];
              }
            };
            auto compute = [&](const float4 (&w)[ST], const float4 (&x)[ST], const int meta) {
#pragma unroll
              for (int u = 0; u < ST; ++u) {
                acc0 = __builtin_amdgcn_mfma_f32_4x4x1f32(w[u].x, x[u].x, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_4x4x1f32(w[u].y, x[u].y, acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_4x4x1f32(w[u].z, x[u].z, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_4x4x1f32(w[u].w, x[u].w, acc1, 0, 0, 0);
              }
              if (meta & 0x8000) {
                // partial slot: [filter of the quad], a float4 over the frames
                float *dst = pf + (meta >> 16) * 16;
                dst[0] = acc0[0] + acc1[0];
                dst[4] = acc0[1] + acc1[1];
                dst[8] = acc0[2] + acc1[2];
                dst[12] = acc0[3] + acc1[3];
                acc0 = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
                acc1 = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
              }
            };
            // (two buffers of 16 bins: 64 registers, which the 256-register geometries have; the others,
            // and longer units, run one round at a time)
            if constexpr (ST == 4 && MINW <= 2) {
              float4 wa[ST], xa[ST], wb[ST], xb[ST];
              int ma = meta_of(0), mb = meta_of(1), mc = meta_of(2);
              fetch(wa, xa, 0, ma);
#pragma unroll 1
              for (int rd = 0; rd <= last; rd += 2) {
                const int md = meta_of(rd + 3), me = meta_of(rd + 4);
                if (rd < last) fetch(wb, xb, rd + 1, mb);
                compute(wa, xa, ma);
                if (rd >= last) break;
                if (rd + 1 < last) fetch(wa, xa, rd + 2, mc);
                compute(wb, xb, mb);
                ma = mc;
                mb = md;
                mc = me;
              }
            } else {
              int m = meta_of(0);
#pragma unroll 1
              for (int rd = 0; rd <= last; ++rd) {
                const int next = meta_of(rd + 1);
                float4 w[ST], x[ST];
                fetch(w, x, rd, m);
                compute(w, x, m);
                m = next;
              }
            }
          };
          if (steps == 4) {  // (mseg_tables.h builds units of 16 or 32 bins)
            walk(inl::Int<4>{});
          } else {
            walk(inl::Int<8>{});
          }
        } else
        for (int q = 0; q < p.seg_rounds; ++q) {
          const int slot = q * 64 + lane;
          const int first = meta_lds[slot];
          const float4 *wrow = reinterpret_cast<const float4 *>(ellw_lds + __mul24(slot, p.seg_len + 4));
          const float4 *pg[G::GROUPS];
          float acc[G::GROUPS];
#pragma unroll
          for (int gg = 0; gg < G::GROUPS; ++gg) {
            pg[gg] = reinterpret_cast<const float4 *>(wbase + gg * PSTR + first);
            acc[gg] = 0.0f;
          }
          // (geometries with the 256-register budget keep twice as many reads in flight)
#pragma unroll SEG_DEPTH
          for (int i = 0; i < steps; ++i) {
            const float4 w = wrow[i];
            float4 x[G::GROUPS];
#pragma unroll
            for (int gg = 0; gg < G::GROUPS; ++gg) x[gg] = pg[gg][i];
#pragma unroll
            for (int gg = 0; gg < G::GROUPS; ++gg) acc[gg] = fmaf(w.x, x[gg].x, acc[gg]);
#pragma unroll
            for (int gg = 0; gg < G::GROUPS; ++gg) acc[gg] = fmaf(w.y, x[gg].y, acc[gg]);
#pragma unroll
            for (int gg = 0; gg < G::GROUPS; ++gg) acc[gg] = fmaf(w.z, x[gg].z, acc[gg]);
#pragma unroll
            for (int gg = 0; gg < G::GROUPS; ++gg) acc[gg] = fmaf(w.w, x[gg].w, acc[gg]);
          }
          float *dst = reinterpret_cast<float *>(part) + slot * G::GROUPS;
#pragma unroll
          for (int gg = 0; gg < G::GROUPS; ++gg) dst[gg] = acc[gg];
        }
        if constexpr (PDS_STAMPS > 2) PDS_STAMP(4, 0);  // (diagnostic: the rounds go to slot 4, the sums and stores stay in 5)
        wave_sync();
        const int *fmeta = meta_lds + p.seg_rounds * (SEG == 2 ? 16 : 64);
        constexpr int PSTEP = SEG == 2 ? 4 : 1;  // a filter's partial sums: consecutive slots, or a quad apart
        for (int f = lane; f < p.num_filts; f += 64) {
          const int fm = fmeta[f];
          const float *src = reinterpret_cast<const float *>(part) + (fm & 0xffff) * G::GROUPS;
          float sum[G::GROUPS];
#pragma unroll
          for (int gg = 0; gg < G::GROUPS; ++gg) sum[gg] = 0.0f;
          if constexpr (PDS_MSEG_RED4) {
            // (long rows have ten and more partial sums: four reads in flight per pass instead of a round
            // trip per partial; reads past the filter's last partial re-read it and are masked out)
            const int cnt = fm >> 16;
            for (int k = 0; k < cnt; k += 4) {
              float v[4][G::GROUPS];
#pragma unroll
              for (int u = 0; u < 4; ++u) {
                const float *at = src + G::GROUPS * PSTEP * (k + u < cnt ? k + u : cnt - 1);
                if constexpr (G::GROUPS == 4) {
                  const float4 t = *reinterpret_cast<const float4 *>(at);
                  v[u][0] = t.x, v[u][1] = t.y, v[u][2] = t.z, v[u][3] = t.w;
                } else if constexpr (G::GROUPS == 2) {
                  const float2 t = *reinterpret_cast<const float2 *>(at);
                  v[u][0] = t.x, v[u][1] = t.y;
                } else {
#pragma unroll
                  for (int gg = 0; gg < G::GROUPS; ++gg) v[u][gg] = at[gg];
                }
              }
#pragma unroll
              for (int u = 0; u < 4; ++u) {
                const bool live = k + u < cnt;
#pragma unroll
                for (int gg = 0; gg < G::GROUPS; ++gg) sum[gg] += live ? v[u][gg] : 0.0f;
              }
            }
          } else
          for (int k = fm >> 16; k > 0; --k) {
#pragma unroll
            for (int gg = 0; gg < G::GROUPS; ++gg) sum[gg] += src[gg];
            src += PSTEP * G::GROUPS;
          }
#pragma unroll
          for (int gg = 0; gg < G::GROUPS; ++gg) {
            float v = sum[gg];
            // max(val, floor) as Python evaluates it: a NaN stays a NaN (compute.py:459)
            if (p.use_log) v = fast_log(p.log_floor > v ? p.log_floor : v);
            if (gg < frames_here) obase[(int64_t)gg * p.out_stride + col0 + f] = (TOUT)v;
            sum[gg] = v;
          }
          if constexpr (STATS) {
            if (stats_on) stat_add4(col0 + f, sum, frames_here);
          }
        }
        PDS_STAMP(5, 0);  // filter walk
        wave_sync();
        continue;
      }
    }
#pragma unroll
    for (int sl = 0; sl < USLOTS; ++sl)
      if (sl < p.ell_slots) run_slot(metas[sl], slot_len[sl], slot_woff[sl]);
    for (int sl = USLOTS; sl < p.ell_slots; ++sl)
      run_slot(meta_lds[sl * N2 + r], load_const(p.ell_len + sl), load_const(p.ell_woff + sl));
    PDS_STAMP(5, 0);  // filter walk
    wave_sync();
  }
  if constexpr (PDS_STAMPS) {
    if (p.stamps && lane == 0) {
      // [0..5] phase sums, [6] items, [7] bookkeeping, then absolute times: kernel entry, loop start, loop end
      unsigned long long *dst = p.stamps + ((size_t)blockIdx.x * p.waves + wave) * 12;
#pragma unroll
      for (int i = 0; i < 8; ++i) dst[i] = st_acc[i];
      dst[8] = st_entry;
      dst[9] = st_loop;
      dst[10] = __builtin_readcyclecounter();
    }
  }
}

}  // namespace pds
