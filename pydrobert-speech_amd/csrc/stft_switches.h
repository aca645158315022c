// Compile-time switches of the fused STFT kernel (stft_wave_kernel.h): scheduling hints, the forms of the lean
// geometries, and the measurement / experiment builds (tools/build_variant.sh passes -D...).  The product build
// defines none of them on the command line: the defaults below ARE the product.  Measured-and-rejected kernel forms
// compile only with -DPDS_EXPERIMENTS=1 (or -DPDS_DYN=1); profiles/HISTORY.md has their numbers.
#pragma once
namespace pds {

// Timing-only builds (tools/build_variant.sh -DPDS_ABLATE=mask) drop one stage each to see what it
// costs; results are wrong by construction.  0 in the product build.
// Wave priority per phase of an item (s_setprio): the four waves of a SIMD run the same sequence
// of phases, and with equal priorities the issue arbiter lets them drift into the same phase, where
// they queue for one unit (VALU in the transforms, LDS in the exchange and the filter walk) while
// the other idles.  Raising the priority as an item ages ("oldest first": loads and the in-lane
// real DFT 0, exchange and N2-point FFT 1, power spectrum + P stores 2, filter walk 3) keeps the
// waves staggered: +10 % on the headline workload, +5 ... 10 % on every other geometry with more
// than one wave per SIMD (tools/ab_libs.sh; flat priorities for the memory phases alone: +7 %,
// youngest first: +3 %).  One hex digit per phase in PDS_PRIO_PACK, from the lowest digit:
// 0 record + sample loads, 1 LDS exchange, 2 power spectrum + P stores, 3 filter walk,
// 4 window + in-lane real DFT, 5 N2-point FFT.  Negative: no hints (tools/build_variant.sh).
// (Round 2, with the row-segment walk: record + sample loads at priority 2 instead of 0 -- a wave gets its 25
// loads out at once and waits for them, instead of queueing for issue slots first -- headline +2.5 % on two
// boxes, Gabor-64 +2.3 %, the other geometries +-0: 0x103212.)
#ifndef PDS_PRIO_PACK
#define PDS_PRIO_PACK 0x103212
#endif
#if PDS_PRIO_PACK == 0xffffff  // (experiment: scheduling barriers at the phase boundaries, no priorities)
#define PDS_PHASE(i) __builtin_amdgcn_sched_barrier(0)
#else
#define PDS_PHASE(i) do { if (PDS_PRIO_PACK >= 0) __builtin_amdgcn_s_setprio(((PDS_PRIO_PACK) >> (4 * (i))) & 3); } while (0)
#endif
#ifndef PDS_ABLATE
#define PDS_ABLATE 0
#endif
// Diagnostic builds (tools/build_variant.sh -DPDS_STAMPS=1, tools/phase_stamps.py): every wave adds up
// the shader-clock time it spends in each phase of an item (s_memtime at the phase boundaries, with
// the loads drained where a phase ends at their arrival) and leaves the sums in a buffer set through
// pds_debug_set_stamp_buffer.  The stamps themselves cost ~10 %; 0 in the product build.
#ifndef PDS_STAMPS
#define PDS_STAMPS 0
#endif
#if PDS_STAMPS
extern unsigned long long *g_stamp_buf;  // (stft_fast.hip: pds_debug_set_stamp_buffer)
// (PDS_STAMPS=2: a wave's entry, loop start, loop end and item count only -- no stamps inside the loop, so the
// build runs like the product; tools/wave_spread.py reads how evenly the waves finish)
#define PDS_STAMP(i, drain)                                               \
  do {                                                                    \
    if (PDS_STAMPS == 2) break;                                           \
    __builtin_amdgcn_sched_barrier(0);                                    \
    if (drain) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           \
    const unsigned long long now_ = __builtin_readcyclecounter();         \
    st_acc[i] += now_ - st_prev;                                          \
    st_prev = now_;                                                       \
    __builtin_amdgcn_sched_barrier(0);                                    \
  } while (0)
#else
#define PDS_STAMP(i, drain) do { } while (0)
#endif
// filter-walk depth of the geometries that run two waves per SIMD or fewer (256 registers and more)
// (measured, tools/ab_libs.sh: segmented walk four steps deep at N = 1024, Gammatone-64 +3 %; ELL walk four
// steps deep: N = 4096 +0.6 %, N = 2048 -0.5 %)
#ifndef PDS_SEG_DEEP_N
#define PDS_SEG_DEEP_N 1024
#endif
#ifndef PDS_ELL_DEEP_N
#define PDS_ELL_DEEP_N 4096
#endif
// transform sizes from which dense banks take the matrix-pipe segment walk by default
#ifndef PDS_MSEG_MIN_N
#define PDS_MSEG_MIN_N 1024
#endif
#ifndef PDS_PAIR_SPLIT  // (N = 4096: a column's 64-point transform over a lane pair; 0: the lower half of the wave alone)
#define PDS_PAIR_SPLIT 1
#endif
#ifndef PDS_N4096_MINW  // (waves per SIMD the 38-row N = 4096 instantiation is built for: with the pair split it needs 270 registers, i.e. 15 spilled dwords at two waves per SIMD, measured +19 % over one)
#define PDS_N4096_MINW 2
#endif
#ifndef PDS_FAST_PROLOGUE  // (tables and wave areas set up 16 bytes at a time)
#define PDS_FAST_PROLOGUE 1
#endif
#ifndef PDS_MSEG_RED4  // (experiment: the partial sums of the matrix-pipe walk read four at a time)
#define PDS_MSEG_RED4 1
#endif
#ifndef PDS_FILTER_UNROLL
#define PDS_FILTER_UNROLL 2
#endif
// PF instantiations (the next item's samples prefetched into registers, see the kernel):
// PDS_PF_TW   inter-stage twiddles: 0 thirty registers (as without PF), 1 regenerated per item from three seeds
// PDS_PF_WIN  window slice: 0 registers, 1 re-read from an LDS table per item (16-byte reads)
// PDS_PF_PLACE where the prefetch loads are issued: 0 behind the exchange (in front of the column transforms),
//             1 in front of the filter walk
// PDS_PF_ILV  vector instructions the scheduler is asked to put between two prefetch loads (0: its own choice)
// Round-3 experiments, measured and NOT in the product build (profiles/r3a_*.txt):
// -DPDS_EXPERIMENTS=1 builds the prefetch instantiation (PF) and its tables, -DPDS_DYN=1 the dynamic item
// distribution inside a workgroup (DYN); tools/ab_pf.sh and tools/ab_dyn.sh run the comparisons.
#ifndef PDS_EXPERIMENTS
#define PDS_EXPERIMENTS 0
#endif
// fused pre-emphasis of the 16-lane geometries: 1 = the predecessor sample as a DPP operand of the multiply-add
// (two per row), 0 = mov_dpp + select + multiply-add (three per row; rounds 1-3a)
// in-lane real transform of the power-of-two geometries: 1 = decimation in time on real data (inl::rdft_dit, unscaled
// outputs, unit twiddles), 0 = complex transform of half the size + untangling (inl::rdft_scaled; rounds 1-3a, and
// what the matrix-pipe front end of the experiments build produces)
#ifndef PDS_RDFT_DIT
#define PDS_RDFT_DIT (!PDS_EXPERIMENTS)
#endif
// 1: window multiplies of the rows that lie wholly inside a frame as ordinary multiplies, which the compiler contracts
// with the transform's first additions (9 vector instructions fewer per item at N = 512 -- and the same time, +-0.7 %,
// profiles/r3r_window_contract_ab.txt: a three-register v_fma_f32 costs 1.3 nJ where the multiply and the add it replaces cost 0.9
// each, profiles/r3p_energy_microbench.txt); 0 (product): v_mul_legacy_f32 for every row
#ifndef PDS_WINDOW_CONTRACT
#define PDS_WINDOW_CONTRACT 0
#endif
// packed real column of the 16-lane geometries handed from lane 0 to the lanes that untangle it: 0 = 32 DPP moves,
// 1 = through the wave's LDS area (eight 16-byte writes of lane 0, two 8-byte reads per lane)
#ifndef PDS_PACKED_LDS
#define PDS_PACKED_LDS 0
#endif
// row-segment walk: a filter's segments added up with the neighbour's sums as DPP operands of the multiply-adds (1) or
// through v_mov_b32_dpp + multiply-add (0: rounds 2-3a)
#ifndef PDS_WALK_DPP_FMAC
#define PDS_WALK_DPP_FMAC 1
#endif
#ifndef PDS_PREEMPH_DPP
#define PDS_PREEMPH_DPP 1
#endif
#ifndef PDS_DYN
#define PDS_DYN 0
#endif
#ifndef PDS_PF_TW
#define PDS_PF_TW 1
#endif
// Every kernel of the 64 x 16 geometry (N = 1024) regenerates its twiddles and reads its window slice from LDS: 146 - 168
// VGPRs instead of 220 - 243, three waves per SIMD (stft_geoms.def) -- this geometry is bound by its waves' latencies,
// not by the vector pipe.  -DPDS_LEAN_1024=0 with the geometry's MINW back at 2: the form of round 2.
#ifndef PDS_LEAN_1024
#define PDS_LEAN_1024 1
#endif
#ifndef PDS_DLT_CHAIN  // (experiment: regenerated twiddles in the one-launch statics + deltas kernel, float32 samples too)
#define PDS_DLT_CHAIN 0
#endif
#ifndef PDS_PF_WIN
#define PDS_PF_WIN 0
#endif
#ifndef PDS_PF_PLACE
#define PDS_PF_PLACE 1
#endif
#ifndef PDS_PF_WINAT  // (WINLDS: the window slice is re-read 0 inside a walk round, behind its reads, 1 behind the rounds)
#define PDS_PF_WINAT 1
#endif
#ifndef PDS_PF_ILV
#define PDS_PF_ILV 0
#endif
}  // namespace pds

// Mirror-image bins of the power spectrum stored from their lowest address up (one address register + immediate
// offsets) instead of counted down from N - k (one hoisted address register per step).
#ifndef PDS_MIRROR_REBASE
#define PDS_MIRROR_REBASE 1
#endif

// float64 samples WITH fused pre-emphasis on the geometries without the 16-byte pair loads (8 / 32 / 64 lanes per
// frame): 8-byte values loaded per batch (a row takes two: predecessor and sample); 0: row by row, as the compiler
// schedules it (which is what the kernels without pre-emphasis keep).
#ifndef PDS_F64_ROW_BATCH
#define PDS_F64_ROW_BATCH 12
#endif

// Fused pre-emphasis of float32 / int16 samples on the 8-lane geometries (N = 128 / 256: 8 kHz audio) with DPP
// operands too (two selects + two multiply-adds per row instead of a second load); 0: predecessors are loaded.
#ifndef PDS_PREEMPH_DPP8
#define PDS_PREEMPH_DPP8 1
#endif
