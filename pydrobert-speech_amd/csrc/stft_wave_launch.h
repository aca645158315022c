// Host side of the fused STFT kernel: launch_wave<N1, N2, ROWS, MINW> picks the instantiation of
// stft_wave_kernel.h for a call (filter walk, sample and feature types, pre-emphasis, statics + deltas, stretch
// scheduling), sizes its workgroups and LDS, and launches it.  Included by stft_geom.hip -- compiled once per geometry
// of stft_geoms.def -- and by stft_fast.hip (tables, dispatch over the geometries).
#pragma once
#include "stft_wave_kernel.h"

namespace pds {

// ----------------------------------------------------------------------- host side ---

// transform sizes with float64-input instantiations of the fused kernel
constexpr bool fast_f64in_kind(int n) { return n == 256 || n == 512 || n == 1024 || n == 2048; }

// transform sizes with fused statics + deltas instantiations (16-lane geometries)
constexpr bool fast_deltas_kind(int n) { return n == 512 || n == 1024; }


template <int N1, int N2, int NROWS, int MINW>
int32_t launch_wave(const pds_stft_plan *plan, const BatchArgs &a) {
  using G = WaveGeom<N1, N2, NROWS>;
  // resident waves per CU that the register budget allows; the fused statics + deltas instantiations
  // hold their window in registers and run three waves per SIMD where the others run four
#ifndef PDS_DLT_MINW  // (experiment: 4 = the one-launch statics + deltas kernel at four waves per SIMD, twiddles regenerated)
#define PDS_DLT_MINW 3
#endif
  // (N = 1024: the one-launch statics + deltas kernel keeps two waves per SIMD -- at three it spills 25 - 33 registers)
  constexpr int DMINW = (N1 == 64 && N2 == 16) ? 2 : MINW > PDS_DLT_MINW ? PDS_DLT_MINW : MINW;
  constexpr int CU_WAVES_STFT = 4 * MINW;
  // (N = 1024: the float64-sample kernels keep two waves per SIMD -- their pair loads hold 120 registers in flight)
  constexpr int F64_MINW = (N1 == 64 && N2 == 16) ? 2 : MINW, F64_WAVES = 4 * F64_MINW;
  const int CU_WAVES = a.dl_K > 0 ? 4 * DMINW : a.in_f64 ? F64_WAVES : CU_WAVES_STFT;
  const FastTables &ft = plan->fast;
  FastParams p;
  p.sig = a.d_signal;
  p.offsets = a.d_offsets;
  p.lengths = a.d_lengths;
  p.nframes = a.d_nframes;
  p.row_off = a.d_row_off;
  p.out = a.d_out;
  p.out_stride = a.out_stride;
  p.win_lane = ft.d_window;
  p.tw_lane = (const float2 *)ft.d_twiddle;
  p.tw_special = (const float2 *)ft.d_tw_special;
  constexpr bool LEAN_ALL = G::LEAN || (PDS_LEAN_1024 && N1 == 64 && N2 == 16);  // every instantiation regenerates its twiddles
  p.win_half = LEAN_ALL ? ft.d_win_half : nullptr;
  p.tw_seed = LEAN_ALL ? (const float2 *)ft.d_tw_seed : nullptr;
  p.ell_w = ft.d_ell_w;
  p.ell_meta = ft.d_ell_meta;
  p.ell_len = ft.d_ell_len;
  p.ell_woff = ft.d_ell_woff;
  p.ell_wfloats = ft.ell_wfloats;
  p.ell_slots = ft.ell_slots;
  p.L = plan->d.frame_length;
  p.S = plan->d.frame_shift;
  p.pad_left = a.pad_left;
  p.include_energy = plan->d.include_energy;
  p.use_power = plan->d.use_power;
  p.use_log = plan->d.use_log;
  p.log_floor = (float)plan->d.log_floor;
  p.inv_L = 1.0f / (float)plan->d.frame_length;
  p.num_utts = a.B;
  const int64_t chunks = (a.max_frames + G::GROUPS - 1) / G::GROUPS;
  if (chunks * a.B > 0x7fffffff || chunks > 0x3fffffff || a.out_stride * G::GROUPS > (a.out_f64 ? 0x0fffffff : 0x1fffffff) ||
      a.max_frames * plan->d.frame_shift > 0x7fffffff) {
    set_error("stft_batch: too many frame chunks in one call");
    return PDS_ERR_INVALID;
  }
  p.chunks_per_utt = (int)chunks;
  // LDS per workgroup: one exchange area per wave, the small tables, and the filter weight
  // rows when they fit.  Workgroup shapes in order of preference -- all CU_WAVES resident as two
  // workgroups, as one workgroup (one copy of the table instead of two), then fewer resident waves
  // with the table still in LDS (measured on the 38 KB gammatone table at N = 1024: 6 waves with
  // LDS weights beat 8 waves reading them through L1/L2 by 33 %).  Tables too large even for that
  // stay in global memory.
  const size_t lds_cu = 160 * 1024;
  const size_t per_wave = (size_t)G::EXCH_F2 * 8;
  // fused CMVN sums (pds_stft_cmvn_batch_f32): [2][coefficients rounded up to 4] float64 per wave, in LDS
  const int stat_c = a.d_stat_part ? plan->d.num_filts + (plan->d.include_energy ? 1 : 0) : 0;
  const int stat_cs = (stat_c + 3) & ~3;
  const size_t stat_pw = (size_t)2 * stat_cs * 8;
  int shapes[4][2] = {{CU_WAVES / 2, 2}, {CU_WAVES, 1}, {CU_WAVES * 3 / 4, 1}, {CU_WAVES * 5 / 8, 1}};
#ifdef PDS_ONE_WG  // (experiment: one workgroup per CU whatever fits)
  shapes[0][0] = CU_WAVES, shapes[0][1] = 1;
#endif
  if (CU_WAVES % 8 != 0) {
    // (three waves per SIMD: two workgroups of six waves do not tile the four SIMDs -- the second one of a
    // CU waited for the first to finish, measured as a launch twice as long -- one of twelve does)
    shapes[0][0] = CU_WAVES, shapes[0][1] = 1;
    shapes[1][0] = CU_WAVES * 2 / 3, shapes[1][1] = 1;
    shapes[2][0] = CU_WAVES / 3, shapes[2][1] = 1;
    shapes[3][0] = CU_WAVES / 3, shapes[3][1] = 1;
  }
  constexpr int CU_WAVES_K = CU_WAVES_STFT;  // launch bound of the instantiations below
  int waves = CU_WAVES / 2, wgs_per_cu = 2;
  size_t area_bytes = per_wave;  // a wave's private LDS area in this launch
  bool in_lds = false;
  // Filter walk: the plan's preferred one (fast_tables_create: fewest 16-byte LDS reads per item, or
  // PDS_STFT_WALK) when its tables fit in LDS beside the waves' areas, else the next: row segments
  // (2), segments of dense banks (1), ELL (0; its tables may also stay in global memory).
  p.seg_rounds = 0;
  p.seg_len = 0;
  p.num_filts = plan->d.num_filts;
  const bool pre = a.preemph != 0.0;
  int walk = 0;
  // (fused deltas exist for the row-segment walk only: take it whatever the plan prefers)
  const bool dl = a.dl_K > 0 && ft.rsn_rounds > 0;  // (its own table: numbered order)
  // prefetch instantiation (PF): 32 x 16 geometry with the row-segment walk, float32 samples, round-robin
  // scheduling (PDS_STFT_PF=0 keeps the kernel without it)
  constexpr bool PFG = (N1 == 32 || N1 == 64) && N2 == 16;
  const char *pf_env = std::getenv("PDS_STFT_PF");
  // (product build: the N = 1024 geometry's matrix-pipe walk -- configs[4], which waits for its loads at two waves per
  // SIMD and is not bound by the vector pipe: +2.7 %, profiles/r3j_prefetch_n1024_ab.txt; the 128-register geometry
  // and the other walks, where it measured -4 % / +-0, with -DPDS_EXPERIMENTS=1 only)
  constexpr bool PF_PRODUCT = N1 == 64 && N2 == 16;
  [[maybe_unused]] constexpr bool PF_MSEG = false;  // (superseded: the matrix-pipe launches of this geometry run three waves per SIMD instead, MSEG3)
  // the matrix-pipe walk at N = 1024: launch bounds of three waves per SIMD (149 - 161 VGPRs), up to twelve waves per CU
  constexpr bool MSEG3 = N1 == 64 && N2 == 16;
  constexpr int MS_WAVES = MSEG3 ? 12 : 4 * MINW, MS_MINW = MSEG3 ? 3 : MINW;
  static_assert(!MSEG3 || !PDS_LEAN_1024 || MINW == 3, "the 64 x 16 geometry is instantiated for three waves per SIMD (stft_geoms.def)");  // (the plain matrix-pipe launch IS the prefetch form)
  const bool pf_ok = (PDS_EXPERIMENTS || PF_PRODUCT) && PFG && !pre && !a.in_f64 && !a.in_i16 && !a.stretch && a.dl_K == 0 &&
                     ft.d_win_half && ft.d_tw_seed && !(pf_env && pf_env[0] == '0');
  constexpr int PF_WSTR = ((NROWS + 3) & ~3) % 8 == 4 ? ((NROWS + 3) & ~3) : ((NROWS + 3) & ~3) + 4;
  const size_t pf_extra = (PDS_PF_WIN == 1) ? (size_t)N2 * PF_WSTR * 4 : 0;  // window table in LDS
  const size_t lean_extra = LEAN_ALL ? (size_t)N2 * win_table_stride(NROWS) * 4 : 0;  // ... of the lean geometries
  const size_t mseg3_extra = LEAN_ALL ? 0 : (size_t)N2 * win_table_stride(NROWS) * 4;  // ... of the matrix-pipe walk at N = 1024 alone
  constexpr bool MSG = G::GROUPS == 4 && inl::is_pow2(N1);  // matrix-pipe segment walk instantiated
  constexpr bool SEGOK = G::GROUPS == 4 || (inl::is_pow2(N1) && N2 >= 32);  // segmented walk instantiated
  for (int cand = (G::GROUPS == 4) ? (dl ? 2 : ft.walk) : (SEGOK && ft.walk == 1 ? 1 : 0); cand >= 0 && !in_lds; --cand) {
    if (cand == 3 && (!MSG || ft.ms_rounds == 0 || pre || a.in_f64 || a.in_i16)) continue;
    // (no segmented variant of the fused pre-emphasis kernel)
    if ((cand == 2 && !dl && ft.rs_rounds == 0) || (cand == 1 && (ft.seg_rounds == 0 || pre || a.in_f64 || a.in_i16))) continue;
    const int meta_ints = cand == 3 ? ft.ms_meta_ints : cand == 2 ? (dl ? ft.rsn_rounds : ft.rs_rounds) * 64 : cand == 1 ? ft.seg_meta_ints : ft.ell_slots * N2;
    const int meta_pad = (std::max(meta_ints, USLOTS * N2) + 3) / 4 * 4;
    const size_t fixed = (size_t)N2 * 8 + (size_t)meta_pad * 4 + (cand == 2 && pf_ok ? pf_extra : 0) + lean_extra +
                         (cand == 3 && MSEG3 ? mseg3_extra : 0);
    const size_t table_bytes = (size_t)(cand == 3 ? ft.ms_wfloats : cand == 2 ? (dl ? ft.rsn_wfloats : ft.rs_wfloats) : cand == 1 ? ft.seg_wfloats : ft.ell_wfloats) * 4;
    // (segment sums live behind P in the wave's area)
    if (cand == 1 && (size_t)G::GROUPS * G::PSTR * 4 + (size_t)ft.seg_rounds * 64 * 4 * G::GROUPS > per_wave) continue;
    if (cand == 3 && (size_t)G::GROUPS * G::PSTR * 4 + (size_t)ft.ms_slots * 64 > per_wave) continue;
    // (the matrix-pipe walk at N = 1024 takes the area it needs -- the exchange, or the power rows + its partial-sum
    // slots -- instead of the geometry's: that is what lets eleven waves sit beside the 38 KB gammatone table)
    const size_t area_c = (cand == 3 && MSEG3)
                              ? std::max((size_t)G::XMIN_F * 4, (size_t)G::GROUPS * G::PSTR * 4 + (size_t)ft.ms_slots * 64)
                              : per_wave;
    const int ms_shapes[5][2] = {{12, 1}, {11, 1}, {10, 1}, {9, 1}, {8, 1}};
    const int (*try_shapes)[2] = (cand == 3 && MSEG3) ? ms_shapes : shapes;
    for (int si = 0; si < ((cand == 3 && MSEG3) ? 5 : 4); ++si) {
      const int *shape = try_shapes[si];
      if (shape[0] * (area_c + stat_pw) + fixed + table_bytes + 48 <= lds_cu / shape[1]) {  // (+ the ticket counter)
        waves = shape[0];
        wgs_per_cu = shape[1];
        in_lds = true;
        area_bytes = area_c;
        break;
      }
    }
    p.ell_meta_pad = meta_pad;
    p.ell_meta_ints = meta_ints;
    if (in_lds && cand == 3) {
      p.ell_w = ft.d_ms_w;
      p.ell_meta = ft.d_ms_meta;
      p.ell_wfloats = ft.ms_wfloats;
      p.seg_rounds = ft.ms_rounds;
      p.seg_len = ft.ms_len;
    } else if (in_lds && cand == 1) {
      p.ell_w = ft.d_seg_w;
      p.ell_meta = ft.d_seg_meta;
      p.ell_wfloats = ft.seg_wfloats;
      p.seg_rounds = ft.seg_rounds;
      p.seg_len = ft.seg_len;
    } else if (in_lds && cand == 2) {
      p.ell_w = dl ? ft.d_rsn_w : ft.d_rs_w;
      p.ell_meta = dl ? ft.d_rsn_meta : ft.d_rs_meta;
      p.ell_wfloats = dl ? ft.rsn_wfloats : ft.rs_wfloats;
      p.seg_rounds = dl ? ft.rsn_rounds : ft.rs_rounds;
      p.seg_len = dl ? ft.rsn_len : ft.rs_len;
    }
    if (in_lds) walk = cand;
  }
#ifdef PDS_FORCE_WAVES  // (experiment: throughput against resident waves per CU, two workgroups per CU)
  if (in_lds && PDS_FORCE_WAVES <= CU_WAVES / 2) waves = PDS_FORCE_WAVES, wgs_per_cu = 2;
#endif
  const size_t fixed = (size_t)N2 * 8 + (size_t)p.ell_meta_pad * 4 + (walk == 2 && in_lds && pf_ok ? pf_extra : 0) + lean_extra +
                       (walk == 3 && in_lds && MSEG3 ? mseg3_extra : 0);
  const size_t table_bytes = (size_t)p.ell_wfloats * 4;
  // the fused pre-emphasis variant exists for LDS-resident tables only; the rare other case
  // (dense complex bank at N >= 1024 plus pre-emphasis) takes the direct-DFT kernel
  if (a.in_i16 && !in_lds) {
    set_error("stft_batch_i16in: not served for this plan (filter table outside LDS)");
    return PDS_ERR_INVALID;
  }
  if (pre && !in_lds) return launch_stft_generic_f32(plan, a);
  p.preemph = (float)a.preemph;
  p.preemph_d = a.preemph;
  p.waves = waves;
  p.mf_tab = ft.d_mf_tab;
#if PDS_STAMPS
  p.stamps = g_stamp_buf;
#else
  p.stamps = nullptr;
#endif
  p.area_f = (int)(area_bytes / 4);
  size_t smem = waves * area_bytes + fixed + (in_lds ? table_bytes : 0);
  smem = (smem + 15) & ~(size_t)15;
  p.lds_ticket_off = (int)(smem / 4);  // the workgroup's ticket counter
  smem += 16;
  p.lds_stat_off = (int)(smem / 4);    // the waves' CMVN sums
  smem += (size_t)waves * stat_pw;
  p.stat_part = nullptr;
  p.stat_c = stat_c;
  p.stat_cs = stat_cs;
  p.waves_rcp = (unsigned)((0x100000000ull + (unsigned)waves - 1) / (unsigned)waves);
  {
    const char *dyn_env = std::getenv("PDS_STFT_DYN");
    p.dyn = (dyn_env && dyn_env[0] == '0') ? 0 : 1;
  }
  constexpr bool W4 = G::GROUPS == 4;  // the walks over four frames exist for the 16-lane geometries
  const bool seg = SEGOK && (walk == 1 || walk == 3), rsg = W4 && walk == 2, mseg = MSG && walk == 3;
  auto kern = pre      ? (rsg ? stft_wave_kernel<N1, N2, NROWS, CU_WAVES_K, MINW, true, true, false, 0, W4>
                              : stft_wave_kernel<N1, N2, NROWS, CU_WAVES_K, MINW, true, true>)
              : rsg    ? stft_wave_kernel<N1, N2, NROWS, CU_WAVES_K, MINW, true, false, false, 0, W4>
              : mseg   ? stft_wave_kernel<N1, N2, NROWS, MS_WAVES, MS_MINW, true, false, MSG ? 2 : 0>
              : seg    ? stft_wave_kernel<N1, N2, NROWS, CU_WAVES_K, MINW, true, false, SEGOK ? 1 : 0>
              : in_lds ? stft_wave_kernel<N1, N2, NROWS, CU_WAVES_K, MINW, true, false>
                       : stft_wave_kernel<N1, N2, NROWS, CU_WAVES_K, MINW, false, false>;
  // matrix-pipe front end: the 32 x 16 geometry with LDS-resident filter tables, when the plan has
  // its tables for this row count
  // (measured -16 % against the in-lane transform, profiles/r2a_front_ab_counters.txt: built with -DPDS_EXPERIMENTS=1 only)
  constexpr int MFS = (PDS_EXPERIMENTS && N1 == 32 && N2 == 16) ? mfma_front_steps(NROWS) : 0;
  bool mf = false;
  if constexpr (MFS > 0) {
    if (in_lds && ft.d_mf_tab && ft.mf_rows == NROWS && !mseg) {
      mf = true;
      kern = pre   ? (rsg ? stft_wave_kernel<N1, N2, NROWS, CU_WAVES_K, MINW, true, true, false, MFS, W4>
                          : stft_wave_kernel<N1, N2, NROWS, CU_WAVES_K, MINW, true, true, false, MFS>)
             : rsg ? stft_wave_kernel<N1, N2, NROWS, CU_WAVES_K, MINW, true, false, false, MFS, W4>
             : seg ? stft_wave_kernel<N1, N2, NROWS, CU_WAVES_K, MINW, true, false, W4, MFS>
                   : stft_wave_kernel<N1, N2, NROWS, CU_WAVES_K, MINW, true, false, false, MFS>;
    }
  }
  bool pf_used = false;
  if (MSEG3 && mseg) {
    p.win_half = ft.d_win_half;  // (regenerated twiddles, window slice from LDS: see the kernel)
    p.tw_seed = (const float2 *)ft.d_tw_seed;
  }
#if PDS_EXPERIMENTS
  if constexpr (PFG) {
    if (pf_ok && in_lds && !mf && (rsg || seg || (mseg && !PF_MSEG))) {
      pf_used = true;
      kern = rsg    ? stft_wave_kernel<N1, N2, NROWS, CU_WAVES_K, MINW, true, false, 0, 0, W4, float, float, 0, false, PFG>
             : mseg ? stft_wave_kernel<N1, N2, NROWS, CU_WAVES_K, MINW, true, false, MSG ? 2 : 0, 0, false, float, float, 0, false, PFG>
                    : stft_wave_kernel<N1, N2, NROWS, CU_WAVES_K, MINW, true, false, SEGOK ? 1 : 0, 0, false, float, float, 0, false, PFG>;
      p.win_half = ft.d_win_half;
      p.tw_seed = (const float2 *)ft.d_tw_seed;
    }
  }
#endif
  // ragged batches (pds_stft_batch_ragged_f32): the same kernels with stretch scheduling (STR); float32 samples
  // without fused pre-emphasis, in-lane front end
  // (power-of-two geometries; the others keep the round-robin order, which skips the chunks short utterances lack)
  bool str_used = false;
  if constexpr (inl::is_pow2(N1))
  // (tables in LDS: a bank whose table stays in global memory keeps the round-robin order)
  if (a.stretch && !pre && !a.in_f64 && !a.in_i16 && a.dl_K == 0 && !mf && a.d_chunk_prefix && in_lds) {
    str_used = true;
    kern = rsg      ? stft_wave_kernel<N1, N2, NROWS, CU_WAVES_K, MINW, true, false, 0, 0, W4, float, float, 0, true>
           : mseg   ? stft_wave_kernel<N1, N2, NROWS, MS_WAVES, MS_MINW, true, false, MSG ? 2 : 0, 0, false, float, float, 0, true>
           : seg    ? stft_wave_kernel<N1, N2, NROWS, CU_WAVES_K, MINW, true, false, SEGOK ? 1 : 0, 0, false, float, float, 0, true>
                    : stft_wave_kernel<N1, N2, NROWS, CU_WAVES_K, MINW, true, false, 0, 0, false, float, float, 0, true>;
    p.chunk_prefix = a.d_chunk_prefix;
  }
  // ... and the flows real ragged batches come in -- fused pre-emphasis, int16 PCM, both -- on the row-segment kernels of
  // the speech geometries (N = 512 and 1024, mel-like banks); the other banks and sizes keep the round-robin order
  int strx_which = -1;
  if constexpr (W4 && (N1 == 32 || N1 == 64) && N2 == 16) {
    if (a.stretch && (pre || a.in_i16) && !a.in_f64 && a.dl_K == 0 && !mf && a.d_chunk_prefix && in_lds && rsg && !a.d_stat_part) {
      using I = int16_t;
      str_used = true;
      kern = a.in_i16 ? (pre ? stft_wave_kernel<N1, N2, NROWS, CU_WAVES_K, MINW, true, true, 0, 0, true, I, float, 0, true>
                             : stft_wave_kernel<N1, N2, NROWS, CU_WAVES_K, MINW, true, false, 0, 0, true, I, float, 0, true>)
                      : stft_wave_kernel<N1, N2, NROWS, CU_WAVES_K, MINW, true, true, 0, 0, true, float, float, 0, true>;
      p.chunk_prefix = a.d_chunk_prefix;
      strx_which = a.in_i16 ? (pre ? 38 : 37) : 36;
    }
  }
  if (a.d_stat_part) {
    // (the sums are taken where ONE lane holds a coefficient of the item's four frames: the row-segment, segmented
    // and matrix-pipe walks of the 16-lane geometries, tables in LDS)
    if (!str_used || !in_lds || walk == 0 || G::GROUPS != 4 || (rsg && MINW > 2)) {
      set_error("stft_cmvn_batch: not served for this plan and call (needs a 16-lane power-of-two geometry, a segment "
                "walk with its tables and the waves' sums in LDS, float32 samples, no fused pre-emphasis)");
      return PDS_ERR_INVALID;
    }
    p.stat_part = a.d_stat_part;
  }
  // float64 samples (pds_stft_batch_f64in): the common power-of-two geometries, LDS-resident tables,
  // ELL or row-segment walk; float64 features without fused pre-emphasis only
  int f64_which = -1;
  if (a.in_f64) {
    constexpr bool F64IN = fast_f64in_kind(N1 * N2);
    if constexpr (F64IN) {
      if (!in_lds || (a.out_f64 && pre)) {
        set_error("stft_batch_f64in: not served for this plan (filter table outside LDS, or float64 features with fused pre-emphasis)");
        return PDS_ERR_INVALID;
      }
      mf = false;
      p.win_half = ft.d_win_half;  // (16-lane geometries: twiddles regenerated from seeds, see the kernel)
      p.tw_seed = (const float2 *)ft.d_tw_seed;
      if (a.out_f64)
        kern = rsg ? stft_wave_kernel<N1, N2, NROWS, F64_WAVES, F64_MINW, true, false, false, 0, W4, double, double>
                   : stft_wave_kernel<N1, N2, NROWS, F64_WAVES, F64_MINW, true, false, false, 0, false, double, double>;
      else if (pre)
        kern = rsg ? stft_wave_kernel<N1, N2, NROWS, F64_WAVES, F64_MINW, true, true, false, 0, W4, double, float>
                   : stft_wave_kernel<N1, N2, NROWS, F64_WAVES, F64_MINW, true, true, false, 0, false, double, float>;
      else
        kern = rsg ? stft_wave_kernel<N1, N2, NROWS, F64_WAVES, F64_MINW, true, false, false, 0, W4, double, float>
                   : stft_wave_kernel<N1, N2, NROWS, F64_WAVES, F64_MINW, true, false, false, 0, false, double, float>;
      f64_which = 12 + (a.out_f64 ? 4 : pre ? 2 : 0) + (rsg ? 1 : 0);
    } else {
      set_error("stft_batch_f64in: no fused float64-input kernel for this transform size");
      return PDS_ERR_INVALID;
    }
  }
  // int16 samples (pds_stft_batch_i16in: PCM as it sits in a WAV file, half the bytes of float32 over PCIe and from
  // HBM): converted as the frame is loaded, then exactly the float32 kernel; the geometries of the float64-sample
  // path, ELL or row-segment walk, LDS-resident tables
  int i16_which = -1;
  if (a.in_i16 && a.dl_K == 0 && strx_which < 0) {
    constexpr bool I16IN = fast_f64in_kind(N1 * N2);
    if constexpr (I16IN) {
      if (a.in_f64 || a.out_f64 || a.d_stat_part) {
        set_error("stft_batch_i16in: float32 features, no fused CMVN sums");
        return PDS_ERR_INVALID;
      }
      mf = false;
      kern = pre ? (rsg ? stft_wave_kernel<N1, N2, NROWS, CU_WAVES_K, MINW, true, true, false, 0, W4, int16_t, float>
                        : stft_wave_kernel<N1, N2, NROWS, CU_WAVES_K, MINW, true, true, false, 0, false, int16_t, float>)
                 : (rsg ? stft_wave_kernel<N1, N2, NROWS, CU_WAVES_K, MINW, true, false, false, 0, W4, int16_t, float>
                        : stft_wave_kernel<N1, N2, NROWS, CU_WAVES_K, MINW, true, false, false, 0, false, int16_t, float>);
      i16_which = 32 + (pre ? 2 : 0) + (rsg ? 1 : 0);
    } else {
      set_error("stft_batch_i16in: no fused int16-input kernel for this transform size");
      return PDS_ERR_INVALID;
    }
  }
  // fused statics + deltas (pds_stft_deltas_batch_f32): row-segment walk with at most two rounds and
  // a spare lane for the energy, see the kernel
  int dl_which = -1;
  if (a.dl_K > 0) {
    constexpr bool DELTAS = fast_deltas_kind(N1 * N2) && G::GROUPS == 4;
    if constexpr (DELTAS) {
      const int staged = ((a.dl_K + 1) * (plan->d.num_filts + (plan->d.include_energy ? 1 : 0)) + 3) / 4 * 16 + 64;  // floats
      if (!in_lds || !rsg || !dl || ft.rsn_rounds > 2 || (plan->d.include_energy && ft.rs_eslot < 0) || a.out_f64 ||
          staged > G::EXCH_F2 * 2) {
        set_error("stft_deltas_batch: not served for this plan and call (needs the row-segment filter walk with at "
                  "most two rounds in LDS and float32 features)");
        return PDS_ERR_INVALID;
      }
      mf = false;
      using D = double;
      using I = int16_t;
      kern = a.in_f64 ? (pre ? stft_wave_kernel<N1, N2, NROWS, 4 * DMINW, DMINW, true, true, false, 0, true, D, float, 2>
                             : stft_wave_kernel<N1, N2, NROWS, 4 * DMINW, DMINW, true, false, false, 0, true, D, float, 2>)
             : a.in_i16 ? (pre ? stft_wave_kernel<N1, N2, NROWS, 4 * DMINW, DMINW, true, true, false, 0, true, I, float, 2>
                               : stft_wave_kernel<N1, N2, NROWS, 4 * DMINW, DMINW, true, false, false, 0, true, I, float, 2>)
                      : (pre ? stft_wave_kernel<N1, N2, NROWS, 4 * DMINW, DMINW, true, true, false, 0, true, float, float, 2>
                             : stft_wave_kernel<N1, N2, NROWS, 4 * DMINW, DMINW, true, false, false, 0, true, float, float, 2>);
      if (a.in_f64 || PDS_DLT_CHAIN) {
        p.win_half = ft.d_win_half;  // (twiddles regenerated from seeds, see the kernel)
        p.tw_seed = (const float2 *)ft.d_tw_seed;
      }
      p.dl_order = a.dl_K;
      dl_which = a.in_i16 ? (pre ? 26 : 21) : 18 + (a.in_f64 ? 1 : 0) + (pre ? 10 : 0);  // 18, 19, 28, 29; int16 samples 21, 26
      for (int j = 0; j < 5; ++j) p.dl_f1[j] = (float)a.dl_taps[j];
      for (int j = 0; j < 9; ++j) p.dl_f2[j] = a.dl_K > 1 ? (float)a.dl_taps[5 + j] : 0.0f;
      p.dl_inner = plan->d.num_filts + (plan->d.include_energy ? 1 : 0);
      p.dl_eslot = plan->d.include_energy ? ft.rs_eslot : -1;
      p.chunk_prefix = a.d_chunk_prefix;
      // (measurement builds only, -DPDS_DL_DEBUG_SWITCH=1: a stray environment variable must not be able to make a
      // product launch skip its stores)
#if defined(PDS_DL_DEBUG_SWITCH) && PDS_DL_DEBUG_SWITCH
      p.dl_debug = std::getenv("PDS_DL_DEBUG") ? std::atoi(std::getenv("PDS_DL_DEBUG")) : 0;
#else
      p.dl_debug = 0;
#endif
    } else {
      set_error("stft_deltas_batch: no fused kernel for this transform size");
      return PDS_ERR_INVALID;
    }
  }
  // the dynamic-LDS limit is an attribute of the kernel on one device: raised once per
  // (instantiation, device, variant) and remembered (relaxed atomics: a lost race repeats the call)
  constexpr int kDevices = 64;
  static std::atomic<size_t> attr_smem[kDevices][40];
  const int which = strx_which >= 0  ? strx_which
                    : dl_which >= 0    ? dl_which
                    : f64_which >= 0 ? f64_which
                    : i16_which >= 0 ? i16_which
                    : pf_used        ? (rsg ? 27 : mseg ? 30 : 31)
                    : str_used       ? 22 + (rsg ? 0 : mseg ? 1 : seg ? 2 : 3)
                    : mseg           ? 20
                                     : (pre ? (rsg ? 5 : 2) : rsg ? 4 : seg ? 3 : (in_lds ? 1 : 0)) + (mf ? 6 : 0);
  const bool cached = plan->device >= 0 && plan->device < kDevices;
  if (!cached || smem > attr_smem[plan->device][which].load(std::memory_order_relaxed)) {
    PDS_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)smem));
    if (cached) attr_smem[plan->device][which].store(smem, std::memory_order_relaxed);
  }
  int64_t grid = (int64_t)ft.num_cus * wgs_per_cu;
  const int64_t need = (chunks * a.B + waves - 1) / waves;
  if (grid > need) grid = need;
  const int64_t grid_waves = grid * waves;
  p.step_utts = (int)(grid_waves / chunks);
  p.step_chunks = (int)(grid_waves % chunks);
  if (a.dl_K > 0 || str_used) {
    // the utterances' chunk counts summed up on the device, then one stretch of chunks per wave
    grid = std::min<int64_t>((int64_t)ft.num_cus * wgs_per_cu, std::max<int64_t>(1, (chunks * a.B + 4 * waves - 1) / (4 * waves)));
    if (!a.prefix_prepared) {
      const int32_t rc_prefix = launch_chunk_prefix(a.d_nframes, a.B, G::GROUPS, a.d_chunk_prefix, a.stream);
      if (rc_prefix != PDS_OK) return rc_prefix;
    }
  }
  if (a.grid_waves_out) *a.grid_waves_out = (int32_t)(grid * waves);
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(waves * 64), smem, a.stream, p);
  PDS_HIP(hipGetLastError());
  return PDS_OK;
}

}  // namespace pds
