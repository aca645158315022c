// Fused STFT filter-bank path, host side: the plan's tables for the kernel of stft_wave_kernel.h
// (fast_tables_create), the dispatch over the instantiated geometries (stft_geoms.def, one object file each:
// stft_geom.hip) and the small helper kernels.
#include "stft_wave_launch.h"

namespace pds {

#if PDS_STAMPS
unsigned long long *g_stamp_buf = nullptr;
extern "C" __attribute__((visibility("default"))) void pds_debug_set_stamp_buffer(unsigned long long *d_buf) {
  g_stamp_buf = d_buf;
}
#endif

extern "C" __attribute__((visibility("default"))) int32_t pds_build_experiments(void) { return PDS_EXPERIMENTS ? 1 : 0; }

bool fast_has_f64in(const pds_stft_plan *plan) { return plan->fast.kind && fast_f64in_kind(plan->fast.kind); }

// (fused CMVN sums: the 16-lane power-of-two geometries with a segment walk; whether the waves' sums fit in LDS
// beside the walk's tables is decided at the launch, which refuses otherwise)
bool fast_has_fused_cmvn(const pds_stft_plan *plan) {
  const FastTables &ft = plan->fast;
  // (the 128-register row-segment kernel of N = 512 does without: see STATS in the kernel)
  return ft.kind && ft.n2 == 16 && ((ft.n1 == 64 && ft.walk != 0) || (ft.n1 == 32 && (ft.walk == 1 || ft.walk == 3)));
}

bool fast_has_fused_deltas(const pds_stft_plan *plan) {
  const FastTables &ft = plan->fast;
  return ft.kind && fast_deltas_kind(ft.kind) && ft.rsn_rounds >= 1 && ft.rsn_rounds <= 2 &&
         (!plan->d.include_energy || ft.rs_eslot >= 0);
}

// chunk_prefix[b] = chunks of GROUPS frames in front of utterance b, [B] = all of them: one workgroup,
// a thread sums a contiguous slice of the utterances, the slices are scanned through LDS
__global__ __launch_bounds__(1024) void chunk_prefix_kernel(const int64_t *nframes, int B, int groups, int64_t *prefix) {
  __shared__ int64_t part[1024];
  const int per = (B + 1023) / 1024, lo = threadIdx.x * per, hi = lo + per < B ? lo + per : B;
  int64_t sum = 0;
  for (int b = lo; b < hi; ++b) sum += (nframes[b] + groups - 1) / groups;
  part[threadIdx.x] = sum;
  __syncthreads();
  for (int step = 1; step < 1024; step <<= 1) {
    const int64_t add = threadIdx.x >= (unsigned)step ? part[threadIdx.x - step] : 0;
    __syncthreads();
    part[threadIdx.x] += add;
    __syncthreads();
  }
  int64_t run = part[threadIdx.x] - sum;  // exclusive
  for (int b = lo; b < hi; ++b) {
    prefix[b] = run;
    run += (nframes[b] + groups - 1) / groups;
  }
  if (threadIdx.x == 1023) prefix[B] = part[1023];
}

int32_t launch_chunk_prefix(const int64_t *d_nframes, int B, int groups, int64_t *d_prefix, hipStream_t stream) {
  hipLaunchKernelGGL(chunk_prefix_kernel, dim3(1), dim3(1024), 0, stream, d_nframes, B, groups, d_prefix);
  PDS_HIP(hipGetLastError());
  return PDS_OK;
}

// The instantiated geometries (stft_geoms.def), one launcher each, defined in its own object file
#define PDS_GEOM(N1, N2, R, MINW) int32_t launch_geom_##N1##_##N2##_##R(const pds_stft_plan *plan, const BatchArgs &a);
#ifdef PDS_DEV_ONLY512
PDS_GEOM(32, 16, 25, 4)
#else
#include "stft_geoms.def"
#endif
#undef PDS_GEOM

int32_t launch_stft_fast_f32(const pds_stft_plan *plan, const BatchArgs &a) {
  const FastTables &ft = plan->fast;
  // the smallest row count of the plan's (N1, N2) that holds the frame
#define PDS_GEOM(N1, N2, R, MINW) \
  if (ft.n1 == N1 && ft.n2 == N2 && ft.rows <= R) return launch_geom_##N1##_##N2##_##R(plan, a);
#ifdef PDS_DEV_ONLY512  // (ISA inspection and variant builds: the headline geometry alone)
  PDS_GEOM(32, 16, 25, 4)
#else
#include "stft_geoms.def"
#endif
#undef PDS_GEOM
  set_error("stft_batch: no fused kernel for this plan");
  return PDS_ERR_INVALID;
}

int32_t fast_tables_create(pds_stft_plan *plan, const double *window, const int32_t *row_ptr,
                           const int32_t *col, const double *val) {
  FastTables &ft = plan->fast;
  ft.kind = 0;
  const pds_stft_desc &d = plan->d;
  int n1 = 0, n2 = 0;
  switch (d.dft_size) {
    case 128: n1 = 16; n2 = 8; break;
    case 256: n1 = 32; n2 = 8; break;
    case 512: n1 = 32; n2 = 16; break;
    case 1024: {
      // 64 x 16 (four frames per wave, 64-point in-lane real transform: 256 registers, 18 KB of LDS per wave)
      // or, with PDS_N1024_GEOM=32x32, 32 x 32 (two frames per wave, a column's transform over a lane pair:
      // 168 registers, 8.7 KB).  The second form was built for the dense banks, whose tables leave the first
      // one six waves per CU: it runs twelve, and measures the same (Gammatone-64 at 48 kHz: 0.339 against
      // 0.335 ms per step) -- with two frames per wave its filter walk is the segmented one at three reads per
      // four bins, 3300 cycles per frame against 1300 for the matrix-pipe walk over four frames, which eats
      // what the transforms gain.  Opt-in.
      const char *geom = PDS_EXPERIMENTS ? std::getenv("PDS_N1024_GEOM") : nullptr;  // (-DPDS_EXPERIMENTS=1 builds only)
      const bool wide = geom && std::strcmp(geom, "32x32") == 0;
      n1 = wide ? 32 : 64;
      n2 = wide ? 32 : 16;
      break;
    }
    case 2048: n1 = 64; n2 = 32; break;
    case 4096: n1 = 64; n2 = 64; break;
    case 160: n1 = 20; n2 = 8; break;
    case 200: n1 = 25; n2 = 8; break;
    case 240: n1 = 30; n2 = 8; break;
    case 320: n1 = 20; n2 = 16; break;
    case 400: n1 = 25; n2 = 16; break;
    case 480: n1 = 30; n2 = 16; break;
    case 640: n1 = 20; n2 = 32; break;
    case 800: n1 = 25; n2 = 32; break;
    case 960: n1 = 30; n2 = 32; break;
    default: return PDS_OK;  // generic kernel
  }
  const bool pow2 = (n1 & (n1 - 1)) == 0;
  // the mixed-radix geometries exist for transforms without zero padding only (N = L)
  if (!pow2 && d.frame_length != d.dft_size) return PDS_OK;
  const char *off = std::getenv("PDS_STFT_GENERIC");
  if (off && off[0] == '1') return PDS_OK;
  if (d.frame_length > d.dft_size || d.frame_length <= d.dft_size / 2) return PDS_OK;
  if (d.frame_shift >= (1 << 22)) return PDS_OK;  // the kernel forms g * S with a 24-bit multiply
  // fast_log() skips the denormal rescue: a floor below the normal range stays on the generic path
  if (d.use_log && !((float)d.log_floor >= 1.17549435e-38f)) return PDS_OK;
  if (d.num_filts > 32767) return PDS_OK;
  const int N = d.dft_size, H1 = n1 / 2, cols = (n1 - 1) / 2 + 1;  // WaveGeom::COLS
  std::vector<float> win((size_t)n1 * n2, 0.0f);
  for (int r = 0; r < n2; ++r)
    for (int k = 0; k < n1; ++k) {
      const int idx = n2 * k + r;
      if (idx < d.frame_length) win[(size_t)r * n1 + k] = (float)window[idx];
    }
  std::vector<float> tw((size_t)n2 * cols * 2, 0.0f);
  for (int r = 0; r < n2; ++r)
    for (int k1 = 1; k1 < cols; ++k1) {
      const double ang = -2.0 * M_PI * (double)((r * k1) % N) / (double)N;
      // undo rdft_scaled's factor (rdft_direct of the other sizes is unscaled)
      const double scale = (!pow2 || PDS_RDFT_DIT) ? 1.0 : (2 * k1 == H1) ? 1.0 : 0.5;
      tw[((size_t)r * cols + k1) * 2 + 0] = (float)(scale * std::cos(ang));
      tw[((size_t)r * cols + k1) * 2 + 1] = (float)(scale * std::sin(ang));
    }
  std::vector<int32_t> order(d.num_filts);
  for (int f = 0; f < d.num_filts; ++f) order[f] = f;
  auto span = [&](int f) {
    return row_ptr[f + 1] > row_ptr[f] ? col[row_ptr[f + 1] - 1] - (col[row_ptr[f]] & ~3) + 1 : 0;
  };
  std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return span(x) > span(y); });
  // ELL form of the filter table: slot s of lane j is filter order[s * n2 + j].  A row is the
  // filter's dense bin range, extended down to a multiple of 4 bins and up to the slot's
  // common length (a multiple of 8); zero weights fill the rest.
  const int slots = (d.num_filts + n2 - 1) / n2;
  const int pstr = ((N / 2 + 2 + 15) / 32) * 32 + 16;  // WaveGeom::PSTR
  std::vector<int32_t> ell_meta((size_t)std::max(slots, 1) * n2, 0), ell_len(std::max(slots, USLOTS), 0),
      ell_woff(std::max(slots, USLOTS), 0);
  std::vector<float> ell_w;
  for (int sl = 0; sl < slots; ++sl) {
    int longest = 8;
    std::vector<int> first(n2, 0);
    for (int j = 0; j < n2 && sl * n2 + j < d.num_filts; ++j) {
      const int f = order[sl * n2 + j];
      if (row_ptr[f + 1] == row_ptr[f]) continue;
      first[j] = col[row_ptr[f]] & ~3;  // cols ascend within a row
      longest = std::max(longest, (span(f) + 7) / 8 * 8);
    }
    // LDS bank spreading: the lanes of a frame read 16 bytes each from their row's current
    // position, all at the same step, so rows whose starts coincide modulo 64 banks (16 groups of
    // 4 bins) collide at every step.  A row may start up to (longest - span) bins early at no
    // cost (the slack is zero weights): take the early start that collides with the fewest rows
    // placed so far, rows with the least slack first.  (Identical starts are one broadcast read.)
    {
      std::vector<int> lanes;
      for (int j = 0; j < n2 && sl * n2 + j < d.num_filts; ++j)
        if (span(order[sl * n2 + j]) > 0) lanes.push_back(j);
      std::stable_sort(lanes.begin(), lanes.end(), [&](int x, int y) {
        return span(order[sl * n2 + x]) > span(order[sl * n2 + y]);
      });
      std::vector<std::vector<int>> taken(16);
      for (int j : lanes) {
        const int slack = std::min((longest - span(order[sl * n2 + j])) / 4, first[j] / 4);
        int best_shift = 0, best_cost = 1 << 30;
        for (int sh = 0; sh <= slack; ++sh) {
          const int q = first[j] / 4 - sh;
          int cost = 0;
          for (int other : taken[q % 16]) cost += other != q;
          if (cost < best_cost) {
            best_cost = cost;
            best_shift = sh;
          }
        }
        first[j] -= 4 * best_shift;
        taken[(first[j] / 4) % 16].push_back(first[j] / 4);
      }
    }
    ell_len[sl] = longest;
    ell_woff[sl] = (int32_t)ell_w.size();
    const int wstride = longest + 4;
    ell_w.resize(ell_w.size() + (size_t)n2 * wstride, 0.0f);
    for (int j = 0; j < n2; ++j) {
      const int f = sl * n2 + j < d.num_filts ? order[sl * n2 + j] : -1;
      if (first[j] + longest > pstr)  // keep every 16-byte read inside the frame's P row
        first[j] = (pstr - longest) & ~3;
      if (first[j] < 0) return PDS_OK;  // cannot happen for N >= 128; stay generic if it does
      ell_meta[(size_t)sl * n2 + j] = first[j] | ((f + 1) << 16);
      if (f < 0) continue;
      for (int q = row_ptr[f]; q < row_ptr[f + 1]; ++q) {
        const int t = col[q] - first[j];
        if (t < 0 || t >= longest) return PDS_OK;
        ell_w[(size_t)ell_woff[sl] + (size_t)j * wstride + t] = (float)val[q];
      }
    }
  }
  if (ell_w.empty()) ell_w.assign(4, 0.0f);
  ft.ell_wfloats = (int)ell_w.size();
  ft.ell_slots = slots;
  int32_t rc = PDS_OK;
  // Segmented walk (16-lane geometries: four frames per wave; see the kernel).  Every filter's
  // dense row, starting at a multiple of 4 bins, is cut into segments of seg_len bins; a filter's
  // segments take consecutive slots, slot = round * 64 + lane.  It is used when it needs a fifth
  // fewer 16-byte LDS reads per item than the ELL walk (two per 4 bins of the longest row of
  // every slot there; five per 4 bins of a segment and FOUR frames here, plus the sums).
  // (Round 2: also for the power-of-two geometries with 32 and 64 lanes per frame, N = 2048 and 4096, two
  // frames and one frame per wave: there the ELL walk's pace is set by its longest row -- the 213-bin top
  // filter of an 80-filter mel bank at N = 4096 while the lanes of short filters idle -- and segments of
  // equal length dealt to all 64 lanes need fewer reads in fewer dependent steps.)
  const int groups = 64 / n2;  // frames per wave
  bool dense_bank = false;  // the bank is one a segment walk pays for (whether or not the plain one fits the wave's area)
  long dense_reads = 0;     // ... and the 16-byte LDS reads per item the plain segment walk would need for it
  if ((n2 == 16 || (pow2 && n2 >= 32)) && d.num_filts > 0 && d.num_filts <= 65535) {
    long ell_reads = 0;
    for (int sl = 0; sl < slots; ++sl) ell_reads += 2L * (ell_len[sl] / 4);  // per lane and item
    ft.ell_reads = ell_reads;
    const int free_slots = (wave_area_floats(n1, n2) - groups * pstr) / groups;  // partial slots behind P
    int best_len = 0, best_rounds = 0;
    long best_reads = 0, any_reads = 0;  // (any_reads: the fewest reads of a segment length whether or not its sums fit)
    const char *only_len = std::getenv("PDS_SEG_LEN");  // (measurement: one segment length)
    for (int len : {16, 32, 64}) {
      if (only_len && std::atoi(only_len) != len) continue;
      long nseg = 0;
      for (int f = 0; f < d.num_filts; ++f) nseg += (span(f) + len - 1) / len;
      const int rounds = (int)((nseg + 63) / 64);
      if (rounds == 0 || rounds * 64 > 65535) continue;
      int longest = 0;
      for (int f = 0; f < d.num_filts; ++f) longest = std::max(longest, (span(f) + len - 1) / len);
      // (reads of the rounds, their epilogues, and the partial sums a filter's lane adds up four at a time)
      const long reads = (long)rounds * (len / 4) * (1 + groups) + rounds + (nseg + 63) / 64 + 4 + 4L * ((longest + 3) / 4);
      if (!any_reads || reads < any_reads) any_reads = reads;
      if (rounds * 64 > free_slots) continue;
      if (!best_len || reads < best_reads) best_len = len, best_rounds = rounds, best_reads = reads;
    }
    const char *force = std::getenv("PDS_STFT_SEGMENTED");  // "1": whenever feasible, "0": never
    if (!force && std::getenv("PDS_STFT_WALK") && std::strcmp(std::getenv("PDS_STFT_WALK"), "seg") == 0) force = "1";
    // (measured: Gammatone-64 at N = 1024, 180 reads against 375: +3.5 %; Gabor-64 at N = 512, 48 against
    // 64: +5 %; the 40-filter mel bank, 26 against 32: -0.5 %; 80 mel filters, more reads: -2 %)
    // with four and more ELL slots their per-slot epilogues (log, scattered stores) weigh in as well
    // (80 mel filters at 48 kHz: N = 4096, one frame per wave, 54 reads against 96: +18 %; N = 2048, two frames
    // per wave, 46 against 60: -6 %, so the 32-lane geometries take it only when forced)
    auto wanted = [&](long reads) {
      return force      ? force[0] == '1'
             : n2 == 64 ? reads < ell_reads
             : n2 == 32 ? 10 * reads <= 7 * ell_reads  // (dense banks; the mel bank above: 46 against 60)
                        : 5 * reads <= 4 * ell_reads || (slots >= 4 && reads < ell_reads);
    };
    const bool want = wanted(best_reads);
    dense_bank = any_reads > 0 && wanted(any_reads);
    dense_reads = any_reads;
    if (best_len && want) {
      const int len = best_len, nslots = best_rounds * 64, wstride = len + 4;
      std::vector<float> seg_w((size_t)nslots * wstride, 0.0f);
      std::vector<int32_t> seg_meta((size_t)nslots + d.num_filts, 0);
      int slot = 0;
      bool ok = true;
      for (int f = 0; f < d.num_filts && ok; ++f) {
        const int count = (span(f) + len - 1) / len;
        seg_meta[(size_t)nslots + f] = slot | (count << 16);
        const int base = count ? col[row_ptr[f]] & ~3 : 0;
        for (int k = 0; k < count; ++k, ++slot) {
          int first = base + k * len;
          if (first + len > pstr) first = (pstr - len) & ~3;  // keep every read inside the P row
          if (first < 0) ok = false;
          seg_meta[slot] = first;
          for (int q = row_ptr[f]; q < row_ptr[f + 1]; ++q) {
            if (col[q] < base + k * len || col[q] >= base + (k + 1) * len) continue;  // this segment's bins
            const int t = col[q] - first;
            if (t < 0 || t >= len) ok = false; else seg_w[(size_t)slot * wstride + t] = (float)val[q];
          }
        }
      }
      if (ok) {
        ft.seg_reads = best_reads;
        ft.seg_rounds = best_rounds;
        ft.seg_len = len;
        ft.seg_wfloats = (int)seg_w.size();
        ft.seg_meta_ints = (int)seg_meta.size();
        if (rc == PDS_OK) rc = upload(&ft.d_seg_w, seg_w.data(), seg_w.size());
        if (rc == PDS_OK) rc = upload(&ft.d_seg_meta, seg_meta.data(), seg_meta.size());
      }
    }
  }
  // Row-segment walk (rseg_tables.h; 16-lane geometries).  PDS_STFT_WALK=ell | seg | rseg forces one
  // of the walks (measurement); by default the one with the fewest 16-byte LDS reads per item runs.
  if (n2 == 16 && rc == PDS_OK) {
    RsegTables rs;
    const int area_floats = wave_area_floats(n1, n2);  // WaveGeom::EXCH_F2 * 2
    if (build_rseg(d.num_filts, row_ptr, col, val, N / 2 + 1, area_floats / 4, 8, rs)) {
      ft.rs_rounds = rs.rounds;
      ft.rs_len = rs.seg_len;
      ft.rs_wfloats = (int)rs.w.size();
      ft.rs_reads = rs.reads_per_lane();
      ft.rs_cost = rs.cost;
      // fused statics + deltas launches (transform sizes with such instantiations, at most two rounds): the
      // table in numbered order, and a lane that finishes no filter to carry the energy column
      RsegTables rn;
      ft.rs_eslot = -1;
      if (rc == PDS_OK && fast_deltas_kind(N) && build_rseg(d.num_filts, row_ptr, col, val, N / 2 + 1, area_floats / 4, 2, rn, true)) {
        ft.rsn_rounds = rn.rounds;
        ft.rsn_len = rn.seg_len;
        ft.rsn_wfloats = (int)rn.w.size();
        for (size_t at = 0; at < rn.meta.size() && ft.rs_eslot < 0; ++at)
          if ((rn.meta[at] >> 16) == 0) ft.rs_eslot = (int)at;
        rc = upload(&ft.d_rsn_w, rn.w.data(), rn.w.size());
        if (rc == PDS_OK) rc = upload(&ft.d_rsn_meta, rn.meta.data(), rn.meta.size());
      }
      if (rc == PDS_OK) rc = upload(&ft.d_rs_w, rs.w.data(), rs.w.size());
      if (rc == PDS_OK) rc = upload(&ft.d_rs_meta, rs.meta.data(), rs.meta.size());
    }
  }
  // Matrix-pipe segment walk (mseg_tables.h): built where the segmented walk is (dense banks), for the
  // power-of-two 16-lane geometries
  if (n2 == 16 && pow2 && rc == PDS_OK && dense_bank) {
    MsegTables ms;
    const int area_floats = wave_area_floats(n1, n2);  // WaveGeom::EXCH_F2 * 2
    const int max_slots = (area_floats - 4 * pstr) / 16;  // partial-sum slots (four float4 each) behind P
    if (build_mseg(d.num_filts, row_ptr, col, val, pstr, max_slots, ms)) {
      ft.ms_rounds = ms.rounds;
      ft.ms_slots = ms.slots;
      ft.ms_len = ms.seg_len;
      ft.ms_wfloats = (int)ms.w.size();
      ft.ms_meta_ints = (int)ms.meta.size();
      ft.ms_reads = ms.reads_per_lane();
      rc = upload(&ft.d_ms_w, ms.w.data(), ms.w.size());
      if (rc == PDS_OK) rc = upload(&ft.d_ms_meta, ms.meta.data(), ms.meta.size());
    }
  }
  {
    const char *force = std::getenv("PDS_STFT_WALK");
    ft.walk = ft.seg_rounds > 0 ? 1 : 0;  // (the segmented walk's own criterion, above)
    // (a dense bank whose plain segment walk does not fit the wave's area still competes with that walk's read
    // count: the matrix-pipe form below stands in for it)
    const bool ms_stands_in = ft.walk == 0 && dense_bank && ft.ms_rounds > 0 && N >= PDS_MSEG_MIN_N;
    const long other = ft.walk == 1 ? ft.seg_reads : ms_stands_in ? dense_reads : ft.ell_reads;
    // (measured: 40 mel filters, 15 reads in one round against 32: +5 %; 80 mel filters, 20 in two rounds
    // against 32: +2 %; Gabor-64, 45 in three rounds against the segmented walk's 48: -1 ... -5 %: every
    // round has an epilogue of its own)
    if (ft.rs_rounds > 0 && 10 * ft.rs_reads <= 7 * other) ft.walk = 2;
    if (force && std::strcmp(force, "ell") == 0) ft.walk = 0;
    if (force && std::strcmp(force, "seg") == 0) ft.walk = ft.seg_rounds > 0 ? 1 : 0;
    if (force && std::strcmp(force, "rseg") == 0) ft.walk = ft.rs_rounds > 0 ? 2 : ft.walk;
    // (matrix-pipe segments: where the segmented walk would run at two waves per SIMD or fewer -- there a
    // wave's walk is a chain of LDS round trips nobody hides; PDS_STFT_WALK=mseg: wherever built)
    if ((ft.walk == 1 || (ft.walk == 0 && ms_stands_in)) && ft.ms_rounds > 0 && N >= PDS_MSEG_MIN_N && !force) ft.walk = 3;
    if (force && std::strcmp(force, "mseg") == 0 && ft.ms_rounds > 0) ft.walk = 3;
    if (std::getenv("PDS_DEBUG_PLAN"))
      std::fprintf(stderr,
                   "pds plan N=%d filters=%d: ell slots %d reads %ld | seg rounds %d len %d reads %ld | rseg rounds %d "
                   "len %d reads %ld cost %ld | mseg rounds %d len %d reads %ld -> walk %d\n",
                   N, d.num_filts, ft.ell_slots, ft.ell_reads, ft.seg_rounds, ft.seg_len, ft.seg_reads, ft.rs_rounds,
                   ft.rs_len, ft.rs_reads, ft.rs_cost, ft.ms_rounds, ft.ms_len, ft.ms_reads, ft.walk);
  }
  if (rc == PDS_OK) rc = upload(&ft.d_ell_w, ell_w.data(), ell_w.size());
  if (rc == PDS_OK) rc = upload(&ft.d_ell_meta, ell_meta.data(), ell_meta.size());
  if (rc == PDS_OK) rc = upload(&ft.d_ell_len, ell_len.data(), ell_len.size());
  if (rc == PDS_OK) rc = upload(&ft.d_ell_woff, ell_woff.data(), ell_woff.size());
  if (rc == PDS_OK) rc = upload(&ft.d_window, win.data(), win.size());
  if (rc == PDS_OK) rc = upload(&ft.d_twiddle, tw.data(), tw.size());
  if (((n1 == 32 || n1 == 64) && n2 == 16) || lean_geometry(n1, n2)) {
    // float64-sample instantiations (and the prefetch experiment): window times 1/2 and the twiddle seeds
    // W_N^r, W_N^4r, W_N^(n1/4 r) (inl::twiddle_chain)
    std::vector<float> wh(win), seed((size_t)n2 * 6);
    for (float &v : wh) v *= PDS_RDFT_DIT ? 1.0f : 0.5f;  // (rdft_dit: unscaled outputs, the whole window)
    const int mult[3] = {1, 4, n1 / 4};
    for (int r = 0; r < n2; ++r)
      for (int j = 0; j < 3; ++j) {
        const double ang = -2.0 * M_PI * (double)((r * mult[j]) % N) / (double)N;
        seed[((size_t)r * 3 + j) * 2 + 0] = (float)std::cos(ang);
        seed[((size_t)r * 3 + j) * 2 + 1] = (float)std::sin(ang);
      }
    if (rc == PDS_OK) rc = upload(&ft.d_win_half, wh.data(), wh.size());
    if (rc == PDS_OK) rc = upload(&ft.d_tw_seed, seed.data(), seed.size());
  }
  std::vector<float> tws((size_t)n2 * 2);
  for (int r = 0; r < n2; ++r) {
    const double ang = -2.0 * M_PI * (double)r / (double)(2 * n2);
    tws[2 * r] = (float)std::cos(ang);
    tws[2 * r + 1] = (float)std::sin(ang);
  }
  if (rc == PDS_OK) rc = upload(&ft.d_tw_special, tws.data(), tws.size());
  // matrix-pipe front end (mfma_front.h): the 32 x 16 geometry, tables for the row count of the
  // instantiation launch_stft_fast_f32 picks.  Opt-in (PDS_STFT_FRONT=mfma) while the in-lane
  // transform measures faster (profiles/r2a_front_ab_counters.txt).
  {
    const char *front = PDS_EXPERIMENTS ? std::getenv("PDS_STFT_FRONT") : nullptr;  // (-DPDS_EXPERIMENTS=1 builds only)
    const int rows = (d.frame_length + n2 - 1) / n2;
    int bucket = 0;
    for (int b : {20, 25, 28, 30, 32})
      if (!bucket && rows <= b) bucket = b;
    MfmaFrontTables mt;
    if (rc == PDS_OK && n1 == 32 && n2 == 16 && bucket && front && std::strcmp(front, "mfma") == 0 &&
        build_mfma_front(n1, bucket, d.frame_length, window, mt)) {
      std::vector<float> image;
      image.reserve(mt.words());
      image.insert(image.end(), mt.win.begin(), mt.win.end());
      for (int32_t v : mt.off) {
        float f;
        std::memcpy(&f, &v, sizeof f);
        image.push_back(f);
      }
      image.insert(image.end(), mt.emask.begin(), mt.emask.end());
      image.insert(image.end(), mt.a_re.begin(), mt.a_re.end());
      image.insert(image.end(), mt.a_im.begin(), mt.a_im.end());
      image.insert(image.end(), mt.tw.begin(), mt.tw.end());
      rc = upload(&ft.d_mf_tab, image.data(), image.size());
      ft.mf_rows = bucket;
    }
  }
  if (rc != PDS_OK) return rc;
  hipDeviceProp_t prop;
  PDS_HIP(hipGetDeviceProperties(&prop, plan->device));
  ft.num_cus = prop.multiProcessorCount;
  ft.n1 = n1;
  ft.n2 = n2;
  ft.rows = (d.frame_length + n2 - 1) / n2;
  ft.kind = d.dft_size;
  return PDS_OK;
}

void fast_tables_destroy(pds_stft_plan *plan) {
  FastTables &ft = plan->fast;
  (void)hipFree(ft.d_window);
  (void)hipFree(ft.d_twiddle);
  (void)hipFree(ft.d_tw_special);
  (void)hipFree(ft.d_win_half);
  (void)hipFree(ft.d_tw_seed);
  (void)hipFree(ft.d_mf_tab);
  (void)hipFree(ft.d_rs_w);
  (void)hipFree(ft.d_ms_w);
  (void)hipFree(ft.d_ms_meta);
  (void)hipFree(ft.d_rsn_w);
  (void)hipFree(ft.d_rsn_meta);
  (void)hipFree(ft.d_rs_meta);
  (void)hipFree(ft.d_seg_w);
  (void)hipFree(ft.d_seg_meta);
  (void)hipFree(ft.d_ell_w);
  (void)hipFree(ft.d_ell_meta);
  (void)hipFree(ft.d_ell_len);
  (void)hipFree(ft.d_ell_woff);
  ft = FastTables();
}

}  // namespace pds
