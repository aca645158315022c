// Tables of the "row-segment" filter walk of the fused STFT kernel (stft_fast.hip, template
// parameter RSG; 16-lane geometries: four frames per wave).
//
// The power spectra of a wave's four frames lie in LDS bin-major, P[bin][frame] (one 16-byte read
// returns a bin of all four frames).  Every filter's dense bin range is cut into segments of T
// bins; the segments are dealt to the 64 lanes, `rounds` segments each, a filter's segments on
// consecutive lanes of one 16-lane row.  A lane reads its T weights (T/4 16-byte reads, stored
// [round][T/4][lane] so that the wave's read is one contiguous KB) and T bins (16 bytes each) and
// holds four partial sums, one per frame; the segments of a filter are added with two DPP steps
// (a filter has at most four segments) and the filter's first lane applies the log and stores four
// coefficients.  Against the ELL walk (a lane = a filter of ONE frame, 16-byte reads of weights and
// powers alike): 40 mel filters at N = 512 need 15 reads per lane and item instead of 32, all of
// them in flight at once (one LDS round trip instead of one per row step), and 48 multiply-adds
// instead of 64.
//
// Host code only (plain C++): included by stft_fast.hip and by tests/csrc/test_rseg_tables.cpp.
#pragma once
#include <algorithm>
#include <cstdint>
#include <vector>

namespace pds {

struct RsegTables {
  int seg_len = 0;   // T, a multiple of 4
  int rounds = 0;    // segments per lane
  int nbp = 0;       // bins the kernel keeps in LDS (>= num_bins, a multiple of 4; bin nbp = dump slot)
  // meta[round * 64 + lane] = first bin | continues(+1) << 14 | continues(+2) << 15 | (filter + 1) << 16
  // (filter + 1 only on the first lane of a filter's run, 0 elsewhere)
  std::vector<int32_t> meta;
  std::vector<float> w;  // [round][T / 4][lane][4]
  long cost = 0;     // LDS cycles per item the layout was priced at (reads incl. bank conflicts + epilogues)
  long reads_per_lane() const { return (long)rounds * (seg_len + seg_len / 4); }
};

// LDS cycles of one round's power reads: a 16-byte read is served in four groups of 16 lanes
// ({0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} and the same + 32: MI355X_MICROARCH.md, LDS), a group in as
// many cycles as its busiest bank quartet has distinct addresses.  A bin is 16 bytes, so its quartet is
// bin mod 16; the lanes of a group step through their segments together.
inline long rseg_read_cycles(const int32_t *meta, int seg_len) {
  static const int kGroup[32] = {0, 0, 0, 0, 1, 1, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0,
                                 1, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1};
  long cycles = 0;
  for (int half = 0; half < 2; ++half)
    for (int grp = 0; grp < 2; ++grp)
      for (int t = 0; t < seg_len; ++t) {
        int seen[16][16], count[16] = {0};
        for (int l = 0; l < 32; ++l) {
          if (kGroup[l] != grp) continue;
          const int bin = (meta[32 * half + l] & 0x3fff) + t, qd = bin & 15;
          bool dup = false;
          for (int i = 0; i < count[qd]; ++i) dup = dup || seen[qd][i] == bin;
          if (!dup) seen[qd][count[qd]++] = bin;
        }
        int worst = 1;
        for (int qd = 0; qd < 16; ++qd) worst = std::max(worst, count[qd]);
        cycles += worst;
      }
  return cycles;
}

// CSR filter table (cols ascending within a row), num_bins = N / 2 + 1.  max_bins: bins (of four
// floats) the wave's LDS area can hold including the dump slot.  False when no segment length fits.
inline bool build_rseg(int num_filts, const int32_t *row_ptr, const int32_t *col, const double *val, int num_bins,
                       int max_bins, int max_rounds, RsegTables &out) {
  if (num_filts <= 0 || num_filts > 65534 || num_bins >= (1 << 14)) return false;
  const int nbp = (num_bins + 3) / 4 * 4;
  if (nbp + 1 > max_bins) return false;
  auto lo = [&](int f) { return col[row_ptr[f]]; };
  auto span = [&](int f) { return row_ptr[f + 1] > row_ptr[f] ? col[row_ptr[f + 1] - 1] - lo(f) + 1 : 0; };
  // lays the segments out for one segment length: meta only (first bins, run flags, filters)
  auto layout = [&](int T, std::vector<int32_t> &meta) -> int {
    long cursor = 0;
    std::vector<long> at(num_filts);
    for (int f = 0; f < num_filts; ++f) {
      const int n = std::max(1, (span(f) + T - 1) / T);
      if (n > 4) return 0;
      if (cursor % 16 + n > 16) cursor = (cursor / 16 + 1) * 16;  // a run stays inside one DPP row
      at[f] = cursor;
      cursor += n;
    }
    const int rounds = (int)((cursor + 63) / 64);
    if (rounds > max_rounds) return 0;
    meta.assign((size_t)rounds * 64, 0);
    // quartet occupancy of the four read groups of every round: a one-segment filter whose span
    // is shorter than T may start up to T - span bins early (zero weights in front); it takes
    // the start whose quartets collide least with the lanes laid out before it
    for (int f = 0; f < num_filts; ++f) {
      const int n = std::max(1, (span(f) + T - 1) / T);
      for (int k = 0; k < n; ++k) {
        const long pos = at[f] + k;
        const int base = span(f) ? lo(f) + k * T : 0;
        int first = std::min(base, nbp - T);  // keep every read inside the kept bins
        if (n == 1 && span(f) > 0) {
          // any start that keeps the filter's last bin inside the segment
          const int lowest = std::max(0, lo(f) + span(f) - T);
          long best = -1;
          int best_first = first;
          for (int cand = first; cand >= lowest; --cand) {
            meta[pos] = cand;
            const long c = rseg_read_cycles(meta.data() + pos / 64 * 64, T);
            if (best < 0 || c < best) best = c, best_first = cand;
          }
          first = best_first;
        }
        int32_t m = first;
        if (k + 1 < n) m |= 1 << 14;
        if (k + 2 < n) m |= 1 << 15;
        if (k == 0) m |= (f + 1) << 16;
        meta[pos] = m;
      }
    }
    return rounds;
  };
  long best_cost = -1;
  int best_len = 0;
  std::vector<int32_t> meta, best_meta;
  for (int len = 4; len <= 64 && len <= nbp; len += 4) {
    const int rounds = layout(len, meta);
    if (!rounds) continue;
    // LDS cycles per item: power reads as laid out, weight reads (conflict-free, 4 cycles per 16 bytes),
    // and the round's epilogue (sums, logs, four predicated stores) priced in the same unit
    long cost = 0;
    for (int rd = 0; rd < rounds; ++rd) cost += rseg_read_cycles(meta.data() + (size_t)rd * 64, len) + len + 40;
    if (best_cost < 0 || cost < best_cost) best_cost = cost, best_len = len, best_meta = meta;
  }
  if (best_cost < 0) return false;
  const int T = best_len, rounds = (int)(best_meta.size() / 64);
  out.seg_len = T;
  out.rounds = rounds;
  out.nbp = nbp;
  out.cost = best_cost;
  out.meta = best_meta;
  out.w.assign((size_t)rounds * T * 64, 0.0f);
  // weights: walk the runs again in layout order (first lane of a run carries the filter)
  for (size_t pos = 0; pos < out.meta.size(); ++pos) {
    const int f = (out.meta[pos] >> 16) - 1;
    if (f < 0) continue;
    const int n = std::max(1, (span(f) + T - 1) / T);
    for (int k = 0; k < n; ++k) {
      const size_t at = pos + k;
      const int round = (int)(at / 64), lane = (int)(at % 64);
      const int first = out.meta[at] & 0x3fff;
      const int base = span(f) ? lo(f) + k * T : 0;  // bins [base, base + T) of the filter belong to this lane
      for (int q = row_ptr[f]; q < row_ptr[f + 1]; ++q) {
        if (col[q] < base || col[q] >= base + T) continue;
        const int t = col[q] - first;
        if (t < 0 || t >= T || col[q] >= num_bins) return false;
        out.w[(((size_t)round * (T / 4) + t / 4) * 64 + lane) * 4 + t % 4] = (float)val[q];
      }
    }
  }
  return true;
}

}  // namespace pds
