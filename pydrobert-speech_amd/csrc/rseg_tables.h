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
#include <utility>
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
inline int rseg_read_group(int lane32) {
  static const int kGroup[32] = {0, 0, 0, 0, 1, 1, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0,
                                 1, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1};
  return kGroup[lane32];
}

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
// numbered: keep the filters in their numbered order along the lanes (no search for fewer bank conflicts): the
// lanes' coefficient stores then ascend with the lane, which the fused statics + deltas launch, with six times
// the stores, needs more than the conflict-free reads
inline bool build_rseg(int num_filts, const int32_t *row_ptr, const int32_t *col, const double *val, int num_bins,
                       int max_bins, int max_rounds, RsegTables &out, bool numbered = false) {
  if (num_filts <= 0 || num_filts > 65534 || num_bins >= (1 << 14)) return false;
  const int nbp = (num_bins + 3) / 4 * 4;
  if (nbp + 1 > max_bins) return false;
  auto lo = [&](int f) { return col[row_ptr[f]]; };
  auto span = [&](int f) { return row_ptr[f + 1] > row_ptr[f] ? col[row_ptr[f + 1] - 1] - lo(f) + 1 : 0; };
  // lays the segments out for one segment length: meta only (first bins, run flags, filters).
  // search: also look for a better assignment of the runs to lanes (below)
  auto layout = [&](int T, bool search, std::vector<int32_t> &meta) -> int {
    std::vector<int> nseg(num_filts);
    for (int f = 0; f < num_filts; ++f) {
      nseg[f] = std::max(1, (span(f) + T - 1) / T);
      if (nseg[f] > 4) return 0;
    }
    // runs in a given order (entries >= num_filts are spare lanes) -> slot of every run's first lane
    std::vector<long> at;
    auto place = [&](const std::vector<int> &order) -> int {
      long cursor = 0;
      at.assign(order.size(), 0);
      for (int e : order) {
        const int n = e < num_filts ? nseg[e] : 1;
        if (cursor % 16 + n > 16) cursor = (cursor / 16 + 1) * 16;  // a run stays inside one DPP row
        at[e] = cursor;
        cursor += n;
      }
      return (int)((cursor + 63) / 64);
    };
    std::vector<int> early(num_filts, 0);  // bins a one-segment filter starts early (zero weights in front)
    auto first_bin = [&](int f, int k) { return std::min(span(f) ? lo(f) + k * T : 0, nbp - T) - early[f]; };
    auto fill = [&](int rounds) {
      meta.assign((size_t)rounds * 64, -1);
      for (int f = 0; f < num_filts; ++f)
        for (int k = 0; k < nseg[f]; ++k) {
          int32_t m = first_bin(f, k);
          if (k + 1 < nseg[f]) m |= 1 << 14;
          if (k + 2 < nseg[f]) m |= 1 << 15;
          if (k == 0) m |= (f + 1) << 16;
          meta[at[f] + k] = m;
        }
      // a lane without a segment reads where a lane of its read group reads anyway (one broadcast)
      for (size_t l = 0; l < meta.size(); ++l) {
        if (meta[l] >= 0) continue;
        int32_t same = 0;
        for (size_t o = l / 32 * 32; o < l / 32 * 32 + 32; ++o)
          if (meta[o] >= 0 && rseg_read_group((int)(o % 32)) == rseg_read_group((int)(l % 32))) same = meta[o] & 0x3fff;
        meta[l] = same;
      }
    };
    auto cycles = [&](int rounds) {
      long c = 0;
      for (int rd = 0; rd < rounds; ++rd) c += rseg_read_cycles(meta.data() + (size_t)rd * 64, T);
      return c;
    };
    std::vector<int> order(num_filts);
    for (int f = 0; f < num_filts; ++f) order[f] = f;
    const int rounds = place(order);
    if (rounds > max_rounds) return 0;
    fill(rounds);
    if (search && num_filts > 1) {
      // Which lanes the runs sit on is free (a filter's coefficient goes where its number says, not
      // where its lane is): swap runs -- and spare lanes -- while that lowers the conflict cycles of
      // the power reads.  The natural order puts neighbouring filters, whose segments start in the
      // same bank quartets, into one read group (40 mel filters at N = 512: 120 cycles as numbered,
      // about 60 after the search, 48 without any conflict).  Deterministic: a fixed linear
      // congruential sequence of swaps.
      long used = 0;
      for (int f = 0; f < num_filts; ++f) used += nseg[f];
      for (long spare = used; spare < (long)rounds * 64 && place(order) <= rounds; ++spare) order.push_back((int)order.size());
      while (place(order) > rounds) order.pop_back();
      fill(rounds);
    }
    uint32_t state = 12345u;
    auto rnd = [&](int n) {
      state = state * 1664525u + 1013904223u;
      return (int)((state >> 8) % (uint32_t)n);
    };
    auto swap_search = [&](int tries) {
      long best = cycles(rounds);
      const int count = (int)order.size();
      for (int it = 0; it < tries && best > 4L * T * rounds; ++it) {
        const int a_ = rnd(count), b_ = rnd(count);
        if (a_ == b_ || (order[a_] >= num_filts && order[b_] >= num_filts)) continue;
        std::swap(order[a_], order[b_]);
        if (place(order) <= rounds) {
          fill(rounds);
          const long c = cycles(rounds);
          if (c <= best) {
            best = c;
            continue;
          }
        }
        std::swap(order[a_], order[b_]);
      }
      place(order);
      fill(rounds);
    };
    // a one-segment filter whose span is shorter than T may start early (zero weights in front): it
    // takes the start whose quartets collide least with the other lanes of its round
    auto early_starts = [&]() {
      for (int f = 0; f < num_filts; ++f) {
        if (nseg[f] != 1 || span(f) == 0) continue;
        const long pos = at[f];
        const int32_t flags = meta[pos] & ~0x3fff;
        early[f] = 0;
        const int highest = first_bin(f, 0), lowest = std::max(0, lo(f) + span(f) - T);
        long best_c = -1;
        int best_first = highest;
        for (int cand = highest; cand >= lowest; --cand) {
          meta[pos] = flags | cand;
          const long c = rseg_read_cycles(meta.data() + pos / 64 * 64, T);
          if (best_c < 0 || c < best_c) best_c = c, best_first = cand;
        }
        meta[pos] = flags | best_first;
        early[f] = highest - best_first;
      }
    };
    early_starts();
    if (search && num_filts > 1) {
      swap_search(4000);  // (with the early starts of the numbered order in place)
      early_starts();
      swap_search(2000);
      early_starts();
    }
    return rounds;
  };
  // price every segment length as numbered, then look for better lane assignments at the two cheapest
  auto price = [&](int len, const std::vector<int32_t> &meta, int rounds) {
    // LDS cycles per item: power reads as laid out, weight reads (conflict-free, 4 cycles per 16 bytes),
    // and the round's epilogue (sums, logs, four predicated stores) priced in the same unit
    long cost = 0;
    for (int rd = 0; rd < rounds; ++rd) cost += rseg_read_cycles(meta.data() + (size_t)rd * 64, len) + len + 40;
    return cost;
  };
  long best_cost = -1;
  int best_len = 0;
  std::vector<int32_t> meta, best_meta;
  std::vector<std::pair<long, int>> priced;
  for (int len = 4; len <= 64 && len <= nbp; len += 4) {
    const int rounds = layout(len, false, meta);
    if (rounds) priced.push_back({price(len, meta, rounds), len});
  }
  std::sort(priced.begin(), priced.end());
  for (size_t i = 0; i < priced.size() && i < 2; ++i) {
#ifdef PDS_RSEG_NO_SEARCH  // (measurement: the numbered lane assignment)
    const bool search = false;
#else
    const bool search = !numbered;
#endif
    const int len = priced[i].second, rounds = layout(len, search, meta);
    const long cost = price(len, meta, rounds);
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
