// Tables of the matrix-pipe filter walk of the fused STFT kernel (stft_fast.hip, template parameter
// SEG = 2; 16-lane geometries: four frames per wave; dense banks such as the 64 gammatone filters at
// 48 kHz, 7000 taps).
//
// v_mfma_f32_4x4x1_16B_f32 multiplies, in each of its 16 blocks, a 4 x 1 column by a 1 x 4 row and adds the
// 4 x 4 product to the block's accumulator: with the four FILTERS of a quad as the column and one bin of
// the four FRAMES of the wave as the row, one instruction applies 16 (quad, bin) pairs to all four frames
// -- 256 multiply-adds on the matrix pipe, in exact float32 arithmetic, for one weight and one power
// value per lane.  Filters are taken in quads of four neighbours (their supports overlap almost
// entirely in the dense banks); a quad's bin range, from a multiple of 4 bins on, is cut into UNITS of
// T bins.  Lane 4 b + i supplies the weight of filter i of block b's quad (A operand), lane 4 b + j the
// power of frame j (B operand), and lane 4 b + j receives rows i = 0..3 of the block's sums for frame j.
//
// Round 3: the units, in quad order, are dealt to the blocks in CONTIGUOUS runs of `rounds` units (block b
// takes units b * rounds ... ), so a block's consecutive rounds mostly belong to one quad and its sums stay in
// the instruction's accumulator across rounds; they leave for a partial-sum slot behind the power spectra only
// where the block's next unit belongs to another quad (or the block ends): a flush, flagged in the round's
// table entry.  A quad's flushes are consecutive slots, which the filter's lane adds up.  (Round 2 dealt the
// units round-robin and stored every unit's sums: 80 slots = 5 KB per wave for the 64 gammatone filters at
// 48 kHz, which with the 38 KB weight table kept that workload at eight waves per CU; now ~31 slots = 2 KB.)
//
// Host code only (plain C++): included by stft_fast.hip and by tests/csrc/test_mseg_tables.cpp.
#pragma once
#include <algorithm>
#include <cstdint>
#include <vector>

namespace pds {

struct MsegTables {
  int seg_len = 0;  // T: 16 or 32
  int rounds = 0;   // units per block
  int slots = 0;    // partial-sum slots (flushes)
  // meta[round * 16 + block] = first bin of the unit (a multiple of 4) | flush << 15 | slot << 16; then, per filter,
  // first partial slot * 4 + filter % 4 | slots << 16 (a filter's partials lie 4 entries apart: one slot each)
  std::vector<int32_t> meta;
  std::vector<float> w;  // [round][T / 4][lane = 4 block + filter % 4][4 bins]
  long reads_per_lane() const { return 2L * rounds * (seg_len / 4); }  // 16-byte LDS reads per item
  long mfmas() const { return (long)rounds * seg_len; }
};

// CSR filter table (cols ascending within a row); pstr: floats of a frame's power row in LDS (every read
// stays inside it); max_slots: partial-sum slots (16 floats each) that fit behind the power rows.
inline bool build_mseg(int num_filts, const int32_t *row_ptr, const int32_t *col, const double *val, int pstr,
                       int max_slots, MsegTables &out) {
  if (num_filts <= 0 || num_filts > 16383) return false;
  const int quads = (num_filts + 3) / 4;
  std::vector<int> qlo(quads, 0), qlen(quads, 0);
  for (int q = 0; q < quads; ++q) {
    int lo = 1 << 30, hi = -1;
    for (int f = 4 * q; f < std::min(num_filts, 4 * q + 4); ++f) {
      if (row_ptr[f + 1] == row_ptr[f]) continue;
      lo = std::min(lo, col[row_ptr[f]] & ~3);
      hi = std::max(hi, col[row_ptr[f + 1] - 1]);
    }
    if (hi >= 0) qlo[q] = lo, qlen[q] = hi - lo + 1;
  }
  // the quad of every unit, in quad order, for a unit length; slots a dealing in runs of `rounds` needs
  auto units_of = [&](int len, std::vector<int> &uq) {
    uq.clear();
    for (int q = 0; q < quads; ++q)
      for (int k = 0; k < std::max(1, (qlen[q] + len - 1) / len); ++k) uq.push_back(q);
  };
  auto count_slots = [&](const std::vector<int> &uq, int rounds) {
    int slots = 0;
    for (size_t u = 0; u < uq.size(); ++u)
      if (u + 1 == uq.size() || (int)((u + 1) % rounds) == 0 || uq[u + 1] != uq[u]) ++slots;
    return slots;
  };
  int best_len = 0, best_rounds = 0;
  long best_cost = -1;
  std::vector<int> uq;
  for (int len : {16, 32}) {  // (the kernel holds a unit's operands in registers: 32 bins at most)
    if (len > pstr) continue;
    units_of(len, uq);
    const int rounds = (int)((uq.size() + 15) / 16);
    if (rounds == 0 || count_slots(uq, rounds) > max_slots || count_slots(uq, rounds) > 65535) continue;
    // matrix instructions, plus a round's exposed LDS round trip and epilogue priced alike (measured on the
    // 64 gammatone filters at N = 1024: five rounds of 32 bins 0.348 ms, nine rounds of 16 bins 0.363 ms)
    const long cost = (long)rounds * len + 8L * rounds;
    if (best_cost < 0 || cost < best_cost) best_cost = cost, best_len = len, best_rounds = rounds;
  }
  if (best_cost < 0) return false;
  const int T = best_len, R = best_rounds;
  units_of(T, uq);
  const int U = (int)uq.size();
  out.seg_len = T;
  out.rounds = R;
  out.meta.assign((size_t)R * 16 + num_filts, 0);
  out.w.assign((size_t)R * T * 64, 0.0f);
  std::vector<int> qfirst(quads, -1), qcount(quads, 0);  // a quad's first slot and slot count
  int slot = 0, k = 0;  // k: index of the unit inside its quad
  for (int u = 0; u < U; ++u) {
    const int q = uq[u];
    k = (u > 0 && uq[u - 1] == q) ? k + 1 : 0;
    const int b = u / R, rd = u % R;
    const int base = qlo[q] + k * T;                  // bins [base, base + T) of the quad belong to this unit
    const int first = std::min(base, (pstr - T) & ~3);  // keep every read inside the power row
    if (first < 0 || first >= (1 << 15)) return false;
    const bool flush = u + 1 == U || (u + 1) % R == 0 || uq[u + 1] != q;
    out.meta[(size_t)rd * 16 + b] = first | (flush ? 1 << 15 : 0) | (slot << 16);
    for (int f = 4 * q; f < std::min(num_filts, 4 * q + 4); ++f)
      for (int at = row_ptr[f]; at < row_ptr[f + 1]; ++at) {
        if (col[at] < base || col[at] >= base + T) continue;
        const int t = col[at] - first;
        if (t < 0 || t >= T) return false;
        out.w[(((size_t)rd * (T / 4) + t / 4) * 64 + 4 * b + f % 4) * 4 + t % 4] = (float)val[at];
      }
    if (qfirst[q] < 0) qfirst[q] = slot;
    if (flush) {
      ++qcount[q];
      ++slot;
    }
  }
  out.slots = slot;
  // rounds of a block past its last unit (the last block only): weights zero, no flush, a harmless read
  for (int u = U; u < R * 16; ++u) out.meta[(size_t)(u % R) * 16 + u / R] = out.meta[(size_t)((U - 1) % R) * 16 + (U - 1) / R] & 0x7fff;
  for (int f = 0; f < num_filts; ++f) {
    const int q = f / 4;
    out.meta[(size_t)R * 16 + f] = (qfirst[q] * 4 + f % 4) | (qcount[q] << 16);
  }
  return true;
}

}  // namespace pds
