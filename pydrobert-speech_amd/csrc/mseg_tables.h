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
// T bins; the units are dealt to the 16 blocks, `rounds` units each.  Lane 4 b + i supplies the weight
// of filter i of block b's quad (A operand), lane 4 b + j the power of frame j (B operand), and lane
// 4 b + j receives rows i = 0..3 of the block's sums for frame j.  A unit's four sums per frame go to
// the partial-sum area behind the power spectra; a filter's partials (one per unit of its quad, 4 slots
// apart) are added up by the filter's lane as in the segmented walk.
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
  // meta[round * 16 + block] = first bin of the unit (a multiple of 4); then, per filter,
  // first partial slot | units << 16 (a filter's partials lie 4 slots apart: slot = unit * 4 + filter % 4)
  std::vector<int32_t> meta;
  std::vector<float> w;  // [round][T / 4][lane = 4 block + filter % 4][4 bins]
  long reads_per_lane() const { return 2L * rounds * (seg_len / 4); }  // 16-byte LDS reads per item
  long mfmas() const { return (long)rounds * seg_len; }
};

// CSR filter table (cols ascending within a row); pstr: floats of a frame's power row in LDS (every read
// stays inside it); max_units: units whose partial sums fit behind the power rows.
inline bool build_mseg(int num_filts, const int32_t *row_ptr, const int32_t *col, const double *val, int pstr,
                       int max_units, MsegTables &out) {
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
  int best_len = 0, best_rounds = 0;
  long best_cost = -1;
  for (int len : {16, 32}) {  // (the kernel holds a unit's operands in registers: 32 bins at most)
    if (len > pstr) continue;
    long units = 0;
    for (int q = 0; q < quads; ++q) units += std::max(1, (qlen[q] + len - 1) / len);
    const int rounds = (int)((units + 15) / 16);
    if (rounds * 16 > max_units) continue;
    // matrix instructions, plus a round's exposed LDS round trip and epilogue priced alike (measured on the
    // 64 gammatone filters at N = 1024: five rounds of 32 bins 0.348 ms, nine rounds of 16 bins 0.363 ms)
    const long cost = (long)rounds * len + 8L * rounds;
    if (best_cost < 0 || cost < best_cost) best_cost = cost, best_len = len, best_rounds = rounds;
  }
  if (best_cost < 0) return false;
  const int T = best_len, R = best_rounds;
  out.seg_len = T;
  out.rounds = R;
  out.meta.assign((size_t)R * 16 + num_filts, 0);
  out.w.assign((size_t)R * T * 64, 0.0f);
  int unit = 0;
  for (int q = 0; q < quads; ++q) {
    const int count = std::max(1, (qlen[q] + T - 1) / T);
    for (int f = 4 * q; f < std::min(num_filts, 4 * q + 4); ++f)
      out.meta[(size_t)R * 16 + f] = (unit * 4 + f % 4) | (count << 16);
    for (int k = 0; k < count; ++k, ++unit) {
      const int base = qlo[q] + k * T;                  // bins [base, base + T) of the quad belong to this unit
      int first = std::min(base, (pstr - T) & ~3);      // keep every read inside the power row
      if (first < 0) return false;
      out.meta[unit] = first;
      const int rd = unit / 16, b = unit % 16;
      for (int f = 4 * q; f < std::min(num_filts, 4 * q + 4); ++f)
        for (int at = row_ptr[f]; at < row_ptr[f + 1]; ++at) {
          if (col[at] < base || col[at] >= base + T) continue;
          const int t = col[at] - first;
          if (t < 0 || t >= T) return false;
          out.w[(((size_t)rd * (T / 4) + t / 4) * 64 + 4 * b + f % 4) * 4 + t % 4] = (float)val[at];
        }
    }
  }
  // blocks without a unit in the last round read where block 0 of the round reads (weights zero)
  for (; unit < R * 16; ++unit) out.meta[unit] = out.meta[unit / 16 * 16];
  return true;
}

}  // namespace pds
