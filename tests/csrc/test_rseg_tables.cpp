// Host replay of the row-segment filter walk's tables (pydrobert-speech_amd/csrc/rseg_tables.h):
// lanes read their segment's weights and the bin-major power rows exactly as the kernel does,
// partial sums are combined with the two shift-and-add steps, and the result is compared with the
// CSR product in float64.  Built and run by tests/test_rseg_tables.py (CPU only).
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../pydrobert-speech_amd/csrc/rseg_tables.h"

static double frand() { return (double)rand() / RAND_MAX; }

static double check(const char *name, int F, const std::vector<int> &start, const std::vector<int> &len, int num_bins) {
  std::vector<int32_t> row_ptr(F + 1, 0), col;
  std::vector<double> val;
  for (int f = 0; f < F; ++f) {
    for (int t = 0; t < len[f]; ++t) {
      col.push_back(start[f] + t);
      val.push_back(0.05 + frand());
    }
    row_ptr[f + 1] = (int32_t)col.size();
  }
  pds::RsegTables rs;
  if (!pds::build_rseg(F, row_ptr.data(), col.data(), val.data(), num_bins, 576, 8, rs)) {
    printf("%s: not served\n", name);
    return name[0] == 'r' ? -1.0 : 1e9;  // (random banks may have rows no segment length serves)
  }
  const int T = rs.seg_len;
  // P[bin][frame]; bins >= num_bins up to nbp are zero in the kernel, the dump slot (bin nbp) is never read
  std::vector<float> P((size_t)(rs.nbp + 1) * 4, 0.0f);
  for (int b = 0; b < num_bins; ++b)
    for (int g = 0; g < 4; ++g) P[(size_t)b * 4 + g] = (float)(1000.0 * frand());
  for (int g = 0; g < 4; ++g) P[(size_t)rs.nbp * 4 + g] = NAN;
  std::vector<double> got((size_t)F * 4, -1.0);
  std::vector<int> seen(F, 0);
  for (int rd = 0; rd < rs.rounds; ++rd) {
    float acc[64][4];
    for (int l = 0; l < 64; ++l) {
      const int m = rs.meta[(size_t)rd * 64 + l], first = m & 0x3fff;
      for (int g = 0; g < 4; ++g) acc[l][g] = 0.0f;
      if (first + T > rs.nbp) return 1e9;
      for (int t = 0; t < T; ++t) {
        const float w = rs.w[(((size_t)rd * (T / 4) + t / 4) * 64 + l) * 4 + t % 4];
        for (int g = 0; g < 4; ++g) acc[l][g] = fmaf(w, P[(size_t)(first + t) * 4 + g], acc[l][g]);
      }
    }
    // DPP row_shl:1 / row_shl:2 with zero fill at the row's end
    for (int step = 1; step <= 2; ++step) {
      float nxt[64][4];
      for (int l = 0; l < 64; ++l) {
        const bool cont = (rs.meta[(size_t)rd * 64 + l] >> (13 + step)) & 1;
        const int src = l + step;
        for (int g = 0; g < 4; ++g) {
          const float other = (src / 16 == l / 16) ? acc[src][g] : 0.0f;
          nxt[l][g] = acc[l][g] + (cont ? other : 0.0f);
        }
      }
      for (int l = 0; l < 64; ++l)
        for (int g = 0; g < 4; ++g) acc[l][g] = nxt[l][g];
    }
    for (int l = 0; l < 64; ++l) {
      const int f = (rs.meta[(size_t)rd * 64 + l] >> 16) - 1;
      if (f < 0) continue;
      ++seen[f];
      for (int g = 0; g < 4; ++g) got[(size_t)f * 4 + g] = acc[l][g];
    }
  }
  double worst = 0;
  for (int f = 0; f < F; ++f) {
    if (seen[f] != 1) return 1e9;
    for (int g = 0; g < 4; ++g) {
      double want = 0;
      for (int q = row_ptr[f]; q < row_ptr[f + 1]; ++q) want += val[q] * P[(size_t)col[q] * 4 + g];
      worst = fmax(worst, fabs(got[(size_t)f * 4 + g] - want) / (fabs(want) + 1.0));
    }
  }
  printf("%-28s F %3d nnz %5zu: T %2d rounds %d reads/lane %ld, worst rel err %.3g\n", name, F, col.size(), T,
         rs.rounds, rs.reads_per_lane(), worst);
  return worst;
}

int main() {
  srand(777);
  double worst = 0;
  {  // the 40-filter Fbank at N = 512 (C1 offsets and lengths)
    const std::vector<int> st = {1, 3, 4, 6, 7, 9, 11, 13, 16, 18, 20, 23, 26, 29, 32, 35, 39, 43, 47, 51,
                                 56, 61, 66, 71, 77, 83, 90, 97, 104, 112, 121, 130, 139, 149, 160, 171, 184, 196, 210, 225};
    const std::vector<int> ln = {3, 3, 3, 3, 4, 4, 5, 5, 4, 5, 6, 6, 6, 6, 7, 8, 8, 8, 9, 10,
                                 10, 10, 11, 12, 13, 14, 14, 15, 17, 18, 18, 19, 21, 22, 24, 25, 26, 29, 30, 32};
    worst = fmax(worst, check("fbank40 @512", 40, st, ln, 257));
  }
  {  // the 64-filter Gabor bank at N = 512 (BASELINE.json configs[3]): dense rows, one long segment per lane
    const std::vector<int> st = {0, 1, 2, 3, 4, 4, 6, 7, 8, 9, 10, 11, 13, 14, 15, 17, 18, 20, 22, 23, 25, 27,
                                 29, 31, 33, 35, 37, 39, 42, 44, 47, 50, 52, 55, 58, 61, 65, 68, 72, 75, 79, 83,
                                 87, 91, 96, 100, 105, 110, 115, 121, 126, 132, 138, 144, 151, 158, 165, 172, 179,
                                 187, 195, 204, 213, 222};
    const std::vector<int> ln = {4, 4, 4, 4, 5, 6, 5, 5, 6, 6, 6, 7, 6, 7, 8, 7, 8, 8, 8, 9, 9, 9, 10, 10, 11, 11,
                                 12, 13, 12, 13, 14, 14, 15, 16, 16, 17, 17, 18, 18, 19, 20, 21, 22, 23, 23, 25, 25,
                                 26, 27, 28, 30, 30, 32, 33, 34, 35, 36, 38, 40, 42, 43, 45, 44, 35};
    worst = fmax(worst, check("gabor64 @512", 64, st, ln, 257));
  }
  for (int trial = 0; trial < 12; ++trial) {
    const int bins = (trial % 3 == 0) ? 257 : (trial % 3 == 1 ? 513 : 201);
    const int F = 1 + rand() % 120;
    std::vector<int> st(F), ln(F);
    const int maxlen = 1 + rand() % 60;
    for (int f = 0; f < F; ++f) {
      ln[f] = (rand() % 7 == 0) ? 0 : 1 + rand() % maxlen;
      st[f] = rand() % (bins - ln[f] + 1);
    }
    char name[64];
    snprintf(name, sizeof name, "random bank %d (%d bins)", trial, bins);
    const double e = check(name, F, st, ln, bins);
    if (e >= 0) worst = fmax(worst, e);
  }
  printf("worst normalised error %.3g\n", worst);
  return worst < 1e-5 ? 0 : 1;
}
