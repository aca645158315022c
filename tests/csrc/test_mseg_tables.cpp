// Host replay of the matrix-pipe segment walk's tables (pydrobert-speech_amd/csrc/mseg_tables.h): every
// block of v_mfma_f32_4x4x1_16B_f32 is emulated as the 4 x 1 by 1 x 4 outer product it is -- lane 4 b + i
// supplies the weight of filter i of the block's quad, lane 4 b + j the power of frame j of the frame-major
// power rows -- a block's sums stay in its accumulator across rounds and go to partial slot `slot` where the round's
// table entry says flush, a filter adds up its quad's slots, and the result is compared with the CSR product in
// float64.  Built and run by tests/test_mseg_tables.py (CPU only).
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../pydrobert-speech_amd/csrc/mseg_tables.h"

static double frand() { return (double)rand() / RAND_MAX; }

static double check(const char *name, int F, const std::vector<int> &start, const std::vector<int> &len, int num_bins,
                    int max_units) {
  std::vector<int32_t> row_ptr(F + 1, 0), col;
  std::vector<double> val;
  for (int f = 0; f < F; ++f) {
    for (int t = 0; t < len[f]; ++t) {
      col.push_back(start[f] + t);
      val.push_back(0.05 + frand());
    }
    row_ptr[f + 1] = (int32_t)col.size();
  }
  const int pstr = ((num_bins + 1 + 15) / 32) * 32 + 16;  // WaveGeom::PSTR
  pds::MsegTables ms;
  if (!pds::build_mseg(F, row_ptr.data(), col.data(), val.data(), pstr, max_units, ms)) {
    printf("%s: not served\n", name);
    return -1.0;
  }
  const int T = ms.seg_len, R = ms.rounds;
  if (ms.slots > max_units || ms.slots <= 0 || (T != 16 && T != 32)) return 1e9;
  std::vector<float> P((size_t)4 * pstr, 0.0f);  // frame-major rows; beyond the bins: zeros (the kernel keeps them finite)
  for (int g = 0; g < 4; ++g)
    for (int b = 0; b < num_bins; ++b) P[(size_t)g * pstr + b] = (float)(1000.0 * frand());
  std::vector<float> part((size_t)ms.slots * 4 * 4, NAN);  // [slot][filter of the quad][frame]
  for (int b = 0; b < 16; ++b) {
    float acc[4][4] = {{0}};
    for (int rd = 0; rd < R; ++rd) {
      const int m = ms.meta[(size_t)rd * 16 + b];
      const int first = m & 0x7fff, slot = m >> 16;
      const bool flush = (m >> 15) & 1;
      if (first % 4 || first + T > pstr) return 1e9;
      for (int t = 0; t < T; ++t)
        for (int i = 0; i < 4; ++i) {
          const float w = ms.w[(((size_t)rd * (T / 4) + t / 4) * 64 + 4 * b + i) * 4 + t % 4];
          for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(w, P[(size_t)j * pstr + first + t], acc[i][j]);
        }
      if (flush) {
        if (slot < 0 || slot >= ms.slots) return 1e9;
        for (int i = 0; i < 4; ++i)
          for (int j = 0; j < 4; ++j) {
            if (part[((size_t)slot * 4 + i) * 4 + j] == part[((size_t)slot * 4 + i) * 4 + j]) return 1e9;  // written twice
            part[((size_t)slot * 4 + i) * 4 + j] = acc[i][j];
            acc[i][j] = 0.0f;
          }
      }
    }
  }
  double worst = 0.0;
  for (int f = 0; f < F; ++f) {
    const int fm = ms.meta[(size_t)R * 16 + f];
    for (int j = 0; j < 4; ++j) {
      double got = 0.0, want = 0.0;
      for (int k = 0; k < (fm >> 16); ++k) got += part[((size_t)(fm & 0xffff) + 4 * k) * 4 + j];
      for (int at = row_ptr[f]; at < row_ptr[f + 1]; ++at) want += val[at] * P[(size_t)j * pstr + col[at]];
      if (!(got == got)) return 1e9;
      worst = std::fmax(worst, std::fabs(got - want) / (1e-30 + std::fabs(want)));
    }
  }
  printf("%s: T %d rounds %d slots %d matrix instructions %ld reads %ld: worst relative error %.3g\n", name, T, R, ms.slots,
         ms.mfmas(), ms.reads_per_lane(), worst);
  return worst;
}

int main() {
  srand(7);
  double worst = 0.0;
  {  // a gammatone-like bank at N = 1024: supports growing from 7 to 300 bins
    const int F = 64, nb = 513;
    std::vector<int> start(F), len(F);
    for (int f = 0; f < F; ++f) {
      len[f] = 7 + f * f * 297 / (63 * 63);
      start[f] = std::min(f * 278 / 63, nb - len[f]);
    }
    worst = std::fmax(worst, check("gammatone-like 64 @ 1024", F, start, len, nb, 36));
  }
  {  // a Gabor-like bank at N = 512
    const int F = 64, nb = 257;
    std::vector<int> start(F), len(F);
    for (int f = 0; f < F; ++f) {
      len[f] = 4 + f * 41 / 63;
      start[f] = std::min(f * 222 / 63, nb - len[f]);
    }
    worst = std::fmax(worst, check("gabor-like 64 @ 512", F, start, len, nb, 76));
  }
  for (int trial = 0; trial < 40; ++trial) {  // random banks, filter counts that are no multiple of four, empty rows
    const int nb = (trial % 2) ? 257 : 513, F = 1 + rand() % 70;
    std::vector<int> start(F), len(F);
    for (int f = 0; f < F; ++f) {
      len[f] = (rand() % 9 == 0) ? 0 : 1 + rand() % (nb / 3);
      start[f] = rand() % (nb - len[f] + 1);
    }
    char name[32];
    snprintf(name, sizeof name, "random %d", trial);
    const double w = check(name, F, start, len, nb, nb == 257 ? 76 : 36);
    if (w >= 0) worst = std::fmax(worst, w);
  }
  printf("worst normalised error %.3g\n", worst);
  return worst < 2e-6 ? 0 : 1;
}
