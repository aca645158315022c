// Host replay of the per-item twiddle regeneration of the prefetch instantiations
// (pydrobert-speech_amd/csrc/fft_inlane.h::twiddle_chain15) against float64 twiddles, for every lane of
// the 32 x 16 geometry (N = 512) and, as the kernel uses them, with seeds rounded to float32.
// Built and run by tests/test_twiddle_chain.py (CPU only).
#include <cmath>
#include <cstdio>

#include "../../pydrobert-speech_amd/csrc/fft_inlane.h"

int main() {
  double worst = 0.0;
  for (int N : {512}) {
    for (int r = 0; r < 16; ++r) {
      float seed[6];
      for (int j = 0; j < 3; ++j) {
        const double ang = -2.0 * M_PI * (double)((r * (j == 0 ? 1 : 4 * j)) % N) / (double)N;
        seed[2 * j] = (float)std::cos(ang);
        seed[2 * j + 1] = (float)std::sin(ang);
      }
      float tr[16], ti[16];
      pds::inl::twiddle_chain15(seed[0], seed[1], seed[2], seed[3], seed[4], seed[5], tr, ti);
      for (int k = 1; k <= 15; ++k) {
        const double ang = -2.0 * M_PI * (double)(r * k) / (double)N, s = k == 8 ? 2.0 : 1.0;
        const double er = std::fabs(tr[k] - s * std::cos(ang)) / s, ei = std::fabs(ti[k] - s * std::sin(ang)) / s;
        worst = std::fmax(worst, std::fmax(er, ei));
      }
    }
  }
  std::printf("worst twiddle error %.3g\n", worst);
  return worst <= 2e-7 ? 0 : 1;
}
