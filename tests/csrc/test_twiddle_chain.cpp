// Host replay of the per-item twiddle regeneration (pydrobert-speech_amd/csrc/fft_inlane.h::twiddle_chain) against
// float64 twiddles, for every lane of the 32 x 16 (N = 512) and 64 x 16 (N = 1024) geometries and, as the kernel
// uses them, with seeds rounded to float32.  Built and run by tests/test_twiddle_chain.py (CPU only).
#include <cmath>
#include <cstdio>

#include "../../pydrobert-speech_amd/csrc/fft_inlane.h"

template <int K>
static double check(int N) {
  constexpr int Q = (K + 1) / 2;
  double worst = 0.0;
  for (int r = 0; r < 16; ++r) {
    float seed[6];
    const int mult[3] = {1, 4, Q};
    for (int j = 0; j < 3; ++j) {
      const double ang = -2.0 * M_PI * (double)((r * mult[j]) % N) / (double)N;
      seed[2 * j] = (float)std::cos(ang);
      seed[2 * j + 1] = (float)std::sin(ang);
    }
    float tr[K + 1], ti[K + 1];
    pds::inl::twiddle_chain<K>(seed[0], seed[1], seed[2], seed[3], seed[4], seed[5], tr, ti);
    for (int k = 1; k <= K; ++k) {
      const double ang = -2.0 * M_PI * (double)(r * k) / (double)N, s = k == Q ? 2.0 : 1.0;
      const double er = std::fabs(tr[k] - s * std::cos(ang)) / s, ei = std::fabs(ti[k] - s * std::sin(ang)) / s;
      worst = std::fmax(worst, std::fmax(er, ei));
    }
  }
  std::printf("N = %d: worst twiddle error %.3g\n", N, worst);
  return worst;
}

int main() {
  const double a = check<15>(512), b = check<31>(1024);
  return (a <= 2e-7 && b <= 3e-7) ? 0 : 1;
}
