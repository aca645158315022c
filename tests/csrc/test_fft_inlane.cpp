// Host-side check of the in-lane FFT templates (pydrobert-speech_amd/csrc/fft_inlane.h)
// against a naive float64 DFT.  Built and run by tests/test_fft_inlane.py (CPU only).
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../pydrobert-speech_amd/csrc/fft_inlane.h"

using namespace pds::inl;

static double g_worst = 0.0;

static double frand() { return 2.0 * rand() / RAND_MAX - 1.0; }

template <int N>
static void check_cfft() {
  float xr[N], xi[N], yr[N], yi[N];
  for (int i = 0; i < N; ++i) { xr[i] = (float)frand(); xi[i] = (float)frand(); }
  CFFT<N, 1>::run(xr, xi, yr, yi);
  double worst = 0;
  for (int k = 0; k < N; ++k) {
    double re = 0, im = 0;
    for (int n = 0; n < N; ++n) {
      const double a = -2.0 * M_PI * n * k / N;
      re += xr[n] * cos(a) - xi[n] * sin(a);
      im += xr[n] * sin(a) + xi[n] * cos(a);
    }
    worst = fmax(worst, fmax(fabs(re - yr[k]), fabs(im - yi[k])));
  }
  printf("cfft<%d> max abs err %.3g\n", N, worst);
  g_worst = fmax(g_worst, worst / sqrt((double)N));
}

template <int M>
static void check_rdft() {
  constexpr int H = M / 2;
  float a[M], Ar[H + 1] = {0}, Ai[H + 1] = {0}, ev, od;
  for (int i = 0; i < M; ++i) a[i] = (float)frand();
  rdft_scaled<M>(a, ev, od, Ar, Ai);
  std::vector<double> re(H + 1), im(H + 1);
  for (int k = 0; k <= H; ++k) {
    re[k] = im[k] = 0;
    for (int n = 0; n < M; ++n) {
      re[k] += a[n] * cos(2.0 * M_PI * n * k / M);
      im[k] -= a[n] * sin(2.0 * M_PI * n * k / M);
    }
  }
  double worst = fmax(fabs(re[0] - (ev + od)), fabs(re[H] - (ev - od)));
  for (int k = 1; k < H; ++k) {
    const double s = (2 * k == H) ? 1.0 : 0.5;
    worst = fmax(worst, fmax(fabs(re[k] - s * Ar[k]), fabs(im[k] - s * Ai[k])));
  }
  // second form: FFT of the packed sequence, then power of every bin
  float zr[H], zi[H], Yr[H], Yi[H];
  for (int m = 0; m < H; ++m) { zr[m] = a[2 * m]; zi[m] = a[2 * m + 1]; }
  CFFT<H, 1>::run(zr, zi, Yr, Yi);
  double pw[H + 1];
  rdft_finish_power<M>(Yr, Yi, [&](auto kk, float r, float i) { pw[decltype(kk)::value] = (double)r * r + (double)i * i; });
  double worst_p = 0;
  for (int k = 0; k <= H; ++k) worst_p = fmax(worst_p, fabs(pw[k] - (re[k] * re[k] + im[k] * im[k])) / M);
  printf("rdft<%d> max abs err %.3g, power err/M %.3g\n", M, worst, worst_p);
  g_worst = fmax(g_worst, fmax(worst, worst_p) / sqrt((double)M));
}

// real decimation in time: unscaled outputs, same interface
template <int M>
static void check_rdft_dit() {
  constexpr int H = M / 2;
  float a[M], Ar[H + 1] = {0}, Ai[H + 1] = {0}, ev, od;
  for (int i = 0; i < M; ++i) a[i] = (float)frand();
  rdft_dit<M>(a, ev, od, Ar, Ai);
  double worst = 0, evs = 0, ods = 0;
  for (int n = 0; n < M; ++n) (n % 2 ? ods : evs) += a[n];
  worst = fmax(fabs(evs - ev), fabs(ods - od));
  for (int k = 1; k < H; ++k) {
    double re = 0, im = 0;
    for (int n = 0; n < M; ++n) {
      re += a[n] * cos(2.0 * M_PI * n * k / M);
      im -= a[n] * sin(2.0 * M_PI * n * k / M);
    }
    worst = fmax(worst, fmax(fabs(re - Ar[k]), fabs(im - Ai[k])));
  }
  printf("rdft_dit<%d> max abs err %.3g\n", M, worst);
  g_worst = fmax(g_worst, worst / sqrt((double)M));
}

// sizes that are not powers of two: unscaled outputs k = 1 .. (M - 1) / 2, and the column sums
template <int M>
static void check_rdft_direct() {
  constexpr int J = (M - 1) / 2;
  float a[M], Ar[J + 1] = {0}, Ai[J + 1] = {0}, ev, od;
  for (int i = 0; i < M; ++i) a[i] = (float)frand();
  rdft_direct<M>(a, ev, od, Ar, Ai);
  double worst = 0, tot = 0, alt = 0;
  for (int n = 0; n < M; ++n) {
    tot += a[n];
    alt += (n % 2 ? -1.0 : 1.0) * a[n];
  }
  worst = fabs(tot - (ev + od));
  if (M % 2 == 0) worst = fmax(worst, fabs(alt - (ev - od)));
  else worst = fmax(worst, fabs((double)od));
  for (int k = 1; k <= J; ++k) {
    double re = 0, im = 0;
    for (int n = 0; n < M; ++n) {
      re += a[n] * cos(2.0 * M_PI * n * k / M);
      im -= a[n] * sin(2.0 * M_PI * n * k / M);
    }
    worst = fmax(worst, fmax(fabs(re - Ar[k]), fabs(im - Ai[k])));
  }
  printf("rdft_direct<%d> max abs err %.3g\n", M, worst);
  g_worst = fmax(g_worst, worst / sqrt((double)M));
}

int main() {
  srand(12345);
  check_cfft<1>(); check_cfft<2>(); check_cfft<4>(); check_cfft<8>(); check_cfft<16>();
  check_cfft<32>(); check_cfft<64>(); check_cfft<128>();
  check_rdft<4>(); check_rdft<8>(); check_rdft<16>(); check_rdft<32>(); check_rdft<64>(); check_rdft<128>();
  check_rdft_dit<8>(); check_rdft_dit<16>(); check_rdft_dit<32>(); check_rdft_dit<64>(); check_rdft_dit<128>();
  check_rdft_direct<10>(); check_rdft_direct<15>(); check_rdft_direct<20>(); check_rdft_direct<25>();
  check_rdft_direct<30>(); check_rdft_direct<12>(); check_rdft_direct<50>();
  printf("worst normalised error %.3g\n", g_worst);
  return g_worst < 2e-6 ? 0 : 1;
}
