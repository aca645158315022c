// Host replay of the matrix-pipe front end of the fused STFT kernel
// (pydrobert-speech_amd/csrc/mfma_front.h): the tables the plan uploads, driven exactly as
// the kernel drives them -- per-lane loads, window, pair sums / differences, the two
// accumulation chains with v_mfma_f32_16x16x4_f32's operand maps (A[i = l & 15][k = l >> 4],
// B[k = l >> 4][j = l & 15], D row 4 (l >> 4) + v, column l & 15), twiddle -- against a
// float64 DFT.  Built and run by tests/test_mfma_front.py (CPU only).
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../pydrobert-speech_amd/csrc/mfma_front.h"

static double frand() { return 2.0 * rand() / RAND_MAX - 1.0; }

// D += A x B for one instruction: a[l], b[l] per-lane operands, acc[l][v] per-lane results
static void mfma16x16x4(const float *a, const float *b, float acc[64][4]) {
  for (int l = 0; l < 64; ++l)
    for (int v = 0; v < 4; ++v) {
      const int i = 4 * (l >> 4) + v, j = l & 15;
      float sum = acc[l][v];
      for (int k = 0; k < 4; ++k) sum = fmaf(a[16 * k + i], b[16 * k + j], sum);  // k-ordered fmaf chain
      acc[l][v] = sum;
    }
}

static double check(int rows_bucket, int L) {
  const int n1 = 32, N = 512;
  std::vector<double> window(L);
  for (int i = 0; i < L; ++i) window[i] = (0.5 - 0.5 * cos(2.0 * M_PI * i / (L - 1))) / (0.5 * (L - 1));
  pds::MfmaFrontTables t;
  if (!pds::build_mfma_front(n1, rows_bucket, L, window.data(), t)) {
    printf("rows %d L %d: not served\n", rows_bucket, L);
    return 1e9;
  }
  const int tp = t.tp;
  std::vector<float> x(rows_bucket * 16 + 64);
  for (auto &v : x) v = (float)(3000.0 * frand());
  // samples past the frame: garbage the window must remove
  for (size_t i = L; i < x.size(); ++i) x[i] = 1e30f;
  float accR[64][4] = {}, accI[64][4] = {};
  std::vector<float> s(64), d(64), u(64, 0.0f), sc(64);
  double energy = 0;
  for (int st = 0; st < tp; ++st) {
    for (int l = 0; l < 64; ++l) {
      const float xp = x[t.off[(2 * st) * 64 + l]], xm = x[t.off[(2 * st + 1) * 64 + l]];
      const float wp = t.win[(2 * st) * 64 + l], wm = t.win[(2 * st + 1) * 64 + l];
      const float a = wp == 0.0f ? 0.0f : xp * wp, b = wm == 0.0f ? 0.0f : xm * wm;  // v_mul_legacy
      s[l] = a + b;
      d[l] = a - b;
      u[l] += s[l];
      energy += (double)(xp * t.emask[(2 * st) * 64 + l]) * (xp * t.emask[(2 * st) * 64 + l]);
      energy += (double)(xm * t.emask[(2 * st + 1) * 64 + l]) * (xm * t.emask[(2 * st + 1) * 64 + l]);
    }
    mfma16x16x4(&t.a_re[st * 64], s.data(), accR);
    mfma16x16x4(&t.a_im[st * 64], d.data(), accI);
  }
  for (int l = 0; l < 64; ++l) {
    const float xc = x[t.off[(2 * tp) * 64 + l]], wc = t.win[(2 * tp) * 64 + l];
    sc[l] = wc == 0.0f ? 0.0f : xc * wc;
    u[l] += sc[l];
    energy += (double)(xc * t.emask[(2 * tp) * 64 + l]) * (xc * t.emask[(2 * tp) * 64 + l]);
  }
  mfma16x16x4(&t.a_re[tp * 64], sc.data(), accR);
  mfma16x16x4(&t.a_im[tp * 64], u.data(), accI);
  // reference
  double worst = 0, scale = 0, want_energy = 0;
  for (int i = 0; i < L; ++i) want_energy += (double)x[i] * x[i];
  for (int r = 0; r < 16; ++r) {
    double ev = 0, od = 0;
    for (int n = 0; n < n1; ++n) {
      const int idx = 16 * n + r;
      const double xw = idx < L ? (double)x[idx] * window[idx] : 0.0;
      (n % 2 ? od : ev) += xw;
    }
    for (int k1 = 0; k1 <= 16; ++k1) {
      double re = 0, im = 0;
      for (int n = 0; n < n1; ++n) {
        const int idx = 16 * n + r;
        if (idx >= L) continue;
        const double xw = (double)x[idx] * window[idx];
        const double a = -2.0 * M_PI * ((double)n * k1 / n1 + (double)r * k1 / N);
        re += xw * cos(a);
        im += xw * sin(a);
      }
      scale = fmax(scale, hypot(re, im));
      if (k1 == 0 || k1 == 16) continue;
      // lane holding output row k1 - 1 of residue r
      const int q = (k1 - 1) >> 2, v = (k1 - 1) & 3, l = 16 * q + r;
      const float tr = t.tw[(v * 64 + l) * 2], ti = t.tw[(v * 64 + l) * 2 + 1];
      const double gr = (double)accR[l][v] * tr - (double)accI[l][v] * ti;
      const double gi = (double)accR[l][v] * ti + (double)accI[l][v] * tr;
      worst = fmax(worst, fmax(fabs(gr - re), fabs(gi - im)));
    }
    const int l = 48 + r;
    worst = fmax(worst, fmax(fabs(accR[l][3] - ev), fabs(accI[l][3] - od)));
  }
  const double eerr = fabs(energy - want_energy) / want_energy;
  printf("rows %2d L %3d tp %d: max abs err %.3g (scale %.3g) energy rel err %.3g\n", rows_bucket, L, tp, worst,
         scale, eerr);
  return fmax(worst / scale, eerr);
}

int main() {
  srand(12345);
  double worst = 0;
  const int cases[][2] = {{25, 400}, {25, 390}, {25, 385}, {20, 320}, {20, 310}, {28, 441}, {28, 448},
                          {30, 480}, {32, 512}, {32, 500}, {20, 272}, {25, 321}};
  for (const auto &c : cases) worst = fmax(worst, check(c[0], c[1]));
  printf("worst normalised error %.3g\n", worst);
  return worst < 2e-6 ? 0 : 1;
}
