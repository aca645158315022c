// Tables of the matrix-pipe front end of the fused STFT kernel (stft_fast.hip, template
// parameter MF): window + the in-lane N1-point real DFT of the 16-lane geometries evaluated as a
// dense product on v_mfma_f32_16x16x4_f32, which runs beside the vector pipe.
//
// Geometry.  N = N1 * 16; a frame is R rows of 16 samples, row n1 = samples 16 n1 .. 16 n1 + 15.
// For every residue r the kernel needs A_r[k1] = sum_n1 xw[16 n1 + r] e^{-2 pi i n1 k1 / N1},
// k1 = 0 .. N1/2 (xw = windowed frame).  With the rows centred on c = R / 2,
//   A_r[k1] = e^{-2 pi i c k1 / N1} (Re'[k1] + i Im'[k1]),
//   Re'[k1] = xw_c + sum_j s[j] cos(2 pi j k1 / N1),   s[j] = xw[c + j] + xw[c - j],
//   Im'[k1] =      - sum_j d[j] sin(2 pi j k1 / N1),   d[j] = xw[c + j] - xw[c - j],  j = 1 .. c,
// so the real parts need c + 1 inputs and the imaginary parts c (instead of R each): two
// accumulation chains of 16 output rows per frame, TP + 1 = ceil(c / 4) + 1 instructions each.
//
// One MFMA tile = one frame: B[k][j] = input k of residue j, D[i][j] = output row i of residue j.
// Lane l = (q = l >> 4, r = l & 15) supplies B[k = q][j = r] and receives D[4 q + v][r], v = 0..3.
//   pair step t < TP : lane (q, r) holds the pair j = 1 + 4 t + q: s[j] for the real chain, d[j] for
//                      the imaginary chain (one pair of loads serves both)
//   last step        : real chain -- the centre sample xw_c in k-slot 1, zeros elsewhere;
//                      imaginary chain -- u(q) = the lane's own s values summed (+ xw_c in slot 1)
//   output rows      : i < 15 is k1 = i + 1 in both chains.  Row 15 of the real chain is
//                      even_sum = sum over even n1 of xw, row 15 of the imaginary chain is odd_sum
//                      (sum over odd n1): all rows a lane holds have the parity of c + 1 + q, the
//                      centre row that of c -- which is why the centre sits in slot q = 1 -- so both
//                      sums are 0/1 combinations of inputs the chains already have.
// The phase e^{-2 pi i c k1 / N1} is folded into the inter-stage twiddle the lane applies anyway.
//
// Host code only (plain C++): included by stft_fast.hip and by tests/csrc/test_mfma_front.cpp,
// which replays the kernel's data flow on the CPU against a float64 DFT.
#pragma once
#include <cmath>
#include <cstdint>
#include <vector>

namespace pds {

struct MfmaFrontTables {
  int tp = 0;       // pair steps
  int center = 0;   // c
  // all arrays are [entry][64 lanes]
  std::vector<float> win;     // [2 tp + 1]: plus / minus partner of step t at 2 t / 2 t + 1, centre last
  std::vector<int32_t> off;   // [2 tp + 1]: sample offset inside the frame (always < frame_length)
  std::vector<float> emask;   // [2 tp + 1]: 1 where the slot is a sample of the frame of its own
  std::vector<float> a_re;    // [tp + 1]: A operand of the real chain
  std::vector<float> a_im;    // [tp + 1]: A operand of the imaginary chain
  std::vector<float> tw;      // [4][64][2]: twiddle of output row 4 q + v (k1 = 4 q + v + 1 <= 15)

  // 32-bit words of the device image: win, off, emask, a_re, a_im, tw back to back
  size_t words() const { return win.size() + off.size() + emask.size() + a_re.size() + a_im.size() + tw.size(); }
};

// pair steps for a frame of `rows` rows (the kernel's template parameter)
constexpr int mfma_front_steps(int rows) { return (rows / 2 + 3) / 4; }

// n1: in-lane DFT size (the front end serves n1 = 32: fifteen complex columns + the packed real one);
// rows: row count the kernel instantiation is built for (>= ceil(L / 16)); L: frame length;
// window: float64[L].  Returns false when the geometry is not served.
inline bool build_mfma_front(int n1, int rows, int L, const double *window, MfmaFrontTables &t) {
  if (n1 != 32 || rows > n1 || rows < 3 || L > rows * 16 || L <= 16 * (rows / 2) + 15) return false;
  const int N = n1 * 16, c = rows / 2, tp = mfma_front_steps(rows);
  t.tp = tp;
  t.center = c;
  const int slots = 2 * tp + 1;
  t.win.assign((size_t)slots * 64, 0.0f);
  t.off.assign((size_t)slots * 64, 0);
  t.emask.assign((size_t)slots * 64, 0.0f);
  t.a_re.assign((size_t)(tp + 1) * 64, 0.0f);
  t.a_im.assign((size_t)(tp + 1) * 64, 0.0f);
  t.tw.assign((size_t)4 * 64 * 2, 0.0f);
  auto even_row = [&](int q) { return ((c + 1 + q) & 1) == 0; };  // parity of the rows lane q holds
  for (int l = 0; l < 64; ++l) {
    const int q = l >> 4, r = l & 15;
    const int safe = 16 * c + r;  // a sample of the frame for slots without one of their own
    for (int s = 0; s < slots; ++s) {
      int row = -1;
      if (s == 2 * tp) {
        if (q == 1) row = c;
      } else {
        const int j = 1 + 4 * (s >> 1) + q;
        if (j <= c) row = (s & 1) ? c - j : c + j;
      }
      const int idx = 16 * row + r;
      const bool have = row >= 0 && row < rows && idx < L;
      t.off[(size_t)s * 64 + l] = have ? idx : safe;
      t.win[(size_t)s * 64 + l] = have ? (float)window[idx] : 0.0f;
      t.emask[(size_t)s * 64 + l] = have ? 1.0f : 0.0f;
    }
    // A operands: lane l holds A[i = l & 15][k = l >> 4]
    const int i = l & 15, k = l >> 4;
    for (int st = 0; st < tp; ++st) {
      const int j = 1 + 4 * st + k;
      if (j > c) continue;
      if (i < 15) {
        const double ang = 2.0 * M_PI * (double)((j * (i + 1)) % n1) / (double)n1;
        t.a_re[(size_t)st * 64 + l] = (float)std::cos(ang);
        t.a_im[(size_t)st * 64 + l] = (float)(-std::sin(ang));
      } else {
        t.a_re[(size_t)st * 64 + l] = even_row(k) ? 1.0f : 0.0f;
      }
    }
    if (k == 1) t.a_re[(size_t)tp * 64 + l] = i < 15 ? 1.0f : (even_row(1) ? 1.0f : 0.0f);
    if (i == 15) t.a_im[(size_t)tp * 64 + l] = even_row(k) ? 0.0f : 1.0f;
    for (int v = 0; v < 4; ++v) {
      const int k1 = 4 * q + v + 1;
      if (k1 > 15) continue;
      // W_N^(r k1) * W_n1^(c k1), reduced exactly before the conversion to an angle
      const long num = ((long)r * k1 + (long)c * k1 * 16) % N;
      const double ang = -2.0 * M_PI * (double)num / (double)N;
      t.tw[((size_t)v * 64 + l) * 2 + 0] = (float)std::cos(ang);
      t.tw[((size_t)v * 64 + l) * 2 + 1] = (float)std::sin(ang);
    }
  }
  return true;
}

}  // namespace pds
