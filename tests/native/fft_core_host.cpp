// Host run of the one-wave FFT core of csrc/spectral.hip (SMT_HD functions), lane by lane, against a double-precision
// DFT.  Built and run by tests/test_fft_core_cpu.py with hipcc (no GPU needed: only host code executes).
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../speech-masters-thesis_amd/csrc/spectral.hip"

namespace smt { void set_error(const char*, ...) {} }   // the library's error slot (common.hip) is not linked here

template <int N, bool INV>
static double run_case(unsigned seed) {
  using W = smt::WaveFft<N>;
  std::vector<smt::cplx> tw(N / 2), buf(smt::frame_padded<N>()), snap;
  for (int k = 0; k < N / 2; ++k) tw[k] = {(float)std::cos(-2.0 * M_PI * k / N), (float)std::sin(-2.0 * M_PI * k / N)};
  std::vector<double> xr(N), xi(N);
  srand(seed);
  for (int n = 0; n < N; ++n) {
    xr[n] = rand() / (double)RAND_MAX - 0.5;
    xi[n] = rand() / (double)RAND_MAX - 0.5;
    buf[smt::fidx<N, 64>(n)] = {(float)xr[n], (float)xi[n]};
  }
  W wc[64];
  for (int lane = 0; lane < 64; ++lane) wc[lane].load(tw.data(), lane);
  auto pass = [&](auto fn) { snap = buf; for (int lane = 0; lane < 64; ++lane) fn(lane); };
  pass([&](int lane) { smt::cplx w1a[W::PER1]; for (auto& w : w1a) w = wc[lane].w2;
                       smt::wave_pass<N, W::R1, W::PER1, INV>(snap.data(), buf.data(), 1, lane, w1a); });
  pass([&](int lane) { smt::cplx w1a[W::PER1]; for (auto& w : w1a) w = wc[lane].w2;
                       smt::wave_pass<N, W::R1, W::PER1, INV>(snap.data(), buf.data(), W::R1, lane, w1a); });
  if constexpr (W::R3 > 1)
    pass([&](int lane) { smt::wave_pass<N, W::R3, W::PER3, INV>(snap.data(), buf.data(), W::R1 * W::R1, lane, wc[lane].w3); });
  double worst = 0.0, scale = 0.0;
  const double sgn = INV ? 1.0 : -1.0;
  for (int k = 0; k < N; ++k) {
    double sr = 0.0, si = 0.0;
    for (int n = 0; n < N; ++n) {
      const double a = sgn * 2.0 * M_PI * (double)((long long)k * n % N) / N, c = std::cos(a), s = std::sin(a);
      sr += xr[n] * c - xi[n] * s;
      si += xr[n] * s + xi[n] * c;
    }
    const smt::cplx z = buf[smt::fidx<N, 64>(k)];
    worst = std::fmax(worst, std::hypot(z.x - sr, z.y - si));
    scale = std::fmax(scale, std::hypot(sr, si));
  }
  return worst / scale;
}

int main() {
  double e[8] = {run_case<256, false>(1), run_case<512, false>(2), run_case<1024, false>(3), run_case<2048, false>(4),
                 run_case<256, true>(5),  run_case<512, true>(6),  run_case<1024, true>(7),  run_case<2048, true>(8)};
  int bad = 0;
  const int ns[4] = {256, 512, 1024, 2048};
  for (int i = 0; i < 8; ++i) {
    printf("N=%d %s rel_err=%.3e\n", ns[i % 4], i < 4 ? "forward" : "inverse", e[i]);
    if (!(e[i] < 2e-6)) bad = 1;
  }
  return bad;
}
