// Diagnostic: phase timing of conv_igemm workgroups via s_memtime stamps (never part of the product).
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -DSTV_STAMPS -I include -I style_transfer_visualizer_amd/csrc
//        tools/conv_stamps.cpp -o tools/conv_stamps
// usage: conv_stamps H W cin cout cfg
#include "../style_transfer_visualizer_amd/csrc/conv_igemm.hip"

#include <algorithm>
#include <cstdio>
#include <vector>

int main(int argc, char** argv) {
  if (argc < 6) return 1;
  const int H = atoi(argv[1]), W = atoi(argv[2]), cin = atoi(argv[3]), cout = atoi(argv[4]);
  setenv("STV_CONV_CFG", argv[5], 1);
  const int wflag = (argc > 6 && atoi(argv[6])) ? STV_W_BLOCKED : 0;   // timing only: data is random either way
  const size_t nx = (size_t)H * W * cin, nw = (size_t)9 * cout * cin, ny = (size_t)H * W * cout;
  std::vector<unsigned short> hx(nx), hw(nw);
  unsigned s = 12345u;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return s; };
  for (auto& v : hx) v = (unsigned short)(0x3c00 + (rnd() >> 22)) ^ (unsigned short)((rnd() >> 31) << 15);
  for (auto& v : hw) v = (unsigned short)(0x3a00 + (rnd() >> 22)) ^ (unsigned short)((rnd() >> 31) << 15);
  void *x, *w, *y;
  float* b;
  hipMalloc(&x, nx * 2); hipMalloc(&w, nw * 2); hipMalloc(&y, ny * 2); hipMalloc(&b, cout * 4);
  hipMemcpy(x, hx.data(), nx * 2, hipMemcpyHostToDevice);
  hipMemcpy(w, hw.data(), nw * 2, hipMemcpyHostToDevice);
  hipMemset(b, 0, cout * 4);
  hipStream_t st;
  hipStreamCreate(&st);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 200; ++i) stv_conv_igemm(x, w, b, nullptr, y, H, W, cin, cout, 9, STV_RELU_OUT | wflag, STV_BF16, st);
  hipStreamSynchronize(st);
  const int reps = 50;
  hipEventRecord(e0, st);
  for (int i = 0; i < reps; ++i) stv_conv_igemm(x, w, b, nullptr, y, H, W, cin, cout, 9, STV_RELU_OUT | wflag, STV_BF16, st);
  hipEventRecord(e1, st);
  hipStreamSynchronize(st);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  const int cfg = stv_conv_config(H, W, cin, cout, 9, STV_BF16);
  static const int th[8] = {8, 8, 4, 4, 4, 8, 4, 2}, bn[8] = {128, 64, 128, 64, 64, 64, 64, 64};
  const int nwg = ((W + 31) / 32) * ((H + th[cfg] - 1) / th[cfg]) * ((cout + bn[cfg] - 1) / bn[cfg]);
  std::vector<unsigned long long> st_h((size_t)nwg * 8);
  hipMemcpyFromSymbol(st_h.data(), HIP_SYMBOL(g_stv_stamps), st_h.size() * 8);
  std::vector<double> ph[5], clk;
  unsigned long long tmin = ~0ull, tmax = 0;
  for (int g = 0; g < nwg; ++g) {
    const unsigned long long* q = &st_h[(size_t)g * 8];
    for (int k = 0; k < 4; ++k) ph[k].push_back((double)(q[k + 1] - q[k]));
    ph[4].push_back((double)(q[4] - q[0]));
    if (q[7] > q[6]) clk.push_back((double)(q[4] - q[0]) / (double)(q[7] - q[6]) * 100.0);
    tmin = std::min(tmin, q[6]); tmax = std::max(tmax, q[7]);
  }
  auto med = [](std::vector<double>& v) { if (v.empty()) return 0.0; std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
  const double us = ms * 1000.0 / reps;
  const double flops = 2.0 * H * W * (double)cout * cin * 9;
  printf("H=%d W=%d cin=%d cout=%d cfg=%d blocked=%d wgs=%d  %.2f us/launch  %.0f TFLOP/s\n", H, W, cin, cout, cfg, wflag ? 1 : 0, nwg, us, flops / us * 1e-6);
  printf("  median cycles: prologue %.0f  mainloop %.0f (%.0f/stage)  ctile %.0f  store %.0f  total %.0f\n", med(ph[0]), med(ph[1]),
         med(ph[1]) / (cin / 16), med(ph[2]), med(ph[3]), med(ph[4]));
  printf("  in-kernel clock (MHz, median) %.0f   first-start..last-end %.2f us\n", med(clk), (double)(tmax - tmin) / 100.0);
  return 0;
}
