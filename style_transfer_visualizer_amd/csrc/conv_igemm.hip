// Implicit-GEMM 3x3 (and 1x1) convolution on the CDNA4 matrix cores, NHWC.
//
//   y[p][n] = sum_{tap} sum_{c} x[p + off(tap)][c] * w[tap][n][c]   (+bias, epilogue)
//
// M = pixels, N = output channels, K = taps * Cin.  A workgroup owns a TH x 32 pixel tile and BN
// output channels.  Per K-stage (32 bytes of K per pixel: 16 bf16 / 8 fp32 channels) the
// (TH+2) x 34 halo tile of the input and the [taps][BN] weight rows of that channel slice sit in
// LDS, and every wave walks the 9 taps as shifted windows of the same LDS tile, so the input is
// fetched ~1.3x (halo) instead of 9x.
//
// Staging is LDS-DMA (buffer_load ... lds): no VGPR round trip and no ds_write - the LDS write
// port was the busiest unit of the register-staged version of this kernel.  One wave-instruction
// moves 64 x 16 B to 1 KiB of consecutive LDS, so the LDS image is lane-linear with a 32-byte row
// pitch; bank conflicts are avoided by an XOR swizzle of the two 16-byte halves of a row
// (half ^= bit 3 of the row index), applied on the source address when the row is fetched and
// again when a fragment is read.  Out-of-range sources (halo pixels outside the image, channel
// rows past Cout, stages past the end of K) are zero-filled by the buffer range check.  A ring of
// three LDS buffers keeps the fetch two K-stages ahead of the MFMAs; every wave issues the same
// number of DMAs per stage, so a counted s_waitcnt vmcnt(N) retires exactly one stage at each
// barrier (one barrier per stage).
//
// MFMA shapes: bf16 -> v_mfma_f32_32x32x16_bf16 (lane (r,h) holds k = 8h..8h+7),
//              fp32 -> v_mfma_f32_32x32x2_f32 x4 with lane (r,h) holding
//              k = 4h..4h+3 (any k permutation is fine as long as A and B agree).
// ReLU-on-load (STV_RELU_IN) is one packed integer max per fragment dword (a negative bf16 / fp32
// is a negative integer), against a scalar that is 0 or INT_MIN - branch-free.
// The MFMAs take the weights as the row operand, so an accumulator lane holds 16 channels of one
// pixel; the epilogue (bias, ReLU, ReLU mask, accumulate, the optional fused 2x2 max-pool) runs in
// registers and v_permlane32_swap pairs the half-waves' channel groups into 16-byte stores.
//
// The same kernel computes the input gradient (dgrad) when handed the flipped, transposed
// weights, and the Gram backward product dF = F * S as a 1x1 conv.
#include <stdlib.h>

#include <mutex>
#include <vector>

#include <type_traits>

#include "stv_common.h"
#include "conv_args.h"

#ifdef STV_STAMPS   // diagnostic build only (tools/conv_stamps.cpp): per-workgroup phase time stamps
__device__ unsigned long long g_stv_stamps[8 * 16384];
#define STV_STAMP(k)                                                                              \
  do {                                                                                            \
    if (threadIdx.x == 0) {                                                                       \
      unsigned long long* s_ = g_stv_stamps + (size_t)blockIdx.x * 8;  \
      s_[(k)] = __builtin_amdgcn_s_memtime();                                                     \
      if ((k) == 0) s_[6] = __builtin_amdgcn_s_memrealtime();                                     \
      if ((k) == 4) s_[7] = __builtin_amdgcn_s_memrealtime();                                     \
    }                                                                                             \
  } while (0)
#endif
#include "conv_igemm_kernel.h"

thread_local const void* g_stv_next_w = nullptr;
thread_local uint32_t g_stv_next_w_bytes = 0;
// Scratch for the K split across workgroups (stv_conv_workspace): the caller's buffer, for the launches of this thread
thread_local void* g_stv_conv_ws = nullptr;
thread_local size_t g_stv_conv_ws_bytes = 0;

namespace {

// ---- generic direct fallback (any Cin/Cout; used for odd shapes in tests) ----
template <typename T, int TAPS>
__global__ __launch_bounds__(256) void conv_direct_kernel(ConvArgs a) {
  const size_t total = (size_t)a.H * a.W * a.cout;
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int n = (int)(idx % a.cout);
  const size_t p = idx / a.cout;
  const int gx = (int)(p % a.W), gy = (int)(p / a.W);
  const T* __restrict__ xin = static_cast<const T*>(a.x);
  const T* __restrict__ wgt = static_cast<const T*>(a.w);
  const bool relu_in = (a.flags & STV_RELU_IN) != 0;
  float s = 0.0f;
  for (int tap = 0; tap < TAPS; ++tap) {
    const int yy = gy + ((TAPS == 9) ? tap / 3 - 1 : 0);
    const int xx = gx + ((TAPS == 9) ? tap % 3 - 1 : 0);
    if (yy < 0 || yy >= a.H || xx < 0 || xx >= a.W) continue;
    const T* xp = xin + ((size_t)yy * a.W + xx) * a.cin;
    const T* wp = wgt + ((size_t)tap * a.cout + n) * a.cin;
    for (int c = 0; c < a.cin; ++c) {
      float xv = elem_traits<T>::load(xp + c);
      if (relu_in) xv = fmaxf(xv, 0.0f);
      s = fmaf(xv, elem_traits<T>::load(wp + c), s);
    }
  }
  if (a.bias) s += a.bias[n];
  if (a.flags & STV_RELU_OUT) s = fmaxf(s, 0.0f);
  T* yout = static_cast<T*>(a.y);
  if (a.flags & STV_MASK) {
    const float m = elem_traits<T>::load(static_cast<const T*>(a.ref) + idx);
    s = (m > 0.0f) ? s : 0.0f;
  }
  if (a.flags & STV_ACCUM) s += elem_traits<T>::load(yout + idx);
  elem_traits<T>::store(yout + idx, s);
}

// ---- tile choice --------------------------------------------------------------------------------
// Configurations: 0 = 8x128, 1 = 8x64, 2 = 4x128, 3 = 4x64 (TH x BN), 4 = 4x64 with K split over two
// wave groups, 5 = 8x64 and 6 = 4x64 on a two-deep LDS ring (two / three workgroups per CU),
// 7 = 2x64 with the K split (the 32x32-pixel layers: four times the workgroups of 4x64 x 2),
// 8 = 1x64 with the K split in four-wave workgroups (a 32x32-pixel layer then covers all 256 CUs).  -1 = the shape is outside the matrix-core tiling (direct fallback).
constexpr int kNumCfg = 19;     // 9 / 10 = 16x64 (four row blocks per wave: half the weight traffic per output) on the three- / two-deep ring
// 11 = 2x32 with the K split (four waves: 2 rows x 2 K groups): on a 32x32-pixel layer still one workgroup per CU, which
// stages 434 KB instead of 1x64's 694 KB;  12 = the same on 4 rows (eight waves)
// 13-17 = 8x64, 16x64, 4x64, 2x32 / K split, 4x32 / K split with the main loop on v_mfma_f32_16x16x32_bf16 (conv_igemm16.hip): two K-stages per
// MFMA out of a ring of four stage slots; bf16 with cin % 32 == 0 only (fp32_cfg names the tile that serves the rest)
// 18 = 16x128 on the two-deep ring (also built in conv_igemm16.hip; bf16 only like its neighbours there)
const int kCfgTH[kNumCfg] = {8, 8, 4, 4, 4, 8, 4, 2, 1, 16, 16, 2, 4, 8, 16, 4, 2, 4, 16},
          kCfgBN[kNumCfg] = {128, 64, 128, 64, 64, 64, 64, 64, 64, 64, 64, 32, 32, 64, 64, 64, 32, 32, 128};
constexpr int kFirstM16 = 13;
bool is_m16(int cfg) { return cfg >= kFirstM16; }       // (= "launched from conv_igemm16.hip": cin % 32 == 0, bf16)

bool cfg_valid(int cfg, int cout) { return cfg >= 0 && cfg < kNumCfg && !(cout <= 64 && kCfgBN[cfg] == 128); }
// fp32 (parity mode) keeps a second accumulator set per K-stage (blocked summation): the eight-wave tiles with 64+
// accumulator registers per lane (8x128, 16x64) would spill at their 256-register budget, so they are served by
// the four-wave 4x128 tile (512 registers per wave) and the 8x64 tiles instead.
int fp32_cfg(int cfg) {
  switch (cfg) {
    case 0: return 2;
    case 9: case 13: case 14: return 1;
    case 10: return 5;
    case 11: case 16: return 7;
    case 12: case 17: return 4;
    case 15: return 3;
    case 18: return 2;
    default: return cfg;
  }
}

// Measured choices (stv_conv_tune), keyed by shape.
struct TuneEntry { int H, W, cin, cout, taps, esize, cfg; };
std::mutex g_tune_mu;
std::vector<TuneEntry> g_tune;

int tuned_cfg(int H, int W, int cin, int cout, int taps, int esize) {
  std::lock_guard<std::mutex> lk(g_tune_mu);
  for (const TuneEntry& e : g_tune)
    if (e.H == H && e.W == W && e.cin == cin && e.cout == cout && e.taps == taps && e.esize == esize) return e.cfg;
  return -1;
}

// Cost model for untuned shapes: waves of workgroups over 256 CUs x work per workgroup / relative
// efficiency of the tile (measured with tools/conv_sweep.py on full-chip layers).
int model_cfg(int H, int W, int cin, int cout) {
  static const float eff[4] = {1.0f, 0.84f, 0.72f, 0.82f};
  int best = 0;
  float best_cost = 3.4e38f;
  for (int i = 0; i < 4; ++i) {
    if (!cfg_valid(i, cout)) continue;
    const long blocks = (long)ceil_div(W, 32) * ceil_div(H, kCfgTH[i]) * ceil_div(cout, kCfgBN[i]);
    const float waves = (float)((blocks + 255) / 256);   // eff is per CU, whatever the residency
    const float cost = waves * (float)(kCfgTH[i] * kCfgBN[i]) / eff[i];
    if (cost < best_cost) { best_cost = cost; best = i; }
  }
  // a grid that cannot give every CU a workgroup, on a deep K: split K inside the workgroup
  const long blocks3 = (long)ceil_div(W, 32) * ceil_div(H, 4) * ceil_div(cout, 64);
  if (best == 3 && blocks3 <= 256 && cin >= 256) best = 4;
  return best;
}

// Tune-table key of the dgrad with the pooling backward in its epilogue (stv_conv_igemm_route): the routed
// epilogue writes four output pixels per accumulator pixel and wants smaller tiles than the plain conv of
// the same shape (round-2 sweep, tools/route_sweep.py: 512^2 128->64 64 us on 4x64 against 87 on 16x64,
// the plain conv's pick; 256^2 256->128 48 against 59 on 8x128), so it is measured as its own "shape".
constexpr int kRouteTaps = STV_TUNE_ROUTE;

int choose_cfg(int H, int W, int cin, int cout, int elem_bytes, int taps = 9) {
  const int kVec = 16 / elem_bytes, CK = 32 / elem_bytes;
  if ((cin % CK) || (cout % kVec)) return -1;
  auto served = [&](int cfg) { return (elem_bytes == 4 || (is_m16(cfg) && cin % 32)) ? fp32_cfg(cfg) : cfg; };
  if (const char* force = getenv("STV_CONV_CFG")) {   // tuning aid (tools/conv_sweep.py)
    const int f = atoi(force);
    if (cfg_valid(f, cout)) return served(f);
  }
  // STV_CONV_TUNE=0 pins the analytic choice even when another caller in this process has
  // measured the shape already (tests that assert near fp32 rounding want one summation order)
  const char* tune = getenv("STV_CONV_TUNE");
  const int t = (tune && atoi(tune) == 0) ? -1 : tuned_cfg(H, W, cin, cout, taps, elem_bytes);
  if (t >= 0) return served(t);
  const int m = model_cfg(H, W, cin, cout);
  return served((taps == kRouteTaps && (m == 0 || m == 2)) ? 3 : m);       // untuned routed dgrad: 4x64 where the model says 128-wide
}

template <typename T, int TAPS>
int launch_mfma(const ConvArgs& a, int cfg, hipStream_t st) {
  if (is_m16(cfg)) {      // two K-stages per MFMA: bf16, whole pairs of 16-channel stages in both K extents
    if (sizeof(T) == 2 && a.cin % 32 == 0 && (a.x2 == nullptr || a.cin2 % 32 == 0)) return stv_conv_launch_m16(a, cfg, TAPS, st);
    cfg = fp32_cfg(cfg);
  }
  switch (cfg) {
    // the two 8-row tiles run 8 waves (two per SIMD: one wave's waits hide under the other's MFMAs)
    case 0: return launch_cfg<Cfg<T, 8, 128, 4, 2, TAPS>>(a, st);   // 64 px x 64 couts per wave
    case 1: return launch_cfg<Cfg<T, 8, 64, 4, 2, TAPS>>(a, st);    // 64 px x 32 couts per wave
    case 2: return launch_cfg<Cfg<T, 4, 128, 1, 4, TAPS>>(a, st);
    case 4: return launch_cfg<Cfg<T, 4, 64, 2, 2, TAPS, 2>>(a, st);   // small layers: K split over two wave groups
    case 5: return launch_cfg<Cfg<T, 8, 64, 4, 2, TAPS, 1, 2>>(a, st);
    case 6: return launch_cfg<Cfg<T, 4, 64, 2, 2, TAPS, 1, 2>>(a, st);
    case 7: return launch_cfg<Cfg<T, 2, 64, 2, 2, TAPS, 2>>(a, st);
    case 8: return launch_cfg<Cfg<T, 1, 64, 1, 2, TAPS, 2>>(a, st);
    case 9: return launch_cfg<Cfg<T, 16, 64, 4, 2, TAPS>>(a, st);
    case 10: return launch_cfg<Cfg<T, 16, 64, 4, 2, TAPS, 1, 2>>(a, st);
    case 11: return launch_cfg<Cfg<T, 2, 32, 2, 1, TAPS, 2>>(a, st);
    case 12: return launch_cfg<Cfg<T, 4, 32, 4, 1, TAPS, 2>>(a, st);
    default: return launch_cfg<Cfg<T, 4, 64, 2, 2, TAPS>>(a, st);
  }
}

// K split across workgroups (ConvArgs::xk): a 3x3 bf16 layer whose output is so small that the 8 x 32-pixel x
// 64-channel tile leaves at least half of the CUs without a workgroup (the 64 x 64-pixel layers of a 512 x 512 image: 128
// tiles), on a K deep enough to halve (Cin >= 256) - every CU then works on one K half of such a tile instead of two
// CUs' worth of workgroups sharing a CU on tiles half the size (0.20 against 0.35 LDS-DMA pieces per MFMA).
// STV_CONV_XK: 0 never, 1 where the rule above holds and the caller provided scratch (stv_conv_workspace); default from
// the measurement recorded in DESIGN.md.
#ifndef STV_CONV_XK_DEFAULT
#define STV_CONV_XK_DEFAULT 0
#endif
template <typename T, int TAPS>
bool xk_wanted(const ConvArgs& a) {
  if (sizeof(T) != 2 || TAPS != 9) return false;
  const char* knob = getenv("STV_CONV_XK");
  if ((knob ? atoi(knob) : STV_CONV_XK_DEFAULT) == 0 || getenv("STV_CONV_CFG") != nullptr) return false;
  if (a.cin < 256 || a.cout % 64 != 0 || g_stv_conv_ws == nullptr) return false;
  const long blocks = (long)ceil_div(a.W, 32) * ceil_div(a.H, 8) * (a.cout / 64);
  const long padded = (blocks + 7) / 8 * 8;
  if (2 * padded > 256 || blocks < 96) return false;                    // every K half on a CU of its own, most CUs busy
  return (size_t)kXkSlabOffset + (size_t)blocks * (8 * 32 * 64 * 4) <= g_stv_conv_ws_bytes && blocks * 8 <= kXkSlabOffset;
}
template <typename T, int TAPS>
int launch_xk(const ConvArgs& a, hipStream_t st) {
  ConvArgs b = a;
  b.xk = 2;
  b.xk_ws = g_stv_conv_ws;
  return launch_cfg<Cfg<T, 8, 64, 4, 2, TAPS>>(b, st);
}

template <typename T, int TAPS>
int launch_typed(const ConvArgs& a, hipStream_t st) {
  // forward forms of the Cin = 64 layers, two waves per SIMD (conv_ws2.hip; STV_CONV_WS2, off by default)
  if (stv_conv_ws2_supported(a, elem_traits<T>::kDtype, TAPS)) {
    g_stv_next_w = nullptr;
    g_stv_next_w_bytes = 0;
    return stv_conv_ws2_launch(a, st);
  }
  // short-K layers (Cin = 64, bf16): weight-stationary persistent kernel (conv_ws.hip)
  if (stv_conv_ws_supported(a, elem_traits<T>::kDtype, TAPS)) {
    g_stv_next_w = nullptr;        // (the weight-stationary kernel does not touch ahead: the hint is consumed, not left for a later launch)
    g_stv_next_w_bytes = 0;
    return stv_conv_ws_launch(a, st);
  }
  if (xk_wanted<T, TAPS>(a)) return launch_xk<T, TAPS>(a, st);
  int cfg = choose_cfg(a.H, a.W, a.cin, a.cout, (int)sizeof(T), TAPS);
  if ((cfg == 7 || cfg == 8 || cfg == 11 || cfg == 12 || cfg == 16 || cfg == 17) && a.pool != nullptr) cfg = 4;      // one row per wave: no pooling window
  if (cfg < 0) {
    const size_t total = (size_t)a.H * a.W * a.cout;
    hipLaunchKernelGGL((conv_direct_kernel<T, TAPS>), dim3((unsigned)((total + 255) / 256)),
                       dim3(256), 0, st, a);
    STV_CHECK_LAUNCH();
    return STV_OK;
  }
  return launch_mfma<T, TAPS>(a, cfg, st);
}

__global__ void tune_fill_kernel(uint32_t* p, size_t n_words, uint32_t seed) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_words) return;
  uint32_t h = (uint32_t)i * 2654435761u + seed;
  h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
  // two bf16 (or one fp32) of magnitude ~1e-2..1, random sign: realistic switching activity
  p[i] = (h & 0x807F807Fu) | 0x3C003C00u;
}

// Turns the caches over between two timed launches (cold tuning): read-modify-write of a buffer larger than
// L2 + Infinity Cache.
__global__ void tune_flush_kernel(uint4* p, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    uint4 v = p[i];
    v.x += 1u;
    p[i] = v;
  }
}

// A measurement must not depend on a hint left for some other launch: cleared for the scope, restored after it.
struct NextWeightsQuiet {
  const void* w = g_stv_next_w;
  uint32_t bytes = g_stv_next_w_bytes;
  NextWeightsQuiet() { g_stv_next_w = nullptr; g_stv_next_w_bytes = 0; }
  ~NextWeightsQuiet() { g_stv_next_w = w; g_stv_next_w_bytes = bytes; }
};

template <typename T>
int tune_typed(int H, int W, int cin, int cout, int key_taps, hipStream_t st) {
  const NextWeightsQuiet quiet;
  const bool route = key_taps == kRouteTaps;       // (bf16 only: the caller checked)
  const int taps = route ? 9 : key_taps;
  const size_t nx = (size_t)H * W * cin, nw = (size_t)taps * cout * cin, ny = (size_t)H * W * cout * (route ? 4 : 1);
  char* buf = nullptr;
  const size_t bx = (nx * sizeof(T) + 255) / 256 * 256, bw = (nw * sizeof(T) + 255) / 256 * 256;
  const size_t bi = route ? ((size_t)H * W * cout + 255) / 256 * 256 : 0;       // arg-max byte map
  if (hipMalloc(reinterpret_cast<void**>(&buf), bx + bw + bi + ny * sizeof(T)) != hipSuccess) {
    (void)hipGetLastError();                       // no room for scratch copies: keep the analytic choice
    return choose_cfg(H, W, cin, cout, (int)sizeof(T), key_taps);
  }
  const size_t words = (bx + bw + bi) / 4;         // (random map bytes: codes 0..255, the routing compares and masks as usual)
  hipLaunchKernelGGL(tune_fill_kernel, dim3((unsigned)((words + 255) / 256)), dim3(256), 0, st,
                     reinterpret_cast<uint32_t*>(buf), words, 0x9E3779B9u);
  ConvArgs a{buf, buf + bx, nullptr, nullptr, buf + bx + bw + bi, H, W, cin, cout, STV_W_BLOCKED, nullptr, nullptr, nullptr, nullptr, 0};
  if (taps == 1) a.flags = 0;
  if (route) {
    a.flags |= STV_MASK;
    a.route_idx = buf + bx + bw;
    a.route_out = a.y;
  }
  // STV_CONV_TUNE=2: rate every tile behind 384 MB of unrelated traffic - inside the step a layer finds its
  // weights and input in HBM, not in L2, and tiles differ in how they take that: the time that
  // counts is (flush + launch) - (flush).
  const char* tmode = getenv("STV_CONV_TUNE");
  uint4* flushbuf = nullptr;
  const size_t flush_n = ((size_t)384 << 20) / sizeof(uint4);
  if (tmode && atoi(tmode) == 2 && hipMalloc(reinterpret_cast<void**>(&flushbuf), flush_n * sizeof(uint4)) != hipSuccess) {
    (void)hipGetLastError();
    flushbuf = nullptr;
  }
  auto flush = [&]() {
    if (flushbuf) hipLaunchKernelGGL(tune_flush_kernel, dim3(2048), dim3(256), 0, st, flushbuf, flush_n);
  };
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  float t_flush = 0.0f;
  if (flushbuf) {
    for (int i = 0; i < 2; ++i) flush();
    (void)hipEventRecord(e0, st);
    for (int i = 0; i < 6; ++i) flush();
    (void)hipEventRecord(e1, st);
    (void)hipEventSynchronize(e1);
    (void)hipEventElapsedTime(&t_flush, e0, e1);
  }
  int base = route ? ((model_cfg(H, W, cin, cout) == 0 || model_cfg(H, W, cin, cout) == 2) ? 3 : model_cfg(H, W, cin, cout))
                   : model_cfg(H, W, cin, cout);
  if (sizeof(T) == 4) base = fp32_cfg(base);
  int best = base;
  float t_best = 3.4e38f, t_base = 3.4e38f;
  int rc = STV_OK;
  // The 16x128 tile (18) is not offered by default: it wins the hot loop by 7-9 % on every shape with >= 512 tiles and
  // LOSES in the step (round 4: closure +1.9 % at 1024^2 with it on the 256^2 layers, nothing at 3840x2160) - see DESIGN 3.8
  int ncfg = kNumCfg - 1;
  if (const char* lim = getenv("STV_CONV_TUNE_CFGS")) ncfg = atoi(lim) < kNumCfg ? atoi(lim) : kNumCfg;   // A/B aid (19: every tile)
  // two interleaved rounds, the faster time of each configuration counts: one round's order effects
  // (clock ramp after the fill, a neighbour's tail) otherwise decide between near-equal tiles
  float t_cfg[kNumCfg];
  for (int cfg = 0; cfg < kNumCfg; ++cfg) t_cfg[cfg] = 3.4e38f;
  for (int round = 0; round < 2 && rc == STV_OK; ++round)
    for (int cfg = 0; cfg < ncfg && rc == STV_OK; ++cfg) {
      if (!cfg_valid(cfg, cout) || (sizeof(T) == 4 && fp32_cfg(cfg) != cfg)) continue;
      const int kWarm = 2, kReps = flushbuf ? 6 : 10;
      for (int i = 0; i < kWarm && rc == STV_OK; ++i)
        rc = taps == 9 ? launch_mfma<T, 9>(a, cfg, st) : launch_mfma<T, 1>(a, cfg, st);
      (void)hipEventRecord(e0, st);
      for (int i = 0; i < kReps && rc == STV_OK; ++i) {
        flush();
        rc = taps == 9 ? launch_mfma<T, 9>(a, cfg, st) : launch_mfma<T, 1>(a, cfg, st);
      }
      (void)hipEventRecord(e1, st);
      if (hipEventSynchronize(e1) != hipSuccess) rc = STV_ERR_LAUNCH;
      float ms = 0.0f;
      (void)hipEventElapsedTime(&ms, e0, e1);
      if (flushbuf) ms -= t_flush;
      if (ms < t_cfg[cfg]) t_cfg[cfg] = ms;
    }
  for (int cfg = 0; cfg < ncfg; ++cfg) {
    if (cfg == base) t_base = t_cfg[cfg];
    if (t_cfg[cfg] < t_best) { t_best = t_cfg[cfg]; best = cfg; }
  }
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  (void)hipFree(buf);
  if (flushbuf) (void)hipFree(flushbuf);
  if (rc != STV_OK) return -(100 + rc);
  // keep the model's choice unless something else is clearly (3 %) faster: fewer flips run to run
  if (best != base && t_best > 0.97f * t_base) best = base;
  std::lock_guard<std::mutex> lk(g_tune_mu);
  for (TuneEntry& e : g_tune)
    if (e.H == H && e.W == W && e.cin == cin && e.cout == cout && e.taps == key_taps && e.esize == (int)sizeof(T)) {
      e.cfg = best;
      return best;
    }
  g_tune.push_back(TuneEntry{H, W, cin, cout, key_taps, (int)sizeof(T), best});
  return best;
}

}  // namespace

extern "C" int stv_conv_config(int H, int W, int cin, int cout, int taps, int dtype) {
  return choose_cfg(H, W, cin, cout, dtype == STV_BF16 ? 2 : 4, taps);
}

extern "C" int stv_conv_uses_ws(int H, int W, int cin, int cout, int taps, int dtype, int flags, int has_ref, int has_pool) {
  static const char dummy = 0;
  ConvArgs a{&dummy, &dummy, nullptr, has_ref ? &dummy : nullptr, const_cast<char*>(&dummy), H, W, cin, cout, flags,
             has_pool ? const_cast<char*>(&dummy) : nullptr, nullptr, has_ref ? &dummy : nullptr, has_ref ? &dummy : nullptr,
             has_ref ? cout : 0};      // (cin2 = the layer's own channel count: the Gram term of its tap)
  return stv_conv_ws_supported(a, dtype, taps) ? 1 : 0;
}

// The tile table as data: 7 ints per entry {H, W, cin, cout, taps (9, 1 or STV_TUNE_ROUTE), element bytes, cfg}.
// Measured choices are PERSISTED by the host (style_transfer_visualizer_amd/conv_tiles_gfx950.json, produced by
// tools/tune_tiles.py on an MI355X) and imported when the library is loaded: which tile a shape runs on - and with
// it the summation order of its results and the kernel name a profile shows - is then the same in every run.
extern "C" int stv_conv_tune_export(int* out7, int max_entries) {
  std::lock_guard<std::mutex> lk(g_tune_mu);
  int n = 0;
  for (const TuneEntry& e : g_tune) {
    if (out7 && n < max_entries) {
      int* o = out7 + 7 * n;
      o[0] = e.H; o[1] = e.W; o[2] = e.cin; o[3] = e.cout; o[4] = e.taps; o[5] = e.esize; o[6] = e.cfg;
    }
    ++n;
  }
  return n;
}

extern "C" int stv_conv_tune_import(const int* in7, int n_entries) {
  if (n_entries < 0 || (n_entries > 0 && !in7)) return STV_ERR_ARG;
  if (n_entries == 0) {                      // empty import: forget everything measured or imported so far
    std::lock_guard<std::mutex> lk(g_tune_mu);
    g_tune.clear();
    return STV_OK;
  }
  for (int i = 0; i < n_entries; ++i) {
    const int* e = in7 + 7 * i;
    if (e[0] <= 0 || e[1] <= 0 || e[2] <= 0 || e[3] <= 0 || (e[5] != 2 && e[5] != 4) || !cfg_valid(e[6], e[3])) return STV_ERR_ARG;
    if (e[4] != 9 && e[4] != 1 && e[4] != kRouteTaps) return STV_ERR_ARG;
  }
  std::lock_guard<std::mutex> lk(g_tune_mu);
  for (int i = 0; i < n_entries; ++i) {
    const int* e = in7 + 7 * i;
    bool found = false;
    for (TuneEntry& t : g_tune)
      if (t.H == e[0] && t.W == e[1] && t.cin == e[2] && t.cout == e[3] && t.taps == e[4] && t.esize == e[5]) { t.cfg = e[6]; found = true; }
    if (!found) g_tune.push_back(TuneEntry{e[0], e[1], e[2], e[3], e[4], e[5], e[6]});
  }
  return STV_OK;
}

extern "C" int stv_conv_tune(int H, int W, int cin, int cout, int taps, int dtype, void* stream) {
  if (H <= 0 || W <= 0 || cin <= 0 || cout <= 0 || (taps != 9 && taps != 1 && taps != kRouteTaps)) return -(100 + STV_ERR_ARG);
  if (dtype != STV_F32 && dtype != STV_BF16) return -(100 + STV_ERR_ARG);
  if (taps == kRouteTaps && dtype != STV_BF16) return -(100 + STV_ERR_ARG);
  if ((size_t)H * W * (size_t)(cin > cout ? cin : cout) >= (size_t)1 << 31) return -(100 + STV_ERR_ARG);
  if (taps == kRouteTaps && (size_t)4 * H * W * (size_t)cout * 2 >= (size_t)1 << 31) return -(100 + STV_ERR_ARG);
  const int esize = dtype == STV_BF16 ? 2 : 4;
  if ((cin % (32 / esize)) || (cout % (16 / esize))) return -1;          // direct-kernel shape: nothing to tune
  // STV_CONV_TUNE: unset = use the table (imported + measured so far), never measure; 0 = analytic choice only;
  // 1 = measure shapes the table does not know; 2 = the same behind a cache turn-over (cold operands)
  const char* mode = getenv("STV_CONV_TUNE");
  if (!mode || atoi(mode) <= 0) return choose_cfg(H, W, cin, cout, esize, taps);
  const int known = tuned_cfg(H, W, cin, cout, taps, esize);
  if (known >= 0) return known;
  hipStream_t st = static_cast<hipStream_t>(stream);
  return dtype == STV_BF16 ? tune_typed<bf16_t>(H, W, cin, cout, taps, st) : tune_typed<float>(H, W, cin, cout, taps, st);
}

extern "C" int stv_conv_igemm(const void* x, const void* w, const float* bias, const void* ref,
                              void* y, int H, int W, int cin, int cout, int taps, int flags,
                              int dtype, void* stream) {
  if (!x || !w || !y || H <= 0 || W <= 0 || cin <= 0 || cout <= 0) return STV_ERR_ARG;
  if ((flags & STV_MASK) && !ref) return STV_ERR_ARG;
  if (flags & STV_POOL_ONLY) return STV_ERR_ARG;          // only stv_conv_igemm_pool has a pooled map to write instead of y
  if (taps != 9 && taps != 1) return STV_ERR_ARG;
  // the direct fallback (shapes the matrix-core tiling does not cover) reads plain weights only
  if ((flags & STV_W_BLOCKED) && choose_cfg(H, W, cin, cout, dtype == STV_F32 ? 4 : 2, taps) < 0) return STV_ERR_ARG;
  if ((size_t)H * W * (size_t)(cin > cout ? cin : cout) >= (size_t)1 << 31) return STV_ERR_ARG;
  ConvArgs a{x, w, bias, ref, y, H, W, cin, cout, flags, nullptr, nullptr, nullptr, nullptr, 0};
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (dtype == STV_F32)
    return taps == 9 ? launch_typed<float, 9>(a, st) : launch_typed<float, 1>(a, st);
  if (dtype == STV_BF16)
    return taps == 9 ? launch_typed<bf16_t, 9>(a, st) : launch_typed<bf16_t, 1>(a, st);
  return STV_ERR_ARG;
}

extern "C" int stv_conv_igemm_pool(const void* x, const void* w, const float* bias, void* y, void* y_pool,
                                   void* pool_idx, int H, int W, int cin, int cout, int flags, int dtype,
                                   void* stream) {
  if (!x || !w || !y_pool || H <= 1 || W <= 1 || cin <= 0 || cout <= 0) return STV_ERR_ARG;
  if (!y && !(flags & STV_POOL_ONLY)) return STV_ERR_ARG;
  if (flags & (STV_MASK | STV_ACCUM)) return STV_ERR_ARG;                     // forward convolutions only
  if ((size_t)H * W * (size_t)(cin > cout ? cin : cout) >= (size_t)1 << 31) return STV_ERR_ARG;
  if (dtype != STV_F32 && dtype != STV_BF16) return STV_ERR_ARG;
  // the fused pool lives in the matrix-core kernel's epilogue: other shapes pool separately
  if (choose_cfg(H, W, cin, cout, dtype == STV_F32 ? 4 : 2, 9) < 0) return STV_ERR_ARG;
  ConvArgs a{x, w, bias, nullptr, y, H, W, cin, cout, flags, y_pool, pool_idx, nullptr, nullptr, 0};
  hipStream_t st = static_cast<hipStream_t>(stream);
  return dtype == STV_F32 ? launch_typed<float, 9>(a, st) : launch_typed<bf16_t, 9>(a, st);
}

extern "C" int stv_conv_num_configs(void) { return kNumCfg; }

extern "C" void stv_conv_workspace(void* ws, size_t bytes) {
  g_stv_conv_ws = (ws != nullptr && bytes > (size_t)kXkSlabOffset) ? ws : nullptr;
  g_stv_conv_ws_bytes = g_stv_conv_ws ? bytes : 0;
}

extern "C" size_t stv_conv_workspace_bytes(void) {
  return (size_t)kXkSlabOffset + (size_t)128 * (8 * 32 * 64 * 4);       // the largest launch that splits K: 128 tiles of 8 x 32 x 64
}

extern "C" void stv_conv_next_weights(const void* w, size_t bytes) {
  g_stv_next_w = (bytes > 0 && bytes < ((size_t)1 << 31)) ? w : nullptr;
  g_stv_next_w_bytes = g_stv_next_w ? (uint32_t)bytes : 0;
}

extern "C" int stv_conv_igemm_route(const void* x, const void* w, const void* pool_idx, void* y_full, int H, int W, int cin,
                                    int cout, int flags, int dtype, void* stream) {
  if (!x || !w || !pool_idx || !y_full || H <= 0 || W <= 0 || cin <= 0 || cout <= 0) return STV_ERR_ARG;
  if (dtype != STV_BF16) return STV_ERR_ARG;                                   // packed-word routing: bf16 storage only
  if (flags & (STV_RELU_IN | STV_RELU_OUT | STV_ACCUM | STV_POOL_ONLY)) return STV_ERR_ARG;
  // 32-bit buffer offsets: the input (H x W x cin) and the routed output (2H x 2W x cout), bf16
  if ((size_t)H * W * (size_t)cin * 2 >= (size_t)1 << 31 || (size_t)4 * H * W * (size_t)cout * 2 >= (size_t)1 << 31) return STV_ERR_ARG;
  if (choose_cfg(H, W, cin, cout, 2, kRouteTaps) < 0) return STV_ERR_ARG;
  ConvArgs a{x, w, nullptr, nullptr, y_full, H, W, cin, cout, flags, nullptr, nullptr, nullptr, nullptr, 0};
  a.route_idx = pool_idx;
  a.route_out = y_full;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (xk_wanted<bf16_t, 9>(a)) return launch_xk<bf16_t, 9>(a, st);
  int cfg = choose_cfg(H, W, cin, cout, 2, kRouteTaps);
  return launch_mfma<bf16_t, 9>(a, cfg, st);                                   // always the general kernel
}

extern "C" int stv_conv_igemm_dual(const void* x, const void* w, const void* x2, const void* w2, const void* ref,
                                   void* y, int H, int W, int cin, int cin2, int cout, int flags, int dtype,
                                   void* stream) {
  if (!x || !w || !x2 || !w2 || !y || H <= 0 || W <= 0 || cin <= 0 || cin2 <= 0 || cout <= 0) return STV_ERR_ARG;
  if ((flags & STV_MASK) && !ref) return STV_ERR_ARG;
  if (flags & (STV_RELU_IN | STV_RELU_OUT | STV_POOL_ONLY)) return STV_ERR_ARG;   // a gradient path: no activations, no pooled output
  if (dtype != STV_F32 && dtype != STV_BF16) return STV_ERR_ARG;
  const size_t cmax = (size_t)(cin > cout ? cin : cout) > (size_t)cin2 ? (size_t)(cin > cout ? cin : cout) : (size_t)cin2;
  if ((size_t)H * W * cmax >= (size_t)1 << 31) return STV_ERR_ARG;
  const int es = dtype == STV_F32 ? 4 : 2;
  // both terms run in the matrix-core kernel: its channel granularity applies to both K extents
  if (choose_cfg(H, W, cin, cout, es, 9) < 0 || cin2 % (32 / es)) return STV_ERR_ARG;
  ConvArgs a{x, w, nullptr, ref, y, H, W, cin, cout, flags, nullptr, nullptr, x2, w2, cin2};
  hipStream_t st = static_cast<hipStream_t>(stream);
  return dtype == STV_F32 ? launch_typed<float, 9>(a, st) : launch_typed<bf16_t, 9>(a, st);
}
