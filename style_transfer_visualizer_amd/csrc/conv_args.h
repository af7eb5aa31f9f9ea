// Launch arguments shared by the implicit-GEMM convolution kernels (conv_igemm.hip, conv_ws.hip).
#pragma once
#include <hip/hip_runtime.h>

struct ConvArgs {
  const void* x;
  const void* w;
  const float* bias;
  const void* ref;
  void* y;
  int H, W, cin, cout, flags;
  void* pool;   // optional second output: MaxPool2d(2,2) of y, [H/2][W/2][cout]
  void* pool_idx;  // optional third output with it: one byte per pooled element (stv.h: stv_conv_igemm_pool)
  // optional fused 1x1 term: y = epilogue(mask(ref) * conv3x3(x, w) + x2 . w2^T), x2 [H][W][cin2],
  // w2 [cout][cin2] plain rows (the Gram backward product riding in the dgrad that shares its output)
  const void* x2;
  const void* w2;
  int cin2;
  // optional (bf16 dgrad in front of a max-pool, stv_conv_igemm_route): instead of storing y, route every
  // element to its window's arg-max position in route_out [2H][2W][cout] (zeros elsewhere), reading the
  // forward pass's arg-max byte map route_idx [H][W][cout] - MaxPool2d's backward without a pass of its own
  const void* route_idx = nullptr;
  void* route_out = nullptr;
  // optional: weights of the conv that runs NEXT (stv_conv_next_weights).  Each lane touches one 128-byte line of
  // them between its main loop and its epilogue, so that launch finds them on chip instead of in HBM.
  const void* pf = nullptr;
  uint32_t pf_bytes = 0;
  // how many XCDs share one spatial tile (1, 2 or 4; conv_igemm_kernel.h, block decode): 1 = every channel block of a
  // tile on the tile's XCD (each L2 fetches ALL weights, the input once), g = the channel blocks dealt to g XCDs
  // (each L2 fetches 1/g of the weights, the input g times)
  int xshare = 1;
  // K split ACROSS workgroups (conv_igemm_kernel.h, "XK"): 2 = two workgroups share an output tile, each walks half of
  // K; the one that finishes second adds the other's fp32 partial sums (exchanged through xk_ws) and runs the epilogue.
  // For layers whose grid cannot give every CU a workgroup on a large tile (the 64 x 64-pixel 512-channel layers of a
  // 512 x 512 image).  xk_ws: caller-owned scratch (stv_conv_workspace), zeroed once; the kernel leaves it zeroed.
  int xk = 1;
  void* xk_ws = nullptr;
};
constexpr int kXkSlabOffset = 65536;       // bytes: [ticket, flag] pairs of up to 8,192 output tiles, then the fp32 slabs

// conv_ws.hip: weight-stationary persistent kernel for 3x3, Cin = 64, bf16 (the short-K layers).
// stv_conv_ws_supported() says whether a launch with these arguments can take it.
bool stv_conv_ws_supported(const ConvArgs& a, int dtype, int taps);
int stv_conv_ws_launch(const ConvArgs& a, hipStream_t st);
// conv_ws2.hip: the forward forms of the Cin = 64 layers with two waves per SIMD (K split between them); STV_CONV_WS2
bool stv_conv_ws2_supported(const ConvArgs& a, int dtype, int taps);
int stv_conv_ws2_launch(const ConvArgs& a, hipStream_t st);

// conv_igemm16.hip: the general kernel's tiles on v_mfma_f32_16x16x32_bf16 (configurations 13 and up; bf16, cin % 32 == 0).
int stv_conv_launch_m16(const ConvArgs& a, int cfg, int taps, hipStream_t st);
