// Weight-stationary 3x3 convolution, Cin = 64, bf16, FORWARD forms - TWO waves per SIMD (round 5).
//
// conv_ws.hip runs one wave per SIMD (the 144 weight registers of a 32-channel output block leave no room for a
// second one) and its tile loop - 33.7 clock ticks per MFMA in isolation (tools/mfma_cadence_probe.cpp) - runs at ~60
// in the kernel: ~2.5 k cycles of serial per-tile work (accumulator rounding, the LDS-DMA burst, index arithmetic,
// barrier) that a wave alone on its SIMD cannot overlap with anything.  Here K is split between the TWO waves of a SIMD:
//
//  * eight waves per workgroup: (row pair wm, channel half wn, K half kh); a wave keeps the weights of two of the four
//    16-channel K-stages (18 fragments = 72 registers, pinned to AGPRs) and works on 2 rows x 32 pixels x 32 channels
//    of a 4 x 32-pixel tile: 36 MFMAs per tile, 32 accumulator registers.  Waves w and w + 4 share a SIMD, a tile and a
//    channel half; they differ in kh;
//  * at the end of a tile the two exchange halves of their partial sums through LDS (double-buffered: 4 KB per wave and
//    parity): wave kh finishes the channel groups 2 kh, 2 kh + 1 (16 of the 32 channels) of BOTH rows - so a pooling
//    window (two rows, two neighbouring lanes) stays inside one wave - adds the partner's sums to its own
//    (own + partner: a + b == b + a, the result does not depend on which wave is which), adds the bias, rounds, applies
//    the ReLU, pools, stores.  All of that, its DMA issue and its index arithmetic run under the PARTNER's MFMAs;
//  * the whole-K halo tile of a later tile (6 x 34 pixels x 128 B = 28 KB, LDS-DMA, zero-filled outside the image)
//    streams into one of three LDS buffers as in conv_ws.hip; every wave issues four of a tile's 28 pieces.
//
// Same arithmetic per output as conv_ws.hip except for the order of the two K halves' sums (stages {0,1} + {2,3}
// instead of 0,1,2,3 in sequence) and the bias entering last instead of first: results agree to rounding of the fp32
// sums (tests/test_gpu_ops.py holds both to the same bound against the CPU oracle).
#include <stdlib.h>
#include <type_traits>

#include "stv_common.h"
#include "conv_args.h"

namespace {

constexpr int TW = 32, IN_W = TW + 2, TH = 4, IN_H = TH + 2, IN_PIX = IN_H * IN_W;   // 204 halo pixels
constexpr int KB = 32, CK = 16, CIN = 64, NSTAGE = CIN / CK, PIX_BYTES = CIN * 2;
constexpr int IN_PIECES = (IN_PIX * 2 + 63) / 64;                 // 7 one-KiB pieces per K-stage
constexpr int IN_STAGE = IN_PIECES * 1024, IN_BYTES = NSTAGE * IN_STAGE;   // 28,672
constexpr int NB = 3;
constexpr int XCH_WAVE = 2 * 8 * 64 * 4;                          // 2 rows x 8 registers x 64 lanes x 4 B = 4 KB
constexpr int XCH_BYTES = 2 * 8 * XCH_WAVE;                       // two parities x eight waves
constexpr int SPARE_OFF = NB * IN_BYTES + XCH_BYTES;
constexpr int LDS_BYTES = SPARE_OFF + 1024;
constexpr int MT = 2, AROWS = MT + 2, NCOL = 6;                   // tap columns per wave and tile: 2 stages x 3 dx
constexpr int PPW = 4;                                            // DMA pieces per wave and tile (28 real ones + 4 dummies)
constexpr int STORES = 5;                                         // per wave and tile: 2 rows + pooled row + 2 map words
constexpr uint32_t kOob = 0x80000000u;
static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

__device__ __forceinline__ bf16x8v relu_frag(bf16x8v v) {
  typedef __attribute__((ext_vector_type(8))) short s16x8;
  return __builtin_bit_cast(bf16x8v, __builtin_elementwise_max(__builtin_bit_cast(s16x8, v), (s16x8)((short)0)));
}

template <bool RELU_IN, bool POOL, bool SKEW>
__global__ __launch_bounds__(512) void conv_ws2_kernel(ConvArgs a, int n_workgroups) {
#if defined(__HIP_DEVICE_COMPILE__)
  using lds_ptr = __attribute__((address_space(3))) void*;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave8 = __builtin_amdgcn_readfirstlane(tid >> 6) & 7;
  const int kh = wave8 >> 2, w4 = wave8 & 3, wm = w4 >> 1, wn = w4 & 1;
  const int r = lane & 31, h = lane >> 5;

  const int tiles_x = (a.W + TW - 1) / TW;
  const int ntiles = tiles_x * ((a.H + TH - 1) / TH);
  const int ncb = a.cout / 64;
  const int cb = (int)blockIdx.x % ncb;
  const int t_first = (int)blockIdx.x / ncb, tstride = n_workgroups / ncb;
  const int nb = cb * 64 + wn * 32;                                  // first output channel of this wave's block

  const bf16_t* __restrict__ xin = static_cast<const bf16_t*>(a.x);
  const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(xin), 0, a.H * a.W * PIX_BYTES, 0x00020000);

  // ---- resident weights: the K-stages 2 kh and 2 kh + 1, nine taps each (row = channel nb + r, k = 8h..8h+7)
  bf16x8v wreg[2][9];
  {
    const bool w_blocked = (a.flags & STV_W_BLOCKED) != 0;
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.w), 0, 9 * a.cout * CIN * 2, 0x00020000);
#pragma unroll
    for (int sl = 0; sl < 2; ++sl)
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        const int s = 2 * kh + sl, n = nb + r;
        const int elem = w_blocked ? (((tap * NSTAGE + s) * a.cout + n) * CK + h * 8) : ((tap * a.cout + n) * CIN + s * CK + h * 8);
        wreg[sl][tap] = __builtin_bit_cast(bf16x8v, __builtin_amdgcn_raw_buffer_load_b128(rs_w, (uint32_t)(elem * 2), 0, 0));
      }
#pragma unroll
    for (int sl = 0; sl < 2; ++sl)
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) asm volatile("" : "+a"(wreg[sl][tap]));        // MFMA operands only: AGPRs
  }
  // the bias rides in as the C operand of the FIRST K half's first MFMAs (wave class kh = 0); the second half starts at zero
  f32x16 acc0;
  {
    const __amdgpu_buffer_rsrc_t rs_b = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(a.bias), 0, (a.bias != nullptr && kh == 0) ? a.cout * 4 : 0, 0x00020000);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const u32x4 t = __builtin_amdgcn_raw_buffer_load_b128(rs_b, (uint32_t)((nb + 8 * j + 4 * h) * 4), 0, 0);
#pragma unroll
      for (int e = 0; e < 4; ++e) acc0[4 * j + e] = __uint_as_float(t[e]);
    }
  }

  // ---- DMA: wave w fetches, of K-stage (w & 3), the pieces kh, kh + 2, ... (4 / 3 of the 7; a fourth dummy for kh = 1)
  const int dstage = w4;
  int in_rel[PPW], in_yx[PPW];
#pragma unroll
  for (int q = 0; q < PPW; ++q) {
    const int p = kh + 2 * q;
    const int pix = p * 32 + (lane >> 1);
    const int half = (lane & 1) ^ ((pix >> 3) & 1);
    const int py = pix / IN_W, px = pix - py * IN_W;
    in_rel[q] = (py * a.W + px) * PIX_BYTES + half * 16;
    in_yx[q] = (p < IN_PIECES && pix < IN_PIX) ? ((py << 8) | px) : (0x3FFF << 8);
  }
  auto issue_tile = [&](int t, char* buf) {          // t >= ntiles: four pieces aimed out of range at the spare KiB
    const bool live = t < ntiles;
    const int ty = t / tiles_x, tx = t - ty * tiles_x;
    const int y0 = ty * TH, x0 = tx * TW;
    const int base = ((y0 - 1) * a.W + (x0 - 1)) * PIX_BYTES;
#pragma unroll
    for (int q = 0; q < PPW; ++q) {
      const int p = kh + 2 * q;
      const int gy = y0 - 1 + (in_yx[q] >> 8), gx = x0 - 1 + (in_yx[q] & 255);
      const bool ok = live && (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W;
      const uint32_t off = ok ? (uint32_t)(base + in_rel[q]) : kOob;
      char* dst = (live && p < IN_PIECES) ? buf + dstage * IN_STAGE + p * 1024 : smem + SPARE_OFF;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, (lds_ptr)dst, 16, off, dstage * KB, 0, 0);
    }
  };

  // lane-constant LDS offsets of this lane's A fragments inside a stage of a tile buffer
  int a_addr[3][AROWS];
#pragma unroll
  for (int dx = 0; dx < 3; ++dx)
#pragma unroll
    for (int j = 0; j < AROWS; ++j) {
      const int pix = (wm * MT + j) * IN_W + dx + r;
      a_addr[dx][j] = pix * KB + ((h ^ ((pix >> 3) & 1)) << 4);
    }

  const bool relu_out = (a.flags & STV_RELU_OUT) != 0;
  const int out_bytes = a.H * a.W * a.cout * 2;
  const __amdgpu_buffer_rsrc_t rs_y = __builtin_amdgcn_make_buffer_rsrc(
      a.y, 0, (a.y != nullptr && !(POOL && (a.flags & STV_POOL_ONLY))) ? out_bytes : 0, 0x00020000);
  const int Hp = a.H >> 1, Wp = a.W >> 1;
  const __amdgpu_buffer_rsrc_t rs_p = __builtin_amdgcn_make_buffer_rsrc(a.pool, 0, POOL ? Hp * Wp * a.cout * 2 : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_i = __builtin_amdgcn_make_buffer_rsrc(
      a.pool_idx, 0, (POOL && a.pool_idx != nullptr) ? Hp * Wp * a.cout : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_none = __builtin_amdgcn_make_buffer_rsrc(a.y, 0, 0, 0x00020000);

  typedef __attribute__((ext_vector_type(2))) short s16x2;
  const s16x2 relu_lo = (s16x2)((short)(relu_out ? 0 : -32768));
  char* const xch = smem + NB * IN_BYTES;

  // ---- a finished tile: own + partner's sums of the channel groups 2 KH, 2 KH + 1 (the bias came in through the first
  // half's accumulators), rounding, ReLU on the packed words, stores (STORES vector-memory operations, always)
  auto finish = [&](auto khc, const float (&own)[MT][8], const char* theirs, int y0, int x0, bool live) {
    constexpr int KH = decltype(khc)::value;
    uint32_t P[MT][4];                               // [row][jj * 2 + pair]: channels nb + 8 (2 KH + jj) + 4 h + 2 pair + {0, 1}
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const f32x4 o = *reinterpret_cast<const f32x4*>(theirs + ((mt * 2 + q) * 64 + lane) * 16);
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = KH == 0 ? own[mt][4 * q + e] + o[e] : o[e] + own[mt][4 * q + e];   // (first half + second half)
        P[mt][q * 2 + 0] = __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(s16x2, pack_bf16x2(v[0], v[1])), relu_lo));
        P[mt][q * 2 + 1] = __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(s16x2, pack_bf16x2(v[2], v[3])), relu_lo));
      }
    // a lane holds, per row, 2 groups x 4 channels; lanes 32-63 of the first group's words <-> lanes 0-31 of the second's:
    // every lane then stores 16 contiguous bytes (channels nb + 16 KH + 8 h .. + 7)
    auto store_row = [&](const uint32_t (&Q)[4], uint32_t poff, const __amdgpu_buffer_rsrc_t& rs_out) {
      const auto sx = __builtin_amdgcn_permlane32_swap(Q[0], Q[2], false, false);     // pairs (0,1) of group jj = 0 / 1
      const auto sy = __builtin_amdgcn_permlane32_swap(Q[1], Q[3], false, false);     // pairs (2,3)
      const u32x4 out = {sx[0], sy[0], sx[1], sy[1]};
      const uint32_t off = poff != kOob ? poff + (uint32_t)((nb + 16 * KH + 8 * h) * 2) : kOob;
      __builtin_amdgcn_raw_buffer_store_b128(out, rs_out, off, 0, 0);
    };
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const int gy = y0 + wm * MT + mt, gx = x0 + r;
      store_row(P[mt], (live && gy < a.H && gx < a.W) ? (uint32_t)(((gy * a.W + gx) * a.cout) * 2) : kOob, rs_y);
    }
    if constexpr (POOL) {
      // MaxPool2d(2,2) + arg-max byte map on the stored (non-negative) words, as conv_ws.hip: window = this wave's two rows
      // x the neighbouring lane; scan order top-left, top-right, bottom-left, bottom-right (first maximum wins)
      auto right = [](uint32_t v) -> uint32_t { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, false); };
      auto gt = [](uint32_t best, uint32_t cand) -> uint32_t {
        return __builtin_bit_cast(uint32_t, __builtin_bit_cast(s16x2, best) - __builtin_bit_cast(s16x2, cand)) & 0x80008000u;
      };
      auto mx = [](uint32_t x, uint32_t y) -> uint32_t {
        return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(s16x2, x), __builtin_bit_cast(s16x2, y)));
      };
      uint32_t Qp[4], code[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const uint32_t tl = P[0][q], bl = P[1][q];
        const uint32_t tr = right(tl), br = right(bl);
        const uint32_t c1 = gt(tl, tr), m1 = mx(tl, tr);
        const uint32_t c2 = gt(m1, bl), m2 = mx(m1, bl);
        const uint32_t c3 = gt(m2, br), m3 = mx(m2, br);
        Qp[q] = m3;
        const uint32_t b1 = c2 | c3;
        const uint32_t b0 = (c2 & c3) | (~c2 & (c1 | c3));
        typedef __attribute__((ext_vector_type(2))) unsigned short u16x2;
        const uint32_t pos = __builtin_bit_cast(uint32_t, __builtin_bit_cast(u16x2, m3) + (u16x2)((unsigned short)0x7FFF)) & 0x80008000u;
        code[q] = (b0 >> 15) | (b1 >> 14) | (pos >> 13);
      }
      const int gyp = (y0 + wm * MT) >> 1, gxp = (x0 + r) >> 1;
      const bool pix_ok = live && (r & 1) == 0 && gyp < Hp && gxp < Wp;
      store_row(Qp, pix_ok ? (uint32_t)(((gyp * Wp + gxp) * a.cout) * 2) : kOob, rs_p);
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) {                // group 2 KH + jj: channels e = 0,1 from code[2 jj], e = 2,3 from code[2 jj + 1]
        const uint32_t word = __builtin_amdgcn_perm(code[2 * jj + 1], code[2 * jj], 0x06040200u);
        const int nn = nb + 8 * (2 * KH + jj) + 4 * h;
        const uint32_t off = pix_ok ? (uint32_t)((gyp * Wp + gxp) * a.cout + nn) : kOob;
        __builtin_amdgcn_raw_buffer_store_b32(word, rs_i, off, 0, 0);
      }
    } else {
      // the vector-memory operation count per tile is the same in both forms (the counted wait below)
#pragma unroll
      for (int k = 0; k < STORES - MT; ++k) __builtin_amdgcn_raw_buffer_store_b32(0u, rs_none, kOob, 0, 0);
    }
  };
  using K0 = std::integral_constant<int, 0>;
  using K1 = std::integral_constant<int, 1>;

  // ---- prologue: three tiles in flight, landed
  int t = t_first;
#pragma unroll
  for (int b = 0; b < NB; ++b) issue_tile(t + b * tstride, smem + b * IN_BYTES);
  wait_vmcnt<0>();
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");

  // wave class B (kh = 1, SKEW): the tile it still has to finish - its kept sums, where it was, the buffer that is free again
  float kept[MT][8];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int i = 0; i < 8; ++i) kept[mt][i] = 0.0f;
  int py0 = 0, px0 = 0, pt = ntiles;
  char* pbuf = smem;
  bool have_prev = false;

  int slot = 0, par = 0;
  for (; t < ntiles; t += tstride, slot = (slot + 1 == NB) ? 0 : slot + 1, par ^= 1) {
    char* const cur = smem + slot * IN_BYTES;
    const int ty = t / tiles_x, tx = t - ty * tiles_x;
    const int y0 = ty * TH, x0 = tx * TW;

    // ---- this wave's two K-stages: 2 x 3 tap columns x 3 rows of taps x 2 row blocks = 36 MFMAs
    bf16x8v af[3][AROWS];
    auto load_col = [&](int col, int set) {            // col = sl * 3 + dx
      const int sl = col / 3, dx = col - sl * 3;
#pragma unroll
      for (int j = 0; j < AROWS; ++j) {
        const bf16x8v v = *reinterpret_cast<const bf16x8v*>(cur + (2 * kh + sl) * IN_STAGE + a_addr[dx][j]);
        af[set][j] = RELU_IN ? relu_frag(v) : v;
      }
    };
    f32x16 acc[MT];
    auto column = [&](int col) {
      const int sl = col / 3, dx = col - sl * 3;
      if (col + 2 < NCOL) load_col(col + 2, (col + 2) % 3);
#pragma unroll
      for (int dy = 0; dy < 3; ++dy)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
          acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wreg[sl][dy * 3 + dx], af[col % 3][mt + dy],
                                                            (col == 0 && dy == 0) ? acc0 : acc[mt], 0, 0, 0);
    };
    load_col(0, 0);
    load_col(1, 1);
#pragma unroll
    for (int col = 0; col < 3; ++col) column(col);
    if (SKEW && kh == 1) {
      // class B finishes the PREVIOUS tile here, half a tile after class A did: each class's DMA issue, rounding and stores
      // fall under the other's MFMAs
      issue_tile(pt, pbuf);
      finish(K1{}, kept, xch + ((par ^ 1) * 8 + (wave8 ^ 4)) * XCH_WAVE, py0, px0, have_prev);
    }
#pragma unroll
    for (int col = 3; col < NCOL; ++col) column(col);

    // ---- exchange: the half this wave does NOT finish goes to the partner; lane-linear, 16 bytes per lane and store
    char* const mine = xch + (par * 8 + wave8) * XCH_WAVE;
    const char* const theirs = xch + (par * 8 + (wave8 ^ 4)) * XCH_WAVE;
    auto give = [&](auto khc) {
      constexpr int G = 8 * (1 - decltype(khc)::value);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const f32x4 v = {acc[mt][G + 4 * q], acc[mt][G + 4 * q + 1], acc[mt][G + 4 * q + 2], acc[mt][G + 4 * q + 3]};
          *reinterpret_cast<f32x4*>(mine + ((mt * 2 + q) * 64 + lane) * 16) = v;
        }
    };
    if (kh == 0) give(K0{}); else give(K1{});
    // The next tile has landed once at most the newest DMA and the two newest tiles' stores are outstanding (vector-memory
    // operations retire in issue order).  The barrier also says: every wave is done with `cur`, every exchange half is written.
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    wait_vmcnt<PPW + 2 * STORES>();
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (kh == 0) {
      float own[MT][8];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int i = 0; i < 8; ++i) own[mt][i] = acc[mt][i];
      issue_tile(t + NB * tstride, cur);
      finish(K0{}, own, theirs, y0, x0, true);
    } else if (!SKEW) {
      float own[MT][8];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int i = 0; i < 8; ++i) own[mt][i] = acc[mt][8 + i];
      issue_tile(t + NB * tstride, cur);
      finish(K1{}, own, theirs, y0, x0, true);
    } else {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int i = 0; i < 8; ++i) kept[mt][i] = acc[mt][8 + i];
      py0 = y0; px0 = x0; pt = t + NB * tstride; pbuf = cur;
      have_prev = true;
    }
  }
  if (SKEW && kh == 1 && have_prev) finish(K1{}, kept, xch + ((par ^ 1) * 8 + (wave8 ^ 4)) * XCH_WAVE, py0, px0, true);
  wait_vmcnt<0>();                                     // (the dummy DMA pieces of the last iterations write LDS too)
#endif
}

template <bool RELU_IN, bool POOL, bool SKEW>
int launch_ws2(const ConvArgs& a, hipStream_t st) {
  if (stv_set_max_lds(reinterpret_cast<const void*>(&conv_ws2_kernel<RELU_IN, POOL, SKEW>), LDS_BYTES) != STV_OK) return STV_ERR_LAUNCH;
  const int ntiles = ceil_div(a.W, TW) * ceil_div(a.H, TH);
  const int ncb = a.cout / 64;
  int dev = 0, cus = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
  int per_cb = cus / ncb;
  if (per_cb < 1) per_cb = 1;
  if (per_cb > ntiles) per_cb = ntiles;
  const int n_wg = per_cb * ncb;
  hipLaunchKernelGGL((conv_ws2_kernel<RELU_IN, POOL, SKEW>), dim3(n_wg), dim3(512), LDS_BYTES, st, a, n_wg);
  STV_CHECK_LAUNCH();
  return STV_OK;
}

}  // namespace

bool stv_conv_ws2_supported(const ConvArgs& a, int dtype, int taps) {
  // STV_CONV_WS2: 0 (default until measured faster) never, 1 the forward forms of the 64 -> 64 layer, 2 every 64 -> 64k layer
  const char* knob = getenv("STV_CONV_WS2");
  const int mode = knob ? atoi(knob) : 0;
  if (mode == 0 || getenv("STV_CONV_CFG") != nullptr) return false;
  if (dtype != STV_BF16 || taps != 9 || a.cin != 64 || a.cout % 64 != 0) return false;
  if (mode == 1 && a.cout != 64) return false;
  if ((a.flags & (STV_MASK | STV_ACCUM)) || a.x2 != nullptr || a.route_out != nullptr) return false;     // forward forms only
  if ((size_t)a.H * a.W * (size_t)(a.cout > 64 ? a.cout : 64) * 2 >= ((size_t)1 << 31)) return false;
  if (a.pool != nullptr && !(a.flags & STV_RELU_OUT)) return false;      // the packed pooling epilogue compares non-negative words
  return true;
}

int stv_conv_ws2_launch(const ConvArgs& a, hipStream_t st) {
  const bool relu = (a.flags & STV_RELU_IN) != 0, pool = a.pool != nullptr;
  const char* sk = getenv("STV_WS2_SKEW");             // A/B: 0 = both waves of a SIMD finish their tile at the same time
  if (sk != nullptr && atoi(sk) == 0) {
    if (relu) return pool ? launch_ws2<true, true, false>(a, st) : launch_ws2<true, false, false>(a, st);
    return pool ? launch_ws2<false, true, false>(a, st) : launch_ws2<false, false, false>(a, st);
  }
  if (relu) return pool ? launch_ws2<true, true, true>(a, st) : launch_ws2<true, false, true>(a, st);
  return pool ? launch_ws2<false, true, true>(a, st) : launch_ws2<false, false, true>(a, st);
}
