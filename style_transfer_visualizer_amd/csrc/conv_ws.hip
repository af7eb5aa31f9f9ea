// Weight-stationary persistent 3x3 convolution for the short-K layers (Cin = 64, bf16), NHWC.
//
// The general kernel (conv_igemm.hip) streams K through an LDS ring: with Cin = 64 that is four
// K-stages per output tile, and a workgroup spends more than a third of its life in the first DMA
// round trip and the final stores.  Here K is short enough to turn the loop
// inside out:
//
//  * all 9 x 64 x 64 weights of a 64-channel output block stay resident for the whole kernel - in
//    REGISTERS: a wave owns 32 output channels, i.e. 36 MFMA B-fragments = 144 VGPRs per lane
//    (four waves per workgroup, one per SIMD, 512 VGPRs each).  Weights never touch LDS, so the
//    LDS read traffic per MFMA is A fragments only: 18 x 16 B per 36 MFMAs against (12 + 9) per 18
//    in the general kernel's 8x64 tile, which ran into the 128 B/clk LDS limit;
//  * workgroups are persistent (one per CU) and walk over 8 x 32 pixel tiles.  The whole-K halo
//    tile of the NEXT tile (10 x 34 pixels x 128 B = 44 KB, LDS-DMA, zero-filled outside the
//    image) lands while the current one is multiplied: one barrier per tile, no K-stage barriers,
//    144 MFMAs back to back per wave;
//  * the tile's stores are issued after that barrier and drain under the next tile's MFMAs.
//
// Forward form: y = [relu](conv3x3([relu]x, w) + bias), optional fused MaxPool2d(2,2) + arg-max map
// (same epilogue arithmetic as conv_igemm.hip).  Backward form (DG): y = mask(z > 0) * conv3x3(dy, wb)
// + z . S^T with z the 64-channel pre-activation map that is both the ReLU mask and the Gram tap's
// features (stv_conv_igemm_dual with ref == x2): its 8 x 32 tile rides in LDS beside the halo tile.
//
// Fragment layouts, swizzle and accumulator layout are those of conv_igemm.hip: MFMA rows = output
// channels, columns = pixels; a lane of an accumulator holds 16 channels (4 groups of 4) of one pixel.
//
// Round 4: the kernel is a template on Cin.  Cin = 128 (conv2_2 and its backward, 128 -> 128): the four waves sit
// side by side along the output channels (32 each = all 128 of the layer: the input is staged once, not once per
// channel block), a wave keeps 8 stages x 9 taps = 72 fragments = 288 weight registers, and the tile shrinks to
// 2 rows x 32 pixels so that accumulators, three A column sets and the packed epilogue words still fit the 512
// registers of a lone wave (DESIGN.md 3.7 worked the budget out); the whole-K halo tile is 4 x 34 pixels x 256 B =
// 40 KB (three buffers forward, two + the 16-KB z tile backward).
#include <stdlib.h>

#include <mutex>

#include "stv_common.h"
#include "conv_args.h"

#ifndef STV_STORE_AUX
#define STV_STORE_AUX 0      // cache-policy bits of the output stores (diagnostic builds: 2 = nt, 16 = sc1)
#endif
#ifndef STV_WS_X_AUX
#define STV_WS_X_AUX 0       // cache-policy bits of the halo-tile DMA (diagnostic builds: 2 = nt)
#endif
#ifndef STV_WS_F_AUX
#define STV_WS_F_AUX 0       // ... of the z-tile DMA (backward form: last use of that map in the step)
#endif
#ifndef STV_WS_SWEEP
// Forward form behind a ReLU: 1 = clamp the staged halo tile ONCE in LDS (a sweep under the previous tile's MFMAs),
// 0 = every A fragment after its read.  Measured (round 5, profiles/r05_ws_sweep_ab.log): the sweep is 1-3 % SLOWER -
// isolated 116.5-118.2 against 114.8 us, step 2.599 against 2.562-2.607 ms at 1024^2 - so the default stays 0
// (profiles/EXPERIMENTS.md 3.9); the sweep is kept for A/B builds (tools/build_variant.sh).
#define STV_WS_SWEEP 0
#endif
#ifndef STV_WS_W_AGPR
#define STV_WS_W_AGPR 1      // resident weights pinned to AGPRs (0: wherever the register allocator puts them)
#endif
#ifndef STV_WS128_DEFAULT
#define STV_WS128_DEFAULT 1  // the 128 -> 128 layer on this kernel unless STV_CONV_WS128=0
#endif

namespace {

constexpr int TW = 32, IN_W = TW + 2;
constexpr int KB = 32, CK = 16;                    // a K-stage = 16 channels = 32 bytes per pixel
constexpr uint32_t kOob = 0x80000000u;             // >= num_records of every tensor accepted here

// Geometry by input channel count.  64: 2 x 2 waves (rows x channel halves), 8-row tiles, 64 output channels per
// workgroup.  128: 1 x 4 waves, 2-row tiles, 128 output channels per workgroup.
template <int CIN> struct WsGeom {
  static_assert(CIN == 64 || CIN == 128, "weight-stationary kernel: Cin 64 or 128");
  static constexpr int NSTAGE = CIN / CK;                    // K-stages of 16 channels
  static constexpr int WN = CIN == 64 ? 2 : 4, WM = 4 / WN;   // waves along output channels / along rows
  static constexpr int TH = CIN == 64 ? 8 : 2;               // tile rows
  static constexpr int MT = TH / WM;                         // rows per wave: 4 / 2
  static constexpr int COUT = 32 * WN;                       // output channels per workgroup
  static constexpr int IN_H = TH + 2, IN_PIX = IN_H * IN_W;  // halo tile
  static constexpr int IN_PIECES = (IN_PIX * 2 + 63) / 64;   // 1-KiB DMA pieces per stage (64 slots of 16 B)
  static constexpr int IN_STAGE = IN_PIECES * 1024;
  static constexpr int IN_BYTES = NSTAGE * IN_STAGE;         // 45,056 / 40,960
  static constexpr int F_PIX = TH * TW;
  static constexpr int F_PIECES = F_PIX * 2 / 64;            // = TH: one piece per tile row
  static constexpr int F_STAGE = F_PIECES * 1024;
  static constexpr int FSTAGES = COUT / CK;                  // the z tile has the layer's output channels
  static constexpr int F_BYTES = FSTAGES * F_STAGE;          // 32,768 / 16,384
  static constexpr int AROWS = MT + 2;
  static constexpr int NCOL = 3 * NSTAGE;                    // tap columns of a tile: (stage, dx)
  static constexpr int NCHUNK = 3 * MT;                      // deferred epilogue chunks: MT row stores + 2 MT pooling half-steps
  static constexpr int SPW = NSTAGE / 4;                     // K-stages each wave fetches per tile
  static constexpr int NSWEEP = IN_BYTES / 4096;             // 16-byte slots of a halo tile per thread (ReLU sweep): 11 / 10
  static_assert(IN_BYTES % 4096 == 0 && NSWEEP < NCOL, "ReLU sweep: whole rounds of 256 slots, one per tap column");
  static_assert(NCHUNK <= NCOL, "one epilogue chunk per tap column");
};

// tile buffers in LDS: the backward form (halo tile + z tile) fits two, the forward form three - the halo
// tile then streams in two tiles ahead of its use
template <int CIN, bool DG> struct WsLds {
  static constexpr int BUF = WsGeom<CIN>::IN_BYTES + (DG ? WsGeom<CIN>::F_BYTES : 0), NB = DG ? 2 : 3, BYTES = NB * BUF;
};
static_assert(WsLds<64, true>::BYTES <= 160 * 1024 && WsLds<64, false>::BYTES <= 160 * 1024, "LDS budget");
static_assert(WsLds<128, true>::BYTES <= 160 * 1024 && WsLds<128, false>::BYTES <= 160 * 1024, "LDS budget");

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

__device__ __forceinline__ bf16x8v relu_frag(bf16x8v v, uint32_t floor) {
  typedef __attribute__((ext_vector_type(8))) short s16x8;
  const s16x8 lo = (s16x8)((short)(floor & 0xFFFFu));
  return __builtin_bit_cast(bf16x8v, __builtin_elementwise_max(__builtin_bit_cast(s16x8, v), lo));
}

template <int CIN, bool DG, bool RELU_IN, bool POOL, bool MASKED = false, bool DUAL = false>
__global__ __launch_bounds__(256) void conv_ws_kernel(ConvArgs a, int n_workgroups, int diag) {
#if defined(__HIP_DEVICE_COMPILE__)
  using lds_ptr = __attribute__((address_space(3))) void*;
  using G = WsGeom<CIN>;
  constexpr int NSTAGE = G::NSTAGE, TH = G::TH, MT = G::MT, IN_PIX = G::IN_PIX, IN_PIECES = G::IN_PIECES, IN_STAGE = G::IN_STAGE,
                IN_BYTES = G::IN_BYTES, F_PIECES = G::F_PIECES, F_STAGE = G::F_STAGE, FSTAGES = G::FSTAGES, AROWS = G::AROWS,
                NCOL = G::NCOL, NCHUNK = G::NCHUNK, COUT = G::COUT, PIX_BYTES = CIN * 2;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6) & 3;     // scalar; also the first K-stage this wave fetches
  const int wm = wave / G::WN, wn = wave % G::WN;
  const int r = lane & 31, h = lane >> 5;

  // persistent mapping: this workgroup owns one 64-channel output block and every `tstride`-th tile
  const int tiles_x = (a.W + TW - 1) / TW;
  const int ntiles = tiles_x * ((a.H + TH - 1) / TH);
  const int ncb = a.cout / COUT;
  const int cb = (int)blockIdx.x % ncb;
  const int t_first = (int)blockIdx.x / ncb, tstride = n_workgroups / ncb;
  const int n0 = cb * COUT;
  const int nb = n0 + wn * 32;                                       // first output channel of this wave

  const bf16_t* __restrict__ xin = static_cast<const bf16_t*>(a.x);
  const bool w_blocked = (a.flags & STV_W_BLOCKED) != 0;
  const int x_bytes = a.H * a.W * PIX_BYTES;
  const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(xin), 0, x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_f = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<void*>(DG ? a.ref : nullptr), 0, DG ? a.H * a.W * COUT * 2 : 0, 0x00020000);

  // ---- resident weights: NSTAGE stages x 9 taps, one 16-byte fragment each (row = channel nb + r, k = 8h..8h+7)
  bf16x8v wreg[NSTAGE][9];
  {
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.w), 0, 9 * a.cout * CIN * 2, 0x00020000);
#pragma unroll
    for (int s = 0; s < NSTAGE; ++s)
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        const int n = nb + r;
        const int elem = w_blocked ? (((tap * NSTAGE + s) * a.cout + n) * CK + h * 8) : ((tap * a.cout + n) * CIN + s * CK + h * 8);
        wreg[s][tap] = __builtin_bit_cast(bf16x8v, __builtin_amdgcn_raw_buffer_load_b128(rs_w, (uint32_t)(elem * 2), 0, 0));
      }
#if STV_WS_W_AGPR
    // The resident weights are MFMA operands only: pinned to the accumulation half of the register file.  Left to the
    // allocator they are loaded as ordinary VGPR values, spilled to AGPRs under the pressure of a 476-register kernel and
    // copied back - four v_accvgpr_read_b32 in front of every group of four MFMAs, 144-227 per tile (round 5, read off
    // the ISA) - although the matrix instruction can read an AGPR operand directly.
#pragma unroll
    for (int s = 0; s < NSTAGE; ++s)
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) asm volatile("" : "+a"(wreg[s][tap]));
#endif
  }
  bf16x8v sreg[DG && DUAL ? FSTAGES : 1];          // DG with a fused 1x1 term: S rows of this wave's channels (plain [cout][cout])
  constexpr bool dual = DG && DUAL;                // (compile-time, like POOL: no branch between MFMA groups)
  if constexpr (dual) {
    const __amdgpu_buffer_rsrc_t rs_s = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.w2), 0, COUT * COUT * 2, 0x00020000);
#pragma unroll
    for (int s = 0; s < FSTAGES; ++s)
      sreg[s] = __builtin_bit_cast(bf16x8v, __builtin_amdgcn_raw_buffer_load_b128(rs_s, (uint32_t)((((nb - n0) + r) * COUT + s * CK + h * 8) * 2), 0, 0));
  }
  const __amdgpu_buffer_rsrc_t rs_b = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.bias), 0, a.bias != nullptr ? a.cout * 4 : 0, 0x00020000);
  f32x4 bias_v[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const u32x4 t = __builtin_amdgcn_raw_buffer_load_b128(rs_b, (uint32_t)((nb + 8 * j + 4 * h) * 4), 0, 0);
#pragma unroll
    for (int e = 0; e < 4; ++e) bias_v[j][e] = __uint_as_float(t[e]);
  }

  // ---- DMA bookkeeping.  Wave w fetches the K-stages w, w + 4, ... of a tile: IN_PIECES pieces of the halo tile
  // each (and F_PIECES of the centre tile of z per stage of z).  Slot v = piece * 64 + lane holds half (v & 1) ^ swizzle of pixel
  // v >> 1; its source offset is tile_base + rel, with rel a per-lane constant of the kernel.
  int in_rel[IN_PIECES], in_yx[IN_PIECES];
#pragma unroll
  for (int p = 0; p < IN_PIECES; ++p) {
    const int pix = p * 32 + (lane >> 1);
    const int half = (lane & 1) ^ ((pix >> 3) & 1);
    const int py = pix / IN_W, px = pix - py * IN_W;
    in_rel[p] = (py * a.W + px) * PIX_BYTES + half * 16;
    in_yx[p] = pix < IN_PIX ? ((py << 8) | px) : (0x3FFF << 8);      // slots past the tile: never in range
  }
  const int f_half = (lane & 1) ^ (((lane >> 1) >> 3) & 1);           // pixel = p * 32 + (lane >> 1): bit 3 is the lane's

  auto issue_tile = [&](int t, char* buf) {
    const int ty = t / tiles_x, tx = t - ty * tiles_x;
    const int y0 = ty * TH, x0 = tx * TW;
    const int base = ((y0 - 1) * a.W + (x0 - 1)) * PIX_BYTES;
#pragma unroll
    for (int sw = 0; sw < G::SPW; ++sw) {
      const int stage = wave + 4 * sw;
      char* dst = buf + stage * IN_STAGE;
#pragma unroll
      for (int p = 0; p < IN_PIECES; ++p) {
        const int gy = y0 - 1 + (in_yx[p] >> 8), gx = x0 - 1 + (in_yx[p] & 255);
        const bool ok = (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W;
        const uint32_t off = ok ? (uint32_t)(base + in_rel[p]) : kOob;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, (lds_ptr)(dst + p * 1024), 16, off, stage * KB, 0, STV_WS_X_AUX);
      }
    }
    if (DG) {
#pragma unroll
      for (int sw = 0; sw < FSTAGES / 4; ++sw) {
        const int stage = wave + 4 * sw;
        char* fdst = buf + IN_BYTES + stage * F_STAGE;
#pragma unroll
        for (int p = 0; p < F_PIECES; ++p) {
          const int gy = y0 + p, gx = x0 + (lane >> 1);
          const bool ok = gy < a.H && gx < a.W;
          const uint32_t off = ok ? (uint32_t)((gy * a.W + gx) * (COUT * 2) + f_half * 16) : kOob;
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_f, (lds_ptr)(fdst + p * 1024), 16, off, stage * KB, 0, STV_WS_F_AUX);
        }
      }
    }
  };
  // vector-memory operations one tile's DMA adds to this wave's queue (the counted wait below)
  constexpr int kTileOps = G::SPW * IN_PIECES + (DG ? (FSTAGES / 4) * F_PIECES : 0);

  // lane-constant LDS offsets of this lane's fragments inside a tile buffer (stage 0)
  int a_addr[3][AROWS];
#pragma unroll
  for (int dx = 0; dx < 3; ++dx)
#pragma unroll
    for (int j = 0; j < AROWS; ++j) {
      const int pix = (wm * MT + j) * IN_W + dx + r;
      a_addr[dx][j] = pix * KB + ((h ^ ((pix >> 3) & 1)) << 4);
    }

  const bool relu_out = (a.flags & STV_RELU_OUT) != 0;
  constexpr bool do_mask = DG && MASKED;
  const int out_bytes = a.H * a.W * a.cout * 2;
  // STV_POOL_ONLY (stv.h): nobody reads the full-resolution map again - a descriptor without records drops its stores
  const __amdgpu_buffer_rsrc_t rs_y = __builtin_amdgcn_make_buffer_rsrc(
      a.y, 0, (a.y != nullptr && !(POOL && (a.flags & STV_POOL_ONLY))) ? out_bytes : 0, 0x00020000);

  // ---- prologue: first two tiles in flight, weights in registers
  constexpr int NB = WsLds<CIN, DG>::NB;
  int t = t_first;
#pragma unroll
  for (int b = 0; b < NB; ++b)
    if (t + b * tstride < ntiles) issue_tile(t + b * tstride, smem + b * WsLds<CIN, DG>::BUF);
  wait_vmcnt<0>();
  __builtin_amdgcn_s_barrier();

  // ---- ReLU on the input, once per staged tile (forward form behind a tapped layer: the stored map is pre-ReLU) -----
  // Per fragment the packed max costs 4 VALU x 3 dx x (MT + 2) rows x NSTAGE stages per wave and tile (288 for Cin = 64:
  // every staged value is clamped six times over - three tap columns, two channel halves of the workgroup); as a sweep
  // over the landed tile it is one 16-byte slot per thread and round, NSWEEP rounds: 44 VALU + 22 LDS operations.  The
  // sweep of tile k + 1 rides under the MFMAs of tile k, one round per tap column (read in one column, clamp + write in
  // the next), and the barrier at the end of tile k publishes it.  For that the tile has to have LANDED when tile k
  // starts: the wait at the end of a tile covers everything this wave has in flight (vmcnt 0: the halo tile requested a
  // whole tile ago, and the previous tile's deferred stores - loads and stores retire out of order with respect to each
  // other, so a counted wait could not tell them apart).
  constexpr bool sweep = !DG && RELU_IN && (STV_WS_SWEEP != 0);
  bf16x8v swv[2];
  auto sweep_read = [&](char* buf, int c, int set) {
    swv[set] = *reinterpret_cast<const bf16x8v*>(buf + (c * 256 + tid) * 16);
  };
  auto sweep_write = [&](char* buf, int c, int set) {      // (whole-vector packed max, as relu_frag does for a fragment)
    *reinterpret_cast<bf16x8v*>(buf + (c * 256 + tid) * 16) = relu_frag(swv[set], 0u);
  };
  if constexpr (sweep) {                             // the first tile has nobody's MFMAs to hide behind
    if (t_first < ntiles) {
#pragma unroll
      for (int c = 0; c < G::NSWEEP; ++c) {
        sweep_read(smem, c, c & 1);
        sweep_write(smem, c, c & 1);
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the sweep's LDS writes are done before the others read
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  }

  // ---- forward form: the epilogue of tile k runs in the shadow of tile k + 1's MFMAs ------------------
  // One wave per SIMD has no other wave to overlap its epilogue with, but its own MFMAs leave seven of eight
  // issue slots to the vector ALU.  At the end of a tile only the rounding (acc -> packed bf16 words Pp, 64
  // VALU) happens in line; swaps, stores, the 2x2 pooling and its arg-max bytes are cut into twelve chunks
  // that the next tile's column loop carries along, one per tap column.
  typedef __attribute__((ext_vector_type(2))) short s16x2;
  const s16x2 relu_lo = (s16x2)((short)(relu_out ? 0 : -32768));
  uint32_t Pp[MT][8];
  int py0 = 0, px0 = 0;
  bool have_prev = false;
  constexpr bool pooling = POOL;                   // compile-time: the chunks below must stay branch-free
  const int Hp = a.H >> 1, Wp = a.W >> 1;
  const __amdgpu_buffer_rsrc_t rs_p = __builtin_amdgcn_make_buffer_rsrc(a.pool, 0, pooling ? Hp * Wp * a.cout * 2 : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_i = __builtin_amdgcn_make_buffer_rsrc(
      a.pool_idx, 0, (pooling && a.pool_idx != nullptr) ? Hp * Wp * a.cout : 0, 0x00020000);
  auto store_rows = [&](const uint32_t (&Q)[8], uint32_t poff, const __amdgpu_buffer_rsrc_t& rs_out) {
#pragma unroll
    for (int jp = 0; jp < 4; jp += 2) {
      // lanes 32-63 of the group-jp register <-> lanes 0-31 of the group-(jp+1) register
      const auto sx = __builtin_amdgcn_permlane32_swap(Q[jp], Q[jp + 1], false, false);
      const auto sy = __builtin_amdgcn_permlane32_swap(Q[4 + jp], Q[4 + jp + 1], false, false);
      const u32x4 out = {sx[0], sy[0], sx[1], sy[1]};
      const int nn = nb + 8 * jp + 8 * h;
      const uint32_t off = (poff != kOob && !(diag & 1)) ? poff + (uint32_t)(nn * 2) : kOob;
      __builtin_amdgcn_raw_buffer_store_b128(out, rs_out, off, 0, STV_STORE_AUX);
    }
  };
  auto right = [](uint32_t v) -> uint32_t {      // the neighbouring lane's word (quad_perm [1,0,3,2])
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, false);
  };
  auto gt = [](uint32_t best, uint32_t cand) -> uint32_t {    // bit 15 of each half: cand > best
    return __builtin_bit_cast(uint32_t, __builtin_bit_cast(s16x2, best) - __builtin_bit_cast(s16x2, cand)) & 0x80008000u;
  };
  auto mx = [](uint32_t x, uint32_t y) -> uint32_t {
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(s16x2, x), __builtin_bit_cast(s16x2, y)));
  };
  uint32_t Qp[MT / 2][8], codep[MT / 2][8];
  // chunk c of the pending tile's epilogue (c = 0..11): rows 0..3 of the full map, then the pooled map
  // (No branch on run-time state in here: the chunk has to sit in the same basic block as the MFMAs it
  // hides behind.  Before the first tile `have_prev` is false and every store is aimed out of range.)
  auto deferred = [&](int c) {
    if (c >= NCHUNK) return;
    if (c < MT) {
      const int gy = py0 + wm * MT + c, gx = px0 + r;
      store_rows(Pp[c], (have_prev && gy < a.H && gx < a.W) ? (uint32_t)(((gy * a.W + gx) * a.cout) * 2) : kOob, rs_y);
      return;
    }
    if (!pooling) return;
    // Fused MaxPool2d(2,2) + arg-max byte map (stv.h: stv_conv_igemm_pool), on the STORED values: the
    // packed bf16 words.  They are >= 0 here (the pool only rides behind a ReLU), so a bf16 compares like
    // its 15-bit integer pattern and `b > a` is the sign of the packed difference a - b: two channels per
    // instruction.  Window scan order (torch's first-maximum rule): top-left, top-right, bottom-left,
    // bottom-right; the right column is the neighbouring lane.  Two of the 16 (window row, word) units per chunk.
#pragma unroll
    for (int u = 2 * (c - MT); u < 2 * (c - MT) + 2; ++u) {
      const int mp = u >> 3, q = u & 7;
      const uint32_t tl = Pp[2 * mp][q], bl = Pp[2 * mp + 1][q];
      const uint32_t tr = right(tl), br = right(bl);
      const uint32_t c1 = gt(tl, tr), m1 = mx(tl, tr);
      const uint32_t c2 = gt(m1, bl), m2 = mx(m1, bl);
      const uint32_t c3 = gt(m2, br), m3 = mx(m2, br);
      Qp[mp][q] = m3;
      // position of the first maximum: c3 ? 3 : c2 ? 2 : c1 ? 1 : 0  ->  bit1 = c2 | c3, bit0 = c3 | (c1 & ~c2)
      const uint32_t b1 = c2 | c3;
      const uint32_t b0 = (c2 & c3) | (~c2 & (c1 | c3));
      // bit 2: the winner is positive (!= 0): adding 0x7FFF carries into bit 15 of a non-zero half
      typedef __attribute__((ext_vector_type(2))) unsigned short u16x2;
      const uint32_t pos = __builtin_bit_cast(uint32_t, __builtin_bit_cast(u16x2, m3) + (u16x2)((unsigned short)0x7FFF)) & 0x80008000u;
      codep[mp][q] = (b0 >> 15) | (b1 >> 14) | (pos >> 13);          // per half: a 3-bit code in bits 0-2 / 16-18
    }
    if (((c - MT) & 3) == 3) {                  // all eight words of a window row are done: store it
      const int mp = (c - MT) >> 2;
      const int gyp = ((py0 + wm * MT) >> 1) + mp, gxp = (px0 + r) >> 1;
      const bool pix_ok = have_prev && (r & 1) == 0 && gyp < Hp && gxp < Wp;
      store_rows(Qp[mp], pix_ok ? (uint32_t)(((gyp * Wp + gxp) * a.cout) * 2) : kOob, rs_p);
      {                                              // (without a map the descriptor has no records: the stores drop)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          // channels e = 0,1 come from code[j] (bytes 0 and 2), e = 2,3 from code[4 + j]
          const uint32_t word = __builtin_amdgcn_perm(codep[mp][4 + j], codep[mp][j], 0x06040200u);
          const int nn = nb + 8 * j + 4 * h;
          const uint32_t off = (pix_ok && !(diag & 1)) ? (uint32_t)((gyp * Wp + gxp) * a.cout + nn) : kOob;
          __builtin_amdgcn_raw_buffer_store_b32(word, rs_i, off, 0, 0);
        }
      }
    }
  };

  // A fragments: three column sets in flight.  The first two columns of a tile are requested as soon as the tile
  // has landed - for every tile but the first that is right behind the barrier at the end of the previous
  // one, so their LDS round trip runs under that tile's rounding VALU instead of in front of the first MFMA.
  bf16x8v af[3][AROWS];
  auto load_col_from = [&](const char* buf, int col, int set) {          // col = stage * 3 + dx
    const int s = col / 3, dx = col - s * 3;
#pragma unroll
    for (int j = 0; j < AROWS; ++j)
      af[set][j] = *reinterpret_cast<const bf16x8v*>(buf + s * IN_STAGE + a_addr[dx][j]);
  };
  f32x16 acc0;                                       // C operand of a tile's first MFMAs: the bias (forward) / zero
#pragma unroll
  for (int i = 0; i < 16; ++i) acc0[i] = DG ? 0.0f : bias_v[i >> 2][i & 3];
  if (t < ntiles) {
    load_col_from(smem, 0, 0);
    load_col_from(smem, 1, 1);
  }
  int slot = 0;
  for (; t < ntiles; t += tstride, slot = (slot + 1 == NB) ? 0 : slot + 1) {
    char* const cur = smem + slot * WsLds<CIN, DG>::BUF;
    const int ty = t / tiles_x, tx = t - ty * tiles_x;
    const int y0 = ty * TH, x0 = tx * TW;

    // accumulators start at the bias (zero for a backward pass): the epilogue has no add left.  Not as 64
    // moves per tile: the first MFMA of each row block takes the bias registers as its C operand.
    f32x16 acc[MT];

    // ---- 4 stages x 3 columns x 3 rows of taps.  One wave per SIMD: nothing else hides an LDS round
    // trip, so the A fragments of a column are requested TWO columns ahead and (forward pass behind a
    // ReLU) clamped one column ahead, in the shadow of the 12 MFMAs in between.
    auto load_col = [&](int col, int set) { load_col_from(cur, col, set); };
    auto relu_col = [&](int set) {
      if (RELU_IN && !sweep) {
#pragma unroll
        for (int j = 0; j < AROWS; ++j) af[set][j] = relu_frag(af[set][j], 0u);
      }
    };
    relu_col(0);                                     // (columns 0 and 1 are on their way already)
    if (diag & 4) {                                  // (timing knock-out: no MFMA loop)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) acc[mt] = acc0;
    }
    if (!(diag & 4))
#pragma unroll
    for (int col = 0; col < NCOL; ++col) {
      const int s = col / 3, dx = col - s * 3;
      if (col + 2 < NCOL) load_col(col + 2, (col + 2) % 3);
      if constexpr (sweep) {                           // the NEXT tile's buffer (no next tile: a buffer nobody reads again)
        char* const nxt_buf = smem + ((slot + 1 == NB) ? 0 : slot + 1) * WsLds<CIN, DG>::BUF;
        if (col < G::NSWEEP) sweep_read(nxt_buf, col, col & 1);
        if (col >= 1 && col - 1 < G::NSWEEP) sweep_write(nxt_buf, col - 1, (col - 1) & 1);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int dy = 0; dy < 3; ++dy)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
          acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wreg[s][dy * 3 + dx], af[col % 3][mt + dy],
                                                            (col == 0 && dy == 0) ? acc0 : acc[mt], 0, 0, 0);
      if (col + 1 < NCOL) relu_col((col + 1) % 3);
      if (!DG || col < MT) deferred(col);              // the previous tile's epilogue, one chunk per column
      if (RELU_IN || ((!DG || col < MT) && col < NCHUNK)) {   // 3 MT MFMAs with the packed max / epilogue VALU in their shadow
#pragma unroll
        for (int k = 0; k < 3 * MT; ++k) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x002, DG ? 2 : 6, 0);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }

    if (DG) {
      // ReLU mask of the first term (z > 0), read from the z tile in the accumulator layout, then the
      // 1x1 term z . S^T onto the same accumulators - both before the tile buffer is handed back
      const char* fb = cur + IN_BYTES;
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const int fp = (wm * MT + mt) * TW + r;
        const int swz = (fp >> 3) & 1;
        if (do_mask) {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int off = (2 * wn + (j >> 1)) * F_STAGE + fp * KB + (((j & 1) ^ swz) << 4) + 8 * h;
            const uint2 m = *reinterpret_cast<const uint2*>(fb + off);
            acc[mt][4 * j + 0] = ((int)(m.x << 16) > 0) ? acc[mt][4 * j + 0] : 0.0f;
            acc[mt][4 * j + 1] = ((int)(m.x & 0xFFFF0000u) > 0) ? acc[mt][4 * j + 1] : 0.0f;
            acc[mt][4 * j + 2] = ((int)(m.y << 16) > 0) ? acc[mt][4 * j + 2] : 0.0f;
            acc[mt][4 * j + 3] = ((int)(m.y & 0xFFFF0000u) > 0) ? acc[mt][4 * j + 3] : 0.0f;
          }
        }
        if constexpr (dual) {
#pragma unroll
          for (int s = 0; s < FSTAGES; ++s) {
            const bf16x8v zf = *reinterpret_cast<const bf16x8v*>(fb + s * F_STAGE + fp * KB + ((h ^ swz) << 4));
            acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(sreg[s], zf, acc[mt], 0, 0, 0);
          }
        }
      }
      // the mask of row mt + 1 (48 VALU) in the shadow of the 1x1 MFMAs of row mt
#pragma unroll
      for (int k = 0; k < (dual ? FSTAGES : 0) * MT; ++k) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 12, 0);
      }
    }

    // ---- hand the buffer back: the next tile has landed, a later one starts streaming into this buffer.
    // Two buffers: everything this wave has in flight is drained (the DMA of the next tile and the
    // previous tile's stores were both issued a whole tile ago).  Three buffers: the DMA issued last
    // (two tiles ahead) may stay in flight - loads retire in order among themselves, so "at most that
    // DMA's IN_PIECES operations outstanding" implies every older load, i.e. the next tile, has landed.
    // The previous tile's deferred stores count too: they can only make the wait longer, never satisfied
    // early (the threshold is exactly the newest DMA's size).
    const bool more = t + NB * tstride < ntiles && !(diag & 2);
    // (STV_WS_SWEEP=2, diagnostic build: the sweep with the counted wait - NOT guaranteed to have the swept tile landed;
    //  isolates what the full drain costs)
    if ((!sweep || STV_WS_SWEEP == 2) && NB == 3 && t + 2 * tstride < ntiles && !(diag & 2)) wait_vmcnt<kTileOps>();
    else wait_vmcnt<0>();
    if constexpr (sweep) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the next tile's sweep: its writes were issued columns ago
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (NB == 2 && more) issue_tile(t + NB * tstride, cur);

    if (NB == 3 && more) issue_tile(t + NB * tstride, cur);      // forward form: the stores follow in the next tile's shadow
    if (t + tstride < ntiles) {                                  // the next tile's first two A columns (it has landed)
      const char* nxt = smem + ((slot + 1 == NB) ? 0 : slot + 1) * WsLds<CIN, DG>::BUF;
      load_col_from(nxt, 0, 0);
      load_col_from(nxt, 1, 1);
    }
    // ---- rounding in line: acc -> packed bf16 words (ReLU after the rounding, on the packed words: a negative
    // bf16 is a negative int16; the bias already sits in the accumulators).  P[mt][0..3] = channel pairs (0,1)
    // of groups j, P[mt][4..7] = pairs (2,3): a lane holds channels nb + 8j + 4h + e of pixel (row mt, column r).
    {
      if (diag & 4) {                                  // (timing knock-out without the MFMA loop: nothing carried the chunks)
#pragma unroll
        for (int c = 0; c < (DG ? MT : NCHUNK); ++c) deferred(c);
      }
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          Pp[mt][j] = __builtin_bit_cast(uint32_t, __builtin_elementwise_max(
              __builtin_bit_cast(s16x2, pack_bf16x2(acc[mt][4 * j + 0], acc[mt][4 * j + 1])), relu_lo));
          Pp[mt][4 + j] = __builtin_bit_cast(uint32_t, __builtin_elementwise_max(
              __builtin_bit_cast(s16x2, pack_bf16x2(acc[mt][4 * j + 2], acc[mt][4 * j + 3])), relu_lo));
        }
      py0 = y0; px0 = x0;
      have_prev = true;
    }
  }
  {                                                    // the last tile's epilogue has no next tile to hide behind
#pragma unroll
    for (int c = 0; c < (DG ? MT : NCHUNK); ++c) deferred(c);
  }
  // every DMA issued was waited for inside the loop (the last two iterations issue none)
#endif
}

int device_cus() {
  static std::mutex mu;
  static int cus[64];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
  std::lock_guard<std::mutex> lk(mu);
  if (cus[dev] == 0) {
    hipDeviceProp_t p;
    cus[dev] = (hipGetDeviceProperties(&p, dev) == hipSuccess && p.multiProcessorCount > 0) ? p.multiProcessorCount : 256;
  }
  return cus[dev];
}

template <int CIN, bool DG, bool RELU_IN, bool POOL, bool MASKED = false, bool DUAL = false>
int launch_ws(const ConvArgs& a, hipStream_t st) {
  using G = WsGeom<CIN>;
  if (stv_set_max_lds(reinterpret_cast<const void*>(&conv_ws_kernel<CIN, DG, RELU_IN, POOL, MASKED, DUAL>), WsLds<CIN, DG>::BYTES) != STV_OK) return STV_ERR_LAUNCH;
  const int ntiles = ceil_div(a.W, TW) * ceil_div(a.H, G::TH);
  const int ncb = a.cout / G::COUT;
  int per_cb = device_cus() / ncb;                   // one persistent workgroup per CU
  if (per_cb < 1) per_cb = 1;
  if (per_cb > ntiles) per_cb = ntiles;
  const int n_wg = per_cb * ncb;
  const char* dg = getenv("STV_WS_DIAG");            // timing experiments only (results are then wrong)
  hipLaunchKernelGGL((conv_ws_kernel<CIN, DG, RELU_IN, POOL, MASKED, DUAL>), dim3(n_wg), dim3(256), (WsLds<CIN, DG>::BYTES), st, a, n_wg, dg ? atoi(dg) : 0);
  STV_CHECK_LAUNCH();
  return STV_OK;
}

template <int CIN>
int launch_ws_cin(const ConvArgs& a, hipStream_t st) {
  const bool has_f = (a.flags & STV_MASK) != 0 || a.x2 != nullptr;
  if (!has_f) {
    const bool relu = (a.flags & STV_RELU_IN) != 0, pool = a.pool != nullptr;
    if (relu) return pool ? launch_ws<CIN, false, true, true>(a, st) : launch_ws<CIN, false, true, false>(a, st);
    return pool ? launch_ws<CIN, false, false, true>(a, st) : launch_ws<CIN, false, false, false>(a, st);
  }
  ConvArgs b = a;
  if (b.ref == nullptr) b.ref = b.x2;              // the z tile is fetched through `ref`
  const bool masked = (b.flags & STV_MASK) != 0, dual = b.x2 != nullptr;
  if (masked) return dual ? launch_ws<CIN, true, false, false, true, true>(b, st) : launch_ws<CIN, true, false, false, true, false>(b, st);
  return launch_ws<CIN, true, false, false, false, true>(b, st);       // has_f without a mask: the 1x1 term is there
}

}  // namespace

bool stv_conv_ws_supported(const ConvArgs& a, int dtype, int taps) {
  // A/B knob; forcing a tile configuration of the general kernel (STV_CONV_CFG) also means: use that kernel
  // STV_CONV_WS: 0 = never, 2 = every supported shape, default (1) = where it measured faster than the
  // general kernel: the 64 -> 64 layers (conv1_2 forward with its pooling epilogue, and its backward) and - round 4,
  // STV_CONV_WS128 (default from the measurement recorded in DESIGN.md) - the 128 -> 128 layer (conv2_2 and its backward)
  const char* knob = getenv("STV_CONV_WS");
  const int mode = knob ? atoi(knob) : 1;
  if (mode == 0 || getenv("STV_CONV_CFG") != nullptr) return false;
  if (dtype != STV_BF16 || taps != 9 || (a.cin != 64 && a.cin != 128)) return false;
  const int cout_wg = a.cin == 64 ? 64 : 128;        // output channels of one workgroup (WsGeom::COUT)
  if (a.cout % cout_wg != 0) return false;
  if (a.cin == 128) {
    // STV_CONV_WS128: 0 never, 2 every 128 -> 128 launch, 1 (default) where it measured faster than the general
    // kernel (round 4, one box, per-op times inside the step): the BACKWARD form (masked dgrad + Gram term) from 8
    // tiles per workgroup up - 1024^2: 86.5 against 97.8 us; at 512^2 (4 tiles per workgroup) 29.1 against 29.3, and
    // the forward form is no faster at either size (80.2 against 78.8 us, 27.0 against 23.4): persistence has
    // little to hide once the general kernel runs two workgroups per CU on this K (8 stages)
    const char* k128 = getenv("STV_CONV_WS128");
    const int mode128 = k128 ? atoi(k128) : STV_WS128_DEFAULT;
    if (mode128 == 0 || a.cout != 128) return false;
    const bool backward = (a.flags & STV_MASK) != 0 || a.x2 != nullptr;
    const long tiles = (long)ceil_div(a.W, TW) * ceil_div(a.H, WsGeom<128>::TH);
    if (mode128 == 1 && !(backward && tiles >= 8L * device_cus())) return false;
  }
  if (mode == 1 && a.cout != cout_wg) return false;
  if (a.flags & STV_ACCUM) return false;
  const size_t cmax = (size_t)(a.cout > a.cin ? a.cout : a.cin);
  if ((size_t)a.H * a.W * cmax * 2 >= ((size_t)1 << 31)) return false;
  if (a.pool != nullptr && !(a.flags & STV_RELU_OUT)) return false;     // the packed pooling epilogue compares non-negative words
  const bool has_f = (a.flags & STV_MASK) != 0 || a.x2 != nullptr;
  if (has_f) {
    // the z tile in LDS is as wide as the workgroup's output channels and serves both as ReLU mask and as the 1x1 term's input
    if (a.cout != cout_wg || a.pool != nullptr) return false;
    if (a.x2 != nullptr && (a.cin2 != cout_wg || a.w2 == nullptr)) return false;
    if ((a.flags & STV_MASK) && a.x2 != nullptr && a.ref != a.x2) return false;
    if (a.flags & (STV_RELU_IN | STV_RELU_OUT)) return false;
    if (a.bias != nullptr) return false;
  }
  return true;
}

int stv_conv_ws_launch(const ConvArgs& a, hipStream_t st) {
  return a.cin == 64 ? launch_ws_cin<64>(a, st) : launch_ws_cin<128>(a, st);
}
