// The implicit-GEMM convolution kernel template and its launcher (description: conv_igemm.hip).
// A header so that the tile configurations can be instantiated in more than one translation unit (conv_igemm.hip: the
// tiles on v_mfma_f32_32x32x16_bf16 / the fp32 tiles; conv_igemm16.hip: the tiles whose main loop runs on
// v_mfma_f32_16x16x32_bf16) and built in parallel.
#pragma once
#include <stdlib.h>

#include <type_traits>

#include "stv_common.h"
#include "conv_args.h"

#ifndef STV_STAMP          // (conv_igemm.hip defines the phase-stamp macro first in its diagnostic build)
#define STV_STAMP(k) do {} while (0)
#endif
#ifndef STV_DIAG   // diagnostic builds knock out parts of the main loop (results are then wrong; timing only)
#define STV_DIAG 0
#endif

#ifndef STV_HOLD_LAST
#define STV_HOLD_LAST 1
#endif
#ifndef STV_X_AUX
#define STV_X_AUX 0          // cache-policy bits of the input-tile DMA (diagnostic builds: 2 = nt); weights keep the default policy
#endif
#ifndef STV_STORE_AUX
#define STV_STORE_AUX 0      // cache-policy bits of the output stores (diagnostic builds: 2 = nt, 16 = sc1)
#endif

// Hint of the caller (the op-program executor): the weights the NEXT conv launch will read.  Consumed by the one
// launch that follows on this thread (defined in conv_igemm.hip).
extern thread_local const void* g_stv_next_w;
extern thread_local uint32_t g_stv_next_w_bytes;

namespace {


template <typename T, int TH_, int BN_, int WM_, int WN_, int TAPS_, int KS_ = 1, int NBUF_ = 3, bool M16_ = false>
struct Cfg {
  // M16: the main loop runs on v_mfma_f32_16x16x32_bf16 (bf16 only).  Same LDS image and DMA pieces, but two K-stages
  // (2 x 16 channels) are consumed together - lanes 0-31 of an operand read the first stage's slot, lanes 32-63 the
  // second's - from a ring of FOUR stage slots (two pairs), and the image is NOT swizzled: the 16-row fragments of
  // this shape are conflict-free on the plain 32-byte pitch (conv_mainloop16 below).
  static constexpr bool M16 = M16_;
  using Elem = T;
  static constexpr int TH = TH_, BN = BN_, WM = WM_, WN = WN_, TAPS = TAPS_;
  static constexpr int NWAVES = WM_ * WN_;           // waves of one K group
  // KS = 2: two wave groups share the output tile and split K between them (group g owns the
  // K-stages c = g mod 2, in LDS buffers of its own); their accumulators meet in the LDS C tile.
  // A layer too small to give every CU two workgroups gets its second wave per SIMD this way.
  static constexpr int KS = KS_;
  static constexpr int GT = 64 * NWAVES;             // threads of one K group
  static constexpr int TW = 32;
  static constexpr int KB = 32;                      // K bytes per pixel per stage = LDS row pitch
  static constexpr int CK = KB / (int)sizeof(T);     // channels per stage
  static constexpr int HALO = (TAPS == 9) ? 1 : 0;
  static constexpr int ND = (TAPS == 9) ? 3 : 1;     // taps per axis
  static constexpr int IN_H = TH + 2 * HALO, IN_W = TW + 2 * HALO;
  static constexpr int IN_PIX = IN_H * IN_W;
  static constexpr int W_ROWS = TAPS * BN;
  static constexpr int MT = TH / WM;                 // image rows (32-pixel MFMA row blocks) per wave
  static constexpr int NT = BN / WN / 32;
  static constexpr int AROWS = MT + 2 * HALO;        // halo-tile rows a wave reads per horizontal tap
  // BIG: eight 32x32 accumulator blocks per wave (128 registers of the 256 a wave has at two waves per SIMD): the bias
  // is fetched after the main loop instead of being held through it (32 registers) and no tap step is held back across
  // the stage barrier (24) - with them the 16x128 tile spilled 60 registers
  static constexpr bool BIG = MT * NT >= 8;
  static constexpr int THREADS = GT * KS;
  // DMA pieces (one wave-instruction = 64 slots of 16 B = 32 rows): the halo tile rounded up to
  // whole pieces, then the weight rows; every wave of a group issues PPW pieces per stage (the
  // surplus ones of the last round are aimed at a spare KiB with a zero-record descriptor)
  static constexpr int IN_PIECES = (IN_PIX * 2 + 63) / 64;
  static constexpr int W_PIECES = (W_ROWS * 2 + 63) / 64;
  static constexpr int PIECES = IN_PIECES + W_PIECES;
  static constexpr int PPW = (PIECES + NWAVES - 1) / NWAVES;
  static constexpr int IN_BYTES = IN_PIECES * 1024;
  static constexpr int SPARE_OFF = PIECES * 1024;
  static constexpr int STAGE_BYTES = SPARE_OFF + 1024;
  // LDS ring: 3 = the DMA runs two stages ahead (one workgroup per CU has to hide its own
  // latencies); 2 = one stage ahead at 2/3 of the LDS, so that two workgroups share a CU and
  // cover each other's prologue, epilogue and waits
  static constexpr int NBUF = NBUF_;
  static constexpr int BM = TH * TW;
  static constexpr int CS = BN + 4;                  // C-tile pitch in floats
  static constexpr int C_BYTES = BM * CS * 4;
  static constexpr int RING_BYTES = NBUF * KS * STAGE_BYTES;
  // the fp32 C tile only exists where two K groups merge their partial sums
  static constexpr int LDS_BYTES = (KS == 1 || RING_BYTES > C_BYTES) ? RING_BYTES : C_BYTES;
  static_assert(NWAVES * KS == 4 || NWAVES * KS == 8, "4 or 8 waves per workgroup");
  static_assert(KS == 1 || KS == 2, "K split");
  static_assert(NBUF >= 2 && NBUF <= 6, "ring depth");
  static_assert(!M16 || (NBUF == 4 && sizeof(T) == 2), "16x16x32 main loop: bf16, two pairs of stage slots");
  static_assert(TH % WM == 0 && BN % (WN * 32) == 0, "tile split");
  static_assert(BN % 16 == 0, "swizzle period");
  static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
};

template <typename T> struct Frag;
template <> struct Frag<bf16_t> { using type = bf16x8v; };
template <> struct Frag<float> { using type = f32x4; };

template <typename T>
__device__ __forceinline__ void mma(const typename Frag<T>::type& a,
                                    const typename Frag<T>::type& b, f32x16& acc);
template <>
__device__ __forceinline__ void mma<bf16_t>(const bf16x8v& a, const bf16x8v& b, f32x16& acc) {
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
}
template <>
__device__ __forceinline__ void mma<float>(const f32x4& a, const f32x4& b, f32x16& acc) {
#pragma unroll
  for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], b[j], acc, 0, 0, 0);
}

// ReLU of a fragment as a packed integer max against `floor` (0: ReLU on, INT_MIN pattern: off)
__device__ __forceinline__ bf16x8v relu_frag(bf16x8v v, uint32_t floor) {
  typedef __attribute__((ext_vector_type(8))) short s16x8;
  const s16x8 lo = (s16x8)((short)(floor & 0xFFFFu));                  // splat
  return __builtin_bit_cast(bf16x8v, __builtin_elementwise_max(__builtin_bit_cast(s16x8, v), lo));
}
__device__ __forceinline__ f32x4 relu_frag(f32x4 v, uint32_t floor) {
  typedef __attribute__((ext_vector_type(4))) int i32x4;
  const i32x4 lo = (i32x4)((int)floor);
  return __builtin_bit_cast(f32x4, __builtin_elementwise_max(__builtin_bit_cast(i32x4, v), lo));
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// One pass of the implicit GEMM over a whole K range: streams `ph` (input tensor, weight tensor,
// their K extent) through the LDS ring and accumulates into `acc`.  The kernel runs it once for
// a plain convolution and a second time - 1x1 geometry, other tensors, same accumulators - for
// the fused Gram-backward term.
template <typename T>
struct Phase {
  const T* x;
  const T* w;
  int cin;
  bool w_blocked;
  uint32_t relu_floor;
  int k_first, k_count;      // the K-stages this workgroup walks (all of them unless K is split across workgroups)
};
struct Geom {
  int H, W, cout, x0, y0, n0;
  int lane, wave, grp, wm, wn, r, h;
};

// RELU: the input passes through a ReLU while it is read (STV_RELU_IN).  A compile-time switch, not a
// scalar floor: the packed max costs four VALU per A fragment (48-72 per K-stage) and the tiles are
// sensitive to exactly that (round-2 A/B: 12 extra VALU per halo row made them 2-18 % slower), while
// only the forward convs behind a tapped (pre-ReLU) layer need it - none of the dgrads do.
template <typename C, bool RELU>
__device__ __forceinline__ void conv_mainloop(const Phase<typename C::Elem>& ph, const Geom& gm, char* smem,
                                              f32x16 (&acc)[C::MT][C::NT]) {
  using T = typename C::Elem;
  using FragT = typename Frag<T>::type;
  using lds_ptr = __attribute__((address_space(3))) void*;
  constexpr int kVec = elem_traits<T>::kVec;
  const int lane = gm.lane, wave = gm.wave, grp = gm.grp, wm = gm.wm, wn = gm.wn, r = gm.r, h = gm.h;
  const int x0 = gm.x0, y0 = gm.y0, n0 = gm.n0;
  const int nchunks = ph.cin / C::CK;
  // ---- DMA pieces of this wave: per-lane source byte offsets (out of range -> zero fill) ----
  constexpr uint32_t kOob = 0x80000000u;   // >= num_records for every tensor this kernel accepts
  const int x_bytes = gm.H * gm.W * ph.cin * (int)sizeof(T);
  const int w_bytes = C::TAPS * gm.cout * ph.cin * (int)sizeof(T);
  // bytes from one K-stage to the next: 32 along a pixel's (or plain weight row's) channels,
  // a whole [cout][CK] slab in the K-blocked weight layout
  const int w_stride = ph.w_blocked ? gm.cout * C::KB : C::KB;
  // piece j of this wave is piece j * NWAVES + wave of the stage: input pieces first, then weights
  auto piece_id = [&](int j) { return j * C::NWAVES + wave; };                  // wave-uniform
  uint32_t p_off[C::PPW];
#pragma unroll
  for (int j = 0; j < C::PPW; ++j) {
    const int g = piece_id(j);
    if (g < C::IN_PIECES) {
      const int v = g * 64 + lane;
      const int pix = v >> 1;
      const int half = (v & 1) ^ ((pix >> 3) & 1);                 // swizzle on the source side
      const int py = pix / C::IN_W, px = pix - py * C::IN_W;
      const int gy = y0 + py - C::HALO, gx = x0 + px - C::HALO;
      const bool ok = pix < C::IN_PIX && gy >= 0 && gy < gm.H && gx >= 0 && gx < gm.W;
      p_off[j] = ok ? (uint32_t)(((gy * gm.W + gx) * ph.cin + half * kVec) * (int)sizeof(T)) : kOob;
    } else {
      const int v = (g - C::IN_PIECES) * 64 + lane;
      const int row = v >> 1;
      const int half = (v & 1) ^ ((row >> 3) & 1);
      const int tap = row / C::BN, nn = row - tap * C::BN;
      const bool ok = row < C::W_ROWS && (n0 + nn) < gm.cout;
      const int elem = ph.w_blocked ? ((tap * nchunks * gm.cout + n0 + nn) * C::CK + half * kVec)
                                 : ((tap * gm.cout + n0 + nn) * ph.cin + half * kVec);
      p_off[j] = ok ? (uint32_t)(elem * (int)sizeof(T)) : kOob;
    }
  }
  // group g walks the K-stages g, g + KS, ...: `l` counts its own stages (every group runs the
  // same number of rounds so that the workgroup barriers match; a round past the end stages zeros)
  const int nrounds = (ph.k_count + C::KS - 1) / C::KS;
  char* const ring = smem + grp * (C::NBUF * C::STAGE_BYTES);
  // issue piece j of this wave's share of round l into ring buffer `buf`
  auto dma = [&](int j, int l, char* buf) {
    const int g = piece_id(j);
    const int own = l * C::KS + grp;               // index among this workgroup's stages
    const int stage = ph.k_first + own;
    const bool in = g < C::IN_PIECES;
    const bool live = g < C::PIECES && own < ph.k_count && !(STV_DIAG & 1);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<T*>(in ? ph.x : ph.w), 0, live ? (in ? x_bytes : w_bytes) : 0, 0x00020000);
    char* dst = buf + (g < C::PIECES ? g * 1024 : C::SPARE_OFF);
    if (STV_X_AUX != 0 && in) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr)dst, 16, p_off[j], stage * C::KB, 0, STV_X_AUX);
    else __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr)dst, 16, p_off[j], stage * (in ? C::KB : w_stride), 0, 0);
  };

  // lane-constant LDS byte offsets of this lane's fragments inside a stage buffer.  A: one per
  // (horizontal tap, halo row) - the swizzle bit depends on the pixel index; B: the row index is
  // r plus multiples of 16, so one offset serves every tap
  int a_addr[C::ND][C::AROWS];
#pragma unroll
  for (int dx = 0; dx < C::ND; ++dx)
#pragma unroll
    for (int j = 0; j < C::AROWS; ++j) {
      const int pix = (wm * C::MT + j) * C::IN_W + dx + r;
      a_addr[dx][j] = pix * C::KB + ((h ^ ((pix >> 3) & 1)) << 4);
    }
  const int b_lane = C::IN_BYTES + (wn * (C::NT * 32) + r) * C::KB + ((h ^ ((r >> 3) & 1)) << 4);

  constexpr int NSTEP = C::ND * C::ND;
  // DMA pieces per tap step.  A two-deep ring under a workgroup that has the CU to itself (BIG) waits for the stage it
  // requested during the SAME stage: all of it is requested in the first half of the steps, the second half is its slack
  // (with two workgroups per CU - the other two-deep tiles - the partner covers that wait instead).
  constexpr int NISS = (C::BIG && C::NBUF == 2 && NSTEP > 1) ? (NSTEP + 1) / 2 : NSTEP;
  constexpr int PER = (C::PPW + NISS - 1) / NISS;
  // B fragments are fetched PFB steps ahead of their MFMAs: an LDS read takes ~190 cycles under
  // load, a step only MT*NT*32 of MFMA issue, so a lone wave on a SIMD needs the deeper queue
  constexpr int PFB = (C::MT * C::NT >= 4) ? 2 : 3;

  // One K-stage out of `cur`, while round l + 2 streams into `fill`.  The taps are walked column
  // by column (dx outer, dy inner): the MT+2 halo-tile rows a wave needs for one dx serve all
  // three dy, so a stage reads 3*(MT+2) A fragments instead of 9*MT.  Fragments are fetched ahead
  // of the MFMAs that use them.
  // The MFMAs of a stage's LAST tap step(s) are held back across the barrier (-DSTV_HOLD_LAST=n steps, 0 = off):
  // their operands sit in registers, so they can be issued behind the next stage's first LDS reads and cover
  // that round trip - right after a barrier both waves of a SIMD would otherwise wait for it with the matrix
  // pipe idle.  Same products in the same order on the same accumulators: results unchanged bit for bit.
  // (Before the first stage the held operands are zero: four MFMAs that add nothing.)
  // fp32 = the parity mode: BLOCKED summation.  One accumulator chain over all 9 Cin products of an output
  // (up to 4,608 sequential fp32 additions) left the fp32 gradient 2.6-6.6x further from its float64 value than
  // the reference's CPU path at 512^2 / 1024^2; here every K-stage (8 channels x 9 taps = 72 products) is summed
  // in a fresh accumulator that is then added to the running sum - chains of 72 and Cin/8 instead of 9 Cin.
  // Speed is not the point of this mode (no held-back step either: it would straddle two stage sums).
  constexpr bool BLOCKED = sizeof(T) == 4;
  constexpr int NHOLD = (BLOCKED || C::BIG) ? 0 : ((STV_HOLD_LAST < NSTEP) ? STV_HOLD_LAST : NSTEP);      // steps held back (0: none)
  FragT hold_a[NHOLD > 0 ? NHOLD : 1][C::MT], hold_b[NHOLD > 0 ? NHOLD : 1][C::NT];
#pragma unroll
  for (int q = 0; q < NHOLD; ++q) {
#pragma unroll
    for (int mt = 0; mt < C::MT; ++mt) hold_a[q][mt] = FragT{};
#pragma unroll
    for (int nt = 0; nt < C::NT; ++nt) hold_b[q][nt] = FragT{};
  }
  auto flush_held = [&]() {
#pragma unroll
    for (int q = 0; q < NHOLD; ++q)
#pragma unroll
      for (int mt = 0; mt < C::MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < C::NT; ++nt) mma<T>(hold_b[q][nt], hold_a[q][mt], acc[mt][nt]);
  };
  auto run_stage = [&](const char* cur, char* fill, int l) {
    FragT af[2][C::AROWS];
    FragT bf[PFB + 1][C::NT];
    f32x16 sacc[BLOCKED ? C::MT : 1][BLOCKED ? C::NT : 1];      // this stage's own sum (fp32 mode)
    if constexpr (BLOCKED) {
#pragma unroll
      for (int mt = 0; mt < C::MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < C::NT; ++nt)
#pragma unroll
          for (int i = 0; i < 16; ++i) sacc[mt][nt][i] = 0.0f;
    }
    auto load_a = [&](int dx, int j, int set) {
      const FragT v = *reinterpret_cast<const FragT*>(cur + a_addr[dx][j]);
      af[set][j] = RELU ? relu_frag(v, 0u) : v;
    };
    auto load_b = [&](int step) {
      const int tap = (step % C::ND) * C::ND + step / C::ND;     // dy * 3 + dx
#pragma unroll
      for (int nt = 0; nt < C::NT; ++nt)
        bf[step % (PFB + 1)][nt] = *reinterpret_cast<const FragT*>(cur + b_lane + (tap * C::BN + nt * 32) * C::KB);
    };
    load_b(0);
#pragma unroll
    for (int j = 0; j < C::AROWS; ++j) load_a(0, j, 0);
#pragma unroll
    for (int q = 1; q < PFB; ++q)
      if (q < NSTEP) load_b(q);
    if (NHOLD > 0) {
      __builtin_amdgcn_sched_barrier(0);
      flush_held();                              // the previous stage's last step, in the shadow of the reads above
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int step = 0; step < NSTEP; ++step) {
      const int dx = step / C::ND, dy = step % C::ND;
      if (step + PFB < NSTEP) load_b(step + PFB);
      if (dx + 1 < C::ND) {        // next column of A rows: first half at dy = 0, the rest at dy = 1
#pragma unroll
        for (int j = 0; j < C::AROWS; ++j)
          if ((j < (C::AROWS + 1) / 2 ? 0 : 1) == dy) load_a(dx + 1, j, (dx + 1) & 1);
      }
      // the next steps' LDS reads and this step's DMAs are issued first, then the MFMAs back to back
#pragma unroll
      for (int k = step * PER; k < (step + 1) * PER; ++k)
        if (k < C::PPW) dma(k, l + C::NBUF - 1, fill);
      __builtin_amdgcn_sched_barrier(0);
      if (step >= NSTEP - NHOLD) {
        constexpr int dummy = 0; (void)dummy;
        const int q = step - (NSTEP - NHOLD);
#pragma unroll
        for (int mt = 0; mt < C::MT; ++mt) hold_a[q][mt] = af[dx & 1][mt + dy];
#pragma unroll
        for (int nt = 0; nt < C::NT; ++nt) hold_b[q][nt] = bf[step % (PFB + 1)][nt];
      } else {
#pragma unroll
        for (int mt = 0; mt < C::MT; ++mt)
#pragma unroll
          for (int nt = 0; nt < C::NT; ++nt) {
            if constexpr (BLOCKED) mma<T>(bf[step % (PFB + 1)][nt], af[dx & 1][mt + dy], sacc[mt][nt]);
            else mma<T>(bf[step % (PFB + 1)][nt], af[dx & 1][mt + dy], acc[mt][nt]);   // D[cout][pixel]
          }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if constexpr (BLOCKED) {
#pragma unroll
      for (int mt = 0; mt < C::MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < C::NT; ++nt)
#pragma unroll
          for (int i = 0; i < 16; ++i) acc[mt][nt][i] += sacc[mt][nt][i];
    }
    // round l + 1 has landed once at most the pieces of the rounds after it are still in flight;
    // the barrier then also says every wave is done reading `cur`, which the next fill overwrites
    wait_vmcnt<(C::NBUF - 2) * C::PPW>();
    __builtin_amdgcn_s_barrier();
  };

#if defined(STV_PRIO_HALF)   // diagnostic build: static priority for the second-dispatched half of an eight-wave workgroup (the arbitration loser)
  if (C::NWAVES * C::KS == 8 && (int)(threadIdx.x >> 8) == 1) __builtin_amdgcn_s_setprio(1);
#endif
  // prologue: rounds 0 .. NBUF-2 in flight, round 0 landed
#pragma unroll
  for (int rnd = 0; rnd + 1 < C::NBUF; ++rnd)
#pragma unroll
    for (int k = 0; k < C::PPW; ++k) dma(k, rnd, ring + rnd * C::STAGE_BYTES);
  wait_vmcnt<(C::NBUF - 2) * C::PPW>();
  __builtin_amdgcn_s_barrier();
  STV_STAMP(1);

  // round c is computed out of ring slot c % NBUF while round c + NBUF - 1 streams into the slot before it
  int c = 0;
  for (; c + C::NBUF <= nrounds; c += C::NBUF) {
#pragma unroll
    for (int k = 0; k < C::NBUF; ++k)
      run_stage(ring + k * C::STAGE_BYTES, ring + ((k + C::NBUF - 1) % C::NBUF) * C::STAGE_BYTES, c + k);
  }
#pragma unroll
  for (int k = 0; k + 1 < C::NBUF; ++k)
    if (c + k < nrounds) run_stage(ring + k * C::STAGE_BYTES, ring + ((k + C::NBUF - 1) % C::NBUF) * C::STAGE_BYTES, c + k);
  if (NHOLD > 0) flush_held();                   // the last stage's held steps
  // the zero-fill DMAs of the rounds past the end still target the ring: drain them before the
  // C tile takes over the same LDS
  wait_vmcnt<0>();
  __syncthreads();

}

// ------------------------------------------------------------------------------------------------------------------
// The same pass on v_mfma_f32_16x16x32_bf16 (Cfg::M16).  Under this chip's power limit the 16x16x32 shape holds a higher
// clock than 32x32x16 at equal cycles per FLOP (MI355X_MICROARCH.md, DVFS give-back item 7).  K = 32 per MFMA = two of
// the 16-channel K-stages: a ROUND consumes a pair of stage slots, lane k-groups 0,1 (lanes 0-31) reading the first
// slot's two 16-byte halves, k-groups 2,3 (lanes 32-63) the second slot's - the pairing is nothing but a per-lane
// address.  Ring: two slot pairs; while round l is multiplied, round l + 1 streams into the other pair (all its DMAs
// issued in the first half of the round, `s_waitcnt vmcnt(0)` + barrier at its end).
// Operands: A = weights (16 couts x 32 k), B = pixels (32 k x 16 pixels); a lane of D holds couts 4 (l >> 4) + e of pixel
// l & 15.  Fragment of 16 rows on the 32-byte pitch: lanes (row r, k-group g) read byte 32 r + 16 (g & 1): every 16-lane
// service group of ds_read_b128 covers 8 consecutive rows x one half twice over distinct banks for ANY first row - no
// swizzle, so every (tap, row, block) offset is a compile-time immediate off one lane address.
// Loop order per round: dx outer; the three taps of a column keep their weight fragments in registers while the halo
// rows stream through (each row's two 16-pixel fragments serve up to three dy).
template <typename C, bool RELU>
__device__ __forceinline__ void conv_mainloop16(const Phase<bf16_t>& ph, const Geom& gm, char* smem,
                                                f32x4 (&acc)[C::MT][2][C::NT][2]) {
  using T = bf16_t;
  using FragT = bf16x8v;
  using lds_ptr = __attribute__((address_space(3))) void*;
  constexpr int kVec = 8;
  const int lane = gm.lane, wave = gm.wave, grp = gm.grp, wm = gm.wm, wn = gm.wn;
  const int x0 = gm.x0, y0 = gm.y0, n0 = gm.n0;
  const int nchunks = ph.cin / C::CK;              // 16-channel K-stages (even: the host checks cin % 32 == 0)
  const int npairs = nchunks >> 1;
  constexpr uint32_t kOob = 0x80000000u;
  const int x_bytes = gm.H * gm.W * ph.cin * (int)sizeof(T);
  const int w_bytes = C::TAPS * gm.cout * ph.cin * (int)sizeof(T);
  const int w_stride = ph.w_blocked ? gm.cout * C::KB : C::KB;
  auto piece_id = [&](int j) { return j * C::NWAVES + wave; };                  // wave-uniform
  uint32_t p_off[C::PPW];
#pragma unroll
  for (int j = 0; j < C::PPW; ++j) {
    const int g = piece_id(j);
    if (g < C::IN_PIECES) {
      const int v = g * 64 + lane;
      const int pix = v >> 1, half = v & 1;                      // plain image: no swizzle
      const int py = pix / C::IN_W, px = pix - py * C::IN_W;
      const int gy = y0 + py - C::HALO, gx = x0 + px - C::HALO;
      const bool ok = pix < C::IN_PIX && gy >= 0 && gy < gm.H && gx >= 0 && gx < gm.W;
      p_off[j] = ok ? (uint32_t)(((gy * gm.W + gx) * ph.cin + half * kVec) * (int)sizeof(T)) : kOob;
    } else {
      const int v = (g - C::IN_PIECES) * 64 + lane;
      const int row = v >> 1, half = v & 1;
      const int tap = row / C::BN, nn = row - tap * C::BN;
      const bool ok = row < C::W_ROWS && (n0 + nn) < gm.cout;
      const int elem = ph.w_blocked ? ((tap * nchunks * gm.cout + n0 + nn) * C::CK + half * kVec)
                                    : ((tap * gm.cout + n0 + nn) * ph.cin + half * kVec);
      p_off[j] = ok ? (uint32_t)(elem * (int)sizeof(T)) : kOob;
    }
  }
  // group g walks the stage PAIRS g, g + KS, ...
  const int nrounds = (npairs + C::KS - 1) / C::KS;
  char* const ring = smem + grp * (C::NBUF * C::STAGE_BYTES);
  constexpr int PAIR_BYTES = 2 * C::STAGE_BYTES;
  // piece k (0 .. 2 PPW - 1) of this wave's share of round l: stage 2 P + k / PPW of pair P = l KS + grp
  auto dma = [&](int k, int l, char* pair) {
    const int sub = k / C::PPW, j = k - sub * C::PPW;
    const int g = piece_id(j);
    const int stage = 2 * (l * C::KS + grp) + sub;
    const bool in = g < C::IN_PIECES;
    const bool live = g < C::PIECES && stage < nchunks && !(STV_DIAG & 1);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<T*>(in ? ph.x : ph.w), 0, live ? (in ? x_bytes : w_bytes) : 0, 0x00020000);
    char* dst = pair + sub * C::STAGE_BYTES + (g < C::PIECES ? g * 1024 : C::SPARE_OFF);
    if (STV_X_AUX != 0 && in) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr)dst, 16, p_off[j], stage * C::KB, 0, STV_X_AUX);
    else __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr)dst, 16, p_off[j], stage * (in ? C::KB : w_stride), 0, 0);
  };

  // this lane's fragment addresses inside a slot pair: everything else is an immediate
  const int px = lane & 15, kg = (lane >> 4) & 1, sub = lane >> 5;
  const int a_lane = sub * C::STAGE_BYTES + ((wm * C::MT) * C::IN_W + px) * C::KB + kg * 16;
  const int b_lane = sub * C::STAGE_BYTES + C::IN_BYTES + (wn * (C::NT * 32) + px) * C::KB + kg * 16;

  constexpr int U = C::ND * C::AROWS;              // row steps of a round: (dx, halo row j)
  constexpr int NISS = (U >= 2) ? U / 2 : 1;       // steps that issue the next round's DMAs (the first half)
  constexpr int PER = (2 * C::PPW + NISS - 1) / NISS;
  constexpr int PFP = (U >= 3) ? 2 : 1;            // halo rows are requested this many steps ahead of their MFMAs

  auto run_round = [&](const char* cur, char* fill, int l) {
    FragT wf[2][C::ND][C::NT][2];                  // weight fragments of a tap column [dx & 1][dy][nt][cb]
    FragT pf[PFP + 1][2];                          // halo-row fragments [step % (PFP + 1)][pb]
    auto load_w = [&](int dx, int dy) {
      const int tap = dy * C::ND + dx;
#pragma unroll
      for (int nt = 0; nt < C::NT; ++nt)
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
          wf[dx & 1][dy][nt][cb] = *reinterpret_cast<const FragT*>(cur + b_lane + (tap * C::BN + nt * 32 + cb * 16) * C::KB);
    };
    auto load_p = [&](int u) {
      const int dx = u / C::AROWS, j = u - dx * C::AROWS;
#pragma unroll
      for (int pb = 0; pb < 2; ++pb) {
        const FragT v = *reinterpret_cast<const FragT*>(cur + a_lane + (j * C::IN_W + dx + pb * 16) * C::KB);
        pf[u % (PFP + 1)][pb] = RELU ? relu_frag(v, 0u) : v;
      }
    };
#pragma unroll
    for (int dy = 0; dy < C::ND; ++dy) load_w(0, dy);
#pragma unroll
    for (int u = 0; u < PFP; ++u)
      if (u < U) load_p(u);
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int dx = u / C::AROWS, j = u - dx * C::AROWS;
      if (u + PFP < U) load_p(u + PFP);
      if (dx + 1 < C::ND && j < C::ND) load_w(dx + 1, j);          // the next column's taps, one per row step
#pragma unroll
      for (int k = u * PER; k < (u + 1) * PER; ++k)
        if (u < NISS && k < 2 * C::PPW) dma(k, l + 1, fill);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int dy = 0; dy < C::ND; ++dy) {
        const int mt = j - dy;
        if (mt < 0 || mt >= C::MT) continue;
#pragma unroll
        for (int nt = 0; nt < C::NT; ++nt)
#pragma unroll
          for (int cb = 0; cb < 2; ++cb)
#pragma unroll
            for (int pb = 0; pb < 2; ++pb)
              acc[mt][pb][nt][cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[dx & 1][dy][nt][cb], pf[u % (PFP + 1)][pb],
                                                                            acc[mt][pb][nt][cb], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    // round l + 1 has landed (all of it was requested in the first half of this round); the barrier also says every
    // wave is done reading `cur`, which the round after next overwrites
    wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
  };

#pragma unroll
  for (int k = 0; k < 2 * C::PPW; ++k) dma(k, 0, ring);
  wait_vmcnt<0>();
  __builtin_amdgcn_s_barrier();
  STV_STAMP(1);
  int l = 0;
  for (; l + 2 <= nrounds; l += 2) {
    run_round(ring, ring + PAIR_BYTES, l);
    run_round(ring + PAIR_BYTES, ring, l + 1);
  }
  if (l < nrounds) run_round(ring, ring + PAIR_BYTES, l);
  wait_vmcnt<0>();
  __syncthreads();
}

// 16x16 accumulator blocks -> the 32x32 accumulator layout the epilogue is written for (lane (r, h): pixel r, channels
// 8 j + 4 h + e): per register pair (pixel block 0 / 1) one v_permlane16_swap (lane bit 4 <-> pixel block) and one
// v_permlane32_swap (lane bit 5 <-> channel-group bit): 16 swaps per 32 x 32 block, once per tile.
template <typename C>
__device__ __forceinline__ void acc16_to_acc32(const f32x4 (&a16)[C::MT][2][C::NT][2], f32x16 (&acc)[C::MT][C::NT], bool add) {
#pragma unroll
  for (int mt = 0; mt < C::MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < C::NT; ++nt)
#pragma unroll
      for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const auto s1 = __builtin_amdgcn_permlane16_swap(__float_as_uint(a16[mt][0][nt][cb][e]), __float_as_uint(a16[mt][1][nt][cb][e]), false, false);
          const auto s2 = __builtin_amdgcn_permlane32_swap(s1[0], s1[1], false, false);
          const float v0 = __uint_as_float(s2[0]), v1 = __uint_as_float(s2[1]);
          if (add) {
            acc[mt][nt][8 * cb + e] += v0;
            acc[mt][nt][8 * cb + 4 + e] += v1;
          } else {
            acc[mt][nt][8 * cb + e] = v0;
            acc[mt][nt][8 * cb + 4 + e] = v1;
          }
        }
}

template <typename C, bool RELU>
__global__ __launch_bounds__(C::THREADS) void conv_igemm_kernel(ConvArgs a) {
#if defined(__HIP_DEVICE_COMPILE__)   // the host pass only needs the launch stub (LDS address-space casts are device-only)
  using T = typename C::Elem;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave_wg = __builtin_amdgcn_readfirstlane(tid >> 6) & (C::NWAVES * C::KS - 1);   // scalar
  const int grp = wave_wg / C::NWAVES;               // K group
  const int wave = wave_wg % C::NWAVES;              // wave within the group
  const int wm = wave / C::WN, wn = wave % C::WN;
  const int r = lane & 31, h = lane >> 5;

  const int tiles_x = (a.W + C::TW - 1) / C::TW;
  // Block -> (spatial tile, channel block).  Workgroups are dealt to the 8 XCDs round-robin and each
  // XCD has its own L2, so the channel blocks of one spatial tile are given ids 8 apart: they run
  // on the same XCD at about the same time and the second one finds the tile's input in that L2
  // (with a plain 2-D grid every channel block streamed the whole input from HBM again).
  const int ntiles = tiles_x * ((a.H + C::TH - 1) / C::TH);
  const int ny = (a.cout + C::BN - 1) / C::BN;
  // K split across workgroups (ConvArgs::xk): blocks b and b + 8 - same XCD, dispatched back to back - are the two K
  // halves of one output tile; the grid is padded to whole groups of eight tiles
  int blk = (int)blockIdx.x, kh = 0;
  constexpr bool kXkTile = C::KS == 1 && !C::M16 && sizeof(T) == 2;
  const bool xk = kXkTile && a.xk == 2;
  if (xk) {
    kh = (blk >> 3) & 1;
    blk = ((blk >> 4) << 3) | (blk & 7);
    if (blk >= ntiles * ny) return;                  // (padding blocks: before any barrier)
  }
  int tile, yb;
  {
    const int round = 8 * ny, b = blk;
    const int grp8 = b / round, within = b - grp8 * round;
    const int left = ntiles - grp8 * 8;                    // tiles in this group of (up to) eight
    const int span = left < 8 ? left : 8;
    const int g = a.xshare;
    if (g > 1 && span == 8 && ny % g == 0) {
      // g XCDs share a tile: XCD x = unit * g + part works on the tiles {unit, unit + 8/g, ...} of the group and on
      // the channel blocks [part * ny/g, (part + 1) * ny/g) - its L2 holds 1/g of the weights (a layer's 4.7 MB of
      // 512 x 512 x 9 bf16 weights do not fit one 4-MB L2; half of them do)
      const int xcd = within & 7, slot = within >> 3, nyp = ny / g;
      const int unit = xcd / g, part = xcd - unit * g;
      tile = grp8 * 8 + unit + (8 / g) * (slot / nyp);
      yb = part * nyp + slot % nyp;
    } else {
      tile = grp8 * 8 + within % span;
      yb = within / span;
    }
  }
  const int tile_x = tile % tiles_x;
  const int tile_y = tile / tiles_x;
  const int x0 = tile_x * C::TW, y0 = tile_y * C::TH;
  const int n0 = yb * C::BN;

  const T* __restrict__ xin = static_cast<const T*>(a.x);
  const T* __restrict__ wgt = static_cast<const T*>(a.w);
  const bool w_blocked = (a.flags & STV_W_BLOCKED) != 0;
  const int nchunks = a.cin / C::CK;
  // ReLU-on-load floor: integer max with 0 clears negative elements, with INT_MIN it is the identity
  const uint32_t relu_floor = (a.flags & STV_RELU_IN) ? 0u : (sizeof(T) == 2 ? 0x80008000u : 0x80000000u);
  STV_STAMP(0);
  // Accumulator layout (MFMA roles: rows = output channels, columns = pixels): this lane owns
  // pixel r of its wave's row blocks and, per 32-channel block, channels 8j + 4h + e (j, e < 4).
  // Its bias values are requested first: a global round trip is ~2 us on a busy chip.
  const int wm_ = (wave / C::WN), wn_ = (wave % C::WN);
  const __amdgpu_buffer_rsrc_t rs_b = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(a.bias), 0, a.bias != nullptr ? a.cout * 4 : 0, 0x00020000);
  f32x4 bias_v[C::NT][4];           // channels past cout (and a null bias) read as zero
  auto load_bias = [&]() {
#pragma unroll
    for (int nt = 0; nt < C::NT; ++nt)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int nn = n0 + wn_ * (C::NT * 32) + nt * 32 + 8 * j + 4 * (lane >> 5);
        const u32x4 t = __builtin_amdgcn_raw_buffer_load_b128(rs_b, (uint32_t)(nn * 4), 0, 0);
#pragma unroll
        for (int e = 0; e < 4; ++e) bias_v[nt][j][e] = __uint_as_float(t[e]);
      }
  };
  if constexpr (!C::BIG) load_bias();

  f32x16 acc[C::MT][C::NT];
#pragma unroll
  for (int mt = 0; mt < C::MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < C::NT; ++nt)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[mt][nt][i] = 0.0f;

  constexpr uint32_t kOob = 0x80000000u;   // >= num_records for every tensor this kernel accepts
  const Geom geom{a.H, a.W, a.cout, x0, y0, n0, lane, wave, grp, wm, wn, r, h};
  const int khalf = (nchunks + 1) >> 1;
  const Phase<T> ph1{xin, wgt, a.cin, w_blocked, relu_floor, (xk && kh) ? khalf : 0, xk ? (kh ? nchunks - khalf : khalf) : nchunks};
  if constexpr (C::M16) {
    f32x4 a16[C::MT][2][C::NT][2];
#pragma unroll
    for (int mt = 0; mt < C::MT; ++mt)
#pragma unroll
      for (int pb = 0; pb < 2; ++pb)
#pragma unroll
        for (int nt = 0; nt < C::NT; ++nt)
#pragma unroll
          for (int cb = 0; cb < 2; ++cb) a16[mt][pb][nt][cb] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    conv_mainloop16<C, RELU>(ph1, geom, smem, a16);
    acc16_to_acc32<C>(a16, acc, false);
  } else {
    conv_mainloop<C, RELU>(ph1, geom, smem, acc);
  }
  STV_STAMP(2);

  // ---- fused second term (3x3 kernels only): the ReLU mask belongs to the first term alone, so it
  // is applied to the accumulators now - `ref` is read in their layout - and the 1x1 product of
  // (x2, w2) then lands on top, same tile, same registers.  The separate launch it replaces also
  // had to read-modify-write this output.
  bool mask_done = false;
  if constexpr (C::TAPS == 9) {
    if (a.x2 != nullptr) {
      if (a.flags & STV_MASK) {
        const int ref_bytes = a.H * a.W * a.cout * (int)sizeof(T);
        const __amdgpu_buffer_rsrc_t rs_m = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.ref), 0, ref_bytes, 0x00020000);
#pragma unroll
        for (int mt = 0; mt < C::MT; ++mt) {
          const int gy = y0 + wm * C::MT + mt, gx = x0 + r;
          const bool pok = gy < a.H && gx < a.W;
#pragma unroll
          for (int nt = 0; nt < C::NT; ++nt)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const int nn = n0 + wn * (C::NT * 32) + nt * 32 + 8 * j + 4 * h;
              const uint32_t off = (pok && nn < a.cout) ? (uint32_t)((((gy * a.W + gx) * a.cout) + nn) * (int)sizeof(T)) : kOob;
              if constexpr (sizeof(T) == 4) {
                const u32x4 m = __builtin_amdgcn_raw_buffer_load_b128(rs_m, off, 0, 0);
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[mt][nt][4 * j + e] = (__uint_as_float(m[e]) > 0.0f) ? acc[mt][nt][4 * j + e] : 0.0f;
              } else {
                const auto m = __builtin_amdgcn_raw_buffer_load_b64(rs_m, off, 0, 0);
                acc[mt][nt][4 * j + 0] = ((int)(m[0] << 16) > 0) ? acc[mt][nt][4 * j + 0] : 0.0f;
                acc[mt][nt][4 * j + 1] = ((int)(m[0] & 0xFFFF0000u) > 0) ? acc[mt][nt][4 * j + 1] : 0.0f;
                acc[mt][nt][4 * j + 2] = ((int)(m[1] << 16) > 0) ? acc[mt][nt][4 * j + 2] : 0.0f;
                acc[mt][nt][4 * j + 3] = ((int)(m[1] & 0xFFFF0000u) > 0) ? acc[mt][nt][4 * j + 3] : 0.0f;
              }
            }
        }
        mask_done = true;
      }
      using C1 = Cfg<T, C::TH, C::BN, C::WM, C::WN, 1, C::KS, C::NBUF, C::M16>;
      static_assert(C1::RING_BYTES <= C::LDS_BYTES, "the 1x1 pass reuses the 3x3 ring");
      const int nch2 = a.cin2 / C::CK, kh2 = (nch2 + 1) >> 1;       // (the 1x1 term's K is split like the first term's)
      const Phase<T> ph2{static_cast<const T*>(a.x2), static_cast<const T*>(a.w2), a.cin2, false,
                         sizeof(T) == 2 ? 0x80008000u : 0x80000000u, (xk && kh) ? kh2 : 0, xk ? (kh ? nch2 - kh2 : kh2) : nch2};
      if constexpr (C::M16) {            // the 1x1 product in blocks of its own, then added in the epilogue's layout
        f32x4 b16[C::MT][2][C::NT][2];
#pragma unroll
        for (int mt = 0; mt < C::MT; ++mt)
#pragma unroll
          for (int pb = 0; pb < 2; ++pb)
#pragma unroll
            for (int nt = 0; nt < C::NT; ++nt)
#pragma unroll
              for (int cb = 0; cb < 2; ++cb) b16[mt][pb][nt][cb] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        conv_mainloop16<C1, false>(ph2, geom, smem, b16);
        acc16_to_acc32<C>(b16, acc, true);
      } else {
        conv_mainloop<C1, false>(ph2, geom, smem, acc);
      }
    }
  }

  // ---- K split across workgroups: the two halves meet here, on raw sums (mask and 1x1 term above are linear in them) ----
  // The workgroup that finishes FIRST (ticket 0) leaves its accumulators in the tile's slab - lane-linear, device-scope
  // write-through stores - raises the tile's flag and exits; the second waits for the flag, adds the slab to its own
  // sums and runs the epilogue.  a + b == b + a in fp32, so the result does not depend on who came first; both words
  // are zero again when the kernel ends.  The spin is bounded (the first workgroup holds ticket 0 only once it is past
  // its main loop: it is resident and a few hundred nanoseconds from raising the flag).
  if constexpr (kXkTile) {
    if (xk) {
      __shared__ int s_role;
      uint32_t* tk = static_cast<uint32_t*>(a.xk_ws) + 2 * blk;
      float* slab = reinterpret_cast<float*>(static_cast<char*>(a.xk_ws) + kXkSlabOffset) + (size_t)blk * (C::BM * C::BN);
      const __amdgpu_buffer_rsrc_t rs_s = __builtin_amdgcn_make_buffer_rsrc(slab, 0, C::BM * C::BN * 4, 0x00020000);
      if (tid == 0) s_role = (int)__hip_atomic_fetch_add(tk, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __syncthreads();
      const int role = s_role;
      constexpr int kSc1 = 16;                           // device-scope cache policy (sc1): coherent across the XCDs' L2s
      if (role == 0) {
#pragma unroll
        for (int mt = 0; mt < C::MT; ++mt)
#pragma unroll
          for (int nt = 0; nt < C::NT; ++nt)
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {
              const u32x4 v = {__float_as_uint(acc[mt][nt][4 * q4]), __float_as_uint(acc[mt][nt][4 * q4 + 1]),
                               __float_as_uint(acc[mt][nt][4 * q4 + 2]), __float_as_uint(acc[mt][nt][4 * q4 + 3])};
              __builtin_amdgcn_raw_buffer_store_b128(v, rs_s, (uint32_t)(((((mt * C::NT + nt) * 4 + q4) * C::THREADS) + tid) * 16), 0, kSc1);
            }
        wait_vmcnt<0>();
        __syncthreads();
        if (tid == 0) __hip_atomic_store(tk + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
      }
      if (tid == 0) {
        int spins = 0;
        while (__hip_atomic_load(tk + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u && ++spins < (1 << 20)) __builtin_amdgcn_s_sleep(2);
      }
      __syncthreads();
#pragma unroll
      for (int mt = 0; mt < C::MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < C::NT; ++nt)
#pragma unroll
          for (int q4 = 0; q4 < 4; ++q4) {
            const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs_s, (uint32_t)(((((mt * C::NT + nt) * 4 + q4) * C::THREADS) + tid) * 16), 0, kSc1);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[mt][nt][4 * q4 + e] += __uint_as_float(v[e]);
          }
      if (tid == 0) {                                    // ready for the next launch that uses this workspace
        __hip_atomic_store(tk, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(tk + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  }

  if constexpr (C::BIG) load_bias();
  // the next conv's weights: one 128-byte line per lane (lines past the end: no traffic), consumed at the very end
  const __amdgpu_buffer_rsrc_t rs_pf = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.pf), 0, (int)a.pf_bytes, 0x00020000);
  const uint32_t pf_word = __builtin_amdgcn_raw_buffer_load_b32(rs_pf, ((uint32_t)blockIdx.x * C::THREADS + (uint32_t)tid) * 128u, 0, 0);

  // ---- epilogue, in registers ----------------------------------------------------------------
  // A lane holds, per (row block, 32-channel block), 4 groups of 4 consecutive channels of ONE
  // pixel.  fp32: each group is a 16-byte store as it is.  bf16: a group packs to 8 bytes;
  // v_permlane32_swap trades groups with the partner lane (same pixel, other h) so that lanes
  // 0-31 end up with channels 8j..8j+7 and lanes 32-63 with 8j+8..8j+15 of a group pair: two
  // 16-byte stores per 32 channels, no LDS round trip and no barrier.  Only the K-split variant
  // still meets in LDS (the second group's partial sums).
  const bool relu_out = (a.flags & STV_RELU_OUT) != 0;
  const bool do_mask = (a.flags & STV_MASK) != 0 && !mask_done && a.route_out == nullptr;
  const bool do_acc = (a.flags & STV_ACCUM) != 0;
  const int out_bytes = a.H * a.W * a.cout * (int)sizeof(T);
  const __amdgpu_buffer_rsrc_t rs_y = __builtin_amdgcn_make_buffer_rsrc(a.y, 0, a.y != nullptr ? out_bytes : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_ref = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<void*>(a.ref), 0, do_mask ? out_bytes : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_old = __builtin_amdgcn_make_buffer_rsrc(a.y, 0, do_acc ? out_bytes : 0, 0x00020000);
  const bool route = sizeof(T) == 2 && a.route_out != nullptr;
  const bool route_mask = (a.flags & STV_MASK) != 0 && route;        // with a route, MASK names the pre-pool ReLU
  const __amdgpu_buffer_rsrc_t rs_ridx = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<void*>(a.route_idx), 0, route ? a.H * a.W * a.cout : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_route = __builtin_amdgcn_make_buffer_rsrc(a.route_out, 0, route ? 4 * out_bytes : 0, 0x00020000);

  if (C::KS == 2) {               // the second K group hands its partial sums over through LDS
    float* cs = reinterpret_cast<float*>(smem);
    auto c_tile = [&](auto mode) {
#pragma unroll
      for (int mt = 0; mt < C::MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < C::NT; ++nt)
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            const int row = (wm * C::MT + mt) * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
            const int col = wn * (C::NT * 32) + nt * 32 + r;
            if (decltype(mode)::value == 0) cs[row * C::CS + col] = acc[mt][nt][i];
            else acc[mt][nt][i] += cs[row * C::CS + col];
          }
    };
    if (grp == 1) c_tile(std::integral_constant<int, 0>{});
    __syncthreads();
    if (grp == 0) c_tile(std::integral_constant<int, 1>{});
  }
  STV_STAMP(3);

  if (grp == 0) {
    // emit one map: `val(mt, nt, i)` yields the raw sum, `pix_ok` / `pix_off` place pixel (mt, r)
    auto emit = [&](auto&& val, auto&& pix_off, const __amdgpu_buffer_rsrc_t& rs_out, int MTN, bool masked, bool accum) {
#pragma unroll
      for (int mt = 0; mt < C::MT; ++mt) {
        if (mt >= MTN) continue;
        const uint32_t poff = pix_off(mt);                  // byte offset of this lane's pixel, or kOob
#pragma unroll
        for (int nt = 0; nt < C::NT; ++nt) {
          const int nb = n0 + wn * (C::NT * 32) + nt * 32;   // first channel of the block
          if constexpr (sizeof(T) == 4) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const int nn = nb + 8 * j + 4 * h;
              const uint32_t off = (poff != kOob && nn < a.cout) ? poff + (uint32_t)(nn * 4) : kOob;
              float v[4];
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                v[e] = val(mt, nt, 4 * j + e) + bias_v[nt][j][e];
                if (relu_out) v[e] = fmaxf(v[e], 0.0f);
              }
              if (masked) {
                const u32x4 m = __builtin_amdgcn_raw_buffer_load_b128(rs_ref, off, 0, 0);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = (__uint_as_float(m[e]) > 0.0f) ? v[e] : 0.0f;
              }
              if (accum) {
                const u32x4 o = __builtin_amdgcn_raw_buffer_load_b128(rs_old, off, 0, 0);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] += __uint_as_float(o[e]);
              }
              const u32x4 out = {__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3])};
              __builtin_amdgcn_raw_buffer_store_b128(out, rs_out, off, 0, 0);
            }
          } else {
            uint32_t px[4], py[4];                            // group j packed: (x = channels 0,1; y = 2,3)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              float v[4];
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                v[e] = val(mt, nt, 4 * j + e) + bias_v[nt][j][e];
                if (relu_out) v[e] = fmaxf(v[e], 0.0f);
              }
              if (accum) {        // exact: fp32 add before the one rounding, in the pre-swap layout
                const int nn = nb + 8 * j + 4 * h;
                const uint32_t off = (poff != kOob && nn < a.cout) ? poff + (uint32_t)(nn * 2) : kOob;
                if (masked) {     // the mask applies to the new term only: it has to come first
                  const auto m = __builtin_amdgcn_raw_buffer_load_b64(rs_ref, off, 0, 0);
                  v[0] = ((int)(m[0] << 16) > 0) ? v[0] : 0.0f;
                  v[1] = ((int)(m[0] & 0xFFFF0000u) > 0) ? v[1] : 0.0f;
                  v[2] = ((int)(m[1] << 16) > 0) ? v[2] : 0.0f;
                  v[3] = ((int)(m[1] & 0xFFFF0000u) > 0) ? v[3] : 0.0f;
                }
                const auto o = __builtin_amdgcn_raw_buffer_load_b64(rs_old, off, 0, 0);
                v[0] += __uint_as_float(o[0] << 16);
                v[1] += __uint_as_float(o[0] & 0xFFFF0000u);
                v[2] += __uint_as_float(o[1] << 16);
                v[3] += __uint_as_float(o[1] & 0xFFFF0000u);
              }
              px[j] = pack_bf16x2(v[0], v[1]);
              py[j] = pack_bf16x2(v[2], v[3]);
            }
#pragma unroll
            for (int jp = 0; jp < 4; jp += 2) {
              // lanes 32-63 of the group-jp register <-> lanes 0-31 of the group-(jp+1) register
              const auto sx = __builtin_amdgcn_permlane32_swap(px[jp], px[jp + 1], false, false);
              const auto sy = __builtin_amdgcn_permlane32_swap(py[jp], py[jp + 1], false, false);
              u32x4 out = {sx[0], sy[0], sx[1], sy[1]};
              const int nn = nb + 8 * jp + 8 * h;             // lanes 0-31: 8jp..8jp+7, lanes 32-63: the next eight
              const uint32_t off = (poff != kOob && nn < a.cout) ? poff + (uint32_t)(nn * 2) : kOob;
              if (masked && !accum) {   // (ref > 0) on packed bf16: positive <=> signed 16-bit value > 0
                typedef __attribute__((ext_vector_type(8))) short s16x8;
                const s16x8 m = __builtin_bit_cast(s16x8, __builtin_amdgcn_raw_buffer_load_b128(rs_ref, off, 0, 0));
                const s16x8 keep = (s16x8)(m > (s16x8)(short)0);          // 0xFFFF where ref > 0
                out = __builtin_bit_cast(u32x4, __builtin_bit_cast(s16x8, out) & keep);
              }
              if (route) {
                // MaxPool2d backward in place of the store: this lane's 8 channels of pooled pixel (gy, gx)
                // go to the window position their arg-max byte names (bits 0-1; bit 2 = the winner was
                // positive, i.e. the ReLU mask of the pre-pool map), zeros to the other three positions.
                const uint32_t ioff = off != kOob ? (off >> 1) : kOob;       // byte map: same element index
                const auto ib = __builtin_amdgcn_raw_buffer_load_b64(rs_ridx, ioff, 0, 0);
                const int gy = y0 + wm * C::MT + mt, gx = x0 + r;            // (only the full-resolution map is ever routed)
                const uint32_t base = (uint32_t)((((2 * gy) * (2 * a.W) + 2 * gx) * a.cout + nn) * 2);
#pragma unroll
                for (int pos = 0; pos < 4; ++pos) {
                  uint32_t keep[4];
#pragma unroll
                  for (int half = 0; half < 2; ++half) {
                    // bytes equal to the wanted code -> 0xFF.  x < 0x80 per byte, so adding 0x7F sets bit 7
                    // exactly in the non-zero bytes, without a carry into the neighbour (the subtract-and-
                    // mask zero-byte test lets a borrow ripple into a byte of value 1)
                    const uint32_t want = route_mask ? 0x01010101u * (uint32_t)(pos | 4) : 0x01010101u * (uint32_t)pos;
                    const uint32_t x = (route_mask ? ib[half] : (ib[half] & 0x03030303u)) ^ want;
                    const uint32_t hit = ((~(x + 0x7F7F7F7Fu) & 0x80808080u) >> 7) * 0xFFu;
                    keep[2 * half] = __builtin_amdgcn_perm(hit, hit, 0x01010000u);       // channels 0,1 of this half
                    keep[2 * half + 1] = __builtin_amdgcn_perm(hit, hit, 0x03030202u);   // channels 2,3
                  }
                  const u32x4 v = {out[0] & keep[0], out[1] & keep[1], out[2] & keep[2], out[3] & keep[3]};
                  const uint32_t o2 = off != kOob ? base + (uint32_t)((((pos >> 1) * 2 * a.W + (pos & 1)) * a.cout) * 2) : kOob;
                  __builtin_amdgcn_raw_buffer_store_b128(v, rs_route, o2, 0, STV_STORE_AUX);
                }
              } else {
                __builtin_amdgcn_raw_buffer_store_b128(out, rs_out, off, 0, STV_STORE_AUX);
              }
            }
          }
        }
      }
    };
    auto full_off = [&](int mt) -> uint32_t {
      const int gy = y0 + wm * C::MT + mt, gx = x0 + r;
      if (STV_DIAG & 2) return kOob;                   // (timing knock-out: no output stores)
      return (gy < a.H && gx < a.W) ? (uint32_t)(((gy * a.W + gx) * a.cout) * (int)sizeof(T)) : kOob;
    };
    // (pooling launches whose full-resolution map nobody reads, stv.h - honoured only where this tile really writes the
    //  pooled map instead: a pooled output is given and a wave owns both rows of a window)
    if (!((a.flags & STV_POOL_ONLY) && a.pool != nullptr && C::MT % 2 == 0))
      emit([&](int mt, int nt, int i) { return acc[mt][nt][i]; }, full_off, rs_y, C::MT, do_mask, do_acc);

    // Fused MaxPool2d(2,2) (forward convs in front of a pool): the vertical pair of a window is
    // two accumulator sets of this wave (MT is even, tile origin even), the horizontal pair the
    // neighbouring lane; max, bias and ReLU commute, so the pooled map is the same arithmetic on
    // max'ed sums - no second pass over HBM.  Even lanes store pooled pixel r / 2.
    if constexpr (C::MT % 2 == 0) if (a.pool != nullptr) {     // (a wave must own both rows of a pooling window)
      const int Hp = a.H >> 1, Wp = a.W >> 1;
      const __amdgpu_buffer_rsrc_t rs_p = __builtin_amdgcn_make_buffer_rsrc(a.pool, 0, Hp * Wp * a.cout * (int)sizeof(T), 0x00020000);
      // the pooled sums are formed where they are consumed (a [MT/2][NT] array of them would be 64 more registers on
      // the 16x128 tile, on top of 128 accumulators: it spilled)
      auto pooled = [&](int mp, int nt, int i) -> float {
        const float v = fmaxf(acc[2 * mp][nt][i], acc[2 * mp + 1][nt][i]);
        return fmaxf(v, __shfl_xor(v, 1, 64));
      };
      auto pool_off = [&](int mp) -> uint32_t {
        const int gyp = ((y0 + wm * C::MT) >> 1) + mp, gxp = (x0 + r) >> 1;
        return ((r & 1) == 0 && gyp < Hp && gxp < Wp) ? (uint32_t)(((gyp * Wp + gxp) * a.cout) * (int)sizeof(T)) : kOob;
      };
      emit([&](int mp, int nt, int i) { return pooled(mp < C::MT / 2 ? mp : 0, nt, i); }, pool_off, rs_p, C::MT / 2, false, false);

      // Arg-max map for the pooling backward (stv_maxpool_bwd with STV_POOL_IDX): per pooled element
      // one byte, bits 0-1 = window position of the FIRST maximum in scan order (row-major, torch),
      // bit 2 = that maximum is positive.  Decided on the values as stored (bias, ReLU, storage
      // rounding applied), so it is the decision a pooling pass over y would take.
      if (a.pool_idx != nullptr) {
        const __amdgpu_buffer_rsrc_t rs_i = __builtin_amdgcn_make_buffer_rsrc(a.pool_idx, 0, Hp * Wp * a.cout, 0x00020000);
        auto stored = [&](float v) -> float {
          if (relu_out) v = fmaxf(v, 0.0f);
          if constexpr (sizeof(T) == 2) v = bf16_to_f32(f32_to_bf16(v));
          return v;
        };
        auto right = [](float v) -> float {       // the neighbouring lane's value (quad_perm [1,0,3,2])
          return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, false));
        };
#pragma unroll
        for (int mp = 0; mp < C::MT / 2; ++mp) {
          const int gyp = ((y0 + wm * C::MT) >> 1) + mp, gxp = (x0 + r) >> 1;
          const bool pix_ok = (r & 1) == 0 && gyp < Hp && gxp < Wp;
#pragma unroll
          for (int nt = 0; nt < C::NT; ++nt) {
            const int nb = n0 + wn * (C::NT * 32) + nt * 32;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              uint32_t word = 0;
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                const float tl = stored(acc[2 * mp][nt][4 * j + e] + bias_v[nt][j][e]);
                const float bl = stored(acc[2 * mp + 1][nt][4 * j + e] + bias_v[nt][j][e]);
                const float tr = right(tl), br = right(bl);
                float best = tl;
                uint32_t code = 0;
                if (tr > best) { best = tr; code = 1; }
                if (bl > best) { best = bl; code = 2; }
                if (br > best) { best = br; code = 3; }
                if (best > 0.0f) code |= 4;
                word |= code << (8 * e);
              }
              const int nn = nb + 8 * j + 4 * h;
              const uint32_t off = (pix_ok && nn < a.cout) ? (uint32_t)((gyp * Wp + gxp) * a.cout + nn) : kOob;
              __builtin_amdgcn_raw_buffer_store_b32(word, rs_i, off, 0, 0);
            }
          }
        }
      }
    }
  }
  STV_STAMP(4);
  asm volatile("" ::"v"(pf_word));
#endif
}

template <typename C>
int launch_cfg(const ConvArgs& a_in, hipStream_t st) {
  ConvArgs a = a_in;
  a.pf = g_stv_next_w;
  a.pf_bytes = g_stv_next_w ? g_stv_next_w_bytes : 0;
  g_stv_next_w = nullptr;          // one shot: the hint belongs to THIS launch, whatever launches next on the thread starts without one
  g_stv_next_w_bytes = 0;
  {
    // XCDs a spatial tile's channel blocks are dealt to (ConvArgs::xshare).  Always 1 unless STV_CONV_XSHARE = 2 / 4
    // asks for more: measured 0.6-1.3 % slower in the step (DESIGN.md 3.8 - the weight fills it saves come out of the
    // Infinity Cache and nobody waits for them), so there is no heuristic; the switch and the block decode stay for
    // A/B runs and are covered by tests/test_gpu_ops.py::test_conv_xshare_changes_nothing.  Read per launch (launches
    // are captured into the step's graph once), so a test can flip it inside one process.
    const char* xs = getenv("STV_CONV_XSHARE");
    const int forced = xs ? atoi(xs) : 0;
    const int ny = ceil_div(a.cout, C::BN);
    int g = forced > 0 ? forced : 1;
    if (g != 1 && g != 2 && g != 4) g = 1;
    while (g > 1 && ny % g != 0) g >>= 1;
    a.xshare = g;
  }
  const bool relu = (a.flags & STV_RELU_IN) != 0;
  const void* fn = relu ? reinterpret_cast<const void*>(&conv_igemm_kernel<C, true>)
                        : reinterpret_cast<const void*>(&conv_igemm_kernel<C, false>);
  if (stv_set_max_lds(fn, C::LDS_BYTES) != STV_OK) return STV_ERR_LAUNCH;
  const int tiles = ceil_div(a.W, C::TW) * ceil_div(a.H, C::TH);
  const int blocks = tiles * ceil_div(a.cout, C::BN);
  if (a.xk == 2 && !(C::KS == 1 && !C::M16 && sizeof(typename C::Elem) == 2 && a.xk_ws != nullptr)) return STV_ERR_ARG;
  dim3 grid(a.xk == 2 ? 2 * ((blocks + 7) / 8 * 8) : blocks);      // decoded XCD-aware in the kernel
  if (relu) hipLaunchKernelGGL((conv_igemm_kernel<C, true>), grid, dim3(C::THREADS), C::LDS_BYTES, st, a);
  else hipLaunchKernelGGL((conv_igemm_kernel<C, false>), grid, dim3(C::THREADS), C::LDS_BYTES, st, a);
  STV_CHECK_LAUNCH();
  return STV_OK;
}

}  // namespace
