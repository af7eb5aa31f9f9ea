// Gram matrix / style loss on the matrix cores.
//
// Reference arithmetic (core_model.py:56-63, 264): with F = features reshaped
// to [C, N] (N = H*W pixels),  R = F F^T,  G = clamp(R, max=5e5) / (C*N),
// loss = mean((G - T)^2).  In NHWC the feature tensor is F^T = [N][C], so
// R[i][j] = sum_p F^T[p][i] F^T[p][j]: a contraction over PIXELS with both
// operands "k-major".  That is exactly the operand shape of the fp32-input
// MFMA v_mfma_f32_32x32x2_f32 (lane (r,h) supplies A[i=r][k=h] as one float),
// so bf16 features are widened to fp32 while being staged to LDS and the Gram
// is accumulated in exact fp32 ("fp32 Gram / bf16 conv", BASELINE config 3).
//
// stv_gram_partial : split-K over pixels, upper-triangular 2-D tiles, fp32
//                    partial slabs (deterministic: no atomics).
// stv_gram_finish  : reduce slabs in fixed order, mirror, clamp, scale, MSE
//                    partial sums, and the backward seed
//                    S = k * [R <= clamp] * (G - T)  (symmetric), so that
//                    dF^T = F^T * S is a plain 1x1 conv (stv_conv_igemm taps=1).
#include <type_traits>

#include <stdlib.h>

#include "stv_common.h"

namespace {

constexpr int PK = 32;  // pixels per LDS stage

template <int TS>
struct GramCfg {
  static constexpr int AT = TS / 64;       // 32x32 accumulators per wave per dim
  static constexpr int PITCH = TS + 4;     // floats per LDS row (p-major)
  static constexpr int TILE_FLOATS = PK * PITCH;
  static constexpr int STAGE_FLOATS = 2 * TILE_FLOATS;  // i-tile and j-tile
  static constexpr int LDS_BYTES = 2 * STAGE_FLOATS * 4;
};

inline int gram_tile(int C) { return C <= 64 ? 64 : 128; }

template <typename T, int TS>
__device__ __forceinline__ void gram_partial_body(const T* __restrict__ F, float* __restrict__ partials, int N, int C,
                                                  int ksplit, int chunk, int bx, int by) {
  using G = GramCfg<TS>;
  constexpr int kVec = elem_traits<T>::kVec;
  constexpr int VPR = TS / kVec;              // 16-byte vectors per tile row
  constexpr int VECS = PK * VPR;              // per operand tile
  constexpr int ITERS = (VECS + 255) / 256;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* smem = reinterpret_cast<float*>(smem_raw);

  // decode upper-triangular tile pair (ti <= tj) from blockIdx.x
  const int nt = (C + TS - 1) / TS;
  int ti = 0, rem = bx;
  while (rem >= nt - ti) { rem -= nt - ti; ++ti; }
  const int tj = ti + rem;
  const int i0 = ti * TS, j0 = tj * TS;
  const int ks = by;
  const int p_begin = ks * chunk;
  const int p_end = min(N, p_begin + chunk);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wi = wave >> 1, wj = wave & 1;
  const int r = lane & 31, h = lane >> 5;

  f32x16 acc[G::AT][G::AT];
#pragma unroll
  for (int a = 0; a < G::AT; ++a)
#pragma unroll
    for (int b = 0; b < G::AT; ++b)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.0f;

  u32x4 reg_i[ITERS], reg_j[ITERS];
  const u32x4 zero4 = {0u, 0u, 0u, 0u};
  auto stage_load = [&](int p0) {
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
      const int v = it * 256 + tid;
      const int p = p0 + v / VPR, cv = (v % VPR) * kVec;
      const bool okp = (v < VECS) && p < p_end;
      reg_i[it] = (okp && i0 + cv < C) ? *reinterpret_cast<const u32x4*>(F + (size_t)p * C + i0 + cv) : zero4;
      reg_j[it] = (okp && j0 + cv < C) ? *reinterpret_cast<const u32x4*>(F + (size_t)p * C + j0 + cv) : zero4;
    }
  };
  auto stage_write = [&](float* buf) {
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
      const int v = it * 256 + tid;
      if (v < VECS) {
        const int pr = v / VPR, cv = (v % VPR) * kVec;
        float fi[kVec], fj[kVec];
        unpack16<T>(reg_i[it], fi);
        unpack16<T>(reg_j[it], fj);
#pragma unroll
        for (int q = 0; q < kVec / 4; ++q) {
          *reinterpret_cast<f32x4*>(buf + pr * G::PITCH + cv + 4 * q) =
              f32x4{fi[4 * q], fi[4 * q + 1], fi[4 * q + 2], fi[4 * q + 3]};
          *reinterpret_cast<f32x4*>(buf + G::TILE_FLOATS + pr * G::PITCH + cv + 4 * q) =
              f32x4{fj[4 * q], fj[4 * q + 1], fj[4 * q + 2], fj[4 * q + 3]};
        }
      }
    }
  };

  const int nstages = (p_end > p_begin) ? (p_end - p_begin + PK - 1) / PK : 0;
  if (nstages > 0) {
    stage_load(p_begin);
    stage_write(smem);
  }
  __syncthreads();
  for (int s = 0; s < nstages; ++s) {
    float* cur = smem + (s & 1) * G::STAGE_FLOATS;
    float* nxt = smem + ((s + 1) & 1) * G::STAGE_FLOATS;
    const bool more = (s + 1) < nstages;
    if (more) stage_load(p_begin + (s + 1) * PK);
    const float* ai = cur + wi * (G::AT * 32) + r;
    const float* bj = cur + G::TILE_FLOATS + wj * (G::AT * 32) + r;
#pragma unroll
    for (int q = 0; q < PK / 2; ++q) {
      float av[G::AT], bv[G::AT];
#pragma unroll
      for (int a = 0; a < G::AT; ++a) av[a] = ai[(2 * q + h) * G::PITCH + a * 32];
#pragma unroll
      for (int b = 0; b < G::AT; ++b) bv[b] = bj[(2 * q + h) * G::PITCH + b * 32];
#pragma unroll
      for (int a = 0; a < G::AT; ++a)
#pragma unroll
        for (int b = 0; b < G::AT; ++b)
          acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[a], bv[b], acc[a][b], 0, 0, 0);
    }
    if (more) stage_write(nxt);
    __syncthreads();
  }

  float* out = partials + (size_t)ks * C * C;
#pragma unroll
  for (int a = 0; a < G::AT; ++a)
#pragma unroll
    for (int b = 0; b < G::AT; ++b)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int row = i0 + wi * (G::AT * 32) + a * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
        const int col = j0 + wj * (G::AT * 32) + b * 32 + r;
        if (row < C && col < C) out[(size_t)row * C + col] = acc[a][b][i];
      }
}

// One batched launch covers several taps (layers): blocks [block0[i], block0[i+1]) belong to tap i.
// The Gram chain of a step is five small, latency-bound problems; side by side in one grid they
// take the time of the slowest instead of the sum.
constexpr int kMaxTaps = 8;
struct PartialMulti {
  int n;
  int block0[kMaxTaps + 1];
  const void* F[kMaxTaps];
  float* partials[kMaxTaps];
  int N[kMaxTaps], C[kMaxTaps], ksplit[kMaxTaps], chunk[kMaxTaps], pairs[kMaxTaps];
};
__device__ __forceinline__ int find_tap(const int* block0, int n, int b) {
  int i = 0;
  while (i + 1 < n && b >= block0[i + 1]) ++i;
  return i;
}

__device__ __forceinline__ void decode_block(int b, int pairs, int ksplit, int& pair, int& ks) {
  pair = b % pairs;
  ks = b / pairs;
}

template <typename T, int TS>
__global__ __launch_bounds__(256) void gram_partial_kernel(const T* __restrict__ F, float* __restrict__ partials,
                                                           int N, int C, int ksplit, int chunk, int pairs) {
  int pair, ks;
  decode_block(blockIdx.x, pairs, ksplit, pair, ks);
  gram_partial_body<T, TS>(F, partials, N, C, ksplit, chunk, pair, ks);
}
template <typename T, int TS>
__global__ __launch_bounds__(256) void gram_partial_multi_kernel(PartialMulti m) {
  const int i = find_tap(m.block0, m.n, blockIdx.x);
  const int b = blockIdx.x - m.block0[i];
  int pair, ks;
  decode_block(b, m.pairs[i], m.ksplit[i], pair, ks);
  gram_partial_body<T, TS>(static_cast<const T*>(m.F[i]), m.partials[i], m.N[i], m.C[i], m.ksplit[i], m.chunk[i],
                           pair, ks);
}

// ---- bf16 features: v_mfma_f32_32x32x16_bf16 fed by transposing LDS reads -------------------------
// The contraction runs over pixels while memory is pixel-major, so each MFMA operand needs, per
// lane, 8 consecutive PIXELS of one channel.  ds_read_b64_tr_b16 delivers exactly that: a 16-lane
// group reads a 4-pixel x 16-channel block (lane 4q+p supplies the address of row q, channels
// 4p..4p+3) and lane i receives channel i of the 4 rows.  bf16 x bf16 products are exact in fp32
// and the MFMA accumulates in fp32, so this is still an fp32-accumulated Gram of the stored
// features; it turns the kernel from fp32-MFMA-bound into an HBM stream.
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

#ifndef STV_GRAM_PKB
#define STV_GRAM_PKB 64
#endif
constexpr int PKB = STV_GRAM_PKB;   // pixels per LDS stage (bf16 path)

template <int TS>
struct GramBCfg {
  static constexpr int AT = TS / 64;
  static constexpr int PITCH = TS * 2 + 64;            // bytes; = 64 (mod 256): tr reads conflict-free
  static constexpr int TILE_BYTES = PKB * PITCH;
  static constexpr int STAGE_BYTES = 2 * TILE_BYTES;
  static constexpr int LDS_BYTES = 2 * STAGE_BYTES;
};

// Diagnostic build (-DSTV_GRAM_STAMPS): wave 0 of block 0 accumulates 100 MHz wall-clock time per
// loop phase (MFMA phase, wait for the next stage's loads, LDS writes, barrier) and the prologue /
// epilogue; tools/gram_stamps.py reads them back.
#ifdef STV_GRAM_STAMPS
__device__ unsigned long long g_gram_stamps[8];
#define GRAM_T() wall_clock64()
#else
#define GRAM_T() 0ull
#endif

template <int TS>
__device__ __forceinline__ void gram_partial_bf16_body(const bf16_t* __restrict__ F, float* __restrict__ partials,
                                                       int N, int C, int ksplit, int chunk, int bx, int by) {
  using G = GramBCfg<TS>;
  constexpr int VPR = TS / 8;                 // 16-byte vectors per tile row
  constexpr int VECS = PKB * VPR;
  constexpr int ITERS = VECS / 256;
  static_assert(VECS % 256 == 0, "whole iterations");
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int nt = (C + TS - 1) / TS;
  int ti = 0, rem = bx;
  while (rem >= nt - ti) { rem -= nt - ti; ++ti; }
  const int tj = ti + rem;
  const bool same = ti == tj;
  const int i0 = ti * TS, j0 = tj * TS;
  const int ks = by;
  const int p_begin = ks * chunk;
  const int p_end = min(N, p_begin + chunk);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wi = wave >> 1, wj = wave & 1;
  const int grp = lane >> 4, lam = lane & 15;
  // byte offset of this lane's tr-read address inside a tile, for k-step 0 / first half
  const int tr_lane = ((grp >> 1) * 8 + (lam >> 2)) * G::PITCH + ((grp & 1) * 16 + (lam & 3) * 4) * 2;

  f32x16 acc[G::AT][G::AT];
#pragma unroll
  for (int a = 0; a < G::AT; ++a)
#pragma unroll
    for (int b = 0; b < G::AT; ++b)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.0f;

  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<bf16_t*>(F), 0, N * C * 2, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_null = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(F), 0, 0, 0x00020000);
  constexpr uint32_t kOob = 0x80000000u;
  uint32_t off_i[ITERS], off_j[ITERS];
  int lds_off[ITERS];
#pragma unroll
  for (int it = 0; it < ITERS; ++it) {
    const int v = it * 256 + tid;
    const int pr = v / VPR, cv = (v % VPR) * 8;
    lds_off[it] = pr * G::PITCH + cv * 2;
    off_i[it] = (i0 + cv < C) ? (uint32_t)((pr * C + i0 + cv) * 2) : kOob;
    off_j[it] = (j0 + cv < C) ? (uint32_t)((pr * C + j0 + cv) * 2) : kOob;
  }
  u32x4 reg_i[ITERS], reg_j[ITERS];
  auto stage_load = [&](int p0, bool live) {
    // rows past p_end must read as zero: clamp through the record count of a per-stage descriptor
    const int rows = min(PKB, p_end - p0);
    const __amdgpu_buffer_rsrc_t r =
        (live && rows > 0) ? __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(F) + (size_t)p0 * C, 0,
                                                               rows * C * 2, 0x00020000)
                           : rs_null;
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
      reg_i[it] = __builtin_amdgcn_raw_buffer_load_b128(r, off_i[it], 0, 0);
      if (!same) reg_j[it] = __builtin_amdgcn_raw_buffer_load_b128(r, off_j[it], 0, 0);
    }
  };
  auto stage_write = [&](char* buf) {
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
      *reinterpret_cast<u32x4*>(buf + lds_off[it]) = reg_i[it];
      if (!same) *reinterpret_cast<u32x4*>(buf + G::TILE_BYTES + lds_off[it]) = reg_j[it];
    }
  };
  (void)rs;

  const int nstages = (p_end > p_begin) ? (p_end - p_begin + PKB - 1) / PKB : 0;
  unsigned long long tt[6] = {0, 0, 0, 0, 0, 0};
  const unsigned long long t_begin = GRAM_T();
  stage_load(p_begin, nstages > 0);
  stage_write(smem);
  __syncthreads();
  const unsigned long long t_loop = GRAM_T();
  for (int s = 0; s < nstages; ++s) {
    const unsigned long long t0 = GRAM_T();
    char* cur = smem + (s & 1) * G::STAGE_BYTES;
    char* nxt = smem + ((s + 1) & 1) * G::STAGE_BYTES;
    stage_load(p_begin + (s + 1) * PKB, (s + 1) < nstages);
    __builtin_amdgcn_sched_barrier(0);
    const char* ta = cur + tr_lane + (wi * (G::AT * 32)) * 2;
    const char* tb = cur + (same ? 0 : G::TILE_BYTES) + tr_lane + (wj * (G::AT * 32)) * 2;
#pragma unroll
    for (int kk = 0; kk < PKB / 16; ++kk) {
      bf16x8v af[G::AT], bfr[G::AT];
#pragma unroll
      for (int a = 0; a < G::AT; ++a) {
        const char* pa = ta + kk * 16 * G::PITCH + a * 64;
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(pa));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(pa + 4 * G::PITCH));
        typedef __attribute__((ext_vector_type(8))) short s16x8;
        const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        af[a] = __builtin_bit_cast(bf16x8v, v);
      }
#pragma unroll
      for (int b = 0; b < G::AT; ++b) {
        const char* pb = tb + kk * 16 * G::PITCH + b * 64;
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(pb));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(pb + 4 * G::PITCH));
        typedef __attribute__((ext_vector_type(8))) short s16x8;
        const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        bfr[b] = __builtin_bit_cast(bf16x8v, v);
      }
#pragma unroll
      for (int a = 0; a < G::AT; ++a)
#pragma unroll
        for (int b = 0; b < G::AT; ++b)
          acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[a], bfr[b], acc[a][b], 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
#ifdef STV_GRAM_STAMPS
    const unsigned long long t1 = GRAM_T();
    __builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(0): the next stage's loads have landed
    const unsigned long long t2 = GRAM_T();
    stage_write(nxt);
    __builtin_amdgcn_s_waitcnt(0xC07F);      // lgkmcnt(0)
    const unsigned long long t3 = GRAM_T();
    __syncthreads();
    const unsigned long long t4 = GRAM_T();
    tt[0] += t1 - t0; tt[1] += t2 - t1; tt[2] += t3 - t2; tt[3] += t4 - t3;
#else
    stage_write(nxt);
    __syncthreads();
#endif
  }
  const unsigned long long t_end = GRAM_T();

  const int r = lane & 31, h = lane >> 5;
  float* out = partials + (size_t)ks * C * C;
#pragma unroll
  for (int a = 0; a < G::AT; ++a)
#pragma unroll
    for (int b = 0; b < G::AT; ++b)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int row = i0 + wi * (G::AT * 32) + a * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
        const int col = j0 + wj * (G::AT * 32) + b * 32 + r;
        if (row < C && col < C) out[(size_t)row * C + col] = acc[a][b][i];
      }
#ifdef STV_GRAM_STAMPS
  if (bx == 0 && by == 0 && tid == 0) {
    g_gram_stamps[0] = tt[0]; g_gram_stamps[1] = tt[1]; g_gram_stamps[2] = tt[2]; g_gram_stamps[3] = tt[3];
    g_gram_stamps[4] = t_loop - t_begin; g_gram_stamps[5] = GRAM_T() - t_end; g_gram_stamps[6] = nstages;
  }
#endif
}

template <int TS>
__global__ __launch_bounds__(256) void gram_partial_bf16_kernel(const bf16_t* __restrict__ F, float* __restrict__ partials,
                                                                int N, int C, int ksplit, int chunk, int pairs) {
  int pair, ks;
  decode_block(blockIdx.x, pairs, ksplit, pair, ks);
  gram_partial_bf16_body<TS>(F, partials, N, C, ksplit, chunk, pair, ks);
}
template <int TS>
__global__ __launch_bounds__(256) void gram_partial_bf16_multi_kernel(PartialMulti m) {
  const int i = find_tap(m.block0, m.n, blockIdx.x);
  const int b = blockIdx.x - m.block0[i];
  int pair, ks;
  decode_block(b, m.pairs[i], m.ksplit[i], pair, ks);
  gram_partial_bf16_body<TS>(static_cast<const bf16_t*>(m.F[i]), m.partials[i], m.N[i], m.C[i], m.ksplit[i],
                             m.chunk[i], pair, ks);
}

// Taps of BOTH tile sizes in one grid (the 128-wide tiles first: they run longer): at 512^2 the 64-channel tap of
// conv1_1 and the four wider taps were two launches of 10 + 16 us, latency-bound each.
__global__ __launch_bounds__(256) void gram_partial_bf16_both_kernel(PartialMulti m128, PartialMulti m64) {
  const int n128 = m128.block0[m128.n];
  int pair, ks;
  if ((int)blockIdx.x < n128) {
    const int i = find_tap(m128.block0, m128.n, blockIdx.x);
    decode_block(blockIdx.x - m128.block0[i], m128.pairs[i], m128.ksplit[i], pair, ks);
    gram_partial_bf16_body<128>(static_cast<const bf16_t*>(m128.F[i]), m128.partials[i], m128.N[i], m128.C[i],
                                m128.ksplit[i], m128.chunk[i], pair, ks);
  } else {
    const int b = (int)blockIdx.x - n128;
    const int i = find_tap(m64.block0, m64.n, b);
    decode_block(b - m64.block0[i], m64.pairs[i], m64.ksplit[i], pair, ks);
    gram_partial_bf16_body<64>(static_cast<const bf16_t*>(m64.F[i]), m64.partials[i], m64.N[i], m64.C[i], m64.ksplit[i],
                               m64.chunk[i], pair, ks);
  }
}

// FIN_E consecutive Gram elements (512 bytes of every slab) per block: 32 lanes x one float4 each,
// FIN_S ks-slices per element group, reduced through LDS in a fixed order (deterministic).  The
// slab walk is latency- and dispatch-bound (a thread's whole job is a handful of loads), hence
// 16-byte loads and FIN_S x FIN_U slabs in flight per block.  loss_part gets one partial per block.
// FIN_S is 32 for long slab walks (small C: hundreds of slabs) and 8 for short ones.
constexpr int FIN_V = 4, FIN_L = 32, FIN_E = FIN_L * FIN_V, FIN_U = 4;
template <typename T, int FIN_S>
__device__ __forceinline__ void gram_finish_body(
    const float* __restrict__ partials, const float* __restrict__ target, float* __restrict__ gram_out,
    float* __restrict__ loss_part, T* __restrict__ sgrad, int C, int TS, int ksplit, float clamp_max,
    float norm, float k_grad, const float* __restrict__ coef_dev, int bx) {
  __shared__ f32x4 red[FIN_S][FIN_L];
  const int le = threadIdx.x % FIN_L;
  const int slice = threadIdx.x / FIN_L;
  const int e = bx * FIN_E + le * FIN_V;       // first of this thread's 4 elements (same row: C % 4 == 0)
  const int CC = C * C;
  // Only tiles with tile(row) <= tile(col) were produced.  Elements of a lower tile have nothing
  // to read: they are finished, as mirror images, by the thread that owns the upper-tile element,
  // which reads the slabs row-wise (a transposed read of the slabs would touch one cache line per
  // element).
  const int i = e / C, j = e - i * C;
  const bool lower = (i / TS) > (j / TS);
  const bool mirror = (i / TS) < (j / TS);
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
  f32x4 s = zero;
  if (e < CC && !lower) {
    f32x4 acc[FIN_U];
#pragma unroll
    for (int u = 0; u < FIN_U; ++u) acc[u] = zero;
    const float* src = partials + e;
    int ks = slice;
    for (; ks + (FIN_U - 1) * FIN_S < ksplit; ks += FIN_U * FIN_S) {
#pragma unroll
      for (int u = 0; u < FIN_U; ++u)
        acc[u] += *reinterpret_cast<const f32x4*>(src + (size_t)(ks + u * FIN_S) * CC);
    }
#pragma unroll
    for (int u = 0; u < FIN_U; ++u)
      if (ks + u * FIN_S < ksplit) acc[u] += *reinterpret_cast<const f32x4*>(src + (size_t)(ks + u * FIN_S) * CC);
    s = (acc[0] + acc[1]) + (acc[2] + acc[3]);
  }
  red[slice][le] = s;
  __syncthreads();
  if (slice == 0) {
    float d2 = 0.0f;
    if (e < CC && !lower) {
      f32x4 R = zero;
#pragma unroll
      for (int q = 0; q < FIN_S; ++q) R += red[q][le];
      const float kk = (target && sgrad) ? k_grad * (coef_dev ? *coef_dev : 1.0f) : 0.0f;
      f32x4 Gv, tv = zero;
#pragma unroll
      for (int v = 0; v < FIN_V; ++v) Gv[v] = fminf(R[v], clamp_max) / norm;
      if (gram_out) *reinterpret_cast<f32x4*>(gram_out + e) = Gv;
      if (target) {
        tv = *reinterpret_cast<const f32x4*>(target + e);
#pragma unroll
        for (int v = 0; v < FIN_V; ++v) {
          const float d = Gv[v] - tv[v];
          d2 += d * d;
          if (sgrad) elem_traits<T>::store(sgrad + e + v, (R[v] <= clamp_max) ? kk * d : 0.0f);
        }
      }
      if (mirror) {
#pragma unroll
        for (int v = 0; v < FIN_V; ++v) {
          const int em = (j + v) * C + i;               // mirror image (strictly upper tiles only)
          if (gram_out) gram_out[em] = Gv[v];
          if (target) {
            const float dm = Gv[v] - target[em];
            d2 += dm * dm;
            if (sgrad) elem_traits<T>::store(sgrad + em, (R[v] <= clamp_max) ? kk * dm : 0.0f);
          }
        }
      }
    }
    // FIN_L = 32 active lanes of wave 0 (lanes 32..63 belong to slice 1 and stay out)
#pragma unroll
    for (int o = FIN_L / 2; o > 0; o >>= 1) d2 += __shfl_xor(d2, o, FIN_L);
    if (le == 0 && loss_part) loss_part[bx] = d2;
  }
}

template <typename T, int FIN_S>
__global__ __launch_bounds__(FIN_L * FIN_S) void gram_finish_kernel(
    const float* __restrict__ partials, const float* __restrict__ target, float* __restrict__ gram_out,
    float* __restrict__ loss_part, T* __restrict__ sgrad, int C, int TS, int ksplit, float clamp_max,
    float norm, float k_grad, const float* __restrict__ coef_dev) {
  gram_finish_body<T, FIN_S>(partials, target, gram_out, loss_part, sgrad, C, TS, ksplit, clamp_max, norm, k_grad, coef_dev,
                             blockIdx.x);
}

struct FinishMulti {
  int n;
  int block0[kMaxTaps + 1];
  const float* partials[kMaxTaps];
  const float* target[kMaxTaps];
  float* gram_out[kMaxTaps];
  float* loss_part[kMaxTaps];
  void* sgrad[kMaxTaps];
  const float* coef_dev[kMaxTaps];
  int C[kMaxTaps], TS[kMaxTaps], ksplit[kMaxTaps];
  float clamp_max[kMaxTaps], norm[kMaxTaps], k_grad[kMaxTaps];
};
template <typename T, int FIN_S>
__global__ __launch_bounds__(FIN_L * FIN_S) void gram_finish_multi_kernel(FinishMulti m) {
  const int i = find_tap(m.block0, m.n, blockIdx.x);
  gram_finish_body<T, FIN_S>(m.partials[i], m.target[i], m.gram_out[i], m.loss_part[i], static_cast<T*>(m.sgrad[i]), m.C[i],
                         m.TS[i], m.ksplit[i], m.clamp_max[i], m.norm[i], m.k_grad[i], m.coef_dev[i],
                         blockIdx.x - m.block0[i]);
}

template <int TS>
int launch_bf16(const bf16_t* F, float* partials, int N, int C, int pairs, int ksplit, hipStream_t st) {
  int chunk = ceil_div(N, ksplit);
  chunk = ceil_div(chunk, PKB) * PKB;
  if (stv_set_max_lds(reinterpret_cast<const void*>(&gram_partial_bf16_kernel<TS>), GramBCfg<TS>::LDS_BYTES) != STV_OK)
    return STV_ERR_LAUNCH;
  hipLaunchKernelGGL((gram_partial_bf16_kernel<TS>), dim3(pairs * ksplit), dim3(256), GramBCfg<TS>::LDS_BYTES, st,
                     F, partials, N, C, ksplit, chunk, pairs);
  STV_CHECK_LAUNCH();
  return STV_OK;
}

template <typename T>
int partial_typed(const void* F, float* partials, int N, int C, hipStream_t st) {
  const int TS = gram_tile(C);
  const int nt = ceil_div(C, TS);
  const int pairs = nt * (nt + 1) / 2;
  const int ksplit = stv_gram_ksplit(N, C);
  if (std::is_same<T, bf16_t>::value && (size_t)N * C * 2 < ((size_t)1 << 31)) {
    const bf16_t* Fb = static_cast<const bf16_t*>(F);
    return TS == 64 ? launch_bf16<64>(Fb, partials, N, C, pairs, ksplit, st)
                    : launch_bf16<128>(Fb, partials, N, C, pairs, ksplit, st);
  }
  int chunk = ceil_div(N, ksplit);
  chunk = ceil_div(chunk, PK) * PK;
  dim3 grid(pairs * ksplit);
  if (TS == 64) {
    hipLaunchKernelGGL((gram_partial_kernel<T, 64>), grid, dim3(256), GramCfg<64>::LDS_BYTES, st,
                       static_cast<const T*>(F), partials, N, C, ksplit, chunk, pairs);
  } else {
    if (stv_set_max_lds(reinterpret_cast<const void*>(&gram_partial_kernel<T, 128>), GramCfg<128>::LDS_BYTES) != STV_OK)
      return STV_ERR_LAUNCH;
    hipLaunchKernelGGL((gram_partial_kernel<T, 128>), grid, dim3(256), GramCfg<128>::LDS_BYTES, st,
                       static_cast<const T*>(F), partials, N, C, ksplit, chunk, pairs);
  }
  STV_CHECK_LAUNCH();
  return STV_OK;
}

}  // namespace

extern "C" int stv_gram_ksplit(int n_pixels, int channels) {
  // Split the pixel axis so that (a) the grid fills the chip, (b) the fp32 partial slabs the
  // finish kernel has to re-read stay below the size of the feature map itself.
  const int TS = gram_tile(channels);
  const int nt = ceil_div(channels, TS);
  const int pairs = nt * (nt + 1) / 2;
  static const int wg_max = getenv("STV_GRAM_WGS") ? atoi(getenv("STV_GRAM_WGS")) : 512;     // tuning aid
  static const int wg_min = getenv("STV_GRAM_WGS_MIN") ? atoi(getenv("STV_GRAM_WGS_MIN")) : 128;
  const int lo = ceil_div(wg_min, pairs), hi = (wg_max / pairs) > 0 ? wg_max / pairs : 1;
  // pixels per slab >= kdiv * channels: the fp32 slabs are written once and re-read once by the finish
  // kernel - at 2 they added up to as much traffic as the feature maps themselves (206 MB at 1024^2);
  // round-2 sweep (bench.py closure, MI355X): 8 is the knee (Gram chain 108 -> 90 us at 1024^2, 48 -> 44 at 512^2)
  static const int kdiv = getenv("STV_GRAM_KDIV") ? atoi(getenv("STV_GRAM_KDIV")) : 8;
  int ks = n_pixels / (kdiv * channels);
  if (ks < lo) ks = lo;
  if (ks > hi) ks = hi;
  const int max_ks = ceil_div(n_pixels, 64);
  if (ks > max_ks) ks = max_ks;
  if (ks < 1) ks = 1;
  return ks;
}

extern "C" size_t stv_gram_partials_bytes(int n_pixels, int channels) {
  return (size_t)stv_gram_ksplit(n_pixels, channels) * channels * channels * sizeof(float);
}

extern "C" int stv_gram_loss_parts(int channels) { return ceil_div(channels * channels, FIN_E); }

extern "C" int stv_gram_partial(const void* F, float* partials, int n_pixels, int C, int dtype,
                                void* stream) {
  if (!F || !partials || n_pixels <= 0 || C <= 0) return STV_ERR_ARG;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (dtype == STV_F32) {
    if (C % 4) return STV_ERR_ARG;
    return partial_typed<float>(F, partials, n_pixels, C, st);
  }
  if (dtype == STV_BF16) {
    if (C % 8) return STV_ERR_ARG;
    return partial_typed<bf16_t>(F, partials, n_pixels, C, st);
  }
  return STV_ERR_ARG;
}

extern "C" int stv_gram_finish(const float* partials, const float* target, float* gram_out,
                               float* loss_part, void* sgrad, int n_pixels, int C, float clamp_max,
                               float norm, float coef, const float* coef_dev, int dtype, void* stream) {
  if (!partials || n_pixels <= 0 || C <= 0 || norm <= 0.0f) return STV_ERR_ARG;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int ksplit = stv_gram_ksplit(n_pixels, C);
  const int TS = gram_tile(C);
  const int blocks = stv_gram_loss_parts(C);
  // d(mean((G-T)^2))/dR = 2/C^2 * (G-T) / norm ; dF = (dR + dR^T) F = 2 dR F
  const float k_grad = coef * 4.0f / ((float)C * (float)C * norm);
  const bool deep = ksplit >= 128;
#define STV_FINISH(T, S)                                                                                     \
  hipLaunchKernelGGL((gram_finish_kernel<T, S>), dim3(blocks), dim3(FIN_L * S), 0, st, partials, target,    \
                     gram_out, loss_part, static_cast<T*>(sgrad), C, TS, ksplit, clamp_max, norm, k_grad, coef_dev)
  if (dtype == STV_F32) {
    if (deep) STV_FINISH(float, 32);
    else STV_FINISH(float, 8);
  } else if (dtype == STV_BF16) {
    if (deep) STV_FINISH(bf16_t, 32);
    else STV_FINISH(bf16_t, 8);
  }
#undef STV_FINISH
  else
    return STV_ERR_ARG;
  STV_CHECK_LAUNCH();
  return STV_OK;
}

// ---- batched Gram chain: all taps of a step in two (partial: one per tile size) + one launches ----
extern "C" int stv_gram_multi(const stv_gram_tap_t* taps, int n_taps, int dtype, void* stream) {
  if (!taps || n_taps <= 0 || n_taps > kMaxTaps) return STV_ERR_ARG;
  if (dtype != STV_F32 && dtype != STV_BF16) return STV_ERR_ARG;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int vec = dtype == STV_F32 ? 4 : 8;
  for (int i = 0; i < n_taps; ++i) {
    const stv_gram_tap_t& t = taps[i];
    if (!t.partials || t.n_pixels <= 0 || t.channels <= 0 || t.channels % vec || t.norm <= 0.0f) return STV_ERR_ARG;
    if (dtype == STV_BF16 && (size_t)t.n_pixels * t.channels * 2 >= ((size_t)1 << 31)) return STV_ERR_ARG;
  }
  // partial sums: one launch per tile size present (bf16 with both sizes present: one launch for both)
  static const bool merge_sizes = !(getenv("STV_GRAM_MERGE") && atoi(getenv("STV_GRAM_MERGE")) == 0);   // A/B aid
  PartialMulti held{};                       // the 64-wide taps, waiting for the 128-wide ones
  for (int TS : {64, 128}) {
    PartialMulti m{};
    for (int i = 0; i < n_taps; ++i) {
      const stv_gram_tap_t& t = taps[i];
      if (gram_tile(t.channels) != TS || !t.F) continue;      // F == NULL: the producer left the slabs (stv_conv_first_fwd_gram)
      const int nt = ceil_div(t.channels, TS), pairs = nt * (nt + 1) / 2;
      const int ksplit = stv_gram_ksplit(t.n_pixels, t.channels);
      const int pk = dtype == STV_BF16 ? PKB : PK;
      int chunk = ceil_div(t.n_pixels, ksplit);
      chunk = ceil_div(chunk, pk) * pk;
      const int k = m.n++;
      m.F[k] = t.F; m.partials[k] = t.partials; m.N[k] = t.n_pixels; m.C[k] = t.channels;
      m.ksplit[k] = ksplit; m.chunk[k] = chunk; m.pairs[k] = pairs;
      m.block0[k + 1] = m.block0[k] + pairs * ksplit;
    }
    if (dtype == STV_BF16 && merge_sizes && TS == 64 && m.n) {
      bool wide = false;
      for (int i = 0; i < n_taps; ++i) wide |= gram_tile(taps[i].channels) == 128 && taps[i].F != nullptr;
      if (wide) { held = m; continue; }
    }
    if (!m.n) continue;
    if (dtype == STV_BF16 && TS == 128 && held.n) {
      constexpr int lds = GramBCfg<128>::LDS_BYTES > GramBCfg<64>::LDS_BYTES ? GramBCfg<128>::LDS_BYTES : GramBCfg<64>::LDS_BYTES;
      if (stv_set_max_lds(reinterpret_cast<const void*>(&gram_partial_bf16_both_kernel), lds) != STV_OK) return STV_ERR_LAUNCH;
      hipLaunchKernelGGL(gram_partial_bf16_both_kernel, dim3(m.block0[m.n] + held.block0[held.n]), dim3(256), lds, st, m, held);
      STV_CHECK_LAUNCH();
      continue;
    }
    const dim3 grid(m.block0[m.n]);
#define STV_SET_LDS(kern, bytes)                                                                             \
  do {                                                                                                       \
    if (stv_set_max_lds(reinterpret_cast<const void*>(&kern), bytes) != STV_OK) return STV_ERR_LAUNCH;      \
  } while (0)
    if (dtype == STV_BF16) {
      if (TS == 64) {
        STV_SET_LDS(gram_partial_bf16_multi_kernel<64>, GramBCfg<64>::LDS_BYTES);
        hipLaunchKernelGGL(gram_partial_bf16_multi_kernel<64>, grid, dim3(256), GramBCfg<64>::LDS_BYTES, st, m);
      } else {
        STV_SET_LDS(gram_partial_bf16_multi_kernel<128>, GramBCfg<128>::LDS_BYTES);
        hipLaunchKernelGGL(gram_partial_bf16_multi_kernel<128>, grid, dim3(256), GramBCfg<128>::LDS_BYTES, st, m);
      }
    } else {
      if (TS == 64) {
        STV_SET_LDS((gram_partial_multi_kernel<float, 64>), GramCfg<64>::LDS_BYTES);
        hipLaunchKernelGGL((gram_partial_multi_kernel<float, 64>), grid, dim3(256), GramCfg<64>::LDS_BYTES, st, m);
      } else {
        STV_SET_LDS((gram_partial_multi_kernel<float, 128>), GramCfg<128>::LDS_BYTES);
        hipLaunchKernelGGL((gram_partial_multi_kernel<float, 128>), grid, dim3(256), GramCfg<128>::LDS_BYTES, st, m);
      }
    }
#undef STV_SET_LDS
    STV_CHECK_LAUNCH();
  }
  // finish: one launch (8 slices per element); with STV_GRAM_FIN_MERGE=0 taps with hundreds of slabs get a launch
  // of their own that walks them with 32 slices per element (1024-thread blocks)
  for (int deep = 0; deep < 2; ++deep) {
    FinishMulti f{};
    for (int i = 0; i < n_taps; ++i) {
      const stv_gram_tap_t& t = taps[i];
      const int ksplit = stv_gram_ksplit(t.n_pixels, t.channels);
      // (one launch for all taps measured faster than a second, 1024-thread launch for the many-slab tap:
      // 14 + 8 us -> ~15 us at 512^2; STV_GRAM_FIN_MERGE=0 restores the two classes)
      static const bool one_finish = !(getenv("STV_GRAM_FIN_MERGE") && atoi(getenv("STV_GRAM_FIN_MERGE")) == 0);
      if (((ksplit >= 128 && !one_finish) ? 1 : 0) != deep) continue;
      const int k = f.n++;
      f.partials[k] = t.partials; f.target[k] = t.target; f.gram_out[k] = t.gram_out; f.loss_part[k] = t.loss_part;
      f.sgrad[k] = t.sgrad; f.coef_dev[k] = t.coef_dev; f.C[k] = t.channels; f.TS[k] = gram_tile(t.channels);
      f.ksplit[k] = ksplit;
      f.clamp_max[k] = t.clamp_max; f.norm[k] = t.norm;
      f.k_grad[k] = t.coef * 4.0f / ((float)t.channels * (float)t.channels * t.norm);
      f.block0[k + 1] = f.block0[k] + stv_gram_loss_parts(t.channels);
    }
    if (!f.n) continue;
    const dim3 grid(f.block0[f.n]);
    if (dtype == STV_F32) {
      if (deep) hipLaunchKernelGGL((gram_finish_multi_kernel<float, 32>), grid, dim3(FIN_L * 32), 0, st, f);
      else hipLaunchKernelGGL((gram_finish_multi_kernel<float, 8>), grid, dim3(FIN_L * 8), 0, st, f);
    } else {
      if (deep) hipLaunchKernelGGL((gram_finish_multi_kernel<bf16_t, 32>), grid, dim3(FIN_L * 32), 0, st, f);
      else hipLaunchKernelGGL((gram_finish_multi_kernel<bf16_t, 8>), grid, dim3(FIN_L * 8), 0, st, f);
    }
    STV_CHECK_LAUNCH();
  }
  return STV_OK;
}

#ifdef STV_GRAM_STAMPS
extern "C" int stv_debug_gram_stamps(unsigned long long* out8) {
  return hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_gram_stamps), sizeof(g_gram_stamps)) == hipSuccess ? 0 : 1;
}
#endif
