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
__global__ __launch_bounds__(256) void gram_partial_kernel(const T* __restrict__ F,
                                                           float* __restrict__ partials, int N, int C,
                                                           int ksplit, int chunk) {
  using G = GramCfg<TS>;
  constexpr int kVec = elem_traits<T>::kVec;
  constexpr int VPR = TS / kVec;              // 16-byte vectors per tile row
  constexpr int VECS = PK * VPR;              // per operand tile
  constexpr int ITERS = (VECS + 255) / 256;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* smem = reinterpret_cast<float*>(smem_raw);

  // decode upper-triangular tile pair (ti <= tj) from blockIdx.x
  const int nt = (C + TS - 1) / TS;
  int ti = 0, rem = blockIdx.x;
  while (rem >= nt - ti) { rem -= nt - ti; ++ti; }
  const int tj = ti + rem;
  const int i0 = ti * TS, j0 = tj * TS;
  const int ks = blockIdx.y;
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

// 64 consecutive Gram elements per block; 4 ks-slices reduced through LDS in a
// fixed order.  loss_part gets one partial per block.
template <typename T>
__global__ __launch_bounds__(256) void gram_finish_kernel(
    const float* __restrict__ partials, const float* __restrict__ target, float* __restrict__ gram_out,
    float* __restrict__ loss_part, T* __restrict__ sgrad, int C, int TS, int ksplit, float clamp_max,
    float norm, float k_grad, const float* __restrict__ coef_dev) {
  __shared__ float red[4][64];
  const int e = blockIdx.x * 64 + (threadIdx.x & 63);
  const int slice = threadIdx.x >> 6;
  const int CC = C * C;
  float s = 0.0f;
  int i = 0, j = 0;
  if (e < CC) {
    i = e / C;
    j = e - i * C;
    // only tiles with tile(row) <= tile(col) were produced; mirror the rest
    const bool upper = (i / TS) <= (j / TS);
    const size_t src = upper ? ((size_t)i * C + j) : ((size_t)j * C + i);
    for (int ks = slice; ks < ksplit; ks += 4) s += partials[(size_t)ks * CC + src];
  }
  red[slice][threadIdx.x & 63] = s;
  __syncthreads();
  if (slice == 0) {
    const int t = threadIdx.x;
    float d2 = 0.0f;
    if (e < CC) {
      const float R = (red[0][t] + red[1][t]) + (red[2][t] + red[3][t]);
      const float Gv = fminf(R, clamp_max) / norm;
      if (gram_out) gram_out[e] = Gv;
      if (target) {
        const float d = Gv - target[e];
        d2 = d * d;
        if (sgrad) {
          const float kk = k_grad * (coef_dev ? *coef_dev : 1.0f);
          elem_traits<T>::store(sgrad + e, (R <= clamp_max) ? kk * d : 0.0f);
        }
      }
    }
    d2 = wave_sum(d2);
    if (t == 0 && loss_part) loss_part[blockIdx.x] = d2;
  }
}

template <typename T>
int partial_typed(const void* F, float* partials, int N, int C, hipStream_t st) {
  const int TS = gram_tile(C);
  const int nt = ceil_div(C, TS);
  const int pairs = nt * (nt + 1) / 2;
  const int ksplit = stv_gram_ksplit(N, C);
  int chunk = ceil_div(N, ksplit);
  chunk = ceil_div(chunk, PK) * PK;
  dim3 grid(pairs, ksplit);
  if (TS == 64) {
    hipLaunchKernelGGL((gram_partial_kernel<T, 64>), grid, dim3(256), GramCfg<64>::LDS_BYTES, st,
                       static_cast<const T*>(F), partials, N, C, ksplit, chunk);
  } else {
    static bool attr = false;
    if (!attr) {
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(&gram_partial_kernel<T, 128>),
                              hipFuncAttributeMaxDynamicSharedMemorySize,
                              GramCfg<128>::LDS_BYTES) != hipSuccess)
        return STV_ERR_LAUNCH;
      attr = true;
    }
    hipLaunchKernelGGL((gram_partial_kernel<T, 128>), grid, dim3(256), GramCfg<128>::LDS_BYTES, st,
                       static_cast<const T*>(F), partials, N, C, ksplit, chunk);
  }
  STV_CHECK_LAUNCH();
  return STV_OK;
}

}  // namespace

extern "C" int stv_gram_ksplit(int n_pixels, int channels) {
  const int TS = gram_tile(channels);
  const int nt = ceil_div(channels, TS);
  const int pairs = nt * (nt + 1) / 2;
  int ks = 512 / pairs;
  const int max_ks = ceil_div(n_pixels, 128);
  if (ks > max_ks) ks = max_ks;
  if (ks < 1) ks = 1;
  return ks;
}

extern "C" size_t stv_gram_partials_bytes(int n_pixels, int channels) {
  return (size_t)stv_gram_ksplit(n_pixels, channels) * channels * channels * sizeof(float);
}

extern "C" int stv_gram_loss_parts(int channels) { return ceil_div(channels * channels, 64); }

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
  if (dtype == STV_F32)
    hipLaunchKernelGGL(gram_finish_kernel<float>, dim3(blocks), dim3(256), 0, st, partials, target,
                       gram_out, loss_part, static_cast<float*>(sgrad), C, TS, ksplit, clamp_max,
                       norm, k_grad, coef_dev);
  else if (dtype == STV_BF16)
    hipLaunchKernelGGL(gram_finish_kernel<bf16_t>, dim3(blocks), dim3(256), 0, st, partials, target,
                       gram_out, loss_part, static_cast<bf16_t*>(sgrad), C, TS, ksplit, clamp_max,
                       norm, k_grad, coef_dev);
  else
    return STV_ERR_ARG;
  STV_CHECK_LAUNCH();
  return STV_OK;
}
