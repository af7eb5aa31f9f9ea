// HBM-bound pointwise / small-reduction kernels around the conv stacks:
// MaxPool2d(2,2) forward/backward, ReLU forward/backward, content MSE and its
// gradient, and the final score combine.  All activation traffic is 16-byte
// vectors per lane (1 KiB per wave instruction); grids are capped and
// grid-strided.
#include <string.h>

#include "stv_common.h"

namespace {

constexpr int kMaxBlocks = 256 * 8;

inline unsigned grid_for(size_t work_items) {
  size_t b = (work_items + 255) / 256;
  if (b < 1) b = 1;
  if (b > (size_t)kMaxBlocks) b = kMaxBlocks;
  return (unsigned)b;
}

// ---------------------------------------------------------------- max pool
// x: [H][W][C] -> y: [H/2][W/2][C] (floor), vectors of kVec channels.
template <typename T>
__global__ __launch_bounds__(256) void maxpool_fwd_kernel(const T* __restrict__ x, T* __restrict__ y,
                                                          int H, int W, int C) {
  constexpr int kVec = elem_traits<T>::kVec;
  const int Ho = H / 2, Wo = W / 2, CV = C / kVec;
  const size_t total = (size_t)Ho * Wo * CV;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int cv = (int)(i % CV);
    const size_t p = i / CV;
    const int ox = (int)(p % Wo), oy = (int)(p / Wo);
    const T* b = x + ((size_t)(2 * oy) * W + 2 * ox) * C + cv * kVec;
    float v00[kVec], v01[kVec], v10[kVec], v11[kVec], o[kVec];
    unpack16<T>(*reinterpret_cast<const u32x4*>(b), v00);
    unpack16<T>(*reinterpret_cast<const u32x4*>(b + C), v01);
    unpack16<T>(*reinterpret_cast<const u32x4*>(b + (size_t)W * C), v10);
    unpack16<T>(*reinterpret_cast<const u32x4*>(b + (size_t)W * C + C), v11);
#pragma unroll
    for (int e = 0; e < kVec; ++e) o[e] = fmaxf(fmaxf(v00[e], v01[e]), fmaxf(v10[e], v11[e]));
    *reinterpret_cast<u32x4*>(y + p * C + cv * kVec) = pack16<T>(o);
  }
}

// dx[window] = dy routed to the first maximum in scan order (torch semantics),
// optionally masked by (x > 0) for a fused ReLU backward; rows/cols dropped by
// the floor get zero.
template <typename T>
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const T* __restrict__ x,
                                                          const T* __restrict__ dy, T* __restrict__ dx,
                                                          int H, int W, int C, int flags) {
  constexpr int kVec = elem_traits<T>::kVec;
  const int Ho = H / 2, Wo = W / 2, CV = C / kVec;
  const int Hc = (H + 1) / 2, Wc = (W + 1) / 2;  // cover odd tails
  const bool mask = (flags & STV_MASK) != 0;
  const bool accum = (flags & STV_ACCUM) != 0;
  const size_t total = (size_t)Hc * Wc * CV;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int cv = (int)(i % CV);
    const size_t p = i / CV;
    const int ox = (int)(p % Wc), oy = (int)(p / Wc);
    const int iy = 2 * oy, ix = 2 * ox;
    const bool inside = oy < Ho && ox < Wo;
    float g[kVec], v[4][kVec], out[4][kVec];
#pragma unroll
    for (int e = 0; e < kVec; ++e) g[e] = 0.0f;
    if (inside) unpack16<T>(*reinterpret_cast<const u32x4*>(dy + ((size_t)oy * Wo + ox) * C + cv * kVec), g);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int yy = iy + (q >> 1), xx = ix + (q & 1);
      if (yy < H && xx < W)
        unpack16<T>(*reinterpret_cast<const u32x4*>(x + ((size_t)yy * W + xx) * C + cv * kVec), v[q]);
      else
#pragma unroll
        for (int e = 0; e < kVec; ++e) v[q][e] = 0.0f;
    }
#pragma unroll
    for (int e = 0; e < kVec; ++e) {
      int best = 0;
      float bv = v[0][e];
      // strict '>' keeps the first maximum; NaN wins like torch's (val > max || isnan(val))
#pragma unroll
      for (int q = 1; q < 4; ++q)
        if (v[q][e] > bv || (v[q][e] != v[q][e] && bv == bv)) { bv = v[q][e]; best = q; }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        float gq = (inside && q == best) ? g[e] : 0.0f;
        if (mask && !(v[q][e] > 0.0f)) gq = 0.0f;
        out[q][e] = gq;
      }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int yy = iy + (q >> 1), xx = ix + (q & 1);
      if (yy < H && xx < W) {
        T* o = dx + ((size_t)yy * W + xx) * C + cv * kVec;
        if (accum) {
          float old[kVec];
          unpack16<T>(*reinterpret_cast<const u32x4*>(o), old);
#pragma unroll
          for (int e = 0; e < kVec; ++e) out[q][e] += old[e];
        }
        *reinterpret_cast<u32x4*>(o) = pack16<T>(out[q]);
      }
    }
  }
}

// Same routing from the arg-max byte map of the fused conv+pool epilogue (one byte per pooled
// element: bits 0-1 = window position of the first maximum, bit 2 = maximum > 0): the pass reads
// dy and kVec bytes per window instead of four activation vectors.
template <typename T>
__global__ __launch_bounds__(256) void maxpool_bwd_idx_kernel(const unsigned char* __restrict__ idx,
                                                              const T* __restrict__ dy, T* __restrict__ dx,
                                                              int H, int W, int C, int flags) {
  constexpr int kVec = elem_traits<T>::kVec;
  const int Ho = H / 2, Wo = W / 2, CV = C / kVec;
  const int Hc = (H + 1) / 2, Wc = (W + 1) / 2;  // cover odd tails
  const bool mask = (flags & STV_MASK) != 0;
  const bool accum = (flags & STV_ACCUM) != 0;
  const size_t total = (size_t)Hc * Wc * CV;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int cv = (int)(i % CV);
    const size_t p = i / CV;
    const int ox = (int)(p % Wc), oy = (int)(p / Wc);
    const int iy = 2 * oy, ix = 2 * ox;
    const bool inside = oy < Ho && ox < Wo;
    float g[kVec], out[4][kVec];
    unsigned char code[kVec];
#pragma unroll
    for (int e = 0; e < kVec; ++e) { g[e] = 0.0f; code[e] = 0; }
    if (inside) {
      const size_t o = ((size_t)oy * Wo + ox) * C + cv * kVec;
      unpack16<T>(*reinterpret_cast<const u32x4*>(dy + o), g);
      if constexpr (kVec == 8) {
        const unsigned long long w = *reinterpret_cast<const unsigned long long*>(idx + o);
#pragma unroll
        for (int e = 0; e < 8; ++e) code[e] = (unsigned char)(w >> (8 * e));
      } else {
        const unsigned int w = *reinterpret_cast<const unsigned int*>(idx + o);
#pragma unroll
        for (int e = 0; e < 4; ++e) code[e] = (unsigned char)(w >> (8 * e));
      }
    }
#pragma unroll
    for (int e = 0; e < kVec; ++e) {
      const bool live = inside && (!mask || (code[e] & 4));
#pragma unroll
      for (int q = 0; q < 4; ++q) out[q][e] = (live && (code[e] & 3) == q) ? g[e] : 0.0f;
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int yy = iy + (q >> 1), xx = ix + (q & 1);
      if (yy < H && xx < W) {
        T* o = dx + ((size_t)yy * W + xx) * C + cv * kVec;
        if (accum) {
          float old[kVec];
          unpack16<T>(*reinterpret_cast<const u32x4*>(o), old);
#pragma unroll
          for (int e = 0; e < kVec; ++e) out[q][e] += old[e];
        }
        *reinterpret_cast<u32x4*>(o) = pack16<T>(out[q]);
      }
    }
  }
}

// scalar variants for channel counts that are not a multiple of the vector width
template <typename T>
__global__ void maxpool_fwd_scalar(const T* __restrict__ x, T* __restrict__ y, int H, int W, int C) {
  const int Ho = H / 2, Wo = W / 2;
  const size_t total = (size_t)Ho * Wo * C;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int c = (int)(i % C);
    const size_t p = i / C;
    const int ox = (int)(p % Wo), oy = (int)(p / Wo);
    const T* b = x + ((size_t)(2 * oy) * W + 2 * ox) * C + c;
    const float m = fmaxf(fmaxf(elem_traits<T>::load(b), elem_traits<T>::load(b + C)),
                          fmaxf(elem_traits<T>::load(b + (size_t)W * C),
                                elem_traits<T>::load(b + (size_t)W * C + C)));
    elem_traits<T>::store(y + i, m);
  }
}
template <typename T>
__global__ void maxpool_bwd_scalar(const T* __restrict__ x, const T* __restrict__ dy,
                                   T* __restrict__ dx, int H, int W, int C, int flags) {
  const int Ho = H / 2, Wo = W / 2;
  const size_t total = (size_t)H * W * C;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int c = (int)(i % C);
    const size_t p = i / C;
    const int xx = (int)(p % W), yy = (int)(p / W);
    const int oy = yy / 2, ox = xx / 2;
    float g = 0.0f;
    const float mine = elem_traits<T>::load(x + i);
    if (oy < Ho && ox < Wo) {
      int best = 0;
      float bv = 0.0f;
      for (int q = 0; q < 4; ++q) {
        const float v = elem_traits<T>::load(x + ((size_t)(2 * oy + (q >> 1)) * W + 2 * ox + (q & 1)) * C + c);
        if (q == 0 || v > bv || (v != v && bv == bv)) { bv = v; best = q; }
      }
      if (best == ((yy & 1) * 2 + (xx & 1))) g = elem_traits<T>::load(dy + ((size_t)oy * Wo + ox) * C + c);
    }
    if ((flags & STV_MASK) && !(mine > 0.0f)) g = 0.0f;
    if (flags & STV_ACCUM) g += elem_traits<T>::load(dx + i);
    elem_traits<T>::store(dx + i, g);
  }
}

// -------------------------------------------------------------------- relu
template <typename T>
__global__ __launch_bounds__(256) void relu_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, size_t n) {
  constexpr int kVec = elem_traits<T>::kVec;
  const size_t nv = n / kVec;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nv; i += (size_t)gridDim.x * 256)
    reinterpret_cast<u32x4*>(y)[i] = relu16<T>(reinterpret_cast<const u32x4*>(x)[i]);
  for (size_t i = nv * kVec + (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
    elem_traits<T>::store(y + i, fmaxf(elem_traits<T>::load(x + i), 0.0f));
}
template <typename T>
__global__ __launch_bounds__(256) void relu_bwd_kernel(const T* __restrict__ x, const T* __restrict__ dy,
                                                       T* __restrict__ dx, size_t n, int flags) {
  const bool accum = (flags & STV_ACCUM) != 0;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    float g = (elem_traits<T>::load(x + i) > 0.0f) ? elem_traits<T>::load(dy + i) : 0.0f;
    if (accum) g += elem_traits<T>::load(dx + i);
    elem_traits<T>::store(dx + i, g);
  }
}

// ----------------------------------------------------------------- content
// One partial sum per workgroup, STV_CONTENT_LOSS_PARTS workgroups (the caller's buffer has that many entries; they are
// added in index order by loss_combine).  Workgroups of 1,024 threads: 256 x 256 threads walking 25-50 MB were four waves
// per CU with one 16-byte load pair in flight each - 3.2 TB/s (15.7 us for 50 MB at 1024^2); sixteen waves per CU
// stream.  Per-thread sums are added wave by wave in fixed order: deterministic.
constexpr int kContentThreads = 1024;
__device__ __forceinline__ float block_sum_1024(float v, float* smem16) {
  v = wave_sum(v);
  const int w = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) smem16[w] = v;
  __syncthreads();
  float s = 0.0f;
#pragma unroll
  for (int i = 0; i < kContentThreads / 64; ++i) s += smem16[i];
  return s;
}
template <typename T>
__global__ __launch_bounds__(kContentThreads) void content_loss_kernel(const T* __restrict__ f, const T* __restrict__ t,
                                                                       float* __restrict__ part, size_t n) {
  constexpr int kVec = elem_traits<T>::kVec;
  constexpr int BT = kContentThreads;
  __shared__ float red[BT / 64];
  float s = 0.0f;
  const size_t nv = n / kVec;
  for (size_t i = (size_t)blockIdx.x * BT + threadIdx.x; i < nv; i += (size_t)gridDim.x * BT) {
    float a[kVec], b[kVec];
    unpack16<T>(reinterpret_cast<const u32x4*>(f)[i], a);
    unpack16<T>(reinterpret_cast<const u32x4*>(t)[i], b);
#pragma unroll
    for (int e = 0; e < kVec; ++e) {
      const float d = a[e] - b[e];
      s = fmaf(d, d, s);
    }
  }
  for (size_t i = nv * kVec + (size_t)blockIdx.x * BT + threadIdx.x; i < n; i += (size_t)gridDim.x * BT) {
    const float d = elem_traits<T>::load(f + i) - elem_traits<T>::load(t + i);
    s = fmaf(d, d, s);
  }
  s = block_sum_1024(s, red);
  if (threadIdx.x == 0) part[blockIdx.x] = s;
}
// Loss and gradient of the content term in one pass over (F, target): the same partial sums in the same
// order as content_loss_kernel (same grid, same stride), and dF = coef * 2/n * (F - target) WRITTEN (the dgrad
// that produces this layer's gradient later accumulates onto it) - one read of the two maps instead of two.
template <typename T>
__global__ __launch_bounds__(kContentThreads) void content_loss_grad_kernel(const T* __restrict__ f, const T* __restrict__ t,
                                                                            float* __restrict__ part, T* __restrict__ df, size_t n,
                                                                            float coef) {
  constexpr int kVec = elem_traits<T>::kVec;
  constexpr int BT = kContentThreads;
  __shared__ float red[BT / 64];
  const float k = coef * (2.0f / (float)n);
  float s = 0.0f;
  const size_t nv = n / kVec;
  for (size_t i = (size_t)blockIdx.x * BT + threadIdx.x; i < nv; i += (size_t)gridDim.x * BT) {
    float a[kVec], b[kVec], o[kVec];
    unpack16<T>(reinterpret_cast<const u32x4*>(f)[i], a);
    unpack16<T>(reinterpret_cast<const u32x4*>(t)[i], b);
#pragma unroll
    for (int e = 0; e < kVec; ++e) {
      const float d = a[e] - b[e];
      s = fmaf(d, d, s);
      o[e] = k * d;
    }
    reinterpret_cast<u32x4*>(df)[i] = pack16<T>(o);
  }
  for (size_t i = nv * kVec + (size_t)blockIdx.x * BT + threadIdx.x; i < n; i += (size_t)gridDim.x * BT) {
    const float d = elem_traits<T>::load(f + i) - elem_traits<T>::load(t + i);
    s = fmaf(d, d, s);
    elem_traits<T>::store(df + i, k * d);
  }
  s = block_sum_1024(s, red);
  if (threadIdx.x == 0) part[blockIdx.x] = s;
}
template <typename T>
__global__ __launch_bounds__(256) void content_grad_kernel(const T* __restrict__ f, const T* __restrict__ t,
                                                           T* __restrict__ df, size_t n, float coef,
                                                           const float* __restrict__ coef_dev, int flags) {
  constexpr int kVec = elem_traits<T>::kVec;
  const float k = coef * (coef_dev ? *coef_dev : 1.0f) * (2.0f / (float)n);
  const bool accum = (flags & STV_ACCUM) != 0;
  const size_t nv = n / kVec;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nv; i += (size_t)gridDim.x * 256) {
    float a[kVec], b[kVec], o[kVec];
    unpack16<T>(reinterpret_cast<const u32x4*>(f)[i], a);
    unpack16<T>(reinterpret_cast<const u32x4*>(t)[i], b);
    if (accum) unpack16<T>(reinterpret_cast<const u32x4*>(df)[i], o);
#pragma unroll
    for (int e = 0; e < kVec; ++e) o[e] = k * (a[e] - b[e]) + (accum ? o[e] : 0.0f);
    reinterpret_cast<u32x4*>(df)[i] = pack16<T>(o);
  }
  for (size_t i = nv * kVec + (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    float g = k * (elem_traits<T>::load(f + i) - elem_traits<T>::load(t + i));
    if (accum) g += elem_traits<T>::load(df + i);
    elem_traits<T>::store(df + i, g);
  }
}

// ------------------------------------------------------------ score combine
// One workgroup of 1024 threads.  losses[k] = scale[k] * sum(parts[off..off+cnt)) with a
// fixed summation tree (deterministic); then the reference's sequential fp32
// stack().sum() per kind and the weighted total.
__global__ __launch_bounds__(1024) void loss_combine_kernel(const float* __restrict__ parts,
                                                            const int32_t* __restrict__ table,
                                                            const float* __restrict__ scale, int n_terms,
                                                            float style_w, float content_w,
                                                            float* __restrict__ losses, float* __restrict__ scores,
                                                            float* __restrict__ log_ring, int log_cap,
                                                            uint32_t* __restrict__ log_count, uint32_t* log_seq) {
  constexpr int MAXT = 64, NW = 16;
  __shared__ int s_off[MAXT], s_cnt[MAXT], s_kind[MAXT], s_start[MAXT + 1];
  __shared__ float s_scale[MAXT], s_term[MAXT];
  __shared__ double s_part[MAXT][NW];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nt = n_terms < MAXT ? n_terms : MAXT;        // the schedule never builds more (5 style + 1 content)
  if (tid < nt) {
    s_off[tid] = table[3 * tid];
    s_cnt[tid] = table[3 * tid + 1];
    s_kind[tid] = table[3 * tid + 2];
    s_scale[tid] = scale[tid];
  }
  for (int i = tid; i < nt * NW; i += 1024) s_part[i / NW][i % NW] = 0.0;
  __syncthreads();
  if (tid == 0) {
    int acc = 0;
    for (int k = 0; k < nt; ++k) { s_start[k] = acc; acc += s_cnt[k]; }
    s_start[nt] = acc;
  }
  __syncthreads();
  // The partials of all terms form one virtual array; wave w reduces its 1/16 of it, term by
  // term, with every load of its share in flight at once (a lone wave walking an 8192-entry term
  // four loads at a time was the whole cost of this kernel).  Fixed assignment -> deterministic.
  const int total = s_start[nt];
  const int chunk = (total + NW - 1) / NW;
  const int lo = wave * chunk, hi = (lo + chunk < total) ? lo + chunk : total;
  for (int k = 0; k < nt; ++k) {
    const int a = s_start[k] > lo ? s_start[k] : lo;
    const int b = s_start[k + 1] < hi ? s_start[k + 1] : hi;
    if (a >= b) continue;                                    // wave-uniform
    const float* __restrict__ src = parts + s_off[k] - s_start[k];
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    int i = a + lane;
    for (; i + 192 < b; i += 256) {
      const float v0 = src[i], v1 = src[i + 64], v2 = src[i + 128], v3 = src[i + 192];
      s0 += (double)v0; s1 += (double)v1; s2 += (double)v2; s3 += (double)v3;
    }
    for (; i < b; i += 64) s0 += (double)src[i];
    const double tot = wave_sum_d((s0 + s1) + (s2 + s3));
    if (lane == 0) s_part[k][wave] = tot;
  }
  __syncthreads();
  if (tid < nt) {
    double tot = 0.0;
#pragma unroll
    for (int w = 0; w < NW; ++w) tot += s_part[tid][w];
    const float v = (float)(tot * (double)s_scale[tid]);
    losses[tid] = v;
    s_term[tid] = v;
  }
  __syncthreads();
  if (tid == 0) {
    float style = 0.0f, content = 0.0f;
    for (int k = 0; k < nt; ++k) {
      if (s_kind[k] == 0) style += s_term[k];
      else content += s_term[k];
    }
    const float total_loss = style_w * style + content_w * content;
    scores[0] = style;
    scores[1] = content;
    scores[2] = total_loss;
    scores[3] = (isfinite(style) && isfinite(content) && isfinite(total_loss)) ? 1.0f : 0.0f;
    if (log_ring) {          // per-evaluation history kept by the producer: slot = evaluations so far (mod capacity)
      const uint32_t k = *log_count;
      const uint32_t slot = k % (uint32_t)log_cap;
      if (log_seq) {
        // the ring lives in host memory the CPU reads while this step is still running: system-scope stores, then the
        // record count with release semantics - a reader that sees count k + 1 sees record k
        __hip_atomic_store(&log_ring[slot], style, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(&log_ring[(size_t)log_cap + slot], content, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(&log_ring[2 * (size_t)log_cap + slot], total_loss, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        *log_count = k + 1;
        __hip_atomic_store(log_seq, k + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
      } else {
        log_ring[slot] = style;
        log_ring[(size_t)log_cap + slot] = content;
        log_ring[2 * (size_t)log_cap + slot] = total_loss;
        *log_count = k + 1;
      }
    }
  }
}

template <typename T>
int pool_fwd_typed(const void* x, void* y, int H, int W, int C, hipStream_t st) {
  const size_t outs = (size_t)(H / 2) * (W / 2) * C;
  if (outs == 0) return STV_OK;
  if (C % elem_traits<T>::kVec == 0)
    hipLaunchKernelGGL(maxpool_fwd_kernel<T>, dim3(grid_for(outs / elem_traits<T>::kVec)), dim3(256), 0,
                       st, static_cast<const T*>(x), static_cast<T*>(y), H, W, C);
  else
    hipLaunchKernelGGL(maxpool_fwd_scalar<T>, dim3(grid_for(outs)), dim3(256), 0, st,
                       static_cast<const T*>(x), static_cast<T*>(y), H, W, C);
  STV_CHECK_LAUNCH();
  return STV_OK;
}
template <typename T>
int pool_bwd_typed(const void* x, const void* dy, void* dx, int H, int W, int C, int flags, hipStream_t st) {
  if (flags & STV_POOL_IDX) {
    if (C % elem_traits<T>::kVec) return STV_ERR_ARG;      // the byte map only exists for matrix-core shapes
    const size_t items = (size_t)((H + 1) / 2) * ((W + 1) / 2) * (C / elem_traits<T>::kVec);
    hipLaunchKernelGGL(maxpool_bwd_idx_kernel<T>, dim3(grid_for(items)), dim3(256), 0, st,
                       static_cast<const unsigned char*>(x), static_cast<const T*>(dy), static_cast<T*>(dx), H, W, C,
                       flags);
    STV_CHECK_LAUNCH();
    return STV_OK;
  }
  if (C % elem_traits<T>::kVec == 0) {
    const size_t items = (size_t)((H + 1) / 2) * ((W + 1) / 2) * (C / elem_traits<T>::kVec);
    hipLaunchKernelGGL(maxpool_bwd_kernel<T>, dim3(grid_for(items)), dim3(256), 0, st,
                       static_cast<const T*>(x), static_cast<const T*>(dy), static_cast<T*>(dx), H, W, C, flags);
  } else {
    hipLaunchKernelGGL(maxpool_bwd_scalar<T>, dim3(grid_for((size_t)H * W * C)), dim3(256), 0, st,
                       static_cast<const T*>(x), static_cast<const T*>(dy), static_cast<T*>(dx), H, W, C, flags);
  }
  STV_CHECK_LAUNCH();
  return STV_OK;
}


// ------------------------------------------------------------- frame export
// prepare_image_for_output + the uint8 conversion of the frame / PNG path (reference
// image_io.py:118-152, optimization.py:445-451, torchvision save_image at runtime/output.py:101):
//   v = x*std + mean (two roundings, as torch evaluates it) when `normalize`;
//   nan -> 0, +inf -> 1, -inf -> 0; clamp to [0,1];
//   round == 0: (uint8)(v*255)            (frames: numpy astype truncation)
//   round == 1: (uint8)clamp(v*255+0.5)   (save_image)
// x: NCHW fp32 [3][H][W] -> out: HWC uint8 [H][W][3].  One thread converts four pixels.
struct FrameStats { float mean[3], std[3]; };

__device__ __forceinline__ uint32_t frame_u8(float v, float mean, float stdv, int normalize, int round) {
#pragma clang fp contract(off)      // torch evaluates mul and add as two ops: an fma here changes last bits -> other bytes
  if (normalize) {
    const float scaled = v * stdv;
    v = scaled + mean;
  }
  if (v != v) v = 0.0f;
  v = fminf(fmaxf(v, 0.0f), 1.0f);                               // +-inf land on 1 / 0 like nan_to_num + clamp
  float s = v * 255.0f;
  if (round) {
    const float r = s + 0.5f;
    s = fminf(fmaxf(r, 0.0f), 255.0f);
  }
  return (uint32_t)s;
}

__global__ __launch_bounds__(256) void frame_u8_kernel(const float* __restrict__ x, uint8_t* __restrict__ out,
                                                       size_t npix, FrameStats st, int normalize, int round) {
  const size_t quads = (npix + 3) / 4;
  const bool vec_ok = (npix % 4) == 0;
  for (size_t q = (size_t)blockIdx.x * 256 + threadIdx.x; q < quads; q += (size_t)gridDim.x * 256) {
    const size_t p0 = q * 4;
    uint32_t b[12];
    if (vec_ok) {
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(x + (size_t)c * npix + p0);
#pragma unroll
        for (int k = 0; k < 4; ++k) b[3 * k + c] = frame_u8(v[k], st.mean[c], st.std[c], normalize, round);
      }
      uint32_t w[3];
#pragma unroll
      for (int j = 0; j < 3; ++j) w[j] = b[4 * j] | (b[4 * j + 1] << 8) | (b[4 * j + 2] << 16) | (b[4 * j + 3] << 24);
      uint32_t* o = reinterpret_cast<uint32_t*>(out + p0 * 3);
      o[0] = w[0]; o[1] = w[1]; o[2] = w[2];
    } else {
      for (int k = 0; k < 4 && p0 + k < npix; ++k)
        for (int c = 0; c < 3; ++c)
          out[(p0 + k) * 3 + c] = (uint8_t)frame_u8(x[(size_t)c * npix + p0 + k], st.mean[c], st.std[c], normalize, round);
    }
  }
}
}  // namespace

extern "C" int stv_maxpool_fwd(const void* x, void* y, int H, int W, int C, int dtype, void* stream) {
  if (!x || !y || H <= 0 || W <= 0 || C <= 0) return STV_ERR_ARG;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (dtype == STV_F32) return pool_fwd_typed<float>(x, y, H, W, C, st);
  if (dtype == STV_BF16) return pool_fwd_typed<bf16_t>(x, y, H, W, C, st);
  return STV_ERR_ARG;
}
extern "C" int stv_maxpool_bwd(const void* x, const void* dy, void* dx, int H, int W, int C, int flags,
                               int dtype, void* stream) {
  if (!x || !dy || !dx || H <= 0 || W <= 0 || C <= 0) return STV_ERR_ARG;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (dtype == STV_F32) return pool_bwd_typed<float>(x, dy, dx, H, W, C, flags, st);
  if (dtype == STV_BF16) return pool_bwd_typed<bf16_t>(x, dy, dx, H, W, C, flags, st);
  return STV_ERR_ARG;
}

extern "C" int stv_relu_fwd(const void* x, void* y, size_t n, int dtype, void* stream) {
  if (!x || !y) return STV_ERR_ARG;
  if (n == 0) return STV_OK;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (dtype == STV_F32)
    hipLaunchKernelGGL(relu_fwd_kernel<float>, dim3(grid_for(n / 4 + 1)), dim3(256), 0, st,
                       static_cast<const float*>(x), static_cast<float*>(y), n);
  else if (dtype == STV_BF16)
    hipLaunchKernelGGL(relu_fwd_kernel<bf16_t>, dim3(grid_for(n / 8 + 1)), dim3(256), 0, st,
                       static_cast<const bf16_t*>(x), static_cast<bf16_t*>(y), n);
  else
    return STV_ERR_ARG;
  STV_CHECK_LAUNCH();
  return STV_OK;
}
extern "C" int stv_relu_bwd(const void* x, const void* dy, void* dx, size_t n, int flags, int dtype,
                            void* stream) {
  if (!x || !dy || !dx) return STV_ERR_ARG;
  if (n == 0) return STV_OK;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (dtype == STV_F32)
    hipLaunchKernelGGL(relu_bwd_kernel<float>, dim3(grid_for(n)), dim3(256), 0, st,
                       static_cast<const float*>(x), static_cast<const float*>(dy), static_cast<float*>(dx), n, flags);
  else if (dtype == STV_BF16)
    hipLaunchKernelGGL(relu_bwd_kernel<bf16_t>, dim3(grid_for(n)), dim3(256), 0, st,
                       static_cast<const bf16_t*>(x), static_cast<const bf16_t*>(dy), static_cast<bf16_t*>(dx), n, flags);
  else
    return STV_ERR_ARG;
  STV_CHECK_LAUNCH();
  return STV_OK;
}

extern "C" int stv_content_loss(const void* F, const void* target, float* loss_part, size_t n, int dtype,
                                void* stream) {
  if (!F || !target || !loss_part || n == 0) return STV_ERR_ARG;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (dtype == STV_F32)
    hipLaunchKernelGGL(content_loss_kernel<float>, dim3(STV_CONTENT_LOSS_PARTS), dim3(kContentThreads), 0, st,
                       static_cast<const float*>(F), static_cast<const float*>(target), loss_part, n);
  else if (dtype == STV_BF16)
    hipLaunchKernelGGL(content_loss_kernel<bf16_t>, dim3(STV_CONTENT_LOSS_PARTS), dim3(kContentThreads), 0, st,
                       static_cast<const bf16_t*>(F), static_cast<const bf16_t*>(target), loss_part, n);
  else
    return STV_ERR_ARG;
  STV_CHECK_LAUNCH();
  return STV_OK;
}
extern "C" int stv_content_loss_grad(const void* F, const void* target, float* loss_part, void* dF, size_t n, float coef,
                                     int dtype, void* stream) {
  if (!F || !target || !loss_part || !dF || n == 0) return STV_ERR_ARG;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (dtype == STV_F32)
    hipLaunchKernelGGL(content_loss_grad_kernel<float>, dim3(STV_CONTENT_LOSS_PARTS), dim3(kContentThreads), 0, st,
                       static_cast<const float*>(F), static_cast<const float*>(target), loss_part, static_cast<float*>(dF), n, coef);
  else if (dtype == STV_BF16)
    hipLaunchKernelGGL(content_loss_grad_kernel<bf16_t>, dim3(STV_CONTENT_LOSS_PARTS), dim3(kContentThreads), 0, st,
                       static_cast<const bf16_t*>(F), static_cast<const bf16_t*>(target), loss_part, static_cast<bf16_t*>(dF), n, coef);
  else
    return STV_ERR_ARG;
  STV_CHECK_LAUNCH();
  return STV_OK;
}
extern "C" int stv_content_grad(const void* F, const void* target, void* dF, size_t n, float coef,
                                const float* coef_dev, int flags, int dtype, void* stream) {
  if (!F || !target || !dF || n == 0) return STV_ERR_ARG;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (dtype == STV_F32)
    hipLaunchKernelGGL(content_grad_kernel<float>, dim3(grid_for(n / 4 + 1)), dim3(256), 0, st,
                       static_cast<const float*>(F), static_cast<const float*>(target),
                       static_cast<float*>(dF), n, coef, coef_dev, flags);
  else if (dtype == STV_BF16)
    hipLaunchKernelGGL(content_grad_kernel<bf16_t>, dim3(grid_for(n / 8 + 1)), dim3(256), 0, st,
                       static_cast<const bf16_t*>(F), static_cast<const bf16_t*>(target),
                       static_cast<bf16_t*>(dF), n, coef, coef_dev, flags);
  else
    return STV_ERR_ARG;
  STV_CHECK_LAUNCH();
  return STV_OK;
}

extern "C" int stv_loss_combine_log(const float* parts, const int32_t* table, const float* scale, int n_terms,
                                    float style_w, float content_w, float* losses, float* scores, float* log_ring,
                                    int log_capacity, uint32_t* log_count, uint32_t* log_seq, void* stream) {
  if (!parts || !table || !scale || !losses || !scores || n_terms < 0) return STV_ERR_ARG;
  if (n_terms > 64) return STV_ERR_ARG;
  if ((log_ring == nullptr) != (log_count == nullptr) || (log_ring && log_capacity <= 0)) return STV_ERR_ARG;
  if (log_seq && !log_ring) return STV_ERR_ARG;
  hipLaunchKernelGGL(loss_combine_kernel, dim3(1), dim3(1024), 0, static_cast<hipStream_t>(stream), parts,
                     table, scale, n_terms, style_w, content_w, losses, scores, log_ring, log_capacity, log_count, log_seq);
  STV_CHECK_LAUNCH();
  return STV_OK;
}

// Host memory the GPU writes while the CPU reads (the loss history of a run in flight): pinned, mapped into the device's
// address space under the same pointer, fine-grained coherent.  Zeroed.
extern "C" int stv_host_mailbox_alloc(size_t bytes, void** out) {
  if (!out || bytes == 0) return STV_ERR_ARG;
  void* p = nullptr;
  if (hipHostMalloc(&p, bytes, hipHostMallocPortable | hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess || !p) {
    (void)hipGetLastError();
    return STV_ERR_ALLOC;
  }
  memset(p, 0, bytes);
  *out = p;
  return STV_OK;
}

extern "C" void stv_host_mailbox_free(void* p) {
  if (p) (void)hipHostFree(p);
}

extern "C" int stv_loss_combine(const float* parts, const int32_t* table, const float* scale, int n_terms,
                                float style_w, float content_w, float* losses, float* scores, void* stream) {
  return stv_loss_combine_log(parts, table, scale, n_terms, style_w, content_w, losses, scores, nullptr, 0, nullptr, nullptr, stream);
}

extern "C" int stv_image_to_u8(const float* x_nchw, uint8_t* out_hwc, int H, int W, const float* mean3,
                               const float* std3, int round, void* stream) {
  if (!x_nchw || !out_hwc || H <= 0 || W <= 0 || (round != 0 && round != 1)) return STV_ERR_ARG;
  if ((mean3 == nullptr) != (std3 == nullptr)) return STV_ERR_ARG;
  FrameStats st{};
  const int normalize = mean3 != nullptr;
  for (int c = 0; c < 3 && normalize; ++c) { st.mean[c] = mean3[c]; st.std[c] = std3[c]; }   // HOST arrays
  const size_t npix = (size_t)H * W;
  hipLaunchKernelGGL(frame_u8_kernel, dim3(grid_for((npix + 3) / 4)), dim3(256), 0, static_cast<hipStream_t>(stream),
                     x_nchw, out_hwc, npix, st, normalize, round);
  STV_CHECK_LAUNCH();
  return STV_OK;
}
