// Shared device/host helpers for the MI355X (gfx950) style-transfer kernels.
// Wavefront = 64 lanes everywhere; no portability layer.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <mutex>
#include <utility>
#include <vector>

#include "../../include/stv.h"

typedef uint16_t bf16_t;  // raw bf16 storage

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8v;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;

#define STV_WAVE 64

#define STV_CHECK_LAUNCH()                                   \
  do {                                                       \
    hipError_t e__ = hipGetLastError();                      \
    if (e__ != hipSuccess) return STV_ERR_LAUNCH;            \
  } while (0)

__device__ __forceinline__ float bf16_to_f32(bf16_t v) {
  return __uint_as_float(((uint32_t)v) << 16);
}

// round-to-nearest-even; NaN stays NaN (hipcc emits v_cvt_pk_bf16_f32 for the cast)
__device__ __forceinline__ bf16_t f32_to_bf16(float f) {
  __bf16 b = (__bf16)f;
  return *reinterpret_cast<bf16_t*>(&b);
}

__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
  return (uint32_t)f32_to_bf16(lo) | ((uint32_t)f32_to_bf16(hi) << 16);
}

// ReLU on two packed bf16 values: clear each half whose sign bit is set.
__device__ __forceinline__ uint32_t relu_bf16x2(uint32_t w) {
  uint32_t s = (w >> 15) & 0x00010001u;  // 1 per negative half
  uint32_t m = s * 0xFFFFu;              // 0xFFFF per negative half
  return w & ~m;
}

template <typename T> struct elem_traits;
template <> struct elem_traits<float> {
  static constexpr int kVec = 4;  // elements per 16-byte vector
  static constexpr int kDtype = STV_F32;
  __device__ static __forceinline__ float load(const float* p) { return *p; }
  __device__ static __forceinline__ void store(float* p, float v) { *p = v; }
};
template <> struct elem_traits<bf16_t> {
  static constexpr int kVec = 8;
  static constexpr int kDtype = STV_BF16;
  __device__ static __forceinline__ float load(const bf16_t* p) { return bf16_to_f32(*p); }
  __device__ static __forceinline__ void store(bf16_t* p, float v) { *p = f32_to_bf16(v); }
};

// Unpack a 16-byte vector of T into floats (4 for f32, 8 for bf16).
template <typename T> __device__ __forceinline__ void unpack16(const u32x4& v, float* f);
template <> __device__ __forceinline__ void unpack16<float>(const u32x4& v, float* f) {
#pragma unroll
  for (int i = 0; i < 4; ++i) f[i] = __uint_as_float(v[i]);
}
template <> __device__ __forceinline__ void unpack16<bf16_t>(const u32x4& v, float* f) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    f[2 * i] = __uint_as_float(v[i] << 16);
    f[2 * i + 1] = __uint_as_float(v[i] & 0xFFFF0000u);
  }
}
template <typename T> __device__ __forceinline__ u32x4 pack16(const float* f);
template <> __device__ __forceinline__ u32x4 pack16<float>(const float* f) {
  u32x4 v;
#pragma unroll
  for (int i = 0; i < 4; ++i) v[i] = __float_as_uint(f[i]);
  return v;
}
template <> __device__ __forceinline__ u32x4 pack16<bf16_t>(const float* f) {
  u32x4 v;
#pragma unroll
  for (int i = 0; i < 4; ++i) v[i] = pack_bf16x2(f[2 * i], f[2 * i + 1]);
  return v;
}

template <typename T> __device__ __forceinline__ u32x4 relu16(u32x4 v);
template <> __device__ __forceinline__ u32x4 relu16<float>(u32x4 v) {
#pragma unroll
  for (int i = 0; i < 4; ++i) v[i] = __float_as_uint(fmaxf(__uint_as_float(v[i]), 0.0f));
  return v;
}
template <> __device__ __forceinline__ u32x4 relu16<bf16_t>(u32x4 v) {
#pragma unroll
  for (int i = 0; i < 4; ++i) v[i] = relu_bf16x2(v[i]);
  return v;
}

// Branch-free ReLU: clear every element whose sign bit is set where `enable` is all-ones.
template <typename T> __device__ __forceinline__ u32x4 relu16_masked(u32x4 v, uint32_t enable);
template <> __device__ __forceinline__ u32x4 relu16_masked<float>(u32x4 v, uint32_t enable) {
#pragma unroll
  for (int i = 0; i < 4; ++i) v[i] &= ~((uint32_t)((int32_t)v[i] >> 31) & enable);
  return v;
}
template <> __device__ __forceinline__ u32x4 relu16_masked<bf16_t>(u32x4 v, uint32_t enable) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const uint32_t s = (v[i] >> 15) & 0x00010001u;
    v[i] &= ~((s * 0xFFFFu) & enable);
  }
  return v;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
// Same total on the DPP path: four in-row butterflies (quad swaps, half-row and row mirrors) and
// the two row broadcasts, no trip through the LDS crossbar (ds_bpermute costs an LDS round trip
// per step).  The total forms in row 3 and is returned wave-uniform.  Different association than
// wave_sum, so the last bits differ; both are deterministic.
__device__ __forceinline__ float wave_sum_dpp(float v) {
#define STV_DPP_ADD(ctrl, rows)                                                                        \
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl, rows, 0xF, false))
  STV_DPP_ADD(0xB1, 0xF);    // quad_perm [1,0,3,2]
  STV_DPP_ADD(0x4E, 0xF);    // quad_perm [2,3,0,1]
  STV_DPP_ADD(0x141, 0xF);   // row_half_mirror
  STV_DPP_ADD(0x140, 0xF);   // row_mirror: every lane holds its row's sum
  STV_DPP_ADD(0x142, 0xA);   // row_bcast:15 into rows 1 and 3
  STV_DPP_ADD(0x143, 0xC);   // row_bcast:31 into rows 2 and 3
#undef STV_DPP_ADD
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
// The same DPP ladder for a double: the two halves travel as 32-bit DPP moves (a masked-out row receives
// 0.0), the addition is one DP add per step - no LDS round trips (wave_sum_d below costs 12 ds_bpermute).
__device__ __forceinline__ double wave_sum_d_dpp(double v) {
#define STV_DPP_ADD_D(ctrl, rows)                                                                         \
  do {                                                                                                  \
    const long long b_ = __double_as_longlong(v);                                                       \
    const int lo_ = __builtin_amdgcn_update_dpp(0, (int)(b_ & 0xFFFFFFFFll), ctrl, rows, 0xF, false);    \
    const int hi_ = __builtin_amdgcn_update_dpp(0, (int)(b_ >> 32), ctrl, rows, 0xF, false);             \
    v += __longlong_as_double(((long long)hi_ << 32) | (unsigned int)lo_);                              \
  } while (0)
  STV_DPP_ADD_D(0xB1, 0xF);
  STV_DPP_ADD_D(0x4E, 0xF);
  STV_DPP_ADD_D(0x141, 0xF);
  STV_DPP_ADD_D(0x140, 0xF);
  STV_DPP_ADD_D(0x142, 0xA);
  STV_DPP_ADD_D(0x143, 0xC);
#undef STV_DPP_ADD_D
  const long long b = __double_as_longlong(v);
  const int lo = __builtin_amdgcn_readlane((int)(b & 0xFFFFFFFFll), 63);
  const int hi = __builtin_amdgcn_readlane((int)(b >> 32), 63);
  return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// Block-wide sum for blockDim.x == 256 (4 waves); result valid in every thread.
__device__ __forceinline__ float block_sum_256(float v, float* smem4) {
  v = wave_sum(v);
  const int w = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) smem4[w] = v;
  __syncthreads();
  return (smem4[0] + smem4[1]) + (smem4[2] + smem4[3]);
}
__device__ __forceinline__ float block_max_256(float v, float* smem4) {
  v = wave_max(v);
  const int w = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) smem4[w] = v;
  __syncthreads();
  return fmaxf(fmaxf(smem4[0], smem4[1]), fmaxf(smem4[2], smem4[3]));
}

static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) once per (kernel, device).  Function attributes
// are per device and callers may launch from several threads / onto several devices, so a plain
// `static bool` guard is neither thread-safe nor sufficient for a second device.
inline int stv_set_max_lds(const void* fn, int bytes) {
  static std::mutex mu;
  static std::vector<std::pair<const void*, int>> done;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return STV_ERR_LAUNCH;
  std::lock_guard<std::mutex> lk(mu);
  for (const auto& e : done)
    if (e.first == fn && e.second == dev) return STV_OK;
  if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) return STV_ERR_LAUNCH;
  done.emplace_back(fn, dev);
  return STV_OK;
}
