// Prototype only (DESIGN.md "Winograd F(2x2,3x3): A/B and go/no-go"): the two data transforms of
// Winograd F(2x2,3x3) for an NHWC bf16 3x3/pad-1 convolution, as plain HBM-bound kernels.  The 16
// element-wise products in between are channel contractions [tiles][Cin] x [Cin][Cout] and run on the
// product library's 1x1 matrix-core kernel (stv_conv_igemm, taps = 1), one launch per transform point.
// Not part of libstv_hip.so; built by tools/winograd/Makefile, driven by tools/winograd_probe.py.
//
//   V = B^T d B   (4x4 input tile d, stride 2, of every channel)      B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1]
//   M_xi = V_xi . U_xi^T   (U = G g G^T precomputed on the host)
//   Y = A^T M A   (2x2 outputs per tile)                                A^T = [1 1 1 0; 0 1 -1 -1]
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef uint16_t bf16_t;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;

__device__ __forceinline__ void unpack8(const u32x4& v, float* f) {
#pragma unroll
  for (int i = 0; i < 4; ++i) { f[2 * i] = __uint_as_float(v[i] << 16); f[2 * i + 1] = __uint_as_float(v[i] & 0xFFFF0000u); }
}
__device__ __forceinline__ uint32_t pack2(float lo, float hi) {
  __bf16 a = (__bf16)lo, b = (__bf16)hi;
  return (uint32_t)(*reinterpret_cast<bf16_t*>(&a)) | ((uint32_t)(*reinterpret_cast<bf16_t*>(&b)) << 16);
}
__device__ __forceinline__ u32x4 pack8(const float* f) {
  u32x4 v;
#pragma unroll
  for (int i = 0; i < 4; ++i) v[i] = pack2(f[2 * i], f[2 * i + 1]);
  return v;
}

// x [H][W][C] -> V [16][Th][Tw][C], Th = H/2, Tw = W/2 (H, W even), one thread per (tile, 8 channels)
__global__ __launch_bounds__(256) void wino_input_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ V, int H, int W, int C, int relu) {
  const int Th = H / 2, Tw = W / 2, CV = C / 8;
  const size_t total = (size_t)Th * Tw * CV;
  const size_t plane = (size_t)Th * Tw * C;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int cv = (int)(i % CV);
    const size_t t = i / CV;
    const int tx = (int)(t % Tw), ty = (int)(t / Tw);
    float d[4][4][8];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int gy = 2 * ty - 1 + r, gx = 2 * tx - 1 + c;
        if (gy >= 0 && gy < H && gx >= 0 && gx < W) {
          unpack8(*reinterpret_cast<const u32x4*>(x + ((size_t)gy * W + gx) * C + cv * 8), d[r][c]);
          if (relu)
#pragma unroll
            for (int e = 0; e < 8; ++e) d[r][c][e] = fmaxf(d[r][c][e], 0.0f);
        } else {
#pragma unroll
          for (int e = 0; e < 8; ++e) d[r][c][e] = 0.0f;
        }
      }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float tmp[4][4];     // B^T d
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        tmp[0][c] = d[0][c][e] - d[2][c][e];
        tmp[1][c] = d[1][c][e] + d[2][c][e];
        tmp[2][c] = d[2][c][e] - d[1][c][e];
        tmp[3][c] = d[1][c][e] - d[3][c][e];
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {   // (B^T d) B, written back into d
        d[r][0][e] = tmp[r][0] - tmp[r][2];
        d[r][1][e] = tmp[r][1] + tmp[r][2];
        d[r][2][e] = tmp[r][2] - tmp[r][1];
        d[r][3][e] = tmp[r][1] - tmp[r][3];
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int c = 0; c < 4; ++c)
        *reinterpret_cast<u32x4*>(V + (size_t)(r * 4 + c) * plane + t * C + cv * 8) = pack8(d[r][c]);
  }
}

// M [16][Th][Tw][C] (bf16 or fp32) -> y [H][W][C] bf16 (+bias, optional ReLU)
template <typename TM>
__global__ __launch_bounds__(256) void wino_output_kernel(const TM* __restrict__ M, const float* __restrict__ bias,
                                                          bf16_t* __restrict__ y, int H, int W, int C, int relu) {
  const int Th = H / 2, Tw = W / 2, CV = C / 8;
  const size_t total = (size_t)Th * Tw * CV;
  const size_t plane = (size_t)Th * Tw * C;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int cv = (int)(i % CV);
    const size_t t = i / CV;
    const int tx = (int)(t % Tw), ty = (int)(t / Tw);
    float m[4][4][8];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const TM* p = M + (size_t)(r * 4 + c) * plane + t * C + cv * 8;
        if constexpr (sizeof(TM) == 2) {
          unpack8(*reinterpret_cast<const u32x4*>(p), m[r][c]);
        } else {
#pragma unroll
          for (int e = 0; e < 8; ++e) m[r][c][e] = p[e];
        }
      }
    float o[2][2][8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float tmp[2][4];   // A^T m
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        tmp[0][c] = m[0][c][e] + m[1][c][e] + m[2][c][e];
        tmp[1][c] = m[1][c][e] - m[2][c][e] - m[3][c][e];
      }
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        const float b = bias ? bias[cv * 8 + e] : 0.0f;
        float v0 = tmp[r][0] + tmp[r][1] + tmp[r][2] + b;
        float v1 = tmp[r][1] - tmp[r][2] - tmp[r][3] + b;
        if (relu) { v0 = fmaxf(v0, 0.0f); v1 = fmaxf(v1, 0.0f); }
        o[r][0][e] = v0; o[r][1][e] = v1;
      }
    }
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
      for (int c = 0; c < 2; ++c)
        *reinterpret_cast<u32x4*>(y + ((size_t)(2 * ty + r) * W + 2 * tx + c) * C + cv * 8) = pack8(o[r][c]);
  }
}

static unsigned grid_for(size_t items) {
  size_t b = (items + 255) / 256;
  return (unsigned)(b < 1 ? 1 : (b > 256 * 16 ? 256 * 16 : b));
}

extern "C" int wino_input_transform(const void* x, void* V, int H, int W, int C, int relu, void* stream) {
  if (!x || !V || H <= 0 || W <= 0 || (H & 1) || (W & 1) || C % 8) return 1;
  hipLaunchKernelGGL(wino_input_kernel, dim3(grid_for((size_t)(H / 2) * (W / 2) * (C / 8))), dim3(256), 0,
                     static_cast<hipStream_t>(stream), static_cast<const bf16_t*>(x), static_cast<bf16_t*>(V), H, W, C, relu);
  return hipGetLastError() == hipSuccess ? 0 : 2;
}
extern "C" int wino_output_transform(const void* M, int m_is_f32, const float* bias, void* y, int H, int W, int C, int relu,
                                     void* stream) {
  if (!M || !y || H <= 0 || W <= 0 || (H & 1) || (W & 1) || C % 8) return 1;
  const unsigned g = grid_for((size_t)(H / 2) * (W / 2) * (C / 8));
  if (m_is_f32)
    hipLaunchKernelGGL(wino_output_kernel<float>, dim3(g), dim3(256), 0, static_cast<hipStream_t>(stream),
                       static_cast<const float*>(M), bias, static_cast<bf16_t*>(y), H, W, C, relu);
  else
    hipLaunchKernelGGL(wino_output_kernel<bf16_t>, dim3(g), dim3(256), 0, static_cast<hipStream_t>(stream),
                       static_cast<const bf16_t*>(M), bias, static_cast<bf16_t*>(y), H, W, C, relu);
  return hipGetLastError() == hipSuccess ? 0 : 2;
}
