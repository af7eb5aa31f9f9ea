// First VGG conv (Cin = 3) at the image boundary: the optimised image and its
// gradient stay NCHW fp32 (the tensors the reference hands to the optimizer,
// core_model.py:66-100); activations are NHWC from here on, so the layout
// change costs nothing extra.  K = 27 is too thin for the matrix cores; these
// are VALU kernels bound by the HBM write (fwd) / read (dgrad) of the 64-channel
// activation.  A wave covers 64/LP consecutive pixels with LP lanes per pixel,
// each lane owning one 16-byte channel vector, so every activation access is a
// fully coalesced 1 KiB wave transaction.
#include "stv_common.h"

namespace {

constexpr int kMaxCin = 4;

template <typename T>
__global__ __launch_bounds__(256) void conv_first_fwd_kernel(
    const float* __restrict__ x, const float* __restrict__ wf, const float* __restrict__ bias,
    T* __restrict__ y, int H, int W, int cin, int cout, int lp_shift) {
  constexpr int kVec = elem_traits<T>::kVec;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* ws = reinterpret_cast<float*>(smem);  // [tap][cin][cout]
  for (int i = threadIdx.x; i < 9 * cin * cout; i += blockDim.x) {
    const int c = i % cin, n = (i / cin) % cout, tap = i / (cin * cout);
    ws[(tap * cin + c) * cout + n] = wf[i];
  }
  __syncthreads();
  const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t pix = gid >> lp_shift;
  const int grp = (int)(gid & ((1u << lp_shift) - 1));
  if (pix >= (size_t)H * W) return;
  const int gy = (int)(pix / W), gx = (int)(pix % W);
  const int n0 = grp * kVec;
  float acc[kVec];
#pragma unroll
  for (int e = 0; e < kVec; ++e) acc[e] = 0.0f;
  for (int tap = 0; tap < 9; ++tap) {
    const int yy = gy + tap / 3 - 1, xx = gx + tap % 3 - 1;
    if (yy < 0 || yy >= H || xx < 0 || xx >= W) continue;
    for (int c = 0; c < cin; ++c) {
      const float xv = x[((size_t)c * H + yy) * W + xx];
      const float* wp = ws + (tap * cin + c) * cout + n0;
#pragma unroll
      for (int e = 0; e < kVec; ++e) acc[e] = fmaf(xv, wp[e], acc[e]);
    }
  }
  if (bias) {
#pragma unroll
    for (int e = 0; e < kVec; ++e) acc[e] += bias[n0 + e];
  }
  *reinterpret_cast<u32x4*>(y + pix * cout + n0) = pack16<T>(acc);
}

template <typename T>
__global__ __launch_bounds__(256) void conv_first_dgrad_kernel(
    const T* __restrict__ dy, const float* __restrict__ wf, float* __restrict__ dx,
    int H, int W, int cin, int cout, int lp_shift) {
  constexpr int kVec = elem_traits<T>::kVec;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* ws = reinterpret_cast<float*>(smem);  // [tap][cin][cout]
  for (int i = threadIdx.x; i < 9 * cin * cout; i += blockDim.x) {
    const int c = i % cin, n = (i / cin) % cout, tap = i / (cin * cout);
    ws[(tap * cin + c) * cout + n] = wf[i];
  }
  __syncthreads();
  const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t pix_raw = gid >> lp_shift;
  const int lp = 1 << lp_shift;
  const int grp = (int)(gid & (lp - 1));
  const bool live = pix_raw < (size_t)H * W;
  const size_t pix = live ? pix_raw : 0;
  const int gy = (int)(pix / W), gx = (int)(pix % W);
  const int n0 = grp * kVec;
  float acc[kMaxCin] = {0.0f, 0.0f, 0.0f, 0.0f};
  // y[p] = sum_tap x[p + off(tap)] w[tap]  =>  dx[q] = sum_tap dy[q - off(tap)] w[tap]
  for (int tap = 0; tap < 9; ++tap) {
    const int yy = gy - (tap / 3 - 1), xx = gx - (tap % 3 - 1);
    if (!live || yy < 0 || yy >= H || xx < 0 || xx >= W) continue;
    float g[kVec];
    unpack16<T>(*reinterpret_cast<const u32x4*>(dy + ((size_t)yy * W + xx) * cout + n0), g);
#pragma unroll
    for (int c = 0; c < kMaxCin; ++c) {
      if (c < cin) {
        const float* wp = ws + (tap * cin + c) * cout + n0;
#pragma unroll
        for (int e = 0; e < kVec; ++e) acc[c] = fmaf(g[e], wp[e], acc[c]);
      }
    }
  }
  // reduce over the LP lanes of this pixel (lp is a power of two <= 64)
#pragma unroll
  for (int c = 0; c < kMaxCin; ++c)
    for (int o = lp >> 1; o > 0; o >>= 1) acc[c] += __shfl_xor(acc[c], o, 64);
  if (live && grp == 0) {
#pragma unroll
    for (int c = 0; c < kMaxCin; ++c)
      if (c < cin) dx[(size_t)c * H * W + pix] = acc[c];
  }
}

// slow generic variants (any cout): one thread per output element
template <typename T>
__global__ void conv_first_fwd_generic(const float* __restrict__ x, const float* __restrict__ wf,
                                       const float* __restrict__ bias, T* __restrict__ y, int H,
                                       int W, int cin, int cout) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (size_t)H * W * cout) return;
  const int n = (int)(idx % cout);
  const size_t pix = idx / cout;
  const int gy = (int)(pix / W), gx = (int)(pix % W);
  float s = 0.0f;
  for (int tap = 0; tap < 9; ++tap) {
    const int yy = gy + tap / 3 - 1, xx = gx + tap % 3 - 1;
    if (yy < 0 || yy >= H || xx < 0 || xx >= W) continue;
    for (int c = 0; c < cin; ++c)
      s = fmaf(x[((size_t)c * H + yy) * W + xx], wf[(tap * cout + n) * cin + c], s);
  }
  if (bias) s += bias[n];
  elem_traits<T>::store(y + idx, s);
}
template <typename T>
__global__ void conv_first_dgrad_generic(const T* __restrict__ dy, const float* __restrict__ wf,
                                         float* __restrict__ dx, int H, int W, int cin, int cout) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (size_t)H * W * cin) return;
  const size_t pix = idx % ((size_t)H * W);
  const int c = (int)(idx / ((size_t)H * W));
  const int gy = (int)(pix / W), gx = (int)(pix % W);
  float s = 0.0f;
  for (int tap = 0; tap < 9; ++tap) {
    const int yy = gy - (tap / 3 - 1), xx = gx - (tap % 3 - 1);
    if (yy < 0 || yy >= H || xx < 0 || xx >= W) continue;
    const T* g = dy + ((size_t)yy * W + xx) * cout;
    for (int n = 0; n < cout; ++n)
      s = fmaf(elem_traits<T>::load(g + n), wf[(tap * cout + n) * cin + c], s);
  }
  dx[idx] = s;
}

inline int lanes_shift(int cout, int kvec) {
  if (cout % kvec) return -1;
  const int lp = cout / kvec;
  if (lp < 1 || lp > 64 || (lp & (lp - 1))) return -1;
  int s = 0;
  while ((1 << s) < lp) ++s;
  return s;
}

template <typename T>
int fwd_typed(const float* x, const float* wf, const float* bias, void* y, int H, int W, int cin,
              int cout, hipStream_t st) {
  const int sh = lanes_shift(cout, elem_traits<T>::kVec);
  const size_t lds = (size_t)9 * cin * cout * sizeof(float);
  if (sh >= 0 && lds <= 64 * 1024) {
    const size_t threads = ((size_t)H * W) << sh;
    hipLaunchKernelGGL(conv_first_fwd_kernel<T>, dim3((unsigned)((threads + 255) / 256)), dim3(256),
                       lds, st, x, wf, bias, static_cast<T*>(y), H, W, cin, cout, sh);
  } else {
    const size_t total = (size_t)H * W * cout;
    hipLaunchKernelGGL(conv_first_fwd_generic<T>, dim3((unsigned)((total + 255) / 256)), dim3(256),
                       0, st, x, wf, bias, static_cast<T*>(y), H, W, cin, cout);
  }
  STV_CHECK_LAUNCH();
  return STV_OK;
}
template <typename T>
int dgrad_typed(const void* dy, const float* wf, float* dx, int H, int W, int cin, int cout,
                hipStream_t st) {
  const int sh = lanes_shift(cout, elem_traits<T>::kVec);
  const size_t lds = (size_t)9 * cin * cout * sizeof(float);
  if (sh >= 0 && lds <= 64 * 1024 && cin <= kMaxCin) {
    const size_t threads = ((size_t)H * W) << sh;
    hipLaunchKernelGGL(conv_first_dgrad_kernel<T>, dim3((unsigned)((threads + 255) / 256)),
                       dim3(256), lds, st, static_cast<const T*>(dy), wf, dx, H, W, cin, cout, sh);
  } else {
    const size_t total = (size_t)H * W * cin;
    hipLaunchKernelGGL(conv_first_dgrad_generic<T>, dim3((unsigned)((total + 255) / 256)),
                       dim3(256), 0, st, static_cast<const T*>(dy), wf, dx, H, W, cin, cout);
  }
  STV_CHECK_LAUNCH();
  return STV_OK;
}

}  // namespace

extern "C" int stv_conv_first_fwd(const float* x_nchw, const float* wf, const float* bias, void* y,
                                  int H, int W, int cin, int cout, int dtype, void* stream) {
  if (!x_nchw || !wf || !y || H <= 0 || W <= 0 || cin <= 0 || cout <= 0) return STV_ERR_ARG;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (dtype == STV_F32) return fwd_typed<float>(x_nchw, wf, bias, y, H, W, cin, cout, st);
  if (dtype == STV_BF16) return fwd_typed<bf16_t>(x_nchw, wf, bias, y, H, W, cin, cout, st);
  return STV_ERR_ARG;
}

extern "C" int stv_conv_first_dgrad(const void* dy, const float* wf, float* dx_nchw, int H, int W,
                                    int cin, int cout, int dtype, void* stream) {
  if (!dy || !wf || !dx_nchw || H <= 0 || W <= 0 || cin <= 0 || cout <= 0) return STV_ERR_ARG;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (dtype == STV_F32) return dgrad_typed<float>(dy, wf, dx_nchw, H, W, cin, cout, st);
  if (dtype == STV_BF16) return dgrad_typed<bf16_t>(dy, wf, dx_nchw, H, W, cin, cout, st);
  return STV_ERR_ARG;
}
