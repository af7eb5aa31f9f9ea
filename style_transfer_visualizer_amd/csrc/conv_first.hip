// First VGG conv (Cin = 3) at the image boundary: the optimised image and its
// gradient stay NCHW fp32 (the tensors the reference hands to the optimizer,
// core_model.py:66-100); activations are NHWC from here on, so the layout
// change costs nothing extra.  K = 27 is too thin for the matrix cores; these
// are VALU kernels bound by the HBM write (fwd) / read (dgrad) of the 64-channel
// activation.  A wave covers 64/LP consecutive pixels with LP lanes per pixel,
// each lane owning one 16-byte channel vector, so every activation access is a
// fully coalesced 1 KiB wave transaction.
#include <stdlib.h>

#include <type_traits>

#include "stv_common.h"

#ifndef STV_FIRST_DIAG   // diagnostic builds knock out phases of the matrix-core forward kernel (timing only)
#define STV_FIRST_DIAG 0
#endif

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

// ---- fast path for the VGG shape (cin = 3, cout = 64): 16x16 pixel tiles ---------------------------
// fwd : the 18x18x3 fp32 halo tile sits in LDS; one lane = one pixel x all 64 output channels.
//       The weight index is wave-uniform, so weights come through the scalar cache as SGPR
//       operands of v_fmac: exactly 27*64 FMAs per pixel and 27 conflict-free LDS reads.  Output
//       rows are transposed through LDS so each wave store is 1 KiB of contiguous NHWC bytes.
// dgrad: the 18x18x64 dy halo tile sits in LDS (pixel pitch padded to 16 B past a power of two);
//       one lane = one pixel x 3 input channels, again with scalar weights.
constexpr int FT = 16;            // tile edge
constexpr int FH = FT + 2;        // halo edge

template <typename T>
__global__ __launch_bounds__(256) void conv_first_fwd_c64(const float* __restrict__ x,
                                                          const float* __restrict__ wt,   // [9][3][64]
                                                          const float* __restrict__ bias, T* __restrict__ y,
                                                          int H, int W) {
  constexpr int CO = 64;
  constexpr int OPB = CO * (int)sizeof(T);          // output bytes per pixel
  constexpr int OPITCH = OPB + 16;
  __shared__ __attribute__((aligned(16))) float xs[3][FH][FH + 1];
  __shared__ __attribute__((aligned(16))) char os[256 * OPITCH];
  const int tid = threadIdx.x;
  const int tiles_x = (W + FT - 1) / FT;
  const int x0 = (blockIdx.x % tiles_x) * FT, y0 = (blockIdx.x / tiles_x) * FT;
  for (int i = tid; i < 3 * FH * FH; i += 256) {
    const int c = i / (FH * FH), r = (i / FH) % FH, q = i % FH;
    const int gy = y0 + r - 1, gx = x0 + q - 1;
    xs[c][r][q] = (gy >= 0 && gy < H && gx >= 0 && gx < W) ? x[((size_t)c * H + gy) * W + gx] : 0.0f;
  }
  __syncthreads();
  const int py = tid / FT, px = tid % FT;
  typedef __attribute__((ext_vector_type(2))) float f32x2;
  f32x2 acc2[CO / 2];
#pragma unroll
  for (int n = 0; n < CO / 2; ++n) acc2[n] = bias ? f32x2{bias[2 * n], bias[2 * n + 1]} : f32x2{0.0f, 0.0f};
  // 27 (tap, channel) steps, NOT unrolled: each step streams its 64 weights through SGPRs;
  // unrolling would hoist all 1728 scalar loads and spill the scalar file.  Packed FMAs
  // (v_pk_fma_f32: two output channels per instruction) halve the VALU issue count.
#pragma unroll 1
  for (int tc = 0; tc < 27; ++tc) {
    const int tap = tc / 3, c = tc - tap * 3;
    const float xv = xs[c][py + tap / 3][px + tap % 3];
    const f32x2 xv2 = {xv, xv};
    const f32x2* __restrict__ wp = reinterpret_cast<const f32x2*>(wt + tc * CO);   // wave-uniform -> scalar loads
#pragma unroll
    for (int n = 0; n < CO / 2; ++n) acc2[n] = __builtin_elementwise_fma(xv2, wp[n], acc2[n]);
  }
  float acc[CO];
#pragma unroll
  for (int n = 0; n < CO / 2; ++n) { acc[2 * n] = acc2[n][0]; acc[2 * n + 1] = acc2[n][1]; }
  // own pixel row -> LDS, then the block stores whole contiguous rows
  char* mine = os + tid * OPITCH;
  constexpr int kVec = elem_traits<T>::kVec;
#pragma unroll
  for (int v = 0; v < CO / kVec; ++v) *reinterpret_cast<u32x4*>(mine + v * 16) = pack16<T>(acc + v * kVec);
  __syncthreads();
  constexpr int VPP = OPB / 16;                       // 16-byte vectors per pixel
  for (int i = tid; i < 256 * VPP; i += 256) {
    const int p = i / VPP, v = i % VPP;
    const int gy = y0 + p / FT, gx = x0 + p % FT;
    if (gy < H && gx < W)
      *reinterpret_cast<u32x4*>(reinterpret_cast<char*>(y) + ((size_t)gy * W + gx) * OPB + v * 16) =
          *reinterpret_cast<const u32x4*>(os + p * OPITCH + v * 16);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void conv_first_dgrad_c64(const T* __restrict__ dy,
                                                            const float* __restrict__ wd,   // [9][3][64], taps as read
                                                            float* __restrict__ dx, int H, int W) {
  constexpr int CO = 64;
  constexpr int PB = CO * (int)sizeof(T);
  constexpr int PITCH = PB + 16;
  constexpr int kVec = elem_traits<T>::kVec;
  constexpr int VPP = PB / 16;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int tiles_x = (W + FT - 1) / FT;
  const int x0 = (blockIdx.x % tiles_x) * FT, y0 = (blockIdx.x / tiles_x) * FT;
  const u32x4 zero4 = {0u, 0u, 0u, 0u};
  for (int i = tid; i < FH * FH * VPP; i += 256) {
    const int p = i / VPP, v = i % VPP;
    const int gy = y0 + p / FH - 1, gx = x0 + p % FH - 1;
    const bool ok = gy >= 0 && gy < H && gx >= 0 && gx < W;
    *reinterpret_cast<u32x4*>(smem + p * PITCH + v * 16) =
        ok ? *reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(dy) + ((size_t)gy * W + gx) * PB + v * 16)
           : zero4;
  }
  __syncthreads();
  const int py = tid / FT, px = tid % FT;
  typedef __attribute__((ext_vector_type(2))) float f32x2;
  f32x2 p0 = {0.0f, 0.0f}, p1 = {0.0f, 0.0f}, p2 = {0.0f, 0.0f};   // (even, odd) channel partial sums
  // dx[q] = sum_tap dy[q - off(tap)] w[tap]; with t' = 8 - tap the read offset is +off(t'), and wd is
  // packed in that (already flipped) order
#pragma unroll 1
  for (int tap = 0; tap < 9; ++tap) {
    const char* src = smem + ((py + tap / 3) * FH + (px + tap % 3)) * PITCH;
    const float* __restrict__ wp = wd + tap * 3 * CO;          // wave-uniform -> scalar loads
    // blocked summation (this is the parity mode's kernel): one tap's 64 products per fresh accumulator,
    // added to the running sum - chains of 32 + 9 instead of 288 sequential additions per lane half
    f32x2 q0 = {0.0f, 0.0f}, q1 = {0.0f, 0.0f}, q2 = {0.0f, 0.0f};
#pragma unroll 2
    for (int v = 0; v < VPP; ++v) {
      float g[kVec];
      unpack16<T>(*reinterpret_cast<const u32x4*>(src + v * 16), g);
#pragma unroll
      for (int e = 0; e < kVec; e += 2) {      // v_pk_fma_f32: two channels per instruction
        const f32x2 g2 = {g[e], g[e + 1]};
        q0 = __builtin_elementwise_fma(g2, *reinterpret_cast<const f32x2*>(wp + 0 * CO + v * kVec + e), q0);
        q1 = __builtin_elementwise_fma(g2, *reinterpret_cast<const f32x2*>(wp + 1 * CO + v * kVec + e), q1);
        q2 = __builtin_elementwise_fma(g2, *reinterpret_cast<const f32x2*>(wp + 2 * CO + v * kVec + e), q2);
      }
    }
    p0 += q0; p1 += q1; p2 += q2;
  }
  const float a0 = p0[0] + p0[1], a1 = p1[0] + p1[1], a2 = p2[0] + p2[1];
  const int gy = y0 + py, gx = x0 + px;
  if (gy < H && gx < W) {
    const size_t plane = (size_t)H * W, o = (size_t)gy * W + gx;
    dx[o] = a0;
    dx[plane + o] = a1;
    dx[2 * plane + o] = a2;
  }
}

// ---- forward on the matrix cores (bf16 activations) ------------------------------------------------
// K = 27 padded to 32 is two v_mfma_f32_32x32x16_bf16 steps; what keeps it fp32-faithful is a
// two-term bf16 split of BOTH operands (v = hi + lo, each bf16): x.w ~ hi.hi + hi.lo + lo.hi, error
// ~2^-16 relative - far below the bf16 rounding of the stored output.  6 MFMAs per 32 px x 32
// couts instead of 27 x 32 VALU FMAs per lane: the kernel is left with the 64-channel NHWC write.
// k = tap * 3 + c (27..31 zero).  One workgroup = 8 rows x 32 px, wave w owns rows 2w, 2w+1.
constexpr int MF_TH = 8, MF_TW = 32, MF_IW = MF_TW + 2 + 2;   // LDS row pitch 36 floats
constexpr int MF_PLANE = (MF_TH + 2) * MF_IW;
constexpr int MF_FRAG_WORDS = 2 * 2 * 2 * 64 * 4;           // [hi|lo][kstep][nt][lane] x 4 dwords

__global__ void pack_first_fragments(const float* __restrict__ wf, uint32_t* __restrict__ frag) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;        // one bf16 pair per thread
  if (i >= MF_FRAG_WORDS) return;
  const int q = i & 3, lane = (i >> 2) & 63, nt = (i >> 8) & 1, ks = (i >> 9) & 1, part = i >> 10;
  const int r = lane & 31, h = lane >> 5;
  uint32_t out = 0;
  for (int e2 = 0; e2 < 2; ++e2) {
    const int k = ks * 16 + 8 * h + 2 * q + e2;
    float v = 0.0f;
    if (k < 27) v = wf[((k / 3) * 64 + nt * 32 + r) * 3 + (k % 3)];
    const bf16_t hi = f32_to_bf16(v);
    const bf16_t lo = f32_to_bf16(v - bf16_to_f32(hi));
    out |= (uint32_t)(part ? lo : hi) << (16 * e2);
  }
  frag[i] = out;
}

// GRAM: the layer is a style tap - the kernel also leaves the split-K slabs of R = Y^T Y that
// stv_gram_partial would compute from the stored map (one [64][64] fp32 slab per workgroup, slab index =
// blockIdx.x), so the 134-MB map (1024^2) is not read back for it.  The Gram product contracts over
// pixels while the accumulators hold [channel rows][pixel columns]: the packed words a lane is about to
// store also go, pixel-major, into a 6-KB per-wave LDS patch, and ds_read_b64_tr_b16 hands them back as
// [channel][8 pixels] operands (6 MFMAs per 32 pixels; wave-private, no barrier).  (First version: the
// transposed accumulators from the same products with the MFMA operands swapped - no LDS at all, but 12
// more MFMAs per 32 pixels: +20 us at 1024^2, as much as the separate Gram pass had cost.)
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
template <bool GRAM>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void conv_first_fwd_mfma(const float* __restrict__ x,
                                                           const uint32_t* __restrict__ frag,
                                                           const float* __restrict__ bias,
                                                           bf16_t* __restrict__ y, float* __restrict__ gram_slabs,
                                                           int H, int W) {
#if defined(__HIP_DEVICE_COMPILE__)
  // [c][row][col]; every pixel value is stored already split: bf16 hi in the upper, bf16 lo (the
  // remainder) in the lower half of the word - split once here, not once per tap it is read for.
  // Last word = 0.
  __shared__ __attribute__((aligned(16))) uint32_t xs[3 * MF_PLANE + 4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int tiles_x = (W + MF_TW - 1) / MF_TW;
  const int ntiles = tiles_x * ((H + MF_TH - 1) / MF_TH);
  // Persistent workgroup: the halo tile of the NEXT image tile is requested (into registers)
  // before the current one is computed, so the HBM round trip of the loads and the drain of the
  // stores overlap the split / MFMA / transpose work instead of bracketing it.
  constexpr int NPRE = (3 * MF_PLANE + 255) / 256;
  float pre[NPRE];
  // which halo-tile element (channel, row, column) each of this thread's NPRE loads fetches is the same for
  // every tile: decoded once (the divisions by the plane / row pitch were ~25 VALU per element per tile)
  int pcyx[NPRE];
#pragma unroll
  for (int k = 0; k < NPRE; ++k) {
    const int i = k * 256 + tid;
    const int c = i / MF_PLANE, rem = i - c * MF_PLANE, py = rem / MF_IW, px = rem - py * MF_IW;
    pcyx[k] = (i < 3 * MF_PLANE && px < MF_TW + 2) ? ((c << 16) | (py << 8) | px) : -1;
  }
  auto request = [&](int t) {
    const int tx0 = (t % tiles_x) * MF_TW, ty0 = (t / tiles_x) * MF_TH;
#pragma unroll
    for (int k = 0; k < NPRE; ++k) {
      const int c = pcyx[k] >> 16, py = (pcyx[k] >> 8) & 255, px = pcyx[k] & 255;
      const int gy = ty0 + py - 1, gx = tx0 + px - 1;
      const bool ok = pcyx[k] >= 0 && gy >= 0 && gy < H && gx >= 0 && gx < W;
      pre[k] = ok ? x[((size_t)c * H + gy) * W + gx] : 0.0f;
    }
  };
  request(blockIdx.x);
  if (tid < 4) xs[3 * MF_PLANE + tid] = 0u;
  // weight fragments: [hi|lo][kstep][nt] for this lane, 16 bytes each
  bf16x8v bw[2][2][2];
#pragma unroll
  for (int part = 0; part < 2; ++part)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        const u32x4 t = *reinterpret_cast<const u32x4*>(frag + ((((part * 2 + ks) * 2 + nt) * 64 + lane) << 2));
        bw[part][ks][nt] = __builtin_bit_cast(bf16x8v, t);
      }
  // LDS word offsets of this lane's 16 K entries relative to (row, r): c * plane + dy * pitch + dx
  int koff[2][8];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks)
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int k = ks * 16 + 8 * h + e;
      const int tap = k / 3, c = k - tap * 3;
      koff[ks][e] = (k < 27) ? c * MF_PLANE + (tap / 3) * MF_IW + (tap % 3) : -1;
    }
  // accumulator layout (weights are the MFMA row operand): lane = pixel r, registers = channels
  // nt * 32 + 8j + 4h + e of that pixel.  The accumulators START at the bias, read from LDS per row block
  // (16 live registers fewer than carrying the lane's 32 bias values across tiles).
  __shared__ __attribute__((aligned(16))) float sb[64];
  if (tid < 64) sb[tid] = bias ? bias[tid] : 0.0f;
  // Gram accumulators of this wave's pixels: 32x32 blocks (0,0), (0,1), (1,1) of the 64x64 matrix
  // (rows c1 = 8(i>>2) + 4h + (i&3), column c2 = r: the layout stv_gram_partial's slabs are written from)
  f32x16 gacc[3];
  constexpr int GT_PITCH = 64 * 2 + 64;                 // bytes per pixel row of a patch; = 64 (mod 256): tr reads conflict-free
  __shared__ __attribute__((aligned(16))) char gt_lds[GRAM ? 4 * 32 * GT_PITCH : 16];
  char* const gpatch = gt_lds + (GRAM ? wave * 32 * GT_PITCH : 0);      // per wave: 32 pixels x 64 channels, no barrier needed
  const int tr_lane = (((lane >> 4) >> 1) * 8 + ((lane & 15) >> 2)) * GT_PITCH + (((lane >> 4) & 1) * 16 + (lane & 3) * 4) * 2;
  if (GRAM) {
#pragma unroll
    for (int b = 0; b < 3; ++b)
#pragma unroll
      for (int i = 0; i < 16; ++i) gacc[b][i] = 0.0f;
  }
  for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
  const int x0 = (t % tiles_x) * MF_TW, y0 = (t / tiles_x) * MF_TH;
  __syncthreads();                       // every wave is done gathering from the previous tile
#pragma unroll
  for (int k = 0; k < NPRE; ++k)
    if (k * 256 + tid < 3 * MF_PLANE) {
      const bf16_t hi = f32_to_bf16(pre[k]);
      xs[k * 256 + tid] = ((uint32_t)hi << 16) | (uint32_t)f32_to_bf16(pre[k] - bf16_to_f32(hi));
    }
  if (t + (int)gridDim.x < ntiles && !(STV_FIRST_DIAG & 4)) request(t + gridDim.x);
  __syncthreads();
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) {
    if (STV_FIRST_DIAG & 1) break;          // timing only: no gathers, no MFMAs
    const int base = (wave * 2 + mt) * MF_IW + r;
    f32x16 acc[2];                          // one row block at a time: product, store, Gram (register budget of two waves per SIMD)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const f32x4 b4 = *reinterpret_cast<const f32x4*>(&sb[nt * 32 + 8 * j + 4 * h]);
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[nt][4 * j + e] = b4[e];
      }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      uint32_t hi4[4], lo4[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        uint32_t w[2];
#pragma unroll
        for (int e2 = 0; e2 < 2; ++e2) {
          const int o = koff[ks][2 * q + e2];
          w[e2] = xs[o >= 0 ? base + o : 3 * MF_PLANE];
        }
        hi4[q] = __builtin_amdgcn_perm(w[1], w[0], 0x07060302u);    // (hi of w1) : (hi of w0)
        lo4[q] = __builtin_amdgcn_perm(w[1], w[0], 0x05040100u);    // (lo of w1) : (lo of w0)
      }
      const bf16x8v a_hi = __builtin_bit_cast(bf16x8v, (u32x4){hi4[0], hi4[1], hi4[2], hi4[3]});
      const bf16x8v a_lo = __builtin_bit_cast(bf16x8v, (u32x4){lo4[0], lo4[1], lo4[2], lo4[3]});
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bw[0][ks][nt], a_lo, acc[nt], 0, 0, 0);
        acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bw[1][ks][nt], a_hi, acc[nt], 0, 0, 0);
        acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bw[0][ks][nt], a_hi, acc[nt], 0, 0, 0);
      }
    }
    // epilogue in registers: bias, pack groups of 4 channels, trade groups with the partner
    // half-wave (v_permlane32_swap) so that every lane stores 16 contiguous bytes - no LDS transpose
    {
    const int gy = y0 + wave * 2 + mt, gx = x0 + r;
    const bool ok = gy < H && gx < W;
    bf16_t* dst = y + ((size_t)gy * W + gx) * 64;
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      uint32_t px[4], py[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        px[j] = pack_bf16x2(acc[nt][4 * j], acc[nt][4 * j + 1]);
        py[j] = pack_bf16x2(acc[nt][4 * j + 2], acc[nt][4 * j + 3]);
      }
#pragma unroll
      for (int jp = 0; jp < 4; jp += 2) {
        const auto sx = __builtin_amdgcn_permlane32_swap(px[jp], px[jp + 1], false, false);
        const auto sy = __builtin_amdgcn_permlane32_swap(py[jp], py[jp + 1], false, false);
        const u32x4 out = {sx[0], sy[0], sx[1], sy[1]};     // lanes 0-31: channels 8jp..8jp+7, lanes 32-63: the next eight
        if (ok && !(STV_FIRST_DIAG & 2)) *reinterpret_cast<u32x4*>(dst + nt * 32 + 8 * jp + 8 * h) = out;
        if ((STV_FIRST_DIAG & 2) && out[0] == 0x12345678u) dst[0] = 1;   // keep the work alive
        if (GRAM)      // the stored words, pixel-major, into this wave's patch (zeros for pixels outside the image)
          *reinterpret_cast<u32x4*>(gpatch + r * GT_PITCH + (nt * 32 + 8 * jp + 8 * h) * 2) = ok ? out : (u32x4){0u, 0u, 0u, 0u};
      }
    }
    if (GRAM) {
      // R += Y^T Y over this row block's 32 pixels: operands [channel][8 pixels] straight from the patch with
      // the hardware transpose read (the gram kernel's addressing, gram.hip: one 4x16 block per ds_read_b64_tr_b16)
      typedef __attribute__((ext_vector_type(8))) short s16x8;
      bf16x8v op[2][2];                      // [channel block][k-step of 16 pixels]
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) {
          const char* pa = gpatch + tr_lane + kk * 16 * GT_PITCH + cb * 64;
          const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(pa));
          const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(pa + 4 * GT_PITCH));
          const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
          op[cb][kk] = __builtin_bit_cast(bf16x8v, v);
        }
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        gacc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(op[0][kk], op[0][kk], gacc[0], 0, 0, 0);
        gacc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(op[0][kk], op[1][kk], gacc[1], 0, 0, 0);
        gacc[2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(op[1][kk], op[1][kk], gacc[2], 0, 0, 0);
      }
    }
  }
  }
  }   // tiles
  if (GRAM) {
    // the four waves' sums meet in LDS in wave order (deterministic); block (1,0) is the mirror image of (0,1)
    __shared__ float gs[64 * 64];
#pragma unroll 1
    for (int w = 0; w < 4; ++w) {
      if (wave == w) {
#pragma unroll
        for (int b = 0; b < 3; ++b)
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            const int row = (b == 2 ? 32 : 0) + 8 * (i >> 2) + 4 * h + (i & 3), col = (b == 0 ? 0 : 32) + r;
            gs[row * 64 + col] = (w == 0 ? 0.0f : gs[row * 64 + col]) + gacc[b][i];
          }
      }
      __syncthreads();
    }
    float* slab = gram_slabs + (size_t)blockIdx.x * (64 * 64);
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const int idx = k * 256 + tid, row = idx >> 6, col = idx & 63;
      slab[idx] = (row >= 32 && col < 32) ? gs[col * 64 + row] : gs[idx];
    }
  }
#endif
}

// ---- input gradient on the matrix cores (bf16 activations) -----------------------------------------
// dx[c][p] = sum_tap sum_n dy[p + off(tap)][n] * wd[tap][c][n]: M = pixels, K = 64 channels per tap, N = 3.
// The three HORIZONTAL taps of a kernel row ride in the 16 columns of v_mfma_f32_16x16x32_bf16 next to the
// three channels: column 4 dxo + c.  One A fragment (16 pixels x 32 channels of a dy row) then serves all
// three horizontal taps at once - a third of the MFMAs and of the LDS reads of a tap-by-tap product
// (9 x 2 x 2 MFMAs per 16 pixels, 13 of 16 columns idle) - and what comes out is D[p'][dxo, c], the
// contribution of input pixel p' to output pixel p' - dxo: a shifted three-term sum finishes the job through
// a 6-KB per-wave LDS patch.  dy is bf16 as stored; the weights are split in two bf16 terms so the product
// stays fp32-faithful.  36 lanes of a weight fragment are non-zero (9 columns x 4 k-groups): a 7 KB table.
#ifndef STV_DG_TH
#define STV_DG_TH 8
#endif
constexpr int DG_TH = STV_DG_TH, DG_TW = 32, DG_IW = DG_TW + 2, DG_PITCH = 128 + 16;
constexpr int DG_RW = DG_TH / 4;                            // output rows per wave
constexpr int DG_CB = 3;                                    // 16-pixel MFMA row blocks across the 34-pixel halo row (48 >= 34)
static_assert(DG_TH % 4 == 0, "four waves split the rows");
constexpr int DG_TILE_BYTES = (DG_TH + 2) * DG_IW * DG_PITCH;
constexpr int DG_FRAGS = 3 * 2 * 2 * 36;                    // [dy][kstep][hi|lo][(dxo * 3 + c) * 4 + g], 16 bytes each
constexpr int DG_FRAG_WORDS = DG_FRAGS * 4;
constexpr int DG_D_BYTES = 4 * DG_RW * (DG_CB * 16) * 16 * 4;   // per workgroup: [wave][row][pixel p'][column] fp32

__global__ void pack_first_dgrad_fragments(const float* __restrict__ wf, uint32_t* __restrict__ frag) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;        // one bf16 pair per thread
  if (i >= DG_FRAG_WORDS) return;
  const int q = i & 3, ent = (i >> 2) % 36, rest = (i >> 2) / 36;
  const int part = rest & 1, ks = (rest >> 1) & 1, dyo = rest >> 2;
  const int g = ent & 3, c = (ent >> 2) % 3, dxo = (ent >> 2) / 3;
  const int tap = dyo * 3 + dxo;
  uint32_t out = 0;
  for (int e2 = 0; e2 < 2; ++e2) {
    const int n = ks * 32 + 8 * g + 2 * q + e2;
    const float v = wf[((8 - tap) * 64 + n) * 3 + c];         // flipped taps: the read offset becomes +off(tap)
    const bf16_t hi = f32_to_bf16(v);
    const bf16_t lo = f32_to_bf16(v - bf16_to_f32(hi));
    out |= (uint32_t)(part ? lo : hi) << (16 * e2);
  }
  frag[i] = out;
}

__global__ __launch_bounds__(256) void conv_first_dgrad_mfma(const bf16_t* __restrict__ dy,
                                                             const uint32_t* __restrict__ frag,
                                                             float* __restrict__ dx, int H, int W) {
#if defined(__HIP_DEVICE_COMPILE__)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* tile = smem;
  u32x4* wtab = reinterpret_cast<u32x4*>(smem + DG_TILE_BYTES);          // DG_FRAGS entries + one zero entry
  float* dpatch = reinterpret_cast<float*>(smem + DG_TILE_BYTES + (DG_FRAGS + 1) * 16);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, g = lane >> 4;
  const int tiles_x = (W + DG_TW - 1) / DG_TW;
  const int ntiles = tiles_x * ((H + DG_TH - 1) / DG_TH);
  const u32x4 zero4 = {0u, 0u, 0u, 0u};
  // Persistent workgroup: the dy halo tile of the NEXT image tile is requested into registers
  // before the current one is multiplied, so the HBM round trip overlaps the MFMA phase.
  constexpr int NVEC = (DG_TH + 2) * DG_IW * 8;              // 16-byte vectors of a halo tile
  constexpr int NPRE = (NVEC + 255) / 256;
  u32x4 pre[NPRE];
  int pyx[NPRE];            // (row << 8 | column) of the halo pixel behind each of this thread's loads: tile-independent
#pragma unroll
  for (int k = 0; k < NPRE; ++k) {
    const int i = k * 256 + tid;
    const int p = i >> 3;
    pyx[k] = i < NVEC ? (((p / DG_IW) << 8) | (p % DG_IW)) : -1;
  }
  auto request = [&](int t) {
    const int tx0 = (t % tiles_x) * DG_TW, ty0 = (t / tiles_x) * DG_TH;
#pragma unroll
    for (int k = 0; k < NPRE; ++k) {
      const int v = tid & 7;                                   // (k * 256 + tid) & 7
      const int gy = ty0 + (pyx[k] >> 8) - 1, gx = tx0 + (pyx[k] & 255) - 1;
      const bool ok = pyx[k] >= 0 && gy >= 0 && gy < H && gx >= 0 && gx < W;
      pre[k] = ok ? *reinterpret_cast<const u32x4*>(dy + ((size_t)gy * W + gx) * 64 + v * 8) : zero4;
    }
  };
  request(blockIdx.x);
  for (int i = tid; i < DG_FRAGS; i += 256) wtab[i] = *reinterpret_cast<const u32x4*>(frag + 4 * i);
  if (tid == 0) wtab[DG_FRAGS] = zero4;
  typedef __attribute__((ext_vector_type(4))) float acc_t;
  // B column = lane & 15 = 4 dxo + c (c < 3): nine live columns, the rest multiply zeros
  const int ent = ((r & 3) < 3 && r < 12) ? ((r >> 2) * 3 + (r & 3)) * 4 + g : -1;
  const size_t plane = (size_t)H * W;
  float* const dw = dpatch + wave * (DG_RW * DG_CB * 16 * 16);
  for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
    const int x0 = (t % tiles_x) * DG_TW, y0 = (t / tiles_x) * DG_TH;
    __syncthreads();                      // every wave is done reading the previous tile
#pragma unroll
    for (int k = 0; k < NPRE; ++k) {
      const int i = k * 256 + tid;
      if (i < NVEC) *reinterpret_cast<u32x4*>(tile + (i >> 3) * DG_PITCH + (i & 7) * 16) = pre[k];
    }
    if (t + (int)gridDim.x < ntiles) request(t + gridDim.x);
    __syncthreads();
    acc_t acc[DG_RW][DG_CB];
#pragma unroll
    for (int rw = 0; rw < DG_RW; ++rw)
#pragma unroll
      for (int cb = 0; cb < DG_CB; ++cb) acc[rw][cb] = acc_t{0.0f, 0.0f, 0.0f, 0.0f};
    // input rows wave * RW .. + RW + 1 of the halo tile feed output rows rw = ir - dyo
#pragma unroll
    for (int ir = 0; ir < DG_RW + 2; ++ir) {
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        bf16x8v a[DG_CB];
#pragma unroll
        for (int cb = 0; cb < DG_CB; ++cb) {
          // pixel p' = 16 cb + r of the halo row (p' >= 34 reads past the row: those D rows are never used)
          const int row = wave * DG_RW + ir, col = cb * 16 + r;
          a[cb] = __builtin_bit_cast(bf16x8v, *reinterpret_cast<const u32x4*>(tile + (row * DG_IW + col) * DG_PITCH + ks * 64 + g * 16));
        }
#pragma unroll
        for (int dyo = 0; dyo < 3; ++dyo) {
          const int rw = ir - dyo;
          if (rw < 0 || rw >= DG_RW) continue;
          const int fb = ((dyo * 2 + ks) * 2) * 36;
          const bf16x8v b_hi = __builtin_bit_cast(bf16x8v, wtab[ent >= 0 ? fb + ent : DG_FRAGS]);
          const bf16x8v b_lo = __builtin_bit_cast(bf16x8v, wtab[ent >= 0 ? fb + 36 + ent : DG_FRAGS]);
#pragma unroll
          for (int cb = 0; cb < DG_CB; ++cb) {
            acc[rw][cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[cb], b_lo, acc[rw][cb], 0, 0, 0);
            acc[rw][cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[cb], b_hi, acc[rw][cb], 0, 0, 0);
          }
        }
      }
    }
    // D: column (lane & 15) = 4 dxo + c, rows 4g..4g+3 = pixels p' of the block.  Through the wave's own
    // LDS patch: dx[x][c] = D[x + 0][0, c] + D[x + 1][1, c] + D[x + 2][2, c]  (x = pixel of the 32-wide tile)
#pragma unroll
    for (int rw = 0; rw < DG_RW; ++rw)
#pragma unroll
      for (int cb = 0; cb < DG_CB; ++cb)
#pragma unroll
        for (int i = 0; i < 4; ++i) dw[(rw * (DG_CB * 16) + cb * 16 + 4 * g + i) * 16 + r] = acc[rw][cb][i];
    // (same wave wrote and reads: program order + the LDS counter suffice, no workgroup barrier)
#pragma unroll
    for (int k = 0; k < (DG_RW * 3 * DG_TW + 63) / 64; ++k) {
      const int o = k * 64 + lane;                  // (row, channel, pixel): consecutive lanes = consecutive pixels
      if (o >= DG_RW * 3 * DG_TW) break;
      const int xl = o % DG_TW, c = (o / DG_TW) % 3, rw = o / (3 * DG_TW);
      const float* base = dw + (rw * (DG_CB * 16) + xl) * 16 + c;
      const float v = base[0] + base[16 + 4] + base[32 + 8];      // p' = xl + dxo, column 4 dxo + c
      const int gy = y0 + wave * DG_RW + rw, gx = x0 + xl;
      if (gy < H && gx < W) dx[c * plane + (size_t)gy * W + gx] = v;
    }
  }
#endif
}

// repack [9][64][3] -> [9][3][64] (fwd) or flipped taps (dgrad); tiny, run per call on the stream
__global__ void repack_first_weights(const float* __restrict__ wf, float* __restrict__ wt, int flip) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= 9 * 3 * 64) return;
  const int n = i % 64, c = (i / 64) % 3, tap = i / 192;
  const int src_tap = flip ? 8 - tap : tap;
  wt[i] = wf[(src_tap * 64 + n) * 3 + c];
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

// `packed` (optional): the caller's buffer from stv_conv_first_pack.  The 3 -> 64 kernels read the weights in their
// kernel-side order only; a caller without that buffer gets the shape-generic kernels below, which read `wf` as it is
// (the library keeps no scratch of its own: nothing hidden is shared between streams, threads or devices).
template <typename T>
int fwd_typed(const float* x, const float* wf, const float* packed, const float* bias, void* y, int H, int W,
              int cin, int cout, hipStream_t st, float* gram_slabs = nullptr) {
  if (cin == 3 && cout == 64 && packed && std::is_same<T, bf16_t>::value && !getenv("STV_FIRST_VALU")) {
    const int tiles = ceil_div(W, MF_TW) * ceil_div(H, MF_TH);
    static const int wg_per_cu = getenv("STV_FIRST_WGS") ? atoi(getenv("STV_FIRST_WGS")) : 2;   // swept 2..8: 2 is fastest at 512^2 and 1024^2
    const int grid = tiles < wg_per_cu * 256 ? tiles : wg_per_cu * 256;   // resident workgroups walk the tiles
    const uint32_t* fr = reinterpret_cast<const uint32_t*>(packed + 2 * 1728);
    if (gram_slabs)     // one slab per workgroup: exactly the split the finish kernel will walk (idle workgroups write zeros)
      hipLaunchKernelGGL(conv_first_fwd_mfma<true>, dim3(stv_gram_ksplit(H * W, 64)), dim3(256), 0, st, x, fr, bias,
                         static_cast<bf16_t*>(y), gram_slabs, H, W);
    else
      hipLaunchKernelGGL(conv_first_fwd_mfma<false>, dim3(grid), dim3(256), 0, st, x, fr, bias, static_cast<bf16_t*>(y),
                         nullptr, H, W);
    STV_CHECK_LAUNCH();
    return STV_OK;
  }
  if (gram_slabs) return STV_ERR_ARG;
  if (cin == 3 && cout == 64 && packed) {
    const float* wt = packed;
    const int tiles = ceil_div(W, FT) * ceil_div(H, FT);
    hipLaunchKernelGGL(conv_first_fwd_c64<T>, dim3(tiles), dim3(256), 0, st, x, wt, bias, static_cast<T*>(y), H, W);
    STV_CHECK_LAUNCH();
    return STV_OK;
  }
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
int dgrad_typed(const void* dy, const float* wf, const float* packed, float* dx, int H, int W, int cin, int cout,
                hipStream_t st) {
  if (cin == 3 && cout == 64 && packed && std::is_same<T, bf16_t>::value && !getenv("STV_FIRST_VALU")) {
    const int tiles = ceil_div(W, DG_TW) * ceil_div(H, DG_TH);
    constexpr int lds = DG_TILE_BYTES + (DG_FRAGS + 1) * 16 + DG_D_BYTES;
    static const int dg_wgs = getenv("STV_FIRST_DG_WGS") ? atoi(getenv("STV_FIRST_DG_WGS")) : 2;
    const int grid = tiles < dg_wgs * 256 ? tiles : dg_wgs * 256;   // resident workgroups walk the tiles
    hipLaunchKernelGGL(conv_first_dgrad_mfma, dim3(grid), dim3(256), lds, st, static_cast<const bf16_t*>(dy),
                       reinterpret_cast<const uint32_t*>(packed + 2 * 1728 + MF_FRAG_WORDS), dx, H, W);
    STV_CHECK_LAUNCH();
    return STV_OK;
  }
  if (cin == 3 && cout == 64 && packed) {
    const float* wd = packed + 1728;
    const int tiles = ceil_div(W, FT) * ceil_div(H, FT);
    const size_t lds = (size_t)FH * FH * (64 * sizeof(T) + 16);
    if (stv_set_max_lds(reinterpret_cast<const void*>(&conv_first_dgrad_c64<T>), (int)lds) != STV_OK) return STV_ERR_LAUNCH;
    hipLaunchKernelGGL(conv_first_dgrad_c64<T>, dim3(tiles), dim3(256), lds, st, static_cast<const T*>(dy), wd, dx,
                       H, W);
    STV_CHECK_LAUNCH();
    return STV_OK;
  }
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

extern "C" size_t stv_conv_first_packed_bytes(int cin, int cout) {
  const size_t plain = (size_t)2 * 9 * (size_t)(cin > 0 ? cin : 0) * (size_t)(cout > 0 ? cout : 0) * sizeof(float);
  return (cin == 3 && cout == 64) ? plain + (size_t)(MF_FRAG_WORDS + DG_FRAG_WORDS) * 4 : plain;   // + matrix-core fragments
}

extern "C" int stv_conv_first_pack(const float* wf, float* packed, int cin, int cout, void* stream) {
  if (!wf || !packed || cin <= 0 || cout <= 0) return STV_ERR_ARG;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (cin == 3 && cout == 64) {
    hipLaunchKernelGGL(repack_first_weights, dim3(7), dim3(256), 0, st, wf, packed, 0);
    hipLaunchKernelGGL(repack_first_weights, dim3(7), dim3(256), 0, st, wf, packed + 1728, 1);
    hipLaunchKernelGGL(pack_first_fragments, dim3(MF_FRAG_WORDS / 256), dim3(256), 0, st, wf,
                       reinterpret_cast<uint32_t*>(packed + 2 * 1728));
    hipLaunchKernelGGL(pack_first_dgrad_fragments, dim3((DG_FRAG_WORDS + 255) / 256), dim3(256), 0, st, wf,
                       reinterpret_cast<uint32_t*>(packed + 2 * 1728 + MF_FRAG_WORDS));
    STV_CHECK_LAUNCH();
    return STV_OK;
  }
  // the other shapes read the weights as they are: the packed form is a copy
  if (hipMemcpyAsync(packed, wf, (size_t)9 * cin * cout * sizeof(float), hipMemcpyDeviceToDevice, st) != hipSuccess)
    return STV_ERR_LAUNCH;
  return STV_OK;
}

extern "C" int stv_conv_first_fwd_packed(const float* x_nchw, const float* packed, const float* bias, void* y,
                                         int H, int W, int cin, int cout, int dtype, void* stream) {
  if (!x_nchw || !packed || !y || H <= 0 || W <= 0 || cin <= 0 || cout <= 0) return STV_ERR_ARG;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (dtype == STV_F32) return fwd_typed<float>(x_nchw, packed, packed, bias, y, H, W, cin, cout, st);
  if (dtype == STV_BF16) return fwd_typed<bf16_t>(x_nchw, packed, packed, bias, y, H, W, cin, cout, st);
  return STV_ERR_ARG;
}

extern "C" int stv_conv_first_fwd_gram(const float* x_nchw, const float* packed, const float* bias, void* y,
                                       float* gram_partials, int H, int W, int cin, int cout, int dtype, void* stream) {
  if (!x_nchw || !packed || !y || !gram_partials || H <= 0 || W <= 0) return STV_ERR_ARG;
  if (cin != 3 || cout != 64 || dtype != STV_BF16) return STV_ERR_ARG;     // the matrix-core first layer only
  if (!stv_conv_first_gram_supported(H, W, cin, cout, dtype)) return STV_ERR_ARG;
  return fwd_typed<bf16_t>(x_nchw, packed, packed, bias, y, H, W, cin, cout, static_cast<hipStream_t>(stream), gram_partials);
}

extern "C" int stv_conv_first_gram_supported(int H, int W, int cin, int cout, int dtype) {
  return (cin == 3 && cout == 64 && dtype == STV_BF16 && H > 0 && W > 0 && !getenv("STV_FIRST_VALU")) ? 1 : 0;
}

extern "C" int stv_conv_first_dgrad_packed(const void* dy, const float* packed, float* dx_nchw, int H, int W,
                                           int cin, int cout, int dtype, void* stream) {
  if (!dy || !packed || !dx_nchw || H <= 0 || W <= 0 || cin <= 0 || cout <= 0) return STV_ERR_ARG;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (dtype == STV_F32) return dgrad_typed<float>(dy, packed, packed, dx_nchw, H, W, cin, cout, st);
  if (dtype == STV_BF16) return dgrad_typed<bf16_t>(dy, packed, packed, dx_nchw, H, W, cin, cout, st);
  return STV_ERR_ARG;
}

extern "C" int stv_conv_first_fwd(const float* x_nchw, const float* wf, const float* bias, void* y,
                                  int H, int W, int cin, int cout, int dtype, void* stream) {
  if (!x_nchw || !wf || !y || H <= 0 || W <= 0 || cin <= 0 || cout <= 0) return STV_ERR_ARG;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (dtype == STV_F32) return fwd_typed<float>(x_nchw, wf, nullptr, bias, y, H, W, cin, cout, st);
  if (dtype == STV_BF16) return fwd_typed<bf16_t>(x_nchw, wf, nullptr, bias, y, H, W, cin, cout, st);
  return STV_ERR_ARG;
}

extern "C" int stv_conv_first_dgrad(const void* dy, const float* wf, float* dx_nchw, int H, int W,
                                    int cin, int cout, int dtype, void* stream) {
  if (!dy || !wf || !dx_nchw || H <= 0 || W <= 0 || cin <= 0 || cout <= 0) return STV_ERR_ARG;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (dtype == STV_F32) return dgrad_typed<float>(dy, wf, nullptr, dx_nchw, H, W, cin, cout, st);
  if (dtype == STV_BF16) return dgrad_typed<bf16_t>(dy, wf, nullptr, dx_nchw, H, W, cin, cout, st);
  return STV_ERR_ARG;
}
