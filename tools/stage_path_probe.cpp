// What does it cost a wave to move one 1-KiB piece from global memory (L2-resident) into LDS between its MFMAs - as ONE
// LDS-DMA instruction (buffer_load ... lds, the conv kernels' staging path) or as a register-staged pair (buffer_load_b128
// into VGPRs, ds_write_b128 four iterations later)?  (diagnostic; round 5: VERDICT r4 item 3 asked for the weights off the
// LDS-DMA path; direct-to-register fragments were costed out (L1 bandwidth x sharing waves) - this is the other variant,
// still shared through LDS.)
//   hipcc --offload-arch=gfx950 -O3 tools/stage_path_probe.cpp -o gpurun_out/stage_path_probe && gpurun_out/stage_path_probe
// One workgroup per CU; 4 or 8 waves (1 or 2 per SIMD).  Per loop iteration a wave moves one piece and issues MPP
// v_mfma_f32_32x32x16_bf16 on four accumulators, each with one ds_read_b128 of its B operand (the conv kernels' ~0.8-1.2 LDS
// fragment reads per MFMA).  MPP = 6 / 3 = the 0.17 / 0.35 pieces per MFMA of the 8x128 / 4x64 tiles.  Prints clock ticks per
// MFMA per wave (s_memtime) and ns per MFMA per SIMD from the wall time.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

constexpr int kSlots = 8;            // 1-KiB LDS slots per wave
constexpr int kDepth = 4;            // pieces in flight per wave

template <int MODE, int MPP, int WAVES, int BAR = 0>
__global__ __launch_bounds__(WAVES * 64) void probe(const unsigned int* src, int src_bytes, float* out, unsigned long long* cycles, int iters, int shared) {
  using lds_ptr = __attribute__((address_space(3))) void*;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  char* mine = smem + wave * (kSlots * 1024);
  for (int i = lane; i < kSlots * 256; i += 64) reinterpret_cast<unsigned int*>(mine)[i] = 0x3C003C00u + (unsigned)i;
  __syncthreads();
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned int*>(src), 0, src_bytes, 0x00020000);
  bf16x8 a;
  for (int i = 0; i < 8; ++i) a[i] = (__bf16)(0.001f * (float)(lane + i));
  f32x16 acc[4];
  for (int k = 0; k < 4; ++k)
    for (int i = 0; i < 16; ++i) acc[k][i] = 0.0f;
  u32x4 st[kDepth];
  for (int d = 0; d < kDepth; ++d) st[d] = u32x4{0u, 0u, 0u, 0u};
  // SHARED = every workgroup reads the same 32 pieces per wave (a filter bank: L2 hits); else each its own (64 MB in all)
  const unsigned int base = (unsigned)(((shared ? 0 : blockIdx.x) * WAVES + wave) * 1024 + lane * 16);
  const unsigned int stride = (unsigned)((shared ? 1 : gridDim.x) * WAVES * 1024);      // 32 pieces per wave, walked round and round
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; it += kDepth) {
#pragma unroll
    for (int d = 0; d < kDepth; ++d) {
      char* slot = mine + ((it + d) & (kSlots - 1)) * 1024;
      const unsigned int off = base + (unsigned)((it + d) & 31) * stride;
      if (MODE == 1) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr)slot, 16, off, 0, 0, 0);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kDepth - 1) : "memory");
      } else if (MODE == 2) {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kDepth - 1) : "memory");      // the load issued kDepth pieces ago
        *reinterpret_cast<u32x4*>(slot + lane * 16) = st[d];
        st[d] = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0);
      }
#pragma unroll
      for (int m = 0; m < MPP; ++m) {
        const bf16x8 b = *reinterpret_cast<const bf16x8*>(mine + ((m + d) & (kSlots - 1)) * 1024 + lane * 16);
        acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[m & 3], 0, 0, 0);
      }
      // BAR: a K-stage boundary every BAR pieces - everything this wave has in flight but the newest piece landed, then
      // the workgroup barrier (the conv kernels' ring hand-over)
      if (BAR > 0 && ((it + d) % BAR) == BAR - 1) {
        asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
        __builtin_amdgcn_s_barrier();
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.0f;
  for (int k = 0; k < 4; ++k) s += acc[k][0] + acc[k][9];
  unsigned int vs = 0;
  for (int d = 0; d < kDepth; ++d) vs ^= st[d][0];
  if (s == 12345.678f || vs == 0x12345679u) out[0] = s;
  if (lane == 0) cycles[blockIdx.x * WAVES + wave] = t1 - t0;
}

template <int MODE, int MPP, int WAVES, int BAR = 0>
void run(const char* name, int cus, const unsigned int* src, int src_bytes, int shared = 1) {
  const int iters = 4000;
  float* out;
  unsigned long long* cyc;
  (void)hipMalloc(&out, 4);
  (void)hipMalloc(&cyc, sizeof(unsigned long long) * cus * WAVES);
  const int lds = WAVES * kSlots * 1024;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&probe<MODE, MPP, WAVES, BAR>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  probe<MODE, MPP, WAVES, BAR><<<cus, WAVES * 64, lds>>>(src, src_bytes, out, cyc, 200, shared);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  probe<MODE, MPP, WAVES, BAR><<<cus, WAVES * 64, lds>>>(src, src_bytes, out, cyc, iters, shared);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms = 0;
  (void)hipEventElapsedTime(&ms, e0, e1);
  unsigned long long* h = (unsigned long long*)malloc(sizeof(unsigned long long) * cus * WAVES);
  (void)hipMemcpy(h, cyc, sizeof(unsigned long long) * cus * WAVES, hipMemcpyDeviceToHost);
  double mean = 0;
  for (int i = 0; i < cus * WAVES; ++i) mean += (double)h[i];
  mean /= cus * WAVES;
  const double mfmas = (double)iters * MPP;
  const double per_simd = mfmas * WAVES / 4.0;
  printf("%-22s %s barrier every %2d pieces, %d MFMAs per piece, %d wave(s) per SIMD: %6.1f ticks per MFMA per wave, %6.2f ns per MFMA per SIMD (32 cycles at 1.78 GHz = 18.0)\n",
         name, shared ? "shared source " : "private source", BAR, MPP, WAVES / 4, mean / mfmas, ms * 1e6 / per_simd);
  free(h);
  (void)hipFree(out);
  (void)hipFree(cyc);
}

int main() {
  hipDeviceProp_t p;
  (void)hipGetDeviceProperties(&p, 0);
  const int cus = p.multiProcessorCount;
  const int src_bytes = 64 << 20;                     // 64 MB: L2 / Infinity-Cache resident after the warm-up launch
  unsigned int* src;
  (void)hipMalloc(&src, src_bytes);
  (void)hipMemset(src, 0x3C, src_bytes);
  printf("%s, %d CUs\n", p.gcnArchName, cus);
  run<0, 6, 4>("no staging", cus, src, src_bytes);
  run<1, 6, 4>("LDS-DMA piece", cus, src, src_bytes);
  run<2, 6, 4>("load + ds_write piece", cus, src, src_bytes);
  run<0, 6, 8>("no staging", cus, src, src_bytes);
  run<1, 6, 8>("LDS-DMA piece", cus, src, src_bytes);
  run<2, 6, 8>("load + ds_write piece", cus, src, src_bytes);
  run<0, 3, 4>("no staging", cus, src, src_bytes);
  run<1, 3, 4>("LDS-DMA piece", cus, src, src_bytes);
  run<2, 3, 4>("load + ds_write piece", cus, src, src_bytes);
  run<0, 3, 8>("no staging", cus, src, src_bytes);
  run<1, 3, 8>("LDS-DMA piece", cus, src, src_bytes);
  run<2, 3, 8>("load + ds_write piece", cus, src, src_bytes);
  run<1, 6, 8>("LDS-DMA piece", cus, src, src_bytes, 0);
  run<2, 6, 8>("load + ds_write piece", cus, src, src_bytes, 0);
  run<1, 12, 8>("LDS-DMA piece", cus, src, src_bytes);
  // the 8x128 tile: 6 pieces and 36 MFMAs per wave and K-stage; the 4x64 tile: ~6 pieces and 18 MFMAs
  run<0, 6, 8, 6>("no staging", cus, src, src_bytes);
  run<1, 6, 8, 6>("LDS-DMA piece", cus, src, src_bytes);
  run<1, 6, 8, 12>("LDS-DMA piece", cus, src, src_bytes);
  run<1, 6, 8, 24>("LDS-DMA piece", cus, src, src_bytes);
  run<0, 3, 4, 6>("no staging", cus, src, src_bytes);
  run<1, 3, 4, 6>("LDS-DMA piece", cus, src, src_bytes);
  run<1, 3, 4, 12>("LDS-DMA piece", cus, src, src_bytes);
  run<1, 3, 8, 6>("LDS-DMA piece", cus, src, src_bytes);
  run<2, 12, 8>("load + ds_write piece", cus, src, src_bytes);
  return 0;
}
