// How often can ONE wave issue independent v_mfma_f32_32x32x16_bf16, and how does that change with a second (fourth) wave on
// the SIMD?  (diagnostic; round 5: the weight-stationary conv kernel runs one wave per SIMD and its matrix pipe is 0.494 busy
// even in the plain form - 64.6 cycles per MFMA against the 32 the pipe needs.)
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_cadence_probe.cpp -o gpurun_out/mfma_cadence_probe && gpurun_out/mfma_cadence_probe
// Every wave runs ITERS x 8 MFMAs back to back on NACC independent accumulators (no memory traffic, operands in registers)
// and, in the "valu" variants, K packed integer maxes between consecutive MFMAs (the shape of the conv kernel's ReLU-on-load);
// one workgroup per CU (grid = CU count), 4 / 8 / 16 waves per workgroup = 1 / 2 / 4 waves per SIMD.  Prints cycles per MFMA
// per wave from s_memtime (100 MHz REFCLK is not what s_memtime counts on gfx9: it counts shader clocks) and from the wall time.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(2))) short s16x2;

template <int NACC, int VALU>
__global__ void probe(float* out, unsigned long long* cycles, int iters) {
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) {
    a[i] = (__bf16)(0.001f * (float)(threadIdx.x + i));
    b[i] = (__bf16)(0.002f * (float)(threadIdx.x + 2 * i));
  }
  f32x16 acc[NACC];
  for (int k = 0; k < NACC; ++k)
    for (int i = 0; i < 16; ++i) acc[k][i] = 0.0f;
  unsigned int v[8];
  for (int i = 0; i < 8; ++i) v[i] = threadIdx.x * 2654435761u + i;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      acc[m % NACC] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[m % NACC], 0, 0, 0);
#pragma unroll
      for (int q = 0; q < VALU; ++q)
        v[(m + q) & 7] = __builtin_bit_cast(unsigned int, __builtin_elementwise_max(
            __builtin_bit_cast(s16x2, v[(m + q) & 7] ^ v[(m + q + 3) & 7]), (s16x2)((short)0)));      // v_xor + v_pk_max_i16: 2 VALU per q
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.0f;
  for (int k = 0; k < NACC; ++k) s += acc[k][0] + acc[k][7];
  unsigned int vs = 0;
  for (int i = 0; i < 8; ++i) vs ^= v[i];
  if (s == 12345.678f || vs == 0x12345678u) out[0] = s;
  if ((threadIdx.x & 63) == 0) cycles[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

// The weight-stationary conv kernel's tile loop without its DMA, epilogue and barrier: 36 resident weight fragments (pinned to
// AGPRs when PIN), a 10 x 34-pixel x 64-channel halo tile in LDS, per tap column six ds_read_b128 requested two columns ahead,
// 12 MFMAs per column on four accumulators - 144 MFMAs and 72 LDS reads per "tile".
template <bool PIN>
__global__ __launch_bounds__(256) void probe_ws(float* out, unsigned long long* cycles, int iters) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int IN_W = 34, KB = 32, IN_STAGE = 11 * 1024;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, r = lane & 31, h = lane >> 5;
  for (int i = tid; i < 4 * IN_STAGE / 4; i += blockDim.x) reinterpret_cast<unsigned int*>(smem)[i] = 0x3C003C00u + i * 2654435761u % 4096u;
  __syncthreads();
  bf16x8 w[4][9];
  for (int s = 0; s < 4; ++s)
    for (int t = 0; t < 9; ++t) {
      for (int i = 0; i < 8; ++i) w[s][t][i] = (__bf16)(0.001f * (float)(lane + s * 9 + t + i));
      if (PIN) asm volatile("" : "+a"(w[s][t]));
    }
  int a_addr[3][6];
  for (int dx = 0; dx < 3; ++dx)
    for (int j = 0; j < 6; ++j) {
      const int pix = (wm * 4 + j) * IN_W + dx + r;
      a_addr[dx][j] = pix * KB + ((h ^ ((pix >> 3) & 1)) << 4);
    }
  f32x16 acc[4];
  for (int k = 0; k < 4; ++k)
    for (int i = 0; i < 16; ++i) acc[k][i] = 0.0f;
  bf16x8 af[3][6];
  auto load_col = [&](int col, int set) {
    const int s = col / 3, dx = col - s * 3;
#pragma unroll
    for (int j = 0; j < 6; ++j) af[set][j] = *reinterpret_cast<const bf16x8*>(smem + s * IN_STAGE + a_addr[dx][j]);
  };
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    load_col(0, 0);
    load_col(1, 1);
#pragma unroll
    for (int col = 0; col < 12; ++col) {
      const int s = col / 3, dx = col - s * 3;
      if (col + 2 < 12) load_col(col + 2, (col + 2) % 3);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int dy = 0; dy < 3; ++dy)
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
          acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[s][dy * 3 + dx], af[col % 3][mt + dy], acc[mt], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float sum = 0.0f;
  for (int k = 0; k < 4; ++k) sum += acc[k][0] + acc[k][7];
  if (sum == 12345.678f) out[0] = sum;
  if ((threadIdx.x & 63) == 0) cycles[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

template <bool PIN>
void run_ws(const char* name, int cus) {
  const int iters = 400, waves_per_wg = 4;
  float* out;
  unsigned long long* cyc;
  (void)hipMalloc(&out, 4);
  (void)hipMalloc(&cyc, sizeof(unsigned long long) * cus * waves_per_wg);
  hipFuncSetAttribute(reinterpret_cast<const void*>(&probe_ws<PIN>), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  probe_ws<PIN><<<cus, 256, 45056>>>(out, cyc, 20);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  probe_ws<PIN><<<cus, 256, 45056>>>(out, cyc, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  unsigned long long* hbuf = (unsigned long long*)malloc(sizeof(unsigned long long) * cus * waves_per_wg);
  hipMemcpy(hbuf, cyc, sizeof(unsigned long long) * cus * waves_per_wg, hipMemcpyDeviceToHost);
  double mean = 0;
  for (int i = 0; i < cus * waves_per_wg; ++i) mean += (double)hbuf[i];
  mean /= cus * waves_per_wg;
  const double mfmas = (double)iters * 144;
  printf("%-46s %7.1f s_memtime ticks per MFMA per wave; wall %.3f ms -> %6.1f ns per MFMA per SIMD\n", name, mean / mfmas, ms, ms * 1e6 / mfmas);
  free(hbuf);
  hipFree(out);
  hipFree(cyc);
}

template <int NACC, int VALU>
void run(const char* name, int waves_per_wg, int cus) {
  const int iters = 4000;
  float* out;
  unsigned long long* cyc;
  (void)hipMalloc(&out, 4);
  (void)hipMalloc(&cyc, sizeof(unsigned long long) * cus * waves_per_wg);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  probe<NACC, VALU><<<cus, 64 * waves_per_wg>>>(out, cyc, 100);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  probe<NACC, VALU><<<cus, 64 * waves_per_wg>>>(out, cyc, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  unsigned long long* h = (unsigned long long*)malloc(sizeof(unsigned long long) * cus * waves_per_wg);
  hipMemcpy(h, cyc, sizeof(unsigned long long) * cus * waves_per_wg, hipMemcpyDeviceToHost);
  double mean = 0;
  for (int i = 0; i < cus * waves_per_wg; ++i) mean += (double)h[i];
  mean /= cus * waves_per_wg;
  const double mfmas = (double)iters * 8;
  const double per_simd = mfmas * waves_per_wg / 4.0;                 // MFMAs one SIMD executed
  printf("%-34s waves/SIMD %d: %7.1f s_memtime ticks per MFMA per wave; wall %.3f ms -> %6.1f ns per MFMA per SIMD = %5.0f TFLOP/s chip-wide\n",
         name, waves_per_wg / 4, mean / mfmas, ms, ms * 1e6 / per_simd,
         per_simd * 4 * cus * 32768.0 / (ms * 1e-3) / 1e12);
  free(h);
  hipFree(out);
  hipFree(cyc);
}

int main() {
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  const int cus = p.multiProcessorCount;
  printf("%s, %d CUs, clock %d MHz\n", p.gcnArchName, cus, p.clockRate / 1000);
  run_ws<false>("WS tile loop (LDS A reads), weights unpinned", cus);
  run_ws<true>("WS tile loop (LDS A reads), weights in AGPRs", cus);
  for (int w : {4, 8, 16}) {
    run<4, 0>("4 independent accumulators", w, cus);
    run<8, 0>("8 independent accumulators", w, cus);
    run<4, 1>("4 acc + ~3 VALU per MFMA", w, cus);
    run<4, 3>("4 acc + ~5 VALU per MFMA", w, cus);
  }
  return 0;
}
