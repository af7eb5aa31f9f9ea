// HBM read-bandwidth yardstick for the L-BFGS history sweeps (diagnostic).
//   hipcc --offload-arch=gfx950 -O3 tools/read_probe.cpp -o gpurun_out/read_probe && gpurun_out/read_probe
// Sums a buffer of `vectors` x `n` floats the way pass_a walks the history: every workgroup owns a
// contiguous tile of each vector and visits the vectors one after another.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int U, int PF>
__global__ __launch_bounds__(256) void sweep(const float* __restrict__ h, float* __restrict__ out, size_t nn, int vectors) {
  const size_t base = (size_t)blockIdx.x * (256 * 4 * U) + threadIdx.x * 4;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  f32x4 buf[PF][U];
#pragma unroll
  for (int p = 0; p < PF; ++p)
#pragma unroll
    for (int u = 0; u < U; ++u) buf[p][u] = *reinterpret_cast<const f32x4*>(h + (size_t)p * nn + base + u * 1024);
  for (int v = 0; v < vectors; v += PF) {
#pragma unroll
    for (int p = 0; p < PF; ++p) {
#pragma unroll
      for (int u = 0; u < U; ++u) acc += buf[p][u];
      const int nv = v + p + PF;
      if (nv < vectors) {
#pragma unroll
        for (int u = 0; u < U; ++u) buf[p][u] = *reinterpret_cast<const f32x4*>(h + (size_t)nv * nn + base + u * 1024);
      }
    }
  }
  if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) out[0] = acc[0];
}
template <int U, int PF>
void run(const float* h, float* out, size_t nn, int vectors, const char* name) {
  const int blocks = (int)(nn / (256 * 4 * U));
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((sweep<U, PF>), dim3(blocks), dim3(256), 0, 0, h, out, nn, vectors);
  hipEventRecord(e0, 0);
  const int reps = 10;
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((sweep<U, PF>), dim3(blocks), dim3(256), 0, 0, h, out, nn, vectors);
  hipEventRecord(e1, 0); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double bytes = (double)nn * vectors * 4;
  printf("%-22s blocks %5d  %8.1f us  %5.2f TB/s\n", name, blocks, ms / reps * 1000, bytes / (ms / reps * 1e-3) / 1e12);
}
int main(int argc, char** argv) {
  const int size = argc > 1 ? atoi(argv[1]) : 1024;
  const size_t nn = (size_t)3 * size * size;
  const int vectors = 200;
  float *h, *out;
  if (hipMalloc(&h, nn * vectors * 4) != hipSuccess) return 1;
  hipMalloc(&out, 64);
  hipMemset(h, 0, nn * vectors * 4);
  printf("size %d: %d vectors x %.1f MB\n", size, vectors, nn * 4 / 1e6);
  run<1, 1>(h, out, nn, vectors, "U1 PF1");
  run<2, 1>(h, out, nn, vectors, "U2 PF1");
  run<4, 1>(h, out, nn, vectors, "U4 PF1");
  run<4, 2>(h, out, nn, vectors, "U4 PF2");
  run<2, 2>(h, out, nn, vectors, "U2 PF2");
  run<2, 4>(h, out, nn, vectors, "U2 PF4");
  run<1, 4>(h, out, nn, vectors, "U1 PF4");
  run<1, 8>(h, out, nn, vectors, "U1 PF8");
  run<8, 1>(h, out, nn, vectors, "U8 PF1");
  return 0;
}
