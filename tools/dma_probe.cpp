// Probe of buffer_load ... lds semantics on gfx950: lane-linear destination, per-lane source, out-of-range lanes
// write ZEROS (the property the conv kernel's halo / tail handling relies on).  build: hipcc --offload-arch=gfx950 -O2 tools/dma_probe.cpp -o tools/dma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
__global__ void k(const uint32_t* g, uint32_t* out, int nrec_bytes) {
  __shared__ __attribute__((aligned(16))) uint32_t lds[64 * 4 * 2];
  const int lane = threadIdx.x;
  for (int i = lane; i < 512; i += 64) lds[i] = 0xDEADBEEFu;
  __syncthreads();
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t*>(g), 0, nrec_bytes, 0x00020000);
  const uint32_t voff = (uint32_t)((63 - lane) * 16);
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)lds, 16, voff, 0, 0, 0);
  // second piece lands 1 KiB further via the immediate offset
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(lds + 256), 16, (uint32_t)(lane * 16), 0, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = lane; i < 512; i += 64) out[i] = lds[i];
}
int main() {
  std::vector<uint32_t> h(256);
  for (int i = 0; i < 256; ++i) h[i] = i;
  uint32_t *g, *o;
  hipMalloc(&g, 1024); hipMalloc(&o, 2048);
  hipMemcpy(g, h.data(), 1024, hipMemcpyHostToDevice);
  for (int nrec : {1024, 512}) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, g, o, nrec);
    std::vector<uint32_t> r(512);
    hipMemcpy(r.data(), o, 2048, hipMemcpyDeviceToHost);
    printf("nrec=%d\n first piece (reversed source): ", nrec);
    for (int l : {0, 1, 2, 31, 32, 62, 63}) printf("[lane-slot %d] %x ", l, r[l * 4]);
    printf("\n second piece: ");
    for (int l : {0, 1, 31, 32, 63}) printf("[slot %d] %x ", l, r[256 + l * 4]);
    printf("\n");
  }
  return 0;
}
