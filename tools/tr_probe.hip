// Probe (not product code): what does ds_read_b64_tr_b16 return?  LDS holds lds[i] = i (16-bit); lane l passes byte address
// addr(l) and gets 4 x 16-bit values.  Printed for a few address patterns so that the lane <-> element map can be read off.
//   hipcc --offload-arch=gfx950 -O2 -o tools/bin/tr_probe tools/tr_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
__global__ void k(const int* addr, uint16_t* out) {
  __shared__ __attribute__((aligned(16))) uint16_t lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (uint16_t)i;
  __syncthreads();
  const unsigned a = (unsigned)(size_t)(&lds[0]) + addr[threadIdx.x];
  typedef __attribute__((ext_vector_type(2))) unsigned u2;
  u2 v;
  asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a) : "memory");
  out[threadIdx.x * 4 + 0] = v.x & 0xffff; out[threadIdx.x * 4 + 1] = v.x >> 16;
  out[threadIdx.x * 4 + 2] = v.y & 0xffff; out[threadIdx.x * 4 + 3] = v.y >> 16;
}
int main() {
  int* d_addr; uint16_t* d_out;
  hipMalloc(&d_addr, 64 * 4); hipMalloc(&d_out, 64 * 4 * 2);
  const char* names[] = {"addr = lane * 8 bytes (lane l points at elements 4l..4l+3)", "addr = (l&15)*256 + (l>>4)*8  (row l&15 of a [16][128] matrix, 4 elements at column 4*(l>>4))",
                         "addr = (l&3)*8 + (l>>2)*256 (row l>>2, 4-element group l&3)"};
  for (int p = 0; p < 3; ++p) {
    int h[64];
    for (int l = 0; l < 64; ++l) h[l] = p == 0 ? l * 8 : p == 1 ? (l & 15) * 256 + (l >> 4) * 8 : (l & 3) * 8 + (l >> 2) * 256;
    hipMemcpy(d_addr, h, sizeof(h), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d_addr, d_out);
    uint16_t o[256];
    hipMemcpy(o, d_out, sizeof(o), hipMemcpyDeviceToHost);
    printf("pattern %d: %s\n", p, names[p]);
    for (int l = 0; l < 64; ++l) printf("  lane %2d (elem addr %4d): %5d %5d %5d %5d%s", l, h[l] / 2, o[l * 4], o[l * 4 + 1], o[l * 4 + 2], o[l * 4 + 3], (l & 1) ? "\n" : "   |");
  }
  return 0;
}
