// Probe: semantics of ds_read_b64_tr_b16 on gfx950.  Each lane supplies its own LDS byte address; we print, for every
// lane and element, which LDS element index it received.  hipcc --offload-arch=gfx950 tools/probe_tr.hip -o /tmp/probe_tr
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef __attribute__((ext_vector_type(4))) unsigned short u16x4;
__global__ void k(const int* addr, unsigned short* out) {
  __shared__ __attribute__((aligned(16))) unsigned short lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (unsigned short)i;
  __syncthreads();
  unsigned a = (unsigned)(uintptr_t)lds + (unsigned)addr[threadIdx.x];
  u16x4 v;
  asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a) : "memory");
  for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = v[j];
}
int main() {
  int h_addr[64]; unsigned short h_out[256];
  int* d_addr; unsigned short* d_out;
  hipMalloc(&d_addr, sizeof(h_addr)); hipMalloc(&d_out, sizeof(h_out));
  for (int pat = 0; pat < 3; ++pat) {
    for (int l = 0; l < 64; ++l) {
      if (pat == 0) h_addr[l] = l * 8;                                   // linear: lane l -> elements 4l..4l+3
      else if (pat == 1) h_addr[l] = ((l & 15) >> 2) * 200 + (l & 3) * 8 + (l >> 4) * 1024;  // 4 rows (stride 100 el) x 16 cols per 16-lane group
      else h_addr[l] = (l & 15) * 136 + (l >> 4) * 8;                    // lane = row d (stride 68 el), group = 4-col block
    }
    hipMemcpy(d_addr, h_addr, sizeof(h_addr), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d_addr, d_out);
    hipMemcpy(h_out, d_out, sizeof(h_out), hipMemcpyDeviceToHost);
    printf("pattern %d (addr bytes: lane0=%d lane1=%d lane4=%d lane16=%d)\n", pat, h_addr[0], h_addr[1], h_addr[4], h_addr[16]);
    for (int l = 0; l < 64; ++l) {
      printf("  lane %2d addr_el %4d -> %4d %4d %4d %4d\n", l, h_addr[l] / 2, h_out[l * 4], h_out[l * 4 + 1], h_out[l * 4 + 2], h_out[l * 4 + 3]);
    }
  }
  return 0;
}
