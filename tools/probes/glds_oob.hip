// Probe: what does an out-of-range `buffer_load ... lds` (LDS-DMA) write into LDS -- zeros or nothing?
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((address_space(3))) void lds_void;
__global__ void k(const char* g, unsigned* out, int nbytes) {
  __shared__ __attribute__((aligned(16))) unsigned smem[256];
  for (int i = threadIdx.x; i < 256; i += 64) smem[i] = 0xDEADBEEFu;
  __syncthreads();
  __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)g, 0, nbytes, 0x00020000);
  // lanes 0..31 in range, lanes 32..63 out of range (offset forced beyond num_records)
  unsigned off = threadIdx.x < 32 ? threadIdx.x * 16 : 0x80000000u;
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void*)smem, 16, off, 0, 0, 0);
  __builtin_amdgcn_s_waitcnt(0);
  __syncthreads();
  for (int i = threadIdx.x; i < 256; i += 64) out[i] = smem[i];
}
int main() {
  char* g; unsigned* o;
  hipMalloc(&g, 4096); hipMalloc(&o, 1024);
  unsigned h[1024]; for (int i = 0; i < 1024; ++i) h[i] = 0x11110000u + i;
  hipMemcpy(g, h, 4096, hipMemcpyHostToDevice);
  k<<<1, 64>>>(g, o, 4096);
  unsigned r[256]; hipMemcpy(r, o, 1024, hipMemcpyDeviceToHost);
  printf("in-range lane0 word0 %08x lane31 word3 %08x | OOB lane32 word0 %08x lane63 word3 %08x\n", r[0], r[31*4+3], r[32*4], r[63*4+3]);
  return 0;
}
