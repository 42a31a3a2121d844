// Development probe: per-CU bandwidth of an L2-resident stream into (a) LDS by LDS-DMA (buffer_load_dwordx4 ... lds), (b) VGPRs by
// buffer_load_dwordx4, (c) both at once (half the bytes each) -- is the LDS-DMA path of the GEMM kernels (21 B/clk/CU measured in
// situ) a limit of the DMA path itself or of the CU's memory pipeline as a whole?
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/probes/l2_bw.hip -o tools/probes/l2_bw.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((ext_vector_type(4))) float f32x4;

template <int MODE>
__global__ __launch_bounds__(512) void stream_kernel(const char* __restrict__ src, long bytes_per_wg, int iters, float* __restrict__ sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(src + (long)blockIdx.x * bytes_per_wg), 0, (int)bytes_per_wg, 0x00020000);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  // each iteration: the workgroup pulls 64 KB (8 waves x 8 x 1 KB)
  for (int it = 0; it < iters; ++it) {
    const uint32_t base = (uint32_t)(((long)it * 65536) % bytes_per_wg);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const uint32_t o = base + (uint32_t)((j * 8 + wave) * 1024 + lane * 16);
      const bool dma = MODE == 0 || (MODE == 2 && (j & 1));
      if (dma) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void*)(smem + ((it & 1) * 65536) + (j * 8 + wave) * 1024), 16, o, 0, 0, 0);
      } else {
        const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, o, 0, 0));
        acc += v;
      }
    }
    if ((it & 3) == 3) __builtin_amdgcn_s_waitcnt(0x0f70 | 8 | (0 << 14));  // vmcnt(8): keep a few in flight
  }
  __builtin_amdgcn_s_waitcnt(0x0f70);
  __syncthreads();
  if (acc[0] + acc[1] + acc[2] + acc[3] == 123.456f || smem[threadIdx.x] == 77) sink[0] = 1.f;
}

template <int MODE>
void run(const char* name, const char* src, long per_wg, float* sink) {
  const int iters = 2000, nwg = 256;
  hipFuncSetAttribute((const void*)stream_kernel<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  stream_kernel<MODE><<<nwg, 512, 131072>>>(src, per_wg, 50, sink);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  stream_kernel<MODE><<<nwg, 512, 131072>>>(src, per_wg, iters, sink);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double bytes = (double)nwg * iters * 65536;
  printf("%-28s working set/WG %6ld KB  %8.1f us  %7.2f TB/s  %6.1f B/ns/CU\n", name, per_wg >> 10, ms * 1e3, bytes / ms / 1e9, bytes / nwg / (ms * 1e6));
}

int main() {
  char* src; float* sink;
  const long total = 256L << 20;
  hipMalloc(&src, total); hipMemset(src, 1, total); hipMalloc(&sink, 4);
  for (long per_wg : {65536L, 131072L, 1L << 20}) {  // 64 KB / 128 KB per workgroup: L2-resident (16 / 32 MB over 8 XCD L2s of 4 MB: 2-4 MB each); 1 MB: Infinity Cache
    run<0>("LDS-DMA dwordx4", src, per_wg, sink);
    run<1>("buffer_load_dwordx4 -> VGPR", src, per_wg, sink);
    run<2>("half DMA, half VGPR", src, per_wg, sink);
  }
  return 0;
}
