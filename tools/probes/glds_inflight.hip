// How the global -> LDS stream rate of a CU depends on the bytes it keeps in flight (gfx950).  One 256-thread workgroup per CU,
// every wave issues 1 KiB LDS-DMA instructions (buffer_load_dwordx4 ... lds) into its own ring of DEPTH slots and waits with a
// counted vmcnt so that DEPTH instructions stay outstanding: 4 x DEPTH KiB in flight per CU.  Sources: a 2 GiB buffer swept once
// (HBM) and a 2 MiB region per XCD read over and over (L2).
// Build: hipcc --offload-arch=gfx950 -O3 tools/probes/glds_inflight.hip -o glds_inflight
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef __attribute__((address_space(3))) void lds_void;

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  __builtin_amdgcn_s_waitcnt((N & 15) | (7 << 4) | (15 << 8) | ((N >> 4) << 14));
}

template <int DEPTH>
__global__ __launch_bounds__(256) void stream_kernel(const char* __restrict__ src, size_t region, size_t wg_stride, int n, unsigned* sink, int share, unsigned stagger) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  // share > 1: `share` workgroups of one XCD (ids b, b + 8, ...: consecutive ids go round the 8 XCDs) read the SAME region in
  // lockstep -- the operand panels of GEMM tiles
  const int xcd = blockIdx.x & 7, xslot = blockIdx.x >> 3;
  const size_t rid = share > 1 ? (size_t)(xslot / share) * 8 + xcd : blockIdx.x;
  const char* base = src + rid * wg_stride;
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)region, 0x00020000);
  char* ring = lds + wave * DEPTH * 1024;
  // instruction i of this wave reads bytes [(4 i + wave) KiB, + 1 KiB) of the workgroup's region (wrapping)
  const unsigned span = (unsigned)region;
  // stagger > 0: sharer j starts j * stagger bytes into the region (and wraps): the sharers are never on the same lines at once
  unsigned off = wave * 1024 + lane * 16 + (share > 1 ? (unsigned)((blockIdx.x >> 3) % share) * stagger : 0u);
  for (int i = 0; i < DEPTH; ++i) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void*)(ring + i * 1024), 16, off, 0, 0, 0);
    off += 4096;
    if (off >= span) off -= span;
  }
  int slot = 0;
  for (int i = DEPTH; i < n; ++i) {
    wait_vmcnt<DEPTH - 1>();
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void*)(ring + slot * 1024), 16, off, 0, 0, 0);
    off += 4096;
    if (off >= span) off -= span;
    slot = slot + 1 == DEPTH ? 0 : slot + 1;
  }
  wait_vmcnt<0>();
  __syncthreads();
  if (sink && threadIdx.x == 0) sink[blockIdx.x] = *reinterpret_cast<unsigned*>(lds + 64);
}

template <int DEPTH>
double run(const char* src, size_t region, size_t wg_stride, int n, int grid, unsigned* sink, int share = 1, unsigned stagger = 0) {
  hipFuncSetAttribute((const void*)stream_kernel<DEPTH>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * DEPTH * 1024);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  stream_kernel<DEPTH><<<grid, 256, 4 * DEPTH * 1024>>>(src, region, wg_stride, n, sink, share, stagger);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  stream_kernel<DEPTH><<<grid, 256, 4 * DEPTH * 1024>>>(src, region, wg_stride, n, sink, share, stagger);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  return (double)grid * n * 4 * 1024 / (ms * 1e-3) / 1e9;  // GB/s
}

int main(int argc, char** argv) {
  const bool one = argc > 1;  // any argument: the 64 KiB row only (for a counter run: rocprofv3 --pmc FETCH_SIZE)
  const int grid = 256;
  const size_t total = 2ull << 30;
  char* buf;
  unsigned* sink;
  if (hipMalloc(&buf, total) != hipSuccess || hipMalloc(&sink, 4096) != hipSuccess) return 1;
  hipMemset(buf, 1, total);
  printf("KiB in flight per CU | HBM sweep (8 MiB per workgroup, 2 GiB) | L2 (2 MiB shared by an XCD's workgroups) | mixed 3 L2 : 1 HBM not measured\n");
#define ROW(D)                                                                                                       \
  {                                                                                                                  \
    const double hbm = run<D>(buf, 8u << 20, 8u << 20, 2048, grid, sink); /* 8 MiB per workgroup, read once */       \
    const double l2 = run<D>(buf, 2u << 20, 0, 8192, grid, sink);         /* everyone reads the same 2 MiB */         \
    const double s4 = run<D>(buf, 24u << 20, 24u << 20, 6144, grid, sink, 4); /* 4 sharers, 24 MiB each group: 1.5 GiB unique */ \
    const double s8 = run<D>(buf, 48u << 20, 48u << 20, 12288, grid, sink, 8); /* 8 sharers, 48 MiB each group */               \
    const double g4 = run<D>(buf, 24u << 20, 24u << 20, 6144, grid, sink, 4, 256u << 10); /* 4 sharers, 256 KiB apart */       \
    const double h4 = run<D>(buf, 24u << 20, 24u << 20, 6144, grid, sink, 4, 6u << 20);   /* 4 sharers, 6 MiB apart */          \
    printf("      4 sharers 256 KiB apart %8.1f (%5.1f per CU)   6 MiB apart (a quarter of the region) %8.1f (%5.1f per CU)\n", g4, g4 / grid, h4, h4 / grid); \
    printf("%4d  HBM %8.1f GB/s (%5.1f per CU)   L2 %8.1f (%5.1f)   4 sharers in lockstep %8.1f (%5.1f per CU; HBM side %6.1f)   8 sharers %8.1f (%5.1f; HBM side %6.1f)\n", 4 * D, hbm, hbm / grid, l2, l2 / grid, s4, s4 / grid, s4 / 4, s8, s8 / grid, s8 / 8); \
  }
  if (one) {
    ROW(16)
    return 0;
  }
  ROW(4) ROW(8) ROW(12) ROW(16) ROW(20) ROW(24) ROW(32) ROW(36)
  return 0;
}
