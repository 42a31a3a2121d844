// Development probe: main-loop rate of a FOUR-wave 256x256-tile bf16 GEMM with 128x128 register tiles per wave
// (0.25 ds_read_b128 per MFMA instead of the 0.375 of the 8-wave kernel, whose main loop is LDS-bandwidth co-limited).
// One wave per SIMD: fragment reads of K step t+1 are interleaved with the MFMAs of step t inside the wave; 32-deep K
// steps, four LDS stages of 32 KB, LDS-DMA issued three steps ahead, one barrier per step.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -Issak_amd/csrc tools/probes/p4_probe.hip -o tools/probes/p4_probe.bin
// -DP4_FLAGS: no workgroup barrier in the loop -- per-stage LDS counters instead ("my share of stage s has landed" / "I have
// read stage s"), so the four waves drift within the slack of the four-stage pipeline instead of meeting every K step.
#include "../../ssak_amd/csrc/common.h"
#include "../../ssak_amd/csrc/gemm_common.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

void ssak_set_error(const char*, ...) {}

namespace {

constexpr int P4_BK = 32;
constexpr int P4_STAGE = 32768;  // A 16 KB | B 16 KB
constexpr int P4_NS = 4;
#ifdef P4_NO_DMA
#define P4_SKIP_DUMMY 1
#else
#define P4_SKIP_DUMMY 0
#endif

// LDS image of an operand stage: [256 rows][64 B]; 16-B chunk c of row r at byte r*64 + ((c ^ ((r >> 2) & 3)) << 4)
struct P4Stager {
  __amdgpu_buffer_rsrc_t rsrc;
  uint32_t off[4];  // this lane's source byte offset for its 4 DMA instructions of the current step
  int wave;
  __device__ __forceinline__ void init(const bf16* base, long ld, int row0, uint32_t extent) {
    rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)extent, 0x00020000);
    wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int o = ((j * 4 + wave) * 64 + lane) * 16;  // LDS byte offset inside the 16 KB operand stage
      const int r = o >> 6, pc = (o >> 4) & 3;
      const int c = pc ^ ((r >> 2) & 3);
      off[j] = (uint32_t)(((long)(row0 + r) * ld + c * 8) * 2);
    }
  }
  __device__ __forceinline__ void issue(char* lds_op) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void*)(lds_op + (j * 4 + wave) * 1024), 16, off[j], 0, 0, 0);
      off[j] += P4_BK * 2;
    }
  }
};

__global__ __launch_bounds__(256) void gemm_p4_probe(const bf16* A, const bf16* B, bf16* C, int M, int N, int K) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wr = wave >> 1, wc = wave & 1;
  const int tiles_n = N / 256;
  const int id = xcd_remap(blockIdx.x, gridDim.x);
  const int bm0 = (id / tiles_n) * 256, bn0 = (id % tiles_n) * 256;
  const int nk = K / P4_BK;
  P4Stager sa, sb;
  sa.init(A, K, bm0, (uint32_t)((size_t)M * K * 2));
  sb.init(B, K, bn0, (uint32_t)((size_t)N * K * 2));
  f32x4 acc[8][8];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  // fragment offsets: row (16 i + lm) of this wave's panel, chunk lq
  const int lm = lane & 15, lq = lane >> 4;
  int fo_a[8], fo_b[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int ra = wr * 128 + 16 * i + lm, rb = wc * 128 + 16 * i + lm;
    fo_a[i] = ra * 64 + ((lq ^ ((ra >> 2) & 3)) << 4);
    fo_b[i] = 16384 + rb * 64 + ((lq ^ ((rb >> 2) & 3)) << 4);
  }
#ifdef P4_FLAGS
  typedef __attribute__((address_space(3))) int lds_int;
  int* const flags = reinterpret_cast<int*>(smem + P4_NS * P4_STAGE);  // [0..3] landed counters, [4..7] read counters
  if (threadIdx.x < 8) flags[threadIdx.x] = 0;
  __syncthreads();
#endif
  // prologue: stages 0..3 in flight
#pragma unroll
  for (int s = 0; s < P4_NS; ++s) {
    sa.issue(smem + s * P4_STAGE);
    sb.issue(smem + s * P4_STAGE + 16384);
  }
  wait_vmcnt<24>();  // stage 0 landed (this wave's share)
  __syncthreads();
#ifdef P4_FLAGS
  // stage 0's landing was settled by the barrier above: start its counter at 4 (one generation complete)
  if (threadIdx.x == 0) flags[0] = 4;
  __syncthreads();
#endif
  bf16x8 fa[2][8], fb[2][8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    fa[0][i] = *reinterpret_cast<const bf16x8*>(smem + fo_a[i]);
    fb[0][i] = *reinterpret_cast<const bf16x8*>(smem + fo_b[i]);
  }
  for (int t = 0; t < nk; t += 2) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int tt = t + u;
      // stage tt+1 landed everywhere; everyone is done reading stage tt (its fragments are in registers)
#ifndef P4_NO_WAIT
      wait_vmcnt<16>();
#endif
#ifdef P4_FLAGS
      {
        // this wave's share of stage tt+1 has landed; this wave finished reading stage tt one step ago (fragments in registers)
        const int s1 = (tt + 1) & 3, s0 = tt & 3;
        __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): the fragment reads of stage tt are in registers
        if (lane == 0) {
          __hip_atomic_fetch_add(flags + s1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          __hip_atomic_fetch_add(flags + 4 + s0, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        const int want_landed = 4 * (((tt + 1) >> 2) + 1), want_read = 4 * ((tt >> 2) + 1);
        int guard = 0;  // (a probe must terminate even if the protocol is wrong)
        while ((__hip_atomic_load(flags + s1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < want_landed ||
                __hip_atomic_load(flags + 4 + s0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < want_read) &&
               ++guard < 200000)
          __builtin_amdgcn_s_sleep(1);
        __builtin_amdgcn_sched_barrier(0);
      }
#elif !defined(P4_NO_BARRIER)
      __syncthreads();
#endif
      char* const nxt = smem + ((tt + 1) & 3) * P4_STAGE;
      char* const freed = smem + (tt & 3) * P4_STAGE;
#ifdef P4_NO_DMA
      if (false) {
#else
      if (tt + 4 < nk) {
#endif
        sa.issue(freed);
        sb.issue(freed + 16384);
      } else if (!P4_SKIP_DUMMY) {  // keep the vmcnt arithmetic uniform past the end: out-of-range dummies
#pragma unroll
        for (int j = 0; j < 8; ++j)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(sa.rsrc, (lds_void*)(freed + (j * 4 + wave) * 1024), 16, 0x80000000u, 0, 0, 0);
      }
      // fragments of step tt+1 while the MFMAs of step tt run
#ifndef P4_NO_DS
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        fa[u ^ 1][i] = *reinterpret_cast<const bf16x8*>(nxt + fo_a[i]);
        fb[u ^ 1][i] = *reinterpret_cast<const bf16x8*>(nxt + fo_b[i]);
      }
#else
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        fa[u ^ 1][i] = fa[u][i];
        fb[u ^ 1][i] = fb[u][i];
      }
#endif
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[u][j], fa[u][i], acc[i][j], 0, 0, 0);
      // interleave: 4 MFMA then 1 LDS read, 16 times
#pragma unroll
      for (int g = 0; g < 16; ++g) {
        __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      }
    }
  }
  wait_vmcnt<0>();
  // epilogue (timing only): lane (lm, lq) of block (i, j) holds row 16 i + lm, columns 16 j + 4 lq ..
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const bf16x4 o = {(bf16)acc[i][j][0], (bf16)acc[i][j][1], (bf16)acc[i][j][2], (bf16)acc[i][j][3]};
      *reinterpret_cast<bf16x4*>(C + (long)(bm0 + wr * 128 + 16 * i + lm) * N + bn0 + wc * 128 + 16 * j + 4 * lq) = o;
    }
}

}  // namespace

int main(int argc, char** argv) {
  const int M = argc > 1 ? atoi(argv[1]) : 4096, N = argc > 2 ? atoi(argv[2]) : 4096, K = argc > 3 ? atoi(argv[3]) : 4096;
  bf16 *A, *B, *C;
  hipMalloc(&A, (size_t)M * K * 2);
  hipMalloc(&B, (size_t)N * K * 2);
  hipMalloc(&C, (size_t)M * N * 2);
  std::vector<unsigned short> ha((size_t)M * K), hb((size_t)N * K);
  srand(1);
  auto tobf = [](float f) { union { float f; unsigned u; } c; c.f = f; return (unsigned short)(c.u >> 16); };
  for (auto& v : ha) v = tobf((float)(rand() % 7 - 3));
  for (auto& v : hb) v = tobf((float)(rand() % 7 - 3));
  hipMemcpy(A, ha.data(), ha.size() * 2, hipMemcpyHostToDevice);
  hipMemcpy(B, hb.data(), hb.size() * 2, hipMemcpyHostToDevice);
  hipFuncSetAttribute((const void*)gemm_p4_probe, hipFuncAttributeMaxDynamicSharedMemorySize, P4_NS * P4_STAGE + 64);
  const int grid = (M / 256) * (N / 256);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  float best = 1e9;
  for (int it = 0; it < 6; ++it) {
    hipEventRecord(e0);
    gemm_p4_probe<<<grid, 256, P4_NS * P4_STAGE + 64>>>(A, B, C, M, N, K);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    best = ms < best ? ms : best;
  }
  printf("p4 probe M=%d N=%d K=%d: %.1f us  %.1f TF/s\n", M, N, K, best * 1e3, 2.0 * M * N * K / (best * 1e-3) / 1e12);
  // spot check a few entries (integer operands: exact)
  std::vector<unsigned short> hc((size_t)M * N);
  hipMemcpy(hc.data(), C, hc.size() * 2, hipMemcpyDeviceToHost);
  auto tof = [](unsigned short h) { union { float f; unsigned u; } c; c.u = (unsigned)h << 16; return c.f; };
  int bad = 0;
  for (int s = 0; s < 200; ++s) {
    const int m = rand() % M, n = rand() % N;
    double ref = 0;
    for (int k = 0; k < K; ++k) ref += (double)tof(ha[(size_t)m * K + k]) * tof(hb[(size_t)n * K + k]);
    const float got = tof(hc[(size_t)m * N + n]);
    if (fabs(got - ref) > 0.01 * fabs(ref) + 2.0) ++bad;
  }
  printf("spot check: %d / 200 mismatches\n", bad);
  return 0;
}
