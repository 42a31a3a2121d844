// Batched bf16 matrix transposition (64 x 64 tiles through LDS): the engine keeps TRANSPOSED copies of the encoder layers'
// projection weights so that the input-gradient products dX = dY W read their B operand K-contiguously -- the layout the
// four-wave GEMM (gemm_p4.hip) is built for -- instead of K-major through transposing LDS reads.  One launch per training
// forward (the optimizer rewrites the bf16 shadow between steps): ~310 MB of traffic for wav2vec2-base.
#include "common.h"
#include "kernels.h"

namespace {

struct TrJob {
  const bf16* src;  // [R][C], row stride ld_src
  bf16* dst;        // [C][R], contiguous
  int R, C, blk0, tiles_c;
};
constexpr int TR_CAP = 112;
struct TrJobs {
  int n;
  TrJob j[TR_CAP];
};
static_assert(sizeof(TrJobs) <= 4096, "kernel arguments are limited to 4 KiB");

__global__ __launch_bounds__(256) void transpose_bf16_kernel(const TrJobs jobs) {
  constexpr int PITCH = 66;  // elements: odd dword pitch, column reads spread over the banks
  __shared__ bf16 tile[64 * PITCH];
  int ji = 0;
  for (int k = 1; k < jobs.n; ++k) ji = (int)blockIdx.x >= jobs.j[k].blk0 ? k : ji;
  const TrJob& J = jobs.j[ji];
  const int blk = (int)blockIdx.x - J.blk0;
  const int r0 = blk / J.tiles_c * 64, c0 = blk % J.tiles_c * 64;
  const bool fast = ((J.C | J.R) & 7) == 0 && (((uintptr_t)J.src | (uintptr_t)J.dst) & 15) == 0;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int r = (int)threadIdx.x / 8 + 32 * i, c8 = ((int)threadIdx.x & 7) * 8;
    bf16x8 v;
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = (bf16)0.f;
    if (r0 + r < J.R) {
      const bf16* s = J.src + (long)(r0 + r) * J.C + c0 + c8;
      if (fast && c0 + c8 + 8 <= J.C) {
        v = *reinterpret_cast<const bf16x8*>(s);
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e)
          if (c0 + c8 + e < J.C) v[e] = s[e];
      }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) tile[r * PITCH + c8 + e] = v[e];
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int c = (int)threadIdx.x / 8 + 32 * i, r8 = ((int)threadIdx.x & 7) * 8;  // output row c, 8 consecutive source rows
    if (c0 + c >= J.C) continue;
    bf16x8 v;
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = tile[(r8 + e) * PITCH + c];
    bf16* d = J.dst + (long)(c0 + c) * J.R + r0 + r8;
    if (fast && r0 + r8 + 8 <= J.R) {
      *reinterpret_cast<bf16x8*>(d) = v;
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e)
        if (r0 + r8 + e < J.R) d[e] = v[e];
    }
  }
}

}  // namespace

int k_transpose_bf16_batched(int n, const bf16* const* src, bf16* const* dst, const int* R, const int* C, hipStream_t st) {
  for (int i0 = 0; i0 < n; i0 += TR_CAP) {
    TrJobs jobs;
    jobs.n = n - i0 < TR_CAP ? n - i0 : TR_CAP;
    int blk = 0;
    for (int i = 0; i < jobs.n; ++i) {
      TrJob& J = jobs.j[i];
      SSAK_REQUIRE(src[i0 + i] && dst[i0 + i] && R[i0 + i] > 0 && C[i0 + i] > 0, "transpose: bad matrix %d", i0 + i);
      J.src = src[i0 + i];
      J.dst = dst[i0 + i];
      J.R = R[i0 + i];
      J.C = C[i0 + i];
      J.blk0 = blk;
      J.tiles_c = ssak_cdiv(J.C, 64);
      blk += ssak_cdiv(J.R, 64) * J.tiles_c;
    }
    for (int i = jobs.n; i < TR_CAP; ++i) jobs.j[i] = jobs.j[0];
    transpose_bf16_kernel<<<blk, 256, 0, st>>>(jobs);
    SSAK_LAUNCH_CHECK();
  }
  return SSAK_OK;
}
