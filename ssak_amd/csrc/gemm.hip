// Batched bf16 MFMA GEMM with fp32 accumulation: the one dense-contraction kernel behind every Linear,
// channels-last Conv1d (overlapping-row "Toeplitz" A operand), attention product and weight gradient of
// the Wav2Vec2 / Whisper encoder path.  gfx950 only.
//
//   C[z][m][n] = alpha * sum_k A(z; m,k) * B(z; n,k)   (+ bias[n], GELU, ...)
//
// Operand storage is chosen per operand: K-contiguous ([rows][K], an nn.Linear weight or a row-major
// activation) or K-major ([K][rows], what a transposed product -- dW = dY^T X, P^T dO -- reads).  Both are
// staged global -> VGPR -> LDS (16-byte chunks, zero-filled at every edge) and double-buffered with one
// workgroup barrier per 64-deep K step; K-contiguous tiles are XOR-swizzled and read with ds_read_b128,
// K-major tiles are row-padded and read with ds_read_b64_tr_b16 (the hardware transpose read), so no
// operand is ever transposed in HBM.  4 waves per workgroup, each owning a (BM/WM)x(BN/WN) sub-tile of
// 16x16x32 bf16 MFMAs.  The MFMA is issued with the operands swapped so each lane ends up with 4
// consecutive output columns (8-/16-byte stores).  Workgroup ids are remapped so that each XCD's L2 sees a
// contiguous run of tiles sharing the same A row panel.
#include <vector>

#include "common.h"

namespace {

constexpr int BK = 64;
constexpr int NTHREADS = 256;

struct GemmParams {
  const bf16* A;
  const bf16* B;
  void* C;
  const float* bias;
  const bf16* aux_in;
  bf16* aux_out;
  float* slab;
  int M, N, K;
  long lda, ldb, ldc;
  int nb2;
  long sa1, sa2, sb1, sb2, sc1, sc2;
  float alpha;
  int epilogue, out_f32, accumulate, split_k;
  int tiles_m, tiles_n, nz, kt_per_split;
  uint32_t drop_thresh, drop_stream;
  float drop_scale;
  uint64_t drop_seed;
  long bias_s2;
};

// ---- LDS tile geometry -------------------------------------------------------------------------
// K-contiguous tile: R rows x 128 B, 16-B chunk c of row r stored at chunk c ^ ((r >> 1) & 7)
//   (16 consecutive rows at one logical chunk land on 16 distinct 16-B slots of the 256-B bank row).
// K-major tile: 64 k-rows x (2R + 32) B; the 32-B pad makes 8 consecutive k-rows cover all 64 banks.
template <int R, bool KM>
struct Tile {
  static constexpr int STRIDE = KM ? (2 * R + 32) : 128;
  static constexpr int BYTES = KM ? BK * STRIDE : R * 128;
  static constexpr int NCHUNK = R * 8 / NTHREADS;  // 16-B chunks per thread per tile
};

__device__ __forceinline__ uint4 mask_chunk(uint4 v, int nvalid) {
  // keep the first nvalid (0..8) bf16 of a 16-B chunk
  uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = nvalid - 2 * i;
    w[i] = (r >= 2) ? w[i] : (r == 1 ? (w[i] & 0xffffu) : 0u);
  }
  return make_uint4(w[0], w[1], w[2], w[3]);
}

template <int N>
struct Regs {
  uint4 v[N];
};

template <int R, bool KM>
__device__ __forceinline__ void load_tile(Regs<Tile<R, KM>::NCHUNK>& regs, const bf16* __restrict__ base, long ld,
                                          int row0, int rows_total, int k0, int k_end) {
  // base already includes the batch offset.  rows_total / k_end bound the valid region.
#pragma unroll
  for (int i = 0; i < Tile<R, KM>::NCHUNK; ++i) {
    const int q = threadIdx.x + i * NTHREADS;
    uint4 v = make_uint4(0u, 0u, 0u, 0u);
    if (!KM) {
      const int r = q >> 3, c = q & 7;
      const int gr = row0 + r, gk = k0 + c * 8;
      if (gr < rows_total && gk < k_end) {
        v = *reinterpret_cast<const uint4*>(base + (long)gr * ld + gk);
        if (gk + 8 > k_end) v = mask_chunk(v, k_end - gk);
      }
    } else {
      constexpr int CPR = R / 8;  // chunks per k-row
      const int kr = q / CPR, c = q % CPR;
      const int gk = k0 + kr, gr = row0 + c * 8;
      if (gk < k_end && gr < rows_total) {
        v = *reinterpret_cast<const uint4*>(base + (long)gk * ld + gr);
        if (gr + 8 > rows_total) v = mask_chunk(v, rows_total - gr);
      }
    }
    regs.v[i] = v;
  }
}

template <int R, bool KM>
__device__ __forceinline__ void store_tile(const Regs<Tile<R, KM>::NCHUNK>& regs, char* lds) {
#pragma unroll
  for (int i = 0; i < Tile<R, KM>::NCHUNK; ++i) {
    const int q = threadIdx.x + i * NTHREADS;
    int off;
    if (!KM) {
      const int r = q >> 3, c = q & 7;
      off = r * 128 + ((c ^ ((r >> 1) & 7)) << 4);
    } else {
      constexpr int CPR = R / 8;
      const int kr = q / CPR, c = q % CPR;
      off = kr * Tile<R, KM>::STRIDE + c * 16;
    }
    *reinterpret_cast<uint4*>(lds + off) = regs.v[i];
  }
}

// fragment for rows [r0, r0+16) and k in [32*kk, 32*kk+32): lane l gets k = 8*(l>>4)+j, row r0 + (l&15)
template <int R, bool KM>
__device__ __forceinline__ bf16x8 read_frag(const char* lds, int r0, int kk, int lane) {
  if (!KM) {
    const int r = r0 + (lane & 15);
    const int c = kk * 4 + (lane >> 4);
    return *reinterpret_cast<const bf16x8*>(lds + r * 128 + ((c ^ ((r >> 1) & 7)) << 4));
  } else {
    const int g = lane >> 4, q = (lane >> 2) & 3, p = lane & 3;
    const int off = (kk * 32 + 8 * g + q) * Tile<R, KM>::STRIDE + (r0 + 4 * p) * 2;
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(lds + off));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(lds + off + 4 * Tile<R, KM>::STRIDE));
    typedef __attribute__((ext_vector_type(8))) short s16x8;
    s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8, v);
  }
}

__device__ __forceinline__ int xcd_remap(int id, int n) {
  // contiguous run of logical ids per XCD (hardware deals consecutive workgroup ids round-robin over 8 XCDs);
  // bijective for any n.  Speed only.
  const int q = n >> 3, r = n & 7;
  const int x = id & 7, i = id >> 3;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
}

template <int BM, int BN, int WM, int WN, bool A_KM, bool B_KM>
__global__ __launch_bounds__(NTHREADS) void gemm_kernel(const GemmParams p) {
  static_assert(WM * WN == 4, "4 waves");
  constexpr int TM = BM / WM, TN = BN / WN;
  constexpr int MI = TM / 16, NI = TN / 16;
  using TA = Tile<BM, A_KM>;
  using TB = Tile<BN, B_KM>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int STAGE = TA::BYTES + TB::BYTES;  // stage s: A at s*STAGE, B at s*STAGE + TA::BYTES

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm0 = (wave / WN) * TM, wn0 = (wave % WN) * TN;

  const int per_z = p.tiles_m * p.tiles_n;
  const int id = xcd_remap(blockIdx.x, gridDim.x);
  const int zs = id / per_z, rem = id % per_z;
  const int tm = rem / p.tiles_n, tn = rem % p.tiles_n;
  const int split = zs % p.split_k, z = zs / p.split_k;
  const int z1 = z / p.nb2, z2 = z % p.nb2;
  const bf16* Ab = p.A + z1 * p.sa1 + z2 * p.sa2;
  const bf16* Bb = p.B + z1 * p.sb1 + z2 * p.sb2;
  const int bm0 = tm * BM, bn0 = tn * BN;

  const int nkt = (p.K + BK - 1) / BK;
  const int kt0 = split * p.kt_per_split;
  const int kt1 = min(nkt, kt0 + p.kt_per_split);

  f32x4 acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  Regs<TA::NCHUNK> ra;
  Regs<TB::NCHUNK> rb;
  if (kt0 < kt1) {
    load_tile<BM, A_KM>(ra, Ab, p.lda, bm0, p.M, kt0 * BK, p.K);
    load_tile<BN, B_KM>(rb, Bb, p.ldb, bn0, p.N, kt0 * BK, p.K);
    store_tile<BM, A_KM>(ra, smem);
    store_tile<BN, B_KM>(rb, smem + TA::BYTES);
  }
  __syncthreads();
  for (int kt = kt0; kt < kt1; ++kt) {
    const int cur = (kt - kt0) & 1;
    const bool more = kt + 1 < kt1;
    if (more) {
      load_tile<BM, A_KM>(ra, Ab, p.lda, bm0, p.M, (kt + 1) * BK, p.K);
      load_tile<BN, B_KM>(rb, Bb, p.ldb, bn0, p.N, (kt + 1) * BK, p.K);
    }
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      bf16x8 fa[MI], fb[NI];
#pragma unroll
      for (int i = 0; i < MI; ++i) fa[i] = read_frag<BM, A_KM>(smem + cur * STAGE, wm0 + 16 * i, kk, lane);
#pragma unroll
      for (int j = 0; j < NI; ++j) fb[j] = read_frag<BN, B_KM>(smem + cur * STAGE + TA::BYTES, wn0 + 16 * j, kk, lane);
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);
    }
    if (more) {
      store_tile<BM, A_KM>(ra, smem + (cur ^ 1) * STAGE);
      store_tile<BN, B_KM>(rb, smem + (cur ^ 1) * STAGE + TA::BYTES);
    }
    __syncthreads();
  }

  // ---- epilogue: lane holds m = .. + (lane & 15), n = .. + 4 * (lane >> 4) + r, r = 0..3
  const int lm = lane & 15, ln = (lane >> 4) * 4;
  if (p.split_k > 1) {
    float* S = p.slab + ((long)split * p.nz + z) * (long)p.M * p.N;
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      const int m = bm0 + wm0 + 16 * i + lm;
      if (m >= p.M) continue;
#pragma unroll
      for (int j = 0; j < NI; ++j) {
        const int n = bn0 + wn0 + 16 * j + ln;
        float* dst = S + (long)m * p.N + n;
        if (n + 3 < p.N && (p.N & 3) == 0) {
          *reinterpret_cast<f32x4*>(dst) = acc[i][j];
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (n + r < p.N) dst[r] = acc[i][j][r];
        }
      }
    }
    return;
  }
  const long coff = z1 * p.sc1 + z2 * p.sc2;
#pragma unroll
  for (int j = 0; j < NI; ++j) {
    const int n = bn0 + wn0 + 16 * j + ln;
    if (n >= p.N) continue;
    const bool full = n + 3 < p.N;
    float bv[4] = {0.f, 0.f, 0.f, 0.f};
    if (p.bias) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (n + r < p.N) bv[r] = p.bias[z2 * p.bias_s2 + n + r];
    }
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      const int m = bm0 + wm0 + 16 * i + lm;
      if (m >= p.M) continue;
      const long o = coff + (long)m * p.ldc + n;
      float v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = acc[i][j][r] * p.alpha + bv[r];
      if (p.epilogue == SSAK_EPI_GELU) {
        if (p.aux_out) {
          if (full) {
            bf16x4 t = {(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]};
            *reinterpret_cast<bf16x4*>(p.aux_out + o) = t;
          } else {
#pragma unroll
            for (int r = 0; r < 4; ++r)
              if (n + r < p.N) p.aux_out[o + r] = (bf16)v[r];
          }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = gelu_f(v[r]);
      } else if (p.epilogue == SSAK_EPI_MUL_GELU_GRAD) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (n + r < p.N) v[r] *= gelu_grad_f((float)p.aux_in[o + r]);
      }
      if (p.drop_thresh) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          v[r] = keep_bit(p.drop_seed, p.drop_stream, (uint64_t)(o + r), p.drop_thresh) ? v[r] * p.drop_scale : 0.f;
      }
      if (p.out_f32) {
        float* dst = reinterpret_cast<float*>(p.C) + o;
        if (p.accumulate) {
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (n + r < p.N) dst[r] += v[r];
        } else if (full) {
          *reinterpret_cast<f32x4*>(dst) = (f32x4){v[0], v[1], v[2], v[3]};
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (n + r < p.N) dst[r] = v[r];
        }
      } else {
        bf16* dst = reinterpret_cast<bf16*>(p.C) + o;
        if (full) {
          bf16x4 t = {(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]};
          *reinterpret_cast<bf16x4*>(dst) = t;
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (n + r < p.N) dst[r] = (bf16)v[r];
        }
      }
    }
  }
}

// deterministic split-K combine: C = alpha * sum_s slab[s] (+ bias) (+ C)
__global__ void splitk_reduce_kernel(const GemmParams p) {
  const long per = (long)p.M * p.N;
  const long total = per * p.nz;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const int z = (int)(e / per);
    const long r = e % per;
    const int m = (int)(r / p.N), n = (int)(r % p.N);
    float s = 0.f;
    for (int k = 0; k < p.split_k; ++k) s += p.slab[((long)k * p.nz + z) * per + r];
    s = s * p.alpha + (p.bias ? p.bias[(z % p.nb2) * p.bias_s2 + n] : 0.f);
    const long o = (long)(z / p.nb2) * p.sc1 + (long)(z % p.nb2) * p.sc2 + (long)m * p.ldc + n;
    if (p.out_f32) {
      float* dst = reinterpret_cast<float*>(p.C) + o;
      *dst = p.accumulate ? (*dst + s) : s;
    } else {
      reinterpret_cast<bf16*>(p.C)[o] = (bf16)s;
    }
  }
}

// ---- optional per-launch timing (bench.py's roofline leg): HIP events around every GEMM launch ----
struct ProfRec {
  hipEvent_t e0, e1;
  int variant;
  double flops;
};
bool g_prof_on = false;
std::vector<ProfRec> g_prof;
std::vector<hipEvent_t> g_event_pool;
hipEvent_t prof_event() {
  if (!g_event_pool.empty()) {
    hipEvent_t e = g_event_pool.back();
    g_event_pool.pop_back();
    return e;
  }
  hipEvent_t e;
  (void)hipEventCreate(&e);
  return e;
}
const char* kVariantNames[8] = {
    "gemm_kernel<128, 128, 2, 2, false, false>", "gemm_kernel<128, 128, 2, 2, false, true>",
    "gemm_kernel<128, 128, 2, 2, true, false>",  "gemm_kernel<128, 128, 2, 2, true, true>",
    "gemm_kernel<128, 64, 2, 2, false, false>",  "gemm_kernel<128, 64, 2, 2, false, true>",
    "gemm_kernel<128, 64, 2, 2, true, false>",   "gemm_kernel<128, 64, 2, 2, true, true>"};

template <int BM, int BN, int WM, int WN, bool A_KM, bool B_KM>
int launch(const GemmParams& p, hipStream_t st) {
  constexpr size_t lds = 2 * (size_t)(Tile<BM, A_KM>::BYTES + Tile<BN, B_KM>::BYTES);
  auto kern = gemm_kernel<BM, BN, WM, WN, A_KM, B_KM>;
  if (lds > 64 * 1024) {
    static bool attr_done = false;  // per instantiation
    if (!attr_done) {
      SSAK_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      attr_done = true;
    }
  }
  const long nblk = (long)p.tiles_m * p.tiles_n * p.nz * p.split_k;
  ProfRec rec;
  if (g_prof_on) {
    rec.e0 = prof_event();
    rec.e1 = prof_event();
    rec.variant = (BN == 128 ? 0 : 4) + (A_KM ? 2 : 0) + (B_KM ? 1 : 0);
    rec.flops = 2.0 * p.M * p.N * (double)p.K * p.nz;
    (void)hipEventRecord(rec.e0, st);
  }
  kern<<<dim3((unsigned)nblk), NTHREADS, lds, st>>>(p);
  SSAK_LAUNCH_CHECK();
  if (g_prof_on) {
    (void)hipEventRecord(rec.e1, st);
    g_prof.push_back(rec);
  }
  return SSAK_OK;
}

template <int BM, int BN, int WM, int WN>
int dispatch_layout(const GemmParams& p, int a_km, int b_km, hipStream_t st) {
  if (!a_km && !b_km) return launch<BM, BN, WM, WN, false, false>(p, st);
  if (!a_km && b_km) return launch<BM, BN, WM, WN, false, true>(p, st);
  if (a_km && b_km) return launch<BM, BN, WM, WN, true, true>(p, st);
  return launch<BM, BN, WM, WN, true, false>(p, st);
}

}  // namespace

extern "C" int ssak_gemm_bf16(const ssak_gemm_desc* d, const void* A, const void* B, void* C, const float* bias,
                              const void* aux_in, void* aux_out, void* workspace, size_t workspace_bytes,
                              void* stream) {
  SSAK_REQUIRE(d && A && B && C, "gemm: null pointer");
  SSAK_REQUIRE(d->M > 0 && d->N > 0 && d->K > 0, "gemm: bad shape M=%d N=%d K=%d", d->M, d->N, d->K);
  SSAK_REQUIRE((d->lda & 7) == 0 && (d->ldb & 7) == 0 && (d->ldc & 3) == 0, "gemm: lda/ldb must be multiples of 8, ldc of 4");
  SSAK_REQUIRE(((uintptr_t)A & 15) == 0 && ((uintptr_t)B & 15) == 0 && ((uintptr_t)C & 15) == 0, "gemm: operands must be 16-byte aligned");
  SSAK_REQUIRE(((d->sa1 | d->sa2 | d->sb1 | d->sb2) & 7) == 0 && ((d->sc1 | d->sc2) & 3) == 0, "gemm: batch strides must keep 16-byte (A,B) / 8-byte (C) alignment");
  SSAK_REQUIRE(d->nb1 > 0 && d->nb2 > 0, "gemm: batch counts must be >= 1");
  SSAK_REQUIRE(d->epilogue >= 0 && d->epilogue <= 2, "gemm: bad epilogue %d", d->epilogue);
  SSAK_REQUIRE(d->epilogue != SSAK_EPI_MUL_GELU_GRAD || aux_in, "gemm: MUL_GELU_GRAD needs aux_in");
  SSAK_REQUIRE(!d->accumulate || d->out_f32, "gemm: accumulate needs fp32 output");
  const int split = d->split_k > 1 ? d->split_k : 1;
  SSAK_REQUIRE(split == 1 || d->epilogue == SSAK_EPI_NONE, "gemm: split_k supports the plain epilogue only");
  GemmParams p;
  p.A = (const bf16*)A;
  p.B = (const bf16*)B;
  p.C = C;
  p.bias = bias;
  p.aux_in = (const bf16*)aux_in;
  p.aux_out = (bf16*)aux_out;
  p.slab = (float*)workspace;
  p.M = d->M;
  p.N = d->N;
  p.K = d->K;
  p.lda = d->lda;
  p.ldb = d->ldb;
  p.ldc = d->ldc;
  p.nb2 = d->nb2;
  p.sa1 = d->sa1;
  p.sa2 = d->sa2;
  p.sb1 = d->sb1;
  p.sb2 = d->sb2;
  p.sc1 = d->sc1;
  p.sc2 = d->sc2;
  p.alpha = d->alpha;
  p.epilogue = d->epilogue;
  p.out_f32 = d->out_f32;
  p.accumulate = d->accumulate;
  p.split_k = split;
  p.nz = d->nb1 * d->nb2;
  p.drop_thresh = d->drop_p > 0.f ? (uint32_t)fminf(4294967295.f, d->drop_p * 4294967296.f) : 0u;
  p.drop_scale = d->drop_p > 0.f ? 1.f / (1.f - d->drop_p) : 1.f;
  p.drop_stream = d->drop_stream;
  p.drop_seed = d->drop_seed;
  p.bias_s2 = d->bias_s2;
  SSAK_REQUIRE(d->drop_p >= 0.f && d->drop_p < 1.f, "gemm: drop_p must be in [0,1)");
  SSAK_REQUIRE(!(d->drop_p > 0.f && d->split_k > 1), "gemm: dropout epilogue is not available with split_k");
  const int nkt = ssak_cdiv(d->K, BK);
  p.kt_per_split = ssak_cdiv(nkt, split);
  if (split > 1)
    SSAK_REQUIRE(workspace && workspace_bytes >= (size_t)split * p.nz * (size_t)d->M * d->N * sizeof(float),
                 "gemm: split_k workspace too small");
  hipStream_t st = (hipStream_t)stream;
  int rc;
  if (d->N > 64) {
    p.tiles_m = ssak_cdiv(d->M, 128);
    p.tiles_n = ssak_cdiv(d->N, 128);
    rc = dispatch_layout<128, 128, 2, 2>(p, d->a_kmajor, d->b_kmajor, st);
  } else {
    p.tiles_m = ssak_cdiv(d->M, 128);
    p.tiles_n = ssak_cdiv(d->N, 64);
    rc = dispatch_layout<128, 64, 2, 2>(p, d->a_kmajor, d->b_kmajor, st);
  }
  if (rc != SSAK_OK) return rc;
  if (split > 1) {
    const long total = (long)p.nz * d->M * d->N;
    const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    splitk_reduce_kernel<<<blocks, 256, 0, st>>>(p);
    SSAK_LAUNCH_CHECK();
  }
  return SSAK_OK;
}

extern "C" int ssak_prof_enable(int on) {
  g_prof_on = on != 0;
  return SSAK_OK;
}

extern "C" int ssak_prof_collect(ssak_prof_entry* out, int cap) {
  SSAK_REQUIRE(out && cap >= 8, "prof_collect: need room for 8 entries");
  for (int i = 0; i < 8; ++i) {
    snprintf(out[i].name, sizeof(out[i].name), "%s", kVariantNames[i]);
    out[i].launches = 0;
    out[i].total_ms = 0.0;
    out[i].total_flops = 0.0;
  }
  for (ProfRec& r : g_prof) {
    SSAK_HIP(hipEventSynchronize(r.e1));
    float ms = 0.f;
    SSAK_HIP(hipEventElapsedTime(&ms, r.e0, r.e1));
    out[r.variant].launches += 1;
    out[r.variant].total_ms += ms;
    out[r.variant].total_flops += r.flops;
    g_event_pool.push_back(r.e0);
    g_event_pool.push_back(r.e1);
  }
  g_prof.clear();
  return 8;
}
