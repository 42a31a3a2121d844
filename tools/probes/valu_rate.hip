// Issue rate of the VALU instructions the attention kernels are made of (gfx950): cycles per wave-instruction with 1, 2 and 4
// waves per SIMD, 8 independent chains per wave.  Build: hipcc --offload-arch=gfx950 -O3 tools/probes/valu_rate.hip -o valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

template <int OP>
__global__ void k(unsigned* out, unsigned long long* cyc, int iters, unsigned seed) {
  unsigned v0 = threadIdx.x + seed, v1 = v0 * 3u + 1u, v2 = v0 * 5u + 2u, v3 = v0 * 7u + 3u, v4 = v0 * 11u, v5 = v0 * 13u, v6 = v0 * 17u, v7 = v0 * 19u;
  float f0 = v0 * 1e-9f, f1 = v1 * 1e-9f, f2 = v2 * 1e-9f, f3 = v3 * 1e-9f, f4 = v4 * 1e-9f, f5 = v5 * 1e-9f, f6 = v6 * 1e-9f, f7 = v7 * 1e-9f;
  unsigned long long w0 = v0, w1 = v1, w2 = v2, w3 = v3;
  const unsigned c = seed | 1u;
  const float fc = 0.999f;
  unsigned long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; ++i) {
    if (OP == 0) {  // v_mul_lo_u32
#define X(n) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(v##n) : "v"(c));
      REP8(X) REP8(X)
#undef X
    } else if (OP == 1) {  // v_xor (full rate reference)
#define X(n) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(v##n) : "v"(c));
      REP8(X) REP8(X)
#undef X
    } else if (OP == 2) {  // v_exp_f32
#define X(n) asm volatile("v_exp_f32 %0, %0" : "+v"(f##n));
      REP8(X) REP8(X)
#undef X
    } else if (OP == 3) {  // v_mad_u64_u32 (64-bit product)
      asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(w0) : "v"(v0), "v"(c) : "vcc");
      asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(w1) : "v"(v1), "v"(c) : "vcc");
      asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(w2) : "v"(v2), "v"(c) : "vcc");
      asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(w3) : "v"(v3), "v"(c) : "vcc");
      asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(w0) : "v"(v4), "v"(c) : "vcc");
      asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(w1) : "v"(v5), "v"(c) : "vcc");
      asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(w2) : "v"(v6), "v"(c) : "vcc");
      asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(w3) : "v"(v7), "v"(c) : "vcc");
      asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(w0) : "v"(v0), "v"(c) : "vcc");
      asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(w1) : "v"(v1), "v"(c) : "vcc");
      asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(w2) : "v"(v2), "v"(c) : "vcc");
      asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(w3) : "v"(v3), "v"(c) : "vcc");
      asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(w0) : "v"(v4), "v"(c) : "vcc");
      asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(w1) : "v"(v5), "v"(c) : "vcc");
      asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(w2) : "v"(v6), "v"(c) : "vcc");
      asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(w3) : "v"(v7), "v"(c) : "vcc");
    } else if (OP == 4) {  // v_fma_f32
#define X(n) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(f##n) : "v"(fc));
      REP8(X) REP8(X)
#undef X
    } else if (OP == 5) {  // v_pk_fma_f32 (two results per lane)
      typedef float fv2 __attribute__((ext_vector_type(2)));
      fv2 a = {f0, f1}, b = {f2, f3}, cc = {f4, f5}, d = {f6, f7}, k2 = {fc, fc};
#define Y(r) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(r) : "v"(k2));
      Y(a) Y(b) Y(cc) Y(d) Y(a) Y(b) Y(cc) Y(d) Y(a) Y(b) Y(cc) Y(d) Y(a) Y(b) Y(cc) Y(d)
#undef Y
      f0 = a[0]; f1 = a[1]; f2 = b[0]; f3 = b[1]; f4 = cc[0]; f5 = cc[1]; f6 = d[0]; f7 = d[1];
    } else if (OP == 6) {  // v_cmp + v_cndmask pairs (vcc dependency)
#define X(n) asm volatile("v_cmp_le_u32 vcc, %1, %0\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(v##n) : "v"(c) : "vcc");
      REP8(X)
#undef X
    } else if (OP == 7) {  // v_mul_u32_u24
#define X(n) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(v##n) : "v"(c));
      REP8(X) REP8(X)
#undef X
    } else if (OP == 8) {  // v_cvt_pk_bf16_f32
#define X(n) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(f##n) : "v"(fc));
      REP8(X) REP8(X)
#undef X
    } else if (OP == 9) {  // v_mov_b32 dpp quad_perm
#define X(n) asm volatile("v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(v##n));
      REP8(X) REP8(X)
#undef X
    }
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  out[blockIdx.x * blockDim.x + threadIdx.x] = v0 ^ v1 ^ v2 ^ v3 ^ v4 ^ v5 ^ v6 ^ v7 ^ __float_as_uint(f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7) ^ (unsigned)(w0 ^ w1 ^ w2 ^ w3);
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int OP>
void run(const char* name, int instr_per_iter) {
  unsigned* out;
  unsigned long long* cyc;
  hipMalloc(&out, 256 * 1024 * 4 * 16);
  hipMalloc(&cyc, 8192 * 8);
  const int iters = 2000;
  printf("%-28s", name);
  for (int waves_per_simd : {1, 2, 4}) {
    const int threads = 256 * waves_per_simd;  // one workgroup per CU: waves_per_simd waves on each of the 4 SIMDs
    hipLaunchKernelGGL(k<OP>, dim3(256), dim3(threads), 0, 0, out, cyc, iters, 12345u);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<OP>, dim3(256), dim3(threads), 0, 0, out, cyc, iters, 12345u);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(256);
    hipMemcpy(h.data(), cyc, 256 * 8, hipMemcpyDeviceToHost);
    double avg = 0;
    for (auto v : h) avg += v;
    avg /= 256;
    // wave-instructions issued per SIMD = waves_per_simd * iters * instr_per_iter
    const double per = (double)ms * 1e6 / ((double)waves_per_simd * iters * instr_per_iter);  // ns per wave-instruction per SIMD
    printf("  %dw: %6.2f ns/instr (%5.1f cyc @2.4GHz, counter %5.1f)", waves_per_simd, per, per * 2.4, avg / ((double)waves_per_simd * iters * instr_per_iter));
  }
  printf("\n");
}

int main() {
  run<1>("v_xor_b32", 16);
  run<4>("v_fma_f32", 16);
  run<5>("v_pk_fma_f32", 16);
  run<0>("v_mul_lo_u32", 16);
  run<7>("v_mul_u32_u24", 16);
  run<3>("v_mad_u64_u32", 16);
  run<2>("v_exp_f32", 16);
  run<6>("v_cmp+v_cndmask (pair)", 16);
  run<8>("v_cvt_pk_bf16_f32", 16);
  run<9>("v_mov_b32_dpp", 16);
  return 0;
}
