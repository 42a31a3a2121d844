// Do MFMA and VALU work of DIFFERENT waves on one SIMD overlap (gfx950)?  One workgroup per CU, 2 waves per SIMD (512 threads):
//   mode 0: every wave runs the MFMA loop            mode 1: every wave runs the VALU loop
//   mode 2: waves 0-3 MFMA loop, waves 4-7 VALU loop (one of each per SIMD)
//   mode 3: every wave runs both, interleaved in one instruction stream (4 VALU per MFMA)
// If the pipes overlap across waves, mode 2 takes max(mode 0, mode 1) / 2-ish of the work each ... printed as ns per loop iteration.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

__global__ __launch_bounds__(512) void k(float* out, int iters, int mode) {
  const int wave = threadIdx.x >> 6;
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) {
    a[i] = (__bf16)(threadIdx.x * 0.001f + i);
    b[i] = (__bf16)(threadIdx.x * 0.002f - i);
  }
  f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
  float f0 = threadIdx.x, f1 = f0 + 1, f2 = f0 + 2, f3 = f0 + 3, f4 = f0 + 4, f5 = f0 + 5, f6 = f0 + 6, f7 = f0 + 7;
  const float fc = 0.999f;
  const bool do_mfma = mode == 0 || mode == 3 || (mode == 2 && wave < 4);
  const bool do_valu = mode == 1 || mode == 3 || (mode == 2 && wave >= 4);
  if (mode == 3) {
    for (int i = 0; i < iters; ++i) {
#define M(c) c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
#define V(n) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(f##n) : "v"(fc));
      M(c0) V(0) V(1) V(2) V(3) M(c1) V(4) V(5) V(6) V(7) M(c2) V(0) V(1) V(2) V(3) M(c3) V(4) V(5) V(6) V(7)
    }
  } else {
    if (do_mfma)
      for (int i = 0; i < iters; ++i) { M(c0) M(c1) M(c2) M(c3) }
    if (do_valu)
      for (int i = 0; i < iters; ++i) { V(0) V(1) V(2) V(3) V(4) V(5) V(6) V(7) V(0) V(1) V(2) V(3) V(4) V(5) V(6) V(7) }
  }
  out[blockIdx.x * 512 + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3] + f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7;
}

int main() {
  float* out;
  (void)hipMalloc(&out, 256 * 512 * 4);
  const int iters = 20000;
  const char* names[4] = {"all waves: 4 MFMA / iter", "all waves: 16 v_fma / iter", "4 waves MFMA + 4 waves VALU", "all waves: 4 MFMA + 16 v_fma interleaved"};
  for (int mode = 0; mode < 4; ++mode) {
    hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, out, iters, mode);
    (void)hipDeviceSynchronize();
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, out, iters, mode);
    (void)hipEventRecord(e1);
    (void)hipDeviceSynchronize();
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    printf("mode %d  %-44s %8.2f ns per iteration\n", mode, names[mode], ms * 1e6 / iters);
  }
  return 0;
}
