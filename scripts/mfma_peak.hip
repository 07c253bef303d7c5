// measured ceiling of the matrix pipe on this box: back-to-back independent MFMAs, 4 waves per SIMD, every CU -- what
// "MFMA busy = 1.0" means in TFLOP/s at the clock the chip actually sustains (hipcc --offload-arch=gfx950 -O3 mfma_peak.hip)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
template <int MODE>
__global__ __launch_bounds__(256) void spin(float* out, int iters, unsigned long long* cyc) {
  f32x16 a0 = {}, a1 = {}, a2 = {}, a3 = {};
  f16x8 h = {(_Float16)1, (_Float16)1, (_Float16)1, (_Float16)1, (_Float16)1, (_Float16)1, (_Float16)1, (_Float16)1};
  float f = 1.0f + threadIdx.x * 1e-9f;
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; ++i) {
    if (MODE == 0) {
      a0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(h, h, a0, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(h, h, a1, 0, 0, 0);
      a2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(h, h, a2, 0, 0, 0);
      a3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(h, h, a3, 0, 0, 0);
    } else {
      a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(f, f, a0, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(f, f, a1, 0, 0, 0);
      a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(f, f, a2, 0, 0, 0);
      a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(f, f, a3, 0, 0, 0);
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  out[blockIdx.x * 256 + threadIdx.x] = a0[0] + a1[1] + a2[2] + a3[3];
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
// dependent-issue latency: ONE wave per SIMD with 1 / 2 / 4 independent accumulator chains of v_mfma_f32_32x32x16_f16
template <int CH>
__global__ __launch_bounds__(64) void chains(float* out, int iters, unsigned long long* cyc) {
  f32x16 a0 = {}, a1 = {}, a2 = {}, a3 = {};
  f16x8 h = {(_Float16)1, (_Float16)1, (_Float16)1, (_Float16)1, (_Float16)1, (_Float16)1, (_Float16)1, (_Float16)1};
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int r = 0; r < 12 / CH; ++r) {
      a0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(h, h, a0, 0, 0, 0);
      if (CH >= 2) a1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(h, h, a1, 0, 0, 0);
      if (CH >= 4) { a2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(h, h, a2, 0, 0, 0); a3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(h, h, a3, 0, 0, 0); }
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  out[blockIdx.x * 64 + threadIdx.x] = a0[0] + a1[1] + a2[2] + a3[3];
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

int main() {
  float* out; unsigned long long* cyc;
  hipMalloc(&out, 4096 * 256 * 4); hipMalloc(&cyc, 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int mode = 0; mode < 2; ++mode)
    for (int blocks : {256, 1024})
      for (int rep = 0; rep < 3; ++rep) {
        const int iters = mode == 0 ? 200000 : 100000;
        hipEventRecord(e0);
        if (mode == 0) hipLaunchKernelGGL(spin<0>, dim3(blocks), dim3(256), 0, 0, out, iters, cyc);
        else hipLaunchKernelGGL(spin<1>, dim3(blocks), dim3(256), 0, 0, out, iters, cyc);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        unsigned long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
        const double flop = (double)blocks * 4 * iters * 4 * (mode == 0 ? 2.0 * 32 * 32 * 16 : 2.0 * 32 * 32 * 2);
        printf("%s blocks=%4d: %8.2f ms, %7.1f TFLOP/s, s_memtime ticks %llu (%.3f GHz if ticks are shader cycles), cycles per MFMA per SIMD %.1f\n",
               mode == 0 ? "f16 32x32x16" : "f32 32x32x2 ", blocks, ms, flop / ms / 1e9, c, c / ms / 1e6,
               (double)c / ((double)iters * 4 * (blocks >= 1024 ? 4 : 1)));
      }
  // one wave per SIMD (1024 single-wave blocks), 12 MFMAs per iteration spread over CH accumulator chains
  for (int ch : {1, 2, 4}) {
    const int iters = 100000;
    hipEventRecord(e0);
    if (ch == 1) hipLaunchKernelGGL(chains<1>, dim3(1024), dim3(64), 0, 0, out, iters, cyc);
    else if (ch == 2) hipLaunchKernelGGL(chains<2>, dim3(1024), dim3(64), 0, 0, out, iters, cyc);
    else hipLaunchKernelGGL(chains<4>, dim3(1024), dim3(64), 0, 0, out, iters, cyc);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("one wave per SIMD, %d accumulator chain(s): %.1f ns per MFMA = %.1f cycles at 2.4 GHz (an independent MFMA: 32)\n", ch,
           ms * 1e6 / (iters * 12.0), ms * 1e6 / (iters * 12.0) * 2.4);
  }
  return 0;
}
