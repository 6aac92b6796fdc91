// Issue rate of v_mfma_f32_16x16x4_f32 on gfx950: NW waves per workgroup (one workgroup per CU), each
// wave runs ITER x 24 independent-by-6 MFMAs from registers (no memory traffic).  Prints cycles per
// MFMA per SIMD (32 = the rate behind the 157.3 TFLOP/s peak).  Not part of the product or the tests.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ __launch_bounds__(1024) void k(float* out, long long* cyc, int iters) {
  f32x4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float a = threadIdx.x * 1e-3f, b = threadIdx.x * 2e-3f;
  __syncthreads();
  const long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int kk = 0; kk < 4; ++kk)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
  for (int i = 0; i < NACC; ++i) s += acc[i].x + acc[i].y + acc[i].z + acc[i].w;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

template <int NACC>
void run(int nw, int iters) {
  float* out; long long* cyc;
  hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&cyc, 256 * 16 * 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<NACC>, dim3(256), dim3(nw * 64), 0, 0, out, cyc, 10);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<NACC>, dim3(256), dim3(nw * 64), 0, 0, out, cyc, iters);
  hipEventRecord(e1); hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  long long h[16]; hipMemcpy(h, cyc, sizeof(long long) * nw, hipMemcpyDeviceToHost);
  long long mx = 0, mn = 1LL << 60; for (int i = 0; i < nw; ++i) { if (h[i] > mx) mx = h[i]; if (h[i] < mn) mn = h[i]; }
  const double mf = (double)iters * 4 * NACC;               // MFMAs per wave
  const double per_simd = mf * (nw / 4.0);                  // MFMAs per SIMD
  printf("NACC=%d waves/WG=%d: wave cycles min %lld max %lld -> %.2f cycles per MFMA per SIMD (slowest wave); %.1f TFLOP/s by events (%.3f ms)\n",
         NACC, nw, mn, mx, mx / per_simd, 256.0 * nw * mf * 2048 / (ms * 1e-3) / 1e12, ms);
  hipFree(out); hipFree(cyc);
}

int main() {
  run<6>(4, 20000); run<6>(8, 20000); run<6>(16, 10000);
  run<12>(4, 10000); run<12>(8, 10000);
  run<3>(8, 20000); run<2>(8, 20000); run<1>(8, 20000);
  return 0;
}
