// Check of the DPP wave reductions used by pmg_fwdc.hip (wave_max, wave_sum63, wave_argmax) against a serial loop.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#define FC_DPP(x, ctrl, rmask) __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, x), __builtin_bit_cast(int, x), ctrl, rmask, 0xf, false))
#define FC_DPPI(x, ctrl, rmask) __builtin_amdgcn_update_dpp(x, x, ctrl, rmask, 0xf, false)
// the builtin is typed int: a float argument would be CONVERTED, not moved
__device__ __forceinline__ float readlane_f(const float x, const int l) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), l));
}
__device__ __forceinline__ float wave_max(float x) {
  x = fmaxf(x, FC_DPP(x, 0xB1, 0xf)); x = fmaxf(x, FC_DPP(x, 0x4E, 0xf));
  x = fmaxf(x, FC_DPP(x, 0x141, 0xf)); x = fmaxf(x, FC_DPP(x, 0x140, 0xf));
  x = fmaxf(x, FC_DPP(x, 0x142, 0xa)); x = fmaxf(x, FC_DPP(x, 0x143, 0xc));
  return readlane_f(x, 63);
}
__device__ __forceinline__ float wave_sum63(float x) {
  x += FC_DPP(x, 0xB1, 0xf); x += FC_DPP(x, 0x4E, 0xf);
  x += FC_DPP(x, 0x141, 0xf); x += FC_DPP(x, 0x140, 0xf);
  const float r15 = FC_DPP(x, 0x142, 0xa);
  x = ((threadIdx.x & 16) ? x + r15 : x);
  const float r31 = FC_DPP(x, 0x143, 0xc);
  x = ((threadIdx.x & 32) ? x + r31 : x);
  return readlane_f(x, 63);
}
__global__ void k(const float* in, float* out) {
  const float x = in[threadIdx.x];
  const float m = wave_max(x), s = wave_sum63(x);
  float a = x; a += FC_DPP(a, 0xB1, 0xf);
  float b = x; b += FC_DPP(b, 0x141, 0xf);
  float c = x; c += FC_DPP(c, 0x140, 0xf);
  float d = FC_DPP(x, 0x142, 0xa), e = FC_DPP(x, 0x143, 0xc);
  if (threadIdx.x == 0) { out[0] = m; out[1] = s; }
  out[8 + threadIdx.x] = a; out[72 + threadIdx.x] = b; out[136 + threadIdx.x] = c; out[200 + threadIdx.x] = d; out[264 + threadIdx.x] = e;
}
int main() {
  float h[64], o[400]; float mx = -1e30f, sm = 0;
  for (int i = 0; i < 64; ++i) { h[i] = (float)((i * 37) % 64) + 0.25f; mx = fmaxf(mx, h[i]); sm += h[i]; }
  float *di, *dout; hipMalloc(&di, 256); hipMalloc(&dout, 1600);
  hipMemcpy(di, h, 256, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, di, dout); hipMemcpy(o, dout, 1600, hipMemcpyDeviceToHost);
  printf("max %g (want %g)  sum %g (want %g)\n", o[0], mx, o[1], sm);
  int bad = 0;
  for (int i = 0; i < 64; ++i) {
    if (o[8 + i] != h[i] + h[i ^ 1]) { if (bad++ < 4) printf("xor1 lane %d got %g want %g\n", i, o[8 + i], h[i] + h[i ^ 1]); }
    const int hm = (i & ~7) | (7 - (i & 7)), rm = (i & ~15) | (15 - (i & 15));
    if (o[72 + i] != h[i] + h[hm]) { if (bad++ < 8) printf("half_mirror lane %d got %g want %g\n", i, o[72 + i], h[i] + h[hm]); }
    if (o[136 + i] != h[i] + h[rm]) { if (bad++ < 12) printf("mirror lane %d got %g want %g\n", i, o[136 + i], h[i] + h[rm]); }
  }
  printf("bcast15 by row: "); for (int r = 0; r < 4; ++r) printf("row%d lane%d=%g ", r, 16 * r, o[200 + 16 * r]); printf("(h15=%g h31=%g h47=%g)\n", h[15], h[31], h[47]);
  printf("bcast31 by row: "); for (int r = 0; r < 4; ++r) printf("row%d lane%d=%g ", r, 16 * r, o[264 + 16 * r]); printf("(h31=%g)\n", h[31]);
  printf("bad %d\n", bad);
  return 0;
}
