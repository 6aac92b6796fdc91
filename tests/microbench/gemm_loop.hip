// The W_hh loop of pmg_fwd.hip in isolation (not part of the product or the tests): 8 waves per CU,
// per "step" two passes of 32 reduction groups x 6 tiles x 4 MFMAs, weights (3 MB, shared by every
// workgroup, L2 resident) through a buffer resource with two register sets, B operand from LDS.
// Template switches remove one ingredient at a time to see what costs MFMA issue slots.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float4 bload(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0);
  return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}

template <bool LOADS, bool LDSX, bool SYNC, int DEPTH>
__global__ __launch_bounds__(512, 2) void k(const float* w, float* out, long long* cyc, int steps) {
  __shared__ __attribute__((aligned(16))) float xs[16 * 520];
  constexpr int NT = 6, KG = 32;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int b = lane & 15, q = lane >> 4;
  for (int i = threadIdx.x; i < 16 * 520; i += 512) xs[i] = i * 1e-4f;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)w, 0, 3u << 20, 0x00020000);
  const unsigned voff = lane * 16u, ts = KG * 1024u;
  const float* xrow = xs + b * 520 + 4 * q;
  f32x4 acc[NT];
  for (int i = 0; i < NT; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float4 wa[NT], wb[NT];
  for (int t = 0; t < NT; ++t) { wa[t] = make_float4(1.f, 2.f, 3.f, 4.f); wb[t] = wa[t]; }
  __syncthreads();
  const long long t0 = __builtin_amdgcn_s_memtime();
#define LOAD(dst, so) if (LOADS) { _Pragma("unroll") for (int t = 0; t < NT; ++t) dst[t] = bload(rs, voff, (so) + t * ts); }
#define STEP(wc, C) _Pragma("unroll") for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(wc[t].C, x.C, acc[t], 0, 0, 0);
#pragma unroll 1
  for (int s = 0; s < steps; ++s) {
#pragma unroll 1
    for (int ps = 0; ps < 2; ++ps) {
      const unsigned soff = (unsigned)((wave * 2 + ps) * NT) * ts;
      const unsigned nsoff = (unsigned)((wave * 2 + (ps ^ 1)) * NT) * ts;
      if (DEPTH != 3 || (s == 0 && ps == 0)) LOAD(wa, soff)
      float4 xa = LDSX ? *reinterpret_cast<const float4*>(xrow) : make_float4(1.f, 1.f, 1.f, 1.f);
#pragma unroll 1
      for (int kg = 0; kg < KG; kg += 2) {
        if (DEPTH == 4) {
          // wb(kg+1) loads ride in the shadow of the first MFMAs of the wa block, wa(kg+2) loads in
          // the shadow of the first MFMAs of the wb block: one VMEM issue per MFMA issue
          const int kn = kg + 2 < KG ? kg + 2 : kg;
          float4 xb = *reinterpret_cast<const float4*>(xrow + 16 * kg + 16);
          float4 wn[NT];
          __builtin_amdgcn_sched_barrier(0);
          {
            const float4 x = xa;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
              acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[t].x, x.x, acc[t], 0, 0, 0);
              wb[t] = bload(rs, voff, soff + (kg + 1) * 1024u + t * ts);
              __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
              __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
            }
            STEP(wa, y) STEP(wa, z) STEP(wa, w)
          }
          __builtin_amdgcn_sched_barrier(0);
          xa = *reinterpret_cast<const float4*>(xrow + 16 * kn);
          {
            const float4 x = xb;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
              acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(wb[t].x, x.x, acc[t], 0, 0, 0);
              wn[t] = bload(rs, voff, soff + kn * 1024u + t * ts);
              __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
              __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
            }
            STEP(wb, y) STEP(wb, z) STEP(wb, w)
          }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int t = 0; t < NT; ++t) wa[t] = wn[t];
          continue;
        }
        if (DEPTH == 5) __builtin_amdgcn_s_setprio(0);
        LOAD(wb, soff + (kg + 1) * 1024u)
        float4 xb = xa;
        if (LDSX) xb = *reinterpret_cast<const float4*>(xrow + 16 * kg + 16);
        __builtin_amdgcn_sched_barrier(0);
        if (DEPTH == 5) __builtin_amdgcn_s_setprio(2);
        { const float4 x = xa; STEP(wa, x) STEP(wa, y) STEP(wa, z) STEP(wa, w) }
        __builtin_amdgcn_sched_barrier(0);
        if (DEPTH == 5) __builtin_amdgcn_s_setprio(0);
        {
          const int kn = kg + 2 < KG ? kg + 2 : kg;
          const unsigned so = (DEPTH == 3 && kg + 2 >= KG) ? nsoff : soff + kn * 1024u;
          LOAD(wa, so)
          if (LDSX) xa = *reinterpret_cast<const float4*>(xrow + 16 * kn);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (DEPTH == 5) __builtin_amdgcn_s_setprio(2);
        { const float4 x = xb; STEP(wb, x) STEP(wb, y) STEP(wb, z) STEP(wb, w) }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (SYNC) __syncthreads();
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  float sum = 0.f;
  for (int i = 0; i < NT; ++i) sum += acc[i].x + acc[i].y + acc[i].z + acc[i].w;
  out[blockIdx.x * 512 + threadIdx.x] = sum;
  if (lane == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}

template <bool LOADS, bool LDSX, bool SYNC, int DEPTH = 2>
void run(const char* name, const float* w, int steps) {
  float* out; long long* cyc;
  hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 256 * 8 * 8);
  auto kern = k<LOADS, LDSX, SYNC, DEPTH>;
  hipLaunchKernelGGL(kern, dim3(256), dim3(512), 0, 0, w, out, cyc, 2);
  hipLaunchKernelGGL(kern, dim3(256), dim3(512), 0, 0, w, out, cyc, steps);
  hipDeviceSynchronize();
  long long h[8]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  const double ideal = (double)steps * 2 * 32 * 24 * 32 * 2;  // MFMAs per wave x 32 cycles x 2 waves per SIMD
  printf("%-28s cycles per step: waves 0-3 %7.0f  waves 4-7 %7.0f   (MFMA-bound: %.0f)  issue %.3f\n", name,
         (double)h[0] / steps, (double)h[4] / steps, ideal / steps, ideal / (double)(h[4] > h[0] ? h[4] : h[0]));
  hipFree(out); hipFree(cyc);
}

int main() {
  float* w; hipMalloc(&w, 3 << 20); hipMemset(w, 0, 3 << 20);
  const int S = 256;
  run<false, false, false>("mfma only", w, S);
  run<false, true, false>("+ B operand from LDS", w, S);
  run<true, false, false>("+ weight loads", w, S);
  run<true, true, false>("+ both", w, S);
  run<true, true, true>("+ both + barrier per step", w, S);
  run<false, false, true>("mfma only + barrier", w, S);
  run<true, true, true, 3>("+ both + barrier, preload", w, S);
  run<true, true, true, 4>("+ both + barrier, interleaved", w, S);
  run<true, true, true, 5>("+ both + barrier, mfma prio", w, S);
  return 0;
}
