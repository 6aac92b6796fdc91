// The OTHER decomposition of the recurrent product (hard part 2), measured in
// isolation; not part of the product or the tests.
//
// Product kernels: batch-partitioned -- a workgroup owns 16 sequences for all steps and streams ALL of
// W_hh (3 MB at hidden 512) from L2 every step; no synchronisation between workgroups.
// Here: hidden-partitioned -- the H/16 CUs of a group each own 16 hidden units = 48 gate rows of W_hh
// (96 KB at hidden 512), RESIDENT in LDS, for ALL sequences of the group; per step a CU reads the whole
// hidden state of the group's S sequences from L2 (packed [seq tile][k group][lane][4], 1 KB blocks),
// multiplies, writes its 16 units of the new state (one 1 KB block per sequence tile) and the group
// synchronises through an L2 atomic counter.  Groups are cut along blockIdx % 8, which is the XCD of
// a workgroup when every CU holds one (printed, from the XCC_ID hardware register), so a group shares
// one L2.
//
// Measured: (1) the group barrier alone; (2) cycles per step of the product loop for S sequences per
// group, with and without the barrier / the state exchange; LDS bytes read per cycle.
// Every spin is bounded: a barrier that does not complete within SPIN_LIMIT polls raises a flag and
// every workgroup leaves, so the grid always drains.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define SPIN_LIMIT 4000000

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

struct Args {
  float* hbuf;          // [2][groups][S/16][H/16][64][4]
  const float* w;       // [H/16 CUs][48 rows][H]  (any values)
  unsigned* ctr;        // [groups] barrier counters (monotonic)
  unsigned* flag;       // [1] set when a barrier timed out
  long long* cyc;       // [blocks] cycles of the timed loop
  unsigned* xcc;        // [blocks] XCC_ID of the workgroup
  int steps, S, cus_per_group, groups_per_xcd;
  std::vector<float> hinit;   // host copy of the initial state (re-uploaded before the timed launch)
};

// one barrier over the `n` workgroups of a group; target = n * (barriers so far + 1)
__device__ __forceinline__ bool group_barrier(unsigned* ctr, unsigned target, unsigned* flag) {
  __syncthreads();
  bool ok = true;
  if (threadIdx.x == 0) {
    __atomic_fetch_add(ctr, 1u, __ATOMIC_RELEASE);       // device scope, resolved in L2
    int spins = 0;
    while (__atomic_load_n(ctr, __ATOMIC_ACQUIRE) < target) {
      if (++spins > SPIN_LIMIT || __atomic_load_n(flag, __ATOMIC_RELAXED)) { __atomic_store_n(flag, 1u, __ATOMIC_RELAXED); ok = false; break; }
      __builtin_amdgcn_s_sleep(1);
    }
  }
  __syncthreads();
  return ok;
}

// The two hand-off forms MI355X_MICROARCH.md lists as valid inside one launch (section "Workgroup
// dispatch, XCD placement & inter-workgroup visibility"); a hand-written buffer_inv sc0 or
// workgroup-scope atomics are NOT among them (tried here first: stale reads whenever the state fits
// the CU's L1, and a compiler that turns an idempotent RMW poll into a plain load).
//   LIGHT = 1  sc1 (write-through) 16-byte stores of the state, every wave s_waitcnt vmcnt(0), workgroup
//              barrier, lane 0: relaxed agent atomic add; poll = relaxed sc1 load + s_sleep; workgroup
//              barrier; EVERY load of the state is a 16-byte sc1 buffer load.  No fence at all.
//   LIGHT = 2  plain stores, every wave s_waitcnt vmcnt(0), workgroup barrier, lane 0: release fence
//              (buffer_wbl2 sc1) + s_waitcnt + relaxed agent atomic add; poll as above; lane 0: acquire
//              fence (buffer_inv sc1) + s_waitcnt; workgroup barrier; plain loads.
template <int LIGHT>
__device__ __forceinline__ bool group_barrier_light(unsigned* ctr, unsigned target, unsigned* flag) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  bool ok = true;
  if (threadIdx.x == 0) {
    if (LIGHT == 2) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    int spins = 0;
    while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      if (++spins > SPIN_LIMIT || __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
        __hip_atomic_store(flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); ok = false; break;
      }
      __builtin_amdgcn_s_sleep(1);
    }
    if (LIGHT == 2) {
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
  }
  __syncthreads();
  return ok;
}

// 16-byte access to the exchanged state through a wave-uniform buffer resource over the whole buffer
// (voff = byte offset of this lane's element): LIGHT = 1 carries the sc1 bit (aux = 16), else plain.
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
template <int LIGHT> __device__ __forceinline__ float4 ldx(const __amdgpu_buffer_rsrc_t r, unsigned voff) {
  const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(r, voff, 0, LIGHT == 1 ? 16 : 0);
  return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}
template <int LIGHT> __device__ __forceinline__ void stx(const __amdgpu_buffer_rsrc_t r, unsigned voff, const f32x4 h) {
  u32x4_t u;
  u.x = __float_as_uint(h[0]); u.y = __float_as_uint(h[1]); u.z = __float_as_uint(h[2]); u.w = __float_as_uint(h[3]);
  __builtin_amdgcn_raw_buffer_store_b128(u, r, voff, 0, LIGHT == 1 ? 16 : 0);
}

template <int LIGHT>
__global__ __launch_bounds__(512) void barrier_only(Args a) {
  extern __shared__ float lds[];     // large: one workgroup per CU
  const int grp = blockIdx.x & 7, n = gridDim.x >> 3;
  if (threadIdx.x == 0) {
    a.xcc[blockIdx.x] = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & 0xf;   // HW_REG_XCC_ID[3:0]
    lds[0] = 0.f;
  }
  group_barrier(a.ctr + grp, n, a.flag);      // start together
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int s = 0; s < a.steps; ++s)
    if (!(LIGHT ? group_barrier_light<LIGHT>(a.ctr + grp, (unsigned)n * (unsigned)(s + 2), a.flag)
                : group_barrier(a.ctr + grp, (unsigned)n * (unsigned)(s + 2), a.flag))) break;
  const long long t1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) a.cyc[blockIdx.x] = t1 - t0;
}

// H: hidden size; a group = H/16 workgroups (CUs).  MODE 0: full (exchange + barrier), 1: no barrier,
// 2: no barrier and no state exchange (B operand re-read from the same buffer, nothing written).
template <int H, int MODE, int NJ, int LIGHT>
__global__ __launch_bounds__(512) void hidden_part(Args a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int KG = H / 16;            // k groups = CUs per group
  float* wl = lds;                      // [3 row tiles][KG][64][4]  packed A operand (48 rows x H)
  float* red = lds + 3 * KG * 256;      // [8 waves][3][256]        K-split partials (small S)
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;          // slot: index of this CU inside its XCD
  const int gsub = slot / KG, cu = slot - gsub * KG;               // group inside the XCD, CU inside the group
  if (gsub >= a.groups_per_xcd) return;
  const int grp = xcd * a.groups_per_xcd + gsub;
  const int S = a.S, ST = S / 16;
  // weights of this CU into LDS, packed (values do not matter)
  for (int i = tid; i < 3 * KG * 256; i += 512) wl[i] = a.w[(size_t)cu * 3 * KG * 256 + i] * 1e-3f;
  __syncthreads();
  const size_t hsz = (size_t)ST * KG * 256;                        // floats of one state buffer of one group
  float* hb0 = a.hbuf + (size_t)(2 * grp) * hsz;
  float* hb1 = hb0 + hsz;
  unsigned bar = 1;
  if (MODE == 0) group_barrier(a.ctr + grp, KG * bar, a.flag);
  ++bar;
  const long long t0 = __builtin_amdgcn_s_memtime();
  const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
  bool ok = true;
#pragma unroll 1
  for (int s = 0; s < a.steps && ok; ++s) {
    const float* hin = (MODE == 2 || !(s & 1)) ? hb0 : hb1;
    float* hout = (s & 1) ? hb0 : hb1;
    const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(hin), 0, (unsigned)(hsz * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rout = __builtin_amdgcn_make_buffer_rsrc(hout, 0, (unsigned)(hsz * 4), 0x00020000);
    if (ST >= 8) {
      // every wave owns ST/8 sequence tiles over the whole reduction
      for (int t0s = wave; t0s < ST; t0s += 8 * NJ) {
        f32x4 acc[NJ][3];
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[j][0] = acc[j][1] = acc[j][2] = z4;
        unsigned hp[NJ];     // byte offset of (sequence tile, k group 0, this lane)
#pragma unroll
        for (int j = 0; j < NJ; ++j) hp[j] = (unsigned)(((t0s + 8 * j < ST) ? t0s + 8 * j : t0s) * KG * 256 + lane * 4) * 4u;
        float4 xa[NJ], xb[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) xa[j] = ldx<LIGHT>(rin, hp[j]);
#pragma unroll 1
        for (int kg = 0; kg < KG; kg += 2) {
#pragma unroll
          for (int j = 0; j < NJ; ++j) xb[j] = ldx<LIGHT>(rin, hp[j] + (unsigned)(kg + 1) * 1024u);
          __builtin_amdgcn_sched_barrier(0);
#define HP_STEP(X, KGI)                                                                              \
          {                                                                                          \
            const float4 w0 = *reinterpret_cast<const float4*>(wl + (0 * KG + (KGI)) * 256 + lane * 4);  \
            const float4 w1 = *reinterpret_cast<const float4*>(wl + (1 * KG + (KGI)) * 256 + lane * 4);  \
            const float4 w2 = *reinterpret_cast<const float4*>(wl + (2 * KG + (KGI)) * 256 + lane * 4);  \
            _Pragma("unroll") for (int j = 0; j < NJ; ++j) {                                          \
              acc[j][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(w0.x, X[j].x, acc[j][0], 0, 0, 0);    \
              acc[j][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(w1.x, X[j].x, acc[j][1], 0, 0, 0);    \
              acc[j][2] = __builtin_amdgcn_mfma_f32_16x16x4f32(w2.x, X[j].x, acc[j][2], 0, 0, 0);    \
              acc[j][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(w0.y, X[j].y, acc[j][0], 0, 0, 0);    \
              acc[j][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(w1.y, X[j].y, acc[j][1], 0, 0, 0);    \
              acc[j][2] = __builtin_amdgcn_mfma_f32_16x16x4f32(w2.y, X[j].y, acc[j][2], 0, 0, 0);    \
              acc[j][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(w0.z, X[j].z, acc[j][0], 0, 0, 0);    \
              acc[j][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(w1.z, X[j].z, acc[j][1], 0, 0, 0);    \
              acc[j][2] = __builtin_amdgcn_mfma_f32_16x16x4f32(w2.z, X[j].z, acc[j][2], 0, 0, 0);    \
              acc[j][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(w0.w, X[j].w, acc[j][0], 0, 0, 0);    \
              acc[j][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(w1.w, X[j].w, acc[j][1], 0, 0, 0);    \
              acc[j][2] = __builtin_amdgcn_mfma_f32_16x16x4f32(w2.w, X[j].w, acc[j][2], 0, 0, 0);    \
            }                                                                                        \
          }
          HP_STEP(xa, kg)
          __builtin_amdgcn_sched_barrier(0);
          const int kn = (kg + 2 < KG) ? kg + 2 : kg;
#pragma unroll
          for (int j = 0; j < NJ; ++j) xa[j] = ldx<LIGHT>(rin, hp[j] + (unsigned)kn * 1024u);
          __builtin_amdgcn_sched_barrier(0);
          HP_STEP(xb, kg + 1)
          __builtin_amdgcn_sched_barrier(0);
        }
        // "gates": new state of this CU's 16 units for these sequence tiles (D layout -> packed block)
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          if (t0s + 8 * j < ST) {
            f32x4 hn;
#pragma unroll
            for (int c = 0; c < 4; ++c) hn[c] = __cosf(40.f * (acc[j][0][c] - acc[j][2][c]) + 17.f * acc[j][1][c] + 0.3f);
            if (MODE != 2) stx<LIGHT>(rout, (unsigned)(((t0s + 8 * j) * KG + cu) * 256 + lane * 4) * 4u, hn);
            else if (hn[0] == 12345.678f) hout[0] = hn[1];
          }
        }
      }
    } else {
      // few sequences: K split over the eight waves, partials reduced through LDS
      for (int t = 0; t < ST; ++t) {
        f32x4 acc0 = z4, acc1 = z4, acc2 = z4;
        const unsigned hp = (unsigned)(t * KG * 256 + lane * 4) * 4u;
        constexpr int KPW = KG / 8;
        float4 x[KPW];
#pragma unroll
        for (int i = 0; i < KPW; ++i) x[i] = ldx<LIGHT>(rin, hp + (unsigned)(wave * KPW + i) * 1024u);
#pragma unroll
        for (int i = 0; i < KPW; ++i) {
          const int kg = wave * KPW + i;
          const float4 w0 = *reinterpret_cast<const float4*>(wl + (0 * KG + kg) * 256 + lane * 4);
          const float4 w1 = *reinterpret_cast<const float4*>(wl + (1 * KG + kg) * 256 + lane * 4);
          const float4 w2 = *reinterpret_cast<const float4*>(wl + (2 * KG + kg) * 256 + lane * 4);
          acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(w0.x, x[i].x, acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(w1.x, x[i].x, acc1, 0, 0, 0);
          acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(w2.x, x[i].x, acc2, 0, 0, 0);
          acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(w0.y, x[i].y, acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(w1.y, x[i].y, acc1, 0, 0, 0);
          acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(w2.y, x[i].y, acc2, 0, 0, 0);
          acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(w0.z, x[i].z, acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(w1.z, x[i].z, acc1, 0, 0, 0);
          acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(w2.z, x[i].z, acc2, 0, 0, 0);
          acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(w0.w, x[i].w, acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(w1.w, x[i].w, acc1, 0, 0, 0);
          acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(w2.w, x[i].w, acc2, 0, 0, 0);
        }
        float* rp = red + (size_t)wave * 768 + lane * 4;
        *reinterpret_cast<f32x4*>(rp) = acc0;
        *reinterpret_cast<f32x4*>(rp + 256) = acc1;
        *reinterpret_cast<f32x4*>(rp + 512) = acc2;
        __syncthreads();
        if (tid < 64) {      // one wave: 16 bytes per lane, every 128-B line written whole by one store instruction
          f32x4 hn;
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            float r = 0.f, zz = 0.f, n = 0.f;
#pragma unroll
            for (int wv = 0; wv < 8; ++wv) { r += red[wv * 768 + 4 * tid + c]; zz += red[wv * 768 + 256 + 4 * tid + c]; n += red[wv * 768 + 512 + 4 * tid + c]; }
            hn[c] = __cosf(40.f * (r - n) + 17.f * zz + 0.3f);
          }
          if (MODE != 2) stx<LIGHT>(rout, (unsigned)((t * KG + cu) * 256 + 4 * tid) * 4u, hn);
          else if (hn[0] == 12345.678f) hout[0] = hn[1];
        }
        __syncthreads();
      }
    }
    if (MODE == 0) {
      if (LIGHT) {
        ok = group_barrier_light<LIGHT>(a.ctr + grp, (unsigned)KG * bar, a.flag);
      } else {
        __threadfence();
        ok = group_barrier(a.ctr + grp, (unsigned)KG * bar, a.flag);
      }
      ++bar;
    }
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  if (tid == 0) a.cyc[blockIdx.x] = t1 - t0;
}

template <int H, int MODE, int NJ, int LIGHT>
static void run_t(const char* what, Args a, int S, int steps, int groups_per_xcd) {
  constexpr int KG = H / 16;
  a.S = S; a.steps = steps; a.cus_per_group = KG; a.groups_per_xcd = groups_per_xcd;
  const int groups = 8 * groups_per_xcd;
  const size_t lds = (size_t)(3 * KG * 256 + 8 * 768) * 4;
  const size_t lds_force = lds < 90 * 1024 ? 90 * 1024 : lds;      // > half of 160 KiB: one workgroup per CU
  CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(hidden_part<H, MODE, NJ, LIGHT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_force));
  CHECK(hipMemset(a.ctr, 0, 64 * sizeof(unsigned)));
  CHECK(hipMemset(a.flag, 0, sizeof(unsigned)));
  {
    const size_t nfl = (size_t)2 * groups * (S / 16) * KG * 256;
    std::vector<float> h0(nfl);
    for (size_t i = 0; i < nfl; ++i) h0[i] = (float)((i * 40503u) >> 7 & 255) / 256.f - 0.5f;
    CHECK(hipMemcpy(a.hbuf, h0.data(), nfl * 4, hipMemcpyHostToDevice));
    a.hinit = h0;
  }
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  hipLaunchKernelGGL((hidden_part<H, MODE, NJ, LIGHT>), dim3(256), dim3(512), lds_force, 0, a);      // warm-up (code, caches)
  CHECK(hipMemset(a.ctr, 0, 64 * sizeof(unsigned)));
  CHECK(hipMemcpy(a.hbuf, a.hinit.data(), a.hinit.size() * 4, hipMemcpyHostToDevice));
  CHECK(hipEventRecord(e0));
  hipLaunchKernelGGL((hidden_part<H, MODE, NJ, LIGHT>), dim3(256), dim3(512), lds_force, 0, a);
  CHECK(hipEventRecord(e1));
  CHECK(hipDeviceSynchronize());
  float ms = 0.f;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  std::vector<long long> cyc(256);
  unsigned flag = 0;
  CHECK(hipMemcpy(cyc.data(), a.cyc, 256 * sizeof(long long), hipMemcpyDeviceToHost));
  CHECK(hipMemcpy(&flag, a.flag, sizeof(flag), hipMemcpyDeviceToHost));
  long long mx = 0;
  int n = 0;
  for (int b = 0; b < 256; ++b) { const int slot = b >> 3; if (slot / KG < groups_per_xcd) { mx = cyc[b] > mx ? cyc[b] : mx; ++n; } }
  // checksum of both state buffers: the two barrier forms must agree (a stale L1 line would show here)
  double sum = 0.0;
  {
    const size_t nfl = (size_t)2 * groups * (S / 16) * KG * 256;
    std::vector<float> hh(nfl);
    CHECK(hipMemcpy(hh.data(), a.hbuf, nfl * 4, hipMemcpyDeviceToHost));
    for (size_t i = 0; i < nfl; ++i) sum += (double)hh[i] * (double)((i % 97) + 1);
  }
  const double per = (double)mx / steps;
  const double mfma_bound = 3.0 * (S / 16) * KG * 4 * 32 / 4.0;       // MFMAs per CU and step x 32 cycles / 4 SIMDs
  const double lds_bytes = 3.0 * KG * 1024 * ((S / 16 >= 8) ? 8.0 : (double)(S / 16));   // A operand: read once per wave (>= 8 tiles) or once per tile
  printf("%-34s H=%4d S/group=%4d groups=%3d (B=%5d) CUs=%3d  cycles/step %9.0f  (MFMA bound %8.0f, frac %.3f)  us/step(event) %7.2f  LDS B/clk/CU %5.1f  checksum %.6e%s\n",
         what, H, S, groups, groups * S, n, per, mfma_bound, mfma_bound / per, ms * 1e3 / steps, lds_bytes / per, sum, flag ? "  ** BARRIER TIMEOUT **" : "");
}

template <int H, int MODE, int LIGHT = 0>
static void run(const char* what, Args a, int S, int steps, int groups_per_xcd) {
  if (S == 512) run_t<H, MODE, 4, LIGHT>(what, a, S, steps, groups_per_xcd);
  else if (S == 256) run_t<H, MODE, 2, LIGHT>(what, a, S, steps, groups_per_xcd);
  else run_t<H, MODE, 1, LIGHT>(what, a, S, steps, groups_per_xcd);
}

int main() {
  Args a;
  CHECK(hipMalloc(&a.hbuf, (size_t)64 << 20));      // 2 buffers x 16 groups x 32 sequence tiles x 32 k groups x 1 KB = 32 MB at most
  float* w;
  CHECK(hipMalloc(&w, (size_t)64 * 3 * 64 * 256 * 4));
  {
    std::vector<float> hw((size_t)64 * 3 * 64 * 256);
    for (size_t i = 0; i < hw.size(); ++i) hw[i] = (float)((i * 2654435761u) >> 20 & 1023) / 1024.f - 0.5f;
    CHECK(hipMemcpy(w, hw.data(), hw.size() * 4, hipMemcpyHostToDevice));
  }
  a.w = w;
  CHECK(hipMalloc(&a.ctr, 64 * sizeof(unsigned)));
  CHECK(hipMalloc(&a.flag, sizeof(unsigned)));
  CHECK(hipMalloc(&a.cyc, 256 * sizeof(long long)));
  CHECK(hipMalloc(&a.xcc, 256 * sizeof(unsigned)));

  // ---- (1) the barrier alone: 32 workgroups per XCD, one per CU
  for (int light = 0; light < 3; ++light) {
    const int steps = 2000;
    a.steps = steps;
    CHECK(hipMemset(a.ctr, 0, 64 * sizeof(unsigned)));
    CHECK(hipMemset(a.flag, 0, sizeof(unsigned)));
    auto kern = light == 1 ? barrier_only<1> : light == 2 ? barrier_only<2> : barrier_only<0>;
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 90 * 1024));
    hipLaunchKernelGGL(kern, dim3(256), dim3(512), 90 * 1024, 0, a);
    CHECK(hipDeviceSynchronize());
    std::vector<long long> cyc(256);
    std::vector<unsigned> xcc(256);
    unsigned flag = 0;
    CHECK(hipMemcpy(cyc.data(), a.cyc, 256 * sizeof(long long), hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(xcc.data(), a.xcc, 256 * sizeof(unsigned), hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(&flag, a.flag, sizeof(flag), hipMemcpyDeviceToHost));
    int same = 0;
    for (int b = 0; b < 256; ++b) same += (xcc[b] == xcc[b & 7]);
    long long mx = 0;
    for (int b = 0; b < 256; ++b) mx = cyc[b] > mx ? cyc[b] : mx;
    if (!light)
      printf("XCC_ID of blocks 0..7: %u %u %u %u %u %u %u %u; blocks with XCC_ID(b) == XCC_ID(b %% 8): %d of 256\n", xcc[0], xcc[1],
             xcc[2], xcc[3], xcc[4], xcc[5], xcc[6], xcc[7], same);
    printf("group barrier alone (32 workgroups of one XCD), %s: %.0f cycles per barrier%s\n",
           light == 1 ? "relaxed counter + sc1 poll, no fences (data form: sc1 stores / sc1 loads)" : light == 2 ? "lane-0 release fence, relaxed counter, lane-0 acquire fence" : "acq/rel atomics + __threadfence by every thread", (double)mx / steps, flag ? "  ** TIMEOUT **" : "");
  }
  // ---- (2) the product loop
  const int steps = 300;
  for (int S : {16, 64, 256, 512}) {
    run<512, 0>("hidden-part., fences by all threads", a, S, steps, 1);
    run<512, 0, 1>("hidden-part., sc1 stores+loads", a, S, steps, 1);
    run<512, 0, 2>("hidden-part., lane-0 fences", a, S, steps, 1);
    run<512, 1>("  without the barrier", a, S, steps, 1);
    run<512, 2>("  without barrier and exchange", a, S, steps, 1);
  }
  // hidden 256 (BASELINE configs[0]): 16 CUs per group, two groups per XCD; B = 64 is 4 groups of 16
  for (int S : {16, 64, 256}) {
    run<256, 0>("hidden-part., fences by all threads", a, S, steps, 2);
    run<256, 0, 1>("hidden-part., sc1 stores+loads", a, S, steps, 2);
    run<256, 0, 2>("hidden-part., lane-0 fences", a, S, steps, 2);
    run<256, 2>("  without barrier and exchange", a, S, steps, 2);
  }
  return 0;
}
