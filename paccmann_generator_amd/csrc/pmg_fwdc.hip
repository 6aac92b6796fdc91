// Clustered ("hidden-partitioned") StackGRU forward for SMALL batches: sampling decode and
// teacher-forced log-probs, same arithmetic and same outputs / history as pmg_fwd.hip.
//
// pmg_fwd.hip gives 16 sequences to ONE workgroup for all time steps: no synchronisation, but at the
// batch sizes the reference ships (64-256: examples/IC50/example_params.json:16,
// examples/affinity/conditional_generator.json:12, BASELINE configs[0]) that is 4-16 of 256 CUs, each
// streaming all of W_hh from L2 every step -- pure latency (24 us per step at hidden 256).  Here the
// 16 sequences of a row group belong to a CLUSTER of C = H/16 workgroups (one per CU): member c keeps
// the 48 gate rows of hidden units 16c..16c+15 (W_hh and the stack-top columns of W_ih) resident in LDS
// and forms those units of the new hidden state for all 16 sequences; member s also owns sequence s
// for everything that is per sequence (extras product, log-softmax / sampling, stack update).  Two
// exchanges per step go through global memory, each closed by a cluster barrier:
//   owners : h(all units) -> controls, push value, logits -> token, new stack, stack top   | barrier A
//   members: tokens, stack tops, h -> gates of the own 16 units -> new h slice             | barrier B
// Hand-off form (MI355X_MICROARCH.md, "inter-workgroup visibility", third row of the table of
// fence-free forms; measured in tests/microbench/hidden_part.hip): EVERY exchanged byte is written by
// sc1 (write-through) stores in whole 128-B lines by one store instruction of one wave and read by sc1
// loads; every storing wave waits vmcnt(0), then a workgroup barrier, then lane 0 adds to the
// cluster's counter (agent-scope atomic) and polls it with sc1 loads; a workgroup barrier separates
// the poll from every load.  No fence, no L2 write-back.  Correctness does not depend on where the
// members run; speed does (blockIdx % 8 is the XCD when the kernel has the chip to itself, so a
// cluster shares one L2).
//
// Residency: the barrier needs every member of a cluster resident at once.  The grid is at most 256
// workgroups of more than half a CU's LDS, so that holds whenever no OTHER spinning kernel holds CUs;
// two clustered launches running concurrently on different streams could starve each other, so a
// caller that overlaps launches passes PMG_FLAG_NO_CLUSTER (ops.TrainSlice does).  Every poll loop is
// bounded: after ~1 s a member gives up, raises the status word, and every member of every cluster
// leaves; the outputs are then poisoned (len_out = -1, logp_out = NaN), never silently wrong.
#include "pmg_kernels.h"

namespace {

#define FC_THREADS 512
#define FC_SPIN_LIMIT 3000000   // polls of >= ~0.3 us each

typedef unsigned fc_u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x4 ld4(const float* p) {
  const float4 v = *reinterpret_cast<const float4*>(p);
  f32x4 r; r.x = v.x; r.y = v.y; r.z = v.z; r.w = v.w; return r;
}
__device__ __forceinline__ void st4(float* p, const f32x4 v) {
  *reinterpret_cast<float4*>(p) = make_float4(v.x, v.y, v.z, v.w);
}
// exchanged data: sc1 on every access (aux = 16)
__device__ __forceinline__ float4 xld4(const __amdgpu_buffer_rsrc_t r, unsigned voff) {
  const fc_u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, voff, 0, 16);
  return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}
__device__ __forceinline__ float xld1(const __amdgpu_buffer_rsrc_t r, unsigned voff) {
  return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, voff, 0, 16));
}
__device__ __forceinline__ void xst4(const __amdgpu_buffer_rsrc_t r, unsigned voff, const f32x4 h) {
  fc_u32x4 u;
  u.x = __float_as_uint(h[0]); u.y = __float_as_uint(h[1]); u.z = __float_as_uint(h[2]); u.w = __float_as_uint(h[3]);
  __builtin_amdgcn_raw_buffer_store_b128(u, r, voff, 0, 16);
}
__device__ __forceinline__ void xst1(const __amdgpu_buffer_rsrc_t r, unsigned voff, const float v) {
  __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), r, voff, 0, 16);
}

__device__ __forceinline__ float fast_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float fast_tanh(float x) {
  const float e = __expf(-2.0f * fabsf(x));
  const float t = (1.0f - e) * __builtin_amdgcn_rcpf(1.0f + e);
  return copysignf(t, x);
}

// Sum over the 16 lanes of a DPP row (every lane gets the total): four row rotations on the VALU
// instead of four ds_bpermute round trips.
__device__ __forceinline__ float row16_sum(float x) {
  x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x128, 0xf, 0xf, false));  // row_ror:8
  x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x124, 0xf, 0xf, false));  // row_ror:4
  x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x122, 0xf, 0xf, false));  // row_ror:2
  x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x121, 0xf, 0xf, false));  // row_ror:1
  return x;
}

// Reductions over the 64 lanes of a wave on the DPP network: xor 1, xor 2 (quad permutes), mirror within 8 and within
// 16 lanes, then row_bcast:15 / row_bcast:31 carry the row results up; lane 63 holds the result, read into a scalar.
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
// sum in the order the network gives (log-softmax of the teacher-forced pass: tolerance, not bit pattern)
__device__ __forceinline__ float wave_sum63(float x) {
  x += FC_DPP(x, 0xB1, 0xf); x += FC_DPP(x, 0x4E, 0xf);
  x += FC_DPP(x, 0x141, 0xf); x += FC_DPP(x, 0x140, 0xf);
  const float r15 = FC_DPP(x, 0x142, 0xa);       // rows 1, 3: the row below's total (rows 0, 2: own value, dropped)
  x = ((threadIdx.x & 16) ? x + r15 : x);
  const float r31 = FC_DPP(x, 0x143, 0xc);       // rows 2, 3: total of rows 0-1
  x = ((threadIdx.x & 32) ? x + r31 : x);
  return readlane_f(x, 63);
}
// index of the largest value, the lowest index among equal values
__device__ __forceinline__ int wave_argmax(float bv, int bi) {
#define FC_AM(ctrl, rmask)                                                          \
  {                                                                                 \
    const float ov = FC_DPP(bv, ctrl, rmask); const int oi = FC_DPPI(bi, ctrl, rmask); \
    if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }                     \
  }
  FC_AM(0xB1, 0xf) FC_AM(0x4E, 0xf) FC_AM(0x141, 0xf) FC_AM(0x140, 0xf) FC_AM(0x142, 0xa) FC_AM(0x143, 0xc)
#undef FC_AM
  return __builtin_amdgcn_readlane(bi, 63);
}

// Cluster barrier in two halves, so that loads which do not depend on the exchange (the owners' extras
// weights) can be issued between them and land while lane 0 polls.  arrive: every wave's sc1 stores
// written through, workgroup barrier, lane 0 adds to the counter.  wait: lane 0 polls (bounded),
// workgroup barrier; returns false (workgroup-uniform) when it gave up.  `okw`: one LDS word.
__device__ __forceinline__ void cluster_arrive(unsigned* ctr) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ bool cluster_wait(unsigned* ctr, const unsigned target, unsigned* status, int* okw) {
  if (threadIdx.x == 0) {
    int ok = 1;
    int spins = 0;
    while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      if (++spins > FC_SPIN_LIMIT || __hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) {
        __hip_atomic_store(status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ok = 0;
        break;
      }
      __builtin_amdgcn_s_sleep(2);
    }
    *okw = ok;
  }
  __syncthreads();
  return *okw != 0;
}

// LDS map (float offsets), shared by the kernel and the host-side size check.  NG = row groups per
// cluster, NOWN = sequences a member may own, DW = stack floats per sequence.
struct FcLds { unsigned red, wlz, bxl, xexs, hls, hl, ghl, stk, ptv, osm, misc, gtl; };
// the z tile of the own units stays in LDS at hidden 512 (registers: r and n tiles only)
__host__ __device__ constexpr bool fc_zreg(const int H) { return H < 512; }
__host__ __device__ inline FcLds fc_lds(const int H, const int NG, const int NOWN, const int DW) {
  FcLds o;
  unsigned p = 0;
  o.red = p; p += 2 * 8 * 2 * 256;          // [2][8 waves][2][256] K-split partials, double-buffered over the groups
  o.wlz = p; p += fc_zreg(H) ? 0u : (unsigned)(H / 16) * 256u;   // [KG][64][4] z tile (hidden 512 only)
  o.bxl = p; p += 256;                      // [256] extras bias (ctrl | push | out)
  o.xexs = p; p += (unsigned)NOWN * 256;    // [NOWN][256] extras of the owned sequences
  o.hls = p; p += (unsigned)NOWN * H;       // [NOWN][H] hidden states of the owned sequences
  o.hl = p; p += (unsigned)NG * 256;        // [NG][64][4] own units of the current hidden state (D layout, wave 0)
  o.ghl = p; p += 4 * 256;                  // [4][64][4] r | z | n | W_hn h of the last group (history, deferred; wave 0)
  o.stk = p; p += (unsigned)NOWN * 2 * DW;  // [NOWN][2][D*W] own sequences' stacks (ping-pong)
  o.ptv = p; p += (unsigned)NOWN * 128;     // [NOWN][2][64] push value | new stack top (history, deferred)
  o.osm = p; p += (unsigned)NOWN * 8;       // [NOWN][8] push, pop, no-op | uniform | token, finished, length
  o.misc = p; p += 64;                      // [0] barrier ok, [1..] scratch
  o.gtl = (p + 3u) & ~3u;                   // [V][3][16] token-table lines of the own units (if they fit)
  return o;
}

// MODE: PMG_FWD_DECODE or PMG_FWD_TEACHER.  One layer.  HM = H / 64: float4 per lane and extras row.
//
// A cluster serves NG consecutive row groups (NG = 1 while there are at least as many clusters as row
// groups; above that each step's two barriers are shared by NG groups): sequence sigma = 16 r + s of
// the cluster (group r, row s) is OWNED by member sigma % C, so a member owns up to NOWN sequences and
// runs their per-sequence work one after the other; the gates of the own units are formed group by group.
// The gate rows of the own units live in REGISTERS (each wave keeps the K quarter of the gate tiles
// it multiplies; at hidden 512 the z tile stays in LDS), so the LDS is left to the owned sequences' stacks.
template <int MODE, int HM>
__global__ __launch_bounds__(FC_THREADS, 2) void pmg_fwdc_kernel(const FwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr bool teacher = (MODE == PMG_FWD_TEACHER);
  constexpr bool decode = (MODE == PMG_FWD_DECODE);
  constexpr int H = 64 * HM, KG = H / 16, C = KG;   // members per cluster = unit tiles
  const int W = a.W, D = a.D, V = a.V, B = a.B, G = a.G, N = a.N, NX = a.NX;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int b = lane & 15, q = lane >> 4;
  // workgroup -> (cluster, member): clusters are dealt over the XCDs first
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int csub = slot / C, cu = slot - csub * C;
  const int cl = csub * 8 + xcd;
  const int NG = a.cluster_ng;
  const int g0 = cl * NG;                        // first row group of this cluster
  if (g0 >= G) return;                           // a whole cluster absent: nobody waits for it
  const int NGc = (G - g0 < NG) ? G - g0 : NG;   // row groups of this cluster
  const int NS = 16 * NGc;                       // its sequences: sigma = 16 r + s, global row = 16 g0 + sigma
  const int NOWN = a.cluster_duo ? NG : (16 * NG + C - 1) / C;   // owned sequences the LDS is laid out for
  // Split ownership ("duo"): with one row group on a cluster of 32 members only 16 of them would own a sequence and each
  // would stream the second half of the extras weights from L2 every step.  Instead member s keeps the control and push
  // rows and does the stack side of sequence s, member s + 16 keeps the vocabulary rows and does its logits / sampling
  // side: each half of the weights stays in registers for good, the two sides run in parallel, and they publish
  // different lines of the sequence's record, so no exchange is added.
  // With several row groups member s (s + 16) does its side for sequence s of every group: sigma = s + 16 k.
  const bool duo = a.cluster_duo != 0;             // hidden 512, 3+W <= 64, V <= 64, NG stacks fit the LDS (host)
  const int nown = duo ? NGc : (cu < NS) ? (NS - cu + C - 1) / C : 0;   // owned by this member: sigma = sq0 + k sqs
  const int sq0 = duo ? (cu & 15) : cu, sqs = duo ? 16 : C;
  const bool roleA = !duo || cu < 16;              // stack side: controls, push value, stack update, its history
  const bool roleB = !duo || cu >= 16;             // vocabulary side: logits -> log-prob / token, its history
  const int rbase = (duo && !roleA) ? 3 + W : 0, rend = (duo && roleA) ? 3 + W : NX;   // extras rows of this member
  const int row00 = g0 * PMG_ROWS;
  const int DW = D * W;

  // ---- LDS
  constexpr bool ZREG = fc_zreg(H);
  const FcLds lo = fc_lds(H, NG, NOWN, DW);
  float* fs = reinterpret_cast<float*>(smem);
  float* red = fs + lo.red;
  float* wlz = fs + lo.wlz;
  float* bxl = fs + lo.bxl;
  float* xexs = fs + lo.xexs;
  float* hls = fs + lo.hls;
  float* hl = fs + lo.hl;
  float* ghl = fs + lo.ghl;
  float* stk = fs + lo.stk;
  float* ptv = fs + lo.ptv;
  float* osm = fs + lo.osm;
  int* osi = reinterpret_cast<int*>(osm);
  int* misc = reinterpret_cast<int*>(fs + lo.misc);
  float* gtl = fs + lo.gtl;
  const bool gt_lds = a.cluster_gtab_lds != 0;
  const bool ht = a.cluster_hist_t != 0;         // stack / push / probability history per sequence (FwdArgs::LAYW)

  // ---- exchange buffers of this cluster: two hidden-state buffers [NG][KG][64][4], records [16 NG][128]
  const size_t hbf = (size_t)NG * H * 16;
  const size_t cxs = 2 * hbf + (size_t)NG * 16 * 128;       // floats per cluster
  float* cx = a.CX + (size_t)cl * cxs;
  const __amdgpu_buffer_rsrc_t hb0 = pmg_rsrc(cx, (unsigned)(hbf * 4)), hb1 = pmg_rsrc(cx + hbf, (unsigned)(hbf * 4));
  const __amdgpu_buffer_rsrc_t rec = pmg_rsrc(cx + 2 * hbf, (unsigned)NG * 16u * 512u);   // per sequence: top[64] | token, finished
  unsigned* ctr = a.CXS + 32 + (size_t)cl * 32;             // counter of this cluster (own 128-B line)
  unsigned* status = a.CXS;                                  // one word for the launch
  unsigned bars = 0;
#define FC_ARRIVE() (++bars, cluster_arrive(ctr))
#define FC_WAIT() cluster_wait(ctr, (unsigned)C * bars, status, misc)
#ifdef PMG_PROFILE
  // -DPMG_PROFILE build + PMG_DEBUG bit 0x8000: per-phase cycle counters of member 0 of cluster 0
  const bool prof = (a.dbg & 0x8000) != 0 && cl == 0 && cu == 0;
  long long tph[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  long long tlast = prof ? (long long)__builtin_amdgcn_s_memtime() : 0;
#define FC_TICK(k) if (prof) { const long long tn_ = (long long)__builtin_amdgcn_s_memtime(); tph[k] += tn_ - tlast; tlast = tn_; }
#else
#define FC_TICK(k)
#endif

  // ---- resident weights (registers): the packed forward operands of pmg_pack.hip, tiles 3*cu .. 3*cu+2.
  // K split: waves 0-3 carry the r and z tiles of K quarter kw, waves 4-7 the n tile.
  const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
  const int kw = wave & 3;
  const int kq0 = kw * HM;                       // KG / 4 = HM reduction groups per wave
  f32x4 wres[ZREG ? 2 * HM : HM], wis0, wis1;
  {
    const int ga = (wave < 4) ? 0 : 2;
    const float* src = a.wbase + a.oWHH[0] / 4 + ((size_t)(3 * cu + ga) * KG + kq0) * 256 + lane * 4;
    const float* srcz = a.wbase + a.oWHH[0] / 4 + ((size_t)(3 * cu + 1) * KG + kq0) * 256 + lane * 4;
#pragma unroll
    for (int m = 0; m < HM; ++m) {
      wres[m] = ld4(src + (size_t)m * 256);
      if constexpr (ZREG) wres[HM + m] = (wave < 4) ? ld4(srcz + (size_t)m * 256) : z4;
    }
    if constexpr (!ZREG) {
      const float* sz = a.wbase + a.oWHH[0] / 4 + (size_t)(3 * cu + 1) * KG * 256;
      for (int i = tid * 4; i < KG * 256; i += FC_THREADS * 4) st4(wlz + i, ld4(sz + i));
    }
    wis0 = ld4(a.wbase + a.oWIH[0] / 4 + ((size_t)(3 * cu + ga) * 4 + kw) * 256 + lane * 4);
    wis1 = (wave < 4) ? ld4(a.wbase + a.oWIH[0] / 4 + ((size_t)(3 * cu + 1) * 4 + kw) * 256 + lane * 4) : z4;
  }
  for (int i = tid; i < NOWN * 2 * DW; i += FC_THREADS) stk[i] = 0.0f;
  if (tid < 64) misc[tid] = 0;
  for (int i = tid; i < NOWN * 8; i += FC_THREADS) {
    if ((i & 7) == 6) osi[i] = N; else osm[i] = 0.0f;        // length N, not finished
  }
  if (gt_lds)
    for (int idx = tid; idx < V * 48; idx += FC_THREADS) {
      const int v = idx / 48, rem = idx - v * 48;
      gtl[idx] = a.GTAB[(size_t)v * 3 * H + (size_t)(rem >> 4) * H + 16 * cu + (rem & 15)];
    }
  if (tid < 256) bxl[tid] = (tid < 3) ? a.b_ctrl[tid] : (tid < 3 + W) ? a.b_push[tid - 3] : (tid < NX) ? a.b_out[tid - 3 - W] : 0.0f;

  // ---- owners' extras product [W_ctrl; W_push; W_out] h: 16 lanes per row, 32 rows at a time; the
  // weights of the first RC rows of each 16-lane group live in registers and are (re)loaded between
  // the two halves of barrier B, i.e. while the new hidden state is still on its way
  constexpr int RC = 16 / HM;                    // rows per register chunk (RC * HM <= 16 float4)
  const int l16 = tid & 15, rgrp = tid >> 4;
  const int RX = (rend - rbase + 31) >> 5;       // rows per 16-lane group
  const int NCH = (RX + RC - 1) / RC;            // chunks (1 at hidden <= 256 and NX <= 128)
  f32x4 wreg[RC * HM];
  auto load_chunk = [&](const int c) {
#pragma unroll
    for (int r = 0; r < RC; ++r) {
      const int n = rbase + rgrp + 32 * (c * RC + r);
      const float* wr = (n < 3) ? a.w_ctrl + (size_t)n * H
                                : (n < 3 + W) ? a.w_push + (size_t)(n - 3) * H : a.w_out + (size_t)(n - 3 - W) * H;
#pragma unroll
      for (int m = 0; m < HM; ++m)
        wreg[r * HM + m] = (n < rend) ? ld4(wr + 4 * l16 + 64 * m) : z4;
    }
  };

  // ---- h0 of the own units for the 16 sequences of every group (D layout: lane (b, q) holds units 4q..4q+3 of row b)
  if (wave == 0) {
    for (int r = 0; r < NGc; ++r) {
      const int row = row00 + 16 * r + b;
      f32x4 hown = z4;
      if (decode) {
        // latent_to_hidden tile cu: packed [tile][Z/16][64][4] (PK_L2H), B operand straight from the latents
        const float* wz = a.wbase + a.oL2H / 4 + (size_t)cu * (a.Z / 16) * 256 + lane * 4;
        f32x4 acc = ld4(a.b_l2h + 16 * cu + 4 * q), acc2 = z4;
        for (int kg = 0; kg < a.Z / 16; kg += 2) {
          const f32x4 w0 = ld4(wz + (size_t)kg * 256), w1 = ld4(wz + (size_t)(kg + 1) * 256);
          f32x4 x0 = z4, x1 = z4;
          if (row < B) {
            x0 = ld4(a.latent + (size_t)row * a.Z + 16 * kg + 4 * q);
            x1 = ld4(a.latent + (size_t)row * a.Z + 16 * (kg + 1) + 4 * q);
          }
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w0[c], x0[c], acc, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(w1[c], x1[c], acc2, 0, 0, 0);
          }
        }
        hown = acc + acc2;
      } else if (row < B) {
        hown = ld4(a.h0 + (size_t)row * H + 16 * cu + 4 * q);
      }
      st4(hl + r * 256 + lane * 4, hown);
      xst4(hb0, (unsigned)((r * KG + cu) * 256 + lane * 4) * 4u, hown);
      if (teacher) {
#pragma unroll
        for (int j = 0; j < 4; ++j) a.HH[0][((size_t)(g0 + r) * H + 16 * cu + 4 * q + j) * 16 + b] = hown[j];
      }
    }
  }
  FC_ARRIVE();
  if (nown > 0) load_chunk(0);
  bool ok = FC_WAIT();                            // h0 of every member visible
  // deferred stack update: thread -> (row offset sd0, column sw), SDS rows apart (no division in the loop)
  const int SDS = FC_THREADS / W;                 // W <= 64: at least 8 rows in flight
  const int sd0 = tid / W, sw = (sd0 < SDS) ? tid - sd0 * W : W;
  const f32x4 bhn = (wave == 0) ? ld4(a.BG[0] + 3 * H + 16 * cu + 4 * q) : z4;

  // hidden part of the gate pre-activations of this wave's K quarter: waves 0-3 r -> p0, z -> p1; waves 4-7 W_hn h -> p1
  auto hidden_part = [&](const float4 (&x4)[HM], f32x4& p0, f32x4& p1) __attribute__((always_inline)) {
    if (wave < 4) {
#pragma unroll
      for (int m = 0; m < HM; ++m) {
        const float4 x = x4[m];
        f32x4 wz;
        if constexpr (ZREG) wz = wres[HM + m]; else wz = ld4(wlz + (size_t)(kq0 + m) * 256 + lane * 4);
        p0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wres[m][0], x.x, p0, 0, 0, 0);
        p1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wz[0], x.x, p1, 0, 0, 0);
        p0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wres[m][1], x.y, p0, 0, 0, 0);
        p1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wz[1], x.y, p1, 0, 0, 0);
        p0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wres[m][2], x.z, p0, 0, 0, 0);
        p1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wz[2], x.z, p1, 0, 0, 0);
        p0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wres[m][3], x.w, p0, 0, 0, 0);
        p1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wz[3], x.w, p1, 0, 0, 0);
      }
    } else {
#pragma unroll
      for (int m = 0; m < HM; ++m) {
        const float4 x = x4[m];
        p1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wres[m][0], x.x, p1, 0, 0, 0);
        p1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wres[m][1], x.y, p1, 0, 0, 0);
        p1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wres[m][2], x.z, p1, 0, 0, 0);
        p1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wres[m][3], x.w, p1, 0, 0, 0);
      }
    }
  };
  FC_TICK(7)
  for (int i = 0; ok; ++i) {
    const __amdgpu_buffer_rsrc_t hin = (i & 1) ? hb1 : hb0, hout = (i & 1) ? hb0 : hb1;
    // ================================================================ owners: per-sequence work
    // (1) hidden states of the owned sequences -> LDS (unit 16*kg + 4*kq + c sits in lane s + 16*kq of block kg)
    for (int t = tid; t < nown * (H / 4); t += FC_THREADS) {
      const int k = t / (H / 4), u4 = t - k * (H / 4);
      const int sg = sq0 + k * sqs, r = sg >> 4, s = sg & 15;
      const float4 v = xld4(hin, (unsigned)(((r * KG + (u4 >> 2)) * 64 + s + 16 * (u4 & 3)) * 4) * 4u);
      *reinterpret_cast<float4*>(hls + (size_t)k * H + 4 * u4) = v;
    }
    __syncthreads();
    FC_TICK(0)
    // (2) extras of every owned sequence: one pass over the weights (registers; later chunks from L2) for all of them
    for (int c = 0; c < NCH; ++c) {
      if (c > 0) load_chunk(c);
      for (int k = 0; k < nown; ++k) {
        f32x4 hv[HM];
#pragma unroll
        for (int m = 0; m < HM; ++m) hv[m] = ld4(hls + (size_t)k * H + 4 * l16 + 64 * m);
#pragma unroll
        for (int rr = 0; rr < RC; ++rr) {
          const int n = rbase + rgrp + 32 * (c * RC + rr);
          float acc = 0.0f;
#pragma unroll
          for (int m = 0; m < HM; ++m) {
            acc = fmaf(wreg[rr * HM + m][0], hv[m][0], acc); acc = fmaf(wreg[rr * HM + m][1], hv[m][1], acc);
            acc = fmaf(wreg[rr * HM + m][2], hv[m][2], acc); acc = fmaf(wreg[rr * HM + m][3], hv[m][3], acc);
          }
          acc = row16_sum(acc);
          if (l16 == 0 && n < rend) xexs[k * 256 + n] = acc + bxl[n];
        }
      }
    }
    __syncthreads();
    FC_TICK(1)
    // (3) one WAVE per owned sequence from here to the record: logits of step i-1 -> token, stack controls,
    // row 0 of the new stack, publish.  Nothing below needs a workgroup barrier.
    for (int k = wave; k < nown; k += 8) {
      const int sg = sq0 + k * sqs;          // wave-uniform: sequence of the cluster,
      const int r = sg >> 4, s = sg & 15;         // its row group and row,
      const int g = g0 + r;                       // the global row group
      const int row = row00 + sg;
      const bool rowok = row < B;
      const float* xex = xexs + k * 256;
      const float* sold = stk + ((size_t)k * 2 + (i & 1)) * DW;
      float* snew = stk + ((size_t)k * 2 + ((i + 1) & 1)) * DW;
      // ---- consume the logits of step i-1: the whole wave, entry v in lane v & 63; reductions on the DPP network
      // (no LDS round trips), results read out of lane 63 into a scalar
      int tok = teacher ? ((rowok && i < N + 1) ? min(max((int)a.tok_in[(size_t)min(i, N) * B + row], 0), V - 1) : 0) : a.start_tok;
      int finflag = rowok ? 0 : 1;
      if (i >= 1 && roleB) {
        const float* lg = xex + 3 + W;
        if (teacher) {
          float m = -INFINITY;
#pragma unroll 1
          for (int v = lane; v < V; v += 64) m = fmaxf(m, lg[v]);
          m = wave_max(m);
          float total = 0.0f;
#pragma unroll 1
          for (int v = lane; v < V; v += 64) total += __expf(lg[v] - m);
          total = wave_sum63(total);
          const int y = tok;       // tokens[i][row]: the target of step i-1 and the input of step i
          if ((y & 63) == lane) {
            const float lp = (lg[y] - m) - __logf(total);
            a.LP[((size_t)(i - 1) * G + g) * 16 + s] = lp;
            if (rowok) a.logp_out[(size_t)(i - 1) * B + row] = lp;
          }
          float* pr = a.PR + ((size_t)(i - 1) * G + g) * a.VP * 16;
          const float inv = __builtin_amdgcn_rcpf(total);
#pragma unroll 2
          for (int v = lane; v < V; v += 64) pr[ht ? s * a.VP + v : v * 16 + s] = __expf(lg[v] - m) * inv;
        } else {
          bool fin = osi[k * 8 + 5] != 0;
          const bool greedy = !(a.temperature > 0.0f);
          if (greedy) {
            float bv = -INFINITY; int bi = 0x7fffffff;
            for (int v = lane; v < V; v += 64) { const float x = lg[v]; if (x > bv) { bv = x; bi = v; } }
            tok = wave_argmax(bv, bi);            // largest value, lowest index among equals
          } else {
            // same slot-wise inverse-CDF draw as pmg_fwd.hip (oracle twin: oracle/philox.py sample_row): 32-entry
            // slots, Hillis-Steele scan inside a slot, running carry across slots.  Two slots ride in the two halves
            // of the wave; every floating-point operation and its order are those of the one-slot-at-a-time form.
            const float T = a.temperature;
            const int l5 = lane & 31, hs = lane >> 5;
            float m = -INFINITY;
            for (int v = lane; v < V; v += 64) m = fmaxf(m, lg[v] / T);
            m = wave_max(m);
            const float u = osm[k * 8 + 3];       // pmg_uniform(seed, global row, offset + i - 1), formed under barrier B
            float cdf[4];                          // pair p holds entries 64p..64p+63 (3+W+V <= 256 on this path)
            float carry = 0.0f;
#pragma unroll
            for (int pr2 = 0; pr2 < 4; ++pr2) {
              cdf[pr2] = 0.0f;
              if (64 * pr2 < V) {
                const int v = 64 * pr2 + lane;
                float e = (v < V) ? __expf(lg[v] / T - m) : 0.0f;
#pragma unroll
                for (int o = 1; o <= 16; o <<= 1) {
                  const float t = __shfl_up(e, o, 32);
                  if (l5 >= o) e += t;
                }
                const float c0 = carry + readlane_f(e, 31);     // carry behind the even slot
                cdf[pr2] = (hs == 0 ? carry : c0) + e;
                // an odd slot past the end adds zeros: the carry stays what the even slot left, as when it is skipped
                carry = (32 * (2 * pr2 + 1) < V) ? readlane_f(cdf[pr2], 63) : c0;
              }
            }
            const float target = u * carry;
            tok = -1;
#pragma unroll
            for (int pr2 = 0; pr2 < 4; ++pr2) {
              if (64 * pr2 < V) {
                const int v = 64 * pr2 + lane;
                const unsigned long long mk = __ballot((v < V) && (target < cdf[pr2]));
                if (tok < 0 && mk) tok = 64 * pr2 + (__ffsll((long long)mk) - 1);
              }
            }
            if (tok < 0) tok = V - 1;
          }
          if (rowok && lane == 0) a.tok_out[(size_t)row * N + (i - 1)] = tok;
          if (!fin && tok == a.stop_tok) {
            fin = true;
            if (lane == 0) { osi[k * 8 + 5] = 1; osi[k * 8 + 6] = i - 1; }
          }
          if (fin) finflag = 1;
        }
      }
      FC_TICK(2)
      if (i < N && roleB) {
        const float meta = (lane == 0) ? __int_as_float(tok) : (lane == 1) ? __int_as_float(finflag) : 0.0f;
        xst1(rec, (unsigned)(sg * 128 + 64 + lane) * 4u, meta);
      }
      if (i < N && roleA) {
        // ---- stack controls; ROW 0 of the new stack (= the stack top the gates need) and the record
        // go out first: the rest of the update and every history store wait until this member has
        // arrived at barrier A (below), off the chain the other members wait on
        const float c0 = xex[0], c1 = xex[1], c2 = xex[2];
        const float m = fmaxf(c0, fmaxf(c1, c2));
        const float e0 = __expf(c0 - m), e1 = __expf(c1 - m), e2 = __expf(c2 - m);
        const float inv = 1.0f / ((e0 + e1) + e2);
        const float apush = e0 * inv, apop = e1 * inv, anoop = e2 * inv;
        float top = 0.0f, pv = 0.0f;
        if (lane < W) {
          pv = fast_tanh(xex[3 + lane]);
          const float nxt = (1 < D) ? sold[W + lane] : 0.0f;
          top = anoop * sold[lane] + apush * pv + apop * nxt;
          snew[lane] = top;
        }
        xst1(rec, (unsigned)(sg * 128 + lane) * 4u, top);
        ptv[k * 128 + lane] = pv; ptv[k * 128 + 64 + lane] = top;
        if (lane == 0) { osm[k * 8] = apush; osm[k * 8 + 1] = apop; osm[k * 8 + 2] = anoop; osi[k * 8 + 4] = tok; }
      }
    }
    FC_TICK(3)
    if (i == N) break;
    FC_ARRIVE();
    // ---- under the wait of barrier A.  The hidden part of the gates needs only h, which has been visible since
    // barrier B: the blocks of group 0 are requested now, fly under the deferred owner work and are multiplied before
    // the wait ends; the part that needs the records (stack tops, tokens) follows behind the barrier.
    // (Up to hidden 256; at hidden 512 the product is longer than the wait and the early form measured slower.)
    constexpr bool EARLY = HM <= 4;
    float4 xs[HM];
    if constexpr (EARLY) {
#pragma unroll
      for (int m = 0; m < HM; ++m) xs[m] = xld4(hin, (unsigned)((kq0 + m) * 256 + lane * 4) * 4u);
    }
    // ---- deferred owner work: stack rows >= 1 and the teacher history
    for (int k = 0; k < (roleA ? nown : 0); ++k) {
      const int sg = sq0 + k * sqs;
      const int s = sg & 15, g = g0 + (sg >> 4);
      const bool rowok = row00 + sg < B;
      const float* sold = stk + ((size_t)k * 2 + (i & 1)) * DW;
      float* snew = stk + ((size_t)k * 2 + ((i + 1) & 1)) * DW;
      const float apush = osm[k * 8], apop = osm[k * 8 + 1], anoop = osm[k * 8 + 2];
      // rows that hold data after this update: d <= min(i, D-1) (row d is zero until step d)
      const int dhi = (i < D - 1) ? i : D - 1;
      for (int d = 1 + sd0; d <= dhi; d += SDS) {       // thread (sd0, sw): column sw of rows 1 + sd0 + j*SDS
        if (sw < W) {
          const int idx = d * W + sw;
          const float nxt = (d + 1 < D) ? sold[idx + W] : 0.0f;
          const float nw = anoop * sold[idx] + apush * sold[idx - W] + apop * nxt;
          snew[idx] = nw;
          // history the BPTT reads: only rows that can still reach a stack top (pmg_fwd.hip, "live rows")
          if (teacher && d <= N - 1 - i) a.SB[((size_t)(i + 1) * G + g) * a.SW + (ht ? (size_t)s * DW + idx : (size_t)idx * 16 + s)] = nw;
        }
      }
      if (teacher) {
        if (wave == 0 && lane < W) {
          const float pv = ptv[k * 128 + lane], top = ptv[k * 128 + 64 + lane];
          a.SB[((size_t)(i + 1) * G + g) * a.SW + (ht ? (size_t)s * DW + lane : (size_t)lane * 16 + s)] = top;
          a.PV[((size_t)i * G + g) * 1024 + (ht ? s * 64 + lane : lane * 16 + s)] = pv;
          a.TP[((size_t)i * G + g) * 2048 + lane * 16 + s] = top;
        }
        if (tid >= 128 && tid < 131)
          a.CT[((size_t)i * G + g) * 64 + (tid - 128) * 16 + s] = (tid == 128) ? apush : (tid == 129) ? apop : anoop;
        // one-hot rows of the fed token behind the stack-top rows (weight-gradient B operand, V <= 64)
        if (tid >= 64 && tid < 128)
          a.TP[((size_t)i * G + g) * 2048 + (size_t)tid * 16 + s] = (rowok && osi[k * 8 + 4] == tid - 64) ? 1.0f : 0.0f;
      }
    }
    f32x4 pa0 = z4, pa1 = z4;
    if constexpr (EARLY) hidden_part(xs, pa0, pa1);
    if (!(ok = FC_WAIT())) break;                // A: tokens and stack tops of every sequence visible

    FC_TICK(4)
    // ================================================================ members: gates of the own 16 units, group by group
    int allfin = 1;                                          // wave 1, decode: every sequence so far has finished
    // (Requesting the operands of group r+1 a group ahead was measured slower: the registers it keeps live across the
    // reduction and the cell spill, and the reloads cost more than the round trip -- profiles/r03_cluster_groups.md.)
    float4 xt;
    float tkn = 0.0f;
    int fl = 1;
    for (int r = 0; r < NGc; ++r) {
      xt = xld4(rec, (unsigned)((r * 16 + b) * 128 + 16 * kw + 4 * q) * 4u);                    // stack-top k group kw
      if (wave == 0) tkn = xld1(rec, (unsigned)((r * 16 + b) * 128 + 64) * 4u);                 // token fed to row b at this step
      if (decode && wave == 1 && lane < 16) fl = __float_as_int(xld1(rec, (unsigned)((r * 16 + lane) * 128 + 65) * 4u));
      float* redr = red + (size_t)(r & 1) * 8 * 2 * 256;
      f32x4 acc0 = pa0, acc1 = pa1;
      if (!EARLY || r > 0) {
#pragma unroll
        for (int m = 0; m < HM; ++m) xs[m] = xld4(hin, (unsigned)((r * KG + kq0 + m) * 256 + lane * 4) * 4u);
        acc0 = z4; acc1 = z4;
        hidden_part(xs, acc0, acc1);
      }
      if (wave < 4) {
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wis0[0], xt.x, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wis1[0], xt.x, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wis0[1], xt.y, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wis1[1], xt.y, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wis0[2], xt.z, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wis1[2], xt.z, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wis0[3], xt.w, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wis1[3], xt.w, acc1, 0, 0, 0);
      } else {
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wis0[0], xt.x, acc0, 0, 0, 0);   // acc0 = W_in[:, E:] top
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wis0[1], xt.y, acc0, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wis0[2], xt.z, acc0, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wis0[3], xt.w, acc0, 0, 0, 0);
      }
      const float tkf = tkn;
      // finished flags of the group's 16 sequences: ONE load per lane of wave 1, combined by a ballot
      if (decode && wave == 1) { allfin = (allfin && __all(fl != 0)) ? 1 : 0; fl = 1; }
      st4(redr + (size_t)(wave * 2 + 0) * 256 + lane * 4, acc0);
      st4(redr + (size_t)(wave * 2 + 1) * 256 + lane * 4, acc1);
      __syncthreads();                            // partials of group r visible (the other buffer is being read by nobody)
      if (wave == 0) {
        f32x4 vr = z4, vz = z4, vi = z4, vh = z4;
#pragma unroll
        for (int wv = 0; wv < 4; ++wv) {
          vr += ld4(redr + (size_t)(wv * 2 + 0) * 256 + lane * 4);
          vz += ld4(redr + (size_t)(wv * 2 + 1) * 256 + lane * 4);
          vi += ld4(redr + (size_t)((wv + 4) * 2 + 0) * 256 + lane * 4);
          vh += ld4(redr + (size_t)((wv + 4) * 2 + 1) * 256 + lane * 4);
        }
        const int tk = __float_as_int(tkf);
        if (gt_lds) {
          const float* gt = gtl + (size_t)tk * 48 + 4 * q;
          vr += ld4(gt); vz += ld4(gt + 16); vi += ld4(gt + 32);
        } else {
          const float* gt = a.GTAB + (size_t)tk * 3 * H + 16 * cu + 4 * q;
          vr += ld4(gt); vz += ld4(gt + H); vi += ld4(gt + 2 * H);
        }
        vh += bhn;
        const f32x4 hown = ld4(hl + r * 256 + lane * 4);
        f32x4 rg, zg, ng, hn;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          rg[j] = fast_sigmoid(vr[j]);
          zg[j] = fast_sigmoid(vz[j]);
          ng[j] = fast_tanh(vi[j] + rg[j] * vh[j]);
          hn[j] = (hown[j] - ng[j]) * zg[j] + ng[j];
        }
        st4(hl + r * 256 + lane * 4, hn);
        xst4(hout, (unsigned)((r * KG + cu) * 256 + lane * 4) * 4u, hn);
        if (teacher) {
          if (r + 1 < NGc) {
            // history of the earlier groups goes out right away (it drains under the next group's product)
            float* gp = a.GT[0] + (((size_t)i * G + g0 + r) * KG + cu) * 1024 + lane * 4;
            st4(gp, rg); st4(gp + 256, zg); st4(gp + 512, ng); st4(gp + 768, vh);
            float* hh = a.HH[0] + (((size_t)(i + 1) * G + g0 + r) * H + 16 * cu + 4 * q) * 16 + b;
#pragma unroll
            for (int j = 0; j < 4; ++j) hh[j * 16] = hn[j];
          } else {
            // the last group's goes behind the arrive of barrier B (through LDS: hl holds the new state)
            st4(ghl + lane * 4, rg); st4(ghl + 256 + lane * 4, zg); st4(ghl + 512 + lane * 4, ng); st4(ghl + 768 + lane * 4, vh);
          }
        }
      }
    }
    if (decode && wave == 1 && lane == 0) misc[4] = (allfin && !(a.flags & PMG_FLAG_NO_EARLY_EXIT)) ? 1 : 0;
    FC_TICK(5)
    FC_ARRIVE();
    if (decode && misc[4]) {
      // every row of this cluster has emitted stop_tok: owners fill their rows' tails and all leave
      for (int k = 0; k < (roleB ? nown : 0); ++k) {
        const int row = row00 + sq0 + k * sqs;
        if (row < B)
          for (int t = i + tid; t < N; t += FC_THREADS) a.tok_out[(size_t)row * N + t] = a.stop_tok;
      }
      break;
    }
    if (teacher && wave == 0) {
      const int r = NGc - 1;
      float* gp = a.GT[0] + (((size_t)i * G + g0 + r) * KG + cu) * 1024 + lane * 4;
      st4(gp, ld4(ghl + lane * 4)); st4(gp + 256, ld4(ghl + 256 + lane * 4));
      st4(gp + 512, ld4(ghl + 512 + lane * 4)); st4(gp + 768, ld4(ghl + 768 + lane * 4));
      const f32x4 hn = ld4(hl + r * 256 + lane * 4);
      float* hh = a.HH[0] + (((size_t)(i + 1) * G + g0 + r) * H + 16 * cu + 4 * q) * 16 + b;
#pragma unroll
      for (int j = 0; j < 4; ++j) hh[j * 16] = hn[j];
    }
    if (nown > 0) {
      // under the wait: what the next step's owner phase needs and the exchange does not carry.  With a single chunk
      // (hidden <= 256 and 3+W+V <= 128) the extras weights simply stay in their registers from step to step.
      if (NCH > 1) load_chunk(0);
      if (decode && roleB && wave == 0 && lane < nown)
        osm[lane * 8 + 3] = pmg_uniform(a.seed, (uint64_t)(a.seq_base + row00 + sq0 + lane * sqs), a.offset + (uint64_t)i);
    }
    if (!(ok = FC_WAIT())) break;                // B: the new hidden state of every member visible
    FC_TICK(6)
  }
#ifdef PMG_PROFILE
  if (prof && tid == 0)
    printf("fwdc member 0 cycles: gather-h %lld extras %lld logits %lld stack+publish %lld barrierA %lld gates %lld barrierB %lld (init %lld)\n",
           tph[0], tph[1], tph[2], tph[3], tph[4], tph[5], tph[6], tph[7]);
#endif

  // ---------------------------------------------------------------- epilogue
  if (!ok) {
    // gave up on a barrier: poison what the caller will look at
    for (int k = 0; k < nown; ++k) {
      const int row = row00 + sq0 + k * sqs;
      if (row < B) {
        if (decode && tid == 0) a.len_out[row] = -1;
        if (teacher) for (int t = tid; t < N; t += FC_THREADS) a.logp_out[(size_t)t * B + row] = __int_as_float(0x7fc00000);
      }
    }
    return;
  }
  __syncthreads();
  if (decode && tid == 0)
    for (int k = 0; k < (roleB ? nown : 0); ++k) {
      const int row = row00 + sq0 + k * sqs;
      if (row < B) a.len_out[row] = osi[k * 8 + 6];
    }
  if (a.h_out != nullptr && wave == 0)
    for (int r = 0; r < NGc; ++r)
      if (row00 + 16 * r + b < B) st4(a.h_out + (size_t)(row00 + 16 * r + b) * H + 16 * cu + 4 * q, ld4(hl + r * 256 + lane * 4));
}

// groups per cluster, owned sequences per member, LDS bytes without / with the token-table lines
// split ownership (see the kernel): 32-member clusters whose control + push rows and vocabulary rows each fit one
// register chunk, as long as a stack per row group fits the LDS of the stack-side members
bool fwdc_duo(const FwdArgs* a, int ng) {
  return a->H == 512 && 3 + a->W <= 64 && a->V <= 64 &&
         (size_t)fc_lds(a->H, ng, ng, a->D * a->W).gtl * sizeof(float) <= (size_t)160 * 1024;
}
int fwdc_nown(const FwdArgs* a, int ng) { const int C = a->H / 16; return fwdc_duo(a, ng) ? ng : (16 * ng + C - 1) / C; }
size_t fwdc_lds_base_bytes(const FwdArgs* a, int ng) {
  return (size_t)fc_lds(a->H, ng, fwdc_nown(a, ng), a->D * a->W).gtl * sizeof(float);
}
bool fwdc_gtab_fits(const FwdArgs* a, int ng) { return fwdc_lds_base_bytes(a, ng) + (size_t)a->V * 48 * 4 <= (size_t)160 * 1024; }
size_t fwdc_lds_bytes(const FwdArgs* a, int ng) {
  return fwdc_lds_base_bytes(a, ng) + (fwdc_gtab_fits(a, ng) ? (size_t)a->V * 48 * 4 : 0);
}

}  // namespace

// Floats of the exchange region and words of the counter region (both modes): NG row groups per
// cluster, ceil(G / NG) clusters.
size_t pmg_fwdc_cx_floats(const PmgDims& d) {
  const size_t ng = pmg_cluster_ng(d.G, d.H), ncl = (d.G + ng - 1) / ng;
  return ncl * ng * ((size_t)2 * d.H * 16 + 16 * 128);
}
size_t pmg_fwdc_sync_words(const PmgDims& d) { return (size_t)32 * (1 + d.G); }

// The clustered kernel serves: one layer, hidden 128..512 (a cluster is H/16 <= 32 CUs, at most
// 8 * floor(32 / C) clusters so that every one is resident at once), up to PMG_CLUSTER_NG_MAX row groups
// per cluster as far as the owned sequences' stacks fit the LDS, extras of at most 256 rows, decode or
// teacher-forced mode.
bool pmg_fwdc_applicable(const FwdArgs* a) {
  if (a->flags & PMG_FLAG_NO_CLUSTER) return false;
  if (a->mode != PMG_FWD_DECODE && a->mode != PMG_FWD_TEACHER) return false;
  if (a->L != 1 || a->H > 512 || a->NX > 256 || a->CX == nullptr) return false;
  const int ng = pmg_cluster_ng(a->G, a->H);
  if (ng > pmg_cluster_ng_limit(a->H)) return false;
  return fwdc_lds_base_bytes(a, ng) <= (size_t)160 * 1024;
}

hipError_t pmg_launch_fwdc(const FwdArgs* a0, hipStream_t st) {
  FwdArgs args = *a0;
  const int ng = pmg_cluster_ng(a0->G, a0->H), ncl = (a0->G + ng - 1) / ng;
  args.cluster_ng = ng;
  args.cluster_gtab_lds = fwdc_gtab_fits(a0, ng) ? 1 : 0;
  args.cluster_duo = fwdc_duo(a0, ng) ? 1 : 0;
  args.cluster_hist_t = (a0->mode == PMG_FWD_TEACHER && a0->LAYW != nullptr &&
                         pmg_bwdc_shape_ok(a0->H, a0->L, a0->NX, a0->NXT, a0->V, a0->W, a0->D, a0->G)) ? 1 : 0;
  const FwdArgs* a = &args;
  const int C = a->H / 16;
  size_t lds = fwdc_lds_bytes(a, ng);
  if (lds < (size_t)81 * 1024) lds = (size_t)81 * 1024;     // more than half a CU: one workgroup per CU
  // counters and the status word start at zero (stream-ordered; 128 B per counter)
  hipError_t e = hipMemsetAsync(a->CXS, 0, (size_t)32 * (1 + ncl) * sizeof(unsigned), st);
  if (e != hipSuccess) return e;
  if (a->mode == PMG_FWD_TEACHER && a->LAYW != nullptr) {
    e = hipMemsetAsync(a->LAYW, a->cluster_hist_t ? 1 : 0, sizeof(unsigned), st);
    if (e != hipSuccess) return e;
  }
  const int grid = 8 * C * ((ncl + 7) / 8);
#define FC_LAUNCH(MODE_, HM_)                                                                              \
  {                                                                                                        \
    static std::atomic<unsigned long long> done{0};                                                        \
    e = pmg_full_lds_once(reinterpret_cast<const void*>(pmg_fwdc_kernel<MODE_, HM_>), &done);              \
    if (e != hipSuccess) return e;                                                                         \
    hipLaunchKernelGGL((pmg_fwdc_kernel<MODE_, HM_>), dim3(grid), dim3(FC_THREADS), lds, st, *a);          \
    return hipGetLastError();                                                                              \
  }
#define FC_LAUNCH_H(MODE_)                        \
  switch (a->H) {                                 \
    case 128: FC_LAUNCH(MODE_, 2)                 \
    case 256: FC_LAUNCH(MODE_, 4)                 \
    case 384: FC_LAUNCH(MODE_, 6)                 \
    case 512: FC_LAUNCH(MODE_, 8)                 \
  }
  if (a->mode == PMG_FWD_DECODE) FC_LAUNCH_H(PMG_FWD_DECODE) else FC_LAUNCH_H(PMG_FWD_TEACHER)
#undef FC_LAUNCH_H
#undef FC_LAUNCH
  return hipErrorInvalidValue;
}
