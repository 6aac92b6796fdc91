// Clustered ("hidden-partitioned") reward-weighted BPTT for SMALL batches: the counterpart of
// pmg_fwdc.hip for the recurrent half of rl_loss.backward() (reference: reinforce_omics.py:335), same
// arithmetic, same outputs (grad_h0, DG / DX gradient slabs for pmg_wgrad.hip, loss partial) as
// pmg_bwd.hip, whose header gives the per-step algebra (restated in oracle/fused_model.py).
//
// A cluster of C = H/16 workgroups (one per CU) walks the 16 sequences of a row group backwards in
// time.  Member c owns hidden units 16c..16c+15 for all 16 sequences: it carries their hidden-state
// gradient, runs the GRU cell backward on them and keeps, resident in LDS, the 16 rows of W_hh^T, the
// own 48 gate rows' part of W_ih[:, E:]^T and the 16 rows of [W_ctrl; W_push; W_out]^T that produce
// those units (all three copied from the transposed packs of pmg_pack.hip).  Member s also owns
// sequence s for what is per sequence: the stack backward (the whole stack gradient lives in its LDS),
// control and push gradients, dlogits, the loss.  Per reverse step p, two exchanges:
//   members: cell backward of the own units -> publish {dr, dz, dghn blocks; partial dtop}   | barrier A
//   owners : dtop = sum of the C partials -> stack backward -> dx = [dctrl | dpush | dlogits] -> publish | barrier B
//            (members form W_hh^T [dr; dz; dghn] for their units under the wait of barrier B)
//   members: carry = z * dh + W_hh^T dgh + [W_ctrl; W_push; W_out]^T dx  (own units)
// Hand-offs, barriers, residency rule and the bounded polls are those of pmg_fwdc.hip.
#include "pmg_kernels.h"

namespace {

#define BC_THREADS 512
#define BC_SPIN_LIMIT 3000000

typedef unsigned bc_u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x4 ld4(const float* p) {
  const float4 v = *reinterpret_cast<const float4*>(p);
  f32x4 r; r.x = v.x; r.y = v.y; r.z = v.z; r.w = v.w; return r;
}
__device__ __forceinline__ void st4(float* p, const f32x4 v) {
  *reinterpret_cast<float4*>(p) = make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ f32x4 xld4(const __amdgpu_buffer_rsrc_t r, unsigned voff) {
  const bc_u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, voff, 0, 16);
  f32x4 o; o.x = __uint_as_float(v.x); o.y = __uint_as_float(v.y); o.z = __uint_as_float(v.z); o.w = __uint_as_float(v.w);
  return o;
}
__device__ __forceinline__ float xld1(const __amdgpu_buffer_rsrc_t r, unsigned voff) {
  return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, voff, 0, 16));
}
__device__ __forceinline__ void xst4(const __amdgpu_buffer_rsrc_t r, unsigned voff, const f32x4 h) {
  bc_u32x4 u;
  u.x = __float_as_uint(h[0]); u.y = __float_as_uint(h[1]); u.z = __float_as_uint(h[2]); u.w = __float_as_uint(h[3]);
  __builtin_amdgcn_raw_buffer_store_b128(u, r, voff, 0, 16);
}
__device__ __forceinline__ void xst1(const __amdgpu_buffer_rsrc_t r, unsigned voff, const float v) {
  __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), r, voff, 0, 16);
}

__device__ __forceinline__ void cluster_arrive(unsigned* ctr) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// RAW: the closing workgroup barrier without the fence of __syncthreads(), which drains the vector-memory counter:
// loads requested on purpose in front of the wait (next step's gates and history) stay in flight across it.
template <bool RAW = false>
__device__ __forceinline__ bool cluster_wait(unsigned* ctr, const unsigned target, unsigned* status, int* okw) {
  if (threadIdx.x == 0) {
    int ok = 1;
    int spins = 0;
    while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      if (++spins > BC_SPIN_LIMIT || __hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) {
        __hip_atomic_store(status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ok = 0;
        break;
      }
      __builtin_amdgcn_s_sleep(2);
    }
    *okw = ok;
  }
  if (RAW) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  else __syncthreads();
  return *okw != 0;
}

// sum over the 64 lanes of a wave on the DPP network (xor 1, xor 2, mirror within 8 and 16 lanes, row_bcast:15 / :31),
// read out of lane 63 into a scalar: no LDS round trips
#define BC_DPP(x, ctrl, rmask) __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, x), __builtin_bit_cast(int, x), ctrl, rmask, 0xf, false))
__device__ __forceinline__ float wave_sum(float x) {
  x += BC_DPP(x, 0xB1, 0xf); x += BC_DPP(x, 0x4E, 0xf);
  x += BC_DPP(x, 0x141, 0xf); x += BC_DPP(x, 0x140, 0xf);
  const float r15 = BC_DPP(x, 0x142, 0xa);       // rows 1, 3: the row below's total
  x = ((threadIdx.x & 16) ? x + r15 : x);
  const float r31 = BC_DPP(x, 0x143, 0xc);       // rows 2, 3: total of rows 0-1
  x = ((threadIdx.x & 32) ? x + r31 : x);
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), 63));   // typed int: move, not convert
}

#define BC_HIST 336   // floats of forward history per owned sequence: PV[64] | push, pop, no-op, logp, token | .. | PR[V] from 72
struct BcLds { unsigned wt_is, wt_x, red, own, dtls, r3s, dxls, hist, sprevs, dsb, misc, end; };
__host__ __device__ inline BcLds bc_lds(const int NXT, const int NOWN, const int DW) {
  BcLds o;
  unsigned p = 0;
  o.wt_is = p; p += 12 * 256;               // [4][3][64][4]  own gate rows' part of W_ih[:, E:]^T
  o.wt_x = p; p += (unsigned)NXT * 256;     // [NXT][64][4]   rows 16cu.. of [ctrl; push; out]^T
  o.red = p; p += 2 * 8 * 256;              // [2][8][256] K-split partials, double-buffered over the groups
  o.own = p; p += (unsigned)NOWN * 4;       // [NOWN][4] reward | sum_p R * logp of the owned sequences
  o.dtls = p; p += (unsigned)NOWN * 64;     // [NOWN][64] dtop
  o.r3s = p; p += (unsigned)NOWN * 32;      // [NOWN][8][4] control-gradient partials per wave
  o.dxls = p; p += (unsigned)NOWN * 256;    // [NOWN][256] dx
  o.hist = p; p += (unsigned)NOWN * BC_HIST;   // [NOWN][..] forward history of the step, fetched under barrier A
  o.sprevs = p; p += (unsigned)NOWN * DW;   // [NOWN][D*W] S_{p-1}, fetched under barrier A
  o.dsb = p; p += (unsigned)NOWN * 2 * DW;  // [NOWN][2][D*W] stack gradient ping-pong
  o.misc = p; p += 64;
  o.end = p;
  return o;
}

// HM = H / 64.  One layer.
//
// Like the forward (pmg_fwdc.hip) a cluster serves NG <= 8 consecutive row groups: the two barriers of a
// step are shared by the groups; sequence sigma = 16 r + s of the cluster is OWNED by member sigma % C.
// What is per (own units, group) -- the carry, the cell backward, the partial dtop, the extras product --
// belongs to WAVE r of the member, in registers, with no workgroup barrier; the W_hh^T product of a
// group is split over the eight waves along K (each keeps its slice of the rows of W_hh^T in registers).
template <int HM>
__global__ __launch_bounds__(BC_THREADS, 2) void pmg_bwdc_kernel(const BwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int H = 64 * HM, KG = H / 16, C = KG, UPW = H / 128 > 0 ? H / 128 : 1, KG3 = 3 * H / 16, KPW = KG3 / 8;
  static_assert(H % 128 == 0, "hidden size is a multiple of 128");
  const int W = a.W, D = a.D, V = a.V, B = a.B, G = a.G, N = a.N, NXT = a.NXT;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int b = lane & 15, q = lane >> 4;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int csub = slot / C, cu = slot - csub * C;
  const int cl = csub * 8 + xcd;
  const int NG = a.cluster_ng;
  const int g0 = cl * NG;
  if (g0 >= G) return;
  const int NGc = (G - g0 < NG) ? G - g0 : NG;
  const int NS = 16 * NGc;
  const int NOWN = (16 * NG + C - 1) / C;
  const int nown = (cu < NS) ? (NS - cu + C - 1) / C : 0;
  const int row00 = g0 * PMG_ROWS;
  const int DW = D * W;
  const bool ht = a.LAYW != nullptr && *a.LAYW != 0u;   // history per sequence (FwdArgs::LAYW), as the clustered forward left it
  const bool mygrp = wave < NGc;                 // this wave carries group r = wave of the own units
  const int gw = g0 + wave;

  // ---- LDS
  const BcLds lo = bc_lds(NXT, NOWN, DW);
  float* fs = reinterpret_cast<float*>(smem);
  float* wt_is = fs + lo.wt_is;
  float* wt_x = fs + lo.wt_x;
  float* red = fs + lo.red;
  float* own = fs + lo.own;
  float* dtls = fs + lo.dtls;
  float* r3s = fs + lo.r3s;
  float* dxls = fs + lo.dxls;
  float* hist = fs + lo.hist;
  float* sprevs = fs + lo.sprevs;
  float* dsb = fs + lo.dsb;
  int* misc = reinterpret_cast<int*>(fs + lo.misc);

  // ---- exchange buffers of this cluster (floats): dg blocks [2][NG][3][C][256] | partial dtop [NG][C][4][256] |
  //      dx records [16 NG][256] | loss records [16 NG][64]
  const size_t dgf = (size_t)NG * 3 * C * 256;
  const size_t o_pt = 2 * dgf, o_dx = o_pt + (size_t)NG * C * 4 * 256, o_ls = o_dx + (size_t)NG * 16 * 256;
  const size_t cxs = o_ls + (size_t)NG * 16 * 64;
  float* cx = a.CX + (size_t)cl * cxs;
  const __amdgpu_buffer_rsrc_t dgb0 = pmg_rsrc(cx, (unsigned)(dgf * 4));
  const __amdgpu_buffer_rsrc_t dgb1 = pmg_rsrc(cx + dgf, (unsigned)(dgf * 4));
  const __amdgpu_buffer_rsrc_t ptb = pmg_rsrc(cx + o_pt, (unsigned)(NG * C * 4 * 256) * 4u);
  const __amdgpu_buffer_rsrc_t dxb = pmg_rsrc(cx + o_dx, (unsigned)NG * 16u * 1024u);
  const __amdgpu_buffer_rsrc_t lsb = pmg_rsrc(cx + o_ls, (unsigned)NG * 16u * 256u);
  unsigned* ctr = a.CXS + 32 + (size_t)cl * 32;
  unsigned* status = a.CXS;
  unsigned bars = 0;
#define BC_ARRIVE() (++bars, cluster_arrive(ctr))
#define BC_WAIT() cluster_wait(ctr, (unsigned)C * bars, status, misc)
#define BC_WAIT_RAW() cluster_wait<true>(ctr, (unsigned)C * bars, status, misc)
#ifdef PMG_PROFILE
  // -DPMG_PROFILE build + PMG_DEBUG bit 0x8000: per-phase cycle counters of member 0 of cluster 0
  const bool prof = (a.dbg & 0x8000) != 0 && cl == 0 && cu == 0;
  long long tph[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  long long tlast = prof ? (long long)__builtin_amdgcn_s_memtime() : 0;
#define BC_TICK(k) if (prof) { const long long tn_ = (long long)__builtin_amdgcn_s_memtime(); tph[k] += tn_ - tlast; tlast = tn_; }
#else
#define BC_TICK(k)
#endif

  // ---- resident transposed weights out of the packs.  W_hh^T rows 16cu..: [KG3][64][4] in chunk order;
  // wave w keeps reduction groups w*KPW .. w*KPW+KPW-1 in registers
  const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
  f32x4 wthh[KPW];
  {
    const float* src = a.wbase + a.oWHHt[0] / 4 + ((size_t)cu * KG3 + wave * KPW) * 256 + lane * 4;
#pragma unroll
    for (int i = 0; i < KPW; ++i) wthh[i] = ld4(src + (size_t)i * 256);
    // W_ih[:, E:]^T: tile t (16 stack columns), reduction group of (piece, unit tile cu) in chunk order
    const int kc = cu % UPW, kwv = cu / UPW;
    for (int i = tid * 4; i < 12 * 256; i += BC_THREADS * 4) {
      const int blk = i >> 8, t = blk / 3, piece = blk - 3 * t;
      const int kgi = kc * 24 + piece * 8 + kwv;
      st4(wt_is + i, ld4(a.wbase + a.oWIHt[0] / 4 + ((size_t)t * KG3 + kgi) * 256 + (i & 255)));
    }
    const float* srcx = a.wbase + a.oWXt / 4 + (size_t)cu * NXT * 256;
    for (int i = tid * 4; i < NXT * 256; i += BC_THREADS * 4) st4(wt_x + i, ld4(srcx + i));
  }
  for (int i = tid; i < NOWN * 2 * DW; i += BC_THREADS) dsb[i] = 0.0f;
  for (int i = tid; i < NOWN * 256; i += BC_THREADS) dxls[i] = 0.0f;
  if (tid < 64) misc[tid] = 0;
  if (tid < nown) {
    const int row = row00 + cu + tid * C;
    own[tid * 4] = (row < B) ? a.reward[row] : 0.0f;
    own[tid * 4 + 1] = 0.0f;
  }
  __syncthreads();

  // stack thread mapping: thread -> (row offset sd0, column sw), rows sd0 + j*SDS
  const int SDS = BC_THREADS / W;
  const int sd0 = tid / W, sw = (sd0 < SDS) ? tid - sd0 * W : W;
  f32x4 car = z4;                       // wave r: hidden-state gradient of the own units of group r (D layout)
  f32x4 dhz = z4;                       // wave r: z * dh of this step
  f32x4 hhs = z4;                       // wave r: W_hh^T [dr; dz; dghn] of this step
  bool ok = true;
  // Forward history of step p for every owned sequence -> LDS: S_{p-1}, push values, controls, log-prob, target
  // token, probabilities.  Plain loads of what the forward wrote long ago; issued under the wait of barrier A.
  // fetch in two halves: fetch_issue requests round m0 of a step's history into registers (8 stack elements per thread
  // ACROSS the owned sequences, and for two sequences the small items), fetch_commit puts them into LDS.
  float fv[8], f1[2], f2[2];
  // thread mapping of a fetch: t = index among the fetching threads, (sdx, swx) its stack row offset and column,
  // SDSx rows apart, rptx rows per thread and sequence
  struct FetchMap { int t, sdx, swx, SDSx, rptx; };
  const int rpt = (D + SDS - 1) / SDS;
  const FetchMap fm_all = {tid, sd0, sw, SDS, rpt};
  // Helper fetch: with few row groups per cluster the waves that carry no group (wave >= NGc) have nothing to do between
  // barrier B and barrier A (carry, cell backward).  They fetch the coming step's history there, so that its HBM latency
  // leaves the wait of barrier A.  Needs at least four such waves, one round of eight stack elements per thread and room
  // for the probabilities behind thread 128.
  const int nhw = 8 - NGc, nth = 64 * nhw, hti = tid - 64 * NGc;
  const int HSDS = nth / W > 0 ? nth / W : 1;
  const int hsd0 = hti >= 0 ? hti / W : 0;
  const FetchMap fm_help = {hti, hsd0, (hti >= 0 && hsd0 < HSDS) ? hti - hsd0 * W : W, HSDS, (D + HSDS - 1) / HSDS};
  constexpr bool HFETCH = true;                  // (20-36 B of scratch per lane at hidden 384 / 512, still 2 % faster there)
  const bool hfetch = HFETCH && nhw >= 4 && nown * fm_help.rptx <= 8 && nown <= 2 && V + 128 <= nth && W <= nth;   // workgroup-uniform
  auto fetch_issue = [&](const FetchMap& fm, const int p, const int m0, const int k0) __attribute__((always_inline)) {
    const int tid = fm.t, sd0 = fm.sdx, sw = fm.swx, SDS = fm.SDSx, rpt = fm.rptx, MF = nown * fm.rptx;
    const bool virt = (p == N);
    // S_{p-1}: rows <= min(p-1, N-p, D-1) hold data (pmg_fwd.hip "live rows"), the rest is zero
    const int s_hi = (virt || p == 0) ? -1 : min(min(p - 1, N - p), D - 1);
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int m = m0 + u, k = m / rpt, d = sd0 + (m - k * rpt) * SDS;
      const int sg = cu + k * C, s = sg & 15, g = g0 + (sg >> 4);
      fv[u] = 0.0f;
      if (m < MF && sw < W && d <= s_hi) {
        const float* Sp = a.SB + ((size_t)p * G + g) * a.SW;
        fv[u] = Sp[ht ? (size_t)s * DW + d * W + sw : (size_t)(d * W + sw) * 16 + s];
      }
    }
    // push values, controls, log-prob, target token (one thread each) and probabilities
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int k = k0 + u;
      const int sg = cu + k * C, s = sg & 15, g = g0 + (sg >> 4), row = row00 + sg;
      f1[u] = 0.0f; f2[u] = 0.0f;
      if (k < nown) {
        if (!virt) {
          if (tid < W) f1[u] = a.PV[((size_t)p * G + g) * 1024 + (ht ? s * 64 + tid : tid * 16 + s)];
          else if (tid >= 64 && tid < 67) f1[u] = a.CT[((size_t)p * G + g) * 64 + (tid - 64) * 16 + s];
        }
        if (p >= 1) {
          if (tid == 67) f1[u] = a.LP[((size_t)(p - 1) * G + g) * 16 + s];
          if (tid == 68) f1[u] = __int_as_float((row < B) ? min(max((int)a.tokens[(size_t)p * B + row], 0), V - 1) : 0);
          if (tid >= 128 && tid < 128 + V)
            f2[u] = a.PR[((size_t)(p - 1) * G + g) * a.VP * 16 + (ht ? s * a.VP + tid - 128 : (tid - 128) * 16 + s)];
        }
      }
    }
  };
  auto fetch_commit = [&](const FetchMap& fm, const int p, const int m0, const int k0) __attribute__((always_inline)) {
    const int tid = fm.t, sd0 = fm.sdx, sw = fm.swx, SDS = fm.SDSx, rpt = fm.rptx, MF = nown * fm.rptx;
    const bool virt = (p == N);
    if (!virt) {
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int m = m0 + u, k = m / rpt, d = sd0 + (m - k * rpt) * SDS;
        if (m < MF && sw < W && d < D) sprevs[(size_t)k * DW + d * W + sw] = fv[u];
      }
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int k = k0 + u;
      if (k < nown) {
        float* hk = hist + k * BC_HIST;
        if (!virt && (tid < W || (tid >= 64 && tid < 67))) hk[tid] = f1[u];
        if (p >= 1) {
          if (tid == 67 || tid == 68) hk[tid] = f1[u];
          if (tid >= 128 && tid < 128 + V) hk[72 + tid - 128] = f2[u];
        }
      }
    }
  };
  // the whole history of a step, round by round (one memory latency per round)
  auto fetch_history = [&](const int p) __attribute__((always_inline)) {
    for (int m0 = 0, k0 = 0; m0 < nown * rpt || k0 < nown; m0 += 8, k0 += 2) {
      fetch_issue(fm_all, p, m0, k0);
      fetch_commit(fm_all, p, m0, k0);
    }
  };
  // saved gates and previous hidden state of the NEXT step's cell backward (wave r, own units of group r): requested
  // under barrier A next to the history fetch, a whole step before they are used
  f32x4 pg0 = z4, pg1 = z4, pg2 = z4, pg3 = z4, pg4 = z4;
  auto load_gates = [&](const int pp) __attribute__((always_inline)) {
    const float* gp = a.GT[0] + (((size_t)pp * G + gw) * KG + cu) * 1024 + lane * 4;
    pg0 = ld4(gp); pg1 = ld4(gp + 256); pg2 = ld4(gp + 512); pg3 = ld4(gp + 768);
    const float* hh = a.HH[0] + (((size_t)pp * G + gw) * H + 16 * cu + 4 * q) * 16 + b;
#pragma unroll
    for (int j = 0; j < 4; ++j) pg4[j] = hh[j * 16];
  };
  if (mygrp && N >= 1) load_gates(N - 1);

  for (int p = N; p >= 0 && ok; --p) {
    const bool virt = (p == N);
    const __amdgpu_buffer_rsrc_t dgw = (p & 1) ? dgb1 : dgb0;
    // ================================================================ members: cell backward of the own units, one wave per group
    if (!virt) {
      f32x4 gr = z4, gz = z4, gn = z4, gg = z4;    // dr_pre, dz_pre, dn_pre, dghn
      if (mygrp) {
        const f32x4 r = pg0, z = pg1, n = pg2, ghn = pg3, hp = pg4;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float d = car[j];
          const float dn = d * (1.0f - z[j]);
          const float dz = d * (hp[j] - n[j]);
          gn[j] = dn * (1.0f - n[j] * n[j]);
          const float dr = gn[j] * ghn[j];
          gg[j] = gn[j] * r[j];
          gz[j] = dz * z[j] * (1.0f - z[j]);
          gr[j] = dr * r[j] * (1.0f - r[j]);
          dhz[j] = d * z[j];
        }
        // dr | dz | dghn blocks of the own units for the other members' W_hh^T products
        xst4(dgw, (unsigned)(((wave * 3 + 0) * C + cu) * 256 + lane * 4) * 4u, gr);
        xst4(dgw, (unsigned)(((wave * 3 + 1) * C + cu) * 256 + lane * 4) * 4u, gz);
        xst4(dgw, (unsigned)(((wave * 3 + 2) * C + cu) * 256 + lane * 4) * 4u, gg);
        // partial dtop of the four 16-column tiles: the own 48 gate rows' share, B operand = own dgi blocks (registers)
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          f32x4 acc = z4;
          const f32x4 w0 = ld4(wt_is + (size_t)(t * 3 + 0) * 256 + lane * 4), w1 = ld4(wt_is + (size_t)(t * 3 + 1) * 256 + lane * 4),
                      w2 = ld4(wt_is + (size_t)(t * 3 + 2) * 256 + lane * 4);
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w0[c], gr[c], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w1[c], gz[c], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w2[c], gn[c], acc, 0, 0, 0);
          }
          xst4(ptb, (unsigned)(((wave * C + cu) * 4 + t) * 256 + lane * 4) * 4u, acc);
        }
      }
      BC_TICK(0)
      BC_ARRIVE();
      if (mygrp) {
        // gate-gradient slab rows of the own units (A operand of the weight-gradient GEMM): off the chain
        float* dg = a.DG[0] + (((size_t)p * G + gw) * 4 * H + 16 * cu + 4 * q) * 16 + b;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          dg[(size_t)j * 16] = gr[j];
          dg[((size_t)H + j) * 16] = gz[j];
          dg[((size_t)2 * H + j) * 16] = gn[j];
          dg[((size_t)3 * H + j) * 16] = gg[j];
        }
      }
      if (!hfetch) fetch_history(p);
      if (mygrp && p >= 1) load_gates(p - 1);
      if (!(ok = BC_WAIT_RAW())) break;          // A: every member's dgate blocks and dtop partials visible
      BC_TICK(1)
    }

    // ================================================================ owners: per-sequence backward
    if (virt) { fetch_history(p); __syncthreads(); }
    // (i) dtop of every owned sequence: wave k sums the C members' partials of sequence k (lane = stack column)
    if (!virt)
      for (int k = wave; k < nown; k += 8) {
        const int sg = cu + k * C, r = sg >> 4, s = sg & 15;
        const int tt = lane >> 4, kq = (lane >> 2) & 3, j = lane & 3;
        constexpr int CH = C < 16 ? C : 16;        // partials requested per round
        float acc = 0.0f;
#pragma unroll
        for (int c0 = 0; c0 < C; c0 += CH) {
          float v[CH];
#pragma unroll
          for (int cm = 0; cm < CH; ++cm)
            v[cm] = (c0 + cm < C) ? xld1(ptb, (unsigned)((((r * C + c0 + cm) * 4 + tt) * 256 + (s + 16 * kq) * 4 + j)) * 4u) : 0.0f;
#pragma unroll
          for (int cm = 0; cm < CH; ++cm) acc += v[cm];
        }
        dtls[k * 64 + lane] = acc;
      }
    __syncthreads();
    // (ii) stack backward of every owned sequence (all threads, LDS only): G(d) = dS_p[d] (+ dtop at row 0);
    // rows beyond N-p carry no gradient
    if (!virt)
      for (int k = 0; k < nown; ++k) {
        const float* hk = hist + k * BC_HIST;
        const float* sprev = sprevs + (size_t)k * DW;
        const float* dsin = dsb + ((size_t)k * 2 + ((p + 1) & 1)) * DW;   // gradient w.r.t. S_p
        float* dsout = dsb + ((size_t)k * 2 + (p & 1)) * DW;              // gradient w.r.t. S_{p-1}
        const float apush = hk[64], apop = hk[65], anoop = hk[66];
        const int dlim = min(N - p, D - 1);
        float sn = 0.0f, sp = 0.0f, sq = 0.0f;
        if (sw < W) {
          const float pv = hk[sw];
          const float dt = dtls[k * 64 + sw];
          for (int d = sd0; d <= dlim; d += SDS) {
            const int idx = d * W + sw;
            const float gc = dsin[idx] + (d == 0 ? dt : 0.0f);
            const float gm = (d == 0) ? 0.0f : dsin[idx - W] + (d == 1 ? dt : 0.0f);
            const float gx = (d + 1 < D) ? dsin[idx + W] : 0.0f;
            const float sc = sprev[idx];
            const float su = (d == 0) ? pv : sprev[idx - W];
            const float sdn = (d + 1 < D) ? sprev[idx + W] : 0.0f;
            sn += gc * sc; sp += gc * su; sq += gc * sdn;
            dsout[idx] = anoop * gc + apush * gx + apop * gm;
            if (d == 0) dxls[k * 256 + 3 + sw] = apush * gc * (1.0f - pv * pv);
          }
        }
        sn = wave_sum(sn); sp = wave_sum(sp); sq = wave_sum(sq);
        if (lane == 0) { r3s[(k * 8 + wave) * 4] = sp; r3s[(k * 8 + wave) * 4 + 1] = sq; r3s[(k * 8 + wave) * 4 + 2] = sn; }
      }
    __syncthreads();
    // (iii) control gradients (thread k), dlogits_{p-1} and the loss term of every owned sequence
    if (tid < nown) {
      const int k = tid;
      const float* hk = hist + k * BC_HIST;
      if (!virt) {
        const float apush = hk[64], apop = hk[65], anoop = hk[66];
        float dpu = 0.f, dpo = 0.f, dno = 0.f;
        for (int wv = 0; wv < 8; ++wv) { dpu += r3s[(k * 8 + wv) * 4]; dpo += r3s[(k * 8 + wv) * 4 + 1]; dno += r3s[(k * 8 + wv) * 4 + 2]; }
        const float sm = apush * dpu + apop * dpo + anoop * dno;
        dxls[k * 256] = apush * (dpu - sm); dxls[k * 256 + 1] = apop * (dpo - sm); dxls[k * 256 + 2] = anoop * (dno - sm);
      } else {
        dxls[k * 256] = 0.0f; dxls[k * 256 + 1] = 0.0f; dxls[k * 256 + 2] = 0.0f;
      }
      if (p >= 1) own[k * 4 + 1] += own[k * 4] * hk[67];
    }
    for (int t = tid; t < nown * 256; t += BC_THREADS) {
      const int k = t >> 8, v = t & 255;
      const float* hk = hist + k * BC_HIST;
      if (v < V) {
        const int y = __float_as_int(hk[68]);
        dxls[k * 256 + 3 + W + v] = (p >= 1) ? a.scale * own[k * 4] * (hk[72 + v] - ((v == y) ? 1.0f : 0.0f)) : 0.0f;
      }
      if (virt && v < W) dxls[k * 256 + 3 + v] = 0.0f;
    }
    __syncthreads();
    // (iv) publish dx of every owned sequence (1-KB record: one 16-byte store per lane) and its DX slab column: wave k
    for (int k = wave; k < nown; k += 8) {
      const int sg = cu + k * C, s = sg & 15, g = g0 + (sg >> 4);
      xst4(dxb, (unsigned)(sg * 256 + lane * 4) * 4u, ld4(dxls + k * 256 + lane * 4));
      float* dxg = a.DX + ((size_t)p * G + g) * NXT * 256 + s;
      for (int idx = lane; idx < NXT * 16; idx += 64) dxg[(size_t)idx * 16] = dxls[k * 256 + idx];
    }
    BC_TICK(2)
    BC_ARRIVE();
    // ---- under the wait of barrier B: W_hh^T [dr; dz; dghn] of every group for the own units, K split over the waves
    if (!virt) {
      // the dgate blocks of group r+1 are requested as soon as the products of group r have consumed the registers:
      // they land under its reduction
      f32x4 xv[KPW];
      auto load_dg = [&](const int r) {
#pragma unroll
        for (int i = 0; i < KPW; ++i) {
          const int kgi = wave * KPW + i;
          const int kc = kgi / 24, rem = kgi - kc * 24, piece = rem >> 3, kwv = rem & 7;
          const int u = kwv * UPW + kc;
          xv[i] = xld4(dgw, (unsigned)(((r * 3 + piece) * C + u) * 256 + lane * 4) * 4u);
        }
      };
      load_dg(0);
      for (int r = 0; r < NGc; ++r) {
        float* redr = red + (size_t)(r & 1) * 8 * 256;
        f32x4 acc0 = z4, acc1 = z4;
#pragma unroll
        for (int i = 0; i < KPW; ++i) {
          if (i & 1) {
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wthh[i][0], xv[i][0], acc1, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wthh[i][1], xv[i][1], acc1, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wthh[i][2], xv[i][2], acc1, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wthh[i][3], xv[i][3], acc1, 0, 0, 0);
          } else {
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wthh[i][0], xv[i][0], acc0, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wthh[i][1], xv[i][1], acc0, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wthh[i][2], xv[i][2], acc0, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wthh[i][3], xv[i][3], acc0, 0, 0, 0);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (r + 1 < NGc) load_dg(r + 1);
        __builtin_amdgcn_sched_barrier(0);
        st4(redr + wave * 256 + lane * 4, acc0 + acc1);
        __syncthreads();                         // partials of group r visible (the other buffer is read by wave r-1 only)
        if (wave == r) {
          f32x4 sum = z4;
#pragma unroll
          for (int wv = 0; wv < 8; ++wv) sum += ld4(redr + wv * 256 + lane * 4);
          hhs = sum;
        }
      }
    }
    BC_TICK(3)
    if (!(ok = BC_WAIT())) break;                // B: dx of every sequence visible
    BC_TICK(4)
    if constexpr (HFETCH) {
      if (hfetch && wave >= NGc && p >= 1) {       // helper waves: the coming step's history, off the chain
        fetch_issue(fm_help, p - 1, 0, 0);
        fetch_commit(fm_help, p - 1, 0, 0);
      }
    }
    // ================================================================ members: the new carry of the own units, one wave per group
    if (mygrp) {
      f32x4 acc0 = virt ? z4 : dhz + hhs, acc1 = z4;
      // every dx block of a round is requested before the first product: one L2 round trip per eight reduction groups
      for (int kg0 = 0; kg0 < NXT; kg0 += 8) {
        f32x4 xr[8];
#pragma unroll
        for (int u = 0; u < 8; ++u)
          xr[u] = (kg0 + u < NXT) ? xld4(dxb, (unsigned)((wave * 16 + b) * 256 + 16 * (kg0 + u) + 4 * q) * 4u) : z4;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          if (kg0 + u < NXT) {
            const f32x4 wv = ld4(wt_x + (size_t)(kg0 + u) * 256 + lane * 4);
            if (u & 1) {
#pragma unroll
              for (int c = 0; c < 4; ++c) acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[c], xr[u][c], acc1, 0, 0, 0);
            } else {
#pragma unroll
              for (int c = 0; c < 4; ++c) acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[c], xr[u][c], acc0, 0, 0, 0);
            }
          }
        }
      }
      car = acc0 + acc1;
    }
    BC_TICK(5)
  }
#ifdef PMG_PROFILE
  if (prof && tid == 0)
    printf("bwdc member 0 cycles: cell+dtop %lld barrierA %lld owners %lld Whh^T product %lld barrierB %lld carry %lld\n", tph[0], tph[1], tph[2],
           tph[3], tph[4], tph[5]);
#endif

  // ---------------------------------------------------------------- outputs
  const float nanv = __int_as_float(0x7fc00000);
  if (!ok) {
    // gave up on a barrier: poison what the caller will look at
    if (mygrp && row00 + 16 * wave + b < B)
      st4(a.grad_h0 + (size_t)(row00 + 16 * wave + b) * H + 16 * cu + 4 * q, (f32x4){nanv, nanv, nanv, nanv});
    if (cu == 0 && tid < NGc) a.LOSSP[g0 + tid] = nanv;
    return;
  }
  if (mygrp && row00 + 16 * wave + b < B) st4(a.grad_h0 + (size_t)(row00 + 16 * wave + b) * H + 16 * cu + 4 * q, car);
  // loss partial of every row group: owners publish sum_p R * logp of their rows, member 0 adds them in row order
  __syncthreads();
  for (int k = wave; k < nown; k += 8) xst1(lsb, (unsigned)((cu + k * C) * 64 + lane) * 4u, lane == 0 ? own[k * 4 + 1] : 0.0f);
  BC_ARRIVE();
  ok = BC_WAIT();
  if (cu == 0 && tid < NGc) {
    float sacc = 0.0f;
    for (int r = 0; r < 16; ++r) sacc += xld1(lsb, (unsigned)((tid * 16 + r) * 64) * 4u);
    a.LOSSP[g0 + tid] = ok ? -a.scale * sacc : nanv;
  }
}

int bwdc_nown(const BwdArgs* a, int ng) { const int C = a->H / 16; return (16 * ng + C - 1) / C; }
size_t bwdc_lds_bytes(const BwdArgs* a, int ng) {
  return (size_t)bc_lds(a->NXT, bwdc_nown(a, ng), a->D * a->W).end * sizeof(float);
}

}  // namespace

// Floats of the exchange region per launch: NG row groups per cluster, ceil(G / NG) clusters.
size_t pmg_bwdc_cx_floats(const PmgDims& d) {
  const size_t C = d.H / 16, ng = pmg_cluster_ng(d.G, d.H), ncl = (d.G + ng - 1) / ng;
  return ncl * ng * ((size_t)2 * 3 * C * 256 + C * 4 * 256 + 16 * 256 + 16 * 64);
}

bool pmg_bwdc_shape_ok(int H, int L, int NX, int NXT, int V, int W, int D, int G) {
  if (L != 1 || H > 512 || NX > 256 || V > 264) return false;
  const int ng = pmg_cluster_ng(G, H);
  if (ng > pmg_cluster_ng_limit(H)) return false;
  if (BC_THREADS / W < 1) return false;
  BwdArgs t;
  t.H = H; t.NXT = NXT; t.D = D; t.W = W;
  return bwdc_lds_bytes(&t, ng) <= (size_t)160 * 1024;
}

bool pmg_bwdc_applicable(const BwdArgs* a) {
  if ((a->flags & PMG_FLAG_NO_CLUSTER) || a->CX == nullptr) return false;
  return pmg_bwdc_shape_ok(a->H, a->L, a->NX, a->NXT, a->V, a->W, a->D, a->G);
}

hipError_t pmg_launch_bwdc(const BwdArgs* a0, hipStream_t st) {
  BwdArgs args = *a0;
  const int ng = pmg_cluster_ng(a0->G, a0->H), ncl = (a0->G + ng - 1) / ng;
  args.cluster_ng = ng;
  const BwdArgs* a = &args;
  const int C = a->H / 16;
  size_t lds = bwdc_lds_bytes(a, ng);
  if (lds < (size_t)81 * 1024) lds = (size_t)81 * 1024;     // more than half a CU: one workgroup per CU
  hipError_t e = hipMemsetAsync(a->CXS, 0, (size_t)32 * (1 + ncl) * sizeof(unsigned), st);
  if (e != hipSuccess) return e;
  const int grid = 8 * C * ((ncl + 7) / 8);
#define BC_LAUNCH(HM_)                                                                                  \
  {                                                                                                     \
    static std::atomic<unsigned long long> done{0};                                                     \
    e = pmg_full_lds_once(reinterpret_cast<const void*>(pmg_bwdc_kernel<HM_>), &done);                  \
    if (e != hipSuccess) return e;                                                                      \
    hipLaunchKernelGGL((pmg_bwdc_kernel<HM_>), dim3(grid), dim3(BC_THREADS), lds, st, *a);              \
    return hipGetLastError();                                                                           \
  }
  switch (a->H) {
    case 128: BC_LAUNCH(2)
    case 256: BC_LAUNCH(4)
    case 384: BC_LAUNCH(6)
    case 512: BC_LAUNCH(8)
  }
#undef BC_LAUNCH
  return hipErrorInvalidValue;
}
