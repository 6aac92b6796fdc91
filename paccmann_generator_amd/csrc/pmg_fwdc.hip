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

// MODE: PMG_FWD_DECODE or PMG_FWD_TEACHER.  One layer.  HM = H / 64: float4 per lane and extras row.
template <int MODE, int HM>
__global__ __launch_bounds__(FC_THREADS, 2) void pmg_fwdc_kernel(const FwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr bool teacher = (MODE == PMG_FWD_TEACHER);
  constexpr bool decode = (MODE == PMG_FWD_DECODE);
  constexpr int H = 64 * HM, KG = H / 16, C = KG;
  const int W = a.W, D = a.D, V = a.V, B = a.B, G = a.G, N = a.N, NX = a.NX;                 // members per cluster = unit tiles
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int b = lane & 15, q = lane >> 4;
  // workgroup -> (cluster, member): clusters are dealt over the XCDs first
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int gsub = slot / C, cu = slot - gsub * C;
  const int g = gsub * 8 + xcd;
  if (g >= G) return;                            // a whole cluster absent: nobody waits for it
  const int row0 = g * PMG_ROWS;
  const int NOWN = (16 + C - 1) / C;             // sequences this member may own: s = cu + k*C
  const int DW = D * W;

  // ---- LDS
  float* wl_hh = reinterpret_cast<float*>(smem);            // [3][KG][64][4] packed gate rows of the own units
  float* wl_is = wl_hh + 3 * KG * 256;                      // [3][4][64][4]  stack-top columns of W_ih
  float* red = wl_is + 3 * 4 * 256;                         // [8 waves][2][256] K-split partials
  float* bxl = red + 8 * 2 * 256;                           // [256] extras bias (ctrl | push | out)
  float* xex = bxl + 256;                                   // [256] extras of the sequence being served
  float* stk = xex + 256;                                   // [NOWN][2][D*W] own sequences' stacks (ping-pong)
  int* misc = reinterpret_cast<int*>(stk + (size_t)NOWN * 2 * DW);   // [0] barrier ok, [1..] scratch
  float* gtl = reinterpret_cast<float*>(misc + 64);         // [V][3][16] token-table lines of the own units (if they fit)
  const bool gt_lds = a.cluster_gtab_lds != 0;

  // ---- exchange buffers of this cluster
  const size_t cxs = (size_t)2 * H * 16 + 16 * 128;         // floats per cluster
  float* cx = a.CX + (size_t)g * cxs;
  const __amdgpu_buffer_rsrc_t hb0 = pmg_rsrc(cx, (unsigned)H * 64u), hb1 = pmg_rsrc(cx + (size_t)H * 16, (unsigned)H * 64u);
  const __amdgpu_buffer_rsrc_t rec = pmg_rsrc(cx + (size_t)2 * H * 16, 16u * 512u);   // [16][128]: top[64] | token, finished
  unsigned* ctr = a.CXS + 32 + (size_t)g * 32;              // counter of this cluster (own 128-B line)
  unsigned* status = a.CXS;                                  // one word for the launch
  unsigned bars = 0;
#define FC_ARRIVE() (++bars, cluster_arrive(ctr))
#define FC_WAIT() cluster_wait(ctr, (unsigned)C * bars, status, misc)
#ifdef PMG_PROFILE
  // -DPMG_PROFILE build + PMG_DEBUG bit 0x8000: per-phase cycle counters of member 0 of cluster 0
  const bool prof = (a.dbg & 0x8000) != 0 && g == 0 && cu == 0;
  long long tph[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  long long tlast = prof ? (long long)__builtin_amdgcn_s_memtime() : 0;
#define FC_TICK(k) if (prof) { const long long tn_ = (long long)__builtin_amdgcn_s_memtime(); tph[k] += tn_ - tlast; tlast = tn_; }
#else
#define FC_TICK(k)
#endif

  // ---- resident weights: the packed forward operands of pmg_pack.hip, tiles 3*cu .. 3*cu+2
  {
    const float* src = a.wbase + a.oWHH[0] / 4 + (size_t)(3 * cu) * KG * 256;
    for (int i = tid * 4; i < 3 * KG * 256; i += FC_THREADS * 4) st4(wl_hh + i, ld4(src + i));
    const float* src2 = a.wbase + a.oWIH[0] / 4 + (size_t)(3 * cu) * 4 * 256;
    for (int i = tid * 4; i < 3 * 4 * 256; i += FC_THREADS * 4) st4(wl_is + i, ld4(src2 + i));
  }
  for (int i = tid; i < NOWN * 2 * DW; i += FC_THREADS) stk[i] = 0.0f;
  if (tid < 64) misc[tid] = 0;
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
  const int RX = (NX + 31) >> 5;                 // rows per 16-lane group
  const int NCH = (RX + RC - 1) / RC;            // chunks (1 at hidden <= 256 and NX <= 128)
  const bool owner = cu < 16;
  f32x4 wreg[RC * HM];
  auto load_chunk = [&](const int c) {
#pragma unroll
    for (int r = 0; r < RC; ++r) {
      const int n = rgrp + 32 * (c * RC + r);
      const float* wr = (n < 3) ? a.w_ctrl + (size_t)n * H
                                : (n < 3 + W) ? a.w_push + (size_t)(n - 3) * H : a.w_out + (size_t)(n - 3 - W) * H;
#pragma unroll
      for (int m = 0; m < HM; ++m)
        wreg[r * HM + m] = (n < NX) ? ld4(wr + 4 * l16 + 64 * m) : (f32x4){0.f, 0.f, 0.f, 0.f};
    }
  };

  // ---- h0 of the own units for the 16 sequences (D layout: lane (b, q) holds units 4q..4q+3 of row b)
  f32x4 hown = {0.f, 0.f, 0.f, 0.f};             // wave 0 only: own slice of the current hidden state
  if (wave == 0) {
    const int row = row0 + b;
    if (decode) {
      // latent_to_hidden tile cu: packed [tile][Z/16][64][4] (PK_L2H), B operand straight from the latents
      const float* wz = a.wbase + a.oL2H / 4 + (size_t)cu * (a.Z / 16) * 256 + lane * 4;
      f32x4 acc = ld4(a.b_l2h + 16 * cu + 4 * q), acc2 = {0.f, 0.f, 0.f, 0.f};
      for (int kg = 0; kg < a.Z / 16; kg += 2) {
        const f32x4 w0 = ld4(wz + (size_t)kg * 256), w1 = ld4(wz + (size_t)(kg + 1) * 256);
        f32x4 x0 = {0.f, 0.f, 0.f, 0.f}, x1 = x0;
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
    xst4(hb0, (unsigned)(cu * 256 + lane * 4) * 4u, hown);
    if (teacher) {
#pragma unroll
      for (int j = 0; j < 4; ++j) a.HH[0][((size_t)g * H + 16 * cu + 4 * q + j) * 16 + b] = hown[j];
    }
  }
  FC_ARRIVE();
  if (owner) load_chunk(0);
  bool ok = FC_WAIT();                            // h0 of every member visible
  float u0 = 0.f, u1 = 0.f;                       // the uniform of the next draw of each owned row (decode)
  float pv0 = 0.f, pv1 = 0.f, top0 = 0.f, top1 = 0.f;   // wave 0: push value / new stack top of each owned row (history, deferred)
  float* fm = reinterpret_cast<float*>(misc + 16);      // [2][4] stack controls of the owned rows
  // deferred stack update: thread -> (row offset sd0, column sw), SDS rows apart (no division in the loop)
  const int SDS = FC_THREADS / W;                 // W <= 64: at least 8 rows in flight
  const int sd0 = tid / W, sw = (sd0 < SDS) ? tid - sd0 * W : W;

  // per owned sequence (k < NOWN): sampling state, identical in the lanes that serve it
  bool fin0 = false, fin1 = false;
  int mylen0 = N, mylen1 = N;
  const f32x4 bhn = (wave == 0) ? ld4(a.BG[0] + 3 * H + 16 * cu + 4 * q) : (f32x4){0.f, 0.f, 0.f, 0.f};

  FC_TICK(7)
  f32x4 gr = {0.f, 0.f, 0.f, 0.f}, gz = gr, gn = gr, gvh = gr;    // wave 0, teacher: gates of this step (history)
  for (int i = 0; ok; ++i) {
    const __amdgpu_buffer_rsrc_t hin = (i & 1) ? hb1 : hb0, hout = (i & 1) ? hb0 : hb1;
    // ================================================================ owners: per-sequence work
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int s = cu + k * C;                   // uniform
      if (k >= NOWN || s >= 16) break;
      bool& fin = k ? fin1 : fin0;
      int& mylen = k ? mylen1 : mylen0;
      const int row = row0 + s;
      const bool rowok = row < B;
      float* sold = stk + ((size_t)k * 2 + (i & 1)) * DW;
      float* snew = stk + ((size_t)k * 2 + ((i + 1) & 1)) * DW;
      // ---- extras of sequence s.  Every lane reads its slice of the hidden state straight out of the
      // packed blocks (unit 16*kg + 4*kq + c sits in lane s + 16*kq of block kg): no staging, no barrier
      {
        f32x4 hv[HM];
#pragma unroll
        for (int m = 0; m < HM; ++m) {
          const float4 v = xld4(hin, (unsigned)((((l16 >> 2) + 4 * m) * 64 + s + 16 * (l16 & 3)) * 4) * 4u);
          hv[m] = (f32x4){v.x, v.y, v.z, v.w};
        }
        for (int c = 0; c < NCH; ++c) {
          if (c > 0) load_chunk(c);
#pragma unroll
          for (int r = 0; r < RC; ++r) {
            const int n = rgrp + 32 * (c * RC + r);
            float acc = 0.0f;
#pragma unroll
            for (int m = 0; m < HM; ++m) {
              acc = fmaf(wreg[r * HM + m][0], hv[m][0], acc); acc = fmaf(wreg[r * HM + m][1], hv[m][1], acc);
              acc = fmaf(wreg[r * HM + m][2], hv[m][2], acc); acc = fmaf(wreg[r * HM + m][3], hv[m][3], acc);
            }
            acc = row16_sum(acc);
            if (l16 == 0 && n < NX) xex[n] = acc + bxl[n];
          }
        }
        if (NCH > 1 && k + 1 < NOWN) load_chunk(0);   // registers hold the last chunk: first one back for the next sequence
      }
      FC_TICK(0)
      __syncthreads();
      FC_TICK(1)
      // ---- consume the logits of step i-1 (wave 0, lanes 0..31: entry v sits in lane v & 31)
      int tok = teacher ? ((rowok && i < N + 1) ? min(max((int)a.tok_in[(size_t)min(i, N) * B + row], 0), V - 1) : 0) : a.start_tok;
      if (i >= 1 && wave == 0 && lane < 32) {
        const int l5 = lane;
        const float* lg = xex + 3 + W;
        if (teacher) {
          float m = -INFINITY;
          for (int v = l5; v < V; v += 32) m = fmaxf(m, lg[v]);
#pragma unroll
          for (int o = 16; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 32));
          float total = 0.0f;
          for (int v = l5; v < V; v += 32) total += __expf(lg[v] - m);
#pragma unroll
          for (int o = 16; o >= 1; o >>= 1) total += __shfl_xor(total, o, 32);
          const int y = tok;       // tokens[i][row]: the target of step i-1 and the input of step i
          if ((y & 31) == l5) {
            const float lp = (lg[y] - m) - __logf(total);
            a.LP[((size_t)(i - 1) * G + g) * 16 + s] = lp;
            if (rowok) a.logp_out[(size_t)(i - 1) * B + row] = lp;
          }
          float* pr = a.PR + ((size_t)(i - 1) * G + g) * a.VP * 16;
          const float inv = __builtin_amdgcn_rcpf(total);
          for (int v = l5; v < V; v += 32) pr[v * 16 + s] = __expf(lg[v] - m) * inv;
        } else {
          const bool greedy = !(a.temperature > 0.0f);
          if (greedy) {
            float bv = -INFINITY; int bi = 0x7fffffff;
            for (int v = l5; v < V; v += 32) { const float x = lg[v]; if (x > bv) { bv = x; bi = v; } }
#pragma unroll
            for (int o = 16; o >= 1; o >>= 1) {
              const float ov = __shfl_xor(bv, o, 32); const int oi = __shfl_xor(bi, o, 32);
              if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
            }
            tok = bi;
          } else {
            // same slot-wise inverse-CDF draw as pmg_fwd.hip (oracle twin: oracle/philox.py sample_row)
            const float T = a.temperature;
            float m = -INFINITY;
            for (int v = l5; v < V; v += 32) m = fmaxf(m, lg[v] / T);
#pragma unroll
            for (int o = 16; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 32));
            const float u = k ? u1 : u0;          // pmg_uniform(seed, global row, offset + i - 1), formed under barrier B
            // running CDF: slot ks holds entries 32ks..32ks+31, one per lane (3+W+V <= 256 on this path)
            float cdf[8];
            float carry = 0.0f;
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
              cdf[ks] = 0.0f;
              if (32 * ks < V) {
                const int v = 32 * ks + l5;
                float e = (v < V) ? __expf(lg[v] / T - m) : 0.0f;
#pragma unroll
                for (int o = 1; o <= 16; o <<= 1) {
                  const float t = __shfl_up(e, o, 32);
                  if (l5 >= o) e += t;
                }
                cdf[ks] = carry + e;
                carry = __shfl(cdf[ks], 31, 32);
              }
            }
            const float target = u * carry;
            tok = -1;
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
              if (32 * ks < V) {
                const int v = 32 * ks + l5;
                const unsigned mk = (unsigned)__ballot((v < V) && (target < cdf[ks]));
                if (tok < 0 && mk) tok = 32 * ks + (__ffs(mk) - 1);
              }
            }
            if (tok < 0) tok = V - 1;
          }
          if (rowok && l5 == 0) a.tok_out[(size_t)row * N + (i - 1)] = tok;
          if (!fin && tok == a.stop_tok) { fin = true; mylen = i - 1; }
        }
        if (l5 == 0) { misc[2] = tok; misc[3] = (fin || !rowok) ? 1 : 0; }
      } else if (i == 0 && tid == 0) {
        misc[2] = tok; misc[3] = rowok ? 0 : 1;
      }
      __syncthreads();
      FC_TICK(2)
      if (i < N) {
        tok = misc[2];
        // ---- stack controls; ROW 0 of the new stack (= the stack top the gates need) and the record
        // go out first: the rest of the update and every history store wait until this member has
        // arrived at barrier A (below), off the chain the other members wait on
        const float c0 = xex[0], c1 = xex[1], c2 = xex[2];
        const float m = fmaxf(c0, fmaxf(c1, c2));
        const float e0 = __expf(c0 - m), e1 = __expf(c1 - m), e2 = __expf(c2 - m);
        const float inv = 1.0f / ((e0 + e1) + e2);
        const float apush = e0 * inv, apop = e1 * inv, anoop = e2 * inv;
        if (wave == 0) {
          float top = 0.0f, pv = 0.0f;
          if (lane < W) {
            pv = fast_tanh(xex[3 + lane]);
            const float nxt = (1 < D) ? sold[W + lane] : 0.0f;
            top = anoop * sold[lane] + apush * pv + apop * nxt;
            snew[lane] = top;
          }
          xst1(rec, (unsigned)(s * 128 + lane) * 4u, top);
          const float meta = (lane == 0) ? __int_as_float(tok) : (lane == 1) ? __int_as_float(misc[3]) : 0.0f;
          xst1(rec, (unsigned)(s * 128 + 64 + lane) * 4u, meta);
          if (k) { pv1 = pv; top1 = top; } else { pv0 = pv; top0 = top; }
          if (lane == 0) { fm[4 * k] = apush; fm[4 * k + 1] = apop; fm[4 * k + 2] = anoop; misc[8 + k] = tok; }
        }
      }
      if (k + 1 < NOWN) __syncthreads();         // xex is reused by the next owned sequence
    }
    FC_TICK(3)
    if (i == N) break;
    FC_ARRIVE();
    // ---- deferred owner work, under the wait of barrier A: stack rows >= 1 and the teacher history
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int s = cu + k * C;
      if (k >= NOWN || s >= 16) break;
      const bool rowok = row0 + s < B;
      const float* sold = stk + ((size_t)k * 2 + (i & 1)) * DW;
      float* snew = stk + ((size_t)k * 2 + ((i + 1) & 1)) * DW;
      const float apush = fm[4 * k], apop = fm[4 * k + 1], anoop = fm[4 * k + 2];
      // rows that hold data after this update: d <= min(i, D-1) (row d is zero until step d)
      const int dhi = (i < D - 1) ? i : D - 1;
      for (int d = 1 + sd0; d <= dhi; d += SDS) {       // thread (sd0, sw): column sw of rows 1 + sd0 + j*SDS
        if (sw < W) {
          const int idx = d * W + sw;
          const float nxt = (d + 1 < D) ? sold[idx + W] : 0.0f;
          const float nw = anoop * sold[idx] + apush * sold[idx - W] + apop * nxt;
          snew[idx] = nw;
          // history the BPTT reads: only rows that can still reach a stack top (pmg_fwd.hip, "live rows")
          if (teacher && d <= N - 1 - i) a.SB[((size_t)(i + 1) * G + g) * a.SW + (size_t)idx * 16 + s] = nw;
        }
      }
      if (teacher) {
        if (wave == 0 && lane < W) {
          const float pv = k ? pv1 : pv0, top = k ? top1 : top0;
          a.SB[((size_t)(i + 1) * G + g) * a.SW + (size_t)lane * 16 + s] = top;
          a.PV[((size_t)i * G + g) * 1024 + lane * 16 + s] = pv;
          a.TP[((size_t)i * G + g) * 2048 + lane * 16 + s] = top;
        }
        if (tid >= 128 && tid < 131)
          a.CT[((size_t)i * G + g) * 64 + (tid - 128) * 16 + s] = (tid == 128) ? apush : (tid == 129) ? apop : anoop;
        // one-hot rows of the fed token behind the stack-top rows (weight-gradient B operand, V <= 64)
        if (tid >= 64 && tid < 128)
          a.TP[((size_t)i * G + g) * 2048 + (size_t)tid * 16 + s] = (rowok && misc[8 + k] == tid - 64) ? 1.0f : 0.0f;
      }
    }
    if (!(ok = FC_WAIT())) break;                // A: tokens and stack tops of all 16 sequences visible

    FC_TICK(4)
    // ================================================================ members: gates of the own 16 units
    {
      // K split: waves 0-3 carry the r and z tiles, waves 4-7 the n tile (input and hidden part apart)
      const int kw = wave & 3;
      const int kq0 = kw * (KG / 4);
      f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
      // every exchanged operand of this wave is requested up front: one L2 round trip, not one per k group
      const float4 xt = xld4(rec, (unsigned)(b * 128 + 16 * kw + 4 * q) * 4u);       // stack-top k group kw
      float4 xs[HM];
#pragma unroll
      for (int m = 0; m < HM; ++m) xs[m] = xld4(hin, (unsigned)((kq0 + m) * 256 + lane * 4) * 4u);
      float tkf = 0.0f;
      if (wave == 0) tkf = xld1(rec, (unsigned)(b * 128 + 64) * 4u);                  // token fed to row b at this step
      if (wave < 4) {
#pragma unroll
        for (int m = 0; m < HM; ++m) {
          const int kg = kq0 + m;
          const float4 x = xs[m];
          const f32x4 wr = ld4(wl_hh + (size_t)(0 * KG + kg) * 256 + lane * 4), wz = ld4(wl_hh + (size_t)(1 * KG + kg) * 256 + lane * 4);
          acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wr[0], x.x, acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wz[0], x.x, acc1, 0, 0, 0);
          acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wr[1], x.y, acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wz[1], x.y, acc1, 0, 0, 0);
          acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wr[2], x.z, acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wz[2], x.z, acc1, 0, 0, 0);
          acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wr[3], x.w, acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wz[3], x.w, acc1, 0, 0, 0);
        }
        const f32x4 wr = ld4(wl_is + (size_t)(0 * 4 + kw) * 256 + lane * 4), wz = ld4(wl_is + (size_t)(1 * 4 + kw) * 256 + lane * 4);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wr[0], xt.x, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wz[0], xt.x, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wr[1], xt.y, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wz[1], xt.y, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wr[2], xt.z, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wz[2], xt.z, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wr[3], xt.w, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wz[3], xt.w, acc1, 0, 0, 0);
      } else {
#pragma unroll
        for (int m = 0; m < HM; ++m) {            // acc1 = W_hn h
          const int kg = kq0 + m;
          const float4 x = xs[m];
          const f32x4 wn = ld4(wl_hh + (size_t)(2 * KG + kg) * 256 + lane * 4);
          acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wn[0], x.x, acc1, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wn[1], x.y, acc1, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wn[2], x.z, acc1, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wn[3], x.w, acc1, 0, 0, 0);
        }
        const f32x4 wn = ld4(wl_is + (size_t)(2 * 4 + kw) * 256 + lane * 4);   // acc0 = W_in[:, E:] top
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wn[0], xt.x, acc0, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wn[1], xt.y, acc0, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wn[2], xt.z, acc0, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wn[3], xt.w, acc0, 0, 0, 0);
      }
      st4(red + (size_t)(wave * 2 + 0) * 256 + lane * 4, acc0);
      st4(red + (size_t)(wave * 2 + 1) * 256 + lane * 4, acc1);
      // finished flags of the 16 sequences: ONE load per lane of wave 1 (lane s reads the flag of sequence
      // s), combined by a ballot and handed to everybody through LDS behind the barrier below
      if (decode && wave == 1) {
        const int fl = (lane < 16) ? __float_as_int(xld1(rec, (unsigned)(lane * 128 + 65) * 4u)) : 1;
        const int all1 = __all(fl != 0);
        if (lane == 0) misc[4] = (all1 && !(a.flags & PMG_FLAG_NO_EARLY_EXIT)) ? 1 : 0;
      }
      __syncthreads();
      const int alld = decode ? misc[4] : 0;
      if (wave == 0) {
        f32x4 vr = {0.f, 0.f, 0.f, 0.f}, vz = vr, vi = vr, vh = vr;
#pragma unroll
        for (int wv = 0; wv < 4; ++wv) {
          vr += ld4(red + (size_t)(wv * 2 + 0) * 256 + lane * 4);
          vz += ld4(red + (size_t)(wv * 2 + 1) * 256 + lane * 4);
          vi += ld4(red + (size_t)((wv + 4) * 2 + 0) * 256 + lane * 4);
          vh += ld4(red + (size_t)((wv + 4) * 2 + 1) * 256 + lane * 4);
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
        f32x4 r, z, n, hn;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          r[j] = fast_sigmoid(vr[j]);
          z[j] = fast_sigmoid(vz[j]);
          n[j] = fast_tanh(vi[j] + r[j] * vh[j]);
          hn[j] = (hown[j] - n[j]) * z[j] + n[j];
        }
        hown = hn;
        if (!alld) xst4(hout, (unsigned)(cu * 256 + lane * 4) * 4u, hn);
        if (teacher) { gr = r; gz = z; gn = n; gvh = vh; }      // history: stored after the arrive of barrier B
      }
      if (alld) {
        // every row of this cluster has emitted stop_tok: owners fill their rows' tails and all leave
        for (int k = 0; k < NOWN; ++k) {
          const int s = cu + k * C;
          if (s < 16 && row0 + s < B)
            for (int t = i + tid; t < N; t += FC_THREADS) a.tok_out[(size_t)(row0 + s) * N + t] = a.stop_tok;
        }
        break;
      }
    }
    FC_TICK(5)
    FC_ARRIVE();
    if (teacher && wave == 0) {
      float* gp = a.GT[0] + (((size_t)i * G + g) * KG + cu) * 1024 + lane * 4;
      st4(gp, gr); st4(gp + 256, gz); st4(gp + 512, gn); st4(gp + 768, gvh);
      float* hh = a.HH[0] + (((size_t)(i + 1) * G + g) * H + 16 * cu + 4 * q) * 16 + b;
#pragma unroll
      for (int j = 0; j < 4; ++j) hh[j * 16] = hown[j];
    }
    if (owner) {
      // under the wait: what the next step's owner phase needs and the exchange does not carry
      load_chunk(0);
      if (decode && wave == 0) {
        u0 = pmg_uniform(a.seed, (uint64_t)(a.seq_base + row0 + cu), a.offset + (uint64_t)i);
        if (NOWN > 1) u1 = pmg_uniform(a.seed, (uint64_t)(a.seq_base + row0 + cu + C), a.offset + (uint64_t)i);
      }
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
    for (int k = 0; k < NOWN; ++k) {
      const int s = cu + k * C;
      if (s < 16 && row0 + s < B) {
        if (decode && tid == 0) a.len_out[row0 + s] = -1;
        if (teacher) for (int t = tid; t < N; t += FC_THREADS) a.logp_out[(size_t)t * B + row0 + s] = __int_as_float(0x7fc00000);
      }
    }
    return;
  }
  if (decode && tid == 0)
    for (int k = 0; k < NOWN; ++k) {
      const int s = cu + k * C;
      if (s < 16 && row0 + s < B) a.len_out[row0 + s] = k ? mylen1 : mylen0;
    }
  if (a.h_out != nullptr && wave == 0 && row0 + b < B) st4(a.h_out + (size_t)(row0 + b) * H + 16 * cu + 4 * q, hown);
}

size_t fwdc_lds_base_bytes(const FwdArgs* a) {
  const int KG = a->H / 16, nown = (16 + KG - 1) / KG;
  return ((size_t)3 * KG * 256 + 3 * 4 * 256 + 8 * 2 * 256 + 256 + 256 + (size_t)nown * 2 * a->D * a->W + 64) * sizeof(float);
}

bool fwdc_gtab_fits(const FwdArgs* a) { return fwdc_lds_base_bytes(a) + (size_t)a->V * 48 * 4 <= (size_t)160 * 1024; }
size_t fwdc_lds_bytes(const FwdArgs* a) {
  return fwdc_lds_base_bytes(a) + (fwdc_gtab_fits(a) ? (size_t)a->V * 48 * 4 : 0);
}

}  // namespace

// Floats of the exchange region and words of the counter region for G clusters (both modes).
size_t pmg_fwdc_cx_floats(const PmgDims& d) { return (size_t)d.G * ((size_t)2 * d.H * 16 + 16 * 128); }
size_t pmg_fwdc_sync_words(const PmgDims& d) { return (size_t)32 * (1 + d.G); }

// The clustered kernel serves: one layer, hidden 128..512 (a cluster is H/16 <= 32 CUs), at most
// 8 * floor(32 / C) row groups (every cluster resident at once), extras of at most 256 rows, decode
// or teacher-forced mode, and a per-workgroup state that fits the LDS.
bool pmg_fwdc_applicable(const FwdArgs* a) {
  if (a->flags & PMG_FLAG_NO_CLUSTER) return false;
  if (a->mode != PMG_FWD_DECODE && a->mode != PMG_FWD_TEACHER) return false;
  if (a->L != 1 || a->H > 512 || a->NX > 256 || a->CX == nullptr) return false;
  const int C = a->H / 16;
  if (a->G > 8 * (32 / C)) return false;
  return fwdc_lds_base_bytes(a) <= (size_t)160 * 1024;
}

hipError_t pmg_launch_fwdc(const FwdArgs* a0, hipStream_t st) {
  FwdArgs args = *a0;
  args.cluster_gtab_lds = fwdc_gtab_fits(a0) ? 1 : 0;
  const FwdArgs* a = &args;
  const int C = a->H / 16;
  size_t lds = fwdc_lds_bytes(a);
  if (lds < (size_t)81 * 1024) lds = (size_t)81 * 1024;     // more than half a CU: one workgroup per CU
  // counters and the status word start at zero (stream-ordered; 128 B per counter)
  hipError_t e = hipMemsetAsync(a->CXS, 0, (size_t)32 * (1 + a->G) * sizeof(unsigned), st);
  if (e != hipSuccess) return e;
  const int grid = 8 * C * ((a->G + 7) / 8);
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
