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

// Cluster barrier; returns false (workgroup-uniform) when it gave up.  `okw`: one LDS word.
__device__ __forceinline__ bool cluster_barrier(unsigned* ctr, const unsigned target, unsigned* status, int* okw) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's sc1 stores have been written through
  __syncthreads();
  if (threadIdx.x == 0) {
    int ok = 1;
    __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
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

// MODE: PMG_FWD_DECODE or PMG_FWD_TEACHER.  One layer.
template <int MODE>
__global__ __launch_bounds__(FC_THREADS, 2) void pmg_fwdc_kernel(const FwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr bool teacher = (MODE == PMG_FWD_TEACHER);
  constexpr bool decode = (MODE == PMG_FWD_DECODE);
  const int H = a.H, W = a.W, D = a.D, V = a.V, B = a.B, G = a.G, N = a.N, NX = a.NX;
  const int KG = H / 16, C = KG;                 // members per cluster = unit tiles
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
  float* hrow = red + 8 * 2 * 256;                          // [H]   hidden state of the sequence being served
  float* xex = hrow + H;                                    // [256] extras of that sequence
  float* stk = xex + 256;                                   // [NOWN][2][D*W] own sequences' stacks (ping-pong)
  int* misc = reinterpret_cast<int*>(stk + (size_t)NOWN * 2 * DW);   // [0] barrier ok, [1..] scratch

  // ---- exchange buffers of this cluster
  const size_t cxs = (size_t)2 * H * 16 + 16 * 128;         // floats per cluster
  float* cx = a.CX + (size_t)g * cxs;
  const __amdgpu_buffer_rsrc_t hb0 = pmg_rsrc(cx, (unsigned)H * 64u), hb1 = pmg_rsrc(cx + (size_t)H * 16, (unsigned)H * 64u);
  const __amdgpu_buffer_rsrc_t rec = pmg_rsrc(cx + (size_t)2 * H * 16, 16u * 512u);   // [16][128]: top[64] | token, finished
  unsigned* ctr = a.CXS + 32 + (size_t)g * 32;              // counter of this cluster (own 128-B line)
  unsigned* status = a.CXS;                                  // one word for the launch
  unsigned bars = 0;
#define FC_BARRIER() (++bars, cluster_barrier(ctr, (unsigned)C * bars, status, misc))

  // ---- resident weights: the packed forward operands of pmg_pack.hip, tiles 3*cu .. 3*cu+2
  {
    const float* src = a.wbase + a.oWHH[0] / 4 + (size_t)(3 * cu) * KG * 256;
    for (int i = tid * 4; i < 3 * KG * 256; i += FC_THREADS * 4) st4(wl_hh + i, ld4(src + i));
    const float* src2 = a.wbase + a.oWIH[0] / 4 + (size_t)(3 * cu) * 4 * 256;
    for (int i = tid * 4; i < 3 * 4 * 256; i += FC_THREADS * 4) st4(wl_is + i, ld4(src2 + i));
  }
  for (int i = tid; i < NOWN * 2 * DW; i += FC_THREADS) stk[i] = 0.0f;
  if (tid < 64) misc[tid] = 0;

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
  bool ok = FC_BARRIER();                         // h0 of every member visible

  // per owned sequence (k < NOWN): sampling state, identical in the lanes that serve it
  bool fin0 = false, fin1 = false;
  int mylen0 = N, mylen1 = N;
  const f32x4 bhn = (wave == 0) ? ld4(a.BG[0] + 3 * H + 16 * cu + 4 * q) : (f32x4){0.f, 0.f, 0.f, 0.f};

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
      // ---- hidden state of sequence s out of the packed blocks: unit 16*kg + 4*kq + c <- lane s + 16*kq
      for (int t = tid; t < H / 4; t += FC_THREADS) {
        const int kg = t >> 2, kq = t & 3;
        const float4 v = xld4(hin, (unsigned)(kg * 256 + (s + 16 * kq) * 4) * 4u);
        *reinterpret_cast<float4*>(hrow + 16 * kg + 4 * kq) = v;
      }
      __syncthreads();
      // ---- extras = [W_ctrl; W_push; W_out] h + b: 16 lanes per row, float4 along the reduction
      {
        const int l16 = tid & 15;
        for (int n = tid >> 4; n < NX; n += FC_THREADS / 16) {
          const float* wr = (n < 3) ? a.w_ctrl + (size_t)n * H
                                    : (n < 3 + W) ? a.w_push + (size_t)(n - 3) * H : a.w_out + (size_t)(n - 3 - W) * H;
          float acc = 0.0f;
          for (int k4 = 4 * l16; k4 < H; k4 += 64) {
            const f32x4 wv = ld4(wr + k4), hv = ld4(hrow + k4);
            acc = fmaf(wv[0], hv[0], acc); acc = fmaf(wv[1], hv[1], acc);
            acc = fmaf(wv[2], hv[2], acc); acc = fmaf(wv[3], hv[3], acc);
          }
#pragma unroll
          for (int o = 8; o >= 1; o >>= 1) acc += __shfl_xor(acc, o, 16);
          if (l16 == 0) xex[n] = acc + ((n < 3) ? a.b_ctrl[n] : (n < 3 + W) ? a.b_push[n - 3] : a.b_out[n - 3 - W]);
        }
      }
      __syncthreads();
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
            const float u = pmg_uniform(a.seed, (uint64_t)(a.seq_base + row), a.offset + (uint64_t)(i - 1));
            const int nslot = (V + 31) >> 5;
            float carry = 0.0f;
#pragma unroll 1
            for (int ks = 0; ks < nslot; ++ks) {
              const int v = 32 * ks + l5;
              float e = (v < V) ? __expf(lg[v] / T - m) : 0.0f;
#pragma unroll
              for (int o = 1; o <= 16; o <<= 1) {
                const float t = __shfl_up(e, o, 32);
                if (l5 >= o) e += t;
              }
              carry = __shfl(carry + e, 31, 32);
            }
            const float target = u * carry;
            carry = 0.0f;
            tok = -1;
#pragma unroll 1
            for (int ks = 0; ks < nslot; ++ks) {
              const int v = 32 * ks + l5;
              float e = (v < V) ? __expf(lg[v] / T - m) : 0.0f;
#pragma unroll
              for (int o = 1; o <= 16; o <<= 1) {
                const float t = __shfl_up(e, o, 32);
                if (l5 >= o) e += t;
              }
              const float cdf = carry + e;
              const unsigned mk = (unsigned)__ballot((v < V) && (target < cdf));
              if (tok < 0 && mk) tok = 32 * ks + (__ffs(mk) - 1);
              carry = __shfl(cdf, 31, 32);
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
      if (i < N) {
        tok = misc[2];
        // ---- stack controls, push value, stack update (whole stack of this sequence is in LDS)
        const float c0 = xex[0], c1 = xex[1], c2 = xex[2];
        const float m = fmaxf(c0, fmaxf(c1, c2));
        const float e0 = __expf(c0 - m), e1 = __expf(c1 - m), e2 = __expf(c2 - m);
        const float inv = 1.0f / ((e0 + e1) + e2);
        const float apush = e0 * inv, apop = e1 * inv, anoop = e2 * inv;
        // rows that hold data: d <= min(i, D-1) after this update (row d is zero until step d)
        const int dhi = (i < D - 1) ? i : D - 1;
        for (int idx = tid; idx < (dhi + 1) * W; idx += FC_THREADS) {
          const int d = idx / W, w = idx - d * W;
          const float cur = sold[idx];
          const float prev = (d == 0) ? fast_tanh(xex[3 + w]) : sold[idx - W];
          const float nxt = (d + 1 < D) ? sold[idx + W] : 0.0f;
          const float nw = anoop * cur + apush * prev + apop * nxt;
          snew[idx] = nw;
          if (teacher) {
            // history the BPTT reads: only rows that can still reach a stack top (pmg_fwd.hip, "live rows")
            if (d <= N - 1 - i) a.SB[((size_t)(i + 1) * G + g) * a.SW + (size_t)idx * 16 + s] = nw;
            if (d == 0) {
              a.PV[((size_t)i * G + g) * 1024 + w * 16 + s] = prev;
              a.TP[((size_t)i * G + g) * 2048 + w * 16 + s] = nw;
            }
          }
        }
        if (teacher) {
          if (tid < 3) a.CT[((size_t)i * G + g) * 64 + tid * 16 + s] = (tid == 0) ? apush : (tid == 1) ? apop : anoop;
          // one-hot rows of the fed token behind the stack-top rows (weight-gradient B operand, V <= 64)
          if (tid >= 64 && tid < 128)
            a.TP[((size_t)i * G + g) * 2048 + (size_t)tid * 16 + s] = (rowok && tok == tid - 64) ? 1.0f : 0.0f;
        }
        __syncthreads();
        // ---- publish: stack top (zero padded to 64), next input token, finished flag; two store
        // instructions of wave 0, each writing two whole 128-B lines
        if (wave == 0) {
          const float top = (lane < W) ? snew[lane] : 0.0f;
          xst1(rec, (unsigned)(s * 128 + lane) * 4u, top);
          const float meta = (lane == 0) ? __int_as_float(tok) : (lane == 1) ? __int_as_float(misc[3]) : 0.0f;
          xst1(rec, (unsigned)(s * 128 + 64 + lane) * 4u, meta);
        }
      }
      __syncthreads();
    }
    if (i == N) break;
    if (!(ok = FC_BARRIER())) break;             // A: tokens and stack tops of all 16 sequences visible

    // ================================================================ members: gates of the own 16 units
    {
      // K split: waves 0-3 carry the r and z tiles, waves 4-7 the n tile (input and hidden part apart)
      const int kw = wave & 3;
      const int kq0 = kw * (KG / 4), kq1 = kq0 + KG / 4;
      f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
      const float4 xt = xld4(rec, (unsigned)(b * 128 + 16 * kw + 4 * q) * 4u);       // stack-top k group kw
      if (wave < 4) {
        for (int kg = kq0; kg < kq1; ++kg) {
          const float4 x = xld4(hin, (unsigned)(kg * 256 + lane * 4) * 4u);
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
        for (int kg = kq0; kg < kq1; ++kg) {      // acc1 = W_hn h
          const float4 x = xld4(hin, (unsigned)(kg * 256 + lane * 4) * 4u);
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
      // tokens and finished flags of the 16 sequences (wave 0 needs its row's token; everyone the flags)
      int alld = 1;
      if (decode) {
        for (int s2 = 0; s2 < 16; ++s2) alld &= (__float_as_int(xld1(rec, (unsigned)(s2 * 128 + 65) * 4u)) != 0);
        if (a.flags & PMG_FLAG_NO_EARLY_EXIT) alld = 0;
      } else {
        alld = 0;
      }
      __syncthreads();
      if (wave == 0) {
        f32x4 vr = {0.f, 0.f, 0.f, 0.f}, vz = vr, vi = vr, vh = vr;
#pragma unroll
        for (int wv = 0; wv < 4; ++wv) {
          vr += ld4(red + (size_t)(wv * 2 + 0) * 256 + lane * 4);
          vz += ld4(red + (size_t)(wv * 2 + 1) * 256 + lane * 4);
          vi += ld4(red + (size_t)((wv + 4) * 2 + 0) * 256 + lane * 4);
          vh += ld4(red + (size_t)((wv + 4) * 2 + 1) * 256 + lane * 4);
        }
        const int tk = __float_as_int(xld1(rec, (unsigned)(b * 128 + 64) * 4u));      // token fed to row b at this step
        const float* gt = a.GTAB + (size_t)tk * 3 * H + 16 * cu + 4 * q;
        vr += ld4(gt); vz += ld4(gt + H); vi += ld4(gt + 2 * H); vh += bhn;
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
        if (teacher) {
          float* gp = a.GT[0] + (((size_t)i * G + g) * KG + cu) * 1024 + lane * 4;
          st4(gp, r); st4(gp + 256, z); st4(gp + 512, n); st4(gp + 768, vh);
          float* hh = a.HH[0] + (((size_t)(i + 1) * G + g) * H + 16 * cu + 4 * q) * 16 + b;
#pragma unroll
          for (int j = 0; j < 4; ++j) hh[j * 16] = hn[j];
        }
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
    if (!(ok = FC_BARRIER())) break;             // B: the new hidden state of every member visible
  }

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

size_t fwdc_lds_bytes(const FwdArgs* a) {
  const int KG = a->H / 16, nown = (16 + KG - 1) / KG;
  return ((size_t)3 * KG * 256 + 3 * 4 * 256 + 8 * 2 * 256 + a->H + 256 + (size_t)nown * 2 * a->D * a->W + 64) * sizeof(float);
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
  return fwdc_lds_bytes(a) <= (size_t)160 * 1024;
}

hipError_t pmg_launch_fwdc(const FwdArgs* a, hipStream_t st) {
  const int C = a->H / 16;
  size_t lds = fwdc_lds_bytes(a);
  if (lds < (size_t)81 * 1024) lds = (size_t)81 * 1024;     // more than half a CU: one workgroup per CU
  // counters and the status word start at zero (stream-ordered; 128 B per counter)
  hipError_t e = hipMemsetAsync(a->CXS, 0, (size_t)32 * (1 + a->G) * sizeof(unsigned), st);
  if (e != hipSuccess) return e;
  const int grid = 8 * C * ((a->G + 7) / 8);
  static std::atomic<unsigned long long> done_d{0}, done_t{0};
  if (a->mode == PMG_FWD_DECODE) {
    e = pmg_full_lds_once(reinterpret_cast<const void*>(pmg_fwdc_kernel<PMG_FWD_DECODE>), &done_d);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(pmg_fwdc_kernel<PMG_FWD_DECODE>, dim3(grid), dim3(FC_THREADS), lds, st, *a);
  } else {
    e = pmg_full_lds_once(reinterpret_cast<const void*>(pmg_fwdc_kernel<PMG_FWD_TEACHER>), &done_t);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(pmg_fwdc_kernel<PMG_FWD_TEACHER>, dim3(grid), dim3(FC_THREADS), lds, st, *a);
  }
  return hipGetLastError();
}
