// Persistent reward-weighted BPTT through the StackGRU (the recurrent half of
// rl_loss.backward(), reference: reinforce_omics.py:335).  Weight gradients are NOT
// formed here: this kernel walks time backwards, produces the per-step gate / extras
// gradients (DG, DX row slabs) and grad_h0; pmg_wgrad.hip contracts those slabs with
// the saved activations in one large-K GEMM.
//
// Same decomposition as the forward: one workgroup = 16 sequences, 8 waves, D layout.
// Per reverse step p and layer l (top first):
//   dh      = carry[l] (+ W_ih^{l+1,T} dgi^{l+1} from the layer above, this step)
//   dgates  = GRU cell backward (r,z,n,ghn,h_prev saved by the forward)
//   carry'[l] = z*dh + W_hh^T dgh          (MFMA, transposed packed weights)
//   dx        = W_ih^T dgi                 (layer 0: only the stack-top columns)
// then the stack backward (blend, softmax, tanh) and
//   carry'[L-1] += [W_ctrl; W_push; W_out]^T [dctrl_p | dpush_p | dlogits_{p-1}].
// The algorithm is restated line by line in oracle/fused_model.py (forward_backward).
#include "pmg_kernels.h"

#ifndef PMG_BWD_XPRE
#define PMG_BWD_XPRE 1
#endif
// cache-policy bits (aux operand of the buffer instructions: 2 = nt) of the once-touched history
// streams: saved gates (GT) and hidden history (HH) loads, gate-gradient (DG) stores.  Tuning knobs;
// measured neutral at cfg2 (profiles/r02_experiments.md), so the default policy stays.
#ifndef PMG_BWD_AUX_GT
#define PMG_BWD_AUX_GT 0
#endif
#ifndef PMG_BWD_AUX_HH
#define PMG_BWD_AUX_HH 0
#endif
#ifndef PMG_BWD_AUX_DG
#define PMG_BWD_AUX_DG 0
#endif

// Helper waves (hidden sizes up to 512): four extra waves per workgroup, one per SIMD, that never touch the weight
// stream.  They take everything of a step that is not on the recurrent chain -- gate-gradient slab
// stores (from the LDS chunk buffers), the stack backward of rows >= 2, dlogits, the DX slab -- and
// run it under the products of the eight MFMA waves; only rows 0/1 of the stack backward (they need
// this step's dtop) and the control gradients stay between the products and the extras product.
// A wave's vector-memory operations return in order, which is why the same work as a side job of
// the MFMA waves themselves lost (profiles/r02_experiments.md): a helper's slow loads only delay
// the helper.
#ifndef PMG_BWD_HELP
#define PMG_BWD_HELP 1
#endif
#define PMG_HELP_THREADS 256
#ifndef PMG_HELP_PRIO
#define PMG_HELP_PRIO 3
#endif
#ifndef PMG_HBLK
#define PMG_HBLK 4      // stack rows per load burst of a helper thread
#endif

namespace {

__device__ __forceinline__ f32x4 ld4(const float* p) {
  const float4 v = *reinterpret_cast<const float4*>(p);
  f32x4 r; r.x = v.x; r.y = v.y; r.z = v.z; r.w = v.w; return r;
}
__device__ __forceinline__ void st4(float* p, const f32x4 v) {
  *reinterpret_cast<float4*>(p) = make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ f32x4 bl4(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
  const float4 v = pmg_bload(r, voff, soff);
  f32x4 o; o.x = v.x; o.y = v.y; o.z = v.z; o.w = v.w; return o;
}
template <int AUX> __device__ __forceinline__ f32x4 bl4a(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
  const pmg_u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, AUX);
  f32x4 o; o.x = __uint_as_float(v.x); o.y = __uint_as_float(v.y); o.z = __uint_as_float(v.z); o.w = __uint_as_float(v.w);
  return o;
}

// buffer resource from values the compiler cannot prove wave-uniform by itself (they are): pinned
// to SGPRs so that the accesses are not wrapped in waterfall loops
__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc_uniform(const float* base, unsigned bytes) {
  const unsigned long long a = reinterpret_cast<unsigned long long>(base);
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
  return pmg_rsrc(reinterpret_cast<const float*>(((unsigned long long)hi << 32) | lo), __builtin_amdgcn_readfirstlane(bytes));
}

// acc[t] += Wt[tile t][24 groups per chunk] * chunk columns.  24 reduction groups per chunk: 8 per
// piece; pcN = LDS column block (of 128) holding piece N of this product.  Weights through the
// buffer resource (voff = lane*16, soff = first tile, first group, ts = bytes per tile).  Two
// register sets + sched_barriers keep the next group's loads in flight under the current MFMAs.
// The same product over `nch` consecutive chunks (chunk buffers 16*PMG_CP floats apart, weights
// contiguous: 24 groups per chunk) as ONE software-pipelined reduction: one exposed weight load at
// the start instead of one per chunk.
template <int NT>
__device__ __forceinline__ void gemm_chunks(f32x4 (&acc)[NT], const __amdgpu_buffer_rsrc_t rs, const unsigned voff,
                                            const unsigned soff, const unsigned ts, const float* xb, const int nch,
                                            const int pc0, const int pc1, const int pc2) {
  float4 wa[NT], wb[NT];
  const int nrg = 24 * nch;
#define PMG_LOAD(dst, rgi) \
  _Pragma("unroll") for (int t = 0; t < NT; ++t) dst[t] = pmg_bload(rs, voff, soff + t * ts + (unsigned)(rgi) * 1024u);
#define PMG_STEP(wc, C)                                                                  \
  _Pragma("unroll") for (int t = 0; t < NT; ++t)                                         \
      acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(wc[t].C, x.C, acc[t], 0, 0, 0);
  PMG_LOAD(wa, 0)
  int cc = 0, rg = 0;  // chunk and group inside the chunk of the running index
#if PMG_BWD_XPRE
  // B operand (LDS) one half group ahead of its MFMAs
  const float* xp = xb + pc0 * PMG_CHUNK;
  float4 xa = *reinterpret_cast<const float4*>(xp);
#pragma unroll 1
  for (int rgl = 0; rgl < nrg; rgl += 2) {
    PMG_LOAD(wb, rgl + 1)
    const float4 xn = *reinterpret_cast<const float4*>(xp + 16);
    __builtin_amdgcn_sched_barrier(0);
    {
      const float4 x = xa;
      PMG_STEP(wa, x) PMG_STEP(wa, y) PMG_STEP(wa, z) PMG_STEP(wa, w)
    }
    __builtin_amdgcn_sched_barrier(0);
    {
      const int rn = (rgl + 2 < nrg) ? rgl + 2 : rgl;
      PMG_LOAD(wa, rn)
      if (rgl + 2 < nrg) {
        rg += 2;
        if (rg == 24) { rg = 0; ++cc; }
      }
      const int piece = rg >> 3;
      const int pc = (piece == 0) ? pc0 : (piece == 1) ? pc1 : pc2;
      xp = xb + (size_t)cc * 16 * PMG_CP + pc * PMG_CHUNK + (rg & 7) * 16;
      xa = *reinterpret_cast<const float4*>(xp);
    }
    __builtin_amdgcn_sched_barrier(0);
    {
      const float4 x = xn;
      PMG_STEP(wb, x) PMG_STEP(wb, y) PMG_STEP(wb, z) PMG_STEP(wb, w)
    }
    __builtin_amdgcn_sched_barrier(0);
  }
#else
#pragma unroll 1
  for (int rgl = 0; rgl < nrg; rgl += 2) {
    const int piece = rg >> 3;
    const int pc = (piece == 0) ? pc0 : (piece == 1) ? pc1 : pc2;
    const float* xp = xb + (size_t)cc * 16 * PMG_CP + pc * PMG_CHUNK + (rg & 7) * 16;
    PMG_LOAD(wb, rgl + 1)
    __builtin_amdgcn_sched_barrier(0);
    {
      const float4 x = *reinterpret_cast<const float4*>(xp);
      PMG_STEP(wa, x) PMG_STEP(wa, y) PMG_STEP(wa, z) PMG_STEP(wa, w)
    }
    __builtin_amdgcn_sched_barrier(0);
    {
      const int rn = (rgl + 2 < nrg) ? rgl + 2 : rgl;
      PMG_LOAD(wa, rn)
    }
    __builtin_amdgcn_sched_barrier(0);
    {
      const float4 x = *reinterpret_cast<const float4*>(xp + 16);
      PMG_STEP(wb, x) PMG_STEP(wb, y) PMG_STEP(wb, z) PMG_STEP(wb, w)
    }
    __builtin_amdgcn_sched_barrier(0);
    rg += 2;
    if (rg == 24) { rg = 0; ++cc; }
  }
#endif
#undef PMG_STEP
#undef PMG_LOAD
}

// acc[t] += Wt[tile t][0..KG) * x[b][0..16*KG)   (plain contiguous reduction)
template <int NT>
__device__ __forceinline__ void gemm_lin(f32x4 (&acc)[NT], const __amdgpu_buffer_rsrc_t rs, const unsigned voff,
                                         const unsigned soff, const unsigned ts, const float* xrow, const int kg0,
                                         const int kg1) {
#pragma unroll 2
  for (int kg = kg0; kg < kg1; ++kg) {
    const float4 x = *reinterpret_cast<const float4*>(xrow + 16 * kg);
    float4 wc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) wc[t] = pmg_bload(rs, voff, soff + t * ts + (unsigned)kg * 1024u);
#define PMG_STEP(C)                                                                      \
  _Pragma("unroll") for (int t = 0; t < NT; ++t)                                         \
      acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(wc[t].C, x.C, acc[t], 0, 0, 0);
    PMG_STEP(x) PMG_STEP(y) PMG_STEP(z) PMG_STEP(w)
#undef PMG_STEP
  }
}

// MULTI: more than one GRU layer (only then is the W_ih^T product of a layer carried to the layer
// below; a single-layer kernel does not hold those accumulators).
// NCB: chunk buffers in LDS = chunks (of 128 hidden units) per round of cell backward + products; a
// divisor of UPW, as large as fits next to the extras tile (a wide vocabulary leaves room for fewer).
template <int UPW, int NCB, bool MULTI, bool HELP>
__global__ __launch_bounds__(HELP ? PMG_THREADS + PMG_HELP_THREADS : PMG_THREADS, HELP ? 3 : 2) void pmg_bwd_kernel(const BwdArgs a) {
  static_assert(UPW % NCB == 0, "chunks per round must divide the chunks of a layer");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int H = a.H, XP = a.XP, W = a.W, D = a.D, V = a.V, L = a.L, B = a.B, G = a.G, N = a.N;
  const int UT = H / 16, NXT = a.NXT;
  const int dbg = PMG_DBG(a.dbg);
  float* cb = reinterpret_cast<float*>(smem);   // [NCB][16][PMG_CP]
  float* dxl = cb + NCB * 16 * PMG_CP;          // [16][XP]   dctrl | dpush | dlogits
  float* dtl = dxl + 16 * XP;                   // [16][PMG_XTP] dtop
  float* red = dtl + 16 * PMG_XTP;              // [8][4][12]
  float* lsl = red + PMG_WAVES * 48;            // [16] loss terms

  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63;
  const int b = lane & 15, q = lane >> 4;
  const int g = blockIdx.x;
  const int row0 = g * PMG_ROWS;
  const unsigned KG3 = 3u * (unsigned)H / 16u;  // reduction groups of the gate products
  const __amdgpu_buffer_rsrc_t wrs = pmg_rsrc(a.wbase, a.wbytes);
  const unsigned voff = (unsigned)lane * 16u;

  constexpr int NTHR = HELP ? PMG_THREADS + PMG_HELP_THREADS : PMG_THREADS;
  const bool helper = HELP && wave >= PMG_WAVES;  // wave-uniform
  if (a.LAYW != nullptr && *a.LAYW != 0u) {
    // the clustered forward left the history in the per-sequence layout (FwdArgs::LAYW) and this launch was
    // told PMG_FLAG_NO_CLUSTER: no plausible-looking numbers out of a workspace this kernel cannot read
    const float nanv = __int_as_float(0x7fc00000);
    for (int i = tid; i < 16 * L * H; i += NTHR) {
      const int r = i / (L * H), rem = i - r * (L * H);
      if (row0 + r < B) a.grad_h0[((size_t)(rem / H) * B + row0 + r) * H + rem % H] = nanv;
    }
    if (tid == 0) a.LOSSP[g] = nanv;
    return;
  }
#if PMG_HELP_PRIO
  // a helper shares its SIMD with two MFMA waves that always have an instruction ready; without
  // priority its few VALU / LDS / memory instructions wait for gaps in their issue
  if (helper) __builtin_amdgcn_s_setprio(PMG_HELP_PRIO);
#endif
  const int ht = tid - PMG_THREADS;               // helper thread index
  const int lt = HELP ? ht : tid;                 // index among the threads that own the per-row scalars

  float* dhc = a.DHC + (size_t)g * L * H * 16;
  // One layer: the hidden-gradient carry never leaves the registers of the MFMA waves.  More
  // layers: it goes through a global slab (own elements only: each lane re-reads what it wrote).
  f32x4 car[UPW];
#pragma unroll
  for (int u = 0; u < UPW; ++u) car[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
  // extras groups below KX0 hold dctrl / dpush rows (need this step's dtop); the groups from KX0 on
  // hold only dlogits rows, which are there before the step's products start
  const int KX0 = (3 + W + 15) / 16;
  if (MULTI && !helper)
    for (int l = 0; l < L; ++l)
#pragma unroll
      for (int u = 0; u < UPW; ++u) {
        const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
        st4(dhc + ((size_t)(l * UT + wave * UPW + u)) * 256 + lane * 4, z4);
      }
  for (int idx = tid; idx < 16 * XP; idx += NTHR) dxl[idx] = 0.0f;
  for (int idx = tid; idx < 16 * PMG_XTP; idx += NTHR) dtl[idx] = 0.0f;
  float lossacc = 0.0f;  // threads lt = 0..15: sum_p R_b * logp[p][b]
  float myR = 0.0f;
  if (lt >= 0 && lt < 16) myR = (row0 + lt < B) ? a.reward[row0 + lt] : 0.0f;
  const float rowR = (row0 + (tid & 15) < B) ? a.reward[row0 + (tid & 15)] : 0.0f;  // reward of row tid&15
  __syncthreads();

#ifdef PMG_PROFILE
  // -DPMG_PROFILE build + PMG_DEBUG bit 0x8000: per-phase cycle counters of workgroup 0, printed at the end
  const bool prof = (dbg & 0x8000) != 0;
  long long tph[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  long long tlast = prof ? (long long)__builtin_amdgcn_s_memtime() : 0;
#define PMG_TICK(k) if (prof) { const long long tn_ = (long long)__builtin_amdgcn_s_memtime(); tph[k] += tn_ - tlast; tlast = tn_; }
#else
#define PMG_TICK(k)
#endif

  // p == N is the virtual step that only injects dlogits_{N-1}.  The helper waves run their own
  // copy of the step loop (same barriers in the same order), so that neither role holds the other's
  // registers.
  if (HELP && helper) {
   for (int p = N; p >= 0; --p) {
    const bool virt = (p == N);
    {
      const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
      const int hb = ht & 15;
      // ---- dlogits_{p-1} and the loss terms: nothing recurrent
      {
        const int dl_row = row0 + hb;
        int y = 0;
        if (p >= 1 && dl_row < B) y = min(max((int)a.tokens[(size_t)p * B + dl_row], 0), V - 1);  // clamped like the forward
        float pr[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        const float* prp = a.PR + (((size_t)(p > 0 ? p - 1 : 0) * G + g) * a.VP) * 16 + hb;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int v = (ht >> 4) + 16 * k;
          if (p >= 1 && v < V) pr[k] = pmg_ld1h<4>(prp + (size_t)v * 16);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int v = (ht >> 4) + 16 * k;
          if (v < V) dxl[hb * XP + 3 + W + v] = (p >= 1) ? a.scale * rowR * (pr[k] - ((v == y) ? 1.0f : 0.0f)) : 0.0f;
        }
        for (int v = (ht >> 4) + 64; v < V; v += PMG_HELP_THREADS / 16) {  // V > 64
          float val = 0.0f;
          if (p >= 1) val = a.scale * rowR * (pmg_ld1h<4>(prp + (size_t)v * 16) - ((v == y) ? 1.0f : 0.0f));
          dxl[hb * XP + 3 + W + v] = val;
        }
        if (p >= 1 && ht < 16) lossacc += myR * a.LP[((size_t)(p - 1) * G + g) * 16 + ht];
      }
      PMG_TICK(0)
      float ct_h[3] = {0.0f, 0.0f, 0.0f};
      if (!virt && ht < 16) {
        const float* ct = a.CT + ((size_t)p * G + g) * 64;
        ct_h[0] = ct[ht]; ct_h[1] = ct[16 + ht]; ct_h[2] = ct[32 + ht];
      }
      // ---- stack backward, part 1: what does not need this step's dtop.  Thread ht owns column
      // (w = ht >> 2, sequences 4*(ht & 3)..+3) over all rows, so every gradient row it reads was
      // written by itself one step earlier.
      // (threads past the last column run the same code on offsets outside every buffer: they read
      // zeros and their stores are dropped, so the control flow below is uniform)
      const int sbq = ht & 3, sw = ht >> 2;
      const bool scol = !virt && sw < W;
      const int swc = sw < W ? sw : W - 1;
      f32x4 sn = zero4, sp = zero4, sq = zero4;
      f32x4 apush = zero4, apop = zero4, anoop = zero4, pv = zero4;
      f32x4 g0in = zero4, g1 = zero4, g2 = zero4, s0 = zero4, s1 = zero4, dS0 = zero4, dS1 = zero4;
      int nrows = 0;
      float* dSout = a.DS + ((size_t)(p & 1) * G + g) * a.SW;
      if (!virt) {
        const float* ctp = a.CT + ((size_t)p * G + g) * 64 + 4 * sbq;
        apush = ld4(ctp); apop = ld4(ctp + 16); anoop = ld4(ctp + 32);
        pv = ld4(a.PV + ((size_t)p * G + g) * 1024 + swc * 16 + 4 * sbq);
      }
      if (!virt) {
        for (int l = L - 1; l >= 0; --l) {
          const __amdgpu_buffer_rsrc_t dgr = pmg_rsrc(a.DG[l] + (((size_t)p * G + g) * 4 * H) * 16, (unsigned)H * 256u);
#pragma unroll
          for (int rd = 0; rd < UPW / NCB; ++rd) {
            if (rd > 0 || (MULTI && l < L - 1)) __syncthreads();  // mirrors the MFMA waves
            __syncthreads();                                       // chunk buffers of this round filled
            // gate-gradient row slabs [gate row][16 sequences] out of the chunk buffers
            // ([sequence][piece*128 + column]): 4 x ds_read_b32 + one 16-byte store, 1 KB contiguous per wave
            const int x0 = ht >> 2;
            const float* src0 = cb + (size_t)(4 * sbq) * PMG_CP + x0;
            const unsigned vo0 = (unsigned)(16 * ((x0 >> 4) * UPW) + (x0 & 15)) * 64u + (unsigned)sbq * 16u;
            const unsigned vo1 = vo0 + (unsigned)(16 * 4 * UPW) * 64u;   // column x0 + 64: four waves further
#pragma unroll
            for (int i = 0; i < NCB * 8; ++i) {
              const int hi = i & 1, gate = (i >> 1) & 3, cc = i >> 3;
              const float* src = src0 + (size_t)cc * 16 * PMG_CP + gate * PMG_CHUNK + 64 * hi;
              pmg_u32x4 u;
              u.x = __float_as_uint(src[0]); u.y = __float_as_uint(src[PMG_CP]);
              u.z = __float_as_uint(src[2 * PMG_CP]); u.w = __float_as_uint(src[3 * PMG_CP]);
              __builtin_amdgcn_raw_buffer_store_b128(u, dgr, hi ? vo1 : vo0,
                                                     (unsigned)(rd * NCB + cc) * 1024u + (unsigned)gate * (unsigned)H * 64u,
                                                     PMG_BWD_AUX_DG);
            }
          }
        }
      }
      PMG_TICK(1)
      // Rows outside the live ranges (see the forward) read as zero through the bounds of the buffer
      // resources instead of per-element selects, addresses are one lane-constant offset plus scalar
      // row offsets, and the burst is unrolled so that the row window (previous / this / next) is
      // renamed, not moved: a helper shares its SIMD's ALUs with two waves of fp32 MFMAs, and every
      // vector instruction it issues is time taken from them.
      const unsigned rs4 = (unsigned)W * 64u;            // bytes per stack row of 16 sequences
      const unsigned cvo = sw < W ? (unsigned)(sw * 16 + 4 * sbq) * 4u : 0x40000000u;
      const __amdgpu_buffer_rsrc_t ors = rsrc_uniform(dSout, (unsigned)D * rs4);
      if (!virt) {
        const int mn = (p < N - p) ? p : N - p;
        nrows = (mn > D - 1) ? D : mn + 1;
        const int s_lim = (p - 1 < N - p) ? p - 1 : N - p;
        const int g_lim = (p < N - 1 - p) ? p : N - 1 - p;
        const int s_hi = s_lim < D - 1 ? s_lim : D - 1;
        const int g_hi = g_lim < D - 1 ? g_lim : D - 1;
        const float* Sprev = a.SB + ((size_t)p * G + g) * a.SW;
        const float* dSin = a.DS + ((size_t)((p + 1) & 1) * G + g) * a.SW;
        const __amdgpu_buffer_rsrc_t srs = rsrc_uniform(Sprev, (p == 0 || s_hi < 0) ? 0u : (unsigned)(s_hi + 1) * rs4);
        const __amdgpu_buffer_rsrc_t grs = rsrc_uniform(dSin, (p == N - 1 || g_hi < 0) ? 0u : (unsigned)(g_hi + 1) * rs4);
        // (only the per-lane offset of a buffer access is checked against the bounds, so the row
        // offsets go there: one add per row)
        constexpr int SAUX = (PMG_NT & 4) ? 2 : 0;
        const unsigned v1 = cvo + rs4, v2 = v1 + rs4;
        g0in = bl4(grs, cvo, 0u); g1 = bl4(grs, v1, 0u); g2 = bl4(grs, v2, 0u);
        s0 = bl4a<SAUX>(srs, cvo, 0u); s1 = bl4a<SAUX>(srs, v1, 0u);
        const f32x4 s2 = bl4a<SAUX>(srs, v2, 0u);
        if (nrows >= 2) {   // row 1: sums now, its gradient row once dtop is there
          sn += g1 * s1;
          sp += g1 * s0;
          sq += g1 * s2;
        }
        f32x4 gm = g1, gc = g2, sm = s1, sc = s2;
        const int nb = (nrows - 2 + PMG_HBLK - 1) / PMG_HBLK;     // bursts covering rows 2 .. nrows-1
        unsigned vr = v2;                                          // first row of the burst
        for (int it = 0; it < nb; ++it) {
          f32x4 gv[PMG_HBLK], sv[PMG_HBLK];
          unsigned vo[PMG_HBLK + 1];
          vo[0] = vr;
#pragma unroll
          for (int k = 0; k < PMG_HBLK; ++k) {
            vo[k + 1] = vo[k] + rs4;
            gv[k] = bl4(grs, vo[k + 1], 0u);
            sv[k] = bl4a<SAUX>(srs, vo[k + 1], 0u);
          }
#pragma unroll
          for (int k = 0; k < PMG_HBLK; ++k) {
            // row r = 2 + it*PMG_HBLK + k; rows >= nrows have a zero gradient (bounds), and their
            // stores land in rows nobody reads or beyond the slab (dropped)
            sn += gc * sc;
            sp += gc * sm;
            sq += gc * sv[k];
            const f32x4 dprev = anoop * gc + apush * gv[k] + apop * gm;
            pmg_u32x4 u;
            u.x = __float_as_uint(dprev[0]); u.y = __float_as_uint(dprev[1]); u.z = __float_as_uint(dprev[2]); u.w = __float_as_uint(dprev[3]);
            __builtin_amdgcn_raw_buffer_store_b128(u, ors, vo[k], 0u, 0);
            gm = gc; gc = gv[k]; sm = sc; sc = sv[k];
          }
          vr = vo[PMG_HBLK];
        }
      }
      PMG_TICK(2)
      if (!virt) {
        __syncthreads();  // dtop visible
        PMG_TICK(3)
        // ---- stack backward, part 2: rows 0 and 1
        if (scol) {
          f32x4 g0;
#pragma unroll
          for (int j = 0; j < 4; ++j) g0[j] = g0in[j] + dtl[(4 * sbq + j) * PMG_XTP + sw];
          sn += g0 * s0;
          sp += g0 * pv;
          sq += g0 * s1;
          dS0 = anoop * g0 + apush * g1;
          dS1 = anoop * g1 + apush * g2 + apop * g0;
#pragma unroll
          for (int j = 0; j < 4; ++j) dxl[(4 * sbq + j) * XP + 3 + sw] = apush[j] * g0[j] * (1.0f - pv[j] * pv[j]);
        }
#pragma unroll
        for (int o = 4; o < 64; o <<= 1) {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            sn[j] += __shfl_xor(sn[j], o);
            sp[j] += __shfl_xor(sp[j], o);
            sq[j] += __shfl_xor(sq[j], o);
          }
        }
        if (lane < 4) {
          float* rp = red + ((wave - PMG_WAVES) * 4 + lane) * 12;
#pragma unroll
          for (int j = 0; j < 4; ++j) { rp[j] = sp[j]; rp[4 + j] = sq[j]; rp[8 + j] = sn[j]; }
        }
      }
      PMG_TICK(4)
      __syncthreads();
      if (!virt && ht < 16) {
        const int bq = ht >> 2, j = ht & 3;
        float dpu = 0.f, dpo = 0.f, dno = 0.f;
        for (int wv = 0; wv < PMG_HELP_THREADS / 64; ++wv) {
          const float* rp = red + (wv * 4 + bq) * 12;
          dpu += rp[j]; dpo += rp[4 + j]; dno += rp[8 + j];
        }
        const float a0 = ct_h[0], a1 = ct_h[1], a2 = ct_h[2];
        const float s = a0 * dpu + a1 * dpo + a2 * dno;
        dxl[ht * XP + 0] = a0 * (dpu - s);
        dxl[ht * XP + 1] = a1 * (dpo - s);
        dxl[ht * XP + 2] = a2 * (dno - s);
      }
      __syncthreads();
      PMG_TICK(5)
      if (!virt) {
        // rows 0 and 1 of the stack gradient (kept back so that no store was pending at the two
        // barriers on the recurrent chain)
        pmg_u32x4 u0, u1;
        u0.x = __float_as_uint(dS0[0]); u0.y = __float_as_uint(dS0[1]); u0.z = __float_as_uint(dS0[2]); u0.w = __float_as_uint(dS0[3]);
        u1.x = __float_as_uint(dS1[0]); u1.y = __float_as_uint(dS1[1]); u1.z = __float_as_uint(dS1[2]); u1.w = __float_as_uint(dS1[3]);
        __builtin_amdgcn_raw_buffer_store_b128(u0, ors, cvo, 0u, 0);
        __builtin_amdgcn_raw_buffer_store_b128(u1, ors, nrows >= 2 ? cvo + rs4 : 0x40000000u, 0u, 0);
      }
      {
        // DX slab [extras row][16 sequences] for the weight-gradient GEMM, next to the extras product
        float* dxg = a.DX + ((size_t)p * G + g) * NXT * 256;
        for (int i4 = ht; i4 < NXT * 64; i4 += PMG_HELP_THREADS) {
          const int nrow = i4 >> 2, bq = i4 & 3;
          f32x4 v;
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] = dxl[(4 * bq + j) * XP + nrow];
          pmg_st4h<8>(dxg + nrow * 16 + 4 * bq, v);
        }
      }
      PMG_TICK(6)
      __syncthreads();  // dxl / dtl / red reusable
      PMG_TICK(7)
    }
   }
  } else {
  for (int p = N; p >= 0; --p) {
    const bool virt = (p == N);
    // dlogits_{p-1} inputs (probabilities, target token; nothing recurrent) are requested here, one
    // step's worth of products before they are used, so their latency is off the serial chain.
    // Thread tid covers row tid&15 and vocabulary entries tid>>4 and 32 + (tid>>4); larger
    // vocabularies load the rest in place.
    const int dl_row = row0 + (tid & 15);
    int dl_y = 0;
    float dl_pr[2] = {0.0f, 0.0f};
    if (!HELP && p >= 1) {
      dl_y = (dl_row < B) ? min(max((int)a.tokens[(size_t)p * B + dl_row], 0), V - 1) : 0;  // clamped like the forward
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int v = (tid >> 4) + 32 * k;
        if (v < V) dl_pr[k] = pmg_ld1h<4>(a.PR + (((size_t)(p - 1) * G + g) * a.VP + v) * 16 + (tid & 15));
      }
    }
    float ct_a[3] = {0.0f, 0.0f, 0.0f};  // stack controls of row tid (threads 0..15), same early request
    if (!HELP && !virt && tid < 16) {
      const float* ct = a.CT + ((size_t)p * G + g) * 64;
      ct_a[0] = ct[tid]; ct_a[1] = ct[16 + tid]; ct_a[2] = ct[32 + tid];
    }
    if (!virt) {
      f32x4 dxa[UPW];
#pragma unroll
      for (int u = 0; u < UPW; ++u) dxa[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
      for (int l = L - 1; l >= 0; --l) {
        f32x4 dh[UPW], acc_hh[UPW], acc_dx[UPW];
        f32x4 acc_top[1];
        acc_top[0] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < UPW; ++u) {
          dh[u] = MULTI ? ld4(dhc + ((size_t)(l * UT + wave * UPW + u)) * 256 + lane * 4) : car[u];
          if (MULTI && l < L - 1) dh[u] += dxa[u];
          acc_hh[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
          acc_dx[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
        // history and gradient slabs of this (step, row group) through buffer resources: one
        // lane-constant byte offset in a VGPR, everything else is a wave-uniform offset
        const __amdgpu_buffer_rsrc_t gtr = pmg_rsrc(a.GT[l] + (((size_t)p * G + g) * UT) * 1024, (unsigned)H * 256u);
        const __amdgpu_buffer_rsrc_t hhr = pmg_rsrc(a.HH[l] + (((size_t)p * G + g) * H) * 16, (unsigned)H * 64u);
        const __amdgpu_buffer_rsrc_t dgr = pmg_rsrc(a.DG[l] + (((size_t)p * G + g) * 4 * H) * 16, (unsigned)H * 256u);
        const unsigned vrow = (unsigned)((16 * wave * UPW + 4 * q) * 16 + b) * 4u;  // element (feature 4q of this wave's first tile, row b)
        // The cell backward of NCB chunks (128 hidden units each) runs first, with every history
        // load of the round in flight at once, into NCB chunk buffers; then one barrier and the
        // products over all of them back to back (one long reduction, no barriers in between).
#pragma unroll
        for (int rd = 0; rd < UPW / NCB; ++rd) {
          if (rd > 0 || (MULTI && l < L - 1)) __syncthreads();  // chunk buffers free again
          f32x4 hr[NCB], hz[NCB], hn[NCB], hg[NCB], hp[NCB];
#pragma unroll
          for (int cc = 0; cc < NCB; ++cc) {
            const unsigned c = (unsigned)(rd * NCB + cc);
            const unsigned gs = ((unsigned)(wave * UPW) + c) * 4096u;  // bytes: tile * 1024 floats
            hr[cc] = bl4a<PMG_BWD_AUX_GT>(gtr, voff, gs); hz[cc] = bl4a<PMG_BWD_AUX_GT>(gtr, voff, gs + 1024u);
            hn[cc] = bl4a<PMG_BWD_AUX_GT>(gtr, voff, gs + 2048u); hg[cc] = bl4a<PMG_BWD_AUX_GT>(gtr, voff, gs + 3072u);
#pragma unroll
            for (int j = 0; j < 4; ++j)
              hp[cc][j] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(hhr, vrow, (c * 256u + j * 16u) * 4u, PMG_BWD_AUX_HH));
          }
#pragma unroll
          for (int cc = 0; cc < NCB; ++cc) {
            const int c = rd * NCB + cc;
            const f32x4 r = hr[cc], z = hz[cc], n = hn[cc], ghn = hg[cc];
            const f32x4 d = dh[c];
            f32x4 dr_pre, dz_pre, dn_pre, dghn;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const float dn = d[j] * (1.0f - z[j]);
              const float dz = d[j] * (hp[cc][j] - n[j]);
              dn_pre[j] = dn * (1.0f - n[j] * n[j]);
              const float dr = dn_pre[j] * ghn[j];
              dghn[j] = dn_pre[j] * r[j];
              dz_pre[j] = dz * z[j] * (1.0f - z[j]);
              dr_pre[j] = dr * r[j] * (1.0f - r[j]);
            }
            acc_hh[c] += d * z;
            float* cbw = cb + (size_t)cc * 16 * PMG_CP + b * PMG_CP + 16 * wave + 4 * q;
            st4(cbw, dr_pre); st4(cbw + PMG_CHUNK, dz_pre); st4(cbw + 2 * PMG_CHUNK, dn_pre);
            st4(cbw + 3 * PMG_CHUNK, dghn);
            if (!HELP && !(dbg & 0x1000))
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const unsigned so = ((unsigned)c * 256u + j * 16u) * 4u, gH = (unsigned)H * 64u;
              __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(dr_pre[j]), dgr, vrow, so, PMG_BWD_AUX_DG);
              __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(dz_pre[j]), dgr, vrow, so + gH, PMG_BWD_AUX_DG);
              __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(dn_pre[j]), dgr, vrow, so + 2u * gH, PMG_BWD_AUX_DG);
              __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(dghn[j]), dgr, vrow, so + 3u * gH, PMG_BWD_AUX_DG);
            }
          }
          PMG_TICK(0)
          __syncthreads();
          PMG_TICK(1)
          {
            // ---- products over the 3*128 reduction rows of each of this round's chunks, one
            // continuous reduction per product
            const float* xb = cb + b * PMG_CP + 4 * q;
            const unsigned c0 = (unsigned)(rd * NCB);
            {
              const unsigned so = a.oWHHt[l] + ((unsigned)(wave * UPW) * KG3 + c0 * 24u) * 1024u;
              if (!(dbg & 0x400)) gemm_chunks<UPW>(acc_hh, wrs, voff, so, KG3 * 1024u, xb, NCB, 0, 1, 3);
            }
            if (MULTI && l > 0) {
              const unsigned so = a.oWIHt[l] + ((unsigned)(wave * UPW) * KG3 + c0 * 24u) * 1024u;
              gemm_chunks<UPW>(acc_dx, wrs, voff, so, KG3 * 1024u, xb, NCB, 0, 1, 2);
            } else if (wave < PMG_KS / 16) {
              const unsigned so = a.oWIHt[0] + ((unsigned)wave * KG3 + c0 * 24u) * 1024u;
              if (!(dbg & 0x800)) gemm_chunks<1>(acc_top, wrs, voff, so, KG3 * 1024u, xb, NCB, 0, 1, 2);
            }
          }
          PMG_TICK(2)
        }
#pragma unroll
        for (int u = 0; u < UPW; ++u) {
          if (MULTI) {
            st4(dhc + ((size_t)(l * UT + wave * UPW + u)) * 256 + lane * 4, acc_hh[u]);
            dxa[u] = acc_dx[u];
          } else {
            car[u] = acc_hh[u];
          }
        }
        if (l == 0 && wave < PMG_KS / 16) st4(dtl + b * PMG_XTP + 16 * wave + 4 * q, acc_top[0]);
      }
      __syncthreads();  // dtop visible; chunk buffers free
      PMG_TICK(3)
      if (HELP && !MULTI) {
        // the dlogits part of the extras product, while the helpers finish rows 0/1 of the stack
        // backward and the control gradients
        const unsigned so = a.oWXt + (unsigned)(wave * UPW) * (unsigned)NXT * 1024u;
        if (!(dbg & 0x200)) gemm_lin<UPW>(car, wrs, voff, so, (unsigned)NXT * 1024u, dxl + b * XP + 4 * q, KX0, NXT);
      }
    }

    // ------------------------------------------------------------ stack backward
    if (!HELP && !virt && !(dbg & 0x100)) {
      // Live rows (see the forward): S_{p-1} holds data in rows <= min(p-1, N-p), the incoming
      // gradient (w.r.t. S_p) in rows <= min(p, N-1-p); gradients of rows that are constant zero
      // or can no longer reach a stack top are neither needed nor formed.
      const bool full = (dbg & 0x4000) != 0;
      const int mn = (p < N - p) ? p : N - p;
      const int nrows = (full || mn > D - 1) ? D : mn + 1;
      const int s_hi = full ? D - 1 : ((p - 1 < N - p ? p - 1 : N - p) < D - 1 ? (p - 1 < N - p ? p - 1 : N - p) : D - 1);
      const int g_hi = full ? D - 1 : ((p < N - 1 - p ? p : N - 1 - p) < D - 1 ? (p < N - 1 - p ? p : N - 1 - p) : D - 1);
      const int ncol = W * 4;
      int dsplit = PMG_THREADS / ncol;
      if (dsplit < 1) dsplit = 1;
      if (dsplit > nrows) dsplit = nrows;
      const int dch = (nrows + dsplit - 1) / dsplit;
      const float* Sprev = a.SB + ((size_t)p * G + g) * a.SW;
      const float* dSin = a.DS + ((size_t)((p + 1) & 1) * G + g) * a.SW;
      float* dSout = a.DS + ((size_t)(p & 1) * G + g) * a.SW;
      const bool zero_S = (p == 0);
      const bool zero_dS = (p == N - 1);
      const float* ct = a.CT + ((size_t)p * G + g) * 64;
      const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
      f32x4 sn = zero4, sp = zero4, sq = zero4;
      for (int item = tid; item < ncol * dsplit; item += PMG_THREADS) {
        const int bq = item & 3;
        const int cw = item >> 2;
        const int w = cw % W;
        const int part = cw / W;
        const int d0 = part * dch;
        const int d1 = (d0 + dch < nrows) ? d0 + dch : nrows;
        if (d0 >= d1) continue;
        const f32x4 apush = ld4(ct + 0 * 16 + 4 * bq), apop = ld4(ct + 1 * 16 + 4 * bq),
                    anoop = ld4(ct + 2 * 16 + 4 * bq);
        const f32x4 pv = ld4(a.PV + ((size_t)p * G + g) * 1024 + w * 16 + 4 * bq);
        f32x4 dtop4;
#pragma unroll
        for (int j = 0; j < 4; ++j) dtop4[j] = dtl[(4 * bq + j) * PMG_XTP + w];
        const size_t cofs = (size_t)w * 16 + 4 * bq;
#define LD_S(dd) (((dd) < 0 || (dd) > s_hi || zero_S) ? zero4 : pmg_ld4h<4>(Sprev + (size_t)(dd) * W * 16 + cofs))
#define LD_G(dd)                                                                              \
  (((dd) < 0 || (dd) >= D) ? zero4                                                            \
                           : (((zero_dS || (dd) > g_hi) ? zero4 : ld4(dSin + (size_t)(dd) * W * 16 + cofs)) + \
                              (((dd) == 0) ? dtop4 : zero4)))
        f32x4 gm = LD_G(d0 - 1), gc = LD_G(d0);
        f32x4 sm = (d0 == 0) ? pv : LD_S(d0 - 1);
        f32x4 sc = LD_S(d0);
        // Bursts of PMG_SBLKB rows: every load of a burst is issued (unconditionally, at a clamped
        // row; rows outside the live range are replaced by zero afterwards) before the first use,
        // so a thread has 2*PMG_SBLKB loads in flight instead of 2.
        const size_t rstr = (size_t)W * 16;
        for (int d = d0; d < d1; d += PMG_SBLKB) {
          f32x4 gv[PMG_SBLKB], sv[PMG_SBLKB];
#pragma unroll
          for (int k = 0; k < PMG_SBLKB; ++k) {
            const int dd = d + 1 + k;
            const int dc = dd < D ? dd : D - 1;
            gv[k] = ld4(dSin + (size_t)dc * rstr + cofs);
            sv[k] = pmg_ld4h<4>(Sprev + (size_t)dc * rstr + cofs);
          }
#pragma unroll
          for (int k = 0; k < PMG_SBLKB; ++k) {
            const int dd = d + 1 + k;
            const f32x4 gn = (zero_dS || dd > g_hi) ? zero4 : gv[k];
            const f32x4 sx = (zero_S || dd > s_hi) ? zero4 : sv[k];
            if (d + k < d1) {
              sn += gc * sc;
              sp += gc * sm;
              sq += gc * sx;
              const f32x4 dprev = anoop * gc + apush * gn + apop * gm;
              st4(dSout + (size_t)(d + k) * rstr + cofs, dprev);
              if (d + k == 0) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                  const float dv = apush[j] * gc[j];
                  dxl[(4 * bq + j) * XP + 3 + w] = dv * (1.0f - pv[j] * pv[j]);
                }
              }
              gm = gc; gc = gn; sm = sc; sc = sx;
            }
          }
        }
#undef LD_S
#undef LD_G
      }
      // reduce the control-gradient partials over the lanes that share bq = lane&3
#pragma unroll
      for (int o = 4; o < 64; o <<= 1) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          sn[j] += __shfl_xor(sn[j], o);
          sp[j] += __shfl_xor(sp[j], o);
          sq[j] += __shfl_xor(sq[j], o);
        }
      }
      if (lane < 4) {
        float* rp = red + (wave * 4 + lane) * 12;
#pragma unroll
        for (int j = 0; j < 4; ++j) { rp[j] = sp[j]; rp[4 + j] = sq[j]; rp[8 + j] = sn[j]; }
      }
    }
    PMG_TICK(4)
    // ------------------------------------------------------------ dlogits_{p-1}, loss
    if (!HELP) {
      const float sc = a.scale;
      const int bb = tid & 15;
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int v = (tid >> 4) + 32 * k;
        if (v < V) dxl[bb * XP + 3 + W + v] = (p >= 1) ? sc * rowR * (dl_pr[k] - ((v == dl_y) ? 1.0f : 0.0f)) : 0.0f;
      }
      for (int idx = tid + 2 * PMG_THREADS; idx < 16 * V; idx += PMG_THREADS) {  // V > 64
        const int v = idx >> 4;
        float val = 0.0f;
        if (p >= 1) {
          const float pr = pmg_ld1h<4>(a.PR + (((size_t)(p - 1) * G + g) * a.VP + v) * 16 + bb);
          val = sc * rowR * (pr - ((v == dl_y) ? 1.0f : 0.0f));
        }
        dxl[bb * XP + 3 + W + v] = val;
      }
      if (p >= 1 && tid < 16) lossacc += myR * a.LP[((size_t)(p - 1) * G + g) * 16 + tid];
    }
    __syncthreads();
    if (!HELP && !virt && tid < 16) {
      const int bq = tid >> 2, j = tid & 3;
      float dpu = 0.f, dpo = 0.f, dno = 0.f;
      for (int wv = 0; wv < PMG_WAVES; ++wv) {
        const float* rp = red + (wv * 4 + bq) * 12;
        dpu += rp[j]; dpo += rp[4 + j]; dno += rp[8 + j];
      }
      const float a0 = ct_a[0], a1 = ct_a[1], a2 = ct_a[2];
      const float s = a0 * dpu + a1 * dpo + a2 * dno;
      dxl[tid * XP + 0] = a0 * (dpu - s);
      dxl[tid * XP + 1] = a1 * (dpo - s);
      dxl[tid * XP + 2] = a2 * (dno - s);
    }
    if (HELP && !MULTI && !virt) {
      // groups 1 .. KX0-1 hold dpush rows only (ready since the last barrier); group 0 also holds
      // the three control gradients the helpers are forming right now
      const unsigned so = a.oWXt + (unsigned)(wave * UPW) * (unsigned)NXT * 1024u;
      if (!(dbg & 0x200)) gemm_lin<UPW>(car, wrs, voff, so, (unsigned)NXT * 1024u, dxl + b * XP + 4 * q, 1, KX0);
    }
    __syncthreads();
    PMG_TICK(5)
    // ------------------------------------------------------------ DX slab + extras product
    {
      if (!HELP) {
        float* dxg = a.DX + ((size_t)p * G + g) * NXT * 256;
        for (int idx = tid; idx < NXT * 256; idx += PMG_THREADS) {
          const int bb = idx & 15, nrow = idx >> 4;
          pmg_st1h<8>(dxg + idx, dxl[bb * XP + nrow]);
        }
      }
      const unsigned so = a.oWXt + (unsigned)(wave * UPW) * (unsigned)NXT * 1024u;
      if (MULTI) {
        f32x4 acc[UPW];
#pragma unroll
        for (int u = 0; u < UPW; ++u) acc[u] = ld4(dhc + ((size_t)((L - 1) * UT + wave * UPW + u)) * 256 + lane * 4);
        if (!(dbg & 0x200)) gemm_lin<UPW>(acc, wrs, voff, so, (unsigned)NXT * 1024u, dxl + b * XP + 4 * q, 0, NXT);
#pragma unroll
        for (int u = 0; u < UPW; ++u) st4(dhc + ((size_t)((L - 1) * UT + wave * UPW + u)) * 256 + lane * 4, acc[u]);
      } else {
        const int kend = (HELP && !virt) ? 1 : NXT;   // the rest was multiplied while the helpers finished the stack
        if (!(dbg & 0x200)) gemm_lin<UPW>(car, wrs, voff, so, (unsigned)NXT * 1024u, dxl + b * XP + 4 * q, 0, kend);
      }
    }
    PMG_TICK(6)
    __syncthreads();  // dxl / dtl / red reusable
    PMG_TICK(7)
  }
  }
#ifdef PMG_PROFILE
  if (prof && g == 0 && (tid & 63) == 0)
    printf("bwd wave %d cycles: cell %lld sync %lld gemm %lld sync2 %lld stack %lld dlogits+ctrl %lld extras %lld sync3 %lld\n", wave,
           tph[0], tph[1], tph[2], tph[3], tph[4], tph[5], tph[6], tph[7]);
#endif

  // ---------------------------------------------------------------- outputs
  if (!helper)
    for (int l = 0; l < L; ++l)
#pragma unroll
      for (int u = 0; u < UPW; ++u) {
        const f32x4 v = MULTI ? ld4(dhc + ((size_t)(l * UT + wave * UPW + u)) * 256 + lane * 4) : car[u];
        const int row = row0 + b;
        if (row < B) st4(a.grad_h0 + ((size_t)l * B + row) * H + 16 * (wave * UPW + u) + 4 * q, v);
      }
  if (lt >= 0 && lt < 16) lsl[lt] = lossacc;
  __syncthreads();
  if (tid == 0) {
    float s = 0.0f;
    for (int i = 0; i < 16; ++i) s += lsl[i];
    a.LOSSP[g] = -a.scale * s;
  }
}

}  // namespace

// Chunk buffers per round: the largest divisor of H/128 that is at most 4 and whose buffers fit the
// CU's 160 KiB next to the extras tile; 0 if not even one fits.
static size_t bwd_lds_bytes_ncb(const BwdArgs* a, int ncb) {
  return ((size_t)ncb * 16 * PMG_CP + (size_t)16 * a->XP + 16 * PMG_XTP + PMG_WAVES * 48 + 16) * sizeof(float);
}
int pmg_bwd_chunk_buffers(const BwdArgs* a) {
  const int upw = a->H / 128;
  for (int ncb = upw < 4 ? upw : 4; ncb >= 1; --ncb)
    if (upw % ncb == 0 && bwd_lds_bytes_ncb(a, ncb) <= (size_t)160 * 1024) return ncb;
  return 0;
}

size_t pmg_bwd_lds_bytes(const BwdArgs* a) { return bwd_lds_bytes_ncb(a, pmg_bwd_chunk_buffers(a)); }

template <int UPW, int NCB, bool MULTI, bool HELP>
static hipError_t launch_bwd_t(const BwdArgs* a, hipStream_t st) {
  const size_t lds = bwd_lds_bytes_ncb(a, NCB);
  auto kern = pmg_bwd_kernel<UPW, NCB, MULTI, HELP>;
  static std::atomic<unsigned long long> lds_done{0};
  hipError_t e = pmg_full_lds_once(reinterpret_cast<const void*>(kern), &lds_done);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(kern, dim3(a->G), dim3(HELP ? PMG_THREADS + PMG_HELP_THREADS : PMG_THREADS), lds, st, *a);
  return hipGetLastError();
}

template <int UPW, int NCB, bool HELP>
static hipError_t launch_bwd_m(const BwdArgs* a, hipStream_t st) {
  return a->L > 1 ? launch_bwd_t<UPW, NCB, true, HELP>(a, st) : launch_bwd_t<UPW, NCB, false, HELP>(a, st);
}

// helper waves up to hidden 512 (168 VGPRs per thread); above, the accumulators need the 8-wave form
template <int UPW>
static hipError_t launch_bwd_u(const BwdArgs* a, hipStream_t st) {
  constexpr bool HELP = (PMG_BWD_HELP != 0) && UPW <= 4;
  switch (pmg_bwd_chunk_buffers(a)) {
    case 4: if constexpr (UPW % 4 == 0) return launch_bwd_m<UPW, 4, HELP>(a, st); break;
    case 3: if constexpr (UPW % 3 == 0) return launch_bwd_m<UPW, 3, HELP>(a, st); break;
    case 2: if constexpr (UPW % 2 == 0) return launch_bwd_m<UPW, 2, HELP>(a, st); break;
    case 1: return launch_bwd_m<UPW, 1, HELP>(a, st);
  }
  return hipErrorInvalidValue;
}

hipError_t pmg_launch_bwd(const BwdArgs* a, hipStream_t st) {
  // small batches: the hidden-partitioned form (a cluster of H/16 CUs per 16 sequences), same results
  if (pmg_bwdc_applicable(a)) return pmg_launch_bwdc(a, st);
  switch (a->H) {
    case 128: return launch_bwd_u<1>(a, st);
    case 256: return launch_bwd_u<2>(a, st);
    case 384: return launch_bwd_u<3>(a, st);
    case 512: return launch_bwd_u<4>(a, st);
    case 640: return launch_bwd_u<5>(a, st);
    case 768: return launch_bwd_u<6>(a, st);
    case 896: return launch_bwd_u<7>(a, st);
    case 1024: return launch_bwd_u<8>(a, st);
  }
  return hipErrorInvalidValue;
}
