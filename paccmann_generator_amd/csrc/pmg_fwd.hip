// Persistent fused StackGRU forward: sampling decode, teacher-forced log-probs and the
// single-step form, one kernel.
//
// One workgroup owns 16 sequences for every time step.  Hidden state lives in LDS, the
// differentiable stack in a private row-slab of the workspace (L2/Infinity-Cache
// resident; in teacher-forced mode the slab of every step is kept because the backward
// needs it), weights stream from L2 in the packed layout of pmg_pack.hip.
//
// Arithmetic per step (reference call sites: reinforce_omics.py:323-329,
// reinforce.py:162-170; decoder restated in oracle/stack_gru.py):
//   extras = [W_ctrl; W_push; W_out] h_{i-1}^{top} + b      (one GEMM, K = H)
//      -> logits of step i-1 (sample / log-softmax), stack controls and push value of step i
//   stack update (push/pop/no-op blend), top row -> GRU input
//   per layer: gates = W_hh h_old (+ W_ih x), layer 0 adds the per-token table GTAB[tok]
//   h_new = (h_old - n) * z + n
#include "pmg_kernels.h"

#ifndef PMG_FWD_ILV
#define PMG_FWD_ILV 1
#endif

namespace {

__device__ __forceinline__ f32x4 ld4(const float* p) {
  const float4 v = *reinterpret_cast<const float4*>(p);
  f32x4 r; r.x = v.x; r.y = v.y; r.z = v.z; r.w = v.w; return r;
}
__device__ __forceinline__ void st4(float* p, const f32x4 v) {
  *reinterpret_cast<float4*>(p) = make_float4(v.x, v.y, v.z, v.w);
}
// 16-byte accesses through a buffer resource: one lane-constant byte offset in a VGPR, everything
// else wave-uniform (no 64-bit address arithmetic in vector registers)
__device__ __forceinline__ f32x4 bl4(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
  const float4 v = pmg_bload(r, voff, soff);
  f32x4 o; o.x = v.x; o.y = v.y; o.z = v.z; o.w = v.w; return o;
}
// history stores of the teacher-forced pass (gates, hidden): written once, read once by the backward
// much later; PMG_FWD_AUX_HIST = cache-policy bits of those stores (2 = nt).  Tuning knob for the shapes whose
// packed weights fit the 4 MB L2s comfortably (measured neutral at cfg2); at hidden 1024 / two layers (40 MB of
// packed weights living in the Infinity Cache) the streams go non-temporal: teacher-forced pass 161 -> 150 ms at
// cfg4's shape on one box, the weights are no longer evicted by the history (profiles/r03_cfg4_nt.md).
#ifndef PMG_FWD_AUX_HIST
#define PMG_FWD_AUX_HIST 0
#endif
template <int AUX>
__device__ __forceinline__ void bs4(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff, const f32x4 v) {
  pmg_u32x4 u;
  u.x = __float_as_uint(v.x); u.y = __float_as_uint(v.y); u.z = __float_as_uint(v.z); u.w = __float_as_uint(v.w);
  __builtin_amdgcn_raw_buffer_store_b128(u, r, voff, soff, AUX);
}
// stack slab streams: cache-policy bits of the buffer instruction (bit 1 = nt) as a tuning knob
#ifndef PMG_STACK_AUX_LD
#define PMG_STACK_AUX_LD 0
#endif
#ifndef PMG_STACK_AUX_ST
#define PMG_STACK_AUX_ST 0
#endif
__device__ __forceinline__ f32x4 bl4s(__amdgpu_buffer_rsrc_t r, unsigned voff) {
  const pmg_u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, voff, 0, PMG_STACK_AUX_LD);
  f32x4 o; o.x = __uint_as_float(v.x); o.y = __uint_as_float(v.y); o.z = __uint_as_float(v.z); o.w = __uint_as_float(v.w);
  return o;
}
__device__ __forceinline__ void bs4s(__amdgpu_buffer_rsrc_t r, unsigned voff, const f32x4 v) {
  pmg_u32x4 u;
  u.x = __float_as_uint(v.x); u.y = __float_as_uint(v.y); u.z = __float_as_uint(v.z); u.w = __float_as_uint(v.w);
  __builtin_amdgcn_raw_buffer_store_b128(u, r, voff, 0, PMG_STACK_AUX_ST);
}
template <int AUX>
__device__ __forceinline__ void bs1(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff, const float v) {
  __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), r, voff, soff, AUX);
}

// acc{r,z,t}[u0..u0+TPP) += W[tiles 3*u0 ..] * X   over KG groups of 16 columns.
// Weights come through the buffer resource `rs`: voff = lane*16, soff = byte offset of
// (first tile, group 0); xrow = this lane's LDS row + 4*(lane>>4).
// Two register sets (wa, wb), loop unrolled by two: the loads of group kg+1 are in flight
// while the MFMAs of group kg run (the sched_barriers keep hipcc from re-fusing load and
// use, which it otherwise does).  KG is even.
// Side work interleaved with a GEMM K loop.  A side job is called once per double K group with a
// STATIC slot number (the loop is unrolled by four), so the value a call loads lands in the
// register it is consumed from four calls later: no register copies of in-flight data, hence no
// early waits.  vmcnt counts loads AND stores in issue order, so a wait for weights that were
// requested after the side job's store also waits for that store's acknowledgement: the side job
// sits right after the loads of the second half group, which gives its store two half groups of
// MFMAs before the next wait that covers it.
#ifndef PMG_STACK_THREADS
#define PMG_STACK_THREADS 512
#endif
struct NoSide {
  static constexpr bool kActive = false;
  template <int U> __device__ __forceinline__ void step() const {}
};

// Stack update of one thread: column (w, 4 sequences), rows d0 .. d1-1, one row per call.
//   new[d] = no_op*old[d] + push*old[d-1] + pop*old[d+1]
// Rows d < KL of the stack live in LDS and are updated IN PLACE (a thread walks upward with
// old[d-1], old[d] in registers and old[d+1] requested ahead, so it never reads what it has
// written); deeper rows live in the workgroup's global ping-pong slab.  A depth part is either all
// LDS or all global (wave-uniform `lds`), and the only rows a thread needs from outside its own
// range -- old[d0-1] and old[d1] -- are read before barrier B1, i.e. before anybody writes
// (s_end = old[d1]).  q[] is a ring of own rows requested ahead (slot of old[d0+1+k] is (k-1)&3).
// a wave-uniform pointer the register allocator may have parked in vector registers: back to SGPRs
// (two v_readfirstlane instead of a waterfall loop around every buffer access built from it)
template <class T> __device__ __forceinline__ T* pmg_uniform_ptr(T* p) {
  const unsigned long long v = reinterpret_cast<unsigned long long>(p);
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
  return reinterpret_cast<T*>(((unsigned long long)hi << 32) | lo);
}

struct StackSide {
  static constexpr bool kActive = true;
  // fp32 MFMA and fp32 VALU share the SIMD's ALUs, so every vector instruction of a side-job call is
  // time taken from the product it rides on.  Row indices are therefore wave-uniform (all address
  // arithmetic is scalar: one descriptor per row, built in SGPRs), rows that hold no data read as
  // zero through the descriptor's bounds (global slab) or because the LDS rows start zeroed, and
  // threads past the last stack column carry a lane offset outside every row (loads give zero,
  // stores are dropped) instead of a predicate.
  const float* Sold;   // old / new global stack slab of this row group (wave-uniform)
  float* Snew;
  unsigned vcol;       // byte offset of this thread's column inside a stack row; 0x40000000 for no column
  float* sl;           // LDS: this thread's column of stack row 0
  f32x4 anoop, apush, apop, s_prev, s_cur, s_end, q[4];
  int d0, d1, rows, old_hi, sj, dch;  // dch: calls after which every thread of the workgroup is done
  unsigned rstr;       // floats per stack row: W*16
  bool on, lds, hist;  // on: this thread has a column (per lane); lds: own rows in LDS (wave-uniform); hist: new rows are also kept in the global slab
  __device__ __forceinline__ f32x4 ldg(const int d) const {   // row d of the old global slab; zero past old_hi
    const __amdgpu_buffer_rsrc_t r = pmg_rsrc(pmg_uniform_ptr(Sold) + (size_t)d * rstr, (d <= old_hi) ? rstr * 4u : 0u);
    return bl4s(r, vcol);
  }
  __device__ __forceinline__ void stg(const int d, const f32x4 v) const {
    const __amdgpu_buffer_rsrc_t r = pmg_rsrc(pmg_uniform_ptr(Snew) + (size_t)d * rstr, rstr * 4u);
    bs4s(r, vcol, v);
  }
  __device__ __forceinline__ f32x4 ldrow(const int d) const {
    if (lds) return ld4(sl + (size_t)d * rstr);
    return ldg(d);
  }
  __device__ __forceinline__ void strow(const int d, const f32x4 v) const {
    if (lds) {
      if (on) st4(sl + (size_t)d * rstr, v);
      if (hist) stg(d, v);
    } else {
      stg(d, v);
    }
  }
  template <int U> __device__ __forceinline__ void step() {
    if (sj >= dch) return;  // uniform: nothing left for anybody
    const int d = d0 + sj;  // uniform
    f32x4 nxt = q[U];
    if (sj + 4 < dch) {  // uniform: somebody will consume this slot again
      const int dn = d + 5;
      q[U] = ldrow(dn < d1 ? dn : d1 - 1);  // own rows only
    }
    if (d + 1 == d1) nxt = s_end;           // uniform: the row above the part comes from the other part
    const f32x4 nw = anoop * s_cur + apush * s_prev + apop * nxt;
    if (sj < rows) strow(d, nw);            // uniform
    s_prev = s_cur;
    s_cur = nxt;
    ++sj;
  }
};

// acc{r,z,t}[u0..u0+TPP) += W[tiles 3*u0 ..] * X   over KG groups of 16 columns.
// Weights come through the buffer resource `rs`: voff = lane*16, soff = byte offset of
// (first tile, group 0); xrow = this lane's LDS row + 4*(lane>>4).
// Two register sets (wa, wb): the loads of group kg+1 are in flight while the MFMAs of group kg
// run (the sched_barriers keep hipcc from re-fusing load and use, which it otherwise does).
// KG is even; with an active side job KG is a multiple of 8 and the loop is unrolled by four.
// `wa` is owned by the caller: with `loaded` it already holds group 0 of this pass, and on return it
// holds group 0 of the weights at (nso, nts) -- the next pass or the next product -- so that no pass
// starts by waiting for an L2 round trip with every wave of the workgroup stalled at once.
template <int UPW, int TPP, class Side>
__device__ __forceinline__ void gemm3_pass(f32x4 (&ar)[UPW], f32x4 (&az)[UPW], f32x4 (&at)[UPW], const int u0,
                                           const __amdgpu_buffer_rsrc_t rs, const unsigned voff,
                                           const unsigned soff, const int KG, const float* xrow, Side& side,
                                           float4 (&wa)[3 * TPP], const bool loaded, const unsigned nso,
                                           const unsigned nts) {
  constexpr int NT = 3 * TPP;
  const unsigned ts = (unsigned)KG * 1024u;  // bytes per tile
  float4 wb[NT];
#define PMG_LOAD0(dst, kgi) \
  _Pragma("unroll") for (int t = 0; t < NT; ++t) dst[t] = pmg_bload(rs, voff, soff + t * ts + (unsigned)(kgi) * 1024u);
#ifdef PMG_ABL_NOLOAD  /* timing experiment: weights loaded once per pass */
#define PMG_LOAD(dst, kgi) asm volatile("" : "+v"(dst[0].x));
#else
#define PMG_LOAD(dst, kgi) PMG_LOAD0(dst, kgi)
#endif
#define PMG_LOADX(dst, so, tsx) \
  _Pragma("unroll") for (int t = 0; t < NT; ++t) dst[t] = pmg_bload(rs, voff, (so) + t * (tsx));
#ifdef PMG_ABL_NOMFMA  /* timing experiment: one VALU fma in place of each MFMA */
#define PMG_STEP(wc, C)                                                                          \
  _Pragma("unroll") for (int uu = 0; uu < TPP; ++uu) {                                           \
    ar[u0 + uu].x += wc[3 * uu + 0].C * x.C;                                                     \
    az[u0 + uu].x += wc[3 * uu + 1].C * x.C;                                                     \
    at[u0 + uu].x += wc[3 * uu + 2].C * x.C;                                                     \
  }
#else
#define PMG_STEP(wc, C)                                                                          \
  _Pragma("unroll") for (int uu = 0; uu < TPP; ++uu) {                                           \
    ar[u0 + uu] = __builtin_amdgcn_mfma_f32_16x16x4f32(wc[3 * uu + 0].C, x.C, ar[u0 + uu], 0, 0, 0); \
    az[u0 + uu] = __builtin_amdgcn_mfma_f32_16x16x4f32(wc[3 * uu + 1].C, x.C, az[u0 + uu], 0, 0, 0); \
    at[u0 + uu] = __builtin_amdgcn_mfma_f32_16x16x4f32(wc[3 * uu + 2].C, x.C, at[u0 + uu], 0, 0, 0); \
  }
#endif
#if PMG_FWD_ILV
  // Interleaved form: the six loads of the next set ride one by one in the shadow of the first six
  // MFMAs of the current set (a block of loads holds the SIMD's issue port while the MFMA pipe
  // drains; tests/microbench/gemm_loop.hip).
#define PMG_MF1(wc, C, t)                                                                        \
  {                                                                                              \
    f32x4& acc_ = ((t) % 3 == 0) ? ar[u0 + (t) / 3] : ((t) % 3 == 1) ? az[u0 + (t) / 3] : at[u0 + (t) / 3]; \
    acc_ = __builtin_amdgcn_mfma_f32_16x16x4f32(wc[t].C, x.C, acc_, 0, 0, 0);                    \
  }
#define PMG_DGROUP(U, kgq)                                                                       \
  {                                                                                              \
    const int kg_ = (kgq);                                                                       \
    const float4 xb = *reinterpret_cast<const float4*>(xrow + 16 * kg_ + 16);                    \
    __builtin_amdgcn_sched_barrier(0);                                                           \
    side.template step<U>();                                                                     \
    __builtin_amdgcn_sched_barrier(0);                                                           \
    {                                                                                            \
      const float4 x = xa;                                                                       \
      _Pragma("unroll") for (int t = 0; t < NT; ++t) {                                           \
        PMG_MF1(wa, x, t)                                                                        \
        wb[t] = pmg_bload(rs, voff, soff + t * ts + (unsigned)(kg_ + 1) * 1024u);                \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                       \
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);                                       \
      }                                                                                          \
      PMG_STEP(wa, y) PMG_STEP(wa, z) PMG_STEP(wa, w)                                            \
    }                                                                                            \
    __builtin_amdgcn_sched_barrier(0);                                                           \
    {                                                                                            \
      const bool more_ = kg_ + 2 < KG;                                                           \
      const unsigned so_ = more_ ? soff + (unsigned)(kg_ + 2) * 1024u : nso;                     \
      const unsigned ts_ = more_ ? ts : nts;                                                     \
      float4 wn_[NT];                                                                            \
      const float4 x = xb;                                                                       \
      xa = *reinterpret_cast<const float4*>(xrow + 16 * (more_ ? kg_ + 2 : kg_));                \
      _Pragma("unroll") for (int t = 0; t < NT; ++t) {                                           \
        PMG_MF1(wb, x, t)                                                                        \
        wn_[t] = pmg_bload(rs, voff, so_ + t * ts_);                                             \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                       \
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);                                       \
      }                                                                                          \
      PMG_STEP(wb, y) PMG_STEP(wb, z) PMG_STEP(wb, w)                                            \
      __builtin_amdgcn_sched_barrier(0);                                                         \
      _Pragma("unroll") for (int t = 0; t < NT; ++t) wa[t] = wn_[t];                             \
    }                                                                                            \
    __builtin_amdgcn_sched_barrier(0);                                                           \
  }
#else
#define PMG_DGROUP(U, kgq)                                                                       \
  {                                                                                              \
    const int kg_ = (kgq);                                                                       \
    PMG_LOAD(wb, kg_ + 1)                                                                        \
    const float4 xb = *reinterpret_cast<const float4*>(xrow + 16 * kg_ + 16);                    \
    __builtin_amdgcn_sched_barrier(0);                                                           \
    side.template step<U>();                                                                     \
    __builtin_amdgcn_sched_barrier(0);                                                           \
    {                                                                                            \
      const float4 x = xa;                                                                       \
      PMG_STEP(wa, x) PMG_STEP(wa, y) PMG_STEP(wa, z) PMG_STEP(wa, w)                            \
    }                                                                                            \
    __builtin_amdgcn_sched_barrier(0);                                                           \
    {                                                                                            \
      const bool more_ = kg_ + 2 < KG;                                                           \
      const unsigned so_ = more_ ? soff + (unsigned)(kg_ + 2) * 1024u : nso;                     \
      const unsigned ts_ = more_ ? ts : nts;                                                     \
      PMG_LOADX(wa, so_, ts_)                                                                    \
      xa = *reinterpret_cast<const float4*>(xrow + 16 * (more_ ? kg_ + 2 : kg_));                \
    }                                                                                            \
    __builtin_amdgcn_sched_barrier(0);                                                           \
    {                                                                                            \
      const float4 x = xb;                                                                       \
      PMG_STEP(wb, x) PMG_STEP(wb, y) PMG_STEP(wb, z) PMG_STEP(wb, w)                            \
    }                                                                                            \
    __builtin_amdgcn_sched_barrier(0);                                                           \
  }
#endif
  if (!loaded) PMG_LOAD0(wa, 0)
  float4 xa = *reinterpret_cast<const float4*>(xrow);  // B operand, one half group ahead of its MFMAs
#ifdef PMG_ABL_NOLOAD
  PMG_LOAD0(wb, 1)
#endif
  if constexpr (Side::kActive) {
#pragma unroll 1
    for (int kg = 0; kg < KG; kg += 8) {
      PMG_DGROUP(0, kg) PMG_DGROUP(1, kg + 2) PMG_DGROUP(2, kg + 4) PMG_DGROUP(3, kg + 6)
    }
  } else {
#pragma unroll 1
    for (int kg = 0; kg < KG; kg += 2) PMG_DGROUP(0, kg)
  }
#undef PMG_DGROUP
#if PMG_FWD_ILV
#undef PMG_MF1
#endif
#undef PMG_STEP
#undef PMG_LOAD
#undef PMG_LOAD0
#undef PMG_LOADX
}

// byte offset of (first tile of this wave, group 0) of a gate product with KG groups
template <int UPW>
__device__ __forceinline__ unsigned gemm3_first(const unsigned obase, const int wave, const int KG) {
  return obase + (unsigned)(3 * wave * UPW) * (unsigned)KG * 1024u;
}
// prefetch of group 0 of a product's first pass into the caller's register set
template <int NT>
__device__ __forceinline__ void gemm3_preload(float4 (&wa)[NT], const __amdgpu_buffer_rsrc_t rs, const unsigned voff,
                                              const unsigned so, const unsigned ts) {
#pragma unroll
  for (int t = 0; t < NT; ++t) wa[t] = pmg_bload(rs, voff, so + t * ts);
}

// nso/nts: where the product that runs next starts (gemm3_first of it, its tile stride); the last
// pass leaves that group in `wa`.  nso = this product's own first group when there is nothing to
// prefetch (a harmless re-read).
template <int UPW, int TPP, class Side>
__device__ __forceinline__ void gemm3(f32x4 (&ar)[UPW], f32x4 (&az)[UPW], f32x4 (&at)[UPW],
                                      const __amdgpu_buffer_rsrc_t rs, const unsigned voff, const unsigned obase,
                                      const int KG, const float* xrow, const int wave, Side& side,
                                      float4 (&wa)[3 * TPP], const bool loaded, const unsigned nso,
                                      const unsigned nts) {
  // this wave's tiles: 3*(wave*UPW + u), u in [0,UPW)
  const unsigned ts = (unsigned)KG * 1024u;
#pragma unroll
  for (int ps = 0; ps < UPW / TPP; ++ps) {
    const int u0 = ps * TPP;
    const unsigned soff = obase + (unsigned)(3 * (wave * UPW + u0)) * ts;
    const bool last = ps == UPW / TPP - 1;
    gemm3_pass<UPW, TPP>(ar, az, at, u0, rs, voff, soff, KG, xrow, side, wa, loaded || ps > 0,
                         last ? nso : soff + (unsigned)(3 * TPP) * ts, last ? nts : ts);
  }
}

// One 16-row output tile, K = KG*16 (KG even).  Two register sets of four groups keep four
// weight loads in flight; two accumulators break the dependent-MFMA chain.
__device__ __forceinline__ f32x4 gemm1(f32x4 acc, const __amdgpu_buffer_rsrc_t rs, const unsigned voff,
                                       const unsigned soff, const int KG, const float* xrow) {
  f32x4 acc2 = {0.f, 0.f, 0.f, 0.f};
  float4 wa[4], wb[4];
#define G1_LOAD(dst, k0)                                                        \
  _Pragma("unroll") for (int j = 0; j < 4; ++j) {                               \
    const int kk = ((k0) + j < KG) ? (k0) + j : KG - 1;                         \
    dst[j] = pmg_bload(rs, voff, soff + (unsigned)kk * 1024u);                  \
  }
#define G1_COMP(src, k0)                                                        \
  _Pragma("unroll") for (int j = 0; j < 4; j += 2) {                            \
    if ((k0) + j < KG) {                                                        \
      const float4 x0 = *reinterpret_cast<const float4*>(xrow + 16 * ((k0) + j));      \
      const float4 x1 = *reinterpret_cast<const float4*>(xrow + 16 * ((k0) + j) + 16); \
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(src[j].x, x0.x, acc, 0, 0, 0);        \
      acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(src[j + 1].x, x1.x, acc2, 0, 0, 0);  \
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(src[j].y, x0.y, acc, 0, 0, 0);        \
      acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(src[j + 1].y, x1.y, acc2, 0, 0, 0);  \
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(src[j].z, x0.z, acc, 0, 0, 0);        \
      acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(src[j + 1].z, x1.z, acc2, 0, 0, 0);  \
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(src[j].w, x0.w, acc, 0, 0, 0);        \
      acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(src[j + 1].w, x1.w, acc2, 0, 0, 0);  \
    }                                                                           \
  }
  G1_LOAD(wa, 0)
#pragma unroll 1
  for (int kg = 0; kg < KG; kg += 8) {
    G1_LOAD(wb, kg + 4)
    __builtin_amdgcn_sched_barrier(0);
    G1_COMP(wa, kg)
    __builtin_amdgcn_sched_barrier(0);
    G1_LOAD(wa, kg + 8)
    __builtin_amdgcn_sched_barrier(0);
    G1_COMP(wb, kg + 4)
    __builtin_amdgcn_sched_barrier(0);
  }
#undef G1_LOAD
#undef G1_COMP
  return acc + acc2;
}

// Gate non-linearities on the hardware exp/rcp units (v_exp_f32, v_rcp_f32: ~1 ulp each).
// Absolute error ~1e-7, the same order as the GEMM summation-order noise.
__device__ __forceinline__ float fast_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float fast_tanh(float x) {
  const float e = __expf(-2.0f * fabsf(x));
  const float t = (1.0f - e) * __builtin_amdgcn_rcpf(1.0f + e);
  return copysignf(t, x);
}

// MODE (decode / teacher / step) and MULTI (more than one layer) are compile-time: each variant
// carries only the registers and branches it needs.
// WV waves per workgroup (8 or 16): with 16 the kernel has to fit in 128 VGPRs, and four waves per
// SIMD are there to fill each other's stalls.
template <int WV, int UPW, int TPP, int MODE, bool MULTI>
__global__ __launch_bounds__(WV * 64, WV / 4) void pmg_fwd_kernel(const FwdArgs a) {
  constexpr int NTHR = WV * 64;
  constexpr int HAUX = (MULTI || WV * UPW * 16 >= 1024) ? (PMG_FWD_AUX_HIST | 2) : PMG_FWD_AUX_HIST;   // history stores
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int H = a.H, HP = a.HP, XP = a.XP, W = a.W, D = a.D, V = a.V, L = a.L, B = a.B, G = a.G, N = a.N;
  const int KGH = a.KGH, UT = H / 16;
  float* h_lds = reinterpret_cast<float*>(smem);       // [L][16][HP]
  float* xtop = h_lds + (size_t)L * 16 * HP;             // [16][PMG_XTP]
  float* xex = xtop + 16 * PMG_XTP;                      // [16][XP]
  int* misc = reinterpret_cast<int*>(xex + 16 * XP);     // [0..15] token, [16] all-done flag
  float* slds = reinterpret_cast<float*>(misc + 32);     // [KL][W*16] top rows of the stack
  // decode only: the latent rows [16][Z+8] of the h0 product live where xtop .. slds will (nothing behind the
  // hidden state is in use before the time loop; pmg_fwd_lds_bytes covers the larger of the two)
  float* lat = xtop;

  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63;
  const int b = lane & 15, q = lane >> 4;
  const int g = blockIdx.x;
  const int row0 = g * PMG_ROWS;
  const __amdgpu_buffer_rsrc_t wrs = pmg_rsrc(a.wbase, a.wbytes);
  const unsigned voff = (unsigned)lane * 16u;
  constexpr bool teacher = (MODE == PMG_FWD_TEACHER);
  constexpr bool decode = (MODE == PMG_FWD_DECODE);
  constexpr bool stepm = (MODE == PMG_FWD_STEP);
  const bool keep = teacher && !(PMG_DBG(a.dbg) & 0x1000);  // keep per-step history for the backward
  const int dbg = PMG_DBG(a.dbg);

  // ---------------------------------------------------------------- init
  if (decode) {
    // h0 = latent_to_hidden(latent): same value for every layer (reinforce.py:160-170 path)
    const int ZP = a.Z + 8;
    for (int idx = tid; idx < 16 * a.Z; idx += NTHR) {
      const int bb = idx / a.Z, k = idx - bb * a.Z;
      lat[bb * ZP + k] = (row0 + bb < B) ? a.latent[(size_t)(row0 + bb) * a.Z + k] : 0.0f;
    }
    __syncthreads();
    for (int t = wave; t < UT; t += WV) {
      f32x4 acc = ld4(a.b_l2h + 16 * t + 4 * q);
      acc = gemm1(acc, wrs, voff, a.oL2H + (unsigned)t * (unsigned)(a.Z / 16) * 1024u, a.Z / 16, lat + b * ZP + 4 * q);
      for (int l = 0; l < L; ++l) st4(&h_lds[(size_t)(l * 16 + b) * HP + 16 * t + 4 * q], acc);
    }
    __syncthreads();                 // the latent rows are dead: their LDS becomes xtop, xex, misc and the stack rows
  } else {
    for (int idx = tid; idx < L * 16 * H; idx += NTHR) {
      const int l = idx / (16 * H);
      const int rem = idx - l * 16 * H;
      const int bb = rem / H, k = rem - bb * H;
      const float v = (row0 + bb < B) ? a.h0[((size_t)l * B + row0 + bb) * H + k] : 0.0f;
      h_lds[(size_t)(l * 16 + bb) * HP + k] = v;
    }
  }
  for (int idx = tid; idx < 16 * PMG_XTP; idx += NTHR) xtop[idx] = 0.0f;
  // LDS stack rows start at zero: the side job reads rows that hold no data yet without a mask
  for (int idx = tid; idx < a.KL * W * 16; idx += NTHR) slds[idx] = 0.0f;
  if (tid >= 16 && tid < 32) misc[tid] = 0;
  if (tid < 16) {
    int t0 = a.start_tok;
    // ids outside [0,V) are clamped (memory safety only: callers validate, ops.reinforce_loss does)
    if (!decode) t0 = (row0 + tid < B) ? min(max((int)a.tok_in[row0 + tid], 0), V - 1) : 0;
    misc[tid] = t0;
  }
  bool s_zero = true;  // stack entering step 0 is all zeros
  if (stepm && a.stack_in != nullptr) {
    s_zero = false;
    float* S0 = a.SB + (size_t)g * a.SW;
    for (int idx = tid; idx < D * W * 16; idx += NTHR) {
      const int bb = idx & 15;
      const int dw = idx >> 4;
      S0[idx] = (row0 + bb < B) ? a.stack_in[(size_t)(row0 + bb) * D * W + dw] : 0.0f;
    }
  }
  __syncthreads();
  if (keep) {
    for (int idx = tid; idx < L * 16 * H; idx += NTHR) {
      const int l = idx / (16 * H);
      const int rem = idx - l * 16 * H;
      const int k = rem >> 4, bb = rem & 15;
      a.HH[l][((size_t)g * H + k) * 16 + bb] = h_lds[(size_t)(l * 16 + bb) * HP + k];
    }
  }

  // logits are consumed by every wave: rows 2*wave and 2*wave+1, 32 lanes per row
  const int cs_bb = (WV == 8) ? 2 * wave + (lane >> 5) : wave;  // 16 waves: both halves of a wave do the same row
  // sampling state of row cs_bb (identical in the 32 lanes of the row)
  bool finished = false;
  int mylen = N;

  f32x4 ar[UPW], az[UPW], ai[UPW], ah[UPW];

  // stack update work split: thread -> (column w, row quad bq, depth part).  Two depth parts, one
  // per half of the workgroup (waves 0-3 / 4-7), so that where a part's rows live (LDS or global) is
  // wave-uniform; the depth range of a part is recomputed every step from the number of live rows.
  const int KL = a.KL;               // stack rows kept in LDS (0: none)
  const int st_part = wave >> 2;     // scalar
  const int st_bq = tid & 3;
  const int st_cw = (tid & 255) >> 2;
  const bool st_col = st_cw < W;
  const int st_w = st_col ? st_cw : 0;
  const size_t st_cofs = (size_t)st_w * 16 + 4 * st_bq;
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  NoSide no_side;
  float4 wpre[3 * TPP];        // weight group prefetched for the product that runs next (see gemm3)
  const bool wpre_ok = false;  // the layer-0 W_hh product loads its own first group
  const __amdgpu_buffer_rsrc_t bgr0 = pmg_rsrc(a.BG[0], (unsigned)H * 16u);
  const __amdgpu_buffer_rsrc_t gtr = pmg_rsrc(a.GTAB, (unsigned)V * 3u * (unsigned)H * 4u);
  const unsigned vq = 4u * q * 4u;                       // feature 4q of a 16-feature tile
  const unsigned vrow = (unsigned)(4 * q * 16 + b) * 4u;  // (feature 4q, row b) of a [feature][16] slab

#ifdef PMG_PROFILE
  // -DPMG_PROFILE build + PMG_DEBUG bit 0x8000: per-phase cycle counters of workgroup 0, printed at the end
  const bool prof = (dbg & 0x8000) != 0;
  long long tph[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  long long tlast = prof ? (long long)__builtin_amdgcn_s_memtime() : 0;
#define PMG_TICK(k) if (prof) { const long long tn_ = (long long)__builtin_amdgcn_s_memtime(); tph[k] += tn_ - tlast; tlast = tn_; }
#else
#define PMG_TICK(k)
#endif

  for (int i = 0;; ++i) {
    // ------------------------------------------------------------ stack: first rows requested early
    const int st_slot_old = teacher ? i : (i & 1);
    const int st_slot_new = teacher ? i + 1 : ((i + 1) & 1);
    const unsigned st_vcol = st_col ? (unsigned)st_cofs * 4u : 0x40000000u;  // no column: outside every row
    // Live rows.  Entering step i only rows 0..i-1 of a stack that started at zero can be non-zero,
    // and row d of the new stack can reach the stack top (the only way the stack is ever read) no
    // earlier than step i+d: rows beyond min(i, N-1-i) are neither computed nor stored, and rows
    // of the old slab beyond st_old_hi are read as zero (they were never written).  The step form
    // returns the whole stack, so it computes every row.
    int st_hi = D - 1, st_old_hi = D - 1;
    if (s_zero) st_old_hi = (i - 1 < st_old_hi) ? i - 1 : st_old_hi;
    if (!stepm && !(dbg & 0x4000)) {
      if (N - 1 - i < st_hi) st_hi = N - 1 - i;
      if (s_zero && i < st_hi) st_hi = i;
    }
    const int st_nrows = st_hi + 1;
    // Depth parts: [0, bnd) for waves 0-3, [bnd, nrows) for waves 4-7.  With more live rows than fit
    // in LDS the cut is the LDS boundary (part 0 all LDS, part 1 all global), otherwise the middle
    // (both parts in LDS); without LDS rows (KL = 0) both parts are global.
    const int st_bnd = (KL > 0 && st_nrows > KL) ? KL : ((st_nrows + 1) >> 1);
    const int st_d0 = st_part ? st_bnd : 0;
    const int st_d1 = st_part ? st_nrows : st_bnd;
    const int st_dch = (st_bnd > st_nrows - st_bnd) ? st_bnd : st_nrows - st_bnd;
    const int st_rows = st_d1 - st_d0;
    const bool st_own_lds = KL > 0 && (st_part == 0 || st_nrows <= KL);  // scalar
    const bool st_act = st_d0 < st_d1 && !(dbg & 0x100) && i < N;  // scalar: this depth part has rows
    const bool st_on = st_col && st_act;
    // target token of step i-1 (teacher): requested here, used after the extras GEMM
    int cs_y = 0;
    if (teacher && i >= 1 && row0 + cs_bb < B) cs_y = min(max((int)a.tok_in[(size_t)i * B + row0 + cs_bb], 0), V - 1);

    // old rows d0-1, d0, d0+1 (first row of this thread), d1 (first row of the part above) and the ring
    // d0+2 .. d0+5: requested here, before barrier B1, i.e. before anybody overwrites an LDS row; only
    // rows that hold data are requested (memory operations are not free here), the rest stay zero
    StackSide stk;
    stk.Sold = a.SB + ((size_t)st_slot_old * G + g) * a.SW; stk.vcol = st_vcol; stk.rstr = (unsigned)W * 16u;
    stk.sl = slds + st_cofs;
    stk.d0 = st_act ? st_d0 : 0; stk.d1 = st_act ? st_d1 : 1; stk.dch = st_dch; stk.rows = st_rows;
    stk.old_hi = st_old_hi; stk.on = st_on; stk.lds = st_own_lds; stk.hist = keep;
    f32x4 st_pm = zero4, st_c0 = zero4, st_n0 = zero4;
    stk.s_end = zero4;
#pragma unroll
    for (int j = 0; j < 4; ++j) stk.q[j] = zero4;
    if (st_act) {   // scalar; a thread without a column reads zeros (global) or column 0 (LDS) and stores nothing
      // rows outside the own range live where THEIR depth says: LDS below KL, the global slab above
      if (st_d0 > 0 && st_d0 - 1 <= st_old_hi)
        st_pm = (st_d0 - 1 < KL) ? ld4(stk.sl + (size_t)(st_d0 - 1) * stk.rstr) : stk.ldg(st_d0 - 1);
      if (st_d1 <= st_old_hi && st_d1 < D)
        stk.s_end = (st_d1 < KL) ? ld4(stk.sl + (size_t)st_d1 * stk.rstr) : stk.ldg(st_d1);
      if (st_d0 <= st_old_hi) st_c0 = stk.ldrow(st_d0);
      if (st_d0 + 1 <= st_old_hi && st_d0 + 1 < st_d1) st_n0 = stk.ldrow(st_d0 + 1);
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (st_d0 + 2 + j <= st_old_hi && st_d0 + 2 + j < st_d1) stk.q[j] = stk.ldrow(st_d0 + 2 + j);
    }

    // ------------------------------------------------------------ extras GEMM
    if (!(dbg & 0x200)) {
      const float* xrow = h_lds + (size_t)((L - 1) * 16 + b) * HP + 4 * q;
      for (int xt = wave; xt < a.NXT; xt += WV) {
        f32x4 acc = ld4(a.BX + 16 * xt + 4 * q);
        acc = gemm1(acc, wrs, voff, a.oWX + (unsigned)xt * (unsigned)KGH * 1024u, KGH, xrow);
        st4(&xex[b * XP + 16 * xt + 4 * q], acc);
      }
    }
    PMG_TICK(0)
    __syncthreads();  // B1
    PMG_TICK(1)

    // ------------------------------------------------------------ consume logits of step i-1
    // Every wave takes two rows; a row's V logits are spread over 32 lanes (entry v sits in lane
    // v&31, slot v>>5), so max / sum / argmax are 5-step butterflies and the sampler's CDF is a
    // 32-lane inclusive scan per slot (oracle twin: oracle/philox.py sample_row).
    if (i >= 1 && !(dbg & 0x2000)) {
      const int bb = cs_bb, l5 = lane & 31;
      const int row = row0 + bb;
      const bool rowok = row < B;
      const float* lg = xex + bb * XP + 3 + W;
      if (!decode) {
        float m = -INFINITY;
        for (int v = l5; v < V; v += 32) m = fmaxf(m, lg[v]);
#pragma unroll
        for (int o = 16; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
        float total = 0.0f;
        for (int v = l5; v < V; v += 32) total += __expf(lg[v] - m);
#pragma unroll
        for (int o = 16; o >= 1; o >>= 1) total += __shfl_xor(total, o);
        if (teacher) {
          const int y = cs_y;
          if ((y & 31) == l5) {
            const float lp = (lg[y] - m) - __logf(total);
            a.LP[((size_t)(i - 1) * G + g) * 16 + bb] = lp;
            if (rowok) a.logp_out[(size_t)(i - 1) * B + row] = lp;
            misc[bb] = y;
          }
          float* pr = a.PR + ((size_t)(i - 1) * G + g) * a.VP * 16;
          const float inv = __builtin_amdgcn_rcpf(total);
          for (int v = l5; v < V; v += 32) pmg_st1h<1>(pr + v * 16 + bb, __expf(lg[v] - m) * inv);
        } else if (a.logits_out != nullptr && rowok) {
          for (int v = l5; v < V; v += 32) a.logits_out[(size_t)row * V + v] = lg[v];
        }
      } else {
        const bool greedy = !(a.temperature > 0.0f);
        int tok;
        if (greedy) {
          float bv = -INFINITY; int bi = 0x7fffffff;
          for (int v = l5; v < V; v += 32) { const float x = lg[v]; if (x > bv) { bv = x; bi = v; } }
#pragma unroll
          for (int o = 16; o >= 1; o >>= 1) {
            const float ov = __shfl_xor(bv, o); const int oi = __shfl_xor(bi, o);
            if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
          }
          tok = bi;
        } else {
          const float T = a.temperature;
          float m = -INFINITY;
          for (int v = l5; v < V; v += 32) m = fmaxf(m, lg[v] / T);
#pragma unroll
          for (int o = 16; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
          const float u = pmg_uniform(a.seed, (uint64_t)(a.seq_base + row), a.offset + (uint64_t)(i - 1));
          tok = -1;
          if (V <= 256) {
            // running CDF: slot k holds entries 32k..32k+31, one per lane
            float cdf[8];
            float carry = 0.0f;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
              cdf[k] = 0.0f;
              if (32 * k < V) {
                const int v = 32 * k + l5;
                float e = (v < V) ? __expf(lg[v] / T - m) : 0.0f;
#pragma unroll
                for (int o = 1; o <= 16; o <<= 1) {
                  const float t = __shfl_up(e, o, 32);
                  if (l5 >= o) e += t;
                }
                cdf[k] = carry + e;
                carry = __shfl(cdf[k], 31, 32);
              }
            }
            const float target = u * carry;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
              if (32 * k < V) {
                const int v = 32 * k + l5;
                const unsigned long long bal = __ballot((v < V) && (target < cdf[k]));
                const unsigned mk = (unsigned)(bal >> (lane & 32));
                if (tok < 0 && mk) tok = 32 * k + (__ffs(mk) - 1);
              }
            }
          } else {
            // larger vocabularies: the same slot-wise CDF, formed twice (first for its total, then
            // for the search) instead of kept in registers; the values are identical both times
            const int nslot = (V + 31) >> 5;     // uniform: both rows of a wave run in lockstep
            float carry = 0.0f;
#pragma unroll 1
            for (int k = 0; k < nslot; ++k) {
              const int v = 32 * k + l5;
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
#pragma unroll 1
            for (int k = 0; k < nslot; ++k) {
              const int v = 32 * k + l5;
              float e = (v < V) ? __expf(lg[v] / T - m) : 0.0f;
#pragma unroll
              for (int o = 1; o <= 16; o <<= 1) {
                const float t = __shfl_up(e, o, 32);
                if (l5 >= o) e += t;
              }
              const float c = carry + e;
              const unsigned long long bal = __ballot((v < V) && (target < c));
              const unsigned mk = (unsigned)(bal >> (lane & 32));
              if (tok < 0 && mk) tok = 32 * k + (__ffs(mk) - 1);
              carry = __shfl(c, 31, 32);
            }
          }
          if (tok < 0) tok = V - 1;
        }
        if (rowok && l5 == 0) a.tok_out[(size_t)row * N + (i - 1)] = tok;
        if (!finished && tok == a.stop_tok) { finished = true; mylen = i - 1; }
        if (l5 == 0) misc[bb] = tok;
        const bool alld = __all(finished || !rowok);
        if (lane == 0) misc[16 + wave] = (alld && !(a.flags & PMG_FLAG_NO_EARLY_EXIT)) ? 1 : 0;
      }
    }
    PMG_TICK(2)
    if (i == N) break;

    // ------------------------------------------------------------ stack update, streamed under the gh GEMM
    // new[d] = no_op*old[d] + push*old[d-1] + pop*old[d+1]   (old[-1] = push value, old[D] = 0).
    // Each thread walks its depth range one row per side-job call; rows d0-1 .. d0+3 were
    // requested before the extras GEMM, every call requests row d+4.
    f32x4 apush = zero4, apop = zero4, anoop = zero4, pv = zero4;
    if (st_on) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float* cx = xex + (4 * st_bq + j) * XP;
        const float c0 = cx[0], c1 = cx[1], c2 = cx[2];
        const float m = fmaxf(c0, fmaxf(c1, c2));
        const float e0 = __expf(c0 - m), e1 = __expf(c1 - m), e2 = __expf(c2 - m);
        const float inv = 1.0f / ((e0 + e1) + e2);
        apush[j] = e0 * inv; apop[j] = e1 * inv; anoop[j] = e2 * inv;
        pv[j] = fast_tanh(cx[3 + st_w]);
      }
      if (st_d0 == 0 && keep) {
        if (st_w == 0) {
          float* ct = a.CT + ((size_t)i * G + g) * 64;
          st4(ct + 0 * 16 + 4 * st_bq, apush);
          st4(ct + 1 * 16 + 4 * st_bq, apop);
          st4(ct + 2 * 16 + 4 * st_bq, anoop);
        }
        st4(a.PV + ((size_t)i * G + g) * 1024 + st_w * 16 + 4 * st_bq, pv);
      }
    }
    // first row of every thread now (row 0 is the stack top the W_ih product needs right after
    // the W_hh product); the remaining rows stream under the W_hh GEMM, one per double K group
    stk.Snew = a.SB + ((size_t)st_slot_new * G + g) * a.SW; stk.anoop = anoop; stk.apush = apush; stk.apop = apop; stk.sj = 1;
    {
      const f32x4 pm = (st_d0 == 0) ? pv : ((st_d0 - 1 <= st_old_hi) ? st_pm : zero4);
      const f32x4 c0 = (st_d0 <= st_old_hi) ? st_c0 : zero4;
      const f32x4 n0 = (st_d0 + 1 <= st_old_hi) ? ((st_d0 + 1 == st_d1) ? stk.s_end : st_n0) : zero4;
      if (st_on) {
        const f32x4 nw = anoop * c0 + apush * pm + apop * n0;
        stk.strow(st_d0, nw);
        if (st_d0 == 0) {
#pragma unroll
          for (int j = 0; j < 4; ++j) xtop[(4 * st_bq + j) * PMG_XTP + st_w] = nw[j];
          if (keep) st4(a.TP + ((size_t)i * G + g) * 2048 + st_w * 16 + 4 * st_bq, nw);
        }
      }
      stk.s_prev = c0;
      stk.s_cur = n0;
    }

    // history slabs of this step (teacher mode) through buffer resources
    __amdgpu_buffer_rsrc_t gth0 = wrs, hhh0 = wrs;
    if (keep) {
      gth0 = pmg_rsrc(a.GT[0] + (((size_t)i * G + g) * UT) * 1024, (unsigned)H * 256u);
      hhh0 = pmg_rsrc(a.HH[0] + (((size_t)(i + 1) * G + g) * H) * 16, (unsigned)H * 64u);
    }
    // ------------------------------------------------------------ layer 0: gh GEMM
    PMG_TICK(7)
    // layer 0 starts every accumulator at zero: b_ih and b_hh{r,z} live in the token table, b_hn
    // (BG[0][3]) is added with the gates, so no product waits for a bias load
#pragma unroll
    for (int u = 0; u < UPW; ++u) ar[u] = az[u] = ah[u] = zero4;
    // the last pass leaves group 0 of the stack-top product in wpre (consumed after barrier B2)
    if (!(dbg & 0x400))
      gemm3<UPW, TPP>(ar, az, ah, wrs, voff, a.oWHH[0], KGH, h_lds + (size_t)b * HP + 4 * q, wave, stk, wpre, wpre_ok,
                      gemm3_first<UPW>(a.oWIH[0], wave, PMG_KS / 16), (unsigned)(PMG_KS / 16) * 1024u);
    PMG_TICK(8)
    while (stk.sj < st_dch) {  // rows the GEMM loop did not reach (four calls keep the ring slots aligned)
      stk.template step<0>(); stk.template step<1>(); stk.template step<2>(); stk.template step<3>();
    }
    PMG_TICK(3)
    __syncthreads();  // B2: stack top + token visible; every read of h_lds[0] (old) is done

    PMG_TICK(4)
    bool all_done = decode;
    if (decode) {
#pragma unroll
      for (int wv = 0; wv < WV; ++wv) all_done = all_done && (misc[16 + wv] != 0);
    }
    if (all_done) {
      // every row of this group has emitted stop_tok: fill the tail and leave
      for (int idx = tid; idx < 16 * (N - i); idx += NTHR) {
        const int bb = idx / (N - i), t = i + (idx - bb * (N - i));
        if (row0 + bb < B) a.tok_out[(size_t)(row0 + bb) * N + t] = a.stop_tok;
      }
      break;
    }

    // ------------------------------------------------------------ layer 0: stack-top GEMM, table, gates
    if (keep && tid < 256) {
      // one-hot rows of the token fed at this step, behind the stack-top rows of the same slab: the
      // B operand [stack top | one-hot(token)] of the weight-gradient job that yields dW_ih[0][:, E:]
      // and the per-token gate-table gradient in one pass (vocabularies of at most 64 tokens)
      const int v = tid >> 2, sq = tid & 3;
      f32x4 oh;
#pragma unroll
      for (int j = 0; j < 4; ++j) oh[j] = (row0 + 4 * sq + j < B && misc[4 * sq + j] == v) ? 1.0f : 0.0f;
      pmg_st4h<1>(a.TP + ((size_t)i * G + g) * 2048 + (64 + v) * 16 + 4 * sq, oh);
    }
    f32x4 bhn[UPW];
#pragma unroll
    for (int u = 0; u < UPW; ++u) {
      ai[u] = zero4;
      bhn[u] = bl4(bgr0, vq, (unsigned)(16 * (wave * UPW + u) + 3 * H) * 4u);
    }
    if (!(dbg & 0x800))
      gemm3<UPW, TPP>(ar, az, ai, wrs, voff, a.oWIH[0], PMG_KS / 16, xtop + b * PMG_XTP + 4 * q, wave, no_side, wpre,
                      !(dbg & 0x400), gemm3_first<UPW>(a.oWIH[0], wave, PMG_KS / 16), (unsigned)(PMG_KS / 16) * 1024u);
    {
      const unsigned vtok = ((unsigned)misc[b] * 3u * (unsigned)H + 4u * q) * 4u;  // this row's table line
#pragma unroll
      for (int u = 0; u < UPW; ++u) {
        const int ug = wave * UPW + u;
        const int j0 = 16 * ug + 4 * q;
        float* hp_ptr = &h_lds[(size_t)b * HP + j0];
        const f32x4 hp = ld4(hp_ptr);
        const unsigned so = (unsigned)(16 * ug) * 4u, gH = (unsigned)H * 4u;
        const f32x4 vr = ar[u] + bl4(gtr, vtok, so);
        const f32x4 vz = az[u] + bl4(gtr, vtok, so + gH);
        const f32x4 vi = ai[u] + bl4(gtr, vtok, so + 2u * gH);
        const f32x4 vh = ah[u] + bhn[u];
        f32x4 r, z, n, hn;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          r[j] = fast_sigmoid(vr[j]);
          z[j] = fast_sigmoid(vz[j]);
          n[j] = fast_tanh(vi[j] + r[j] * vh[j]);
          hn[j] = (hp[j] - n[j]) * z[j] + n[j];
        }
        st4(hp_ptr, hn);
        if (keep) {
          const unsigned gs = (unsigned)ug * 4096u;
          bs4<HAUX>(gth0, voff, gs, r); bs4<HAUX>(gth0, voff, gs + 1024u, z); bs4<HAUX>(gth0, voff, gs + 2048u, n);
          bs4<HAUX>(gth0, voff, gs + 3072u, vh);
#pragma unroll
          for (int j = 0; j < 4; ++j) bs1<HAUX>(hhh0, vrow, (unsigned)((16 * ug + j) * 16) * 4u, hn[j]);
        }
      }
    }
    PMG_TICK(5)
    __syncthreads();  // B3
    PMG_TICK(6)

    // ------------------------------------------------------------ layers 1..L-1
    for (int l = 1; MULTI && l < L; ++l) {
#pragma unroll
      for (int u = 0; u < UPW; ++u) {
        const int j0 = 16 * (wave * UPW + u) + 4 * q;
        ar[u] = ld4(a.BG[l] + 0 * H + j0);
        az[u] = ld4(a.BG[l] + 1 * H + j0);
        ai[u] = ld4(a.BG[l] + 2 * H + j0);
        ah[u] = ld4(a.BG[l] + 3 * H + j0);
      }
      gemm3<UPW, TPP>(ar, az, ah, wrs, voff, a.oWHH[l], KGH, h_lds + (size_t)(l * 16 + b) * HP + 4 * q, wave, no_side,
                      wpre, false, gemm3_first<UPW>(a.oWIH[l], wave, KGH), (unsigned)KGH * 1024u);
      gemm3<UPW, TPP>(ar, az, ai, wrs, voff, a.oWIH[l], KGH, h_lds + (size_t)((l - 1) * 16 + b) * HP + 4 * q, wave, no_side,
                      wpre, true, gemm3_first<UPW>(a.oWIH[l], wave, KGH), (unsigned)KGH * 1024u);
      __syncthreads();  // every read of h_lds[l] (old) is done
#pragma unroll
      for (int u = 0; u < UPW; ++u) {
        const int ug = wave * UPW + u;
        const int j0 = 16 * ug + 4 * q;
        float* hp_ptr = &h_lds[(size_t)(l * 16 + b) * HP + j0];
        const f32x4 hp = ld4(hp_ptr);
        f32x4 r, z, n, hn;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          r[j] = fast_sigmoid(ar[u][j]);
          z[j] = fast_sigmoid(az[u][j]);
          n[j] = fast_tanh(ai[u][j] + r[j] * ah[u][j]);
          hn[j] = (hp[j] - n[j]) * z[j] + n[j];
        }
        st4(hp_ptr, hn);
        if (keep) {
          float* gp = a.GT[l] + ((((size_t)i * G + g) * UT + ug) * 4) * 256 + lane * 4;
          pmg_st4h<1>(gp, r); pmg_st4h<1>(gp + 256, z); pmg_st4h<1>(gp + 512, n); pmg_st4h<1>(gp + 768, ah[u]);
          float* hh = a.HH[l] + (((size_t)(i + 1) * G + g) * H + j0) * 16 + b;
#pragma unroll
          for (int j = 0; j < 4; ++j) pmg_st1h<1>(hh + j * 16, hn[j]);
        }
      }
      __syncthreads();
    }
  }

#ifdef PMG_PROFILE
  if (prof && g == 0 && lane == 0)
    printf("fwd wave %d cycles: extras %lld syncB1 %lld logits %lld stackprep %lld gh %lld tail %lld syncB2 %lld top+gates %lld syncB3 %lld\n", wave,
           tph[0], tph[1], tph[2], tph[7], tph[8], tph[3], tph[4], tph[5], tph[6]);
#endif
  // ---------------------------------------------------------------- epilogue
  if (decode && (lane & 31) == 0 && row0 + cs_bb < B) a.len_out[row0 + cs_bb] = mylen;
  if (a.h_out != nullptr) {
    for (int idx = tid; idx < L * 16 * H; idx += NTHR) {
      const int l = idx / (16 * H);
      const int rem = idx - l * 16 * H;
      const int bb = rem / H, k = rem - bb * H;
      if (row0 + bb < B) a.h_out[((size_t)l * B + row0 + bb) * H + k] = h_lds[(size_t)(l * 16 + bb) * HP + k];
    }
  }
  if (stepm && a.stack_out != nullptr) {
    // step mode ran exactly one update: the new stack is in slot 1 (written by this workgroup)
    __syncthreads();
    const float* S1 = a.SB + ((size_t)1 * G + g) * a.SW;
    for (int idx = tid; idx < D * W * 16; idx += NTHR) {
      const int bb = idx & 15;
      const int dw = idx >> 4;
      if (row0 + bb < B) a.stack_out[(size_t)(row0 + bb) * D * W + dw] = S1[idx];
    }
  }
}

}  // namespace

static size_t fwd_lds_base_bytes(const FwdArgs* a) {
  return ((size_t)a->L * 16 * a->HP + 16 * PMG_XTP + (size_t)16 * a->XP + 32) * sizeof(float);
}
// decode: hidden state + the latent rows of the h0 product, which alias everything behind the hidden state
static size_t fwd_lds_init_bytes(const FwdArgs* a) {
  return a->mode == PMG_FWD_DECODE ? ((size_t)a->L * 16 * a->HP + (size_t)16 * (a->Z + 8)) * sizeof(float) : 0;
}

// PMG_FWD_LDS_STACK=0 keeps the whole stack in the global slab (A/B knob)
#ifndef PMG_FWD_LDS_STACK
#define PMG_FWD_LDS_STACK 1
#endif

// Stack rows kept in LDS: as many of the top rows as fit next to the hidden state in the CU's
// 160 KiB (a row is W x 16 sequences x 4 B).  The single-step form keeps none: its stack comes from
// and goes back to the caller's memory.
int pmg_fwd_stack_rows(const FwdArgs* a) {
  if (!PMG_FWD_LDS_STACK || a->mode == PMG_FWD_STEP) return 0;
  const size_t base = fwd_lds_base_bytes(a), row = (size_t)a->W * 64;
  if (base + row > (size_t)160 * 1024) return 0;
  const size_t k = ((size_t)160 * 1024 - base) / row;
  return (int)(k < (size_t)a->D ? k : (size_t)a->D);
}

size_t pmg_fwd_lds_bytes(const FwdArgs* a) {
  const size_t loop = fwd_lds_base_bytes(a) + (size_t)a->KL * a->W * 64, init = fwd_lds_init_bytes(a);
  return loop > init ? loop : init;
}

template <int WV, int UPW, int TPP, int MODE, bool MULTI>
static hipError_t launch_fwd_t(const FwdArgs* a, hipStream_t st) {
  const size_t lds = pmg_fwd_lds_bytes(a);
  auto kern = pmg_fwd_kernel<WV, UPW, TPP, MODE, MULTI>;
  static std::atomic<unsigned long long> lds_done{0};
  hipError_t e = pmg_full_lds_once(reinterpret_cast<const void*>(kern), &lds_done);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(kern, dim3(a->G), dim3(WV * 64), lds, st, *a);
  return hipGetLastError();
}

template <int WV, int UPW, int TPP>
static hipError_t launch_fwd_h(const FwdArgs* a, hipStream_t st) {
  const bool m = a->L > 1;
  switch (a->mode) {
    case PMG_FWD_DECODE:
      return m ? launch_fwd_t<WV, UPW, TPP, PMG_FWD_DECODE, true>(a, st) : launch_fwd_t<WV, UPW, TPP, PMG_FWD_DECODE, false>(a, st);
    case PMG_FWD_TEACHER:
      return m ? launch_fwd_t<WV, UPW, TPP, PMG_FWD_TEACHER, true>(a, st) : launch_fwd_t<WV, UPW, TPP, PMG_FWD_TEACHER, false>(a, st);
    case PMG_FWD_STEP:
      return m ? launch_fwd_t<WV, UPW, TPP, PMG_FWD_STEP, true>(a, st) : launch_fwd_t<WV, UPW, TPP, PMG_FWD_STEP, false>(a, st);
  }
  return hipErrorInvalidValue;
}

// PMG_FWD_WV512: waves per workgroup of the H=512 variant (tuning knob: 8 or 16)
#ifndef PMG_FWD_WV512
#define PMG_FWD_WV512 8
#endif
#ifndef PMG_FWD_TPP16
#define PMG_FWD_TPP16 1
#endif

hipError_t pmg_launch_fwd(const FwdArgs* a, hipStream_t st) {
  // small batches: the hidden-partitioned form (a cluster of H/16 CUs per 16 sequences), same results
  if (pmg_fwdc_applicable(a)) return pmg_launch_fwdc(a, st);
  if (a->mode == PMG_FWD_TEACHER && a->LAYW != nullptr) {   // history in the row-group-interleaved layout
    const hipError_t e = hipMemsetAsync(a->LAYW, 0, sizeof(unsigned), st);
    if (e != hipSuccess) return e;
  }
  switch (a->H) {
    case 128: return launch_fwd_h<8, 1, 1>(a, st);
    case 256: return launch_fwd_h<8, 2, 2>(a, st);
    // hidden sizes between the power-of-two ones (model_params.json is free to name them): the same
    // kernel with another number of unit tiles per wave
    case 384: return launch_fwd_h<8, 3, 1>(a, st);
    case 640: return launch_fwd_h<8, 5, 1>(a, st);
    case 768: return launch_fwd_h<8, 6, 2>(a, st);
    case 896: return launch_fwd_h<8, 7, 1>(a, st);
#if PMG_FWD_WV512 == 16
    case 512: return launch_fwd_h<16, 2, PMG_FWD_TPP16>(a, st);
#else
    case 512: return launch_fwd_h<8, 4, 2>(a, st);
#endif
    case 1024: return launch_fwd_h<8, 8, 2>(a, st);
  }
  return hipErrorInvalidValue;
}
