// Shared definitions for the gfx950 kernels behind include/pmg.h.
//
// Work decomposition (DESIGN.md "Kernels"): one workgroup = 16 sequences ("a row
// slab") for ALL time steps; 8 waves; every GEMM is  D[feature][row] = W * X^T  on
// v_mfma_f32_16x16x4_f32 with the weight tile as the A operand (streamed from L2 in
// a pre-packed lane-major layout) and the 16 rows' activations as the B operand
// (resident in LDS).  Lane l of a wave holds row b = l&15 and features 4*(l>>4)..+3
// of each 16-feature tile ("D layout").
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/pmg.h"

#define PMG_ROWS 16     // sequences per workgroup
#define PMG_WAVES 8     // waves per workgroup
#define PMG_THREADS 512 // PMG_WAVES * 64
#define PMG_KS 64       // stack width padded to a multiple of 16 (K of the stack-top GEMM)
#define PMG_MAX_VOCAB 1024
#define PMG_XTP 72      // LDS pitch of the stack-top tile (64 + 8: conflict-free ds_read_b128)
#define PMG_CHUNK 128   // hidden units per backward chunk (PMG_WAVES * 16)
#define PMG_CP 520      // LDS pitch of a backward chunk row (4*128 + 8)
#ifndef PMG_SBLKB
#define PMG_SBLKB 5     // stack rows per load burst of the backward (2 loads per row in flight)
#endif

typedef float f32x4 __attribute__((ext_vector_type(4)));

// -DPMG_PROFILE (in-kernel cycle counters) implies -DPMG_ABLATE (phase-skipping bits); neither is
// part of the default library.
#if defined(PMG_PROFILE) && !defined(PMG_ABLATE)
#define PMG_ABLATE 1
#endif
#ifdef PMG_ABLATE
#define PMG_DBG(x) (x)
#else
#define PMG_DBG(x) 0
#endif

// error codes (negative returns of the C ABI)
enum {
  PMG_OK = 0,
  PMG_E_NULL = -1,
  PMG_E_SHAPE = -2,
  PMG_E_HIDDEN = -3,
  PMG_E_STACK = -4,
  PMG_E_VOCAB = -5,
  PMG_E_LAYERS = -6,
  PMG_E_ALIGN = -7,
  PMG_E_MODE = -8,
  PMG_E_ARG = -9,
  PMG_E_LDS = -10,
  PMG_E_SCRATCH = -11,
};

// Derived dimensions, identical on host and device.
struct PmgDims {
  int B, G, T, H, E, W, D, V, L, Z, pad_idx;
  int NX;   // 3 + W + V  : rows of the "extras" matrix [ctrl | push | out]
  int NXT;  // ceil(NX/16)
  int XP;   // LDS pitch of the extras tile: NXT*16 + 8
  int HP;   // LDS pitch of a hidden tile: H + 8
  int KGH;  // H/16
  int UT;   // H/16 unit tiles
  int UPW;  // unit tiles per wave = H/128
  int VP;   // V rounded up to 16
  int SW;   // D*W*16 floats: one stack slab
};

static inline __host__ __device__ int pmg_ceil_div(int a, int b) { return (a + b - 1) / b; }

static inline __host__ PmgDims pmg_dims(const pmg_shape* s) {
  PmgDims d;
  d.B = s->B; d.G = pmg_ceil_div(s->B, PMG_ROWS); d.T = s->T; d.H = s->H; d.E = s->E; d.W = s->W;
  d.D = s->D; d.V = s->V; d.L = s->L; d.Z = s->Z; d.pad_idx = s->pad_idx;
  d.NX = 3 + s->W + s->V;
  d.NXT = pmg_ceil_div(d.NX, 16);
  d.XP = d.NXT * 16 + 8;
  d.HP = s->H + 8;
  d.KGH = s->H / 16;
  d.UT = s->H / 16;
  d.UPW = s->H / 128;
  d.VP = pmg_ceil_div(s->V, 16) * 16;
  d.SW = s->D * s->W * 16;
  return d;
}

// ---------------------------------------------------------------------------------
// Workspace layout (float offsets from the 256-B aligned base).  Filled by
// pmg_make_layout() in pmg_api.hip; every kernel gets the pointers it needs.
// ---------------------------------------------------------------------------------
struct PmgLayout {
  // forward packs
  size_t WXp, BX, GTAB;
  size_t WHHp[PMG_MAX_LAYERS], WIHp[PMG_MAX_LAYERS], BG[PMG_MAX_LAYERS];
  // backward packs
  size_t WXt, WHHt[PMG_MAX_LAYERS], WIHt[PMG_MAX_LAYERS];
  // state
  size_t H0;  // [L][Bp][H] decode: latent_to_hidden output
  size_t SB;  // decode: 2 stack slabs per row group;  train: T slabs
  // history (train)
  size_t HH[PMG_MAX_LAYERS];  // [T][G][H][16]
  size_t GT[PMG_MAX_LAYERS];  // [T-1][G][UT][4][64][4]
  size_t CT, PV, TP, PR;      // [T-1][G][{4,64,128,VP}][16]  (TP: stack top rows, then one-hot rows of the fed token)
  size_t DG[PMG_MAX_LAYERS];  // [T-1][G][4H][16]
  size_t DX;                  // [T][G][NXT*16][16]
  size_t DS;                  // [2][G][SW]
  size_t LOSSP;               // [G] per-group loss partials
  size_t LP;                  // [T-1][G][16] log-probs kept for the loss
  size_t DHC;                 // [G][L][H*16] hidden-gradient carry
  size_t WG;                  // weight-gradient accumulators (packed "natural" matrices)
  size_t WGP;                 // split-K partials
  size_t CX, CXS, LAYW;             // clustered forward: exchange buffers, barrier counters (0: none)
  size_t pack_end;            // end of the packed-weight region (covered by one buffer resource)
  size_t total;               // floats
};

// ---------------------------------------------------------------------------------
// device helpers
// ---------------------------------------------------------------------------------
__device__ __forceinline__ f32x4 mfma4(f32x4 acc, const float4 w, const float4 x) {
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w.x, x.x, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w.y, x.y, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w.z, x.z, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w.w, x.w, acc, 0, 0, 0);
  return acc;
}

// 16-byte load through a buffer resource: SGPR descriptor + one 32-bit per-lane offset
// (voff) + a wave-uniform byte offset (soff).  No 64-bit address VGPRs; out-of-range reads
// return 0.
typedef unsigned int pmg_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 pmg_bload(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
  const pmg_u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0);
  return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t pmg_rsrc(const void* base, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, bytes, 0x00020000);
}

// History streams (written once by the forward, read once by the backward / weight-gradient
// GEMM much later) can be issued non-temporal so that they do not evict the packed weights from
// the 4 MB L2s.  PMG_NT is a compile-time mask choosing which streams are (tuning knob).
#ifndef PMG_NT
#define PMG_NT 12
#endif
template <int BIT> __device__ __forceinline__ void pmg_st4h(float* p, const f32x4 v) {
  if (PMG_NT & BIT) __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(p));
  else *reinterpret_cast<f32x4*>(p) = v;
}
template <int BIT> __device__ __forceinline__ void pmg_st1h(float* p, const float v) {
  if (PMG_NT & BIT) __builtin_nontemporal_store(v, p);
  else *p = v;
}
template <int BIT> __device__ __forceinline__ f32x4 pmg_ld4h(const float* p) {
  if (PMG_NT & BIT) return __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p));
  return *reinterpret_cast<const f32x4*>(p);
}
template <int BIT> __device__ __forceinline__ float pmg_ld1h(const float* p) {
  if (PMG_NT & BIT) return __builtin_nontemporal_load(p);
  return *p;
}

__device__ __forceinline__ float pmg_sigmoid(float x) { return 1.0f / (1.0f + expf(-x)); }

// Philox4x32-10 (Salmon et al. 2011).  Returns the first output word.
__host__ __device__ inline uint32_t pmg_philox_first(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                                     uint32_t k0, uint32_t k1) {
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
    const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
    const uint32_t n1 = (uint32_t)p1;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    const uint32_t n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  return c0;
}

// Uniform in (0,1) with 24 random bits, exactly representable in fp32.
__host__ __device__ inline float pmg_uniform(uint64_t seed, uint64_t seq, uint64_t step) {
  const uint32_t x = pmg_philox_first((uint32_t)seq, (uint32_t)(seq >> 32), (uint32_t)step, (uint32_t)(step >> 32),
                                      (uint32_t)seed, (uint32_t)(seed >> 32));
  return ((float)(x >> 8) + 0.5f) * (1.0f / 16777216.0f);
}
