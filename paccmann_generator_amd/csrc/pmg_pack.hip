// Weight packing: torch layouts -> MFMA A-operand tiles.
//
// A packed matrix is [tile][kg][lane][4]: tile = 16 output rows, kg = 16 reduction
// columns; lane l holds row (l&15), columns 16*kg + 4*(l>>4) + {0..3}.  One wave
// loads one (tile,kg) block as a single contiguous 1 KiB global_load_dwordx4, and
// the four values feed four v_mfma_f32_16x16x4_f32 whose hardware k index (l>>4)
// stands for column 4*(l>>4)+c in both operands.
//
// Runs once per entry-point call (weights change after every optimiser step,
// reference: reinforce_omics.py:341); ~2x the parameter bytes of traffic.
#include "pmg_kernels.h"

// What a packing job reads, in registers (a copy of PackArgs in private memory would land in scratch).
struct PackJob {
  int kind, tiles, KG, H, E, W, NX, UPW, Z;
  const float* src; const float* src2; const float* src3;
  float* dst;
};

// Row n, column k of the virtual extras matrix [ctrl | push | out] (K = H).
__device__ __forceinline__ float extras_at(const float* s1, const float* s2, const float* s3, int H, int W, int NX, int n, int k) {
  if (n < 3) return s1[(size_t)n * H + k];
  if (n < 3 + W) return s2[(size_t)(n - 3) * H + k];
  if (n < NX) return s3[(size_t)(n - 3 - W) * H + k];
  return 0.0f;
}

// Backward reduction position rr (chunk order) -> gate row n in torch's [3H] order.
// chunk c holds, for each of the 3 pieces, the c-th unit tile of every wave.
__device__ __forceinline__ int chunk_row(int rr, int H, int UPW) {
  const int c = rr / (3 * PMG_CHUNK);
  const int rem = rr - c * (3 * PMG_CHUNK);
  const int piece = rem / PMG_CHUNK;
  const int within = rem - piece * PMG_CHUNK;
  const int wv = within >> 4, ii = within & 15;
  return piece * H + 16 * (wv * UPW + c) + ii;
}

// One packing job, worked by blocks lb = 0..nb-1 of the prepare launch.
__device__ __forceinline__ void pack_job(const PackJob a, const int lb, const int nb) {
  const size_t total = (size_t)a.tiles * a.KG * 64;
  for (size_t idx = (size_t)lb * blockDim.x + threadIdx.x; idx < total; idx += (size_t)nb * blockDim.x) {
    const int lane = (int)(idx & 63);
    const size_t tk = idx >> 6;
    const int kg = (int)(tk % a.KG);
    const int tile = (int)(tk / a.KG);
    const int i = lane & 15;
    const int kb = 16 * kg + 4 * (lane >> 4);
    float v[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int k = kb + c;
      float x = 0.0f;
      switch (a.kind) {
        case PK_WX: x = extras_at(a.src, a.src2, a.src3, a.H, a.W, a.NX, 16 * tile + i, k); break;
        case PK_WHH: {
          const int u = tile / 3, g = tile - 3 * u;
          x = a.src[(size_t)(g * a.H + 16 * u + i) * a.H + k];
        } break;
        case PK_WIS: {
          const int u = tile / 3, g = tile - 3 * u;
          x = (k < a.W) ? a.src[(size_t)(g * a.H + 16 * u + i) * (a.E + a.W) + a.E + k] : 0.0f;
        } break;
        case PK_WIH: {
          const int u = tile / 3, g = tile - 3 * u;
          x = a.src[(size_t)(g * a.H + 16 * u + i) * a.H + k];
        } break;
        case PK_L2H: x = a.src[(size_t)(16 * tile + i) * a.Z + k]; break;
        case PK_WXT: x = extras_at(a.src, a.src2, a.src3, a.H, a.W, a.NX, k, 16 * tile + i); break;
        case PK_WHHT: x = a.src[(size_t)chunk_row(k, a.H, a.UPW) * a.H + 16 * tile + i]; break;
        case PK_WIST: {
          const int w = 16 * tile + i;
          x = (w < a.W) ? a.src[(size_t)chunk_row(k, a.H, a.UPW) * (a.E + a.W) + a.E + w] : 0.0f;
        } break;
        case PK_WIHT: x = a.src[(size_t)chunk_row(k, a.H, a.UPW) * a.H + 16 * tile + i]; break;
      }
      v[c] = x;
    }
    reinterpret_cast<float4*>(a.dst)[idx] = make_float4(v[0], v[1], v[2], v[3]);
  }
}

// Bias vectors and the per-token gate table.
//   BX[n]        = extras bias (ctrl | push | out), zero padded to NXT*16
//   BG[l][4][H]  = l==0: (0, 0, 0, b_hn)   (b_ih and b_hh{r,z} live in GTAB)
//                  l>0 : (b_ir+b_hr, b_iz+b_hz, b_in, b_hn)
//   GTAB[v][3H]  = W_ih0[:, :E] @ emb[v] + b_ih0 + (b_hh0 for the r,z rows)
__device__ __forceinline__ void bias_job(const AuxArgs& a, const int lb, const int nb) {
  const int H = a.H;
  const size_t nbx = (size_t)a.NXT * 16;
  const size_t nbg = (size_t)a.L * 4 * H;
  for (size_t idx = (size_t)lb * blockDim.x + threadIdx.x; idx < nbx + nbg; idx += (size_t)nb * blockDim.x) {
    if (idx < nbx) {
      const int n = (int)idx;
      float x = 0.0f;
      if (n < 3) x = a.b_ctrl[n];
      else if (n < 3 + a.W) x = a.b_push[n - 3];
      else if (n < a.NX) x = a.b_out[n - 3 - a.W];
      a.BX[n] = x;
    } else {
      const size_t r = idx - nbx;
      const int l = (int)(r / (4 * H));
      const int piece = (int)((r / H) & 3);
      const int j = (int)(r % H);
      const float* bih = a.b_ih[0]; const float* bhh = a.b_hh[0]; float* bg = a.BG[0];
#pragma unroll
      for (int k = 1; k < PMG_MAX_LAYERS; ++k)
        if (k == l) { bih = a.b_ih[k]; bhh = a.b_hh[k]; bg = a.BG[k]; }
      float x;
      if (piece == 3) x = bhh[2 * H + j];
      else if (l == 0) x = 0.0f;
      else if (piece == 2) x = bih[2 * H + j];
      else x = bih[piece * H + j] + bhh[piece * H + j];
      bg[piece * H + j] = x;
    }
  }
}

// Token-table job: block lb forms 16 gate rows x 64 tokens.  The 16 weight rows and the 64 embedding rows are staged
// through LDS in slices of 128 embedding columns (coalesced reads; the row stride of W_ih0 is E+W floats, which a
// thread-per-entry loop would walk with one cache line per load); every entry is one fused multiply-add chain over
// e = 0..E-1 in that order, then + b_ih (+ b_hh).
#define PMG_TAB_EC 128
__device__ __forceinline__ void table_job(const AuxArgs& a, const int lb, float* wl, float* el) {
  const int H = a.H, E = a.E, V = a.V, tid = threadIdx.x;
  const int nt = 3 * H / 16;
  const int n0 = 16 * (lb % nt), v0 = 64 * (lb / nt);
  const int nl = tid & 15, vl = tid >> 4;
  float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
  for (int e0 = 0; e0 < E; e0 += PMG_TAB_EC) {
    const int ec = min(PMG_TAB_EC, E - e0);
    __syncthreads();
    for (int i = tid; i < 16 * ec; i += 256) {
      const int r = i / ec, e = i - r * ec;
      wl[r * (PMG_TAB_EC + 1) + e] = a.w_ih0[(size_t)(n0 + r) * (E + a.W) + e0 + e];
    }
    for (int i = tid; i < 64 * ec; i += 256) {
      const int r = i / ec, e = i - r * ec;
      el[r * (PMG_TAB_EC + 1) + e] = (v0 + r < V) ? a.emb[(size_t)(v0 + r) * E + e0 + e] : 0.0f;
    }
    __syncthreads();
    for (int e = 0; e < ec; ++e) {
      const float w = wl[nl * (PMG_TAB_EC + 1) + e];
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[j] = fmaf(w, el[(vl + 16 * j) * (PMG_TAB_EC + 1) + e], acc[j]);
    }
  }
  const int n = n0 + nl;
  const float bias = a.b_ih[0][n];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int v = v0 + vl + 16 * j;
    if (v < V) {
      float s = acc[j] + bias;
      if (n < 2 * H) s += a.b_hh[0][n];
      a.GTAB[(size_t)v * 3 * H + n] = s;
    }
  }
}

// Everything an entry point prepares from the caller's weights, in ONE launch: the packing jobs, the bias vectors
// and the token table each own a range of blocks (PrepArgs::first).  Ten dependent launches of a few microseconds
// each cost a small-batch step 5 % (profiles/r03_prepare.md).
__global__ void __launch_bounds__(256) pmg_prep_kernel(PrepArgs a) {
  __shared__ float wl[16 * (PMG_TAB_EC + 1)];
  __shared__ float el[64 * (PMG_TAB_EC + 1)];
  // constant indices throughout: a dynamic index into the kernel argument block would move it to scratch
  int job = 0, f0 = 0, f1 = a.first[1];
#pragma unroll
  for (int j = 1; j < PMG_PREP_MAX + 2; ++j)
    if (j <= a.npack + 1 && (int)blockIdx.x >= a.first[j]) { job = j; f0 = a.first[j]; f1 = a.first[j + 1]; }
  const int lb = (int)blockIdx.x - f0, nb = f1 - f0;
  if (job < a.npack) {
    // the shape fields are the same in every job
    PackJob pj = {a.p[0].kind, a.p[0].tiles, a.p[0].KG, a.p[0].H, a.p[0].E, a.p[0].W, a.p[0].NX, a.p[0].UPW, a.p[0].Z,
                  a.p[0].src, a.p[0].src2, a.p[0].src3, a.p[0].dst};
#pragma unroll
    for (int j = 1; j < PMG_PREP_MAX; ++j)
      if (j == job) {
        pj.kind = a.p[j].kind; pj.tiles = a.p[j].tiles; pj.KG = a.p[j].KG;
        pj.src = a.p[j].src; pj.src2 = a.p[j].src2; pj.src3 = a.p[j].src3; pj.dst = a.p[j].dst;
      }
    pack_job(pj, lb, nb);
  }
  else if (job == a.npack) bias_job(a.x, lb, nb);
  else table_job(a.x, lb, wl, el);
}

void pmg_prep_add(PrepArgs* a, const PackArgs* p) {
  const size_t total = (size_t)p->tiles * p->KG * 64;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 2048) blocks = 2048;
  if (blocks < 1) blocks = 1;
  a->p[a->npack] = *p;
  a->first[a->npack + 1] = a->first[a->npack] + blocks;
  ++a->npack;
}

hipError_t pmg_launch_prep(PrepArgs* a, hipStream_t st) {
  const AuxArgs& x = a->x;
  const size_t nbias = (size_t)x.NXT * 16 + (size_t)x.L * 4 * x.H;
  const int bias_blocks = (int)((nbias + 255) / 256);
  const int tab_blocks = (3 * x.H / 16) * ((x.V + 63) / 64);
  a->first[a->npack + 1] = a->first[a->npack] + bias_blocks;
  a->first[a->npack + 2] = a->first[a->npack + 1] + tab_blocks;
  hipLaunchKernelGGL(pmg_prep_kernel, dim3(a->first[a->npack + 2]), dim3(256), 0, st, *a);
  return hipGetLastError();
}
