// Weight gradients of the REINFORCE step: one large-K "row-slab GEMM".
//
// The backward recurrence leaves, per time step p and row group g (16 sequences), the
// gate/extras gradients as slabs [row][16] and the forward left the activations as
// slabs [col][16].  Every parameter gradient is
//     C[row][col] = sum_{p,g} sum_{b<16} A(p,g)[row][b] * B(p,g)[col][b]
// i.e. a GEMM whose reduction runs over all B*T tokens.  Both operands are contiguous
// along the reduction inside a slab, so a lane's four reduction values are one
// dwordx4 load and feed four v_mfma_f32_16x16x4_f32 directly -- no LDS transpose.
//
// Jobs (reference: the autograd of reinforce_omics.py:322-335 w.r.t. each parameter):
//   HH[l]  dW_hh[l]      = [dr|dz|dghn]^T  h^l_{p-1}
//   IH[l]  dW_ih[l>0]    = [dr|dz|dn]^T    h^{l-1}_p
//   IS     dW_ih[0][:,E:]= [dr|dz|dn]^T    stack_top_p
//   TAB    dGTAB         = [dr|dz|dn]^T    onehot(tok_p)   -> embedding, W_ih[0][:, :E]
//          (vocabularies of at most 64 tokens: IS and TAB are ONE plain job, the forward having left
//          [stack top | one-hot(tok)] as one 128-row slab; larger ones: TAB is a job of its own whose B
//          operand is generated from the token ids)
//   X      d[W_ctrl;W_push;W_out] = [dctrl_p|dpush_p|dlogits_{p-1}]^T h^{top}_{p-1}
// Bias gradients are the row sums of the A operands.
// Split-K over slabs into per-split partial tiles, then a deterministic reduce and a
// small finalize kernel that writes torch-layout gradients.
#include "pmg_kernels.h"
#include <string.h>

#define WG_TM 128
#define WG_THREADS 256
#define WG_MAXJOBS (2 + 2 * PMG_MAX_LAYERS + 1)
#define WG_PLAIN 0
#define WG_ONEHOT 1

struct WJob {
  const float* A;   // slabs [P][G][a_rows][16]
  const float* B;   // slabs [P(+shift)][G][b_rows][16]  (unused when mode == WG_ONEHOT)
  int mode;         // WG_PLAIN or WG_ONEHOT (B = one-hot of the tokens, generated in the kernel)
  int a_rows;       // rows of one A slab
  int skip_mid;     // 1: job row r >= 2H maps to slab row r + H (dgh: skip the dn piece)
  int b_rows;       // rows of one B slab
  int b_pshift;     // time shift of the B slabs
  int rows, cols;   // logical C size
  int tn;           // column tile: 128 (wide kernel) or 64 (narrow kernel)
  int row_tiles, col_tiles;
  int tile0;        // first tile index within its kernel variant's grid
  int rs0;          // offset of this job's row sums in the row-sum vector
  int P;            // time slabs
  int S;            // split-K factor of the launch this job belongs to
  size_t c_off;     // offset of the dense C in the accumulator region
  size_t pt_off;    // offset of this job's first partial tile within one split's partial block
};

struct WgradArgs {
  int njobs, total_rs, S;
  int G, B, H;
  int tiles_wide, tiles_narrow;
  size_t split_stride;  // floats of partial tiles per split
  const int32_t* tok32;  // [N][G][16] token of (step, row group, row), -1 for rows past the batch (pmg_tok32_kernel)
  WJob jobs[WG_MAXJOBS];
  float* partial;     // [S][split_stride]
  float* rs_partial;  // [S][total_rs]
  float* dense;       // accumulators
};

namespace {

// Wave tile = (16*MI) x (16*NJ); workgroup = WR x WC waves; TM = 16*MI*WR = 128.
// ONEHOT: the B operand is generated from the token ids (embedding-table job).
// The hot loop is branch-free: out-of-range row/column fragments are clamped to tile 0 of
// the slab and their (garbage) results are simply never read back.  Operands stream
// through buffer resources (SGPR descriptor + one lane-constant VGPR offset + a uniform
// byte offset per fragment); two register sets keep the next slab in flight.
template <int MI, int NJ, int WR, int WC, int MODE>
__global__ __launch_bounds__(WG_THREADS, 2) void pmg_wgrad_kernel(const WgradArgs a) {
  constexpr int TN = 16 * NJ * WC;
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63;
  const int n16 = lane & 15, q = lane >> 4;
  const int wr = wave / WC, wc = wave - wr * WC;
  // tiles of this variant
  int ntiles = 0;
  for (int k = 0; k < a.njobs; ++k)
    if (a.jobs[k].tn == TN && a.jobs[k].mode == MODE) ntiles += a.jobs[k].row_tiles * a.jobs[k].col_tiles;
  // XCD-aware remap: blocks b and b+8 share an XCD; give each XCD a contiguous run of
  // virtual ids so that the tiles of one split (same slabs) meet in one L2.
  const int nblk = gridDim.x;  // multiple of 8
  int jS = 1;
  for (int k = 0; k < a.njobs; ++k)
    if (a.jobs[k].tn == TN && a.jobs[k].mode == MODE) jS = a.jobs[k].S;
  const int v = (blockIdx.x & 7) * (nblk >> 3) + (blockIdx.x >> 3);
  if (v >= ntiles * jS) return;
  const int s = v / ntiles;
  int t = v - s * ntiles;
  int ji = 0;
  for (int k = 0; k < a.njobs; ++k) {
    if (a.jobs[k].tn == TN && a.jobs[k].mode == MODE) {
      const int nt = a.jobs[k].row_tiles * a.jobs[k].col_tiles;
      if (t < nt) { ji = k; break; }
      t -= nt;
    }
  }
  const WJob& jb = a.jobs[ji];
  const int tl = t;
  const int rt = tl / jb.col_tiles, ct = tl - rt * jb.col_tiles;
  const int r0 = rt * WG_TM + wr * 16 * MI;
  const int c0 = ct * TN + wc * 16 * NJ;
  const int H = a.H, G = a.G;

  const int nslab = jb.P * G;
  const int per = (nslab + jb.S - 1) / jb.S;
  const int sl0 = s * per;
  const int sl1 = (sl0 + per < nslab) ? sl0 + per : nslab;
  const int ns = sl1 - sl0;

  // uniform byte offsets of each fragment's 16-row tile inside a slab
  unsigned arow[MI], brow[NJ];
#pragma unroll
  for (int i = 0; i < MI; ++i) {
    int rr = r0 + 16 * i;
    if (rr >= jb.rows) rr = 0;  // clamp: result rows are ignored
    if (jb.skip_mid && rr >= 2 * H) rr += H;
    arow[i] = (unsigned)rr * 64u;
  }
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    int cc = c0 + 16 * j;
    if (cc >= jb.cols) cc = 0;
    brow[j] = (unsigned)cc * 64u;
  }
  // one-hot columns: token id that column n16 of fragment j stands for
  constexpr bool onehot = (MODE == WG_ONEHOT);
  const int oh0 = c0;
  const unsigned voff = (unsigned)(n16 * 16 + 4 * q) * 4u;
  const unsigned astr = (unsigned)jb.a_rows * 64u, bstr = (unsigned)jb.b_rows * 64u;
  const int nsafe = ns > 0 ? ns : 0;
  const __amdgpu_buffer_rsrc_t ars = pmg_rsrc(jb.A + (size_t)sl0 * jb.a_rows * 16, (unsigned)nsafe * astr);
  const __amdgpu_buffer_rsrc_t brs =
      (MODE == WG_ONEHOT) ? ars : pmg_rsrc(jb.B + ((size_t)sl0 + (size_t)jb.b_pshift * G) * jb.b_rows * 16, (unsigned)nsafe * bstr);

  f32x4 acc[MI][NJ];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float rsum[MI];
#pragma unroll
  for (int i = 0; i < MI; ++i) rsum[i] = 0.f;

#define WG_LOAD_A(dst, k) \
  _Pragma("unroll") for (int i = 0; i < MI; ++i) dst[i] = pmg_bload(ars, voff, (unsigned)(k) * astr + arow[i]);
  // token ids of the slab rows 4q..4q+3, one dwordx4 per lane from the per-slab table
  const __amdgpu_buffer_rsrc_t trs = pmg_rsrc(reinterpret_cast<const float*>(a.tok32) + (size_t)sl0 * 16, (unsigned)nsafe * 64u);
  auto load_b = [&](int k, float4 (&fb)[NJ]) {
    if (!onehot) {
#pragma unroll
      for (int j = 0; j < NJ; ++j) fb[j] = pmg_bload(brs, voff, (unsigned)k * bstr + brow[j]);
    } else {
      const float4 tf = pmg_bload(trs, (unsigned)q * 16u, (unsigned)k * 64u);
      const int tk[4] = {__float_as_int(tf.x), __float_as_int(tf.y), __float_as_int(tf.z), __float_as_int(tf.w)};
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const int col = oh0 + 16 * j + n16;
        fb[j] = make_float4(tk[0] == col ? 1.f : 0.f, tk[1] == col ? 1.f : 0.f, tk[2] == col ? 1.f : 0.f,
                            tk[3] == col ? 1.f : 0.f);
      }
    }
  };
#define WG_STEP(fa, fb, C)                                                                           \
  _Pragma("unroll") for (int i = 0; i < MI; ++i) _Pragma("unroll") for (int j = 0; j < NJ; ++j)      \
      acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[i].C, fb[j].C, acc[i][j], 0, 0, 0);
#define WG_COMPUTE(fa, fb)                                                                           \
  _Pragma("unroll") for (int i = 0; i < MI; ++i) rsum[i] += (fa[i].x + fa[i].y) + (fa[i].z + fa[i].w); \
  WG_STEP(fa, fb, x) WG_STEP(fa, fb, y) WG_STEP(fa, fb, z) WG_STEP(fa, fb, w)

  if (ns > 0) {
    float4 fa0[MI], fb0[NJ], fa1[MI], fb1[NJ];
    WG_LOAD_A(fa0, 0)
    load_b(0, fb0);
#pragma unroll 1
    for (int k = 0; k < ns; k += 2) {
      const int k1 = (k + 1 < ns) ? k + 1 : k;
      WG_LOAD_A(fa1, k1)
      load_b(k1, fb1);
      __builtin_amdgcn_sched_barrier(0);
      WG_COMPUTE(fa0, fb0)
      __builtin_amdgcn_sched_barrier(0);
      const int k2 = (k + 2 < ns) ? k + 2 : k;
      WG_LOAD_A(fa0, k2)
      load_b(k2, fb0);
      __builtin_amdgcn_sched_barrier(0);
      if (k + 1 < ns) {
        WG_COMPUTE(fa1, fb1)
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
#undef WG_COMPUTE
#undef WG_STEP
#undef WG_LOAD_A

  // partial tile: [128][TN], element (row_local, col_local)
  float* pt = a.partial + (size_t)s * a.split_stride + jb.pt_off + (size_t)tl * (WG_TM * TN);
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        const int rl = wr * 16 * MI + 16 * i + 4 * q + rg;
        const int cl = wc * 16 * NJ + 16 * j + n16;
        pt[rl * TN + cl] = acc[i][j][rg];
      }
  // row sums (bias gradients): only the first column tile's wc==0 waves own them
  if (ct == 0 && wc == 0) {
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      float v2 = rsum[i];
      v2 += __shfl_xor(v2, 16);
      v2 += __shfl_xor(v2, 32);
      const int r = r0 + 16 * i + n16;
      if (q == 0 && r < jb.rows) a.rs_partial[(size_t)s * a.total_rs + jb.rs0 + r] = v2;
    }
  }
}

// tok32[p][g][r] = token fed at step p to row 16g+r (int32), -1 for rows past the batch.
__global__ void pmg_tok32_kernel(const int64_t* __restrict__ tokens, int B, int G, int N, int V, int32_t* __restrict__ out) {
  const size_t total = (size_t)N * G * 16;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
    const int r = (int)(idx & 15);
    const size_t sl = idx >> 4;
    const int p = (int)(sl / G), g = (int)(sl - (size_t)p * G);
    const int row = g * 16 + r;
    out[idx] = row < B ? (int32_t)min(max(tokens[(size_t)p * B + row], (int64_t)0), (int64_t)(V - 1)) : -1;  // clamped like the forward
  }
}

// dense[job] = sum over splits of the partial tiles; row sums likewise.
__global__ void pmg_wgrad_reduce_kernel(const WgradArgs a) {
  size_t total = 0;
  for (int k = 0; k < a.njobs; ++k) total += (size_t)a.jobs[k].rows * a.jobs[k].cols;
  const size_t total_all = total + a.total_rs;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total_all;
       idx += (size_t)gridDim.x * blockDim.x) {
    if (idx < total) {
      size_t rem = idx;
      int ji = 0;
      for (; ji < a.njobs; ++ji) {
        const size_t sz = (size_t)a.jobs[ji].rows * a.jobs[ji].cols;
        if (rem < sz) break;
        rem -= sz;
      }
      const WJob& jb = a.jobs[ji];
      const int r = (int)(rem / jb.cols), c = (int)(rem % jb.cols);
      const int tl = (r / WG_TM) * jb.col_tiles + (c / jb.tn);
      const size_t e = jb.pt_off + (size_t)tl * (WG_TM * jb.tn) + (size_t)(r % WG_TM) * jb.tn + (c % jb.tn);
      float sum = 0.0f;
      for (int s = 0; s < jb.S; ++s) sum += a.partial[(size_t)s * a.split_stride + e];
      a.dense[jb.c_off + rem] = sum;
    } else {
      const size_t r = idx - total;
      int jS = 1;
      for (int k = 0; k < a.njobs; ++k)
        if ((size_t)a.jobs[k].rs0 <= r && r < (size_t)(a.jobs[k].rs0 + a.jobs[k].rows)) jS = a.jobs[k].S;
      float sum = 0.0f;
      for (int s = 0; s < jS; ++s) sum += a.rs_partial[(size_t)s * a.total_rs + r];
      a.dense[total + r] = sum;  // row sums live right after the dense matrices
    }
  }
}

struct FinArgs {
  int H, E, W, V, L, NX, NXT, VP, G, pad_idx, accumulate;
  int ld_tab, ld_is;  // row pitch of the dGTAB block and of the dW_ih0[:, E:] block inside dense
  const float* dense;
  size_t c_hh[PMG_MAX_LAYERS], c_ih[PMG_MAX_LAYERS], c_tab, c_x;  // c_ih[0] = IS job
  size_t rs_hh[PMG_MAX_LAYERS], rs_gi[PMG_MAX_LAYERS], rs_x;       // absolute offsets in dense
  const float* emb; const float* w_ih0;
  const float* lossp;
  pmg_grads g;
  float* loss_out;
};

__device__ __forceinline__ void put(float* p, float v, int acc) { *p = acc ? (*p + v) : v; }

__global__ void pmg_finalize_kernel(const FinArgs a) {
  const int H = a.H, E = a.E, W = a.W, V = a.V, L = a.L;
  const size_t n_emb = (size_t)V * E;
  const size_t n_wih0 = (size_t)3 * H * (E + W);
  const size_t n_whh = (size_t)3 * H * H;
  const size_t n_b = (size_t)3 * H;
  const size_t n_x = (size_t)a.NX * H + a.NX;  // ctrl/push/out weights + biases
  size_t total = n_emb + n_wih0 + (size_t)(L - 1) * n_whh + (size_t)L * n_whh + (size_t)2 * L * n_b + n_x + 1;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (size_t)gridDim.x * blockDim.x) {
    size_t r = idx;
    if (r < n_emb) continue;  // embedding gradient: pmg_embgrad_kernel
    r -= n_emb;
    if (r < n_wih0) {
      const int n = (int)(r / (E + W)), e = (int)(r % (E + W));
      float s = 0.0f;
      if (e < E) {
        const float* tab = a.dense + a.c_tab + (size_t)n * a.ld_tab;
        for (int v = 0; v < V; ++v) s = fmaf(tab[v], a.emb[(size_t)v * E + e], s);
      } else {
        s = a.dense[a.c_ih[0] + (size_t)n * a.ld_is + (e - E)];
      }
      put(a.g.w_ih[0] + r, s, a.accumulate);
      continue;
    }
    r -= n_wih0;
    bool done = false;
    for (int l = 1; l < L && !done; ++l) {
      if (r < n_whh) { put(a.g.w_ih[l] + r, a.dense[a.c_ih[l] + r], a.accumulate); done = true; }
      else r -= n_whh;
    }
    if (done) continue;
    for (int l = 0; l < L && !done; ++l) {
      if (r < n_whh) { put(a.g.w_hh[l] + r, a.dense[a.c_hh[l] + r], a.accumulate); done = true; }
      else r -= n_whh;
    }
    if (done) continue;
    for (int l = 0; l < L && !done; ++l) {
      if (r < n_b) { put(a.g.b_ih[l] + r, a.dense[a.rs_gi[l] + r], a.accumulate); done = true; }
      else r -= n_b;
    }
    if (done) continue;
    for (int l = 0; l < L && !done; ++l) {
      if (r < n_b) { put(a.g.b_hh[l] + r, a.dense[a.rs_hh[l] + r], a.accumulate); done = true; }
      else r -= n_b;
    }
    if (done) continue;
    if (r < (size_t)a.NX * H) {
      const int n = (int)(r / H), k = (int)(r % H);
      const float v = a.dense[a.c_x + r];
      if (n < 3) put(a.g.w_ctrl + (size_t)n * H + k, v, a.accumulate);
      else if (n < 3 + W) put(a.g.w_push + (size_t)(n - 3) * H + k, v, a.accumulate);
      else put(a.g.w_out + (size_t)(n - 3 - W) * H + k, v, a.accumulate);
      continue;
    }
    r -= (size_t)a.NX * H;
    if (r < (size_t)a.NX) {
      const int n = (int)r;
      const float v = a.dense[a.rs_x + n];
      if (n < 3) put(a.g.b_ctrl + n, v, a.accumulate);
      else if (n < 3 + W) put(a.g.b_push + (n - 3), v, a.accumulate);
      else put(a.g.b_out + (n - 3 - W), v, a.accumulate);
      continue;
    }
    // the loss
    float s = 0.0f;
    for (int g = 0; g < a.G; ++g) s += a.lossp[g];
    put(a.loss_out, s, a.accumulate);
  }
}

// d embedding[v][e] = sum_n dGTAB[n][v] * W_ih0[n][e]  (n over the 3H gate rows).  One block per
// (token v, 64 columns e); 16 segments of the reduction per block, combined in a fixed order.
__global__ __launch_bounds__(1024) void pmg_embgrad_kernel(const FinArgs a) {
  __shared__ float part[16][64];
  const int v = blockIdx.x, e = blockIdx.y * 64 + (threadIdx.x & 63), seg = threadIdx.x >> 6;
  const int K = 3 * a.H, per = (K + 15) / 16;
  const int n0 = seg * per, n1 = (n0 + per < K) ? n0 + per : K;
  float s = 0.0f;
  if (e < a.E && v != a.pad_idx) {
    const float* tab = a.dense + a.c_tab + v;
    const float* wi = a.w_ih0 + e;
    for (int n = n0; n < n1; ++n) s = fmaf(tab[(size_t)n * a.ld_tab], wi[(size_t)n * (a.E + a.W)], s);
  }
  part[seg][threadIdx.x & 63] = s;
  __syncthreads();
  if (seg == 0 && e < a.E) {
    float t = 0.0f;
    for (int k = 0; k < 16; ++k) t += part[k][threadIdx.x];
    put(a.g.embedding + (size_t)v * a.E + e, t, a.accumulate);
  }
}

struct JobPlan {
  WgradArgs w;
  FinArgs f;
  size_t dense_floats, partial_floats, rs_partial_floats;
};

// Split-K factor of one launch with `tiles` output tiles.  512 blocks are resident at a time (two per
// CU), so the grid should be close to a whole number of such rounds: take the factor whose last
// round is fullest among those that give 1..4 rounds.
int choose_splits(const PmgDims& d, int tiles) {
  const long slabs = (long)d.T * d.G;
  if (tiles < 1) tiles = 1;
  long best = 1;
  double best_fill = -1.0;
  for (long S = 1; S <= 64; ++S) {
    const long blocks = S * tiles;
    const long rounds = (blocks + 511) / 512;
    if (rounds > 4 && S > 1) break;
    const double fill = (double)blocks / (double)(rounds * 512);
    // prefer fuller rounds; among equals the larger grid (shorter tail per block)
    if (fill > best_fill + 1e-9 || (fill > best_fill - 1e-9 && S > best)) { best_fill = fill; best = S; }
  }
  long S = best;
  const long maxS = (slabs + 3) / 4;  // at least ~4 slabs per split
  if (S > maxS) S = maxS;
  if (S < 1) S = 1;
  // one split's slab range is addressed through a 32-bit buffer resource: keep it < 2 GiB
  const long slab_bytes = (long)4 * d.H * 64;
  while (((slabs + S - 1) / S) * slab_bytes >= (1L << 31)) S *= 2;
  return (int)S;
}

void build_plan(const PmgDims& d, JobPlan* P) {
  memset(P, 0, sizeof(*P));
  WgradArgs& w = P->w;
  FinArgs& f = P->f;
  const int H = d.H, N = d.T - 1;
  int nj = 0, tile_w = 0, tile_n = 0, rs = 0;
  size_t coff = 0, ptoff = 0;
  auto add = [&](int rows, int cols, int a_rows, int skip_mid, int b_rows, int b_pshift, int Psl, int mode) -> WJob& {
    WJob& j = w.jobs[nj++];
    j.mode = mode;
    j.rows = rows; j.cols = cols; j.a_rows = a_rows; j.skip_mid = skip_mid; j.b_rows = b_rows;
    j.b_pshift = b_pshift; j.P = Psl;
    j.tn = (cols <= 64) ? 64 : 128;
    j.row_tiles = pmg_ceil_div(rows, WG_TM); j.col_tiles = pmg_ceil_div(cols, j.tn);
    const int nt = j.row_tiles * j.col_tiles;
    if (j.tn == 128) { j.tile0 = tile_w; tile_w += nt; } else { j.tile0 = tile_n; tile_n += nt; }
    j.rs0 = rs; rs += rows;
    j.c_off = coff; coff += (size_t)rows * cols;
    j.pt_off = ptoff; ptoff += (size_t)nt * WG_TM * j.tn;
    return j;
  };
  // job order is fixed; pointers are filled by the caller
  for (int l = 0; l < d.L; ++l) { WJob& j = add(3 * H, H, 4 * H, 1, H, 0, N, WG_PLAIN); f.c_hh[l] = j.c_off; f.rs_hh[l] = j.rs0; }
  // layer 0: the stack-top columns of W_ih and the per-token gate table share their A operand
  // ([dr|dz|dn]); with a vocabulary of at most 64 they are the two halves of one 128-wide tile column
  // whose B operand is the forward's [stack top | one-hot(token)] slab
  const bool mixed = d.VP <= 64;
  for (int l = 0; l < d.L; ++l) {
    // the TP slab holds [64 stack-top rows | 64 one-hot rows of the fed token] (written by the forward)
    WJob& j = (l == 0) ? (mixed ? add(3 * H, 128, 4 * H, 0, 128, 0, N, WG_PLAIN)
                                : add(3 * H, PMG_KS, 4 * H, 0, 128, 0, N, WG_PLAIN))
                       : add(3 * H, H, 4 * H, 0, H, 1, N, WG_PLAIN);
    f.c_ih[l] = j.c_off; f.rs_gi[l] = j.rs0;
  }
  if (mixed) {
    f.c_tab = f.c_ih[0] + 64; f.ld_tab = 128; f.ld_is = 128;
  } else {
    WJob& j = add(3 * H, d.VP, 4 * H, 0, 0, 0, N, WG_ONEHOT);
    f.c_tab = j.c_off; f.ld_tab = d.VP; f.ld_is = PMG_KS;
  }
  { WJob& j = add(d.NXT * 16, H, d.NXT * 16, 0, H, 0, N + 1, WG_PLAIN); f.c_x = j.c_off; f.rs_x = j.rs0; }
  w.njobs = nj; w.tiles_wide = tile_w; w.tiles_narrow = tile_n; w.total_rs = rs;
  {
    int cnt[2][2] = {{0, 0}, {0, 0}}, Sl[2][2];
    for (int j = 0; j < nj; ++j) cnt[w.jobs[j].tn == 128][w.jobs[j].mode] += w.jobs[j].row_tiles * w.jobs[j].col_tiles;
    w.S = 1;
    for (int a2 = 0; a2 < 2; ++a2)
      for (int m = 0; m < 2; ++m) {
        Sl[a2][m] = cnt[a2][m] ? choose_splits(d, cnt[a2][m]) : 1;
        if (Sl[a2][m] > w.S) w.S = Sl[a2][m];
      }
    for (int j = 0; j < nj; ++j) w.jobs[j].S = Sl[w.jobs[j].tn == 128][w.jobs[j].mode];
  }
  w.split_stride = ptoff;
  w.G = d.G; w.B = d.B; w.H = H;
  // row sums are stored right behind the dense matrices
  for (int l = 0; l < d.L; ++l) { f.rs_hh[l] += coff; f.rs_gi[l] += coff; }
  f.rs_x += coff;
  P->dense_floats = coff + rs;
  P->partial_floats = (size_t)w.S * ptoff;
  P->rs_partial_floats = (size_t)w.S * rs;
}

}  // namespace

size_t pmg_wgrad_acc_floats(const PmgDims& d) {
  JobPlan P;
  build_plan(d, &P);
  return P.dense_floats;
}

size_t pmg_wgrad_partial_floats(const PmgDims& d) {
  JobPlan P;
  build_plan(d, &P);
  return ((P.partial_floats + 63) & ~(size_t)63) + ((P.rs_partial_floats + 63) & ~(size_t)63) + (size_t)d.T * d.G * 16;
}

hipError_t pmg_run_wgrad(const PmgDims& d, const PmgLayout& lay, float* ws, const pmg_weights* wts,
                         const int64_t* tokens, uint32_t flags, const pmg_grads* grads, float* loss_out,
                         hipStream_t st) {
  JobPlan P;
  build_plan(d, &P);
  WgradArgs& w = P.w;
  w.dense = ws + lay.WG;
  w.partial = ws + lay.WGP;
  w.rs_partial = ws + lay.WGP + ((P.partial_floats + 63) & ~(size_t)63);
  int32_t* tok32 = reinterpret_cast<int32_t*>(w.rs_partial + ((P.rs_partial_floats + 63) & ~(size_t)63));
  w.tok32 = tok32;
  int k = 0;
  for (int l = 0; l < d.L; ++l) { w.jobs[k].A = ws + lay.DG[l]; w.jobs[k].B = ws + lay.HH[l]; ++k; }
  for (int l = 0; l < d.L; ++l) {
    w.jobs[k].A = ws + lay.DG[l];
    w.jobs[k].B = (l == 0) ? ws + lay.TP : ws + lay.HH[l - 1];
    ++k;
  }
  if (w.jobs[k].mode == WG_ONEHOT) { w.jobs[k].A = ws + lay.DG[0]; w.jobs[k].B = ws + lay.DG[0]; ++k; }
  w.jobs[k].A = ws + lay.DX; w.jobs[k].B = ws + lay.HH[d.L - 1]; ++k;

  hipError_t e = hipSuccess;
  bool need_tok32 = false;
  for (int j = 0; j < w.njobs; ++j) need_tok32 = need_tok32 || w.jobs[j].mode == WG_ONEHOT;
  if (need_tok32) {
    const size_t total = (size_t)(d.T - 1) * d.G * 16;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(pmg_tok32_kernel, dim3(blocks), dim3(256), 0, st, tokens, d.B, d.G, d.T - 1, d.V, tok32);
    e = hipGetLastError();
    if (e != hipSuccess) return e;
  }
  {
    int nt[2][2] = {{0, 0}, {0, 0}};  // [wide][mode]
    for (int j = 0; j < w.njobs; ++j)
      nt[w.jobs[j].tn == 128][w.jobs[j].mode] += w.jobs[j].row_tiles * w.jobs[j].col_tiles;
#define PMG_WG_LAUNCH(wide_, mode_, ...)                                                     \
  if (nt[wide_][mode_] > 0) {                                                                \
    int jS = 1;                                                                              \
    for (int j = 0; j < w.njobs; ++j)                                                        \
      if ((w.jobs[j].tn == 128) == (wide_ != 0) && w.jobs[j].mode == mode_) jS = w.jobs[j].S; \
    const int nb = ((nt[wide_][mode_] * jS + 7) / 8) * 8;                                    \
    hipLaunchKernelGGL((pmg_wgrad_kernel<__VA_ARGS__>), dim3(nb), dim3(WG_THREADS), 0, st, w); \
    e = hipGetLastError();                                                                   \
    if (e != hipSuccess) return e;                                                           \
  }
    PMG_WG_LAUNCH(1, WG_PLAIN, 4, 4, 2, 2, WG_PLAIN)
    PMG_WG_LAUNCH(1, WG_ONEHOT, 4, 4, 2, 2, WG_ONEHOT)
    PMG_WG_LAUNCH(0, WG_PLAIN, 2, 4, 4, 1, WG_PLAIN)
    PMG_WG_LAUNCH(0, WG_ONEHOT, 2, 4, 4, 1, WG_ONEHOT)
#undef PMG_WG_LAUNCH
  }
  {
    size_t total = P.dense_floats;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(pmg_wgrad_reduce_kernel, dim3(blocks), dim3(256), 0, st, w);
    e = hipGetLastError();
    if (e != hipSuccess) return e;
  }
  FinArgs& f = P.f;
  f.H = d.H; f.E = d.E; f.W = d.W; f.V = d.V; f.L = d.L; f.NX = d.NX; f.NXT = d.NXT; f.VP = d.VP; f.G = d.G;
  f.pad_idx = d.pad_idx; f.accumulate = (flags & PMG_FLAG_ACCUMULATE) ? 1 : 0;
  f.dense = w.dense;
  f.emb = wts->embedding; f.w_ih0 = wts->w_ih[0];
  f.lossp = ws + lay.LOSSP;
  f.g = *grads;
  f.loss_out = loss_out;
  {
    const size_t H = d.H;
    size_t total = (size_t)d.V * d.E + 3 * H * (d.E + d.W) + (size_t)(2 * d.L - 1) * 3 * H * H +
                   (size_t)2 * d.L * 3 * H + (size_t)d.NX * H + d.NX + 1;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(pmg_finalize_kernel, dim3(blocks), dim3(256), 0, st, f);
    e = hipGetLastError();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(pmg_embgrad_kernel, dim3(d.V, (d.E + 63) / 64), dim3(1024), 0, st, f);
    e = hipGetLastError();
  }
  return e;
}
