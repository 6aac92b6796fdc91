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

// Row n, column k of the virtual extras matrix [ctrl | push | out] (K = H).
__device__ __forceinline__ float extras_at(const PackArgs& a, int n, int k) {
  if (n < 3) return a.src[(size_t)n * a.H + k];
  if (n < 3 + a.W) return a.src2[(size_t)(n - 3) * a.H + k];
  if (n < a.NX) return a.src3[(size_t)(n - 3 - a.W) * a.H + k];
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

__global__ void pmg_pack_kernel(PackArgs a) {
  const size_t total = (size_t)a.tiles * a.KG * 64;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (size_t)gridDim.x * blockDim.x) {
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
        case PK_WX: x = extras_at(a, 16 * tile + i, k); break;
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
        case PK_WXT: x = extras_at(a, k, 16 * tile + i); break;
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
__global__ void pmg_aux_kernel(AuxArgs a) {
  const int H = a.H;
  const size_t nbx = (size_t)a.NXT * 16;
  const size_t nbg = (size_t)a.L * 4 * H;
  const size_t ntab = (size_t)a.V * 3 * H;
  const size_t total = nbx + nbg + ntab;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (size_t)gridDim.x * blockDim.x) {
    if (idx < nbx) {
      const int n = (int)idx;
      float x = 0.0f;
      if (n < 3) x = a.b_ctrl[n];
      else if (n < 3 + a.W) x = a.b_push[n - 3];
      else if (n < a.NX) x = a.b_out[n - 3 - a.W];
      a.BX[n] = x;
    } else if (idx < nbx + nbg) {
      const size_t r = idx - nbx;
      const int l = (int)(r / (4 * H));
      const int piece = (int)((r / H) & 3);
      const int j = (int)(r % H);
      float x;
      if (piece == 3) x = a.b_hh[l][2 * H + j];
      else if (l == 0) x = 0.0f;
      else if (piece == 2) x = a.b_ih[l][2 * H + j];
      else x = a.b_ih[l][piece * H + j] + a.b_hh[l][piece * H + j];
      a.BG[l][piece * H + j] = x;
    } else {
      const size_t r = idx - nbx - nbg;
      const int v = (int)(r / (3 * H));
      const int n = (int)(r % (3 * H));
      const float* wr = a.w_ih0 + (size_t)n * (a.E + a.W);
      const float* er = a.emb + (size_t)v * a.E;
      float s = 0.0f;
      for (int e = 0; e < a.E; ++e) s = fmaf(wr[e], er[e], s);
      s += a.b_ih[0][n];
      if (n < 2 * H) s += a.b_hh[0][n];
      a.GTAB[r] = s;
    }
  }
}

hipError_t pmg_launch_pack(const PackArgs* a, hipStream_t st) {
  const size_t total = (size_t)a->tiles * a->KG * 64;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 2048) blocks = 2048;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(pmg_pack_kernel, dim3(blocks), dim3(256), 0, st, *a);
  return hipGetLastError();
}

hipError_t pmg_launch_aux(const AuxArgs* a, hipStream_t st) {
  const size_t total = (size_t)a->NXT * 16 + (size_t)a->L * 4 * a->H + (size_t)a->V * 3 * a->H;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(pmg_aux_kernel, dim3(blocks), dim3(256), 0, st, *a);
  return hipGetLastError();
}
