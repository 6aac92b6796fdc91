// pmg_detokenize: batched token-index -> string concatenation on the device.
// Replaces the per-molecule host loop `smiles_language.token_indexes_to_smiles(mol_num.tolist())` of
// paccmann_generator/reinforce.py:172-176 for a whole decoded batch: the host then fetches one packed
// byte blob and B+1 offsets instead of B*len Python dictionary look-ups.
//
// Byte work, HBM/latency bound and tiny next to the decode: three launches, one wave per row.
//   1. row byte lengths      out_off[b+1] = sum_t |table[tokens[b,t]]|,  t < len[b]
//   2. inclusive scan        out_off[b+1] += out_off[b]   (one workgroup, in place; out_off[0] = 0)
//   3. packed write          out_bytes[out_off[b] ..) = concat_t table[tokens[b,t]]
// A token outside [0,V) or without a table entry contributes nothing, like `index_to_token.get(i, '')`.
#include "pmg_common.h"

namespace {

constexpr int DT_WAVES = 8;
constexpr int DT_THREADS = DT_WAVES * 64;
constexpr int DT_SCAN = 1024;

__device__ __forceinline__ int tok_bytes(const int64_t tok, const int32_t* __restrict__ tab_off, int V, int max_tok,
                                         int* start) {
  if (tok < 0 || tok >= (int64_t)V) {
    *start = 0;
    return 0;
  }
  const int a = tab_off[tok], b = tab_off[tok + 1];
  *start = a;
  const int n = b - a;
  return n < 0 ? 0 : (n > max_tok ? max_tok : n);  // clamp: a malformed table cannot overrun out_bytes
}

__device__ __forceinline__ int row_len(const int32_t* __restrict__ len, int b, int max_len) {
  const int n = len[b];
  return n < 0 ? 0 : (n > max_len ? max_len : n);
}

__global__ __launch_bounds__(DT_THREADS) void detok_rowlen_kernel(const int64_t* __restrict__ tokens, int64_t ld,
                                                                   const int32_t* __restrict__ len, int B, int max_len,
                                                                   const int32_t* __restrict__ tab_off, int V,
                                                                   int max_tok, int64_t* __restrict__ out_off) {
  const int lane = threadIdx.x & 63;
  const int b = blockIdx.x * DT_WAVES + (threadIdx.x >> 6);
  if (blockIdx.x == 0 && threadIdx.x == 0) out_off[0] = 0;
  if (b >= B) return;
  const int n = row_len(len, b, max_len);
  int sum = 0;
  for (int t = lane; t < n; t += 64) {
    int st;
    sum += tok_bytes(tokens[(int64_t)b * ld + t], tab_off, V, max_tok, &st);
  }
  for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
  if (lane == 0) out_off[b + 1] = sum;
}

// In-place inclusive scan of out_off[1..B] by one workgroup, DT_SCAN elements per pass.
__global__ __launch_bounds__(DT_SCAN) void detok_scan_kernel(int64_t* __restrict__ out_off, int B) {
  __shared__ int64_t wsum[DT_SCAN / 64];
  __shared__ int64_t carry_s;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (threadIdx.x == 0) carry_s = 0;
  __syncthreads();
  for (int base = 0; base < B; base += DT_SCAN) {
    const int i = base + threadIdx.x;
    int64_t v = i < B ? out_off[i + 1] : 0;
    for (int o = 1; o < 64; o <<= 1) {
      const int64_t u = __shfl_up(v, o, 64);
      if (lane >= o) v += u;
    }
    if (lane == 63) wsum[wave] = v;
    __syncthreads();
    int64_t pre = carry_s;
    for (int w = 0; w < wave; ++w) pre += wsum[w];
    if (i < B) out_off[i + 1] = v + pre;
    __syncthreads();
    if (threadIdx.x == DT_SCAN - 1) carry_s = v + pre;
    __syncthreads();
  }
}

__global__ __launch_bounds__(DT_THREADS) void detok_write_kernel(const int64_t* __restrict__ tokens, int64_t ld,
                                                                  const int32_t* __restrict__ len, int B, int max_len,
                                                                  const int32_t* __restrict__ tab_off,
                                                                  const uint8_t* __restrict__ tab_bytes, int V,
                                                                  int max_tok, const int64_t* __restrict__ out_off,
                                                                  uint8_t* __restrict__ out_bytes) {
  const int lane = threadIdx.x & 63;
  const int b = blockIdx.x * DT_WAVES + (threadIdx.x >> 6);
  if (b >= B) return;
  const int n = row_len(len, b, max_len);
  int64_t run = out_off[b];
  for (int t0 = 0; t0 < n; t0 += 64) {  // wave-uniform trip count
    const int t = t0 + lane;
    int st = 0, nb = 0;
    if (t < n) nb = tok_bytes(tokens[(int64_t)b * ld + t], tab_off, V, max_tok, &st);
    int inc = nb;
    for (int o = 1; o < 64; o <<= 1) {
      const int u = __shfl_up(inc, o, 64);
      if (lane >= o) inc += u;
    }
    const int64_t pos = run + (inc - nb);
    for (int j = 0; j < nb; ++j) out_bytes[pos + j] = tab_bytes[st + j];
    run += __shfl(inc, 63, 64);
  }
}

}  // namespace

extern "C" int pmg_detokenize(const int64_t* tokens, int64_t ld_tokens, const int32_t* len, int B, int max_len,
                              const int32_t* tab_off, const uint8_t* tab_bytes, int V, int max_tok_bytes,
                              int64_t* out_off, uint8_t* out_bytes, size_t out_cap, pmg_stream stream) {
  if (!tokens || !len || !tab_off || !tab_bytes || !out_off || !out_bytes) return PMG_E_NULL;
  if (B < 1 || max_len < 1 || V < 1 || max_tok_bytes < 1 || ld_tokens < max_len) return PMG_E_ARG;
  if (out_cap / (size_t)max_tok_bytes / (size_t)max_len < (size_t)B) return PMG_E_ARG;  // worst case must fit
  hipStream_t st = (hipStream_t)stream;
  const int grid = (B + DT_WAVES - 1) / DT_WAVES;
  hipLaunchKernelGGL(detok_rowlen_kernel, dim3(grid), dim3(DT_THREADS), 0, st, tokens, ld_tokens, len, B, max_len,
                     tab_off, V, max_tok_bytes, out_off);
  hipLaunchKernelGGL(detok_scan_kernel, dim3(1), dim3(DT_SCAN), 0, st, out_off, B);
  hipLaunchKernelGGL(detok_write_kernel, dim3(grid), dim3(DT_THREADS), 0, st, tokens, ld_tokens, len, B, max_len,
                     tab_off, tab_bytes, V, max_tok_bytes, out_off, out_bytes);
  return (int)hipGetLastError();
}
