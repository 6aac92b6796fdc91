// Argument blocks and launchers of the gfx950 kernels (internal; the public ABI is
// include/pmg.h).
#pragma once
#include "pmg_common.h"
#include <atomic>

// hipFuncAttributeMaxDynamicSharedMemorySize raised to the CU's whole 160 KiB ONCE per kernel
// instance and device (`done`: one bit per device, a static of the launcher's instantiation), not per
// launch: a launch path without the call is cheaper and safe under HIP graph capture.
static inline hipError_t pmg_full_lds_once(const void* kern, std::atomic<unsigned long long>* done) {
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  const unsigned long long bit = 1ull << (dev & 63);
  if (done->load(std::memory_order_acquire) & bit) return hipSuccess;
  e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  if (e == hipSuccess) done->fetch_or(bit, std::memory_order_release);
  return e;
}

// ------------------------------------------------------------------ packing
enum PackKind {
  PK_WX = 0,  // [ctrl | push | out] rows (natural order), K = H
  PK_WHH,     // gate rows in (unit tile, gate) order, K = H
  PK_WIS,     // layer-0 W_ih[:, E:E+W], gate rows in (unit tile, gate) order, K = 64 (zero padded)
  PK_WIH,     // layer>0 W_ih, gate rows in (unit tile, gate) order, K = H
  PK_L2H,     // latent_to_hidden, natural rows, K = Z
  PK_WXT,     // transpose of the PK_WX source: out rows = H, reduction = NXT*16 (natural)
  PK_WHHT,    // out rows = H, reduction over gate rows in backward chunk order
  PK_WIST,    // out rows = 64 (stack width, padded), reduction over gate rows in chunk order
  PK_WIHT,    // out rows = H (layer l-1 features), reduction over gate rows in chunk order
};

struct PackArgs {
  int kind;
  int tiles, KG;  // dst geometry: [tiles][KG][64][4]
  int H, E, W, V, NX, UPW, Z;
  const float* src;   // main source
  const float* src2;  // PK_WX*: w_push
  const float* src3;  // PK_WX*: w_out
  float* dst;
};

struct AuxArgs {
  int H, E, W, V, L, NX, NXT;
  const float* b_ctrl; const float* b_push; const float* b_out;
  const float* b_ih[PMG_MAX_LAYERS]; const float* b_hh[PMG_MAX_LAYERS];
  const float* emb; const float* w_ih0;
  float* BX; float* BG[PMG_MAX_LAYERS]; float* GTAB;
};

// One prepare launch per entry-point call: up to PMG_PREP_MAX packing jobs, the bias vectors and the token table;
// job j owns blocks [first[j], first[j+1]) -- packing jobs first, then the biases, then the table.
#define PMG_PREP_MAX (2 + 4 * PMG_MAX_LAYERS)
struct PrepArgs {
  int npack;
  int first[PMG_PREP_MAX + 3];
  PackArgs p[PMG_PREP_MAX];
  AuxArgs x;
};

void pmg_prep_add(PrepArgs* a, const PackArgs* p);       // a zeroed PrepArgs, then one call per packing job
hipError_t pmg_launch_prep(PrepArgs* a, hipStream_t st);  // a->x filled in by the caller

// ------------------------------------------------------------------ forward
#define PMG_FWD_DECODE 0
#define PMG_FWD_TEACHER 1
#define PMG_FWD_STEP 2

struct FwdArgs {
  int mode;   // PMG_FWD_*
  int dbg;    // ablation bits, -DPMG_ABLATE builds only (always 0 in the default library)
  int B, G, N;  // N decoder steps
  int H, W, D, V, L, Z, NX, NXT, XP, HP, KGH, VP, SW;
  int KL;     // top rows of the stack kept in LDS (pmg_fwd_stack_rows)
  int start_tok, stop_tok;
  float temperature;
  uint32_t flags;
  uint64_t seed, offset;
  int64_t seq_base;
  // packed weights: one buffer resource over [wbase, wbase + wbytes), byte offsets per matrix
  const float* wbase; unsigned wbytes;
  unsigned oWX, oWHH[PMG_MAX_LAYERS], oWIH[PMG_MAX_LAYERS], oL2H;
  const float* BX; const float* GTAB; const float* BG[PMG_MAX_LAYERS];
  const float* b_l2h;  // decode
  // inputs
  const float* latent;     // decode: [B][Z]
  const float* h0;         // teacher/step: [L][B][H]
  const int64_t* tok_in;   // teacher: [T][B]; step: [B]
  const float* stack_in;   // step: [B][D][W] or NULL
  // outputs
  int64_t* tok_out;        // decode: [B][N]
  int32_t* len_out;        // decode: [B]
  float* logp_out;         // teacher: [N][B]
  float* h_out;            // [L][B][H] or NULL
  float* stack_out;        // step: [B][D][W] or NULL
  float* logits_out;       // step: [B][V] or NULL
  // state / history (workspace)
  float* SB;               // stack slabs: decode/step 2 per group, teacher T per group ([slot][G][SW])
  float* HH[PMG_MAX_LAYERS];  // teacher: [T][G][H][16]
  float* GT[PMG_MAX_LAYERS];  // teacher: [T-1][G][UT][4][64][4]
  float* CT; float* PV; float* TP; float* PR;
  float* LP;               // teacher: [T-1][G][16] copy of logp for the backward's loss
  // clustered forward (pmg_fwdc.hip): exchange buffers, counters, and the extras in torch layout
  float* CX; unsigned* CXS;
  int cluster_gtab_lds;    // set by pmg_launch_fwdc: the token-table lines of a member's units fit its LDS
  int cluster_ng;          // set by pmg_launch_fwdc: row groups per cluster
  int cluster_duo;         // set by pmg_launch_fwdc: split ownership (stack side / vocabulary side), pmg_fwdc.hip
  int cluster_hist_t;      // set by pmg_launch_fwdc: stack / push / probability history per SEQUENCE (see LAYW)
  // History layout word (training workspaces of cluster-capable shapes, else NULL): 0 = the stack (SB), push
  // value (PV) and probability (PR) slabs interleave the 16 sequences of a row group innermost (what the
  // batch-partitioned kernels read and write); nonzero = one contiguous run per sequence, written by the
  // clustered forward when the clustered BPTT will read it.  Set by the forward launch (stream-ordered),
  // checked by the BPTT kernels: the batch-partitioned one poisons its outputs when it finds nonzero.
  unsigned* LAYW;
  const float* w_ctrl; const float* b_ctrl; const float* w_push; const float* b_push;
  const float* w_out; const float* b_out;
};

hipError_t pmg_launch_fwd(const FwdArgs* a, hipStream_t st);
// Clustered kernels: a cluster is H/16 workgroups, at most 8 * floor(32 / (H/16)) clusters are resident
// at once; more row groups than that are served NG per cluster.
#define PMG_CLUSTER_NG_MAX 8
static inline int pmg_cluster_ng(int G, int H) {
  const int ncl = 8 * (32 / (H / 16));
  return (G + ncl - 1) / ncl;
}
// Row groups per cluster up to which the clustered kernels beat the batch-partitioned ones (measured,
// profiles/r03_cluster_groups.md): the smaller the hidden size, the cheaper a batch-partitioned step and the
// larger the share of the per-sequence owner work.
static inline int pmg_cluster_ng_limit(int H) { return H >= 384 ? PMG_CLUSTER_NG_MAX : H >= 256 ? 4 : 1; }
// clustered forward for small batches (pmg_fwdc.hip); pmg_launch_fwd picks it by shape
bool pmg_fwdc_applicable(const FwdArgs* a);
hipError_t pmg_launch_fwdc(const FwdArgs* a, hipStream_t st);
size_t pmg_fwdc_cx_floats(const PmgDims& d);
size_t pmg_fwdc_sync_words(const PmgDims& d);
size_t pmg_fwd_lds_bytes(const FwdArgs* a);
int pmg_fwd_stack_rows(const FwdArgs* a);

// ------------------------------------------------------------------ backward
struct BwdArgs {
  int B, G, N;  // N = T-1 decoder steps
  int H, W, D, V, L, NX, NXT, XP, KGH, VP, SW, UPW;
  int dbg;      // ablation bits, -DPMG_ABLATE builds only
  float scale;
  const int64_t* tokens;  // [T][B]
  const float* reward;    // [B]
  // packed transposed weights: one buffer resource, byte offsets per matrix
  const float* wbase; unsigned wbytes;
  unsigned oWXt, oWHHt[PMG_MAX_LAYERS], oWIHt[PMG_MAX_LAYERS];
  // history from the forward
  const float* SB; const float* HH[PMG_MAX_LAYERS]; const float* GT[PMG_MAX_LAYERS];
  const float* CT; const float* PV; const float* PR; const float* LP;
  // outputs / scratch
  float* DG[PMG_MAX_LAYERS];  // [N][G][4H][16]   rows: dr | dz | dn | dghn
  float* DX;                  // [N+1][G][NXT*16][16]  rows: dctrl_p | dpush_p | dlogits_{p-1}
  float* DS;                  // [2][G][SW] stack-gradient ping-pong
  float* DHC;                 // [G][L][UT][64][4] hidden-gradient carry (D layout)
  float* grad_h0;             // [L][B][H]
  float* LOSSP;               // [G] loss partials
  // clustered BPTT (pmg_bwdc.hip): exchange buffers, counters, caller's flags
  float* CX; unsigned* CXS; uint32_t flags;
  int cluster_ng;             // set by pmg_launch_bwdc: row groups per cluster
  const unsigned* LAYW;       // history layout word (FwdArgs::LAYW)
};

hipError_t pmg_launch_bwd(const BwdArgs* a, hipStream_t st);
size_t pmg_bwd_lds_bytes(const BwdArgs* a);
int pmg_bwd_chunk_buffers(const BwdArgs* a);
// clustered BPTT for small batches (pmg_bwdc.hip); pmg_launch_bwd picks it by shape
bool pmg_bwdc_applicable(const BwdArgs* a);
hipError_t pmg_launch_bwdc(const BwdArgs* a, hipStream_t st);
size_t pmg_bwdc_cx_floats(const PmgDims& d);
// would the clustered BPTT serve this shape (flags aside)?  The clustered forward asks before it picks the history layout.
bool pmg_bwdc_shape_ok(int H, int L, int NX, int NXT, int V, int W, int D, int G);
