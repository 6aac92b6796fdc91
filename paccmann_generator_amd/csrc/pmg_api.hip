// Host side of the C ABI (include/pmg.h): shape checks, workspace layout, weight
// packing and kernel sequencing.  Allocates nothing, never synchronises.
#include "pmg_kernels.h"
#include <string.h>
#include <stdlib.h>
#include <mutex>

// Ablation bits for timing experiments (skip a phase of the kernels; results are wrong when set)
// exist only in -DPMG_ABLATE builds (tuning variants made by __graft_entry__.build(extra_flags=...,
// libname=...)).  The default library reads no environment variable and its kernels carry no
// ablation branches (PMG_DBG() is the constant 0 there).
#ifdef PMG_ABLATE
static int debug_bits() {
  const char* e = getenv("PMG_DEBUG");
  return e ? (int)strtol(e, nullptr, 16) : 0;
}
#else
static int debug_bits() { return 0; }
#endif

// implemented in pmg_wgrad.hip
struct WgradArgs;
hipError_t pmg_run_wgrad(const PmgDims& d, const PmgLayout& lay, float* ws, const pmg_weights* w,
                         const int64_t* tokens, uint32_t flags, const pmg_grads* grads, float* loss_out,
                         hipStream_t st);
size_t pmg_wgrad_acc_floats(const PmgDims& d);
size_t pmg_wgrad_partial_floats(const PmgDims& d);

static size_t align64(size_t x) { return (x + 63) & ~(size_t)63; }

// shapes the clustered forward may serve (pmg_fwdc_applicable decides per call): room for its
// exchange buffers is laid out only then
static bool cluster_shape(const PmgDims& d) { return d.L == 1 && d.H <= 512 && pmg_cluster_ng(d.G, d.H) <= pmg_cluster_ng_limit(d.H); }

static void make_layout(const PmgDims& d, int mode, PmgLayout* L) {
  memset(L, 0, sizeof(*L));
  size_t off = 0;
  auto take = [&](size_t n) { size_t o = off; off = align64(off + n); return o; };
  const size_t H = d.H, G = d.G, T = d.T;
  L->WXp = take((size_t)d.NXT * 16 * H);
  L->BX = take((size_t)d.NXT * 16);
  L->GTAB = take((size_t)d.V * 3 * H);
  for (int l = 0; l < d.L; ++l) {
    L->WHHp[l] = take(3 * H * H);
    L->WIHp[l] = take(l == 0 ? 3 * H * PMG_KS : 3 * H * H);
    L->BG[l] = take(4 * H);
  }
  if (mode == PMG_MODE_DECODE) {
    L->H0 = take((size_t)H * d.Z);  // packed latent_to_hidden
    L->pack_end = off;
    if (cluster_shape(d)) { L->CX = take(pmg_fwdc_cx_floats(d)); L->CXS = take(pmg_fwdc_sync_words(d)); }
    L->SB = take(2 * G * (size_t)d.SW);
  } else {
    L->WXt = take((size_t)d.NXT * 16 * H);
    for (int l = 0; l < d.L; ++l) {
      L->WHHt[l] = take(3 * H * H);
      L->WIHt[l] = take(l == 0 ? (size_t)PMG_KS * 3 * H : 3 * H * H);
    }
    L->pack_end = off;
    if (cluster_shape(d)) {   // one exchange region serves the forward, then the backward
      const size_t f = pmg_fwdc_cx_floats(d), b = pmg_bwdc_cx_floats(d);
      L->CX = take(f > b ? f : b);
      L->CXS = take(pmg_fwdc_sync_words(d));
      L->LAYW = take(64);
    }
    L->SB = take(T * G * (size_t)d.SW);
    for (int l = 0; l < d.L; ++l) {
      L->HH[l] = take(T * G * H * 16);
      L->GT[l] = take((T - 1) * G * 4 * H * 16);
      L->DG[l] = take((T - 1) * G * 4 * H * 16);
    }
    L->CT = take((T - 1) * G * 64);
    L->PV = take((T - 1) * G * 1024);
    L->TP = take((T - 1) * G * 2048);  // [64 stack-top rows | 64 one-hot token rows][16]
    L->PR = take((T - 1) * G * (size_t)d.VP * 16);
    L->DX = take(T * G * (size_t)d.NXT * 16 * 16);
    L->DS = take(2 * G * (size_t)d.SW);
    L->LOSSP = take(G);
    L->LP = take((T - 1) * G * 16);
    L->DHC = take(G * (size_t)d.L * H * 16);
    L->WG = take(pmg_wgrad_acc_floats(d));
    L->WGP = take(pmg_wgrad_partial_floats(d));
  }
  L->total = off;
}

extern "C" int pmg_abi_version(void) { return PMG_ABI_VERSION; }

extern "C" const char* pmg_strerror(int code) {
  switch (code) {
    case PMG_OK: return "ok";
    case PMG_E_NULL: return "null pointer argument";
    case PMG_E_SHAPE: return "bad shape: need B>=1, T>=1 (train: T>=2), E>=1, D>=2, Z%32==0";
    case PMG_E_HIDDEN: return "unsupported rnn_cell_size: H must be a multiple of 128 between 128 and 1024";
    case PMG_E_STACK: return "unsupported stack: need 1<=W<=64, 2<=D";
    case PMG_E_VOCAB: return "unsupported vocabulary: need 2<=V<=1024";
    case PMG_E_LAYERS: return "unsupported n_layers: need 1<=L<=4";
    case PMG_E_ALIGN: return "pointer not 16-byte aligned (b_l2h of pmg_decode, grad_h0 of pmg_reinforce_bwd)";
    case PMG_E_MODE: return "unknown workspace mode";
    case PMG_E_ARG: return "bad argument";
    case PMG_E_LDS: return "shape needs more than 160 KiB of LDS per workgroup";
    case PMG_E_SCRATCH: return "shape needs more scratch than the 32 MiB static array of pmg_decode_noalloc holds";
  }
  return code > 0 ? "HIP runtime error (see hipGetErrorString)" : "unknown error";
}

extern "C" int pmg_check_shape(const pmg_shape* s, int mode) {
  if (!s) return PMG_E_NULL;
  if (mode != PMG_MODE_DECODE && mode != PMG_MODE_TRAIN) return PMG_E_MODE;
  if (s->H < 128 || s->H > 1024 || s->H % 128 != 0) return PMG_E_HIDDEN;
  if (s->L < 1 || s->L > PMG_MAX_LAYERS) return PMG_E_LAYERS;
  if (s->W < 1 || s->W > PMG_KS || s->D < 2) return PMG_E_STACK;
  if (s->V < 2 || s->V > PMG_MAX_VOCAB) return PMG_E_VOCAB;
  if (s->B < 1 || s->T < 1 || s->E < 1) return PMG_E_SHAPE;
  if (mode == PMG_MODE_TRAIN && s->T < 2) return PMG_E_SHAPE;
  if (mode == PMG_MODE_DECODE && (s->Z < 32 || s->Z % 32 != 0)) return PMG_E_SHAPE;
  const PmgDims d = pmg_dims(s);
  // pmg_fwd.hip: hidden state of every layer + extras tiles; the decode's latent rows alias what is behind the hidden state
  size_t lds = ((size_t)d.L * 16 * d.HP + 16 * PMG_XTP + (size_t)16 * d.XP + 32) * 4;
  if (mode == PMG_MODE_DECODE) {
    const size_t init = ((size_t)d.L * 16 * d.HP + (size_t)16 * (d.Z + 8)) * 4;
    if (init > lds) lds = init;
  }
  if (lds > 160 * 1024) return PMG_E_LDS;
  if (mode == PMG_MODE_TRAIN) {   // the BPTT needs at least one chunk buffer next to its extras tile
    BwdArgs b;
    memset(&b, 0, sizeof(b));
    b.H = d.H; b.XP = d.XP;
    if (pmg_bwd_chunk_buffers(&b) < 1) return PMG_E_LDS;
  }
  return PMG_OK;
}

extern "C" int pmg_workspace_bytes(const pmg_shape* s, int mode, size_t* out) {
  if (!out) return PMG_E_NULL;
  const int rc = pmg_check_shape(s, mode);
  if (rc) return rc;
  PmgLayout lay;
  make_layout(pmg_dims(s), mode, &lay);
  *out = lay.total * sizeof(float) + 256;
  return PMG_OK;
}

static float* ws_base(void* ws) {
  return reinterpret_cast<float*>(((uintptr_t)ws + 255) & ~(uintptr_t)255);
}

static bool misaligned16(const void* p) { return ((uintptr_t)p & 15u) != 0; }

#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return (int)e_; } while (0)

// Pack the forward operands (and, for training, the transposed ones).
static int pack_weights(const PmgDims& d, const PmgLayout& lay, float* base, const pmg_weights* w, int mode,
                        bool with_l2h, hipStream_t st) {
  PrepArgs prep;
  memset(&prep, 0, sizeof(prep));
  PackArgs p;
  memset(&p, 0, sizeof(p));
  p.H = d.H; p.E = d.E; p.W = d.W; p.V = d.V; p.NX = d.NX; p.UPW = d.UPW; p.Z = d.Z;
  // extras
  p.kind = PK_WX; p.tiles = d.NXT; p.KG = d.KGH; p.src = w->w_ctrl; p.src2 = w->w_push; p.src3 = w->w_out;
  p.dst = base + lay.WXp;
  pmg_prep_add(&prep, &p);
  for (int l = 0; l < d.L; ++l) {
    p.kind = PK_WHH; p.tiles = 3 * d.UT; p.KG = d.KGH; p.src = w->w_hh[l]; p.dst = base + lay.WHHp[l];
    pmg_prep_add(&prep, &p);
    if (l == 0) { p.kind = PK_WIS; p.KG = PMG_KS / 16; } else { p.kind = PK_WIH; p.KG = d.KGH; }
    p.src = w->w_ih[l]; p.dst = base + lay.WIHp[l];
    pmg_prep_add(&prep, &p);
  }
  if (mode == PMG_MODE_DECODE) {
    if (with_l2h) {
    p.kind = PK_L2H; p.tiles = d.UT; p.KG = d.Z / 16; p.src = w->w_l2h; p.dst = base + lay.H0;
    pmg_prep_add(&prep, &p);
    }
  } else {
    p.kind = PK_WXT; p.tiles = d.UT; p.KG = d.NXT; p.src = w->w_ctrl; p.src2 = w->w_push; p.src3 = w->w_out;
    p.dst = base + lay.WXt;
    pmg_prep_add(&prep, &p);
    for (int l = 0; l < d.L; ++l) {
      p.kind = PK_WHHT; p.tiles = d.UT; p.KG = 3 * d.H / 16; p.src = w->w_hh[l]; p.dst = base + lay.WHHt[l];
      pmg_prep_add(&prep, &p);
      if (l == 0) { p.kind = PK_WIST; p.tiles = PMG_KS / 16; } else { p.kind = PK_WIHT; p.tiles = d.UT; }
      p.src = w->w_ih[l]; p.dst = base + lay.WIHt[l];
      pmg_prep_add(&prep, &p);
    }
  }
  AuxArgs& x = prep.x;
  x.H = d.H; x.E = d.E; x.W = d.W; x.V = d.V; x.L = d.L; x.NX = d.NX; x.NXT = d.NXT;
  x.b_ctrl = w->b_ctrl; x.b_push = w->b_push; x.b_out = w->b_out;
  for (int l = 0; l < d.L; ++l) { x.b_ih[l] = w->b_ih[l]; x.b_hh[l] = w->b_hh[l]; x.BG[l] = base + lay.BG[l]; }
  x.emb = w->embedding; x.w_ih0 = w->w_ih[0];
  x.BX = base + lay.BX; x.GTAB = base + lay.GTAB;
  HIPCHK(pmg_launch_prep(&prep, st));
  return PMG_OK;
}

static int check_weights(const pmg_shape* s, const pmg_weights* w, bool need_l2h) {
  if (!w || !w->embedding || !w->w_ctrl || !w->b_ctrl || !w->w_push || !w->b_push || !w->w_out || !w->b_out)
    return PMG_E_NULL;
  for (int l = 0; l < s->L; ++l)
    if (!w->w_ih[l] || !w->w_hh[l] || !w->b_ih[l] || !w->b_hh[l]) return PMG_E_NULL;
  if (need_l2h && (!w->w_l2h || !w->b_l2h)) return PMG_E_NULL;
  return PMG_OK;
}

static void fill_fwd_common(FwdArgs* a, const PmgDims& d, const PmgLayout& lay, float* base) {
  memset(a, 0, sizeof(*a));
  a->dbg = debug_bits();
  a->B = d.B; a->G = d.G; a->H = d.H; a->W = d.W; a->D = d.D; a->V = d.V; a->L = d.L; a->Z = d.Z;
  a->NX = d.NX; a->NXT = d.NXT; a->XP = d.XP; a->HP = d.HP; a->KGH = d.KGH; a->VP = d.VP; a->SW = d.SW;
  a->wbase = base;
  a->wbytes = (unsigned)(lay.pack_end * sizeof(float));
  a->oWX = (unsigned)(lay.WXp * 4);
  a->BX = base + lay.BX;
  a->GTAB = base + lay.GTAB;
  for (int l = 0; l < d.L; ++l) {
    a->oWHH[l] = (unsigned)(lay.WHHp[l] * 4);
    a->oWIH[l] = (unsigned)(lay.WIHp[l] * 4);
    a->BG[l] = base + lay.BG[l];
  }
  a->SB = base + lay.SB;
  if (lay.CXS) {
    a->CX = base + lay.CX;
    a->CXS = reinterpret_cast<unsigned*>(base + lay.CXS);
  }
  if (lay.LAYW) a->LAYW = reinterpret_cast<unsigned*>(base + lay.LAYW);
}

extern "C" int pmg_uses_cluster(const pmg_shape* s, int mode, uint32_t flags) {
  const int rc = pmg_check_shape(s, mode);
  if (rc) return rc;
  const PmgDims d = pmg_dims(s);
  PmgLayout lay;
  make_layout(d, mode, &lay);
  FwdArgs a;
  fill_fwd_common(&a, d, lay, reinterpret_cast<float*>(256));   // addresses are only compared with null
  a.mode = (mode == PMG_MODE_DECODE) ? PMG_FWD_DECODE : PMG_FWD_TEACHER;
  a.flags = flags;
  int r = pmg_fwdc_applicable(&a) ? 1 : 0;
  if (mode == PMG_MODE_TRAIN) {
    BwdArgs b;
    memset(&b, 0, sizeof(b));
    b.B = d.B; b.G = d.G; b.N = d.T - 1; b.H = d.H; b.W = d.W; b.D = d.D; b.V = d.V; b.L = d.L;
    b.NX = d.NX; b.NXT = d.NXT; b.flags = flags;
    b.CX = lay.CXS ? reinterpret_cast<float*>(256) : nullptr;
    if (pmg_bwdc_applicable(&b)) r |= 2;
  }
  return r;
}

static void fill_extras(FwdArgs* a, const pmg_weights* w) {
  a->w_ctrl = w->w_ctrl; a->b_ctrl = w->b_ctrl; a->w_push = w->w_push; a->b_push = w->b_push;
  a->w_out = w->w_out; a->b_out = w->b_out;
}

extern "C" int pmg_decode(const pmg_shape* s, const pmg_weights* w, const float* latent, int start_tok,
                          int stop_tok, int gen_len, float temperature, uint64_t philox_seed,
                          uint64_t philox_offset, int64_t seq_index_base, uint32_t flags, int64_t* tokens_out,
                          int32_t* len_out, void* ws, pmg_stream stream) {
  int rc = pmg_check_shape(s, PMG_MODE_DECODE);
  if (rc) return rc;
  if (!latent || !tokens_out || !len_out || !ws) return PMG_E_NULL;
  if ((rc = check_weights(s, w, true))) return rc;
  if (gen_len < 1 || gen_len != s->T) return PMG_E_ARG;
  if (start_tok < 0 || start_tok >= s->V || stop_tok < 0 || stop_tok >= s->V) return PMG_E_ARG;
  if (misaligned16(w->b_l2h)) return PMG_E_ALIGN;  // read with 16-byte loads
  hipStream_t st = (hipStream_t)stream;
  const PmgDims d = pmg_dims(s);
  PmgLayout lay;
  make_layout(d, PMG_MODE_DECODE, &lay);
  float* base = ws_base(ws);
  if ((rc = pack_weights(d, lay, base, w, PMG_MODE_DECODE, true, st))) return rc;
  FwdArgs a;
  fill_fwd_common(&a, d, lay, base);
  a.mode = PMG_FWD_DECODE;
  fill_extras(&a, w);
  a.KL = pmg_fwd_stack_rows(&a);
  a.N = gen_len;
  a.start_tok = start_tok; a.stop_tok = stop_tok; a.temperature = temperature; a.flags = flags;
  a.seed = philox_seed; a.offset = philox_offset; a.seq_base = seq_index_base;
  a.oL2H = (unsigned)(lay.H0 * 4);
  a.b_l2h = w->b_l2h;
  a.latent = latent;
  a.tok_out = tokens_out; a.len_out = len_out;
  HIPCHK(pmg_launch_fwd(&a, st));
  return PMG_OK;
}

// ---------------------------------------------------------------------------------
// pmg_decode_noalloc: pmg_decode on a workspace that is a static device array, for callers that
// cannot allocate (HIP graph capture, tight serving loops, harnesses that forbid hipMalloc).
// Every GPU the library runs on holds its own copy of the array at its own address; the address is
// looked up once per device, with that device current, and only that pointer is ever handed to
// kernels (the array's name is a host-side stand-in, not a device address).
// ---------------------------------------------------------------------------------
#define PMG_NOALLOC_SCRATCH_BYTES ((size_t)32 << 20)
#define PMG_NOALLOC_MAX_DEVICES 64
static __device__ __align__(256) unsigned char scratch[PMG_NOALLOC_SCRATCH_BYTES];
static std::mutex g_scratch_mu;
static void* g_scratch_ptr[PMG_NOALLOC_MAX_DEVICES];  // per device, filled on the first call on that device

static int scratch_for_current_device(void** out) {
  int dev = -1;
  HIPCHK(hipGetDevice(&dev));
  if (dev < 0 || dev >= PMG_NOALLOC_MAX_DEVICES) return PMG_E_ARG;
  std::lock_guard<std::mutex> lock(g_scratch_mu);
  if (!g_scratch_ptr[dev]) {
    void* p = nullptr;
    HIPCHK(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
    if (!p) return PMG_E_NULL;
    g_scratch_ptr[dev] = p;
  }
  *out = g_scratch_ptr[dev];
  return PMG_OK;
}

extern "C" size_t pmg_noalloc_scratch_bytes(void) { return PMG_NOALLOC_SCRATCH_BYTES; }

extern "C" int pmg_decode_noalloc(const pmg_shape* s, const pmg_weights* w, const float* latent, int start_tok,
                                  int stop_tok, int gen_len, float temperature, uint64_t philox_seed,
                                  uint64_t philox_offset, int64_t seq_index_base, uint32_t flags,
                                  int64_t* tokens_out, int32_t* len_out, pmg_stream stream) {
  size_t need = 0;
  int rc = pmg_workspace_bytes(s, PMG_MODE_DECODE, &need);
  if (rc) return rc;
  if (need > PMG_NOALLOC_SCRATCH_BYTES) return PMG_E_SCRATCH;
  void* ws = nullptr;
  if ((rc = scratch_for_current_device(&ws))) return rc;
  // Nothing in the decode workspace has to start at zero: the packed weights are rewritten by every
  // call and stack rows are only read after this call wrote them (rows never written are masked).
  return pmg_decode(s, w, latent, start_tok, stop_tok, gen_len, temperature, philox_seed, philox_offset,
                    seq_index_base, flags, tokens_out, len_out, ws, stream);
}

static void fill_history(FwdArgs* a, const PmgDims& d, const PmgLayout& lay, float* base) {
  for (int l = 0; l < d.L; ++l) { a->HH[l] = base + lay.HH[l]; a->GT[l] = base + lay.GT[l]; }
  a->CT = base + lay.CT; a->PV = base + lay.PV; a->TP = base + lay.TP; a->PR = base + lay.PR;
  a->LP = base + lay.LP;
}

extern "C" int pmg_logprob_fwd(const pmg_shape* s, const pmg_weights* w, const int64_t* tokens, const float* h0,
                               float* logp_out, float* h_out, uint32_t flags, void* ws, pmg_stream stream) {
  int rc = pmg_check_shape(s, PMG_MODE_TRAIN);
  if (rc) return rc;
  if (!tokens || !h0 || !logp_out || !ws) return PMG_E_NULL;
  if ((rc = check_weights(s, w, false))) return rc;
  hipStream_t st = (hipStream_t)stream;
  const PmgDims d = pmg_dims(s);
  PmgLayout lay;
  make_layout(d, PMG_MODE_TRAIN, &lay);
  float* base = ws_base(ws);
  // the packed-weight region sits at the start of the workspace and its layout depends only on the
  // model dimensions, so a caller that runs a batch in chunks on one workspace packs once
  if (!(flags & PMG_FLAG_REUSE_PACK))
    if ((rc = pack_weights(d, lay, base, w, PMG_MODE_TRAIN, false, st))) return rc;
  FwdArgs a;
  fill_fwd_common(&a, d, lay, base);
  fill_history(&a, d, lay, base);
  a.mode = PMG_FWD_TEACHER;
  a.flags = flags;
  fill_extras(&a, w);
  a.KL = pmg_fwd_stack_rows(&a);
  a.N = d.T - 1;
  a.h0 = h0; a.tok_in = tokens; a.logp_out = logp_out; a.h_out = h_out;
  HIPCHK(pmg_launch_fwd(&a, st));
  return PMG_OK;
}

extern "C" int pmg_step(const pmg_shape* s, const pmg_weights* w, const int64_t* tok, const float* hidden_in,
                        const float* stack_in, float* hidden_out, float* stack_out, float* logits_out, void* ws,
                        pmg_stream stream) {
  if (!s) return PMG_E_NULL;
  pmg_shape s1 = *s;
  s1.T = 1;
  if (s1.Z < 32 || s1.Z % 32) s1.Z = 32;  // latent_to_hidden is not used by a step
  int rc = pmg_check_shape(&s1, PMG_MODE_DECODE);
  if (rc) return rc;
  if (!tok || !hidden_in || !hidden_out || !ws) return PMG_E_NULL;
  if ((rc = check_weights(&s1, w, false))) return rc;
  hipStream_t st = (hipStream_t)stream;
  const PmgDims d = pmg_dims(&s1);
  PmgLayout lay;
  make_layout(d, PMG_MODE_DECODE, &lay);
  float* base = ws_base(ws);
  if ((rc = pack_weights(d, lay, base, w, PMG_MODE_DECODE, false, st))) return rc;
  FwdArgs a;
  fill_fwd_common(&a, d, lay, base);
  a.mode = PMG_FWD_STEP;
  a.KL = 0;
  a.N = 1;
  a.h0 = hidden_in; a.tok_in = tok; a.stack_in = stack_in;
  a.h_out = hidden_out; a.stack_out = stack_out; a.logits_out = logits_out;
  HIPCHK(pmg_launch_fwd(&a, st));
  return PMG_OK;
}

// shared argument checks + kernel arguments of the two halves of the backward pass
static int bwd_setup(const pmg_shape* s, const pmg_weights* w, const int64_t* tokens, void* ws, PmgDims* d,
                     PmgLayout* lay, float** base) {
  int rc = pmg_check_shape(s, PMG_MODE_TRAIN);
  if (rc) return rc;
  if (!tokens || !ws) return PMG_E_NULL;
  if ((rc = check_weights(s, w, false))) return rc;
  *d = pmg_dims(s);
  make_layout(*d, PMG_MODE_TRAIN, lay);
  *base = ws_base(ws);
  return PMG_OK;
}

extern "C" int pmg_reinforce_bptt(const pmg_shape* s, const pmg_weights* w, const int64_t* tokens,
                                  const float* reward, float scale, uint32_t flags, float* grad_h0, void* ws,
                                  pmg_stream stream) {
  PmgDims d;
  PmgLayout lay;
  float* base = nullptr;
  int rc = bwd_setup(s, w, tokens, ws, &d, &lay, &base);
  if (rc) return rc;
  if (!reward || !grad_h0) return PMG_E_NULL;
  if (misaligned16(grad_h0)) return PMG_E_ALIGN;  // written with 16-byte stores
  hipStream_t st = (hipStream_t)stream;
  BwdArgs a;
  memset(&a, 0, sizeof(a));
  a.B = d.B; a.G = d.G; a.N = d.T - 1; a.H = d.H; a.W = d.W; a.D = d.D; a.V = d.V; a.L = d.L;
  a.NX = d.NX; a.NXT = d.NXT; a.XP = d.XP; a.KGH = d.KGH; a.VP = d.VP; a.SW = d.SW; a.UPW = d.UPW;
  a.scale = scale; a.tokens = tokens; a.reward = reward; a.dbg = debug_bits();
  a.wbase = base;
  a.wbytes = (unsigned)(lay.pack_end * sizeof(float));
  a.oWXt = (unsigned)(lay.WXt * 4);
  for (int l = 0; l < d.L; ++l) {
    a.oWHHt[l] = (unsigned)(lay.WHHt[l] * 4);
    a.oWIHt[l] = (unsigned)(lay.WIHt[l] * 4);
    a.HH[l] = base + lay.HH[l]; a.GT[l] = base + lay.GT[l]; a.DG[l] = base + lay.DG[l];
  }
  a.SB = base + lay.SB; a.CT = base + lay.CT; a.PV = base + lay.PV; a.PR = base + lay.PR;
  a.LP = base + lay.LP;
  a.DX = base + lay.DX; a.DS = base + lay.DS; a.DHC = base + lay.DHC; a.grad_h0 = grad_h0;
  a.LOSSP = base + lay.LOSSP;
  a.flags = flags;
  if (lay.CXS) {
    a.CX = base + lay.CX;
    a.CXS = reinterpret_cast<unsigned*>(base + lay.CXS);
  }
  if (lay.LAYW) a.LAYW = reinterpret_cast<const unsigned*>(base + lay.LAYW);
  HIPCHK(pmg_launch_bwd(&a, st));
  return PMG_OK;
}

extern "C" int pmg_reinforce_wgrad(const pmg_shape* s, const pmg_weights* w, const int64_t* tokens, uint32_t flags,
                                   const pmg_grads* grad_out, float* loss_out, void* ws, pmg_stream stream) {
  PmgDims d;
  PmgLayout lay;
  float* base = nullptr;
  int rc = bwd_setup(s, w, tokens, ws, &d, &lay, &base);
  if (rc) return rc;
  if (!grad_out || !loss_out) return PMG_E_NULL;
  if (!grad_out->embedding || !grad_out->w_ctrl || !grad_out->b_ctrl || !grad_out->w_push || !grad_out->b_push ||
      !grad_out->w_out || !grad_out->b_out)
    return PMG_E_NULL;
  for (int l = 0; l < s->L; ++l)
    if (!grad_out->w_ih[l] || !grad_out->w_hh[l] || !grad_out->b_ih[l] || !grad_out->b_hh[l]) return PMG_E_NULL;
  HIPCHK(pmg_run_wgrad(d, lay, base, w, tokens, flags, grad_out, loss_out, (hipStream_t)stream));
  return PMG_OK;
}

extern "C" int pmg_reinforce_bwd(const pmg_shape* s, const pmg_weights* w, const int64_t* tokens,
                                 const float* reward, float scale, uint32_t flags, const pmg_grads* grad_out,
                                 float* grad_h0, float* loss_out, void* ws, pmg_stream stream) {
  // argument errors of either half are reported before anything is launched
  if (!grad_out || !loss_out || !reward || !grad_h0) return PMG_E_NULL;
  if (!grad_out->embedding || !grad_out->w_ctrl || !grad_out->b_ctrl || !grad_out->w_push || !grad_out->b_push ||
      !grad_out->w_out || !grad_out->b_out)
    return PMG_E_NULL;
  if (s)
    for (int l = 0; l < s->L && l < PMG_MAX_LAYERS; ++l)
      if (!grad_out->w_ih[l] || !grad_out->w_hh[l] || !grad_out->b_ih[l] || !grad_out->b_hh[l]) return PMG_E_NULL;
  int rc = pmg_reinforce_bptt(s, w, tokens, reward, scale, flags, grad_h0, ws, stream);
  if (rc) return rc;
  return pmg_reinforce_wgrad(s, w, tokens, flags, grad_out, loss_out, ws, stream);
}
