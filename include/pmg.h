/* pmg.h -- C ABI of the MI355X-native REINFORCE hot path of paccmann_generator.
 *
 * The reference has no FFI: its hot path is a Python loop over a duck-typed decoder
 * protocol.  Each entry point below replaces one piece of that loop; the citation
 * names the reference lines (relative to /root/reference/) whose arithmetic it
 * performs.  Conventions:
 *   - the caller owns every buffer; all pointers are DEVICE pointers unless noted;
 *   - the library allocates nothing: scratch comes in through `ws`, sized by
 *     pmg_workspace_bytes();
 *   - every call is stream-ordered on the caller's hipStream_t and never synchronises;
 *   - return value: 0 ok, <0 argument/shape error (pmg_strerror), >0 a hipError_t;
 *   - weights are in torch layouts (nn.GRU weight_ih_l{k} [3H,in], weight_hh_l{k}
 *     [3H,H], gate order r,z,n; nn.Linear [out,in]; nn.Embedding [V,E]) so that
 *     checkpoints load unchanged;
 *   - fp32 everywhere; token ids are int64 (torch.long), sequence-major [T,B] where
 *     the reference uses pad_sequence (reinforce_omics.py:304).
 */
#ifndef PMG_H
#define PMG_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PMG_MAX_LAYERS 4
#define PMG_ABI_VERSION 3

/* workspace modes */
#define PMG_MODE_DECODE 0 /* pmg_decode */
#define PMG_MODE_TRAIN 1  /* pmg_logprob_fwd followed by pmg_reinforce_bwd on the same ws */

/* flags */
#define PMG_FLAG_NO_EARLY_EXIT 1u /* decode: run all gen_len steps even after every row hit stop */
#define PMG_FLAG_ACCUMULATE 2u    /* bwd / wgrad: add into grad_out AND loss_out instead of overwriting them (chunked batches) */
#define PMG_FLAG_REUSE_PACK 16u   /* logprob_fwd: `ws` still holds the packed weights of the previous
                                     pmg_logprob_fwd call on it (same weights, same model dimensions; B and T
                                     may differ): skip the packing kernels.  For batches run in chunks. */

#define PMG_FLAG_NO_CLUSTER 32u   /* decode / logprob_fwd / reinforce_bwd / reinforce_bptt: never take the clustered small-batch kernels (see below).
                                     Pass it on launches that may run CONCURRENTLY with another pmg launch on the
                                     same GPU (other streams or processes), and pass it to pmg_reinforce_bwd /
                                     pmg_reinforce_bptt whenever the pmg_logprob_fwd that filled the workspace had
                                     it and vice versa: a backward with the flag on a workspace whose forward ran
                                     without it returns NaN (the clustered forward keeps part of the history in a
                                     layout only the clustered backward reads). */

typedef void* pmg_stream; /* hipStream_t */

/* Problem shape.  B = sequences in this call, T = time steps:
 *   decode : T = generate_len (reinforce.py:168)
 *   train  : T = rows of padded_nums (reinforce_omics.py:304); T-1 decoder steps run.
 * H rnn_cell_size, E embedding_size, W stack_width, D stack_depth, V vocab_size,
 * L n_layers, Z decoder latent_dim, pad_idx = nn.Embedding padding_idx (-1: none). */
typedef struct pmg_shape {
  int32_t B, T, H, E, W, D, V, L, Z, pad_idx;
} pmg_shape;

/* Decoder parameters, torch layouts, fp32 device pointers. */
typedef struct pmg_weights {
  const float* embedding;            /* [V,E]                                  */
  const float* w_ih[PMG_MAX_LAYERS]; /* l=0: [3H,E+W]; l>0: [3H,H]             */
  const float* w_hh[PMG_MAX_LAYERS]; /* [3H,H]                                 */
  const float* b_ih[PMG_MAX_LAYERS]; /* [3H]                                   */
  const float* b_hh[PMG_MAX_LAYERS]; /* [3H]                                   */
  const float* w_ctrl;               /* stack_controls_layer.weight [3,H]      */
  const float* b_ctrl;               /* [3]                                    */
  const float* w_push;               /* stack_input_layer.weight [W,H]         */
  const float* b_push;               /* [W]                                    */
  const float* w_out;                /* output_layer.weight [V,H]              */
  const float* b_out;                /* [V]                                    */
  const float* w_l2h;                /* latent_to_hidden.weight [H,Z] (decode) */
  const float* b_l2h;                /* [H]                         (decode)   */
} pmg_weights;

/* Same layout, writable: gradients of the loss w.r.t. each parameter. */
typedef struct pmg_grads {
  float* embedding;
  float* w_ih[PMG_MAX_LAYERS];
  float* w_hh[PMG_MAX_LAYERS];
  float* b_ih[PMG_MAX_LAYERS];
  float* b_hh[PMG_MAX_LAYERS];
  float* w_ctrl;
  float* b_ctrl;
  float* w_push;
  float* b_push;
  float* w_out;
  float* b_out;
} pmg_grads;

/* ABI version of the loaded library (== PMG_ABI_VERSION). */
int pmg_abi_version(void);

/* Text for a negative return code. */
const char* pmg_strerror(int code);

/* 0 if `s` is a shape the kernels support -- what StackGRUDecoder(params) of the reference's
 * drivers admits (examples/IC50/train_paccmann_rl.py:93-97) within: H a multiple of 128 in
 * [128,1024], 1<=W<=64, D>=2, 2<=V<=1024, 1<=L<=PMG_MAX_LAYERS, Z a multiple of 32 (decode), and the
 * per-workgroup state fitting the CU's 160 KiB of LDS: the hidden state of every layer is kept there
 * (L x 16 x H floats), so L x H up to about 2048 fits -- e.g. H = 1024: one layer with any V, two
 * layers with 3 + W + V <= 400; H = 512: three layers with V <= 859, four with V <= 331 at W = 50
 * (table in DESIGN.md section 6) -- else the error code every entry point would return for it. */
int pmg_check_shape(const pmg_shape* s, int mode);

/* Which launches would take the clustered small-batch kernels for shape `s` with these flags:
 * PMG_MODE_DECODE: 1 if pmg_decode does; PMG_MODE_TRAIN: bit 0 for pmg_logprob_fwd, bit 1 for
 * pmg_reinforce_bptt (the recurrent half of pmg_reinforce_bwd); 0 = the batch-partitioned kernels,
 * negative for an unsupported shape.  Clustered: one layer, H <= 512, 3+W+V <= 256, and at most NG row
 * groups of 16 sequences per cluster where there are 8 * floor(512 / H) clusters: NG = 1 at H = 128,
 * up to 4 at H = 256 (B <= 1024), above that as many as the owned sequences' stacks fit the CU's LDS
 * (6 at H = 512 with a 50 x 50 stack: B <= 768; the BPTT stops a little earlier).  A cluster of H/16
 * workgroups (one per CU) synchronises through device memory twice per time step, so every cluster
 * member must be resident at once: see PMG_FLAG_NO_CLUSTER.  Same results either way (to the tolerances of the parity
 * tests; greedy tokens exact). */
int pmg_uses_cluster(const pmg_shape* s, int mode, uint32_t flags);

/* Bytes of device scratch needed for `mode` at shape `s` (256-byte aligned base). */
int pmg_workspace_bytes(const pmg_shape* s, int mode, size_t* out);

/* Batched autoregressive sampling -- replaces generator.generate(...) as called at
 * reinforce.py:162-170 (TeacherVAE.generate / StackGRUDecoder.generate_from_latent /
 * SamplingSearch of paccmann_chemistry): h0 = latent_to_hidden(latent) for every
 * layer, zero stack, feed start_tok, then gen_len x {decoder step, output_layer,
 * softmax(logits/temperature), one categorical draw}.  temperature <= 0 selects
 * argmax (GreedySearch).  Draws use Philox4x32-10 keyed by philox_seed with counter
 * (seq_index_base + row, philox_offset + step), so a batch sharded over GPUs samples
 * the same tokens as the unsharded batch.
 *   latent     [B,Z]   (w->b_l2h must be 16-byte aligned; 0 <= start_tok, stop_tok < V)
 *   tokens_out [B,gen_len] every sampled token (start token not included)
 *   len_out    [B] tokens before the first stop_tok (gen_len if none): the crop
 *              TeacherVAE.generate applies (consumed at reinforce.py:172-187)
 * A workgroup stops early once all of its rows have emitted stop_tok (remaining
 * positions are filled with stop_tok) unless PMG_FLAG_NO_EARLY_EXIT is set.
 * Small batches (pmg_uses_cluster) run on clusters of workgroups with bounded spin barriers: if the
 * members of a cluster cannot all become resident (another spinning kernel holds the CUs for about a
 * second) the launch gives up and POISONS its outputs -- len_out = -1 for every row here, NaN in
 * logp_out of pmg_logprob_fwd -- instead of hanging or returning plausible numbers. */
int pmg_decode(const pmg_shape* s, const pmg_weights* w, const float* latent,
               int start_tok, int stop_tok, int gen_len, float temperature,
               uint64_t philox_seed, uint64_t philox_offset, int64_t seq_index_base,
               uint32_t flags, int64_t* tokens_out, int32_t* len_out,
               void* ws, pmg_stream stream);

/* pmg_decode for callers that cannot allocate device memory (code under HIP graph capture, tight
 * serving loops, harnesses that forbid hipMalloc).  NO COUNTERPART IN THE REFERENCE: it is
 * pmg_decode with the workspace argument replaced by a 32 MiB static device array inside the
 * library (pmg_noalloc_scratch_bytes()), which every GPU the library runs on holds until the process
 * ends.  Shapes whose pmg_workspace_bytes(s, PMG_MODE_DECODE) exceed that size are rejected with a
 * negative code (at the configs[1] model shape: up to 1328 sequences per call).  Calls no
 * hipMalloc/hipFree or any allocating routine; same results as pmg_decode, bit for bit.  All calls
 * on one GPU share that GPU's array: calls that overlap on one GPU (different streams or threads)
 * race; calls ordered on one stream are safe. */
int pmg_decode_noalloc(const pmg_shape* s, const pmg_weights* w, const float* latent,
                       int start_tok, int stop_tok, int gen_len, float temperature,
                       uint64_t philox_seed, uint64_t philox_offset, int64_t seq_index_base,
                       uint32_t flags, int64_t* tokens_out, int32_t* len_out, pmg_stream stream);

/* Size in bytes of the static scratch array behind pmg_decode_noalloc. */
size_t pmg_noalloc_scratch_bytes(void);

/* Teacher-forced trajectory log-probs -- replaces the loop at
 * reinforce_omics.py:319-329 (reinforce_proteins.py:307-318): zero stack, then for
 * p in 0..T-2: decoder(tokens[p], hidden, stack); output_layer; log_softmax;
 * gather(tokens[p+1]).  No padding mask (as the reference).  Leaves in `ws` what
 * pmg_reinforce_bwd needs.
 *   tokens   [T,B] int64 (pad_sequence layout)
 *   h0       [L,B,H] = latent_to_hidden(...) (reinforce_omics.py:313-317)
 *   logp_out [T-1,B]
 *   h_out    [L,B,H] final hidden, may be NULL
 *   flags    0 or PMG_FLAG_REUSE_PACK
 * A batch too large for one workspace runs as chunks on one `ws`: per chunk pmg_logprob_fwd then
 * pmg_reinforce_bwd, chunks after the first with PMG_FLAG_REUSE_PACK / PMG_FLAG_ACCUMULATE and the
 * same `scale` (BASELINE.json configs[3]: B=16384, T=258 as 4 chunks of 4096). */
int pmg_logprob_fwd(const pmg_shape* s, const pmg_weights* w, const int64_t* tokens,
                    const float* h0, float* logp_out, float* h_out, uint32_t flags,
                    void* ws, pmg_stream stream);

/* Reward-weighted BPTT -- replaces rl_loss.backward() (reinforce_omics.py:335) for
 * rl_loss = -sum_p mean_b(logp[p,b] * reward[b]) (:330-332), with mean = scale * sum
 * (scale = 1/B_valid; a data-parallel caller passes 1/B_valid_global).  Must follow
 * pmg_logprob_fwd with the same shape, tokens, weights and ws.
 *   reward   [B]
 *   grad_out gradients of every decoder parameter reachable from the loss
 *            (latent_to_hidden is reached through grad_h0 by the caller's autograd)
 *   grad_h0  [L,B,H], 16-byte aligned
 *   loss_out [1]; with PMG_FLAG_ACCUMULATE the call's loss is added to it, like the gradients
 * Token ids outside [0,V) are clamped by the kernels (memory safety only; validate on the host). */
int pmg_reinforce_bwd(const pmg_shape* s, const pmg_weights* w, const int64_t* tokens,
                      const float* reward, float scale, uint32_t flags,
                      const pmg_grads* grad_out, float* grad_h0, float* loss_out,
                      void* ws, pmg_stream stream);

/* The two halves of pmg_reinforce_bwd as entry points of their own (pmg_reinforce_bwd is exactly
 * pmg_reinforce_bptt followed by pmg_reinforce_wgrad on the same stream), for callers that time or
 * schedule them apart: the BPTT is a latency-bound recurrence, the weight-gradient GEMM a
 * throughput-bound contraction over all B*T tokens.
 *   pmg_reinforce_bptt : walks time backwards from the trajectory pmg_logprob_fwd left in `ws`; writes
 *                        grad_h0 and, into `ws`, the per-step gate / extras gradient slabs and the
 *                        loss partials; flags: 0 or PMG_FLAG_NO_CLUSTER (small batches run on clusters
 *                        of workgroups like the forward, see pmg_uses_cluster; a launch that gave up
 *                        leaves NaN in grad_h0 and in the loss).
 *   pmg_reinforce_wgrad: contracts those slabs with the saved activations into grad_out (every
 *                        decoder parameter) and loss_out; flags: 0 or PMG_FLAG_ACCUMULATE. */
int pmg_reinforce_bptt(const pmg_shape* s, const pmg_weights* w, const int64_t* tokens,
                       const float* reward, float scale, uint32_t flags, float* grad_h0, void* ws,
                       pmg_stream stream);
int pmg_reinforce_wgrad(const pmg_shape* s, const pmg_weights* w, const int64_t* tokens, uint32_t flags,
                        const pmg_grads* grad_out, float* loss_out, void* ws, pmg_stream stream);

/* One decoder step -- the duck-typed `decoder(tok[1,B], hidden, stack)` call of
 * reinforce_omics.py:323-325 plus `decoder.output_layer` (:326).
 *   tok [B] int64; hidden_in/out [L,B,H]; stack_in/out [B,D,W]; logits_out [B,V]
 *   (any of stack_in == NULL means a zero stack; logits_out may be NULL). */
int pmg_step(const pmg_shape* s, const pmg_weights* w, const int64_t* tok,
             const float* hidden_in, const float* stack_in, float* hidden_out,
             float* stack_out, float* logits_out, void* ws, pmg_stream stream);

/* Batched detokenisation -- replaces the per-molecule host loop
 * `smiles_language.token_indexes_to_smiles(mol_num.tolist())` of reinforce.py:172-176 for
 * the whole batch pmg_decode produced: row b of the result is the
 * concatenation of the table strings of tokens[b, 0..len[b]), packed back to back.
 *   tokens    [B, ld_tokens] int64 (tokens_out of pmg_decode, ld_tokens = gen_len)
 *   len       [B] int32 (len_out of pmg_decode), clamped to [0, max_len]
 *   tab_off   [V+1] int32, tab_bytes: token i is tab_bytes[tab_off[i] .. tab_off[i+1]) (UTF-8);
 *             ids outside [0,V) and empty entries add nothing (index_to_token.get(i, ''))
 *   max_tok_bytes  longest table string; longer entries are truncated to it
 *   out_off   [B+1] int64: row b is out_bytes[out_off[b] .. out_off[b+1]); out_off[B] = total
 *   out_bytes [out_cap] uint8, out_cap >= B * max_len * max_tok_bytes (checked on the host,
 *             so no device-side overflow is possible and no synchronisation is needed)
 * No workspace. */
int pmg_detokenize(const int64_t* tokens, int64_t ld_tokens, const int32_t* len, int B, int max_len,
                   const int32_t* tab_off, const uint8_t* tab_bytes, int V, int max_tok_bytes,
                   int64_t* out_off, uint8_t* out_bytes, size_t out_cap, pmg_stream stream);

#ifdef __cplusplus
}
#endif
#endif /* PMG_H */
