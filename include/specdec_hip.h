/*
 * specdec_hip.h — C-ABI of the MI355X (gfx950) speculative-decoding hot path.
 *
 * This is the drop-in boundary for GogoRit/llm-inference-lab's kernel registry
 * (`src/kernels/registry.py:11-123`, ops "verify_prefix" and "kv_append",
 * `src/kernels/__init__.py:84-112`) and for the model-wrapper forward that the
 * pipeline calls through `LanguageModel.generate_tokens`
 * (`src/specdec/utils/interfaces.py:14-138`, call sites
 * `src/specdec/core/pipeline.py:2397, 2585`).
 *
 * Conventions
 *   - plain pointers + sizes only; no torch / STL types cross this boundary
 *   - every pointer named `*_dev`, `logits`, `ids`, cache/new/out is DEVICE memory
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream)
 *   - every entry point returns 0 on success, non-zero on error; the message is
 *     available from sd_last_error() (thread-local); nothing throws
 *   - no allocation, no synchronisation and no host<->device copy inside an
 *     entry point unless its comment says so: all of them are graph-capturable
 *   - inputs are borrowed and never written; outputs are caller-owned
 */
#ifndef SPECDEC_HIP_H
#define SPECDEC_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* element types (values are part of the ABI) */
enum sd_dtype {
  SD_F32 = 0,
  SD_F16 = 1,
  SD_BF16 = 2,
  SD_I32 = 3,
  SD_I64 = 4,
  SD_U8 = 5,
  SD_FP8_E4M3 = 6
};

#define SD_ABI_VERSION 1

/* ABI version of the loaded library (== SD_ABI_VERSION it was built with). */
int sd_abi_version(void);

/* Last error message of the calling thread ("" if none). Never NULL. */
const char* sd_last_error(void);

/* ------------------------------------------------------------------------
 * verify_prefix — replaces the registry op "verify_prefix"
 *   reference contract: verify_prefix_ref, src/kernels/reference.py:13-56
 *   reference callers : LongestPrefixPolicy.accept_tokens, policies.py:128-134
 *   (the dead CUDA attempt it supersedes: src/kernels/cuda/verify.cu:34-215)
 *
 *   pred[b,k]      = argmax_v logits[b,k,v]      (lowest index wins ties, NaN
 *                                                 counts as the maximum — the
 *                                                 torch.argmax rule)
 *   accept_len[b]  = length of the longest prefix with pred[b,k] == ids[b,k]
 *   mask[b,k]      = 1 for k < accept_len[b], else 0   (prefix-only mask)
 *
 *   logits  : [B][K][V], dtype SD_F32 | SD_F16 | SD_BF16, last dim contiguous,
 *             element strides stride_b / stride_k
 *   ids     : [B][K] contiguous, dtype SD_I32 | SD_I64
 *   pred_out: optional [B][K] int32 (NULL to skip) — the argmax ids
 *   workspace: device scratch of at least sd_verify_prefix_workspace(B,K,V) bytes
 * ------------------------------------------------------------------------ */
size_t sd_verify_prefix_workspace(int B, int K, int V);

int sd_verify_prefix(const void* logits, int logits_dtype,
                     const void* ids, int ids_dtype,
                     int32_t* accept_len, uint8_t* mask, int32_t* pred_out,
                     int B, int K, int V,
                     int64_t stride_b, int64_t stride_k,
                     void* workspace, size_t workspace_bytes,
                     void* stream);

/* Sampled token from logits rows: temperature -> top-k -> top-p -> one categorical draw.
 * Replaces sample_bonus_token_from_logits (src/specdec/core/pipeline.py:48-147) for
 * do_sample=True. Entry b reads row  b*rows_per_b + (pos ? pos[b] : 0)  of `logits`
 * ([rows][V], row_stride elements apart, f32 / bf16 / f16) and writes out_ids[b]; entries with
 * active[b] == 0 (if given) are skipped. Order among equal logits: value descending, index
 * ascending. The nucleus cut drops a token when the inclusive cumulative probability (descending
 * order, inside the top-k) exceeds top_p; the first token is always kept. The draw is one
 * Philox4x32-10 value with key = seed and counter = (draw index, stream, 0, tag), inverted through
 * the cumulative weights; draw index = draw_counters[b] (then incremented) or draw0 when
 * draw_counters is NULL; stream = stream_id[b] or b. top_k in 1..1024; top_k <= 0 with
 * top_p >= 1 draws from the whole row (Gumbel-max, one Philox value per element); top_k <= 0
 * with top_p < 1 is the full-vocabulary nucleus (the reference with top_k = None): the row is
 * consumed in rank blocks of 1024 (exact radix select per block) until the cumulative probability
 * — over the softmax of the WHOLE row — passes top_p. Asynchronous on `stream`, no allocation, graph-capturable. */
int sd_sample_token(const void* logits, int logits_dtype, int64_t row_stride, int B, int V,
                    const int32_t* pos, int rows_per_b, const int32_t* active, float temperature,
                    int top_k, float top_p, uint64_t seed, uint32_t* draw_counters, uint32_t draw0,
                    const int32_t* stream_id, int32_t* out_ids, void* stream);

/* ------------------------------------------------------------------------
 * kv_append (in place) — the KV-append path
 *   reference contract: kv_append_ref, src/kernels/reference.py:59-93
 *   reference callers : HFWrapper._append_kv_with_kernel, hf_wrappers.py:985-1029;
 *                       SafeKVCacheManager.update_*_cache, kv_cache_manager.py:194-273
 *
 *   cache_k/v : [B][H][Lmax][D] preallocated, contiguous
 *   new_k/v   : [B][H][K][D] contiguous
 *   row_len   : optional device int32[B]: row b is appended at row_len[b];
 *               NULL → every row is appended at `L`
 *   Rows [row_len[b], row_len[b]+K) of the cache are written; nothing else is
 *   touched, so views of the first row_len[b] rows stay valid (the
 *   `torch.cat` of the reference re-copies all L rows instead).
 *   elem_size : bytes per element (2 or 4); the copy is type-agnostic
 * ------------------------------------------------------------------------ */
int sd_kv_append(void* cache_k, void* cache_v,
                 const void* new_k, const void* new_v,
                 const int32_t* row_len, int L,
                 int elem_size, int B, int H, int Lmax, int K, int D,
                 void* stream);

/* ------------------------------------------------------------------------
 * kv_concat (out of place) — exact output shape of the registry op "kv_append"
 *   out[b,h,0:L]   = base[b,h,0:L]
 *   out[b,h,L:L+K] = new[b,h,0:K]          out: [B][H][out_cap][D], out_cap >= L+K
 *   (out_cap == L+K gives the contiguous reference shape; a larger out_cap lands
 *    the result in a cache with head-room for later in-place appends)
 *   base may be a strided view: element strides base_sb / base_sh, rows of D
 *   contiguous elements at stride D. K and V are moved by one launch.
 * ------------------------------------------------------------------------ */
int sd_kv_concat(void* out_k, void* out_v,
                 const void* base_k, const void* base_v,
                 const void* new_k, const void* new_v,
                 int elem_size, int B, int H, int L, int K, int D, int out_cap,
                 int64_t base_sb, int64_t base_sh,
                 void* stream);

/* ------------------------------------------------------------------------
 * kv_append_masked — compacting append
 *   reference contract: kv_append_with_mask_ref, src/kernels/reference.py:96-159
 *   (supersedes src/kernels/cuda/kv_cache.cu:14-173, including its zero-accept
 *    bug at :40 — a zero-accept row keeps its base rows here, as the reference
 *    test tests/test_kv_cache.py:164-186 requires)
 *
 *   out : [B][H][L+K][D] contiguous; every element is written:
 *         rows [0,L)          = base
 *         rows L+j, j<n_b     = draft row of the j-th set bit of mask[b,:]
 *         rows L+n_b .. L+K-1 = 0
 *         n_b = accept_len[b]==0 ? 0 : min(popcount(mask[b,:]), accept_len[b] < 0 ? 1 : accept_len[b])
 *               (a negative accept_len is invalid input; the reference loop writes one row for it, so does this)
 *   mask : device uint8[B][K] (K <= 64), accept_len : device int32[B]
 * ------------------------------------------------------------------------ */
int sd_kv_append_masked(void* out_k, void* out_v,
                        const void* base_k, const void* base_v,
                        const void* draft_k, const void* draft_v,
                        const uint8_t* mask, const int32_t* accept_len,
                        int elem_size, int B, int H, int L, int K, int D,
                        int64_t base_sb, int64_t base_sh,
                        void* stream);

/* ========================================================================
 * Decoder forward + draft-then-verify loop
 *
 * The reference reaches the model through LanguageModel.generate_tokens
 * (src/specdec/utils/interfaces.py:30-62; HFWrapper._generate_tokens_async,
 * src/specdec/models/hf_wrappers.py:272-627 — one HF forward per generated token,
 * argmax of the last position) and drives it from the step loop of
 * SpeculativePipeline.generate_batch / generate (src/specdec/core/pipeline.py:
 * 1984-3733, 984-1275). These entry points are what a ctypes binding of that path
 * binds instead: a model instance over caller-owned weights and caches, a forward
 * that appends to the KV cache in place, and one call per draft-then-verify step.
 * ======================================================================== */

enum sd_arch { SD_ARCH_LLAMA = 0, SD_ARCH_GPT2 = 1 };

/* Per-layer weights. Linear weights are [out_features][in_features] row-major
 * (HF nn.Linear layout), bf16. Biases / LayerNorm biases may be NULL (Llama). */
typedef struct sd_layer_weights {
  const void* attn_norm_w;  /* [d]  input_layernorm / ln_1 */
  const void* attn_norm_b;  /* [d]  GPT-2 only */
  const void* wqkv;         /* [(Hq+2*Hkv)*D][d]  q rows, then k rows, then v rows */
  const void* bqkv;
  const void* wo;           /* [d][Hq*D] */
  const void* bo;
  const void* mlp_norm_w;   /* [d]  post_attention_layernorm / ln_2 */
  const void* mlp_norm_b;
  const void* w_up;         /* Llama: [2*ff][d] gate rows then up rows; GPT-2: c_fc [ff][d] */
  const void* b_up;
  const void* w_down;       /* [d][ff] */
  const void* b_down;
} sd_layer_weights;

typedef struct sd_model_config {
  int arch;                 /* enum sd_arch */
  int n_layers, d_model, n_heads, n_kv_heads, head_dim, d_ff, vocab, max_pos;
  float norm_eps;
  int weight_dtype;         /* SD_BF16: stream the matrices as given. SD_FP8_E4M3: the matrices are still passed as
                             * bf16; sd_pack_weights quantises them (OCP e4m3, per-output-row absmax/448 scale) into
                             * `packed` and the forward streams that copy: half the HBM bytes, bf16 activations */
  const void* tok_emb;      /* [vocab][d] */
  const void* pos_emb;      /* [max_pos][d], GPT-2 only */
  const void* final_norm_w; /* [d] */
  const void* final_norm_b;
  const void* lm_head;      /* [vocab][d] (may alias tok_emb) */
  const float* rope_cos;    /* [max_pos][D/2] fp32, Llama only */
  const float* rope_sin;
  const sd_layer_weights* layers; /* host array [n_layers] (copied by sd_model_create) */
  const void* packed;       /* optional: buffer written by sd_pack_weights for THIS config; the
                               GEMVs then stream the packed copy (the row-major matrices are no
                               longer read by the forward, except tok_emb for the gather) */
} sd_model_config;

/* The weight quantiser of the fp8 storage mode on a row-major bf16 matrix [N][K]:
 * scales[r] = max|w[r][:]| / 448 (1 for an all-zero row), q[r][k] = OCP e4m3 (round to nearest even) of
 * w[r][k] / scales[r]. sd_pack_weights applies exactly this before re-ordering. Asynchronous on `stream`. */
int sd_quantize_fp8_rows(const void* w_bf16, int N, int K, void* q_fp8, float* scales, void* stream);

/* One vocabulary-sized output matrix [vocab][d_model] (bf16, row-major) outside a model — a Medusa head — in the
 * engine's tile-stream order, bf16 or fp8 e4m3 with row scales (the lm_head's layout and quantiser). */
size_t sd_packed_head_bytes(int vocab, int d_model, int weight_dtype);
int sd_pack_head(const void* w_bf16, int vocab, int d_model, int weight_dtype, void* dst, size_t dst_bytes, void* stream);

/* Weight pre-packing: the engine's private copy of all Linear weights in the order the
 * streaming GEMV consumes them (csrc/pack.hip), so that every wave reads one contiguous
 * region of HBM. sd_packed_bytes gives the buffer size (device memory, 256-byte aligned);
 * sd_pack_weights fills it (asynchronous on `stream`). */
size_t sd_packed_bytes(const sd_model_config* cfg);
int sd_pack_weights(const sd_model_config* cfg, void* dst, size_t dst_bytes, void* stream);

typedef struct sd_model sd_model;

/* Create a model over caller-owned device weights (borrowed for the model's life). */
int sd_model_create(const sd_model_config* cfg, sd_model** out);
int sd_model_destroy(sd_model* m);

/* Tokens one pass of sd_model_forward covers: 128 or 64 when every matrix of the model has a shape the
 * multi-token kernel (gemm_skinny.hip) handles at that size, else 9 or fewer (x rows must fit the LDS). Larger B*M are tiled into passes. */
int sd_model_pass_tokens(const sd_model* m);

/* Scratch the forward needs (activations of one pass + argmax partials). */
size_t sd_model_workspace_bytes(const sd_model* m);
/* Bytes of ONE of the two cache tensors, bf16:
 *   K: [n_layers][B][Hkv][Lmax][D]      V: [n_layers][B][Hkv][D][Lmax] (transposed)
 * Lmax must be a multiple of 8. */
size_t sd_model_kv_bytes(const sd_model* m, int B, int Lmax);
/* Attach caller-owned KV caches and workspace. */
int sd_model_bind(sd_model* m, void* k_cache, void* v_cache, int B, int Lmax,
                  void* workspace, size_t workspace_bytes);

/* Paged KV (the counterpart of the reference's per-sequence cache tensors that grow by torch.cat
 * and are realigned to a common length, src/specdec/cache/kv_cache_manager.py:194-199 and :353-479): instead of Lmax
 * positions per row, rows own PAGES of page_len positions (a power of two >= 32) out of pools shared by all rows:
 *   K pool: [n_layers][n_pages][Hkv][page_len][D]    V pool: [n_layers][n_pages][Hkv][D][page_len]   (bf16)
 * and position pos of row b lives in page block_table[b * max_pages_per_row + pos / page_len] (the same page index in
 * every layer) at offset pos % page_len. `block_table` is device memory, caller-owned and caller-maintained: an entry
 * must be valid before a forward reads or writes a position in it (write entries on the stream the forward runs on, or
 * synchronise); entries of positions a row has not reached are never read. The row's capacity is
 * max_pages_per_row * page_len. Everything else (forward, step loop, C-ABI) is unchanged; sd_model_probe_gemv's QKV
 * probe is not available on a paged model. */
size_t sd_model_kv_pool_bytes(const sd_model* m, int n_pages, int page_len);   /* ONE of the two pools */
int sd_model_bind_paged(sd_model* m, void* k_pool, void* v_pool, int n_pages, int page_len,
                        const int32_t* block_table, int max_pages_per_row, int B,
                        void* workspace, size_t workspace_bytes);

/* One forward over M new tokens per batch row, appended to the KV cache in place.
 *   tokens   : device int32, token (b,m) at tokens[b*tok_stride + m]
 *   pos_base : device int32[B]; token (b,m) sits at position pos_base[b] + pos_off + m
 *              and attends to cache positions 0 .. that position (its own K/V are
 *              written first — the KV-append path of kv_append_ref, fused)
 *   ids_out  : optional device int32, argmax of token (b,m) at ids_out[b*ids_stride+m]
 *   logits_out: optional [B][M][vocab] (SD_BF16 | SD_F32)
 *   skip_head: 1 = only fill the cache (prefill of all but the last token)
 *   row0, B  : the call works on rows [row0, row0+B) of the bound batch; element 0 of
 *              tokens / pos_base / ids_out / logits_out belongs to row row0
 * B*M may exceed the 9 tokens one pass holds; the call then makes several passes. */
int sd_model_forward(sd_model* m, const int32_t* tokens, int tok_stride,
                     const int32_t* pos_base, int pos_off, int row0, int B, int M,
                     int32_t* ids_out, int ids_stride,
                     void* logits_out, int logits_dtype, int skip_head, void* stream);

/* The residual-stream rows (bf16 [n][d_model], BEFORE the final norm) that the last pass of the last sd_model_forward
 * left in the workspace: token (b, m) of that pass is row b*M + m. Rows [row0, row0+n) are copied to `out` (device
 * memory), asynchronously on `stream`. This is `outputs.hidden_states` of the reference's draft modes one norm earlier:
 * _run_medusa_hf / _run_eagle_hf read hidden_states[-1][:, -1] (src/specdec/core/pipeline.py:674-686, 788-800), i.e.
 * final_norm(row). */
int sd_model_hidden_rows(sd_model* m, int row0, int n, void* out, void* stream);

/* Measurement hook for bench.py's roofline leg: launches ONE of the forward's weight-
 * streaming GEMVs (which: 1 = attention out-proj, 2 = norm+gate/up+SwiGLU, 3 = down-proj,
 * 4 = final norm+lm_head+argmax) `iters` times, round-robin over the layers so the weights
 * stream from HBM, between two HIP events on `stream`; returns the average launch duration
 * and the algorithmic bytes of one launch (N*K*2). Synchronises on the second event. */
int sd_model_probe_gemv(sd_model* m, int which, int T, int iters, void* stream,
                        float* avg_usec, double* bytes_per_launch);

/* ---- persistent forward (csrc/persist.hip) ---------------------------------
 * Passes of <= sd_model_persist_tokens(m) tokens of a dense-KV Llama model with packed bf16 weights run as ONE launch
 * (an LDS-DMA loader wave per CU streams the weights ahead of every dependency; activations cross CUs as tagged 8-byte
 * granules) instead of five launches per layer. It replaces the same reference code as sd_model_forward does (the k-step
 * draft loop, src/specdec/models/hf_wrappers.py:417-539). 0 = not available for this model / device (other
 * architectures, fp8 or row-major weights, paged KV, a GPU without 256 CUs, SPECDEC_NO_PERSIST=1); the environment
 * variable SPECDEC_PERSIST_MAX_T (read by sd_model_bind) sets the token limit (default: 2 for models with d_model <= 2048, where the
 * launch is measured faster than the launch path; 0 for wider models, where it is not; at most 8). Whether a given pass takes it
 * also depends on the rows' current length: sd_model_set_length_hint below. */
int sd_model_persist_tokens(const sd_model* m);

/* Health word of the persistent launches of `m` since it was bound: 0 = every launch ran to completion. Every wait
 * inside the launch is bounded (50 ms); a workgroup that gives up ORs a reason into this word (1 loader blocked, 2 weights
 * never landed, 4 / 8 intra-workgroup hand-over, 16 granules of another CU never arrived, 32 attention hand-over) and
 * leaves, and the pass's outputs are then invalid. Synchronises `stream` (one 4-byte copy to the host). The step loop
 * also carries the word of its models in every step record (sd_specdec_record). */
int sd_model_engine_status(sd_model* m, uint32_t* status_out, void* stream);

/* The same word as a pinned host location (NULL before the model is bound): a workgroup that gives up also stores its
 * reason there, so whoever has just synchronised with a pass to read its ids / logits can check the pass's health with a
 * plain load, no copy, no further synchronisation. 0 = every persistent launch since the bind / the last clear completed. */
const uint32_t* sd_model_status_word(const sd_model* m);

/* Recovery after a non-zero health word: synchronises `stream`, clears the device word and the host word and moves the launch
 * counter past the failed launch (so that granules it left behind never match again). The passes since the failure are
 * invalid and must be repeated — typically after sd_model_set_persist_tokens(m, 0), i.e. on the launch path.
 * PRECONDITION of the persistent launch, for direct C callers: it needs all 256 CUs to itself while it runs (one workgroup
 * per CU, each declaring the CU's whole LDS, spinning on the others' hand-offs). A second process on the same GPU, or a
 * second stream of this process running another LDS-heavy kernel at the same time, can split the CUs between the two;
 * the launch then gives up after its 50 ms bound (it never hangs) and reports here. Share a GPU only with
 * SPECDEC_NO_PERSIST=1 or sd_model_set_persist_tokens(m, 0). */
int sd_model_engine_status_clear(sd_model* m, void* stream);

/* Tokens per pass the persistent launch takes from now on: min(max_tokens, what the model / cache allows); 0 = launch path
 * only. Host-side state: steps already captured (sd_specdec_step) keep the kernels they were captured with — call
 * sd_specdec_invalidate on the loops that use the model. */
int sd_model_set_persist_tokens(sd_model* m, int max_tokens);

/* The caller's bound on the CURRENT length (cached positions) of the rows the coming passes touch; max_len <= 0 or beyond
 * the cache: the cache size (the default after a bind). The persistent launch walks a head's whole cache on one CU and is
 * the faster path up to 1280 positions only, so a session bound for a long context starts on it and moves to the launch
 * path when its rows pass that length (host-side state, as above: re-capture after crossing). */
int sd_model_set_length_hint(sd_model* m, int max_len);

/* 1 when a pass of T tokens of one row would run as the persistent launch right now, else 0. */
int sd_model_persist_active(const sd_model* m, int T);

/* Rows [row0, row0+n) of one of the workspace buffers the last pass left behind (bf16, asynchronous copy on `stream`):
 * which = 0 residual stream [d_model] (= sd_model_hidden_rows), 1 q after RoPE [Hq*D], 2 attention rows [Hq*D],
 * 3 MLP activation [d_ff]; all of the LAST layer. For stage-by-stage checks of the persistent launch against the
 * launch-per-operator forward. (The draft model of an sd_specdec loop runs the persistent launch WITHOUT these stores unless
 * SPECDEC_PERSIST_TAPS is set when the loop is created: its rows are then not updated by the loop's passes.) */
int sd_model_debug_rows(sd_model* m, int which, int row0, int n, void* out, void* stream);

/* Measurement hook: `iters` whole forwards of one row of M tokens (token id 0, positions pos0..pos0+M-1 of cache row 0,
 * which they overwrite; the attention reads the pos0 positions below, whatever the cache holds) between two HIP events on `stream`; returns the average duration of a forward and the bytes of weights
 * it streams. skip_head = 1 leaves the lm_head out. timeline (optional, host memory, `timeline_cap` uint64): in-kernel
 * 100 MHz stamps of ONE further persistent forward, [256 workgroups][12 * n_ops + 4] (per op: gather start, input staged,
 * MFMA + epilogue done, attention done, third consumer's start and MFMA end, leader's MFMA end, loader done issuing the
 * op; then s_memrealtime / s_memtime at the workgroup's start and end: the shader clock it ran at); allocates and frees a device buffer and synchronises. */
int sd_model_probe_forward(sd_model* m, int M, int pos0, int iters, int skip_head, void* stream, float* avg_usec,
                           double* bytes_per_forward, unsigned long long* timeline, size_t timeline_cap);

/* ---- the step loop ------------------------------------------------------- */
typedef struct sd_specdec sd_specdec;

enum sd_emit_mode {
  SD_EMIT_BONUS = 0,   /* generate_batch: base tokens t_0..t_{a-1} + bonus t_a (pipeline.py:3059-3292) */
  SD_EMIT_DRAFT = 1    /* generate: draft tokens d_1..d_a, or t_0 if a == 0 (pipeline.py:1190-1235) */
};

/* Creation allocates the (tiny) device loop state and a pinned host mirror. Both
 * models must be bound with the same B.
 * draft == NULL selects the self-draft mode: Medusa-lite with heads tied to the lm_head, greedy — the
 * reference draftor (src/specdec/modes/medusa.py:71-186) evaluates head 0 on the same last hidden state for
 * all K proposals, i.e. proposes K copies of the target's own next token; the step is then one 1-token target
 * forward + the K+1-token verify forward. */
int sd_specdec_create(sd_model* draft, sd_model* target, int B, int K, int emit_mode,
                      sd_specdec** out);
int sd_specdec_destroy(sd_specdec* s);

/* (Re)initialise row b: `seq_len` tokens are final, the last two of them are
 * (prev_tok, last_tok); caches must hold positions [0, seq_len-1) of the target and
 * [0, seq_len-2] or more of the draft. Asynchronous on `stream`. */
int sd_specdec_set_row(sd_specdec* s, int b, int seq_len, int prev_tok, int last_tok,
                       int active, void* stream);

/* Sampled bonus token inside the step (generate_batch with do_sample=True: the reference
 * samples ONLY the token after the accepted prefix, pipeline.py:3140-3160 and :3351-3361;
 * drafting and verification stay greedy, :2400/:2645). With enable=1 the verify forward stores
 * its logits ([B][K+1][V] bf16 in logits_buf, caller-owned), and between the accept scan and the
 * state advance one sampler launch draws row b's token from the logits of position
 * accept_len[b] (sd_sample_token semantics; draw index = draw_counters[b], incremented per
 * sampled step; Philox stream = stream_ids[b] or b). enable=0 returns to the greedy step.
 * Either call drops the captured graph (it is re-captured on the next step). */
int sd_specdec_set_sampling(sd_specdec* s, int enable, float temperature, int top_k, float top_p,
                            uint64_t seed, void* logits_buf, size_t logits_bytes,
                            uint32_t* draw_counters, const int32_t* stream_ids);

/* Per-row adaptive K inside the captured step ("AdaptiveKController driving per-row K inside a
 * captured graph"; the rule is the reference's AdaptiveKController.get_k, src/specdec/policies/controllers.py:100-126,
 * which the reference applies to one K for the whole batch from the host, pipeline.py:1994-2014). The loop keeps the
 * SHAPE K it was created with (= max_k): every step still drafts and verifies K positions per row, but for row b only
 * the first k_row[b] proposals count — accept length = min(longest matching prefix, k_row[b]), emitted tokens and the
 * bonus token follow from that length exactly as in a step run at K = k_row[b] — and after the accept scan the device
 * applies the controller rule to the row's own counters (accepted / proposed so far, strict), so the next step can
 * be launched without a host round trip. History starts as [0.0] (the reference's first get_k reports rate 0.0), the
 * mean of the last four rates is compared with target_rate +- 0.1, K moves by step_size inside [min_k, max_k];
 * double arithmetic in the reference's order. Requires 1 <= min_k <= initial_k <= max_k <= K. The step record gains a
 * last int per row: the k that counted in that step (K when adaptive K is off). enable = 0 returns to fixed K.
 * Either call drops the captured graph. Not available with persistent Medusa heads / EAGLE-lite. */
int sd_specdec_set_adaptive(sd_specdec* s, int enable, int initial_k, int min_k, int max_k, int step_size,
                            double target_rate, void* stream);
/* (Re)write row b's controller state: a new sequence in the row's slot (k = initial_k, 0, 0, hist = NULL -> [0.0]), or
 * the host's in-order view after its rules overrode steps that had been launched ahead. `hist`: hist_n <= 4 rates,
 * oldest first. Asynchronous on `stream` (synchronises it first: one pinned staging slot per row). */
int sd_specdec_set_adaptive_row(sd_specdec* s, int b, int k, int accepted, int proposed, int hist_n,
                                const double* hist, void* stream);

/* EAGLE-lite drafting (the reference's _run_eagle_hf, src/specdec/core/pipeline.py:765-889, under greedy decoding), for
 * a loop created with draft = NULL: every step runs a 1-token target forward for the residual row of the last token,
 * h_t = final_norm(row); extrapolates K hidden rows h_1 = h_t + alpha (h_t - E), h_2 = h_1 + alpha (h_1 - h_t), ...
 * (E = the last extrapolated row of the previous step of that batch row; on a row's first step h_t stands in for it;
 * every operation rounds to bf16 as the reference's bf16 tensors do); scores all K rows with ONE lm_head launch
 * (d_i = argmax lm_head(h_i)); then verifies (last, d_1..d_K) as usual. K = min(k, eagle.max_draft) is the caller's
 * choice at sd_specdec_create. `workspace` (sd_specdec_eagle_bytes, caller-owned device memory, must outlive the loop)
 * holds the per-row state (at offsets independent of K: loops of different K may share one workspace sized for the
 * largest) and the extrapolated rows. Call before the first step, then sd_specdec_reset_eagle, which forgets the state
 * of all rows (start of a new set of sequences; enqueued on `stream`) — set_eagle itself does not touch the state. */
size_t sd_specdec_eagle_bytes(int B, int K, int d_model);
int sd_specdec_set_eagle(sd_specdec* s, float alpha, void* workspace, size_t workspace_bytes);
int sd_specdec_reset_eagle(sd_specdec* s, void* stream);

/* Persistent multi-head (Medusa) drafting for a loop created with draft = NULL: K heads, each a
 * vocabulary-sized matrix [V][d_model] packed by sd_pack_head in `weight_dtype`. After the accept scan of a step
 * the heads read the target's final-norm input row of the position that produced the last emitted token and
 * propose d_{i+1} = argmax head_i(norm(h)) for the NEXT step, no draft forwards. Heads that sit at a CONSTANT byte
 * stride (packed_heads[i+1] - packed_heads[i] equal for all i, e.g. one buffer of K slots) are evaluated by ONE launch
 * (one matrix per grid row over the same hidden rows) + one argmax finalize; otherwise one launch per head. Up to 9
 * rows read the hidden rows in place; more (within one verify pass: B*(K+1) <= 64) are gathered first and take the
 * multi-token kernel, one launch per head.
 * (Not in the reference: its Medusa mode re-creates random heads per call, pipeline.py:689-705)
 * Requires B*(K+1) within one verify pass. Drops the captured graph. */
int sd_specdec_set_medusa(sd_specdec* s, int n_heads, const void* const* packed_heads, int weight_dtype);

/* Enqueue ONE draft-then-verify step for all rows: K draft forwards (the first over
 * (prev,last), the rest over one token), one verify forward over (last,d_1..d_K),
 * the accept scan (wave ballot), the in-place state advance, and the copy of the
 * step record to pinned host memory. Draft work goes to stream_draft, verify to
 * stream_target, ordered by events; with use_graph=1 the whole step is captured
 * into a hipGraph on first use and replayed afterwards. */
int sd_specdec_step(sd_specdec* s, void* stream_target, void* stream_draft, int use_graph);

/* Steps launched so far, and a wait for ONE of the last two launches (0-based index) without draining the stream:
 * the host may keep a second step queued behind the running one (the device advances its own state) and still read
 * the record of the older one — slot (index & 1) of sd_specdec_record. */
long sd_specdec_launches(const sd_specdec* s);
int sd_specdec_wait(sd_specdec* s, long launch_index);

/* Drop the captured step (its kernels were chosen at capture time: after sd_model_set_persist_tokens /
 * sd_model_set_length_hint changed a model's path, or a model was re-bound). The next sd_specdec_step runs eagerly, the
 * one after it captures again. The caller has drained the loop's streams. */
int sd_specdec_invalidate(sd_specdec* s);

/* Block until everything enqueued on `stream` is done (hipStreamSynchronize). */
int sd_specdec_sync(sd_specdec* s, void* stream);

/* Pinned host records, written by the device at the end of each step: TWO slots of [B][record_ints], the
 * record of launch i (0-based, sd_specdec_launches) is in slot i & 1. Ints per row:
 *   [0] accept_len  [1] n_new  [2] cur_len after the step
 *   [3 .. 3+K]            emitted tokens (n_new valid, -1 padded)
 *   [4+K .. 3+2K]         draft tokens d_1..d_K
 *   [4+2K .. 4+3K]        target argmax t_0..t_K
 *   [5+3K]                proposals that counted for the row in this step (K unless sd_specdec_set_adaptive)
 *   [6+3K]                health word of the models' persistent launches (sd_model_engine_status), 0 = all completed
 * row stride = sd_specdec_record_ints(). */
const int32_t* sd_specdec_record(const sd_specdec* s);
int sd_specdec_record_ints(const sd_specdec* s);

#ifdef __cplusplus
}
#endif
#endif /* SPECDEC_HIP_H */
