// Decoder forward orchestration and the draft-then-verify step for gfx950.
//
// Host-side C++ behind the C-ABI (include/specdec_hip.h). It enqueues the kernels
// of gemv.hip / attention.hip / misc.hip on caller-provided HIP streams, never
// synchronises, and is graph-capturable: all per-row dynamic quantities (lengths,
// tokens) live in device memory and are read by the kernels.
//
// What it replaces in the reference: HFWrapper._generate_tokens_async
// (hf_wrappers.py:272-627: one HF forward + argmax per generated token, the whole
// prefix re-fed unless a KV cache is threaded through) and the device-facing half of
// the step loop of SpeculativePipeline.generate_batch (pipeline.py:2306-2846: draft
// stream / verify stream / event waits). The reference verifies by letting the base
// model generate K tokens autoregressively (speculative_scheduler.py:192-199); here
// the target scores all K+1 positions in ONE forward over (last, d_1..d_K), which
// under greedy decoding yields the same accepted prefix and bonus token.

#include <hip/hip_runtime.h>

#include <stdlib.h>

#include <new>
#include <vector>

#include "engine.h"
#include "persist.h"
#include "prefill_gemm.h"

struct sd_model {
  sd_model_config cfg;
  std::vector<sd_layer_weights> layers;
  // bound storage
  uint16_t* k_cache = nullptr;
  uint16_t* v_cache = nullptr;
  int B = 0, Lmax = 0;
  // paged KV (sd_model_bind_paged): k_cache / v_cache are page pools of n_pages pages per layer, Lmax = max_pages * page_len
  const int32_t* block_table = nullptr;   // device [B][max_pages], caller-owned and caller-maintained
  int page_shift = 0, max_pages = 0, n_pages = 0;
  // workspace carve
  uint16_t* x = nullptr;     // [64][d]
  uint16_t* q = nullptr;     // [64][Hq*D]
  uint16_t* attn = nullptr;  // [64][Hq*D]
  uint16_t* act = nullptr;   // [64][ff]
  int32_t* probe_pos = nullptr;  // [1] zero: position base of the QKV probe
  float* attn_ws = nullptr;      // split-KV partial tiles (attention.hip)
  unsigned* attn_cnt = nullptr;  // arrival counters, zero between launches
  float* xstat = nullptr;    // [2][128][256]: row statistics handed from an EPI_RESID launch to the next norm-fused one (> 9 tokens)
  float* part_val = nullptr; // [64][512]
  int small_t = sd::kGemvMaxT; // tokens per pass of gemv.hip for this model's widest activation row (<= 9)
  int max_t = sd::kGemvMaxT; // tokens per pass: 64 when every matrix of the model is covered by gemm_skinny.hip
  int* part_idx = nullptr;
  int head_grid = 0;         // grid of the last lm_head launch (partials per token)
  const int32_t* skip_k = nullptr;   // set around a draft forward of the adaptive step (enqueue_step): every launch of the
  int skip_i = 0;                    // forward returns at entry when *skip_k <= skip_i
  std::vector<const void*> packed;  // per matrix (4 per layer + lm_head) or empty: row-major weights
  std::vector<const float*> scales; // fp8 storage: fp32 row scales per matrix
  const void* mat(int index, const void* row_major) const { return packed.empty() ? row_major : packed[index]; }
  int is_packed() const { return packed.empty() ? 0 : 1; }
  int w8() const { return scales.empty() ? 0 : 1; }
  const float* scale(int index) const { return scales.empty() ? nullptr : scales[index]; }
  // persistent forward (csrc/persist.hip): passes of <= persist_t tokens run as ONE launch
  int persist_t = 0;                       // 0 = not available for this model / device
  bool persist_taps = true;                // stage rows written by persistent passes (sd_specdec_create turns them off for its draft)
  sd::PersistOp* p_ops = nullptr;          // device: 4 per layer + lm_head, stream order
  unsigned long long* p_gran = nullptr;    // granule buffers, two parities
  unsigned p_gran_parity = 0;
  unsigned* p_sync = nullptr;              // [0] launch counter, [1] status
  unsigned long long* p_debug = nullptr;   // optional timeline (sd_model_probe_persist)
  void* prefill_ws = nullptr;              // lazily allocated workspace of the GEMM prefill path (csrc/prefill_gemm.hip)
  unsigned* host_status = nullptr;         // pinned host word: a persistent launch that gives up stores its reason here as well
  int persist_cap = 0;                     // tokens per persistent pass this model / device / cache can take (0: none)
  int len_hint = 0;                        // caller's bound on the rows' current lengths (sd_model_set_length_hint; default Lmax)
  int ctx_limit = 0;                       // longest rows the persistent launch serves (kPersistMaxCtx; lifted by SPECDEC_PERSIST_MAX_T)
  ~sd_model() {
    if (host_status) (void)hipHostFree(host_status);
    if (prefill_ws) (void)hipFree(prefill_ws);
  }
};

namespace sd {
// One CU (three waves) walks a head's whole cache in the persistent launch, where the launch path splits long rows over up to 32
// workgroups — 1B dimensions, 1 token, us per forward persistent / launches at 128 ... 4096 cached positions: 588 / 675,
// 605 / 694, 651 / 721, 723 / 764, 888 / 767, 1186 / 777 (profiles/round3_persist_ab.md): the persistent launch serves rows of
// up to this many positions (where the two lines cross: 723 + 0.161 (L - 1024) against 764 + 0.003 (L - 1024) us at L = 1283;
// round 4's context sweep: 5.07 ms per step on the persistent launch at 1400-1536 positions against 4.96 on the launch path at 2048).
constexpr int kPersistMaxCtx = 1280;
static bool persist_pass_ok(const sd_model* m, int T, int Bc, int Mc) {
  return m->persist_t > 0 && T <= m->persist_t && Mc <= 8 && !m->block_table && Bc * m->cfg.n_heads <= kPersistCUs && m->len_hint <= m->ctx_limit;
}
}  // namespace sd

namespace sd {

constexpr int kMaxPartials = 512;

static size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

static size_t workspace_bytes(const sd_model_config& c) {
  const size_t T = kSkinnyMaxT;
  size_t n = 0;
  n += align_up(T * c.d_model * 2, 256);
  n += align_up(T * c.n_heads * c.head_dim * 2, 256) * 2;
  n += align_up(T * c.d_ff * 2, 256);
  n += align_up(T * kMaxPartials * 4, 256) * 2;
  n += align_up(attention_split_ws_bytes(c.head_dim), 256);
  n += align_up(sizeof(float) * 2 * kStatPlane, 256);
  n += align_up(persist_workspace_bytes(c), 256);
  return n + 256;
}

// one pass: Bc rows x Mc tokens, Bc*Mc <= m->max_t
static int forward_pass(sd_model* m, const int32_t* tokens, int tok_stride, const int32_t* pos_base,
                        int pos_off, int row0, int b0, int Bc, int Mc, int32_t* ids_out, int ids_stride,
                        void* logits_out, int logits_dtype, int logits_stride, int skip_head,
                        hipStream_t st) {
  const sd_model_config& c = m->cfg;
  const int T = Bc * Mc;
  const int d = c.d_model, Hq = c.n_heads, Hkv = c.n_kv_heads, D = c.head_dim, ff = c.d_ff;
  const bool llama = (c.arch == SD_ARCH_LLAMA);
  const int pro = llama ? PRO_RMSNORM : PRO_LAYERNORM;

  // ---- the whole pass as ONE persistent launch (csrc/persist.hip): small passes of a dense-KV Llama model
  if (persist_pass_ok(m, T, Bc, Mc)) {
    PersistArgs pa{};
    pa.ops = m->p_ops;
    pa.n_ops = 4 * c.n_layers + (skip_head ? 0 : 1);
    pa.d_model = d; pa.n_q_heads = Hq; pa.n_kv_heads = Hkv; pa.head_dim = D; pa.d_ff = ff; pa.vocab = c.vocab;
    pa.n_layers = c.n_layers; pa.max_pos = c.max_pos;
    pa.norm_eps = c.norm_eps;
    pa.attn_scale = 1.0f / sqrtf(static_cast<float>(D));
    pa.tok_emb = c.tok_emb;
    pa.rope_cos = c.rope_cos;
    pa.rope_sin = c.rope_sin;
    pa.tokens = tokens + static_cast<size_t>(b0) * tok_stride;
    pa.tok_stride = tok_stride;
    pa.pos_base = pos_base + b0;
    pa.pos_off = pos_off;
    pa.B = Bc;
    pa.M = Mc;
    const size_t row_off = static_cast<size_t>(row0 + b0) * Hkv * m->Lmax * D;
    pa.k_cache = m->k_cache + row_off;
    pa.v_cache = m->v_cache + row_off;
    pa.layer_kv = static_cast<size_t>(m->B) * Hkv * m->Lmax * D;
    pa.l_max = m->Lmax;
    pa.logits = logits_out;
    pa.logits_dtype = logits_dtype;
    pa.logits_stride = logits_stride;
    pa.part_val = m->part_val;
    pa.part_idx = m->part_idx;
    pa.x = m->x; pa.q = m->q; pa.attn = m->attn; pa.act = m->act;
    pa.gran = m->p_gran;
    pa.gran_parity = m->p_gran_parity;
    pa.sync = m->p_sync;
    pa.host_status = m->host_status;
    pa.skip_k = m->skip_k;
    pa.skip_i = m->skip_i;
    pa.taps = (m->persist_taps || skip_head) ? 1 : 0;   // (a pass without the head is run FOR its hidden rows)
    pa.debug_ts = m->p_debug;
    if (int rc = launch_persist_forward(pa, st)) return rc;
    if (skip_head) return 0;
    m->head_grid = kPersistCUs;
    if (ids_out) {
      if (int rc = launch_argmax_finalize(m->part_val, m->part_idx, T, m->head_grid, Mc, ids_stride,
                                          ids_out + static_cast<size_t>(b0) * ids_stride, st, m->skip_k, m->skip_i))
        return rc;
    }
    return 0;
  }

  EmbedArgs e{};
  e.tok_emb = c.tok_emb;
  e.pos_emb = llama ? nullptr : c.pos_emb;
  e.tokens = tokens + static_cast<size_t>(b0) * tok_stride;
  e.tok_stride = tok_stride;
  e.pos_base = pos_base + b0;
  e.pos_off = pos_off;
  e.M = Mc;
  e.T = T;
  e.d = d;
  e.vocab = c.vocab;
  e.max_pos = c.max_pos;
  e.x = m->x;
  e.skip_k = m->skip_k;
  e.skip_i = m->skip_i;
  if (int rc = launch_embed(e, st)) return rc;

  const bool paged = m->block_table != nullptr;
  const size_t layer_kv = paged ? (static_cast<size_t>(m->n_pages) * Hkv * D << m->page_shift) : static_cast<size_t>(m->B) * Hkv * m->Lmax * D;
  const size_t row_kv = paged ? 0 : static_cast<size_t>(row0 + b0) * Hkv * m->Lmax * D;    // paged: rows are found through the table
  const int32_t* bt = paged ? m->block_table + static_cast<size_t>(row0 + b0) * m->max_pages : nullptr;
  // > 9 tokens: the launch that writes the residual stream hands its row statistics to the launch that normalises it
  const float* stat_in = nullptr;
  int stat_n = 0;
  auto publish = [&](GemvArgs& prod) {      // prod: an EPI_RESID launch about to be issued
    stat_in = nullptr;
    if (m->xstat && gemm_resid_publishes_stats(prod)) {
      prod.xstat_out = m->xstat;
      stat_in = m->xstat;
      int ppw = 1;
      stat_n = gemv_grid(prod, &ppw);
    }
  };
  for (int l = 0; l < c.n_layers; ++l) {
    const sd_layer_weights& w = m->layers[l];
    uint16_t* kc = m->k_cache + l * layer_kv + row_kv;
    uint16_t* vc = m->v_cache + l * layer_kv + row_kv;

    GemvArgs g{};
    g.T = T;
    g.M = Mc;
    g.pos_base = pos_base + b0;
    g.pos_off = pos_off;
    g.head_dim = D;
    g.n_q_heads = Hq;
    g.n_kv_heads = Hkv;
    g.max_pos = c.max_pos;
    g.l_max = m->Lmax;
    g.out_dtype = SD_BF16;
    g.w8 = m->w8();
    g.block_table = bt;
    g.page_shift = m->page_shift;
    g.skip_k = m->skip_k;
    g.skip_i = m->skip_i;

    // 1. norm + QKV projection + RoPE + in-place KV append
    g.packed = m->is_packed();
    GemvArgs a1 = g;
    a1.W = m->mat(4 * l + 0, w.wqkv);
    a1.w_scale = m->scale(4 * l + 0);
    a1.bias = w.bqkv;
    a1.N = (Hq + 2 * Hkv) * D;
    a1.K = d;
    a1.n_pairs = a1.N / 2;
    a1.x = m->x;
    a1.x_stride = d;
    a1.prologue = pro;
    a1.norm_w = w.attn_norm_w;
    a1.norm_b = w.attn_norm_b;
    a1.norm_eps = c.norm_eps;
    a1.out = m->q;
    a1.out_stride = Hq * D;
    a1.rope_cos = llama ? c.rope_cos : nullptr;
    a1.rope_sin = llama ? c.rope_sin : nullptr;
    a1.k_cache = kc;
    a1.v_cache = vc;
    a1.xstat_in = stat_in;
    a1.xstat_n = stat_n;
    if (int rc = launch_gemv(a1, EPI_QKV_ROPE, st)) return rc;

    // 2. attention of the Mc new positions over the appended cache
    AttnArgs at{};
    at.q = m->q;
    at.k_cache = kc;
    at.v_cache = vc;
    at.out = m->attn;
    at.pos_base = pos_base + b0;
    at.pos_off = pos_off;
    at.B = Bc;
    at.M = Mc;
    at.n_q_heads = Hq;
    at.n_kv_heads = Hkv;
    at.head_dim = D;
    at.l_max = m->Lmax;
    at.scale = 1.0f / sqrtf(static_cast<float>(D));
    at.split_ws = m->attn_ws;
    at.split_cnt = m->attn_cnt;
    at.split_slots = kAttnSplitSlots;
    at.block_table = bt;
    at.page_shift = m->page_shift;
    at.skip_k = m->skip_k;
    at.skip_i = m->skip_i;
    if (int rc = launch_attention(at, st)) return rc;

    // 3. output projection + residual
    GemvArgs a3 = g;
    a3.W = m->mat(4 * l + 1, w.wo);
    a3.w_scale = m->scale(4 * l + 1);
    a3.bias = w.bo;
    a3.N = d;
    a3.K = Hq * D;
    a3.n_pairs = d / 2;
    a3.x = m->attn;
    a3.x_stride = Hq * D;
    a3.prologue = PRO_NONE;
    a3.out = m->x;
    a3.out_stride = d;

    // 4. norm + up projection (+ gate) + activation
    GemvArgs a4 = g;
    a4.W = m->mat(4 * l + 2, w.w_up);
    a4.w_scale = m->scale(4 * l + 2);
    a4.bias = w.b_up;
    a4.K = d;
    a4.x = m->x;
    a4.x_stride = d;
    a4.prologue = pro;
    a4.norm_w = w.mlp_norm_w;
    a4.norm_b = w.mlp_norm_b;
    a4.norm_eps = c.norm_eps;
    a4.out = m->act;
    a4.out_stride = ff;
    publish(a3);
    a4.xstat_in = stat_in;
    a4.xstat_n = stat_n;
    if (llama) {
      a4.N = 2 * ff;
      a4.n_pairs = ff;
      if (int rc = launch_gemv(a3, EPI_RESID, st)) return rc;
      if (int rc = launch_gemv(a4, EPI_SWIGLU, st)) return rc;
    } else {
      a4.N = ff;
      a4.n_pairs = ff / 2;
      if (int rc = launch_gemv(a3, EPI_RESID, st)) return rc;
      if (int rc = launch_gemv(a4, EPI_GELU, st)) return rc;
    }

    // 5. down projection + residual
    GemvArgs a5 = g;
    a5.W = m->mat(4 * l + 3, w.w_down);
    a5.w_scale = m->scale(4 * l + 3);
    a5.bias = w.b_down;
    a5.N = d;
    a5.K = ff;
    a5.n_pairs = d / 2;
    a5.x = m->act;
    a5.x_stride = ff;
    a5.prologue = PRO_NONE;
    a5.out = m->x;
    a5.out_stride = d;
    publish(a5);
    if (int rc = launch_gemv(a5, EPI_RESID, st)) return rc;
  }
  if (skip_head) return 0;

  // final norm + lm_head with the argmax fused into the epilogue
  GemvArgs h{};
  h.packed = m->is_packed();
  h.w8 = m->w8();
  h.w_scale = m->scale(4 * c.n_layers);
  h.W = m->mat(4 * c.n_layers, c.lm_head);
  h.N = c.vocab;
  h.K = d;
  h.n_pairs = (c.vocab + 1) / 2;
  h.x = m->x;
  h.x_stride = d;
  h.T = T;
  h.M = Mc;
  h.prologue = pro;
  h.norm_w = c.final_norm_w;
  h.norm_b = c.final_norm_b;
  h.norm_eps = c.norm_eps;
  h.out = logits_out;
  h.out_stride = logits_stride;
  h.out_dtype = logits_dtype;
  h.part_val = m->part_val;
  h.part_idx = m->part_idx;
  h.xstat_in = stat_in;
  h.xstat_n = stat_n;
  h.skip_k = m->skip_k;
  h.skip_i = m->skip_i;
  int ks = 1;
  m->head_grid = gemv_grid(h, &ks);
  if (int rc = launch_gemv(h, EPI_ARGMAX, st)) return rc;
  if (ids_out) {
    if (int rc = launch_argmax_finalize(m->part_val, m->part_idx, T, m->head_grid, Mc, ids_stride,
                                        ids_out + static_cast<size_t>(b0) * ids_stride, st, m->skip_k, m->skip_i))
      return rc;
  }
  return 0;
}

// final norm + lm_head + fused argmax over n <= 128 residual rows (bf16 [n][d_model]) -> ids[0..n)
static int head_pass(sd_model* m, const uint16_t* xrows, int n, int32_t* ids, hipStream_t st) {
  const sd_model_config& c = m->cfg;
  GemvArgs h{};
  h.packed = m->is_packed();
  h.w8 = m->w8();
  h.w_scale = m->scale(4 * c.n_layers);
  h.W = m->mat(4 * c.n_layers, c.lm_head);
  h.N = c.vocab;
  h.K = c.d_model;
  h.n_pairs = (c.vocab + 1) / 2;
  h.x = xrows;
  h.x_stride = c.d_model;
  h.T = n;
  h.M = n;
  h.prologue = (c.arch == SD_ARCH_LLAMA) ? PRO_RMSNORM : PRO_LAYERNORM;
  h.norm_w = c.final_norm_w;
  h.norm_b = c.final_norm_b;
  h.norm_eps = c.norm_eps;
  h.out = nullptr;
  h.out_dtype = SD_BF16;
  h.part_val = m->part_val;
  h.part_idx = m->part_idx;
  int ks = 1;
  m->head_grid = gemv_grid(h, &ks);
  if (int rc = launch_gemv(h, EPI_ARGMAX, st)) return rc;
  return launch_argmax_finalize(m->part_val, m->part_idx, n, m->head_grid, n, n, ids, st);
}

// rows [row0, row0+B) of the bound batch; tokens / pos_base / ids_out / logits_out are
// indexed from row0 (element 0 of each array belongs to row row0)
static int model_forward(sd_model* m, const int32_t* tokens, int tok_stride, const int32_t* pos_base,
                         int pos_off, int row0, int B, int M, int32_t* ids_out, int ids_stride, void* logits_out,
                         int logits_dtype, int skip_head, hipStream_t st) {
  SD_REQUIRE(m && m->k_cache && m->x, "forward: model not bound (sd_model_bind)");
  SD_REQUIRE(row0 >= 0 && B >= 1 && row0 + B <= m->B, "forward: rows [%d,%d) exceeds bound batch %d", row0, row0 + B, m->B);
  SD_REQUIRE(M >= 1, "forward: M=%d", M);
  SD_REQUIRE(tokens && pos_base, "forward: NULL tokens/pos_base");
  const int esz = (logits_dtype == SD_F32) ? 4 : 2;
  SD_REQUIRE(!logits_out || logits_dtype == SD_F32 || logits_dtype == SD_BF16, "forward: logits dtype %d", logits_dtype);
  const int V = m->cfg.vocab;
  const int cap = (B * M <= m->small_t) ? m->small_t : m->max_t;  // tokens per pass
  // A prompt (M >= 96 positions per row, whatever the pass size of the decode-shaped kernels). A prompt of a Llama model with bf16 row-major weights and dense KV is absorbed as GEMMs (csrc/prefill_gemm.hip:
  // <= 512 positions per chunk, every matrix product one library GEMM, this repo's norm / epilogue / attention kernels around them) when
  // the caller wants no logits of the prompt positions (skip_head, or ids only — the head then runs over the LAST chunk's rows below).
  static const int prefill_min = getenv(debug_env::kPrefillMinTokens) ? atoi(getenv(debug_env::kPrefillMinTokens)) : kPrefillMinTokens;
  if (M >= prefill_min && m->cfg.arch == SD_ARCH_LLAMA && !m->w8() && !m->block_table && !logits_out && m->cfg.weight_dtype == SD_BF16 &&
      !getenv(debug_env::kNoGemmPrefill) && !m->skip_k && prefill_gemm_available()) {
    hipStreamCaptureStatus cap_st = hipStreamCaptureStatusNone;
    (void)hipStreamIsCapturing(st, &cap_st);
    if (cap_st == hipStreamCaptureStatusNone) {
      if (!m->prefill_ws) SD_HIP_CHECK(hipMalloc(&m->prefill_ws, prefill_gemm_workspace_bytes(m->cfg)));
      PrefillModel pm{&m->cfg, m->k_cache, m->v_cache, m->B, m->Lmax, m->attn_ws, m->attn_cnt};
      for (int b0 = 0; b0 < B; ++b0) {
        for (int m0 = 0; m0 < M; m0 += kPrefillChunk) {
          const int mc = (M - m0 < kPrefillChunk) ? M - m0 : kPrefillChunk;
          uint16_t* xr = nullptr;
          if (int rc = prefill_gemm_chunk(pm, tokens + static_cast<size_t>(b0) * tok_stride + m0, pos_base + b0, pos_off + m0, row0 + b0, mc,
                                          m->prefill_ws, &xr, st))
            return rc;
          // the residual rows of the chunk's last <= 128 positions go where every other pass leaves them (hidden rows, the head)
          const int keep = mc < 128 ? mc : 128;
          SD_HIP_CHECK(hipMemcpyAsync(m->x, xr + static_cast<size_t>(mc - keep) * m->cfg.d_model, static_cast<size_t>(keep) * m->cfg.d_model * 2,
                                      hipMemcpyDeviceToDevice, st));
          if (!skip_head && ids_out) {
            // ids of the prompt positions: the lm_head over the chunk's rows in groups of <= 128 (the decode-shaped head kernel)
            for (int s0 = 0; s0 < mc; s0 += 128) {
              const int n = (mc - s0 < 128) ? mc - s0 : 128;
              if (int rc = head_pass(m, xr + static_cast<size_t>(s0) * m->cfg.d_model, n, ids_out + static_cast<size_t>(b0) * ids_stride + m0 + s0, st)) return rc;
            }
          }
        }
      }
      return 0;
    }
  }
  if (M <= cap) {
    const int Bc = cap / M;
    for (int b0 = 0; b0 < B; b0 += Bc) {
      const int nb = (B - b0 < Bc) ? B - b0 : Bc;
      void* lo = logits_out ? static_cast<char*>(logits_out) + static_cast<size_t>(b0) * M * V * esz : nullptr;
      if (int rc = forward_pass(m, tokens, tok_stride, pos_base, pos_off, row0, b0, nb, M, ids_out, ids_stride, lo,
                                logits_dtype, V, skip_head, st))
        return rc;
    }
    return 0;
  }
  // long M (prefill): chunks of `cap` positions, one row at a time, in position order
  for (int b0 = 0; b0 < B; ++b0) {
    for (int m0 = 0; m0 < M; m0 += cap) {
      const int mc = (M - m0 < cap) ? M - m0 : cap;
      void* lo = logits_out ? static_cast<char*>(logits_out) + (static_cast<size_t>(b0) * M + m0) * V * esz : nullptr;
      if (int rc = forward_pass(m, tokens + m0, tok_stride, pos_base, pos_off + m0, row0, b0, 1, mc,
                                ids_out ? ids_out + m0 : nullptr, ids_stride, lo, logits_dtype, V, skip_head, st))
        return rc;
    }
  }
  return 0;
}

}  // namespace sd

// ============================================================================ C-ABI
using namespace sd;

extern "C" int sd_model_create(const sd_model_config* cfg, sd_model** out) {
  clear_error();
  SD_REQUIRE(cfg && out, "model_create: NULL argument");
  SD_REQUIRE(cfg->arch == SD_ARCH_LLAMA || cfg->arch == SD_ARCH_GPT2, "model_create: arch %d", cfg->arch);
  SD_REQUIRE(cfg->weight_dtype == SD_BF16 || cfg->weight_dtype == SD_FP8_E4M3,
             "model_create: weight_dtype %d (SD_BF16, or SD_FP8_E4M3 = stream the fp8 copy made by sd_pack_weights)", cfg->weight_dtype);
  SD_REQUIRE(cfg->weight_dtype != SD_FP8_E4M3 || cfg->packed, "model_create: fp8 storage needs the packed buffer of sd_pack_weights");
  SD_REQUIRE(cfg->n_layers > 0 && cfg->d_model > 0 && cfg->n_heads > 0 && cfg->n_kv_heads > 0 &&
                 cfg->head_dim > 0 && cfg->d_ff > 0 && cfg->vocab > 0 && cfg->max_pos > 0,
             "model_create: non-positive dimension");
  SD_REQUIRE(cfg->d_model % 8 == 0 && cfg->d_ff % 8 == 0 && (cfg->n_heads * cfg->head_dim) % 8 == 0,
             "model_create: d_model, d_ff and Hq*D must be multiples of 8");
  SD_REQUIRE(cfg->head_dim % 2 == 0 && cfg->n_heads % cfg->n_kv_heads == 0, "model_create: head layout");
  // the kernels' argument blocks carry these in 8-bit fields (GemvArgs / AttnArgs), and the attention kernel is instantiated
  // for these head sizes only: a direct C caller gets an error, not a truncated geometry
  SD_REQUIRE(cfg->n_heads <= 255 && cfg->n_kv_heads <= 255, "model_create: at most 255 heads (got %d / %d kv)", cfg->n_heads, cfg->n_kv_heads);
  SD_REQUIRE(cfg->head_dim == 32 || cfg->head_dim == 64 || cfg->head_dim == 128, "model_create: head_dim %d (32, 64 or 128)", cfg->head_dim);
  SD_REQUIRE(cfg->tok_emb && cfg->lm_head && cfg->final_norm_w && cfg->layers, "model_create: NULL weights");
  if (cfg->arch == SD_ARCH_LLAMA) SD_REQUIRE(cfg->rope_cos && cfg->rope_sin, "model_create: Llama needs rope tables");
  if (cfg->arch == SD_ARCH_GPT2) SD_REQUIRE(cfg->pos_emb && cfg->final_norm_b, "model_create: GPT-2 needs pos_emb and ln_f bias");
  sd_model* m = new (std::nothrow) sd_model();
  SD_REQUIRE(m, "model_create: out of memory");
  m->cfg = *cfg;
  m->layers.assign(cfg->layers, cfg->layers + cfg->n_layers);
  m->cfg.layers = m->layers.data();
  for (int l = 0; l < cfg->n_layers; ++l) {
    const sd_layer_weights& w = m->layers[l];
    if (!(w.attn_norm_w && w.wqkv && w.wo && w.mlp_norm_w && w.w_up && w.w_down)) {
      delete m;
      SD_REQUIRE(false, "model_create: layer %d has NULL weights", l);
    }
  }
  {
    // 64-token passes need every matrix of the model to be a shape gemm_skinny.hip covers
    const sd_model_config& c = m->cfg;
    const bool llama = (c.arch == SD_ARCH_LLAMA);
    const int HqD = c.n_heads * c.head_dim;
    const bool w8 = cfg->weight_dtype == SD_FP8_E4M3;
    auto covers = [&](int T) {
      return gemm_skinny_covers(T, (c.n_heads + 2 * c.n_kv_heads) * c.head_dim / 2, c.d_model, w8) &&
             gemm_skinny_covers(T, c.d_model / 2, HqD, w8) && gemm_skinny_covers(T, llama ? c.d_ff : c.d_ff / 2, c.d_model, w8) &&
             gemm_skinny_covers(T, c.d_model / 2, c.d_ff, w8) && gemm_skinny_covers(T, (c.vocab + 1) / 2, c.d_model, w8);
    };
    // the most tokens every matrix of the model can take in one pass: 128, 64 (x chunks of a pass must fit the LDS) ...
    int cover_t = 0;
    for (int T = kSkinnyMaxT; T >= 16 && !cover_t; T >>= 1)
      if (covers(T)) cover_t = T;
    const bool ok = cover_t != 0;
    const char* env = getenv(debug_env::kMaxPassTokens);  // testing knob: 9 forces the small-T kernel everywhere
    int want = env ? atoi(env) : cover_t;
    if (want > cover_t) want = cover_t;
    int kmax = c.d_model > HqD ? c.d_model : HqD;
    if (c.d_ff > kmax) kmax = c.d_ff;
    m->small_t = gemv_max_tokens(kmax);  // e.g. 5 for d_ff = 14336: x rows must fit the CU's LDS
    if (want < m->small_t) want = m->small_t;
    m->max_t = ok ? want : m->small_t;
  }
  if (cfg->weight_dtype == SD_FP8_E4M3) {
    // every matrix must split into whole 64-k steps per K slice
    const sd_model_config& c = m->cfg;
    const bool llama = (c.arch == SD_ARCH_LLAMA);
    const int shapes[5][2] = {{(c.n_heads + 2 * c.n_kv_heads) * c.head_dim / 2, c.d_model}, {c.d_model / 2, c.n_heads * c.head_dim},
                              {llama ? c.d_ff : c.d_ff / 2, c.d_model}, {c.d_model / 2, c.d_ff}, {(c.vocab + 1) / 2, c.d_model}};
    for (const auto& sh : shapes) {
      const GemvGeom q = gemv_geometry(sh[0], sh[1]);
      if (sh[1] % 64 != 0 || q.kw % 64 != 0 || q.kw * q.ksplit != sh[1]) {
        delete m;
        SD_REQUIRE(false, "model_create: fp8 storage does not cover a matrix with K=%d (K slice %d)", sh[1], q.kw);
      }
    }
  }
  if (cfg->packed) {
    const char* base = static_cast<const char*>(cfg->packed);
    for (int i = 0; i <= 4 * cfg->n_layers; ++i) {
      m->packed.push_back(base + packed_offset(m->cfg, i));
      if (cfg->weight_dtype == SD_FP8_E4M3)
        m->scales.push_back(reinterpret_cast<const float*>(base + packed_offset(m->cfg, i) + packed_scale_offset(m->cfg, i)));
    }
  }
  *out = m;
  return 0;
}

extern "C" int sd_model_destroy(sd_model* m) {
  delete m;
  return 0;
}

extern "C" int sd_model_pass_tokens(const sd_model* m) { return m ? m->max_t : 0; }

extern "C" size_t sd_model_workspace_bytes(const sd_model* m) { return m ? workspace_bytes(m->cfg) : 0; }

extern "C" size_t sd_model_kv_bytes(const sd_model* m, int B, int Lmax) {
  if (!m || B <= 0 || Lmax <= 0) return 0;
  return static_cast<size_t>(m->cfg.n_layers) * B * m->cfg.n_kv_heads * Lmax * m->cfg.head_dim * 2;
}

static int carve_workspace(sd_model* m, void* workspace);

extern "C" int sd_model_bind(sd_model* m, void* k_cache, void* v_cache, int B, int Lmax, void* workspace,
                             size_t workspace_bytes_) {
  clear_error();
  SD_REQUIRE(m && k_cache && v_cache && workspace, "model_bind: NULL argument");
  SD_REQUIRE(B >= 1 && Lmax >= 1, "model_bind: B=%d Lmax=%d", B, Lmax);
  SD_REQUIRE(workspace_bytes_ >= workspace_bytes(m->cfg), "model_bind: workspace too small");
  SD_REQUIRE((reinterpret_cast<uintptr_t>(k_cache) & 15) == 0 && (reinterpret_cast<uintptr_t>(v_cache) & 15) == 0,
             "model_bind: caches must be 16-byte aligned");
  m->k_cache = static_cast<uint16_t*>(k_cache);
  m->v_cache = static_cast<uint16_t*>(v_cache);
  m->B = B;
  m->Lmax = Lmax;
  m->block_table = nullptr;
  m->page_shift = m->max_pages = m->n_pages = 0;
  return carve_workspace(m, workspace);
}

extern "C" size_t sd_model_kv_pool_bytes(const sd_model* m, int n_pages, int page_len) {
  if (!m || n_pages <= 0 || page_len <= 0) return 0;
  return static_cast<size_t>(m->cfg.n_layers) * n_pages * m->cfg.n_kv_heads * page_len * m->cfg.head_dim * 2;
}

// Paged KV: rows do not own Lmax positions each; they own pages of page_len positions out of a pool shared by all
// rows, through a caller-maintained table (the counterpart of the reference's cache manager growing / realigning
// per-sequence tensors, kv_cache_manager.py:194-199, :353-479 — here a row grows by a table entry).
extern "C" int sd_model_bind_paged(sd_model* m, void* k_pool, void* v_pool, int n_pages, int page_len, const int32_t* block_table,
                                   int max_pages_per_row, int B, void* workspace, size_t workspace_bytes_) {
  clear_error();
  SD_REQUIRE(m && k_pool && v_pool && block_table && workspace, "model_bind_paged: NULL argument");
  SD_REQUIRE(B >= 1 && n_pages >= 1 && max_pages_per_row >= 1, "model_bind_paged: B=%d pages=%d max_pages=%d", B, n_pages, max_pages_per_row);
  int shift = 0;
  while ((1 << shift) < page_len) ++shift;
  SD_REQUIRE((1 << shift) == page_len && page_len >= 32 && page_len <= 65536, "model_bind_paged: page_len %d must be a power of two >= 32", page_len);
  SD_REQUIRE(workspace_bytes_ >= workspace_bytes(m->cfg), "model_bind_paged: workspace too small");
  SD_REQUIRE((reinterpret_cast<uintptr_t>(k_pool) & 15) == 0 && (reinterpret_cast<uintptr_t>(v_pool) & 15) == 0,
             "model_bind_paged: pools must be 16-byte aligned");
  m->k_cache = static_cast<uint16_t*>(k_pool);
  m->v_cache = static_cast<uint16_t*>(v_pool);
  m->B = B;
  m->Lmax = max_pages_per_row * page_len;
  m->block_table = block_table;
  m->page_shift = shift;
  m->max_pages = max_pages_per_row;
  m->n_pages = n_pages;
  return carve_workspace(m, workspace);
}

static int carve_workspace(sd_model* m, void* workspace) {
  const sd_model_config& c = m->cfg;
  const size_t T = kSkinnyMaxT;
  char* p = reinterpret_cast<char*>(align_up(reinterpret_cast<uintptr_t>(workspace), 256));
  m->x = reinterpret_cast<uint16_t*>(p);
  p += align_up(T * c.d_model * 2, 256);
  m->q = reinterpret_cast<uint16_t*>(p);
  p += align_up(T * c.n_heads * c.head_dim * 2, 256);
  m->attn = reinterpret_cast<uint16_t*>(p);
  p += align_up(T * c.n_heads * c.head_dim * 2, 256);
  m->act = reinterpret_cast<uint16_t*>(p);
  p += align_up(T * c.d_ff * 2, 256);
  m->part_val = reinterpret_cast<float*>(p);
  p += align_up(T * kMaxPartials * 4, 256);
  m->part_idx = reinterpret_cast<int*>(p);
  p += align_up(T * kMaxPartials * 4, 256);
  m->attn_ws = reinterpret_cast<float*>(p);
  m->attn_cnt = reinterpret_cast<unsigned*>(p + static_cast<size_t>(kAttnSplitSlots) * 16 * (c.head_dim + 2) * sizeof(float));
  p += align_up(attention_split_ws_bytes(c.head_dim), 256);
  m->xstat = reinterpret_cast<float*>(p);
  p += align_up(sizeof(float) * 2 * kStatPlane, 256);
  SD_HIP_CHECK(hipMemset(m->attn_cnt, 0, kAttnSplitSlots * sizeof(unsigned)));
  m->probe_pos = reinterpret_cast<int32_t*>(m->attn_cnt);   // a zero word (the counters rest at zero between launches)
  // ---- persistent forward: sync words, op table, granule buffers (all zero: tag 0 is never a valid tag)
  {
    char* pp = p;
    const size_t pbytes = persist_workspace_bytes(c);
    SD_HIP_CHECK(hipMemset(pp, 0, pbytes));
    m->p_sync = reinterpret_cast<unsigned*>(pp);
    pp += 256;
    m->p_ops = reinterpret_cast<PersistOp*>(pp);
    pp += align_up(static_cast<size_t>(4 * c.n_layers + 1) * sizeof(PersistOp), 256);
    m->p_gran = reinterpret_cast<unsigned long long*>(pp);
    m->p_gran_parity = static_cast<unsigned>((pbytes - 256 - align_up(static_cast<size_t>(4 * c.n_layers + 1) * sizeof(PersistOp), 256)) / 16);
    m->persist_t = 0;
    m->persist_cap = 0;
    m->len_hint = m->Lmax;
    if (!m->host_status) SD_HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&m->host_status), 64, hipHostMallocDefault));
    *m->host_status = 0u;
    int dev = 0;
    hipDeviceProp_t prop{};
    SD_HIP_CHECK(hipGetDevice(&dev));
    SD_HIP_CHECK(hipGetDeviceProperties(&prop, dev));
    if (persist_model_ok(c, m->is_packed() != 0, m->w8() != 0, prop.multiProcessorCount)) {
      const int HqD = c.n_heads * c.head_dim;
      std::vector<PersistOp> ops;
      auto add = [&](int index, const void* norm_w, int n_pairs, int K, int kind, int layer) {
        const GemvGeom q = gemv_geometry(n_pairs, K);
        PersistOp o{};
        o.W = m->packed[index];
        o.norm_w = norm_w;
        o.pair_bytes = static_cast<unsigned>(4 * K);
        o.n_pairs = n_pairs;
        o.ppw = q.ppw;
        o.tile_pairs = q.tile_pairs;
        o.K = K;
        o.kind = kind;
        o.layer = layer;
        ops.push_back(o);
      };
      for (int l = 0; l < c.n_layers; ++l) {
        const sd_layer_weights& w = m->layers[l];
        add(4 * l + 0, w.attn_norm_w, (c.n_heads + 2 * c.n_kv_heads) * c.head_dim / 2, c.d_model, POP_QKV, l);
        add(4 * l + 1, nullptr, c.d_model / 2, HqD, POP_OUT, l);
        add(4 * l + 2, w.mlp_norm_w, c.d_ff, c.d_model, POP_GATEUP, l);
        add(4 * l + 3, nullptr, c.d_model / 2, c.d_ff, POP_DOWN, l);
      }
      add(4 * c.n_layers, c.final_norm_w, (c.vocab + 1) / 2, c.d_model, POP_HEAD, c.n_layers);
      SD_HIP_CHECK(hipMemcpy(m->p_ops, ops.data(), ops.size() * sizeof(PersistOp), hipMemcpyHostToDevice));
      // (a cache whose rows are not whole 8-key vectors cannot be walked by the launch's attention: launch path)
      m->persist_cap = persist_cache_ok(m->Lmax) ? persist_max_tokens(c) : 0;
      // tokens per pass the persistent launch takes (0 = off), by measurement (same box, 128 cached positions, us per forward,
      // persistent / launches; profiles/round3_persist_ab.md): Llama-3.2-1B dimensions 1 token 596 / 698, 2 tokens 677 / 718,
      // 3 tokens 838 / 741; Llama-3.2-3B dimensions 1 token 1506 / 1495, 2 tokens 1640 / 1540. So: 2 tokens up to
      // d_model = 2048 (the draft), off above it (the target's passes stay on launches).
      // And by context (kPersistMaxCtx): by the rows' CURRENT lengths as the caller bounds them (sd_model_set_length_hint; the
      // default bound is the cache size: sessions that size the cache to prompt + budget need not say anything).
      const char* env = getenv(debug_env::kPersistMaxT);
      const int want = env ? atoi(env) : (c.d_model <= 2048 ? 2 : 0);
      m->persist_t = m->persist_cap < want ? m->persist_cap : want;
      m->ctx_limit = env ? m->Lmax : kPersistMaxCtx;   // (the measurement override also lifts the context bound)
    }
  }
  return 0;
}

extern "C" int sd_model_set_persist_tokens(sd_model* m, int max_tokens) {
  clear_error();
  SD_REQUIRE(m && m->x, "set_persist_tokens: NULL argument / model not bound");
  SD_REQUIRE(max_tokens >= 0, "set_persist_tokens: max_tokens=%d", max_tokens);
  m->persist_t = max_tokens < m->persist_cap ? max_tokens : m->persist_cap;
  return 0;
}

extern "C" int sd_model_set_length_hint(sd_model* m, int max_len) {
  clear_error();
  SD_REQUIRE(m && m->x, "set_length_hint: NULL argument / model not bound");
  m->len_hint = (max_len <= 0 || max_len > m->Lmax) ? m->Lmax : max_len;
  return 0;
}

extern "C" int sd_model_persist_active(const sd_model* m, int T) {
  // 1 when a pass of T tokens of one row would run as the persistent launch right now (tokens, length hint, paging)
  return (m && T >= 1 && sd::persist_pass_ok(m, T, 1, T)) ? 1 : 0;
}

extern "C" const uint32_t* sd_model_status_word(const sd_model* m) { return m ? m->host_status : nullptr; }

extern "C" int sd_model_engine_status_clear(sd_model* m, void* stream) {
  clear_error();
  SD_REQUIRE(m, "engine_status_clear: NULL argument");
  if (!m->p_sync) return 0;
  hipStream_t st = static_cast<hipStream_t>(stream);
  SD_HIP_CHECK(hipStreamSynchronize(st));   // every workgroup of a failed launch has left
  unsigned w[2] = {0u, 0u};
  SD_HIP_CHECK(hipMemcpy(w, m->p_sync, 8, hipMemcpyDeviceToHost));
  // past the failed launch's tags (it did not advance the counter): granules it left behind can never match again
  w[0] = (w[0] + 2u) & 0x7fffffu;
  w[1] = 0u;
  SD_HIP_CHECK(hipMemcpy(m->p_sync, w, 8, hipMemcpyHostToDevice));
  if (m->host_status) *m->host_status = 0u;
  return 0;
}

extern "C" int sd_model_forward(sd_model* m, const int32_t* tokens, int tok_stride, const int32_t* pos_base,
                                int pos_off, int row0, int B, int M, int32_t* ids_out, int ids_stride, void* logits_out,
                                int logits_dtype, int skip_head, void* stream) {
  clear_error();
  return model_forward(m, tokens, tok_stride, pos_base, pos_off, row0, B, M, ids_out, ids_stride, logits_out,
                       logits_dtype, skip_head, static_cast<hipStream_t>(stream));
}

extern "C" int sd_model_hidden_rows(sd_model* m, int row0, int n, void* out, void* stream) {
  clear_error();
  SD_REQUIRE(m && m->x && out, "hidden_rows: NULL argument / model not bound");
  SD_REQUIRE(row0 >= 0 && n >= 1 && row0 + n <= kSkinnyMaxT, "hidden_rows: rows [%d,%d) outside the %d rows of a pass", row0, row0 + n, kSkinnyMaxT);
  const size_t d = static_cast<size_t>(m->cfg.d_model);
  SD_HIP_CHECK(hipMemcpyAsync(out, m->x + static_cast<size_t>(row0) * d, static_cast<size_t>(n) * d * 2, hipMemcpyDeviceToDevice,
                              static_cast<hipStream_t>(stream)));
  return 0;
}

// Measurement hook (bench.py "roofline" leg): one GEMV of the forward, launched `iters`
// times round-robin over the layers (so the weights come from HBM, not from the 256 MiB
// Infinity Cache) between two HIP events on `stream`.
extern "C" int sd_model_probe_gemv(sd_model* m, int which, int T, int iters, void* stream, float* avg_usec,
                                   double* bytes_per_launch) {
  clear_error();
  SD_REQUIRE(m && m->x && avg_usec && bytes_per_launch, "probe_gemv: NULL argument / model not bound");
  SD_REQUIRE(T >= 1 && T <= kSkinnyMaxT && iters >= 1, "probe_gemv: T=%d iters=%d", T, iters);
  const sd_model_config& c = m->cfg;
  const bool llama = (c.arch == SD_ARCH_LLAMA);
  const int d = c.d_model, ff = c.d_ff, Hq = c.n_heads, D = c.head_dim;
  hipStream_t st = static_cast<hipStream_t>(stream);
  hipEvent_t e0, e1;
  SD_HIP_CHECK(hipEventCreate(&e0));
  SD_HIP_CHECK(hipEventCreate(&e1));
  unsigned long long* dbg = nullptr;
  const bool timeline = getenv(debug_env::kGemvTimeline) != nullptr;
  if (timeline) {
    SD_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&dbg), 256 * 16 * sizeof(unsigned long long)));
    SD_HIP_CHECK(hipMemset(dbg, 0, 256 * 16 * sizeof(unsigned long long)));
  }
  // SPECDEC_PROBE_HOT=n: cycle over n layers only (n=1: the same matrix every launch, i.e. cache-resident weights)
  const int hot = getenv(debug_env::kProbeHot) ? atoi(getenv(debug_env::kProbeHot)) : 0;
  // > 9 tokens: as in the forward, the norm-fused launches read the row statistics the previous launch left, and the
  // residual launches leave them (SPECDEC_PROBE_NO_XSTAT=1: every launch computes its own, as for the first layer)
  const bool use_xstat = T > kGemvMaxT && T <= 64 && !getenv(debug_env::kProbeNoXstat);
  auto launch = [&](int l) -> int {
    if (hot > 0) l %= hot;
    const sd_layer_weights& w = m->layers[l % c.n_layers];
    GemvArgs g{};
    g.debug_ts = dbg;
    g.packed = m->is_packed();
    g.w8 = m->w8();
    g.w_scale = m->scale(which == 4 ? 4 * c.n_layers : 4 * (l % c.n_layers) + which);
    g.M = T;
    const int li = l % c.n_layers;
    g.T = T;
    g.M = T;
    g.out_dtype = SD_BF16;
    g.norm_eps = c.norm_eps;
    if (use_xstat && (which == 0 || which == 2 || which == 4)) { g.xstat_in = m->xstat; g.xstat_n = 256; }
    switch (which) {
      case 0: {  // norm + QKV projection + RoPE + in-place KV append (row 0 of the cache, positions 0..T-1)
        SD_REQUIRE(!m->block_table, "probe_gemv: the QKV probe writes dense cache rows (not available on a paged model)");
        const int Hkv = c.n_kv_heads;
        g.W = m->mat(4 * li + 0, w.wqkv); g.bias = w.bqkv; g.N = (Hq + 2 * Hkv) * D; g.K = d; g.n_pairs = g.N / 2;
        g.x = m->x; g.x_stride = d; g.prologue = llama ? PRO_RMSNORM : PRO_LAYERNORM; g.norm_w = w.attn_norm_w; g.norm_b = w.attn_norm_b;
        g.out = m->q; g.out_stride = Hq * D; g.head_dim = D; g.n_q_heads = Hq; g.n_kv_heads = Hkv; g.max_pos = c.max_pos; g.l_max = m->Lmax;
        g.rope_cos = llama ? c.rope_cos : nullptr; g.rope_sin = llama ? c.rope_sin : nullptr;
        g.pos_base = m->probe_pos; g.pos_off = 0; g.M = T;
        const size_t layer_kv = static_cast<size_t>(m->B) * Hkv * m->Lmax * D;
        g.k_cache = m->k_cache + li * layer_kv; g.v_cache = m->v_cache + li * layer_kv;
        return launch_gemv(g, EPI_QKV_ROPE, st);
      }
      case 1:  // attention output projection + residual
        g.W = m->mat(4 * li + 1, w.wo); g.bias = w.bo; g.N = d; g.K = Hq * D; g.n_pairs = d / 2;
        g.x = m->attn; g.x_stride = Hq * D; g.prologue = PRO_NONE; g.out = m->x; g.out_stride = d;
        if (use_xstat && gemm_resid_publishes_stats(g)) g.xstat_out = m->xstat;
        return launch_gemv(g, EPI_RESID, st);
      case 2:  // norm + gate/up + SwiGLU (GELU for GPT-2)
        g.W = m->mat(4 * li + 2, w.w_up); g.bias = w.b_up; g.K = d; g.x = m->x; g.x_stride = d;
        g.prologue = llama ? PRO_RMSNORM : PRO_LAYERNORM; g.norm_w = w.mlp_norm_w; g.norm_b = w.mlp_norm_b;
        g.out = m->act; g.out_stride = ff;
        if (llama) { g.N = 2 * ff; g.n_pairs = ff; return launch_gemv(g, EPI_SWIGLU, st); }
        g.N = ff; g.n_pairs = ff / 2;
        return launch_gemv(g, EPI_GELU, st);
      case 3:  // down projection + residual
        g.W = m->mat(4 * li + 3, w.w_down); g.bias = w.b_down; g.N = d; g.K = ff; g.n_pairs = d / 2;
        g.x = m->act; g.x_stride = ff; g.prologue = PRO_NONE; g.out = m->x; g.out_stride = d;
        if (use_xstat && gemm_resid_publishes_stats(g)) g.xstat_out = m->xstat;
        return launch_gemv(g, EPI_RESID, st);
      case 4:  // final norm + lm_head + fused argmax
        g.W = m->mat(4 * c.n_layers, c.lm_head); g.N = c.vocab; g.K = d; g.n_pairs = (c.vocab + 1) / 2; g.x = m->x; g.x_stride = d;
        g.prologue = llama ? PRO_RMSNORM : PRO_LAYERNORM; g.norm_w = c.final_norm_w; g.norm_b = c.final_norm_b;
        g.part_val = m->part_val; g.part_idx = m->part_idx;
        return launch_gemv(g, EPI_ARGMAX, st);
      default:
        set_error("probe_gemv: which=%d (0=qkv 1=o_proj 2=gate_up 3=down 4=lm_head)", which);
        return 1;
    }
  };
  double bytes = 0;
  switch (which) {
    case 0: bytes = 2.0 * (Hq + 2 * c.n_kv_heads) * D * d; break;
    case 1: bytes = 2.0 * d * Hq * D; break;
    case 2: bytes = 2.0 * (llama ? 2 : 1) * ff * d; break;
    case 3: bytes = 2.0 * d * ff; break;
    default: bytes = 2.0 * c.vocab * d; break;
  }
  if (m->w8()) bytes *= 0.5;  // one byte per weight
  for (int i = 0; i < 3; ++i)
    if (int rc = launch(i)) return rc;
  SD_HIP_CHECK(hipEventRecord(e0, st));
  for (int i = 0; i < iters; ++i)
    if (int rc = launch(i + 3)) return rc;
  SD_HIP_CHECK(hipEventRecord(e1, st));
  SD_HIP_CHECK(hipEventSynchronize(e1));
  float ms = 0.f;
  SD_HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  *avg_usec = ms * 1000.0f / iters;
  *bytes_per_launch = bytes;
  if (timeline) {  // stamps of the LAST launch: offsets from the earliest workgroup entry, in us
    std::vector<unsigned long long> h(256 * 8);
    SD_HIP_CHECK(hipMemcpy(h.data(), dbg, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    (void)hipFree(dbg);
    unsigned long long t0 = ~0ull;
    int n = 0;
    for (int b = 0; b < 256; ++b)
      if (h[b * 8]) { t0 = h[b * 8] < t0 ? h[b * 8] : t0; ++n; }
    static const char* names[8] = {"entry", "issued", "staged", "mfma_done", "reduced", "epilogue", "end", "(stats)"};  // gemm_pipe: issued = first loads issued, (stats) = statistics done, staged = chunk 0
    fprintf(stderr, "[timeline which=%d T=%d] %d workgroups, us from first entry (min / mean / max):\n", which, T, n);
    for (int s = 0; s < 8; ++s) {
      double mn = 1e30, mx = 0, sum = 0;
      for (int b = 0; b < 256; ++b) {
        if (!h[b * 8]) continue;
        const double v = (h[b * 8 + s] - t0) / 100.0;
        mn = v < mn ? v : mn; mx = v > mx ? v : mx; sum += v;
      }
      fprintf(stderr, "  %-10s %7.2f %7.2f %7.2f\n", names[s], mn, sum / (n ? n : 1), mx);
    }
    if (getenv(debug_env::kGemvTimeline)[0] == '2') {   // per-workgroup: when its K loop and the workgroup itself were done
      for (int s : {3, 6}) {
        fprintf(stderr, "[timeline-wg which=%d T=%d %s]", which, T, names[s]);
        for (int b = 0; b < 256; ++b) fprintf(stderr, " %.2f", h[b * 8] ? (h[b * 8 + s] - t0) / 100.0 : -1.0);
        fprintf(stderr, "\n");
      }
    }
  }
  return 0;
}

extern "C" int sd_model_persist_tokens(const sd_model* m) { return m ? m->persist_t : 0; }

extern "C" int sd_model_engine_status(sd_model* m, uint32_t* status_out, void* stream) {
  clear_error();
  SD_REQUIRE(m && status_out, "engine_status: NULL argument");
  *status_out = 0;
  if (!m->p_sync) return 0;
  SD_HIP_CHECK(hipMemcpyAsync(status_out, m->p_sync + 1, 4, hipMemcpyDeviceToHost, static_cast<hipStream_t>(stream)));
  SD_HIP_CHECK(hipStreamSynchronize(static_cast<hipStream_t>(stream)));
  return 0;
}

extern "C" int sd_model_debug_rows(sd_model* m, int which, int row0, int n, void* out, void* stream) {
  clear_error();
  SD_REQUIRE(m && m->x && out, "debug_rows: NULL argument / model not bound");
  SD_REQUIRE(row0 >= 0 && n >= 1 && row0 + n <= kSkinnyMaxT, "debug_rows: rows [%d,%d) outside the %d rows of a pass", row0, row0 + n, kSkinnyMaxT);
  const sd_model_config& c = m->cfg;
  const uint16_t* src = nullptr;
  size_t w = 0;
  switch (which) {
    case 0: src = m->x; w = c.d_model; break;
    case 1: src = m->q; w = static_cast<size_t>(c.n_heads) * c.head_dim; break;
    case 2: src = m->attn; w = static_cast<size_t>(c.n_heads) * c.head_dim; break;
    case 3: src = m->act; w = c.d_ff; break;
    default: SD_REQUIRE(false, "debug_rows: which=%d (0 x, 1 q, 2 attn, 3 act)", which);
  }
  SD_HIP_CHECK(hipMemcpyAsync(out, src + static_cast<size_t>(row0) * w, static_cast<size_t>(n) * w * 2, hipMemcpyDeviceToDevice,
                              static_cast<hipStream_t>(stream)));
  return 0;
}

extern "C" int sd_model_probe_forward(sd_model* m, int M, int pos0, int iters, int skip_head, void* stream, float* avg_usec,
                                      double* bytes_per_forward, unsigned long long* timeline, size_t timeline_cap) {
  clear_error();
  SD_REQUIRE(m && m->x && avg_usec && bytes_per_forward, "probe_forward: NULL argument / model not bound");
  SD_REQUIRE(M >= 1 && M <= kSkinnyMaxT && iters >= 1, "probe_forward: M=%d iters=%d", M, iters);
  SD_REQUIRE(pos0 >= 0 && pos0 + M <= m->Lmax, "probe_forward: positions [%d, %d) outside the cache rows (%d)", pos0, pos0 + M, m->Lmax);
  SD_REQUIRE(!m->block_table, "probe_forward: dense KV only");
  const sd_model_config& c = m->cfg;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int32_t* zeros = reinterpret_cast<const int32_t*>(m->attn_cnt);   // kAttnSplitSlots zero words: token ids and the position base
  auto fwd = [&]() { return model_forward(m, zeros, M, zeros, pos0, 0, 1, M, nullptr, M, nullptr, SD_BF16, skip_head, st); };
  hipEvent_t e0, e1;
  SD_HIP_CHECK(hipEventCreate(&e0));
  SD_HIP_CHECK(hipEventCreate(&e1));
  for (int i = 0; i < 2; ++i)
    if (int rc = fwd()) return rc;
  SD_HIP_CHECK(hipEventRecord(e0, st));
  for (int i = 0; i < iters; ++i)
    if (int rc = fwd()) return rc;
  SD_HIP_CHECK(hipEventRecord(e1, st));
  SD_HIP_CHECK(hipEventSynchronize(e1));
  float ms = 0.f;
  SD_HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  *avg_usec = ms * 1000.0f / iters;
  const bool llama = (c.arch == SD_ARCH_LLAMA);
  double per_layer = 2.0 * ((c.n_heads + 2.0 * c.n_kv_heads) * c.head_dim * c.d_model + static_cast<double>(c.d_model) * c.n_heads * c.head_dim +
                            (llama ? 3.0 : 2.0) * c.d_ff * c.d_model);
  double bytes = per_layer * c.n_layers + (skip_head ? 0.0 : 2.0 * c.vocab * c.d_model);
  if (m->w8()) bytes *= 0.5;
  *bytes_per_forward = bytes;
  if (timeline && m->persist_t >= M) {
    const size_t n_ops = static_cast<size_t>(4 * c.n_layers + (skip_head ? 0 : 1));
    const size_t words = static_cast<size_t>(kPersistCUs) * (12 * n_ops + 4);
    SD_REQUIRE(timeline_cap >= words, "probe_forward: timeline needs %zu words", words);
    unsigned long long* dbg = nullptr;
    SD_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&dbg), words * 8));
    SD_HIP_CHECK(hipMemsetAsync(dbg, 0, words * 8, st));
    m->p_debug = dbg;
    const int rc = fwd();
    m->p_debug = nullptr;
    if (rc) { (void)hipFree(dbg); return rc; }
    SD_HIP_CHECK(hipStreamSynchronize(st));
    SD_HIP_CHECK(hipMemcpy(timeline, dbg, words * 8, hipMemcpyDeviceToHost));
    (void)hipFree(dbg);
  }
  return 0;
}

// ---------------------------------------------------------------------------- step loop
static constexpr int kStageInts = 24;   // per row: 8 ints of set_row, then (accepted, proposed, hist_n, k) + 4 doubles

struct sd_specdec {
  sd_model* draft = nullptr;
  sd_model* target = nullptr;
  int B = 0, K = 0, mode = 0;
  SpecState st{};
  int32_t* dev_block = nullptr;   // all device state in one allocation
  int32_t* step_counter = nullptr; // device: steps executed (its parity selects the record slot)
  int32_t* host_record = nullptr;  // pinned, device-accessible: [2 slots][B][rec]
  hipEvent_t ev_done[2] = {nullptr, nullptr};   // recorded after launch i on the target stream (i & 1)
  long launches = 0;
  int32_t* host_stage = nullptr;  // pinned staging for set_row / set_adaptive_row: [B][kStageInts]
  int a_initial = 0;
  int rec = 0;
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  hipGraph_t graph = nullptr;
  hipGraphExec_t exec = nullptr;
  hipStream_t graph_st_t = nullptr, graph_st_d = nullptr;
  long steps = 0;
  // sampled bonus token (sd_specdec_set_sampling); all buffers are caller-owned
  int sample = 0;
  float temperature = 1.0f, top_p = 1.0f;
  int top_k = 0;
  uint64_t seed = 0;
  void* logits = nullptr;          // [B][K+1][V] bf16
  uint32_t* draw = nullptr;        // [B]
  const int32_t* stream_id = nullptr;
  // persistent Medusa heads (sd_specdec_set_medusa): K packed [V][d] matrices + their fp32 row scales (fp8)
  std::vector<const void*> heads;
  std::vector<const float*> head_scales;
  int32_t* head_rows = nullptr;    // [B] device: row of the target's residual stream each head reads
  size_t head_stride = 0;          // bytes between consecutive heads when they sit at a constant stride (one launch), else 0
  // EAGLE-lite (sd_specdec_set_eagle): extrapolated hidden rows instead of a draft model; caller-owned workspace
  bool draft_taps = false;         // the draft's persistent passes also store their stage rows (debug)
  bool want_fwd0_select = false;   // device-selected form of draft forward 0 (storage carved at create; eligibility per capture)
  int eagle = 0;
  float eagle_alpha = 0.7f;
  uint16_t* eagle_H = nullptr;     // [B*K][d]
  uint16_t* eagle_prev = nullptr;  // [B][d] last extrapolated row of the previous step
  int32_t* eagle_has = nullptr;    // [B]
};

namespace sd {

// draft tokens of the NEXT step from the heads: d_{i+1} = argmax head_i(final_norm(h)), h = the residual row of the
// position that produced the last emitted token. Heads the caller packed at a constant stride (sd_specdec_set_medusa)
// are evaluated by ONE lm_head-shaped launch (grid y = head: same x rows, same geometry, per-head argmax partials) and
// one finalize over K x B results; otherwise one launch + finalize per head.
static int enqueue_medusa_heads(sd_specdec* s, hipStream_t st) {
  sd_model* m = s->target;
  const sd_model_config& c = m->cfg;
  const int B = s->B, K = s->K;
  if (int rc = launch_medusa_rows(s->st, s->head_rows, st)) return rc;
  GemvArgs h{};
  h.packed = 1;
  h.w8 = s->head_scales.empty() ? 0 : 1;
  h.N = c.vocab;
  h.K = c.d_model;
  h.n_pairs = (c.vocab + 1) / 2;
  h.x = m->x;
  h.x_stride = c.d_model;
  h.x_row = s->head_rows;
  h.T = B;
  h.M = 1;
  h.prologue = (c.arch == SD_ARCH_LLAMA) ? PRO_RMSNORM : PRO_LAYERNORM;
  h.norm_w = c.final_norm_w;
  h.norm_b = c.final_norm_b;
  h.norm_eps = c.norm_eps;
  h.out = nullptr;
  h.out_dtype = SD_BF16;
  h.part_val = m->part_val;
  h.part_idx = m->part_idx;
  if (B > m->small_t) {   // more rows than a GEMV pass: gather them (the attention-output buffer is free after the verify forward)
    if (int rc = launch_medusa_gather(m->x, s->head_rows, m->attn, B, c.d_model, st)) return rc;
    h.x = m->attn;
    h.x_row = nullptr;
  }
  int ppw = 1;
  const int grid = gemv_grid(h, &ppw);
  if (B <= m->small_t && s->head_stride && static_cast<size_t>(K) * B * grid <= static_cast<size_t>(kSkinnyMaxT) * kMaxPartials) {
    h.W = s->heads[0];
    h.w_scale = s->head_scales.empty() ? nullptr : s->head_scales[0];
    h.batch_bytes = s->head_stride;
    h.n_batch = K;
    if (int rc = launch_gemv(h, EPI_ARGMAX, st)) return rc;
    // result (head j, row b) = "token" j * B + b of the partials -> verify_tok[b][j + 1]
    if (int rc = launch_argmax_finalize(m->part_val, m->part_idx, K * B, grid, -B, K + 1, s->st.verify_tok + 1, st)) return rc;
    return launch_medusa_commit(s->st, st);
  }
  for (int i = 0; i < K; ++i) {
    h.w_scale = s->head_scales.empty() ? nullptr : s->head_scales[i];
    h.W = s->heads[i];
    if (int rc = launch_gemv(h, EPI_ARGMAX, st)) return rc;
    // token of head i of row b -> verify_tok[b][i+1]
    if (int rc = launch_argmax_finalize(m->part_val, m->part_idx, B, grid, 1, K + 1, s->st.verify_tok + i + 1, st)) return rc;
  }
  return launch_medusa_commit(s->st, st);
}

static int enqueue_step(sd_specdec* s, hipStream_t st_t, hipStream_t st_d) {
  const int B = s->B, K = s->K;
  const bool two = (st_d != st_t) && s->draft;
  if (!s->draft && s->eagle) {
    // EAGLE-lite: residual row of `last` (1-token forward, head skipped) -> K extrapolated rows -> one lm_head launch
    sd_model* m = s->target;
    const sd_model_config& c = m->cfg;
    if (int rc = model_forward(m, s->st.verify_tok, K + 1, s->st.cur_len, 0, 0, B, 1, nullptr, 2, nullptr, SD_BF16, 1, st_t)) return rc;
    if (int rc = launch_eagle_extrapolate(m->x, s->eagle_H, s->eagle_prev, s->eagle_has, c.final_norm_w, c.final_norm_b, c.norm_eps,
                                          s->eagle_alpha, c.d_model, B, K, c.arch == SD_ARCH_LLAMA ? 1 : 0, st_t))
      return rc;
    GemvArgs h{};
    h.packed = m->is_packed();
    h.w8 = m->w8();
    h.w_scale = m->scale(4 * c.n_layers);
    h.W = m->mat(4 * c.n_layers, c.lm_head);
    h.N = c.vocab;
    h.K = c.d_model;
    h.n_pairs = (c.vocab + 1) / 2;
    h.x = s->eagle_H;
    h.x_stride = c.d_model;
    h.T = B * K;
    h.M = K;
    h.prologue = PRO_NONE;     // the rows are final-norm outputs already
    h.out = nullptr;
    h.out_dtype = SD_BF16;
    h.part_val = m->part_val;
    h.part_idx = m->part_idx;
    int ppw = 1;
    const int grid = gemv_grid(h, &ppw);
    if (int rc = launch_gemv(h, EPI_ARGMAX, st_t)) return rc;
    if (int rc = launch_argmax_finalize(m->part_val, m->part_idx, B * K, grid, K, K + 1, s->st.verify_tok + 1, st_t)) return rc;
    if (int rc = launch_medusa_commit(s->st, st_t)) return rc;
  } else if (!s->draft && s->heads.empty()) {
    // self-draft (Medusa-lite, tied heads): the target's own next token, K times
    if (int rc = model_forward(s->target, s->st.verify_tok, K + 1, s->st.cur_len, 0, 0, B, 1, s->st.draft_ids, 2, nullptr, SD_BF16, 0, st_t))
      return rc;
    if (int rc = launch_medusa_fill(s->st, st_t)) return rc;
  }
  if (two) {
    SD_HIP_CHECK(hipEventRecord(s->ev_fork, st_t));
    SD_HIP_CHECK(hipStreamWaitEvent(st_d, s->ev_fork, 0));
  }
  // draft: forward 0 over (prev, last) at positions cur_len-1, cur_len; then one token each
  struct TapsGuard {   // the draft's stage taps are this loop's choice, for the duration of its forwards only
    sd_model* m; bool saved;
    TapsGuard(sd_model* m_, bool v) : m(m_), saved(m_ ? m_->persist_taps : false) { if (m) m->persist_taps = v; }
    ~TapsGuard() { if (m) m->persist_taps = saved; }
  } taps_guard(s->draft, s->draft_taps);
  // both forms of forward 0 with the device picking one: only while the draft's 1- and 2-token passes are persistent launches
  // (a pass that is not needed then costs one launch that returns at entry, not 80) — decided per capture, the model may have
  // been re-bound or its persistent passes switched off since the loop was created
  const bool fwd0_select = s->draft && s->st.fwd0_w && B == 1 && persist_pass_ok(s->draft, 2, 1, 2);
  // a draft forward that is ONE pass leaves the lm_head partials of all its tokens in the model's workspace: ids and the hand-over
  // of d_{i+1} are then one launch (draft_finalize_kernel) instead of two
  const bool one_pass_d = s->draft && B * 2 <= s->draft->max_t;
  const auto finalize = [&](int M, int i, int32_t* ids, const int32_t* skip_k, int skip_i) -> int {
    if (one_pass_d)
      return launch_draft_finalize(s->draft->part_val, s->draft->part_idx, s->draft->head_grid, M, i, ids, s->st, skip_k, skip_i, st_d);
    return 0;
  };
  int32_t* const ids_d = one_pass_d ? nullptr : s->st.draft_ids;
  for (int i = 0; s->draft && i < K; ++i) {
    const int M = (i == 0) ? 2 : 1;
    const int32_t* toks = (i == 0) ? s->st.tok2 : s->st.next_tok;
    const int off = (i == 0) ? -1 : i;
    // per-row adaptive K: forward i >= 1 only matters while some row proposes more than i tokens
    s->draft->skip_k = (s->st.adaptive && i >= 1) ? s->st.k_active : nullptr;
    s->draft->skip_i = i;
    int rc_f;
    if (i == 0 && fwd0_select) {
      // both forms of forward 0, the device picks (SpecState::fwd0_w, written by accept_kernel): the 2-token pass, then the
      // 1-token pass over `last` alone, whose id lands where the 2-token pass leaves its second one
      s->draft->skip_k = s->st.fwd0_w;
      s->draft->skip_i = 1;
      rc_f = model_forward(s->draft, toks, 2, s->st.cur_len, -1, 0, B, 2, ids_d, 2, nullptr, SD_BF16, 0, st_d);
      if (!rc_f) rc_f = finalize(2, 0, s->st.draft_ids, s->st.fwd0_w, 1);
      if (!rc_f) {
        s->draft->skip_k = s->st.fwd0_w + 1;
        rc_f = model_forward(s->draft, toks + 1, 2, s->st.cur_len, 0, 0, B, 1, ids_d ? ids_d + 1 : nullptr, 2, nullptr, SD_BF16, 0, st_d);
      }
      if (!rc_f) rc_f = finalize(1, 0, s->st.draft_ids + 1, s->st.fwd0_w + 1, 1);
    } else {
      rc_f = model_forward(s->draft, toks, M, s->st.cur_len, off, 0, B, M, ids_d, 2, nullptr, SD_BF16, 0, st_d);
      if (!rc_f) rc_f = finalize(M, i, s->st.draft_ids, s->draft->skip_k, i);
    }
    s->draft->skip_k = nullptr;
    if (rc_f) return rc_f;
    if (!one_pass_d)
      if (int rc = launch_draft_next(M, i, s->st, st_d)) return rc;
  }
  if (two) {
    SD_HIP_CHECK(hipEventRecord(s->ev_join, st_d));
    SD_HIP_CHECK(hipStreamWaitEvent(st_t, s->ev_join, 0));
  }
  // verify: one forward over (last, d_1..d_K)
  // greedy steps whose verify forward is ONE pass: ids, accept scan, state advance and the step record are one launch over the
  // lm_head's partials (verify_tail_kernel) instead of three
  const unsigned* const st_word_d = (s->draft && s->draft->p_sync) ? s->draft->p_sync + 1 : nullptr;
  const unsigned* const st_word_t = s->target->p_sync ? s->target->p_sync + 1 : nullptr;
  const bool tail = !s->sample && verify_tail_fits(s->st) && B * (K + 1) <= s->target->max_t;
  if (int rc = model_forward(s->target, s->st.verify_tok, K + 1, s->st.cur_len, 0, 0, B, K + 1, tail ? nullptr : s->st.target_ids,
                             K + 1, s->sample ? s->logits : nullptr, SD_BF16, 0, st_t))
    return rc;
  if (tail) {
    if (int rc = launch_verify_tail(s->st, s->target->part_val, s->target->part_idx, s->target->head_grid, s->mode, s->host_record, s->rec,
                                    s->step_counter, st_word_d, st_word_t, st_t))
      return rc;
    if (!s->heads.empty())
      if (int rc = enqueue_medusa_heads(s, st_t)) return rc;
    return 0;
  }
  if (s->sample) {
    // accept length -> draw the token after the accepted prefix from the stored logits of that position
    if (int rc = launch_accept_len(s->st, st_t)) return rc;
    if (int rc = launch_sample_step(s->st, s->logits, s->target->cfg.vocab, s->temperature, s->top_k, s->top_p, s->seed,
                                    s->draw, s->stream_id, st_t))
      return rc;
  }
  if (int rc = launch_accept(s->st, s->mode, s->sample, st_t)) return rc;
  if (int rc = launch_pack_record(s->st, s->host_record, s->rec, s->step_counter, st_word_d, st_word_t, st_t)) return rc;
  // persistent Medusa heads: the proposals of the next step, after the record of this one has left
  if (!s->heads.empty())
    if (int rc = enqueue_medusa_heads(s, st_t)) return rc;
  return 0;
}

}  // namespace sd

extern "C" int sd_specdec_create(sd_model* draft, sd_model* target, int B, int K, int emit_mode, sd_specdec** out) {
  clear_error();
  SD_REQUIRE(target && out, "specdec_create: NULL argument");   // draft == NULL: self-draft (Medusa-lite, tied heads)
  SD_REQUIRE(B >= 1 && K >= 1 && K <= 8, "specdec_create: B=%d K=%d (K in 1..8)", B, K);
  SD_REQUIRE(B * (K + 1) <= 65535, "specdec_create: batch too large");
  SD_REQUIRE((!draft || draft->B >= B) && target->B >= B, "specdec_create: models must be bound with batch >= %d", B);
  SD_REQUIRE(!draft || draft->cfg.vocab == target->cfg.vocab, "specdec_create: draft/target vocabularies differ");
  SD_REQUIRE(emit_mode == SD_EMIT_BONUS || emit_mode == SD_EMIT_DRAFT, "specdec_create: emit_mode %d", emit_mode);
  // verify of B rows must fit the passes of the target forward: any B works (tiled)
  sd_specdec* s = new (std::nothrow) sd_specdec();
  SD_REQUIRE(s, "specdec_create: out of memory");
  s->draft = draft;
  s->target = target;
  // nobody reads the hidden rows of a loop's draft: its persistent passes skip the stage taps (csrc/persist.hip, TAPS). The
  // choice belongs to the LOOP (enqueue_step sets it around the draft's forwards): the same model object used elsewhere — as a
  // standalone model, or as another loop's target — keeps its taps.
  s->draft_taps = getenv(debug_env::kPersistTaps) != nullptr;
  s->B = B;
  s->K = K;
  s->mode = emit_mode;
  s->rec = 7 + 3 * K;
  const size_t n_state = static_cast<size_t>(B) * (1 + 1 + 2 + 1 + 2 + K + (K + 1) + (K + 1) + 1 + 1 + (K + 1) + 1);
  const size_t n_adapt = static_cast<size_t>(B) * (1 + 4 + 8) + 2 + 2 + 2;   // k_row, ctl, ctl_hist (doubles), k_active (+ alignment), fwd0_w
  const size_t n_total = n_state + 4 + n_adapt;
  hipError_t e = hipMalloc(&s->dev_block, n_total * sizeof(int32_t));
  if (e != hipSuccess) {
    delete s;
    SD_REQUIRE(false, "specdec_create: hipMalloc failed: %s", hipGetErrorString(e));
  }
  (void)hipMemset(s->dev_block, 0, n_total * sizeof(int32_t));
  int32_t* p = s->dev_block;
  SpecState& st = s->st;
  st.B = B;
  st.K = K;
  st.cur_len = p; p += B;
  st.active = p; p += B;
  st.tok2 = p; p += 2 * B;
  st.next_tok = p; p += B;
  st.draft_ids = p; p += 2 * B;
  st.draft_tok = p; p += static_cast<size_t>(B) * K;
  st.verify_tok = p; p += static_cast<size_t>(B) * (K + 1);
  st.target_ids = p; p += static_cast<size_t>(B) * (K + 1);
  st.accept_len = p; p += B;
  st.n_new = p; p += B;
  st.new_tok = p; p += static_cast<size_t>(B) * (K + 1);
  st.sampled = p; p += B;
  s->step_counter = p; p += 4;
  if ((p - s->dev_block) & 1) ++p;         // the doubles below: 8-byte aligned (hipMalloc is 256-byte aligned)
  st.ctl_hist = reinterpret_cast<double*>(p); p += static_cast<size_t>(B) * 8;
  st.ctl = p; p += static_cast<size_t>(B) * 4;
  st.k_row = p; p += B;
  st.k_active = p; p += 1;
  st.adaptive = 0;
  // device-selected form of draft forward 0: one sequence, a draft whose 1- and 2-token passes are persistent launches (a pass
  // that is not needed then costs one launch that returns at entry, not 80)
  st.fwd0_w = nullptr;
  if (draft && B == 1 && draft->persist_cap >= 2 && !getenv(debug_env::kNoFwd0Select)) {
    st.fwd0_w = p; p += 2;
    const int32_t init[2] = {2, 1};
    (void)hipMemcpy(st.fwd0_w, init, sizeof(init), hipMemcpyHostToDevice);
  }
  if (hipHostMalloc(reinterpret_cast<void**>(&s->host_record), sizeof(int32_t) * 2 * B * s->rec, hipHostMallocDefault) != hipSuccess ||
      hipEventCreateWithFlags(&s->ev_done[0], hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&s->ev_done[1], hipEventDisableTiming) != hipSuccess ||
      hipHostMalloc(reinterpret_cast<void**>(&s->host_stage), sizeof(int32_t) * (B * kStageInts + 4), hipHostMallocDefault) != hipSuccess ||
      hipEventCreateWithFlags(&s->ev_fork, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&s->ev_join, hipEventDisableTiming) != hipSuccess) {
    sd_specdec_destroy(s);
    SD_REQUIRE(false, "specdec_create: pinned memory / event creation failed");
  }
  *out = s;
  return 0;
}

extern "C" int sd_specdec_destroy(sd_specdec* s) {
  if (!s) return 0;
  if (s->exec) (void)hipGraphExecDestroy(s->exec);
  if (s->graph) (void)hipGraphDestroy(s->graph);
  if (s->ev_done[0]) (void)hipEventDestroy(s->ev_done[0]);
  if (s->ev_done[1]) (void)hipEventDestroy(s->ev_done[1]);
  if (s->ev_fork) (void)hipEventDestroy(s->ev_fork);
  if (s->ev_join) (void)hipEventDestroy(s->ev_join);
  if (s->host_record) (void)hipHostFree(s->host_record);
  if (s->host_stage) (void)hipHostFree(s->host_stage);
  if (s->dev_block) (void)hipFree(s->dev_block);
  if (s->head_rows) (void)hipFree(s->head_rows);
  delete s;
  return 0;
}

extern "C" int sd_specdec_set_row(sd_specdec* s, int b, int seq_len, int prev_tok, int last_tok, int active,
                                  void* stream) {
  clear_error();
  SD_REQUIRE(s, "specdec_set_row: NULL");
  SD_REQUIRE(b >= 0 && b < s->B, "specdec_set_row: row %d out of range", b);
  SD_REQUIRE(seq_len >= 1, "specdec_set_row: seq_len=%d (need at least one token)", seq_len);
  hipStream_t st = static_cast<hipStream_t>(stream);
  // the staging slot of row b must not be rewritten while a previous copy is in flight
  SD_HIP_CHECK(hipStreamSynchronize(st));
  int32_t* h = s->host_stage + b * kStageInts;
  h[0] = seq_len - 1;  // cur_len: position of `last`
  h[1] = active ? 1 : 0;
  h[2] = prev_tok;
  h[3] = last_tok;
  SD_HIP_CHECK(hipMemcpyAsync(s->st.cur_len + b, h + 0, 4, hipMemcpyHostToDevice, st));
  SD_HIP_CHECK(hipMemcpyAsync(s->st.active + b, h + 1, 4, hipMemcpyHostToDevice, st));
  SD_HIP_CHECK(hipMemcpyAsync(s->st.tok2 + 2 * b, h + 2, 8, hipMemcpyHostToDevice, st));
  SD_HIP_CHECK(hipMemcpyAsync(s->st.verify_tok + static_cast<size_t>(b) * (s->K + 1), h + 3, 4, hipMemcpyHostToDevice, st));
  if (s->st.fwd0_w) {   // whatever the caller did to the caches: the next step recomputes prev's K/V
    h[4] = 2;
    h[5] = 1;
    SD_HIP_CHECK(hipMemcpyAsync(s->st.fwd0_w, h + 4, 8, hipMemcpyHostToDevice, st));
  }
  return 0;
}

// Per-row adaptive K. The captured step keeps the shape K = max_k; row b's proposals past k_row[b]
// do not count (accept length clamped, bonus token = the target's token after the clamped prefix), and accept_kernel
// moves k_row[b] by the reference's rule. enable: all rows restart (k = initial_k, history = [0.0] — the reference's
// first get_k call reports acceptance_rate 0.0 before any step).
extern "C" int sd_specdec_set_adaptive(sd_specdec* s, int enable, int initial_k, int min_k, int max_k, int step_size,
                                       double target_rate, void* stream) {
  clear_error();
  SD_REQUIRE(s, "specdec_set_adaptive: NULL");
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (s->exec) {   // the captured kernels hold the loop state by value
    (void)hipGraphExecDestroy(s->exec);
    (void)hipGraphDestroy(s->graph);
    s->exec = nullptr;
    s->graph = nullptr;
  }
  if (!enable) {
    s->st.adaptive = 0;
    return 0;
  }
  SD_REQUIRE(min_k >= 1 && min_k <= max_k && max_k <= s->K, "specdec_set_adaptive: need 1 <= min_k (%d) <= max_k (%d) <= K of the loop (%d)", min_k, max_k, s->K);
  SD_REQUIRE(initial_k >= min_k && initial_k <= max_k, "specdec_set_adaptive: initial_k %d outside [%d, %d]", initial_k, min_k, max_k);
  SD_REQUIRE(step_size >= 1, "specdec_set_adaptive: step_size %d", step_size);
  SD_REQUIRE(target_rate == target_rate, "specdec_set_adaptive: target rate is NaN");
  SD_REQUIRE(s->heads.empty() && !s->eagle, "specdec_set_adaptive: stateful draft modes keep a fixed K");
  s->st.adaptive = 1;
  s->st.a_min = min_k;
  s->st.a_max = max_k;
  s->st.a_step = step_size;
  s->st.a_hi = target_rate + 0.1;
  s->st.a_lo = target_rate - 0.1;
  s->a_initial = initial_k;
  for (int b = 0; b < s->B; ++b)
    if (int rc = sd_specdec_set_adaptive_row(s, b, initial_k, 0, 0, 1, nullptr, st)) return rc;
  return 0;
}

// (Re)write one row's controller state — a new sequence in the row's slot, or the host's in-order view after the host
// rules overrode steps that were launched ahead. hist: hist_n <= 4 rates, oldest first (NULL: [0.0]).
extern "C" int sd_specdec_set_adaptive_row(sd_specdec* s, int b, int k, int accepted, int proposed, int hist_n, const double* hist,
                                           void* stream) {
  clear_error();
  SD_REQUIRE(s && s->st.adaptive, "specdec_set_adaptive_row: adaptive K is not enabled");
  SD_REQUIRE(b >= 0 && b < s->B, "specdec_set_adaptive_row: row %d out of range", b);
  SD_REQUIRE(k >= s->st.a_min && k <= s->st.a_max, "specdec_set_adaptive_row: k=%d outside [%d, %d]", k, s->st.a_min, s->st.a_max);
  SD_REQUIRE(hist_n >= 0 && hist_n <= 4 && accepted >= 0 && proposed >= 0, "specdec_set_adaptive_row: bad counters");
  hipStream_t st = static_cast<hipStream_t>(stream);
  SD_HIP_CHECK(hipStreamSynchronize(st));   // the staging slot must not be rewritten under a copy in flight
  int32_t* h = s->host_stage + b * kStageInts + 8;
  double* hd = reinterpret_cast<double*>(h + 4);
  h[0] = accepted;
  h[1] = proposed;
  h[2] = hist ? hist_n : 1;
  h[3] = k;
  for (int i = 0; i < 4; ++i) hd[i] = (hist && i < hist_n) ? hist[i] : 0.0;
  SD_HIP_CHECK(hipMemcpyAsync(s->st.ctl + 4 * b, h, 16, hipMemcpyHostToDevice, st));
  SD_HIP_CHECK(hipMemcpyAsync(s->st.ctl_hist + 4 * b, hd, 32, hipMemcpyHostToDevice, st));
  SD_HIP_CHECK(hipMemcpyAsync(s->st.k_row + b, h + 3, 4, hipMemcpyHostToDevice, st));
  // k_active (which draft forwards run) is recomputed at the end of every step; until then: all of them
  int32_t* ka = s->host_stage + s->B * kStageInts;
  *ka = s->st.a_max;
  SD_HIP_CHECK(hipMemcpyAsync(s->st.k_active, ka, 4, hipMemcpyHostToDevice, st));
  return 0;
}

extern "C" int sd_specdec_set_sampling(sd_specdec* s, int enable, float temperature, int top_k, float top_p,
                                       uint64_t seed, void* logits_buf, size_t logits_bytes, uint32_t* draw_counters,
                                       const int32_t* stream_ids) {
  clear_error();
  SD_REQUIRE(s, "specdec_set_sampling: NULL");
  if (s->exec) {  // the captured step holds the sampling launches and their parameters
    (void)hipGraphExecDestroy(s->exec);
    (void)hipGraphDestroy(s->graph);
    s->exec = nullptr;
    s->graph = nullptr;
  }
  if (!enable) {
    s->sample = 0;
    return 0;
  }
  SD_REQUIRE(s->mode == SD_EMIT_BONUS, "specdec_set_sampling: only the bonus-token emit mode (generate_batch) samples");
  SD_REQUIRE(logits_buf && draw_counters, "specdec_set_sampling: NULL logits buffer / draw counters");
  const size_t need = static_cast<size_t>(s->B) * (s->K + 1) * s->target->cfg.vocab * 2;
  SD_REQUIRE(logits_bytes >= need, "specdec_set_sampling: logits buffer %zu B < %zu B ([B][K+1][V] bf16)", logits_bytes, need);
  SD_REQUIRE(temperature == temperature && temperature >= 0.f, "specdec_set_sampling: temperature %g", temperature);
  SD_REQUIRE(top_k <= 0 || (top_k < s->target->cfg.vocab ? top_k : s->target->cfg.vocab) <= 1024, "specdec_set_sampling: top_k=%d > 1024", top_k);
  s->sample = 1;
  s->temperature = temperature;
  s->top_k = top_k;
  s->top_p = top_p;
  s->seed = seed;
  s->logits = logits_buf;
  s->draw = draw_counters;
  s->stream_id = stream_ids;
  return 0;
}

extern "C" size_t sd_specdec_eagle_bytes(int B, int K, int d_model) {
  if (B <= 0 || K <= 0 || d_model <= 0) return 0;
  return (static_cast<size_t>(B) * K * d_model + static_cast<size_t>(B) * d_model) * 2 + static_cast<size_t>(B) * 4 + 1024;
}

// workspace layout: [state rows B x d bf16][has_prev B x int32][extrapolated rows B*K x d bf16] — the state sits at
// offsets that do not depend on K, so loops of different K over the same workspace (adaptive K) share it
extern "C" int sd_specdec_set_eagle(sd_specdec* s, float alpha, void* workspace, size_t workspace_bytes_) {
  clear_error();
  SD_REQUIRE(s && workspace, "specdec_set_eagle: NULL argument");
  SD_REQUIRE(!s->draft, "specdec_set_eagle: the loop was created with a draft model (pass draft = NULL)");
  SD_REQUIRE(s->heads.empty(), "specdec_set_eagle: the loop already has Medusa heads");
  const int d = s->target->cfg.d_model;
  SD_REQUIRE(workspace_bytes_ >= sd_specdec_eagle_bytes(s->B, s->K, d), "specdec_set_eagle: workspace too small");
  SD_REQUIRE(s->B * s->K <= s->target->max_t, "specdec_set_eagle: B*K = %d rows exceed one lm_head pass (%d)", s->B * s->K, s->target->max_t);
  SD_REQUIRE(!s->exec, "specdec_set_eagle: call before the first captured step");
  char* p = reinterpret_cast<char*>((reinterpret_cast<uintptr_t>(workspace) + 255) & ~static_cast<uintptr_t>(255));
  s->eagle_prev = reinterpret_cast<uint16_t*>(p);
  p += (static_cast<size_t>(s->B) * d * 2 + 255) & ~static_cast<size_t>(255);
  s->eagle_has = reinterpret_cast<int32_t*>(p);
  p += (static_cast<size_t>(s->B) * 4 + 255) & ~static_cast<size_t>(255);
  s->eagle_H = reinterpret_cast<uint16_t*>(p);
  s->eagle_alpha = alpha;
  s->eagle = 1;
  return 0;
}

extern "C" int sd_specdec_reset_eagle(sd_specdec* s, void* stream) {
  clear_error();
  SD_REQUIRE(s && s->eagle, "specdec_reset_eagle: EAGLE mode is not set");
  SD_HIP_CHECK(hipMemsetAsync(s->eagle_has, 0, static_cast<size_t>(s->B) * 4, static_cast<hipStream_t>(stream)));
  return 0;
}

extern "C" int sd_specdec_set_medusa(sd_specdec* s, int n_heads, const void* const* packed_heads, int weight_dtype) {
  clear_error();
  SD_REQUIRE(s, "specdec_set_medusa: NULL");
  SD_REQUIRE(!s->draft, "specdec_set_medusa: the loop was created with a draft model (pass draft = NULL)");
  SD_REQUIRE(n_heads == s->K && packed_heads, "specdec_set_medusa: need K = %d heads, got %d", s->K, n_heads);
  SD_REQUIRE(weight_dtype == SD_BF16 || weight_dtype == SD_FP8_E4M3, "specdec_set_medusa: weight_dtype %d", weight_dtype);
  const sd_model_config& c = s->target->cfg;
  SD_REQUIRE(s->B <= s->target->small_t || (s->B <= s->target->max_t && c.n_heads * c.head_dim >= c.d_model),
             "specdec_set_medusa: batch %d exceeds one pass of the head kernels (%d rows)", s->B, s->target->max_t);
  SD_REQUIRE(s->B * (s->K + 1) <= s->target->max_t, "specdec_set_medusa: the verify pass must be a single pass (B*(K+1) = %d > %d)",
             s->B * (s->K + 1), s->target->max_t);
  if (s->exec) {
    (void)hipGraphExecDestroy(s->exec);
    (void)hipGraphDestroy(s->graph);
    s->exec = nullptr;
    s->graph = nullptr;
  }
  s->heads.clear();
  s->head_scales.clear();
  for (int i = 0; i < n_heads; ++i) {
    SD_REQUIRE(packed_heads[i], "specdec_set_medusa: head %d is NULL", i);
    s->heads.push_back(packed_heads[i]);
    if (weight_dtype == SD_FP8_E4M3) {
      const size_t off = packed_any_matrix_bytes((c.vocab + 1) / 2, c.d_model, SD_FP8_E4M3) -
                         ((static_cast<size_t>((c.vocab + 1) / 2) * 2 * 4 + 255) & ~static_cast<size_t>(255));
      s->head_scales.push_back(reinterpret_cast<const float*>(static_cast<const char*>(packed_heads[i]) + off));
    }
  }
  s->head_stride = 0;
  if (n_heads >= 2 && !getenv(debug_env::kMedusaPerHead)) {
    const char* h0 = static_cast<const char*>(packed_heads[0]);
    const char* h1 = static_cast<const char*>(packed_heads[1]);
    bool even = h1 > h0;
    for (int i = 2; even && i < n_heads; ++i)
      even = static_cast<const char*>(packed_heads[i]) - static_cast<const char*>(packed_heads[i - 1]) == h1 - h0;
    if (even) s->head_stride = static_cast<size_t>(h1 - h0);
  }
  if (!s->head_rows) SD_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&s->head_rows), sizeof(int32_t) * s->B));
  return 0;
}

extern "C" int sd_specdec_step(sd_specdec* s, void* stream_target, void* stream_draft, int use_graph) {
  clear_error();
  SD_REQUIRE(s, "specdec_step: NULL");
  hipStream_t st_t = static_cast<hipStream_t>(stream_target);
  hipStream_t st_d = stream_draft ? static_cast<hipStream_t>(stream_draft) : st_t;
  s->steps++;
  if (!use_graph || s->steps == 1) {
    // the first step always runs eagerly: it also performs the one-time kernel
    // attribute setup that must not happen inside a capture
    if (int rc = enqueue_step(s, st_t, st_d)) return rc;
    SD_HIP_CHECK(hipEventRecord(s->ev_done[s->launches & 1], st_t));
    s->launches++;
    return 0;
  }
  if (s->exec && (s->graph_st_t != st_t || s->graph_st_d != st_d)) {
    (void)hipGraphExecDestroy(s->exec);
    (void)hipGraphDestroy(s->graph);
    s->exec = nullptr;
    s->graph = nullptr;
  }
  if (!s->exec) {
    SD_HIP_CHECK(hipStreamBeginCapture(st_t, hipStreamCaptureModeRelaxed));
    const int rc = enqueue_step(s, st_t, st_d);
    hipGraph_t g = nullptr;
    const hipError_t ee = hipStreamEndCapture(st_t, &g);
    if (rc != 0) {
      if (g) (void)hipGraphDestroy(g);
      return rc;
    }
    SD_HIP_CHECK(ee);
    s->graph = g;
    SD_HIP_CHECK(hipGraphInstantiate(&s->exec, s->graph, nullptr, nullptr, 0));
    s->graph_st_t = st_t;
    s->graph_st_d = st_d;
  }
  SD_HIP_CHECK(hipGraphLaunch(s->exec, st_t));
  SD_HIP_CHECK(hipEventRecord(s->ev_done[s->launches & 1], st_t));   // completion of THIS step (sd_specdec_wait)
  s->launches++;
  return 0;
}

extern "C" int sd_specdec_invalidate(sd_specdec* s) {
  clear_error();
  SD_REQUIRE(s, "specdec_invalidate: NULL");
  if (s->exec) {
    (void)hipGraphExecDestroy(s->exec);
    (void)hipGraphDestroy(s->graph);
    s->exec = nullptr;
    s->graph = nullptr;
  }
  s->steps = 0;   // the next step runs eagerly (kernels of the other path may still need their one-time attribute setup)
  return 0;
}

extern "C" long sd_specdec_launches(const sd_specdec* s) { return s ? s->launches : 0; }

extern "C" int sd_specdec_wait(sd_specdec* s, long launch_index) {
  clear_error();
  SD_REQUIRE(s, "specdec_wait: NULL");
  SD_REQUIRE(launch_index >= 0 && launch_index < s->launches && launch_index + 2 >= s->launches,
             "specdec_wait: step %ld is not one of the last two launches (%ld launched)", launch_index, s->launches);
  SD_HIP_CHECK(hipEventSynchronize(s->ev_done[launch_index & 1]));
  return 0;
}

extern "C" int sd_specdec_sync(sd_specdec* s, void* stream) {
  clear_error();
  (void)s;
  SD_HIP_CHECK(hipStreamSynchronize(static_cast<hipStream_t>(stream)));
  return 0;
}

// two slots of [B][record_ints]: the record of launch i is in slot i & 1
extern "C" const int32_t* sd_specdec_record(const sd_specdec* s) { return s ? s->host_record : nullptr; }

extern "C" int sd_specdec_record_ints(const sd_specdec* s) { return s ? s->rec : 0; }
