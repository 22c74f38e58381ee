// Internal kernel-launch interfaces shared by the engine (not part of the C-ABI).
#pragma once

#include <vector>

#include "common.h"
#include "debug_env.h"

namespace sd {

// ---- skinny GEMM (gemv.hip) ---------------------------------------------------
enum GemvPrologue { PRO_NONE = 0, PRO_RMSNORM = 1, PRO_LAYERNORM = 2 };
enum GemvEpilogue { EPI_QKV_ROPE = 0, EPI_RESID = 1, EPI_SWIGLU = 2, EPI_GELU = 3, EPI_ARGMAX = 4 };

constexpr int kGemvMaxT = 9;     // tokens per launch of gemv.hip (K+1 for K = 8)
constexpr int kSkinnyMaxT = 128; // tokens per launch of gemm_skinny.hip (batched verify, chunked prefill)
static_assert(kSkinnyMaxT <= 255, "GemvArgs::T / M are 8-bit fields");

struct GemvArgs {
  // Field ORDER and WIDTHS are part of the design: these kernels are latency chains, every launch starts by pulling its
  // arguments through the scalar cache, and each further 64-byte line of the argument block that the first instructions
  // touch is a further miss (measured: two fields appended at byte 320 instead of byte 72 cost 0.8 % of the batch-1 step).
  // Lines 0 and 1 hold what EVERY variant needs before its first loads; the epilogue-specific fields follow.
  // ---- line 0: pointers
  const void* W;         // weights: bf16 [N][K] row-major, or the packed tile streams of csrc/pack.hip
  const void* x;         // activations in: bf16 [T][x_stride]
  void* out;             // q buffer / residual stream / activation / logits (may be null for ARGMAX)
  const void* norm_w;    // fused normalisation of x
  const void* norm_b;
  const void* bias;      // bf16 [N] or null
  const int32_t* x_row;  // nullable, device [T]: token t reads row x_row[t] of x instead of row t (gemv.hip only)
  // per-row adaptive K: a launch of draft forward skip_i leaves at once when *skip_k <= skip_i (null: never)
  const int32_t* skip_k;
  // ---- line 1: scalars
  unsigned long long* debug_ts;  // diagnostic timeline stamps [grid][8] or null
  int N, K;
  int n_pairs;       // row pairs processed (see pair_rows)
  int kw;            // K span per slice (set by launch_gemv)
  int x_stride;
  int out_stride;
  unsigned m_magic;  // ceil(65536 / M): t / M == (t * m_magic) >> 16 for t < 65536 / M... (exact for t < 512)
  float norm_eps;
  // small scalars as bitfields of 32-bit units (plain uint8_t / uint16_t kernel-argument fields made hipcc fetch some of
  // them with VECTOR byte loads; a bitfield is read as its dword through the scalar cache and extracted with s_bfe)
  unsigned T : 8;             // tokens in this pass (<= kSkinnyMaxT)
  unsigned M : 8;             // tokens per batch row (t = b*M + m)
  unsigned ppw : 16;          // pairs per workgroup (set by launch_gemv)
  unsigned n_tiles_full : 16; // tiles of a workgroup that owns a full share of ppw pairs
  unsigned tile_pairs : 4;    // pairs per tile, <= 8 (set by launch_gemv)
  unsigned ksplit : 5;        // K slices per tile, power of two <= 16 (set by launch_gemv)
  unsigned ks_shift : 3;      // log2(ksplit)   (derived by the launchers, gemv_derive: index arithmetic without runtime divisions)
  unsigned alias_part : 1;    // partial sums alias the staged x rows (set by launch_gemv)
  unsigned packed : 1;        // W is in the packed tile-stream order of csrc/pack.hip
  unsigned w8 : 1;            // W holds OCP fp8 e4m3 values (packed only); acc of row r is scaled by w_scale[r]
  unsigned : 1;
  unsigned prologue : 2;      // GemvPrologue
  unsigned out_dtype : 2;     // SD_BF16 | SD_F32 (logits only)
  unsigned skip_i : 4;
  int half_shift : 8;         // log2(head_dim / 2), or -1 when head_dim / 2 is not a power of two (division fallback)
  // ---- epilogue-specific
  const float* w_scale;          // fp32 [N] row scales (w8)
  // QKV epilogue: one 64-byte line (bytes 128..191)
  const int32_t* pos_base;  // device [B]
  const float* rope_cos;    // [max_pos][head_dim/2] or null (no RoPE)
  const float* rope_sin;
  void* k_cache;            // [B][n_kv_heads][l_max][head_dim] bf16
  void* v_cache;
  // paged KV (sd_model_bind_paged): k_cache / v_cache are page POOLS ([pages][Hkv][P][D] / [pages][Hkv][D][P]) and row b's
  // position pos lives in page block_table[b * (l_max >> page_shift) + (pos >> page_shift)] at offset pos & (P - 1).
  // null = dense rows of l_max positions.
  const int32_t* block_table;
  int pos_off;
  int max_pos;
  int l_max;
  unsigned head_dim : 8, n_q_heads : 8, n_kv_heads : 8, page_shift : 8;
  // ARGMAX epilogue: per-workgroup partials [T][grid]
  float* part_val;
  int* part_idx;
  // EPI_ARGMAX, gemv.hip only: n_batch equally shaped matrices at a constant byte stride (the K Medusa heads), one per
  // blockIdx.y, over the SAME x rows; partials of matrix j at part_val/part_idx + j * T * grid. 0 = a single matrix.
  size_t batch_bytes;
  int n_batch;
  // Row statistics handed from the kernel that WRITES the residual stream to the kernel that normalises it (> 9 tokens):
  // an EPI_RESID launch with xstat_out != null leaves, per token t and workgroup c, the sum of squares and the sum of the
  // new row values over the columns c owns: xstat_out[t * kStatStride + c] and xstat_out[kStatPlane + t * kStatStride + c].
  // A norm-fused launch with xstat_in != null adds the xstat_n partials of a token in a fixed order instead of reading
  // the whole row again (every one of its 256 workgroups would: 63 MB of L2 -> CU traffic at 40 tokens, 6-9 us).
  int xstat_n;
  float* xstat_out;
  const float* xstat_in;
};
constexpr int kStatStride = 256;                    // partials per token (>= workgroups of the producing launch)
constexpr int kStatPlane = kSkinnyMaxT * kStatStride;   // floats per plane (plane 0: sum of squares, plane 1: sum)
bool gemm_resid_publishes_stats(const GemvArgs& a);  // true when launch_gemv(a, EPI_RESID) takes a kernel that honours xstat_out


// fields of GemvArgs every launcher derives from the caller's (idempotent)
inline void gemv_derive(GemvArgs& a) {
  const int half = a.head_dim >> 1;
  a.half_shift = -1;
  if (half > 0 && (half & (half - 1)) == 0)
    for (int s = 0; s < 31; ++s)
      if ((1 << s) == half) a.half_shift = s;
  const int M = a.M > 0 ? a.M : 1;
  a.m_magic = static_cast<unsigned>((65536 + M - 1) / M);
  a.ks_shift = 0;
  while ((1 << a.ks_shift) < a.ksplit) ++a.ks_shift;
}

// work split of one matrix over the chip (shared by the launcher and the weight packer)
struct GemvGeom {
  int grid, ppw, n_tiles, tile_pairs, ksplit, kw;
};
GemvGeom gemv_geometry(int n_pairs, int K);
size_t packed_matrix_bytes(int n_pairs, int K);
size_t packed_any_matrix_bytes(int n_pairs, int K, int weight_dtype);   // bf16 or fp8 (+ scales)
int pack_one_matrix(const void* w_bf16, int N, int K, int n_pairs, int epi, int head_dim, int weight_dtype, void* dst, hipStream_t st);
size_t packed_scale_offset(const sd_model_config& c, int index);  // fp8: row scales follow the packed bytes of a matrix
size_t packed_offset(const sd_model_config& c, int index);  // index: 4*layer + {0 qkv,1 out,2 up,3 down}; 4*n_layers = lm_head
int gemv_grid(const GemvArgs& a, int* ppw_out);
int gemv_max_tokens(int K);                                          // tokens gemv.hip can stage for rows of K elements (<= 9)
int launch_gemv(const GemvArgs& a, int epi, hipStream_t st);          // T <= 9: gemv.hip, else gemm_skinny.hip
int launch_gemm_skinny(const GemvArgs& a, int epi, hipStream_t st);   // T <= 64
bool gemm_skinny_covers(int T, int n_pairs, int K, bool w8 = false);                  // shape handled by gemm_skinny.hip
// gemm_pipe.hip: the statically scheduled chunk pipeline for <= 64 tokens (tried first by launch_gemm_skinny, which has
// set the work split fields of `a`)
bool gemm_pipe_covers(int T, int n_pairs, int K, bool w8 = false);
int launch_gemm_pipe(const GemvArgs& a, const GemvGeom& q, int epi, hipStream_t st);

// ---- attention over the appended KV cache (attention.hip) -------------------------
struct AttnArgs {
  // line 0 (64 bytes): everything a short-context launch reads (see GemvArgs on why the order matters)
  const void* q;        // bf16 [T][n_q_heads*head_dim]
  const void* k_cache;  // bf16 [B][n_kv_heads][l_max][head_dim]
  const void* v_cache;
  void* out;            // bf16 [T][n_q_heads*head_dim]
  const int32_t* pos_base;
  const int32_t* skip_k;   // per-row adaptive K, as GemvArgs
  int pos_off;
  int l_max;
  float scale;
  unsigned M : 8, n_q_heads : 8, n_kv_heads : 8, n_split : 8;   // n_split: set by launch_attention
  // line 1
  unsigned head_dim : 8, page_shift : 8, skip_i : 4;
  int B;
  // split-KV over workgroups for long contexts (attention.hip): partial (max, sum, O) tiles and one
  // arrival counter per (row, kv head, query tile); null / 0 = every tile is one workgroup
  float* split_ws;
  unsigned* split_cnt;
  int split_slots;      // partial tiles the workspace holds
  const int32_t* block_table;   // paged KV, as GemvArgs (null = dense rows of l_max positions)
};
int launch_attention(const AttnArgs& a, hipStream_t st);
constexpr int kAttnSplitSlots = 1024;   // partial tiles of the split-KV workspace
size_t attention_split_ws_bytes(int head_dim);   // workspace for kAttnSplitSlots partial tiles (+ counters)

// ---- small kernels (misc.hip) -------------------------------------------------------
struct EmbedArgs {
  const void* tok_emb;   // bf16 [vocab][d]
  const void* pos_emb;   // bf16 [max_pos][d] or null
  const int32_t* tokens; // device, token (b,m) at tokens[b*tok_stride + m]
  int tok_stride;
  const int32_t* pos_base;
  int pos_off, M, T, d, vocab, max_pos;
  void* x;               // bf16 [T][d]
  const int32_t* skip_k;   // as GemvArgs
  int skip_i;
};
int launch_embed(const EmbedArgs& a, hipStream_t st);

// reduce the per-workgroup argmax partials: ids[b*ids_stride + m] = argmax over grid
int launch_argmax_finalize(const float* part_val, const int* part_idx, int T, int grid, int M,
                           int ids_stride, int32_t* ids_out, hipStream_t st, const int32_t* skip_k = nullptr, int skip_i = 0);

}  // namespace sd
