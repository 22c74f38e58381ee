// Persistent decode forward (csrc/persist.hip): ONE launch per forward for passes of <= kPersistMaxT tokens.
// Internal interface between the engine (engine.hip) and the kernel; not part of the C-ABI.
#pragma once

#include "kernels.h"

namespace sd {

enum PersistKind { POP_QKV = 0, POP_OUT = 1, POP_GATEUP = 2, POP_DOWN = 3, POP_HEAD = 4 };

// One matrix of the forward, in the order the weights are streamed (device array, read through the scalar cache).
struct PersistOp {
  const void* W;        // packed tile streams of the matrix (csrc/pack.hip): workgroup c owns pairs [c*ppw, (c+1)*ppw)
  const void* norm_w;   // RMSNorm weight applied to the op's input rows (QKV, GATEUP, HEAD), else null
  unsigned pair_bytes;  // bytes of one row pair in the packed stream (4 * K)
  int n_pairs, ppw, tile_pairs;
  int K;                // input length (elements, multiple of 128)
  int kind, layer;
  int pad_[5];
};
static_assert(sizeof(PersistOp) == 64, "PersistOp is read as whole 64-byte records");

constexpr int kPersistMaxT = 8;      // tokens per pass (epilogue items of a tile: 8 pairs x 8 tokens = one wave)
constexpr int kPersistCUs = 256;     // one workgroup per CU; the packed layout is cut for 256 workgroups
constexpr int kPersistMaxQkvTiles = 4;

// hand-off edges of a layer (granule buffers); tag = (launch << 9) | (layer * 8 + edge + 1)
enum PersistEdge { PE_X = 0, PE_QKV = 1, PE_ATTN = 2, PE_X2 = 3, PE_ACT = 4 };

struct PersistArgs {
  const PersistOp* ops;
  int n_ops;
  int d_model, n_q_heads, n_kv_heads, head_dim, d_ff, vocab, n_layers, max_pos;
  float norm_eps, attn_scale;
  const void* tok_emb;
  const float* rope_cos;
  const float* rope_sin;
  const int32_t* tokens;   // token (b, m) at tokens[b * tok_stride + m]
  int tok_stride;
  const int32_t* pos_base; // [B]
  int pos_off;
  int B, M;                // T = B * M tokens, t = b * M + m
  void* k_cache;           // layer 0, first row of this pass: [B][Hkv][l_max][D]
  void* v_cache;           //                                  [B][Hkv][D][l_max]
  size_t layer_kv;         // elements between the caches of consecutive layers
  int l_max;
  void* logits;            // optional [T][logits_stride]
  int logits_dtype, logits_stride;
  float* part_val;         // [T][256] per-workgroup argmax partials of the lm_head
  int* part_idx;
  void* x;                 // workspace taps, same meaning as in the launch-per-operator forward (last layer's values)
  void* q;
  void* attn;
  void* act;
  unsigned long long* gran;   // [2 parities][gran_parity] 8-byte {tag, value} granules
  unsigned gran_parity;
  unsigned off_edge[5];       // granule offset of each edge inside a parity (in storage granules)
  // Granule g of an edge is stored at (g >> 4) * gran_unit + (g & 15): with gran_unit = 16 the vector is dense; with
  // gran_unit = 528 (4 KiB + 128 B) every 128-byte unit sits in its own page, so the 256 CUs that all sweep the same vector
  // spread over the memory-side cache slices instead of queueing on the few a dense 8-32 KiB buffer maps to
  unsigned gran_unit;
  unsigned* sync;             // [0] launch counter, [1] status (0 = ok), [2..] reserved
  unsigned* host_status;      // optional, pinned host memory: a workgroup that gives up also stores its reason here
  const int32_t* skip_k;      // per-row adaptive K, as GemvArgs
  int skip_i;
  // LDS carve (bytes), set by launch_persist_forward
  unsigned lds_rope, lds_resid, lds_part, lds_attn, lds_u, lds_ring;
  unsigned ring_bytes;        // multiple of 1 KiB
  unsigned u_stride;          // bytes between token rows of the staged input
  unsigned resid_ppw;         // pairs per workgroup of the d_model-wide matrices
  int taps;                       // 1: also write the stage rows x / q / attn / act (tests, callers that read hidden rows); 0: the draft of a loop
  unsigned long long* debug_ts;   // optional [256][4 * n_ops + n_ops] 100 MHz stamps
};

// bytes of workspace the persistent path needs for a model of these dimensions (granules + op table + sync words)
size_t persist_workspace_bytes(const sd_model_config& c);
constexpr unsigned kGranUnitMax = 528;   // storage granules per 16-granule unit the buffers are sized for
// static eligibility of a model (architecture, dtype, shapes); T-dependent limits are checked per pass
bool persist_model_ok(const sd_model_config& c, bool packed, bool w8, int n_cus);
// cache rows the attention of the persistent launch can walk (16-byte V^T vectors of 8 keys)
inline bool persist_cache_ok(int l_max) { return l_max >= 8 && l_max % 8 == 0; }
// tokens one pass can hold for this model (LDS: staged rows + ring), 0 = none
int persist_max_tokens(const sd_model_config& c);
int launch_persist_forward(PersistArgs a, hipStream_t st);

}  // namespace sd
