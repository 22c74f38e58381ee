// Prompt prefill as GEMMs: passes of more than 128 tokens of ONE row (a prompt being absorbed into the KV cache).
//
// What it replaces: the first full-prefix forward of the reference (/root/reference/src/specdec/models/hf_wrappers.py:417: one HF
// forward over the whole prompt). Rounds 1-3 fed a prompt through the decode-shaped kernels in passes of <= 128 tokens: every pass
// streams every weight once and the multi-token body at 128 tokens retires them at ~110 TFLOP/s, so a 512-token prompt cost 4
// passes = 41 ms for the 3B + 1B pair (profiles/round3_context_scaling.md) — ~10 % of what the bytes alone would take. A prompt is
// the one place in this path where the work IS a plain GEMM (hundreds of rows against every weight matrix), and a plain GEMM is what
// the design rules hand to the library: each matrix product of a <= 512-token chunk is one rocBLAS GEMM (bf16 operands, fp32
// accumulate, fp32 result), and everything that gives this engine its numerics stays in this repo's kernels:
//   * rms_rows_kernel           — the RMSNorm of the GEMV prologue (HF LlamaRMSNorm rounding points: rmsnorm_pair, common.h)
//   * epilogue_rows_kernel<EPI> — the SAME fused epilogues as the GEMV / multi-token kernels (gemv_device.h: epilogue<EPI>), applied
//                                 to the fp32 products: RoPE + q store + in-place K / V^T append, residual add, SwiGLU — so the
//                                 bf16 rounding points are those of every other path (RoPE on the fp32 sum, rounded once)
//   * attention                 — the existing MFMA attention kernel, in sub-passes of 128 query positions over the cache the QKV
//                                 epilogue has just appended to (causal by position)
// Sums differ from the GEMV path in fp32 summation order only; tests/test_hip_prefill_gemm_gpu.py holds the chunked path against the
// 128-token passes (next tokens, logits, K / V rows).
//
// rocBLAS is opened at first use (dlopen: the C-ABI library itself does not link it); when it is missing, or for GPT-2 models, fp8
// storage or paged KV, prefill falls back to the 128-token passes. Llama, bf16 row-major weights (the caller's HF-layout tensors:
// sd_layer_weights), dense KV.

#include <dlfcn.h>

#include "gemv_device.h"
#include "prefill_gemm.h"

namespace sd {

namespace {

// ---- rocBLAS through dlopen (the five entry points used; types reduced to what crosses the boundary) -----------------------
typedef void* rb_handle;
typedef int (*rb_create_t)(rb_handle*);
typedef int (*rb_destroy_t)(rb_handle);
typedef int (*rb_set_stream_t)(rb_handle, hipStream_t);
typedef int (*rb_gemm_ex_t)(rb_handle, int, int, int, int, int, const void*, const void*, int, int, const void*, int, int, const void*,
                            const void*, int, int, void*, int, int, int, int, int32_t, uint32_t);
constexpr int kOpN = 111, kOpT = 112;                 // rocblas_operation_none / _transpose
constexpr int kF32 = 151, kBf16 = 168;                // rocblas_datatype_f32_r / _bf16_r
constexpr int kAlgoStandard = 0;

struct Blas {
  void* lib = nullptr;
  rb_handle h = nullptr;
  rb_create_t create = nullptr;
  rb_destroy_t destroy = nullptr;
  rb_set_stream_t set_stream = nullptr;
  rb_gemm_ex_t gemm_ex = nullptr;
  bool tried = false, ok = false;
};
Blas g_blas;

bool blas_ready() {
  Blas& b = g_blas;
  if (b.tried) return b.ok;
  b.tried = true;
  for (const char* name : {"librocblas.so.5", "librocblas.so", "/opt/rocm/lib/librocblas.so"}) {
    b.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
    if (b.lib) break;
  }
  if (!b.lib) return false;
  b.create = reinterpret_cast<rb_create_t>(dlsym(b.lib, "rocblas_create_handle"));
  b.destroy = reinterpret_cast<rb_destroy_t>(dlsym(b.lib, "rocblas_destroy_handle"));
  b.set_stream = reinterpret_cast<rb_set_stream_t>(dlsym(b.lib, "rocblas_set_stream"));
  b.gemm_ex = reinterpret_cast<rb_gemm_ex_t>(dlsym(b.lib, "rocblas_gemm_ex"));
  if (!b.create || !b.destroy || !b.set_stream || !b.gemm_ex) return false;
  if (b.create(&b.h) != 0 || !b.h) return false;
  b.ok = true;
  return true;
}

// Y[T][N] (fp32, row-major) = X[T][K] (bf16, rows ldx apart) x W[N][K]^T (bf16 row-major: the HF Linear layout).
// Column-major view: Y^T (N x T, ld N) = op_T(W as K x N, ld K) x (X^T as K x T, ld ldx).
int gemm_rows(const void* W, const void* X, float* Y, int T, int N, int K, int ldx, hipStream_t st) {
  Blas& b = g_blas;
  SD_REQUIRE(b.ok, "prefill: rocBLAS is not available");
  SD_REQUIRE(b.set_stream(b.h, st) == 0, "prefill: rocblas_set_stream failed");
  const float alpha = 1.0f, beta = 0.0f;
  const int rc = b.gemm_ex(b.h, kOpT, kOpN, N, T, K, &alpha, W, kBf16, K, X, kBf16, ldx, &beta, Y, kF32, N, Y, kF32, N, kF32, kAlgoStandard, 0, 0);
  SD_REQUIRE(rc == 0, "prefill: rocblas_gemm_ex(%d x %d x %d) failed with status %d", N, T, K, rc);
  return 0;
}

// ---- RMSNorm of T rows: out[t] = weight * (x[t] * rsqrt(mean(x^2) + eps)).to(bf16), rounded again (rmsnorm_pair) -----------------
__global__ __launch_bounds__(256) void rms_rows_kernel(const uint16_t* x, int x_stride, const uint16_t* w, float eps, int d, uint16_t* out) {
  __shared__ float red[4];
  const int t = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const uint32_t* row = reinterpret_cast<const uint32_t*>(x + static_cast<size_t>(t) * x_stride);
  const uint32_t* w2 = reinterpret_cast<const uint32_t*>(w);
  const int np = d >> 1;
  f32x2_t s2 = {0.f, 0.f};
  for (int i = tid; i < np; i += 256) {
    const f32x2_t f = bf16x2_unpack(row[i]);
    s2 += f * f;
  }
  const float part = wave_reduce_sum(s2.x + s2.y);
  if (lane == 0) red[wave] = part;
  __syncthreads();
  const float sq = (red[0] + red[1]) + (red[2] + red[3]);
  const float rs = rsqrtf(sq / static_cast<float>(d) + eps);
  uint32_t* o = reinterpret_cast<uint32_t*>(out + static_cast<size_t>(t) * d);
  for (int i = tid; i < np; i += 256) o[i] = rmsnorm_pair(row[i], rs, w2[i]);
}

// ---- the fused epilogues of the weight-streaming kernels, applied to fp32 products ---------------------------------------------------
// Y: [T][N] fp32. GemvArgs carries token counts in 8-bit fields, so every group of 128 tokens gets its own view of the arguments
// (pos_off and the token-indexed output moved on by 128 tokens) and the epilogue sees token indices below 128 — exactly what it
// sees in a 128-token pass.
template <int EPI>
__global__ __launch_bounds__(256) void epilogue_rows_kernel(GemvArgs a, const float* Y, int T, int out_elem_stride) {
  // one token per block row, consecutive threads = consecutive pairs: the fp32 products of a token are read along the row
  // (pairs of the residual epilogues are adjacent columns, SwiGLU's are (p, p + ff), RoPE's (i, i + D/2) of a head)
  const int t = blockIdx.y;
  const int sub = t >> 7, t_local = t & 127;
  const int p = static_cast<int>(blockIdx.x) * 256 + static_cast<int>(threadIdx.x);
  if (p >= a.n_pairs) return;
  a.pos_off += sub * 128;
  a.out = static_cast<char*>(a.out) + static_cast<size_t>(sub) * 128 * out_elem_stride;
  int r0, r1;
  pair_rows<EPI>(a, p, r0, r1);
  const float* yr = Y + static_cast<size_t>(t) * a.N;
  float best_v = 0.f;
  int best_i = 0;
  epilogue<EPI>(a, p, r0, r1, t_local, yr[r0], (r1 < a.N) ? yr[r1] : 0.f, best_v, best_i);
}

template <int EPI>
int launch_epilogue_rows(GemvArgs a, const float* Y, int T, int out_elem_stride, hipStream_t st) {
  gemv_derive(a);
  hipLaunchKernelGGL((epilogue_rows_kernel<EPI>), dim3((a.n_pairs + 255) / 256, T), dim3(256), 0, st, a, Y, T, out_elem_stride);
  SD_LAUNCH_CHECK();
  return 0;
}

}  // namespace

bool prefill_gemm_available() { return blas_ready(); }

size_t prefill_gemm_workspace_bytes(const sd_model_config& c) {
  const size_t T = kPrefillChunk;
  const size_t HqD = static_cast<size_t>(c.n_heads) * c.head_dim;
  size_t nmax = static_cast<size_t>(2) * c.d_ff;
  if ((c.n_heads + 2 * static_cast<size_t>(c.n_kv_heads)) * c.head_dim > nmax) nmax = (c.n_heads + 2 * static_cast<size_t>(c.n_kv_heads)) * c.head_dim;
  auto up = [](size_t v) { return (v + 255) & ~static_cast<size_t>(255); };
  return up(T * c.d_model * 2) * 2 + up(T * HqD * 2) * 2 + up(T * c.d_ff * 2) + up(T * nmax * 4) + 256;
}

// One chunk of Mc <= kPrefillChunk positions of row `row` (absolute cache row), positions pos_base[row] + pos_off + [0, Mc).
// ws: prefill_gemm_workspace_bytes(c) bytes. Leaves the residual rows in ws (x) — the caller takes the hidden rows / runs the head.
int prefill_gemm_chunk(const PrefillModel& m, const int32_t* tokens, const int32_t* pos_base_row, int pos_off, int cache_row, int Mc, void* ws,
                       uint16_t** x_out, hipStream_t st) {
  const sd_model_config& c = *m.cfg;
  SD_REQUIRE(Mc >= 1 && Mc <= kPrefillChunk, "prefill: chunk of %d positions", Mc);
  const int d = c.d_model, Hq = c.n_heads, Hkv = c.n_kv_heads, D = c.head_dim, ff = c.d_ff, HqD = Hq * D;
  auto up = [](size_t v) { return (v + 255) & ~static_cast<size_t>(255); };
  char* p = reinterpret_cast<char*>((reinterpret_cast<uintptr_t>(ws) + 255) & ~static_cast<uintptr_t>(255));
  const size_t T = kPrefillChunk;
  uint16_t* x = reinterpret_cast<uint16_t*>(p); p += up(T * d * 2);
  uint16_t* xn = reinterpret_cast<uint16_t*>(p); p += up(T * d * 2);
  uint16_t* q = reinterpret_cast<uint16_t*>(p); p += up(T * HqD * 2);
  uint16_t* attn = reinterpret_cast<uint16_t*>(p); p += up(T * HqD * 2);
  uint16_t* act = reinterpret_cast<uint16_t*>(p); p += up(T * ff * 2);
  float* Y = reinterpret_cast<float*>(p);

  // embedding rows (Llama: no position table), in groups of <= 128 tokens (EmbedArgs / the kernel index tokens as t = b * M + m)
  for (int s0 = 0; s0 < Mc; s0 += 128) {
    EmbedArgs e{};
    e.tok_emb = c.tok_emb;
    e.pos_emb = nullptr;
    e.tokens = tokens + s0;
    e.tok_stride = Mc;
    e.pos_base = pos_base_row;
    e.pos_off = pos_off + s0;
    e.M = (Mc - s0 < 128) ? Mc - s0 : 128;
    e.T = e.M;
    e.d = d;
    e.vocab = c.vocab;
    e.max_pos = c.max_pos;
    e.x = x + static_cast<size_t>(s0) * d;
    if (int rc = launch_embed(e, st)) return rc;
  }
  const size_t layer_kv = static_cast<size_t>(m.B) * Hkv * m.Lmax * D;
  const size_t row_kv = static_cast<size_t>(cache_row) * Hkv * m.Lmax * D;
  for (int l = 0; l < c.n_layers; ++l) {
    const sd_layer_weights& w = c.layers[l];
    uint16_t* kc = m.k_cache + l * layer_kv + row_kv;
    uint16_t* vc = m.v_cache + l * layer_kv + row_kv;
    GemvArgs g{};
    g.T = 128;                 // (per 128-token group: see epilogue_rows_kernel)
    g.M = 128;
    g.pos_base = pos_base_row;
    g.pos_off = pos_off;
    g.head_dim = D;
    g.n_q_heads = Hq;
    g.n_kv_heads = Hkv;
    g.max_pos = c.max_pos;
    g.l_max = m.Lmax;
    g.out_dtype = SD_BF16;

    // 1. norm, QKV product, RoPE + q store + in-place K / V^T append
    hipLaunchKernelGGL(rms_rows_kernel, dim3(Mc), dim3(256), 0, st, x, d, static_cast<const uint16_t*>(w.attn_norm_w), c.norm_eps, d, xn);
    SD_LAUNCH_CHECK();
    const int Nqkv = (Hq + 2 * Hkv) * D;
    if (int rc = gemm_rows(w.wqkv, xn, Y, Mc, Nqkv, d, d, st)) return rc;
    GemvArgs a1 = g;
    a1.N = Nqkv;
    a1.K = d;
    a1.n_pairs = Nqkv / 2;
    a1.bias = w.bqkv;
    a1.out = q;
    a1.out_stride = HqD;
    a1.rope_cos = c.rope_cos;
    a1.rope_sin = c.rope_sin;
    a1.k_cache = kc;
    a1.v_cache = vc;
    if (int rc = launch_epilogue_rows<EPI_QKV_ROPE>(a1, Y, Mc, HqD * 2, st)) return rc;

    // 2. attention of the chunk's positions over the cache they have just been appended to, 128 query positions per launch
    for (int s0 = 0; s0 < Mc; s0 += 128) {
      AttnArgs at{};
      at.q = q + static_cast<size_t>(s0) * HqD;
      at.k_cache = kc;
      at.v_cache = vc;
      at.out = attn + static_cast<size_t>(s0) * HqD;
      at.pos_base = pos_base_row;
      at.pos_off = pos_off + s0;
      at.B = 1;
      at.M = (Mc - s0 < 128) ? Mc - s0 : 128;
      at.n_q_heads = Hq;
      at.n_kv_heads = Hkv;
      at.head_dim = D;
      at.l_max = m.Lmax;
      at.scale = 1.0f / sqrtf(static_cast<float>(D));
      at.split_ws = m.attn_ws;
      at.split_cnt = m.attn_cnt;
      at.split_slots = kAttnSplitSlots;
      if (int rc = launch_attention(at, st)) return rc;
    }

    // 3. output projection + residual
    if (int rc = gemm_rows(w.wo, attn, Y, Mc, d, HqD, HqD, st)) return rc;
    GemvArgs a3 = g;
    a3.N = d;
    a3.K = HqD;
    a3.n_pairs = d / 2;
    a3.bias = w.bo;
    a3.out = x;
    a3.out_stride = d;
    if (int rc = launch_epilogue_rows<EPI_RESID>(a3, Y, Mc, d * 2, st)) return rc;

    // 4. norm, gate / up product, SwiGLU
    hipLaunchKernelGGL(rms_rows_kernel, dim3(Mc), dim3(256), 0, st, x, d, static_cast<const uint16_t*>(w.mlp_norm_w), c.norm_eps, d, xn);
    SD_LAUNCH_CHECK();
    if (int rc = gemm_rows(w.w_up, xn, Y, Mc, 2 * ff, d, d, st)) return rc;
    GemvArgs a4 = g;
    a4.N = 2 * ff;
    a4.K = d;
    a4.n_pairs = ff;
    a4.bias = w.b_up;
    a4.out = act;
    a4.out_stride = ff;
    if (int rc = launch_epilogue_rows<EPI_SWIGLU>(a4, Y, Mc, ff * 2, st)) return rc;

    // 5. down projection + residual
    if (int rc = gemm_rows(w.w_down, act, Y, Mc, d, ff, ff, st)) return rc;
    GemvArgs a5 = g;
    a5.N = d;
    a5.K = ff;
    a5.n_pairs = d / 2;
    a5.bias = w.b_down;
    a5.out = x;
    a5.out_stride = d;
    if (int rc = launch_epilogue_rows<EPI_RESID>(a5, Y, Mc, d * 2, st)) return rc;
  }
  *x_out = x;
  return 0;
}

}  // namespace sd
