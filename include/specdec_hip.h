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
 *         n_b = accept_len[b]==0 ? 0 : min(popcount(mask[b,:]), accept_len[b])
 *   mask : device uint8[B][K] (K <= 64), accept_len : device int32[B]
 * ------------------------------------------------------------------------ */
int sd_kv_append_masked(void* out_k, void* out_v,
                        const void* base_k, const void* base_v,
                        const void* draft_k, const void* draft_v,
                        const uint8_t* mask, const int32_t* accept_len,
                        int elem_size, int B, int H, int L, int K, int D,
                        int64_t base_sb, int64_t base_sh,
                        void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SPECDEC_HIP_H */
