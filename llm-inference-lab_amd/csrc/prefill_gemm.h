// Prompt prefill as library GEMMs + this repo's norm / epilogue / attention kernels (csrc/prefill_gemm.hip). Internal interface.
#pragma once

#include "kernels.h"

namespace sd {

constexpr int kPrefillChunk = 512;   // positions per GEMM chunk (workspace: ~110 KiB per position at Llama-3.2-3B dimensions)
// Shorter passes keep the decode-shaped kernels: the GEMM path's ~14 launches per layer cost ~4.5 ms for the 3B + 1B pair before the
// first product (160 tokens 5.9 ms, 192 6.2, 256 6.4, 512 8.9), the decode-shaped passes 5.0 / 7.1 / 9.3 ms at 64 / 96 / 128 tokens
// (profiles/round4_context_scaling.md): they cross below 96.
constexpr int kPrefillMinTokens = 96;

struct PrefillModel {
  const sd_model_config* cfg;
  uint16_t* k_cache;    // [layer][B][Hkv][Lmax][D]
  uint16_t* v_cache;    // [layer][B][Hkv][D][Lmax]
  int B, Lmax;
  float* attn_ws;       // split-KV workspace of the attention kernel
  unsigned* attn_cnt;
};

bool prefill_gemm_available();                               // rocBLAS could be opened (dlopen at first use)
size_t prefill_gemm_workspace_bytes(const sd_model_config& c);
int prefill_gemm_chunk(const PrefillModel& m, const int32_t* tokens, const int32_t* pos_base_row, int pos_off, int cache_row, int Mc, void* ws,
                       uint16_t** x_out, hipStream_t st);

}  // namespace sd
