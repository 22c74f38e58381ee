// K-token parallel attention over the appended KV cache (draft decode: M = 1..2,
// verify: M = K+1) for gfx950 — MFMA for the QK^T and PV contractions.
//
// Replaces what the reference gets from HF transformers' attention inside its k
// sequential forwards (hf_wrappers.py:417/478): here all M new positions of a row are
// scored against the cache in one pass, causally (query m sees keys 0 .. pos0+m).
//
// Cache layout (owned by the engine):
//   K: [B][Hkv][Lmax][D]   bf16 — a key is one contiguous row
//   V: [B][Hkv][D][Lmax]   bf16 — TRANSPOSED, so that 8 consecutive keys of one
//      channel are 16 contiguous bytes: exactly the MFMA operand shape of P·V.
//      The append writes 2-byte elements either way (gemv.hip QKV epilogue), so the
//      transpose is free at write time and saves an LDS transpose at read time.
//
// One workgroup = one (batch row, kv head, tile of 16 query rows); the query rows of a
// kv head are its G = Hq/Hkv query heads x M positions (15 rows for Llama-3.2-3B at
// K = 4: one MFMA tile). The 4 waves split the keys in blocks of 32 (flash-decoding
// inside the workgroup) and merge their (max, sum, O) partials through LDS.
// Per block of 32 keys a wave does
//   S^T[key][q] = K[key][:] . Q[q][:]      2 tiles x D/32  v_mfma_f32_16x16x32_bf16,
//                                          A = K rows loaded straight from HBM/L2
//                                          (16 B per lane), B = Q fragments (registers)
//   online softmax on the 8 scores each lane holds for ITS query row (the key order of
//   the two tiles is chosen so that a lane ends up with keys 8g..8g+7: its scores are,
//   unpermuted, the A fragment of the next MFMA — no LDS, no shuffles for P)
//   O[q][d] += P[q][key] . V[key][d]       D/16 MFMAs, B = V^T rows (16 B per lane)
// All K and V loads of a block are issued before any arithmetic: one memory round
// trip per block, and at decode lengths (<= 128 keys per wave) one per kernel.

#include <stdlib.h>

#include "attention_device.h"

namespace sd {

template <int D, int NW, bool PAGED>
__global__ __launch_bounds__(NW * 64) void attention_mfma_kernel(const AttnArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int ns = a.n_split > 1 ? a.n_split : 1;
  attention_tile<D, NW, PAGED>(a, blockIdx.x, blockIdx.y, blockIdx.z / ns, smem, true, blockIdx.z % ns);
}

template <int D, int NW, bool PAGED>
static void launch_attention_one_p(const AttnArgs& a, dim3 grid, hipStream_t st) {
  static unsigned long long attr_set = 0;
  const size_t smem = attention_smem_bytes(D, NW);
  if (smem > 64 * 1024) (void)opt_in_dynamic_lds(reinterpret_cast<const void*>(&attention_mfma_kernel<D, NW, PAGED>), static_cast<int>(smem), attr_set);   // (a failure shows at the launch check)
  hipLaunchKernelGGL((attention_mfma_kernel<D, NW, PAGED>), grid, dim3(NW * 64), smem, st, a);
}

template <int D, int NW>
static void launch_attention_one(const AttnArgs& a, dim3 grid, hipStream_t st) {
  if (a.block_table) launch_attention_one_p<D, NW, true>(a, grid, st);
  else launch_attention_one_p<D, NW, false>(a, grid, st);
}

template <int D>
static void launch_attention_d(const AttnArgs& a, int waves, dim3 grid, hipStream_t st) {
  (void)waves;   // always 4 (8- and 16-wave instances were measured and dropped, see launch_attention)
  launch_attention_one<D, 4>(a, grid, st);
  // (also tried: the first block's K/V loads hoisted above the row-position load (addresses are clamped, so they
  // are always safe): 4.79 -> 4.92 us (D = 64) and 5.9 -> 6.9 us (D = 128) per launch — waves whose first block
  // does not exist then wait for cold lines they do not need;
  // and: a wave issuing the loads of two 32-key blocks per pass — 5.05 vs 4.95 ms/step at short
  // contexts, 5.48 vs 5.40 at 512 keys: the pass is not bound by its memory round trip)
}

size_t attention_split_ws_bytes(int head_dim) {
  return static_cast<size_t>(kAttnSplitSlots) * kAttnRows * (head_dim + 2) * sizeof(float) + static_cast<size_t>(kAttnSplitSlots) * sizeof(unsigned);
}

int launch_attention(const AttnArgs& a_in, hipStream_t st) {
  AttnArgs a = a_in;
  SD_REQUIRE(a.head_dim == 32 || a.head_dim == 64 || a.head_dim == 128,
             "attention: head_dim %d not in {32,64,128}", a.head_dim);
  SD_REQUIRE(a.n_kv_heads > 0 && a.n_q_heads % a.n_kv_heads == 0, "attention: Hq %% Hkv != 0");
  SD_REQUIRE(a.B >= 1 && a.M >= 1, "attention: empty batch");
  SD_REQUIRE(a.l_max % 8 == 0 && a.l_max >= 8, "attention: l_max=%d must be a multiple of 8", a.l_max);
  SD_REQUIRE(!a.block_table || (a.page_shift >= 5 && a.page_shift <= 16 && a.l_max >= (1 << a.page_shift) && a.l_max % (1 << a.page_shift) == 0),
             "attention: paged cache needs pages of 32 * 2^n positions and l_max = max_pages * page_len");
  const int G = a.n_q_heads / a.n_kv_heads;
  const int R = G * a.M;
  const int tiles = (R + kAttnRows - 1) / kAttnRows;
  const int D = a.head_dim;
  // 4 waves per workgroup. 8 and 16 were measured and lose at every context length: the LDS merge grows with the wave
  // count and a 16-wave workgroup holds a CU's LDS alone (3B + 1B, K=4: 8 K context 6.8 ms/step with 4 waves, 9.3 with
  // 16; short contexts equal within noise).
  const int waves = 4;
  // split-KV: as many workgroups per tile as the cache could keep busy (one per 256 keys of l_max),
  // within the partial-tile workspace and ~one wave of workgroups over the chip; the kernel uses
  // fewer while the row is short. Fixed per launch site, so a captured step stays valid as rows grow.
  const int base = a.n_kv_heads * a.B * tiles;
  int ns = 1;
  if (a.split_ws && a.split_cnt && !getenv(debug_env::kNoAttnSplit)) {
    ns = a.l_max / (kAttnBlock * kAttnSplitBlocks);
    if (ns > kAttnMaxSplit) ns = kAttnMaxSplit;
    if (ns > a.split_slots / base) ns = a.split_slots / base;
    if (ns > 512 / base) ns = 512 / base;
    if (ns < 1) ns = 1;
  }
  a.n_split = ns;
  const dim3 grid(a.n_kv_heads, a.B, tiles * ns);
  switch (D) {
    case 32: launch_attention_d<32>(a, waves, grid, st); break;
    case 64: launch_attention_d<64>(a, waves, grid, st); break;
    default: launch_attention_d<128>(a, waves, grid, st); break;
  }
  SD_LAUNCH_CHECK();
  return 0;
}

}  // namespace sd
