// Weight-streaming skinny GEMM ("GEMV for T <= 9 tokens") for gfx950 decode/verify.
//
// y[t][n] = sum_k W[n][k] * x[t][k],  W bf16 [N][K] row-major (HF Linear layout),
// T = B*M tokens of one draft / verify forward (1..9). The weights are read ONCE
// from HBM per forward — this kernel IS the decode HBM roofline: its algorithmic
// bytes are N*K*2 and everything else (x, epilogue) is L2/LDS traffic.
//
// Structure (MI355X-first; the reference has no such kernel, its forward lives in HF
// transformers — hf_wrappers.py:417/478):
//   * v_mfma_f32_16x16x32_bf16 with the WEIGHTS as the A operand (16 rows x 32 k) and
//     the tokens as the B operand (32 k x 16 token columns, columns >= T are zero).
//     One MFMA retires 1 KiB of weights in 16 cycles, so the matrix pipe is idle most
//     of the time and the kernel stays bandwidth-bound for every T up to 16 (a first
//     VALU v_dot2c version was issue-bound at T = 5: v_dot2c_f32_bf16 is quarter rate);
//   * each lane loads its A fragment straight from HBM into VGPRs with one
//     global_load_dwordx4 (nt): lane (g = lane>>4, n = lane&15) reads 16 bytes of row n
//     at k = 32*step + 8*g, i.e. a wave-instruction covers 16 rows x 64 contiguous
//     bytes and consecutive steps continue each row. No LDS round trip for weights;
//   * latency is hidden by OCCUPANCY, not by a software pipeline: a workgroup is 16
//     waves (one per CU, 4 per SIMD); every wave issues a batch of up to 16
//     independent 1-KiB loads (16 KiB per wave, 256 KiB per CU in flight), waits for
//     them once, and runs its 16 MFMAs. (A two-buffer pipeline inside 4-wave
//     workgroups lost half its depth to hipcc's conservative vmcnt(0) at the loop
//     back-edge; this form needs no counted waits.)
//   * the T activation rows are staged ONCE per CU into LDS as bf16, with the
//     RMSNorm / LayerNorm fused into the staging pass; B fragments are ds_read_b128;
//   * work split: every workgroup owns a CONTIGUOUS range of ceil(pairs/256) row
//     pairs — equal bytes per CU for any N — cut into n_tiles tiles of <= 8 pairs
//     (16 MFMA rows); the 16 waves take (tile, K-slice) units, ksplit = 16 / n_tiles
//     slices per tile, and the slices' partial 16x16 tiles are summed through LDS
//     (one barrier per round of 16 units);
//   * a "pair" is what the fused epilogue needs in one place: rows (i, i+D/2) of a
//     head for RoPE, (gate_n, up_n) for SwiGLU, adjacent columns (2p, 2p+1) for the
//     residual / logits stores. Rows 0..7 of a tile are the first rows of its
//     pairs, rows 8..15 the second rows.

#include <stdlib.h>

#include "gemv_device.h"

namespace sd {

// Loads in flight per lane (template parameter KB) and how many of them are issued before the prologue (kPre).
// Swept on the whole step with a low-noise bench (3B + 1B, K=4, ms/step): KB = 4..12 -> 4.66 4.69 4.64 4.68 4.67 4.77
// 4.77 4.80 4.82 (14 spills): the many small launches of a step want a SHALLOW batch (less queueing in front
// of their few loads), while a long stream alone is ~5 % faster with 12 (lm_head 120 vs 126 us). So: 12 for matrices
// with >= kDeepSteps steps per wave, 6 otherwise (threshold swept on one box: never / 48 / 24 / 16 -> 4.65 / 4.65 /
// 4.61 / 4.67 ms per step).
constexpr int kBatchShallow = 6, kBatchDeep = 12;
constexpr int kPre = 8;
constexpr int kDeepSteps = 24, kDeepStepsVerify = 16;

// ------------------------------------------------------------------------------
// staging of x into LDS (bf16 [T][K + pad]) with the fused normalisation
// ------------------------------------------------------------------------------
__device__ __forceinline__ void stage_x(const GemvArgs& a, uint16_t* xs, int KP, float* red) {
  const int K = a.K, T = a.T;
  const int tid = threadIdx.x;
  const int nvec = K >> 3;
  const uint16_t* xin = static_cast<const uint16_t*>(a.x);
  const int lane = tid & 63, wave = tid >> 6;

  for (int t = 0; t < T; ++t) {
    const uint4* src = reinterpret_cast<const uint4*>(xin + static_cast<size_t>(a.x_row ? a.x_row[t] : t) * a.x_stride);
    uint4* dst = reinterpret_cast<uint4*>(xs + static_cast<size_t>(t) * KP);
    float s1 = 0.f, s2 = 0.f;
    for (int v = tid; v < nvec; v += kGemvThreads) {
      const uint4 q = src[v];
      dst[v] = q;
      if (a.prologue != PRO_NONE) {
        const uint32_t w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float lo = __uint_as_float(w[j] << 16), hi = __uint_as_float(w[j] & 0xffff0000u);
          s1 += lo + hi;
          s2 += lo * lo + hi * hi;
        }
      }
    }
    if (a.prologue != PRO_NONE) {
      s1 = wave_reduce_sum(s1);
      s2 = wave_reduce_sum(s2);
      if (lane == 0) {
        red[(t * kGemvWaves + wave) * 2 + 0] = s1;
        red[(t * kGemvWaves + wave) * 2 + 1] = s2;
      }
    }
  }
  __syncthreads();
  if (a.prologue == PRO_NONE) return;
  // pass 2: normalise in place. HF LlamaRMSNorm: weight * (x * rsqrt(var+eps)).to(bf16);
  // GPT-2 LayerNorm: (x-mean)*rsqrt(var+eps)*w + b computed in fp32, rounded once.
  const uint16_t* nw = static_cast<const uint16_t*>(a.norm_w);
  const uint16_t* nb = static_cast<const uint16_t*>(a.norm_b);
  const float invK = 1.0f / static_cast<float>(K);
  for (int t = 0; t < T; ++t) {
    const float2 pr = *reinterpret_cast<const float2*>(red + (t * kGemvWaves + (lane & 15)) * 2);
    const float sum = row16_reduce_sum(pr.x), sq = row16_reduce_sum(pr.y);
    uint16_t* row = xs + static_cast<size_t>(t) * KP;
    if (a.prologue == PRO_RMSNORM) {
      const float rs = rsqrtf(sq * invK + a.norm_eps);
      for (int k = tid * 2; k < K; k += kGemvThreads * 2) {
        const uint32_t xv = *reinterpret_cast<const uint32_t*>(row + k);
        const uint32_t wv = *reinterpret_cast<const uint32_t*>(nw + k);
        const float x0 = bf16_bits_to_float(float_to_bf16_bits(__uint_as_float(xv << 16) * rs));
        const float x1 = bf16_bits_to_float(float_to_bf16_bits(__uint_as_float(xv & 0xffff0000u) * rs));
        const uint16_t o0 = float_to_bf16_bits(x0 * __uint_as_float(wv << 16));
        const uint16_t o1 = float_to_bf16_bits(x1 * __uint_as_float(wv & 0xffff0000u));
        *reinterpret_cast<uint32_t*>(row + k) = static_cast<uint32_t>(o0) | (static_cast<uint32_t>(o1) << 16);
      }
    } else {  // PRO_LAYERNORM
      const float mean = sum * invK;
      const float var = fmaxf(sq * invK - mean * mean, 0.f);
      const float rs = rsqrtf(var + a.norm_eps);
      for (int k = tid * 2; k < K; k += kGemvThreads * 2) {
        const uint32_t xv = *reinterpret_cast<const uint32_t*>(row + k);
        const uint32_t wv = *reinterpret_cast<const uint32_t*>(nw + k);
        const uint32_t bv = *reinterpret_cast<const uint32_t*>(nb + k);
        const float y0 = (__uint_as_float(xv << 16) - mean) * rs * __uint_as_float(wv << 16) + __uint_as_float(bv << 16);
        const float y1 = (__uint_as_float(xv & 0xffff0000u) - mean) * rs * __uint_as_float(wv & 0xffff0000u) + __uint_as_float(bv & 0xffff0000u);
        *reinterpret_cast<uint32_t*>(row + k) =
            static_cast<uint32_t>(float_to_bf16_bits(y0)) | (static_cast<uint32_t>(float_to_bf16_bits(y1)) << 16);
      }
    }
  }
  __syncthreads();
}

// ------------------------------------------------------------------------------
// main kernel
//   MASK = false: every K slice is whole 32-element steps inside the row (all Llama
//   production shapes); MASK = true: any K % 8 == 0 (lanes past K load nothing).
// ------------------------------------------------------------------------------
//   TT = compile-time bound on the token count (1, 2, 3, 5 or 9): the prologue's loads are
//   unconditional straight-line code, which lets hipcc wait for them with a counted vmcnt
//   while the weight batch issued after them is still in flight.
//   W8 = the weights are OCP fp8 e4m3 in the packed order of csrc/pack.hip: a lane's 16-byte load
//   holds its A fragments of two consecutive 32-k steps; they are widened to bf16 in registers
//   (v_cvt_scalef32_pk_bf16_fp8, exact) and feed two bf16 MFMAs; the fp32 sum of row r is scaled by
//   w_scale[r] in the epilogue. Half the HBM bytes per token, bf16 activations unchanged.
template <int EPI, bool MASK, int TT, bool W8, int KB>
__global__ __launch_bounds__(kGemvThreads) void gemv_mfma_kernel(const GemvArgs a) {
  constexpr int kBatch = KB;
  pin_gemv_args<EPI, W8>(a);
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int K = a.K, T = a.T;
  const int KP = K + kXPad;
  uint16_t* xs = reinterpret_cast<uint16_t*>(smem);
  // LDS: [x rows][partials 16 x 16x16 f32][norm sums / argmax fold]. When x alone nearly
  // fills the CU's 160 KiB (T=5, K=14336) the partials alias the x rows (single-round
  // launches only; one extra barrier before they are written).
  const size_t xs_bytes = (static_cast<size_t>(T) * KP * 2 + 15) & ~static_cast<size_t>(15);
  float* part = reinterpret_cast<float*>(a.alias_part ? smem : smem + xs_bytes);  // [16 waves][16][16]
  float* red = reinterpret_cast<float*>(smem + xs_bytes + (a.alias_part ? 0 : sizeof(float) * kGemvWaves * 256));

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave-uniform, kept in SGPRs
  const int g = lane >> 4, n = lane & 15;
  const int ksplit = a.ksplit;                 // power of two, <= 16
  const int tiles_per_round = kGemvWaves >> a.ks_shift;
  const int kpart = wave & (ksplit - 1);
  const int tslot = wave >> a.ks_shift;
  const int kw = a.kw;                         // K span of one slice (multiple of 32)
  const int k_begin = kpart * kw;
  const int steps = W8 ? (kw >> 6) : (kw >> 5);  // loads per slice (an fp8 load covers 64 k)
  const uint16_t* W = static_cast<const uint16_t*>(a.W);
  const float* wsc = a.w_scale;
  size_t part_off = 0;
  if constexpr (EPI == EPI_ARGMAX) {
    if (a.batch_bytes) {   // Medusa heads: matrix blockIdx.y of n_batch, same geometry, same x rows
      W = reinterpret_cast<const uint16_t*>(reinterpret_cast<const char*>(W) + blockIdx.y * a.batch_bytes);
      if constexpr (W8) wsc = reinterpret_cast<const float*>(reinterpret_cast<const char*>(wsc) + blockIdx.y * a.batch_bytes);
      part_off = static_cast<size_t>(blockIdx.y) * T * gridDim.x;
    }
  }

  // contiguous pair range of this workgroup, cut into n_tiles tiles of tile_pairs pairs
  const int p_lo = static_cast<int>(blockIdx.x) * a.ppw;
  const int p_hi = min(p_lo + a.ppw, a.n_pairs);
  const int tile_pairs = a.tile_pairs;
  // (only the last workgroup can own less than ppw pairs: everyone else takes the host's tile count, no division)
  const int n_tiles = (p_lo + a.ppw <= a.n_pairs) ? a.n_tiles_full : (p_hi - p_lo + tile_pairs - 1) / tile_pairs;
  const int rounds = (n_tiles + tiles_per_round - 1) >> (4 - a.ks_shift);

  // lane's weight row inside a tile: rows 0..7 = first rows of the pairs, 8..15 = second
  // rows. Lanes without a pair alias the tile's first row (same address as lane 0: no
  // extra traffic); their results are never read.
  // Address of a lane's A fragment = wave-uniform base (SGPRs) + 32-bit lane offset (one VGPR)
  // + step * wstride. Packed weights (csrc/pack.hip): every 32-k step of a tile is one
  // contiguous block [g][row] of 2*np*64 bytes; row-major: the lane offset selects the row.
  int wstride = 32;  // elements between a lane's consecutive k-steps
  const int K32 = (K + 31) & ~31;
  unsigned lane_off = 0;
  auto tile_base = [&](int tile) -> const uint16_t* {
    const int p0 = p_lo + tile * tile_pairs;
    if (a.packed) {
      int np = min(tile_pairs, p_hi - p0);
      if (np < 1) np = 1;
      int jp = n & 7, second = n >> 3;
      if (jp >= np) { jp = 0; second = 0; }  // alias a valid lane: same address, no extra traffic
      wstride = np * 64;
      lane_off = static_cast<unsigned>((g * 2 * np + second * np + jp) * 8);
      if constexpr (W8) return W + static_cast<size_t>(p0) * ((K + 63) & ~63) + static_cast<size_t>(k_begin >> 6) * wstride;  // 1 byte per weight
      return W + static_cast<size_t>(p0) * 2 * K32 + static_cast<size_t>(k_begin >> 5) * wstride;
    }
    int p = p0 + (n & 7);
    int second = n >> 3;
    if ((n & 7) >= tile_pairs || p >= p_hi) { p = min(p0, p_hi - 1); second = 0; }
    int r0, r1;
    pair_rows<EPI>(a, p, r0, r1);
    int r = second ? r1 : r0;
    if (r >= a.N) r = r0;  // odd N (vocabulary): second row of the last pair
    lane_off = static_cast<unsigned>(r) * static_cast<unsigned>(K) + static_cast<unsigned>(g * 8);
    return W + k_begin;
  };

  u32x4 buf[kBatch];
  // first batch: unconditional (steps past the slice re-read its last step; never used). It is
  // issued in two parts: kPre loads per wave before the prologue (64 KiB per CU: enough to keep
  // HBM busy, few enough not to fill the CU's memory queue, which would stall the waves at
  // issue and hold up the prologue's barriers), the rest right after it.
  auto issue_first = [&](const uint16_t* ubase, int j0, int j1) {
#pragma unroll
    for (int j = 0; j < kBatch; ++j) {
      if (j >= j0 && j < j1) {
        // (32-bit element offsets from the wave-uniform tile base: a workgroup's share of a matrix is far below 2^31 elements; as
        //  size_t products every load of the batch cost ~11 scalar instructions of 64-bit address arithmetic in the launch's preamble)
        const unsigned s = (j < steps) ? j : steps - 1;
        buf[j] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(ubase) + (s * static_cast<unsigned>(wstride) + lane_off) * 2u));
      }
    }
  };
  auto issue = [&](const uint16_t* ubase, int s0) {
#pragma unroll
    for (int j = 0; j < kBatch; ++j) {
      const int s = s0 + j;
      bool ok = s < steps;  // wave-uniform
      if constexpr (MASK && !W8) ok = ok && (a.packed ? (k_begin + s * 32 < K32) : (k_begin + s * 32 + g * 8 + 8 <= K));
      if (ok) buf[j] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(ubase) + (static_cast<unsigned>(s) * static_cast<unsigned>(wstride) + lane_off) * 2u));
      else buf[j] = u32x4{0u, 0u, 0u, 0u};
    }
  };

  // diagnostic timeline (sd_model_probe_gemv with SPECDEC_GEMV_TIMELINE=1): 100 MHz stamps
  // of wave 0 / lane 0 into a buffer nothing else reads; no stamp executes otherwise
  auto stamp = [&](int slot) {
    if (a.debug_ts && tid == 0) a.debug_ts[static_cast<size_t>(blockIdx.x) * 8 + slot] = __builtin_amdgcn_s_memrealtime();
  };
  stamp(0);
  // per-row adaptive K: a launch of a draft forward no row needs leaves here — after the index arithmetic above, which
  // runs under the latency of the argument loads (at kernel entry the branch kept it behind them: 0.5 % of the step)
  SD_SKIP_IF_INACTIVE(a.skip_k, a.skip_i);

  // ---- x staging, fast path (K <= 8192): thread `tid` owns the 16-byte chunk `tid` of every
  // token row. Its loads (x rows, norm weights, the residual value of its epilogue item) are
  // issued BEFORE the weight batch: vmcnt completes in order, so they return first and the
  // whole prologue runs while the weights are still in flight.
  const int nvec = K >> 3;
  constexpr bool fast_stage = !MASK;  // the launcher sends K > 8192 to the MASK variant
  const bool has_chunk = tid < nvec;
  const int cidx = has_chunk ? tid : nvec - 1;  // clamp: every thread loads, owners store
  u32x4 xr[TT];
  u32x4 nw4 = {0u, 0u, 0u, 0u}, nb4 = {0u, 0u, 0u, 0u};
  bool have_old = false;
  uint32_t old_pre = 0;
  if constexpr (fast_stage) {
    const uint16_t* xin = static_cast<const uint16_t*>(a.x);
    if (a.x_row) {   // kernel-uniform: gathered rows (Medusa heads read the accepted position's hidden row)
      int rows[TT];
#pragma unroll
      for (int t = 0; t < TT; ++t) rows[t] = a.x_row[(t < T) ? t : T - 1];
#pragma unroll
      for (int t = 0; t < TT; ++t) xr[t] = *reinterpret_cast<const u32x4*>(xin + static_cast<size_t>(rows[t]) * a.x_stride + cidx * 8);
    } else {
#pragma unroll
      for (int t = 0; t < TT; ++t) {
        const int tt = (t < T) ? t : T - 1;
        xr[t] = *reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(xin) + (static_cast<unsigned>(tt) * static_cast<unsigned>(a.x_stride) + static_cast<unsigned>(cidx) * 8u) * 2u);   // (32-bit offsets: <= 64 rows of <= 16384 elements)
      }
    }
    if (a.prologue != PRO_NONE) {  // kernel-uniform
      nw4 = *reinterpret_cast<const u32x4*>(static_cast<const uint16_t*>(a.norm_w) + cidx * 8);
      if (a.prologue == PRO_LAYERNORM) nb4 = *reinterpret_cast<const u32x4*>(static_cast<const uint16_t*>(a.norm_b) + cidx * 8);
    }
    // residual epilogue: the old x value of this thread's (pair, token) item of round 0
    if constexpr (EPI == EPI_RESID) {
      const int ts = tid >> 7, jp = (tid >> 4) & 7, t = tid & 15;
      const int p = p_lo + ts * tile_pairs + jp;
      have_old = ts < tiles_per_round && ts < n_tiles && jp < tile_pairs && p < p_hi && t < T;
      const int pc = have_old ? p : p_lo, tc = have_old ? t : 0;
      old_pre = *reinterpret_cast<const uint32_t*>(reinterpret_cast<const char*>(a.out) + (static_cast<unsigned>(tc) * static_cast<unsigned>(a.out_stride) + 2u * static_cast<unsigned>(pc)) * 2u);
    }
  }

  // start the weight stream: its HBM latency hides under the rest of the prologue
  const bool first_valid = tslot < n_tiles;
  const uint16_t* wrow0 = tile_base(first_valid ? tslot : 0);
  if constexpr (MASK) {
    if (first_valid) issue(wrow0, 0);
  } else {
    issue_first(wrow0, 0, kPre);  // waves without a tile re-read tile 0's first steps (L2 hits)
  }
  stamp(1);

  if constexpr (!fast_stage) {
    stage_x(a, xs, KP, red);
  } else if (a.prologue == PRO_NONE) {
#pragma unroll
    for (int t = 0; t < TT; ++t)
      if (t < T && has_chunk) *reinterpret_cast<u32x4*>(xs + static_cast<size_t>(t) * KP + tid * 8) = xr[t];
    __syncthreads();
  } else {
    // per-row sum / sum of squares: own chunk -> wave -> LDS -> every thread. Only the first nvec/64 waves own
    // chunks (6 of 16 at K = 3072); the others skip the arithmetic — all 16 waves running the ~100 VALU
    // instructions per token on clamped duplicates made the prologue VALU-bound (4 us at 5 tokens).
    const bool wave_has_chunk = (wave << 6) < nvec;   // wave-uniform
#pragma unroll
    for (int t = 0; t < TT; ++t) {
      if (t < T) {
        float s1 = 0.f, s2 = 0.f;
        if (wave_has_chunk) {
          const bool need_mean = a.prologue == PRO_LAYERNORM;   // kernel-uniform: RMSNorm only needs the squares
          f32x2_t a1 = {0.f, 0.f}, a2 = {0.f, 0.f};
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const f32x2_t v = bf16x2_unpack(xr[t][j]);
            if (need_mean) a1 += v;
            a2 += v * v;
          }
          s1 = a1.x + a1.y;
          s2 = a2.x + a2.y;
          if (!has_chunk) { s1 = 0.f; s2 = 0.f; }  // clamped (duplicate) loads do not count
          if (need_mean) s1 = wave_reduce_sum(s1);
          s2 = wave_reduce_sum(s2);
        }
        if (lane == 0) {
          red[(t * kGemvWaves + wave) * 2 + 0] = s1;
          red[(t * kGemvWaves + wave) * 2 + 1] = s2;
        }
      }
    }
    __syncthreads();
    const float invK = 1.0f / static_cast<float>(K);
#pragma unroll
    for (int t = 0; t < TT; ++t) {
      if (t < T && wave_has_chunk) {
        // the 16 per-wave partials: lane l reads partial l & 15 (one 8-byte LDS read), rows of 16 lanes add up
        const float2 pr = *reinterpret_cast<const float2*>(red + (t * kGemvWaves + (lane & 15)) * 2);
        const float sum = row16_reduce_sum(pr.x), sq = row16_reduce_sum(pr.y);
        u32x4 o;
        if (a.prologue == PRO_RMSNORM) {
          // HF LlamaRMSNorm: weight * (x * rsqrt(var + eps)).to(bf16)
          const float rs = rsqrtf(sq * invK + a.norm_eps);
#pragma unroll
          for (int j = 0; j < 4; ++j) o[j] = rmsnorm_pair(xr[t][j], rs, nw4[j]);   // packed math (common.h): same bits, half the VALU work
        } else {
          // GPT-2 LayerNorm in fp32, rounded once
          const float mean = sum * invK;
          const float rs = rsqrtf(fmaxf(sq * invK - mean * mean, 0.f) + a.norm_eps);
#pragma unroll
          for (int j = 0; j < 4; ++j) o[j] = layernorm_pair(xr[t][j], mean, rs, nw4[j], nb4[j]);
        }
        if (has_chunk) *reinterpret_cast<u32x4*>(xs + static_cast<size_t>(t) * KP + tid * 8) = o;
      }
    }
    __syncthreads();
  }
  if constexpr (!MASK) issue_first(wrow0, kPre, kBatch);
  stamp(2);

  float best_v = -INFINITY;
  int best_i = 0x7fffffff;
  // token columns >= T read token T-1's row: their output columns are never used
  const uint16_t* xrow = xs + static_cast<size_t>(n < T ? n : T - 1) * KP + k_begin + g * 8;

  for (int r = 0; r < rounds; ++r) {
    const int tile = r * tiles_per_round + tslot;
    f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
    // (Round 2, measured and removed: "rolling halves" — the batch as two register sets, one multiplied while the other is in
    // flight, the issue order pinned with scheduling barriers as in gemm_pipe.hip, so that a wave keeps KB/2..KB loads in
    // flight instead of the 0..KB saw-tooth below: 4.55 against 4.49 ms/step at 1 row, 6.05 against 6.04 at 8 rows. With 16
    // waves per CU the saw-teeth of the waves interleave already; the pinned order only costs MFMA / LDS slack.)
    if (tile < n_tiles) {
      const uint16_t* wrow = (r == 0) ? wrow0 : tile_base(tile);
      for (int s0 = 0; s0 < steps; s0 += kBatch) {
        if (r != 0 || s0 != 0) issue(wrow, s0);
#pragma unroll
        for (int j = 0; j < kBatch; ++j) {
          const int s = s0 + j;
          if (s < steps) {
            if constexpr (W8) {
              // 16 fp8 -> two bf16x8 fragments (k = 64 s + 8 g + 0..7 and + 32)
              u32x4 lo, hi;
#pragma unroll
              for (int e = 0; e < 2; ++e) {
                lo[2 * e] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(buf[j][e], 1.0f, false));
                lo[2 * e + 1] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(buf[j][e], 1.0f, true));
                hi[2 * e] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(buf[j][2 + e], 1.0f, false));
                hi[2 * e + 1] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(buf[j][2 + e], 1.0f, true));
              }
              const u32x4 xb0 = *reinterpret_cast<const u32x4*>(xrow + s * 64);
              const u32x4 xb1 = *reinterpret_cast<const u32x4*>(xrow + s * 64 + 32);
              acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, lo), __builtin_bit_cast(bf16x8_t, xb0), acc, 0, 0, 0);
              acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, hi), __builtin_bit_cast(bf16x8_t, xb1), acc, 0, 0, 0);
            } else {
              u32x4 xb;
              if constexpr (MASK) {
                const bool ok = (k_begin + s * 32 + g * 8 + 8 <= K);
                xb = ok ? *reinterpret_cast<const u32x4*>(xrow + s * 32) : u32x4{0u, 0u, 0u, 0u};
              } else {
                xb = *reinterpret_cast<const u32x4*>(xrow + s * 32);
              }
              acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, buf[j]),
                                                            __builtin_bit_cast(bf16x8_t, xb), acc, 0, 0, 0);
            }
          }
        }
      }
    }
    if (r == 0) stamp(3);
    // publish this wave's partial 16x16 (row = 4g+reg, col = token n)
    if (a.alias_part) __syncthreads();  // every wave is done reading x
    float* slot = part + wave * 256;
#pragma unroll
    for (int q = 0; q < 4; ++q) slot[(4 * g + q) * 16 + n] = acc[q];
    __syncthreads();
    if (r == 0) stamp(4);
    // epilogue items: thread -> (tile slot, pair, token); sums the ksplit slices
    for (int it = tid; it < tiles_per_round * 128; it += kGemvThreads) {
      const int ts = it >> 7, jp = (it >> 4) & 7, t = it & 15;  // t == tid & 15 on every trip
      const int etile = r * tiles_per_round + ts;
      const int p = p_lo + etile * tile_pairs + jp;
      if (etile < n_tiles && jp < tile_pairs && p < p_hi && t < T) {
        float y0, y1;
        sum_slices(part + (ts * ksplit) * 256 + jp * 16 + t, 256, ksplit, y0, y1);   // (gemv_device.h)
        int r0, r1;
        pair_rows<EPI>(a, p, r0, r1);
        if constexpr (W8) {
          y0 *= wsc[r0];
          y1 *= (r1 < a.N) ? wsc[r1] : 0.f;
        }
        epilogue<EPI>(a, p, r0, r1, t, y0, y1, best_v, best_i, have_old && r == 0 && it == tid, old_pre);
      }
    }
    if (r == 0) stamp(5);
    if (r + 1 < rounds) __syncthreads();  // partial slots are rewritten next round
  }
  stamp(6);

  if constexpr (EPI == EPI_ARGMAX) {
    // thread tid holds a running best for token tid & 15 (pairs (tid>>4)&7 of its tile
    // slot): fold the 64 candidates per token through LDS
    __syncthreads();
    float* sv = part;                                  // [16 tokens][64]
    int* si = reinterpret_cast<int*>(part + 16 * 64);
    sv[(tid & 15) * 64 + (tid >> 4)] = best_v;
    si[(tid & 15) * 64 + (tid >> 4)] = best_i;
    __syncthreads();
    if (wave < T) {  // wave t folds token t
      float bv = sv[wave * 64 + lane];
      int bi = si[wave * 64 + lane];
      wave_reduce_argmax(bv, bi);
      if (lane == 0) {
        a.part_val[part_off + static_cast<size_t>(wave) * gridDim.x + blockIdx.x] = bv;
        a.part_idx[part_off + static_cast<size_t>(wave) * gridDim.x + blockIdx.x] = bi;
      }
    }
  }
}

// ------------------------------------------------------------------------------
// host-side launch
// ------------------------------------------------------------------------------
constexpr size_t kLdsLimit = 160 * 1024;
static size_t gemv_smem(int T, int K, bool alias) {
  const size_t xs = (static_cast<size_t>(T) * (K + kXPad) * 2 + 15) & ~static_cast<size_t>(15);
  const size_t part = sizeof(float) * kGemvWaves * 256;
  const size_t red = sizeof(float) * (kGemvMaxT * kGemvWaves * 2 + 64);
  return (alias ? (xs > part ? xs : part) : xs + part) + red;
}

// most tokens one launch can stage for rows of K elements (x rows + aliased partials in 160 KiB)
int gemv_max_tokens(int K) {
  int t = kGemvMaxT;
  while (t > 1 && gemv_smem(t, K, true) > kLdsLimit) --t;
  return t;
}

int gemv_grid(const GemvArgs& a, int* ppw_out) {
  const GemvGeom q = gemv_geometry(a.n_pairs, a.K);
  *ppw_out = q.ppw;
  return q.grid;
}

template <int EPI, bool MASK, int TT, bool W8, int KB>
static int launch_one(const GemvArgs& a, int grid, size_t smem, hipStream_t st) {
  // dynamic LDS above 64 KiB has to be opted into once per kernel
  static unsigned long long attr_set = 0;
  if (int rc = opt_in_dynamic_lds(reinterpret_cast<const void*>(&gemv_mfma_kernel<EPI, MASK, TT, W8, KB>), 160 * 1024, attr_set)) return rc;  // whole LDS of the CU
  hipLaunchKernelGGL((gemv_mfma_kernel<EPI, MASK, TT, W8, KB>), dim3(grid, a.batch_bytes ? a.n_batch : 1), dim3(kGemvThreads), smem, st, a);
  SD_LAUNCH_CHECK();
  return 0;
}

template <int EPI, bool W8, int KB>
static int launch_epi_w(const GemvArgs& a, bool mask, int grid, size_t smem, hipStream_t st) {
  if (mask) return launch_one<EPI, true, kGemvMaxT, W8, KB>(a, grid, smem, st);  // generic shapes: one variant
  if (a.T <= 1) return launch_one<EPI, false, 1, W8, KB>(a, grid, smem, st);
  if (a.T <= 2) return launch_one<EPI, false, 2, W8, KB>(a, grid, smem, st);
  if (a.T <= 3) return launch_one<EPI, false, 3, W8, KB>(a, grid, smem, st);
  if (a.T <= 5) return launch_one<EPI, false, 5, W8, KB>(a, grid, smem, st);
  return launch_one<EPI, false, kGemvMaxT, W8, KB>(a, grid, smem, st);
}

template <int EPI>
static int launch_epi(const GemvArgs& a, bool mask, int grid, size_t smem, hipStream_t st) {
  // (3+ token launches, i.e. the verify forward: 16 — threshold 24 / 16 / 12 / 6 -> 4.62 / 4.60 / 4.62 / 4.63 ms per step)
  const int nsteps = a.kw >> (a.w8 ? 6 : 5);   // loads per wave over the whole K slice
  const bool deep = nsteps >= (a.T >= 3 ? kDeepStepsVerify : kDeepSteps);
  if (a.w8) return deep ? launch_epi_w<EPI, true, kBatchDeep>(a, mask, grid, smem, st) : launch_epi_w<EPI, true, kBatchShallow>(a, mask, grid, smem, st);
  return deep ? launch_epi_w<EPI, false, kBatchDeep>(a, mask, grid, smem, st) : launch_epi_w<EPI, false, kBatchShallow>(a, mask, grid, smem, st);
}

int launch_gemv(const GemvArgs& a_in, int epi, hipStream_t st) {
  if (a_in.T > kGemvMaxT) {
    SD_REQUIRE(!a_in.batch_bytes, "gemv: batched matrices need T <= %d", kGemvMaxT);
    return launch_gemm_skinny(a_in, epi, st);
  }
  GemvArgs a = a_in;
  SD_REQUIRE(a.T >= 1 && a.T <= kGemvMaxT, "gemv: T=%d out of range 1..%d", a.T, kGemvMaxT);
  SD_REQUIRE(a.K % 8 == 0 && a.x_stride % 8 == 0, "gemv: K=%d / x_stride=%d must be multiples of 8", a.K, a.x_stride);
  SD_REQUIRE(a.n_pairs > 0, "gemv: no rows");
  SD_REQUIRE(!a.batch_bytes || (epi == EPI_ARGMAX && a.n_batch >= 1 && a.n_batch <= 64 && !a.out), "gemv: batched matrices are an ARGMAX-epilogue feature (no logits store)");
  // one workgroup (16 waves) per CU with an equal, contiguous share of the row pairs, cut into
  // a power-of-two count of <= 8-pair tiles; the 16 waves take (tile, K-slice) units
  const GemvGeom q = gemv_geometry(a.n_pairs, a.K);
  const int grid = q.grid, n_tiles = q.n_tiles, ksplit = q.ksplit;
  a.ppw = q.ppw;
  a.tile_pairs = q.tile_pairs;
  a.ksplit = q.ksplit;
  a.kw = q.kw;
  a.n_tiles_full = (q.ppw + q.tile_pairs - 1) / q.tile_pairs;
  gemv_derive(a);
  // MASK variant: slices that are not whole 32-k steps, or rows longer than the one-chunk-per-
  // thread register staging covers
  const bool mask = (a.kw * ksplit != a.K) || (a.K / 8 > kGemvThreads);
  if (a.w8) {
    SD_REQUIRE(a.packed && a.w_scale, "gemv: fp8 weights need the packed layout and row scales");
    SD_REQUIRE(a.K % 64 == 0 && a.kw % 64 == 0 && a.kw * ksplit == a.K, "gemv: fp8 weights need K (%d) and the K slice (%d) in whole 64-k steps", a.K, a.kw);
  }
  size_t smem = gemv_smem(a.T, a.K, false);
  a.alias_part = 0;
  if (smem > kLdsLimit && n_tiles <= kGemvWaves / ksplit) {  // single round: partials may alias x
    a.alias_part = 1;
    smem = gemv_smem(a.T, a.K, true);
  }
  // rows too long to stage T of them whole (d_ff = 14336 takes 5): the chunked multi-token kernel covers the rest
  // of 6..9 tokens — a pass may mix both kernels, they share layout, work split and epilogues
  if (smem > kLdsLimit && !a.x_row && !a.batch_bytes && gemm_skinny_covers(a.T, a.n_pairs, a.K, a.w8 != 0)) return launch_gemm_skinny(a_in, epi, st);
  SD_REQUIRE(smem <= kLdsLimit, "gemv: T=%d x K=%d does not fit LDS", a.T, a.K);
  switch (epi) {
    case EPI_QKV_ROPE: return launch_epi<EPI_QKV_ROPE>(a, mask, grid, smem, st);
    case EPI_RESID: return launch_epi<EPI_RESID>(a, mask, grid, smem, st);
    case EPI_SWIGLU: return launch_epi<EPI_SWIGLU>(a, mask, grid, smem, st);
    case EPI_GELU: return launch_epi<EPI_GELU>(a, mask, grid, smem, st);
    case EPI_ARGMAX: return launch_epi<EPI_ARGMAX>(a, mask, grid, smem, st);
    default: SD_REQUIRE(false, "gemv: unknown epilogue %d", epi);
  }
  return 0;
}

}  // namespace sd
