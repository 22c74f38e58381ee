// Weight-streaming skinny GEMM for 10..64 tokens as a statically scheduled chunk pipeline (gfx950).
//
// Same contract, weight layout (row-major or csrc/pack.hip tile streams), work split (gemv_geometry), chunking
// of K and fused prologue / epilogues as gemm_skinny.hip — this is its replacement for the shapes the batched
// verify pass and the multi-row draft passes run (the reference has no such kernel: its verify is K sequential HF
// forwards per row, speculative_scheduler.py:192-199). What changed, and why:
//
//   * gemm_skinny_kernel issued every weight load under a condition (`if (s < steps)`); each conditional load
//     became its own basic block and hipcc put an `s_waitcnt vmcnt(0)` in front of it. A wave therefore never had
//     more than one or two 1-KiB loads in flight and the kernel streamed at latency, not bandwidth: 2.8 TB/s
//     for the 3B gate/up at 40 tokens. Here every load is UNCONDITIONAL (indices clamped to the last valid step /
//     chunk; the surplus loads hit L2) and the issue order is pinned with scheduling barriers, so the waits are
//     counted (`vmcnt(n)`, n = loads of the younger set) and two whole chunk sets per wave stay in flight.
//   * a chunk boundary was barrier -> L2 round trip for the x chunk -> normalise -> barrier, with the weight
//     stream drained across it. Here the x chunk is DOUBLE-BUFFERED in LDS: the rows of chunk c+1 are loaded
//     into registers when chunk c starts, written (normalised) to the other buffer when chunk c's MFMAs are
//     done, and ONE barrier per chunk publishes them.
//   * the prologue was three dependent round trips with no weight traffic behind them (row statistics x2, then
//     chunk 0). Here the raw rows of chunk 0 and the first weight set are issued FIRST and the statistics pass
//     runs under them, balanced over the waves (segments of 512 16-byte pieces) instead of whole rows per wave.
//   * the normalisation (every workgroup normalises all T x K activations: ~8 us of VALU time at 40 tokens) uses the
//     packed-math path (common.h: rmsnorm_pair): half the instructions, same bits.
//   (Tried and dropped: RMSNorm with the row scale deferred to the epilogue — no statistics pass at all, 35 -> 30 us at 40
//   tokens — because it moves a rounding point: bf16(x * w) instead of HF's bf16(bf16(x * rstd) * w); the logits' RMS
//   distance to the oracle went from 0.9 % to 1.5 %. Parity first.)
//
// Work of a wave: its (tile, K slice) unit walks the chunks; inside chunk c it owns SC steps of 32 k (64 k for fp8
// storage). SC is a template parameter so that a set of weights is a fixed-size register array.

#include <stdlib.h>

#include "gemv_device.h"

namespace sd {

struct PipeGeom {
  int kc;        // chunk width in columns: divides K, = KS * ksplit * SC
  int nchunks;   // K / kc
};

static __host__ __device__ size_t pipe_x_bytes(int T, int kc) { return (static_cast<size_t>(T) * (kc + kXPad) * 2 + 15) & ~static_cast<size_t>(15); }
static __host__ __device__ size_t pipe_part_bytes(int TG) { return sizeof(float) * kGemvWaves * TG * 256; }
static size_t pipe_smem(int T, int TG, int kc) {
  const size_t xs = 2 * pipe_x_bytes(T, kc), part = pipe_part_bytes(TG);
  return (xs > part ? xs : part) + sizeof(float) * 2 * 64 + sizeof(float) * kGemvWaves * 4 * 32;   // + stat[] + row-statistics scratch
}

#define SD_PIN_ORDER()                 \
  do {                                 \
    asm volatile("" ::: "memory");     \
    __builtin_amdgcn_sched_barrier(0); \
  } while (0)

// Row statistics (mean, rstd) of the T rows into stat[] — its own function, NOT inlined: the statistics need a few dozen
// registers for a microsecond, and inlined into the kernel they pushed values that live across the K loop (accumulators,
// fragment offsets, the rows in flight) into scratch, whose reloads then sat in the loop, in order behind the weight
// loads. A call keeps the damage where it happens. Called by every thread of the workgroup, before anything is in flight.
template <int TG>
__device__ __noinline__ void pipe_row_stats(const uint16_t* xin, int T, int K, int prologue, float norm_eps, const float* xstat_in,
                                            int xstat_n, float* stat, unsigned char* smem) {
  // (scalars, not the argument struct: a by-reference struct would be copied to the stack for the call)
  struct { int prologue; float norm_eps; const float* xstat_in; int xstat_n; } a{prologue, norm_eps, xstat_in, xstat_n};
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (a.xstat_in) {
      // the launch that wrote these rows left one partial (sum of squares, sum) per token and workgroup: a wave takes
      // tokens wave, wave + 16, ...; lane l adds partials l, l + 64, l + 128, l + 192, the wave folds: one round trip
      // of <= 1 KiB per token instead of the whole row, same order in every workgroup and every run
      const float invK = 1.0f / static_cast<float>(K);
      const int np = a.xstat_n;
      auto fold = [&](const float* plane, float (&out)[TG]) {     // out[u] = sum of the np partials of token wave + 16 u
        float pp[TG][4];
#pragma unroll
        for (int u = 0; u < TG; ++u) {
          const int t = wave + u * kGemvWaves;
          const float* src = plane + static_cast<size_t>(t < T ? t : T - 1) * kStatStride;
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int idx = lane + i * kWave;
            pp[u][i] = src[idx < np ? idx : 0];
          }
        }
#pragma unroll
        for (int u = 0; u < TG; ++u) {
          float v = 0.f;
#pragma unroll
          for (int i = 0; i < 4; ++i) v += (lane + i * kWave < np) ? pp[u][i] : 0.f;
          out[u] = wave_reduce_sum(v);
        }
      };
      float s2[TG];
      fold(a.xstat_in, s2);
      if (a.prologue == PRO_RMSNORM) {
#pragma unroll
        for (int u = 0; u < TG; ++u) {
          const int t = wave + u * kGemvWaves;
          if (lane == 0 && t < T) {
            stat[2 * t] = 0.f;
            stat[2 * t + 1] = rsqrtf(s2[u] * invK + a.norm_eps);
          }
        }
      } else {
        float s1[TG];
        fold(a.xstat_in + kStatPlane, s1);
#pragma unroll
        for (int u = 0; u < TG; ++u) {
          const int t = wave + u * kGemvWaves;
          if (lane == 0 && t < T) {
            const float mean = s1[u] * invK;
            stat[2 * t] = mean;
            stat[2 * t + 1] = rsqrtf(fmaxf(s2[u] * invK - mean * mean, 0.f) + a.norm_eps);
          }
        }
      }
  } else {
      // Row statistics over the T x K block, balanced over the waves: x is contiguous per row, so the block is cut into
      // SEGMENTS of 512 16-byte pieces (8 per lane = 32 registers); segment s = 16 * trip + wave. A segment touches the
      // tail of one row and the head of the next one(s); per touched row the wave reduces its lanes' masked sums and
      // leaves the partial in LDS, and row t finally adds its partials in segment order (deterministic). T = 40, K = 3072:
      // two balanced trips of 8 loads per lane (rows split two-per-wave took the same two trips with 64 registers: spills).
      const int nvec = K >> 3;                        // 16-byte pieces per row (x_stride == K is required by the launcher)
      const int total = T * nvec;
      const float invK = 1.0f / static_cast<float>(K);
      constexpr int kSeg = 512;
      const int nseg = (total + kSeg - 1) / kSeg;
      float2* pstat = reinterpret_cast<float2*>(smem);   // [nseg][kMaxRowsPerSeg] partial (sum, sum of squares); chunk buffers are still unused
      constexpr int kMaxRowsPerSeg = 5;                  // 512 pieces over rows of >= 128 pieces (K >= 1024, launcher)
      for (int sg = wave; sg < nseg; sg += kGemvWaves) {
        const int b0 = sg * kSeg;
        u32x4 q[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int bi = b0 + i * kWave + lane;
          q[i] = *reinterpret_cast<const u32x4*>(xin + static_cast<size_t>(bi < total ? bi : total - 1) * 8);
        }
        const int row_lo = b0 / nvec, row_hi = min(b0 + kSeg - 1, total - 1) / nvec;
        for (int row = row_lo; row <= row_hi; ++row) {     // wave-uniform, <= kMaxRowsPerSeg trips
          const int lo_b = row * nvec, hi_b = lo_b + nvec;
          f32x2_t a1 = {0.f, 0.f}, a2 = {0.f, 0.f};
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            const int bi = b0 + i * kWave + lane;
            const bool in = bi >= lo_b && bi < hi_b && bi < total;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const f32x2_t v = bf16x2_unpack(in ? q[i][j] : 0u);
              a1 += v;
              a2 += v * v;
            }
          }
          float s1 = a1.x + a1.y, s2 = a2.x + a2.y;
          s1 = wave_reduce_sum(s1);
          s2 = wave_reduce_sum(s2);
          if (lane == 0) pstat[sg * kMaxRowsPerSeg + (row - row_lo)] = float2{s1, s2};
        }
      }
      __syncthreads();
      if (tid < T) {
        const int t = tid;
        const int s_lo = (t * nvec) / kSeg, s_hi = (t * nvec + nvec - 1) / kSeg;
        float s1 = 0.f, s2 = 0.f;
        for (int sg = s_lo; sg <= s_hi; ++sg) {
          const float2 pp = pstat[sg * kMaxRowsPerSeg + (t - (sg * kSeg) / nvec)];
          s1 += pp.x;
          s2 += pp.y;
        }
        if (a.prologue == PRO_RMSNORM) {
          stat[2 * t] = 0.f;
          stat[2 * t + 1] = rsqrtf(s2 * invK + a.norm_eps);
        } else {
          const float mean = s1 * invK;
          stat[2 * t] = mean;
          stat[2 * t + 1] = rsqrtf(fmaxf(s2 * invK - mean * mean, 0.f) + a.norm_eps);
        }
      }
  }
  __syncthreads();                          // stat[] visible; the scratch inside the chunk buffers is dead
}

template <int EPI, int TG, bool W8, int SC>
__global__ __launch_bounds__(kGemvThreads) void gemm_pipe_kernel(const GemvArgs a, const PipeGeom pg) {
  constexpr int KS = W8 ? 64 : 32;   // k per weight step
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int K = a.K, T = a.T;
  const int kc = pg.kc, KP = kc + kXPad, nchunks = pg.nchunks;
  const size_t xs_bytes = pipe_x_bytes(T, kc), part_bytes = pipe_part_bytes(TG);
  uint16_t* xs0 = reinterpret_cast<uint16_t*>(smem);                    // [2][T][kc + pad] bf16
  const int xs_elems = static_cast<int>(xs_bytes >> 1);
  float* part = reinterpret_cast<float*>(smem);                         // aliases the chunk buffers: [16 waves][TG][16][16]
  float* stat = reinterpret_cast<float*>(smem + (2 * xs_bytes > part_bytes ? 2 * xs_bytes : part_bytes));  // [T][2] mean, rstd

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, n = lane & 15;
  const int ksplit = a.ksplit;
  const int tiles_per_round = kGemvWaves >> a.ks_shift;
  const int kpart = wave & (ksplit - 1);
  const int tslot = wave >> a.ks_shift;
  const uint16_t* W = static_cast<const uint16_t*>(a.W);

  const int p_lo = static_cast<int>(blockIdx.x) * a.ppw;
  const int p_hi = min(p_lo + a.ppw, a.n_pairs);
  const int tile_pairs = a.tile_pairs;
  const int n_tiles = (p_lo + a.ppw <= a.n_pairs) ? a.n_tiles_full : (p_hi - p_lo + tile_pairs - 1) / tile_pairs;
  const int rounds = (n_tiles + tiles_per_round - 1) >> (4 - a.ks_shift);
  // (here, not at entry: the arguments are needed from this point on anyway, so the check adds no wait of its own)
  SD_SKIP_IF_INACTIVE(a.skip_k, a.skip_i);

  // address of a lane's A fragment of global step gs: tile start + gs * wstride + lane_off
  int wstride = 32;
  unsigned lane_off = 0;
  auto tile_start = [&](int tile) -> const uint16_t* {
    const int p0 = p_lo + tile * tile_pairs;
    if (a.packed) {
      int np = min(tile_pairs, p_hi - p0);
      if (np < 1) np = 1;
      int jp = n & 7, second = n >> 3;
      if (jp >= np) { jp = 0; second = 0; }
      wstride = np * 64;
      lane_off = static_cast<unsigned>((g * 2 * np + second * np + jp) * 8);
      return W + static_cast<size_t>(p0) * (W8 ? 1 : 2) * K;   // in 2-byte units: one byte per fp8 weight
    }
    int p = p0 + (n & 7);
    int second = n >> 3;
    if ((n & 7) >= tile_pairs || p >= p_hi) { p = min(p0, p_hi - 1); second = 0; }
    int r0, r1;
    pair_rows<EPI>(a, p, r0, r1);
    int r = second ? r1 : r0;
    if (r >= a.N) r = r0;
    wstride = 32;
    lane_off = static_cast<unsigned>(r) * static_cast<unsigned>(K) + static_cast<unsigned>(g * 8);
    return W;
  };

  struct Set { u32x4 w[SC]; };
  Set A, B;
  // weights of chunk c (clamped: chunks past the end re-read the last one, never used)
  auto issue_range = [&](Set& s, const uint16_t* ts, int c, int j0, int j1) {
    const int cc = c < nchunks ? c : nchunks - 1;
    const int gs0 = (cc * ksplit + kpart) * SC;
#pragma unroll
    for (int j = 0; j < SC; ++j)
      if (j >= j0 && j < j1)   // compile-time after unrolling
        s.w[j] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(ts + static_cast<size_t>(gs0 + j) * wstride + lane_off));
  };
  auto issue = [&](Set& s, const uint16_t* ts, int c) { issue_range(s, ts, c, 0, SC); };
  // SC = 8 (ksplit 1 or 2: lm_head, the 8B gate/up): two sets of 8 do not fit the register budget next to the accumulators;
  // ONE set is refilled in halves right after each half has been multiplied (4..8 loads in flight, as two sets of 4)
  constexpr bool kOneSet = (SC >= 8);

  // diagnostic timeline (sd_model_probe_gemv with SPECDEC_GEMV_TIMELINE=1): 0 entry, 1 first loads issued + statistics,
  // 2 chunk 0 staged, 3 K loop done, 4 partials exchanged, 5 epilogue, 6 end
  // (kept in scalar registers and written at the very end: a global store in flight would make the next barrier's
  // release fence wait for vmcnt(0), i.e. for the weight loads too, and the timeline would show that instead)
  unsigned long long ts_[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  auto stamp = [&](int slot) {
    if (a.debug_ts) ts_[slot] = __builtin_amdgcn_s_memrealtime();
  };
  stamp(0);
  if (a.prologue != PRO_NONE)
    pipe_row_stats<TG>(static_cast<const uint16_t*>(a.x), T, K, a.prologue, a.norm_eps, a.xstat_in, a.xstat_n, stat, smem);
  stamp(7);

  // ---- staging of a chunk: thread -> fixed 8-column block kv, token rows t0, t0 + 16 * (64 / kvec) ...
  const uint16_t* xin = static_cast<const uint16_t*>(a.x);
  const int kvec = kc >> 3;                  // 16-byte blocks per row of a chunk: power of two, 16..128
  const int kv = tid & (kvec - 1);
  const int t0 = tid / kvec, tpi = kGemvThreads / kvec;
  // rows a thread stages per chunk: ceil(T / tpi) <= TG for kc = 512 (tpi = 16); kc = 256 needs half of them (the rest
  // are clamped duplicates: no branch around a load, see the header)
  constexpr int XR = TG;
  u32x4 xr[XR], nw4 = {0u, 0u, 0u, 0u}, nb4 = {0u, 0u, 0u, 0u};

  auto xload = [&](int c) {
    const int cc = c < nchunks ? c : nchunks - 1;
    const int col = cc * kc + kv * 8;
#pragma unroll
    for (int u = 0; u < XR; ++u) {
      const int t = t0 + u * tpi;
      xr[u] = *reinterpret_cast<const u32x4*>(xin + static_cast<size_t>(t < T ? t : T - 1) * a.x_stride + col);
    }
    if (a.prologue != PRO_NONE) {
      nw4 = *reinterpret_cast<const u32x4*>(static_cast<const uint16_t*>(a.norm_w) + col);
      if (a.prologue == PRO_LAYERNORM) nb4 = *reinterpret_cast<const u32x4*>(static_cast<const uint16_t*>(a.norm_b) + col);
    }
  };
  auto xstore = [&](int buf) {
    uint16_t* xs = xs0 + buf * xs_elems;
#pragma unroll
    for (int u = 0; u < XR; ++u) {
      const int t = t0 + u * tpi;
      if (t < T) {
        u32x4 q = xr[u];
        if (a.prologue == PRO_RMSNORM) {
          const float rs = stat[2 * t + 1];
#pragma unroll
          for (int j = 0; j < 4; ++j) q[j] = rmsnorm_pair(q[j], rs, nw4[j]);
        } else if (a.prologue == PRO_LAYERNORM) {
          const float mean = stat[2 * t], rs = stat[2 * t + 1];
#pragma unroll
          for (int j = 0; j < 4; ++j) q[j] = layernorm_pair(q[j], mean, rs, nw4[j], nb4[j]);
        }
        *reinterpret_cast<u32x4*>(xs + static_cast<size_t>(t) * KP + kv * 8) = q;
      }
    }
  };

  float best_v[TG];
  int best_i[TG];
#pragma unroll
  for (int q = 0; q < TG; ++q) { best_v[q] = -INFINITY; best_i[q] = 0x7fffffff; }
  float st_sq[TG] = {}, st_sum[TG] = {};     // EPI_RESID: row statistics of the new residual values (xstat_out)

  for (int r = 0; r < rounds; ++r) {
    const int tile = r * tiles_per_round + tslot;
    const bool valid = tile < n_tiles;
    const uint16_t* ts = tile_start(valid ? tile : 0);
    // ---- prime: the first two weight sets and the raw rows of chunk 0, then (round 0) the statistics under them
    // (register budget: 128 per wave at 16 waves per CU. The statistics hold RR rows x 8 loads per lane; next to them
    // there is room for ONE weight set of <= 4 steps and the rows of chunk 0 — the second set follows the statistics)
    if (r != 0) __syncthreads();             // the previous round's epilogue is done with the partials (aliasing the chunk buffers)
    xload(0);                               // first: loads return in order, and these come from L2 while the weights come from HBM
    SD_PIN_ORDER();
    if constexpr (SC < 8) {
      issue(A, ts, 0);
      SD_PIN_ORDER();
    }
    if (r == 0) stamp(1);
    if constexpr (SC >= 8) {
      issue(A, ts, 0);
      SD_PIN_ORDER();
    } else {
      issue(B, ts, 1);
      SD_PIN_ORDER();
    }
    xstore(0);
    __syncthreads();
    if (r == 0) stamp(2);

    // B fragment rows of this lane: token 16 q + n (columns >= T read row T-1; never used). (Defined here, after the
    // prologue: nothing that only the K loop needs should be live across the statistics — 128 registers per wave.)
    int xrow_off[TG];
#pragma unroll
    for (int q = 0; q < TG; ++q) {
      const int t = 16 * q + n;
      xrow_off[q] = (t < T ? t : T - 1) * KP + kpart * SC * KS + g * 8;
    }
    f32x4_t acc[TG];
#pragma unroll
    for (int q = 0; q < TG; ++q) acc[q] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    auto mfma_steps = [&](Set& s, const uint16_t* xs, int j0, int j1) {
#pragma unroll
      for (int j = 0; j < SC; ++j) {
        if (j < j0 || j >= j1) continue;      // compile-time after unrolling
        const int koff = j * KS;
        if constexpr (W8) {
          u32x4 lo, hi;
#pragma unroll
          for (int e = 0; e < 2; ++e) {
            lo[2 * e] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(s.w[j][e], 1.0f, false));
            lo[2 * e + 1] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(s.w[j][e], 1.0f, true));
            hi[2 * e] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(s.w[j][2 + e], 1.0f, false));
            hi[2 * e + 1] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(s.w[j][2 + e], 1.0f, true));
          }
#pragma unroll
          for (int q = 0; q < TG; ++q) {
            const u32x4 xb0 = *reinterpret_cast<const u32x4*>(xs + xrow_off[q] + koff);
            const u32x4 xb1 = *reinterpret_cast<const u32x4*>(xs + xrow_off[q] + koff + 32);
            acc[q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, lo), __builtin_bit_cast(bf16x8_t, xb0), acc[q], 0, 0, 0);
            acc[q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, hi), __builtin_bit_cast(bf16x8_t, xb1), acc[q], 0, 0, 0);
          }
        } else {
#pragma unroll
          for (int q = 0; q < TG; ++q) {
            const u32x4 xb = *reinterpret_cast<const u32x4*>(xs + xrow_off[q] + koff);
            acc[q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, s.w[j]), __builtin_bit_cast(bf16x8_t, xb), acc[q], 0, 0, 0);
          }
        }
      }
    };
    // one chunk: start the rows of the next one, multiply this one, restart this set's weights two chunks ahead (so
    // that a set stays in flight across the barrier: the wait for the rows then counts it), publish the next chunk
    auto chunk = [&](Set& s, int c) {
      if (c >= nchunks) return;               // workgroup-uniform
      const bool more = c + 1 < nchunks;
      xload(c + 1);
      SD_PIN_ORDER();
      const uint16_t* xs = xs0 + (c & 1) * xs_elems;
      if constexpr (kOneSet) {
        if (valid) mfma_steps(s, xs, 0, SC / 2);
        SD_PIN_ORDER();
        issue_range(s, ts, c + 1, 0, SC / 2);
        SD_PIN_ORDER();
        if (valid) mfma_steps(s, xs, SC / 2, SC);
        SD_PIN_ORDER();
        issue_range(s, ts, c + 1, SC / 2, SC);
      } else {
        if (valid) mfma_steps(s, xs, 0, SC);
        SD_PIN_ORDER();
        issue(s, ts, c + 2);
      }
      SD_PIN_ORDER();
      if (more) xstore((c + 1) & 1);
      __syncthreads();                        // chunk c+1 published; everyone is done reading chunk c
    };
    if constexpr (kOneSet) {
      for (int c = 0; c < nchunks; ++c) {
        chunk(A, c);
        SD_PIN_ORDER();
      }
    } else
    for (int c = 0; c < nchunks; c += 2) {
      chunk(A, c);
      SD_PIN_ORDER();
      chunk(B, c + 1);
      SD_PIN_ORDER();
    }

    // K-slice partials through LDS (aliasing the chunk buffers: the last chunk ended with a barrier)
    if (r == 0) stamp(3);
    float* slot = part + static_cast<size_t>(wave) * TG * 256;
#pragma unroll
    for (int q = 0; q < TG; ++q)
#pragma unroll
      for (int e = 0; e < 4; ++e) slot[q * 256 + (4 * g + e) * 16 + n] = acc[q][e];
    __syncthreads();
    if (r == 0) stamp(4);
    for (int it = tid; it < tiles_per_round * 128; it += kGemvThreads) {
      const int tsl = it >> 7, jp = (it >> 4) & 7, tl = it & 15;  // tl == tid & 15 on every trip
      const int etile = r * tiles_per_round + tsl;
      const int p = p_lo + etile * tile_pairs + jp;
      if (etile < n_tiles && jp < tile_pairs && p < p_hi) {
        int r0, r1;
        pair_rows<EPI>(a, p, r0, r1);
        // residual epilogue: the old values of all token groups first (one round trip instead of TG)
        uint32_t oldv[TG];
        if constexpr (EPI == EPI_RESID) {
#pragma unroll
          for (int q = 0; q < TG; ++q) {
            const int t = 16 * q + tl;
            oldv[q] = *reinterpret_cast<const uint32_t*>(static_cast<const uint16_t*>(a.out) + static_cast<size_t>(t < T ? t : T - 1) * a.out_stride + r0);
          }
        }
#pragma unroll
        for (int q = 0; q < TG; ++q) {
          const int t = 16 * q + tl;
          if (t < T) {
            float y0, y1;
            sum_slices(part + static_cast<size_t>(tsl * ksplit) * TG * 256 + q * 256 + jp * 16 + tl, TG * 256, ksplit, y0, y1);
            if constexpr (W8) {
              y0 *= a.w_scale[r0];
              y1 *= (r1 < a.N) ? a.w_scale[r1] : 0.f;
            }
            if constexpr (EPI == EPI_RESID) epilogue<EPI>(a, p, r0, r1, t, y0, y1, best_v[q], best_i[q], true, oldv[q], &st_sq[q], &st_sum[q]);
            else epilogue<EPI>(a, p, r0, r1, t, y0, y1, best_v[q], best_i[q]);
          }
        }
      }
    }
    if (r == 0) stamp(5);
  }
  stamp(6);
  if constexpr (EPI == EPI_RESID)
    if (a.xstat_out) resid_stats_publish<TG>(a, st_sq, st_sum, stat + 2 * 64, tid);   // its own LDS scratch (pipe_smem)
  if (a.debug_ts && tid == 0) {
#pragma unroll
    for (int i = 0; i < 8; ++i) a.debug_ts[static_cast<size_t>(blockIdx.x) * 8 + i] = ts_[i];
  }

  if constexpr (EPI == EPI_ARGMAX) {
    // thread tid holds a running best for tokens 16 q + (tid & 15): fold the 64 candidates per token
    __syncthreads();
    float* sv = part;                                              // [16 TG tokens][64]
    int* si = reinterpret_cast<int*>(part + 16 * TG * 64);
#pragma unroll
    for (int q = 0; q < TG; ++q) {
      sv[(16 * q + (tid & 15)) * 64 + (tid >> 4)] = best_v[q];
      si[(16 * q + (tid & 15)) * 64 + (tid >> 4)] = best_i[q];
    }
    __syncthreads();
    for (int t = wave; t < T; t += kGemvWaves) {
      float bv = sv[t * 64 + lane];
      int bi = si[t * 64 + lane];
      wave_reduce_argmax(bv, bi);
      if (lane == 0) {
        a.part_val[static_cast<size_t>(t) * gridDim.x + blockIdx.x] = bv;
        a.part_idx[static_cast<size_t>(t) * gridDim.x + blockIdx.x] = bi;
      }
    }
  }
}
#undef SD_PIN_ORDER

// ------------------------------------------------------------------------------ host side
// chunk width for (T, K, ksplit): 512 columns (256 when a wave would otherwise own more than 8 steps of a chunk: ksplit = 1,
// the lm_head); must divide K, give a wave 1, 2, 4 or 8 steps per chunk and fit two chunk buffers into the LDS
static int pipe_chunk(int T, int TG, int K, int ksplit, int kw, bool w8, int* sc_out) {
  const int ks = w8 ? 64 : 32;
  if (K % ks != 0 || kw % ks != 0 || kw * ksplit != K || K < 1024) return 0;   // (K >= 1024: rows per statistics segment)
  for (int kc = 512; kc >= 256; kc >>= 1) {
    if (K % kc != 0 || kc % (ks * ksplit) != 0) continue;
    const int sc = kc / (ks * ksplit);
    if (sc != 1 && sc != 2 && sc != 4 && sc != 8) continue;
    const int tpi = kGemvThreads / (kc >> 3);             // token rows staged per pass of the 1024 threads: 16 or 32
    if ((T + tpi - 1) / tpi > TG) continue;
    if (pipe_smem(T, TG, kc) > 160 * 1024) continue;
    *sc_out = sc;
    return kc;
  }
  return 0;
}

template <int EPI, int TG, bool W8, int SC>
static int launch_pipe_one(const GemvArgs& a, const PipeGeom& pg, int grid, size_t smem, hipStream_t st) {
  static unsigned long long attr_set = 0;
  if (int rc = opt_in_dynamic_lds(reinterpret_cast<const void*>(&gemm_pipe_kernel<EPI, TG, W8, SC>), 160 * 1024, attr_set)) return rc;
  hipLaunchKernelGGL((gemm_pipe_kernel<EPI, TG, W8, SC>), dim3(grid), dim3(kGemvThreads), smem, st, a, pg);
  SD_LAUNCH_CHECK();
  return 0;
}

template <int EPI, int TG, bool W8>
static int launch_pipe_sc(const GemvArgs& a, const PipeGeom& pg, int sc, int grid, size_t smem, hipStream_t st) {
  switch (sc) {
    case 1: return launch_pipe_one<EPI, TG, W8, 1>(a, pg, grid, smem, st);
    case 2: return launch_pipe_one<EPI, TG, W8, 2>(a, pg, grid, smem, st);
    case 4: return launch_pipe_one<EPI, TG, W8, 4>(a, pg, grid, smem, st);
    default: return launch_pipe_one<EPI, TG, W8, 8>(a, pg, grid, smem, st);
  }
}

template <int EPI, bool W8>
static int launch_pipe_tg(const GemvArgs& a, const PipeGeom& pg, int sc, int grid, size_t smem, hipStream_t st) {
  switch ((a.T + 15) / 16) {
    case 1: return launch_pipe_sc<EPI, 1, W8>(a, pg, sc, grid, smem, st);
    case 2: return launch_pipe_sc<EPI, 2, W8>(a, pg, sc, grid, smem, st);
    case 3: return launch_pipe_sc<EPI, 3, W8>(a, pg, sc, grid, smem, st);
    default: return launch_pipe_sc<EPI, 4, W8>(a, pg, sc, grid, smem, st);
  }
}

template <int EPI>
static int launch_pipe_epi(const GemvArgs& a, const PipeGeom& pg, int sc, int grid, size_t smem, hipStream_t st) {
  return a.w8 ? launch_pipe_tg<EPI, true>(a, pg, sc, grid, smem, st) : launch_pipe_tg<EPI, false>(a, pg, sc, grid, smem, st);
}

bool gemm_pipe_covers(int T, int n_pairs, int K, bool w8) {
  if (T < 1 || T > 64 || n_pairs < 1) return false;
  const GemvGeom q = gemv_geometry(n_pairs, K);
  int sc = 0;
  return pipe_chunk(T, (T + 15) / 16, K, q.ksplit, q.kw, w8, &sc) != 0;
}

// The launcher of gemm_skinny.hip tries this first (a.ppw .. a.kw and the derived fields already set by it).
int launch_gemm_pipe(const GemvArgs& a, const GemvGeom& q, int epi, hipStream_t st) {
  const int TG = (a.T + 15) / 16;
  int sc = 0;
  PipeGeom pg{};
  pg.kc = pipe_chunk(a.T, TG, a.K, q.ksplit, q.kw, a.w8 != 0, &sc);
  SD_REQUIRE(pg.kc != 0, "gemm_pipe: shape T=%d K=%d (ksplit %d) is not covered", a.T, a.K, q.ksplit);
  SD_REQUIRE(a.prologue == PRO_NONE || a.x_stride == a.K, "gemm_pipe: normalised rows must be contiguous (x_stride %d != K %d)", a.x_stride, a.K);
  pg.nchunks = a.K / pg.kc;
  const size_t smem = pipe_smem(a.T, TG, pg.kc);
  switch (epi) {
    case EPI_QKV_ROPE: return launch_pipe_epi<EPI_QKV_ROPE>(a, pg, sc, q.grid, smem, st);
    case EPI_RESID: return launch_pipe_epi<EPI_RESID>(a, pg, sc, q.grid, smem, st);
    case EPI_SWIGLU: return launch_pipe_epi<EPI_SWIGLU>(a, pg, sc, q.grid, smem, st);
    case EPI_GELU: return launch_pipe_epi<EPI_GELU>(a, pg, sc, q.grid, smem, st);
    case EPI_ARGMAX: return launch_pipe_epi<EPI_ARGMAX>(a, pg, sc, q.grid, smem, st);
    default: SD_REQUIRE(false, "gemm_pipe: unknown epilogue %d", epi);
  }
  return 0;
}

}  // namespace sd
