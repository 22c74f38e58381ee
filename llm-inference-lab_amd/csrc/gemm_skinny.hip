// Weight-streaming skinny GEMM for 10..64 tokens: batched verify (B rows x (K+1) positions)
// and chunked prefill on gfx950.
//
// Same contract, weight layout (row-major or csrc/pack.hip tile streams), work split
// (gemv_geometry) and fused epilogues as gemv.hip, which covers T <= 9. What changes with T:
//   * the token columns no longer fit one MFMA tile: a wave keeps TG = ceil(T/16) accumulators
//     and runs TG MFMAs per 1-KiB weight fragment (the fragment is loaded once);
//   * the activations no longer fit LDS (64 tokens x 8192 x 2 B = 1 MiB): K is walked in
//     contiguous CHUNKS of kc <= 2048 columns. A chunk of all T rows is staged (normalised) into
//     LDS by the whole workgroup, every wave multiplies its sc = kc / (32 ksplit) steps of it,
//     and the next chunk replaces it;
//   * the norm statistics need whole rows before the first chunk: a prologue pass reads x once
//     (rows split over the 16 waves) while the first weight batch is in flight.
// The weight stream is decoupled from the chunks: every wave keeps a double buffer of 8-step
// batches in flight (16 KiB per wave, 256 KiB per CU) and crosses chunk boundaries inside a
// batch. The reference has no such kernel: its verify is K sequential HF forwards per row
// (speculative_scheduler.py:192-199) and its prefill is HF's.

#include <stdlib.h>

#include "gemv_device.h"

namespace sd {

// weight steps per batch NB (two batches in flight): 4 for <= 16 tokens when the chunk allows, else 2. Chunk
// boundaries fall on batch starts (sc % NB == 0), so the staging code exists twice in the kernel, not once per
// step: the kernels are instruction-fetch sensitive (36 -> 33 us for the 3B gate/up at 40 tokens from code size alone).
constexpr int kSkMaxKc = 2048;

struct SkinnyGeom {
  int kc;        // chunk width in columns: power of two, multiple of 32*ksplit, divides K
  int sc_shift;  // log2(steps per wave per chunk)
};

static __host__ __device__ size_t skinny_x_bytes(int T, int kc) { return (static_cast<size_t>(T) * (kc + kXPad) * 2 + 15) & ~static_cast<size_t>(15); }
static __host__ __device__ size_t skinny_part_bytes(int TG) { return sizeof(float) * kGemvWaves * TG * 256; }
static size_t skinny_smem(int T, int TG, int kc) {
  const size_t xs = skinny_x_bytes(T, kc), part = skinny_part_bytes(TG);
  return (xs > part ? xs : part) + sizeof(float) * 2 * kSkinnyMaxT;
}

// chunk width for (T, K, ksplit), or 0 when the shape is not covered
static int skinny_tg(int T) {
  const int tg = (T + 15) / 16;
  return tg <= 4 ? tg : (tg <= 6 ? 6 : 8);
}

static int skinny_chunk(int T, int K, int ksplit, int kw, bool w8 = false) {
  const int ks = w8 ? 64 : 32;   // k covered by one 16-byte weight load
  if (K % ks != 0 || kw % ks != 0 || kw * ksplit != K) return 0;
  const int TG = skinny_tg(T);
  int best = 0;
  for (int kc = ks * ksplit; kc <= kSkMaxKc; kc <<= 1) {
    if (K % kc != 0) break;
    if (skinny_smem(T, TG, kc) > 160 * 1024) break;
    best = kc;
  }
  return best;
}

// W8: fp8 e4m3 weight storage (csrc/pack.hip) — a "step" is then one 16-byte load = 64 k = two MFMAs per
// token group, widened to bf16 in registers as in gemv.hip; the row sum is scaled in the epilogue.
template <int EPI, int TG, bool W8, int NB>
__global__ __launch_bounds__(kGemvThreads) void gemm_skinny_kernel(const GemvArgs a, const SkinnyGeom sg) {
  constexpr int KS = W8 ? 64 : 32;   // k per weight step
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int K = a.K, T = a.T;
  const int kc = sg.kc, KP = kc + kXPad;
  uint16_t* xs = reinterpret_cast<uint16_t*>(smem);                    // [T][kc + pad] bf16, one chunk
  float* part = reinterpret_cast<float*>(smem);                       // aliases xs: [16 waves][TG][16][16]
  const size_t xs_bytes = skinny_x_bytes(T, kc), part_bytes = skinny_part_bytes(TG);
  float* stat = reinterpret_cast<float*>(smem + (xs_bytes > part_bytes ? xs_bytes : part_bytes));  // [T][2] mean, rstd

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, n = lane & 15;
  const int ksplit = a.ksplit;
  const int tiles_per_round = kGemvWaves / ksplit;
  const int kpart = wave & (ksplit - 1);
  const int tslot = wave / ksplit;
  const int steps_w = a.kw / KS;                 // steps of one wave over the whole K
  const int sc_shift = sg.sc_shift, sc = 1 << sc_shift;
  const uint16_t* W = static_cast<const uint16_t*>(a.W);

  SD_SKIP_IF_INACTIVE(a.skip_k, a.skip_i);   // (where the arguments are first needed: no wait of its own)
  const int p_lo = static_cast<int>(blockIdx.x) * a.ppw;
  const int p_hi = min(p_lo + a.ppw, a.n_pairs);
  const int tile_pairs = a.tile_pairs;
  const int n_tiles = (p_hi - p_lo + tile_pairs - 1) / tile_pairs;
  const int rounds = (n_tiles + tiles_per_round - 1) / tiles_per_round;

  // address of a lane's A fragment of global step gs (k = 32 gs): tile start + gs * wstride + lane_off
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
  // wave-local step s (chunk-major) -> global step
  auto gstep = [&](int s) { return ((s >> sc_shift) * ksplit + kpart) * sc + (s & (sc - 1)); };

  constexpr int kSkBatch = NB;
  u32x4 bufA[kSkBatch], bufB[kSkBatch];
  auto issue = [&](u32x4 (&buf)[kSkBatch], const uint16_t* ts, int s0) {
#pragma unroll
    for (int j = 0; j < kSkBatch; ++j) {
      const int s = s0 + j;
      if (s < steps_w) buf[j] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(ts + static_cast<size_t>(gstep(s)) * wstride + lane_off));
    }
  };

  // diagnostic timeline (sd_model_probe_gemv with SPECDEC_GEMV_TIMELINE=1), as in gemv.hip:
  // 0 entry, 1 weights issued + statistics done, 2 first chunk staged, 3 K loop done, 4 partials exchanged, 5 epilogue, 6 end
  auto stamp = [&](int slot) {
    if (a.debug_ts && tid == 0) a.debug_ts[static_cast<size_t>(blockIdx.x) * 8 + slot] = __builtin_amdgcn_s_memrealtime();
  };
  stamp(0);
  // ---- weights of round 0 first, then the norm statistics under their latency
  const uint16_t* ts0 = tile_start(tslot < n_tiles ? tslot : 0);

  const uint16_t* xin = static_cast<const uint16_t*>(a.x);
  if (a.prologue != PRO_NONE) {
    const int nvec = K >> 3;
    const float invK = 1.0f / static_cast<float>(K);
    // a wave takes rows wave, wave + 16, ... two at a time; the loads of both rows (<= 8 per lane each) are all
    // issued before the first sum, so a pair costs one L2 round trip (row by row, load by load it was 6-8 us
    // at 40 tokens: a third of the kernel)
    constexpr int kMaxPer = 8;                       // 16-byte loads per lane per row held in registers (K <= 4096)
    const int nper = (nvec + kWave - 1) / kWave;
    constexpr int RR = 2;   // rows per pass
    for (int tp = wave; tp < T; tp += RR * kGemvWaves) {
      float s1[RR] = {0.f, 0.f}, s2[RR] = {0.f, 0.f};
      if (nper <= kMaxPer) {
        u32x4 q[RR][kMaxPer];
#pragma unroll
        for (int rr = 0; rr < RR; ++rr) {
          const int t = tp + rr * kGemvWaves;
          const u32x4* src = reinterpret_cast<const u32x4*>(xin + static_cast<size_t>(t < T ? t : T - 1) * a.x_stride);
#pragma unroll
          for (int i = 0; i < kMaxPer; ++i) {
            const int v = lane + i * kWave;
            q[rr][i] = (i < nper && v < nvec) ? src[v] : u32x4{0u, 0u, 0u, 0u};
          }
        }
#pragma unroll
        for (int rr = 0; rr < RR; ++rr)
#pragma unroll
          for (int i = 0; i < kMaxPer; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const float lo = __uint_as_float(q[rr][i][j] << 16), hi = __uint_as_float(q[rr][i][j] & 0xffff0000u);
              s1[rr] += lo + hi;
              s2[rr] += lo * lo + hi * hi;
            }
      } else {
        for (int rr = 0; rr < RR; ++rr) {
          const int t = tp + rr * kGemvWaves;
          if (t >= T) break;
          const u32x4* src = reinterpret_cast<const u32x4*>(xin + static_cast<size_t>(t) * a.x_stride);
          for (int v = lane; v < nvec; v += kWave) {
            const u32x4 qq = src[v];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const float lo = __uint_as_float(qq[j] << 16), hi = __uint_as_float(qq[j] & 0xffff0000u);
              s1[rr] += lo + hi;
              s2[rr] += lo * lo + hi * hi;
            }
          }
        }
      }
#pragma unroll
      for (int rr = 0; rr < RR; ++rr) {
      const int t = tp + rr * kGemvWaves;
      const float s1r = wave_reduce_sum(s1[rr]);
      const float s2r = wave_reduce_sum(s2[rr]);
      if (lane == 0 && t < T) {
        if (a.prologue == PRO_RMSNORM) {
          stat[2 * t] = 0.f;
          stat[2 * t + 1] = rsqrtf(s2r * invK + a.norm_eps);
        } else {
          const float mean = s1r * invK;
          stat[2 * t] = mean;
          stat[2 * t + 1] = rsqrtf(fmaxf(s2r * invK - mean * mean, 0.f) + a.norm_eps);
        }
      }
      }
    }
  }
  // (the first chunk boundary's barrier publishes stat[])
  if (tslot < n_tiles) issue(bufA, ts0, 0);   // after the statistics: their row loads need the registers
  stamp(1);

  // ---- staging of chunk c: thread -> fixed 8-column block kv, tokens t0, t0 + tpi, ...
  const int kvec = kc >> 3;                 // power of two <= 256
  const int kv = tid & (kvec - 1);
  const int t0 = tid / kvec, tpi = kGemvThreads / kvec;
  auto stage = [&](int c) {
    const int col = c * kc + kv * 8;
    u32x4 nw4 = {0u, 0u, 0u, 0u}, nb4 = {0u, 0u, 0u, 0u};
    if (a.prologue != PRO_NONE) {
      nw4 = *reinterpret_cast<const u32x4*>(static_cast<const uint16_t*>(a.norm_w) + col);
      if (a.prologue == PRO_LAYERNORM) nb4 = *reinterpret_cast<const u32x4*>(static_cast<const uint16_t*>(a.norm_b) + col);
    }
    // rows t0, t0 + tpi, ... GS at a time: their loads are issued together (one L2 round trip per group;
    // load-normalise-store per row cost ~6 us per chunk at 40 tokens)
    constexpr int GS = (TG <= 2) ? 4 : 2;   // rows per group (registers: the weight double buffer is live here)
    for (int tb = t0; tb < T; tb += GS * tpi) {
      u32x4 qg[GS];
#pragma unroll
      for (int u = 0; u < GS; ++u) {
        const int t = tb + u * tpi;
        qg[u] = *reinterpret_cast<const u32x4*>(xin + static_cast<size_t>(t < T ? t : T - 1) * a.x_stride + col);
      }
#pragma unroll
      for (int u = 0; u < GS; ++u) {
      const int t = tb + u * tpi;
      if (t >= T) break;
      u32x4 q = qg[u];
      if (a.prologue == PRO_RMSNORM) {
        // HF LlamaRMSNorm: weight * (x * rsqrt(var + eps)).to(bf16)
        const float rs = stat[2 * t + 1];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float x0 = bf16_bits_to_float(float_to_bf16_bits(__uint_as_float(q[j] << 16) * rs));
          const float x1 = bf16_bits_to_float(float_to_bf16_bits(__uint_as_float(q[j] & 0xffff0000u) * rs));
          q[j] = static_cast<uint32_t>(float_to_bf16_bits(x0 * __uint_as_float(nw4[j] << 16))) |
                 (static_cast<uint32_t>(float_to_bf16_bits(x1 * __uint_as_float(nw4[j] & 0xffff0000u))) << 16);
        }
      } else if (a.prologue == PRO_LAYERNORM) {
        const float mean = stat[2 * t], rs = stat[2 * t + 1];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float y0 = (__uint_as_float(q[j] << 16) - mean) * rs * __uint_as_float(nw4[j] << 16) + __uint_as_float(nb4[j] << 16);
          const float y1 = (__uint_as_float(q[j] & 0xffff0000u) - mean) * rs * __uint_as_float(nw4[j] & 0xffff0000u) +
                           __uint_as_float(nb4[j] & 0xffff0000u);
          q[j] = static_cast<uint32_t>(float_to_bf16_bits(y0)) | (static_cast<uint32_t>(float_to_bf16_bits(y1)) << 16);
        }
      }
      *reinterpret_cast<u32x4*>(xs + static_cast<size_t>(t) * KP + kv * 8) = q;
      }
    }
  };

  float best_v[TG];
  int best_i[TG];
#pragma unroll
  for (int q = 0; q < TG; ++q) { best_v[q] = -INFINITY; best_i[q] = 0x7fffffff; }

  // B fragment rows of this lane: token 16 grp + n (columns >= T read row T-1; never used)
  int xrow_off[TG];
#pragma unroll
  for (int q = 0; q < TG; ++q) {
    const int t = 16 * q + n;
    xrow_off[q] = (t < T ? t : T - 1) * KP + kpart * sc * KS + g * 8;
  }

  for (int r = 0; r < rounds; ++r) {
    const int tile = r * tiles_per_round + tslot;
    const bool valid = tile < n_tiles;
    const uint16_t* ts = (r == 0) ? ts0 : tile_start(valid ? tile : 0);
    if (r != 0 && valid) issue(bufA, ts, 0);
    f32x4_t acc[TG];
#pragma unroll
    for (int q = 0; q < TG; ++q) acc[q] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    auto consume = [&](u32x4 (&buf)[kSkBatch], int s0) {
      if (s0 < steps_w && (s0 & (sc - 1)) == 0) {  // chunk boundary (always a batch start): replace the staged chunk
        __syncthreads();                           // every wave is done with the previous chunk (or partials)
        stage(s0 >> sc_shift);
        __syncthreads();
        if (r == 0 && s0 == 0) stamp(2);
      }
#pragma unroll
      for (int j = 0; j < kSkBatch; ++j) {
        const int s = s0 + j;
        if (s < steps_w) {                       // workgroup-uniform
          if (valid) {
            const int koff = (s & (sc - 1)) * KS;
            if constexpr (W8) {
              u32x4 lo, hi;
#pragma unroll
              for (int e = 0; e < 2; ++e) {
                lo[2 * e] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(buf[j][e], 1.0f, false));
                lo[2 * e + 1] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(buf[j][e], 1.0f, true));
                hi[2 * e] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(buf[j][2 + e], 1.0f, false));
                hi[2 * e + 1] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(buf[j][2 + e], 1.0f, true));
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
                acc[q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, buf[j]),
                                                                 __builtin_bit_cast(bf16x8_t, xb), acc[q], 0, 0, 0);
              }
            }
          }
        }
      }
    };
    for (int s0 = 0; s0 < steps_w; s0 += 2 * kSkBatch) {
      if (valid) issue(bufB, ts, s0 + kSkBatch);
      consume(bufA, s0);
      if (valid) issue(bufA, ts, s0 + 2 * kSkBatch);
      consume(bufB, s0 + kSkBatch);
    }

    // K-slice partials through LDS (aliasing the chunk buffer)
    if (r == 0) stamp(3);
    __syncthreads();
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
            if constexpr (EPI == EPI_RESID) epilogue<EPI>(a, p, r0, r1, t, y0, y1, best_v[q], best_i[q], true, oldv[q]);
            else epilogue<EPI>(a, p, r0, r1, t, y0, y1, best_v[q], best_i[q]);
          }
        }
      }
    }
    // the next round's first chunk boundary (or the fold below) starts with a barrier
    if (r == 0) stamp(5);
  }
  stamp(6);

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

// ------------------------------------------------------------------------------
// Direct-operand variant for matrices WITHOUT a fused normalisation whose workgroup share is a single
// round of tiles (out-projection and down-projection: N = d_model, <= 8 pairs per workgroup,
// ksplit = 16): there a staged x chunk is read by exactly one wave per K slice, so LDS staging buys
// no reuse and its two workgroup barriers per chunk are pure cost (26 us for the 3B down-projection
// at 40 tokens). Here every wave loads its B fragments (16 tokens x 32 k, 16 bytes per lane, L2
// resident) straight into registers next to the A fragments, batch by batch, and runs free of the
// other waves until the K-slice reduction — the structure of gemv.hip with TG token groups.
// ------------------------------------------------------------------------------
template <int EPI, int TG>
__global__ __launch_bounds__(kGemvThreads) void gemm_direct_kernel(const GemvArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int NB = (TG == 1) ? 10 : (TG == 2) ? 6 : (TG == 3) ? 5 : 4;   // steps per batch: NB * (1 + TG) 16-byte loads in flight per lane
  const int K = a.K, T = a.T;
  float* part = reinterpret_cast<float*>(smem);  // [16 waves][TG][16][16]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, n = lane & 15;
  const int ksplit = a.ksplit;
  const int tiles_per_round = kGemvWaves / ksplit;
  const int kpart = wave & (ksplit - 1);
  const int tslot = wave / ksplit;
  const int k_begin = kpart * a.kw;
  const int steps = a.kw >> 5;
  const uint16_t* W = static_cast<const uint16_t*>(a.W);
  const int p_lo = static_cast<int>(blockIdx.x) * a.ppw;
  const int p_hi = min(p_lo + a.ppw, a.n_pairs);
  const int tile_pairs = a.tile_pairs;
  const int n_tiles = (p_hi - p_lo + tile_pairs - 1) / tile_pairs;   // <= tiles_per_round (launcher)
  SD_SKIP_IF_INACTIVE(a.skip_k, a.skip_i);   // (where the arguments are first needed: no wait of its own)

  // A fragment address: base + step * wstride + lane_off (as gemv.hip)
  int wstride = 32;
  unsigned lane_off = 0;
  const uint16_t* wbase;
  {
    const int tile = tslot < n_tiles ? tslot : 0;
    const int p0 = p_lo + tile * tile_pairs;
    if (a.packed) {
      int np = min(tile_pairs, p_hi - p0);
      if (np < 1) np = 1;
      int jp = n & 7, second = n >> 3;
      if (jp >= np) { jp = 0; second = 0; }
      wstride = np * 64;
      lane_off = static_cast<unsigned>((g * 2 * np + second * np + jp) * 8);
      wbase = W + static_cast<size_t>(p0) * 2 * K + static_cast<size_t>(k_begin >> 5) * wstride;
    } else {
      int p = p0 + (n & 7);
      int second = n >> 3;
      if ((n & 7) >= tile_pairs || p >= p_hi) { p = min(p0, p_hi - 1); second = 0; }
      int r0, r1;
      pair_rows<EPI>(a, p, r0, r1);
      int r = second ? r1 : r0;
      if (r >= a.N) r = r0;
      lane_off = static_cast<unsigned>(r) * static_cast<unsigned>(K) + static_cast<unsigned>(g * 8);
      wbase = W + k_begin;
    }
  }
  // B fragment rows of this lane: token 16 q + n (columns >= T read row T-1; never used)
  const uint16_t* xin = static_cast<const uint16_t*>(a.x);
  const uint16_t* xrow[TG];
#pragma unroll
  for (int q = 0; q < TG; ++q) {
    const int t = 16 * q + n;
    xrow[q] = xin + static_cast<size_t>(t < T ? t : T - 1) * a.x_stride + k_begin + g * 8;
  }

  f32x4_t acc[TG];
#pragma unroll
  for (int q = 0; q < TG; ++q) acc[q] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  if (tslot < n_tiles) {
    for (int s0 = 0; s0 < steps; s0 += NB) {
      u32x4 wb[NB], xb[NB][TG];
#pragma unroll
      for (int j = 0; j < NB; ++j) {
        const int s = s0 + j;
        if (s < steps) {  // wave-uniform
#pragma unroll
          for (int q = 0; q < TG; ++q) xb[j][q] = *reinterpret_cast<const u32x4*>(xrow[q] + s * 32);
          wb[j] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(wbase + static_cast<size_t>(s) * wstride + lane_off));
        }
      }
#pragma unroll
      for (int j = 0; j < NB; ++j) {
        if (s0 + j < steps) {
#pragma unroll
          for (int q = 0; q < TG; ++q)
            acc[q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, wb[j]), __builtin_bit_cast(bf16x8_t, xb[j][q]),
                                                             acc[q], 0, 0, 0);
        }
      }
    }
  }

  float* slot = part + static_cast<size_t>(wave) * TG * 256;
#pragma unroll
  for (int q = 0; q < TG; ++q)
#pragma unroll
    for (int e = 0; e < 4; ++e) slot[q * 256 + (4 * g + e) * 16 + n] = acc[q][e];
  __syncthreads();
  float best_v[TG];
  int best_i[TG];
#pragma unroll
  for (int q = 0; q < TG; ++q) { best_v[q] = -INFINITY; best_i[q] = 0x7fffffff; }
  float st_sq[TG] = {}, st_sum[TG] = {};   // row statistics of the new residual values (xstat_out)
  for (int it = tid; it < tiles_per_round * 128; it += kGemvThreads) {
    const int tsl = it >> 7, jp = (it >> 4) & 7, tl = it & 15;
    const int p = p_lo + tsl * tile_pairs + jp;
    if (tsl < n_tiles && jp < tile_pairs && p < p_hi) {
      int r0, r1;
      pair_rows<EPI>(a, p, r0, r1);
#pragma unroll
      for (int q = 0; q < TG; ++q) {
        const int t = 16 * q + tl;
        if (t < T) {
          float y0, y1;
          sum_slices(part + static_cast<size_t>(tsl * ksplit) * TG * 256 + q * 256 + jp * 16 + tl, TG * 256, ksplit, y0, y1);
          if constexpr (EPI == EPI_RESID) epilogue<EPI>(a, p, r0, r1, t, y0, y1, best_v[q], best_i[q], false, 0u, &st_sq[q], &st_sum[q]);
          else epilogue<EPI>(a, p, r0, r1, t, y0, y1, best_v[q], best_i[q]);
        }
      }
    }
  }
  if constexpr (EPI == EPI_RESID)
    if (a.xstat_out) resid_stats_publish<TG>(a, st_sq, st_sum, part + static_cast<size_t>(kGemvWaves) * TG * 256, tid);   // scratch after the partials
}

// ------------------------------------------------------------------------------
// Wave-private operand staging for the same shapes (un-normalised, single round of tiles: out / down projections)
// at 17..64 tokens. With N = d_model rows over 256 workgroups a CU owns ~12 weight rows but needs ALL of x: per CU the
// activations (T x K x 2 B) outweigh the weights 2-3x, and each x element is consumed by exactly ONE wave (the one
// that owns its K slice). So there is nothing to share through a workgroup-wide staged chunk — the two workgroup
// barriers and the L2 round trip per chunk of the staged kernel were pure cost (13.5 us for the 3B out-projection at
// 40 tokens = 0.17 of the HBM roofline). Here every wave runs alone until the final K-slice reduction:
//   * its K slice is walked in sub-slices of W = 64 (32 above 48 tokens) columns; the x rows of a sub-slice are
//     loaded with full 16-byte lanes (8 or 16 token rows x 128 / 64 contiguous bytes per wave-instruction, not the
//     16 x 64-byte gather of the direct variant), written to the wave's OWN LDS region (no workgroup barrier: DS
//     operations of a wave execute in order) and read back as MFMA B fragments;
//   * two register sets (A / B) alternate, each holding the weights and the x rows of one sub-slice: while one is
//     consumed the other is in flight (16 waves x ~7 KiB per CU);
//   * one barrier in the whole kernel, before the K-slice partials are summed.
// ------------------------------------------------------------------------------
template <int TG>
struct SliceCfg {
  static constexpr int W = (TG <= 3) ? 64 : 32;          // columns per sub-slice
  static constexpr int RPI = 1024 / (W * 2);             // token rows per wave-wide 16-byte load instruction
  static constexpr int NX = 16 * TG / RPI;               // x load instructions per sub-slice
  static constexpr int NWS = W / 32;                     // weight steps per sub-slice
  static constexpr int XP = W + kXPad;                   // padded row length in LDS (elements)
  static constexpr int REGION = 16 * TG * XP * 2;        // bytes of a wave's private x buffer
};

template <int EPI, int TG>
__global__ __launch_bounds__(kGemvThreads) void gemm_slice_kernel(const GemvArgs a) {
  using C = SliceCfg<TG>;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int K = a.K, T = a.T;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, n = lane & 15;
  uint16_t* xw = reinterpret_cast<uint16_t*>(smem + static_cast<size_t>(wave) * C::REGION);   // this wave's [16 TG][W + pad]
  float* part = reinterpret_cast<float*>(smem + static_cast<size_t>(kGemvWaves) * C::REGION);  // [16 waves][TG][16][16]
  const int ksplit = a.ksplit;
  const int tiles_per_round = kGemvWaves / ksplit;
  const int kpart = wave & (ksplit - 1);
  const int tslot = wave / ksplit;
  const int k_begin = kpart * a.kw;
  const int nsub = a.kw / C::W;                          // sub-slices of this wave (launcher: kw % W == 0)
  const uint16_t* Wt = static_cast<const uint16_t*>(a.W);
  const int p_lo = static_cast<int>(blockIdx.x) * a.ppw;
  const int p_hi = min(p_lo + a.ppw, a.n_pairs);
  const int tile_pairs = a.tile_pairs;
  const int n_tiles = (p_hi - p_lo + tile_pairs - 1) / tile_pairs;   // <= tiles_per_round (launcher)
  const bool valid = tslot < n_tiles;
  SD_SKIP_IF_INACTIVE(a.skip_k, a.skip_i);   // (where the arguments are first needed: no wait of its own)

  // A fragment address: base + step * wstride + lane_off (as gemv.hip); waves without a tile read tile 0 (L2 hits)
  int wstride = 32;
  unsigned lane_off = 0;
  const uint16_t* wbase;
  {
    const int tile = valid ? tslot : 0;
    const int p0 = p_lo + tile * tile_pairs;
    if (a.packed) {
      int np = min(tile_pairs, p_hi - p0);
      if (np < 1) np = 1;
      int jp = n & 7, second = n >> 3;
      if (jp >= np) { jp = 0; second = 0; }
      wstride = np * 64;
      lane_off = static_cast<unsigned>((g * 2 * np + second * np + jp) * 8);
      wbase = Wt + static_cast<size_t>(p0) * 2 * K + static_cast<size_t>(k_begin >> 5) * wstride;
    } else {
      int p = p0 + (n & 7);
      int second = n >> 3;
      if ((n & 7) >= tile_pairs || p >= p_hi) { p = min(p0, p_hi - 1); second = 0; }
      int r0, r1;
      pair_rows<EPI>(a, p, r0, r1);
      int r = second ? r1 : r0;
      if (r >= a.N) r = r0;
      lane_off = static_cast<unsigned>(r) * static_cast<unsigned>(K) + static_cast<unsigned>(g * 8);
      wbase = Wt + k_begin;
    }
  }
  // x rows of a sub-slice: load instruction i covers token rows i*RPI .. i*RPI + RPI-1; lane -> (row, 16-byte chunk)
  constexpr int CPR = C::W / 8;                           // chunks per row
  const int xrow_in = lane / CPR, xchunk = lane % CPR;
  const uint16_t* xin = static_cast<const uint16_t*>(a.x) + k_begin + xchunk * 8;
  unsigned xoff[C::NX];                                   // element offset of this lane's row for load instruction i
#pragma unroll
  for (int i = 0; i < C::NX; ++i) {
    const int t = i * C::RPI + xrow_in;
    xoff[i] = static_cast<unsigned>(t < T ? t : T - 1) * static_cast<unsigned>(a.x_stride);
  }
  const int lds_w = xrow_in * C::XP + xchunk * 8;         // + i * RPI * XP
  int lds_r[TG];                                          // B fragment of token 16 q + n, k-group g (+ step * 32)
#pragma unroll
  for (int q = 0; q < TG; ++q) lds_r[q] = (16 * q + n) * C::XP + g * 8;

  struct Set { u32x4 w[C::NWS]; u32x4 x[C::NX]; };
  Set A, B;
  // Unconditional (a conditional load would cost the counted waits), but the loads of sub-slices past the end — up to two
  // sets at the tail of the two-set pipeline, 40 % of all loads when a wave owns three sub-slices — are pointed at ONE
  // line (offset 0 of the wave's slice, by masking the offsets: no branch): the kernel is bound by what a CU can pull
  // through its L1 (~55-60 GB/s measured), and a re-read of a whole sub-slice cost as much as a real one.
  auto issue = [&](Set& s, int c) {
    const unsigned live = c < nsub ? 0xffffffffu : 0u;   // wave-uniform
    const unsigned xcol = static_cast<unsigned>(c * C::W) & live;
    const unsigned wstep = static_cast<unsigned>(c * C::NWS) & live;
#pragma unroll
    for (int i = 0; i < C::NX; ++i) s.x[i] = *reinterpret_cast<const u32x4*>(xin + ((xoff[i] & live) + xcol));
#pragma unroll
    for (int j = 0; j < C::NWS; ++j)
      s.w[j] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(wbase + (static_cast<size_t>(wstep + j) * wstride + (lane_off & live))));
  };
  f32x4_t acc[TG];
#pragma unroll
  for (int q = 0; q < TG; ++q) acc[q] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  auto consume = [&](Set& s, int c) {
    if (c >= nsub) return;                                 // wave-uniform
#pragma unroll
    for (int i = 0; i < C::NX; ++i) *reinterpret_cast<u32x4*>(xw + lds_w + i * C::RPI * C::XP) = s.x[i];
#pragma unroll
    for (int j = 0; j < C::NWS; ++j)
#pragma unroll
      for (int q = 0; q < TG; ++q) {
        const u32x4 xb = *reinterpret_cast<const u32x4*>(xw + lds_r[q] + j * 32);
        acc[q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, s.w[j]), __builtin_bit_cast(bf16x8_t, xb), acc[q], 0, 0, 0);
      }
  };
  // The loads of the OTHER set must be in flight while one set is consumed. Left alone, the machine scheduler sinks
  // every load to its first use (register pressure heuristic) and the loop degenerates to issue -> s_waitcnt vmcnt(0)
  // -> consume with ONE set live; a scheduling barrier after each issue pins the order, and the waits become counted
  // (vmcnt = loads of the younger set).
#define SD_PIN_ORDER()                    \
  do {                                    \
    asm volatile("" ::: "memory");        \
    __builtin_amdgcn_sched_barrier(0);    \
  } while (0)
  // diagnostic timeline (sd_model_probe_gemv with SPECDEC_GEMV_TIMELINE=1), stamps of thread 0: 0 entry, 1 first set issued,
  // 3 K loop done, 4 partials exchanged, 5 epilogue, 6 end
  auto stamp = [&](int slot) {
    if (a.debug_ts && tid == 0) a.debug_ts[static_cast<size_t>(blockIdx.x) * 8 + slot] = __builtin_amdgcn_s_memrealtime();
  };
  stamp(0);
  issue(A, 0);
  SD_PIN_ORDER();
  stamp(1);
  for (int c = 0; c < nsub; c += 2) {
    issue(B, c + 1);
    SD_PIN_ORDER();
    consume(A, c);
    SD_PIN_ORDER();
    issue(A, c + 2);
    SD_PIN_ORDER();
    consume(B, c + 1);
    SD_PIN_ORDER();
  }
#undef SD_PIN_ORDER

  stamp(3);
  float* slot = part + static_cast<size_t>(wave) * TG * 256;
#pragma unroll
  for (int q = 0; q < TG; ++q)
#pragma unroll
    for (int e = 0; e < 4; ++e) slot[q * 256 + (4 * g + e) * 16 + n] = valid ? acc[q][e] : 0.f;
  __syncthreads();
  stamp(4);
  float best_v[TG];
  int best_i[TG];
#pragma unroll
  for (int q = 0; q < TG; ++q) { best_v[q] = -INFINITY; best_i[q] = 0x7fffffff; }
  float st_sq[TG] = {}, st_sum[TG] = {};   // row statistics of the new residual values (xstat_out)
  for (int it = tid; it < tiles_per_round * 128; it += kGemvThreads) {
    const int tsl = it >> 7, jp = (it >> 4) & 7, tl = it & 15;
    const int p = p_lo + tsl * tile_pairs + jp;
    if (tsl < n_tiles && jp < tile_pairs && p < p_hi) {
      int r0, r1;
      pair_rows<EPI>(a, p, r0, r1);
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
          if constexpr (EPI == EPI_RESID) epilogue<EPI>(a, p, r0, r1, t, y0, y1, best_v[q], best_i[q], true, oldv[q], &st_sq[q], &st_sum[q]);
          else epilogue<EPI>(a, p, r0, r1, t, y0, y1, best_v[q], best_i[q]);
        }
      }
    }
  }
  stamp(5);
  // the wave-private x regions are dead (every wave passed the barrier above): scratch for the row statistics
  if constexpr (EPI == EPI_RESID)
    if (a.xstat_out) resid_stats_publish<TG>(a, st_sq, st_sum, reinterpret_cast<float*>(smem), tid);
  stamp(6);
}

template <int EPI, int TG>
static int launch_slice_one(const GemvArgs& a, int grid, hipStream_t st) {
  using C = SliceCfg<TG>;
  const size_t smem = static_cast<size_t>(kGemvWaves) * C::REGION + skinny_part_bytes(TG);
  static_assert(static_cast<size_t>(kGemvWaves) * C::REGION + sizeof(float) * kGemvWaves * TG * 256 <= 160 * 1024, "LDS");
  static unsigned long long attr_set = 0;
  if (int rc = opt_in_dynamic_lds(reinterpret_cast<const void*>(&gemm_slice_kernel<EPI, TG>), 160 * 1024, attr_set)) return rc;
  hipLaunchKernelGGL((gemm_slice_kernel<EPI, TG>), dim3(grid), dim3(kGemvThreads), smem, st, a);
  SD_LAUNCH_CHECK();
  return 0;
}

// shapes the wave-private kernel takes: whole sub-slices per wave, equal K slices
static bool slice_covers(const GemvArgs& a, const GemvGeom& q, int TG) {
  const int W = TG <= 3 ? 64 : 32;
  return TG >= 1 && TG <= 4 && q.kw % W == 0 && q.kw * q.ksplit == a.K && q.n_tiles <= kGemvWaves / q.ksplit;
}

template <int EPI>
static int launch_slice(const GemvArgs& a, int grid, hipStream_t st) {
  switch ((a.T + 15) / 16) {
    case 1: return launch_slice_one<EPI, 1>(a, grid, st);
    case 2: return launch_slice_one<EPI, 2>(a, grid, st);
    case 3: return launch_slice_one<EPI, 3>(a, grid, st);
    default: return launch_slice_one<EPI, 4>(a, grid, st);
  }
}

template <int EPI>
static int launch_direct(const GemvArgs& a, int grid, hipStream_t st) {
  const int TG = (a.T + 15) / 16;
  const size_t smem = skinny_part_bytes(TG) + sizeof(float) * kGemvWaves * TG * 32;   // partials + row-statistics scratch, <= 72 KiB
  static unsigned long long attr_set = 0;   // (only the 4-tile instance passes 64 KiB)
  if (int rc = opt_in_dynamic_lds(reinterpret_cast<const void*>(&gemm_direct_kernel<EPI, 4>), 80 * 1024, attr_set)) return rc;
  switch (TG) {
    case 1: hipLaunchKernelGGL((gemm_direct_kernel<EPI, 1>), dim3(grid), dim3(kGemvThreads), smem, st, a); break;
    case 2: hipLaunchKernelGGL((gemm_direct_kernel<EPI, 2>), dim3(grid), dim3(kGemvThreads), smem, st, a); break;
    case 3: hipLaunchKernelGGL((gemm_direct_kernel<EPI, 3>), dim3(grid), dim3(kGemvThreads), smem, st, a); break;
    default: hipLaunchKernelGGL((gemm_direct_kernel<EPI, 4>), dim3(grid), dim3(kGemvThreads), smem, st, a); break;
  }
  SD_LAUNCH_CHECK();
  return 0;
}

template <int EPI, int TG, bool W8, int NB>
static int launch_skinny_one(const GemvArgs& a, const SkinnyGeom& sg, int grid, size_t smem, hipStream_t st) {
  static unsigned long long attr_set = 0;
  if (int rc = opt_in_dynamic_lds(reinterpret_cast<const void*>(&gemm_skinny_kernel<EPI, TG, W8, NB>), 160 * 1024, attr_set)) return rc;
  hipLaunchKernelGGL((gemm_skinny_kernel<EPI, TG, W8, NB>), dim3(grid), dim3(kGemvThreads), smem, st, a, sg);
  SD_LAUNCH_CHECK();
  return 0;
}

template <int EPI, bool W8>
static int launch_skinny_w(const GemvArgs& a, const SkinnyGeom& sg, int grid, size_t smem, hipStream_t st) {
  const int sc = 1 << sg.sc_shift;   // steps per wave per chunk: batches must not straddle a chunk boundary
  const int tg = (a.T + 15) / 16;   // token groups: 1..4, then 6 and 8 (5 and 7 round up)
  if (sc == 1) {
    switch (tg) {
      case 1: return launch_skinny_one<EPI, 1, W8, 1>(a, sg, grid, smem, st);
      case 2: return launch_skinny_one<EPI, 2, W8, 1>(a, sg, grid, smem, st);
      case 3: return launch_skinny_one<EPI, 3, W8, 1>(a, sg, grid, smem, st);
      case 4: return launch_skinny_one<EPI, 4, W8, 1>(a, sg, grid, smem, st);
      case 5: case 6: return launch_skinny_one<EPI, 6, W8, 1>(a, sg, grid, smem, st);
      default: return launch_skinny_one<EPI, 8, W8, 1>(a, sg, grid, smem, st);
    }
  }
  switch (tg) {
    case 1:
      if (sc % 4 == 0) return launch_skinny_one<EPI, 1, W8, 4>(a, sg, grid, smem, st);
      return launch_skinny_one<EPI, 1, W8, 2>(a, sg, grid, smem, st);
    case 2: return launch_skinny_one<EPI, 2, W8, 2>(a, sg, grid, smem, st);
    case 3: return launch_skinny_one<EPI, 3, W8, 2>(a, sg, grid, smem, st);
    case 4: return launch_skinny_one<EPI, 4, W8, 2>(a, sg, grid, smem, st);
    case 5: case 6: return launch_skinny_one<EPI, 6, W8, 2>(a, sg, grid, smem, st);
    default: return launch_skinny_one<EPI, 8, W8, 2>(a, sg, grid, smem, st);
  }
}

template <int EPI>
static int launch_skinny_epi(const GemvArgs& a, const SkinnyGeom& sg, int grid, size_t smem, hipStream_t st) {
  return a.w8 ? launch_skinny_w<EPI, true>(a, sg, grid, smem, st) : launch_skinny_w<EPI, false>(a, sg, grid, smem, st);
}

// Which body a multi-token launch takes — ONE function, used by the launcher and by the forward's question "does this
// EPI_RESID launch publish the row statistics" (a second copy of these conditions would let the two drift apart: the
// consumer would then fold stale partial sums with no error). The knobs are read once.
enum SkinnyBody { BODY_DIRECT, BODY_SLICE, BODY_PIPE, BODY_CHUNKED };
static SkinnyBody choose_skinny_body(const GemvArgs& a, const GemvGeom& q, int epi) {
  static const bool no_direct = getenv(debug_env::kNoDirect) != nullptr;
  static const bool no_pipe = getenv(debug_env::kNoPipe) != nullptr;
  constexpr int slice_min_t = 17;
  const int TG = (a.T + 15) / 16;
  const bool plain_resid = !a.w8 && a.prologue == PRO_NONE && epi == EPI_RESID;
  // un-normalised, single-round shapes (out / down projections): operands straight to registers (measured on the 3B shapes:
  // ahead of the staged kernel up to 16 tokens, behind it from 24 — its B loads touch 16 rows x 64 bytes per instruction)
  if (TG == 1 && plain_resid && q.n_tiles <= kGemvWaves / q.ksplit && !no_direct) return BODY_DIRECT;
  // ... and from 17 tokens the wave-private staging
  if (a.T >= slice_min_t && a.T <= 48 && plain_resid && slice_covers(a, q, TG)) return BODY_SLICE;
  // the statically scheduled chunk pipeline (gemm_pipe.hip) for everything else up to 64 tokens
  if (!no_pipe && a.T <= 64 && gemm_pipe_covers(a.T, a.n_pairs, a.K, a.w8 != 0)) return BODY_PIPE;
  return BODY_CHUNKED;
}

// direct, slice and pipe bodies publish the row statistics of an EPI_RESID launch (xstat_out), the chunked fallback does not
bool gemm_resid_publishes_stats(const GemvArgs& a) {
  if (a.T <= kGemvMaxT || a.T > 64 || a.x_row) return false;
  return choose_skinny_body(a, gemv_geometry(a.n_pairs, a.K), EPI_RESID) != BODY_CHUNKED;
}

bool gemm_skinny_covers(int T, int n_pairs, int K, bool w8) {
  if (T < 1 || T > kSkinnyMaxT || n_pairs < 1) return false;
  const GemvGeom q = gemv_geometry(n_pairs, K);
  return skinny_chunk(T, K, q.ksplit, q.kw, w8) != 0;
}

int launch_gemm_skinny(const GemvArgs& a_in, int epi, hipStream_t st) {
  GemvArgs a = a_in;
  SD_REQUIRE(a.T >= 1 && a.T <= kSkinnyMaxT, "gemm_skinny: T=%d out of range 1..%d", a.T, kSkinnyMaxT);
  SD_REQUIRE(a.K % 8 == 0 && a.x_stride % 8 == 0, "gemm_skinny: K=%d / x_stride=%d must be multiples of 8", a.K, a.x_stride);
  SD_REQUIRE(a.n_pairs > 0, "gemm_skinny: no rows");
  SD_REQUIRE(!a.x_row, "gemm_skinny: gathered activation rows are a gemv.hip (<= 9 tokens) feature");
  SD_REQUIRE(!a.w8 || (a.packed && a.w_scale), "gemm_skinny: fp8 weights need the packed layout and row scales");
  const GemvGeom q = gemv_geometry(a.n_pairs, a.K);
  a.ppw = q.ppw;
  a.tile_pairs = q.tile_pairs;
  a.ksplit = q.ksplit;
  a.kw = q.kw;
  a.n_tiles_full = (q.ppw + q.tile_pairs - 1) / q.tile_pairs;
  gemv_derive(a);   // the shared epilogues index with the derived shifts
  SkinnyGeom sg{};
  sg.kc = skinny_chunk(a.T, a.K, q.ksplit, q.kw, a.w8 != 0);
  SD_REQUIRE(sg.kc != 0, "gemm_skinny: shape T=%d K=%d (ksplit %d) is not covered", a.T, a.K, q.ksplit);
  const int sc = sg.kc / ((a.w8 ? 64 : 32) * q.ksplit);
  sg.sc_shift = 0;
  while ((1 << sg.sc_shift) < sc) ++sg.sc_shift;
  const int TG = skinny_tg(a.T);
  switch (choose_skinny_body(a, q, epi)) {
    case BODY_DIRECT: return launch_direct<EPI_RESID>(a, q.grid, st);
    case BODY_SLICE: return launch_slice<EPI_RESID>(a, q.grid, st);
    case BODY_PIPE: return launch_gemm_pipe(a, q, epi, st);
    default: break;
  }
  const size_t smem = skinny_smem(a.T, TG, sg.kc);
  switch (epi) {
    case EPI_QKV_ROPE: return launch_skinny_epi<EPI_QKV_ROPE>(a, sg, q.grid, smem, st);
    case EPI_RESID: return launch_skinny_epi<EPI_RESID>(a, sg, q.grid, smem, st);
    case EPI_SWIGLU: return launch_skinny_epi<EPI_SWIGLU>(a, sg, q.grid, smem, st);
    case EPI_GELU: return launch_skinny_epi<EPI_GELU>(a, sg, q.grid, smem, st);
    case EPI_ARGMAX: return launch_skinny_epi<EPI_ARGMAX>(a, sg, q.grid, smem, st);
    default: SD_REQUIRE(false, "gemm_skinny: unknown epilogue %d", epi);
  }
  return 0;
}

}  // namespace sd
