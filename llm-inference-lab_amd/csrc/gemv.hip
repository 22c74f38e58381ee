// Weight-streaming skinny GEMM ("GEMV for T <= 9 tokens") for gfx950 decode/verify.
//
// y[t][n] = sum_k W[n][k] * x[t][k],  W bf16 [N][K] row-major (HF Linear layout),
// T = B*M tokens of one draft / verify forward (1..9). The weights are read ONCE
// from HBM per forward — this kernel IS the decode HBM roofline: its algorithmic
// bytes are N*K*2 and everything else (x, epilogue) is L2/LDS traffic.
//
// Structure (MI355X-first, not a translation of anything in the reference, whose
// forward lives in HF transformers — hf_wrappers.py:417/478):
//   * one wave owns a PAIR of output rows at a time and all 64 lanes stride along K
//     with global_load_dwordx4: every wave-instruction moves 1 KiB of contiguous
//     weight bytes (full 128-B lines, no fragment-shaped loads);
//   * the T activation rows are staged ONCE per workgroup into LDS as bf16 (with
//     the RMSNorm / LayerNorm fused into the staging pass), and every 16-byte x
//     vector read from LDS is used for both rows of the pair;
//   * products use v_dot2c_f32_bf16 (2 MACs per lane-op, fp32 accumulate), so the
//     VALU stays far from being the limiter;
//   * loads are software-pipelined in two register buffers of 8 dwordx4 each:
//     16 KiB in flight per wave, 64 KiB per CU;
//   * the grid is sized to the chip (256 CUs), workgroups walk the row pairs with a
//     grid stride so x is staged once per workgroup, and for the small matrices of
//     Llama-3.2-1B/3B the K dimension is split over the waves of a workgroup
//     (KSPLIT) so that every CU gets the same number of bytes;
//   * the pair structure serves the fused epilogues: RoPE rotates rows (i, i+D/2)
//     of a head, SwiGLU combines (gate_n, up_n), residual/logit stores write the
//     two adjacent columns (2p, 2p+1) as one dword.

#include "kernels.h"

namespace sd {

constexpr int kGemvThreads = 256;
constexpr int kGemvWaves = kGemvThreads / kWave;
constexpr int kChunk = 4;  // k-steps (of 64 lanes x 8 elements) per register buffer

typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float dot2(uint32_t w, uint32_t x, float acc) {
  return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, w),
                                         __builtin_bit_cast(bf16x2_t, x), acc, false);
}

__device__ __forceinline__ float dot8(const u32x4& w, const u32x4& x, float acc) {
  acc = dot2(w.x, x.x, acc);
  acc = dot2(w.y, x.y, acc);
  acc = dot2(w.z, x.z, acc);
  acc = dot2(w.w, x.w, acc);
  return acc;
}

__device__ __forceinline__ float gelu_new(float x) {
  // GPT-2 "gelu_new": 0.5 x (1 + tanh( sqrt(2/pi) (x + 0.044715 x^3) ))
  const float c = 0.7978845608028654f;
  return 0.5f * x * (1.0f + tanhf(c * (x + 0.044715f * x * x * x)));
}

// row indices of pair p for each epilogue
template <int EPI>
__device__ __forceinline__ void pair_rows(const GemvArgs& a, int p, int& r0, int& r1) {
  if constexpr (EPI == EPI_QKV_ROPE) {
    const int half = a.head_dim >> 1;
    const int h = p / half, i = p - h * half;
    r0 = h * a.head_dim + i;
    r1 = r0 + half;
  } else if constexpr (EPI == EPI_SWIGLU) {
    r0 = p;
    r1 = p + a.n_pairs;  // up rows follow the gate rows
  } else {
    r0 = 2 * p;
    r1 = 2 * p + 1;
  }
}

// ------------------------------------------------------------------------------
// staging of x into LDS (bf16 [T][K]) with the fused normalisation
// ------------------------------------------------------------------------------
template <int TT>
__device__ __forceinline__ void stage_x(const GemvArgs& a, uint16_t* xs, float* red) {
  const int K = a.K, T = a.T;
  const int tid = threadIdx.x;
  const int nvec = K >> 3;
  const uint16_t* xin = static_cast<const uint16_t*>(a.x);

  if (a.prologue == PRO_NONE) {
    for (int t = 0; t < T; ++t) {
      const uint4* src = reinterpret_cast<const uint4*>(xin + static_cast<size_t>(t) * a.x_stride);
      uint4* dst = reinterpret_cast<uint4*>(xs + static_cast<size_t>(t) * K);
      for (int v = tid; v < nvec; v += kGemvThreads) dst[v] = src[v];
    }
    __syncthreads();
    return;
  }

  // pass 1: raw copy + per-token sum / sum of squares
  float s1[TT], s2[TT];
#pragma unroll
  for (int t = 0; t < TT; ++t) { s1[t] = 0.f; s2[t] = 0.f; }
#pragma unroll
  for (int t = 0; t < TT; ++t) {
    if (t < T) {
      const uint4* src = reinterpret_cast<const uint4*>(xin + static_cast<size_t>(t) * a.x_stride);
      uint4* dst = reinterpret_cast<uint4*>(xs + static_cast<size_t>(t) * K);
      for (int v = tid; v < nvec; v += kGemvThreads) {
        const uint4 q = src[v];
        dst[v] = q;
        const uint32_t w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float lo = __uint_as_float(w[j] << 16), hi = __uint_as_float(w[j] & 0xffff0000u);
          s1[t] += lo + hi;
          s2[t] += lo * lo + hi * hi;
        }
      }
    }
  }
  const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
  for (int t = 0; t < TT; ++t) {
    if (t < T) {
      const float a1 = wave_reduce_sum(s1[t]);
      const float a2 = wave_reduce_sum(s2[t]);
      if (lane == 0) {
        red[(wave * TT + t) * 2 + 0] = a1;
        red[(wave * TT + t) * 2 + 1] = a2;
      }
    }
  }
  __syncthreads();
  // pass 2: normalise in place. HF LlamaRMSNorm: weight * (x * rsqrt(var+eps)).to(bf16);
  // GPT-2 LayerNorm: (x-mean)*rsqrt(var+eps)*w + b computed in fp32, rounded once.
  const uint16_t* nw = static_cast<const uint16_t*>(a.norm_w);
  const uint16_t* nb = static_cast<const uint16_t*>(a.norm_b);
  const float invK = 1.0f / static_cast<float>(K);
#pragma unroll
  for (int t = 0; t < TT; ++t) {
    if (t < T) {
      float sum = 0.f, sq = 0.f;
#pragma unroll
      for (int w = 0; w < kGemvWaves; ++w) {
        sum += red[(w * TT + t) * 2 + 0];
        sq += red[(w * TT + t) * 2 + 1];
      }
      uint16_t* row = xs + static_cast<size_t>(t) * K;
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
  }
  __syncthreads();
}

// ------------------------------------------------------------------------------
// epilogues: lane t (< T) finishes token t of the pair (y0 = row r0, y1 = row r1)
// ------------------------------------------------------------------------------
template <int EPI>
__device__ __forceinline__ void epilogue(const GemvArgs& a, int p, int r0, int r1, int t, float y0,
                                         float y1, float& best_v, int& best_i) {
  const int b = t / a.M, m = t - b * a.M;
  if constexpr (EPI == EPI_QKV_ROPE) {
    if (a.bias) {
      const uint16_t* bs = static_cast<const uint16_t*>(a.bias);
      y0 += bf16_bits_to_float(bs[r0]);
      y1 += bf16_bits_to_float(bs[r1]);
    }
    const int D = a.head_dim, half = D >> 1;
    const int h = p / half, i = p - h * half;
    const int pos = a.pos_base[b] + a.pos_off + m;
    float o0 = y0, o1 = y1;
    if (a.rope_cos && h < a.n_q_heads + a.n_kv_heads && pos >= 0 && pos < a.max_pos) {
      const float c = a.rope_cos[static_cast<size_t>(pos) * half + i];
      const float s = a.rope_sin[static_cast<size_t>(pos) * half + i];
      o0 = y0 * c - y1 * s;
      o1 = y1 * c + y0 * s;
    }
    const uint16_t u0 = float_to_bf16_bits(o0), u1 = float_to_bf16_bits(o1);
    if (h < a.n_q_heads) {
      uint16_t* q = static_cast<uint16_t*>(a.out) + static_cast<size_t>(t) * a.out_stride + h * D + i;
      q[0] = u0;
      q[half] = u1;
    } else if (pos >= 0 && pos < a.l_max) {
      const bool is_k = h < a.n_q_heads + a.n_kv_heads;
      const int kvh = is_k ? h - a.n_q_heads : h - a.n_q_heads - a.n_kv_heads;
      uint16_t* cache = static_cast<uint16_t*>(is_k ? a.k_cache : a.v_cache);
      uint16_t* dst = cache + ((static_cast<size_t>(b) * a.n_kv_heads + kvh) * a.l_max + pos) * D + i;
      dst[0] = u0;
      dst[half] = u1;
    }
  } else if constexpr (EPI == EPI_RESID) {
    if (a.bias) {
      const uint16_t* bs = static_cast<const uint16_t*>(a.bias);
      y0 += bf16_bits_to_float(bs[r0]);
      y1 += bf16_bits_to_float(bs[r1]);
    }
    uint32_t* px = reinterpret_cast<uint32_t*>(static_cast<uint16_t*>(a.out) + static_cast<size_t>(t) * a.out_stride + r0);
    const uint32_t old = *px;
    const float n0 = __uint_as_float(old << 16) + y0;
    const float n1 = __uint_as_float(old & 0xffff0000u) + y1;
    *px = static_cast<uint32_t>(float_to_bf16_bits(n0)) | (static_cast<uint32_t>(float_to_bf16_bits(n1)) << 16);
  } else if constexpr (EPI == EPI_SWIGLU) {
    const float g = y0, u = y1;
    const float act = g / (1.0f + __expf(-g)) * u;
    static_cast<uint16_t*>(a.out)[static_cast<size_t>(t) * a.out_stride + p] = float_to_bf16_bits(act);
  } else if constexpr (EPI == EPI_GELU) {
    const uint16_t* bs = static_cast<const uint16_t*>(a.bias);
    if (bs) {
      y0 += bf16_bits_to_float(bs[r0]);
      y1 += bf16_bits_to_float(bs[r1]);
    }
    uint32_t* po = reinterpret_cast<uint32_t*>(static_cast<uint16_t*>(a.out) + static_cast<size_t>(t) * a.out_stride + r0);
    *po = static_cast<uint32_t>(float_to_bf16_bits(gelu_new(y0))) | (static_cast<uint32_t>(float_to_bf16_bits(gelu_new(y1))) << 16);
  } else {  // EPI_ARGMAX: logits are the bf16-rounded products, as a bf16 lm_head returns
    const uint16_t u0 = float_to_bf16_bits(y0), u1 = float_to_bf16_bits(y1);
    const float f0 = bf16_bits_to_float(u0), f1 = bf16_bits_to_float(u1);
    if (argmax_better(f0, r0, best_v, best_i)) { best_v = f0; best_i = r0; }
    if (r1 < a.N && argmax_better(f1, r1, best_v, best_i)) { best_v = f1; best_i = r1; }
    if (a.out) {
      if (a.out_dtype == SD_F32) {
        float* lo = static_cast<float*>(a.out) + static_cast<size_t>(t) * a.out_stride;
        lo[r0] = f0;
        if (r1 < a.N) lo[r1] = f1;
      } else {
        uint16_t* lo = static_cast<uint16_t*>(a.out) + static_cast<size_t>(t) * a.out_stride;
        lo[r0] = u0;
        if (r1 < a.N) lo[r1] = u1;
      }
    }
  }
}

// ------------------------------------------------------------------------------
// main kernel
// ------------------------------------------------------------------------------
template <int TT, int EPI, int KSPLIT>
__global__ __launch_bounds__(kGemvThreads) void gemv_pairs_kernel(const GemvArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  uint16_t* xs = reinterpret_cast<uint16_t*>(smem);
  float* red = reinterpret_cast<float*>(smem + static_cast<size_t>(TT) * a.K * 2);  // [2][waves][2][TT]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  constexpr int PPI = kGemvWaves / KSPLIT;  // pairs per workgroup iteration
  const int sub = wave / KSPLIT;            // which pair of the iteration
  const int kpart = wave % KSPLIT;          // which K slice
  const int K = a.K, T = a.T;
  // k-steps of 512 elements (64 lanes x 8) for this wave; K only has to be a
  // multiple of 8: lanes past the end of the row are masked off
  const int steps = (((K + 511) >> 9) + KSPLIT - 1) / KSPLIT;
  const int kbase = kpart * steps * 512 + lane * 8;
  const uint16_t* W = static_cast<const uint16_t*>(a.W);

  const int n_groups = (a.n_pairs + PPI - 1) / PPI;
  const int n_iter = (n_groups - static_cast<int>(blockIdx.x) + static_cast<int>(gridDim.x) - 1) / static_cast<int>(gridDim.x);
  const int n_chunks = (steps + kChunk - 1) / kChunk;
  const int total = n_iter * n_chunks;  // flattened (iteration, chunk) stream of this wave

  // one register buffer = kChunk k-steps x 2 rows
  u32x4 bufA[2 * kChunk], bufB[2 * kChunk];

  auto issue = [&](u32x4* buf, int idx) {
    const int it = idx / n_chunks, c = idx - it * n_chunks;
    const int p = (it * static_cast<int>(gridDim.x) + static_cast<int>(blockIdx.x)) * PPI + sub;
    if (p >= a.n_pairs) return;
    int r0, r1;
    pair_rows<EPI>(a, p, r0, r1);
    if (r1 >= a.N) r1 = r0;  // odd N (lm_head tail): read a valid row, result is dropped
    const uint16_t* w0 = W + static_cast<size_t>(r0) * K + kbase;
    const uint16_t* w1 = W + static_cast<size_t>(r1) * K + kbase;
#pragma unroll
    for (int j = 0; j < kChunk; ++j) {
      const int s = c * kChunk + j;
      if (s < steps && kbase + s * 512 < K) {
        buf[2 * j + 0] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(w0 + s * 512));
        buf[2 * j + 1] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(w1 + s * 512));
      }
    }
  };

  // start the weight stream before x is staged: HBM latency hides under the prologue
  if (total > 0) issue(bufA, 0);

  stage_x<TT>(a, xs, red);
  __syncthreads();

  float acc0[TT], acc1[TT];
#pragma unroll
  for (int t = 0; t < TT; ++t) { acc0[t] = 0.f; acc1[t] = 0.f; }
  float best_v = -INFINITY;
  int best_i = 0x7fffffff;

  auto consume = [&](const u32x4* buf, int idx) {
    const int it = idx / n_chunks, c = idx - it * n_chunks;
    const int p = (it * static_cast<int>(gridDim.x) + static_cast<int>(blockIdx.x)) * PPI + sub;
    const bool valid = p < a.n_pairs;
    if (valid) {
#pragma unroll
      for (int j = 0; j < kChunk; ++j) {
        const int s = c * kChunk + j;
        if (s < steps && kbase + s * 512 < K) {
          const uint16_t* xk = xs + kbase + s * 512;
#pragma unroll
          for (int t = 0; t < TT; ++t) {
            if (t < T) {
              const u32x4 xv = *reinterpret_cast<const u32x4*>(xk + static_cast<size_t>(t) * K);
              acc0[t] = dot8(buf[2 * j + 0], xv, acc0[t]);
              acc1[t] = dot8(buf[2 * j + 1], xv, acc1[t]);
            }
          }
        }
      }
    }
    if (c != n_chunks - 1) return;
    // pair finished: wave reduce, (cross-wave reduce), epilogue
    float my0 = 0.f, my1 = 0.f;
#pragma unroll
    for (int t = 0; t < TT; ++t) {
      if (t < T) {
        const float v0 = wave_reduce_sum(acc0[t]);
        const float v1 = wave_reduce_sum(acc1[t]);
        if (lane == t) { my0 = v0; my1 = v1; }
      }
      acc0[t] = 0.f;
      acc1[t] = 0.f;
    }
    if constexpr (KSPLIT > 1) {
      float* slot = red + (it & 1) * (kGemvWaves * 2 * TT);
      if (lane < T) {
        slot[(wave * 2 + 0) * TT + lane] = my0;
        slot[(wave * 2 + 1) * TT + lane] = my1;
      }
      __syncthreads();  // uniform: every wave of the workgroup has the same trip count
      if (kpart != 0) return;
      if (lane < T) {
#pragma unroll
        for (int w = 1; w < KSPLIT; ++w) {
          my0 += slot[((wave + w) * 2 + 0) * TT + lane];
          my1 += slot[((wave + w) * 2 + 1) * TT + lane];
        }
      }
    }
    if (valid && lane < T) {
      int r0, r1;
      pair_rows<EPI>(a, p, r0, r1);
      epilogue<EPI>(a, p, r0, r1, lane, my0, my1, best_v, best_i);
    }
  };

  for (int i = 0; i < total; i += 2) {
    if (i + 1 < total) issue(bufB, i + 1);
    consume(bufA, i);
    if (i + 2 < total) issue(bufA, i + 2);
    if (i + 1 < total) consume(bufB, i + 1);
  }

  if constexpr (EPI == EPI_ARGMAX) {
    // lane t of the leading waves holds token t's running best: fold the waves via LDS
    __syncthreads();
    float* sv = red;
    int* si = reinterpret_cast<int*>(red + kGemvWaves * TT);
    if (lane < T) {
      sv[wave * TT + lane] = best_v;
      si[wave * TT + lane] = best_i;
    }
    __syncthreads();
    if (wave == 0 && lane < T) {
#pragma unroll
      for (int w = 1; w < kGemvWaves; ++w) {
        const float ov = sv[w * TT + lane];
        const int oi = si[w * TT + lane];
        if (argmax_better(ov, oi, best_v, best_i)) { best_v = ov; best_i = oi; }
      }
      a.part_val[static_cast<size_t>(lane) * gridDim.x + blockIdx.x] = best_v;
      a.part_idx[static_cast<size_t>(lane) * gridDim.x + blockIdx.x] = best_i;
    }
  }
}

// ------------------------------------------------------------------------------
// host-side launch
// ------------------------------------------------------------------------------
static int pick_ksplit(int n_pairs, int K) {
  // Give all 1024 waves (256 CUs x 4) the same number of (pair, K-slice) units.
  const int steps = (K + 511) >> 9;
  for (int ks : {1, 2, 4}) {
    if (steps % ks) break;
    const long units = static_cast<long>(n_pairs) * ks;
    if (units >= 1024 && (units % 1024 == 0 || units >= 8 * 1024)) return ks;
  }
  int best = 1;
  for (int ks : {1, 2, 4})
    if (steps % ks == 0 && steps / ks >= 1) best = ks;
  return (static_cast<long>(n_pairs) >= 4096) ? 1 : best;
}

template <int TT, int EPI, int KS>
static int launch_one(const GemvArgs& a, int grid, size_t smem, hipStream_t st) {
  // dynamic LDS above 64 KiB has to be opted into once per kernel
  static bool attr_set = false;
  if (!attr_set) {
    SD_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemv_pairs_kernel<TT, EPI, KS>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_set = true;
  }
  hipLaunchKernelGGL((gemv_pairs_kernel<TT, EPI, KS>), dim3(grid), dim3(kGemvThreads), smem, st, a);
  SD_LAUNCH_CHECK();
  return 0;
}

template <int TT, int EPI>
static int launch_tt(const GemvArgs& a, int ksplit, int grid, size_t smem, hipStream_t st) {
  switch (ksplit) {
    case 1: return launch_one<TT, EPI, 1>(a, grid, smem, st);
    case 2: return launch_one<TT, EPI, 2>(a, grid, smem, st);
    default: return launch_one<TT, EPI, 4>(a, grid, smem, st);
  }
}

template <int EPI>
static int launch_epi(const GemvArgs& a, int tt, int ksplit, int grid, size_t smem, hipStream_t st) {
  switch (tt) {
    case 1: return launch_tt<1, EPI>(a, ksplit, grid, smem, st);
    case 2: return launch_tt<2, EPI>(a, ksplit, grid, smem, st);
    case 3: return launch_tt<3, EPI>(a, ksplit, grid, smem, st);
    case 5: return launch_tt<5, EPI>(a, ksplit, grid, smem, st);
    default: return launch_tt<9, EPI>(a, ksplit, grid, smem, st);
  }
}

int gemv_tile_for(int T) {
  if (T <= 1) return 1;
  if (T <= 2) return 2;
  if (T <= 3) return 3;
  if (T <= 5) return 5;
  return 9;
}

int gemv_grid(const GemvArgs& a, int* ksplit_out) {
  const int ks = pick_ksplit(a.n_pairs, a.K);
  const int ppi = kGemvWaves / ks;
  const int groups = (a.n_pairs + ppi - 1) / ppi;
  int grid = groups < 256 ? groups : 256;
  // big matrices (lm_head, gate/up): two workgroups per CU when LDS allows
  const size_t smem = static_cast<size_t>(gemv_tile_for(a.T)) * a.K * 2 + 1024;
  if (groups >= 2048 && smem <= 72 * 1024) grid = 512;
  *ksplit_out = ks;
  return grid;
}

int launch_gemv(const GemvArgs& a, int epi, hipStream_t st) {
  SD_REQUIRE(a.T >= 1 && a.T <= kGemvMaxT, "gemv: T=%d out of range 1..%d", a.T, kGemvMaxT);
  SD_REQUIRE(a.K % 8 == 0 && a.x_stride % 8 == 0, "gemv: K=%d / x_stride=%d must be multiples of 8", a.K, a.x_stride);
  SD_REQUIRE(a.n_pairs > 0, "gemv: no rows");
  const int tt = gemv_tile_for(a.T);
  int ks = 1;
  const int grid = gemv_grid(a, &ks);
  const size_t smem = static_cast<size_t>(tt) * a.K * 2 + 1024;
  SD_REQUIRE(smem <= 160 * 1024, "gemv: T=%d x K=%d does not fit LDS", a.T, a.K);
  switch (epi) {
    case EPI_QKV_ROPE: return launch_epi<EPI_QKV_ROPE>(a, tt, ks, grid, smem, st);
    case EPI_RESID: return launch_epi<EPI_RESID>(a, tt, ks, grid, smem, st);
    case EPI_SWIGLU: return launch_epi<EPI_SWIGLU>(a, tt, ks, grid, smem, st);
    case EPI_GELU: return launch_epi<EPI_GELU>(a, tt, ks, grid, smem, st);
    case EPI_ARGMAX: return launch_epi<EPI_ARGMAX>(a, tt, ks, grid, smem, st);
    default: SD_REQUIRE(false, "gemv: unknown epilogue %d", epi);
  }
  return 0;
}

}  // namespace sd
