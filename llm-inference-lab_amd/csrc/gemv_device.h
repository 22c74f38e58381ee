// Device pieces shared by the weight-streaming kernels (gemv.hip: T <= 9 tokens,
// gemm_skinny.hip: T <= 64): constants, MFMA operand types, the row-pair mapping and the
// fused epilogues. Included by .hip files only.
#pragma once

#include "kernels.h"

namespace sd {

constexpr int kGemvThreads = 1024;
constexpr int kGemvWaves = kGemvThreads / kWave;  // 16 waves: 4 per SIMD
constexpr int kXPad = 8;                          // bf16 elements of padding per staged x row

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float gelu_new(float x) {
  // GPT-2 "gelu_new": 0.5 x (1 + tanh( sqrt(2/pi) (x + 0.044715 x^3) ))
  const float c = 0.7978845608028654f;
  return 0.5f * x * (1.0f + tanhf(c * (x + 0.044715f * x * x * x)));
}

// Kernel arguments into SGPRs at kernel entry: hipcc loads a by-value argument struct lazily, next to each first
// use, which puts 5-6 DEPENDENT scalar-cache round trips (s_load -> s_waitcnt) in front of the first vector load of
// these latency-bound kernels (0.7 us from wave entry to the first weight load in the in-kernel timeline). One empty
// asm statement that takes the fields as SGPR INPUT operands makes all of them live at that point: the s_loads are issued
// back to back (merged into x8 / x16 loads) and waited for once. (Inputs only: as in/out operands the pointers lose
// their address space, every load becomes a flat_load and the counted vmcnt waits of the prologue are gone.)
// (SD_PIN: common.h)
template <int EPI, bool W8>
__device__ __forceinline__ void pin_gemv_args(const GemvArgs& a) {
  SD_PIN("s"(a.W), "s"(a.x), "s"(a.out), "s"(a.norm_w), "s"(a.norm_b), "s"(a.bias), "s"(a.x_row), "s"(a.skip_k), "s"(a.debug_ts),
         "s"(a.N), "s"(a.K), "s"(a.n_pairs), "s"(a.kw), "s"(a.x_stride), "s"(a.out_stride), "s"(a.m_magic), "s"(a.norm_eps),
         "s"(static_cast<int>(a.T)), "s"(static_cast<int>(a.M)), "s"(static_cast<int>(a.ppw)), "s"(static_cast<int>(a.n_tiles_full)),
         "s"(static_cast<int>(a.tile_pairs)), "s"(static_cast<int>(a.ksplit)), "s"(static_cast<int>(a.alias_part)),
         "s"(static_cast<int>(a.packed)), "s"(static_cast<int>(a.prologue)), "s"(static_cast<int>(a.ks_shift)),
         "s"(static_cast<int>(a.skip_i)));
  if constexpr (W8) SD_PIN("s"(a.w_scale));
  if constexpr (EPI == EPI_QKV_ROPE)
    SD_PIN("s"(static_cast<int>(a.head_dim)), "s"(static_cast<int>(a.n_q_heads)), "s"(static_cast<int>(a.n_kv_heads)), "s"(a.pos_base), "s"(a.pos_off), "s"(a.rope_cos),
           "s"(a.rope_sin), "s"(a.max_pos), "s"(a.k_cache), "s"(a.v_cache), "s"(a.l_max), "s"(static_cast<int>(a.half_shift)),
           "s"(a.block_table), "s"(static_cast<int>(a.page_shift)));   // (left out, the epilogue waited for a scalar load of its own: 0.5 % of the step)
  if constexpr (EPI == EPI_ARGMAX) SD_PIN("s"(static_cast<int>(a.out_dtype)), "s"(a.part_val), "s"(a.part_idx), "s"(a.batch_bytes));
}

// row indices of pair p for each epilogue
template <int EPI>
__device__ __forceinline__ void pair_rows(const GemvArgs& a, int p, int& r0, int& r1) {
  if constexpr (EPI == EPI_QKV_ROPE) {
    const int half = a.head_dim >> 1;
    const int h = (a.half_shift >= 0) ? (p >> a.half_shift) : p / half, i = p - h * half;
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
// epilogues: one lane finishes token t of pair p (y0 = row r0, y1 = row r1)
// ------------------------------------------------------------------------------
//   COH (EPI_RESID only): the new residual row is stored with an agent-scope relaxed atomic store (write-through to
//   memory), for a consumer on another XCD inside the same launch.
template <int EPI, bool COH = false>
__device__ __forceinline__ void epilogue(const GemvArgs& a, int p, int r0, int r1, int t, float y0,
                                         float y1, float& best_v, int& best_i, bool have_old = false,
                                         uint32_t old_pre = 0, float* st_sq = nullptr, float* st_sum = nullptr) {
  const int b = static_cast<int>((static_cast<unsigned>(t) * a.m_magic) >> 16), m = t - b * a.M;   // t / M (gemv_derive)
  if constexpr (EPI == EPI_QKV_ROPE) {
    if (a.bias) {
      const uint16_t* bs = static_cast<const uint16_t*>(a.bias);
      y0 += bf16_bits_to_float(bs[r0]);
      y1 += bf16_bits_to_float(bs[r1]);
    }
    const int D = a.head_dim, half = D >> 1;
    const int h = (a.half_shift >= 0) ? (p >> a.half_shift) : p / half, i = p - h * half;
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
      // in-place KV append (the fused form of kv_append_ref, reference.py:59-93):
      // K rows are [Lmax][D], V is kept transposed [D][Lmax] (see attention.hip)
      const bool is_k = h < a.n_q_heads + a.n_kv_heads;
      // dense: "page" b of l_max positions; paged: the row's page for this position, P positions each
      size_t slab = static_cast<size_t>(b);
      int off = pos, plen = a.l_max;
      if (a.block_table) {
        slab = static_cast<size_t>(a.block_table[b * (a.l_max >> a.page_shift) + (pos >> a.page_shift)]);
        plen = 1 << a.page_shift;
        off = pos & (plen - 1);
      }
      if (is_k) {
        const int kvh = h - a.n_q_heads;
        uint16_t* dst = static_cast<uint16_t*>(a.k_cache) + ((slab * a.n_kv_heads + kvh) * plen + off) * D + i;
        dst[0] = u0;
        dst[half] = u1;
      } else {
        const int kvh = h - a.n_q_heads - a.n_kv_heads;
        uint16_t* dst = static_cast<uint16_t*>(a.v_cache) + ((slab * a.n_kv_heads + kvh) * D + i) * plen + off;
        dst[0] = u0;
        dst[static_cast<size_t>(half) * plen] = u1;
      }
    }
  } else if constexpr (EPI == EPI_RESID) {
    if (a.bias) {
      const uint16_t* bs = static_cast<const uint16_t*>(a.bias);
      y0 += bf16_bits_to_float(bs[r0]);
      y1 += bf16_bits_to_float(bs[r1]);
    }
    uint32_t* px = reinterpret_cast<uint32_t*>(static_cast<uint16_t*>(a.out) + static_cast<size_t>(t) * a.out_stride + r0);
    const uint32_t old = have_old ? old_pre : *px;  // prefetched at kernel entry when possible
    const float n0 = __uint_as_float(old << 16) + y0;
    const float n1 = __uint_as_float(old & 0xffff0000u) + y1;
    const uint32_t nv = static_cast<uint32_t>(float_to_bf16_bits(n0)) | (static_cast<uint32_t>(float_to_bf16_bits(n1)) << 16);
    if constexpr (COH) __hip_atomic_store(px, nv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else *px = nv;
    if (st_sq) {   // statistics of the row AS STORED (bf16-rounded), for the launch that normalises it (GemvArgs::xstat_out)
      const float r0f = __uint_as_float(nv << 16), r1f = __uint_as_float(nv & 0xffff0000u);
      *st_sq += r0f * r0f + r1f * r1f;
      *st_sum += r0f + r1f;
    }
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

// EPI_RESID launches at > 9 tokens: thread `tid` holds, for tokens 16 q + (tid & 15), the (sum of squares, sum) of the new
// row values of the pairs it finished. Folded over the workgroup in a fixed order (lanes that share a token: xor 16, 32;
// then the 16 waves through `scratch`, >= 16 * TG * 32 floats of LDS nobody else is using) and written as this
// workgroup's partial of every token. Called by all threads of the workgroup.
template <int TG>
__device__ __forceinline__ void resid_stats_publish(const GemvArgs& a, const float (&sq)[TG], const float (&sm)[TG], float* scratch, int tid) {
  const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
  for (int q = 0; q < TG; ++q) {
    float v2 = sq[q], v1 = sm[q];
    v2 += __shfl_xor(v2, 16, 64);
    v2 += __shfl_xor(v2, 32, 64);
    v1 += __shfl_xor(v1, 16, 64);
    v1 += __shfl_xor(v1, 32, 64);
    if (lane < 16) {
      scratch[((wave * TG + q) * 16 + lane) * 2 + 0] = v2;
      scratch[((wave * TG + q) * 16 + lane) * 2 + 1] = v1;
    }
  }
  __syncthreads();
  if (tid < 16 * TG) {
    const int q = tid >> 4, tl = tid & 15, t = 16 * q + tl;
    if (t < a.T) {
      float v2 = 0.f, v1 = 0.f;
      for (int w = 0; w < kGemvWaves; ++w) {
        v2 += scratch[((w * TG + q) * 16 + tl) * 2 + 0];
        v1 += scratch[((w * TG + q) * 16 + tl) * 2 + 1];
      }
      a.xstat_out[static_cast<size_t>(t) * kStatStride + blockIdx.x] = v2;
      a.xstat_out[kStatPlane + static_cast<size_t>(t) * kStatStride + blockIdx.x] = v1;
    }
  }
}

// Sum of the K-slice partials of one (pair, token) item: slice w's 16x16 tile starts `stride` floats after slice w-1's;
// first row of the pair at base[0], second row 8 rows (128 floats) further. The slice count is dispatched to a compile-time
// constant so that all 2 x ksplit LDS reads are issued before the first add — as a loop over the run-time count they were
// up to 32 dependent LDS round trips per item (0.7 us of a 5-token out- or down-projection launch). Ascending slice order in
// both forms: the sums are bit-identical.
template <int KS>
__device__ __forceinline__ void sum_slices_fixed(const float* base, int stride, float& y0, float& y1) {
  float a[KS], b[KS];
#pragma unroll
  for (int w = 0; w < KS; ++w) {
    a[w] = base[w * stride];
    b[w] = base[w * stride + 128];
  }
  y0 = 0.f;
  y1 = 0.f;
#pragma unroll
  for (int w = 0; w < KS; ++w) {
    y0 += a[w];
    y1 += b[w];
  }
}
__device__ __forceinline__ void sum_slices(const float* base, int stride, int ksplit, float& y0, float& y1) {
  switch (ksplit) {   // a power of two <= 16 (gemv_geometry)
    case 16: sum_slices_fixed<16>(base, stride, y0, y1); break;
    case 8: sum_slices_fixed<8>(base, stride, y0, y1); break;
    case 4: sum_slices_fixed<4>(base, stride, y0, y1); break;
    case 2: sum_slices_fixed<2>(base, stride, y0, y1); break;
    default: sum_slices_fixed<1>(base, stride, y0, y1); break;
  }
}

}  // namespace sd
