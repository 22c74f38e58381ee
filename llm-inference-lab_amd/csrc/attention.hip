// K-token parallel attention over the appended KV cache (draft decode: M = 1..2,
// verify: M = K+1) for gfx950.
//
// Replaces what the reference gets from HF transformers' attention inside its k
// sequential forwards (hf_wrappers.py:417/478): here all M new positions of a row are
// scored against the cache in one pass, causally (query m sees keys 0 .. pos0+m).
//
// One workgroup = one (batch row, kv head, tile of up to 8 query rows); the query
// rows of a kv head are its G = Hq/Hkv query heads x M positions, so K and V are
// streamed once per tile instead of once per query head. K/V rows are read with
// 16-byte loads, consecutive lanes on consecutive bytes of consecutive keys
// (keys are contiguous in the [B][Hkv][Lmax][D] cache), the query tile is staged
// in LDS pre-scaled, softmax is online over 64-key chunks (wave shuffles for
// max/sum), and the PV accumulators live in registers.
//
// v1 uses VALU FMAs: at decode sizes this kernel moves <1% of a step's bytes.

#include "kernels.h"

namespace sd {

constexpr int kAttnThreads = 256;
constexpr int kAttnRows = 8;     // query rows per workgroup
constexpr int kAttnChunk = 64;   // keys per online-softmax chunk

__global__ __launch_bounds__(kAttnThreads) void attention_kernel(const AttnArgs a) {
  const int kvh = blockIdx.x, b = blockIdx.y, tile = blockIdx.z;
  const int D = a.head_dim, G = a.n_q_heads / a.n_kv_heads, M = a.M;
  const int R = G * M;
  const int r_base = tile * kAttnRows;
  const int rows = min(kAttnRows, R - r_base);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int LPK = D >> 3;                    // lanes per key (8 elements each)
  const int keys_per_pass = kAttnThreads / LPK;

  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* q_s = reinterpret_cast<float*>(smem);                 // [kAttnRows][D]
  float* p_s = q_s + kAttnRows * D;                            // [kAttnRows][kAttnChunk]
  float* alpha_s = p_s + kAttnRows * kAttnChunk;               // [kAttnRows]
  float* m_s = alpha_s + kAttnRows;                            // running max
  float* l_s = m_s + kAttnRows;                                // running sum
  float* o_s = l_s + kAttnRows;                                // [4 waves][kAttnRows][D]

  const int pos0 = a.pos_base[b] + a.pos_off;  // position of query m = 0
  const int n_keys = min(pos0 + M, a.l_max);   // keys visible to the last query

  // stage the query tile (row r = g*M + m -> head kvh*G+g, token b*M+m), pre-scaled
  const uint16_t* q = static_cast<const uint16_t*>(a.q);
  const int qstride = a.n_q_heads * D;
  for (int i = tid; i < kAttnRows * D; i += kAttnThreads) {
    const int r = i / D, d = i - r * D;
    float v = 0.f;
    if (r < rows) {
      const int rr = r_base + r, g = rr / M, m = rr - g * M;
      v = bf16_bits_to_float(q[static_cast<size_t>(b * M + m) * qstride + (kvh * G + g) * D + d]) * a.scale;
    }
    q_s[i] = v;
  }
  if (tid < kAttnRows) { m_s[tid] = -INFINITY; l_s[tid] = 0.f; }
  __syncthreads();

  const uint16_t* kc = static_cast<const uint16_t*>(a.k_cache) + (static_cast<size_t>(b) * a.n_kv_heads + kvh) * a.l_max * D;
  const uint16_t* vc = static_cast<const uint16_t*>(a.v_cache) + (static_cast<size_t>(b) * a.n_kv_heads + kvh) * a.l_max * D;

  // PV mapping: dv = 8-wide slice of D, kg = key group
  const int DV = LPK;
  const int dv = tid % DV, kg = tid / DV;
  const int n_kg = kAttnThreads / DV;
  float acc[kAttnRows][8];
#pragma unroll
  for (int r = 0; r < kAttnRows; ++r)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[r][j] = 0.f;

  for (int c0 = 0; c0 < n_keys; c0 += kAttnChunk) {
    // ---- scores for keys [c0, c0+64) ------------------------------------------------
    const int part = tid % LPK;
    for (int kk = tid / LPK; kk < kAttnChunk; kk += keys_per_pass) {
      const int key = c0 + kk;
      float kf[8];
      if (key < n_keys) {
        const uint4 kv = *reinterpret_cast<const uint4*>(kc + static_cast<size_t>(key) * D + part * 8);
        const uint32_t w[4] = {kv.x, kv.y, kv.z, kv.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          kf[2 * j] = __uint_as_float(w[j] << 16);
          kf[2 * j + 1] = __uint_as_float(w[j] & 0xffff0000u);
        }
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) kf[j] = 0.f;
      }
#pragma unroll
      for (int r = 0; r < kAttnRows; ++r) {
        const float4 qa = *reinterpret_cast<const float4*>(q_s + r * D + part * 8);
        const float4 qb = *reinterpret_cast<const float4*>(q_s + r * D + part * 8 + 4);
        float s = qa.x * kf[0] + qa.y * kf[1] + qa.z * kf[2] + qa.w * kf[3] +
                  qb.x * kf[4] + qb.y * kf[5] + qb.z * kf[6] + qb.w * kf[7];
        for (int off = 1; off < LPK; off <<= 1) s += __shfl_xor(s, off, 64);
        if (part == 0) {
          // causal mask: row r is query m, which sees keys <= pos0 + m
          const int rr = r_base + r, m = rr % M;
          const bool vis = (r < rows) && (key < n_keys) && (key <= pos0 + m);
          p_s[r * kAttnChunk + kk] = vis ? s : -INFINITY;
        }
      }
    }
    __syncthreads();
    // ---- online softmax: wave w owns rows w, w+4 ------------------------------------
    for (int r = wave; r < kAttnRows; r += kAttnThreads / kWave) {
      const float s = p_s[r * kAttnChunk + lane];
      const float mx = wave_reduce_max(s);
      const float m_old = m_s[r];
      const float m_new = fmaxf(m_old, mx);
      float p = 0.f, alpha = 1.f;
      if (m_new > -INFINITY) {
        p = (s > -INFINITY) ? __expf(s - m_new) : 0.f;
        alpha = (m_old > -INFINITY) ? __expf(m_old - m_new) : 0.f;
      }
      const float sum = wave_reduce_sum(p);
      p_s[r * kAttnChunk + lane] = p;
      if (lane == 0) {
        alpha_s[r] = alpha;
        m_s[r] = m_new;
        l_s[r] = l_s[r] * alpha + sum;
      }
    }
    __syncthreads();
    // ---- PV --------------------------------------------------------------------------
#pragma unroll
    for (int r = 0; r < kAttnRows; ++r) {
      const float al = alpha_s[r];
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[r][j] *= al;
    }
    for (int kk = kg; kk < kAttnChunk; kk += n_kg) {
      const int key = c0 + kk;
      if (key < n_keys) {
        const uint4 vv = *reinterpret_cast<const uint4*>(vc + static_cast<size_t>(key) * D + dv * 8);
        const uint32_t w[4] = {vv.x, vv.y, vv.z, vv.w};
        float vf[8];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          vf[2 * j] = __uint_as_float(w[j] << 16);
          vf[2 * j + 1] = __uint_as_float(w[j] & 0xffff0000u);
        }
#pragma unroll
        for (int r = 0; r < kAttnRows; ++r) {
          const float p = p_s[r * kAttnChunk + kk];
#pragma unroll
          for (int j = 0; j < 8; ++j) acc[r][j] += p * vf[j];
        }
      }
    }
    __syncthreads();
  }

  // ---- fold the key groups: lanes that share dv inside a wave, then the 4 waves ------
#pragma unroll
  for (int r = 0; r < kAttnRows; ++r)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float v = acc[r][j];
      for (int off = DV; off < kWave; off <<= 1) v += __shfl_xor(v, off, 64);
      acc[r][j] = v;
    }
  if (lane < DV) {
#pragma unroll
    for (int r = 0; r < kAttnRows; ++r)
#pragma unroll
      for (int j = 0; j < 8; ++j) o_s[(wave * kAttnRows + r) * D + lane * 8 + j] = acc[r][j];
  }
  __syncthreads();
  uint16_t* out = static_cast<uint16_t*>(a.out);
  for (int i = tid; i < rows * D; i += kAttnThreads) {
    const int r = i / D, d = i - r * D;
    float v = 0.f;
#pragma unroll
    for (int w = 0; w < kAttnThreads / kWave; ++w) v += o_s[(w * kAttnRows + r) * D + d];
    const float l = l_s[r];
    v = (l > 0.f) ? v / l : 0.f;
    const int rr = r_base + r, g = rr / M, m = rr - g * M;
    out[static_cast<size_t>(b * M + m) * qstride + (kvh * G + g) * D + d] = float_to_bf16_bits(v);
  }
}

int launch_attention(const AttnArgs& a, hipStream_t st) {
  SD_REQUIRE(a.head_dim == 32 || a.head_dim == 64 || a.head_dim == 128 || a.head_dim == 256,
             "attention: head_dim %d not in {32,64,128,256}", a.head_dim);
  SD_REQUIRE(a.n_kv_heads > 0 && a.n_q_heads % a.n_kv_heads == 0, "attention: Hq %% Hkv != 0");
  SD_REQUIRE(a.B >= 1 && a.M >= 1, "attention: empty batch");
  const int G = a.n_q_heads / a.n_kv_heads;
  const int R = G * a.M;
  const int tiles = (R + kAttnRows - 1) / kAttnRows;
  const int D = a.head_dim;
  const size_t smem = sizeof(float) * (static_cast<size_t>(kAttnRows) * D + kAttnRows * kAttnChunk +
                                       3 * kAttnRows + 4 * kAttnRows * D);
  hipLaunchKernelGGL(attention_kernel, dim3(a.n_kv_heads, a.B, tiles), dim3(kAttnThreads), smem, st, a);
  SD_LAUNCH_CHECK();
  return 0;
}

}  // namespace sd
