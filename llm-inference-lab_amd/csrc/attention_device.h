// Device-side body of the K-token attention (shared by attention.hip and the fused
// attention + out-projection variant of gemv.hip). See attention.hip for the design notes.
#pragma once

#include "kernels.h"

namespace sd {

// four floats another workgroup wrote in THIS launch: two agent-scope relaxed 8-byte loads (L2 bypass)
__device__ __forceinline__ float4 coherent_load_f4(const float* p) {
  const unsigned long long* q = reinterpret_cast<const unsigned long long*>(p);
  const unsigned long long lo = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const unsigned long long hi = __hip_atomic_load(q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return float4{__uint_as_float(static_cast<unsigned>(lo)), __uint_as_float(static_cast<unsigned>(lo >> 32)),
                __uint_as_float(static_cast<unsigned>(hi)), __uint_as_float(static_cast<unsigned>(hi >> 32))};
}


constexpr int kAttnMaxWaves = 16;  // waves per workgroup: 4, 8 or 16 (template parameter NW)
constexpr int kAttnRows = 16;    // query rows per workgroup (MFMA tile)
constexpr int kAttnBlock = 32;   // keys per block
constexpr int kAttnSplitBlocks = 16;  // a workgroup is worth adding per this many key blocks (512 keys)
constexpr int kAttnMaxSplit = 32;

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
  return static_cast<uint32_t>(float_to_bf16_bits(lo)) | (static_cast<uint32_t>(float_to_bf16_bits(hi)) << 16);
}

// One (batch row, kv head, 16-query-row tile). Called by a whole workgroup of NW waves, which split the
// 32-key blocks round-robin; every thread takes the barriers.
// With a.n_split > 1 the keys of a tile are shared by up to n_split workgroups (`split` = this one's
// index): the number actually used, s_eff, follows the row's current length (one workgroup per 512
// keys), the others leave at once. Each computes an un-normalised partial over its blocks; the last to
// arrive (device-scope counter) merges them — flash-decoding across CUs, for contexts where one
// workgroup per kv head would walk thousands of keys while 248 CUs idle.
// PAGED (sd_model_bind_paged): the caches are page pools; a 32-key block never straddles a page (P is a multiple of 32),
// so each block costs one wave-uniform table read and the V^T row stride is P instead of l_max.
template <int D, int NW, bool PAGED = false>
__device__ __forceinline__ void attention_tile(const AttnArgs& a, int kvh, int b, int tile, unsigned char* smem,
                                               bool active, int split = 0) {
  constexpr int kAttnWaves = NW, kAttnThreads = NW * 64;
  constexpr int NKS = D / 32;  // k-steps of the QK^T contraction
  constexpr int NDT = D / 16;  // 16-wide tiles of the output channels
  const int G = a.n_q_heads / a.n_kv_heads, M = a.M;
  const int R = G * M;
  const int r_base = tile * kAttnRows;
  const int rows = min(kAttnRows, R - r_base);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 4, n = lane & 15;

  float* o_s = reinterpret_cast<float*>(smem);            // [waves][16][D]
  float* m_s = o_s + kAttnWaves * kAttnRows * D;          // [waves][16]
  float* l_s = m_s + kAttnWaves * kAttnRows;              // [waves][16]

  // per-row adaptive K: a launch of a draft forward no row needs leaves here (the arguments are first needed here, so
  // the check adds no wait of its own; workgroup-uniform, before any barrier)
  SD_SKIP_IF_INACTIVE(a.skip_k, a.skip_i);
  const int pos0 = a.pos_base[b] + a.pos_off;             // position of query m = 0
  const int n_keys = max(0, min(pos0 + M, a.l_max));      // keys visible to the last query
  const int n_blocks = (n_keys + kAttnBlock - 1) / kAttnBlock;
  int s_eff = 1;
  if (a.n_split > 1) {
    s_eff = min(a.n_split, max(1, (n_blocks + kAttnSplitBlocks - 1) / kAttnSplitBlocks));
    if (split >= s_eff) return;  // workgroup-uniform, before any barrier
  }

  // ---- Q fragments: lane (g, n) holds Q[row n][32 s + 8 g .. +8] for s < NKS -------
  const int qstride = a.n_q_heads * D;
  u32x4 qf[NKS];
  int my_m = 0;  // query position index of row n (for the causal mask)
  {
    const uint16_t* q = static_cast<const uint16_t*>(a.q);
    const bool valid = n < rows;
    const int rr = r_base + (valid ? n : 0);
    const int gh = rr / M, m = rr - gh * M;
    my_m = m;
    const uint16_t* qrow = q + static_cast<size_t>(b * M + m) * qstride + (kvh * G + gh) * D + g * 8;
#pragma unroll
    for (int s = 0; s < NKS; ++s) {
      if (valid) qf[s] = *reinterpret_cast<const u32x4*>(qrow + s * 32);
      else qf[s] = u32x4{0u, 0u, 0u, 0u};
    }
  }

  const uint16_t* kc = static_cast<const uint16_t*>(a.k_cache) + (static_cast<size_t>(b) * a.n_kv_heads + kvh) * a.l_max * D;
  const uint16_t* vt = static_cast<const uint16_t*>(a.v_cache) + (static_cast<size_t>(b) * a.n_kv_heads + kvh) * D * a.l_max;
  (void)kc; (void)vt;
  const int vstride = PAGED ? (1 << a.page_shift) : a.l_max;   // positions per V^T row

  f32x4_t oacc[NDT];
#pragma unroll
  for (int i = 0; i < NDT; ++i) oacc[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  float m_run = -INFINITY, l_run = 0.f;  // for query row n (replicated over g)

  const int my_limit = pos0 + my_m;  // last key this lane's query may see

  for (int blk = active ? split * kAttnWaves + wave : n_blocks; blk < n_blocks; blk += kAttnWaves * s_eff) {
    const int key0 = blk * kAttnBlock;
    int kbase = key0;                       // position of the block's first key inside its slab
    if constexpr (PAGED) {
      const int pg = __builtin_amdgcn_readfirstlane(a.block_table[b * (a.l_max >> a.page_shift) + (key0 >> a.page_shift)]);
      kc = static_cast<const uint16_t*>(a.k_cache) + ((static_cast<size_t>(pg) * a.n_kv_heads + kvh) << a.page_shift) * D;
      vt = static_cast<const uint16_t*>(a.v_cache) + ((static_cast<size_t>(pg) * a.n_kv_heads + kvh) * D << a.page_shift);
      kbase = key0 & (vstride - 1);
    }
    // ---- issue every load of the block -------------------------------------------
    // S^T tile u (u = 0,1): MFMA row i (= lane n) is key  key0 + 8*(i>>2) + 4*u + (i&3)
    u32x4 kf[2][NKS];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      int key = kbase + 8 * (n >> 2) + 4 * u + (n & 3);
      if (!PAGED && key >= a.l_max) key = a.l_max - 1;  // stay inside the cache; masked below (a page holds the whole block)
      const uint16_t* krow = kc + static_cast<size_t>(key) * D + g * 8;
#pragma unroll
      for (int s = 0; s < NKS; ++s) kf[u][s] = *reinterpret_cast<const u32x4*>(krow + s * 32);
    }
    // V^T fragment of channel tile i: lane (g, n) holds V^T[16 i + n][key0 + 8 g .. +8]
    u32x4 vf[NDT];
    {
      int kofs = kbase + 8 * g;
      if (!PAGED && kofs + 8 > a.l_max) kofs = a.l_max - 8;  // l_max % 8 == 0 (checked by the host)
      const uint16_t* vrow = vt + static_cast<size_t>(n) * vstride + kofs;
#pragma unroll
      for (int i = 0; i < NDT; ++i) vf[i] = *reinterpret_cast<const u32x4*>(vrow + static_cast<size_t>(16 * i) * vstride);
    }
    // ---- S^T = K Q^T -----------------------------------------------------------------
    f32x4_t st[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      st[u] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < NKS; ++s)
        st[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, kf[u][s]),
                                                        __builtin_bit_cast(bf16x8_t, qf[s]), st[u], 0, 0, 0);
    }
    // lane (g, n) now holds, for query row n, the scores of keys key0 + 8g + (4u + r)
    float sc[8];
    float mx = -INFINITY;
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int key = key0 + 8 * g + 4 * u + r;
        const bool vis = (key <= my_limit) && (key < n_keys);
        const float v = vis ? st[u][r] * a.scale : -INFINITY;
        sc[4 * u + r] = v;
        mx = fmaxf(mx, v);
      }
    // row max / sum across the 4 lane groups that share n
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float m_new = fmaxf(m_run, mx);
    float alpha = 1.f, psum = 0.f;
    float p[8];
    if (m_new > -INFINITY) {
      alpha = (m_run > -INFINITY) ? __expf(m_run - m_new) : 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        p[j] = (sc[j] > -INFINITY) ? __expf(sc[j] - m_new) : 0.f;
        psum += p[j];
      }
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) p[j] = 0.f;
    }
    psum += __shfl_xor(psum, 16, 64);
    psum += __shfl_xor(psum, 32, 64);
    l_run = l_run * alpha + psum;
    m_run = m_new;
    // P as the A operand: lane (g, n = q) holds P[q][key0 + 8g + j], j = 0..7
    const u32x4 pf = {pack_bf16x2(p[0], p[1]), pack_bf16x2(p[2], p[3]), pack_bf16x2(p[4], p[5]), pack_bf16x2(p[6], p[7])};
    // rescale O: lane (g, n = d) holds O[q = 4g + r][d]; alpha of row q lives in lane q
    float al[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) al[r] = __shfl(alpha, 4 * g + r, 64);
#pragma unroll
    for (int i = 0; i < NDT; ++i) {
#pragma unroll
      for (int r = 0; r < 4; ++r) oacc[i][r] *= al[r];
      oacc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, pf),
                                                        __builtin_bit_cast(bf16x8_t, vf[i]), oacc[i], 0, 0, 0);
    }
  }

  // ---- merge the 4 waves ----------------------------------------------------------------
  if (active) {
  if (g == 0) {
    m_s[wave * kAttnRows + n] = m_run;
    l_s[wave * kAttnRows + n] = l_run;
  }
#pragma unroll
  for (int i = 0; i < NDT; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) o_s[(wave * kAttnRows + 4 * g + r) * D + 16 * i + n] = oacc[i][r];
  }
  __syncthreads();
  uint16_t* out = static_cast<uint16_t*>(a.out);
  if (s_eff > 1) {
    // ---- partial of this workgroup -> workspace; the last arrival merges all s_eff partials
    const int tiles = (R + kAttnRows - 1) / kAttnRows;
    const int group = (b * a.n_kv_heads + kvh) * tiles + tile;
    constexpr int PS = kAttnRows * (D + 2);  // floats per partial: O[16][D], max[16], sum[16]
    float* mine = a.split_ws + (static_cast<size_t>(group) * a.n_split + split) * PS;
    for (int i = tid; i < kAttnRows * D; i += kAttnThreads) {
      const int r = i / D, d = i - r * D;
      float mm = -INFINITY;
#pragma unroll
      for (int w = 0; w < kAttnWaves; ++w) mm = fmaxf(mm, m_s[w * kAttnRows + r]);
      float num = 0.f, den = 0.f;
#pragma unroll
      for (int w = 0; w < kAttnWaves; ++w) {
        const float mw = m_s[w * kAttnRows + r];
        const float f = (mw > -INFINITY) ? __expf(mw - mm) : 0.f;
        num += f * o_s[(w * kAttnRows + r) * D + d];
        den += f * l_s[w * kAttnRows + r];
      }
      // the workspace is only ever touched with agent-scope RELAXED atomics (write-through stores, L2-bypassing
      // loads); the hand-over to the last arrival is ordered by thread 0's release / acquire arrival below
      __hip_atomic_store(mine + i, num, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (d == 0) {
        __hip_atomic_store(mine + kAttnRows * D + r, mm, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(mine + kAttnRows * D + kAttnRows + r, den, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    // Hand-over to the last arrival, in the HIP / HSA memory model: the workgroup barrier orders every thread's stores
    // before thread 0 (workgroup scope), thread 0's arrival on the device-scope counter is an agent-scope RELEASE
    // (cumulative over what the barrier made visible to it) and ACQUIRE (the last arrival synchronises with every
    // earlier one), and the second barrier passes that on to the other threads of the merging workgroup. ONE thread
    // per workgroup fences: an agent-scope fence is a cache-wide L2 write-back / invalidate on this part (with every
    // thread fencing 8 K / 32 K contexts took 6.34 / 8.63 ms per step; this form 5.66 / 7.09). The partial tiles
    // themselves still move with agent-scope relaxed atomics (write-through stores, L2-bypassing loads), which leaves
    // the fences nothing to write back. (Measured once and not kept: dropping the release / acquire and relying on s_waitcnt
    // + the barriers alone — gfx950 behaviour, not a memory-model guarantee — 5.6 / 6.89 ms.)
    constexpr int kArrive = __ATOMIC_ACQ_REL;
    __syncthreads();
    unsigned* flag = reinterpret_cast<unsigned*>(m_s);  // LDS scratch (m_s is dead after the loop above)
    if (tid == 0) {
      const unsigned old = __hip_atomic_fetch_add(a.split_cnt + group, 1u, kArrive, __HIP_MEMORY_SCOPE_AGENT);
      const bool last = (old == static_cast<unsigned>(s_eff - 1));
      *flag = last ? 1u : 0u;
      if (last) __hip_atomic_store(a.split_cnt + group, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // ready for the next launch
    }
    __syncthreads();
    if (*flag == 0u) return;
    const float* base = a.split_ws + static_cast<size_t>(group) * a.n_split * PS;
    // (max, sum) of every partial into LDS, then each thread owns 8 consecutive channels of one row and
    // streams its slice of the s_eff partial tiles with independent 16-byte loads
    float* pm = o_s;                           // [s_eff][16] max   (o_s is dead: the partial has been written)
    float* pl = o_s + kAttnMaxSplit * kAttnRows;  // [s_eff][16] sum
    for (int i = tid; i < s_eff * kAttnRows; i += kAttnThreads) {
      const int sidx = i / kAttnRows, r = i - sidx * kAttnRows;
      pm[i] = __hip_atomic_load(base + static_cast<size_t>(sidx) * PS + kAttnRows * D + r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      pl[i] = __hip_atomic_load(base + static_cast<size_t>(sidx) * PS + kAttnRows * D + kAttnRows + r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    constexpr int CPR = D / 8;                 // 8-channel chunks per row
    for (int c = tid; c < rows * CPR; c += kAttnThreads) {
      const int r = c / CPR, d0 = (c - r * CPR) * 8;
      float mm = -INFINITY;
      for (int sidx = 0; sidx < s_eff; ++sidx) mm = fmaxf(mm, pm[sidx * kAttnRows + r]);
      float num[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      float den = 0.f;
      const float* src = base + r * D + d0;
#pragma unroll 4
      for (int sidx = 0; sidx < s_eff; ++sidx) {
        const float4 v0 = coherent_load_f4(src + static_cast<size_t>(sidx) * PS);
        const float4 v1 = coherent_load_f4(src + static_cast<size_t>(sidx) * PS + 4);
        const float ms = pm[sidx * kAttnRows + r];
        const float f = (ms > -INFINITY) ? __expf(ms - mm) : 0.f;
        den += f * pl[sidx * kAttnRows + r];
        num[0] += f * v0.x; num[1] += f * v0.y; num[2] += f * v0.z; num[3] += f * v0.w;
        num[4] += f * v1.x; num[5] += f * v1.y; num[6] += f * v1.z; num[7] += f * v1.w;
      }
      const float inv = (den > 0.f) ? 1.0f / den : 0.f;
      const int rr = r_base + r, gh = rr / M, m = rr - gh * M;
      uint16_t* dst = out + static_cast<size_t>(b * M + m) * qstride + (kvh * G + gh) * D + d0;
      const u32x4 o = {pack_bf16x2(num[0] * inv, num[1] * inv), pack_bf16x2(num[2] * inv, num[3] * inv),
                       pack_bf16x2(num[4] * inv, num[5] * inv), pack_bf16x2(num[6] * inv, num[7] * inv)};
      *reinterpret_cast<u32x4*>(dst) = o;
    }
    return;
  }
  for (int i = active ? tid : rows * D; i < rows * D; i += kAttnThreads) {
    const int r = i / D, d = i - r * D;
    float mm = -INFINITY;
#pragma unroll
    for (int w = 0; w < kAttnWaves; ++w) mm = fmaxf(mm, m_s[w * kAttnRows + r]);
    float num = 0.f, den = 0.f;
#pragma unroll
    for (int w = 0; w < kAttnWaves; ++w) {
      const float mw = m_s[w * kAttnRows + r];
      const float f = (mw > -INFINITY) ? __expf(mw - mm) : 0.f;
      num += f * o_s[(w * kAttnRows + r) * D + d];
      den += f * l_s[w * kAttnRows + r];
    }
    const float v = (den > 0.f) ? num / den : 0.f;
    const int rr = r_base + r, gh = rr / M, m = rr - gh * M;
    out[static_cast<size_t>(b * M + m) * qstride + (kvh * G + gh) * D + d] = float_to_bf16_bits(v);
  }
}


inline size_t attention_smem_bytes(int D, int waves) {
  size_t n = sizeof(float) * (static_cast<size_t>(waves) * kAttnRows * D + 2 * waves * kAttnRows);
  const size_t merge = sizeof(float) * 2 * kAttnMaxSplit * kAttnRows;   // (max, sum) of the split-KV partials
  return n > merge ? n : merge;
}

}  // namespace sd
