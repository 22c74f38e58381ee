// Sampled bonus token on the device: temperature -> top-k -> top-p -> one categorical draw.
//
// Replaces sample_bonus_token_from_logits (src/specdec/core/pipeline.py:48-147, do_sample=True),
// which the reference runs on the host per row per step with a full sort over the vocabulary
// (:107) and torch.multinomial on the global generator. Here one 1024-thread workgroup per row:
//
//   1. a lower bound of the k-th largest logit on a 52-bit composite key
//        [32-bit order-preserving value key][20-bit inverted index]
//      (so "value descending, then index ascending" is a total order and ties at the cut are
//      decided the same way everywhere). Fast path: every thread keeps the maximum of the elements
//      it scans (16-byte loads; the row is L2/MALL resident, the lm_head epilogue has just written
//      it); the k-th largest of the 1024 thread maxima (bitonic sort in LDS) cannot exceed the k-th
//      largest element, so everything below it is out. For real logits that leaves ~k..2k
//      candidates. If more than 4096 survive (maxima concentrated in few threads), fall back to a
//      radix select: each pass histograms the next 10-11 key bits of the elements that match the
//      prefix decided so far in LDS, one wave picks the digit holding the k-th element, and the
//      passes stop as soon as that digit's bucket fits the candidate buffer;
//   2. one gather scan of everything at or above the bound, bitonic sort in LDS (<= 4096 keys);
//   3. the k survivors: x/T, exp(x/T - max), nucleus cut on the inclusive cumulative probability
//      (first token always kept), inversion of ONE Philox4x32-10 uniform through the cumulative
//      sums — in float64, sequentially in sorted order, exactly as oracle/sampling_ref.py.
//
// Without top-k and top-p the draw is a Gumbel-max over the whole row (one Philox value per
// element, block argmax): no ordered prefix sum over 128 K probabilities.
// Full-vocabulary nucleus sampling (top_p < 1 without top_k): sample_nucleus_kernel below, rank blocks of 1024.

#include "engine.h"

namespace sd {

constexpr int kSampleThreads = 1024;
constexpr int kSampleMaxK = 1024;     // top_k limit
constexpr int kSampleCap = 2048;      // bucket size at which the radix passes stop
constexpr int kSampleSort = 4096;     // >= kSampleMaxK + kSampleCap, power of two
constexpr int kIdxBits = 20;
constexpr uint32_t kTagCdf = 0x5EED0001u, kTagGumbel = 0x5EED0002u;

struct SampleArgs {
  const void* logits;
  int dtype;               // SD_F32 / SD_BF16 / SD_F16
  int64_t row_stride;      // elements between rows
  int V;
  const int32_t* pos;      // nullable: row of batch entry b = b * rows_per_b + pos[b]
  int rows_per_b;
  const int32_t* active;   // nullable: entries with active[b] == 0 are skipped (no draw consumed)
  float temperature;
  int top_k;
  float top_p;             // >= 1: no nucleus cut
  uint32_t seed_lo, seed_hi;
  uint32_t* draw;          // nullable: per-entry draw counters, read then incremented
  uint32_t draw0;          // draw index when `draw` is NULL
  const int32_t* stream_id;  // nullable: Philox stream of entry b (default b)
  int32_t* out;            // [B]
};

__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                              uint32_t k1, uint32_t& r0) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = static_cast<uint64_t>(0xD2511F53u) * c0;
    const uint64_t p1 = static_cast<uint64_t>(0xCD9E8D57u) * c2;
    const uint32_t n0 = static_cast<uint32_t>(p1 >> 32) ^ c1 ^ k0;
    const uint32_t n1 = static_cast<uint32_t>(p1);
    const uint32_t n2 = static_cast<uint32_t>(p0 >> 32) ^ c3 ^ k1;
    const uint32_t n3 = static_cast<uint32_t>(p0);
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  r0 = c0;
}

__device__ __forceinline__ float load_logit(const void* row, int dtype, int i) {
  if (dtype == SD_F32) return static_cast<const float*>(row)[i];
  if (dtype == SD_BF16) return bf16_bits_to_float(static_cast<const uint16_t*>(row)[i]);
  return __half2float(static_cast<const __half*>(row)[i]);
}

// larger key = larger value; NaN largest; -0 == +0
__device__ __forceinline__ uint32_t order_key(float x) {
  if (x != x) return 0xFFFFFFFFu;
  if (x == 0.f) x = 0.f;
  const uint32_t u = __float_as_uint(x);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

__device__ __forceinline__ uint64_t composite_key(float x, int i) {
  return (static_cast<uint64_t>(order_key(x)) << kIdxBits) | static_cast<uint64_t>(((1u << kIdxBits) - 1u) - static_cast<uint32_t>(i));
}

// f(value, index) for every element of the row, 16-byte loads when the row allows it.
// The visiting order differs between the two forms; every use below is order-independent.
template <typename F>
__device__ __forceinline__ void for_each_logit(const void* row, int dtype, int V, int tid, F&& f) {
  const bool aligned = (reinterpret_cast<uintptr_t>(row) & 15) == 0;
  if (dtype == SD_F32 && aligned && (V & 3) == 0) {
    const float4* p = static_cast<const float4*>(row);
    for (int v = tid; v < (V >> 2); v += kSampleThreads) {
      const float4 q = p[v];
      f(q.x, 4 * v); f(q.y, 4 * v + 1); f(q.z, 4 * v + 2); f(q.w, 4 * v + 3);
    }
  } else if (dtype == SD_BF16 && aligned && (V & 7) == 0) {
    const uint4* p = static_cast<const uint4*>(row);
    for (int v = tid; v < (V >> 3); v += kSampleThreads) {
      const uint4 q = p[v];
      const uint32_t w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        f(__uint_as_float(w[j] << 16), 8 * v + 2 * j);
        f(__uint_as_float(w[j] & 0xffff0000u), 8 * v + 2 * j + 1);
      }
    }
  } else {
    for (int i = tid; i < V; i += kSampleThreads) f(load_logit(row, dtype, i), i);
  }
}

__global__ __launch_bounds__(kSampleThreads) void sample_topk_kernel(const SampleArgs a) {
  __shared__ uint32_t hist[2048];
  __shared__ uint64_t sel[kSampleSort];
  __shared__ double ev[kSampleMaxK];
  __shared__ uint32_t s_cnt, s_digit, s_above, s_bucket;
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (a.active && a.active[b] == 0) return;
  const int rowi = b * a.rows_per_b + (a.pos ? a.pos[b] : 0);
  const char* row = static_cast<const char*>(a.logits) + static_cast<size_t>(rowi) * a.row_stride * (a.dtype == SD_F32 ? 4 : 2);
  const int V = a.V;
  const int k = min(a.top_k, V);

  // gather every element whose composite key, shifted right by `shift`, is >= `bound`
  auto gather = [&](uint64_t bound, int shift) {
    if (tid == 0) s_cnt = 0;
    for (int i = tid; i < kSampleSort; i += kSampleThreads) sel[i] = 0;
    __syncthreads();
    for_each_logit(row, a.dtype, V, tid, [&](float x, int i) {
      const uint64_t c = composite_key(x, i);
      if ((c >> shift) >= bound) {
        const uint32_t slot = atomicAdd(&s_cnt, 1u);
        if (slot < static_cast<uint32_t>(kSampleSort)) sel[slot] = c + 1;  // 0 stays "empty" and sorts last
      }
    });
    __syncthreads();
  };
  // descending bitonic sort of sel[0, n), n a power of two
  auto sort_desc = [&](int n) {
    for (int size = 2; size <= n; size <<= 1) {
      for (int stride = size >> 1; stride > 0; stride >>= 1) {
        for (int t = tid; t < n / 2; t += kSampleThreads) {
          const int lo = 2 * t - (t & (stride - 1));
          const int hi = lo + stride;
          const bool desc = ((lo & size) == 0);
          const uint64_t x = sel[lo], y = sel[hi];
          if ((x < y) == desc) { sel[lo] = y; sel[hi] = x; }
        }
        __syncthreads();
      }
    }
  };

  // ---- 1a. fast path: the k-th largest thread maximum bounds the k-th largest element from below
  {
    uint64_t best = 0;
    for_each_logit(row, a.dtype, V, tid, [&](float x, int i) {
      const uint64_t c = composite_key(x, i);
      best = c > best ? c : best;
    });
    sel[tid] = best;  // threads without an element hold 0: below every real key
    __syncthreads();
    sort_desc(kSampleThreads);
    const uint64_t tau = sel[k - 1];  // k <= min(1024, V): at least k threads saw an element
    __syncthreads();
    gather(tau, 0);
  }

  // ---- 1b. fallback: radix select on the composite key, most significant digits first
  if (s_cnt > static_cast<uint32_t>(kSampleSort)) {
    const int widths[5] = {11, 11, 10, 10, 10};
    uint64_t prefix = 0;       // decided high bits (right-aligned)
    int decided = 0;           // number of decided bits (of 52)
    int need = k;              // rank of the wanted element inside the current bucket (1-based from the top)
    for (int p = 0; p < 5; ++p) {
      const int w = widths[p];
      const int shift = 52 - decided - w;
      for (int i = tid; i < 2048; i += kSampleThreads) hist[i] = 0;
      __syncthreads();
      for_each_logit(row, a.dtype, V, tid, [&](float x, int i) {
        const uint64_t c = composite_key(x, i);
        if ((c >> (shift + w)) == prefix) atomicAdd(&hist[(c >> shift) & ((1u << w) - 1u)], 1u);
      });
      __syncthreads();
      if (wave == 0) {
        // lane l owns digits [32 l, 32 l + 32); scan from the top digit down
        const int nb = 1 << w;
        uint32_t local = 0;
        for (int j = 0; j < 32; ++j) {
          const int d = lane * 32 + j;
          if (d < nb) local += hist[d];
        }
        uint32_t incl = local;  // suffix sum over lanes >= this one
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
          const uint32_t o = __shfl_down(incl, off, 64);
          if (lane + off < 64) incl += o;
        }
        uint32_t above = incl - local;
        if (above < static_cast<uint32_t>(need) && incl >= static_cast<uint32_t>(need)) {
          for (int j = 31; j >= 0; --j) {
            const int d = lane * 32 + j;
            if (d >= nb) continue;
            const uint32_t h = hist[d];
            if (above + h >= static_cast<uint32_t>(need)) {
              s_digit = d;
              s_above = above;
              s_bucket = h;
              break;
            }
            above += h;
          }
        }
      }
      __syncthreads();
      prefix = (prefix << w) | s_digit;
      decided += w;
      need -= static_cast<int>(s_above);
      const uint32_t bucket = s_bucket;
      __syncthreads();
      if (bucket <= static_cast<uint32_t>(kSampleCap)) break;
    }
    gather(prefix, 52 - decided);
  }

  // ---- 2. sort the candidates (>= k of them, the k largest among them) descending
  int n_sort = 2;  // next power of two >= gathered count (workgroup-uniform)
  while (n_sort < static_cast<int>(s_cnt)) n_sort <<= 1;
  sort_desc(n_sort);

  // ---- 3. the k survivors in sorted order: weights in float64
  const double T = static_cast<double>(a.temperature);
  const bool scale = (a.temperature > 0.f) && (a.temperature != 1.0f);
  int my_idx = 0;
  if (tid < k) {
    const uint64_t c = sel[tid] - 1;
    my_idx = static_cast<int>(((1u << kIdxBits) - 1u) - static_cast<uint32_t>(c & ((1u << kIdxBits) - 1u)));
    double v = static_cast<double>(load_logit(row, a.dtype, my_idx));
    if (scale) v = v / T;
    ev[tid] = v;
  }
  __syncthreads();
  const double m = ev[0];
  __syncthreads();
  if (tid < k) {
    double e = exp(ev[tid] - m);
    if (e != e) e = 0.0;
    ev[tid] = e;
  }
  // sel[] is reused for the token ids of the survivors
  __syncthreads();
  if (tid < k) sel[tid] = static_cast<uint64_t>(my_idx);
  __syncthreads();
  if (tid == 0) {
    int pick = 0;
    const bool finite = (m == m) && (m - m == 0.0);
    if (finite) {
      int n_keep = k;
      if (a.top_p < 1.0f) {
        const double tp = static_cast<double>(a.top_p);
        double z = 0.0;
        for (int i = 0; i < k; ++i) z += ev[i];
        double cum = 0.0;
        n_keep = 0;
        for (int i = 0; i < k; ++i) {
          cum += ev[i] / z;
          if (i == 0 || !(cum > tp)) n_keep = i + 1;
          else break;
        }
      }
      double z2 = 0.0;
      for (int i = 0; i < n_keep; ++i) z2 += ev[i];
      const uint32_t d = a.draw ? a.draw[b] : a.draw0;
      const uint32_t sid = a.stream_id ? static_cast<uint32_t>(a.stream_id[b]) : static_cast<uint32_t>(b);
      uint32_t r0;
      philox4x32_10(d, sid, 0u, kTagCdf, a.seed_lo, a.seed_hi, r0);
      const double target = static_cast<double>(r0) * 2.3283064365386963e-10 * z2;  // 2^-32
      pick = n_keep - 1;
      double c = 0.0;
      for (int i = 0; i < n_keep; ++i) {
        c += ev[i];
        if (target < c) { pick = i; break; }
      }
    }
    a.out[b] = static_cast<int32_t>(sel[pick]);
    if (a.draw) a.draw[b] = a.draw[b] + 1u;
  }
}

// ------------------------------------------------------------------------------
// Full-vocabulary nucleus sampling: top_p < 1 WITHOUT top_k (sample_bonus_token_from_logits with top_k = None,
// src/specdec/core/pipeline.py:105-125: sort the whole row, cumulative softmax, drop a token when the inclusive
// cumulative probability exceeds top_p, the first is always kept). No full sort here: the row is consumed in RANK BLOCKS
// of 1024 — an exact radix select (5 passes over the L2-resident row) finds the composite key of rank 1024 (r+1), the
// elements between it and the previous block's key are gathered and sorted in LDS, and thread 0 continues the cumulative
// sum where the previous block stopped. A peaked distribution (any trained model) ends inside the first block; a flat one
// walks on, 1024 ranks at a time, up to the whole vocabulary. Arithmetic as oracle/sampling_ref.py (float64):
//   v_i = x_i / T, m = max v, e_i = exp(v_i - m),
//   Z   = sum_s ( sum_j e[s + 1024 j] )  — per-slot sums s = 0..1023 (j ascending), then the slots in order,
//   cum += e_i / Z in sorted order; kept while i == 0 or !(cum > top_p); the draw inverts one Philox uniform through
//   the kept weights in sorted order.
// ------------------------------------------------------------------------------
__global__ __launch_bounds__(kSampleThreads) void sample_nucleus_kernel(const SampleArgs a) {
  __shared__ uint32_t hist[2048];
  __shared__ uint64_t sel[kSampleThreads];
  __shared__ double ev[kSampleThreads];
  __shared__ uint32_t s_cnt, s_digit, s_above;
  __shared__ double s_z, s_cum, s_z2, s_target, s_c;
  __shared__ int s_done, s_keep, s_pick;
  __shared__ uint64_t s_best;
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (a.active && a.active[b] == 0) return;
  const int rowi = b * a.rows_per_b + (a.pos ? a.pos[b] : 0);
  const char* row = static_cast<const char*>(a.logits) + static_cast<size_t>(rowi) * a.row_stride * (a.dtype == SD_F32 ? 4 : 2);
  const int V = a.V;
  const double T = static_cast<double>(a.temperature);
  const bool scale = (a.temperature > 0.f) && (a.temperature != 1.0f);
  auto scaled = [&](int i) {
    double v = static_cast<double>(load_logit(row, a.dtype, i));
    return scale ? v / T : v;
  };
  constexpr uint32_t kIdxMask = (1u << kIdxBits) - 1u;
  auto key_index = [&](uint64_t c) { return static_cast<int>(kIdxMask - static_cast<uint32_t>(c & kIdxMask)); };

  // ---- the top element (largest composite key) and the normaliser
  if (tid == 0) s_best = 0;
  __syncthreads();
  {
    uint64_t best = 0;
    for (int i = tid; i < V; i += kSampleThreads) {
      const uint64_t c = composite_key(load_logit(row, a.dtype, i), i);
      best = c > best ? c : best;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      const uint64_t o = __shfl_xor(best, off, 64);
      best = o > best ? o : best;
    }
    if (lane == 0) atomicMax(reinterpret_cast<unsigned long long*>(&s_best), static_cast<unsigned long long>(best));
  }
  __syncthreads();
  const int top_idx = key_index(s_best);
  const double m = scaled(top_idx);
  const bool finite = (m == m) && (m - m == 0.0);
  if (!finite) {                       // -inf / NaN / +inf on top: the reference falls back to argmax (:129-131)
    if (tid == 0) {
      a.out[b] = top_idx;
      if (a.draw) a.draw[b] = a.draw[b] + 1u;
    }
    return;
  }
  {
    double p = 0.0;
    for (int i = tid; i < V; i += kSampleThreads) {
      double e = exp(scaled(i) - m);
      if (e != e) e = 0.0;
      p += e;
    }
    ev[tid] = p;
  }
  __syncthreads();
  if (tid == 0) {
    double z = 0.0;
    for (int s2 = 0; s2 < kSampleThreads; ++s2) z += ev[s2];
    s_z = z;
    s_cum = 0.0;
    s_z2 = 0.0;
    s_keep = 0;
    s_done = 0;
  }
  __syncthreads();

  // exact composite key of rank k (1-based from the top): radix select over all 52 bits
  auto select_kth = [&](int k) -> uint64_t {
    const int widths[5] = {11, 11, 10, 10, 10};
    uint64_t prefix = 0;
    int decided = 0, need = k;
    for (int p = 0; p < 5; ++p) {
      const int w = widths[p];
      const int shift = 52 - decided - w;
      for (int i = tid; i < 2048; i += kSampleThreads) hist[i] = 0;
      __syncthreads();
      for (int i = tid; i < V; i += kSampleThreads) {
        const uint64_t c = composite_key(load_logit(row, a.dtype, i), i);
        if ((c >> (shift + w)) == prefix) atomicAdd(&hist[(c >> shift) & ((1u << w) - 1u)], 1u);
      }
      __syncthreads();
      if (wave == 0) {
        const int nb = 1 << w;
        uint32_t local = 0;
        for (int j = 0; j < 32; ++j) {
          const int d = lane * 32 + j;
          if (d < nb) local += hist[d];
        }
        uint32_t incl = local;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
          const uint32_t o = __shfl_down(incl, off, 64);
          if (lane + off < 64) incl += o;
        }
        uint32_t above = incl - local;
        if (above < static_cast<uint32_t>(need) && incl >= static_cast<uint32_t>(need)) {
          for (int j = 31; j >= 0; --j) {
            const int d = lane * 32 + j;
            if (d >= nb) continue;
            const uint32_t h = hist[d];
            if (above + h >= static_cast<uint32_t>(need)) {
              s_digit = d;
              s_above = above;
              break;
            }
            above += h;
          }
        }
      }
      __syncthreads();
      prefix = (prefix << w) | s_digit;
      decided += w;
      need -= static_cast<int>(s_above);
      __syncthreads();
    }
    return prefix;
  };
  // block r: the elements of ranks (1024 r, min(1024 (r+1), V)], sorted descending into sel[] / ev[]; returns their count
  auto load_block = [&](int r, uint64_t upper_excl, uint64_t& thr_out) -> int {
    const int k_hi = min((r + 1) * kSampleThreads, V);
    const uint64_t thr = select_kth(k_hi);
    thr_out = thr;
    if (tid == 0) s_cnt = 0;
    sel[tid] = 0;
    __syncthreads();
    for (int i = tid; i < V; i += kSampleThreads) {
      const uint64_t c = composite_key(load_logit(row, a.dtype, i), i);
      if (c >= thr && c < upper_excl) {
        const uint32_t slot = atomicAdd(&s_cnt, 1u);
        if (slot < static_cast<uint32_t>(kSampleThreads)) sel[slot] = c + 1;
      }
    }
    __syncthreads();
    for (int size = 2; size <= kSampleThreads; size <<= 1) {        // descending bitonic sort of the 1024 slots (0 = empty, last)
      for (int stride = size >> 1; stride > 0; stride >>= 1) {
        if (tid < kSampleThreads / 2) {
          const int lo = 2 * tid - (tid & (stride - 1));
          const int hi = lo + stride;
          const bool desc = ((lo & size) == 0);
          const uint64_t x = sel[lo], y = sel[hi];
          if ((x < y) == desc) { sel[lo] = y; sel[hi] = x; }
        }
        __syncthreads();
      }
    }
    const int n = k_hi - r * kSampleThreads;
    if (tid < n) {
      const int idx = key_index(sel[tid] - 1);
      double e = exp(scaled(idx) - m);
      if (e != e) e = 0.0;
      ev[tid] = e;
      sel[tid] = static_cast<uint64_t>(idx);
    }
    __syncthreads();
    return n;
  };

  // ---- phase A: how many tokens the nucleus keeps, and their total weight
  const double tp = static_cast<double>(a.top_p);
  const int n_blocks = (V + kSampleThreads - 1) / kSampleThreads;
  uint64_t upper = ~0ull;
  int blocks_used = 0;
  for (int r = 0; r < n_blocks; ++r) {
    uint64_t thr;
    const int n = load_block(r, upper, thr);
    upper = thr;
    blocks_used = r + 1;
    if (tid == 0) {
      double cum = s_cum, z2 = s_z2;
      int keep = s_keep, done = 0;
      for (int i = 0; i < n; ++i) {
        cum += ev[i] / s_z;
        if ((r == 0 && i == 0) || !(cum > tp)) {
          ++keep;
          z2 += ev[i];
        } else {
          done = 1;
          break;
        }
      }
      s_cum = cum;
      s_z2 = z2;
      s_keep = keep;
      s_done = done;
    }
    __syncthreads();
    if (s_done) break;
  }
  // ---- phase B: invert one uniform through the kept weights, in sorted order
  if (tid == 0) {
    const uint32_t d = a.draw ? a.draw[b] : a.draw0;
    const uint32_t sid = a.stream_id ? static_cast<uint32_t>(a.stream_id[b]) : static_cast<uint32_t>(b);
    uint32_t r0;
    philox4x32_10(d, sid, 0u, kTagCdf, a.seed_lo, a.seed_hi, r0);
    s_target = static_cast<double>(r0) * 2.3283064365386963e-10 * s_z2;  // 2^-32
    s_c = 0.0;
    s_pick = -1;
  }
  __syncthreads();
  const int n_keep = s_keep;
  if (blocks_used == 1) {              // the common case: block 0 is still in LDS
    if (tid == 0) {
      int pick = n_keep - 1;
      double c = 0.0;
      for (int i = 0; i < n_keep; ++i) {
        c += ev[i];
        if (s_target < c) { pick = i; break; }
      }
      s_pick = static_cast<int>(sel[pick]);
    }
  } else {
    upper = ~0ull;
    int last_tok = top_idx;
    for (int r = 0; r * kSampleThreads < n_keep; ++r) {
      uint64_t thr;
      const int n = load_block(r, upper, thr);
      upper = thr;
      const int lim = min(n, n_keep - r * kSampleThreads);
      if (tid == 0) {
        double c = s_c;
        for (int i = 0; i < lim; ++i) {
          c += ev[i];
          if (s_target < c) { s_pick = static_cast<int>(sel[i]); break; }
        }
        s_c = c;
      }
      last_tok = static_cast<int>(sel[lim - 1]);   // (every thread reads the same LDS word)
      __syncthreads();
      if (s_pick >= 0) break;
    }
    if (tid == 0 && s_pick < 0) s_pick = last_tok;  // target == total weight (rounding): the last kept token
  }
  __syncthreads();
  if (tid == 0) {
    a.out[b] = s_pick;
    if (a.draw) a.draw[b] = a.draw[b] + 1u;
  }
}

__device__ __forceinline__ bool better_d(double v, int i, double bv, int bi) {
  const bool vn = (v != v), bn = (bv != bv);
  if (vn | bn) {
    if (vn & bn) return i < bi;
    return vn;
  }
  return (v > bv) | ((v == bv) & (i < bi));
}

__global__ __launch_bounds__(kSampleThreads) void sample_gumbel_kernel(const SampleArgs a) {
  __shared__ double sv[kSampleThreads / kWave];
  __shared__ int si[kSampleThreads / kWave];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (a.active && a.active[b] == 0) return;
  const int rowi = b * a.rows_per_b + (a.pos ? a.pos[b] : 0);
  const char* row = static_cast<const char*>(a.logits) + static_cast<size_t>(rowi) * a.row_stride * (a.dtype == SD_F32 ? 4 : 2);
  const double T = static_cast<double>(a.temperature);
  const bool scale = (a.temperature > 0.f) && (a.temperature != 1.0f);
  const uint32_t d = a.draw ? a.draw[b] : a.draw0;
  const uint32_t sid = a.stream_id ? static_cast<uint32_t>(a.stream_id[b]) : static_cast<uint32_t>(b);
  double bv = -INFINITY;
  int bi = 0x7fffffff;
  for (int i = tid; i < a.V; i += kSampleThreads) {
    double v = static_cast<double>(load_logit(row, a.dtype, i));
    if (scale) v = v / T;
    uint32_t r0;
    philox4x32_10(d, sid, static_cast<uint32_t>(i), kTagGumbel, a.seed_lo, a.seed_hi, r0);
    const double u = (static_cast<double>(r0) + 0.5) * 2.3283064365386963e-10;
    const double s = v + (-log(-log(u)));
    if (better_d(s, i, bv, bi)) { bv = s; bi = i; }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const double ov = __shfl_xor(bv, off, 64);
    const int oi = __shfl_xor(bi, off, 64);
    if (better_d(ov, oi, bv, bi)) { bv = ov; bi = oi; }
  }
  if (lane == 0) { sv[wave] = bv; si[wave] = bi; }
  __syncthreads();
  if (tid == 0) {
    for (int w = 1; w < kSampleThreads / kWave; ++w)
      if (better_d(sv[w], si[w], bv, bi)) { bv = sv[w]; bi = si[w]; }
    a.out[b] = bi;
    if (a.draw) a.draw[b] = a.draw[b] + 1u;
  }
}

int launch_sample(const SampleArgs& a, int B, hipStream_t st) {
  SD_REQUIRE(a.logits && a.out, "sample: NULL logits/out");
  SD_REQUIRE(a.dtype == SD_F32 || a.dtype == SD_BF16 || a.dtype == SD_F16, "sample: logits dtype %d", a.dtype);
  SD_REQUIRE(B >= 1 && B <= 65535 && a.V >= 1 && a.V <= (1 << kIdxBits), "sample: B=%d V=%d out of range", B, a.V);
  SD_REQUIRE(a.rows_per_b >= 1, "sample: rows_per_b=%d", a.rows_per_b);
  SD_REQUIRE(a.temperature == a.temperature, "sample: temperature is NaN");
  if (a.top_k > 0) {
    SD_REQUIRE((a.top_k < a.V ? a.top_k : a.V) <= kSampleMaxK, "sample: top_k=%d > %d is not supported", a.top_k, kSampleMaxK);
    hipLaunchKernelGGL(sample_topk_kernel, dim3(B), dim3(kSampleThreads), 0, st, a);
  } else if (a.top_p < 1.0f) {
    hipLaunchKernelGGL(sample_nucleus_kernel, dim3(B), dim3(kSampleThreads), 0, st, a);
  } else {
    hipLaunchKernelGGL(sample_gumbel_kernel, dim3(B), dim3(kSampleThreads), 0, st, a);
  }
  SD_LAUNCH_CHECK();
  return 0;
}

// the step's draw: entry b samples from row b*(K+1) + accept_len[b] of the verify logits
int launch_sample_step(const SpecState& s, const void* logits, int V, float temperature, int top_k, float top_p,
                       uint64_t seed, uint32_t* draw, const int32_t* stream_id, hipStream_t st) {
  SampleArgs a{};
  a.logits = logits;
  a.dtype = SD_BF16;
  a.row_stride = V;
  a.V = V;
  a.pos = s.accept_len;
  a.rows_per_b = s.K + 1;
  a.active = s.active;
  a.temperature = temperature;
  a.top_k = top_k;
  a.top_p = top_p;
  a.seed_lo = static_cast<uint32_t>(seed);
  a.seed_hi = static_cast<uint32_t>(seed >> 32);
  a.draw = draw;
  a.stream_id = stream_id;
  a.out = s.sampled;
  return launch_sample(a, s.B, st);
}

}  // namespace sd

using namespace sd;

extern "C" int sd_sample_token(const void* logits, int logits_dtype, int64_t row_stride, int B, int V,
                               const int32_t* pos, int rows_per_b, const int32_t* active, float temperature,
                               int top_k, float top_p, uint64_t seed, uint32_t* draw_counters, uint32_t draw0,
                               const int32_t* stream_id, int32_t* out_ids, void* stream) {
  clear_error();
  SampleArgs a{};
  a.logits = logits;
  a.dtype = logits_dtype;
  a.row_stride = row_stride;
  a.V = V;
  a.pos = pos;
  a.rows_per_b = rows_per_b;
  a.active = active;
  a.temperature = temperature;
  a.top_k = top_k;
  a.top_p = top_p;
  a.seed_lo = static_cast<uint32_t>(seed);
  a.seed_hi = static_cast<uint32_t>(seed >> 32);
  a.draw = draw_counters;
  a.draw0 = draw0;
  a.stream_id = stream_id;
  a.out = out_ids;
  return launch_sample(a, B, static_cast<hipStream_t>(stream));
}
