// verify_prefix for gfx950: argmax over the vocabulary + longest-accepted-prefix.
//
// Contract: verify_prefix_ref, /root/reference/src/kernels/reference.py:13-56
// (argmax(-1) → equality with the draft ids → longest matching prefix, prefix-only
// mask). Written from that contract; it is not a translation of verify.cu, which
// walks K rows serially inside one block per batch row and keeps half the block
// idle behind a 128-float tile.
//
// Shape of the work: B*K independent rows of V logits (V = 128256 for Llama-3.2,
// 256 KB per bf16 row) — a pure HBM/L2 stream with a tiny reduction behind it.
//   pass 1  grid (nsplit, B*K): each workgroup streams one 16-byte-aligned chunk of
//           one row with dwordx4 loads (8 bf16 / 4 f32 per lane), keeps a per-lane
//           running (max, index), reduces over the wave with shuffles and over the
//           4 waves through LDS, and writes one (value, index) partial.
//   pass 2  grid B, one wave per batch row: lane k folds the nsplit partials of
//           position k (independent loads: one round trip), then holds match[k];
//           __ballot + ctz gives the longest accepted prefix without a serial scan.
// Ties resolve to the lowest index and NaN is the maximum, as torch.argmax does.

#include "common.h"

namespace sd {

constexpr int kVerifyThreads = 256;
constexpr int kVerifyMaxSplit = 64;

template <typename T>
struct VecOf;
template <>
struct VecOf<float> {
  static constexpr int N = 4;
  __device__ static float get(const uint4& v, int j) {
    const uint32_t w = (j == 0) ? v.x : (j == 1) ? v.y : (j == 2) ? v.z : v.w;
    return __uint_as_float(w);
  }
  __device__ static float scalar(const void* p, int64_t i) {
    return static_cast<const float*>(p)[i];
  }
};
template <>
struct VecOf<__half> {
  static constexpr int N = 8;
  __device__ static float get(const uint4& v, int j) {
    const uint32_t w = (j < 2) ? v.x : (j < 4) ? v.y : (j < 6) ? v.z : v.w;
    const uint16_t h = (j & 1) ? (w >> 16) : (w & 0xffffu);
    return __half2float(__ushort_as_half(h));
  }
  __device__ static float scalar(const void* p, int64_t i) {
    return __half2float(static_cast<const __half*>(p)[i]);
  }
};
template <>
struct VecOf<__hip_bfloat16> {
  static constexpr int N = 8;
  __device__ static float get(const uint4& v, int j) {
    const uint32_t w = (j < 2) ? v.x : (j < 4) ? v.y : (j < 6) ? v.z : v.w;
    return (j & 1) ? __uint_as_float(w & 0xffff0000u) : __uint_as_float(w << 16);
  }
  __device__ static float scalar(const void* p, int64_t i) {
    return bf16_bits_to_float(static_cast<const uint16_t*>(p)[i]);
  }
};

// pass 1: partial argmax of elements [c0, c1) of row (b,k)
template <typename T>
__global__ __launch_bounds__(kVerifyThreads) void verify_partial_kernel(
    const T* __restrict__ logits, int K, int V, int64_t stride_b, int64_t stride_k,
    int chunk, float* __restrict__ ws_val, int* __restrict__ ws_idx) {
  using VT = VecOf<T>;
  const int row = blockIdx.y;
  const int split = blockIdx.x;
  const int nsplit = gridDim.x;
  const int b = row / K, k = row - b * K;
  const T* rowp = logits + b * stride_b + k * stride_k;

  const int c0 = split * chunk;
  const int c1 = min(V, c0 + chunk);

  float bv = -INFINITY;
  int bi = 0x7fffffff;

  if (c0 < c1) {
    // scalar head up to the first 16-byte boundary, vector body, scalar tail
    const uintptr_t addr = reinterpret_cast<uintptr_t>(rowp + c0);
    int head = static_cast<int>(((16 - (addr & 15)) & 15) / sizeof(T));
    head = min(head, c1 - c0);
    const int body0 = c0 + head;
    const int nvec = (c1 - body0) / VT::N;
    const int tail0 = body0 + nvec * VT::N;

    if (static_cast<int>(threadIdx.x) < head) {
      const int i = c0 + threadIdx.x;
      const float v = VT::scalar(rowp, i);
      if (argmax_better(v, i, bv, bi)) { bv = v; bi = i; }
    }
    const uint4* vp = reinterpret_cast<const uint4*>(rowp + body0);
    // four independent 16-byte loads in flight per lane per trip (16 KiB per workgroup): a chunk of the default
    // size is ONE memory round trip; the compares run after all four have been issued
    constexpr int kInFlight = 4;
    for (int j0 = 0; j0 < nvec; j0 += kInFlight * kVerifyThreads) {
      uint4 q[kInFlight];
#pragma unroll
      for (int u = 0; u < kInFlight; ++u) {
        const int j = j0 + u * kVerifyThreads + static_cast<int>(threadIdx.x);
        q[u] = (j < nvec) ? vp[j] : make_uint4(0u, 0u, 0u, 0u);
      }
#pragma unroll
      for (int u = 0; u < kInFlight; ++u) {
        const int j = j0 + u * kVerifyThreads + static_cast<int>(threadIdx.x);
        if (j < nvec) {
          const int ia = body0 + j * VT::N;
#pragma unroll
          for (int e = 0; e < VT::N; ++e) {
            const float v = VT::get(q[u], e);
            if (argmax_better(v, ia + e, bv, bi)) { bv = v; bi = ia + e; }
          }
        }
      }
    }
    {
      const int i = tail0 + threadIdx.x;
      if (i < c1) {
        const float v = VT::scalar(rowp, i);
        if (argmax_better(v, i, bv, bi)) { bv = v; bi = i; }
      }
    }
  }

  wave_reduce_argmax(bv, bi);
  __shared__ float s_v[kVerifyThreads / kWave];
  __shared__ int s_i[kVerifyThreads / kWave];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) { s_v[wave] = bv; s_i[wave] = bi; }
  __syncthreads();
  if (threadIdx.x == 0) {
#pragma unroll
    for (int w = 1; w < kVerifyThreads / kWave; ++w) {
      if (argmax_better(s_v[w], s_i[w], bv, bi)) { bv = s_v[w]; bi = s_i[w]; }
    }
    ws_val[row * nsplit + split] = bv;
    ws_idx[row * nsplit + split] = bi;
  }
}

// pass 2: one wave per batch row. Lane k folds the nsplit partials of position k itself — the loads of a lane are
// independent of each other and of the other lanes', so the whole fold is one memory round trip (a wave-wide reduce
// per position, one after the other, cost a dependent L2 miss per k: +1 us per draft token) — then holds
// match[k]; __ballot + ctz gives the longest accepted prefix without a serial scan.
template <typename IdT>
__global__ __launch_bounds__(kWave) void verify_finalize_kernel(
    const float* __restrict__ ws_val, const int* __restrict__ ws_idx, int nsplit,
    const IdT* __restrict__ ids, int K, int32_t* __restrict__ accept_len,
    uint8_t* __restrict__ mask, int32_t* __restrict__ pred_out) {
  const int b = blockIdx.x;
  const int lane = threadIdx.x;
  int accepted = 0;
  bool open = true;  // prefix still unbroken (wave-uniform)
  for (int k0 = 0; k0 < K; k0 += kWave) {
    const int kn = min(kWave, K - k0);
    int my_pred = -1;
    bool match = false;
    if (lane < kn) {
      const size_t row = static_cast<size_t>(b) * K + k0 + lane;
      const long long want = static_cast<long long>(ids[row]);
      float v = -INFINITY;
      int i = 0x7fffffff;
      for (int s0 = 0; s0 < nsplit; s0 += 8) {       // 8 partials (16 loads) in flight
        float sv[8];
        int si[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int s = (s0 + u < nsplit) ? s0 + u : nsplit - 1;
          sv[u] = ws_val[row * nsplit + s];
          si[u] = ws_idx[row * nsplit + s];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
          if (s0 + u < nsplit && argmax_better(sv[u], si[u], v, i)) { v = sv[u]; i = si[u]; }
      }
      my_pred = i;
      match = (static_cast<long long>(my_pred) == want);
      if (pred_out) pred_out[row] = my_pred;
    }
    const unsigned long long m = __ballot(match);
    const unsigned long long valid = (kn == 64) ? ~0ull : ((1ull << kn) - 1ull);
    const unsigned long long miss = (~m) & valid;
    const int run = open ? (miss ? __builtin_ctzll(miss) : kn) : 0;
    if (lane < kn) mask[static_cast<size_t>(b) * K + k0 + lane] = (lane < run) ? 1 : 0;
    accepted += run;
    if (run < kn) open = false;
  }
  if (lane == 0) accept_len[b] = accepted;
}

static int pick_nsplit(int B, int K, int V, int esize) {
  const int64_t row_bytes = static_cast<int64_t>(V) * esize;
  int nsplit = static_cast<int>((row_bytes + 16383) / 16384);  // ~16 KB per workgroup
  if (nsplit < 1) nsplit = 1;
  if (nsplit > kVerifyMaxSplit) nsplit = kVerifyMaxSplit;
  const int64_t rows = static_cast<int64_t>(B) * K;
  while (nsplit > 1 && rows * nsplit > 8192) nsplit >>= 1;
  return nsplit;
}

}  // namespace sd

extern "C" size_t sd_verify_prefix_workspace(int B, int K, int V) {
  (void)V;
  if (B <= 0 || K <= 0) return 0;
  return static_cast<size_t>(B) * K * sd::kVerifyMaxSplit * 8 + 256;
}

extern "C" int sd_verify_prefix(const void* logits, int logits_dtype, const void* ids,
                                int ids_dtype, int32_t* accept_len, uint8_t* mask,
                                int32_t* pred_out, int B, int K, int V, int64_t stride_b,
                                int64_t stride_k, void* workspace, size_t workspace_bytes,
                                void* stream) {
  using namespace sd;
  clear_error();
  SD_REQUIRE(B >= 0 && K >= 0 && V >= 0, "verify_prefix: negative dimension");
  if (B == 0) return 0;
  SD_REQUIRE(accept_len != nullptr, "verify_prefix: accept_len is NULL");
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (K == 0) {  // nothing proposed: accept_len = 0, empty mask
    SD_HIP_CHECK(hipMemsetAsync(accept_len, 0, sizeof(int32_t) * B, st));
    return 0;
  }
  SD_REQUIRE(V > 0, "verify_prefix: empty vocabulary (V=0)");
  SD_REQUIRE(logits && ids && mask, "verify_prefix: NULL pointer");
  SD_REQUIRE(logits_dtype == SD_F32 || logits_dtype == SD_F16 || logits_dtype == SD_BF16,
             "verify_prefix: logits dtype %d not supported (f32/f16/bf16)", logits_dtype);
  SD_REQUIRE(ids_dtype == SD_I32 || ids_dtype == SD_I64,
             "verify_prefix: ids dtype %d not supported (i32/i64)", ids_dtype);
  SD_REQUIRE(static_cast<int64_t>(B) * K <= 65535, "verify_prefix: B*K=%lld exceeds grid.y limit",
             static_cast<long long>(B) * K);
  const size_t need = sd_verify_prefix_workspace(B, K, V);
  SD_REQUIRE(workspace && workspace_bytes >= need,
             "verify_prefix: workspace too small (%zu < %zu)", workspace_bytes, need);

  const int esize = dtype_size(logits_dtype);
  const int nsplit = pick_nsplit(B, K, V, esize);
  const int vecn = 16 / esize;
  int chunk = (V + nsplit - 1) / nsplit;
  chunk = ((chunk + vecn - 1) / vecn) * vecn;

  // 256-byte aligned carve of the workspace: values then indices
  uintptr_t base = (reinterpret_cast<uintptr_t>(workspace) + 255) & ~static_cast<uintptr_t>(255);
  float* ws_val = reinterpret_cast<float*>(base);
  int* ws_idx = reinterpret_cast<int*>(ws_val + static_cast<size_t>(B) * K * kVerifyMaxSplit);

  dim3 grid(nsplit, B * K), block(kVerifyThreads);
  switch (logits_dtype) {
    case SD_F32:
      hipLaunchKernelGGL(verify_partial_kernel<float>, grid, block, 0, st,
                         static_cast<const float*>(logits), K, V, stride_b, stride_k, chunk,
                         ws_val, ws_idx);
      break;
    case SD_F16:
      hipLaunchKernelGGL(verify_partial_kernel<__half>, grid, block, 0, st,
                         static_cast<const __half*>(logits), K, V, stride_b, stride_k, chunk,
                         ws_val, ws_idx);
      break;
    default:
      hipLaunchKernelGGL(verify_partial_kernel<__hip_bfloat16>, grid, block, 0, st,
                         static_cast<const __hip_bfloat16*>(logits), K, V, stride_b, stride_k,
                         chunk, ws_val, ws_idx);
      break;
  }
  SD_LAUNCH_CHECK();
  if (ids_dtype == SD_I64) {
    hipLaunchKernelGGL(verify_finalize_kernel<int64_t>, dim3(B), dim3(kWave), 0, st, ws_val,
                       ws_idx, nsplit, static_cast<const int64_t*>(ids), K, accept_len, mask,
                       pred_out);
  } else {
    hipLaunchKernelGGL(verify_finalize_kernel<int32_t>, dim3(B), dim3(kWave), 0, st, ws_val,
                       ws_idx, nsplit, static_cast<const int32_t*>(ids), K, accept_len, mask,
                       pred_out);
  }
  SD_LAUNCH_CHECK();
  return 0;
}
