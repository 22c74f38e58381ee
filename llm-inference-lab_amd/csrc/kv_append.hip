// KV-append path for gfx950: in-place append, out-of-place concat, masked compaction.
//
// Contracts: kv_append_ref / kv_append_with_mask_ref,
// /root/reference/src/kernels/reference.py:59-93, 96-159. Written from those
// contracts. All three are pure byte movement (HBM-bound), so they are written
// type-agnostically over "units" of 16, 4 or 2 bytes: one unit per lane per trip,
// consecutive lanes on consecutive units of a row so that a wave moves 1 KiB of
// contiguous bytes per dwordx4 instruction. K and V travel in ONE launch
// (blockIdx.y selects the tensor) — at decode sizes the launch boundary costs more
// than the bytes.

#include "common.h"

namespace sd {

constexpr int kCopyThreads = 256;

struct KvPair {
  void* dst[2];
  const void* a[2];  // base (or unused)
  const void* b[2];  // new / draft
};

// ---- in-place append: cache[b,h,off_b + r,:] = new[b,h,r,:] ---------------------
template <typename U>
__global__ __launch_bounds__(kCopyThreads) void kv_append_inplace_kernel(
    KvPair p, const int32_t* __restrict__ row_len, int L, int H, int Lmax, int K, int row_units,
    int64_t total_units) {
  U* __restrict__ dst = static_cast<U*>(p.dst[blockIdx.y]);
  const U* __restrict__ src = static_cast<const U*>(p.b[blockIdx.y]);
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * kCopyThreads + threadIdx.x; i < total_units;
       i += static_cast<int64_t>(gridDim.x) * kCopyThreads) {
    const int u = static_cast<int>(i % row_units);
    const int64_t t = i / row_units;
    const int r = static_cast<int>(t % K);
    const int64_t bh = t / K;
    const int b = static_cast<int>(bh / H);
    const int off = row_len ? row_len[b] : L;
    if (off < 0 || off + r >= Lmax) continue;  // never write outside the cache
    dst[(bh * Lmax + off + r) * row_units + u] = src[i];
  }
}

// ---- out-of-place concat: out = cat(base[:, :, :L], new) along rows -------------
template <typename U>
__global__ __launch_bounds__(kCopyThreads) void kv_concat_kernel(
    KvPair p, int H, int L, int K, int out_cap, int row_units, int64_t base_sb_units,
    int64_t base_sh_units, int64_t total_units) {
  U* __restrict__ dst = static_cast<U*>(p.dst[blockIdx.y]);
  const U* __restrict__ base = static_cast<const U*>(p.a[blockIdx.y]);
  const U* __restrict__ nw = static_cast<const U*>(p.b[blockIdx.y]);
  const int R = L + K;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * kCopyThreads + threadIdx.x; i < total_units;
       i += static_cast<int64_t>(gridDim.x) * kCopyThreads) {
    const int u = static_cast<int>(i % row_units);
    const int64_t t = i / row_units;
    const int r = static_cast<int>(t % R);
    const int64_t bh = t / R;
    const int b = static_cast<int>(bh / H), h = static_cast<int>(bh - static_cast<int64_t>(b) * H);
    U v;
    if (r < L) {
      v = base[b * base_sb_units + h * base_sh_units + static_cast<int64_t>(r) * row_units + u];
    } else {
      v = nw[(bh * K + (r - L)) * row_units + u];
    }
    dst[(bh * out_cap + r) * row_units + u] = v;
  }
}

// ---- masked compaction ------------------------------------------------------------
template <typename U>
__device__ __forceinline__ U zero_unit();
template <>
__device__ __forceinline__ uint4 zero_unit<uint4>() { return make_uint4(0, 0, 0, 0); }
template <>
__device__ __forceinline__ uint32_t zero_unit<uint32_t>() { return 0u; }
template <>
__device__ __forceinline__ uint16_t zero_unit<uint16_t>() { return 0; }

template <typename U>
__global__ __launch_bounds__(kCopyThreads) void kv_append_masked_kernel(
    KvPair p, const uint8_t* __restrict__ mask, const int32_t* __restrict__ accept_len, int H,
    int L, int K, int row_units, int64_t base_sb_units, int64_t base_sh_units,
    int64_t total_units) {
  U* __restrict__ dst = static_cast<U*>(p.dst[blockIdx.y]);
  const U* __restrict__ base = static_cast<const U*>(p.a[blockIdx.y]);
  const U* __restrict__ draft = static_cast<const U*>(p.b[blockIdx.y]);
  const int R = L + K;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * kCopyThreads + threadIdx.x; i < total_units;
       i += static_cast<int64_t>(gridDim.x) * kCopyThreads) {
    const int u = static_cast<int>(i % row_units);
    const int64_t t = i / row_units;
    const int r = static_cast<int>(t % R);
    const int64_t bh = t / R;
    const int b = static_cast<int>(bh / H), h = static_cast<int>(bh - static_cast<int64_t>(b) * H);
    U v = zero_unit<U>();
    if (r < L) {
      v = base[b * base_sb_units + h * base_sh_units + static_cast<int64_t>(r) * row_units + u];
    } else {
      // slot j of the appended region takes the draft row of the j-th set mask bit,
      // for j < accept_len[b] (reference.py:146-157)
      const int j = r - L;
      // (a NEGATIVE accept_len still writes the row of the first set bit there: the loop tests
      // `accepted_count >= num_accepted` only after a write, reference.py:148-157)
      int want = accept_len[b];
      if (want < 0) want = 1;
      if (j < want) {
        int seen = 0, srck = -1;
        for (int k = 0; k < K; ++k) {
          if (mask[b * K + k]) {
            if (seen == j) { srck = k; break; }
            ++seen;
          }
        }
        if (srck >= 0) v = draft[(bh * K + srck) * row_units + u];
      }
    }
    dst[i] = v;
  }
}

static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

static inline int pick_unit(int row_bytes, int elem_size, std::initializer_list<const void*> ptrs,
                            std::initializer_list<int64_t> byte_strides) {
  bool ok16 = (row_bytes % 16 == 0);
  for (const void* p : ptrs) ok16 = ok16 && aligned16(p);
  for (int64_t s : byte_strides) ok16 = ok16 && (s % 16 == 0);
  if (ok16) return 16;
  return (elem_size % 4 == 0) ? 4 : 2;
}

static inline int grid_for(int64_t total_units) {
  int64_t g = (total_units + kCopyThreads - 1) / kCopyThreads;
  if (g < 1) g = 1;
  if (g > 2048) g = 2048;  // 256 CUs x 8 resident blocks; grid-stride the rest
  return static_cast<int>(g);
}

}  // namespace sd

extern "C" int sd_kv_append(void* cache_k, void* cache_v, const void* new_k, const void* new_v,
                            const int32_t* row_len, int L, int elem_size, int B, int H, int Lmax,
                            int K, int D, void* stream) {
  using namespace sd;
  clear_error();
  SD_REQUIRE(elem_size == 2 || elem_size == 4, "kv_append: elem_size %d not in {2,4}", elem_size);
  SD_REQUIRE(B >= 0 && H >= 0 && K >= 0 && D >= 0 && Lmax >= 0, "kv_append: negative dimension");
  if (B == 0 || H == 0 || K == 0 || D == 0) return 0;
  SD_REQUIRE(cache_k && cache_v && new_k && new_v, "kv_append: NULL pointer");
  if (!row_len) SD_REQUIRE(L >= 0 && L + K <= Lmax, "kv_append: L=%d + K=%d exceeds Lmax=%d", L, K, Lmax);
  const int row_bytes = D * elem_size;
  const int unit = pick_unit(row_bytes, elem_size, {cache_k, cache_v, new_k, new_v}, {});
  const int row_units = row_bytes / unit;
  const int64_t total = static_cast<int64_t>(B) * H * K * row_units;
  KvPair p{{cache_k, cache_v}, {nullptr, nullptr}, {new_k, new_v}};
  dim3 grid(grid_for(total), 2), block(kCopyThreads);
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (unit == 16)
    hipLaunchKernelGGL(kv_append_inplace_kernel<uint4>, grid, block, 0, st, p, row_len, L, H, Lmax, K, row_units, total);
  else if (unit == 4)
    hipLaunchKernelGGL(kv_append_inplace_kernel<uint32_t>, grid, block, 0, st, p, row_len, L, H, Lmax, K, row_units, total);
  else
    hipLaunchKernelGGL(kv_append_inplace_kernel<uint16_t>, grid, block, 0, st, p, row_len, L, H, Lmax, K, row_units, total);
  SD_LAUNCH_CHECK();
  return 0;
}

extern "C" int sd_kv_concat(void* out_k, void* out_v, const void* base_k, const void* base_v,
                            const void* new_k, const void* new_v, int elem_size, int B, int H,
                            int L, int K, int D, int out_cap, int64_t base_sb, int64_t base_sh,
                            void* stream) {
  using namespace sd;
  clear_error();
  SD_REQUIRE(out_cap >= L + K, "kv_concat: out_cap=%d < L+K=%d", out_cap, L + K);
  SD_REQUIRE(elem_size == 2 || elem_size == 4, "kv_concat: elem_size %d not in {2,4}", elem_size);
  SD_REQUIRE(B >= 0 && H >= 0 && K >= 0 && D >= 0 && L >= 0, "kv_concat: negative dimension");
  if (B == 0 || H == 0 || D == 0 || L + K == 0) return 0;
  SD_REQUIRE(out_k && out_v, "kv_concat: NULL output");
  SD_REQUIRE(L == 0 || (base_k && base_v), "kv_concat: NULL base with L>0");
  SD_REQUIRE(K == 0 || (new_k && new_v), "kv_concat: NULL new with K>0");
  const int row_bytes = D * elem_size;
  const int unit = pick_unit(row_bytes, elem_size,
                             {out_k, out_v, L ? base_k : out_k, L ? base_v : out_v,
                              K ? new_k : out_k, K ? new_v : out_v},
                             {base_sb * elem_size, base_sh * elem_size});
  const int row_units = row_bytes / unit;
  const int64_t total = static_cast<int64_t>(B) * H * (L + K) * row_units;
  KvPair p{{out_k, out_v}, {base_k, base_v}, {new_k, new_v}};
  dim3 grid(grid_for(total), 2), block(kCopyThreads);
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int64_t sbu = base_sb * elem_size / unit, shu = base_sh * elem_size / unit;
  if (unit == 16)
    hipLaunchKernelGGL(kv_concat_kernel<uint4>, grid, block, 0, st, p, H, L, K, out_cap, row_units, sbu, shu, total);
  else if (unit == 4)
    hipLaunchKernelGGL(kv_concat_kernel<uint32_t>, grid, block, 0, st, p, H, L, K, out_cap, row_units, sbu, shu, total);
  else
    hipLaunchKernelGGL(kv_concat_kernel<uint16_t>, grid, block, 0, st, p, H, L, K, out_cap, row_units, sbu, shu, total);
  SD_LAUNCH_CHECK();
  return 0;
}

extern "C" int sd_kv_append_masked(void* out_k, void* out_v, const void* base_k,
                                   const void* base_v, const void* draft_k, const void* draft_v,
                                   const uint8_t* mask, const int32_t* accept_len, int elem_size,
                                   int B, int H, int L, int K, int D, int64_t base_sb,
                                   int64_t base_sh, void* stream) {
  using namespace sd;
  clear_error();
  SD_REQUIRE(elem_size == 2 || elem_size == 4, "kv_append_masked: elem_size %d not in {2,4}", elem_size);
  SD_REQUIRE(B >= 0 && H >= 0 && K >= 0 && D >= 0 && L >= 0, "kv_append_masked: negative dimension");
  if (B == 0 || H == 0 || D == 0 || L + K == 0) return 0;
  SD_REQUIRE(out_k && out_v, "kv_append_masked: NULL output");
  SD_REQUIRE(L == 0 || (base_k && base_v), "kv_append_masked: NULL base with L>0");
  SD_REQUIRE(K == 0 || (draft_k && draft_v && mask && accept_len), "kv_append_masked: NULL draft/mask/accept_len");
  const int row_bytes = D * elem_size;
  const int unit = pick_unit(row_bytes, elem_size,
                             {out_k, out_v, L ? base_k : out_k, L ? base_v : out_v,
                              K ? draft_k : out_k, K ? draft_v : out_v},
                             {base_sb * elem_size, base_sh * elem_size});
  const int row_units = row_bytes / unit;
  const int64_t total = static_cast<int64_t>(B) * H * (L + K) * row_units;
  KvPair p{{out_k, out_v}, {base_k, base_v}, {draft_k, draft_v}};
  dim3 grid(grid_for(total), 2), block(kCopyThreads);
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int64_t sbu = base_sb * elem_size / unit, shu = base_sh * elem_size / unit;
  if (unit == 16)
    hipLaunchKernelGGL(kv_append_masked_kernel<uint4>, grid, block, 0, st, p, mask, accept_len, H, L, K, row_units, sbu, shu, total);
  else if (unit == 4)
    hipLaunchKernelGGL(kv_append_masked_kernel<uint32_t>, grid, block, 0, st, p, mask, accept_len, H, L, K, row_units, sbu, shu, total);
  else
    hipLaunchKernelGGL(kv_append_masked_kernel<uint16_t>, grid, block, 0, st, p, mask, accept_len, H, L, K, row_units, sbu, shu, total);
  SD_LAUNCH_CHECK();
  return 0;
}
