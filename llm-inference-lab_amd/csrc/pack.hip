// Weight pre-packing for the streaming GEMV (csrc/gemv.hip).
//
// HF stores a Linear weight as [out][in] row-major. An MFMA A-fragment load of gemv.hip
// touches 16 rows x 64 bytes, i.e. 16 different DRAM pages per wave-instruction; measured,
// that pattern streams ~20 % below a row-contiguous one on MI355X. The engine therefore keeps
// a private copy of every matrix in the order the kernel consumes it: for each workgroup's
// pair range, tile by tile, every 32-k step of a tile is ONE contiguous block
//     [g = 0..3][row = 0..2*np-1] x 16 bytes      (np = pairs in the tile, <= 8)
// so a wave-instruction reads 2*np*64 contiguous bytes and consecutive steps continue the same
// stream: each wave walks one contiguous region of HBM. Rows of a tile are its pairs' first
// rows, then their second rows (pair_rows). K is padded to a multiple of 32 with zeros, a
// missing second row (odd vocabulary) is a zero row. The packed size is n_pairs*2*K32*2 bytes.
//
// fp8 storage (sd_model_config.weight_dtype = SD_FP8_E4M3): the same blocks, but a lane's 16
// bytes are 16 OCP e4m3 values — its fragments of TWO consecutive 32-k steps (k = 64 S + 8 g + 0..7
// and k = 64 S + 32 + 8 g + 0..7) — so one load feeds two MFMAs and the stream is half as long.
// Quantisation happens here, on the device, from the bf16 matrices the caller passes:
// per output row r, scale[r] = max|w[r][:]| / 448 (1 for an all-zero row), q = rne_e4m3(w / scale[r]);
// the kernel multiplies the fp32 accumulator of row r by scale[r]. The scales (fp32 [N]) follow the
// packed bytes of their matrix. K is padded to a multiple of 64.

#include "kernels.h"

namespace sd {

GemvGeom gemv_geometry(int n_pairs, int K) {
  GemvGeom q{};
  int grid = 256;  // one workgroup per CU
  if (n_pairs < grid) grid = n_pairs;
  q.ppw = (n_pairs + grid - 1) / grid;
  q.grid = (n_pairs + q.ppw - 1) / q.ppw;
  q.n_tiles = 1;
  while (q.n_tiles * 8 < q.ppw) q.n_tiles <<= 1;
  q.tile_pairs = (q.ppw + q.n_tiles - 1) / q.n_tiles;
  int ksplit = 16 / (q.n_tiles < 16 ? q.n_tiles : 16);
  while (ksplit > 1 && (K + ksplit * 32 - 1) / (ksplit * 32) < 2) ksplit >>= 1;  // >= 2 steps per slice
  q.ksplit = ksplit;
  q.kw = ((K + ksplit * 32 - 1) / (ksplit * 32)) * 32;
  return q;
}

struct PackJob {
  const uint16_t* src;
  uint16_t* dst;
  int N, K, n_pairs, epi, head_dim, ppw, tile_pairs;
};

__device__ __forceinline__ void pack_pair_rows(const PackJob& j, int p, int& r0, int& r1) {
  if (j.epi == EPI_QKV_ROPE) {
    const int half = j.head_dim >> 1;
    const int h = p / half, i = p - h * half;
    r0 = h * j.head_dim + i;
    r1 = r0 + half;
  } else if (j.epi == EPI_SWIGLU) {
    r0 = p;
    r1 = p + j.n_pairs;
  } else {
    r0 = 2 * p;
    r1 = 2 * p + 1;
  }
}

__global__ __launch_bounds__(256) void pack_kernel(PackJob j) {
  const int K32 = (j.K + 31) & ~31;
  const int qn = K32 >> 3;  // 16-byte chunks per packed row
  const size_t total = static_cast<size_t>(j.n_pairs) * 2 * qn;
  for (size_t i = static_cast<size_t>(blockIdx.x) * 256 + threadIdx.x; i < total;
       i += static_cast<size_t>(gridDim.x) * 256) {
    const int q = static_cast<int>(i % qn);
    const size_t t = i / qn;
    const int second = static_cast<int>(t & 1);
    const int p = static_cast<int>(t >> 1);
    const int c = p / j.ppw;
    const int p_lo = c * j.ppw;
    const int p_hi = min(p_lo + j.ppw, j.n_pairs);
    const int tile = (p - p_lo) / j.tile_pairs;
    const int p0 = p_lo + tile * j.tile_pairs;
    const int np = min(j.tile_pairs, p_hi - p0);
    const int jp = p - p0;
    const int S = q >> 2, g = q & 3;
    const size_t dst_chunk = static_cast<size_t>(p0) * 2 * qn + static_cast<size_t>(S) * (2 * np * 4) + g * 2 * np + second * np + jp;
    int r0, r1;
    pack_pair_rows(j, p, r0, r1);
    const int r = second ? r1 : r0;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (r < j.N && q * 8 + 8 <= j.K) v = *reinterpret_cast<const uint4*>(j.src + static_cast<size_t>(r) * j.K + q * 8);
    *reinterpret_cast<uint4*>(j.dst + dst_chunk * 8) = v;
  }
}

// scale[r] = max |w[r][:]| / 448, one wave per row
__global__ __launch_bounds__(kWave) void row_scale_kernel(const uint16_t* __restrict__ w, int N, int K, float* __restrict__ scale) {
  const int r = blockIdx.x, lane = threadIdx.x;
  float m = 0.f;
  for (int k = lane; k < K; k += kWave) m = fmaxf(m, fabsf(bf16_bits_to_float(w[static_cast<size_t>(r) * K + k])));
  m = wave_reduce_max(m);
  if (lane == 0) scale[r] = m > 0.f ? m / 448.0f : 1.0f;
}

__global__ __launch_bounds__(256) void pack_fp8_kernel(PackJob j, const float* __restrict__ scale) {
  const int K64 = (j.K + 63) & ~63;
  const int qn = K64 >> 4;  // 16-byte chunks (16 fp8) per packed row
  const size_t total = static_cast<size_t>(j.n_pairs) * 2 * qn;
  uint8_t* dst = reinterpret_cast<uint8_t*>(j.dst);
  for (size_t i = static_cast<size_t>(blockIdx.x) * 256 + threadIdx.x; i < total;
       i += static_cast<size_t>(gridDim.x) * 256) {
    const int q = static_cast<int>(i % qn);
    const size_t t = i / qn;
    const int second = static_cast<int>(t & 1);
    const int p = static_cast<int>(t >> 1);
    const int c = p / j.ppw;
    const int p_lo = c * j.ppw;
    const int p_hi = min(p_lo + j.ppw, j.n_pairs);
    const int tile = (p - p_lo) / j.tile_pairs;
    const int p0 = p_lo + tile * j.tile_pairs;
    const int np = min(j.tile_pairs, p_hi - p0);
    const int jp = p - p0;
    const int S = q >> 2, g = q & 3;  // double step (64 k), k-group
    const size_t dst_chunk = static_cast<size_t>(p0) * 2 * qn + static_cast<size_t>(S) * (2 * np * 4) + g * 2 * np + second * np + jp;
    int r0, r1;
    pack_pair_rows(j, p, r0, r1);
    const int r = second ? r1 : r0;
    uint32_t out[4] = {0u, 0u, 0u, 0u};
    if (r < j.N) {
      const float sc = scale[r];
      const uint16_t* row = j.src + static_cast<size_t>(r) * j.K;
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        const int k0 = 64 * S + 32 * half + 8 * g;
        float f[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) f[e] = (k0 + e < j.K) ? bf16_bits_to_float(row[k0 + e]) / sc : 0.f;
        int lo = 0, hi = 0;
        lo = __builtin_amdgcn_cvt_pk_fp8_f32(f[0], f[1], lo, false);
        lo = __builtin_amdgcn_cvt_pk_fp8_f32(f[2], f[3], lo, true);
        hi = __builtin_amdgcn_cvt_pk_fp8_f32(f[4], f[5], hi, false);
        hi = __builtin_amdgcn_cvt_pk_fp8_f32(f[6], f[7], hi, true);
        out[2 * half] = static_cast<uint32_t>(lo);
        out[2 * half + 1] = static_cast<uint32_t>(hi);
      }
    }
    *reinterpret_cast<uint4*>(dst + dst_chunk * 16) = make_uint4(out[0], out[1], out[2], out[3]);
  }
}

// row-major fp8 image of a matrix with the scales of row_scale_kernel (sd_quantize_fp8_rows): the same
// arithmetic as pack_fp8_kernel, kept separately so that the quantiser can be checked bit for bit
__global__ __launch_bounds__(256) void quantize_rows_kernel(const uint16_t* __restrict__ w, int N, int K,
                                                            const float* __restrict__ scale, uint8_t* __restrict__ q) {
  const size_t total = static_cast<size_t>(N) * (K >> 2);
  for (size_t i = static_cast<size_t>(blockIdx.x) * 256 + threadIdx.x; i < total; i += static_cast<size_t>(gridDim.x) * 256) {
    const size_t r = i / (K >> 2);
    const int k0 = static_cast<int>(i % (K >> 2)) * 4;
    const float sc = scale[r];
    const uint16_t* row = w + r * K + k0;
    int v = 0;
    v = __builtin_amdgcn_cvt_pk_fp8_f32(bf16_bits_to_float(row[0]) / sc, bf16_bits_to_float(row[1]) / sc, v, false);
    v = __builtin_amdgcn_cvt_pk_fp8_f32(bf16_bits_to_float(row[2]) / sc, bf16_bits_to_float(row[3]) / sc, v, true);
    *reinterpret_cast<int*>(q + r * K + k0) = v;
  }
}

// matrices of the model in packing order: per layer qkv, out, up, down; then lm_head
struct MatDesc {
  const void* w;
  int N, K, n_pairs, epi;
};

static void model_matrices(const sd_model_config& c, std::vector<MatDesc>& out) {
  const bool llama = (c.arch == SD_ARCH_LLAMA);
  const int d = c.d_model, Hq = c.n_heads, Hkv = c.n_kv_heads, D = c.head_dim, ff = c.d_ff;
  for (int l = 0; l < c.n_layers; ++l) {
    const sd_layer_weights& w = c.layers[l];
    out.push_back({w.wqkv, (Hq + 2 * Hkv) * D, d, (Hq + 2 * Hkv) * D / 2, EPI_QKV_ROPE});
    out.push_back({w.wo, d, Hq * D, d / 2, EPI_RESID});
    if (llama) out.push_back({w.w_up, 2 * ff, d, ff, EPI_SWIGLU});
    else out.push_back({w.w_up, ff, d, ff / 2, EPI_GELU});
    out.push_back({w.w_down, d, ff, d / 2, EPI_RESID});
  }
  out.push_back({c.lm_head, c.vocab, d, (c.vocab + 1) / 2, EPI_ARGMAX});
}

size_t packed_matrix_bytes(int n_pairs, int K) {
  const size_t K32 = (static_cast<size_t>(K) + 31) & ~static_cast<size_t>(31);
  return (static_cast<size_t>(n_pairs) * 2 * K32 * 2 + 255) & ~static_cast<size_t>(255);
}

// fp8: packed bytes, then the fp32 row scales
size_t packed_fp8_weight_bytes(int n_pairs, int K) {
  const size_t K64 = (static_cast<size_t>(K) + 63) & ~static_cast<size_t>(63);
  return (static_cast<size_t>(n_pairs) * 2 * K64 + 255) & ~static_cast<size_t>(255);
}
static size_t packed_fp8_matrix_bytes(int n_pairs, int K) {
  return packed_fp8_weight_bytes(n_pairs, K) + ((static_cast<size_t>(n_pairs) * 2 * 4 + 255) & ~static_cast<size_t>(255));
}
static size_t matrix_bytes(const sd_model_config& c, const MatDesc& m) {
  return c.weight_dtype == SD_FP8_E4M3 ? packed_fp8_matrix_bytes(m.n_pairs, m.K) : packed_matrix_bytes(m.n_pairs, m.K);
}

// byte offset of matrix `index` (same order as model_matrices) inside the packed buffer
size_t packed_offset(const sd_model_config& c, int index) {
  std::vector<MatDesc> mats;
  model_matrices(c, mats);
  size_t off = 0;
  for (int i = 0; i < index && i < static_cast<int>(mats.size()); ++i) off += matrix_bytes(c, mats[i]);
  return off;
}

// offset of the fp32 row scales of matrix `index` from the start of its packed bytes
size_t packed_scale_offset(const sd_model_config& c, int index) {
  std::vector<MatDesc> mats;
  model_matrices(c, mats);
  return packed_fp8_weight_bytes(mats[index].n_pairs, mats[index].K);
}

size_t packed_any_matrix_bytes(int n_pairs, int K, int weight_dtype) {
  return weight_dtype == SD_FP8_E4M3 ? packed_fp8_matrix_bytes(n_pairs, K) : packed_matrix_bytes(n_pairs, K);
}

int pack_one_matrix(const void* w_bf16, int N, int K, int n_pairs, int epi, int head_dim, int weight_dtype, void* dst, hipStream_t st) {
  SD_REQUIRE(w_bf16 && dst, "pack: NULL matrix");
  SD_REQUIRE(K % 8 == 0, "pack: K=%d must be a multiple of 8", K);
  const GemvGeom q = gemv_geometry(n_pairs, K);
  PackJob j{static_cast<const uint16_t*>(w_bf16), static_cast<uint16_t*>(dst), N, K, n_pairs, epi, head_dim, q.ppw, q.tile_pairs};
  if (weight_dtype == SD_FP8_E4M3) {
    float* scale = reinterpret_cast<float*>(static_cast<char*>(dst) + packed_fp8_weight_bytes(n_pairs, K));
    hipLaunchKernelGGL(row_scale_kernel, dim3(N), dim3(kWave), 0, st, static_cast<const uint16_t*>(w_bf16), N, K, scale);
    SD_LAUNCH_CHECK();
    hipLaunchKernelGGL(pack_fp8_kernel, dim3(2048), dim3(256), 0, st, j, scale);
  } else {
    hipLaunchKernelGGL(pack_kernel, dim3(2048), dim3(256), 0, st, j);
  }
  SD_LAUNCH_CHECK();
  return 0;
}

}  // namespace sd

using namespace sd;

// A vocabulary-sized output matrix [V][d] outside a model (Medusa heads): same layout and quantiser as the lm_head.
extern "C" size_t sd_packed_head_bytes(int vocab, int d_model, int weight_dtype) {
  return packed_any_matrix_bytes((vocab + 1) / 2, d_model, weight_dtype);
}

extern "C" int sd_pack_head(const void* w_bf16, int vocab, int d_model, int weight_dtype, void* dst, size_t dst_bytes, void* stream) {
  clear_error();
  SD_REQUIRE(weight_dtype == SD_BF16 || weight_dtype == SD_FP8_E4M3, "pack_head: weight_dtype %d", weight_dtype);
  SD_REQUIRE(dst_bytes >= sd_packed_head_bytes(vocab, d_model, weight_dtype), "pack_head: destination too small");
  SD_REQUIRE((reinterpret_cast<uintptr_t>(dst) & 255) == 0, "pack_head: destination must be 256-byte aligned");
  return pack_one_matrix(w_bf16, vocab, d_model, (vocab + 1) / 2, EPI_ARGMAX, 0, weight_dtype, dst, static_cast<hipStream_t>(stream));
}

extern "C" size_t sd_packed_bytes(const sd_model_config* cfg) {
  if (!cfg || !cfg->layers) return 0;
  std::vector<MatDesc> mats;
  model_matrices(*cfg, mats);
  size_t n = 0;
  for (const MatDesc& m : mats) n += matrix_bytes(*cfg, m);
  return n;
}

extern "C" int sd_pack_weights(const sd_model_config* cfg, void* dst, size_t dst_bytes, void* stream) {
  clear_error();
  SD_REQUIRE(cfg && cfg->layers && dst, "pack_weights: NULL argument");
  SD_REQUIRE(cfg->weight_dtype == SD_BF16 || cfg->weight_dtype == SD_FP8_E4M3, "pack_weights: weight_dtype %d (bf16 or fp8 e4m3)", cfg->weight_dtype);
  const bool fp8 = cfg->weight_dtype == SD_FP8_E4M3;
  SD_REQUIRE(dst_bytes >= sd_packed_bytes(cfg), "pack_weights: destination too small");
  SD_REQUIRE((reinterpret_cast<uintptr_t>(dst) & 255) == 0, "pack_weights: destination must be 256-byte aligned");
  std::vector<MatDesc> mats;
  model_matrices(*cfg, mats);
  hipStream_t st = static_cast<hipStream_t>(stream);
  char* p = static_cast<char*>(dst);
  for (const MatDesc& m : mats) {
    SD_REQUIRE(m.w, "pack_weights: NULL matrix");
    SD_REQUIRE(m.K % 8 == 0, "pack_weights: K=%d must be a multiple of 8", m.K);
    const GemvGeom q = gemv_geometry(m.n_pairs, m.K);
    PackJob j{static_cast<const uint16_t*>(m.w), reinterpret_cast<uint16_t*>(p), m.N, m.K, m.n_pairs, m.epi,
              cfg->head_dim, q.ppw, q.tile_pairs};
    if (fp8) {
      float* scale = reinterpret_cast<float*>(p + packed_fp8_weight_bytes(m.n_pairs, m.K));
      hipLaunchKernelGGL(row_scale_kernel, dim3(m.N), dim3(kWave), 0, st, static_cast<const uint16_t*>(m.w), m.N, m.K, scale);
      SD_LAUNCH_CHECK();
      hipLaunchKernelGGL(pack_fp8_kernel, dim3(2048), dim3(256), 0, st, j, scale);
    } else {
      hipLaunchKernelGGL(pack_kernel, dim3(2048), dim3(256), 0, st, j);
    }
    SD_LAUNCH_CHECK();
    p += matrix_bytes(*cfg, m);
  }
  return 0;
}

extern "C" int sd_quantize_fp8_rows(const void* w_bf16, int N, int K, void* q_fp8, float* scales, void* stream) {
  clear_error();
  SD_REQUIRE(w_bf16 && q_fp8 && scales, "quantize_fp8_rows: NULL argument");
  SD_REQUIRE(N >= 1 && K >= 4 && K % 4 == 0, "quantize_fp8_rows: N=%d K=%d (K must be a multiple of 4)", N, K);
  hipStream_t st = static_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(row_scale_kernel, dim3(N), dim3(kWave), 0, st, static_cast<const uint16_t*>(w_bf16), N, K, scales);
  SD_LAUNCH_CHECK();
  hipLaunchKernelGGL(quantize_rows_kernel, dim3(1024), dim3(256), 0, st, static_cast<const uint16_t*>(w_bf16), N, K, scales,
                     static_cast<uint8_t*>(q_fp8));
  SD_LAUNCH_CHECK();
  return 0;
}
