// Shared host/device helpers for the gfx950 speculative-decoding library.
// Not part of the C-ABI (see include/specdec_hip.h for that).
#pragma once

#include <hip/hip_runtime.h>
#include <hip/hip_bf16.h>
#include <hip/hip_fp16.h>
#include <stdint.h>
#include <stdio.h>
#include <string>

#include "../../include/specdec_hip.h"

namespace sd {

// scalar kernel arguments made live at one point (see pin_gemv_args, gemv_device.h): the s_loads are issued back to back
#define SD_PIN(...) asm volatile("" ::__VA_ARGS__)

// Per-row adaptive K (sd_specdec_set_adaptive): launches of draft forward i >= 1 are in the captured step for every
// i < K, but when no row proposes more than *k_active tokens the ones with i >= *k_active leave at once (one scalar
// load; k_active == null, i.e. every other launch, costs a compare). Uniform: every wave takes the same path.
// As ONE opaque statement (scalar compare, conditional scalar load, s_endpgm): written as C++ (`if (...) return;`) the
// early exit splits the kernel's entry block, and everything the scheduler used to place under the latency of the
// argument loads (lane / tile index arithmetic) then waits behind the branch — 0.5 % of the batch-1 step with the
// check compiled in and never taken. Nothing is in flight that matters (the hardware drains a wave's outstanding
// loads at s_endpgm).
#define SD_SKIP_IF_INACTIVE(kptr, level)                                          \
  do {                                                                            \
    int sd_skip_tmp_;                                                             \
    asm volatile(                                                                 \
        "s_cmp_eq_u64 %1, 0\n\t"                                                  \
        "s_cbranch_scc1 1f\n\t"                                                   \
        "s_load_dword %0, %1, 0x0\n\t"                                            \
        "s_waitcnt lgkmcnt(0)\n\t"                                                \
        "s_cmp_gt_i32 %0, %2\n\t"                                                 \
        "s_cbranch_scc1 1f\n\t"                                                   \
        "s_endpgm\n"                                                              \
        "1:"                                                                      \
        : "=&s"(sd_skip_tmp_)                                                     \
        : "s"(kptr), "s"(static_cast<int>(level))                                 \
        : "scc");                                                                 \
  } while (0)

// ---- error plumbing -------------------------------------------------------
void set_error(const char* fmt, ...);
void clear_error();

#define SD_REQUIRE(cond, ...)            \
  do {                                   \
    if (!(cond)) {                       \
      ::sd::set_error(__VA_ARGS__);      \
      return 1;                          \
    }                                    \
  } while (0)

#define SD_HIP_CHECK(expr)                                                   \
  do {                                                                       \
    hipError_t _e = (expr);                                                  \
    if (_e != hipSuccess) {                                                  \
      ::sd::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), \
                      __FILE__, __LINE__);                                   \
      return 2;                                                              \
    }                                                                        \
  } while (0)

// A launch inside a stream capture cannot be followed by hipGetLastError-style
// polling of a sticky error only; hipGetLastError is capture-safe.
#define SD_LAUNCH_CHECK() SD_HIP_CHECK(hipGetLastError())

// Dynamic LDS above 64 KiB has to be opted into per kernel AND per device (a process that drives several GPUs sets it on each).
// `seen`: the call site's static bitmask of the devices it has done. 0 = ok.
inline int opt_in_dynamic_lds(const void* kernel, int bytes, unsigned long long& seen) {
  int dev = 0;
  SD_HIP_CHECK(hipGetDevice(&dev));
  if (dev < 0 || dev >= 64 || !((seen >> dev) & 1ull)) {
    SD_HIP_CHECK(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    if (dev >= 0 && dev < 64) seen |= 1ull << dev;
  }
  return 0;
}

inline int dtype_size(int dt) {
  switch (dt) {
    case SD_F32: return 4;
    case SD_F16: return 2;
    case SD_BF16: return 2;
    case SD_I32: return 4;
    case SD_I64: return 8;
    case SD_U8: return 1;
    case SD_FP8_E4M3: return 1;
    default: return 0;
  }
}

constexpr int kWave = 64;  // gfx950 wavefront

// ---- device helpers ---------------------------------------------------------
#if defined(__HIPCC__)

__device__ __forceinline__ float bf16_bits_to_float(uint16_t b) {
  return __uint_as_float(static_cast<uint32_t>(b) << 16);
}

// round-to-nearest-even f32 -> bf16 that keeps NaN a NaN (plain cast lowers to
// v_cvt_pk_bf16_f32 on gfx950)
__device__ __forceinline__ uint16_t float_to_bf16_bits(float f) {
  __hip_bfloat16 h = __float2bfloat16(f);
  return *reinterpret_cast<uint16_t*>(&h);
}

// Two bf16 values of one 32-bit word <-> two floats, on the packed-math VALU path of gfx950: the unpack is two bit
// operations, the arithmetic in between is v_pk_mul_f32 / v_pk_fma_f32 (two lanes of fp32 per instruction) and the
// re-pack is ONE v_cvt_pk_bf16_f32 (round to nearest even, NaN stays NaN) — half the instructions of the scalar
// sequence. The normalisation fused into the staging of x is VALU-bound (every workgroup normalises all T x K
// activations), so this is where the instruction count matters.
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2_t bf16x2_unpack(uint32_t u) { return f32x2_t{__uint_as_float(u << 16), __uint_as_float(u & 0xffff0000u)}; }
__device__ __forceinline__ uint32_t bf16x2_pack(f32x2_t v) { return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t)); }
// HF LlamaRMSNorm on a pair: weight * (x * rstd).to(bf16), rounded to bf16 again
__device__ __forceinline__ uint32_t rmsnorm_pair(uint32_t x, float rs, uint32_t w) {
  const f32x2_t xn = bf16x2_unpack(bf16x2_pack(bf16x2_unpack(x) * rs));
  return bf16x2_pack(xn * bf16x2_unpack(w));
}
// GPT-2 LayerNorm on a pair: ((x - mean) * rstd) * w + b in fp32, rounded once
__device__ __forceinline__ uint32_t layernorm_pair(uint32_t x, float mean, float rs, uint32_t w, uint32_t b) {
  return bf16x2_pack((bf16x2_unpack(x) - mean) * rs * bf16x2_unpack(w) + bf16x2_unpack(b));
}

// Sum over the 64 lanes, result in every lane. Data-parallel-primitive moves on the VALU (v_add_f32 with a DPP
// operand), not __shfl_xor: that lowers to ds_bpermute_b32, which occupies the LDS pipe — with 16 waves per CU
// doing 2 x T reductions in the norm prologue of every GEMV the LDS pipe, not HBM, set the prologue's length
// (7 us at 5 tokens, measured with the in-kernel timeline).
#define SD_DPP_F32(v, old, ctrl, row_mask) \
  __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(v), ctrl, row_mask, 0xf, false))

// sum over each row of 16 lanes, result in every lane of the row
__device__ __forceinline__ float row16_reduce_sum(float v) {
  v += SD_DPP_F32(v, v, 0xb1, 0xf);    // quad_perm:[1,0,3,2]
  v += SD_DPP_F32(v, v, 0x4e, 0xf);    // quad_perm:[2,3,0,1]
  v += SD_DPP_F32(v, v, 0x124, 0xf);   // row_ror:4
  v += SD_DPP_F32(v, v, 0x128, 0xf);   // row_ror:8
  return v;
}

__device__ __forceinline__ float wave_reduce_sum(float v) {
  v = row16_reduce_sum(v);
  v += SD_DPP_F32(v, 0.f, 0x142, 0xa);  // row_bcast:15 into rows 1 and 3
  v += SD_DPP_F32(v, 0.f, 0x143, 0xc);  // row_bcast:31 into rows 2 and 3: lane 63 holds the total
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

__device__ __forceinline__ float wave_reduce_max(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
  return v;
}

// torch.argmax ordering: NaN is the maximum, ties go to the lower index.
__device__ __forceinline__ bool argmax_better(float v, int i, float bv, int bi) {
  const bool vn = (v != v), bn = (bv != bv);
  if (vn | bn) {
    if (vn & bn) return i < bi;
    return vn;
  }
  return (v > bv) | ((v == bv) & (i < bi));
}

__device__ __forceinline__ void wave_reduce_argmax(float& v, int& i) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    float ov = __shfl_xor(v, off, 64);
    int oi = __shfl_xor(i, off, 64);
    if (argmax_better(ov, oi, v, i)) {
      v = ov;
      i = oi;
    }
  }
}

#endif  // __HIPCC__

}  // namespace sd
