// Two dependent weight-streaming GEMVs in ONE launch, separated by an in-kernel grid barrier:
//   [attention out-projection + residual]  ->  [norm + gate/up + SwiGLU]
//   [down projection + residual]           ->  [norm + QKV + RoPE + in-place KV append of the next layer]
//
// Why (profiles/round1_grid_barrier_microbench.txt): at batch 1 every GEMV of a step costs its stream
// time plus ~4.5 us of fixed latency (1.57 us kernel boundary, kernel arguments, the x rows, reduce, epilogue) during
// which HBM idles; 465 launches make that 2.1 ms of a 4.5 ms step. A kernel boundary cannot overlap anything. An
// in-kernel barrier can: the second matrix's first weight batch (up to 12 KiB per wave, 46 MB over the chip) is issued
// BEFORE the barrier and streams while the first phase's results cross the chip. The barrier that makes this pay is
// FENCE-FREE: an agent-scope release/acquire pair costs a whole-L2 write-back and invalidate (7.3 us per barrier
// measured), so the only data exchanged inside the launch — the new residual rows, a few KiB — moves with agent-scope
// RELAXED atomics instead (stores write through to memory, loads bypass this XCD's L2), ordered by waiting for the
// stores before the arrival and by the control dependency on the poll: 2.9 us alone, +0.8..1.3 us on top of a weight
// stream in flight.
//
// Restrictions (the launcher falls back to two launches otherwise): bf16 weights, T <= 9 tokens, K <= 8192 in whole
// 32-k steps per slice, both matrices split over exactly 256 workgroups on a 256-CU device (co-residency of the
// grid is what makes the barrier legal: one 1024-thread workgroup per CU, nothing else running on the stream's device
// share). The spin is bounded: after kSpinLimit polls a workgroup sets *err and carries on (wrong data, no hang).
//
// The phase bodies are the gemv.hip kernel (same work split, same operand order, same epilogues => identical bits).

#include <stdlib.h>

#include "gemv_device.h"

namespace sd {

constexpr int kChainPre = 8;            // phase 1: weight loads issued before the x staging (as in gemv.hip)
constexpr int kSpinLimit = 100000;      // polls (~1 us each) before a workgroup gives up on the barrier
constexpr int kSyncGroups = 16;         // arrival counters, 1 KiB apart (different channels); [0] is the top counter
constexpr int kSyncStride = 256;        // in 4-byte words

struct ChainArgs {
  GemvArgs a;       // phase 1: EPI_RESID, no prologue
  GemvArgs b;       // phase 2: norm prologue + EPI_QKV_ROPE / EPI_SWIGLU
  unsigned* sync;   // (1 + kSyncGroups) * kSyncStride words, zeroed once; monotonic, never reset
  unsigned* err;    // set to 1 when a barrier timed out
};

// 16 bytes of x written earlier in THIS launch by another workgroup: two agent-scope relaxed 8-byte loads
__device__ __forceinline__ u32x4 coherent_load16(const uint16_t* p) {
  const unsigned long long* q = reinterpret_cast<const unsigned long long*>(p);
  const unsigned long long lo = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const unsigned long long hi = __hip_atomic_load(q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return u32x4{static_cast<uint32_t>(lo), static_cast<uint32_t>(lo >> 32), static_cast<uint32_t>(hi), static_cast<uint32_t>(hi >> 32)};
}

// One phase. PH = 1: x with plain loads at entry, coherent residual stores. PH = 2: the first weight batch is
// issued, THEN the grid barrier is crossed, then x is read with coherent loads.
template <int EPI, int TT, int KB, int PH>
__device__ __forceinline__ void chain_phase(const GemvArgs& a, unsigned char* smem, unsigned* sync, unsigned epoch, unsigned* err) {
  constexpr int kBatch = KB;
  const int K = a.K, T = a.T;
  const int KP = K + kXPad;
  uint16_t* xs = reinterpret_cast<uint16_t*>(smem);
  const size_t xs_bytes = (static_cast<size_t>(T) * KP * 2 + 15) & ~static_cast<size_t>(15);
  float* part = reinterpret_cast<float*>(a.alias_part ? smem : smem + xs_bytes);  // [16 waves][16][16]
  float* red = reinterpret_cast<float*>(smem + xs_bytes + (a.alias_part ? 0 : sizeof(float) * kGemvWaves * 256));

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, n = lane & 15;
  const int ksplit = a.ksplit;
  const int tiles_per_round = kGemvWaves / ksplit;
  const int kpart = wave & (ksplit - 1);
  const int tslot = wave / ksplit;
  const int kw = a.kw;
  const int k_begin = kpart * kw;
  const int steps = kw >> 5;
  const uint16_t* W = static_cast<const uint16_t*>(a.W);

  const int p_lo = static_cast<int>(blockIdx.x) * a.ppw;
  const int p_hi = min(p_lo + a.ppw, a.n_pairs);
  const int tile_pairs = a.tile_pairs;
  const int n_tiles = (p_hi - p_lo + tile_pairs - 1) / tile_pairs;
  const int rounds = (n_tiles + tiles_per_round - 1) / tiles_per_round;

  int wstride = 32;
  const int K32 = (K + 31) & ~31;
  unsigned lane_off = 0;
  auto tile_base = [&](int tile) -> const uint16_t* {
    const int p0 = p_lo + tile * tile_pairs;
    if (a.packed) {
      int np = min(tile_pairs, p_hi - p0);
      if (np < 1) np = 1;
      int jp = n & 7, second = n >> 3;
      if (jp >= np) { jp = 0; second = 0; }
      wstride = np * 64;
      lane_off = static_cast<unsigned>((g * 2 * np + second * np + jp) * 8);
      return W + static_cast<size_t>(p0) * 2 * K32 + static_cast<size_t>(k_begin >> 5) * wstride;
    }
    int p = p0 + (n & 7);
    int second = n >> 3;
    if ((n & 7) >= tile_pairs || p >= p_hi) { p = min(p0, p_hi - 1); second = 0; }
    int r0, r1;
    pair_rows<EPI>(a, p, r0, r1);
    int r = second ? r1 : r0;
    if (r >= a.N) r = r0;
    lane_off = static_cast<unsigned>(r) * static_cast<unsigned>(K) + static_cast<unsigned>(g * 8);
    return W + k_begin;
  };

  // diagnostic timeline (SPECDEC_GEMV_TIMELINE=1 through sd_model_probe_gemv): 100 MHz stamps of the polling thread
  auto stamp = [&](int slot) {
    if (a.debug_ts && tid == (kGemvWaves - 1) * 64) a.debug_ts[static_cast<size_t>(blockIdx.x) * 16 + (PH - 1) * 8 + slot] = __builtin_amdgcn_s_memrealtime();
  };
  stamp(0);

  u32x4 buf[kBatch];
  auto issue_first = [&](const uint16_t* ubase, int j0, int j1) {
#pragma unroll
    for (int j = 0; j < kBatch; ++j) {
      if (j >= j0 && j < j1) {
        const int s = (j < steps) ? j : steps - 1;
        buf[j] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(ubase + static_cast<size_t>(s) * wstride + lane_off));
      }
    }
  };
  auto issue = [&](const uint16_t* ubase, int s0) {
#pragma unroll
    for (int j = 0; j < kBatch; ++j) {
      const int s = s0 + j;
      if (s < steps) buf[j] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(ubase + static_cast<size_t>(s) * wstride + lane_off));
      else buf[j] = u32x4{0u, 0u, 0u, 0u};
    }
  };

  const int nvec = K >> 3;
  const bool has_chunk = tid < nvec;
  const int cidx = has_chunk ? tid : nvec - 1;
  u32x4 xr[TT];
  u32x4 nw4 = {0u, 0u, 0u, 0u}, nb4 = {0u, 0u, 0u, 0u};
  bool have_old = false;
  uint32_t old_pre = 0;
  const uint16_t* xin = static_cast<const uint16_t*>(a.x);
  const bool first_valid = tslot < n_tiles;
  const uint16_t* wrow0 = tile_base(first_valid ? tslot : 0);
  // the wave whose lane 0 polls the barrier keeps its memory queue empty: vmcnt retires in order, a poll behind
  // 12 KiB of weights would see the barrier only after they have arrived
  const bool sync_wave = (wave == kGemvWaves - 1);

  if constexpr (PH == 1) {
#pragma unroll
    for (int t = 0; t < TT; ++t) {
      const int tt = (t < T) ? t : T - 1;
      xr[t] = *reinterpret_cast<const u32x4*>(xin + static_cast<size_t>(tt) * a.x_stride + cidx * 8);
    }
    if constexpr (EPI == EPI_RESID) {
      const int ts = tid >> 7, jp = (tid >> 4) & 7, t = tid & 15;
      const int p = p_lo + ts * tile_pairs + jp;
      have_old = ts < tiles_per_round && ts < n_tiles && jp < tile_pairs && p < p_hi && t < T;
      const int pc = have_old ? p : p_lo, tc = have_old ? t : 0;
      old_pre = *reinterpret_cast<const uint32_t*>(static_cast<const uint16_t*>(a.out) + static_cast<size_t>(tc) * a.out_stride + 2 * pc);
    }
    issue_first(wrow0, 0, kChainPre);
  } else {
    // constants of the prologue and the whole first weight batch go out before the barrier
    if (a.prologue != PRO_NONE) {
      nw4 = *reinterpret_cast<const u32x4*>(static_cast<const uint16_t*>(a.norm_w) + cidx * 8);
      if (a.prologue == PRO_LAYERNORM) nb4 = *reinterpret_cast<const u32x4*>(static_cast<const uint16_t*>(a.norm_b) + cidx * 8);
    }
    if (!sync_wave) issue_first(wrow0, 0, kBatch);
    stamp(1);
    __syncthreads();   // every thread of this workgroup has waited for its phase-1 stores (see the kernel)
    if (tid == (kGemvWaves - 1) * 64) {
      unsigned* grp = sync + kSyncStride * (1 + (blockIdx.x & (kSyncGroups - 1)));
      const unsigned per = gridDim.x / kSyncGroups;
      const unsigned old = __hip_atomic_fetch_add(grp, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (old == (epoch + 1) * per - 1) __hip_atomic_fetch_add(sync, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned target = (epoch + 1) * kSyncGroups;
      int it = 0;
      while (static_cast<int>(__hip_atomic_load(sync, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - target) < 0) {
        if (++it > kSpinLimit) { *err = 1u; break; }
      }
    }
    __syncthreads();
    stamp(2);
#pragma unroll
    for (int t = 0; t < TT; ++t) {
      const int tt = (t < T) ? t : T - 1;
      xr[t] = coherent_load16(xin + static_cast<size_t>(tt) * a.x_stride + cidx * 8);
    }
    if (sync_wave) issue_first(wrow0, 0, kBatch);
  }

  // ---- x staging (as gemv.hip's fast path)
  if (a.prologue == PRO_NONE) {
#pragma unroll
    for (int t = 0; t < TT; ++t)
      if (t < T && has_chunk) *reinterpret_cast<u32x4*>(xs + static_cast<size_t>(t) * KP + tid * 8) = xr[t];
    __syncthreads();
  } else {
    const bool wave_has_chunk = (wave << 6) < nvec;
#pragma unroll
    for (int t = 0; t < TT; ++t) {
      if (t < T) {
        float s1 = 0.f, s2 = 0.f;
        if (wave_has_chunk) {
          const bool need_mean = a.prologue == PRO_LAYERNORM;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float lo = __uint_as_float(xr[t][j] << 16), hi = __uint_as_float(xr[t][j] & 0xffff0000u);
            if (need_mean) s1 += lo + hi;
            s2 += lo * lo + hi * hi;
          }
          if (!has_chunk) { s1 = 0.f; s2 = 0.f; }
          if (need_mean) s1 = wave_reduce_sum(s1);
          s2 = wave_reduce_sum(s2);
        }
        if (lane == 0) {
          red[(t * kGemvWaves + wave) * 2 + 0] = s1;
          red[(t * kGemvWaves + wave) * 2 + 1] = s2;
        }
      }
    }
    __syncthreads();
    const float invK = 1.0f / static_cast<float>(K);
#pragma unroll
    for (int t = 0; t < TT; ++t) {
      if (t < T && wave_has_chunk) {
        const float2 pr = *reinterpret_cast<const float2*>(red + (t * kGemvWaves + (lane & 15)) * 2);
        const float sum = row16_reduce_sum(pr.x), sq = row16_reduce_sum(pr.y);
        u32x4 o;
        if (a.prologue == PRO_RMSNORM) {
          const float rs = rsqrtf(sq * invK + a.norm_eps);
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float x0 = bf16_bits_to_float(float_to_bf16_bits(__uint_as_float(xr[t][j] << 16) * rs));
            const float x1 = bf16_bits_to_float(float_to_bf16_bits(__uint_as_float(xr[t][j] & 0xffff0000u) * rs));
            o[j] = static_cast<uint32_t>(float_to_bf16_bits(x0 * __uint_as_float(nw4[j] << 16))) |
                   (static_cast<uint32_t>(float_to_bf16_bits(x1 * __uint_as_float(nw4[j] & 0xffff0000u))) << 16);
          }
        } else {
          const float mean = sum * invK;
          const float rs = rsqrtf(fmaxf(sq * invK - mean * mean, 0.f) + a.norm_eps);
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float y0 = (__uint_as_float(xr[t][j] << 16) - mean) * rs * __uint_as_float(nw4[j] << 16) + __uint_as_float(nb4[j] << 16);
            const float y1 = (__uint_as_float(xr[t][j] & 0xffff0000u) - mean) * rs * __uint_as_float(nw4[j] & 0xffff0000u) +
                             __uint_as_float(nb4[j] & 0xffff0000u);
            o[j] = static_cast<uint32_t>(float_to_bf16_bits(y0)) | (static_cast<uint32_t>(float_to_bf16_bits(y1)) << 16);
          }
        }
        if (has_chunk) *reinterpret_cast<u32x4*>(xs + static_cast<size_t>(t) * KP + tid * 8) = o;
      }
    }
    __syncthreads();
  }
  if constexpr (PH == 1) issue_first(wrow0, kChainPre, kBatch);
  stamp(3);

  float best_v = -INFINITY;
  int best_i = 0x7fffffff;
  const uint16_t* xrow = xs + static_cast<size_t>(n < T ? n : T - 1) * KP + k_begin + g * 8;

  for (int r = 0; r < rounds; ++r) {
    const int tile = r * tiles_per_round + tslot;
    f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
    if (tile < n_tiles) {
      const uint16_t* wrow = (r == 0) ? wrow0 : tile_base(tile);
      for (int s0 = 0; s0 < steps; s0 += kBatch) {
        if (r != 0 || s0 != 0) issue(wrow, s0);
#pragma unroll
        for (int j = 0; j < kBatch; ++j) {
          const int s = s0 + j;
          if (s < steps) {
            const u32x4 xb = *reinterpret_cast<const u32x4*>(xrow + s * 32);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, buf[j]), __builtin_bit_cast(bf16x8_t, xb), acc, 0, 0, 0);
          }
        }
      }
    }
    if (r == 0) stamp(4);
    if (a.alias_part) __syncthreads();
    float* slot = part + wave * 256;
#pragma unroll
    for (int q = 0; q < 4; ++q) slot[(4 * g + q) * 16 + n] = acc[q];
    __syncthreads();
    for (int it = tid; it < tiles_per_round * 128; it += kGemvThreads) {
      const int ts = it >> 7, jp = (it >> 4) & 7, t = it & 15;
      const int etile = r * tiles_per_round + ts;
      const int p = p_lo + etile * tile_pairs + jp;
      if (etile < n_tiles && jp < tile_pairs && p < p_hi && t < T) {
        const float* base = part + (ts * ksplit) * 256;
        float y0 = 0.f, y1 = 0.f;
        for (int w = 0; w < ksplit; ++w) {
          y0 += base[w * 256 + jp * 16 + t];
          y1 += base[w * 256 + (jp + 8) * 16 + t];
        }
        int r0, r1;
        pair_rows<EPI>(a, p, r0, r1);
        epilogue<EPI, PH == 1>(a, p, r0, r1, t, y0, y1, best_v, best_i, have_old && r == 0 && it == tid, old_pre);
      }
    }
    if (r + 1 < rounds) __syncthreads();
  }
  stamp(5);
}

template <int EPI_B, int TT, int KBA, int KBB>
__global__ __launch_bounds__(kGemvThreads) void gemv_chain_kernel(const ChainArgs c) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  pin_gemv_args<EPI_RESID, false>(c.a);
  pin_gemv_args<EPI_B, false>(c.b);
  // barriers completed before this launch: the top counter gains kSyncGroups per barrier and cannot pass
  // (epoch + 1) * kSyncGroups before every workgroup of THIS launch has arrived, so every workgroup reads the same epoch
  // (every thread loads it, unconditionally and first: inside a branch hipcc waits for the value on the spot, a
  // memory round trip in front of phase 1; here it retires with the first x loads)
  // and through a lane offset the compiler cannot see is zero: a load from a uniform address is moved to an SGPR with
  // v_readfirstlane right away, which is the same wait)
  unsigned lane_zero;
  asm volatile("v_mov_b32 %0, 0" : "=v"(lane_zero));
  const unsigned top = __hip_atomic_load(c.sync + lane_zero, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  chain_phase<EPI_RESID, TT, KBA, 1>(c.a, smem, c.sync, 0u, c.err);
  // the residual rows of this workgroup must be in memory before it arrives: every thread waits for its own stores
  // (on gfx950 stores count in vmcnt); only then the second phase starts issuing loads
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_s_waitcnt(0);
  if (c.a.debug_ts && threadIdx.x == (kGemvWaves - 1) * 64) c.a.debug_ts[static_cast<size_t>(blockIdx.x) * 16 + 6] = __builtin_amdgcn_s_memrealtime();
  chain_phase<EPI_B, TT, KBB, 2>(c.b, smem, c.sync, top / kSyncGroups, c.err);
}

// ------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------
constexpr size_t kLdsLimitChain = 160 * 1024;
static size_t chain_smem(int T, int K, bool alias) {
  const size_t xs = (static_cast<size_t>(T) * (K + kXPad) * 2 + 15) & ~static_cast<size_t>(15);
  const size_t part = sizeof(float) * kGemvWaves * 256;
  const size_t red = sizeof(float) * (kGemvMaxT * kGemvWaves * 2 + 64);
  return (alias ? (xs > part ? xs : part) : xs + part) + red;
}

size_t chain_sync_bytes() { return sizeof(unsigned) * (1 + kSyncGroups) * kSyncStride + 256; }

static bool prepare_phase(GemvArgs& a, size_t* smem) {
  const GemvGeom q = gemv_geometry(a.n_pairs, a.K);
  if (q.grid != 256) return false;
  a.ppw = q.ppw;
  a.tile_pairs = q.tile_pairs;
  a.ksplit = q.ksplit;
  a.kw = q.kw;
  a.n_tiles_full = (q.ppw + q.tile_pairs - 1) / q.tile_pairs;
  gemv_derive(a);
  if (a.kw * q.ksplit != a.K || a.K / 8 > kGemvThreads || a.K % 8 != 0 || a.x_stride % 8 != 0) return false;   // gemv.hip's MASK shapes
  size_t s = chain_smem(a.T, a.K, false);
  a.alias_part = 0;
  if (s > kLdsLimitChain && q.n_tiles <= kGemvWaves / q.ksplit) {
    a.alias_part = 1;
    s = chain_smem(a.T, a.K, true);
  }
  if (s > kLdsLimitChain) return false;
  *smem = s;
  return true;
}

// true when the pair can run as one chained launch (the caller launches the two GEMVs separately otherwise)
bool gemv_chain_covers(const GemvArgs& a_in, const GemvArgs& b_in, int epi_b) {
  static int cus = -1;
  if (cus < 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    cus = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ? prop.multiProcessorCount : 0;
  }
  if (cus != 256) return false;   // one workgroup per CU, all co-resident
  if (a_in.w8 || b_in.w8 || a_in.x_row || b_in.x_row) return false;
  if (a_in.T != b_in.T || a_in.T < 1 || a_in.T > kGemvMaxT) return false;
  if (a_in.prologue != PRO_NONE || b_in.prologue == PRO_NONE) return false;
  if (epi_b != EPI_QKV_ROPE && epi_b != EPI_SWIGLU) return false;
  GemvArgs a = a_in, b = b_in;
  size_t sa = 0, sb = 0;
  return prepare_phase(a, &sa) && prepare_phase(b, &sb);
}

template <int EPI_B, int TT, int KBA, int KBB>
static int launch_chain_one(const ChainArgs& c, size_t smem, hipStream_t st) {
  static bool attr_set = false;
  if (!attr_set) {
    SD_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemv_chain_kernel<EPI_B, TT, KBA, KBB>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_set = true;
  }
  hipLaunchKernelGGL((gemv_chain_kernel<EPI_B, TT, KBA, KBB>), dim3(256), dim3(kGemvThreads), smem, st, c);
  SD_LAUNCH_CHECK();
  return 0;
}

template <int EPI_B, int KBA, int KBB>
static int launch_chain_t(const ChainArgs& c, size_t smem, hipStream_t st) {
  const int T = c.a.T;
  if (T <= 1) return launch_chain_one<EPI_B, 1, KBA, KBB>(c, smem, st);
  if (T <= 2) return launch_chain_one<EPI_B, 2, KBA, KBB>(c, smem, st);
  if (T <= 3) return launch_chain_one<EPI_B, 3, KBA, KBB>(c, smem, st);
  if (T <= 5) return launch_chain_one<EPI_B, 5, KBA, KBB>(c, smem, st);
  return launch_chain_one<EPI_B, kGemvMaxT, KBA, KBB>(c, smem, st);
}

template <int EPI_B>
static int launch_chain_epi(const ChainArgs& c, size_t smem, hipStream_t st) {
  // batch depth per phase as gemv.hip chooses it: deep (12 loads in flight) for long slices, shallow (6) otherwise
  // The second phase always takes the deep batch (when its slices are long enough to use it): what it has in flight
  // before the barrier is all that streams while the barrier and the x rows take their ~6 us.
  auto deep = [](const GemvArgs& g) { return (g.kw >> 5) >= (g.T >= 3 ? 16 : 24); };
  const bool da = deep(c.a), db = (c.b.kw >> 5) > 6;
  if (da && db) return launch_chain_t<EPI_B, 12, 12>(c, smem, st);
  if (da) return launch_chain_t<EPI_B, 12, 6>(c, smem, st);
  if (db) return launch_chain_t<EPI_B, 6, 12>(c, smem, st);
  return launch_chain_t<EPI_B, 6, 6>(c, smem, st);
}

int launch_gemv_chain(const GemvArgs& a_in, const GemvArgs& b_in, int epi_b, unsigned* sync, unsigned* err, hipStream_t st) {
  SD_REQUIRE(sync && err, "gemv_chain: NULL sync buffer");
  SD_REQUIRE(gemv_chain_covers(a_in, b_in, epi_b), "gemv_chain: shape not covered (call gemv_chain_covers first)");
  ChainArgs c{};
  c.a = a_in;
  c.b = b_in;
  c.sync = sync;
  c.err = err;
  size_t sa = 0, sb = 0;
  prepare_phase(c.a, &sa);
  prepare_phase(c.b, &sb);
  const size_t smem = sa > sb ? sa : sb;
  if (epi_b == EPI_QKV_ROPE) return launch_chain_epi<EPI_QKV_ROPE>(c, smem, st);
  return launch_chain_epi<EPI_SWIGLU>(c, smem, st);
}

}  // namespace sd
