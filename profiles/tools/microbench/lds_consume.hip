// Microbenchmark: how fast can ONE wave per SIMD retire weight fragments that are already in LDS?
// (The persistent engine's consumers ran at ~9 KiB/us per wave; this isolates the loop from the loader and the hand-offs.)
//   hipcc --offload-arch=gfx950 -O3 lds_consume.hip -o lds_consume && ./lds_consume
// One workgroup per CU, 4 waves (wave 0 idle or running the LDS-DMA loader, waves 1-3 consume a 128 KiB region
// round-robin in 8-step chunks, `reps` times). Variants: step bytes (np), B operand from LDS or registers, MFMA on/off,
// reads batched behind a scheduling barrier or left to the compiler, loader live or not.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int kRing = 128 * 1024, kU = 16 * 1024 + 16;

__device__ __forceinline__ void dma_piece(const char* sbase, unsigned voff, unsigned lds_dst) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 nt" ::"v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}

// MODE bit 0: B from LDS (else registers), bit 1: MFMA (else xor-fold), bit 2: sched_barrier batching, bit 3: loader live
template <int MODE, int CH>
__global__ __launch_bounds__(256) void consume_kernel(const char* __restrict__ W, int np, int reps, float* out, unsigned long long* cyc) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* U = smem;
  unsigned char* ring = smem + kU;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  for (int i = tid; i < (kU + kRing) / 16; i += 256) reinterpret_cast<u32x4*>(smem)[i] = u32x4{0x3f803f80u + i, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
  __syncthreads();
  if (wave == 0) {
    if constexpr (MODE & 8) {   // live loader: keeps streaming into the ring (overwrites what the consumers read: timing only)
      const char* src = W + static_cast<size_t>(blockIdx.x) * (4u << 20);
      unsigned rpos = 0;
      for (int s = 0; s < reps * 8; ++s) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
          dma_piece(src, lane * 16, kU + rpos);
          src += 1024;
          rpos = (rpos + 1024) & (kRing - 1);
        }
        asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
        if ((s & 255) == 255) src = W + static_cast<size_t>(blockIdx.x) * (4u << 20);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    return;
  }
  const int cw = wave - 1, g = lane >> 4, n = lane & 15;
  const unsigned sb = np * 128u;
  int jp = n & 7, second = n >> 3;
  if (jp >= np) { jp = 0; second = 0; }
  const unsigned lane_off = (g * 2 * np + second * np + jp) * 16u;
  const int n_chunk = kRing / (CH * sb);
  f32x4_t acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
  u32x4 fold = {0, 0, 0, 0};
  const unsigned char* xrow = U + g * 16;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int r = 0; r < reps; ++r) {
    for (int ch = cw; ch < n_chunk; ch += 3) {
      const unsigned char* wb = ring + ch * CH * sb + lane_off;
      const unsigned char* xb = xrow + (ch & 31) * (CH * 64);
      u32x4 wf[CH], xf[CH];
#pragma unroll
      for (int j = 0; j < CH; ++j) {
        wf[j] = *reinterpret_cast<const u32x4*>(wb + j * sb);
        if constexpr (MODE & 1) xf[j] = *reinterpret_cast<const u32x4*>(xb + j * 64);
        else xf[j] = u32x4{0x3f803f80u, 0x3f803f80u + j, 0x3f803f80u, 0x3f803f80u};
      }
      if constexpr (MODE & 4) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < CH; j += 2) {
        if constexpr (MODE & 2) {
          acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, wf[j]), __builtin_bit_cast(bf16x8_t, xf[j]), acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, wf[j + 1]), __builtin_bit_cast(bf16x8_t, xf[j + 1]), acc1, 0, 0, 0);
        } else {
          fold ^= wf[j] ^ xf[j] ^ wf[j + 1] ^ xf[j + 1];
        }
      }
      if constexpr (MODE & 4) __builtin_amdgcn_sched_barrier(0);
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (lane == 0) cyc[blockIdx.x * 3 + cw] = t1 - t0;
  out[(blockIdx.x * 3 + cw) * 64 + lane] = acc0[0] + acc1[1] + __uint_as_float(fold[0] ^ fold[3]);
}

template <int MODE, int CH>
static void run(const char* name, const char* W, int np, int reps, float* out, unsigned long long* cyc) {
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&consume_kernel<MODE, CH>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  hipLaunchKernelGGL((consume_kernel<MODE, CH>), dim3(256), dim3(256), kU + kRing, 0, W, np, reps, out, cyc);
  CK(hipDeviceSynchronize());
  unsigned long long h[768];
  CK(hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost));
  double sum = 0;
  for (auto v : h) sum += v;
  const double per_wave = sum / 768.0;
  const double bytes_per_wave = static_cast<double>(reps) * kRing / 3.0;
  printf("%-52s np=%d CH=%d: %7.0f cycles per KiB-step... %6.2f B/cycle per wave, %6.1f cycles per %d-step chunk\n", name, np, CH,
         per_wave / (bytes_per_wave / (np * 128.0)), bytes_per_wave / per_wave, per_wave / (bytes_per_wave / (CH * np * 128.0)), CH);
  fflush(stdout);
}

int main() {
  char* W;
  float* out;
  unsigned long long* cyc;
  CK(hipMalloc(&W, 256ull * (4u << 20) + (1u << 20)));
  CK(hipMemset(W, 0x3f, 256ull * (4u << 20) + (1u << 20)));
  CK(hipMalloc(&out, 768 * 64 * 4));
  CK(hipMalloc(&cyc, 768 * 8));
  const int reps = 64;
  for (int np : {8, 4}) {
    run<1 | 2, 8>("B from LDS, MFMA, compiler order", W, np, reps, out, cyc);
    run<1 | 2 | 4, 8>("B from LDS, MFMA, batched reads", W, np, reps, out, cyc);
    run<2 | 4, 8>("B in registers, MFMA, batched reads", W, np, reps, out, cyc);
    run<1 | 4, 8>("B from LDS, no MFMA (xor), batched reads", W, np, reps, out, cyc);
    run<4, 8>("A reads only (xor), batched", W, np, reps, out, cyc);
    run<1 | 2 | 4, 4>("B from LDS, MFMA, batched reads", W, np, reps, out, cyc);
    run<1 | 2 | 4 | 8, 8>("B from LDS, MFMA, batched reads, LIVE LOADER", W, np, reps, out, cyc);
    run<2 | 4 | 8, 8>("B in registers, MFMA, batched, LIVE LOADER", W, np, reps, out, cyc);
  }
  return 0;
}
