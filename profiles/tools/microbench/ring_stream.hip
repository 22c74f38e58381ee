// Microbenchmark: one persistent workgroup per CU, 1 LDS-DMA loader wave + 3 MFMA consumer waves, a byte ring in LDS.
// Measures what the persistent decode engine (csrc/persist.hip) is built on: the rate at which one loader wave per CU
// streams a contiguous weight stream through LDS while the consumers retire it with v_mfma_f32_16x16x32_bf16.
//   hipcc --offload-arch=gfx950 -O3 ring_stream.hip -o ring_stream && ./ring_stream [MiB per CU] [iters]
// The stream of CU c is [tile][step][g][row] 1-KiB MFMA A fragments (8 row pairs per tile, K = 2048), the layout of
// csrc/pack.hip; y = W x is checked against the host for a few tiles.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int kK = 2048, kSteps = kK / 32;       // 64 steps of 1 KiB per tile
constexpr int kPiece = 1024;
constexpr unsigned kTimeoutTicks = 20u * 1000u * 100u;   // 20 ms of the 100 MHz clock

struct Ctl {            // LDS control words
  unsigned landed;      // pieces landed (loader)
  unsigned consumed[3]; // first piece each consumer still needs
  unsigned abort;
  unsigned done[3];     // tiles whose partial a consumer has written
  unsigned lead_done;   // tiles the leader has finished
};

template <bool NT>
__device__ __forceinline__ void dma_piece(const char* sbase, unsigned voff, unsigned lds_dst) {
  if constexpr (NT)
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 nt" ::"v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
  else
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}

__device__ __forceinline__ unsigned lds_load(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void lds_store(unsigned* p, unsigned v) { __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP); }

// F = slots (16 pieces) the loader keeps in flight
template <int F, bool NT>
__global__ __launch_bounds__(256) void ring_kernel(const char* __restrict__ W, size_t bytes_per_cu, const uint16_t* __restrict__ x,
                                                   float* __restrict__ out, unsigned ring_pieces, unsigned* status, int consume) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  Ctl* ctl = reinterpret_cast<Ctl*>(smem);
  uint16_t* U = reinterpret_cast<uint16_t*>(smem + 256);                  // x row, 4 KiB
  float* part = reinterpret_cast<float*>(smem + 256 + 4096);              // [2][3][16] f32 (column 0 only)
  unsigned char* ring = smem + 256 + 4096 + 512;                          // ring_pieces KiB
  const unsigned ring_base = 256 + 4096 + 512;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  if (tid < 16) reinterpret_cast<unsigned*>(ctl)[tid] = 0;
  for (int i = tid; i < kK / 8; i += 256) reinterpret_cast<u32x4*>(U)[i] = reinterpret_cast<const u32x4*>(x)[i];
  __syncthreads();
  const unsigned total = static_cast<unsigned>(bytes_per_cu / kPiece);
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  auto timed_out = [&]() { return static_cast<unsigned>(__builtin_amdgcn_s_memrealtime() - t0) > kTimeoutTicks; };

  if (wave == 0) {
    // ---------------- loader
    const char* src = W + static_cast<size_t>(blockIdx.x) * bytes_per_cu;
    const unsigned voff = lane * 16;
    unsigned issued = 0, rpos = 0;   // pieces issued; ring position (pieces)
    while (issued < total) {
      const unsigned n = min(16u, total - issued);
      for (unsigned spins = 0;; ++spins) {
        const unsigned c = min(lds_load(&ctl->consumed[0]), min(lds_load(&ctl->consumed[1]), lds_load(&ctl->consumed[2])));
        if (!consume || issued + n - c <= ring_pieces) break;
        __builtin_amdgcn_s_sleep(2);
        if ((spins & 63) == 63 && (lds_load(&ctl->abort) || timed_out())) { lds_store(&ctl->abort, 1u); if (lane == 0) atomicOr(status, 1u); return; }
      }
      if (n == 16) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
          dma_piece<NT>(src, voff, ring_base + rpos * kPiece);
          src += kPiece;
          rpos = (rpos + 1 == ring_pieces) ? 0 : rpos + 1;
        }
        issued += 16;
        if constexpr (F == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if constexpr (F == 2) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        if constexpr (F == 3) asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
        if constexpr (F == 4) asm volatile("s_waitcnt vmcnt(48)" ::: "memory");
        if (issued > 16u * (F - 1)) lds_store(&ctl->landed, issued - 16u * (F - 1));
      } else {
        for (unsigned j = 0; j < n; ++j) {
          dma_piece<NT>(src, voff, ring_base + rpos * kPiece);
          src += kPiece;
          rpos = (rpos + 1 == ring_pieces) ? 0 : rpos + 1;
        }
        issued += n;
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    lds_store(&ctl->landed, total);
    return;
  }
  if (!consume) return;
  // ---------------- consumers
  const int w = wave - 1;
  const int g = lane >> 4;
  const unsigned n_tiles = total / kSteps;
  unsigned landed = 0;
  for (unsigned tile = 0; tile < n_tiles; ++tile) {
    f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
    for (int grp = w; grp < kSteps / 4; grp += 3) {
      const unsigned p0 = tile * kSteps + grp * 4;
      if (landed < p0 + 4) {
        for (unsigned spins = 0;; ++spins) {
          landed = lds_load(&ctl->landed);
          if (landed >= p0 + 4) break;
          __builtin_amdgcn_s_sleep(1);
          if ((spins & 63) == 63 && (lds_load(&ctl->abort) || timed_out())) { lds_store(&ctl->abort, 1u); if (lane == 0) atomicOr(status, 2u); return; }
        }
      }
      unsigned rp = p0 % ring_pieces;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const u32x4 a = *reinterpret_cast<const u32x4*>(ring + rp * kPiece + lane * 16);
        const u32x4 b = *reinterpret_cast<const u32x4*>(U + (grp * 4 + j) * 32 + g * 8);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), acc, 0, 0, 0);
        rp = (rp + 1 == ring_pieces) ? 0 : rp + 1;
      }
      // next piece this wave needs: its next group (this tile or the next)
      const unsigned nxt = (grp + 3 < kSteps / 4) ? p0 + 12 : (tile + 1) * kSteps + w * 4;
      lds_store(&ctl->consumed[w], nxt);
    }
    // partial (column 0: lanes n == 0 hold rows 4g..4g+3) -> LDS, double-buffered by tile parity
    if (tile >= 2) {
      for (unsigned spins = 0; lds_load(&ctl->lead_done) + 1 < tile; ++spins) {
        __builtin_amdgcn_s_sleep(1);
        if ((spins & 63) == 63 && (lds_load(&ctl->abort) || timed_out())) { lds_store(&ctl->abort, 1u); if (lane == 0) atomicOr(status, 4u); return; }
      }
    }
    float* pb = part + ((tile & 1) * 3 + w) * 16;
    if ((lane & 15) == 0) {
#pragma unroll
      for (int q = 0; q < 4; ++q) pb[4 * g + q] = acc[q];
    }
    lds_store(&ctl->done[w], tile + 1);
    if (w == 0) {   // leader: fold the three partials
      for (unsigned spins = 0; lds_load(&ctl->done[1]) < tile + 1 || lds_load(&ctl->done[2]) < tile + 1; ++spins) {
        __builtin_amdgcn_s_sleep(1);
        if ((spins & 63) == 63 && (lds_load(&ctl->abort) || timed_out())) { lds_store(&ctl->abort, 1u); if (lane == 0) atomicOr(status, 8u); return; }
      }
      if (lane < 16) {
        const float* p = part + (tile & 1) * 48;
        out[(static_cast<size_t>(blockIdx.x) * n_tiles + tile) * 16 + lane] = p[lane] + p[16 + lane] + p[32 + lane];
      }
      lds_store(&ctl->lead_done, tile + 1);
    }
  }
}

static uint16_t f2bf(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7fff + ((u >> 16) & 1); return static_cast<uint16_t>(u >> 16); }
static float bf2f(uint16_t b) { uint32_t u = static_cast<uint32_t>(b) << 16; float f; memcpy(&f, &u, 4); return f; }

template <int F, bool NT>
static void run(const char* name, const char* W, size_t bytes_per_cu, const uint16_t* x, float* out, unsigned ring_pieces, unsigned* status,
                int consume, int iters, int nbuf, size_t buf_stride) {
  const size_t smem = 256 + 4096 + 512 + static_cast<size_t>(ring_pieces) * kPiece;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&ring_kernel<F, NT>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((ring_kernel<F, NT>), dim3(256), dim3(256), smem, 0, W + (i % nbuf) * buf_stride, bytes_per_cu, x, out, ring_pieces, status, consume);
  CK(hipEventRecord(e0, 0));
  for (int i = 0; i < iters; ++i) hipLaunchKernelGGL((ring_kernel<F, NT>), dim3(256), dim3(256), smem, 0, W + (i % nbuf) * buf_stride, bytes_per_cu, x, out, ring_pieces, status, consume);
  CK(hipEventRecord(e1, 0));
  CK(hipEventSynchronize(e1));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  unsigned st = 0;
  CK(hipMemcpy(&st, status, 4, hipMemcpyDeviceToHost));
  const double us = ms * 1000.0 / iters, gb = 256.0 * bytes_per_cu / 1e9;
  printf("%-34s ring %3u KiB  %8.2f us/launch  %6.2f TB/s  status %u\n", name, ring_pieces, us, gb / us * 1e6 / 1e3, st);
  fflush(stdout);
}

int main(int argc, char** argv) {
  const size_t mib = argc > 1 ? atoi(argv[1]) : 2;
  const int iters = argc > 2 ? atoi(argv[2]) : 20;
  const size_t bytes_per_cu = mib << 20;
  const int nbuf = 3;
  const size_t buf_stride = 256 * bytes_per_cu;
  std::vector<uint16_t> hw(buf_stride / 2 * nbuf), hx(kK);
  srand(1);
  for (auto& v : hw) v = f2bf((rand() % 2001 - 1000) / 4000.0f);
  for (auto& v : hx) v = f2bf((rand() % 2001 - 1000) / 1000.0f);
  char* W;
  uint16_t* x;
  float* out;
  unsigned* status;
  CK(hipMalloc(&W, buf_stride * nbuf));
  CK(hipMalloc(&x, kK * 2));
  const size_t n_tiles = bytes_per_cu / kPiece / kSteps;
  CK(hipMalloc(&out, 256 * n_tiles * 16 * 4));
  CK(hipMalloc(&status, 4));
  CK(hipMemset(status, 0, 4));
  CK(hipMemcpy(W, hw.data(), buf_stride * nbuf, hipMemcpyHostToDevice));
  CK(hipMemcpy(x, hx.data(), kK * 2, hipMemcpyHostToDevice));

  // correctness first (F = 3, nt): buffer 0, every 37th tile of every 16th CU
  {
    const unsigned rp = 128;
    const size_t smem = 256 + 4096 + 512 + rp * kPiece;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&ring_kernel<3, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipLaunchKernelGGL((ring_kernel<3, true>), dim3(256), dim3(256), smem, 0, W, bytes_per_cu, x, out, rp, status, 1);
    CK(hipDeviceSynchronize());
    std::vector<float> ho(256 * n_tiles * 16);
    CK(hipMemcpy(ho.data(), out, ho.size() * 4, hipMemcpyDeviceToHost));
    double maxerr = 0;
    for (int c = 0; c < 256; c += 16)
      for (size_t t = 0; t < n_tiles; t += 37)
        for (int row = 0; row < 16; ++row) {
          double ref = 0;
          for (int s = 0; s < kSteps; ++s)
            for (int g = 0; g < 4; ++g)
              for (int e = 0; e < 8; ++e) {
                const size_t off = (static_cast<size_t>(c) * bytes_per_cu + (t * kSteps + s) * kPiece + (g * 16 + row) * 16) / 2 + e;
                ref += static_cast<double>(bf2f(hw[off])) * bf2f(hx[s * 32 + g * 8 + e]);
              }
          const double err = fabs(ref - ho[(c * n_tiles + t) * 16 + row]);
          if (err > maxerr) maxerr = err;
        }
    unsigned st = 0;
    CK(hipMemcpy(&st, status, 4, hipMemcpyDeviceToHost));
    printf("check: max |err| = %.3g (status %u)\n", maxerr, st);
    if (maxerr > 1e-2 || st) { printf("FAILED\n"); return 1; }
  }
  printf("%zu MiB per CU, %d iters, %d buffers\n", mib, iters, nbuf);
  run<3, true>("F=3 nt consume", W, bytes_per_cu, x, out, 128, status, 1, iters, nbuf, buf_stride);
  run<3, false>("F=3 default consume", W, bytes_per_cu, x, out, 128, status, 1, iters, nbuf, buf_stride);
  run<2, true>("F=2 nt consume", W, bytes_per_cu, x, out, 128, status, 1, iters, nbuf, buf_stride);
  run<4, true>("F=4 nt consume (vmcnt 48)", W, bytes_per_cu, x, out, 128, status, 1, iters, nbuf, buf_stride);
  run<1, true>("F=1 nt consume", W, bytes_per_cu, x, out, 128, status, 1, iters, nbuf, buf_stride);
  run<3, true>("F=3 nt consume ring 64", W, bytes_per_cu, x, out, 64, status, 1, iters, nbuf, buf_stride);
  run<3, true>("F=3 nt consume ring 112", W, bytes_per_cu, x, out, 112, status, 1, iters, nbuf, buf_stride);
  run<3, true>("F=3 nt loader only (floor)", W, bytes_per_cu, x, out, 128, status, 0, iters, nbuf, buf_stride);
  run<4, true>("F=4 nt loader only (floor)", W, bytes_per_cu, x, out, 128, status, 0, iters, nbuf, buf_stride);
  return 0;
}
