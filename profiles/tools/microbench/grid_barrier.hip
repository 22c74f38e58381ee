// How much does a device-wide barrier cost on MI355X when every CU holds one 1024-thread
// workgroup (the GEMV shape)? Compared against the floor of back-to-back kernel launches
// (~4.5 us per dependent launch in the bench trace). Build: hipcc --offload-arch=gfx950 -O3
// grid_barrier.hip -o grid_barrier ; run: ./grid_barrier
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

// monotonic counter barrier: phase p completes when counter >= (p+1)*nwg
template <int MODE>
__global__ __launch_bounds__(1024) void barrier_kernel(unsigned* counter, float* data, int phases, int payload) {
  const unsigned nwg = gridDim.x;
  float acc = 0.f;
  for (int p = 0; p < phases; ++p) {
    // a little "work": each WG writes a value every other WG reads after the barrier
    if (payload && threadIdx.x < 64) data[(p & 1) * nwg * 64 + blockIdx.x * 64 + threadIdx.x] = acc * 0.5f + p;
    __syncthreads();
    if (threadIdx.x == 0) {
      if (MODE == 0) {
        __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned target = (p + 1) * nwg;
        while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {}
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      } else if (MODE == 1) {
        __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned target = (p + 1) * nwg;
        while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(2);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      } else if (MODE == 3) {
        // two levels: 16 group counters 1 KiB apart (different channels), the last arrival of a group bumps the top
        unsigned* grp = counter + 256 * (1 + (blockIdx.x & 15));
        const unsigned per = (nwg + 15) / 16;   // nwg is a multiple of 16 here
        const unsigned old = __hip_atomic_fetch_add(grp, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        if (old == (p + 1) * per - 1) __hip_atomic_fetch_add(counter, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned target = (p + 1) * 16;
        while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {}
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      }
    }
    if (MODE == 2) {
      // no read-modify-write at all: every workgroup publishes its phase in its own slot, lane l of wave 0 of every
      // workgroup polls slots 4l..4l+3 (one 1-KiB wave load covers 256 slots)
      unsigned* flags = counter + 256;
      if (threadIdx.x == 0) __hip_atomic_store(flags + blockIdx.x, (unsigned)(p + 1), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      if (threadIdx.x < 64) {
        const unsigned base = threadIdx.x * 4;
        for (;;) {
          bool ok = true;
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if (base + j < nwg) ok = ok && (__hip_atomic_load(flags + base + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= (unsigned)(p + 1));
          if (__all(ok)) break;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      }
    }
    __syncthreads();
    if (payload) {
      const unsigned src = (blockIdx.x + 97) % nwg;   // another WG's (likely another XCD's) data
      if (threadIdx.x < 64) acc += data[(p & 1) * nwg * 64 + src * 64 + threadIdx.x];
    }
  }
  if (payload && threadIdx.x < 64) data[2 * nwg * 64 + blockIdx.x * 64 + threadIdx.x] = acc;
}

// MODE 4/5: no agent-scope fences at all (no L2 write-back / invalidate): the exchanged data itself moves with
// agent-scope RELAXED atomics (stores write through to memory, loads bypass the XCD's L2); ordering comes from
// waiting for the stores (workgroup-scope release = s_waitcnt) before the arrival and from the control dependency
// on the poll. 4 = single counter, 5 = two-level counters.
template <int MODE>
__global__ __launch_bounds__(1024) void barrier_nofence_kernel(unsigned* counter, float* data, int phases, int payload) {
  const unsigned nwg = gridDim.x;
  float acc = 0.f;
  for (int p = 0; p < phases; ++p) {
    if (payload && threadIdx.x < 64)
      __hip_atomic_store(&data[(p & 1) * nwg * 64 + blockIdx.x * 64 + threadIdx.x], (float)(p * 512 + (int)blockIdx.x), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __syncthreads();
    if (threadIdx.x == 0) {
      if (MODE == 4) {
        __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned target = (p + 1) * nwg;
        while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {}
      } else {
        unsigned* grp = counter + 256 * (1 + (blockIdx.x & 15));
        const unsigned per = (nwg + 15) / 16;
        const unsigned old = __hip_atomic_fetch_add(grp, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (old == (p + 1) * per - 1) __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned target = (p + 1) * 16;
        while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {}
      }
    }
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    if (payload) {
      const unsigned src = (blockIdx.x + 97) % nwg;
      if (threadIdx.x < 64) {
        const float v = __hip_atomic_load(&data[(p & 1) * nwg * 64 + src * 64 + threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (v != (float)(p * 512 + (int)src)) acc += 1.f;   // a stale or torn value
      }
    }
  }
  if (payload && threadIdx.x < 64) data[2 * nwg * 64 + blockIdx.x * 64 + threadIdx.x] = acc;
}

// MODE 6: does the fence-free barrier overlap with a weight stream in flight? Per phase, waves 1..15 of every workgroup
// issue NB independent 1-KiB loads (the GEMV's weight batch) BEFORE the barrier and consume them after it; wave 0 only
// synchronises. with_barrier = 0 gives the pure stream time of the same loads.
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
template <int NB>
__global__ __launch_bounds__(1024) void barrier_stream_kernel(unsigned* counter, const u32x4* big, size_t big_vec, unsigned* sink,
                                                              int phases, int with_barrier) {
  const unsigned nwg = gridDim.x;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  unsigned acc = 0;
  for (int p = 0; p < phases; ++p) {
    u32x4 buf[NB];
    if (wave != 0) {
      size_t base = ((static_cast<size_t>(p) * nwg + blockIdx.x) * 15 + (wave - 1)) * NB * 64;
      base %= (big_vec - NB * 64);
#pragma unroll
      for (int j = 0; j < NB; ++j) buf[j] = __builtin_nontemporal_load(big + base + j * 64 + lane);
    }
    if (with_barrier) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      if (threadIdx.x == 0) {
        unsigned* grp = counter + 256 * (1 + (blockIdx.x & 15));
        const unsigned per = (nwg + 15) / 16;
        const unsigned old = __hip_atomic_fetch_add(grp, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (old == (p + 1) * per - 1) __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned target = (p + 1) * 16;
        while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {}
      }
    }
    __syncthreads();
    if (wave != 0) {
#pragma unroll
      for (int j = 0; j < NB; ++j) acc += buf[j][0] ^ buf[j][1] ^ buf[j][2] ^ buf[j][3];
    }
  }
  if (acc == 0x12345678u) sink[0] = acc;
}

// MODE 7: a PACED stream — every wave keeps only W 1-KiB loads in flight (rolling window, straight-line code so hipcc
// emits counted vmcnt waits) while wave 0 runs the fence-free barrier concurrently. Does a shallow queue (chip-wide
// 256 x 15 x W KiB in flight) still reach the stream's bandwidth, and what does the barrier cost next to it?
template <int NB, int W>
__global__ __launch_bounds__(1024) void barrier_paced_kernel(unsigned* counter, const u32x4* big, size_t big_vec, unsigned* sink,
                                                             unsigned long long* lat, int phases, int with_barrier) {
  const unsigned nwg = gridDim.x;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  unsigned acc = 0;
  unsigned long long lat_sum = 0;
  for (int p = 0; p < phases; ++p) {
    if (wave != 0) {
      size_t base = ((static_cast<size_t>(p) * nwg + blockIdx.x) * 15 + (wave - 1)) * NB * 64;
      base %= (big_vec - NB * 64);
      u32x4 buf[W];
#pragma unroll
      for (int j = 0; j < W; ++j) buf[j] = __builtin_nontemporal_load(big + base + j * 64 + lane);
#pragma unroll
      for (int j = 0; j < NB; ++j) {
        acc += buf[j % W][0] ^ buf[j % W][1] ^ buf[j % W][2] ^ buf[j % W][3];
        if (j + W < NB) buf[j % W] = __builtin_nontemporal_load(big + base + (j + W) * 64 + lane);
      }
    } else if (with_barrier && threadIdx.x == 0) {
      const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
      unsigned* grp = counter + 256 * (1 + (blockIdx.x & 15));
      const unsigned per = (nwg + 15) / 16;
      const unsigned old = __hip_atomic_fetch_add(grp, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (old == (p + 1) * per - 1) __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned target = (p + 1) * 16;
      while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {}
      lat_sum += __builtin_amdgcn_s_memrealtime() - t0;
    }
    __syncthreads();
  }
  if (acc == 0x12345678u) sink[0] = acc;
  if (with_barrier && threadIdx.x == 0) lat[blockIdx.x] = lat_sum;
}

__global__ void empty_kernel(float* d) { if (d == nullptr) *d = 0; }

int main() {
  int dev = 0; CK(hipSetDevice(dev));
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, dev));
  const int ncu = prop.multiProcessorCount;
  printf("device %s, %d CUs\n", prop.name, ncu);
  unsigned* counter; float* data;
  CK(hipMalloc(&counter, 32768)); CK(hipMalloc(&data, 3 * ncu * 64 * 4));
  hipStream_t st; CK(hipStreamCreate(&st));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int phases = 2000;   // p * 512 + wg stays exact in fp32
  for (int mode = 0; mode < 6; ++mode)
    for (int payload = 0; payload < 2; ++payload)
      for (int rep = 0; rep < 2; ++rep) {
        CK(hipMemsetAsync(counter, 0, 32768, st));
        CK(hipMemsetAsync(data, 0, 3 * ncu * 64 * 4, st));
        CK(hipEventRecord(e0, st));
        void* args[] = {&counter, &data, (void*)&phases, &payload};
        // cooperative launch: co-residency of all workgroups is guaranteed or the launch fails
        void* fn = mode == 0 ? (void*)barrier_kernel<0> : mode == 1 ? (void*)barrier_kernel<1> : mode == 2 ? (void*)barrier_kernel<2> : mode == 3 ? (void*)barrier_kernel<3> : mode == 4 ? (void*)barrier_nofence_kernel<4> : (void*)barrier_nofence_kernel<5>;
        CK(hipLaunchCooperativeKernel(fn, dim3(ncu), dim3(1024), args, 0, st));
        CK(hipEventRecord(e1, st));
        CK(hipStreamSynchronize(st));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep) printf("mode %d (%s) payload %d: %.3f us per barrier\n", mode, mode == 0 ? "spin" : mode == 1 ? "sleep" : mode == 2 ? "flag per workgroup" : mode == 3 ? "two-level counters" : mode == 4 ? "no fences, one counter" : "no fences, two-level counters", payload, ms * 1e3 / phases);
        if (payload && rep) {
          std::vector<float> h(ncu * 64);
          CK(hipMemcpy(h.data(), data + 2 * ncu * 64, ncu * 64 * 4, hipMemcpyDeviceToHost));
          // every phase p adds (acc_src + p); all WGs symmetric -> acc identical everywhere
          bool same = true; for (int i = 1; i < ncu * 64; ++i) same &= (h[i] == h[0]);
          if (mode >= 4) { double bad = 0; for (int i = 0; i < ncu * 64; ++i) bad += h[i]; printf("   exact payload check: %g stale values in %d phases\n", bad, phases); }
          else printf("   payload check: %s (acc=%g)\n", same ? "consistent" : "MISMATCH", h[0]);
        }
      }
  {
    const size_t big_bytes = 2ull << 30;
    u32x4* big; unsigned* sink;
    CK(hipMalloc(&big, big_bytes)); CK(hipMalloc(&sink, 4));
    CK(hipMemset(big, 1, big_bytes));
    const size_t big_vec = big_bytes / 16;
    const int ph = 400;
    for (int nb = 6; nb <= 12; nb += 6)
      for (int wb = 0; wb < 2; ++wb)
        for (int rep = 0; rep < 2; ++rep) {
          CK(hipMemsetAsync(counter, 0, 32768, st));
          CK(hipEventRecord(e0, st));
          void* args[] = {&counter, &big, (void*)&big_vec, &sink, (void*)&ph, &wb};
          void* fn = nb == 6 ? (void*)barrier_stream_kernel<6> : (void*)barrier_stream_kernel<12>;
          CK(hipLaunchCooperativeKernel(fn, dim3(ncu), dim3(1024), args, 0, st));
          CK(hipEventRecord(e1, st));
          CK(hipStreamSynchronize(st));
          float ms; CK(hipEventElapsedTime(&ms, e0, e1));
          const double mb = (double)ncu * 15 * nb * 1024 / 1e6;
          if (rep) printf("stream of %d x 1 KiB per wave (%.1f MB per phase) %s: %.3f us per phase = %.2f TB/s\n", nb, mb,
                          wb ? "+ fence-free barrier" : "alone (workgroup barrier only)", ms * 1e3 / ph, mb / (ms * 1e3 / ph));
        }
  }
  {
    const size_t big_bytes = 2ull << 30;
    u32x4* big; unsigned* sink; unsigned long long* lat;
    CK(hipMalloc(&big, big_bytes)); CK(hipMalloc(&sink, 4)); CK(hipMalloc(&lat, ncu * 8));
    CK(hipMemset(big, 1, big_bytes));
    const size_t big_vec = big_bytes / 16;
    const int ph = 400;
    const int Ws[6] = {1, 2, 3, 4, 6, 12};
    for (int wi = 0; wi < 6; ++wi)
      for (int wb = 0; wb < 2; ++wb)
        for (int rep = 0; rep < 2; ++rep) {
          CK(hipMemsetAsync(counter, 0, 32768, st));
          CK(hipMemsetAsync(lat, 0, ncu * 8, st));
          CK(hipEventRecord(e0, st));
          void* args[] = {&counter, &big, (void*)&big_vec, &sink, &lat, (void*)&ph, &wb};
          void* fn = nullptr;
          switch (Ws[wi]) {
            case 1: fn = (void*)barrier_paced_kernel<12, 1>; break;
            case 2: fn = (void*)barrier_paced_kernel<12, 2>; break;
            case 3: fn = (void*)barrier_paced_kernel<12, 3>; break;
            case 4: fn = (void*)barrier_paced_kernel<12, 4>; break;
            case 6: fn = (void*)barrier_paced_kernel<12, 6>; break;
            default: fn = (void*)barrier_paced_kernel<12, 12>; break;
          }
          CK(hipLaunchCooperativeKernel(fn, dim3(ncu), dim3(1024), args, 0, st));
          CK(hipEventRecord(e1, st));
          CK(hipStreamSynchronize(st));
          float ms; CK(hipEventElapsedTime(&ms, e0, e1));
          const double mb = (double)ncu * 15 * 12 * 1024 / 1e6;
          if (rep) {
            double bl = 0;
            if (wb) {
              std::vector<unsigned long long> h(ncu);
              CK(hipMemcpy(h.data(), lat, ncu * 8, hipMemcpyDeviceToHost));
              for (int i = 0; i < ncu; ++i) bl += (double)h[i];
              bl = bl / ncu / ph / 100.0;   // 100 MHz ticks -> us
            }
            printf("paced stream, window %2d loads per wave (%.1f MB in flight chip-wide, %.1f MB per phase) %s: %.3f us per phase = %.2f TB/s",
                   Ws[wi], (double)ncu * 15 * Ws[wi] * 1024 / 1e6, mb, wb ? "+ concurrent barrier" : "alone", ms * 1e3 / ph, mb / (ms * 1e3 / ph));
            if (wb) printf(", barrier latency seen by wave 0: %.2f us", bl);
            printf("\n");
          }
        }
  }
  // floor of dependent kernel launches in a captured graph
  {
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < 500; ++i) hipLaunchKernelGGL(empty_kernel, dim3(ncu), dim3(1024), 0, st, data);
    CK(hipStreamEndCapture(st, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int rep = 0; rep < 3; ++rep) {
      CK(hipEventRecord(e0, st));
      CK(hipGraphLaunch(ge, st));
      CK(hipEventRecord(e1, st));
      CK(hipStreamSynchronize(st));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (rep) printf("graph of 500 empty 1024-thread x %d-WG kernels: %.3f us per launch\n", ncu, ms * 1e3 / 500);
    }
  }
  return 0;
}
