// What a MINIMAL dependent weight-streaming launch costs on this chip: the floor a launch-path GEMV can be held against.
// Each launch: every workgroup (one per CU) first issues the loads of x (T rows x K bf16, written by the previous launch),
// then streams its contiguous share of a weight matrix with 16-byte non-temporal loads (DEPTH loads per thread in flight),
// stages x in LDS behind a barrier, folds weights and x with xors (no MFMA: the arithmetic is not the subject), reduces over
// the workgroup and writes T x 8 bytes of "output" into the next launch's x. A chain of such launches over DIFFERENT weight
// buffers (a pool larger than the Infinity Cache) is captured into a hipGraph; us per launch = graph time / launches.
//   hipcc -O3 --offload-arch=gfx950 profiles/tools/microbench/stream_floor.hip -o gpurun_out/stream_floor && gpurun_out/stream_floor
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int THREADS, int DEPTH>
__global__ __launch_bounds__(THREADS) void stream_hop(const u32x4* __restrict__ w, size_t share_vec, const u32x4* __restrict__ x_in, int x_vec,
                                                       unsigned* __restrict__ x_out, unsigned rot_vec, unsigned long long* t_end) {
  extern __shared__ u32x4 xs[];
  const int tid = threadIdx.x;
  // x rows of the previous launch: issued first, so that they return first
  u32x4 xr[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) { const int i = tid + j * THREADS; xr[j] = x_in[i < x_vec ? i : x_vec - 1]; }
  const u32x4* p = w + static_cast<size_t>(blockIdx.x) * share_vec;
  // rot_vec != 0: workgroup b walks its share starting rot_vec * b vectors in (wrapping), so that the workgroups do not sit at the
  // same offset of their equally sized, equally spaced shares at the same time
  const size_t rot = rot_vec ? (static_cast<size_t>(blockIdx.x) * rot_vec) % share_vec : 0;
  auto at = [&](size_t k) -> const u32x4* { size_t q = k + rot; if (q >= share_vec) q -= share_vec; return p + q; };
  u32x4 acc = {0u, 0u, 0u, 0u};
  u32x4 buf[DEPTH];
  size_t i = tid;
#pragma unroll
  for (int j = 0; j < DEPTH; ++j) buf[j] = __builtin_nontemporal_load(at(i + static_cast<size_t>(j) * THREADS < share_vec ? i + static_cast<size_t>(j) * THREADS : share_vec - 1));
#pragma unroll
  for (int j = 0; j < 4; ++j) { const int k = tid + j * THREADS; if (k < x_vec) xs[k] = xr[j]; }
  __syncthreads();
  for (; i < share_vec; i += static_cast<size_t>(DEPTH) * THREADS) {
#pragma unroll
    for (int j = 0; j < DEPTH; ++j) {
      const size_t k = i + static_cast<size_t>(j) * THREADS;
      acc ^= buf[j] & xs[(k + j) % x_vec];
      const size_t nk = k + static_cast<size_t>(DEPTH) * THREADS;
      buf[j] = __builtin_nontemporal_load(at(nk < share_vec ? nk : share_vec - 1));
    }
  }
  unsigned v = acc.x ^ acc.y ^ acc.z ^ acc.w;
  for (int off = 32; off > 0; off >>= 1) v ^= __shfl_xor(v, off, 64);
  __shared__ unsigned red[16];
  if ((tid & 63) == 0) red[tid >> 6] = v;
  __syncthreads();
  if (tid < 16) {   // the "output": 16 dwords per workgroup into the next launch's x (4096 dwords = 16 KiB for 256 workgroups)
    unsigned r = 0;
    for (int k = 0; k < THREADS / 64; ++k) r ^= red[k];
    x_out[blockIdx.x * 16 + tid] = (r & 0x3f803f80u) | tid;   // (keeps the values small: they are bf16 pairs to the next launch)
  }
  if (t_end && tid == 0) t_end[blockIdx.x] = __builtin_amdgcn_s_memrealtime();
}

template <int THREADS, int DEPTH>
static float run(size_t bytes, int n, int reps, const char* pool, size_t pool_bytes, unsigned* xa, unsigned* xb, int x_bytes, hipStream_t st,
                 unsigned rot_vec = 0, unsigned long long* t_end = nullptr, float* spread_us = nullptr) {
  const int grid = 256;
  const size_t share_vec = bytes / 16 / grid;
  const size_t nbuf = pool_bytes / bytes;
  hipGraph_t g; hipGraphExec_t ge;
  CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
  for (int i = 0; i < n; ++i)
    hipLaunchKernelGGL((stream_hop<THREADS, DEPTH>), dim3(grid), dim3(THREADS), x_bytes, st, reinterpret_cast<const u32x4*>(pool + (i % nbuf) * bytes), share_vec,
                       reinterpret_cast<const u32x4*>((i & 1) ? xb : xa), x_bytes / 16, (i & 1) ? xa : xb, rot_vec, t_end);
  CK(hipStreamEndCapture(st, &g));
  CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  CK(hipGraphLaunch(ge, st)); CK(hipStreamSynchronize(st));
  float best = 1e9f;
  for (int r = 0; r < reps; ++r) {
    CK(hipEventRecord(e0, st)); CK(hipGraphLaunch(ge, st)); CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
  }
  CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
  if (t_end && spread_us) {   // end stamps of the LAST launch of the last replay (100 MHz): last workgroup - first workgroup
    unsigned long long h[256];
    CK(hipMemcpy(h, t_end, sizeof(h), hipMemcpyDeviceToHost));
    unsigned long long lo = h[0], hi = h[0];
    for (int i = 1; i < 256; ++i) { lo = h[i] < lo ? h[i] : lo; hi = h[i] > hi ? h[i] : hi; }
    *spread_us = static_cast<float>(hi - lo) / 100.f;
  }
  return best * 1000.f / n;
}

int main() {
  hipStream_t st; CK(hipStreamCreate(&st));
  const size_t pool_bytes = 3ull << 30;
  char* pool; CK(hipMalloc(&pool, pool_bytes)); CK(hipMemset(pool, 0x3c, pool_bytes));
  unsigned *xa, *xb; CK(hipMalloc(&xa, 1 << 20)); CK(hipMalloc(&xb, 1 << 20)); CK(hipMemset(xa, 0x3c, 1 << 20)); CK(hipMemset(xb, 0x3c, 1 << 20));
  const int n = 112, reps = 8;
  const int x_bytes = 5 * 3072 * 2;   // five 3072-wide bf16 rows (the 3B verify pass)
  printf("us per launch, chain of %d dependent streaming launches (256 workgroups, x = %d bytes staged in LDS), min of %d replays\n", n, x_bytes, reps);
  printf("%-26s %10s %10s %10s %10s   (bytes / 6.9 TB/s: 2.74 / 4.57 / 7.29 / 14.59 us)\n", "threads x loads in flight", "18.9 MB", "31.5 MB", "50.3 MB", "100.7 MB");
  const size_t sizes[4] = {18874368, 31457280, 50331648, 100663296};
  for (int cfg = 0; cfg < 6; ++cfg) {
    const char* names[6] = {"256 x 8", "256 x 16", "512 x 8", "512 x 12", "1024 x 6", "1024 x 12"};
    printf("%-26s ", names[cfg]);
    for (size_t b : sizes) {
      float us = 0;
      switch (cfg) {
        case 0: us = run<256, 8>(b, n, reps, pool, pool_bytes, xa, xb, x_bytes, st); break;
        case 1: us = run<256, 16>(b, n, reps, pool, pool_bytes, xa, xb, x_bytes, st); break;
        case 2: us = run<512, 8>(b, n, reps, pool, pool_bytes, xa, xb, x_bytes, st); break;
        case 3: us = run<512, 12>(b, n, reps, pool, pool_bytes, xa, xb, x_bytes, st); break;
        case 4: us = run<1024, 6>(b, n, reps, pool, pool_bytes, xa, xb, x_bytes, st); break;
        default: us = run<1024, 12>(b, n, reps, pool, pool_bytes, xa, xb, x_bytes, st); break;
      }
      printf("%10.2f ", us);
    }
    printf("\n");
  }
  unsigned long long* t_end; CK(hipMalloc(&t_end, 256 * 8));
  printf("\nstart offsets rotated per workgroup (512 threads x 8 loads in flight): us per launch (spread of the 256 end stamps of one launch)\n");
  printf("%-26s %18s %18s %18s %18s\n", "rotation (bytes / workgroup)", "18.9 MB", "31.5 MB", "50.3 MB", "100.7 MB");
  for (unsigned rot_bytes : {0u, 4096u, 9216u, 33792u, 66560u}) {
    printf("%-26u ", rot_bytes);
    for (size_t b : sizes) {
      float sp = 0;
      const float us = run<512, 8>(b, n, reps, pool, pool_bytes, xa, xb, x_bytes, st, rot_bytes / 16, t_end, &sp);
      printf("%9.2f (%6.2f) ", us, sp);
    }
    printf("\n");
  }
  return 0;
}
