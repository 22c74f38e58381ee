// Cost of a dependent kernel boundary as a function of the workgroup shape (threads, dynamic LDS) — what a launch of the
// launch path pays whatever its kernel does. A chain of N trivial kernels (each workgroup reads one dword written by the
// previous launch and writes one) is captured into a hipGraph and replayed; time per launch = graph time / N.
//   hipcc -O3 --offload-arch=gfx950 profiles/tools/microbench/boundary.hip -o gpurun_out/boundary && gpurun_out/boundary
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int THREADS>
__global__ __launch_bounds__(THREADS) void hop(const unsigned* in, unsigned* out, int work) {
  extern __shared__ unsigned lds[];
  unsigned v = in[blockIdx.x];
  if (work) {   // touch the LDS so that the allocation is real
    lds[threadIdx.x] = v + threadIdx.x;
    __syncthreads();
    v = lds[(threadIdx.x + 1) % THREADS];
  }
  if (threadIdx.x == 0) out[blockIdx.x] = v + 1;
}

template <int THREADS>
static float run(int grid, size_t lds, int work, int n, int reps, unsigned* a, unsigned* b, hipStream_t st) {
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&hop<THREADS>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  hipGraph_t g; hipGraphExec_t ge;
  CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
  for (int i = 0; i < n; ++i) {
    hipLaunchKernelGGL((hop<THREADS>), dim3(grid), dim3(THREADS), lds, st, (i & 1) ? b : a, (i & 1) ? a : b, work);
  }
  CK(hipStreamEndCapture(st, &g));
  CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int r = 0; r < 3; ++r) CK(hipGraphLaunch(ge, st));
  CK(hipStreamSynchronize(st));
  float best = 1e9f;
  for (int r = 0; r < reps; ++r) {
    CK(hipEventRecord(e0, st)); CK(hipGraphLaunch(ge, st)); CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
  }
  CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
  return best * 1000.f / n;
}

int main() {
  hipStream_t st; CK(hipStreamCreate(&st));
  unsigned *a, *b; CK(hipMalloc(&a, 4096 * 4)); CK(hipMalloc(&b, 4096 * 4)); CK(hipMemset(a, 0, 4096 * 4)); CK(hipMemset(b, 0, 4096 * 4));
  const int n = 400, reps = 10;
  printf("us per launch in a chain of %d dependent launches (hipGraph), min of %d replays\n", n, reps);
  printf("%-28s %8s %8s %8s %8s\n", "grid x threads", "0 KiB", "64 KiB", "128 KiB", "160 KiB");
  const size_t ldss[4] = {0, 64 * 1024, 128 * 1024, 160 * 1024};
  for (int grid : {256, 512, 1024}) {
    for (int th : {64, 256, 512, 1024}) {
      printf("%5d x %-4d (touch LDS)     ", grid, th);
      for (size_t l : ldss) {
        if (grid > 256 && l > 64 * 1024 && false) { printf("%8s ", "-"); continue; }
        float us = 0;
        const size_t eff = l ? l : th * 4;
        switch (th) {
          case 64: us = run<64>(grid, eff, 1, n, reps, a, b, st); break;
          case 256: us = run<256>(grid, eff, 1, n, reps, a, b, st); break;
          case 512: us = run<512>(grid, eff, 1, n, reps, a, b, st); break;
          default: us = run<1024>(grid, eff, 1, n, reps, a, b, st); break;
        }
        printf("%8.2f ", us);
      }
      printf("\n");
    }
  }
  return 0;
}
