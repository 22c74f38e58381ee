// Where do the argument blocks of a captured graph's kernel nodes live? Each kernel records its own kernarg segment
// pointer; the host prints the differences between consecutive launches (eager, then graph replays).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

struct Block { const void* p[8]; int v[48]; };   // 256 bytes
struct Small { const void* p[2]; int v[12]; };   // 64 bytes

__global__ void k_big(const Block b, unsigned long long* out, int slot) {
  if (threadIdx.x == 0 && blockIdx.x == 0) out[slot] = reinterpret_cast<unsigned long long>(__builtin_amdgcn_kernarg_segment_ptr());
}
__global__ void k_small(const Small b, unsigned long long* out, int slot) {
  if (threadIdx.x == 0 && blockIdx.x == 0) out[slot] = reinterpret_cast<unsigned long long>(__builtin_amdgcn_kernarg_segment_ptr());
}

int main() {
  hipStream_t st;
  hipStreamCreate(&st);
  const int n = 12;
  unsigned long long* out;
  hipMalloc(&out, 8 * n);
  Block b{};
  Small s{};
  auto enqueue = [&]() {
    for (int i = 0; i < n; ++i) {
      if (i % 3 == 2) hipLaunchKernelGGL(k_small, dim3(256), dim3(64), 0, st, s, out, i);
      else hipLaunchKernelGGL(k_big, dim3(256), dim3(64), 0, st, b, out, i);
    }
  };
  std::vector<unsigned long long> h(n);
  auto show = [&](const char* what) {
    hipStreamSynchronize(st);
    hipMemcpy(h.data(), out, 8 * n, hipMemcpyDeviceToHost);
    printf("%s\n", what);
    for (int i = 0; i < n; ++i) printf("  launch %2d (%s): %#llx  %+lld\n", i, i % 3 == 2 ? "64-B block " : "256-B block", h[i], i ? (long long)(h[i] - h[i - 1]) : 0LL);
  };
  enqueue();
  show("eager launches");
  hipGraph_t g;
  hipGraphExec_t ge;
  hipStreamBeginCapture(st, hipStreamCaptureModeRelaxed);
  enqueue();
  hipStreamEndCapture(st, &g);
  hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
  hipGraphLaunch(ge, st);
  show("graph replay 1");
  hipGraphLaunch(ge, st);
  show("graph replay 2");
  return 0;
}
