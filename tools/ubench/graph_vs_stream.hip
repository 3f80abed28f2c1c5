// Would a hipGraph shorten a SMALL blocking call of libisx?  The call's device-side sequence -- H2D of a short list, three small
// memsets, one kernel of ~0.3 ms, two small D2H copies into pinned memory, one memset -- issued (a) as stream operations, as
// isx_api.hip does, and (b) as ONE hipGraphLaunch of the captured sequence with the kernel's parameters updated per call
// (hipGraphExecKernelNodeSetParams: seed / first ray change every call).  Prints wall time per call minus the kernel's own time.
//   hipcc -O2 --offload-arch=gfx950 -o graph_vs_stream graph_vs_stream.hip && ./graph_vs_stream
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void spin_kernel(unsigned long long* out, unsigned long long first, int spin) {
  // ~`spin` dependent fma per lane: stands in for the trace kernel of a 5e4-ray call (latency-bound, ~0.3 ms)
  double x = (double)(first + threadIdx.x) * 1e-9 + 1.0;
  for (int i = 0; i < spin; ++i) x = __builtin_fma(x, 1.0000001, 1e-9);
  if (x == 12345.678) out[0] = 1;   // never
  if (blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(out, first & 1ull);
}

int main() {
  hipStream_t s;
  CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  unsigned long long *d_hist, *d_stats, *d_aux; unsigned int* d_ctr; unsigned char* h_pin;
  CK(hipMalloc(&d_hist, 64)); CK(hipMalloc(&d_stats, 64)); CK(hipMalloc(&d_aux, 4096)); CK(hipMalloc(&d_ctr, 64));
  CK(hipHostMalloc((void**)&h_pin, 1 << 18, hipHostMallocDefault));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int spin = 60000, reps = 300;
  auto issue = [&](unsigned long long first) {
    hipMemcpyAsync(d_aux, h_pin + 4096, 48, hipMemcpyHostToDevice, s);
    hipMemsetAsync(d_hist, 0, 8, s);
    hipMemsetAsync(d_ctr, 0, 16, s);
    hipEventRecord(e0, s);
    hipLaunchKernelGGL(spin_kernel, dim3(261), dim3(256), 0, s, d_hist, first, spin);
    hipEventRecord(e1, s);
    hipMemcpyAsync(h_pin + 64, d_hist, 8, hipMemcpyDeviceToHost, s);
    hipMemcpyAsync(h_pin, d_stats, 64, hipMemcpyDeviceToHost, s);
    hipMemsetAsync(d_stats, 0, 64, s);
  };
  for (int k = 0; k < 5; ++k) { issue(k); CK(hipStreamSynchronize(s)); }
  float kms = 0; CK(hipEventElapsedTime(&kms, e0, e1));
  auto t0 = std::chrono::steady_clock::now();
  for (int k = 0; k < reps; ++k) { issue(k); CK(hipStreamSynchronize(s)); }
  const double stream_us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / reps;
  // (b) the same sequence as a graph
  hipGraph_t g; hipGraphExec_t ge;
  CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
  issue(0);
  CK(hipStreamEndCapture(s, &g));
  CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  size_t nn = 0; CK(hipGraphGetNodes(g, nullptr, &nn));
  std::vector<hipGraphNode_t> nodes(nn); CK(hipGraphGetNodes(g, nodes.data(), &nn));
  hipGraphNode_t kn = nullptr;
  for (auto n : nodes) { hipGraphNodeType t; CK(hipGraphNodeGetType(n, &t)); if (t == hipGraphNodeTypeKernel) kn = n; }
  if (!kn) { std::printf("no kernel node captured\n"); return 1; }
  hipKernelNodeParams kp; CK(hipGraphKernelNodeGetParams(kn, &kp));
  unsigned long long first = 0; int sp = spin; void* args[3] = {&d_hist, &first, &sp};
  kp.kernelParams = args;
  for (int k = 0; k < 5; ++k) { first = k; CK(hipGraphExecKernelNodeSetParams(ge, kn, &kp)); CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s)); }
  t0 = std::chrono::steady_clock::now();
  for (int k = 0; k < reps; ++k) { first = k; CK(hipGraphExecKernelNodeSetParams(ge, kn, &kp)); CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s)); }
  const double graph_us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / reps;
  std::printf("kernel %.1f us; per call: stream operations %.1f us (+%.1f over the kernel), hipGraph %.1f us (+%.1f), %zu graph nodes\n",
              kms * 1e3, stream_us, stream_us - kms * 1e3, graph_us, graph_us - kms * 1e3, nn);
  return 0;
}
