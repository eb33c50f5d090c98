// Per-launch floor on the GPU box: empty kernel and a 25 MB -> 25 MB copy, 768 x 256 threads, 20 back-to-back launches.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void empty_k(int* p) { if (p && threadIdx.x == 1024) *p = 1; }
__global__ void copy_k(const uint4* a, uint4* b, int per) {
  const size_t base = (size_t)blockIdx.x * blockDim.x * per + threadIdx.x;
  uint4 v[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = a[base + (size_t)i * blockDim.x];
#pragma unroll
  for (int i = 0; i < 8; ++i) b[base + (size_t)i * blockDim.x] = v[i];
}
__global__ void copy1_k(const uint4* a, uint4* b, size_t n16) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n16) b[i] = a[i];
}
int main() {
  const size_t n = 25165824;  // bytes
  void *a, *b; hipMalloc(&a, n); hipMalloc(&b, n); hipMemset(a, 1, n);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float ms;
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0);
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(empty_k, dim3(768), dim3(256), 0, 0, (int*)nullptr);
    hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
    printf("empty kernel 768x256: %.2f us per launch\n", ms / 20 * 1e3);
    hipEventRecord(e0);
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(copy_k, dim3(768), dim3(256), 0, 0, (const uint4*)a, (uint4*)b, 8);
    hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
    printf("copy 25 MB -> 25 MB, 768x256 (128 B per thread): %.2f us per launch (%.2f TB/s)\n", ms / 20 * 1e3, 2.0 * n / (ms / 20 * 1e-3) / 1e12);
    hipEventRecord(e0);
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(copy1_k, dim3(6144), dim3(256), 0, 0, (const uint4*)a, (uint4*)b, n / 16);
    hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
    printf("copy 25 MB -> 25 MB, 6144x256 (16 B per thread): %.2f us per launch\n", ms / 20 * 1e3);
  }
  return 0;
}
