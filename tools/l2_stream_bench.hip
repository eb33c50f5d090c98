// L2 -> CU fill-rate microbenchmark (GPU box): how many bytes per second can ONE workgroup per CU pull from an L2-resident buffer
//   dma : global_load_lds_dwordx4 into a 128 KB LDS ring (the GEMM / attention operand path), D instructions in flight per wave
//   reg : global_load_dwordx4 into registers (8 or 16 in flight per lane), results XOR-folded so that nothing is dead
// The GEMM ring kernels measure ~45 GB/s per CU (11.5 TB/s chip-wide) on the dma path whatever the schedule; this tool asks whether
// that is the path's ceiling or the kernels'.   hipcc --offload-arch=gfx950 -O3 tools/l2_stream_bench.hip -o build/l2_stream_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void glb_void_t;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

// every workgroup streams `span` bytes starting at its own offset (wrapping inside `total`), `passes` times
template <int DEPTH>
__global__ __launch_bounds__(512) void k_dma(const char* __restrict__ src, size_t total, size_t span, int passes, unsigned* sink) {
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  const int wid = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const size_t start = ((size_t)blockIdx.x * 262144) % total;
  char* ring = smem + wid * 16384;                       // 16 x 1 KB slots per wave
  const size_t steps = span / 8192;                      // 8 waves x 1 KB per step
  int slot = 0;
  for (int ps = 0; ps < passes; ++ps) {
    for (size_t s = 0; s < steps; ++s) {
      size_t off = start + s * 8192 + wid * 1024 + lane * 16;
      off = off >= total ? off - total : off;
      __builtin_amdgcn_global_load_lds((glb_void_t*)(src + off), (lds_void_t*)(ring + slot * 1024), 16, 0, 0);
      slot = (slot + 1) & 15;
      asm volatile("s_waitcnt vmcnt(%0)" :: "n"(DEPTH - 1) : "memory");
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (sink && threadIdx.x == 0) sink[blockIdx.x] = *reinterpret_cast<unsigned*>(smem + 64);
}

template <int DEPTH>
__global__ __launch_bounds__(512) void k_reg(const char* __restrict__ src, size_t total, size_t span, int passes, unsigned* sink) {
  const int wid = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const size_t start = ((size_t)blockIdx.x * 262144) % total;
  const size_t steps = span / (8192 * DEPTH);
  uint4 acc = {0, 0, 0, 0};
  for (int ps = 0; ps < passes; ++ps) {
    for (size_t s = 0; s < steps; ++s) {
      uint4 v[DEPTH];
#pragma unroll
      for (int d = 0; d < DEPTH; ++d) {
        size_t off = start + (s * DEPTH + d) * 8192 + wid * 1024 + lane * 16;
        off = off >= total ? off - total : off;
        v[d] = *reinterpret_cast<const uint4*>(src + off);
      }
#pragma unroll
      for (int d = 0; d < DEPTH; ++d) { acc.x ^= v[d].x; acc.y ^= v[d].y; acc.z ^= v[d].z; acc.w ^= v[d].w; }
    }
  }
  if (sink && (acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[blockIdx.x] = lane + wid;
}

template <typename F>
static double time_ms(F f, int reps) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  f(); CK(hipDeviceSynchronize());
  CK(hipEventRecord(a));
  for (int i = 0; i < reps; ++i) f();
  CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  return ms / reps;
}

int main() {
  hipDeviceProp_t pr; CK(hipGetDeviceProperties(&pr, 0));
  const int cus = pr.multiProcessorCount;
  unsigned* sink; CK(hipMalloc(&sink, 4096 * 4));
  const size_t sizes[] = {2u << 20, 16u << 20, 512u << 20};          // L2-resident per XCD, MALL-resident, HBM
  const char* names[] = {"2 MB (L2)", "16 MB (L2+MALL)", "512 MB (HBM)"};
  for (int si = 0; si < 3; ++si) {
    const size_t total = sizes[si];
    char* buf; CK(hipMalloc(&buf, total)); CK(hipMemset(buf, 1, total));
    const size_t span = 2u << 20; const int passes = 8;
    const double bytes = (double)cus * span * passes;
    for (int wgs = 1; wgs <= 2; ++wgs) {
      const int grid = cus * wgs;
      const double b = bytes * wgs;
      const size_t lds = wgs == 1 ? 131072 : 65536;       // 2 workgroups per CU: 64 KB each (waves use the first 8 slots only)
      auto rep = [&](const char* what, double ms) {
        printf("%-16s %-10s wg/CU %d: %7.3f ms  %7.1f GB/s per CU  %6.2f TB/s chip\n", names[si], what, wgs, ms, b / grid * wgs / ms / 1e6, b / ms / 1e9);
      };
      if (wgs == 1) {
        CK(hipFuncSetAttribute((const void*)k_dma<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
        CK(hipFuncSetAttribute((const void*)k_dma<8>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
        CK(hipFuncSetAttribute((const void*)k_dma<16>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
        rep("dma d4", time_ms([&] { hipLaunchKernelGGL(k_dma<4>, dim3(grid), dim3(512), lds, 0, buf, total, span, passes, sink); }, 5));
        rep("dma d8", time_ms([&] { hipLaunchKernelGGL(k_dma<8>, dim3(grid), dim3(512), lds, 0, buf, total, span, passes, sink); }, 5));
        rep("dma d16", time_ms([&] { hipLaunchKernelGGL(k_dma<16>, dim3(grid), dim3(512), lds, 0, buf, total, span, passes, sink); }, 5));
      }
      rep("reg d4", time_ms([&] { hipLaunchKernelGGL(k_reg<4>, dim3(grid), dim3(512), 0, 0, buf, total, span, passes, sink); }, 5));
      rep("reg d8", time_ms([&] { hipLaunchKernelGGL(k_reg<8>, dim3(grid), dim3(512), 0, 0, buf, total, span, passes, sink); }, 5));
      rep("reg d16", time_ms([&] { hipLaunchKernelGGL(k_reg<16>, dim3(grid), dim3(512), 0, 0, buf, total, span, passes, sink); }, 5));
    }
    CK(hipFree(buf));
  }
  return 0;
}
