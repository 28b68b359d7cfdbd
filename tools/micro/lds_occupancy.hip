// How many 256-thread workgroups share a CU as a function of their LDS size (gfx950)?
// hipcc --offload-arch=gfx950 -O3 -o lds_occupancy lds_occupancy.hip && ./lds_occupancy
// Found: the budget is 160 KB per CU but allocations are rounded up, so "two per CU" ends near 74.8 KB
// and "three per CU" near 49 KB - a kernel one KB over the edge silently runs at half the occupancy.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(256) void k(float* out) {
  extern __shared__ float lds[];
  lds[threadIdx.x] = threadIdx.x;
  __syncthreads();
  out[threadIdx.x] = lds[255 - threadIdx.x];
}
__global__ __launch_bounds__(256) void spin(float* out, long long ticks) {
  extern __shared__ float lds[];
  lds[threadIdx.x] = threadIdx.x;
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks / 10) __builtin_amdgcn_s_sleep(8);      // wall clock: 100 MHz
  __syncthreads();
  if (blockIdx.x == 0) out[threadIdx.x] = lds[255 - threadIdx.x];
}
int main() {
  hipFuncSetAttribute(reinterpret_cast<const void*>(spin), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  int prev = -1;
  for (int b = 1024; b <= 160 * 1024; b += 256) {
    int n = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k, 256, b) != hipSuccess) { printf("%d: error\n", b); break; }
    if (n != prev) { printf("LDS %6d B -> %d workgroups/CU\n", b, n); prev = n; }
  }
  // what the hardware actually does: 2 x (#CUs) workgroups that each spin ~200 us; one round if two fit per CU
  hipDeviceProp_t pr; hipGetDeviceProperties(&pr, 0);
  float* d; hipMalloc(&d, 1 << 20);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int b = 64 * 1024; b <= 84 * 1024; b += 1024) {
    spin<<<2 * pr.multiProcessorCount, 256, b>>>(d, 200000);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    spin<<<2 * pr.multiProcessorCount, 256, b>>>(d, 200000);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("LDS %6d B: 2 workgroups per CU launched, %.3f ms\n", b, ms);
  }
  return 0;
}
