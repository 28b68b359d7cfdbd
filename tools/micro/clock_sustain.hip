// Sustained shader clock and f32 MFMA rate under continuous load (does the 2.4 GHz / 157.3 TFLOP/s peak hold for seconds?):
//   hipcc --offload-arch=gfx950 -O3 tools/micro/clock_sustain.hip -o /tmp/clock_sustain && /tmp/clock_sustain
// One MFMA-loop launch of ~10 ms after another for ~3 s; per launch: TFLOP/s from HIP events, effective shader clock from
// s_memtime (shader cycles) against the constant-rate wall clock (100 MHz).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(256) void k32(float* out, long long* clk, int iters, float a, float b) {
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = (float)i;
  const long long c0 = clock64(), w0 = wall_clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 16; ++u)
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
  }
  const long long c1 = clock64(), w1 = wall_clock64();
  float s = 0.f;
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = c1 - c0; clk[1] = w1 - w0; }
}

int main() {
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  const int grid = p.multiProcessorCount * 2, iters = 20000;      // 2 workgroups x 4 waves per CU: two waves per SIMD
  float* out; long long* clk; hipMalloc(&out, sizeof(float) * grid * 256); hipMalloc(&clk, 16);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const double flop = (double)grid * 4 /*waves*/ * iters * 64.0 /*mfma per iter*/ * 4096.0;
  double t_total = 0;
  for (int l = 0; l < 400; ++l) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(k32, dim3(grid), dim3(256), 0, 0, out, clk, iters, 1.0f, 0.5f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long h[2]; hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
    t_total += ms;
    if (l % 25 == 0 || l == 399)
      printf("t=%7.1f ms  launch %3d: %6.2f ms  %6.1f TFLOP/s  shader clock %6.0f MHz (clock64 / wall_clock64 at 100 MHz)\n", t_total, l, ms,
             flop / (ms * 1e-3) / 1e12, (double)h[0] / ((double)h[1] / 100.0));
  }
  return 0;
}
