// Calibration of rocprofv3's WRITE_SIZE on gfx950 for the store patterns of the training step (MI355X_MICROARCH.md: "WRITE_SIZE is
// uncalibrated: calibrate on a known byte count in your own access pattern").  Every mode writes exactly rows x 1536 bytes:
//   0  float4 per lane, fully coalesced (1 KB per wave instruction)
//   1  the recurrent kernels' pattern: 512-thread workgroup, wave w owns columns [16 w, 16 w + 16) of each of the three 128-column gates,
//      lane = 16 g + j writes ONE float of row 4 g' + r: 64-byte segments per wave instruction, the 8 waves of the workgroup cover the row
//   2  as 1, but every workgroup walks its 16 rows 'steps' times apart in time (a barrier and ~1 us of ALU work between row groups)
// usage: write_calib <mode> [rows]      (run under rocprofv3 --pmc WRITE_SIZE; prints the bytes written)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

__global__ __launch_bounds__(256) void k0(float4* dst, size_t n4) {
  for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) dst[i] = make_float4(1.f, 2.f, 3.f, 4.f);
}
template <int SPIN>
__global__ __launch_bounds__(512) void k1(float* dst, int rows) {
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63, g4 = 4 * (lane >> 4), col = 16 * w + (lane & 15);
  float acc = (float)threadIdx.x;
  for (int r0 = blockIdx.x * 16; r0 < rows; r0 += gridDim.x * 16) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float* g = dst + (size_t)(r0 + g4 + r) * 384;
      g[col] = acc; g[128 + col] = acc + 1.f; g[256 + col] = acc + 2.f;
    }
    if (SPIN) {
      for (int i = 0; i < SPIN; ++i) acc = __builtin_fmaf(acc, 1.000001f, 0.5f);
      __syncthreads();
    }
  }
}
int main(int argc, char** argv) {
  const int mode = argc > 1 ? atoi(argv[1]) : 0;
  const int rows = argc > 2 ? atoi(argv[2]) : 230960;
  float* d; hipMalloc(&d, (size_t)rows * 1536 + 4096);
  hipMemset(d, 0, (size_t)rows * 1536);
  hipDeviceSynchronize();
  for (int it = 0; it < 3; ++it) {
    if (mode == 0) hipLaunchKernelGGL(k0, dim3(2048), dim3(256), 0, 0, (float4*)d, (size_t)rows * 96);
    else if (mode == 1) hipLaunchKernelGGL(k1<0>, dim3(256), dim3(512), 0, 0, d, rows);
    else hipLaunchKernelGGL(k1<2000>, dim3(256), dim3(512), 0, 0, d, rows);
  }
  hipDeviceSynchronize();
  printf("mode %d: %zu bytes per launch\n", mode, (size_t)rows * 1536);
  return 0;
}
