// Calibration of SQ_VALU_MFMA_BUSY_CYCLES / SQ_BUSY_CYCLES: pure-MFMA loops (registers only, four independent accumulators, every SIMD of the chip
// busy) for each instruction family the engine uses.  Run under the SAME --pmc set as the bench (tools/profile_gpu.sh) - the ratio a kernel of
// 100 % matrix-pipe issue reports is what the per-kernel ratios of profiles/rNN_sq_counters.md are divided by ("calibrated MFMA %").
//   hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_calib.hip -o tools/micro/mfma_calib && tools/micro/mfma_calib
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x16 __attribute__((ext_vector_type(16)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef double f64x4 __attribute__((ext_vector_type(4)));

#define LOOP(ACC_T, NR, ZERO, CALL)                                                        \
  ACC_T acc[4];                                                                            \
  for (int i = 0; i < 4; ++i) for (int r = 0; r < NR; ++r) acc[i][r] = ZERO;               \
  for (int it = 0; it < iters; ++it) {                                                     \
    _Pragma("unroll") for (int u = 0; u < 8; ++u) {                                        \
      _Pragma("unroll") for (int i = 0; i < 4; ++i) acc[i] = CALL;                         \
    }                                                                                      \
  }                                                                                        \
  double s = 0.0;                                                                          \
  for (int i = 0; i < 4; ++i) for (int r = 0; r < NR; ++r) s += (double)acc[i][r];         \
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;

__global__ __launch_bounds__(256) void calib_bf16_32x32x16(double* out, int iters, s16x8 a, s16x8 b) { LOOP(f32x16, 16, 0.f, __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0)) }
__global__ __launch_bounds__(256) void calib_bf16_16x16x32(double* out, int iters, s16x8 a, s16x8 b) { LOOP(f32x4, 4, 0.f, __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0)) }
__global__ __launch_bounds__(256) void calib_f16_32x32x16(double* out, int iters, h16x8 a, h16x8 b) { LOOP(f32x16, 16, 0.f, __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[i], 0, 0, 0)) }
__global__ __launch_bounds__(256) void calib_i8_16x16x64(double* out, int iters, i32x4 a, i32x4 b) { LOOP(i32x4, 4, 0, __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, acc[i], 0, 0, 0)) }
__global__ __launch_bounds__(256) void calib_i8_32x32x32(double* out, int iters, i32x4 a, i32x4 b) { LOOP(i32x16, 16, 0, __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, acc[i], 0, 0, 0)) }
__global__ __launch_bounds__(256) void calib_f32_32x32x2(double* out, int iters, float a, float b) { LOOP(f32x16, 16, 0.f, __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0)) }
__global__ __launch_bounds__(256) void calib_f64_16x16x4(double* out, int iters, double a, double b) { LOOP(f64x4, 4, 0.0, __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0)) }

template <class F>
static void run(const char* name, double ops_per_mfma, F launch) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 4000, grid = 256 * 2;      // two 4-wave workgroups per CU: two waves per SIMD
  launch(grid, 100);
  hipDeviceSynchronize();
  hipEventRecord(e0); launch(grid, iters); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms = 0.f; hipEventElapsedTime(&ms, e0, e1);
  const double n_mfma = (double)grid * 4 * iters * 32;
  printf("%-22s %8.3f ms  %8.1f T(FL)OP/s  %6.2f cycles per MFMA and SIMD at 2.4 GHz\n", name, ms, n_mfma * ops_per_mfma / (ms * 1e-3) / 1e12,
         ms * 1e-3 * 2.4e9 / (n_mfma / 1024.0));
}

int main() {
  double* out; hipMalloc(&out, sizeof(double) * 256 * 2 * 256);
  s16x8 sa = {0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80};
  h16x8 ha = {1, 1, 1, 1, 1, 1, 1, 1};
  i32x4 ia = {0x01010101, 0x01010101, 0x01010101, 0x01010101};
  run("bf16 32x32x16", 2.0 * 32 * 32 * 16, [&](int g, int it) { hipLaunchKernelGGL(calib_bf16_32x32x16, dim3(g), dim3(256), 0, 0, out, it, sa, sa); });
  run("bf16 16x16x32", 2.0 * 16 * 16 * 32, [&](int g, int it) { hipLaunchKernelGGL(calib_bf16_16x16x32, dim3(g), dim3(256), 0, 0, out, it, sa, sa); });
  run("f16 32x32x16", 2.0 * 32 * 32 * 16, [&](int g, int it) { hipLaunchKernelGGL(calib_f16_32x32x16, dim3(g), dim3(256), 0, 0, out, it, ha, ha); });
  run("i8 16x16x64", 2.0 * 16 * 16 * 64, [&](int g, int it) { hipLaunchKernelGGL(calib_i8_16x16x64, dim3(g), dim3(256), 0, 0, out, it, ia, ia); });
  run("i8 32x32x32", 2.0 * 32 * 32 * 32, [&](int g, int it) { hipLaunchKernelGGL(calib_i8_32x32x32, dim3(g), dim3(256), 0, 0, out, it, ia, ia); });
  run("f32 32x32x2", 2.0 * 32 * 32 * 2, [&](int g, int it) { hipLaunchKernelGGL(calib_f32_32x32x2, dim3(g), dim3(256), 0, 0, out, it, 1.f, 1.f); });
  run("f64 16x16x4", 2.0 * 16 * 16 * 4, [&](int g, int it) { hipLaunchKernelGGL(calib_f64_16x16x4, dim3(g), dim3(256), 0, 0, out, it, 1.0, 1.0); });
  hipFree(out);
  return 0;
}
