// What fits in the shadow of an int8 MFMA on gfx950?  One wave stream: 1 v_mfma_i32_16x16x64_i8 followed by NV independent vector instructions
// (float32 FMA / float64 FMA / int32 lshl_add), 1 or 2 waves per SIMD.  Reports cycles per (MFMA + NV VALU) group per SIMD.
//   hipcc --offload-arch=gfx950 -O3 -w tools/micro/shadow.hip -o tools/micro/shadow
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x16 __attribute__((ext_vector_type(16)));
template <int NV, int KIND, int MF>
__global__ __launch_bounds__(512) void k(double* out, int iters, double a, double b, i32x4 fa, i32x4 fb) {
  double x[8]; float y[8]; int q[8];
  for (int i = 0; i < 8; ++i) { x[i] = a + i + threadIdx.x; y[i] = (float)x[i]; q[i] = threadIdx.x + i; }
  const float fa2 = (float)a, fb2 = (float)b;
  i32x4 acc[6];
  i32x16 big[3];
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 16; ++j) big[i][j] = i + j;
  for (int i = 0; i < 6; ++i) acc[i] = i32x4{i, 1, 2, 3};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 12; ++u) {
      if (MF == 1) acc[u % 6] = __builtin_amdgcn_mfma_i32_16x16x64_i8(fa, fb, acc[u % 6], 0, 0, 0);
      if (MF == 2) big[u % 3] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa, fb, big[u % 3], 0, 0, 0);
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        const int i = (u * NV + v) & 7;
        if (KIND == 0) y[i] = __builtin_fmaf(y[i], fb2, fa2);
        if (KIND == 1) x[i] = __builtin_fma(x[i], b, a);
        if (KIND == 2) q[i] = (q[i] << 3) + q[(i + 1) & 7];
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  double s = 0;
  for (int i = 0; i < 8; ++i) s += x[i] + y[i] + q[i];
  for (int i = 0; i < 6; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 16; ++j) s += big[i][j];
  out[blockIdx.x * 512 + threadIdx.x] = s;
}
template <int NV, int KIND, int MF>
static double run(double* out, int grid, int threads) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  i32x4 f = {0x01020304, 0x01010101, 0x02020202, 0x01000100};
  const int it = 4000;
  k<NV, KIND, MF><<<grid, threads>>>(out, it, 1.0, 0.5, f, f);
  hipEventRecord(e0); k<NV, KIND, MF><<<grid, threads>>>(out, it, 1.0, 0.5, f, f); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e-3 * 2.4e9 / ((double)it * 12) / (threads / 256);      // cycles per group per wave of a SIMD
}
template <int KIND>
static void test(const char* name, double* out, int grid) {
  for (int threads = 256; threads <= 512; threads += 256) {
    printf("%-14s %d wave(s)/SIMD: cycles per wave and group   MFMA alone %.1f |", name, threads / 256, run<0, KIND, 1>(out, grid, threads));
    printf(" NV=2: valu %.1f both %.1f |", run<2, KIND, 0>(out, grid, threads), run<2, KIND, 1>(out, grid, threads));
    printf(" NV=4: valu %.1f both %.1f |", run<4, KIND, 0>(out, grid, threads), run<4, KIND, 1>(out, grid, threads));
    printf(" NV=8: valu %.1f both %.1f\n", run<8, KIND, 0>(out, grid, threads), run<8, KIND, 1>(out, grid, threads));
  }
}
int main() {
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  double* out; hipMalloc(&out, sizeof(double) * p.multiProcessorCount * 512);
  for (int threads = 256; threads <= 512; threads += 256)
    printf("32x32x32 i8, %d wave(s)/SIMD: MFMA alone %.1f | + 4 v_fma_f64: %.1f (valu %.1f) | + 8 v_fma_f64: %.1f (valu %.1f) | + 8 v_fma_f32: %.1f (valu %.1f) | + 4 lshl_add %.1f (valu %.1f)\n", threads / 256,
           run<0, 1, 2>(out, p.multiProcessorCount, threads), run<4, 1, 2>(out, p.multiProcessorCount, threads), run<4, 1, 0>(out, p.multiProcessorCount, threads),
           run<8, 1, 2>(out, p.multiProcessorCount, threads), run<8, 1, 0>(out, p.multiProcessorCount, threads),
           run<8, 0, 2>(out, p.multiProcessorCount, threads), run<8, 0, 0>(out, p.multiProcessorCount, threads),
           run<4, 2, 2>(out, p.multiProcessorCount, threads), run<4, 2, 0>(out, p.multiProcessorCount, threads));
  test<0>("v_fma_f32", out, p.multiProcessorCount);
  test<1>("v_fma_f64", out, p.multiProcessorCount);
  test<2>("v_lshl_add_u32", out, p.multiProcessorCount);
  return 0;
}
