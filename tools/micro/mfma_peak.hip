// Sustained issue rate of the f32 MFMA shapes used by the engine (registers only, no memory traffic):
//   hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_peak.hip -o /tmp/mfma_peak && /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ __launch_bounds__(256) void k32(float* out, int iters, float a, float b) {
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = (float)i;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 16; ++u)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NACC>
__global__ __launch_bounds__(256) void k16(float* out, int iters, float a, float b) {
  f32x4 acc[NACC];
  for (int i = 0; i < NACC; ++i) for (int r = 0; r < 4; ++r) acc[i][r] = (float)i;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 16; ++u)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < NACC; ++i) for (int r = 0; r < 4; ++r) s += acc[i][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
// gemm_nt-style inner loop fed from LDS: per k-group 2 A + 2 B float4 reads (issued one group ahead) for 16 MFMAs
template <bool PIN>
__global__ __launch_bounds__(256, 2) void klds(float* out, int iters) {
  __shared__ __align__(16) float As[128][36];
  __shared__ __align__(16) float Bs[128][36];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, li = lane & 31, h = lane >> 5;
  for (int e = tid; e < 128 * 36; e += 256) { (&As[0][0])[e] = 0.001f * (e % 7); (&Bs[0][0])[e] = 0.002f * (e % 5); }
  __syncthreads();
  const int wm = (w >> 1) * 64, wn = (w & 1) * 64;
  f32x16 acc[2][2];
  for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  for (int it = 0; it < iters; ++it) {
    float4 a[2][2], b[2][2];
    for (int i = 0; i < 2; ++i) a[0][i] = *reinterpret_cast<const float4*>(&As[wm + 32 * i + li][4 * h]);
    for (int j = 0; j < 2; ++j) b[0][j] = *reinterpret_cast<const float4*>(&Bs[wn + 32 * j + li][4 * h]);
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      if (m + 1 < 4) {
        for (int i = 0; i < 2; ++i) a[(m + 1) & 1][i] = *reinterpret_cast<const float4*>(&As[wm + 32 * i + li][8 * (m + 1) + 4 * h]);
        for (int j = 0; j < 2; ++j) b[(m + 1) & 1][j] = *reinterpret_cast<const float4*>(&Bs[wn + 32 * j + li][8 * (m + 1) + 4 * h]);
      }
      if (PIN) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const float4 x = a[m & 1][i], y = b[m & 1][j];
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(x.x, y.x, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(x.y, y.y, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(x.z, y.z, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(x.w, y.w, acc[i][j], 0, 0, 0);
        }
      if (PIN) __builtin_amdgcn_sched_barrier(0);
    }
  }
  float s = 0.f;
  for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) s += acc[i][j][r];
  out[blockIdx.x * 256 + tid] = s;
}
template <typename K>
static void run_lds(const char* name, K kern, int wg_per_cu) {
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  const int grid = p.multiProcessorCount * wg_per_cu, iters = 20000;
  float* out; hipMalloc(&out, sizeof(float) * grid * 256);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, out, 10);
  hipEventRecord(e0); hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, out, iters); hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double fl = (double)grid * 4 * iters * 64.0 * 4096.0;
  printf("%-28s wg/cu=%d: %7.1f TFLOP/s\n", name, wg_per_cu, fl / (ms * 1e-3) / 1e12);
  hipFree(out);
}
template <typename K>
static void run(const char* name, K kern, int nacc, double flop_per_mfma, int wg_per_cu) {
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  const int grid = p.multiProcessorCount * wg_per_cu, iters = 4000;
  float* out; hipMalloc(&out, sizeof(float) * grid * 256);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, out, 10, 1.0f, 1.0f);
  hipEventRecord(e0); hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, out, iters, 1.0f, 1.0f); hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double fl = (double)grid * 4 /*waves*/ * iters * 16.0 * nacc * flop_per_mfma;
  printf("%-28s acc=%d wg/cu=%d: %7.1f TFLOP/s\n", name, nacc, wg_per_cu, fl / (ms * 1e-3) / 1e12);
  hipFree(out);
}
int main() {
  for (int wg : {1, 2}) {
    run("v_mfma_f32_32x32x2_f32", k32<1>, 1, 4096.0, wg);
    run("v_mfma_f32_32x32x2_f32", k32<4>, 4, 4096.0, wg);
    run("v_mfma_f32_16x16x4_f32", k16<1>, 1, 2048.0, wg);
    run("v_mfma_f32_16x16x4_f32", k16<4>, 4, 2048.0, wg);
  }
  for (int wg : {1, 2}) { run_lds("LDS-fed 128x128 tile, pinned", klds<true>, wg); run_lds("LDS-fed 128x128 tile, free", klds<false>, wg); }
  return 0;
}
