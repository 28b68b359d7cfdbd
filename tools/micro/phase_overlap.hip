// Two waves per SIMD, barrier-synchronised phases (the structure of te_rec_fwdx): per phase each wave has one block of NM int8 MFMAs and one
// block of NV float64 FMAs.  "lockstep": both waves MFMA first, then FMA.  "anti-phase": waves 0-3 MFMA then FMA, waves 4-7 FMA then MFMA.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/phase_overlap.hip -o tools/micro/phase_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int i32x4 __attribute__((ext_vector_type(4)));
template <int NM, int NV>
__global__ __launch_bounds__(512) void k(double* out, int iters, int anti, double a, double b, i32x4 fa, i32x4 fb) {
  const int w = threadIdx.x >> 6;
  double x[8];
  for (int i = 0; i < 8; ++i) x[i] = a + i + threadIdx.x;
  i32x4 acc[5];
  for (int i = 0; i < 5; ++i) acc[i] = i32x4{i, 1, 2, 3};
  const bool mfma_first = !anti || w < 4;
  auto M = [&]() {
#pragma unroll
    for (int u = 0; u < NM; ++u) acc[u % 5] = __builtin_amdgcn_mfma_i32_16x16x64_i8(fa, fb, acc[u % 5], 0, 0, 0);
  };
  auto V = [&]() {
#pragma unroll
    for (int u = 0; u < NV / 8; ++u)
#pragma unroll
      for (int i = 0; i < 8; ++i) x[i] = __builtin_fma(x[i], b, a);
  };
  for (int it = 0; it < iters; ++it) {
    if (mfma_first) { M(); __builtin_amdgcn_sched_barrier(0); V(); } else { V(); __builtin_amdgcn_sched_barrier(0); M(); }
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
  }
  double s = 0;
  for (int i = 0; i < 8; ++i) s += x[i];
  for (int i = 0; i < 5; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * 512 + threadIdx.x] = s;
}
template <int NM, int NV>
static float run(double* out, int grid, int iters, int anti) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  i32x4 f = {0x01020304, 0x01010101, 0x02020202, 0x01000100};
  k<NM, NV><<<grid, 512>>>(out, iters, anti, 1.0, 0.5, f, f);
  hipEventRecord(e0); k<NM, NV><<<grid, 512>>>(out, iters, anti, 1.0, 0.5, f, f); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}
int main() {
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  double* out; hipMalloc(&out, sizeof(double) * p.multiProcessorCount * 512);
  const int it = 20000;
  const float a = run<30, 256>(out, p.multiProcessorCount, it, 0), b = run<30, 256>(out, p.multiProcessorCount, it, 1);
  const float m = run<30, 0>(out, p.multiProcessorCount, it, 0), v = run<0, 256>(out, p.multiProcessorCount, it, 0);
  printf("per phase (2 waves / SIMD, 30 MFMA + 256 f64 FMA each): lockstep %.0f cycles, anti-phase %.0f cycles; MFMA only %.0f, FMA only %.0f\n",
         a * 1e-3 * 2.4e9 / it, b * 1e-3 * 2.4e9 / it, m * 1e-3 * 2.4e9 / it, v * 1e-3 * 2.4e9 / it);
  return 0;
}
