// Do MFMA and VALU instructions of DIFFERENT waves on one SIMD overlap?  Workgroup = 8 waves (2 per SIMD):
// waves 0-3 run an MFMA loop, waves 4-7 a VALU fma loop.  Times: MFMA only, VALU only, both.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_valu_overlap.hip -o tools/micro/mfma_valu_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(512) void k(float* out, int n_mfma, int n_valu, float a, float b) {
  const int w = threadIdx.x >> 6;
  float s = 0.f;
  if (w < 4) {
    f32x16 acc[2];
    for (int i = 0; i < 2; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = (float)r;
    for (int it = 0; it < n_mfma; ++it) {
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[1], 0, 0, 0);
      }
    }
    for (int r = 0; r < 16; ++r) s += acc[0][r] + acc[1][r];
  } else {
    float x[8];
    for (int i = 0; i < 8; ++i) x[i] = a + i + threadIdx.x;
    for (int it = 0; it < n_valu; ++it) {
#pragma unroll
      for (int u = 0; u < 16; ++u)
#pragma unroll
        for (int i = 0; i < 8; ++i) x[i] = fmaf(x[i], b, a);
    }
    for (int i = 0; i < 8; ++i) s += x[i];
  }
  out[blockIdx.x * 512 + threadIdx.x] = s;
}
static float run(float* out, int grid, int nm, int nv) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<<<grid, 512>>>(out, nm, nv, 1.0f, 0.5f);
  hipEventRecord(e0); k<<<grid, 512>>>(out, nm, nv, 1.0f, 0.5f); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}
int main() {
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  const int grid = p.multiProcessorCount;
  float* out; hipMalloc(&out, sizeof(float) * grid * 512);
  // 16 MFMAs (64 cycles each) per iteration = 1024 cycles; 128 VALU fma (4 cycles each?) per iteration
  const int NM = 20000, NV = 40000;
  const float tm = run(out, grid, NM, 0), tv = run(out, grid, 0, NV), tb = run(out, grid, NM, NV);
  printf("MFMA only %.3f ms, VALU only %.3f ms, both %.3f ms  (sum %.3f, max %.3f)\n", tm, tv, tb, tm + tv, tm > tv ? tm : tv);
  printf("VALU: %.2f cycles per wave64 fma at 2.4 GHz\n", tv * 1e-3 * 2.4e9 / ((double)NV * 128));
  return 0;
}
