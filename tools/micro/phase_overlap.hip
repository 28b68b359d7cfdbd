// Two waves per SIMD, barrier-synchronised phases (the structure of te_rec_fwdx): per phase each wave has one block of NM int8 MFMAs and one
// block of NV float64 FMAs.  "lockstep": both waves MFMA first, then FMA.  "anti-phase": waves 0-3 MFMA then FMA, waves 4-7 FMA then MFMA.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/phase_overlap.hip -o tools/micro/phase_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int i32x4 __attribute__((ext_vector_type(4)));
template <int NM, int NV, int F32 = 0>
__global__ __launch_bounds__(512) void k(double* out, int iters, int anti, int prio, double a, double b, i32x4 fa, i32x4 fb) {
  const int w = threadIdx.x >> 6;
  double x[8];
  for (int i = 0; i < 8; ++i) x[i] = a + i + threadIdx.x;
  float y[8]; const float fa2 = (float)a, fb2 = (float)b;
  for (int i = 0; i < 8; ++i) y[i] = (float)x[i];
  i32x4 acc[5];
  for (int i = 0; i < 5; ++i) acc[i] = i32x4{i, 1, 2, 3};
  const bool mfma_first = anti == 0 || (anti == 1 ? w < 4 : anti == 2 ? (w & 1) == 0 : ((w >> 1) & 1) == 0);
  auto M = [&]() {
    if (prio) __builtin_amdgcn_s_setprio(3);
#pragma unroll
    for (int u = 0; u < NM; ++u) acc[u % 5] = __builtin_amdgcn_mfma_i32_16x16x64_i8(fa, fb, acc[u % 5], 0, 0, 0);
    if (prio) __builtin_amdgcn_s_setprio(0);
  };
  auto V = [&]() {
#pragma unroll
    for (int u = 0; u < NV / 8; ++u)
#pragma unroll
      for (int i = 0; i < 8; ++i) { if (F32) y[i] = __builtin_fmaf(y[i], fb2, fa2); else x[i] = __builtin_fma(x[i], b, a); }
  };
  for (int it = 0; it < iters; ++it) {
    if (mfma_first) { M(); __builtin_amdgcn_sched_barrier(0); V(); } else { V(); __builtin_amdgcn_sched_barrier(0); M(); }
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
  }
  double s = 0;
  for (int i = 0; i < 8; ++i) s += x[i] + y[i];
  for (int i = 0; i < 5; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * 512 + threadIdx.x] = s;
}
__global__ void hwid(unsigned* out) { unsigned v; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(v)); if ((threadIdx.x & 63) == 0) out[threadIdx.x >> 6] = v; }
template <int NM, int NV, int F32 = 0>
static float run(double* out, int grid, int iters, int anti, int prio = 0) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  i32x4 f = {0x01020304, 0x01010101, 0x02020202, 0x01000100};
  k<NM, NV, F32><<<grid, 512>>>(out, iters, anti, prio, 1.0, 0.5, f, f);
  hipEventRecord(e0); k<NM, NV, F32><<<grid, 512>>>(out, iters, anti, prio, 1.0, 0.5, f, f); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}
int main() {
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  double* out; hipMalloc(&out, sizeof(double) * p.multiProcessorCount * 512);
  const int it = 20000;
  const float a = run<30, 256>(out, p.multiProcessorCount, it, 0), b = run<30, 256>(out, p.multiProcessorCount, it, 1);
  const float b2 = run<30, 256>(out, p.multiProcessorCount, it, 2), b3 = run<30, 256>(out, p.multiProcessorCount, it, 3);
  printf("anti-phase by (w & 1): %.0f cycles, by ((w >> 1) & 1): %.0f cycles\n", b2 * 1e-3 * 2.4e9 / it, b3 * 1e-3 * 2.4e9 / it);
  for (int a2 = 0; a2 < 2; ++a2) printf("with s_setprio(3) around the MFMA block: %s %.0f cycles\n", a2 ? "anti-phase" : "lockstep", run<30, 256>(out, p.multiProcessorCount, it, a2, 1) * 1e-3 * 2.4e9 / it);
  printf("float32 FMAs (30 MFMA + 512 v_fma_f32): MFMA only %.0f, FMA only %.0f, lockstep %.0f, anti-phase %.0f, anti-phase + setprio %.0f\n",
         run<30, 0, 1>(out, p.multiProcessorCount, it, 0) * 1e-3 * 2.4e9 / it, run<0, 512, 1>(out, p.multiProcessorCount, it, 0) * 1e-3 * 2.4e9 / it,
         run<30, 512, 1>(out, p.multiProcessorCount, it, 0) * 1e-3 * 2.4e9 / it, run<30, 512, 1>(out, p.multiProcessorCount, it, 1) * 1e-3 * 2.4e9 / it,
         run<30, 512, 1>(out, p.multiProcessorCount, it, 1, 1) * 1e-3 * 2.4e9 / it);
  { unsigned* ids; hipMalloc(&ids, 8 * sizeof(unsigned)); hwid<<<1, 512>>>(ids); unsigned h[8]; hipMemcpy(h, ids, sizeof(h), hipMemcpyDeviceToHost);
    for (int i = 0; i < 8; ++i) printf("wave %d: HW_ID %08x simd %u wave slot %u\n", i, h[i], (h[i] >> 4) & 3, h[i] & 15); }
  const float m = run<30, 0>(out, p.multiProcessorCount, it, 0), v = run<0, 256>(out, p.multiProcessorCount, it, 0);
  printf("per phase (2 waves / SIMD, 30 MFMA + 256 f64 FMA each): lockstep %.0f cycles, anti-phase %.0f cycles; MFMA only %.0f, FMA only %.0f\n",
         a * 1e-3 * 2.4e9 / it, b * 1e-3 * 2.4e9 / it, m * 1e-3 * 2.4e9 / it, v * 1e-3 * 2.4e9 / it);
  return 0;
}
