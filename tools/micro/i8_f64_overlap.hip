// Does the int8 matrix pipe overlap with float64 / int32 vector work on gfx950 - (a) between the two waves of a SIMD, (b) inside ONE wave whose
// instruction stream alternates?  And what does a wave64 float64 instruction cost?  (The exact forward, te_xfwd.hip, is ~90 i8 MFMAs + ~650
// vector instructions per wave and step.)   hipcc --offload-arch=gfx950 -O3 tools/micro/i8_f64_overlap.hip -o tools/micro/i8_f64_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int i32x4 __attribute__((ext_vector_type(4)));
// mode 0: waves 0-3 MFMA, waves 4-7 VALU (different waves, same SIMDs); mode 1: every wave alternates 1 MFMA + NV VALU
template <int VK>
__device__ __forceinline__ void valu(double (&x)[8], int (&q)[8], double b, double a) {
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    if (VK == 0) x[i] = __builtin_fma(x[i], b, a);                       // v_fma_f64
    if (VK == 1) q[i] = (q[i] << 3) + q[(i + 1) & 7];                    // v_lshl_add_u32
    if (VK == 2) x[i] = (double)q[i] + x[i];                             // v_cvt_f64_i32 + v_add_f64
    if (VK == 3) x[i] = __builtin_amdgcn_rcp(x[i]);                      // v_rcp_f64
    if (VK == 4) x[i] = __builtin_amdgcn_ldexp(x[i], q[i] & 1);          // v_ldexp_f64
    if (VK == 5) x[i] = __builtin_rint(x[i] * b);                        // v_mul_f64 + v_rndne_f64
  }
}
template <int VK>
__global__ __launch_bounds__(512) void k(double* out, int n_mfma, int n_valu, int mode, double a, double b, i32x4 fa, i32x4 fb) {
  const int w = threadIdx.x >> 6;
  double s = 0.0;
  double x[8]; int q[8];
  for (int i = 0; i < 8; ++i) { x[i] = a + i + threadIdx.x; q[i] = threadIdx.x + i; }
  i32x4 acc[4];
  for (int i = 0; i < 4; ++i) acc[i] = i32x4{i, 1, 2, 3};
  if (mode == 0) {
    if (w < 4) {
      for (int it = 0; it < n_mfma; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          acc[0] = __builtin_amdgcn_mfma_i32_16x16x64_i8(fa, fb, acc[0], 0, 0, 0); acc[1] = __builtin_amdgcn_mfma_i32_16x16x64_i8(fa, fb, acc[1], 0, 0, 0);
          acc[2] = __builtin_amdgcn_mfma_i32_16x16x64_i8(fa, fb, acc[2], 0, 0, 0); acc[3] = __builtin_amdgcn_mfma_i32_16x16x64_i8(fa, fb, acc[3], 0, 0, 0);
        }
      }
    } else {
      for (int it = 0; it < n_valu; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) valu<VK>(x, q, b, a);
      }
    }
  } else {
    for (int it = 0; it < n_mfma; ++it) {
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        acc[u & 3] = __builtin_amdgcn_mfma_i32_16x16x64_i8(fa, fb, acc[u & 3], 0, 0, 0);
        if (n_valu) valu<VK>(x, q, b, a);      // 8 vector instructions behind every MFMA
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  for (int i = 0; i < 8; ++i) s += x[i] + q[i];
  for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * 512 + threadIdx.x] = s;
}
template <int VK>
static float run(double* out, int grid, int nm, int nv, int mode) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  i32x4 f = {0x01020304, 0x01010101, 0x02020202, 0x01000100};
  k<VK><<<grid, 512>>>(out, nm, nv, mode, 1.0, 0.5, f, f);
  hipEventRecord(e0); k<VK><<<grid, 512>>>(out, nm, nv, mode, 1.0, 0.5, f, f); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}
template <int VK>
static void test(const char* name, double* out, int grid) {
  const int NM = 20000, NV = 5000;
  const float tm = run<VK>(out, grid, NM, 0, 0), tv = run<VK>(out, grid, 0, NV, 0), tb = run<VK>(out, grid, NM, NV, 0);
  printf("%-28s two waves: MFMA only %.3f ms (%.1f cycles each), VALU only %.3f ms (%.2f cycles per wave64 instruction), both %.3f ms (sum %.3f, max %.3f)\n", name, tm,
         tm * 1e-3 * 2.4e9 / ((double)NM * 16), tv, tv * 1e-3 * 2.4e9 / ((double)NV * 128), tb, tm + tv, tm > tv ? tm : tv);
  const int N1 = 5000;
  const float a0 = run<VK>(out, grid, N1, 0, 1), a1 = run<VK>(out, grid, N1, 1, 1);
  printf("%-28s one stream (2 waves / SIMD, 1 MFMA + 8 VALU alternating): MFMA only %.3f ms, with VALU %.3f ms = %.1f cycles per (MFMA + 8 VALU) pair of both waves\n", "", a0, a1,
         a1 * 1e-3 * 2.4e9 / ((double)N1 * 16));
}
// accuracy of v_rcp_f64 (the seed of te_xfwd.hip's 1 / (1 + e^x)): max relative error over 2^20 arguments in [1, 2^40]
__global__ void rcp_err(double* out) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  const double d = (1.0 + (double)i * 9.5367431640625e-7) * (double)(1ull << (i % 41));
  const double y = __builtin_amdgcn_rcp(d), e0 = fabs(__builtin_fma(-d, y, 1.0));
  const double y1 = __builtin_fma(y, __builtin_fma(-d, y, 1.0), y), e1 = fabs(__builtin_fma(-d, y1, 1.0));
  atomicMax((unsigned long long*)out, (unsigned long long)__double_as_longlong(e0));
  atomicMax((unsigned long long*)out + 1, (unsigned long long)__double_as_longlong(e1));
}
int main() {
  { double* o; hipMalloc(&o, 16); hipMemset(o, 0, 16); rcp_err<<<4096, 256>>>(o); double h[2]; hipMemcpy(h, o, 16, hipMemcpyDeviceToHost);
    printf("v_rcp_f64: max |1 - d rcp(d)| = %.3e; after one Newton step %.3e\n", h[0], h[1]); }
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  const int grid = p.multiProcessorCount;
  double* out; hipMalloc(&out, sizeof(double) * grid * 512);
  test<0>("v_fma_f64", out, grid); test<1>("v_lshl_add_u32", out, grid); test<2>("v_cvt_f64_i32 + v_add_f64", out, grid);
  test<3>("v_rcp_f64", out, grid); test<4>("v_ldexp_f64", out, grid); test<5>("v_mul_f64 + v_rndne_f64", out, grid);
  return 0;
}
