// What limits te_head's MFMA phases?  The kernel's inner loop in isolation: 32-row A tile from LDS (one ds_read_b128 per k-group),
// packed B fragments streamed from a 114 KB global buffer (L2-resident, one k-group of prefetch), NTW accumulators, three workgroups
// of four waves per CU (46 KB of LDS each).  Variants switch the B stream / the A reads off (registers instead).
//   hipcc --offload-arch=gfx950 -O3 tools/micro/head_loop.hip -o /tmp/head_loop && /tmp/head_loop
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0); }

template <int NTW, int K8, bool BG, bool AL, int WGPC>
__global__ __launch_bounds__(256, WGPC) void k(const float4* __restrict__ bp, float* out, int tiles, int ntile_b) {
  extern __shared__ __align__(16) float lds[];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, li = lane & 31, h = lane >> 5;
  constexpr int LDA = K8 * 8 + 4;
  for (int e = tid; e < 32 * LDA; e += 256) lds[e] = 0.001f * (e % 13);
  __syncthreads();
  const float* arow = lds + li * LDA + 4 * h;
  f32x16 acc[NTW];
  for (int j = 0; j < NTW; ++j) for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  for (int t = 0; t < tiles; ++t) {
    const float4* bj[NTW];
    float4 bc[NTW], bn[NTW], ac, an;
#pragma unroll
    for (int j = 0; j < NTW; ++j) { bj[j] = bp + ((size_t)((w + 4 * j + t) % ntile_b) * K8) * 64 + lane; bc[j] = BG ? *bj[j] : make_float4(1.f, 2.f, 3.f, 4.f); }
    ac = AL ? *reinterpret_cast<const float4*>(arow) : make_float4(.1f, .2f, .3f, .4f);
#pragma unroll 2
    for (int m = 0; m < K8; ++m) {
      const int mn = m + 1 < K8 ? m + 1 : m;
#pragma unroll
      for (int j = 0; j < NTW; ++j) bn[j] = BG ? bj[j][(size_t)mn * 64] : bc[j];
      an = AL ? *reinterpret_cast<const float4*>(arow + 8 * mn) : ac;
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < NTW; ++j) {
        acc[j] = mfma32(ac.x, bc[j].x, acc[j]);
        acc[j] = mfma32(ac.y, bc[j].y, acc[j]);
        acc[j] = mfma32(ac.z, bc[j].z, acc[j]);
        acc[j] = mfma32(ac.w, bc[j].w, acc[j]);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < NTW; ++j) bc[j] = bn[j];
      ac = an;
    }
  }
  float s = 0.f;
  for (int j = 0; j < NTW; ++j) for (int r = 0; r < 16; ++r) s += acc[j][r];
  out[blockIdx.x * 256 + tid] = s;
}

template <int NTW, int K8, bool BG, bool AL, int WGPC>
static void run(const char* name, const float4* bp, float* out, int cus) {
  const int tiles = 2000, grid = cus * WGPC;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const size_t lds = 46 * 1024;
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<NTW, K8, BG, AL, WGPC>), dim3(grid), dim3(256), lds, 0, bp, out, tiles, 7);
    hipEventRecord(e1); hipEventSynchronize(e1);
  }
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double flop = (double)grid * 4 * tiles * K8 * NTW * 4 * 4096.0;
  printf("%-58s %7.2f ms  %6.1f TFLOP/s\n", name, ms, flop / (ms * 1e-3) / 1e12);
}

int main() {
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  const int cus = p.multiProcessorCount;
  float4* bp; float* out;
  hipMalloc(&bp, sizeof(float4) * 7 * 28 * 64 * 2); hipMemset(bp, 0, sizeof(float4) * 7 * 28 * 64 * 2);
  hipMalloc(&out, sizeof(float) * cus * 4 * 256);
  run<2, 16, true, true, 3>("logits loop: NTW 2, K8 16, B global, A LDS, 3 WG/CU", bp, out, cus);
  run<2, 16, false, true, 3>("              B in registers", bp, out, cus);
  run<2, 16, true, false, 3>("              A in registers", bp, out, cus);
  run<2, 16, false, false, 3>("              both in registers", bp, out, cus);
  run<1, 28, true, true, 3>("DH loop: NTW 1, K8 28, B global, A LDS, 3 WG/CU", bp, out, cus);
  run<1, 28, false, true, 3>("              B in registers", bp, out, cus);
  run<1, 28, true, false, 3>("              A in registers", bp, out, cus);
  run<1, 28, false, false, 3>("              both in registers", bp, out, cus);
  run<2, 16, true, true, 2>("logits loop, 2 WG/CU", bp, out, cus);
  run<1, 28, true, true, 2>("DH loop, 2 WG/CU", bp, out, cus);
  run<2, 16, true, true, 4>("logits loop, 4 WG/CU (LDS permitting)", bp, out, cus);
  run<1, 28, true, true, 4>("DH loop, 4 WG/CU", bp, out, cus);
  return 0;
}
