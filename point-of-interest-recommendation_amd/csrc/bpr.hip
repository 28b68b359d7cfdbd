// BPR-MF step (OboBpr.bpr_train, public/BPR.py:201-241): u = ux[u].(lt[p]-lt[q]),
// g = -sigmoid(-u); ux[u] -= a(g(xp-xq) + l ux[u]); lt[p] -= a(g usr + l xp); lt[q] -= a(-g usr + l xq).
// The whole step is three row gathers + three row scatters: HBM-bound (6*D*4 + 12 bytes / triple).
#include "poi_common.h"
#include "poi_kernels.h"

namespace poi {

template <int LPT>   // lanes per triple (16 / 32 / 64); float4 per lane per pass
__device__ __forceinline__ float group_sum(float v) {
#pragma unroll
  for (int o = LPT / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, LPT);
  return v;
}

// HOGWILD: in-place.  One pass: 3 rows in, 3 rows out.
template <int LPT>
__global__ __launch_bounds__(POI_BLOCK) void bpr_hogwild_kernel(BprArgs A) {
  const int D = A.dim;
  const int gl = threadIdx.x % LPT;
  const int gpb = POI_BLOCK / LPT;
  for (int i = blockIdx.x * gpb + threadIdx.x / LPT; i < A.n; i += gridDim.x * gpb) {
    float* ur = A.ux + (size_t)A.uidx[i] * D;
    float* pr = A.lt + (size_t)A.p[i] * D;
    float* qr = A.lt + (size_t)A.q[i] * D;
    float dot = 0.f;
    for (int j = gl * 4; j < D; j += LPT * 4) {
      const float4 u = *reinterpret_cast<const float4*>(ur + j);
      const float4 a = *reinterpret_cast<const float4*>(pr + j);
      const float4 b = *reinterpret_cast<const float4*>(qr + j);
      dot += u.x * (a.x - b.x) + u.y * (a.y - b.y) + u.z * (a.z - b.z) + u.w * (a.w - b.w);
    }
    dot = group_sum<LPT>(dot);
    const float g = -sigmoidf_(-dot);
    const float al = A.alpha, lm = A.lambda;
    for (int j = gl * 4; j < D; j += LPT * 4) {
      const float4 u = *reinterpret_cast<const float4*>(ur + j);
      const float4 a = *reinterpret_cast<const float4*>(pr + j);
      const float4 b = *reinterpret_cast<const float4*>(qr + j);
      float4 un, an, bn;
      un.x = u.x - al * (g * (a.x - b.x) + lm * u.x); un.y = u.y - al * (g * (a.y - b.y) + lm * u.y);
      un.z = u.z - al * (g * (a.z - b.z) + lm * u.z); un.w = u.w - al * (g * (a.w - b.w) + lm * u.w);
      an.x = a.x - al * (g * u.x + lm * a.x); an.y = a.y - al * (g * u.y + lm * a.y);
      an.z = a.z - al * (g * u.z + lm * a.z); an.w = a.w - al * (g * u.w + lm * a.w);
      bn.x = b.x - al * (-g * u.x + lm * b.x); bn.y = b.y - al * (-g * u.y + lm * b.y);
      bn.z = b.z - al * (-g * u.z + lm * b.z); bn.w = b.w - al * (-g * u.w + lm * b.w);
      *reinterpret_cast<float4*>(ur + j) = un;
      *reinterpret_cast<float4*>(pr + j) = an;
      *reinterpret_cast<float4*>(qr + j) = bn;
    }
    if (gl == 0) A.loss[i] = -log_sigmoidf_(dot);
  }
}

// SNAPSHOT phase 1: gradients of every triple at the launch-entry values -> gradient tables.
template <int LPT>
__global__ __launch_bounds__(POI_BLOCK) void bpr_grad_kernel(BprArgs A) {
  const int D = A.dim;
  const int gl = threadIdx.x % LPT;
  const int gpb = POI_BLOCK / LPT;
  for (int i = blockIdx.x * gpb + threadIdx.x / LPT; i < A.n; i += gridDim.x * gpb) {
    const int u = A.uidx[i], p = A.p[i], q = A.q[i];
    const float* ur = A.ux + (size_t)u * D;
    const float* pr = A.lt + (size_t)p * D;
    const float* qr = A.lt + (size_t)q * D;
    float dot = 0.f;
    for (int j = gl * 4; j < D; j += LPT * 4) {
      const float4 uu = *reinterpret_cast<const float4*>(ur + j);
      const float4 a = *reinterpret_cast<const float4*>(pr + j);
      const float4 b = *reinterpret_cast<const float4*>(qr + j);
      dot += uu.x * (a.x - b.x) + uu.y * (a.y - b.y) + uu.z * (a.z - b.z) + uu.w * (a.w - b.w);
    }
    dot = group_sum<LPT>(dot);
    const float g = -sigmoidf_(-dot);
    float* gu = A.g_ux + (size_t)u * D;
    float* gp = A.g_lt + (size_t)p * D;
    float* gq = A.g_lt + (size_t)q * D;
    for (int j = gl * 4; j < D; j += LPT * 4) {
      const float4 uu = *reinterpret_cast<const float4*>(ur + j);
      const float4 a = *reinterpret_cast<const float4*>(pr + j);
      const float4 b = *reinterpret_cast<const float4*>(qr + j);
      atomicAdd(gu + j + 0, g * (a.x - b.x)); atomicAdd(gu + j + 1, g * (a.y - b.y));
      atomicAdd(gu + j + 2, g * (a.z - b.z)); atomicAdd(gu + j + 3, g * (a.w - b.w));
      atomicAdd(gp + j + 0, g * uu.x); atomicAdd(gp + j + 1, g * uu.y);
      atomicAdd(gp + j + 2, g * uu.z); atomicAdd(gp + j + 3, g * uu.w);
      atomicAdd(gq + j + 0, -g * uu.x); atomicAdd(gq + j + 1, -g * uu.y);
      atomicAdd(gq + j + 2, -g * uu.z); atomicAdd(gq + j + 3, -g * uu.w);
    }
    if (gl == 0) {
      A.loss[i] = -log_sigmoidf_(dot);
      atomicAdd(&A.cnt_ux[u], 1); atomicAdd(&A.cnt_lt[p], 1); atomicAdd(&A.cnt_lt[q], 1);
    }
  }
}

template <int LPT>
__device__ __forceinline__ void bpr_claim(float* T, float* G, int* cnt, int row, int D, int gl, float al, float lm, float cap) {
  int got = 0;
  if (gl == 0) got = atomicExch(&cnt[row], 0);
  got = __shfl(got, 0, LPT);
  if (got <= 0) return;
  const float inv = 1.0f / (float)got;
  al *= fminf((float)got, cap);            // batch rule: min(got, cap) of the touching triples' updates count
  float* t = T + (size_t)row * D;
  float* g = G + (size_t)row * D;
  for (int j = gl * 4; j < D; j += LPT * 4) {
    float4 tv = *reinterpret_cast<float4*>(t + j);
    const float4 gv = *reinterpret_cast<float4*>(g + j);
    tv.x -= al * (gv.x * inv + lm * tv.x); tv.y -= al * (gv.y * inv + lm * tv.y);
    tv.z -= al * (gv.z * inv + lm * tv.z); tv.w -= al * (gv.w * inv + lm * tv.w);
    *reinterpret_cast<float4*>(t + j) = tv;
    *reinterpret_cast<float4*>(g + j) = make_float4(0.f, 0.f, 0.f, 0.f);
  }
}

// SNAPSHOT phase 2: row <- row - alpha * (mean over touching triples of their gradient + lambda*row)
template <int LPT>
__global__ __launch_bounds__(POI_BLOCK) void bpr_apply_kernel(BprArgs A) {
  const int D = A.dim;
  const int gl = threadIdx.x % LPT;
  const int gpb = POI_BLOCK / LPT;
  for (int i = blockIdx.x * gpb + threadIdx.x / LPT; i < A.n; i += gridDim.x * gpb) {
    bpr_claim<LPT>(A.ux, A.g_ux, A.cnt_ux, A.uidx[i], D, gl, A.alpha, A.lambda, A.bcap);
    bpr_claim<LPT>(A.lt, A.g_lt, A.cnt_lt, A.p[i], D, gl, A.alpha, A.lambda, A.bcap);
    bpr_claim<LPT>(A.lt, A.g_lt, A.cnt_lt, A.q[i], D, gl, A.alpha, A.lambda, A.bcap);
  }
}

template <int LPT>
static hipError_t launch_bpr_t(const BprArgs& A, int mode, hipStream_t st, Timing* tm) {
  const int gpb = POI_BLOCK / LPT;
  int grid = (A.n + gpb - 1) / gpb;
  if (grid > 256 * 16) grid = 256 * 16;
  if (grid < 1) grid = 1;
  if (mode == 1) {
    tm->begin("bpr_hogwild", st);
    hipLaunchKernelGGL(bpr_hogwild_kernel<LPT>, dim3(grid), dim3(POI_BLOCK), 0, st, A);
    tm->end(st);
  } else {
    tm->begin("bpr_grad", st);
    hipLaunchKernelGGL(bpr_grad_kernel<LPT>, dim3(grid), dim3(POI_BLOCK), 0, st, A);
    tm->end(st);
    tm->begin("bpr_apply", st);
    hipLaunchKernelGGL(bpr_apply_kernel<LPT>, dim3(grid), dim3(POI_BLOCK), 0, st, A);
    tm->end(st);
  }
  return hipGetLastError();
}

hipError_t launch_bpr(const BprArgs& A, int mode, hipStream_t st, Timing* tm) {
  if (A.dim <= 64) return launch_bpr_t<16>(A, mode, st, tm);
  if (A.dim <= 128) return launch_bpr_t<32>(A, mode, st, tm);
  return launch_bpr_t<64>(A, mode, st, tm);
}

}  // namespace poi
