// Device-side helpers shared by the gfx950 kernels: 64-lane wavefront reductions (DPP), block
// reductions through LDS, and the two GEMV shapes the per-sequence engine uses.
#pragma once
#include <hip/hip_runtime.h>
#include <mutex>
#include <hip/hip_fp16.h>
#include <stdint.h>

#define POI_WAVE 64
#define POI_BLOCK 256
#define POI_NWAVE (POI_BLOCK / POI_WAVE)

namespace poi {

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }
__device__ __forceinline__ int wave_id() { return threadIdx.x >> 6; }

template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, false));
}
__device__ __forceinline__ float readlane_f(float v, int l) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l));
}

// Sum over the 64 lanes of a wavefront; result is wave-uniform (returned in every lane).
// Four DPP adds reduce each 16-lane row (quad swap, quad-pair swap, half-mirror, mirror), then the
// four row sums are read through SGPRs.
__device__ __forceinline__ float wave_sum(float v) {
  v += dpp_f<0xB1>(v);   // quad_perm [1,0,3,2]
  v += dpp_f<0x4E>(v);   // quad_perm [2,3,0,1]
  v += dpp_f<0x141>(v);  // row_half_mirror
  v += dpp_f<0x140>(v);  // row_mirror
  return (readlane_f(v, 0) + readlane_f(v, 16)) + (readlane_f(v, 32) + readlane_f(v, 48));
}
__device__ __forceinline__ float wave_max(float v) {
  v = fmaxf(v, dpp_f<0xB1>(v));
  v = fmaxf(v, dpp_f<0x4E>(v));
  v = fmaxf(v, dpp_f<0x141>(v));
  v = fmaxf(v, dpp_f<0x140>(v));
  return fmaxf(fmaxf(readlane_f(v, 0), readlane_f(v, 16)), fmaxf(readlane_f(v, 32), readlane_f(v, 48)));
}
// Reference implementation through ds_bpermute, used only by the self-test.
__device__ __forceinline__ float wave_sum_shfl(float v) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// Block-wide (256 threads) reductions; `red` is an LDS array of >= POI_NWAVE floats.  Contains barriers.
__device__ __forceinline__ float block_sum(float v, float* red) {
  v = wave_sum(v);
  __syncthreads();
  if (lane_id() == 0) red[wave_id()] = v;
  __syncthreads();
  return (red[0] + red[1]) + (red[2] + red[3]);
}
__device__ __forceinline__ float block_max(float v, float* red) {
  v = wave_max(v);
  __syncthreads();
  if (lane_id() == 0) red[wave_id()] = v;
  __syncthreads();
  return fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}

// ---- table element access: float32 or IEEE half storage, float32 arithmetic (config X: fp16 embedding tables) ----------------
// Offsets are in ELEMENTS.  A half row of 4 elements is one 8-byte load / store; conversion is round-to-nearest-even.
__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ float4 ld4(const __half* p) {
  const uint2 u = *reinterpret_cast<const uint2*>(p);
  const __half2 a = *reinterpret_cast<const __half2*>(&u.x), b = *reinterpret_cast<const __half2*>(&u.y);
  const float2 fa = __half22float2(a), fb = __half22float2(b);
  return make_float4(fa.x, fa.y, fb.x, fb.y);
}
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
__device__ __forceinline__ void st4(__half* p, float4 v) {
  const __half2 a = __floats2half2_rn(v.x, v.y), b = __floats2half2_rn(v.z, v.w);
  uint2 u; u.x = *reinterpret_cast<const unsigned*>(&a); u.y = *reinterpret_cast<const unsigned*>(&b);
  *reinterpret_cast<uint2*>(p) = u;
}
// Stochastic rounding float32 -> half (poi_ctx_set_f16_rounding): the 13 mantissa bits a half drops decide - a uniform 13-bit number is
// added below the half's last place and the sum truncated, so the value rounds up with probability (v - floor) / ulp and the EXPECTED stored
// value is v.  With round-to-nearest an SGD update below half an fp16 ulp of the element (the whole L2 decay alpha lambda |x| = 1e-5 |x|
// against a half ulp of 2.4e-4 |x| .. 4.9e-4 |x|) is lost every time; stochastically it is applied in expectation.  `rnd` = 32 random
// bits per 4 elements (hash of the element index and the launch's salt): reproducible for a given launch sequence.  Values below the
// half's normal range (2^-14) are rounded to nearest.
__device__ __forceinline__ unsigned sr_hash(unsigned x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
__device__ __forceinline__ __half sr_half(float v, unsigned r13) {
  const unsigned u = __float_as_uint(v);
  if ((u & 0x7f800000u) < 0x38800000u) return __float2half_rn(v);            // |v| < 2^-14 (or zero): nearest
  return __float2half_rz(__uint_as_float((u + (r13 & 0x1FFFu)) & ~0x1FFFu)); // exact conversion: the low 13 bits are zero
}
__device__ __forceinline__ void st4_sr(__half* p, float4 v, unsigned seed) {
  const unsigned r0 = sr_hash(seed), r1 = sr_hash(seed ^ 0x9E3779B9u);
  const __half2 a = __halves2half2(sr_half(v.x, r0), sr_half(v.y, r0 >> 13)), b = __halves2half2(sr_half(v.z, r1), sr_half(v.w, r1 >> 13));
  uint2 w; w.x = *reinterpret_cast<const unsigned*>(&a); w.y = *reinterpret_cast<const unsigned*>(&b);
  *reinterpret_cast<uint2*>(p) = w;
}
// runtime-typed (f16 is wave-uniform)
__device__ __forceinline__ float4 ld4t(const void* base, size_t off, int f16) {
  return f16 ? ld4(reinterpret_cast<const __half*>(base) + off) : ld4(reinterpret_cast<const float*>(base) + off);
}
__device__ __forceinline__ void st4t(void* base, size_t off, int f16, float4 v) {
  if (f16) st4(reinterpret_cast<__half*>(base) + off, v); else st4(reinterpret_cast<float*>(base) + off, v);
}
// ... with stochastic rounding of a half table when salt != 0 (`eidx` = element index of v.x in the table)
__device__ __forceinline__ void st4t_sr(void* base, size_t off, int f16, float4 v, unsigned salt, size_t eidx) {
  if (f16 && salt) st4_sr(reinterpret_cast<__half*>(base) + off, v, (unsigned)(eidx >> 2) * 0x85ebca6bu + (unsigned)(eidx >> 34) + salt);
  else st4t(base, off, f16, v);
}

__device__ __forceinline__ float dot4(const float4 a, const float4 b) {
  return fmaf(a.x, b.x, fmaf(a.y, b.y, fmaf(a.z, b.z, a.w * b.w)));
}
__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + __expf(-x)); }
// log(sigmoid(x)) = -softplus(-x), stable on both sides.
__device__ __forceinline__ float log_sigmoidf_(float x) {
  return x >= 0.f ? -log1pf(expf(-x)) : x - log1pf(expf(x));
}

// Row-parallel GEMV: out[r] = act(W1[r,:K1].x1 + W2[r,:K2].x2 + bias[r]) for r in [0, nrows).
// One wavefront per row (4 rows in flight per wave), lanes stride the row in float4 (coalesced
// 1-KiB row segments from HBM/L2), wave_sum closes the dot product.  x1/x2 live in LDS.
// ACT: 0 none, 1 sigmoid, 2 tanh.  W2 may be nullptr (K2 = 0).
template <int ACT>
__device__ __forceinline__ void gemv_rows(const float* __restrict__ W1, int K1, const float* x1,
                                          const float* __restrict__ W2, int K2, const float* x2,
                                          const float* __restrict__ bias, int nrows, float* out) {
  const int lane = lane_id(), w = wave_id();
  for (int r0 = w * 4; r0 < nrows; r0 += POI_NWAVE * 4) {
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int r = r0 + u;
      if (r < nrows) {
        const float* w1 = W1 + (size_t)r * K1;
        for (int j = lane * 4; j < K1; j += 256)
          acc[u] += dot4(*reinterpret_cast<const float4*>(w1 + j), *reinterpret_cast<const float4*>(x1 + j));
        if (W2) {
          const float* w2 = W2 + (size_t)r * K2;
          for (int j = lane * 4; j < K2; j += 256)
            acc[u] += dot4(*reinterpret_cast<const float4*>(w2 + j), *reinterpret_cast<const float4*>(x2 + j));
        }
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const float s = wave_sum(acc[u]);
      const int r = r0 + u;
      if (lane == 0 && r < nrows) {
        float v = s + (bias ? bias[r] : 0.f);
        if (ACT == 1) v = sigmoidf_(v);
        if (ACT == 2) v = tanhf(v);
        out[r] = v;
      }
    }
  }
}

// Column-parallel transposed GEMV: out[j] (+)= sum_i W[i, j] * v[i], W row-major (rows x cols),
// cols % 4 == 0, cols <= 1024.  Thread t owns the float4 column c = t % (cols/4) for the row group
// rg = t / (cols/4); the RG partial sums meet in LDS `part` (>= 1024 floats).  v and out are LDS.
// Contains barriers; `out` is overwritten (ACCUM=false) or incremented (ACCUM=true).
template <bool ACCUM>
__device__ __forceinline__ void gemv_cols(const float* __restrict__ W, int rows, int cols,
                                          const float* v, float* out, float* part) {
  const int c4n = cols >> 2;
  const int RG = POI_BLOCK / c4n > 0 ? POI_BLOCK / c4n : 1;
  const int tid = threadIdx.x;
  const int c = tid % c4n, rg = tid / c4n;
  if (rg < RG) {
    float4 a0 = {0, 0, 0, 0}, a1 = {0, 0, 0, 0};
    int i = rg;
    for (; i + RG < rows; i += 2 * RG) {
      const float4 w0 = *reinterpret_cast<const float4*>(W + (size_t)i * cols + 4 * c);
      const float4 w1 = *reinterpret_cast<const float4*>(W + (size_t)(i + RG) * cols + 4 * c);
      const float v0 = v[i], v1 = v[i + RG];
      a0.x = fmaf(w0.x, v0, a0.x); a0.y = fmaf(w0.y, v0, a0.y); a0.z = fmaf(w0.z, v0, a0.z); a0.w = fmaf(w0.w, v0, a0.w);
      a1.x = fmaf(w1.x, v1, a1.x); a1.y = fmaf(w1.y, v1, a1.y); a1.z = fmaf(w1.z, v1, a1.z); a1.w = fmaf(w1.w, v1, a1.w);
    }
    if (i < rows) {
      const float4 w0 = *reinterpret_cast<const float4*>(W + (size_t)i * cols + 4 * c);
      const float v0 = v[i];
      a0.x = fmaf(w0.x, v0, a0.x); a0.y = fmaf(w0.y, v0, a0.y); a0.z = fmaf(w0.z, v0, a0.z); a0.w = fmaf(w0.w, v0, a0.w);
    }
    float4 s = {a0.x + a1.x, a0.y + a1.y, a0.z + a1.z, a0.w + a1.w};
    *reinterpret_cast<float4*>(part + rg * cols + 4 * c) = s;
  }
  __syncthreads();
  for (int j = tid; j < cols; j += POI_BLOCK) {
    float s = 0.f;
    for (int g = 0; g < RG; ++g) s += part[g * cols + j];
    if (ACCUM) out[j] += s; else out[j] = s;
  }
  __syncthreads();
}

// cos(x) for the small angle differences of nearby POIs: a 6-term Taylor/Horner polynomial (error
// < 1e-27 for |x| < 1/32, evaluation rounding <= 1 ulp); larger arguments take the library cos.
__device__ __forceinline__ double cos_small(double x) {
  if (fabs(x) < 0.03125) {
    const double z = x * x;
    double p = -1.0 / 479001600.0;
    p = fma(p, z, 1.0 / 3628800.0);
    p = fma(p, z, -1.0 / 40320.0);
    p = fma(p, z, 1.0 / 720.0);
    p = fma(p, z, -1.0 / 24.0);
    p = fma(p, z, 0.5);
    return fma(-z, p, 1.0);
  }
  return cos(x);
}


// Distance bin of a Haversine `c` (public/Load_Data_by_length.py:32-39) through the exact host thresholds (data.bin_thresholds):
// bin = #{t : c >= thr[t]}, thr ascending (LDS or global).  asin(x) ~ x at these distances, so int(sqrt(c) * 12742e3 / dd) is within
// one bin of the answer; the thresholds then decide.
__device__ __forceinline__ int bin_of_c(double c, const double* thr, int n_dist, float scale) {
  int g = (int)(sqrtf((float)c) * scale);
  g = g < 0 ? 0 : (g > n_dist ? n_dist : g);
  while (g > 0 && c < thr[g - 1]) --g;
  while (g < n_dist && c >= thr[g]) ++g;
  return g;
}

// Batch rule of a launch (include/poi_hip.h): a table row touched by `nseq` sequences with L2 multiplicity `mult` moves by
//   row -= sc * (G + lm * row),   G = sum of the touching sequences' loss gradients.
// cap >= 1: sc = alpha min(nseq, cap) / nseq, lm = lambda mult (capped sum; 1 = mean of the sequences' reference updates).
// cap < 0 encodes the MINI-BATCH rule of public/GRU.py:452-466 for a launch of n = -cap sequences - cost = -sum(loss) / n + L2 over
// every gathered row: row -= alpha (G / n + lambda mult row), i.e. sc = alpha / n, lm = lambda mult n.
__device__ __forceinline__ void rule_scales(float alpha, float lambda, int nseq, int mult, float cap, float& sc, float& lm) {
  if (cap < 0.f) { sc = alpha / -cap; lm = lambda * (float)mult * -cap; }
  else { sc = alpha * fminf((float)nseq, cap) / (float)max(nseq, 1); lm = lambda * (float)mult; }
}


// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is an attribute of the function ON ONE DEVICE: a process with contexts on several devices
// must opt in on each of them, and two host threads may race to it.  One flag per device, under a mutex.
struct DeviceOnce {
  std::mutex mu;
  bool done[64] = {};
  template <class F> hipError_t run(F&& f) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev < 0 || dev >= 64) return hipErrorInvalidDevice;
    std::lock_guard<std::mutex> lock(mu);
    if (done[dev]) return hipSuccess;
    e = f();
    if (e == hipSuccess) done[dev] = true;
    return e;
  }
};
}  // namespace poi
