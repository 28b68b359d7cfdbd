// Device helpers shared by the per-sequence kernels (seq_engine.hip, carnn.hip): row staging, table-touch bookkeeping
// of the batch rule, and the per-row SGD write-back.
#pragma once
#include "poi_common.h"

namespace poi {

__device__ __forceinline__ void load_row4(float* dst, const float* __restrict__ src, int n) {
  for (int j = threadIdx.x * 4; j < n; j += POI_BLOCK * 4)
    *reinterpret_cast<float4*>(dst + j) = *reinterpret_cast<const float4*>(src + j);
}


// Count the sequence's table touches: multiplicity (every occurrence; L2 weight of the reference's
// multiplicity-weighted row decay) and distinct-sequence count (batch mean rule), plus the analytic
// padding-row terms.  ids: two concatenated id lists of length L each (second may be null).
template <bool TWO>
__device__ __forceinline__ int row_at(const int* __restrict__ a, const int* __restrict__ b, int L, int e) {
  if (TWO) return e < L ? a[e] : b[e - L];
  return a[e];
}
template <bool TWO>
__device__ __forceinline__ void count_rows(const int* __restrict__ a, const int* __restrict__ b, int L,
                                           int pad_row, int pad_mult, int* __restrict__ mult, int* __restrict__ nseq) {
  const int n = TWO ? 2 * L : L;
  for (int e = threadIdx.x; e < n; e += POI_BLOCK) {
    const int row = row_at<TWO>(a, b, L, e);
    atomicAdd(&mult[row], 1);
    int dup = 0;
    for (int j = 0; j < e; ++j) dup |= (row_at<TWO>(a, b, L, j) == row) ? 1 : 0;
    if (!dup) atomicAdd(&nseq[row], 1);
  }
  if (pad_mult > 0 && threadIdx.x == 0) {
    int seen = 0;
    for (int j = 0; j < n; ++j) seen |= (row_at<TWO>(a, b, L, j) == pad_row) ? 1 : 0;
    atomicAdd(&mult[pad_row], pad_mult);
    if (!seen) atomicAdd(&nseq[pad_row], 1);
  }
}


// One wavefront per table row: row <- row - sc * (G[row] + lm * row) with the batch rule's scales (rule_scales); G / mult / nseq are
// re-zeroed.  `D` is the row width in floats (a CA-RNN interval matrix is one row of H * D floats).
__device__ __forceinline__ void apply_row(float* __restrict__ T, float* __restrict__ G, int* __restrict__ mult,
                                          int* __restrict__ nseq, int row, int D, float alpha, float lambda, float cap) {
  const int got = nseq[row];
  if (got <= 0) return;
  const int m = mult[row];
  float sc, lm; rule_scales(alpha, lambda, got, m, cap, sc, lm);
  float* t = T + (size_t)row * D;
  float* g = G + (size_t)row * D;
  for (int j = lane_id() * 4; j < D; j += 256) {
    float4 tv = *reinterpret_cast<float4*>(t + j);
    const float4 gv = *reinterpret_cast<float4*>(g + j);
    tv.x -= sc * (gv.x + lm * tv.x); tv.y -= sc * (gv.y + lm * tv.y);
    tv.z -= sc * (gv.z + lm * tv.z); tv.w -= sc * (gv.w + lm * tv.w);
    *reinterpret_cast<float4*>(t + j) = tv;
    *reinterpret_cast<float4*>(g + j) = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  if (lane_id() == 0) { nseq[row] = 0; mult[row] = 0; }
}


}  // namespace poi
