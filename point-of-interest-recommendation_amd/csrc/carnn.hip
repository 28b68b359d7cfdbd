// CA-RNN (flag 3 of prog_bpr_gru_spatial.py:141-151): OboCARNN of public/CA_RNN.py:46-227 - a sigmoid RNN whose
// recurrent matrix is selected per step by the distance interval of the hop (wd[(n_dist+1), H, D]), trained with a BPR
// loss whose scores are bilinear in the NEXT interval matrix:
//      h_t = sigmoid(M x_t + wd[dp_t] h_{t-1})
//      y_t = (wd[dp_{t+1}] h_t) . (M xp_{t+1}) - (wd[dq_{t+1}] h_t) . (M xq_{t+1}),   loss -= log sigmoid(y_t)
// One 256-thread workgroup walks one sequence (persistent grid over the launch), same structure as seq_engine.hip:
// rows gathered from the tables into LDS, GEMVs with one wavefront per matrix row (coalesced row reads, DPP wave sums),
// BPTT, sparse gradients of lt rows and of whole interval MATRICES (a matrix is one "row" of H*D floats of the wd
// table) into zero-initialised gradient tables with float atomics, the dense gradient of M into the workgroup's slab.
// Batch rule of include/poi_hip.h (n_seq == 1: exactly the reference step).  Backward as derived in
// oracle/poi_oracle.py::carnn_step (checked against autograd).
#include "poi_common.h"
#include "poi_kernels.h"
#include "seq_common.h"

namespace poi {

struct CaLds {
  float *x, *xp, *xq, *hp, *h, *mp, *mq, *vp, *vq, *dh, *da, *t0, *t1, *t2, *t3, *part, *red;
  __device__ CaLds(float* b, int D) {
    float* q = b;
    x = q; q += D; xp = q; q += D; xq = q; q += D; hp = q; q += D; h = q; q += D;
    mp = q; q += D; mq = q; q += D; vp = q; q += D; vq = q; q += D; dh = q; q += D; da = q; q += D;
    t0 = q; q += D; t1 = q; q += D; t2 = q; q += D; t3 = q; q += D;
    part = q; q += 1024; red = q; q += 8;
  }
};
__host__ __device__ inline int ca_lds_floats(int D) { return 15 * D + 1024 + 8; }

// G[r][c] += s * a[r] * b[c]   (rows x cols, row-major), a / b in LDS.  ATOMIC: the target is shared between workgroups.
template <bool ATOMIC>
__device__ __forceinline__ void outer_add(float* __restrict__ G, int rows, int cols, float s, const float* a, const float* b) {
  const int c4n = cols >> 2;
  for (int e = threadIdx.x; e < rows * c4n; e += POI_BLOCK) {
    const int r = e / c4n, c = (e % c4n) * 4;
    const float ar = s * a[r];
    const float4 bv = *reinterpret_cast<const float4*>(b + c);
    float* g = G + (size_t)r * cols + c;
    if (ATOMIC) { atomicAdd(g, ar * bv.x); atomicAdd(g + 1, ar * bv.y); atomicAdd(g + 2, ar * bv.z); atomicAdd(g + 3, ar * bv.w); }
    else { float4 v = *reinterpret_cast<float4*>(g); v.x += ar * bv.x; v.y += ar * bv.y; v.z += ar * bv.z; v.w += ar * bv.w; *reinterpret_cast<float4*>(g) = v; }
  }
}

__global__ __launch_bounds__(POI_BLOCK) void carnn_train_kernel(CaArgs A) {
  extern __shared__ __align__(16) float lds_raw[];
  const int D = A.dim, HD = D * D, tid = threadIdx.x;
  CaLds S(lds_raw, D);
  float* ws = A.ws + (size_t)blockIdx.x * A.ws_stride;
  float* wsH = ws;                                   // (cap + 1) x D, H[0] = h0 = 0
  float* wsV = wsH + (size_t)(A.cap + 1) * D;        // cap x 4D : mp | mq | vp | vq
  float* wsY = wsV + (size_t)A.cap * 4 * D;          // cap
  float* slab = A.slab + (size_t)blockIdx.x * HD;    // d M of this workgroup's sequences
  for (int k = blockIdx.x; k < A.n_seq; k += gridDim.x) {
    const int u = A.uidx[k], base = A.off[u], L = A.off[u + 1] - base, ns = L > 0 ? L - 1 : 0;
    const int *p = A.p + base, *q = A.q + base, *dp = A.dp + base, *dq = A.dq + base;
    // table-touch bookkeeping: lt rows of p U q, interval matrices of dp U dq, both padded to len_max (public/CA_RNN.py:147)
    count_rows<true>(p, q, L, A.n_item, 2 * (A.len_max - L), A.mult_lt, A.nseq_lt);
    count_rows<true>(dp, dq, L, A.n_dist, 2 * (A.len_max - L), A.mult_wd, A.nseq_wd);
    for (int j = tid; j < D; j += POI_BLOCK) { S.hp[j] = 0.f; wsH[j] = 0.f; }
    float tot = 0.f;       // meaningful in thread 0
    __syncthreads();
    // ------------------------------------------------------------------ forward
    for (int t = 0; t < ns; ++t) {
      load_row4(S.x, A.lt + (size_t)p[t] * D, D);
      load_row4(S.xp, A.lt + (size_t)p[t + 1] * D, D);
      load_row4(S.xq, A.lt + (size_t)q[t + 1] * D, D);
      __syncthreads();
      gemv_rows<1>(A.M, D, S.x, A.wd + (size_t)dp[t] * HD, D, S.hp, nullptr, D, S.h);         // h_t  (:131)
      gemv_rows<0>(A.M, D, S.xp, nullptr, 0, nullptr, nullptr, D, S.mp);
      gemv_rows<0>(A.M, D, S.xq, nullptr, 0, nullptr, nullptr, D, S.mq);
      __syncthreads();
      gemv_rows<0>(A.wd + (size_t)dp[t + 1] * HD, D, S.h, nullptr, 0, nullptr, nullptr, D, S.vp);
      gemv_rows<0>(A.wd + (size_t)dq[t + 1] * HD, D, S.h, nullptr, 0, nullptr, nullptr, D, S.vq);
      __syncthreads();
      float part = 0.f;
      for (int j = tid; j < D; j += POI_BLOCK) {
        part += S.vp[j] * S.mp[j] - S.vq[j] * S.mq[j];                                          // yp - yq (:132-133)
        wsH[(size_t)(t + 1) * D + j] = S.h[j];
        float* v = wsV + (size_t)t * 4 * D;
        v[j] = S.mp[j]; v[D + j] = S.mq[j]; v[2 * D + j] = S.vp[j]; v[3 * D + j] = S.vq[j];
        S.hp[j] = S.h[j];
      }
      const float y = block_sum(part, S.red);
      if (tid == 0) { wsY[t] = y; tot += log_sigmoidf_(y); }
      __syncthreads();
    }
    if (tid == 0) A.out[k] = -tot;                                                              // los (:148)
    // ------------------------------------------------------------------ backward (BPTT)
    for (int j = tid; j < D; j += POI_BLOCK) S.dh[j] = 0.f;
    __syncthreads();
    for (int t = ns - 1; t >= 0; --t) {
      const float g = -sigmoidf_(-wsY[t]);
      const float* v = wsV + (size_t)t * 4 * D;
      load_row4(S.mp, v, D); load_row4(S.mq, v + D, D); load_row4(S.vp, v + 2 * D, D); load_row4(S.vq, v + 3 * D, D);
      load_row4(S.h, wsH + (size_t)(t + 1) * D, D); load_row4(S.hp, wsH + (size_t)t * D, D);
      load_row4(S.x, A.lt + (size_t)p[t] * D, D);
      load_row4(S.xp, A.lt + (size_t)p[t + 1] * D, D);
      load_row4(S.xq, A.lt + (size_t)q[t + 1] * D, D);
      __syncthreads();
      const float* Wp = A.wd + (size_t)dp[t + 1] * HD; const float* Wq = A.wd + (size_t)dq[t + 1] * HD;
      const float* Wt = A.wd + (size_t)dp[t] * HD;
      gemv_cols<false>(Wp, D, D, S.mp, S.t0, S.part);        // Wp^T mp
      gemv_cols<false>(Wq, D, D, S.mq, S.t1, S.part);        // Wq^T mq
      gemv_cols<false>(A.M, D, D, S.vp, S.t2, S.part);       // M^T vp
      gemv_cols<false>(A.M, D, D, S.vq, S.t3, S.part);       // M^T vq
      {
        float* gp1 = A.g_lt + (size_t)p[t + 1] * D; float* gq1 = A.g_lt + (size_t)q[t + 1] * D;
        for (int j = tid; j < D; j += POI_BLOCK) {
          const float d = S.dh[j] + g * (S.t0[j] - S.t1[j]);
          S.dh[j] = d;
          S.da[j] = d * S.h[j] * (1.0f - S.h[j]);
          atomicAdd(gp1 + j, g * S.t2[j]);
          atomicAdd(gq1 + j, -g * S.t3[j]);
        }
      }
      __syncthreads();
      outer_add<true>(A.g_wd + (size_t)dp[t + 1] * HD, D, D, g, S.mp, S.h);
      outer_add<true>(A.g_wd + (size_t)dq[t + 1] * HD, D, D, -g, S.mq, S.h);
      outer_add<true>(A.g_wd + (size_t)dp[t] * HD, D, D, 1.0f, S.da, S.hp);
      outer_add<false>(slab, D, D, g, S.vp, S.xp);
      __syncthreads();
      outer_add<false>(slab, D, D, -g, S.vq, S.xq);
      __syncthreads();
      outer_add<false>(slab, D, D, 1.0f, S.da, S.x);
      gemv_cols<false>(A.M, D, D, S.da, S.t0, S.part);       // M^T da  -> d lt[p_t]
      gemv_cols<false>(Wt, D, D, S.da, S.t1, S.part);        // W_t^T da -> dh_{t-1}
      {
        float* gp0 = A.g_lt + (size_t)p[t] * D;
        for (int j = tid; j < D; j += POI_BLOCK) { atomicAdd(gp0 + j, S.t0[j]); S.dh[j] = S.t1[j]; }
      }
      __syncthreads();
    }
    __syncthreads();
  }
}

// write-back of the lt rows and of the interval matrices: one wavefront per table row (seq_common.h apply_row)
__global__ __launch_bounds__(POI_BLOCK) void carnn_rows_apply_kernel(CaArgs A, float alpha, float lambda) {
  const int n_lt = A.n_item + 1, n_wd = A.n_dist + 1, HD = A.dim * A.dim;
  for (int r = blockIdx.x * POI_NWAVE + wave_id(); r < n_lt + n_wd; r += gridDim.x * POI_NWAVE) {
    if (r < n_lt) apply_row(A.lt, A.g_lt, A.mult_lt, A.nseq_lt, r, A.dim, alpha, lambda, A.bcap);
    else apply_row(A.wd, A.g_wd, A.mult_wd, A.nseq_wd, r - n_lt, HD, alpha, lambda, A.bcap);
  }
}

// M <- M - alpha * min(n, cap) * (mean_k d M_k + lambda * M)   (public/CA_RNN.py:151-152 for n_seq == 1); slabs re-zeroed
__global__ __launch_bounds__(POI_BLOCK) void carnn_dense_apply_kernel(CaArgs A, int n_slab, float alpha, float lambda) {
  const int HD = A.dim * A.dim, i = blockIdx.x * POI_BLOCK + threadIdx.x;
  if (i >= HD) return;
  float g = 0.f;
  for (int s = 0; s < n_slab; ++s) { float* p = A.slab + (size_t)s * HD + i; g += *p; *p = 0.f; }
  g /= (float)A.n_seq;
  const float a = alpha * fminf((float)A.n_seq, A.bcap), v = A.M[i];
  A.M[i] = v - a * (g + lambda * v);
}

// row sums of every interval matrix (predict: public/CA_RNN.py:191 adds-then-sums) and the sum of all its elements
// (scoring: :97-100):  wrs[b][i] = sum_k wd[b][i][k],  wsum[b] = sum_i wrs[b][i];  msum[k] = sum_i M[i][k]
__global__ __launch_bounds__(POI_BLOCK) void carnn_sums_kernel(const float* __restrict__ wd, const float* __restrict__ M, int n_dist, int D,
                                                               float* __restrict__ wrs, float* __restrict__ wsum, float* __restrict__ msum) {
  __shared__ float red[8];
  const int b = blockIdx.x;
  if (b <= n_dist) {
    float tot = 0.f;
    for (int i0 = 0; i0 < D; i0 += POI_NWAVE) {
      const int i = i0 + wave_id();
      float s = 0.f;
      if (i < D) for (int k = lane_id(); k < D; k += 64) s += wd[((size_t)b * D + i) * D + k];
      s = wave_sum(s);
      if (i < D && lane_id() == 0) { if (wrs) wrs[(size_t)b * D + i] = s; tot += s; }
    }
    tot = block_sum(lane_id() == 0 ? tot : 0.f, red);
    if (threadIdx.x == 0 && wsum) wsum[b] = tot;
  } else if (msum) {
    for (int k = threadIdx.x; k < D; k += POI_BLOCK) { float s = 0.f; for (int i = 0; i < D; ++i) s += M[(size_t)i * D + k]; msum[k] = s; }
  }
}

// seq_predict (public/CA_RNN.py:172-217), literally: h_t = sigmoid(M p_t + rowsum(wd[d_t]) + sum(h_{t-1})) over all L positions
__global__ __launch_bounds__(POI_BLOCK) void carnn_predict_kernel(CaArgs A, const float* __restrict__ wrs) {
  extern __shared__ __align__(16) float lds_raw[];
  const int D = A.dim, tid = threadIdx.x;
  CaLds S(lds_raw, D);
  for (int k = blockIdx.x; k < A.n_seq; k += gridDim.x) {
    const int u = A.uidx[k], base = A.off[u], L = A.off[u + 1] - base;
    for (int j = tid; j < D; j += POI_BLOCK) S.h[j] = 0.f;
    __syncthreads();
    for (int t = 0; t < L; ++t) {
      load_row4(S.x, A.lt + (size_t)A.p[base + t] * D, D);
      float part = 0.f;
      for (int j = tid; j < D; j += POI_BLOCK) part += S.h[j];
      const float hs = block_sum(part, S.red);                  // (contains the barrier that publishes S.x)
      gemv_rows<0>(A.M, D, S.x, nullptr, 0, nullptr, nullptr, D, S.t0);
      __syncthreads();
      const float* wr = wrs + (size_t)A.dp[base + t] * D;
      for (int j = tid; j < D; j += POI_BLOCK) S.h[j] = sigmoidf_(S.t0[j] + wr[j] + hs);
      __syncthreads();
    }
    for (int j = tid; j < D; j += POI_BLOCK) A.hts[(size_t)k * D + j] = S.h[j];
    __syncthreads();
  }
}

// m[j] = sum_i (M x_j)_i = msum . x_j  for every POI (one wavefront per POI)
__global__ __launch_bounds__(POI_BLOCK) void carnn_item_term_kernel(const float* __restrict__ items, const float* __restrict__ msum, int N, int D,
                                                                    float* __restrict__ m) {
  for (int j = blockIdx.x * POI_NWAVE + wave_id(); j < N; j += gridDim.x * POI_NWAVE) {
    float s = 0.f;
    for (int k = lane_id(); k < D; k += 64) s += items[(size_t)j * D + k] * msum[k];
    s = wave_sum(s);
    if (lane_id() == 0) m[j] = s;
  }
}

// compute_sub_all_scores (public/CA_RNN.py:91-101), literally: score[u][j] = -(wsum[bin(u, j)] + H * sum(user_u) + m[j]) with
// bin(u, j) = usrs_last_poi_to_all_intervals[u][j] computed on the fly from the coordinates with the exact host thresholds
// (same arithmetic as dist_prob_kernel in misc.hip; the U x N bin matrix of the reference is never materialised)
__global__ __launch_bounds__(POI_BLOCK) void carnn_score_kernel(const float* __restrict__ users, const float* __restrict__ m, const float* __restrict__ wsum,
                                                                const double* __restrict__ coords, const double* __restrict__ cphi,
                                                                const double* __restrict__ thr, const int* __restrict__ last_poi, int n, int N, int D,
                                                                int n_dist, double dd, float* __restrict__ out) {
  extern __shared__ __align__(16) double s_thr[];       // n_dist thresholds, then n_dist + 1 interval sums (float)
  __shared__ float red[8];
  const int k = blockIdx.y;
  float* s_w = reinterpret_cast<float*>(s_thr + n_dist);
  for (int i = threadIdx.x; i < n_dist; i += POI_BLOCK) s_thr[i] = thr[i];
  for (int i = threadIdx.x; i <= n_dist; i += POI_BLOCK) s_w[i] = wsum[i];
  float part = 0.f;
  for (int j = threadIdx.x; j < D; j += POI_BLOCK) part += users[(size_t)k * D + j];
  const float su = (float)D * block_sum(part, red);
  const int lp = last_poi[k];
  const double lat1 = coords[2 * lp], lon1 = coords[2 * lp + 1], c1 = cphi[lp], pr = 0.017453292519943295;
  const float scale = (float)(12742.0 * 1000.0 / dd);
  for (int j = blockIdx.x * POI_BLOCK + threadIdx.x; j < N; j += gridDim.x * POI_BLOCK) {
    int bin;
    {
#pragma clang fp contract(off)
      const double a = (lat1 - coords[2 * j]) * pr;
      const double b = (lon1 - coords[2 * j + 1]) * pr;
      const double c = (1.0 - cos_small(a)) / 2 + c1 * cphi[j] * (1.0 - cos_small(b)) / 2;
      bin = bin_of_c(c, s_thr, n_dist, scale);
    }
    out[(size_t)k * N + j] = -((s_w[bin] + su) + m[j]);
  }
}

// ---------------------------------------------------------------------------------------------------------------
size_t carnn_ws_floats(int D, int cap) { return (size_t)(cap + 1) * D + (size_t)cap * 4 * D + (size_t)cap + 16; }

hipError_t launch_carnn_train(const CaArgs& A, int grid, float alpha, float lambda, hipStream_t st, Timing* tm) {
  const int D = A.dim;
  tm->begin("carnn_train", st);
  hipLaunchKernelGGL(carnn_train_kernel, dim3(grid), dim3(POI_BLOCK), sizeof(float) * ca_lds_floats(D), st, A);
  int rows = A.n_item + 1 + A.n_dist + 1, g2 = (rows + POI_NWAVE - 1) / POI_NWAVE;
  if (g2 > 8192) g2 = 8192;
  hipLaunchKernelGGL(carnn_rows_apply_kernel, dim3(g2), dim3(POI_BLOCK), 0, st, A, alpha, lambda);
  hipLaunchKernelGGL(carnn_dense_apply_kernel, dim3((D * D + POI_BLOCK - 1) / POI_BLOCK), dim3(POI_BLOCK), 0, st, A, grid, alpha, lambda);
  tm->end(st);
  return hipGetLastError();
}

hipError_t launch_carnn_predict(const CaArgs& A, int grid, float* wrs, hipStream_t st, Timing* tm) {
  tm->begin("carnn_predict", st);
  hipLaunchKernelGGL(carnn_sums_kernel, dim3(A.n_dist + 1), dim3(POI_BLOCK), 0, st, A.wd, A.M, A.n_dist, A.dim, wrs, (float*)nullptr, (float*)nullptr);
  hipLaunchKernelGGL(carnn_predict_kernel, dim3(grid), dim3(POI_BLOCK), sizeof(float) * ca_lds_floats(A.dim), st, A, (const float*)wrs);
  tm->end(st);
  return hipGetLastError();
}

hipError_t launch_carnn_score(const float* users, const float* items, const float* M, const float* dists, const double* coords, const double* cphi,
                              const double* thr, const int* last_poi, int n, int N, int n_dist, int D, double dd, float* scratch, float* out,
                              hipStream_t st, Timing* tm) {
  // scratch: wsum (n_dist + 1, padded to 256) | msum (D) | m (N)
  float* wsum = scratch; float* msum = scratch + ((n_dist + 1 + 255) & ~255); float* m = msum + ((D + 255) & ~255);
  tm->begin("carnn_score", st);
  hipLaunchKernelGGL(carnn_sums_kernel, dim3(n_dist + 2), dim3(POI_BLOCK), 0, st, dists, M, n_dist, D, (float*)nullptr, wsum, msum);
  int g = (N + POI_NWAVE - 1) / POI_NWAVE; if (g > 4096) g = 4096;
  hipLaunchKernelGGL(carnn_item_term_kernel, dim3(g), dim3(POI_BLOCK), 0, st, items, (const float*)msum, N, D, m);
  int gx = (N + POI_BLOCK * 4 - 1) / (POI_BLOCK * 4); if (gx < 1) gx = 1;
  const size_t lds = sizeof(double) * n_dist + sizeof(float) * (n_dist + 1) + 16;
  hipLaunchKernelGGL(carnn_score_kernel, dim3(gx, n), dim3(POI_BLOCK), lds, st, users, (const float*)m, (const float*)wsum, coords, cphi, thr, last_poi,
                     n, N, D, n_dist, dd, out);
  tm->end(st);
  return hipGetLastError();
}

}  // namespace poi
