// Per-sequence engine ("parity engine"): one 256-thread workgroup walks one user sequence at a time
// (persistent grid over the launch's sequences): gathers the POI / distance-bin rows straight from
// the HBM tables into LDS, runs the GRU cell with wavefront-reduced GEMVs, the distance-softmax
// head, the BPR + survival losses, BPTT, and accumulates
//   - sparse row gradients into the zero-initialised gradient tables (float atomics),
//   - dense gradients into the workgroup's private slab (plain read-modify-write).
// rows_apply_kernel / dense_apply_kernel then perform the SGD write-back.
//
// Math: public/GRU_Spatial.py:127-229 (SPATIAL) and public/GRU.py:313-389 (plain); backward as
// derived in SURVEY.md 2.1 and checked by oracle/poi_oracle.py (tests/test_oracle_autograd.py).
#include "poi_common.h"
#include "poi_kernels.h"
#include "seq_common.h"

namespace poi {

// ---------------------------------------------------------------------------------------------
// LDS carve-up (floats).  D = dim, XW = input width (2D spatial, D plain), NB = n_dist + 1.
// ---------------------------------------------------------------------------------------------
struct Lds {
  float *xs, *hcur, *rh, *act, *es, *os, *dh, *dacc, *mvec, *part, *red;
  __device__ Lds(float* base, int D, int XW, int NBpad) {
    float* q = base;
    xs = q; q += XW;
    hcur = q; q += D;
    rh = q; q += D;
    act = q; q += 3 * D;      // z | r | c   (forward)   /  da_z | da_r | da_c (backward)
    es = q; q += D;
    dh = q; q += D;
    dacc = q; q += XW;        // dx
    mvec = q; q += D;
    os = q; q += NBpad;       // head logits / softmax / d logits
    part = q; q += 1024;
    red = q; q += 8;
  }
};

__host__ __device__ inline int seq_lds_floats(int D, int XW, int NBpad) {
  return XW + D + D + 3 * D + D + D + XW + D + NBpad + 1024 + 8;
}

// out[row][col] += sum_t A[t*lda + row] * Bt(t)[col]; rows x cols slab block, cols % 4 == 0.
// BFN(t, col4) returns the float4 of the right-hand vector at step t.
template <typename BFN>
__device__ __forceinline__ void outer_acc(float* __restrict__ slab, int rows, int cols,
                                          const float* __restrict__ A, int lda, int nstep, BFN bfn) {
  const int c4n = cols >> 2;
  const int RG = POI_BLOCK / c4n > 0 ? POI_BLOCK / c4n : 1;
  const int tid = threadIdx.x;
  const int c = tid % c4n, rg = tid / c4n;
  if (rg >= RG) return;
  for (int r0 = rg; r0 < rows; r0 += 4 * RG) {
    float4 acc[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) acc[u] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int t = 0; t < nstep; ++t) {
      const float4 b = bfn(t, c);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int r = r0 + u * RG;
        const float a = r < rows ? A[(size_t)t * lda + r] : 0.f;
        acc[u].x = fmaf(a, b.x, acc[u].x); acc[u].y = fmaf(a, b.y, acc[u].y);
        acc[u].z = fmaf(a, b.z, acc[u].z); acc[u].w = fmaf(a, b.w, acc[u].w);
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int r = r0 + u * RG;
      if (r < rows) {
        float4* o = reinterpret_cast<float4*>(slab + (size_t)r * cols + 4 * c);
        float4 v = *o;
        v.x += acc[u].x; v.y += acc[u].y; v.z += acc[u].z; v.w += acc[u].w;
        *o = v;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// forward cell: consumes xs (LDS, XW) and hcur (LDS, D); leaves z|r|c in act and the new state in
// hcur; stores z, r, c, h to the workspace when ws pointers are non-null.  Contains barriers.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void cell_forward(const SeqArgs& A, Lds& S, int D, int XW,
                                             float* wsZ, float* wsR, float* wsC, float* wsH) {
  gemv_rows<1>(A.ui, XW, S.xs, A.wh, D, S.hcur, A.bi, 2 * D, S.act);
  __syncthreads();
  for (int j = threadIdx.x; j < D; j += POI_BLOCK) S.rh[j] = S.act[D + j] * S.hcur[j];
  __syncthreads();
  gemv_rows<2>(A.ui + (size_t)2 * D * XW, XW, S.xs, A.wh + (size_t)2 * D * D, D, S.rh, A.bi + 2 * D, D, S.act + 2 * D);
  __syncthreads();
  for (int j = threadIdx.x; j < D; j += POI_BLOCK) {
    const float z = S.act[j], r = S.act[D + j], c = S.act[2 * D + j], hp = S.hcur[j];
    const float hn = (1.0f - z) * hp + z * c;
    if (wsZ) { wsZ[j] = z; wsR[j] = r; wsC[j] = c; wsH[j] = hn; }
    S.hcur[j] = hn;
  }
  __syncthreads();
}

// softmax(vs.h + bs) over NB bins into S.os (LDS).  Contains barriers.
__device__ __forceinline__ void head_softmax(const SeqArgs& A, Lds& S, int D, int NB) {
  gemv_rows<0>(A.vs, D, S.hcur, nullptr, 0, nullptr, A.bs, NB, S.os);
  __syncthreads();
  float m = -INFINITY;
  for (int k = threadIdx.x; k < NB; k += POI_BLOCK) m = fmaxf(m, S.os[k]);
  m = block_max(m, S.red);
  float sum = 0.f;
  for (int k = threadIdx.x; k < NB; k += POI_BLOCK) { const float e = expf(S.os[k] - m); S.os[k] = e; sum += e; }
  sum = block_sum(sum, S.red);
  const float inv = 1.0f / sum;
  for (int k = threadIdx.x; k < NB; k += POI_BLOCK) S.os[k] *= inv;
  __syncthreads();
}

// ---------------------------------------------------------------------------------------------
// backward cell: dh (LDS) is d cost / d h_t; reads the stored z, r, c, h_{t-1}; writes da (3D) to
// S.act and to DA[t], dx to S.dacc, and d cost / d h_{t-1} back into S.dh.  Contains barriers.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void cell_backward(const SeqArgs& A, Lds& S, int D, int XW,
                                              const float* wsZ, const float* wsR, const float* wsC,
                                              const float* wsHp, float* wsDA) {
  // dim <= 256 (checked by the ABI): one hidden column per thread
  const int j = threadIdx.x;
  const bool on = j < D;
  float dz = 0.f, dhp = 0.f, z = 0.f, r = 0.f, hp = 0.f;
  if (on) {
    z = wsZ[j]; r = wsR[j]; hp = wsHp[j];
    const float c = wsC[j], d = S.dh[j];
    dz = d * (c - hp);
    dhp = d * (1.0f - z);
    S.act[2 * D + j] = d * z * (1.0f - c * c);          // da_c
  }
  __syncthreads();
  gemv_cols<false>(A.wh + (size_t)2 * D * D, D, D, S.act + 2 * D, S.mvec, S.part);   // m = wh[2]^T da_c
  if (on) {
    const float m = S.mvec[j];
    const float dr = m * hp;
    dhp += m * r;
    S.act[j] = dz * z * (1.0f - z);                      // da_z
    S.act[D + j] = dr * r * (1.0f - r);                  // da_r
  }
  __syncthreads();
  gemv_cols<false>(A.wh, 2 * D, D, S.act, S.mvec, S.part);                            // wh[0,1]^T da_{z,r}
  gemv_cols<false>(A.ui, 3 * D, XW, S.act, S.dacc, S.part);                           // dx = ui^T da
  if (on) S.dh[j] = dhp + S.mvec[j];
  for (int i = threadIdx.x; i < 3 * D; i += POI_BLOCK) wsDA[i] = S.act[i];
  __syncthreads();
}

// ---------------------------------------------------------------------------------------------
// training kernel
// ---------------------------------------------------------------------------------------------
template <bool SPATIAL>
__global__ __launch_bounds__(POI_BLOCK) void seq_train_kernel(SeqArgs A) {
  extern __shared__ __align__(16) float lds_raw[];
  const int D = A.dim, XW = SPATIAL ? 2 * D : D, NB = SPATIAL ? A.n_dist + 1 : 0;
  const int NBpad = (NB + 3) & ~3;
  Lds S(lds_raw, D, XW, NBpad);
  const int tid = threadIdx.x;
  const DenseLayout dl = dense_layout(D, XW, NB);

  // per-workgroup scratch
  float* ws = A.ws + (size_t)blockIdx.x * A.ws_stride;
  float* wsH = ws;                                   // (cap+1) x D ; H[0] = h0 = 0
  float* wsZ = wsH + (size_t)(A.cap + 1) * D;
  float* wsR = wsZ + (size_t)A.cap * D;
  float* wsC = wsR + (size_t)A.cap * D;
  float* wsDA = wsC + (size_t)A.cap * D;             // cap x 3D
  float* wsS = wsDA + (size_t)A.cap * 3 * D;         // cap x NBpad (softmax, then d logits)
  float* wsU = wsS + (size_t)A.cap * NBpad;          // cap
  float* slab = A.slab + (size_t)blockIdx.x * dl.total;

  float ls0 = 0.f, ls1 = 1.f, wd = 0.f;
  if (SPATIAL) {   // ls = softmax(loss_weight)  public/GRU_Spatial.py:156
    const float a = A.lw[0], b = A.lw[1], m = fmaxf(a, b);
    const float ea = expf(a - m), eb = expf(b - m);
    ls0 = ea / (ea + eb); ls1 = eb / (ea + eb);
    wd = A.wd[0];
  }

  for (int k = blockIdx.x; k < A.n_seq; k += gridDim.x) {
    const int u = A.uidx[k];
    const int base = A.off[u];
    const int L = A.off[u + 1] - base;
    const int* p = A.p + base;
    const int* q = A.q + base;
    const int* dp = SPATIAL ? A.dp + base : nullptr;
    const int* dq = SPATIAL ? A.dq + base : nullptr;
    const int nstep = SPATIAL ? (L > 0 ? L - 1 : 0) : L;

    // table-touch bookkeeping (multiplicity-weighted L2, batch mean rule, analytic padding rows)
    count_rows<true>(p, q, L, A.n_item, 2 * (A.len_max - L), A.mult_lt, A.nseq_lt);
    if (SPATIAL) count_rows<false>(dp, dp, L, A.n_dist, A.len_max - L, A.mult_di, A.nseq_di);

    for (int j = tid; j < D; j += POI_BLOCK) { S.hcur[j] = 0.f; wsH[j] = 0.f; }
    float sur = 0.f, bpr = 0.f;   // meaningful in thread 0
    __syncthreads();

    // ------------------------------------------------------------------ forward
    for (int t = 0; t < nstep; ++t) {
      const float* xp = A.lt + (size_t)p[t] * D;
      load_row4(S.xs, xp, D);
      if (SPATIAL) load_row4(S.xs + D, A.di + (size_t)dp[t] * D, D);
      if (!SPATIAL) {   // u_t = h_{t-1}.(xp_t - xq_t)   public/GRU.py:352
        const float* xq = A.lt + (size_t)q[t] * D;
        float part = 0.f;
        for (int j = tid; j < D; j += POI_BLOCK) part += S.hcur[j] * (xp[j] - xq[j]);
        const float ut = block_sum(part, S.red);
        if (tid == 0) { wsU[t] = ut; bpr += log_sigmoidf_(ut); }
      }
      __syncthreads();
      cell_forward(A, S, D, XW, wsZ + (size_t)t * D, wsR + (size_t)t * D, wsC + (size_t)t * D, wsH + (size_t)(t + 1) * D);
      if (SPATIAL) {
        head_softmax(A, S, D, NB);
        const int a = dp[t + 1], b = dq[t + 1];
        const float* xp1 = A.lt + (size_t)p[t + 1] * D;
        const float* xq1 = A.lt + (size_t)q[t + 1] * D;
        float part = 0.f, cum = 0.f;
        for (int j = tid; j < D; j += POI_BLOCK) part += S.hcur[j] * (xp1[j] - xq1[j]);
        for (int kk = tid; kk < NB; kk += POI_BLOCK) { const float s = S.os[kk]; wsS[(size_t)t * NBpad + kk] = s; if (kk <= a) cum += s; }
        const float he = block_sum(part, S.red);
        const float cs = block_sum(cum, S.red);
        if (tid == 0) {
          const float sa = S.os[a], sb = S.os[b];
          const float ut = he + wd * (sa - sb);           // public/GRU_Spatial.py:184
          wsU[t] = ut;
          bpr += log_sigmoidf_(ut);                       // :186
          sur += cs - logf(sa);                           // :189
        }
        __syncthreads();
      }
    }
    if (tid == 0) {
      if (SPATIAL) {
        const float upq = -bpr;
        float* o = A.out + (size_t)k * 5;
        o[0] = ls0 * sur + ls1 * upq; o[1] = sur; o[2] = upq; o[3] = ls0; o[4] = ls1;
        slab[dl.sur] += sur; slab[dl.upq] += upq;
      } else {
        A.out[k] = -bpr;
      }
    }

    // ------------------------------------------------------------------ backward (BPTT)
    for (int j = tid; j < D; j += POI_BLOCK) S.dh[j] = 0.f;
    __syncthreads();
    for (int t = nstep - 1; t >= 0; --t) {
      const float* h_t = wsH + (size_t)(t + 1) * D;
      const float* h_p = wsH + (size_t)t * D;
      float g_plain = 0.f;
      if (SPATIAL) {
        const int a = dp[t + 1], b = dq[t + 1];
        const float ut = wsU[t];
        const float g = -ls1 * sigmoidf_(-ut);
        float* st = wsS + (size_t)t * NBpad;
        const float sa = st[a], sb = st[b];
        float part = 0.f;
        for (int kk = tid; kk < NB; kk += POI_BLOCK) {
          const float s = st[kk];
          float ds = (kk <= a ? ls0 : 0.f);
          if (kk == a) ds += g * wd - ls0 / sa;
          if (kk == b) ds -= g * wd;
          S.os[kk] = ds;
          part += ds * s;
        }
        const float dot = block_sum(part, S.red);
        for (int kk = tid; kk < NB; kk += POI_BLOCK) {
          const float dlog = st[kk] * (S.os[kk] - dot);
          S.os[kk] = dlog;
          st[kk] = dlog;                      // kept for the deferred d vs
          slab[dl.bs + kk] += dlog;
        }
        if (tid == 0) slab[dl.wd] += g * (sa - sb);
        const float* xp1 = A.lt + (size_t)p[t + 1] * D;
        const float* xq1 = A.lt + (size_t)q[t + 1] * D;
        float* gp = A.g_lt + (size_t)p[t + 1] * D;
        float* gq = A.g_lt + (size_t)q[t + 1] * D;
        for (int j = tid; j < D; j += POI_BLOCK) {
          const float hv = h_t[j];
          S.dh[j] += g * (xp1[j] - xq1[j]);
          atomicAdd(gp + j, g * hv);
          atomicAdd(gq + j, -g * hv);
        }
        __syncthreads();
        gemv_cols<true>(A.vs, NB, D, S.os, S.dh, S.part);        // dh += vs^T d logits
      } else {
        g_plain = -sigmoidf_(-wsU[t]);
      }
      cell_backward(A, S, D, XW, wsZ + (size_t)t * D, wsR + (size_t)t * D, wsC + (size_t)t * D, h_p, wsDA + (size_t)t * 3 * D);
      {
        float* gp = A.g_lt + (size_t)p[t] * D;
        for (int j = tid; j < D; j += POI_BLOCK) atomicAdd(gp + j, S.dacc[j]);
        if (SPATIAL) {
          float* gd = A.g_di + (size_t)dp[t] * D;
          for (int j = tid; j < D; j += POI_BLOCK) atomicAdd(gd + j, S.dacc[D + j]);
        } else {
          const float* xp = A.lt + (size_t)p[t] * D;
          const float* xq = A.lt + (size_t)q[t] * D;
          float* gq = A.g_lt + (size_t)q[t] * D;
          for (int j = tid; j < D; j += POI_BLOCK) {
            const float hv = h_p[j];
            atomicAdd(gp + j, g_plain * hv);
            atomicAdd(gq + j, -g_plain * hv);
            S.dh[j] += g_plain * (xp[j] - xq[j]);
          }
        }
      }
      __syncthreads();
    }

    // ------------------------------------------------------------------ dense gradients (deferred outer products)
    if (nstep > 0) {
      const float* lt = A.lt; const float* di = A.di;
      // d ui (3D x XW) = sum_t da_t (x) x_t, x_t re-gathered from the tables (L1/L2 hits)
      outer_acc(slab + dl.ui, 3 * D, XW, wsDA, 3 * D, nstep, [&](int t, int c) {
        const int col = 4 * c;
        const float* src = (!SPATIAL || col < D) ? lt + (size_t)p[t] * D + col : di + (size_t)dp[t] * D + (col - D);
        return *reinterpret_cast<const float4*>(src);
      });
      // d wh[0,1] (2D x D) = sum_t da_{z,r} (x) h_{t-1}
      outer_acc(slab + dl.wh, 2 * D, D, wsDA, 3 * D, nstep, [&](int t, int c) {
        return *reinterpret_cast<const float4*>(wsH + (size_t)t * D + 4 * c);
      });
      // d wh[2] (D x D) = sum_t da_c (x) (r_t * h_{t-1})
      outer_acc(slab + dl.wh + (size_t)2 * D * D, D, D, wsDA + 2 * D, 3 * D, nstep, [&](int t, int c) {
        const float4 h = *reinterpret_cast<const float4*>(wsH + (size_t)t * D + 4 * c);
        const float4 r = *reinterpret_cast<const float4*>(wsR + (size_t)t * D + 4 * c);
        return make_float4(h.x * r.x, h.y * r.y, h.z * r.z, h.w * r.w);
      });
      for (int r = tid; r < 3 * D; r += POI_BLOCK) {
        float s = 0.f;
        for (int t = 0; t < nstep; ++t) s += wsDA[(size_t)t * 3 * D + r];
        slab[dl.bi + r] += s;
      }
      if (SPATIAL)   // d vs (NB x D) = sum_t d logits_t (x) h_t
        outer_acc(slab + dl.vs, NB, D, wsS, NBpad, nstep, [&](int t, int c) {
          return *reinterpret_cast<const float4*>(wsH + (size_t)(t + 1) * D + 4 * c);
        });
    }
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------------------------
// predict kernel: forward over ALL L positions, hts = h_{L-1}, sts = softmax(vs.h + bs)
// public/GRU_Spatial.py:231-288, public/GRU.py:154-205
// ---------------------------------------------------------------------------------------------
template <bool SPATIAL>
__global__ __launch_bounds__(POI_BLOCK) void seq_predict_kernel(SeqArgs A) {
  extern __shared__ __align__(16) float lds_raw[];
  const int D = A.dim, XW = SPATIAL ? 2 * D : D, NB = SPATIAL ? A.n_dist + 1 : 0;
  const int NBpad = (NB + 3) & ~3;
  Lds S(lds_raw, D, XW, NBpad);
  const int tid = threadIdx.x;
  for (int k = blockIdx.x; k < A.n_seq; k += gridDim.x) {
    const int u = A.uidx[k];
    const int base = A.off[u];
    const int L = A.off[u + 1] - base;
    for (int j = tid; j < D; j += POI_BLOCK) S.hcur[j] = 0.f;
    __syncthreads();
    for (int t = 0; t < L; ++t) {
      load_row4(S.xs, A.lt + (size_t)A.p[base + t] * D, D);
      if (SPATIAL) load_row4(S.xs + D, A.di + (size_t)A.dp[base + t] * D, D);
      __syncthreads();
      cell_forward(A, S, D, XW, nullptr, nullptr, nullptr, nullptr);
    }
    const int ko = A.out_row ? A.out_row[k] : k;
    for (int j = tid; j < D; j += POI_BLOCK) A.hts[(size_t)ko * D + j] = S.hcur[j];
    if (SPATIAL && A.sts) {
      head_softmax(A, S, D, NB);
      for (int kk = tid; kk < NB; kk += POI_BLOCK) A.sts[(size_t)ko * NB + kk] = S.os[kk];
    }
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------------------------
// sparse write-back: every table touch of the launch tries to claim its row (atomicExch on the
// distinct-sequence counter); the single winner applies
//     row <- row - alpha * (G[row] + lambda * mult[row] * row) / nseq[row]
// and re-zeroes G / mult.  With n_seq == 1 this is the reference's unique(p U q) write-back
// (public/GRU_Spatial.py:149-153,212-215) including the padding rows.
// ---------------------------------------------------------------------------------------------
// One wavefront per table row: rows whose distinct-sequence counter is non-zero were touched by this
// launch and get   row <- row - alpha * (G[row] + lambda * mult[row] * row) / nseq[row]   ; G / mult /
// nseq are re-zeroed.  No atomics (each row has exactly one owner), deterministic, and the scan of
// the counters costs 4 bytes per table row.
template <bool SPATIAL>
__global__ __launch_bounds__(POI_BLOCK) void rows_apply_kernel(SeqArgs A, float alpha, float lambda) {
  const int D = A.dim;
  const int n_lt = A.n_item + 1, n_di = SPATIAL ? A.n_dist + 1 : 0;
  for (int r = blockIdx.x * POI_NWAVE + wave_id(); r < n_lt + n_di; r += gridDim.x * POI_NWAVE) {
    if (r < n_lt) apply_row(A.lt, A.g_lt, A.mult_lt, A.nseq_lt, r, D, alpha, lambda, A.bcap);
    else apply_row(A.di, A.g_di, A.mult_di, A.nseq_di, r - n_lt, D, alpha, lambda, A.bcap);
  }
}

// ---------------------------------------------------------------------------------------------
// dense write-back: theta <- theta - alpha * (mean_k grad_k + lambda * theta); slabs re-zeroed.
// public/GRU_Spatial.py:210-211 (n_seq == 1: identical).
// ---------------------------------------------------------------------------------------------
template <bool SPATIAL>
__global__ __launch_bounds__(POI_BLOCK) void dense_apply_kernel(SeqArgs A, int n_slab, int n_slab_head, float alpha, float lambda) {
  const int D = A.dim, XW = SPATIAL ? 2 * D : D, NB = SPATIAL ? A.n_dist + 1 : 0;
  const DenseLayout dl = dense_layout(D, XW, NB);
  const float inv_n = 1.0f / (float)A.n_seq;
  alpha *= A.bcap < 0.f ? 1.0f : fminf((float)A.n_seq, A.bcap);      // batch rule: min(n, cap) of the n sequences' updates count (cap = 1: their mean == the mini-batch rule for dense tensors)
  const int i = blockIdx.x * POI_BLOCK + threadIdx.x;
  if (i >= dl.total) return;
  if (SPATIAL && i == dl.upq) return;   // consumed together with dl.sur by one thread (below)
  // regions vs | bs | wd are written by up to n_slab_head workgroups, everything else by n_slab - unless te_wgrad chose the K-chunk
  // counts on the device (A.kc_dev = {chunks of the other jobs, chunks of the d ui jobs}: per-POI regrouping)
  if (A.kc_dev) { n_slab = n_slab_head = A.kc_dev[0]; }
  const int ns = (i >= dl.vs && i <= dl.wd) ? n_slab_head : (i < dl.wh && A.kc_dev) ? A.kc_dev[1] : n_slab;
  float g = 0.f;
  {
    float* base = A.slab + i;
    const size_t st = dl.total;
    int s = 0;
    for (; s + 16 <= ns; s += 16) {         // the tile engine splits K into ~100 chunks (one slab each): sixteen loads in flight
      float* p0 = base + (size_t)s * st;
      float v[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) v[u] = p0[(size_t)u * st];
#pragma unroll
      for (int u = 0; u < 16; ++u) p0[(size_t)u * st] = 0.f;
      g += (((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]))) + (((v[8] + v[9]) + (v[10] + v[11])) + ((v[12] + v[13]) + (v[14] + v[15])));
    }
    for (; s + 4 <= ns; s += 4) {           // four independent loads in flight
      float* p0 = base + (size_t)s * st;
      const float v0 = p0[0], v1 = p0[st], v2 = p0[2 * st], v3 = p0[3 * st];
      p0[0] = 0.f; p0[st] = 0.f; p0[2 * st] = 0.f; p0[3 * st] = 0.f;
      g += (v0 + v1) + (v2 + v3);
    }
    for (; s < ns; ++s) { float* p0 = base + (size_t)s * st; g += *p0; *p0 = 0.f; }
  }
  g *= inv_n;
  float* theta = nullptr;
  if (i < dl.wh) theta = A.ui + (i - dl.ui);
  else if (i < dl.bi) theta = A.wh + (i - dl.wh);
  else if (i < dl.vs) theta = A.bi + (i - dl.bi);
  else if (i < dl.bs) theta = A.vs + (i - dl.vs);
  else if (i < dl.wd) theta = A.bs + (i - dl.bs);
  else if (i == dl.wd) { if (SPATIAL) theta = A.wd; }
  if (theta) { const float v = *theta; *theta = v - alpha * (g + lambda * v); return; }
  if (SPATIAL && i == dl.sur) {
    // loss_weight: d ls = [mean sur, mean upq] + lambda * ls ; d lw = ls * (d ls - d ls . ls)
    // (thread dl.sur also consumes the upq slot to keep the update atomic w.r.t. the old values)
    float upq = 0.f;
    for (int s = 0; s < n_slab; ++s) { float* ptr = A.slab + (size_t)s * dl.total + dl.upq; upq += *ptr; *ptr = 0.f; }
    upq *= inv_n;
    const float a = A.lw[0], b = A.lw[1], m = fmaxf(a, b);
    const float ea = expf(a - m), eb = expf(b - m);
    const float ls0 = ea / (ea + eb), ls1 = eb / (ea + eb);
    const float d0 = g + lambda * ls0, d1 = upq + lambda * ls1;
    const float dot = d0 * ls0 + d1 * ls1;
    A.lw[0] = a - alpha * ls0 * (d0 - dot);
    A.lw[1] = b - alpha * ls1 * (d1 - dot);
  }
}

// ---------------------------------------------------------------------------------------------
// host-side launchers
// ---------------------------------------------------------------------------------------------
size_t seq_ws_floats(int D, int NB, int cap) {
  const int NBpad = (NB + 3) & ~3;
  return (size_t)(cap + 1) * D + (size_t)3 * cap * D + (size_t)cap * 3 * D + (size_t)cap * NBpad + (size_t)cap + 16;
}

hipError_t launch_seq_train(const SeqArgs& A, bool spatial, int grid, float alpha, float lambda, hipStream_t st, Timing* tm) {
  const int D = A.dim, XW = spatial ? 2 * D : D, NB = spatial ? A.n_dist + 1 : 0;
  const size_t lds = sizeof(float) * seq_lds_floats(D, XW, (NB + 3) & ~3);
  tm->begin("seq_train", st);
  if (spatial) hipLaunchKernelGGL(seq_train_kernel<true>, dim3(grid), dim3(POI_BLOCK), lds, st, A);
  else hipLaunchKernelGGL(seq_train_kernel<false>, dim3(grid), dim3(POI_BLOCK), lds, st, A);
  tm->end(st);
  hipError_t e = launch_rows_apply(A, spatial, grid, alpha, lambda, st, tm);
  if (e != hipSuccess) return e;
  return launch_dense_apply(A, spatial, grid, grid, alpha, lambda, st, tm);
}

hipError_t launch_rows_apply(const SeqArgs& A, bool spatial, int grid, float alpha, float lambda, hipStream_t st, Timing* tm) {
  const int rows = A.n_item + 1 + (spatial ? A.n_dist + 1 : 0);
  grid = (rows + POI_NWAVE - 1) / POI_NWAVE;
  if (grid > 8192) grid = 8192;
  tm->begin("rows_apply", st);
  if (spatial) hipLaunchKernelGGL(rows_apply_kernel<true>, dim3(grid), dim3(POI_BLOCK), 0, st, A, alpha, lambda);
  else hipLaunchKernelGGL(rows_apply_kernel<false>, dim3(grid), dim3(POI_BLOCK), 0, st, A, alpha, lambda);
  tm->end(st);
  return hipGetLastError();
}

hipError_t launch_dense_apply(const SeqArgs& A, bool spatial, int n_slab, int n_slab_head, float alpha, float lambda, hipStream_t st, Timing* tm) {
  const int D = A.dim, XW = spatial ? 2 * D : D, NB = spatial ? A.n_dist + 1 : 0;
  const DenseLayout dl = dense_layout(D, XW, NB);
  const int dgrid = (dl.total + POI_BLOCK - 1) / POI_BLOCK;
  tm->begin("dense_apply", st);
  if (spatial) hipLaunchKernelGGL(dense_apply_kernel<true>, dim3(dgrid), dim3(POI_BLOCK), 0, st, A, n_slab, n_slab_head, alpha, lambda);
  else hipLaunchKernelGGL(dense_apply_kernel<false>, dim3(dgrid), dim3(POI_BLOCK), 0, st, A, n_slab, n_slab_head, alpha, lambda);
  tm->end(st);
  return hipGetLastError();
}

hipError_t launch_seq_predict(const SeqArgs& A, bool spatial, int grid, hipStream_t st, Timing* tm) {
  const int D = A.dim, XW = spatial ? 2 * D : D, NB = spatial ? A.n_dist + 1 : 0;
  const size_t lds = sizeof(float) * seq_lds_floats(D, XW, (NB + 3) & ~3);
  tm->begin("seq_predict", st);
  if (spatial) hipLaunchKernelGGL(seq_predict_kernel<true>, dim3(grid), dim3(POI_BLOCK), lds, st, A);
  else hipLaunchKernelGGL(seq_predict_kernel<false>, dim3(grid), dim3(POI_BLOCK), lds, st, A);
  tm->end(st);
  return hipGetLastError();
}

}  // namespace poi
