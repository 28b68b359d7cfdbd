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

// =================================================================================================================
// Outer-product path (dims 64 / 128).  The per-sequence kernel above spends its time in six D x D rank-one updates per step - three
// of them float atomics on interval matrices that most workgroups hit at once.  Here the recurrence kernel only RECORDS the
// vectors of every step (EA: g mp | -g mq | da | g vp | -g vq; Hpk: the hidden states) and six entries (matrix id, a, b) per step;
// the entries are sorted by matrix id (stable radix sort, te_scatter.hip) and every matrix gradient is the product
//      d W[id] = sum over the entries of id of a (x) b  =  A_id^T . B_id
// computed on the matrix cores in 512-entry chunks (ca_outer_kernel) whose partial products are added in chunk order
// (ca_outer_reduce_kernel): no float atomics on matrices, reproducible, and the recurrence kernel is left with its GEMVs.
// Matrix ids: 0 .. n_dist = the interval matrices, n_dist + 1 = M.
// =================================================================================================================
#define CA_CK 512       // entries per chunk
typedef float f32x16 __attribute__((ext_vector_type(16)));

// soff = exclusive scan of the step counts (one block; contiguous runs per thread), cnt[0] = 6 * total steps
__global__ __launch_bounds__(1024) void ca_scan_kernel(CaArgs A) {
  __shared__ int wtot[16];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, n = A.n_seq;
  const int per = (n + 1023) / 1024;
  const int b = min(n, tid * per), e = min(n, b + per);
  int s = 0;
  for (int k = b; k < e; ++k) { const int u = A.uidx[k], L = A.off[u + 1] - A.off[u]; s += L > 0 ? L - 1 : 0; }
  int inc = s;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) { const int v = __shfl_up(inc, o, 64); inc += lane >= o ? v : 0; }
  if (lane == 63) wtot[w] = inc;
  __syncthreads();
  if (w == 0) {
    int t = lane < 16 ? wtot[lane] : 0;
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) { const int v = __shfl_up(t, o, 64); t += lane >= o ? v : 0; }
    if (lane < 16) wtot[lane] = t;
  }
  __syncthreads();
  int run = inc - s + (w > 0 ? wtot[w - 1] : 0);
  for (int k = b; k < e; ++k) { const int u = A.uidx[k], L = A.off[u + 1] - A.off[u]; A.soff[k] = run; run += L > 0 ? L - 1 : 0; }
  if (tid == 1023) { A.soff[n] = wtot[15]; A.cnt[0] = 6 * wtot[15]; A.cnt[1] = 3 * wtot[15]; }
}

// PM[r][i] = sum_c M[i][c] lt[r][c] for every table row: the three M x products of a step (x_t, xp_{t+1}, xq_{t+1}) become row
// gathers in the recurrence kernel (a launch's ~231 k steps draw their rows from a 100 k-row table).  64 rows per workgroup pass;
// M^T and the row tile in LDS, thread = (output column i, 4 consecutive rows).
template <int D>
__global__ __launch_bounds__(POI_BLOCK) void ca_pm_kernel(CaArgs A) {
  extern __shared__ __align__(16) float lds_pm[];
  float* Mt = lds_pm;                    // [c][i], pitch D + 1... (read by consecutive i: conflict-free at any pitch)
  float* Xt = Mt + D * D;                // [c][64 rows] of the current pass
  const int tid = threadIdx.x, R = A.n_item + 1;
  for (int e = tid; e < D * D; e += POI_BLOCK) { const int i = e / D, c = e % D; Mt[c * D + i] = A.M[e]; }
  constexpr int TPR = POI_BLOCK / D;     // row groups handled at once (2 at D = 128, 4 at D = 64): each covers 16 / TPR x 4 rows
  const int i = tid % D, rg = tid / D;
  for (int r0 = blockIdx.x * 64; r0 < R; r0 += gridDim.x * 64) {
    __syncthreads();
    for (int e = tid; e < 64 * (D / 4); e += POI_BLOCK) {
      const int r = e / (D / 4), c = (e % (D / 4)) * 4;
      const float4 v = *reinterpret_cast<const float4*>(A.lt + (size_t)min(r0 + r, R - 1) * D + c);
      Xt[(c + 0) * 64 + r] = v.x; Xt[(c + 1) * 64 + r] = v.y; Xt[(c + 2) * 64 + r] = v.z; Xt[(c + 3) * 64 + r] = v.w;
    }
    __syncthreads();
    for (int q = rg; q < 16; q += TPR) {           // 16 groups of 4 rows
      float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll 8
      for (int c = 0; c < D; ++c) {
        const float m = Mt[c * D + i];
        const float4 x = *reinterpret_cast<const float4*>(Xt + c * 64 + 4 * q);
        a0 = fmaf(m, x.x, a0); a1 = fmaf(m, x.y, a1); a2 = fmaf(m, x.z, a2); a3 = fmaf(m, x.w, a3);
      }
      const int r = r0 + 4 * q;
      if (r < R) A.PM[(size_t)r * D + i] = a0;
      if (r + 1 < R) A.PM[(size_t)(r + 1) * D + i] = a1;
      if (r + 2 < R) A.PM[(size_t)(r + 2) * D + i] = a2;
      if (r + 3 < R) A.PM[(size_t)(r + 3) * D + i] = a3;
    }
  }
}

// gemv_rows for rows of at most 128 floats: a HALF-wave per row (gemv_rows strides a row with 64 float4 lanes - at 128 columns half of
// them idle), eight rows in flight per wave, sums closed inside the half-waves.  out[r] = act(W[r, :K] . x + bias[r]).
template <int ACT>
__device__ __forceinline__ void gemv_rows_half(const float* __restrict__ W, int K, const float* x, const float* __restrict__ bias, int nrows, float* out) {
  const int lane = lane_id(), w = wave_id(), hl = lane & 31, hh = lane >> 5;
  for (int r0 = w * 8; r0 < nrows; r0 += POI_NWAVE * 8) {
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int r = r0 + 2 * u + hh;
      if (r < nrows && hl * 4 < K) acc[u] = dot4(*reinterpret_cast<const float4*>(W + (size_t)r * K + hl * 4), *reinterpret_cast<const float4*>(x + hl * 4));
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      float s = acc[u];
#pragma unroll
      for (int o = 16; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
      const int r = r0 + 2 * u + hh;
      if (hl == 0 && r < nrows) {
        float v = s + (bias ? bias[r] : 0.f);
        if (ACT == 1) v = sigmoidf_(v);
        out[r] = v;
      }
    }
  }
}

// Transposed GEMV without barriers: out[j] = sum_i W[i, j] v[i] (W row-major rows x cols, cols = 64 or 128).  A wave owns cols / 4
// consecutive columns as CW = cols / 16 float4 lanes and splits the rows over its 64 / CW lane groups; the group sums meet through
// shuffles inside the wave (gemv_cols: two workgroup barriers and an LDS round trip per call).  The caller publishes `out`.
__device__ __forceinline__ void gemv_cols_wave(const float* __restrict__ W, int rows, int cols, const float* v, float* out) {
  const int lane = lane_id(), w = wave_id();
  const int CW = cols >> 4, RG = 64 / CW;          // float4 columns per wave, row groups
  const int cc = lane % CW, rg = lane / CW;
  const int c = (w * CW + cc) * 4;
  float4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
  int i = rg;
  for (; i + RG < rows; i += 2 * RG) {
    const float4 w0 = *reinterpret_cast<const float4*>(W + (size_t)i * cols + c);
    const float4 w1 = *reinterpret_cast<const float4*>(W + (size_t)(i + RG) * cols + c);
    const float v0 = v[i], v1 = v[i + RG];
    a0.x = fmaf(w0.x, v0, a0.x); a0.y = fmaf(w0.y, v0, a0.y); a0.z = fmaf(w0.z, v0, a0.z); a0.w = fmaf(w0.w, v0, a0.w);
    a1.x = fmaf(w1.x, v1, a1.x); a1.y = fmaf(w1.y, v1, a1.y); a1.z = fmaf(w1.z, v1, a1.z); a1.w = fmaf(w1.w, v1, a1.w);
  }
  if (i < rows) {
    const float4 w0 = *reinterpret_cast<const float4*>(W + (size_t)i * cols + c);
    const float v0 = v[i];
    a0.x = fmaf(w0.x, v0, a0.x); a0.y = fmaf(w0.y, v0, a0.y); a0.z = fmaf(w0.z, v0, a0.z); a0.w = fmaf(w0.w, v0, a0.w);
  }
  float4 s = {a0.x + a1.x, a0.y + a1.y, a0.z + a1.z, a0.w + a1.w};
  for (int o = CW; o < 64; o <<= 1) {
    s.x += __shfl_xor(s.x, o, 64); s.y += __shfl_xor(s.y, o, 64); s.z += __shfl_xor(s.z, o, 64); s.w += __shfl_xor(s.w, o, 64);
  }
  if (rg == 0) *reinterpret_cast<float4*>(out + c) = s;
}

// gemv_cols_wave with TWO vectors against one pass over the matrix: out1 = W^T v1, out2 = W^T v2 (the interval matrices are the
// kernel's L2 stream: every product that shares a matrix shares its read)
__device__ __forceinline__ void gemv_cols_wave2(const float* __restrict__ W, int rows, int cols, const float* v1, const float* v2,
                                                float* out1, float* out2) {
  const int lane = lane_id(), w = wave_id();
  const int CW = cols >> 4, RG = 64 / CW;
  const int cc = lane % CW, rg = lane / CW;
  const int c = (w * CW + cc) * 4;
  float4 a = {0.f, 0.f, 0.f, 0.f}, b = {0.f, 0.f, 0.f, 0.f};
  for (int i = rg; i < rows; i += RG) {
    const float4 w0 = *reinterpret_cast<const float4*>(W + (size_t)i * cols + c);
    const float x = v1[i], y = v2[i];
    a.x = fmaf(w0.x, x, a.x); a.y = fmaf(w0.y, x, a.y); a.z = fmaf(w0.z, x, a.z); a.w = fmaf(w0.w, x, a.w);
    b.x = fmaf(w0.x, y, b.x); b.y = fmaf(w0.y, y, b.y); b.z = fmaf(w0.z, y, b.z); b.w = fmaf(w0.w, y, b.w);
  }
  for (int o = CW; o < 64; o <<= 1) {
    a.x += __shfl_xor(a.x, o, 64); a.y += __shfl_xor(a.y, o, 64); a.z += __shfl_xor(a.z, o, 64); a.w += __shfl_xor(a.w, o, 64);
    b.x += __shfl_xor(b.x, o, 64); b.y += __shfl_xor(b.y, o, 64); b.z += __shfl_xor(b.z, o, 64); b.w += __shfl_xor(b.w, o, 64);
  }
  if (rg == 0) { *reinterpret_cast<float4*>(out1 + c) = a; *reinterpret_cast<float4*>(out2 + c) = b; }
}

__global__ __launch_bounds__(POI_BLOCK) void carnn_train2_kernel(CaArgs A) {
  extern __shared__ __align__(16) float lds_raw[];
  const int D = A.dim, HD = D * D, tid = threadIdx.x, NB = A.n_dist + 1;
  (void)HD;
  CaLds S(lds_raw, D);
  float* ws = A.ws + (size_t)blockIdx.x * A.ws_stride;
  float* wsV = ws + (size_t)(A.cap + 1) * D;         // cap x 4D : mp | mq | vp | vq   (same carve as carnn_train_kernel)
  float* wsY = wsV + (size_t)A.cap * 4 * D;          // cap
  for (int k = blockIdx.x; k < A.n_seq; k += gridDim.x) {
    const int u = A.uidx[k], base = A.off[u], L = A.off[u + 1] - base, ns = L > 0 ? L - 1 : 0;
    const int *p = A.p + base, *q = A.q + base, *dp = A.dp + base, *dq = A.dq + base;
    const int r0 = A.soff[k], hr0 = r0 + k;            // packed step rows, rows of Hpk
    float* H = A.Hpk + (size_t)hr0 * D;
    count_rows<true>(p, q, L, A.n_item, 2 * (A.len_max - L), A.mult_lt, A.nseq_lt);
    count_rows<true>(dp, dq, L, A.n_dist, 2 * (A.len_max - L), A.mult_wd, A.nseq_wd);
    for (int j = tid; j < D; j += POI_BLOCK) { S.hp[j] = 0.f; H[j] = 0.f; }
    float tot = 0.f;       // meaningful in thread 0
    __syncthreads();
    // ------------------------------------------------------------------ forward
    for (int t = 0; t < ns; ++t) {
      if (A.PM) {
        // M x_t, M xp_{t+1}, M xq_{t+1}: rows of the per-launch table PM = lt . M^T
        load_row4(S.mp, A.PM + (size_t)p[t + 1] * D, D);
        load_row4(S.mq, A.PM + (size_t)q[t + 1] * D, D);
        // h_t = sigmoid(M x_t + W[dp_t] h_{t-1})  (:131): the product is the previous step's vp = W[dp_t] h_{t-1} (same matrix,
        // same vector, same routine: the same bits), and W[dp_0] h_0 = 0 - no recurrent GEMV at all
        const float* pm = A.PM + (size_t)p[t] * D;
        for (int j = tid; j < D; j += POI_BLOCK) S.h[j] = sigmoidf_((t > 0 ? S.vp[j] : 0.f) + pm[j]);
      } else {      // small launches: the table would cost more than the three products of the launch's steps
        load_row4(S.x, A.lt + (size_t)p[t] * D, D);
        load_row4(S.xp, A.lt + (size_t)p[t + 1] * D, D);
        load_row4(S.xq, A.lt + (size_t)q[t + 1] * D, D);
        __syncthreads();
        gemv_rows<1>(A.M, D, S.x, A.wd + (size_t)dp[t] * D * D, D, S.hp, nullptr, D, S.h);
        gemv_rows<0>(A.M, D, S.xp, nullptr, 0, nullptr, nullptr, D, S.mp);
        gemv_rows<0>(A.M, D, S.xq, nullptr, 0, nullptr, nullptr, D, S.mq);
      }
      __syncthreads();
      gemv_rows_half<0>(A.wd + (size_t)dp[t + 1] * D * D, D, S.h, nullptr, D, S.vp);
      gemv_rows_half<0>(A.wd + (size_t)dq[t + 1] * D * D, D, S.h, nullptr, D, S.vq);
      __syncthreads();
      float part = 0.f;
      for (int j = tid; j < D; j += POI_BLOCK) {
        part += S.vp[j] * S.mp[j] - S.vq[j] * S.mq[j];                                          // yp - yq (:132-133)
        H[(size_t)(t + 1) * D + j] = S.h[j];
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
      load_row4(S.h, H + (size_t)(t + 1) * D, D);
      __syncthreads();
      const float* Wp = A.wd + (size_t)dp[t + 1] * D * D; const float* Wq = A.wd + (size_t)dq[t + 1] * D * D;
      const float* Wt = A.wd + (size_t)dp[t] * D * D;
      if (t == ns - 1) gemv_cols_wave(Wp, D, D, S.mp, S.t0);        // Wp^T mp (later steps: computed with the previous iteration's W_t pass)
      else for (int j = tid; j < D; j += POI_BLOCK) S.t0[j] = S.xq[j];
      gemv_cols_wave(Wq, D, D, S.mq, S.t1);                        // Wq^T mq
      __syncthreads();
      {
        const size_t r = (size_t)(r0 + t);
        float* ea = A.EA + r * 5 * D;
        for (int j = tid; j < D; j += POI_BLOCK) {
          const float d = S.dh[j] + g * (S.t0[j] - S.t1[j]);
          S.dh[j] = d;
          const float da = d * S.h[j] * (1.0f - S.h[j]);
          S.da[j] = da;
          ea[j] = g * S.mp[j]; ea[D + j] = -g * S.mq[j]; ea[2 * D + j] = da; ea[3 * D + j] = g * S.vp[j]; ea[4 * D + j] = -g * S.vq[j];
        }
        if (tid < 6) {
          // entries: (matrix id, EA vector, b source) - d W[dp_{t+1}] += (g mp) (x) h_t, d W[dq_{t+1}] += (-g mq) (x) h_t,
          // d W[dp_t] += da (x) h_{t-1}, d M += (g vp) (x) xp + (-g vq) (x) xq + da (x) x_t
          const int e = tid;
          const int key = e == 0 ? dp[t + 1] : e == 1 ? dq[t + 1] : e == 2 ? dp[t] : NB;
          const int av = (int)r * 5 + (e == 0 ? 0 : e == 1 ? 1 : e == 2 ? 2 : e == 3 ? 3 : e == 4 ? 4 : 2);
          const int bs = e < 2 ? hr0 + t + 1 : e == 2 ? hr0 + t : e == 3 ? ~p[t + 1] : e == 4 ? ~q[t + 1] : ~p[t];
          const size_t eid = r * 6 + e;
          A.keys0[eid] = key; A.ent_a[eid] = av; A.ent_b[eid] = bs;
        } else if (tid < 9) {
          // row entries: d lt[p_{t+1}] += M^T (g vp), d lt[q_{t+1}] += M^T (-g vq), d lt[p_t] += M^T da - M^T is applied once per ROW to the
          // sum of its vectors (ca_ltgrad_kernel), not per step
          const int j = tid - 6;
          A.k2a[r * 3 + j] = j == 0 ? p[t + 1] : j == 1 ? q[t + 1] : p[t];
        }
      }
      __syncthreads();
      // W_t^T da -> dh_{t-1}, and - the same matrix W[dp_t] is step t-1's Wp - the next iteration's Wp^T mp in the same pass
      if (t > 0) { load_row4(S.xp, wsV + (size_t)(t - 1) * 4 * D, D); __syncthreads(); gemv_cols_wave2(Wt, D, D, S.da, S.xp, S.t1, S.xq); }
      else gemv_cols_wave(Wt, D, D, S.da, S.t1);
      __syncthreads();
      for (int j = tid; j < D; j += POI_BLOCK) S.dh[j] = S.t1[j];
      __syncthreads();
    }
    __syncthreads();
  }
}

// segment [seg_start, seg_end) of every matrix id in the sorted entry list (the arrays are zero on entry: absent ids stay empty)
__global__ __launch_bounds__(256) void ca_bounds_kernel(const int* __restrict__ n_ptr, const int* __restrict__ ks, int* __restrict__ seg_start,
                                                        int* __restrict__ seg_end) {
  const int Ne = *n_ptr;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < Ne; i += gridDim.x * 256) {
    const int k = ks[i];
    if (i == 0 || ks[i - 1] != k) seg_start[k] = i;
    if (i == Ne - 1 || ks[i + 1] != k) seg_end[k] = i + 1;
  }
}

// Sums of the step vectors per POI row, first level: one wave per 64-entry window of the row-sorted entry list walks its entries in
// order (lane = D / 64 columns of the vector) and stores the sum of every run of equal rows at the sorted position of the run's
// first entry inside the window (a run that crosses windows leaves one partial per window).
template <int D>
__global__ __launch_bounds__(POI_BLOCK) void ca_vsum_kernel(CaArgs A, const int* __restrict__ ks, const int* __restrict__ vs) {
  constexpr int C = D / 64;
  const int N2 = A.cnt[1], w = blockIdx.x * POI_NWAVE + wave_id(), lane = lane_id();
  const int b = w * 64;
  if (b >= N2) return;
  const int n = min(64, N2 - b);
  const int myk = lane < n ? ks[b + lane] : -1, mye = lane < n ? vs[b + lane] : 0;
  float acc[C];
#pragma unroll
  for (int c = 0; c < C; ++c) acc[c] = 0.f;
  int cur = __builtin_amdgcn_readfirstlane(myk), first = 0;
  for (int i = 0; i < n; ++i) {
    const int k = __builtin_amdgcn_readlane(myk, i), e = __builtin_amdgcn_readlane(mye, i);
    if (k != cur) {
      float* o = A.vpart + (size_t)(b + first) * D + lane * C;
#pragma unroll
      for (int c = 0; c < C; ++c) { o[c] = acc[c]; acc[c] = 0.f; }
      cur = k; first = i;
    }
    const int r = e / 3, j = e - 3 * r;
    const float* a = A.EA + ((size_t)r * 5 + (j == 0 ? 3 : j == 1 ? 4 : 2)) * D + lane * C;
#pragma unroll
    for (int c = 0; c < C; ++c) acc[c] += a[c];
  }
  float* o = A.vpart + (size_t)(b + first) * D + lane * C;
#pragma unroll
  for (int c = 0; c < C; ++c) o[c] = acc[c];
}

// second level + the product with M: one wave per touched POI row adds the row's partial sums in order (its first entry's position,
// then every window boundary inside its segment) and writes d lt[row] = M^T v into the (zero) gradient table
template <int D>
__global__ __launch_bounds__(POI_BLOCK) void ca_ltgrad_kernel(CaArgs A) {
  constexpr int C = D / 64;
  __shared__ float sv[POI_NWAVE][D];
  const int row = blockIdx.x * POI_NWAVE + wave_id(), lane = lane_id(), w = wave_id();
  if (row > A.n_item) return;
  const int s0 = A.seg2_start[row], s1 = A.seg2_end[row];
  if (s1 == 0) return;
  float v[C];
#pragma unroll
  for (int c = 0; c < C; ++c) v[c] = A.vpart[(size_t)s0 * D + lane * C + c];
  for (int pos = (s0 / 64 + 1) * 64; pos < s1; pos += 64) {
#pragma unroll
    for (int c = 0; c < C; ++c) v[c] += A.vpart[(size_t)pos * D + lane * C + c];
  }
#pragma unroll
  for (int c = 0; c < C; ++c) sv[w][lane * C + c] = v[c];
  __builtin_amdgcn_wave_barrier();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  float g[C];
#pragma unroll
  for (int c = 0; c < C; ++c) g[c] = 0.f;
#pragma unroll 8
  for (int i = 0; i < D; ++i) {
    const float x = sv[w][i];
    const float* m = A.M + (size_t)i * D + lane * C;
#pragma unroll
    for (int c = 0; c < C; ++c) g[c] = fmaf(m[c], x, g[c]);
  }
#pragma unroll
  for (int c = 0; c < C; ++c) A.g_lt[(size_t)row * D + lane * C + c] = g[c];
}

// chunk_first[id] = first 512-entry chunk of matrix id (exclusive scan of the chunk counts), chunk_first[n_id] = total
__global__ __launch_bounds__(1024) void ca_chunks_kernel(CaArgs A) {
  __shared__ int s_cnt[2080];
  const int NK = A.n_dist + 2, tid = threadIdx.x;
  for (int k = tid; k < NK; k += 1024) s_cnt[k] = (A.seg_end[k] - A.seg_start[k] + CA_CK - 1) / CA_CK;
  __syncthreads();
  if (tid == 0) {       // (<= 2050 ids)
    int run = 0;
    for (int k = 0; k < NK; ++k) { const int c = s_cnt[k]; A.chunk_first[k] = run; run += c; }
    A.chunk_first[NK] = run;
  }
}

// partial[c] = sum over the entries of chunk c of a (x) b   (D x D, MFMA 32x32x2: K = the entries, staged 32 at a time through LDS
// as [entry][component] tiles - te_wgrad's transposed-GEMM scheme).  Waves form a 2 x 2 grid of (D/2) x (D/2) quadrants.
template <int D>
__global__ __launch_bounds__(POI_BLOCK) void ca_outer_kernel(CaArgs A, const int* __restrict__ vs) {
  constexpr int LDT = D + 4, Q = D / 64, F4 = 32 * (D / 4) / POI_BLOCK;
  __shared__ __align__(16) float At[32][LDT];
  __shared__ __align__(16) float Bt[32][LDT];
  __shared__ int s_key;
  const int NK = A.n_dist + 2, c = blockIdx.x, tid = threadIdx.x;
  if (c >= A.chunk_first[NK]) return;
  if (tid == 0) {       // the id whose chunk range holds c: last id with chunk_first <= c (empty ids share their successor's value)
    int lo = 0, hi = NK;          // invariant: chunk_first[lo] <= c < chunk_first[hi]
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (A.chunk_first[mid] <= c) lo = mid; else hi = mid; }
    s_key = lo;
  }
  __syncthreads();
  const int key = s_key;
  const int s0 = A.seg_start[key] + (c - A.chunk_first[key]) * CA_CK, s1 = min(A.seg_end[key], s0 + CA_CK);
  const int lane = lane_id(), w = wave_id(), li = lane & 31, h = lane >> 5;
  const int wm = (w >> 1) * (D / 2), wn = (w & 1) * (D / 2);
  f32x16 acc[Q][Q];
#pragma unroll
  for (int i = 0; i < Q; ++i)
#pragma unroll
    for (int j = 0; j < Q; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  for (int e0 = s0; e0 < s1; e0 += 32) {
    float4 ra[F4], rb[F4];
#pragma unroll
    for (int s = 0; s < F4; ++s) {
      const int x = tid + s * POI_BLOCK, r = x / (D / 4), cc = (x % (D / 4)) * 4;
      const bool in = e0 + r < s1;
      const int eid = vs[min(e0 + r, s1 - 1)];
      const int av = A.ent_a[eid], bs = A.ent_b[eid];
      const float* bp = bs >= 0 ? A.Hpk + (size_t)bs * D : A.lt + (size_t)(~bs) * D;
      const float4 a4 = *reinterpret_cast<const float4*>(A.EA + (size_t)av * D + cc);
      const float4 b4 = *reinterpret_cast<const float4*>(bp + cc);
      ra[s] = make_float4(in ? a4.x : 0.f, in ? a4.y : 0.f, in ? a4.z : 0.f, in ? a4.w : 0.f);
      rb[s] = make_float4(in ? b4.x : 0.f, in ? b4.y : 0.f, in ? b4.z : 0.f, in ? b4.w : 0.f);
    }
    __syncthreads();          // the previous stage's MFMAs are done with the tiles
#pragma unroll
    for (int s = 0; s < F4; ++s) {
      const int x = tid + s * POI_BLOCK, r = x / (D / 4), cc = (x % (D / 4)) * 4;
      *reinterpret_cast<float4*>(&At[r][cc]) = ra[s];
      *reinterpret_cast<float4*>(&Bt[r][cc]) = rb[s];
    }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) {
      float av[Q], bv[Q];
#pragma unroll
      for (int i = 0; i < Q; ++i) av[i] = At[2 * kk + h][wm + 32 * i + li];
#pragma unroll
      for (int j = 0; j < Q; ++j) bv[j] = Bt[2 * kk + h][wn + 32 * j + li];
#pragma unroll
      for (int i = 0; i < Q; ++i)
#pragma unroll
        for (int j = 0; j < Q; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[j], acc[i][j], 0, 0, 0);
    }
  }
  float* out = A.partial + (size_t)c * D * D;
#pragma unroll
  for (int i = 0; i < Q; ++i)
#pragma unroll
    for (int j = 0; j < Q; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = wm + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * h;
        out[(size_t)m * D + wn + 32 * j + li] = acc[i][j][r];
      }
}

// gradient of matrix id = its chunks' partial products added in chunk order: eight lanes take every eighth chunk, their sums are
// combined in lane order (fixed tree: reproducible) - into g_wd[id] (zero between launches) or, for M, into slab 0 of the dense apply
template <int D>
__global__ __launch_bounds__(POI_BLOCK) void ca_outer_reduce_kernel(CaArgs A) {
  __shared__ float red[8][32];
  const int NK = A.n_dist + 2, key = blockIdx.x, HD = D * D;
  const int first = A.chunk_first[key], nch = A.chunk_first[key + 1] - first;
  if (nch <= 0) return;
  const int el = blockIdx.y * 32 + (threadIdx.x & 31), ln = threadIdx.x >> 5;
  float s = 0.f;
  for (int q = ln; q < nch; q += 8) s += A.partial[(size_t)(first + q) * HD + el];
  red[ln][threadIdx.x & 31] = s;
  __syncthreads();
  if (ln == 0) {
    float t = red[0][threadIdx.x];
#pragma unroll
    for (int l = 1; l < 8; ++l) t += red[l][threadIdx.x];
    float* dst = key < NK - 1 ? A.g_wd + (size_t)key * HD : A.slab;
    dst[el] += t;
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

template <int D>
static hipError_t carnn_train2_t(const CaArgs& A, int grid, float alpha, float lambda, hipStream_t st, Timing* tm) {
  const int NK = A.n_dist + 2;
  tm->begin("carnn_train", st);
  hipLaunchKernelGGL(ca_scan_kernel, dim3(1), dim3(1024), 0, st, A);
  if (A.PM) hipLaunchKernelGGL(ca_pm_kernel<D>, dim3(1024), dim3(POI_BLOCK), sizeof(float) * (D * D + D * 64), st, A);
  hipLaunchKernelGGL(carnn_train2_kernel, dim3(grid), dim3(POI_BLOCK), sizeof(float) * ca_lds_floats(D), st, A);
  tm->end(st);
  tm->begin("carnn_outer", st);
  int bits = 1; while ((1 << bits) < NK) ++bits;
  const int *ks = nullptr, *vs = nullptr;
  hipError_t e = launch_radix_sort(A.keys0, A.keys1, A.vals0, A.vals1, A.cnt, bits, A.hist, st, &ks, &vs);
  if (e != hipSuccess) return e;
  // (the workspace is carved per launch size: clear the segment tables instead of relying on the previous launch)
  if (hipMemsetAsync(A.seg_start, 0, sizeof(int) * (NK + 4), st) != hipSuccess || hipMemsetAsync(A.seg_end, 0, sizeof(int) * (NK + 4), st) != hipSuccess) return hipGetLastError();
  hipLaunchKernelGGL(ca_bounds_kernel, dim3(1024), dim3(256), 0, st, A.cnt, ks, A.seg_start, A.seg_end);
  hipLaunchKernelGGL(ca_chunks_kernel, dim3(1), dim3(1024), 0, st, A);
  hipLaunchKernelGGL(ca_outer_kernel<D>, dim3(A.n_chunk_cap), dim3(POI_BLOCK), 0, st, A, vs);
  hipLaunchKernelGGL(ca_outer_reduce_kernel<D>, dim3(NK, D * D / 32), dim3(POI_BLOCK), 0, st, A);
  tm->end(st);
  tm->begin("carnn_ltgrad", st);
  {
    int b2 = 1; while ((1 << b2) <= A.n_item) ++b2;
    const int *ks2 = nullptr, *vs2 = nullptr;
    e = launch_radix_sort(A.k2a, A.k2b, A.v2a, A.v2b, A.cnt + 1, b2, A.hist, st, &ks2, &vs2);
    if (e != hipSuccess) return e;
    if (hipMemsetAsync(A.seg2_start, 0, sizeof(int) * (A.n_item + 2), st) != hipSuccess || hipMemsetAsync(A.seg2_end, 0, sizeof(int) * (A.n_item + 2), st) != hipSuccess) return hipGetLastError();
    hipLaunchKernelGGL(ca_bounds_kernel, dim3(1024), dim3(256), 0, st, A.cnt + 1, ks2, A.seg2_start, A.seg2_end);
    hipLaunchKernelGGL(ca_vsum_kernel<D>, dim3((A.n_chunk_cap * 8 + POI_NWAVE - 1) / POI_NWAVE), dim3(POI_BLOCK), 0, st, A, ks2, vs2);
    hipLaunchKernelGGL(ca_ltgrad_kernel<D>, dim3((A.n_item + 1 + POI_NWAVE - 1) / POI_NWAVE), dim3(POI_BLOCK), 0, st, A);
  }
  tm->end(st);
  tm->begin("carnn_apply", st);
  int rows = A.n_item + 1 + A.n_dist + 1, g2 = (rows + POI_NWAVE - 1) / POI_NWAVE;
  if (g2 > 8192) g2 = 8192;
  hipLaunchKernelGGL(carnn_rows_apply_kernel, dim3(g2), dim3(POI_BLOCK), 0, st, A, alpha, lambda);
  hipLaunchKernelGGL(carnn_dense_apply_kernel, dim3((D * D + POI_BLOCK - 1) / POI_BLOCK), dim3(POI_BLOCK), 0, st, A, 1, alpha, lambda);
  tm->end(st);
  return hipGetLastError();
}

hipError_t launch_carnn_train2(const CaArgs& A, int grid, float alpha, float lambda, hipStream_t st, Timing* tm) {
  if (A.dim == 64) return carnn_train2_t<64>(A, grid, alpha, lambda, st, tm);
  if (A.dim == 128) return carnn_train2_t<128>(A, grid, alpha, lambda, st, tm);
  return hipErrorInvalidValue;
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
