// Small kernels around the hot path: AUC preference, sum of squares, last-POI distance-bin
// probabilities, replica-delta helpers, primitive self-test.
#include "poi_common.h"
#include "poi_kernels.h"

namespace poi {

// compute_sub_auc_preference (public/GRU.py:98-110): one wavefront per (user, test position).
__global__ __launch_bounds__(POI_BLOCK) void auc_kernel(const float* __restrict__ users, const float* __restrict__ items, int items_f16,
                                                         int n, int D, const int* __restrict__ tp, const int* __restrict__ tq,
                                                         const int* __restrict__ tm, int len, uint8_t* __restrict__ out) {
  const int e = blockIdx.x * POI_NWAVE + wave_id();
  if (e >= n * len) return;
  const int u = e / len;
  const float* ur = users + (size_t)u * D;
  // float64: the differences and products of float32 values are EXACT in float64 (25 + 24 bits), only the D-term sum rounds (1e-16 of its
  // mass) - the flag is the sign of the reference's own float64 margin (public/GRU.py:98-110) whatever its size; a float32 sum flipped the
  // flags of margins below ~1e-4 (rounds 1 - 4 skipped those in the parity test).  n x len dot products: the cost is nothing.
  double acc = 0.0;
  for (int j = lane_id() * 4; j < D; j += 256) {
    const float4 a = *reinterpret_cast<const float4*>(ur + j);
    const float4 b = ld4t(items, (size_t)tp[e] * D + j, items_f16);
    const float4 c = ld4t(items, (size_t)tq[e] * D + j, items_f16);
    acc = __builtin_fma((double)a.x, (double)b.x - (double)c.x, acc); acc = __builtin_fma((double)a.y, (double)b.y - (double)c.y, acc);
    acc = __builtin_fma((double)a.z, (double)b.z - (double)c.z, acc); acc = __builtin_fma((double)a.w, (double)b.w - (double)c.w, acc);
  }
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
  if (lane_id() == 0) out[e] = (acc * (double)tm[e] > 0.0) ? 1 : 0;
}

__global__ __launch_bounds__(POI_BLOCK) void sumsq_kernel(const float* __restrict__ x, int64_t n, double* out) {
  __shared__ double red[POI_NWAVE];
  double acc = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * POI_BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * POI_BLOCK) {
    const double v = (double)x[i];
    acc += v * v;
  }
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
  if (lane_id() == 0) red[wave_id()] = acc;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(out, (red[0] + red[1]) + (red[2] + red[3]));
}

__global__ __launch_bounds__(POI_BLOCK) void sumsq_f16_kernel(const __half* __restrict__ x, int64_t n, double* out) {
  __shared__ double red[POI_NWAVE];
  double acc = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * POI_BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * POI_BLOCK) {
    const double v = (double)__half2float(x[i]);
    acc += v * v;
  }
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
  if (lane_id() == 0) red[wave_id()] = acc;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(out, (red[0] + red[1]) + (red[2] + red[3]));
}

// cal_dis (public/Load_Data_by_length.py:24-42) in float64, operation order as written there;
// contraction off so that products and sums round separately like the Python expression.
__device__ __forceinline__ int cal_dis_bin(double lat1, double lon1, double lat2, double lon2, double dd, int dist_num) {
#pragma clang fp contract(off)
  const double d = 12742.0;
  const double p = 0.017453292519943295;
  const double a = (lat1 - lat2) * p;
  const double b = (lon1 - lon2) * p;
  const double c = (1.0 - cos(a)) / 2 + cos(lat1 * p) * cos(lat2 * p) * (1.0 - cos(b)) / 2;
  const double dist = d * asin(sqrt(c));
  const double q = dist * 1000 / dd;
  int interval = q >= 2147483647.0 ? 2147483647 : (int)q;
  return interval < dist_num ? interval : dist_num;
}

// fun_compute_distance + fun_acquire_prob (public/Load_Data_by_length.py:183-235) for a user batch.
// With `cphi` (host cos(lat*pi/180) per POI) and `thr` (bin_thresholds of data.py: the smallest c at
// which each bin starts, found with the reference's own libm) the per-pair work is two small-angle
// cosines, the Haversine `c` in the reference's operation order, and a binary search - exactly
// equivalent to cal_dis for every c.  Without them the kernel evaluates cal_dis literally.
__global__ __launch_bounds__(POI_BLOCK) void dist_prob_kernel(const double* __restrict__ coords, const double* __restrict__ cphi,
                                                               const double* __restrict__ thr, const int* __restrict__ last_poi,
                                                               const float* __restrict__ sts, int n, int N, int n_dist,
                                                               double dd, float* __restrict__ prob) {
  extern __shared__ __align__(16) double s_thr[];       // n_dist thresholds, then n_dist+1 probabilities (float)
  const int k = blockIdx.y;
  const int lp = last_poi[k];
  const double lat1 = coords[2 * lp], lon1 = coords[2 * lp + 1];
  const float* s = sts + (size_t)k * (n_dist + 1);
  if (!thr) {
    for (int j = blockIdx.x * POI_BLOCK + threadIdx.x; j < N; j += gridDim.x * POI_BLOCK) {
      const int bin = cal_dis_bin(lat1, lon1, coords[2 * j], coords[2 * j + 1], dd, n_dist);
      prob[(size_t)k * N + j] = bin < n_dist ? s[bin] : 0.f;
    }
    return;
  }
  float* s_p = reinterpret_cast<float*>(s_thr + n_dist);
  for (int i = threadIdx.x; i < n_dist; i += POI_BLOCK) s_thr[i] = thr[i];
  for (int i = threadIdx.x; i <= n_dist; i += POI_BLOCK) s_p[i] = i < n_dist ? s[i] : 0.f;
  __syncthreads();
  const double pr = 0.017453292519943295;
  const double c1 = cphi[lp];
  const float scale = (float)(12742.0 * 1000.0 / dd);
  for (int j = blockIdx.x * POI_BLOCK + threadIdx.x; j < N; j += gridDim.x * POI_BLOCK) {
    int bin;
    {
#pragma clang fp contract(off)
      const double a = (lat1 - coords[2 * j]) * pr;
      const double b = (lon1 - coords[2 * j + 1]) * pr;
      const double c = (1.0 - cos_small(a)) / 2 + c1 * cphi[j] * (1.0 - cos_small(b)) / 2;
      // bin = #{ t : c >= thr[t] } (thr ascending).  asin(x) ~ x for these distances, so
      // int(sqrt(c) * 12742e3/dd) is within one bin of the answer; the exact thresholds then decide.
      int g = (int)(sqrtf((float)c) * scale);
      g = g < 0 ? 0 : (g > n_dist ? n_dist : g);
      while (g > 0 && c < s_thr[g - 1]) --g;
      while (g < n_dist && c >= s_thr[g]) ++g;
      bin = g;
    }
    prob[(size_t)k * N + j] = s_p[bin];
  }
}

// ---------------------------------------------------------------------------------------------
// usrs_last_poi_to_all_intervals ("ulptai", prog_bpr_gru_spatial.py:90; fun_compute_distance,
// public/Load_Data_by_length.py:183-216): the distance bin of every (user's last POI, POI) pair,
// computed ONCE per data set as in the reference and kept resident.  Stored in the scoring kernel's
// accumulator order: for user tile ut (32 users) and item tile it (32 POIs), lane l of the scoring
// wavefront finds its 16 bins contiguous at  out[((ut * ntile + it) * 64 + l) * 16 + r]  (register r
// of the 32x32 MFMA result = user row (r&3) + 8*(r>>2) + 4*(l>>5), POI column l&31), so that one
// 16-byte (uint8 bins) or 32-byte (uint16) load per lane and tile replaces the float prob matrix.
// Pairs outside the matrix (user >= n, POI >= N) hold n_dist (= "too far": probability 0).
// ---------------------------------------------------------------------------------------------
template <typename BT>
__global__ __launch_bounds__(POI_BLOCK) void ulptai_kernel(const double* __restrict__ coords, const double* __restrict__ cphi,
                                                            const double* __restrict__ thr, const int* __restrict__ last_poi,
                                                            int n, int N, int n_dist, double dd, BT* __restrict__ out) {
  extern __shared__ __align__(16) double s_thr[];       // n_dist thresholds, then 32 x {lat, lon, cos(lat)} of the user tile
  double* s_u = s_thr + n_dist;
  const int ut = blockIdx.y, lane = lane_id(), w = wave_id(), li = lane & 31, h = lane >> 5;
  const int ntile = (N + 31) / 32;
  for (int i = threadIdx.x; i < n_dist; i += POI_BLOCK) s_thr[i] = thr[i];
  if (threadIdx.x < 32) {
    const int u = ut * 32 + threadIdx.x;
    const int lp = last_poi[min(u, n - 1)];
    s_u[3 * threadIdx.x] = coords[2 * lp]; s_u[3 * threadIdx.x + 1] = coords[2 * lp + 1]; s_u[3 * threadIdx.x + 2] = cphi[lp];
  }
  __syncthreads();
  const double pr = 0.017453292519943295;
  const float scale = (float)(12742.0 * 1000.0 / dd);
  for (int it = blockIdx.x * POI_NWAVE + w; it < ntile; it += gridDim.x * POI_NWAVE) {
    const int j = it * 32 + li, jc = min(j, N - 1);
    const double lat2 = coords[2 * jc], lon2 = coords[2 * jc + 1], c2 = cphi[jc];
    BT b[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int i = (r & 3) + 8 * (r >> 2) + 4 * h;
      int g;
      {
#pragma clang fp contract(off)
        const double a = (s_u[3 * i] - lat2) * pr;
        const double bb = (s_u[3 * i + 1] - lon2) * pr;
        const double c = (1.0 - cos_small(a)) / 2 + s_u[3 * i + 2] * c2 * (1.0 - cos_small(bb)) / 2;
        g = (int)(sqrtf((float)c) * scale);
        g = g < 0 ? 0 : (g > n_dist ? n_dist : g);
        while (g > 0 && c < s_thr[g - 1]) --g;
        while (g < n_dist && c >= s_thr[g]) ++g;
      }
      if (j >= N || ut * 32 + i >= n) g = n_dist;
      b[r] = (BT)g;
    }
    BT* o = out + (((size_t)ut * ntile + it) * 64 + lane) * 16;
#pragma unroll
    for (int r = 0; r < 16; ++r) o[r] = b[r];
  }
}

// ---------------------------------------------------------------------------------------------
// Per-epoch negative refresh on the device (prog_bpr_gru_spatial.py:221-228):
//   fun_random_neg_masks_tra / _tes (public/Load_Data_by_length.py:127-162): one uniform draw over
//   [0, n_item) per valid position, redrawn while it hits one of the user's own train (and, for the
//   test negative, test) items;
//   fun_compute_dist_neg (:165-180): dq[t] = cal_dis(neg_t, pos_{t-1}), dq[0] = n_dist.
// Counter-based RNG (splitmix64 of seed, position, attempt): reproducible for a seed, independent of
// launch geometry.  The reference draws from Python's Mersenne Twister, whose stream cannot be
// matched; the CONTRACT (support, exclusions, bins) is what the tests pin.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned long long splitmix64(unsigned long long x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}
__device__ __forceinline__ int draw_item(unsigned long long seed, unsigned long long pos, unsigned attempt, int n_item) {
  const unsigned long long r = splitmix64(splitmix64(seed ^ (pos * 0xD1342543DE82EF95ull)) + attempt);
  return (int)__umul64hi(r, (unsigned long long)n_item);      // floor(r / 2^64 * n_item): bias < 2^-47
}

#define NEG_MAX_L 256      // items per user staged in LDS; longer sequences fall back to global reads
__global__ __launch_bounds__(POI_BLOCK) void sample_neg_kernel(const int* __restrict__ off, const int* __restrict__ p, int n_user,
                                                                int n_item, const int* __restrict__ tes_p, const int* __restrict__ tes_mask,
                                                                int len_tes, unsigned long long seed, int* __restrict__ q_out,
                                                                int* __restrict__ tes_q_out) {
  __shared__ int s_items[POI_NWAVE][NEG_MAX_L];
  const int w = wave_id(), lane = lane_id();
  const int u = blockIdx.x * POI_NWAVE + w;
  if (u >= n_user) return;
  const int base = off[u], L = off[u + 1] - base;
  const bool in_lds = L <= NEG_MAX_L;
  if (in_lds) {
    for (int i = lane; i < L; i += 64) s_items[w][i] = p[base + i];
    __builtin_amdgcn_wave_barrier();
  }
  auto owns = [&](int j) {
    bool hit = false;
    if (in_lds) { for (int i = 0; i < L; ++i) hit |= s_items[w][i] == j; }
    else { for (int i = 0; i < L; ++i) hit |= p[base + i] == j; }
    return hit;
  };
  for (int t = lane; t < L; t += 64) {
    unsigned attempt = 0;
    int j = draw_item(seed, (unsigned long long)(base + t), attempt, n_item);
    while (owns(j)) j = draw_item(seed, (unsigned long long)(base + t), ++attempt, n_item);
    q_out[base + t] = j;
  }
  if (tes_q_out) {
    for (int t = lane; t < len_tes; t += 64) {
      const int e = u * len_tes + t;
      if (!tes_mask[e]) { tes_q_out[e] = n_item; continue; }     // padded test position keeps the padding id
      unsigned attempt = 0;
      const unsigned long long pos = 0x8000000000000000ull | (unsigned long long)e;
      auto in_test = [&](int j) { bool h = false; for (int i = 0; i < len_tes; ++i) h |= tes_mask[u * len_tes + i] && tes_p[u * len_tes + i] == j; return h; };
      int j = draw_item(seed, pos, attempt, n_item);
      while (owns(j) || in_test(j)) j = draw_item(seed, pos, ++attempt, n_item);
      tes_q_out[e] = j;
    }
  }
}

// dq[t] = bin(coord[q_t], coord[p_{t-1}]) for t >= 1 inside a sequence, n_dist at t = 0 (one thread per position)
__global__ __launch_bounds__(POI_BLOCK) void neg_dist_kernel(const int* __restrict__ off, const int* __restrict__ p, const int* __restrict__ q,
                                                              int n_user, const double* __restrict__ coords, const double* __restrict__ cphi,
                                                              const double* __restrict__ thr, int n_dist, double dd, int* __restrict__ dq) {
  extern __shared__ __align__(16) double s_thr[];
  for (int i = threadIdx.x; i < n_dist; i += POI_BLOCK) s_thr[i] = thr[i];
  __syncthreads();
  const int total = off[n_user];
  const double pr = 0.017453292519943295;
  const float scale = (float)(12742.0 * 1000.0 / dd);
  for (int e = blockIdx.x * POI_BLOCK + threadIdx.x; e < total; e += gridDim.x * POI_BLOCK) {
    // owning user by binary search over the offsets (first position of a sequence -> n_dist)
    int lo = 0, hi = n_user;
    while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (off[mid] <= e) lo = mid; else hi = mid - 1; }
    int bin = n_dist;
    if (e > off[lo]) {
#pragma clang fp contract(off)
      const int a_ = q[e], b_ = p[e - 1];
      const double a = (coords[2 * a_] - coords[2 * b_]) * pr;
      const double b = (coords[2 * a_ + 1] - coords[2 * b_ + 1]) * pr;
      const double c = (1.0 - cos_small(a)) / 2 + cphi[a_] * cphi[b_] * (1.0 - cos_small(b)) / 2;
      int g = (int)(sqrtf((float)c) * scale);
      g = g < 0 ? 0 : (g > n_dist ? n_dist : g);
      while (g > 0 && c < s_thr[g - 1]) --g;
      while (g < n_dist && c >= s_thr[g]) ++g;
      bin = g;
    }
    dq[e] = bin;
  }
}

// ---------------------------------------------------------------------------------------------
// Rank metrics on the device (public/Valuate.py:23-88,149-172): one thread per user walks its K
// recommendations once and, at every cut-off at_nums[m], adds its hit count, average precision and
// NDCG to the accumulators acc[m*3 + {0,1,2}] (double; block reduction + one atomic per block).
// ---------------------------------------------------------------------------------------------
#define RM_MAX_AT 8
__global__ __launch_bounds__(POI_BLOCK) void rank_metrics_kernel(const int* __restrict__ ranks, int n, int K, const int* __restrict__ tes_p,
                                                                  const int* __restrict__ tes_mask, int len_tes, const int* __restrict__ at_nums,
                                                                  int n_at, double* __restrict__ acc) {
  __shared__ double red[POI_NWAVE];
  const int u = blockIdx.x * POI_BLOCK + threadIdx.x;
  double v[RM_MAX_AT][3];
#pragma unroll
  for (int m = 0; m < RM_MAX_AT; ++m) v[m][0] = v[m][1] = v[m][2] = 0.0;
  if (u < n) {
    int n_test = 0;
    for (int i = 0; i < len_tes; ++i) n_test += tes_mask[u * len_tes + i] != 0;
    int hits = 0, m = 0;
    double ap = 0.0, dcg = 0.0;
    for (int r = 0; r < K && m < n_at; ++r) {
      const int item = ranks[(size_t)u * K + r];
      bool hit = false;
      for (int i = 0; i < len_tes; ++i) hit |= tes_mask[u * len_tes + i] && tes_p[u * len_tes + i] == item;
      if (hit) { ++hits; ap += (double)hits / (double)(r + 1); dcg += 1.0 / log2((double)r + 2.0); }
      while (m < n_at && at_nums[m] == r + 1) {
        double ideal = 0.0;
        const int lim = n_test < r + 1 ? n_test : r + 1;
        for (int i = 0; i < lim; ++i) ideal += 1.0 / log2((double)i + 2.0);
        v[m][0] = hits;
        v[m][1] = n_test > 0 ? ap / (double)n_test : 0.0;
        v[m][2] = (hits > 0 && ideal > 0.0) ? dcg / ideal : 0.0;
        ++m;
      }
    }
  }
  for (int m = 0; m < n_at; ++m)
    for (int j = 0; j < 3; ++j) {
      double x = v[m][j];
      for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o, 64);
      __syncthreads();
      if (lane_id() == 0) red[wave_id()] = x;
      __syncthreads();
      if (threadIdx.x == 0) atomicAdd(&acc[m * 3 + j], (red[0] + red[1]) + (red[2] + red[3]));
    }
}

__global__ void delta_make_kernel(const float* __restrict__ cur, const float* __restrict__ base, float* __restrict__ d, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) d[i] = cur[i] - base[i];
}
__global__ void delta_apply_kernel(float* __restrict__ cur, const float* __restrict__ base, const float* __restrict__ d, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) cur[i] = base[i] + d[i];
}

// Self-test: DPP wave_sum/wave_max vs the ds_bpermute versions, block_sum, float atomics.
__global__ __launch_bounds__(POI_BLOCK) void selftest_kernel(float* buf, int* fail) {
  __shared__ float red[8];
  const int tid = threadIdx.x;
  // pseudo-random but exactly representable values so that sums are order-independent
  const float v = (float)(((tid * 2654435761u) >> 20) & 1023) - 512.0f;
  const float a = wave_sum(v), b = wave_sum_shfl(v);
  if (a != b) atomicAdd(fail, 1);
  float m = v;
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  if (wave_max(v) != m) atomicAdd(fail, 1);
  const float bs = block_sum(v, red);
  float ref = 0.f;
  for (int t = 0; t < POI_BLOCK; ++t) ref += (float)(((t * 2654435761u) >> 20) & 1023) - 512.0f;
  if (bs != ref) atomicAdd(fail, 1);
  atomicAdd(&buf[tid & 7], 1.0f);
  // operand / result layouts of the two f32 MFMA shapes the tile engine relies on (small integers: exact)
  if (wave_id() == 0) {
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    typedef float f32x16 __attribute__((ext_vector_type(16)));
    const int l = lane_id();
    auto Av = [](int i, int k) { return (float)((i * 7 + k * 3) % 11 - 5); };
    auto Bv = [](int k, int j) { return (float)((k * 5 + j * 2) % 13 - 6); };
    {   // 16x16x4: A[i][k] in lane i + 16k, B[k][j] in lane j + 16k, D[4*(lane/16) + r][lane%16] in register r
      const int i = l & 15, g = l >> 4;
      f32x4 c = {0.f, 0.f, 0.f, 0.f};
      c = __builtin_amdgcn_mfma_f32_16x16x4f32(Av(i, g), Bv(g, i), c, 0, 0, 0);
      for (int r = 0; r < 4; ++r) {
        float e = 0.f;
        for (int k = 0; k < 4; ++k) e += Av(4 * g + r, k) * Bv(k, i);
        if (c[r] != e) atomicAdd(fail, 1);
      }
    }
    {   // 32x32x2: A[i][k] in lane i + 32k, B[k][j] in lane j + 32k, D[(r&3) + 8*(r>>2) + 4*(lane/32)][lane%32] in register r
      const int i = l & 31, h = l >> 5;
      f32x16 c;
      for (int r = 0; r < 16; ++r) c[r] = 0.f;
      c = __builtin_amdgcn_mfma_f32_32x32x2f32(Av(i, h), Bv(h, i), c, 0, 0, 0);
      for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
        const float e = Av(row, 0) * Bv(0, i) + Av(row, 1) * Bv(1, i);
        if (c[r] != e) atomicAdd(fail, 1);
      }
    }
  }
}

hipError_t launch_auc(const float* users, const float* items, int items_f16, int n, int dim, const int* tp, const int* tq,
                      const int* tm, int len, uint8_t* out, hipStream_t st) {
  const int e = n * len;
  if (e <= 0) return hipSuccess;
  hipLaunchKernelGGL(auc_kernel, dim3((e + POI_NWAVE - 1) / POI_NWAVE), dim3(POI_BLOCK), 0, st, users, items, items_f16, n, dim, tp, tq, tm, len, out);
  return hipGetLastError();
}
hipError_t launch_sumsq_f16(const void* x, int64_t n, double* out, hipStream_t st) {
  int64_t g = (n + POI_BLOCK * 8 - 1) / (POI_BLOCK * 8);
  if (g > 2048) g = 2048;
  if (g < 1) g = 1;
  hipLaunchKernelGGL(sumsq_f16_kernel, dim3((unsigned)g), dim3(POI_BLOCK), 0, st, (const __half*)x, n, out);
  return hipGetLastError();
}
hipError_t launch_sumsq(const float* x, int64_t n, double* out, hipStream_t st) {
  int64_t g = (n + POI_BLOCK * 8 - 1) / (POI_BLOCK * 8);
  if (g > 2048) g = 2048;
  if (g < 1) g = 1;
  hipLaunchKernelGGL(sumsq_kernel, dim3((unsigned)g), dim3(POI_BLOCK), 0, st, x, n, out);
  return hipGetLastError();
}
hipError_t launch_dist_prob(const double* coords, const double* cphi, const double* thr, const int* last_poi, const float* sts,
                            int n, int n_item, int n_dist, double dd, float* prob, hipStream_t st) {
  int gx = (n_item + POI_BLOCK * 4 - 1) / (POI_BLOCK * 4);
  if (gx > 32) gx = 32;
  if (gx < 1) gx = 1;
  const size_t lds = thr ? sizeof(double) * n_dist + sizeof(float) * (n_dist + 1) : 0;
  hipLaunchKernelGGL(dist_prob_kernel, dim3(gx, n), dim3(POI_BLOCK), lds, st, coords, cphi, thr, last_poi, sts, n, n_item, n_dist, dd, prob);
  return hipGetLastError();
}
hipError_t launch_ulptai(const double* coords, const double* cphi, const double* thr, const int* last_poi, int n, int n_item,
                         int n_dist, double dd, void* out, int bin_bytes, hipStream_t st) {
  const int n_utile = (n + 31) / 32, ntile = (n_item + 31) / 32;
  int gx = (ntile + POI_NWAVE - 1) / POI_NWAVE;
  if (gx > 64) gx = 64;
  const size_t lds = sizeof(double) * (n_dist + 96);
  for (int u0 = 0; u0 < n_utile; u0 += 32768) {          // gridDim.y limit
    const int m = n_utile - u0 < 32768 ? n_utile - u0 : 32768;
    const size_t off = (size_t)u0 * ntile * 1024;
    if (bin_bytes == 1)
      hipLaunchKernelGGL(ulptai_kernel<uint8_t>, dim3(gx, m), dim3(POI_BLOCK), lds, st, coords, cphi, thr, last_poi + (size_t)u0 * 32,
                         n - u0 * 32, n_item, n_dist, dd, (uint8_t*)out + off);
    else
      hipLaunchKernelGGL(ulptai_kernel<uint16_t>, dim3(gx, m), dim3(POI_BLOCK), lds, st, coords, cphi, thr, last_poi + (size_t)u0 * 32,
                         n - u0 * 32, n_item, n_dist, dd, (uint16_t*)out + off);
  }
  return hipGetLastError();
}
hipError_t launch_rank_metrics(const int* ranks, int n, int K, const int* tes_p, const int* tes_mask, int len_tes, const int* at_nums,
                               int n_at, double* acc, hipStream_t st) {
  hipLaunchKernelGGL(rank_metrics_kernel, dim3((n + POI_BLOCK - 1) / POI_BLOCK), dim3(POI_BLOCK), 0, st, ranks, n, K, tes_p, tes_mask, len_tes,
                     at_nums, n_at, acc);
  return hipGetLastError();
}
hipError_t launch_sample_neg(const int* off, const int* p, int n_user, int n_item, const int* tes_p, const int* tes_mask, int len_tes,
                             unsigned long long seed, int* q_out, int* tes_q_out, hipStream_t st) {
  hipLaunchKernelGGL(sample_neg_kernel, dim3((n_user + POI_NWAVE - 1) / POI_NWAVE), dim3(POI_BLOCK), 0, st, off, p, n_user, n_item, tes_p,
                     tes_mask, len_tes, seed, q_out, tes_q_out);
  return hipGetLastError();
}
hipError_t launch_neg_dist(const int* off, const int* p, const int* q, int n_user, const double* coords, const double* cphi,
                           const double* thr, int n_dist, double dd, int* dq, hipStream_t st) {
  hipLaunchKernelGGL(neg_dist_kernel, dim3(2048), dim3(POI_BLOCK), sizeof(double) * n_dist, st, off, p, q, n_user, coords, cphi, thr, n_dist, dd, dq);
  return hipGetLastError();
}
hipError_t launch_delta_make(const float* cur, const float* base, float* delta, int64_t n, hipStream_t st) {
  hipLaunchKernelGGL(delta_make_kernel, dim3(2048), dim3(256), 0, st, cur, base, delta, n);
  return hipGetLastError();
}
hipError_t launch_delta_apply(float* cur, const float* base, const float* dsum, int64_t n, hipStream_t st) {
  hipLaunchKernelGGL(delta_apply_kernel, dim3(2048), dim3(256), 0, st, cur, base, dsum, n);
  return hipGetLastError();
}
hipError_t launch_selftest(float* buf, int* fail, hipStream_t st) {
  hipLaunchKernelGGL(selftest_kernel, dim3(8), dim3(POI_BLOCK), 0, st, buf, fail);
  return hipGetLastError();
}

}  // namespace poi
