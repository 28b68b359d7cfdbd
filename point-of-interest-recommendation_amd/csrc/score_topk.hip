// All-POI scoring (+ fused top-K) on the f32-input matrix cores.
//   scores = trained_users[ids] . trained_items[:-1]^T (+ wd * prob[ids])
//     public/GRU.py:93-96, public/BPR.py:76-79, public/GRU_Spatial.py:117-125
//   top-K  = argpartition(-K) + argsort of the K, descending  (public/Valuate.py:91-100,132-146)
// v_mfma_f32_32x32x2_f32 keeps exact f32 products/accumulation (a k-ordered fmaf chain), which the
// bit-exact-rank requirement needs; bf16 inputs would be 16x faster and flip near-tied ranks.
//
// One wavefront owns a 32-user tile and walks a contiguous range of 32-item tiles.  Operands go
// straight from HBM/L2 to VGPRs (f32 MFMA needs only 16 B/clk/CU of operands - no LDS staging):
// lane (i = lane&31, h = lane>>5) holds row i of the tile, k-columns {8m+4h .. 8m+4h+3}, one float4
// per m; MFMA step s = 4m+c consumes component c, so A and B agree on a permuted k order and every
// byte loaded is used.  The 32x32 result tile stays in registers; the top-K filter compares it with
// the per-user thresholds and appends survivors to per-user LDS candidate lists (f32 score + 16-bit
// item offset), compacted by rank counting over v_readlane whenever a list could overflow.
#include "poi_common.h"
#include "poi_kernels.h"
#include <limits.h>

namespace poi {

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define TK_CAP 80   // candidate slots per user per wave (>= K + 32; compaction when cnt > TK_CAP - 32).
                    // 4 waves x 32 users x 80 x 6 B = 61 KB per workgroup (+ 16 KB of A fragments: two workgroups per CU)

__device__ __forceinline__ bool better(float s, int i, float ps, int pi) {
  return (s > ps) || (s == ps && i < pi);
}

// 64-lane bitonic sort, best (highest score, then lowest index) first.
__device__ __forceinline__ void wave_sort_desc(float& s, int& idx) {
  const int lane = lane_id();
#pragma unroll
  for (int k = 2; k <= 64; k <<= 1) {
#pragma unroll
    for (int j = k >> 1; j > 0; j >>= 1) {
      const float ps = __shfl_xor(s, j, 64);
      const int pi = __shfl_xor(idx, j, 64);
      const bool up = ((lane & k) == 0);
      const bool lower = ((lane & j) == 0);
      const bool mine = better(s, idx, ps, pi);
      const bool keep = (up == lower) ? mine : !mine;
      if (!keep) { s = ps; idx = pi; }
    }
  }
}

struct WaveTopk {   // LDS state of one wavefront
  float cs[32][TK_CAP];
  unsigned short ci[32][TK_CAP];   // item index relative to the first item of this wave's range (< 65536: host-checked)
  int cnt[32];
  float thr[32];
  unsigned gseen[32];   // last global bound seen / published per user (order-mapped)
  unsigned gtmp[32];    // staging of a bound refresh
};

// Orders this wavefront's LDS traffic only (LDS is in-order per wave; the waitcnt makes returned data
// available, the clobber stops compiler reordering).  Deliberately NOT a memory fence: a wavefront-
// scope __builtin_amdgcn_fence also drains vmcnt, i.e. waits for every outstanding global load /
// atomic (the published bounds below), which costs microseconds per compaction.
__device__ __forceinline__ void wave_fence() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_wave_barrier();
}

// Order-preserving map float -> uint (larger float <=> larger uint).
__device__ __forceinline__ unsigned f2ord(float f) {
  const unsigned u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ord2f(unsigned o) { return __uint_as_float((o & 0x80000000u) ? (o & 0x7fffffffu) : ~o); }
__device__ __forceinline__ int popc64(unsigned long long m) { return __builtin_popcountll(m); }

// Keep the best K of user i's (unsorted, <= 128 entries) candidate list and raise its threshold to the
// K-th best score.  Selection by rank counting on the vector ALUs: every candidate (two per lane)
// counts how many candidates beat it (higher score, or equal score and lower index) while the list
// is broadcast from LDS; rank < K survives, rank == K-1 is the new threshold.  No sort, no scalar-unit
// dependency chain (a ballot-based radix select measured ~400 cycles per bit with 8 waves per CU
// sharing the one scalar ALU).
__device__ __forceinline__ void compact_user(WaveTopk& T, int i, int K, unsigned* gbound = nullptr, int dbg = 0) {
  const int lane = lane_id();
  const int n = __builtin_amdgcn_readfirstlane(T.cnt[i]);
  if (n <= K) { if (lane == 0) T.thr[i] = -INFINITY; return; }
  const bool v0 = lane < n, v1 = lane + 64 < n;
  const float s0 = v0 ? T.cs[i][lane] : -INFINITY, s1 = v1 ? T.cs[i][lane + 64] : -INFINITY;
  const int i0 = v0 ? (int)T.ci[i][lane] : 0xFFFF, i1 = v1 ? (int)T.ci[i][lane + 64] : 0xFFFF;
  // one 64-bit key per candidate (order-mapped score, then reversed index): "beats" is a single compare.
  // The list is walked with v_readlane (the candidates already sit two per lane) - no LDS round trip per
  // element, which made this loop ~10 k cycles.
  const unsigned long long k0 = ((unsigned long long)f2ord(s0) << 32) | (unsigned)(0xFFFF - i0);
  const unsigned long long k1 = ((unsigned long long)f2ord(s1) << 32) | (unsigned)(0xFFFF - i1);
  const unsigned h0 = (unsigned)(k0 >> 32), l0 = (unsigned)k0, h1 = (unsigned)(k1 >> 32), l1 = (unsigned)k1;
  int r0 = 0, r1 = 0;
  const int na = n < 64 ? n : 64;
#pragma unroll 4
  for (int j = 0; j < na; ++j) {
    const unsigned long long kj = ((unsigned long long)__builtin_amdgcn_readlane(h0, j) << 32) | (unsigned)__builtin_amdgcn_readlane(l0, j);
    r0 += kj > k0 ? 1 : 0;
    r1 += kj > k1 ? 1 : 0;
  }
  for (int j = 64; j < n; ++j) {
    const unsigned long long kj = ((unsigned long long)__builtin_amdgcn_readlane(h1, j - 64) << 32) | (unsigned)__builtin_amdgcn_readlane(l1, j - 64);
    r0 += kj > k0 ? 1 : 0;
    r1 += kj > k1 ? 1 : 0;
  }
  const bool keep0 = v0 && r0 < K, keep1 = v1 && r1 < K;
  // the K-th best (rank K-1) sits in exactly one slot
  const unsigned long long kb0 = __ballot(v0 && r0 == K - 1), kb1 = __ballot(v1 && r1 == K - 1);
  const float kth = kb0 ? readlane_f(s0, __builtin_ctzll(kb0)) : readlane_f(s1, __builtin_ctzll(kb1 | (1ull << 63)));
  wave_fence();
  if (keep0) { T.cs[i][r0] = s0; T.ci[i][r0] = (unsigned short)i0; }     // ranks 0..K-1 are distinct: sorted, gap-free
  if (keep1) { T.cs[i][r1] = s1; T.ci[i][r1] = (unsigned short)i1; }
  if (lane == 0) {
    T.cnt[i] = K;
    T.thr[i] = kth;
    // Any subset's K-th best score bounds the global K-th best from below: publish (kth - 1 ulp, so
    // that `score > bound` keeps ties) for the waves that scan other item ranges of the same user,
    // but only when it improves on the bound last seen (keeps atomic traffic on hot words low).
    if (gbound && dbg != 4) {
      const unsigned o = f2ord(kth);
      if (o > 1u && o - 1u > T.gseen[i]) { atomicMax(gbound, o - 1u); T.gseen[i] = o - 1u; }
    }
  }
  wave_fence();
}

// Initial thresholds of a wave from the seeded per-user bounds (topk_seed_kernel): the same fold as the periodic refresh in
// tile_epilogue, once, before the first tile.
__device__ __forceinline__ void seed_thresholds(const ScoreArgs& A, WaveTopk& T, float (&thr)[16], int ut, int h) {
  if (!(A.k > 0 && A.gbound && A.seeded)) return;
  const int lane = lane_id();
  if (lane < 32) {
    const unsigned g = __hip_atomic_load(A.gbound + ut * 32 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    T.gtmp[lane] = g; T.gseen[lane] = g;
  }
  wave_fence();
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const unsigned g = T.gtmp[(r & 3) + 8 * (r >> 2) + 4 * h];
    thr[r] = g ? fmaxf(thr[r], ord2f(g)) : thr[r];
  }
}

// Seeded thresholds (poi_ctx_set_topk_seed): the caller hands over, per user, k_seed >= K DISTINCT item ids - in practice the user's
// top-K of the previous evaluation.  Their scores under the CURRENT model bound the current K-th best score from below (any subset's
// K-th best does), so the scoring kernels may start from that bound instead of -inf and append almost nothing but the final top-K:
// the streaming top-K otherwise inserts ~K ln(items / K) records per user and item range.  Exactness does not depend on the seed's
// quality: the bound is a true lower bound - each seed item's score is bounded below by
//   fl(dot) - 4 D u sum|u_i v_i| - 4 u |score|  +  min_b wd sts[user][b]      (u = 2^-24; two float32 evaluations of the same dot
// product differ by at most 2 gamma_D sum|u_i v_i|; the distance term is the one the scoring kernel will add - dense prob element, bin
// matrix element or Haversine bin -, or at least its minimum over the bins) - and the K-th largest of those bounds is
// published.  Out-of-range or repeated ids: no bound for that user (the kernels then start from -inf as before).  One wave per user.
__global__ __launch_bounds__(POI_BLOCK) void topk_seed_kernel(ScoreArgs A, const int* __restrict__ seed, int k_seed) {
  const int u = blockIdx.x * POI_NWAVE + wave_id(), lane = lane_id();
  if (u >= A.n) return;
  const int D = A.dim, K = A.k, N = A.n_item;
  const int id = lane < k_seed ? seed[(size_t)u * k_seed + lane] : -1;
  bool ok = lane >= k_seed || (id >= 0 && id < N);
  for (int j = 0; j < k_seed; ++j) {           // distinct ids
    const int oj = __builtin_amdgcn_readlane(id, j);
    if (lane < k_seed && lane != j && id == oj) ok = false;
  }
  if (__ballot(!ok)) return;
  const float wd = ((A.prob || A.sts) && A.wd) ? A.wd[0] : 0.f;
  float dlb = 0.f, dmax = 0.f;                 // lower bound / magnitude of the distance term
  if (A.sts) {
    const int NB = A.n_dist + 1;
    float mn = 0.f, mx = 0.f;
    for (int b = lane; b < NB; b += 64) { const float v = wd * A.sts[(size_t)u * NB + b]; mn = fminf(mn, v); mx = fmaxf(mx, fabsf(v)); }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { mn = fminf(mn, __shfl_xor(mn, o, 64)); mx = fmaxf(mx, __shfl_xor(mx, o, 64)); }
    dlb = mn; dmax = mx;
  }
  // the k_seed dot products: lanes split the columns, one wave reduction per seed item
  const float* up = A.users + (size_t)u * D;
  float lb = -INFINITY, lad = 0.f;             // lane j ends up with the dot product / absolute sum (then the bound) of seed item j
  for (int j = 0; j < k_seed; ++j) {
    const int it = __builtin_amdgcn_readlane(id, j);
    float d = 0.f, ad = 0.f;
    for (int c = lane * 4; c < D; c += 256) {
      const float4 a = *reinterpret_cast<const float4*>(up + c);
      const float4 b = ld4t(A.items, (size_t)it * D + c, A.items_f16);
      d = fmaf(a.x, b.x, d); d = fmaf(a.y, b.y, d); d = fmaf(a.z, b.z, d); d = fmaf(a.w, b.w, d);
      ad += fabsf(a.x * b.x) + fabsf(a.y * b.y) + fabsf(a.z * b.z) + fabsf(a.w * b.w);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { d += __shfl_xor(d, o, 64); ad += __shfl_xor(ad, o, 64); }
    if (lane == j) { lb = d; lad = ad; }
  }
  // the distance term of lane j's seed item: exact where the kernel can look it up (dense prob row, resident bin matrix - same
  // element the scoring kernel reads -, or the Haversine bin from the coordinates), else its lower bound over the bins
  if (lane < k_seed) {
    float dist = dlb, dm = dmax;
    if (A.prob) { dist = wd * A.prob[(size_t)u * N + id]; dm = fabsf(dist); }
    else if (A.ulptai) {
      // accumulator-order bin matrix (misc.hip::ulptai_kernel): tile (u / 32, id / 32), lane = id % 32 + 32 h, register r with
      // (r & 3) + 8 (r >> 2) + 4 h == u % 32
      const int ur = u & 31, hh = (ur >> 2) & 1, r = (ur & 3) + 4 * (ur >> 3), ntile = A.bins_ntile ? A.bins_ntile : (N + 31) / 32;
      const size_t cell = ((size_t)(u >> 5) * ntile + (id >> 5)) * 64 + (id & 31) + 32 * hh;
      int bin;
      if (A.bin_bytes == 1) bin = reinterpret_cast<const unsigned char*>(A.ulptai)[cell * 16 + r];
      else bin = reinterpret_cast<const unsigned short*>(A.ulptai)[cell * 16 + r];
      dist = wd * A.sts[(size_t)u * (A.n_dist + 1) + bin]; dm = fabsf(dist);
    } else if (A.geo) {
      const int lp = A.last_poi[u];
      int bin;
      {
#pragma clang fp contract(off)
        const double pr = 0.017453292519943295;
        const double a = (A.coords[2 * lp] - A.coords[2 * id]) * pr;
        const double b = (A.coords[2 * lp + 1] - A.coords[2 * id + 1]) * pr;
        const double c = (1.0 - cos_small(a)) / 2 + A.cphi[lp] * A.cphi[id] * (1.0 - cos_small(b)) / 2;
        bin = bin_of_c(c, A.thr, A.n_dist, (float)(12742.0 * 1000.0 / A.dd));
      }
      dist = wd * A.sts[(size_t)u * (A.n_dist + 1) + bin]; dm = fabsf(dist);
    }
    const float eps = 5.9604645e-8f;           // 2^-24
    lb = lb + dist - (4.0f * (float)D * eps * lad + 8.0f * eps * (fabsf(lb) + dm) + 1e-30f);
  }
  // K-th largest of the k_seed bounds (rank counting; ties broken by lane)
  int rank = 0;
  for (int j = 0; j < k_seed; ++j) {
    const float oj = readlane_f(lb, j);
    rank += (oj > lb || (oj == lb && j < lane)) ? 1 : 0;
  }
  const unsigned long long at = __ballot(lane < k_seed && rank == K - 1);
  if (at && lane == 0) {
    const float tau = readlane_f(lb, __builtin_ctzll(at));
    const unsigned o = f2ord(tau);
    if (o > 1u) A.gbound[u] = o - 1u;           // (as compact_user publishes: `score > bound` keeps ties)
  }
}

hipError_t launch_topk_seed(const ScoreArgs& A, const int* seed, int k_seed, hipStream_t st) {
  hipLaunchKernelGGL(topk_seed_kernel, dim3((A.n + POI_NWAVE - 1) / POI_NWAVE), dim3(POI_BLOCK), 0, st, A, seed, k_seed);
  return hipGetLastError();
}

// Per-tile epilogue shared by the scoring kernels.  The 32x32 result tile is in `acc` (C layout:
// item = lane&31, user row = (r&3) + 8*(r>>2) + 4*(lane>>5)).  The per-user thresholds of this lane's
// 16 rows live in registers (`thr`, reloaded only after a compaction); the pass flags are computed
// branch-free and ONE wave-uniform branch skips the whole insertion block when no lane passes - the
// common case once the thresholds have risen - so the fast path has no LDS round trips at all.
__device__ __forceinline__ void tile_epilogue(const ScoreArgs& A, WaveTopk& T, const f32x16& acc, const float (&pv)[16],
                                              float (&thr)[16], float wd, int ut, int j, int jrel, bool jvalid, int K, int tile_no) {
  const int lane = lane_id(), h = lane >> 5, N = A.n_item;
  float sc[16];
  bool any = false;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    sc[r] = __fmaf_rn(wd, pv[r], acc[r]);      // ONE rounding of acc + wd * pv: the expression score_rescore_kernel repeats (score_filter.hip)
    any |= sc[r] > thr[r];
  }
  if (A.scores) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int urow = ut * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
      if (jvalid && urow < A.n) A.scores[(size_t)urow * N + j] = sc[r];
    }
  }
  if (K <= 0 || !jvalid) any = false;
  if (A.dbg == 1) { if (__any(any) && lane == 0) T.cnt[0] = 1; return; }     // tuning: filter check only
  if (K > 0 && A.gbound && (tile_no & 7) == 7 && A.dbg != 2) {
    // fold the bounds published by the other item ranges into the register thresholds (stale values
    // are merely weaker bounds, so relaxed device-scope loads suffice)
    // ONE coalesced load of the tile's 32 bounds, redistributed through LDS (sixteen dependent
    // device-scope loads per lane serialise: ~24 k cycles per refresh)
    if (lane < 32) {
      const unsigned g = __hip_atomic_load(A.gbound + ut * 32 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      T.gtmp[lane] = g;
      if (g > T.gseen[lane]) T.gseen[lane] = g;
    }
    wave_fence();
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const unsigned g = T.gtmp[(r & 3) + 8 * (r >> 2) + 4 * h];
      thr[r] = g ? fmaxf(thr[r], ord2f(g)) : thr[r];
    }
  }
  if (!__any(any)) return;
  // Row base of this half-wave, made opaque inside every block: otherwise the compiler hoists the 32
  // per-row LDS addresses out of the tile loop, spills them (the loop is at the register limit) and
  // reloads them from scratch right here.
  const int urow0 = ut * 32 + 4 * h;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int ic = (r & 3) + 8 * (r >> 2);
    if (any && sc[r] > thr[r] && urow0 + ic < A.n) {
      int hb = 4 * h;
      asm volatile("" : "+v"(hb));
      const int i = ic + hb;
      const int pos = atomicAdd(&T.cnt[i], 1);
      T.cs[i][pos] = sc[r]; T.ci[i][pos] = (unsigned short)jrel;
    }
  }
  wave_fence();
  const int c = lane < 32 ? T.cnt[lane] : 0;
  unsigned long long need = __ballot(c > TK_CAP - 32);
  if (A.dbg == 3 && need) {            // tuning: inserts, but compaction replaced by a reset
    if (lane < 32 && c > TK_CAP - 32) T.cnt[lane] = K;
    wave_fence();
    return;
  }
  if (need) {
    while (need) {
      const int i = __builtin_ctzll(need);
      need &= need - 1;
      compact_user(T, i, K, A.gbound ? A.gbound + ut * 32 + i : nullptr, A.dbg);
    }
    if (A.dbg != 6) {
#pragma unroll
      for (int r = 0; r < 16; ++r) thr[r] = fmaxf(thr[r], T.thr[(r & 3) + 8 * (r >> 2) + 4 * h]);
    }
  }
}

// GEO: the distance term wd * (bin < n_dist ? sts[user][bin] : 0) with bin = cal_dis(last train POI of the user, item) computed
// HERE from the coordinates (float64 Haversine `c` in the reference's operation order + the exact host thresholds, as
// dist_prob_kernel): neither the reference's U x N bin matrix nor a dense prob matrix exists - the only form that scales to
// 1 M users x 10 M POIs.  A lane owns one item of the tile (its coordinates live in registers) and 16 users (LDS).
template <int D8, bool DB, bool GEO>
__global__ __launch_bounds__(POI_BLOCK) void score_kernel(ScoreArgs A) {
  __shared__ WaveTopk tk[POI_NWAVE];
  extern __shared__ __align__(16) double s_geo[];        // GEO: thr[n_dist] | user lat[32] | lon[32] | cos(lat)[32]
  const int lane = lane_id(), w = wave_id();
  const int li = lane & 31, h = lane >> 5;
  const int D = A.dim, N = A.n_item, K = A.k;
  const int ut = blockIdx.x;
  if (A.tile_flag && !A.tile_flag[ut]) return;       // two-stage path: only the tiles whose survivor lists overflowed
  const int split = blockIdx.y * POI_NWAVE + w;
  const int ntile = (N + 31) / 32;
  const int tps = (ntile + A.n_split - 1) / A.n_split;
  const int t_begin = split * tps;
  const int t_end = min(ntile, t_begin + tps);
  WaveTopk& T = tk[w];
  if (K > 0) {
    if (lane < 32) { T.cnt[lane] = 0; T.thr[lane] = -INFINITY; T.gseen[lane] = 0; }
    wave_fence();
  }
  float thr[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) thr[r] = -INFINITY;
  seed_thresholds(A, T, thr, ut, h);

  // A fragment: user row (clamped), k-slices 8m+4h
  float4 af[D8];
  {
    const int urow = min(ut * 32 + li, A.n - 1);
    const float* up = A.users + (size_t)urow * D;
#pragma unroll
    for (int m = 0; m < D8; ++m) {
      const int k0 = 8 * m + 4 * h;
      af[m] = k0 < D ? *reinterpret_cast<const float4*>(up + k0) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
  const float wd = ((A.prob || GEO) && A.wd) ? A.wd[0] : 0.f;
  double* s_ulat = s_geo + A.n_dist; double* s_ulon = s_ulat + 32; double* s_ucp = s_ulon + 32;
  const float gscale = GEO ? (float)(12742.0 * 1000.0 / A.dd) : 0.f;
  const int NBg = A.n_dist + 1;
  if (GEO) {
    for (int i = threadIdx.x; i < A.n_dist; i += POI_BLOCK) s_geo[i] = A.thr[i];
    if (threadIdx.x < 32) {
      const int lp = A.last_poi[min(ut * 32 + (int)threadIdx.x, A.n - 1)];
      s_ulat[threadIdx.x] = A.coords[2 * lp]; s_ulon[threadIdx.x] = A.coords[2 * lp + 1]; s_ucp[threadIdx.x] = A.cphi[lp];
    }
    __syncthreads();
  }

  auto load_b = [&](float4 (&bf)[D8], int tile) {
    const int irow = min(tile * 32 + li, N - 1);
#pragma unroll
    for (int m = 0; m < D8; ++m) {
      const int k0 = 8 * m + 4 * h;
      bf[m] = k0 < D ? ld4t(A.items, (size_t)irow * D + k0, A.items_f16) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };

  float4 b0[D8], b1[DB ? D8 : 1];
  if (t_begin < t_end) load_b(b0, t_begin);

  for (int tile = t_begin; tile < t_end; ++tile) {
    if constexpr (DB) { if (tile + 1 < t_end) load_b(b1, tile + 1); }
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int m = 0; m < D8; ++m) {
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[m].x, b0[m].x, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[m].y, b0[m].y, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[m].z, b0[m].z, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[m].w, b0[m].w, acc, 0, 0, 0);
    }
    {
      const int j = tile * 32 + li;
      const bool jvalid = j < N;
      float pv[16];
      if constexpr (GEO) {
        const int jc = min(j, N - 1);
        const double jlat = A.coords[2 * jc], jlon = A.coords[2 * jc + 1], jcp = A.cphi[jc];
        const double pr = 0.017453292519943295;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int ul = (r & 3) + 8 * (r >> 2) + 4 * h;
          int bin;
          {
#pragma clang fp contract(off)
            const double a = (s_ulat[ul] - jlat) * pr;
            const double b = (s_ulon[ul] - jlon) * pr;
            const double c = (1.0 - cos_small(a)) / 2 + s_ucp[ul] * jcp * (1.0 - cos_small(b)) / 2;
            bin = bin_of_c(c, s_geo, A.n_dist, gscale);
          }
          // (the caller's table has a zero in column n_dist and is readable for whole 32-user tiles: as the BINS path)
          pv[r] = A.sts[(size_t)(ut * 32 + ul) * NBg + bin];
        }
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int urow = ut * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
          pv[r] = (A.prob && jvalid && urow < A.n) ? A.prob[(size_t)urow * N + j] : 0.f;
        }
      }
      tile_epilogue(A, T, acc, pv, thr, wd, ut, j, j - t_begin * 32, jvalid, K, tile - t_begin);
    }
    if constexpr (DB) {
#pragma unroll
      for (int m = 0; m < D8; ++m) b0[m] = b1[m];
    } else {
      if (tile + 1 < t_end) load_b(b0, tile + 1);
    }
  }

  if (K > 0) {   // final per-split lists -> global candidates (split, user, K)
    const int n_pad = gridDim.x * 32;
    for (int i = 0; i < 32; ++i) {
      compact_user(T, i, K);
      const int n = T.cnt[i];
      if (lane < K) {
        const size_t o = ((size_t)split * n_pad + ut * 32 + i) * K + lane;
        A.cand_score[o] = lane < n ? T.cs[i][lane] : -INFINITY;
        A.cand_idx[o] = lane < n ? t_begin * 32 + (int)T.ci[i][lane] : INT_MAX;
      }
    }
  }
}

// Pack the item table into MFMA B-fragment order (see tile_engine.hip):
//   P[(tile * D8 + m) * 64 + lane] = float4{ items[32 tile + j][8m + 4h + c], c = 0..3 }, lane = 32h + j
// so that a wave's B loads are contiguous 1-KiB streams.  Rows >= n_item and k >= D are zero.
__global__ __launch_bounds__(POI_BLOCK) void pack_items_kernel(const float* __restrict__ items, int items_f16, int n_item, int D, int D8,
                                                               float4* __restrict__ out, size_t total) {
  for (size_t e = (size_t)blockIdx.x * POI_BLOCK + threadIdx.x; e < total; e += (size_t)gridDim.x * POI_BLOCK) {
    const int lane = (int)(e & 63);
    const size_t tm = e >> 6;
    const int m = (int)(tm % D8);
    const size_t tile = tm / D8;
    const size_t row = tile * 32 + (lane & 31);
    const int k0 = 8 * m + 4 * (lane >> 5);
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (row < (size_t)n_item && k0 < D) v = ld4t(items, row * D + k0, items_f16);
    out[e] = v;
  }
}

// Large-n variant on the packed item table: every wavefront is independent (own 32-user tile, own
// item range, B fragments double-buffered in registers straight from the packed stream); no LDS
// staging and no workgroup barrier, so a wave that stops to compact a candidate list only delays
// itself while its SIMD partner keeps the matrix core busy.
// BINS: 0 = distance term from the float `prob` matrix (or none); 1 / 2 = from the resident bin matrix
// (uint8 / uint16 bins in accumulator order, misc.hip::ulptai_kernel) and the users' bin probabilities:
// the bins of the NEXT tile are fetched one iteration ahead (one 16/32-byte load per lane), the 16
// probability gathers of a tile are issued before its MFMAs.
template <int D8, int BINS>
__global__ __launch_bounds__(POI_BLOCK, 2) void score_kernel_packed(ScoreArgs A) {
  extern __shared__ __align__(16) float dyn[];
  WaveTopk* tk = reinterpret_cast<WaveTopk*>(dyn);
  // the user tile's A fragments (identical for the four waves of the workgroup: same users, different
  // item ranges) live in LDS, [k-group][lane] float4: 64 registers freed for a full tile of B prefetch
  float4* af = reinterpret_cast<float4*>(tk + POI_NWAVE);
  const int lane = lane_id(), w = wave_id(), li = lane & 31, h = lane >> 5;
  const int D = A.dim, N = A.n_item, K = A.k;
  const int ut = blockIdx.x;
  if (A.tile_flag && !A.tile_flag[ut]) return;       // two-stage path: only the tiles whose survivor lists overflowed
  const int split = blockIdx.y * POI_NWAVE + w;
  const int ntile = (N + 31) / 32;
  const int tps = (ntile + A.n_split - 1) / A.n_split;
  const int t_begin = split * tps;
  const int t_end = min(ntile, t_begin + tps);
  WaveTopk& T = tk[w];
  if (K > 0) {
    if (lane < 32) { T.cnt[lane] = 0; T.thr[lane] = -INFINITY; T.gseen[lane] = 0; }
  }
  {
    const int urow = min(ut * 32 + li, A.n - 1);
    const float* up = A.users + (size_t)urow * D;
    for (int m = w; m < D8; m += POI_NWAVE) {
      const int k0 = 8 * m + 4 * h;
      af[m * 64 + lane] = k0 < D ? *reinterpret_cast<const float4*>(up + k0) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
  __syncthreads();
  float thr[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) thr[r] = -INFINITY;
  seed_thresholds(A, T, thr, ut, h);
  const float wd = ((A.prob || BINS) && A.wd) ? A.wd[0] : 0.f;
  const float4* bp = A.items_packed + lane;
  const float4* ap = af + lane;
  const int NB = A.n_dist + 1;
  constexpr int QN = BINS ? BINS : 1, DH = D8 / 2;
  const uint4* qp = reinterpret_cast<const uint4*>(A.ulptai) + ((size_t)ut * (A.bins_ntile ? A.bins_ntile : ntile) * 64 + lane) * QN;
  uint4 qn[QN];
  const int sbase = (ut * 32 + 4 * h) * NB;      // sts offsets fit 32 bits (checked by the host)
  // The packed item stream runs a full tile ahead: two register sets of two half tiles each; while tile t
  // is multiplied out of set p, tile t+1 lands in set 1-p.
  float4 bA[2][DH], bB[2][DH];
  if (t_begin < t_end) {
#pragma unroll
    for (int m = 0; m < DH; ++m) { bA[0][m] = bp[((size_t)t_begin * D8 + m) * 64]; bB[0][m] = bp[((size_t)t_begin * D8 + DH + m) * 64]; }
    if (BINS) {
#pragma unroll
      for (int q = 0; q < QN; ++q) qn[q] = qp[(size_t)t_begin * 64 * QN + q];
    }
  }
  auto half = [&](f32x16& acc, const float4 (&b)[DH], int m0) {
    float4 a2[2];
    a2[0] = ap[m0 * 64];
#pragma unroll
    for (int m = 0; m < DH; ++m) {
      if (m + 1 < DH) a2[(m + 1) & 1] = ap[(m0 + m + 1) * 64];
      __builtin_amdgcn_sched_barrier(0);
      const float4 a = a2[m & 1];
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b[m].x, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b[m].y, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b[m].z, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b[m].w, acc, 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  auto do_tile = [&](int tile, float4 (&cA)[DH], float4 (&cB)[DH], float4 (&nA)[DH], float4 (&nB)[DH]) {
    const bool more = tile + 1 < t_end;
    if (more) {
#pragma unroll
      for (int m = 0; m < DH; ++m) nA[m] = bp[((size_t)(tile + 1) * D8 + m) * 64];
    }
    const int j = tile * 32 + li;
    const bool jvalid = j < N;
    float pv[16];
    if (BINS) {
      // prob = sts[user][bin]: the bins are <= n_dist by construction and the caller's table has a zero
      // in column n_dist ("too far": fun_acquire_prob's mask, Load_Data_by_length.py:231) and is readable
      // for whole 32-user tiles, so the gather needs no clamp and no select
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        int bin;
        if (BINS == 1) { const unsigned wv = r < 4 ? qn[0].x : r < 8 ? qn[0].y : r < 12 ? qn[0].z : qn[0].w; bin = (wv >> (8 * (r & 3))) & 255u; }
        else { const uint4 qq = qn[(r >> 3) & (QN - 1)]; const int e = r & 7; const unsigned wv = e < 2 ? qq.x : e < 4 ? qq.y : e < 6 ? qq.z : qq.w; bin = (wv >> (16 * (e & 1))) & 65535u; }
        pv[r] = A.sts[sbase + ((r & 3) + 8 * (r >> 2)) * NB + bin];
      }
      if (more) {
#pragma unroll
        for (int q = 0; q < QN; ++q) qn[q] = qp[(size_t)(tile + 1) * 64 * QN + q];
      }
    } else {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int urow = ut * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        pv[r] = (A.prob && jvalid && urow < A.n) ? A.prob[(size_t)urow * N + j] : 0.f;
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    half(acc, cA, 0);
    if (more) {
#pragma unroll
      for (int m = 0; m < DH; ++m) nB[m] = bp[((size_t)(tile + 1) * D8 + DH + m) * 64];
    }
    __builtin_amdgcn_sched_barrier(0);
    half(acc, cB, DH);
    tile_epilogue(A, T, acc, pv, thr, wd, ut, j, j - t_begin * 32, jvalid, K, tile - t_begin);
  };
  for (int tile = t_begin; tile < t_end; tile += 2) {
    do_tile(tile, bA[0], bB[0], bA[1], bB[1]);
    if (tile + 1 < t_end) do_tile(tile + 1, bA[1], bB[1], bA[0], bB[0]);
  }
  if (K > 0) {
    const int n_pad = gridDim.x * 32;
    for (int i = 0; i < 32; ++i) {
      compact_user(T, i, K);
      const int n = T.cnt[i];
      if (lane < K) {
        const size_t o = ((size_t)split * n_pad + ut * 32 + i) * K + lane;
        A.cand_score[o] = lane < n ? T.cs[i][lane] : -INFINITY;
        A.cand_idx[o] = lane < n ? t_begin * 32 + (int)T.ci[i][lane] : INT_MAX;
      }
    }
  }
}

// -------------------------------------------------------------------------------------------------
// GEO on the packed item stream, for the 10 M-POI / dim-256 configuration (and any dim >= 128): the row-per-lane GEO kernel above
// keeps the user tile's A fragments (128 registers at dim 256) AND a tile of B in registers - one wave per SIMD, every tile's item
// rows fetched with nothing to hide the latency, and a float64 Haversine for every (user, item) pair: 19 % of the f32 matrix peak.
// Here
//  * a workgroup is EIGHT waves (two per SIMD) on one 32-user tile: A fragments in LDS (shared), eight item ranges;
//  * the packed B stream runs a full tile ahead in ONE register set: the k-group that has just been multiplied is reloaded with the
//    same k-group of the next tile (a ring: D8 loads in flight, 4 D8 registers instead of 8 D8);
//  * the distance term is computed only where it can matter: score <= dot + ub[user], ub = max_b wd * sts[user][b] (>= 0: column
//    n_dist of the table is zero), so a pair with dot + ub <= the user's K-th best so far can never enter the list and gets the
//    term 0 - exactly the same candidates, ranks and scores as computing every bin.  Once the thresholds have risen (a few hundred
//    items into a range) almost no pair passes: the float64 block runs for a handful of (tile, row) pairs.
// LDS: 8 candidate-list blocks (124 KB) + A fragments (32 KB at dim 256) + thresholds / user coordinates: one workgroup per CU.
// -------------------------------------------------------------------------------------------------
#define SG_NW 8
template <int D8>
__global__ __launch_bounds__(SG_NW * 64) void score_kernel_geo_stream(ScoreArgs A) {
  extern __shared__ __align__(16) float dyn[];
  WaveTopk* tk = reinterpret_cast<WaveTopk*>(dyn);
  float4* af = reinterpret_cast<float4*>(tk + SG_NW);
  double* s_geo = reinterpret_cast<double*>(af + D8 * 64);       // thr[n_dist] | user lat[32] | lon[32] | cos(lat)[32]
  double* s_ulat = s_geo + A.n_dist; double* s_ulon = s_ulat + 32; double* s_ucp = s_ulon + 32;
  float* s_ub = reinterpret_cast<float*>(s_ucp + 32);             // 32 upper bounds of the distance term
  const int lane = lane_id(), w = threadIdx.x >> 6, li = lane & 31, h = lane >> 5;
  const int D = A.dim, N = A.n_item, K = A.k;
  const int ut = blockIdx.x;
  if (A.tile_flag && !A.tile_flag[ut]) return;       // two-stage path: only the tiles whose survivor lists overflowed
  const int split = blockIdx.y * SG_NW + w;
  const int ntile = (N + 31) / 32;
  const int tps = (ntile + A.n_split - 1) / A.n_split;
  const int t_begin = split * tps;
  const int t_end = min(ntile, t_begin + tps);
  WaveTopk& T = tk[w];
  if (lane < 32) { T.cnt[lane] = 0; T.thr[lane] = -INFINITY; T.gseen[lane] = 0; }
  const int NB = A.n_dist + 1;
  const float wd = A.wd[0];
  {
    const int urow = min(ut * 32 + li, A.n - 1);
    const float* up = A.users + (size_t)urow * D;
    for (int m = w; m < D8; m += SG_NW) {
      const int k0 = 8 * m + 4 * h;
      af[m * 64 + lane] = k0 < D ? *reinterpret_cast<const float4*>(up + k0) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    for (int i = threadIdx.x; i < A.n_dist; i += SG_NW * 64) s_geo[i] = A.thr[i];
    if (threadIdx.x < 32) {
      const int lp = A.last_poi[min(ut * 32 + (int)threadIdx.x, A.n - 1)];
      s_ulat[threadIdx.x] = A.coords[2 * lp]; s_ulon[threadIdx.x] = A.coords[2 * lp + 1]; s_ucp[threadIdx.x] = A.cphi[lp];
    }
    // ub[user] = max over the bins of wd * sts[user][bin] (the table is readable for whole 32-user tiles): wave w takes users 4w .. 4w+3
    for (int q = 0; q < 4; ++q) {
      const int u = 4 * w + q;
      float mx = 0.f;
      for (int b = lane; b < NB; b += 64) mx = fmaxf(mx, wd * A.sts[(size_t)(ut * 32 + u) * NB + b]);
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
      // The epilogue's score is ONE rounding of the exact acc + wd * pv (hipcc contracts it to an FMA), while mx is a maximum of ROUNDED
      // products: widened by 2^-21 it is >= every exact product wd * sts[b] (fl(x) >= x (1 - 2^-24)), and rounding is monotone, so
      // fl(acc + ub) >= fl(acc + wd * pv) for every pair - a pair the bound prunes could not have beaten the threshold.
      if (lane == 0) s_ub[u] = mx * 1.00000048f;
    }
  }
  __syncthreads();
  float thr[16], ub[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) { thr[r] = -INFINITY; ub[r] = s_ub[(r & 3) + 8 * (r >> 2) + 4 * h]; }
  seed_thresholds(A, T, thr, ut, h);
  const float gscale = (float)(12742.0 * 1000.0 / A.dd);
  const float4* bp = A.items_packed + lane;
  const float4* ap = af + lane;
  float4 b[D8];
  if (t_begin < t_end) {
#pragma unroll
    for (int m = 0; m < D8; ++m) b[m] = bp[((size_t)t_begin * D8 + m) * 64];
  }
  for (int tile = t_begin; tile < t_end; ++tile) {
    const size_t nb = (size_t)min(tile + 1, t_end - 1) * D8;        // (branch-free: the last tile reloads itself)
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    float4 a2[2];
    a2[0] = ap[0];
#pragma unroll
    for (int m = 0; m < D8; ++m) {
      if (m + 1 < D8) a2[(m + 1) & 1] = ap[(m + 1) * 64];
      __builtin_amdgcn_sched_barrier(0);
      const float4 a = a2[m & 1];
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b[m].x, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b[m].y, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b[m].z, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b[m].w, acc, 0, 0, 0);
      b[m] = bp[(nb + m) * 64];                                    // this k-group of the NEXT tile
      __builtin_amdgcn_sched_barrier(0);
    }
    const int j = tile * 32 + li;
    const bool jvalid = j < N;
    float pv[16];
    bool cand = false;
#pragma unroll
    for (int r = 0; r < 16; ++r) { pv[r] = 0.f; cand |= acc[r] + ub[r] > thr[r]; }
    if (__any(cand && jvalid)) {
      const int jc = min(j, N - 1);
      const double jlat = A.coords[2 * jc], jlon = A.coords[2 * jc + 1], jcp = A.cphi[jc];
      const double pr = 0.017453292519943295;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        if (!__any(jvalid && acc[r] + ub[r] > thr[r])) continue;
        const int ul = (r & 3) + 8 * (r >> 2) + 4 * h;
        int bin;
        {
#pragma clang fp contract(off)
          const double a = (s_ulat[ul] - jlat) * pr;
          const double bb = (s_ulon[ul] - jlon) * pr;
          const double c = (1.0 - cos_small(a)) / 2 + s_ucp[ul] * jcp * (1.0 - cos_small(bb)) / 2;
          bin = bin_of_c(c, s_geo, A.n_dist, gscale);
        }
        pv[r] = A.sts[(size_t)(ut * 32 + ul) * NB + bin];
      }
    }
    tile_epilogue(A, T, acc, pv, thr, wd, ut, j, j - t_begin * 32, jvalid, K, tile - t_begin);
  }
  const int n_pad = gridDim.x * 32;
  for (int i = 0; i < 32; ++i) {
    compact_user(T, i, K);
    const int n = T.cnt[i];
    if (lane < K) {
      const size_t o = ((size_t)split * n_pad + ut * 32 + i) * K + lane;
      A.cand_score[o] = lane < n ? T.cs[i][lane] : -INFINITY;
      A.cand_idx[o] = lane < n ? t_begin * 32 + (int)T.ci[i][lane] : INT_MAX;
    }
  }
}

// Merge n_lists sorted K-lists per user into the final top-K (one wavefront per user).
__global__ __launch_bounds__(POI_BLOCK) void topk_merge_kernel(ScoreArgs A, int n_lists, int n_pad) {
  const int lane = lane_id();
  const int u = blockIdx.x * POI_NWAVE + wave_id();
  if (u >= A.n) return;
  if (A.tile_flag && !A.tile_flag[u >> 5]) return;   // two-stage path: this user's list was written by score_rescore_kernel
  const int K = A.k;
  const int total = n_lists * K;
  float s = -INFINITY; int idx = INT_MAX;
  const int room = 64 - K;
  for (int base = 0; base < total; base += room) {
    // lanes [0,K) keep the running best; lanes [K,64) take the next `room` candidates
    const int c = base + (lane - K);
    if (lane >= K) {
      if (c < total) {
        const int l = c / K, e = c % K;
        const size_t o = ((size_t)l * n_pad + u) * K + e;
        s = A.cand_score[o]; idx = A.cand_idx[o];
      } else { s = -INFINITY; idx = INT_MAX; }
    }
    wave_sort_desc(s, idx);
  }
  if (lane < K) {
    A.idx_out[(size_t)u * K + lane] = idx == INT_MAX ? -1 : idx;
    if (A.score_out) A.score_out[(size_t)u * K + lane] = s;
  }
}

// Top-K of explicit score rows (one wavefront per row): threshold filter + ballot compaction.
__global__ __launch_bounds__(POI_BLOCK) void topk_rows_kernel(const float* __restrict__ scores, int n, int N, int K,
                                                              int* __restrict__ idx_out, float* __restrict__ score_out) {
  __shared__ float cs[POI_NWAVE][128];
  __shared__ int ci[POI_NWAVE][128];
  const int lane = lane_id(), w = wave_id();
  const int row = blockIdx.x * POI_NWAVE + w;
  if (row >= n) return;
  const float* sr = scores + (size_t)row * N;
  int cnt = 0;
  float thr = -INFINITY;
  auto compact = [&]() {
    // best K (<= 64) of up to 128 pending candidates: sort each half best-first; max(a[i], b[63 - i]) then holds the 64
    // best of the 128 (bitonic partition), one more sort orders them
    float s0 = lane < cnt ? cs[w][lane] : -INFINITY; int i0 = lane < cnt ? ci[w][lane] : INT_MAX;
    float s1 = lane + 64 < cnt ? cs[w][lane + 64] : -INFINITY; int i1 = lane + 64 < cnt ? ci[w][lane + 64] : INT_MAX;
    wave_sort_desc(s0, i0);
    wave_sort_desc(s1, i1);
    const float t1 = __shfl(s1, 63 - lane, 64); const int j1 = __shfl(i1, 63 - lane, 64);
    const bool take0 = better(s0, i0, t1, j1) || (s0 == t1 && i0 == j1);
    float s = take0 ? s0 : t1;
    int idx = take0 ? i0 : j1;
    wave_sort_desc(s, idx);
    wave_fence();
    if (lane < K) { cs[w][lane] = s; ci[w][lane] = idx; }
    const int nn = cnt < K ? cnt : K;
    const float kth = __shfl(s, K - 1, 64);
    thr = cnt >= K ? kth : -INFINITY;
    cnt = nn;
    wave_fence();
  };
  for (int c0 = 0; c0 < N; c0 += 64) {
    const int j = c0 + lane;
    const float v = j < N ? sr[j] : -INFINITY;
    const bool pass = j < N && v > thr;
    const unsigned long long mask = __ballot(pass);
    if (mask) {
      const int pos = cnt + __builtin_popcountll(mask & ((1ull << lane) - 1ull));
      if (pass) { cs[w][pos] = v; ci[w][pos] = j; }
      cnt += __builtin_popcountll(mask);
      wave_fence();
      if (cnt > 64) compact();
    }
  }
  compact();
  if (lane < K) {
    idx_out[(size_t)row * K + lane] = lane < cnt ? ci[w][lane] : -1;
    if (score_out) score_out[(size_t)row * K + lane] = lane < cnt ? cs[w][lane] : -INFINITY;
  }
}

template <int D8, bool DB>
static hipError_t launch_score_t(const ScoreArgs& A, hipStream_t st, Timing* tm) {
  dim3 grid((A.n + 31) / 32, A.n_split / POI_NWAVE);
  tm->begin(A.k > 0 ? "score_topk" : "score_all", st);
  if (A.geo) {
    // static candidate lists (62 KB) + the dynamic threshold / user-coordinate block exceed the default 64 KB of LDS per workgroup
    static DeviceOnce once;      // (per device: poi_common.h)
    const hipError_t oe = once.run([]() { return hipFuncSetAttribute(reinterpret_cast<const void*>(&score_kernel<D8, DB, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024); });
    if (oe != hipSuccess) return oe;
    hipLaunchKernelGGL((score_kernel<D8, DB, true>), grid, dim3(POI_BLOCK), sizeof(double) * (A.n_dist + 96), st, A);
  }
  else hipLaunchKernelGGL((score_kernel<D8, DB, false>), grid, dim3(POI_BLOCK), 0, st, A);
  tm->end(st);
  return hipGetLastError();
}

template <int D8>
static hipError_t launch_score_packed_t(const ScoreArgs& A, hipStream_t st, Timing* tm) {
  const size_t total = (size_t)((A.n_item + 31) / 32) * D8 * 64;
  tm->begin("pack_items", st);
  hipLaunchKernelGGL(pack_items_kernel, dim3(2048), dim3(POI_BLOCK), 0, st, A.items, A.items_f16, A.n_item, A.dim, D8, A.items_packed, total);
  tm->end(st);
  dim3 grid((A.n + 31) / 32, A.n_split / POI_NWAVE);
  tm->begin(A.k > 0 ? "score_topk" : "score_all", st);
  if (A.ulptai && A.bin_bytes == 1) hipLaunchKernelGGL((score_kernel_packed<D8, 1>), grid, dim3(POI_BLOCK), sizeof(WaveTopk) * POI_NWAVE + sizeof(float4) * D8 * 64, st, A);
  else if (A.ulptai) hipLaunchKernelGGL((score_kernel_packed<D8, 2>), grid, dim3(POI_BLOCK), sizeof(WaveTopk) * POI_NWAVE + sizeof(float4) * D8 * 64, st, A);
  else hipLaunchKernelGGL((score_kernel_packed<D8, 0>), grid, dim3(POI_BLOCK), sizeof(WaveTopk) * POI_NWAVE + sizeof(float4) * D8 * 64, st, A);
  tm->end(st);
  return hipGetLastError();
}

hipError_t launch_score_packed(const ScoreArgs& A, hipStream_t st, Timing* tm) {
  if (A.dim <= 32) return launch_score_packed_t<4>(A, st, tm);
  if (A.dim <= 64) return launch_score_packed_t<8>(A, st, tm);
  if (A.dim <= 128) return launch_score_packed_t<16>(A, st, tm);
  return hipErrorInvalidValue;
}


size_t score_geo_stream_lds(int dim, int n_dist) {
  const int d8 = dim <= 128 ? 16 : 32;
  return sizeof(WaveTopk) * SG_NW + sizeof(float4) * d8 * 64 + sizeof(double) * (n_dist + 96) + sizeof(float) * 32;
}

template <int D8>
static hipError_t launch_score_geo_stream_t(const ScoreArgs& A, hipStream_t st, Timing* tm) {
  const size_t total = (size_t)((A.n_item + 31) / 32) * D8 * 64;
  const size_t lds = score_geo_stream_lds(A.dim, A.n_dist);
  static DeviceOnce once;
  const hipError_t oe = once.run([]() { return hipFuncSetAttribute(reinterpret_cast<const void*>(&score_kernel_geo_stream<D8>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); });
  if (oe != hipSuccess) return oe;
  tm->begin("pack_items", st);
  hipLaunchKernelGGL(pack_items_kernel, dim3(2048), dim3(POI_BLOCK), 0, st, A.items, A.items_f16, A.n_item, A.dim, D8, A.items_packed, total);
  tm->end(st);
  dim3 grid((A.n + 31) / 32, A.n_split / SG_NW);
  tm->begin("score_topk", st);
  hipLaunchKernelGGL((score_kernel_geo_stream<D8>), grid, dim3(SG_NW * 64), lds, st, A);
  tm->end(st);
  return hipGetLastError();
}

// GEO + top-K on the packed stream (n_split must be a multiple of 8; A.items_packed sized for d8 = 16 (dim <= 128) or 32)
hipError_t launch_score_geo_stream(const ScoreArgs& A, hipStream_t st, Timing* tm) {
  if (A.dim <= 128) return launch_score_geo_stream_t<16>(A, st, tm);
  if (A.dim <= 256) return launch_score_geo_stream_t<32>(A, st, tm);
  return hipErrorInvalidValue;
}

hipError_t launch_score(const ScoreArgs& A, hipStream_t st, Timing* tm) {
  if (A.dim <= 32) return launch_score_t<4, true>(A, st, tm);
  if (A.dim <= 64) return launch_score_t<8, true>(A, st, tm);
  if (A.dim <= 128) return launch_score_t<16, true>(A, st, tm);
  if (A.dim <= 256) return launch_score_t<32, false>(A, st, tm);
  return hipErrorInvalidValue;
}

hipError_t launch_topk_merge(const ScoreArgs& A, int n_lists, int n_pad, hipStream_t st) {
  hipLaunchKernelGGL(topk_merge_kernel, dim3((A.n + POI_NWAVE - 1) / POI_NWAVE), dim3(POI_BLOCK), 0, st, A, n_lists, n_pad);
  return hipGetLastError();
}

hipError_t launch_topk_rows(const float* scores, int n, int n_item, int k, int* idx_out, float* score_out, hipStream_t st) {
  hipLaunchKernelGGL(topk_rows_kernel, dim3((n + POI_NWAVE - 1) / POI_NWAVE), dim3(POI_BLOCK), 0, st, scores, n, n_item, k, idx_out, score_out);
  return hipGetLastError();
}

}  // namespace poi
