// Two-stage fused scoring + top-K (public/Valuate.py:132-146 over public/GRU_Spatial.py:117-125): an f16 FILTER pass on the 16x-faster
// matrix rate followed by an EXACT float32 rescoring of the survivors - the same lists, bit for bit, as the one-stage float32 kernel
// (score_topk.hip), which remains the cold path and the fallback.
//
//   stage 1  score_filter_kernel   users / items rounded to IEEE half, v_mfma_f32_32x32x16_f16 (8 instructions per 32 x 32 tile at dim
//            128 instead of 64 f32 ones), the distance term added exactly as the float32 kernel adds it, and a RIGOROUS bound on
//            |approximate - float32 score| per pair:
//                c1 |u|_2 |v|_2 + 2^-25 (|u|_1 + |v|_1) + rounding slack,   c1 = 2^-10 + 2^-22 + D (2^-22 + 2^-23) + 2^-21
//            (half rounding of both operands: relative 2^-11 each, absolute 2^-25 in the subnormal range; D f32 accumulation steps of
//            the f16 MFMA at <= 2^-22 each and of the f32 MFMA at <= 2^-23; Cauchy-Schwarz for sum |u_i v_i|).  A pair SURVIVES when
//            approximate + bound can exceed the user's threshold - a true lower bound of the final K-th best score, seeded from the
//            previous evaluation's list by topk_seed_kernel (poi_ctx_set_topk_seed).  Survivors (~K + a few per user) go to per-user
//            lists in global memory; a user whose list overflows (bad seed) flags its 32-user tile.
//   stage 2  score_rescore_kernel  per user tile, the survivors of its 32 users in chunks of 32 items: item rows gathered straight into
//            B fragments, the SAME v_mfma_f32_32x32x2_f32 sequence over k as the one-stage kernels (an element of the result tile
//            depends only on its row of A, its column of B and the k order: same bits), the same fused distance term, then per user
//            the K best by (score desc, id asc).  Flagged tiles are left to the one-stage kernel (ScoreArgs.tile_flag), which starts
//            from the same seeded thresholds.
// Exactness never depends on the seed: thresholds are true lower bounds, every pair that could beat them is rescored exactly, and
// overflowing tiles take the exact path.  tests: test_gpu_parity.py (two-stage == one-stage, ids AND scores), tools/fuzz_score.py.
#include "poi_common.h"
#include "poi_kernels.h"
#include <limits.h>
#include <stdlib.h>

namespace poi {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ float sf_ord2f(unsigned o) { return __uint_as_float((o & 0x80000000u) ? (o & 0x7fffffffu) : ~o); }
__device__ __forceinline__ unsigned sf_f2ord(float f) { const unsigned u = __float_as_uint(f); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); }

// Items -> IEEE half in the fragment order of v_mfma_f32_32x32x16_f16 + the two norms of the bound.
//   P[(tile * KG + m) * 64 + lane] = 8 halfs items[32 tile + j][16 m + 8 h + e], e = 0..7, lane = 32 h + j        (KG = D / 16)
//   inorm[item] = { |v|_2 (rounded up), 2^-25 * 1.01 * |v|_1 }; an item with a value outside the half range gets |v|_2 = +inf
// One workgroup per 32-item tile: thread (row j = t / 8, segment s = t % 8) converts D / 8 consecutive columns.
template <int D>
__global__ __launch_bounds__(256) void pack_items_f16_kernel(const float* __restrict__ items, int items_f16, int n_item, uint4* __restrict__ out,
                                                             float2* __restrict__ inorm) {
  constexpr int KG = D / 16, CPT = D / 8;
  const int tile = blockIdx.x, t = threadIdx.x, j = t >> 3, s = t & 7;
  const int gi = tile * 32 + j;
  const bool valid = gi < n_item;
  float x[CPT];
#pragma unroll
  for (int q = 0; q < CPT / 4; ++q) {
    const float4 v = valid ? ld4t(items, (size_t)gi * D + s * CPT + 4 * q, items_f16) : make_float4(0.f, 0.f, 0.f, 0.f);
    x[4 * q] = v.x; x[4 * q + 1] = v.y; x[4 * q + 2] = v.z; x[4 * q + 3] = v.w;
  }
  float n2 = 0.f, n1 = 0.f; bool bad = false;
#pragma unroll
  for (int g = 0; g < CPT / 8; ++g) {
    h8 hv;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float v = x[8 * g + e];
      hv[e] = (_Float16)v;                       // round to nearest even
      n2 = fmaf(v, v, n2); n1 += fabsf(v);
      bad |= !(fabsf(v) <= 65000.f);             // (also NaN)
    }
    const int col0 = s * CPT + 8 * g, m = col0 >> 4, h = (col0 >> 3) & 1;
    out[((size_t)tile * KG + m) * 64 + 32 * h + j] = __builtin_bit_cast(uint4, hv);
  }
#pragma unroll
  for (int o = 1; o < 8; o <<= 1) { n2 += __shfl_xor(n2, o, 64); n1 += __shfl_xor(n1, o, 64); bad |= __shfl_xor((int)bad, o, 64) != 0; }
  if (s == 0 && valid) inorm[gi] = make_float2(bad ? INFINITY : sqrtf(n2) * 1.000002f, n1 * (2.98023224e-8f * 1.01f));
}

#define SF_CAP 4096       // survivor slots per user in global memory (good seeds leave ~K + 1 of them, the self-seeding pre-pass ~16 K; an overflow flags the tile).  8 bytes per slot: a 65536-user call pins 2 GB of grow-only context memory

// stage 1.  BINS: 0 = no distance term, 1 / 2 = resident bin matrix (uint8 / uint16, misc.hip::ulptai_kernel) + the users' bin probabilities,
// 3 = GEO: bins computed on the fly from the coordinates (poi_score_topk_geo: no U x N matrix; config X) - only for the pairs whose
// approximate score + bound + the LARGEST possible distance term of the user can exceed the threshold (as score_kernel_geo_stream)
// UT = 2 (resident bin matrix / no distance term): the workgroup holds TWO user tiles and every wave scores both against each item tile it
// loads - the kernel's time is proportional to the item fragments it pulls out of L2 (loading them twice doubles it: 2.8 -> 5.3 ms at the
// Gowalla shape), so one load now feeds sixteen MFMAs instead of eight.  68 KB of LDS and ~190 registers: two workgroups per CU.
// MAXP (the self-seeding pass of an UNSEEDED call, round 4): no thresholds, no survivors - every lane keeps, per user of its 16 rows, the
// running maximum of (approximate score - the pair's bound) over the items it sees: item range `split`, items = lane (mod 32).  These
// n_split x 32 item BLOCKS are disjoint, so the K-th largest block maximum of a user is a rigorous lower bound of his K-th best exact
// score (sf_select_kernel turns it into the filter's threshold) - and with 512+ blocks the 20 best items almost always sit in 20 different
// blocks: ~1.3 K survivors per user where the float32 one-stage pre-pass over 1/16 of the items left 17 K (and cost 2.6 ms).
template <int D, int BINS, int UT, bool MAXP = false>
__global__ __launch_bounds__(256, ((BINS == 3 || UT == 2) ? 2 : 3)) void score_filter_kernel(ScoreArgs A) {
  constexpr int KG = D / 16, CPT = D / 8, QN = (BINS == 1 || BINS == 2) ? BINS : 1;
  constexpr bool GEO = BINS == 3;
  static_assert(!(GEO && UT != 1), "the GEO filter keeps one user tile per workgroup");
  extern __shared__ __align__(16) float dyn[];
  const int lane = lane_id(), w = wave_id(), li = lane & 31, h = lane >> 5, t = threadIdx.x;
  const int N = A.n_item, NB = A.n_dist + 1;
  // per user tile: [KG][64] half fragments | 32: c1 |u|_2 | 32: threshold - user part of the bound | 32 + 32 | 32 x NB (BINS)
  const int ustride = KG * 64 * 4 + 128 + (BINS ? ((32 * NB + 3) & ~3) : 0);
  auto af_of = [&](int u) { return reinterpret_cast<uint4*>(dyn + u * ustride); };
  auto au_of = [&](int u) { return dyn + u * ustride + KG * 64 * 4; };
  auto c_of = [&](int u) { return au_of(u) + 32; };
  auto n2_of = [&](int u) { return au_of(u) + 64; };
  auto n1_of = [&](int u) { return au_of(u) + 96; };
  auto sts_of = [&](int u) { return au_of(u) + 128; };
  float* s_sts = sts_of(0);
  // GEO: thr[n_dist] | user lat[32] | lon[32] | cos(lat)[32] (doubles, 8-byte aligned behind the float block) | ub[32]
  double* s_geo = reinterpret_cast<double*>(s_sts + ((32 * NB + 1) & ~1));
  double* s_ulat = s_geo + A.n_dist; double* s_ulon = s_ulat + 32; double* s_ucp = s_ulon + 32;
  float* s_ub = reinterpret_cast<float*>(s_ucp + 32);
  const int n_utile = (A.n + 31) / 32;
  int utv[UT];                                             // (a workgroup past the last user tile scores the last one again and drops the result)
#pragma unroll
  for (int u = 0; u < UT; ++u) utv[u] = min((int)blockIdx.x * UT + u, n_utile - 1);
  const bool phantom = UT == 2 && (int)blockIdx.x * UT + 1 >= n_utile;
  const int split = blockIdx.y * POI_NWAVE + w;
  const int ntile = (N + 31) / 32;
  const int tps = (ntile + A.n_split - 1) / A.n_split;
  const int t_begin = split * tps, t_end = min(ntile, t_begin + tps);
  const float wd = (BINS && A.wd) ? A.wd[0] : 0.f;
#pragma unroll
  for (int u = 0; u < UT; ++u) {
    const int ut = utv[u];
    uint4* af = af_of(u);
    const int j = t >> 3, s = t & 7;
    const int urow = min(ut * 32 + j, A.n - 1);
    const float* up = A.users + (size_t)urow * D + s * CPT;
    float n2 = 0.f, n1 = 0.f;
#pragma unroll
    for (int g = 0; g < CPT / 8; ++g) {
      const float4 v0 = *reinterpret_cast<const float4*>(up + 8 * g), v1 = *reinterpret_cast<const float4*>(up + 8 * g + 4);
      const float x[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
      h8 hv;
#pragma unroll
      for (int e = 0; e < 8; ++e) { hv[e] = (_Float16)x[e]; n2 = fmaf(x[e], x[e], n2); n1 += fabsf(x[e]); }
      const int col0 = s * CPT + 8 * g, m = col0 >> 4, hh = (col0 >> 3) & 1;
      af[m * 64 + 32 * hh + j] = __builtin_bit_cast(uint4, hv);
    }
#pragma unroll
    for (int o = 1; o < 8; o <<= 1) { n2 += __shfl_xor(n2, o, 64); n1 += __shfl_xor(n1, o, 64); }
    if (s == 0) { n2_of(u)[j] = n2; n1_of(u)[j] = n1; }
    if (BINS) for (int i = t; i < 32 * NB; i += 256) sts_of(u)[i] = A.sts[(size_t)ut * 32 * NB + i];
    if (GEO) {
      for (int i = t; i < A.n_dist; i += 256) s_geo[i] = A.thr[i];
      if (t < 32) {
        const int lp = A.last_poi[min(ut * 32 + t, A.n - 1)];
        s_ulat[t] = A.coords[2 * lp]; s_ulon[t] = A.coords[2 * lp + 1]; s_ucp[t] = A.cphi[lp];
      }
    }
  }
  __syncthreads();
  if (t < 32 * UT) {
    const int u = t >> 5, tt = t & 31;
    const float* sts = sts_of(u);
    const int urow = utv[u] * 32 + tt;
    float pmax = 0.f;
    if (BINS) for (int b = 0; b < NB; ++b) pmax = fmaxf(pmax, fabsf(sts[tt * NB + b]));
    const unsigned g = urow < A.n ? A.gbound[urow] : 0u;
    const float thr = g ? sf_ord2f(g) : -INFINITY;
    const float nu2 = sqrtf(n2_of(u)[tt]) * 1.000002f;
    constexpr float c1 = (9.765625e-4f * 1.0005f + (float)D * (2.38418579e-7f * 1.001f + 1.19209290e-7f) + 4.76837158e-7f) * 1.00001f;
    const float bu = n1_of(u)[tt] * (2.98023224e-8f * 1.01f) + 4.76837158e-7f * (fabsf(wd) * pmax + (g ? fabsf(thr) : 0.f)) + 1e-30f;
    au_of(u)[tt] = c1 * nu2;
    c_of(u)[tt] = urow < A.n ? thr - bu : INFINITY;        // rows past n: nothing survives
    // an unseeded user (no bound: malformed / missing seed row) - or one whose row may not survive the rounding to half: an element
    // beyond the half range becomes +-inf, an all-(-inf) product sum would DROP a pair whose float32 score is finite (|x| > 65000 implies
    // |u|_2^2 > 4e9; the converse costs only speed) - hands its tile to the one-stage kernel
    n2_of(u)[tt] = (urow < A.n && (!g || !(n2_of(u)[tt] < 4.0e9f))) ? 1.f : 0.f;
    if (GEO) {                                             // ub = max_b wd sts[b] (>= 0: column n_dist is zero), widened: >= every exact product
      float mx = 0.f;
      for (int b = 0; b < NB; ++b) mx = fmaxf(mx, wd * sts[tt * NB + b]);
      s_ub[tt] = mx * 1.00000048f;
    }
  }
  __syncthreads();
  bool dead[UT];
  {
    // a tile with an unseeded user would keep every pair of that user: it goes to the one-stage kernel as a whole, at once
    bool all_dead = true;
#pragma unroll
    for (int u = 0; u < UT; ++u) {
      bool unseeded = false;
      if (!MAXP) for (int i = 0; i < 32; ++i) unseeded |= n2_of(u)[i] != 0.f;
      if (unseeded && t == 0 && !(u == 1 && phantom)) A.tile_flag[utv[u]] = 1;
      dead[u] = unseeded || (u == 1 && phantom);
      all_dead &= dead[u];
    }
    if (all_dead) return;
  }
  // (the norm part of the bound uses the LARGEST c1 |u|_2 of the tile's 32 users: one fma per lane and tile instead of one per pair and
  // sixteen registers less - the hidden states of a model have similar norms, so the bound loosens by a few per cent at most)
  float cc[UT][16], au[UT];
#pragma unroll
  for (int u = 0; u < UT; ++u) {
    au[u] = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) cc[u][r] = c_of(u)[(r & 3) + 8 * (r >> 2) + 4 * h];
    for (int i = 0; i < 32; ++i) au[u] = fmaxf(au[u], au_of(u)[i]);
  }
  float ubt = 0.f;                 // GEO: the largest distance term any user of the tile can get (one register instead of sixteen)
  if (GEO) for (int i = 0; i < 32; ++i) ubt = fmaxf(ubt, s_ub[i]);
  const float gscale = GEO ? (float)(12742.0 * 1000.0 / A.dd) : 0.f;
  const uint4* bp = A.items_packed16 + lane;
  const uint4* qp[UT];
#pragma unroll
  for (int u = 0; u < UT; ++u) qp[u] = reinterpret_cast<const uint4*>(A.ulptai) + ((size_t)utv[u] * ntile * 64 + lane) * QN;
  const int sbase = 4 * h * NB;
  uint4 b[KG], qn[UT][QN];
  float2 nm = make_float2(0.f, 0.f);
  if (t_begin < t_end) {
#pragma unroll
    for (int m = 0; m < KG; ++m) b[m] = bp[((size_t)t_begin * KG + m) * 64];
    if (BINS == 1 || BINS == 2) {
#pragma unroll
      for (int u = 0; u < UT; ++u)
#pragma unroll
        for (int q = 0; q < QN; ++q) qn[u][q] = qp[u][(size_t)t_begin * 64 * QN + q];
    }
    nm = A.inorm[min(t_begin * 32 + li, N - 1)];
  }
  int flagv[UT];
  float mx[UT][MAXP ? 16 : 1];
#pragma unroll
  for (int u = 0; u < UT; ++u) {
    flagv[u] = 0;
#pragma unroll
    for (int r = 0; r < (MAXP ? 16 : 1); ++r) mx[u][r] = -INFINITY;
  }
  // (MAXP: every max_stride-th item tile - the K-th best of a SUBSET of the items is a lower bound too; see launch_maxpass_t)
  const int tstep = MAXP ? max(A.max_stride, 1) : 1;
  for (int tile = t_begin; tile < t_end; tile += tstep) {
    // a tile whose survivor lists overflowed (useless seeds: thresholds far below the final ones) is rescored by the one-stage kernel
    // anyway: the flag is polled every 16 item tiles, one poll period ahead (no wait on the load), and the wave stops
    if (!MAXP && ((tile - t_begin) & 15) == 0) {
      bool all_dead = true;
#pragma unroll
      for (int u = 0; u < UT; ++u) {
        dead[u] |= flagv[u] != 0;
        all_dead &= dead[u];
        flagv[u] = __hip_atomic_load(A.tile_flag + utv[u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      if (all_dead) break;
    }
    const int nt = min(tile + tstep, t_end - 1);           // (branch-free: the last tile reloads itself)
    const int j = tile * 32 + li;
    const bool jvalid = j < N;
#pragma unroll
    for (int u = 0; u < UT; ++u) {
      const uint4* af = af_of(u);
      const float* sts = sts_of(u);
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
      for (int m = 0; m < KG; ++m) {
        const h8 a = __builtin_bit_cast(h8, af[m * 64 + lane]);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, __builtin_bit_cast(h8, b[m]), acc, 0, 0, 0);
        if (u == UT - 1) b[m] = bp[((size_t)nt * KG + m) * 64];               // this k-group of the NEXT tile
      }
      const float tb = __fmaf_rn(au[u], nm.x, nm.y);
      unsigned pass = 0;
      if constexpr (GEO) {
        // coarse test with the largest distance term the user can get; the float64 Haversine bin only for the (tile, row) pairs that pass
        unsigned coarse = 0;
#pragma unroll
        for (int r = 0; r < 16; ++r) coarse |= !((acc[r] + ubt) + tb <= cc[u][r]) ? (1u << r) : 0u;
        if (!jvalid) coarse = 0;
        if (__any(coarse != 0)) {
          const int jc = min(j, N - 1);
          const double jlat = A.coords[2 * jc], jlon = A.coords[2 * jc + 1], jcp = A.cphi[jc];
          const double pr = 0.017453292519943295;
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            if (!__any((coarse >> r) & 1u)) continue;
            const int ul = (r & 3) + 8 * (r >> 2) + 4 * h;
            int bin;
            {
#pragma clang fp contract(off)
              const double a = (s_ulat[ul] - jlat) * pr;
              const double bb = (s_ulon[ul] - jlon) * pr;
              const double c = (1.0 - cos_small(a)) / 2 + s_ucp[ul] * jcp * (1.0 - cos_small(bb)) / 2;
              bin = bin_of_c(c, s_geo, A.n_dist, gscale);
            }
            const float pv = sts[ul * NB + bin];
            const float up = __fmaf_rn(wd, pv, acc[r]) + tb;
            pass |= (((coarse >> r) & 1u) && !(up <= cc[u][r])) ? (1u << r) : 0u;
          }
        }
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          if (r == 8) __builtin_amdgcn_sched_barrier(0);     // two batches of eight gathers: sixteen in flight cost the occupancy
          float pv = 0.f;
          if (BINS) {
            int bin;
            if (BINS == 1) { const unsigned wv = r < 4 ? qn[u][0].x : r < 8 ? qn[u][0].y : r < 12 ? qn[u][0].z : qn[u][0].w; bin = (wv >> (8 * (r & 3))) & 255u; }
            else { const uint4 qq = qn[u][(r >> 3) & (QN - 1)]; const int e = r & 7; const unsigned wv = e < 2 ? qq.x : e < 4 ? qq.y : e < 6 ? qq.z : qq.w; bin = (wv >> (16 * (e & 1))) & 65535u; }
            pv = sts[sbase + ((r & 3) + 8 * (r >> 2)) * NB + bin];
          }
          if constexpr (MAXP) { mx[u][r] = fmaxf(mx[u][r], jvalid ? __fmaf_rn(wd, pv, acc[r]) - tb : -INFINITY); continue; }      // (a NaN is ignored: lower)
          const float up = __fmaf_rn(wd, pv, acc[r]) + tb;
          pass |= !(up <= cc[u][r]) ? (1u << r) : 0u;        // (a NaN survives)
        }
      }
      if (BINS == 1 || BINS == 2) {
#pragma unroll
        for (int q = 0; q < QN; ++q) qn[u][q] = qp[u][(size_t)nt * 64 * QN + q];
      }
      if (!jvalid || dead[u]) pass = 0;
      if (__any(pass != 0)) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          if (pass & (1u << r)) {
            const int urow = utv[u] * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            const int pos = atomicAdd(A.surv_cnt + urow, 1);
            if (pos < SF_CAP) A.surv_idx[(size_t)urow * SF_CAP + pos] = j;
            else A.tile_flag[utv[u]] = 1;
          }
        }
      }
    }
    nm = A.inorm[min(nt * 32 + li, N - 1)];
  }
  if constexpr (MAXP) {      // block maxima -> surv_sc as scratch: [user][n_split x 32], one 128-byte line per (user, item range)
    const int bw = A.n_split * 32;
#pragma unroll
    for (int u = 0; u < UT; ++u) {
      if (u == 1 && phantom) continue;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int urow = utv[u] * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        A.surv_sc[(size_t)urow * bw + split * 32 + li] = mx[u][r];
      }
    }
  }
}

// K-th largest of a user's block maxima (score_filter_kernel<MAXP>) -> the filter's threshold gbound[u].  One wave per user, the BWL * 64
// values in registers, K rounds of (wave maximum, its first owner drops it).  What is subtracted on top of the per-pair bound the maxima
// already carry: the user part of the filter's bound (2^-25 |u|_1) and the float32 evaluation of the comparison itself, both doubled.
template <int BWL>
__global__ __launch_bounds__(256) void sf_select_kernel(ScoreArgs A, int k) {
  const int u = blockIdx.x * POI_NWAVE + wave_id(), lane = lane_id();
  if (u >= A.n) return;
  const float* bm = A.surv_sc + (size_t)u * (BWL * 64);
  float v[BWL];
#pragma unroll
  for (int i = 0; i < BWL; ++i) v[i] = bm[lane + 64 * i];
  float kth = -INFINITY;
  for (int it = 0; it < k; ++it) {
    float m = v[0];
#pragma unroll
    for (int i = 1; i < BWL; ++i) m = fmaxf(m, v[i]);
    const float wm = wave_max(m);
    kth = wm;
    if (!(wm > -INFINITY)) break;                              // fewer than K finite blocks: no bound
    const unsigned long long own = __ballot(m == wm);
    if (lane == __ffsll((long long)own) - 1) {                 // the first owner drops ONE instance
      bool done = false;
#pragma unroll
      for (int i = 0; i < BWL; ++i) { const bool hit = !done && v[i] == wm; v[i] = hit ? -INFINITY : v[i]; done |= hit; }
    }
  }
  if (!(kth > -INFINITY)) return;
  float n1 = 0.f, pmax = 0.f;
  for (int i = lane; i < A.dim; i += 64) n1 += fabsf(A.users[(size_t)u * A.dim + i]);
  n1 = wave_sum(n1);
  const bool bins = A.ulptai != nullptr || A.geo;
  if (bins && A.sts) { for (int b = lane; b <= A.n_dist; b += 64) pmax = fmaxf(pmax, fabsf(A.sts[(size_t)u * (A.n_dist + 1) + b])); pmax = wave_max(pmax); }
  const float wd = (bins && A.wd) ? A.wd[0] : 0.f;
  const float slack = 2.f * (n1 * (2.98023224e-8f * 1.01f) + 4.76837158e-7f * (fabsf(wd) * pmax + fabsf(kth))) + 1e-30f;
  const float thr = kth - slack;
  if (lane == 0 && thr > -INFINITY) {
    const unsigned o = sf_f2ord(thr);
    if (o > 1u && o - 1u > A.gbound[u]) A.gbound[u] = o - 1u;
  }
}

// -------------------------------------------------------------------------------------------------------------------------------------
// ITEM-STATIONARY filter pass for the GEO path with a huge item table and few users (config X's evaluation: 8192 users x 10 M POIs).
// score_filter_kernel gives every user tile its own pass over the item table: 256 user tiles x 5.1 GB of half fragments = 1.3 TB per
// call, HBM / MALL-bound (165 ms).  Here the loops are swapped: a workgroup keeps the B fragments of FOUR item tiles in registers (one
// per wave) and walks the USER tiles, whose half fragments (4 MB in all: L2-resident) are staged through LDS once per workgroup and
// shared by its four waves - the item table is read once, the users stream from L2 at a quarter of the volume.
//   sf_users_prep_kernel   per user tile, once per call: the half fragments, and per user { c1 |u|_2, threshold - bound, largest distance
//                          term }, the last-POI coordinates, and the tile flag of unseeded users - the prologue of score_filter_kernel,
//                          the same expressions
//   score_filter_items_kernel   the same tests as score_filter_kernel<D, 3> (coarse with the tile's largest distance term, float64
//                          Haversine bin only for the (tile, row) pairs that pass), survivors into the same per-user lists
// -------------------------------------------------------------------------------------------------------------------------------------
template <int D>
__global__ __launch_bounds__(256) void sf_users_prep_kernel(ScoreArgs A, uint4* __restrict__ upk, float4* __restrict__ ub, double* __restrict__ ugeo) {
  constexpr int KG = D / 16, CPT = D / 8;
  __shared__ float s_n2[32], s_n1[32];
  const int t = threadIdx.x, ut = blockIdx.x, NB = A.n_dist + 1;
  const float wd = A.wd ? A.wd[0] : 0.f;
  {
    const int j = t >> 3, s = t & 7;
    const int urow = min(ut * 32 + j, A.n - 1);
    const float* up = A.users + (size_t)urow * D + s * CPT;
    float n2 = 0.f, n1 = 0.f;
#pragma unroll
    for (int g = 0; g < CPT / 8; ++g) {
      const float4 v0 = *reinterpret_cast<const float4*>(up + 8 * g), v1 = *reinterpret_cast<const float4*>(up + 8 * g + 4);
      const float x[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
      h8 hv;
#pragma unroll
      for (int e = 0; e < 8; ++e) { hv[e] = (_Float16)x[e]; n2 = fmaf(x[e], x[e], n2); n1 += fabsf(x[e]); }
      const int col0 = s * CPT + 8 * g, m = col0 >> 4, hh = (col0 >> 3) & 1;
      upk[((size_t)ut * KG + m) * 64 + 32 * hh + j] = __builtin_bit_cast(uint4, hv);
    }
#pragma unroll
    for (int o = 1; o < 8; o <<= 1) { n2 += __shfl_xor(n2, o, 64); n1 += __shfl_xor(n1, o, 64); }
    if (s == 0) { s_n2[j] = n2; s_n1[j] = n1; }
  }
  __syncthreads();
  bool unseeded = false;
  if (t < 32) {
    const int urow = ut * 32 + t, ur = min(urow, A.n - 1);
    const float* srow = A.sts + (size_t)(ut * 32 + t) * NB;        // (sts has n_pad rows)
    float pmax = 0.f, mx = 0.f;
    for (int b = 0; b < NB; ++b) { const float p = srow[b]; pmax = fmaxf(pmax, fabsf(p)); mx = fmaxf(mx, wd * p); }
    const unsigned g = urow < A.n ? A.gbound[urow] : 0u;
    const float thr = g ? sf_ord2f(g) : -INFINITY;
    const float nu2 = sqrtf(s_n2[t]) * 1.000002f;
    constexpr float c1 = (9.765625e-4f * 1.0005f + (float)D * (2.38418579e-7f * 1.001f + 1.19209290e-7f) + 4.76837158e-7f) * 1.00001f;
    const float bu = s_n1[t] * (2.98023224e-8f * 1.01f) + 4.76837158e-7f * (fabsf(wd) * pmax + (g ? fabsf(thr) : 0.f)) + 1e-30f;
    // .w: the tile's maxima (user 0: largest c1 |u|_2, user 1: largest distance term) - what the filter's coarse test uses
    float au = c1 * nu2, um = mx * 1.00000048f;
    float aum = au, umm = um;
#pragma unroll
    for (int o = 1; o < 32; o <<= 1) { aum = fmaxf(aum, __shfl_xor(aum, o, 64)); umm = fmaxf(umm, __shfl_xor(umm, o, 64)); }
    ub[ut * 32 + t] = make_float4(au, urow < A.n ? thr - bu : INFINITY, um, t == 0 ? aum : t == 1 ? umm : 0.f);
    unseeded = urow < A.n && (!g || !(s_n2[t] < 4.0e9f));      // (no bound, or a row that may overflow the half range: see score_filter_kernel)
    const int lp = A.last_poi[ur];
    double* gq = ugeo + (size_t)(ut * 32 + t) * 3;
    gq[0] = A.coords[2 * lp]; gq[1] = A.coords[2 * lp + 1]; gq[2] = A.cphi[lp];
  }
  if (t < 64 && __any(unseeded) && t == 0) A.tile_flag[ut] = 1;      // an unseeded user: the whole tile goes to the one-stage kernel
}

// Workgroup barrier that orders LDS traffic only (tile_engine.hip's lds_barrier): __syncthreads() also waits for every outstanding GLOBAL
// load of the wave (vmcnt(0)) - here the user tile requested two iterations ahead, i.e. a full L2 round trip per iteration: 2.7 us per
// 32 x 32 x 256 tile product instead of 0.5.
__device__ __forceinline__ void sf_lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

template <int D>
__global__ __launch_bounds__(256, 2) void score_filter_items_kernel(ScoreArgs A, const uint4* __restrict__ upk, const float4* __restrict__ ub,
                                                                    const double* __restrict__ ugeo, int n_utile) {
  constexpr int KG = D / 16, PF = KG * 64 / 256;      // uint4 of a user tile per thread
  extern __shared__ __align__(16) float dyn[];
  uint4* afb = reinterpret_cast<uint4*>(dyn);                        // [2][KG * 64]
  float* s_cu = dyn + 2 * KG * 64 * 4;                               // [2][32] threshold - bound
  float* s_mx = s_cu + 64;                                           // [2][2] tile maxima: c1 |u|_2, distance term
  double* s_ug = reinterpret_cast<double*>(s_mx + 4);                // [2][32][3] last-POI latitude, longitude, cos(latitude)
  double* s_geo = s_ug + 2 * 96;                                     // thr[n_dist]
  int* s_flag = reinterpret_cast<int*>(s_geo + A.n_dist + 2);        // [n_utile] tile flags, read once per item group (a tile that overflows
                                                                     // meanwhile only costs wasted work: it is redone by the one-stage kernel anyway)
  const int lane = lane_id(), w = wave_id(), li = lane & 31, h = lane >> 5, t = threadIdx.x;
  const int N = A.n_item, NB = A.n_dist + 1, ntile = (N + 31) / 32, groups = (ntile + 3) / 4;
  const float wd = A.wd ? A.wd[0] : 0.f;
  const float gscale = (float)(12742.0 * 1000.0 / A.dd);
  for (int i = t; i < A.n_dist; i += 256) s_geo[i] = A.thr[i];
  // the next user tile is requested while the current one is multiplied (one register set; the barriers do not wait for it: sf_lds_barrier)
  // (native vector types: an array of HIP's uint4 struct captured by the lambdas stayed in scratch memory - a store, a vmcnt(0) and a
  // reload per iteration)
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  typedef float f32x4v __attribute__((ext_vector_type(4)));
  u32x4 f0[PF]; f32x4v b0 = {0.f, INFINITY, 0.f, 0.f}; double g0 = 0.0;
  const u32x4* upk4 = reinterpret_cast<const u32x4*>(upk);
  const f32x4v* ub4 = reinterpret_cast<const f32x4v*>(ub);
  u32x4* afb4 = reinterpret_cast<u32x4*>(afb);
  auto fetch = [&](int ut) {
    ut = min(ut, n_utile - 1);
#pragma unroll
    for (int q = 0; q < PF; ++q) f0[q] = upk4[(size_t)ut * KG * 64 + t + 256 * q];
    b0 = ub4[ut * 32 + (t & 31)];
    g0 = ugeo[(size_t)ut * 96 + min(t, 95)];
  };
  auto put = [&](int buf) {
#pragma unroll
    for (int q = 0; q < PF; ++q) afb4[buf * KG * 64 + t + 256 * q] = f0[q];
    if (t < 32) {
      s_cu[buf * 32 + t] = b0.y;
      if (t < 2) s_mx[buf * 2 + t] = b0.w;           // the tile's maxima (sf_users_prep_kernel)
    }
    if (t < 96) s_ug[buf * 96 + t] = g0;
  };
  // Work unit = (chunk of user tiles, group of four item tiles), the chunk the SLOW index: every workgroup of the chip is on the same ~1 MB
  // of user fragments at a time, which then live in the 4 MB L2 of every XCD; the item table is re-read once per chunk (x 4 at 8192 users:
  // 20 GB from HBM).
  const int uc = max(1, (1 << 20) / (KG * 1024)), n_chunk = (n_utile + uc - 1) / uc;
  for (long long work = blockIdx.x; work < (long long)groups * n_chunk; work += gridDim.x) {
    const int g = (int)(work % groups), u_lo = (int)(work / groups) * uc, u_hi = min(n_utile, u_lo + uc);
    const int tile = min(4 * g + w, ntile - 1);
    const bool tvalid = 4 * g + w < ntile;
    uint4 b[KG];
#pragma unroll
    for (int m = 0; m < KG; ++m) b[m] = A.items_packed16[((size_t)tile * KG + m) * 64 + lane];
    const int j = tile * 32 + li;
    const bool jvalid = tvalid && j < N;
    const float2 nm = A.inorm[min(j, N - 1)];
    __syncthreads();                                 // (every wave is done with the buffers of the previous unit)
    for (int i = t; i < u_hi - u_lo; i += 256) s_flag[i] = __hip_atomic_load(A.tile_flag + u_lo + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    fetch(u_lo); put(0);
    __syncthreads();
    for (int ut = u_lo; ut < u_hi; ++ut) {
      const int buf = (ut - u_lo) & 1;
      fetch(ut + 1);
      if (!s_flag[ut - u_lo]) {                      // (unseeded / overflowed tiles: the one-stage kernel takes them)
        // all A fragments of the tile are requested before the first MFMA (read next to its MFMA, every product waited a full LDS round trip)
        uint4 a[KG];
#pragma unroll
        for (int m = 0; m < KG; ++m) a[m] = afb[buf * KG * 64 + m * 64 + lane];
        const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        f32x16 acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8, a[0]), __builtin_bit_cast(h8, b[0]), zero, 0, 0, 0);
#pragma unroll
        for (int m = 1; m < KG; ++m) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8, a[m]), __builtin_bit_cast(h8, b[m]), acc, 0, 0, 0);
        const float au = s_mx[buf * 2], ubt = s_mx[buf * 2 + 1];
        const float tb = __fmaf_rn(au, nm.x, nm.y);
        unsigned coarse = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {                // rows 8 q + 4 h .. + 3 of the tile: registers 4 q .. 4 q + 3
          const float4 c4 = *reinterpret_cast<const float4*>(s_cu + buf * 32 + 8 * q + 4 * h);
          coarse |= !((acc[4 * q] + ubt) + tb <= c4.x) ? (1u << (4 * q)) : 0u;
          coarse |= !((acc[4 * q + 1] + ubt) + tb <= c4.y) ? (2u << (4 * q)) : 0u;
          coarse |= !((acc[4 * q + 2] + ubt) + tb <= c4.z) ? (4u << (4 * q)) : 0u;
          coarse |= !((acc[4 * q + 3] + ubt) + tb <= c4.w) ? (8u << (4 * q)) : 0u;
        }
        if (!jvalid) coarse = 0;
        if (__any(coarse != 0)) {
          unsigned pass = 0;
          const int jc = min(j, N - 1);
          const double jlat = A.coords[2 * jc], jlon = A.coords[2 * jc + 1], jcp = A.cphi[jc];
          const double pr = 0.017453292519943295;
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            if (!__any((coarse >> r) & 1u)) continue;
            const int ul = (r & 3) + 8 * (r >> 2) + 4 * h;
            const double* ug = s_ug + buf * 96 + 3 * ul;
            int bin;
            {
#pragma clang fp contract(off)
              const double aa = (ug[0] - jlat) * pr;
              const double bb = (ug[1] - jlon) * pr;
              const double c = (1.0 - cos_small(aa)) / 2 + ug[2] * jcp * (1.0 - cos_small(bb)) / 2;
              bin = bin_of_c(c, s_geo, A.n_dist, gscale);
            }
            const float pv = A.sts[(size_t)(ut * 32 + ul) * NB + bin];
            const float up = __fmaf_rn(wd, pv, acc[r]) + tb;
            pass |= (((coarse >> r) & 1u) && !(up <= s_cu[buf * 32 + ul])) ? (1u << r) : 0u;
          }
          if (__any(pass != 0)) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              if (pass & (1u << r)) {
                const int urow = ut * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                const int pos = atomicAdd(A.surv_cnt + urow, 1);
                if (pos < SF_CAP) A.surv_idx[(size_t)urow * SF_CAP + pos] = j;
                else A.tile_flag[ut] = 1;
              }
            }
          }
        }
      }
      put(buf ^ 1);
      sf_lds_barrier();
    }
  }
}

// 64-lane bitonic sort, best (highest score, then lowest id) first - as score_topk.hip's merge
__device__ __forceinline__ void sf_wave_sort_desc(float& s, int& idx) {
  const int lane = lane_id();
#pragma unroll
  for (int k = 2; k <= 64; k <<= 1) {
#pragma unroll
    for (int j = k >> 1; j > 0; j >>= 1) {
      const float ps = __shfl_xor(s, j, 64);
      const int pi = __shfl_xor(idx, j, 64);
      const bool up = ((lane & k) == 0), lower = ((lane & j) == 0);
      const bool mine = (s > ps) || (s == ps && idx < pi);
      const bool keep = (up == lower) ? mine : !mine;
      if (!keep) { s = ps; idx = pi; }
    }
  }
}

// stage 2.  D8 = k-groups of 8 of the one-stage kernels (dim 64: 8, dim 128: 16, dim 256: 32): the same MFMA sequence, hence the same bits.
// The exact score of survivor `slot` of user u goes to surv_sc[u * SF_CAP + slot] (every survivor has its own slot: no atomics); after
// the workgroup's barrier one wave per user keeps the K best by (score desc, id asc) with the merge of topk_merge_kernel: lanes [0, K)
// hold the running best, lanes [K, 64) take the next 64 - K candidates, one bitonic sort per round.
template <int D8, int BINS>
__global__ __launch_bounds__(256) void score_rescore_kernel(ScoreArgs A) {
  extern __shared__ __align__(16) float dyn[];
  float4* af = reinterpret_cast<float4*>(dyn);             // [D8][64]
  __shared__ int s_pre[33];
  const int lane = lane_id(), w = wave_id(), li = lane & 31, h = lane >> 5, t = threadIdx.x;
  const int D = A.dim, N = A.n_item, K = A.k, NB = A.n_dist + 1;
  const int ut = blockIdx.x;
  if (A.tile_flag[ut]) return;                              // overflow: the one-stage kernel takes this tile
  const int ntile = (N + 31) / 32;
  {
    const int urow = min(ut * 32 + li, A.n - 1);
    const float* up = A.users + (size_t)urow * D;
    for (int m = w; m < D8; m += POI_NWAVE) {
      const int k0 = 8 * m + 4 * h;
      af[m * 64 + lane] = k0 < D ? *reinterpret_cast<const float4*>(up + k0) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
  if (t < 32) {
    const int urow = ut * 32 + t;
    const int c = urow < A.n ? min(A.surv_cnt[urow], SF_CAP) : 0;
    int inc = c;
#pragma unroll
    for (int o = 1; o < 32; o <<= 1) { const int v = __shfl_up(inc, o, 64); inc += t >= o ? v : 0; }
    s_pre[t + 1] = inc;
    if (t == 0) s_pre[0] = 0;
  }
  __syncthreads();
  const int total = s_pre[32];
  const float wd = (BINS && A.wd) ? A.wd[0] : 0.f;
  for (int c0 = 32 * w; c0 < total; c0 += 32 * POI_NWAVE) {
    const int flat = c0 + li;
    const bool valid = flat < total;
    int own = 0;
    if (valid) {
#pragma unroll
      for (int sft = 16; sft > 0; sft >>= 1) if (own + sft < 32 && s_pre[own + sft] <= flat) own += sft;      // largest i with pre[i] <= flat
    }
    const size_t slot = (size_t)(ut * 32 + own) * SF_CAP + (valid ? flat - s_pre[own] : 0);
    const int id = valid ? A.surv_idx[slot] : 0;
    float4 bf[D8];
#pragma unroll
    for (int m = 0; m < D8; ++m) {
      const int k0 = 8 * m + 4 * h;
      bf[m] = k0 < D ? ld4t(A.items, (size_t)id * D + k0, A.items_f16) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int m = 0; m < D8; ++m) {
      const float4 a = af[m * 64 + lane];
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, bf[m].x, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, bf[m].y, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, bf[m].z, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, bf[m].w, acc, 0, 0, 0);
    }
    // the pair (owner, item) sits in half-wave (owner >> 2) & 1, register (i & 3) + 4 (i >> 3) with i = owner - 4 h
    const bool mine = valid && h == ((own >> 2) & 1);
    const int ip = own - 4 * h, r_own = (ip & 3) + 4 * (ip >> 3);
    float a = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) a = r == r_own ? acc[r] : a;
    if (mine) {
      float pv = 0.f;
      if (BINS == 1 || BINS == 2) {
        const size_t cell = ((size_t)ut * ntile + (id >> 5)) * 64 + (id & 31) + 32 * h;
        int bin;
        if (BINS == 1) bin = reinterpret_cast<const unsigned char*>(A.ulptai)[cell * 16 + r_own];
        else bin = reinterpret_cast<const unsigned short*>(A.ulptai)[cell * 16 + r_own];
        pv = A.sts[(size_t)(ut * 32 + own) * NB + bin];
      } else if (BINS == 3) {         // GEO: the bin of (the owner's last train POI, item) from the coordinates, as the one-stage GEO kernels
        const int lp = A.last_poi[ut * 32 + own];
        int bin;
        {
#pragma clang fp contract(off)
          const double pr = 0.017453292519943295;
          const double aa = (A.coords[2 * lp] - A.coords[2 * id]) * pr;
          const double bb = (A.coords[2 * lp + 1] - A.coords[2 * id + 1]) * pr;
          const double c = (1.0 - cos_small(aa)) / 2 + A.cphi[lp] * A.cphi[id] * (1.0 - cos_small(bb)) / 2;
          bin = bin_of_c(c, A.thr, A.n_dist, (float)(12742.0 * 1000.0 / A.dd));
        }
        pv = A.sts[(size_t)(ut * 32 + own) * NB + bin];
      }
      A.surv_sc[slot] = __fmaf_rn(wd, pv, a);               // the one-stage kernels' expression (tile_epilogue)
    }
  }
  __syncthreads();                                          // (also orders the global score stores before the reads below)
  for (int i = w; i < 32; i += POI_NWAVE) {
    const int urow = ut * 32 + i;
    if (urow >= A.n) continue;
    const int n = s_pre[i + 1] - s_pre[i];
    const size_t base0 = (size_t)urow * SF_CAP;
    float sc = -INFINITY; int idx = INT_MAX;
    const int room = 64 - K;
    for (int base = 0; base < n; base += room) {
      const int c = base + (lane - K);
      if (lane >= K) {
        if (c < n) { sc = A.surv_sc[base0 + c]; idx = A.surv_idx[base0 + c]; }
        else { sc = -INFINITY; idx = INT_MAX; }
      }
      sf_wave_sort_desc(sc, idx);
    }
    if (lane < K) {
      A.idx_out[(size_t)urow * K + lane] = idx == INT_MAX ? -1 : idx;
      if (A.score_out) A.score_out[(size_t)urow * K + lane] = sc;
    }
  }
}

bool score_two_stage_supported(const ScoreArgs& A) {
  if (!(A.k > 0 && A.seeded && !A.prob && !A.scores && A.n >= 128 && A.n_dist + 1 <= 1024)) return false;
  return A.geo ? (A.dim == 64 || A.dim == 128 || A.dim == 256) : (A.dim == 64 || A.dim == 128);      // (the users' bin probabilities live in LDS: 32 x (n_dist + 1) floats)
}

// user tiles per workgroup: two while both fit twice into a CU's LDS (two workgroups per CU), else one
static int score_filter_ut(int dim, int n_dist, bool bins, bool geo) {
  const size_t per_tile = sizeof(float) * ((size_t)(dim / 16) * 64 * 4 + 128 + (bins ? ((32 * (size_t)(n_dist + 1) + 3) & ~(size_t)3) : 0));
  return (!geo && 2 * per_tile <= 79 * 1024) ? 2 : 1;
}
size_t score_filter_lds(int dim, int n_dist, bool bins, bool geo) {
  const size_t per_tile = sizeof(float) * ((size_t)(dim / 16) * 64 * 4 + 128 + (bins ? ((32 * (size_t)(n_dist + 1) + 3) & ~(size_t)3) : 0));
  return score_filter_ut(dim, n_dist, bins, geo) * per_tile + (geo ? sizeof(double) * (size_t)(n_dist + 96) + sizeof(float) * 32 : 0);
}

template <int D>
static hipError_t launch_two_stage_t(const ScoreArgs& A, int n_split_f, hipStream_t st, Timing* tm) {
  const int ntile = (A.n_item + 31) / 32, n_utile = (A.n + 31) / 32;
  const int bins = A.geo ? 3 : A.ulptai ? (A.bin_bytes == 1 ? 1 : 2) : 0;
  tm->begin("pack_items", st);
  hipLaunchKernelGGL(pack_items_f16_kernel<D>, dim3(ntile), dim3(256), 0, st, A.items, A.items_f16, A.n_item, const_cast<uint4*>(A.items_packed16), const_cast<float2*>(A.inorm));
  tm->end(st);
  ScoreArgs F = A; F.n_split = n_split_f;
  const size_t lds = score_filter_lds(D, A.n_dist, bins != 0, bins == 3);
  const int ut = score_filter_ut(D, A.n_dist, bins != 0, bins == 3);
  const dim3 grid((n_utile + ut - 1) / ut, n_split_f / POI_NWAVE);
  static DeviceOnce once;      // (per device: poi_common.h)
  {
    const hipError_t oe = once.run([&]() -> hipError_t {
    hipError_t e = hipSuccess;
    auto big = [&](const void* f) { if (e == hipSuccess) e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); };
    big(reinterpret_cast<const void*>(&score_filter_kernel<D, 3, 1>));
    if constexpr (D <= 128) {
      big(reinterpret_cast<const void*>(&score_filter_kernel<D, 2, 1>)); big(reinterpret_cast<const void*>(&score_filter_kernel<D, 1, 1>));
      big(reinterpret_cast<const void*>(&score_filter_kernel<D, 2, 2>)); big(reinterpret_cast<const void*>(&score_filter_kernel<D, 1, 2>));
      big(reinterpret_cast<const void*>(&score_filter_kernel<D, 0, 2>));
    }
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&score_rescore_kernel<D / 8, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
    return e;
    });
    if (oe != hipSuccess) return oe;
  }
  tm->begin("score_filter", st);
  if (bins == 3 && A.users_packed16) {        // item-stationary (huge item table, few users): see score_filter_items_kernel
    hipLaunchKernelGGL(sf_users_prep_kernel<D>, dim3(n_utile), dim3(256), 0, st, F, A.users_packed16, A.ubound, A.ugeo);
    const size_t ldsi = sizeof(uint4) * 2 * (D / 16) * 64 + sizeof(float) * (64 + 4) + sizeof(double) * (2 * 96 + (size_t)A.n_dist + 2) + sizeof(int) * ((size_t)n_utile + 4);
    const int groups = (ntile + 3) / 4;
    const long long units = (long long)groups * ((n_utile + ((1 << 20) / ((D / 16) * 1024)) - 1) / ((1 << 20) / ((D / 16) * 1024)));
    const int gridi = units < A.n_cu * 2 ? (int)units : A.n_cu * 2;
    hipLaunchKernelGGL(score_filter_items_kernel<D>, dim3(gridi), dim3(256), ldsi, st, F, A.users_packed16, A.ubound, A.ugeo, n_utile);
  } else
  if (bins == 3) hipLaunchKernelGGL((score_filter_kernel<D, 3, 1>), grid, dim3(256), lds, st, F);
  else if constexpr (D <= 128) {
    if (bins == 1 && ut == 2) hipLaunchKernelGGL((score_filter_kernel<D, 1, 2>), grid, dim3(256), lds, st, F);
    else if (bins == 1) hipLaunchKernelGGL((score_filter_kernel<D, 1, 1>), grid, dim3(256), lds, st, F);
    else if (bins == 2 && ut == 2) hipLaunchKernelGGL((score_filter_kernel<D, 2, 2>), grid, dim3(256), lds, st, F);
    else if (bins == 2) hipLaunchKernelGGL((score_filter_kernel<D, 2, 1>), grid, dim3(256), lds, st, F);
    else hipLaunchKernelGGL((score_filter_kernel<D, 0, 2>), grid, dim3(256), lds, st, F);
  } else return hipErrorInvalidValue;
  tm->end(st);
  constexpr int D8 = D / 8;
  const size_t lds2 = sizeof(float) * ((size_t)D8 * 64 * 4);
  tm->begin("score_rescore", st);
  if (bins == 3) hipLaunchKernelGGL((score_rescore_kernel<D8, 3>), dim3(n_utile), dim3(256), lds2, st, A);
  else if constexpr (D <= 128) {
    if (bins == 1) hipLaunchKernelGGL((score_rescore_kernel<D8, 1>), dim3(n_utile), dim3(256), lds2, st, A);
    else if (bins == 2) hipLaunchKernelGGL((score_rescore_kernel<D8, 2>), dim3(n_utile), dim3(256), lds2, st, A);
    else hipLaunchKernelGGL((score_rescore_kernel<D8, 0>), dim3(n_utile), dim3(256), lds2, st, A);
  }
  tm->end(st);
  return hipGetLastError();
}

// self-seeding of an unseeded call (resident bin matrix or no distance term, dims 64 / 128): block maxima of the approximate lower bounds +
// K-th largest per user -> gbound.  Uses surv_sc as scratch and leaves the packed items / norms for the filter pass.
template <int D>
static hipError_t launch_maxpass_t(const ScoreArgs& A, int n_split_f, hipStream_t st, Timing* tm) {
  if constexpr (D > 128) { return hipErrorInvalidValue; } else {
  const int ntile = (A.n_item + 31) / 32, n_utile = (A.n + 31) / 32;
  const int bins = A.ulptai ? (A.bin_bytes == 1 ? 1 : 2) : 0;
  int nsm = n_split_f >= 64 ? 64 : n_split_f >= 32 ? 32 : 16;      // 512 .. 2048 blocks per user
  if (const char* e = getenv("POI_SF_NSM")) { const int v = atoi(e); if (v == 16 || v == 32 || v == 64) nsm = v; }      // tuning switch
  tm->begin("pack_items", st);
  hipLaunchKernelGGL(pack_items_f16_kernel<D>, dim3(ntile), dim3(256), 0, st, A.items, A.items_f16, A.n_item, const_cast<uint4*>(A.items_packed16), const_cast<float2*>(A.inorm));
  tm->end(st);
  ScoreArgs F = A; F.n_split = nsm;
  // every second item tile: the K-th largest block maximum over half of the items is about the 2 K-th best score - 40 survivors per user
  // instead of 21 for half of this pass (measured at the Gowalla shape, strides 1 / 2 / 4 / 8: 6.5 / 5.2 / 5.3 / 6.6 ms per unseeded call)
  F.max_stride = ntile >= 64 * 8 ? 2 : 1;
  if (const char* e = getenv("POI_SF_MAXSTRIDE")) { const int v = atoi(e); if (v >= 1 && v <= 16) F.max_stride = v; }      // tuning switch
  const size_t lds = score_filter_lds(D, A.n_dist, bins != 0, false);
  const int ut = score_filter_ut(D, A.n_dist, bins != 0, false);
  const dim3 grid((n_utile + ut - 1) / ut, nsm / POI_NWAVE);
  static DeviceOnce once;
  {
    const hipError_t oe = once.run([&]() -> hipError_t {
    hipError_t e = hipSuccess;
    auto big = [&](const void* f) { if (e == hipSuccess) e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); };
    big(reinterpret_cast<const void*>(&score_filter_kernel<D, 2, 1, true>)); big(reinterpret_cast<const void*>(&score_filter_kernel<D, 1, 1, true>));
    big(reinterpret_cast<const void*>(&score_filter_kernel<D, 2, 2, true>)); big(reinterpret_cast<const void*>(&score_filter_kernel<D, 1, 2, true>));
    big(reinterpret_cast<const void*>(&score_filter_kernel<D, 0, 2, true>));
    return e;
    });
    if (oe != hipSuccess) return oe;
  }
  tm->begin("score_maxpass", st);
  if (bins == 1 && ut == 2) hipLaunchKernelGGL((score_filter_kernel<D, 1, 2, true>), grid, dim3(256), lds, st, F);
  else if (bins == 1) hipLaunchKernelGGL((score_filter_kernel<D, 1, 1, true>), grid, dim3(256), lds, st, F);
  else if (bins == 2 && ut == 2) hipLaunchKernelGGL((score_filter_kernel<D, 2, 2, true>), grid, dim3(256), lds, st, F);
  else if (bins == 2) hipLaunchKernelGGL((score_filter_kernel<D, 2, 1, true>), grid, dim3(256), lds, st, F);
  else hipLaunchKernelGGL((score_filter_kernel<D, 0, 2, true>), grid, dim3(256), lds, st, F);
  const dim3 gs((A.n + POI_NWAVE - 1) / POI_NWAVE);
  if (nsm == 64) hipLaunchKernelGGL(sf_select_kernel<32>, gs, dim3(256), 0, st, F, A.k);
  else if (nsm == 32) hipLaunchKernelGGL(sf_select_kernel<16>, gs, dim3(256), 0, st, F, A.k);
  else hipLaunchKernelGGL(sf_select_kernel<8>, gs, dim3(256), 0, st, F, A.k);
  tm->end(st);
  return hipGetLastError();
  }
}
bool score_maxpass_supported(const ScoreArgs& A) { return !A.geo && (A.dim == 64 || A.dim == 128) && A.k > 0 && A.k <= 64 && (A.n_item + 31) / 32 >= 64 * 4; }
hipError_t launch_score_maxpass(const ScoreArgs& A, int n_split_f, hipStream_t st, Timing* tm) {
  if (A.dim == 64) return launch_maxpass_t<64>(A, n_split_f, st, tm);
  if (A.dim == 128) return launch_maxpass_t<128>(A, n_split_f, st, tm);
  return hipErrorInvalidValue;
}

// stage 1 + stage 2; the caller then runs the one-stage kernel + merge with A.tile_flag set (they skip every tile that is not flagged)
hipError_t launch_score_two_stage(const ScoreArgs& A, int n_split_f, hipStream_t st, Timing* tm) {
  if (A.dim == 64) return launch_two_stage_t<64>(A, n_split_f, st, tm);
  if (A.dim == 128) return launch_two_stage_t<128>(A, n_split_f, st, tm);
  if (A.dim == 256) return launch_two_stage_t<256>(A, n_split_f, st, tm);
  return hipErrorInvalidValue;
}

int score_filter_cap() { return SF_CAP; }

// gbound[u] = max(gbound[u], order-mapped (K-th best score of the pre-pass list - 1 ulp)): the K-th best exact score of a SUBSET of the items
// bounds the final K-th best from below (`score > bound` keeps ties, as compact_user publishes)
__global__ __launch_bounds__(256) void topk_bound_kernel(const float* __restrict__ score_k, int n, int k, unsigned* __restrict__ gbound) {
  const int u = blockIdx.x * 256 + threadIdx.x;
  if (u >= n) return;
  const float s = score_k[(size_t)u * k + (k - 1)];
  if (!(s > -INFINITY)) return;                 // fewer than K items in the subset (or NaN): no bound
  const unsigned o = sf_f2ord(s);
  if (o > 1u && o - 1u > gbound[u]) gbound[u] = o - 1u;
}
hipError_t launch_topk_bound(const float* score_k, int n, int k, unsigned* gbound, hipStream_t st) {
  hipLaunchKernelGGL(topk_bound_kernel, dim3((n + 255) / 256), dim3(256), 0, st, score_k, n, k, gbound);
  return hipGetLastError();
}

}  // namespace poi
