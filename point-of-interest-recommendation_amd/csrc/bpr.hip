// BPR-MF step (OboBpr.bpr_train, public/BPR.py:201-241): u = ux[u] . (lt[p] - lt[q]), g = -sigmoid(-u);
//   ux[u] -= a (g (xp - xq) + l ux[u]);   lt[p] -= a (g usr + l xp);   lt[q] -= a (-g usr + l xq).
// The whole step is three row gathers + three row scatters: HBM-bound, 6 D e + 12 bytes per triple.
//
// SNAPSHOT mode (round 5: rebuilt; no float atomics - identical launches give bitwise identical tables).  Every triple's gradient is
// evaluated at the launch-entry values and a row touched by k triples moves by min(k, cap) / k of the sum of their updates (the batch
// rule of include/poi_hip.h).  A launch of n triples is 3 n table touches - (user row, +g (xp - xq)), (positive row, +g usr),
// (negative row, -g usr) - which a stable radix sort groups by row (te_scatter.hip's sort; user rows first).  Then
//   bpr_chunk<users>  one wave per 64 sorted user touches: for every run of equal rows the user row is read ONCE, each triple's two POI
//                     rows are gathered, g_i / loss_i come out of the dot product and the run's sum of g_i (xp - xq) stays in registers;
//                     a complete run writes the new user row to a SHADOW table (the POI pass still needs the entry values of ux);
//   bpr_chunk<items>  the same over the 2 n POI touches: sum of +-g_i ux[u_i], new POI row written in place (half tables: float32
//                     arithmetic, nearest or stochastic rounding on the way back);
//   bpr_span          runs cut by a 64-touch window leave their partial sums behind (one opening and one closing partial per window,
//                     as te_psum) - summed in window order and applied;
//   bpr_commit        shadow rows -> ux.
// Rows moved: 2 per triple (user pass) + 2 per triple (POI pass) + every touched row read and written once: ~4.4 rows per triple at the
// Gowalla shape against the 6 of the per-triple formulation - the user row of a run and the target rows are not re-read per triple.
// HOGWILD mode: the round-1 in-place kernel (racy by definition; identical to the reference whenever no row is shared inside the launch).
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
    // (an id outside its table: the reference raises IndexError - here the triple is skipped, counted, its loss NaN: include/poi_hip.h, ABI 6)
    if ((unsigned)A.uidx[i] >= (unsigned)A.n_user || (unsigned)A.p[i] > (unsigned)A.n_item || (unsigned)A.q[i] > (unsigned)A.n_item) {
      if (gl == 0) { atomicAdd(A.bad, 1); A.loss[i] = __int_as_float(0x7fc00000); }
      continue;
    }
    float* ur = A.ux + (size_t)A.uidx[i] * D;
    float* pr = reinterpret_cast<float*>(A.lt) + (size_t)A.p[i] * D;      // (float32 tables only: launch_bpr)
    float* qr = reinterpret_cast<float*>(A.lt) + (size_t)A.q[i] * D;
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

// -------------------------------------------------------------------------------------------------
// SNAPSHOT mode
// -------------------------------------------------------------------------------------------------
// touch e < 3 n: e / n = 0 user, 1 positive, 2 negative row of triple e % n; key = row in the unified space [users | POI rows]
__global__ __launch_bounds__(256) void bpr_keys_kernel(BprArgs A) {
  const int n = A.n;
  if (blockIdx.x == 0 && threadIdx.x == 0) A.cnt[0] = 3 * n;
  for (int e = blockIdx.x * 256 + threadIdx.x; e < 3 * n; e += gridDim.x * 256) {
    const int kind = e >= 2 * n ? 2 : e >= n ? 1 : 0, i = e - kind * n;
    // (ids clamped into their tables: the reference raises IndexError on a bad id, a device kernel must not write outside; the triple of a bad id
    // gets g = 0 and a NaN loss in bpr_chunk<users> and is counted here: poi_ctx_take_bad_ids)
    const unsigned raw = kind == 0 ? (unsigned)A.uidx[i] : (unsigned)(kind == 1 ? A.p[i] : A.q[i]);
    if (kind == 0 ? raw >= (unsigned)A.n_user : raw > (unsigned)A.n_item) atomicAdd(A.bad, 1);
    A.keys0[e] = kind == 0 ? (int)min((unsigned)A.uidx[i], (unsigned)(A.n_user - 1))
                           : A.n_user + (int)min((unsigned)(kind == 1 ? A.p[i] : A.q[i]), (unsigned)A.n_item);
  }
}

template <int NT> struct BprVec { float4 v[NT]; };

template <int LPR, int NT>
__device__ __forceinline__ void bpr_zero(BprVec<NT>& a) {
#pragma unroll
  for (int t = 0; t < NT; ++t) a.v[t] = make_float4(0.f, 0.f, 0.f, 0.f);
}
// row <- row - alpha min(k, cap) (sum / k + lambda row)
__device__ __forceinline__ float4 bpr_rule(float4 r, float4 s, float sc, float inv, float lm) {
  return make_float4(r.x - sc * (s.x * inv + lm * r.x), r.y - sc * (s.y * inv + lm * r.y), r.z - sc * (s.z * inv + lm * r.z), r.w - sc * (s.w * inv + lm * r.w));
}
// one finished row: PART 0 -> shadow[user] from ux[user]; PART 1 -> lt[row] in place.  Lanes gl < LPR of ONE group call it.
template <int PART, int LPR, int NT>
__device__ __forceinline__ void bpr_apply(const BprArgs& A, int key, const BprVec<NT>& sum, int k, int gl) {
  const int D = A.dim;
  const float sc = A.alpha * fminf((float)k, A.bcap), inv = 1.0f / (float)k;
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int col = (t * LPR + gl) * 4;
    if (col >= D) continue;
    if (PART == 0) {
      const size_t off = (size_t)key * D + col;
      *reinterpret_cast<float4*>(A.shadow + off) = bpr_rule(*reinterpret_cast<const float4*>(A.ux + off), sum.v[t], sc, inv, A.lambda);
    } else {
      const size_t off = (size_t)(key - A.n_user) * D + col;
      st4t_sr(A.lt, off, A.lt_f16, bpr_rule(ld4t(A.lt, off, A.lt_f16), sum.v[t], sc, inv, A.lambda), A.sr_salt, off);
    }
  }
}

// One wave per window of 64 sorted touches of PART (0: user rows [0, n), 1: POI rows [n, 3n)).  LPR lanes per table row (float4 each,
// NT column passes), EPW = 64 / LPR touches of a run in flight per pass, two passes unrolled.
template <int PART, int LPR, int NT>
__global__ __launch_bounds__(256) void bpr_chunk_kernel(BprArgs A) {
  constexpr int EPW = 64 / LPR, U = 2;
  const int lane = lane_id(), grp = lane / LPR, gl = lane % LPR;
  const int n = A.n, D = A.dim;
  const int part_b = PART ? n : 0, part_e = PART ? 3 * n : n;
  const int n_chunk = (part_e - part_b + 63) / 64;
  const int chunk0 = PART ? (n + 63) / 64 : 0;
  for (int c = blockIdx.x * 4 + wave_id(); c < n_chunk; c += gridDim.x * 4) {
    const int j0 = part_b + 64 * c, nv = min(64, part_e - j0);
    const bool valid = lane < nv;
    const int key = valid ? A.ks[j0 + lane] : -1;
    const int val = valid ? A.vs[j0 + lane] : 0;
    const int up = __shfl_up(key, 1, 64);
    const int prev = lane == 0 ? (c > 0 ? A.ks[j0 - 1] : -2) : up;
    const int nextk = (j0 + nv < part_e) ? A.ks[j0 + nv] : -3;
    const unsigned long long starts = __ballot(valid && key != prev);
    int lead_cnt = 0, lead_more = 0, trail_cnt = 0, trail_row = -1;
    int a = 0;
    while (a < nv) {
      const unsigned long long above = a + 1 < 64 ? (starts >> (a + 1)) << (a + 1) : 0ull;
      const int b = above ? min(nv, (int)__builtin_ctzll(above)) : nv;
      const int row = __builtin_amdgcn_readfirstlane(__shfl(key, a, 64));
      const bool cont_before = a == 0 && !(starts & 1ull);
      const bool cont_after = b == nv && nextk == row;
      BprVec<NT> acc; bpr_zero<LPR, NT>(acc);
      BprVec<NT> ur;      // PART 0: the run's user row (entry values), read once
      if (PART == 0) {
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          const int col = (t * LPR + gl) * 4;
          ur.v[t] = col < D ? *reinterpret_cast<const float4*>(A.ux + (size_t)row * D + col) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
      }
      for (int e0 = a; e0 < b; e0 += EPW * U) {
        bool ok[U]; int tri[U]; float sg[U];
        BprVec<NT> x[U];      // PART 0: xp - xq; PART 1: ux[u_i]
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int idx = e0 + u * EPW + grp;
          ok[u] = idx < b;
          const int e = __shfl(val, idx & 63, 64);
          if (PART == 0) {
            tri[u] = ok[u] ? e : 0; sg[u] = 0.f;
            const size_t pr = (size_t)min((unsigned)A.p[tri[u]], (unsigned)A.n_item) * D, qr = (size_t)min((unsigned)A.q[tri[u]], (unsigned)A.n_item) * D;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
              const int col = (t * LPR + gl) * 4;
              if (ok[u] && col < D) {
                const float4 xp = ld4t(A.lt, pr + col, A.lt_f16), xq = ld4t(A.lt, qr + col, A.lt_f16);
                x[u].v[t] = make_float4(xp.x - xq.x, xp.y - xq.y, xp.z - xq.z, xp.w - xq.w);
              } else x[u].v[t] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
          } else {
            const int kind = e >= 2 * n ? 2 : 1;
            tri[u] = ok[u] ? e - kind * n : 0;
            sg[u] = ok[u] ? (kind == 2 ? -A.g[tri[u]] : A.g[tri[u]]) : 0.f;
            const size_t us = (size_t)min((unsigned)A.uidx[tri[u]], (unsigned)(A.n_user - 1)) * D;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
              const int col = (t * LPR + gl) * 4;
              x[u].v[t] = (ok[u] && col < D) ? *reinterpret_cast<const float4*>(A.ux + us + col) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
          }
        }
        if (PART == 0) {
#pragma unroll
          for (int u = 0; u < U; ++u) {
            float dot = 0.f;
#pragma unroll
            for (int t = 0; t < NT; ++t) dot += dot4(ur.v[t], x[u].v[t]);
            dot = group_sum<LPR>(dot);
            const bool bad = (unsigned)A.uidx[tri[u]] >= (unsigned)A.n_user || (unsigned)A.p[tri[u]] > (unsigned)A.n_item || (unsigned)A.q[tri[u]] > (unsigned)A.n_item;
            sg[u] = (ok[u] && !bad) ? -sigmoidf_(-dot) : 0.f;      // (a triple with an id outside its table moves nothing)
            if (ok[u] && gl == 0) { A.g[tri[u]] = sg[u]; A.loss[tri[u]] = bad ? __int_as_float(0x7fc00000) : -log_sigmoidf_(dot); }
          }
        }
#pragma unroll
        for (int u = 0; u < U; ++u)      // (touches of a group in ascending order: a fixed summation order)
#pragma unroll
          for (int t = 0; t < NT; ++t) {
            acc.v[t].x = fmaf(sg[u], x[u].v[t].x, acc.v[t].x); acc.v[t].y = fmaf(sg[u], x[u].v[t].y, acc.v[t].y);
            acc.v[t].z = fmaf(sg[u], x[u].v[t].z, acc.v[t].z); acc.v[t].w = fmaf(sg[u], x[u].v[t].w, acc.v[t].w);
          }
      }
      // the groups' sums -> one total, the same in every group (pairwise, commutative at each level)
#pragma unroll
      for (int o = LPR; o < 64; o <<= 1)
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          acc.v[t].x += __shfl_xor(acc.v[t].x, o, 64); acc.v[t].y += __shfl_xor(acc.v[t].y, o, 64);
          acc.v[t].z += __shfl_xor(acc.v[t].z, o, 64); acc.v[t].w += __shfl_xor(acc.v[t].w, o, 64);
        }
      if (!cont_before && !cont_after) {
        if (grp == 0) bpr_apply<PART, LPR, NT>(A, row, acc, b - a, gl);
      } else {
        float* dst = (cont_before ? A.lead : A.trail) + (size_t)(chunk0 + c) * D;
        if (grp == 0) {
#pragma unroll
          for (int t = 0; t < NT; ++t) {
            const int col = (t * LPR + gl) * 4;
            if (col < D) *reinterpret_cast<float4*>(dst + col) = acc.v[t];
          }
        }
        if (cont_before) { lead_cnt = b - a; lead_more = cont_after ? 1 : 0; }
        else { trail_cnt = b - a; trail_row = row; }
      }
      a = b;
    }
    if (lane == 0) A.meta[chunk0 + c] = make_int4(lead_cnt, lead_more, trail_cnt, trail_row);
  }
}

// rows cut by window boundaries: the window where the row STARTS (its closing run) owns it and walks the following windows' opening runs
template <int PART, int LPR, int NT>
__global__ __launch_bounds__(256) void bpr_span_kernel(BprArgs A) {
  constexpr int EPW = 64 / LPR;
  const int lane = lane_id(), grp = lane / LPR, gl = lane % LPR;
  const int n = A.n, D = A.dim;
  const int n_chunk = ((PART ? 2 * n : n) + 63) / 64, chunk0 = PART ? (n + 63) / 64 : 0;
  for (int c = (blockIdx.x * 4 + wave_id()) * EPW + grp; c < n_chunk; c += gridDim.x * 4 * EPW) {
    const int4 m = A.meta[chunk0 + c];
    if (m.z == 0) continue;
    BprVec<NT> sum;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int col = (t * LPR + gl) * 4;
      sum.v[t] = col < D ? *reinterpret_cast<const float4*>(A.trail + (size_t)(chunk0 + c) * D + col) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    int k = m.z;
    for (int c2 = c + 1; c2 < n_chunk; ++c2) {
      const int4 m2 = A.meta[chunk0 + c2];
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const int col = (t * LPR + gl) * 4;
        if (col < D) {
          const float4 v = *reinterpret_cast<const float4*>(A.lead + (size_t)(chunk0 + c2) * D + col);
          sum.v[t] = make_float4(sum.v[t].x + v.x, sum.v[t].y + v.y, sum.v[t].z + v.z, sum.v[t].w + v.w);
        }
      }
      k += m2.x;
      if (!m2.y) break;
    }
    bpr_apply<PART, LPR, NT>(A, m.w, sum, k, gl);
  }
}

// shadow rows of the launch's users -> ux (after the POI pass has read the entry values)
template <int LPR, int NT>
__global__ __launch_bounds__(256) void bpr_commit_kernel(BprArgs A) {
  constexpr int EPW = 64 / LPR;
  const int lane = lane_id(), grp = lane / LPR, gl = lane % LPR;
  const int n = A.n, D = A.dim;
  const int n_chunk = (n + 63) / 64;
  for (int c = blockIdx.x * 4 + wave_id(); c < n_chunk; c += gridDim.x * 4) {
    const int j0 = 64 * c, nv = min(64, n - j0);
    const bool valid = lane < nv;
    const int key = valid ? A.ks[j0 + lane] : -1;
    const int up = __shfl_up(key, 1, 64);
    const int prev = lane == 0 ? (c > 0 ? A.ks[j0 - 1] : -2) : up;
    const int st = (valid && key != prev) ? key : -1;
    for (int l0 = 0; l0 < nv; l0 += EPW) {
      const int row = __shfl(st, (l0 + grp) & 63, 64);
      if (l0 + grp >= nv || row < 0) continue;
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const int col = (t * LPR + gl) * 4;
        if (col < D) *reinterpret_cast<float4*>(A.ux + (size_t)row * D + col) = *reinterpret_cast<const float4*>(A.shadow + (size_t)row * D + col);
      }
    }
  }
}

template <int LPR, int NT>
static hipError_t launch_bpr_snapshot_t(BprArgs& A, int num_cu, hipStream_t st, Timing* tm) {
  const int n = A.n;
  tm->begin("bpr_sort", st);
  hipLaunchKernelGGL(bpr_keys_kernel, dim3(min(num_cu * 8, (3 * n + 255) / 256)), dim3(256), 0, st, A);
  int bits = 1;
  while ((1ll << bits) < (long long)A.n_user + A.n_item + 2) ++bits;
  const int *ks = nullptr, *vs = nullptr;
  hipError_t e = launch_radix_sort(A.keys0, A.keys1, A.vals0, A.vals1, A.cnt, bits, A.hist, st, &ks, &vs);
  if (e != hipSuccess) return e;
  A.ks = ks; A.vs = vs;
  tm->end(st);
  const int cu = (n + 63) / 64, ci = (2 * n + 63) / 64;
  auto grid = [&](int chunks, int per) { return dim3(max(1, min(num_cu * 16, (chunks + per - 1) / per))); };
  tm->begin("bpr_users", st);
  hipLaunchKernelGGL((bpr_chunk_kernel<0, LPR, NT>), grid(cu, 4), dim3(256), 0, st, A);
  hipLaunchKernelGGL((bpr_span_kernel<0, LPR, NT>), grid(cu, 4 * (64 / LPR)), dim3(256), 0, st, A);
  tm->end(st);
  tm->begin("bpr_items", st);
  hipLaunchKernelGGL((bpr_chunk_kernel<1, LPR, NT>), grid(ci, 4), dim3(256), 0, st, A);
  hipLaunchKernelGGL((bpr_span_kernel<1, LPR, NT>), grid(ci, 4 * (64 / LPR)), dim3(256), 0, st, A);
  hipLaunchKernelGGL((bpr_commit_kernel<LPR, NT>), grid(cu, 4), dim3(256), 0, st, A);
  tm->end(st);
  return hipGetLastError();
}

template <int LPT>
static hipError_t launch_bpr_hogwild_t(const BprArgs& A, hipStream_t st, Timing* tm) {
  const int gpb = POI_BLOCK / LPT;
  int grid = (A.n + gpb - 1) / gpb;
  if (grid > 256 * 16) grid = 256 * 16;
  if (grid < 1) grid = 1;
  tm->begin("bpr_hogwild", st);
  hipLaunchKernelGGL(bpr_hogwild_kernel<LPT>, dim3(grid), dim3(POI_BLOCK), 0, st, A);
  tm->end(st);
  return hipGetLastError();
}

// workspace of a snapshot launch of n triples (see abi.hip poi_bpr_step): ints, floats
void bpr_ws_sizes(int n, int dim, size_t* n_int, size_t* n_float) {
  const size_t chunks = (size_t)(n + 63) / 64 + (size_t)(2 * (size_t)n + 63) / 64 + 2;
  *n_int = 4 * (3 * (size_t)n + 64) + RS_HIST_INTS + RS_MAXBIN + 64 + 4 * chunks;
  *n_float = (((size_t)n + 64 + 3) & ~(size_t)3) + 2 * chunks * (size_t)dim;
}

hipError_t launch_bpr(BprArgs& A, int mode, int num_cu, hipStream_t st, Timing* tm) {
  if (mode == 1) {
    if (A.lt_f16) return hipErrorInvalidValue;
    if (A.dim <= 64) return launch_bpr_hogwild_t<16>(A, st, tm);
    if (A.dim <= 128) return launch_bpr_hogwild_t<32>(A, st, tm);
    return launch_bpr_hogwild_t<64>(A, st, tm);
  }
  if (A.dim <= 64) return launch_bpr_snapshot_t<16, 1>(A, num_cu, st, tm);
  if (A.dim <= 128) return launch_bpr_snapshot_t<32, 1>(A, num_cu, st, tm);
  if (A.dim <= 256) return launch_bpr_snapshot_t<64, 1>(A, num_cu, st, tm);
  if (A.dim <= 512) return launch_bpr_snapshot_t<64, 2>(A, num_cu, st, tm);
  if (A.dim <= 1024) return launch_bpr_snapshot_t<64, 4>(A, num_cu, st, tm);
  return hipErrorInvalidValue;
}

}  // namespace poi
