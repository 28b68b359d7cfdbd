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
// the per-user thresholds and appends survivors to per-user LDS candidate lists, compacted by a
// 64-lane bitonic sort (wavefront shuffles) whenever a list could overflow.
#include "poi_common.h"
#include "poi_kernels.h"
#include <limits.h>

namespace poi {

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define TK_CAP 64   // candidate slots per user per wave (>= K + 32)

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
  int ci[32][TK_CAP];
  int cnt[32];
  float thr[32];
};

__device__ __forceinline__ void wave_fence() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

// Keep the best K of user i's candidate list; update its threshold.
__device__ __forceinline__ void compact_user(WaveTopk& T, int i, int K) {
  const int lane = lane_id();
  const int n = T.cnt[i];
  float s = lane < n ? T.cs[i][lane] : -INFINITY;
  int idx = lane < n ? T.ci[i][lane] : INT_MAX;
  wave_sort_desc(s, idx);
  wave_fence();
  if (lane < K) { T.cs[i][lane] = s; T.ci[i][lane] = idx; }
  const float kth = __shfl(s, K - 1, 64);
  if (lane == 0) { T.cnt[i] = n < K ? n : K; T.thr[i] = n >= K ? kth : -INFINITY; }
  wave_fence();
}

template <int D8, bool DB>
__global__ __launch_bounds__(POI_BLOCK) void score_kernel(ScoreArgs A) {
  __shared__ WaveTopk tk[POI_NWAVE];
  const int lane = lane_id(), w = wave_id();
  const int li = lane & 31, h = lane >> 5;
  const int D = A.dim, N = A.n_item, K = A.k;
  const int ut = blockIdx.x;
  const int split = blockIdx.y * POI_NWAVE + w;
  const int ntile = (N + 31) / 32;
  const int tps = (ntile + A.n_split - 1) / A.n_split;
  const int t_begin = split * tps;
  const int t_end = min(ntile, t_begin + tps);
  WaveTopk& T = tk[w];
  if (K > 0) {
    if (lane < 32) { T.cnt[lane] = 0; T.thr[lane] = -INFINITY; }
    wave_fence();
  }

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
  const float wd = (A.prob && A.wd) ? A.wd[0] : 0.f;

  auto load_b = [&](float4 (&bf)[D8], int tile) {
    const int irow = min(tile * 32 + li, N - 1);
    const float* ip = A.items + (size_t)irow * D;
#pragma unroll
    for (int m = 0; m < D8; ++m) {
      const int k0 = 8 * m + 4 * h;
      bf[m] = k0 < D ? *reinterpret_cast<const float4*>(ip + k0) : make_float4(0.f, 0.f, 0.f, 0.f);
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
    // epilogue: C layout col = lane&31 (item), row = (r&3) + 8*(r>>2) + 4*h (user)
    const int j = tile * 32 + li;
    const bool jvalid = j < N;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int i = (r & 3) + 8 * (r >> 2) + 4 * h;
      const int urow = ut * 32 + i;
      float sc = acc[r];
      const bool valid = jvalid && urow < A.n;
      if (valid) {
        if (A.prob) sc = sc + wd * A.prob[(size_t)urow * N + j];
        if (A.scores) A.scores[(size_t)urow * N + j] = sc;
        if (K > 0 && sc > T.thr[i]) {
          const int pos = atomicAdd(&T.cnt[i], 1);
          T.cs[i][pos] = sc; T.ci[i][pos] = j;
        }
      }
    }
    if (K > 0) {
      wave_fence();
      // lists that could overflow on the next tile (cnt > CAP - 32) are compacted now
      const int c = lane < 32 ? T.cnt[lane] : 0;
      unsigned long long need = __ballot(c > TK_CAP - 32);
      while (need) {
        const int i = __builtin_ctzll(need);
        need &= need - 1;
        compact_user(T, i, K);
      }
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
        A.cand_idx[o] = lane < n ? T.ci[i][lane] : INT_MAX;
      }
    }
  }
}

// Merge n_lists sorted K-lists per user into the final top-K (one wavefront per user).
__global__ __launch_bounds__(POI_BLOCK) void topk_merge_kernel(ScoreArgs A, int n_lists, int n_pad) {
  const int lane = lane_id();
  const int u = blockIdx.x * POI_NWAVE + wave_id();
  if (u >= A.n) return;
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
    // best K of up to 128 pending candidates: sort each half, then the 2K survivors
    float s0 = lane < cnt ? cs[w][lane] : -INFINITY; int i0 = lane < cnt ? ci[w][lane] : INT_MAX;
    float s1 = lane + 64 < cnt ? cs[w][lane + 64] : -INFINITY; int i1 = lane + 64 < cnt ? ci[w][lane + 64] : INT_MAX;
    wave_sort_desc(s0, i0);
    wave_sort_desc(s1, i1);
    // lanes [0,K) <- first list, lanes [K,2K) <- second list (2K <= 64)
    const float t1 = __shfl(s1, (lane - K) & 63, 64); const int j1 = __shfl(i1, (lane - K) & 63, 64);
    float s = lane < K ? s0 : (lane < 2 * K ? t1 : -INFINITY);
    int idx = lane < K ? i0 : (lane < 2 * K ? j1 : INT_MAX);
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
  hipLaunchKernelGGL((score_kernel<D8, DB>), grid, dim3(POI_BLOCK), 0, st, A);
  tm->end(st);
  return hipGetLastError();
}

hipError_t launch_score(const ScoreArgs& A, hipStream_t st, Timing* tm) {
  if (A.dim <= 32) return launch_score_t<4, true>(A, st, tm);
  if (A.dim <= 64) return launch_score_t<8, true>(A, st, tm);
  if (A.dim <= 128) return launch_score_t<16, true>(A, st, tm);
  if (A.dim <= 256) return launch_score_t<32, false>(A, st, tm);
  return hipErrorInvalidValue;
}

hipError_t launch_topk_merge(const ScoreArgs& A, int n_lists, hipStream_t st) {
  const int n_pad = ((A.n + 31) / 32) * 32;
  hipLaunchKernelGGL(topk_merge_kernel, dim3((A.n + POI_NWAVE - 1) / POI_NWAVE), dim3(POI_BLOCK), 0, st, A, n_lists, n_pad);
  return hipGetLastError();
}

hipError_t launch_topk_rows(const float* scores, int n, int n_item, int k, int* idx_out, float* score_out, hipStream_t st) {
  hipLaunchKernelGGL(topk_rows_kernel, dim3((n + POI_NWAVE - 1) / POI_NWAVE), dim3(POI_BLOCK), 0, st, scores, n, n_item, k, idx_out, score_out);
  return hipGetLastError();
}

}  // namespace poi
