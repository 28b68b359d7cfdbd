// Sorted segmented scatter for the tile engine: the sparse SGD write-back of
// public/GRU_Spatial.py:149-153,212-215 (T.set_subtensor on the unique rows of p U q and of the
// distance bins) for a whole launch, without float atomics and with a fixed summation order.
//
// Every table touch of the launch (POI ids of p and q, distance bins of dp, at every position of
// every sequence) is one slot: key = unified row id (POI rows, then distance-bin rows), value = slot
// index.  te_rowmap writes the slots; a stable LSD radix sort groups them by row; te_segment finds
// each row's [start, end) range and marks the first entry of every sequence in it (= the distinct-
// sequence count of the batch rule); te_reduce walks the rows: a row with <= 64 entries is summed
// and updated by one wavefront, longer ("hot") rows - popular POIs, every distance bin - are cut into
// 256-entry chunks summed by whole workgroups and combined in chunk order.  An entry names the packed
// step row whose dx (te_gemm_dx) and/or g*h (te_head) it adds, so the gradient tables of the
// per-sequence engine are not used at all and the result does not depend on scheduling.
#include "poi_common.h"
#include "poi_kernels.h"

namespace poi {

#define RS_BLOCK 256

__device__ __forceinline__ float4 f4_add(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }

// -------------------------------------------------------------------------------------------------
// stable LSD radix sort of (key, value) pairs, element count read from device memory
// -------------------------------------------------------------------------------------------------
__device__ __forceinline__ void rs_range(int N, int& b, int& e) {
  const int G = gridDim.x;
  const int per = ((((N + G - 1) / G) + RS_BLOCK - 1) / RS_BLOCK) * RS_BLOCK;
  b = min(N, (int)blockIdx.x * per);
  e = min(N, b + per);
}

__global__ __launch_bounds__(RS_BLOCK) void rs_hist_kernel(const int* __restrict__ keys, const int* __restrict__ n_ptr,
                                                           int shift, int nbin, int* __restrict__ hist) {
  __shared__ int cnt[RS_MAXBIN];
  int b, e;
  rs_range(*n_ptr, b, e);
  for (int d = threadIdx.x; d < nbin; d += RS_BLOCK) cnt[d] = 0;
  __syncthreads();
  // One LDS atomic per DISTINCT digit of a wave's 64 keys (the lanes holding a digit are matched with ballots, the first of them adds
  // their count): the keys are Zipf-popular POIs and three distance bins that hold 80 % of the steps - a per-key atomicAdd serialised
  // up to 64 ways on one LDS address (96 us per pass for 730 k keys, WAIT_ANY 92 %: profiles/r03_sq_counters.md).
  const unsigned long long below = (1ull << lane_id()) - 1ull;
  for (int t0 = b; t0 < e; t0 += RS_BLOCK) {
    const int i = t0 + threadIdx.x;
    const bool valid = i < e;
    const int d = valid ? (keys[i] >> shift) & (nbin - 1) : 0;
    unsigned long long m = __ballot(valid);
    for (int bit = 1; bit < nbin; bit <<= 1) {
      const unsigned long long bal = __ballot((d & bit) != 0);
      m &= (d & bit) ? bal : ~bal;
    }
    if (valid && (m & below) == 0ull) atomicAdd(&cnt[d], __builtin_popcountll(m));
  }
  __syncthreads();
  for (int d = threadIdx.x; d < nbin; d += RS_BLOCK) hist[d * gridDim.x + blockIdx.x] = cnt[d];
}

// block d: exclusive scan over the blocks' counts of digit d (in place) + the digit total
__global__ __launch_bounds__(RS_GRID) void rs_digit_scan_kernel(int* __restrict__ hist, int* __restrict__ total) {
  __shared__ int wsum[RS_GRID / 64];
  const int d = blockIdx.x, tid = threadIdx.x, lane = lane_id(), w = wave_id();
  const int v = hist[d * RS_GRID + tid];
  int inc = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(inc, o, 64); if (lane >= o) inc += t; }
  if (lane == 63) wsum[w] = inc;
  __syncthreads();
  int pre = 0;
#pragma unroll
  for (int i = 0; i < RS_GRID / 64; ++i) if (i < w) pre += wsum[i];
  hist[d * RS_GRID + tid] = pre + inc - v;
  if (tid == RS_GRID - 1) total[d] = pre + inc;
}

// vin == nullptr: values are the element indices (first pass)
__global__ __launch_bounds__(RS_BLOCK) void rs_scatter_kernel(const int* __restrict__ kin, const int* __restrict__ vin,
                                                              int* __restrict__ kout, int* __restrict__ vout,
                                                              const int* __restrict__ n_ptr, int shift, int nbin,
                                                              const int* __restrict__ hist, const int* __restrict__ total) {
  __shared__ int base[RS_MAXBIN];
  __shared__ int wcnt[RS_BLOCK / 64][RS_MAXBIN];
  __shared__ int wsum[RS_BLOCK / 64];
  int b, e;
  rs_range(*n_ptr, b, e);
  const int lane = lane_id(), w = wave_id();
  // base[d] = (keys with a smaller digit) + (keys with digit d in earlier blocks): exclusive scan of
  // the digit totals (two digits per thread) + this block's entry of the per-digit block scan
  {
    const int d0 = 2 * threadIdx.x, d1 = d0 + 1;
    const int t0 = d0 < nbin ? total[d0] : 0, t1 = d1 < nbin ? total[d1] : 0;
    int inc = t0 + t1;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(inc, o, 64); if (lane >= o) inc += t; }
    if (lane == 63) wsum[w] = inc;
    __syncthreads();
    int pre = 0;
#pragma unroll
    for (int i = 0; i < RS_BLOCK / 64; ++i) if (i < w) pre += wsum[i];
    const int ex = pre + inc - (t0 + t1);
    if (d0 < nbin) base[d0] = ex + hist[d0 * gridDim.x + blockIdx.x];
    if (d1 < nbin) base[d1] = ex + t0 + hist[d1 * gridDim.x + blockIdx.x];
  }
  for (int d = threadIdx.x; d < nbin; d += RS_BLOCK) {
#pragma unroll
    for (int ww = 0; ww < RS_BLOCK / 64; ++ww) wcnt[ww][d] = 0;
  }
  __syncthreads();
  const unsigned long long below = (1ull << lane) - 1ull;
  for (int t0 = b; t0 < e; t0 += RS_BLOCK) {
    const int i = t0 + threadIdx.x;
    const bool valid = i < e;
    const int key = valid ? kin[i] : 0;
    const int val = valid ? (vin ? vin[i] : i) : 0;
    const int d = (key >> shift) & (nbin - 1);
    // lanes of this wave holding the same digit (stable rank = number of such lanes below this one)
    unsigned long long m = __ballot(valid);
    for (int bit = 1; bit < nbin; bit <<= 1) {
      const unsigned long long bal = __ballot((d & bit) != 0);
      m &= (d & bit) ? bal : ~bal;
    }
    const int rank = __builtin_popcountll(m & below), tot = __builtin_popcountll(m);
    if (valid && rank == 0) wcnt[w][d] = tot;
    __syncthreads();
    if (valid) {
      int o = base[d] + rank;
      for (int w2 = 0; w2 < w; ++w2) o += wcnt[w2][d];
      kout[o] = key; vout[o] = val;
    }
    __syncthreads();
    if (valid && rank == 0) { atomicAdd(&base[d], tot); wcnt[w][d] = 0; }
    __syncthreads();
  }
}

// -------------------------------------------------------------------------------------------------
// te_segment: sorted (key, slot) -> entry codes + per-row ranges
// -------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void te_segment_kernel(TeArgs A, const int* __restrict__ Ks, const int* __restrict__ Vs) {
  const int N = A.cnt[0], R = A.n_item + 1 + A.n_dist + 1;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < N; i += gridDim.x * 256) {
    const int k = Ks[i];
    if (k >= R) continue;                  // sentinel slots (sorted last)
    const int v = Vs[i];
    const bool same = i > 0 && Ks[i - 1] == k;
    const bool first = !same || A.slot_seq[Vs[i - 1]] != A.slot_seq[v];
    A.ent[i] = A.code[v] | (first ? (int)TE_ENT_FIRST : 0);
    if (!same) {
      A.seg_start[k] = i;
      // touched-row list for tables much larger than a launch's footprint (te_reduce then walks the touched rows instead of
      // scanning every table row; the order of the list is irrelevant - each row is reduced on its own)
      if (A.urow) A.urow[atomicAdd(&A.cnt[3], 1)] = k;
    }
    if (i == N - 1 || Ks[i + 1] != k) A.seg_end[k] = i + 1;
  }
}

// (round 6) The hot rows of the write-back - more than TE_COLD_MAX entries - and their 64-entry chunks, listed right behind te_segment (te_reduce used to build
// the list with atomics while it walked the rows: the chunk sums could not start before it).  One wave per 64 rows; a hot row's chunk list is written by the
// whole wave.  The order of the list depends on scheduling; every row is reduced on its own, in chunk order: the result does not.
__global__ __launch_bounds__(256) void te_hotlist_kernel(TeArgs A) {
  const int RT = A.bintab ? A.n_item + 1 : A.n_item + 1 + A.n_dist + 1;      // bintab: the distance-bin rows are written by te_dapply
  const int R = A.urow ? A.cnt[3] : RT;
  const int lane = lane_id();
  for (int i0 = (blockIdx.x * 4 + wave_id()) * 64; i0 < R; i0 += gridDim.x * 256) {
    const int i = i0 + lane;
    int row = -1, start = 0, cnt = 0;
    if (i < R) {
      row = A.urow ? A.urow[i] : i;
      if (row >= 0 && row < RT) { const int e = A.seg_end[row]; if (e) { start = A.seg_start[row]; cnt = e - start; } }
    }
    unsigned long long m = __ballot(cnt > TE_COLD_MAX);
    while (m) {
      const int src = __builtin_ctzll(m); m &= m - 1;
      const int r = __shfl(row, src, 64), st = __shfl(start, src, 64), c = __shfl(cnt, src, 64);
      const int nch = (c + TE_HOT_CHUNK - 1) / TE_HOT_CHUNK;
      int h = 0, c0 = 0;
      if (lane == 0) { h = atomicAdd(&A.cnt[1], 1); c0 = atomicAdd(&A.cnt[2], nch); A.hot_rows[h] = make_int4(r, st, c, c0); }
      h = __shfl(h, 0, 64); c0 = __shfl(c0, 0, 64);
      for (int k = lane; k < nch; k += 64) A.hot_chunks[c0 + k] = make_int2(h, k);
    }
  }
}

// -------------------------------------------------------------------------------------------------
// per-POI regrouping (TeArgs.ppoi): S[r] = sum of DA over the steps whose input POI is row r.
//   te_pcount / te_passign  one exclusive scan over the sorted slots (per-block counts, then block prefix + ballot ranks inside
//                           256-wide strips) ranks the POI rows that are step inputs (pmark, set by te_slots) in sorted-key order
//                           -> S row index (pmark[row] <- index + 1, urow_p), AND compacts their dx entries into one list
//                           (dxe = packed step row, dxs = S row, dstart = first list position of every S row).  The order of the S
//                           rows - the summation order of te_wgrad's d ui jobs - does not depend on scheduling.
//   te_psum                 streams the list like te_dsum streams the bins: one workgroup per 64 consecutive entries, thread =
//                           column of DA, 16 rows in flight; a run of equal S rows inside the range is summed in order and flushed -
//                           complete rows straight to S, the run that opens the range to pfirst[range], the one that closes it to
//                           plast[range] (a hot POI spans many ranges)
//   te_pfin                 stitches the rows that cross range boundaries: head fragment + whole ranges + tail fragment, in order
// -------------------------------------------------------------------------------------------------
#define TE_PBLK 1024
__device__ __forceinline__ bool te_pflag(const TeArgs& A, int i, int N) {
  if (i >= N) return false;
  const int k = A.ks[i];
  return k <= A.n_item && (i == 0 || A.ks[i - 1] != k) && A.pmark[k] != 0;
}
__device__ __forceinline__ bool te_dxflag(const TeArgs& A, int i, int N) {
  return i < N && A.ks[i] <= A.n_item && (A.ent[i] & TE_ENT_DX) != 0;
}
__device__ __forceinline__ int te_pper(int N) { return (((N + TE_PBLK - 1) / TE_PBLK) + 255) & ~255; }
__global__ __launch_bounds__(256) void te_pcount_kernel(TeArgs A) {
  __shared__ int red[2][4];
  const int N = A.cnt[0], per = te_pper(N);
  const int b0 = blockIdx.x * per;
  int c = 0, d = 0;
  for (int i = b0 + threadIdx.x; i < min(N, b0 + per); i += 256) { c += te_pflag(A, i, N) ? 1 : 0; d += te_dxflag(A, i, N) ? 1 : 0; }
  for (int o = 32; o > 0; o >>= 1) { c += __shfl_xor(c, o, 64); d += __shfl_xor(d, o, 64); }
  if (lane_id() == 0) { red[0][wave_id()] = c; red[1][wave_id()] = d; }
  __syncthreads();
  if (threadIdx.x == 0) {
    A.pblk[blockIdx.x] = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
    A.pblk[TE_PBLK + blockIdx.x] = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
  }
}
__global__ __launch_bounds__(256) void te_passign_kernel(TeArgs A) {
  __shared__ int s_w[2][4];
  __shared__ int s_base[2];
  const int N = A.cnt[0], per = te_pper(N), tid = threadIdx.x, lane = lane_id(), w = wave_id();
  const int b0 = blockIdx.x * per;
  {
    int v = 0, v2 = 0;
    for (int j = tid; j < (int)blockIdx.x; j += 256) { v += A.pblk[j]; v2 += A.pblk[TE_PBLK + j]; }
    for (int o = 32; o > 0; o >>= 1) { v += __shfl_xor(v, o, 64); v2 += __shfl_xor(v2, o, 64); }
    if (lane == 0) { s_w[0][w] = v; s_w[1][w] = v2; }
    __syncthreads();
    if (tid == 0) { s_base[0] = (s_w[0][0] + s_w[0][1]) + (s_w[0][2] + s_w[0][3]); s_base[1] = (s_w[1][0] + s_w[1][1]) + (s_w[1][2] + s_w[1][3]); }
    __syncthreads();
  }
  int rbase = s_base[0], dbase = s_base[1];
  const unsigned long long below = (1ull << lane) - 1ull;
  for (int i0 = b0; i0 < min(N, b0 + per); i0 += 256) {
    const int i = i0 + tid;
    const bool f = te_pflag(A, i, N), dx = te_dxflag(A, i, N);
    const unsigned long long m = __ballot(f), md = __ballot(dx);
    __syncthreads();
    if (lane == 0) { s_w[0][w] = __builtin_popcountll(m); s_w[1][w] = __builtin_popcountll(md); }
    __syncthreads();
    int rb = 0, db = 0;
    for (int j = 0; j < w; ++j) { rb += s_w[0][j]; db += s_w[1][j]; }
    const int rinc = rbase + rb + __builtin_popcountll(m & below) + (f ? 1 : 0);     // row flags up to and including this position
    const int dpos = dbase + db + __builtin_popcountll(md & below);                  // dx entries before this position
    if (f) {
      const int idx = rinc - 1, row = A.ks[i];
      A.pmark[row] = idx + 1; A.urow_p[idx] = row; A.dstart[idx] = dpos;
    }
    if (dx) { A.dxe[dpos] = A.ent[i] & TE_ENT_ROW; A.dxs[dpos] = rinc - 1; }           // (a dx entry's row is flagged at or before it)
    rbase += (s_w[0][0] + s_w[0][1]) + (s_w[0][2] + s_w[0][3]);
    dbase += (s_w[1][0] + s_w[1][1]) + (s_w[1][2] + s_w[1][3]);
  }
  if (blockIdx.x == gridDim.x - 1 && tid == 0) { A.cnt[4] = rbase; A.cnt[5] = dbase; A.dstart[rbase] = dbase; }
}

template <int D>
__global__ __launch_bounds__(3 * D) void te_psum_kernel(TeArgs A) {
  const int Ndx = A.cnt[5], col = threadIdx.x, lane = lane_id();
  const int nr = (Ndx + 63) / 64;
  // hot bins (TeArgs.dhot): this workgroup's sums of the DA rows whose step-input bin is one of them - the rows are in registers here anyway
  const bool hot_on = A.dhot_on != 0;
  int hb[TE_HB];
  float hacc[TE_HB];
#pragma unroll
  for (int k = 0; k < TE_HB; ++k) { hb[k] = hot_on ? A.dhot[1 + k] : -1; hacc[k] = 0.f; }
  for (int r = blockIdx.x; r < nr; r += gridDim.x) {
    const int j0 = 64 * r, j1 = min(Ndx, j0 + 64);
    // the range's entries, one per lane (every wave holds the same lists), handed out through v_readlane: scalar row
    // addresses, scalar run detection, and all 64 row loads in flight behind one dependent index load
    const int me = A.dxe[min(j0 + lane, j1 - 1)], ms = A.dxs[min(j0 + lane, j1 - 1)];
    const int mb = hot_on ? A.row_dp[me] : -2;
    float v[64];
#pragma unroll
    for (int u = 0; u < 64; ++u) v[u] = A.G[(size_t)__builtin_amdgcn_readlane(me, u) * 3 * D + col];
    int cur = __builtin_amdgcn_readlane(ms, 0);
    bool first = true;
    float acc = 0.f;
#pragma unroll
    for (int u = 0; u < 64; ++u) {
      const int sr = __builtin_amdgcn_readlane(ms, u);     // (clamped past j1: the last entry's S row - no flush; its value is masked)
      if (sr != cur) {                                      // (block-uniform control flow: every thread sees the same entries)
        if (first) A.pfirst[(size_t)r * 3 * D + col] = acc; else A.S[(size_t)cur * 3 * D + col] = acc;
        first = false; cur = sr; acc = 0.f;
      }
      const float vu = j0 + u < j1 ? v[u] : 0.f;
      acc += vu;
      if (hot_on) {                                         // (scalar compares: the bin of entry u is the same for every thread)
        const int bu = __builtin_amdgcn_readlane(mb, u);
#pragma unroll
        for (int k = 0; k < TE_HB; ++k) if (bu == hb[k]) hacc[k] += vu;
      }
    }
    if (first) A.pfirst[(size_t)r * 3 * D + col] = acc; else A.plast[(size_t)r * 3 * D + col] = acc;
  }
  if (hot_on) {
#pragma unroll
    for (int k = 0; k < TE_HB; ++k)
      if (hb[k] >= 0) A.dpart[(size_t)(A.dch0[hb[k]] + (int)blockIdx.x) * 3 * D + col] = hacc[k];
  }
}

// rows whose dx entries are not strictly inside one range: first / last run of a range, or spanning several ranges
template <int D>
__global__ __launch_bounds__(3 * D) void te_pfin_kernel(TeArgs A) {
  const int P = A.cnt[4], Ndx = A.cnt[5], col = threadIdx.x;
  for (int idx = blockIdx.x; idx < P; idx += gridDim.x) {
    const int d0 = A.dstart[idx], d1 = A.dstart[idx + 1];
    const int ra = d0 >> 6, rb = (d1 - 1) >> 6;
    const bool first_a = (d0 & 63) == 0, closes_b = d1 == min(Ndx, 64 * (rb + 1));
    if (ra == rb) {
      if (first_a) A.S[(size_t)idx * 3 * D + col] = A.pfirst[(size_t)ra * 3 * D + col];
      else if (closes_b) A.S[(size_t)idx * 3 * D + col] = A.plast[(size_t)ra * 3 * D + col];
      continue;                                             // (strictly inside: te_psum wrote S)
    }
    float s = first_a ? A.pfirst[(size_t)ra * 3 * D + col] : A.plast[(size_t)ra * 3 * D + col];
    for (int r0 = ra + 1; r0 <= rb; r0 += 16) {           // whole ranges and the tail fragment: all "first run of their range"
      float v[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) v[u] = A.pfirst[(size_t)min(r0 + u, rb) * 3 * D + col];
#pragma unroll
      for (int u = 0; u < 16; ++u) s += r0 + u <= rb ? v[u] : 0.f;
    }
    A.S[(size_t)idx * 3 * D + col] = s;
  }
}

// S-row assignment (needs the sorted slots).  With a side stream it follows the slot sort there, next to te_rec_fwd (one workgroup per CU and no
// room for a second: two small LDS-free kernels fit beside it; next to te_head's persistent grid they cost it a workgroup slot, +30 us) -
// 22 us off the main stream; without one, right before te_psum.
hipError_t launch_te_passign(TeArgs& A, hipStream_t st) {
  hipLaunchKernelGGL(te_pcount_kernel, dim3(TE_PBLK), dim3(256), 0, st, A);
  hipLaunchKernelGGL(te_passign_kernel, dim3(TE_PBLK), dim3(256), 0, st, A);
  return hipGetLastError();
}
hipError_t launch_te_psum(TeArgs& A, int num_cu, hipStream_t st) {
  if (!A.side || (A.dbg & 1024)) { hipError_t e = launch_te_passign(A, st); if (e != hipSuccess) return e; }      // (POI_TE_DBG bit 1024: on the main stream, for A/B runs)
  if (A.dim == 128) {
    hipLaunchKernelGGL(te_psum_kernel<128>, dim3(A.npw), dim3(384), 0, st, A);
    hipLaunchKernelGGL(te_pfin_kernel<128>, dim3(num_cu * 16), dim3(384), 0, st, A);
  } else if (A.dim == 256) {
    hipLaunchKernelGGL(te_psum_kernel<256>, dim3(A.npw), dim3(768), 0, st, A);
    hipLaunchKernelGGL(te_pfin_kernel<256>, dim3(num_cu * 16), dim3(768), 0, st, A);
  } else return hipErrorInvalidValue;
  return hipGetLastError();
}

hipError_t launch_radix_sort(int* keys0, int* keys1, int* vals0, int* vals1, const int* n_ptr, int bits, int* hist, hipStream_t st,
                             const int** ks, const int** vs) {
  const int npass = bits <= 9 ? 1 : bits <= 18 ? 2 : bits <= 27 ? 3 : 4;
  const int w = (bits + npass - 1) / npass, nbin = 1 << w;
  const int *kin = keys0, *vin = nullptr;
  int *kout = keys1, *vout = vals1;
  for (int p = 0; p < npass; ++p) {
    hipLaunchKernelGGL(rs_hist_kernel, dim3(RS_GRID), dim3(RS_BLOCK), 0, st, kin, n_ptr, p * w, nbin, hist);
    hipLaunchKernelGGL(rs_digit_scan_kernel, dim3(nbin), dim3(RS_GRID), 0, st, hist, hist + RS_HIST_INTS);
    hipLaunchKernelGGL(rs_scatter_kernel, dim3(RS_GRID), dim3(RS_BLOCK), 0, st, kin, vin, kout, vout, n_ptr, p * w, nbin, hist, hist + RS_HIST_INTS);
    kin = kout; vin = vout;
    if (kout == keys1) { kout = keys0; vout = vals0; } else { kout = keys1; vout = vals1; }
  }
  *ks = kin; *vs = vin;
  return hipGetLastError();
}

hipError_t launch_te_sort(TeArgs& A, hipStream_t st) {
  const int bits = A.key_bits;
  const int npass = bits <= 9 ? 1 : bits <= 18 ? 2 : bits <= 27 ? 3 : 4;
  const int w = (bits + npass - 1) / npass, nbin = 1 << w;
  const int *kin = A.keys0, *vin = nullptr;
  int *kout = A.keys1, *vout = A.vals1;
  for (int p = 0; p < npass; ++p) {
    hipLaunchKernelGGL(rs_hist_kernel, dim3(RS_GRID), dim3(RS_BLOCK), 0, st, kin, A.cnt, p * w, nbin, A.hist);
    hipLaunchKernelGGL(rs_digit_scan_kernel, dim3(nbin), dim3(RS_GRID), 0, st, A.hist, A.hist + RS_HIST_INTS);
    hipLaunchKernelGGL(rs_scatter_kernel, dim3(RS_GRID), dim3(RS_BLOCK), 0, st, kin, vin, kout, vout, A.cnt, p * w, nbin, A.hist, A.hist + RS_HIST_INTS);
    kin = kout; vin = vout;
    if (kout == A.keys1) { kout = A.keys0; vout = A.vals0; } else { kout = A.keys1; vout = A.vals1; }
  }
  hipLaunchKernelGGL(te_segment_kernel, dim3(1024), dim3(256), 0, st, A, kin, vin);
  hipLaunchKernelGGL(te_hotlist_kernel, dim3(256), dim3(256), 0, st, A);
  A.ks = kin;
  return hipGetLastError();
}

// -------------------------------------------------------------------------------------------------
// reduction of a row's entries
// -------------------------------------------------------------------------------------------------
template <int D>
__device__ __forceinline__ float4 ent_contrib(const TeArgs& A, int e, int doff, int c) {
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  const size_t rr = (size_t)(e & TE_ENT_ROW);
  // (bintab: a distance-bin row's dx sum comes from the per-bin sums of DA, see te_dsum below - no di half of X exists)
  if ((e & TE_ENT_DX) && !(A.bintab && doff) && !A.ppoi) v = *reinterpret_cast<const float4*>(A.X + rr * A.xw + doff + c);
  if (e & TE_ENT_GH) {
    float g = A.gcoef[rr - 1];
    if (e & TE_ENT_NEG) g = -g;
    const float4 h = *reinterpret_cast<const float4*>(A.H + (rr - 1) * D + c);
    v.x = fmaf(g, h.x, v.x); v.y = fmaf(g, h.y, v.y); v.z = fmaf(g, h.z, v.z); v.w = fmaf(g, h.w, v.w);
  }
  return v;
}

// One wavefront sums entries ent[s, s + cnt), cnt <= 64, in a fixed order: D/4 lanes per entry
// (float4 each), 64/(D/4) entries per pass, eight passes in flight.  The sum ends up in every lane
// group (all groups hold the same total); *nfirst = number of TE_ENT_FIRST entries.
template <int D>
__device__ __forceinline__ float4 seg_sum(const TeArgs& A, int s, int cnt, int doff, int* nfirst) {
  constexpr int LPR = D / 4, EPW = 64 / LPR;
  const int lane = lane_id(), grp = lane / LPR, c = (lane % LPR) * 4;
  const int mine = lane < cnt ? A.ent[s + lane] : 0;
  *nfirst = __builtin_popcountll(__ballot(mine < 0));
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int i0 = 0; i0 < cnt; i0 += 8 * EPW) {
    int e[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int idx = i0 + u * EPW + grp;
      const int got = __shfl(mine, idx & 63, 64);
      e[u] = idx < cnt ? got : 0;
    }
    float4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = ent_contrib<D>(A, e[u], doff, c);
    acc = f4_add(acc, f4_add(f4_add(f4_add(v[0], v[1]), f4_add(v[2], v[3])), f4_add(f4_add(v[4], v[5]), f4_add(v[6], v[7]))));
  }
#pragma unroll
  for (int o = LPR; o < 64; o <<= 1) {
    acc.x += __shfl_xor(acc.x, o, 64); acc.y += __shfl_xor(acc.y, o, 64);
    acc.z += __shfl_xor(acc.z, o, 64); acc.w += __shfl_xor(acc.w, o, 64);
  }
  return acc;
}

// row <- row - alpha * min(nseq, cap) / nseq * (G + lambda * mult * row)      (batch rule, include/poi_hip.h)
template <int D>
__device__ __forceinline__ void apply_sum(float* __restrict__ trow, int f16, float4 g, int mult, int nseq, float alpha, float lambda, float cap,
                                          unsigned salt = 0, size_t erow = 0) {
  constexpr int LPR = D / 4;
  const int lane = lane_id();
  if (lane >= LPR) return;
  float sc, lm; rule_scales(alpha, lambda, nseq, mult, cap, sc, lm);
  float4 tv = ld4t(trow, (size_t)lane * 4, f16);
  tv.x -= sc * (g.x + lm * tv.x); tv.y -= sc * (g.y + lm * tv.y);
  tv.z -= sc * (g.z + lm * tv.z); tv.w -= sc * (g.w + lm * tv.w);
  st4t_sr(trow, (size_t)lane * 4, f16, tv, salt, erow + (size_t)lane * 4);
}

struct RowInfo { float* trow; int* pm; int* pn; int doff; int f16; };      // f16: trow addresses IEEE half elements
__device__ __forceinline__ RowInfo row_info(const TeArgs& A, int row) {
  RowInfo r;
  const int D = A.dim;
  if (row <= A.n_item) {
    r.f16 = A.lt_f16;
    r.trow = A.lt_f16 ? reinterpret_cast<float*>(reinterpret_cast<__half*>(A.lt) + (size_t)row * D) : A.lt + (size_t)row * D; r.doff = 0;
    r.pm = row == A.n_item ? A.mult_lt + A.n_item : nullptr;
    r.pn = row == A.n_item ? A.nseq_lt + A.n_item : nullptr;
  } else {
    const int b = row - A.n_item - 1;
    r.f16 = 0;
    r.trow = A.di + (size_t)b * D; r.doff = D;
    r.pm = b == A.n_dist ? A.mult_di + A.n_dist : nullptr;
    r.pn = b == A.n_dist ? A.nseq_di + A.n_dist : nullptr;
  }
  return r;
}

// Cold rows (<= TE_COLD_MAX entries; the bulk: ~5 entries per touched POI row).  The kernel is bound by the number of
// loads in flight, not by bytes: a row's update is a chain seg_end -> seg_start -> entry codes -> dx / h rows -> table row.
// So a wavefront works on 64 / (D/4) rows at once (D/4 lanes x float4 = one row), takes a row's entries EIGHT at a time, and
// every load of a batch is issued unconditionally before the first use: an entry without a dx (or g*h) term reads a
// resident all-zero row instead of branching around the load (a branch would make the waitcnt pass drain the queue).
// Entries are added in batch order ((e0+e1)+(e2+e3))+((e4+e5)+(e6+e7)), batches in order: reproducible.
template <int D, bool PPOI>       // PPOI: per-POI regrouping - a POI row's dx sum is ONE row of X (S . ui), no per-entry dx rows exist
__global__ __launch_bounds__(256) void te_reduce_kernel(TeArgs A, float alpha, float lambda) {
  constexpr int LPR = D / 4, RPW = 64 / LPR;
  const int RT = A.bintab ? A.n_item + 1 : A.n_item + 1 + A.n_dist + 1;     // bintab: the distance-bin rows are written by te_dapply
  // scan mode: every table row; list mode (A.urow): the launch's touched rows + the padding rows (which may be touched only
  // analytically, i.e. have no segment) - a padding row that does have a segment is in the list already and skipped at the tail
  const int n_u = A.urow ? A.cnt[3] : 0, n_pad = A.urow ? (A.bintab || A.n_dist < 0 ? 1 : 2) : 0;
  const int R = A.urow ? n_u + n_pad : RT;
  const int lane = lane_id(), sub = lane / LPR, c = (lane % LPR) * 4, lead = sub * LPR;
  const float* __restrict__ zrow = A.zrow;
  const int nw = gridDim.x * 4;
  // (round 5) a row's update is a chain of dependent loads - segment bounds -> entry codes -> h rows -> table row - and a wave walks its rows one
  // after the other: the bounds of the NEXT row pair are requested at the top of the current one (scan mode; the list mode's rows come through
  // another indirection and keep the plain order): te_scatter 131 -> 124 us per 12500-user launch.  (Also the first eight entry codes of the
  // next pair, bounds two pairs ahead: 140 us - 26 registers more and loads for rows without entries.)
  const int first0 = (blockIdx.x * 4 + wave_id()) * RPW;
  // (round 6) ... and so is the row's S-row mark, and the loads that depend on the row alone - its table values, its dx-sum row of X - are issued at
  // the top, next to the entry codes: a row pair's update was FIVE dependent round trips (codes -> h rows -> mark -> X row -> table row), now two.  Measured: 77 -> 78 us
  // per 12500-user launch - nothing: the kernel moves its 357 MB at ~80 % of what the memory system gives random 512-byte rows (tools/micro/gather_rate.hip), it does not wait on the chain
  int p_end = 0, p_start = 0, p_pmk = 0;
  if (!A.urow && first0 + sub < R) { p_end = A.seg_end[first0 + sub]; p_start = A.seg_start[first0 + sub]; if (PPOI && first0 + sub <= A.n_item) p_pmk = A.pmark[first0 + sub]; }
  for (int row0 = first0; row0 < R; row0 += nw * RPW) {
    const int idx = min(row0 + sub, R - 1);
    bool in = row0 + sub < R;
    int row = idx;
    if (A.urow) {
      if (idx < n_u) { row = A.urow[idx]; in = in && row < RT; row = min(row, RT - 1); }
      else { row = idx == n_u ? A.n_item : A.n_item + 1 + A.n_dist; in = in && A.seg_end[row] == 0; }
    }
    const int end = A.urow ? (in ? A.seg_end[row] : 0) : (in ? p_end : 0);
    const int start_pre = p_start;
    // per-POI regrouping: the summed dx of the row's step inputs = S[row] . ui, row pmark[row] - 1 of X (te_gemm_dx over S)
    const int pmk = !PPOI ? 0 : A.urow ? ((in && row <= A.n_item) ? A.pmark[row] : 0) : (in ? p_pmk : 0);          // S row + 1 (te_passign), 0: not a step input
    if (!A.urow) {
      const int nx = row0 + nw * RPW + sub;
      const int nxc = min(nx, R - 1);
      p_end = A.seg_end[nxc]; p_start = A.seg_start[nxc]; p_pmk = (PPOI && nxc <= A.n_item) ? A.pmark[nxc] : 0;
      if (nx >= R) { p_end = 0; p_pmk = 0; }
    }
    const RowInfo ri = row_info(A, row);
    float4 tv = ld4t(ri.trow, (size_t)c, ri.f16);
    const float4 xs = *reinterpret_cast<const float4*>(pmk ? A.X + (size_t)(pmk - 1) * A.xw + c : zrow + c);
    // padding rows: analytic multiplicity / sequence count from te_rowmap
    const int am = (in && ri.pm) ? *ri.pm : 0, an = (in && ri.pn) ? *ri.pn : 0;
    const int start = end ? (A.urow ? A.seg_start[row] : start_pre) : 0, cnt = end - start;
    const bool hot = cnt > TE_COLD_MAX;
    const int n_e = hot ? 0 : cnt;       // (hot rows: te_hotlist listed them behind te_segment; te_hot_reduce / te_hot_apply take them)
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    int nf = 0;
    for (int i0 = 0; i0 < n_e; i0 += 8) {
      int e[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) { const int v = A.ent[start + min(i0 + u, n_e - 1)]; e[u] = i0 + u < n_e ? v : 0; }
      float4 x[8], hh[8]; float g[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const size_t rr = (size_t)(e[u] & TE_ENT_ROW);
        const bool dx = (e[u] & TE_ENT_DX) != 0, gh = (e[u] & TE_ENT_GH) != 0;
        const float* px = dx ? A.X + rr * A.xw + ri.doff + c : zrow + c;
        const float* ph = gh ? A.H + (rr - 1) * D + c : zrow + c;
        const float* pg = gh ? A.gcoef + (rr - 1) : zrow;
        if (PPOI) x[u] = make_float4(0.f, 0.f, 0.f, 0.f); else x[u] = *reinterpret_cast<const float4*>(px);
        hh[u] = *reinterpret_cast<const float4*>(ph);
        g[u] = *pg;
      }
      float4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const float gg = (e[u] & TE_ENT_NEG) ? -g[u] : g[u];
        v[u] = make_float4(fmaf(gg, hh[u].x, x[u].x), fmaf(gg, hh[u].y, x[u].y), fmaf(gg, hh[u].z, x[u].z), fmaf(gg, hh[u].w, x[u].w));
        nf += e[u] < 0 ? 1 : 0;
      }
      acc = f4_add(acc, f4_add(f4_add(f4_add(v[0], v[1]), f4_add(v[2], v[3])), f4_add(f4_add(v[4], v[5]), f4_add(v[6], v[7]))));
    }
    if (pmk && !hot) {
      acc = f4_add(acc, xs);
      if (lane == lead) A.pmark[row] = 0;
    }
    if (in && !hot && (end != 0 || an != 0)) {
      const int nseq = ri.pn ? an : nf, mult = cnt + am;
      float sc, lm; rule_scales(alpha, lambda, nseq, mult, A.bcap, sc, lm);
      tv.x -= sc * (acc.x + lm * tv.x); tv.y -= sc * (acc.y + lm * tv.y);
      tv.z -= sc * (acc.z + lm * tv.z); tv.w -= sc * (acc.w + lm * tv.w);
      st4t_sr(ri.trow, (size_t)c, ri.f16, tv, A.sr_salt, (size_t)row * D + c);
      if (lane == lead) { A.seg_end[row] = 0; if (ri.pm) { *ri.pm = 0; *ri.pn = 0; } }
    }
  }
}

// (round 6) one WAVE per 64-entry chunk (a workgroup took 256 entries, its four waves 64 each, and combined them through LDS): four times the
// jobs for the same rows - the kernel is a handful of hot rows deep and was ~500 workgroups on 256 CUs.
template <int D>
__global__ __launch_bounds__(256) void te_hot_reduce_kernel(TeArgs A) {
  static_assert(TE_HOT_CHUNK == 64, "te_hot_reduce: a chunk is what seg_sum takes in one call");
  constexpr int LPR = D / 4;
  const int nchunk = A.cnt[2];
  const int lane = lane_id();
  for (int ci = blockIdx.x * 4 + wave_id(); ci < nchunk; ci += gridDim.x * 4) {
    const int2 item = A.hot_chunks[ci];
    const int4 hr = A.hot_rows[item.x];
    const int doff = hr.x <= A.n_item ? 0 : D;
    const int s = hr.y + item.y * TE_HOT_CHUNK;
    const int wc = min(TE_HOT_CHUNK, hr.z - item.y * TE_HOT_CHUNK);
    int nf = 0;
    const float4 g = seg_sum<D>(A, s, wc, doff, &nf);
    if (lane < LPR) *reinterpret_cast<float4*>(A.hot_part + (size_t)ci * D + lane * 4) = g;
    if (lane == 0) A.hot_nf[ci] = nf;
  }
}

// (round 6) one WORKGROUP per hot row: its four waves sum a quarter of the row's chunk partials each (a fixed split, 16 partial rows in flight
// per wave), LDS combine in wave order - one wave walked all of them, and the hottest POI of a 12500-user launch has hundreds of chunks.
template <int D>
__global__ __launch_bounds__(256) void te_hot_apply_kernel(TeArgs A, float alpha, float lambda) {
  constexpr int LPR = D / 4, EPW = 64 / LPR;
  __shared__ __align__(16) float part[4][D];
  __shared__ int s_nf[4];
  const int nhot = A.cnt[1];
  const int lane = lane_id(), grp = lane / LPR, c = (lane % LPR) * 4, w = wave_id();
  for (int h = blockIdx.x; h < nhot; h += gridDim.x) {
    const int4 hr = A.hot_rows[h];
    const int row = hr.x, cnt = hr.z, c0 = hr.w, nch = (cnt + TE_HOT_CHUNK - 1) / TE_HOT_CHUNK;
    const int per = (nch + 3) / 4, k0 = w * per, k1 = min(nch, k0 + per);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int i0 = k0; i0 < k1; i0 += 8 * EPW) {
      float4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int idx = i0 + u * EPW + grp;
        v[u] = idx < k1 ? *reinterpret_cast<const float4*>(A.hot_part + (size_t)(c0 + idx) * D + c) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
      acc = f4_add(acc, f4_add(f4_add(f4_add(v[0], v[1]), f4_add(v[2], v[3])), f4_add(f4_add(v[4], v[5]), f4_add(v[6], v[7]))));
    }
#pragma unroll
    for (int o = LPR; o < 64; o <<= 1) {
      acc.x += __shfl_xor(acc.x, o, 64); acc.y += __shfl_xor(acc.y, o, 64);
      acc.z += __shfl_xor(acc.z, o, 64); acc.w += __shfl_xor(acc.w, o, 64);
    }
    int nf = 0;
    for (int i = k0 + lane; i < k1; i += 64) nf += A.hot_nf[c0 + i];
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) nf += __shfl_xor(nf, o, 64);
    __syncthreads();                                        // (the previous row's partials have been read)
    if (lane < LPR) *reinterpret_cast<float4*>(&part[w][lane * 4]) = acc;
    if (lane == 0) s_nf[w] = nf;
    __syncthreads();
    if (w == 0) {
      const RowInfo ri = row_info(A, row);
      const int am = ri.pm ? *ri.pm : 0, an = ri.pn ? *ri.pn : 0;
      float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
      if (lane < LPR) {
        const float4 p0 = *reinterpret_cast<const float4*>(&part[0][lane * 4]), p1 = *reinterpret_cast<const float4*>(&part[1][lane * 4]);
        const float4 p2 = *reinterpret_cast<const float4*>(&part[2][lane * 4]), p3 = *reinterpret_cast<const float4*>(&part[3][lane * 4]);
        g = f4_add(f4_add(p0, p1), f4_add(p2, p3));
      }
      const int nft = (s_nf[0] + s_nf[1]) + (s_nf[2] + s_nf[3]);
      const int pmk = (A.ppoi && row <= A.n_item) ? A.pmark[row] : 0;
      if (pmk) {
        if (lane < LPR) g = f4_add(g, *reinterpret_cast<const float4*>(A.X + (size_t)(pmk - 1) * A.xw + lane * 4));
        if (lane == 0) A.pmark[row] = 0;
      }
      apply_sum<D>(ri.trow, ri.f16, g, cnt + am, ri.pn ? an : nft, alpha, lambda, A.bcap, A.sr_salt, (size_t)row * D);
      if (lane == 0) { A.seg_end[row] = 0; if (ri.pm) { *ri.pm = 0; *ri.pn = 0; } }
    }
  }
}

// -------------------------------------------------------------------------------------------------
// Per-bin sums of DA (bintab, see te_ztab_kernel in tile_engine.hip).  The distance-bin half of the step input
// takes n_dist + 1 values, so everything the backward pass needs from it is linear in
//   S[b] = sum over the packed rows t with dp_t = b of DA_t                      ((n_dist + 1) x 3D):
//   d di[b]        (the row's dx sum of the batch rule)  = S[b] . ui[:, D:2D]    -> dgd, used by te_reduce / te_hot_apply
//   d ui[:, D:2D]  (dense gradient)                      = S^T . di              -> slab 0
// which replaces the di halves of te_gemm_dx and of te_wgrad's d ui jobs (a quarter of the step's flops).
// The rows of a bin are the DX entries of its segment in the sorted entry list, cut into 64-entry chunks (one
// workgroup iteration each; the bins are very uneven), whose partial sums are added in chunk order: reproducible.
// -------------------------------------------------------------------------------------------------
// chunk tables: dch0[b] = first 64-entry chunk of bin b (exclusive scan of the bins' chunk counts), dch0[NB] = total;
// dch1[b] = first SUPER-chunk (TE_DSUPER consecutive chunks of one bin) of bin b, dch1[NB] = total.  The bins are very
// uneven on real check-in data (most hops are short: a few bins hold almost every step), so the per-bin sum is a
// three-level tree - 64-entry chunks (te_dsum), TE_DSUPER-chunk groups (te_dred), groups of a bin (te_dfin) - every level
// added in index order: reproducible, and no level walks more than a few dozen partials serially.
#define TE_DSUPER 32
#define TE_DPREP_T 1024           // two bins per thread: up to 2048 bins (te_supported's limit)
__global__ __launch_bounds__(TE_DPREP_T) void te_dprep_kernel(TeArgs A) {
  __shared__ int s[2 * TE_DPREP_T], s2[2 * TE_DPREP_T], s3[2 * TE_DPREP_T];      // chunks, super-chunks, chunks of the cold bins (te_dsum's work list)
  const int NB = A.n_dist + 1, t = threadIdx.x;
  int n[2], n2[2];
  int ent[2], hot[2] = {-1, -1};
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int b = t + u * TE_DPREP_T;
    n[u] = 0; ent[u] = 0;
    if (b < NB) { const int row = A.n_item + 1 + b, end = A.seg_end[row]; ent[u] = end ? end - A.seg_start[row] : 0; n[u] = (ent[u] + 63) / 64; }
  }
  // hot bins (TeArgs.dhot): the TE_HB bins with the most entries (ties: lower bin first), at least TE_HOT_BIN_MIN each - te_psum sums their DA rows,
  // one partial per te_psum workgroup in the place of the bin's chunk partials.  A function of the launch's own counts: reproducible.
  {
    __shared__ long long s_w[TE_DPREP_T / 64];
    __shared__ long long s_pick;
    int n_hot = 0;
    for (int k = 0; k < TE_HB; ++k) {
      long long key = -1;
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int b = t + u * TE_DPREP_T;
        if (A.dhot_on && b < NB && hot[u] < 0 && ent[u] >= TE_HOT_BIN_MIN) key = max(key, ((long long)ent[u] << 12) | (long long)(4095 - b));
      }
      for (int o = 32; o > 0; o >>= 1) key = max(key, __shfl_xor(key, o, 64));
      if (lane_id() == 0) s_w[wave_id()] = key;
      __syncthreads();
      if (t == 0) { long long m = -1; for (int q = 0; q < TE_DPREP_T / 64; ++q) m = max(m, s_w[q]); s_pick = m; }
      __syncthreads();
      const long long pick = s_pick;
      __syncthreads();
      if (pick < 0) break;                      // (block-uniform)
      const int pb = 4095 - (int)(pick & 4095);
#pragma unroll
      for (int u = 0; u < 2; ++u) if (t + u * TE_DPREP_T == pb) { hot[u] = k; n[u] = A.npw; }
      if (t == 0) A.dhot[1 + k] = pb;
      ++n_hot;
    }
    if (t == 0) { A.dhot[0] = n_hot; for (int k = n_hot; k < TE_HB; ++k) A.dhot[1 + k] = -1; }
  }
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int b = t + u * TE_DPREP_T;
    if (b < NB) A.dhot[8 + b] = hot[u];
    n2[u] = (n[u] + TE_DSUPER - 1) / TE_DSUPER;
    s[b] = n[u]; s2[b] = n2[u]; s3[b] = hot[u] >= 0 ? 0 : n[u];
  }
  __syncthreads();
  for (int o = 1; o < 2 * TE_DPREP_T; o <<= 1) {          // inclusive Hillis-Steele scan over the 2048 slots
    int v[2], v2[2], v3[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) { const int b = t + u * TE_DPREP_T; v[u] = b >= o ? s[b - o] : 0; v2[u] = b >= o ? s2[b - o] : 0; v3[u] = b >= o ? s3[b - o] : 0; }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < 2; ++u) { const int b = t + u * TE_DPREP_T; s[b] += v[u]; s2[b] += v2[u]; s3[b] += v3[u]; }
    __syncthreads();
  }
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int b = t + u * TE_DPREP_T;
    if (b < NB) { A.dch0[b] = s[b] - n[u]; A.dch1[b] = s2[b] - n2[u]; A.dcc0[b] = s3[b] - (hot[u] >= 0 ? 0 : n[u]); }
  }
  if (t == TE_DPREP_T - 1) { A.dch0[NB] = s[2 * TE_DPREP_T - 1]; A.dch1[NB] = s2[2 * TE_DPREP_T - 1]; A.dcc0[NB] = s3[2 * TE_DPREP_T - 1]; }
#pragma unroll
  for (int u = 0; u < 2; ++u) if (hot[u] >= 0) A.dnf[s[t + u * TE_DPREP_T] - n[u]] = 0;      // te_dhot adds the slices' counts here
}

// one 64-entry chunk of one bin per workgroup iteration, thread = column of DA; thread 0 also counts the chunk's
// TE_ENT_FIRST flags (= distinct sequences of the batch rule: a sequence's entries are contiguous in a row segment)
template <int D>
__global__ __launch_bounds__(3 * D) void te_dsum_kernel(TeArgs A) {
  const int NB = A.n_dist + 1, col = threadIdx.x, lane = lane_id();
  const int total = A.dcc0[NB];                 // the chunks of the COLD bins (a hot bin's chunk partials are te_psum's per-workgroup sums)
  for (int cc = blockIdx.x; cc < total; cc += gridDim.x) {
    int lo = 0, hi = NB - 1;                    // last bin with dcc0[b] <= cc: the bin whose range holds cc (empty ranges - hot or unused bins - start
    while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (A.dcc0[mid] <= cc) lo = mid; else hi = mid - 1; }      // where the next non-empty one does and lose)
    const int row = A.n_item + 1 + lo;
    const int ci = A.dch0[lo] + (cc - A.dcc0[lo]);
    const int c0 = A.seg_start[row] + 64 * (cc - A.dcc0[lo]), ce = min(A.seg_end[row], c0 + 64);
    // the chunk's entries: one per lane (every wave holds the same list), handed out through v_readlane - the row
    // addresses are scalar, and all 64 row loads are in flight behind ONE dependent index load (16 rows behind
    // each of four index batches left the memory pipe idle half of the time: 3.6 TB/s)
    const int mine = A.ent[min(c0 + lane, ce - 1)];
    const bool in = c0 + lane < ce;
    const unsigned long long mdx = __ballot(in && (mine & TE_ENT_DX) != 0);
    const int nf = __builtin_popcountll(__ballot(in && mine < 0));
    float v[64];
#pragma unroll
    for (int u = 0; u < 64; ++u) v[u] = A.G[(size_t)(__builtin_amdgcn_readlane(mine, u) & TE_ENT_ROW) * 3 * D + col];
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll
    for (int b = 0; b < 64; b += 16) {          // (the summation order of the four-batch version)
#pragma unroll
      for (int u = 0; u < 16; ++u) v[b + u] = ((mdx >> (b + u)) & 1ull) ? v[b + u] : 0.f;
      s0 += (v[b + 0] + v[b + 4]) + (v[b + 8] + v[b + 12]); s1 += (v[b + 1] + v[b + 5]) + (v[b + 9] + v[b + 13]);
      s2 += (v[b + 2] + v[b + 6]) + (v[b + 10] + v[b + 14]); s3 += (v[b + 3] + v[b + 7]) + (v[b + 11] + v[b + 15]);
    }
    A.dpart[(size_t)ci * 3 * D + col] = (s0 + s1) + (s2 + s3);
    if (col == 0) A.dnf[ci] = nf;
  }
}

// hot bins: the distinct-sequence count of the batch rule (TE_ENT_FIRST flags of the bin's entries) - the DA rows themselves went through te_psum.
// Workgroup (k, y) = slice y of hot bin k's entries; the slices' counts meet in the bin's first chunk slot (integer atomics: order-free; te_dprep
// zeroed it), the other npw - 1 slots count nothing.
#define TE_DHOT_Y 32
__global__ __launch_bounds__(256) void te_dhot_kernel(TeArgs A) {
  __shared__ int s_w[4];
  const int k = blockIdx.x, t = threadIdx.x;
  if (k >= A.dhot[0]) return;
  const int b = A.dhot[1 + k], row = A.n_item + 1 + b;
  const int s0 = A.seg_start[row], s1 = A.seg_end[row];
  const int per = (((s1 - s0 + TE_DHOT_Y - 1) / TE_DHOT_Y) + 255) & ~255;
  const int lo = s0 + (int)blockIdx.y * per, hi = min(s1, lo + per);
  int c = 0;
  for (int i = lo + t; i < hi; i += 256) c += A.ent[i] < 0 ? 1 : 0;
  for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
  if ((t & 63) == 0) s_w[t >> 6] = c;
  __syncthreads();
  const int base = A.dch0[b];
  if (t == 0) { const int tot = (s_w[0] + s_w[1]) + (s_w[2] + s_w[3]); if (tot) atomicAdd(&A.dnf[base], tot); }
  for (int w = 1 + (int)blockIdx.y * 256 + t; w < A.npw; w += TE_DHOT_Y * 256) A.dnf[base + w] = 0;
}

// one super-chunk (<= TE_DSUPER consecutive chunk partials of one bin) per workgroup iteration
template <int D>
__global__ __launch_bounds__(3 * D) void te_dred_kernel(TeArgs A) {
  const int NB = A.n_dist + 1, col = threadIdx.x;
  const int total = A.dch1[NB];
  for (int si = blockIdx.x; si < total; si += gridDim.x) {
    int lo = 0, hi = NB - 1;
    while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (A.dch1[mid] <= si) lo = mid; else hi = mid - 1; }
    const int c0 = A.dch0[lo] + TE_DSUPER * (si - A.dch1[lo]), c1 = min(A.dch0[lo + 1], c0 + TE_DSUPER);
    float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    int nf = 0;
#pragma unroll
    for (int g = 0; g < TE_DSUPER / 8; ++g)
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int c = c0 + 8 * g + u;
        const float v = A.dpart[(size_t)min(c, c1 - 1) * 3 * D + col];
        a[u] += c < c1 ? v : 0.f;
      }
    if (col == 0) for (int c = c0; c < c1; ++c) nf += A.dnf[c];
    A.dpart2[(size_t)si * 3 * D + col] = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
    if (col == 0) A.dnf2[si] = nf;
  }
}

// S[b] = sum of the bin's super-chunk partials (in order); dgd[b] = S[b] . ui[:, D:2D]; dbn[b] = distinct sequences
template <int D>
__global__ __launch_bounds__(3 * D) void te_dfin_kernel(TeArgs A) {
  __shared__ float S[3 * D];
  __shared__ float gp[3][D];
  const int b = blockIdx.x, col = threadIdx.x;
  const int c0 = A.dch1[b], c1 = A.dch1[b + 1];
  float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  int c = c0;
  for (; c + 7 < c1; c += 8) {
#pragma unroll
    for (int u = 0; u < 8; ++u) a[u] += A.dpart2[(size_t)(c + u) * 3 * D + col];
  }
  for (; c < c1; ++c) a[0] += A.dpart2[(size_t)c * 3 * D + col];
  const float s = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
  S[col] = s;
  A.dsum[(size_t)b * 3 * D + col] = s;
  if (col == 0) { int nf = 0; for (int k = c0; k < c1; ++k) nf += A.dnf2[k]; A.dbn[b] = nf; }
  __syncthreads();
  {   // matvec: thread (part, cc) sums k in [part*D, (part+1)*D), three parts added in order
    const int part = col / D, cc = col % D;
    const float* u = A.ui + (size_t)part * D * 2 * D + D + cc;      // ui[part*D + k][D + cc], row pitch 2D
    float g[4] = {0.f, 0.f, 0.f, 0.f};
    for (int k = 0; k < D; k += 4) {
#pragma unroll
      for (int q = 0; q < 4; ++q) g[q] = fmaf(S[part * D + k + q], u[(size_t)(k + q) * 2 * D], g[q]);
    }
    gp[part][cc] = (g[0] + g[1]) + (g[2] + g[3]);
  }
  __syncthreads();
  if (col < D) A.dgd[(size_t)b * D + col] = (gp[0][col] + gp[1][col]) + gp[2][col];
}

// write-back of the distance-bin rows (bintab): di[b] <- di[b] - alpha * min(n, cap) / n * (dgd[b] + lambda * mult * di[b]) with
// mult = the row's entries (+ the padding row's analytic multiplicity) and n = its distinct sequences (te_dfin; padding row:
// te_rowmap's counter).  Runs AFTER te_dui, which needs the old di.  Re-zeroes the row's segment / padding bookkeeping.
template <int D>
__global__ __launch_bounds__(D) void te_dapply_kernel(TeArgs A, float alpha, float lambda) {
  const int b = blockIdx.x, c = threadIdx.x, row = A.n_item + 1 + b;
  const int end = A.seg_end[row], cnt = end ? end - A.seg_start[row] : 0;
  const bool pad = b == A.n_dist;
  const int am = pad ? A.mult_di[A.n_dist] : 0, an = pad ? A.nseq_di[A.n_dist] : 0;
  const int nseq = pad ? an : A.dbn[b];
  __syncthreads();                              // every thread has read the counters before thread 0 clears them
  if (end != 0 || an != 0) {
    float sc, lm; rule_scales(alpha, lambda, nseq, cnt + am, A.bcap, sc, lm);
    float* t = A.di + (size_t)b * D + c;
    const float v = *t;
    *t = v - sc * (A.dgd[(size_t)b * D + c] + lm * v);
  }
  if (c == 0) { A.seg_end[row] = 0; if (pad) { A.mult_di[A.n_dist] = 0; A.nseq_di[A.n_dist] = 0; } }
}

// d ui[k][D + c] = sum_b S[b][k] * di[b][c]  (di BEFORE this launch's write-back) -> slab 0 (zero on entry: plain store).
// Workgroup = one row k, thread = (quarter of the bins, column c).
template <int D>
__global__ __launch_bounds__(4 * D) void te_dui_kernel(TeArgs A) {
  __shared__ float qp[4][D];
  const int k = blockIdx.x, c = threadIdx.x % D, qtr = threadIdx.x / D, NB = A.n_dist + 1;
  const int per = (NB + 3) / 4, b0 = qtr * per, b1 = min(NB, b0 + per);
  float a[4] = {0.f, 0.f, 0.f, 0.f};
  int b = b0;
  for (; b + 3 < b1; b += 4) {
#pragma unroll
    for (int u = 0; u < 4; ++u) a[u] = fmaf(A.dsum[(size_t)(b + u) * 3 * D + k], A.di[(size_t)(b + u) * D + c], a[u]);
  }
  for (; b < b1; ++b) a[0] = fmaf(A.dsum[(size_t)b * 3 * D + k], A.di[(size_t)b * D + c], a[0]);
  qp[qtr][c] = (a[0] + a[1]) + (a[2] + a[3]);
  __syncthreads();
  if (qtr == 0) A.slab[A.dl.ui + (size_t)k * 2 * D + D + c] = (qp[0][c] + qp[1][c]) + (qp[2][c] + qp[3][c]);
}

// The distance-bin chain: per-bin sums of DA -> the two small dense products -> the bin rows' write-back.  It needs DA (te_rec_bwd), the
// sorted entries and the OLD ui / di, and writes only its own buffers, the di rows and the di half of slab 0's d ui.
// (early: the chunk offsets of the bins' entry segments - te_dprep, a one-workgroup scan that needs only the sorted segments - were formed
// behind the slot sort already: launch_te_dprep)
hipError_t launch_te_dprep(TeArgs& A, hipStream_t st) {
  hipLaunchKernelGGL(te_dprep_kernel, dim3(1), dim3(TE_DPREP_T), 0, st, A);
  return hipGetLastError();
}
template <int D>
static hipError_t te_bins_t(TeArgs& A, float alpha, float lambda, int num_cu, hipStream_t sb, Timing* tm, bool early = false) {
  // the per-bin reduction of DA rows is scatter traffic (te_dsum: one pass over DA at HBM speed); the two small
  // dense products that follow (S . ui[:, D:], S^T . di) are timed on their own
  tm->begin("te_dsum", sb);
  if (!early) hipLaunchKernelGGL(te_dprep_kernel, dim3(1), dim3(TE_DPREP_T), 0, sb, A);
  hipLaunchKernelGGL(te_dsum_kernel<D>, dim3(num_cu * 8), dim3(3 * D), 0, sb, A);
  if (A.dhot_on) hipLaunchKernelGGL(te_dhot_kernel, dim3(TE_HB, TE_DHOT_Y), dim3(256), 0, sb, A);
  tm->end(sb);
  tm->begin("te_bin_gemm", sb);
  hipLaunchKernelGGL(te_dred_kernel<D>, dim3(num_cu * 2), dim3(3 * D), 0, sb, A);
  hipLaunchKernelGGL(te_dfin_kernel<D>, dim3(A.n_dist + 1), dim3(3 * D), 0, sb, A);
  hipLaunchKernelGGL(te_dui_kernel<D>, dim3(3 * D), dim3(4 * D), 0, sb, A);
  hipLaunchKernelGGL(te_dapply_kernel<D>, dim3(A.n_dist + 1), dim3(D), 0, sb, A, alpha, lambda);
  tm->end(sb);
  return hipGetLastError();
}
hipError_t launch_te_bins(TeArgs& A, float alpha, float lambda, int num_cu, hipStream_t sb, Timing* tm) {      // (the early chain: te_dprep ran behind the sort)
  if (A.dim == 64) return te_bins_t<64>(A, alpha, lambda, num_cu, sb, tm, true);
  if (A.dim == 128) return te_bins_t<128>(A, alpha, lambda, num_cu, sb, tm, true);
  if (A.dim == 256) return te_bins_t<256>(A, alpha, lambda, num_cu, sb, tm, true);
  return hipErrorInvalidValue;
}

template <int D>
static hipError_t te_scatter_t(TeArgs& A, float alpha, float lambda, int num_cu, hipStream_t st, Timing* tm) {
  const int R = A.n_item + 1 + A.n_dist + 1;
  int grid = (R + 3) / 4;
  if (grid > num_cu * 32) grid = num_cu * 32;
  if (A.side && hipStreamWaitEvent(st, A.ev_sorted, 0) != hipSuccess) return hipGetLastError();     // the sorted entries
  // The distance-bin chain and the POI rows' reduction below touch disjoint rows and share only read-only inputs (the sorted entries,
  // DA, H): with a side stream they run next to each other - te_dsum streams DA at HBM speed while te_reduce is a chain of dependent
  // loads, and the five small kernels of the bin chain hide behind the reduction.  Large launches (A.early_bins): launch_te_train has
  // already started the chain on the side stream, next to te_gemm_dx, and only the join is left here.  Otherwise (early_bins off) it forks here
  // (>= 2048 sequences) or runs inline (the two cross-stream dependencies cost ~35 us, more than the whole tail of a one-sequence launch).
  // Timing: forked regions OVERLAP the main stream's and stretch each other; `te_tail` spans this function's fork to join.
  const bool early = A.bintab && A.early_bins;
  const bool fork = !early && A.bintab && A.side && !(A.dbg & 1) && A.n_seq >= 2048;
  hipStream_t sb = fork ? A.side : st;
  const long tail = tm->span_begin("te_tail", st);
  if (fork && (hipEventRecord(A.ev_bwd, st) != hipSuccess || hipStreamWaitEvent(sb, A.ev_bwd, 0) != hipSuccess)) return hipGetLastError();
  if (A.bintab && !early) {
    hipError_t be = te_bins_t<D>(A, alpha, lambda, num_cu, sb, tm);
    if (be != hipSuccess) return be;
  }
  if (fork && hipEventRecord(A.ev_fin, sb) != hipSuccess) return hipGetLastError();
  tm->begin("te_scatter", st);
  if (A.ppoi) hipLaunchKernelGGL((te_reduce_kernel<D, true>), dim3(grid), dim3(256), 0, st, A, alpha, lambda);
  else hipLaunchKernelGGL((te_reduce_kernel<D, false>), dim3(grid), dim3(256), 0, st, A, alpha, lambda);
  if (A.hot_early) { if (hipStreamWaitEvent(st, A.ev_hr1, 0) != hipSuccess) return hipGetLastError(); }      // (the chunk sums ran beside te_rec_bwd: launch_te_train)
  else hipLaunchKernelGGL(te_hot_reduce_kernel<D>, dim3(num_cu * 8), dim3(256), 0, st, A);
  hipLaunchKernelGGL(te_hot_apply_kernel<D>, dim3(num_cu * 8), dim3(256), 0, st, A, alpha, lambda);
  tm->end(st);
  if (fork && hipStreamWaitEvent(st, A.ev_fin, 0) != hipSuccess) return hipGetLastError();       // join: dense_apply reads te_dui's slab
  if (early && hipStreamWaitEvent(st, A.ev_slots, 0) != hipSuccess) return hipGetLastError();     // (recorded behind the chain by launch_te_train)
  tm->span_end(tail, st);
  return hipGetLastError();
}

hipError_t launch_te_hot_reduce(TeArgs& A, int num_cu, hipStream_t st) {
  if (A.dim == 64) hipLaunchKernelGGL(te_hot_reduce_kernel<64>, dim3(num_cu * 8), dim3(256), 0, st, A);
  else if (A.dim == 128) hipLaunchKernelGGL(te_hot_reduce_kernel<128>, dim3(num_cu * 8), dim3(256), 0, st, A);
  else if (A.dim == 256) hipLaunchKernelGGL(te_hot_reduce_kernel<256>, dim3(num_cu * 8), dim3(256), 0, st, A);
  else return hipErrorInvalidValue;
  return hipGetLastError();
}

hipError_t launch_te_scatter(TeArgs& A, float alpha, float lambda, int num_cu, hipStream_t st, Timing* tm) {
  if (A.dim == 64) return te_scatter_t<64>(A, alpha, lambda, num_cu, st, tm);
  if (A.dim == 128) return te_scatter_t<128>(A, alpha, lambda, num_cu, st, tm);
  if (A.dim == 256) return te_scatter_t<256>(A, alpha, lambda, num_cu, st, tm);
  return hipErrorInvalidValue;
}

}  // namespace poi
