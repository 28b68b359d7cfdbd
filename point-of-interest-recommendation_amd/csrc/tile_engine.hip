// Tile engine ("throughput mode"): the Distance2Pre training step for a BATCH of sequences, decomposed
// so that every heavy contraction is a tile GEMM on the f32-input matrix cores (v_mfma_f32_32x32x2_f32
// and v_mfma_f32_16x16x4_f32: exact f32 products / accumulation), with the recurrence confined to two
// small per-tile kernels:
//
//   te_len/scan  per-sequence step counts -> exclusive scan -> packed row offsets
//   te_rowmap    packed row -> CSR position; sorted-scatter slots (one per table touch); padding-row bookkeeping
//   te_pack      weights -> MFMA B-fragment order (every weight load is then a coalesced 1-KiB stream)
//   te_sort      (te_scatter.hip) stable radix sort of the slots by table row -> per-row entry segments
//   te_gather    E[r] = lt[p_{t+1}] - lt[q_{t+1}]                                         (HBM-bound)
//   te_gemm_ax   G[r] = [lt[p_t] | di[dp_t]] . ui^T + bi   (te_gemm_nt: the input rows are gathered from the
//                tables straight into the LDS tiles; all steps at once, 128 x 128 tiles, K = 2D)
//   te_rec_fwd16 per 16-sequence tile, t ascending: gates from G + h_{t-1} . wh^T  -> G := z|r|c, H, RH
//   te_head      per 32-row tile: logits = H . vs^T + bs, softmax, BPR + survival losses, d logits -> DL,
//                DH = dlogits . vs + g * E, g, d bs / d wd partials
//   te_rec_bwd16 per 16-sequence tile, t descending (BPTT): G := da_z|da_r|da_c, d bi partials
//   te_wgrad     split-K  d ui = DA^T . X,  d wh = DA^T . [Hprev | RH],  d vs = DL^T . H   -> per-chunk slabs
//   te_gemm_dx   dx = DA . ui  (te_gemm_nt; stored over X)
//   te_finalize  per-sequence losses, loss-weight statistics
// followed by te_scatter (te_scatter.hip: ordered per-row sums of dx / g*h + sparse SGD write-back) and
// the shared dense_apply (seq_engine.hip).
//
// Math and semantics are identical to the per-sequence engine (public/GRU_Spatial.py:127-229, batch
// rule of include/poi_hip.h); only the summation order differs.
#include "poi_common.h"
#include "poi_kernels.h"

#include <stdlib.h>
#include <stdio.h>

namespace poi {

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define TE_BLOCK 256
// D >= 256: the recurrent weights no longer fit the register file of one workgroup - streaming kernels (te_rec_fwd32 / bwd32);
// TeArgs.rec32 also selects them at D = 128 (engine 3 of poi_ctx_set_engine: same arithmetic, different tiling - tests).

__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}
// Workgroup barrier that orders LDS traffic only.  __syncthreads() also carries a release fence that
// waits for every outstanding GLOBAL store / atomic of the wave (vmcnt(0)) - microseconds per barrier
// in these loops, where global results are consumed only by later kernels and just LDS tiles cross
// waves.  (Loads feeding ds_write are still waited for by the data dependence.)
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}
// tanh on the hardware exp: 1 - 2 / (1 + e^{2x}); saturates correctly (e^{2x} -> inf gives 1, -> 0 gives -1);
// absolute error ~1e-7, far inside the 1e-5 parity bar.  The library tanhf is a long branchy routine.
// (v_rcp_f32, 1 ulp, instead of the IEEE division sequence: the gate math sits on the per-step latency chain)
__device__ __forceinline__ float fast_tanh(float x) { return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __expf(2.0f * x)); }
__device__ __forceinline__ float fast_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
// C/D layout of the 32x32 tile: element (row, col) of register r in lane l
__device__ __forceinline__ int c_row(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

// -------------------------------------------------------------------------------------------------
// Packed B operand: logical B[k][n], k < K (padded to K8*8), n < N (padded to NT*32):
//   P[(nt * K8 + m) * 64 + lane] = float4{ B[8m + 4h + c][32 nt + j], c = 0..3 },  lane = h*32 + j.
// MFMA step 4m+c consumes component c of both operands, so A (read as float4 at k = 8m+4h from a
// row-major LDS tile) and B agree on a permuted k order and every loaded byte is used.
// -------------------------------------------------------------------------------------------------
// n16 == 1: fragments of the 16x16x4 MFMA (n tiles of 16 columns, k groups of 16):
//   P[(nt * KG + m) * 64 + lane] = float4{ B[16m + 4g + c][16 nt + j], c = 0..3 },  lane = g*16 + j  (K8 field = KG).
struct PackJob { const float* src; int sk, sn, K, N, K8, NT; float4* dst; int n16; };
struct PackJobs { PackJob j[12]; int n; };

// n16 == 2: fragments of v_mfma_f32_16x16x32_bf16 for the split-operand recurrent kernels (te_rec_fwd16 / bwd16 <SP>): every float32
// weight as THREE bf16 planes (split3: w = w1 + w2 + w3 exactly), n tiles of 16 columns, k groups of 32 (K8 field = K / 32):
//   P[((nt * KG + m) * 3 + plane) * 64 + lane] = 8 x bf16 { B_plane[32m + 8g + c][16 nt + j], c = 0..7 },  lane = g*16 + j.
// split3: w1 = bf16(w) rounded to nearest (v_cvt_pk_bf16_f32), w2 = the top 8 bits of the exact remainder r1 = w - w1, w3 = r1 - w2
// (<= 8 significant bits: exact), so w = w1 + w2 + w3 exactly and |w2| <= 2^-9 |w|, |w3| <= 2^-17 |w|.  The rounding of w1 gives the
// remainder a random sign relative to w: the dropped cross terms do not add up.  (Truncating w1 as well is one instruction shorter,
// but every dropped term then has the sign of the product: measured 7.0e-6 instead of 4.7e-6 on the timed Gowalla launch.)
// Returns the planes as float32 bit patterns with zero low halves (bf16 = the high 16 bits).
__device__ __forceinline__ void split3(float v, unsigned& u1, unsigned& u2, unsigned& u3) {
  const __bf16 h1 = (__bf16)v;
  u1 = (unsigned)__builtin_bit_cast(unsigned short, h1) << 16;
  const float r1 = v - __uint_as_float(u1);
  u2 = __float_as_uint(r1) & 0xFFFF0000u;
  u3 = __float_as_uint(r1 - __uint_as_float(u2));
}

// two values -> one packed dword per plane (low half = x0): split3's planes for a pair, v_cvt_pk_bf16_f32 / v_perm_b32 doing the packing
__device__ __forceinline__ void wg_split2(float x0, float x1, unsigned& p1, unsigned& p2, unsigned& p3) {
  typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  const f32x2 xv = {x0, x1};
  p1 = __builtin_bit_cast(unsigned, __builtin_convertvector(xv, bf16x2));
  const float r0 = x0 - __uint_as_float(p1 << 16), r1 = x1 - __uint_as_float(p1 & 0xFFFF0000u);
  p2 = __builtin_amdgcn_perm(__float_as_uint(r1), __float_as_uint(r0), 0x07060302u);      // {r1.hi16, r0.hi16}
  const float s0 = r0 - __uint_as_float(p2 << 16), s1 = r1 - __uint_as_float(p2 & 0xFFFF0000u);
  p3 = __builtin_amdgcn_perm(__float_as_uint(s1), __float_as_uint(s0), 0x07060302u);
}
// two planes only, BOTH rounded to nearest (x - x1 - x2 <= 2^-18 |x|): the backward-only products of te_head3
__device__ __forceinline__ void wg_split2r(float x0, float x1, unsigned& p1, unsigned& p2) {
  typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  const f32x2 xv = {x0, x1};
  p1 = __builtin_bit_cast(unsigned, __builtin_convertvector(xv, bf16x2));
  const f32x2 rv = {x0 - __uint_as_float(p1 << 16), x1 - __uint_as_float(p1 & 0xFFFF0000u)};
  p2 = __builtin_bit_cast(unsigned, __builtin_convertvector(rv, bf16x2));
}

// (vb of nvb: the virtual block of the job - te_pack_kernel's own grid, or a slice of te_one_in_kernel's)
__device__ __forceinline__ void te_pack_block(const PackJob& j, const int vb, const int nvb) {
  if (j.n16 == 3) {      // plain transposed copy for the per-sequence recurrent kernels: dst[n * K + k] = B[k][n]
    float* dst = reinterpret_cast<float*>(j.dst);
    for (int e = vb * TE_BLOCK + threadIdx.x; e < j.K * j.N; e += nvb * TE_BLOCK) {
      const int k = e / j.N, n = e % j.N;                  // (reads coalesced along n, writes strided: 64 K elements, once per launch)
      dst[(size_t)n * j.K + k] = j.src[(size_t)k * j.sk + (size_t)n * j.sn];
    }
    return;
  }
  if (j.n16 == 4 || j.n16 == 5) {      // bf16 x 3 planes in v_mfma_f32_32x32x16_bf16 fragment order (streaming recurrent kernels <SP>): K8 field = K / 16,
    // (n16 == 5: column n of B is row (n % 3) * (N / 3) + n / 3 of the source - the gate-interleaved columns of the forward table, te_uiperm's order)
    //   P[((nt * KG + m) * 3 + plane) * 64 + lane] = 8 x bf16 { B_plane[16m + 8h + c][32 nt + j], c = 0..7 },  lane = 32 h + j
    const int total4 = j.NT * j.K8 * 3 * 64;
    for (int e = vb * TE_BLOCK + threadIdx.x; e < total4; e += nvb * TE_BLOCK) {
      const int lane = e & 63, f = e >> 6, pl = f % 3, m = (f / 3) % j.K8, nt = (f / 3) / j.K8;
      const int n = nt * 32 + (lane & 31), k0 = 16 * m + 8 * (lane >> 5);
      const int ns = j.n16 == 5 ? (n % 3) * (j.N / 3) + n / 3 : n;
      unsigned h[8];
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        const int k = k0 + c;
        const float v = (k < j.K && n < j.N) ? j.src[(size_t)k * j.sk + (size_t)ns * j.sn] : 0.f;
        unsigned u[3];
        split3(v, u[0], u[1], u[2]);
        h[c] = (pl == 0 ? u[0] : pl == 1 ? u[1] : u[2]) >> 16;
      }
      reinterpret_cast<uint4*>(j.dst)[e] = make_uint4(h[0] | (h[1] << 16), h[2] | (h[3] << 16), h[4] | (h[5] << 16), h[6] | (h[7] << 16));
    }
    return;
  }
  if (j.n16 == 2) {
    const int total3 = j.NT * j.K8 * 3 * 64;
    for (int e = vb * TE_BLOCK + threadIdx.x; e < total3; e += nvb * TE_BLOCK) {
      const int lane = e & 63, f = e >> 6, pl = f % 3, m = (f / 3) % j.K8, nt = (f / 3) / j.K8;
      const int n = nt * 16 + (lane & 15), k0 = 32 * m + 8 * (lane >> 4);
      unsigned h[8];
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        const int k = k0 + c;
        const float v = (k < j.K && n < j.N) ? j.src[(size_t)k * j.sk + (size_t)n * j.sn] : 0.f;
        unsigned u[3];
        split3(v, u[0], u[1], u[2]);
        h[c] = (pl == 0 ? u[0] : pl == 1 ? u[1] : u[2]) >> 16;
      }
      reinterpret_cast<uint4*>(j.dst)[e] = make_uint4(h[0] | (h[1] << 16), h[2] | (h[3] << 16), h[4] | (h[5] << 16), h[6] | (h[7] << 16));
    }
    return;
  }
  const int total = j.NT * j.K8 * 64;
  for (int e = vb * TE_BLOCK + threadIdx.x; e < total; e += nvb * TE_BLOCK) {
    const int lane = e & 63, m = (e >> 6) % j.K8, nt = (e >> 6) / j.K8;
    const int n = j.n16 ? nt * 16 + (lane & 15) : nt * 32 + (lane & 31);
    const int k0 = j.n16 ? 16 * m + 4 * (lane >> 4) : 8 * m + 4 * (lane >> 5);
    float v[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int k = k0 + c;
      v[c] = (k < j.K && n < j.N) ? j.src[(size_t)k * j.sk + (size_t)n * j.sn] : 0.f;
    }
    j.dst[e] = make_float4(v[0], v[1], v[2], v[3]);
  }
}

__global__ __launch_bounds__(TE_BLOCK) void te_pack_kernel(PackJobs J) { te_pack_block(J.j[blockIdx.y], blockIdx.x, gridDim.x); }

// acc[i][j] += A_i (32 x 8*K8, LDS row-major, leading dim lda) . B_j (packed n-tile j of `bp`)
// Explicit software pipeline, one k-group deep: the packed B fragments (L2) and the A fragments (LDS)
// of k-group m+1 are requested before the MFMAs of k-group m are issued; sched_barriers pin that
// order, because the compiler otherwise sinks every operand read next to its first MFMA (exposing the
// LDS latency once per k-group) or, fully unrolled, hoists them all and spills.  The loop is unrolled
// by two only (register renaming of the double buffers); the kernels using it run 2-3 waves per SIMD,
// which covers the L2 latency of the B stream.
template <int MT, int NTW, int K8>
__device__ __forceinline__ void mma_lds_packed(f32x16 (&acc)[MT][NTW], const float* __restrict__ ldsA, int lda,
                                               const float4* __restrict__ bp, const int (&nt)[NTW]) {
  const int lane = lane_id(), li = lane & 31, h = lane >> 5;
  const float* arow = ldsA + li * lda + 4 * h;
  const float4* bj[NTW];
  float4 bc[NTW], bn[NTW], ac[MT], an[MT];
#pragma unroll
  for (int j = 0; j < NTW; ++j) { bj[j] = bp + ((size_t)nt[j] * K8) * 64 + lane; bc[j] = *bj[j]; }
#pragma unroll
  for (int i = 0; i < MT; ++i) ac[i] = *reinterpret_cast<const float4*>(arow + (size_t)i * 32 * lda);
#pragma unroll 2
  for (int m = 0; m < K8; ++m) {
    const int mn = m + 1 < K8 ? m + 1 : m;
#pragma unroll
    for (int j = 0; j < NTW; ++j) bn[j] = bj[j][(size_t)mn * 64];
#pragma unroll
    for (int i = 0; i < MT; ++i) an[i] = *reinterpret_cast<const float4*>(arow + (size_t)i * 32 * lda + 8 * mn);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NTW; ++j) {
        acc[i][j] = mfma32(ac[i].x, bc[j].x, acc[i][j]);
        acc[i][j] = mfma32(ac[i].y, bc[j].y, acc[i][j]);
        acc[i][j] = mfma32(ac[i].z, bc[j].z, acc[i][j]);
        acc[i][j] = mfma32(ac[i].w, bc[j].w, acc[i][j]);
      }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < NTW; ++j) bc[j] = bn[j];
#pragma unroll
    for (int i = 0; i < MT; ++i) ac[i] = an[i];
  }
}

template <int K8>
__device__ __forceinline__ void load_bfrag(float4 (&b)[K8], const float4* __restrict__ bp, int nt) {
#pragma unroll
  for (int m = 0; m < K8; ++m) b[m] = bp[((size_t)nt * K8 + m) * 64 + lane_id()];
}

typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}
// 16-row tile x resident 16-column B fragments on v_mfma_f32_16x16x4_f32: A[i][k] is read as float4 at
// k = 16m + 4g (lane = 16g + i), matching the n16 packing, so four MFMAs consume one LDS read.  NG
// gates share the A fragments; with one gate the k-groups alternate between two accumulators (two
// independent MFMA chains).  Result layout: acc[g][r] = D[4*(lane/16) + r][lane%16].
template <int KG, int NG>
__device__ __forceinline__ void mma16_regb(f32x4 (&acc)[NG], const float* __restrict__ ldsA, int lda, const float4 (&b)[NG][KG]) {
  const int lane = lane_id();
  const float* arow = ldsA + (lane & 15) * lda + 4 * (lane >> 4);
  float4 aq[2];
  aq[0] = *reinterpret_cast<const float4*>(arow);
  f32x4 alt = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int m = 0; m < KG; ++m) {
    if (m + 1 < KG) aq[(m + 1) & 1] = *reinterpret_cast<const float4*>(arow + 16 * (m + 1));
    __builtin_amdgcn_sched_barrier(0);
    const float4 a = aq[m & 1];
    if (NG == 1 && (m & 1)) {
      alt = mfma16(a.x, b[0][m].x, alt); alt = mfma16(a.y, b[0][m].y, alt);
      alt = mfma16(a.z, b[0][m].z, alt); alt = mfma16(a.w, b[0][m].w, alt);
    } else {
#pragma unroll
      for (int g = 0; g < NG; ++g) acc[g] = mfma16(a.x, b[g][m].x, acc[g]);
#pragma unroll
      for (int g = 0; g < NG; ++g) acc[g] = mfma16(a.y, b[g][m].y, acc[g]);
#pragma unroll
      for (int g = 0; g < NG; ++g) acc[g] = mfma16(a.z, b[g][m].z, acc[g]);
#pragma unroll
      for (int g = 0; g < NG; ++g) acc[g] = mfma16(a.w, b[g][m].w, acc[g]);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  if (NG == 1) acc[0] += alt;
}

// Split-operand contraction of the recurrent kernels (<SP>): the float32-input MFMA runs at the vector rate on gfx950 (1/16 of the
// bf16 rate), and a recurrent step IS one CU's matrix throughput.  Both operands are therefore held as three bf16 planes
// (split3: x = x1 + x2 + x3 exactly, 8 + 8 + 8 mantissa bits) and the product is the six partial products down to 2^-16
//   x1 y1  +  (x1 y2 + x2 y1)  +  (x1 y3 + x2 y2 + x3 y1)
// on v_mfma_f32_16x16x32_bf16 (exact bf16 products, float32 accumulation): 6 MFMAs of 16 cycles per 32 k against 8 float32 ones of
// 32 cycles - 2.7x less matrix time.  Dropped: x2 y3 + x3 y2 + x3 y3 <= 2^-25 |x y| per product, signs random (measured on random
// operands: 6e-9 rms of the result against 2e-7 of float32 accumulation noise for the same product).  The leading products go to one accumulator,
// the five small ones to another (added once at the end).
// A planes in LDS: plane p of tile row i at ldsA[(16 p + i) * ldh ..], bf16; B planes resident (te_pack, n16 == 2).
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ f32x4 mfma16b(const uint4& a, const uint4& b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
// planes 1 and 2 of n tile `nt` -> registers; plane 3 -> LDS in fragment order (dst[m * 64 + lane]: lane-linear 16-byte reads).
// The register file of a CU cannot hold all three planes of the recurrent weights next to the working set of a step (288 of 512 KB);
// the third plane enters one MFMA in six, so it is the one that is read from LDS every step.
template <int KG>
__device__ __forceinline__ void load_bfrag3(uint4 (&b)[KG][2], uint4* __restrict__ dst, const float4* __restrict__ bp, int nt) {
  const uint4* q = reinterpret_cast<const uint4*>(bp) + ((size_t)nt * KG) * 3 * 64 + lane_id();
#pragma unroll
  for (int m = 0; m < KG; ++m) {
    b[m][0] = q[(m * 3 + 0) * 64]; b[m][1] = q[(m * 3 + 1) * 64];
    dst[m * 64 + lane_id()] = q[(m * 3 + 2) * 64];
  }
}
// x -> its three planes at LDS element `at` (plane stride ps elements); ds_write_b16_d16_hi stores the high halves directly
__device__ __forceinline__ void split3_store(unsigned short* __restrict__ at, int ps, float x) {
  unsigned u1, u2, u3;
  split3(x, u1, u2, u3);
  at[0] = (unsigned short)(u1 >> 16); at[ps] = (unsigned short)(u2 >> 16); at[2 * ps] = (unsigned short)(u3 >> 16);
}
// two gates sharing the A planes: hi / lo accumulators of gate 0 and gate 1 (four independent MFMA chains); c0 / c1: the gates'
// third B planes in LDS (already offset by the lane)
template <int KG>
__device__ __forceinline__ void mma16s_g2(f32x4& h0, f32x4& l0, f32x4& h1, f32x4& l1, const unsigned short* __restrict__ ldsA, int ldh,
                                          const uint4 (&b0)[KG][2], const uint4 (&b1)[KG][2], const uint4* __restrict__ c0, const uint4* __restrict__ c1) {
  const int lane = lane_id(), ps = 16 * ldh;
  const unsigned short* arow = ldsA + (lane & 15) * ldh + 8 * (lane >> 4);
#pragma unroll
  for (int m = 0; m < KG; ++m) {
    const uint4 a1 = *reinterpret_cast<const uint4*>(arow + 32 * m), a2 = *reinterpret_cast<const uint4*>(arow + ps + 32 * m),
                a3 = *reinterpret_cast<const uint4*>(arow + 2 * ps + 32 * m);
    const uint4 b03 = c0[m * 64], b13 = c1[m * 64];
    h0 = mfma16b(a1, b0[m][0], h0); h1 = mfma16b(a1, b1[m][0], h1);
    l0 = mfma16b(a1, b0[m][1], l0); l1 = mfma16b(a1, b1[m][1], l1);
    l0 = mfma16b(a2, b0[m][0], l0); l1 = mfma16b(a2, b1[m][0], l1);
    l0 = mfma16b(a2, b0[m][1], l0); l1 = mfma16b(a2, b1[m][1], l1);
    l0 = mfma16b(a3, b0[m][0], l0); l1 = mfma16b(a3, b1[m][0], l1);
    l0 = mfma16b(a1, b03, l0); l1 = mfma16b(a1, b13, l1);
  }
}
// one gate: the small products alternate between two accumulators (three chains)
// TR: operands swapped - the weight fragments go in as the A operand (the fragment layouts of the two operands of the 16x16x32 MFMA are the
// same), so the result is transposed: a lane then holds four consecutive output UNITS of one tile row instead of one unit of four rows
template <int KG, bool TR = false>
__device__ __forceinline__ void mma16s_g1(f32x4& h, f32x4& l, f32x4& l2, const unsigned short* __restrict__ ldsA, int ldh, const uint4 (&b)[KG][2],
                                          const uint4* __restrict__ c) {
  const int lane = lane_id(), ps = 16 * ldh;
  const unsigned short* arow = ldsA + (lane & 15) * ldh + 8 * (lane >> 4);
#pragma unroll
  for (int m = 0; m < KG; ++m) {
    const uint4 a1 = *reinterpret_cast<const uint4*>(arow + 32 * m), a2 = *reinterpret_cast<const uint4*>(arow + ps + 32 * m),
                a3 = *reinterpret_cast<const uint4*>(arow + 2 * ps + 32 * m);
    const uint4 b3 = c[m * 64];
    if constexpr (TR) {
      h = mfma16b(b[m][0], a1, h);
      l = mfma16b(b[m][1], a1, l); l2 = mfma16b(b[m][0], a2, l2);
      l = mfma16b(b[m][0], a3, l); l2 = mfma16b(b[m][1], a2, l2);
      l = mfma16b(b3, a1, l);
    } else {
    h = mfma16b(a1, b[m][0], h);
    l = mfma16b(a1, b[m][1], l); l2 = mfma16b(a2, b[m][0], l2);
    l = mfma16b(a3, b[m][0], l); l2 = mfma16b(a2, b[m][1], l2);
    l = mfma16b(a1, b3, l);
    }
  }
}

// Split products for the STREAMING recurrent kernels (te_rec_fwd32 / bwd32 <SP>: D = 256): A planes (bf16, 32 rows) in LDS, B planes streamed
// from L2 in v_mfma_f32_32x32x16_bf16 fragment order (te_pack n16 == 4), one k group of prefetch as mma_lds_packed.  Six products per k group
// into ONE accumulator per n tile (float32-level accuracy; the register file has no room for a second set).
__device__ __forceinline__ f32x16 mfma32b(const uint4& a, const uint4& b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
template <int NTW, int KG>
__device__ __forceinline__ void mma_lds_packed_s3(f32x16 (&acc)[1][NTW], const unsigned short* __restrict__ ldsA, int ldh,
                                                  const float4* __restrict__ bp, const int (&nt)[NTW], const int kstride = KG) {
  // (kstride: k groups per n tile of the packed operand - more than KG when this call contracts over a K slice of it)
  const int lane = lane_id(), li = lane & 31, h = lane >> 5, ps = 32 * ldh;
  const unsigned short* arow = ldsA + li * ldh + 8 * h;
  const uint4* bj[NTW];
  uint4 bc[NTW][3], bn[NTW][3], ac[3], an[3];
#pragma unroll
  for (int j = 0; j < NTW; ++j) {
    bj[j] = reinterpret_cast<const uint4*>(bp) + ((size_t)nt[j] * kstride) * 3 * 64 + lane;
#pragma unroll
    for (int p = 0; p < 3; ++p) bc[j][p] = bj[j][p * 64];
  }
#pragma unroll
  for (int p = 0; p < 3; ++p) ac[p] = *reinterpret_cast<const uint4*>(arow + p * ps);
#pragma unroll 2
  for (int m = 0; m < KG; ++m) {
    const int mn = m + 1 < KG ? m + 1 : m;
#pragma unroll
    for (int j = 0; j < NTW; ++j)
#pragma unroll
      for (int p = 0; p < 3; ++p) bn[j][p] = bj[j][((size_t)mn * 3 + p) * 64];
#pragma unroll
    for (int p = 0; p < 3; ++p) an[p] = *reinterpret_cast<const uint4*>(arow + p * ps + 16 * mn);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < NTW; ++j) {
      acc[0][j] = mfma32b(ac[0], bc[j][2], acc[0][j]); acc[0][j] = mfma32b(ac[1], bc[j][1], acc[0][j]); acc[0][j] = mfma32b(ac[2], bc[j][0], acc[0][j]);
      acc[0][j] = mfma32b(ac[0], bc[j][1], acc[0][j]); acc[0][j] = mfma32b(ac[1], bc[j][0], acc[0][j]);
      acc[0][j] = mfma32b(ac[0], bc[j][0], acc[0][j]);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < NTW; ++j)
#pragma unroll
      for (int p = 0; p < 3; ++p) bc[j][p] = bn[j][p];
#pragma unroll
    for (int p = 0; p < 3; ++p) ac[p] = an[p];
  }
}

// The same with the B fragments of PF k groups in flight (a ring of PF + 1 register sets): for callers whose
// B stream comes from L2 at ~1 us under load while one k group is 0.16 us of MFMA work (te_head_big3).
template <int NTW, int KG, int PF>
__device__ __forceinline__ void mma_lds_packed_s3p(f32x16 (&acc)[1][NTW], const unsigned short* __restrict__ ldsA, int ldh,
                                                   const float4* __restrict__ bp, const int (&nt)[NTW], const int kstride = KG) {
  constexpr int R = PF + 1;
  const int lane = lane_id(), li = lane & 31, h = lane >> 5, ps = 32 * ldh;
  const unsigned short* arow = ldsA + li * ldh + 8 * h;
  const uint4* bj[NTW];
  uint4 bq[R][NTW][3], ac[3], an[3];
#pragma unroll
  for (int j = 0; j < NTW; ++j) bj[j] = reinterpret_cast<const uint4*>(bp) + ((size_t)nt[j] * kstride) * 3 * 64 + lane;
#pragma unroll
  for (int u = 0; u < PF; ++u)
#pragma unroll
    for (int j = 0; j < NTW; ++j)
#pragma unroll
      for (int p = 0; p < 3; ++p) bq[u][j][p] = bj[j][((size_t)u * 3 + p) * 64];
#pragma unroll
  for (int p = 0; p < 3; ++p) ac[p] = *reinterpret_cast<const uint4*>(arow + p * ps);
  for (int m0 = 0; m0 < KG; m0 += R) {
#pragma unroll
    for (int u = 0; u < R; ++u) {
      const int m = m0 + u, ml = min(m + PF, KG - 1), mn = min(m + 1, KG - 1);
      if (KG % R != 0 && m >= KG) break;
      // (ring slot of group m + PF = the slot group m - 1 has just left)
#pragma unroll
      for (int j = 0; j < NTW; ++j)
#pragma unroll
        for (int p = 0; p < 3; ++p) bq[(u + PF) % R][j][p] = bj[j][((size_t)ml * 3 + p) * 64];
#pragma unroll
      for (int p = 0; p < 3; ++p) an[p] = *reinterpret_cast<const uint4*>(arow + p * ps + 16 * mn);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < NTW; ++j) {
        acc[0][j] = mfma32b(ac[0], bq[u][j][2], acc[0][j]); acc[0][j] = mfma32b(ac[1], bq[u][j][1], acc[0][j]); acc[0][j] = mfma32b(ac[2], bq[u][j][0], acc[0][j]);
        acc[0][j] = mfma32b(ac[0], bq[u][j][1], acc[0][j]); acc[0][j] = mfma32b(ac[1], bq[u][j][0], acc[0][j]);
        acc[0][j] = mfma32b(ac[0], bq[u][j][0], acc[0][j]);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int p = 0; p < 3; ++p) ac[p] = an[p];
    }
  }
}

// te_head3's two products.  (1) logits: the A tile stays float32 in LDS and a lane cuts its 8 consecutive k into three planes when it reads
// them; B = planes 1, 2 of the n16 == 4 fragments: five partial products a3 b1 + a2 b2 + a2 b1 + a1 b2 + a1 b1 (dropped: a1 b3, 2^-17).
template <int NTW, int KG>
__device__ __forceinline__ void mma_f32a_s2(f32x16 (&acc)[NTW], const float* __restrict__ ldsA, int ldh, const float4* __restrict__ bp, const int (&nt)[NTW]) {
  const int lane = lane_id(), li = lane & 31, h = lane >> 5;
  const float* arow = ldsA + li * ldh + 8 * h;
  const uint4* bj[NTW];
  uint4 bc[NTW][2], bn[NTW][2];
#pragma unroll
  for (int j = 0; j < NTW; ++j) {
    bj[j] = reinterpret_cast<const uint4*>(bp) + ((size_t)nt[j] * KG) * 3 * 64 + lane;
    bc[j][0] = bj[j][0]; bc[j][1] = bj[j][64];
  }
  float4 xc0 = *reinterpret_cast<const float4*>(arow), xc1 = *reinterpret_cast<const float4*>(arow + 4);
#pragma unroll 1
  for (int m = 0; m < KG; ++m) {
    const int mn = min(m + 1, KG - 1);
#pragma unroll
    for (int j = 0; j < NTW; ++j) { bn[j][0] = bj[j][(size_t)mn * 3 * 64]; bn[j][1] = bj[j][((size_t)mn * 3 + 1) * 64]; }
    const float4 xn0 = *reinterpret_cast<const float4*>(arow + 16 * mn), xn1 = *reinterpret_cast<const float4*>(arow + 16 * mn + 4);
    uint4 a1, a2, a3;
    wg_split2(xc0.x, xc0.y, a1.x, a2.x, a3.x); wg_split2(xc0.z, xc0.w, a1.y, a2.y, a3.y);
    wg_split2(xc1.x, xc1.y, a1.z, a2.z, a3.z); wg_split2(xc1.z, xc1.w, a1.w, a2.w, a3.w);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < NTW; ++j) acc[j] = mfma32b(a3, bc[j][0], acc[j]);
#pragma unroll
    for (int j = 0; j < NTW; ++j) acc[j] = mfma32b(a2, bc[j][1], acc[j]);
#pragma unroll
    for (int j = 0; j < NTW; ++j) acc[j] = mfma32b(a2, bc[j][0], acc[j]);
#pragma unroll
    for (int j = 0; j < NTW; ++j) acc[j] = mfma32b(a1, bc[j][1], acc[j]);
#pragma unroll
    for (int j = 0; j < NTW; ++j) acc[j] = mfma32b(a1, bc[j][0], acc[j]);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < NTW; ++j) { bc[j][0] = bn[j][0]; bc[j][1] = bn[j][1]; }
    xc0 = xn0; xc1 = xn1;
  }
}
// (2) d h: two A planes in LDS (arow: this lane's row of plane 1, + 8 h; ps: elements between the planes), all three B planes: five partial
// products a1 b3 + a2 b2 + a2 b1 + a1 b2 + a1 b1.  What is lost is the A operand's third plane, 2^-18 (both A planes are rounded to nearest):
// a backward product - the recurrence does not amplify its error, the bars on the updates are 1e-4.  (Without a1 b3 - B truncated to two
// planes, 2^-17 - the hot-POI step of test_exact_forward_pass_is_far_inside_the_bar lands at 2.7e-6 instead of under 2e-6: 229 vs 250 us.)
template <int NTW, int KG>
__device__ __forceinline__ void mma_p2_s3(f32x16 (&acc)[NTW], const unsigned short* __restrict__ arow, int ps, const float4* __restrict__ bp, const int (&nt)[NTW], const int kg_run = KG) {
  // (kg_run <= KG: k groups that hold real bins - the padding bins' d logits are zero, their groups need not be multiplied)
  const int lane = lane_id();
  const uint4* bj[NTW];
  uint4 bc[NTW][3], bn[NTW][3];
#pragma unroll
  for (int j = 0; j < NTW; ++j) {
    bj[j] = reinterpret_cast<const uint4*>(bp) + ((size_t)nt[j] * KG) * 3 * 64 + lane;
#pragma unroll
    for (int p = 0; p < 3; ++p) bc[j][p] = bj[j][p * 64];
  }
  uint4 ac0 = *reinterpret_cast<const uint4*>(arow), ac1 = *reinterpret_cast<const uint4*>(arow + ps);
#pragma unroll 1
  for (int m = 0; m < kg_run; ++m) {
    const int mn = min(m + 1, kg_run - 1);
#pragma unroll
    for (int j = 0; j < NTW; ++j)
#pragma unroll
      for (int p = 0; p < 3; ++p) bn[j][p] = bj[j][((size_t)mn * 3 + p) * 64];
    const uint4 an0 = *reinterpret_cast<const uint4*>(arow + 16 * mn), an1 = *reinterpret_cast<const uint4*>(arow + ps + 16 * mn);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < NTW; ++j) acc[j] = mfma32b(ac0, bc[j][2], acc[j]);
#pragma unroll
    for (int j = 0; j < NTW; ++j) acc[j] = mfma32b(ac1, bc[j][1], acc[j]);
#pragma unroll
    for (int j = 0; j < NTW; ++j) acc[j] = mfma32b(ac1, bc[j][0], acc[j]);
#pragma unroll
    for (int j = 0; j < NTW; ++j) acc[j] = mfma32b(ac0, bc[j][1], acc[j]);
#pragma unroll
    for (int j = 0; j < NTW; ++j) acc[j] = mfma32b(ac0, bc[j][0], acc[j]);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < NTW; ++j)
#pragma unroll
      for (int p = 0; p < 3; ++p) bc[j][p] = bn[j][p];
    ac0 = an0; ac1 = an1;
  }
}

// -------------------------------------------------------------------------------------------------
// bookkeeping kernels
// -------------------------------------------------------------------------------------------------
// soff[k] = sum_{k' < k} (L_k' - 1); soff[n] = total packed rows.  te_len writes the step counts
// (one thread per sequence: the uidx -> off chain is two dependent loads), te_scan (one 1024-thread
// block, contiguous runs per thread) turns them into offsets in place.
__global__ __launch_bounds__(256) void te_len_kernel(TeArgs A) {
  const int k = blockIdx.x * 256 + threadIdx.x;
  if (k >= A.n_seq) return;
  const int u = A.uidx[k];
  const int L = A.off[u + 1] - A.off[u];
  A.soff[k] = A.predict ? L : (L > 0 ? L - 1 : 0);
}
// Hybrid recurrences (TeArgs.hyb): the split of the launch between the per-sequence kernels (leading sequences) and the 16-sequence tiles (the rest),
// chosen from the launch's own lengths at the end of te_scan (a thread of the scan holds exactly one tile: its 16 step counts and the steps in front
// of it).  Candidates are the tile boundaries hyb = 16 j: cost(j) = max(c_tile x longest sequence among k >= hyb, c_seq x (steps of the sequences
// k < hyb / workgroups left for them + a quarter of the longest)) with num_cu - tiles workgroups for the per-sequence kernel (each kernel's workgroup
// holds a CU).  Step costs measured under both kernels at once (1563-user launches, forced splits): forward 4.5 us per tile step,
// 2.6 us per sequence step; backward 2.9 / 1.35.  Ties go to the smaller j.  hyb_dev = {hyb_fwd, grid_fwd, hyb_bwd, grid_bwd}.
__device__ __forceinline__ void te_hybrid_choose(const TeArgs& A, int num_cu, int my_max, int my_pre, int total, int* s_max, int* s_pre, float* s_cost, int* s_arg,
                                                 const unsigned short* s_len) {
  const int n = A.n_seq, nt = (n + 15) / 16, tid = threadIdx.x;
  s_max[tid] = tid < nt ? my_max : 0; s_pre[tid] = tid < nt ? my_pre : total;
  __syncthreads();
  for (int o = 1; o < 1024; o <<= 1) {            // suffix maxima over the tiles
    const int v = (tid + o < 1024) ? s_max[tid + o] : 0;
    __syncthreads();
    s_max[tid] = max(s_max[tid], v);
    __syncthreads();
  }
  for (int pass = 0; pass < 2; ++pass) {
    const float c_tile = pass ? 2.9f : 4.5f, c_seq = pass ? 1.35f : 2.6f;
    float cost = 3.0e38f;
    if (tid <= nt) {
      const int hyb = min(16 * tid, n), ntile = nt - min(tid, nt);
      const int g = min(num_cu - ntile, hyb);
      const float t_tile = ntile > 0 ? c_tile * (float)s_max[min(tid, 1023)] : 0.f;
      if (hyb == 0) cost = t_tile;
      else if (g >= 1) {
        // the per-sequence kernel walks the leading sequences in snake order over g workgroups (k = j g + b on even legs, j g + g - 1 - b on odd ones):
        // its time is the heaviest workgroup's - the first, the last and the middle one are evaluated exactly from the step counts
        int worst = 0;
        const int bs[3] = {0, g - 1, g / 2};
#pragma unroll
        for (int q = 0; q < 3; ++q) {
          int load = 0;
          for (int j = 0;; ++j) {
            const int k = j * g + ((j & 1) ? g - 1 - bs[q] : bs[q]);
            if (k >= hyb) break;
            load += s_len[k];
          }
          worst = max(worst, load);
        }
        cost = fmaxf(t_tile, c_seq * (float)worst);
      }
    }
    s_cost[tid] = cost; s_arg[tid] = tid;
    __syncthreads();
    for (int o = 512; o > 0; o >>= 1) {
      if (tid < o && (s_cost[tid + o] < s_cost[tid] || (s_cost[tid + o] == s_cost[tid] && s_arg[tid + o] < s_arg[tid]))) { s_cost[tid] = s_cost[tid + o]; s_arg[tid] = s_arg[tid + o]; }
      __syncthreads();
    }
    if (tid == 0) {
      const int j = A.hyb_force > 0 ? min((A.hyb_force + 15) / 16, nt) : s_arg[0], hyb = min(16 * j, n), ntile = nt - min(j, nt);
      A.hyb_dev[2 * pass] = hyb; A.hyb_dev[2 * pass + 1] = max(1, min(num_cu - ntile, hyb));
    }
    __syncthreads();
  }
}

__global__ __launch_bounds__(1024) void te_scan_kernel(TeArgs A, int num_cu) {
  __shared__ int wtot[16];
  __shared__ int s_hmax[1024], s_hpre[1024], s_harg[1024];
  __shared__ float s_hcost[1024];
  __shared__ unsigned short s_hlen[TE_HYB_NMAX];      // step counts of the launch's sequences (hybrid recurrences)
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, n = A.n_seq;
  // exclusive scan of the step counts, 16384 sequences per pass: a thread holds 16 consecutive counts in registers (independent
  // loads; a load - store - load loop over soff serialised on the possible aliasing: 22 us for 12500 sequences), then a shuffle scan
  // inside each wave and over the 16 wave totals (two barriers instead of the twenty of a Hillis-Steele pass over LDS)
  int carry = 0, my_max = 0, my_pre = 0;
  for (int base = 0; base < n; base += 16384) {
    const int i0 = base + tid * 16;
    int v[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) v[u] = i0 + u < n ? A.soff[i0 + u] : 0;
    int s = 0;
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const int t = v[u];
      if (base == 0) { my_max = max(my_max, t); if (A.hyb && i0 + u < TE_HYB_NMAX) s_hlen[i0 + u] = (unsigned short)min(t, 65535); }
      v[u] = s; s += t;
    }
    int inc = s;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(inc, o, 64); inc += lane >= o ? t : 0; }
    if (lane == 63) wtot[w] = inc;
    __syncthreads();
    if (w == 0) {
      int t = lane < 16 ? wtot[lane] : 0;
#pragma unroll
      for (int o = 1; o < 16; o <<= 1) { const int t2 = __shfl_up(t, o, 64); t += lane >= o ? t2 : 0; }
      if (lane < 16) wtot[lane] = t;             // inclusive totals of the waves
    }
    __syncthreads();
    const int run = carry + inc - s + (w > 0 ? wtot[w - 1] : 0);
    if (base == 0) my_pre = run;                 // steps in front of this thread's 16 sequences (hybrid recurrences: one tile per thread)
#pragma unroll
    for (int u = 0; u < 16; ++u) if (i0 + u < n) A.soff[i0 + u] = run + v[u];
    carry += wtot[15];
    __syncthreads();
  }
  if (tid == 1023) {
    const int total = carry;
    A.soff[n] = total;
    if (A.cnt) { A.cnt[0] = (A.spatial ? 3 : 2) * (total + n); A.cnt[1] = 0; A.cnt[2] = 0; A.cnt[3] = 0; A.cnt[4] = 0; A.cnt[5] = 0; }   // slots, hot rows, hot chunks, touched rows, S rows, dx entries of the POI rows
  }
  if (A.hyb && n <= TE_HYB_NMAX) te_hybrid_choose(A, num_cu, my_max, my_pre, carry, s_hmax, s_hpre, s_hcost, s_harg, s_hlen);      // (uniform)
}

#ifndef TE_SEQ_PER_WAVE
#define TE_SEQ_PER_WAVE 4
#endif
// Sorted-scatter slots of one sequence (te_scatter.hip): 3 * (ns + 1) slots at 3 * (r0 + k):
// [0, L) POI ids of p, [L, 2L) negatives q, [2L, 3L) distance bins dp, rest sentinels.  Returns the
// literal occurrences of the two padding rows (their analytic multiplicity is added by the caller).
__device__ __forceinline__ void te_slots(const TeArgs& A, int k, int base, int L, int ns, int r0, int* pads_lt, int* pads_di) {
  const int lane = lane_id();
  const int nsec = A.spatial ? 3 : 2;      // the plain GRU has no distance-bin section (n_dist == -1)
  const int S0 = nsec * (r0 + k), nslot = nsec * (ns + 1), sentinel = A.n_item + 1 + A.n_dist + 1;
  int plt = 0, pdi = 0;
  for (int e0 = 0; e0 < nslot; e0 += 64) {
    const int e = e0 + lane;
    int key = sentinel, code = 0;
    bool is_plt = false, is_pdi = false;
    if (e < nslot && e < nsec * L) {
      const int sec = e / L, j = e - sec * L;
      code = r0 + j;
      if (sec == 0) { key = A.p[base + j]; code |= (j < ns ? TE_ENT_DX : 0) | (j >= 1 ? TE_ENT_GH : 0); is_plt = key == A.n_item; }
      else if (sec == 1) { key = A.q[base + j]; code |= (j >= 1 ? (TE_ENT_GH | TE_ENT_NEG) : 0); is_plt = key == A.n_item; }
      else { const int b = A.dp[base + j]; key = A.n_item + 1 + b; code |= (j < ns ? TE_ENT_DX : 0); is_pdi = b == A.n_dist; }
    }
    if (e < nslot) { A.keys0[S0 + e] = key; A.code[S0 + e] = code; A.slot_seq[S0 + e] = k; }
    if (A.ppoi && e < L && e < ns) A.pmark[key] = 1;          // POI rows that are step inputs (dx entries): rows of S (te_passign: S row + 1)
    plt += __builtin_popcountll(__ballot(is_plt));
    pdi += __builtin_popcountll(__ballot(is_pdi));
  }
  *pads_lt = plt; *pads_di = pdi;
}

// one wavefront per sequence: packed row -> CSR position maps, sorted-scatter slots, and the analytic
// part of the padding rows' bookkeeping (every sequence shorter than len_max touches the padding rows
// 2*(len_max-L) / (len_max-L) times: public/GRU_Spatial.py:202-203), aggregated per workgroup because
// same-address device atomics serialise.
__global__ __launch_bounds__(TE_BLOCK) void te_rowmap_kernel(TeArgs A) {
  __shared__ int s_pad[POI_NWAVE][4];
  const int w = wave_id(), lane = lane_id();
  int m_lt = 0, n_lt = 0, m_di = 0, n_di = 0;
  // the headers of the wave's sequences: lane i fetches sequence i's (uidx -> off is a dependent chain: once per wave, not per sequence)
  const int kbase = (blockIdx.x * POI_NWAVE + w) * TE_SEQ_PER_WAVE;
  int hb = 0, hL = 0, hr = 0;
  if (lane < TE_SEQ_PER_WAVE && kbase + lane < A.n_seq) {
    const int u = A.uidx[kbase + lane];
    hb = A.off[u]; hL = A.off[u + 1] - hb; hr = A.soff[kbase + lane];
  }
  for (int i = 0; i < TE_SEQ_PER_WAVE; ++i) {
    const int k = kbase + i;
    if (k >= A.n_seq) break;
    const int base = __shfl(hb, i, 64), L = __shfl(hL, i, 64), ns = A.predict ? L : (L > 0 ? L - 1 : 0), r0 = __shfl(hr, i, 64);
    for (int t = lane; t < ns; t += 64) {
      A.row_src[r0 + t] = base + t; A.row_t[r0 + t] = t;
      A.row_p[r0 + t] = A.p[base + t];                       // table rows of the step's input: the GEMMs gather them
      if (A.spatial) A.row_dp[r0 + t] = A.dp[base + t];      // straight into their LDS tiles (no packed copy of X)
      // target bins of the step (positive | negative << 16) for te_head: one load instead of a dependent chain
      if (A.spatial && !A.predict) A.row_ab[r0 + t] = A.dp[base + t + 1] | (A.dq[base + t + 1] << 16);
      if (A.efuse) A.row_pq[r0 + t] = make_int2(A.p[base + t + 1], A.q[base + t + 1]);      // the rows of E = lt[p'] - lt[q']: te_head3 gathers them itself
    }
    if (A.predict) continue;
    int plt, pdi;
    te_slots(A, k, base, L, ns, r0, &plt, &pdi);
    // literal occurrences of a padding id are counted by its row segment; only the analytic part goes here
    m_lt += 2 * (A.len_max - L); n_lt += (2 * (A.len_max - L) + plt) > 0;
    m_di += (A.len_max - L); n_di += ((A.len_max - L) + pdi) > 0;
  }
  if (A.predict) return;
  if (lane == 0) { s_pad[w][0] = m_lt; s_pad[w][1] = n_lt; s_pad[w][2] = m_di; s_pad[w][3] = n_di; }
  __syncthreads();
  if (threadIdx.x < (A.spatial ? 4 : 2)) {
    const int t = s_pad[0][threadIdx.x] + s_pad[1][threadIdx.x] + s_pad[2][threadIdx.x] + s_pad[3][threadIdx.x];
    int* dst = threadIdx.x == 0 ? A.mult_lt + A.n_item : threadIdx.x == 1 ? A.nseq_lt + A.n_item
             : threadIdx.x == 2 ? A.mult_di + A.n_dist : A.nseq_di + A.n_dist;
    if (t) atomicAdd(dst, t);
  }
}

// Exact forward table over the launch's step-input POIs only (TeArgs.xcomp): the rows te_slots marked (pmark) are ranked in row order -
// xidx[lt row] = table row, xlist[table row] = lt row, xcnt = number of rows - by one scan in two small kernels (per-block counts, then
// block prefix + ballot ranks); te_gather translates the steps' POI ids (row_pc).  45 k of the 100 k Gowalla rows per 12500-user launch:
// te_gemmx multiplies and writes less than half of the float64 table (110 -> 55 us) and the table fits the 256 MB cache behind L2.
#define TE_XBLK 256
__device__ __forceinline__ int te_xper(int N) { return (((N + TE_XBLK - 1) / TE_XBLK) + 255) & ~255; }
__global__ __launch_bounds__(256) void te_xcount_kernel(TeArgs A) {
  __shared__ int red[4];
  const int N = A.n_item + 1, per = te_xper(N), b0 = blockIdx.x * per;
  int c = 0;
  for (int i = b0 + threadIdx.x; i < min(N, b0 + per); i += 256) c += A.pmark[i] != 0 ? 1 : 0;
  for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
  if (lane_id() == 0) red[wave_id()] = c;
  __syncthreads();
  if (threadIdx.x == 0) A.xblk[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}
__global__ __launch_bounds__(256) void te_xassign_kernel(TeArgs A) {
  __shared__ int s_w[4];
  __shared__ int s_base;
  const int N = A.n_item + 1, per = te_xper(N), tid = threadIdx.x, lane = lane_id(), w = wave_id(), b0 = blockIdx.x * per;
  {
    int v = tid < (int)blockIdx.x ? A.xblk[tid] : 0;      // (TE_XBLK == block size: one earlier block per thread)
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    if (lane == 0) s_w[w] = v;
    __syncthreads();
    if (tid == 0) s_base = (s_w[0] + s_w[1]) + (s_w[2] + s_w[3]);
    __syncthreads();
  }
  int base = s_base;
  const unsigned long long below = (1ull << lane) - 1ull;
  for (int i0 = b0; i0 < min(N, b0 + per); i0 += 256) {
    const int i = i0 + tid;
    const bool f = i < N && A.pmark[i] != 0;
    const unsigned long long m = __ballot(f);
    __syncthreads();
    if (lane == 0) s_w[w] = __builtin_popcountll(m);
    __syncthreads();
    int wb = 0;
    for (int j = 0; j < w; ++j) wb += s_w[j];
    if (f) { const int idx = base + wb + __builtin_popcountll(m & below); A.xidx[i] = idx; A.xlist[idx] = i; }
    base += (s_w[0] + s_w[1]) + (s_w[2] + s_w[3]);
  }
  if (blockIdx.x == gridDim.x - 1 && tid == 0) A.xcnt[0] = base;
}

// E[r] = lt[p_{t+1}] - lt[q_{t+1}] (the BPR difference row of step t).  LPR lanes per row, float4 per lane.
// The step's input x_t = [lt[p_t] | di[dp_t]] is NOT materialised: te_gemm_nt / te_wgrad gather those table
// rows straight into their LDS tiles through row_p / row_dp.
template <int D>
__global__ __launch_bounds__(TE_BLOCK) void te_gather_kernel(TeArgs A) {
  constexpr int LPR = D / 4;                    // lanes per table row
  constexpr int RPB = TE_BLOCK / LPR;           // rows per block pass
  const int T = A.soff[A.n_seq];
  const int sub = threadIdx.x / LPR, c = (threadIdx.x % LPR) * 4;
  if (A.efuse) {      // (round 6) no E rows: the training head gathers lt[p'] / lt[q'] itself (row_pq, te_rowmap) - only the compact table's ids are left here
    for (int r = blockIdx.x * TE_BLOCK + threadIdx.x; A.xcomp && r < T; r += gridDim.x * TE_BLOCK) A.row_pc[r] = A.xidx[min((unsigned)A.row_p[r], (unsigned)A.n_item)];
  } else
  for (int r = blockIdx.x * RPB + sub; r < T; r += gridDim.x * RPB) {
    const int s = A.row_src[r];
    const float4 a = ld4t(A.lt, (size_t)A.p[s + 1] * D + c, A.lt_f16);
    const float4 b = ld4t(A.lt, (size_t)A.q[s + 1] * D + c, A.lt_f16);
    *reinterpret_cast<float4*>(A.E + (size_t)r * D + c) = make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w);
    if (A.xcomp && c == 0) A.row_pc[r] = A.xidx[min((unsigned)A.row_p[r], (unsigned)A.n_item)];      // the step's row of the compact forward table
  }
  // the spare packed row T: finished sequences of a tile read its table row unconditionally (te_rec_fwdx / te_rec_fwdd) - row 0 of the compact
  // table always exists, whatever an earlier launch with another n_item left in this buffer
  if (A.xcomp && blockIdx.x == 0 && threadIdx.x == 0) A.row_pc[T] = 0;
}

// Plain GRU + BPR head (public/GRU.py:349-357): u = h_t . (x_p' - x_q'), loss -= log sigmoid(u),
// g = -sigmoid(-u), DH = g * E.  D/4 lanes per packed row, float4 per lane (HBM-bound).
template <int D>
__global__ __launch_bounds__(TE_BLOCK) void te_bpr_head_kernel(TeArgs A) {
  constexpr int LPR = D / 4, RPB = TE_BLOCK / LPR;
  const int T = A.soff[A.n_seq];
  const int sub = threadIdx.x / LPR, c = (threadIdx.x % LPR) * 4;
  for (int r0 = blockIdx.x * RPB; r0 < T; r0 += gridDim.x * RPB) {
    const int r = min(r0 + sub, T - 1);
    const float4 h = *reinterpret_cast<const float4*>(A.H + (size_t)r * D + c);
    const float4 e = *reinterpret_cast<const float4*>(A.E + (size_t)r * D + c);
    float u = (h.x * e.x + h.y * e.y) + (h.z * e.z + h.w * e.w);
#pragma unroll
    for (int o = 1; o < LPR; o <<= 1) u += __shfl_xor(u, o, 64);
    const float g = -sigmoidf_(-u);
    if (r0 + sub < T) {
      *reinterpret_cast<float4*>(A.DH + (size_t)r * D + c) = make_float4(g * e.x, g * e.y, g * e.z, g * e.w);
      if (c == 0) { A.gcoef[r] = g; A.rowloss[2 * (size_t)r] = 0.f; A.rowloss[2 * (size_t)r + 1] = log_sigmoidf_(u); }
    }
  }
}

// -------------------------------------------------------------------------------------------------
// te_gemm_nt: C[r][n] = sum_k A[r][k] * B[n][k] (+ bias[n]) for the packed rows r < T: 128 x 128 output
// tiles (persistent grid, n fastest so that neighbouring workgroups share the A rows in L2), both
// operands staged in 32-wide k-chunks through LDS as [row][k] tiles (row pitch 36 floats: the float4
// operand reads of 16 lanes hit 16 different bank groups).  Waves form a 2 x 2 grid of 64 x 64
// sub-tiles.  Global loads run two chunks ahead in two register sets (written to LDS one iteration
// later), LDS operand reads one k-group ahead of the MFMAs; rows >= T are clamped on load and stored
// to the spare row T.  Used for G = X . ui^T + bi (te_gemm_ax) and dx = DA . ui (te_gemm_dx).
// -------------------------------------------------------------------------------------------------
#define NT_LDK 36
// GATHER: the A operand is not a packed matrix but x_r = [tab0[idx0[r]] | tab1[idx1[r]]] (Dg columns each):
// the embedding gather of the training step (lt[p_t] | di[dp_t]) happens here, straight into the LDS tile.
struct NtArgs {
  const float* A; int lda;            // packed A (GATHER == false)
  const float *tab0, *tab1; const int *idx0, *idx1; int Dg;      // gathered A
  const float* B; int ldb; float* C; int ldc; const float* bias; const int* Tptr; int N, K;
  const float* ztab; const int* zidx;         // ZADD: C[r][:] += ztab[zidx[r]][:]   (ztab rows of N floats)
};
template <bool BIAS, bool GATHER, bool F16 = false>      // F16: tab0 (the POI table) holds IEEE half
__global__ __launch_bounds__(TE_BLOCK, 2) void te_gemm_nt_kernel(NtArgs P) {
  __shared__ __align__(16) float As[2][128][NT_LDK];
  __shared__ __align__(16) float Bs[2][128][NT_LDK];
  __shared__ int s_idx[2][128];
  const float* __restrict__ Ag = P.A; const float* __restrict__ Bg = P.B; float* __restrict__ C = P.C;
  const float* __restrict__ bias = P.bias;
  const int lda = P.lda, ldb = P.ldb, ldc = P.ldc, N = P.N, K = P.K;
  const int T = *P.Tptr;
  const int lane = lane_id(), w = wave_id(), li = lane & 31, h = lane >> 5, tid = threadIdx.x;
  const int wm = (w >> 1) * 64, wn = (w & 1) * 64;
  const int ntl = (N + 127) / 128, mtl = (T + 127) / 128, nchunk = K / 32;
  // XCD-aware tile order: workgroup b runs on XCD b % 8 (round-robin dispatch) and every XCD has its own
  // L2, so the n-tiles of one 128-row block are handed to neighbouring workgroups of the SAME XCD: the A
  // rows then come from HBM once instead of once per n-tile.  (gridDim.x is a multiple of 8.)
  const int xcd = blockIdx.x & 7, gx = gridDim.x >> 3;
  for (int L = blockIdx.x >> 3;; L += gx) {
    const int tm = (L / ntl) * 8 + xcd;
    if (tm >= mtl) break;
    const int r0 = tm * 128, n0 = (L % ntl) * 128;
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    float4 ra0[4], rb0[4], ra1[4], rb1[4];
    __syncthreads();                       // previous tile's MFMAs are done with both LDS buffers (and s_idx)
    if (GATHER) {
      if (tid < 128) {
        const int r = min(r0 + tid, T - 1);
        s_idx[0][tid] = P.idx0[r]; s_idx[1][tid] = P.idx1 ? P.idx1[r] : 0;
      }
      __syncthreads();
    }
    auto gload = [&](int kc, float4 (&ra)[4], float4 (&rb)[4]) {
      const int half = (GATHER && kc * 32 >= P.Dg) ? 1 : 0;                 // chunk-uniform: which table
      const float* tab = half ? P.tab1 : P.tab0;
      const int coff = kc * 32 - half * P.Dg;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const int e = tid + s * TE_BLOCK, row = e >> 3, c = (e & 7) * 4;
        if (GATHER) {
          if (F16 && !half) ra[s] = ld4(reinterpret_cast<const __half*>(P.tab0) + (size_t)s_idx[0][row] * P.Dg + coff + c);
          else ra[s] = *reinterpret_cast<const float4*>(tab + (size_t)s_idx[half][row] * P.Dg + coff + c);
        }
        else ra[s] = *reinterpret_cast<const float4*>(Ag + (size_t)min(r0 + row, T - 1) * lda + kc * 32 + c);
        rb[s] = *reinterpret_cast<const float4*>(Bg + (size_t)min(n0 + row, N - 1) * ldb + kc * 32 + c);
      }
    };
    auto lstore = [&](int buf, const float4 (&ra)[4], const float4 (&rb)[4]) {
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const int e = tid + s * TE_BLOCK, row = e >> 3, c = (e & 7) * 4;
        // (component-wise: a whole-struct copy of the HIP float4 keeps the staging arrays in scratch)
        *reinterpret_cast<float4*>(&As[buf][row][c]) = make_float4(ra[s].x, ra[s].y, ra[s].z, ra[s].w);
        *reinterpret_cast<float4*>(&Bs[buf][row][c]) = make_float4(rb[s].x, rb[s].y, rb[s].z, rb[s].w);
      }
    };
    auto mma = [&](int buf) {
      float4 a[2][2], b[2][2];
#pragma unroll
      for (int i = 0; i < 2; ++i) a[0][i] = *reinterpret_cast<const float4*>(&As[buf][wm + 32 * i + li][4 * h]);
#pragma unroll
      for (int j = 0; j < 2; ++j) b[0][j] = *reinterpret_cast<const float4*>(&Bs[buf][wn + 32 * j + li][4 * h]);
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        if (m + 1 < 4) {
#pragma unroll
          for (int i = 0; i < 2; ++i) a[(m + 1) & 1][i] = *reinterpret_cast<const float4*>(&As[buf][wm + 32 * i + li][8 * (m + 1) + 4 * h]);
#pragma unroll
          for (int j = 0; j < 2; ++j) b[(m + 1) & 1][j] = *reinterpret_cast<const float4*>(&Bs[buf][wn + 32 * j + li][8 * (m + 1) + 4 * h]);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            const float4 x = a[m & 1][i], y = b[m & 1][j];
            acc[i][j] = mfma32(x.x, y.x, acc[i][j]);
            acc[i][j] = mfma32(x.y, y.y, acc[i][j]);
            acc[i][j] = mfma32(x.z, y.z, acc[i][j]);
            acc[i][j] = mfma32(x.w, y.w, acc[i][j]);
          }
        __builtin_amdgcn_sched_barrier(0);
      }
    };
    gload(0, ra0, rb0); lstore(0, ra0, rb0);
    if (1 < nchunk) gload(1, ra0, rb0);
    __syncthreads();
    for (int kc = 0; kc < nchunk; kc += 2) {
      if (kc + 2 < nchunk) gload(kc + 2, ra1, rb1);
      mma(0);
      if (kc + 1 < nchunk) lstore(1, ra0, rb0);
      __syncthreads();
      if (kc + 1 >= nchunk) break;
      if (kc + 3 < nchunk) gload(kc + 3, ra0, rb0);
      mma(1);
      if (kc + 2 < nchunk) lstore(0, ra1, rb1);
      __syncthreads();
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = n0 + wn + 32 * j + li;
      if (n0 + wn + 32 * j >= N) continue;          // wave-uniform (N % 32 == 0)
      const float bv = BIAS ? bias[col] : 0.f;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = min(r0 + wm + 32 * i + c_row(r, lane), T);
          C[(size_t)row * ldc + col] = acc[i][j][r] + bv;
        }
    }
  }
}

// te_gemm_ntk: the same product with K (and, when gathering, the table width DG) known at compile time, for
// K >= 128.  Differences that matter (measured: ax 1.79 -> see DESIGN.md):
//  * the chunk loop is fully unrolled, so one tile is straight-line code and the compiler's s_waitcnt
//    vmcnt(n) values are exact - the runtime-K loop above has branches around its loads, and every
//    join there degrades to vmcnt(0);
//  * the operand pipeline runs ACROSS tiles: the last two stages of a tile already fetch chunks 0 and 1
//    of the workgroup's next tile (and stage 0 fetches its gather indices), so neither the load latency
//    at the start of a tile nor the 64 C stores per lane at its end are exposed (on gfx9 stores count
//    in vmcnt too; the loads of the next tile are issued BEFORE them and vmcnt retires in order);
//  * the epilogue is branch-free: columns beyond N (N % 128 != 0) go to the spare row T.
//  * ZADD: the epilogue adds row zidx[r] of a small table instead of a bias (te_gemm_ax: the distance-bin half
//    of the step input only takes n_dist + 1 values, so its product with ui is a table, see te_ztab_kernel).
// SP3 (round 4): split products - every staged float4 (four consecutive k of a row) is cut into three bf16 planes on its way to LDS (8 bytes per
// plane), a lane's fragment - eight consecutive k of its row - is one 16-byte read per plane, six partial products on v_mfma_f32_32x32x16_bf16;
// 16-k stages in the same 72 KB of LDS ([buffer][plane][row][16 + 8 pad] bf16).
template <bool BIAS, bool GATHER, int K, int DG, int N, int LDB = K, int LDC = N, bool ZADD = false, bool F16 = false, bool SP3 = false>
__global__ __launch_bounds__(TE_BLOCK, 2) void te_gemm_ntk_kernel(NtArgs P) {
  constexpr int KC = SP3 ? 16 : 32, NCH = K / KC, ldb = LDB, ldc = LDC;       // B is N x K (row pitch LDB), C is T x N (row pitch LDC)
  constexpr int F4S = 128 * (KC / 4) / TE_BLOCK, LPK = KC / 4, LDP = KC + 8;     // float4 per thread, operand and stage; float4 per row and stage; plane row pitch (bf16)
  static_assert(K % 64 == 0 && NCH >= 4, "te_gemm_ntk: K must be a multiple of 64, >= 128");
  __shared__ __align__(16) float As[2][128][NT_LDK];
  __shared__ __align__(16) float Bs[2][128][NT_LDK];
  // gather indices of 128 rows x 2 slots.  LDS is the occupancy limit here (two workgroups per CU fit only up
  // to ~74.8 KB each - a third KB of indices halves the occupancy and costs 25%), so there is ONE KB of them:
  //  * two tables (DG < K): slot = table.  Table 0 is read by the fetches of chunks < NCH/2, i.e. in stages
  //    NCH-2 .. NCH/2-3 (wrapping over the tile boundary), table 1 in stages NCH/2-2 .. NCH-3; the next tile's
  //    indices are written in the gaps (stage NCH/2-1 and stage NCH-2), a barrier away from any reader.
  //  * one table (DG == K): no gap, slot = tile parity.
  constexpr bool ONE_TAB = (DG == K);
  __shared__ int s_idx[2][128];
  __shared__ int s_z[ZADD ? 2 : 1][128];       // ZADD: table rows of the tile's 128 rows, by tile parity
  const float* __restrict__ Ag = P.A; const float* __restrict__ Bg = P.B; float* __restrict__ C = P.C;
  const float* __restrict__ bias = P.bias;
  const int lda = P.lda;
  const int T = *P.Tptr;
  const int lane = lane_id(), w = wave_id(), li = lane & 31, h = lane >> 5, tid = threadIdx.x;
  const int wm = (w >> 1) * 64, wn = (w & 1) * 64;
  constexpr int ntl = (N + 127) / 128;
  const int mtl = (T + 127) / 128;
  const int xcd = blockIdx.x & 7, gx = gridDim.x >> 3;          // XCD-aware tile order, as above
  int L = blockIdx.x >> 3;
  if ((L / ntl) * 8 + xcd >= mtl) return;
  int r0 = ((L / ntl) * 8 + xcd) * 128, n0 = (L % ntl) * 128, par = 0;
  const int irow = tid & 127;                                   // (both halves of the workgroup: duplicates are harmless)
  const int* __restrict__ idx0 = P.idx0; const int* __restrict__ idx1 = ONE_TAB ? P.idx0 : P.idx1;
  float4 ra[2][F4S], rb[2][F4S];
  unsigned short* Ap = reinterpret_cast<unsigned short*>(&As[0][0][0]);      // SP3: [buffer][plane][128][LDP]
  unsigned short* Bp = reinterpret_cast<unsigned short*>(&Bs[0][0][0]);
  static_assert(!SP3 || 2 * 3 * 128 * (16 + 8) * 2 <= (int)sizeof(float) * 2 * 128 * NT_LDK, "te_gemm_ntk: the planes fit the float32 tiles' LDS");
  // chunk kc of the tile at (tr0, tn0), whose gather indices are in s_idx[tp], -> register set `set`
  auto gload = [&](int set, int tr0, int tn0, int tp, int kc) {
    const int half = (GATHER && kc * KC >= DG) ? 1 : 0;
    const float* tab = half ? P.tab1 : P.tab0;
    const int coff = kc * KC - half * DG;
#pragma unroll
    for (int s = 0; s < F4S; ++s) {
      const int e = tid + s * TE_BLOCK, row = e / LPK, c = (e % LPK) * 4;
      if (GATHER) {
        if (F16 && !half) ra[set][s] = ld4(reinterpret_cast<const __half*>(P.tab0) + (size_t)s_idx[ONE_TAB ? tp : 0][row] * DG + coff + c);
        else ra[set][s] = *reinterpret_cast<const float4*>(tab + (size_t)s_idx[ONE_TAB ? tp : half][row] * DG + coff + c);
      }
      else ra[set][s] = *reinterpret_cast<const float4*>(Ag + (size_t)min(tr0 + row, T - 1) * lda + kc * KC + c);
      rb[set][s] = *reinterpret_cast<const float4*>(Bg + (size_t)min(tn0 + row, N - 1) * ldb + kc * KC + c);
    }
  };
  auto lstore = [&](int buf, int set) {
#pragma unroll
    for (int s = 0; s < F4S; ++s) {
      const int e = tid + s * TE_BLOCK, row = e / LPK, c = (e % LPK) * 4;
      if constexpr (SP3) {
        uint2 pa[3], pb[3];
        wg_split2(ra[set][s].x, ra[set][s].y, pa[0].x, pa[1].x, pa[2].x); wg_split2(ra[set][s].z, ra[set][s].w, pa[0].y, pa[1].y, pa[2].y);
        wg_split2(rb[set][s].x, rb[set][s].y, pb[0].x, pb[1].x, pb[2].x); wg_split2(rb[set][s].z, rb[set][s].w, pb[0].y, pb[1].y, pb[2].y);
#pragma unroll
        for (int p = 0; p < 3; ++p) {
          *reinterpret_cast<uint2*>(Ap + ((buf * 3 + p) * 128 + row) * LDP + c) = pa[p];
          *reinterpret_cast<uint2*>(Bp + ((buf * 3 + p) * 128 + row) * LDP + c) = pb[p];
        }
      } else {
      *reinterpret_cast<float4*>(&As[buf][row][c]) = make_float4(ra[set][s].x, ra[set][s].y, ra[set][s].z, ra[set][s].w);
      *reinterpret_cast<float4*>(&Bs[buf][row][c]) = make_float4(rb[set][s].x, rb[set][s].y, rb[set][s].z, rb[set][s].w);
      }
    }
  };
  f32x16 acc[2][2];
  auto mma = [&](int buf) {
    if constexpr (SP3) {
      uint4 pa[2][3], pb[2][3];
#pragma unroll
      for (int p = 0; p < 3; ++p)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          pa[i][p] = *reinterpret_cast<const uint4*>(Ap + ((buf * 3 + p) * 128 + wm + 32 * i + li) * LDP + 8 * h);
          pb[i][p] = *reinterpret_cast<const uint4*>(Bp + ((buf * 3 + p) * 128 + wn + 32 * i + li) * LDP + 8 * h);
        }
#pragma unroll
      for (int t = 0; t < 6; ++t) {
        constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[i][j] = mfma32b(pa[i][PA[t]], pb[j][PB[t]], acc[i][j]);
      }
      return;
    }
    float4 a[2][2], b[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i) a[0][i] = *reinterpret_cast<const float4*>(&As[buf][wm + 32 * i + li][4 * h]);
#pragma unroll
    for (int j = 0; j < 2; ++j) b[0][j] = *reinterpret_cast<const float4*>(&Bs[buf][wn + 32 * j + li][4 * h]);
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      if (m + 1 < 4) {
#pragma unroll
        for (int i = 0; i < 2; ++i) a[(m + 1) & 1][i] = *reinterpret_cast<const float4*>(&As[buf][wm + 32 * i + li][8 * (m + 1) + 4 * h]);
#pragma unroll
        for (int j = 0; j < 2; ++j) b[(m + 1) & 1][j] = *reinterpret_cast<const float4*>(&Bs[buf][wn + 32 * j + li][8 * (m + 1) + 4 * h]);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const float4 x = a[m & 1][i], y = b[m & 1][j];
          acc[i][j] = mfma32(x.x, y.x, acc[i][j]);
          acc[i][j] = mfma32(x.y, y.y, acc[i][j]);
          acc[i][j] = mfma32(x.z, y.z, acc[i][j]);
          acc[i][j] = mfma32(x.w, y.w, acc[i][j]);
        }
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  // prologue: LDS buffer 0 <- chunk 0, register set 1 <- chunk 1 (in flight): the state every tile starts from
  if (GATHER || ZADD) {
    if (GATHER) s_idx[0][irow] = idx0[min(r0 + irow, T - 1)];
    if (GATHER && !ONE_TAB) s_idx[1][irow] = idx1[min(r0 + irow, T - 1)];
    if (ZADD) s_z[0][irow] = P.zidx[min(r0 + irow, T - 1)];
    __syncthreads();
  }
  gload(0, r0, n0, 0, 0); lstore(0, 0); gload(1, r0, n0, 0, 1);
  __syncthreads();
  for (;;) {
    // the workgroup's next tile; past the end the current one is fetched again (harmless, never used)
    const int Ln = L + gx, tmn = (Ln / ntl) * 8 + xcd;
    const bool more = tmn < mtl;
    const int r0n = more ? tmn * 128 : r0, n0n = more ? (Ln % ntl) * 128 : n0;
    int nidx0 = 0, nidx1 = 0;
    if (GATHER) { nidx0 = idx0[min(r0n + irow, T - 1)]; if (!ONE_TAB) nidx1 = idx1[min(r0n + irow, T - 1)]; }
    int nzi = 0;
    if (ZADD) nzi = P.zidx[min(r0n + irow, T - 1)];
    float bvj[2];                            // fetched here so that the epilogue never waits on a load
#pragma unroll
    for (int j = 0; j < 2; ++j) bvj[j] = BIAS ? bias[min(n0 + wn + 32 * j + li, N - 1)] : 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
#pragma unroll
    for (int kc = 0; kc < NCH; ++kc) {
      // stage kc: fetch chunk kc + 2 (set kc & 1 is free: chunk kc went to LDS at the end of stage kc - 1),
      // multiply chunk kc, then move chunk kc + 1 from its registers to the other LDS buffer
      if (kc + 2 < NCH) gload(kc & 1, r0, n0, par, kc + 2);
      else gload(kc & 1, r0n, n0n, par ^ 1, kc + 2 - NCH);
      mma(kc & 1);
      if (GATHER && ONE_TAB && kc == 1) s_idx[par ^ 1][irow] = nidx0;        // read from stage NCH - 2 on
      if (GATHER && !ONE_TAB && kc == NCH / 2 - 1) s_idx[0][irow] = nidx0;
      if (GATHER && !ONE_TAB && kc == NCH - 2) s_idx[1][irow] = nidx1;
      if (ZADD && kc == 1) s_z[(par ^ 1) & (ZADD ? 1 : 0)][irow] = nzi;          // read by the NEXT tile's epilogue
      lstore((kc + 1) & 1, (kc + 1) & 1);
      if (kc == NCH - 1) {
        // C tile.  One base pointer per 32x32 block and compile-time row offsets; the sched_barrier keeps
        // the address arithmetic HERE (hoisted to the top of the tile it spills).  Rows past T of the last
        // tile land in the spare rows of C; columns past N (N % 128 != 0) are sent there too.
        __builtin_amdgcn_sched_barrier(0);
        if (ZADD) {
          // all table loads of the tile are issued before the first store (one L2 round trip per tile)
          float zr[2][2][16];
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            int zi[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) zi[r] = s_z[par & (ZADD ? 1 : 0)][wm + 32 * i + 4 * h + (r & 3) + 8 * (r >> 2)];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
              const int cc = min(n0 + wn + 32 * j + li, N - 1);
#pragma unroll
              for (int r = 0; r < 16; ++r) zr[i][j][r] = P.ztab[(size_t)zi[r] * N + cc];
            }
          }
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            const int col = n0 + wn + 32 * j + li;
            const bool cok = col < N;
            const int cc = cok ? col : N - 1;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
              float* cb = C + (size_t)(cok ? r0 + wm + 32 * i + 4 * h : T) * ldc + cc;
#pragma unroll
              for (int r = 0; r < 16; ++r) cb[((r & 3) + 8 * (r >> 2)) * ldc] = acc[i][j][r] + zr[i][j][r];
            }
          }
        } else {
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            const int col = n0 + wn + 32 * j + li;
            const bool cok = col < N;
            const int cc = cok ? col : N - 1;
            const float bv = bvj[j];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
              float* cb = C + (size_t)(cok ? r0 + wm + 32 * i + 4 * h : T) * ldc + cc;
#pragma unroll
              for (int r = 0; r < 16; ++r) cb[((r & 3) + 8 * (r >> 2)) * ldc] = acc[i][j][r] + bv;
            }
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      __syncthreads();
    }
    if (!more) break;
    L = Ln; r0 = r0n; n0 = n0n; par ^= 1;
  }
}

// ztab[b][n] = bi[n] + sum_c di[b][c] * ui[n][D + c]: the distance-bin half of x_t . ui^T (+ bias) for each of the
// n_dist + 1 bins.  x_t = [lt[p_t] | di[dp_t]] takes only that many values in its second half, so te_gemm_ax
// multiplies the POI half alone (K = D instead of 2D) and adds row dp_t of this table in its epilogue; in the
// same way the backward pass needs only per-bin sums of DA (te_dsum in te_scatter.hip) for d di and the
// di half of d ui, instead of the di halves of dx = DA . ui and of d ui = DA^T . x.
__global__ __launch_bounds__(1024) void te_ztab_kernel(TeArgs A, float* __restrict__ ztab) {
  __shared__ float drow[256];
  const int D = A.dim, XW = A.xw, b = blockIdx.x, n = threadIdx.x;
  if (n < D) drow[n] = A.di[(size_t)b * D + n];
  __syncthreads();
  if (n >= 3 * D) return;
  const float* u = A.ui + (size_t)n * XW + D;
  float z0 = A.bi[n], z1 = 0.f, z2 = 0.f, z3 = 0.f;
  for (int c = 0; c < D; c += 4) {
    const float4 uv = *reinterpret_cast<const float4*>(u + c);
    z0 = fmaf(drow[c], uv.x, z0); z1 = fmaf(drow[c + 1], uv.y, z1); z2 = fmaf(drow[c + 2], uv.z, z2); z3 = fmaf(drow[c + 3], uv.w, z3);
  }
  // forward-table mode: gate-interleaved columns ([c][gate]) - a lane of te_rec_fwd16 reads z | r | c of its column with one 12-byte load
  const int o = A.fwd_tab ? 3 * (n % D) + n / D : n;
  ztab[(size_t)b * 3 * D + o] = (z0 + z1) + (z2 + z3);
}

// uiP[3 c + g][k] = ui[g D + c][k], k < D: the POI half of ui with gate-interleaved rows, B operand of the forward table's product
__global__ __launch_bounds__(256) void te_uiperm_kernel(TeArgs A) {
  const int D = A.dim, XW = A.xw;
  for (int e = blockIdx.x * 256 + threadIdx.x; e < 3 * D * D; e += gridDim.x * 256) {
    const int n = e / D, k = e % D, g = n % 3, c = n / 3;
    A.uiP[e] = A.ui[(size_t)(g * D + c) * XW + k];
  }
}

// iota[i] = i for i < n, iota[n] = n (the row count te_gemm_ntk reads through its T pointer): the forward table's identity gather
__global__ __launch_bounds__(256) void te_iota_kernel(int* __restrict__ buf, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i <= n) buf[i] = i;
}
void launch_te_iota(int* buf, int n, hipStream_t st) { hipLaunchKernelGGL(te_iota_kernel, dim3(n / 256 + 1), dim3(256), 0, st, buf, n); }

// uiT[c][r] = ui[r][c]   (ui is 3D x 2D row-major): the K-contiguous B operand of dx = DA . ui
__global__ __launch_bounds__(256) void te_transpose_kernel(const float* __restrict__ src, float* __restrict__ dst, int rows, int cols) {
  __shared__ float t[32][33];
  const int bx = blockIdx.x * 32, by = blockIdx.y * 32, x = threadIdx.x & 31, y = threadIdx.x >> 5;
  for (int k = y; k < 32; k += 8) if (by + k < rows && bx + x < cols) t[k][x] = src[(size_t)(by + k) * cols + bx + x];
  __syncthreads();
  for (int k = y; k < 32; k += 8) if (bx + k < cols && by + x < rows) dst[(size_t)(bx + k) * rows + by + x] = t[x][k];
}

// -------------------------------------------------------------------------------------------------
// te_rec_fwd16: one workgroup (D/16 waves) per tile of 16 sequences; wave w owns hidden columns
// [16w, 16w+16) of z, r, c and h (96 registers of resident weights at D = 128), so the state update is
// lane-local and two waves share each SIMD: one wave's gate math and stores overlap the other's
// MFMAs.  The recurrence is a latency chain (longest sequence x step time), so the tile is kept small:
// a 16-row step is half the matrix work of a 32-row step.  h_{t-1} / r*h_{t-1} cross waves through LDS
// (two barriers per step).
// -------------------------------------------------------------------------------------------------
// FT (forward table): the pre-activations of a step are not read from G (filled by te_gemm_ax, one row per step) but assembled here
// as ptab[p_t] + ztab[dp_t] - the step input takes far fewer distinct values than there are steps (231 k steps hit 100 k POIs),
// so te_gemm_ax multiplies the TABLE once (te_launch_ax_t) and the recurrence gathers.  Same operands, same single addition as the
// table epilogue of te_gemm_ax: bitwise the same pre-activations.  The row ids of step t+2 are fetched while step t computes
// (a dependent load chain of two).
// SP (split operands, mma16s_*): h_{t-1} / r*h_{t-1} live in LDS as three bf16 planes, the weights as three resident bf16 planes.
template <int D, bool predict, bool FT = false, bool SP = false>      // (compile-time: a runtime flag puts a branch around every store of the step)
__global__ __launch_bounds__(D * 4) void te_rec_fwd16_kernel(TeArgs A) {
  extern __shared__ __align__(16) float lds[];
  constexpr int KG = D / 16, LDA = D + 4, NW = D / 16, KG2 = D / 32, LDH = D + 8, PS = 16 * LDH;
  float* Hb = lds;                       // h_{t-1}, overwritten by h_t   16 x LDA
  float* RHb = Hb + 16 * LDA;            // r * h_{t-1}
  unsigned short* Hs = reinterpret_cast<unsigned short*>(lds);      // SP: 3 planes x 16 x LDH (bf16)
  unsigned short* RHs = Hs + 3 * PS;
  uint4* B3 = reinterpret_cast<uint4*>(RHs + 3 * PS) + (size_t)wave_id() * 3 * KG2 * 64;      // SP: this wave's third weight planes (z | r | c)
  __shared__ int s_r0[16], s_ns[16];
  const int lane = lane_id(), w = wave_id(), tid = threadIdx.x, g4 = 4 * (lane >> 4);
  const int col = 16 * w + (lane & 15);
  const int tile = blockIdx.x;
  if (tid < 16) {
    const int k = tile * 16 + tid;
    int r0 = 0, ns = 0;
    if (k < A.n_seq) { r0 = A.soff[k]; ns = A.soff[k + 1] - r0; }
    s_r0[tid] = r0; s_ns[tid] = ns;
  }
  if constexpr (SP) { for (int e = tid; e < 3 * PS; e += blockDim.x) Hs[e] = 0; }
  else { for (int e = tid; e < 16 * LDA; e += blockDim.x) Hb[e] = 0.f; }
  lds_barrier();
  int ns_max = 0;
  for (int i = 0; i < 16; ++i) ns_max = max(ns_max, s_ns[i]);
  int rowb[4], nsr[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) { rowb[r] = s_r0[g4 + r]; nsr[r] = s_ns[g4 + r]; }
  float4 wz[1][SP ? 1 : KG], wr[1][SP ? 1 : KG], wc[1][SP ? 1 : KG];
  uint4 bz[SP ? KG2 : 1][2], br[SP ? KG2 : 1][2], bc[SP ? KG2 : 1][2];
  const uint4 *cz3 = B3 + lane, *cr3 = B3 + KG2 * 64 + lane, *cc3 = B3 + 2 * KG2 * 64 + lane;
  if constexpr (SP) {
    load_bfrag3<KG2>(bz, B3, A.pWhT16, w);
    load_bfrag3<KG2>(br, B3 + KG2 * 64, A.pWhT16, NW + w);
    load_bfrag3<KG2>(bc, B3 + 2 * KG2 * 64, A.pWhT16, 2 * NW + w);
  } else {
    load_bfrag<KG>(wz[0], A.pWhT16, w);
    load_bfrag<KG>(wr[0], A.pWhT16, NW + w);
    load_bfrag<KG>(wc[0], A.pWhT16, 2 * NW + w);
  }
  // pre-activations of the NEXT step are fetched while the current step computes (G still holds
  // X.ui^T + bi for rows not yet visited)
  float cz[4], cr[4], cc[4];
  // Finished sequences read / write the spare packed row Tsp (= total rows; allocated, never
  // consumed): every access is unconditional, so no branch - and no conservative s_waitcnt vmcnt(0)
  // that would serialise the stores or turn the prefetch into a blocking load.
  const int Tsp = A.soff[A.n_seq];
  float hcur[4] = {0.f, 0.f, 0.f, 0.f};      // this lane's elements of h_{t-1} (it wrote them itself)
  // one step of the recurrence from the pre-activations cz / cr / cc
  auto compute = [&](int t) {
    if constexpr (SP) {
      f32x4 hr, lr = {0.f, 0.f, 0.f, 0.f}, hz, lz = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int r = 0; r < 4; ++r) { hr[r] = cr[r]; hz[r] = cz[r]; }
      mma16s_g2<KG2>(hr, lr, hz, lz, Hs, LDH, br, bz, cr3, cz3);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = g4 + r;
        const float rv = fast_sigmoid(hr[r] + lr[r]);
        const float rh = rv * hcur[r];
        split3_store(RHs + i * LDH + col, PS, rh);
        const size_t row = (size_t)(t < nsr[r] ? rowb[r] + t : Tsp);
        if (!predict) { A.G[row * 3 * D + D + col] = rv; A.RH[row * D + col] = rh; }
      }
      lds_barrier();
      f32x4 hc, lc = {0.f, 0.f, 0.f, 0.f}, lc2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int r = 0; r < 4; ++r) hc[r] = cc[r];
      mma16s_g1<KG2>(hc, lc, lc2, RHs, LDH, bc, cc3);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = g4 + r;
        const bool on = t < nsr[r];
        const float zv = fast_sigmoid(hz[r] + lz[r]);
        const float c = fast_tanh(hc[r] + (lc[r] + lc2[r]));
        const float hn = on ? (1.0f - zv) * hcur[r] + zv * c : hcur[r];
        split3_store(Hs + i * LDH + col, PS, hn);
        hcur[r] = hn;
        if (!predict) {
          const size_t row = (size_t)(on ? rowb[r] + t : Tsp);
          A.G[row * 3 * D + col] = zv; A.G[row * 3 * D + 2 * D + col] = c;
          A.H[row * D + col] = hn;
        }
      }
      lds_barrier();
    } else {
    // r gate first: its sigmoid, the r*h exchange and the stores then overlap the z-gate MFMAs (of this
    // wave and of its SIMD partner) instead of sitting between the MFMA block and the barrier
    f32x4 ar[1], az[1], ac[1];
#pragma unroll
    for (int r = 0; r < 4; ++r) ar[0][r] = cr[r];
    mma16_regb<KG, 1>(ar, Hb, LDA, wr);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int i = g4 + r;
      const float rv = fast_sigmoid(ar[0][r]);
      const float rh = rv * hcur[r];
      RHb[i * LDA + col] = rh;
      const size_t row = (size_t)(t < nsr[r] ? rowb[r] + t : Tsp);
      if (!predict) { A.G[row * 3 * D + D + col] = rv; A.RH[row * D + col] = rh; }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) az[0][r] = cz[r];
    mma16_regb<KG, 1>(az, Hb, LDA, wz);
    lds_barrier();
    float zv[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) { zv[r] = fast_sigmoid(az[0][r]); ac[0][r] = cc[r]; }
    mma16_regb<KG, 1>(ac, RHb, LDA, wc);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int i = g4 + r;
      const bool on = t < nsr[r];
      const float c = fast_tanh(ac[0][r]);
      const float hn = on ? (1.0f - zv[r]) * hcur[r] + zv[r] * c : hcur[r];
      Hb[i * LDA + col] = hn;          // nobody reads Hb between the two barriers of a step
      hcur[r] = hn;
      if (!predict) {
        const size_t row = (size_t)(on ? rowb[r] + t : Tsp);
        A.G[row * 3 * D + col] = zv[r]; A.G[row * 3 * D + 2 * D + col] = c;
        A.H[row * D + col] = hn;
      }
    }
    lds_barrier();
    }
  };
  if constexpr (!FT) {
    float nz[4], nr[4], nc[4];
    auto fetch = [&](int t, float (&z)[4], float (&r)[4], float (&c)[4]) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float* g = A.G + (size_t)(t < nsr[q] ? rowb[q] + t : Tsp) * 3 * D;
        z[q] = g[col]; r[q] = g[D + col]; c[q] = g[2 * D + col];
      }
    };
    fetch(0, cz, cr, cc);
    for (int t = 0; t < ns_max; ++t) {
      fetch(t + 1, nz, nr, nc);
      compute(t);
      // Opaque use of the prefetched values HERE: the wait for them is then counted in straight-line code behind
      // this step's stores (vmcnt(#stores)).  Left to the first use at the top of the next iteration, it merges
      // with the loop-entry state and becomes a wait for most of the stores as well - a store round trip per step.
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        asm volatile("" : "+v"(nz[r]), "+v"(nr[r]), "+v"(nc[r]));
        cz[r] = nz[r]; cr[r] = nr[r]; cc[r] = nc[r];
      }
    }
  } else {
    // Forward table, TWO steps of prefetch: the table rows of step t+2 are requested at the top of step t (random 1.5 KB rows: a
    // longer latency than the sequential G rows of the plain path, and one workgroup per CU has nothing else to hide it), the
    // row ids of step t+3 right after.  Two register sets alternate by step parity (the loop is unrolled by two, so they are
    // renamed, not moved).  Row ids: lane (lane & 3) of every quad loads those of tile row g4 + (lane & 3); quad broadcasts hand
    // them to the other three lanes when they are consumed, a step later.
    struct f3 { float x, y, z; };
    struct Pre { f3 g[4], y[4]; };
    const int sub = lane & 3;
    const int my_rowb = sub == 0 ? rowb[0] : sub == 1 ? rowb[1] : sub == 2 ? rowb[2] : rowb[3];
    const int my_ns = sub == 0 ? nsr[0] : sub == 1 ? nsr[1] : sub == 2 ? nsr[2] : nsr[3];
    int rp = 0, rz = 0;                        // raw row ids of this lane's row (in flight)
    auto ids = [&](int t) {
      const int rr = t < my_ns ? my_rowb + t : Tsp;              // (row Tsp holds whatever an earlier launch left: clamped on use)
      rp = A.row_p[rr]; rz = A.row_dp[rr];
    };
    // consume rp / rz (row ids of step tn - 1), request the row ids of step tn, THEN the table rows: vmcnt retires in order, so the
    // wait for the ids at the top of the next step covers only what is older than them - the table rows get two full steps
    auto rows = [&](Pre& X, int tn) {
      const int p1 = (int)min((unsigned)rp, (unsigned)A.n_item), z1 = (int)min((unsigned)rz, (unsigned)A.n_dist);
      int pi[4], zi[4];
      pi[0] = __builtin_amdgcn_mov_dpp(p1, 0x00, 0xF, 0xF, true); zi[0] = __builtin_amdgcn_mov_dpp(z1, 0x00, 0xF, 0xF, true);
      pi[1] = __builtin_amdgcn_mov_dpp(p1, 0x55, 0xF, 0xF, true); zi[1] = __builtin_amdgcn_mov_dpp(z1, 0x55, 0xF, 0xF, true);
      pi[2] = __builtin_amdgcn_mov_dpp(p1, 0xAA, 0xF, 0xF, true); zi[2] = __builtin_amdgcn_mov_dpp(z1, 0xAA, 0xF, 0xF, true);
      pi[3] = __builtin_amdgcn_mov_dpp(p1, 0xFF, 0xF, 0xF, true); zi[3] = __builtin_amdgcn_mov_dpp(z1, 0xFF, 0xF, 0xF, true);
      ids(tn);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        X.g[q] = *reinterpret_cast<const f3*>(A.ptab + (size_t)pi[q] * 3 * D + 3 * col);      // (gate-interleaved columns)
        X.y[q] = *reinterpret_cast<const f3*>(A.ztab + (size_t)zi[q] * 3 * D + 3 * col);
      }
    };
    auto take = [&](Pre& X) {                  // pre-activations of the next step <- a landed register set
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        asm volatile("" : "+v"(X.g[q].x), "+v"(X.g[q].y), "+v"(X.g[q].z), "+v"(X.y[q].x), "+v"(X.y[q].y), "+v"(X.y[q].z));
        cz[q] = X.g[q].x + X.y[q].x; cr[q] = X.g[q].y + X.y[q].y; cc[q] = X.g[q].z + X.y[q].z;
      }
    };
    Pre B0, B1;
    ids(0); rows(B0, 1); take(B0);             // step 0
    rows(B1, 2);                               // step 1 in flight, ids of step 2
    int t = 0;
    for (; t + 1 < ns_max; t += 2) {
      rows(B0, t + 3); compute(t); take(B1);                 // B0 <- step t+2; step t+1 <- B1
      rows(B1, t + 4); compute(t + 1); take(B0);             // B1 <- step t+3; step t+2 <- B0
    }
    if (t < ns_max) compute(t);                // odd length: the pre-activations of the last step are already in place
  }
  if (predict) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int i = g4 + r, k = tile * 16 + i;
      if (k < A.n_seq) A.hts[(size_t)(A.out_row ? A.out_row[k] : k) * D + col] = hcur[r];
    }
  }
}

// -------------------------------------------------------------------------------------------------
// te_rec_bwd16: BPTT per 16-sequence tile, t descending; wave w owns hidden columns [16w, 16w+16)
// (same layout and reasoning as te_rec_fwd16).
// -------------------------------------------------------------------------------------------------
template <int D, bool SP = false>
__global__ __launch_bounds__(D * 4) void te_rec_bwd16_kernel(TeArgs A) {
  extern __shared__ __align__(16) float lds[];
  constexpr int KG = D / 16, LDA = D + 4, LDB = 2 * D + 4, KG2 = D / 32, LHA = D + 8, LHB = 2 * D + 8;
  float* Ac = lds;                       // da_c           16 x LDA
  float* Azr = Ac + 16 * LDA;            // da_z | da_r    16 x LDB
  unsigned short* Acs = reinterpret_cast<unsigned short*>(lds);      // SP: 3 planes x 16 x LHA (bf16)
  unsigned short* Azrs = Acs + 3 * 16 * LHA;                          //     3 planes x 16 x LHB
  uint4* B3 = reinterpret_cast<uint4*>(Azrs + 3 * 16 * LHB) + (size_t)wave_id() * 3 * KG2 * 64;      // this wave's third weight planes (c | zr)
  __shared__ int s_r0[16], s_ns[16];
  const int lane = lane_id(), w = wave_id(), tid = threadIdx.x, g4 = 4 * (lane >> 4);
  const int col = 16 * w + (lane & 15);
  const int tile = blockIdx.x;
  if (tid < 16) {
    const int k = tile * 16 + tid;
    int r0 = 0, ns = 0;
    if (k < A.n_seq) { r0 = A.soff[k]; ns = A.soff[k + 1] - r0; }
    s_r0[tid] = r0; s_ns[tid] = ns;
  }
  lds_barrier();
  int ns_max = 0;
  for (int i = 0; i < 16; ++i) ns_max = max(ns_max, s_ns[i]);
  int rowb[4], nsr[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) { rowb[r] = s_r0[g4 + r]; nsr[r] = s_ns[g4 + r]; }
  float4 wcb[1][SP ? 1 : KG], wzrb[1][SP ? 1 : 2 * KG];
  uint4 bcb[SP ? KG2 : 1][2], bzrb[SP ? 2 * KG2 : 1][2];
  const uint4 *cc3 = B3 + lane, *czr3 = B3 + KG2 * 64 + lane;
  if constexpr (SP) {
    load_bfrag3<KG2>(bcb, B3, A.pWhc16, w);
    load_bfrag3<2 * KG2>(bzrb, B3 + KG2 * 64, A.pWhzr16, w);
  } else {
    load_bfrag<KG>(wcb[0], A.pWhc16, w);
    load_bfrag<2 * KG>(wzrb[0], A.pWhzr16, w);
  }
  float dhn[4], sbz = 0.f, sbr = 0.f, sbc = 0.f;
#pragma unroll
  for (int r = 0; r < 4; ++r) dhn[r] = 0.f;
  // operands of step t-2 (z, r, c, h_{t-3}, DH) are fetched while step t computes: TWO steps of prefetch in two register sets that
  // alternate by step parity (loop unrolled by two: renamed, not moved) - one workgroup per CU has nothing else to hide the HBM
  // latency of the five rows a step reads, and a step is shorter than that latency under load (te_rec_fwd16<FT>: -7 %)
  struct Ops { float z[4], r[4], c[4], h[4], d[4]; };
  float fz[4], fr[4], fc[4], fh[4], fd[4];       // masked operands of the current step
  // unconditional accesses (inactive lanes use the spare packed row Tsp): see te_rec_fwd16
  const int Tsp = A.soff[A.n_seq];
  auto fetch = [&](int t, Ops& X) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const bool on = t >= 0 && t < nsr[q];
      const size_t row = (size_t)(on ? rowb[q] + t : Tsp);
      const float* g = A.G + row * 3 * D;
      // raw values; take() masks them (a select here would wait for the load at once)
      X.z[q] = g[col]; X.r[q] = g[D + col]; X.c[q] = g[2 * D + col];
      X.h[q] = A.H[(on && t > 0 ? row - 1 : (size_t)Tsp) * D + col];
      X.d[q] = A.DH[row * D + col];
    }
  };
  auto take = [&](Ops& X, int t) {               // operands of step t <- a landed register set
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      asm volatile("" : "+v"(X.z[r]), "+v"(X.r[r]), "+v"(X.c[r]), "+v"(X.h[r]), "+v"(X.d[r]));
      const bool on = t >= 0 && t < nsr[r];
      // the spare row holds arbitrary bits: select, do not multiply by zero
      fz[r] = on ? X.z[r] : 0.f; fr[r] = on ? X.r[r] : 0.f; fc[r] = on ? X.c[r] : 0.f;
      fh[r] = (on && t > 0) ? X.h[r] : 0.f; fd[r] = on ? X.d[r] : 0.f;
    }
  };
  auto compute = [&](int t) {
    float zv[4], rv[4], hp[4], dz[4], dhp[4], dacv[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int i = g4 + r;
      const bool on = t < nsr[r];
      const float z = fz[r], rr = fr[r], c = fc[r], h = fh[r];
      const float dh = on ? dhn[r] + fd[r] : 0.f;
      zv[r] = z; rv[r] = rr; hp[r] = h;
      dz[r] = dh * (c - h);
      dhp[r] = dh * (1.0f - z);
      dacv[r] = dh * z * (1.0f - c * c);
      if constexpr (SP) split3_store(Acs + i * LHA + col, 16 * LHA, dacv[r]);
      else Ac[i * LDA + col] = dacv[r];
    }
    lds_barrier();
    f32x4 m[1];
#pragma unroll
    for (int r = 0; r < 4; ++r) m[0][r] = 0.f;
    if constexpr (SP) {
      f32x4 ml = {0.f, 0.f, 0.f, 0.f}, ml2 = {0.f, 0.f, 0.f, 0.f};
      mma16s_g1<KG2>(m[0], ml, ml2, Acs, LHA, bcb, cc3);
      m[0] += ml + ml2;
    } else {
      mma16_regb<KG, 1>(m, Ac, LDA, wcb);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int i = g4 + r;
      const float mv = m[0][r];
      const float dr = mv * hp[r];
      dhp[r] += mv * rv[r];
      const float daz = dz[r] * zv[r] * (1.0f - zv[r]);
      const float dar = dr * rv[r] * (1.0f - rv[r]);
      if constexpr (SP) {
        split3_store(Azrs + i * LHB + col, 16 * LHB, daz);
        split3_store(Azrs + i * LHB + D + col, 16 * LHB, dar);
      } else {
        Azr[i * LDB + col] = daz;
        Azr[i * LDB + D + col] = dar;
      }
      float* g = A.G + (size_t)(t < nsr[r] ? rowb[r] + t : Tsp) * 3 * D;
      g[col] = daz; g[D + col] = dar; g[2 * D + col] = dacv[r];
      sbz += daz; sbr += dar; sbc += dacv[r];        // zero for inactive steps (dh == 0)
    }
    lds_barrier();
    f32x4 acc[1];
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[0][r] = 0.f;
    if constexpr (SP) {
      f32x4 al = {0.f, 0.f, 0.f, 0.f}, al2 = {0.f, 0.f, 0.f, 0.f};
      mma16s_g1<2 * KG2>(acc[0], al, al2, Azrs, LHB, bzrb, czr3);
      acc[0] += al + al2;
    } else {
      mma16_regb<2 * KG, 1>(acc, Azr, LDB, wzrb);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) dhn[r] = (t < nsr[r]) ? dhp[r] + acc[0][r] : 0.f;
    // No barrier here: Ac is free once every wave passed the second barrier (its readers ran before
    // it), and the next step's Azr writes come after the next first barrier, which every wave reaches
    // only after finishing this step's MFMA block on Azr.
  };
  Ops B0, B1;
  int t = ns_max - 1;
  fetch(t, B0); take(B0, t);
  fetch(t - 1, B1);
  for (; t >= 1; t -= 2) {
    fetch(t - 2, B0); compute(t); take(B1, t - 1);
    fetch(t - 3, B1); compute(t - 1); take(B0, t - 2);
  }
  if (t == 0) compute(0);
  // d bi partial sums of this tile: the four 16-lane groups hold different sequences of the same column;
  // written per tile and summed in tile order by te_parts_kernel (no float atomics: reproducible)
  sbz += __shfl_xor(sbz, 16, 64); sbr += __shfl_xor(sbr, 16, 64); sbc += __shfl_xor(sbc, 16, 64);
  sbz += __shfl_xor(sbz, 32, 64); sbr += __shfl_xor(sbr, 32, 64); sbc += __shfl_xor(sbc, 32, 64);
  if (lane < 16) {
    float* bp = A.bi_part + (size_t)tile * 3 * D;
    bp[col] = sbz; bp[D + col] = sbr; bp[2 * D + col] = sbc;
  }
}

// te_rec_bwd16 on split products with TRANSPOSED products (round 4, as te_rec_fwdx): the weight fragments are the A operand, so a lane owns
// FOUR CONSECUTIVE UNITS of ONE sequence instead of one unit of four sequences - every operand of a step is one 16-byte load (5 instead of 20
// 4-byte ones), d a goes out as three 16-byte stores (12), the bf16 planes of a value quad are one 8-byte LDS write per plane (3 instead of twelve
// 2-byte ones, which conflict two ways), one activity mask per lane.  Same formulas on the same MFMA products: d a, d h agree with te_rec_bwd16<SP>
// to float32 rounding (tests: 1e-6; the compiler contracts the gate-derivative expressions differently).
template <int D>
__global__ __launch_bounds__(D * 4) void te_rec_bwd16t_kernel(TeArgs A) {
  extern __shared__ __align__(16) float lds[];
  constexpr int KG2 = D / 32, LHA = D + 8, LHB = 2 * D + 8;
  unsigned short* Acs = reinterpret_cast<unsigned short*>(lds);      // 3 planes x 16 x LHA (bf16): d a_c
  unsigned short* Azrs = Acs + 3 * 16 * LHA;                          // 3 planes x 16 x LHB: d a_z | d a_r
  uint4* B3 = reinterpret_cast<uint4*>(Azrs + 3 * 16 * LHB) + (size_t)wave_id() * 3 * KG2 * 64;      // this wave's third weight planes (c | zr)
  __shared__ int s_r0[16], s_ns[16];
  const int lane = lane_id(), w = wave_id(), tid = threadIdx.x;
  const int sq = lane & 15, u0 = 16 * w + 4 * (lane >> 4);            // this lane's sequence of the tile and its first unit
  const int tile = blockIdx.x;
  const int k_lo = A.hyb ? A.hyb_dev[2] : 0;       // hybrid recurrences: the tiles start behind the sequences of the per-sequence kernel
  if (k_lo + tile * 16 >= A.n_seq) return;         // (uniform; the grid is sized for k_lo == 0)
  if (tid < 16) {
    const int k = k_lo + tile * 16 + tid;
    int r0 = 0, ns = 0;
    if (k < A.n_seq) { r0 = A.soff[k]; ns = A.soff[k + 1] - r0; }
    s_r0[tid] = r0; s_ns[tid] = ns;
  }
  lds_barrier();
  int ns_max = 0;
  for (int i = 0; i < 16; ++i) ns_max = max(ns_max, s_ns[i]);
  const int rowb = s_r0[sq], nsr = s_ns[sq];
  uint4 bcb[KG2][2], bzrb[2 * KG2][2];
  const uint4 *cc3 = B3 + lane, *czr3 = B3 + KG2 * 64 + lane;
  load_bfrag3<KG2>(bcb, B3, A.pWhc16, w);
  load_bfrag3<2 * KG2>(bzrb, B3 + KG2 * 64, A.pWhzr16, w);
  float dhn[4], sbz[4], sbr[4], sbc[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) { dhn[r] = 0.f; sbz[r] = 0.f; sbr[r] = 0.f; sbc[r] = 0.f; }
  // two steps of operand prefetch in two register sets that alternate by step parity (see te_rec_bwd16)
  struct Ops { float4 z, r, c, h, d; };
  float fz[4], fr[4], fc[4], fh[4], fd[4];       // masked operands of the current step
  const int Tsp = A.soff[A.n_seq];               // unconditional accesses: inactive lanes use the spare packed row
  auto fetch = [&](int t, Ops& X) {
    const bool on = t >= 0 && t < nsr;
    const size_t row = (size_t)(on ? rowb + t : Tsp);
    const float* g = A.G + row * 3 * D + u0;
    X.z = *reinterpret_cast<const float4*>(g); X.r = *reinterpret_cast<const float4*>(g + D); X.c = *reinterpret_cast<const float4*>(g + 2 * D);
    X.h = *reinterpret_cast<const float4*>(A.H + (on && t > 0 ? row - 1 : (size_t)Tsp) * D + u0);
    X.d = *reinterpret_cast<const float4*>(A.DH + row * D + u0);
  };
  auto take = [&](Ops& X, int t) {               // operands of step t <- a landed register set (the spare row holds arbitrary bits: select)
    asm volatile("" : "+v"(X.z.x), "+v"(X.z.y), "+v"(X.z.z), "+v"(X.z.w), "+v"(X.r.x), "+v"(X.r.y), "+v"(X.r.z), "+v"(X.r.w), "+v"(X.c.x), "+v"(X.c.y), "+v"(X.c.z), "+v"(X.c.w));
    asm volatile("" : "+v"(X.h.x), "+v"(X.h.y), "+v"(X.h.z), "+v"(X.h.w), "+v"(X.d.x), "+v"(X.d.y), "+v"(X.d.z), "+v"(X.d.w));
    const bool on = t >= 0 && t < nsr, onh = on && t > 0;
    fz[0] = on ? X.z.x : 0.f; fz[1] = on ? X.z.y : 0.f; fz[2] = on ? X.z.z : 0.f; fz[3] = on ? X.z.w : 0.f;
    fr[0] = on ? X.r.x : 0.f; fr[1] = on ? X.r.y : 0.f; fr[2] = on ? X.r.z : 0.f; fr[3] = on ? X.r.w : 0.f;
    fc[0] = on ? X.c.x : 0.f; fc[1] = on ? X.c.y : 0.f; fc[2] = on ? X.c.z : 0.f; fc[3] = on ? X.c.w : 0.f;
    fh[0] = onh ? X.h.x : 0.f; fh[1] = onh ? X.h.y : 0.f; fh[2] = onh ? X.h.z : 0.f; fh[3] = onh ? X.h.w : 0.f;
    fd[0] = on ? X.d.x : 0.f; fd[1] = on ? X.d.y : 0.f; fd[2] = on ? X.d.z : 0.f; fd[3] = on ? X.d.w : 0.f;
  };
  // four consecutive values -> their three planes at LDS element `at`: one 8-byte write per plane (wg_split2 == split3 on a pair)
  auto store4 = [&](unsigned short* at, int ps, const float (&v)[4]) {
    uint2 p1, p2, p3;
    wg_split2(v[0], v[1], p1.x, p2.x, p3.x); wg_split2(v[2], v[3], p1.y, p2.y, p3.y);
    *reinterpret_cast<uint2*>(at) = p1; *reinterpret_cast<uint2*>(at + ps) = p2; *reinterpret_cast<uint2*>(at + 2 * ps) = p3;
  };
  auto compute = [&](int t) {
    const bool on = t < nsr;
    float dz[4], dhp[4], dacv[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float dh = on ? dhn[r] + fd[r] : 0.f;
      dz[r] = dh * (fc[r] - fh[r]);
      dhp[r] = dh * (1.0f - fz[r]);
      dacv[r] = dh * fz[r] * (1.0f - fc[r] * fc[r]);
    }
    store4(Acs + sq * LHA + u0, 16 * LHA, dacv);
    lds_barrier();
    f32x4 m = {0.f, 0.f, 0.f, 0.f}, ml = {0.f, 0.f, 0.f, 0.f}, ml2 = {0.f, 0.f, 0.f, 0.f};
    mma16s_g1<KG2, true>(m, ml, ml2, Acs, LHA, bcb, cc3);
    m += ml + ml2;
    float daz[4], dar[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float mv = m[r];
      const float dr = mv * fh[r];
      dhp[r] += mv * fr[r];
      daz[r] = dz[r] * fz[r] * (1.0f - fz[r]);
      dar[r] = dr * fr[r] * (1.0f - fr[r]);
      sbz[r] += daz[r]; sbr[r] += dar[r]; sbc[r] += dacv[r];        // zero for inactive steps (dh == 0)
    }
    store4(Azrs + sq * LHB + u0, 16 * LHB, daz);
    store4(Azrs + sq * LHB + D + u0, 16 * LHB, dar);
    {
      float* g = A.G + (size_t)(on ? rowb + t : Tsp) * 3 * D + u0;
      *reinterpret_cast<float4*>(g) = make_float4(daz[0], daz[1], daz[2], daz[3]);
      *reinterpret_cast<float4*>(g + D) = make_float4(dar[0], dar[1], dar[2], dar[3]);
      *reinterpret_cast<float4*>(g + 2 * D) = make_float4(dacv[0], dacv[1], dacv[2], dacv[3]);
    }
    lds_barrier();
    f32x4 acc = {0.f, 0.f, 0.f, 0.f}, al = {0.f, 0.f, 0.f, 0.f}, al2 = {0.f, 0.f, 0.f, 0.f};
    mma16s_g1<2 * KG2, true>(acc, al, al2, Azrs, LHB, bzrb, czr3);
    acc += al + al2;
#pragma unroll
    for (int r = 0; r < 4; ++r) dhn[r] = on ? dhp[r] + acc[r] : 0.f;
    // (no barrier here: see te_rec_bwd16)
  };
  Ops B0, B1;
  int t = ns_max - 1;
  fetch(t, B0); take(B0, t);
  fetch(t - 1, B1);
  for (; t >= 1; t -= 2) {
    fetch(t - 2, B0); compute(t); take(B1, t - 1);
    fetch(t - 3, B1); compute(t - 1); take(B0, t - 2);
  }
  if (t == 0) compute(0);
  // d bi partial sums of this tile: the sixteen lanes of a group hold the sequences of the same four units; written per tile and summed in
  // tile order by te_parts_kernel (no float atomics: reproducible)
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) { sbz[r] += __shfl_xor(sbz[r], o, 64); sbr[r] += __shfl_xor(sbr[r], o, 64); sbc[r] += __shfl_xor(sbc[r], o, 64); }
  if (sq == 0) {
    float* bp = A.bi_part + (size_t)(k_lo + tile) * 3 * D + u0;      // (rows [0, k_lo): one per sequence of the per-sequence kernel)
    *reinterpret_cast<float4*>(bp) = make_float4(sbz[0], sbz[1], sbz[2], sbz[3]);
    *reinterpret_cast<float4*>(bp + D) = make_float4(sbr[0], sbr[1], sbr[2], sbr[3]);
    *reinterpret_cast<float4*>(bp + 2 * D) = make_float4(sbc[0], sbc[1], sbc[2], sbc[3]);
  }
}

// -------------------------------------------------------------------------------------------------
// te_rec_fwd1 / te_rec_bwd1: the recurrence of ONE sequence per workgroup on the vector ALUs, for launches too small to fill the
// chip with 16-sequence tiles (TeArgs.rec1: n_seq <= rec1_max).  The float32-input MFMA has no rate advantage over v_fma_f32 on
// gfx950, and a 16-row MFMA tile that holds one sequence wastes 15/16 of it: a tile step is ~2.5 us whatever it holds, a step here
// ~0.5 us (3 D^2 FMAs over 512 threads with the weights RESIDENT in registers - 96 per thread at D = 128 - and h_{t-1} broadcast
// from LDS).  A launch of n <= 256 sequences runs them all at once, one per CU: 49 steps in ~30 us against ~125 us; the reference
// schedule (one user per step) spends 2 x 15 us here instead of 2 x 65 us.
//   forward : 4 D threads; a thread owns FOUR outputs and one k-slice (8 slices in the z|r phase, 16 in the c phase; adjacent lanes:
//             DPP sums) - one 16-byte LDS read of h feeds 16 FMAs (one output per thread made the kernel LDS-bound: 1.25 us per
//             step).  Reads the pre-activations from G (te_gemm_ax), writes z | r | c over them, H, RH - the buffers the rest of
//             the step reads, as the tile kernels do.
//   backward: thread (g, s) owns hidden columns 4g .. 4g+3 and the j-slice s of 16: m = da_c . Wc and dh_{t-1} += [da_z | da_r] . Wzr
//             with the TRANSPOSED weights (te_pack n16 == 3 -> pWhc16 / pWhzr16 as plain float matrices) contiguous per thread.
//   d bi partials: one row per sequence (bi_part is sized for it), summed in order by te_parts.
// Same formulas as te_rec_fwd16 / bwd16; summation order differs (tolerance-tested against the oracle and the tile kernels).
// -------------------------------------------------------------------------------------------------
template <int N> __device__ __forceinline__ float group_sum(float v) {      // sum over N = 8 / 16 adjacent lanes, in every lane
  v += dpp_f<0xB1>(v);                     // quad_perm [1,0,3,2]
  v += dpp_f<0x4E>(v);                     // quad_perm [2,3,0,1]
  v += dpp_f<0x141>(v);                    // row_half_mirror (quads already uniform)
  if (N >= 16) v += dpp_f<0x140>(v);       // row_mirror
  return v;
}
// a[o] += w[o][:] . x[:] for FOUR outputs sharing the LDS slice x (one 16-byte LDS read feeds 16 FMAs).  The weights are held as PAIRS
// (outputs 0|1 and 2|3) so that the products are v_pk_fma_f32 - two float32 FMAs per lane and issue slot, the only form that reaches the
// float32 vector peak on gfx950 (a scalar v_fma_f32 stream runs at half of it); x is broadcast into both halves by op_sel.
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int L>
__device__ __forceinline__ void dot4_reg_lds(const f32x2 (&w)[2][L], const float* __restrict__ x, float (&a)[4]) {
  f32x2 p0 = {a[0], a[1]}, p1 = {a[2], a[3]};
#pragma unroll
  for (int i = 0; i < L; i += 4) {
    const float4 u = *reinterpret_cast<const float4*>(x + i);
    p0 = __builtin_elementwise_fma(w[0][i], f32x2{u.x, u.x}, p0); p1 = __builtin_elementwise_fma(w[1][i], f32x2{u.x, u.x}, p1);
    p0 = __builtin_elementwise_fma(w[0][i + 1], f32x2{u.y, u.y}, p0); p1 = __builtin_elementwise_fma(w[1][i + 1], f32x2{u.y, u.y}, p1);
    p0 = __builtin_elementwise_fma(w[0][i + 2], f32x2{u.z, u.z}, p0); p1 = __builtin_elementwise_fma(w[1][i + 2], f32x2{u.z, u.z}, p1);
    p0 = __builtin_elementwise_fma(w[0][i + 3], f32x2{u.w, u.w}, p0); p1 = __builtin_elementwise_fma(w[1][i + 3], f32x2{u.w, u.w}, p1);
  }
  a[0] = p0.x; a[1] = p0.y; a[2] = p1.x; a[3] = p1.y;
}
template <int L>
__device__ __forceinline__ void load_rows4(f32x2 (&w)[2][L], const float* __restrict__ src, int ld) {       // w[o / 2][i][o % 2] = src[o * ld + i]
#pragma unroll
  for (int i = 0; i < L; i += 4) {
    const float4 r0 = *reinterpret_cast<const float4*>(src + i), r1 = *reinterpret_cast<const float4*>(src + (size_t)ld + i);
    const float4 r2 = *reinterpret_cast<const float4*>(src + (size_t)2 * ld + i), r3 = *reinterpret_cast<const float4*>(src + (size_t)3 * ld + i);
    w[0][i] = f32x2{r0.x, r1.x}; w[0][i + 1] = f32x2{r0.y, r1.y}; w[0][i + 2] = f32x2{r0.z, r1.z}; w[0][i + 3] = f32x2{r0.w, r1.w};
    w[1][i] = f32x2{r2.x, r3.x}; w[1][i + 1] = f32x2{r2.y, r3.y}; w[1][i + 2] = f32x2{r2.z, r3.z}; w[1][i + 3] = f32x2{r2.w, r3.w};
  }
}
__device__ __forceinline__ float pick4(const float (&a)[4], int o) { return o == 0 ? a[0] : o == 1 ? a[1] : o == 2 ? a[2] : a[3]; }

// 4 D threads.  A thread owns FOUR outputs and one k-slice: z|r phase 8 slices of D / 8, c phase 16 slices of D / 16 (adjacent lanes: DPP
// sums); lanes 0 - 3 of a group then finish one output each (activation, state update, stores).
template <int D, bool predict>
__global__ __launch_bounds__(4 * D) void te_rec_fwd1_kernel(TeArgs A) {
  constexpr int LZ = D / 8, LC = D / 16;
  __shared__ __align__(16) float hs[D], rhs[D], zs[D];
  const int tid = threadIdx.x, k = blockIdx.x;
  const int gz = tid >> 3, sz = tid & 7, gc = tid >> 4, sc = tid & 15;
  const int jz = 4 * gz + (sz & 3), jc = 4 * gc + (sc & 3);        // the output this lane finishes (lanes sz / sc < 4)
  const bool isr = jz >= D;
  const int jr = isr ? jz - D : jz;
  const int r0 = A.soff[k], ns = A.soff[k + 1] - r0;
  f32x2 wzr[2][LZ], wc[2][LC];
  load_rows4<LZ>(wzr, A.wh + (size_t)4 * gz * D + sz * LZ, D);
  load_rows4<LC>(wc, A.wh + (size_t)(2 * D + 4 * gc) * D + sc * LC, D);
  if (tid < D) hs[tid] = 0.f;
  __syncthreads();
  // pre-activations of the next step are requested a step ahead (G row r0 + t + 1 is not written before step t + 1)
  float gzr = 0.f, gcc = 0.f;
  if (ns > 0) { gzr = A.G[(size_t)r0 * 3 * D + jz]; gcc = A.G[(size_t)r0 * 3 * D + 2 * D + jc]; }
  // Every lane issues every global store of a step - the lanes that do not own an output write their (duplicate) value into the spare
  // packed row - so that the stores sit in straight-line code: vmcnt counts loads AND stores in order, and with the stores inside the
  // owner branches the compiler's wait for the prefetched pre-activations also waited for the stores issued after them (a store round
  // trip per step: 1.07 us per step instead of 0.6).
  const size_t Tsp = (size_t)A.soff[A.n_seq];
  const bool ownz = sz < 4, ownc = sc < 4;
  float* const dG = A.G + Tsp * 3 * D + (tid % (3 * D));
  float* const dH = A.H + Tsp * D + (tid % D);
  float* const dR = A.RH + Tsp * D + (tid % D);
  for (int t = 0; t < ns; ++t) {
    const size_t row = (size_t)(r0 + t), rn = (size_t)(r0 + min(t + 1, ns - 1));
    float nzr = A.G[rn * 3 * D + jz], nc = A.G[rn * 3 * D + 2 * D + jc];
    float a[4] = {0.f, 0.f, 0.f, 0.f};
    dot4_reg_lds<LZ>(wzr, hs + sz * LZ, a);
#pragma unroll
    for (int o = 0; o < 4; ++o) a[o] = group_sum<8>(a[o]);
    {
      const float v = fast_sigmoid(pick4(a, sz & 3) + gzr);
      const float rh = v * hs[jr];
      if (ownz) { if (isr) rhs[jr] = rh; else zs[jr] = v; }
      if (!predict) {
        const bool st = ownz && isr;
        *(st ? A.G + row * 3 * D + D + jr : dG) = v;
        *(st ? A.RH + row * D + jr : dR) = rh;
      }
    }
    lds_barrier();
    float b[4] = {0.f, 0.f, 0.f, 0.f};
    dot4_reg_lds<LC>(wc, rhs + sc * LC, b);
#pragma unroll
    for (int o = 0; o < 4; ++o) b[o] = group_sum<16>(b[o]);
    {
      const float c = fast_tanh(pick4(b, sc & 3) + gcc);
      const float z = zs[jc];
      const float hn = (1.0f - z) * hs[jc] + z * c;
      if (ownc) hs[jc] = hn;               // (the c phase reads rhs only)
      if (!predict) {
        *(ownc ? A.G + row * 3 * D + jc : dG) = z;
        *(ownc ? A.G + row * 3 * D + 2 * D + jc : dG) = c;
        *(ownc ? A.H + row * D + jc : dH) = hn;
      }
    }
    lds_barrier();
    asm volatile("" : "+v"(nzr), "+v"(nc));      // the wait for the prefetch is counted HERE, behind this step's stores
    gzr = nzr; gcc = nc;
  }
  if (predict && tid < D) A.hts[(size_t)(A.out_row ? A.out_row[k] : k) * D + tid] = hs[tid];
}

// thread (g, s): hidden columns 4g .. 4g+3, j-slice s of 16 - of D / 16 (m = da_c . Wc) and of D / 8 ([da_z | da_r] . Wzr)
// (round 5: persistent, as te_rec_fwd1x - one workgroup per CU slot walks the launch's sequences in snake order, the transposed weights loaded once)
template <int D>
__global__ __launch_bounds__(4 * D) void te_rec_bwd1_kernel(TeArgs A) {
  constexpr int LC = D / 16, LZ = D / 8;
  __shared__ __align__(16) float dac[D], dazr[2 * D];
  const int tid = threadIdx.x;
  const int g = tid >> 4, s = tid & 15;
  const int kk = 4 * g + (s & 3);          // the column this lane finishes (lanes s < 4)
  f32x2 wc[2][LC], wzr[2][LZ];
  // hybrid recurrences (TeArgs.hyb): the leading hyb_dev[2] sequences on hyb_dev[3] workgroups, the rest in tiles (te_rec_bwd16t) at the same time
  const int n1 = A.hyb ? A.hyb_dev[2] : A.n_seq, NG = A.hyb ? A.hyb_dev[3] : (int)gridDim.x, bq = blockIdx.x;
  if (bq >= NG) return;
  load_rows4<LC>(wc, reinterpret_cast<const float*>(A.pWhc1) + (size_t)4 * g * D + s * LC, D);              // Wc^T:  [k][j]
  load_rows4<LZ>(wzr, reinterpret_cast<const float*>(A.pWhzr1) + (size_t)4 * g * 2 * D + s * LZ, 2 * D);     // Wzr^T: [k][j], j < 2D
  const bool own = s < 4;
  float* const dG = A.G + ((size_t)A.soff[A.n_seq] + 1 + (blockIdx.x & 127)) * 3 * D + (tid % (3 * D));      // a spare packed row of its own: see te_rec_fwd1x
  for (int j = 0;; ++j) {
    const int k = j * NG + ((j & 1) ? NG - 1 - bq : bq);
    if (k >= n1) break;
    const int r0 = A.soff[k], ns = A.soff[k + 1] - r0;
    float dhn = 0.f, sbz = 0.f, sbr = 0.f, sbc = 0.f;
    // operands of step t - 1 are requested while step t computes
    float z = 0.f, r = 0.f, c = 0.f, hp = 0.f, dd = 0.f;
    auto fetch = [&](int t, float& fz, float& fr, float& fc, float& fh, float& fd) {
      const size_t row = (size_t)(r0 + max(t, 0));
      const float* gp = A.G + row * 3 * D;
      fz = gp[kk]; fr = gp[D + kk]; fc = gp[2 * D + kk];
      fh = A.H[(row - (t > 0 ? 1 : 0)) * D + kk];
      fd = A.DH[row * D + kk];
    };
    if (ns > 0) fetch(ns - 1, z, r, c, hp, dd);
    for (int t = ns - 1; t >= 0; --t) {
      float nz, nr, nc, nh, nd;
      fetch(t - 1, nz, nr, nc, nh, nd);
      const float h = t > 0 ? hp : 0.f;
      const float dh = dhn + dd;
      const float dz = dh * (c - h);
      float dhp = dh * (1.0f - z);
      const float dacv = dh * z * (1.0f - c * c);
      if (own) dac[kk] = dacv;
      lds_barrier();
      float a[4] = {0.f, 0.f, 0.f, 0.f};
      dot4_reg_lds<LC>(wc, dac + s * LC, a);
#pragma unroll
      for (int o = 0; o < 4; ++o) a[o] = group_sum<16>(a[o]);
      const float m = pick4(a, s & 3);
      const float dr = m * h;
      dhp += m * r;
      const float daz = dz * z * (1.0f - z), dar = dr * r * (1.0f - r);
      if (own) { dazr[kk] = daz; dazr[D + kk] = dar; }
      {
        float* gp = A.G + (size_t)(r0 + t) * 3 * D;
        *(own ? gp + kk : dG) = daz; *(own ? gp + D + kk : dG) = dar; *(own ? gp + 2 * D + kk : dG) = dacv;
        sbz += daz; sbr += dar; sbc += dacv;      // (every lane of a column holds the same values; lanes s < 4 write them)
      }
      lds_barrier();
      float b[4] = {0.f, 0.f, 0.f, 0.f};
      dot4_reg_lds<LZ>(wzr, dazr + s * LZ, b);
#pragma unroll
      for (int o = 0; o < 4; ++o) b[o] = group_sum<16>(b[o]);
      dhn = dhp + pick4(b, s & 3);
      asm volatile("" : "+v"(nz), "+v"(nr), "+v"(nc), "+v"(nh), "+v"(nd));      // (wait for the prefetch behind this step's stores)
      z = nz; r = nr; c = nc; hp = nh; dd = nd;
    }
    if (s < 4) {
      float* bp = A.bi_part + (size_t)k * 3 * D;
      bp[kk] = sbz; bp[D + kk] = sbr; bp[2 * D + kk] = sbc;
    }
    // (no barrier between sequences: the next one's first LDS write - dac - follows this one's last barrier, and its dazr write sits behind its own
    // first barrier, which no wave passes before every wave has finished the dazr reads above)
  }
}

// -------------------------------------------------------------------------------------------------
// te_rec_fwd32 / te_rec_bwd32: the recurrent kernels for D = 256 (config X; also instantiable at D = 128).  At D = 256 the
// recurrent weights are 786 KB: the register-resident scheme of the 16-sequence kernels would need 192 registers per wave
// with four waves per SIMD.  Here a workgroup (4 waves) owns a tile of 32 sequences, keeps h_{t-1} / r*h_{t-1} in LDS and
// STREAMS the weights from L2 in MFMA B-fragment order every step (mma_lds_packed, one k-group of prefetch): a step of a
// 32-row tile is 12.6 MFLOP = ~20 us of the CU's matrix pipe against ~5 us to pull 786 KB at 64 B/clk, so the stream
// hides behind the 32x32x2 MFMAs.  Wave w owns hidden columns [D/4 * w, D/4 * (w + 1)) of all three gates, i.e.
// TPW = D/128 n-tiles per gate: the state update is lane-local in the MFMA C layout.  Two barriers per step.
// -------------------------------------------------------------------------------------------------
template <int D, bool predict, int NWV, bool SP = false>      // NWV waves per workgroup: wave w owns hidden columns [D / NWV * w, D / NWV * (w + 1)); SP: split products
__global__ __launch_bounds__(NWV * 64) void te_rec_fwd32_kernel(TeArgs A) {
  extern __shared__ __align__(16) float lds[];
  constexpr int K8 = D / 8, LDA = D + 4, TPW = D / 32 / NWV, NTG = D / 32;       // NTG: n-tiles per gate
  constexpr int LDH = D + 8, PS = 32 * LDH;
  float* Hb = lds;                       // h_{t-1}, overwritten by h_t   32 x LDA
  float* RHb = Hb + 32 * LDA;            // r * h_{t-1}
  unsigned short* Hs = reinterpret_cast<unsigned short*>(lds);      // SP: 3 bf16 planes x 32 x LDH
  unsigned short* RHs = Hs + 3 * PS;
  __shared__ int s_r0[32], s_ns[32];
  const int lane = lane_id(), w = wave_id(), li = lane & 31, tid = threadIdx.x;
  const int tile = blockIdx.x;
  if (tid < 32) {
    const int k = tile * 32 + tid;
    int r0 = 0, ns = 0;
    if (k < A.n_seq) { r0 = A.soff[k]; ns = A.soff[k + 1] - r0; }
    s_r0[tid] = r0; s_ns[tid] = ns;
  }
  if constexpr (SP) { for (int e = tid; e < 3 * PS; e += NWV * 64) Hs[e] = 0; }
  else { for (int e = tid; e < 32 * LDA; e += NWV * 64) Hb[e] = 0.f; }
  lds_barrier();
  int ns_max = 0;
  for (int i = 0; i < 32; ++i) ns_max = max(ns_max, s_ns[i]);
  const int Tsp = A.soff[A.n_seq];       // spare packed row: finished sequences read / write it unconditionally
  int ntzr[2 * TPW], ntc[TPW];
#pragma unroll
  for (int i = 0; i < TPW; ++i) { ntzr[i] = w * TPW + i; ntzr[TPW + i] = NTG + w * TPW + i; ntc[i] = 2 * NTG + w * TPW + i; }
  float hcur[TPW][16];
#pragma unroll
  for (int i = 0; i < TPW; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) hcur[i][r] = 0.f;
  for (int t = 0; t < ns_max; ++t) {
    size_t grow[16];
    bool on[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) { const int i = c_row(r, lane); on[r] = t < s_ns[i]; grow[r] = (size_t)(on[r] ? s_r0[i] + t : Tsp); }
    // pre-activations X.ui^T + bi of this step (te_gemm_ax): fetched now, added after the MFMA loop
    float gz[TPW][16], gr[TPW][16];
#pragma unroll
    for (int i = 0; i < TPW; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float* g = A.G + grow[r] * 3 * D + (w * TPW + i) * 32 + li;
        gz[i][r] = g[0]; gr[i][r] = g[D];
      }
    f32x16 azr[1][2 * TPW];
#pragma unroll
    for (int j = 0; j < 2 * TPW; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) azr[0][j][r] = 0.f;
    if constexpr (SP) mma_lds_packed_s3<2 * TPW, D / 16>(azr, Hs, LDH, A.pWhT16, ntzr);
    else mma_lds_packed<1, 2 * TPW, K8>(azr, Hb, LDA, A.pWhT16, ntzr);
    float zv[TPW][16];
#pragma unroll
    for (int i = 0; i < TPW; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int col = (w * TPW + i) * 32 + li;
        zv[i][r] = fast_sigmoid(azr[0][i][r] + gz[i][r]);
        const float rv = fast_sigmoid(azr[0][TPW + i][r] + gr[i][r]);
        const float rh = rv * hcur[i][r];
        if constexpr (SP) split3_store(RHs + c_row(r, lane) * LDH + col, PS, rh);
        else RHb[c_row(r, lane) * LDA + col] = rh;
        if (!predict) { A.G[grow[r] * 3 * D + D + col] = rv; A.RH[grow[r] * D + col] = rh; }
      }
    float gc[TPW][16];
#pragma unroll
    for (int i = 0; i < TPW; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) gc[i][r] = A.G[grow[r] * 3 * D + 2 * D + (w * TPW + i) * 32 + li];
    lds_barrier();
    f32x16 ac[1][TPW];
#pragma unroll
    for (int j = 0; j < TPW; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) ac[0][j][r] = 0.f;
    if constexpr (SP) mma_lds_packed_s3<TPW, D / 16>(ac, RHs, LDH, A.pWhT16, ntc);
    else mma_lds_packed<1, TPW, K8>(ac, RHb, LDA, A.pWhT16, ntc);
#pragma unroll
    for (int i = 0; i < TPW; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int col = (w * TPW + i) * 32 + li;
        const float c = fast_tanh(ac[0][i][r] + gc[i][r]);
        const float hn = on[r] ? (1.0f - zv[i][r]) * hcur[i][r] + zv[i][r] * c : hcur[i][r];
        if constexpr (SP) split3_store(Hs + c_row(r, lane) * LDH + col, PS, hn);
        else Hb[c_row(r, lane) * LDA + col] = hn;     // nobody reads Hb between the two barriers of a step
        hcur[i][r] = hn;
        if (!predict) { A.G[grow[r] * 3 * D + col] = zv[i][r]; A.G[grow[r] * 3 * D + 2 * D + col] = c; A.H[grow[r] * D + col] = hn; }
      }
    lds_barrier();
  }
  if (predict) {
#pragma unroll
    for (int i = 0; i < TPW; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int k = tile * 32 + c_row(r, lane);
        if (k < A.n_seq) A.hts[(size_t)(A.out_row ? A.out_row[k] : k) * D + (w * TPW + i) * 32 + li] = hcur[i][r];
      }
  }
}

template <int D, int NWV, bool SP = false>
__global__ __launch_bounds__(NWV * 64) void te_rec_bwd32_kernel(TeArgs A) {
  extern __shared__ __align__(16) float lds[];
  constexpr int K8 = D / 8, LDA = D + 4, LDB = 2 * D + 4, TPW = D / 32 / NWV;
  constexpr int LHA = D + 8, LHB = 2 * D + 8;
  float* Ac = lds;                       // da_c           32 x LDA
  float* Azr = Ac + 32 * LDA;            // da_z | da_r    32 x LDB
  unsigned short* Acs = reinterpret_cast<unsigned short*>(lds);      // SP: 3 bf16 planes x 32 x LHA
  unsigned short* Azrs = Acs + 3 * 32 * LHA;                         //     3 planes x 32 x LHB
  __shared__ int s_r0[32], s_ns[32];
  const int lane = lane_id(), w = wave_id(), li = lane & 31, tid = threadIdx.x;
  const int tile = blockIdx.x;
  if (tid < 32) {
    const int k = tile * 32 + tid;
    int r0 = 0, ns = 0;
    if (k < A.n_seq) { r0 = A.soff[k]; ns = A.soff[k + 1] - r0; }
    s_r0[tid] = r0; s_ns[tid] = ns;
  }
  lds_barrier();
  int ns_max = 0;
  for (int i = 0; i < 32; ++i) ns_max = max(ns_max, s_ns[i]);
  const int Tsp = A.soff[A.n_seq];
  int ntw[TPW];
#pragma unroll
  for (int i = 0; i < TPW; ++i) ntw[i] = w * TPW + i;
  float dhn[TPW][16], sbz[TPW], sbr[TPW], sbc[TPW];
#pragma unroll
  for (int i = 0; i < TPW; ++i) {
    sbz[i] = sbr[i] = sbc[i] = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) dhn[i][r] = 0.f;
  }
  for (int t = ns_max - 1; t >= 0; --t) {
    size_t grow[16];
    bool on[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) { const int i = c_row(r, lane); on[r] = t < s_ns[i]; grow[r] = (size_t)(on[r] ? s_r0[i] + t : Tsp); }
    float daz[TPW][16], rv[TPW][16], hp[TPW][16], dhp[TPW][16], dacv[TPW][16];
#pragma unroll
    for (int i = 0; i < TPW; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int col = (w * TPW + i) * 32 + li;
        const float* g = A.G + grow[r] * 3 * D + col;
        // the spare row holds arbitrary bits: select, do not multiply by zero
        const float z = on[r] ? g[0] : 0.f, rr = on[r] ? g[D] : 0.f, c = on[r] ? g[2 * D] : 0.f;
        const float hraw = A.H[(on[r] && t > 0 ? grow[r] - 1 : (size_t)Tsp) * D + col];
        const float h = (on[r] && t > 0) ? hraw : 0.f;
        const float draw = A.DH[grow[r] * D + col];
        const float dh = on[r] ? dhn[i][r] + draw : 0.f;
        rv[i][r] = rr; hp[i][r] = h;
        daz[i][r] = dh * (c - h) * z * (1.0f - z);
        dhp[i][r] = dh * (1.0f - z);
        dacv[i][r] = dh * z * (1.0f - c * c);
        if constexpr (SP) split3_store(Acs + c_row(r, lane) * LHA + col, 32 * LHA, dacv[i][r]);
        else Ac[c_row(r, lane) * LDA + col] = dacv[i][r];
      }
    lds_barrier();
    f32x16 m[1][TPW];
#pragma unroll
    for (int j = 0; j < TPW; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) m[0][j][r] = 0.f;
    if constexpr (SP) mma_lds_packed_s3<TPW, D / 16>(m, Acs, LHA, A.pWhc16, ntw);
    else mma_lds_packed<1, TPW, K8>(m, Ac, LDA, A.pWhc16, ntw);
#pragma unroll
    for (int i = 0; i < TPW; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int col = (w * TPW + i) * 32 + li, ri = c_row(r, lane);
        const float mv = m[0][i][r];
        const float dr = mv * hp[i][r];
        dhp[i][r] += mv * rv[i][r];
        const float dar = dr * rv[i][r] * (1.0f - rv[i][r]);
        if constexpr (SP) {
          split3_store(Azrs + ri * LHB + col, 32 * LHB, daz[i][r]);
          split3_store(Azrs + ri * LHB + D + col, 32 * LHB, dar);
        } else {
          Azr[ri * LDB + col] = daz[i][r];
          Azr[ri * LDB + D + col] = dar;
        }
        float* g = A.G + grow[r] * 3 * D + col;
        g[0] = daz[i][r]; g[D] = dar; g[2 * D] = dacv[i][r];
        sbz[i] += daz[i][r]; sbr[i] += dar; sbc[i] += dacv[i][r];        // zero for inactive steps (dh == 0)
      }
    lds_barrier();
    f32x16 acc[1][TPW];
#pragma unroll
    for (int j = 0; j < TPW; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[0][j][r] = 0.f;
    if constexpr (SP) mma_lds_packed_s3<TPW, 2 * D / 16>(acc, Azrs, LHB, A.pWhzr16, ntw);
    else mma_lds_packed<1, TPW, 2 * K8>(acc, Azr, LDB, A.pWhzr16, ntw);
#pragma unroll
    for (int i = 0; i < TPW; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) dhn[i][r] = on[r] ? dhp[i][r] + acc[0][i][r] : 0.f;
    // (no third barrier: Ac is rewritten only after every wave passed the second barrier, Azr only after the next first one)
  }
  // d bi partials of this tile: a lane holds 16 rows of its columns, the two half-waves the other 16
#pragma unroll
  for (int i = 0; i < TPW; ++i) {
    sbz[i] += __shfl_xor(sbz[i], 32, 64); sbr[i] += __shfl_xor(sbr[i], 32, 64); sbc[i] += __shfl_xor(sbc[i], 32, 64);
    if (lane < 32) {
      float* bp = A.bi_part + (size_t)tile * 3 * D + (w * TPW + i) * 32 + li;
      bp[0] = sbz[i]; bp[D] = sbr[i]; bp[2 * D] = sbc[i];
    }
  }
}

// -------------------------------------------------------------------------------------------------
// te_head: 32 packed rows per iteration (persistent grid).  NBT = number of 32-bin tiles (bins padded).
// mode 0: training - losses, d logits (stored to DL for the d vs job of te_wgrad), DH = d logits . vs
// + g * E, g (for the sorted scatter), d bs / d wd partials; mode 1: predict (sts only, rows =
// sequences, H = hts).  Small enough in registers (no d vs accumulators) and LDS (E never staged) for
// three workgroups per CU, whose MFMA / softmax / staging phases overlap.
// -------------------------------------------------------------------------------------------------
// -DTE_HEAD_PROF: per-phase cycle counts of te_head (tools/head_phases.sh); never defined in the product build
#ifdef TE_HEAD_PROF
__device__ unsigned long long g_head_prof[4][10];
#define HP_INIT long long hp_t = clock64(); long long hp[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#define HP(i) { const long long hp_n = clock64(); hp[i] += hp_n - hp_t; hp_t = hp_n; }
#define HP_END if (lane_id() == 0) { for (int i = 0; i < 10; ++i) atomicAdd(&g_head_prof[wave_id()][i], (unsigned long long)hp[i]); }
#else
#define HP_INIT
#define HP(i)
#define HP_END
#endif
template <int D, int NBT, int MODE>
__global__ __launch_bounds__(TE_BLOCK, (D >= 256 ? 2 : 3)) void te_head_kernel(TeArgs A) {
  extern __shared__ __align__(16) float lds[];
  constexpr int K8 = D / 8, LDH = D + 4, NBP = NBT * 32, LDO = NBP + 4, NTW = (NBT + 3) / 4, NTD = D / 32;
  constexpr int KB8 = NBP / 8, DTW = (NTD + 3) / 4, LPR = D / 4;
  float* Ht = lds;                  // 32 x LDH
  float* Ot = Ht + 32 * LDH;        // 32 x LDO : logits -> softmax -> d logits
  __shared__ float s_g[32], s_he[32], s_red[8];
  __shared__ int s_a[32], s_b[32];   // target bins of the tile's rows
  const int NB = A.n_dist + 1;
  const int T = MODE ? A.n_seq : A.soff[A.n_seq];
  const float* __restrict__ Hsrc = MODE ? A.hts : A.H;
  const float* __restrict__ Esrc = A.E;
  const int lane = lane_id(), w = wave_id(), li = lane & 31, tid = threadIdx.x;
  if ((int)blockIdx.x * 32 >= T) return;        // (grid <= number of tiles: not taken; keeps T >= 1 below)
  float ls0 = 0.f, ls1 = 1.f, wd = 0.f;
  {
    const float a = A.lw[0], b = A.lw[1], m = fmaxf(a, b);
    const float ea = expf(a - m), eb = expf(b - m);
    ls0 = ea / (ea + eb); ls1 = eb / (ea + eb); wd = A.wd[0];
  }
  int nto[NTW];
  float bsv[NTW];                   // bias of this lane's bins (-inf for the padding bins): loop-invariant
#pragma unroll
  for (int j = 0; j < NTW; ++j) {
    nto[j] = min(w + 4 * j, NBT - 1);
    const int bin = nto[j] * 32 + li;
    bsv[j] = bin < NB ? A.bs[bin] : -INFINITY;
  }
  float dbs_acc = 0.f;      // thread tid < NBP accumulates d bs[tid]
  float dwd_acc = 0.f;      // meaningful in the row-owner lanes, reduced at the end

  // The kernel is a latency chain per workgroup (five barriers per tile), so nothing in the loop may wait for
  // a load it has just issued.  Every global access is therefore branch-free - rows past T are clamped to
  // row T - 1 for loads and redirected to the spare row T for stores, and masked where they are consumed -
  // which lets the compiler count its s_waitcnt vmcnt exactly; and the NEXT tile's H / E rows and target bins
  // are fetched in the middle of the current tile (before its second MFMA block) and only touched at the top
  // of the next iteration.
  constexpr int SF4 = 32 * LPR / TE_BLOCK;          // float4 per thread per staged tile
  float4 ph[SF4], pe[SF4];
  int pab = 0;
  auto prefetch = [&](int r0) {
#pragma unroll
    for (int q = 0; q < SF4; ++q) {
      const int e = tid + q * TE_BLOCK, r = e / LPR, c = (e % LPR) * 4;
      const size_t gr = (size_t)min(r0 + r, T - 1);
      ph[q] = *reinterpret_cast<const float4*>(Hsrc + gr * D + c);
      if (!MODE) pe[q] = *reinterpret_cast<const float4*>(Esrc + gr * D + c);
    }
    if (!MODE) pab = A.row_ab[min(r0 + (tid & 31), T - 1)];
  };
  // registers -> LDS (H rows, h . e, target bins).  Called at the END of an iteration for the next tile: the
  // wait for the prefetch then sits in straight-line code behind the stores that followed it and is counted
  // exactly; at the top of the loop it would merge with the first iteration's "just issued" state, i.e. be a
  // vmcnt(0) that also waits for the previous tile's DH stores.
  auto stage = [&](int tid) {
#pragma unroll
    for (int q = 0; q < SF4; ++q) {
      const int e = tid + q * TE_BLOCK, r = e / LPR, c = (e % LPR) * 4;
      *reinterpret_cast<float4*>(Ht + r * LDH + c) = make_float4(ph[q].x, ph[q].y, ph[q].z, ph[q].w);
      if (!MODE) {     // h . e of row r: sum over the LPR lanes that hold the row
        float d = (ph[q].x * pe[q].x + ph[q].y * pe[q].y) + (ph[q].z * pe[q].z + ph[q].w * pe[q].w);
        // (butterfly inside 16 lanes on DPP - VALU latency; only the steps across 16-lane rows go through ds_bpermute: five serial
        // LDS round trips per staged float4 were most of this function)
        d += dpp_f<0xB1>(d); d += dpp_f<0x4E>(d); d += dpp_f<0x141>(d); d += dpp_f<0x140>(d);
#pragma unroll
        for (int o = 16; o < LPR; o <<= 1) d += __shfl_xor(d, o, 64);
        if ((tid % LPR) == 0) s_he[r] = d;
      }
    }
    if (!MODE && tid < 32) { s_a[tid] = pab & 0xffff; s_b[tid] = pab >> 16; }
  };
  prefetch(blockIdx.x * 32);
  stage(tid);
  HP_INIT
  for (int r0 = blockIdx.x * 32; r0 < T; r0 += gridDim.x * 32) {
    // A per-iteration copy of the thread id the compiler cannot see through: the LDS / global addresses derived from it are
    // recomputed in every tile (a few VALU instructions).  As loop invariants they were hoisted, spilled (the kernel sat at its
    // 168-register limit) and reloaded with scratch_load + s_waitcnt vmcnt(0) - a wait for EVERY older vector memory operation:
    // in stage() for the DH stores just issued (13 % of the kernel, tools/head_phases.sh), in the softmax phase for the prefetch.
    // Without them: 153 registers, no spill, 352 -> 329 us.
    int tl = tid;
    asm volatile("" : "+v"(tl));
    const int lane = tl & 63, li = tl & 31;
    lds_barrier();        // staged tile visible; every wave is done with Ot (DH MFMAs of the previous tile)
    HP(0)
    {   // logits
      f32x16 acc[1][NTW];
#pragma unroll
      for (int j = 0; j < NTW; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[0][j][r] = 0.f;
      mma_lds_packed<1, NTW, K8>(acc, Ht, LDH, A.pVsT, nto);
#pragma unroll
      for (int j = 0; j < NTW; ++j) {
        if (w + 4 * j >= NBT) continue;
        const int bin = nto[j] * 32 + li;
#pragma unroll
        for (int r = 0; r < 16; ++r) Ot[c_row(r, lane) * LDO + bin] = acc[0][j][r] + bsv[j];
      }
    }
    HP(1)
    lds_barrier();
    HP(2)
    // next tile's staging rows: requested BEFORE the softmax phase (LDS + VALU only), so that they have landed when the second MFMA
    // block starts waiting for its first B fragment - vmcnt is in order, that wait covers every older load (measured: -3 %)
    prefetch(min(r0 + (int)gridDim.x * 32, T - 1));
    {   // row-wise softmax + losses: 8 lanes per row
      const int row = tl >> 3, sub = tl & 7, gr = r0 + row;
      float* o = Ot + row * LDO;
      float mx = -INFINITY;
#pragma unroll
      for (int i = 0; i < NBP / 8; ++i) mx = fmaxf(mx, o[sub + 8 * i]);     // (uniform trip counts: unrollable)
      mx = fmaxf(mx, dpp_f<0xB1>(mx)); mx = fmaxf(mx, dpp_f<0x4E>(mx)); mx = fmaxf(mx, dpp_f<0x141>(mx));
      // exp pass: e_k stays in LDS un-normalised; the sum and (training) the sum over the bins <= the target come out of the same pass,
      // the probabilities are formed where they are consumed (three passes over the row instead of four)
      const int a = MODE ? 0 : s_a[row], b = MODE ? 0 : s_b[row];
      float sum = 0.f, cum = 0.f;
#pragma unroll
      // (training: v_exp_f32 of the scaled argument, 2 instructions - expf's range reduction and overflow selects are 17, on 28 elements
      // per lane and tile, and f32 VALU time is MFMA pipe time; prediction keeps expf: its probabilities are ranked)
      for (int i = 0; i < NBP / 8; ++i) { const int k = sub + 8 * i; const float e = MODE ? expf(o[k] - mx) : __expf(o[k] - mx); o[k] = e; sum += e; cum += k <= a ? e : 0.f; }
      sum += dpp_f<0xB1>(sum); sum += dpp_f<0x4E>(sum); sum += dpp_f<0x141>(sum);
      const float inv = 1.0f / sum;
      if (MODE) {
        if (gr < T) for (int k = sub; k < NB; k += 8) A.sts[(size_t)gr * NB + k] = o[k] * inv;
      } else {
        const float he = s_he[row];
        const bool live = gr < T;
        cum += dpp_f<0xB1>(cum); cum += dpp_f<0x4E>(cum); cum += dpp_f<0x141>(cum);
        cum *= inv;
        // (the 8 lanes of a row now hold identical cum; they all wrote disjoint o[k])
        __builtin_amdgcn_wave_barrier();
        const float sa = o[a] * inv, sb = o[b] * inv;
        const float u = he + wd * (sa - sb);
        const float g = live ? -ls1 * sigmoidf_(-u) : 0.f;
        const float dot = ls0 * cum - ls0 + g * wd * (sa - sb);
        {   // all 8 lanes of the row store the same values (no lane branch around the stores); dead rows -> row T
          const size_t rs = (size_t)min(gr, T);
          // (reported losses only: v_log_f32 / v_exp_f32 forms, branch-free - logf and the two-sided log1pf(expf()) are ~300 instructions)
          A.rowloss[2 * rs] = cum - __logf(sa);
          A.rowloss[2 * rs + 1] = fminf(u, 0.f) - __logf(1.0f + __expf(-fabsf(u)));
          A.gcoef[rs] = g;
        }
        dwd_acc += sub == 0 ? g * (sa - sb) : 0.f;
        if (sub == 0) s_g[row] = g;
        __builtin_amdgcn_wave_barrier();
        // d logits: the row's e_k are read in one batch and every element is a select chain (with the LDS read inside an `if` the
        // compiler emits a branch, a read and a wait per element).  No select on the element either: a dead row's scale is zero,
        // and a padding bin's e_k is exp(-inf) = 0.
        const float gwd = g * wd, ca = gwd - ls0 / sa, invl = live ? inv : 0.f;
        float pv[NBP / 8];
#pragma unroll
        for (int i = 0; i < NBP / 8; ++i) pv[i] = o[sub + 8 * i];
#pragma unroll
        for (int i = 0; i < NBP / 8; ++i) {
          const int k = sub + 8 * i;
          float ds = (k <= a ? ls0 : 0.f);
          ds = k == a ? ds + ca : ds;
          ds = k == b ? ds - gwd : ds;
          o[k] = (pv[i] * invl) * (ds - dot);
        }
      }
    }
    HP(3)
    lds_barrier();
    HP(4)
    if (!MODE) {
      // g * E of this lane's DH elements: issued now, consumed after the MFMAs
      int ntd[DTW];
#pragma unroll
      for (int j = 0; j < DTW; ++j) ntd[j] = min(w + 4 * j, NTD - 1);
      float ge[DTW][16];
#pragma unroll
      for (int j = 0; j < DTW; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          ge[j][r] = Esrc[(size_t)min(r0 + c_row(r, lane), T - 1) * D + ntd[j] * 32 + li];
      // d logits -> DL (A operand of the d vs job of te_wgrad), d bs partials
      for (int e = tl; e < 32 * (NBP / 4); e += TE_BLOCK) {
        const int r = e / (NBP / 4), c = (e % (NBP / 4)) * 4;
        *reinterpret_cast<float4*>(A.DL + (size_t)min(r0 + r, T) * NBP + c) = *reinterpret_cast<const float4*>(Ot + r * LDO + c);
      }
      if (tid < NBP) {
        float s = 0.f;
#pragma unroll
        for (int r = 0; r < 32; ++r) s += Ot[r * LDO + tid];
        dbs_acc += s;
      }
      HP(5)
      // DH = d logits . vs + g * E
      f32x16 acc[1][DTW];
#pragma unroll
      for (int j = 0; j < DTW; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[0][j][r] = 0.f;
      mma_lds_packed<1, DTW, KB8>(acc, Ot, LDO, A.pVs, ntd);
      HP(6)
#pragma unroll
      for (int j = 0; j < DTW; ++j) {
        if (w + 4 * j >= NTD) continue;
        const int col = ntd[j] * 32 + li;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int i = c_row(r, lane);
          A.DH[(size_t)min(r0 + i, T) * D + col] = acc[0][j][r] + s_g[i] * ge[j][r];
        }
      }
    }
    HP(7)
    stage(tl);            // Ht / s_he / s_a / s_b were last read before the previous barrier
    HP(8)
  }
  HP_END
  if (!MODE) {
    float* hs = A.hslab + (size_t)blockIdx.x * A.hstride;
    if (tid < NB) hs[tid] += dbs_acc;
    const float dw = block_sum(dwd_acc, s_red);
    if (tid == 0) hs[NB] += dw;
  }
}

// -------------------------------------------------------------------------------------------------
// te_head3 (round 4): the training head on split products (poi_ctx_set_split_products; <= 256 bins).  Same tile loop, same LDS budget
// (three workgroups per CU), the float32-input MFMAs - 3/4 of te_head's time at the vector rate - replaced:
//   logits   = h . vs^T: the h tile stays float32 in LDS, a lane cuts its fragment into three bf16 planes as it reads it (mma_f32a_s2);
//   softmax  : a thread keeps its 4 x NBT logits (bins 32 i + 4 sub + e of its row) in REGISTERS through all three passes - one 16-byte LDS
//              read per four bins instead of three passes of 4-byte reads and writes; the target probabilities by compare + DPP sums;
//              d logits go to DL as float4 stores from the registers and to LDS as two bf16 planes (both rounded to nearest) that overwrite
//              the wave's OWN eight float32 rows (8 rows x LDO floats = 2 planes x 8 rows x LDO bf16: no barrier in between);
//   d h      = d logits . vs from the planes (mma_p2_s3); d bs = column sums of the planes.
// vs fragments: te_pack n16 == 4 in both orientations (TeArgs.head_split); the logits read planes 1 and 2, d h all three.
// -------------------------------------------------------------------------------------------------
// EF (round 6, TeArgs.efuse, dim 128): E = lt[p'] - lt[q'] is gathered HERE instead of written by te_gather (118 MB per 12500-user launch) and read
// back twice.  The staging map changes with it: a wave owns 32 COLUMNS of the tile (eight lanes x float4 per row, eight rows per pass) - the h tile
// goes to Ht, the E values stay in registers through the logits and are then parked in the wave's OWN columns of Ht (dead once the logits exist),
// where the DH epilogue - wave w writes column tile w - reads them back in the MFMA's C layout: no second pass over E in global memory, no
// cross-wave hazard on Ht, no extra barrier.  h . E is four per-wave partial dots (s_he4) summed in a fixed order.
template <int D, int NBT, bool EF = false>
__global__ __launch_bounds__(TE_BLOCK, (D >= 256 ? 2 : 3)) void te_head3_kernel(TeArgs A) {
  static_assert(!EF || D == 128, "te_head3<EF>: a wave owns D / 4 = 32 columns = its DH column tile");
  extern __shared__ __align__(16) float lds[];
  constexpr int KG = D / 16, LDH = D + 4, NBP = NBT * 32, LDO = NBP + 8, NTW = (NBT + 3) / 4, NTD = D / 32;
  constexpr int KBG = NBP / 16, DTW = (NTD + 3) / 4, LPR = D / 4;
  float* Ht = lds;                  // 32 x LDH
  float* Ot = Ht + 32 * LDH;        // 32 x LDO logits; then per wave region (8 rows): [2 planes][8 rows][LDO] bf16 d logits
  __shared__ float s_g[32], s_he[32], s_red[8];
  __shared__ float s_he4[EF ? 4 : 1][32];
  __shared__ int s_a[32], s_b[32];   // target bins of the tile's rows
  const int NB = A.n_dist + 1;
  const int T = A.soff[A.n_seq];
  const float* __restrict__ Hsrc = A.H;
  const float* __restrict__ Esrc = A.E;
  const int w = wave_id(), tid = threadIdx.x;
  if ((int)blockIdx.x * 32 >= T) return;        // (grid <= number of tiles: not taken; keeps T >= 1 below)
  float ls0, ls1, wd;
  {
    const float a = A.lw[0], b = A.lw[1], m = fmaxf(a, b);
    const float ea = expf(a - m), eb = expf(b - m);
    ls0 = ea / (ea + eb); ls1 = eb / (ea + eb); wd = A.wd[0];
  }
  int nto[NTW];
  float bsv[NTW];                   // bias of this lane's bins (-inf for the padding bins): loop-invariant
#pragma unroll
  for (int j = 0; j < NTW; ++j) {
    nto[j] = min(w + 4 * j, NBT - 1);
    const int bin = nto[j] * 32 + (tid & 31);
    bsv[j] = bin < NB ? A.bs[bin] : -INFINITY;
  }
  float dbs0 = 0.f, dbs1 = 0.f;     // thread tid < NBP / 2 accumulates d bs[2 tid], d bs[2 tid + 1]
  float dwd_acc = 0.f;
  // (global accesses branch-free, next tile fetched in the middle of the current one, staged at its end: see te_head)
  constexpr int SF4 = 32 * LPR / TE_BLOCK;
  constexpr int EW = 8, ERP = 64 / EW;         // EF: lanes per row inside a wave's 32 columns, rows per pass (SF4 = 32 / ERP passes)
  float4 ph[SF4], pe[SF4], pb[EF ? SF4 : 1];
  int2 pid[EF ? SF4 : 1];
  int pab = 0;
  auto ids = [&](int r0, int tl) {             // EF: the tile's (p', q') pairs, one tile ahead of its rows
#pragma unroll
    for (int q = 0; q < SF4; ++q) pid[q] = A.row_pq[min(r0 + q * ERP + ((tl & 63) / EW), T - 1)];
  };
  auto prefetch = [&](int r0) {
    if constexpr (EF) {
#pragma unroll
      for (int q = 0; q < SF4; ++q) {
        const int r = q * ERP + ((tid & 63) / EW), c = 32 * w + 4 * (tid & (EW - 1));
        const size_t gr = (size_t)min(r0 + r, T - 1);
        ph[q] = *reinterpret_cast<const float4*>(Hsrc + gr * D + c);
        pe[q] = ld4t(A.lt, (size_t)pid[q].x * D + c, A.lt_f16);
        pb[q] = ld4t(A.lt, (size_t)pid[q].y * D + c, A.lt_f16);
      }
    } else {
#pragma unroll
    for (int q = 0; q < SF4; ++q) {
      const int e = tid + q * TE_BLOCK, r = e / LPR, c = (e % LPR) * 4;
      const size_t gr = (size_t)min(r0 + r, T - 1);
      ph[q] = *reinterpret_cast<const float4*>(Hsrc + gr * D + c);
      pe[q] = *reinterpret_cast<const float4*>(Esrc + gr * D + c);
    }
    }
    pab = A.row_ab[min(r0 + (tid & 31), T - 1)];
  };
  auto stage = [&](int tid) {
    if constexpr (EF) {
#pragma unroll
      for (int q = 0; q < SF4; ++q) {
        const int r = q * ERP + ((tid & 63) / EW), c = 32 * (tid >> 6) + 4 * (tid & (EW - 1));
        *reinterpret_cast<float4*>(Ht + r * LDH + c) = make_float4(ph[q].x, ph[q].y, ph[q].z, ph[q].w);
        pe[q] = make_float4(pe[q].x - pb[q].x, pe[q].y - pb[q].y, pe[q].z - pb[q].z, pe[q].w - pb[q].w);
        float d = (ph[q].x * pe[q].x + ph[q].y * pe[q].y) + (ph[q].z * pe[q].z + ph[q].w * pe[q].w);
        d += dpp_f<0xB1>(d); d += dpp_f<0x4E>(d); d += dpp_f<0x141>(d);      // the eight lanes of the row
        if ((tid & (EW - 1)) == 0) s_he4[tid >> 6][r] = d;
      }
    } else {
#pragma unroll
    for (int q = 0; q < SF4; ++q) {
      const int e = tid + q * TE_BLOCK, r = e / LPR, c = (e % LPR) * 4;
      *reinterpret_cast<float4*>(Ht + r * LDH + c) = make_float4(ph[q].x, ph[q].y, ph[q].z, ph[q].w);
      float d = (ph[q].x * pe[q].x + ph[q].y * pe[q].y) + (ph[q].z * pe[q].z + ph[q].w * pe[q].w);
      d += dpp_f<0xB1>(d); d += dpp_f<0x4E>(d); d += dpp_f<0x141>(d); d += dpp_f<0x140>(d);
#pragma unroll
      for (int o = 16; o < LPR; o <<= 1) d += __shfl_xor(d, o, 64);
      if ((tid % LPR) == 0) s_he[r] = d;
    }
    }
    if (tid < 32) { s_a[tid] = pab & 0xffff; s_b[tid] = pab >> 16; }
  };
  if constexpr (EF) ids(blockIdx.x * 32, tid);
  prefetch(blockIdx.x * 32);
  stage(tid);
  for (int r0 = blockIdx.x * 32; r0 < T; r0 += gridDim.x * 32) {
    int tl = tid;                   // (a per-iteration copy the compiler cannot hoist address arithmetic out of: see te_head)
    asm volatile("" : "+v"(tl));
    const int lane = tl & 63, li = tl & 31;
    lds_barrier();        // staged tile visible; every wave is done with the planes (d h MFMAs of the previous tile)
    if constexpr (EF) ids(min(r0 + (int)gridDim.x * 32, T - 1), tl);      // the next tile's (p', q'): its rows are requested behind the logits
    {   // logits
      f32x16 acc[NTW];
#pragma unroll
      for (int j = 0; j < NTW; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
      // (a wave whose last bin tile does not exist - 7 tiles on 4 waves - multiplies one tile less instead of a duplicate: 1/8 of the vs^T stream)
      if (NTW > 1 && w + 4 * (NTW - 1) >= NBT) {
        f32x16 (&acc1)[NTW - 1 > 0 ? NTW - 1 : 1] = reinterpret_cast<f32x16 (&)[NTW - 1 > 0 ? NTW - 1 : 1]>(acc);
        int nto1[NTW - 1 > 0 ? NTW - 1 : 1];
#pragma unroll
        for (int j = 0; j < (NTW - 1 > 0 ? NTW - 1 : 1); ++j) nto1[j] = nto[j];
        mma_f32a_s2<(NTW - 1 > 0 ? NTW - 1 : 1), KG>(acc1, Ht, LDH, A.pVsT, nto1);
      } else mma_f32a_s2<NTW, KG>(acc, Ht, LDH, A.pVsT, nto);
#pragma unroll
      for (int j = 0; j < NTW; ++j) {
        if (w + 4 * j >= NBT) continue;
        const int bin = nto[j] * 32 + li;
#pragma unroll
        for (int r = 0; r < 16; ++r) Ot[c_row(r, lane) * LDO + bin] = acc[j][r] + bsv[j];
      }
    }
    lds_barrier();
    if constexpr (EF) {     // the logits are out: Ht is dead until the next stage - this wave's 32 columns of it take the tile's E values
#pragma unroll
      for (int q = 0; q < SF4; ++q)
        *reinterpret_cast<float4*>(Ht + (q * ERP + ((tl & 63) / EW)) * LDH + 32 * w + 4 * (tl & (EW - 1))) = pe[q];
    }
    prefetch(min(r0 + (int)gridDim.x * 32, T - 1));
    {   // row-wise softmax, losses, d logits: 8 lanes per row, the row's logits in registers
      const int row = tl >> 3, sub = tl & 7, gr = r0 + row;
      const float* o = Ot + row * LDO + 4 * sub;
      float v[NBT][4];
#pragma unroll
      for (int i = 0; i < NBT; ++i) {
        const float4 t = *reinterpret_cast<const float4*>(o + 32 * i);
        v[i][0] = t.x; v[i][1] = t.y; v[i][2] = t.z; v[i][3] = t.w;
      }
      float mx = -INFINITY;
#pragma unroll
      for (int i = 0; i < NBT; ++i) mx = fmaxf(fmaxf(mx, fmaxf(v[i][0], v[i][1])), fmaxf(v[i][2], v[i][3]));
      mx = fmaxf(mx, dpp_f<0xB1>(mx)); mx = fmaxf(mx, dpp_f<0x4E>(mx)); mx = fmaxf(mx, dpp_f<0x141>(mx));
      const int a = s_a[row], b = s_b[row];
      float sum = 0.f, cum = 0.f, ea = 0.f, eb = 0.f;
#pragma unroll
      for (int i = 0; i < NBT; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int k = 32 * i + 4 * sub + e;
          const float x = __expf(v[i][e] - mx);      // (a padding bin's logit is -inf: 0)
          v[i][e] = x; sum += x; cum += k <= a ? x : 0.f; ea += k == a ? x : 0.f; eb += k == b ? x : 0.f;
        }
      sum += dpp_f<0xB1>(sum); sum += dpp_f<0x4E>(sum); sum += dpp_f<0x141>(sum);
      cum += dpp_f<0xB1>(cum); cum += dpp_f<0x4E>(cum); cum += dpp_f<0x141>(cum);
      ea += dpp_f<0xB1>(ea); ea += dpp_f<0x4E>(ea); ea += dpp_f<0x141>(ea);
      eb += dpp_f<0xB1>(eb); eb += dpp_f<0x4E>(eb); eb += dpp_f<0x141>(eb);
      const float inv = 1.0f / sum;
      const float he = EF ? (s_he4[0][row] + s_he4[EF ? 1 : 0][row]) + (s_he4[EF ? 2 : 0][row] + s_he4[EF ? 3 : 0][row]) : s_he[row];
      const bool live = gr < T;
      cum *= inv;
      const float sa = ea * inv, sb = eb * inv;
      const float u = he + wd * (sa - sb);
      const float g = live ? -ls1 * sigmoidf_(-u) : 0.f;
      const float dot = ls0 * cum - ls0 + g * wd * (sa - sb);
      const size_t rs = (size_t)min(gr, T);      // all 8 lanes of the row store the same values; dead rows -> row T
      A.rowloss[2 * rs] = cum - __logf(sa);
      A.rowloss[2 * rs + 1] = fminf(u, 0.f) - __logf(1.0f + __expf(-fabsf(u)));
      A.gcoef[rs] = g;
      dwd_acc += sub == 0 ? g * (sa - sb) : 0.f;
      if (sub == 0) s_g[row] = g;
      const float gwd = g * wd, ca = gwd - ls0 / sa, invl = live ? inv : 0.f;
      // the planes overwrite the float32 rows of THIS wave only, which all its lanes have read above
      unsigned short* pr = reinterpret_cast<unsigned short*>(Ot + (row & ~7) * LDO) + (row & 7) * LDO + 4 * sub;
      float* dlr = A.DL + rs * NBP + 4 * sub;
#pragma unroll
      for (int i = 0; i < NBT; ++i) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int k = 32 * i + 4 * sub + e;
          float ds = (k <= a ? ls0 : 0.f);
          ds = k == a ? ds + ca : ds;
          ds = k == b ? ds - gwd : ds;
          v[i][e] = (v[i][e] * invl) * (ds - dot);
        }
        *reinterpret_cast<float4*>(dlr + 32 * i) = make_float4(v[i][0], v[i][1], v[i][2], v[i][3]);
        uint2 p1, p2;
        wg_split2r(v[i][0], v[i][1], p1.x, p2.x); wg_split2r(v[i][2], v[i][3], p1.y, p2.y);
        *reinterpret_cast<uint2*>(pr + 32 * i) = p1;
        *reinterpret_cast<uint2*>(pr + 8 * LDO + 32 * i) = p2;
      }
    }
    lds_barrier();
    {
      // g * E of this lane's DH elements: issued now, consumed after the MFMAs
      int ntd[DTW];
#pragma unroll
      for (int j = 0; j < DTW; ++j) ntd[j] = min(w + 4 * j, NTD - 1);
      float ge[DTW][16];
      if constexpr (!EF) {
#pragma unroll
      for (int j = 0; j < DTW; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          ge[j][r] = Esrc[(size_t)min(r0 + c_row(r, lane), T - 1) * D + ntd[j] * 32 + li];
      }
      if (tl < NBP / 2) {      // d bs partials: column sums of the two planes (their sum is the float32 value to 2^-18), two bins per thread
        const unsigned* q = reinterpret_cast<const unsigned*>(Ot) + tl;      // (a plane row is LDO / 2 dwords)
        float s0 = 0.f, s1 = 0.f, t0 = 0.f, t1 = 0.f;
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
          for (int r = 0; r < 8; ++r) {
            const unsigned u1 = q[(g * 8 * LDO * 2 + r * LDO) / 2], u2 = q[(g * 8 * LDO * 2 + (8 + r) * LDO) / 2];
            s0 += __uint_as_float(u1 << 16); s1 += __uint_as_float(u1 & 0xFFFF0000u);
            t0 += __uint_as_float(u2 << 16); t1 += __uint_as_float(u2 & 0xFFFF0000u);
          }
        dbs0 += s0 + t0; dbs1 += s1 + t1;
      }
      f32x16 acc[DTW];
#pragma unroll
      for (int j = 0; j < DTW; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
      const unsigned short* arow = reinterpret_cast<const unsigned short*>(Ot + (li & ~7) * LDO) + (li & 7) * LDO + 8 * (lane >> 5);
      mma_p2_s3<DTW, KBG>(acc, arow, 8 * LDO, A.pVs, ntd, (NB + 15) / 16);
      if constexpr (EF) {     // the tile's E values, parked in this wave's columns of Ht (DTW == 1, ntd[0] == w)
#pragma unroll
        for (int r = 0; r < 16; ++r) ge[0][r] = Ht[c_row(r, lane) * LDH + 32 * w + li];
      }
#pragma unroll
      for (int j = 0; j < DTW; ++j) {
        if (w + 4 * j >= NTD) continue;
        const int col = ntd[j] * 32 + li;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int i = c_row(r, lane);
          A.DH[(size_t)min(r0 + i, T) * D + col] = acc[j][r] + s_g[i] * ge[j][r];
        }
      }
    }
    stage(tl);            // Ht / s_he / s_a / s_b were last read before the previous barrier
  }
  {
    float* hs = A.hslab + (size_t)blockIdx.x * A.hstride;
    if (tid < NBP / 2) {
      if (2 * tid < NB) hs[2 * tid] += dbs0;
      if (2 * tid + 1 < NB) hs[2 * tid + 1] += dbs1;
    }
    const float dw = block_sum(dwd_acc, s_red);
    if (tid == 0) hs[NB] += dw;
  }
}

// -------------------------------------------------------------------------------------------------
// te_head_big: the same head for MORE THAN 256 distance bins (the reference's dd = 25 m configuration has 1520,
// public/GRU_Spatial.py:247): a 32 x (n_dist + 1) logits tile no longer fits LDS, so the bins go through it in chunks of 256 -
//   pass A  logits chunk -> LDS -> per-lane running (max, sum of exp, sum of exp over bins <= a) + the two target logits:
//           an online softmax, combined across the 8 lanes of a row after the last chunk;
//   pass B  logits chunk again (recomputed: cheaper than an HBM round trip of the 32 x 1536 tile) -> probabilities ->
//           d logits -> DL chunk, d bs, DH += d logits . vs[chunk]   (mode 1: probabilities -> sts).
// Bins are padded to a multiple of 256 (te_nbp_dev); everything else as te_head_kernel.
// -------------------------------------------------------------------------------------------------
template <int MT, int NTW, int K8>
__device__ __forceinline__ void mma_lds_packed_s(f32x16 (&acc)[MT][NTW], const float* __restrict__ ldsA, int lda,
                                                 const float4* __restrict__ bp, const int (&nt)[NTW], int kstride) {
  // mma_lds_packed over K8 k-groups of a packed operand whose n-tiles are `kstride` k-groups apart (a K-slice of a longer contraction)
  const int lane = lane_id(), li = lane & 31, h = lane >> 5;
  const float* arow = ldsA + li * lda + 4 * h;
  const float4* bj[NTW];
  float4 bc[NTW], bn[NTW], ac[MT], an[MT];
#pragma unroll
  for (int j = 0; j < NTW; ++j) { bj[j] = bp + ((size_t)nt[j] * kstride) * 64 + lane; bc[j] = *bj[j]; }
#pragma unroll
  for (int i = 0; i < MT; ++i) ac[i] = *reinterpret_cast<const float4*>(arow + (size_t)i * 32 * lda);
#pragma unroll 2
  for (int m = 0; m < K8; ++m) {
    const int mn = m + 1 < K8 ? m + 1 : m;
#pragma unroll
    for (int j = 0; j < NTW; ++j) bn[j] = bj[j][(size_t)mn * 64];
#pragma unroll
    for (int i = 0; i < MT; ++i) an[i] = *reinterpret_cast<const float4*>(arow + (size_t)i * 32 * lda + 8 * mn);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NTW; ++j) {
        acc[i][j] = mfma32(ac[i].x, bc[j].x, acc[i][j]);
        acc[i][j] = mfma32(ac[i].y, bc[j].y, acc[i][j]);
        acc[i][j] = mfma32(ac[i].z, bc[j].z, acc[i][j]);
        acc[i][j] = mfma32(ac[i].w, bc[j].w, acc[i][j]);
      }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < NTW; ++j) bc[j] = bn[j];
#pragma unroll
    for (int i = 0; i < MT; ++i) ac[i] = an[i];
  }
}

template <int D, int MODE>
__global__ __launch_bounds__(TE_BLOCK, (D >= 256 ? 2 : 3)) void te_head_big_kernel(TeArgs A) {
  extern __shared__ __align__(16) float lds[];
  constexpr int K8 = D / 8, LDH = D + 4, CH = 256, LDO = CH + 4, NTD = D / 32, DTW = (NTD + 3) / 4, LPR = D / 4;
  const int NB = A.n_dist + 1, NBP = te_nbp_dev(A.n_dist), NCH = NBP / CH, KB8 = NBP / 8;
  float* Ht = lds;                  // 32 x LDH
  float* Ot = Ht + 32 * LDH;        // 32 x LDO : one 256-bin chunk of logits -> d logits
  float* s_dbs = Ot + 32 * LDO;     // NBP: d bs partial of this workgroup
  __shared__ float s_g[32], s_he[32], s_red[8];
  __shared__ int s_a[32], s_b[32];
  const int T = MODE ? A.n_seq : A.soff[A.n_seq];
  const float* __restrict__ Hsrc = MODE ? A.hts : A.H;
  const float* __restrict__ Esrc = A.E;
  const int w = wave_id();
  int tid = threadIdx.x, lane = tid & 63, li = tid & 31;      // (re-derived per tile, see the loop)
  if ((int)blockIdx.x * 32 >= T) return;
  float ls0 = 0.f, ls1 = 1.f, wd = 0.f;
  {
    const float a = A.lw[0], b = A.lw[1], m = fmaxf(a, b);
    const float ea = expf(a - m), eb = expf(b - m);
    ls0 = ea / (ea + eb); ls1 = eb / (ea + eb); wd = A.wd[0];
  }
  for (int e = tid; e < NBP; e += TE_BLOCK) s_dbs[e] = 0.f;
  float dwd_acc = 0.f;
  constexpr int SF4 = 32 * LPR / TE_BLOCK;
  float4 ph[SF4], pe[SF4];
  int pab = 0;
  auto prefetch = [&](int r0) {
#pragma unroll
    for (int q = 0; q < SF4; ++q) {
      const int e = tid + q * TE_BLOCK, r = e / LPR, c = (e % LPR) * 4;
      const size_t gr = (size_t)min(r0 + r, T - 1);
      ph[q] = *reinterpret_cast<const float4*>(Hsrc + gr * D + c);
      if (!MODE) pe[q] = *reinterpret_cast<const float4*>(Esrc + gr * D + c);
    }
    if (!MODE) pab = A.row_ab[min(r0 + (tid & 31), T - 1)];
  };
  auto stage = [&]() {
#pragma unroll
    for (int q = 0; q < SF4; ++q) {
      const int e = tid + q * TE_BLOCK, r = e / LPR, c = (e % LPR) * 4;
      *reinterpret_cast<float4*>(Ht + r * LDH + c) = make_float4(ph[q].x, ph[q].y, ph[q].z, ph[q].w);
      if (!MODE) {
        float d = (ph[q].x * pe[q].x + ph[q].y * pe[q].y) + (ph[q].z * pe[q].z + ph[q].w * pe[q].w);
#pragma unroll
        for (int o = 1; o < LPR; o <<= 1) d += __shfl_xor(d, o, 64);
        if ((tid % LPR) == 0) s_he[r] = d;
      }
    }
    if (!MODE && tid < 32) { s_a[tid] = pab & 0xffff; s_b[tid] = (pab >> 16) & 0xffff; }
  };
  // logits of chunk c -> Ot (bias added; padding bins -inf)
  auto logits = [&](int c) {
    int nto[2];
    float bsv[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      nto[j] = c * 8 + w + 4 * j;
      const int bin = nto[j] * 32 + li;
      bsv[j] = A.bs[min(bin, NB - 1)];
      bsv[j] = bin < NB ? bsv[j] : -INFINITY;
    }
    f32x16 acc[1][2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[0][j][r] = 0.f;
    mma_lds_packed<1, 2, K8>(acc, Ht, LDH, A.pVsT, nto);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = (w + 4 * j) * 32 + li;
#pragma unroll
      for (int r = 0; r < 16; ++r) Ot[c_row(r, lane) * LDO + col] = acc[0][j][r] + bsv[j];
    }
  };
  prefetch(blockIdx.x * 32);
  stage();
  for (int r0 = blockIdx.x * 32; r0 < T; r0 += gridDim.x * 32) {
    // (per-iteration copy of the thread id, as in te_head: address arithmetic is redone per tile instead of hoisted, spilled and
    // reloaded behind s_waitcnt vmcnt(0); the lambdas above capture tid / lane / li by reference)
    tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    lane = tid & 63; li = tid & 31;
    lds_barrier();
    const int row = tid >> 3, sub = tid & 7, gr = r0 + row;
    const int a = MODE ? 0 : s_a[row], b = MODE ? 0 : s_b[row];
    float* o = Ot + row * LDO;
    // ---- pass A: online softmax statistics of this lane's bins (k = sub mod 8) ----
    float m_l = -INFINITY, S_l = 0.f, C_l = 0.f, la = 0.f, lb = 0.f;
    for (int c = 0; c < NCH; ++c) {
      logits(c);
      lds_barrier();
      float mc = -INFINITY;
#pragma unroll
      for (int i = 0; i < CH / 8; ++i) mc = fmaxf(mc, o[sub + 8 * i]);
      const float mn = fmaxf(m_l, mc);
      if (mn > -INFINITY) {                     // (a lane whose bins so far are all padding keeps its empty state)
        const float sc = expf(m_l - mn);        // exp(-inf) = 0 on the first chunk
        float s = 0.f, cs = 0.f;
#pragma unroll
        for (int i = 0; i < CH / 8; ++i) {
          const int k = c * CH + sub + 8 * i;
          const float l = o[sub + 8 * i], e = expf(l - mn);
          s += e; cs += k <= a ? e : 0.f;
          la = k == a ? l : la; lb = k == b ? l : lb;
        }
        S_l = S_l * sc + s; C_l = C_l * sc + cs; m_l = mn;
      }
      lds_barrier();                            // Ot is rewritten by the next chunk / pass
    }
    float mx = m_l;
    mx = fmaxf(mx, dpp_f<0xB1>(mx)); mx = fmaxf(mx, dpp_f<0x4E>(mx)); mx = fmaxf(mx, dpp_f<0x141>(mx));
    const float rs = m_l > -INFINITY ? expf(m_l - mx) : 0.f;
    float sum = S_l * rs, cum = C_l * rs;
    sum += dpp_f<0xB1>(sum); sum += dpp_f<0x4E>(sum); sum += dpp_f<0x141>(sum);
    cum += dpp_f<0xB1>(cum); cum += dpp_f<0x4E>(cum); cum += dpp_f<0x141>(cum);
    la += dpp_f<0xB1>(la); la += dpp_f<0x4E>(la); la += dpp_f<0x141>(la);      // exactly one lane of the row captured each
    lb += dpp_f<0xB1>(lb); lb += dpp_f<0x4E>(lb); lb += dpp_f<0x141>(lb);
    const float inv = 1.0f / sum;
    float g = 0.f, dot = 0.f, sa = 1.f;
    const bool live = gr < T;
    if (!MODE) {
      cum *= inv;
      sa = expf(la - mx) * inv;
      const float sb = expf(lb - mx) * inv;
      const float he = s_he[row];
      const float u = he + wd * (sa - sb);
      g = live ? -ls1 * sigmoidf_(-u) : 0.f;
      dot = ls0 * cum - ls0 + g * wd * (sa - sb);
      {
        const size_t rsx = (size_t)min(gr, T);
        A.rowloss[2 * rsx] = cum - logf(sa);
        A.rowloss[2 * rsx + 1] = log_sigmoidf_(u);
        A.gcoef[rsx] = g;
      }
      dwd_acc += sub == 0 ? g * (sa - sb) : 0.f;
      if (sub == 0) s_g[row] = g;
    }
    // ---- pass B ----
    int ntd[DTW];
#pragma unroll
    for (int j = 0; j < DTW; ++j) ntd[j] = min(w + 4 * j, NTD - 1);
    prefetch(min(r0 + (int)gridDim.x * 32, T - 1));
    f32x16 dh[1][DTW];
#pragma unroll
    for (int j = 0; j < DTW; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) dh[0][j][r] = 0.f;
    for (int c = 0; c < NCH; ++c) {
      logits(c);
      lds_barrier();
#pragma unroll
      for (int i = 0; i < CH / 8; ++i) {
        const int k = c * CH + sub + 8 * i;
        const float s = expf(o[sub + 8 * i] - mx) * inv;
        if (MODE) {
          if (live && k < NB) A.sts[(size_t)gr * NB + k] = s;
        } else {
          float ds = (k <= a ? ls0 : 0.f);
          if (k == a) ds += g * wd - ls0 / sa;
          if (k == b) ds -= g * wd;
          o[sub + 8 * i] = (live && k < NB) ? s * (ds - dot) : 0.f;
        }
      }
      lds_barrier();
      if (!MODE) {
        for (int e = tid; e < 32 * (CH / 4); e += TE_BLOCK) {
          const int r = e / (CH / 4), cc = (e % (CH / 4)) * 4;
          *reinterpret_cast<float4*>(A.DL + (size_t)min(r0 + r, T) * NBP + c * CH + cc) = *reinterpret_cast<const float4*>(Ot + r * LDO + cc);
        }
        {
          float sd = 0.f;
#pragma unroll
          for (int r = 0; r < 32; ++r) sd += Ot[r * LDO + tid];
          s_dbs[c * CH + tid] += sd;              // (thread tid owns bin c * 256 + tid)
        }
        mma_lds_packed_s<1, DTW, CH / 8>(dh, Ot, LDO, A.pVs + (size_t)c * (CH / 8) * 64, ntd, KB8);
        lds_barrier();                            // the next chunk's logits overwrite Ot
      }
    }
    if (!MODE) {
#pragma unroll
      for (int j = 0; j < DTW; ++j) {
        if (w + 4 * j >= NTD) continue;
        const int col = ntd[j] * 32 + li;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int i = c_row(r, lane);
          // (g * E fetched here, after the chunk loop: registers; the kernel is MFMA-bound over 2 NCH + NCH products per tile)
          const float ge = Esrc[(size_t)min(r0 + i, T - 1) * D + col];
          A.DH[(size_t)min(r0 + i, T) * D + col] = dh[0][j][r] + s_g[i] * ge;
        }
      }
    }
    stage();
  }
  if (!MODE) {
    __syncthreads();
    float* hs = A.hslab + (size_t)blockIdx.x * A.hstride;
    for (int e = tid; e < NB; e += TE_BLOCK) hs[e] += s_dbs[e];
    const float dw = block_sum(dwd_acc, s_red);
    if (tid == 0) hs[NB] += dw;
  }
}

// exp(x) for x <= 0 in six instructions: x log2(e) as a head and a tail (the fma recovers the product's rounding error, the second term
// log2(e)'s), v_exp_f32 on the head, first-order correction for the tail; ~1 ulp like expf, without its range reduction and denormal
// scaling - results below 2^-126 flush to zero (softmax terms), -inf and anything below -150 give exactly 0.
__device__ __forceinline__ float exp_neg(float x) {
  x = fmaxf(x, -150.f);
  const float hi = x * 1.44269502162933349609375f;
  const float lo = fmaf(x, 1.925963033500e-8f, fmaf(x, 1.44269502162933349609375f, -hi));
  const float r = __builtin_amdgcn_exp2f(hi);
  return fmaf(r, lo * 0.693147182464599609375f, r);
}
// te_head_big on split products (poi_ctx_set_split_products, the default): every float32 product formed from three bf16 planes per
// operand (six v_mfma_f32_32x32x16_bf16 per 16 k, float32 accumulate: 2.7x the float32 matrix rate).  h is split when the tile is
// staged.  Training: pass A leaves every logits chunk in DL, pass B reads it back (same thread: four adjacent bins x eight), turns it
// into d logits in place and splits them into three bf16 planes that ALIAS pass A's float32 chunk - 77 KB per workgroup at D = 128,
// two workgroups per CU - for the d h product; d bs is summed from the planes (their sum is the float32 value to 2^-24) straight
// into the workgroup's slab.  Predict: the logits are recomputed for the probabilities.  vs packed as te_pack n16 == 4 in both
// orientations.
#ifndef TE_HB3_PFL
#define TE_HB3_PFL 2      // k groups of B in flight: logits (two n tiles per wave), DH (one)
#define TE_HB3_PFD 3
#endif
template <int D, int MODE>
__global__ __launch_bounds__(TE_BLOCK, (D >= 256 ? 1 : 2)) void te_head_big3_kernel(TeArgs A) {
  extern __shared__ __align__(16) float lds[];
  constexpr int KG = D / 16, LDH = D + 8, CH = 256, LDO = CH + 4, LDP = CH + 8, NTD = D / 32, DTW = (NTD + 3) / 4, LPR = D / 4;
  const int NB = A.n_dist + 1, NBP = te_nbp_dev(A.n_dist), NCH = NBP / CH, KBG = NBP / 16;
  unsigned short* Hp = reinterpret_cast<unsigned short*>(lds);      // 3 planes x 32 x LDH
  float* Ot = reinterpret_cast<float*>(Hp + 3 * 32 * LDH);          // 32 x LDO: one 256-bin chunk of logits ...
  unsigned short* Op = reinterpret_cast<unsigned short*>(Ot);        // ... then 3 planes x 32 x LDP of d logits
  __shared__ float s_g[32], s_he[32], s_red[8];
  __shared__ int s_a[32], s_b[32];
  const int T = MODE ? A.n_seq : A.soff[A.n_seq];
  const float* __restrict__ Hsrc = MODE ? A.hts : A.H;
  const float* __restrict__ Esrc = A.E;
  const int w = wave_id();
  int tid = threadIdx.x, lane = tid & 63, li = tid & 31;
  if ((int)blockIdx.x * 32 >= T) return;
  float ls0 = 0.f, ls1 = 1.f, wd = 0.f;
  {
    const float a = A.lw[0], b = A.lw[1], m = fmaxf(a, b);
    const float ea = expf(a - m), eb = expf(b - m);
    ls0 = ea / (ea + eb); ls1 = eb / (ea + eb); wd = A.wd[0];
  }
  float* hs = A.hslab + (size_t)blockIdx.x * A.hstride;
  float dwd_acc = 0.f;
  constexpr int SF4 = 32 * LPR / TE_BLOCK;
  float4 ph[SF4], pe[SF4];
  int pab = 0;
  auto prefetch = [&](int r0) {
#pragma unroll
    for (int q = 0; q < SF4; ++q) {
      const int e = tid + q * TE_BLOCK, r = e / LPR, c = (e % LPR) * 4;
      const size_t gr = (size_t)min(r0 + r, T - 1);
      ph[q] = *reinterpret_cast<const float4*>(Hsrc + gr * D + c);
      if (!MODE) pe[q] = *reinterpret_cast<const float4*>(Esrc + gr * D + c);
    }
    if (!MODE) pab = A.row_ab[min(r0 + (tid & 31), T - 1)];
  };
  auto stage = [&]() {
#pragma unroll
    for (int q = 0; q < SF4; ++q) {
      const int e = tid + q * TE_BLOCK, r = e / LPR, c = (e % LPR) * 4;
      unsigned u[4][3];
      split3(ph[q].x, u[0][0], u[0][1], u[0][2]); split3(ph[q].y, u[1][0], u[1][1], u[1][2]);
      split3(ph[q].z, u[2][0], u[2][1], u[2][2]); split3(ph[q].w, u[3][0], u[3][1], u[3][2]);
#pragma unroll
      for (int p = 0; p < 3; ++p)
        *reinterpret_cast<uint2*>(Hp + (p * 32 + r) * LDH + c) = make_uint2((u[0][p] >> 16) | (u[1][p] & 0xFFFF0000u), (u[2][p] >> 16) | (u[3][p] & 0xFFFF0000u));
      if (!MODE) {
        float d = (ph[q].x * pe[q].x + ph[q].y * pe[q].y) + (ph[q].z * pe[q].z + ph[q].w * pe[q].w);
#pragma unroll
        for (int o = 1; o < LPR; o <<= 1) d += __shfl_xor(d, o, 64);
        if ((tid % LPR) == 0) s_he[r] = d;
      }
    }
    if (!MODE && tid < 32) { s_a[tid] = pab & 0xffff; s_b[tid] = (pab >> 16) & 0xffff; }
  };
  auto logits = [&](int c) {
    int nto[2];
    float bsv[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      nto[j] = c * 8 + w + 4 * j;
      const int bin = nto[j] * 32 + li;
      bsv[j] = A.bs[min(bin, NB - 1)];
      bsv[j] = bin < NB ? bsv[j] : -INFINITY;
    }
    f32x16 acc[1][2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[0][j][r] = 0.f;
    mma_lds_packed_s3p<2, KG, TE_HB3_PFL>(acc, Hp, LDH, A.pVsT, nto);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = (w + 4 * j) * 32 + li;
#pragma unroll
      for (int r = 0; r < 16; ++r) Ot[c_row(r, lane) * LDO + col] = acc[0][j][r] + bsv[j];
    }
  };
  HP_INIT
  prefetch(blockIdx.x * 32);
  stage();
  for (int r0 = blockIdx.x * 32; r0 < T; r0 += gridDim.x * 32) {
    tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    lane = tid & 63; li = tid & 31;
    lds_barrier();
    HP(0)
    const int row = tid >> 3, sub = tid & 7, gr = r0 + row;
    const int a = MODE ? 0 : s_a[row], b = MODE ? 0 : s_b[row];
    const float* o = Ot + row * LDO;
    // ---- pass A: online softmax statistics of this lane's bins (k = sub mod 8) ----
    float m_l = -INFINITY, S_l = 0.f, C_l = 0.f, la = 0.f, lb = 0.f;
    for (int c = 0; c < NCH; ++c) {
      logits(c);
      HP(1)
      lds_barrier();
      HP(2)
      float mc = -INFINITY;
#pragma unroll
      for (int i = 0; i < CH / 8; ++i) mc = fmaxf(mc, o[sub + 8 * i]);
      const float mn = fmaxf(m_l, mc);
      if (mn > -INFINITY) {
        const float sc = exp_neg(m_l - mn);     // exp(-inf) = 0 on the first chunk
        float s = 0.f, cs = 0.f;
#pragma unroll
        for (int i = 0; i < CH / 8; ++i) {
          const int k = c * CH + sub + 8 * i;
          const float e = exp_neg(o[sub + 8 * i] - mn);
          s += e; cs += k <= a ? e : 0.f;
        }
        S_l = S_l * sc + s; C_l = C_l * sc + cs; m_l = mn;
      }
      if (!MODE) {                              // the two target logits, read by every lane of the row
        if ((unsigned)(a - c * CH) < (unsigned)CH) la = o[a - c * CH];
        if ((unsigned)(b - c * CH) < (unsigned)CH) lb = o[b - c * CH];
        // the chunk's logits wait in DL for pass B (the thread that stores them reads them back: no visibility question), which
        // overwrites them with the d logits - a 2 x 6 KB round trip per row instead of a third pass over vs
        float* dlr = A.DL + (size_t)min(gr, T) * NBP + c * CH + 4 * sub;
#pragma unroll
        for (int i = 0; i < CH / 32; ++i) *reinterpret_cast<float4*>(dlr + 32 * i) = *reinterpret_cast<const float4*>(o + 4 * sub + 32 * i);
      }
      HP(3)
      lds_barrier();
      HP(4)
    }
    float mx = m_l;
    mx = fmaxf(mx, dpp_f<0xB1>(mx)); mx = fmaxf(mx, dpp_f<0x4E>(mx)); mx = fmaxf(mx, dpp_f<0x141>(mx));
    const float rs = m_l > -INFINITY ? exp_neg(m_l - mx) : 0.f;
    float sum = S_l * rs, cum = C_l * rs;
    sum += dpp_f<0xB1>(sum); sum += dpp_f<0x4E>(sum); sum += dpp_f<0x141>(sum);
    cum += dpp_f<0xB1>(cum); cum += dpp_f<0x4E>(cum); cum += dpp_f<0x141>(cum);
    const float inv = 1.0f / sum;
    float g = 0.f, dot = 0.f, sa = 1.f;
    const bool live = gr < T;
    if (!MODE) {
      cum *= inv;
      sa = expf(la - mx) * inv;
      const float sb = expf(lb - mx) * inv;
      const float he = s_he[row];
      const float u = he + wd * (sa - sb);
      g = live ? -ls1 * sigmoidf_(-u) : 0.f;
      dot = ls0 * cum - ls0 + g * wd * (sa - sb);
      {
        const size_t rsx = (size_t)min(gr, T);
        A.rowloss[2 * rsx] = cum - logf(sa);
        A.rowloss[2 * rsx + 1] = log_sigmoidf_(u);
        A.gcoef[rsx] = g;
      }
      dwd_acc += sub == 0 ? g * (sa - sb) : 0.f;
      if (sub == 0) s_g[row] = g;
    }
    // ---- pass B ----
    int ntd[DTW];
#pragma unroll
    for (int j = 0; j < DTW; ++j) ntd[j] = min(w + 4 * j, NTD - 1);
    prefetch(min(r0 + (int)gridDim.x * 32, T - 1));
    f32x16 dh[1][DTW];
#pragma unroll
    for (int j = 0; j < DTW; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) dh[0][j][r] = 0.f;
    if constexpr (MODE) {
      for (int c = 0; c < NCH; ++c) {             // predict: logits again -> probabilities
        logits(c);
        lds_barrier();
#pragma unroll
        for (int i = 0; i < CH / 8; ++i) {
          const int k = c * CH + sub + 8 * i;
          const float s = exp_neg(o[sub + 8 * i] - mx) * inv;
          if (live && k < NB) A.sts[(size_t)gr * NB + k] = s;
        }
        lds_barrier();
      }
    } else {
      for (int c = 0; c < NCH; ++c) {
        // this thread's 32 logits of the chunk, as pass A left them: bins 32 i + 4 sub + e
        float* dlr = A.DL + (size_t)min(gr, T) * NBP + c * CH + 4 * sub;
        float4 lv[CH / 32];
#pragma unroll
        for (int i = 0; i < CH / 32; ++i) lv[i] = *reinterpret_cast<const float4*>(dlr + 32 * i);
        HP(5)
        unsigned short* op = Op + row * LDP + 4 * sub;
#pragma unroll
        for (int i = 0; i < CH / 32; ++i) {
          float v[4] = {lv[i].x, lv[i].y, lv[i].z, lv[i].w};
          unsigned u[4][3];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int k = c * CH + 32 * i + 4 * sub + e;
            const float s = exp_neg(v[e] - mx) * inv;
            float ds = (k <= a ? ls0 : 0.f);
            if (k == a) ds += g * wd - ls0 / sa;
            if (k == b) ds -= g * wd;
            v[e] = (live && k < NB) ? s * (ds - dot) : 0.f;
            split3(v[e], u[e][0], u[e][1], u[e][2]);
          }
          *reinterpret_cast<float4*>(dlr + 32 * i) = make_float4(v[0], v[1], v[2], v[3]);
#pragma unroll
          for (int p = 0; p < 3; ++p)
            *reinterpret_cast<uint2*>(op + p * 32 * LDP + 32 * i) = make_uint2((u[0][p] >> 16) | (u[1][p] & 0xFFFF0000u), (u[2][p] >> 16) | (u[3][p] & 0xFFFF0000u));
        }
        lds_barrier();
        HP(7)
        {
          // (thread tid owns bin c * 256 + tid: two rows of a plane per v_dot2c_f32_bf16 against (1, 1) - the bf16 pair goes in as it is)
          typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
          const bf16x2 ones = __builtin_bit_cast(bf16x2, 0x3f803f80u);
          float sd[3] = {0.f, 0.f, 0.f};
#pragma unroll 8
          for (int r = 0; r < 32; r += 2) {
            const unsigned short* q = Op + r * LDP + tid;
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) {
              const unsigned pk = (unsigned)q[pl * 32 * LDP] | ((unsigned)q[pl * 32 * LDP + LDP] << 16);
              sd[pl] = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, pk), ones, sd[pl], false);
            }
          }
          const int k = c * CH + tid;
          if (k < NB) hs[k] += (sd[0] + sd[1]) + sd[2];
        }
        mma_lds_packed_s3p<DTW, CH / 16, (DTW > 1 ? 1 : TE_HB3_PFD)>(dh, Op, LDP, A.pVs + (size_t)c * (CH / 16) * 3 * 64, ntd, KBG);
        lds_barrier();                            // the next chunk's planes overwrite these
        HP(8)
      }
    }
    if (!MODE) {
#pragma unroll
      for (int j = 0; j < DTW; ++j) {
        if (w + 4 * j >= NTD) continue;
        const int col = ntd[j] * 32 + li;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int i = c_row(r, lane);
          const float ge = Esrc[(size_t)min(r0 + i, T - 1) * D + col];
          A.DH[(size_t)min(r0 + i, T) * D + col] = dh[0][j][r] + s_g[i] * ge;
        }
      }
    }
    stage();
    HP(9)
  }
  HP_END
  if (!MODE) {
    const float dw = block_sum(dwd_acc, s_red);
    if (tid == 0) hs[NB] += dw;
  }
}

// -------------------------------------------------------------------------------------------------
// te_wgrad: split-K transposed GEMMs  out[m][n] = sum_r DA[r][m0+m] * Bsrc[r][n0+n]  on TxT output
// blocks (T = 128 when D % 128 == 0, else 64); K-chunk c covers packed rows [c*chunk, (c+1)*chunk).
// job -> (A column block, B source).  Waves form a 2x2 grid, each owning (T/2)x(T/2) = Q x Q 32x32
// accumulators; each 16-row stage is loaded from HBM/L2 into registers BEFORE the MFMA block of the
// previous stage and written to the other LDS buffer after it (async-stage split, one barrier/stage).
// -------------------------------------------------------------------------------------------------

// Split products (round 4): the float32-input MFMA runs at the vector rate on gfx950, 1/16 of the bf16 rate.  Every staged value is cut
// into three bf16 planes ONCE per workgroup, when it goes to LDS (wg_split2: x = x1 + x2 + x3 exactly; a thread stages rows k and k + 1
// of four columns, so v_cvt_pk_bf16_f32 / v_perm_b32 pack the pair into the dword the MFMA wants - no transposition), and a product is
// the six partial products down to 2^-16 on v_mfma_f32_32x32x16_bf16 (exact products, float32 accumulation; dropped terms <= 2^-25 |x y|,
// random signs - the rule of the recurrent kernels): 12 MFMAs of 8 passes per 32 k-rows and output block against 16 of 16 passes.
// Measured at the Gowalla launch: 385 us (float32 MFMAs) -> 285 us (planes cut by every wave that reads a value) -> see DESIGN.md.
#ifndef WG_OCC
#define WG_OCC 2
#endif
template <int D, int T, bool F16 = false>       // F16: the POI table (gather source of the d ui jobs) holds IEEE half
__global__ __launch_bounds__(TE_BLOCK, WG_OCC) void te_wgrad_kernel(TeArgs A, int nkc) {
  constexpr int LDT = T + 8, Q = T / 64;         // Q x Q accumulators per wave; (4 LDT) % 64 == 32
  const int XW = A.xw;
  constexpr int ITEMS = 8 * (T / 4);             // (row pair, 4 columns) items of a 16-row stage and operand: one per thread (T = 64: threads 0 .. 127)
  static_assert(ITEMS <= TE_BLOCK && TE_BLOCK % ITEMS == 0, "te_wgrad: one staging item per thread");
  // a stage = 16 k-rows of both operands as three bf16 planes, rows k and k + 1 of a column packed in one dword: [buffer][plane][row pair][column]
  // (a thread's four dwords of a plane are one 16-byte write; a lane's MFMA fragment - k = 8 h .. 8 h + 7 of its column - four 4-byte reads,
  // conflict-free.  Measured against it: the pairs of a row quad side by side, [row quad][column][pair], for two 8-byte reads - with 8-byte
  // global loads, 4 rows x 2 columns per thread, 256 -> 281 us; with 16-byte loads and four 4-byte writes per plane, 8-way bank conflicts: 348 us)
  __shared__ __align__(16) unsigned At[2][3][8][LDT];
  __shared__ __align__(16) unsigned Bt[2][3][8][LDT];
  constexpr int NB_ZR = (2 * D / T) * (D / T), NB_C = (D / T) * (D / T);
  const int XWJ = A.bintab ? D : XW;                         // d ui columns that are GEMM jobs (bintab: POI half only)
  const int NB_UI = (3 * D / T) * (XWJ / T);
  // (an XCD-aware (chunk, job) order - all jobs of a K-chunk on one XCD - measured 10 % slower than this plain
  // order: it needs a chunk count that is a multiple of 8, which leaves CU slots empty)
  // grid: A.wg_slots workgroups = a 1-D list of (job, K-chunk) pairs.  Under the per-POI regrouping the d ui jobs contract over the S
  // rows (~T/5 of them), so they get proportionally fewer K-chunks - the split is computed HERE from the launch's own counts
  // (te_wgrad_split), identically in every workgroup and in dense_apply, which reads it back from kc_dev
  const int Tsteps = A.soff[A.n_seq];
  const int NB_TOT = NB_UI + NB_ZR + NB_C + (A.spatial ? ((te_nbp_dev(A.n_dist) + T - 1) / T) * (D / T) : 0);
  int n_o = nkc, n_u = nkc;
  if (A.ppoi) te_wgrad_split(A.wg_slots, NB_UI, NB_TOT, A.cnt[4], Tsteps, &n_o, &n_u, !(A.dbg & 2));      // (POI_TE_DBG bit 2: plain split, for A/B runs)
  if (A.kc_dev && blockIdx.x == 0 && threadIdx.x == 0) { A.kc_dev[0] = n_o; A.kc_dev[1] = n_u; }
  int job, kc;
  {
    const int b = blockIdx.x, nu = NB_UI * n_u;
    if (b < nu) { job = b / n_u; kc = b % n_u; }
    else { job = NB_UI + (b - nu) / n_o; kc = (b - nu) % n_o; }
    // (round 4, split products: the d wh / d vs jobs of one K-chunk placed on one XCD - block ids 8 apart - so that the four readers of the same
    // h rows share that XCD's L2: 260 us either way.  The kernel waits on its stage chain, not on HBM bandwidth.)
    if (job >= NB_TOT) return;                       // (the grid is sized for the worst case)
  }
  const bool pp = A.ppoi && job < NB_UI;
  nkc = pp ? n_u : n_o;
  int m0, n0, ldo, bsel; size_t oo;
  if (job < NB_UI) { const int bn = XWJ / T; m0 = (job / bn) * T; n0 = (job % bn) * T; ldo = XW; oo = A.dl.ui; bsel = 0; }
  else if (job < NB_UI + NB_ZR) { const int j = job - NB_UI, bn = D / T; m0 = (j / bn) * T; n0 = (j % bn) * T; ldo = D; oo = A.dl.wh; bsel = 1; }
  else if (job < NB_UI + NB_ZR + NB_C) { const int j = job - NB_UI - NB_ZR, bn = D / T; m0 = 2 * D + (j / bn) * T; n0 = (j % bn) * T; ldo = D; oo = (size_t)A.dl.wh + (size_t)2 * D * D; bsel = 2; }
  else { const int j = job - NB_UI - NB_ZR - NB_C, bn = D / T; m0 = (j / bn) * T; n0 = (j % bn) * T; ldo = D; oo = A.dl.vs; bsel = 3; }   // d vs = DL^T . H
  const int NBP = te_nbp_dev(A.n_dist), NB = A.n_dist + 1;
  const int Trows = pp ? A.cnt[4] : Tsteps, rmax = max(Trows - 1, 0);        // (loads are clamped to row rmax)
  const int chunk = (((Trows + nkc - 1) / nkc) + 63) & ~63;
  const int rb = kc * chunk, re = min(Trows, rb + chunk);
  const int lane = lane_id(), w = wave_id(), li = lane & 31, h = lane >> 5, tid = threadIdx.x;
  const int wm = (w >> 1) * (T / 2), wn = (w & 1) * (T / 2);
  f32x16 acc[Q][Q];
#pragma unroll
  for (int i = 0; i < Q; ++i)
#pragma unroll
    for (int j = 0; j < Q; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  // two register sets: the stage loaded during iteration i is written to LDS at the end of iteration
  // i+1 and consumed in iteration i+2, so a global load has two MFMA blocks to land.  Loads are
  // unconditional (clamped addresses) and masked when they are written to LDS: no branch, no wait.
  const float* Ap = pp ? A.S + m0 : bsel != 3 ? A.G + m0 : A.DL + m0;
  const int lda = bsel != 3 ? 3 * D : NBP;
  const float* Bp = bsel == 2 ? A.RH + n0 : A.H + n0;
  const int ldb = D;
  const int acols = bsel != 3 ? T : max(0, min(T, NBP - m0));     // valid columns of the A block
  const int bshift = bsel == 1 ? 1 : 0;
  // d ui jobs (bsel 0): the B operand is the step input x = [lt[p_t] | di[dp_t]], gathered from the tables
  // through row_p / row_dp (no packed copy of X exists).  The row indices of a stage are loaded one gload
  // call earlier (`ni`), so the gather is not a dependent load inside the pipeline.
  const bool gdi = n0 >= D;
  const int goff = n0 - (gdi ? D : 0);
  const float* gtab = (gdi ? A.di : A.lt) + goff;        // (F16 and !gdi: A.lt is re-read as half below, offsets in elements)
  const int* gidx = bsel == 0 ? (gdi ? A.row_dp : pp ? A.urow_p : A.row_p) : A.row_t;
  // this thread's staging item: rows 2 rp, 2 rp + 1 of the stage, columns c .. c + 3 (a wave covers two row pairs: 512 contiguous bytes per row)
  const int item = tid % ITEMS, rp = item / (T / 4), c = (item % (T / 4)) * 4;
  const int ca = c < acols ? c : 0;
  float4 ra0[2], rb0[2], ra1[2], rb1[2];
  uint2 rh0[2], rh1[2];            // F16: the raw half rows of the d ui jobs (unused otherwise)
  unsigned rt0[2], rt1[2];         // the rows' clamped indices (d wh jobs: position t of the step - its h_{t-1} operand does not exist at t = 0)
  unsigned ni[2];      // unsigned: a signed index is sign-extended right behind its load, i.e. the wave waits for it there
  // (clamped to the table: a launch WITHOUT steps - every sequence a single position - reads entry 0 of an index array nobody wrote;
  // whatever an earlier launch left there must still be a valid row.  Found by tools/fuzz_engines.py: an aperture violation)
  const unsigned nimax = (unsigned)(gdi ? A.n_dist : A.n_item);
#pragma unroll
  for (int s = 0; s < 2; ++s) ni[s] = (unsigned)gidx[min(rb + 2 * rp + s, rmax)];
  auto gload = [&](int r0, float4 (&ra)[2], float4 (&rbv)[2], unsigned (&rt)[2], uint2 (&rh)[2]) {
    // The indices of the NEXT stage are requested first, the rows of this stage after them: vmcnt retires in order, so the wait for the
    // indices at the top of the next call then leaves this call's row loads in flight.  (ni holds the RAW loaded index and is clamped
    // here, where it is consumed: clamped where it is loaded, the min sits right behind the load and the wave waits a memory latency
    // for it before its MFMA block)
#pragma unroll
    for (int s = 0; s < 2; ++s) rt[s] = min(ni[s], nimax);
#pragma unroll
    for (int s = 0; s < 2; ++s) ni[s] = (unsigned)gidx[min(r0 + 16 + 2 * rp + s, rmax)];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const int gr = min(r0 + 2 * rp + s, rmax);
      ra[s] = *reinterpret_cast<const float4*>(Ap + (size_t)gr * lda + ca);
      if constexpr (F16) {
        // branch-free (a branch around a load drains the queue): both typed loads are always issued from valid addresses - the
        // half one from row 0 of the table when this job does not gather it, the float one from H when it does - and selected
        const bool hb = bsel == 0 && !gdi;
        const float* bptr = (bsel == 0 && gdi) ? gtab + (size_t)rt[s] * D + c : Bp + (size_t)max(gr - bshift, 0) * ldb + c;
        // (both stay RAW in registers until lstore: converted and selected here, the consumer sits right behind the loads and the wave
        // waits a memory latency in front of its MFMA block - tools/scan_waits.py)
        rbv[s] = *reinterpret_cast<const float4*>(bptr);
        rh[s] = *reinterpret_cast<const uint2*>(reinterpret_cast<const __half*>(A.lt) + (hb ? goff + (size_t)rt[s] * D + c : (size_t)c));
      } else {
        const float* bptr = bsel == 0 ? gtab + (size_t)rt[s] * D + c : Bp + (size_t)max(gr - bshift, 0) * ldb + c;
        rbv[s] = *reinterpret_cast<const float4*>(bptr);
      }
    }
  };
  // masks, the three-plane split (once per element and workgroup) and the LDS writes
  auto lstore = [&](int buf, int r0, const float4 (&ra)[2], const float4 (&rbv)[2], const unsigned (&rt)[2], const uint2 (&rh)[2]) {
    float xa[2][4], xb[2][4];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const bool in = r0 + 2 * rp + s < re;
      // (component-wise selects: a select between float4 aggregates sends the arrays to scratch)
      const bool oa = in && c < acols, ob = in && (bsel != 1 || rt[s] >= 1u);      // (bsel 1: gidx is row_t)
      xa[s][0] = oa ? ra[s].x : 0.f; xa[s][1] = oa ? ra[s].y : 0.f; xa[s][2] = oa ? ra[s].z : 0.f; xa[s][3] = oa ? ra[s].w : 0.f;
      float4 bq = rbv[s];
      if constexpr (F16) {
        const bool hb = bsel == 0 && !gdi;
        const float2 fa = __half22float2(*reinterpret_cast<const __half2*>(&rh[s].x)), fb = __half22float2(*reinterpret_cast<const __half2*>(&rh[s].y));
        bq = make_float4(hb ? fa.x : bq.x, hb ? fa.y : bq.y, hb ? fb.x : bq.z, hb ? fb.y : bq.w);
      }
      xb[s][0] = ob ? bq.x : 0.f; xb[s][1] = ob ? bq.y : 0.f; xb[s][2] = ob ? bq.z : 0.f; xb[s][3] = ob ? bq.w : 0.f;
    }
    unsigned pa[3][4], pb[3][4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      wg_split2(xa[0][e], xa[1][e], pa[0][e], pa[1][e], pa[2][e]);
      wg_split2(xb[0][e], xb[1][e], pb[0][e], pb[1][e], pb[2][e]);
    }
    if (ITEMS == TE_BLOCK || tid < ITEMS) {
#pragma unroll
      for (int p = 0; p < 3; ++p) {
        *reinterpret_cast<uint4*>(&At[buf][p][rp][c]) = make_uint4(pa[p][0], pa[p][1], pa[p][2], pa[p][3]);
        *reinterpret_cast<uint4*>(&Bt[buf][p][rp][c]) = make_uint4(pb[p][0], pb[p][1], pb[p][2], pb[p][3]);
      }
    }
  };
  // one 16-row stage: a lane's fragment of a plane = the four row pairs 4 h .. 4 h + 3 of its column (k = 8 h .. 8 h + 7); six partial
  // products per output block, small ones first; the Q x Q blocks are independent chains
  auto mma = [&](int buf) {
    uint4 pa[Q][3], pb[Q][3];
#pragma unroll
    for (int p = 0; p < 3; ++p) {
#pragma unroll
      for (int i = 0; i < Q; ++i) {
        const unsigned* q = &At[buf][p][4 * h][wm + 32 * i + li];
        pa[i][p] = make_uint4(q[0], q[LDT], q[2 * LDT], q[3 * LDT]);
      }
#pragma unroll
      for (int j = 0; j < Q; ++j) {
        const unsigned* q = &Bt[buf][p][4 * h][wn + 32 * j + li];
        pb[j][p] = make_uint4(q[0], q[LDT], q[2 * LDT], q[3 * LDT]);
      }
    }
#pragma unroll
    for (int t = 0; t < 6; ++t) {
      constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
      for (int i = 0; i < Q; ++i)
#pragma unroll
        for (int j = 0; j < Q; ++j) acc[i][j] = mfma32b(pa[i][PA[t]], pb[j][PB[t]], acc[i][j]);
    }
  };
  // Branch-free pipeline (every load is clamped, every LDS write masked by `r < re`): with no control flow
  // around the loads the s_waitcnt vmcnt(n) before each LDS write is exact, i.e. it leaves the stage that
  // was fetched just before the MFMA block in flight.  A chunk is a multiple of 64 rows, so the odd stage
  // of the last iteration is the only work that can be empty.
  gload(rb, ra0, rb0, rt0, rh0); lstore(0, rb, ra0, rb0, rt0, rh0);
  gload(rb + 16, ra0, rb0, rt0, rh0);
  __syncthreads();
  for (int r0 = rb; r0 < re; r0 += 32) {
    // even stage: LDS buffer 0; set 0 holds stage +1, set 1 receives stage +2
    gload(r0 + 32, ra1, rb1, rt1, rh1);
    mma(0);
    lstore(1, r0 + 16, ra0, rb0, rt0, rh0);
    __syncthreads();
    // odd stage: LDS buffer 1; set 1 holds stage +1, set 0 receives stage +2
    gload(r0 + 48, ra0, rb0, rt0, rh0);
    mma(1);
    lstore(0, r0 + 32, ra1, rb1, rt1, rh1);
    __syncthreads();
  }
  float* out = A.slab + (size_t)kc * A.dl.total + oo;
  const int mbase = (bsel == 2 ? m0 - 2 * D : m0) + wm;
#pragma unroll
  for (int i = 0; i < Q; ++i)
#pragma unroll
    for (int j = 0; j < Q; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = mbase + 32 * i + c_row(r, lane);
        // plain stores: the slabs are zero on entry (dense_apply re-zeroes what it reads) and every element has
        // one writer per step - a read-modify-write here is 64 dependent round trips per lane
        if (bsel != 3 || m < NB) out[(size_t)m * ldo + n0 + wn + 32 * j + li] = acc[i][j][r];
      }
}

// per-sequence losses (deterministic row order) + loss-weight statistics
__global__ __launch_bounds__(TE_BLOCK) void te_finalize_kernel(TeArgs A) {
  __shared__ float red[8];
  const int k = blockIdx.x * TE_BLOCK + threadIdx.x;
  float sur = 0.f, bpr = 0.f;
  if (!A.spatial) {
    // plain GRU (public/GRU.py:352-357,380): loss = -sum_t log sigmoid(u_t); the t = 0 term has
    // h_0 = 0, i.e. u = 0 and no gradient - only its constant log sigmoid(0) = -ln 2
    if (k < A.n_seq) {
      for (int r = A.soff[k]; r < A.soff[k + 1]; ++r) bpr += A.rowloss[2 * (size_t)r + 1];
      const int u = A.uidx[k];
      A.out[k] = -(bpr + (A.off[u + 1] > A.off[u] ? -0.69314718056f : 0.f));
    }
    return;
  }
  float ls0, ls1;
  {
    const float a = A.lw[0], b = A.lw[1], m = fmaxf(a, b);
    const float ea = expf(a - m), eb = expf(b - m);
    ls0 = ea / (ea + eb); ls1 = eb / (ea + eb);
  }
  if (k < A.n_seq) {
    for (int r = A.soff[k]; r < A.soff[k + 1]; ++r) { sur += A.rowloss[2 * (size_t)r]; bpr += A.rowloss[2 * (size_t)r + 1]; }
    float* o = A.out + (size_t)k * 5;
    o[0] = ls0 * sur - ls1 * bpr; o[1] = sur; o[2] = -bpr; o[3] = ls0; o[4] = ls1;
  }
  const float s1 = block_sum(sur, red);
  const float s2 = block_sum(-bpr, red);
  if (threadIdx.x == 0) { A.fin_part[2 * blockIdx.x] = s1; A.fin_part[2 * blockIdx.x + 1] = s2; }   // summed in order by te_parts_kernel
}

// Per-workgroup / per-tile partial sums -> slab 0 in a fixed order (lane-strided sums + DPP tree), one
// wavefront per output element: d bi (3D elements, one partial row per recurrent tile), and for the
// Distance2Pre model d bs | d wd (te_head workgroups; re-zeroed) and the two loss sums (te_finalize blocks).
__global__ __launch_bounds__(TE_BLOCK) void te_parts_kernel(TeArgs A, int n_tile, int n_fin) {
  const int j = blockIdx.x * POI_NWAVE + wave_id(), D3 = 3 * A.dim, NB = A.n_dist + 1;
  const int lane = lane_id();
  float s = 0.f;
  if (A.hyb) n_tile = A.hyb_dev[2] + (A.n_seq - A.hyb_dev[2] + 15) / 16;      // hybrid recurrences: one row per leading sequence + one per tile
  if (j < D3) {
    for (int k = lane; k < n_tile; k += 64) s += A.bi_part[(size_t)k * D3 + j];
    s = wave_sum(s);
    if (lane == 0) A.slab[A.dl.bi + j] += s;
  } else if (A.spatial && j < D3 + NB + 1) {
    const int e = j - D3;
    for (int k = lane; k < A.n_head; k += 64) { float* p = A.hslab + (size_t)k * A.hstride + e; s += *p; *p = 0.f; }
    s = wave_sum(s);
    if (lane == 0) A.slab[A.dl.bs + e] += s;
  } else if (A.spatial && j < D3 + NB + 3) {
    const int e = j - (D3 + NB + 1);
    for (int k = lane; k < n_fin; k += 64) s += A.fin_part[2 * k + e];
    s = wave_sum(s);
    if (lane == 0) A.slab[(e ? A.dl.upq : A.dl.sur)] += s;
  }
}

// -------------------------------------------------------------------------------------------------
// One-sequence path (TeArgs n_seq == 1: the reference schedule, prog_bpr_gru_spatial.py:249-250 - one user per step).  The batched
// pipeline spends a launch of one sequence in ~40 dependent dispatches of 4 - 30 us each (sort, regrouping, split-K slabs, write-back
// chains - machinery that pays at thousands of sequences); here the step is FIVE kernels:
//   te_one_in   prep + gather + G = X . ui^T + bi on the vector ALUs (24 workgroups of 16 gate rows), E rows, weight packs for te_head /
//               te_rec_bwd1, and two snapshots the last kernel needs: the step inputs X (T x 2D) and ui (before its update)
//   te_rec_fwd1, te_head, te_rec_bwd1                      (as in the batched path)
//   te_one_out  everything after BPTT in one launch: d ui / d wh / d vs as K = T products with the SGD step applied in the epilogue
//               (no slabs), bi / bs / wd / loss_weight / losses by one workgroup, and the sparse write-back with one workgroup per
//               table touch (3 L slots: first occurrence of a row sums the row's touches in slot order - dx on the fly from DA and the
//               ui snapshot, +- g h - and applies the step; the padding rows' analytic multiplicities as in te_rowmap).
// Same formulas and batch rule (n_seq = 1) as te_scatter / dense_apply; T <= 64 steps.
// -------------------------------------------------------------------------------------------------
#define ONE_TMAX 160      // steps of the one-sequence path (te_one_out keeps a 16 x ONE_TMAX and a 64 x ONE_TMAX operand tile in LDS: 54 KB); 161 positions cover the len_max 157 the reference mentions for Foursquare (public/GRU.py:171)
__device__ __forceinline__ void one_header(const TeArgs& A, int& base, int& L, int& ns) {
  const int u = A.uidx[0];
  base = A.off[u]; L = min(A.off[u + 1] - base, ONE_TMAX + 1); ns = L > 0 ? L - 1 : 0;      // (the host checked max_len <= ONE_TMAX + 1: the clamp only guards the LDS tiles against inconsistent tables)
}

template <int D>
__global__ __launch_bounds__(TE_BLOCK) void te_one_in_kernel(TeArgs A, PackJobs J, int n_ax, int n_pk) {
  constexpr int LDX = 2 * D + 4;
  __shared__ __align__(16) float Xc[16][LDX], U[16][LDX];
  const int tid = threadIdx.x, b = blockIdx.x, XW = A.xw;        // 2 D (Distance2Pre: POI | distance bin) or D (plain GRU)
  int base, L, ns; one_header(A, base, L, ns);
  if (b >= n_ax + 1) {                       // weight packs (te_head's vs fragments, the transposes of te_rec_bwd1)
    const int v = b - n_ax - 1;
    if (v / n_pk < J.n) te_pack_block(J.j[v / n_pk], v % n_pk, n_pk);
    return;
  }
  if (b == n_ax) {                           // prep (te_len / te_scan / te_rowmap of one sequence) + E rows
    if (tid == 0) { A.soff[0] = 0; A.soff[1] = ns; }
    for (int t = tid; t < ns; t += TE_BLOCK) {
      A.row_src[t] = base + t; A.row_t[t] = t; A.row_p[t] = A.p[base + t];
      if (A.spatial) { A.row_dp[t] = A.dp[base + t]; A.row_ab[t] = A.dp[base + t + 1] | (A.dq[base + t + 1] << 16); }
    }
    constexpr int LPR = D / 4;
    for (int e = tid; e < ns * LPR; e += TE_BLOCK) {
      const int t = e / LPR, c = (e % LPR) * 4;
      const float4 a = *reinterpret_cast<const float4*>(A.lt + (size_t)A.p[base + t + 1] * D + c);
      const float4 q = *reinterpret_cast<const float4*>(A.lt + (size_t)A.q[base + t + 1] * D + c);
      *reinterpret_cast<float4*>(A.E + (size_t)t * D + c) = make_float4(a.x - q.x, a.y - q.y, a.z - q.z, a.w - q.w);
    }
    return;
  }
  // gate rows m0 .. m0 + 15 of G for every step: thread (tm, tt) -> G[t0 + tt][m0 + tm]
  const int m0 = 16 * b, tm = tid & 15, tt = tid >> 4;
  for (int e = tid; e < 16 * (XW / 4); e += TE_BLOCK) {
    const int r = e / (XW / 4), c = (e % (XW / 4)) * 4;
    const float4 v = *reinterpret_cast<const float4*>(A.ui + (size_t)(m0 + r) * XW + c);
    *reinterpret_cast<float4*>(&U[r][c]) = v;
    *reinterpret_cast<float4*>(A.uiT + (size_t)(m0 + r) * XW + c) = v;          // ui snapshot, plain layout (read by te_one_out after ui moved)
  }
  const float bias = A.bi[m0 + tm];
  for (int t0 = 0; t0 < ns; t0 += 16) {
    __syncthreads();
    for (int e = tid; e < 16 * (XW / 4); e += TE_BLOCK) {
      const int r = e / (XW / 4), c = (e % (XW / 4)) * 4, t = min(t0 + r, ns - 1);
      const float* src = c < D ? A.lt + (size_t)A.p[base + t] * D + c : A.di + (size_t)A.dp[base + t] * D + (c - D);
      const float4 v = *reinterpret_cast<const float4*>(src);
      *reinterpret_cast<float4*>(&Xc[r][c]) = v;
      if (b == 0 && t0 + r < ns) *reinterpret_cast<float4*>(A.X + (size_t)(t0 + r) * XW + c) = v;       // X snapshot
    }
    __syncthreads();
    if (A.xfwd) {      // exact forward: the input product in float64 (products of float32 values are exact), gate-major row of gx
      double a0 = 0.0, a1 = 0.0;
#pragma unroll 8
      for (int c = 0; c < XW; c += 4) {
        const float4 x = *reinterpret_cast<const float4*>(&Xc[tt][c]), u = *reinterpret_cast<const float4*>(&U[tm][c]);
        a0 = __builtin_fma((double)x.x, (double)u.x, a0); a1 = __builtin_fma((double)x.y, (double)u.y, a1);
        a0 = __builtin_fma((double)x.z, (double)u.z, a0); a1 = __builtin_fma((double)x.w, (double)u.w, a1);
      }
      if (t0 + tt < ns) A.gx[(size_t)(t0 + tt) * 3 * D + m0 + tm] = (a0 + a1) + (double)bias;
      continue;
    }
    float a0 = 0.f, a1 = 0.f;
#pragma unroll 8
    for (int c = 0; c < XW; c += 4) {
      const float4 x = *reinterpret_cast<const float4*>(&Xc[tt][c]), u = *reinterpret_cast<const float4*>(&U[tm][c]);
      a0 = __fmaf_rn(x.x, u.x, a0); a1 = __fmaf_rn(x.y, u.y, a1); a0 = __fmaf_rn(x.z, u.z, a0); a1 = __fmaf_rn(x.w, u.w, a1);
    }
    if (t0 + tt < ns) A.G[(size_t)(t0 + tt) * 3 * D + m0 + tm] = (a0 + a1) + bias;
  }
}

// out[m][n] = sum_t Aop[t][m] * Bop[t][n] on a 16 x 64 block, then theta[m][n] -= aeff * (out + lambda * theta[m][n])
struct OneJob { const float* a; int lda, m0, mvalid; const float* b; int ldb, n0, bshift; float* theta; int ldt; };
__device__ __forceinline__ void one_dense_block(const OneJob& j, int ns, float aeff, float lambda, float (*As)[17], float (*Bs)[68]) {
  const int tid = threadIdx.x, tm = tid & 15, tq = tid >> 4;
  for (int e = tid; e < ns * 16; e += TE_BLOCK) {
    const int t = e >> 4, m = e & 15;
    As[t][m] = (j.m0 + m < j.mvalid) ? j.a[(size_t)t * j.lda + j.m0 + m] : 0.f;
  }
  for (int e = tid; e < ns * 16; e += TE_BLOCK) {
    const int t = e >> 4, c = (e & 15) * 4;
    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
    *reinterpret_cast<float4*>(&Bs[t][c]) = (t >= j.bshift) ? *reinterpret_cast<const float4*>(j.b + (size_t)(t - j.bshift) * j.ldb + j.n0 + c) : z;
  }
  __syncthreads();
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int t = 0; t < ns; ++t) {
    const float a = As[t][tm];
    const float4 v = *reinterpret_cast<const float4*>(&Bs[t][4 * tq]);
    acc.x = __fmaf_rn(a, v.x, acc.x); acc.y = __fmaf_rn(a, v.y, acc.y); acc.z = __fmaf_rn(a, v.z, acc.z); acc.w = __fmaf_rn(a, v.w, acc.w);
  }
  if (j.m0 + tm < j.mvalid) {
    float4* th = reinterpret_cast<float4*>(j.theta + (size_t)(j.m0 + tm) * j.ldt + j.n0 + 4 * tq);
    float4 w = *th;
    w.x -= aeff * (acc.x + lambda * w.x); w.y -= aeff * (acc.y + lambda * w.y);
    w.z -= aeff * (acc.z + lambda * w.z); w.w -= aeff * (acc.w + lambda * w.w);
    *th = w;
  }
}

template <int D>
__global__ __launch_bounds__(TE_BLOCK) void te_one_out_kernel(TeArgs A, float alpha, float lambda, int n_hwg, int l_cap) {
  const int XW = A.xw;                               // 2 D (Distance2Pre) or D (plain GRU: no distance-bin rows, no vs / bs / wd / loss weights)
  __shared__ __align__(16) float As[ONE_TMAX][17];
  __shared__ __align__(16) float Bs[ONE_TMAX][68];
  __shared__ int s_key[3 * (ONE_TMAX + 1)], s_hit[3 * (ONE_TMAX + 1)], s_cnt;
  __shared__ float s_red[8];
  const int tid = threadIdx.x;
  int base, L, ns; one_header(A, base, L, ns);
  const int NB = A.n_dist + 1, NBP = te_nbp_dev(A.n_dist);
  const float aeff = alpha * (A.bcap < 0.f ? 1.0f : fminf(1.0f, A.bcap));       // dense rule of dense_apply_kernel at n_seq = 1
  // ---- dense gradients + SGD step: 16 x 64 output blocks ----
  const int nb_ui = (3 * D / 16) * (XW / 64), nb_zr = (2 * D / 16) * (D / 64), nb_c = (D / 16) * (D / 64), nb_vs = A.spatial ? ((NB + 15) / 16) * (D / 64) : 0;
  int b = blockIdx.x;
  if (b < nb_ui + nb_zr + nb_c + nb_vs) {
    OneJob j;
    if (b < nb_ui) {                     // d ui = DA^T . X (snapshot)
      j = OneJob{A.G, 3 * D, 16 * (b / (XW / 64)), 3 * D, A.X, XW, 64 * (b % (XW / 64)), 0, A.ui, XW};
    } else if ((b -= nb_ui) < nb_zr) {   // d wh[z | r] = [da_z | da_r]^T . h_{t-1}
      j = OneJob{A.G, 3 * D, 16 * (b / (D / 64)), 2 * D, A.H, D, 64 * (b % (D / 64)), 1, A.wh, D};
    } else if ((b -= nb_zr) < nb_c) {    // d wh[c] = da_c^T . (r * h_{t-1})
      j = OneJob{A.G + 2 * D, 3 * D, 16 * (b / (D / 64)), D, A.RH, D, 64 * (b % (D / 64)), 0, A.wh + (size_t)2 * D * D, D};
    } else {                             // d vs = DL^T . H
      b -= nb_c;
      j = OneJob{A.DL, NBP, 16 * (b / (D / 64)), NB, A.H, D, 64 * (b % (D / 64)), 0, A.vs, D};
    }
    one_dense_block(j, ns, aeff, lambda, As, Bs);
    return;
  }
  b -= nb_ui + nb_zr + nb_c + nb_vs;
  if (b == 0) {
    // ---- bi | bs | wd | losses | loss_weight (te_finalize + te_parts + dense_apply of one sequence) ----
    for (int e = tid; e < 3 * D; e += TE_BLOCK) { const float w = A.bi[e]; A.bi[e] = w - aeff * (A.bi_part[e] + lambda * w); }
    for (int e = tid; A.spatial && e <= NB; e += TE_BLOCK) {
      float g = 0.f;
      for (int k = 0; k < n_hwg; ++k) { float* p = A.hslab + (size_t)k * A.hstride + e; g += *p; *p = 0.f; }
      float* th = e < NB ? A.bs + e : A.wd;
      const float w = *th; *th = w - aeff * (g + lambda * w);
    }
    float sur = 0.f, bpr = 0.f;
    for (int r = tid; r < ns; r += TE_BLOCK) { sur += A.rowloss[2 * (size_t)r]; bpr += A.rowloss[2 * (size_t)r + 1]; }
    sur = block_sum(sur, s_red); bpr = block_sum(bpr, s_red);
    if (tid == 0 && !A.spatial) A.out[0] = -(bpr + (L > 0 ? -0.69314718056f : 0.f));      // te_finalize, plain GRU (public/GRU.py:352-357,380)
    if (tid == 0 && A.spatial) {
      const float a = A.lw[0], c = A.lw[1], m = fmaxf(a, c);
      const float ea = expf(a - m), eb = expf(c - m);
      const float ls0 = ea / (ea + eb), ls1 = eb / (ea + eb);
      float* o = A.out;
      o[0] = ls0 * sur - ls1 * bpr; o[1] = sur; o[2] = -bpr; o[3] = ls0; o[4] = ls1;
      const float d0 = sur + lambda * ls0, d1 = -bpr + lambda * ls1, dot = d0 * ls0 + d1 * ls1;
      A.lw[0] = a - aeff * ls0 * (d0 - dot);
      A.lw[1] = c - aeff * ls1 * (d1 - dot);
    }
    return;
  }
  // ---- sparse write-back: one workgroup per table touch; slot e = section * L + j as te_slots ----
  const int nsec = A.spatial ? 3 : 2;                  // the plain GRU has no distance-bin section
  const int e = b - 1, n3 = nsec * L, pad_lt = A.n_item, pad_di = A.n_item + 1 + A.n_dist;
  auto key_of = [&](int i) { const int sec = i / L, jj = i - sec * L; return sec == 0 ? A.p[base + jj] : sec == 1 ? A.q[base + jj] : A.n_item + 1 + A.dp[base + jj]; };
  int key;
  const bool padwg = e >= 3 * l_cap;                  // the two padding rows without a literal touch: analytic multiplicity only
  if (padwg) { if (e > 3 * l_cap && !A.spatial) return; key = e == 3 * l_cap ? pad_lt : pad_di; }
  else { if (e >= n3) return; key = key_of(e); }
  if (tid == 0) s_cnt = 0;
  for (int i = tid; i < n3; i += TE_BLOCK) s_key[i] = key_of(i);
  __syncthreads();
  int dup = 0;
  for (int i = tid; i < n3; i += TE_BLOCK) { const bool hit = s_key[i] == key; s_hit[i] = hit; dup |= hit && (padwg || i < e); }
  if (__syncthreads_or(dup)) return;                  // an earlier slot owns the row (or a literal touch owns the padding row)
  int cnt = 0;
  for (int i = 0; i < n3; ++i) cnt += s_hit[i];       // (<= 195 LDS reads)
  const int am = key == pad_lt ? 2 * (A.len_max - L) : (A.spatial && key == pad_di) ? (A.len_max - L) : 0;
  if (cnt + am == 0) return;
  // thread (cq, part): columns 4 cq .. 4 cq + 3, every NP-th gate row of the dx products - 256 threads keep 3 D / NP independent
  // 16-byte loads of the ui snapshot in flight each (one thread per column walked the 3 D rows behind a handful of loads: 40 us)
  constexpr int CQ = D / 4, NP = TE_BLOCK / CQ;
  const int cq = tid % CQ, part = tid / CQ;
  const int doff = key <= A.n_item ? 0 : D;
  // The row's dx sum is linear in DA: sum the DA rows of its step-input touches first (a distance-bin row can have as many as the
  // sequence has steps - 80 % of the transitions of a user fall into a few bins), then ONE pass over the ui snapshot
  float* Ssum = &Bs[0][0];                            // 3 D floats
  float4 a4 = make_float4(0.f, 0.f, 0.f, 0.f);
  {
    float s0 = 0.f, s1 = 0.f;                         // m = tid, tid + 256 (3 D <= 384)
    bool any = false;
    for (int i = padwg ? n3 : e; i < n3; ++i) {
      if (!s_hit[i]) continue;
      const int sec = i / L, jj = i - sec * L;
      if (sec != 1 && jj < ns) {
        const float* da = A.G + (size_t)jj * 3 * D;
        s0 += da[min(tid, 3 * D - 1)]; s1 += da[min(tid + TE_BLOCK, 3 * D - 1)];
        any = true;
      }
      if (sec != 2 && jj >= 1 && part == 0) {         // +- g h of step jj - 1
        const float g0 = A.gcoef[jj - 1], g = sec == 1 ? -g0 : g0;
        const float4 h = *reinterpret_cast<const float4*>(A.H + (size_t)(jj - 1) * D + 4 * cq);
        a4.x = __fmaf_rn(g, h.x, a4.x); a4.y = __fmaf_rn(g, h.y, a4.y); a4.z = __fmaf_rn(g, h.z, a4.z); a4.w = __fmaf_rn(g, h.w, a4.w);
      }
    }
    if (tid < 3 * D) Ssum[tid] = s0;
    if (tid + TE_BLOCK < 3 * D) Ssum[tid + TE_BLOCK] = s1;
    __syncthreads();
    if (any) {                                        // + S . ui_old[:, doff + c]     (uniform: every thread saw the same slots)
      const float* u = A.uiT + doff + 4 * cq;
      float4 d0 = make_float4(0.f, 0.f, 0.f, 0.f), d1 = d0;
#pragma unroll 8
      for (int m = part; m < 3 * D; m += 2 * NP) {
        const float w0 = Ssum[m], w1 = Ssum[m + NP];
        const float4 v0 = *reinterpret_cast<const float4*>(u + (size_t)m * XW), v1 = *reinterpret_cast<const float4*>(u + (size_t)(m + NP) * XW);
        d0.x = __fmaf_rn(w0, v0.x, d0.x); d0.y = __fmaf_rn(w0, v0.y, d0.y); d0.z = __fmaf_rn(w0, v0.z, d0.z); d0.w = __fmaf_rn(w0, v0.w, d0.w);
        d1.x = __fmaf_rn(w1, v1.x, d1.x); d1.y = __fmaf_rn(w1, v1.y, d1.y); d1.z = __fmaf_rn(w1, v1.z, d1.z); d1.w = __fmaf_rn(w1, v1.w, d1.w);
      }
      a4.x += d0.x + d1.x; a4.y += d0.y + d1.y; a4.z += d0.z + d1.z; a4.w += d0.w + d1.w;
    }
  }
  float* red = &As[0][0];                             // NP x D partial sums (<= 1088 floats)
  *reinterpret_cast<float4*>(red + part * D + 4 * cq) = a4;
  __syncthreads();
  const int c = tid;
  if (c >= D) return;
  float acc = 0.f;
#pragma unroll
  for (int q = 0; q < NP; ++q) acc += red[q * D + c];
  float sc, lm; rule_scales(alpha, lambda, 1, cnt + am, A.bcap, sc, lm);
  float* row = key <= A.n_item ? A.lt + (size_t)key * D : A.di + (size_t)(key - A.n_item - 1) * D;
  const float w = row[c];
  row[c] = w - sc * (acc + lm * w);
}

// -------------------------------------------------------------------------------------------------
// host side
// -------------------------------------------------------------------------------------------------
int te_wgrad_jobs(int D, int n_dist, bool spatial, bool bintab) {
  const int T = (D % 128 == 0) ? 128 : 64, XW = (spatial && !bintab) ? 2 * D : D;
  return (3 * D / T) * (XW / T) + (2 * D / T) * (D / T) + (D / T) * (D / T) + (spatial ? ((te_nbp_dev(n_dist) + T - 1) / T) * (D / T) : 0);
}

int te_wgrad_ui_jobs(int D, int n_dist, bool spatial, bool bintab) {
  (void)n_dist;
  const int T = (D % 128 == 0) ? 128 : 64, XW = (spatial && !bintab) ? 2 * D : D;
  return (3 * D / T) * (XW / T);
}

// Distance2Pre at D >= 128: the distance-bin half of the input goes through per-bin tables (te_ztab / te_dsum)
bool te_bintab(int D, bool spatial, int n_dist) { return spatial && D >= 128 && n_dist + 1 <= 2048; }      // (te_dprep: two bins per thread)

bool te_supported(int D, int n_dist) { return (D == 64 || D == 128 || D == 256) && n_dist + 1 <= 2048; }   // plain GRU: n_dist == -1

int te_nbp(int n_dist);
static int nbt_for(int nb) { return te_nbp_dev(nb - 1) / 32; }

int te_nbp(int n_dist) { return te_nbp_dev(n_dist); }

template <int D, int NBT>
static hipError_t te_launch_head(const TeArgs& A, int mode, int grid, hipStream_t st) {
  const size_t lds = sizeof(float) * (32 * (D + 4) + 32 * (NBT * 32 + 4));
  if (mode) hipLaunchKernelGGL((te_head_kernel<D, NBT, 1>), dim3(grid), dim3(TE_BLOCK), lds, st, A);
  else hipLaunchKernelGGL((te_head_kernel<D, NBT, 0>), dim3(grid), dim3(TE_BLOCK), lds, st, A);
  return hipGetLastError();
}

template <int D, int NBT>
static hipError_t te_launch_head3(const TeArgs& A, int grid, hipStream_t st) {
  const size_t lds = sizeof(float) * (32 * (D + 4) + 32 * (NBT * 32 + 8));
  if constexpr (D == 128) {
    if (A.efuse) { hipLaunchKernelGGL((te_head3_kernel<D, NBT, true>), dim3(grid), dim3(TE_BLOCK), lds, st, A); return hipGetLastError(); }
  }
  if (A.efuse) return hipErrorInvalidValue;
  hipLaunchKernelGGL((te_head3_kernel<D, NBT>), dim3(grid), dim3(TE_BLOCK), lds, st, A);
  return hipGetLastError();
}

template <int D>
static hipError_t te_head_dispatch(const TeArgs& A, int mode, int grid, hipStream_t st) {
  if (A.n_dist + 1 <= 256 && A.head_split && mode == 0) {      // training head on split products (te_head3)
    switch (nbt_for(A.n_dist + 1)) {
      case 1: return te_launch_head3<D, 1>(A, grid, st);
      case 2: return te_launch_head3<D, 2>(A, grid, st);
      case 4: return te_launch_head3<D, 4>(A, grid, st);
      case 7: return te_launch_head3<D, 7>(A, grid, st);
      default: return te_launch_head3<D, 8>(A, grid, st);
    }
  }
  if (A.n_dist + 1 > 256 && A.head_split) {      // chunked head on split products
    const size_t lds = sizeof(short) * 3 * 32 * (D + 8) + sizeof(short) * 3 * 32 * (256 + 8);
    if (mode) hipLaunchKernelGGL((te_head_big3_kernel<D, 1>), dim3(grid), dim3(TE_BLOCK), lds, st, A);
    else hipLaunchKernelGGL((te_head_big3_kernel<D, 0>), dim3(grid), dim3(TE_BLOCK), lds, st, A);
    return hipGetLastError();
  }
  if (A.n_dist + 1 > 256) {      // chunked head (te_head_big_kernel)
    const size_t lds = sizeof(float) * (32 * (D + 4) + 32 * (256 + 4) + te_nbp_dev(A.n_dist));
    if (mode) hipLaunchKernelGGL((te_head_big_kernel<D, 1>), dim3(grid), dim3(TE_BLOCK), lds, st, A);
    else hipLaunchKernelGGL((te_head_big_kernel<D, 0>), dim3(grid), dim3(TE_BLOCK), lds, st, A);
    return hipGetLastError();
  }
  switch (nbt_for(A.n_dist + 1)) {
    case 1: return te_launch_head<D, 1>(A, mode, grid, st);
    case 2: return te_launch_head<D, 2>(A, mode, grid, st);
    case 4: return te_launch_head<D, 4>(A, mode, grid, st);
    case 7: return te_launch_head<D, 7>(A, mode, grid, st);
    default: return te_launch_head<D, 8>(A, mode, grid, st);
  }
}

static void te_pack_jobs(const TeArgs& A, PackJobs& J, bool train) {
  const int D = A.dim, NBP = nbt_for(A.n_dist + 1) * 32, NB = A.n_dist + 1;
  int n = 0;
  // B[k][n] = vs[n][k]   (K = D, N = NB -> NBP)
  if (A.spatial && A.head_split) J.j[n++] = PackJob{A.vs, 1, D, D, NB, D / 16, NBP / 32, A.pVsT, 4};
  else if (A.spatial) J.j[n++] = PackJob{A.vs, 1, D, D, NB, D / 8, NBP / 32, A.pVsT};
  if (train) {
    // B[k][n] = vs[k][n]   (K = NB -> NBP, N = D)
    if (A.spatial && A.head_split) J.j[n++] = PackJob{A.vs, D, 1, NB, D, NBP / 16, D / 32, A.pVs, 4};
    else if (A.spatial) J.j[n++] = PackJob{A.vs, D, 1, NB, D, NBP / 8, D / 32, A.pVs};
    // 16-column fragments of the recurrent kernels (16x16x4 MFMA): B[k][n] = wh[2][k][n] (K = D, N = D) and
    // B[k][n] = wh_flat[k][n], k < 2D (K = 2D, N = D)
    if (A.hyb) {        // hybrid recurrences: the per-sequence backward kernel's transposes NEXT TO the tile kernel's fragments (below)
      J.j[n++] = PackJob{A.wh + (size_t)2 * D * D, D, 1, D, D, 0, 0, A.pWhc1, 3};
      J.j[n++] = PackJob{A.wh, D, 1, 2 * D, D, 0, 0, A.pWhzr1, 3};
    }
    if (A.rec1) {       // per-sequence kernels: plain transposes (the forward kernel reads wh itself)
      J.j[n++] = PackJob{A.wh + (size_t)2 * D * D, D, 1, D, D, 0, 0, A.pWhc16, 3};
      J.j[n++] = PackJob{A.wh, D, 1, 2 * D, D, 0, 0, A.pWhzr16, 3};
    } else if (A.rec32 && A.rec_split) {      // streaming kernels on split products: bf16 x 3 planes, 32x32x16 fragments
      J.j[n++] = PackJob{A.wh + (size_t)2 * D * D, D, 1, D, D, D / 16, D / 32, A.pWhc16, 4};
      J.j[n++] = PackJob{A.wh, D, 1, 2 * D, D, 2 * D / 16, D / 32, A.pWhzr16, 4};
    } else if (A.rec32) {      // 32-column fragments of the streaming recurrent kernels (same buffers)
      J.j[n++] = PackJob{A.wh + (size_t)2 * D * D, D, 1, D, D, D / 8, D / 32, A.pWhc16, 0};
      J.j[n++] = PackJob{A.wh, D, 1, 2 * D, D, 2 * D / 8, D / 32, A.pWhzr16, 0};
    } else if (A.rec_split) {      // bf16 x 3 planes, 16x16x32 fragments (same buffers, 1.5 x the bytes)
      J.j[n++] = PackJob{A.wh + (size_t)2 * D * D, D, 1, D, D, D / 32, D / 16, A.pWhc16, 2};
      J.j[n++] = PackJob{A.wh, D, 1, 2 * D, D, 2 * D / 32, D / 16, A.pWhzr16, 2};
    } else {
      J.j[n++] = PackJob{A.wh + (size_t)2 * D * D, D, 1, D, D, D / 16, D / 16, A.pWhc16, 1};
      J.j[n++] = PackJob{A.wh, D, 1, 2 * D, D, 2 * D / 16, D / 16, A.pWhzr16, 1};
    }
  }
  // B[k][n] = wh_flat[n][k]   (K = D, N = 3D), 16-column fragments (32-column ones for the streaming kernels)
  if (A.rec1 || A.xfwd) {}      // (te_rec_fwd1 reads wh directly; the exact forward has its own digit fragments: te_xpack)
  else if (A.rec32 && A.rec_split) J.j[n++] = PackJob{A.wh, 1, D, D, 3 * D, D / 16, 3 * D / 32, A.pWhT16, 4};
  else if (A.rec32) J.j[n++] = PackJob{A.wh, 1, D, D, 3 * D, D / 8, 3 * D / 32, A.pWhT16, 0};
  else if (A.rec_split && (!A.fwd_tab || A.predict)) J.j[n++] = PackJob{A.wh, 1, D, D, 3 * D, D / 32, 3 * D / 16, A.pWhT16, 2};      // (forward-table TRAINING launches keep the float32 kernel)
  else J.j[n++] = PackJob{A.wh, 1, D, D, 3 * D, D / 16, 3 * D / 16, A.pWhT16, 1};
  // forward table on split products (te_ptab_s3): B[k][n] = ui[(n % 3) D + n / 3][k], k < D - the POI half of ui, gate-interleaved columns
  if (A.fwd_tab && A.rec_split && D == 128) J.j[n++] = PackJob{A.ui, 1, A.xw, D, 3 * D, D / 16, 3 * D / 32, A.pUiP3, 5};
  J.n = n;
}

// The forward table on split products (poi_ctx_set_split_products): ptab[r][n] = sum_k lt[r][k] uiP[n][k] over the n_item + 1 table rows,
// dim 128.  A workgroup takes 128 table rows; a wave keeps ITS 32 rows as resident A fragments - float32 (or half) rows fetched straight
// into the fragment layout, split into three bf16 planes in registers (96 registers) - and walks the twelve 32-column tiles of uiP, whose
// bf16 x 3 fragments (te_pack n16 == 4: 24 KB per column tile, already in lane order) all four waves read from LDS: one pass over uiP per
// 128 rows out of L2 (a 32-row tile per pass would ask L2 for 0.9 GB per launch).  The fragments come in by LDS DMA
// (global_load_lds_dwordx4: no staging registers - the kernel sits at the register cap) into a ring of three buffers, two column tiles
// ahead of the one the MFMAs read: an L2 round trip is ~3 column tiles of matrix work.  The DMA is issued from inline asm, so the
// waits are counted here (s_waitcnt vmcnt(N): the queue retires in order, N = what was issued behind the awaited tile - the C stores
// of the tiles in between and the next tile's six requests) and the barriers are raw (no vmcnt(0) in front of them).  Six
// v_mfma_f32_32x32x16_bf16 per 16 k, the leading product in its own accumulator.  Rows past the table's last go to the spare row
// behind it, as in te_gemm_ntk.
template <int N>
__device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst) {      // lds_dst: the WAVE's destination (lane l lands at + 16 l)
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
template <int D, bool F16>
__global__ __launch_bounds__(TE_BLOCK, 2) void te_ptab_s3_kernel(const void* __restrict__ tab, const int* __restrict__ n_ptr, const float4* __restrict__ Bp,
                                                                 float* __restrict__ C) {
  constexpr int KG = D / 16, N = 3 * D, NT = N / 32, FR = KG * 3 * 64, PT = FR / TE_BLOCK;      // uint4 fragments per column tile; per thread
  static_assert(FR % TE_BLOCK == 0 && NT >= 4, "te_ptab_s3: fragment block per thread");
  extern __shared__ __align__(16) float lds[];
  uint4* s_b = reinterpret_cast<uint4*>(lds);                 // [3][FR]
  const int n_rows = *n_ptr;
  const int tid = threadIdx.x, lane = lane_id(), w = wave_id(), li = lane & 31, h = lane >> 5;
  const uint4* bsrc = reinterpret_cast<const uint4*>(Bp);
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) void*)s_b;
  const unsigned wbase = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)(w * 64) * 16u);
  auto request = [&](int j) {                                  // column tile j -> ring slot j % 3: PT requests per thread
    const unsigned dst = wbase + (unsigned)((j % 3) * FR) * 16u;
#pragma unroll
    for (int q = 0; q < PT; ++q) glds16(bsrc + (size_t)j * FR + tid + q * TE_BLOCK, dst + (unsigned)(q * TE_BLOCK) * 16u);
  };
  const int n_tile = (n_rows + 127) / 128;
  for (int t = blockIdx.x; t < n_tile; t += gridDim.x) {
    uint4 a[KG][3];
    {
      const size_t row = (size_t)min(t * 128 + w * 32 + li, n_rows - 1);
#pragma unroll
      for (int m = 0; m < KG; ++m) {
        const float4 x0 = ld4t(tab, row * D + 16 * m + 8 * h, F16), x1 = ld4t(tab, row * D + 16 * m + 8 * h + 4, F16);
        const float x[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
        unsigned u[8][3];
#pragma unroll
        for (int e = 0; e < 8; ++e) split3(x[e], u[e][0], u[e][1], u[e][2]);
#pragma unroll
        for (int p = 0; p < 3; ++p)
          a[m][p] = make_uint4((u[0][p] >> 16) | (u[1][p] & 0xFFFF0000u), (u[2][p] >> 16) | (u[3][p] & 0xFFFF0000u),
                               (u[4][p] >> 16) | (u[5][p] & 0xFFFF0000u), (u[6][p] >> 16) | (u[7][p] & 0xFFFF0000u));
      }
    }
    // (the compiler has waited for the row loads above: nothing of this tile is in the queue yet; the previous tile's ring slots were
    // released by its last barrier)
    request(0); request(1);
    for (int j = 0; j < NT; ++j) {
      // tile j has landed when at most the requests / stores issued behind it are outstanding
      if (j == 0) wait_vm<PT>();                               // behind tile 0: tile 1
      else if (j == 1) wait_vm<PT + 16>();                     // tile 2, C stores of tile 0
      else if (j == NT - 1) wait_vm<32>();                     // C stores of tiles NT - 3, NT - 2
      else wait_vm<16 + PT + 16>();                            // C stores of j - 2, tile j + 1, C stores of j - 1
      lds_barrier();                                           // every wave's part of tile j is in; slot (j + 2) % 3 (tile j - 1) has been read
      if (j + 2 < NT) request(j + 2);
      const uint4* cur = s_b + (j % 3) * FR + lane;
      f32x16 hi, lo;
#pragma unroll
      for (int r = 0; r < 16; ++r) { hi[r] = 0.f; lo[r] = 0.f; }
#pragma unroll
      for (int m = 0; m < KG; ++m) {
        const uint4 b1 = cur[(m * 3 + 0) * 64], b2 = cur[(m * 3 + 1) * 64], b3 = cur[(m * 3 + 2) * 64];
        hi = mfma32b(a[m][0], b1, hi);
        lo = mfma32b(a[m][0], b3, lo); lo = mfma32b(a[m][1], b2, lo); lo = mfma32b(a[m][2], b1, lo);
        lo = mfma32b(a[m][0], b2, lo); lo = mfma32b(a[m][1], b1, lo);
      }
      const int col = j * 32 + li;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int rr = min(t * 128 + w * 32 + c_row(r, lane), n_rows);
        C[(size_t)rr * N + col] = hi[r] + lo[r];
      }
    }
    lds_barrier();                                             // the last column tile has been read: the ring is free
  }
}

// ax = x . ui^T + bi.  Spatial: POI half through the GEMM (K = D, B = the first D columns of ui), distance-bin
// half + bias from the per-bin table (te_ztab_kernel).  Plain GRU: one table, bias in the epilogue.
template <int D, bool F16>
static void te_launch_ax_t(const TeArgs& A, int num_cu, hipStream_t st) {
  const dim3 grid(((num_cu * 2 + 7) / 8) * 8), block(TE_BLOCK);
  const int n = A.n_seq;
  if (A.bintab) {
    if constexpr (D >= 128) {
      hipLaunchKernelGGL(te_ztab_kernel, dim3(A.n_dist + 1), dim3(3 * D), 0, st, A, A.ztab);
      if (A.fwd_tab) {
        // forward table: ptab = lt . ui[:, :D]^T over the n_item + 1 table rows (identity "gather": the same kernel reads the rows of a
        // half table too), columns gate-interleaved like ztab's; te_rec_fwd16<FT> gathers ptab[p_t] + ztab[dp_t]
        if constexpr (D == 128) {
          if (A.rec_split) {      // split products: te_pack has left the gate-interleaved POI half of ui as bf16 x 3 fragments (te_pack_jobs)
            hipLaunchKernelGGL((te_ptab_s3_kernel<D, F16>), dim3(num_cu * 2), block, sizeof(uint4) * 3 * (D / 16) * 3 * 64, st, A.lt, A.iota + A.n_item + 1, A.pUiP3, A.ptab);
            return;
          }
        }
        hipLaunchKernelGGL(te_uiperm_kernel, dim3(3 * D * D / 1024), dim3(256), 0, st, A);
        NtArgs P{nullptr, 0, A.lt, nullptr, A.iota, nullptr, D, A.uiP, D, A.ptab, 3 * D, nullptr, A.iota + A.n_item + 1, 3 * D, D, nullptr, nullptr};
        hipLaunchKernelGGL((te_gemm_ntk_kernel<false, true, D, D, 3 * D, D, 3 * D, false, F16>), grid, block, 0, st, P);
      } else {
        NtArgs P{nullptr, 0, A.lt, nullptr, A.row_p, nullptr, D, A.ui, 2 * D, A.G, 3 * D, nullptr, A.soff + n, 3 * D, D, A.ztab, A.row_dp};
        // split products for TRAINING launches (predict keeps the float32-input MFMAs: its sts are held to 1e-5 at dim 256 too - 1.07e-5 on split products)
        if (A.rec_split && !A.predict && !(A.dbg & 512)) hipLaunchKernelGGL((te_gemm_ntk_kernel<false, true, D, D, 3 * D, 2 * D, 3 * D, true, F16, true>), grid, block, 0, st, P);
        else hipLaunchKernelGGL((te_gemm_ntk_kernel<false, true, D, D, 3 * D, 2 * D, 3 * D, true, F16>), grid, block, 0, st, P);      // (POI_TE_DBG bit 512: float32-input MFMAs, for A/B runs)
      }
    }
  } else if (A.spatial) {
    NtArgs P{nullptr, 0, A.lt, A.di, A.row_p, A.row_dp, D, A.ui, 2 * D, A.G, 3 * D, A.bi, A.soff + n, 3 * D, 2 * D, nullptr, nullptr};
    hipLaunchKernelGGL((te_gemm_ntk_kernel<true, true, 2 * D, D, 3 * D, 2 * D, 3 * D, false, F16>), grid, block, 0, st, P);
  } else {
    NtArgs P{nullptr, 0, A.lt, A.di, A.row_p, nullptr, D, A.ui, D, A.G, 3 * D, A.bi, A.soff + n, 3 * D, D, nullptr, nullptr};
    if constexpr (D >= 128) hipLaunchKernelGGL((te_gemm_ntk_kernel<true, true, D, D, 3 * D, D, 3 * D, false, F16>), grid, block, 0, st, P);
    else hipLaunchKernelGGL((te_gemm_nt_kernel<true, true, F16>), grid, block, 0, st, P);
  }
}

// ax = x . ui^T + bi.  Spatial: POI half through the GEMM (K = D, B = the first D columns of ui), distance-bin
// half + bias from the per-bin table (te_ztab_kernel).  Plain GRU: one table, bias in the epilogue.
template <int D>
static void te_launch_ax(const TeArgs& A, int num_cu, hipStream_t st) {
  if (A.lt_f16) te_launch_ax_t<D, true>(A, num_cu, st);
  else te_launch_ax_t<D, false>(A, num_cu, st);
}

template <int D>
static hipError_t te_one_t(TeArgs& A, float alpha, float lambda, int l_cap, hipStream_t st, Timing* tm) {
  if constexpr (D > 128) { return hipErrorInvalidValue; } else {
  PackJobs J; te_pack_jobs(A, J, true);
  const int n_ax = 3 * D / 16, n_pk = 8, NB = A.n_dist + 1;
  tm->begin("te_prep", st);
  hipLaunchKernelGGL(te_one_in_kernel<D>, dim3(n_ax + 1 + n_pk * J.n), dim3(TE_BLOCK), 0, st, A, J, n_ax, n_pk);
  tm->end(st);
  if (A.xfwd) {      // exact forward: the recurrence of the one sequence in float64 on the vector ALUs (te_rec_fwd1x)
    hipError_t xe = A.xrec1 ? hipSuccess : launch_te_xfwd(A, 64, st, tm, 2);      // (16-row tile recurrence forced: it needs wh's digit fragments)
    if (xe == hipSuccess) xe = launch_te_xfwd(A, 64, st, tm, 1);      // (te_one_in has written gx: the input product in float64)
    if (xe != hipSuccess) return xe;
  } else {
  tm->begin("te_rec_fwd", st);
  hipLaunchKernelGGL((te_rec_fwd1_kernel<D, false>), dim3(1), dim3(4 * D), 0, st, A);
  tm->end(st);
  }
  tm->begin("te_head", st);
  const int n_hwg = (l_cap + 30) / 32 > 0 ? (l_cap + 30) / 32 : 1;
  if (A.spatial) {
    hipError_t e = te_head_dispatch<D>(A, 0, n_hwg, st);
    if (e != hipSuccess) return e;
  } else {
    hipLaunchKernelGGL(te_bpr_head_kernel<D>, dim3(8), dim3(TE_BLOCK), 0, st, A);
  }
  tm->end(st);
  tm->begin("te_rec_bwd", st);
  hipLaunchKernelGGL(te_rec_bwd1_kernel<D>, dim3(1), dim3(4 * D), 0, st, A);
  tm->end(st);
  tm->begin("te_tail", st);
  const int nb = (3 * D / 16) * (A.xw / 64) + (2 * D / 16) * (D / 64) + (D / 16) * (D / 64) + (A.spatial ? ((NB + 15) / 16) * (D / 64) : 0);
  hipLaunchKernelGGL(te_one_out_kernel<D>, dim3(nb + 1 + 3 * l_cap + 2), dim3(TE_BLOCK), 0, st, A, alpha, lambda, n_hwg, l_cap);
  tm->end(st);
  return hipGetLastError();
  }
}

// The whole step of ONE sequence (Distance2Pre or plain GRU + BPR; write-back included): launch_te_train + launch_te_scatter + launch_dense_apply in five kernels.
// l_cap: the longest sequence of the tables (<= ONE_TMAX + 1).
hipError_t launch_te_one(TeArgs& A, float alpha, float lambda, int l_cap, hipStream_t st, Timing* tm) {
  if (A.dim == 64) return te_one_t<64>(A, alpha, lambda, l_cap, st, tm);
  if (A.dim == 128) return te_one_t<128>(A, alpha, lambda, l_cap, st, tm);
  return hipErrorInvalidValue;
}
bool te_one_supported(int D, bool spatial, int max_len) { (void)spatial; return (D == 64 || D == 128) && max_len <= ONE_TMAX + 1; }

template <int D>
static hipError_t te_train_t(TeArgs& A, int num_cu, hipStream_t st, Timing* tm) {
  const int n = A.n_seq, tiles = (n + 31) / 32;
  PackJobs J; te_pack_jobs(A, J, true);
  const int XW = A.xw;
  // The weight packs (MFMA fragment orders, the exact forward's digit planes and per-bin table) depend on nothing the index preparation
  // produces: with a side stream they run there, next to te_len .. te_gather (four to six small dependent kernels off the main stream's
  // chain); the side stream waits for the caller's stream first - whatever wrote the parameters is ordered before the launch there.
  const bool packs_side = A.side && A.ev_pack && !(A.dbg & 256);      // (POI_TE_DBG bit 256: inline, for A/B runs)
  if (packs_side) {
    if (hipEventRecord(A.ev_start, st) != hipSuccess || hipStreamWaitEvent(A.side, A.ev_start, 0) != hipSuccess) return hipGetLastError();
    if (J.n) hipLaunchKernelGGL(te_pack_kernel, dim3(64, J.n), dim3(TE_BLOCK), 0, A.side, J);
    hipLaunchKernelGGL(te_transpose_kernel, dim3((XW + 31) / 32, (3 * D + 31) / 32), dim3(256), 0, A.side, A.ui, A.uiT, 3 * D, XW);
    if (A.xfwd) { hipError_t xe = launch_te_xfwd(A, num_cu, A.side, tm, 3); if (xe != hipSuccess) return xe; }
    if (hipEventRecord(A.ev_pack, A.side) != hipSuccess) return hipGetLastError();
  }
  tm->begin("te_prep", st);
  hipLaunchKernelGGL(te_len_kernel, dim3((n + 255) / 256), dim3(256), 0, st, A);
  hipLaunchKernelGGL(te_scan_kernel, dim3(1), dim3(1024), 0, st, A, num_cu);
  hipLaunchKernelGGL(te_rowmap_kernel, dim3((n + POI_NWAVE * TE_SEQ_PER_WAVE - 1) / (POI_NWAVE * TE_SEQ_PER_WAVE)), dim3(TE_BLOCK), 0, st, A);
  if (A.xcomp) {      // (behind te_rowmap's marks)
    hipLaunchKernelGGL(te_xcount_kernel, dim3(TE_XBLK), dim3(256), 0, st, A);
    hipLaunchKernelGGL(te_xassign_kernel, dim3(TE_XBLK), dim3(256), 0, st, A);
  }
  if (!packs_side) {
    if (J.n) hipLaunchKernelGGL(te_pack_kernel, dim3(64, J.n), dim3(TE_BLOCK), 0, st, J);
    hipLaunchKernelGGL(te_transpose_kernel, dim3((XW + 31) / 32, (3 * D + 31) / 32), dim3(256), 0, st, A.ui, A.uiT, 3 * D, XW);
  }
  if (!A.side) { hipError_t se = launch_te_sort(A, st); if (se != hipSuccess) return se; }
  tm->end(st);
  // The slot sort (+ the S-row assignment and the bin chain's chunk offsets behind it) is needed from te_psum on: it runs on the side stream,
  // next to te_rec_fwd - a latency chain that leaves most of the chip idle.  NOT next to the float32 te_gemm_ax of rounds 1 - 3: its grid is
  // exactly two persistent workgroups per CU, and a co-resident sort kernel whose LDS displaces one of them pushes that workgroup's whole
  // tile list into a second round (measured: ax 1.6 -> 2.3 ms when its LDS grew by 1 KB).  Exact forward (round 4): forked right behind the
  // index preparation, next to te_gather / te_gemmx as well - the chain then ends inside te_rec_fwdx instead of beside te_head.
  auto fork_sort = [&]() -> hipError_t {
    if (hipEventRecord(A.ev_slots, st) != hipSuccess || hipStreamWaitEvent(A.side, A.ev_slots, 0) != hipSuccess) return hipGetLastError();
    const long sp = tm->span_begin("te_sort", A.side);      // (a span: it overlaps the main stream's regions - the slot sort is part of the scatter, hidden beside te_rec_fwd)
    hipError_t se = launch_te_sort(A, A.side); if (se != hipSuccess) return se;
    if (A.ppoi && !(A.dbg & 1024)) { se = launch_te_passign(A, A.side); if (se != hipSuccess) return se; }
    if (A.early_bins && A.bintab) { se = launch_te_dprep(A, A.side); if (se != hipSuccess) return se; }
    tm->span_end(sp, A.side);
    return hipEventRecord(A.ev_sorted, A.side);
  };
  const bool sort_early = A.side && A.xfwd && !(A.dbg & 2048);      // (POI_TE_DBG bit 2048: behind te_gemm_ax, for A/B runs)
  if (sort_early) { hipError_t se = fork_sort(); if (se != hipSuccess) return se; }
  tm->begin("te_gather", st);
  hipLaunchKernelGGL(te_gather_kernel<D>, dim3(num_cu * 8), dim3(TE_BLOCK), 0, st, A);
  tm->end(st);
  if (packs_side && hipStreamWaitEvent(st, A.ev_pack, 0) != hipSuccess) return hipGetLastError();
  if (A.xfwd) {      // exact forward (te_xfwd.hip): the input product in fixed point on the int8 matrix cores, float64 tables
    hipError_t xe = launch_te_xfwd(A, num_cu, st, tm, packs_side ? 4 : 0);
    if (xe != hipSuccess) return xe;
  } else {
  tm->begin("te_gemm_ax", st);
  te_launch_ax<D>(A, num_cu, st);
  tm->end(st);
  }
  if (A.side && !sort_early) { hipError_t se = fork_sort(); if (se != hipSuccess) return se; }
  if (A.xfwd) {
    hipError_t xe = launch_te_xfwd(A, num_cu, st, tm, 1);
    if (xe != hipSuccess) return xe;
  } else {
  tm->begin("te_rec_fwd", st);
  if constexpr (D >= 128) {
    if (A.rec32 && A.rec_split) hipLaunchKernelGGL((te_rec_fwd32_kernel<D, false, D / 32, true>), dim3((n + 31) / 32), dim3(D * 2), sizeof(short) * 2 * 3 * 32 * (D + 8), st, A);
    else if (A.rec32) hipLaunchKernelGGL((te_rec_fwd32_kernel<D, false, D / 32>), dim3((n + 31) / 32), dim3(D * 2), sizeof(float) * 2 * 32 * (D + 4), st, A);
  }
  if constexpr (D <= 128) {
    if (!A.rec32) {
      const dim3 g((n + 15) / 16), b(D * 4);
      const size_t ldsf = sizeof(float) * 2 * 16 * (D + 4), ldss = sizeof(short) * (2 * 3 * 16 * (D + 8) + 3 * D * D);
      if (A.rec1) hipLaunchKernelGGL((te_rec_fwd1_kernel<D, false>), dim3(n), dim3(4 * D), 0, st, A);
      else
      // (forward table = large launches, where the kernel is bound by its HBM streams, not by the matrix pipe: 300 us for 782 tiles
      // either way - and the two-step table prefetch next to the split planes does not fit the register file: float32-input MFMAs)
      if (A.fwd_tab) hipLaunchKernelGGL((te_rec_fwd16_kernel<D, false, true>), g, b, ldsf, st, A);
      else if (A.rec_split) hipLaunchKernelGGL((te_rec_fwd16_kernel<D, false, false, true>), g, b, ldss, st, A);
      else hipLaunchKernelGGL((te_rec_fwd16_kernel<D, false>), g, b, ldsf, st, A);
    }
  }
  tm->end(st);
  }
  tm->begin("te_head", st);
  if (A.spatial) {
    hipError_t e = te_head_dispatch<D>(A, 0, A.n_head, st);
    if (e != hipSuccess) return e;
  } else {
    hipLaunchKernelGGL(te_bpr_head_kernel<D>, dim3(num_cu * 8), dim3(TE_BLOCK), 0, st, A);
  }
  tm->end(st);
  if (A.hot_early) {      // the hot rows' chunk sums need the sorted entries (side stream, long done), gcoef and H: beside te_rec_bwd instead of in the tail
    if (hipEventRecord(A.ev_hr0, st) != hipSuccess || hipStreamWaitEvent(A.side, A.ev_hr0, 0) != hipSuccess) return hipGetLastError();
    const long sp = tm->span_begin("te_hot_early", A.side);
    hipError_t he = launch_te_hot_reduce(A, num_cu, A.side); if (he != hipSuccess) return he;
    tm->span_end(sp, A.side);
    if (hipEventRecord(A.ev_hr1, A.side) != hipSuccess) return hipGetLastError();
  }
  tm->begin("te_rec_bwd", st);
  if constexpr (D >= 128) {
    if (A.rec32 && A.rec_split) hipLaunchKernelGGL((te_rec_bwd32_kernel<D, D / 32, true>), dim3((n + 31) / 32), dim3(D * 2), sizeof(short) * 3 * 32 * (3 * D + 16), st, A);
    else if (A.rec32) hipLaunchKernelGGL((te_rec_bwd32_kernel<D, D / 32>), dim3((n + 31) / 32), dim3(D * 2), sizeof(float) * (32 * (D + 4) + 32 * (2 * D + 4)), st, A);
  }
  if constexpr (D <= 128) {
    if (!A.rec32) {
      if (A.hyb) {        // hybrid recurrences: the leading sequences per sequence on side2 WHILE the rest runs in tiles here (the split: te_scan)
        const bool serial = (A.dbg & 8192) != 0;      // (POI_TE_DBG bit 8192: one after the other on the main stream, for A/B runs)
        hipStream_t s2 = serial ? st : A.side2;
        if (!serial && (hipEventRecord(A.ev_h2, st) != hipSuccess || hipStreamWaitEvent(A.side2, A.ev_h2, 0) != hipSuccess)) return hipGetLastError();
        hipLaunchKernelGGL(te_rec_bwd1_kernel<D>, dim3(min(n, num_cu)), dim3(4 * D), 0, s2, A);
        if (!serial && hipEventRecord(A.ev_h3, A.side2) != hipSuccess) return hipGetLastError();
        hipLaunchKernelGGL((te_rec_bwd16t_kernel<D>), dim3((n + 15) / 16), dim3(D * 4), sizeof(short) * (3 * 16 * (3 * D + 16) + 3 * D * D), st, A);
        if (!serial && hipStreamWaitEvent(st, A.ev_h3, 0) != hipSuccess) return hipGetLastError();
      } else
      if (A.rec1) hipLaunchKernelGGL(te_rec_bwd1_kernel<D>, dim3(min(n, num_cu * (D <= 64 ? 4 : 1))), dim3(4 * D), 0, st, A);      // persistent: the workgroups one CU holds at a time
      else if (A.rec_split && !(A.dbg & 128)) hipLaunchKernelGGL((te_rec_bwd16t_kernel<D>), dim3((n + 15) / 16), dim3(D * 4), sizeof(short) * (3 * 16 * (3 * D + 16) + 3 * D * D), st, A);
      else if (A.rec_split) hipLaunchKernelGGL((te_rec_bwd16_kernel<D, true>), dim3((n + 15) / 16), dim3(D * 4), sizeof(short) * (3 * 16 * (3 * D + 16) + 3 * D * D), st, A);      // (POI_TE_DBG bit 128: one unit of four sequences per lane, for A/B runs)
      else hipLaunchKernelGGL((te_rec_bwd16_kernel<D, false>), dim3((n + 15) / 16), dim3(D * 4), sizeof(float) * (16 * (D + 4) + 16 * (2 * D + 4)), st, A);
    }
  }
  tm->end(st);
  // per-sequence losses and the fixed-order partial sums only need te_head / te_rec_bwd: two small kernels that
  // fit beside the GEMMs' workgroups (14 registers, no LDS) instead of two more serial launches at the end
  const int n_fin = (n + TE_BLOCK - 1) / TE_BLOCK, n_out = 3 * D + (A.spatial ? A.n_dist + 4 : 0);
  auto finalize = [&](hipStream_t s) {
    tm->begin("te_finalize", s);
    hipLaunchKernelGGL(te_finalize_kernel, dim3(n_fin), dim3(TE_BLOCK), 0, s, A);
    hipLaunchKernelGGL(te_parts_kernel, dim3((n_out + POI_NWAVE - 1) / POI_NWAVE), dim3(TE_BLOCK), 0, s, A, A.rec1 ? n : A.rec32 ? (n + 31) / 32 : (n + 15) / 16, n_fin);
    tm->end(s);
  };
  if (A.side) {
    if (hipEventRecord(A.ev_bwd, st) != hipSuccess || hipStreamWaitEvent(A.side, A.ev_bwd, 0) != hipSuccess) return hipGetLastError();
    finalize(A.side);
    if (hipEventRecord(A.ev_fin, A.side) != hipSuccess) return hipGetLastError();
  }
  if (A.ppoi) {
    // per-POI sums of DA over the sorted entries (the sort ran on the side stream next to te_rec_fwd)
    if (A.side && hipStreamWaitEvent(st, A.ev_sorted, 0) != hipSuccess) return hipGetLastError();
    tm->begin("te_psum", st);
    hipError_t pe = launch_te_psum(A, num_cu, st);
    if (pe != hipSuccess) return pe;
    tm->end(st);
  }
  // the distance-bin chain of the write-back needs nothing the main stream still has to produce once te_psum has formed the hot bins' partials
  auto bins_fork = [&]() -> hipError_t {
    if (hipEventRecord(A.ev_bwd, st) != hipSuccess || hipStreamWaitEvent(A.side, A.ev_bwd, 0) != hipSuccess) return hipGetLastError();
    return launch_te_bins(A, A.bin_alpha, A.bin_lambda, num_cu, A.side, tm);
    // (the caller puts the dense write-back behind the chain on the side stream and records ev_slots: abi.hip)
  };
  const bool bins_side = A.side && A.early_bins && A.bintab;
  // (round 6) beside te_wgrad: with the hot bins summed by te_psum the chain reads a fifth of DA (57 us instead of 177) and now costs te_wgrad +14 us for
  // -22 us of te_tail (12500 users: -9 us per launch, 4096 users: -15).  POI_TE_DBG bit 4096: beside te_gemm_dx as in round 5, for A/B runs
  const bool bins_first = bins_side && !(A.dbg & 4096);
  if (bins_first) { hipError_t be = bins_fork(); if (be != hipSuccess) return be; }
  tm->begin("te_wgrad", st);
  {
    constexpr int T = (D % 128 == 0) ? 128 : 64;
    const int jobs = te_wgrad_jobs(D, A.n_dist, A.spatial != 0, A.bintab != 0);
    const dim3 grid(A.ppoi ? A.wg_slots : jobs * A.n_kc);
    if (A.lt_f16) hipLaunchKernelGGL((te_wgrad_kernel<D, T, true>), grid, dim3(TE_BLOCK), 0, st, A, A.n_kc);
    else hipLaunchKernelGGL((te_wgrad_kernel<D, T, false>), grid, dim3(TE_BLOCK), 0, st, A, A.n_kc);
  }
  tm->end(st);
  // (the dense write-back the caller queues behind the chain on the side stream reads te_wgrad's slabs: the side stream waits for te_wgrad HERE, behind the
  // chain's kernels - they run beside it, the write-back after it)
  if (bins_first && (hipEventRecord(A.ev_bwd, st) != hipSuccess || hipStreamWaitEvent(A.side, A.ev_bwd, 0) != hipSuccess)) return hipGetLastError();
  // it starts here, next to te_gemm_dx (350 tiles on 512 workgroup slots), instead of at the fork in launch_te_scatter, which joins on ev_slots
  // (ev_bwd / ev_slots: both streams passed them long ago).  (Round 4, te_wgrad on split products: started right behind te_rec_bwd instead, the chain
  // costs te_wgrad +105 us and te_psum +16 for -73 us of te_tail: 1870 -> 1855 us per launch, inside the noise between boxes - left here.)
  if (bins_side && !bins_first) { hipError_t be = bins_fork(); if (be != hipSuccess) return be; }
  tm->begin("te_gemm_dx", st);
  {
    // dx = DA . ui: K = 3D is always wide enough for the compile-time-K kernel.  With the per-bin table only the
    // POI half of dx is needed (N = D, written into the first D columns of the 2D-wide X rows).
    // (ppoi: one row per distinct step-input POI - S . ui[:, :D] - instead of one per step)
    NtArgs P{A.ppoi ? A.S : A.G, 3 * D, nullptr, nullptr, nullptr, nullptr, 0, A.uiT, 3 * D, A.X, XW, nullptr, A.ppoi ? A.cnt + 4 : A.soff + n, A.bintab ? D : XW, 3 * D, nullptr, nullptr};
    const dim3 grid(((num_cu * 2 + 7) / 8) * 8), block(TE_BLOCK);
    // split products (POI_TE_DBG bit 512: float32-input MFMAs, for A/B runs); dims <= 128: at K = 768 the unrolled 48-stage tile spills (404 -> 823 us at dim 256)
    const bool sp3 = A.rec_split && D <= 128 && !(A.dbg & 512);
    if (A.bintab && sp3) hipLaunchKernelGGL((te_gemm_ntk_kernel<false, false, 3 * D, 0, D, 3 * D, 2 * D, false, false, true>), grid, block, 0, st, P);
    else if (A.bintab) hipLaunchKernelGGL((te_gemm_ntk_kernel<false, false, 3 * D, 0, D, 3 * D, 2 * D>), grid, block, 0, st, P);
    else if (A.spatial && sp3) hipLaunchKernelGGL((te_gemm_ntk_kernel<false, false, 3 * D, 0, 2 * D, 3 * D, 2 * D, false, false, true>), grid, block, 0, st, P);
    else if (A.spatial) hipLaunchKernelGGL((te_gemm_ntk_kernel<false, false, 3 * D, 0, 2 * D>), grid, block, 0, st, P);
    else if (sp3) hipLaunchKernelGGL((te_gemm_ntk_kernel<false, false, 3 * D, 0, D, 3 * D, D, false, false, true>), grid, block, 0, st, P);
    else hipLaunchKernelGGL((te_gemm_ntk_kernel<false, false, 3 * D, 0, D>), grid, block, 0, st, P);
  }
  tm->end(st);
  if (A.side) { if (hipStreamWaitEvent(st, A.ev_fin, 0) != hipSuccess) return hipGetLastError(); }
  else finalize(st);
  return hipGetLastError();
}

static hipError_t te_optin_lds() {
  // the streaming recurrent kernels need more than the default 64 KB of dynamic LDS at D = 256 (99 KB backward)
  static DeviceOnce once;      // (per device, thread-safe: poi_common.h)
  return once.run([]() -> hipError_t {
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&te_rec_bwd32_kernel<256, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, 112 * 1024);
  if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&te_rec_fwd32_kernel<256, false, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
  if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&te_rec_fwd32_kernel<256, true, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
  if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&te_rec_bwd32_kernel<128, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
  // split-operand recurrent kernels at D = 128: 96 KB of third weight planes next to the operand planes
  auto optin = [&](const void* f) { if (e == hipSuccess) e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024); };
  optin(reinterpret_cast<const void*>(&te_rec_fwd16_kernel<128, false, false, true>)); optin(reinterpret_cast<const void*>(&te_rec_fwd16_kernel<128, true, false, true>));
  optin(reinterpret_cast<const void*>(&te_rec_fwd16_kernel<128, true, true, true>));
  optin(reinterpret_cast<const void*>(&te_rec_bwd16_kernel<128, true>)); optin(reinterpret_cast<const void*>(&te_rec_bwd16t_kernel<128>));
  // streaming kernels on split products: three bf16 planes per operand tile (101 KB forward, 147 KB backward at D = 256)
  optin(reinterpret_cast<const void*>(&te_rec_fwd32_kernel<256, false, 8, true>)); optin(reinterpret_cast<const void*>(&te_rec_fwd32_kernel<256, true, 8, true>));
  optin(reinterpret_cast<const void*>(&te_rec_bwd32_kernel<256, 8, true>));
  optin(reinterpret_cast<const void*>(&te_rec_fwd32_kernel<128, false, 4, true>)); optin(reinterpret_cast<const void*>(&te_rec_fwd32_kernel<128, true, 4, true>));
  optin(reinterpret_cast<const void*>(&te_rec_bwd32_kernel<128, 4, true>));
  // chunked head on split products: 77 KB at D = 128, 101 KB at D = 256
  optin(reinterpret_cast<const void*>(&te_head_big3_kernel<128, 0>)); optin(reinterpret_cast<const void*>(&te_head_big3_kernel<128, 1>));
  optin(reinterpret_cast<const void*>(&te_head_big3_kernel<256, 0>)); optin(reinterpret_cast<const void*>(&te_head_big3_kernel<256, 1>));
  optin(reinterpret_cast<const void*>(&te_head_big3_kernel<64, 0>)); optin(reinterpret_cast<const void*>(&te_head_big3_kernel<64, 1>));
  // heads of dim 256 with 8 bin tiles: 67 KB
  optin(reinterpret_cast<const void*>(&te_head3_kernel<256, 8>)); optin(reinterpret_cast<const void*>(&te_head_kernel<256, 8, 0>)); optin(reinterpret_cast<const void*>(&te_head_kernel<256, 8, 1>));
  // forward table on split products: three 24 KB ring slots
  optin(reinterpret_cast<const void*>(&te_ptab_s3_kernel<128, false>)); optin(reinterpret_cast<const void*>(&te_ptab_s3_kernel<128, true>));
  return e;
  });
}

#ifdef TE_HEAD_PROF
static void te_head_prof_dump(const char* what) {
  unsigned long long h[4][10];
  if (hipDeviceSynchronize() != hipSuccess || hipMemcpyFromSymbol(h, HIP_SYMBOL(g_head_prof), sizeof(h)) != hipSuccess) return;
  static const char* nm[10] = {"wait top barrier", "logits mfma + Ot write", "wait barrier 2", "prefetch + softmax", "wait barrier 3", "ge loads, DL copy, d bs", "DH mfma", "DH store", "stage", "(te_head_big3: stage)"};
  for (int w = 0; w < 4; ++w) {
    unsigned long long tot = 0;
    for (int i = 0; i < 10; ++i) tot += h[w][i];
    fprintf(stderr, "[te_head prof %s] wave %d: total %.3e cycles;", what, w, (double)tot);
    for (int i = 0; i < 10; ++i) fprintf(stderr, " %s %.1f%%;", nm[i], 100.0 * (double)h[w][i] / (double)(tot ? tot : 1));
    fprintf(stderr, "\n");
  }
  unsigned long long z[4][10] = {};
  if (hipMemcpyToSymbol(HIP_SYMBOL(g_head_prof), z, sizeof(z)) != hipSuccess) return;
}
#endif

hipError_t launch_te_train(TeArgs& A, int num_cu, hipStream_t st, Timing* tm) {
  hipError_t e = te_optin_lds();
  if (e != hipSuccess) return e;
#ifdef TE_HEAD_PROF
  struct Dump { ~Dump() { te_head_prof_dump("train launch"); } } dump_at_exit;
#endif
  if (A.dim == 64) return te_train_t<64>(A, num_cu, st, tm);
  if (A.dim == 128) return te_train_t<128>(A, num_cu, st, tm);
  if (A.dim == 256) return te_train_t<256>(A, num_cu, st, tm);
  return hipErrorInvalidValue;
}

template <int D>
static hipError_t te_predict_t(TeArgs& A, int num_cu, hipStream_t st, Timing* tm) {
  const int n = A.n_seq, tiles = (n + 31) / 32;
  PackJobs J; te_pack_jobs(A, J, false);
  tm->begin("te_predict", st);
  hipLaunchKernelGGL(te_len_kernel, dim3((n + 255) / 256), dim3(256), 0, st, A);
  hipLaunchKernelGGL(te_scan_kernel, dim3(1), dim3(1024), 0, st, A, num_cu);
  hipLaunchKernelGGL(te_rowmap_kernel, dim3((n + POI_NWAVE * TE_SEQ_PER_WAVE - 1) / (POI_NWAVE * TE_SEQ_PER_WAVE)), dim3(TE_BLOCK), 0, st, A);
  if (J.n) hipLaunchKernelGGL(te_pack_kernel, dim3(64, J.n), dim3(TE_BLOCK), 0, st, J);
  if (A.xfwd) {      // exact forward (te_xfwd.hip): the same input product and recurrence as the training launches, final state -> hts
    hipError_t xe = launch_te_xfwd(A, num_cu, st, tm, 0);
    if (xe == hipSuccess) xe = launch_te_xfwd(A, num_cu, st, tm, 1);
    if (xe != hipSuccess) return xe;
  } else {
  te_launch_ax<D>(A, num_cu, st);
  if constexpr (D >= 128) {
    if (A.rec32 && A.rec_split) hipLaunchKernelGGL((te_rec_fwd32_kernel<D, true, D / 32, true>), dim3((n + 31) / 32), dim3(D * 2), sizeof(short) * 2 * 3 * 32 * (D + 8), st, A);
    else if (A.rec32) hipLaunchKernelGGL((te_rec_fwd32_kernel<D, true, D / 32>), dim3((n + 31) / 32), dim3(D * 2), sizeof(float) * 2 * 32 * (D + 4), st, A);
  }
  if constexpr (D <= 128) {
    if (!A.rec32) {
      const dim3 g((n + 15) / 16), b(D * 4);
      const size_t ldsf = sizeof(float) * 2 * 16 * (D + 4), ldss = sizeof(short) * (2 * 3 * 16 * (D + 8) + 3 * D * D);
      if (A.rec1) hipLaunchKernelGGL((te_rec_fwd1_kernel<D, true>), dim3(n), dim3(4 * D), 0, st, A);
      // (prediction has no per-step stores: with the forward table it is the matrix pipe that bounds it, not the memory system - split
      // products although the table prefetch spills ~30 registers next to the planes)
      else if (A.fwd_tab && A.rec_split) hipLaunchKernelGGL((te_rec_fwd16_kernel<D, true, true, true>), g, b, ldss, st, A);
      else if (A.fwd_tab) hipLaunchKernelGGL((te_rec_fwd16_kernel<D, true, true>), g, b, ldsf, st, A);
      else if (A.rec_split) hipLaunchKernelGGL((te_rec_fwd16_kernel<D, true, false, true>), g, b, ldss, st, A);
      else hipLaunchKernelGGL((te_rec_fwd16_kernel<D, true>), g, b, ldsf, st, A);
    }
  }
  }
  hipError_t e = hipSuccess;
  if (A.sts) e = te_head_dispatch<D>(A, 1, num_cu * 2 < tiles ? num_cu * 2 : tiles, st);
  tm->end(st);
  return e != hipSuccess ? e : hipGetLastError();
}

hipError_t launch_te_predict(TeArgs& A, int num_cu, hipStream_t st, Timing* tm) {
  hipError_t e = te_optin_lds();
  if (e != hipSuccess) return e;
  if (A.dim == 64) return te_predict_t<64>(A, num_cu, st, tm);
  if (A.dim == 128) return te_predict_t<128>(A, num_cu, st, tm);
  if (A.dim == 256) return te_predict_t<256>(A, num_cu, st, tm);
  return hipErrorInvalidValue;
}

}  // namespace poi
