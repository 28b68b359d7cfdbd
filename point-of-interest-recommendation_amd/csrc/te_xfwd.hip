// Exact forward pass of the tile engine ("x" kernels): the forward recurrence of the GRU and its input product in ~40-bit fixed point
// on the INT8 matrix cores (v_mfma_i32_16x16x64_i8 / 32x32x32_i8: exact products, exact int32 accumulation), gate math in float64.
//
// Why: the reference runs float64 (Theano floatX, public/GRU.py:57, public/GRU_Spatial.py:52) and BASELINE.json asks for weights
// within 1e-5 after one step.  At dim 128 / 50 positions with the reference's uniform(-0.5, 0.5) init the GRU is saturated and the
// forward map h_{t-1} -> h_t EXPANDS perturbations: a float32 forward pass (any summation order, libm-exact gates) leaves the late
// hidden states 1e-5 off, and the whole update inherits that (tools/precision_split.py: float32 forward + float64 everything else
// 2e-5 of the max-norm; float64 forward + float32 everything else 2e-7).  The backward pass is LINEAR in its carry - its rounding
// errors are amplified exactly like the signal - and stays float32.  So only two things need more than float32: the input product
// G_t = ui . x_t + bi and the chain h_t = GRU(G_t, h_{t-1}).
//
// How (the Ozaki scheme on integer matrix cores): an operand row is scaled by a power of two to |Q| <= 2^38 and cut into five signed
// base-256 digits (int8 planes); the product of two rows is sum_{i + j <= 4} 256^-(i+j) (d_i . e_j) with every digit product summed
// EXACTLY in int32 by the matrix core (|d e| <= 2^14, K <= 256 terms, <= 5 digit pairs per accumulator: < 2^25); the five
// accumulators are combined in float64.  Dropped: digit pairs with i + j >= 5, <= 2^-39 of the row scales - the result carries ~2e-9
// of relative error where float32 accumulation carries 1e-7, at 15 int8 MFMAs of 16 cycles per 64 k against 16 float32 ones of 32
// (the int8 matrix rate is 32x the float32 one on gfx950; a v_mfma_f64_16x16x4_f64 path would cost 4.3x the matrix time of this one).
// h and r * h lie in [-1, 1]: their scale is a constant (2^-38), no exponent search on the chain.
//
// Kernels: te_xpack (weights -> digit planes in MFMA fragment order + per-row scales, once per launch), te_xztab (distance-bin half
// of the input product + bias per bin, float64 FMAs), te_gemmx (POI half of the input product: table rows or gathered step rows),
// te_rec_fwdx (the recurrence, 16-sequence tiles as te_rec_fwd16; same outputs: G := z | r | c, H, RH in float32 for the head and BPTT).
#include "poi_common.h"
#include "poi_kernels.h"

#include <stdio.h>
#include <type_traits>

namespace poi {

typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x16 __attribute__((ext_vector_type(16)));

#define XS 5                       // signed base-256 digits per operand
#define XQB (8 * XS - 2)           // |Q| <= 2^XQB: the leading digit stays inside [-64, 64] (+ carry)

__device__ __forceinline__ void x_lds_barrier() {      // orders LDS traffic only (tile_engine.hip: lds_barrier)
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}
__device__ __forceinline__ double x_pow2(int e) { return __longlong_as_double((long long)(1023 + e) << 52); }
// exponent e with |m| < 2^e (m = 0 and subnormals: -126)
__device__ __forceinline__ int x_exponent(float m) { return (int)((__float_as_uint(m) >> 23) & 255u) - 126; }

// Signed base-256 digits of rint(t), |t| <= 2^XQB: adding 1.5 * 2^52 leaves rint(t) in the low mantissa bits (two's complement);
// adding 0x80 to every digit position makes the digits unsigned bytes u_i = d_i + 128, and u_i ^ 0x80 is d_i as int8.
// lo: digits 4 (byte 0, least significant) .. 1 (byte 3); hi: digit 0 in byte 0.
__device__ __forceinline__ void x_digits(double t, unsigned& lo, unsigned& hi) {
  const unsigned long long b = (unsigned long long)__double_as_longlong(t + 0x1.8p52) + 0x8080808080ull;
  lo = (unsigned)b ^ 0x80808080u;
  hi = ((unsigned)(b >> 32) ^ 0x80u) & 0xFFu;
}
// the same for v in [-1, 1] at the chain's fixed scale 2^38 (one FMA: v 2^38 is exact, the sum rounds once)
__device__ __forceinline__ void x_digits38(double v, unsigned& lo, unsigned& hi) {
  const unsigned long long b = (unsigned long long)__double_as_longlong(__builtin_fma(v, 0x1p38, 0x1.8p52)) + 0x8080808080ull;
  lo = (unsigned)b ^ 0x80808080u;
  hi = ((unsigned)(b >> 32) ^ 0x80u) & 0xFFu;
}
__device__ __forceinline__ unsigned x_digit(unsigned lo, unsigned hi, int s) { return s == 0 ? hi : (lo >> (8 * (XS - 1 - s))) & 0xFFu; }

// 256 sum_w acc_w 256^-w in float64: the digit-pair classes 0|1 and 2|3 are merged in int32 first ((a0 << 8) + a1 and (a2 << 8) + a3 stay
// below 2^31 for K <= 128: |a0| <= 65^2 K, |a2| <= 2^15 K + 2^14 K), then three conversions and two FMAs (every step exact: < 53 bits).
template <int K>
__device__ __forceinline__ double x_merge(int a0, int a1, int a2, int a3, int a4) {
  const int m01 = (a0 << 8) + a1;
  if constexpr (K <= 128) {
    const int m23 = (a2 << 8) + a3;
    return __builtin_fma(__builtin_fma((double)a4, 0x1p-8, (double)m23), 0x1p-16, (double)m01);
  } else {      // K = 256: |a2| <= 2^23.1, (a2 << 8) + a3 no longer fits - classes 2, 3, 4 one by one (the last FMA rounds at 2^-53 of the sum)
    const double t = __builtin_fma(__builtin_fma((double)a4, 0x1p-8, (double)a3), 0x1p-8, (double)a2);
    return __builtin_fma(t, 0x1p-8, (double)m01);
  }
}
// K = 256, digit-pair classes 0 .. 6 (te_gemmx<256>): 256 sum_w a_w 256^-w by Horner from the smallest class (exact up to 53 bits, then rounded at
// 2^-53 of the running sum)
__device__ __forceinline__ double x_merge7(int a0, int a1, int a2, int a3, int a4, int a5, int a6) {
  double t = __builtin_fma((double)a6, 0x1p-8, (double)a5);
  t = __builtin_fma(t, 0x1p-8, (double)a4);
  t = __builtin_fma(t, 0x1p-8, (double)a3);
  t = __builtin_fma(t, 0x1p-8, (double)a2);
  return __builtin_fma(t, 0x1p-8, (double)((a0 << 8) + a1));
}
__device__ __forceinline__ double x_combine(const i32x4 (&acc)[XS], int r) { return x_merge<128>(acc[0][r], acc[1][r], acc[2][r], acc[3][r], acc[4][r]); }

// float64 exp / sigmoid / tanh, branch-free (the gate math sits on the per-step chain): e^x = 2^m T[j] p(r) with n = rint(64 x / ln 2) =
// 64 m + j, T[j] = 2^(j / 64) (64 doubles in LDS), r = x - n ln2 / 64 (Cody-Waite, |r| <= 0.0055) and the degree-4 Taylor polynomial
// (truncation 4e-14 relative); 1 / d by v_rcp_f64 + one Newton step.  ~1e-13 relative: the chain needs ~1e-10 (see the header).
__device__ __forceinline__ double x_exp(double x, const double* __restrict__ T) {
  const double n = __builtin_rint(x * 0x1.71547652b82fep+6);
  double r = __builtin_fma(n, -0x1.62e42fee00000p-7, x);
  r = __builtin_fma(n, -2.9815858269852933e-12, r);
  const int ni = (int)n;
  const double tj = T[ni & 63];
  double p = __builtin_fma(r, 1.0 / 24.0, 1.0 / 6.0);
  p = __builtin_fma(p, r, 0.5);
  p = __builtin_fma(p, r, 1.0);
  p = __builtin_fma(p, r, 1.0);
  return __builtin_amdgcn_ldexp(tj * p, ni >> 6);
}
__device__ __forceinline__ double x_rcp(double d) {       // v_rcp_f64 is good to 4.6e-8 (tools/micro/i8_f64_overlap.hip): one Newton step -> 2e-15
  const double y = __builtin_amdgcn_rcp(d);
  return __builtin_fma(y, __builtin_fma(-d, y, 1.0), y);
}
// (the clamps: the reduction of x_exp needs |x| < 2^31 ln2 / 64.  v_max_f64 / v_min_f64 return the non-NaN operand: a NaN pre-activation -
// diverged or NaN weights - would come out as a finite gate where the float64 reference propagates it.  NANP: x * 0 is 0 for every finite x
// and NaN for a non-finite one - one FMA on the result hands NaN (and, stricter than the reference, +-inf) through.  te_rec_fwdx sits at its
// register limit and cannot keep x alive that long: there the launch is poisoned as a whole through TeArgs.xflag, see x_flag_bad.)
template <bool NANP = false>
__device__ __forceinline__ double x_sigmoid(double x, const double* __restrict__ T) {
  const double v = x_rcp(1.0 + x_exp(-__builtin_fmin(__builtin_fmax(x, -700.0), 700.0), T));
  return NANP ? __builtin_fma(x, 0.0, v) : v;
}
template <bool NANP = false>
__device__ __forceinline__ double x_tanh(double x, const double* __restrict__ T) {
  const double v = __builtin_fma(-2.0, x_rcp(1.0 + x_exp(2.0 * __builtin_fmin(__builtin_fmax(x, -350.0), 350.0), T)), 1.0);
  return NANP ? __builtin_fma(x, 0.0, v) : v;
}
// A non-finite weight or table row seen while the operands of a launch are prepared (te_xpack, te_xztab, te_gemmx: they scan every value
// anyway): the launch id goes into TeArgs.xflag, and te_rec_fwdx starts every sequence of a flagged launch from h_0 = NaN - every hidden
// state, loss and gradient of the launch is NaN, as in the float64 reference one dense SGD step later at the latest.  (No reset needed:
// the id only grows.)
__device__ __forceinline__ void x_flag_bad(const TeArgs& A) { if (A.xflag) atomicMax(A.xflag, A.xlaunch); }

// -------------------------------------------------------------------------------------------------
// te_xpack: rows of a float32 weight matrix -> digit planes in MFMA B-fragment order + the row scales 2^(e - 12)
// (product of two scaled rows: 2^(e_a - 38) 2^(e_b - 38) 256^8 sum_w 256^-w acc_w = 2^(e_a + e_b - 12) sum_w ...; the A side
// contributes 2^e_a - or 2^0 for the fixed-point chain operands h, r * h, whose Q = rint(h 2^38)).
//   frag32 == 0 (te_rec_fwdx, v_mfma_i32_16x16x64_i8): dst[((nt KB + kb) XS + s) 64 + 16 g + j] byte b = digit s of row 16 nt + j, k = 64 kb + 16 g + b
//   frag32 == 1 (te_gemmx,    v_mfma_i32_32x32x32_i8): dst[((nt KB + kb) XS + s) 64 + 32 h + j] byte b = digit s of row 32 nt + j, k = 32 kb + 16 h + b
// A and B fragments use the SAME (lane group, byte) -> k assignment, which is all the contraction needs.
// inter: destination row n = 3 c + g is source row g (rows / 3) + c (the gate-interleaved columns of the forward table).
// -------------------------------------------------------------------------------------------------
struct XPackJob { const float* src; int ld, koff, rows, K, inter, frag32; unsigned char* dst; double* scale; int* flag; int launch; };
struct XPackJobs { XPackJob j[2]; int n; };

__global__ __launch_bounds__(256) void te_xpack_kernel(XPackJobs J) {
  const XPackJob j = J.j[blockIdx.y];
  const int lane = lane_id();
  for (int n = blockIdx.x * 4 + wave_id(); n < j.rows; n += gridDim.x * 4) {
    const int sr = j.inter ? (n % 3) * (j.rows / 3) + n / 3 : n;
    const float* src = j.src + (size_t)sr * j.ld + j.koff;
    float v[4], m = 0.f;
    bool bad = false;      // a NaN / inf weight has no digits: the row's SCALE carries it into every product of the row (the float64 reference propagates it)
#pragma unroll
    for (int i = 0; i < 4; ++i) { const int k = lane + 64 * i; v[i] = k < j.K ? src[k] : 0.f; m = fmaxf(m, fabsf(v[i])); bad |= !(fabsf(v[i]) <= 3.0e38f); }
    m = wave_max(m);
    bad = __any(bad);
    const int e = bad ? 0 : x_exponent(m);
    const double sc = x_pow2(XQB - e);
    if (lane == 0) { j.scale[n] = bad ? __longlong_as_double(0x7FF8000000000000ll) : x_pow2(e - 12); if (bad && j.flag) atomicMax(j.flag, j.launch); }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int k = lane + 64 * i;
      if (k >= j.K) break;
      unsigned lo, hi;
      x_digits((double)v[i] * sc, lo, hi);
#pragma unroll
      for (int s = 0; s < XS; ++s) {
        size_t at;
        if (j.frag32) { const int KB = j.K / 32; at = ((((size_t)(n / 32) * KB + k / 32) * XS + s) * 64 + 32 * ((k % 32) / 16) + n % 32) * 16 + k % 16; }
        else { const int KB = j.K / 64; at = ((((size_t)(n / 16) * KB + k / 64) * XS + s) * 64 + 16 * ((k % 64) / 16) + n % 16) * 16 + k % 16; }
        j.dst[at] = (unsigned char)x_digit(lo, hi, s);
      }
    }
  }
}

// -------------------------------------------------------------------------------------------------
// te_xztab: ztabx[b][g D + c] = di[b] . ui[g D + c][D:] + bi[g D + c] in float64 (products of float32 values are exact in float64).
// Plain GRU (no distance-bin table): one row, the bias.
// -------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(768) void te_xztab_kernel(TeArgs A) {
  __shared__ float drow[256];
  const int D = A.dim, XW = A.xw, b = blockIdx.x, n = threadIdx.x;
  if (A.spatial && n < D) drow[n] = A.di[(size_t)b * D + n];
  __syncthreads();
  if (n >= 3 * D) return;
  const int r = n;                                  // gate-major columns: n = g D + c, the row of ui / bi itself
  double a0 = (double)A.bi[r], a1 = 0.0;
  if (A.spatial) {
    const float* u = A.ui + (size_t)r * XW + D;
    for (int k = 0; k < D; k += 4) {
      const float4 uv = *reinterpret_cast<const float4*>(u + k);
      a0 = __builtin_fma((double)drow[k], (double)uv.x, a0); a1 = __builtin_fma((double)drow[k + 1], (double)uv.y, a1);
      a0 = __builtin_fma((double)drow[k + 2], (double)uv.z, a0); a1 = __builtin_fma((double)drow[k + 3], (double)uv.w, a1);
    }
  }
  const double zv = a0 + a1;
  A.ztabx[(size_t)b * 3 * D + n] = zv;
  if (!(__builtin_fabs(zv) <= 1.0e300)) x_flag_bad(A);      // (bi / di / the distance half of ui hold a NaN or inf)
}

// -------------------------------------------------------------------------------------------------
// te_gemmx: C[r][n] = lt[idx[r]] . uiP[n]  (+ ztabx[zidx[r]][n]),  n = g D + c < 3 D (gate-major columns), in float64 from the digit products.
// A workgroup item = 128 rows x a group of 32-column tiles; a wave keeps ITS 32 rows as resident digit planes (rows fetched straight
// into the fragment layout, row scale = the row's largest exponent, digits cut in registers) and walks the column tiles, whose digit
// fragments (te_xpack, frag32) are staged through LDS once per workgroup.  Rows past the end land in the spare row behind C.
// -------------------------------------------------------------------------------------------------
__device__ __forceinline__ i32x16 x_mfma32(const i32x4& a, const i32x4& b, i32x16 c) { return __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c, 0, 0, 0); }
__device__ __forceinline__ i32x4 x_mfma16(const i32x4& a, const i32x4& b, i32x4 c) { return __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c, 0, 0, 0); }
__device__ __forceinline__ int x_crow(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }      // C/D layout of the 32x32 tile
// byte `b` of four words -> one word
__device__ __forceinline__ unsigned x_gather_byte(unsigned w0, unsigned w1, unsigned w2, unsigned w3, int b) {
  const unsigned sel = (unsigned)b | ((unsigned)(4 + b) << 8);
  const unsigned t01 = __builtin_amdgcn_perm(w1, w0, sel), t23 = __builtin_amdgcn_perm(w3, w2, sel);
  return __builtin_amdgcn_perm(t23, t01, 0x05040100u);
}

struct XGemmArgs {
  const void* tab; int f16;                 // POI table (float32 or IEEE half)
  const int* idx;                           // row r of C <- table row idx[r] (null: r)
  const int* n_ptr; int idx_max;            // number of rows (device); idx clamp
  const uint4* B8; const double* Bs;        // digit fragments + scales of uiP (te_xpack, frag32)
  const double* ztabx; const int* zidx; int z_max;      // epilogue: + ztabx[min(zidx[r], z_max)] (null: nothing)
  double* C;
  int* flag; int launch;                    // x_flag_bad: a non-finite input row poisons the launch
  int ncg;                                  // column groups per row tile (work items = row tiles x ncg)
};

// LDS DMA (tile_engine.hip: te_ptab_s3): 16 bytes per lane straight from global memory into LDS, no staging registers; lane l of the
// wave lands at lds_dst + 16 l.  Issued from inline asm: the waits are counted by hand (s_waitcnt vmcnt(N), the queue retires in order).
template <int N>
__device__ __forceinline__ void x_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void x_glds16(const void* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

// D = 256 (config X): a wave's resident row digits are 160 registers and the two column-tile slots 80 KB - one workgroup per CU, dynamic LDS;
// the int32 merge of the digit-pair classes 2|3 would overflow at K = 256 (x_combine_wide).
template <int D>
__global__ __launch_bounds__(256, (D <= 128 ? 2 : 1)) void te_gemmx_kernel(XGemmArgs P) {
  constexpr int KB = D / 32, N = 3 * D, NT = N / 32, FR = KB * XS * 64;      // uint4 fragments per column tile
  constexpr int PT = (FR + 255) / 256;
  // digit pairs (i, j) with i + j <= XW.  Dims <= 128: XW = 4, the products carry ~2^-39 of the row scales.  Dim 256 (config X): the reference's init
  // makes the chain amplify a perturbation of the pre-activations ~10^6-fold over 50 positions (tests/test_gpu_dim256.py) - the classes 5 and 6 are
  // kept as well (22 MFMAs per 32 k instead of 15): ~2^-55, the float64 rounding of the sum itself.
  constexpr int XW = D <= 128 ? XS - 1 : XS + 1, NC = XW + 1;
  static_assert(D <= 256 && D % 32 == 0, "te_gemmx: int32 accumulators hold five digit pairs of K <= 256 terms");
  extern __shared__ __align__(16) uint4 xg_lds[];
  uint4 (*s_b)[FR] = reinterpret_cast<uint4 (*)[FR]>(xg_lds);   // two column tiles: tile j + 1 lands while tile j is multiplied
  __shared__ double s_rs[4][32];
  const int tid = threadIdx.x, lane = lane_id(), w = wave_id(), li = lane & 31, h = lane >> 5;
  const int n_rows = *P.n_ptr;
  const int n_tile = (n_rows + 127) / 128, ncg = P.ncg, tpg = NT / ncg;
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) void*)&s_b[0][0];
  const unsigned wbase = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)(w * 64) * 16u);
  auto request = [&](int j, int slot) {                        // column tile j -> slot: every wave its 64-fragment pieces
#pragma unroll
    for (int q = 0; q < PT; ++q)
      if (FR % 256 == 0 || (w * 64 + q * 256) < FR)            // (wave-uniform: a wave's piece lies inside the tile or not)
        x_glds16(P.B8 + (size_t)j * FR + tid + q * 256, wbase + (unsigned)(slot * FR + q * 256) * 16u);
  };
  for (int it = blockIdx.x; it < n_tile * ncg; it += gridDim.x) {
    const int t = it / ncg, j0 = (it % ncg) * tpg;
    request(j0, 0);
    // ---- this wave's 32 rows -> digit planes a[kb][s] (lane: row li, k = 32 kb + 16 h + 0..15) ----
    i32x4 a[KB][XS];
    {
      const int row = min(t * 128 + w * 32 + li, n_rows - 1);
      const size_t src = (size_t)(P.idx ? min((unsigned)P.idx[row], (unsigned)P.idx_max) : row) * D;
      float4 x[KB][4];
      unsigned mb = 0u;      // largest |x| as a BIT PATTERN (integer max: a NaN element - pattern above inf's - survives it, v_max_f32 would drop it)
#pragma unroll
      for (int kb = 0; kb < KB; ++kb)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          x[kb][q] = ld4t(P.tab, src + 32 * kb + 16 * h + 4 * q, P.f16);
          mb = max(max(mb, max(__float_as_uint(x[kb][q].x) & 0x7FFFFFFFu, __float_as_uint(x[kb][q].y) & 0x7FFFFFFFu)),
                   max(__float_as_uint(x[kb][q].z) & 0x7FFFFFFFu, __float_as_uint(x[kb][q].w) & 0x7FFFFFFFu));
        }
      mb = max(mb, (unsigned)__shfl_xor((int)mb, 32, 64));
      const bool bad = mb >= 0x7F800000u;      // a NaN / inf element: the row scale carries it into every product of the row
      const int e = bad ? 0 : x_exponent(__uint_as_float(mb));
      const double sc = x_pow2(XQB - e);
      if (h == 0) s_rs[w][li] = bad ? __longlong_as_double(0x7FF8000000000000ll) : x_pow2(e);
      if (bad && P.flag) atomicMax(P.flag, P.launch);
#pragma unroll
      for (int kb = 0; kb < KB; ++kb) {
        unsigned lo[16], hi[16];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          x_digits((double)x[kb][q].x * sc, lo[4 * q], hi[4 * q]); x_digits((double)x[kb][q].y * sc, lo[4 * q + 1], hi[4 * q + 1]);
          x_digits((double)x[kb][q].z * sc, lo[4 * q + 2], hi[4 * q + 2]); x_digits((double)x[kb][q].w * sc, lo[4 * q + 3], hi[4 * q + 3]);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          a[kb][0][q] = (int)x_gather_byte(hi[4 * q], hi[4 * q + 1], hi[4 * q + 2], hi[4 * q + 3], 0);
#pragma unroll
          for (int s = 1; s < XS; ++s) a[kb][s][q] = (int)x_gather_byte(lo[4 * q], lo[4 * q + 1], lo[4 * q + 2], lo[4 * q + 3], XS - 1 - s);
        }
      }
    }
    for (int j = j0; j < j0 + tpg; ++j) {
      const int slot = (j - j0) & 1;
      // tile j has landed when at most what was issued behind its request is outstanding: the 16 C stores of tile j - 1 (first tile: the
      // compiler has already waited for the row loads that followed the request - nothing younger is in flight)
      if (j == j0) x_wait_vm<0>(); else x_wait_vm<16>();
      x_lds_barrier();                                         // every wave's pieces are in; the other slot (tile j - 1) has been read; s_rs is visible
      if (j + 1 < j0 + tpg) request(j + 1, slot ^ 1);
      const uint4* cur = s_b[slot];
      i32x16 acc[NC];
#pragma unroll
      for (int s = 0; s < NC; ++s)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[s][r] = 0;
#pragma unroll
      for (int kb = 0; kb < KB; ++kb) {
        i32x4 b[XS];
#pragma unroll
        for (int s = 0; s < XS; ++s) b[s] = __builtin_bit_cast(i32x4, cur[(kb * XS + s) * 64 + lane]);
#pragma unroll
        for (int sa = 0; sa < XS; ++sa)
#pragma unroll
          for (int sb = 0; sb < XS; ++sb)
            if (sa + sb <= XW) acc[sa + sb] = x_mfma32(a[kb][sa], b[sb], acc[sa + sb]);
      }
      const int col = j * 32 + li;
      const double cs = P.Bs[col];
      double zadd[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int grow = t * 128 + w * 32 + x_crow(r, lane);
        zadd[r] = P.ztabx ? P.ztabx[(size_t)(P.zidx ? min((unsigned)P.zidx[min(grow, n_rows - 1)], (unsigned)P.z_max) : 0) * N + col] : 0.0;
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int rr = x_crow(r, lane), grow = t * 128 + w * 32 + rr;
        double v;
        if constexpr (NC == 7) v = x_merge7(acc[0][r], acc[1][r], acc[2][r], acc[3][r], acc[4][r], acc[5][r], acc[6][r]);
        else v = x_merge<D>(acc[0][r], acc[1][r], acc[2][r], acc[3][r], acc[4][r]);
        P.C[(size_t)min(grow, n_rows) * N + col] = __builtin_fma(v, s_rs[w][rr] * (cs * 0x1p-8), zadd[r]);      // (exactly 16 stores, last in the tile)
      }
    }
    x_wait_vm<0>();
    x_lds_barrier();                                           // both slots and s_rs are free for the next item
  }
}

// -------------------------------------------------------------------------------------------------
// te_rec_fwdx: te_rec_fwd16's tiling (one workgroup of D / 16 waves per 16 sequences, wave w owns hidden units [16 w, 16 w + 16) of
// z, r, c and h; two barriers per step), with h_{t-1} / r * h_{t-1} in LDS as five int8 digit planes (fixed point, scale 2^-38), the
// recurrent weights resident as digit fragments (digits 0 - 2 in registers, 3 - 4 - three MFMAs in fifteen - in LDS), the
// pre-activations and the state in float64.  The products are formed TRANSPOSED - the weights are the MFMA's A operand (16 hidden
// units), the state its B operand (16 sequences) - so that a lane holds FOUR CONSECUTIVE hidden units of ONE sequence: the digits of
// its four state values are one 4-byte LDS write per plane (per-value byte writes conflicted four ways and were a microsecond per
// step), its outputs 16-byte stores, its pre-activations 96 contiguous bytes.
// FT: pre-activations = ptabx[p_t] + ztabx[dp_t] (forward table); else gx[row].
// Outputs exactly as te_rec_fwd16: G := z | r | c, H, RH (float32 roundings of the float64 values).
// -------------------------------------------------------------------------------------------------
#ifdef TE_XPROF
// -DTE_XPROF (never in the product build): clock64() stamps between the phases of a step, summed per wave over all workgroups
__device__ unsigned long long g_xprof[8][10];
#define XP_INIT long long xp_t = clock64(); long long xp[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#define XP(i) { const long long xp_n = clock64(); xp[i] += xp_n - xp_t; xp_t = xp_n; }
#define XP_END if (lane_id() == 0) { for (int q_ = 0; q_ < 10; ++q_) atomicAdd(&g_xprof[wave_id() & 7][q_], (unsigned long long)xp[q_]); }
#else
#define XP_INIT
#define XP(i)
#define XP_END
#endif
// z | r | c of four consecutive hidden units: v[4 g + r] = gate g of unit u0 + r.  The float64 tables are GATE-MAJOR (column g D + u): a lane's
// 32 bytes per gate sit next to its neighbours' - the 16 lanes of a row read one 128-byte line per wave and gate.  (Gate-interleaved
// columns, 96 bytes per lane, made every 16-byte load instruction of a wave touch 48 lines: the kernel was bound by the address path.)
struct XG12 { double v[12]; };

template <int D, bool FT, bool PRED = false>      // PRED: poi_gru_predict - all L positions, no per-step stores, the final state -> hts
__global__ __launch_bounds__(D * 4) void te_rec_fwdx_kernel(TeArgs A) {
  constexpr int KB = D / 64, NW = D / 16, SR = 3, SL = XS - SR, LDP = D + 16, PSZ = 16 * LDP;
  static_assert(D <= 128 && D % 64 == 0, "te_rec_fwdx: resident digit fragments and x_combine's int32 merge are sized for K <= 128");
  extern __shared__ __align__(16) unsigned char xlds[];
  unsigned char* Hq = xlds;                          // XS planes x 16 rows x LDP bytes
  unsigned char* RHq = Hq + XS * PSZ;
  uint4* Bl = reinterpret_cast<uint4*>(RHq + XS * PSZ) + (size_t)wave_id() * (3 * KB * SL * 64);      // this wave's low digit fragments [gate][kb][sl]
  __shared__ int s_r0[16], s_ns[16];
  __shared__ double s_t64[64];                       // 2^(j / 64): x_exp
  __shared__ __align__(16) double s_cn[3 * D];       // weight-row scales x 2^-8 (x_combine returns 256 x the digit-pair sum)
  const int lane = lane_id(), w = wave_id(), tid = threadIdx.x, i = lane & 15;
  const int u0 = 16 * w + 4 * (lane >> 4);           // this lane: hidden units u0 .. u0 + 3 of sequence i
  const int tile = blockIdx.x;
  const int k_lo = (!PRED && A.hyb) ? A.hyb_dev[0] : 0;      // hybrid recurrences (TeArgs.hyb): the tiles start behind the sequences of te_rec_fwd1x
  if (k_lo + tile * 16 >= A.n_seq) return;                   // (uniform; the grid is sized for k_lo == 0)
  if (tid < 16) {
    const int k = k_lo + tile * 16 + tid;
    int r0 = 0, ns = 0;
    if (k < A.n_seq) { r0 = A.soff[k]; ns = A.soff[k + 1] - r0; }
    s_r0[tid] = r0; s_ns[tid] = ns;
  }
  if (tid >= 64 && tid < 128) s_t64[tid - 64] = exp2((double)(tid - 64) * (1.0 / 64.0));
  for (int e = tid; e < 3 * D; e += blockDim.x) s_cn[e] = A.xWhS[e] * 0x1p-8;
  for (int e = tid; e < 2 * XS * PSZ / 4; e += blockDim.x) reinterpret_cast<unsigned*>(xlds)[e] = 0u;      // h_0 = 0: all digits zero
  i32x4 bw[3][KB][SR];
#pragma unroll
  for (int gt = 0; gt < 3; ++gt) {
    const int nt = gt * NW + w;
#pragma unroll
    for (int kb = 0; kb < KB; ++kb)
#pragma unroll
      for (int s = 0; s < XS; ++s) {
        const uint4 v = A.xWh8[((size_t)(nt * KB + kb) * XS + s) * 64 + lane];
        if (s < SR) bw[gt][kb][s] = __builtin_bit_cast(i32x4, v);
        else Bl[((gt * KB + kb) * SL + (s - SR)) * 64 + lane] = v;
      }
  }
  __syncthreads();
  int ns_max = 0;
  for (int q = 0; q < 16; ++q) ns_max = max(ns_max, s_ns[q]);
  const int rowb = s_r0[i], nsr = s_ns[i];
  const int Tsp = A.soff[A.n_seq];                   // spare packed row: finished sequences read / write it unconditionally
  // (a launch whose weights / input rows hold a NaN or inf - x_flag_bad - starts from h_0 = NaN: the digits of a state cannot carry it)
  const double h_init = (A.xflag && *A.xflag == A.xlaunch) ? __longlong_as_double(0x7FF8000000000000ll) : 0.0;
  double hcur[4] = {h_init, h_init, h_init, h_init};
  XG12 gc;                                           // pre-activations of the current step
  struct XG4 { double v[4]; };
  auto load3 = [&](XG12& o, const double* __restrict__ rowp) {
#pragma unroll
    for (int g = 0; g < 3; ++g) {
      const XG4 q = *reinterpret_cast<const XG4*>(rowp + g * D + u0);
#pragma unroll
      for (int r = 0; r < 4; ++r) o.v[4 * g + r] = q.v[r];
    }
  };
  XP_INIT

  const unsigned char* hrow = Hq + i * LDP + 16 * (lane >> 4);       // B fragments: sequence i, k = 64 kb + 16 (lane >> 4) + 0 .. 15
  const unsigned char* rrow = RHq + i * LDP + 16 * (lane >> 4);
  auto put_digits = [&](unsigned char* plane0, const double (&v)[4]) {       // four values in [-1, 1] -> 4 bytes per plane at (row i, units u0 ..)
    unsigned lo[4], hi[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) x_digits38(v[r], lo[r], hi[r]);
    unsigned* at = reinterpret_cast<unsigned*>(plane0 + i * LDP + u0);
    at[0] = x_gather_byte(hi[0], hi[1], hi[2], hi[3], 0);
#pragma unroll
    for (int s = 1; s < XS; ++s) at[s * (PSZ / 4)] = x_gather_byte(lo[0], lo[1], lo[2], lo[3], XS - 1 - s);
  };
  // digit products of NG gates that share the state operand, transposed: acc[g][w][r] = sum over the digit pairs (sa, sb), sa + sb = w,
  // of W_sb[gate gt0 + g, unit u0 + r] . H_sa[sequence i]  (15 MFMAs per gate and 64 k; consecutive MFMAs go to different accumulators)
  auto mma = [&](auto& acc, const unsigned char* __restrict__ h0, const int gt0, auto ng) {
    constexpr int NG = decltype(ng)::value;
#pragma unroll
    for (int g = 0; g < NG; ++g)
#pragma unroll
      for (int s = 0; s < XS; ++s) acc[g][s] = i32x4{0, 0, 0, 0};
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
      i32x4 a[XS], lb[NG][SL];
#pragma unroll
      for (int s = 0; s < XS; ++s) a[s] = *reinterpret_cast<const i32x4*>(h0 + s * PSZ + 64 * kb);
#pragma unroll
      for (int g = 0; g < NG; ++g)
#pragma unroll
        for (int s = 0; s < SL; ++s) lb[g][s] = __builtin_bit_cast(i32x4, Bl[(((gt0 + g) * KB + kb) * SL + s) * 64 + lane]);
#pragma unroll
      for (int sa = 0; sa < XS; ++sa)
#pragma unroll
        for (int sb = 0; sb < XS - sa; ++sb)
#pragma unroll
          for (int g = 0; g < NG; ++g)
            acc[g][sa + sb] = x_mfma16(sb < SR ? bw[gt0 + g][kb][sb < SR ? sb : 0] : lb[g][sb >= SR ? sb - SR : 0], a[sa], acc[g][sa + sb]);
    }
  };
  // one step: z | r products (they share the state's digit fragments: one LDS read) -> r gate, r * h digits -> barrier -> c products,
  // z gate (off the chain: behind the c products' issue), c gate, h, h digits -> barrier
  auto compute = [&](int t) {
    i32x4 azr[2][XS], ac[1][XS];
    const bool on = t < nsr;
    const size_t row = (size_t)(on ? rowb + t : Tsp);
    XP(0)
    mma(azr, hrow, 0, std::integral_constant<int, 2>());
    XP(1)
    {
      double rh[4];
      float rv4[4], rh4[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const double rv = x_sigmoid(__builtin_fma(x_combine(azr[1], r), s_cn[D + u0 + r], gc.v[4 + r]), s_t64);
        rh[r] = rv * hcur[r];
        rv4[r] = (float)rv; rh4[r] = (float)rh[r];
      }
      put_digits(RHq, rh);
      if constexpr (!PRED) {
        *reinterpret_cast<float4*>(A.G + row * 3 * D + D + u0) = make_float4(rv4[0], rv4[1], rv4[2], rv4[3]);
        *reinterpret_cast<float4*>(A.RH + row * D + u0) = make_float4(rh4[0], rh4[1], rh4[2], rh4[3]);
      }
    }
    XP(2)
    x_lds_barrier();
    XP(3)
    mma(ac, rrow, 2, std::integral_constant<int, 1>());
    XP(4)
    double zv[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) zv[r] = x_sigmoid(__builtin_fma(x_combine(azr[0], r), s_cn[u0 + r], gc.v[r]), s_t64);
    XP(5)
    float z4[4], c4[4], h4[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const double c = x_tanh(__builtin_fma(x_combine(ac[0], r), s_cn[2 * D + u0 + r], gc.v[8 + r]), s_t64);
      const double hn = on ? __builtin_fma(zv[r], c - hcur[r], hcur[r]) : hcur[r];
      hcur[r] = hn;
      z4[r] = (float)zv[r]; c4[r] = (float)c; h4[r] = (float)hn;
    }
    put_digits(Hq, hcur);
    if constexpr (!PRED) {
      *reinterpret_cast<float4*>(A.G + row * 3 * D + u0) = make_float4(z4[0], z4[1], z4[2], z4[3]);
      *reinterpret_cast<float4*>(A.G + row * 3 * D + 2 * D + u0) = make_float4(c4[0], c4[1], c4[2], c4[3]);
      *reinterpret_cast<float4*>(A.H + row * D + u0) = make_float4(h4[0], h4[1], h4[2], h4[3]);
    }
    XP(6)
    x_lds_barrier();
    XP(7)
  };

  if constexpr (!FT) {
    auto fetch = [&](int t, XG12& n) { load3(n, A.gx + (size_t)(t < nsr ? rowb + t : Tsp) * 3 * D); };
    XG12 nx;
    fetch(0, gc);
    for (int t = 0; t < ns_max; ++t) {
      fetch(t + 1, nx);
      compute(t);
#pragma unroll
      for (int q = 0; q < 12; ++q) {
        asm volatile("" : "+v"(nx.v[q]));      // the wait for the prefetch is counted here, behind the step's stores
        gc.v[q] = nx.v[q];
      }
    }
  } else {
    // forward table: the row ids of step t + 2 and the table rows of step t + 1 are requested at the top of step t.  (Measured and not
    // kept: touching the lines of the row of step t + 3 - one dword per 128-byte line, at the top or at the very end of a step - to bring
    // them to L2 two steps early: 412 -> 444 .. 481 us per 12500-user launch; vmcnt retires in order, a far prefetch sits in every wait.)
    int rp = 0, rz = 0;
    const int* __restrict__ xrow = A.xcomp ? A.row_pc : A.row_p;      // (compact table: te_gather translated the ids)
    auto ids = [&](int t) { const int rr = t < nsr ? rowb + t : Tsp; rp = xrow[rr]; rz = A.spatial ? A.row_dp[rr] : 0; };
    auto rows = [&](XG12& pp, XG12& zz) {
      const int p1 = (int)min((unsigned)rp, (unsigned)A.n_item), z1 = A.spatial ? (int)min((unsigned)rz, (unsigned)A.n_dist) : 0;
      load3(pp, A.ptabx + (size_t)p1 * 3 * D);
      load3(zz, A.ztabx + (size_t)z1 * 3 * D);
    };
    XG12 pn, zn;
    ids(0); rows(pn, zn);
#pragma unroll
    for (int q = 0; q < 12; ++q) gc.v[q] = pn.v[q] + zn.v[q];
    ids(1);
    for (int t = 0; t < ns_max; ++t) {
      rows(pn, zn);                            // step t + 1 (ids loaded a step ago)
      ids(t + 2);
      compute(t);
#pragma unroll
      for (int q = 0; q < 12; ++q) {
        asm volatile("" : "+v"(pn.v[q]), "+v"(zn.v[q]));
        gc.v[q] = pn.v[q] + zn.v[q];
      }
    }
  }
  if constexpr (PRED) {
    const int k = tile * 16 + i;
    if (k < A.n_seq)
      *reinterpret_cast<float4*>(A.hts + (size_t)(A.out_row ? A.out_row[k] : k) * D + u0) = make_float4((float)hcur[0], (float)hcur[1], (float)hcur[2], (float)hcur[3]);
  }
  XP_END
}

// -------------------------------------------------------------------------------------------------
// te_rec_fwd1x: the exact forward recurrence of ONE sequence per workgroup on the vector ALUs, in float64 - small launches (at most one
// workgroup per CU or two) and the one-sequence path (the reference's schedule).  te_rec_fwd1's tiling: 4 D threads, a thread owns FOUR
// outputs and one k-slice (z | r phase: 8 slices of D / 8, c phase: 16 slices of D / 16; adjacent lanes, DPP sums), the recurrent weights
// RESIDENT in registers as float64 (96 per thread at dim 128: converted once - float32 -> float64 is exact), h / r h broadcast from LDS.
// A 16-row tile step of te_rec_fwdx costs 3.6 us whether the tile holds 16 sequences or one (float64 gate math for 16 x 3 D values);
// here a step is 96 FMAs + two gate values per thread: ~1 us.  Same inputs (gx, gate-major) and outputs as te_rec_fwdx.
// -------------------------------------------------------------------------------------------------
template <int CTRL>
__device__ __forceinline__ double x_dpp(double v) {
  const long long b = __double_as_longlong(v);
  const int lo = __builtin_amdgcn_update_dpp(0, (int)b, CTRL, 0xF, 0xF, false), hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), CTRL, 0xF, 0xF, false);
  return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}
template <int N> __device__ __forceinline__ double x_group_sum(double v) {      // sum over N = 8 / 16 adjacent lanes, in every lane
  v += x_dpp<0xB1>(v);                     // quad_perm [1,0,3,2]
  v += x_dpp<0x4E>(v);                     // quad_perm [2,3,0,1]
  v += x_dpp<0x141>(v);                    // row_half_mirror
  if (N >= 16) v += x_dpp<0x140>(v);       // row_mirror
  return v;
}
__device__ __forceinline__ double x_pick4(const double (&a)[4], int o) { return o == 0 ? a[0] : o == 1 ? a[1] : o == 2 ? a[2] : a[3]; }

// Round 5: PERSISTENT - the grid is one workgroup per CU and a workgroup walks the launch's sequences k = b, 2 G - 1 - b, 2 G + b, ... (snake
// order over the length-sorted launch: every workgroup gets one sequence of each length class) with its weights loaded ONCE: a launch of
// 1563 sequences was six rounds of workgroups, each paying the longest sequence of its round and a 196 KB weight fetch (te_rec_fwd 217 us;
// the 16-row tiles: 221 us), now ~113 steps per CU.  FT: pre-activations from the forward table (ptabx[p_t] + ztabx[dp_t]) instead of gx.
template <int D, bool PRED = false, bool FT = false>
__global__ __launch_bounds__(4 * D) void te_rec_fwd1x_kernel(TeArgs A) {
  constexpr int LZ = D / 8, LC = D / 16;
  __shared__ __align__(16) double hs[D], rhs[D], zs[D];
  __shared__ double s_t64[64];
  const int tid = threadIdx.x;
  // hybrid recurrences (TeArgs.hyb, training): the leading hyb_dev[0] sequences on hyb_dev[1] workgroups, the rest in tiles (te_rec_fwdx) at the same time
  const int n1 = (!PRED && A.hyb) ? A.hyb_dev[0] : A.n_seq, G = (!PRED && A.hyb) ? A.hyb_dev[1] : (int)gridDim.x, bq = blockIdx.x;
  if (bq >= G) return;
  const int gz = tid >> 3, sz = tid & 7, gc = tid >> 4, sc = tid & 15;
  const int jz = 4 * gz + (sz & 3), jc = 4 * gc + (sc & 3);        // the output this lane finishes (lanes sz / sc < 4)
  const bool isr = jz >= D;
  const int jr = isr ? jz - D : jz;
  double wzr[4][LZ], wc[4][LC];
#pragma unroll
  for (int o = 0; o < 4; ++o) {
#pragma unroll
    for (int i = 0; i < LZ; i += 4) {
      const float4 v = *reinterpret_cast<const float4*>(A.wh + (size_t)(4 * gz + o) * D + sz * LZ + i);
      wzr[o][i] = (double)v.x; wzr[o][i + 1] = (double)v.y; wzr[o][i + 2] = (double)v.z; wzr[o][i + 3] = (double)v.w;
    }
#pragma unroll
    for (int i = 0; i < LC; i += 4) {
      const float4 v = *reinterpret_cast<const float4*>(A.wh + (size_t)(2 * D + 4 * gc + o) * D + sc * LC + i);
      wc[o][i] = (double)v.x; wc[o][i + 1] = (double)v.y; wc[o][i + 2] = (double)v.z; wc[o][i + 3] = (double)v.w;
    }
  }
  if (tid >= 64 && tid < 128) s_t64[tid - 64] = exp2((double)(tid - 64) * (1.0 / 64.0));
  // every lane issues every global store of a step (non-owners write their duplicate into the spare packed row): straight-line code,
  // exact vmcnt waits (te_rec_fwd1)
  // (one of the 128 spare rows behind the packed rows per workgroup: 256 CUs storing their duplicates into ONE row serialised on its lines)
  const size_t Tsp = (size_t)A.soff[A.n_seq] + 1 + (blockIdx.x & 127);
  const bool ownz = sz < 4, ownc = sc < 4;
  float* const dG = A.G + Tsp * 3 * D + (tid % (3 * D));
  float* const dH = A.H + Tsp * D + (tid % D);
  float* const dR = A.RH + Tsp * D + (tid % D);
  const int* __restrict__ xrow = A.xcomp ? A.row_pc : A.row_p;
  // pre-activations of packed row rr, columns jz (z | r) and 2 D + jc (c)
  auto pre = [&](size_t rr, double& ozr, double& oc) {
    if constexpr (FT) {
      const size_t p1 = (size_t)min((unsigned)xrow[rr], (unsigned)A.n_item) * 3 * D, z1 = (size_t)(A.spatial ? min((unsigned)A.row_dp[rr], (unsigned)A.n_dist) : 0u) * 3 * D;
      ozr = A.ptabx[p1 + jz] + A.ztabx[z1 + jz]; oc = A.ptabx[p1 + 2 * D + jc] + A.ztabx[z1 + 2 * D + jc];
    } else { ozr = A.gx[rr * 3 * D + jz]; oc = A.gx[rr * 3 * D + 2 * D + jc]; }
  };
  for (int j = 0;; ++j) {
    const int k = j * G + ((j & 1) ? G - 1 - bq : bq);
    if (k >= n1) break;                      // (workgroup-uniform; the snake's odd legs run downwards: a later even leg may still exist)
    if (tid < D) hs[tid] = 0.0;
    const int r0 = A.soff[k], ns = A.soff[k + 1] - r0;
    __syncthreads();
    double gzr = 0.0, gcc = 0.0;
    if (ns > 0) pre((size_t)r0, gzr, gcc);
    for (int t = 0; t < ns; ++t) {
      const size_t row = (size_t)(r0 + t), rn = (size_t)(r0 + min(t + 1, ns - 1));
      double nzr, nc;
      pre(rn, nzr, nc);
      double a[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int i = 0; i < LZ; i += 2) {
        const double2 x = *reinterpret_cast<const double2*>(hs + sz * LZ + i);
#pragma unroll
        for (int o = 0; o < 4; ++o) a[o] = __builtin_fma(wzr[o][i + 1], x.y, __builtin_fma(wzr[o][i], x.x, a[o]));
      }
#pragma unroll
      for (int o = 0; o < 4; ++o) a[o] = x_group_sum<8>(a[o]);
      {
        const double v = x_sigmoid<true>(x_pick4(a, sz & 3) + gzr, s_t64);
        const double rh = v * hs[jr];
        if (ownz) { if (isr) rhs[jr] = rh; else zs[jr] = v; }
        const bool st = ownz && isr;
        if constexpr (!PRED) {
          *(st ? A.G + row * 3 * D + D + jr : dG) = (float)v;
          *(st ? A.RH + row * D + jr : dR) = (float)rh;
        }
      }
      x_lds_barrier();
      double b[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int i = 0; i < LC; i += 2) {
        const double2 x = *reinterpret_cast<const double2*>(rhs + sc * LC + i);
#pragma unroll
        for (int o = 0; o < 4; ++o) b[o] = __builtin_fma(wc[o][i + 1], x.y, __builtin_fma(wc[o][i], x.x, b[o]));
      }
#pragma unroll
      for (int o = 0; o < 4; ++o) b[o] = x_group_sum<16>(b[o]);
      {
        const double c = x_tanh<true>(x_pick4(b, sc & 3) + gcc, s_t64);
        const double z = zs[jc], hp = hs[jc];
        const double hn = __builtin_fma(z, c - hp, hp);
        if (ownc) hs[jc] = hn;               // (the c phase reads rhs only; the lanes that share jc are in one wave)
        if constexpr (!PRED) {
          *(ownc ? A.G + row * 3 * D + jc : dG) = (float)z;
          *(ownc ? A.G + row * 3 * D + 2 * D + jc : dG) = (float)c;
          *(ownc ? A.H + row * D + jc : dH) = (float)hn;
        }
      }
      x_lds_barrier();
      asm volatile("" : "+v"(nzr), "+v"(nc));      // the wait for the prefetch is counted HERE, behind this step's stores
      gzr = nzr; gcc = nc;
    }
    if (PRED && tid < D) A.hts[(size_t)(A.out_row ? A.out_row[k] : k) * D + tid] = (float)hs[tid];
    __syncthreads();                         // hs is reset for the next sequence: every lane has read its last values
  }
}

// -------------------------------------------------------------------------------------------------
// te_rec_fwdd: the exact forward recurrence for D = 256 (config X) - the recurrent products in FLOAT64 on the matrix cores
// (v_mfma_f64_16x16x4_f64), weights streamed from L2.
//
// Why not the digit scheme of te_rec_fwdx: at dim 256 the digit planes of wh are 983 KB - not resident in one CU - and the reference's
// uniform(-0.5, 0.5) init makes the chain expand perturbations ~10^6-fold over 50 positions (tools/precision_split.py with PS_D=256: a
// float32 forward pass leaves the update O(1) off; a chain held to 2^-38 ~1e-6; float64 forward + float32 everything else 0.5 - 4e-6).
// The f64 MFMA runs at the float64 vector rate (a 16x16x4 block is 16 passes), i.e. 2.3x the matrix time of fifteen streamed int8 digit
// products - but it needs no digit cutting, no scales and a quarter of the weight bytes (float32 fragments converted on the fly, 786 KB
// per step of a tile against 983 KB of digit planes), its 64-cycle issue shadow hides the float64 gate math of the other waves, and its
// result is the float64 product itself: no conditioning of the chain can push it outside the bar.
//
// Tiling: 16 sequences per workgroup of D / 32 waves; wave w owns hidden units [32 w, 32 w + 32) of z, r and c (two 16-unit MFMA tiles
// per gate).  Products TRANSPOSED as in te_rec_fwdx: weights = A operand (16 units x 4 k), state = B operand (4 k x 16 sequences);
// the weight rows are packed in the order unit(rho) = 4 (rho & 3) + (rho >> 2) so that the f64 C layout (row = (lane >> 4) + 4 reg) gives
// a lane FOUR CONSECUTIVE units of ONE sequence - the gate / store code of te_rec_fwdx.  State h_{t-1}, r * h_{t-1}: float64 in LDS,
// k-major (hT[k][16 sequences]: the four lane groups of a B fragment read four consecutive 128-byte rows, no bank conflicts).
// MFMA j of k-block kq contracts k = 16 kq + 4 j + (lane >> 4): the A fragments of four MFMAs are ONE 16-byte load per lane (te_xwpackd).
// Pre-activations (te_gemmx: int8 digits, float64 tables), gates, outputs: as te_rec_fwdx.
// -------------------------------------------------------------------------------------------------
typedef double f64x4 __attribute__((ext_vector_type(4)));

// wh (3, D, D) float32 -> dst[((gate NT + ut) KQ + kq) 64 + lane] = float4 { wh[gate][16 ut + 4 (i & 3) + (i >> 2)][16 kq + 4 j + g] : j = 0..3 }, i = lane & 15, g = lane >> 4
__global__ __launch_bounds__(256) void te_xwpackd_kernel(const float* __restrict__ wh, float4* __restrict__ dst, int D) {
  const int NT = D / 16, KQ = D / 16, total = 3 * NT * KQ * 64;
  for (int e = blockIdx.x * 256 + threadIdx.x; e < total; e += gridDim.x * 256) {
    const int lane = e & 63, kq = (e >> 6) % KQ, tu = (e >> 6) / KQ;      // tu = gate NT + ut
    const int i = lane & 15, g = lane >> 4;
    const int gate = tu / NT, ut = tu % NT;
    const float* src = wh + ((size_t)gate * D + 16 * ut + 4 * (i & 3) + (i >> 2)) * D + 16 * kq + g;
    dst[e] = make_float4(src[0], src[4], src[8], src[12]);
  }
}

template <int D, bool FT, bool PRED = false>
__global__ __launch_bounds__(D * 2) void te_rec_fwdd_kernel(TeArgs A) {
  constexpr int NW = D / 32, NT = D / 16, KQ = D / 16;
  static_assert(D % 32 == 0 && D >= 64, "te_rec_fwdd: a wave owns 32 units of every gate");
  extern __shared__ __align__(16) unsigned char xlds[];
  double* hT = reinterpret_cast<double*>(xlds);      // [D][16]
  double* rhT = hT + D * 16;
  __shared__ int s_r0[16], s_ns[16];
  __shared__ double s_t64[64];
  const int lane = lane_id(), w = wave_id(), tid = threadIdx.x, i = lane & 15, g = lane >> 4;
  const int tile = blockIdx.x;
  if (tid < 16) {
    const int k = tile * 16 + tid;
    int r0 = 0, ns = 0;
    if (k < A.n_seq) { r0 = A.soff[k]; ns = A.soff[k + 1] - r0; }
    s_r0[tid] = r0; s_ns[tid] = ns;
  }
  if (tid >= 64 && tid < 128) s_t64[tid - 64] = exp2((double)(tid - 64) * (1.0 / 64.0));
  for (int e = tid; e < 2 * D * 16; e += blockDim.x) hT[e] = 0.0;      // h_0 = 0
  __syncthreads();
  int ns_max = 0;
  for (int q = 0; q < 16; ++q) ns_max = max(ns_max, s_ns[q]);
  const int rowb = s_r0[i], nsr = s_ns[i];
  const int Tsp = A.soff[A.n_seq];                   // spare packed row: finished sequences read / write it unconditionally
  const float4* __restrict__ Wp = reinterpret_cast<const float4*>(A.xWh8);
  // this lane: units ub[u] .. ub[u] + 3 (u = 0, 1: the wave's two unit tiles) of sequence i
  int ub[2];
#pragma unroll
  for (int u = 0; u < 2; ++u) ub[u] = 32 * w + 16 * u + 4 * g;
  double hcur[2][4] = {{0.0, 0.0, 0.0, 0.0}, {0.0, 0.0, 0.0, 0.0}};
  struct XG4 { double v[4]; };
  // products of NG gates (gt0 ..) x this wave's two unit tiles with the state in sT: acc[gate][tile] (C layout: reg r = unit ub + r)
  auto mma = [&](auto& acc, const double* __restrict__ sT, const int gt0, auto ng) {
    constexpr int NG = decltype(ng)::value;
    // the weight fragments of k-block kq + 2 are requested while kq is multiplied (an L2 round trip under a full chip is longer than one
    // block's sixteen MFMAs); the four state values of a block are read from LDS in front of its MFMAs
    float4 ac[NG][2], an[NG][2], af[NG][2];
    auto fetch = [&](float4 (&o)[NG][2], int kq) {
#pragma unroll
      for (int q = 0; q < NG; ++q)
#pragma unroll
        for (int u = 0; u < 2; ++u) o[q][u] = Wp[(((size_t)(gt0 + q) * NT + 2 * w + u) * KQ + kq) * 64 + lane];
    };
#pragma unroll
    for (int q = 0; q < NG; ++q)
#pragma unroll
      for (int u = 0; u < 2; ++u) acc[q][u] = f64x4{0.0, 0.0, 0.0, 0.0};
    fetch(ac, 0); fetch(an, 1);
#pragma unroll 2
    for (int kq = 0; kq < KQ; ++kq) {
      fetch(af, min(kq + 2, KQ - 1));
      const double* bp = sT + (size_t)(16 * kq + g) * 16 + i;
      double b[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) b[j] = bp[64 * j];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
#pragma unroll
        for (int q = 0; q < NG; ++q)
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            const float a = j == 0 ? ac[q][u].x : j == 1 ? ac[q][u].y : j == 2 ? ac[q][u].z : ac[q][u].w;
            acc[q][u] = __builtin_amdgcn_mfma_f64_16x16x4f64((double)a, b[j], acc[q][u], 0, 0, 0);
          }
      }
#pragma unroll
      for (int q = 0; q < NG; ++q)
#pragma unroll
        for (int u = 0; u < 2; ++u) { ac[q][u] = an[q][u]; an[q][u] = af[q][u]; }
    }
  };
  // pre-activations of gate gt, both unit tiles, of packed row `row` (FT: table rows p1 / z1)
  auto pre = [&](XG4 (&o)[2], int gt, size_t row, int p1, int z1) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      if constexpr (FT) {
        const XG4 a = *reinterpret_cast<const XG4*>(A.ptabx + (size_t)p1 * 3 * D + gt * D + ub[u]);
        const XG4 b = *reinterpret_cast<const XG4*>(A.ztabx + (size_t)z1 * 3 * D + gt * D + ub[u]);
#pragma unroll
        for (int r = 0; r < 4; ++r) o[u].v[r] = a.v[r] + b.v[r];
      } else {
        o[u] = *reinterpret_cast<const XG4*>(A.gx + row * 3 * D + gt * D + ub[u]);
      }
    }
  };
  const int* __restrict__ xrow = A.xcomp ? A.row_pc : A.row_p;
  for (int t = 0; t < ns_max; ++t) {
    const bool on = t < nsr;
    const size_t row = (size_t)(on ? rowb + t : Tsp);
    int p1 = 0, z1 = 0;
    if constexpr (FT) {
      p1 = (int)min((unsigned)xrow[row], (unsigned)A.n_item);
      z1 = A.spatial ? (int)min((unsigned)A.row_dp[row], (unsigned)A.n_dist) : 0;
    }
    XG4 gz[2], gr[2], gcc[2];
    pre(gz, 0, row, p1, z1); pre(gr, 1, row, p1, z1);
    f64x4 azr[2][2], ac1[1][2];
    mma(azr, hT, 0, std::integral_constant<int, 2>());
    double zv[2][4];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      double rh[4];
      float rv4[4], rh4[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        zv[u][r] = x_sigmoid<true>(azr[0][u][r] + gz[u].v[r], s_t64);
        const double rv = x_sigmoid<true>(azr[1][u][r] + gr[u].v[r], s_t64);
        rh[r] = rv * hcur[u][r];
        rv4[r] = (float)rv; rh4[r] = (float)rh[r];
        rhT[(size_t)(ub[u] + r) * 16 + i] = rh[r];
      }
      if constexpr (!PRED) {
        *reinterpret_cast<float4*>(A.G + row * 3 * D + D + ub[u]) = make_float4(rv4[0], rv4[1], rv4[2], rv4[3]);
        *reinterpret_cast<float4*>(A.RH + row * D + ub[u]) = make_float4(rh4[0], rh4[1], rh4[2], rh4[3]);
      }
    }
    pre(gcc, 2, row, p1, z1);
    x_lds_barrier();
    mma(ac1, rhT, 2, std::integral_constant<int, 1>());
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      float z4[4], c4[4], h4[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const double c = x_tanh<true>(ac1[0][u][r] + gcc[u].v[r], s_t64);
        const double hn = on ? __builtin_fma(zv[u][r], c - hcur[u][r], hcur[u][r]) : hcur[u][r];
        hcur[u][r] = hn;
        z4[r] = (float)zv[u][r]; c4[r] = (float)c; h4[r] = (float)hn;
        hT[(size_t)(ub[u] + r) * 16 + i] = hn;
      }
      if constexpr (!PRED) {
        *reinterpret_cast<float4*>(A.G + row * 3 * D + ub[u]) = make_float4(z4[0], z4[1], z4[2], z4[3]);
        *reinterpret_cast<float4*>(A.G + row * 3 * D + 2 * D + ub[u]) = make_float4(c4[0], c4[1], c4[2], c4[3]);
        *reinterpret_cast<float4*>(A.H + row * D + ub[u]) = make_float4(h4[0], h4[1], h4[2], h4[3]);
      }
    }
    x_lds_barrier();
  }
  if constexpr (PRED) {
    const int k = tile * 16 + i;
    if (k < A.n_seq)
#pragma unroll
      for (int u = 0; u < 2; ++u)
        *reinterpret_cast<float4*>(A.hts + (size_t)(A.out_row ? A.out_row[k] : k) * D + ub[u]) =
            make_float4((float)hcur[u][0], (float)hcur[u][1], (float)hcur[u][2], (float)hcur[u][3]);
  }
}

// -------------------------------------------------------------------------------------------------
// host side
// -------------------------------------------------------------------------------------------------
bool te_xfwd_supported(int D) { return D == 64 || D == 128 || D == 256; }

size_t te_xfwd_lds(int D) {
  const int KB = D / 64, NW = D / 16, SL = XS - 3;
  return (size_t)2 * XS * 16 * (D + 16) + (size_t)NW * 3 * KB * SL * 64 * 16;
}

// bytes of the digit fragments / doubles of the scales for a (rows x K) operand
size_t te_xfrag_bytes(int rows, int K) { return (size_t)rows * K * XS; }

// D <= 128: te_rec_fwdx (int8 digit products, resident fragments); D = 256: te_rec_fwdd (float64 MFMA, streamed float32 fragments)
template <int D> struct XRec {
  static constexpr bool F64 = D > 128;
  static size_t lds() { return F64 ? sizeof(double) * 2 * D * 16 : te_xfwd_lds(D <= 128 ? D : 128); }
  static dim3 block() { return dim3(F64 ? D * 2 : D * 4); }
  template <bool FT, bool PRED> static const void* fn() {
    if constexpr (F64) return reinterpret_cast<const void*>(&te_rec_fwdd_kernel<D, FT, PRED>);
    else return reinterpret_cast<const void*>(&te_rec_fwdx_kernel<D, FT, PRED>);
  }
  template <bool FT, bool PRED> static void launch(const TeArgs& A, hipStream_t st) {
    const dim3 grid((A.n_seq + 15) / 16);
    if constexpr (F64) hipLaunchKernelGGL((te_rec_fwdd_kernel<D, FT, PRED>), grid, block(), lds(), st, A);
    else hipLaunchKernelGGL((te_rec_fwdx_kernel<D, FT, PRED>), grid, block(), lds(), st, A);
  }
};

template <int D>
static hipError_t te_xfwd_t(const TeArgs& A, int num_cu, hipStream_t st, Timing* tm, int phase) {
  // the LDS opt-in is a per-device attribute of the function (DeviceOnce, poi_common.h)
  static DeviceOnce once;
  {
    const hipError_t oe = once.run([&]() -> hipError_t {
      const void* fns[5] = {XRec<D>::template fn<false, false>(), XRec<D>::template fn<true, false>(), XRec<D>::template fn<false, true>(), XRec<D>::template fn<true, true>(),
                            reinterpret_cast<const void*>(&te_gemmx_kernel<D>)};
      for (const void* f : fns) {
        const hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024);
        if (e != hipSuccess) return e;
      }
      return hipSuccess;
    });
    if (oe != hipSuccess) return oe;
  }
  // the recurrent weights for the recurrence kernel: int8 digit fragments (te_rec_fwdx) or float32 fragments in f64-MFMA order (te_rec_fwdd)
  auto pack_wh = [&](hipStream_t s) {
    if constexpr (XRec<D>::F64) hipLaunchKernelGGL(te_xwpackd_kernel, dim3(3 * D * D / 4 / 256), dim3(256), 0, s, A.wh, reinterpret_cast<float4*>(A.xWh8), D);
  };
  const int n = A.n_seq;
  if (phase == 2) {      // fragments of the recurrent weights only (one-sequence path with the tile recurrence: te_one_in forms gx itself)
    if constexpr (XRec<D>::F64) pack_wh(st);
    else {
      XPackJobs J;
      J.j[0] = XPackJob{A.wh, D, 0, 3 * D, D, 0, 0, reinterpret_cast<unsigned char*>(A.xWh8), A.xWhS, A.xflag, A.xlaunch};
      J.n = 1;
      hipLaunchKernelGGL(te_xpack_kernel, dim3(3 * D / 4, 1), dim3(256), 0, st, J);
    }
    return hipGetLastError();
  }
  if (phase == 0 || phase == 3 || phase == 4) {      // 3: the weight digits and the per-bin table only (no timed region: the side stream); 4: the product only
  if (!A.predict && phase != 3) tm->begin("te_gemm_ax", st);
  if (phase != 4) {
    XPackJobs J;
    J.j[0] = XPackJob{A.ui, A.xw, 0, 3 * D, D, 0, 1, reinterpret_cast<unsigned char*>(A.xUi8), A.xUiS, A.xflag, A.xlaunch};
    J.j[1] = XPackJob{A.wh, D, 0, 3 * D, D, 0, 0, reinterpret_cast<unsigned char*>(A.xWh8), A.xWhS, A.xflag, A.xlaunch};
    J.n = XRec<D>::F64 ? 1 : 2;
    hipLaunchKernelGGL(te_xpack_kernel, dim3(3 * D / 4, J.n), dim3(256), 0, st, J);
    pack_wh(st);
    hipLaunchKernelGGL(te_xztab_kernel, dim3(A.spatial ? A.n_dist + 1 : 1), dim3(3 * D), 0, st, A);
  }
  if (phase != 3) {
    XGemmArgs P;
    P.flag = A.xflag; P.launch = A.xlaunch;
    P.tab = A.lt; P.f16 = A.lt_f16; P.idx_max = A.n_item; P.B8 = A.xUi8; P.Bs = A.xUiS; P.z_max = A.spatial ? A.n_dist : 0;
    int rows_est;
    if (A.xft && A.xcomp) { P.idx = A.xlist; P.n_ptr = A.xcnt; P.ztabx = nullptr; P.zidx = nullptr; P.C = A.ptabx; rows_est = min(A.n_item + 1, A.x_rows_est * 2 / 5 + 1); }
    else if (A.xft) { P.idx = nullptr; P.n_ptr = A.iota + A.n_item + 1; P.ztabx = nullptr; P.zidx = nullptr; P.C = A.ptabx; rows_est = A.n_item + 1; }
    else { P.idx = A.row_p; P.n_ptr = A.soff + n; P.ztabx = A.ztabx; P.zidx = A.spatial ? A.row_dp : nullptr; P.C = A.gx; rows_est = A.x_rows_est; }
    // few row tiles: split the column tiles of a row tile over several workgroups (the slicing of the rows is repeated, it is cheap)
    // (per-step rows: the host only knows the launch's step CAPACITY; sequences average ~40 % of the longest)
    if (!A.xft) rows_est = rows_est * 2 / 5 + 1;
    const int n_tile_est = (rows_est + 127) / 128, NT = 3 * D / 32;
    int ncg = 1;
    while (ncg < NT && n_tile_est * ncg < num_cu && NT % (ncg * 2) == 0) ncg *= 2;
    if (ncg * 2 <= NT && NT % (ncg * 3) == 0 && n_tile_est * ncg < num_cu) ncg *= 3;
    P.ncg = ncg;
    const int grid = min(num_cu * 2, n_tile_est * ncg > 0 ? n_tile_est * ncg : 1);
    hipLaunchKernelGGL(te_gemmx_kernel<D>, dim3(D <= 128 ? grid : min(grid, num_cu)), dim3(256), sizeof(uint4) * 2 * (D / 32) * XS * 64, st, P);
  }
  if (!A.predict && phase != 3) tm->end(st);
  return hipGetLastError();
  }
#ifdef TE_XPROF
  struct Dump { ~Dump() {
    unsigned long long h[8][10];
    if (hipDeviceSynchronize() != hipSuccess || hipMemcpyFromSymbol(h, HIP_SYMBOL(g_xprof), sizeof(h)) != hipSuccess) return;
    static const char* nm[10] = {"prefetch issue / loop", "z|r products", "r gate, digits, stores", "barrier 1", "c products", "z gate", "c gate, h, digits, stores", "barrier 2", "", ""};
    for (int w = 0; w < 8; ++w) {
      unsigned long long tot = 0;
      for (int i = 0; i < 10; ++i) tot += h[w][i];
      if (!tot) continue;
      fprintf(stderr, "[te_rec_fwdx prof] wave %d: total %.3e cycles;", w, (double)tot);
      for (int i = 0; i < 8; ++i) fprintf(stderr, " %s %.1f%%;", nm[i], 100.0 * (double)h[w][i] / (double)tot);
      fprintf(stderr, "\n");
    }
    unsigned long long z[8][10] = {};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_xprof), z, sizeof(z));
  } } dump_at_exit;
#endif
  if (A.predict) {      // (inside the caller's te_predict region)
    if constexpr (!XRec<D>::F64) {
      if (A.xrec1) {
        if (A.xft) hipLaunchKernelGGL((te_rec_fwd1x_kernel<D, true, true>), dim3(min(n, num_cu * (D <= 64 ? 3 : 1))), dim3(4 * D), 0, st, A);
        else hipLaunchKernelGGL((te_rec_fwd1x_kernel<D, true, false>), dim3(min(n, num_cu * (D <= 64 ? 3 : 1))), dim3(4 * D), 0, st, A);
        return hipGetLastError();
      }
    }
    if (A.xft) XRec<D>::template launch<true, true>(A, st);
    else XRec<D>::template launch<false, true>(A, st);
    return hipGetLastError();
  }
  tm->begin("te_rec_fwd", st);
  bool done = false;
  if constexpr (!XRec<D>::F64) {
    if (A.hyb) {        // hybrid recurrences: the leading sequences per sequence on side2 WHILE the rest runs in tiles here (the split: te_hybrid_kernel)
      if (hipEventRecord(A.ev_h0, st) != hipSuccess || hipStreamWaitEvent(A.side2, A.ev_h0, 0) != hipSuccess) return hipGetLastError();
      if (A.xft) hipLaunchKernelGGL((te_rec_fwd1x_kernel<D, false, true>), dim3(min(n, num_cu)), dim3(4 * D), 0, A.side2, A);
      else hipLaunchKernelGGL((te_rec_fwd1x_kernel<D, false, false>), dim3(min(n, num_cu)), dim3(4 * D), 0, A.side2, A);
      if (hipEventRecord(A.ev_h1, A.side2) != hipSuccess) return hipGetLastError();
      if (A.xft) XRec<D>::template launch<true, false>(A, st);
      else XRec<D>::template launch<false, false>(A, st);
      if (hipStreamWaitEvent(st, A.ev_h1, 0) != hipSuccess) return hipGetLastError();
      done = true;
    } else
    if (A.xrec1) {      // persistent: one workgroup per CU walks the launch's sequences
      if (A.xft) hipLaunchKernelGGL((te_rec_fwd1x_kernel<D, false, true>), dim3(min(n, num_cu * (D <= 64 ? 3 : 1))), dim3(4 * D), 0, st, A);
      else hipLaunchKernelGGL((te_rec_fwd1x_kernel<D, false, false>), dim3(min(n, num_cu * (D <= 64 ? 3 : 1))), dim3(4 * D), 0, st, A);
      done = true;
    }
  }
  if (!done) {
    if (A.xft) XRec<D>::template launch<true, false>(A, st);
    else XRec<D>::template launch<false, false>(A, st);
  }
  tm->end(st);
  return hipGetLastError();
}

// te_gemm_ax (phase 0) / te_rec_fwd (phase 1) of a training launch in the exact-forward mode (TeArgs.xfwd); phase 2: the recurrent weights' digit fragments only
hipError_t launch_te_xfwd(const TeArgs& A, int num_cu, hipStream_t st, Timing* tm, int phase) {
  if (A.dim == 64) return te_xfwd_t<64>(A, num_cu, st, tm, phase);
  if (A.dim == 128) return te_xfwd_t<128>(A, num_cu, st, tm, phase);
  if (A.dim == 256) return te_xfwd_t<256>(A, num_cu, st, tm, phase);
  return hipErrorInvalidValue;
}

}  // namespace poi
