// What the memory system gives the gather / scatter kernels of the training step (round 6): rows of ROWB bytes fetched through an index
// list out of a buffer of `mb` megabytes - the access pattern of te_psum (1536-byte DA rows in POI order), te_scatter (512-byte h rows in
// table-row order) and te_gather (512-byte table rows) - against the same bytes streamed.  A wave owns whole rows (16 bytes per lane),
// U rows in flight per wave, the rows of a group summed and ONE row written per group of 64 (the segmented sums' output side).
//   gather_rate <row_bytes: 512 | 1536> <buffer MB> <mode: 0 sequential, 1 random permutation, 2 random with repeats (zipf-like reuse)> [flush 0 none | 1 write 512 MB | 2 read 512 MB | 3 read-flush, then the buffer itself written]
// Prints GB/s of the bytes fetched; run a few sizes to see L2 (4 MB / XCD), the 256 MB cache behind it, and HBM.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

template <int LPR, int U>      // LPR lanes per row (16 B each), U row-loads in flight per lane group
__global__ __launch_bounds__(256) void gather_k(const float4* __restrict__ buf, const int* __restrict__ idx, int n, float4* __restrict__ out) {
  constexpr int GPW = 64 / LPR > 0 ? 64 / LPR : 1;      // row groups per wave
  const int lane = threadIdx.x & 63, wave = (blockIdx.x * 256 + threadIdx.x) >> 6, nw = (gridDim.x * 256) >> 6;
  const int grp = lane / LPR, c = lane % LPR;
  float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int i0 = wave * GPW * U; i0 < n; i0 += nw * GPW * U) {
    int id[U];
#pragma unroll
    for (int u = 0; u < U; ++u) id[u] = idx[min(i0 + u * GPW + grp, n - 1)];
    float4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = buf[(size_t)id[u] * LPR + c];
#pragma unroll
    for (int u = 0; u < U; ++u) { a.x += v[u].x; a.y += v[u].y; a.z += v[u].z; a.w += v[u].w; }
  }
  out[(size_t)wave * 64 + lane] = a;      // (every load feeds the result: one row per wave written)
}
// 1536-byte rows: 96 lanes per row = thread = one float4 column group, 384-thread... here: 3 x 32-lane thirds, a wave takes two rows' halves -
// simpler and equivalent for the memory system: treat the row as three 512-byte pieces handled by three lane groups of one wave pair
template <int U>
__global__ __launch_bounds__(384) void gather3_k(const float* __restrict__ buf, const int* __restrict__ idx, int n, float* __restrict__ out) {
  // te_psum's own shape: one workgroup per 64 consecutive list entries, thread = column, U rows in flight
  const int col = threadIdx.x;
  for (int r = blockIdx.x; r * 64 < n; r += gridDim.x) {
    float acc = 0.f;
    for (int u0 = 0; u0 < 64; u0 += U) {
      float v[U];
#pragma unroll
      for (int u = 0; u < U; ++u) v[u] = buf[(size_t)idx[min(r * 64 + u0 + u, n - 1)] * 384 + col];
#pragma unroll
      for (int u = 0; u < U; ++u) acc += v[u];
    }
    out[(size_t)r * 384 + col] = acc;
  }
}
template <int U>
__global__ __launch_bounds__(384) void gather3v_k(const float4* __restrict__ buf, const int* __restrict__ idx, int n, float4* __restrict__ out) {
  // the same rows with 16-byte loads: 96 threads per row, four rows per pass of a 384-thread workgroup
  const int sub = threadIdx.x / 96, c = threadIdx.x % 96;
  for (int r = blockIdx.x; r * 64 < n; r += gridDim.x) {
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int u0 = 0; u0 < 16; u0 += U) {
      float4 v[U];
#pragma unroll
      for (int u = 0; u < U; ++u) v[u] = buf[(size_t)idx[min(r * 64 + 4 * (u0 + u) + sub, n - 1)] * 96 + c];
#pragma unroll
      for (int u = 0; u < U; ++u) { a.x += v[u].x; a.y += v[u].y; a.z += v[u].z; a.w += v[u].w; }
    }
    out[(size_t)r * 384 + threadIdx.x] = a;
  }
}
__global__ void flush_k(float4* p, size_t n4) {
  for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) p[i] = make_float4(1.f, 1.f, 1.f, 1.f);
}
__global__ void rflush_k(const float4* p, size_t n4, float4* out) {      // flush by READING (clean lines)
  float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
  for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) { const float4 v = p[i]; a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w; }
  if (a.x == 123.f) out[0] = a;
}

int main(int argc, char** argv) {
  const int rowb = argc > 1 ? atoi(argv[1]) : 512;
  const size_t mb = argc > 2 ? atoll(argv[2]) : 118;
  const int mode = argc > 3 ? atoi(argv[3]) : 1;
  const int flush = argc > 4 ? atoi(argv[4]) : 1;
  const size_t rows = mb * 1000000ull / rowb;
  const int n = (int)rows;                       // every row fetched once (modes 0, 1); mode 2: n draws with repeats
  std::vector<int> idx(n);
  std::mt19937 rng(12345);
  for (int i = 0; i < n; ++i) idx[i] = i;
  if (mode == 1) std::shuffle(idx.begin(), idx.end(), rng);
  if (mode == 2) { for (int i = 0; i < n; ++i) idx[i] = (int)(rng() % rows); std::sort(idx.begin(), idx.end()); std::shuffle(idx.begin(), idx.end(), rng); }
  if (mode == 3) {      // sorted-by-key access with local disorder: blocks of 64 consecutive rows visited in random order (a sequence's rows are neighbours)
    std::vector<int> blk((n + 63) / 64); for (size_t b = 0; b < blk.size(); ++b) blk[b] = (int)b;
    std::shuffle(blk.begin(), blk.end(), rng);
    for (int i = 0; i < n; ++i) idx[i] = std::min(n - 1, blk[i / 64] * 64 + i % 64);
  }
  float4* buf; int* didx; float4* out; float4* fl;
  hipMalloc(&buf, rows * rowb + 4096); hipMalloc(&didx, n * 4ull); hipMalloc(&out, rows * rowb / 8 + (64 << 20)); hipMalloc(&fl, 512ull << 20);
  hipMemset(buf, 0, rows * rowb);
  hipMemcpy(didx, idx.data(), n * 4ull, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto run = [&](const char* name, auto launch) {
    float best = 1e30f;
    for (int it = 0; it < 5; ++it) {
      if (flush == 1) hipLaunchKernelGGL(flush_k, dim3(2048), dim3(256), 0, 0, fl, (512ull << 20) / 16);
      if (flush == 2) hipLaunchKernelGGL(rflush_k, dim3(2048), dim3(256), 0, 0, fl, (512ull << 20) / 16, out);
      if (flush == 3) {      // the producer pattern: the buffer itself is WRITTEN right before it is gathered (te_rec_bwd -> te_psum), behind a read-flush
        hipLaunchKernelGGL(rflush_k, dim3(2048), dim3(256), 0, 0, fl, (512ull << 20) / 16, out);
        hipLaunchKernelGGL(flush_k, dim3(2048), dim3(256), 0, 0, buf, rows * rowb / 16);
      }
      hipEventRecord(e0, 0); launch(); hipEventRecord(e1, 0); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); if (it) best = std::min(best, ms);
    }
    printf("rows of %4d B, buffer %4zu MB, mode %d, flush %d, %-28s %8.1f us  %7.1f GB/s\n", rowb, mb, mode, flush, name, best * 1e3, (double)n * rowb / best / 1e6);
  };
  if (rowb == 512) {
    for (int g : {1024, 2048, 4096}) {
      char nm[64];
      snprintf(nm, 64, "32 lanes/row U=4 grid %d", g); run(nm, [&] { hipLaunchKernelGGL((gather_k<32, 4>), dim3(g), dim3(256), 0, 0, buf, didx, n, out); });
      snprintf(nm, 64, "32 lanes/row U=8 grid %d", g); run(nm, [&] { hipLaunchKernelGGL((gather_k<32, 8>), dim3(g), dim3(256), 0, 0, buf, didx, n, out); });
      snprintf(nm, 64, "32 lanes/row U=16 grid %d", g); run(nm, [&] { hipLaunchKernelGGL((gather_k<32, 16>), dim3(g), dim3(256), 0, 0, buf, didx, n, out); });
    }
  } else {
    for (int g : {768, 1024, 2048}) {
      char nm[64];
      snprintf(nm, 64, "col/thread U=16 grid %d", g); run(nm, [&] { hipLaunchKernelGGL((gather3_k<16>), dim3(g), dim3(384), 0, 0, (const float*)buf, didx, n, (float*)out); });
      snprintf(nm, 64, "col/thread U=64 grid %d", g); run(nm, [&] { hipLaunchKernelGGL((gather3_k<64>), dim3(g), dim3(384), 0, 0, (const float*)buf, didx, n, (float*)out); });
      snprintf(nm, 64, "16 B loads U=8 grid %d", g); run(nm, [&] { hipLaunchKernelGGL((gather3v_k<8>), dim3(g), dim3(384), 0, 0, buf, didx, n, out); });
      snprintf(nm, 64, "16 B loads U=16 grid %d", g); run(nm, [&] { hipLaunchKernelGGL((gather3v_k<16>), dim3(g), dim3(384), 0, 0, buf, didx, n, out); });
    }
  }
  return 0;
}
