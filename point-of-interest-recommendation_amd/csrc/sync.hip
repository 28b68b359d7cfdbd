// Replica reconciliation of the user-sharded multi-GPU path (new - the reference is single-process):
// every rank trains its user shard on a full parameter replica with no data-path collective, and ONCE PER EPOCH
//      delta_r = theta_r - theta_start            (poi_sync_make_delta; per table row also "did this replica move it")
//      D       = sum_r delta_r                    (poi_allreduce_tables: one RCCL all-reduce over xGMI of ONE flat buffer)
//      theta   = theta_start + combine(D)         (poi_sync_apply; the result is the next epoch's theta_start)
// with a combine rule per tensor: SUM (every replica's epoch counts in full: to first order in alpha the epoch
// of a single process over all users), MEAN (model averaging) or MEAN_TOUCHED (per table row: mean over the replicas
// that moved the row - the launch-level batch rule of include/poi_hip.h one level up).
//
// RCCL is bound at run time (dlopen of the librccl the process already holds - torch maps its copy as librccl.so - or the system
// one), so the library itself links against nothing but the HIP runtime and both builds and loads on a box without RCCL: the
// handful of nccl types / enum values the five entry points need are declared here (ABI-stable since NCCL 2.0), not included.
#include "../../include/poi_hip.h"
#include "poi_common.h"

#include <dlfcn.h>
#include <stdio.h>
#include <string.h>

#include <mutex>
#include <string>
#include <vector>

// ---- the slice of the NCCL / RCCL C API this file binds (rccl.h: ncclResult_t, ncclUniqueId, ncclComm_t, ncclDataType_t, ncclRedOp_t)
typedef int ncclResult_t;
enum { ncclSuccess = 0 };
typedef struct { char internal[128]; } ncclUniqueId;
typedef struct ncclComm* ncclComm_t;
typedef int ncclDataType_t;
enum { ncclFloat16 = 6, ncclFloat32 = 7 };
typedef int ncclRedOp_t;
enum { ncclSum = 0 };

namespace poi {

struct Rccl {
  void* h = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  std::string err;
};

static Rccl* rccl() {
  static Rccl R;
  static std::once_flag once;
  std::call_once(once, [] {
    const char* names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so.1"};
    // first the copy this process already mapped, under either soname (torch ships torch/lib/librccl.so): a second RCCL of another
    // version in one process is what must not happen
    void* h = nullptr;
    for (int i = 0; !h && i < 2; ++i) h = dlopen(names[i], RTLD_NOW | RTLD_NOLOAD);
    for (int i = 0; !h && i < 3; ++i) h = dlopen(names[i], RTLD_NOW | RTLD_GLOBAL);
    if (!h) { R.err = std::string("cannot load librccl: ") + (dlerror() ? dlerror() : "?"); return; }
    R.h = h;
    *(void**)&R.GetUniqueId = dlsym(h, "ncclGetUniqueId");
    *(void**)&R.CommInitRank = dlsym(h, "ncclCommInitRank");
    *(void**)&R.CommDestroy = dlsym(h, "ncclCommDestroy");
    *(void**)&R.AllReduce = dlsym(h, "ncclAllReduce");
    *(void**)&R.GetErrorString = dlsym(h, "ncclGetErrorString");
    if (!R.GetUniqueId || !R.CommInitRank || !R.CommDestroy || !R.AllReduce || !R.GetErrorString) { R.err = "librccl lacks the nccl* entry points"; R.h = nullptr; }
  });
  return &R;
}

// ---------------------------------------------------------------------------------------------------------------
// kernels: one wavefront per row of a segment (rows x width floats, width % 4 == 0 or handled by the scalar tail)
// ---------------------------------------------------------------------------------------------------------------
// (f16: `cur` holds IEEE half - config X's fp16 tables; snapshot, deltas and the all-reduce stay float32, the combined value is
// rounded to half once, and the snapshot keeps exactly that rounded value so that replicas stay bit-identical)
__device__ __forceinline__ float ldc(const float* cur, long long i, int f16) { return f16 ? __half2float(reinterpret_cast<const __half*>(cur)[i]) : cur[i]; }

// H (half segments, round 3): a table STORED as half keeps its snapshot as half too (exact: the values are halves) and its delta as half -
// the combined value is rounded to half anyway, and a sum of `world` half deltas carries an error of world x 2^-11 of the DELTA, far
// below the 2^-11 of the VALUE that rounding costs; half the snapshot memory, a quarter... half the all-reduce bytes (config X: 10 -> 5 GB).
template <bool H> struct SyncT { typedef float T; };
template <> struct SyncT<true> { typedef __half T; };
__device__ __forceinline__ float sy_ld(const float* p) { return *p; }
__device__ __forceinline__ float sy_ld(const __half* p) { return __half2float(*p); }
__device__ __forceinline__ void sy_st(float* p, float v) { *p = v; }
__device__ __forceinline__ void sy_st(__half* p, float v) { *p = __float2half_rn(v); }

template <bool H>
__global__ __launch_bounds__(256) void sync_delta_kernel(const float* __restrict__ cur, const typename SyncT<H>::T* __restrict__ base,
                                                         typename SyncT<H>::T* __restrict__ delta, float* __restrict__ touched,
                                                         long long rows, long long width) {
  const int lane = lane_id();
  for (long long r = (long long)blockIdx.x * 4 + wave_id(); r < rows; r += (long long)gridDim.x * 4) {
    bool any = false;
    for (long long j = lane; j < width; j += 64) {
      const float c = ldc(cur, r * width + j, H ? 1 : 0), b = sy_ld(base + r * width + j);
      sy_st(delta + r * width + j, c - b);
      any |= (c != b);               // (the row moved - also when the half delta of a tiny step rounds to zero)
    }
    if (touched) { const bool t = __ballot(any) != 0ull; if (lane == 0) touched[r] = t ? 1.f : 0.f; }
  }
}

// cur <- base + scale * dsum ; base <- cur.   rule 0: scale 1, 1: 1 / world, 2: 1 / max(count[row], 1)
template <bool H>
__global__ __launch_bounds__(256) void sync_apply_kernel(float* __restrict__ cur, typename SyncT<H>::T* __restrict__ base,
                                                         const typename SyncT<H>::T* __restrict__ dsum, const float* __restrict__ count,
                                                         long long rows, long long width, int rule, float inv_world) {
  const int lane = lane_id();
  for (long long r = (long long)blockIdx.x * 4 + wave_id(); r < rows; r += (long long)gridDim.x * 4) {
    float sc = rule == 1 ? inv_world : 1.f;
    if (rule == 2) { const float n = count[r]; sc = 1.f / fmaxf(n, 1.f); }
    for (long long j = lane; j < width; j += 64) {
      const long long i = r * width + j;
      float v = fmaf(sc, sy_ld(dsum + i), sy_ld(base + i));
      if (H) { const __half hv = __float2half_rn(v); reinterpret_cast<__half*>(cur)[i] = hv; base[i] = hv; }      // (the snapshot keeps exactly the rounded value)
      else { cur[i] = v; sy_st(base + i, v); }
    }
  }
}

__global__ __launch_bounds__(256) void sync_copy_kernel(const float* __restrict__ src, float* __restrict__ dst, long long n) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) dst[i] = src[i];
}

// order-independent checksum of a float buffer: 64-bit sum of the 32-bit patterns (equal buffers <=> equal sums,
// up to collisions; used to show that replicas are bit-identical after a reconciliation)
__global__ __launch_bounds__(256) void checksum_kernel(const unsigned* __restrict__ x, long long n, unsigned long long* __restrict__ out) {
  unsigned long long s = 0;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) s += x[i];
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if (lane_id() == 0) atomicAdd(out, s);
}

}  // namespace poi

struct poi_comm {
  ncclComm_t comm = nullptr;
  int world = 1, rank = 0, device = 0;
};

struct poi_sync {
  poi_ctx* ctx = nullptr;
  int device = 0;
  std::vector<poi_sync_seg> segs;
  std::vector<long long> off, cnt_off;          // element offsets into the flat buffers (cnt_off < 0: no touch counts); half segments: into base16 / delta16
  long long n_data = 0, n_total = 0, n_half = 0;
  bool asked16 = false;                         // the caller has seen poi_sync_buffer16 (a caller that owns the collective must all-reduce it too)
  float *base = nullptr, *delta = nullptr;
  __half *base16 = nullptr, *delta16 = nullptr; // snapshot / deltas of the segments stored as half
  hipEvent_t e0 = nullptr, e1 = nullptr;
  double last_ms = 0;
  std::string err;
};

namespace {
thread_local std::string g_sync_err;
int sfail(poi_sync* s, int code, const std::string& m) { g_sync_err = m; if (s) s->err = m; return code; }
int grid_rows(long long rows) { long long g = (rows + 3) / 4; return (int)(g < 1 ? 1 : g > 8192 ? 8192 : g); }
}

extern "C" {

const char* poi_sync_last_error(void) { return g_sync_err.c_str(); }

// Can this process bind librccl (dlopen + the nccl* entry points)?  No RCCL call is made: ncclGetUniqueId on a rank that is not the root
// would leave a bootstrap listener (socket + thread) behind that nothing uses or releases.
int poi_comm_available(void) {
  poi::Rccl* R = poi::rccl();
  if (!R->h) return sfail(nullptr, POI_ENOTSUP, R->err);
  return POI_OK;
}

int poi_comm_unique_id(char* id_host) {
  if (!id_host) return sfail(nullptr, POI_EINVAL, "poi_comm_unique_id: NULL");
  poi::Rccl* R = poi::rccl();
  if (!R->h) return sfail(nullptr, POI_ENOTSUP, R->err);
  ncclUniqueId id;
  const ncclResult_t rc = R->GetUniqueId(&id);
  if (rc != ncclSuccess) return sfail(nullptr, POI_EHIP, std::string("ncclGetUniqueId: ") + R->GetErrorString(rc));
  static_assert(sizeof(id) == POI_UNIQUE_ID_BYTES, "ncclUniqueId size");
  memcpy(id_host, &id, sizeof id);
  return POI_OK;
}

int poi_comm_init_rank(const char* id_host, int world, int rank, int device, poi_comm** out) {
  if (!id_host || !out || world < 1 || rank < 0 || rank >= world) return sfail(nullptr, POI_EINVAL, "poi_comm_init_rank: bad argument");
  poi::Rccl* R = poi::rccl();
  if (!R->h) return sfail(nullptr, POI_ENOTSUP, R->err);
  if (hipSetDevice(device) != hipSuccess) return sfail(nullptr, POI_EHIP, "hipSetDevice failed");
  ncclUniqueId id;
  memcpy(&id, id_host, sizeof id);
  poi_comm* c = new poi_comm();
  c->world = world; c->rank = rank; c->device = device;
  const ncclResult_t rc = R->CommInitRank(&c->comm, world, id, rank);
  if (rc != ncclSuccess) { delete c; return sfail(nullptr, POI_EHIP, std::string("ncclCommInitRank: ") + R->GetErrorString(rc)); }
  *out = c;
  return POI_OK;
}

int poi_comm_destroy(poi_comm* c) {
  if (!c) return POI_OK;
  poi::Rccl* R = poi::rccl();
  if (R->h && c->comm) R->CommDestroy(c->comm);
  delete c;
  return POI_OK;
}

int poi_comm_world(const poi_comm* c) { return c ? c->world : POI_EINVAL; }
int poi_comm_rank(const poi_comm* c) { return c ? c->rank : POI_EINVAL; }

int poi_allreduce_tables(poi_ctx* ctx, poi_comm* comm, float* buf, int64_t n, void* stream) {
  (void)ctx;
  if (!comm || !buf || n < 0) return sfail(nullptr, POI_EINVAL, "poi_allreduce_tables: bad argument");
  if (n == 0) return POI_OK;
  poi::Rccl* R = poi::rccl();
  if (!R->h) return sfail(nullptr, POI_ENOTSUP, R->err);
  if (hipSetDevice(comm->device) != hipSuccess) return sfail(nullptr, POI_EHIP, "hipSetDevice failed");
  const ncclResult_t rc = R->AllReduce(buf, buf, (size_t)n, ncclFloat32, ncclSum, comm->comm, (hipStream_t)stream);
  if (rc != ncclSuccess) return sfail(nullptr, POI_EHIP, std::string("ncclAllReduce: ") + R->GetErrorString(rc));
  return POI_OK;
}

int poi_sync_create(poi_ctx* ctx, int device, const poi_sync_seg* segs_host, int32_t n_seg, poi_sync** out) {
  if (!segs_host || n_seg <= 0 || !out) return sfail(nullptr, POI_EINVAL, "poi_sync_create: bad argument");
  if (hipSetDevice(device) != hipSuccess) return sfail(nullptr, POI_EHIP, "hipSetDevice failed");
  poi_sync* s = new poi_sync();
  s->ctx = ctx; s->device = device;
  long long o = 0, oh = 0;
  for (int i = 0; i < n_seg; ++i) {
    const poi_sync_seg& g = segs_host[i];
    if (!g.cur || g.rows <= 0 || g.width <= 0 || g.rule < POI_SYNC_SUM || g.rule > POI_SYNC_MEAN_TOUCHED || (g.dtype != POI_F32 && g.dtype != POI_F16)) { delete s; return sfail(nullptr, POI_EINVAL, "poi_sync_create: bad segment"); }
    s->segs.push_back(g);
    if (g.dtype == POI_F16) { s->off.push_back(oh); oh += g.rows * g.width; oh = (oh + 7) & ~7ll; }
    else { s->off.push_back(o); o += g.rows * g.width; o = (o + 3) & ~3ll; }
  }
  s->n_data = o; s->n_half = oh;
  for (int i = 0; i < n_seg; ++i) {
    if (segs_host[i].rule == POI_SYNC_MEAN_TOUCHED) { s->cnt_off.push_back(o); o += segs_host[i].rows; o = (o + 3) & ~3ll; }
    else s->cnt_off.push_back(-1);
  }
  s->n_total = o;
  bool ok = hipMalloc(&s->base, sizeof(float) * (size_t)(s->n_data + 4)) == hipSuccess && hipMalloc(&s->delta, sizeof(float) * (size_t)(s->n_total + 4)) == hipSuccess &&
            hipEventCreate(&s->e0) == hipSuccess && hipEventCreate(&s->e1) == hipSuccess;
  if (ok && oh) ok = hipMalloc(&s->base16, sizeof(__half) * (size_t)oh) == hipSuccess && hipMalloc(&s->delta16, sizeof(__half) * (size_t)oh) == hipSuccess;
  // the alignment gaps between the segments are never written by the kernels but travel through the all-reduce: zero them once
  if (ok) ok = hipMemset(s->delta, 0, sizeof(float) * (size_t)(s->n_total + 4)) == hipSuccess && (!oh || hipMemset(s->delta16, 0, sizeof(__half) * (size_t)oh) == hipSuccess);
  if (!ok) {
    (void)hipGetLastError();
    if (s->base) (void)hipFree(s->base);
    if (s->delta) (void)hipFree(s->delta);
    if (s->base16) (void)hipFree(s->base16);
    if (s->delta16) (void)hipFree(s->delta16);
    delete s;
    return sfail(nullptr, POI_ENOMEM, "poi_sync_create: allocation failed");
  }
  *out = s;
  return POI_OK;
}

int poi_sync_destroy(poi_sync* s) {
  if (!s) return POI_OK;
  (void)hipSetDevice(s->device);
  (void)hipDeviceSynchronize();
  if (s->base) (void)hipFree(s->base);
  if (s->delta) (void)hipFree(s->delta);
  if (s->base16) (void)hipFree(s->base16);
  if (s->delta16) (void)hipFree(s->delta16);
  if (s->e0) (void)hipEventDestroy(s->e0);
  if (s->e1) (void)hipEventDestroy(s->e1);
  delete s;
  return POI_OK;
}

int poi_sync_begin_epoch(poi_sync* s, void* stream) {
  if (!s) return sfail(s, POI_EINVAL, "poi_sync_begin_epoch: NULL");
  if (hipSetDevice(s->device) != hipSuccess) return sfail(s, POI_EHIP, "hipSetDevice failed");
  for (size_t i = 0; i < s->segs.size(); ++i) {
    const long long n = s->segs[i].rows * s->segs[i].width;
    if (s->segs[i].dtype == POI_F16) {       // the snapshot of a half table is the table (a device copy)
      if (hipMemcpyAsync(s->base16 + s->off[i], s->segs[i].cur, sizeof(__half) * (size_t)n, hipMemcpyDeviceToDevice, (hipStream_t)stream) != hipSuccess)
        return sfail(s, POI_EHIP, "poi_sync_begin_epoch: copy failed");
      continue;
    }
    hipLaunchKernelGGL(poi::sync_copy_kernel, dim3((unsigned)((n + 255) / 256 > 4096 ? 4096 : (n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       s->segs[i].cur, s->base + s->off[i], n);
  }
  return hipGetLastError() == hipSuccess ? POI_OK : sfail(s, POI_EHIP, "poi_sync_begin_epoch: launch failed");
}

int poi_sync_make_delta(poi_sync* s, void* stream) {
  if (!s) return sfail(s, POI_EINVAL, "poi_sync_make_delta: NULL");
  if (hipSetDevice(s->device) != hipSuccess) return sfail(s, POI_EHIP, "hipSetDevice failed");
  for (size_t i = 0; i < s->segs.size(); ++i) {
    const poi_sync_seg& g = s->segs[i];
    float* cnt = s->cnt_off[i] >= 0 ? s->delta + s->cnt_off[i] : nullptr;
    if (g.dtype == POI_F16)
      hipLaunchKernelGGL(poi::sync_delta_kernel<true>, dim3(grid_rows(g.rows)), dim3(256), 0, (hipStream_t)stream, g.cur, s->base16 + s->off[i],
                         s->delta16 + s->off[i], cnt, (long long)g.rows, (long long)g.width);
    else
      hipLaunchKernelGGL(poi::sync_delta_kernel<false>, dim3(grid_rows(g.rows)), dim3(256), 0, (hipStream_t)stream, g.cur, s->base + s->off[i],
                         s->delta + s->off[i], cnt, (long long)g.rows, (long long)g.width);
  }
  return hipGetLastError() == hipSuccess ? POI_OK : sfail(s, POI_EHIP, "poi_sync_make_delta: launch failed");
}

int poi_sync_buffer(poi_sync* s, float** delta, int64_t* n) {
  if (!s || !delta || !n) return sfail(s, POI_EINVAL, "poi_sync_buffer: NULL");
  *delta = s->delta; *n = s->n_total;
  return POI_OK;
}

int poi_sync_buffer16(poi_sync* s, void** delta16_dev, int64_t* n) {
  if (!s || !delta16_dev || !n) return sfail(s, POI_EINVAL, "poi_sync_buffer16: NULL");
  *delta16_dev = s->delta16; *n = s->n_half;
  s->asked16 = true;
  return POI_OK;
}

int poi_sync_apply(poi_sync* s, int32_t world, void* stream) {
  if (!s || world < 1) return sfail(s, POI_EINVAL, "poi_sync_apply: bad argument");
  // A caller written against ABI 2 all-reduces poi_sync_buffer only: the half segments would then combine each replica's OWN delta with
  // the global touch counts and diverge silently.  (poi_sync_end_epoch reduces both buffers itself.)
  if (s->n_half > 0 && world > 1 && !s->asked16)
    return sfail(s, POI_EINVAL, "poi_sync_apply: the set has IEEE-half segments whose deltas live in poi_sync_buffer16, which was never queried - all-reduce it as well (ABI 3)");
  if (hipSetDevice(s->device) != hipSuccess) return sfail(s, POI_EHIP, "hipSetDevice failed");
  for (size_t i = 0; i < s->segs.size(); ++i) {
    const poi_sync_seg& g = s->segs[i];
    const float* cnt = s->cnt_off[i] >= 0 ? s->delta + s->cnt_off[i] : nullptr;
    if (g.dtype == POI_F16)
      hipLaunchKernelGGL(poi::sync_apply_kernel<true>, dim3(grid_rows(g.rows)), dim3(256), 0, (hipStream_t)stream, g.cur, s->base16 + s->off[i],
                         s->delta16 + s->off[i], cnt, (long long)g.rows, (long long)g.width, (int)g.rule, 1.0f / (float)world);
    else
      hipLaunchKernelGGL(poi::sync_apply_kernel<false>, dim3(grid_rows(g.rows)), dim3(256), 0, (hipStream_t)stream, g.cur, s->base + s->off[i],
                         s->delta + s->off[i], cnt, (long long)g.rows, (long long)g.width, (int)g.rule, 1.0f / (float)world);
  }
  return hipGetLastError() == hipSuccess ? POI_OK : sfail(s, POI_EHIP, "poi_sync_apply: launch failed");
}

int poi_sync_end_epoch(poi_sync* s, poi_comm* comm, void* stream) {
  if (!s || !comm) return sfail(s, POI_EINVAL, "poi_sync_end_epoch: NULL");
  int rc = poi_sync_make_delta(s, stream);
  if (rc) return rc;
  (void)hipEventRecord(s->e0, (hipStream_t)stream);
  if ((rc = poi_allreduce_tables(s->ctx, comm, s->delta, s->n_total, stream))) return rc;
  if (s->n_half) {       // the half segments' deltas: a second all-reduce on the same stream, half elements
    poi::Rccl* R = poi::rccl();
    const ncclResult_t nr = R->AllReduce(s->delta16, s->delta16, (size_t)s->n_half, ncclFloat16, ncclSum, comm->comm, (hipStream_t)stream);
    if (nr != ncclSuccess) return sfail(s, POI_EHIP, std::string("ncclAllReduce (half): ") + R->GetErrorString(nr));
  }
  (void)hipEventRecord(s->e1, (hipStream_t)stream);
  s->asked16 = true;      // (both buffers were reduced above)
  return poi_sync_apply(s, comm->world, stream);
}

int poi_sync_stats(poi_sync* s, double* allreduce_ms, int64_t* allreduce_bytes) {
  if (!s) return sfail(s, POI_EINVAL, "poi_sync_stats: NULL");
  if (allreduce_bytes) *allreduce_bytes = (int64_t)s->n_total * 4 + (int64_t)s->n_half * 2;
  if (allreduce_ms) {
    float ms = 0.f;
    *allreduce_ms = (hipEventSynchronize(s->e1) == hipSuccess && hipEventElapsedTime(&ms, s->e0, s->e1) == hipSuccess) ? ms : -1.0;
  }
  return POI_OK;
}

int poi_checksum(poi_ctx* ctx, const float* x, int64_t n, uint64_t* out_dev, void* stream) {
  (void)ctx;
  if (!x || !out_dev || n < 0) return sfail(nullptr, POI_EINVAL, "poi_checksum: bad argument");
  if (n == 0) return POI_OK;
  const long long g = (n + 255) / 256;
  hipLaunchKernelGGL(poi::checksum_kernel, dim3((unsigned)(g > 2048 ? 2048 : g)), dim3(256), 0, (hipStream_t)stream, (const unsigned*)x, (long long)n,
                     (unsigned long long*)out_dev);
  return hipGetLastError() == hipSuccess ? POI_OK : sfail(nullptr, POI_EHIP, "poi_checksum: launch failed");
}

}  // extern "C"
