// C-ABI of libpoi_hip.so (see include/poi_hip.h): context, scratch ownership, argument checks,
// kernel dispatch.  No torch types, no host allocation handed to the caller.
#include "../../include/poi_hip.h"
#include "poi_kernels.h"

#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <utility>
#include <vector>

namespace {
thread_local std::string g_err;
}

struct DevBuf {
  void* p = nullptr;
  size_t bytes = 0;
};

struct poi_ctx {
  int device = 0;
  int num_cu = 0;
  int wg_per_cu = 2;
  std::string err;
  // per-sequence engine
  DevBuf ws, slab, te_ws, hslab, zrow;
  DevBuf ex_ws, ex_slab, ex_glt, ex_gdi;      // exact (float64) engine
  int engine = 0;   // 0 auto, 1 per-sequence, 2 tile, 3 tile with streaming recurrent kernels, 4 exact (float64)
  float batch_cap = 1.0f;   // poi_ctx_set_batch_cap
  int wgrad_rounds = 2;
  int head_rounds = 3;      // workgroups per CU for te_head (POI_HEAD_ROUNDS, tuning)     // workgroups per CU for te_wgrad (POI_WGRAD_ROUNDS, tuning)
  int score_variant = -1;   // -1 auto; POI_SCORE_VARIANT=0|1 (tuning only)
  DevBuf g_lt, mult_lt, nseq_lt, g_di, mult_di, nseq_di;
  DevBuf g_wd, mult_wd, nseq_wd, ca_ws, ca_slab, ca_scr, ca2;      // CA-RNN (ca2: workspace of the outer-product path)
  int carnn_fast = 1;       // POI_CARNN_FAST=0: the per-sequence kernel with float atomics on the interval matrices (A/B)
  hipEvent_t ev_hr0 = nullptr, ev_hr1 = nullptr;      // early chunk sums of the write-back's hot rows (TeArgs.hot_early)
  int hot_early = 1;        // POI_TE_HOT_EARLY=0: in the tail, as up to round 5 (A/B)
  hipStream_t side2 = nullptr; hipEvent_t ev_h0 = nullptr, ev_h1 = nullptr, ev_h2 = nullptr, ev_h3 = nullptr;      // hybrid recurrences of mid-size launches (TeArgs.hyb)
  int hybrid = 1, hyb_min = 1150, hyb_max = 2300, hyb_force = 0;      // POI_TE_HYBRID=0 / option "hybrid"; launches of hyb_min .. hyb_max sequences (POI_TE_HYB_MIN / _MAX; measured: below ~1200 the per-sequence kernels alone are faster, above ~2600 the tiles alone - the fork / join costs ~15 us)
  hipStream_t side = nullptr; hipEvent_t ev_slots = nullptr, ev_sorted = nullptr, ev_bwd = nullptr, ev_fin = nullptr, ev_start = nullptr, ev_pack = nullptr;   // slot sort next to the GEMMs (POI_TE_SIDE=0: inline)
  DevBuf seg_s, seg_e;      // per table row [start, end) of the sorted scatter (te_scatter.hip); seg_e is all-zero between launches
  DevBuf xc;                // exact forward over the step-input POIs only: rank tables + per-step table rows (TeArgs.xcomp)
  DevBuf pmark;             // per-POI regrouping: per lt row, S row + 1 of a step-input POI of this launch (te_passign; all-zero between launches)
  int ppoi = 1;             // POI_TE_PPOI=0 disables the regrouping (A/B)
  int hot_bins = 1;         // te_psum also sums the DA rows of the most frequent distance bins (TeArgs.dhot); POI_TE_HOTBINS=0 / option "hot_bins"
  int early_bins = 1;       // distance-bin chain of the write-back starts next to te_gemm_dx on the side stream; POI_TE_EARLY_BINS=0: at the tail (A/B)
  int early_min = 1024;     // ... for launches of at least this many sequences (POI_TE_EARLY_MIN; 1300 .. 2000 users: -5 % per launch against the inline chain)
  DevBuf kc_dev;            // te_wgrad's K-chunk split, chosen on the device per launch
  DevBuf ptab, iota;        // forward table (te_rec_fwd16<FT>): lt . ui[:, :D]^T per table row; 0..n_item, n_item + 1
  int iota_n = -1;          // rows the iota buffer currently describes
  int fwd_tab = 1;          // POI_TE_FWDTAB=0 disables (A/B)
  int bintab_min = 1280;    // launches below this many sequences take the two-table path (no per-bin tables / per-POI regrouping); POI_TE_BINTAB_MIN
  int one_path = 1;         // launches of ONE sequence (Distance2Pre, plain GRU) take the five-kernel path (te_one_*); POI_TE_ONE=0 -> the batched pipeline
  int rec1_max = 1800;      // launches of at most this many sequences run the per-sequence recurrent kernels (te_rec_fwd1 / bwd1: persistent since round 5 - crossover with the 16-sequence tiles measured at ~1800 for the backward, ~1100 for the float64 forward pass); POI_TE_REC1
  int rec_split = 1;        // recurrent kernels on bf16 x 3 split operands; POI_TE_SPLIT=0 -> float32-input MFMA (A/B)
  int xlaunch = 0;          // launch id of the exact forward's non-finite-input flag (TeArgs.xflag)
  int xfwd = 1;             // exact forward (te_xfwd.hip: fixed point on the int8 matrix cores / float64 MFMA + float64 gates) for dims 64 / 128 / 256; POI_TE_XFWD=0 / poi_ctx_set_exact_forward
  int xcomp = 1;            // exact forward table over the step-input POIs only; POI_TE_XCOMP=0: every row of the POI table (A/B)
  int xcomp_min = 1536;     // ... for launches of at least this many sequences (below: one row per step - the table form of te_rec_fwdx costs 0.6 us more per step of the latency chain than the ranking saves in te_gemmx; 1300 / 1563 / 2048 / 3125 users: +9 / -6 / -38 / -45 us); POI_TE_XCOMP_MIN
  int efuse = 1;            // E = lt[p'] - lt[q'] gathered inside te_head3 (dim 128) instead of written by te_gather and read back twice; POI_TE_EFUSE
  int head3 = 1;            // training head on split products for <= 256 bins (te_head3); POI_TE_HEAD3
  int xrec1_max = 1100;     // ... launches of at most this many sequences run its recurrence per sequence in float64 on the vector ALUs (te_rec_fwd1x); POI_TE_XREC1
  DevBuf bad_ids;           // out-of-range ids seen by poi_bpr_step (poi_ctx_take_bad_ids)
  DevBuf xflag;             // launch id of the last launch whose operands held a NaN / inf (TeArgs.xflag)
  DevBuf xw, xg;            // its digit fragments, scales and per-bin table | per-step pre-activations or the forward table (float64)
  // hipGraph replay of the tile engine's training launch (poi_ctx_set_graph): ~40 kernels on two streams become one graph launch.
  // A launch is captured the second time its key (every pointer / size / scalar the kernels receive) is seen; the caller's uidx /
  // out are staged through context buffers so that the key does not depend on them.
  struct StepGraph { std::vector<uint64_t> key; hipGraph_t graph; hipGraphExec_t exec; uint64_t stamp; };
  std::vector<StepGraph> graphs;
  std::vector<uint64_t> seen_key;
  int graph_mode = 0;       // off by default (no gain measured on ROCm 7.0: DESIGN.md section 5); POI_GRAPH=1 / poi_ctx_set_graph enable
  int graph_min_n = 0, graph_max_n = 1 << 30;
  uint64_t graph_stamp = 0, graph_replays = 0, graph_captures = 0;
  hipStream_t cap = nullptr;   // capture stream (the caller's stream may be the null stream, which cannot capture)
  DevBuf uidx_stage, out_stage;
  // BPR
  DevBuf g_ux, cnt_ux, g_blt, cnt_blt;
  // scoring
  DevBuf cand_s, cand_i, items_pk, gbound;
  DevBuf items_pk16, inorm, surv_cnt, surv_idx, surv_sc, tflag, pre_idx, pre_sc;      // two-stage fused top-K (score_filter.hip)
  DevBuf users_pk16, ubound, ugeo;      // item-stationary GEO filter: users' half fragments, per-user bound terms, last-POI coordinates
  int sf_items = -1;        // poi_ctx_set_topk_filter(ctx, 2 / 3): force / forbid the item-stationary GEO filter (-1: by shape)
  int f16_rounding = 0;     // poi_ctx_set_f16_rounding: 0 nearest, 1 stochastic (write-back of a half POI table)
  unsigned sr_counter = 0;  // launches so far (salt of the stochastic rounding)
  int topk_filter = 1;      // poi_ctx_set_topk_filter / POI_TOPK_FILTER=0: one-stage float32 kernel only
  int last_two_n = 0, last_two_tiles = 0;      // users / user tiles of the last two-stage call (poi_ctx_topk_filter_stats)
  const int32_t* seed_idx = nullptr; int seed_k = 0;      // poi_ctx_set_topk_seed: consumed by the next fused top-K call
  // selftest
  DevBuf st;
  poi::Timing tm;
  std::vector<std::pair<const char*, size_t>> f16;      // registered IEEE-half table buffers (poi_ctx_register_f16)
};

static int is_f16(const poi_ctx* c, const void* p) {
  if (!c || !p) return 0;
  for (auto& r : c->f16) if ((const char*)p >= r.first && (const char*)p < r.first + r.second) return 1;
  return 0;
}

static int fail(poi_ctx* c, int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_err = buf;
  if (c) c->err = buf;
  return code;
}

#define HIPCHK(c, expr)                                                                      \
  do {                                                                                       \
    hipError_t e_ = (expr);                                                                  \
    if (e_ != hipSuccess) return fail(c, POI_EHIP, "%s: %s", #expr, hipGetErrorString(e_));  \
  } while (0)

// Grow-only zero-initialised device buffer.  Contents are all-zero after (re)allocation; users
// of gradient tables rely on that and restore the zeros themselves after each launch.
static int ensure(poi_ctx* c, DevBuf& b, size_t bytes, hipStream_t st) {
  if (bytes <= b.bytes) return POI_OK;
  if (b.p) { HIPCHK(c, hipStreamSynchronize(st)); HIPCHK(c, hipFree(b.p)); b.p = nullptr; b.bytes = 0; }
  size_t want = bytes + bytes / 8;
  if (hipMalloc(&b.p, want) != hipSuccess) {
    (void)hipGetLastError();
    want = bytes;
    if (hipMalloc(&b.p, want) != hipSuccess) { b.p = nullptr; return fail(c, POI_ENOMEM, "hipMalloc(%zu) failed", bytes); }
  }
  b.bytes = want;
  HIPCHK(c, hipMemsetAsync(b.p, 0, want, st));
  return POI_OK;
}

extern "C" {

int poi_abi_version(void) { return POI_ABI_VERSION; }

const char* poi_last_error(const poi_ctx* ctx) { return ctx ? ctx->err.c_str() : g_err.c_str(); }

int poi_ctx_create(poi_ctx** out, int device) {
  if (!out) return fail(nullptr, POI_EINVAL, "poi_ctx_create: out is NULL");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(nullptr, POI_EHIP, "no HIP device visible");
  if (device < 0 || device >= ndev) return fail(nullptr, POI_EINVAL, "device %d out of range (%d visible)", device, ndev);
  poi_ctx* c = new poi_ctx();
  c->device = device;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) != hipSuccess) { delete c; return fail(nullptr, POI_EHIP, "hipGetDeviceProperties failed"); }
  c->num_cu = prop.multiProcessorCount;
  if (const char* e = getenv("POI_SEQ_WG_PER_CU")) { int v = atoi(e); if (v >= 1 && v <= 8) c->wg_per_cu = v; }
  if (const char* e = getenv("POI_HEAD_ROUNDS")) { int v = atoi(e); if (v >= 1 && v <= 4) c->head_rounds = v; }
  if (const char* e = getenv("POI_WGRAD_ROUNDS")) { int v = atoi(e); if (v >= 1 && v <= 4) c->wgrad_rounds = v; }
  if (const char* e = getenv("POI_SCORE_VARIANT")) { int v = atoi(e); if (v >= 0 && v <= 1) c->score_variant = v; }
  if (const char* e = getenv("POI_TE_PPOI")) c->ppoi = atoi(e) != 0;
  if (const char* e = getenv("POI_TE_EARLY_BINS")) c->early_bins = atoi(e) != 0;
  if (const char* e = getenv("POI_TE_HOTBINS")) c->hot_bins = atoi(e) != 0;
  if (const char* e = getenv("POI_TE_HYBRID")) c->hybrid = atoi(e) != 0;
  if (const char* e = getenv("POI_TE_HOT_EARLY")) c->hot_early = atoi(e) != 0;
  if (const char* e = getenv("POI_TE_HYB_MIN")) c->hyb_min = atoi(e);
  if (const char* e = getenv("POI_TE_HYB_MAX")) c->hyb_max = atoi(e);
  if (const char* e = getenv("POI_TE_HYB_FORCE")) c->hyb_force = atoi(e);
  if (const char* e = getenv("POI_TE_EARLY_MIN")) c->early_min = atoi(e);
  if (const char* e = getenv("POI_TE_FWDTAB")) c->fwd_tab = atoi(e) != 0;
  if (const char* e = getenv("POI_TE_SPLIT")) c->rec_split = atoi(e) != 0;
  if (const char* e = getenv("POI_TE_XFWD")) c->xfwd = atoi(e) != 0;
  if (const char* e = getenv("POI_TE_XREC1")) c->xrec1_max = atoi(e);
  if (const char* e = getenv("POI_TE_HEAD3")) c->head3 = atoi(e) != 0;
  if (const char* e = getenv("POI_TE_EFUSE")) c->efuse = atoi(e) != 0;
  if (const char* e = getenv("POI_TE_XCOMP")) c->xcomp = atoi(e) != 0;
  if (const char* e = getenv("POI_TE_XCOMP_MIN")) c->xcomp_min = atoi(e);
  if (const char* e = getenv("POI_TE_REC1")) c->rec1_max = atoi(e);
  if (const char* e = getenv("POI_TE_ONE")) c->one_path = atoi(e) != 0;
  if (const char* e = getenv("POI_TE_BINTAB_MIN")) c->bintab_min = atoi(e);
  if (const char* e = getenv("POI_CARNN_FAST")) c->carnn_fast = atoi(e) != 0;
  if (const char* e = getenv("POI_GRAPH")) c->graph_mode = atoi(e) != 0;
  if (const char* e = getenv("POI_TOPK_FILTER")) c->topk_filter = atoi(e) != 0;
  if (const char* e = getenv("POI_ENGINE")) { if (!strcmp(e, "seq")) c->engine = 1; else if (!strcmp(e, "tile")) c->engine = 2; else if (!strcmp(e, "exact")) c->engine = 4; }
  if (hipSetDevice(device) != hipSuccess) { delete c; return fail(nullptr, POI_EHIP, "hipSetDevice failed"); }
  const char* sd = getenv("POI_TE_SIDE");
  if (!sd || atoi(sd) != 0) {
    // (a high or a low stream priority for the side stream changes nothing: 1727 - 1746 us per 12500-user launch either way)
    if (hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_slots, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_sorted, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_bwd, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_fin, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_start, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_pack, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_hr0, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&c->ev_hr1, hipEventDisableTiming) != hipSuccess) {
      (void)hipGetLastError();
      c->side = nullptr;                     // fall back to the inline sort
    }
    if (c->side && (hipStreamCreateWithFlags(&c->side2, hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&c->ev_h0, hipEventDisableTiming) != hipSuccess ||
                    hipEventCreateWithFlags(&c->ev_h1, hipEventDisableTiming) != hipSuccess ||
                    hipEventCreateWithFlags(&c->ev_h2, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&c->ev_h3, hipEventDisableTiming) != hipSuccess)) {
      (void)hipGetLastError();
      c->side2 = nullptr;                    // no hybrid recurrences
    }
  }
  if (c->graph_mode && hipStreamCreateWithFlags(&c->cap, hipStreamNonBlocking) != hipSuccess) { (void)hipGetLastError(); c->cap = nullptr; c->graph_mode = 0; }
  *out = c;
  return POI_OK;
}

static void drop_graphs(poi_ctx* c) {
  for (auto& g : c->graphs) { (void)hipGraphExecDestroy(g.exec); (void)hipGraphDestroy(g.graph); }
  c->graphs.clear(); c->seen_key.clear();
}

int poi_ctx_destroy(poi_ctx* c) {
  if (!c) return POI_OK;
  DevBuf* all[] = {&c->ex_ws, &c->ex_slab, &c->ex_glt, &c->ex_gdi, &c->ws, &c->slab, &c->te_ws, &c->hslab, &c->zrow, &c->g_lt, &c->mult_lt, &c->nseq_lt, &c->g_di, &c->mult_di, &c->nseq_di, &c->seg_s, &c->seg_e, &c->pmark, &c->xc, &c->kc_dev, &c->uidx_stage, &c->out_stage, &c->ptab, &c->iota, &c->xw, &c->xg, &c->xflag, &c->bad_ids,
                   &c->g_wd, &c->mult_wd, &c->nseq_wd, &c->ca_ws, &c->ca_slab, &c->ca_scr, &c->ca2, &c->g_ux, &c->cnt_ux, &c->g_blt, &c->cnt_blt, &c->cand_s, &c->cand_i, &c->items_pk, &c->gbound, &c->st,
                   &c->items_pk16, &c->inorm, &c->surv_cnt, &c->surv_idx, &c->surv_sc, &c->tflag, &c->pre_idx, &c->pre_sc, &c->users_pk16, &c->ubound, &c->ugeo};
  (void)hipDeviceSynchronize();
  c->tm.clear();
  drop_graphs(c);
  if (c->cap) (void)hipStreamDestroy(c->cap);
  if (c->side2) { (void)hipStreamDestroy(c->side2); (void)hipEventDestroy(c->ev_h0); (void)hipEventDestroy(c->ev_h1); (void)hipEventDestroy(c->ev_h2); (void)hipEventDestroy(c->ev_h3); }
  if (c->side) { (void)hipStreamDestroy(c->side); (void)hipEventDestroy(c->ev_slots); (void)hipEventDestroy(c->ev_sorted); (void)hipEventDestroy(c->ev_bwd); (void)hipEventDestroy(c->ev_fin); (void)hipEventDestroy(c->ev_start); (void)hipEventDestroy(c->ev_pack); (void)hipEventDestroy(c->ev_hr0); (void)hipEventDestroy(c->ev_hr1); }
  for (DevBuf* b : all) if (b->p) (void)hipFree(b->p);
  delete c;
  return POI_OK;
}

int poi_ctx_num_cu(const poi_ctx* c) { return c ? c->num_cu : POI_EINVAL; }

// ---------------------------------------------------------------------------------------------
static int check_gru(poi_ctx* c, const poi_gru_params* P, const poi_seq_tables* T, bool spatial, bool need_q) {
  if (!c || !P || !T) return fail(c, POI_EINVAL, "NULL ctx/params/tables");
  if (P->dim <= 0 || P->dim % 4 != 0 || P->dim > 256) return fail(c, POI_ENOTSUP, "dim must be a multiple of 4 in [4, 256] (got %d)", P->dim);
  if (!P->lt || !P->ui || !P->wh || !P->bi) return fail(c, POI_EINVAL, "lt/ui/wh/bi must be non-NULL");
  if (spatial && (!P->di || !P->vs || !P->bs || !P->wd || !P->lw || P->n_dist <= 0)) return fail(c, POI_EINVAL, "spatial model needs di/vs/bs/wd/lw and n_dist > 0");
  if (!T->off || !T->p) return fail(c, POI_EINVAL, "tables: off/p must be non-NULL");
  if (need_q && !T->q) return fail(c, POI_EINVAL, "tables: q must be non-NULL");
  if (spatial && (!T->dp || (need_q && !T->dq))) return fail(c, POI_EINVAL, "tables: dp/dq must be non-NULL for the spatial model");
  if (T->max_len <= 0 || T->len_max < T->max_len) return fail(c, POI_EINVAL, "tables: need 0 < max_len <= len_max");
  return POI_OK;
}

static void fill_args(poi::SeqArgs& A, const poi_gru_params* P, const poi_seq_tables* T, const int32_t* uidx, int n) {
  memset(&A, 0, sizeof A);
  A.lt = P->lt; A.di = P->di; A.ui = P->ui; A.wh = P->wh; A.bi = P->bi; A.vs = P->vs; A.bs = P->bs; A.wd = P->wd; A.lw = P->lw;
  A.n_item = P->n_item; A.n_dist = P->n_dist; A.dim = P->dim;
  A.off = T->off; A.p = T->p; A.q = T->q; A.dp = T->dp; A.dq = T->dq;
  A.len_max = T->len_max; A.cap = T->max_len;
  A.uidx = uidx; A.n_seq = n;
  A.bcap = 1.0f;
}

// Carve the tile engine's packed-row workspace out of one grow-only buffer.
static int te_setup(poi_ctx* c, poi::TeArgs& A, const poi_gru_params* P, const poi_seq_tables* T, const int32_t* uidx, int n,
                    bool predict, bool spatial, hipStream_t st) {
  memset(&A, 0, sizeof A);
  const int n_dist = spatial ? P->n_dist : -1;            // the plain GRU has no distance-bin table: zero rows
  const int D = P->dim, NBP = poi::te_nbp(n_dist);
  A.lt = P->lt; A.di = P->di; A.ui = P->ui; A.wh = P->wh; A.bi = P->bi; A.vs = P->vs; A.bs = P->bs; A.wd = P->wd; A.lw = P->lw;
  A.n_item = P->n_item; A.n_dist = n_dist; A.dim = D;
  A.spatial = spatial ? 1 : 0; A.xw = spatial ? 2 * D : D;
  A.lt_f16 = is_f16(c, P->lt);
  A.bintab = (poi::te_bintab(D, spatial, n_dist) && (predict || n >= c->bintab_min)) ? 1 : 0;
  A.rec32 = (D >= 256 || (D == 128 && c->engine == 3)) ? 1 : 0;
  A.rec_split = c->rec_split ? 1 : 0;          // (16-sequence tiles and the streaming kernels of dim 256 alike)
  A.head_split = (c->rec_split && A.spatial && (P->n_dist + 1 > 256 || (c->head3 && !predict))) ? 1 : 0;      // (<= 256 bins: te_head3, training launches only)
  A.efuse = (c->efuse && A.head_split && !predict && P->n_dist + 1 <= 256 && D == 128) ? 1 : 0;      // (the one-sequence path clears it: te_one_in writes E)
  A.rec1 = (!A.rec32 && n <= c->rec1_max) ? 1 : 0;
  // hybrid recurrences (poi_kernels.h TeArgs.hyb): training launches of hyb_min .. hyb_max sequences on the exact forward, dim 128.  (NOT dim 64: there a
  // workgroup of each kernel fits the same CU, and with the two backward kernels co-resident identical launches differed in the last bits of DA in 5 - 20 % of
  // the runs - inside every parity bar, but not bitwise reproducible; serialised, or at dim 128 where each workgroup fills its CU, 0 of 240: tools/repro_check.py)
  A.hyb = (c->hybrid && c->side2 && !predict && !A.rec32 && D == 128 && c->xfwd && poi::te_xfwd_supported(D) && c->rec_split && n >= c->hyb_min && n <= c->hyb_max && n <= TE_HYB_NMAX && n > 1) ? 1 : 0;
  A.hyb_force = c->hyb_force;
  if (A.hyb) A.rec1 = 0;                    // (the tile kernels' weight packs; the per-sequence backward kernel gets its transposes in pWhc1 / pWhzr1)
  A.ppoi = (A.bintab && !predict && c->ppoi) ? 1 : 0;
  A.off = T->off; A.p = T->p; A.q = T->q; A.dp = T->dp; A.dq = T->dq; A.len_max = T->len_max;
  A.uidx = uidx; A.n_seq = n; A.predict = predict ? 1 : 0;
  if (const char* e = getenv("POI_TE_DBG")) A.dbg = atoi(e);
  A.dl = poi::dense_layout(D, A.xw, n_dist + 1);
  const size_t Tcap = (size_t)n * (size_t)(predict ? T->max_len : (T->max_len > 1 ? T->max_len - 1 : 1)) + 192;   // + spare rows: row T and the rest of the last 128-row tile
  // uiT 6 + uiP 3 + pWhT16 4.5 + pWhc16 1.5 + pWhzr16 3 + pUiP3 4.5 (x D^2) + spare, pVsT + pVs 2 x 1.5 NBP D, + the rounding of each carve
  const size_t pk = (size_t)30 * D * D + (size_t)3 * NBP * D + 128;      // (+ 3 D^2: pWhc1 / pWhzr1 of the hybrid recurrences)
  // sorted scatter (training): 3 slots per sequence position
  const bool sorted = !predict;
  const size_t Ncap = sorted ? 3 * (Tcap + (size_t)n) : 0;
  const size_t n_hot = Ncap / (TE_COLD_MAX + 1) + 2, n_chunk = Ncap / TE_HOT_CHUNK + n_hot + 2;
  const size_t sfl = sorted ? Tcap + n_chunk * D + Tcap * (size_t)NBP + (size_t)((n <= c->rec1_max || n <= c->hyb_max) ? n : (n + 15) / 16) * 3 * D + (size_t)2 * ((n + 255) / 256) + 128 : 0;
  const int Rrows = P->n_item + 1 + n_dist + 1;
  const bool listed = sorted && (size_t)Rrows > 4 * Ncap;      // table much larger than the launch's footprint: touched-row list
  const size_t sin = sorted ? 7 * Ncap + (listed ? Ncap : 0) + (size_t)RS_HIST_INTS + RS_MAXBIN + 16 + 4 * n_hot + 3 * n_chunk + 64 : 0;
  // per-bin tables (bintab): ztab + per-bin sums + d di sums, and (training) the sliced partial sums of DA
  const size_t NBt = (size_t)(n_dist + 1), n_dchunk = (Tcap + (size_t)n) / 64 + NBt + 2 + (size_t)TE_HB * (size_t)TE_PSUM_WG;      // (+ the hot bins' per-workgroup partials: TeArgs.dhot)
  A.npw = TE_PSUM_WG;         // workgroups of te_psum: a CONSTANT (the hot bins' partial sums are formed per workgroup: their summation order must not depend on the device's CU count); 96 registers x 384 threads - three or four of them fit a CU at a time      // 64-entry chunks of the bins' entry segments
  const size_t n_dsuper = n_dchunk / 32 + NBt + 2;
  const size_t bfl = A.bintab ? NBt * (size_t)(3 * D) * 2 + NBt * D + (sorted ? (n_dchunk + n_dsuper) * (size_t)(3 * D) : 0) + 64 : 0;
  const size_t n_prange = Tcap / 64 + 4;
  const size_t pfl = A.ppoi ? Tcap * (size_t)(3 * D) + 2 * n_prange * (size_t)(3 * D) + 64 : 0;
  const size_t nfl = Tcap * (size_t)(9 * D + 2) + pk + 64 + sfl + bfl + pfl;
  const size_t nin = Tcap * 7 + (size_t)n + 48 + sin + 512 + (A.bintab && sorted ? n_dchunk + n_dsuper + 4 * NBt + 64 : 0) + 3 * NBt + 48 + (A.ppoi ? 4 * Tcap + 2 * 1024 + 128 : 0);
  int rc = ensure(c, c->te_ws, nfl * 4 + nin * 4 + 1024, st);
  if (rc) return rc;
  const int R = P->n_item + 1 + n_dist + 1;
  if (sorted && ((rc = ensure(c, c->seg_s, sizeof(int) * (size_t)(R + 1), st)) || (rc = ensure(c, c->seg_e, sizeof(int) * (size_t)(R + 1), st)))) return rc;
  if (A.ppoi && (rc = ensure(c, c->pmark, sizeof(int) * (size_t)(P->n_item + 2), st))) return rc;
  // forward table: worth it when the table has clearly fewer rows than the launch has steps (Tcap is the upper bound: sequences
  // average ~40 % of the longest) - te_gemm_ax then multiplies n_item + 1 rows instead of one row per step
  // exact forward (dims 64 / 128): input product and forward recurrence in fixed point / float64 (te_xfwd.hip)
  // dim 256 (config X): the streaming backward kernels keep their 32-sequence tiles, the forward pass runs te_gemmx<256> + te_rec_fwdd (float64 MFMA)
  A.xfwd = (c->xfwd && (!A.rec32 || D == 256) && poi::te_xfwd_supported(D)) ? 1 : 0;      // (training launches and predict alike)
  const bool want_ft = c->fwd_tab && !A.rec32 && 2 * (size_t)(P->n_item + 1) <= Tcap;
  A.fwd_tab = (!A.xfwd && want_ft && A.bintab && !A.rec1) ? 1 : 0;      // (the per-sequence kernels read G)
  A.xrec1 = (A.xfwd && D <= 128 && n <= c->xrec1_max && !A.hyb) ? 1 : 0;      // one workgroup per sequence, float64 on the vector ALUs (te_rec_fwd1x)
  // (the table over the launch's step-input POIs only - te_slots marks them for the per-POI regrouping - never has more rows than the launch has steps)
  const bool want_xc = c->xcomp && A.ppoi && P->n_item + 1 <= (1 << 22) && n >= c->xcomp_min;
  A.xft = (A.xfwd && (want_ft || want_xc) && n > 1) ? 1 : 0;      // (n == 1: the one-sequence path writes gx itself, te_one_in; the per-sequence kernel has a table variant too)
  if (A.fwd_tab || A.xft) {
    if ((rc = ensure(c, c->iota, sizeof(int) * (size_t)(P->n_item + 8), st))) return rc;
    if (c->iota_n != P->n_item + 1) { poi::launch_te_iota((int*)c->iota.p, P->n_item + 1, st); c->iota_n = P->n_item + 1; }
    A.iota = (const int*)c->iota.p;
  }
  if (A.fwd_tab) {
    // (+ spare rows: te_gemm_ntk writes whole 128-row tiles - rows past the table's last land behind it, as in G)
    if ((rc = ensure(c, c->ptab, sizeof(float) * ((size_t)(P->n_item + 1 + 127) / 128 * 128 + 128) * 3 * D, st))) return rc;
    A.ptab = (float*)c->ptab.p;
  }
  if (A.xfwd) {
    const size_t frag = (size_t)3 * D * D * 5, nz = (size_t)(spatial ? n_dist + 1 : 1) * 3 * D;      // five int8 digit planes per weight
    if ((rc = ensure(c, c->xw, 2 * frag + 64 + sizeof(double) * (6 * (size_t)D + nz + 8) + 64, st))) return rc;
    A.xcomp = (A.xft && want_xc) ? 1 : 0;
    const size_t xrows = A.xcomp ? std::min((size_t)P->n_item + 1, Tcap) + 2 : A.xft ? (size_t)P->n_item + 1 + 2 : Tcap;
    if ((rc = ensure(c, c->xg, sizeof(double) * xrows * 3 * D, st))) return rc;
    char* xp = (char*)c->xw.p;
    A.xWh8 = (uint4*)xp; A.xUi8 = (uint4*)(xp + ((frag + 15) & ~(size_t)15));
    double* xd = (double*)(xp + 2 * ((frag + 15) & ~(size_t)15));
    A.xWhS = xd; A.xUiS = xd + 3 * D; A.ztabx = xd + 6 * D;
    // (its own allocation: the address must not move with the dim / bin count of the launch - whatever an earlier launch left at a recycled
    // offset would be read as a launch id; zero-filled at allocation, ids start at 1)
    if ((rc = ensure(c, c->xflag, 64, st))) return rc;
    A.xflag = (int*)c->xflag.p; A.xlaunch = ++c->xlaunch;
    if (A.xft) A.ptabx = (double*)c->xg.p; else A.gx = (double*)c->xg.p;
    A.x_rows_est = (int)(Tcap < (size_t)1 << 30 ? Tcap : (size_t)1 << 30);
    if (A.xcomp) {
      const size_t nr = ((size_t)P->n_item + 2 + 3) & ~(size_t)3;
      if ((rc = ensure(c, c->xc, sizeof(int) * (2 * nr + 256 + 8 + Tcap), st))) return rc;
      int* xi = (int*)c->xc.p;
      A.xidx = xi; A.xlist = xi + nr; A.xblk = xi + 2 * nr; A.xcnt = xi + 2 * nr + 256; A.row_pc = xi + 2 * nr + 256 + 8;
    }
  }
  float* f = (float*)c->te_ws.p;
  auto take = [&](size_t cnt) { float* r = f; f += (cnt + 3) & ~(size_t)3; return r; };
  A.X = take(Tcap * 2 * D); A.E = take(Tcap * D); A.G = take(Tcap * 3 * D); A.H = take(Tcap * D);
  A.RH = take(Tcap * D); A.DH = take(Tcap * D); A.rowloss = take(Tcap * 2);
  A.uiT = take((size_t)6 * D * D); A.uiP = take((size_t)3 * D * D);
  A.pUiP3 = (float4*)take((size_t)9 * D * D / 2);
  A.pVsT = (float4*)take((size_t)NBP * D * 3 / 2); A.pVs = (float4*)take((size_t)NBP * D * 3 / 2);      // (x 1.5: te_head_big3 reads bf16 x 3 planes)
  // (x 1.5: the split-operand recurrent kernels keep every weight as three bf16 planes)
  A.pWhT16 = (float4*)take((size_t)9 * D * D / 2); A.pWhc16 = (float4*)take((size_t)3 * D * D / 2); A.pWhzr16 = (float4*)take((size_t)3 * D * D);
  if (A.hyb) { A.pWhc1 = (float4*)take((size_t)D * D); A.pWhzr1 = (float4*)take((size_t)2 * D * D); } else { A.pWhc1 = A.pWhc16; A.pWhzr1 = A.pWhzr16; }
  if (A.bintab) {
    A.ztab = take(NBt * 3 * D); A.dsum = take(NBt * 3 * D); A.dgd = take(NBt * D);
    if (sorted) { A.dpart = take(n_dchunk * (size_t)(3 * D)); A.dpart2 = take(n_dsuper * (size_t)(3 * D)); }
  }
  if (A.ppoi) { A.S = take(Tcap * (size_t)(3 * D)); A.pfirst = take(n_prange * (size_t)(3 * D)); A.plast = take(n_prange * (size_t)(3 * D)); }
  if (sorted) {
    A.gcoef = take(Tcap); A.hot_part = take(n_chunk * D); A.DL = take(Tcap * (size_t)NBP);
    A.bi_part = take((size_t)((A.rec1 || A.hyb) ? n : (n + 15) / 16) * 3 * D); A.fin_part = take((size_t)2 * ((n + 255) / 256) + 8);
  }
  int* ip = (int*)f;
  auto itake = [&](size_t cnt) { int* r = ip; ip += (cnt + 3) & ~(size_t)3; return r; };
  A.hyb_dev = itake(8);
  A.soff = itake(n + 1); A.row_src = itake(Tcap); A.row_t = itake(Tcap); A.row_p = itake(Tcap); A.row_dp = itake(Tcap); A.row_ab = itake(Tcap); A.row_pq = (int2*)itake(2 * Tcap);
  if (sorted) {
    int bits = 1; while ((1 << bits) <= R) ++bits;      // keys 0..R (R = sentinel)
    A.key_bits = bits;
    A.keys0 = itake(Ncap); A.keys1 = itake(Ncap); A.vals0 = itake(Ncap); A.vals1 = itake(Ncap);
    A.code = itake(Ncap); A.slot_seq = itake(Ncap); A.ent = itake(Ncap);
    A.hist = itake(RS_HIST_INTS + RS_MAXBIN);   // + per-digit totals
    A.cnt = itake(8);
    A.urow = listed ? itake(Ncap) : nullptr;
    A.hot_rows = (int4*)itake(4 * n_hot); A.hot_chunks = (int2*)itake(2 * n_chunk); A.hot_nf = itake(n_chunk);
    A.seg_start = (int*)c->seg_s.p; A.seg_end = (int*)c->seg_e.p;
    A.dch0 = itake(NBt + 8); A.dhot = itake(NBt + 16); A.dcc0 = itake(NBt + 8);
    if (A.ppoi) {
      A.urow_p = itake(Tcap); A.pblk = itake(2 * 1024); A.dxe = itake(Tcap); A.dxs = itake(Tcap); A.dstart = itake(Tcap + 4);
      A.pmark = (int*)c->pmark.p;
    }
    if (A.bintab) { A.dch1 = itake(NBt + 8); A.dnf = itake(n_dchunk); A.dnf2 = itake(n_dsuper); A.dbn = itake(NBt + 4); }
  }
  return POI_OK;
}

// auto: the tile engine whenever it supports the shape - measured on MI355X (Gowalla shape, D = 128) it is 4.9x faster than the
// per-sequence engine already at ONE sequence per launch (2.6 k vs 0.53 k steps/s: resident recurrent weights + MFMA against
// GEMVs streamed from L2 by one workgroup), 7x at 16 and 64.  The per-sequence engine serves the other dims / bin counts.
static bool use_tile(const poi_ctx* c, const poi_gru_params* P, bool spatial, int n) {
  (void)n;
  if (c->engine == 1 || c->engine == 4 || !poi::te_supported(P->dim, spatial ? P->n_dist : -1)) return false;
  return true;
}

// The caller's float32 alpha / lambda are the nearest floats to short decimals (0.01, 0.001: prog_bpr_gru_spatial.py:66-67); the
// float64 engine takes the SHORTEST decimal that rounds to the given float - 0.01, not 0.00999999977648 - so that its step equals
// the float64 reference's to rounding, not to 2e-8.
static double shortest_decimal(float x) {
  char buf[64];
  for (int prec = 1; prec <= 9; ++prec) {
    snprintf(buf, sizeof buf, "%.*g", prec, (double)x);
    if (strtof(buf, nullptr) == x) return strtod(buf, nullptr);
  }
  return (double)x;
}

static void fill_ex(poi::ExArgs& A, const poi_gru_params* P, const poi_seq_tables* T, const int32_t* uidx, int n) {
  memset(&A, 0, sizeof A);
  A.lt = P->lt; A.di = P->di; A.ui = P->ui; A.wh = P->wh; A.bi = P->bi; A.vs = P->vs; A.bs = P->bs; A.wd = P->wd; A.lw = P->lw;
  A.n_item = P->n_item; A.n_dist = P->n_dist; A.dim = P->dim;
  A.off = T->off; A.p = T->p; A.q = T->q; A.dp = T->dp; A.dq = T->dq;
  A.len_max = T->len_max; A.cap = T->max_len;
  A.uidx = uidx; A.n_seq = n; A.bcap = 1.0f;
}

// engine 4: float64 arithmetic end to end (exact_engine.hip)
static int exact_step(poi_ctx* c, const poi_gru_params* P, const poi_seq_tables* T, const int32_t* uidx, int32_t n,
                      float alpha, float lambda, float* out, hipStream_t st, bool spatial) {
  if (is_f16(c, P->lt)) return fail(c, POI_ENOTSUP, "the exact engine supports float32 tables only");
  const int D = P->dim, XW = spatial ? 2 * D : D, NB = spatial ? P->n_dist + 1 : 0;
  if (NB > 4096) return fail(c, POI_ENOTSUP, "the exact engine supports at most 4095 distance bins");
  int grid = c->num_cu * c->wg_per_cu;
  if (grid > n) grid = n;
  const poi::DenseLayout dl = poi::dense_layout(D, XW, NB);
  const size_t wsd = poi::ex_ws_doubles(D, NB, T->max_len);
  int rc;
  if ((rc = ensure(c, c->ex_ws, sizeof(double) * wsd * grid, st))) return rc;
  if ((rc = ensure(c, c->ex_slab, sizeof(double) * (size_t)dl.total * grid, st))) return rc;
  if ((rc = ensure(c, c->ex_glt, sizeof(double) * (size_t)(P->n_item + 1) * D, st))) return rc;
  if ((rc = ensure(c, c->mult_lt, sizeof(int) * (size_t)(P->n_item + 1), st))) return rc;
  if ((rc = ensure(c, c->nseq_lt, sizeof(int) * (size_t)(P->n_item + 1), st))) return rc;
  if (spatial) {
    if ((rc = ensure(c, c->ex_gdi, sizeof(double) * (size_t)(P->n_dist + 1) * D, st))) return rc;
    if ((rc = ensure(c, c->mult_di, sizeof(int) * (size_t)(P->n_dist + 1), st))) return rc;
    if ((rc = ensure(c, c->nseq_di, sizeof(int) * (size_t)(P->n_dist + 1), st))) return rc;
  }
  poi::ExArgs A;
  fill_ex(A, P, T, uidx, n);
  A.out = out; A.bcap = c->batch_cap == 0.0f ? -(float)n : c->batch_cap;
  A.ws = (double*)c->ex_ws.p; A.ws_stride = wsd; A.slab = (double*)c->ex_slab.p;
  A.g_lt = (double*)c->ex_glt.p; A.g_di = (double*)c->ex_gdi.p;
  A.mult_lt = (int*)c->mult_lt.p; A.nseq_lt = (int*)c->nseq_lt.p; A.mult_di = (int*)c->mult_di.p; A.nseq_di = (int*)c->nseq_di.p;
  HIPCHK(c, poi::launch_ex_train(A, spatial, grid, shortest_decimal(alpha), shortest_decimal(lambda), st, &c->tm));
  return POI_OK;
}

static int seq_step(poi_ctx* c, const poi_gru_params* P, const poi_seq_tables* T, const int32_t* uidx, int32_t n,
                    float alpha, float lambda, float* out, void* stream, bool spatial) {
  int rc = check_gru(c, P, T, spatial, true);
  if (rc) return rc;
  if (!uidx || !out || n < 0) return fail(c, POI_EINVAL, "uidx/out NULL or n < 0");
  if (n == 0) return POI_OK;
  hipStream_t st = (hipStream_t)stream;
  HIPCHK(c, hipSetDevice(c->device));
  c->tm.tick();
  if (c->engine == 4) return exact_step(c, P, T, uidx, n, alpha, lambda, out, st, spatial);
  const int D = P->dim, XW = spatial ? 2 * D : D, NB = spatial ? P->n_dist + 1 : 0;
  int grid = c->num_cu * c->wg_per_cu;
  if (grid > n) grid = n;
  const poi::DenseLayout dl = poi::dense_layout(D, XW, NB);
  const size_t wsf = poi::seq_ws_floats(D, NB, T->max_len);
  const bool tile = use_tile(c, P, spatial, n);
  if (!tile && is_f16(c, P->lt)) return fail(c, POI_ENOTSUP, "a half POI table needs the tile engine (dim 64 / 128 / 256, <= 2048 bins)");
  int n_head = 0, n_kc = 0, n_kc_ui = 0, n_slab = grid;
  if (tile) {
    n_head = c->num_cu * (D >= 256 && c->head_rounds > 2 ? 2 : c->head_rounds);      // D = 256: 62 KB of LDS per te_head workgroup, two per CU
    // te_wgrad launches (output-tile jobs) x n_kc K-chunks: fill the CUs exactly (no ragged second round)
    const bool bt = poi::te_bintab(D, spatial, spatial ? P->n_dist : -1) && n >= c->bintab_min;
    const int jobs = poi::te_wgrad_jobs(D, spatial ? P->n_dist : -1, spatial, bt), nui = poi::te_wgrad_ui_jobs(D, spatial ? P->n_dist : -1, spatial, bt);
    n_kc = (c->num_cu * c->wgrad_rounds) / jobs;
    if (n_kc < 1) n_kc = 1;
    n_kc_ui = n_kc;
    if (bt && c->ppoi) {
      // per-POI regrouping: te_wgrad picks the K-chunk split on the device from the launch's own S-row count; the slabs are sized for
      // the extremes (no S rows: every slot goes to the T-row jobs; as many S rows as steps: the uniform split)
      int a, b;
      poi::te_wgrad_split(c->num_cu * c->wgrad_rounds, nui, jobs, 1, 1 << 20, &a, &b);
      n_kc_ui = n_kc;
      n_kc = a > n_kc ? a : n_kc;
      if ((rc = ensure(c, c->kc_dev, 64, st))) return rc;
    }
    n_slab = n_kc > n_kc_ui ? n_kc : n_kc_ui;
    if ((rc = ensure(c, c->hslab, sizeof(float) * (size_t)n_head * ((NB + 4) & ~3), st))) return rc;
  } else if ((rc = ensure(c, c->ws, sizeof(float) * wsf * grid, st))) return rc;
  if ((rc = ensure(c, c->slab, sizeof(float) * (size_t)dl.total * n_slab, st))) return rc;
  if (!tile && (rc = ensure(c, c->g_lt, sizeof(float) * (size_t)(P->n_item + 1) * D, st))) return rc;      // (10 GB at 10 M POIs x 256)
  if ((rc = ensure(c, c->mult_lt, sizeof(int) * (size_t)(P->n_item + 1), st))) return rc;
  if ((rc = ensure(c, c->nseq_lt, sizeof(int) * (size_t)(P->n_item + 1), st))) return rc;
  if (spatial) {
    if (!tile && (rc = ensure(c, c->g_di, sizeof(float) * (size_t)(P->n_dist + 1) * D, st))) return rc;
    if ((rc = ensure(c, c->mult_di, sizeof(int) * (size_t)(P->n_dist + 1), st))) return rc;
    if ((rc = ensure(c, c->nseq_di, sizeof(int) * (size_t)(P->n_dist + 1), st))) return rc;
  }
  poi::SeqArgs A;
  fill_args(A, P, T, uidx, n);
  const float bcap = c->batch_cap == 0.0f ? -(float)n : c->batch_cap;      // 0: mini-batch rule, encoded as -n (rule_scales)
  A.out = out; A.bcap = bcap;
  A.ws = (float*)c->ws.p; A.ws_stride = wsf;
  A.slab = (float*)c->slab.p;
  A.g_lt = (float*)c->g_lt.p; A.mult_lt = (int*)c->mult_lt.p; A.nseq_lt = (int*)c->nseq_lt.p;
  A.g_di = (float*)c->g_di.p; A.mult_di = (int*)c->mult_di.p; A.nseq_di = (int*)c->nseq_di.p;
  if (tile) {
    poi::TeArgs E;
    if ((rc = te_setup(c, E, P, T, uidx, n, false, spatial, st))) return rc;
    if ((rc = ensure(c, c->zrow, sizeof(float) * 1024, st))) return rc;
    E.zrow = (const float*)c->zrow.p;
    E.sr_salt = (c->f16_rounding && E.lt_f16) ? (++c->sr_counter) * 0x9E3779B1u | 1u : 0u;
    E.out = out; E.bcap = bcap; E.slab = A.slab; E.n_slab = n_slab; E.n_head = n_head; E.n_kc = n_kc; E.wg_slots = c->num_cu * c->wgrad_rounds;
    E.kc_dev = (E.bintab && c->ppoi) ? (int*)c->kc_dev.p : nullptr;
    E.hslab = (float*)c->hslab.p; E.hstride = (NB + 4) & ~3;
    E.side2 = c->side2; E.ev_h0 = c->ev_h0; E.ev_h1 = c->ev_h1; E.ev_h2 = c->ev_h2; E.ev_h3 = c->ev_h3;
    E.side = c->side; E.ev_slots = c->ev_slots; E.ev_sorted = c->ev_sorted; E.ev_bwd = c->ev_bwd; E.ev_fin = c->ev_fin; E.ev_start = c->ev_start; E.ev_pack = c->ev_pack;
    E.early_bins = (c->early_bins && E.bintab && c->side && n >= c->early_min) ? 1 : 0; E.bin_alpha = alpha; E.bin_lambda = lambda;
    E.dhot_on = (E.early_bins && E.ppoi && c->hot_bins) ? 1 : 0;
    E.hot_early = (c->hot_early && E.side && E.ppoi) ? 1 : 0; E.ev_hr0 = c->ev_hr0; E.ev_hr1 = c->ev_hr1;      // (ppoi: the hot rows' entries carry no per-entry dx rows - nothing te_gemm_dx produces is read)      // (te_dprep - the selection - must have run before te_psum: the early chain)
    E.mult_lt = A.mult_lt; E.nseq_lt = A.nseq_lt; E.mult_di = A.mult_di; E.nseq_di = A.nseq_di;
    A.kc_dev = E.kc_dev;                 // (dense_apply reads te_wgrad's K-chunk counts from the device)
    // one sequence (the reference schedule): the whole step in five kernels (tile_engine.hip, te_one_*)
    const bool one = n == 1 && c->one_path && E.rec1 && !E.lt_f16 && poi::te_one_supported(D, spatial, T->max_len);
    // captured: a replay re-issues the launch id this launch got when it was captured - atomicMax against whatever a LATER launch left in xflag
    // would keep the newer id and lose this replay's NaN flag.  The captured sequence therefore starts by clearing the flag: only
    // its own pack kernels can raise it to its own id (they run behind ev_start, i.e. behind this node).
    bool captured = false;
    auto run = [&](hipStream_t s) -> hipError_t {
      if (captured && E.xflag) { hipError_t me = hipMemsetAsync(E.xflag, 0, sizeof(int), s); if (me != hipSuccess) return me; }
      if (one) { E.efuse = 0; return poi::launch_te_one(E, alpha, lambda, T->max_len, s, &c->tm); }
      hipError_t e = poi::launch_te_train(E, c->num_cu, s, &c->tm);
      // early distance-bin chain (launch_te_train started it on the side stream behind te_wgrad): the dense write-back needs te_wgrad's slabs,
      // te_finalize's parts and te_dui's - all of them older on the side stream - and nothing the POI rows' reduction on `s` reads: it
      // follows the chain there, and launch_te_scatter's join (ev_slots) waits for both
      const bool dense_side = E.early_bins && E.bintab && E.side;
      if (e == hipSuccess && dense_side) {
        e = poi::launch_dense_apply(A, spatial, n_slab, n_slab, alpha, lambda, E.side, &c->tm);
        if (e == hipSuccess) e = hipEventRecord(E.ev_slots, E.side);
      }
      if (e == hipSuccess) e = poi::launch_te_scatter(E, alpha, lambda, c->num_cu, s, &c->tm);
      if (e == hipSuccess && !dense_side) e = poi::launch_dense_apply(A, spatial, n_slab, n_slab, alpha, lambda, s, &c->tm);
      return e;
    };
    const size_t out_bytes = sizeof(float) * (size_t)n * (spatial ? 5 : 1);
    // (timing: only the SAMPLED launches carry event pairs - the others replay; a sampled launch runs eagerly)
    if (!c->graph_mode || (c->tm.on && c->tm.active) || !c->side || n < c->graph_min_n || n > c->graph_max_n || E.sr_salt) { HIPCHK(c, run(st)); return POI_OK; }
    // ---- graph replay ----
    if ((rc = ensure(c, c->uidx_stage, sizeof(int32_t) * (size_t)n, st)) || (rc = ensure(c, c->out_stage, out_bytes, st))) return rc;
    std::vector<uint64_t> key;
    {
      auto add = [&](const void* p, size_t bytes) { const size_t w = (bytes + 7) / 8, o = key.size(); key.resize(o + w, 0); memcpy(key.data() + o, p, bytes); };
      add(P, sizeof *P); add(T, sizeof *T);
      const uint64_t sc[] = {(uint64_t)n, (uint64_t)spatial, (uint64_t)c->engine, (uint64_t)c->ppoi, (uint64_t)n_head, (uint64_t)n_kc, (uint64_t)n_slab,
                             (uint64_t)c->wgrad_rounds, (uint64_t)is_f16(c, P->lt)};
      (void)bcap;
      add(sc, sizeof sc);
      const float fs[] = {alpha, lambda, c->batch_cap, 0.f};
      add(fs, sizeof fs);
      const void* bufs[] = {c->te_ws.p, c->slab.p, c->hslab.p, c->zrow.p, c->seg_s.p, c->seg_e.p, c->pmark.p, c->xc.p, c->kc_dev.p, c->mult_lt.p, c->nseq_lt.p,
                            c->mult_di.p, c->nseq_di.p, c->uidx_stage.p, c->out_stage.p, c->ptab.p, c->iota.p, c->xw.p, c->xg.p, c->xflag.p};
      add(bufs, sizeof bufs);
      // every switch that decides which kernels / which stream topology the launch takes, each in its own word (packed into one
      // word they overlapped and early_min was missing), plus what te_setup derived from them for THIS launch
      const uint64_t sw[] = {(uint64_t)c->fwd_tab, (uint64_t)c->rec_split, (uint64_t)c->one_path, (uint64_t)(unsigned)c->rec1_max, (uint64_t)(unsigned)c->bintab_min,
                             (uint64_t)c->early_bins, (uint64_t)(unsigned)c->early_min, (uint64_t)c->xfwd, (uint64_t)E.early_bins, (uint64_t)E.bintab, (uint64_t)E.rec1,
                             (uint64_t)E.fwd_tab, (uint64_t)E.xfwd, (uint64_t)E.xft, (uint64_t)E.xrec1, (uint64_t)E.xcomp, (uint64_t)(unsigned)c->xcomp_min, (uint64_t)E.head_split, (uint64_t)(unsigned)c->xrec1_max, (uint64_t)E.ppoi, (uint64_t)one, (uint64_t)E.hyb, (uint64_t)E.efuse, (uint64_t)E.hot_early};
      add(sw, sizeof sw);
    }
    poi_ctx::StepGraph* g = nullptr;
    for (auto& e : c->graphs) if (e.key == key) { g = &e; break; }
    if (!g && c->seen_key != key) {       // first sight: run eagerly (first-use initialisation happens outside any capture)
      c->seen_key = key;
      HIPCHK(c, run(st));
      return POI_OK;
    }
    HIPCHK(c, hipMemcpyAsync(c->uidx_stage.p, uidx, sizeof(int32_t) * (size_t)n, hipMemcpyDeviceToDevice, st));
    if (!g) {
      E.uidx = (const int32_t*)c->uidx_stage.p; E.out = (float*)c->out_stage.p;
      A.uidx = E.uidx; A.out = E.out;
      hipGraph_t graph = nullptr; hipGraphExec_t exec = nullptr;
      hipError_t e = hipStreamBeginCapture(c->cap, hipStreamCaptureModeRelaxed);
      if (e == hipSuccess) {
        captured = true;
        e = run(c->cap);
        captured = false;
        const hipError_t e2 = hipStreamEndCapture(c->cap, &graph);
        if (e == hipSuccess) e = e2;
      }
      if (e == hipSuccess) e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
      if (e != hipSuccess) {
        (void)hipGetLastError();
        if (graph) (void)hipGraphDestroy(graph);
        c->graph_mode = 0;                 // this runtime cannot capture the launch: stay eager
        E.uidx = uidx; E.out = out; A.uidx = uidx; A.out = out;
        HIPCHK(c, run(st));
        return POI_OK;
      }
      if (c->graphs.size() >= 8) {         // evict the least recently used
        size_t lru = 0;
        for (size_t i = 1; i < c->graphs.size(); ++i) if (c->graphs[i].stamp < c->graphs[lru].stamp) lru = i;
        HIPCHK(c, hipStreamSynchronize(st));
        (void)hipGraphExecDestroy(c->graphs[lru].exec); (void)hipGraphDestroy(c->graphs[lru].graph);
        c->graphs.erase(c->graphs.begin() + lru);
      }
      c->graphs.push_back(poi_ctx::StepGraph{key, graph, exec, 0});
      g = &c->graphs.back();
      ++c->graph_captures;
    }
    g->stamp = ++c->graph_stamp;
    HIPCHK(c, hipGraphLaunch(g->exec, st));
    ++c->graph_replays;
    HIPCHK(c, hipMemcpyAsync(out, c->out_stage.p, out_bytes, hipMemcpyDeviceToDevice, st));
    return POI_OK;
  }
  HIPCHK(c, poi::launch_seq_train(A, spatial, grid, alpha, lambda, st, &c->tm));
  return POI_OK;
}

int poi_spatial_step(poi_ctx* c, const poi_gru_params* P, const poi_seq_tables* T, const int32_t* uidx, int32_t n,
                     float alpha, float lambda, float* out, void* stream) {
  return seq_step(c, P, T, uidx, n, alpha, lambda, out, stream, true);
}

int poi_gru_step(poi_ctx* c, const poi_gru_params* P, const poi_seq_tables* T, const int32_t* uidx, int32_t n,
                 float alpha, float lambda, float* out, void* stream) {
  return seq_step(c, P, T, uidx, n, alpha, lambda, out, stream, false);
}

int poi_gru_predict(poi_ctx* c, const poi_gru_params* P, const poi_seq_tables* T, const int32_t* uidx, const int32_t* out_row, int32_t n,
                    float* hts, float* sts, void* stream) {
  const bool spatial = P && P->di != nullptr;
  int rc = check_gru(c, P, T, spatial, false);
  if (rc) return rc;
  if (!uidx || !hts || n < 0) return fail(c, POI_EINVAL, "uidx/hts NULL or n < 0");
  if (n == 0) return POI_OK;
  HIPCHK(c, hipSetDevice(c->device));
  if (use_tile(c, P, spatial, n)) {
    poi::TeArgs E;
    if ((rc = te_setup(c, E, P, T, uidx, n, true, spatial, (hipStream_t)stream))) return rc;
    E.hts = hts; E.sts = sts; E.out_row = out_row;
    HIPCHK(c, poi::launch_te_predict(E, c->num_cu, (hipStream_t)stream, &c->tm));
    return POI_OK;
  }
  if (is_f16(c, P->lt)) return fail(c, POI_ENOTSUP, "a half POI table needs the tile engine (dim 64 / 128 / 256, <= 2048 bins)");
  int grid = c->num_cu * c->wg_per_cu;
  if (grid > n) grid = n;
  if (c->engine == 4) {
    if (spatial && P->n_dist + 1 > 4096) return fail(c, POI_ENOTSUP, "the exact engine supports at most 4095 distance bins");
    poi::ExArgs X;
    fill_ex(X, P, T, uidx, n);
    X.hts = hts; X.sts = sts; X.out_row = out_row;
    HIPCHK(c, poi::launch_ex_predict(X, spatial, grid, (hipStream_t)stream, &c->tm));
    return POI_OK;
  }
  poi::SeqArgs A;
  fill_args(A, P, T, uidx, n);
  A.hts = hts; A.sts = sts; A.out_row = out_row;
  HIPCHK(c, poi::launch_seq_predict(A, spatial, grid, (hipStream_t)stream, &c->tm));
  return POI_OK;
}

// ---------------------------------------------------------------------------------------------
static int check_carnn(poi_ctx* c, const poi_carnn_params* P, const poi_seq_tables* T, bool need_q) {
  if (!c || !P || !T) return fail(c, POI_EINVAL, "NULL ctx/params/tables");
  if (P->dim <= 0 || P->dim % 4 != 0 || P->dim > 256) return fail(c, POI_ENOTSUP, "dim must be a multiple of 4 in [4, 256] (got %d)", P->dim);
  if (!P->lt || !P->wd || !P->M || P->n_dist <= 0 || P->n_item <= 0) return fail(c, POI_EINVAL, "CA-RNN needs lt / wd / M, n_item > 0 and n_dist > 0");
  if (!T->off || !T->p || !T->dp) return fail(c, POI_EINVAL, "tables: off/p/dp must be non-NULL");
  if (need_q && (!T->q || !T->dq)) return fail(c, POI_EINVAL, "tables: q/dq must be non-NULL");
  if (T->max_len <= 0 || T->len_max < T->max_len) return fail(c, POI_EINVAL, "tables: need 0 < max_len <= len_max");
  return POI_OK;
}

static void fill_carnn(poi::CaArgs& A, const poi_carnn_params* P, const poi_seq_tables* T, const int32_t* uidx, int n) {
  memset(&A, 0, sizeof A);
  A.lt = P->lt; A.wd = P->wd; A.M = P->M; A.n_item = P->n_item; A.n_dist = P->n_dist; A.dim = P->dim;
  A.off = T->off; A.p = T->p; A.q = T->q; A.dp = T->dp; A.dq = T->dq; A.len_max = T->len_max; A.cap = T->max_len;
  A.uidx = uidx; A.n_seq = n; A.bcap = 1.0f;
}

int poi_carnn_step(poi_ctx* c, const poi_carnn_params* P, const poi_seq_tables* T, const int32_t* uidx, int32_t n,
                   float alpha, float lambda, float* out, void* stream) {
  int rc = check_carnn(c, P, T, true);
  if (rc) return rc;
  if (is_f16(c, P->lt)) return fail(c, POI_ENOTSUP, "CA-RNN supports float32 tables only");
  if (c->batch_cap == 0.0f) return fail(c, POI_ENOTSUP, "the mini-batch rule (batch cap 0) applies to poi_gru_step / poi_spatial_step only");
  if (!uidx || !out || n < 0) return fail(c, POI_EINVAL, "uidx/out NULL or n < 0");
  if (n == 0) return POI_OK;
  hipStream_t st = (hipStream_t)stream;
  HIPCHK(c, hipSetDevice(c->device));
  const int D = P->dim;
  const bool fast = c->carnn_fast && (D == 64 || D == 128) && P->n_dist + 2 <= 2048;
  int grid = c->num_cu * (fast ? (getenv("POI_SEQ_WG_PER_CU") ? c->wg_per_cu : 8) : c->wg_per_cu);
  if (grid > n) grid = n;
  const size_t wsf = poi::carnn_ws_floats(D, T->max_len);
  if ((rc = ensure(c, c->ca_ws, sizeof(float) * wsf * grid, st))) return rc;
  if ((rc = ensure(c, c->ca_slab, sizeof(float) * (size_t)D * D * grid, st))) return rc;
  if ((rc = ensure(c, c->g_lt, sizeof(float) * (size_t)(P->n_item + 1) * D, st))) return rc;
  if ((rc = ensure(c, c->mult_lt, sizeof(int) * (size_t)(P->n_item + 1), st))) return rc;
  if ((rc = ensure(c, c->nseq_lt, sizeof(int) * (size_t)(P->n_item + 1), st))) return rc;
  if ((rc = ensure(c, c->g_wd, sizeof(float) * (size_t)(P->n_dist + 1) * D * D, st))) return rc;
  if ((rc = ensure(c, c->mult_wd, sizeof(int) * (size_t)(P->n_dist + 1), st))) return rc;
  if ((rc = ensure(c, c->nseq_wd, sizeof(int) * (size_t)(P->n_dist + 1), st))) return rc;
  poi::CaArgs A;
  fill_carnn(A, P, T, uidx, n);
  A.out = out; A.bcap = c->batch_cap;
  A.ws = (float*)c->ca_ws.p; A.ws_stride = wsf; A.slab = (float*)c->ca_slab.p;
  A.g_lt = (float*)c->g_lt.p; A.mult_lt = (int*)c->mult_lt.p; A.nseq_lt = (int*)c->nseq_lt.p;
  A.g_wd = (float*)c->g_wd.p; A.mult_wd = (int*)c->mult_wd.p; A.nseq_wd = (int*)c->nseq_wd.p;
  if (fast) {
    // (the recurrence kernel is bound by the L2 stream of the interval matrices: eight workgroups per CU instead of two, +29 %)
    // outer-product path (carnn.hip): packed per-step state + sorted entries, matrix gradients on the matrix cores
    const size_t Tcap = (size_t)n * (size_t)(T->max_len > 1 ? T->max_len - 1 : 1), Ne = 6 * Tcap, NK = (size_t)P->n_dist + 2;
    const size_t n_chunk = Ne / 512 + NK + 1;
    const size_t N2 = 3 * Tcap;
    const size_t nint = (size_t)n + 8 + 6 * Ne + (size_t)RS_HIST_INTS + RS_MAXBIN + 16 + 3 * (NK + 8) + 4 * N2 + 2 * ((size_t)P->n_item + 8) + 32;
    const size_t nfl = (Tcap + (size_t)n + 1) * D + Tcap * 5 * D + n_chunk * (size_t)D * D + ((size_t)P->n_item + 4) * D + N2 * D + 64;
    if ((rc = ensure(c, c->ca2, 4 * (nint + nfl) + 256, st))) return rc;
    float* f = (float*)c->ca2.p;
    auto take = [&](size_t cnt) { float* r = f; f += (cnt + 3) & ~(size_t)3; return r; };
    A.Hpk = take((Tcap + (size_t)n + 1) * D); A.EA = take(Tcap * 5 * D); A.partial = take(n_chunk * (size_t)D * D);
    A.PM = take(((size_t)P->n_item + 2) * D);
    A.vpart = take(N2 * D);
    if (n < 16) A.PM = nullptr;      // a handful of sequences: three GEMVs per step are cheaper than the table (measured: a wash at one)
    int* ip = (int*)f;
    auto itake = [&](size_t cnt) { int* r = ip; ip += (cnt + 3) & ~(size_t)3; return r; };
    A.soff = itake((size_t)n + 1); A.keys0 = itake(Ne); A.keys1 = itake(Ne); A.vals0 = itake(Ne); A.vals1 = itake(Ne);
    A.ent_a = itake(Ne); A.ent_b = itake(Ne); A.hist = itake((size_t)RS_HIST_INTS + RS_MAXBIN); A.cnt = itake(8);
    A.seg_start = itake(NK + 4); A.seg_end = itake(NK + 4); A.chunk_first = itake(NK + 4);
    A.k2a = itake(N2); A.k2b = itake(N2); A.v2a = itake(N2); A.v2b = itake(N2);
    A.seg2_start = itake((size_t)P->n_item + 4); A.seg2_end = itake((size_t)P->n_item + 4);
    A.n_chunk_cap = (int)n_chunk;
    HIPCHK(c, poi::launch_carnn_train2(A, grid, alpha, lambda, st, &c->tm));
    return POI_OK;
  }
  HIPCHK(c, poi::launch_carnn_train(A, grid, alpha, lambda, st, &c->tm));
  return POI_OK;
}

int poi_carnn_predict(poi_ctx* c, const poi_carnn_params* P, const poi_seq_tables* T, const int32_t* uidx, int32_t n, float* hts, void* stream) {
  int rc = check_carnn(c, P, T, false);
  if (rc) return rc;
  if (!uidx || !hts || n < 0) return fail(c, POI_EINVAL, "uidx/hts NULL or n < 0");
  if (n == 0) return POI_OK;
  hipStream_t st = (hipStream_t)stream;
  HIPCHK(c, hipSetDevice(c->device));
  if ((rc = ensure(c, c->ca_scr, sizeof(float) * ((size_t)(P->n_dist + 1) * P->dim + (size_t)P->n_item + 2048), st))) return rc;
  poi::CaArgs A;
  fill_carnn(A, P, T, uidx, n);
  A.hts = hts;
  int grid = c->num_cu * 4;
  if (grid > n) grid = n;
  HIPCHK(c, poi::launch_carnn_predict(A, grid, (float*)c->ca_scr.p, st, &c->tm));
  return POI_OK;
}

int poi_carnn_score_all(poi_ctx* c, const float* users, const float* items, const float* M, const float* dists, const double* coords,
                        const double* cphi, const double* thr, const int32_t* last_poi, int32_t n, int32_t n_item, int32_t n_dist, int32_t dim,
                        double dd, float* scores_out, void* stream) {
  if (!c || !users || !items || !M || !dists || !coords || !cphi || !thr || !last_poi || !scores_out) return fail(c, POI_EINVAL, "poi_carnn_score_all: NULL argument");
  if (n < 0 || n_item <= 0 || n_dist <= 0 || dim <= 0 || !(dd > 0)) return fail(c, POI_EINVAL, "bad sizes");
  if (n == 0) return POI_OK;
  hipStream_t st = (hipStream_t)stream;
  HIPCHK(c, hipSetDevice(c->device));
  int rc;
  if ((rc = ensure(c, c->ca_scr, sizeof(float) * ((size_t)(n_dist + 1) * dim + (size_t)n_item + 2048), st))) return rc;
  for (int32_t o = 0; o < n; o += 32768) {
    const int32_t m = n - o < 32768 ? n - o : 32768;
    HIPCHK(c, poi::launch_carnn_score(users + (size_t)o * dim, items, M, dists, coords, cphi, thr, last_poi + o, m, n_item, n_dist, dim, dd,
                                      (float*)c->ca_scr.p, scores_out + (size_t)o * n_item, st, &c->tm));
  }
  return POI_OK;
}

// ---------------------------------------------------------------------------------------------
int poi_bpr_step(poi_ctx* c, float* ux, float* lt, int32_t n_user, int32_t n_item, int32_t dim,
                 const int32_t* uidx, const int32_t* p, const int32_t* q, int32_t n,
                 float alpha, float lambda, float* loss_out, int mode, void* stream) {
  if (!c || !ux || !lt || !uidx || !p || !q || !loss_out) return fail(c, POI_EINVAL, "poi_bpr_step: NULL argument");
  if (is_f16(c, ux)) return fail(c, POI_ENOTSUP, "BPR-MF keeps the user table in float32 (a half POI table is supported in snapshot mode)");
  if (is_f16(c, lt) && mode != POI_BPR_SNAPSHOT) return fail(c, POI_ENOTSUP, "a half POI table needs POI_BPR_SNAPSHOT");
  if (dim <= 0 || dim % 4 != 0 || dim > 1024) return fail(c, POI_ENOTSUP, "dim must be a multiple of 4 in [4, 1024] (got %d)", dim);
  if (n < 0 || n_user <= 0 || n_item <= 0) return fail(c, POI_EINVAL, "bad sizes");
  if ((int64_t)n * 3 >= (int64_t)1 << 31) return fail(c, POI_ENOTSUP, "at most 2^31 / 3 triples per launch");
  if (mode != POI_BPR_SNAPSHOT && mode != POI_BPR_HOGWILD) return fail(c, POI_EINVAL, "unknown mode %d", mode);
  if (c->batch_cap == 0.0f) return fail(c, POI_ENOTSUP, "the mini-batch rule (batch cap 0) applies to poi_gru_step / poi_spatial_step only");
  if (n == 0) return POI_OK;
  hipStream_t st = (hipStream_t)stream;
  HIPCHK(c, hipSetDevice(c->device));
  poi::BprArgs A;
  memset(&A, 0, sizeof A);
  A.ux = ux; A.lt = lt; A.n_user = n_user; A.n_item = n_item; A.dim = dim;
  A.lt_f16 = is_f16(c, lt);
  A.sr_salt = (c->f16_rounding && A.lt_f16) ? (++c->sr_counter) * 0x9E3779B1u | 1u : 0u;
  A.uidx = uidx; A.p = p; A.q = q; A.n = n; A.alpha = alpha; A.lambda = lambda; A.loss = loss_out; A.bcap = c->batch_cap;
  { int rc0 = ensure(c, c->bad_ids, 64, st); if (rc0) return rc0; }      // (zero-filled at allocation)
  A.bad = (int*)c->bad_ids.p;
  if (mode == POI_BPR_SNAPSHOT) {
    // workspace: the 3 n touches' sort buffers, per-triple coefficients, per-window partial sums; the shadow user table (grow-only, ctx-owned)
    int rc;
    size_t ni = 0, nf = 0;
    poi::bpr_ws_sizes(n, dim, &ni, &nf);
    if ((rc = ensure(c, c->g_ux, sizeof(float) * (size_t)n_user * dim, st))) return rc;
    if ((rc = ensure(c, c->g_blt, sizeof(int) * ni + sizeof(float) * nf + 256, st))) return rc;
    A.shadow = (float*)c->g_ux.p;
    const size_t chunks = (size_t)(n + 63) / 64 + (size_t)(2 * (size_t)n + 63) / 64 + 2, per = 3 * (size_t)n + 64;
    int* ip = (int*)c->g_blt.p;
    A.keys0 = ip; A.keys1 = ip + per; A.vals0 = ip + 2 * per; A.vals1 = ip + 3 * per; ip += 4 * per;
    A.hist = ip; ip += RS_HIST_INTS + RS_MAXBIN;
    A.cnt = ip; ip += 64;
    A.meta = (int4*)ip; ip += 4 * chunks;
    float* fp = (float*)ip;
    A.g = fp; fp += ((size_t)n + 64 + 3) & ~(size_t)3;      // (lead / trail rows are read as float4)
    A.lead = fp; fp += chunks * (size_t)dim;
    A.trail = fp;
  }
  HIPCHK(c, poi::launch_bpr(A, mode, c->num_cu, st, &c->tm));
  return POI_OK;
}

// ---------------------------------------------------------------------------------------------
struct UlptaiArg { const void* bins; int bin_bytes; const float* sts; int n_dist; const double *coords, *cphi, *thr; const int* last_poi; double dd; };

static int score_common(poi_ctx* c, const float* users, const float* items, int32_t n, int32_t n_item, int32_t dim,
                        const float* wd, const float* prob, float* scores, int32_t k, int32_t* idx_out, float* score_out,
                        void* stream, const UlptaiArg* U = nullptr) {
  // the top-K seed is "consumed by the next call" whatever that call does: taken (and cleared) before any early return, so a failed or
  // empty call can never leave a stale pointer - sized for another n - armed for a later one
  const int32_t* seed_idx = c ? c->seed_idx : nullptr; const int seed_k = c ? c->seed_k : 0;
  if (c) { c->seed_idx = nullptr; c->seed_k = 0; }
  if (!c || !users || !items) return fail(c, POI_EINVAL, "score: NULL argument");
  if (dim <= 0 || dim % 4 != 0 || dim > 256) return fail(c, POI_ENOTSUP, "dim must be a multiple of 4 in [4, 256] (got %d)", dim);
  if (n < 0 || n_item <= 0) return fail(c, POI_EINVAL, "bad sizes");
  if (prob && !wd) return fail(c, POI_EINVAL, "prob given without wd");
  if (k < 0 || k > 32 || (k > 0 && k > n_item)) return fail(c, POI_ENOTSUP, "top-K supports 1 <= k <= min(32, n_item) (got %d)", k);
  if (n == 0) return POI_OK;
  hipStream_t st = (hipStream_t)stream;
  HIPCHK(c, hipSetDevice(c->device));
  poi::ScoreArgs A;
  memset(&A, 0, sizeof A);
  A.users = users; A.items = items; A.items_f16 = is_f16(c, items); A.n = n; A.n_item = n_item; A.dim = dim; A.wd = wd; A.prob = prob;
  A.scores = scores; A.k = k; A.idx_out = idx_out; A.score_out = score_out;
  if (U) { A.ulptai = U->bins; A.bin_bytes = U->bin_bytes; A.sts = U->sts; A.n_dist = U->n_dist; }
  if (U && !U->bins) { A.geo = 1; A.coords = U->coords; A.cphi = U->cphi; A.thr = U->thr; A.last_poi = U->last_poi; A.dd = U->dd; }
  if (const char* e = getenv("POI_SCORE_DBG")) A.dbg = atoi(e);
  const int ntile = (n_item + 31) / 32;
  // variant: 0 = one item stream per wave (row-per-lane loads; small n or dim > 128), 1 = packed item
  // stream per wave with the user tile's A fragments in LDS (default for n >= 128)
  int variant = (n >= 128 && dim <= 128) ? 1 : 0;
  if (c->score_variant >= 0 && dim <= 128) variant = c->score_variant;
  if (U && U->bins) variant = 1;      // the bin matrix is laid out for the packed-stream kernel
  if (U && !U->bins) variant = 0;     // bins on the fly: the row-per-lane kernel (any dim <= 256)
  // ... or, for enough users to fill the chip with eight-wave workgroups and a wide model, the packed-stream GEO kernel
  if (U && !U->bins && k > 0 && n >= 1024 && dim >= 128 && poi::score_geo_stream_lds(dim, U->n_dist) <= 160 * 1024 && c->score_variant != 0) variant = 2;
  const int n_utile = (n + 31) / 32;
  // item ranges per user tile for a table of nt item tiles: long item streams keep per-user thresholds high (few top-K compactions)
  auto splits_for = [&](int nt) {
    int want = (c->num_cu * 8 + n_utile - 1) / n_utile;   // 8 waves per CU
    if (want > nt / 8) want = nt / 8;          // >= 8 tiles per stream: amortise the per-wave top-K epilogue
    if (want < 1) want = 1;
    if (want < (nt + 2046) / 2047) want = (nt + 2046) / 2047;      // candidate lists hold 16-bit item offsets: < 65536 items per range
    return variant == 2 ? ((want + 7) / 8) * 8 : ((want + 3) / 4) * 4;
  };
  const int n_split = splits_for(ntile);
  A.n_split = n_split;
  const int n_pad = n_utile * 32;
  int rc;
  // one-stage kernel of the chosen variant + merge of the per-range lists, on the item table X describes
  auto one_stage = [&](poi::ScoreArgs& X) -> int {
    const int nt = (X.n_item + 31) / 32;
    if (variant == 2) {
      const int d8 = dim <= 128 ? 16 : 32;
      int r2 = ensure(c, c->items_pk, sizeof(float) * 4 * (size_t)nt * d8 * 64, st);
      if (r2) return r2;
      X.items_packed = (float4*)c->items_pk.p;
      HIPCHK(c, poi::launch_score_geo_stream(X, st, &c->tm));
    } else if (variant == 1) {
      const int d8 = dim <= 32 ? 4 : dim <= 64 ? 8 : 16;
      int r2 = ensure(c, c->items_pk, sizeof(float) * 4 * (size_t)nt * d8 * 64, st);
      if (r2) return r2;
      X.items_packed = (float4*)c->items_pk.p;
      HIPCHK(c, poi::launch_score_packed(X, st, &c->tm));
    } else {
      HIPCHK(c, poi::launch_score(X, st, &c->tm));
    }
    if (k > 0) HIPCHK(c, poi::launch_topk_merge(X, X.n_split, n_pad, st));
    return POI_OK;
  };
  bool two_stage = false, use_maxpass = false;
  if (k > 0) {
    const size_t cand = (size_t)n_split * n_pad * k;
    if ((rc = ensure(c, c->cand_s, sizeof(float) * cand, st))) return rc;
    if ((rc = ensure(c, c->cand_i, sizeof(int) * cand, st))) return rc;
    A.cand_score = (float*)c->cand_s.p; A.cand_idx = (int*)c->cand_i.p;
    const bool seeded = seed_idx && seed_k >= k && seed_k <= 64;
    {
      poi::ScoreArgs probe = A; probe.seeded = 1;
      two_stage = c->topk_filter && (variant == 1 || A.geo) && poi::score_two_stage_supported(probe);
    }
    // SELF-SEEDING pre-pass of the two-stage path: the one-stage kernel on the first 1/16 of the item tiles (1/64 when the caller's
    // seed already gave bounds) - the K-th best EXACT score of any subset is a lower bound of the final K-th best, so the filter pass
    // starts from thresholds that leave ~16 K (64 K) survivors per user whatever the seed holds: an unseeded call (the first evaluation
    // of a run) or a useless seed (a model that moved a lot) no longer sends its tiles to the one-stage kernel.  Unseeded calls always
    // take it; seeded ones when the table has >= 2^20 items (there it costs < 2 % of the call).
    const int sub_tiles = !two_stage ? 0 : !seeded ? (ntile >= 256 ? ntile / 16 : 0) : (n_item >= (1 << 20) ? ntile / 64 : 0);
    if (two_stage && !seeded && sub_tiles == 0) two_stage = false;      // (a small table and no seed: one-stage)
    if (n_split > 1 || seeded || two_stage) {
      if ((rc = ensure(c, c->gbound, sizeof(unsigned) * (size_t)n_pad, st))) return rc;
      HIPCHK(c, hipMemsetAsync(c->gbound.p, 0, sizeof(unsigned) * (size_t)n_pad, st));
      A.gbound = (unsigned*)c->gbound.p;
    }
    if (seeded) {
      c->tm.begin("topk_seed", st);
      HIPCHK(c, poi::launch_topk_seed(A, seed_idx, seed_k, st));
      c->tm.end(st);
      A.seeded = 1;
    }
    // unseeded, resident bin matrix / no distance term, dims 64 / 128: thresholds from the block maxima of the f16 lower bounds instead
    // (score_filter.hip, MAXP: one f16 pass over ALL items, ~1.3 K survivors per user against ~17 K of the float32 prefix pre-pass)
    use_maxpass = two_stage && !seeded && sub_tiles > 0 && poi::score_maxpass_supported(A);
    if (const char* e = getenv("POI_SF_MAXPASS")) use_maxpass = use_maxpass && atoi(e) != 0;
    if (two_stage && sub_tiles > 0 && !use_maxpass) {
      if ((rc = ensure(c, c->pre_idx, sizeof(int) * (size_t)n_pad * k, st)) || (rc = ensure(c, c->pre_sc, sizeof(float) * (size_t)n_pad * k, st))) return rc;
      poi::ScoreArgs S = A;
      S.n_item = sub_tiles * 32; S.bins_ntile = ntile; S.n_split = splits_for(sub_tiles);
      S.idx_out = (int*)c->pre_idx.p; S.score_out = (float*)c->pre_sc.p;
      if ((size_t)S.n_split * n_pad * k > cand) {
        if ((rc = ensure(c, c->cand_s, sizeof(float) * (size_t)S.n_split * n_pad * k, st)) || (rc = ensure(c, c->cand_i, sizeof(int) * (size_t)S.n_split * n_pad * k, st))) return rc;
        A.cand_score = S.cand_score = (float*)c->cand_s.p; A.cand_idx = S.cand_idx = (int*)c->cand_i.p;
      }
      if ((rc = one_stage(S))) return rc;
      HIPCHK(c, poi::launch_topk_bound(S.score_out, n, k, A.gbound, st));
      A.seeded = 1;
    }
  }
  if (two_stage) {
    // two-stage: f16 filter pass + exact float32 rescoring of the survivors (score_filter.hip); the one-stage kernel below then only
    // runs the user tiles whose survivor lists overflowed (A.tile_flag)
    const int kg = dim / 16, cap = poi::score_filter_cap();
    if ((rc = ensure(c, c->items_pk16, sizeof(uint4) * (size_t)ntile * kg * 64, st)) || (rc = ensure(c, c->inorm, sizeof(float2) * (size_t)ntile * 32, st)) ||
        (rc = ensure(c, c->surv_cnt, sizeof(int) * (size_t)n_pad, st)) || (rc = ensure(c, c->surv_idx, sizeof(int) * (size_t)n_pad * cap, st)) ||
        (rc = ensure(c, c->surv_sc, sizeof(float) * (size_t)n_pad * cap, st)) ||
        (rc = ensure(c, c->tflag, sizeof(int) * (size_t)n_utile, st))) return rc;
    HIPCHK(c, hipMemsetAsync(c->surv_cnt.p, 0, sizeof(int) * (size_t)n_pad, st));
    HIPCHK(c, hipMemsetAsync(c->tflag.p, 0, sizeof(int) * (size_t)n_utile, st));
    A.items_packed16 = (const uint4*)c->items_pk16.p; A.inorm = (const float2*)c->inorm.p;
    A.surv_cnt = (int*)c->surv_cnt.p; A.surv_idx = (int*)c->surv_idx.p; A.surv_sc = (float*)c->surv_sc.p; A.tile_flag = (int*)c->tflag.p;
    // GEO with a huge item table and few users (config X's evaluation): the item-stationary filter - every user tile's own pass over the
    // item table is 1.3 TB at 8192 users x 10 M POIs; forced (2) / forbidden (0) by POI_SF_ITEMS for tests and A/B runs
    A.n_cu = c->num_cu;
    {
      int items_mode = (A.geo && n_item >= (1 << 20) && n_utile <= 4096) ? 1 : 0;
      if (const char* e = getenv("POI_SF_ITEMS")) items_mode = A.geo ? (atoi(e) != 0) : 0;
      if (c->sf_items >= 0) items_mode = A.geo ? c->sf_items : 0;
      if (items_mode) {
        if ((rc = ensure(c, c->users_pk16, sizeof(uint4) * (size_t)n_utile * kg * 64, st)) || (rc = ensure(c, c->ubound, sizeof(float) * 4 * (size_t)n_pad, st)) ||
            (rc = ensure(c, c->ugeo, sizeof(double) * 3 * (size_t)n_pad, st))) return rc;
        A.users_packed16 = (uint4*)c->users_pk16.p; A.ubound = (float4*)c->ubound.p; A.ugeo = (double*)c->ugeo.p;
      }
    }
    int nsf = ((4 * c->num_cu + n_utile - 1) / n_utile) * 4;      // >= 4 workgroups (16 waves) per CU
    if (nsf < 16) nsf = 16;      // (swept at the Gowalla shape: 8 / 16 / 32 / 64 / 128 ranges -> 3.09 / 2.84 / 2.80 / 2.86 / 3.21 ms of filter time)
    if (const char* e = getenv("POI_SF_NSPLIT")) { const int v = atoi(e); if (v >= 4) nsf = (v / 4) * 4; }      // tuning switch
    if (nsf > (ntile / 4) * 4) nsf = (ntile / 4) * 4;
    if (nsf < 4) nsf = 4;
    if (use_maxpass) {
      HIPCHK(c, poi::launch_score_maxpass(A, nsf, st, &c->tm));
      A.seeded = 1;
    }
    HIPCHK(c, poi::launch_score_two_stage(A, nsf, st, &c->tm));
    c->last_two_n = n; c->last_two_tiles = n_utile;
  }
  return one_stage(A);
}

int poi_score_all(poi_ctx* c, const float* users, const float* items, int32_t n, int32_t n_item, int32_t dim,
                  const float* wd, const float* prob, float* scores_out, void* stream) {
  if (!scores_out) return fail(c, POI_EINVAL, "scores_out is NULL");
  return score_common(c, users, items, n, n_item, dim, wd, prob, scores_out, 0, nullptr, nullptr, stream);
}

int poi_score_topk(poi_ctx* c, const float* users, const float* items, int32_t n, int32_t n_item, int32_t dim,
                   const float* wd, const float* prob, int32_t k, int32_t* idx_out, float* score_out, void* stream) {
  if (!idx_out || k <= 0) return fail(c, POI_EINVAL, "idx_out NULL or k <= 0");
  return score_common(c, users, items, n, n_item, dim, wd, prob, nullptr, k, idx_out, score_out, stream);
}

int poi_ulptai_build(poi_ctx* c, const double* coords, const double* cphi, const double* thr, const int32_t* last_poi,
                     int32_t n_user, int32_t n_item, int32_t n_dist, double dd, void* out, int32_t bin_bytes, void* stream) {
  if (!c || !coords || !cphi || !thr || !last_poi || !out) return fail(c, POI_EINVAL, "poi_ulptai_build: NULL argument");
  if (n_user <= 0 || n_item <= 0 || n_dist <= 0 || !(dd > 0)) return fail(c, POI_EINVAL, "bad sizes");
  if (bin_bytes != 1 && bin_bytes != 2) return fail(c, POI_EINVAL, "bin_bytes must be 1 or 2");
  if ((bin_bytes == 1 && n_dist > 255) || n_dist > 65535) return fail(c, POI_EINVAL, "n_dist %d does not fit %d-byte bins", n_dist, bin_bytes);
  HIPCHK(c, hipSetDevice(c->device));
  c->tm.begin("ulptai_build", (hipStream_t)stream);
  HIPCHK(c, poi::launch_ulptai(coords, cphi, thr, last_poi, n_user, n_item, n_dist, dd, out, bin_bytes, (hipStream_t)stream));
  c->tm.end((hipStream_t)stream);
  return POI_OK;
}

int poi_score_topk_ulptai(poi_ctx* c, const float* users, const float* items, int32_t n, int32_t n_item, int32_t dim,
                          const float* wd, const float* sts, const void* ulptai, int32_t bin_bytes, int32_t n_dist,
                          int32_t k, int32_t* idx_out, float* score_out, void* stream) {
  if (!idx_out || k <= 0) return fail(c, POI_EINVAL, "idx_out NULL or k <= 0");
  if (!wd || !sts || !ulptai) return fail(c, POI_EINVAL, "poi_score_topk_ulptai: wd / sts / ulptai NULL");
  if (bin_bytes != 1 && bin_bytes != 2) return fail(c, POI_EINVAL, "bin_bytes must be 1 or 2");
  if (dim > 128) return fail(c, POI_ENOTSUP, "the bin-matrix path supports dim <= 128 (got %d)", dim);
  if (n_dist <= 0 || (int64_t)n * (n_dist + 1) >= (int64_t)1 << 31) return fail(c, POI_EINVAL, "n * (n_dist + 1) must stay below 2^31: score in batches");
  const UlptaiArg U{ulptai, bin_bytes, sts, n_dist, nullptr, nullptr, nullptr, nullptr, 0.0};
  return score_common(c, users, items, n, n_item, dim, wd, nullptr, nullptr, k, idx_out, score_out, stream, &U);
}

int poi_score_topk_geo(poi_ctx* c, const float* users, const float* items, int32_t n, int32_t n_item, int32_t dim, const float* wd, const float* sts,
                       const double* coords, const double* cphi, const double* thr, const int32_t* last_poi, int32_t n_dist, double dd,
                       int32_t k, int32_t* idx_out, float* score_out, void* stream) {
  if (!idx_out || k <= 0) return fail(c, POI_EINVAL, "idx_out NULL or k <= 0");
  if (!wd || !sts || !coords || !cphi || !thr || !last_poi) return fail(c, POI_EINVAL, "poi_score_topk_geo: NULL argument");
  if (n_dist <= 0 || !(dd > 0)) return fail(c, POI_EINVAL, "bad n_dist / dd");
  const UlptaiArg U{nullptr, 0, sts, n_dist, coords, cphi, thr, last_poi, dd};
  return score_common(c, users, items, n, n_item, dim, wd, nullptr, nullptr, k, idx_out, score_out, stream, &U);
}

int poi_topk(poi_ctx* c, const float* scores, int32_t n, int32_t n_item, int32_t k, int32_t* idx_out, float* score_out,
             void* stream) {
  if (!c || !scores || !idx_out) return fail(c, POI_EINVAL, "poi_topk: NULL argument");
  if (k <= 0 || k > 64 || k > n_item) return fail(c, POI_ENOTSUP, "poi_topk supports 1 <= k <= min(64, n_item) (got %d)", k);
  if (n < 0) return fail(c, POI_EINVAL, "n < 0");
  if (n == 0) return POI_OK;
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, poi::launch_topk_rows(scores, n, n_item, k, idx_out, score_out, (hipStream_t)stream));
  return POI_OK;
}

int poi_auc_preference(poi_ctx* c, const float* users, const float* items, int32_t n, int32_t dim,
                       const int32_t* tp, const int32_t* tq, const int32_t* tm, int32_t len, uint8_t* out, void* stream) {
  if (!c || !users || !items || !tp || !tq || !tm || !out) return fail(c, POI_EINVAL, "poi_auc_preference: NULL argument");
  if (dim <= 0 || dim % 4 != 0) return fail(c, POI_ENOTSUP, "dim must be a positive multiple of 4");
  if (n < 0 || len < 0) return fail(c, POI_EINVAL, "bad sizes");
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, poi::launch_auc(users, items, is_f16(c, items), n, dim, tp, tq, tm, len, out, (hipStream_t)stream));
  return POI_OK;
}

int poi_sumsq(poi_ctx* c, const float* x, int64_t n, double* out, void* stream) {
  if (!c || !x || !out || n < 0) return fail(c, POI_EINVAL, "poi_sumsq: bad argument");
  if (n == 0) return POI_OK;
  HIPCHK(c, hipSetDevice(c->device));
  if (is_f16(c, x)) HIPCHK(c, poi::launch_sumsq_f16(x, n, out, (hipStream_t)stream));
  else HIPCHK(c, poi::launch_sumsq(x, n, out, (hipStream_t)stream));
  return POI_OK;
}

int poi_dist_prob(poi_ctx* c, const double* coords, const double* cphi, const double* thr, const int32_t* last_poi,
                  const float* sts, int32_t n, int32_t n_item, int32_t n_dist, double dd, float* prob_out, void* stream) {
  if (!c || !coords || !last_poi || !sts || !prob_out) return fail(c, POI_EINVAL, "poi_dist_prob: NULL argument");
  if ((cphi == nullptr) != (thr == nullptr)) return fail(c, POI_EINVAL, "poi_dist_prob: cphi and thr go together");
  if (n < 0 || n_item <= 0 || n_dist <= 0 || !(dd > 0)) return fail(c, POI_EINVAL, "bad sizes");
  HIPCHK(c, hipSetDevice(c->device));
  for (int32_t o = 0; o < n; o += 32768) {
    const int32_t m = n - o < 32768 ? n - o : 32768;
    c->tm.begin("dist_prob", (hipStream_t)stream);
    HIPCHK(c, poi::launch_dist_prob(coords, cphi, thr, last_poi + o, sts + (size_t)o * (n_dist + 1), m, n_item, n_dist, dd,
                                    prob_out + (size_t)o * n_item, (hipStream_t)stream));
    c->tm.end((hipStream_t)stream);
  }
  return POI_OK;
}

int poi_rank_metrics(poi_ctx* c, const int32_t* ranks, int32_t n, int32_t k, const int32_t* tes_p, const int32_t* tes_mask,
                     int32_t len_tes, const int32_t* at_nums, int32_t n_at, double* acc, void* stream) {
  if (!c || !ranks || !tes_p || !tes_mask || !at_nums || !acc) return fail(c, POI_EINVAL, "poi_rank_metrics: NULL argument");
  if (n < 0 || k <= 0 || len_tes <= 0 || n_at <= 0 || n_at > 8) return fail(c, POI_EINVAL, "poi_rank_metrics: bad sizes (n_at <= 8)");
  if (n == 0) return POI_OK;
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, poi::launch_rank_metrics(ranks, n, k, tes_p, tes_mask, len_tes, at_nums, n_at, acc, (hipStream_t)stream));
  return POI_OK;
}

int poi_sample_negatives(poi_ctx* c, const int32_t* off, const int32_t* p, int32_t n_user, int32_t n_item, const int32_t* tes_p,
                         const int32_t* tes_mask, int32_t len_tes, uint64_t seed, int32_t* q_out, int32_t* tes_q_out, void* stream) {
  if (!c || !off || !p || !q_out) return fail(c, POI_EINVAL, "poi_sample_negatives: NULL argument");
  if (tes_q_out && (!tes_p || !tes_mask || len_tes <= 0)) return fail(c, POI_EINVAL, "poi_sample_negatives: test tables missing");
  if (n_user < 0 || n_item <= 0) return fail(c, POI_EINVAL, "bad sizes");
  if (n_user == 0) return POI_OK;
  HIPCHK(c, hipSetDevice(c->device));
  c->tm.begin("sample_neg", (hipStream_t)stream);
  HIPCHK(c, poi::launch_sample_neg(off, p, n_user, n_item, tes_p, tes_mask, len_tes, seed, q_out, tes_q_out, (hipStream_t)stream));
  c->tm.end((hipStream_t)stream);
  return POI_OK;
}

int poi_neg_dist_bins(poi_ctx* c, const int32_t* off, const int32_t* p, const int32_t* q, int32_t n_user, const double* coords,
                      const double* cphi, const double* thr, int32_t n_dist, double dd, int32_t* dq_out, void* stream) {
  if (!c || !off || !p || !q || !coords || !cphi || !thr || !dq_out) return fail(c, POI_EINVAL, "poi_neg_dist_bins: NULL argument");
  if (n_user < 0 || n_dist <= 0 || !(dd > 0)) return fail(c, POI_EINVAL, "bad sizes");
  if (n_user == 0) return POI_OK;
  HIPCHK(c, hipSetDevice(c->device));
  c->tm.begin("neg_dist", (hipStream_t)stream);
  HIPCHK(c, poi::launch_neg_dist(off, p, q, n_user, coords, cphi, thr, n_dist, dd, dq_out, (hipStream_t)stream));
  c->tm.end((hipStream_t)stream);
  return POI_OK;
}

int poi_delta_make(poi_ctx* c, const float* cur, const float* base, float* delta, int64_t n, void* stream) {
  if (!c || !cur || !base || !delta || n < 0) return fail(c, POI_EINVAL, "poi_delta_make: bad argument");
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, poi::launch_delta_make(cur, base, delta, n, (hipStream_t)stream));
  return POI_OK;
}

int poi_delta_apply(poi_ctx* c, float* cur, const float* base, const float* delta_sum, int64_t n, void* stream) {
  if (!c || !cur || !base || !delta_sum || n < 0) return fail(c, POI_EINVAL, "poi_delta_apply: bad argument");
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, poi::launch_delta_apply(cur, base, delta_sum, n, (hipStream_t)stream));
  return POI_OK;
}

int poi_ctx_set_engine(poi_ctx* c, int engine) {
  if (!c || engine < 0 || engine > 4) return fail(c, POI_EINVAL, "engine must be 0 (auto), 1 (per-sequence), 2 (tile), 3 (tile, streaming recurrent kernels) or 4 (exact: float64)");
  c->engine = engine;
  return POI_OK;
}

int poi_ctx_set_f16_rounding(poi_ctx* c, int mode, uint32_t seed) {
  if (!c || mode < 0 || mode > 1) return fail(c, POI_EINVAL, "poi_ctx_set_f16_rounding: mode must be 0 (nearest) or 1 (stochastic)");
  c->f16_rounding = mode; c->sr_counter = seed;
  return POI_OK;
}

int poi_ctx_set_split_products(poi_ctx* c, int on) {
  if (!c || on < 0 || on > 1) return fail(c, POI_EINVAL, "poi_ctx_set_split_products: on must be 0 or 1");
  c->rec_split = on;
  return POI_OK;
}

int poi_ctx_set_exact_forward(poi_ctx* c, int on, int per_sequence_max) {
  if (!c || on < 0 || on > 1) return fail(c, POI_EINVAL, "poi_ctx_set_exact_forward: on must be 0 or 1");
  c->xfwd = on;
  if (per_sequence_max >= 0) c->xrec1_max = per_sequence_max;
  drop_graphs(c);
  return POI_OK;
}

int poi_ctx_set_option(poi_ctx* c, const char* name, int value) {
  if (!c || !name) return fail(c, POI_EINVAL, "poi_ctx_set_option: null argument");
  struct Opt { const char* name; int* p; int lo, hi; };
  const Opt opts[] = {{"forward_table_compact", &c->xcomp, 0, 1}, {"forward_table_compact_min", &c->xcomp_min, 0, 1 << 30}, {"head_split", &c->head3, 0, 1},
                      {"early_bins", &c->early_bins, 0, 1}, {"hot_bins", &c->hot_bins, 0, 1}, {"hybrid", &c->hybrid, 0, 1}, {"hybrid_min", &c->hyb_min, 0, 1 << 30}, {"hybrid_max", &c->hyb_max, 0, 1 << 30}, {"hybrid_force", &c->hyb_force, 0, 1 << 30}};
  for (const Opt& o : opts)
    if (!strcmp(name, o.name)) {
      if (value < o.lo || value > o.hi) return fail(c, POI_EINVAL, "poi_ctx_set_option: %s must be in [%d, %d] (got %d)", name, o.lo, o.hi, value);
      *o.p = value;
      drop_graphs(c);
      return POI_OK;
    }
  return fail(c, POI_EINVAL, "poi_ctx_set_option: unknown option '%s'", name);
}

int poi_ctx_set_small_launch(poi_ctx* c, int max_sequences) {
  if (!c || max_sequences < 0) return fail(c, POI_EINVAL, "poi_ctx_set_small_launch: max_sequences must be >= 0");
  c->rec1_max = max_sequences;
  return POI_OK;
}

int poi_ctx_set_regroup_min(poi_ctx* c, int min_sequences) {
  if (!c || min_sequences < 0) return fail(c, POI_EINVAL, "poi_ctx_set_regroup_min: min_sequences must be >= 0");
  c->bintab_min = min_sequences;
  return POI_OK;
}

int poi_ctx_set_one_sequence_path(poi_ctx* c, int on) {
  if (!c || on < 0 || on > 1) return fail(c, POI_EINVAL, "poi_ctx_set_one_sequence_path: on must be 0 or 1");
  c->one_path = on;
  return POI_OK;
}

int poi_ctx_set_topk_filter(poi_ctx* c, int on) {
  if (!c || on < 0 || on > 3) return fail(c, POI_EINVAL, "poi_ctx_set_topk_filter: on must be 0 (one-stage only), 1 (two-stage), 2 / 3 (two-stage, item-stationary GEO filter forced / forbidden)");
  c->topk_filter = on != 0;
  c->sf_items = on == 2 ? 1 : on == 3 ? 0 : -1;
  return POI_OK;
}

int poi_ctx_topk_filter_stats(poi_ctx* c, int64_t* users, int64_t* survivors, int64_t* tiles, int64_t* tiles_flagged) {
  if (!c || !users || !survivors || !tiles || !tiles_flagged) return fail(c, POI_EINVAL, "poi_ctx_topk_filter_stats: NULL argument");
  *users = c->last_two_n; *tiles = c->last_two_tiles; *survivors = 0; *tiles_flagged = 0;
  if (c->last_two_n <= 0) return POI_OK;
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipDeviceSynchronize());
  std::vector<int> cnt((size_t)c->last_two_n), fl((size_t)c->last_two_tiles);
  HIPCHK(c, hipMemcpy(cnt.data(), c->surv_cnt.p, sizeof(int) * cnt.size(), hipMemcpyDeviceToHost));
  HIPCHK(c, hipMemcpy(fl.data(), c->tflag.p, sizeof(int) * fl.size(), hipMemcpyDeviceToHost));
  for (int v : cnt) *survivors += v;
  for (int v : fl) *tiles_flagged += v != 0;
  return POI_OK;
}

int poi_ctx_set_topk_seed(poi_ctx* c, const int32_t* seed_idx, int32_t k_seed) {
  if (!c || (seed_idx && (k_seed <= 0 || k_seed > 64))) return fail(c, POI_EINVAL, "poi_ctx_set_topk_seed: k_seed must be in [1, 64]");
  c->seed_idx = seed_idx; c->seed_k = seed_idx ? k_seed : 0;
  return POI_OK;
}

int poi_ctx_set_graph(poi_ctx* c, int on, int min_n, int max_n) {
  if (!c || on < 0 || on > 1 || min_n < 0 || max_n < min_n) return fail(c, POI_EINVAL, "poi_ctx_set_graph: on must be 0 / 1 and 0 <= min_n <= max_n");
  if (on && !c->cap) {
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamCreateWithFlags(&c->cap, hipStreamNonBlocking));
  }
  c->graph_mode = on; c->graph_min_n = min_n; c->graph_max_n = max_n;
  return POI_OK;
}

int64_t poi_ctx_graph_replays(const poi_ctx* c) { return c ? (int64_t)c->graph_replays : POI_EINVAL; }
int64_t poi_ctx_take_bad_ids(poi_ctx* c, void* stream) {
  if (!c) return POI_EINVAL;
  if (!c->bad_ids.p) return 0;
  hipStream_t st = (hipStream_t)stream;
  int v = 0;
  if (hipSetDevice(c->device) != hipSuccess || hipMemcpyAsync(&v, c->bad_ids.p, sizeof v, hipMemcpyDeviceToHost, st) != hipSuccess ||
      hipStreamSynchronize(st) != hipSuccess) return fail(c, POI_EHIP, "poi_ctx_take_bad_ids: %s", hipGetErrorString(hipGetLastError()));
  if (v && (hipMemsetAsync(c->bad_ids.p, 0, sizeof v, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess)) return fail(c, POI_EHIP, "poi_ctx_take_bad_ids: reset failed");
  return v;
}

int poi_ctx_register_f16(poi_ctx* c, const void* ptr, int64_t bytes) {
  if (!c || !ptr || bytes <= 0) return fail(c, POI_EINVAL, "poi_ctx_register_f16: bad argument");
  for (auto& r : c->f16) if (r.first == (const char*)ptr) { r.second = (size_t)bytes; return POI_OK; }
  c->f16.emplace_back((const char*)ptr, (size_t)bytes);
  return POI_OK;
}

int poi_ctx_unregister_f16(poi_ctx* c, const void* ptr) {
  if (!c) return fail(c, POI_EINVAL, "NULL ctx");
  for (size_t i = 0; i < c->f16.size(); ++i) if (c->f16[i].first == (const char*)ptr) { c->f16.erase(c->f16.begin() + i); return POI_OK; }
  return POI_OK;
}

int poi_ctx_set_batch_cap(poi_ctx* c, float cap) {
  if (!c || !(cap >= 1.0f || cap == 0.0f)) return fail(c, POI_EINVAL, "batch cap must be >= 1 (1 = mean rule) or 0 (mini-batch rule)");
  c->batch_cap = cap;
  return POI_OK;
}

int poi_timing_enable(poi_ctx* c, int on) {
  if (!c) return fail(c, POI_EINVAL, "NULL ctx");
  c->tm.on = on != 0;
  c->tm.period = on > 1 ? on : 1; c->tm.count = 0; c->tm.active = true;
  return POI_OK;
}

int poi_timing_reset(poi_ctx* c) {
  if (!c) return fail(c, POI_EINVAL, "NULL ctx");
  HIPCHK(c, hipDeviceSynchronize());
  c->tm.clear();
  return POI_OK;
}

int poi_timing_get(poi_ctx* c, const char* kernel, double* total_ms, int64_t* launches) {
  if (!c || !kernel || !total_ms || !launches) return fail(c, POI_EINVAL, "poi_timing_get: NULL argument");
  HIPCHK(c, hipDeviceSynchronize());
  double tot = 0; int64_t n = 0;
  for (auto& r : c->tm.recs) {
    if (r.name != kernel) continue;
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) { tot += ms; ++n; }
  }
  *total_ms = tot; *launches = n;
  return POI_OK;
}

int poi_selftest(poi_ctx* c, void* stream) {
  if (!c) return fail(c, POI_EINVAL, "NULL ctx");
  hipStream_t st = (hipStream_t)stream;
  HIPCHK(c, hipSetDevice(c->device));
  int rc = ensure(c, c->st, 64, st);
  if (rc) return rc;
  HIPCHK(c, hipMemsetAsync(c->st.p, 0, 64, st));
  float* buf = (float*)c->st.p;
  int* flag = (int*)((char*)c->st.p + 32);
  HIPCHK(c, poi::launch_selftest(buf, flag, st));
  float hb[8]; int hf = -1;
  HIPCHK(c, hipMemcpyAsync(hb, buf, sizeof hb, hipMemcpyDeviceToHost, st));
  HIPCHK(c, hipMemcpyAsync(&hf, flag, sizeof hf, hipMemcpyDeviceToHost, st));
  HIPCHK(c, hipStreamSynchronize(st));
  if (hf != 0) return fail(c, POI_EHIP, "selftest: %d wave/block reduction mismatches", hf);
  for (int i = 0; i < 8; ++i) if (hb[i] != 256.0f) return fail(c, POI_EHIP, "selftest: float atomic count %g != 256", hb[i]);
  return POI_OK;
}

}  // extern "C"
