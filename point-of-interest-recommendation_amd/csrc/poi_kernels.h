// Internal (non-ABI) declarations shared between the .hip translation units.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include <string>
#include <vector>

namespace poi {

// Optional per-kernel timing with HIP events recorded on the launch stream (bench.py's live
// roofline measurement).  Disabled by default: begin()/end() are then no-ops.
struct Timing {
  struct Rec { std::string name; hipEvent_t a, b; };
  bool on = false;
  // sampling: with period N only every N-th training launch is instrumented (tick() at the launch's entry decides) - an event pair
  // per kernel costs ~7 us of stream serialisation, 7 % of a Gowalla epoch when every launch carries them; the average duration of
  // the sampled launches is the launch duration either way
  int period = 1; unsigned long count = 0; bool active = true;
  void tick() { active = period <= 1 || (count++ % (unsigned long)period) == 0; }
  std::vector<Rec> recs;
  size_t limit = 32768;
  void begin(const char* name, hipStream_t st) {
    if (!on || !active || recs.size() >= limit) { open_ = false; return; }
    Rec r; r.name = name;
    if (hipEventCreate(&r.a) != hipSuccess || hipEventCreate(&r.b) != hipSuccess) { open_ = false; return; }
    (void)hipEventRecord(r.a, st);
    recs.push_back(r); open_ = true;
  }
  void end(hipStream_t st) { if (open_) { (void)hipEventRecord(recs.back().b, st); open_ = false; } }
  // a region that spans other regions (a fork / join over two streams): closed through its index; -1 = not recorded
  long span_begin(const char* name, hipStream_t st) {
    if (!on || !active || recs.size() >= limit) return -1;
    Rec r; r.name = name;
    if (hipEventCreate(&r.a) != hipSuccess || hipEventCreate(&r.b) != hipSuccess) return -1;
    (void)hipEventRecord(r.a, st);
    (void)hipEventRecord(r.b, st);          // (valid even if span_end is never reached)
    recs.push_back(r);
    return (long)recs.size() - 1;
  }
  void span_end(long idx, hipStream_t st) { if (idx >= 0 && (size_t)idx < recs.size()) (void)hipEventRecord(recs[(size_t)idx].b, st); }
  void clear() { for (auto& r : recs) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); } recs.clear(); }
 private:
  bool open_ = false;
};

// Arguments of the per-sequence engine kernels (passed by value).
struct SeqArgs {
  // parameters (device)
  float *lt, *di, *ui, *wh, *bi, *vs, *bs, *wd, *lw;
  int n_item, n_dist, dim;
  // CSR index tables (device)
  const int *off, *p, *q, *dp, *dq;
  int len_max;       // padded row length of the reference tables (analytic padding-row decay)
  int cap;           // capacity in steps of the per-workgroup scratch (>= longest sequence)
  // launch
  const int* uidx;
  int n_seq;
  float* out;        // spatial: 5 per sequence; plain: 1 per sequence
  // scratch (owned by poi_ctx)
  float* ws; size_t ws_stride;      // per-workgroup activations
  float* slab;                      // per-workgroup dense-gradient slabs
  float *g_lt, *g_di;               // zero-initialised gradient tables
  int *mult_lt, *nseq_lt, *mult_di, *nseq_di;
  // predict outputs (row k of the launch goes to output row out_row[k], or k when out_row is null)
  float *hts, *sts; const int* out_row;
  const int* kc_dev; // tile launches: {slabs of the other regions, slabs of the d ui region}, written by te_wgrad (null: n_slab)
  float bcap;        // batch rule: at most `bcap` of the touching sequences' updates count (1 = mean rule), include/poi_hip.h
};

// te_wgrad's (job, K-chunk) grid: `slots` workgroups, nui d ui jobs contracting over P rows and (jobs - nui) jobs over T rows.  Every
// workgroup should get the same number of rows: n_o chunks for the T-row jobs, n_u ~ n_o * P / T for the ui jobs, filling `slots`.
__host__ __device__ inline void te_wgrad_split(int slots, int nui, int jobs, int P, int T, int* n_o, int* n_u, bool xcd = true) {
  float rho = (T > 0 && P > 0) ? (float)P / (float)T : 1.f;
  rho = rho < 0.02f ? 0.02f : rho > 1.f ? 1.f : rho;
  int o = (int)((float)slots / ((float)nui * rho + (float)(jobs - nui)));
  if (o < 1) o = 1;
  int u = (int)((float)o * rho + 0.5f);
  if (u < 1) u = 1;
  while (o > 1 && nui * u + (jobs - nui) * o > slots) --o;
  // A chunk count that is a multiple of 8 puts the T-row jobs of one K-chunk on ONE XCD (workgroup ids go round the 8 XCDs and the
  // jobs of chunk kc sit n_o ids apart): they stream the same h / r*h rows at the same time, so four of the five reads of every H row hit
  // that XCD's L2 instead of HBM.  The slots this frees go to the d ui jobs.
  // (Only where it costs little: at >= 64 chunks the rounding lengthens a chunk by < 11 %; the d ui jobs never get more chunks than
  // the T-row jobs, which is what the slabs are sized for.)
  if (xcd && o >= 64) {
    o &= ~7;
    const int uu = (slots - (jobs - nui) * o) / (nui > 0 ? nui : 1);
    if (nui > 0 && uu > u) u = uu < 2 * u ? uu : 2 * u;
    if (u > o) u = o;
    if (u >= 16) u &= ~7;          // (the d ui jobs of one chunk gather the same POI rows: same XCD for them too)
  }
  *n_o = o; *n_u = u;
}

// Layout of one dense-gradient slab (offsets in floats).
struct DenseLayout { int ui, wh, bi, vs, bs, wd, sur, upq, total; };
__host__ __device__ inline DenseLayout dense_layout(int D, int XW, int NB) {
  DenseLayout l;
  l.ui = 0;
  l.wh = l.ui + 3 * D * XW;
  l.bi = l.wh + 3 * D * D;
  l.vs = l.bi + 3 * D;
  l.bs = l.vs + NB * D;
  l.wd = l.bs + NB;
  l.sur = l.wd + 1;
  l.upq = l.sur + 1;
  l.total = (l.upq + 1 + 3) & ~3;
  return l;
}

// Arguments of the tile engine (tile_engine.hip).
struct TeArgs {
  float *lt, *di, *ui, *wh, *bi, *vs, *bs, *wd, *lw;
  int n_item, n_dist, dim;
  const int *off, *p, *q, *dp, *dq;
  int len_max;
  const int* uidx;
  int n_seq;
  float* out;
  int predict;                        // 1: forward over all L positions, no bookkeeping
  int spatial, xw;                    // 1 / 2D: Distance2Pre (POI + distance-bin input); 0 / D: plain GRU + BPR (n_dist == -1)
  int lt_f16;                         // the POI table `lt` is stored as IEEE half (float32 arithmetic): config X
  int rec32;                          // streaming recurrent kernels on 32-sequence tiles (D = 256; D = 128 on request)
  int bintab;                         // spatial && D >= 128: distance-bin half through per-bin tables (te_ztab / te_dsum)
  float *ztab, *dpart, *dsum, *dgd;   // (n_dist+1) x 3D table; per-chunk partial sums of DA; per-bin sums; per-bin d di sums
  int *dch0, *dch1;                   // first 64-entry chunk / first super-chunk of each bin (+ total)
  // hot bins (round 5): on check-in data a few distance bins hold most steps (short hops) - te_psum, which streams every DA row once for the per-POI
  // sums anyway, also adds the rows of the (<= TE_HB) most frequent bins into per-workgroup partials that take the place of those bins' 64-entry chunk
  // partials (dpart); te_dsum then reads only the rows of the remaining bins.  dhot[0] = number of hot bins, dhot[1 .. TE_HB] = their ids,
  // dhot[8 + b] = hot index of bin b or -1 (te_dprep); npw = workgroups of te_psum (= chunk partials of a hot bin); dhot_on: host switch
  int* dhot; int npw, dhot_on;
  int* dcc0;                          // exclusive scan of the COLD bins' chunk counts (+ total): te_dsum's work list
  float* dpart2; int *dnf, *dnf2, *dbn;   // super-chunk partial sums; distinct-sequence counts per chunk / super-chunk / bin
  int dbg;                            // tuning switch (POI_TE_DBG), 0 in production
  int early_bins; float bin_alpha, bin_lambda;      // the distance-bin chain of the write-back starts on the side stream behind te_wgrad (launch_te_train)
  // packed-row workspace
  int *soff, *row_src, *row_t, *row_p, *row_dp, *row_ab;   // packed row -> CSR position, step index, input table rows (lt, di)
  float *X, *E, *G, *H, *RH, *DH, *DL, *rowloss;   // DL: d logits (T x padded bins)
  float4 *pVsT, *pVs;
  // forward table (te_rec_fwd16<FT>): ptab = lt . ui[:, :D]^T over all n_item + 1 table rows; iota = 0..n_item, then n_item + 1
  float* ptab; const int* iota; int fwd_tab; float* uiP;       // uiP: ui's POI half with gate-interleaved rows
  float4* pUiP3;                      // uiP as bf16 x 3 fragments (te_pack n16 == 4): B operand of te_ptab_s3 (the forward table on split products)
  float* uiT;                         // ui transposed (2D x 3D), K-contiguous B operand of te_gemm_dx
  float4 *pWhT16, *pWhc16, *pWhzr16;  // 16-column fragments (16x16x4 MFMA) for the recurrent kernels
  int rec_split;                      // recurrent kernels on bf16 x 3 split operands (te_rec_fwd16 / bwd16 <SP>)
  int head_split;                     // chunked head (more than 256 bins) on split products: pVsT / pVs hold bf16 x 3 fragments
  int rec1;                           // per-sequence recurrent kernels on the vector ALUs (te_rec_fwd1 / bwd1): small launches
  float* slab;
  int n_slab, n_head, n_kc;
  // the slot sort only feeds te_scatter (the last kernel): it runs on a side stream next to the GEMMs
  hipStream_t side; hipEvent_t ev_slots, ev_sorted;       // side == nullptr: inline on the main stream
  hipEvent_t ev_start, ev_pack;                           // the weight packs of a launch run on the side stream next to its index preparation
  hipEvent_t ev_bwd, ev_fin;                              // te_finalize / te_parts run on the side stream next to te_wgrad / te_gemm_dx
  // (round 6) hot_early: the hot rows' g*h chunk sums (te_hot_reduce: needs the sorted entries, gcoef and H - nothing later) run on the side stream right behind te_head,
  // beside te_rec_bwd, instead of in the tail; the hot-row list is built behind te_segment (te_hotlist) instead of by te_reduce.  Per-POI regrouping only (no per-entry dx rows)
  int hot_early; hipEvent_t ev_hr0, ev_hr1;
  // (round 6) HYBRID recurrences of mid-size launches (hyb != 0; poi_ctx option "hybrid"): a 1563-sequence launch is 98 tiles of 16 - its two recurrences
  // are the 49-step chains of the longest tile on 98 CUs while 158 CUs idle.  The leading hyb_dev[0] (forward) / hyb_dev[2] (backward) sequences of the
  // launch - the longest ones of a length-sorted launch - run on the per-sequence kernels (te_rec_fwd1x / te_rec_bwd1: 2.0 / 1.0 us per step, persistent
  // grids of hyb_dev[1] / hyb_dev[3] workgroups on stream side2) WHILE the shorter rest runs in 16-sequence tiles (4.5 / 2.9 us per step) on the other CUs;
  // the split is chosen per launch ON THE DEVICE from the launch's own lengths (te_hybrid_kernel: min over the tile boundaries of max(chain of the
  // tiles' longest sequence, steps of the leading sequences / free CUs)).  pWhc1 / pWhzr1: the plain transposes te_rec_bwd1 reads (its own buffers: the
  // tile kernel's bf16 x 3 fragments live in pWhc16 / pWhzr16 at the same time)
  int hyb, hyb_force; int* hyb_dev; float4 *pWhc1, *pWhzr1;      // hyb_force > 0 (option "hybrid_force", tests): that many leading sequences, whatever the cost model says
  hipStream_t side2; hipEvent_t ev_h0, ev_h1, ev_h2, ev_h3;      // (forward fork / join, backward fork / join: an event is never re-recorded inside a launch)
  float *bi_part, *fin_part;           // per recurrent tile d bi partials (n_tile x 3D); per te_finalize block loss sums
  float* hslab; int hstride;          // te_head's per-workgroup d bs | d wd partials (n_head x hstride)
  DenseLayout dl;
  int *mult_lt, *nseq_lt, *mult_di, *nseq_di;    // only the padding rows' entries are used (analytic touches)
  float *hts, *sts; const int* out_row;     // predict: row k of the launch -> output row out_row[k] (null: k)
  // sorted segmented scatter (te_scatter.hip): every table touch of the launch becomes one (row key,
  // entry) pair; a stable radix sort groups them by row, and each row's update is a plain ordered sum
  int key_bits;                       // bits of the largest key (= sentinel = number of table rows)
  int *keys0, *keys1, *vals0, *vals1; // sort ping-pong (key = unified row id, value = slot)
  int *code, *slot_seq;               // per slot: entry code, sequence index in the launch
  int *ent;                           // sorted entry codes (bit 31: first entry of its sequence in the row)
  int *seg_start, *seg_end;           // per unified row: [start, end) in `ent`; end == 0 <=> untouched (persistent, re-zeroed)
  int *hist;                          // radix histogram (bins x blocks)
  int *cnt;                           // [0] number of slots, [1] hot rows, [2] hot chunks, [3] touched rows (urow), [4] S rows, [5] dx entries of the POI rows
  int *urow;                          // list of touched table rows (null: te_reduce scans the tables), see te_segment
  int4* hot_rows;                     // {row, start, count, first chunk}
  int2* hot_chunks;                   // {hot row index, chunk index}
  float* hot_part;                    // (hot chunks, D) partial sums
  int* hot_nf;                        // per hot chunk: distinct-sequence count
  float* gcoef;                       // per packed row: d loss / d (h . e) (te_head)
  float bcap;                         // batch rule cap (see SeqArgs)
  // per-POI regrouping (ppoi; bintab only): the step input lt[p_t] takes far fewer distinct values than there are steps (231 k steps hit
  // ~45 k POIs per launch), and everything the backward pass needs from it is linear in S[p] = sum over the steps with p_t = p of DA_t:
  //   d lt[p] (dx part) = S[p] . ui[:, :D]   (te_gemm_dx over S rows)      d ui[:, :D] = S^T . lt[rows of S]   (te_wgrad, K = S rows)
  int ppoi;
  int *pmark;                         // per lt row: has a dx entry in this launch (zero between launches; te_slots sets, the write-back clears)
  int *urow_p;                        // S row -> lt row (lt row -> S row + 1 is pmark itself after te_passign)
  int *pblk;                          // per-block counts of the scan (S rows | dx entries)
  const int *ks;                      // sorted keys (set by launch_te_sort)
  int *dxe, *dxs, *dstart;            // compacted dx entries of the POI rows: packed step row, S row; first list position of every S row (+ end)
  float *S, *pfirst, *plast;          // S (rows x 3D); per 64-entry range: sum of its opening / closing run
  int wg_slots;                       // te_wgrad: workgroups of the (job, K-chunk) grid (2 per CU)
  int* kc_dev;                        // {K-chunks of the non-ui jobs, of the d ui jobs}: chosen ON THE DEVICE from the launch's own S-row / step counts
                                      // (te_wgrad_split) - the split-K order is a function of the launch alone, not of its history
  unsigned sr_salt;                   // != 0: a half POI table is written back with stochastic rounding (poi_ctx_set_f16_rounding), salt of this launch
  const float* zrow;                  // resident all-zero row (>= 2 * dim floats), never written: target of branch-free "no contribution" loads
  // exact forward (te_xfwd.hip): input product + forward recurrence in ~40-bit fixed point on the int8 matrix cores, float64 gate math
  int xcomp; int *xidx, *xlist, *xblk, *xcnt, *row_pc;      // exact forward table over the launch's step-input POIs only (te_xcount / te_xassign): lt row -> table row, table row -> lt row, per-block counts, row count, per-step table row
  int efuse;                          // (round 6) te_head3 gathers E = lt[p'] - lt[q'] itself (row_pq: the step's positive / negative POI): no E rows in HBM, te_gather only translates ids; dim 128, training, <= 256 bins; POI_TE_EFUSE
  int2* row_pq;
  int xrec1;                          // the recurrence of every sequence in its own workgroup on the float64 vector ALUs (te_rec_fwd1x): small launches
  int xfwd, xft;                      // on; pre-activations from the forward table ptabx[p_t] + ztabx[dp_t] (else gx[row])
  double *gx, *ptabx, *ztabx;         // (T + spare) x 3D per step | (n_item + 1 + spare) x 3D | (n_dist + 1) x 3D, gate-major columns (g D + unit)
  uint4 *xWh8, *xUi8; double *xWhS, *xUiS;      // digit fragments + row scales of wh (16x16x64 order) and of ui's POI half (32x32x32 order)
  int x_rows_est;                     // host-side bound of the packed row count (grid sizing of te_gemmx)
  int* xflag; int xlaunch;            // != launch id: no non-finite weight / input row seen while this launch's operands were prepared (te_xfwd.hip x_flag_bad)
};
// entry code: packed-row index of the position (28 bits) + what the position contributes
#define TE_ENT_ROW 0x0FFFFFFF
#define TE_ENT_DX 0x10000000      // + dx[row] (lt half for POI rows, di half for distance-bin rows)
#define TE_ENT_GH 0x20000000      // + g[row-1] * h[row-1]
#define TE_ENT_NEG 0x40000000     // the g*h term enters with a minus sign (negative sample)
#define TE_ENT_FIRST 0x80000000u  // first entry of its sequence within the row segment
#define TE_HYB_NMAX 4096          // hybrid recurrences: te_scan keeps the launch's step counts in LDS for the split
#define TE_PSUM_WG 1024          // workgroups of te_psum = chunk partials of a hot bin: device-independent (was num_cu * 4)
#define TE_HB 4                   // hot distance bins summed by te_psum (TeArgs.dhot)
#define TE_HOT_BIN_MIN 4096       // ... when they hold at least this many steps of the launch
#define TE_COLD_MAX 64            // rows with more entries are reduced in 256-entry chunks by whole workgroups
#define TE_HOT_CHUNK 64
#define RS_MAXBIN 512
#define RS_GRID 256
#define RS_HIST_INTS (RS_MAXBIN * RS_GRID)   // radix histogram: bins x blocks
bool te_supported(int D, int n_dist);
bool te_bintab(int D, bool spatial, int n_dist);
// distance bins padded for the head / d vs tiles: 32 x {1, 2, 4, 7, 8} up to 256 bins, a multiple of 256 beyond (te_head_big's chunks)
__host__ __device__ inline int te_nbp_dev(int n_dist) {
  const int t = (n_dist + 1 + 31) / 32;
  return t <= 8 ? 32 * (t <= 1 ? 1 : t <= 2 ? 2 : t <= 4 ? 4 : t <= 7 ? 7 : 8) : 256 * ((n_dist + 1 + 255) / 256);
}
int te_wgrad_jobs(int D, int n_dist, bool spatial, bool bintab);
hipError_t launch_te_sort(TeArgs& A, hipStream_t st);
hipError_t launch_te_psum(TeArgs& A, int num_cu, hipStream_t st);
hipError_t launch_te_passign(TeArgs& A, hipStream_t st);
hipError_t launch_te_hot_reduce(TeArgs& A, int num_cu, hipStream_t st);      // chunk sums of the hot rows (te_scatter.hip); early on the side stream when TeArgs.hot_early
hipError_t launch_te_dprep(TeArgs& A, hipStream_t st);        // chunk offsets of the distance-bin chain (behind the slot sort; launch_te_bins expects them)      // S rows of the per-POI regrouping (behind the slot sort)
int te_wgrad_ui_jobs(int D, int n_dist, bool spatial, bool bintab);
hipError_t launch_te_scatter(TeArgs& A, float alpha, float lambda, int num_cu, hipStream_t st, Timing* tm);
hipError_t launch_te_bins(TeArgs& A, float alpha, float lambda, int num_cu, hipStream_t sb, Timing* tm);      // te_scatter.hip: per-bin sums of DA -> d di, d ui[:, D:]
int te_nbp(int n_dist);
void launch_te_iota(int* buf, int n, hipStream_t st);
hipError_t launch_te_one(TeArgs& A, float alpha, float lambda, int l_cap, hipStream_t st, Timing* tm);      // n_seq == 1, whole step
bool te_one_supported(int D, bool spatial, int max_len);
hipError_t launch_te_train(TeArgs& A, int num_cu, hipStream_t st, Timing* tm);
hipError_t launch_te_predict(TeArgs& A, int num_cu, hipStream_t st, Timing* tm);
bool te_xfwd_supported(int D);
hipError_t launch_te_xfwd(const TeArgs& A, int num_cu, hipStream_t st, Timing* tm, int phase);      // te_xfwd.hip: phase 0 = input product, 1 = recurrence
hipError_t launch_rows_apply(const SeqArgs& A, bool spatial, int grid, float alpha, float lambda, hipStream_t st, Timing* tm);
hipError_t launch_dense_apply(const SeqArgs& A, bool spatial, int n_slab, int n_slab_head, float alpha, float lambda, hipStream_t st, Timing* tm);

size_t seq_ws_floats(int D, int NB, int cap);

// Arguments of the exact (float64) engine, exact_engine.hip: SeqArgs with float64 scratch.  Tables stay float32.
struct ExArgs {
  float *lt, *di, *ui, *wh, *bi, *vs, *bs, *wd, *lw;
  int n_item, n_dist, dim;
  const int *off, *p, *q, *dp, *dq;
  int len_max, cap;
  const int* uidx;
  int n_seq;
  float* out;
  double* ws; size_t ws_stride;     // per-workgroup activations (float64)
  double* slab;                     // per-workgroup dense-gradient slabs (float64)
  double *g_lt, *g_di;              // zero-initialised float64 gradient tables
  int *mult_lt, *nseq_lt, *mult_di, *nseq_di;
  float *hts, *sts; const int* out_row;
  float bcap;
};
size_t ex_ws_doubles(int D, int NB, int cap);
hipError_t launch_ex_train(const ExArgs& A, bool spatial, int grid, double alpha, double lambda, hipStream_t st, Timing* tm);
hipError_t launch_ex_predict(const ExArgs& A, bool spatial, int grid, hipStream_t st, Timing* tm);
hipError_t launch_seq_train(const SeqArgs& A, bool spatial, int grid, float alpha, float lambda, hipStream_t st, Timing* tm);
hipError_t launch_seq_predict(const SeqArgs& A, bool spatial, int grid, hipStream_t st, Timing* tm);

// CA-RNN (carnn.hip)
struct CaArgs {
  float *lt, *wd, *M;                 // POI table (n_item+1, D), interval matrices (n_dist+1, H, D), input matrix (H, D); H == D
  int n_item, n_dist, dim;
  const int *off, *p, *q, *dp, *dq;
  int len_max, cap;
  const int* uidx;
  int n_seq;
  float* out;                         // train: loss per sequence
  float* ws; size_t ws_stride;        // per-workgroup activations
  float* slab;                        // per-workgroup d M
  float *g_lt, *g_wd;                 // zero-initialised gradient tables
  int *mult_lt, *nseq_lt, *mult_wd, *nseq_wd;
  float* hts;                         // predict output
  float bcap;
  // ---- outer-product path (carnn_train2: dims 64 / 128): packed per-step state and the (matrix id, a, b) entries whose sorted,
  // segmented products a (x) b are the gradients of the interval matrices and of M (no float atomics on matrices)
  int* soff;                          // n + 1: packed step offsets (step t of sequence k = row soff[k] + t)
  float* Hpk;                         // (T + n) x D: h_0 .. h_ns of sequence k at rows soff[k] + k + t
  float* EA;                          // T x 5 x D: g mp | -g mq | da | g vp | -g vq of every step
  int *keys0, *keys1, *vals0, *vals1, *hist, *cnt;      // radix sort of the 6 T entries by matrix id; cnt[0] = 6 T
  int *ent_a, *ent_b;                 // per entry: EA vector (row of D floats), b source (>= 0: Hpk row, < 0: ~row of lt)
  int *seg_start, *seg_end, *chunk_first;               // per matrix id (n_dist + 2 ids: the interval matrices, then M)
  float* partial;                     // per 512-entry chunk: D x D partial product
  int n_chunk_cap;
  float* PM;                          // (n_item + 1) x D: lt . M^T, the input product of every POI (computed once per launch)
  // gradient of the POI rows: d lt[row] = M^T (sum of the step vectors that name the row) - three entries (row, EA vector) per step,
  // sorted by row; 64-entry windows of the sorted list are summed per run (vpart), a wave per row adds its runs in order
  int *k2a, *k2b, *v2a, *v2b;         // radix sort of the 3 T row entries (cnt[1] = 3 T)
  int *seg2_start, *seg2_end;         // per POI row (n_item + 2)
  float* vpart;                       // 3 T x D: partial sums, stored at the sorted position of a run's first entry inside its window
};
size_t carnn_ws_floats(int D, int cap);
hipError_t launch_carnn_train2(const CaArgs& A, int grid, float alpha, float lambda, hipStream_t st, Timing* tm);
// stable LSD radix sort of (keys, element index) pairs on `bits` key bits; n_ptr[0] = element count (device).  *ks / *vs: the buffers
// that hold the sorted keys / the original indices in sorted order.  hist: RS_HIST_INTS + RS_MAXBIN ints.
hipError_t launch_radix_sort(int* keys0, int* keys1, int* vals0, int* vals1, const int* n_ptr, int bits, int* hist, hipStream_t st,
                             const int** ks, const int** vs);
hipError_t launch_carnn_train(const CaArgs& A, int grid, float alpha, float lambda, hipStream_t st, Timing* tm);
hipError_t launch_carnn_predict(const CaArgs& A, int grid, float* wrs, hipStream_t st, Timing* tm);
hipError_t launch_carnn_score(const float* users, const float* items, const float* M, const float* dists, const double* coords, const double* cphi,
                              const double* thr, const int* last_poi, int n, int N, int n_dist, int D, double dd, float* scratch, float* out,
                              hipStream_t st, Timing* tm);

// BPR-MF
struct BprArgs {
  float* ux; void* lt;    // lt: float32, or IEEE half when lt_f16 (float32 arithmetic; config X's table storage)
  int lt_f16; unsigned sr_salt;      // != 0: stochastic rounding of the half write-back (poi_ctx_set_f16_rounding)
  int n_user, n_item, dim;
  const int *uidx, *p, *q;
  int n;
  float alpha, lambda;
  float* loss;
  int* bad;               // device counter of out-of-range ids (poi_ctx_take_bad_ids)
  float bcap;             // batch rule cap (see SeqArgs)
  // snapshot mode (bpr.hip): 3 n table touches sorted by row
  int *keys0, *keys1, *vals0, *vals1, *hist, *cnt;
  const int *ks, *vs;     // sorted keys / touch ids (set by launch_bpr)
  int4* meta;             // per 64-touch window: {touches of its opening run, that run goes on, touches of its closing run, its row}
  float *g;               // per triple: d loss / d u
  float *shadow;          // (n_user, D): new user rows until the POI pass has read the entry values
  float *lead, *trail;    // per window: partial sum of its opening / closing run
};
hipError_t launch_bpr(BprArgs& A, int mode, int num_cu, hipStream_t st, Timing* tm);
void bpr_ws_sizes(int n, int dim, size_t* n_int, size_t* n_float);

// scoring / top-K
struct ScoreArgs {
  const float *users, *items; int items_f16;      // items: float32, or IEEE half when items_f16
  float4* items_packed;     // MFMA B-fragment order (large-n kernel), ctx scratch
  int n, n_item, dim;
  const float *wd, *prob;
  // distance term through the resident bin matrix (ulptai_kernel layout) instead of `prob`:
  // score += wd * (bin < n_dist ? sts[user][bin] : 0)
  const void* ulptai; int bin_bytes; const float* sts; int n_dist;
  // ... or with the bins computed on the fly from the coordinates (no U x N matrix at all): geo != 0
  int geo; const double *coords, *cphi, *thr; const int* last_poi; double dd;
  float* scores;            // (n, n_item) or null
  int k;                    // 0 = no top-K
  int n_split;              // item splits for the fused top-K
  float* cand_score; int* cand_idx;   // (n_split, n_pad, k) partial top-K lists
  int dbg;                  // tuning switch (POI_SCORE_DBG), 0 in production
  int max_stride;           // score_filter_kernel<MAXP>: every max_stride-th item tile
  unsigned* gbound;         // (n_pad) per-user lower bound of the K-th best score shared by all item ranges
  int seeded;               // gbound starts from the caller's seed items' scores (topk_seed_kernel), not from zero
  int* idx_out; float* score_out;
  // two-stage path (score_filter.hip): half-precision item fragments + per-item norms of the bound, per-user survivor lists, and the
  // per-user-tile overflow flags; the one-stage kernels and the merge skip every tile whose flag is clear when tile_flag is set
  const uint4* items_packed16; const float2* inorm;
  int* surv_cnt; int* surv_idx; float* surv_sc; int* tile_flag;
  // item-stationary GEO filter (score_filter_items_kernel): the users' half fragments, per-user bound terms and last-POI coordinates
  // (ctx scratch; users_packed16 == nullptr: the user-stationary filter), and the CU count for its persistent grid
  uint4* users_packed16; float4* ubound; double* ugeo; int n_cu;
  int bins_ntile;           // item tiles per user-tile row of the bin matrix when the call scores only a PREFIX of the item table (0: = tiles of n_item)
};
bool score_two_stage_supported(const ScoreArgs& A);
hipError_t launch_score_two_stage(const ScoreArgs& A, int n_split_f, hipStream_t st, Timing* tm);
bool score_maxpass_supported(const ScoreArgs& A);
hipError_t launch_score_maxpass(const ScoreArgs& A, int n_split_f, hipStream_t st, Timing* tm);      // unseeded calls: thresholds from block maxima of the f16 lower bounds
int score_filter_cap();
hipError_t launch_topk_bound(const float* score_k, int n, int k, unsigned* gbound, hipStream_t st);
hipError_t launch_ulptai(const double* coords, const double* cphi, const double* thr, const int* last_poi, int n, int n_item,
                         int n_dist, double dd, void* out, int bin_bytes, hipStream_t st);
hipError_t launch_score(const ScoreArgs& A, hipStream_t st, Timing* tm);
hipError_t launch_score_packed(const ScoreArgs& A, hipStream_t st, Timing* tm);
hipError_t launch_topk_seed(const ScoreArgs& A, const int* seed, int k_seed, hipStream_t st);
hipError_t launch_score_geo_stream(const ScoreArgs& A, hipStream_t st, Timing* tm);
size_t score_geo_stream_lds(int dim, int n_dist);
hipError_t launch_topk_merge(const ScoreArgs& A, int n_lists, int n_pad, hipStream_t st);
hipError_t launch_topk_rows(const float* scores, int n, int n_item, int k, int* idx_out, float* score_out, hipStream_t st);

// misc
hipError_t launch_auc(const float* users, const float* items, int items_f16, int n, int dim, const int* tp, const int* tq,
                      const int* tm, int len, uint8_t* out, hipStream_t st);
hipError_t launch_sumsq_f16(const void* x, int64_t n, double* out, hipStream_t st);
hipError_t launch_sumsq(const float* x, int64_t n, double* out, hipStream_t st);
hipError_t launch_dist_prob(const double* coords, const double* cphi, const double* thr, const int* last_poi, const float* sts,
                            int n, int n_item, int n_dist, double dd, float* prob, hipStream_t st);
hipError_t launch_rank_metrics(const int* ranks, int n, int K, const int* tes_p, const int* tes_mask, int len_tes, const int* at_nums,
                               int n_at, double* acc, hipStream_t st);
hipError_t launch_sample_neg(const int* off, const int* p, int n_user, int n_item, const int* tes_p, const int* tes_mask, int len_tes,
                             unsigned long long seed, int* q_out, int* tes_q_out, hipStream_t st);
hipError_t launch_neg_dist(const int* off, const int* p, const int* q, int n_user, const double* coords, const double* cphi,
                           const double* thr, int n_dist, double dd, int* dq, hipStream_t st);
hipError_t launch_delta_make(const float* cur, const float* base, float* delta, int64_t n, hipStream_t st);
hipError_t launch_delta_apply(float* cur, const float* base, const float* dsum, int64_t n, hipStream_t st);
hipError_t launch_selftest(float* buf, int* fail, hipStream_t st);

}  // namespace poi
