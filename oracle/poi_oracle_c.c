/* Plain-C float64 restatement of the reference's per-sequence training step and all-POI top-K.
 * TEST / BASELINE INFRASTRUCTURE ONLY: linked by nothing in the product; used by
 *   - tests/test_oracle_c.py  (checked against oracle/poi_oracle.py, which is checked against autograd),
 *   - bench.py's cpu_baseline leg ("kind": "port" - Theano itself cannot be built or shipped).
 * Follows public/GRU_Spatial.py:127-229 (one seq_train(uidx) call == one poi_oracle_spatial_seq call,
 * applied in place like Theano's `updates`) and public/Valuate.py:91-100 (top-K).
 * Single-threaded by construction: the reference's CPU path is one compiled graph per call on one
 * thread, with every step's dense update feeding the next (prog_bpr_gru_spatial.py:249-250).
 * Parity status: see oracle/poi_oracle.py header (Theano half unpinned).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

static double sigm(double x) { return 1.0 / (1.0 + exp(-x)); }
static double logsig(double x) { return x >= 0 ? -log1p(exp(-x)) : x - log1p(exp(x)); }

/* y[r] = sum_j W[r*K + j] x[j] */
static void gemv(const double* W, int rows, int K, const double* x, double* y, int accumulate) {
  for (int r = 0; r < rows; ++r) {
    const double* w = W + (size_t)r * K;
    double s = 0;
    for (int j = 0; j < K; ++j) s += w[j] * x[j];
    y[r] = accumulate ? y[r] + s : s;
  }
}
/* y[j] += sum_r W[r*K + j] v[r] */
static void gemvT_acc(const double* W, int rows, int K, const double* v, double* y) {
  for (int r = 0; r < rows; ++r) {
    const double* w = W + (size_t)r * K;
    const double vr = v[r];
    for (int j = 0; j < K; ++j) y[j] += w[j] * vr;
  }
}
/* G[r*K + j] += a[r] b[j] */
static void outer_acc(double* G, int rows, int K, const double* a, const double* b) {
  for (int r = 0; r < rows; ++r) {
    double* g = G + (size_t)r * K;
    const double ar = a[r];
    for (int j = 0; j < K; ++j) g[j] += ar * b[j];
  }
}

/* Gradients of one Distance2Pre step on sequence (p,q,dp,dq)[0..L), evaluated at the given parameter
 * values (nothing is modified).  Dense gradients (without the L2 term) go to gd = [ui | wh | bi | vs | bs],
 * position gradients to gp / gq / gdd ((L+1) x D each: the row gradient contributed at position t of p / q /
 * dp, duplicates not yet merged), scalars to sc = {sur, upq, g_wd, ls0, ls1}. */
typedef struct { double *gd, *gp, *gq, *gdd; size_t nd; double sur, upq, g_wd, ls0, ls1; } seq_grad;

static void seq_grad_free(seq_grad* G) { free(G->gd); free(G->gp); }

static void spatial_seq_grad(const double* lt, const double* di, const double* ui, const double* wh, const double* bi,
                             const double* vs, const double* bs, double wd, const double* lw, int n_dist, int D,
                             const int* p, const int* q, const int* dp, const int* dq, int L, seq_grad* G) {
  const int NB = n_dist + 1, XW = 2 * D, ns = L > 0 ? L - 1 : 0;
  double m = lw[0] > lw[1] ? lw[0] : lw[1];
  double e0 = exp(lw[0] - m), e1 = exp(lw[1] - m);
  const double ls0 = e0 / (e0 + e1), ls1 = e1 / (e0 + e1);
  size_t nd = (size_t)3 * D * XW + (size_t)3 * D * D + 3 * D + (size_t)NB * D + NB;
  double* H = calloc((size_t)(ns + 1) * D, sizeof(double));
  double* Z = calloc((size_t)(ns + 1) * D * 3, sizeof(double));   /* z | r | c per step */
  double* S = calloc((size_t)(ns + 1) * NB, sizeof(double));
  double* U = calloc((size_t)ns + 1, sizeof(double));
  double* X = calloc((size_t)(ns + 1) * XW, sizeof(double));
  double* gd = calloc(nd, sizeof(double));
  double *g_ui = gd, *g_wh = g_ui + (size_t)3 * D * XW, *g_bi = g_wh + (size_t)3 * D * D, *g_vs = g_bi + 3 * D, *g_bs = g_vs + (size_t)NB * D;
  /* row gradients per position: gp[t], gq[t], gdp[t] (scattered with duplicates merged at the end) */
  double* gp = calloc((size_t)(L + 1) * D * 3, sizeof(double));
  double *gq = gp + (size_t)(L + 1) * D, *gdd = gq + (size_t)(L + 1) * D;
  double* tmp = calloc((size_t)8 * D + 2 * XW + 2 * NB, sizeof(double));
  double *a = tmp, *rh = a + 3 * D, *dh = rh + D, *dhn = dh + D, *da = dhn + D /*3D*/, *dx = da + 3 * D, *o = dx + XW, *ds = o + NB;
  double sur = 0, bpr = 0, g_wd = 0;
  for (int t = 0; t < ns; ++t) {
    double* x = X + (size_t)t * XW;
    memcpy(x, lt + (size_t)p[t] * D, D * sizeof(double));
    memcpy(x + D, di + (size_t)dp[t] * D, D * sizeof(double));
    const double* hp = H + (size_t)t * D;
    double* h = H + (size_t)(t + 1) * D;
    double *z = Z + (size_t)t * 3 * D, *r = z + D, *c = r + D;
    gemv(ui, 2 * D, XW, x, a, 0); gemv(wh, 2 * D, D, hp, a, 1);
    for (int j = 0; j < D; ++j) { z[j] = sigm(a[j] + bi[j]); r[j] = sigm(a[D + j] + bi[D + j]); rh[j] = r[j] * hp[j]; }
    gemv(ui + (size_t)2 * D * XW, D, XW, x, a, 0); gemv(wh + (size_t)2 * D * D, D, D, rh, a, 1);
    for (int j = 0; j < D; ++j) { c[j] = tanh(a[j] + bi[2 * D + j]); h[j] = (1.0 - z[j]) * hp[j] + z[j] * c[j]; }
    double* s = S + (size_t)t * NB;
    gemv(vs, NB, D, h, s, 0);
    double mx = -INFINITY, sum = 0;
    for (int k = 0; k < NB; ++k) { s[k] += bs[k]; if (s[k] > mx) mx = s[k]; }
    for (int k = 0; k < NB; ++k) { s[k] = exp(s[k] - mx); sum += s[k]; }
    for (int k = 0; k < NB; ++k) s[k] /= sum;
    const int ai = dp[t + 1], bq = dq[t + 1];
    const double *xp1 = lt + (size_t)p[t + 1] * D, *xq1 = lt + (size_t)q[t + 1] * D;
    double u = 0;
    for (int j = 0; j < D; ++j) u += h[j] * (xp1[j] - xq1[j]);
    u += wd * (s[ai] - s[bq]);
    U[t] = u;
    bpr += logsig(u);
    double cs = 0;
    for (int k = 0; k <= ai; ++k) cs += s[k];
    sur += cs - log(s[ai]);
  }
  const double upq = -bpr;
  memset(dhn, 0, D * sizeof(double));
  for (int t = ns - 1; t >= 0; --t) {
    const double *x = X + (size_t)t * XW, *hp = H + (size_t)t * D, *h = H + (size_t)(t + 1) * D;
    const double *z = Z + (size_t)t * 3 * D, *r = z + D, *c = r + D, *s = S + (size_t)t * NB;
    const int ai = dp[t + 1], bq = dq[t + 1];
    const double g = -ls1 * sigm(-U[t]);
    const double *xp1 = lt + (size_t)p[t + 1] * D, *xq1 = lt + (size_t)q[t + 1] * D;
    for (int j = 0; j < D; ++j) {
      dh[j] = dhn[j] + g * (xp1[j] - xq1[j]);
      gp[(size_t)(t + 1) * D + j] += g * h[j];
      gq[(size_t)(t + 1) * D + j] -= g * h[j];
    }
    g_wd += g * (s[ai] - s[bq]);
    double dot = 0;
    for (int k = 0; k < NB; ++k) { ds[k] = (k <= ai ? ls0 : 0.0); }
    ds[ai] += g * wd - ls0 / s[ai];
    ds[bq] -= g * wd;
    for (int k = 0; k < NB; ++k) dot += ds[k] * s[k];
    for (int k = 0; k < NB; ++k) { o[k] = s[k] * (ds[k] - dot); g_bs[k] += o[k]; }
    outer_acc(g_vs, NB, D, o, h);
    gemvT_acc(vs, NB, D, o, dh);
    for (int j = 0; j < D; ++j) {
      const double dz = dh[j] * (c[j] - hp[j]);
      dhn[j] = dh[j] * (1.0 - z[j]);
      da[2 * D + j] = dh[j] * z[j] * (1.0 - c[j] * c[j]);
      da[j] = dz * z[j] * (1.0 - z[j]);
      rh[j] = 0;
    }
    gemvT_acc(wh + (size_t)2 * D * D, D, D, da + 2 * D, rh);      /* m = wh[2]^T da_c */
    for (int j = 0; j < D; ++j) {
      const double dr = rh[j] * hp[j];
      dhn[j] += rh[j] * r[j];
      da[D + j] = dr * r[j] * (1.0 - r[j]);
      rh[j] = r[j] * hp[j];
    }
    gemvT_acc(wh, 2 * D, D, da, dhn);
    memset(dx, 0, XW * sizeof(double));
    gemvT_acc(ui, 3 * D, XW, da, dx);
    outer_acc(g_ui, 3 * D, XW, da, x);
    outer_acc(g_wh, 2 * D, D, da, hp);
    outer_acc(g_wh + (size_t)2 * D * D, D, D, da + 2 * D, rh);
    for (int j = 0; j < 3 * D; ++j) g_bi[j] += da[j];
    for (int j = 0; j < D; ++j) { gp[(size_t)t * D + j] += dx[j]; gdd[(size_t)t * D + j] += dx[D + j]; }
  }
  free(H); free(Z); free(S); free(U); free(X); free(tmp);
  G->gd = gd; G->nd = nd; G->gp = gp; G->gq = gq; G->gdd = gdd;
  G->sur = sur; G->upq = upq; G->g_wd = g_wd; G->ls0 = ls0; G->ls1 = ls1;
}

/* The reference's write-back of one step (public/GRU_Spatial.py:149-153,210-215), expressed as DELTAS so that the
 * same code serves the sequential step (delta added to the parameters at once) and the batch rule (delta
 * accumulated per row together with a touch count).  For every unique row of p U q (and the padding row when
 * len_max > L) / of dp:  delta = -alpha * (sum of the row's position gradients + lam * multiplicity * row).
 * emit(ctx, table, row, delta[D]) is called once per unique row; table 0 = lt, 1 = di. */
typedef void (*row_emit)(void* ctx, int table, int row, const double* delta);

static void spatial_row_deltas(const double* lt, const double* di, int n_item, int n_dist, int D, const int* p, const int* q,
                               const int* dp, int L, int len_max, double alpha, double lam, const seq_grad* G,
                               row_emit emit, void* ctx) {
  const int ne = 2 * L;
  int* ids = malloc(sizeof(int) * (size_t)(ne + 1));
  char* done = calloc((size_t)ne + 1, 1);
  double* acc = malloc(sizeof(double) * D);
  int pad_done = 0;
  for (int e = 0; e < ne; ++e) ids[e] = e < L ? p[e] : q[e - L];
  for (int e = 0; e < ne; ++e) {
    if (done[e]) continue;
    const int row = ids[e];
    const double* tr = lt + (size_t)row * D;
    int mult = 0;
    memset(acc, 0, sizeof(double) * D);
    for (int f = e; f < ne; ++f) if (ids[f] == row) {
      done[f] = 1; ++mult;
      const double* gsrc = f < L ? G->gp + (size_t)f * D : G->gq + (size_t)(f - L) * D;
      for (int j = 0; j < D; ++j) acc[j] += gsrc[j];
    }
    if (row == n_item) { mult += 2 * (len_max - L); pad_done = 1; }      /* a literal padding id inside the sequence */
    for (int j = 0; j < D; ++j) acc[j] = -alpha * (acc[j] + lam * mult * tr[j]);
    emit(ctx, 0, row, acc);
  }
  if (!pad_done && len_max > L) {
    const double* tr = lt + (size_t)n_item * D; const double mm = 2.0 * (len_max - L);
    for (int j = 0; j < D; ++j) acc[j] = -alpha * lam * mm * tr[j];
    emit(ctx, 0, n_item, acc);
  }
  memset(done, 0, (size_t)ne + 1);
  pad_done = 0;
  for (int e = 0; e < L; ++e) {
    if (done[e]) continue;
    const int row = dp[e];
    const double* tr = di + (size_t)row * D;
    int mult = 0;
    memset(acc, 0, sizeof(double) * D);
    for (int f = e; f < L; ++f) if (dp[f] == row) { done[f] = 1; ++mult; for (int j = 0; j < D; ++j) acc[j] += G->gdd[(size_t)f * D + j]; }
    if (row == n_dist) { mult += len_max - L; pad_done = 1; }
    for (int j = 0; j < D; ++j) acc[j] = -alpha * (acc[j] + lam * mult * tr[j]);
    emit(ctx, 1, row, acc);
  }
  if (!pad_done && len_max > L) {
    const double* tr = di + (size_t)n_dist * D; const double mm = (double)(len_max - L);
    for (int j = 0; j < D; ++j) acc[j] = -alpha * lam * mm * tr[j];
    emit(ctx, 1, n_dist, acc);
  }
  free(ids); free(done); free(acc);
}

typedef struct { double *lt, *di; int D; } inplace_ctx;
static void emit_inplace(void* c, int table, int row, const double* d) {
  inplace_ctx* x = c;
  double* tr = (table ? x->di : x->lt) + (size_t)row * x->D;
  for (int j = 0; j < x->D; ++j) tr[j] += d[j];
}

/* One Distance2Pre SGD step on sequence (p,q,dp,dq)[0..L) with analytic padding to len_max.
 * Parameters are updated in place.  out5 = {los, sur, upq, ls0, ls1}. Returns 0. */
int poi_oracle_spatial_seq(double* lt, double* di, double* ui, double* wh, double* bi, double* vs, double* bs,
                           double* wd_p, double* lw, int n_item, int n_dist, int D,
                           const int* p, const int* q, const int* dp, const int* dq, int L, int len_max,
                           double alpha, double lam, double* out5) {
  const int NB = n_dist + 1, XW = 2 * D;
  const double wd = *wd_p;
  seq_grad G;
  spatial_seq_grad(lt, di, ui, wh, bi, vs, bs, wd, lw, n_dist, D, p, q, dp, dq, L, &G);
  const double ls0 = G.ls0, ls1 = G.ls1, sur = G.sur, upq = G.upq;
  out5[0] = ls0 * sur + ls1 * upq; out5[1] = sur; out5[2] = upq; out5[3] = ls0; out5[4] = ls1;
  double *g_ui = G.gd, *g_wh = g_ui + (size_t)3 * D * XW, *g_bi = g_wh + (size_t)3 * D * D, *g_vs = g_bi + 3 * D, *g_bs = g_vs + (size_t)NB * D;
  /* sparse updates first: they read the OLD table rows (Theano `updates`: everything at the old values), and the
   * dense tensors below are disjoint from the tables */
  {
    inplace_ctx c = {lt, di, D};
    /* deltas are computed from the old rows; a row is emitted once, so in-place addition is safe */
    spatial_row_deltas(lt, di, n_item, n_dist, D, p, q, dp, L, len_max, alpha, lam, &G, emit_inplace, &c);
  }
  /* dense updates (public/GRU_Spatial.py:210-211) */
  for (size_t i = 0; i < (size_t)3 * D * XW; ++i) ui[i] -= alpha * (g_ui[i] + lam * ui[i]);
  for (size_t i = 0; i < (size_t)3 * D * D; ++i) wh[i] -= alpha * (g_wh[i] + lam * wh[i]);
  for (int i = 0; i < 3 * D; ++i) bi[i] -= alpha * (g_bi[i] + lam * bi[i]);
  for (size_t i = 0; i < (size_t)NB * D; ++i) vs[i] -= alpha * (g_vs[i] + lam * vs[i]);
  for (int i = 0; i < NB; ++i) bs[i] -= alpha * (g_bs[i] + lam * bs[i]);
  *wd_p = wd - alpha * (G.g_wd + lam * wd);
  {
    const double d0 = sur + lam * ls0, d1 = upq + lam * ls1, dt = d0 * ls0 + d1 * ls1;
    lw[0] -= alpha * ls0 * (d0 - dt);
    lw[1] -= alpha * ls1 * (d1 - dt);
  }
  seq_grad_free(&G);
  return 0;
}

/* Batch rule of include/poi_hip.h ("throughput mode"), oracle side: every sequence's REFERENCE update is
 * evaluated at the same (unmodified) parameter values; per table row the deltas of the sequences that touch it
 * are SUMMED into acc_lt / acc_di with a touch count in cnt_lt / cnt_di, the dense deltas are summed into
 * acc_dense = [ui | wh | bi | vs | bs | wd | lw0 | lw1].  The caller divides by the counts (rows) / n (dense) -
 * possibly after adding up the accumulators of several threads, each of which ran a slice of the launch.
 * out5 (n x 5) receives the per-sequence losses.  Nothing in the parameter arrays is modified. */
typedef struct { double *lt, *di; int *clt, *cdi; int D; double *alt, *adi; } acc_ctx;
static void emit_acc(void* c, int table, int row, const double* d) {
  acc_ctx* x = c;
  double* tr = (table ? x->di : x->lt) + (size_t)row * x->D;
  for (int j = 0; j < x->D; ++j) tr[j] += d[j];
  (table ? x->cdi : x->clt)[row] += 1;
  double* ab = table ? x->adi : x->alt;
  if (ab) { ab += (size_t)row * x->D; for (int j = 0; j < x->D; ++j) ab[j] += fabs(d[j]); }
}

int poi_oracle_spatial_batch(const double* lt, const double* di, const double* ui, const double* wh, const double* bi,
                             const double* vs, const double* bs, const double* wd_p, const double* lw, int n_item, int n_dist, int D,
                             const int* off, const int* p, const int* q, const int* dp, const int* dq,
                             const int* ids, int n, int len_max, double alpha, double lam,
                             double* acc_lt, int* cnt_lt, double* acc_di, int* cnt_di, double* acc_dense, double* out5,
                             double* abs_lt, double* abs_di, double* abs_dense) {
  /* abs_* (optional, may be NULL): the same accumulators over |delta| - a row's "absolute mass", the scale float32
   * summation noise of a hot row is proportional to (tests/gpu_util.delta_excess) */
  const int NB = n_dist + 1, XW = 2 * D;
  const double wd = *wd_p;
  const size_t n_ui = (size_t)3 * D * XW, n_wh = (size_t)3 * D * D, n_bi = 3 * D, n_vs = (size_t)NB * D, n_bs = NB;
  double *a_ui = acc_dense, *a_wh = a_ui + n_ui, *a_bi = a_wh + n_wh, *a_vs = a_bi + n_bi, *a_bs = a_vs + n_vs, *a_sc = a_bs + n_bs;
  acc_ctx c = {acc_lt, acc_di, cnt_lt, cnt_di, D, abs_lt, abs_di};
  double* b_ui = abs_dense; double *b_wh = NULL, *b_bi = NULL, *b_vs = NULL, *b_bs = NULL, *b_sc = NULL;
  if (abs_dense) { b_wh = b_ui + n_ui; b_bi = b_wh + n_wh; b_vs = b_bi + n_bi; b_bs = b_vs + n_vs; b_sc = b_bs + n_bs; }
  for (int k = 0; k < n; ++k) {
    const int u = ids[k], b = off[u], L = off[u + 1] - b;
    seq_grad G;
    spatial_seq_grad(lt, di, ui, wh, bi, vs, bs, wd, lw, n_dist, D, p + b, q + b, dp + b, dq + b, L, &G);
    double* o = out5 + (size_t)5 * k;
    o[0] = G.ls0 * G.sur + G.ls1 * G.upq; o[1] = G.sur; o[2] = G.upq; o[3] = G.ls0; o[4] = G.ls1;
    spatial_row_deltas(lt, di, n_item, n_dist, D, p + b, q + b, dp + b, L, len_max, alpha, lam, &G, emit_acc, &c);
    const double *g_ui = G.gd, *g_wh = g_ui + n_ui, *g_bi = g_wh + n_wh, *g_vs = g_bi + n_bi, *g_bs = g_vs + n_vs;
    for (size_t i = 0; i < n_ui; ++i) a_ui[i] -= alpha * (g_ui[i] + lam * ui[i]);
    for (size_t i = 0; i < n_wh; ++i) a_wh[i] -= alpha * (g_wh[i] + lam * wh[i]);
    for (size_t i = 0; i < n_bi; ++i) a_bi[i] -= alpha * (g_bi[i] + lam * bi[i]);
    for (size_t i = 0; i < n_vs; ++i) a_vs[i] -= alpha * (g_vs[i] + lam * vs[i]);
    for (size_t i = 0; i < n_bs; ++i) a_bs[i] -= alpha * (g_bs[i] + lam * bs[i]);
    a_sc[0] -= alpha * (G.g_wd + lam * wd);
    {
      const double d0 = G.sur + lam * G.ls0, d1 = G.upq + lam * G.ls1, dt = d0 * G.ls0 + d1 * G.ls1;
      a_sc[1] -= alpha * G.ls0 * (d0 - dt);
      a_sc[2] -= alpha * G.ls1 * (d1 - dt);
      if (abs_dense) { b_sc[1] += fabs(alpha * G.ls0 * (d0 - dt)); b_sc[2] += fabs(alpha * G.ls1 * (d1 - dt)); }
    }
    if (abs_dense) {
      for (size_t i = 0; i < n_ui; ++i) b_ui[i] += fabs(alpha * (g_ui[i] + lam * ui[i]));
      for (size_t i = 0; i < n_wh; ++i) b_wh[i] += fabs(alpha * (g_wh[i] + lam * wh[i]));
      for (size_t i = 0; i < n_bi; ++i) b_bi[i] += fabs(alpha * (g_bi[i] + lam * bi[i]));
      for (size_t i = 0; i < n_vs; ++i) b_vs[i] += fabs(alpha * (g_vs[i] + lam * vs[i]));
      for (size_t i = 0; i < n_bs; ++i) b_bs[i] += fabs(alpha * (g_bs[i] + lam * bs[i]));
      b_sc[0] += fabs(alpha * (G.g_wd + lam * wd));
    }
    seq_grad_free(&G);
  }
  return 0;
}

/* Sequential epoch: seq_train(uidx) for uidx in order[0..n) (prog_bpr_gru_spatial.py:249-250). */
int poi_oracle_spatial_epoch(double* lt, double* di, double* ui, double* wh, double* bi, double* vs, double* bs,
                             double* wd, double* lw, int n_item, int n_dist, int D,
                             const int* off, const int* p, const int* q, const int* dp, const int* dq,
                             const int* order, int n, int len_max, double alpha, double lam, double* out5) {
  for (int k = 0; k < n; ++k) {
    const int u = order[k], b = off[u], L = off[u + 1] - b;
    poi_oracle_spatial_seq(lt, di, ui, wh, bi, vs, bs, wd, lw, n_item, n_dist, D, p + b, q + b, dp + b, dq + b, L, len_max,
                           alpha, lam, out5 + (size_t)5 * k);
  }
  return 0;
}

/* All-POI scores + top-K for n users (public/GRU.py:93-96 + public/Valuate.py:91-100): float64 dot
 * products, then selection of the K best (descending score, ties by ascending index). */
int poi_oracle_score_topk(const double* users, const double* items, int n, int n_item, int D, int K, int* idx_out) {
  double* sc = malloc(sizeof(double) * (size_t)n_item);
  for (int u = 0; u < n; ++u) {
    const double* ur = users + (size_t)u * D;
    for (int j = 0; j < n_item; ++j) {
      const double* it = items + (size_t)j * D;
      double s = 0;
      for (int d = 0; d < D; ++d) s += ur[d] * it[d];
      sc[j] = s;
    }
    int* out = idx_out + (size_t)u * K;
    int cnt = 0;
    for (int j = 0; j < n_item; ++j) {      /* insertion into a sorted K-list */
      if (cnt == K && !(sc[j] > sc[out[K - 1]])) continue;
      int pos = cnt < K ? cnt++ : K - 1;
      while (pos > 0 && sc[j] > sc[out[pos - 1]]) { out[pos] = out[pos - 1]; --pos; }
      out[pos] = j;
    }
  }
  free(sc);
  return 0;
}
