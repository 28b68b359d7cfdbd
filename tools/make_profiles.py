#!/usr/bin/env python
"""Turn the rocprofv3 CSVs collected by tools/profile_gpu.sh into the committed summaries under profiles/:
  <tag>_kernel_stats.csv / .md   per-kernel time (kernel trace + stats)
  <tag>_pmc_traffic.json         HBM bytes per launch (FETCH_SIZE / WRITE_SIZE, separate passes; gfx950 correction)
  <tag>_sq_counters.md           SQ wave / wait / MFMA-busy ratios
usage: python tools/make_profiles.py gpurun_out/prof_<tag> <tag>"""
import csv
import glob
import json
import os
import re
import shutil
import sys
from collections import defaultdict

src, tag = sys.argv[1], sys.argv[2]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, "profiles")


def short(name):
    m = re.match(r"_Z\d+(calib_[a-z0-9_]+?)P", name)      # (rocprofv3 leaves names with _Float16 vector arguments mangled)
    if m:
        return m.group(1)
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(.*", "", name)
    return name.replace("poi::", "")


def find(sub, pat):
    g = glob.glob(os.path.join(src, sub, "**", pat), recursive=True)
    return g[0] if g else None


# ---- kernel stats -------------------------------------------------------------------------------
st = find("stats", "*kernel_stats.csv")
rows = list(csv.DictReader(open(st)))
shutil.copy(st, os.path.join(P, tag + "_kernel_stats.csv"))
bench = None
try:
    bench = json.load(open(os.path.join(src, "bench_full.json")))      # (the FULL record: stdout's last line is the compact one since round 6)
except Exception:
    pass
with open(os.path.join(P, tag + "_kernel_stats.md"), "w") as f:
    f.write("# %s - rocprofv3 --kernel-trace --stats of `python bench.py --steps 2 --warmup 1 --eval-steps 2 --no-cpu-baseline --no-quality --no-secondary --no-exact --no-x1`\n\n" % tag)
    f.write("Gowalla-shape synthetic (100 k POIs, 50 k users, L <= 50, D = 128, 200 bins; 80 %% of the transitions local), one MI355X; 4 training\n"
            "epochs of 4 launches (12500 users each) + 3 evaluation passes.  Full CSV: `%s_kernel_stats.csv`.\n\n" % tag)
    f.write("| kernel | calls | total ms | avg us | % |\n|---|---|---|---|---|\n")
    for r in rows:
        if float(r["Percentage"]) < 0.05:
            continue
        f.write("| `%s` | %s | %.3f | %.1f | %.2f |\n" % (short(r["Name"]), r["Calls"], float(r["TotalDurationNs"]) / 1e6,
                                                         float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
    if bench:
        f.write("\nLive HIP-event timings of the same build (`bench.py`, defaults): %.0f seq/s, %.2f ms/epoch, eval %.0f users/s.\n\n"
                % (bench["value"], bench["ms_per_step"], bench.get("eval_users_per_s") or 0))
        f.write("| timed region | ms/epoch | achieved | of peak |\n|---|---|---|---|\n")
        for k, v in bench["kernels"].items():
            f.write("| %s | %.3f | %s | %s |\n" % (k, v["ms_per_step"], ("%.1f %s" % (v["achieved"], v["unit"])) if "achieved" in v else "",
                                                  ("%.0f %%" % (100 * v["frac"])) if "frac" in v else ""))


# ---- PMC traffic --------------------------------------------------------------------------------
BY_GRID = {}      # sub -> {(kernel, grid size): {counter: sum, "_n": dispatches}} for the BPR-MF kernels: bench.py launches them at several sizes


def counters(sub):
    fn = find(sub, "*counter_collection.csv")
    acc = defaultdict(lambda: defaultdict(float))
    calls = defaultdict(set)
    byg = defaultdict(lambda: defaultdict(float)); bygc = defaultdict(set)
    for r in csv.DictReader(open(fn)):
        k = short(r["Kernel_Name"])
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        calls[k].add(r["Dispatch_Id"])
        if k.startswith("bpr_") and "Grid_Size" in r:
            byg[(k, r["Grid_Size"])][r["Counter_Name"]] += float(r["Counter_Value"]); bygc[(k, r["Grid_Size"])].add(r["Dispatch_Id"])
    BY_GRID[sub] = {k: dict(v, _n=len(bygc[k])) for k, v in byg.items()}
    return acc, {k: len(v) for k, v in calls.items()}


fa, fc = counters("fetch")
wa, wc = counters("write")
# timed regions of bench.py -> kernels
REGION = {"te_gather": ["te_gather_kernel"], "te_gemm_ax": ["te_gemm_nt_kernel<true", "te_gemm_ntk_kernel<true", "te_gemm_ntk_kernel<false, true", "te_ptab_s3_kernel", "te_ztab_kernel", "te_xpack_kernel", "te_xwpackd_kernel", "te_xztab_kernel", "te_gemmx_kernel", "te_xcount_kernel", "te_xassign_kernel"], "te_gemm_dx": ["te_gemm_nt_kernel<false", "te_gemm_ntk_kernel<false, false"],
          "te_rec_fwd": ["te_rec_fwd16_kernel", "te_rec_fwdx_kernel", "te_rec_fwd1x_kernel", "te_rec_fwd1_kernel", "te_rec_fwd32_kernel", "te_rec_fwdd_kernel"],
          "te_rec_bwd": ["te_rec_bwd16_kernel", "te_rec_bwd16t_kernel", "te_rec_bwd1_kernel", "te_rec_bwd32_kernel"],
          # the TRAINING head (te_head3 / te_head_big3 on split products; te_head_kernel<.., 0> when split products are off); the predict head is its own row
          "te_head": ["te_head3_kernel", "te_head_big3_kernel", "te_bpr_head_kernel"], "te_predict_head": ["te_head_kernel", "te_head_big_kernel"],
          "te_wgrad": ["te_wgrad_kernel"], "te_psum": ["te_pcount_kernel", "te_passign_kernel", "te_psum_kernel", "te_pfin_kernel"], "te_scatter": ["te_reduce_kernel", "te_hot_reduce_kernel", "te_hot_apply_kernel"], "te_dsum": ["te_dprep_kernel", "te_dsum_kernel"], "te_bin_gemm": ["te_dred_kernel", "te_dfin_kernel", "te_dui_kernel", "te_dapply_kernel"],
          "dense_apply": ["dense_apply_kernel"], "te_finalize": ["te_finalize_kernel", "te_parts_kernel"],
          "te_prep": ["te_len_kernel", "te_scan_kernel", "te_rowmap_kernel", "te_pack_kernel", "te_transpose_kernel", "rs_hist_kernel",
                      "rs_digit_scan_kernel", "rs_scatter_kernel", "te_segment_kernel"],
          "score_topk": ["score_kernel_packed", "score_filter_kernel", "score_rescore_kernel", "sf_select_kernel", "score_merge_kernel"], "te_predict": [],
          # secondary_bpr block of bench.py (BPR-MF step, csrc/bpr.hip; its radix-sort passes are the rs_* kernels listed under te_prep)
          "bpr_users": ["bpr_keys_kernel", "bpr_chunk_kernel<0", "bpr_span_kernel<0"], "bpr_items": ["bpr_chunk_kernel<1", "bpr_span_kernel<1", "bpr_commit_kernel"],
          "bpr_hogwild": ["bpr_hogwild_kernel"]}
out = {"config": "bench.py default (gowalla shape, batch_users 12500)",
       "note": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE in separate passes; FETCH_SIZE doubled per MI355X_MICROARCH.md "
               "(gfx950 reports half of wide coalesced reads); WRITE_SIZE calibrated with tools/micro/write_calib.hip (exact for the 64-byte-segment pattern of the recurrent kernels); counter unit KB (x 1024); per launch of the TIMED REGION (sum over its kernels), each pass normalised by its own launch count",
       "kernels": {}}
for reg, pats in REGION.items():
    fk = [k for k in fa if any(k.startswith(p) for p in pats)]
    if not fk:
        continue
    n_launch = max(fc[k] for k in fk)
    if reg == "te_prep":
        n_launch = fc.get("te_rowmap_kernel", n_launch)
    fetch = sum(fa[k]["FETCH_SIZE"] for k in fk)
    # the two passes are separate runs of a bench with time-based sections: each pass is normalised by ITS OWN launch count
    # (round 3 found the write pass 1.49 x as long as the fetch pass - every write figure of the first r03 set was inflated by that)
    wk = [k for k in wa if any(k.startswith(p) for p in pats)]
    n_w = max([wc[k] for k in wk], default=n_launch)
    if reg == "te_prep":
        n_w = wc.get("te_rowmap_kernel", n_w)
    write = sum(wa[k]["WRITE_SIZE"] for k in wk)
    out["kernels"][reg] = {"launches": n_launch, "launches_write_pass": n_w, "fetch_kb_raw": fetch / n_launch, "write_kb": write / max(n_w, 1),
                           "hbm_bytes_per_launch": (2.0 * fetch / n_launch + write / max(n_w, 1)) * 1024.0}
# BPR-MF kernels per launch size (grid = work-groups x 256 threads: 974 k-triple launches, 262144-triple launches and the epoch's remainder have different grids)
bpr_sizes = {}
for (k, g), c in sorted(BY_GRID.get("fetch", {}).items()):
    w = BY_GRID.get("write", {}).get((k, g), {})
    bpr_sizes["%s [grid %s]" % (k, g)] = {"launches": c["_n"], "fetch_kb_raw": c.get("FETCH_SIZE", 0.0) / max(c["_n"], 1), "write_kb": w.get("WRITE_SIZE", 0.0) / max(w.get("_n", 1), 1),
                                          "hbm_bytes_per_launch": (2.0 * c.get("FETCH_SIZE", 0.0) / max(c["_n"], 1) + w.get("WRITE_SIZE", 0.0) / max(w.get("_n", 1), 1)) * 1024.0}
if bpr_sizes:
    out["bpr_by_launch_size"] = bpr_sizes
json.dump(out, open(os.path.join(P, tag + "_pmc_traffic.json"), "w"), indent=1)

# ---- SQ counters --------------------------------------------------------------------------------
sa, sc = counters("sq")
# calibration (tools/micro/mfma_calib.hip under the same --pmc set): SQ_VALU_MFMA_BUSY_CYCLES / SQ_BUSY_CYCLES of a pure-MFMA loop = 100 % issue
calib = {}
if find("calib", "*counter_collection.csv"):
    ca, _ = counters("calib")
    for k, c in ca.items():
        if k.startswith("calib_") and c["SQ_BUSY_CYCLES"] > 0:
            calib[k[len("calib_"):]] = c["SQ_VALU_MFMA_BUSY_CYCLES"] / c["SQ_BUSY_CYCLES"]
# measured rate of the same loops (calib.log: event-timed, no profiler) against the nominal dense peaks of MI355X_MICROARCH.md: under a pure-MFMA load the
# chip sustains ~83 - 94 % of nominal (clocks) - bench.py's `frac` divides by the NOMINAL peak, so frac ~= calibrated % x that ratio
NOMINAL = {"bf16": 2500.0, "f16": 2500.0, "i8": 5000.0, "f32": 157.3, "f64": 78.6}
rate = {}
try:
    for line in open(os.path.join(src, "calib.log")):
        m = re.match(r"^(bf16|f16|i8|f32|f64) (\S+)\s+([\d.]+) ms\s+([\d.]+) T", line)
        if m:
            rate["%s_%s" % (m.group(1), m.group(2))] = float(m.group(4))
except Exception:
    pass
FAMILY = [("te_rec_fwdx", "i8_16x16x64"), ("te_gemmx", "i8_32x32x32"), ("te_rec_fwdd", "f64_16x16x4"), ("te_wgrad", "bf16_32x32x16"), ("te_head3", "bf16_32x32x16"),
          ("te_head_big3", "bf16_32x32x16"), ("te_rec_bwd16t", "bf16_16x16x32"), ("te_rec_bwd16", "bf16_16x16x32"), ("te_rec_bwd32", "bf16_32x32x16"),
          ("te_rec_fwd32", "bf16_32x32x16"), ("te_gemm_ntk", "bf16_32x32x16"), ("te_ptab_s3", "bf16_32x32x16"), ("score_filter", "f16_32x32x16"),
          ("score_rescore", "f32_32x32x2"), ("score_kernel", "f32_32x32x2"), ("te_head_kernel", "f32_32x32x2"), ("te_gemm_nt", "f32_32x32x2")]
bench_frac = {}
if bench:
    for reg, pats in REGION.items():
        if reg in bench.get("kernels", {}) and bench["kernels"][reg].get("bound") == "mfma":
            for p_ in pats:
                bench_frac[p_] = (reg, bench["kernels"][reg]["frac"])
    ts = (bench.get("eval") or {}).get("two_stage") or {}
    if ts.get("filter_frac_of_f16_mfma_peak"):
        bench_frac["score_filter_kernel"] = ("eval filter", ts["filter_frac_of_f16_mfma_peak"])
with open(os.path.join(P, tag + "_sq_counters.md"), "w") as f:
    f.write("# %s - SQ counters per kernel (rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY "
            "SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS)\n\n" % tag)
    f.write("Same command as the kernel stats.  Percentages are of SQ_WAVE_CYCLES: WAIT_ANY = wave parked at s_waitcnt / barrier, WAIT_INST = issue "
            "stall (MFMA dependency / pipe), ACTIVE = issuing.\n\n")
    if calib:
        f.write("**Calibration** (`tools/micro/mfma_calib.hip`, pure-MFMA loops - four independent accumulators, two waves per SIMD on all 256 CUs - under the "
                "same counter set): MFMA_BUSY/BUSY at 100 %% issue of the matrix pipe, and the rate the same loop sustains (event-timed, no profiler)\n\n"
                "| instruction | MFMA_BUSY/BUSY at full issue | sustained T(FL)OP/s | of the nominal dense peak |\n|---|---|---|---|\n")
        for k in sorted(calib):
            nom = NOMINAL.get(k.split("_")[0])
            f.write("| `v_mfma_*_%s` | %.2f | %s | %s |\n" % (k, calib[k], ("%.0f" % rate[k]) if k in rate else "", ("%.0f %%" % (100 * rate[k] / nom)) if k in rate and nom else ""))
        f.write("\n`calibrated MFMA %` = the kernel's MFMA_BUSY/BUSY over the full-issue ratio of its instruction family: the counter-backed utilisation of the matrix "
                "pipe; `x sustained/nominal` = that utilisation expressed against the NOMINAL peak (what bench.py's `frac` - executed partial products over the dense peak of "
                "`MI355X_MICROARCH.md` - divides by): the two agree when the accounting of executed products is right.\n\n")
    f.write("| kernel | launches | WAIT_ANY % | WAIT_INST % | ACTIVE % | MFMA_BUSY/BUSY | calibrated MFMA % | x sustained/nominal | bench.py frac (region) | LDS_BANK_CONFLICT % | WAIT_INST_LDS % |\n|---|---|---|---|---|---|---|---|---|---|---|\n")
    order = sorted(sa, key=lambda k: -sa[k]["SQ_BUSY_CYCLES"])
    for k in order:
        c = sa[k]
        wv = c["SQ_WAVE_CYCLES"]
        if wv <= 0 or not (k.startswith("te_") or k.startswith("score") or k.startswith("rs_") or k.startswith("sf_") or "apply" in k or "topk" in k or "pack" in k):
            continue
        ratio = c["SQ_VALU_MFMA_BUSY_CYCLES"] / max(c["SQ_BUSY_CYCLES"], 1)
        fam = next((fm for pre, fm in FAMILY if k.startswith(pre)), None)
        cal = ("%.0f (%s)" % (100.0 * ratio / calib[fam], fam)) if fam in calib and ratio > 0 else ""
        nom = NOMINAL.get(fam.split("_")[0]) if fam else None
        caln = ("%.0f" % (100.0 * ratio / calib[fam] * rate[fam] / nom)) if fam in calib and fam in rate and nom and ratio > 0 else ""
        bf = next((v for p_, v in bench_frac.items() if k.startswith(p_)), None)
        f.write("| `%s` | %d | %.0f | %.0f | %.0f | %.2f | %s | %s | %s | %.1f | %.1f |\n" % (
            k, sc[k], 100 * c["SQ_WAIT_ANY"] / wv, 100 * c["SQ_WAIT_INST_ANY"] / wv, 100 * c["SQ_ACTIVE_INST_ANY"] / wv,
            ratio, cal, caln, ("%.0f %% (%s)" % (100 * bf[1], bf[0])) if bf else "", 100 * c["SQ_LDS_BANK_CONFLICT"] / wv, 100 * c["SQ_WAIT_INST_LDS"] / wv))
print("wrote", [x for x in os.listdir(P) if x.startswith(tag)])
