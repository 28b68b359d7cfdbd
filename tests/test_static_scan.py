"""Static guard on the compiled gfx950 code of the training step's hot kernels (tools/scan_waits.py: llvm-objdump on the library's
embedded code objects; no GPU).  It pins what DESIGN.md 5 describes as fixed in round 2: no spilled registers in te_head / te_wgrad /
the recurrent kernels (a spill reload is `scratch_load` + `s_waitcnt vmcnt(0)` = a wait for every older store), and no load in te_wgrad's
pipeline that is waited for before the next MFMA block (the index prefetch).  Skipped when the ROCm binary tools are not installed."""
import importlib.util
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("scan_waits", os.path.join(ROOT, "tools", "scan_waits.py"))
scan_waits = importlib.util.module_from_spec(spec)
spec.loader.exec_module(scan_waits)

HOT = ["te_head_kernel<128, 7, 0>",                          # float32-input MFMA head: predict, and training with split products off
       "te_head3_kernel<128, 7, true>",                            # training head on split products (round 4): three workgroups per CU
       "te_wgrad_kernel<128, 128, false>", "te_wgrad_kernel<128, 128, true>",
       "te_rec_fwdx_kernel<128, true, false>", "te_rec_fwdx_kernel<128, false, false>",      # exact forward: table / per-step rows
       "te_rec_fwd16_kernel<128, false, true, false>",       # forward table, float32-input MFMA (exact forward off)
       "te_rec_fwd16_kernel<128, false, false, true>",       # split products
       "te_rec_bwd16t_kernel<128>",                          # BPTT tiles, transposed split products (round 4)
       "te_rec_bwd16_kernel<128, false>", "te_rec_bwd16_kernel<128, true>",
       "te_rec_fwd1_kernel<128, false>", "te_rec_bwd1_kernel<128>", "te_one_in_kernel<128>", "te_one_out_kernel<128>",
       "te_head_big3_kernel<128, 0>",                        # chunked head (1520 bins) on split products, two workgroups per CU
       "te_ptab_s3_kernel<128, false>",                      # forward table on split products: 96 registers of resident A planes, LDS-DMA ring
       "te_gemmx_kernel<128>",                               # exact forward's input product
       "te_gemm_ntk_kernel<false, true, 128, 128, 384, 128, 384, false, false, false>",
       "te_gemm_ntk_kernel<false, false, 384, 0, 128, 384, 256, false, false, true>"]       # te_gemm_dx on split products
# te_rec_bwd16<SP> (one unit of four sequences per lane; the A/B form of te_rec_bwd16t since round 4): 96 registers of resident weight planes
# + two sets of operand prefetch + the split temporaries exceed the 256 registers of two waves per SIMD by a few values the compiler keeps
# in scratch.  te_gemmx: resident digit planes of 128 rows + LDS-DMA bookkeeping (still faster than the version that fits).
# Pinned so that they do not grow.
SPILL_ALLOWED = {"te_rec_bwd16_kernel<128, true>": 24, "te_gemmx_kernel<128>": 36}


@pytest.fixture(scope="module")
def records(tmp_path_factory):
    import poi_amd
    poi_amd.build.build_lib()                      # (no-op when the library is up to date)
    if not scan_waits.available():
        pytest.skip("llvm-objdump / clang-offload-bundler not installed")
    return scan_waits.scan(HOT, tmp=str(tmp_path_factory.mktemp("scan")))


def test_hot_kernels_are_found_and_do_not_spill(records):
    kernels = {r["kernel"]: r for r in records if "loop" not in r}
    for name in HOT:
        assert name in kernels, "kernel %s not found in the library (renamed? update HOT)" % name
        assert kernels[name]["spill"] <= SPILL_ALLOWED.get(name, 0), "%s spills %d registers" % (name, kernels[name]["spill"])
    # three te_head workgroups per CU need <= 168 registers; two te_wgrad / GEMM workgroups <= 256
    assert kernels["te_head_kernel<128, 7, 0>"]["vgpr"] <= 168
    assert kernels["te_head3_kernel<128, 7, true>"]["vgpr"] <= 168


def test_wgrad_pipeline_never_waits_for_a_load_it_has_just_issued(records):
    for r in records:
        if "loop" in r and r["kernel"].startswith("te_wgrad_kernel") and r["mfma"] >= 40:      # (two 16-row stages: 48 MFMAs)
            assert r["full_drains"] == 0, r
            assert r["loads_waited_before_next_mfma"] == 0, r
