"""Build libpoi_hip.so (gfx950) in-tree with hipcc.  hipcc cross-compiles without a GPU."""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libpoi_hip.so")
SOURCES = ["abi.hip", "seq_engine.hip", "tile_engine.hip", "te_scatter.hip", "bpr.hip", "score_topk.hip", "misc.hip", "sync.hip", "carnn.hip"]
HEADERS = ["poi_common.h", "poi_kernels.h", "seq_common.h", os.path.join("..", "..", "include", "poi_hip.h")]


def _hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC or install ROCm)")


def is_stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS]
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build_lib(force=False, verbose=True):
    """Compile every HIP translation unit for gfx950 into one shared library next to this file."""
    if not force and not is_stale():
        return LIB
    cmd = [_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
           "-ldl", "-o", LIB + ".tmp"] + os.environ.get("POI_HIPCC_FLAGS", "").split() + [os.path.join(CSRC, s) for s in SOURCES]
    if verbose:
        print("[build]", " ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    os.replace(LIB + ".tmp", LIB)
    return LIB


if __name__ == "__main__":
    print(build_lib(force=True))
