"""Build libpoi_hip.so (gfx950) in-tree with hipcc.  hipcc cross-compiles without a GPU.

Every translation unit is compiled to its own object (in parallel, cached under csrc/_obj by source / header mtime and
flags) and the objects are linked into one shared library next to this file."""
import concurrent.futures as cf
import hashlib
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "_obj")
LIB = os.path.join(HERE, "libpoi_hip.so")
SOURCES = ["abi.hip", "seq_engine.hip", "exact_engine.hip", "tile_engine.hip", "te_scatter.hip", "te_xfwd.hip", "bpr.hip", "score_topk.hip", "score_filter.hip",
           "misc.hip", "sync.hip", "carnn.hip"]
HEADERS = ["poi_common.h", "poi_kernels.h", "seq_common.h", os.path.join("..", "..", "include", "poi_hip.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC"]


def _hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC or install ROCm)")


def _sources():
    return [s for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]


def is_stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in _sources() + HEADERS]
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def _compile_one(hipcc, src, extra, force, verbose):
    path = os.path.join(CSRC, src)
    tag = hashlib.sha1((" ".join(FLAGS + extra)).encode()).hexdigest()[:8]
    obj = os.path.join(OBJ, "%s.%s.o" % (os.path.splitext(src)[0], tag))
    deps = [path] + [os.path.join(CSRC, h) for h in HEADERS]
    if not force and os.path.exists(obj) and all(os.path.getmtime(d) <= os.path.getmtime(obj) for d in deps if os.path.exists(d)):
        return obj
    cmd = [hipcc] + FLAGS + extra + ["-c", path, "-o", obj + ".tmp"]
    if verbose:
        print("[build]", " ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    os.replace(obj + ".tmp", obj)
    return obj


def build_lib(force=False, verbose=True):
    """Compile every HIP translation unit for gfx950 into one shared library next to this file."""
    if not force and not is_stale():
        return LIB
    hipcc = _hipcc()
    extra = os.environ.get("POI_HIPCC_FLAGS", "").split()
    os.makedirs(OBJ, exist_ok=True)
    srcs = _sources()
    with cf.ThreadPoolExecutor(max(1, min(len(srcs), os.cpu_count() or 1))) as ex:
        objs = list(ex.map(lambda s: _compile_one(hipcc, s, extra, force and not os.environ.get("POI_BUILD_INCREMENTAL"), verbose), srcs))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB + ".tmp"] + objs + ["-ldl"]
    if verbose:
        print("[build]", " ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    os.replace(LIB + ".tmp", LIB)
    return LIB


if __name__ == "__main__":
    print(build_lib(force=True))
