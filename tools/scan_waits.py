#!/usr/bin/env python3
"""Static scan of libpoi_hip.so's gfx950 code (llvm-objdump on the embedded code objects).  Per kernel loop:
  * `s_waitcnt vmcnt(0|1)` INSIDE the loop (between a backward branch and its target): a full drain of the vector-memory queue in a hot
    loop means some wait is not counted exactly - a value first used at the top of the next iteration, a reload from scratch, a load
    issued last but needed first (vmcnt retires in order);
  * in MFMA loops, a wait that targets a LOAD issued since the last MFMA: its consumer was scheduled right behind the load and the wave
    sits out a memory latency instead of covering it with the MFMA block.
These are the patterns behind the te_head / te_wgrad fixes of round 2.  Also reports registers / spills per kernel.
Usage: tools/scan_waits.py [name-substring ...]          (tests/test_static_scan.py keeps the training step's hot kernels clean)"""
import collections
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO = os.path.join(ROOT, "point-of-interest-recommendation_amd", "libpoi_hip.so")
LLVM = "/opt/rocm/lib/llvm/bin"
VM_LOAD = ("global_load", "buffer_load", "scratch_load")
VM_ANY = VM_LOAD + ("global_store", "buffer_store", "scratch_store", "global_atomic")


def available():
    return os.path.exists(SO) and all(os.path.exists(os.path.join(LLVM, t)) for t in ("llvm-objcopy", "clang-offload-bundler", "llvm-objdump", "llvm-readelf"))


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
    res = {}
    for n, d in zip(names, out):
        d = re.sub(r"^void ", "", d).replace("poi::", "")
        res[n] = re.sub(r"\(.*$", "", d)
    return res


def code_objects(tmp):
    os.makedirs(tmp, exist_ok=True)
    subprocess.run([LLVM + "/llvm-objcopy", "--dump-section", ".hip_fatbin=%s/fat.bin" % tmp, SO], check=True)
    blob = open(tmp + "/fat.bin", "rb").read()
    pos = [m.start() for m in re.finditer(b"__CLANG_OFFLOAD_BUNDLE__", blob)] + [len(blob)]
    cos = []
    for i in range(len(pos) - 1):
        open("%s/b%d.bin" % (tmp, i), "wb").write(blob[pos[i]:pos[i + 1]])
        co = "%s/k%d.co" % (tmp, i)
        subprocess.run([LLVM + "/clang-offload-bundler", "--unbundle", "--type=o", "--input=%s/b%d.bin" % (tmp, i),
                        "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co], check=True)
        cos.append(co)
    return cos


def resources(co):
    """{mangled kernel name: (vgpr, agpr, spilled vgprs, lds bytes)} from the code object's metadata note."""
    t = subprocess.run([LLVM + "/llvm-readelf", "--notes", co], capture_output=True, text=True).stdout
    res = {}
    for blk in t.split("  - .agpr_count:")[1:]:
        g = lambda key: int(re.search(r"\.%s:\s+(\d+)" % key, blk).group(1)) if re.search(r"\.%s:\s+(\d+)" % key, blk) else 0
        name = re.search(r"\.name:\s+(\S+)", blk).group(1)
        res[name] = (g("vgpr_count"), int(blk.split()[0]), g("vgpr_spill_count"), g("group_segment_fixed_size"))
    return res


def scan(filters=(), tmp="/tmp/poi_scan"):
    """-> list of dicts, one per (kernel, loop) with a finding, plus one {'kernel', 'vgpr', 'spill'} record per kernel (key 'loop' absent)."""
    out = []
    for co in code_objects(tmp):
        dis = subprocess.run([LLVM + "/llvm-objdump", "-d", co], capture_output=True, text=True).stdout
        cur, funcs = None, collections.OrderedDict()
        for line in dis.splitlines():
            m = re.match(r"^[0-9a-f]+ <(.+)>:", line)
            if m:
                cur = m.group(1); funcs[cur] = []; continue
            m = re.match(r"\s+([a-z_0-9]+)\s*(.*?)\s*//\s*([0-9A-F]+):", line)
            if m and cur:
                funcs[cur].append((int(m.group(3), 16), m.group(1), m.group(2)))
        names = demangle(list(funcs))
        res = resources(co)
        for name, ins in funcs.items():
            dem = names[name]
            if filters and not any(a in dem for a in filters):
                continue
            if name in res:
                out.append({"kernel": dem, "vgpr": res[name][0], "agpr": res[name][1], "spill": res[name][2], "lds": res[name][3]})
            for a, op, args in ins:
                if not (op.startswith("s_cbranch") or op == "s_branch"):
                    continue
                m = re.search(r"(-?\d+)", args)
                if not m:
                    continue
                off = int(m.group(1)); off -= 65536 if off >= 32768 else 0
                t = a + 4 + off * 4
                if t > a:
                    continue
                body = [x for x in ins if t <= x[0] <= a]
                mf = sum(1 for x in body if x[1].startswith("v_mfma"))
                vm = sum(1 for x in body if x[1].startswith(VM_ANY))
                full = sum(1 for x in body if x[1] == "s_waitcnt" and re.search(r"vmcnt\((0|1)\)", x[2]))
                hot = 0
                if mf >= 16:
                    ops, nm = [], 0
                    for x in body:
                        if x[1].startswith("v_mfma"):
                            nm += 1
                        elif x[1].startswith(VM_ANY):
                            ops.append((nm, x[1].startswith(VM_LOAD)))
                        elif x[1] == "s_waitcnt":
                            w = re.search(r"vmcnt\((\d+)\)", x[2])
                            if w and len(ops) > int(w.group(1)):
                                at, is_load = ops[len(ops) - 1 - int(w.group(1))]
                                hot += 1 if (is_load and at == nm) else 0
                if (full and (mf or vm >= 8)) or hot:
                    out.append({"kernel": dem, "loop": len(body), "mfma": mf, "vmem": vm, "full_drains": full, "loads_waited_before_next_mfma": hot})
    return out


if __name__ == "__main__":
    for r in scan(sys.argv[1:]):
        if "loop" in r:
            print("%-72s loop %5d instr, %3d mfma, %3d vmem: %d x vmcnt(0|1), %d load(s) waited for before the next MFMA"
                  % (r["kernel"][:72], r["loop"], r["mfma"], r["vmem"], r["full_drains"], r["loads_waited_before_next_mfma"]))
        else:
            print("%-72s vgpr %3d agpr %3d spilled %3d lds %6d" % (r["kernel"][:72], r["vgpr"], r["agpr"], r["spill"], r["lds"]))
