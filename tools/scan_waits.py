#!/usr/bin/env python3
"""Static scan of libpoi_hip.so's gfx950 code: for every kernel, the s_waitcnt vmcnt(0) / vmcnt(1) that sit INSIDE a loop (between a backward
branch and its target), next to the loop's MFMA / vector-memory counts.  A full drain inside a hot loop means some wait is not counted
exactly (a value consumed at the top of the next iteration, a reload from scratch, a load issued last but needed first) - the pattern
behind three fixes of round 2 (te_head's spilled addresses, te_wgrad's index prefetch).  Usage: tools/scan_waits.py [name-substring ...]"""
import os, re, subprocess, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO = os.path.join(ROOT, "point-of-interest-recommendation_amd", "libpoi_hip.so")
LLVM = "/opt/rocm/lib/llvm/bin"
TMP = "/tmp/poi_scan"
os.makedirs(TMP, exist_ok=True)
subprocess.run([LLVM + "/llvm-objcopy", "--dump-section", ".hip_fatbin=%s/fat.bin" % TMP, SO], check=True)
blob = open(TMP + "/fat.bin", "rb").read()
pos = [m.start() for m in re.finditer(b"__CLANG_OFFLOAD_BUNDLE__", blob)] + [len(blob)]
for i in range(len(pos) - 1):
    open("%s/b%d.bin" % (TMP, i), "wb").write(blob[pos[i]:pos[i + 1]])
    subprocess.run([LLVM + "/clang-offload-bundler", "--unbundle", "--type=o", "--input=%s/b%d.bin" % (TMP, i),
                    "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=%s/k%d.co" % (TMP, i)], check=True)
    dis = subprocess.run([LLVM + "/llvm-objdump", "-d", "%s/k%d.co" % (TMP, i)], capture_output=True, text=True).stdout
    cur, funcs = None, collections.OrderedDict()
    for line in dis.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.+)>:", line)
        if m:
            cur = m.group(1); funcs[cur] = []; continue
        m = re.match(r"\s+([a-z_0-9]+)\s*(.*?)\s*//\s*([0-9A-F]+):", line)
        if m and cur:
            funcs[cur].append((int(m.group(3), 16), m.group(1), m.group(2)))
    for name, ins in funcs.items():
        dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        dem = re.sub(r"^void ", "", dem).replace("poi::", ""); dem = re.sub(r"\(.*$", "", dem)
        if len(sys.argv) > 1 and not any(a in dem for a in sys.argv[1:]):
            continue
        loops = []
        for a, op, args in ins:
            if op.startswith("s_cbranch") or op == "s_branch":
                m = re.search(r"(-?\d+)", args)
                if m:
                    off = int(m.group(1)); off -= 65536 if off >= 32768 else 0
                    t = a + 4 + off * 4
                    if t <= a:
                        loops.append((t, a))
        for t, a in loops:
            body = [x for x in ins if t <= x[0] <= a]
            full = [x for x in body if x[1] == "s_waitcnt" and re.search(r"vmcnt\((0|1)\)", x[2])]
            mf = sum(1 for x in body if x[1].startswith("v_mfma"))
            vm = sum(1 for x in body if x[1].startswith(("global_", "buffer_", "scratch_")))
            if full and (mf or vm >= 8):
                print("%-70s loop %5d instr, %3d mfma, %3d vmem: %d x vmcnt(0|1)" % (dem[:70], len(body), mf, vm, len(full)))
            # second pattern: in an MFMA loop, a wait that targets a load issued with NO MFMA in between (its consumer was scheduled
            # right behind the load: the wave sits out a memory latency instead of covering it with the MFMA block)
            if mf >= 16:
                ops, nm, hot = [], 0, 0
                for x in body:
                    if x[1].startswith("v_mfma"):
                        nm += 1
                    elif x[1].startswith(("global_load", "buffer_load", "scratch_load", "global_store", "buffer_store", "scratch_store", "global_atomic")):
                        ops.append((nm, x[1].startswith(("global_load", "buffer_load", "scratch_load"))))
                    elif x[1] == "s_waitcnt":
                        m = re.search(r"vmcnt\((\d+)\)", x[2])
                        if m and len(ops) > int(m.group(1)):
                            at, is_load = ops[len(ops) - 1 - int(m.group(1))]
                            if is_load and at == nm:
                                hot += 1
                if hot:
                    print("%-70s loop %5d instr, %3d mfma: %d wait(s) for a load issued since the last MFMA" % (dem[:70], len(body), mf, hot))
