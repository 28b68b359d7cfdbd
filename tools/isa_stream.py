#!/usr/bin/env python
"""One character per instruction of a kernel's ISA, in program order - to SEE whether MFMA and VALU work alternate:
M mfma, . valu, t transcendental / f64 rcp, r / w LDS read / write, L / S global load / store, ~ s_waitcnt, | s_barrier, X scratch.
    python tools/isa_stream.py point-of-interest-recommendation_amd/csrc/te_xfwd.hip 'te_rec_fwdx_kernelILi128ELb0'"""
import re, subprocess, sys, tempfile, os
src, pat = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
with tempfile.TemporaryDirectory() as d:
    out = os.path.join(d, "x.s")
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", "-I", os.path.join(root, "include"),
                    "-o", out, src], check=True, stderr=subprocess.DEVNULL)
    s = open(out).read()
m = re.search(r"^(\S*%s\S*):.*?\n(.*?)\n\s*s_endpgm" % re.escape(pat), s, re.S | re.M)
if not m:
    sys.exit("kernel not found")
seq = []
for l in m.group(2).splitlines():
    l = l.strip()
    if not l or l.startswith(";") or l.startswith("."):
        continue
    op = l.split()[0]
    if l.endswith(":") or re.match(r"^\.?LBB", op): seq.append("\n" + op + "\n")
    elif op.startswith("v_mfma"): seq.append("M")
    elif op.startswith("ds_write") or op.startswith("ds_store"): seq.append("w")
    elif op.startswith("ds_"): seq.append("r")
    elif op.startswith("global_load") or op.startswith("buffer_load"): seq.append("L")
    elif op.startswith("global_store") or op.startswith("buffer_store"): seq.append("S")
    elif op.startswith("s_barrier"): seq.append("|")
    elif op.startswith("s_waitcnt"): seq.append("~")
    elif op.startswith("scratch"): seq.append("X")
    elif op.startswith("v_rcp") or op.startswith("v_exp") or op.startswith("v_log") or op.startswith("v_rsq") or op.startswith("v_sqrt"): seq.append("t")
    elif op.startswith("v_"): seq.append(".")
print("".join(seq))
