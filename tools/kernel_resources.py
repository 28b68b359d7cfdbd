#!/usr/bin/env python
"""Register / scratch / occupancy table of every kernel in a HIP source (hipcc -Rpass-analysis).
usage: python tools/kernel_resources.py point-of-interest-recommendation_amd/csrc/tile_engine.hip [filter]"""
import re
import subprocess
import sys

src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
out = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-c", src, "-o", "/dev/null",
                      "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True).stderr
if "error" in out:
    print("\n".join(l for l in out.splitlines() if "error" in l)); sys.exit(1)
demangle = lambda n: subprocess.run(["c++filt", n], capture_output=True, text=True).stdout.strip()
cur = None
rows = []
for line in out.splitlines():
    m = re.search(r"remark:\s+(.*?) \[-Rpass", line)
    if not m:
        continue
    t = m.group(1).strip()
    if t.startswith("Function Name:"):
        cur = {"name": t.split(":", 1)[1].strip()}; rows.append(cur)
    elif cur is not None and ":" in t:
        k, v = t.split(":", 1); cur[k.strip()] = v.strip()
print("%-52s %5s %5s %8s %4s %6s" % ("kernel", "VGPR", "AGPR", "scratch", "occ", "LDS"))
for r in rows:
    n = re.sub(r"\(.*", "", demangle(r["name"])).replace("void poi::", "").replace("poi::", "")
    if flt in n:
        print("%-52s %5s %5s %8s %4s %6s" % (n[:52], r.get("VGPRs"), r.get("AGPRs"), r.get("ScratchSize [bytes/lane]"),
                                            r.get("Occupancy [waves/SIMD]"), r.get("LDS Size [bytes/block]")))
