cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
tools/micro/phase_overlap
echo "--- default"; python tools/launch_sizes.py 256 1563 3125 2>&1 | tail -3
echo "--- graph"; LS_GRAPH=1 python tools/launch_sizes.py 256 1563 3125 2>&1 | tail -3
echo "--- per-seq 2048"; POI_TE_XREC1=2048 POI_TE_REC1=2048 python tools/launch_sizes.py 1563 2>&1 | tail -1
echo "--- xrec1 2048 only"; POI_TE_XREC1=2048 python tools/launch_sizes.py 1563 2>&1 | tail -1
echo "--- rec1 2048 only"; POI_TE_REC1=2048 python tools/launch_sizes.py 1563 2>&1 | tail -1
rocprofv3 --kernel-trace -f csv -d gpurun_out/mid_trace -o k -- python tools/launch_sizes.py 1563 > /dev/null 2>&1
python tools/chain_gaps.py $(ls gpurun_out/mid_trace/*/k_kernel_trace.csv gpurun_out/mid_trace/k_kernel_trace.csv 2>/dev/null | head -1) te_len
