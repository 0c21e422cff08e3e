#!/bin/bash
# TOOLS ONLY: which library kernels torch.mm (hipBLASLt) runs on the 14B Linear shapes -- their names encode tile / pipeline choices.
cd "$(dirname "$0")/.." || exit 1
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
out=$GRAFT_REPO_ROOT/gpurun_out/r4/yardstick_trace
mkdir -p "$out"
cat > /tmp/mm_shapes.py <<'PY'
import torch
dev = torch.device("cuda:0")
for M, N, K in [(67080, 10240, 5120), (67080, 5120, 5120), (67080, 13824, 5120), (67080, 5120, 13824), (8392, 5120, 5120), (8392, 5120, 13824), (2304, 1536, 8960)]:
    a = torch.randn(M, K, device=dev, dtype=torch.bfloat16); w = torch.randn(N, K, device=dev, dtype=torch.bfloat16)
    o = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    for _ in range(3): torch.mm(a, w.t(), out=o)
    torch.cuda.synchronize()
PY
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -- python /tmp/mm_shapes.py > "$out/trace.log" 2>&1
f=$(find "$out/trace" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" "$out/kernel_stats.csv"
t=$(find "$out/trace" -name "*kernel_trace.csv" | head -1); [ -n "$t" ] && python - "$t" > "$out/kernel_trace_summary.txt" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    n = r.get("Kernel_Name", "")
    if "Cijk" in n or "gemm" in n.lower():
        print(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r.get("Grid_Size_X", r.get("Grid_Size")), r.get("Workgroup_Size_X", r.get("Workgroup_Size")), r.get("LDS_Block_Size"), r.get("VGPR_Count"), r.get("Accum_VGPR_Count"), r.get("SGPR_Count"), n)
PY
rm -rf "$out/trace"
cat "$out/kernel_trace_summary.txt" | cut -c1-600
