#!/bin/bash
# kernel trace of tools/bench_configs.py (C2 / C4 informational configs)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/c4_trace -o p -- python3 tools/bench_configs.py > gpurun_out/c4_trace.log 2>&1
python3 - <<'PY'
import csv, collections
rows = list(csv.DictReader(open("gpurun_out/c4_trace/p_kernel_trace.csv")))
acc = collections.defaultdict(list)
for r in rows:
    n = r["Kernel_Name"]
    if "step_" in n:
        acc[(n.split("(")[0][:44], r.get("Grid_Size_X", r.get("Grid_Size", "")))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in acc.items():
    print(k, len(v), "avg us %.1f" % (sum(v) / len(v)))
PY
grep -E "C2|C4" gpurun_out/c4_trace.log
