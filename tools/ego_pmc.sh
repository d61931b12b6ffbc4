#!/bin/bash
# rocprofv3 passes over tools/ego_prof.py (65 536 observations of the C3 batch): kernel trace + stats, then PMC sets,
# each in its own run.  Summaries land in gpurun_out/ego_*; copy what is to be judged into profiles/.
set -e
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ego_trace -o p -- python3 tools/ego_prof.py > gpurun_out/ego_trace.log 2>&1
grep -E "Name|ego_|goal_n" gpurun_out/ego_trace/p_kernel_stats.csv > gpurun_out/ego_kernel_stats.csv
rm -f gpurun_out/ego_pmc_summary.txt
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "GRBM_GUI_ACTIVE SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --output-format csv --pmc $set -d gpurun_out/ego_pmc_$i -o p -- python3 tools/ego_prof.py > gpurun_out/ego_pmc_$i.log 2>&1
  python3 - <<PY
import csv, glob, collections
f = glob.glob("gpurun_out/ego_pmc_$i/**/*counter_collection.csv", recursive=True)
acc = collections.defaultdict(lambda: [0, 0.0])
for fn in f:
    for r in csv.DictReader(open(fn)):
        if "ego_costmap" in r["Kernel_Name"]:
            a = acc[r["Counter_Name"]]
            a[0] += 1
            a[1] += float(r["Counter_Value"])
with open("gpurun_out/ego_pmc_summary.txt", "a") as o:
    for k, (c, v) in acc.items():
        o.write("%s per launch %.6g (launch records %d)\n" % (k, v / max(c, 1), c))
PY
done
cat gpurun_out/ego_kernel_stats.csv gpurun_out/ego_pmc_summary.txt
