#!/bin/bash
# BASELINE configs[3] (C4) under rocprofv3: kernel trace + stats, then FETCH_SIZE / WRITE_SIZE passes (one run each),
# converted to bytes with tools/pmc_calib's factors (run tools/step_pmc.sh first: it leaves the calibration passes in
# gpurun_out/calib_pmc_*).  Writes gpurun_out/c4_kernel_stats.csv and gpurun_out/c4_pmc.json -> profiles/r<round>_c4_*.
set -e
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/c4_trace -o p -- python3 tools/pmc_run_c4.py > gpurun_out/c4_trace.log 2>&1
grep -E '^"Name"|step_|edt_|pack_bitmap|path_' gpurun_out/c4_trace/p_kernel_stats.csv > gpurun_out/c4_kernel_stats.csv
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --output-format csv --pmc $c -d gpurun_out/c4_pmc_$c -o p -- python3 tools/pmc_run_c4.py > gpurun_out/c4_pmc_$c.log 2>&1
done
python3 tools/pmc_summary.py c4 gpurun_out/c4_pmc_ gpurun_out/calib_pmc_ gpurun_out/c4_pmc.json 65536 $((4183 * 65536))
cat gpurun_out/c4_kernel_stats.csv
