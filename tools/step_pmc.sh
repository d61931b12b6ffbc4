#!/bin/bash
# rocprofv3 over the metric workload (run on the GPU box from the repo root):
#   1. kernel trace + stats of `python3 bench.py` without / with its aux legs;
#   2. FETCH_SIZE and WRITE_SIZE passes (one run each: they do not fit one pass) over tools/pmc_run.py AND over
#      tools/pmc_calib, the known-byte-count kernels in the step's own access pattern.
# The counters are turned into bytes with the factors the calibration kernels give for THIS access pattern
# (bytes actually moved / counter value of calib_step), as MI355X_MICROARCH.md's HBM section prescribes.
# Writes gpurun_out/step_kernel_stats.csv, gpurun_out/aux_kernel_stats.csv, gpurun_out/step_pmc.json; copy them to
# profiles/ (r<round>_bench_kernel_stats.csv, r<round>_aux_kernel_stats.csv, r<round>_pmc_summary.json).
set -e
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
KEEP='^"Name"|step_|ego_|goal_n|mini_world|edt_|pack_bitmap|path_|calib_'
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/step_trace -o p -- python3 bench.py --no-cpu-baseline --no-aux > gpurun_out/step_trace.log 2>&1
grep -E "$KEEP" gpurun_out/step_trace/p_kernel_stats.csv > gpurun_out/step_kernel_stats.csv
grep "^{\"metric" gpurun_out/step_trace.log > gpurun_out/step_trace_bench_line.json || true
if [ "$1" != "--no-aux" ]; then
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/aux_trace -o p -- python3 bench.py --no-cpu-baseline > gpurun_out/aux_trace.log 2>&1
  grep -E "$KEEP" gpurun_out/aux_trace/p_kernel_stats.csv > gpurun_out/aux_kernel_stats.csv
fi
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --output-format csv --pmc $c -d gpurun_out/step_pmc_$c -o p -- python3 tools/pmc_run.py > gpurun_out/step_pmc_$c.log 2>&1
  rocprofv3 --kernel-trace --output-format csv --pmc $c -d gpurun_out/calib_pmc_$c -o p -- tools/pmc_calib > gpurun_out/calib_pmc_$c.log 2>&1
done
python3 tools/pmc_summary.py step gpurun_out/step_pmc_ gpurun_out/calib_pmc_ gpurun_out/step_pmc.json
cat gpurun_out/step_kernel_stats.csv
