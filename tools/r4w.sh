#!/bin/bash
# traffic check after a C4 / pool change: calibration + step PMC (no aux trace), then C4 and pool PMC
bash tools/step_pmc.sh --no-aux > gpurun_out/r4w_step_pmc.log 2>&1; echo "step_pmc rc=$?"
bash tools/c4_pmc.sh > gpurun_out/r4w_c4_pmc.log 2>&1; echo "c4_pmc rc=$?"
bash tools/pool_pmc.sh > gpurun_out/r4w_pool_pmc.log 2>&1; echo "pool_pmc rc=$?"
cat gpurun_out/step_pmc.json gpurun_out/c4_pmc.json gpurun_out/pool_pmc.json 2>/dev/null
grep step_local gpurun_out/c4_kernel_stats.csv gpurun_out/pool_kernel_stats.csv gpurun_out/step_kernel_stats.csv | cut -c1-160
