#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for tag in r3 cur r3b curb; do
  lib=tools/libbcplan_a5488a0.so; if [ $tag = cur ] || [ $tag = curb ]; then lib=-; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ab_trace_$tag -o p -- python3 tools/step_time.py $lib > gpurun_out/ab_trace_$tag.log 2>&1
  grep -E "step_local" gpurun_out/ab_trace_$tag/p_kernel_stats.csv | cut -c1-140
  grep "n=65536" gpurun_out/ab_trace_$tag.log
done
