#!/bin/bash
# world sampler at 128 / 64 VGPRs (amdgpu_waves_per_eu 4 / 8, spilling) against the 174-VGPR build: endless pool step rate, one refresh,
# and the longest step launch of a traced run
O=gpurun_out/r4am; mkdir -p $O; rm -f $O/*
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for v in base s4 s8; do
  lib=tools/libbcplan_$v.so; if [ $v = base ]; then lib=bc_gym_planning_env_amd/libbcplan.so; fi
  echo "== $v" >> $O/endless.txt
  BCP_LIB=$lib python tools/bench_endless.py 2>&1 | grep -E "ms/step|one refresh|status" >> $O/endless.txt
  BCP_LIB=$lib rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r4am_$v -o p -- python3 tools/bench_endless.py 65536 4 128 1024 > $O/trace_$v.log 2>&1
  echo "== $v" >> $O/trace.txt; grep -E "step_local|mini_world_sample" gpurun_out/r4am_$v/p_kernel_stats.csv | cut -c1-60,150-260 >> $O/trace.txt
done
cat $O/endless.txt $O/trace.txt
