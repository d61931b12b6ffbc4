#!/bin/bash
# instruction-scheduling strategies of the compiler (-mllvm -amdgpu-sched-strategy=max-ilp / max-memory-clause, -amdgpu-schedule-metric-bias=0)
# against the default build: C3 step time, C4 / pool configs
O=gpurun_out/r4as; mkdir -p $O; rm -f $O/*
for rep in 1 2 3; do
  for lib in - tools/libbcplan_f_ilp.so tools/libbcplan_f_clause.so tools/libbcplan_f_bias0.so; do python tools/step_time.py $lib 2>&1 | grep n=65536 >> $O/step_time.txt; done
done
for lib in bc_gym_planning_env_amd/libbcplan.so tools/libbcplan_f_ilp.so tools/libbcplan_f_clause.so tools/libbcplan_f_bias0.so; do echo "== $lib" >> $O/configs.txt; BCP_LIB=$lib python tools/bench_configs.py 2>&1 | grep "exact_mode': 0}" >> $O/configs.txt; BCP_LIB=$lib python tools/bench_pool.py 65536 65536 1 2>&1 | grep "ms/step" | head -1 >> $O/configs.txt; done
cut -c1-110 $O/step_time.txt; cat $O/configs.txt
