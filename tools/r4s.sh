#!/bin/bash
O=gpurun_out/r4s; mkdir -p $O; rm -f $O/*
for rep in 1 2 3; do for lib in tools/libbcplan_base.so -; do python tools/step_time.py $lib 2>&1 | grep n=65536 >> $O/step_time.txt; done; done
for rep in 1 2; do for lib in tools/libbcplan_base.so bc_gym_planning_env_amd/libbcplan.so; do BCP_LIB=$lib python tools/bench_configs.py 2>&1 | grep "exact_mode': 0}" >> $O/configs.txt; done; done
cat $O/step_time.txt $O/configs.txt
