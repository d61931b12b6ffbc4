#!/bin/bash
# edge set-up of the sparse exact test without integer / float64 divisions, extents by packed 16-bit reductions: parity, A/B, timeline
O=gpurun_out/r4at; mkdir -p $O; rm -f $O/*
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_seams.py tests/test_gpu_c4_full.py tests/test_gpu_pool.py tests/test_gpu_rollout.py tests/test_gpu_delays.py -m gpu -x -q --timeout 300 > $O/tests.log 2>&1; echo "tests rc=$?" > $O/rc.txt
if grep -q "Memory access fault" $O/tests.log; then echo FAULT; tail -n 20 $O/tests.log; exit 1; fi
if ! grep -q "rc=0" $O/rc.txt; then tail -n 30 $O/tests.log; exit 1; fi
for rep in 1 2 3; do
  for lib in tools/libbcplan_v0.so -; do python tools/step_time.py $lib 2>&1 | grep n=65536 >> $O/step_time.txt; done
done
for rep in 1 2; do for lib in tools/libbcplan_v0.so bc_gym_planning_env_amd/libbcplan.so; do echo "== $lib" >> $O/configs.txt; BCP_LIB=$lib python tools/bench_configs.py 2>&1 | grep "exact_mode': 0}" >> $O/configs.txt; BCP_LIB=$lib python tools/bench_pool.py 65536 65536 1 2>&1 | grep "ms/step" | head -1 >> $O/configs.txt; done; done
python tools/diag_local.py > $O/diag.txt 2>&1
cat $O/rc.txt; tail -n 1 $O/tests.log; cut -c1-110 $O/step_time.txt; cat $O/configs.txt; grep -E "exact test of helper|ticket -> verdict|out of tickets" $O/diag.txt
