#!/bin/bash
# lane / G and 64 / G of the sparse exact test as shifts: parity, digest against the previous build, A/B step times
O=gpurun_out/r4aw; mkdir -p $O; rm -f $O/*
python tools/state_digest.py tools/libbcplan_v7.so > $O/digest.txt 2>&1
python tools/state_digest.py - >> $O/digest.txt 2>&1
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_c4_full.py tests/test_gpu_pool.py -m gpu -x -q --timeout 300 > $O/tests.log 2>&1; echo "tests rc=$?" > $O/rc.txt
if grep -q "Memory access fault" $O/tests.log; then echo FAULT; tail -n 20 $O/tests.log; exit 1; fi
if ! grep -q "rc=0" $O/rc.txt; then tail -n 30 $O/tests.log; exit 1; fi
for rep in 1 2 3; do
  for lib in tools/libbcplan_v7.so -; do python tools/step_time.py $lib 2>&1 | grep n=65536 >> $O/step_time.txt; done
done
for lib in tools/libbcplan_v7.so bc_gym_planning_env_amd/libbcplan.so; do echo "== $lib" >> $O/configs.txt; BCP_LIB=$lib python tools/bench_configs.py 2>&1 | grep "exact_mode': 0}" >> $O/configs.txt; BCP_LIB=$lib python tools/bench_pool.py 65536 65536 1 2>&1 | grep "ms/step" | head -1 >> $O/configs.txt; done
cat $O/rc.txt; tail -n 1 $O/tests.log; grep digest $O/digest.txt; cut -c1-112 $O/step_time.txt; cat $O/configs.txt
