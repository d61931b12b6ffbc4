#!/bin/bash
# coarse near tiles (BCP_NEAR_SHIFT 0 / 1 / 2): tests, C4 and pool time, pool traffic
O=gpurun_out/r4y; mkdir -p $O; rm -f $O/*
python -m pytest tests -m gpu -x -q --timeout 300 > $O/tests.log 2>&1; echo "tests rc=$?" > $O/rc.txt
if grep -q "Memory access fault" $O/tests.log; then echo FAULT; tail -n 20 $O/tests.log; exit 1; fi
if ! grep -q "rc=0" $O/rc.txt; then tail -n 30 $O/tests.log; exit 1; fi
BCP_NEAR_SHIFT=2 python -m pytest tests/test_gpu_parity.py tests/test_gpu_c4_full.py tests/test_gpu_pool.py tests/test_gpu_rollout.py -m gpu -x -q --timeout 300 > $O/tests2.log 2>&1; echo "tests shift 2 rc=$?" >> $O/rc.txt
if grep -q "Memory access fault" $O/tests2.log; then echo FAULT; tail -n 20 $O/tests2.log; exit 1; fi
tail -n 3 $O/tests2.log
for lib in ${BASE:-tools/libbcplan_base.so} -; do python tools/step_time.py $lib 2>&1 | grep n=65536 >> $O/step_time.txt; done
for sh in 0 1 2; do
  echo "== shift $sh" >> $O/configs.txt
  BCP_NEAR_SHIFT=$sh python tools/bench_configs.py 2>&1 | grep "exact_mode': 0}" >> $O/configs.txt
  BCP_NEAR_SHIFT=$sh bash tools/pool_pmc.sh > $O/pool_pmc_$sh.log 2>&1
  echo "== shift $sh" >> $O/pool.txt; grep "bytes_per_env_step" $O/pool_pmc_$sh.log >> $O/pool.txt; grep step_local gpurun_out/pool_kernel_stats.csv | cut -c1-120 >> $O/pool.txt
done
cat $O/rc.txt; tail -n 3 $O/tests.log; cat $O/step_time.txt $O/configs.txt $O/pool.txt
