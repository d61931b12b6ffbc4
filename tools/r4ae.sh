#!/bin/bash
# 1-bit classification tiles at 1/8 of the resolution (BCP_NEAR_SHIFT=3) against the default 1/4: pool traffic and time, C4 time
O=gpurun_out/r4ae; mkdir -p $O; rm -f $O/*
BCP_NEAR_SHIFT=3 python -m pytest tests/test_gpu_parity.py tests/test_gpu_c4_full.py tests/test_gpu_pool.py -m gpu -x -q --timeout 300 > $O/tests3.log 2>&1; echo "tests shift 3 rc=$?" >> $O/rc.txt
if grep -q "Memory access fault" $O/tests3.log; then echo FAULT; tail -n 20 $O/tests3.log; exit 1; fi
tail -n 3 $O/tests3.log
for sh in 2 3; do
  echo "== shift $sh" >> $O/configs.txt
  BCP_NEAR_SHIFT=$sh python tools/bench_configs.py 2>&1 | grep "exact_mode': 0}" >> $O/configs.txt
  BCP_NEAR_SHIFT=$sh python tools/bench_pool.py 65536 65536 1 2>&1 | grep "ms/step" | head -1 >> $O/configs.txt
  BCP_NEAR_SHIFT=$sh bash tools/pool_pmc.sh > $O/pool_pmc_$sh.log 2>&1
  echo "== shift $sh" >> $O/pool.txt; grep "bytes_per_env_step" $O/pool_pmc_$sh.log >> $O/pool.txt; grep step_local gpurun_out/pool_kernel_stats.csv | cut -c1-120 >> $O/pool.txt
done
cat $O/rc.txt $O/configs.txt $O/pool.txt
