#!/bin/bash
# world sampler: the two pose_collides of a candidate by the sparse exact test (cells under the image one by one) instead of the rasteriser
O=gpurun_out/r4ao; mkdir -p $O; rm -f $O/*
timeout -k 10 600 python -m pytest tests/test_gpu_pool.py tests/test_gpu_sharding.py -m gpu -x -q --timeout 300 > $O/tests.log 2>&1; echo "tests rc=$?" > $O/rc.txt
if grep -q "Memory access fault" $O/tests.log; then echo FAULT; tail -n 20 $O/tests.log; exit 1; fi
if ! grep -q "rc=0" $O/rc.txt; then tail -n 30 $O/tests.log; exit 1; fi
timeout -k 10 300 python tools/soak_endless.py > $O/soak.txt 2>&1; echo "soak rc=$?" >> $O/rc.txt
for rep in 1 2; do for lib in tools/libbcplan_base.so bc_gym_planning_env_amd/libbcplan.so; do
  echo "== $lib" >> $O/endless.txt
  BCP_LIB=$lib python tools/bench_endless.py 2>&1 | grep -E "setup|side stream|high-priority|one refresh|status" >> $O/endless.txt
done; done
python tools/sampler_latency.py > $O/latency.txt 2>&1
cat $O/rc.txt; tail -n 3 $O/soak.txt; cat $O/endless.txt; tail -n 8 $O/latency.txt
