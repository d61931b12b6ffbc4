#!/bin/bash
# step counter among the launch arguments: tests that touch the counter (noise, graphs, rollouts, shards), then A/B step times
O=gpurun_out/r4ab; mkdir -p $O; rm -f $O/*
python -m pytest tests/test_gpu_noise.py tests/test_gpu_state.py tests/test_gpu_parity.py tests/test_gpu_rollout.py tests/test_gpu_delays.py tests/test_gpu_sharding.py tests/test_gpu_pool.py -m gpu -x -q --timeout 300 > $O/tests.log 2>&1; echo "tests rc=$?" > $O/rc.txt
if grep -q "Memory access fault" $O/tests.log; then echo FAULT; tail -n 20 $O/tests.log; exit 1; fi
if ! grep -q "rc=0" $O/rc.txt; then tail -n 30 $O/tests.log; exit 1; fi
for rep in 1 2 3; do
  python tools/step_time.py tools/libbcplan_base.so 2>&1 | grep n=65536 >> $O/step_time.txt
  python tools/step_time.py - 2>&1 | grep n=65536 >> $O/step_time.txt
  BCP_TICK_ON_DEVICE=1 python tools/step_time.py - 2>&1 | grep n=65536 | sed 's/^/tick on device: /' >> $O/step_time.txt
done
for rep in 1 2; do for lib in tools/libbcplan_base.so bc_gym_planning_env_amd/libbcplan.so; do echo "== $lib" >> $O/configs.txt; BCP_LIB=$lib python tools/bench_configs.py 2>&1 | grep "exact_mode': 0}" >> $O/configs.txt; BCP_LIB=$lib python tools/bench_pool.py 65536 65536 1 2>&1 | grep "ms/step" | head -3 >> $O/configs.txt; done; done
cat $O/rc.txt; tail -n 1 $O/tests.log; cat $O/step_time.txt $O/configs.txt
