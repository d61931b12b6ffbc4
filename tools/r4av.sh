#!/bin/bash
# divisions by dt, L and pi as Markstein's sequence (div_by_const): same bits as the divisions (digest against the previous build),
# parity tests, A/B step times, C4 / pool configs, timeline
O=gpurun_out/r4av; mkdir -p $O; rm -f $O/*
python tools/state_digest.py tools/libbcplan_v6.so > $O/digest.txt 2>&1
python tools/state_digest.py - >> $O/digest.txt 2>&1
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_seams.py tests/test_gpu_delays.py tests/test_gpu_rollout.py tests/test_gpu_state.py tests/test_gpu_noise.py -m gpu -x -q --timeout 300 > $O/tests.log 2>&1; echo "tests rc=$?" > $O/rc.txt
if grep -q "Memory access fault" $O/tests.log; then echo FAULT; tail -n 20 $O/tests.log; exit 1; fi
if ! grep -q "rc=0" $O/rc.txt; then tail -n 30 $O/tests.log; exit 1; fi
for rep in 1 2 3; do
  for lib in tools/libbcplan_v6.so -; do python tools/step_time.py $lib 2>&1 | grep n=65536 >> $O/step_time.txt; done
done
for lib in tools/libbcplan_v6.so -; do python tools/step_time.py $lib 1048576 2>&1 | grep n=1048576 >> $O/step_time.txt; done
for lib in tools/libbcplan_v6.so bc_gym_planning_env_amd/libbcplan.so; do echo "== $lib" >> $O/configs.txt; BCP_LIB=$lib python tools/bench_configs.py 2>&1 | grep "exact_mode': 0}" >> $O/configs.txt; BCP_LIB=$lib python tools/bench_pool.py 65536 65536 1 2>&1 | grep "ms/step" | head -1 >> $O/configs.txt; done
cat $O/rc.txt; tail -n 1 $O/tests.log; grep digest $O/digest.txt; cut -c1-112 $O/step_time.txt; cat $O/configs.txt
