#!/bin/bash
# delay queues: slot = (k - 1) % delay without the integer division (fifo_slot): delay / state / rollout tests, digests of a delayed
# configuration under both builds, step rates
O=gpurun_out/r4ax; mkdir -p $O; rm -f $O/*
timeout -k 10 600 python -m pytest tests/test_gpu_delays.py tests/test_gpu_state.py tests/test_gpu_rollout.py tests/test_gpu_parity.py -m gpu -x -q --timeout 300 > $O/tests.log 2>&1; echo "tests rc=$?" > $O/rc.txt
if grep -q "Memory access fault" $O/tests.log; then echo FAULT; tail -n 20 $O/tests.log; exit 1; fi
if ! grep -q "rc=0" $O/rc.txt; then tail -n 30 $O/tests.log; exit 1; fi
for lib in tools/libbcplan_v8.so bc_gym_planning_env_amd/libbcplan.so; do echo "== $lib" >> $O/delays.txt; BCP_LIB=$lib python tools/bench_delays.py 2>&1 | grep -E "ms/step|env-steps" >> $O/delays.txt; done
python tests/soak.py 600 16384 19 > $O/soak.txt 2>&1; echo "soak rc=$?" >> $O/rc.txt
cat $O/rc.txt; tail -n 1 $O/tests.log; cat $O/delays.txt; grep -E "delays|pursuit|soak ok" $O/soak.txt
