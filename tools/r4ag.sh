#!/bin/bash
# initial states of the envs that may end, fetched while the mover waits for verdicts: whole suite, A/B step times, C4 / pool configs
O=gpurun_out/r4ag; mkdir -p $O; rm -f $O/*
timeout -k 10 600 python -m pytest tests -m gpu -x -q --timeout 300 > $O/tests.log 2>&1; echo "tests rc=$?" > $O/rc.txt
if grep -q "Memory access fault" $O/tests.log; then echo FAULT; tail -n 20 $O/tests.log; exit 1; fi
if ! grep -q "rc=0" $O/rc.txt; then tail -n 30 $O/tests.log; exit 1; fi
for rep in 1 2 3; do
  for lib in tools/libbcplan_base.so -; do python tools/step_time.py $lib 2>&1 | grep n=65536 >> $O/step_time.txt; done
done
for rep in 1 2; do for lib in tools/libbcplan_base.so bc_gym_planning_env_amd/libbcplan.so; do echo "== $lib" >> $O/configs.txt; BCP_LIB=$lib python tools/bench_configs.py 2>&1 | grep "exact_mode': 0}" >> $O/configs.txt; BCP_LIB=$lib python tools/bench_pool.py 65536 65536 1 2>&1 | grep "ms/step" | head -1 >> $O/configs.txt; done; done
cat $O/rc.txt; tail -n 1 $O/tests.log; cat $O/step_time.txt $O/configs.txt
