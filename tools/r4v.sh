#!/bin/bash
# A/B of the one-sector path record against ${BASE:-tools/libbcplan_base.so}: parity tests, C3 step time, C4 / pool configs, pool PMC
O=gpurun_out/r4v; mkdir -p $O; rm -f $O/*
python -m pytest tests -m gpu -x -q --timeout 300 > $O/tests.log 2>&1; echo "tests rc=$?" > $O/rc.txt
if grep -q "Memory access fault" $O/tests.log; then echo FAULT; tail -n 20 $O/tests.log; exit 1; fi
if ! grep -q "rc=0" $O/rc.txt; then tail -n 30 $O/tests.log; exit 1; fi
for rep in 1 2; do for lib in ${BASE:-tools/libbcplan_base.so} -; do python tools/step_time.py $lib 2>&1 | grep n=65536 >> $O/step_time.txt; done; done
for rep in 1 2; do for lib in ${BASE:-tools/libbcplan_base.so} bc_gym_planning_env_amd/libbcplan.so; do echo "== $lib" >> $O/configs.txt; BCP_LIB=$lib python tools/bench_configs.py 2>&1 | grep -v amdgpu >> $O/configs.txt; done; done
cat $O/rc.txt; tail -n 3 $O/tests.log; cat $O/step_time.txt $O/configs.txt
bash tools/pool_pmc.sh > $O/pool_pmc.log 2>&1; tail -n 12 $O/pool_pmc.log
