#!/bin/bash
# the current build once more where the pool's path kernel matters: pool / sharding tests, endless soak, one refresh
O=gpurun_out/r4ar; mkdir -p $O; rm -f $O/*
timeout -k 10 600 python -m pytest tests/test_gpu_pool.py tests/test_gpu_sharding.py tests/test_gpu_state.py -m gpu -x -q --timeout 300 > $O/tests.log 2>&1; echo "tests rc=$?" > $O/rc.txt
if grep -q "Memory access fault" $O/tests.log; then echo FAULT; tail -n 20 $O/tests.log; exit 1; fi
if ! grep -q "rc=0" $O/rc.txt; then tail -n 30 $O/tests.log; exit 1; fi
timeout -k 10 300 python tools/soak_endless.py > $O/soak.txt 2>&1; echo "soak rc=$?" >> $O/rc.txt
python tools/bench_endless.py 2>&1 | grep -E "side stream|high-priority|one refresh|status" > $O/endless.txt
strings bc_gym_planning_env_amd/libbcplan.so | grep -c "mini_world_paths_kernel" >> $O/rc.txt
cat $O/rc.txt; tail -n 1 $O/tests.log; tail -n 3 $O/soak.txt; cat $O/endless.txt
