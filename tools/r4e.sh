#!/bin/bash
O=gpurun_out/r4e; mkdir -p $O
timeout -k 5 120 tools/dispatch_rate > $O/dispatch_rate.txt 2>&1
python -m pytest tests/test_gpu_parity.py tests/test_gpu_state.py -m gpu -x -q --timeout 300 > $O/tests.log 2>&1; echo "tests rc=$?" > $O/rc.txt
for lib in tools/libbcplan_a5488a0.so - tools/libbcplan_a5488a0.so -; do python tools/step_time.py $lib >> $O/step_time.txt 2>&1; done
python tools/bench_lib.py - >> $O/bench_lib.txt 2>&1
cat $O/rc.txt; tail -3 $O/tests.log; grep -h "n=65536\|ms_per_step" $O/step_time.txt $O/bench_lib.txt
