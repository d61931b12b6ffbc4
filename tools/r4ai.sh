#!/bin/bash
# profiles of the round's last build, part 1: whole GPU suite, the default bench line, C3 kernel trace / aux trace / FETCH + WRITE passes
# with calibration, C4 trace and passes, pool passes
O=gpurun_out/r4ai; mkdir -p $O; rm -f $O/*
timeout -k 10 600 python -m pytest tests -m gpu -x -q --timeout 300 > $O/tests.log 2>&1; echo "tests rc=$?" > $O/rc.txt
if grep -q "Memory access fault" $O/tests.log; then echo FAULT; tail -n 20 $O/tests.log; exit 1; fi
if ! grep -q "rc=0" $O/rc.txt; then tail -n 30 $O/tests.log; exit 1; fi
python bench.py > $O/bench_default.json 2> $O/bench.err; echo "bench rc=$?" >> $O/rc.txt
bash tools/step_pmc.sh > $O/step_pmc.log 2>&1; echo "step_pmc rc=$?" >> $O/rc.txt
bash tools/c4_pmc.sh > $O/c4_pmc.log 2>&1; echo "c4_pmc rc=$?" >> $O/rc.txt
bash tools/pool_pmc.sh > $O/pool_pmc.log 2>&1; echo "pool_pmc rc=$?" >> $O/rc.txt
cat $O/rc.txt; tail -n 1 $O/tests.log; cut -c1-600 $O/bench_default.json; cat gpurun_out/step_kernel_stats.csv gpurun_out/c4_kernel_stats.csv gpurun_out/pool_kernel_stats.csv | cut -c1-130; tail -n 5 $O/pool_pmc.log
