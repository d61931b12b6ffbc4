#!/bin/bash
# profiles of the round's LAST build: whole suite, bench line, C3 / C4 / pool traces and FETCH + WRITE passes, SQ counters, N sweep, soak
O=gpurun_out/r4au; mkdir -p $O; rm -f $O/*
timeout -k 10 600 python -m pytest tests -m gpu -x -q --timeout 300 > $O/tests.log 2>&1; echo "tests rc=$?" > $O/rc.txt
if grep -q "Memory access fault" $O/tests.log; then echo FAULT; tail -n 20 $O/tests.log; exit 1; fi
if ! grep -q "rc=0" $O/rc.txt; then tail -n 30 $O/tests.log; exit 1; fi
python bench.py > $O/bench_default.json 2> $O/bench.err; echo "bench rc=$?" >> $O/rc.txt
bash tools/step_pmc.sh > $O/step_pmc.log 2>&1; echo "step_pmc rc=$?" >> $O/rc.txt
bash tools/c4_pmc.sh > $O/c4_pmc.log 2>&1; echo "c4_pmc rc=$?" >> $O/rc.txt
bash tools/pool_pmc.sh > $O/pool_pmc.log 2>&1; echo "pool_pmc rc=$?" >> $O/rc.txt
bash tools/step_alu_pmc.sh > $O/alu_c3.log 2>&1; echo "alu c3 rc=$?" >> $O/rc.txt
bash tools/step_alu_pmc.sh c4 > $O/alu_c4.log 2>&1; echo "alu c4 rc=$?" >> $O/rc.txt
cp gpurun_out/step_alu_pmc*.json $O/ 2>/dev/null
python tools/n_sweep.py > $O/n_sweep.txt 2>&1; echo "n_sweep rc=$?" >> $O/rc.txt
python tests/soak.py 1500 32768 17 > $O/soak.txt 2>&1; echo "soak rc=$?" >> $O/rc.txt
python tools/soak_endless.py >> $O/soak.txt 2>&1; echo "soak endless rc=$?" >> $O/rc.txt
cat $O/rc.txt; tail -n 1 $O/tests.log; python tools/bench_summary.py $O/bench_default.json; cat gpurun_out/step_kernel_stats.csv gpurun_out/c4_kernel_stats.csv gpurun_out/pool_kernel_stats.csv | grep step_local | cut -c1-130; grep -v enqueued $O/soak.txt | tail -n 6
