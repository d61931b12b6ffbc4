#!/bin/bash
# final validation + profiles of a round: GPU suite (default workgroup size, then 8- and 4-wave), bench.py, all profile passes
O=gpurun_out/r4t; mkdir -p $O; rm -f $O/*
python -m pytest tests -m gpu -x -q --timeout 300 > $O/gpu_tests.log 2>&1; echo "tests rc=$?" > $O/rc.txt
if grep -q "Memory access fault" $O/gpu_tests.log; then echo FAULT; tail -n 20 $O/gpu_tests.log; exit 1; fi
BCP_LOCAL_PAIRS=2 python -m pytest tests -m gpu -x -q --timeout 300 > $O/gpu_tests_p2.log 2>&1; echo "tests pairs=2 rc=$?" >> $O/rc.txt
BCP_LOCAL_PAIRS=1 python -m pytest tests -m gpu -x -q --timeout 300 > $O/gpu_tests_p1.log 2>&1; echo "tests pairs=1 rc=$?" >> $O/rc.txt
python bench.py > $O/bench_default.json 2> $O/bench.err; echo "bench rc=$?" >> $O/rc.txt
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/rc.txt
bash tools/r4q.sh >> $O/rc.txt 2>&1
cat $O/rc.txt; for f in gpu_tests gpu_tests_p2 gpu_tests_p1; do tail -n 1 $O/$f.log; done; python tools/bench_summary.py $O/bench_default.json
