#!/bin/bash
# last validation of the round: whole GPU suite, the default bench line, kernel trace of bench.py with its aux legs
O=gpurun_out/r4aq; mkdir -p $O; rm -f $O/*
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests -m gpu -x -q --timeout 300 > $O/tests.log 2>&1; echo "tests rc=$?" > $O/rc.txt
if grep -q "Memory access fault" $O/tests.log; then echo FAULT; tail -n 20 $O/tests.log; exit 1; fi
if ! grep -q "rc=0" $O/rc.txt; then tail -n 30 $O/tests.log; exit 1; fi
python bench.py > $O/bench_default.json 2> $O/bench.err; echo "bench rc=$?" >> $O/rc.txt
KEEP='^"Name"|step_|ego_|goal_n|mini_world|edt_|pack_bitmap|path_|near_|calib_'
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/aux_trace -o p -- python3 bench.py --no-cpu-baseline > gpurun_out/aux_trace.log 2>&1
grep -E "$KEEP" gpurun_out/aux_trace/p_kernel_stats.csv > gpurun_out/aux_kernel_stats.csv
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; echo "smoke rc=$?" >> $O/rc.txt
cat $O/rc.txt; tail -n 1 $O/tests.log; tail -n 1 $O/smoke.txt; python tools/bench_summary.py $O/bench_default.json; cut -c1-150 gpurun_out/aux_kernel_stats.csv
