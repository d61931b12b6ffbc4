#!/bin/bash
# state of the tree on a fresh box: whole GPU suite, the default bench line, C4 / pool quick figures
O=gpurun_out/r4aa; mkdir -p $O; rm -f $O/*
python -m pytest tests -m gpu -x -q --timeout 300 > $O/tests.log 2>&1; echo "tests rc=$?" > $O/rc.txt
if grep -q "Memory access fault" $O/tests.log; then echo FAULT; tail -n 20 $O/tests.log; exit 1; fi
if ! grep -q "rc=0" $O/rc.txt; then tail -n 30 $O/tests.log; exit 1; fi
python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" >> $O/rc.txt
python tools/step_time.py - 2>&1 | grep n=65536 >> $O/step_time.txt
cat $O/rc.txt; tail -n 2 $O/tests.log; cat $O/step_time.txt; cut -c1-1500 $O/bench.json
