#!/bin/bash
# the tree as committed at the end of round 4: whole GPU suite, smoke, the default bench line
O=gpurun_out/r4ay; mkdir -p $O; rm -f $O/*
timeout -k 10 600 python -m pytest tests -m gpu -x -q --timeout 300 > $O/tests.log 2>&1; echo "tests rc=$?" > $O/rc.txt
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; echo "smoke rc=$?" >> $O/rc.txt
python bench.py --no-aux > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" >> $O/rc.txt
cat $O/rc.txt; tail -n 1 $O/tests.log; tail -n 1 $O/smoke.txt; cut -c1-260 $O/bench.json
