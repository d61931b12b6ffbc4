#!/bin/bash
O=gpurun_out/r4o; mkdir -p $O; rm -f $O/*
timeout -k 10 300 python -m pytest tests/test_gpu_rollout.py -m gpu -x -q --timeout 120 > $O/tests.log 2>&1; rc=$?; echo "tests rc=$rc" > $O/rc.txt
if grep -q "Memory access fault" $O/tests.log; then echo "FAULT in tests"; tail -n 20 $O/tests.log; exit 1; fi
if [ $rc -ne 0 ]; then tail -n 30 $O/tests.log; exit 1; fi
timeout -k 10 300 python tools/bench_rollout.py > $O/rollout.txt 2>&1 || { tail -n 5 $O/rollout.txt; exit 1; }
timeout -k 10 300 python -m pytest tests/test_gpu_egocentric.py -m gpu -x -q --timeout 300 > $O/tests_ego.log 2>&1; echo "ego tests rc=$?" >> $O/rc.txt
python tools/bench_ego_aisle.py > $O/ego_colored.txt 2>&1
cat $O/rc.txt; tail -n 3 $O/tests.log $O/tests_ego.log; grep -v amdgpu $O/ego_colored.txt $O/rollout.txt | tail -n 12
