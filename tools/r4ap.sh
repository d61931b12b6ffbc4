#!/bin/bash
# pack_bitmap_kernel with 16-byte loads: tests that go through the lethal masks, endless pool / one refresh against the round's base library
O=gpurun_out/r4ap; mkdir -p $O; rm -f $O/*
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_pool.py tests/test_gpu_c4_full.py tests/test_gpu_seams.py tests/test_gpu_egocentric.py -m gpu -x -q --timeout 300 > $O/tests.log 2>&1; echo "tests rc=$?" > $O/rc.txt
if grep -q "Memory access fault" $O/tests.log; then echo FAULT; tail -n 20 $O/tests.log; exit 1; fi
if ! grep -q "rc=0" $O/rc.txt; then tail -n 30 $O/tests.log; exit 1; fi
for rep in 1 2; do for lib in tools/libbcplan_base.so bc_gym_planning_env_amd/libbcplan.so; do
  echo "== $lib" >> $O/endless.txt
  BCP_LIB=$lib python tools/bench_endless.py 2>&1 | grep -E "side stream|high-priority|one refresh|status" >> $O/endless.txt
done; done
python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" >> $O/rc.txt
cat $O/rc.txt; tail -n 1 $O/tests.log; cat $O/endless.txt; python tools/bench_summary.py $O/bench.json
