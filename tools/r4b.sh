#!/bin/bash
# round-4 A/B session: see DESIGN.md (workgroup sizes of step_local_kernel, the cost of a second atomic, ego legs)
O=gpurun_out/r4b; mkdir -p $O
python -m pytest tests/test_gpu_egocentric.py tests/test_gpu_parity.py tests/test_gpu_pool.py -m gpu -x -q --timeout 300 > $O/tests.log 2>&1; echo "tests rc=$?" > $O/rc.txt
for lib in tools/libbcplan_a5488a0.so tools/libbcplan_HEAD.so - tools/libbcplan_a5488a0.so -; do python tools/step_time.py $lib >> $O/step_time.txt 2>&1; done
for p in 4 2 1 4 2 1; do BCP_PAIRS=$p python tools/step_time.py - >> $O/step_time_pairs.txt 2>&1; done
python tools/bench_lib.py - >> $O/bench_lib.txt 2>&1
BCP_LOCAL_PAIRS=2 python tools/bench_lib.py - >> $O/bench_lib.txt 2>&1
BCP_LOCAL_PAIRS=1 python tools/bench_lib.py - >> $O/bench_lib.txt 2>&1
for p in 4 2 1; do echo "pairs $p" >> $O/n_sweep.txt; BCP_PAIRS=$p python tools/n_sweep.py - >> $O/n_sweep.txt 2>&1; done
python tools/bench_ego_aisle.py > $O/ego_colored.txt 2>&1
python tools/bench_ego_aisle.py g10_ego_aisle.npz > $O/ego_aisle.txt 2>&1
python tools/bench_ego.py > $O/ego_mini.txt 2>&1
python tools/bench_ego_cells.py > $O/ego_cells.txt 2>&1
cat $O/rc.txt; tail -3 $O/tests.log
