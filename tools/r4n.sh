#!/bin/bash
# round-4 validation session: the whole GPU suite under each workgroup size of step_local_kernel, bench.py, and the
# co-residency cases the smaller workgroups were built for (endless pool refresh, the RCCL branch at world size 1)
O=gpurun_out/r4n; mkdir -p $O
python -m pytest tests -m gpu -x -q --timeout 300 > $O/gpu_tests.log 2>&1; echo "tests rc=$?" > $O/rc.txt
BCP_LOCAL_PAIRS=2 python -m pytest tests -m gpu -x -q --timeout 300 > $O/gpu_tests_p2.log 2>&1; echo "tests pairs=2 rc=$?" >> $O/rc.txt
BCP_LOCAL_PAIRS=1 python -m pytest tests -m gpu -x -q --timeout 300 > $O/gpu_tests_p1.log 2>&1; echo "tests pairs=1 rc=$?" >> $O/rc.txt
python bench.py > $O/bench_default.json 2> $O/bench.err; echo "bench rc=$?" >> $O/rc.txt
for p in 4 2; do echo "== pairs $p" >> $O/endless.txt; BCP_LOCAL_PAIRS=$p python tools/bench_endless.py 2>&1 | grep -v amdgpu >> $O/endless.txt; done
for p in 4 2; do for r in 8 128; do echo "== pairs $p gather-every $r" >> $O/rccl_ws1.txt; BCP_LOCAL_PAIRS=$p BCP_DIST_FORCE=1 python bench.py --gpus 1 --gather-every $r --no-aux --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['timed_region']['device_ms_per_step'], d.get('gather_every_8'))" >> $O/rccl_ws1.txt; done; done
cat $O/rc.txt; tail -2 $O/gpu_tests.log $O/gpu_tests_p2.log $O/gpu_tests_p1.log
