#!/bin/bash
# world sampler in time slices: pool tests (bit-for-bit worlds against the host sampler / the oracle), endless soak, step rate of the
# endless pool and the longest step launch, for slice counts / lengths; BCP_SAMPLE_SLICES=0 is the single launch over the list
O=gpurun_out/r4an; mkdir -p $O; rm -f $O/*
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_pool.py tests/test_gpu_sharding.py -m gpu -x -q --timeout 300 > $O/tests.log 2>&1; echo "tests rc=$?" > $O/rc.txt
if grep -q "Memory access fault" $O/tests.log; then echo FAULT; tail -n 20 $O/tests.log; exit 1; fi
if ! grep -q "rc=0" $O/rc.txt; then tail -n 30 $O/tests.log; exit 1; fi
timeout -k 10 300 python tools/soak_endless.py > $O/soak.txt 2>&1; echo "soak rc=$?" >> $O/rc.txt
for cfg in "0 40" "40 40" "80 20" "20 80"; do
  set -- $cfg
  echo "== slices $1 x $2 us" >> $O/endless.txt
  BCP_SAMPLE_SLICES=$1 BCP_SAMPLE_SLICE_US=$2 python tools/bench_endless.py 2>&1 | grep -E "side stream|high-priority|one refresh|status" >> $O/endless.txt
  BCP_SAMPLE_SLICES=$1 BCP_SAMPLE_SLICE_US=$2 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r4an_$1_$2 -o p -- python3 tools/bench_endless.py 65536 4 128 1024 > $O/trace_$1_$2.log 2>&1
  echo "== slices $1 x $2 us" >> $O/trace.txt; grep -E "step_local|mini_world_sample" gpurun_out/r4an_$1_$2/p_kernel_stats.csv | sed 's/(StepArgs)//; s/(bcp::DevParams.*int)//' | cut -c1-170 >> $O/trace.txt
done
echo "== round's base library" >> $O/endless.txt
BCP_LIB=tools/libbcplan_base.so python tools/bench_endless.py 2>&1 | grep -E "side stream|high-priority|one refresh|status" >> $O/endless.txt
cat $O/rc.txt; tail -n 3 $O/soak.txt; cat $O/endless.txt $O/trace.txt
