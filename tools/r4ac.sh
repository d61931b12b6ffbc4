#!/bin/bash
# timeline of a workgroup with the counter among the arguments / on the device; step time of the variants
O=gpurun_out/r4ac; mkdir -p $O; rm -f $O/*
python tools/diag_local.py > $O/diag_known.txt 2>&1
BCP_TICK_ON_DEVICE=1 python tools/diag_local.py > $O/diag_device.txt 2>&1
for rep in 1 2; do
  for lib in tools/libbcplan_base.so tools/libbcplan_v1.so tools/libbcplan_v2.so; do python tools/step_time.py $lib 2>&1 | grep n=65536 >> $O/step_time.txt; done
done
python -m pytest tests/test_gpu_parity.py tests/test_gpu_noise.py -m gpu -x -q --timeout 300 > $O/tests.log 2>&1; echo "tests rc=$?" > $O/rc.txt
cat $O/rc.txt; tail -n 1 $O/tests.log; cat $O/step_time.txt; head -50 $O/diag_known.txt
