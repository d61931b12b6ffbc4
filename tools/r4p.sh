#!/bin/bash
O=gpurun_out/r4p; mkdir -p $O; rm -f $O/*
python -m pytest tests -m gpu -x -q --timeout 300 > $O/gpu_tests.log 2>&1; echo "tests rc=$?" > $O/rc.txt
if grep -q "Memory access fault" $O/gpu_tests.log; then echo FAULT; tail -n 20 $O/gpu_tests.log; exit 1; fi
python bench.py > $O/bench_default.json 2> $O/bench.err; echo "bench rc=$?" >> $O/rc.txt
cat $O/rc.txt; tail -n 3 $O/gpu_tests.log; python - <<'PY'
import json
d=json.load(open('gpurun_out/r4p/bench_default.json'))
aux=d.pop('aux'); d.pop('cpu_baseline')
print('value %.4e ms %.5f' % (d['value'], d['ms_per_step']))
for k,v in aux.items():
    print(k, {kk: (round(vv,5) if isinstance(vv,float) else vv) for kk,vv in v.items() if kk in ('ms_per_step','ms_per_call','kernel','setup_detail','setup_s')})
PY
