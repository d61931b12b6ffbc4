#!/bin/bash
# which lanes' initial states are worth fetching ahead: with the parked lanes (v4 = the build) or without them (v5); time and FETCH_SIZE
O=gpurun_out/r4ak; mkdir -p $O; rm -f $O/*
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do
  for lib in tools/libbcplan_base.so tools/libbcplan_v4.so tools/libbcplan_v5.so; do python tools/step_time.py $lib 2>&1 | grep n=65536 >> $O/step_time.txt; done
done
for v in v4 v5; do
  BCP_LIB=tools/libbcplan_$v.so rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d gpurun_out/r4ak_$v -o p -- python3 tools/pmc_run.py > $O/pmc_$v.log 2>&1
  python3 - $v <<'PY' >> $O/fetch.txt
import csv, glob, sys
v = sys.argv[1]
vals = []
for fn in glob.glob("gpurun_out/r4ak_%s/**/*counter_collection.csv" % v, recursive=True):
    for r in csv.DictReader(open(fn)):
        if "step_local" in r["Kernel_Name"]:
            vals.append(float(r["Counter_Value"]))
print(v, "FETCH_SIZE KiB per launch (last 40):", sum(vals[-40:]) / 40, "launches", len(vals))
PY
done
for lib in tools/libbcplan_base.so tools/libbcplan_v4.so tools/libbcplan_v5.so; do python tools/step_time.py $lib 1048576 2>&1 | grep n=1048576 >> $O/step_time.txt; done
python tools/bench_rollout.py > $O/rollout.txt 2>&1
cut -c1-110 $O/step_time.txt; cat $O/fetch.txt; tail -n 6 $O/rollout.txt
