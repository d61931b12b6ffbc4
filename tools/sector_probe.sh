#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d gpurun_out/sector_probe -o p -- tools/sector_probe > gpurun_out/sector_probe.log 2>&1
python3 - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(list)
for fn in glob.glob("gpurun_out/sector_probe/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        if "probe" in r["Kernel_Name"]:
            acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc):
    v = acc[k]
    print("%-40s FETCH_SIZE %s KiB -> %.1f bytes per thread (counter x 1024 / 2 M threads; x 2 with the stream calibration)" % (k, ["%.0f" % x for x in v], v[-1] * 1024 / (1 << 21)))
PY
