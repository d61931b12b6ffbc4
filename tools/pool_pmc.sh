#!/bin/bash
# rocprofv3 over the geometry-pool step (one private world per env): kernel trace, then FETCH_SIZE and WRITE_SIZE passes.
set -e
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/pool_trace -o p -- python3 tools/pmc_pool_run.py > gpurun_out/pool_trace.log 2>&1
grep -E "^\"Name\"|step_" gpurun_out/pool_trace/p_kernel_stats.csv > gpurun_out/pool_kernel_stats.csv
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --output-format csv --pmc $c -d gpurun_out/pool_pmc_$c -o p -- python3 tools/pmc_pool_run.py > gpurun_out/pool_pmc_$c.log 2>&1
done
python3 - <<'PY'
import csv, glob, collections, json
out = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    acc = collections.defaultdict(list)
    for fn in glob.glob("gpurun_out/pool_pmc_%s/**/*counter_collection.csv" % c, recursive=True):
        for r in csv.DictReader(open(fn)):
            if "step_" in r["Kernel_Name"]:
                acc[r["Kernel_Name"].split("(")[0].replace("void ", "")].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        last = v[-40:]
        out.setdefault(k, {})[c + "_KiB_per_launch_last40"] = sum(last) / len(last)
tot = 0
for k, v in out.items():
    v["bytes_per_launch_corrected"] = v.get("FETCH_SIZE_KiB_per_launch_last40", 0) * 2 * 1024 + v.get("WRITE_SIZE_KiB_per_launch_last40", 0) * 1024
    tot += v["bytes_per_launch_corrected"]
out["total_bytes_per_step"] = tot
out["bytes_per_env_step"] = tot / 65536
json.dump(out, open("gpurun_out/pool_pmc.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
cat gpurun_out/pool_kernel_stats.csv
