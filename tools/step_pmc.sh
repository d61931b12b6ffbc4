#!/bin/bash
# rocprofv3 over the metric workload: (1) kernel trace + stats of `python3 bench.py`, (2) FETCH_SIZE and WRITE_SIZE
# passes (one run each) over tools/pmc_run.py.  Writes gpurun_out/step_kernel_stats.csv, gpurun_out/aux_kernel_stats.csv and gpurun_out/step_pmc.json;
# copy them to profiles/ (r<round>_bench_kernel_stats.csv, r<round>_pmc_summary.json).
set -e
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
# metric workload alone (the averages of the step kernels are the metric's), then the whole default run with its aux legs
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/step_trace -o p -- python3 bench.py --no-cpu-baseline --no-aux > gpurun_out/step_trace.log 2>&1
grep -E "^\"Name\"|step_|ego_|goal_n|mini_world|edt_|pack_bitmap|path_" gpurun_out/step_trace/p_kernel_stats.csv > gpurun_out/step_kernel_stats.csv
tail -1 gpurun_out/step_trace.log > gpurun_out/step_trace_bench_line.json || true
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/aux_trace -o p -- python3 bench.py --no-cpu-baseline > gpurun_out/aux_trace.log 2>&1
grep -E "^\"Name\"|step_|ego_|goal_n|mini_world|edt_|pack_bitmap|path_" gpurun_out/aux_trace/p_kernel_stats.csv > gpurun_out/aux_kernel_stats.csv
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --output-format csv --pmc $c -d gpurun_out/step_pmc_$c -o p -- python3 tools/pmc_run.py > gpurun_out/step_pmc_$c.log 2>&1
done
python3 - <<'PY'
import csv, glob, collections, json
raw = collections.defaultdict(dict)
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    acc = collections.defaultdict(lambda: [0, 0.0])
    for fn in glob.glob("gpurun_out/step_pmc_%s/**/*counter_collection.csv" % c, recursive=True):
        for r in csv.DictReader(open(fn)):
            name = r["Kernel_Name"]
            for key in ("step_fast", "step_pending_kernel", "step_kernel", "robot_step_kernel", "copyBuffer"):
                if key in name:
                    a = acc[name.split("(")[0].replace("void ", "")]
                    a[0] += 1
                    a[1] += float(r["Counter_Value"])
    for k, (n, v) in acc.items():
        # the first launches of a kernel belong to the pre-roll: average the LAST 40 step launches / last 3 calibration ones
        raw[k][c + "_KiB_per_launch_all_launches_mean"] = v / max(n, 1)
        raw[k]["launches"] = n
n = 65536
out = {"source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (two separate runs) -- python3 tools/pmc_run.py, MI355X",
       "raw": raw, "correction": "FETCH_SIZE x2 (gfx950, MI355X_MICROARCH.md HBM section; confirmed by the copyBuffer and robot_step_kernel calibration launches), WRITE_SIZE as is",
       "algorithmic_bytes_per_step": 163 * n}
tot = 0.0
per = {}
for k, v in raw.items():
    if "step_fast" in k or "step_pending" in k:
        rd = v.get("FETCH_SIZE_KiB_per_launch_all_launches_mean", 0) * 2 * 1024
        wr = v.get("WRITE_SIZE_KiB_per_launch_all_launches_mean", 0) * 1024
        per[k] = {"read": rd, "write": wr, "total": rd + wr}
        tot += rd + wr
per["total"] = tot
per["algorithmic_bytes_per_step"] = 163 * n
out["corrected_bytes_per_step"] = per
json.dump(out, open("gpurun_out/step_pmc.json", "w"), indent=1)
print(json.dumps(per, indent=1))
PY
cat gpurun_out/step_kernel_stats.csv
