#!/bin/bash
# kernel trace + WRITE_SIZE / FETCH_SIZE passes over the egocentric observation on the two AisleTurn maps
# (tools/ego_prof_aisle.py); writes gpurun_out/ego_<tag>_kernel_stats.csv and gpurun_out/ego_aisle_pmc.json
set -e
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for tag in colored aisle; do
  fx=g12_colored_ego.npz; if [ $tag = aisle ]; then fx=g10_ego_aisle.npz; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ego_${tag}_trace -o p -- python3 tools/ego_prof_aisle.py $fx > gpurun_out/ego_${tag}_trace.log 2>&1
  grep -E "Name|ego_|goal_" gpurun_out/ego_${tag}_trace/p_kernel_stats.csv > gpurun_out/ego_${tag}_kernel_stats.csv
  for c in WRITE_SIZE FETCH_SIZE; do
    rocprofv3 --kernel-trace --output-format csv --pmc $c -d gpurun_out/ego_${tag}_pmc_$c -o p -- python3 tools/ego_prof_aisle.py $fx > gpurun_out/ego_${tag}_pmc_$c.log 2>&1
  done
done
python3 - <<'PY'
import csv, glob, json
out = {"source": "rocprofv3 --kernel-trace --stats, then --pmc WRITE_SIZE and --pmc FETCH_SIZE (one run each) -- python3 tools/ego_prof_aisle.py <fixture>; 65 536 envs at steady state, 10 observation calls; WRITE_SIZE in KiB as is, FETCH_SIZE KiB x 2 (gfx950 correction of MI355X_MICROARCH.md, calibrated in tools/pmc_calib)"}
for tag, px in (("colored", 133 * 133), ("aisle", 133 * 117)):
    row = {}
    for r in csv.DictReader(open("gpurun_out/ego_%s_kernel_stats.csv" % tag)):
        if "ego_sparse" in r["Name"] or "ego_costmap" in r["Name"]:
            row = {"kernel": r["Name"].split("(")[0], "calls": int(r["Calls"]), "avg_us": float(r["AverageNs"]) / 1e3, "min_us": float(r["MinNs"]) / 1e3, "max_us": float(r["MaxNs"]) / 1e3}
    for c in ("WRITE_SIZE", "FETCH_SIZE"):
        vals = []
        for fn in glob.glob("gpurun_out/ego_%s_pmc_%s/**/*counter_collection.csv" % (tag, c), recursive=True):
            for r in csv.DictReader(open(fn)):
                if "ego_sparse" in r["Kernel_Name"] or "ego_costmap" in r["Kernel_Name"]:
                    vals.append(float(r["Counter_Value"]))
        row[c + "_KiB_per_launch"] = sum(vals) / max(len(vals), 1)
    alg = 65536 * px
    row["algorithmic_bytes_per_launch"] = alg
    row["hbm_write_bytes_per_launch"] = row["WRITE_SIZE_KiB_per_launch"] * 1024
    row["hbm_read_bytes_per_launch"] = row["FETCH_SIZE_KiB_per_launch"] * 1024 * 2
    row["written_over_algorithmic"] = row["hbm_write_bytes_per_launch"] / alg
    row["written_TB_per_s"] = alg / row["avg_us"] / 1e6
    row["fraction_of_8_TB_per_s"] = row["written_TB_per_s"] / 8.0
    out[tag] = row
json.dump(out, open("gpurun_out/ego_aisle_pmc.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
