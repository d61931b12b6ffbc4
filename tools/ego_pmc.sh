#!/bin/bash
# rocprofv3 passes over tools/ego_prof.py (65 536 observations of the C3 batch): kernel trace + stats, then PMC sets,
# each in its own run.  Summaries land in gpurun_out/ego_*; copy what is to be judged into profiles/.
set -e
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ego_trace -o p -- python3 tools/ego_prof.py > gpurun_out/ego_trace.log 2>&1
grep -E "Name|ego_|goal_n" gpurun_out/ego_trace/p_kernel_stats.csv > gpurun_out/ego_kernel_stats.csv
rm -f gpurun_out/ego_pmc_summary.txt
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "GRBM_GUI_ACTIVE SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --output-format csv --pmc $set -d gpurun_out/ego_pmc_$i -o p -- python3 tools/ego_prof.py > gpurun_out/ego_pmc_$i.log 2>&1
  python3 - <<PY
import csv, glob, collections
f = glob.glob("gpurun_out/ego_pmc_$i/**/*counter_collection.csv", recursive=True)
acc = collections.defaultdict(lambda: [0, 0.0])
for fn in f:
    for r in csv.DictReader(open(fn)):
        if "ego_costmap" in r["Kernel_Name"] or "ego_sparse" in r["Kernel_Name"]:
            a = acc[r["Counter_Name"]]
            a[0] += 1
            a[1] += float(r["Counter_Value"])
with open("gpurun_out/ego_pmc_summary.txt", "a") as o:
    for k, (c, v) in acc.items():
        o.write("%s per launch %.6g (launch records %d)\n" % (k, v / max(c, 1), c))
PY
done
python3 - <<'PY'
import csv, json, re
vals = {}
for line in open("gpurun_out/ego_pmc_summary.txt"):
    m = re.match(r"(\S+) per launch (\S+)", line)
    if m:
        vals[m.group(1)] = float(m.group(2))
avg_ns = None
for r in csv.DictReader(open("gpurun_out/ego_kernel_stats.csv")):
    if "ego_costmap" in r["Name"] or "ego_sparse" in r["Name"]:
        avg_ns, name = float(r["AverageNs"]), r["Name"]
n, px = 65536, 133 * 117
out = {"source": "rocprofv3 --kernel-trace --pmc <set> (one run per set) -- python3 tools/ego_prof.py; MI355X; kernel %s, %d images of 133 x 117 px per launch (C3 batch at steady state)" % (name, n),
       "per_launch": vals, "kernel_avg_us_from_kernel_trace": avg_ns / 1e3, "algorithmic_bytes_per_launch": n * px,
       "hbm_bytes_per_launch": {"write": vals.get("WRITE_SIZE", 0) * 1024, "read": vals.get("FETCH_SIZE", 0) * 1024 * 2,
                                "note": "WRITE_SIZE in KiB as is, FETCH_SIZE KiB x2 (gfx950 correction of MI355X_MICROARCH.md)"},
       "derived": {"written_TB_per_s": n * px / avg_ns / 1e3, "fraction_of_8_TB_per_s": n * px / avg_ns / 1e3 / 8.0,
                   "valu_instructions_per_image": vals.get("SQ_INSTS_VALU", 0) / n,
                   "lds_array_busy_fraction (SQ_LDS_IDX_ACTIVE / 256 CUs / kernel cycles at GRBM_GUI_ACTIVE / 8)":
                       vals.get("SQ_LDS_IDX_ACTIVE", 0) / 256 / max(vals.get("GRBM_GUI_ACTIVE", 1) / 8, 1),
                   "lds_conflict_fraction_of_lds_cycles": vals.get("SQ_LDS_BANK_CONFLICT", 0) / max(vals.get("SQ_LDS_IDX_ACTIVE", 1), 1),
                   "valu_busy_fraction (SQ_ACTIVE_INST_VALU quad-cycles x 4 / 1024 SIMDs / kernel cycles)":
                       vals.get("SQ_ACTIVE_INST_VALU", 0) * 4 / 1024 / max(vals.get("GRBM_GUI_ACTIVE", 1) / 8, 1)}}
json.dump(out, open("gpurun_out/ego_pmc.json", "w"), indent=1)
print(json.dumps(out["derived"], indent=1))
PY
cat gpurun_out/ego_kernel_stats.csv gpurun_out/ego_pmc_summary.txt
