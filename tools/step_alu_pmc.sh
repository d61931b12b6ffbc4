#!/bin/bash
# Where the step kernels' time goes: SQ counter sets (one rocprofv3 run each) over the metric workload (tools/pmc_run.py).
# Writes gpurun_out/step_alu_pmc.json; copy to profiles/r<round>_step_alu_pmc.json.
set -e
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA" "SQ_BUSY_CYCLES SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --output-format csv --pmc $set -d gpurun_out/step_alu_$i -o p -- python3 tools/pmc_run.py > gpurun_out/step_alu_$i.log 2>&1 || echo "set $i failed: $set"
done
python3 - <<'PY'
import csv, glob, collections, json
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for fn in glob.glob("gpurun_out/step_alu_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        for key in ("step_fast_pair_kernel", "step_pending_kernel"):
            if key in r["Kernel_Name"]:
                acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {"source": "rocprofv3 --kernel-trace --pmc <set> (one run per set) -- python3 tools/pmc_run.py; metric workload C3, "
                 "65536 envs, averages over the last 40 launches of each kernel", "per_launch": {}}
for k, d in acc.items():
    out["per_launch"][k] = {c: sum(v[-40:]) / len(v[-40:]) for c, v in d.items()}
for k, d in out["per_launch"].items():
    der = {}
    if d.get("SQ_WAVE_CYCLES"):
        der["valu_active_fraction_of_wave_cycles"] = d.get("SQ_ACTIVE_INST_VALU", 0) / d["SQ_WAVE_CYCLES"]
        der["waiting_fraction_of_wave_cycles"] = d.get("SQ_WAIT_ANY", 0) / d["SQ_WAVE_CYCLES"]
        der["issuing_fraction_of_wave_cycles"] = d.get("SQ_ACTIVE_INST_ANY", 0) / d["SQ_WAVE_CYCLES"]
    if d.get("SQ_WAVES"):
        der["valu_instructions_per_wave"] = d.get("SQ_INSTS_VALU", 0) / d["SQ_WAVES"]
        der["waves_per_launch"] = d["SQ_WAVES"]
    if d.get("SQ_LDS_IDX_ACTIVE"):
        der["lds_conflict_fraction_of_lds_cycles"] = d.get("SQ_LDS_BANK_CONFLICT", 0) / d["SQ_LDS_IDX_ACTIVE"]
    out.setdefault("derived", {})[k] = der
json.dump(out, open("gpurun_out/step_alu_pmc.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
