#!/bin/bash
# Where the step kernel's time goes: SQ counter sets (one rocprofv3 run each; 8 SQ slots per pass) over the metric workload
# (tools/pmc_run.py) or, with `c4` as the argument, over BASELINE configs[3] (tools/pmc_run_c4.py).
# Writes gpurun_out/step_alu_pmc[_c4].json; copy to profiles/r<round>_step_alu_pmc[_c4].json.
set -e
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
TAG=""; RUN=tools/pmc_run.py
if [ "$1" = "c4" ]; then TAG="_c4"; RUN=tools/pmc_run_c4.py; fi
# BCP_LOCAL_PAIRS=1|2|4 in the environment: the same counters for another workgroup size of step_local_kernel (bcp_create reads it)
if [ -n "$BCP_LOCAL_PAIRS" ]; then TAG="${TAG}_p${BCP_LOCAL_PAIRS}"; fi
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE" \
           "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQC_TC_INST_REQ SQ_IFETCH SQ_IFETCH_LEVEL" \
           "SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_DCACHE_MISSES_DUPLICATE SQC_TC_DATA_READ_REQ SQ_INST_LEVEL_SMEM SQ_INST_LEVEL_VMEM"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --output-format csv --pmc $set -d gpurun_out/step_alu${TAG}_$i -o p -- python3 $RUN > gpurun_out/step_alu${TAG}_$i.log 2>&1 || echo "set $i failed: $set"
done
python3 - "$TAG" "$RUN" <<'PY'
import csv, glob, collections, json, sys
tag, run = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for fn in glob.glob("gpurun_out/step_alu%s_[0-9]/**/*counter_collection.csv" % tag, recursive=True):
    for r in csv.DictReader(open(fn)):
        for key in ("step_local_kernel", "step_fast_pair_kernel", "step_pending_kernel"):
            if key in r["Kernel_Name"]:
                acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {"source": "rocprofv3 --kernel-trace --pmc <set> (one run per set) -- python3 %s; 65536 envs, steady state, averages "
                 "over the last 40 launches of each kernel.  SQ_*_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles "
                 "summed over waves (MI355X_MICROARCH.md)" % run, "per_launch": {}}
for k, d in acc.items():
    out["per_launch"][k] = {c: sum(v[-40:]) / len(v[-40:]) for c, v in d.items()}
for k, d in out["per_launch"].items():
    der = {}
    wc = d.get("SQ_WAVE_CYCLES")
    if wc:
        for name, c in (("waiting", "SQ_WAIT_ANY"), ("issue_stalled", "SQ_WAIT_INST_ANY"), ("issuing", "SQ_ACTIVE_INST_ANY"),
                        ("valu_active", "SQ_ACTIVE_INST_VALU"), ("scalar_active", "SQ_ACTIVE_INST_SCA"),
                        ("lds_active", "SQ_ACTIVE_INST_LDS"), ("lds_issue_stalled", "SQ_WAIT_INST_LDS"),
                        ("vmem_cycles", "SQ_INST_CYCLES_VMEM")):
            if c in d:
                der[name + "_fraction_of_wave_cycles"] = d[c] / wc
    if d.get("SQ_WAVES"):
        w = d["SQ_WAVES"]
        der["waves_per_launch"] = w
        for name, c in (("valu", "SQ_INSTS_VALU"), ("salu", "SQ_INSTS_SALU"), ("lds", "SQ_INSTS_LDS"), ("smem", "SQ_INSTS_SMEM"),
                        ("vmem_rd", "SQ_INSTS_VMEM_RD"), ("vmem_wr", "SQ_INSTS_VMEM_WR")):
            if c in d:
                der[name + "_instructions_per_wave"] = d[c] / w
        if wc:
            der["wave_quad_cycles_per_wave"] = wc / w
    if d.get("SQC_ICACHE_REQ"):
        der["icache_miss_fraction_of_requests"] = (d.get("SQC_ICACHE_MISSES", 0) + d.get("SQC_ICACHE_MISSES_DUPLICATE", 0)) / d["SQC_ICACHE_REQ"]
        der["icache_requests_per_wave"] = d["SQC_ICACHE_REQ"] / max(d.get("SQ_WAVES", 4096.0), 1.0)
    if d.get("SQ_IFETCH"):
        der["ifetch_mean_latency_quad_cycles"] = d.get("SQ_IFETCH_LEVEL", 0) / d["SQ_IFETCH"]
    if d.get("SQC_DCACHE_REQ"):
        der["scalar_cache_miss_fraction_of_requests"] = (d.get("SQC_DCACHE_MISSES", 0) + d.get("SQC_DCACHE_MISSES_DUPLICATE", 0)) / d["SQC_DCACHE_REQ"]
    if d.get("SQ_LDS_IDX_ACTIVE"):
        der["lds_conflict_fraction_of_lds_cycles"] = d.get("SQ_LDS_BANK_CONFLICT", 0) / d["SQ_LDS_IDX_ACTIVE"]
    out.setdefault("derived", {})[k] = der
json.dump(out, open("gpurun_out/step_alu_pmc%s.json" % tag, "w"), indent=1)
print(json.dumps(out, indent=1))
PY
