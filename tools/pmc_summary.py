"""rocprofv3 FETCH_SIZE / WRITE_SIZE passes -> bytes per step.
python3 tools/pmc_summary.py <tag> <dir prefix of the workload passes> <dir prefix of the pmc_calib passes> <out.json>
(the passes live in <prefix>FETCH_SIZE and <prefix>WRITE_SIZE).  The counters (KiB) are converted with the factors
tools/pmc_calib measures for the step kernels' own access pattern (one 8 / 4 / 1 byte element per lane and array):
factor = bytes the calibration kernel really moves / counter value."""
import collections, csv, glob, json, re, sys

tag, work, calib, out_path = sys.argv[1:5]
n_envs = int(sys.argv[5]) if len(sys.argv) > 5 else 65536
algorithmic = int(sys.argv[6]) if len(sys.argv) > 6 else 163 * n_envs


def passes(prefix, keys):
    """{kernel: {counter: (launches, mean KiB per launch over the LAST half of the launches)}}"""
    res = collections.defaultdict(dict)
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        vals = collections.defaultdict(list)
        for fn in glob.glob(prefix + c + "/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(fn)):
                name = r["Kernel_Name"].split("(")[0].replace("void ", "")
                if any(k in name for k in keys):
                    vals[name].append(float(r["Counter_Value"]))
        for name, v in vals.items():
            tail = v[len(v) // 2:]   # (the first launches belong to the pre-roll)
            res[name][c] = (len(v), sum(tail) / max(len(tail), 1))
    return res


cal = passes(calib, ("calib_",))
log = " ".join(open(f).read() for f in glob.glob(calib + "FETCH_SIZE.log"))
m = re.search(r"\{\"lanes\".*\}", log)
truth = json.loads(m.group(0))["bytes"] if m else {}
factors = {}
for name, (rd, wr) in truth.items():
    got = cal.get(name, {})
    f = got.get("FETCH_SIZE", (0, 0.0))[1] * 1024.0
    w = got.get("WRITE_SIZE", (0, 0.0))[1] * 1024.0
    factors[name] = {"read_bytes": rd, "FETCH_SIZE_bytes": f, "read_factor": (rd / f) if f and rd else None,
                     "write_bytes": wr, "WRITE_SIZE_bytes": w, "write_factor": (wr / w) if w and wr else None}
rf = (factors.get("calib_step", {}).get("read_factor") or 2.0)
wf = (factors.get("calib_step", {}).get("write_factor") or 1.0)

raw = passes(work, ("step_", "copyBuffer", "robot_step_kernel"))
per, total = {}, 0.0
for name, v in raw.items():
    if name.startswith("step_"):   # the step kernels proper (robot_step_kernel / copyBuffer are calibration launches)
        rd = v.get("FETCH_SIZE", (0, 0.0))[1] * 1024.0 * rf
        wr = v.get("WRITE_SIZE", (0, 0.0))[1] * 1024.0 * wf
        per[name] = {"launches": v.get("FETCH_SIZE", (0, 0))[0], "read": rd, "write": wr, "total": rd + wr}
        total += rd + wr
per["total"] = total
per["algorithmic_bytes_per_step"] = algorithmic
per["ratio_to_algorithmic"] = total / algorithmic if algorithmic else None
out = {"workload": tag,
       "source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate runs), MI355X; counters converted "
                 "with the factors of tools/pmc_calib's calib_step kernel (the step's own access pattern)",
       "calibration": factors, "read_factor_used": rf, "write_factor_used": wf,
       "raw_KiB_per_launch": {k: {c: v[c][1] for c in v} for k, v in raw.items()},
       "corrected_bytes_per_step": per}
json.dump(out, open(out_path, "w"), indent=1)
print(json.dumps({"factors": {k: (v["read_factor"], v["write_factor"]) for k, v in factors.items()}, "per_step": per}, indent=1))
