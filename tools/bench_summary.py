"""One line per leg of a bench.py JSON line: python tools/bench_summary.py file.json"""
import json, sys
d = json.loads([x for x in open(sys.argv[1]).read().splitlines() if x.startswith("{")][-1])
aux = d.pop("aux", {})
cb = d.pop("cpu_baseline", None)
print("value %.4e  ms_per_step %.5f  device %.5f  frac %.4f  parked %.4f" % (
    d["value"], d["ms_per_step"], d["timed_region"]["device_ms_per_step"], d["roofline"]["frac"], d.get("parked_pose_fraction", -1)))
if cb:
    print("cpu_baseline %.3e on %d cores" % (cb["value"], cb["cores"]))
for k, v in aux.items():
    if not isinstance(v, dict):
        print(k, v)
        continue
    keep = {kk: (round(vv, 5) if isinstance(vv, float) else vv) for kk, vv in v.items()
            if kk in ("ms_per_step", "ms_per_call", "kernel", "env_steps_per_s", "parked_pose_fraction", "step_plus_observation_ms")}
    r = v.get("roofline") or {}
    print(k, keep, "frac", r.get("frac"))
