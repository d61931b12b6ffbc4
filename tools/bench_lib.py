"""bench.py's metric leg with a given build of the library (A/B of kernel variants under the bench's own workload and timing):
python tools/bench_lib.py path/to/libbcplan_variant.so [bench.py arguments]   -- prints ms_per_step of the JSON line."""
import json, os, sys, io, contextlib
sys.path.insert(0, '.')
from bc_gym_planning_env_amd import _lib
lib = sys.argv[1]
if lib not in ('', '-'):
    _lib.LIB_PATH = os.path.abspath(lib)
import bench
sys.argv = ["bench.py", "--no-aux", "--no-cpu-baseline"] + sys.argv[2:]
buf = io.StringIO()
with contextlib.redirect_stdout(buf):
    bench.main()
line = json.loads([x for x in buf.getvalue().splitlines() if x.startswith("{")][-1])
print("%-28s ms_per_step %.5f (device %.5f)  value %.4e" % (os.path.basename(_lib.LIB_PATH), line["ms_per_step"],
      line["timed_region"]["device_ms_per_step"], line["value"]), flush=True)
