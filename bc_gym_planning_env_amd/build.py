"""Builds libbcplan.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "csrc", "bcplan.hip")
DEPS = [SRC] + [os.path.join(HERE, "csrc", f) for f in ("bcp_device.h", "bcp_raster.h", "bcp_coop.h", "bcp_step.h", "bcp_ego.h", "bcp_sample.h")] + \
       [os.path.join(os.path.dirname(HERE), "include", "bcplan.h")]
OUT = os.path.join(HERE, "libbcplan.so")

# -ffp-contract=off: numpy rounds every product and sum on its own; hipcc's default would fuse them into FMAs.
FLAGS = ["--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-fPIC", "-shared", "-std=c++17", "-Wall"]


def hipcc():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    return "hipcc"


def build(force=False, verbose=False):
    if not force and os.path.exists(OUT) and all(os.path.getmtime(OUT) >= os.path.getmtime(d) for d in DEPS):
        return OUT
    cmd = [hipcc()] + FLAGS + [SRC, "-o", OUT]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    print(build(force=True, verbose=True))
