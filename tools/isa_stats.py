#!/usr/bin/env python
"""Per-kernel register / spill / scratch figures and instruction mix of libbcplan, from hipcc -save-temps.

    python tools/isa_stats.py [name-filter] [--mix]

Compiles csrc/bcplan.hip into a scratch directory with the flags of bc_gym_planning_env_amd/build.py plus -save-temps
and reads the .amdhsa metadata of the .s file (what DESIGN.md quotes for VGPRs, spills and scratch)."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def compile_s(extra=()):
    from bc_gym_planning_env_amd import build
    d = tempfile.mkdtemp(prefix="bcp_isa_")
    cmd = [build.hipcc()] + build.FLAGS + list(extra) + ["-save-temps", build.SRC, "-o", os.path.join(d, "lib.so")]
    subprocess.check_call(cmd, cwd=d)
    return os.path.join(d, "bcplan-hip-amdgcn-amd-amdhsa-gfx950.s")


def kernels(path):
    text = open(path).read()
    meta = {}
    for m in re.finditer(r"  - \.agpr_count:.*?\.wavefront_size:\s+\d+", text, re.S):
        blk = m.group(0)
        name = re.search(r"\.name:\s+(\S+)", blk).group(1)
        meta[name] = dict((k, int(v)) for k, v in re.findall(r"\.(\w+):\s+(\d+)\s*$", blk, re.M))
    bodies = {}
    for name in meta:
        m = re.search(r"^%s:\n(.*?)^\s*\.section|^%s:\n(.*?)\.Lfunc_end" % (re.escape(name), re.escape(name)), text, re.S | re.M)
        if m:
            bodies[name] = m.group(1) or m.group(2)
    return meta, bodies


def mix(body):
    ops = re.findall(r"^\s+([a-z_0-9]+)\s", body, re.M)
    c = {}
    for o in ops:
        key = ("v_" if o.startswith("v_") else "s_" if o.startswith("s_") else "ds_" if o.startswith("ds_") else
               "flat_" if o.startswith("flat_") else "global_" if o.startswith("global_") else
               "scratch_" if o.startswith("scratch_") else "buffer_" if o.startswith("buffer_") else "other")
        c[key] = c.get(key, 0) + 1
    for o in ("v_writelane_b32", "v_readlane_b32", "s_cbranch_execz", "s_and_saveexec_b64", "s_mov_b32", "s_mov_b64"):
        c[o] = ops.count(o)
    c["total"] = len(ops)
    return c


def main():
    flt = [a for a in sys.argv[1:] if not a.startswith("--")]
    meta, bodies = kernels(compile_s())
    for name in sorted(meta):
        if flt and not any(f in name for f in flt):
            continue
        k = meta[name]
        print("%-58s vgpr %3d sgpr %3d  spills v %3d s %3d  scratch %4d B  lds %6d B"
              % (name[:58], k.get("vgpr_count", -1), k.get("sgpr_count", -1), k.get("vgpr_spill_count", 0),
                 k.get("sgpr_spill_count", 0), k.get("private_segment_fixed_size", 0), k.get("group_segment_fixed_size", 0)))
        if "--mix" in sys.argv and name in bodies:
            print("    ", mix(bodies[name]))


if __name__ == "__main__":
    main()
