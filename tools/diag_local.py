"""In-kernel s_memtime stamps of step_local_kernel on the metric workload (`c4` as an argument: on BASELINE configs[3])
(needs the -DBCP_DIAG build at tools/libbcplan_diag.so, see tools/ablate.py): where a workgroup's time goes, in shader cycles."""
import sys, os, ctypes as C, numpy as np, torch
sys.path.insert(0, '.')
from bc_gym_planning_env_amd import _lib
_lib.LIB_PATH = os.path.join('tools', 'libbcplan_diag.so')
import bench
n = 65536
pairs = int(os.environ.get("BCP_LOCAL_PAIRS", "4"))   # workgroup size of step_local_kernel (bcp_create reads the same variable)
groups = n // (64 * pairs)
rng = np.random.RandomState(0)
if "c4" in sys.argv[1:]:   # BASELINE configs[3]: private AisleTurn costmaps and paths
    env = bench.make_c4_env(n, 0)
    pool = torch.from_numpy(np.stack([env.action_space.sample_batch(n, rng) for _ in range(16)])).cuda()
    bench.steady_state(env, pool, rng)
else:
    env, g = bench.make_env(n, 0, 0, 1)
    pool = torch.from_numpy(np.stack([env.action_space.sample_batch(n, rng) for _ in range(16)])).cuda()
    bench.steady_state(env, pool, rng)
for k in range(50):
    env.step(pool[k % 16])
torch.cuda.synchronize()
L = _lib.load()
L.bcp_diag_clear()          # (the maxima accumulate: look at ONE step)
env.step(pool[3])
torch.cuda.synchronize()
buf = (C.c_ulonglong * (4096 * 16))()
L.bcp_diag_read.argtypes = [C.c_void_p]
L.bcp_diag_read(buf)
raw_u = np.frombuffer(buf, dtype=np.uint64).reshape(4096, 16).copy()   # (the wall-clock rows hold complements: keep them unsigned)
raw = raw_u.astype(np.int64)
G = min(groups, 512)                # (stamps of the first 512 workgroups: the per-wave tables below start at row 512)
a = raw[:G]
upper = raw[2048:2048 + G]   # maxima over the waves of a workgroup (DIAG_MAX)
t0 = a[:, 0].min()
names = {1: "mover: loads issued .. robot model starts", 2: "mover: robot model done", 3: "mover: past barrier 1",
         4: "mover: classified and parked", 5: "mover: past barrier 2", 6: "mover: reward provider done", 7: "mover: decided envs finished",
         13: "mover: out of tickets", 8: "scorer: scanned", 9: "helper: past barrier 2", 14: "helper: out of tickets"}
print("workgroups %d; start skew (stamp 0 - earliest): median %d max %d cycles" % (len(a), np.median(a[:, 0] - t0), (a[:, 0] - t0).max()))
for k in (1, 2, 3, 8, 4, 5, 6, 9, 7, 13, 14):
    d = a[:, k] - a[:, 0]
    print("  %-45s since kernel entry: median %6d  p90 %6d  max %6d" % (names[k], np.median(d), np.percentile(d, 90), d.max()))
had = a[:, 10] > a[:, 0]
print("helper wave 8 had a ticket in %d of %d workgroups; parked poses per workgroup: mean %.2f max %d" % (had.sum(), len(a), a[:, 15].mean(), a[:, 15].max()))
h = a[had] if had.any() else a[:1]
print("  helper: ticket -> verdict  median %d  p90 %d ;  verdict -> verdict posted  median %d  p90 %d" % (
    np.median(h[:, 11] - h[:, 10]), np.percentile(h[:, 11] - h[:, 10], 90), np.median(h[:, 12] - h[:, 11]), np.percentile(h[:, 12] - h[:, 11], 90)))
unames = {1: "scorer: noise drawn", 2: "helper: cos / sin of the old heading", 3: "mover: state loads landed (diag wait)",
          4: "mover: cos / sin of the new heading", 5: "mover: classified", 6: "scorer: candidate window known",
          7: "helper 1: scanned", 8: "helper 2: scanned", 10: "mover: prologue loads issued", 11: "scorer: prologue loads issued",
          12: "mover: first half of the robot model"}
for k in sorted(unames):
    d = upper[:, k] - a[:, 0]
    print("  %-45s since kernel entry: median %6d  p90 %6d  max %6d" % (unames[k], np.median(d), np.percentile(d, 90), d.max()))
print("  longest candidate window among the lanes 0 of a workgroup's waves: median %d  p90 %d  max %d way points" % (
    np.median(upper[:, 9]), np.percentile(upper[:, 9], 90), upper[:, 9].max()))
W = min(groups, 256)                # (per-wave stamps: the first 256 workgroups)
first = raw[512:512 + W, :4 * pairs].min(axis=1, keepdims=True)   # the first instruction of the workgroup's earliest wave
print("  stamp 0 (wave 0, arguments fetched) since the workgroup's first instruction: median %d" % np.median(a[:W, 0] - first[:, 0]))
for base, what in ((512, "first instruction"), (768, "launch arguments fetched"), (1280, "prologue loads issued"), (1024, "arrival at barrier 0"), (1536, "arrival at barrier 1")):
    w = raw[base:base + W, :4 * pairs] - first
    print("  %-26s by wave (median cycles since the workgroup's first instruction): %s" % (what, " ".join("%5d" % v for v in np.median(w, axis=0))))
# wall clock (s_memrealtime, 100 MHz, one counter chip-wide): when does a workgroup enter and leave, against the first entry of the launch
real = raw_u[3072:3072 + min(groups, 1024)]
entry = (~real[:, 0]).astype(np.int64)
leave = real[:, 1].astype(np.int64)
ok = real[:, 1] > 0
t_first = entry[ok].min()
print("wall clock, microseconds since the first workgroup's entry (%d workgroups):" % ok.sum())
print("   entry: median %.2f  p90 %.2f  max %.2f   exit: median %.2f  p90 %.2f  max %.2f   life: median %.2f p90 %.2f max %.2f" % (
    np.median(entry[ok] - t_first) / 100.0, np.percentile(entry[ok] - t_first, 90) / 100.0, (entry[ok] - t_first).max() / 100.0,
    np.median(leave[ok] - t_first) / 100.0, np.percentile(leave[ok] - t_first, 90) / 100.0, (leave[ok] - t_first).max() / 100.0,
    np.median((leave - entry)[ok]) / 100.0, np.percentile((leave - entry)[ok], 90) / 100.0, (leave - entry)[ok].max() / 100.0))
for k in range(0, len(entry), 256):   # (round-4 probe: workgroups b, b + 256, ... share a compute unit)
    sel = ok[k:k + 256]
    e, l = entry[k:k + 256][sel] - t_first, leave[k:k + 256][sel] - t_first
    print("   workgroups %4d .. %4d: entry median %.2f max %.2f, exit median %.2f max %.2f" % (k, k + 255, np.median(e) / 100.0, e.max() / 100.0, np.median(l) / 100.0, l.max() / 100.0))
end = np.maximum(a[:, 13], a[:, 14]) - t0
print("workgroup end since the earliest start: median %d  p90 %d  max %d cycles" % (np.median(end), np.percentile(end, 90), end.max()))

# the kernel ends with its slowest workgroup: end time against the number of parked poses
endw = np.maximum(a[:, 13], a[:, 14]) - a[:, 0]
for k in sorted(set(a[:, 15].tolist())):
    sel = a[:, 15] == k
    print("  parked %2d: %3d workgroups, done after median %6d  max %6d cycles" % (k, sel.sum(), np.median(endw[sel]), endw[sel].max()))
# the longest exact test of each workgroup: cycles and kind (0 free, 1 hit, 2 too many cells -> row-by-row rasteriser)
dur, kind = upper[:, 0] >> 4, upper[:, 0] & 15
for k in (0, 1, 2):
    sel = (kind == k) & (a[:, 15] > 0)
    if sel.any():
        print("  longest test of a workgroup, kind %d: %3d workgroups, median %6d  max %6d cycles; those workgroups end after median %6d max %6d" % (
            k, sel.sum(), np.median(dur[sel]), dur[sel].max(), np.median(endw[sel]), endw[sel].max()))
# wave 8's (last) exact test by phase: edge set-up | listing the lethal cells under the image | filter + cell tests
ph = upper[had, 13]
if had.any():
    setup, listing, rest, cells = ph >> 40, (ph >> 20) & 0xFFFFF, ph & 0xFFFFF, upper[had, 15]
    print("  exact test of helper wave 8 by phase (cycles): edge set-up median %d p90 %d | cell list median %d p90 %d | filter + tests median %d p90 %d ; cells listed median %d p90 %d max %d" % (
        np.median(setup), np.percentile(setup, 90), np.median(listing), np.percentile(listing, 90), np.median(rest), np.percentile(rest, 90),
        np.median(cells), np.percentile(cells, 90), cells.max()))
# the ten workgroups that end last: where did they lose the time?
order = np.argsort(-endw)[:10]
print("  latest workgroups: parked | robot model done, barrier 1, parked, scans in, reward done, decided done, out | longest test (kind)")
for b in order:
    d = a[b] - a[b, 0]
    print("   wg %3d: %2d | %6d %6d %6d %6d %6d %6d %6d | %6d (%d)" % (b, a[b, 15], d[2], d[3], d[4], d[5], d[6], d[7], d[13], dur[b], kind[b]))
