"""Debug aid for the single-launch step: a few steps at increasing batch sizes with a sync + wall time each, the
watchdog count and the queue tallies after every step.  python tools/debug_fused.py [n ...]"""
import ctypes as C, os, sys, time
sys.path.insert(0, '.')
import numpy as np, torch
import bench
from bc_gym_planning_env_amd import _lib

def queues(env):
    out = (C.c_int32 * (2 * 112))()
    _lib.check(env._lib.bcp_step_queues(env._h, out))
    a = np.array(out[:]).reshape(2, 112)
    return a

for n in [int(x) for x in sys.argv[1:]] or [1024, 8192, 65536]:
    env, g = bench.make_env(n, 0, 0, 2024)
    rng = np.random.RandomState(1)
    # scatter the robots along the path so that walls are near from the first step
    path = g["path"]
    k = rng.randint(0, len(path), n)
    st = np.zeros((7, n)); st[0] = path[k, 0] + rng.normal(0, .3, n); st[1] = path[k, 1] + rng.normal(0, .3, n); st[2] = rng.uniform(-3, 3, n)
    env.state.robot.copy_(torch.from_numpy(st))
    pool = torch.from_numpy(np.stack([env.action_space.sample_batch(n, rng) for _ in range(4)])).cuda()
    print("n", n, "form", env.step_kernels(), flush=True)
    for t in range(6):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        env.step(pool[t % 4])
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        ev = C.c_uint64(); _lib.check(env._lib.bcp_step_health(env._h, C.byref(ev)))
        q = queues(env)[t & 1]
        tally = q[16:].reshape(32, 3)
        if ev.value:
            print("  ", env._lib.bcp_last_error())
        print("  step %d: %.3f ms  watchdog %d  reserve %s head %s  tally min %s max %s" % (
            t, dt * 1e3, ev.value, q[:8].tolist(), q[8:16].tolist(), tally.min(0).tolist(), tally.max(0).tolist()), flush=True)
        if ev.value or dt > 0.5:
            print("  -> unhealthy, stopping"); sys.exit(1)
    env.close()
