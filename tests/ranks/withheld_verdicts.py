"""Child process of tests/test_gpu_errors.py::test_a_wait_that_never_ends_gives_up: steps a batch with a -DBCP_DIAG build of
the library (argv[1]) whose ticket loop tests the parked poses but never posts their verdicts (flag bit 23).  The movers'
bounded waits must give up: the step returns, the envs whose verdict never came are finished as free with BCP_ERR_INTERNAL,
bcp_expired_waits counts the waits, check_errors() raises.  Prints one JSON line."""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    from bc_gym_planning_env_amd import _lib
    _lib.LIB_PATH = os.path.abspath(sys.argv[1])
    import torch
    from ranks.sharded_rank import global_actions, make_shard
    n = 4096
    env = make_shard(n, 0, 0, timeout=1000)
    acts = global_actions(n, 40)
    for k in range(30):   # ordinary steps: robots reach the walls, poses get parked and settled
        env.step(torch.from_numpy(acts[k]).cuda())
    env.check_errors()
    before = C.c_int64()
    _lib.check(env._lib.bcp_expired_waits(env._h, C.byref(before), None))
    # steps with the verdicts withheld, until one of them has parked a pose (a few steps at most)
    after = C.c_int64()
    seconds, steps, flagged, flagged_hits = 0.0, 0, 0, 0
    acts = global_actions(n, 80)
    while after.value == 0 and steps < 40:
        a = torch.from_numpy(acts[30 + steps]).cuda()
        env._io.actions = a.data_ptr()
        env._io.noise_z = None
        env._io.noise_z_out = None
        env._io.done = env.done.data_ptr()
        torch.cuda.synchronize()
        t0 = time.time()
        rc = env._bcp_step(env._h, env._io_ref, env._flags_f32 | (1 << 23), torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        seconds = max(seconds, time.time() - t0)
        steps += 1
        _lib.check(env._lib.bcp_expired_waits(env._h, C.byref(after), None))
    err = env.err.cpu().numpy()
    raised = False
    try:
        env.check_errors()
    except RuntimeError:
        raised = True
    print(json.dumps({"rc": int(rc), "seconds": seconds, "withheld_steps": steps, "expired_before": before.value, "expired_after": after.value,
                      "envs_flagged": int((err & _lib.ERR_INTERNAL != 0).sum()),
                      "flagged_collided_now": int(env.collided_now.cpu().numpy()[err & _lib.ERR_INTERNAL != 0].sum()),
                      "check_errors_raised": raised}))


if __name__ == "__main__":
    main()
