"""Endless geometry pool (BatchedRandomMiniEnv(endless=True)): step rate with refresh() every R steps, and what one
refresh costs.  Usage: python tools/bench_endless.py [n_envs] [episodes] [refresh_every] [steps] [side_cu_percent] [near_dilate]
(side_cu_percent: share of the compute units the overlapped refresh may use, bcp_side_stream; 100 = an ordinary stream;
near_dilate: BCP_TUNE_NEAR_DILATE, 0 = every refresh computes the uint8 distance fields, 1 = tiles by dilation only)"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from bc_gym_planning_env_amd import _lib, mini_env  # noqa: E402
if os.environ.get("BCP_LIB"):   # A/B of kernel variants: another build of the library
    _lib.LIB_PATH = os.path.abspath(os.environ["BCP_LIB"])

n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
episodes = int(sys.argv[2]) if len(sys.argv) > 2 else 4
every = int(sys.argv[3]) if len(sys.argv) > 3 else 128
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 2048
share = int(sys.argv[5]) if len(sys.argv) > 5 else 100
near_dilate = int(sys.argv[6]) if len(sys.argv) > 6 else 1

t0 = time.time()
env = mini_env.BatchedRandomMiniEnv(n, episodes=episodes, endless=True, auto_reset=True, seed=1)
env.side_cu_percent = share
env.set_tuning(near_dilate=near_dilate)
torch.cuda.synchronize()
print("setup: %d envs x %d entries in %.2f s; overlapped refresh on %d %% of the compute units; near_dilate %d" % (n, episodes, time.time() - t0, share, near_dilate), flush=True)
g = torch.Generator(device="cuda").manual_seed(0)
acts = [torch.rand(n, 2, device="cuda", generator=g, dtype=torch.float64) * torch.tensor([1.0, 1.0], device="cuda",
        dtype=torch.float64) - torch.tensor([0.0, 0.5], device="cuda", dtype=torch.float64) for _ in range(16)]
dones = torch.zeros(n, dtype=torch.int64, device="cuda")
for k in range(64):       # warm-up: spread the envs over their episodes
    env.step(acts[k % 16])
    if k % every == every - 1:
        env.refresh()
torch.cuda.synchronize()
hi = torch.cuda.Stream(priority=-1)
hi.wait_stream(torch.cuda.current_stream())
for label, do_refresh, overlap, prio in (("step only (envs run into their guards)", False, False, False),
                                         ("step + refresh in stream order", True, False, False),
                                         ("step + refresh on a side stream", True, True, False),
                                         ("same, steps on a high-priority stream", True, True, True)):
  with torch.cuda.stream(hi if prio else torch.cuda.current_stream()):
      new = waiting = 0
      e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
      infos = []
      dones.zero_()
      e0.record()
      for k in range(steps):
          env.step(acts[k % 16])
          dones += env.done
          if do_refresh and k % every == every - 1:
              info = env.refresh(overlap=overlap)
              if info is not None:
                  infos.append(info.clone())
      env.finish_refresh()
      e1.record()
      torch.cuda.synchronize()
      ms = e0.elapsed_time(e1)
      if infos:
          inf = torch.stack(infos).cpu().numpy()
          new, waiting = int(inf[:, 0].sum()), int(inf[:, 1].sum())
      print("%-42s %.4f ms/step  %.3g env-steps/s  episodes ended %d  refreshes %d  worlds re-sampled %d  "
            "waiting-at-guard sightings %d" % (label, ms / steps, n * steps / ms * 1e3, int(dones.sum()), len(infos), new,
                                               waiting), flush=True)
# one refresh in isolation, after `every` steps
ts = []
for rep in range(10):
    for k in range(every):
        env.step(acts[k % 16])
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    info = env.refresh()
    e1.record()
    torch.cuda.synchronize()
    ts.append((e0.elapsed_time(e1), int(info[0])))
ms = np.median([t for t, _ in ts])
cnt = np.median([c for _, c in ts])
print("one refresh after %d steps: %.3f ms for ~%d worlds (%.3g worlds/s)" % (every, ms, cnt, cnt / ms * 1e3))
print("status ok:", int(env._ring_status.sum()) == 0, int(env._ring_path_status.max()) == 0)
