"""SHA-256 of the state, rewards and flags after N steps of the metric workload with a given build of the library:
python tools/state_digest.py [path/to/libbcplan_variant.so] [steps] -- two builds whose arithmetic is the same print the same digest."""
import hashlib, sys
sys.path.insert(0, '.')
sys.path.insert(0, 'tools')
from _variant import use_lib
use_lib(sys.argv[1] if len(sys.argv) > 1 else '')
import numpy as np, torch
import bench
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 400
env, g = bench.make_env(16384, 0, 0, 2024)
rng = np.random.RandomState(7)
pool = torch.from_numpy(np.stack([env.action_space.sample_batch(16384, rng) for _ in range(16)])).cuda()
h = hashlib.sha256()
for k in range(steps):
    env.step(pool[k % 16])
    if k % 50 == 49:
        h.update(env.reward.cpu().numpy().tobytes())
        h.update(env.done.cpu().numpy().tobytes())
torch.cuda.synchronize()
h.update(env.state.robot.cpu().numpy().tobytes())
h.update(env.state.min_spat_dist_so_far.cpu().numpy().tobytes())
h.update(env.state.target_idx.cpu().numpy().tobytes())
print("digest after %d steps x 16384 envs: %s" % (steps, h.hexdigest()))
