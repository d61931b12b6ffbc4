import sys, time, numpy as np, torch
sys.path.insert(0, '.')
sys.path.insert(0, 'tools')
from diag_flags import time_steps_with_flags
from bc_gym_planning_env_amd import mini_env
n = 65536
pool = mini_env.sample_pool(None, list(range(1024)), 4)
env = mini_env.BatchedRandomMiniEnv(n, pool=pool, auto_reset=True, seed=3)
rng = np.random.RandomState(0)
acts = torch.from_numpy(np.stack([env.action_space.sample_batch(n, rng) for _ in range(8)])).cuda()
env.state.current_iter.copy_(torch.from_numpy(rng.randint(0, 1200, n).astype(np.int32)).cuda())
for k in range(1200):
    env.step(acts[k % 8])
torch.cuda.synchronize()
st = env.get_state()
print("ambiguous/park stats: collided_now mean %.5f" % float(env.collided_now.float().mean()))
for name, fl in [('full', 0), ('no_park', 1 << 21), ('no_classify', 1 << 22), ('no_collision', 1 << 16), ('no_reward', 1 << 17), ('neither', 3 << 16)]:
    env.set_state(st)
    ms = [time_steps_with_flags(env, acts[i % 8], 20, fl) for i in range(4)]
    print(name, ['%.4f' % m for m in ms], flush=True)
