"""The C3 step with stages switched off (ablation flags of bcplan.hip: results are wrong by construction, only the
timing is of interest).  HIP-event timing of 20 back-to-back steps, 5 repetitions each."""
import sys, numpy as np, torch
sys.path.insert(0, '.')
import bench
env, g = bench.make_env(65536, 0, 0, 1)
rng = np.random.RandomState(0)
pool = torch.from_numpy(np.stack([env.action_space.sample_batch(65536, rng) for _ in range(16)])).cuda()
env.state.current_iter.copy_(torch.from_numpy(rng.randint(0, 1200, 65536).astype(np.int32)).cuda())
for k in range(1200):
    env.step(pool[k % 16])
torch.cuda.synchronize()
st = env.get_state()
for name, fl in [('full', 0), ('no_park', 1 << 21), ('no_classify', 1 << 22), ('no_collision', 1 << 16),
                 ('no_reward', 1 << 17), ('neither', 3 << 16), ('k2_no_coop', 1 << 19)]:
    env.set_state(st)
    env._debug_flags = fl
    ms = [env.time_steps(pool[i % 16], 20) for i in range(5)]
    print(name, ['%.4f' % m for m in ms], flush=True)
