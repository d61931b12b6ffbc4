"""Step time of the C3 geometry at small batch sizes (is the two-kernel split worth it below 1024 waves?)."""
import sys, numpy as np, torch
sys.path.insert(0, '.')
import bench
for n in (256, 1024, 4096, 16384, 32768):
    env, g = bench.make_env(n, 0, 0, 1)
    rng = np.random.RandomState(0)
    pool = torch.from_numpy(np.stack([env.action_space.sample_batch(n, rng) for _ in range(8)])).cuda()
    env.state.current_iter.copy_(torch.from_numpy(rng.randint(0, 1200, n).astype(np.int32)).cuda())
    for k in range(1200):
        env.step(pool[k % 8])
    torch.cuda.synchronize()
    ms = [env.time_steps(pool[i % 8], 50) for i in range(4)]
    print(n, ['%.4f' % m for m in ms], "env-steps/s %.3e" % (n / min(ms) * 1e3), flush=True)
