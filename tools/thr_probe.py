import sys, numpy as np, torch
sys.path.insert(0,'.')
import bench
env,g = bench.make_env(65536,0,0,1)
rng=np.random.RandomState(0)
pool=torch.from_numpy(np.stack([env.action_space.sample_batch(65536,rng) for _ in range(16)])).cuda()
env.state.current_iter.copy_(torch.from_numpy(rng.randint(0,1200,65536).astype(np.int32)).cuda())
for k in range(1200): env.step(pool[k%16])
torch.cuda.synchronize()
st=env.get_state()
for thr in (64, 16, 6, 2, 0):
    env.set_state(st); env.set_tuning(dense_threshold=thr)
    ms=[env.time_steps(pool[i%16],50) for i in range(3)]
    print('dense_threshold',thr, ['%.4f'%m for m in ms], flush=True)
