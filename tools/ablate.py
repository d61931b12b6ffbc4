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
for name,fl in [('full',0),('no_raster',1<<16),('no_reward',1<<17),('neither',3<<16)]:
    env.set_state(st); env._debug_flags=fl
    ms=[env.time_steps(pool[i%16],20) for i in range(5)]
    print(name, ['%.3f'%m for m in ms])
