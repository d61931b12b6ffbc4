import sys, numpy as np, torch
sys.path.insert(0,'.')
import bench
env,g = bench.make_env(65536,0,0,1)
rng=np.random.RandomState(0)
pool=torch.from_numpy(np.stack([env.action_space.sample_batch(65536,rng) for _ in range(16)])).cuda()
env.state.current_iter.copy_(torch.from_numpy(rng.randint(0,1200,65536).astype(np.int32)).cuda())
env._debug_flags=0
for k in range(1200): env.step(pool[k%16])
torch.cuda.synchronize()
st=env.get_state()
for name,fl in [('no_park',1<<21),('no_classify',1<<22)]:
    env.set_state(st); env._debug_flags=fl
    ms=[env.time_steps(pool[i%16],10) for i in range(3)]
    print(name, ['%.4f'%m for m in ms], flush=True)
