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
for name,fl in [('full',0),('k2_no_coop',1<<19),('k2_no_coop_no_inner',(1<<19)|(1<<20))]:
    env.set_state(st); env._debug_flags=fl
    for rep in range(2):
        k1,k2=env.time_step_kernels(pool[rep],100)
        print(name,'k1 %.4f  k2+gap %.4f'%(k1,k2), flush=True)
