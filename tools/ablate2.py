import sys, numpy as np, torch
sys.path.insert(0,'.')
import bench
from bc_gym_planning_env_amd import NativeOps
env,g = bench.make_env(65536,0,0,1)
rng=np.random.RandomState(0)
pool=torch.from_numpy(np.stack([env.action_space.sample_batch(65536,rng) for _ in range(16)])).cuda()
env.state.current_iter.copy_(torch.from_numpy(rng.randint(0,1200,65536).astype(np.int32)).cuda())
for k in range(1200): env.step(pool[k%16])
torch.cuda.synchronize()
st=env.get_state()
rob=st.robot.cpu().numpy()
res=float(g['resolution']); org=g['origin']
px=np.rint((rob[0]-org[0])/res); py=np.rint((rob[1]-org[1])/res)
inmap=(px>=0)&(px<183)&(py>=0)&(py<183)
reach=47
near=(px+reach>=0)&(px-reach<183)&(py+reach>=0)&(py-reach<183)
print('center in map %.3f  bbox touches map %.3f' % (inmap.mean(), near.mean()))
print('iter hist', np.histogram(st.current_iter.cpu().numpy(), bins=6, range=(0,1200))[0])
for name,kw in [('auto',dict(exact_mode=0)),('coop',dict(exact_mode=1)),('dense',dict(exact_mode=2)),('auto_thr4',dict(exact_mode=0,dense_threshold=4)),('auto_thr32',dict(exact_mode=0,dense_threshold=32))]:
    env.set_state(st); env.set_tuning(**kw); env._debug_flags=0
    ms=[env.time_steps(pool[i%16],20) for i in range(3)]
    print(name, ['%.3f'%m for m in ms])
# collision op timing on the steady-state poses
ops=NativeOps(); 
poses=torch.stack([st.robot[0],st.robot[1],st.robot[2]],1).contiguous()
for cull in (1,0):
    ops.set_tuning(cull=cull, exact_mode=1); ops.set_costmap(g['costmap'],org,res)
    out=ops.pose_collides(poses); torch.cuda.synchronize()
    e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record(); 
    for _ in range(20): out=ops.pose_collides(poses)
    e1.record(); torch.cuda.synchronize()
    print('pose_collides cull=%d coop: %.3f ms, hits %d'%(cull, e0.elapsed_time(e1)/20, int(out.sum())))
