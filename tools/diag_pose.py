import sys, os, numpy as np, torch
sys.path.insert(0, '.')
from bc_gym_planning_env_amd import NativeOps
g = np.load('tests/golden/g6_pose_collides.npz')
tag = 'mini0'
def say(*a):
    print(*a, flush=True)
ops = NativeOps('industrial_tricycle_v1')
mode = sys.argv[1] if len(sys.argv) > 1 else 'default'
if mode == 'nocull': ops.set_tuning(cull=0)
if mode == 'coop': ops.set_tuning(exact_mode=1)
if mode == 'dense': ops.set_tuning(exact_mode=2)
if mode == 'nocull_dense': ops.set_tuning(cull=0, exact_mode=2)
if mode == 'nocull_coop': ops.set_tuning(cull=0, exact_mode=1)
say('mode', mode)
ops.set_costmap(g[tag + '_map'], g[tag + '_origin'], float(g[tag + '_res']))
torch.cuda.synchronize(); say('set_costmap ok')
out = ops.pose_collides(g[tag + '_poses'][:64])
torch.cuda.synchronize(); say('pose_collides 64 ok', int(out.sum()), int(g[tag+'_collides'][:64].sum()))
out = ops.pose_collides(g[tag + '_poses'])
torch.cuda.synchronize(); say('pose_collides all ok', int((out.cpu().numpy() != g[tag+'_collides']).sum()))
