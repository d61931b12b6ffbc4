"""In-kernel s_memtime stamps of step_pending_kernel (needs a -DBCP_DIAG build of the library at tools/libbcplan_diag.so:
hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -shared -std=c++17 -DBCP_DIAG -Iinclude
bc_gym_planning_env_amd/csrc/bcplan.hip -o tools/libbcplan_diag.so)."""
import sys, os, ctypes as C, numpy as np, torch
sys.path.insert(0,'.')
from bc_gym_planning_env_amd import _lib
_lib.LIB_PATH = os.path.join('tools', os.environ.get('BCP_DIAG_LIB','libbcplan_diag.so'))   # diagnostic build with in-kernel stamps
import bench
env,g = bench.make_env(65536,0,0,1)
rng=np.random.RandomState(0)
pool=torch.from_numpy(np.stack([env.action_space.sample_batch(65536,rng) for _ in range(16)])).cuda()
env.state.current_iter.copy_(torch.from_numpy(rng.randint(0,1200,65536).astype(np.int32)).cuda())
for k in range(1300): env.step(pool[k%16])
torch.cuda.synchronize()
L=_lib.load()
buf=(C.c_ulonglong*(4096*8))()
L.bcp_diag_read.argtypes=[C.c_void_p]
L.bcp_diag_read(buf)
a=np.array(buf[:]).reshape(4096,8).astype(np.int64)
a=a[a[:,0]>0]   # workgroups that ran
work=a[:,1]>a[:,0]
t0=a[:,0].min()
print('blocks with work', work.sum())
d=a[work]
print('start spread (cycles@100MHz?) min/max of stamp0 rel', (d[:,0]-t0).min(), (d[:,0]-t0).max())
for k in range(1,5):
    print('stamp%d - stamp%d: median %d  p90 %d  max %d'%(k,k-1, np.median(d[:,k]-d[:,k-1]), np.percentile(d[:,k]-d[:,k-1],90), (d[:,k]-d[:,k-1]).max()))
print('total stamp4-stamp0 median', np.median(d[:,4]-d[:,0]), 'max', (d[:,4]-d[:,0]).max(), ' last end rel t0', (d[:,4]-t0).max())
idle=a[~work]
print('idle blocks', len(idle))
