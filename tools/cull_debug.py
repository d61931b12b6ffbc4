import sys, numpy as np, torch
sys.path.insert(0,'.')
import bench
from scipy import ndimage
env,g = bench.make_env(65536,0,0,1)
rng=np.random.RandomState(0)
pool=torch.from_numpy(np.stack([env.action_space.sample_batch(65536,rng) for _ in range(16)])).cuda()
env.state.current_iter.copy_(torch.from_numpy(rng.randint(0,1200,65536).astype(np.int32)).cuda())
for k in range(1200): env.step(pool[k%16])
torch.cuda.synchronize()
rob=env.state.robot.cpu().numpy(); n=rob.shape[1]
res=float(g['resolution']); origin=g['origin']; leth=(g['costmap']==254)
import oracle as O
fp=O.TRICYCLE_FOOTPRINT
reach=int(np.ceil(np.hypot(fp[:,0],fp[:,1]).max()/res))+2; pad=2*reach+4
lp=np.pad(leth,pad); edt=np.floor(np.minimum(255,ndimage.distance_transform_edt(~lp))).astype(int)
x,y,th=rob[0],rob[1],rob[2]
px=np.rint((x-origin[0])/res).astype(int); py=np.rint((y-origin[1])/res).astype(int)
off=(px+reach<0)|(px-reach>=183)|(py+reach<0)|(py-reach>=183)
ay=0.5*(fp[:,1].min()+fp[:,1].max()); hw=0.5*(fp[:,1].max()-fp[:,1].min())
a0=fp[:,0].min()+hw; a1=fp[:,0].max()-hw
xx=np.clip(fp[:,0],a0,a1); rho=np.hypot(fp[:,0]-xx,fp[:,1]-ay).max()
nout=min(8,max(2,int(np.ceil((a1-a0)/(0.5*rho)))+1)); h=(a1-a0)/(nout-1)
rout=np.sqrt(rho**2+h*h/4)/res+1.9244; tout=int(np.floor(rout))+1
print('reach',reach,'pad',pad,'rho',rho,'a0',a0,'a1',a1,'nout',nout,'tout',tout)
c,s=np.cos(th),np.sin(th)
allfar=np.ones(n,bool); mind=np.full(n,999)
for i in range(nout):
    ox=(a0+i*h)/res; oy=ay/res
    du=np.rint(ox*c-oy*s).astype(int); dv=np.rint(ox*s+oy*c).astype(int)
    cx=np.clip(px+pad+du,0,lp.shape[1]-1); cy=np.clip(py+pad+dv,0,lp.shape[0]-1)
    d=edt[cy,cx]; allfar&=d>=tout; mind=np.minimum(mind,d)
free=off|allfar
print('off-map %.3f far %.3f not-free %.4f'%(off.mean(),(allfar&~off).mean(),1-free.mean()))
print('hist of min edt among in-range:', np.histogram(mind[~off],bins=[0,5,10,15,20,25,30,40,60,100,256])[0])
