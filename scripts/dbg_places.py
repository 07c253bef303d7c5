import sys, math, numpy as np, torch
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
from helpers import load_golden, t
from cnrma_amd import rma
g=load_golden('tiny'); dev=torch.device('cuda:0')
feats=rma.to_nhwc(t(g['features'],dev)); pinv=t(g['proj_inv'],dev); tsdf=t(g['tsdf'],dev)
rows,pv,samples=rma.rma_view_rows(feats,pinv,tsdf,g['dims'],g['voxel_size'],g['origin'],g['n_steps'],g['thr'],with_samples=True)
rows=rows.cpu().numpy(); s=samples.cpu().numpy()
n0=int(g['neus_counts'][0]); f32=np.float32
ray=s[:n0,0]; step=s[:n0,1].astype(np.int64)
o=g['ray_o'][0]; d=g['ray_d'][0][:,ray]
X,Y,Z=g['dims']; t_one=math.sqrt(X*X+Y*Y+Z*Z)*g['voxel_size']/g['n_steps']
tt=(step.astype(f32)*f32(t_one)).astype(f32)
pl=(o[:,None]+(d*tt).astype(f32)).astype(f32).T
def fma(a,b,c): return (a.astype(np.float64)*b.astype(np.float64)+c.astype(np.float64)).astype(np.float32)
pl2=fma(d,np.broadcast_to(tt,d.shape),np.broadcast_to(o[:,None],d.shape)).T
print('gpu vs unfused', (rows[:n0,:3]!=pl).sum(), 'gpu vs fused', (rows[:n0,:3]!=pl2).sum())
exp=g['neus_rows'][:n0]
print('gpu vs golden', (rows[:n0,:3]!=exp[:,:3]).sum(), 'w mismatch', (rows[:n0,3]!=exp[:,3]).sum())
i=np.nonzero((rows[:n0,:3]!=exp[:,:3]).any(1))[0][:5]
for k in i: print(k, ray[k], step[k], rows[k,:3], exp[k,:3], tt[k])
