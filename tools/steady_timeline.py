import sys, os, time; sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import torch, bench
dev=torch.device('cuda',0); torch.cuda.set_device(0)
from nmf_amd import synthetic, hip
from nmf_amd.noise import DeviceNoise
from nmf_amd.trainer import Trainer
nerf, params = bench.build(dev)
tr=Trainer(nerf, params); noise=DeviceNoise(dev, 1)
rays, focal = synthetic.camera_rays(4096, seed=1); rays=rays.to(dev); gt=torch.rand(4096,3,device=dev)
for i in range(5): tr.step(rays, gt, focal, noise=noise, update_controllers=False, fixed_chunk=4096)
torch.cuda.synchronize()
ev=[]
orig_cpu=torch.Tensor.cpu
def cpu(self,*a,**k):
    t0=time.perf_counter(); r=orig_cpu(self,*a,**k); t1=time.perf_counter()
    if self.is_cuda: ev.append(('sync',t0,t1))
    return r
torch.Tensor.cpu=cpu
import torch.autograd
orig_bwd=torch.Tensor.backward
def bwd(self,*a,**k):
    t0=time.perf_counter(); r=orig_bwd(self,*a,**k); t1=time.perf_counter(); ev.append(('bwd',t0,t1)); return r
torch.Tensor.backward=bwd
N=30; acc={}
tprev=time.perf_counter()
t_begin=tprev
for it in range(N):
    ev.clear()
    t0=time.perf_counter()
    tr.step(rays, gt, focal, noise=noise, update_controllers=False, fixed_chunk=4096)
    t1=time.perf_counter()
    prev=t0; out=[]; si=0
    for (k,a,b) in ev:
        if k=='sync':
            out.append((f'host{si}',a-prev)); out.append((f'wait{si}',b-a)); si+=1; prev=b
        else:
            out.append(('host_fwd_tail',a-prev)); out.append(('bwd_issue',b-a)); prev=b
    out.append(('opt+rest',t1-prev))
    for k,v in out: acc[k]=acc.get(k,0)+v/N*1e6
torch.cuda.synchronize()
print('ms/step', (time.perf_counter()-t_begin)/N*1e3)
print({k: round(v) for k,v in acc.items()})
