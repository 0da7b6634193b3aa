import sys, time; sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import torch, bench
bench.GRID=300
dev=torch.device('cuda',0); torch.cuda.set_device(0)
from nmf_amd import synthetic
from nmf_amd.noise import DeviceNoise
from nmf_amd.trainer import Trainer
t0=time.time()
nerf, params = bench.build(dev)
print('build', round(time.time()-t0,1),'s', 'nSamples', nerf.rf.nSamples, 'mem GB', torch.cuda.max_memory_allocated()/1e9)
tr=Trainer(nerf, params); noise=DeviceNoise(dev, 1)
batches=[(synthetic.camera_rays(4096, seed=100+i)[0].to(dev), torch.rand(4096,3,device=dev)) for i in range(15)]
focal=synthetic.camera_rays(8,seed=0)[1]
for i in range(5): out=tr.step(*batches[i], focal, noise=noise, update_controllers=False, fixed_chunk=4096)
torch.cuda.synchronize(); t0=time.perf_counter()
for i in range(5,15): out=tr.step(*batches[i], focal, noise=noise, update_controllers=False, fixed_chunk=4096)
torch.cuda.synchronize(); dt=(time.perf_counter()-t0)/10
print('G=300:', round(dt*1e3,2),'ms/step', round(4096/dt),'rays/s', 'samples', out['n_samples'], 'loss', out['loss'], 'mem GB', round(torch.cuda.max_memory_allocated()/1e9,2))
