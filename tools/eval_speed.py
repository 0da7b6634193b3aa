import sys, time; sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import torch, bench
from nmf_amd import synthetic
from nmf_amd.noise import DeviceNoise
dev=torch.device('cuda',0); torch.cuda.set_device(0)
nerf,_=bench.build(dev); nerf.eval()
noise=DeviceNoise(dev,3)
for chunk in (4096, 16384, 65536):
    rays,focal=synthetic.camera_rays(chunk*6, seed=5); rays=rays.to(dev)
    with torch.no_grad():
        for w in range(2):
            nerf(rays[:chunk], focal, bg_col=torch.ones(3,device=dev), is_train=False, ndc_ray=False, noise=noise, draw_debug=False)
        torch.cuda.synchronize(); t0=time.perf_counter()
        for i in range(1,6):
            ims,st=nerf(rays[i*chunk:(i+1)*chunk], focal, bg_col=torch.ones(3,device=dev), is_train=False, ndc_ray=False, noise=noise, draw_debug=False)
        torch.cuda.synchronize(); dt=time.perf_counter()-t0
    print(f"eval chunk {chunk}: {5*chunk/dt:.0f} rays/s, {dt/5*1e3:.2f} ms/chunk, samples {st['n_samples']}")
