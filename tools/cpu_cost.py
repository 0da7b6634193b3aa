import sys, os, time, collections; sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import torch, bench
dev=torch.device('cuda',0); torch.cuda.set_device(0)
from nmf_amd import synthetic, hip, functional
from nmf_amd.noise import DeviceNoise
from nmf_amd.trainer import Trainer
nerf, params = bench.build(dev)
tr=Trainer(nerf, params); noise=DeviceNoise(dev, 1)
rays, focal = synthetic.camera_rays(4096, seed=1); rays=rays.to(dev); gt=torch.rand(4096,3,device=dev)
for i in range(3): tr.step(rays, gt, focal, noise=noise, update_controllers=False, fixed_chunk=4096)
torch.cuda.synchronize()
acc=collections.defaultdict(lambda:[0,0.0])
def wrap(obj, name, label=None):
    f=getattr(obj,name)
    def g(*a,**k):
        t=time.perf_counter(); r=f(*a,**k); dt=time.perf_counter()-t
        e=acc[label or name]; e[0]+=1; e[1]+=dt
        return r
    setattr(obj,name,g)
for n in ['march_params','march_count','march_scan','march_fill','vm_query_fwd','composite_fwd','heads_fwd','select_bounces','bounce_index','expand_segments','bounce_prep_fwd','ggx_rays_fwd','brdf_mlp_fwd','sat_lookup_fwd','shade_mix_fwd','segment_sum','ray_compose_fwd',
          'vm_query_bwd','vm_query_bwd_segments','loss_mix_fwd','loss_mix_bwd','composite_bwd','heads_bwd','bounce_prep_bwd','ggx_rays_bwd','brdf_mlp_bwd','sat_lookup_bwd','shade_mix_bwd','ray_compose_bwd','segment_sum_wide','sat_build_bwd','vm_unpack_density_grad','adam_step','l1_mean_fwd','l1_mean_bwd','sqerr_fwd','sqerr_bwd','sat_build','vm_pack_density']:
    wrap(hip,n,'hip.'+n)
import nmf_amd.functional as F_
for cls in ['VMQuery','Composite','BouncePrep','GgxRays','BrdfMLP','MaterialHeads','ShadeMix','RayCompose','BounceRays','VMAppQuery','LossMix','ShadeCompose','VMQueryWeights','EnvLookup','FieldGrads','SatBuild','ParamGrads','StackedHeadGrads','L1Mean','SquaredError']:
    c=getattr(F_,cls)
    for m in ('forward','backward'):
        f=getattr(c,m)
        def mk(f,lab):
            def g(*a,**k):
                t=time.perf_counter(); r=f(*a,**k); dt=time.perf_counter()-t
                e=acc[lab]; e[0]+=1; e[1]+=dt
                return r
            return staticmethod(g)
        setattr(c,m,mk(f,f'{cls}.{m}'))
wrap(nerf.model.__class__,'shade_compact','Microfacet.shade_compact(incl)')
wrap(nerf.sampler.__class__,'sample_compact','sample_compact(incl sync)')
wrap(nerf.bg_module.__class__,'get_spherical_harmonics')
wrap(nerf.rf.__class__,'_tables','rf._tables')
wrap(nerf.rf.__class__,'_pass_token','rf._pass_token')
wrap(tr.optimizer.__class__,'step','optimizer.step')
N=20
t0=time.perf_counter()
for i in range(N): tr.step(rays, gt, focal, noise=noise, update_controllers=False, fixed_chunk=4096)
torch.cuda.synchronize()
print('ms/step', (time.perf_counter()-t0)/N*1e3)
rows=sorted(acc.items(), key=lambda kv:-kv[1][1])
for k,(n,t) in rows[:60]:
    print(f"{k:38s} {n/N:5.1f}x {t/N*1e6:8.1f} us/step")
hip_tot=sum(t for k,(n,t) in acc.items() if k.startswith('hip.'))/N*1e6
fn_f=sum(t for k,(n,t) in acc.items() if k.endswith('.forward'))/N*1e6
fn_b=sum(t for k,(n,t) in acc.items() if k.endswith('.backward'))/N*1e6
print(f"hip wrappers total {hip_tot:.0f} us/step over {sum(n for k,(n,t) in acc.items() if k.startswith('hip.'))/N:.0f} calls; Function.forward incl {fn_f:.0f}; Function.backward incl {fn_b:.0f}")
