import os, sys, time, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from attentionshift_amd import ops, roi_head as RH
torch.cuda.set_device(0)
def T(name, fn, reps=10):
    for _ in range(2): fn()
    torch.cuda.synchronize(); ts=[]
    for _ in range(reps):
        t0=time.perf_counter(); r=fn(); torch.cuda.synchronize(); ts.append((time.perf_counter()-t0)*1e3)
    print(f"{name:46s} med {np.median(ts):8.3f} ms  max {max(ts):8.3f}", flush=True); return r
g=torch.Generator().manual_seed(0)
prot=torch.randn(3,20,768,generator=g).cuda(); keep=(torch.rand(3,20,generator=g)>0.3).cuda()
T("merge_parts", lambda: RH.merge_parts(prot, keep, 0.85))
u=T("  _unit(prot)", lambda: RH._unit(prot))
link=T("  bmm u u^T >= thr", lambda: (u @ u.transpose(1,2)) >= 0.85)
host=T("  cat.cpu().numpy()", lambda: torch.cat((keep[:,None,:], link),dim=1).cpu().numpy())
wgt=np.zeros((3,5,20),dtype=np.float32); wgt[:,:,3]=1
w=T("  from_numpy.to(device)", lambda: torch.from_numpy(wgt).to(prot.device))
T("  bmm(wgt, prot)/(sum+eps)", lambda: torch.bmm(w, prot)/(w.sum(-1,keepdim=True)+1e-8))
feat=torch.randn(768,64,64,generator=g).cuda()
merged=[torch.randn(2,768,device='cuda'), [], torch.randn(3,768,device='cuda')]
sp=T("part_similarity", lambda: RH.part_similarity(merged, feat))
rois=torch.tensor([[20.,200,330,700],[360,210,670,710],[700,200,1010,700]]).cuda()
labels=torch.tensor([1,2,3]).cuda()
T("part_centers", lambda: RH.part_centers(sp, rois, labels, feat, num_max_obj=5))
maps=torch.rand(3,1024,1024,generator=g).cuda()
T("mask_sample_points", lambda: RH.mask_sample_points(maps, maps.flip(0).contiguous(), rois, 0.35, 0.8, 10, 21))
T("sample_point_grid bg", lambda: RH.sample_point_grid(maps, 20, 0.1, False))
fgb=(torch.rand(3,64,64,generator=g)>0.5).float().cuda()
T("grid_seed_coords", lambda: RH.grid_seed_coords(fgb, rois))
T("seed_features", lambda: RH.seed_features(torch.randint(0,1024,(4,20,2),generator=g).cuda(), feat))
T("feat.flatten(1).t().contiguous()", lambda: feat.flatten(1).t().contiguous())
