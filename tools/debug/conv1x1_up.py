import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, torch.nn.functional as F
from cooperative_training_and_latent_space_data_augmentation_amd import _ffi, ops
def dev(x): return x.cuda().contiguous(memory_format=torch.channels_last) if x.dim()==4 else x.cuda()
n,cin,cout,h,w=16,16,16,64,64
g=torch.Generator().manual_seed(1)
x=torch.randn(n,cin,h,w,generator=g); wt=torch.randn(cout,cin,1,1,generator=g)*0.3; b=torch.randn(cout,generator=g)
for up in (1,):
  ho,wo=(2*h,2*w) if up else (h,w)
  v=torch.randn(n,cout,ho,wo,generator=g); rs,rh=torch.rand(cout,generator=g)+0.5, torch.randn(cout,generator=g)
  xi=F.interpolate(x,scale_factor=2,mode="nearest") if up else x
  for name,flags in (("res",_ffi.EPI_RES),("accum",_ffi.EPI_ACCUM)):
    d=_ffi.conv_desc(n=n,hin=h,win=w,cin=cin,hout=ho,wout=wo,cout=cout,ks=1,in_mode=_ffi.IN_UP2 if up else 0,epi_flags=flags)
    kw=dict(bias=dev(b))
    if flags&_ffi.EPI_RES: kw.update(res=dev(v),res_scale=dev(rs),res_shift=dev(rh))
    kw={}
    if flags&_ffi.EPI_RES: kw.update(res=dev(v),res_scale=dev(rs),res_shift=dev(rh))
    if flags&_ffi.EPI_ACCUM: kw.update(y=dev(v.clone()))
    y,_=ops.conv_forward(d,dev(x),ops.pack_oihw_fwd(dev(wt)),**kw)
    conv=F.conv2d(xi,wt)
    ref=conv+((v*rs.view(1,-1,1,1)+rh.view(1,-1,1,1)) if flags&_ffi.EPI_RES else v)
    e=(y.cpu()-ref).abs()
    bad=(e>1e-3)
    print(f"up={up} {name}: max err {e.max():.3e} bad frac {bad.float().mean():.4f}")
    if bad.any():
        idx=bad.nonzero()
        print("  bad n:",sorted(set(idx[:,0].tolist()))[:20]," ch:",sorted(set(idx[:,1].tolist()))[:16])
        for k in range(3):
            a,c,r,q=idx[k*997 % len(idx)].tolist()
            print("   at",(a,c,r,q),"got",float(y[a,c,r,q]),"ref",float(ref[a,c,r,q]),"conv",float(conv[a,c,r,q]),"v",float(v[a,c,r,q]),"rs",float(rs[c]),"rh",float(rh[c]), "v nearby ch", v[a,:,r,q].tolist()[:16])
        print("  bad rows:",sorted(set(idx[:,2].tolist()))[:40]); print("  bad cols:",sorted(set(idx[:,3].tolist()))[:40])
