import sys, os, torch
sys.path.insert(0, "/root/repo" if os.path.exists("/root/repo/bench.py") else os.getcwd())
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__))))
import _variant
_variant.use_variant()
from cooperative_training_and_latent_space_data_augmentation_amd import _ffi, ops
from cooperative_training_and_latent_space_data_augmentation_amd._ffi import lib, check
DEV="cuda"
for n, cin in ((16,1),(16,4),(32,4)):
    cout,h=16,256
    x=torch.randn(n,cin,h,h,device=DEV); x = x.contiguous(memory_format=torch.channels_last) if cin>1 else x.contiguous()
    wt=torch.randn(cout,cin,3,3,device=DEV)*0.3
    total=((cout+15)//16)*3*256
    table=torch.tensor([[0,0,cout,cin,3,0,cin*9,9,3,1,total,4]],dtype=torch.int64,device=DEV)
    wp=torch.zeros(total,device=DEV)
    check(lib.ctl_pack_weights_batched(wt.contiguous().data_ptr(),wp.data_ptr(),table.data_ptr(),1,total,ops.stream_ptr()))
    d=_ffi.conv_desc(n=n,hin=h,win=h,cin=cin,hout=h,wout=h,cout=cout,ks=3,in_mode=_ffi.IN_C4,epi_flags=_ffi.EPI_STATS)
    y=ops.empty_nhwc(n,cout,h,h,DEV); st=torch.empty(lib.ctl_conv_stats_floats(_ffi.desc_ptr(d)),device=DEV)
    run=lambda: check(lib.ctl_conv_forward(_ffi.desc_ptr(d),x.data_ptr(),wp.data_ptr(),None,None,None,None,None,None,y.data_ptr(),st.data_ptr(),ops.stream_ptr()))
    for _ in range(5): run()
    torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30): run()
    e1.record(); torch.cuda.synchronize()
    us=e0.elapsed_time(e1)*1e3/30
    by=4.0*n*h*h*(cin+cout)
    print(f"C4 conv n{n} {cin}->16 @256: {us:.1f} us  {by/us/1e3:.0f} GB/s")
