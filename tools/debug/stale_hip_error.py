"""Debug: which call leaves a stale HIP error behind that the next CTL_LAUNCH_CHECK picks up?  (build() followed by smoke() in one process)"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__ as g
hip = None
def last(tag):
    global hip
    import torch
    if hip is None:
        hip = ctypes.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))
        hip.hipGetErrorString.restype = ctypes.c_char_p
    e = hip.hipPeekAtLastError()
    print(f"{tag}: hipPeekAtLastError = {e} ({hip.hipGetErrorString(e).decode()})", flush=True)
g.build()
last("after build()")
import torch
torch.cuda.set_device(0)
last("after set_device")
x = torch.zeros(4, device="cuda")
last("after first allocation")
from cooperative_training_and_latent_space_data_augmentation_amd._ffi import lib
print("ctl_spin rc", lib.ctl_spin(1, torch.cuda.current_stream().cuda_stream), lib.ctl_last_error())
last("after ctl_spin")
