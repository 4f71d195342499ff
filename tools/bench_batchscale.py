"""How does one STN pass (E_s -> D_s, forward + full backward) scale with batch?  Decides whether grouped STN passes pay."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cooperative_training_and_latent_space_data_augmentation_amd import nets
torch.manual_seed(0)
m = nets.build_networks(device="cuda")
enc, dec = m["shape_encoder"], m["shape_decoder"]
for n in (16, 32, 64):
    x = torch.rand(n, 4, 256, 256, device="cuda").contiguous(memory_format=torch.channels_last).requires_grad_(True)
    def run():
        y = dec(enc(x)); y.backward(torch.ones_like(y))
    for _ in range(3): run()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): run()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
    print(f"batch {n}: {dt*1e3:.2f} ms per fwd+bwd  ({dt*1e3/n*16:.2f} ms per 16 slices)")
