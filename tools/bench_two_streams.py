"""Do two independent decoder passes (D_img and D_seg on the two halves of the FTN code, forward + backward) overlap when
issued on two HIP streams?  Decides whether the solver should run them concurrently."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cooperative_training_and_latent_space_data_augmentation_amd import nets
torch.manual_seed(0)
m = nets.build_networks(device="cuda")
dimg, dseg = m["image_decoder"], m["segmentation_decoder"]
za = torch.relu(torch.randn(16, 128, 16, 16, device="cuda")).contiguous(memory_format=torch.channels_last).requires_grad_(True)
zb = torch.relu(torch.randn(16, 128, 16, 16, device="cuda")).contiguous(memory_format=torch.channels_last).requires_grad_(True)
side = torch.cuda.Stream()

def seq():
    ya = dimg(za); yb = dseg(zb)
    torch.autograd.backward([ya, yb], [torch.ones_like(ya), torch.ones_like(yb)])

def par():
    cur = torch.cuda.current_stream()
    side.wait_stream(cur)
    with torch.cuda.stream(side):
        ya = dimg(za)
        ga = torch.ones_like(ya)
    yb = dseg(zb)
    gb = torch.ones_like(yb)
    torch.autograd.backward([ya, yb], [ga, gb])      # each backward node runs on the stream of its forward
    cur.wait_stream(side)

def single(net, z):
    y = net(z); y.backward(torch.ones_like(y))

def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3

print(f"D_img fwd+bwd alone {timeit(lambda: single(dimg, za)):.2f} ms, D_seg alone {timeit(lambda: single(dseg, zb)):.2f} ms")
print(f"both, one stream   {timeit(seq):.2f} ms")
print(f"both, two streams  {timeit(par):.2f} ms")
