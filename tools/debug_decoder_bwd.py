"""GPU debug helper: compare per-block input gradients of the image decoder (ConvTranspose path) with the oracle."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import ref_cpu as O
from cooperative_training_and_latent_space_data_augmentation_amd import nets

sd = torch.load("tests/golden/state_dicts_seed0.pt", weights_only=False)
name = sys.argv[1] if len(sys.argv) > 1 else "image_decoder"
onet = O.build_networks(init=False)[name]; onet.load_state_dict(sd[name]); onet.train()
hnet = nets.build_networks(device="cuda", state_dicts={name: sd[name]})[name]; hnet.train()
for seed in range(int(sys.argv[2]) if len(sys.argv) > 2 else 12):
    print('==== seed', seed)
    for p_ in hnet.parameters(): p_.grad.zero_()
    onet.zero_grad()
    g = torch.Generator().manual_seed(seed)
    x = torch.relu(torch.randn(3, 128, 4, 4, generator=g))
    xo = x.clone().requires_grad_(True)
    grads = {}
    cur = xo
    feats = []
    for i, blk in enumerate([onet.up1, onet.up2, onet.up3, onet.up4]):
        cur = blk(cur); cur.retain_grad(); feats.append(cur)
    yo = onet.final_conv(cur)
    if onet.last_act is not None: yo = onet.last_act(yo)
    dy = torch.randn(yo.shape, generator=g)
    yo.backward(dy)
    xh = x.cuda().contiguous(memory_format=torch.channels_last).requires_grad_(True)
    yh = hnet(xh)
    print("fwd err", float((yh.cpu() - yo).abs().max()))
    yh.backward(dy.cuda().contiguous(memory_format=torch.channels_last))
    plan, bscr = hnet._dbg_last
    def fetch(t):
        (slot, off), n, h, w, c = t[0], t.n, t.h, t.w, t.c
        if slot != nets.S_BSCR: return None
        return bscr[off:off + 4 * n * h * w * c].view(torch.float32).view(n, h, w, c).permute(0, 3, 1, 2).cpu()
    for i in (4, 3, 2, 1):
        t = fetch(plan.rec[f"d_out{i}"])
        ref = feats[i - 1].grad
        print(f"d_out{i}", tuple(ref.shape), "err", float((t - ref).abs().max()), "max", float(ref.abs().max()))
    print("dx err", float((xh.grad.cpu() - xo.grad).abs().max()), "max", float(xo.grad.abs().max()))
    hp = dict(hnet.named_parameters())
    for n, p in onet.named_parameters():
        e = float((hp[n].grad.cpu() - p.grad).abs().max()); m = float(p.grad.abs().max())
        if e > 1e-3 * m + 1e-5: print("  grad", n, "err", e, "max", m)
