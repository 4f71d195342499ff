#!/usr/bin/env python3
"""Experiment (round 5): the two launch chains of the cooperative step on DISJOINT halves of the chip (CU-masked HIP streams, hipExtStreamCreateWithCUMask)
instead of two chip-filling chains that time-slice.  Idea: ~850 launches of ~20 us, of which ~10 us are per-launch latency (launch gap, exposed first
loads, tails): on half the CUs a kernel's variable part doubles but its fixed part does not, and the other half of the chip works meanwhile.
Needs the tuning build (CTL_TOOL_LIB=tuning) with CTL_NUM_CUS=128 so that the persistent grids match the masked streams.
    python tools/cumask_probe.py [split|none] [steps]"""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import _variant
_variant.use_variant()
from cooperative_training_and_latent_space_data_augmentation_amd.solver import AdvancedTripletReconSegmentationModel
import bench

mode = sys.argv[1] if len(sys.argv) > 1 else "split"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
hip = ctypes.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))


def masked_stream(bits):
    words = (ctypes.c_uint32 * 8)(*[(bits >> (32 * i)) & 0xffffffff for i in range(8)])
    s = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(s), 8, words)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(s.value)


torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
torch.manual_seed(0)
solver = AdvancedTripletReconSegmentationModel(network_type="FCN_16_standard", image_ch=1, num_classes=4, learning_rate=1e-4, use_gpu=True)
clean, label, noisy, _ = bench.synthetic(16, 256, 256, 1000, dev)
full = (1 << 256) - 1
if mode == "split":          # bit i -> XCC i % 8, CU i / 8 of that XCC (kfd: the mask is dealt round-robin over the XCCs): lower half = CUs 0-15 of every XCD
    lo = (1 << 128) - 1
    main, side = masked_stream(lo), masked_stream(full ^ lo)
elif mode == "xcd":          # XCDs 0-3 / 4-7
    m = 0
    for i in range(256):
        if (i % 8) < 4:
            m |= 1 << i
    main, side = masked_stream(m), masked_stream(full ^ m)
else:
    main, side = torch.cuda.Stream(), torch.cuda.Stream()
solver._side = side
solver._queue_checked = main.cuda_stream          # (no re-probing of the pair: the streams are the experiment)
IMG, SEG = bench.DROP_IMG, bench.DROP_SEG
torch.cuda.synchronize()          # (weights and inputs were produced on the default stream)
with torch.cuda.stream(main):
    for _ in range(5):
        solver.cooperative_step(clean, label, noisy, IMG, SEG)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        losses = solver.cooperative_step(clean, label, noisy, IMG, SEG)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
print(f"mode {mode}  CTL_NUM_CUS={os.environ.get('CTL_NUM_CUS', '256')}: {1e3 * dt:.3f} ms/step  {16 / dt:.1f} slices/s  losses {[round(float(v), 4) for v in losses][:3]}")
