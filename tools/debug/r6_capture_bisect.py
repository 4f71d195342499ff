"""Round 6 bring-up: which stacking variant survives hipGraph capture of the dropout-scheme step?  Each variant runs in a child process
(a crash in hipStreamEndCapture takes the process down)."""
import os
import subprocess
import sys

VARIANTS = {
    "none": "()",
    "default": "(('image_decoder', 1), ('segmentation_decoder', 0), ('image_encoder', 0))",
    "all_main": "(('image_decoder', 0), ('segmentation_decoder', 0), ('image_encoder', 0))",
    "img_dec_side": "(('image_decoder', 1),)",
    "img_dec_main": "(('image_decoder', 0),)",
    "seg_dec": "(('segmentation_decoder', 0),)",
    "enc": "(('image_encoder', 0),)",
    "decs": "(('image_decoder', 1), ('segmentation_decoder', 0))",
}


def child(name):
    import torch
    ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    sys.path.insert(0, ROOT)
    from oracle import ref_cpu as O
    from cooperative_training_and_latent_space_data_augmentation_amd.graph import CooperativeStepGraph
    from cooperative_training_and_latent_space_data_augmentation_amd.solver import AdvancedTripletReconSegmentationModel
    torch.manual_seed(0)
    s = AdvancedTripletReconSegmentationModel(use_gpu=True)
    if VARIANTS[name] is not None:
        s.stack_passes = eval(VARIANTS[name])
    dev = lambda t: t.cuda().contiguous(memory_format=torch.channels_last) if t.dim() == 4 else t.cuda()
    clean, label, noisy = (dev(t) for t in O.synthetic_batch(4, 64, 64, seed=7))
    drop_i = {"loss_name": "mse", "mask_type": "dropout", "max_threshold": 0.5, "random_threshold": False, "if_soft": False}
    drop_s = {"loss_name": "ce", "mask_type": "dropout", "max_threshold": 0.5, "random_threshold": False, "if_soft": False}
    mode = os.environ.get("R6_MODE", "graph")
    if mode == "eager":
        for _ in range(3):
            losses = s.cooperative_step(clean, label, noisy, drop_i, drop_s)
    else:
        g = CooperativeStepGraph(s, drop_i, drop_s, replay=os.environ.get("R6_REPLAY", "runtime"))
        for _ in range(3):
            losses = g(clean, label, noisy)
    torch.cuda.synchronize()
    print(name, "ok", [round(float(v), 5) for v in losses], flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1:
        child(sys.argv[1])
    else:
        for name in VARIANTS:
            r = subprocess.run([sys.executable, __file__, name], capture_output=True, text=True)
            tail = (r.stdout.strip().splitlines() or [""])[-1]
            err = [l for l in r.stderr.splitlines() if "Error" in l or "error" in l or "fault" in l][-2:]
            print(f"{name:14s} rc={r.returncode} {tail} {err}", flush=True)
