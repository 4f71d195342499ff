"""Debug probe: one captured cooperative step with CTL_SIDE_STREAM=1 (run under rocgdb for a native backtrace)."""
import os, sys, faulthandler
faulthandler.enable()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from cooperative_training_and_latent_space_data_augmentation_amd.solver import AdvancedTripletReconSegmentationModel
from cooperative_training_and_latent_space_data_augmentation_amd.graph import CooperativeStepGraph
torch.manual_seed(0)
s = AdvancedTripletReconSegmentationModel(use_gpu=True)
g = torch.Generator().manual_seed(7)
clean = torch.rand(4, 1, 96, 80, generator=g).cuda().contiguous(memory_format=torch.channels_last)
label = torch.randint(0, 4, (4, 96, 80), generator=g).cuda()
ci = {"loss_name": "mse", "mask_type": "channel", "max_threshold": 0.5, "random_threshold": False, "if_soft": False}
cs = {"loss_name": "ce", "mask_type": "spatial", "max_threshold": 0.5, "random_threshold": False, "if_soft": False}
step = CooperativeStepGraph(s, ci, cs)
print("capturing", flush=True)
for i in range(3):
    print(i, [float(v) for v in step(clean, label, clean)], flush=True)
