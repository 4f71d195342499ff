#!/bin/bash
# pipelined fp32 weight gradient: kernel parity on the new cases, then the whole step: default (two-tensor launches with >= 8 tiles per block)
# vs the single-image loop everywhere vs the pipelined loop everywhere
timeout 900 python -m pytest tests/test_bf16_gpu.py -x -q -m gpu -k "virtual_output_gradient or bn_backward_prologue" 2>&1 | tail -3
bash tools/ab.sh gpurun_out/ab_wgpipe -r 3 "pipe_dy2|" "single_image|--lib cooperative_training_and_latent_space_data_augmentation_amd/csrc/variants/libctl_nopipe.so" "pipe_all|--lib cooperative_training_and_latent_space_data_augmentation_amd/csrc/variants/libctl_pipeall.so"
