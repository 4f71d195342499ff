for v in "$@"; do echo "=== $v"; CTL_HIP_LIB=$PWD/cooperative_training_and_latent_space_data_augmentation_amd/csrc/variants/libctl_$v.so python tools/debug/conv1x1_up.py 2>&1 | grep -E "^up="; done
