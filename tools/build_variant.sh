#!/bin/bash
# usage: tools/build_variant.sh NAME "-DCTL_LB_MID=2 ..." [file.hip ...]   -> csrc/variants/libctl_NAME.so
#   A/B builds: bench.py --lib <path>, tools with CTL_TOOL_LIB=NAME (tools/_variant.py).  `tools/build_variant.sh tuning -DCTL_TUNING` gives the
#   library whose tuning hooks (CTL_FORCE_CFG, CTL_PERSIST, CTL_WGRAD_SLOTS, CTL_MASK_SPLIT, CTL_PROF_TIMELINE, ...) read the environment;
#   the shipped build has none (tests/test_cabi.py).
# With a file list only those sources are recompiled with the flags; the other objects come from the default build (csrc/build/).
set -e
cd "$(dirname "$0")/../cooperative_training_and_latent_space_data_augmentation_amd/csrc"
name=$1; flags=$2; shift; shift
all="ctl_conv.hip ctl_conv_x3.hip ctl_conv_bf16.hip ctl_wgrad_bf16.hip ctl_wgrad_x3.hip ctl_elem.hip ctl_mask.hip ctl_io.hip ctl_plan.cpp"
files=${*:-$all}
mkdir -p variants/obj_$name
F="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-gpu-rdc -Wall -Wno-unused-function -I../../include -I. $flags"
objs=""
for s in $all; do
  if echo " $files " | grep -q " $s "; then
    /opt/rocm/bin/hipcc $F -x hip -c $s -o variants/obj_$name/$s.o 2> variants/obj_$name/$s.log &
    objs="$objs variants/obj_$name/$s.o"
  else
    objs="$objs build/$s.o"
  fi
done
wait
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o variants/libctl_$name.so $objs
rm -rf variants/obj_$name
echo built variants/libctl_$name.so
