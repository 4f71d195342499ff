#!/bin/bash
# usage: tools/build_variant.sh NAME "-DCTL_LB_MID=2 ..."   -> csrc/variants/libctl_NAME.so (A/B builds; load with CTL_HIP_LIB)
set -e
cd "$(dirname "$0")/../cooperative_training_and_latent_space_data_augmentation_amd/csrc"
mkdir -p variants/obj_$1
F="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-gpu-rdc -Wall -Wno-unused-function -I../../include -I. $2"
for s in ctl_conv.hip ctl_elem.hip ctl_mask.hip ctl_io.hip; do /opt/rocm/bin/hipcc $F -c $s -o variants/obj_$1/$s.o & done
/opt/rocm/bin/hipcc $F -x hip -c ctl_plan.cpp -o variants/obj_$1/ctl_plan.o &
wait
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o variants/libctl_$1.so variants/obj_$1/*.o
rm -rf variants/obj_$1
echo built variants/libctl_$1.so
