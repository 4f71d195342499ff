# latent-mask kernel at the configured size: blocks per image (S) vs time in a graph replay (tuning build)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
bash tools/build_variant.sh tuning "-DCTL_TUNING" ctl_mask.hip ctl_plan.cpp > /dev/null 2>&1
for S in 1 2 4 8; do
CTL_MASK_S=$S CTL_TOOL_LIB=tuning python3 - <<'PY'
import os, sys, torch
sys.path.insert(0, "tools"); sys.path.insert(0, ".")
from _variant import use_variant
use_variant("tuning")
import bench
r = bench.latent_mask_roofline(torch.device("cuda"))["configured_16x128x16x16"]
print("S =", os.environ["CTL_MASK_S"], "eager us %.2f  graph-replay us %.2f  (frac %.3f)" % (r["us_per_call"], r["graph_replay_us_per_call"], r["graph_replay_frac"]))
PY
done
