"""Opt-in execution paths that are OFF by default (measured slower or neutral on MI355X, kept for the record and for other parts):
they must stay correct.  Each runs in a subprocess because the switches are read once per process.
  CTL_FUSE_FINALIZE=1   BatchNorm finalize folded into the producing conv / reduction by the plan executor (ctl_plan.cpp)
  CTL_FUSE_CONSUMER=0|1 forward BatchNorm finalize inside the first blocks of the convolution that consumes the coefficients (ctl_bn_consume)
  CTL_FUSE_BNBWD=1      fp32 BatchNorm-backward reduction inside the data-gradient epilogue
  CTL_SIDE_STREAM=1     weight gradients on a library-owned side stream per launch chain (eager plans)
  CTL_SIDE_STREAM=2     side lanes also inside a captured step (correct; the replay is slower with the extra cross-stream edges)"""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r"""
import json, sys, torch
sys.path.insert(0, %r)
from cooperative_training_and_latent_space_data_augmentation_amd.solver import AdvancedTripletReconSegmentationModel
from cooperative_training_and_latent_space_data_augmentation_amd import _ffi
torch.manual_seed(0)
s = AdvancedTripletReconSegmentationModel(use_gpu=True)
g = torch.Generator().manual_seed(7)
clean = torch.rand(4, 1, 96, 80, generator=g).cuda().contiguous(memory_format=torch.channels_last)
label = torch.randint(0, 4, (4, 96, 80), generator=g).cuda()
ci = {"loss_name": "mse", "mask_type": "channel", "max_threshold": 0.5, "random_threshold": False, "if_soft": False}
cs = {"loss_name": "ce", "mask_type": "spatial", "max_threshold": 0.5, "random_threshold": False, "if_soft": False}
out = []
n0 = _ffi.lib.ctl_launch_count()
if GRAPH:
    from cooperative_training_and_latent_space_data_augmentation_amd.graph import CooperativeStepGraph
    step = CooperativeStepGraph(s, ci, cs)
    for _ in range(3):
        out.append([float(v) for v in step(clean, label, clean)])
else:
    for _ in range(2):
        out.append([float(v) for v in s.cooperative_step(clean, label, clean, ci, cs)])
torch.cuda.synchronize()
launches = int(_ffi.lib.ctl_launch_count() - n0) // 2
sums = {k: float(m._flat_data.double().sum()) for k, m in s.model.items()}
bufs = {k: float(m._bflat.double().sum()) for k, m in s.model.items()}
print("RESULT " + json.dumps({"losses": out, "sums": sums, "bufs": bufs, "launches": launches, "chain_overlap": s.chain_overlap}))
""" % ROOT


def run(env_extra, graph=False):
    env = dict(os.environ, **env_extra)
    r = subprocess.run([sys.executable, "-c", f"GRAPH = {graph}\n" + SCRIPT], capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    line = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")]
    assert line, r.stderr[-2000:]
    return json.loads(line[0][7:])


@pytest.fixture(scope="module")
def default_run():
    return run({})


@pytest.mark.parametrize("env", [{"CTL_FUSE_FINALIZE": "1"}, {"CTL_FUSE_BNBWD": "1"}, {"CTL_SIDE_STREAM": "1"}, {"CTL_FUSE_CONSUMER": "1"},
                                 {"CTL_FUSE_CONSUMER": "0"}])
def test_optin_path_matches_default(env, default_run):
    got = run(env)
    for a, b in zip(got["losses"], default_run["losses"]):
        for x, y in zip(a, b):
            assert abs(x - y) <= 2e-5 * max(1.0, abs(y)), (env, a, b)
    for k in default_run["sums"]:
        # two Adam steps move every weight by ~+-lr = 1e-4: a gradient that is rounding noise around 0 (the order of the fp32 sums differs
        # between the paths) may flip its sign and move the SUM of a network's ~1e6 weights by 2e-4 per flipped element
        assert abs(got["sums"][k] - default_run["sums"][k]) <= 5e-2, (env, k, got["sums"][k], default_run["sums"][k])
        assert abs(got["bufs"][k] - default_run["bufs"][k]) <= 1e-3 + 1e-4 * abs(default_run["bufs"][k]), (env, k)
    if "CTL_FUSE_FINALIZE" in env:          # the stand-alone finalize launches are gone (one table-write launch per plan instead)
        assert got["launches"] < default_run["launches"] - 100, (got["launches"], default_run["launches"])
    from cooperative_training_and_latent_space_data_augmentation_amd._ffi import lib
    if env.get("CTL_FUSE_CONSUMER") == "1" and lib.ctl_consumer_finalize_built():  # the forward finalize launches are gone
        other = run({"CTL_FUSE_CONSUMER": "0"})
        assert got["launches"] < other["launches"] - 50, (got["launches"], other["launches"])


def test_side_stream_inside_a_captured_step():
    """hipGraph mode with CTL_SIDE_STREAM=2: the fork / join events of the side lanes become graph dependencies (the lanes are created by
    the eager warm-up step in front of the capture).  Same kernels on the same data: the replays' losses and the weights after three
    steps equal the default graph run bit for bit."""
    ref = run({}, graph=True)
    got = run({"CTL_SIDE_STREAM": "2"}, graph=True)
    assert got["losses"] == ref["losses"], (got["losses"], ref["losses"])
    assert got["sums"] == ref["sums"] and got["bufs"] == ref["bufs"]


def test_two_chains_sit_on_two_hardware_queues(default_run):
    """solver._ensure_chains_overlap: before the first two-chain step the pair of streams is probed with two idle kernels (ctl_spin) and
    the second chain is moved to another stream until they run side by side.  With ONE hardware queue for the whole process
    (GPU_MAX_HW_QUEUES=1) no stream can overlap the main one: the probe reports it after 8 streams and the step is still the same step."""
    ov = default_run["chain_overlap"]
    assert ov is not None and ov["overlap"] and ov["probe_ratio"] < 1.5, ov
    one = run({"GPU_MAX_HW_QUEUES": "1"})
    assert one["chain_overlap"]["overlap"] is False and one["chain_overlap"]["streams_tried"] == 8, one["chain_overlap"]
    for a, b in zip(one["losses"], default_run["losses"]):
        for x, y in zip(a, b):
            assert abs(x - y) <= 2e-5 * max(1.0, abs(y)), (a, b)
