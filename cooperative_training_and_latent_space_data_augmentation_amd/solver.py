"""`AdvancedTripletReconSegmentationModel` -- drop-in for the reference solver
(medseg/models/advanced_triplet_recon_segmentation_model.py:24-813) on the MI355X engine.

Same constructor, method names, argument meaning and return values as upstream, so
`medseg/train_adv_supervised_segmentation_triplet.py` can import this class instead.  The five
networks live in `self.model` (dict name -> nn.Module, reference state_dict keys); compute goes through compiled HIP
plans (nets.py), losses / STN inputs through fused HIP ops (autograd.py), the optimizer is one flat Adam launch per
network (optim.py).  Optional keyword-only `override` dicts inject the reference's random draws for parity tests.
"""
from __future__ import annotations

import gc
import itertools
import os
import random
from os.path import join
from typing import Dict, Optional

import torch
import torch.nn as nn

from . import ops
from .autograd import cross_entropy_2D, net_apply, scaled_mse, softmax_t, split_halves
from .metrics import runningScore
from .model_util import (_disable_tracking_bn_stats, _draw_seed, mask_latent_code_channel_wise,
                         mask_latent_code_spatial_wise, set_grad)
from .nets import build_networks
from .optim import FlatAdam

_DEFAULT_IMG_CFG = {"loss_name": "mse", "mask_type": "random", "max_threshold": 0.5, "random_threshold": True, "if_soft": True}
_DEFAULT_SEG_CFG = {"loss_name": "ce", "mask_type": "random", "max_threshold": 0.5, "random_threshold": True, "if_soft": True}


def basic_loss_fn(pred, target, loss_type="cross entropy"):
    """custom_loss.py:8-19 -- the solver only ever asks for 'cross entropy'."""
    if loss_type != "cross entropy":
        raise NotImplementedError(loss_type)
    return cross_entropy_2D(pred, target)


SPLIT_BACKWARD = True       # one backward() sweep per launch chain, the standard branch first (see _cooperative_step)
# Networks whose standard and hard-example pass of a cooperative step run their backward STACKED (nets.PassStack, round 6; VERDICT r5 "next" #1):
# one launch chain over both passes (n = 32 in two BatchNorm groups) instead of two n = 16 chains, one weight-gradient contraction over both.
# Entries: (network, launch chain its stacked backward is issued on: 0 = main stream, 1 = second stream), in issue order (a network comes after
# every network that consumes its outputs), e.g. STACK_ALL_FTN below.
# MEASURED AND NOT ADOPTED as the default (profiles/r6_stack_experiments.txt): parity-green (tests/test_stack_gpu.py) and 142 launches fewer per
# step, but the serial kernel time only drops 2.5 % (the grouped weight-gradient launches and the persistent-grid convs already run at n = 16
# what they run at n = 32), while the two SYMMETRIC per-pass chains become one dependent chain STN -> D_seg -> E_i with D_img beside it: time with
# two kernels in flight 10.5 -> 8.0 ms of the step.  fp32 14.46 -> 14.80 ms, bf16 9.20 -> 11.4 ms same-box.  () = every pass runs its own backward.
STACK_ALL_FTN = (("image_decoder", 1), ("segmentation_decoder", 0), ("image_encoder", 0))
STACK_PASSES = ()
SPLIT_WGRAD_TAIL = True     # a stacked backward runs its weight-gradient family on the OTHER launch chain, behind its data-gradient chain

class AdvancedTripletReconSegmentationModel(nn.Module):
    def __init__(self, network_type="FCN_16_standard", image_ch=1, learning_rate=1e-4, encoder_dropout=None,
                 decoder_dropout=None, num_classes=4, n_iter=1, checkpoint_dir=None, use_gpu=True, debug=False, *, compute_dtype=None):
        """`compute_dtype` (keyword-only, no upstream counterpart): "fp32" = the reference's arithmetic (default, BASELINE config 2);
        "bf16" = BASELINE config 3: network-internal activations / gradients stored as bf16, convolutions on bf16 MFMA with fp32
        accumulation, fp32 master weights / BatchNorm statistics / losses."""
        super().__init__()
        self.compute_dtype = compute_dtype or "fp32"
        if network_type not in ("FCN_16_standard", "FCN_16_standard_w_o_filter", "FCN_16_standard_share_code"):
            raise NotImplementedError(network_type)
        if not use_gpu:
            raise RuntimeError("this engine runs on MI355X only (use_gpu=True); the CPU path is the oracle under oracle/")
        self.network_type, self.image_ch, self.checkpoint_dir = network_type, image_ch, checkpoint_dir
        self.num_classes, self.learning_rate, self.n_iter = num_classes, learning_rate, n_iter
        self.encoder_dropout, self.decoder_dropout = encoder_dropout, decoder_dropout
        self.use_gpu, self.debug = use_gpu, debug
        self.device = torch.device("cuda", torch.cuda.current_device())
        self.model = self.get_network(checkpoint_dir=checkpoint_dir)
        self.optimizers = None
        self.reset_all_optimizers()
        self.latent_code = {"image": None, "segmentation": None, "shape": None}
        self.running_metric = self.set_running_metric()
        self.cur_eval_images = self.cur_eval_predicts = self.cur_eval_gts = None
        self.cur_time_predicts = {}
        self.loss = 0.0
        self.z_i = self.z_s = None
        self.last_masks = {}
        self.grad_scale = 1.0          # set to 1/world_size by the data-parallel wrapper
        self._dp = None                # dist.DataParallel: per-network gradient exchange, waited for in front of each Adam launch
        # independent STN passes of one step that share the BatchNorm mode run as one grouped pass (recon_shape_pair); off = one
        # pass per reference call
        self.group_stn_passes = True
        # the image decoder consumes z_i only: its launch chain (forward, loss, and through autograd its backward) can run on a
        # second HIP stream next to D_seg -> STN on the main stream
        self.two_streams = True
        self.split_backward = SPLIT_BACKWARD and self.compute_dtype != "bf16"      # (bf16: no gain eager, and the captured step replays 16 % slower)
        self.stack_passes = tuple(STACK_PASSES)      # () = every pass runs its own backward (rounds 1-5)
        self.split_wgrad_tail = SPLIT_WGRAD_TAIL     # the weight-gradient family of a stacked backward runs on the OTHER launch chain
        # parameter gradients of the passes of a step are parked and added with one launch per network after backward (nets.py)
        self.defer_param_grads = True
        self._side = torch.cuda.Stream(device=self.device)
        if self.two_streams and hasattr(torch.autograd.graph, "set_warn_on_accumulate_grad_stream_mismatch"):
            # the flat-parameter leaves live on the main stream while part of their gradient is produced on the second one: intended
            torch.autograd.graph.set_warn_on_accumulate_grad_stream_mismatch(False)
        self._side_pending = False
        self._queue_checked, self.chain_overlap = None, None
        self._chain_events = None        # list collecting per-step stream events (tools/chain_timing.py)
        self._in_side = False            # True while the hard-example branch of cooperative_step is being issued on the side stream
        # HIP-graph capture (graph.py): while set, nothing that changes from step to step may be a launch argument -- RNG seeds and the
        # Adam step count come from the device int64[3] `_gstate` (advanced by ops.step_tick), random thresholds k from the device
        # int32 buffers `_gk[...]` that the host refreshes before every replay
        self._gstate = None
        self._gk = None
        self._perturb_calls = 0          # call-site salt of the device RNG within one step
        self._step_schemes = []          # mask schemes drawn since the start of the current cooperative_step
        self.training = True

    # ------------------------------------------------------------------ construction / checkpoints
    def get_network(self, checkpoint_dir=None):
        """model.py:76-149: fresh weights follow the reference's init for the current torch seed; with a checkpoint
        directory, `<name>.pth` state dicts are loaded (model.py:114-131,157-173)."""
        model = build_networks(self.image_ch, self.num_classes, 4, device=self.device, dtype=self.compute_dtype,
                               encoder_dropout=self.encoder_dropout, decoder_dropout=self.decoder_dropout)
        if checkpoint_dir:
            for name, net in model.items():
                self.init_model(net, resume_path=join(checkpoint_dir, name + ".pth"))
        return model

    def init_model(self, model, resume_path=None):
        if resume_path:
            assert os.path.exists(resume_path), "path: {} must exist".format(resume_path)
            sd = torch.load(resume_path, map_location="cpu")
            if isinstance(sd, dict) and "model_state" in sd:
                sd = sd["model_state"]
            model.load_state_dict(sd)
        return model

    def parameters(self):
        return itertools.chain(*[m.parameters() for m in self.model.values()])

    def named_parameters(self):
        return itertools.chain(*[m.named_parameters() for m in self.model.values()])

    def save_model(self, save_dir, epoch_iter, model_prefix=None, save_optimizers=False):
        """model.py:666-678: <save_dir>/<epoch>/checkpoints/<name>.pth (+ <name>_optim.pth)."""
        epoch_path = join(save_dir, str(epoch_iter), "checkpoints")
        os.makedirs(epoch_path, exist_ok=True)
        for name, net in self.model.items():
            torch.save({k: v.detach().cpu().clone() for k, v in net.state_dict().items()}, join(epoch_path, f"{name}.pth"))
        if save_optimizers:
            for name, opt in self.optimizers.items():
                torch.save(opt.state_dict(), join(epoch_path, f"{name}_optim.pth"))

    def save_snapshots(self, save_dir, epoch, model_prefix="interrupted"):
        """model.py:680-701."""
        epoch_path = join(save_dir, "interrupted", "checkpoints")
        os.makedirs(epoch_path, exist_ok=True)
        save_path = join(epoch_path, self.network_type + ".pkl")
        state = {"network_type": self.network_type, "epoch": epoch,
                 "model_state": {k: {n: v.detach().cpu().clone() for n, v in m.state_dict().items()} for k, m in self.model.items()},
                 "optimizer_state": {k: o.state_dict() for k, o in self.optimizers.items()}}
        torch.save(state, save_path)
        return save_path

    def load_snapshots(self, file_path):
        """model.py:703-738."""
        if not file_path or not os.path.exists(file_path):
            print(f"warning: {file_path} does not exists")
            return 0
        ckpt = torch.load(file_path, map_location="cpu")
        for k, net in self.model.items():
            net.load_state_dict(ckpt["model_state"][k])
        for k, opt in self.optimizers.items():
            opt.load_state_dict(ckpt["optimizer_state"][k])
        return ckpt["epoch"]

    # ------------------------------------------------------------------ mode switches / optimizers (model.py:740-795)
    def train(self, if_testing=False):
        self.training = True          # upstream never clears this inside train(); eval() clears it first
        for net in self.model.values():
            if not if_testing:
                net.train()
                set_grad(net, True)
            else:
                net.eval()
        return self

    def eval(self):
        self.training = False
        self.train(if_testing=True)
        return self

    def set_optimizers(self):
        self.optimizers = {name: FlatAdam(net, lr=self.learning_rate) for name, net in self.model.items()}

    def reset_all_optimizers(self):
        if self.optimizers is None:
            self.set_optimizers()
        for o in self.optimizers.values():
            o.zero_grad()

    def zero_grad(self):
        self.reset_all_optimizers()

    def get_optimizer(self, model_name=None):
        return self.optimizers if model_name is None else self.optimizers[model_name]

    def optimize_all_params(self):
        for name, o in self.optimizers.items():
            if self._dp is not None:
                self._dp.wait(name)              # this network's range of the gradient bucket (no-op if the exchange was already waited for)
            o.step(grad_scale=self.grad_scale, state=self._gstate)

    def optimize_params(self, model_name):
        if self._dp is not None:
            self._dp.wait(model_name)
        self.optimizers[model_name].step(grad_scale=self.grad_scale, state=self._gstate)

    def reset_optimizer(self, model_name):
        self.optimizers[model_name].zero_grad()

    def set_running_metric(self):
        return runningScore(n_classes=self.num_classes)

    def get_modules(self):
        return self.model.values()

    # ------------------------------------------------------------------ network plumbing (model.py:182-287)
    def _enc(self, x, disable_track_bn_stats=False):
        enc = self.model["image_encoder"]
        if disable_track_bn_stats:
            with _disable_tracking_bn_stats(enc):
                z_i, z_s = enc(x)
        else:
            z_i, z_s = enc(x)
        if "share_code" in self.network_type:
            z_i = z_s
        elif "w_o_filter" in self.network_type:
            z_s = z_i
        return z_i, z_s

    @staticmethod
    def _call(net, x, disable_track_bn_stats=False):
        if disable_track_bn_stats:
            with _disable_tracking_bn_stats(net):
                return net(x)
        return net(x)

    def encode_image(self, input, disable_track_bn_stats=False):
        z_i, z_s = self._enc(input, disable_track_bn_stats)
        self.latent_code["image"], self.latent_code["segmentation"] = z_i, z_s
        return z_i, z_s

    def decode_image(self, latent_code, disable_track_bn_stats=False):
        return self._call(self.model["image_decoder"], latent_code, disable_track_bn_stats)

    # ------------------------------------------------------------------ the two launch chains need two HARDWARE queues
    @staticmethod
    def _overlap_probe(a, b, us=300) -> float:
        """Two idle 300-us kernels, one per stream, between two events: ~1.0 = they ran side by side, ~2.0 = one behind the other."""
        from .hipgraph import streams_overlap_ratio
        return streams_overlap_ratio(a, b, us)

    def _ensure_chains_overlap(self):
        """HIP streams share a few hardware queues (4 by default) in creation order, and two streams on one queue run IN ORDER: the
        second chain then overlaps nothing (measured: 18.2 -> 22.5 ms per step once RCCL's streams had shifted the assignment,
        profiles/README.md round 2).  Before the first two-chain step on a stream the pair is probed with two idle kernels; on a
        collision the second chain moves to the next stream of torch's pool (the next queue) until the probe shows overlap."""
        if torch.cuda.is_current_stream_capturing():
            return
        cur = torch.cuda.current_stream()
        if self._queue_checked == cur.cuda_stream:
            return
        self._queue_checked = cur.cuda_stream
        ratio, attempts = self._overlap_probe(cur, self._side), 1
        while ratio > 1.5 and attempts < 8:
            self._side = torch.cuda.Stream(device=self.device)
            ratio, attempts = self._overlap_probe(cur, self._side), attempts + 1
        self.chain_overlap = {"probe_ratio": round(ratio, 2), "streams_tried": attempts, "overlap": ratio <= 1.5}

    def _fork_side(self, fn, *inputs):
        """Run fn() on the side stream (after everything issued so far on the current stream) when two_streams is on; the result
        must not be touched on the main stream before _join_side()."""
        if not self.two_streams:
            return fn()
        if self._in_side:
            return fn()
        cur = torch.cuda.current_stream()
        self._side.wait_stream(cur)
        for t in inputs:
            t.record_stream(self._side)
        with torch.cuda.stream(self._side):
            out = fn()
        for t in (out if isinstance(out, tuple) else (out,)):
            t.record_stream(cur)
        self._side_pending = True
        return out

    def _join_side(self):
        if self._side_pending:
            torch.cuda.current_stream().wait_stream(self._side)
            self._side_pending = False

    def _image_recon_loss(self, z_i, clean_image_l, disable_track_bn_stats=False):
        def work():
            ev = getattr(self, "_img_std_done", None)
            if self._in_side and ev is not None:
                # two-chain step: the image decoder is the one network whose BatchNorm running statistics are written on BOTH
                # chains (standard pass on the main chain, this pass on the side chain; upstream's decode_image always tracks,
                # model.py:444).  The standard pass comes first in the reference: make that an event, not a matter of timing.
                torch.cuda.current_stream().wait_event(ev)
            return scaled_mse(self.decode_image(z_i, disable_track_bn_stats), clean_image_l, 0.5)
        return self._fork_side(work, z_i, clean_image_l)

    def encode_shape(self, segmentation, is_label_map=False, disable_track_bn_stats=False, temperature=2):
        """construct_input (basic_operations.py:110-158): one-hot(label) or softmax(logit / T)."""
        if is_label_map:
            inp = ops.onehot(segmentation, self.num_classes)
        else:
            inp = softmax_t(segmentation, temperature)
        code = self._call(self.model["shape_encoder"], inp, disable_track_bn_stats)
        self.latent_code["shape"] = code
        return code

    def decode_shape(self, latent_code, disable_track_bn_stats=False):
        return self._call(self.model["shape_decoder"], latent_code, disable_track_bn_stats)

    def recon_shape(self, segmentation_logit, is_label_map=False, disable_track_bn_stats=False):
        return self.decode_shape(self.encode_shape(segmentation_logit, is_label_map, disable_track_bn_stats), disable_track_bn_stats)

    def recon_shape_pair(self, seg_a, a_is_label_map, seg_b, b_is_label_map, disable_track_bn_stats=False, temperature=2):
        """recon_shape(seg_a, a_is_label_map) followed by recon_shape(seg_b, b_is_label_map) in the same BatchNorm mode, run as
        ONE pass over the stacked batch with per-call BatchNorm statistics (ctl_conv.groups): the same numbers as the two
        calls (running statistics updated in call order), half the launches.  Returns (recon_a, recon_b)."""
        def stn_input(seg, is_label_map):
            return ops.onehot(seg, self.num_classes) if is_label_map else softmax_t(seg, temperature)
        ia, ib = stn_input(seg_a, a_is_label_map), stn_input(seg_b, b_is_label_map)
        n = ia.shape[0]
        if ib.shape != ia.shape:
            raise ValueError("recon_shape_pair: both inputs must have the same shape")
        inp = torch.cat([ia, ib], 0)
        enc, dec = self.model["shape_encoder"], self.model["shape_decoder"]
        if disable_track_bn_stats:
            with _disable_tracking_bn_stats(enc), _disable_tracking_bn_stats(dec):
                code = net_apply(enc, inp, groups=2)[0]
                out = net_apply(dec, code, groups=2)[0]
        else:
            code = net_apply(enc, inp, groups=2)[0]
            out = net_apply(dec, code, groups=2)[0]
        self.latent_code["shape"] = code[n:]
        return split_halves(out)

    def decode_segmentation_from_image_code(self, latent_code_i, disable_track_bn_stats=False):
        """FTN: z_i -> z_s -> segmentation (model.py:208-221).  Forward only (`Dual_Branch_Encoder.filter_code`)."""
        enc, dec = self.model["image_encoder"], self.model["segmentation_decoder"]
        if disable_track_bn_stats:
            with _disable_tracking_bn_stats(enc):
                z_s = enc.filter_code(latent_code_i)
            with _disable_tracking_bn_stats(dec):
                return dec(z_s)
        return dec(enc.filter_code(latent_code_i))

    def predict_w_reconstructed_image(self, image):
        """model.py:603-606: segment the FTN's own reconstruction of the image."""
        return self.fast_predict(self.recon_image(image))[1]

    def requires_grad_(self, requires_grad=True):
        for module in self.model.values():
            for p in module.parameters():
                p.requires_grad = requires_grad

    def recon_image(self, image, disable_track_bn_stats=False):
        z_i, _ = self.encode_image(image, disable_track_bn_stats)
        return self.decode_image(z_i, disable_track_bn_stats)

    def run(self, input):
        zi, zs = self.encode_image(input)
        init_predict = self.model["segmentation_decoder"](zs)
        return self.decode_image(zi), init_predict, self.recon_shape(init_predict)

    def forward(self, input):
        return self.fast_predict(input)[1]

    def fast_predict(self, input, disable_track_bn_stats=False):
        """model.py:561-601 -> ((z_i, z_s), y_0)."""
        if not self.training:
            with torch.no_grad():
                z_i, z_s = self._enc(input)
                y_0 = self.model["segmentation_decoder"](z_s)
        else:
            z_i, z_s = self._enc(input, disable_track_bn_stats)
            y_0 = self._call(self.model["segmentation_decoder"], z_s, disable_track_bn_stats)
        return (z_i, z_s), y_0

    def decoder_inference(self, decoder, latent_code, eval=False, disable_track_bn_stats=False):
        """model.py:396-412."""
        state = decoder.training
        if eval:
            decoder.eval()
            with torch.no_grad():
                logit = decoder(latent_code)
        else:
            logit = self._call(decoder, latent_code, disable_track_bn_stats)
        decoder.train(mode=state)
        return logit

    # ------------------------------------------------------------------ losses of one step (model.py:414-467, 525-559)
    def standard_training(self, clean_image_l, label_l, perturbed_image, separate_training=False, compute_gt_recon=True,
                          update_latent=True, disable_track_bn_stats=False, _pre=None):
        """`_pre = (z_i, z_s, image_recon_loss)`: the FTN encoder pass and the image branch were already issued by the caller
        (cooperative_step runs the image branch and the whole hard-example branch on the second stream)."""
        zero = torch.zeros((), device=clean_image_l.device)
        if _pre is not None:
            z_i, z_s, image_recon_loss, y_0 = _pre
            if y_0 is None:
                y_0 = self._call(self.model["segmentation_decoder"], z_s, disable_track_bn_stats)
        elif self.two_streams and self.training and not self._in_side:
            z_i, z_s = self._enc(perturbed_image, disable_track_bn_stats)
            image_recon_loss = self._image_recon_loss(z_i, clean_image_l)      # side stream, next to D_seg -> STN below
            y_0 = self._call(self.model["segmentation_decoder"], z_s, disable_track_bn_stats)
        else:
            (z_i, z_s), y_0 = self.fast_predict(perturbed_image, disable_track_bn_stats=disable_track_bn_stats)
            image_recon_loss = None
        if update_latent:
            self.z_i, self.z_s = z_i, z_s
        standard_supervised_loss = basic_loss_fn(y_0, label_l.detach(), "cross entropy")
        if image_recon_loss is None and _pre is None:
            image_recon = self.decode_image(z_i)                       # always BN mode A, as upstream (model.py:444)
            image_recon_loss = scaled_mse(image_recon, clean_image_l, 0.5)
        y_0_new = y_0.detach() if separate_training else y_0
        if compute_gt_recon and self.group_stn_passes and not disable_track_bn_stats and self.training:
            # the two STN passes are independent and share the BatchNorm mode: one grouped pass (same order: gt, then p)
            gt_recon, p_recon = self.recon_shape_pair(label_l.detach(), True, y_0_new, False)
            gt_shape_recon_loss = basic_loss_fn(gt_recon, label_l, "cross entropy")
        else:
            if compute_gt_recon:
                gt_recon = self.recon_shape(label_l.detach(), is_label_map=True)
                gt_shape_recon_loss = basic_loss_fn(gt_recon, label_l, "cross entropy")
            else:
                gt_shape_recon_loss = zero
            p_recon = self.recon_shape(y_0_new, is_label_map=False, disable_track_bn_stats=disable_track_bn_stats)
        pred_shape_recon_loss = basic_loss_fn(p_recon, label_l, "cross entropy")
        self._join_side()
        return standard_supervised_loss, image_recon_loss, gt_shape_recon_loss, pred_shape_recon_loss

    def hard_example_training(self, perturbed_image, clean_image_l, perturbed_seg, label_l, separate_training=False, use_gpu=True):
        dev = clean_image_l.device
        zero = torch.zeros((), device=dev)
        seg_loss = recon_loss = shape_loss = perturbed_p_recon_loss = zero
        if perturbed_image is not None and perturbed_seg is not None and self.group_stn_passes and self.training:
            # FTN pass on the hard image, then both STN passes (prediction of the hard image, corrupted segmentation: independent,
            # both with frozen BatchNorm statistics tracking) as one grouped pass
            z_i, z_s = self._enc(perturbed_image.detach(), True)
            recon_loss = self._image_recon_loss(z_i, clean_image_l)              # (side stream when two_streams)
            y_0 = self._call(self.model["segmentation_decoder"], z_s, True)
            seg_loss = basic_loss_fn(y_0, label_l.detach(), "cross entropy")
            if separate_training:
                perturbed_seg = perturbed_seg.detach()
            p_recon, perturbed_p_recon = self.recon_shape_pair(y_0.detach() if separate_training else y_0, False, perturbed_seg, False,
                                                               disable_track_bn_stats=True)
            shape_loss = basic_loss_fn(p_recon, label_l, "cross entropy")
            perturbed_p_recon_loss = basic_loss_fn(perturbed_p_recon, label_l, "cross entropy")
            self._join_side()
            return seg_loss, recon_loss, shape_loss, perturbed_p_recon_loss
        if perturbed_image is not None:
            seg_loss, recon_loss, _, shape_loss = self.standard_training(
                clean_image_l=clean_image_l, label_l=label_l, perturbed_image=perturbed_image.detach(), compute_gt_recon=False,
                separate_training=separate_training, update_latent=False, disable_track_bn_stats=True)
        if perturbed_seg is not None:
            if separate_training:
                perturbed_seg = perturbed_seg.detach()
            perturbed_p_recon = self.recon_shape(perturbed_seg, is_label_map=False, disable_track_bn_stats=True)
            perturbed_p_recon_loss = basic_loss_fn(perturbed_p_recon, label_l, "cross entropy")
        return seg_loss, recon_loss, shape_loss, perturbed_p_recon_loss

    # ------------------------------------------------------------------ latent-space hard examples (model.py:300-350, 469-523)
    def perturb_latent_code(self, latent_code, decoder_function, label_y=None, perturb_type="random", threshold=0.5,
                            if_soft=False, random_threshold=False, loss_type="mse", if_detach=False, *, override=None,
                            _keep_mask=False):
        """model.py:300-350.  `mask`: [N,C,1,1] / [N,1,H,W] for the targeted schemes; for 'dropout' upstream's full-size mask
        (1 where the dropped-out code EQUALS the input element, model.py:334-336) -- `_keep_mask=True` (the solver's own calls,
        which only log the mask) returns the [N,C,1,1] keep pattern instead and skips that extra pass."""
        assert perturb_type in ["random", "dropout", "spatial", "channel"], "invalid method name"
        ov = override or {}
        perturb_type = ov.get("scheme", perturb_type)
        if perturb_type == "random":
            cands = ["dropout", "spatial", "channel"]
            random.shuffle(cands)
            perturb_type = cands[0]
        self.last_scheme = perturb_type
        self._step_schemes.append(perturb_type)      # (every scheme drawn in this step, see _cooperative_step)
        self._perturb_calls += 1
        salt, gstate = self._perturb_calls, self._gstate
        if perturb_type == "dropout":
            res = ops.dropout2d(latent_code.detach(), threshold, keep=ov.get("keep"), seed=salt if gstate is not None else _draw_seed(),
                                state=gstate, want_mask=not _keep_mask)
            masked, keep = res[0], res[1]
            keep4 = keep.view(keep.shape[0], keep.shape[1], 1, 1)
            mask = keep4 if _keep_mask else res[2]
            if not if_detach and latent_code.requires_grad:
                masked = latent_code * (keep4 / (1.0 - threshold))
        else:
            assert loss_type in ["mse", "ce", "corr"], "not implemented loss"
            fn = mask_latent_code_spatial_wise if perturb_type == "spatial" else mask_latent_code_channel_wise
            k, noise = ov.get("k"), ov.get("soft_noise")
            if gstate is not None:
                n, c, h, w = latent_code.shape
                L = c if perturb_type == "channel" else h * w
                if k is None and random_threshold:
                    k = self._gk[salt]                       # device int32: the host draws np.random per replay (graph.py)
                if if_soft and noise is None:
                    noise = ops.uniform((n, L), latent_code.device, seed=salt, state=gstate)
            masked, mask = fn(latent_code, num_classes=self.num_classes, decoder_function=decoder_function, label=label_y,
                              percentile=threshold, random=random_threshold, loss_type=loss_type, if_detach=if_detach,
                              if_soft=if_soft, k=k, soft_noise=noise)
        if if_detach:
            masked = masked.detach()
        return masked, mask

    def hard_example_generation(self, clean_image_l, label_l, gen_corrupted_seg=True, gen_corrupted_image=True,
                                corrupted_image_DA_config=None, corrupted_seg_DA_config=None, *, image_override=None,
                                seg_override=None):
        img_cfg = corrupted_image_DA_config or _DEFAULT_IMG_CFG
        seg_cfg = corrupted_seg_DA_config or _DEFAULT_SEG_CFG
        d_seg, d_img = self.model["segmentation_decoder"], self.model["image_decoder"]
        set_grad(d_seg, requires_grad=False)
        set_grad(d_img, requires_grad=False)
        perturbed_image_0 = perturbed_y_0 = None
        try:
            if gen_corrupted_image:
                self.reset_all_optimizers()
                z, m = self.perturb_latent_code(self.z_i, d_img, label_y=clean_image_l, perturb_type=img_cfg["mask_type"],
                                                loss_type=img_cfg["loss_name"], threshold=img_cfg["max_threshold"],
                                                random_threshold=img_cfg["random_threshold"], if_detach=True,
                                                if_soft=img_cfg["if_soft"], override=image_override, _keep_mask=True)
                self.last_masks["image"] = m
                zi_masked = z
                gen_img = lambda: self.decoder_inference(d_img, zi_masked, eval=False, disable_track_bn_stats=True)
                perturbed_image_0 = gen_img() if self._in_side else self._fork_side(gen_img, zi_masked)
            if gen_corrupted_seg:
                self.reset_all_optimizers()
                z, m = self.perturb_latent_code(self.z_s, d_seg, label_y=label_l, perturb_type=seg_cfg["mask_type"],
                                                loss_type=seg_cfg["loss_name"], threshold=seg_cfg["max_threshold"],
                                                random_threshold=seg_cfg["random_threshold"], if_detach=True,
                                                if_soft=seg_cfg["if_soft"], override=seg_override, _keep_mask=True)
                self.last_masks["seg"] = m
                perturbed_y_0 = self.decoder_inference(d_seg, z, eval=False, disable_track_bn_stats=True)
            self._join_side()
        finally:
            set_grad(d_seg, requires_grad=True)
            set_grad(d_img, requires_grad=True)
            # the standard pass' activations were kept for the saliency pass of a decoder whose code was perturbed above
            # (CtlNet.reuse_pass); nothing after this point may re-use them, and the record pins a whole activation arena per decoder
            # (289 MB at bs16 256x256).  The OTHER decoder's record stays: the two-chain step perturbs the two codes in two calls.
            if gen_corrupted_seg:
                d_seg.forget_pass()
            if gen_corrupted_image:
                d_img.forget_pass()
        return perturbed_image_0, perturbed_y_0

    # ------------------------------------------------------------------ inference (model.py:375-394, 608-664)
    def slow_refinement(self, pred_logit, n_steps=1, auto_stop=False, save_internal_predicts=False):
        """model.py:608-641.  Every pass of ONE call re-feeds the call's input (`pred_logit.detach().clone()`, :629): with eval-mode
        BatchNorm all passes give the same tensor, so one is run; with training-mode BatchNorm every pass moves the running
        statistics, so all `n_steps` run.  (`auto_stop` compares consecutive passes, which are equal: it never changes the result.)"""
        n_steps = self.n_iter if n_steps is None else n_steps
        s_t = pred_logit
        stn_training = self.model["shape_encoder"].training or self.model["shape_decoder"].training
        for _ in range(max(n_steps, 0) if stn_training else min(max(n_steps, 0), 1)):
            s_t = self.recon_shape(pred_logit.detach())
        return s_t, {0: [pred_logit]}

    def predict(self, input, softmax=False, n_iter=None):
        """model.py:375-394: FTN prediction, then n_iter-1 calls of slow_refinement, each fed with the previous call's result
        (n_iter = 3 composes two STN passes)."""
        self.eval()
        n_iter = self.n_iter if n_iter is None else n_iter
        with torch.no_grad():
            _, pred = self.fast_predict(input)
            for _ in range(max(n_iter - 1, 0)):
                pred, _ = self.slow_refinement(pred, n_steps=n_iter)
        if softmax:
            pred = softmax_t(pred, 1.0)
        return pred

    def evaluate(self, input, targets_npy, n_iter=None):
        """model.py:643-664: argmax on device (uint8), confusion matrix on host."""
        n_iter = self.n_iter if n_iter is None else n_iter
        self.train(if_testing=True)
        pred = self.predict(input, n_iter=n_iter)
        pred_lab = ops.argmax_c(pred)
        if torch.is_tensor(targets_npy) and targets_npy.is_cuda:
            # ground truth already on the device -> confusion matrix accumulated there, nothing is copied to
            # the host per batch (the cur_eval_* visualisation copies are made lazily by whoever asks for them)
            self.running_metric.update(label_trues=targets_npy, label_preds=pred_lab)
            self.cur_eval_images, self.cur_eval_predicts, self.cur_eval_gts = input.detach()[:, 0], pred_lab, targets_npy
            return pred
        pred_npy = pred_lab.cpu().numpy()
        self.running_metric.update(label_trues=targets_npy, label_preds=pred_npy)
        self.cur_eval_images = input.detach().cpu().numpy()[:, 0, :, :]
        self.cur_eval_predicts, self.cur_eval_gts = pred_npy, targets_npy
        return pred

    def get_recon_diff(self, input):
        self.eval()
        with torch.no_grad():
            (z_i, _), first = self.fast_predict(input)
            refined = self.recon_shape(first, is_label_map=False)
            recon = self.decode_image(z_i)
        return torch.abs(input - recon), torch.abs(refined - first), first, refined, recon

    # ------------------------------------------------------------------ one full iteration (train...py:171-237)
    def _two_chain_forward(self, clean_image_l, label_l, image_l, img_cfg, seg_cfg, separate_training, image_override, seg_override):
        """Forward of one iteration as two launch chains.  After the FTN encoder everything the hard-example branch needs exists
        (z_i, z_s): image decoder + its loss, hard-example generation and the whole hard-example training forward go to the
        second stream; D_seg -> STN of the standard phase stay on the main stream.  Same calls, same order per network (so the
        BatchNorm running statistics see the same sequence), same numbers -- the launch-latency-bound low-resolution layers of one
        chain fill the gaps of the other.  autograd runs every backward node on the stream of its forward, so the backward
        overlaps the same way.  Weights are packed before the fork (the pack kernels must not race between the streams)."""
        for net in self.model.values():
            net.ensure_packed()
        cur, side = torch.cuda.current_stream(), self._side
        self._main = cur
        z_i, z_s = self._enc(image_l)
        self.z_i, self.z_s = z_i, z_s
        if self._chain_events is not None:
            self._fork_event = torch.cuda.Event(enable_timing=True)
            self._fork_event.record(cur)
        side.wait_stream(cur)                       # fork point: right after the encoder
        for t in (z_i, z_s, clean_image_l, label_l):
            t.record_stream(side)
        # Assignment of the work to the two chains, from the measured timeline (tools/timeline.py): pieces with a backward are split
        # so that BOTH directions balance -- main: D_seg -> STN (standard) + image decoder (standard), then the FTN encoder's backward;
        # side: FTN on the hard image, its D_seg, image decoder and STN pair.  The forward-only generation passes go where the
        # forward has room: image part at the head of the side chain (the hard image is needed first), segmentation part at the
        # head of the main chain (its result is needed last, by the hard STN pair).  Python order = reference order (image
        # perturbation before segmentation perturbation: same RNG draws).
        gen_kw = dict(corrupted_image_DA_config=img_cfg, corrupted_seg_DA_config=seg_cfg, image_override=image_override,
                      seg_override=seg_override)
        xh = yh = None
        img_saliency_done = None
        # Targeted masks: the saliency forward of a code is the standard pass of that decoder over again (same code, same weights, training
        # mode; model_util._saliency_grad) -- issue the standard pass FIRST and the generator re-uses its activations (one decoder forward
        # less per code).  'random' may still draw dropout: then the early pass only changed the issue order.
        from . import model_util as _mu
        scheme_of = lambda cfg, ov: None if cfg is None else (ov or {}).get("scheme", cfg["mask_type"])
        early_img = _mu.REUSE_SALIENCY_FORWARD and scheme_of(img_cfg, image_override) not in (None, "dropout")
        early_seg = _mu.REUSE_SALIENCY_FORWARD and scheme_of(seg_cfg, seg_override) not in (None, "dropout")
        image_recon_loss = y_0 = None
        if early_img:
            image_recon_loss = scaled_mse(self.decode_image(z_i), clean_image_l, 0.5)
            self._img_std_done = torch.cuda.Event()
            self._img_std_done.record(cur)
            side.wait_event(self._img_std_done)         # the image saliency pass reads these activations, and replays this pass' running update
        if early_seg:
            y_0 = self._call(self.model["segmentation_decoder"], z_s)
        self._in_side = True
        try:
            with torch.cuda.stream(side):
                if img_cfg is not None:
                    xh, _ = self.hard_example_generation(clean_image_l.detach(), label_l.detach(), gen_corrupted_seg=False,
                                                         gen_corrupted_image=True, **gen_kw)
                    if self.last_scheme != "dropout" and not early_img:
                        # targeted masks: the saliency pass decoded z_i through the image decoder with TRACKING BatchNorm (upstream
                        # calls decoder_function(code) in train mode, model_util.py:214) -- a third writer of that network's running
                        # statistics, on this chain.  The standard pass on the main chain must not overlap it: it waits for this
                        # event (both passes decode the same z_i, so their two updates commute bit-exactly; only overlap is wrong).
                        img_saliency_done = torch.cuda.Event()
                        img_saliency_done.record(side)
        finally:
            self._in_side = False
        if seg_cfg is not None:
            self._in_side = True               # (no nested fork inside the generation)
            try:
                _, yh = self.hard_example_generation(clean_image_l.detach(), label_l.detach(), gen_corrupted_seg=True,
                                                     gen_corrupted_image=False, **gen_kw)
            finally:
                self._in_side = False
            yh.record_stream(side)
        seg_ready = torch.cuda.Event()
        seg_ready.record(cur)
        # the standard image decoder goes FIRST on the main chain: autograd replays a chain in reverse, and where two passes of a
        # network sit on different streams the later gradient has to wait for the earlier one at the accumulation -- as the first
        # backward node of the main chain it stalled 5 ms on the hard image decoder's backward, which comes late on the side chain
        if image_recon_loss is None:
            if img_saliency_done is not None:
                cur.wait_event(img_saliency_done)
            image_recon_loss = scaled_mse(self.decode_image(z_i), clean_image_l, 0.5)
            self._img_std_done = torch.cuda.Event()
            self._img_std_done.record(cur)           # the hard phase's image-decoder pass (side chain) waits for this one
        # CPU issue order (one Python thread feeds both streams): the side chain's long part is issued BEFORE the main chain's
        # D_seg -> STN, while the GPU is still busy with what both streams already have -- issued after it, the side stream sat
        # idle for ~1.5 ms waiting for the host (tools/timeline.py)
        self._in_side = True
        try:
            with torch.cuda.stream(side):
                side.wait_event(seg_ready)
                hard = self.hard_example_training(perturbed_image=xh, perturbed_seg=yh, clean_image_l=clean_image_l, label_l=label_l,
                                                  separate_training=separate_training)
        finally:
            self._in_side = False
        std = self.standard_training(clean_image_l, label_l, perturbed_image=image_l, separate_training=separate_training,
                                     _pre=(z_i, z_s, None, y_0))
        if self._chain_events is not None:          # tools/chain_timing.py: when does each chain finish its forward?
            ev = {k: torch.cuda.Event(enable_timing=True) for k in ("fork", "main_done", "side_done")}
            ev["fork"] = self._fork_event
            ev["main_done"].record(cur)
            ev["side_done"].record(side)
            self._chain_events.append(ev)
        if not self.split_backward:
            cur.wait_stream(side)           # (split_backward: the join comes behind the two backward sweeps, see _cooperative_step)
        self._img_std_done = None
        for t in tuple(hard):
            t.record_stream(cur)
        return (std[0], image_recon_loss, std[2], std[3]), hard

    def cooperative_step(self, clean_image_l, label_l, image_l, img_cfg=None, seg_cfg=None, latent_DA=True, separate_training=False,
                         image_override=None, seg_override=None, do_optim=True, grad_hook=None):
        """The loop body of `train_network`, without its ten `.item()` syncs / `empty_cache()` stalls.  Returns the
        8 loss tensors (device scalars): standard (seg, image, gt_shape, shape) + hard (seg, image, shape, perturbed)."""
        self.train()
        self._perturb_calls = 0
        self._step_schemes = []
        if self._gstate is not None:
            ops.step_tick(self._gstate)           # (graph capture) RNG counter and Adam step advance on the device
        self.reset_all_optimizers()
        for net in self.model.values():
            net._defer_grads, net._deferred, net._pending_bwd = self.defer_param_grads, [], 0
        if self.defer_param_grads and latent_DA:
            from .nets import PassStack
            for name, _ in self.stack_passes:
                if self.model[name].drop_p is None:
                    self.model[name]._stack = PassStack(self.model[name], 2)
        if self._dp is not None:
            self._dp.begin_step(grad_hook is not None)
        try:
            return self._cooperative_step(clean_image_l, label_l, image_l, img_cfg, seg_cfg, latent_DA, separate_training, image_override,
                                          seg_override, do_optim, grad_hook)
        finally:
            for net in self.model.values():
                net._defer_grads, net._deferred, net._stack = False, [], None
                net.forget_pass()

    def _has_stacks(self) -> bool:
        return any(m._stack is not None and m._stack.filled for m in self.model.values())

    def _backward(self, loss):
        # (retain_graph with stacks: a sweep passes through the nodes upstream of a stacked pass without a gradient for them -- the stacked
        #  pass hands its input gradient back later, in a second sweep from its inputs -- and must not release what they saved)
        loss.backward(retain_graph=self._has_stacks())
        self._backward_stacks(two_chains=self.two_streams)
        for net in self.model.values():             # the parked per-pass parameter gradients: one accumulation launch per network
            net.collect_deferred_grads()

    def _backward_stacks(self, two_chains=False):
        """The stacked backward sweeps (nets.PassStack) behind the autograd sweep(s) that delivered the gradients of the stacked passes'
        outputs: per network, in STACK_PASSES order, ONE launch chain over all its passes of this step, then autograd continues from the
        passes' inputs (which reaches the stacks of the networks upstream).  two_chains: the networks marked for the second chain are
        issued on the second stream; a stack waits for the events of the gradients it received, whichever stream they came from."""
        cur, side = torch.cuda.current_stream(), self._side
        used_side = False
        for name, chain in self.stack_passes:
            net = self.model[name]
            st = net._stack
            if st is None or st.done or not st.filled:
                continue
            # (under stream capture every stack is issued on the capture's origin stream: with a stack on the second chain hipStreamEndCapture
            #  crashed on this image -- tools/debug/r6_capture_bisect.py -- while the all-on-one-stream form captures and replays bit-exactly)
            on_side = two_chains and chain == 1 and not torch.cuda.is_current_stream_capturing()
            tail = None
            if two_chains and self.split_wgrad_tail and not torch.cuda.is_current_stream_capturing():
                tail = cur if on_side else side
            with torch.cuda.stream(side if on_side else cur):
                cont = net.backward_stack(st, tail_stream=tail)
                if cont:
                    torch.autograd.backward([r for r, _ in cont], [g for _, g in cont], retain_graph=True)
            used_side = used_side or on_side or tail is not None
        if used_side:
            cur.wait_stream(side)

    def _cooperative_step(self, clean_image_l, label_l, image_l, img_cfg, seg_cfg, latent_DA, separate_training, image_override,
                          seg_override, do_optim, grad_hook):
        if self.two_streams and latent_DA:
            self._ensure_chains_overlap()
            std, hard = self._two_chain_forward(clean_image_l, label_l, image_l, img_cfg, seg_cfg, separate_training, image_override,
                                                seg_override)
            # Under stream capture the two-sweep form only pays for the dropout scheme: with targeted masks the hipGraph replay of the
            # two-sweep step is 13 % SLOWER than that of the one-sweep step (21.4 vs 18.9 ms), while the live streams gain 1.3 % (17.77 vs
            # 18.00): the replay's overlap depends on the topology the capture produces (profiles/r3_split_backward_ab4.txt)
            targeted = any(t != "dropout" for t in self._step_schemes)
            # (with neither code perturbed -- img_cfg and seg_cfg both None -- hard_example_training returns four constant zeros: nothing
            # to sweep on the side chain, the one-sweep form below handles it.  The hard sum itself must NOT be formed here, on the main
            # stream: its add nodes would tie the hard sweep to the main chain, see below -- 18.8 instead of 16.5 ms)
            if self.split_backward and self.defer_param_grads and any(t.requires_grad for t in hard) and \
                    not (targeted and torch.cuda.is_current_stream_capturing()):
                # Each branch's backward is its own sweep on its own chain: the standard branch lives on the main chain alone (FTN encoder,
                # its two decoders, the standard STN pair) and its forward ends ~1.5 ms before the hard branch's does on the side chain
                # (tools/timeline.py), so its sweep starts right away.  The root gradient of the hard sweep is created ON the side
                # stream: created on the main stream, the sweep's first node waits for everything main has queued (measured: 20.4 vs
                # 17.4 ms).  Parameter gradients are parked per pass and summed in forward order either way: same bits.
                # fp32 17.41 -> 17.31 ms same-box.  bf16 (launch-rate bound): eager unchanged (10.61 vs 10.65), but the hipGraph replay of the
                # two-sweep step takes 13.25 ms instead of 11.45 -- the one-sweep form stays there (profiles/r3_split_backward_ab3.txt).
                self.reset_all_optimizers()
                keep = self._has_stacks()
                (std[0] + std[1] + std[3] + std[2]).backward(retain_graph=keep)
                with torch.cuda.stream(self._side):
                    (hard[0] + hard[1] + hard[2] + hard[3]).backward(retain_graph=keep)
                self._backward_stacks(two_chains=True)
                torch.cuda.current_stream().wait_stream(self._side)
                for net in self.model.values():
                    net.collect_deferred_grads()
            else:
                if self.split_backward:
                    torch.cuda.current_stream().wait_stream(self._side)     # the join _two_chain_forward left to the sweeps
                loss = (std[0] + std[1] + std[3] + std[2]) + (hard[0] + hard[1] + hard[2] + hard[3])
                self.reset_all_optimizers()
                self._backward(loss)
            if grad_hook is not None:
                grad_hook(self)
            if do_optim:
                self.optimize_all_params()
            return self._loss_record(std, hard)
        std = self.standard_training(clean_image_l, label_l, perturbed_image=image_l, separate_training=separate_training)
        loss = std[0] + std[1] + std[3] + std[2]
        zero = torch.zeros((), device=clean_image_l.device)
        hard = (zero, zero, zero, zero)
        if latent_DA:
            xh, yh = self.hard_example_generation(clean_image_l.detach(), label_l.detach(), gen_corrupted_seg=seg_cfg is not None,
                                                  gen_corrupted_image=img_cfg is not None, corrupted_image_DA_config=img_cfg,
                                                  corrupted_seg_DA_config=seg_cfg, image_override=image_override,
                                                  seg_override=seg_override)
            hard = self.hard_example_training(perturbed_image=xh, perturbed_seg=yh, clean_image_l=clean_image_l, label_l=label_l,
                                              separate_training=separate_training)
            loss = loss + (hard[0] + hard[1] + hard[2] + hard[3])
        self.reset_all_optimizers()
        self._backward(loss)
        if grad_hook is not None:
            grad_hook(self)          # data-parallel gradient all-reduce goes here
        if do_optim:
            self.optimize_all_params()
        return self._loss_record(std, hard)

    @staticmethod
    def _loss_record(std, hard):
        """The 8 losses as DETACHED device scalars: backward has run, and a caller that keeps the tuple until the next step (any training
        loop does) would otherwise keep the whole autograd graph of this step alive through their grad_fn -- and with it the activation
        arenas of every pass, so that the next step needs a second set (seen as 8 device allocations in bench.py's second timed step)."""
        return tuple(v.detach() for v in tuple(std) + tuple(hard))
