"""CPU ORACLE -- test infrastructure only, never the product path.

A torch-CPU (fp32) restatement of the reference's cooperative-training hot path,
written from the reference's *behaviour* (file:line cited per function, paths are
relative to the upstream repo cherise215/Cooperative_Training_and_Latent_Space_Data_Augmentation).

Who may import this file: ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` -- as the checker / the reported CPU baseline,
never as the thing that is shipped or measured as the product.

Parity pin: the reference holds no unit tests or golden vectors of its own
(SURVEY.md section 4), so this oracle is pinned against outputs of the reference
itself, imported in the build container by ``tools/gen_golden.py``; the resulting
fixtures live in ``tests/golden/`` and ``tests/test_oracle_golden.py`` replays them.

Everything here works on logical NCHW tensors exactly like the reference.  Module
attribute names mirror the reference so that its ``state_dict`` files load
unchanged (e.g. ``general_encoder.down1.conv.0.weight``).

Randomness is *injected*: every place where the reference draws from python
``random`` / ``numpy.random`` / torch RNG takes an optional override (``scheme``,
``k``, ``noise``, ``keep``) so that the HIP path and the oracle can be driven with
identical decisions.
"""
from __future__ import annotations

import math
import random as _pyrandom
from typing import Callable, Dict, Optional, Tuple

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

SLOPE = 0.2          # nn.LeakyReLU(0.2) everywhere: medseg/models/ebm/encoder_decoder.py:46,57,337,405,472
NET_NAMES = ("image_encoder", "segmentation_decoder", "shape_encoder", "shape_decoder", "image_decoder")


# --------------------------------------------------------------------------- networks
def _bn(c: int) -> nn.BatchNorm2d:
    return nn.BatchNorm2d(c)  # eps 1e-5, momentum 0.1, affine: model.py:93-106 passes norm=nn.BatchNorm2d


def _double_conv(cin: int, cout: int) -> nn.Sequential:
    """conv3x3-BN-LeakyReLU-conv3x3-BN (encoder_decoder.py:43-49, 322-328, 370-378)."""
    return nn.Sequential(
        nn.Conv2d(cin, cout, 3, padding=1, bias=True), _bn(cout), nn.LeakyReLU(SLOPE),
        nn.Conv2d(cout, cout, 3, padding=1, bias=True), _bn(cout))


def _init_direct_conv_children(mod: nn.Module) -> None:
    """`normal_init` applied to *direct* children only (encoder_decoder.py:13-16, 400-401, 442-443, 493-494)."""
    for child in mod._modules.values():
        if isinstance(child, (nn.Conv2d, nn.ConvTranspose2d)):
            child.weight.data.normal_(0.0, 0.02)
            child.bias.data.zero_()


def _block_dropout(mod: nn.Module, res_x: torch.Tensor) -> torch.Tensor:
    """`res_x = self.drop(res_x)` (encoder_decoder.py:58-66, 338-347): nn.Dropout2d in training mode.  `mod.keep` (a [n, c] tensor of 0/1,
    test hook) replaces the Bernoulli draw so that two implementations can be compared on the same pattern."""
    p = getattr(mod, "drop_p", None)
    if p is None or not mod.training:
        return res_x
    keep = getattr(mod, "keep", None)
    if keep is None:
        keep = (torch.rand(res_x.shape[0], res_x.shape[1]) >= p).to(res_x.dtype)
    mod.last_keep = keep
    return res_x * keep.to(res_x.dtype)[:, :, None, None] * (1.0 / (1.0 - p))


class DownBlock(nn.Module):
    """`res_convdown` (encoder_decoder.py:19-68): x'=conv3x3 s2; out=LReLU(conv1x1(x') + dconv(x'))."""

    def __init__(self, cin: int, cout: int):
        super().__init__()
        self.down = nn.Conv2d(cin, cin, 3, stride=2, padding=1, bias=True)
        self.conv = _double_conv(cin, cout)
        self.conv_input = nn.Conv2d(cin, cout, kernel_size=1, stride=1, padding=0, bias=True)
        self.last_act = nn.LeakyReLU(SLOPE)

    def forward(self, x):
        x = self.down(x)
        return _block_dropout(self, self.last_act(self.conv_input(x) + self.conv(x)))


class UpBlock(nn.Module):
    """`res_up_family` (encoder_decoder.py:285-348) for up_type in {'NN','Conv2'}."""

    def __init__(self, cin: int, cout: int, up_type: str):
        super().__init__()
        if up_type == "NN":
            self.up = nn.Sequential(nn.UpsamplingNearest2d(scale_factor=2))
        elif up_type == "Conv2":
            self.up = nn.ConvTranspose2d(cin, cin, kernel_size=2, stride=2)
        else:
            raise NotImplementedError(up_type)
        self.conv = _double_conv(cin, cout)
        self.conv_input = nn.Conv2d(cin, cout, kernel_size=1, stride=1, padding=0, bias=True)
        self.last_act = nn.LeakyReLU(SLOPE)

    def forward(self, x):
        x = self.up(x)
        return _block_dropout(self, self.last_act(self.conv_input(x) + self.conv(x)))


class Encoder(nn.Module):
    """`MyEncoder` (encoder_decoder.py:351-415) with feature_reduce=4 -> ladder 16-32-64-128-128."""

    def __init__(self, input_channel: int, feature_reduce: int = 4, act: Optional[nn.Module] = None):
        super().__init__()
        c = [64 // feature_reduce, 128 // feature_reduce, 256 // feature_reduce, 512 // feature_reduce]
        self.inc = _double_conv(input_channel, c[0])
        self.down1 = DownBlock(c[0], c[1])
        self.down2 = DownBlock(c[1], c[2])
        self.down3 = DownBlock(c[2], c[3])
        self.down4 = DownBlock(c[3], c[3])
        self.final_conv = nn.Sequential(nn.Conv2d(c[3], c[3], kernel_size=1, stride=1, padding=0), _bn(c[3]))
        self.act = act
        _init_direct_conv_children(self)

    def forward(self, x):
        x = F.leaky_relu(self.inc(x), negative_slope=SLOPE)
        x = self.down4(self.down3(self.down2(self.down1(x))))
        x = self.final_conv(x)
        return x if self.act is None else self.act(x)


class DualEncoder(nn.Module):
    """`Dual_Branch_Encoder` (encoder_decoder.py:456-503): z_i = Enc(x); z_s = code_decoupler(z_i)."""

    def __init__(self, input_channel: int, z1: int, z2: int, feature_reduce: int = 4):
        super().__init__()
        self.general_encoder = Encoder(input_channel, feature_reduce, act=nn.ReLU())
        self.code_decoupler = nn.Sequential(
            nn.Conv2d(z1, z2, 3, padding=1, bias=True), _bn(z2), nn.LeakyReLU(SLOPE),
            nn.Conv2d(z2, z2, 3, padding=1, bias=True), _bn(z2), nn.ReLU())
        _init_direct_conv_children(self)

    def filter_code(self, z):
        return self.code_decoupler(z)

    def forward(self, x):
        z_i = self.general_encoder(x)
        return z_i, self.filter_code(z_i)


class Decoder(nn.Module):
    """`MyDecoder` (encoder_decoder.py:418-453)."""

    def __init__(self, input_channel: int, output_channel: int, feature_reduce: int = 4,
                 up_type: str = "NN", last_act: Optional[nn.Module] = None):
        super().__init__()
        c = [256 // feature_reduce, 128 // feature_reduce, 64 // feature_reduce]
        self.up1 = UpBlock(input_channel, c[0], up_type)
        self.up2 = UpBlock(c[0], c[1], up_type)
        self.up3 = UpBlock(c[1], c[2], up_type)
        self.up4 = UpBlock(c[2], c[2], up_type)
        self.final_conv = nn.Conv2d(c[2], output_channel, kernel_size=1, stride=1, padding=0)
        self.last_act = last_act
        _init_direct_conv_children(self)

    def forward(self, x):
        x = self.final_conv(self.up4(self.up3(self.up2(self.up1(x)))))
        return x if self.last_act is None else self.last_act(x)


def set_dropout(net: nn.Module, p: Optional[float], patterns=None) -> None:
    """`encoder_dropout` / `decoder_dropout` of model.py:92-106 on one network: Dropout2d(p) behind every residual block; `patterns`
    (list of [n, c] tensors in forward order) injects the keep patterns."""
    blocks = [m for m in net.modules() if isinstance(m, (DownBlock, UpBlock))]
    for i, m in enumerate(blocks):
        m.drop_p = p
        m.keep = None if patterns is None else patterns[i]


def kaiming_init_(net: nn.Module) -> None:
    """`init_weights(net,'kaiming')` (init_weight.py:30-39,54-65): Conv2d kaiming fan_in, BN N(1,.02)/0.

    ConvTranspose2d is not an nn.Conv2d instance, so it keeps torch's default init -- same as upstream.
    """
    def fn(m):
        if isinstance(m, (nn.Conv2d, nn.Linear)):
            nn.init.kaiming_normal_(m.weight.data, a=0, mode="fan_in")
        elif isinstance(m, nn.BatchNorm2d):
            nn.init.normal_(m.weight.data, 1.0, 0.02)
            nn.init.constant_(m.bias.data, 0.0)
    net.apply(fn)


def build_networks(image_ch: int = 1, num_classes: int = 4, reduce_factor: int = 4,
                   init: bool = True) -> Dict[str, nn.Module]:
    """`get_network('FCN_16_standard')` (model.py:76-149): construction order and the order of the
    `init_model` calls are kept so that, for a given torch seed, the RNG stream -- and therefore every
    initial weight -- is identical to the reference's."""
    z = 512 // reduce_factor
    image_encoder = DualEncoder(image_ch, z, z, reduce_factor)
    segmentation_decoder = Decoder(z, num_classes, reduce_factor, up_type="NN")
    image_decoder = Decoder(z, image_ch, reduce_factor, up_type="Conv2", last_act=nn.Sigmoid())
    shape_encoder = Encoder(num_classes, reduce_factor, act=nn.ReLU())
    shape_decoder = Decoder(z, num_classes, reduce_factor, up_type="NN")
    if init:  # model.py:122-131 order
        for net in (image_encoder, shape_decoder, shape_encoder, segmentation_decoder, image_decoder):
            kaiming_init_(net)
    return {"image_encoder": image_encoder, "segmentation_decoder": segmentation_decoder,
            "shape_encoder": shape_encoder, "shape_decoder": shape_decoder, "image_decoder": image_decoder}


# --------------------------------------------------------------------------- bf16 rounding-point emulation (BASELINE config 3)
def rb16(t: torch.Tensor) -> torch.Tensor:
    """Round to bf16 (round-to-nearest-even), keep the dtype: what `v_cvt_pk_bf16_f32` does to an MFMA operand / a stored tensor."""
    return t.to(torch.bfloat16).to(t.dtype)


class bf16_rounding_points:
    """Context manager: run the oracle networks with the ROUNDING POINTS of the engine's bf16 path (config 3) and fp32 arithmetic
    everywhere else -- the checker for that path (tests/test_bf16_engine_gpu.py):
      * every convolution's input and weights are rounded to bf16 (the MFMA operands; the input AFTER the fp32 BatchNorm + LeakyReLU of
        its producer), products accumulate in fp32, bias / residual / activation in fp32;
      * a BatchNorm takes its batch statistics from the UNROUNDED conv output (the kernels reduce the fp32 accumulators) but normalises
        the tensor as STORED (bf16);
      * `conv3x3(nearest_up(x))` of the 'NN' up blocks runs as the engine runs it: four 2x2 phase convs whose weights are sums of the
        3x3 taps formed in fp32 and rounded ONCE;
      * Dropout2d behind a block (encoder_dropout / decoder_dropout) reads the block's output as stored (bf16), scales in fp32 and stores
        bf16 again (the second rounding happens at the next convolution's operand rounding);
      * network inputs / outputs stay fp32.
    Forward emulation (inference, losses) by default.  `backward=True`: every network pass -- forward AND backward -- runs through the
    explicit plan emulation of oracle/bf16_plan.py instead (the same forward rounding points, plus the rounding points of the engine's
    backward plans: stored gradient tensors, dgrad / wgrad operands, BatchNorm-backward sums); that is the checker for bf16 gradients."""

    def __init__(self, backward: bool = False):
        self.backward = backward

    def __enter__(self):
        if self.backward:
            from . import bf16_plan
            self._saved_nets = (Encoder.forward, DualEncoder.forward, Decoder.forward)
            fwd = lambda m, x: bf16_plan.net_apply(m, x)
            Encoder.forward = DualEncoder.forward = Decoder.forward = fwd
            return self
        self._saved = (nn.Conv2d.forward, nn.ConvTranspose2d.forward, nn.BatchNorm2d.forward, UpBlock.forward)
        conv_fwd, convt_fwd, bn_fwd, _ = self._saved

        def conv(m, x):
            return m._conv_forward(rb16(x), rb16(m.weight), m.bias)

        def convt(m, x):
            return F.conv_transpose2d(rb16(x), rb16(m.weight), m.bias, m.stride, m.padding, m.output_padding, m.groups, m.dilation)

        def bn(m, u):
            if not (m.training or not m.track_running_stats):           # eval mode: running statistics
                return F.batch_norm(rb16(u), m.running_mean, m.running_var, m.weight, m.bias, False, 0.0, m.eps)
            mean = u.mean((0, 2, 3))
            var = u.var((0, 2, 3), unbiased=False)
            if m.track_running_stats and m.running_mean is not None:
                with torch.no_grad():
                    cnt = u.numel() / u.shape[1]
                    m.running_mean.mul_(1 - m.momentum).add_(mean.detach() * m.momentum)
                    m.running_var.mul_(1 - m.momentum).add_(var.detach() * cnt / max(cnt - 1, 1) * m.momentum)
                    m.num_batches_tracked.add_(1)
            scale = m.weight / torch.sqrt(var + m.eps)
            return rb16(u) * scale.view(1, -1, 1, 1) + (m.bias - mean * scale).view(1, -1, 1, 1)

        def up_forward(m, x):
            if not isinstance(m.up, nn.Sequential):                      # 'Conv2' (ConvTranspose2d): no re-formulation
                x = m.up(x)
                return _block_dropout(m, m.last_act(m.conv_input(x) + m.conv(x)))
            c0 = m.conv[0]
            n, _, h, w = x.shape
            xp = F.pad(rb16(x), (1, 1, 1, 1))
            u = x.new_zeros(n, c0.out_channels, 2 * h, 2 * w)
            for a in range(2):
                for b in range(2):
                    k = c0.weight.new_zeros(c0.out_channels, c0.in_channels, 2, 2)
                    for kh in range(3):
                        for kw in range(3):
                            k[:, :, (a + kh + 1) // 2 - a, (b + kw + 1) // 2 - b] += c0.weight[:, :, kh, kw]
                    u[:, :, a::2, b::2] = F.conv2d(xp[:, :, a:a + h + 1, b:b + w + 1], rb16(k), c0.bias)
            main = u
            for layer in list(m.conv)[1:]:
                main = layer(main)
            return _block_dropout(m, m.last_act(m.conv_input(m.up(x)) + main))

        def block_dropout(mod, res_x):
            if getattr(mod, "drop_p", None) is None or not mod.training:
                return res_x
            return plain_dropout(mod, rb16(res_x))       # the block's output is STORED (rounded) before the Dropout2d launch reads it

        nn.Conv2d.forward, nn.ConvTranspose2d.forward, nn.BatchNorm2d.forward, UpBlock.forward = conv, convt, bn, up_forward
        g = globals()
        plain_dropout = self._saved_dropout = g["_block_dropout"]
        g["_block_dropout"] = block_dropout
        return self

    def __exit__(self, *exc):
        if self.backward:
            Encoder.forward, DualEncoder.forward, Decoder.forward = self._saved_nets
            return False
        nn.Conv2d.forward, nn.ConvTranspose2d.forward, nn.BatchNorm2d.forward, UpBlock.forward = self._saved
        globals()["_block_dropout"] = self._saved_dropout
        return False


# --------------------------------------------------------------------------- small functions
def set_grad(module: nn.Module, requires_grad: bool) -> None:
    """model_util.py:163-165 / basic_operations.py:82-84."""
    for p in module.parameters():
        p.requires_grad = requires_grad


def one_hot(label: torch.Tensor, num_classes: int, dtype=torch.float32) -> torch.Tensor:
    """model_util.py:168-177 and basic_operations.py:135-140: int64 [N,H,W] -> float [N,C,H,W].
    (`dtype` exists only so that the whole oracle can also be run in fp64 as an accuracy yardstick.)"""
    return F.one_hot(label.long(), num_classes).permute(0, 3, 1, 2).to(dtype)


def stn_input(seg: torch.Tensor, num_classes: int, is_label_map: bool, temperature: float = 2.0,
              dtype=torch.float32) -> torch.Tensor:
    """`construct_input` as called by `encode_shape` (basic_operations.py:110-158; model.py:233-246)."""
    if is_label_map:
        return one_hot(seg, num_classes, dtype)
    return torch.softmax(seg / temperature, dim=1)


def ce2d(logit: torch.Tensor, label: torch.Tensor) -> torch.Tensor:
    """`cross_entropy_2D` with a 3-D int64 target (custom_loss.py:706-740; twin model_util.py:104-115):
    sum over pixels of -log_softmax(logit)[label] divided by N*H*W."""
    n, c, h, w = logit.shape
    logp = F.log_softmax(logit, dim=1).permute(0, 2, 3, 1).reshape(-1, c)
    return F.nll_loss(logp, label.reshape(-1), reduction="sum") / float(n * h * w)


def half_mse(pred: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
    """0.5 * MSELoss(mean) (model.py:445-447)."""
    return 0.5 * torch.mean((pred - target) ** 2)


class bn_no_track:
    """`_disable_tracking_bn_stats` (model_util.py:414-451) -- 'mode B' of SURVEY 8a row 4:
    batch statistics are used, running buffers are not updated, gamma/beta do not receive gradients
    from this pass.  On exit requires_grad is restored to the saved *track* flag (i.e. True)."""

    def __init__(self, model: nn.Module):
        self.bns = [m for m in model.modules() if isinstance(m, nn.BatchNorm2d)]

    def __enter__(self):
        self.saved = [m.track_running_stats for m in self.bns]
        for m in self.bns:
            m.track_running_stats = False
            m.weight.requires_grad_(False)
            m.bias.requires_grad_(False)

    def __exit__(self, *exc):
        for m, s in zip(self.bns, self.saved):
            m.track_running_stats = s
            m.weight.requires_grad_(s)
            m.bias.requires_grad_(s)
        return False


def _run(net: nn.Module, x, no_track: bool):
    if no_track:
        with bn_no_track(net):
            return net(x)
    return net(x)


# --------------------------------------------------------------------------- latent masking
def _as_float(t: torch.Tensor) -> torch.Tensor:
    """`makeVariable(type='float')` (model_util.py:603-618) casts to fp32; fp64 inputs are kept for the fp64 yardstick."""
    return t if t.dtype == torch.float64 else t.float()


def saliency_grad(code: torch.Tensor, decoder: Callable, label: torch.Tensor, num_classes: int,
                  loss_type: str) -> torch.Tensor:
    """dL/dz of model_util.py:202-223 (same lines 263-283 for the spatial variant)."""
    code = _as_float(code.detach()).requires_grad_(True)
    gt = one_hot(label, num_classes, code.dtype) if label.dim() < code.dim() else label
    out = decoder(code)
    if loss_type == "mse":
        loss = torch.mean((out - gt) ** 2)
    elif loss_type == "ce":
        loss = ce2d(out, label)
    elif loss_type == "corr":
        loss = torch.mean(out * gt)
    else:
        raise NotImplementedError(loss_type)
    return torch.autograd.grad(loss, [code])[0]


def rank_select_mask(score: torch.Tensor, k: int, soft_noise: Optional[torch.Tensor]) -> torch.Tensor:
    """model_util.py:231-244 / 293-306.  score [N,L]; threshold = k-th largest (descending sort, index k);
    entries strictly greater are masked: 0 (hard) or 0.5*noise with noise~U[0,1) (soft); others 1."""
    thr = torch.sort(score, dim=1, descending=True)[0][:, k].view(-1, 1)
    hit = score > thr
    if soft_noise is not None:
        return torch.where(hit, 0.5 * soft_noise, torch.ones_like(score))
    return torch.where(hit, torch.zeros_like(score), torch.ones_like(score))


def mask_latent_code_channel_wise(latent_code, decoder_function, label, num_classes=2, percentile=1 / 3.0,
                                  random=False, loss_type="corr", if_detach=True, if_soft=False,
                                  k: Optional[int] = None, soft_noise: Optional[torch.Tensor] = None,
                                  return_aux: bool = False):
    """model_util.py:180-255.  `k` / `soft_noise` override the numpy / torch draws of lines 228-230, 239."""
    n, c = latent_code.shape[:2]
    code = _as_float(latent_code.detach())
    grad = saliency_grad(code, decoder_function, label, num_classes, loss_type)
    score = grad.view(n, c, -1).mean(dim=2)                                   # signed mean, not |grad|: :224-225
    if k is None:
        if random:
            percentile = np.random.rand() * percentile
        k = int(c * percentile)
    if if_soft and soft_noise is None:
        soft_noise = torch.rand_like(score)
    vec = rank_select_mask(score, k, soft_noise if if_soft else None)
    mask = vec.view(n, c, 1, 1)
    masked = (code if if_detach else latent_code) * mask
    if hasattr(decoder_function, "zero_grad"):
        decoder_function.zero_grad()
    if return_aux:
        return masked, mask, {"grad": grad, "score": score, "k": k}
    return masked, mask


def mask_latent_code_spatial_wise(latent_code, decoder_function, label, num_classes, percentile=1 / 3.0,
                                  random=False, loss_type="corr", if_detach=True, if_soft=False,
                                  k: Optional[int] = None, soft_noise: Optional[torch.Tensor] = None,
                                  return_aux: bool = False):
    """model_util.py:258-318: mean over channels, ranks the H*W positions, mask [N,1,H,W]."""
    n, c, h, w = latent_code.shape
    code = _as_float(latent_code.detach())
    grad = saliency_grad(code, decoder_function, label, num_classes, loss_type)
    score = grad.mean(dim=1).reshape(n, h * w)
    if k is None:
        if random:
            percentile = np.random.rand() * percentile
        k = int(h * w * percentile)
    if if_soft and soft_noise is None:
        soft_noise = torch.rand_like(score)
    vec = rank_select_mask(score, k, soft_noise if if_soft else None)
    mask = vec.view(n, 1, h, w)
    masked = (code if if_detach else latent_code) * mask
    if hasattr(decoder_function, "zero_grad"):
        decoder_function.zero_grad()
    if return_aux:
        return masked, mask, {"grad": grad, "score": score, "k": k}
    return masked, mask


def dropout2d_with_keep(z: torch.Tensor, p: float, keep: Optional[torch.Tensor]) -> Tuple[torch.Tensor, torch.Tensor]:
    """`F.dropout2d(z, p)` (model.py:332-336): per-(n,c) Bernoulli keep, survivors scaled by 1/(1-p).
    `keep` [N,C] in {0,1} overrides the torch draw.  Returns (masked, mask) with the reference's mask
    definition: 1 where masked == input else 0 (model.py:334-336)."""
    n, c = z.shape[:2]
    if keep is None:
        keep = (torch.rand(n, c) >= p)
    out = z * (keep.to(z.dtype).view(n, c, 1, 1) / (1.0 - p))
    mask = torch.where(out == z, torch.ones_like(out), torch.zeros_like(out))
    return out, mask


# --------------------------------------------------------------------------- solver
class OracleSolver:
    """Restatement of `AdvancedTripletReconSegmentationModel` (model.py:24-813), hot-path methods only."""

    def __init__(self, image_ch: int = 1, num_classes: int = 4, learning_rate: float = 1e-4, n_iter: int = 1,
                 state_dicts: Optional[Dict[str, dict]] = None, network_type: str = "FCN_16_standard"):
        self.network_type = network_type        # + "_share_code" / "_w_o_filter": the ablation variants of model.py:199-203
        self.num_classes = num_classes
        self.n_iter = n_iter
        self.learning_rate = learning_rate
        self.model = build_networks(image_ch, num_classes, init=state_dicts is None)
        if state_dicts is not None:
            for k, sd in state_dicts.items():
                self.model[k].load_state_dict(sd)
        # one Adam per module, torch defaults (model.py:774-781)
        self.optimizers = {k: torch.optim.Adam(m.parameters(), lr=learning_rate) for k, m in self.model.items()}
        self.z_i = self.z_s = None
        self.last_masks = {}
        self.last_scores = []
        self.dtype = torch.float32
        self.train()

    def double(self):
        """fp64 copy of the whole path: the accuracy yardstick for fp32 results (reference's and HIP's alike)."""
        for m in self.model.values():
            m.double()
        self.optimizers = {k: torch.optim.Adam(m.parameters(), lr=self.learning_rate) for k, m in self.model.items()}
        self.dtype = torch.float64
        return self

    # -- mode switches (model.py:740-752); `self.training` stays True upstream, grad suppression in
    #    predict comes from its own no_grad (model.py:382)
    def train(self):
        for m in self.model.values():
            m.train()
            set_grad(m, True)

    def eval(self):
        for m in self.model.values():
            m.eval()

    def reset_all_optimizers(self):
        for o in self.optimizers.values():
            o.zero_grad()

    def optimize_all_params(self):
        for o in self.optimizers.values():
            o.step()

    # -- building blocks
    def fast_predict(self, x, no_track=False):
        """model.py:561-601."""
        z_i, z_s = _run(self.model["image_encoder"], x, no_track)
        if "share_code" in self.network_type:         # model.py:199-203 (ablation study: one shared code)
            z_i = z_s
        elif "w_o_filter" in self.network_type:
            z_s = z_i
        y0 = _run(self.model["segmentation_decoder"], z_s, no_track)
        return (z_i, z_s), y0

    def recon_shape(self, seg, is_label_map=False, no_track=False):
        """model.py:262-269 -> 233-260."""
        inp = stn_input(seg, self.num_classes, is_label_map, dtype=self.dtype)
        code = _run(self.model["shape_encoder"], inp, no_track)
        return _run(self.model["shape_decoder"], code, no_track)

    def standard_training(self, clean, label, perturbed, separate_training=False, compute_gt_recon=True,
                          update_latent=True, no_track=False):
        """model.py:414-467.  Note `decode_image` is called WITHOUT the no-track flag (model.py:444)."""
        (z_i, z_s), y0 = self.fast_predict(perturbed, no_track)
        if update_latent:
            self.z_i, self.z_s = z_i, z_s
        l_seg = ce2d(y0, label)
        l_img = half_mse(self.model["image_decoder"](z_i), clean)
        if compute_gt_recon:
            l_gt = ce2d(self.recon_shape(label.detach().clone(), is_label_map=True), label)
        else:
            l_gt = torch.tensor(0.0)
        y0_in = y0.detach().clone() if separate_training else y0
        l_shape = ce2d(self.recon_shape(y0_in, False, no_track), label)
        return l_seg, l_img, l_gt, l_shape

    def perturb_latent_code(self, z, decoder, label_y, perturb_type, threshold, if_soft, random_threshold,
                            loss_type, override: Optional[dict] = None):
        """model.py:300-350 with if_detach=True (the only way hard_example_generation calls it)."""
        ov = override or {}
        scheme = ov.get("scheme", perturb_type)
        if scheme == "random":
            cands = ["dropout", "spatial", "channel"]
            _pyrandom.shuffle(cands)
            scheme = cands[0]
        if scheme == "dropout":
            out, mask = dropout2d_with_keep(z, threshold, ov.get("keep"))
        else:
            fn = mask_latent_code_spatial_wise if scheme == "spatial" else mask_latent_code_channel_wise
            out, mask, aux = fn(z, decoder, label_y, num_classes=self.num_classes, percentile=threshold,
                                random=random_threshold, loss_type=loss_type, if_detach=True, if_soft=if_soft,
                                k=ov.get("k"), soft_noise=ov.get("soft_noise"), return_aux=True)
            self.last_scores.append({"scheme": scheme, "score": aux["score"].detach().clone(), "k": aux["k"]})      # (tests: near-tie margins of the ranking)
        if ov.get("mask") is not None:          # a selection made elsewhere (the fp64 yardstick re-uses the fp32 run's masks: a near-tie
            mask = ov["mask"].to(z.dtype)       # in the ranking must not make the two runs train on different hard examples)
            out = z * mask
        return out.detach().clone(), mask

    def hard_example_generation(self, clean, label, gen_corrupted_seg=True, gen_corrupted_image=True,
                                corrupted_image_DA_config=None, corrupted_seg_DA_config=None,
                                image_override: Optional[dict] = None, seg_override: Optional[dict] = None):
        """model.py:469-523."""
        d_seg, d_img = self.model["segmentation_decoder"], self.model["image_decoder"]
        set_grad(d_seg, False)
        set_grad(d_img, False)
        x_hard = y_hard = None
        if gen_corrupted_image:
            c = corrupted_image_DA_config
            zt, m = self.perturb_latent_code(self.z_i, d_img, clean, c["mask_type"], c["max_threshold"], c["if_soft"],
                                             c["random_threshold"], c["loss_name"], image_override)
            self.last_masks["image"] = m
            x_hard = _run(d_img, zt, True)          # decoder_inference(eval=False, disable_track_bn_stats=True)
        if gen_corrupted_seg:
            c = corrupted_seg_DA_config
            zt, m = self.perturb_latent_code(self.z_s, d_seg, label, c["mask_type"], c["max_threshold"], c["if_soft"],
                                             c["random_threshold"], c["loss_name"], seg_override)
            self.last_masks["seg"] = m
            y_hard = _run(d_seg, zt, True)
        set_grad(d_seg, True)
        set_grad(d_img, True)
        return x_hard, y_hard

    def hard_example_training(self, x_hard, clean, y_hard, label, separate_training=False):
        """model.py:525-559."""
        zero = torch.tensor(0.0)
        l_seg = l_img = l_shape = l_pert = zero
        if x_hard is not None:
            l_seg, l_img, _, l_shape = self.standard_training(clean, label, x_hard.detach().clone(), separate_training,
                                                              compute_gt_recon=False, update_latent=False, no_track=True)
        if y_hard is not None:
            if separate_training:
                y_hard = y_hard.detach().clone()
            l_pert = ce2d(self.recon_shape(y_hard, False, True), label)
        return l_seg, l_img, l_shape, l_pert

    def cooperative_step(self, clean, label, noisy, img_cfg, seg_cfg, latent_DA=True,
                         image_override=None, seg_override=None, do_optim=True, separate_training=False):
        """One iteration of `train_network` (train_adv_supervised_segmentation_triplet.py:171-237), with the
        noisy input (`image_l`, :185-189) supplied by the caller.  Returns the 8 loss terms."""
        self.train()
        self.reset_all_optimizers()
        std = self.standard_training(clean, label, noisy, separate_training=separate_training)
        loss = std[0] + std[1] + std[3] + std[2]
        hard = (torch.tensor(0.0),) * 4
        if latent_DA:
            self.reset_all_optimizers()
            xh, yh = self.hard_example_generation(clean.detach().clone(), label.detach().clone(),
                                                  gen_corrupted_seg=seg_cfg is not None,
                                                  gen_corrupted_image=img_cfg is not None,
                                                  corrupted_image_DA_config=img_cfg, corrupted_seg_DA_config=seg_cfg,
                                                  image_override=image_override, seg_override=seg_override)
            hard = self.hard_example_training(xh, clean, yh, label, separate_training=separate_training)
            loss = loss + (hard[0] + hard[1] + hard[2] + hard[3])
        self.reset_all_optimizers()
        loss.backward()
        if do_optim:
            self.optimize_all_params()
        return tuple(float(v.detach()) if torch.is_tensor(v) else float(v) for v in tuple(std) + tuple(hard))

    def predict(self, x, softmax=False, n_iter=None):
        """model.py:375-394 with 608-641: `predict` calls `slow_refinement(pred, n_steps=n_iter)` n_iter-1 times, feeding each
        call's result to the next (:387-389); INSIDE a call every pass re-feeds that call's input (`pred_logit.detach().clone()`,
        :629), so with eval-mode BatchNorm a call equals one STN pass: n_iter = 1 + number of composed STN passes."""
        self.eval()
        n_iter = self.n_iter if n_iter is None else n_iter
        with torch.no_grad():
            _, pred = self.fast_predict(x)
            for _ in range(max(n_iter - 1, 0)):
                pred = self.recon_shape(pred.detach().clone())
        return torch.softmax(pred, dim=1) if softmax else pred


# --------------------------------------------------------------------------- metrics
def dice(result: np.ndarray, reference: np.ndarray) -> float:
    """medpy `dc` as vendored in medseg/common_utils/measure.py:52-99: 2|A&B| / (|A|+|B|), NaN if both empty."""
    a = np.asarray(result).astype(bool)
    b = np.asarray(reference).astype(bool)
    den = int(a.sum()) + int(b.sum())
    return float("nan") if den == 0 else 2.0 * int((a & b).sum()) / den


def confusion_hist(label_true: np.ndarray, label_pred: np.ndarray, n_class: int) -> np.ndarray:
    """`runningScore._fast_hist` (medseg/common_utils/metrics.py:18-23)."""
    lt, lp = label_true.reshape(-1), label_pred.reshape(-1)
    m = (lt >= 0) & (lt < n_class)
    return np.bincount(n_class * lt[m].astype(int) + lp[m], minlength=n_class ** 2).reshape(n_class, n_class)


# ---------------------------------------------------------------------------------------------- SURVEY 8(f) rows 1 and 3
def rescale_intensity(data: torch.Tensor, new_min: float = 0.0, new_max: float = 1.0, eps: float = 1e-20) -> torch.Tensor:
    """Min-max rescale per (n, c) plane (medseg/common_utils/basic_operations.py:232-245)."""
    bs, c, h, w = data.shape
    flat = data.reshape(bs * c, -1)
    old_max = flat.max(dim=1, keepdim=True).values
    old_min = flat.min(dim=1, keepdim=True).values
    return ((flat - old_min) / (old_max - old_min + eps) * (new_max - new_min) + new_min).reshape(bs, c, h, w)


def crop_or_pad(image: np.ndarray, crop_size, label: np.ndarray = None):
    """Centre crop / zero pad [n,h,w] arrays to crop_size (medseg/common_utils/basic_operations.py:173-220): per axis the source
    index is `dst + (size - new)//2` (floor division, negative when padding), zero outside the source."""
    def one(a):
        n, h, w = a.shape
        nh, nw = crop_size
        hs, ws = (h - nh) // 2, (w - nw) // 2
        out = np.zeros((n, nh, nw), dtype=a.dtype)
        ys, xs = np.arange(nh) + hs, np.arange(nw) + ws
        vy, vx = (ys >= 0) & (ys < h), (xs >= 0) & (xs < w)
        out[:, np.ix_(vy, vx)[0], np.ix_(vy, vx)[1]] = a[:, ys[vy]][:, :, xs[vx]]
        return out
    return one(image), (None if label is None else one(label))


def noise_clamp(clean: torch.Tensor, noise: torch.Tensor, lo: float = 0.0, hi: float = 1.0) -> torch.Tensor:
    """`clamp(clean + noise, 0, 1)` with noise = 0.05 * N(0,1) (medseg/train_adv_supervised_segmentation_triplet.py:185-187)."""
    return torch.clamp(clean + noise, lo, hi)


def running_scores(confusion: np.ndarray):
    """`runningScore.get_scores` (medseg/common_utils/metrics.py:33-54) from an accumulated confusion matrix."""
    h = np.asarray(confusion, dtype=np.float64)
    with np.errstate(divide="ignore", invalid="ignore"):
        acc = np.diag(h).sum() / h.sum()
        acc_cls = np.nanmean(np.diag(h) / h.sum(axis=1))
        iu = np.diag(h) / (h.sum(axis=1) + h.sum(axis=0) - np.diag(h))
        freq = h.sum(axis=1) / h.sum()
    return ({"Overall Acc: \t": acc, "Mean Acc : \t": acc_cls, "FreqW Acc : \t": (freq[freq > 0] * iu[freq > 0]).sum(),
             "Mean IoU : \t": np.nanmean(iu)}, dict(zip(range(h.shape[0]), iu)))


def dice_from_confusion(confusion: np.ndarray) -> np.ndarray:
    """Per-class Dice of one volume from its confusion matrix: 2*h_cc / (row_c + col_c) = `dc(pred == c, gt == c)` (measure.py:52-99)."""
    h = np.asarray(confusion, dtype=np.float64)
    with np.errstate(divide="ignore", invalid="ignore"):
        return 2.0 * np.diag(h) / (h.sum(axis=1) + h.sum(axis=0))


def patient_scores(preds: np.ndarray, gts: np.ndarray, classes, metrics=("Dice", "VolError", "VolSim"), foreground_only: bool = False):
    """One patient's row of `runningMySegmentationScore.update` (medseg/common_utils/metrics.py:185-252), mask by mask as the
    reference does it: per class c > 0 binarise both volumes, then Dice (medpy 0.4.0 `metric.binary.dc`: 2|A&B|/(|A|+|B|), 0.0 when
    both are empty), VolError (pred - gt) / gt and VolSim 1 - |v1 - v2| / |v1 + v2| (measure.py:668-722)."""
    row = []
    for c in classes:
        if c == 0:
            continue
        g = (gts > 0) if foreground_only else (gts == c)
        p = (preds > 0) if foreground_only else (preds == c)
        v1, v2, inter = int(np.count_nonzero(p)), int(np.count_nonzero(g)), int(np.count_nonzero(p & g))
        for m in metrics:
            if m == "Dice":
                row.append(2.0 * inter / float(v1 + v2) if v1 + v2 else 0.0)
            elif m == "VolError":
                with np.errstate(divide="ignore", invalid="ignore"):
                    row.append(float(np.float64(v1 - v2) / np.float64(1.0 * v2)))
            elif m == "VolSim":
                row.append(float(1 - np.abs(v1 - v2) / np.abs(float(v2 + v1))))
            else:
                raise ValueError(m)
    return row


def surface_scores(preds: np.ndarray, gts: np.ndarray, classes, spacing, foreground_only: bool = False):
    """Per class c > 0: [HD, ASD] as `runningMySegmentationScore.update` asks for them (metrics.py:224-236): HD = mean over
    the slices where both masks are non-empty of the in-plane symmetric Hausdorff distance with spacing[:2] (-1 if none),
    ASD = mean distance from the prediction's surface voxels to the ground truth's surface over the volume with the full
    spacing (1e100 if a mask is empty); surfaces = mask minus its erosion with the 8- / 26-neighbourhood; distances from the
    Euclidean distance transform of the other surface's complement (medpy 0.4.0 metric.binary, measure.py:1096-1128)."""
    from scipy import ndimage as ndi

    def sd(a, b, sp):
        st = ndi.generate_binary_structure(a.ndim, 2)
        ea, eb = a & ~ndi.binary_erosion(a, structure=st), b & ~ndi.binary_erosion(b, structure=st)
        return ndi.distance_transform_edt(~eb, sampling=sp)[ea]

    row = []
    for c in classes:
        if c == 0:
            continue
        g = (gts > 0) if foreground_only else (gts == c)
        p = (preds > 0) if foreground_only else (preds == c)
        per_slice = [max(sd(p[z], g[z], spacing[:2]).max(), sd(g[z], p[z], spacing[:2]).max())
                     for z in range(p.shape[0]) if p[z].any() and g[z].any()]
        row.append(float(sum(per_slice) / len(per_slice)) if per_slice else -1.0)
        row.append(float(sd(p, g, spacing).mean()) if p.any() and g.any() else 1e100)
    return row


def synthetic_batch(n: int, h: int, w: int, num_classes: int = 4, seed: int = 0, structured: bool = False):
    """SURVEY 8d synthetic inputs: U[0,1) images, randint labels (or a concentric-ellipse phantom),
    0.05*N(0,1) input noise clamped to [0,1] (train...py:185-187)."""
    g = torch.Generator().manual_seed(seed)
    clean = torch.rand(n, 1, h, w, generator=g)
    if structured:
        yy, xx = torch.meshgrid(torch.linspace(-1, 1, h), torch.linspace(-1, 1, w), indexing="ij")
        label = torch.zeros(n, h, w, dtype=torch.int64)
        for i in range(n):
            cx, cy = (torch.rand(2, generator=g) - 0.5) * 0.4
            r = torch.sqrt(((xx - cx) / 0.9) ** 2 + ((yy - cy) / 0.7) ** 2)
            label[i] = (r < 0.75).long() + (r < 0.5).long() + (r < 0.25).long()
        label = label % num_classes
    else:
        label = torch.randint(0, num_classes, (n, h, w), generator=g)
    noisy = torch.clamp(clean + 0.05 * torch.randn(n, 1, h, w, generator=g), 0, 1)
    return clean, label, noisy
