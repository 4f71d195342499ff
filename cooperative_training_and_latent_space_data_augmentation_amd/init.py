"""Initial weights, drawn on the host with exactly the reference's RNG consumption.

The reference builds five torch modules and then runs `init_weights(net,'kaiming')` over them
(model.py:93-131,157-180; encoder_decoder.py:13-16,400-401,442-443,493-494; init_weight.py:30-39,54-65).
To make `torch.manual_seed(s)` give the same starting point as upstream, this module replays the *sequence of draws*:
default construction of every Conv2d / ConvTranspose2d (kaiming_uniform(a=sqrt 5) + uniform bias), `normal_init` on the
direct Conv children (only MyDecoder.final_conv qualifies), then the kaiming pass in the order of `init_model` calls.
torch.nn layers are used purely as RNG-faithful initialisers here -- no compute goes through them.
"""
from __future__ import annotations

from collections import OrderedDict
from typing import Dict, List, Tuple

import torch
import torch.nn as nn

# (key prefix, kind, args) in module-registration order; kind in {"conv","convT","bn"}
Spec = List[Tuple[str, str, tuple]]


def _double_conv(prefix: str, cin: int, cout: int) -> Spec:
    return [(f"{prefix}.0", "conv", (cin, cout, 3)), (f"{prefix}.1", "bn", (cout,)),
            (f"{prefix}.3", "conv", (cout, cout, 3)), (f"{prefix}.4", "bn", (cout,))]


def encoder_spec(prefix: str, cin: int, reduce: int = 4) -> Spec:
    c = [64 // reduce, 128 // reduce, 256 // reduce, 512 // reduce, 512 // reduce]
    p = prefix
    spec = _double_conv(f"{p}inc", cin, c[0])
    for i in range(4):
        spec += [(f"{p}down{i + 1}.down", "conv", (c[i], c[i], 3))]
        spec += _double_conv(f"{p}down{i + 1}.conv", c[i], c[i + 1])
        spec += [(f"{p}down{i + 1}.conv_input", "conv", (c[i], c[i + 1], 1))]
    spec += [(f"{p}final_conv.0", "conv", (c[4], c[4], 1)), (f"{p}final_conv.1", "bn", (c[4],))]
    return spec


def dual_encoder_spec(cin: int, z: int, reduce: int = 4) -> Spec:
    spec = encoder_spec("general_encoder.", cin, reduce)
    spec += [("code_decoupler.0", "conv", (z, z, 3)), ("code_decoupler.1", "bn", (z,)),
             ("code_decoupler.3", "conv", (z, z, 3)), ("code_decoupler.4", "bn", (z,))]
    return spec


def decoder_spec(cin: int, cout: int, up_type: str, reduce: int = 4) -> Spec:
    c = [cin, 256 // reduce, 128 // reduce, 64 // reduce, 64 // reduce]
    spec: Spec = []
    for i in range(4):
        if up_type == "Conv2":
            spec += [(f"up{i + 1}.up", "convT", (c[i], c[i], 2))]
        spec += _double_conv(f"up{i + 1}.conv", c[i], c[i + 1])
        spec += [(f"up{i + 1}.conv_input", "conv", (c[i], c[i + 1], 1))]
    spec += [("final_conv", "conv", (c[4], cout, 1))]
    return spec


def network_specs(image_ch: int = 1, num_classes: int = 4, reduce: int = 4) -> "OrderedDict[str, Spec]":
    z = 512 // reduce
    # construction order of get_network (model.py:93-106)
    return OrderedDict([
        ("image_encoder", dual_encoder_spec(image_ch, z, reduce)),
        ("segmentation_decoder", decoder_spec(z, num_classes, "NN", reduce)),
        ("image_decoder", decoder_spec(z, image_ch, "Conv2", reduce)),
        ("shape_encoder", encoder_spec("", num_classes, reduce)),
        ("shape_decoder", decoder_spec(z, num_classes, "NN", reduce)),
    ])


def _construct(spec: Spec) -> "OrderedDict[str, nn.Module]":
    mods: "OrderedDict[str, nn.Module]" = OrderedDict()
    for key, kind, a in spec:
        if kind == "conv":
            mods[key] = nn.Conv2d(a[0], a[1], a[2], padding=1 if a[2] == 3 else 0)
        elif kind == "convT":
            mods[key] = nn.ConvTranspose2d(a[0], a[1], kernel_size=2, stride=2)
        else:
            mods[key] = nn.BatchNorm2d(a[0])
    return mods


def reference_init_state_dicts(image_ch: int = 1, num_classes: int = 4, reduce: int = 4) -> Dict[str, "OrderedDict[str, torch.Tensor]"]:
    """State dicts (CPU) for the five networks, bit-identical to the reference's for the current torch seed."""
    specs = network_specs(image_ch, num_classes, reduce)
    built = OrderedDict()
    for name, spec in specs.items():
        mods = _construct(spec)
        # `normal_init` touches direct Conv children of the top module only: MyDecoder.final_conv
        if "final_conv" in mods and isinstance(mods["final_conv"], nn.Conv2d):
            mods["final_conv"].weight.data.normal_(0.0, 0.02)
            mods["final_conv"].bias.data.zero_()
        built[name] = mods
    for name in ("image_encoder", "shape_decoder", "shape_encoder", "segmentation_decoder", "image_decoder"):
        for m in built[name].values():   # net.apply visits leaves in registration order
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight.data, a=0, mode="fan_in")
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.normal_(m.weight.data, 1.0, 0.02)
                nn.init.constant_(m.bias.data, 0.0)
    out = {}
    for name, mods in built.items():
        sd = OrderedDict()
        for key, m in mods.items():
            for k, v in m.state_dict().items():
                sd[f"{key}.{k}"] = v.detach().clone()
        out[name] = sd
    return out
