"""Flat multi-tensor Adam: ONE kernel launch per network updates every parameter (the reference uses one
`optim.Adam(model.parameters(), lr)` per network: model.py:774-785).  State dicts are laid out like torch.optim.Adam's
(`state` keyed by parameter index with `step`, `exp_avg`, `exp_avg_sq`; one param group) so upstream `_optim.pth` /
snapshot files round-trip."""
from __future__ import annotations

import torch

from . import ops


class FlatAdam:
    def __init__(self, net, lr=1e-4, betas=(0.9, 0.999), eps=1e-8):
        self.net = net
        self.defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=0, amsgrad=False)
        self.param_groups = [dict(self.defaults, params=list(net.parameters()))]
        self.exp_avg = torch.zeros_like(net._flat_data)
        self.exp_avg_sq = torch.zeros_like(net._flat_data)
        self.step_count = 0

    def zero_grad(self, set_to_none: bool = False):
        self.net.zero_grad()                 # gradients are views of one flat buffer that is never re-allocated

    def step(self, closure=None, grad_scale: float = 1.0, state=None):
        """`state`: device int64[3] whose [2] holds the (already advanced) step count -- HIP-graph capture, where the host-side
        count must not be a launch argument; the host count is still advanced as the mirror that state dicts report.
        Like torch.optim.Adam (which skips parameters whose `.grad` is None) the step is skipped when no backward pass has written
        this network's gradient since the last zero_grad() (set_to_none semantics of torch >= 2.0)."""
        if not self.net._grad_written:
            return
        g = self.param_groups[0]
        self.step_count += 1
        ops.adam_step(self.net._flat_data, self.net._flat.grad, self.exp_avg, self.exp_avg_sq, g["lr"], g["betas"][0],
                      g["betas"][1], g["eps"], self.step_count, grad_scale, state=state)
        self.net.weights_changed()

    def _views(self, flat):
        net = self.net
        return [flat[net._poff[n]:net._poff[n] + p.numel()].view(p.shape) for n, p in net.named_parameters()]

    def state_dict(self):
        state = {}
        if self.step_count > 0:
            for i, (m, v) in enumerate(zip(self._views(self.exp_avg), self._views(self.exp_avg_sq))):
                state[i] = {"step": torch.tensor(float(self.step_count)), "exp_avg": m.detach().cpu().clone(),
                            "exp_avg_sq": v.detach().cpu().clone()}
        g = {k: v for k, v in self.param_groups[0].items() if k != "params"}
        g["params"] = list(range(len(self.param_groups[0]["params"])))
        return {"state": state, "param_groups": [g]}

    def load_state_dict(self, sd):
        g = sd["param_groups"][0]
        for k in ("lr", "betas", "eps"):
            if k in g:
                self.param_groups[0][k] = tuple(g[k]) if k == "betas" else g[k]
        ms, vs = self._views(self.exp_avg), self._views(self.exp_avg_sq)
        self.step_count = 0
        for i, st in sd["state"].items():
            ms[int(i)].copy_(st["exp_avg"])
            vs[int(i)].copy_(st["exp_avg_sq"])
            self.step_count = int(float(st["step"]))
