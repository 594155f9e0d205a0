"""Fused SGD-with-momentum over flat parameter groups (`torch.optim.SGD` semantics of
`src/upstream/delores_s/upstream_expert.py:236-243`: wd added to the gradient, first step buf = g, no dampening)."""
import torch

from src import _native as N


class HipSGD(torch.optim.Optimizer):
    def __init__(self, flat_groups, params, lr, momentum=0.9, weight_decay=0.0, grad_scale=1.0):
        super().__init__(params, dict(lr=lr, momentum=momentum, weight_decay=weight_decay))
        self.flat_groups = list(flat_groups)
        self.grad_scale = grad_scale
        self.grad_scale_tensor = None      # optional device scalar multiplied in (loss.backward(g))
        self.steps = 0

    @torch.no_grad()
    def step(self, closure=None):
        g0 = self.param_groups[0]
        for fg in self.flat_groups:
            first = fg.momentum is None
            if first:
                fg.momentum = torch.empty_like(fg.data)
            N.call("sgd_momentum", fg.data, fg.grad, fg.momentum, fg.numel, float(g0["lr"]), float(g0["momentum"]),
                   float(g0["weight_decay"]), int(first), float(self.grad_scale), self.grad_scale_tensor)
        self.steps += 1

    def zero_grad(self, set_to_none=True):
        for fg in self.flat_groups:
            for p in fg.params:
                p.grad = None

    def state_dict(self):
        return {"steps": self.steps, "momentum": [None if fg.momentum is None else fg.momentum.clone() for fg in self.flat_groups],
                "param_groups": [{k: v for k, v in g.items() if k != "params"} for g in self.param_groups]}

    def load_state_dict(self, sd):
        self.steps = sd["steps"]
        for fg, m in zip(self.flat_groups, sd["momentum"]):
            fg.momentum = None if m is None else m.to(fg.data.device).clone()
