"""Adam over flat buffers (torch.optim.Adam's update rule - the reference's optimizers, trainer_objective.py / pretrain_DAMSM.py:
Adam(lr, betas=(0.5, 0.999)) - as ONE launch per parameter group).

The trainers already keep every gradient of a network in one flat fp32 bucket (parallel.FlatGradBucket).  `FlatAdam` gives the
parameters and the two moments the same layout: each parameter's storage becomes a view of one flat buffer (`p.data` is re-pointed
once, at construction - module state_dicts, `load_state_dict`, `.copy_` and every kernel keep working on the views), so the update
is `tgsr::adam_flat_` over four dense buffers: 28 bytes per element in one pass.  torch's fused Adam walks the same tensors in ~10
multi-tensor launches per group (1.0 ms per G/D step over the discriminators' 100 M parameters; the flat pass is HBM-bound).

The step count lives on the device and is advanced by the kernel's own one-thread preamble, so a captured update keeps counting
across hipGraph replays; `state_dict()` / `load_state_dict()` speak torch.optim.Adam's format (per-parameter `step`, `exp_avg`,
`exp_avg_sq`), loading COPIES into the flat buffers - addresses a capture has baked in stay valid.
"""
import torch

from . import custom_ops as C


class FlatAdam(torch.optim.Optimizer):
    def __init__(self, params, grads_flat, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        """params: the parameters IN THE ORDER of `grads_flat` (a FlatGradBucket's `.params` / `.flat`): element k of the flat
        gradient belongs to element k of the flat parameter buffer built here."""
        params = list(params)
        if not params:
            raise ValueError("FlatAdam: no parameters")
        super().__init__(params, dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay))
        if len(self.param_groups) != 1:
            raise ValueError("FlatAdam takes one parameter group (one flat gradient buffer)")
        # FlatGradBucket's layout: every parameter on a multiple of 4 elements (the kernels want 16-byte aligned weights)
        n = sum((p.numel() + 3) & ~3 for p in params)
        if grads_flat.numel() != n or grads_flat.dtype != torch.float32 or not grads_flat.is_contiguous():
            raise ValueError("FlatAdam: the flat gradient holds %d elements, the parameters' layout %d" % (grads_flat.numel(), n))
        dev = grads_flat.device
        self.grads_flat = grads_flat
        self.flat = torch.zeros(n, dtype=torch.float32, device=dev)
        self.exp_avg = torch.zeros(n, dtype=torch.float32, device=dev)
        self.exp_avg_sq = torch.zeros(n, dtype=torch.float32, device=dev)
        self.counters = torch.tensor([0.0, 1.0, 1.0], dtype=torch.float32, device=dev)       # [step, 1 - b1^t, sqrt(1 - b2^t)]
        step_view = self.counters[0:1].view(())
        off = 0
        with torch.no_grad():
            for p in params:
                if p.dtype != torch.float32 or p.device != dev:
                    raise ValueError("FlatAdam: fp32 parameters on %s expected" % dev)
                k = p.numel()
                home = self.flat[off:off + k].view_as(p)
                home.copy_(p.data)
                p.data = home                                   # the parameter now LIVES in the flat buffer
                self.state[p] = {"step": step_view, "exp_avg": self.exp_avg[off:off + k].view_as(p),
                                 "exp_avg_sq": self.exp_avg_sq[off:off + k].view_as(p)}
                self._last_off = off
                off += (k + 3) & ~3

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        g = self.param_groups[0]
        # the parameters must still LIVE in the flat buffer (a module.to(...) or `p.data = ...` after construction moves them out, and the
        # update would run over memory nobody reads): the first and the last one are checked, two pointer reads per step
        for p, off in ((g["params"][0], 0), (g["params"][-1], self._last_off)):
            if p.data_ptr() != self.flat.data_ptr() + 4 * off:
                raise RuntimeError("FlatAdam: a parameter no longer lives in the optimizer's flat buffer (moved or re-assigned after "
                                   "the optimizer was built); build the optimizer after the modules are in place")
        C.adam_flat_(self.flat, self.grads_flat, self.exp_avg, self.exp_avg_sq, self.counters, float(g["lr"]), float(g["betas"][0]),
                     float(g["betas"][1]), float(g["eps"]), float(g["weight_decay"]), True)
        return loss

    def load_state_dict(self, state_dict):
        """torch.optim.Adam's format; the values are copied INTO the flat buffers (the views in `self.state` stay what they are)."""
        groups = state_dict["param_groups"]
        if len(groups) != 1 or len(groups[0]["params"]) != len(self.param_groups[0]["params"]):
            raise ValueError("FlatAdam.load_state_dict: another parameter list")
        for k, v in groups[0].items():
            if k != "params":
                self.param_groups[0][k] = v
        with torch.no_grad():
            step = None
            for idx, p in zip(groups[0]["params"], self.param_groups[0]["params"]):
                st = state_dict["state"].get(idx)
                if st is None:
                    continue
                self.state[p]["exp_avg"].copy_(st["exp_avg"])
                self.state[p]["exp_avg_sq"].copy_(st["exp_avg_sq"])
                step = float(st["step"])
            if step is not None:
                b1, b2 = self.param_groups[0]["betas"]
                self.counters.copy_(torch.tensor([step, 1.0 - b1 ** step if step else 1.0, (1.0 - b2 ** step) ** 0.5 if step else 1.0]))
