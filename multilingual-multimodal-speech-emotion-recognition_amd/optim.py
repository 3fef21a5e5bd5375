"""Flat AdamW + the reference's warm-up/cosine schedule (ref src/train.py:72-83,114-121,169-177).

Same update rule as torch.optim.AdamW (checked against a torch trajectory fixture), but executed by
`ser_adamw` over contiguous segments of the modules' flat parameter buckets: ten launches for the
reference's ten parameter groups instead of one per tensor.  The per-step scalars (lr, bias
corrections) live in a small device tensor that is refreshed by `prepare_step()`, so the kernel
launches themselves (`launch()`) can sit inside a captured hipGraph.
"""
import math

import torch

from . import _ops as O
from .models._flat import find_bucket


class FlatAdamW:
    def __init__(self, param_groups, lr=1e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.05):
        self.base_lr = lr
        self.lr_factor = 1.0          # multiplied in by the scheduler
        self.betas, self.eps = betas, eps
        self.t = 0
        self.groups = []
        for grp in param_groups:
            params = [p for p in grp["params"]]
            self.groups.append(dict(params=params, lr_mult=grp.get("lr", lr) / lr, weight_decay=grp.get("weight_decay", weight_decay)))
        self.param_groups = [dict(lr=g["lr_mult"] * lr, weight_decay=g["weight_decay"]) for g in self.groups]
        self._plan = None
        self._state = {}              # id(bucket) or id(param) -> (m, v)
        self.hyper = None
        self._host = None
        self.gates = {}               # id(param) -> one-element int32 device tensor: non-zero = no update in this step (set_gates)

    # ---- planning: group -> contiguous segments of flat buckets (+ loose tensors) ------------------------------
    def _build_plan(self):
        plan = []
        for gi, grp in enumerate(self.groups):
            segs, loose = [], []
            for p in grp["params"]:
                if not p.requires_grad:
                    continue
                b, i = find_bucket(p)
                if b is None:
                    loose.append(p)
                    continue
                b.ensure()
                start, end = b.offsets[i], b.offsets[i] + b.padded_numel(i)
                if segs and segs[-1][0] is b and segs[-1][2] == start:
                    segs[-1][2] = end
                else:
                    segs.append([b, start, end])
            plan.append((grp, segs, loose))
        self._plan = plan

    def set_gates(self, gates):
        """gates: {id(param): one-element int32 device tensor}.  A gated parameter whose word is non-zero when the update kernel
        runs is left untouched, moments included - torch.optim.AdamW's treatment of a parameter whose gradient is None, for steps
        that cannot express "no gradient" by leaving the kernel out (captured graphs: LayerDrop, models/_finetune.py)."""
        self.gates = dict(gates)

    def _mv(self, key, like):
        if key not in self._state:
            self._state[key] = (torch.zeros_like(like), torch.zeros_like(like))
        return self._state[key]

    def prepare_step(self, device):
        """Host side of a step: advance t and upload {lr, 1-b1^t, sqrt(1-b2^t)}."""
        self.t += 1
        if self.hyper is None:
            self.hyper = torch.zeros(4, dtype=torch.float32, device=device)
            self._host = torch.zeros(4, dtype=torch.float32).pin_memory() if torch.cuda.is_available() else torch.zeros(4)
        b1, b2 = self.betas
        self._host[0] = self.base_lr * self.lr_factor
        self._host[1] = 1.0 - b1 ** self.t
        self._host[2] = math.sqrt(1.0 - b2 ** self.t)
        self.hyper.copy_(self._host, non_blocking=True)
        for pg, g in zip(self.param_groups, self.groups):
            pg["lr"] = self.base_lr * self.lr_factor * g["lr_mult"]

    def launch(self, only=None, skip=None):
        """Device side of a step (capturable).  only / skip: sets of id(parameter) among the parameters outside the flat buckets -
        `only` updates just those (the early update of an encoder whose backward is complete, TrainStepper), `skip` leaves them
        out of the regular launch."""
        if self._plan is None:
            self._build_plan()
        b1, b2 = self.betas
        segs_all = []                                   # (p, g, m, v, lr_mult, weight_decay) of every live segment
        for grp, segs, loose in self._plan:
            for b, s, e in segs:
                if only is not None:
                    break
                if b.params[0].grad is None and all(p.grad is None for p in b.params):
                    continue
                m, v = self._mv(id(b), b.flat)
                segs_all.append((b.flat[s:e], b.gflat[s:e], m[s:e], v[s:e], grp["lr_mult"], grp["weight_decay"]))
            for p in loose:
                if p.grad is None or (only is not None and id(p) not in only) or (skip is not None and id(p) in skip):
                    continue
                m, v = self._mv(id(p), p.data)
                segs_all.append((p.data, p.grad.contiguous(), m, v, grp["lr_mult"], grp["weight_decay"], self.gates.get(id(p))))
        if segs_all:
            O.adamw_multi_(segs_all, self.hyper, b1, b2, self.eps)

    def step(self):
        dev = None
        for g in self.groups:
            for p in g["params"]:
                if p.requires_grad:
                    dev = p.device
                    break
            if dev is not None:
                break
        self.prepare_step(dev)
        self.launch()

    def zero_grad(self, set_to_none=True):
        for g in self.groups:
            for p in g["params"]:
                if set_to_none:
                    p.grad = None
                elif p.grad is not None:
                    p.grad.zero_()

    def state_dict(self):
        if self._plan is None:
            self._build_plan()
        st = {}
        for gi, (grp, segs, loose) in enumerate(self._plan):
            for si, (b, s, e) in enumerate(segs):
                if id(b) in self._state:
                    m, v = self._state[id(b)]
                    st[f"g{gi}.s{si}"] = dict(m=m[s:e].clone(), v=v[s:e].clone())
            for li, p in enumerate(loose):
                if id(p) in self._state:
                    m, v = self._state[id(p)]
                    st[f"g{gi}.l{li}"] = dict(m=m.clone(), v=v.clone())
        return dict(t=self.t, base_lr=self.base_lr, lr_factor=self.lr_factor, state=st)

    # ---- torch.optim.AdamW format (what the reference's checkpoints hold, ref train.py:258) ---------------------------
    def _moments_of(self, p):
        """(m, v) views of parameter p inside this optimizer's state (allocating it when absent)."""
        if self._plan is None:
            self._build_plan()
        b, i = find_bucket(p)
        if b is not None:
            b.ensure()
            m, v = self._mv(id(b), b.flat)
            s = b.offsets[i]
            return m[s:s + p.numel()].view(p.shape), v[s:s + p.numel()].view(p.shape)
        return self._mv(id(p), p.data)

    def load_torch_state_dict(self, sd, order, params_by_name):
        """sd: torch.optim.AdamW.state_dict(); order: [(module key, name)] per optimizer index (SERSystem.torch_param_order
        of the checkpoint); params_by_name: {(module key, name): parameter} of this system.  Entries of parameters this
        system does not hold are ignored."""
        steps = []
        for idx, st in sd["state"].items():
            key = order[int(idx)]
            p = params_by_name.get(key)
            if p is None or not p.requires_grad:
                continue
            m, v = self._moments_of(p)
            m.copy_(st["exp_avg"].to(m.device))
            v.copy_(st["exp_avg_sq"].to(v.device))
            steps.append(int(float(st["step"])))
        self.t = max(steps) if steps else 0
        g0 = sd["param_groups"][2]                       # the `cross` group runs at the base learning rate (multiplier 1)
        self.base_lr = float(g0.get("initial_lr", g0["lr"]))
        if "initial_lr" in g0 and g0["initial_lr"] > 0:
            self.lr_factor = float(g0["lr"]) / float(g0["initial_lr"])

    def torch_state_dict(self, order, params_by_name):
        """This optimizer's state as a torch.optim.AdamW state dict over the parameters in `order`."""
        index = {k: i for i, k in enumerate(order)}
        state = {}
        for key, p in params_by_name.items():
            if key not in index or not p.requires_grad:
                continue
            b, _ = find_bucket(p)
            if (id(b) if b is not None else id(p)) not in self._state:
                continue
            m, v = self._moments_of(p)
            state[index[key]] = dict(step=torch.tensor(float(self.t)), exp_avg=m.detach().clone(), exp_avg_sq=v.detach().clone())
        groups, pos = [], 0
        for g, names in zip(self.groups, self._group_sizes(order)):
            lr = self.base_lr * self.lr_factor * g["lr_mult"]
            groups.append(dict(lr=lr, initial_lr=self.base_lr * g["lr_mult"], betas=self.betas, eps=self.eps, weight_decay=g["weight_decay"],
                               amsgrad=False, maximize=False, foreach=None, capturable=False, differentiable=False, fused=None,
                               decoupled_weight_decay=True, params=list(range(pos, pos + names))))
            pos += names
        return dict(state=state, param_groups=groups)

    def _group_sizes(self, order):
        from .system import SERSystem
        sizes = []
        for key, prefix in SERSystem.OPT_GROUPS:
            sizes.append(sum(1 for k, n in order if k == key and n.startswith(prefix)
                             and (prefix or key != "classifier")))
        return sizes

    def load_state_dict(self, sd, order=None, params_by_name=None):
        if "param_groups" in sd:                              # torch.optim.AdamW format (a reference checkpoint)
            assert order is not None and params_by_name is not None, "a torch-format optimizer state needs the parameter order"
            return self.load_torch_state_dict(sd, order, params_by_name)
        self.t, self.base_lr, self.lr_factor = sd["t"], sd["base_lr"], sd["lr_factor"]
        if self._plan is None:
            self._build_plan()
        for gi, (grp, segs, loose) in enumerate(self._plan):
            for si, (b, s, e) in enumerate(segs):
                ent = sd["state"].get(f"g{gi}.s{si}")
                if ent is not None:
                    m, v = self._mv(id(b), b.flat)
                    m[s:e].copy_(ent["m"]); v[s:e].copy_(ent["v"])
            for li, p in enumerate(loose):
                ent = sd["state"].get(f"g{gi}.l{li}")
                if ent is not None:
                    m, v = self._mv(id(p), p.data)
                    m.copy_(ent["m"]); v.copy_(ent["v"])


class WarmupCosine:
    """LambdaLR of ref train.py:114-121: linear warm-up then cosine to zero; like torch's LambdaLR the factor is
    applied at construction, so the first optimizer step runs with lambda(0)."""

    def __init__(self, optimizer, total_steps, warmup_ratio):
        self.opt, self.total, self.W = optimizer, total_steps, int(total_steps * warmup_ratio)
        self.last_epoch = 0
        self.opt.lr_factor = self.factor(0)

    def factor(self, step):
        if step < self.W:
            return float(step) / max(1, self.W)
        prog = (step - self.W) / max(1, self.total - self.W)
        return 0.5 * (1.0 + torch.cos(torch.tensor(prog * 3.1415926535)).item())

    def step(self):
        self.last_epoch += 1
        self.opt.lr_factor = self.factor(self.last_epoch)

    def state_dict(self):
        return dict(last_epoch=self.last_epoch, total=self.total, W=self.W)

    def load_state_dict(self, sd):
        """Own format, or torch's LambdaLR state dict (a reference checkpoint: only the step counter carries over; the
        schedule's length comes from this run's arguments, as in the reference where the lambda is rebuilt from args)."""
        self.last_epoch = sd["last_epoch"]
        if "total" in sd:
            self.total, self.W = sd["total"], sd["W"]
        self.opt.lr_factor = self.factor(self.last_epoch)
