"""``FusedRMSprop``: ``torch.optim.RMSprop`` (the reference's optimiser, utils/training.py:52-53) whose step is ONE launch.

torch's foreach implementation issues 5 multi-tensor launches per step per ~30 tensors (a U-Net step: 10 launches, ~250 us of a 7 ms
step; the CNN: 5, ~85 us of 2 ms); ``nastar_rmsprop_multi_f32`` walks a device table of (param, grad, square_avg, count) rows instead.
Same state (``square_avg``, ``step``) and ``state_dict`` as torch's class, so checkpoints are interchangeable; anything the kernel does
not cover (momentum, centered, weight decay, maximize, CPU / non-fp32 / sparse tensors) takes torch's own step."""
from __future__ import annotations

import torch


class FusedRMSprop(torch.optim.RMSprop):
    _tables = None  # {id(group): (key, device table)}: one cached table per parameter group

    @torch.no_grad()
    def step(self, closure=None):
        # the closure first (Lightning's runs zero_grad + backward inside it): eligibility and row pointers are taken from the
        # gradients of THIS step, never from the previous step's
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        plain = all(g["momentum"] == 0 and not g["centered"] and g["weight_decay"] == 0 and not g.get("maximize", False)
                    and not g.get("differentiable", False) and not g.get("capturable", False) for g in self.param_groups)
        params = [p for g in self.param_groups for p in g["params"] if p.grad is not None]
        ok = plain and len(params) > 0 and all(
            p.is_cuda and p.dtype == torch.float32 and p.is_contiguous() and p.grad.is_cuda and p.grad.dtype == torch.float32
            and p.grad.is_contiguous() and not p.grad.is_sparse and p.grad.device == p.device for p in params)
        if ok and len({p.device for p in params}) != 1:
            ok = False
        if not ok:
            super().step(None)  # torch's own step on the gradients the closure (if any) just produced
            return loss
        from .. import _native
        lib = _native.load()
        dev = params[0].device
        stream = torch.cuda.current_stream(dev).cuda_stream
        if self._tables is None:
            self._tables = {}
        for group in self.param_groups:
            ps = [p for p in group["params"] if p.grad is not None]
            if not ps:
                continue
            rows = []
            for p in ps:
                st = self.state[p]
                if len(st) == 0:  # as torch.optim.RMSprop._init_group
                    st["step"] = torch.tensor(0.0)
                    st["square_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["step"] += 1
                rows.append((p.data_ptr(), p.grad.data_ptr(), st["square_avg"].data_ptr(), p.numel()))
            key = tuple(rows)
            cached = self._tables.get(id(group))
            if cached is None or cached[0] != key:  # the caching allocator usually hands the gradients the same blocks every step
                cached = (key, torch.tensor(rows, dtype=torch.int64).to(dev, non_blocking=True))
                self._tables[id(group)] = cached
            with torch.cuda.device(dev):
                rc = lib.nastar_rmsprop_multi_f32(cached[1].data_ptr(), len(rows), float(group["lr"]), float(group["alpha"]),
                                                  float(group["eps"]), stream)
            _native.check(rc, "nastar_rmsprop_multi_f32")
        return loss
