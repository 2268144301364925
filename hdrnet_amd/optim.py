"""The optimizer update of the reference's training loop (``tf.train.AdamOptimizer``, hdrnet/bin/train.py:108-115) over ONE
flat parameter buffer.

``torch.optim.Adam(fused=True)`` walks the model's 35 parameter tensors with a multi-tensor launch whose few workgroups
each grind through 64 k elements: ~40 us of a 0.7-ms training step on MI355X for 2 MB of state.  ``FlatAdam`` re-binds
every parameter's storage to a view of one flat fp32 buffer laid out exactly like the step's flat gradient bucket
(``dist.GradBucket``, 16-byte aligned segments), so that the update is one elementwise kernel over the buffer
(``hdrnet_adam_step_f32``, include/hdrnet_amd_train.h) -- the same arithmetic, element by element, as
``torch.optim.Adam`` without amsgrad / weight decay.  The step count lives on the device: a captured hipGraph replays it.
``epsilon_hat=True`` places epsilon where ``tf.train.AdamOptimizer`` does (``lr sqrt(1 - b2^t) / (1 - b1^t) * m / (sqrt(v) +
eps)``, tensorflow/python/training/adam.py) -- the reference's optimizer to the letter; torch's form is the default.

The device-side step count is a float32 (the kernel's ABI, include/hdrnet_amd_train.h): it stops incrementing at 2^24 =
16.7 M steps.  By then both bias corrections ``1 - beta^t`` have been exactly 1.0 in float32 for millions of steps (beta2 =
0.999: t > ~17 000), so the update is unaffected; only ``state_dict()["steps"]`` saturates.

Construct it AFTER the module is on its device and in its memory format (``module.to(...)`` afterwards would re-allocate the
parameters and detach them from the flat buffer).
"""
from __future__ import annotations

from typing import Iterable, Tuple

import torch

from . import _state
from . import dist as hd

__all__ = ["FlatAdam"]


class FlatAdam:
    def __init__(self, params: Iterable[torch.nn.Parameter], lr: float = 1e-3, betas: Tuple[float, float] = (0.9, 0.999),
                 eps: float = 1e-8, epsilon_hat: bool = False):
        self.epsilon_hat = bool(epsilon_hat)
        self.bucket = hd.GradBucket(params, align=4)  # runtime.TrainStep picks this bucket up instead of making its own
        self.lr, self.betas, self.eps = float(lr), (float(betas[0]), float(betas[1])), float(eps)
        b = self.bucket
        self.flat = torch.zeros_like(b.flat)
        with torch.no_grad():
            for p, off, gview in zip(b.params, b.offsets, b.views):
                seg = self.flat[off:off + p.numel()]
                view = seg.as_strided(p.shape, gview.stride())  # the gradient view's strides = the parameter's own
                view.copy_(p.data)
                p.data = view
        self._writer = _state.Writer(b.params)  # this optimizer's generation cell on the parameters it updates
        self.exp_avg = torch.zeros_like(self.flat)
        self.exp_avg_sq = torch.zeros_like(self.flat)
        self.steps = torch.zeros((1,), dtype=torch.float32, device=self.flat.device)

    def state_dict(self) -> dict:
        """The Adam slots and the step count (the reference's checkpoints carry them: tf.train.Saver over the optimizer's
        variables, hdrnet/bin/train.py:140-150) + the hyper-parameters; the parameters themselves are the module's."""
        return {"exp_avg": self.exp_avg.clone(), "exp_avg_sq": self.exp_avg_sq.clone(), "steps": self.steps.clone(),
                "lr": self.lr, "betas": self.betas, "eps": self.eps, "epsilon_hat": self.epsilon_hat,
                "numel": int(self.flat.numel())}

    def load_state_dict(self, state: dict, hyperparameters: bool = True) -> None:
        """Restore the Adam slots and the step count.  ``hyperparameters=True`` (the default, what ``torch.optim``'s
        ``load_state_dict`` does with its param groups and what a resumed ``hdrnet/bin/train.py`` run gets from its
        checkpoint): lr, betas, eps and the epsilon form of the CHECKPOINT replace the constructor's; ``False`` keeps the
        constructor's -- resuming with a new learning rate."""
        if int(state["numel"]) != int(self.flat.numel()):
            raise ValueError(f"optimizer state of {state['numel']} elements does not fit {self.flat.numel()}")
        with torch.no_grad():
            self.exp_avg.copy_(state["exp_avg"])
            self.exp_avg_sq.copy_(state["exp_avg_sq"])
            self.steps.copy_(state["steps"])
        if hyperparameters:
            self.lr, self.betas, self.eps = float(state["lr"]), tuple(float(b) for b in state["betas"]), float(state["eps"])
            self.epsilon_hat = bool(state["epsilon_hat"])

    @property
    def param_groups(self):  # enough of torch.optim's surface for code that reads the learning rate
        return [{"params": self.bucket.params, "lr": self.lr, "betas": self.betas, "eps": self.eps}]

    def zero_grad(self, set_to_none: bool = False) -> None:
        self.bucket.zero_()  # never drops the views

    @torch.no_grad()
    def step(self) -> None:
        """One update from the bucket's flat gradient (call ``bucket.gather()`` first if backward ran on released
        gradients -- ``runtime.TrainStep`` does)."""
        g = self.bucket.flat
        b1, b2 = self.betas
        b = self.bucket
        if any(p.data_ptr() != self.flat.data_ptr() + off * self.flat.element_size() for p, off in zip(b.params, b.offsets)):
            raise RuntimeError("FlatAdam: a parameter no longer lives in the flat buffer (module.to(...) or a re-allocated "
                               "parameter after construction): the update would not reach it")
        self._writer.bump()  # the update below does not touch the parameters' version counters: invalidate THEIR caches
        if self.flat.is_cuda:
            from . import _lib
            from .hdrnet_ops import _stream
            lib = _lib.load()
            with torch.cuda.device(self.flat.device):
                fn = lib.hdrnet_adam_step_tf_f32 if self.epsilon_hat else lib.hdrnet_adam_step_f32
                rc = fn(self.flat.data_ptr(), g.data_ptr(), self.exp_avg.data_ptr(),
                        self.exp_avg_sq.data_ptr(), self.flat.numel(), self.steps.data_ptr(),
                        self.lr, b1, b2, self.eps, _stream(self.flat.device))
            if rc != 0:
                raise RuntimeError(f"hdrnet_adam_step{'_tf' if self.epsilon_hat else ''}_f32 failed (rc={rc})")
            return
        # CPU (the gloo tests): the same formula with torch ops
        self.steps += 1
        t = float(self.steps)
        self.exp_avg.mul_(b1).add_(g, alpha=1 - b1)
        self.exp_avg_sq.mul_(b2).addcmul_(g, g, value=1 - b2)
        rs2 = 1.0 / (1 - b2 ** t) ** 0.5
        denom = (self.exp_avg_sq.sqrt() * rs2).add_(self.eps * rs2 if self.epsilon_hat else self.eps)
        self.flat.addcdiv_(self.exp_avg, denom, value=-self.lr / (1 - b1 ** t))
