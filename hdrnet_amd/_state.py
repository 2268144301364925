"""Generation counters of PARAMETER STATE, scoped to the tensors that were actually written.

The models cache arrays derived from their parameters (batch norm folded into the guide network and the coefficient
network, the exported layouts the HIP kernels read) keyed on ``(data_ptr, _version)`` of every parameter and buffer.
Two writers of this package change parameters WITHOUT touching those version counters: ``optim.FlatAdam`` updates the
flat buffer through a raw pointer, and a replayed hipGraph (``runtime.GraphedTrainStep``: capturable optimizers, batch
norm's running statistics) writes from inside the graph.

Each such writer owns one ``Writer`` cell, attached once to the tensors it may write (an attribute on the tensor
object: it survives ``p.data = ...`` and ``module.to(...)``), and bumps it per step -- O(1).  A cache key or
``runtime.GraphedInference``'s staleness check reads ``generation_of(t)`` of ITS OWN tensors only, so a train step on
model A leaves the caches and captured graphs of an untouched model B (a frozen teacher, an EMA copy evaluated during
training) valid.  (Through round 5 the counter was process-wide: any optimizer step invalidated every model's
caches, and every ``GraphedInference`` in the process raised "parameters changed".)
"""
from typing import Iterable

_ATTR = "_hdrnet_writers"


class Writer:
    """One writer's generation cell (a FlatAdam instance, a GraphedTrainStep instance)."""

    __slots__ = ("n",)

    def __init__(self, tensors: Iterable = ()):
        self.n = 0
        self.attach(tensors)

    def attach(self, tensors: Iterable) -> None:
        for t in tensors:
            cells = getattr(t, _ATTR, None)
            if cells is None:
                setattr(t, _ATTR, [self])
            elif not any(c is self for c in cells):
                cells.append(self)

    def bump(self) -> None:
        self.n += 1


def generation_of(t) -> int:
    """Sum of the generations of the writers attached to tensor ``t`` (0: no raw writer ever touched it)."""
    cells = getattr(t, _ATTR, None)
    if not cells:
        return 0
    n = 0
    for c in cells:
        n += c.n
    return n
