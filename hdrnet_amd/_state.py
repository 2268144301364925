"""Process-wide generation counter of PARAMETER STATE.

The models cache arrays derived from their parameters (batch norm folded into the guide network and the coefficient
network, the exported layouts the HIP kernels read) keyed on ``(data_ptr, _version)`` of every parameter and buffer.
Two writers of this package change parameters WITHOUT touching those version counters: ``optim.FlatAdam`` updates the
flat buffer through a raw pointer, and a replayed hipGraph (``runtime.GraphedTrainStep``: capturable optimizers, batch
norm's running statistics) writes from inside the graph.  Both call ``bump()``; every cache key and
``runtime.GraphedInference``'s staleness check include ``generation()``.
"""
_generation = 0


def bump() -> None:
    global _generation
    _generation += 1


def generation() -> int:
    return _generation
