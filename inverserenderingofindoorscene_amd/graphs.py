"""HIP-graph replay of a step of the path, for callers whose step is launch-bound.

At the reference's default batch (5, trainLight.py:28) the GPU work of layer + render loss is ~0.16 ms against ~0.14 ms of host
enqueue for the same eleven launches (profiles/r04u_host_overhead.txt): the step sits at the host bound.  Every operator of the
package is capturable -- workspaces come from torch's capture-aware allocator, the constant tables are built before the capture, the
render loss's last-arrival ticket is re-armed inside the captured launches, nothing synchronises or reads a device value on the host
(tests/test_gpu_graph.py) -- so such a caller can capture its step once and replay it.  From batch 16 up the step is GPU-bound and a
replay buys nothing (bench.py: config3.ms_per_step_config3_hipgraph).

    step = sgr.capture_step(lambda: train_step(static_batch))     # warm-up on a side stream, one capture
    for batch in loader:
        for k in static_batch: static_batch[k].copy_(batch[k])      # the captured launches read the SAME tensors
        outs = step()                                               # = step.replay(); the tensors the captured call returned, refilled

This is torch's own CUDA-graph contract (static input tensors, outputs alias the capture's), nothing of the package's making; the
helper only packages torch's recipe (side-stream warm-up, private pool kept alive with the graph) and builds the layer's constant
tables before the capture, where a first use would otherwise fail loudly.  An optimizer inside ``fn`` must be capturable
(``torch.optim.Adam(..., capturable=True)``)."""
from __future__ import annotations

from typing import Any, Callable

import torch

__all__ = ["CapturedStep", "capture_step"]


class CapturedStep:
    """One captured call of ``fn``; ``replay()`` (or calling the object) re-issues its launches and returns the captured outputs."""

    def __init__(self, fn: Callable[[], Any], warmup: int = 3, device=None):
        if not torch.cuda.is_available():
            raise RuntimeError("sgrender: capture_step needs a GPU (HIP graphs)")
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        cur = torch.cuda.current_stream(self.device)
        side = torch.cuda.Stream(device=self.device)
        side.wait_stream(cur)
        with torch.cuda.stream(side):          # first uses (table uploads, allocator growth, autograd engine start-up) happen here
            for _ in range(max(1, int(warmup))):
                fn()
        cur.wait_stream(side)
        torch.cuda.synchronize(self.device)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.outputs = fn()
        # the constant tables the captured launches read live in a bounded cache of the extension (64 view-vector tables, oldest evicted:
        # testReal.py builds a layer per image size); holding them here keeps their memory alive for as long as this graph can replay
        self._tables = list(torch.ops.sgrender.cached_tables())
        self.replays = 0

    def replay(self):
        self.graph.replay()
        self.replays += 1
        return self.outputs

    __call__ = replay


def capture_step(fn: Callable[[], Any], warmup: int = 3, device=None) -> CapturedStep:
    """Capture one call of ``fn`` -- a step built from this package's operators (and any other capturable work) on tensors that stay
    in place -- in a HIP graph.  See the module docstring for the contract."""
    return CapturedStep(fn, warmup=warmup, device=device)
