"""hipGraph capture/replay of a whole decoder forward (launch-bound: 25-35 kernels of 3-120 us each).

One graph per input signature (pointers, shapes) and parameter version; replay re-executes every
kernel on the live input buffers — nothing is cached but the launch sequence.  Capture uses PyTorch's
graph support only as the owner of the capture stream and of the private memory pool."""
import torch

from . import ops as _ops


class GraphCache:
    def __init__(self, max_entries=8):
        self._entries = {}
        self._max = max_entries

    def clear(self):
        self._entries.clear()

    def run(self, fn, inputs, params, extra_key=()):
        key = tuple((t.data_ptr(), tuple(t.shape)) for t in inputs) + tuple((p.data_ptr(), p._version) for p in params) \
            + tuple(extra_key) + (_ops.pack_generation(),)
        ent = self._entries.get(key)
        if ent is None:
            if len(self._entries) >= self._max:
                self._entries.clear()
            # eager warm-up on a side stream (autotuning, lazy init), then capture
            s = torch.cuda.Stream()
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                fn(inputs)
                fn(inputs)
            torch.cuda.current_stream().wait_stream(s)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                out = fn(inputs)
            ent = (g, out, list(inputs))  # keep the inputs alive: the graph reads their storage
            self._entries[key] = ent
        ent[0].replay()
        return dict(ent[1]) if isinstance(ent[1], dict) else ent[1]
