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
        self.captures = 0
        self.shared_pool = None   # set to a torch.cuda.graph_pool_handle(): every capture allocates from it (replays of one cache are never concurrent)

    def clear(self):
        self._entries.clear()

    def key_of(self, inputs, params, extra_key=()):
        return tuple((t.data_ptr(), tuple(t.shape), tuple(t.stride())) for t in inputs) + tuple((p.data_ptr(), p._version) for p in params) \
            + tuple(extra_key) + (_ops.pack_generation(),)

    def has(self, key):
        return key in self._entries

    def run(self, fn, inputs, params, extra_key=(), retain_inputs=True, key=None):
        """Replay the capture of `fn(inputs)` for this input signature (device pointers, shapes, strides, parameter versions),
        capturing it first if there is none.  retain_inputs=False: the entry does NOT keep the input tensors alive -- for callers
        whose inputs come back at RECURRING addresses (a caching allocator in steady state: decoder.bind_inputs); the graph reads
        whatever lives at the captured addresses at replay time, and that is the tensors of the call that hits the key."""
        if key is None:
            key = self.key_of(inputs, params, extra_key)
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
            with torch.cuda.graph(g, pool=self.shared_pool):
                out = fn(inputs)
            self.captures += 1
            ent = (g, out, list(inputs) if retain_inputs else None)  # retained inputs stay alive: the graph reads their storage
            self._entries[key] = ent
        ent[0].replay()
        return dict(ent[1]) if isinstance(ent[1], dict) else ent[1]


class TrainStepGraph:
    """One whole training step -- forward, backward and the optimizer update -- as hipGraph replays.

    For steps that are bound by their python launches -- small images, small batches: ResNet18 at 160x96, batch 2 takes
    6.07 ms eagerly and 2.90 ms replayed.  At the benchmark sizes (1024x320 batch 8, 640x192 batch 12, NYUv2 640x480 batch 4)
    the eager step already keeps the queue full and the replay takes as long (34.2 vs 33.6, 10.5 vs 10.4, 44.3 vs 42.9 ms:
    the ~1500 kernels of a step cost the same dispatch gaps either way), so bench.py reports it beside the eager figure
    only.  Usage::

        opt = torch.optim.Adam(params, lr=1e-4, capturable=True)
        g = TrainStepGraph(lambda img, tgt: loss_of(dec(enc(img)), tgt), opt, inputs=(img, tgt))
        for img, tgt in loader:
            loss = g.step(img, tgt)          # copies into the static input buffers, replays; loss is a device scalar

    `inputs` become the graph's static input buffers (used in place).  `warmup` (>= 1) eager steps run first on a side
    stream (they are real optimisation steps: autotuning, MIOpen's solver choice and lazy initialisation must happen
    outside the capture).

    No autograd graph of an earlier eager step may still be alive when this is built: its AccumulateGrad nodes are bound to
    the stream that step ran on -- normally the legacy default stream, which must not be touched during a capture (ROCm
    aborts the process in hipStreamEndCapture).  Typical holders are a stored loss and the `outputs` dictionary the decoder
    modules keep like the reference's; pass the networks as `modules` to have those dictionaries emptied.  A graph that is
    still alive is detected during the warm-up and reported as a RuntimeError instead.

    With a `GradientExchange` the step is two graphs: forward+backward accumulating into the flat gradient buckets (the
    per-parameter hooks are bypassed), then the bucket all-reduces as ordinary RCCL launches, then the optimizer graph --
    the all-reduce is then not overlapped with the backward, which costs less than the launch gaps it removes up to a
    few GPUs per node; the eager step (hooks, overlap) stays available for the other regime.

    Every replay rewrites the parameters without bumping their autograd version counters, so `step` ends with
    `ops.invalidate_packs()`: packed-weight memos and inference graphs keyed on the weights are rebuilt on next use."""

    def __init__(self, loss_fn, optimizer, inputs=(), exchange=None, warmup=3, modules=()):
        import warnings
        for grp in optimizer.param_groups:
            if grp.get("capturable") is False:
                raise ValueError("TrainStepGraph needs an optimizer whose step can be captured: construct it with capturable=True")
        self.loss_fn, self.optimizer, self.exchange = loss_fn, optimizer, exchange
        self.inputs = list(inputs)
        for m in modules:
            for sub in m.modules():
                if isinstance(getattr(sub, "outputs", None), dict):
                    sub.outputs = {}
        cur = torch.cuda.current_stream()
        side = torch.cuda.Stream()
        side.wait_stream(cur)
        warn_always = torch.is_warn_always_enabled()
        torch.set_warn_always(True)
        try:
            with warnings.catch_warnings(record=True) as caught, torch.cuda.stream(side):
                warnings.simplefilter("always")
                for _ in range(max(1, warmup)):
                    self.eager_step()
        finally:
            torch.set_warn_always(warn_always)
        cur.wait_stream(side)
        if any("AccumulateGrad node's stream does not match" in str(w.message) for w in caught):
            raise RuntimeError("TrainStepGraph: an autograd graph of an earlier eager step is still alive (a stored loss, a "
                               "module's `outputs` dictionary ...): its AccumulateGrad nodes would run on the stream of that "
                               "step during the capture.  Drop those references (or pass the networks as modules=) first.")
        _ops.invalidate_packs()          # no memoised weight image may be reused inside the capture
        self.graph = torch.cuda.CUDAGraph()
        self.update = None
        with warnings.catch_warnings():
            # the warm-up's nodes (side stream) may be reused by the capture stream: harmless, both are capture-safe
            warnings.filterwarnings("ignore", message=".*AccumulateGrad node's stream does not match.*")
            if exchange is None:
                optimizer.zero_grad(set_to_none=True)      # the captured backward allocates the gradients in the graph's pool
                with torch.cuda.graph(self.graph):
                    self.loss = loss_fn(*self.inputs)
                    self.loss.backward()
                    optimizer.step()
            else:
                with torch.cuda.graph(self.graph):
                    exchange.zero_grad()
                    with exchange.no_sync():
                        self.loss = loss_fn(*self.inputs)
                        self.loss.backward()
                self.update = torch.cuda.CUDAGraph()
                with torch.cuda.graph(self.update, pool=self.graph.pool()):
                    optimizer.step()
        # the capture itself executed nothing: parameters, gradients and optimizer state are those after the warm-up steps

    def eager_step(self):
        """The same step without a graph (used for the warm-up; also the reference protocol the replay must match)."""
        if self.exchange is not None:
            self.exchange.zero_grad()
        else:
            self.optimizer.zero_grad(set_to_none=True)
        loss = self.loss_fn(*self.inputs)
        loss.backward()
        if self.exchange is not None:
            self.exchange.finish()
        self.optimizer.step()
        return loss

    def step(self, *inputs):
        if inputs:
            if len(inputs) != len(self.inputs):
                raise ValueError("TrainStepGraph.step takes the %d inputs it was captured with" % len(self.inputs))
            for dst, src in zip(self.inputs, inputs):
                if src is not dst:
                    dst.copy_(src, non_blocking=True)
        self.graph.replay()
        if self.update is not None:
            self.exchange.finish()
            self.update.replay()
        _ops.invalidate_packs()
        return self.loss
