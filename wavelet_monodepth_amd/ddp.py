"""Data-parallel gradient exchange for the wavelet-monodepth training step.

The reference is single-GPU (KITTI/trainer.py:45).  Here: one process per GPU, the minibatch is sharded across
ranks, and ONE exchange step per iteration sums the gradients:

  * parameters are grouped into flat fp32 buckets in REVERSE forward order — the decoder first (its gradients
    are complete before the encoder backward has started), then encoder stages layer4 .. conv1;
  * every parameter's `.grad` is a view into its bucket, so autograd accumulates straight into the flat buffer
    (no gather copy);
  * a post-accumulate hook counts a bucket down; when it is full the bucket is all-reduced (sum, then 1/world)
    on a SIDE stream after an event recorded on the compute stream — the collective overlaps the rest of the
    backward pass;
  * `finish()` launches whatever is left (buckets holding unused parameters such as `encoder.fc`, whose
    gradient stays zero) and makes the compute stream wait for the side stream before `optimizer.step()`.

Backends: "rccl" = wmd_comm_* (libwmd_hip.so -> RCCL over xGMI); "torch" = torch.distributed.all_reduce on
whatever process group is initialised (gloo on CPU: used by the multi-process CPU tests).
"""
import ctypes as C
import os

import torch
import torch.distributed as dist


class _RcclBackend:
    def __init__(self, world, rank, store):
        from . import _lib
        self._lib = _lib
        l = _lib.lib()
        uid = C.create_string_buffer(128)
        if rank == 0:
            _lib.check(l.wmd_comm_unique_id(uid), "wmd_comm_unique_id")
            store.set("wmd_comm_uid", uid.raw)
        raw = store.get("wmd_comm_uid")
        self.comm = C.c_void_p()
        _lib.check(l.wmd_comm_init(C.byref(self.comm), raw, world, rank), "wmd_comm_init")
        self.side = torch.cuda.Stream()

    def allreduce(self, buf, scale):
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream())
        self.side.wait_event(ev)
        self._lib.check(self._lib.lib().wmd_comm_allreduce(self.comm, buf.data_ptr(), buf.numel(), float(scale),
                                                           self.side.cuda_stream), "wmd_comm_allreduce")

    def wait(self):
        torch.cuda.current_stream().wait_stream(self.side)

    def close(self):
        if self.comm:
            self._lib.lib().wmd_comm_destroy(self.comm)
            self.comm = None


class _TorchBackend:
    def __init__(self, group=None):
        self.group = group
        self.work = []

    def allreduce(self, buf, scale):
        w = dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        self.work.append((w, buf, scale))

    def wait(self):
        for w, buf, scale in self.work:
            w.wait()
            if scale != 1.0:
                buf.mul_(scale)
        self.work = []

    def close(self):
        pass


class GradientExchange:
    """groups: list of (name, iterable of parameters) in the order their gradients become ready
    (decoder first).  Call `finish()` between `loss.backward()` and `optimizer.step()`."""

    def __init__(self, groups, world=None, rank=None, backend="torch", store=None, process_group=None):
        self.world = dist.get_world_size() if world is None else world
        self.rank = dist.get_rank() if rank is None else rank
        self.buckets = []
        for name, params in groups:
            params = [p for p in params if p.requires_grad]
            if not params:
                continue
            n = sum(p.numel() for p in params)
            flat = torch.zeros(n, device=params[0].device, dtype=torch.float32)
            off = 0
            for p in params:
                p.grad = flat[off:off + p.numel()].view_as(p)   # autograd accumulates in place into the bucket
                off += p.numel()
            self.buckets.append({"name": name, "params": params, "flat": flat, "pending": len(params), "sent": False})
        if backend == "rccl":
            if store is None:
                store = dist.TCPStore(os.environ.get("MASTER_ADDR", "127.0.0.1"), int(os.environ.get("MASTER_PORT", "29500")) + 17,
                                      self.world, self.rank == 0)
            self.backend = _RcclBackend(self.world, self.rank, store)
        else:
            self.backend = _TorchBackend(process_group)
        self._hooks = []
        for b in self.buckets:
            for p in b["params"]:
                self._hooks.append(p.register_post_accumulate_grad_hook(self._make_hook(b)))

    def _make_hook(self, bucket):
        def hook(_param):
            bucket["pending"] -= 1
            if bucket["pending"] == 0 and not bucket["sent"]:
                self._send(bucket)
        return hook

    def _send(self, bucket):
        bucket["sent"] = True
        self.backend.allreduce(bucket["flat"], 1.0 / self.world)

    def zero_grad(self):
        """Keep the bucket views alive: zero the flat buffers instead of dropping `.grad`."""
        for b in self.buckets:
            b["flat"].zero_()
            b["pending"] = len(b["params"])
            b["sent"] = False
            for p in b["params"]:
                lo = b["flat"].data_ptr()
                if p.grad is None or not (lo <= p.grad.data_ptr() < lo + 4 * b["flat"].numel()):
                    raise RuntimeError("parameter gradient of bucket %s was re-allocated; use GradientExchange.zero_grad()" % b["name"])

    def finish(self):
        for b in self.buckets:
            if not b["sent"]:       # buckets with parameters that received no gradient this step
                self._send(b)
        self.backend.wait()

    def message_bytes(self):
        return {b["name"]: 4 * b["flat"].numel() for b in self.buckets}

    def close(self):
        for h in self._hooks:
            h.remove()
        self.backend.close()


def monodepth_groups(encoder, decoder):
    """Bucket order for an encoder/decoder depth network: decoder, then ResNet stages in backward order."""
    groups = [("decoder", list(decoder.parameters()))]
    enc = getattr(encoder, "encoder", encoder)
    seen = set()
    for stage in ("layer4", "layer3", "layer2", "layer1"):
        if hasattr(enc, stage):
            ps = list(getattr(enc, stage).parameters())
            seen.update(id(p) for p in ps)
            groups.append(("encoder." + stage, ps))
    rest = [p for p in encoder.parameters() if id(p) not in seen]
    groups.append(("encoder.stem", rest))
    return groups
