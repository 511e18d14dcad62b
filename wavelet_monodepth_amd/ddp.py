"""Data-parallel gradient exchange for the wavelet-monodepth training step.

The reference is single-GPU (KITTI/trainer.py:45).  Here: one process per GPU, the minibatch is sharded across
ranks, and ONE exchange step per iteration sums the gradients:

  * parameters are grouped into flat fp32 buckets of ~25 MB in REVERSE forward order — the decoder first (its
    gradients are complete before the encoder backward has started), then the encoder's parameters from its last
    registered module back to its stem, whatever the encoder is (ResNet stages, 160 DenseNet layers, MobileNet blocks);
  * every parameter's `.grad` is a view into its bucket, so autograd accumulates straight into the flat buffer
    (no gather copy);
  * a post-accumulate hook counts a bucket down; when it is full the bucket is all-reduced (sum, then 1/world)
    on a SIDE stream after an event recorded on the compute stream — the collective overlaps the rest of the
    backward pass (from the second step on: the first step sends everything in `finish()`, after the ranks have
    agreed that they saw the same gradient arrivals per bucket — see `static_graph`) (xGMI rings are per-link bound: a handful of 25 MB messages keeps every link busy without
    paying the per-collective latency 160 times);
  * `finish()` launches whatever is left (buckets holding unused parameters such as `encoder.fc`, whose
    gradient stays zero), makes the compute stream wait for the side stream before `optimizer.step()`, and re-arms
    every bucket for the next step — whatever the caller does with `zero_grad` afterwards;
  * construction broadcasts parameters and floating-point buffers (BatchNorm running statistics) from rank 0, so
    the replicas start identical like under torch's DistributedDataParallel; `sync_buffers()` repeats the buffer
    broadcast on demand (e.g. before evaluation / checkpointing).

Backends: "rccl" = wmd_comm_* (libwmd_hip.so -> RCCL over xGMI); "torch" = torch.distributed collectives on
whatever process group is initialised (gloo on CPU: used by the multi-process CPU tests).
"""
import contextlib
import ctypes as C
import os

import warnings

import torch
import torch.distributed as dist

BUCKET_BYTES = 25 << 20


class _GroupStore:
    """Hands rank 0's RCCL unique id to the other ranks through the torch.distributed process group that is already up
    (no second rendezvous, no extra port); the set / get pair of a c10d store, for this one key."""

    def __init__(self, group=None):
        self.group, self.value = group, None

    def set(self, key, value):
        self.value = value

    def get(self, key):
        box = [self.value]
        dist.broadcast_object_list(box, src=0, group=self.group)
        return box[0]


class _RcclBackend:
    def __init__(self, world, rank, store):
        from . import _lib
        self._lib = _lib
        l = _lib.lib()
        # dmabuf IPC: without HSA_ENABLE_IPC_MODE_LEGACY=0 RCCL's communicator set-up fails on this driver (hipIpcGetMemHandle:
        # invalid argument).  The HIP runtime reads the variable when it initialises, so it belongs in the launcher's
        # environment (bench.py sets it before its first HIP call); setting it here would be too late -- say so instead.
        if world > 1 and os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY") != "0":
            warnings.warn("HSA_ENABLE_IPC_MODE_LEGACY=0 is not set in this process: RCCL communicator set-up between processes is "
                          "known to fail on this driver without it (export it before the first HIP call)")
        uid = C.create_string_buffer(128)
        if rank == 0:
            _lib.check(l.wmd_comm_unique_id(uid), "wmd_comm_unique_id")
            store.set("wmd_comm_uid", uid.raw)
        raw = store.get("wmd_comm_uid")
        self.comm = C.c_void_p()
        try:
            _lib.check(l.wmd_comm_init(C.byref(self.comm), raw, world, rank), "wmd_comm_init")
        except _lib.WmdError as e:
            #  name the known cause instead of leaving the caller with RCCL's `hipIpcGetMemHandle: invalid argument`
            if world > 1 and os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY") != "0":
                raise _lib.WmdError("%s -- rank %d of %d: HSA_ENABLE_IPC_MODE_LEGACY=0 is not in this process' environment; the host "
                                    "driver only supports dmabuf IPC and RCCL's communicator set-up between processes fails without "
                                    "it.  Export it in the LAUNCHER (before the first HIP call; INTEGRATION.md, 'Multi-GPU launch')"
                                    % (e, rank, world)) from e
            raise _lib.WmdError("%s -- rank %d of %d (HSA_ENABLE_IPC_MODE_LEGACY=%s)" % (e, rank, world, os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY"))) from e
        self.world = world
        self.side = torch.cuda.Stream()

    def _fork(self):
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream())
        self.side.wait_event(ev)

    def allreduce(self, buf, scale):
        self._fork()
        self._lib.check(self._lib.lib().wmd_comm_allreduce(self.comm, buf.data_ptr(), buf.numel(), float(scale),
                                                           self.side.cuda_stream), "wmd_comm_allreduce")

    def broadcast(self, buf, root=0):
        self._fork()
        self._lib.check(self._lib.lib().wmd_comm_broadcast(self.comm, buf.data_ptr(), buf.numel(), root,
                                                           self.side.cuda_stream), "wmd_comm_broadcast")

    def wait(self):
        torch.cuda.current_stream().wait_stream(self.side)

    def info(self):
        v, w, r = C.c_int(), C.c_int(), C.c_int()
        self._lib.check(self._lib.lib().wmd_comm_info(self.comm, C.byref(v), C.byref(w), C.byref(r)), "wmd_comm_info")
        return {"backend": "rccl", "rccl_version": v.value, "comm_world": w.value, "comm_rank": r.value}

    def timed_allreduce(self, buf, reps):
        """-> mean milliseconds of `reps` all-reduces of buf on the side stream (hipEvents on that stream)."""
        self._fork()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        l = self._lib.lib()
        self._lib.check(l.wmd_comm_allreduce(self.comm, buf.data_ptr(), buf.numel(), 1.0, self.side.cuda_stream), "wmd_comm_allreduce")
        e0.record(self.side)
        for _ in range(reps):
            self._lib.check(l.wmd_comm_allreduce(self.comm, buf.data_ptr(), buf.numel(), 1.0, self.side.cuda_stream), "wmd_comm_allreduce")
        e1.record(self.side)
        e1.synchronize()
        return e0.elapsed_time(e1) / reps

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

    def broadcast(self, buf, root=0):
        dist.broadcast(buf, src=root, group=self.group)

    def wait(self):
        for w, buf, scale in self.work:
            w.wait()
            if scale != 1.0:
                buf.mul_(scale)
        self.work = []

    def info(self):
        return {"backend": "torch:" + str(dist.get_backend(self.group)), "comm_world": dist.get_world_size(self.group),
                "comm_rank": dist.get_rank(self.group)}

    def timed_allreduce(self, buf, reps):
        import time
        dist.all_reduce(buf, group=self.group)
        if buf.is_cuda:
            torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            dist.all_reduce(buf, group=self.group)
        if buf.is_cuda:
            torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps * 1e3

    def close(self):
        pass


class _NullBackend:
    """world == 1 without a process group: nothing to exchange."""

    def allreduce(self, buf, scale):
        if scale != 1.0:
            buf.mul_(scale)

    def broadcast(self, buf, root=0):
        pass

    def wait(self):
        pass

    def close(self):
        pass


class GradientExchange:
    """groups: list of (name, iterable of parameters) in the order their gradients become ready
    (decoder first; `bucket_groups` builds it).  Call `finish()` between `loss.backward()` and `optimizer.step()`.

    modules: the nn.Modules whose parameters AND buffers rank 0 broadcasts at construction (default: only the
    parameters of `groups` are broadcast).

    static_graph (default True, torch DDP's `static_graph` made explicit): every step reaches the same parameters on every
    rank.  A bucket that holds parameters the loss never reaches (resnet.fc, densenet.norm5 / classifier) then learns on the
    first step not to wait for them and leaves as soon as its used parameters are in -- AFTER the ranks have agreed on the
    per-bucket arrival counts (one tiny all-reduce on that step; a mismatch raises on every rank instead of letting the
    ranks issue their collectives in different orders).  static_graph=False never learns: incomplete buckets always leave
    in finish(), in bucket order, on every rank."""

    def __init__(self, groups, world=None, rank=None, backend="torch", store=None, process_group=None, modules=(),
                 broadcast_from_rank0=True, static_graph=True):
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
            # arm = how many gradient arrivals complete the bucket; learned down on the first step for buckets that hold
            # parameters the loss never reaches (resnet.fc, densenet.norm5/classifier): see finish()
            self.buckets.append({"name": name, "params": params, "flat": flat, "arm": len(params), "count": 0, "sent": False})
        if backend == "rccl":
            if store is None and dist.is_available() and dist.is_initialized() and dist.get_world_size(process_group) == self.world:
                store = _GroupStore(process_group)
            if store is None:
                store = dist.TCPStore(os.environ.get("MASTER_ADDR", "127.0.0.1"), int(os.environ.get("MASTER_PORT", "29500")) + 17,
                                      self.world, self.rank == 0)
            self.backend = _RcclBackend(self.world, self.rank, store)
        elif self.world == 1 and not (dist.is_available() and dist.is_initialized()):
            self.backend = _NullBackend()
        else:
            self.backend = _TorchBackend(process_group)
        self.static_graph = bool(static_graph)
        self._arms_agreed = False
        self.enabled = True        # False: the all-reduces are skipped (local gradients; used to time the exposed cost)
        self._accumulating = False
        self._modules = list(modules)
        self._hooks = []
        for b in self.buckets:
            for p in b["params"]:
                self._hooks.append(p.register_post_accumulate_grad_hook(self._make_hook(b)))
        if broadcast_from_rank0 and self.world > 1:
            self.sync_parameters()
            self.sync_buffers()

    # -- replica synchronisation -----------------------------------------------------------------
    def _broadcast_tensors(self, tensors):
        """Rank 0's values of `tensors` to every rank, packed into flat messages of <= BUCKET_BYTES."""
        tensors = [t for t in tensors if t.numel() and t.dtype == torch.float32]
        i = 0
        while i < len(tensors):
            chunk, size = [], 0
            while i < len(tensors) and (not chunk or size + 4 * tensors[i].numel() <= BUCKET_BYTES):
                chunk.append(tensors[i])
                size += 4 * tensors[i].numel()
                i += 1
            flat = torch.cat([t.detach().reshape(-1) for t in chunk])
            self.backend.broadcast(flat, 0)
            self.backend.wait()
            off = 0
            with torch.no_grad():
                for t in chunk:
                    t.copy_(flat[off:off + t.numel()].view_as(t))
                    off += t.numel()

    def sync_parameters(self):
        seen, ps = set(), []
        for b in self.buckets:
            ps += b["params"]
        for m in self._modules:
            ps += list(m.parameters())
        uniq = [p for p in ps if not (id(p) in seen or seen.add(id(p)))]
        self._broadcast_tensors(uniq)

    def sync_buffers(self):
        """BatchNorm running statistics (every fp32 buffer of `modules`) from rank 0."""
        bufs = [b for m in self._modules for b in m.buffers()]
        if bufs:
            self._broadcast_tensors(bufs)

    # -- per-step protocol -----------------------------------------------------------------------
    def _make_hook(self, bucket):
        def hook(_param):
            if self._accumulating:
                return
            if bucket["sent"]:
                raise RuntimeError("gradient bucket %s received a gradient after it was all-reduced: call finish() once per "
                                   "backward pass, or wrap extra backward passes in no_sync()" % bucket["name"])
            bucket["count"] += 1
            # a full bucket leaves at once (overlapping the rest of the backward pass) only once the ranks are known to see
            # the same arrivals: from the second step of a static graph on.  On the first step, and always with
            # static_graph=False, every bucket leaves in finish(), in bucket order, on every rank.
            if bucket["count"] == bucket["arm"] and self.static_graph and (self._arms_agreed or self.world == 1):
                self._send(bucket)
        return hook

    def _send(self, bucket):
        bucket["sent"] = True
        if self.enabled:
            self.backend.allreduce(bucket["flat"], 1.0 / self.world)

    @contextlib.contextmanager
    def no_sync(self):
        """Gradient accumulation: backward passes inside this context only accumulate into the buckets."""
        self._accumulating = True
        try:
            yield
        finally:
            self._accumulating = False

    def zero_grad(self):
        """Keep the bucket views alive: zero the flat buffers instead of dropping `.grad`.  (The optimizer's own
        `zero_grad(set_to_none=False)` is equivalent; `set_to_none=True` would detach the views and is refused here.)"""
        for b in self.buckets:
            b["flat"].zero_()
            lo, hi = b["flat"].data_ptr(), b["flat"].data_ptr() + 4 * b["flat"].numel()
            for p in b["params"]:
                if p.grad is None or not (lo <= p.grad.data_ptr() < hi):
                    raise RuntimeError("parameter gradient of bucket %s was re-allocated; use GradientExchange.zero_grad() "
                                       "or optimizer.zero_grad(set_to_none=False)" % b["name"])

    def _agree_on_arms(self):
        """First step of a static graph: every rank must have seen the same number of gradient arrivals per bucket before
        any bucket's arm is lowered (ranks with different arms would launch their all-reduces in different orders: a hang,
        or sums over mismatched buffers).  Sum and sum of squares of the counts over the ranks: equal counts <=> W * sum(c^2)
        == (sum c)^2 per bucket (small integers, exact in fp32)."""
        counts = [float(b["count"]) for b in self.buckets]
        dev = self.buckets[0]["flat"].device
        v = torch.tensor(counts + [c * c for c in counts], device=dev, dtype=torch.float32)
        self.backend.allreduce(v, 1.0)
        self.backend.wait()
        tot = v.tolist()
        n = len(counts)
        bad = [self.buckets[i]["name"] for i in range(n) if abs(self.world * tot[n + i] - tot[i] * tot[i]) > 0.5]
        if bad:
            raise RuntimeError("GradientExchange(static_graph=True): the ranks reached different parameter sets in buckets %s "
                               "(rank %d saw %s arrivals); use static_graph=False for data-dependent graphs"
                               % (bad, self.rank, [int(c) for c in counts]))
        self._arms_agreed = True

    def finish(self):
        if self.static_graph and not self._arms_agreed and self.enabled and self.world > 1 and self.buckets \
                and not self._accumulating:
            self._agree_on_arms()
        for b in self.buckets:
            if not b["sent"]:
                # some parameters of this bucket received no gradient (their slice stays zero): send it now, and -- static
                # graphs only, after the ranks agreed on the counts -- from the next step on do not wait for them: the
                # bucket then leaves as soon as its used parameters are in (a parameter that turns up later raises in the hook)
                if self.static_graph and (self._arms_agreed or self.world == 1) and 0 < b["count"] < b["arm"]:
                    b["arm"] = b["count"]
                self._send(b)
        self.backend.wait()
        for b in self.buckets:      # re-arm: the next backward starts from a clean count whatever zero_grad the caller uses
            b["count"] = 0
            b["sent"] = False

    def message_bytes(self):
        return {b["name"]: 4 * b["flat"].numel() for b in self.buckets}

    def comm_info(self):
        """What the communicator this rank exchanges through reports about itself (backend, RCCL version code, world, rank)."""
        return self.backend.info() if hasattr(self.backend, "info") else {"backend": "none", "comm_world": 1, "comm_rank": 0}

    def bucket_timing(self, reps=3):
        """Stand-alone all-reduce time of every bucket's message size (a scratch buffer, nothing overlapped; every rank must
        call it at the same point): -> {bucket: {"bytes", "ms", "bus_GBps"}}, bus bandwidth = 2 (W - 1) / W x bytes / time,
        the figure a ring is bound by per link (xGMI: ~153 GB/s per link and direction)."""
        out = {}
        if not hasattr(self.backend, "timed_allreduce"):
            return out
        for b in self.buckets:
            scratch = torch.zeros_like(b["flat"])
            ms = self.backend.timed_allreduce(scratch, reps)
            nbytes = 4 * scratch.numel()
            out[b["name"]] = {"bytes": nbytes, "ms": round(ms, 4),
                              "bus_GBps": round(2.0 * (self.world - 1) / max(self.world, 1) * nbytes / (ms * 1e-3) / 1e9, 2) if ms > 0 else None}
        return out

    def close(self):
        for h in self._hooks:
            h.remove()
        self.backend.close()


def _chunks(name, params, bucket_bytes):
    """Split an ordered parameter list into consecutive buckets of <= bucket_bytes (a single larger tensor gets its own)."""
    out, cur, size = [], [], 0
    for p in params:
        nb = 4 * p.numel()
        if cur and size + nb > bucket_bytes:
            out.append(cur)
            cur, size = [], 0
        cur.append(p)
        size += nb
    if cur:
        out.append(cur)
    return [(name if k == 0 else "%s.%d" % (name, k), ps) for k, ps in enumerate(out)]


def bucket_groups(encoder, decoder, bucket_bytes=BUCKET_BYTES):
    """Bucket order for an encoder/decoder depth network: the decoder's parameters (reverse registration order = the
    order its backward produces them: heads and fine levels first), then the encoder's in reverse registration order —
    torch registers modules in forward order, so this is the order the encoder backward completes them for ResNet,
    DenseNet and MobileNet alike — cut into ~bucket_bytes messages."""
    seen = set()

    def fresh(ps):
        out = [p for p in ps if id(p) not in seen]
        seen.update(id(p) for p in out)
        return out

    groups = _chunks("decoder", fresh(reversed(list(decoder.parameters()))), bucket_bytes)
    groups += _chunks("encoder", fresh(reversed(list(encoder.parameters()))), bucket_bytes)
    return groups


monodepth_groups = bucket_groups   # round-1 name
