"""One camera view per GPU, gradient exchange over RCCL/xGMI.

The reference is single-process, single-GPU (train.py:48-57 renders one view
per optimizer step).  The only place the rasterizer path shards is BY VIEW:
every rank keeps a full replica of the Gaussian parameters, renders its own
view forward+backward, and the step ends with ONE exchange:

* all-reduce(mean) of the parameter gradients -- pws 3 + shs 48 + alphas 1 +
  scales 3 + rots 4 = 59 fp32 per Gaussian (236 MB at N = 1 M);
* all-reduce(sum) of the per-view densification statistics the reference
  accumulates in ``GSModel.update_density_info`` (gsmodel.py:214-230): the norm
  of dL/du per Gaussian (the norm is per view, so norms are reduced, not dus)
  and the visibility count.

No data-path collective exists anywhere else (binning/sort/draw are per view).
``torch.distributed`` backend "nccl" is RCCL on ROCm; the CPU tests run the same
code over "gloo".
"""
from __future__ import annotations

from typing import Dict, Iterable, List, Optional, Sequence

import os
import threading

import torch
import torch.distributed as dist

PARAM_ORDER = ("pws", "shs", "alphas", "scales", "rots")
GRAD_FLOATS_PER_GAUSSIAN = {"pws": 3, "shs": 48, "alphas": 1, "scales": 3, "rots": 4}


def views_for_rank(n_views: int, rank: int, world: int) -> List[int]:
    """Static round-robin assignment of camera views to ranks (view v -> rank v % world)."""
    return [v for v in range(n_views) if v % world == rank]


def _world(group=None) -> int:
    return dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1


def allreduce_mean_(tensors: Sequence[torch.Tensor], group=None) -> None:
    """In-place mean over ranks of every tensor, issued as asynchronous
    collectives and waited together (the 192-MB SH gradient dominates)."""
    world = _world(group)
    if world == 1:
        return
    backend = dist.get_backend(group)
    avg = backend == "nccl"  # RCCL implements ncclAvg; gloo has no AVG
    op = dist.ReduceOp.AVG if avg else dist.ReduceOp.SUM
    handles = [dist.all_reduce(t, op=op, group=group, async_op=True) for t in tensors]
    for h in handles:
        h.wait()
    if not avg:
        for t in tensors:
            t.div_(world)


def allreduce_sum_(tensors: Sequence[torch.Tensor], group=None) -> None:
    if _world(group) == 1:
        return
    handles = [dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group, async_op=True) for t in tensors]
    for h in handles:
        h.wait()


def flat_grad_buffer(tensors):
    """One 1-D tensor over the storage behind the ``.grad`` of the given parameter tensors, when they all
    came out of one ``backward`` call of this module and tile that storage exactly (up to the 16-B
    padding between slices); else None."""
    grads = [t.grad for t in tensors]
    if any(g is None or not g.is_contiguous() or g.dtype != torch.float32 for g in grads):
        return None
    st = grads[0].untyped_storage()
    if any(g.untyped_storage().data_ptr() != st.data_ptr() for g in grads):
        return None
    total = st.nbytes() // 4
    at = 0
    for off, cnt in sorted((g.storage_offset(), g.numel()) for g in grads):
        if not (at <= off < at + 4):      # slices may be padded to 16 B: gaps of up to 3 floats
            return None
        at = off + cnt
    if not (at <= total < at + 4):
        return None
    return torch.empty(0, dtype=torch.float32, device=grads[0].device).set_(st, 0, (total,))


class ChunkedExchange:
    """Gradient exchange overlapped with the tail of the backward pass ("overlap the all-reduce of
    early buckets with the tail of backward").

    While attached, ``fused.backward`` computes the per-Gaussian chain rule in ``chunks`` row chunks and calls
    ``on_chunk`` with the gradient slices of each chunk right after enqueueing its kernel; the slices are
    all-reduced (SUM) on a side stream that waits for exactly that kernel, so chunk k travels over xGMI while
    chunk k + 1 is computed -- with 4 chunks three quarters of the 236 MB are on their way before the backward
    pass ends.  ``finish()`` makes the caller's stream wait for the exchange and turns the sums into means.
    One collective per chunk (its five or six slices as one coalesced RCCL group call).  Every rank must
    attach for the same backward passes (the collectives have to match).

    RESTRICTION (checked, not assumed).  The buffers handed to ``on_chunk`` are the tensors ``fused.backward``
    RETURNS to autograd: they are reduced in place while autograd still owns them.  That is only right when
    autograd adopts them as ``.grad`` without copying and nothing else reads them before ``finish()``:

    * exactly ONE backward pass per ``attach()`` ... ``finish()`` -- a second view's ``.grad += new`` would run on
      the compute stream next to the in-flight all-reduce of the same storage (``begin_backward`` raises);
      a rank that renders several views per step accumulates them first and exchanges the flat buffer afterwards
      (``exchange_gradients`` / ``coalesce_grads``: one collective for V views);
    * the differentiated inputs are LEAF tensors without a ``.grad`` yet (``zero_grad(set_to_none=True)``):
      with torch activations in front of ``GSFunction`` (non-leaf inputs) downstream nodes would read the slices
      mid-reduce.  ``finish(params)`` verifies that every ``p.grad`` lives in the storage that was exchanged and
      raises otherwise -- unreduced gradients never go unnoticed.
    ``on_chunk`` is always handed FRESH view objects (never the tensor objects returned to autograd: a second
    reference would make ``AccumulateGrad`` clone them and the reduced values would not reach ``.grad``)."""

    def __init__(self, world=None, group=None, chunks=4):
        self.group = group
        self.world = _world(group) if world is None else world
        self.chunks = max(1, int(chunks))
        self.side = None
        self.works, self.tensors = [], []
        self.used = False
        self.backwards = 0          # backward passes seen since attach() / finish()
        self.storages = set()       # data_ptr of every storage handed over since the last finish()

    def attach(self):
        import contextlib
        from . import fused

        @contextlib.contextmanager
        def cm():
            prev, fused._exchange_hook = fused._exchange_hook, self
            self.backwards = 0
            try:
                yield self
            finally:
                fused._exchange_hook = prev
        return cm()

    def begin_step(self):
        """Start a step without ``attach()`` (a caller that names this object in its ``RenderOptions``)."""
        self.backwards = 0
        return self

    def begin_backward(self):
        """Called by ``fused.backward`` before it hands over the first chunk."""
        self.backwards += 1
        if self.backwards > 1:
            raise RuntimeError(
                "ChunkedExchange: a second backward pass inside one attach()/finish() -- its gradients would be "
                "accumulated into buffers that are being all-reduced.  Render ONE view per attach(), or accumulate "
                "the local views first and call exchange_gradients() afterwards (see the class docstring)")

    def on_chunk(self, tensors):
        tensors = [t for t in tensors if t is not None and t.numel() > 0]
        if not tensors or not (dist.is_available() and dist.is_initialized()):
            return
        if self.side is None:
            self.side = torch.cuda.Stream(device=tensors[0].device)
        ev = torch.cuda.Event()
        ev.record()                              # behind the kernel that produced this chunk
        self.side.wait_event(ev)
        with torch.cuda.stream(self.side):
            for t in tensors:
                t.record_stream(self.side)
            try:
                w = dist.all_reduce_coalesced(tensors, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
                self.works.append(w)
            except (RuntimeError, NotImplementedError, AttributeError):
                self.works += [dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
                               for t in tensors]
        self.tensors += tensors
        self.storages.update(t.untyped_storage().data_ptr() for t in tensors)
        self.used = True

    def finish(self, params=None) -> bool:
        """Wait (stream-wise) for everything handed over since the last call; False when nothing was.
        ``params`` (iterable of the differentiated leaf tensors): verify that their ``.grad`` IS the exchanged
        storage -- raises when autograd copied instead of adopting (non-leaf inputs, a pre-existing ``.grad``,
        a second reference to the returned tensors), because the reduced values would then be lost."""
        self.backwards = 0
        if not self.works:
            self.storages = set()
            return False
        with torch.cuda.stream(self.side):
            for w in self.works:
                if w is not None:
                    w.wait()
            if self.world > 1:
                torch._foreach_div_(self.tensors, float(self.world))
        torch.cuda.current_stream().wait_stream(self.side)
        storages, self.storages = self.storages, set()
        self.works, self.tensors = [], []
        if params is not None:
            for p in params:
                g = p.grad
                if g is None or g.untyped_storage().data_ptr() not in storages:
                    raise RuntimeError(
                        "ChunkedExchange.finish: a parameter's .grad is not the buffer that was all-reduced (autograd "
                        "copied or accumulated instead of adopting it: non-leaf input, existing .grad, or several "
                        "backward passes) -- the exchanged values did not reach it")
        return True


def coalesce_grads(tensors: Sequence[torch.Tensor]) -> List[torch.Tensor]:
    """``[flat]`` when the gradients of ``tensors`` are slices of one buffer (the fused backward allocates
    them that way: one 236-MB collective instead of five or six latency-bound ones), else the gradients."""
    flat = flat_grad_buffer(tensors)
    return [flat] if flat is not None else [t.grad for t in tensors]


def exchange_gradients(params: Dict[str, torch.Tensor], group=None, names: Optional[Iterable[str]] = None) -> None:
    """Mean-reduce ``p.grad`` of the parameter groups (default: the five activated ones) across ranks."""
    names = tuple(names) if names is not None else PARAM_ORDER
    for k in names:
        if params[k].grad is None:
            raise RuntimeError("parameter %r has no gradient to exchange" % k)
    if _world(group) == 1:
        return
    allreduce_mean_(coalesce_grads([params[k] for k in names]), group)


def density_stats(dloss_dus: torch.Tensor, mask: torch.Tensor, group=None):
    """Per-view ||dL/du|| and visibility, summed over ranks (gsmodel.py:219-228)."""
    grad_norm = torch.norm(dloss_dus.reshape(-1, 2), dim=-1)
    grad_norm = torch.where(mask, grad_norm, torch.zeros_like(grad_norm))
    count = mask.to(torch.int32)
    allreduce_sum_([grad_norm, count], group)
    return grad_norm, count


def grad_exchange_bytes(n_gaussians: int) -> int:
    return 4 * n_gaussians * sum(GRAD_FLOATS_PER_GAUSSIAN.values())


def factored_exchange_pays(world: int, views_per_rank: int, sh_dim: int = 48) -> bool:
    """Does a step move fewer bytes per link with its SH gradient FACTORED (``FactoredShGrad``) than with the rows
    all-reduced?  An all-gather of 3 floats per Gaussian and VIEW receives ``(world - 1) * views * 3`` floats per
    Gaussian; a (ring or direct) all-reduce of the rows moves ``2 (world - 1) / world * sh_dim``.  One rank: the
    factored form still saves HBM traffic once a rank renders several views (12 B instead of 2 x 192 B per Gaussian
    and accumulated view), and costs one extra kernel with a single view."""
    if world <= 1:
        return views_per_rank > 1
    return (world - 1) * views_per_rank * 3 < 2.0 * (world - 1) / world * sh_dim


class FactoredShGrad:
    """The SH-coefficient gradient of a step, kept in the form it is born in.

    ``dL/dshs`` of ONE view is an outer product per Gaussian (backward.md eq (5), gsmodel.py:84-85):
    ``dL/dcolour[rgb] * basis_c(pw - camera centre)`` -- 3 numbers and a direction every rank can recompute from the
    replicated ``pws``.  48 of the 59 gradient floats per Gaussian of SURVEY 8e's exchange are these rows.  While
    attached, every ``GSFunction`` / ``GSRawFunction`` backward (fused path) leaves its view's ``dL/dcolour [N,3]``
    and camera centre in a row of this object instead of forming the 48-float rows (autograd gets ``None`` for the
    SH tensors; the 11 other floats go on as before), and ``finish()``

    * all-gathers the rows of every rank (one collective: ``views * (12 N + 16)`` bytes per rank -- against an
      all-reduce of ``192 N`` bytes; RCCL over xGMI on GPUs, gloo in the CPU tests),
    * forms ``scale * sum over all views of dL/dcolour (x) basis`` ONCE (``egs_sh_grad_views``) into the ``.grad`` of
      the SH tensors (allocated, or added to when one exists).

    The result equals the all-reduced rows up to the order of the float sums.  Whether it pays: see
    ``factored_exchange_pays``.

        fx = FactoredShGrad(views=len(my_cams))
        with fx.attach():
            for cam in my_cams: GSFunction.apply(pws, shs, ..., cam)[0].backward(dl)
        fx.finish(pws, shs)                                        # shs.grad = mean over ranks of the sum over views
        exchange_gradients(params, names=("pws", "alphas", "scales", "rots"))     # 44 of the 236 bytes per Gaussian

    With ``ViewStreams``: attach around the lanes, call ``vs.finish()`` first (it orders every lane before the
    caller's stream), then ``finish``.  Only ``.backward()`` passes that accumulate into the leaves use it
    (``torch.autograd.grad`` gets its rows as always); not together with ``ChunkedExchange``."""

    def __init__(self, views: int, group=None):
        self.views = max(1, int(views))
        self.group = group
        self.rows = None          # [views, stride] float32: row v = {dL/dcolour [N,3], twc[3], padding}
        self.n = 0
        self.sh_dim = None
        self._next = 0
        self._lock = threading.Lock()     # backward runs on autograd's thread(s)

    @staticmethod
    def row_stride(n: int) -> int:
        return (3 * n + 3 + 3) // 4 * 4   # floats; a multiple of four keeps every row 16-B aligned

    def attach(self):
        import contextlib
        from . import fused

        @contextlib.contextmanager
        def cm():
            prev, fused._sh_sink = fused._sh_sink, self
            self._next = 0
            try:
                yield self
            finally:
                fused._sh_sink = prev
        return cm()

    def begin_step(self, n: Optional[int] = None, device=None):
        """Start a step without ``attach()`` (a caller that names this object in its ``RenderOptions``): no row is
        taken yet.  ``n`` / ``device``: allocate the row buffer NOW, on the caller's stream, before its views fork
        onto side streams -- allocated lazily by the first backward pass it would come from the caching allocator on
        whichever lane runs first, and a faster lane could write into a block the main stream's still-queued kernels
        use."""
        with self._lock:
            self._next = 0
            if n is not None and (self.rows is None or self.n != n or
                                  (device is not None and self.rows.device != torch.device(device))):
                self.rows = torch.empty((self.views, self.row_stride(n)), dtype=torch.float32,
                                        device=device if device is not None else "cuda")
                self.n = n
        return self

    def restart(self):
        """Forget the rows of the step so far (the step is rendered again)."""
        with self._lock:
            self._next = 0

    def slot(self, n: int, sh_dim: int, cam) -> torch.Tensor:
        """Called by ``fused.backward``: the [N,3] tensor this view's dL/dcolour goes to (the first 3 n floats of a
        row; the kernel writes the camera centre into the three floats behind them)."""
        with self._lock:
            v = self._next
            if v >= self.views:
                raise RuntimeError("FactoredShGrad(views=%d): a backward pass of view %d -- every rank gathers "
                                   "exactly `views` rows per step" % (self.views, v + 1))
            self._next = v + 1
            stride = self.row_stride(n)
            dev = cam.twc.device
            if self.rows is None or self.n != n or self.rows.device != dev:
                # (not zero-filled: a fill enqueued on THIS lane's stream would race the rows other lanes write; every
                # word a reader touches is written by the backward kernel, by the copy below or by gathered())
                self.rows = torch.empty((self.views, stride), dtype=torch.float32, device=dev)
                self.n = n
            if v > 0 and self.sh_dim != sh_dim:
                raise RuntimeError("FactoredShGrad: views of one step with different SH widths")
            self.sh_dim = sh_dim
        # (the backward kernel stores the view's camera centre behind the 3 n floats itself: EGS_BWD_FACTORED_SH)
        return self.rows[v][:3 * n].view(n, 3)

    def gathered(self):
        """-> (rows of every rank [world * views, stride], world).  Rows no backward pass filled count as zeros."""
        if self._next < self.views:
            self.rows[self._next:].zero_()
        world = _world(self.group)
        initialised = dist.is_available() and dist.is_initialized()
        # (EGS_FORCE_EXCHANGE=1: a one-rank process group still runs its collective -- bench.py's way of exercising
        # the exchange code on a single GPU)
        if world == 1 and not (initialised and os.environ.get("EGS_FORCE_EXCHANGE", "0") == "1"):
            return self.rows, 1
        out = torch.empty((world * self.views, self.rows.shape[1]), dtype=torch.float32, device=self.rows.device)
        try:
            dist.all_gather_into_tensor(out.view(-1), self.rows.view(-1), group=self.group)
        except (RuntimeError, NotImplementedError):     # a backend without the flat form (gloo on device tensors)
            dist.all_gather(list(out.view(world, -1).unbind(0)), self.rows.view(-1), group=self.group)
        return out, world

    def take(self):
        """The collective half of ``finish``: -> (rows of every rank, world) and the object is ready for the next step;
        None when this rank rendered nothing (one rank only: its peers would wait in the all-gather).  For a consumer
        that never needs the rows in memory (``FusedAdam.step(factored_sh=...)``: ``egs_adam_sh_factored``)."""
        if self.rows is None or self._next == 0:
            if _world(self.group) > 1:
                raise RuntimeError("FactoredShGrad: no backward pass on this rank in this step (its peers would wait "
                                   "in the all-gather)")
            return None
        rows, world = self.gathered()
        self._next = 0
        return rows, world

    def finish(self, pws: torch.Tensor, shs: torch.Tensor, high_shs: Optional[torch.Tensor] = None,
               average: bool = True, on_gathered=None) -> None:
        """``shs.grad`` (raw layout: ``shs`` = low_shs [N,3] and ``high_shs`` [N,K-3]) += scale * the step's SH gradient
        over the views of ALL ranks; scale = 1 / ranks (``average``: what ``exchange_gradients`` does to the other
        tensors) or 1 (``Trainer``, whose loss already carries 1 / views).  A collective: every rank calls it."""
        from . import _lib
        taken = self.take()
        if on_gathered is not None:      # (bench.py: an event between the all-gather and the kernel that forms the rows)
            on_gathered()
        if taken is None:
            return
        rows, world = taken
        n = self.n
        raw = high_shs is not None
        K = self.sh_dim
        lib = _lib.load()
        outs = [shs, high_shs] if (raw and K > 3) else [shs]
        widths = [3, K - 3] if (raw and K > 3) else [K]
        grads = [t.grad for t in outs]
        usable = all(g is not None and g.dtype == torch.float32 and g.is_contiguous() and g.shape == (n, w)
                     and not (g.data_ptr() & 15) for g, w in zip(grads, widths))
        accumulate = usable
        if not usable:
            fresh = [torch.empty((n, w), dtype=torch.float32, device=rows.device) for w in widths]
        tgt = grads if usable else fresh
        st = torch.cuda.current_stream(rows.device).cuda_stream
        _lib.check(lib.egs_sh_grad_views(
            n, K, rows.shape[0], pws.data_ptr(), rows.data_ptr(), rows.shape[1],
            (1.0 / world) if average else 1.0, tgt[0].data_ptr(), tgt[1].data_ptr() if len(tgt) > 1 else None,
            1 if accumulate else 0, st))
        if not usable:
            for t, g, f in zip(outs, grads, fresh):
                if g is None:
                    t.grad = f
                else:
                    g.add_(f)




class ViewStreams:
    """The views of ONE rank's share of a step, dealt round-robin to ``n_streams`` HIP streams.

    One view is a chain of dependent kernels: the per-Gaussian pass, the two sorts and the scans between them are
    short and latency-bound (a quarter of a view's time with most of the chip idle), the two draw kernels issue-bound.
    Views are independent until their gradients are added, so two of them on two streams fill each other's gaps:
    measured on the 1 M / 1080p scene, independent steps take 0.87 ms each on one stream, 0.76 on two, 0.74 on three
    and 0.77 on four (tools/lab/lab_two_streams.py).

    Every stream gets its own autograd leaves -- detached aliases of the parameters (same storage, separate
    ``.grad``) -- so the views of a stream accumulate among themselves (``fused.accumulate_in_kernel`` works per
    lane) and no two streams write one gradient buffer; ``finish()`` makes the caller's stream wait for the others
    and adds their accumulators to the parameters' ``.grad`` (one flat add per extra stream: 3 x 236 MB of traffic
    at N = 1 M, ~0.1 ms per step against the ~0.1 ms saved per VIEW).

        vs = ViewStreams(params, 2)
        vs.begin()                                  # parameters are final: side streams may start
        for i, cam in enumerate(cams):
            with vs.lane(i) as leaves:              # torch.cuda.stream(...) of lane i % n_streams
                image, mask = GSFunction.apply(*leaves, us_i, cam); loss(image).backward()
        vs.finish()                                 # params[k].grad = sum over the lanes

    Per-view results the caller keeps (losses, statistics) are produced on the lane's stream: accumulate them per
    lane (``lane_index(i)``) and combine after ``finish()``, which orders every lane before the caller's stream.
    On a CPU device (tests) the lanes run one after the other on the host; the leaf aliasing and the final sum are
    the same code.
    """

    def __init__(self, params: Sequence[torch.Tensor], n_streams: int = 4):
        self.params = list(params)
        if not self.params:
            raise ValueError("ViewStreams needs at least one parameter tensor")
        for p in self.params:
            if not (p.is_leaf and p.requires_grad):
                raise ValueError("ViewStreams parameters must be autograd leaves that require grad")
        self.n = max(1, int(n_streams))
        dev = self.params[0].device
        self.cuda = dev.type == "cuda"
        # experiment knobs: EGS_VIEW_STREAM_PRIO = 1: side lanes alternate high / low stream priority around
        # the caller's (lane 0) normal one; EGS_VIEW_STAGGER_US = t: lane k of a step starts k * t microseconds late
        # (a one-workgroup spin kernel), so that the lanes' latency-bound and issue-bound phases do not coincide
        self._stagger = float(os.environ.get("EGS_VIEW_STAGGER_US", "0") or 0)
        if self.cuda and os.environ.get("EGS_VIEW_STREAM_PRIO", "0") == "1":
            least, greatest = torch.cuda.Stream.priority_range()
            prios = [greatest if (k % 2 == 0) else least for k in range(self.n - 1)]
            self.streams = [torch.cuda.Stream(device=dev, priority=pr) for pr in prios]
        else:
            self.streams = [torch.cuda.Stream(device=dev) for _ in range(self.n - 1)] if self.cuda else []
        # lane 0 = the caller's stream and the parameters themselves
        self.leaves = [self.params] + [[p.detach().requires_grad_(True) for p in self.params]
                                       for _ in range(self.n - 1)]
        self._main = None
        self._open = False

    def lane_index(self, i: int) -> int:
        return i % self.n

    def begin(self):
        """Call when the parameters hold their values for this step (after the optimizer step that produced them was
        enqueued on the current stream).  Side streams wait for that point; their accumulators start empty."""
        for k in range(1, self.n):
            lane = self.leaves[k]
            if any(q.data_ptr() != p.data_ptr() or q.shape != p.shape for p, q in zip(self.params, lane)):
                # the caller re-allocated a parameter (densification): alias the new storage
                lane = self.leaves[k] = [p.detach().requires_grad_(True) for p in self.params]
            for q in lane:
                q.grad = None
        if self.cuda:
            self._main = torch.cuda.current_stream(self.params[0].device)
            for k, s in enumerate(self.streams):
                s.wait_stream(self._main)
                if self._stagger > 0:
                    with torch.cuda.stream(s):      # ~2.1 cycles per ns at the clocks a training run sees
                        torch.cuda._sleep(int((k + 1) * self._stagger * 2100))
        self._open = True

    class _Lane:
        def __init__(self, vs, k):
            self.vs, self.k, self.ctx = vs, k, None

        def __enter__(self):
            if self.vs.cuda and self.k > 0:
                self.ctx = torch.cuda.stream(self.vs.streams[self.k - 1])
                self.ctx.__enter__()
            return self.vs.leaves[self.k]

        def __exit__(self, et, ev, tb):
            if self.ctx is not None:
                self.ctx.__exit__(et, ev, tb)
            return False

    def lane(self, i: int):
        if not self._open:
            raise RuntimeError("ViewStreams.lane() outside begin() ... finish()")
        return ViewStreams._Lane(self, i % self.n)

    def finish(self):
        """The caller's stream waits for every lane; the lanes' accumulators are added to the parameters' ``.grad``
        (a parameter no lane-0 view touched adopts the first accumulator instead of adding to zeros)."""
        if not self._open:
            raise RuntimeError("ViewStreams.finish() without begin()")
        self._open = False
        if self.cuda:
            for s in self.streams:
                self._main.wait_stream(s)
        for k, lane in enumerate(self.leaves[1:]):
            if all(q.grad is None for q in lane):
                continue
            # (tensors without a gradient in this lane -- the SH tensors under FactoredShGrad -- are left out: the
            # others still tile one buffer)
            pq = [(p, q) for p, q in zip(self.params, lane) if q.grad is not None]
            have = all(p.grad is not None for p, _ in pq)
            fa = flat_grad_buffer([p for p, _ in pq]) if have else None
            fb = flat_grad_buffer([q for _, q in pq]) if have else None
            if fa is not None and fb is not None and fa.numel() == fb.numel() and \
                    all(p.grad.storage_offset() == q.grad.storage_offset() for p, q in pq):
                if self.cuda:
                    fb.record_stream(self._main)
                fa.add_(fb)                     # both came out of fused.backward: ONE add over 59 N floats
            else:
                for p, q in zip(self.params, lane):
                    if q.grad is None:
                        continue
                    if self.cuda:
                        q.grad.record_stream(self._main)
                    if p.grad is None:
                        p.grad = q.grad
                    else:
                        p.grad.add_(q.grad)
            for q in lane:
                q.grad = None
