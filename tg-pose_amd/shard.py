"""Object sharding across GPUs (one process per GPU, torch.distributed; backend "nccl" is RCCL on ROCm).

In eval mode every object is independent (BatchNorm uses running statistics),
so the forward path shards by object with NO data-path collective: rank r owns a contiguous slice of
the objects, weights are replicated (110 MB).  The only communication is control: a barrier and a
max-reduction of the wall time around a timed region, and an optional gather of the (B,3)-sized pose
outputs to rank 0.
"""
import torch
import torch.distributed as dist


def object_range(total, rank, world):
    """Contiguous, balanced slice [lo, hi) of `total` objects for `rank` (first ranks take the remainder)."""
    base, rem = divmod(total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def max_over_ranks(seconds, device):
    """Wall time of the slowest rank (the job's time)."""
    if not (dist.is_available() and dist.is_initialized()):
        return float(seconds)
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_rows_to_rank0(rows, total, device):
    """Collect each rank's (n_r, C) per-object rows on rank 0 in object order; other ranks get None."""
    if not (dist.is_available() and dist.is_initialized()):
        return rows
    world, rank = dist.get_world_size(), dist.get_rank()
    C = rows.shape[1]
    sizes = [object_range(total, r, world) for r in range(world)]
    pad = max(hi - lo for lo, hi in sizes)
    buf = torch.zeros(pad, C, dtype=rows.dtype, device=device)
    buf[: rows.shape[0]] = rows
    out = [torch.zeros_like(buf) for _ in range(world)] if rank == 0 else None
    dist.gather(buf, out, dst=0)
    if rank != 0:
        return None
    return torch.cat([o[: hi - lo] for o, (lo, hi) in zip(out, sizes)], dim=0)


# ------------------------------------------------------------------------------------------------- training
def allreduce_gradients(params, bucket_bytes=64 << 20, average=True):
    """Data-parallel gradient exchange (BASELINE config 4): every rank holds the gradients of its
    own shard of the objects; sum (or average) them over the ranks so that all replicas take the same optimizer step.

    The 27.43 M fp32 gradients (109.7 MB) go out in a few large buckets -- xGMI is point to point, so RCCL's ring
    all-reduce is bound per link and wants few, large messages rather than one collective per tensor.  Buckets follow the
    reverse parameter order (the order backward produces them).  Call after loss.backward() and BEFORE
    clip_grad_norm_ (trainer/RL_TDA.py:223 clips the gradients the optimizer will see: the averaged ones).
    Returns the number of buckets sent.  Single process / uninitialised process group: no-op."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return 0
    world = dist.get_world_size()
    grads = [p.grad for p in reversed(list(params)) if p.grad is not None]
    buckets, cur, size = [], [], 0
    for g in grads:
        cur.append(g)
        size += g.numel() * g.element_size()
        if size >= bucket_bytes:
            buckets.append(cur)
            cur, size = [], 0
    if cur:
        buckets.append(cur)
    if grads and grads[0].is_cuda and dist.get_backend() == "gloo":
        # rehearsal path (CPU collectives over device tensors, ranks possibly sharing one GPU): gloo stages through the host
        # anyway; draining the device first keeps two processes' deep queues from waiting on each other across the exchange
        # (measured on the one-GPU box: 8.6 s -> 0.1 s per step with the backward replayed as a graph)
        torch.cuda.current_stream(grads[0].device).synchronize()
    for b in buckets:
        flat = torch.cat([g.reshape(-1) for g in b])
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
        if average:
            flat.div_(world)
        o = 0
        for g in b:
            g.copy_(flat[o:o + g.numel()].view_as(g))
            o += g.numel()
    return len(buckets)


class GradBuckets(object):
    """Flat gradient storage for the data-parallel step, in the order the backward pass completes it.

    Bucket 0 holds the layers AFTER the encoder (PH predictor 60 MB, the three heads 20 MB, the decoder 4 MB: 77 % of the
    109.7 MB); their gradients are final when the first backward segment ends, so their exchange runs on RCCL's stream WHILE
    the encoder's backward (bucket 1: 12 MB + the never-used proj_layer's 13 MB, which stays zero and is not sent) is still
    computing (autograd.GraphedStep's two-segment form).  Every ``p.grad`` is a view into its bucket's flat fp32 buffer: one
    collective per bucket, no packing copies.

    The exchange is reduce-scatter + all-gather on the flat buffer (each rank averages its 1/world slice in between): with RCCL
    over xGMI both halves move (world-1)/world of the bucket per GPU spread over all peers' links, where a call per tensor would
    pay 164 launch latencies and a ring over one link.  Backends without reduce_scatter_tensor (gloo: the CPU rehearsal and
    the tests) use one all-reduce per bucket.  Unmeasured on more than one GPU in this repository (no multi-GPU box is available
    to the build): correctness is covered by the two-rank gloo test; the RCCL branch itself (in-place reduce-scatter on a view
    of the flat buffer, all-gather, exchange-stream ordering against the second graph) executes on the one-GPU box at world
    size 1 under ``force_collectives`` (tests/test_gpu_parity.py::test_rccl_exchange_branch_world_size_one)."""

    def __init__(self, named_params, late_prefixes, skip=("proj_layer",)):
        named = [(n, p) for n, p in named_params if p.requires_grad]
        groups = [[(n, p) for n, p in named if n.startswith(tuple(late_prefixes)) and not any(s in n for s in skip)],
                  [(n, p) for n, p in named if not n.startswith(tuple(late_prefixes)) and not any(s in n for s in skip)]]
        self.unsent = [p for n, p in named if any(s in n for s in skip)]
        self.flat, self.params = [], []
        self._xstream = None
        world = dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1
        for g in groups:
            numel = sum(p.numel() for _, p in g)
            pad = (-numel) % (world * 4)                       # equal 16-byte-aligned slices for reduce-scatter
            dev = g[0][1].device if g else "cpu"
            flat = torch.zeros(numel + pad, device=dev, dtype=torch.float32)
            o = 0
            for _, p in g:
                p.grad = flat[o:o + p.numel()].view_as(p)
                o += p.numel()
            self.flat.append(flat)
            self.params.append([p for _, p in g])
        for p in self.unsent:
            if p.grad is None:
                p.grad = torch.zeros_like(p)

    def zero_(self):
        for f in self.flat:
            f.zero_()

    # tests / rehearsal: run the collectives even when the process group has a single rank (they then move no data but execute
    # the same calls on the same streams in the same order), so that the RCCL branch is exercised on a one-GPU box
    force_collectives = False

    def _active(self):
        if not (dist.is_available() and dist.is_initialized()):
            return False
        return dist.get_world_size() > 1 or self.force_collectives

    def reduce(self, i, async_op=True):
        """Start the exchange of bucket i (average over the ranks) behind the work already enqueued on the current stream;
        returns a handle for wait().  No-op without a process group.

        RCCL: the WHOLE exchange -- reduce-scatter into this rank's 1/world slice, the slice's division by world, all-gather
        back into the flat buffer -- is enqueued here on a dedicated exchange stream that first waits for the current stream
        (the gradients are complete there).  The calling stream is not blocked: whatever it enqueues next (the encoder's
        backward, graph 2 of the overlapped step) runs beside all three pieces; wait() only makes it wait for the recorded
        end event.  (Round 2 issued the all-gather from wait(), i.e. after the encoder's backward: only the reduce-scatter
        half was overlapped.)"""
        if not self._active():
            return None
        flat, world, rank = self.flat[i], dist.get_world_size(), dist.get_rank()
        if dist.get_backend() == "nccl":
            cur = torch.cuda.current_stream(flat.device)
            if self._xstream is None:
                self._xstream = torch.cuda.Stream(device=flat.device)
            xs = self._xstream
            xs.wait_stream(cur)
            with torch.cuda.stream(xs):
                shard = flat.view(world, -1)[rank]
                # blocking form = "the current (exchange) stream waits for the collective": nothing blocks on the host
                dist.reduce_scatter_tensor(shard, flat, op=dist.ReduceOp.SUM)
                shard.div_(world)
                dist.all_gather_into_tensor(flat, shard)
                done = torch.cuda.Event()
                done.record(xs)
            return ("rs", i, done, None)
        if flat.is_cuda:
            torch.cuda.current_stream(flat.device).synchronize()     # gloo stages device tensors through the host (rehearsal only)
        return ("ar", i, dist.all_reduce(flat, op=dist.ReduceOp.SUM, async_op=async_op), None)

    def wait(self, handle):
        """the current stream (clip_grad_norm_ and the optimizer run on it next) waits for the exchange started by reduce()"""
        if handle is None:
            return
        kind, i, work, _ = handle
        world = dist.get_world_size()
        if kind == "rs":
            torch.cuda.current_stream(self.flat[i].device).wait_event(work)
        else:
            if work is not None:
                work.wait()
            self.flat[i].div_(world)
