"""Data-parallel sharding of independent HSI patches across the GPUs of one node (one process per GPU).

The path has no exchange inside the reverse loop: every (patch, spectral-group) latent is an independent
chain sharing read-only weights.  Collectives (RCCL over xGMI via torch.distributed "nccl";
"gloo" in the CPU tests) are therefore only: one broadcast of the weights at start-up and one all-gather
of the decoded cubes at the end.  The reference's only multi-GPU mode on this path is nn.DataParallel
(model/networks.py:113-115), which re-broadcasts all parameters on every forward.
"""
import torch
import torch.distributed as dist


def shard_range(n_items, rank, world):
    """Contiguous block of items for `rank`; the remainder goes to the low ranks.  -> (start, stop)"""
    base, rem = divmod(n_items, world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def shard_counts(n_items, world):
    return [shard_range(n_items, r, world)[1] - shard_range(n_items, r, world)[0] for r in range(world)]


def broadcast_module_(module, src=0, bucket_bytes=256 << 20):
    """One-time weight broadcast: parameters and buffers are flattened into large buckets (xGMI links are
    per-peer, so few large messages beat 374 small ones) and broadcast from `src`."""
    if not (dist.is_available() and dist.is_initialized()):
        return module
    # the tensors themselves, not `.data`: copy_ below must bump `_version` so that packed-weight caches notice
    tensors = list(module.parameters()) + [b for b in module.buffers() if b.is_floating_point()]
    bucket, size = [], 0

    def flush():
        nonlocal bucket, size
        if not bucket:
            return
        with torch.no_grad():
            flat = torch.cat([t.detach().reshape(-1) for t in bucket])
            dist.broadcast(flat, src=src)
            off = 0
            for t in bucket:
                n = t.numel()
                t.copy_(flat[off:off + n].view_as(t))
                off += n
        bucket, size = [], 0

    dtype = None
    for t in tensors:
        if dtype is not None and (t.dtype != dtype or size + t.numel() * t.element_size() > bucket_bytes):
            flush()
        dtype = t.dtype
        bucket.append(t)
        size += t.numel() * t.element_size()
    flush()
    return module


def grad_buckets(n_floats, bucket_bytes=64 << 20):
    """[(start, stop)] element ranges of a flat fp32 gradient buffer, last bucket first (= the order in which the backward
    pass completes them: the flat buffer follows the module order)."""
    per = max(1, bucket_bytes // 4)
    cuts = list(range(0, n_floats, per)) + [n_floats]
    return [(cuts[i], cuts[i + 1]) for i in reversed(range(len(cuts) - 1))]


def allreduce_grads_(flat_grad, bucket_bytes=64 << 20):
    """Sum the flat gradient buffer over the ranks of the default group in a few large buckets (xGMI links are per peer:
    few large messages; SURVEY 5: 391 MB fp32 per step for the shipped UNet), all in flight at once.  The caller divides by
    the world size (hsidm_adam_step's grad_scale).  The reference's counterpart is nn.DataParallel's reduce-add of the
    replicas' gradients onto GPU 0 (model/networks.py:113-115)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return flat_grad
    works = [dist.all_reduce(flat_grad[a:b], op=dist.ReduceOp.SUM, async_op=True) for a, b in grad_buckets(flat_grad.numel(), bucket_bytes)]
    for w in works:
        w.wait()
    return flat_grad


class GradReducer:
    """Overlaps the gradient all-reduce with the backward pass: the flat gradient buffer follows the module order, the backward
    pass fills it from the end, and every bucket is reduced (async, on RCCL's stream) as soon as everything above its lower
    edge is final.  `ready(lowest_final_offset)` after each layer's backward; `finish()` before the optimiser step."""

    def __init__(self, flat_grad, bucket_bytes=64 << 20):
        self.flat = flat_grad
        self.on = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
        self.buckets = grad_buckets(flat_grad.numel(), bucket_bytes)
        self.reset()

    def reset(self):
        self.next, self.works = 0, []

    def ready(self, lowest_final_offset):
        while self.next < len(self.buckets) and self.buckets[self.next][0] >= lowest_final_offset:
            a, b = self.buckets[self.next]
            if self.on:
                self.works.append(dist.all_reduce(self.flat[a:b], op=dist.ReduceOp.SUM, async_op=True))
            self.next += 1

    def finish(self):
        self.ready(0)
        for w in self.works:
            w.wait()
        fired = self.next
        self.reset()
        return fired


def all_gather_patches(local, n_total):
    """local: [n_local, ...] results of this rank's shard (shard_range order) -> [n_total, ...] on every rank."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return local
    world = dist.get_world_size()
    counts = shard_counts(n_total, world)
    cap = max(counts)
    pad = torch.zeros((cap,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[:local.shape[0]] = local
    out = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(out, pad)
    return torch.cat([o[:c] for o, c in zip(out, counts)], dim=0)


def run_sharded(items, fn, rank=None, world=None):
    """Apply fn to this rank's contiguous shard of `items` (a tensor, first axis = patches) and all-gather."""
    if rank is None:
        rank = dist.get_rank() if dist.is_initialized() else 0
    if world is None:
        world = dist.get_world_size() if dist.is_initialized() else 1
    lo, hi = shard_range(items.shape[0], rank, world)
    local = fn(items[lo:hi]) if hi > lo else None
    if local is None:   # empty shard: need a correctly-typed empty result
        probe = fn(items[:1])
        local = probe[:0]
    return all_gather_patches(local, items.shape[0])
