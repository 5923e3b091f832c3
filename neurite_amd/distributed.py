"""
Data-parallel use of the hot path: one process per GPU (torch.distributed, backend "nccl" = RCCL over
xGMI), volumes sharded over ranks by batch entry.

The path shards by independent units: warping and per-(batch, label) Dice need
no communication at all.  Only the reductions that cross batch entries do:

  mean_dice  = mean over ALL B*L entries of dice*weights   (neurite/tf/metrics.py:499-510)
  cce        = mean over ALL B*V voxels                      (Keras SUM_OVER_BATCH_SIZE)

Both are ONE all-reduce of a handful of floats per step (latency-bound, ~2L+2 floats): each rank
contributes [sum of its weighted dice entries, number of entries] (or [loss sum, voxel count]).
For a batch entry split spatially across ranks the Dice numerator/denominator partials
`sums [B, 3, L]` are all-reduced instead, before the division (reduce_dice_sums).

The reference's own multi-device code is keras.utils.multi_gpu_model (neurite/tf/utils/model.py:298-321),
single-process tower replication; nothing of it is reproduced here.
"""

import torch
import torch.distributed as dist

__all__ = ['shard_range', 'PendingMean', 'all_reduce_mean_dice', 'mean_dice_pair', 'all_reduce_mean_pair', 'all_reduce_mean', 'reduce_dice_sums', 'dice_from_sums',
           'all_reduce_gradients', 'GradientBucket']


def _world(group=None):
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(group), dist.get_world_size(group)
    return 0, 1


_count_cache = {}


def _count_tensor(n, device):
    key = (int(n), str(device))
    t = _count_cache.get(key)
    if t is None:
        t = torch.full((1,), float(n), dtype=torch.float32, device=device)
        _count_cache[key] = t
    return t


def shard_range(n_items, rank=None, world_size=None):
    """Contiguous shard [lo, hi) of n_items for `rank`; remainders go to the first ranks."""
    r, w = _world()
    rank = r if rank is None else rank
    world_size = w if world_size is None else world_size
    base, rem = divmod(n_items, world_size)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


class PendingMean:
    """
    A mean whose all-reduce is in flight on RCCL's own stream.  `result()` makes the CURRENT stream wait for it (no host
    sync) and returns the 0-d tensor: call it after the next step's kernels have been enqueued and the collective's
    latency hides behind them.
    """

    def __init__(self, buf, work, value=None):
        self._buf, self._work, self._value = buf, work, value

    def result(self):
        if self._value is not None:
            return self._value
        if self._work is not None:
            self._work.wait()
            self._work = None
        return self._buf[0] / self._buf[1]


def all_reduce_mean_dice(local_dice, weights=None, group=None, async_op=False):
    """
    local_dice [B_local, L] (this rank's batch entries).  Returns the global mean over all ranks'
    entries of dice*weights -- what Dice.mean_dice would return on the gathered batch.  One all-reduce
    of 2 floats.  With async_op a PendingMean is returned instead (also when there is nothing to reduce).
    """
    d = local_dice
    _, w = _world(group)
    if d.device.type == 'cuda' and d.dim() == 2 and not (torch.is_grad_enabled() and d.requires_grad):
        # one launch writes [sum of dice * weights, number of entries] (csrc/dice.hip: dice_mean_pair) -- the buffer the
        # collective reduces; no sum / cat / scale launches around a ~1 ms step
        buf3 = _mean3(d, weights)
        if w == 1 and not (dist.is_available() and dist.is_initialized()):
            # alone: no collective to wait for, and the kernel has already divided
            return PendingMean(buf3[:2], None, value=buf3[2]) if async_op else buf3[2]
        buf = buf3[:2]
    else:
        # host tensors (the gloo tests of the collective logic), inputs of another rank than [B, L], and values a gradient is being
        # recorded for (the kernel's output carries no autograd graph): plain torch arithmetic
        if weights is not None:
            d = d * torch.as_tensor(weights, dtype=d.dtype, device=d.device)
        buf = torch.cat([d.sum(dtype=torch.float32).reshape(1), _count_tensor(d.numel(), d.device)])
    return all_reduce_mean_pair(buf, group=group, async_op=async_op)


def mean_dice_pair(local_dice, weights=None):
    """[sum of dice * weights, number of entries] of this rank's [B_local, L] Dice values as a 2-float device buffer
    (one kernel launch, csrc/dice.hip: dice_mean_pair) -- the operand of `all_reduce_mean_pair`.  Split from the
    collective so that the compute part of a step can be captured in a hipGraph (bench.py --graph)."""
    return _mean3(local_dice, weights)[:2]


def all_reduce_mean_pair(buf, group=None, async_op=False):
    """Global mean from a per-rank [sum, count] buffer: ONE all-reduce of 2 floats (in place), then sum / count."""
    _, w = _world(group)
    if w == 1 and not (dist.is_available() and dist.is_initialized()):
        if async_op:
            return PendingMean(buf, None)
        return buf[0] / buf[1]
    if async_op:
        return PendingMean(buf, dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group, async_op=True))
    dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group)
    return buf[0] / buf[1]


def _mean3(dice, weights):
    """[sum of dice * weights, number of entries, their quotient] as a 3-float device buffer, one launch"""
    from . import _lib
    lib = _lib.lib()
    dev = _lib.require_device(dice)
    if dice.dim() != 2:
        raise ValueError('local_dice must be [B_local, L]')
    B, L = dice.shape
    d = dice.contiguous().to(torch.float32)
    wt, per_batch = None, 0
    if weights is not None:                       # metrics.py:499-506: weights [1, L] (or [L]) or [B, L]
        wt = torch.as_tensor(weights, dtype=torch.float32, device=dev).contiguous()
        if wt.numel() == L:
            per_batch = 0
        elif wt.numel() == B * L:
            per_batch = 1
        else:
            raise ValueError('weights must be [L], [1, L] or [B, L]; got %s for dice %s' % (tuple(wt.shape), (B, L)))
    buf = torch.empty(3, dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        rc = lib.nrt_dice_mean_f32(_lib.ptr(d), _lib.ptr(wt), int(L), int(B), per_batch, _lib.ptr(buf), _lib.stream_ptr(dev))
    _lib.check(rc, 'nrt_dice_mean_f32')
    return buf


def all_reduce_mean(local_sum, local_count, group=None):
    """Global mean from per-rank (sum, count): the cross-entropy reduction over all B*V voxels."""
    _, w = _world(group)
    if w == 1:
        return local_sum.reshape(()).to(torch.float32) / float(local_count)
    buf = torch.cat([local_sum.reshape(1).to(torch.float32), _count_tensor(local_count, local_sum.device)])
    dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group)
    return buf[0] / buf[1]


def reduce_dice_sums(sums, group=None):
    """All-reduce the numerator/denominator partials [B, 3, L] of spatially split batch entries."""
    _, w = _world(group)
    if w > 1:
        sums = sums.clone()
        dist.all_reduce(sums, op=dist.ReduceOp.SUM, group=group)
    return sums


def dice_from_sums(sums, laplace_smoothing=0.):
    """
    dice [B, L] from (all-reduced) sums [B, 3, L] = sum t*p, sum t^2, sum p^2 (metrics.py:476-482),
    computed by the HIP finalize kernel (ROCm tensors only, like every compute entry point).
    """
    from . import _lib
    lib = _lib.lib()
    dev = _lib.require_device(sums)
    B, three, L = sums.shape
    assert three == 3, 'sums must be [B, 3, L]'
    s = sums.contiguous().to(torch.float32)
    out = torch.empty((B, L), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        rc = lib.nrt_dice_from_sums_f32(_lib.ptr(s), L, B, float(laplace_smoothing), _lib.ptr(out),
                                        _lib.stream_ptr(dev))
    _lib.check(rc, 'nrt_dice_from_sums_f32')
    return out


class GradientBucket:
    """
    ONE flat float32 buffer that IS the gradient storage of a set of parameters: every `p.grad` is a view into it, so the
    data-parallel step is `bucket.all_reduce()` -- one RCCL all-reduce of the whole buffer, no packing, no copy back, no allocation,
    and therefore capturable into the same hipGraph as the step itself.  xGMI is point-to-point (7 links x ~153 GB/s): a ring
    all-reduce of a small buffer is latency-bound, so a unet (a few MB of weights) goes out as one message.

        bucket = GradientBucket(net.parameters())        # once, after the optimizer exists; p.grad now alias the bucket
        loss.backward(); bucket.all_reduce(); opt.step(); bucket.zero_()

    (The reference's only multi-device code is neurite/tf/utils/model.py:298-321, Keras multi_gpu_model.)  Autograd accumulates into
    an existing .grad in place, so the views survive backward passes; `optimizer.zero_grad(set_to_none=True)` would drop them -- use
    `bucket.zero_()` (or zero_grad(set_to_none=False)).
    """

    def __init__(self, params, group=None, average=True):
        self.params = [p for p in params if p.requires_grad]
        self.group, self.average = group, average
        if not self.params:
            raise ValueError('GradientBucket: no parameter requires a gradient')
        dev = self.params[0].device
        if any(p.device != dev for p in self.params):
            raise ValueError('GradientBucket: all parameters must live on one device')
        # (every check BEFORE the first .grad is re-pointed: a refused model is left exactly as it was -- ADVICE r5)
        bad = [p.dtype for p in self.params if p.dtype != torch.float32]
        if bad:
            raise ValueError('GradientBucket: float32 parameters only (got %s)' % bad[0])
        self.sizes = [p.numel() for p in self.params]
        self.flat = torch.zeros(sum(self.sizes), dtype=torch.float32, device=dev)
        off = 0
        for p, n in zip(self.params, self.sizes):
            view = self.flat[off:off + n].view_as(p)
            if p.grad is not None:
                view.copy_(p.grad.to(torch.float32))
            p.grad = view
            off += n

    def intact(self):
        """every p.grad still aliases the bucket (False after zero_grad(set_to_none=True) or a re-assigned .grad)"""
        off = 0
        for p, n in zip(self.params, self.sizes):
            g = p.grad
            if g is None or g.data_ptr() != self.flat.data_ptr() + 4 * off or g.numel() != n:
                return False
            off += n
        return True

    def _require_intact(self):
        if not self.intact():
            raise RuntimeError('GradientBucket: a parameter gradient no longer aliases the bucket (zero_grad(set_to_none=True)?)')

    def zero_(self):
        # (at every world size: zeroing a buffer the optimizer no longer sees would silently keep accumulating gradients)
        self._require_intact()
        self.flat.zero_()

    def all_reduce(self, async_op=False):
        """sum (average) over the ranks, in place; returns the number of collective calls issued (0 at world size 1) or, with
        async_op, the work handle (None at world size 1)"""
        rank, world = _world(self.group)
        self._require_intact()
        if world == 1:
            return None if async_op else 0
        if self.average:
            # pre-divide: sum of g / W; one pass, and the collective's result needs no second kernel
            self.flat.div_(world)
        work = dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=self.group, async_op=async_op)
        return work if async_op else 1


def all_reduce_gradients(params, group=None, bucket_mb=256, average=True):
    """
    Data-parallel training step of the conv stack: sum (or average) the gradients of `params` over the ranks.
    Gradients are packed into flat float32 buckets so that a unet (a few MB of weights) is ONE RCCL all-reduce --
    xGMI is point-to-point (7 links x ~153 GB/s), a ring all-reduce of a small buffer is latency-bound, so fewer and
    larger messages win; `bucket_mb` only matters for models larger than a bucket.  In place; returns the number of
    all-reduce calls issued (0 at world size 1).  This form packs and unpacks (three passes and an allocation per step): for a
    training loop build a `GradientBucket` once instead -- the gradients then live in the flat buffer and nothing is copied.
    """
    rank, world = _world(group)
    grads = [p.grad for p in params if getattr(p, 'grad', None) is not None]
    if world == 1 or not grads:
        return 0
    limit = int(bucket_mb * (1 << 20) // 4)
    calls, i = 0, 0
    while i < len(grads):
        bucket, n = [], 0
        while i < len(grads) and (not bucket or n + grads[i].numel() <= limit):
            bucket.append(grads[i]); n += grads[i].numel(); i += 1
        flat = torch.cat([g.reshape(-1).to(torch.float32) for g in bucket])
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
        if average:
            flat /= world
        off = 0
        for g in bucket:
            g.copy_(flat[off:off + g.numel()].view_as(g))
            off += g.numel()
        calls += 1
    return calls
