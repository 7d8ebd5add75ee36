"""Data-parallel plumbing: one process per GPU, image batch sharded over ranks.

The forward path has NO data-path collective: images are independent through the whole model (no BatchNorm; the only
coupling is the mean in the loss).  Ranks only meet for (1) the max-over-ranks timing of bench.py, (2) the sum of the
per-rank scalars that are logged (bits, squared error), and -- in the training step -- (3) ONE all-reduce(sum) of the flat
fp32 gradient bucket (RCCL over xGMI when the backend is "nccl").  All of it goes through torch.distributed so the same
code runs on gloo (CPU tests, world_size 2) and on RCCL.
"""
import os

import torch
import torch.distributed as dist


def env_rank():
    return int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1)), int(os.environ.get("LOCAL_RANK", 0))


def init(backend=None, device=None):
    """Initialise the default process group from the torchrun environment (no-op for world_size 1)."""
    rank, world, _ = env_rank()
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if (device is not None and device.type == "cuda") else "gloo"
        kw = {"device_id": device} if backend == "nccl" and device is not None else {}
        dist.init_process_group(backend, **kw)
    return rank, world


def backend_name():
    """'nccl' (= RCCL over xGMI on ROCm), 'gloo', or 'none' when the job is a single process."""
    return dist.get_backend() if dist.is_initialized() else "none"


def shard_range(n_items, rank, world):
    """Contiguous shard [lo, hi) of n_items for ``rank``; sizes differ by at most one (ragged batches allowed)."""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_batch(x, rank, world):
    lo, hi = shard_range(x.shape[0], rank, world)
    return x[lo:hi]


def rank_seed(seed, rank):
    """Per-rank RNG seed for the quantisation noise (each rank must draw its own noise stream)."""
    return int(seed) + 7919 * int(rank)


def barrier():
    if dist.is_initialized():
        dist.barrier()


def _all_reduce(t, op):
    """dist.all_reduce; with the gloo backend (CPU rehearsals of the N > 1 path, also with ranks sharing one GPU) a device
    tensor is staged through the host, so the call does not depend on gloo having been built with GPU support."""
    if dist.get_backend() == "gloo" and t.device.type != "cpu":
        h = t.detach().cpu()
        dist.all_reduce(h, op=op)
        t.copy_(h)
    else:
        dist.all_reduce(t, op=op)
    return t


def max_over_ranks(value, device="cpu"):
    if not dist.is_initialized():
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    _all_reduce(t, dist.ReduceOp.MAX)
    return float(t)


def all_ranks_ok(ok, device="cpu"):
    """True iff ``ok`` holds on EVERY rank.  Call it at a point every rank reaches whatever happened before (outside the
    try block): a rank that failed locally must not leave its peers waiting in a collective it will never join."""
    if not dist.is_initialized():
        return bool(ok)
    t = torch.tensor([1.0 if ok else 0.0], dtype=torch.float64, device=device)
    _all_reduce(t, dist.ReduceOp.MIN)
    return bool(float(t) > 0.5)


def mean_over_ranks(values, device="cpu"):
    """Mean over ranks of a few host scalars (logged means, validation loss) -> list of floats, identical on every
    rank.  Decisions that steer training (ReduceLROnPlateau step, the D -> RD loss switch, is_best) must be taken on
    these, never on a rank's local value, or the replicas drift apart."""
    vals = [float(v) for v in values]
    if not dist.is_initialized():
        return vals
    t = torch.tensor(vals, dtype=torch.float64, device=device)
    _all_reduce(t, dist.ReduceOp.SUM)
    return (t / dist.get_world_size()).tolist()


def rank():
    return dist.get_rank() if dist.is_initialized() else int(os.environ.get("RANK", 0))


def is_rank0():
    return (not dist.is_initialized()) or dist.get_rank() == 0


def broadcast_parameters(module, src=0):
    """Make every rank's replica identical to rank ``src``'s (after a checkpoint load on rank 0, or at start-up)."""
    if dist.is_initialized():
        for t in list(module.parameters()) + list(module.buffers()):
            if t.numel():
                if dist.get_backend() == "gloo" and t.device.type != "cpu":
                    h = t.data.cpu()
                    dist.broadcast(h, src=src)
                    t.data.copy_(h)
                else:
                    dist.broadcast(t.data, src=src)


def sum_over_ranks(t):
    """In-place all-reduce(sum) of a tensor (scalars that are logged, or the flat gradient bucket)."""
    if dist.is_initialized():
        _all_reduce(t, dist.ReduceOp.SUM)
    return t


class FlatGradBucket:
    """One flat fp32 buffer aliasing every parameter's .grad, so the gradient exchange is a single all-reduce
    (7.2 M floats = 29 MB for the 1x1 auto-encoder model: ~0.4 ms on one xGMI ring)."""

    def __init__(self, params):
        self.params = [p for p in params if p.requires_grad]
        n = sum(p.numel() for p in self.params)
        dev = self.params[0].device if self.params else "cpu"
        self.flat = torch.zeros(n, dtype=torch.float32, device=dev)
        self.flat_p = None                   # parameter arena (relayout): every p.data a view into it
        self.group_views = {}                # param_arena: group key -> (parameter slice, gradient slice, numel per member)
        self.rejected = set()                # group keys that cannot be laid out (a member already belongs to another group)
        off = 0
        for p in self.params:
            p.grad = self.flat[off:off + p.numel()].view_as(p)
            off += p.numel()

    def relayout(self, groups):
        """Lay the given groups of parameters (lists; the members of a group have one shape) out ADJACENT and in order, in the
        gradient bucket and in a parameter arena (param_arena.py): a stack of a group is then a slice of the arena and its
        gradient a slice of the bucket.  Gradients and parameter values are carried over; parameters outside any group follow
        the groups in their old order.  A parameter that appears in several groups stays with the first."""
        mine = {id(p) for p in self.params}
        seen, order, kept = set(), [], []
        for g in groups:
            g = list(g)
            if not g or any(id(p) not in mine or id(p) in seen for p in g) or any(p.shape != g[0].shape for p in g):
                self.rejected.add(tuple(id(p) for p in g))          # never laid out: param_arena stops recording it
                continue
            seen.update(id(p) for p in g)
            order.extend(g)
            kept.append(g)
        order.extend(p for p in self.params if id(p) not in seen)
        n = sum(p.numel() for p in order)
        flat = torch.zeros(n, dtype=torch.float32, device=self.flat.device)
        flat_p = torch.empty(n, dtype=torch.float32, device=self.flat.device)
        off, where = 0, {}
        with torch.no_grad():
            for p in order:
                k = p.numel()
                flat[off:off + k].copy_(p.grad.reshape(-1))
                flat_p[off:off + k].copy_(p.data.reshape(-1))
                p.grad = flat[off:off + k].view_as(p)
                p.data = flat_p[off:off + k].view_as(p)
                where[id(p)] = off
                off += k
        self.params, self.flat, self.flat_p = order, flat, flat_p
        self.group_views = {}
        for g in kept:
            o, k = where[id(g[0])], g[0].numel()
            shape = (len(g),) + tuple(g[0].shape)
            self.group_views[tuple(id(p) for p in g)] = (flat_p[o:o + k * len(g)].view(shape), flat[o:o + k * len(g)].view(shape), k)
        self._kept_groups = kept

    def bump_version(self):
        """After the optimizer wrote the parameters through their own views: the arena's version counter (shared by every stack
        handed out) moves too, so that caches keyed on (data_ptr, _version) of a stacked tensor see the update."""
        if self.flat_p is not None:
            torch.autograd.graph.increment_version(self.flat_p)

    def zero_(self):
        self.flat.zero_()

    def all_reduce_mean(self):
        if dist.is_initialized():
            _all_reduce(self.flat, dist.ReduceOp.SUM)
            self.flat.div_(dist.get_world_size())
        return self.flat
