"""Stacks of per-plane parameters without a copy and without one gradient add per parameter.

The reference keeps one network per colour plane (graphs/models/LiftingBasedDWT_net.py:43-62); the HIP kernels take the
planes' parameters STACKED (one launch covers all planes).  With ordinary tensors every training step paid ~190 ``torch.stack``
copies of parameters and, in the backward, one ``grad.add_`` per parameter tensor (595 of them: the un-stacked slices added into
the flat gradient bucket) -- 4 ms of 4 us kernels.

Here the groups of parameters that are stacked together are laid out ADJACENT, in stack order, in two flat buffers owned by
``parallel.FlatGradBucket``: a parameter arena (every ``p.data`` is a view into it) and the gradient bucket (every ``p.grad`` is
a view into it, as before).  ``stack_leaf(params)`` then returns the group's slice of the parameter arena as a fresh autograd
LEAF whose ``.grad`` IS the group's slice of the bucket: no copy in the forward, one in-place accumulation per stack in the
backward, and the per-parameter ``.grad`` views (what Adam, the all-reduce and the tests read) see the result without any
further op.

The layout is learned from the first training step: a group that is not (or no longer -- ``.to()``, a foreign ``.data``
assignment) laid out falls back to ``torch.stack`` (differentiable, as before) and is recorded; the agent hands the recorded
groups to the bucket after the step (``FlatGradBucket.relayout``).  Data-parallel: the groups are found in program order, which
is the same on every rank, so every rank builds the same layout and the flat all-reduce stays element-for-element aligned.
"""
import collections

import torch

_active = None                               # the FlatGradBucket of the training step in progress
_pending = collections.OrderedDict()         # groups met without a layout: key -> list of parameters


def set_active(bucket):
    global _active
    _active = bucket


def take_pending():
    out = list(_pending.values())
    _pending.clear()
    return out


def stack_leaf(params):
    """``torch.stack(params, 0)`` for leaf parameters of one shape -- as a view of the arena when the group is laid out."""
    params = list(params)
    b = _active
    if b is not None and torch.is_grad_enabled():
        key = tuple(id(p) for p in params)
        hit = b.group_views.get(key)
        if hit is not None:
            pview, gview, n = hit
            # still aliased?  (first and last member: a module moved with .to() or re-assigned through .data breaks it)
            if params[0].data_ptr() == pview.data_ptr() and params[-1].data_ptr() == pview.data_ptr() + 4 * n * (len(params) - 1) \
                    and params[0].grad is not None and params[0].grad.data_ptr() == gview.data_ptr():
                s = pview.detach().requires_grad_(True)
                s.grad = gview
                return s
            b.group_views.pop(key, None)
        if key not in b.rejected and all(isinstance(p, torch.nn.Parameter) and p.requires_grad and p.is_cuda and
                                         p.dtype == torch.float32 and p.shape == params[0].shape for p in params):
            _pending[key] = params
    return torch.stack(params, 0)
