"""Per-module cache of stacked / packed device parameters (MFMA operand packs, folded cgp weights, EntropyBottleneck
tables) that the kernels consume.

The cache lives ON the owning ``nn.Module`` (a plain attribute, not a parameter/buffer), so it dies with the module and
can never be handed to another model that happens to reuse an ``id()`` or a device pointer.  An entry is rebuilt when
one of its source tensors changed identity (``data_ptr``) or version (``_version``: bumped by every in-place autograd-
visible write, e.g. ``optimizer.step()``, ``copy_`` under ``no_grad``).

Writes THROUGH ``.data`` (``w.data.mul_()``, ``p.data = ...``) do not bump ``_version``: code that does this must call
``invalidate_packed(model)`` afterwards.  The library's own ``.data`` writer, ``MaskedConv2d.apply_mask_``, does; so do
``load_state_dict`` (post-hook) and ``_apply`` (``.to()``, ``.cuda()``, ``.float()``) of every module built on
``PackedOwnerMixin``.
"""
_ATTR = "_lldwt_packed"


def cached(owner, tag, tensors, build):
    """Value of ``build()`` cached on ``owner`` under ``tag`` until one of ``tensors`` changes."""
    store = owner.__dict__.get(_ATTR)
    if store is None:
        store = owner.__dict__[_ATTR] = {}
    key = tuple((t.data_ptr(), t._version) for t in tensors)
    hit = store.get(tag)
    if hit is not None and hit[0] == key:
        return hit[1]
    val = build()
    store[tag] = (key, val)
    return val


def invalidate_packed(module):
    """Drop every cached pack owned by ``module`` or one of its submodules (call after writing parameters via .data)."""
    for m in module.modules():
        m.__dict__.pop(_ATTR, None)
        if "_masked_version" in m.__dict__:
            m.__dict__["_masked_version"] = None


class PackedOwnerMixin:
    """nn.Module mixin: invalidate the packs on load_state_dict and on _apply (device / dtype moves)."""

    def _init_packed_owner(self):
        self.register_load_state_dict_post_hook(lambda module, incompatible_keys: invalidate_packed(module))

    def _apply(self, fn, *args, **kw):
        out = super()._apply(fn, *args, **kw)
        invalidate_packed(self)
        return out
