"""Strip tiler for large frames (BASELINE configs[4]: one 3840x2160 frame as 8 vertical strips of 480x2160).

The reference has no tiler: its model is fully convolutional and an image is scaled only by cropping
(dataloaders/image_dl.py:80), so a strip is exactly "the reference run on that crop" (zero-padded borders).
Strips are therefore independent images: they go through the path as a batch (one launch covers all strips), or one
strip per rank under ``parallel.shard_range`` -- no halo exchange, no collective except the final sum of bit counts.
"""
import torch


def split_strips(frame, n_strips):
    """(B,3,H,W) -> (B*n_strips,3,H,W/n_strips): vertical strips, strip-major inside each image (b*n + s).
    W/n_strips must keep the strip width divisible by 2**dwtlevels (checked by the transform's own shape asserts)."""
    B, C, H, W = frame.shape
    if W % n_strips:
        raise ValueError("frame width %d is not divisible into %d strips" % (W, n_strips))
    ws = W // n_strips
    return frame.reshape(B, C, H, n_strips, ws).permute(0, 3, 1, 2, 4).reshape(B * n_strips, C, H, ws).contiguous()


def merge_strips(strips, n_strips):
    """Inverse of split_strips: (B*n,3,H,ws) -> (B,3,H,n*ws)."""
    Bn, C, H, ws = strips.shape
    B = Bn // n_strips
    return strips.reshape(B, n_strips, C, H, ws).permute(0, 2, 3, 1, 4).reshape(B, C, H, n_strips * ws).contiguous()


def strips_for_rank(frame, n_strips, rank, world):
    """The strips this rank owns (contiguous shard of the strip batch, parallel.shard_range)."""
    from . import parallel
    s = split_strips(frame, n_strips)
    lo, hi = parallel.shard_range(s.shape[0], rank, world)
    return s[lo:hi].contiguous(), (lo, hi)


def frame_rate_bits(net, frame, n_strips):
    """Estimated bits of a frame coded as independent strips: -> (total_bits float64 tensor (1,), per-strip bits (n,)).
    net: LiftingBasedDWTNetWrapper (clrch == 1) in eval mode."""
    from . import ops
    from .graphs.models.LiftingBasedDWT_net import rate_planes
    s = split_strips(frame, n_strips)
    with torch.no_grad():
        y = ops.rgb_to_ycc(s)
        si_xe, si_xo = rate_planes(net.nets(), y, False)
    per = si_xe.double().sum(dim=(0, 2, 3, 4))
    for t in si_xo:
        per = per + t.double().sum(dim=(0, 2, 3, 4))
    return per.sum().reshape(1), per
