"""Oracle: subband auto-encoders, GDN, LowerBound, NonNegativeParametrizer -- test infrastructure only.

Follows (paths relative to /root/reference):
  utils/bound_ops.py:22-65          LowerBound (fwd max, bwd pass-through rule)
  utils/parametrizers.py:23-48      NonNegativeParametrizer
  graphs/layers/gdn.py:41-92        GDN (vendored copy of compressai.layers.GDN)
  graphs/layers/lifting_dwt_nets.py:82-124   SubbandAutoEncoder (grouped 1x1 scalar MLP)
  graphs/layers/lifting_dwt_nets.py:126-164  SubbandAutoEncoderBerk (3x3 convs + GDN)
"""
import torch
import torch.nn.functional as F


class _LowerBoundFn(torch.autograd.Function):
    """utils/bound_ops.py:22-44."""

    @staticmethod
    def forward(ctx, x, bound):
        ctx.save_for_backward(x, bound)
        return torch.max(x, bound)

    @staticmethod
    def backward(ctx, g):
        x, bound = ctx.saved_tensors
        return ((x >= bound) | (g < 0)) * g, None


def lower_bound(x, bound):
    b = torch.tensor([float(bound)], dtype=x.dtype)
    return _LowerBoundFn.apply(x, b)


def lower_bound_bwd(x, bound, g):
    """utils/bound_ops.py:26-28."""
    return ((x >= bound) | (g < 0)) * g


REPARAM_OFFSET = 2.0 ** -18


def nonneg_bound(minimum=0.0):
    """utils/parametrizers.py:37-40: bound = sqrt(minimum + pedestal), both stored as fp32 buffers."""
    pedestal = REPARAM_OFFSET ** 2
    return (float(minimum) + pedestal) ** 0.5, pedestal


def nonneg_param(x, minimum=0.0):
    """utils/parametrizers.py:45-48: lower_bound(x)**2 - pedestal."""
    bound, pedestal = nonneg_bound(minimum)
    out = lower_bound(x, bound)
    return out ** 2 - torch.tensor([pedestal], dtype=x.dtype)


def nonneg_init(x):
    """utils/parametrizers.py:42-43."""
    ped = torch.tensor([REPARAM_OFFSET ** 2], dtype=x.dtype)
    return torch.sqrt(torch.max(x + ped, ped))


def gdn(x, beta, gamma, inverse=False, beta_min=1e-6):
    """graphs/layers/gdn.py:77-92."""
    C = x.shape[1]
    b = nonneg_param(beta, beta_min)
    g = nonneg_param(gamma, 0.0).reshape(C, C, 1, 1)
    norm = F.conv2d(x ** 2, g, b)
    norm = torch.sqrt(norm) if inverse else torch.rsqrt(norm)
    return x * norm


def subband_ae_encode(x, sd, prefix):
    """SubbandAutoEncoder.encode (lifting_dwt_nets.py:99-104,112-117): Sequential indices 0,2,4,6 are convs."""
    iC = x.shape[1]
    t = x
    for n in (0, 2, 4, 6):
        t = F.conv2d(t, sd[prefix + "ae_down.%d.weight" % n], sd[prefix + "ae_down.%d.bias" % n], groups=iC)
        if n != 6:
            t = torch.tanh(t)
    return t


def subband_ae_decode(y, sd, prefix):
    """SubbandAutoEncoder.decode (lifting_dwt_nets.py:105-110,119-124): grouped 1x1 ConvTranspose2d."""
    iC = y.shape[1]
    t = y
    for n in (0, 2, 4, 6):
        t = F.conv_transpose2d(t, sd[prefix + "ae_up.%d.weight" % n], sd[prefix + "ae_up.%d.bias" % n], groups=iC)
        if n != 6:
            t = torch.tanh(t)
    return t


def berk_ae_encode(x, sd, prefix):
    """SubbandAutoEncoderBerk.encode (lifting_dwt_nets.py:139-144): conv3x3 / GDN alternating (0,1,2,3,4,5,6)."""
    t = x
    for n in (0, 2, 4, 6):
        t = F.conv2d(t, sd[prefix + "ae_down.%d.weight" % n], sd[prefix + "ae_down.%d.bias" % n], padding=1)
        if n != 6:
            t = gdn(t, sd[prefix + "ae_down.%d.beta" % (n + 1)], sd[prefix + "ae_down.%d.gamma" % (n + 1)], False)
    return t


def berk_ae_decode(y, sd, prefix):
    """SubbandAutoEncoderBerk.decode (lifting_dwt_nets.py:145-150)."""
    t = y
    for n in (0, 2, 4, 6):
        t = F.conv_transpose2d(t, sd[prefix + "ae_up.%d.weight" % n], sd[prefix + "ae_up.%d.bias" % n], padding=1)
        if n != 6:
            t = gdn(t, sd[prefix + "ae_up.%d.beta" % (n + 1)], sd[prefix + "ae_up.%d.gamma" % (n + 1)], True)
    return t


def ae_encode(x, sd, prefix, kind):
    return subband_ae_encode(x, sd, prefix) if kind == "SubbandAutoEncoder" else berk_ae_encode(x, sd, prefix)


def ae_decode(y, sd, prefix, kind):
    return subband_ae_decode(y, sd, prefix) if kind == "SubbandAutoEncoder" else berk_ae_decode(y, sd, prefix)
