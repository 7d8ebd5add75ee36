"""Oracle: fixed CDF 9/7 (bior4.4) multi-level 2-D DWT, periodization mode -- test infrastructure only.

Restates what ``DWTPytorchWaveletsLayer`` (graphs/layers/lifting_dwt_nets.py:212-277) obtains from
``pytorch_wavelets.DWTForward(J, mode='periodization', wave='bior4.4')`` / ``DWTInverse``
(:228-231,250,274).  pytorch_wavelets is unpinned in requirements.txt:36 and absent from the image ->
**parity unpinned**; the filter bank equals the reference's own table ``get_cdf97_filters``
(lifting_dwt_nets.py:414-418) and the transform is checked against PyWavelets 1.1.1 (pywt.wavedec2,
mode='periodization') run in the build container (tests/golden/cdf97_pywt.npz).

Subband order: Yh[i][:, :, 0/1/2] = LH/HL/HH = pywt cH/cV/cD, finest level first.
"""
import torch
import torch.nn.functional as F

# analysis / synthesis filters, lifting_dwt_nets.py:415-418 (== pywt bior4.4 dec_lo, dec_hi, rec_lo, rec_hi)
DEC_LO = [0.0, 0.037828455507264, -0.023849465019557, -0.110624404418437, 0.377402855612831,
          0.852698679008894, 0.377402855612831, -0.110624404418437, -0.023849465019557, 0.037828455507264]
DEC_HI = [0.0, -0.064538882628697, 0.040689417609164, 0.418092273221617, -0.788485616405583,
          0.418092273221617, 0.040689417609164, -0.064538882628697, 0.0, 0.0]
REC_LO = [0.0, -0.064538882628697, -0.040689417609164, 0.418092273221617, 0.788485616405583,
          0.418092273221617, -0.040689417609164, -0.064538882628697, 0.0, 0.0]
REC_HI = [0.0, -0.037828455507264, -0.023849465019557, 0.110624404418437, 0.377402855612831,
          -0.852698679008894, 0.377402855612831, 0.110624404418437, -0.023849465019557, -0.037828455507264]
L = 10


def _afb1d(x, dim):
    """Analysis filter bank along ``dim`` (2 or 3), periodization: returns (lo, hi), each half length."""
    C = x.shape[1]
    N = x.shape[dim]
    assert N % 2 == 0
    h0 = torch.tensor(DEC_LO[::-1], dtype=x.dtype)
    h1 = torch.tensor(DEC_HI[::-1], dtype=x.dtype)
    shape = [1, 1, 1, 1]
    shape[dim] = L
    h = torch.cat([h0.reshape(shape), h1.reshape(shape)] * C, dim=0)
    x = torch.roll(x, -(L // 2), dims=dim)
    pad = (L - 1, 0) if dim == 2 else (0, L - 1)
    s = (2, 1) if dim == 2 else (1, 2)
    lohi = F.conv2d(x, h, padding=pad, stride=s, groups=C)
    N2, L2 = N // 2, L // 2
    if dim == 2:
        lohi[:, :, :L2] = lohi[:, :, :L2] + lohi[:, :, N2:N2 + L2]
        lohi = lohi[:, :, :N2]
    else:
        lohi[:, :, :, :L2] = lohi[:, :, :, :L2] + lohi[:, :, :, N2:N2 + L2]
        lohi = lohi[:, :, :, :N2]
    return lohi[:, 0::2], lohi[:, 1::2]


def _sfb1d(lo, hi, dim):
    C = lo.shape[1]
    N = 2 * lo.shape[dim]
    g0 = torch.tensor(REC_LO, dtype=lo.dtype)
    g1 = torch.tensor(REC_HI, dtype=lo.dtype)
    shape = [1, 1, 1, 1]
    shape[dim] = L
    g0 = torch.cat([g0.reshape(shape)] * C, dim=0)
    g1 = torch.cat([g1.reshape(shape)] * C, dim=0)
    s = (2, 1) if dim == 2 else (1, 2)
    y = F.conv_transpose2d(lo, g0, stride=s, groups=C) + F.conv_transpose2d(hi, g1, stride=s, groups=C)
    if dim == 2:
        y[:, :, :L - 2] = y[:, :, :L - 2] + y[:, :, N:N + L - 2]
        y = y[:, :, :N]
    else:
        y[:, :, :, :L - 2] = y[:, :, :, :L - 2] + y[:, :, :, N:N + L - 2]
        y = y[:, :, :, :N]
    return torch.roll(y, 1 - L // 2, dims=dim)


def _afb1d_periodic(x, dim):
    """The periodic analysis itself, for any even length: lo[k] = sum_m dec_lo[m] x[(2k + 5 - m) mod N] (same for hi).
    Equals ``_afb1d`` whenever N >= L = 10; for the shorter level inputs (2, 4, 6, 8 samples) ``_afb1d``'s single fold of the
    linear convolution drops the taps that wrap more than one period (what pytorch_wavelets' afb1d computes, restated from its
    source; library absent -> unpinned), whereas PyWavelets' periodization and the HIP kernels use the formula here
    (tests/golden/cdf97_pywt_small.npz)."""
    N = x.shape[dim]
    assert N % 2 == 0
    k = torch.arange(N // 2)
    lo = hi = 0
    for m in range(L):
        v = x.index_select(dim, (2 * k + 5 - m) % N)
        lo = lo + DEC_LO[m] * v
        hi = hi + DEC_HI[m] * v
    return lo, hi


def _sfb1d_periodic(lo, hi, dim):
    """x[n] = sum over taps t with (n + 4 - t) mod N even of lo[q/2] rec_lo[t] + hi[q/2] rec_hi[t], q = (n + 4 - t) mod N."""
    N = 2 * lo.shape[dim]
    n = torch.arange(N)
    y = 0
    for t in range(L):
        q = (n + 4 - t) % N
        even = (q % 2 == 0).to(lo.dtype)
        shape = [1, 1, 1, 1]
        shape[dim] = N
        y = y + even.reshape(shape) * (REC_LO[t] * lo.index_select(dim, q // 2) + REC_HI[t] * hi.index_select(dim, q // 2))
    return y


def dwt_forward(x, levels, periodic=False):
    """-> (Yl, [Yh_0..]) with Yh_i (B, C, 3, h, w) = (LH, HL, HH), finest first.  periodic=True: the exact periodic
    transform at every length (see _afb1d_periodic); the two agree unless a level input is shorter than 10 samples."""
    afb = _afb1d_periodic if periodic else _afb1d
    Yh = []
    ll = x
    for _ in range(levels):
        lo_w, hi_w = afb(ll, 3)
        ll_, lh = afb(lo_w, 2)         # low along width:  (lo_h, hi_h) -> LL, LH
        hl, hh = afb(hi_w, 2)          # high along width: HL, HH
        Yh.append(torch.stack((lh, hl, hh), dim=2))
        ll = ll_
    return ll, Yh


def dwt_inverse(Yl, Yh, periodic=False):
    sfb = _sfb1d_periodic if periodic else _sfb1d
    ll = Yl
    for h in Yh[::-1]:
        lh, hl, hh = h[:, :, 0], h[:, :, 1], h[:, :, 2]
        lo = sfb(ll, lh, 2)
        hi = sfb(hl, hh, 2)
        ll = sfb(lo, hi, 3)
    return ll
