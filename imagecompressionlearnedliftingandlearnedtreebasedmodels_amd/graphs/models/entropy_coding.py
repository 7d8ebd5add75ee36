"""Real entropy coding of DWTConditioned2EntropyLayerZTsepSubbands -- replaces the per-pixel Python loops of the
reference's ``test`` / ``compress_ar`` / ``decompress_ar`` (graphs/models/LiftingBasedDWT_net.py:374-556).

What the reference does, per tensor: walk the pixels in raster order; at each pixel run the context CNN on a k x k crop of
the values decoded so far, read (sigma, mu) at the crop's centre, code ``round(y - mu)`` with the Gaussian CDF selected by
sigma (64-entry scale table), and write ``round(y - mu) + mu`` back for the pixels that follow.  That is O(pixels) tiny
sequential CNN calls -- minutes per image.

What this module does: the SAME per-pixel maths, scheduled as an anti-diagonal WAVEFRONT.  A pixel only depends on
already-coded neighbours inside the causal footprint of its context model, so every pixel with the same
``t = w + s*h`` is independent of the others once steps < t are done:
  * 3x3-crop stacks (``csc_xe`` and the coarsest ``csc_list`` entry, :311-317,299-305): the crop reaches (h-1, w+1) -> s = 2;
  * masked 5x5 type-A conv + tree context + cgp (:275-289): the mask reaches (h-1, w+2) -> s = 3.
One wavefront step is evaluated for all planes, images, subbands and pixels of the step in a handful of launches of the
kernels the rate path already has (conv engine on the batch of crops; fused cgp kernel with the context conv folded into
its first layer on the gathered taps), so a 256 x 256 subband needs ~1 000 steps instead of 65 536 x 3 Python iterations.
The tree-context conv (243 channels, the expensive part) does not depend on the tensor being coded and runs once, fully
parallel, per level.

Stream layout: one range-ANS stream per (plane, image, tensor) (the reference: per plane and tensor at batch 1), symbols
in WAVEFRONT order -- step t ascending, inside a step h ascending, channels innermost -- because that is the order a
parallel decoder can consume them in (the reference's raster order would serialise it again).  The coder itself is
ans.py (host C++ behind the C-ABI); symbols and CDF indexes are produced on the GPU.  Encoder and decoder evaluate the
context with the same kernels on the same operand shapes, so their (sigma, mu) agree bit for bit; not-yet-coded positions
hold 0 on both sides (the reference's encoder holds the unquantised value there, which the masks multiply by 0).
"""
import numpy as np
import torch

from ... import ops
from ...ans import BufferedRansEncoder, RansDecoder

_WF_CACHE = {}


def wavefront(H, W, slope, device):
    """-> (hs, ws, starts): pixel coordinates sorted by (t = w + slope*h, h) as device int64 tensors, and the python
    list of step boundaries (len = steps + 1)."""
    key = (H, W, slope, str(device))
    hit = _WF_CACHE.get(key)
    if hit is None:
        h = np.repeat(np.arange(H), W)
        w = np.tile(np.arange(W), H)
        t = w + slope * h
        order = np.lexsort((h, t))
        h, w, t = h[order], w[order], t[order]
        nsteps = W + slope * (H - 1)
        starts = np.searchsorted(t, np.arange(nsteps + 1)).tolist()
        hit = (torch.from_numpy(h).to(device), torch.from_numpy(w).to(device), starts)
        _WF_CACHE[key] = hit
    return hit


def _live_taps(mask_bits, K):
    return [(t // K, t % K) for t in range(K * K) if (mask_bits >> t) & 1]


class _Tables:
    """The Gaussian CDF tables of one entropy model as host int32 arrays (shared by every stream of a tensor)."""

    def __init__(self, emodel, scale_table):
        emodel.update_scale_table(scale_table)
        self.cdf = emodel.quantized_cdf.cpu().numpy().astype(np.int32)
        self.sizes = emodel.cdf_length.cpu().numpy().astype(np.int32)
        self.offsets = emodel.offset.cpu().numpy().astype(np.int32)


class _Sink:
    """Where the symbols of a tensor go (encoder) or come from (decoder), one stream per (plane, image)."""

    def __init__(self, P, B, tables, strings=None):
        self.P, self.B, self.t = P, B, tables
        self.decoding = strings is not None
        self._handles = None
        if self.decoding:
            self.dec = [[RansDecoder() for _ in range(B)] for _ in range(P)]
            for p in range(P):
                for b in range(B):
                    self.dec[p][b].set_stream(strings[p][b])
        else:
            self.sym, self.idx = [], []

    def step(self, idx, sym=None):
        """idx / sym: (P,B,N,g) int32 device tensors of one wavefront step.  Encoder: buffers them (device, no sync).
        Decoder: pops the step's symbols from the streams -> (P,B,N,g) int32 device tensor (one host round trip)."""
        if not self.decoding:
            self.idx.append(idx)
            self.sym.append(sym)
            return sym
        import ctypes as C
        from ... import _lib
        ih = np.ascontiguousarray(idx.cpu().numpy(), dtype=np.int32)
        out = np.empty(ih.shape, dtype=np.int32)
        if self._handles is None:
            self._handles = (C.c_void_p * (self.P * self.B))(*[self.dec[p][b]._h for p in range(self.P) for b in range(self.B)])
        per = int(ih.shape[2] * ih.shape[3])
        pv = lambda a_: a_.ctypes.data_as(C.c_void_p)
        ops.check(_lib.load().lldwt_rans_decode_multi(self._handles, self.P * self.B, pv(ih), per, per, pv(self.t.cdf),
                                                      self.t.cdf.shape[0], self.t.cdf.shape[1], pv(self.t.sizes),
                                                      pv(self.t.offsets), pv(out)), "rans_decode_multi")
        return torch.from_numpy(out).to(idx.device)

    def note(self, idx_host, sym_host):
        """Decoder of code_tree_level: called with every step's (indexes, symbols) as host int32 arrays once they are decoded.
        A hook for diagnostics (the code-length test replaces it); does nothing."""

    def flush(self):
        """Encoder: -> strings[p][b]."""
        idx = torch.cat(self.idx, 2).cpu().numpy()          # (P,B,Npix,g) in wavefront order
        sym = torch.cat(self.sym, 2).cpu().numpy()
        out = []
        for p in range(self.P):
            row = []
            for b in range(self.B):
                e = BufferedRansEncoder()
                e.encode_with_indexes(sym[p, b].reshape(-1), idx[p, b].reshape(-1), self.t.cdf, self.t.sizes, self.t.offsets)
                row.append(e.flush())
            out.append(row)
        return out


def _finish_step(emodel, sink, sigma, mu, yv):
    """sigma, mu (P,B,g,N); yv (P,B,g,N) or None when decoding -> dequantised values (P,B,g,N)."""
    idx = emodel.build_indexes(sigma).permute(0, 1, 3, 2).contiguous()                 # (P,B,N,g)
    sym = None
    if yv is not None:
        sym = torch.round(yv - mu).int().permute(0, 1, 3, 2).contiguous()              # quantize(..., "symbols", mu)
    sym = sink.step(idx, sym)
    return sym.permute(0, 1, 3, 2).float() + mu                                         # dequantize: symbol + mu


def code_crop_stack(seq_stack, emodels, seqs, y, shape, tables, strings=None):
    """The 3x3-crop context (LiftingBasedDWT_net.py:388-401 / :424-433 with compress_ar / decompress_ar at
    network_kernel_size 3).  y: (P,B,g,H,W) coefficients (encoder) or None (decoder); -> (strings or None, dequantised)."""
    P, B, g, H, W = shape
    dev = y.device if y is not None else next(seqs[0].parameters()).device
    hs, ws, starts = wavefront(H, W, 2, dev)
    sink = _Sink(P, B, tables, strings)
    yhat = torch.zeros(P, B, g, H + 2, W + 2, device=dev, dtype=torch.float32)       # 1-pixel zero frame = the crop padding
    ar = torch.arange(3, device=dev)
    for t in range(len(starts) - 1):
        a, b = starts[t], starts[t + 1]
        if a == b:
            continue
        h, w = hs[a:b], ws[a:b]
        N = b - a
        rows = (h[:, None, None] + ar[None, :, None]).expand(N, 3, 3)
        cols = (w[:, None, None] + ar[None, None, :]).expand(N, 3, 3)
        crop = yhat[:, :, :, rows, cols]                                              # (P,B,g,N,3,3)
        crop = crop.permute(0, 1, 3, 2, 4, 5).reshape(P, B * N, g, 3, 3).contiguous()
        ms = seq_stack(seqs, crop)[:, :, :, 1, 1].reshape(P, B, N, 2 * g)             # centre of every crop
        sigma = ms[..., 0::2].permute(0, 1, 3, 2)                                     # (P,B,g,N): even = sigma, odd = mu
        mu = ms[..., 1::2].permute(0, 1, 3, 2)
        yv = y[:, :, :, h, w] if y is not None else None
        yhat[:, :, :, h + 1, w + 1] = _finish_step(emodels[0], sink, sigma, mu, yv)
    out = yhat[:, :, :, 1:-1, 1:-1].contiguous()
    return (None if sink.decoding else sink.flush()), out


def _step_offsets(H, W, slope):
    """starts[t] = number of pixels in the wavefront steps before t (pixels of step t: rows ascending)."""
    h = np.repeat(np.arange(H), W)
    w = np.tile(np.arange(W), H)
    t = np.sort(w + slope * h)
    return np.searchsorted(t, np.arange(W + slope * (H - 1) + 1)).tolist()


def code_tree_level(emodels, plc, packed16, K, tap_bits, y, shape, tables, strings=None):
    """A level with tree context + masked KxK context + cgp (:402-417 / :440-454 with kernel size 5).  plc: (P,B,G*81,H,W)
    tree-context features of the (decoded) parent; packed16: the cgp stack with the masked conv folded into its first layer,
    packed for the register-chain kernel (_fold_csc_into_cgp).

    ONE launch per wavefront step (lldwt_cgp16_wavefront_step): the step's pixels are enumerated in the kernel, the causal
    taps are gathered from the decoded-so-far tensor, the cgp chain runs on the matrix cores and the epilogue produces CDF
    index, symbol and dequantised value -- no per-step tensor ops on the host side.  Encoder: the launches are queued back to
    back (no synchronisation until the streams are written).  Decoder: per step one launch, the step's indexes down to the
    host (pinned buffer), ONE C call that pops the step's symbols from every (plane, image) stream, the symbols up, one
    small launch that writes symbol + mu."""
    import ctypes as C
    from ... import _lib
    lib = _lib.load()
    P, B, G, H, W = shape
    dev = plc.device
    slope = K // 2 + 1
    starts = _step_offsets(H, W, slope)
    nsteps, ntot = len(starts) - 1, H * W
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    table63 = emodels[0].scale_table.to(dev).float()[:63].contiguous()
    yhat = torch.zeros(P, B, G, H, W, device=dev, dtype=torch.float32)
    sink = _Sink(P, B, tables, strings)
    ptr = lambda t_: C.c_void_p(t_.data_ptr())
    if not sink.decoding:
        yc = y.contiguous()
        idx_all = torch.empty(P, B, ntot, G, device=dev, dtype=torch.int32)
        sym_all = torch.empty(P, B, ntot, G, device=dev, dtype=torch.int32)
        for t in range(nsteps):
            ops.check(lib.lldwt_cgp16_wavefront_step(ptr(plc), ptr(yhat), ptr(yc), ptr(packed16), ptr(table63), ptr(idx_all),
                                                     ptr(sym_all), None, P, B, H, W, G, K, int(tap_bits), t, ntot, starts[t], st),
                      "cgp16_wavefront_step")
        sink.idx, sink.sym = [idx_all], [sym_all]
        return sink.flush(), yhat
    nmax = max(starts[t + 1] - starts[t] for t in range(nsteps))
    Z = P * B
    idx_d = torch.empty(Z * nmax * G, device=dev, dtype=torch.int32)
    sym_d = torch.empty(Z * nmax * G, device=dev, dtype=torch.int32)
    mu_d = torch.empty(Z * G * nmax, device=dev, dtype=torch.float32)
    idx_h = torch.empty(Z * nmax * G, dtype=torch.int32).pin_memory()
    sym_h = torch.empty(Z * nmax * G, dtype=torch.int32).pin_memory()
    handles = (C.c_void_p * Z)(*[sink.dec[p][b]._h for p in range(P) for b in range(B)])
    cdf, sizes, offs = tables.cdf, tables.sizes, tables.offsets
    pv = lambda a_: a_.ctypes.data_as(C.c_void_p)
    for t in range(nsteps):
        n = starts[t + 1] - starts[t]
        if n == 0:
            continue
        cnt = Z * n * G
        ops.check(lib.lldwt_cgp16_wavefront_step(ptr(plc), ptr(yhat), None, ptr(packed16), ptr(table63), ptr(idx_d), None, ptr(mu_d),
                                                 P, B, H, W, G, K, int(tap_bits), t, n, 0, st), "cgp16_wavefront_step")
        idx_h[:cnt].copy_(idx_d[:cnt], non_blocking=True)
        torch.cuda.current_stream().synchronize()
        ops.check(lib.lldwt_rans_decode_multi(handles, Z, C.c_void_p(idx_h.data_ptr()), n * G, n * G, pv(cdf), cdf.shape[0],
                                              cdf.shape[1], pv(sizes), pv(offs), C.c_void_p(sym_h.data_ptr())), "rans_decode_multi")
        sink.note(idx_h[:cnt].numpy(), sym_h[:cnt].numpy())
        sym_d[:cnt].copy_(sym_h[:cnt], non_blocking=True)
        ops.check(lib.lldwt_wavefront_apply(ptr(sym_d), ptr(mu_d), ptr(yhat), P, B, H, W, G, K, t, n, 0, st), "wavefront_apply")
    return None, yhat


def code_tree_level_generic(emodels, plc, cgp_packed, cgp_dims, K, tap_bits, y, shape, tables, strings=None):
    """A level with tree context + masked KxK context + cgp (:402-417 / :440-454 with kernel size 5).  plc: (P,B,G*81,H,W)
    tree-context features of the (decoded) parent; cgp_packed/dims: the cgp stack with the masked conv folded into its
    first layer (_fold_csc_into_cgp).  The host-stepped schedule (a handful of tensor ops and launches per step) for cgp widths
    other than the reference's 93 -> 162 -> 54 -> 18 -> 2, which the fused step kernel of code_tree_level is built for."""
    P, B, G, H, W = shape
    dev = plc.device
    R = K // 2
    taps = _live_taps(tap_bits, K)
    cpl = plc.shape[2] // G
    hs, ws, starts = wavefront(H, W, R + 1, dev)
    sink = _Sink(P, B, tables, strings)
    yhat = torch.zeros(P, B, G, H + 2 * R, W + 2 * R, device=dev, dtype=torch.float32)
    dy = torch.tensor([t[0] for t in taps], device=dev)
    dx = torch.tensor([t[1] for t in taps], device=dev)
    for t in range(len(starts) - 1):
        a, b = starts[t], starts[t + 1]
        if a == b:
            continue
        h, w = hs[a:b], ws[a:b]
        N = b - a
        ctx = yhat[:, :, :, h[None, :] + dy[:, None], w[None, :] + dx[:, None]]       # (P,B,G,ntaps,N): causal neighbours
        feat = plc[:, :, :, h, w].reshape(P, B, G, cpl, N)                            # (P,B,G,81,N)
        cat = torch.cat((feat, ctx), 3).reshape(P, B, G * (cpl + len(taps)), 1, N).contiguous()
        yv = y[:, :, :, h, w] if y is not None else None
        xarg = (yv if yv is not None else torch.zeros(P, B, G, N, device=dev)).reshape(P, B, G, 1, N).contiguous()
        _, params = ops.cgp_rate(cat, xarg, cgp_packed, cgp_dims, want_params=True)   # (P,B,2G,1,N)
        sigma, mu = params[:, :, 0::2, 0, :], params[:, :, 1::2, 0, :]
        yhat[:, :, :, h + R, w + R] = _finish_step(emodels[0], sink, sigma, mu, yv)
    out = yhat[:, :, :, R:-R, R:-R].contiguous()
    return (None if sink.decoding else sink.flush()), out


class _FactorizedTables:
    """The per-channel CDF tables of one EntropyBottleneck as host int32 arrays."""

    def __init__(self, emodel):
        emodel.update()
        self.cdf = emodel.quantized_cdf.cpu().numpy().astype(np.int32)
        self.sizes = emodel.cdf_length.cpu().numpy().astype(np.int32)
        self.offsets = emodel.offset.cpu().numpy().astype(np.int32)


def code_factorized(emodels, y, shape, strings=None):
    """A tensor coded with a factorized prior (EntropyBottleneck; compressai compress / decompress): every coefficient is
    independent, so symbols and indexes of the whole tensor are produced in one pass; one rANS stream per (plane, image),
    raster order, channels outermost.  emodels: one per plane.  -> (strings or None, dequantised (P,B,C,h,w))."""
    P, B, Cc, H, W = shape
    out = []
    strs = []
    for p in range(P):
        em = emodels[p]
        tabs = _FactorizedTables(em)
        if strings is None:
            sym, idx = em.symbols_and_indexes(y[p])                                     # (B,C,h,w)
            sh, ih = sym.cpu().numpy(), idx.cpu().numpy()
            row = []
            for b in range(B):
                e = BufferedRansEncoder()
                e.encode_with_indexes(sh[b].reshape(-1), ih[b].reshape(-1), tabs.cdf, tabs.sizes, tabs.offsets)
                row.append(e.flush())
            strs.append(row)
            out.append(em.dequantize_symbols(sym))
        else:
            dev = em.quantiles.device
            idx = np.broadcast_to(np.arange(Cc, dtype=np.int32).reshape(Cc, 1, 1), (Cc, H, W)).reshape(-1)
            syms = np.empty((B, Cc, H, W), dtype=np.int32)
            for b in range(B):
                d = RansDecoder()
                d.set_stream(strings[p][b])
                syms[b] = d.decode_stream(idx, tabs.cdf, tabs.sizes, tabs.offsets, as_numpy=True).reshape(Cc, H, W)
            out.append(em.dequantize_symbols(torch.from_numpy(syms).to(dev)))
    return (strs if strings is None else None), torch.stack(out, 0)


def code_gaussian_parallel(emodels, params, y, shape, tables, strings=None):
    """A level whose (sigma, mu) depend only on already decoded tensors (onlyEZWT: the tree context of the parent level,
    LiftingBasedDWT_net.py:822-835): fully parallel -- indexes and symbols of the whole tensor in one pass, raster order.
    params (P,B,2C,h,w): sigma on the even, mu on the odd channels.  -> (strings or None, dequantised (P,B,C,h,w))."""
    P, B, Cc, H, W = shape
    sigma, mu = params[:, :, 0::2].contiguous(), params[:, :, 1::2].contiguous()
    idx = emodels[0].build_indexes(sigma)                                               # (P,B,C,h,w) int32
    ih = idx.cpu().numpy()
    if strings is None:
        sym = torch.round(y - mu).int()
        sh = sym.cpu().numpy()
        strs = []
        for p in range(P):
            row = []
            for b in range(B):
                e = BufferedRansEncoder()
                e.encode_with_indexes(sh[p, b].reshape(-1), ih[p, b].reshape(-1), tables.cdf, tables.sizes, tables.offsets)
                row.append(e.flush())
            strs.append(row)
        return strs, sym.float() + mu
    syms = np.empty(ih.shape, dtype=np.int32)
    for p in range(P):
        for b in range(B):
            d = RansDecoder()
            d.set_stream(strings[p][b])
            syms[p, b] = d.decode_stream(ih[p, b].reshape(-1), tables.cdf, tables.sizes, tables.offsets,
                                         as_numpy=True).reshape(ih.shape[2:])
    return None, torch.from_numpy(syms).to(mu.device).float() + mu


def ideal_bits(sym, idx, tables):
    """Code length (bits) of integer symbols under the quantised tables; escapes are not modelled (returns their count)."""
    cdf, sizes, offs = tables.cdf, tables.sizes, tables.offsets
    v = sym - offs[idx]
    inside = (v >= 0) & (v < sizes[idx] - 2)
    vi = np.where(inside, v, 0)
    freq = cdf[idx, vi + 1] - cdf[idx, vi]
    return float(-np.log2(freq[inside] / 65536.0).sum()), int((~inside).sum())
