"""CPU: the C-ABI range-ANS coder and CDF builder vs the pure-Python oracle (bit-exact bytes), round trips, escapes,
and the scale-table index rule (SCALES_MIN / MAX / LEVELS, LiftingBasedDWT_net.py:12-14,32-33)."""
import math

import numpy as np
import pytest
import torch

from oracle import rans as orans
from imagecompressionlearnedliftingandlearnedtreebasedmodels_amd import ans


def _tables(seed=0, ncdf=5):
    g = np.random.default_rng(seed)
    cdfs, sizes, offs = [], [], []
    for i in range(ncdf):
        n = int(g.integers(3, 40))
        pmf = g.random(n).astype(np.float32) ** 3 + 1e-7
        pmf /= pmf.sum()
        c = orans.pmf_to_quantized_cdf(pmf.tolist())
        cdfs.append(c)
        sizes.append(len(c))
        offs.append(-int(g.integers(0, n)))
    return cdfs, sizes, offs


def test_pmf_to_quantized_cdf_matches_oracle_and_is_strictly_increasing():
    g = np.random.default_rng(3)
    for n in (2, 3, 17, 200, 1500):
        pmf = (g.random(n).astype(np.float32) ** 8)
        pmf[g.integers(0, n, n // 3)] = 0.0                      # zero-probability symbols must still get a slot
        pmf[0] = 1.0
        pmf /= pmf.sum()
        c = ans.pmf_to_quantized_cdf(pmf)
        assert c == orans.pmf_to_quantized_cdf(pmf.tolist())
        assert c[0] == 0 and c[-1] == 1 << 16 and all(b > a for a, b in zip(c, c[1:]))


def test_encode_bytes_equal_the_oracle_and_round_trip():
    cdfs, sizes, offs = _tables(1)
    g = np.random.default_rng(2)
    n = 3000
    idx = g.integers(0, len(cdfs), n).astype(np.int32)
    sym = np.array([int(g.integers(offs[i] - 2, offs[i] + sizes[i])) for i in idx], dtype=np.int32)   # some escape both ways
    sym[10] = 100000
    sym[11] = -77777                                             # long bypass runs (> 15 digits of count)
    enc = ans.BufferedRansEncoder()
    enc.encode_with_indexes(sym[:1000], idx[:1000], cdfs, sizes, offs)     # buffered in pieces, like compress_ar's extend()
    enc.encode_with_indexes(sym[1000:], idx[1000:], cdfs, sizes, offs)
    stream = enc.flush()
    assert stream == orans.encode(sym.tolist(), idx.tolist(), cdfs, sizes, offs)
    assert len(stream) % 4 == 0
    dec = ans.RansDecoder()
    dec.set_stream(stream)
    got = dec.decode_stream(idx[:5].tolist(), cdfs, sizes, offs) + dec.decode_stream(idx[5:], cdfs, sizes, offs)
    assert got == sym.tolist()
    assert orans.Decoder(stream).decode(idx.tolist(), cdfs, sizes, offs) == sym.tolist()
    # ideal code length under the tables vs bytes written: within 0.1 % + the 8-byte state
    bits = 0.0
    for s, i in zip(sym.tolist(), idx.tolist()):
        v = s - offs[i]
        if 0 <= v < sizes[i] - 2:
            bits += -math.log2((cdfs[i][v + 1] - cdfs[i][v]) / 65536.0)
    esc = sum(1 for s, i in zip(sym.tolist(), idx.tolist()) if not (0 <= s - offs[i] < sizes[i] - 2))
    assert 8 * len(stream) >= bits
    assert 8 * len(stream) <= bits * 1.001 + 64 + esc * 60


def test_different_tables_per_call_share_a_stream_and_decoder_cache_is_safe():
    """ADVICE r2.  compressai resolves each buffered symbol with the tables passed IN ITS call: two calls with different
    tables must give one stream that decodes chunk by chunk with the matching tables (bytes = the oracle coder run on the
    stacked tables).  The decoder's converted-table cache must not outlive set_stream, must not confuse a new list with a
    freed one of the same id, and must see an in-place edit of a list."""
    ta, tb = _tables(5, ncdf=4), _tables(6, ncdf=3)
    g = np.random.default_rng(9)

    def draw(t, n):
        idx = g.integers(0, len(t[0]), n).astype(np.int32)
        sym = np.array([int(g.integers(t[2][i], t[2][i] + t[1][i] - 2)) for i in idx], dtype=np.int32)
        return sym, idx
    sa, ia = draw(ta, 400)
    sb, ib = draw(tb, 300)
    enc = ans.BufferedRansEncoder()
    enc.encode_with_indexes(sa, ia, *ta)
    enc.encode_with_indexes(sb, ib, *tb)
    enc.encode_with_indexes(sa[:50], ia[:50], *ta)                 # the first tables again: no third copy
    stream = enc.flush()
    stacked = (ta[0] + tb[0], ta[1] + tb[1], ta[2] + tb[2])
    ref = orans.encode(sa.tolist() + sb.tolist() + sa[:50].tolist(),
                       ia.tolist() + (ib + len(ta[0])).tolist() + ia[:50].tolist(), *stacked)
    assert stream == ref
    dec = ans.RansDecoder()
    dec.set_stream(stream)
    assert dec.decode_stream(ia.tolist(), *ta) == sa.tolist()
    assert dec.decode_stream(ib.tolist(), *tb) == sb.tolist()      # fresh lists: converted again, not the cached first tables
    assert dec.decode_stream(ia[:50].tolist(), *ta) == sa[:50].tolist()
    with pytest.raises(Exception):
        enc.encode_with_indexes(sa, ia + 100, *ta)                 # an index outside this call's tables
    # numpy tables are cached by identity; a list edited in place between calls is seen
    na = tuple(np.asarray(x, dtype=np.int32) if not isinstance(x[0], list) else None for x in ta)
    width = max(len(r) for r in ta[0])
    m = np.zeros((len(ta[0]), width), dtype=np.int32)
    for i, r in enumerate(ta[0]):
        m[i, :len(r)] = r
    e2 = ans.RansEncoder()
    st2 = e2.encode_with_indexes(sa, ia, m, na[1], na[2])
    assert st2 == orans.encode(sa.tolist(), ia.tolist(), *ta)
    d2 = ans.RansDecoder()
    d2.set_stream(st2)
    assert d2.decode_stream(ia[:100], m, na[1], na[2]) == sa[:100].tolist()
    assert d2._tab_key is not None and d2._tab_key[0] is m
    assert d2.decode_stream(ia[100:], m, na[1], na[2]) == sa[100:].tolist()
    d2.set_stream(st2)
    assert d2._tab_key is None                                     # a new stream starts without cached tables
    cd = [list(r) for r in ta[0]]
    assert d2.decode_stream(ia[:10].tolist(), cd, ta[1], ta[2]) == sa[:10].tolist()
    assert d2._tab_key is None                                     # lists are never cached


def test_empty_and_corrupt_streams():
    cdfs, sizes, offs = _tables(4)
    e = ans.BufferedRansEncoder()
    e.encode_with_indexes([], [], cdfs, sizes, offs)
    s = e.flush()
    assert len(s) == 8                                           # just the flushed state
    d = ans.RansDecoder()
    d.set_stream(s)
    assert d.decode_stream([], cdfs, sizes, offs) == []
    from imagecompressionlearnedliftingandlearnedtreebasedmodels_amd._lib import LLDWTError
    with pytest.raises(LLDWTError):
        ans.RansDecoder().set_stream(b"\x00\x01")
    with pytest.raises(LLDWTError):
        ans.RansEncoder().encode_with_indexes([0], [99], cdfs, sizes, offs)     # cdf index out of range
    d.set_stream(s)
    with pytest.raises(LLDWTError):
        d.decode_stream([0] * 64, cdfs, sizes, offs)             # reads past the end


def test_scale_table_and_indexes():
    from imagecompressionlearnedliftingandlearnedtreebasedmodels_amd.entropy_models import GaussianConditional
    from imagecompressionlearnedliftingandlearnedtreebasedmodels_amd.graphs.models.LiftingBasedDWT_net import get_scale_table
    t = get_scale_table()
    assert torch.equal(t, orans.get_scale_table()) and len(t) == 64
    assert abs(float(t[0]) - 0.11) < 1e-6 and abs(float(t[-1]) - 256.0) < 1e-3
    gc = GaussianConditional(scale_table=None, scale_bound=0.11)
    gc.update_scale_table(t)
    cdf, ln, off = orans.gaussian_tables(t)
    assert torch.equal(gc.quantized_cdf.cpu(), cdf) and torch.equal(gc.cdf_length.cpu(), ln) and torch.equal(gc.offset.cpu(), off)
    # every row: increasing over its length, ends at 2^16, symmetric support around 0
    for i in range(64):
        row = cdf[i, :int(ln[i])].tolist()
        assert row[0] == 0 and row[-1] == 65536 and all(b > a for a, b in zip(row, row[1:]))
        assert int(ln[i]) == 2 * (-int(off[i])) + 1 + 2
    s = torch.tensor([0.0, 0.05, 0.11, 0.1100001, 0.5, 1.0, 3.7, 255.9, 256.0, 1000.0, float(t[17]), float(t[17]) * 1.0000001])
    idx = gc.build_indexes(s)
    assert torch.equal(idx.cpu(), orans.build_indexes(s, t))
    assert idx[0] == 0 and idx[2] == 0 and idx[-3] == 63 and idx[-2] == 17 and idx[-1] == 18
    # state_dict carries the tables under compressai's buffer names
    sd = gc.state_dict()
    assert {"_offset", "_quantized_cdf", "_cdf_length", "scale_table"} <= set(sd)
    gc2 = GaussianConditional(scale_table=None, scale_bound=0.11)
    gc2.load_state_dict(sd)
    assert torch.equal(gc2.quantized_cdf, gc.quantized_cdf)


def test_entropy_bottleneck_tables_cpu():
    """EntropyBottleneck.update() (compressai's algorithm; used by the onlyEZWT coding extension): per channel the table is a
    strictly increasing 16-bit CDF of pmf_length + 2 entries, offset = -ceil(median - lower quantile), and its frequencies
    are the density the ORACLE's eb_likelihood gives at the integers median + offset + k; a stream coded with these tables
    decodes to the same symbols."""
    from oracle import entropy as oent
    from imagecompressionlearnedliftingandlearnedtreebasedmodels_amd.entropy_models import EntropyBottleneck
    torch.manual_seed(3)
    eb = EntropyBottleneck(3)
    with torch.no_grad():
        eb.quantiles.copy_(torch.tensor([[[-7.3, 0.4, 6.1]], [[-2.2, -0.3, 3.9]], [[-11.0, 1.7, 9.5]]]))
        for i in range(5):
            getattr(eb, "_bias%d" % i).uniform_(-0.5, 0.5)
    assert eb.update() is True and eb.update() is False and eb.update(force=True) is True
    cdf, sizes, offs = eb.quantized_cdf.numpy(), eb.cdf_length.numpy(), eb.offset.numpy()
    sd = {"m." + k: v.detach() for k, v in eb.state_dict().items()}
    for c in range(3):
        q = eb.quantiles[c, 0].detach()
        minima, maxima = math.ceil(float(q[1] - q[0])), math.ceil(float(q[2] - q[1]))
        assert offs[c] == -minima and sizes[c] == minima + maxima + 1 + 2
        row = cdf[c, :sizes[c]]
        assert row[0] == 0 and row[-1] == 65536 and np.all(np.diff(row) > 0)
        n = sizes[c] - 2
        v = (float(q[1]) + offs[c] + torch.arange(n, dtype=torch.float32)).reshape(1, 1, n).expand(3, 1, n).contiguous()
        lik = oent.eb_likelihood(v, sd, "m.")[c, 0].numpy()
        # pmf_to_quantized_cdf renormalises (in-support mass + tail -> 2^16; these quantiles are not the density's own 1e-9
        # tails, so the mass outside the support is not negligible): compare the SHAPES
        f = np.diff(row)[:n].astype(np.float64)
        assert np.abs(f / f.sum() - lik / lik.sum()).max() < 3e-4
    g = np.random.default_rng(0)
    idx = g.integers(0, 3, 500).astype(np.int32)
    sym = np.array([g.integers(offs[i] - 2, offs[i] + sizes[i]) for i in idx], dtype=np.int32)     # some outside: escapes
    e = ans.BufferedRansEncoder()
    e.encode_with_indexes(sym, idx, cdf, sizes, offs)
    d = ans.RansDecoder()
    d.set_stream(e.flush())
    assert np.array_equal(d.decode_stream(idx, cdf, sizes, offs, as_numpy=True), sym)
