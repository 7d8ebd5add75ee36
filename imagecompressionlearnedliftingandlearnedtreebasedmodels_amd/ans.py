"""Range-ANS entropy coder -- the two classes of ``compressai.ans`` the reference uses (``BufferedRansEncoder``,
``RansDecoder``; graphs/models/LiftingBasedDWT_net.py:9,466,502-505,516-517,540-546) over the C-ABI host functions
``lldwt_rans_*`` (include/lldwt.h, csrc/rans.hip).  Same method names and argument order as compressai's; symbols,
indexes and tables may be Python lists (as the reference passes them) or int32 tensors / numpy arrays (no per-symbol
Python objects: that is how the wavefront coder calls it).
"""
import ctypes as C

import numpy as np
import torch

from . import _lib


def _i32(a):
    if isinstance(a, torch.Tensor):
        a = a.detach().cpu().numpy()
    return np.ascontiguousarray(np.asarray(a, dtype=np.int32))


def _tables(cdfs, cdfs_sizes, offsets):
    if isinstance(cdfs, (list, tuple)):                 # ragged python lists -> padded matrix
        n = max(len(r) for r in cdfs)
        m = np.zeros((len(cdfs), n), dtype=np.int32)
        for i, r in enumerate(cdfs):
            m[i, :len(r)] = r
        cdfs = m
    cdfs = _i32(cdfs)
    if cdfs.ndim != 2:
        raise _lib.LLDWTError("cdf table must be 2-D (ncdf, max_length)")
    return cdfs, _i32(cdfs_sizes), _i32(offsets)


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


class BufferedRansEncoder:
    """compressai.ans.BufferedRansEncoder: buffer symbols with ``encode_with_indexes`` (may be called repeatedly),
    ``flush()`` runs the rANS state machine over them in reverse and returns the byte string.  Like compressai's, every
    call resolves its symbols against the tables passed IN THAT CALL: calls with different tables may share a stream
    (the distinct tables are stacked at flush and each call's indexes shifted to its table's rows)."""

    def __init__(self):
        self._sym, self._idx, self._tabs, self._rows = [], [], [], []

    def _table_slot(self, tab):
        for k, t in enumerate(self._tabs):
            if all(a.shape == b.shape and np.array_equal(a, b) for a, b in zip(t, tab)):
                return k
        self._tabs.append(tab)
        self._rows.append(sum(t[0].shape[0] for t in self._tabs[:-1]))
        return len(self._tabs) - 1

    def encode_with_indexes(self, symbols, indexes, cdfs, cdfs_sizes, offsets):
        s, i = _i32(symbols).reshape(-1), _i32(indexes).reshape(-1)
        if s.shape != i.shape:
            raise _lib.LLDWTError("symbols and indexes differ in length")
        tab = _tables(cdfs, cdfs_sizes, offsets)
        if i.size and (int(i.min()) < 0 or int(i.max()) >= tab[0].shape[0]):
            raise _lib.LLDWTError("index outside the %d tables of this call" % tab[0].shape[0])
        k = self._table_slot(tab)
        self._sym.append(s)
        self._idx.append(i + self._rows[k] if self._rows[k] else i)

    def flush(self):
        if not self._tabs:
            return b""
        s, i = np.concatenate(self._sym), np.ascontiguousarray(np.concatenate(self._idx), dtype=np.int32)
        if len(self._tabs) == 1:
            cdfs, sizes, offs = self._tabs[0]
        else:
            width = max(t[0].shape[1] for t in self._tabs)
            cdfs = np.zeros((sum(t[0].shape[0] for t in self._tabs), width), dtype=np.int32)
            r = 0
            for t in self._tabs:
                cdfs[r:r + t[0].shape[0], :t[0].shape[1]] = t[0]
                r += t[0].shape[0]
            sizes = np.ascontiguousarray(np.concatenate([t[1] for t in self._tabs]), dtype=np.int32)
            offs = np.ascontiguousarray(np.concatenate([t[2] for t in self._tabs]), dtype=np.int32)
        cap = 16 * s.size + 64                            # worst case: every symbol escapes with long bypass runs
        out = np.empty(cap, dtype=np.uint8)
        n = _lib.load().lldwt_rans_encode(_p(s), _p(i), s.size, _p(cdfs), cdfs.shape[0], cdfs.shape[1], _p(sizes), _p(offs),
                                          _p(out), cap)
        if n < 0:
            _lib.check(int(n), "rans_encode")
        self._sym, self._idx, self._tabs, self._rows = [], [], [], []
        return out[:n].tobytes()


class RansEncoder:
    """compressai.ans.RansEncoder: one-shot form."""

    def encode_with_indexes(self, symbols, indexes, cdfs, cdfs_sizes, offsets):
        e = BufferedRansEncoder()
        e.encode_with_indexes(symbols, indexes, cdfs, cdfs_sizes, offsets)
        return e.flush()


class RansDecoder:
    """compressai.ans.RansDecoder: ``set_stream(bytes)`` then ``decode_stream(indexes, ...)`` pops symbols in coding order."""

    def __init__(self):
        self._h = None
        self._tab_key, self._tab = None, None

    def set_stream(self, stream):
        self._free()
        self._tab_key, self._tab = None, None
        buf = np.frombuffer(stream, dtype=np.uint8)
        h = _lib.load().lldwt_rans_decoder_new(_p(buf), buf.size)
        if not h:
            _lib.check(-1, "rans_decoder_new")
        self._h = C.c_void_p(h)

    def decode_stream(self, indexes, cdfs, cdfs_sizes, offsets, as_numpy=False):
        if self._h is None:
            raise _lib.LLDWTError("RansDecoder: set_stream first")
        # the wavefront coder passes the same three arrays for every call of a stream: convert them once.  The cache holds
        # the objects themselves (compared with ``is``: an id() can be reused by a new object once the old one is freed) and
        # only immutable-by-convention containers; python lists (the reference's call pattern passes fresh
        # ``quantized_cdf.tolist()`` lists, which may also be mutated in place) are converted on every call
        cacheable = all(isinstance(t, (np.ndarray, torch.Tensor)) for t in (cdfs, cdfs_sizes, offsets))
        key = (cdfs, cdfs_sizes, offsets)
        if cacheable and self._tab_key is not None and all(a is b for a, b in zip(key, self._tab_key)):
            cdf, sizes, offs = self._tab
        else:
            cdf, sizes, offs = _tables(cdfs, cdfs_sizes, offsets)
            self._tab_key, self._tab = (key, (cdf, sizes, offs)) if cacheable else (None, None)
        i = _i32(indexes).reshape(-1)
        out = np.empty(i.size, dtype=np.int32)
        _lib.check(_lib.load().lldwt_rans_decode(self._h, _p(i), i.size, _p(cdf), cdf.shape[0], cdf.shape[1], _p(sizes),
                                                _p(offs), _p(out)), "rans_decode")
        return out if as_numpy else out.tolist()

    def _free(self):
        if self._h is not None:
            _lib.load().lldwt_rans_decoder_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self._free()
        except Exception:
            pass


def pmf_to_quantized_cdf(pmf, precision=16):
    """compressai._CXX.pmf_to_quantized_cdf: list / 1-D tensor of float -> list of len(pmf)+1 ints."""
    a = np.ascontiguousarray(np.asarray(pmf.detach().cpu().numpy() if isinstance(pmf, torch.Tensor) else pmf, dtype=np.float32))
    out = np.empty(a.size + 1, dtype=np.uint32)
    _lib.check(_lib.load().lldwt_pmf_to_quantized_cdf(_p(a), a.size, precision, _p(out)), "pmf_to_quantized_cdf")
    return out.astype(np.int64).tolist()
