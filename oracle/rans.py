"""Oracle: range-ANS coder + CDF tables of the reference's real-coding path -- pure Python, test infrastructure only.

Follows (the packages are absent from the image -> restated from the published algorithms; **parity unpinned** against
compressai's own bytes):
  compressai==1.2.1  compressai/cpp_exts/rans/rans_interface.cpp   BufferedRansEncoder.encode_with_indexes / flush,
                     RansDecoder.decode_stream (precision 16, bypass precision 4)  -- call sites
                     graphs/models/LiftingBasedDWT_net.py:466,502-505,516-517,540-546
  ryg_rans rans64.h (public domain)                                 Rans64EncPut / EncFlush / DecInit / DecGet / DecAdvance
  compressai/cpp_exts/ops/ops.cpp                                   pmf_to_quantized_cdf
  compressai/entropy_models/entropy_models.py                       GaussianConditional.update / build_indexes /
                                                                    _standardized_quantile, EntropyModel._pmf_to_cdf
  graphs/models/LiftingBasedDWT_net.py:12-14,32-33                  SCALES_MIN / SCALES_MAX / SCALES_LEVELS, get_scale_table
"""
import math
import struct

import torch

PRECISION = 16
BYPASS_PRECISION = 4
MAX_BYPASS = (1 << BYPASS_PRECISION) - 1
RANS_L = 1 << 31
M64 = (1 << 64) - 1

SCALES_MIN, SCALES_MAX, SCALES_LEVELS = 0.11, 256, 64


def get_scale_table(lo=SCALES_MIN, hi=SCALES_MAX, levels=SCALES_LEVELS):
    """LiftingBasedDWT_net.py:32-33."""
    return torch.exp(torch.linspace(math.log(lo), math.log(hi), levels))


def pmf_to_quantized_cdf(pmf, precision=PRECISION):
    """compressai ops.cpp pmf_to_quantized_cdf: list of float -> list of n+1 ints."""
    # std::round (half away from zero; p >= 0) of the FLOAT32 product p * 2^precision (exact: a power-of-two scaling)
    cdf = [0] + [int(math.floor(float(p) * (1 << precision) + 0.5)) for p in pmf]
    total = sum(cdf)
    cdf = [((1 << precision) * c) // total for c in cdf]
    for i in range(1, len(cdf)):
        cdf[i] += cdf[i - 1]
    cdf[-1] = 1 << precision
    n = len(cdf) - 1
    for i in range(n):
        if cdf[i] == cdf[i + 1]:
            best_freq, best = None, -1
            for j in range(n):
                f = cdf[j + 1] - cdf[j]
                if f > 1 and (best_freq is None or f < best_freq):
                    best_freq, best = f, j
            assert best != -1
            if best < i:
                for j in range(best + 1, i + 1):
                    cdf[j] -= 1
            else:
                for j in range(i + 1, best + 1):
                    cdf[j] += 1
    return cdf


def std_cumulative(z):
    return 0.5 * torch.erfc(-(2 ** -0.5) * z)


def gaussian_tables(scale_table, tail_mass=1e-9):
    """GaussianConditional.update(): -> (quantized_cdf (n, max_len+2) int32, cdf_length (n,), offset (n,))."""
    from scipy.stats import norm
    multiplier = -float(norm.ppf(tail_mass / 2))
    pmf_center = torch.ceil(scale_table * multiplier).int()
    pmf_length = 2 * pmf_center + 1
    max_length = int(pmf_length.max())
    samples = torch.abs(torch.arange(max_length).int() - pmf_center[:, None]).float()
    scale = scale_table.unsqueeze(1).float()
    upper = std_cumulative((0.5 - samples) / scale)
    lower = std_cumulative((-0.5 - samples) / scale)
    pmf = upper - lower
    tail = 2 * lower[:, :1]
    cdf = torch.zeros(len(pmf_length), max_length + 2, dtype=torch.int32)
    for i in range(len(pmf_length)):
        prob = torch.cat((pmf[i, :int(pmf_length[i])], tail[i]), 0)
        c = pmf_to_quantized_cdf(prob.tolist())
        cdf[i, :len(c)] = torch.tensor(c, dtype=torch.int32)
    return cdf, (pmf_length + 2).int(), (-pmf_center).int()


def build_indexes(scales, scale_table, scale_bound=0.11):
    """GaussianConditional.build_indexes: index of the first table entry >= max(scale, bound)."""
    s = torch.clamp(scales, min=scale_bound)          # lower_bound_scale
    idx = torch.full(s.shape, len(scale_table) - 1, dtype=torch.int32)
    for t in scale_table[:-1]:
        idx -= (s <= t).int()
    return idx


# ----------------------------------------------------------------------------- rANS (rans64 + compressai interface)
def _expand(symbols, indexes, cdfs, cdf_sizes, offsets):
    out = []
    for sym, ci in zip(symbols, indexes):
        cdf = cdfs[ci]
        max_value = cdf_sizes[ci] - 2
        value = sym - offsets[ci]
        raw = 0
        if value < 0:
            raw = -2 * value - 1
            value = max_value
        elif value >= max_value:
            raw = 2 * (value - max_value)
            value = max_value
        out.append((cdf[value], cdf[value + 1] - cdf[value], False))
        if value == max_value:
            nb = 0
            while (raw >> (nb * BYPASS_PRECISION)) != 0:
                nb += 1
            val = nb
            while val >= MAX_BYPASS:
                out.append((MAX_BYPASS, MAX_BYPASS + 1, True))
                val -= MAX_BYPASS
            out.append((val, val + 1, True))
            for j in range(nb):
                v = (raw >> (j * BYPASS_PRECISION)) & MAX_BYPASS
                out.append((v, v + 1, True))
    return out


def encode(symbols, indexes, cdfs, cdf_sizes, offsets):
    """-> bytes.  symbols / indexes: lists of int; cdfs: list of lists."""
    syms = _expand(symbols, indexes, cdfs, cdf_sizes, offsets)
    words = []
    x = RANS_L
    for start, rng, bypass in reversed(syms):
        if not bypass:
            x_max = ((RANS_L >> PRECISION) << 32) * rng
            if x >= x_max:
                words.append(x & 0xFFFFFFFF)
                x >>= 32
            x = ((x // rng) << PRECISION) + (x % rng) + start
        else:
            freq = 1 << (16 - BYPASS_PRECISION)
            x_max = ((RANS_L >> 16) << 32) * freq
            if x >= x_max:
                words.append(x & 0xFFFFFFFF)
                x >>= 32
            x = ((x << BYPASS_PRECISION) | start) & M64
    words.append((x >> 32) & 0xFFFFFFFF)
    words.append(x & 0xFFFFFFFF)
    words.reverse()
    return struct.pack("<%dI" % len(words), *words)


class Decoder:
    def __init__(self, stream):
        self.words = list(struct.unpack("<%dI" % (len(stream) // 4), stream))
        self.pos = 2
        self.x = self.words[0] | (self.words[1] << 32)

    def _renorm(self):
        if self.x < RANS_L:
            self.x = (self.x << 32) | self.words[self.pos]
            self.pos += 1

    def _bits(self, n):
        v = self.x & ((1 << n) - 1)
        self.x >>= n
        self._renorm()
        return v

    def decode(self, indexes, cdfs, cdf_sizes, offsets):
        out = []
        for ci in indexes:
            cdf = cdfs[ci]
            max_value = cdf_sizes[ci] - 2
            cum = self.x & ((1 << PRECISION) - 1)
            s = 0
            while s + 1 < cdf_sizes[ci] and cdf[s + 1] <= cum:
                s += 1
            start, rng = cdf[s], cdf[s + 1] - cdf[s]
            self.x = rng * (self.x >> PRECISION) + cum - start
            self._renorm()
            value = s
            if value == max_value:
                val = self._bits(BYPASS_PRECISION)
                nb = val
                while val == MAX_BYPASS:
                    val = self._bits(BYPASS_PRECISION)
                    nb += val
                raw = 0
                for j in range(nb):
                    raw |= self._bits(BYPASS_PRECISION) << (j * BYPASS_PRECISION)
                value = raw >> 1
                value = -value - 1 if raw & 1 else value + max_value
            out.append(value + offsets[ci])
        return out
