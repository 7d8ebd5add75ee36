// rans.hip -- range-ANS entropy coder, HOST code only (no kernel in this file; it is compiled with the library so that the
// C-ABI is one shared object).
//
// Replaces the reference's use of compressai==1.2.1 `compressai.ans` (C++ extension, absent from the image and not
// vendored): BufferedRansEncoder.encode_with_indexes / flush and RansDecoder.set_stream / decode_stream, call sites
// graphs/models/LiftingBasedDWT_net.py:466,502-505 (compress_ar) and :516-517,540-546 (decompress_ar).  Restated from the
// published algorithm: compressai's rans_interface (precision 16, bypass precision 4, out-of-range symbols escape through
// the last CDF slot and are sent as 4-bit "bypass" digits) on top of Fabian Giesen's public-domain rans64 (64-bit state,
// 32-bit renormalisation words, lower bound 2^31).  Symbols are pushed in coding order and the state machine runs over
// them in REVERSE, so the decoder pops them in forward order.
// Also pmf_to_quantized_cdf (compressai _CXX): float pmf -> 16-bit CDF with every symbol given a non-zero frequency.
// parity unpinned against compressai's bytes (package absent, the reference holds no bitstream fixture); pinned instead by
// an independent pure-Python restatement held by the tests, by decode(encode(x)) == x and by the code length.
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <numeric>
#include <vector>

#include "common.h"

namespace {

constexpr int kPrecision = 16;           // CDF precision (bits)
constexpr int kBypassPrecision = 4;      // bits per bypass digit
constexpr int kMaxBypass = (1 << kBypassPrecision) - 1;
constexpr uint64_t kRansL = 1ull << 31;  // lower bound of the normalisation interval

struct Sym {
    uint16_t start;
    uint16_t range;
    bool bypass;
};

inline void enc_put(uint64_t& x, std::vector<uint32_t>& out, uint32_t start, uint32_t freq, uint32_t scale_bits) {
    const uint64_t x_max = ((kRansL >> scale_bits) << 32) * freq;
    if (x >= x_max) {
        out.push_back((uint32_t)x);
        x >>= 32;
    }
    x = ((x / freq) << scale_bits) + (x % freq) + start;
}

inline void enc_put_bits(uint64_t& x, std::vector<uint32_t>& out, uint32_t val, uint32_t nbits) {
    const uint32_t freq = 1u << (16 - nbits);
    const uint64_t x_max = ((kRansL >> 16) << 32) * freq;
    if (x >= x_max) {
        out.push_back((uint32_t)x);
        x >>= 32;
    }
    x = (x << nbits) | val;
}

struct Decoder {
    std::vector<uint32_t> words;
    size_t pos;
    uint64_t x;
    bool bad;

    uint32_t next_word() {
        if (pos >= words.size()) {
            bad = true;
            return 0;
        }
        return words[pos++];
    }
    void init() {
        pos = 0;
        bad = false;
        x = (uint64_t)next_word();
        x |= (uint64_t)next_word() << 32;
    }
    uint32_t get(uint32_t scale_bits) const { return (uint32_t)(x & ((1u << scale_bits) - 1)); }
    void advance(uint32_t start, uint32_t freq, uint32_t scale_bits) {
        const uint64_t mask = (1ull << scale_bits) - 1;
        x = freq * (x >> scale_bits) + (x & mask) - start;
        if (x < kRansL) x = (x << 32) | next_word();
    }
    uint32_t get_bits(uint32_t nbits) {
        const uint32_t val = (uint32_t)(x & ((1u << nbits) - 1));
        x >>= nbits;
        if (x < kRansL) x = (x << 32) | next_word();
        return val;
    }
};

}  // namespace

extern "C" int lldwt_pmf_to_quantized_cdf(const float* pmf, int n, int precision, uint32_t* cdf) {
    LLDWT_REQUIRE(pmf && cdf && n > 0 && precision > 0 && precision <= 16, "pmf_to_quantized_cdf: bad arguments");
    std::vector<uint32_t> c(n + 1);
    c[0] = 0;
    for (int i = 0; i < n; ++i) c[i + 1] = (uint32_t)std::lround((double)(pmf[i] * (float)(1 << precision)));
    const uint64_t total = std::accumulate(c.begin(), c.end(), (uint64_t)0);
    LLDWT_REQUIRE(total > 0, "pmf_to_quantized_cdf: pmf sums to zero");
    for (auto& p : c) p = (uint32_t)((((uint64_t)1 << precision) * p) / total);
    std::partial_sum(c.begin(), c.end(), c.begin());
    c.back() = 1u << precision;
    for (int i = 0; i < n; ++i) {
        if (c[i] == c[i + 1]) {          // zero frequency: steal from the least frequent symbol that can spare one
            uint32_t best_freq = ~0u;
            int best = -1;
            for (int j = 0; j < n; ++j) {
                const uint32_t f = c[j + 1] - c[j];
                if (f > 1 && f < best_freq) {
                    best_freq = f;
                    best = j;
                }
            }
            LLDWT_REQUIRE(best != -1, "pmf_to_quantized_cdf: cannot give every symbol a non-zero frequency");
            if (best < i)
                for (int j = best + 1; j <= i; ++j) c[j]--;
            else
                for (int j = i + 1; j <= best; ++j) c[j]++;
        }
    }
    memcpy(cdf, c.data(), sizeof(uint32_t) * (n + 1));
    return LLDWT_OK;
}

extern "C" int64_t lldwt_rans_encode(const int32_t* symbols, const int32_t* indexes, int64_t n, const int32_t* cdfs,
                                     int32_t ncdf, int32_t cdf_stride, const int32_t* cdf_sizes, const int32_t* offsets,
                                     uint8_t* out, int64_t out_cap) {
    if (!(symbols && indexes && cdfs && cdf_sizes && offsets && out && n >= 0 && ncdf > 0 && cdf_stride > 1)) {
        lldwt::set_error("rans_encode: bad arguments");
        return LLDWT_EINVAL;
    }
    std::vector<Sym> syms;
    syms.reserve((size_t)n + 16);
    for (int64_t i = 0; i < n; ++i) {
        const int32_t ci = indexes[i];
        if (ci < 0 || ci >= ncdf) {
            lldwt::set_error("rans_encode: cdf index %d out of range at symbol %lld", ci, (long long)i);
            return LLDWT_EINVAL;
        }
        const int32_t* cdf = cdfs + (int64_t)ci * cdf_stride;
        const int32_t max_value = cdf_sizes[ci] - 2;
        if (max_value < 0 || max_value + 1 >= cdf_stride) {
            lldwt::set_error("rans_encode: bad cdf size %d", cdf_sizes[ci]);
            return LLDWT_EINVAL;
        }
        int32_t value = symbols[i] - offsets[ci];
        uint32_t raw_val = 0;
        if (value < 0) {
            raw_val = (uint32_t)(-2 * value - 1);
            value = max_value;
        } else if (value >= max_value) {
            raw_val = (uint32_t)(2 * (value - max_value));
            value = max_value;
        }
        syms.push_back({(uint16_t)cdf[value], (uint16_t)(cdf[value + 1] - cdf[value]), false});
        if (value == max_value) {                  // escape: Elias-gamma-like bypass digits
            int32_t n_bypass = 0;
            while (n_bypass * kBypassPrecision < 32 && (raw_val >> (n_bypass * kBypassPrecision)) != 0) ++n_bypass;
            int32_t val = n_bypass;
            while (val >= kMaxBypass) {
                syms.push_back({(uint16_t)kMaxBypass, (uint16_t)(kMaxBypass + 1), true});
                val -= kMaxBypass;
            }
            syms.push_back({(uint16_t)val, (uint16_t)(val + 1), true});
            for (int32_t j = 0; j < n_bypass; ++j) {
                const int32_t v = (raw_val >> (j * kBypassPrecision)) & kMaxBypass;
                syms.push_back({(uint16_t)v, (uint16_t)(v + 1), true});
            }
        }
    }
    std::vector<uint32_t> words;
    words.reserve(syms.size() / 2 + 4);
    uint64_t x = kRansL;
    for (size_t k = syms.size(); k-- > 0;) {
        const Sym& s = syms[k];
        if (!s.bypass)
            enc_put(x, words, s.start, s.range, kPrecision);
        else
            enc_put_bits(x, words, s.start, kBypassPrecision);
    }
    words.push_back((uint32_t)(x >> 32));          // flush: high word then low word are the LAST written = first read
    words.push_back((uint32_t)x);
    const int64_t nbytes = (int64_t)words.size() * 4;
    if (nbytes > out_cap) {
        lldwt::set_error("rans_encode: output buffer too small (%lld > %lld)", (long long)nbytes, (long long)out_cap);
        return LLDWT_EINVAL;
    }
    // the encoder wrote backwards: the stream is the word sequence reversed
    uint32_t* o = reinterpret_cast<uint32_t*>(out);
    for (size_t k = 0; k < words.size(); ++k) {
        const uint32_t w = words[words.size() - 1 - k];
        memcpy(o + k, &w, 4);
    }
    return nbytes;
}

extern "C" void* lldwt_rans_decoder_new(const uint8_t* stream, int64_t nbytes) {
    if (!stream || nbytes < 8 || (nbytes & 3)) {
        lldwt::set_error("rans_decoder_new: a stream is at least two 32-bit words");
        return nullptr;
    }
    Decoder* d = new Decoder();
    d->words.resize((size_t)nbytes / 4);
    memcpy(d->words.data(), stream, (size_t)nbytes);
    d->init();
    return d;
}

extern "C" void lldwt_rans_decoder_free(void* dec) { delete reinterpret_cast<Decoder*>(dec); }

extern "C" int lldwt_rans_decode(void* dec, const int32_t* indexes, int64_t n, const int32_t* cdfs, int32_t ncdf,
                                 int32_t cdf_stride, const int32_t* cdf_sizes, const int32_t* offsets, int32_t* symbols) {
    LLDWT_REQUIRE(dec && indexes && cdfs && cdf_sizes && offsets && symbols && n >= 0, "rans_decode: bad arguments");
    Decoder& d = *reinterpret_cast<Decoder*>(dec);
    for (int64_t i = 0; i < n; ++i) {
        const int32_t ci = indexes[i];
        LLDWT_REQUIRE(ci >= 0 && ci < ncdf, "rans_decode: cdf index %d out of range", ci);
        const int32_t* cdf = cdfs + (int64_t)ci * cdf_stride;
        const int32_t csize = cdf_sizes[ci];
        const int32_t max_value = csize - 2;
        const uint32_t cum = d.get(kPrecision);
        // first entry greater than cum, minus one (the table is increasing over its first csize entries)
        const int32_t* it = std::upper_bound(cdf, cdf + csize, (int32_t)cum);
        const int32_t s = (int32_t)(it - cdf) - 1;
        LLDWT_REQUIRE(s >= 0 && s <= max_value, "rans_decode: corrupt stream (slot %d)", s);
        d.advance((uint32_t)cdf[s], (uint32_t)(cdf[s + 1] - cdf[s]), kPrecision);
        int32_t value = s;
        if (value == max_value) {
            int32_t val = (int32_t)d.get_bits(kBypassPrecision);
            int32_t n_bypass = val;
            while (val == kMaxBypass) {
                val = (int32_t)d.get_bits(kBypassPrecision);
                n_bypass += val;
            }
            uint32_t raw_val = 0;
            for (int32_t j = 0; j < n_bypass; ++j) {
                val = (int32_t)d.get_bits(kBypassPrecision);
                raw_val |= (uint32_t)val << (j * kBypassPrecision);
            }
            value = (int32_t)(raw_val >> 1);
            if (raw_val & 1)
                value = -value - 1;
            else
                value += max_value;
        }
        symbols[i] = value + offsets[ci];
        LLDWT_REQUIRE(!d.bad, "rans_decode: read past the end of the stream");
    }
    return LLDWT_OK;
}

// One call for a whole wavefront step: stream k (of nstreams decoders) pops n symbols for the indexes at indexes + k * stride,
// into symbols + k * stride.  The same tables for every stream.  (A Python loop over the streams of a step cost more than the
// decoding itself: ~10 us of call overhead per stream and step.)
extern "C" int lldwt_rans_decode_multi(void* const* decs, int64_t nstreams, const int32_t* indexes, int64_t n, int64_t stride,
                                       const int32_t* cdfs, int32_t ncdf, int32_t cdf_stride, const int32_t* cdf_sizes,
                                       const int32_t* offsets, int32_t* symbols) {
    LLDWT_REQUIRE(decs && nstreams >= 0 && stride >= n, "rans_decode_multi: bad arguments");
    for (int64_t k = 0; k < nstreams; ++k) {
        const int r = lldwt_rans_decode(decs[k], indexes + k * stride, n, cdfs, ncdf, cdf_stride, cdf_sizes, offsets, symbols + k * stride);
        if (r) return r;
    }
    return LLDWT_OK;
}
