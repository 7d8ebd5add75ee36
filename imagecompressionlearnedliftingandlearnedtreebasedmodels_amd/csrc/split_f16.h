// split_f16.h -- fp32 -> (hi, lo) fp16 pairs for the split-fp16 MFMA kernels: hi = fp16(v), lo = fp16(v - hi), both
// round-to-nearest, |v - hi - lo| <= 2^-22 |v| while lo stays normal.  Written on 2-element vectors so that hipcc emits
// v_cvt_pk_f16_f32 (one instruction per pair, gfx950): 2.5 vector instructions per element instead of ~5.
#pragma once
#include "common.h"

namespace lldwt {

typedef _Float16 h2_t __attribute__((ext_vector_type(2)));
typedef _Float16 h4_t __attribute__((ext_vector_type(4)));
typedef _Float16 h8_t __attribute__((ext_vector_type(8)));
typedef float f2_t __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void split2(float a, float b, h2_t& hi, h2_t& lo) {
    const f2_t v = {a, b};
    hi = __builtin_convertvector(v, h2_t);
    const f2_t r = v - __builtin_convertvector(hi, f2_t);
    lo = __builtin_convertvector(r, h2_t);
}

__device__ __forceinline__ void split4v(const float (&v)[4], h4_t& hi, h4_t& lo) {
    h2_t a, b, c, d;
    split2(v[0], v[1], a, b);
    split2(v[2], v[3], c, d);
    hi = __builtin_shufflevector(a, c, 0, 1, 2, 3);
    lo = __builtin_shufflevector(b, d, 0, 1, 2, 3);
}

__device__ __forceinline__ void split8v(const float (&v)[8], h8_t& hi, h8_t& lo) {
    h4_t a, b, c, d;
    const float v0[4] = {v[0], v[1], v[2], v[3]}, v1[4] = {v[4], v[5], v[6], v[7]};
    split4v(v0, a, b);
    split4v(v1, c, d);
    hi = __builtin_shufflevector(a, c, 0, 1, 2, 3, 4, 5, 6, 7);
    lo = __builtin_shufflevector(b, d, 0, 1, 2, 3, 4, 5, 6, 7);
}

}  // namespace lldwt
