// split_f16.h -- fp32 -> (hi, lo) fp16 pairs for the split-fp16 MFMA kernels: hi = fp16(v), lo = fp16(v - hi), both
// round-to-nearest, |v - hi - lo| <= 2^-22 |v| while lo stays normal.  hi on 2-element vectors so that hipcc emits
// v_cvt_pk_f16_f32 (one instruction per pair, gfx950).
#pragma once
#include "common.h"

namespace lldwt {

typedef _Float16 h2_t __attribute__((ext_vector_type(2)));
typedef _Float16 h4_t __attribute__((ext_vector_type(4)));
typedef _Float16 h8_t __attribute__((ext_vector_type(8)));
typedef float f2_t __attribute__((ext_vector_type(2)));

// lo through v_fma_mixlo_f16 / v_fma_mixhi_f16: D.f16 = round(v * 1.0 - hi) with v read as fp32 and hi as the fp16 half it
// already is (op_sel_hi / op_sel), one instruction per element and no fp16 -> fp32 conversion: 1.5 vector instructions per
// element.  v - hi is exact in fp32 (|v - hi| <= ulp_f16(v) / 2, a multiple of ulp_f32(v)), so the single rounding to fp16
// gives the same bits as (half)(v - (float)hi); tools/check_split.hip compares the two on 2^21 values.  The compiler does not
// select these instructions from C (it folds the fused form back into a subtraction), hence the inline assembly.
__device__ __forceinline__ void split2(float a, float b, h2_t& hi, h2_t& lo) {
    const f2_t v = {a, b};
    hi = __builtin_convertvector(v, h2_t);
    const unsigned hu = __builtin_bit_cast(unsigned, hi);
    unsigned lu;
    asm("v_fma_mixlo_f16 %0, %1, 1.0, -%2 op_sel_hi:[0,0,1]" : "=v"(lu) : "v"(a), "v"(hu));
    asm("v_fma_mixhi_f16 %0, %1, 1.0, -%2 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(lu) : "v"(b), "v"(hu));
    lo = __builtin_bit_cast(h2_t, lu);
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

// ---- the one-product modes (lldwt_set_precision 1 = fp16, 2 = bf16): 16-byte fragments travel as h8_t whatever they hold
typedef __bf16 bf8_t __attribute__((ext_vector_type(8)));
typedef float f16_t __attribute__((ext_vector_type(16)));

// one MFMA product of the 32x32x16 shape on fp16 (PREC 0, 1) or bf16 (PREC 2) operands
template <int PREC>
__device__ __forceinline__ f16_t mma32(const h8_t& a, const h8_t& b, const f16_t& acc) {
    if constexpr (PREC == 2)
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf8_t, a), __builtin_bit_cast(bf8_t, b), acc, 0, 0, 0);
    else
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
}
// the 16x16x32 shape
typedef float f4_t __attribute__((ext_vector_type(4)));
template <int PREC>
__device__ __forceinline__ f4_t mma16(const h8_t& a, const h8_t& b, const f4_t& acc) {
    if constexpr (PREC == 2)
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf8_t, a), __builtin_bit_cast(bf8_t, b), acc, 0, 0, 0);
    else
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc, 0, 0, 0);
}
// 8 fp32 values -> 8 fp16 (PREC 1) or bf16 (PREC 2) operands, round to nearest
template <int PREC>
__device__ __forceinline__ h8_t cvt8(const float (&v)[8]) {
    typedef float f8 __attribute__((ext_vector_type(8)));
    const f8 x = {v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7]};
    if constexpr (PREC == 2) return __builtin_bit_cast(h8_t, __builtin_convertvector(x, bf8_t));
    else return __builtin_convertvector(x, h8_t);
}

}  // namespace lldwt
