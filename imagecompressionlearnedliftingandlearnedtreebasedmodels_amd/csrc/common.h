// common.h -- shared helpers for the lldwt HIP library (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include "../../include/lldwt.h"

namespace lldwt {

void set_error(const char* fmt, ...);

inline int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: %s", what, hipGetErrorString(e));
        return LLDWT_EHIP;
    }
    return LLDWT_OK;
}

#define LLDWT_REQUIRE(cond, ...)          \
    do {                                  \
        if (!(cond)) {                    \
            lldwt::set_error(__VA_ARGS__); \
            return LLDWT_EINVAL;          \
        }                                 \
    } while (0)

// compute units of the current device (256 on MI355X); 256 if the query fails
static inline int lldwt_num_cus() {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0)
        return 256;
    return n;
}

static inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }
static inline int64_t round_up(int64_t a, int64_t b) { return cdiv(a, b) * b; }

// tanh(x) = sign(x) * (1 - 2 / (exp(2|x|) + 1)) on the hardware exp2 / rcp units (v_exp_f32, v_rcp_f32): ~8 instructions
// instead of ocml's ~40; absolute error <= 3e-7 over the whole range (checked against torch.tanh in the parity tests),
// exact saturation to +-1 for |x| > 10.
__device__ __forceinline__ float fast_tanh(float x) {
    // exp(2|x|) = 2^(2|x| log2 e); no clamp needed: a huge |x| gives e = inf, rcp = 0, t = 1
    const float e = __builtin_amdgcn_exp2f(fabsf(x) * 2.88539008177792681472f);
    const float t = __builtin_fmaf(-2.f, __builtin_amdgcn_rcpf(e + 1.f), 1.f);    // = 1 - 2r (2r is exact: same rounding)
    return copysignf(t, x);
}

__device__ __forceinline__ float act_apply(float v, int act) {
    if (act == LLDWT_ACT_TANH) return fast_tanh(v);
    if (act == LLDWT_ACT_LRELU) return v >= 0.f ? v : 0.01f * v;
    if (act == LLDWT_ACT_RELU) return v > 0.f ? v : 0.f;
    return v;
}

// 64-lane wavefront sum via DPP-free shuffles (wave = 64 on CDNA4)
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}

}  // namespace lldwt
