// ops.hip -- elementwise / small kernels of the hot path for gfx950: colour transforms, subband MLP,
// reference-order direct conv (fallback + cross-check of the MFMA conv engine), GDN, bound ops, rate estimation,
// reductions.  Every kernel cites the reference code it replaces.
#include "common.h"
#include <string.h>
#include <math.h>

namespace lldwt {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

// ------------------------------------------------------------------------------------------ colour
// compressai.transforms.functional rgb2ycbcr / ycbcr2rgb (BT.709), agents/liftingDWT_agent.py:86-87,90-94
constexpr float KR = 0.2126f, KG = 0.7152f, KB = 0.0722f;

__global__ void k_rgb_to_ycc(const float* __restrict__ rgb, float* __restrict__ ycc, int64_t B, int64_t hw) {
    const int64_t n = B * hw;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t b = i / hw, p = i - b * hw;
        const float r = rgb[(b * 3 + 0) * hw + p], g = rgb[(b * 3 + 1) * hw + p], bl = rgb[(b * 3 + 2) * hw + p];
        const float y = KR * r + KG * g + KB * bl;
        const float cb = 0.5f * (bl - y) / (1.f - KB) + 0.5f;
        const float cr = 0.5f * (r - y) / (1.f - KR) + 0.5f;
        ycc[(0 * B + b) * hw + p] = y - 0.5f;
        ycc[(1 * B + b) * hw + p] = cb;
        ycc[(2 * B + b) * hw + p] = cr;
    }
}

// uint8 HWC image batch (as decoded by PIL) -> float NCHW in [0,1]: torchvision ToTensor semantics (x / 255 in fp32),
// dataloaders/image_dl.py:81.  One thread per pixel: reads 3 consecutive bytes, writes three coalesced planes.
__global__ void k_u8hwc_to_f32chw(const uint8_t* __restrict__ src, float* __restrict__ dst, int64_t B, int64_t hw) {
    const int64_t n = B * hw;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t b = i / hw, p = i - b * hw;
        const uint8_t* s = src + i * 3;
        dst[(b * 3 + 0) * hw + p] = (float)s[0] / 255.0f;
        dst[(b * 3 + 1) * hw + p] = (float)s[1] / 255.0f;
        dst[(b * 3 + 2) * hw + p] = (float)s[2] / 255.0f;
    }
}

__global__ void k_ycc_to_rgb(const float* __restrict__ ycc, float* __restrict__ rgb, int64_t B, int64_t hw, int clamp) {
    const int64_t n = B * hw;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t b = i / hw, p = i - b * hw;
        const float y = ycc[(0 * B + b) * hw + p] + 0.5f, cb = ycc[(1 * B + b) * hw + p], cr = ycc[(2 * B + b) * hw + p];
        float r = y + (2.f - 2.f * KR) * (cr - 0.5f);
        float bl = y + (2.f - 2.f * KB) * (cb - 0.5f);
        float g = (y - KR * r - KB * bl) / KG;
        r -= 0.5f; g -= 0.5f; bl -= 0.5f;
        if (clamp) {
            r = fminf(fmaxf(r, -0.5f), 0.5f);
            g = fminf(fmaxf(g, -0.5f), 0.5f);
            bl = fminf(fmaxf(bl, -0.5f), 0.5f);
        }
        rgb[(b * 3 + 0) * hw + p] = r;
        rgb[(b * 3 + 1) * hw + p] = g;
        rgb[(b * 3 + 2) * hw + p] = bl;
    }
}

// ------------------------------------------------------------------------------------------ subband MLP
// SubbandAutoEncoder (lifting_dwt_nets.py:99-110): 1 -> 32 -> 32 -> 32 -> 1 per coefficient, tanh between, on the matrix
// cores: the two 32x32 layers run on v_mfma_f32_16x16x4_f32 with everything in registers.
// A wave takes 64 coefficients (4 column tiles of 16).  Layer outputs come out of the MFMA as D[row = oc][col = coef]
// with lane (col, kk) holding rows 4kk..4kk+3 of each 16-row tile -- and that is directly usable as the NEXT layer's
// B operand if its 8 k-steps are taken in the order (tile m', r): k index kk <-> input channel m'*16 + 4kk + r.  So the
// weights are loaded once per wave in that permuted order (32 registers for both layers) and no activation ever goes
// through LDS or a lane shuffle; the 1 -> 32 layer is computed straight into that layout, the 32 -> 1 layer is a dot
// product over the lane's 8 rows plus two xor-shuffles across kk.
typedef float floatx4m __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k_subband_mlp_mfma(const float* __restrict__ x, float* __restrict__ y, int batch, int C,
                                                          int64_t hw, const float* __restrict__ w0,
                                                          const float* __restrict__ b0, const float* __restrict__ w1,
                                                          const float* __restrict__ b1, const float* __restrict__ w2,
                                                          const float* __restrict__ b2, const float* __restrict__ w3,
                                                          const float* __restrict__ b3, int transposed) {
    constexpr int HD = 32;
    const int c = blockIdx.y;
    const int64_t z = blockIdx.z;
    const int plane = (int)(z / batch);
    const int64_t pc = (int64_t)plane * C + c;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int col = lane & 15, kk = lane >> 4;
    // A operands: A[m][q], q = (m', r): weight W[oc = m*16 + col][ic = m'*16 + 4kk + r]
    float A1[2][8], A2[2][8], w0b[8], b0b[8], b1d[8], b2d[8], w3d[8];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int oc = m * 16 + col, ic = (q >> 2) * 16 + 4 * kk + (q & 3);
            const int64_t src = transposed ? (pc * HD + ic) * HD + oc : (pc * HD + oc) * HD + ic;
            A1[m][q] = w1[src];
            A2[m][q] = w2[src];
        }
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int ch = (q >> 2) * 16 + 4 * kk + (q & 3);       // the lane's 8 rows / input channels
        w0b[q] = w0[pc * HD + ch];
        b0b[q] = b0[pc * HD + ch];
        b1d[q] = b1[pc * HD + ch];
        b2d[q] = b2[pc * HD + ch];
        w3d[q] = w3[pc * HD + ch];
    }
    const float bb3 = b3[pc];
    const float* xp = x + (z * C + c) * hw;
    float* yp = y + (z * C + c) * hw;
    for (int64_t i0 = ((int64_t)blockIdx.x * 4 + wave) * 64; i0 < hw; i0 += (int64_t)gridDim.x * 256) {
        float xv[4];
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            const int64_t i = i0 + n * 16 + col;
            xv[n] = xp[i < hw ? i : 0];
        }
        float h[4][8];                                          // activations in B-operand layout: [column tile][q]
#pragma unroll
        for (int n = 0; n < 4; ++n)
#pragma unroll
            for (int q = 0; q < 8; ++q) h[n][q] = fast_tanh(fmaf(w0b[q], xv[n], b0b[q]));
#pragma unroll
        for (int layer = 0; layer < 2; ++layer) {
            floatx4m acc[2][4];
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int n = 0; n < 4; ++n) {
                    const float* bd = layer == 0 ? b1d : b2d;
                    acc[m][n] = floatx4m{bd[m * 4 + 0], bd[m * 4 + 1], bd[m * 4 + 2], bd[m * 4 + 3]};
                }
#pragma unroll
            for (int q = 0; q < 8; ++q)
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int n = 0; n < 4; ++n)
                        acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(layer == 0 ? A1[m][q] : A2[m][q], h[n][q], acc[m][n],
                                                                         0, 0, 0);
#pragma unroll
            for (int n = 0; n < 4; ++n)
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int r = 0; r < 4; ++r) h[n][m * 4 + r] = fast_tanh(acc[m][n][r]);
        }
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            float o = 0.f;
#pragma unroll
            for (int q = 0; q < 8; ++q) o = fmaf(w3d[q], h[n][q], o);
            o += __shfl_xor(o, 16, 64);
            o += __shfl_xor(o, 32, 64);
            const int64_t i = i0 + n * 16 + col;
            if (kk == 0 && i < hw) yp[i] = o + bb3;
        }
    }
}

// Backward of the same MLP (training), also in registers: the forward is recomputed, then the gradient runs back through
// the transposed 32x32 layers with the same permuted-k trick (a layer's gradient in D layout is the next MFMA's B operand).
// Written out for the weight-gradient GEMMs (lldwt_conv2d_wgrad with groups == C): the activations h0,h1,h2 (their x
// operands) and d0,d1,d2, the gradients at the pre-activation outputs of layers 0..2 (their dy operands), all
// (Z, C*32, hw); gx (Z, C, hw) is the gradient at the input.  Weights in Conv2d layout (as lldwt_subband_mlp, encode).
__global__ __launch_bounds__(256) void k_subband_mlp_bwd(const float* __restrict__ x, const float* __restrict__ gy,
                                                         float* __restrict__ gx, float* __restrict__ h0o,
                                                         float* __restrict__ h1o, float* __restrict__ h2o,
                                                         float* __restrict__ d0o, float* __restrict__ d1o,
                                                         float* __restrict__ d2o, int batch, int C, int64_t hw,
                                                         const float* __restrict__ w0, const float* __restrict__ b0,
                                                         const float* __restrict__ w1, const float* __restrict__ b1,
                                                         const float* __restrict__ w2, const float* __restrict__ b2,
                                                         const float* __restrict__ w3) {
    constexpr int HD = 32, NC = 2;                       // NC column tiles (16 coefficients each) per wave and iteration
    const int c = blockIdx.y;
    const int64_t z = blockIdx.z;
    const int plane = (int)(z / batch);
    const int64_t pc = (int64_t)plane * C + c;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int col = lane & 15, kk = lane >> 4;
    float A1[2][8], A2[2][8], A1T[2][8], A2T[2][8], w0b[8], b0b[8], b1d[8], b2d[8], w3d[8];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int row = m * 16 + col, ch = (q >> 2) * 16 + 4 * kk + (q & 3);
            A1[m][q] = w1[(pc * HD + row) * HD + ch];            // forward: row = oc, k <-> ic
            A2[m][q] = w2[(pc * HD + row) * HD + ch];
            A1T[m][q] = w1[(pc * HD + ch) * HD + row];           // backward: row = ic, k <-> oc
            A2T[m][q] = w2[(pc * HD + ch) * HD + row];
        }
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int ch = (q >> 2) * 16 + 4 * kk + (q & 3);
        w0b[q] = w0[pc * HD + ch];
        b0b[q] = b0[pc * HD + ch];
        b1d[q] = b1[pc * HD + ch];
        b2d[q] = b2[pc * HD + ch];
        w3d[q] = w3[pc * HD + ch];
    }
    const float* xp = x + (z * C + c) * hw;
    const float* gp = gy + (z * C + c) * hw;
    float* gxp = gx + (z * C + c) * hw;
    const int64_t hb = (z * C + c) * (int64_t)HD * hw;      // base of this (image, channel)'s 32 hidden channels
#define LLDWT_MLP_GEMM(OUT_, AW_, IN_, BIAS_)                                                                    \
    {                                                                                                            \
        floatx4m acc_[2][NC];                                                                                    \
        _Pragma("unroll") for (int m = 0; m < 2; ++m)                                                            \
            _Pragma("unroll") for (int n = 0; n < NC; ++n)                                                       \
                acc_[m][n] = floatx4m{BIAS_[m * 4 + 0], BIAS_[m * 4 + 1], BIAS_[m * 4 + 2], BIAS_[m * 4 + 3]};   \
        _Pragma("unroll") for (int q = 0; q < 8; ++q)                                                            \
            _Pragma("unroll") for (int m = 0; m < 2; ++m)                                                        \
                _Pragma("unroll") for (int n = 0; n < NC; ++n)                                                   \
                    acc_[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(AW_[m][q], IN_[n][q], acc_[m][n], 0, 0, 0); \
        _Pragma("unroll") for (int n = 0; n < NC; ++n)                                                           \
            _Pragma("unroll") for (int m = 0; m < 2; ++m)                                                        \
                _Pragma("unroll") for (int r = 0; r < 4; ++r) OUT_[n][m * 4 + r] = acc_[m][n][r];                \
    }
#define LLDWT_MLP_WRITE(PTR_, V_)                                                                                \
    _Pragma("unroll") for (int n = 0; n < NC; ++n) {                                                             \
        const int64_t i_ = i0 + n * 16 + col;                                                                    \
        if (i_ < hw) {                                                                                           \
            _Pragma("unroll") for (int q = 0; q < 8; ++q)                                                        \
                PTR_[hb + (int64_t)((q >> 2) * 16 + 4 * kk + (q & 3)) * hw + i_] = V_[n][q];                     \
        }                                                                                                        \
    }
    const float zero8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int64_t i0 = ((int64_t)blockIdx.x * 4 + wave) * (16 * NC); i0 < hw; i0 += (int64_t)gridDim.x * 4 * 16 * NC) {
        float xv[NC], gv[NC];
#pragma unroll
        for (int n = 0; n < NC; ++n) {
            const int64_t i = i0 + n * 16 + col;
            xv[n] = xp[i < hw ? i : 0];
            gv[n] = gp[i < hw ? i : 0];
        }
        float h0[NC][8], h1[NC][8], h2[NC][8], d[NC][8], t[NC][8];
#pragma unroll
        for (int n = 0; n < NC; ++n)
#pragma unroll
            for (int q = 0; q < 8; ++q) h0[n][q] = fast_tanh(fmaf(w0b[q], xv[n], b0b[q]));
        LLDWT_MLP_GEMM(t, A1, h0, b1d)
#pragma unroll
        for (int n = 0; n < NC; ++n)
#pragma unroll
            for (int q = 0; q < 8; ++q) h1[n][q] = fast_tanh(t[n][q]);
        LLDWT_MLP_GEMM(t, A2, h1, b2d)
#pragma unroll
        for (int n = 0; n < NC; ++n)
#pragma unroll
            for (int q = 0; q < 8; ++q) h2[n][q] = fast_tanh(t[n][q]);
        LLDWT_MLP_WRITE(h0o, h0)
        LLDWT_MLP_WRITE(h1o, h1)
        LLDWT_MLP_WRITE(h2o, h2)
        // y = w3 . h2 + b3  ->  d2 = w3 * gy * tanh'(pre2)
#pragma unroll
        for (int n = 0; n < NC; ++n)
#pragma unroll
            for (int q = 0; q < 8; ++q) d[n][q] = w3d[q] * gv[n] * (1.f - h2[n][q] * h2[n][q]);
        LLDWT_MLP_WRITE(d2o, d)
        LLDWT_MLP_GEMM(t, A2T, d, zero8)
#pragma unroll
        for (int n = 0; n < NC; ++n)
#pragma unroll
            for (int q = 0; q < 8; ++q) d[n][q] = t[n][q] * (1.f - h1[n][q] * h1[n][q]);
        LLDWT_MLP_WRITE(d1o, d)
        LLDWT_MLP_GEMM(t, A1T, d, zero8)
#pragma unroll
        for (int n = 0; n < NC; ++n)
#pragma unroll
            for (int q = 0; q < 8; ++q) d[n][q] = t[n][q] * (1.f - h0[n][q] * h0[n][q]);
        LLDWT_MLP_WRITE(d0o, d)
#pragma unroll
        for (int n = 0; n < NC; ++n) {
            float o = 0.f;
#pragma unroll
            for (int q = 0; q < 8; ++q) o = fmaf(w0b[q], d[n][q], o);
            o += __shfl_xor(o, 16, 64);
            o += __shfl_xor(o, 32, 64);
            const int64_t i = i0 + n * 16 + col;
            if (kk == 0 && i < hw) gxp[i] = o;
        }
    }
#undef LLDWT_MLP_GEMM
#undef LLDWT_MLP_WRITE
}

// The same backward with the weight gradients formed IN the kernel (training default): nothing but x, gy and gx touches HBM -- the
// six (Z, C*32, hw) tensors of the form above are 768 B per coefficient written and read again by four GEMM launches.
//   dW1 = sum over coefficients d1 (x) h0,  dW2 = d2 (x) h1    (32 x 32, on v_mfma_f32_16x16x4_f32 with K = coefficients)
//   dw0 = sum d0 x, db0 = sum d0, db1 = sum d1, db2 = sum d2, dw3 = sum gy h2, db3 = sum gy     (per lane, reduced at the end)
// The matrix products contract over coefficients, which the register layout above spreads over lanes (a lane = one coefficient
// column x 8 channels): each wave passes d and h of its 32 coefficients through a private LDS tile [coefficient][channel] (row
// pitch 48 floats: the transposed 4-byte reads of an MFMA operand -- lane (channel, coefficient mod 4) -- are conflict-free) --
// no barrier, a wave's LDS operations execute in order.  A workgroup = (slice of the plane, channel, plane) loops over the
// batch; every WAVE ends by storing its partial sums as one row of `part` (no atomics: k_subband_mlp_wsum adds the rows in a
// fixed order, the gradients need no zero-fill and do not depend on scheduling).
constexpr int MLPW_ROW = 2240;          // dW1 1024 | dW2 1024 | dw0 32 | db0 32 | db1 32 | db2 32 | dw3 32 | db3 1 (+ padding)
constexpr int MLPW_PITCH = 48;
__global__ __launch_bounds__(256) void k_subband_mlp_bwdw(const float* __restrict__ x, const float* __restrict__ gy,
                                                          float* __restrict__ gx, float* __restrict__ part, int batch, int C,
                                                          int64_t hw, const float* __restrict__ w0, const float* __restrict__ b0,
                                                          const float* __restrict__ w1, const float* __restrict__ b1,
                                                          const float* __restrict__ w2, const float* __restrict__ b2,
                                                          const float* __restrict__ w3) {
    constexpr int HD = 32, NC = 2;
    __shared__ __attribute__((aligned(16))) float tbuf[4][2][32 * MLPW_PITCH];
    const int c = blockIdx.y, plane = blockIdx.z;
    const int64_t pc = (int64_t)plane * C + c;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int col = lane & 15, kk = lane >> 4;
    float* bufA = tbuf[wave][0];
    float* bufB = tbuf[wave][1];
    float A1[2][8], A2[2][8], A1T[2][8], A2T[2][8], w0b[8], b0b[8], b1d[8], b2d[8], w3d[8];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int row = m * 16 + col, ch = (q >> 2) * 16 + 4 * kk + (q & 3);
            A1[m][q] = w1[(pc * HD + row) * HD + ch];
            A2[m][q] = w2[(pc * HD + row) * HD + ch];
            A1T[m][q] = w1[(pc * HD + ch) * HD + row];
            A2T[m][q] = w2[(pc * HD + ch) * HD + row];
        }
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int ch = (q >> 2) * 16 + 4 * kk + (q & 3);
        w0b[q] = w0[pc * HD + ch];
        b0b[q] = b0[pc * HD + ch];
        b1d[q] = b1[pc * HD + ch];
        b2d[q] = b2[pc * HD + ch];
        w3d[q] = w3[pc * HD + ch];
    }
    floatx4m aw1[2][2], aw2[2][2];          // [oc tile][ic tile], D layout: row oc = 4 kk + r, column ic = col
    float sdw0[8], sdb0[8], sdb1[8], sdb2[8], sdw3[8], sdb3 = 0.f;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) aw1[a][b] = aw2[a][b] = floatx4m{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int q = 0; q < 8; ++q) sdw0[q] = sdb0[q] = sdb1[q] = sdb2[q] = sdw3[q] = 0.f;
#define LLDWT_MLPW_GEMM(OUT_, AW_, IN_, BIAS_)                                                                   \
    {                                                                                                            \
        floatx4m acc_[2][NC];                                                                                    \
        _Pragma("unroll") for (int m = 0; m < 2; ++m)                                                            \
            _Pragma("unroll") for (int n = 0; n < NC; ++n)                                                       \
                acc_[m][n] = floatx4m{BIAS_[m * 4 + 0], BIAS_[m * 4 + 1], BIAS_[m * 4 + 2], BIAS_[m * 4 + 3]};   \
        _Pragma("unroll") for (int q = 0; q < 8; ++q)                                                            \
            _Pragma("unroll") for (int m = 0; m < 2; ++m)                                                        \
                _Pragma("unroll") for (int n = 0; n < NC; ++n)                                                   \
                    acc_[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(AW_[m][q], IN_[n][q], acc_[m][n], 0, 0, 0); \
        _Pragma("unroll") for (int n = 0; n < NC; ++n)                                                           \
            _Pragma("unroll") for (int m = 0; m < 2; ++m)                                                        \
                _Pragma("unroll") for (int r = 0; r < 4; ++r) OUT_[n][m * 4 + r] = acc_[m][n][r];                \
    }
    // tensor in the register layout -> this wave's LDS tile [coefficient n * 16 + col][channel]
#define LLDWT_MLPW_PUT(BUF_, V_)                                                                                 \
    _Pragma("unroll") for (int n = 0; n < NC; ++n)                                                               \
        _Pragma("unroll") for (int m = 0; m < 2; ++m)                                                            \
            *reinterpret_cast<floatx4m*>(BUF_ + (n * 16 + col) * MLPW_PITCH + m * 16 + 4 * kk) =                 \
                floatx4m{V_[n][m * 4 + 0], V_[n][m * 4 + 1], V_[n][m * 4 + 2], V_[n][m * 4 + 3]};
    // ACC_[oc tile][ic tile] += (bufA as d)[oc][coefficient] . (bufB as h)[ic][coefficient] over the wave's 32 coefficients
#define LLDWT_MLPW_OUTER(ACC_)                                                                                   \
    _Pragma("unroll") for (int j = 0; j < 4 * NC; ++j) {                                                         \
        const float a0_ = bufA[(4 * j + kk) * MLPW_PITCH + col], a1_ = bufA[(4 * j + kk) * MLPW_PITCH + 16 + col]; \
        const float b0_ = bufB[(4 * j + kk) * MLPW_PITCH + col], b1_ = bufB[(4 * j + kk) * MLPW_PITCH + 16 + col]; \
        ACC_[0][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0_, b0_, ACC_[0][0], 0, 0, 0);                        \
        ACC_[0][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0_, b1_, ACC_[0][1], 0, 0, 0);                        \
        ACC_[1][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1_, b0_, ACC_[1][0], 0, 0, 0);                        \
        ACC_[1][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1_, b1_, ACC_[1][1], 0, 0, 0);                        \
    }
    const float zero8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int b = 0; b < batch; ++b) {
        const int64_t z = (int64_t)plane * batch + b;
        const float* xp = x + (z * C + c) * hw;
        const float* gp = gy + (z * C + c) * hw;
        float* gxp = gx + (z * C + c) * hw;
        for (int64_t i0 = ((int64_t)blockIdx.x * 4 + wave) * (16 * NC); i0 < hw; i0 += (int64_t)gridDim.x * 4 * 16 * NC) {
            float xv[NC], gv[NC];
#pragma unroll
            for (int n = 0; n < NC; ++n) {
                const int64_t i = i0 + n * 16 + col;
                xv[n] = xp[i < hw ? i : 0];
                gv[n] = i < hw ? gp[i] : 0.f;            // past the end: no gradient, so every sum below gets zeros
            }
            float h0[NC][8], h1[NC][8], h2[NC][8], d[NC][8], t[NC][8];
#pragma unroll
            for (int n = 0; n < NC; ++n)
#pragma unroll
                for (int q = 0; q < 8; ++q) h0[n][q] = fast_tanh(fmaf(w0b[q], xv[n], b0b[q]));
            LLDWT_MLPW_GEMM(t, A1, h0, b1d)
#pragma unroll
            for (int n = 0; n < NC; ++n)
#pragma unroll
                for (int q = 0; q < 8; ++q) h1[n][q] = fast_tanh(t[n][q]);
            LLDWT_MLPW_GEMM(t, A2, h1, b2d)
#pragma unroll
            for (int n = 0; n < NC; ++n)
#pragma unroll
                for (int q = 0; q < 8; ++q) h2[n][q] = fast_tanh(t[n][q]);
#pragma unroll
            for (int n = 0; n < NC; ++n) {
                if (kk == 0) sdb3 += gv[n];
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    sdw3[q] = fmaf(gv[n], h2[n][q], sdw3[q]);
                    d[n][q] = w3d[q] * gv[n] * (1.f - h2[n][q] * h2[n][q]);           // d2
                    sdb2[q] += d[n][q];
                }
            }
            LLDWT_MLPW_PUT(bufA, d)
            LLDWT_MLPW_PUT(bufB, h1)
            LLDWT_MLPW_OUTER(aw2)
            LLDWT_MLPW_GEMM(t, A2T, d, zero8)
#pragma unroll
            for (int n = 0; n < NC; ++n)
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    d[n][q] = t[n][q] * (1.f - h1[n][q] * h1[n][q]);                    // d1
                    sdb1[q] += d[n][q];
                }
            LLDWT_MLPW_PUT(bufA, d)
            LLDWT_MLPW_PUT(bufB, h0)
            LLDWT_MLPW_OUTER(aw1)
            LLDWT_MLPW_GEMM(t, A1T, d, zero8)
#pragma unroll
            for (int n = 0; n < NC; ++n) {
                float o = 0.f;
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const float d0 = t[n][q] * (1.f - h0[n][q] * h0[n][q]);
                    sdb0[q] += d0;
                    sdw0[q] = fmaf(d0, xv[n], sdw0[q]);
                    o = fmaf(w0b[q], d0, o);
                }
                o += __shfl_xor(o, 16, 64);
                o += __shfl_xor(o, 32, 64);
                const int64_t i = i0 + n * 16 + col;
                if (kk == 0 && i < hw) gxp[i] = o;
            }
        }
    }
#undef LLDWT_MLPW_GEMM
#undef LLDWT_MLPW_PUT
#undef LLDWT_MLPW_OUTER
    // this wave's row of partial sums
    float* row = part + ((pc * gridDim.x + blockIdx.x) * 4 + wave) * (int64_t)MLPW_ROW;
#pragma unroll
    for (int mo = 0; mo < 2; ++mo)
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int j = (mo * 16 + 4 * kk + r) * HD + mi * 16 + col;
                row[j] = aw1[mo][mi][r];
                row[1024 + j] = aw2[mo][mi][r];
            }
#pragma unroll
    for (int q = 0; q < 8; ++q) {                       // sums over the 16 coefficient columns of the lane's channel
        float v[5] = {sdw0[q], sdb0[q], sdb1[q], sdb2[q], sdw3[q]};
#pragma unroll
        for (int k = 0; k < 5; ++k) {
#pragma unroll
            for (int o = 1; o < 16; o <<= 1) v[k] += __shfl_xor(v[k], o, 64);
            if (col == 0) row[2048 + 32 * k + (q >> 2) * 16 + 4 * kk + (q & 3)] = v[k];
        }
    }
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) sdb3 += __shfl_xor(sdb3, o, 64);
    if (lane == 0) row[2048 + 160] = sdb3;
}

// rows of partial sums -> the eight gradient tensors of one (plane, channel): out = alpha-free plain sums in row order
__global__ __launch_bounds__(256) void k_subband_mlp_wsum(const float* __restrict__ part, int nrows, float* __restrict__ dw0,
                                                          float* __restrict__ db0, float* __restrict__ dw1,
                                                          float* __restrict__ db1, float* __restrict__ dw2,
                                                          float* __restrict__ db2, float* __restrict__ dw3,
                                                          float* __restrict__ db3) {
    const int64_t pc = blockIdx.y;
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= 2048 + 161) return;
    const float* p = part + pc * nrows * (int64_t)MLPW_ROW + j;
    float s = 0.f;
    for (int r = 0; r < nrows; ++r) s += p[(int64_t)r * MLPW_ROW];
    if (j < 1024) dw1[pc * 1024 + j] = s;
    else if (j < 2048) dw2[pc * 1024 + j - 1024] = s;
    else {
        const int k = (j - 2048) >> 5, ch = (j - 2048) & 31;
        float* dst = k == 0 ? dw0 : k == 1 ? db0 : k == 2 ? db1 : k == 3 ? db2 : k == 4 ? dw3 : db3;
        if (k < 5) dst[pc * 32 + ch] = s;
        else if (ch == 0) dst[pc] = s;
    }
}

// ------------------------------------------------------------------------------------------ direct conv (reference order)
constexpr int OCB = 8;
__global__ __launch_bounds__(256) void k_conv_direct(const float* __restrict__ x, float* __restrict__ y,
                                                     const float* __restrict__ w, const float* __restrict__ bias,
                                                     lldwt_conv_desc d, int batch, int h, int wd) {
    const int K = d.K, P = K / 2, KK = K * K;
    const int cin_g = d.cin / d.groups, cout_g = d.cout / d.groups;
    const int64_t z = blockIdx.z;
    const int plane = (int)(z / batch);
    const int nblk = (cout_g + OCB - 1) / OCB;       // blockIdx.y = group * nblk + block-in-group
    const int g = blockIdx.y / nblk;
    const int ocl0 = (blockIdx.y - g * nblk) * OCB;  // first output channel inside the group
    const int oc0 = g * cout_g + ocl0;
    const int64_t hw = (int64_t)h * wd;
    const int hi = d.upsample2 ? h / 2 : h, wi = d.upsample2 ? wd / 2 : wd;
    const int64_t hwi = (int64_t)hi * wi;
    const float* wp = w + (int64_t)plane * d.cout * cin_g * KK;
    const float* xz = x + (z * d.cin + (int64_t)g * cin_g) * hwi;
    for (int64_t p = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; p < hw; p += (int64_t)gridDim.x * blockDim.x) {
        const int py = (int)(p / wd), px = (int)(p - (int64_t)py * wd);
        float acc[OCB];
#pragma unroll
        for (int o = 0; o < OCB; ++o) acc[o] = (bias && ocl0 + o < cout_g) ? bias[plane * d.cout + oc0 + o] : 0.f;
        for (int ic = 0; ic < cin_g; ++ic) {
            for (int t = 0; t < KK; ++t) {
                if (!((d.tap_mask >> t) & 1u)) continue;
                const int ky = t / K, kx = t - ky * K;
                const int yy = py + ky - P, xx = px + kx - P;
                float v = 0.f;
                if (yy >= 0 && yy < h && xx >= 0 && xx < wd) {
                    const int sy = d.upsample2 ? (yy >> 1) : yy, sx = d.upsample2 ? (xx >> 1) : xx;
                    v = xz[ic * hwi + (int64_t)sy * wi + sx];
                }
#pragma unroll
                for (int o = 0; o < OCB; ++o) {
                    const int oc = oc0 + o;
                    if (ocl0 + o < cout_g) {
                        const float wv = d.transposed
                                             ? wp[((int64_t)(g * cin_g + ic) * cout_g + (ocl0 + o)) * KK + (KK - 1 - t)]
                                             : wp[((int64_t)oc * cin_g + ic) * KK + t];
                        acc[o] = fmaf(wv, v, acc[o]);
                    }
                }
            }
        }
#pragma unroll
        for (int o = 0; o < OCB; ++o) {
            const int oc = oc0 + o;
            if (ocl0 + o < cout_g) {
                const int ocp = (oc / d.oc_block) * d.oc_stride + d.oc_off + oc % d.oc_block;
                y[(z * d.ytot + ocp) * hw + p] = act_apply(acc[o], d.act);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------ GDN
// graphs/layers/gdn.py:77-92 with utils/parametrizers.py:45-48 applied to beta/gamma in-kernel.
__device__ __forceinline__ float nonneg(float v, float bound, float pedestal) {
    const float m = fmaxf(v, bound);
    return m * m - pedestal;
}

__global__ __launch_bounds__(256) void k_gdn(const float* __restrict__ x, float* __restrict__ y,
                                             const float* __restrict__ beta, const float* __restrict__ gamma, int batch,
                                             int C, int64_t hw, int inverse, float beta_bound, float gamma_bound,
                                             float pedestal) {
    const int64_t z = blockIdx.z;
    const int plane = (int)(z / batch);
    const int c = blockIdx.y;
    const float* gp = gamma + ((int64_t)plane * C + c) * C;
    const float bt = nonneg(beta[plane * C + c], beta_bound, pedestal);
    const float* xz = x + z * C * hw;
    for (int64_t p = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; p < hw; p += (int64_t)gridDim.x * blockDim.x) {
        float nrm = bt;
        for (int j = 0; j < C; ++j) {
            const float v = xz[j * hw + p];
            nrm = fmaf(nonneg(gp[j], gamma_bound, pedestal), v * v, nrm);
        }
        nrm = inverse ? sqrtf(nrm) : 1.0f / sqrtf(nrm);
        y[(z * C + c) * hw + p] = xz[c * hw + p] * nrm;
    }
}

// ------------------------------------------------------------------------------------------ bound ops
// utils/bound_ops.py:22-28, utils/parametrizers.py:45-48
__global__ void k_lower_bound_fwd(const float* __restrict__ x, float* __restrict__ y, int64_t n, float b) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        y[i] = fmaxf(x[i], b);
}
__global__ void k_lower_bound_bwd(const float* __restrict__ x, const float* __restrict__ gy, float* __restrict__ gx,
                                  int64_t n, float b) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float g = gy[i];
        gx[i] = ((x[i] >= b) || (g < 0.f)) ? g : 0.f;
    }
}
__global__ void k_nonneg_fwd(const float* __restrict__ x, float* __restrict__ y, int64_t n, float b, float ped) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        y[i] = nonneg(x[i], b, ped);
}
__global__ void k_nonneg_bwd(const float* __restrict__ x, const float* __restrict__ gy, float* __restrict__ gx,
                             int64_t n, float b) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        // d/dx [lower_bound(x)^2 - ped] = 2*lower_bound(x) * LowerBound'(x)  with the pass-through rule on the inner grad
        const float xv = x[i];
        const float g = gy[i] * 2.f * fmaxf(xv, b);
        gx[i] = ((xv >= b) || (g < 0.f)) ? g : 0.f;
    }
}

// ------------------------------------------------------------------------------------------ block reduce -> double atomic
__device__ __forceinline__ void block_accumulate(double v, double* out) {
    __shared__ double part[16];
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (lane == 0) part[wv] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        double s = 0;
        for (int i = 0; i < (int)(blockDim.x >> 6); ++i) s += part[i];
        atomicAdd(out, s);
    }
}

// ------------------------------------------------------------------------------------------ Gaussian rate
// erfc without branches (the library erfcf picks one of several range forms per lane: a wave with mixed arguments runs them all,
// ~70 vector instructions; this is ~35).  For a = |x|: (1 + 2a) exp(a^2) erfc(a) - 1 is a smooth function of q = (a - 2) / (a + 2)
// on [-1, 0.668] (a <= 10.05, beyond which erfc underflows fp32); a degree-9 polynomial fitted at 400 Chebyshev nodes against
// scipy.special.erfcx in double (fit error 1.6e-8 relative).  The two reciprocals are the hardware's (1 ulp) followed by one
// residual step each; exp(-a^2) = 2^(-h) (1 - l ln 2) with h + l = a^2 log2(e) carried in two floats and (a^2 rounded) - a^2
// folded in at the end.  Relative error <= 2.3e-7 over [0, 9.1] (2 M points against scipy erfc; erfc(9.1) = 7e-38).
__device__ __forceinline__ float erfc_fast(float x) {
    const float a = fabsf(x);
    float r = __builtin_amdgcn_rcpf(a + 2.f);
    float q = __builtin_fmaf(-4.f, r, 1.f);                        // (a - 2) / (a + 2) = 1 - 4 / (a + 2)
    const float t0 = __builtin_fmaf(q + 1.f, -2.f, a);
    q = __builtin_fmaf(r, __builtin_fmaf(-a, q, t0), q);            // + residual / (a + 2)
    float p = -4.095828658e-04f;
    p = __builtin_fmaf(p, q, -1.241535299e-03f);
    p = __builtin_fmaf(p, q, 1.320674849e-03f);
    p = __builtin_fmaf(p, q, 8.642993991e-03f);
    p = __builtin_fmaf(p, q, -8.061684272e-03f);
    p = __builtin_fmaf(p, q, -5.420781631e-02f);
    p = __builtin_fmaf(p, q, 1.640555406e-01f);
    p = __builtin_fmaf(p, q, -1.660310709e-01f);
    p = __builtin_fmaf(p, q, -9.276399062e-02f);
    p = __builtin_fmaf(p, q, 2.769783880e-01f);
    r = __builtin_amdgcn_rcpf(__builtin_fmaf(2.f, a, 1.f));
    const float g = __builtin_fmaf(p, r, r);                       // (p + 1) / (1 + 2a)
    const float e = __builtin_fmaf(__builtin_fmaf(g, -a, 0.5f), 2.f, p - g);
    const float sc = __builtin_fmaf(e, r, g);                      // exp(a^2) erfc(a)
    const float s = a * a;
    const float ds = __builtin_fmaf(-a, a, s);                     // s - a^2, exact
    const float L = 1.4426950408889634f;
    const float h = s * L;
    float l = __builtin_fmaf(s, L, -h);
    l = __builtin_fmaf(s, 1.9259629911e-08f, l);                   // log2(e) - (float)log2(e)
    float ex = __builtin_amdgcn_exp2f(-h);
    ex = __builtin_fmaf(-ex, l * 0.6931471805599453f, ex);
    float res = __builtin_fmaf(sc, ex, sc * ex * ds);
    res = a > 10.0546875f ? 0.f : res;
    return x < 0.f ? 2.f - res : res;
}

// compressai GaussianConditional.forward/_likelihood as called at LiftingBasedDWT_net.py:334,345,364,832
// grid: x over the pixels of one channel plane (4 per lane when aligned), y over (image, channel) -- no per-element
// division (the first version's two 64-bit divisions per element cost more than the two erfc)
// MODE: 1 = training (noise read), 2 = bits written, 4 = q written -- compile-time, so that the loop's waits count its own stores
template <int MODE>
__global__ __launch_bounds__(256) void k_gauss_rate(const float* __restrict__ x, const float* __restrict__ params,
                                                    const float* __restrict__ noise, float* __restrict__ bits,
                                                    float* __restrict__ qout, double* __restrict__ bit_sum, int C,
                                                    int64_t hw, int64_t ZC) {
    constexpr bool train = (MODE & 1) != 0, WB = (MODE & 2) != 0, WQ = (MODE & 4) != 0;
    double local = 0;
    auto one = [&](float xv, float sg, float mu, float nz, bool train, float& b, float& v) {
        v = train ? xv + nz : rintf(xv - mu) + mu;
        const float a = fabsf(v - mu);
        const float s = fmaxf(sg, 0.11f);
        // one reciprocal (v_rcp_f32 + a Newton step: < 1 ulp) instead of two IEEE divisions; the hardware log2 (1 ulp): the
        // likelihood moves by a few 1e-7 relative, the bits by < 1e-6 (parity bar 1e-4)
        float rs = __builtin_amdgcn_rcpf(s);
        rs = __builtin_fmaf(__builtin_fmaf(-s, rs, 1.f), rs, rs);
        const float k = -0.70710678118654752440f * rs;
        const float up = 0.5f * erfc_fast(k * (0.5f - a));
        const float lo = 0.5f * erfc_fast(k * (-0.5f - a));
        const float lik = fmaxf(up - lo, 1e-9f);
        b = -__builtin_amdgcn_logf(lik);
    };
    for (int64_t zc = blockIdx.y; zc < ZC; zc += gridDim.y) {
        const int64_t z = zc / C;
        const int c = (int)(zc - z * C);
        const float* ps = params + (z * 2 * C + 2 * c) * hw;
        const float* pm = ps + hw;
        const float* xp = x + zc * hw;
        const float* np = train ? noise + zc * hw : nullptr;
        float* bp = WB ? bits + zc * hw : nullptr;
        float* qp = WQ ? qout + zc * hw : nullptr;
        const bool vec = (hw & 3) == 0 && ((((uintptr_t)xp) | ((uintptr_t)ps) | ((uintptr_t)(bp ? bp : xp)) |
                                            ((uintptr_t)(qp ? qp : xp)) | ((uintptr_t)(np ? np : xp))) & 15) == 0;
        if (vec) {
            // one 16-byte group per lane and iteration, the NEXT group's three (four) loads issued before this group's arithmetic:
            // a launch at the BASELINE batch is a single resident round whose waves all start together, so without the look-ahead
            // the chip spends a load phase (no arithmetic) and then an arithmetic phase (no loads) -- the sum of the two, not the max
            const int64_t n4 = hw >> 2, stride = (int64_t)gridDim.x * blockDim.x;
            int64_t p = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
            float4 xa = {0.f, 0.f, 0.f, 0.f}, sa = xa, ma = xa, na = xa;
            if (p < n4) {
                xa = reinterpret_cast<const float4*>(xp)[p];
                sa = reinterpret_cast<const float4*>(ps)[p];
                ma = reinterpret_cast<const float4*>(pm)[p];
                if (train) na = reinterpret_cast<const float4*>(np)[p];
            }
            while (p < n4) {
                const int64_t pn = p + stride;
                float4 xb = xa, sb = sa, mb = ma, nb = na;
                if (pn < n4) {
                    xb = reinterpret_cast<const float4*>(xp)[pn];
                    sb = reinterpret_cast<const float4*>(ps)[pn];
                    mb = reinterpret_cast<const float4*>(pm)[pn];
                    if (train) nb = reinterpret_cast<const float4*>(np)[pn];
                }
                float4 b, v;
                one(xa.x, sa.x, ma.x, na.x, train, b.x, v.x);
                one(xa.y, sa.y, ma.y, na.y, train, b.y, v.y);
                one(xa.z, sa.z, ma.z, na.z, train, b.z, v.z);
                one(xa.w, sa.w, ma.w, na.w, train, b.w, v.w);
                if (WB) reinterpret_cast<float4*>(bp)[p] = b;
                if (WQ) reinterpret_cast<float4*>(qp)[p] = v;
                local += (double)((b.x + b.y) + (b.z + b.w));      // four values of at most ~30 bits each: fp32 partial, fp64 running sum
                xa = xb; sa = sb; ma = mb; na = nb;
                p = pn;
            }
        } else {
            for (int64_t p = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; p < hw; p += (int64_t)gridDim.x * blockDim.x) {
                float b, v;
                one(xp[p], ps[p], pm[p], train ? np[p] : 0.f, train, b, v);
                if (WB) bp[p] = b;
                if (WQ) qp[p] = v;
                local += (double)b;
            }
        }
    }
    if (bit_sum) block_accumulate(local, bit_sum);
}

// backward of k_gauss_rate (training: v = x + noise).  With p = Phi(u) - Phi(l), u = (.5-a)/s, l = (-.5-a)/s, a = |v-mu|,
// s = LowerBound(0.11)(sigma), bits = -log2(LowerBound(1e-9)(p)); both LowerBounds use the pass-through rule of
// utils/bound_ops.py:26-28.  dparams gets (dsigma, dmu) interleaved like params.
__global__ __launch_bounds__(256) void k_gauss_rate_bwd(const float* __restrict__ x, const float* __restrict__ params,
                                                        const float* __restrict__ noise, const float* __restrict__ gbits,
                                                        float* __restrict__ dx, float* __restrict__ dparams, int C,
                                                        int64_t hw, int64_t n) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t zc = i / hw, p = i - zc * hw;
        const int64_t z = zc / C;
        const int c = (int)(zc - z * C);
        const int64_t is = (z * 2 * C + 2 * c) * hw + p, im = is + hw;
        const float sg = params[is], mu = params[im];
        const float xv = x[i];
        const bool train = noise != nullptr;
        const float v = train ? xv + noise[i] : rintf(xv - mu) + mu;
        const float dlt = v - mu;
        const float a = fabsf(dlt);
        const float s = fmaxf(sg, 0.11f);
        const float u = (0.5f - a) / s, l = (-0.5f - a) / s;
        const float cst = -0.70710678118654752440f;
        const float pr = 0.5f * erfcf(cst * u) - 0.5f * erfcf(cst * l);
        const float pb = fmaxf(pr, 1e-9f);
        const float g = gbits[i];
        float gp = -g / (pb * 0.69314718055994530942f);            // d bits / d p_bounded
        if (!(pr >= 1e-9f || gp < 0.f)) gp = 0.f;                  // LowerBound(1e-9) gradient rule
        const float inv = 0.39894228040143267794f;                 // 1/sqrt(2 pi)
        const float phu = inv * expf(-0.5f * u * u), phl = inv * expf(-0.5f * l * l);
        const float dpda = (phl - phu) / s;
        float gs = gp * (l * phl - u * phu) / s;                   // d/ds
        if (!(sg >= 0.11f || gs < 0.f)) gs = 0.f;                  // LowerBound(0.11) gradient rule
        const float sgn = dlt > 0.f ? 1.f : (dlt < 0.f ? -1.f : 0.f);
        const float gv = train ? gp * dpda * sgn : 0.f;            // eval: round() has zero gradient, v - mu is constant
        dx[i] = gv;
        dparams[is] = gs;
        dparams[im] = -gv;
    }
}

__global__ void k_quantize(const float* __restrict__ x, const float* __restrict__ noise, float* __restrict__ q, int64_t n) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        q[i] = noise ? x[i] + noise[i] : rintf(x[i]);
}

// ------------------------------------------------------------------------------------------ factorized rate
// compressai EntropyBottleneck.forward/_logits_cumulative/_likelihood (call sites LiftingBasedDWT_net.py:225,229,815,818)
__device__ __forceinline__ float softplusf(float v) { return v > 20.f ? v : log1pf(expf(v)); }   // torch softplus, threshold 20

struct EbParams {   // processed (softplus / tanh applied)
    float m0[3], b0[3], f0[3], m1[9], b1[3], f1[3], m2[9], b2[3], f2[3], m3[9], b3[3], f3[3], m4[3], b4, median;
};

__device__ __forceinline__ float eb_logits(const float* __restrict__ e, float v) {
    // e: processed params in LDS, same order as the packed layout
    float l[3], t[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        l[j] = e[j] * v + e[3 + j];
        l[j] += e[6 + j] * tanhf(l[j]);
    }
    const float* q = e + 9;
#pragma unroll
    for (int layer = 0; layer < 3; ++layer) {
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            float a = q[j * 3 + 0] * l[0];
            a += q[j * 3 + 1] * l[1];
            a += q[j * 3 + 2] * l[2];
            a += q[9 + j];
            t[j] = a + q[12 + j] * tanhf(a);
        }
        l[0] = t[0]; l[1] = t[1]; l[2] = t[2];
        q += 15;
    }
    float o = q[0] * l[0];
    o += q[1] * l[1];
    o += q[2] * l[2];
    return o + q[3];
}

// raw packed parameters of one (plane, channel) -> the processed ones in LDS (softplus of the matrices, tanh of the factors)
__device__ __forceinline__ void eb_process(const float* __restrict__ src, float* e) {
    if (threadIdx.x < LLDWT_EB_FLOATS) {
        const int i = threadIdx.x;
        float v = src[i];
        // layout: m0(3) b0(3) f0(3) | m1(9) b1(3) f1(3) | m2 .. | m3 .. | m4(3) b4(1) median(1)
        bool is_m, is_f;
        if (i < 9) { is_m = i < 3; is_f = i >= 6; }
        else if (i < 54) { const int r = (i - 9) % 15; is_m = r < 9; is_f = r >= 12; }
        else { is_m = i < 57; is_f = false; }
        if (is_m) v = softplusf(v);
        else if (is_f) v = tanhf(v);
        e[i] = v;
    }
}
__device__ __forceinline__ float eb_bits(const float* e, float v) {
    const float lower = eb_logits(e, v - 0.5f);
    const float upper = eb_logits(e, v + 0.5f);
    const float sm = lower + upper;
    const float sign = sm > 0.f ? -1.f : (sm < 0.f ? 1.f : 0.f);
    const float su = 1.f / (1.f + expf(-sign * upper));
    const float sl = 1.f / (1.f + expf(-sign * lower));
    const float lik = fmaxf(fabsf(su - sl), 1e-9f);
    return -log2f(lik);
}
constexpr int EB_TR = 127;                                       // integer offsets -127 .. 127 around the median come from a table

// the eval table of every (plane, channel): bits of median + o for o = -127 .. 127 (256 floats per row, the last unused).
// Depends on the parameters only: the host keeps it while they are unchanged (compressai's update() keeps its CDFs the same way)
__global__ __launch_bounds__(256) void k_factorized_table(const float* __restrict__ eb, float* __restrict__ table) {
    __shared__ float e[LLDWT_EB_FLOATS + 5];
    eb_process(eb + (int64_t)blockIdx.x * LLDWT_EB_FLOATS, e);
    __syncthreads();
    table[(int64_t)blockIdx.x * 256 + threadIdx.x] =
        threadIdx.x < 2 * EB_TR + 1 ? eb_bits(e, (float)((int)threadIdx.x - EB_TR) + e[58]) : 0.f;
}

// WB / WQ: bits / q written -- compile-time, so that the eval loop's waits count its own stores
template <bool WB, bool WQ>
__global__ __launch_bounds__(256) void k_factorized_rate(const float* __restrict__ x, const float* __restrict__ eb,
                                                         const float* __restrict__ noise, float* __restrict__ bits,
                                                         float* __restrict__ qout, double* __restrict__ bit_sum,
                                                         int batch, int C, int64_t hw, const float* __restrict__ table) {
    __shared__ float e[LLDWT_EB_FLOATS + 5];
    const int c = blockIdx.y;
    const int64_t z = blockIdx.z;
    const int plane = (int)(z / batch);
    // eval, 16-byte path: the first group of the stream is requested BEFORE the parameter / table prologue (a global round trip, the
    // softplus / tanh of the parameters and a barrier that would otherwise stand in front of the first load of every workgroup)
    const int64_t base0 = (z * C + c) * hw;
    const bool vec = (hw & 3) == 0 && ((((uintptr_t)(x + base0)) | ((uintptr_t)(WB ? bits + base0 : x + base0)) |
                                        ((uintptr_t)(WQ ? qout + base0 : x + base0))) & 15) == 0;
    const int64_t n4 = hw >> 2, p0 = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    float4 xa = {0.f, 0.f, 0.f, 0.f};
    if (!noise && vec && p0 < n4) xa = reinterpret_cast<const float4*>(x + base0)[p0];
    eb_process(eb + ((int64_t)plane * C + c) * LLDWT_EB_FLOATS, e);
    constexpr int TR = EB_TR;
    __shared__ float tab[2 * TR + 1];
    if (table && threadIdx.x < 2 * TR + 1) tab[threadIdx.x] = table[((int64_t)plane * C + c) * 256 + threadIdx.x];
    __syncthreads();
    const float med = e[58];
    const int64_t base = (z * C + c) * hw;
    double local = 0;
    auto bits_of = [&](float v) { return eb_bits(e, v); };
    if (!noise) {
        // Eval: v = round(x - median) + median takes one value per integer offset, so the 24 tanh + 2 exp + log of the
        // chain are evaluated ONCE per offset and channel into an LDS table (the same arithmetic as the direct path, so the
        // values are identical) and the element loop is load -> round -> table -> store: HBM-bound, 16 bytes per lane.
        // With a precomputed table (k_factorized_table; lldwt_factorized_rate_tab) the workgroup only copies its row.
        if (!table) {
            if (threadIdx.x < 2 * TR + 1) tab[threadIdx.x] = bits_of((float)((int)threadIdx.x - TR) + med);
            __syncthreads();
        }
        auto one = [&](float xv, float& b, float& q) {
            const float r = rintf(xv - med);
            q = r + med;
            b = fabsf(r) <= (float)TR ? tab[(int)r + TR] : bits_of(q);
        };
        if (vec) {
            const float4* x4 = reinterpret_cast<const float4*>(x + base);
            float4* b4 = WB ? reinterpret_cast<float4*>(bits + base) : nullptr;
            float4* q4 = WQ ? reinterpret_cast<float4*>(qout + base) : nullptr;
            const int64_t stride = (int64_t)gridDim.x * blockDim.x;
            for (int64_t p = p0; p < n4;) {                      // the next group's load in flight under this group's look-ups
                const int64_t pn = p + stride;
                float4 xb = xa;
                if (pn < n4) xb = x4[pn];
                float4 b, q;
                one(xa.x, b.x, q.x);
                one(xa.y, b.y, q.y);
                one(xa.z, b.z, q.z);
                one(xa.w, b.w, q.w);
                if (WB) b4[p] = b;
                if (WQ) q4[p] = q;
                local += (double)b.x + (double)b.y + (double)b.z + (double)b.w;
                xa = xb;
                p = pn;
            }
        } else {
            for (int64_t p = p0; p < hw; p += (int64_t)gridDim.x * blockDim.x) {
                float b, q;
                one(x[base + p], b, q);
                if (WB) bits[base + p] = b;
                if (WQ) qout[base + p] = q;
                local += (double)b;
            }
        }
    } else {
        for (int64_t p = p0; p < hw; p += (int64_t)gridDim.x * blockDim.x) {
            const float v = x[base + p] + noise[base + p];
            const float b = bits_of(v);
            if (WB) bits[base + p] = b;
            if (WQ) qout[base + p] = v;
            local += (double)b;
        }
    }
    if (bit_sum) block_accumulate(local, bit_sum);
}

// ---- backward of the factorized rate (training: v = x + noise) -------------------------------------------------------
// forward chain on processed params e (softplus / tanh applied), keeping what the backward needs
struct EbTrace {
    float a[4][3];    // pre-gate activations of layers 0..3
    float h[4][3];    // gated outputs
};

__device__ __forceinline__ float eb_logits_trace(const float* __restrict__ e, float v, EbTrace& tr) {
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        tr.a[0][j] = e[j] * v + e[3 + j];
        tr.h[0][j] = tr.a[0][j] + e[6 + j] * tanhf(tr.a[0][j]);
    }
    const float* q = e + 9;
#pragma unroll
    for (int layer = 1; layer < 4; ++layer) {
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            float a = q[j * 3 + 0] * tr.h[layer - 1][0];
            a += q[j * 3 + 1] * tr.h[layer - 1][1];
            a += q[j * 3 + 2] * tr.h[layer - 1][2];
            a += q[9 + j];
            tr.a[layer][j] = a;
            tr.h[layer][j] = a + q[12 + j] * tanhf(a);
        }
        q += 15;
    }
    return q[0] * tr.h[3][0] + q[1] * tr.h[3][1] + q[2] * tr.h[3][2] + q[3];
}

// accumulates d(out)/d(processed params) * g into ge[59] (same layout as e) and returns d(out)/dv * g
__device__ __forceinline__ float eb_logits_bwd(const float* __restrict__ e, float v, const EbTrace& tr, float g, float* ge) {
    float gh[3];
    const float* q = e + 54;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        ge[54 + k] += tr.h[3][k] * g;
        gh[k] = q[k] * g;
    }
    ge[57] += g;
#pragma unroll
    for (int layer = 3; layer >= 1; --layer) {
        q = e + 9 + 15 * (layer - 1);
        float* gq = ge + 9 + 15 * (layer - 1);
        float ga[3], ghp[3] = {0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const float th = tanhf(tr.a[layer][j]);
            ga[j] = gh[j] * (1.f + q[12 + j] * (1.f - th * th));
            gq[12 + j] += gh[j] * th;
            gq[9 + j] += ga[j];
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                gq[j * 3 + k] += ga[j] * tr.h[layer - 1][k];
                ghp[k] += q[j * 3 + k] * ga[j];
            }
        }
        gh[0] = ghp[0]; gh[1] = ghp[1]; gh[2] = ghp[2];
    }
    float gv = 0.f;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const float th = tanhf(tr.a[0][j]);
        const float ga = gh[j] * (1.f + e[6 + j] * (1.f - th * th));
        ge[6 + j] += gh[j] * th;
        ge[3 + j] += ga;
        ge[j] += ga * v;
        gv += e[j] * ga;
    }
    return gv;
}

// grads wrt x (Z,C,hw) and wrt the RAW packed EntropyBottleneck parameters deb (planes,C,59) (atomics; zero it first).
__global__ __launch_bounds__(256) void k_factorized_rate_bwd(const float* __restrict__ x, const float* __restrict__ eb,
                                                             const float* __restrict__ noise, const float* __restrict__ gbits,
                                                             float* __restrict__ dx, float* __restrict__ deb, int batch,
                                                             int C, int64_t hw) {
    __shared__ float e[LLDWT_EB_FLOATS + 5];
    __shared__ float dfac[LLDWT_EB_FLOATS + 5];     // d processed / d raw (sigmoid for softplus, 1 - t^2 for tanh, 1 else)
    __shared__ float red[4][LLDWT_EB_FLOATS + 1];
    const int c = blockIdx.y;
    const int64_t z = blockIdx.z;
    const int plane = (int)(z / batch);
    const float* src = eb + ((int64_t)plane * C + c) * LLDWT_EB_FLOATS;
    if (threadIdx.x < LLDWT_EB_FLOATS) {
        const int i = threadIdx.x;
        float v = src[i], f = 1.f;
        bool is_m, is_f;
        if (i < 9) { is_m = i < 3; is_f = i >= 6; }
        else if (i < 54) { const int r = (i - 9) % 15; is_m = r < 9; is_f = r >= 12; }
        else { is_m = i < 57; is_f = false; }
        if (is_m) { f = 1.f / (1.f + expf(-v)); v = softplusf(v); }
        else if (is_f) { v = tanhf(v); f = 1.f - v * v; }
        if (i == 58) f = 0.f;      // the median does not enter the training-mode forward
        e[i] = v;
        dfac[i] = f;
    }
    __syncthreads();
    const int64_t base = (z * C + c) * hw;
    float ge[LLDWT_EB_FLOATS];
#pragma unroll
    for (int i = 0; i < LLDWT_EB_FLOATS; ++i) ge[i] = 0.f;
    for (int64_t p = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; p < hw; p += (int64_t)gridDim.x * blockDim.x) {
        const float v = x[base + p] + (noise ? noise[base + p] : 0.f);
        EbTrace tl, tu;
        const float lower = eb_logits_trace(e, v - 0.5f, tl);
        const float upper = eb_logits_trace(e, v + 0.5f, tu);
        const float sm = lower + upper;
        const float sign = sm > 0.f ? -1.f : (sm < 0.f ? 1.f : 0.f);
        const float su = 1.f / (1.f + expf(-sign * upper)), sl = 1.f / (1.f + expf(-sign * lower));
        const float diff = su - sl;
        const float pr = fabsf(diff), pb = fmaxf(pr, 1e-9f);
        float gp = -gbits[base + p] / (pb * 0.69314718055994530942f);
        if (!(pr >= 1e-9f || gp < 0.f)) gp = 0.f;                       // LowerBound(1e-9) rule
        const float sd = diff > 0.f ? 1.f : (diff < 0.f ? -1.f : 0.f);
        const float gu = gp * sd * su * (1.f - su) * sign;
        const float gl = -gp * sd * sl * (1.f - sl) * sign;
        float gv = eb_logits_bwd(e, v + 0.5f, tu, gu, ge);
        gv += eb_logits_bwd(e, v - 0.5f, tl, gl, ge);
        dx[base + p] = noise ? gv : 0.f;                                 // eval: round() has zero gradient
    }
    // block reduction of the 58 parameter partials, then one atomic per parameter
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < LLDWT_EB_FLOATS - 1; ++i) {
        float v = ge[i];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
        if (lane == 0) red[wv][i] = v;
    }
    __syncthreads();
    if (threadIdx.x < LLDWT_EB_FLOATS - 1) {
        const int i = threadIdx.x;
        const float s = (red[0][i] + red[1][i] + red[2][i] + red[3][i]) * dfac[i];
        atomicAdd(deb + ((int64_t)plane * C + c) * LLDWT_EB_FLOATS + i, s);
    }
}

__global__ __launch_bounds__(256) void k_sq_err_sum(const float* __restrict__ a, const float* __restrict__ b, int64_t n,
                                                    double* __restrict__ out) {
    double local = 0;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float d = a[i] - b[i];
        local += (double)d * d;
    }
    block_accumulate(local, out);
}
__global__ __launch_bounds__(256) void k_sum(const float* __restrict__ x, int64_t n, double* __restrict__ out) {
    double local = 0;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        local += (double)x[i];
    block_accumulate(local, out);
}

static inline unsigned ew_grid(int64_t n) {
    int64_t g = cdiv(n, 256);
    return (unsigned)(g < 1 ? 1 : (g > 2048 ? 2048 : g));
}

}  // namespace lldwt

using namespace lldwt;

extern "C" const char* lldwt_last_error(void) { return g_err; }
extern "C" int lldwt_version(void) { return 100; }
extern "C" int lldwt_device_ok(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return 0;
    hipDeviceProp_t p;
    if (hipGetDeviceProperties(&p, 0) != hipSuccess) return 0;
    return strncmp(p.gcnArchName, "gfx950", 6) == 0 ? 1 : 0;
}

extern "C" int lldwt_rgb_to_ycc(const float* rgb, float* ycc, int64_t B, int64_t H, int64_t W, void* stream) {
    LLDWT_REQUIRE(rgb && ycc && B > 0 && H > 0 && W > 0, "rgb_to_ycc: bad arguments");
    hipLaunchKernelGGL(k_rgb_to_ycc, dim3(ew_grid(B * H * W)), dim3(256), 0, (hipStream_t)stream, rgb, ycc, B, H * W);
    return check_launch("rgb_to_ycc");
}
extern "C" int lldwt_u8hwc_to_f32chw(const uint8_t* src, float* dst, int64_t B, int64_t H, int64_t W, void* stream) {
    LLDWT_REQUIRE(src && dst && B > 0 && H > 0 && W > 0, "u8hwc_to_f32chw: bad arguments");
    hipLaunchKernelGGL(k_u8hwc_to_f32chw, dim3(ew_grid(B * H * W)), dim3(256), 0, (hipStream_t)stream, src, dst, B, H * W);
    return check_launch("u8hwc_to_f32chw");
}
extern "C" int lldwt_ycc_to_rgb(const float* ycc, float* rgb, int64_t B, int64_t H, int64_t W, int clamp, void* stream) {
    LLDWT_REQUIRE(rgb && ycc && B > 0 && H > 0 && W > 0, "ycc_to_rgb: bad arguments");
    hipLaunchKernelGGL(k_ycc_to_rgb, dim3(ew_grid(B * H * W)), dim3(256), 0, (hipStream_t)stream, ycc, rgb, B, H * W, clamp);
    return check_launch("ycc_to_rgb");
}

extern "C" int lldwt_subband_mlp(const float* x, float* y, int64_t planes, int64_t batch, int C, int64_t hw, int Hd,
                                 const float* w0, const float* b0, const float* w1, const float* b1, const float* w2,
                                 const float* b2, const float* w3, const float* b3, int transposed, void* stream) {
    LLDWT_REQUIRE(x && y && w0 && b0 && w1 && b1 && w2 && b2 && w3 && b3, "subband_mlp: null pointer");
    LLDWT_REQUIRE(planes > 0 && batch > 0 && C > 0 && hw > 0, "subband_mlp: bad dims");
    LLDWT_REQUIRE(Hd == 32, "subband_mlp: hidden width %d unsupported (reference uses H=32, lifting_dwt_nets.py:98)", Hd);
    LLDWT_REQUIRE(planes * batch <= 65535 && C <= 65535, "subband_mlp: grid too large");
    int64_t gx = cdiv(hw, 256);
    if (gx > 1024) gx = 1024;
    dim3 grid((unsigned)gx, (unsigned)C, (unsigned)(planes * batch));
    hipLaunchKernelGGL(k_subband_mlp_mfma, grid, dim3(256), 0, (hipStream_t)stream, x, y, (int)batch, C, hw, w0, b0, w1, b1,
                       w2, b2, w3, b3, transposed);
    return check_launch("subband_mlp");
}

extern "C" int lldwt_subband_mlp_bwd(const float* x, const float* gy, float* gx, float* h0, float* h1, float* h2, float* d0,
                                     float* d1, float* d2, int64_t planes, int64_t batch, int C, int64_t hw, int Hd,
                                     const float* w0, const float* b0, const float* w1, const float* b1, const float* w2,
                                     const float* b2, const float* w3, void* stream) {
    LLDWT_REQUIRE(x && gy && gx && h0 && h1 && h2 && d0 && d1 && d2 && w0 && b0 && w1 && b1 && w2 && b2 && w3,
                  "subband_mlp_bwd: null pointer");
    LLDWT_REQUIRE(planes > 0 && batch > 0 && C > 0 && hw > 0, "subband_mlp_bwd: bad dims");
    LLDWT_REQUIRE(Hd == 32, "subband_mlp_bwd: hidden width %d unsupported (reference uses H=32, lifting_dwt_nets.py:98)", Hd);
    LLDWT_REQUIRE(planes * batch <= 65535 && C <= 65535, "subband_mlp_bwd: grid too large");
    int64_t gxn = cdiv(hw, 128);
    if (gxn > 1024) gxn = 1024;
    dim3 grid((unsigned)gxn, (unsigned)C, (unsigned)(planes * batch));
    hipLaunchKernelGGL(k_subband_mlp_bwd, grid, dim3(256), 0, (hipStream_t)stream, x, gy, gx, h0, h1, h2, d0, d1, d2, (int)batch,
                       C, hw, w0, b0, w1, b1, w2, b2, w3);
    return check_launch("subband_mlp_bwd");
}

static inline int64_t mlpw_slices(int64_t planes, int C, int64_t hw) {
    // about two workgroups per CU in all; a slice is at least one round of the four waves (128 coefficients)
    int64_t s = (int64_t)lldwt_num_cus() * 2 / (planes * C);
    const int64_t most = cdiv(hw, 128);
    s = s > most ? most : s;
    return s < 1 ? 1 : s;
}
extern "C" int64_t lldwt_subband_mlp_bwd_w_ws_bytes(int64_t planes, int C, int64_t hw) {
    return (int64_t)sizeof(float) * planes * C * mlpw_slices(planes, C, hw) * 4 * MLPW_ROW;
}
extern "C" int lldwt_subband_mlp_bwd_w(const float* x, const float* gy, float* gx, int64_t planes, int64_t batch, int C, int64_t hw,
                                       int Hd, const float* w0, const float* b0, const float* w1, const float* b1,
                                       const float* w2, const float* b2, const float* w3, float* dw0, float* db0, float* dw1,
                                       float* db1, float* dw2, float* db2, float* dw3, float* db3, void* ws, int64_t ws_bytes,
                                       void* stream) {
    LLDWT_REQUIRE(x && gy && gx && w0 && b0 && w1 && b1 && w2 && b2 && w3 && dw0 && db0 && dw1 && db1 && dw2 && db2 && dw3 && db3 &&
                  ws, "subband_mlp_bwd_w: null pointer");
    LLDWT_REQUIRE(planes > 0 && batch > 0 && C > 0 && hw > 0, "subband_mlp_bwd_w: bad dims");
    LLDWT_REQUIRE(Hd == 32, "subband_mlp_bwd_w: hidden width %d unsupported (reference uses H=32, lifting_dwt_nets.py:98)", Hd);
    LLDWT_REQUIRE(planes <= 65535 && C <= 65535, "subband_mlp_bwd_w: grid too large");
    if (ws_bytes < lldwt_subband_mlp_bwd_w_ws_bytes(planes, C, hw)) {
        set_error("subband_mlp_bwd_w: workspace %ld < %ld bytes", (long)ws_bytes, (long)lldwt_subband_mlp_bwd_w_ws_bytes(planes, C, hw));
        return LLDWT_EWS;
    }
    const int64_t sl = mlpw_slices(planes, C, hw);
    hipLaunchKernelGGL(k_subband_mlp_bwdw, dim3((unsigned)sl, (unsigned)C, (unsigned)planes), dim3(256), 0, (hipStream_t)stream, x,
                       gy, gx, (float*)ws, (int)batch, C, hw, w0, b0, w1, b1, w2, b2, w3);
    hipLaunchKernelGGL(k_subband_mlp_wsum, dim3((unsigned)cdiv(2048 + 161, 256), (unsigned)(planes * C)), dim3(256), 0,
                       (hipStream_t)stream, (const float*)ws, (int)(sl * 4), dw0, db0, dw1, db1, dw2, db2, dw3, db3);
    return check_launch("subband_mlp_bwd_w");
}

extern "C" int lldwt_conv2d_direct(const float* x, float* y, const float* w, const float* bias, const lldwt_conv_desc* d,
                                   int64_t planes, int64_t batch, int64_t h, int64_t w_, void* stream) {
    LLDWT_REQUIRE(x && y && w && d, "conv2d: null pointer");
    LLDWT_REQUIRE(d->K == 1 || d->K == 3 || d->K == 5, "conv2d: K=%d unsupported", d->K);
    LLDWT_REQUIRE(d->groups > 0 && d->cin % d->groups == 0 && d->cout % d->groups == 0, "conv2d: bad groups");
    LLDWT_REQUIRE(d->ic_block == 0, "conv2d_direct: input placement not supported by the cross-check kernel");
    LLDWT_REQUIRE(!d->upsample2 || (h % 2 == 0 && w_ % 2 == 0), "conv2d: upsample2 needs even output dims");
    LLDWT_REQUIRE(d->oc_block > 0 && d->ytot >= d->cout, "conv2d: bad output placement");
    LLDWT_REQUIRE(planes * batch <= 65535, "conv2d: planes*batch exceeds grid.z");
    int64_t gx = cdiv(h * w_, 256);
    if (gx > 4096) gx = 4096;
    dim3 grid((unsigned)gx, (unsigned)(d->groups * cdiv(d->cout / d->groups, OCB)), (unsigned)(planes * batch));
    hipLaunchKernelGGL(k_conv_direct, grid, dim3(256), 0, (hipStream_t)stream, x, y, w, bias, *d, (int)batch, (int)h, (int)w_);
    return check_launch("conv2d_direct");
}

extern "C" int lldwt_gdn(const float* x, float* y, const float* beta, const float* gamma, int64_t planes, int64_t batch,
                         int C, int64_t hw, int inverse, float beta_min, void* stream) {
    LLDWT_REQUIRE(x && y && beta && gamma && planes > 0 && batch > 0 && C > 0 && hw > 0, "gdn: bad arguments");
    LLDWT_REQUIRE(planes * batch <= 65535 && C <= 65535, "gdn: grid too large");
    const float ped = (float)(3.814697265625e-06 * 3.814697265625e-06);       // (2^-18)^2, parametrizers.py:31-36
    const float bb = (float)sqrt((double)beta_min + 3.814697265625e-06 * 3.814697265625e-06);
    const float gb = (float)sqrt(0.0 + 3.814697265625e-06 * 3.814697265625e-06);
    int64_t gx = cdiv(hw, 256);
    if (gx > 1024) gx = 1024;
    dim3 grid((unsigned)gx, (unsigned)C, (unsigned)(planes * batch));
    hipLaunchKernelGGL(k_gdn, grid, dim3(256), 0, (hipStream_t)stream, x, y, beta, gamma, (int)batch, C, hw, inverse, bb, gb, ped);
    return check_launch("gdn");
}

extern "C" int lldwt_lower_bound_fwd(const float* x, float* y, int64_t n, float bound, void* stream) {
    LLDWT_REQUIRE(x && y && n >= 0, "lower_bound_fwd: bad arguments");
    if (n == 0) return 0;
    hipLaunchKernelGGL(k_lower_bound_fwd, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, x, y, n, bound);
    return check_launch("lower_bound_fwd");
}
extern "C" int lldwt_lower_bound_bwd(const float* x, const float* gy, float* gx, int64_t n, float bound, void* stream) {
    LLDWT_REQUIRE(x && gy && gx && n >= 0, "lower_bound_bwd: bad arguments");
    if (n == 0) return 0;
    hipLaunchKernelGGL(k_lower_bound_bwd, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, x, gy, gx, n, bound);
    return check_launch("lower_bound_bwd");
}
static inline void nonneg_consts(float minimum, float* b, float* ped) {
    const double p = 3.814697265625e-06 * 3.814697265625e-06;
    *ped = (float)p;
    *b = (float)sqrt((double)minimum + p);
}
extern "C" int lldwt_nonneg_param_fwd(const float* x, float* y, int64_t n, float minimum, void* stream) {
    LLDWT_REQUIRE(x && y && n >= 0, "nonneg_param_fwd: bad arguments");
    if (n == 0) return 0;
    float b, ped;
    nonneg_consts(minimum, &b, &ped);
    hipLaunchKernelGGL(k_nonneg_fwd, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, x, y, n, b, ped);
    return check_launch("nonneg_param_fwd");
}
extern "C" int lldwt_nonneg_param_bwd(const float* x, const float* gy, float* gx, int64_t n, float minimum, void* stream) {
    LLDWT_REQUIRE(x && gy && gx && n >= 0, "nonneg_param_bwd: bad arguments");
    if (n == 0) return 0;
    float b, ped;
    nonneg_consts(minimum, &b, &ped);
    hipLaunchKernelGGL(k_nonneg_bwd, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, x, gy, gx, n, b);
    return check_launch("nonneg_param_bwd");
}

extern "C" int lldwt_gauss_rate(const float* x, const float* params, const float* noise, float* bits, float* qout,
                                double* bit_sum, int64_t Z, int C, int64_t hw, void* stream) {
    LLDWT_REQUIRE(x && params && Z > 0 && C > 0 && hw > 0, "gauss_rate: bad arguments");
    const int64_t ZC = Z * C;
    // Many short workgroups: their start times spread by themselves, so the loads of some overlap the ~100 vector instructions per
    // coefficient of others (one 16-byte group per lane; a launch of a few long-lived workgroups runs in step -- a load phase, then
    // an arithmetic phase: 16.3 vs 18.2 us at the BASELINE batch).  With bit_sum every workgroup ends in one double atomic on ONE
    // address (thousands of them serialise at the memory side): at most 16 workgroups per CU then, every lane taking k groups with the
    // next one's loads in flight.
    int64_t gy = ZC > 65535 ? 65535 : ZC;
    const int64_t cap = (int64_t)lldwt_num_cus() * (bit_sum ? 16 : 128);
    int64_t gx = cdiv(hw, 1024);
    for (int k = 2; gx * gy > cap && gx > 1; ++k) gx = cdiv(hw, 1024 * (int64_t)k);   // k iterations for every lane alike
    gx = gx < 1 ? 1 : (gx > 1024 ? 1024 : gx);
    const int mode = (noise ? 1 : 0) | (bits ? 2 : 0) | (qout ? 4 : 0);
    const dim3 grid((unsigned)gx, (unsigned)gy);
#define LLDWT_GR_LAUNCH(M) \
    case M: hipLaunchKernelGGL(k_gauss_rate<M>, grid, dim3(256), 0, (hipStream_t)stream, x, params, noise, bits, qout, bit_sum, C, hw, ZC); break;
    switch (mode) {
        LLDWT_GR_LAUNCH(0) LLDWT_GR_LAUNCH(1) LLDWT_GR_LAUNCH(2) LLDWT_GR_LAUNCH(3)
        LLDWT_GR_LAUNCH(4) LLDWT_GR_LAUNCH(5) LLDWT_GR_LAUNCH(6) LLDWT_GR_LAUNCH(7)
    }
#undef LLDWT_GR_LAUNCH
    return check_launch("gauss_rate");
}
extern "C" int lldwt_quantize(const float* x, const float* noise, float* q, int64_t n, void* stream) {
    LLDWT_REQUIRE(x && q && n >= 0, "quantize: bad arguments");
    if (n == 0) return 0;
    hipLaunchKernelGGL(k_quantize, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, x, noise, q, n);
    return check_launch("quantize");
}
extern "C" int lldwt_factorized_rate(const float* x, const float* eb, const float* noise, float* bits, float* qout,
                                     double* bit_sum, int64_t planes, int64_t batch, int C, int64_t hw, void* stream) {
    LLDWT_REQUIRE(x && eb && planes > 0 && batch > 0 && C > 0 && hw > 0, "factorized_rate: bad arguments");
    LLDWT_REQUIRE(planes * batch <= 65535 && C <= 65535, "factorized_rate: grid too large");
    // eval: every workgroup first builds the per-offset table (255 chain evaluations), so give it >= 8192 elements to
    // stream afterwards (4 per lane per iteration)
    int64_t gx = noise ? cdiv(hw, 256) : cdiv(hw, 8192);
    if (gx > 1024) gx = 1024;
    dim3 grid((unsigned)gx, (unsigned)C, (unsigned)(planes * batch));
    {
        const dim3 blk(256);
        if (bits && qout) hipLaunchKernelGGL((k_factorized_rate<true, true>), grid, blk, 0, (hipStream_t)stream, x, eb, noise, bits, qout, bit_sum, (int)batch, C, hw, (const float*)nullptr);
        else if (bits) hipLaunchKernelGGL((k_factorized_rate<true, false>), grid, blk, 0, (hipStream_t)stream, x, eb, noise, bits, qout, bit_sum, (int)batch, C, hw, (const float*)nullptr);
        else if (qout) hipLaunchKernelGGL((k_factorized_rate<false, true>), grid, blk, 0, (hipStream_t)stream, x, eb, noise, bits, qout, bit_sum, (int)batch, C, hw, (const float*)nullptr);
        else hipLaunchKernelGGL((k_factorized_rate<false, false>), grid, blk, 0, (hipStream_t)stream, x, eb, noise, bits, qout, bit_sum, (int)batch, C, hw, (const float*)nullptr);
    }
    return check_launch("factorized_rate");
}
extern "C" int lldwt_factorized_table(const float* eb, float* table, int64_t planes, int C, void* stream) {
    LLDWT_REQUIRE(eb && table && planes > 0 && C > 0 && planes * C <= 65535, "factorized_table: bad arguments");
    hipLaunchKernelGGL(k_factorized_table, dim3((unsigned)(planes * C)), dim3(256), 0, (hipStream_t)stream, eb, table);
    return check_launch("factorized_table");
}
extern "C" int lldwt_factorized_rate_tab(const float* x, const float* eb, const float* table, float* bits, float* qout,
                                         double* bit_sum, int64_t planes, int64_t batch, int C, int64_t hw, void* stream) {
    LLDWT_REQUIRE(x && eb && table && planes > 0 && batch > 0 && C > 0 && hw > 0, "factorized_rate_tab: bad arguments");
    LLDWT_REQUIRE(planes * batch <= 65535 && C <= 65535, "factorized_rate_tab: grid too large");
    // A workgroup pays a prologue (parameters, table row, barrier) before it streams: eight 16-byte groups per lane behind it
    // (sweep at 72 and 864 rows of 64 K coefficients: 1 / 2 / 4 / 8 / 64 groups -> 15.2 / 12.4 / 11.6 / 10.9 / - us and
    // 179 / 125 / - / 124 / 142 us), fewer while that leaves the chip under three workgroups per CU; with bit_sum (one double
    // atomic per workgroup on ONE address) at most 8 per CU.
    const int64_t cus = lldwt_num_cus(), rows = (int64_t)C * planes * batch;
    int k = 8;
    int64_t gx = cdiv(hw, 1024 * (int64_t)k);
    while (k > 1 && gx * rows < 3 * cus) gx = cdiv(hw, 1024 * (int64_t)--k);
    if (bit_sum)
        while (gx * rows > 8 * cus && gx > 1) gx = cdiv(hw, 1024 * (int64_t)++k);
    gx = gx < 1 ? 1 : (gx > 1024 ? 1024 : gx);
    dim3 grid((unsigned)gx, (unsigned)C, (unsigned)(planes * batch));
    {
        const dim3 blk(256);
        if (bits && qout) hipLaunchKernelGGL((k_factorized_rate<true, true>), grid, blk, 0, (hipStream_t)stream, x, eb, (const float*)nullptr, bits, qout, bit_sum, (int)batch, C, hw, table);
        else if (bits) hipLaunchKernelGGL((k_factorized_rate<true, false>), grid, blk, 0, (hipStream_t)stream, x, eb, (const float*)nullptr, bits, qout, bit_sum, (int)batch, C, hw, table);
        else if (qout) hipLaunchKernelGGL((k_factorized_rate<false, true>), grid, blk, 0, (hipStream_t)stream, x, eb, (const float*)nullptr, bits, qout, bit_sum, (int)batch, C, hw, table);
        else hipLaunchKernelGGL((k_factorized_rate<false, false>), grid, blk, 0, (hipStream_t)stream, x, eb, (const float*)nullptr, bits, qout, bit_sum, (int)batch, C, hw, table);
    }
    return check_launch("factorized_rate_tab");
}
extern "C" int lldwt_sq_err_sum(const float* a, const float* b, int64_t n, double* out, void* stream) {
    LLDWT_REQUIRE(a && b && out && n > 0, "sq_err_sum: bad arguments");
    hipLaunchKernelGGL(k_sq_err_sum, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, a, b, n, out);
    return check_launch("sq_err_sum");
}
extern "C" int lldwt_sum(const float* x, int64_t n, double* out, void* stream) {
    LLDWT_REQUIRE(x && out && n > 0, "sum: bad arguments");
    hipLaunchKernelGGL(k_sum, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, x, n, out);
    return check_launch("sum");
}


extern "C" int lldwt_gauss_rate_bwd(const float* x, const float* params, const float* noise, const float* gbits, float* dx,
                                    float* dparams, int64_t Z, int C, int64_t hw, void* stream) {
    LLDWT_REQUIRE(x && params && gbits && dx && dparams && Z > 0 && C > 0 && hw > 0, "gauss_rate_bwd: bad arguments");
    const int64_t n = Z * C * hw;
    hipLaunchKernelGGL(k_gauss_rate_bwd, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, x, params, noise, gbits, dx,
                       dparams, C, hw, n);
    return check_launch("gauss_rate_bwd");
}

// out[i] = alpha * a[i] + beta * b[i]  (b may be null); used for d(mse), d(sum) and the colour-transform backward glue
__global__ void k_axpby(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out, int64_t n,
                        float alpha, float beta) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        out[i] = alpha * a[i] + (b ? beta * b[i] : 0.f);
}
extern "C" int lldwt_axpby(const float* a, const float* b, float* out, int64_t n, float alpha, float beta, void* stream) {
    LLDWT_REQUIRE(a && out && n >= 0, "axpby: bad arguments");
    if (n == 0) return 0;
    hipLaunchKernelGGL(k_axpby, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, a, b, out, n, alpha, beta);
    return check_launch("axpby");
}

// backward of lldwt_ycc_to_rgb (no clamp): grgb (B,3,hw) -> gycc plane-major (3,B,hw); transpose of the BT.709 matrix
__global__ void k_ycc_to_rgb_bwd(const float* __restrict__ grgb, float* __restrict__ gycc, int64_t B, int64_t hw) {
    const int64_t n = B * hw;
    const float cr_r = 2.f - 2.f * KR, cb_b = 2.f - 2.f * KB;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t b = i / hw, p = i - b * hw;
        const float gr = grgb[(b * 3 + 0) * hw + p], gg = grgb[(b * 3 + 1) * hw + p], gb = grgb[(b * 3 + 2) * hw + p];
        // r = y + cr_r*(cr-.5); b = y + cb_b*(cb-.5); g = (y - KR*r - KB*b)/KG
        const float gr_t = gr - gg * KR / KG, gb_t = gb - gg * KB / KG;     // total grads wrt r and b
        gycc[(0 * B + b) * hw + p] = gg / KG + gr_t + gb_t;
        gycc[(1 * B + b) * hw + p] = gb_t * cb_b;
        gycc[(2 * B + b) * hw + p] = gr_t * cr_r;
    }
}
extern "C" int lldwt_ycc_to_rgb_bwd(const float* grgb, float* gycc, int64_t B, int64_t H, int64_t W, void* stream) {
    LLDWT_REQUIRE(grgb && gycc && B > 0 && H > 0 && W > 0, "ycc_to_rgb_bwd: bad arguments");
    hipLaunchKernelGGL(k_ycc_to_rgb_bwd, dim3(ew_grid(B * H * W)), dim3(256), 0, (hipStream_t)stream, grgb, gycc, B, H * W);
    return check_launch("ycc_to_rgb_bwd");
}

extern "C" int lldwt_factorized_rate_bwd(const float* x, const float* eb, const float* noise, const float* gbits, float* dx,
                                         float* deb, int64_t planes, int64_t batch, int C, int64_t hw, void* stream) {
    LLDWT_REQUIRE(x && eb && gbits && dx && deb && planes > 0 && batch > 0 && C > 0 && hw > 0, "factorized_rate_bwd: bad arguments");
    LLDWT_REQUIRE(planes * batch <= 65535 && C <= 65535, "factorized_rate_bwd: grid too large");
    int64_t gx = cdiv(hw, 256 * 8);
    if (gx < 1) gx = 1;
    if (gx > 256) gx = 256;
    dim3 grid((unsigned)gx, (unsigned)C, (unsigned)(planes * batch));
    hipLaunchKernelGGL(k_factorized_rate_bwd, grid, dim3(256), 0, (hipStream_t)stream, x, eb, noise, gbits, dx, deb,
                       (int)batch, C, hw);
    return check_launch("factorized_rate_bwd");
}

// ---- small elementwise pieces used by the differentiable (training) composition of GDN -------------------------------------
// out = scale * a * b
__global__ void k_ew_mul(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out, int64_t n, float scale) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        out[i] = scale * a[i] * b[i];
}
// y = x * rsqrt(nrm)  (inverse: x * sqrt(nrm))   -- graphs/layers/gdn.py:85-90
__global__ void k_gdn_apply(const float* __restrict__ x, const float* __restrict__ nrm, float* __restrict__ y, int64_t n, int inverse) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float r = inverse ? sqrtf(nrm[i]) : 1.0f / sqrtf(nrm[i]);
        y[i] = x[i] * r;
    }
}
__global__ void k_gdn_apply_bwd(const float* __restrict__ x, const float* __restrict__ nrm, const float* __restrict__ g,
                                float* __restrict__ dx, float* __restrict__ dn, int64_t n, int inverse) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float nv = nrm[i], xv = x[i], gv = g[i];
        if (inverse) {
            const float r = sqrtf(nv);
            dx[i] = gv * r;
            dn[i] = gv * xv * 0.5f / r;
        } else {
            const float r = 1.0f / sqrtf(nv);
            dx[i] = gv * r;
            dn[i] = -0.5f * gv * xv * r * r * r;
        }
    }
}
extern "C" int lldwt_ew_mul(const float* a, const float* b, float* out, int64_t n, float scale, void* stream) {
    LLDWT_REQUIRE(a && b && out && n >= 0, "ew_mul: bad arguments");
    if (n == 0) return 0;
    hipLaunchKernelGGL(k_ew_mul, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, a, b, out, n, scale);
    return check_launch("ew_mul");
}
extern "C" int lldwt_gdn_apply(const float* x, const float* nrm, float* y, int64_t n, int inverse, void* stream) {
    LLDWT_REQUIRE(x && nrm && y && n >= 0, "gdn_apply: bad arguments");
    if (n == 0) return 0;
    hipLaunchKernelGGL(k_gdn_apply, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, x, nrm, y, n, inverse);
    return check_launch("gdn_apply");
}
extern "C" int lldwt_gdn_apply_bwd(const float* x, const float* nrm, const float* g, float* dx, float* dn, int64_t n,
                                   int inverse, void* stream) {
    LLDWT_REQUIRE(x && nrm && g && dx && dn && n >= 0, "gdn_apply_bwd: bad arguments");
    if (n == 0) return 0;
    hipLaunchKernelGGL(k_gdn_apply_bwd, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, x, nrm, g, dx, dn, n, inverse);
    return check_launch("gdn_apply_bwd");
}
