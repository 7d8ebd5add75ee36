// lifting_f16.hip -- ONE launch per lifting step: the whole P/U block of the learned lifting (reference:
// graphs/layers/wavelet_forward_v2.py:58-74, wavelet_inverse_v2.py:76-90, graphs/layers/P_block_v2.py:40-55) for a
// 16 x 32 output tile, every 16-channel intermediate kept in LDS, the 16 -> 16 5x5 convolutions on the fp16 matrix cores
// with split-fp16 operands (hi + lo fp16 per fp32 value, power-of-two scaled, three MFMA products per MAC, fp32
// accumulate: fp32-level accuracy at 16/3 x the fp32 MFMA rate -- see conv_f16x3.hip for the arithmetic).
//
//   dst_out = dst_in + sign * (skip + rw * net),   skip = 3-tap filter of src along the lifting direction,
//   net = conv4(conv3(tanh(conv2(tanh(r)))) + r),  r = conv1(skip);   every conv zero-pads at the IMAGE border.
//
// Replaces the three launches k_lift_a_mfma / k_lift_b_mfma / k_lift_c of lifting.hip (fp32 MFMA, intermediates through
// HBM: ~11 GB per 8x3x512x512 forward) on the eval path; those kernels remain for training (they save the
// intermediates), for other channel counts / kernel sizes and as the exact-fp32 fallback (LLDWT_LIFT_MODE=f32).
//
// Tile anatomy (halo recompute instead of HBM round trips): output 16x32; t3 on 20x36; t2 on 24x40; t1 on 28x44; skip on
// 32x48.  A "T-image" holds a 16-channel tensor over a region as four arrays [pixel][8 x fp16] (hi ch 0-7, hi ch 8-15,
// lo ch 0-7, lo ch 8-15): a lane's MFMA B fragment (8 consecutive channels of one pixel) is one conflict-free
// ds_read_b128, and the D fragment of v_mfma_f32_16x16x32_f16 (4 consecutive channels of one pixel per lane) is one
// ds_write_b64 per part.  MFMA roles: A = weights (16 output channels x 32 = 2 taps x 16 channels, fragments held in
// registers for the whole phase), B = activations (k x 16 pixels; pixel = flattened index in the region, so tiles may
// wrap rows -- addressing is per lane anyway).  conv1 (1 -> 16) gathers its 25 taps from the fp32 skip patch straight
// into a B fragment; conv4 (16 -> 1) is computed as D[dx][pixel] = sum over (dy, channel) and finished by a 5-term
// shifted sum, 9 MFMAs per 16 pixels instead of 39.
// LDS: skip 6 KB + T1 77 KB + T2 60 KB (+ T3 aliasing T1, D aliasing T2) = 143 KB: one 512-thread workgroup per CU.
#include "lifting_f16.h"
#include "split_f16.h"

namespace lldwt {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int TH = 16, TW = 32, NTH = 512, NWAVE = 8;
constexpr int SH = TH + 16, SW = TW + 16, NS = SH * SW;              // 32 x 48 skip patch (fp32)
constexpr int R1W = TW + 12, N1 = (TH + 12) * R1W;                    // 28 x 44 = 1232
constexpr int R2W = TW + 8, N2 = (TH + 8) * R2W;                      // 24 x 40 =  960
constexpr int R3W = TW + 4, N3 = (TH + 4) * R3W;                      // 20 x 36 =  720
constexpr int RDW = TW + 4, ND = TH * RDW;                            // 16 x 36 =  576
static_assert(N1 % 16 == 0 && N2 % 16 == 0 && N3 % 16 == 0 && ND % 16 == 0, "regions are whole 16-pixel tiles");
constexpr int NT1 = N1 / 16, NT2 = N2 / 16, NT3 = N3 / 16, NTD = ND / 16;   // 77, 60, 45, 36
constexpr int LDS_S = 0;
constexpr int LDS_RED = NS * 4;                                       // 64 floats of scratch
constexpr int LDS_T1 = LDS_RED + 256;
constexpr int LDS_T2 = LDS_T1 + 4 * N1 * 16;
constexpr int LDS_TOTAL = LDS_T2 + 4 * N2 * 16;                       // 146,688 B
constexpr int LDS_T3 = LDS_T1, LDS_D = LDS_T2;
static_assert(4 * N3 * 16 <= 4 * N1 * 16 && ND * 8 * 4 <= 4 * N2 * 16, "aliases fit");
constexpr float ACT_SCALE = 16384.f;                                  // tanh outputs: |t| <= 1 -> |t * 2^14| < fp16 max

__device__ __forceinline__ float pow2_scale(float amax) {             // s = 2^k with amax * s in [2^14, 2^15)
    if (!(amax > 0.f) || !(amax < 3.0e38f)) return 1.f;
    int e;
    (void)frexpf(amax, &e);
    int k = 15 - e;
    k = k > 120 ? 120 : (k < -120 ? -120 : k);
    return ldexpf(1.f, k);
}

__device__ __forceinline__ void split4(const float (&v)[4], half4& hi, half4& lo) { split4v(v, hi, lo); }

// effective tap t = dy*5+dx of an orientation -> index into the PyTorch (kh,kw) weight
__device__ __forceinline__ int srctap(int t, int orient) { return orient == 0 ? t : (t % LF_K) * LF_K + t / LF_K; }

__global__ void k_lift_f16_pack(const float* __restrict__ w1, const float* __restrict__ w2, const float* __restrict__ w3,
                                const float* __restrict__ w4, float* __restrict__ packed, int64_t plane_stride, int f16_off) {
    const int orient = blockIdx.x, plane = blockIdx.y, tid = threadIdx.x;
    w1 += (int64_t)plane * LF_C * LF_KK;
    w2 += (int64_t)plane * LF_C * LF_C * LF_KK;
    w3 += (int64_t)plane * LF_C * LF_C * LF_KK;
    w4 += (int64_t)plane * LF_C * LF_KK;
    __shared__ float red[4][4];
    float m[4] = {0.f, 0.f, 0.f, 0.f};
    for (int i = tid; i < LF_C * LF_KK; i += 256) m[0] = fmaxf(m[0], fabsf(w1[i]));
    for (int i = tid; i < LF_C * LF_C * LF_KK; i += 256) {
        m[1] = fmaxf(m[1], fabsf(w2[i]));
        m[2] = fmaxf(m[2], fabsf(w3[i]));
    }
    for (int i = tid; i < LF_C * LF_KK; i += 256) m[3] = fmaxf(m[3], fabsf(w4[i]));
#pragma unroll
    for (int q = 0; q < 4; ++q) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) m[q] = fmaxf(m[q], __shfl_xor(m[q], o, 64));
        if ((tid & 63) == 0) red[q][tid >> 6] = m[q];
    }
    __syncthreads();
    float sw[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) sw[q] = pow2_scale(fmaxf(fmaxf(red[q][0], red[q][1]), fmaxf(red[q][2], red[q][3])));
    float* dst = packed + (int64_t)plane * plane_stride + f16_off + (int64_t)orient * LF_ORIENT_FLOATS;
    _Float16* hp = reinterpret_cast<_Float16*>(dst);
    // ---- the composed kernels of this orientation (effective taps): P_block_v2.py:50-55 has no nonlinearity between
    // conv3 and conv4, so away from the image border conv4(conv3(t2) + b3 + r) + b4 =
    //   (w4 o w3) * t2  +  (w4 o w1) * skip  +  b4 + sum_oc (b3 + b1)[oc] * sum_p w4[oc][p]
    // wc[ic][u][v] = sum_oc sum_{p + q = (u, v)} w4[oc][p] w3[oc][ic][q]  (9x9);  wr[u][v] the same with w1 (1 -> 16)
    __shared__ float wc[LF_C * 81];
    __shared__ float wr[81];
    __shared__ float redc[4];
    for (int i = tid; i < LF_C * 81 + 81; i += 256) {
        const bool is_r = i >= LF_C * 81;
        const int ic = is_r ? 0 : i / 81, s = is_r ? i - LF_C * 81 : i % 81;
        const int u = s / 9, vv = s % 9;
        double acc = 0.0;
        for (int oc = 0; oc < LF_C; ++oc)
            for (int py = 0; py < LF_K; ++py) {
                const int qy = u - py;
                if (qy < 0 || qy >= LF_K) continue;
                for (int px = 0; px < LF_K; ++px) {
                    const int qx = vv - px;
                    if (qx < 0 || qx >= LF_K) continue;
                    const float a4 = w4[oc * LF_KK + srctap(py * LF_K + px, orient)];
                    const float b = is_r ? w1[oc * LF_KK + srctap(qy * LF_K + qx, orient)]
                                         : w3[(oc * LF_C + ic) * LF_KK + srctap(qy * LF_K + qx, orient)];
                    acc += (double)a4 * (double)b;
                }
            }
        if (is_r) wr[s] = (float)acc; else wc[i] = (float)acc;
    }
    __syncthreads();
    float mc = 0.f;
    for (int i = tid; i < LF_C * 81; i += 256) mc = fmaxf(mc, fabsf(wc[i]));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mc = fmaxf(mc, __shfl_xor(mc, o, 64));
    if ((tid & 63) == 0) redc[tid >> 6] = mc;
    __syncthreads();
    const float swc = pow2_scale(fmaxf(fmaxf(redc[0], redc[1]), fmaxf(redc[2], redc[3])));
    for (int i = tid; i < LF_H_END / 2; i += 256) {          // one (hi, lo) pair per iteration
        int rem = i;
        const int j = rem % 8; rem /= 8;
        const int lane = rem % 64; rem /= 64;
        const int step = rem;                                 // global k-step index over conv1 | conv2 | conv3 | conv4
        const int row = lane & 15, kg = lane >> 4;
        float v = 0.f;
        if (step == 0) {                                      // conv1: k = tap
            const int t = 8 * kg + j;
            if (t < LF_KK) v = w1[row * LF_KK + srctap(t, orient)] * sw[0];
        } else if (step < 1 + 2 * LF_KS) {                    // conv2 / conv3: k = (tap pair, channel)
            const int which = (step - 1) / LF_KS, ks = (step - 1) % LF_KS;
            const int t = 2 * ks + (kg >> 1), ic = 8 * (kg & 1) + j;
            const float* w = which == 0 ? w2 : w3;
            if (t < LF_KK) v = w[(row * LF_C + ic) * LF_KK + srctap(t, orient)] * sw[1 + which];
        } else if (step < 1 + 2 * LF_KS + LF_KS4) {           // conv4: rows = dx, k = (dy, channel)
            const int ks = step - 1 - 2 * LF_KS;
            const int k = 32 * ks + 8 * kg + j, dy = k / LF_C, ic = k % LF_C;
            if (dy < LF_K && row < LF_K) v = w4[ic * LF_KK + srctap(dy * LF_K + row, orient)] * sw[3];
        } else {                                              // conv4 o conv3: rows = dx (0..8), k = (dy 0..8, channel)
            const int ks = step - 1 - 2 * LF_KS - LF_KS4;
            const int k = 32 * ks + 8 * kg + j, dy = k / LF_C, ic = k % LF_C;
            if (dy < 9 && row < 9) v = wc[ic * 81 + dy * 9 + row] * swc;
        }
        const _Float16 hi = (_Float16)v;
        const _Float16 lo = (_Float16)(v - (float)hi);
        hp[(step * 2 + 0) * 512 + lane * 8 + j] = hi;
        hp[(step * 2 + 1) * 512 + lane * 8 + j] = lo;
    }
    float* tail = dst + LF_H_END / 2;
    if (tid < 4) tail[tid] = sw[tid];
    if (tid == 4) tail[4] = swc;
    if (tid < LF_C) {
        float s4 = 0.f;
        for (int t = 0; t < LF_KK; ++t) s4 += w4[tid * LF_KK + t];
        tail[16 + tid] = s4;
    }
    if (tid < 81) tail[32 + tid] = wr[tid];
}

struct LfArgs {
    LiftF16Views v;
    const float* taps;
    const float* packed;
    int64_t pstride;
    int orient_fp32;      // float offset of this orientation's fp32 section (biases)
    int b1, b2, b3, b4;   // float offsets of the biases inside an fp32 orientation section
    int f16;              // float offset of this orientation's f16 section
    int batch, h, w, vertical;
    float sign, rw;
    int dbg;              // diagnostics only (LLDWT_LF_DBG): bit i set = skip the tile loop of phase P(i+1); results are then wrong
};

// 13 k-steps of one 16-pixel tile: B fragments from a T-image (input region width WIN, NIN pixels), A fragments in registers
template <int WIN, int NIN>
__device__ __forceinline__ floatx4 conv16_tile(const uint8_t* __restrict__ img, int basein, bool hi_tap,
                                               const half8 (&ah)[LF_KS], const half8 (&al)[LF_KS]) {
    floatx4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < LF_KS; ++ks) {
        constexpr int dummy = 0;
        (void)dummy;
        const int ta = 2 * ks, tb = (2 * ks + 1) < LF_KK ? 2 * ks + 1 : LF_KK - 1;      // tap 25 does not exist: weight 0
        const int offa = ((ta / LF_K) * WIN + ta % LF_K) * 16, offb = ((tb / LF_K) * WIN + tb % LF_K) * 16;
        const int off = basein + (hi_tap ? offb : offa);
        const half8 bh = *reinterpret_cast<const half8*>(img + off);
        const half8 bl = *reinterpret_cast<const half8*>(img + 2 * NIN * 16 + off);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[ks], bh, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[ks], bl, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[ks], bh, acc, 0, 0, 0);
    }
    return acc;
}

// two tiles at once: the A fragments are shared, the two MFMA chains are independent.  The B fragments of k-step ks+1 are
// read from LDS into a second register set BEFORE the 6 MFMAs of k-step ks (the sched_group_barrier sequence pins that
// order and leaves room for 2 vector instructions of a neighbouring epilogue after every MFMA), so the matrix pipe never
// waits for an LDS round trip.
template <int WIN, int NIN>
__device__ __forceinline__ void conv16_tile2(const uint8_t* __restrict__ img, int base0, int base1, bool hi_tap,
                                             const half8 (&ah)[LF_KS], const half8 (&al)[LF_KS], floatx4& acc0, floatx4& acc1) {
    acc0 = floatx4{0.f, 0.f, 0.f, 0.f};
    acc1 = floatx4{0.f, 0.f, 0.f, 0.f};
    half8 bh0[2], bl0[2], bh1[2], bl1[2];
    auto load = [&](int ks, int set) {
        const int ta = 2 * ks, tb = (2 * ks + 1) < LF_KK ? 2 * ks + 1 : LF_KK - 1;
        const int offa = ((ta / LF_K) * WIN + ta % LF_K) * 16, offb = ((tb / LF_K) * WIN + tb % LF_K) * 16;
        const int off = hi_tap ? offb : offa;
        bh0[set] = *reinterpret_cast<const half8*>(img + base0 + off);
        bl0[set] = *reinterpret_cast<const half8*>(img + 2 * NIN * 16 + base0 + off);
        bh1[set] = *reinterpret_cast<const half8*>(img + base1 + off);
        bl1[set] = *reinterpret_cast<const half8*>(img + 2 * NIN * 16 + base1 + off);
    };
    load(0, 0);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int ks = 0; ks < LF_KS; ++ks) {
        const int c = ks & 1;
        if (ks + 1 < LF_KS) load(ks + 1, c ^ 1);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[ks], bh0[c], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[ks], bh1[c], acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[ks], bl0[c], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[ks], bl1[c], acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[ks], bh0[c], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[ks], bh1[c], acc1, 0, 0, 0);
        // this k-step is a scheduling region of its own (fenced): its 4 LDS reads can only be the NEXT step's, one after
        // each of the first four MFMAs
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            if (i < 4) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

// scheduling hint for a region that holds NM MFMAs next to LDS reads and vector work: one MFMA, one LDS read, a few VALU
template <int NM>
__device__ __forceinline__ void interleave_hint() {
#pragma unroll
    for (int i = 0; i < NM; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
        __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
    }
}

// conv1 of one 16-pixel tile: gather this lane's 8 taps from the fp32 skip patch, split, 3 MFMAs
__device__ __forceinline__ floatx4 conv1_tile(const float* __restrict__ S, int sbase, const int (&soff)[8], float s_skip,
                                              const half8& a1h, const half8& a1l) {
    half8 bh, bl;
    float g[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) g[j] = S[sbase + soff[j]] * s_skip;
    split8v(g, bh, bl);
    floatx4 acc = {0.f, 0.f, 0.f, 0.f};
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1l, bh, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1h, bl, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1h, bh, acc, 0, 0, 0);
    return acc;
}

// store 4 consecutive channels (oc0 .. oc0+3) of pixel p into a T-image of N pixels
template <int N>
__device__ __forceinline__ void timg_store(uint8_t* __restrict__ img, int p, int oc0, const float (&v)[4]) {
    half4 hi, lo;
    split4(v, hi, lo);
    uint8_t* d = img + (oc0 >> 3) * (N * 16) + p * 16 + (oc0 & 7) * 2;
    *reinterpret_cast<half4*>(d) = hi;
    *reinterpret_cast<half4*>(d + 2 * N * 16) = lo;
}

__global__ __launch_bounds__(NTH) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_lift_fused_f16(LfArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    float* S = reinterpret_cast<float*>(lds + LDS_S);
    float* RED = reinterpret_cast<float*>(lds + LDS_RED);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int kg = lane >> 4, pl = lane & 15, oc0 = 4 * kg;
    const bool hi_tap = kg >= 2;
    const int halfsel = kg & 1;
    const int64_t z = blockIdx.z;
    const int plane = (int)(z / a.batch);
    const int y0 = blockIdx.y * TH, x0 = blockIdx.x * TW;
    const int h = a.h, w = a.w;
    const float* pk = a.packed + (int64_t)plane * a.pstride;
    const float* bias = pk + a.orient_fp32;
    const _Float16* frag = reinterpret_cast<const _Float16*>(pk + a.f16);
    const float* scales = pk + a.f16 + LF_H_END / 2;
    const float sw1 = scales[0], sw2 = scales[1], sw3 = scales[2], sw4 = scales[3];

    // ---------------- P0: skip patch (32 x 48) and its |max|
    {
        const float t0 = a.taps[plane * 3 + 0], t1 = a.taps[plane * 3 + 1], t2 = a.taps[plane * 3 + 2];
        const float* sp = a.v.src + z * a.v.src_sz;
        float amax = 0.f;
#pragma unroll
        for (int k = 0; k < NS / NTH; ++k) {
            const int i = tid + k * NTH;
            const int sy = i / SW, sx = i - sy * SW;
            const int gy = y0 - 8 + sy, gx = x0 - 8 + sx;
            float v = 0.f;
            if (gy >= 0 && gy < h && gx >= 0 && gx < w) {
                const float c = sp[(int64_t)gy * a.v.src_sy + (int64_t)gx * a.v.src_sx];
                float m = 0.f, p = 0.f;
                if (a.vertical) {
                    if (gy > 0) m = sp[(int64_t)(gy - 1) * a.v.src_sy + (int64_t)gx * a.v.src_sx];
                    if (gy + 1 < h) p = sp[(int64_t)(gy + 1) * a.v.src_sy + (int64_t)gx * a.v.src_sx];
                } else {
                    if (gx > 0) m = sp[(int64_t)gy * a.v.src_sy + (int64_t)(gx - 1) * a.v.src_sx];
                    if (gx + 1 < w) p = sp[(int64_t)gy * a.v.src_sy + (int64_t)(gx + 1) * a.v.src_sx];
                }
                v = t0 * m + t1 * c + t2 * p;
            }
            S[i] = v;
            amax = fmaxf(amax, fabsf(v));
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o, 64));
        if (lane == 0) RED[wave] = amax;
    }
    __syncthreads();
    float s_skip;
    {
        float m = RED[0];
#pragma unroll
        for (int i = 1; i < NWAVE; ++i) m = fmaxf(m, RED[i]);
        s_skip = pow2_scale(m);
    }
    const float inv1 = (1.f / s_skip) * (1.f / sw1);

    // conv1 operands: this lane's 8 taps (k = 8*kg + j) as offsets into the skip patch
    int soff[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int t = 8 * kg + j;
        soff[j] = t < LF_KK ? (t / LF_K) * SW + t % LF_K : 0;
    }
    const half8 a1h = *reinterpret_cast<const half8*>(frag + LF_H_C1 + lane * 8);
    const half8 a1l = *reinterpret_cast<const half8*>(frag + LF_H_C1 + 512 + lane * 8);
    float b1v[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) b1v[q] = bias[a.b1 + oc0 + q];

    // ---------------- P1: t1 = tanh(conv1(skip) + b1) on 28 x 44
    for (int tile = wave; tile < ((a.dbg & 1) ? 0 : NT1); tile += NWAVE) {
        const int p = tile * 16 + pl;
        const int r = p / R1W, c = p - r * R1W;
        const floatx4 acc = conv1_tile(S, r * SW + c, soff, s_skip, a1h, a1l);
        const int gy = y0 - 6 + r, gx = x0 - 6 + c;
        const bool in = gy >= 0 && gy < h && gx >= 0 && gx < w;
        float v[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = in ? fast_tanh(acc[q] * inv1 + b1v[q]) * ACT_SCALE : 0.f;
        timg_store<N1>(lds + LDS_T1, p, oc0, v);
    }
    __syncthreads();

    // ---------------- P2: t2 = tanh(conv2(t1) + b2) on 24 x 40
    {
        half8 ah[LF_KS], al[LF_KS];
#pragma unroll
        for (int ks = 0; ks < LF_KS; ++ks) {
            ah[ks] = *reinterpret_cast<const half8*>(frag + LF_H_C2 + (ks * 2 + 0) * 512 + lane * 8);
            al[ks] = *reinterpret_cast<const half8*>(frag + LF_H_C2 + (ks * 2 + 1) * 512 + lane * 8);
        }
        float bv[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) bv[q] = bias[a.b2 + oc0 + q];
        const float inv2 = (1.f / ACT_SCALE) * (1.f / sw2);
        // software pipeline over PAIRS of tiles (wave + 16 it, wave + 16 it + 8): the MFMA chains of pair it+1 are issued
        // next to the tanh / split / store work of pair it, so the vector ALU and the matrix pipe overlap inside one wave
        constexpr int NIT2 = (NT2 + 2 * NWAVE - 1) / (2 * NWAVE);      // 4
        auto tile_base = [&](int tile, int& p_out) {
            const int tcl = tile < NT2 ? tile : NT2 - 1;               // past the end: the last tile again (same bytes stored twice)
            const int p = tcl * 16 + pl;
            const int r = p / R2W, c = p - r * R2W;
            p_out = p;
            return (r * R1W + c) * 16 + halfsel * (N1 * 16);
        };
        auto epilogue2 = [&](const floatx4& acc, int p) {              // branch-free: one basic block per pipeline stage
            const int r = p / R2W, c = p - r * R2W;
            const int gy = y0 - 4 + r, gx = x0 - 4 + c;
            const bool in = gy >= 0 && gy < h && gx >= 0 && gx < w;
            float v[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] = in ? fast_tanh(acc[q] * inv2 + bv[q]) * ACT_SCALE : 0.f;
            timg_store<N2>(lds + LDS_T2, p, oc0, v);
        };
        if (!(a.dbg & 2)) {
            floatx4 c0, c1;
            int p0, p1;
            {
                const int b0 = tile_base(wave, p0), b1 = tile_base(wave + NWAVE, p1);
                conv16_tile2<R1W, N1>(lds + LDS_T1, b0, b1, hi_tap, ah, al, c0, c1);
            }
#pragma unroll
            for (int it = 0; it < NIT2; ++it) {
                floatx4 n0 = c0, n1 = c1;
                int q0 = p0, q1 = p1;
                if (it + 1 < NIT2) {
                    const int b0 = tile_base(wave + 16 * (it + 1), q0), b1 = tile_base(wave + 16 * (it + 1) + NWAVE, q1);
                    conv16_tile2<R1W, N1>(lds + LDS_T1, b0, b1, hi_tap, ah, al, n0, n1);
                }
                epilogue2(c0, p0);
                epilogue2(c1, p1);
                __builtin_amdgcn_sched_barrier(0);
                c0 = n0; c1 = n1; p0 = q0; p1 = q1;
            }
        }
    }
    __syncthreads();

    // ---------------- interior tiles: conv4(conv3(t2) + b3 + r) + b4 through the COMPOSED 9x9 kernels (packed by
    // k_lift_f16_pack): exact algebra wherever the t3 region (20 x 36 around the tile) lies inside the image, because
    // the only thing between conv3 and conv4 is the zero padding at the image border.  5 k-steps x 3 MFMAs per 16 pixels on
    // 16 x 40 pixels instead of 13 x 3 on 20 x 36 plus conv4, no t3 image, no second pass over conv1.
    const bool interior = y0 >= 2 && y0 + TH + 2 <= h && x0 >= 2 && x0 + TW + 2 <= w && !(a.dbg & 16);
    if (interior) {
        constexpr int NPC = TH * R2W, NTC = NPC / 16;           // 640 pixels (16 rows x 40 T2 columns), 40 tiles
        constexpr int DP = 12;                                  // floats per pixel in the D image (9 used)
        static_assert(NPC % 16 == 0 && NPC * DP * 4 <= 4 * N1 * 16, "D image fits the T1 region");
        const float* tail = pk + a.f16 + LF_H_END / 2;
        const float swc = tail[4];
        float* D = reinterpret_cast<float*>(lds + LDS_T1);      // T1 is dead after P2
        {
            half8 ah[LF_KSC], al[LF_KSC];
#pragma unroll
            for (int ks = 0; ks < LF_KSC; ++ks) {
                ah[ks] = *reinterpret_cast<const half8*>(frag + LF_H_CC + (ks * 2 + 0) * 512 + lane * 8);
                al[ks] = *reinterpret_cast<const half8*>(frag + LF_H_CC + (ks * 2 + 1) * 512 + lane * 8);
            }
            const uint8_t* img = lds + LDS_T2;
            for (int tile = wave; tile < NTC; tile += NWAVE) {
                const int p = tile * 16 + pl;
                const int r = p / R2W, c = p - r * R2W;         // D(r, c) <-> output row r, T2 column c; tap dy -> T2 row r + dy
                const int basein = (r * R2W + c) * 16 + halfsel * (N2 * 16);
                floatx4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < LF_KSC; ++ks) {
                    const int dya = 2 * ks, dyb = 2 * ks + 1 < 9 ? 2 * ks + 1 : 8;            // dy 9 does not exist: weight 0
                    const int off = basein + (hi_tap ? dyb : dya) * (R2W * 16);
                    const half8 bh = *reinterpret_cast<const half8*>(img + off);
                    const half8 bl = *reinterpret_cast<const half8*>(img + 2 * N2 * 16 + off);
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[ks], bh, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[ks], bl, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[ks], bh, acc, 0, 0, 0);
                }
                // rows = dx: lanes kg 0 hold dx 0-3, kg 1 dx 4-7, kg 2 dx 8 (register 0)
                if (kg < 2) *reinterpret_cast<floatx4*>(D + p * DP + 4 * kg) = acc;
                if (kg == 2) D[p * DP + 8] = acc[0];
            }
        }
        __syncthreads();
        {
            const int oy = tid / TW, ox = tid - oy * TW;
            const int gy = y0 + oy, gx = x0 + ox;                // inside the image by the interior condition
            const float invc = (1.f / ACT_SCALE) * (1.f / swc);
            float net = 0.f;
#pragma unroll
            for (int dx = 0; dx < 9; ++dx) net += D[(oy * R2W + ox + dx) * DP + dx];
            net *= invc;
            float cst = bias[a.b4];
#pragma unroll
            for (int oc = 0; oc < LF_C; ++oc) cst += (bias[a.b3 + oc] + bias[a.b1 + oc]) * tail[16 + oc];
            float rs = 0.f;                                       // (w4 o w1) * skip: 81 taps on the skip patch
#pragma unroll
            for (int u = 0; u < 9; ++u)
#pragma unroll
                for (int v = 0; v < 9; ++v) rs += tail[32 + u * 9 + v] * S[(oy + 4 + u) * SW + ox + 4 + v];
            net += rs + cst;
            const float skip = S[(oy + 8) * SW + ox + 8];
            const float din = a.v.din[z * a.v.din_sz + (int64_t)gy * a.v.din_sy + (int64_t)gx * a.v.din_sx];
            a.v.dout[z * a.v.dout_sz + (int64_t)gy * a.v.dout_sy + (int64_t)gx * a.v.dout_sx] = din + a.sign * (skip + a.rw * net);
        }
        return;
    }

    // ---------------- P3: t3 = conv3(t2) + b3 + r,  r = conv1(skip) + b1 (recomputed), on 20 x 36; dynamic scale
    constexpr int IT3 = (NT3 + NWAVE - 1) / NWAVE;      // 6
    float t3v[IT3][4];
    float s_t3;
    {
        half8 ah[LF_KS], al[LF_KS];
#pragma unroll
        for (int ks = 0; ks < LF_KS; ++ks) {
            ah[ks] = *reinterpret_cast<const half8*>(frag + LF_H_C3 + (ks * 2 + 0) * 512 + lane * 8);
            al[ks] = *reinterpret_cast<const half8*>(frag + LF_H_C3 + (ks * 2 + 1) * 512 + lane * 8);
        }
        float bv[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) bv[q] = bias[a.b3 + oc0 + q] + b1v[q];
        const float inv3 = (1.f / ACT_SCALE) * (1.f / sw3);
        float amax = 0.f;
#pragma unroll
        for (int it = 0; it < IT3; ++it) {
            const int tile = wave + it * NWAVE;
#pragma unroll
            for (int q = 0; q < 4; ++q) t3v[it][q] = 0.f;
            if (tile < ((a.dbg & 4) ? 0 : NT3)) {
                const int p = tile * 16 + pl;
                const int r = p / R3W, c = p - r * R3W;
                const floatx4 acc = conv16_tile<R2W, N2>(lds + LDS_T2, (r * R2W + c) * 16 + halfsel * (N2 * 16), hi_tap, ah, al);
                const floatx4 accr = conv1_tile(S, (r + 4) * SW + c + 4, soff, s_skip, a1h, a1l);
                const int gy = y0 - 2 + r, gx = x0 - 2 + c;
                const bool in = gy >= 0 && gy < h && gx >= 0 && gx < w;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float v = in ? acc[q] * inv3 + accr[q] * inv1 + bv[q] : 0.f;
                    t3v[it][q] = v;
                    amax = fmaxf(amax, fabsf(v));
                }
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o, 64));
        if (lane == 0) RED[8 + wave] = amax;
    }
    __syncthreads();            // also: every wave is done reading T1 (P2) -- T3 may now overwrite it
    {
        float m = RED[8];
#pragma unroll
        for (int i = 1; i < NWAVE; ++i) m = fmaxf(m, RED[8 + i]);
        s_t3 = pow2_scale(m);
#pragma unroll
        for (int it = 0; it < IT3; ++it) {
            const int tile = wave + it * NWAVE;
            if (tile < NT3) {
                float v[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) v[q] = t3v[it][q] * s_t3;
                timg_store<N3>(lds + LDS_T3, tile * 16 + pl, oc0, v);
            }
        }
    }
    __syncthreads();            // T3 complete; every wave is done reading T2 -- D may now overwrite it

    // ---------------- P4: D[dx][pixel] = sum over (dy, channel) of t3 * w4 on 16 x 36
    {
        half8 ah[LF_KS4], al[LF_KS4];
#pragma unroll
        for (int ks = 0; ks < LF_KS4; ++ks) {
            ah[ks] = *reinterpret_cast<const half8*>(frag + LF_H_C4 + (ks * 2 + 0) * 512 + lane * 8);
            al[ks] = *reinterpret_cast<const half8*>(frag + LF_H_C4 + (ks * 2 + 1) * 512 + lane * 8);
        }
        float* D = reinterpret_cast<float*>(lds + LDS_D);
        const uint8_t* img = lds + LDS_T3;
        for (int tile = wave; tile < ((a.dbg & 8) ? 0 : NTD); tile += NWAVE) {
            const int p = tile * 16 + pl;
            const int r = p / RDW, c = p - r * RDW;
            const int basein = (r * R3W + c) * 16 + halfsel * (N3 * 16);
            floatx4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < LF_KS4; ++ks) {
                const int dya = 2 * ks < LF_K ? 2 * ks : LF_K - 1, dyb = 2 * ks + 1 < LF_K ? 2 * ks + 1 : LF_K - 1;   // dy >= 5: weight 0
                const int off = basein + (hi_tap ? dyb : dya) * (R3W * 16);
                const half8 bh = *reinterpret_cast<const half8*>(img + off);
                const half8 bl = *reinterpret_cast<const half8*>(img + 2 * N3 * 16 + off);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[ks], bh, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[ks], bl, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[ks], bh, acc, 0, 0, 0);
            }
            // D rows = dx: lanes kg == 0 hold dx 0..3, lanes kg == 1 hold dx 4 in register 0
            if (kg == 0) *reinterpret_cast<floatx4*>(D + p * 8) = acc;
            if (kg == 1) D[p * 8 + 4] = acc[0];
        }
    }
    __syncthreads();

    // ---------------- P5: net = b4 + sum_dx D[dx][x + dx];  dst_out = dst_in + sign * (skip + rw * net)
    {
        const float* D = reinterpret_cast<const float*>(lds + LDS_D);
        const int oy = tid / TW, ox = tid - oy * TW;
        const int gy = y0 + oy, gx = x0 + ox;
        if (gy < h && gx < w) {
            const float inv4 = (1.f / s_t3) * (1.f / sw4);
            float net = 0.f;
#pragma unroll
            for (int dx = 0; dx < LF_K; ++dx) net += D[(oy * RDW + ox + dx) * 8 + dx];
            net = net * inv4 + bias[a.b4];
            const float skip = S[(oy + 8) * SW + ox + 8];
            const float din = a.v.din[z * a.v.din_sz + (int64_t)gy * a.v.din_sy + (int64_t)gx * a.v.din_sx];
            a.v.dout[z * a.v.dout_sz + (int64_t)gy * a.v.dout_sy + (int64_t)gx * a.v.dout_sx] = din + a.sign * (skip + a.rw * net);
        }
    }
}

}  // namespace

int lift_f16_pack(const float* w1, const float* w2, const float* w3, const float* w4, float* packed, int64_t plane_stride,
                  int f16_off, int planes, hipStream_t st) {
    hipLaunchKernelGGL(k_lift_f16_pack, dim3(2, (unsigned)planes), dim3(256), 0, st, w1, w2, w3, w4, packed, plane_stride, f16_off);
    return check_launch("lift_f16_pack");
}

int lift_f16_step(const LiftF16Views& v, int64_t Z, int64_t batch, int64_t h, int64_t w, const float* taps,
                  const float* packed, int64_t pstride, int fp32_orient_floats, int f16_off, int vertical, float sign, float rw,
                  hipStream_t st) {
    static bool attr = false;
    if (!attr) {
        if (hipFuncSetAttribute((const void*)k_lift_fused_f16, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_TOTAL) != hipSuccess) {
            set_error("lift_f16_step: cannot reserve %d bytes of LDS", LDS_TOTAL);
            return LLDWT_EHIP;
        }
        attr = true;
    }
    LfArgs a;
    a.v = v;
    a.taps = taps;
    a.packed = packed;
    a.pstride = pstride;
    const int orient = vertical ? 0 : 1;
    a.orient_fp32 = orient * fp32_orient_floats;
    // bias offsets inside an fp32 orientation section (mirror of pack_off in lifting.hip for C = 16, K = 5)
    auto pad16 = [](int n) { return (n + 15) & ~15; };
    const int w1o = 0, b1o = w1o + pad16(LF_KK * LF_C), w2o = b1o + pad16(LF_C), b2o = w2o + pad16(LF_C * LF_KK * LF_C);
    const int w3o = b2o + pad16(LF_C), b3o = w3o + pad16(LF_C * LF_KK * LF_C), w4o = b3o + pad16(LF_C), b4o = w4o + pad16(LF_C * LF_KK);
    a.b1 = b1o; a.b2 = b2o; a.b3 = b3o; a.b4 = b4o;
    a.f16 = f16_off + orient * LF_ORIENT_FLOATS;
    a.batch = (int)batch; a.h = (int)h; a.w = (int)w; a.vertical = vertical;
    a.sign = sign; a.rw = rw;
    const char* dbg = getenv("LLDWT_LF_DBG");
    a.dbg = dbg ? atoi(dbg) : 0;
    dim3 grid((unsigned)cdiv(w, TW), (unsigned)cdiv(h, TH), (unsigned)Z);
    hipLaunchKernelGGL(k_lift_fused_f16, grid, dim3(NTH), LDS_TOTAL, st, a);
    return check_launch("lift_f16_step");
}

}  // namespace lldwt
