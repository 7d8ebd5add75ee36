// cgp_f16x3.hip -- the context-fusion MLP of the tree + intra-subband model (cgp_out_xo_list, reference
// graphs/models/LiftingBasedDWT_net.py:282-289,357-365) on the fp16 matrix cores with split-fp16 operands, as ONE
// register-resident chain per wave (LDS holds only a copy of the biases; one barrier, at the start).
//
// Per subband g (groups = 3) and pixel:  [81 tree-context features | 12 causal taps of the quantised subband] (the masked
// context conv folded into layer 0 on the host, _fold_csc_into_cgp)  -> 162 -> 54 -> 18 -> 2 = (sigma, mu), LeakyReLU(0.01)
// between layers.  fp32 reference arithmetic: k_cgp_rate (cgp_fused.hip), which runs at 36 % of the fp32 MFMA roof.
//
// v_mfma_f32_32x32x16_f16, A = weights (32 output channels x 16 inputs per step), B = activations (16 inputs x 32 pixels):
//   * layer 0's B fragments come STRAIGHT from global memory: lane (pixel = l & 31, half = l >> 5) of k-step s needs input
//     channels 16 s + 8 half .. + 7 of ITS pixel -- 8 coalesced dword loads from the planar fp32 tensors, split to hi/lo
//     fp16 in registers.  Every input element is loaded exactly once by exactly one lane.
//   * a layer's 32x32 output tile D is, as it stands in the accumulator registers, the B operand of the next layer: registers
//     8 s .. 8 s + 7 of a lane form the fragment of k-step s, with the k order inside the step permuted
//     (k = 8 half + j  <->  output row 16 s + 8 (j >> 2) + 4 half + (j & 3); MI355X guide, "an accumulator tile as the next
//     MFMA's operand").  The next layer's WEIGHTS are packed in that permuted order, so no lane ever moves a value:
//     bias + LeakyReLU + split happen in place.
//   * weight fragments are streamed per wave from L2 (1 KB coalesced per load, pre-packed in step order, ring of 4 kept
//     three steps ahead by a sched_barrier after each load: left alone, the scheduler sinks the loads to their uses).
//   * one 32-pixel block per wave, two waves per SIMD: the partner wave computes while a wave waits for its inputs.
// Scales: every operand tensor is multiplied by a power of two before the split so that it cannot overflow fp16; the
// activations' bounds come from the wave's input maximum and the layers' max row L1 norms (fp16's exponent keeps the full
// 22-bit split precision over 18 binades, so a loose bound costs nothing).
// Output: params (planes, batch, 2*groups, h, w) = (sigma, mu) interleaved per subband, consumed by lldwt_gauss_rate.
#include "common.h"
#include "split_f16.h"
#include "lifting_f16.h"      // split_precision()

namespace lldwt {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));

namespace {

// the reference's dimensions (after the fold): 93 -> 162 -> 54 -> 18 -> 2
constexpr int C0 = 93, C1 = 162, C2 = 54, C3 = 18, C4 = 2, CPLC = 81;
constexpr int NK0 = 6;                 // k-steps of layer 0 (96 >= 93)
constexpr int NM0 = 6;                 // 32-channel output blocks of layer 0 (192 >= 162)
constexpr int NM1 = 2;                 // layer 1 (64 >= 54)
constexpr int NSTEP = NM0 * (NK0 + 2 * NM1) + 2 * NM1 + 2;     // 60 + 4 + 2 = 66 weight steps
constexpr int STEP_BYTES = 2048;       // hi fragment + lo fragment
constexpr int HDR_FLOATS = 64 + 192 + 64 + 32 + 32;            // scalars | b0 | b1 | b2 | b3
constexpr int BF_OFF = HDR_FLOATS * 4 + (NSTEP + 3) * STEP_BYTES;        // + 3 steps of padding for the prefetch ring
constexpr int GROUP_BYTES = BF_OFF + (NSTEP + 3) * (STEP_BYTES / 2);     // then a bf16 copy of the scaled weights (one-product bf16 mode)
constexpr int NB = 1;                  // pixel blocks of 32 per wave.  One block and two waves per SIMD (232 VGPRs): the other wave
                                       // computes while this one waits for its 48 input loads (two blocks in one wave at one wave
                                       // per SIMD share the weight stream, but nothing hides the input latency: 12 % slower)

__device__ __forceinline__ float pow2_scale(float amax) {      // s = 2^k with amax * s in [2^14, 2^15)
    if (!(amax > 0.f) || !(amax < 3.0e38f)) return 1.f;
    int e;
    (void)frexpf(amax, &e);
    int k = 15 - e;
    k = k > 120 ? 120 : (k < -120 ? -120 : k);
    return ldexpf(1.f, k);
}

// row of a 32x32 D tile held by (register q, lane half h)
__device__ __host__ __forceinline__ int drow(int q, int h) { return (q & 3) + 8 * (q >> 2) + 4 * h; }

// ---- pack: one workgroup per (plane, group)
__global__ void k_cgp16_pack(const float* __restrict__ w0, const float* __restrict__ b0, const float* __restrict__ w1,
                             const float* __restrict__ b1, const float* __restrict__ w2, const float* __restrict__ b2,
                             const float* __restrict__ w3, const float* __restrict__ b3, uint8_t* __restrict__ packed, int groups) {
    const int plane = blockIdx.y, g = blockIdx.x, tid = threadIdx.x;
    // per-plane PyTorch layouts: w_l (groups*c_{l+1}, c_l), b_l (groups*c_{l+1})
    const float* W[4] = {w0 + ((int64_t)plane * groups + g) * C1 * C0, w1 + ((int64_t)plane * groups + g) * C2 * C1,
                         w2 + ((int64_t)plane * groups + g) * C3 * C2, w3 + ((int64_t)plane * groups + g) * C4 * C3};
    const float* Bv[4] = {b0 + ((int64_t)plane * groups + g) * C1, b1 + ((int64_t)plane * groups + g) * C2,
                          b2 + ((int64_t)plane * groups + g) * C3, b3 + ((int64_t)plane * groups + g) * C4};
    const int cin[4] = {C0, C1, C2, C3}, cout[4] = {C1, C2, C3, C4};
    uint8_t* dst = packed + ((int64_t)plane * groups + g) * GROUP_BYTES;
    float* hdr = reinterpret_cast<float*>(dst);
    __shared__ float red[3][4];
    __shared__ float sc[4][3];           // per layer: max |w|, max row L1 norm, max |b|
    for (int l = 0; l < 4; ++l) {
        float mw = 0.f, ml1 = 0.f, mb = 0.f;
        for (int r = tid; r < cout[l]; r += 256) {
            float s = 0.f;
            for (int c = 0; c < cin[l]; ++c) {
                const float v = fabsf(W[l][r * cin[l] + c]);
                s += v;
                mw = fmaxf(mw, v);
            }
            ml1 = fmaxf(ml1, s);
            mb = fmaxf(mb, fabsf(Bv[l][r]));
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            mw = fmaxf(mw, __shfl_xor(mw, o, 64));
            ml1 = fmaxf(ml1, __shfl_xor(ml1, o, 64));
            mb = fmaxf(mb, __shfl_xor(mb, o, 64));
        }
        if ((tid & 63) == 0) { red[0][tid >> 6] = mw; red[1][tid >> 6] = ml1; red[2][tid >> 6] = mb; }
        __syncthreads();
        if (tid < 3) sc[l][tid] = fmaxf(fmaxf(red[tid][0], red[tid][1]), fmaxf(red[tid][2], red[tid][3]));
        __syncthreads();
    }
    float sw[4];
#pragma unroll
    for (int l = 0; l < 4; ++l) sw[l] = pow2_scale(sc[l][0]);
    if (tid < 4) {
        hdr[tid] = sw[tid];              // [0..3]  weight scales
        hdr[4 + tid] = sc[tid][1];       // [4..7]  max row L1 norm
        hdr[8 + tid] = sc[tid][2];       // [8..11] max |bias|
    }
    float* hb = hdr + 64;
    for (int i = tid; i < 192; i += 256) hb[i] = i < C1 ? Bv[0][i] : 0.f;
    for (int i = tid; i < 64; i += 256) hb[192 + i] = i < C2 ? Bv[1][i] : 0.f;
    for (int i = tid; i < 32; i += 256) hb[256 + i] = i < C3 ? Bv[2][i] : 0.f;
    for (int i = tid; i < 32; i += 256) hb[288 + i] = i < C4 ? Bv[3][i] : 0.f;
    _Float16* fr = reinterpret_cast<_Float16*>(dst + HDR_FLOATS * 4);
    for (int i = tid; i < (NSTEP + 3) * 512; i += 256) {       // one (hi, lo) pair per iteration
        const int j = i & 7, lane = (i >> 3) & 63, step = i >> 9;
        const int row = lane & 31, h = lane >> 5;
        float v = 0.f;
        if (step < NM0 * (NK0 + 2 * NM1)) {
            const int m0 = step / (NK0 + 2 * NM1), r = step % (NK0 + 2 * NM1);
            if (r < NK0) {               // layer 0: natural k order
                const int oc = 32 * m0 + row, ic = 16 * r + 8 * h + j;
                if (oc < C1 && ic < C0) v = W[0][oc * C0 + ic] * sw[0];
            } else {                     // layer 1 fed by block m0 of layer 0: permuted k order
                const int s = (r - NK0) / NM1, m1 = (r - NK0) % NM1;
                const int oc = 32 * m1 + row, ic = 32 * m0 + 16 * s + 8 * (j >> 2) + 4 * h + (j & 3);
                if (oc < C2 && ic < C1) v = W[1][oc * C1 + ic] * sw[1];
            }
        } else if (step < NM0 * (NK0 + 2 * NM1) + 2 * NM1) {   // layer 2 fed by block m1 of layer 1
            const int r = step - NM0 * (NK0 + 2 * NM1), m1 = r / 2, s = r % 2;
            const int ic = 32 * m1 + 16 * s + 8 * (j >> 2) + 4 * h + (j & 3);
            if (row < C3 && ic < C2) v = W[2][row * C2 + ic] * sw[2];
        } else if (step < NSTEP) {                              // layer 3
            const int s = step - NM0 * (NK0 + 2 * NM1) - 2 * NM1;
            const int ic = 16 * s + 8 * (j >> 2) + 4 * h + (j & 3);
            if (row < C4 && ic < C3) v = W[3][row * C3 + ic] * sw[3];
        }
        const _Float16 hi = (_Float16)v;
        fr[step * 1024 + lane * 8 + j] = hi;
        fr[step * 1024 + 512 + lane * 8 + j] = (_Float16)(v - (float)hi);
        reinterpret_cast<__bf16*>(dst + BF_OFF)[step * 512 + lane * 8 + j] = (__bf16)v;
    }
}

struct Cgp16Args {
    const float* plc;      // (Z, groups*81, h, w)
    const float* xq;       // (Z, groups, h, w)
    float* params;         // (Z, 2*groups, h, w)
    const uint8_t* packed;
    int batch, groups, h, w, K;
    int ntaps;
    int tap_dy[25], tap_dx[25];
    int cols;              // 64-pixel columns per image
    // ---- wavefront mode (real entropy coding, lldwt_cgp16_wavefront_step): the pixels of ONE anti-diagonal step t = x + slope * y
    // instead of a whole image; xq is the (Z, groups, h, w) tensor of values decoded so far (0 where not yet coded)
    int wf_t, wf_slope, wf_y0, wf_n;      // step, slope (K/2 + 1), first row of the step, pixels of the step
    const float* wf_y;                    // encoder: the coefficients (Z, groups, h, w); decoder: null
    float* wf_yhat;                       // encoder: == xq, receives symbol + mu at the step's pixels
    const float* wf_table;                // the scale table's first 63 entries (CDF index = number of entries < max(sigma, 0.11))
    int* wf_idx;                          // (Z, ntot, groups) CDF indexes in wavefront order; this step starts at wf_off
    int* wf_sym;                          // encoder: symbols, same layout
    float* wf_mu;                         // decoder: (Z, groups, wf_n) means of this step
    int64_t wf_ntot, wf_off;
    unsigned long long* stamps;   // diagnostics only (lldwt_set_diagnostics kind 2): [z][group][column][8] s_memtime stamps
    // TRAIN: the hidden activations after LeakyReLU in the layout of the unfused convs, h1 (Z, groups*162, hw), h2 (Z, groups*54,
    // hw), h3 (Z, groups*18, hw): what lldwt_cgp_bwd_split gates with and the 1x1 weight-gradient GEMMs read
    float* h1; float* h2; float* h3;
};
#define CGP_STAMP(i)                                                                                                    \
    if (a.stamps && lane == 0)                                                                                          \
        a.stamps[(((int64_t)blockIdx.z * gridDim.y + blockIdx.y) * a.cols + col) * 8 + (i)] =                           \
            (i) >= 6 ? __builtin_amdgcn_s_memrealtime() : __builtin_amdgcn_s_memtime();

// bias + LeakyReLU + rescale + split of one D tile into the two B fragments (k-steps 0 and 1) of the next layer.
// The next layer's power-of-two scale commutes with LeakyReLU, so it is folded into the dequantisation factor and into the
// bias (bsc = bias * snext, prepared once per tile for both pixel blocks): per value one FMA, LeakyReLU as max(t, 0.01 t),
// then the split.
// The biases are read from an LDS copy (filled once per workgroup, before the weight stream starts): a per-lane vector
// load at this point is waited for immediately, and because vmcnt retires in order that wait also drains the whole weight
// prefetch ring -- one full memory latency per tile (tools/cgp_stamps.py).  LDS reads count on lgkmcnt instead.
__device__ __forceinline__ void load_bias_scaled(const float* bias_lds, int h, float snext, float (&bsc)[16]) {
#pragma unroll
    for (int q = 0; q < 16; ++q) bsc[q] = bias_lds[drow(q, h)] * snext;
}
// TRAIN: the activation (v / snext: exact, a power of two) also goes to hout[row * hw] for the rows below nrows -- hout = this
// lane's pixel in channel 0 of the tile, or null for a pixel past the image; a register's 32 lanes of one half write 128 contiguous bytes
template <int PREC, bool TRAIN = false>
__device__ __forceinline__ void next_frags(const floatx16& acc, float k, const float (&bsc)[16], half8 (&bh)[2], half8 (&bl)[2],
                                           float* hout = nullptr, int64_t hw = 0, int h5 = 0, int nrows = 0, float inv_snext = 1.f) {
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float t = __builtin_fmaf(acc[8 * s + j], k, bsc[8 * s + j]);
            v[j] = fmaxf(t, 0.01f * t);
            if constexpr (TRAIN) {
                const int row = drow(8 * s + j, h5);
                if (hout && row < nrows) hout[(int64_t)row * hw] = v[j] * inv_snext;
            }
        }
        if constexpr (PREC == 0) split8v(v, bh[s], bl[s]);
        else bh[s] = cvt8<PREC>(v);
    }
}

// PREC (lldwt_set_precision): 0 = three MFMA products per MAC (split fp16), 1 / 2 = one product on fp16 / bf16 operands
// WF: one wavefront step of the real entropy coder (see Cgp16Args): the pixel list is the step's anti-diagonal and the epilogue
// turns (sigma, mu) into CDF index + symbol + dequantised value (LiftingBasedDWT_net.py:458-506: compress_ar's per-pixel work)
// TRAIN: the training forward -- also writes the hidden activations (see Cgp16Args); always the fp32-accurate PREC 0
template <int PREC, bool WF = false, bool TRAIN = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(TRAIN ? 2 : 3, TRAIN ? 2 : 3))) void k_cgp16(Cgp16Args a) {
    // (TRAIN: two waves per SIMD -- the store addresses do not fit the 168 registers of three)
    static_assert(!TRAIN || (PREC == 0 && !WF), "the training forward runs the split-fp16 arithmetic on whole images");
    constexpr int SB = PREC == 2 ? STEP_BYTES / 2 : STEP_BYTES;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int h5 = lane >> 5, pl = lane & 31;
    const int64_t z = blockIdx.z;
    const int plane = (int)(z / a.batch), g = blockIdx.y;
    const int col = blockIdx.x * 4 + wave;
    const int64_t hw = (int64_t)a.h * a.w;
    const uint8_t* grp = a.packed + ((int64_t)plane * a.groups + g) * GROUP_BYTES;
    const float* hdr = reinterpret_cast<const float*>(grp);
    __shared__ float sbias[320];                                 // header floats 64 .. 383: the four layers' biases
    for (int i = threadIdx.x; i < 320; i += 256) sbias[i] = hdr[64 + i];
    __syncthreads();                                             // the only barrier: before any wave can leave
    if (col >= a.cols) return;                                   // whole wave
    const float* bias0 = sbias, * bias1 = sbias + 192, * bias2 = sbias + 256, * bias3 = sbias + 288;
    const uint8_t* wst = grp + (PREC == 2 ? BF_OFF : HDR_FLOATS * 4) + lane * 16;
    CGP_STAMP(0)
    CGP_STAMP(6)

    // ---- layer-0 B fragments straight from global memory
    const float* plc = a.plc + (z * a.groups * CPLC + (int64_t)g * CPLC) * hw;
    const float* xq = a.xq + (z * a.groups + g) * hw;
    const int R = a.K / 2;
    float xin[NB][NK0][8];
    int pix[NB];
    bool valid[NB];
    float amax = 0.f;
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        int pc, y, x;
        if constexpr (WF) {
            const int i = col * (32 * NB) + nb * 32 + pl;
            valid[nb] = i < a.wf_n;
            y = a.wf_y0 + (valid[nb] ? i : 0);
            x = a.wf_t - a.wf_slope * y;
            pc = y * a.w + x;
        } else {
            const int64_t p = (int64_t)col * (32 * NB) + nb * 32 + pl;
            valid[nb] = p < hw;
            pc = (int)(valid[nb] ? p : hw - 1);
            y = pc / a.w;
            x = pc - y * a.w;
        }
        pix[nb] = pc;
#pragma unroll
        for (int k = 0; k < NK0; ++k) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                // channel of this element: 16k + j for the lanes of half 0, 16k + 8 + j for half 1.  Both are compile-time
                // constants per branch, so the tap tables are read with constant indexes (scalar loads, no scratch).
                constexpr int dummy_ = 0;
                (void)dummy_;
                const int c_lo = 16 * k + j, c_hi = 16 * k + 8 + j;
                float v = 0.f;
                auto fetch = [&](int c) -> float {          // c is a constant at every call site after unrolling
                    if (c < CPLC) return plc[(int64_t)c * hw + pc];
                    if (c < C0) {
                        const int yy = y + a.tap_dy[c - CPLC] - R, xx = x + a.tap_dx[c - CPLC] - R;
                        return (yy >= 0 && yy < a.h && xx >= 0 && xx < a.w) ? xq[(int64_t)yy * a.w + xx] : 0.f;
                    }
                    return 0.f;
                };
                if (c_hi < CPLC) {                          // both halves: tree-context features (one load, runtime channel)
                    v = plc[(int64_t)(c_lo + 8 * h5) * hw + pc];
                } else if (h5 == 0) {
                    v = fetch(c_lo);
                } else {
                    v = fetch(c_hi);
                }
                xin[nb][k][j] = v;
                amax = fmaxf(amax, fabsf(v));
            }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o, 64));
    // ---- scales: input from its maximum, hidden layers from bounds |h_l| <= |h_{l-1}|max * L1max_l + |b_l|max
    CGP_STAMP(1)
    const float s_in = pow2_scale(amax);
    const float bound0 = amax * hdr[4] + hdr[8];
    const float bound1 = bound0 * hdr[5] + hdr[9];
    const float bound2 = bound1 * hdr[6] + hdr[10];
    const float s1 = pow2_scale(bound0), s2 = pow2_scale(bound1), s3 = pow2_scale(bound2);
    const float inv0 = (1.f / s_in) * (1.f / hdr[0]), inv1 = (1.f / s1) * (1.f / hdr[1]);
    const float inv2 = (1.f / s2) * (1.f / hdr[2]), inv3 = (1.f / s3) * (1.f / hdr[3]);
    half8 b0h[NB][NK0], b0l[NB][NK0];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int k = 0; k < NK0; ++k) {
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = xin[nb][k][j] * s_in;
            if constexpr (PREC == 0) split8v(v, b0h[nb][k], b0l[nb][k]);
            else b0h[nb][k] = cvt8<PREC>(v);
        }

    // ---- weight stream: ring of 4 steps
    half8 ah[4], al[4];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        ah[i] = *reinterpret_cast<const half8*>(wst + i * SB);
        if constexpr (PREC == 0) al[i] = *reinterpret_cast<const half8*>(wst + i * SB + 1024);
    }
    int step = 0;
#define CGP16_NEXT()                                                                                  \
    {                                                                                                 \
        ah[(step + 3) & 3] = *reinterpret_cast<const half8*>(wst + (step + 3) * SB);                  \
        if constexpr (PREC == 0) al[(step + 3) & 3] = *reinterpret_cast<const half8*>(wst + (step + 3) * SB + 1024); \
        __builtin_amdgcn_sched_barrier(0);   /* the scheduler otherwise sinks these loads down to their use (3 steps later) */ \
    }
#define CGP16_MMA(ACC, BH, BL)                                                                        \
    {                                                                                                 \
        if constexpr (PREC == 0) {                                                                    \
            ACC = mma32<0>(al[step & 3], BH, ACC);                                                    \
            ACC = mma32<0>(ah[step & 3], BL, ACC);                                                    \
        }                                                                                             \
        ACC = mma32<PREC>(ah[step & 3], BH, ACC);                                                     \
    }

    floatx16 acc1[NB][NM1];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int m = 0; m < NM1; ++m)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc1[nb][m][q] = 0.f;

#pragma unroll
    for (int m0 = 0; m0 < NM0; ++m0) {
        floatx16 acc0[NB];
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc0[nb][q] = 0.f;
#pragma unroll
        for (int k = 0; k < NK0; ++k) {
            CGP16_NEXT()
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) CGP16_MMA(acc0[nb], b0h[nb][k], b0l[nb][k])
            ++step;
        }
        half8 b1h[NB][2], b1l[NB][2];
        {
            float bsc[16];
            load_bias_scaled(bias0 + 32 * m0, h5, s1, bsc);
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                if constexpr (TRAIN)
                    next_frags<PREC, true>(acc0[nb], inv0 * s1, bsc, b1h[nb], b1l[nb],
                                           valid[nb] ? a.h1 + ((z * a.groups + g) * C1 + 32 * m0) * hw + pix[nb] : nullptr, hw, h5,
                                           C1 - 32 * m0, 1.f / s1);
                else next_frags<PREC>(acc0[nb], inv0 * s1, bsc, b1h[nb], b1l[nb]);
            }
        }
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int m1 = 0; m1 < NM1; ++m1) {
                CGP16_NEXT()
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) CGP16_MMA(acc1[nb][m1], b1h[nb][s], b1l[nb][s])
                ++step;
            }
    }
    CGP_STAMP(2)
    // ---- layer 2 (54 -> 18)
    floatx16 acc2[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc2[nb][q] = 0.f;
#pragma unroll
    for (int m1 = 0; m1 < NM1; ++m1) {
        half8 b2h[NB][2], b2l[NB][2];
        {
            float bsc[16];
            load_bias_scaled(bias1 + 32 * m1, h5, s2, bsc);
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                if constexpr (TRAIN)
                    next_frags<PREC, true>(acc1[nb][m1], inv1 * s2, bsc, b2h[nb], b2l[nb],
                                           valid[nb] ? a.h2 + ((z * a.groups + g) * C2 + 32 * m1) * hw + pix[nb] : nullptr, hw, h5,
                                           C2 - 32 * m1, 1.f / s2);
                else next_frags<PREC>(acc1[nb][m1], inv1 * s2, bsc, b2h[nb], b2l[nb]);
            }
        }
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            CGP16_NEXT()
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) CGP16_MMA(acc2[nb], b2h[nb][s], b2l[nb][s])
            ++step;
        }
    }
    CGP_STAMP(3)
    // ---- layer 3 (18 -> 2)
    floatx16 acc3[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc3[nb][q] = 0.f;
    {
        half8 b3h[NB][2], b3l[NB][2];
        {
            float bsc[16];
            load_bias_scaled(bias2, h5, s3, bsc);
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                if constexpr (TRAIN)
                    next_frags<PREC, true>(acc2[nb], inv2 * s3, bsc, b3h[nb], b3l[nb],
                                           valid[nb] ? a.h3 + (z * a.groups + g) * C3 * hw + pix[nb] : nullptr, hw, h5, C3, 1.f / s3);
                else next_frags<PREC>(acc2[nb], inv2 * s3, bsc, b3h[nb], b3l[nb]);
            }
        }
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            CGP16_NEXT()
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) CGP16_MMA(acc3[nb], b3h[nb][s], b3l[nb][s])
            ++step;
        }
    }
#undef CGP16_NEXT
#undef CGP16_MMA
    CGP_STAMP(4)
    // ---- rows 0 (sigma) and 1 (mu) of the last tile: registers 0, 1 of the lanes with h == 0
    if constexpr (WF) {
        if (h5 == 0) {
#pragma unroll
            for (int nb = 0; nb < NB; ++nb)
                if (valid[nb]) {
                    const float sigma = acc3[nb][0] * inv3 + bias3[0], mu = acc3[nb][1] * inv3 + bias3[1];
                    const float sb = fmaxf(sigma, 0.11f);                         // GaussianConditional's scale bound (:32-33)
                    int idx = 0;
                    for (int k = 0; k < 63; ++k) idx += a.wf_table[k] < sb ? 1 : 0;   // build_indexes
                    const int i = col * (32 * NB) + nb * 32 + pl;
                    const int64_t o = (z * a.wf_ntot + a.wf_off + i) * a.groups + g;
                    a.wf_idx[o] = idx;
                    if (a.wf_y) {                                                 // encoder: symbol = round(y - mu), value = symbol + mu
                        const int64_t e = (z * a.groups + g) * hw + pix[nb];
                        const int sym = (int)rintf(a.wf_y[e] - mu);
                        a.wf_sym[o] = sym;
                        a.wf_yhat[e] = (float)sym + mu;
                    } else {
                        a.wf_mu[(z * a.groups + g) * a.wf_n + i] = mu;
                    }
                }
        }
    } else if (h5 == 0) {
        float* out = a.params + (z * 2 * a.groups + 2 * g) * hw;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
            if (valid[nb]) {
                out[pix[nb]] = acc3[nb][0] * inv3 + bias3[0];
                out[hw + pix[nb]] = acc3[nb][1] * inv3 + bias3[1];
            }
    }
    CGP_STAMP(5)
    CGP_STAMP(7)
}
#undef CGP_STAMP


// ================================================================================================================
// Backward-data of the stack on the same register chain (training; autograd of LiftingBasedDWT_net.py:282-289,357-365).
//   d3 = lrelu'(h3) . W3^T dparams (2 -> 18),  d2 = lrelu'(h2) . W2^T d3 (18 -> 54),  d1 = lrelu'(h1) . W1^T d2 (54 -> 162),
//   [dplc | dtaps] = W0^T d1 (162 -> 81 + 12)
// -- the transposed weights packed in step order with the permuted k order of an accumulator tile used as the next operand, exactly
// as the forward chain; the layers WIDEN here, so the 162-wide d1 is never held whole: each of its six 32-row tiles is gated, written
// out, split and at once accumulated into the three output tiles (65 weight steps: 1 + 4 + 6 x (4 + 6)).  The gates come from the
// SIGN of the stored activations (LeakyReLU: 1 or 0.01), loaded 16 rows per lane -- 128 contiguous bytes per half-wave and channel,
// like the stores of d1 / d2 / d3 (what the 1x1 weight-gradient GEMMs read).  Operand scales: the wave's max |dparams| and the
// transposed layers' max row L1 norms (gates <= 1), as the forward bounds.  Replaces k_cgp_bwd (fp32 MFMA, 7.3 ms per training step).
constexpr int NSTEPB = 1 + 2 * NM1 + NM0 * (2 * NM1 + 2 * 3);                 // 65
constexpr int HDRB_BYTES = 64 * 4;
constexpr int GROUPB_BYTES = HDRB_BYTES + (NSTEPB + 3) * STEP_BYTES;

__global__ void k_cgp16_pack_bwd(const float* __restrict__ w0, const float* __restrict__ w1, const float* __restrict__ w2,
                                 const float* __restrict__ w3, uint8_t* __restrict__ packed, int groups) {
    const int plane = blockIdx.y, g = blockIdx.x, tid = threadIdx.x;
    const float* W[4] = {w0 + ((int64_t)plane * groups + g) * C1 * C0, w1 + ((int64_t)plane * groups + g) * C2 * C1,
                         w2 + ((int64_t)plane * groups + g) * C3 * C2, w3 + ((int64_t)plane * groups + g) * C4 * C3};
    const int cin[4] = {C0, C1, C2, C3}, cout[4] = {C1, C2, C3, C4};
    uint8_t* dst = packed + ((int64_t)plane * groups + g) * GROUPB_BYTES;
    float* hdr = reinterpret_cast<float*>(dst);
    __shared__ float red[2][4];
    __shared__ float sc[4][2];           // per forward layer: max |w|, max COLUMN L1 norm (= row norm of the transposed layer)
    for (int l = 0; l < 4; ++l) {
        float mw = 0.f, ml1 = 0.f;
        for (int c = tid; c < cin[l]; c += 256) {
            float s_ = 0.f;
            for (int r = 0; r < cout[l]; ++r) {
                const float v = fabsf(W[l][r * cin[l] + c]);
                s_ += v;
                mw = fmaxf(mw, v);
            }
            ml1 = fmaxf(ml1, s_);
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            mw = fmaxf(mw, __shfl_xor(mw, o, 64));
            ml1 = fmaxf(ml1, __shfl_xor(ml1, o, 64));
        }
        if ((tid & 63) == 0) { red[0][tid >> 6] = mw; red[1][tid >> 6] = ml1; }
        __syncthreads();
        if (tid < 2) sc[l][tid] = fmaxf(fmaxf(red[tid][0], red[tid][1]), fmaxf(red[tid][2], red[tid][3]));
        __syncthreads();
    }
    // backward layer b uses forward layer 3 - b
    float sw[4];
#pragma unroll
    for (int b = 0; b < 4; ++b) sw[b] = pow2_scale(sc[3 - b][0]);
    if (tid < 4) {
        hdr[tid] = sw[tid];                    // [0..3] weight scales of the backward layers
        hdr[4 + tid] = sc[3 - tid][1];         // [4..7] max row L1 norm of the transposed layers
    }
    _Float16* fr = reinterpret_cast<_Float16*>(dst + HDRB_BYTES);
    for (int i = tid; i < (NSTEPB + 3) * 512; i += 256) {
        const int j = i & 7, lane = (i >> 3) & 63, step = i >> 9;
        const int row = lane & 31, h = lane >> 5;
        const int perm = 8 * (j >> 2) + 4 * h + (j & 3);          // row (inside a 16-row half tile) this k of a chained step holds
        float v = 0.f;
        if (step == 0) {                                           // 2 -> 18: natural k order
            const int out = row, in = 8 * h + j;
            if (out < C3 && in < C4) v = W[3][in * C3 + out] * sw[0];
        } else if (step < 1 + 2 * NM1) {                           // 18 -> 54: steps (m2, s)
            const int r = step - 1, m2 = r / 2, s_ = r % 2;
            const int out = 32 * m2 + row, in = 16 * s_ + perm;
            if (out < C2 && in < C3) v = W[2][in * C2 + out] * sw[1];
        } else if (step < NSTEPB) {
            const int r = step - 1 - 2 * NM1, m = r / 10, q = r % 10;
            if (q < 4) {                                           // 54 -> 162, output tile m: steps (t, s)
                const int t = q / 2, s_ = q % 2;
                const int out = 32 * m + row, in = 32 * t + 16 * s_ + perm;
                if (out < C1 && in < C2) v = W[1][in * C1 + out] * sw[2];
            } else {                                               // 162 -> 93 fed by tile m of d1: steps (s, n)
                const int s_ = (q - 4) / 3, n = (q - 4) % 3;
                const int out = 32 * n + row, in = 32 * m + 16 * s_ + perm;
                if (out < C0 && in < C1) v = W[0][in * C0 + out] * sw[3];
            }
        }
        const _Float16 hi = (_Float16)v;
        fr[step * 1024 + lane * 8 + j] = hi;
        fr[step * 1024 + 512 + lane * 8 + j] = (_Float16)(v - (float)hi);
    }
}

struct Cgp16BwdArgs {
    const float* dparams;   // (Z, 2*groups, hw)
    const float* h1; const float* h2; const float* h3;
    const uint8_t* packed;
    float* d1; float* d2; float* d3;
    float* dplc;            // (Z, groups*81, hw)
    float* dtaps;           // (Z, groups*12, hw)
    int batch, groups, cols;
    int64_t hw;
};

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_cgp16_bwd(Cgp16BwdArgs a) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int h5 = lane >> 5, pl = lane & 31;
    const int64_t z = blockIdx.z;
    const int plane = (int)(z / a.batch), g = blockIdx.y;
    const int col = blockIdx.x * 4 + wave;
    if (col >= a.cols) return;                                   // whole wave; no barrier in this kernel
    const int64_t hw = a.hw;
    const uint8_t* grp = a.packed + ((int64_t)plane * a.groups + g) * GROUPB_BYTES;
    const float* hdr = reinterpret_cast<const float*>(grp);
    const uint8_t* wst = grp + HDRB_BYTES + lane * 16;
    const int64_t p = (int64_t)col * 32 + pl;
    const bool valid = p < hw;
    const int64_t pc = valid ? p : hw - 1;
    const int64_t zg = z * a.groups + g;
    const float* dp = a.dparams + zg * 2 * hw;
    const float v0 = valid ? dp[pc] : 0.f, v1 = valid ? dp[hw + pc] : 0.f;
    float amax = fmaxf(fabsf(v0), fabsf(v1));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o, 64));
    const float s_in = pow2_scale(amax);
    const float bound3 = amax * hdr[4], bound2 = bound3 * hdr[5], bound1 = bound2 * hdr[6];
    const float s3 = pow2_scale(bound3), s2 = pow2_scale(bound2), s1 = pow2_scale(bound1);
    const float inv0 = (1.f / s_in) * (1.f / hdr[0]), inv1 = (1.f / s3) * (1.f / hdr[1]);
    const float inv2 = (1.f / s2) * (1.f / hdr[2]), inv3 = (1.f / s1) * (1.f / hdr[3]);

    half8 ah[4], al[4];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        ah[i] = *reinterpret_cast<const half8*>(wst + i * STEP_BYTES);
        al[i] = *reinterpret_cast<const half8*>(wst + i * STEP_BYTES + 1024);
    }
    int step = 0;
#define CGPB_NEXT()                                                                                   \
    {                                                                                                 \
        ah[(step + 3) & 3] = *reinterpret_cast<const half8*>(wst + (step + 3) * STEP_BYTES);          \
        al[(step + 3) & 3] = *reinterpret_cast<const half8*>(wst + (step + 3) * STEP_BYTES + 1024);   \
        __builtin_amdgcn_sched_barrier(0);                                                            \
    }
#define CGPB_MMA(ACC, BH, BL)                                                                         \
    {                                                                                                 \
        ACC = mma32<0>(al[step & 3], BH, ACC);                                                        \
        ACC = mma32<0>(ah[step & 3], BL, ACC);                                                        \
        ACC = mma32<0>(ah[step & 3], BH, ACC);                                                        \
    }
    // the stored activations of one tile (rows below nrows; 1 elsewhere: those rows carry zero weights)
    auto load_h = [&](const float* hbase, int nrows, float (&hv)[16]) {
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int row = drow(q, h5);
            hv[q] = row < nrows ? hbase[(int64_t)row * hw + pc] : 1.f;
        }
    };
    // gate, write out (value / snext), split into the next layer's two k-steps
    auto gate_frags = [&](const floatx16& acc, float k, const float (&hv)[16], float* dbase, int nrows, float inv_snext,
                          half8 (&bh)[2], half8 (&bl)[2]) {
#pragma unroll
        for (int s_ = 0; s_ < 2; ++s_) {
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int q = 8 * s_ + j, row = drow(q, h5);
                v[j] = acc[q] * k * (hv[q] > 0.f ? 1.f : 0.01f);
                if (valid && row < nrows) dbase[(int64_t)row * hw + pc] = v[j] * inv_snext;
            }
            split8v(v, bh[s_], bl[s_]);
        }
    };
    floatx16 zero16;
#pragma unroll
    for (int q = 0; q < 16; ++q) zero16[q] = 0.f;

    // ---- 2 -> 18
    float hv3[16];
    load_h(a.h3 + zg * C3 * hw, C3, hv3);
    half8 f3h[2], f3l[2];
    {
        float vin[8] = {h5 == 0 ? v0 * s_in : 0.f, h5 == 0 ? v1 * s_in : 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        half8 bh, bl;
        split8v(vin, bh, bl);
        floatx16 acc = zero16;
        CGPB_NEXT()
        CGPB_MMA(acc, bh, bl)
        ++step;
        gate_frags(acc, inv0 * s3, hv3, a.d3 + zg * C3 * hw, C3, 1.f / s3, f3h, f3l);
    }
    // ---- 18 -> 54
    half8 f2h[NM1][2], f2l[NM1][2];
#pragma unroll
    for (int m2 = 0; m2 < NM1; ++m2) {
        float hv2[16];
        load_h(a.h2 + (zg * C2 + 32 * m2) * hw, C2 - 32 * m2, hv2);
        floatx16 acc = zero16;
#pragma unroll
        for (int s_ = 0; s_ < 2; ++s_) {
            CGPB_NEXT()
            CGPB_MMA(acc, f3h[s_], f3l[s_])
            ++step;
        }
        gate_frags(acc, inv1 * s2, hv2, a.d2 + (zg * C2 + 32 * m2) * hw, C2 - 32 * m2, 1.f / s2, f2h[m2], f2l[m2]);
    }
    // ---- 54 -> 162 -> 93, one 32-row tile of d1 at a time
    floatx16 accO[3] = {zero16, zero16, zero16};
    float hvn[16];                                   // the gates of the NEXT tile: requested a whole tile (10 weight steps) ahead
    load_h(a.h1 + zg * C1 * hw, C1, hvn);
#pragma unroll
    for (int m = 0; m < NM0; ++m) {
        float hv1[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) hv1[q] = hvn[q];
        if (m + 1 < NM0) load_h(a.h1 + (zg * C1 + 32 * (m + 1)) * hw, C1 - 32 * (m + 1), hvn);
        floatx16 acc = zero16;
#pragma unroll
        for (int t = 0; t < NM1; ++t)
#pragma unroll
            for (int s_ = 0; s_ < 2; ++s_) {
                CGPB_NEXT()
                CGPB_MMA(acc, f2h[t][s_], f2l[t][s_])
                ++step;
            }
        half8 f1h[2], f1l[2];
        gate_frags(acc, inv2 * s1, hv1, a.d1 + (zg * C1 + 32 * m) * hw, C1 - 32 * m, 1.f / s1, f1h, f1l);
#pragma unroll
        for (int s_ = 0; s_ < 2; ++s_)
#pragma unroll
            for (int n = 0; n < 3; ++n) {
                CGPB_NEXT()
                CGPB_MMA(accO[n], f1h[s_], f1l[s_])
                ++step;
            }
    }
#undef CGPB_NEXT
#undef CGPB_MMA
    // ---- the input gradient: rows 0..80 -> dplc, rows 81..92 -> dtaps
    if (valid) {
        float* dpl = a.dplc + zg * CPLC * hw + pc;
        float* dtp = a.dtaps + zg * (C0 - CPLC) * hw + pc;
#pragma unroll
        for (int n = 0; n < 3; ++n)
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int row = 32 * n + drow(q, h5);
                const float v = accO[n][q] * inv3;
                if (row < CPLC) dpl[(int64_t)row * hw] = v;
                else if (row < C0) dtp[(int64_t)(row - CPLC) * hw] = v;
            }
    }
}

}  // namespace
}  // namespace lldwt
using namespace lldwt;

static unsigned long long* g_cgp_stamps = nullptr;
static int64_t g_cgp_stamps_bytes = 0;
namespace lldwt { void cgp16_set_stamps(void* p, int64_t nbytes) { g_cgp_stamps = reinterpret_cast<unsigned long long*>(p); g_cgp_stamps_bytes = p ? nbytes : 0; } }

extern "C" int64_t lldwt_cgp16_packed_bytes(int c0, int c1, int c2, int c3, int groups) {
    if (c0 != C0 || c1 != C1 || c2 != C2 || c3 != C3 || groups <= 0) return -1;     // the reference's dimensions only
    return (int64_t)groups * GROUP_BYTES;
}

extern "C" int lldwt_cgp16_pack(const float* w0, const float* b0, const float* w1, const float* b1, const float* w2,
                                const float* b2, const float* w3, const float* b3, void* packed, int64_t planes, int c0,
                                int c1, int c2, int c3, int groups, void* stream) {
    LLDWT_REQUIRE(lldwt_cgp16_packed_bytes(c0, c1, c2, c3, groups) > 0, "cgp16_pack: built for 93 -> 162 -> 54 -> 18 -> 2 (got %d,%d,%d,%d)", c0, c1, c2, c3);
    LLDWT_REQUIRE(w0 && b0 && w1 && b1 && w2 && b2 && w3 && b3 && packed && planes > 0 && planes <= 65535, "cgp16_pack: bad arguments");
    hipLaunchKernelGGL(k_cgp16_pack, dim3((unsigned)groups, (unsigned)planes), dim3(256), 0, (hipStream_t)stream, w0, b0, w1, b1,
                       w2, b2, w3, b3, reinterpret_cast<uint8_t*>(packed), groups);
    return check_launch("cgp16_pack");
}

static int cgp16_params_impl(const float* plc, const float* xq, const void* packed, float* params, float* h1, float* h2, float* h3,
                             int64_t planes, int64_t batch, int64_t h, int64_t w_, int groups, int K, uint32_t tap_mask, void* stream);

extern "C" int lldwt_cgp16_params(const float* plc, const float* xq, const void* packed, float* params, int64_t planes,
                                  int64_t batch, int64_t h, int64_t w_, int groups, int K, uint32_t tap_mask, void* stream) {
    return cgp16_params_impl(plc, xq, packed, params, nullptr, nullptr, nullptr, planes, batch, h, w_, groups, K, tap_mask, stream);
}

// the training forward of the cgp stack on the same register chain (always the fp32-accurate split-fp16 arithmetic): (sigma, mu) as
// lldwt_cgp16_params + the hidden activations after LeakyReLU, h1 (Z, groups*162, hw), h2 (Z, groups*54, hw), h3 (Z, groups*18, hw) --
// what lldwt_cgp_rate_train_ctx writes on fp32 MFMA (6.0 ms per training step of configs[2] against 1.4 for the eval chain)
extern "C" int lldwt_cgp16_params_train(const float* plc, const float* xq, const void* packed, float* params, float* h1, float* h2,
                                        float* h3, int64_t planes, int64_t batch, int64_t h, int64_t w_, int groups, int K,
                                        uint32_t tap_mask, void* stream) {
    LLDWT_REQUIRE(h1 && h2 && h3, "cgp16_params_train: null output");
    return cgp16_params_impl(plc, xq, packed, params, h1, h2, h3, planes, batch, h, w_, groups, K, tap_mask, stream);
}

static int cgp16_params_impl(const float* plc, const float* xq, const void* packed, float* params, float* h1, float* h2, float* h3,
                             int64_t planes, int64_t batch, int64_t h, int64_t w_, int groups, int K, uint32_t tap_mask, void* stream) {
    LLDWT_REQUIRE(plc && xq && packed && params && planes > 0 && batch > 0 && h > 0 && w_ > 0 && groups > 0, "cgp16_params: bad arguments");
    LLDWT_REQUIRE(K == 3 || K == 5, "cgp16_params: K=%d unsupported", K);
    LLDWT_REQUIRE(planes * batch <= 65535 && groups <= 65535, "cgp16_params: grid too large");
    LLDWT_REQUIRE((int64_t)CPLC * h * w_ < ((int64_t)1 << 31), "cgp16_params: image too large for 32-bit offsets");
    Cgp16Args a;
    a.plc = plc; a.xq = xq; a.params = params; a.packed = reinterpret_cast<const uint8_t*>(packed);
    a.h1 = h1; a.h2 = h2; a.h3 = h3;
    a.batch = (int)batch; a.groups = groups; a.h = (int)h; a.w = (int)w_; a.K = K;
    int n = 0;
    for (int t = 0; t < K * K; ++t)
        if ((tap_mask >> t) & 1u) {
            LLDWT_REQUIRE(n < 25, "cgp16_params: too many taps");
            a.tap_dy[n] = t / K;
            a.tap_dx[n] = t % K;
            ++n;
        }
    LLDWT_REQUIRE(n == C0 - CPLC, "cgp16_params: %d live taps, the folded first layer expects %d", n, C0 - CPLC);
    a.ntaps = n;
    a.cols = (int)cdiv(h * w_, 32 * NB);
    {   // diagnostics (tools/cgp_stamps.py, lldwt_set_diagnostics): only when the registered buffer holds this grid's stamps
        const int64_t need = (int64_t)planes * batch * groups * a.cols * 8 * 8;
        a.stamps = (g_cgp_stamps && g_cgp_stamps_bytes >= need) ? g_cgp_stamps : nullptr;
    }
    dim3 grid((unsigned)cdiv(a.cols, 4), (unsigned)groups, (unsigned)(planes * batch));
    const int prec = split_precision();
    if (h1) hipLaunchKernelGGL((k_cgp16<0, false, true>), grid, dim3(256), 0, (hipStream_t)stream, a);
    else if (prec == 1) hipLaunchKernelGGL((k_cgp16<1, false>), grid, dim3(256), 0, (hipStream_t)stream, a);
    else if (prec == 2) hipLaunchKernelGGL((k_cgp16<2, false>), grid, dim3(256), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL((k_cgp16<0, false>), grid, dim3(256), 0, (hipStream_t)stream, a);
    return check_launch("cgp16_params");
}

// the decoder's second half of a wavefront step: yhat[pixel] = symbol + mu
__global__ void k_wf_apply(const int* __restrict__ sym, const float* __restrict__ mu, float* __restrict__ yhat, int groups, int h,
                           int w, int t, int slope, int y0, int n, int64_t ntot, int64_t off) {
    const int64_t z = blockIdx.z;
    const int g = blockIdx.y, i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int y = y0 + i, x = t - slope * y;
    yhat[(z * groups + g) * (int64_t)h * w + (int64_t)y * w + x] = (float)sym[(z * ntot + off + i) * groups + g] + mu[(z * groups + g) * n + i];
}

// pixels of wavefront step t: rows y0 .. y0 + n - 1 with x = t - slope * y inside [0, w)
static inline void wf_range(int t, int slope, int h, int w, int& y0, int& n) {
    int lo = t - (w - 1);
    lo = lo <= 0 ? 0 : (lo + slope - 1) / slope;
    int hi = t / slope;
    if (hi > h - 1) hi = h - 1;
    y0 = lo;
    n = hi >= lo ? hi - lo + 1 : 0;
}

extern "C" int lldwt_cgp16_wavefront_step(const float* plc, float* yhat, const float* y, const void* packed, const float* table63,
                                          int* idx, int* sym, float* mu, int64_t planes, int64_t batch, int64_t h, int64_t w_,
                                          int groups, int K, uint32_t tap_mask, int t, int64_t ntot, int64_t off, void* stream) {
    LLDWT_REQUIRE(plc && yhat && packed && table63 && idx && (y ? sym != nullptr : mu != nullptr), "cgp16_wavefront_step: null pointer");
    LLDWT_REQUIRE(planes > 0 && batch > 0 && h > 0 && w_ > 0 && groups > 0 && (K == 3 || K == 5), "cgp16_wavefront_step: bad arguments");
    LLDWT_REQUIRE(planes * batch <= 65535 && groups <= 65535, "cgp16_wavefront_step: grid too large");
    LLDWT_REQUIRE((int64_t)CPLC * h * w_ < ((int64_t)1 << 31), "cgp16_wavefront_step: image too large for 32-bit offsets");
    const int slope = K / 2 + 1;
    LLDWT_REQUIRE(t >= 0 && t <= (int)(w_ - 1 + slope * (h - 1)), "cgp16_wavefront_step: step %d outside 0 .. %ld", t, (long)(w_ - 1 + slope * (h - 1)));
    Cgp16Args a;
    a.plc = plc; a.xq = yhat; a.params = nullptr; a.packed = reinterpret_cast<const uint8_t*>(packed);
    a.batch = (int)batch; a.groups = groups; a.h = (int)h; a.w = (int)w_; a.K = K;
    int n = 0;
    for (int q = 0; q < K * K; ++q)
        if ((tap_mask >> q) & 1u) {
            LLDWT_REQUIRE(n < 25, "cgp16_wavefront_step: too many taps");
            a.tap_dy[n] = q / K;
            a.tap_dx[n] = q % K;
            ++n;
        }
    LLDWT_REQUIRE(n == C0 - CPLC, "cgp16_wavefront_step: %d live taps, the folded first layer expects %d", n, C0 - CPLC);
    a.ntaps = n;
    a.wf_t = t; a.wf_slope = slope;
    wf_range(t, slope, (int)h, (int)w_, a.wf_y0, a.wf_n);
    if (a.wf_n == 0) return LLDWT_OK;
    LLDWT_REQUIRE(off >= 0 && off + a.wf_n <= ntot, "cgp16_wavefront_step: step does not fit the output (off %ld + %d > %ld)", (long)off, a.wf_n, (long)ntot);
    a.wf_y = y; a.wf_yhat = yhat; a.wf_table = table63; a.wf_idx = idx; a.wf_sym = sym; a.wf_mu = mu; a.wf_ntot = ntot; a.wf_off = off;
    a.cols = (int)cdiv(a.wf_n, 32 * NB);
    a.stamps = nullptr;
    dim3 grid((unsigned)cdiv(a.cols, 4), (unsigned)groups, (unsigned)(planes * batch));
    // always the fp32-accurate arithmetic: encoder and decoder must agree bit for bit whatever the precision mode of the process
    hipLaunchKernelGGL((k_cgp16<0, true>), grid, dim3(256), 0, (hipStream_t)stream, a);
    return check_launch("cgp16_wavefront_step");
}

extern "C" int lldwt_wavefront_apply(const int* sym, const float* mu, float* yhat, int64_t planes, int64_t batch, int64_t h,
                                     int64_t w_, int groups, int K, int t, int64_t ntot, int64_t off, void* stream) {
    LLDWT_REQUIRE(sym && mu && yhat && planes > 0 && batch > 0 && h > 0 && w_ > 0 && groups > 0 && (K == 3 || K == 5), "wavefront_apply: bad arguments");
    const int slope = K / 2 + 1;
    int y0, n;
    wf_range(t, slope, (int)h, (int)w_, y0, n);
    if (n == 0) return LLDWT_OK;
    LLDWT_REQUIRE(off >= 0 && off + n <= ntot, "wavefront_apply: step does not fit");
    dim3 grid((unsigned)cdiv(n, 64), (unsigned)groups, (unsigned)(planes * batch));
    hipLaunchKernelGGL(k_wf_apply, grid, dim3(64), 0, (hipStream_t)stream, sym, mu, yhat, groups, (int)h, (int)w_, t, slope, y0, n, ntot, off);
    return check_launch("wavefront_apply");
}

extern "C" int64_t lldwt_cgp16_bwd_packed_bytes(int c0, int c1, int c2, int c3, int groups) {
    if (c0 != C0 || c1 != C1 || c2 != C2 || c3 != C3 || groups <= 0) return -1;
    return (int64_t)groups * GROUPB_BYTES;
}

extern "C" int lldwt_cgp16_pack_bwd(const float* w0, const float* w1, const float* w2, const float* w3, void* packed, int64_t planes,
                                    int c0, int c1, int c2, int c3, int groups, void* stream) {
    LLDWT_REQUIRE(lldwt_cgp16_bwd_packed_bytes(c0, c1, c2, c3, groups) > 0, "cgp16_pack_bwd: built for 93 -> 162 -> 54 -> 18 -> 2 (got %d,%d,%d,%d)", c0, c1, c2, c3);
    LLDWT_REQUIRE(w0 && w1 && w2 && w3 && packed && planes > 0 && planes <= 65535, "cgp16_pack_bwd: bad arguments");
    hipLaunchKernelGGL(k_cgp16_pack_bwd, dim3((unsigned)groups, (unsigned)planes), dim3(256), 0, (hipStream_t)stream, w0, w1, w2, w3,
                       reinterpret_cast<uint8_t*>(packed), groups);
    return check_launch("cgp16_pack_bwd");
}

extern "C" int lldwt_cgp16_bwd(const float* dparams, const float* h1, const float* h2, const float* h3, const void* packed_bwd,
                               float* d1, float* d2, float* d3, float* dplc, float* dtaps, int64_t planes, int64_t batch, int64_t hw,
                               int groups, void* stream) {
    LLDWT_REQUIRE(dparams && h1 && h2 && h3 && packed_bwd && d1 && d2 && d3 && dplc && dtaps && planes > 0 && batch > 0 && hw > 0 &&
                      groups > 0 && planes * batch <= 65535 && groups <= 65535, "cgp16_bwd: bad arguments");
    Cgp16BwdArgs a;
    a.dparams = dparams; a.h1 = h1; a.h2 = h2; a.h3 = h3; a.packed = reinterpret_cast<const uint8_t*>(packed_bwd);
    a.d1 = d1; a.d2 = d2; a.d3 = d3; a.dplc = dplc; a.dtaps = dtaps;
    a.batch = (int)batch; a.groups = groups; a.hw = hw; a.cols = (int)cdiv(hw, 32);
    dim3 grid((unsigned)cdiv(a.cols, 4), (unsigned)groups, (unsigned)(planes * batch));
    hipLaunchKernelGGL(k_cgp16_bwd, grid, dim3(256), 0, (hipStream_t)stream, a);
    return check_launch("cgp16_bwd");
}
