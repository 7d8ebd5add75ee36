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
// LDS: skip 6 KB + T1 77 KB + T2 60 KB (+ T3 aliasing T1, D aliasing T2) + 12 KB fp16 skip images = 155.5 KB: one
// 512-thread workgroup per CU.
// conv1's B operand: the skip patch is also kept as SCALED SPLIT fp16 (hi and lo images, plus a copy of each shifted by
// one element), and conv1's k order is (row dy = lane group, three dx PAIRS (0,1) (2,3) (4,5*)) + one pair of row 4 per
// lane group (* = a real neighbour value against a zero weight): a lane's 8 k values are four aligned 4-byte LDS reads
// that land as packed fp16 pairs -- no per-tap gather, no scaling, no split in the loop.
#include "lifting_f16.h"
#include "split_f16.h"

namespace lldwt {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

namespace {

// Arithmetic of the 16-channel convolutions (lldwt_set_precision).  PREC 0: split fp16, three MFMA products per fp32 MAC (hi hi +
// hi lo + lo hi; fp32-level accuracy, the default and what every fp32 parity bar is checked with).  PREC 1 / 2: ONE product per
// MAC on fp16 / bf16 operands (BASELINE configs[4] "fp16", configs[1] "bf16"; their own tolerance class): the lo images, lo
// weight fragments and two thirds of the MFMAs are not touched at all.  16-byte fragments travel as half8 whatever they hold.
template <int PREC>
__device__ __forceinline__ floatx4 mma1(const half8& a, const half8& b, const floatx4& acc) {
    if constexpr (PREC == 2)
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc, 0, 0, 0);
    else
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc, 0, 0, 0);
}
template <int PREC>
__device__ __forceinline__ floatx4 mma3(const half8& ah, const half8& al, const half8& bh, const half8& bl, floatx4 acc) {
    if constexpr (PREC == 0) {
        acc = mma1<0>(al, bh, acc);
        acc = mma1<0>(ah, bl, acc);
    }
    return mma1<PREC>(ah, bh, acc);
}
// weight fragments of global k-step `step` (conv1 | conv2 | conv3 | conv4 | composed): hi and lo fp16 in the f16 section, a bf16
// copy of the (scaled) weights behind the section's tail
template <int PREC>
__device__ __forceinline__ void ldA(const _Float16* __restrict__ frag, int step, int lane, half8& h, half8& l) {
    if constexpr (PREC == 2) {
        h = *reinterpret_cast<const half8*>(frag + LF_H_BF + step * 512 + lane * 8);
    } else {
        h = *reinterpret_cast<const half8*>(frag + step * LF_FRAG + lane * 8);
        if constexpr (PREC == 0) l = *reinterpret_cast<const half8*>(frag + step * LF_FRAG + 512 + lane * 8);
    }
}
// fp32 -> the 2-byte operand type of PREC (bits): fp16 hi, or bf16
template <int PREC>
__device__ __forceinline__ unsigned short to16(float v) {
    if constexpr (PREC == 2) return __builtin_bit_cast(unsigned short, (__bf16)v);
    else return __builtin_bit_cast(unsigned short, (_Float16)v);
}

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
constexpr int LDS_S16 = LDS_T2 + 4 * N2 * 16;                         // skip patch as scaled fp16: [copy 0|1][hi|lo][NS16]
constexpr int NS16 = NS + 48;                                         // copy 1 is copy 0 shifted by one element (aligned pair reads
constexpr int LDS_TOTAL = LDS_S16 + 4 * NS16 * 2;                     // at odd columns); tails zeroed; copy 1 starts 16 banks after
static_assert((2 * NS16 * 2) % 128 == 64, "even- and odd-column lanes read different banks");   // copy 0.  159,360 B
// sequential path (debug check, TRAIN, BWD): T3 takes the front of the T1 region and the D image sits behind it, so that the T2 image
// stays whole until the tile ends -- its last rows are handed down to the tile below like on the composed path (T1 is not: T3
// overwrites it), and the next tile's staged operands use the composed path's slots in T2 rows 8..15
constexpr int LDS_T3 = LDS_T1, LDS_D = LDS_T1 + 4 * N3 * 16;
static_assert(4 * N3 * 16 + ND * 8 * 4 <= 4 * N1 * 16, "T3 and the D image of the sequential path fit the T1 region");
// ---- vertical reuse (composed path).  A workgroup walks DOWN a column of tiles (a "run"); the tile below needs t1 on image
// rows y0 + 10 .. y0 + 38 and t2 on y0 + 12 .. y0 + 36, of which the first 12 (t1) and 8 (t2) rows are the LAST rows of this
// tile's T images: they are handed down (moved to the top of the images through registers while the skip patch of the next
// tile is built), and the tile below computes 16 new rows of each instead of 28 / 24 -- 44 conv1 tiles instead of 77, 40
// conv2 tiles instead of 60.  Everything that used to alias T1 / T2 now lives in the rows that die:
//   T1 rows 0..15 of each of the four arrays (11 264 B): a quarter of the D image (output rows 4k..4k+3, 7 680 B), then a
//   quarter of the border strips' t3v (56 positions x 64 B);  T2 rows 8..15 (5 120 B per array): the next tile's staged
//   operands (20 480 B across the four arrays).
constexpr int KEEP1 = 12, KEEP2 = 8;                                  // rows handed down
constexpr int T1A = N1 * 16, T2A = N2 * 16;                           // bytes per array
constexpr int NT1C = (N1 - KEEP1 * R1W) / 16, NT2C = (N2 - KEEP2 * R2W) / 16;      // 44, 40
// sequential path: only T2 is handed down, so a continuing tile needs t1 on rows 8..27 (the 20 rows under its 16 new t2 rows)
constexpr int NT1S = (N1 - KEEP2 * R1W) / 16;                                      // 55
static_assert((KEEP2 * R1W) % 16 == 0, "a continuing tile of the sequential path starts t1 on a whole MFMA tile");
static_assert((KEEP1 * R1W) % 16 == 0 && (KEEP2 * R2W) % 16 == 0, "a continuing tile starts on a whole MFMA tile");
constexpr int DPIECE_PX = 4 * R2W, DPIECE_BYTES = DPIECE_PX * 12 * 4;  // 160 px, 7 680 B
constexpr int T3V_PER = 56;
constexpr int T2HALF = KEEP2 * R2W * 16;                              // 5 120 B: rows 0..7 (and rows 8..15) of a T2 array
static_assert(DPIECE_BYTES + T3V_PER * LF_C * 4 <= (N1 - KEEP1 * R1W) * 16, "D quarter + t3v quarter fit the dying rows of a T1 array");
static_assert(10 * NTH * 4 <= 4 * T2HALF && 2 * T2HALF == (N2 - KEEP2 * R2W) * 16, "a staging quarter fits half of the dying rows of a T2 array");
static_assert(4 * T3V_PER >= 4 * R3W + 4 * (TH + 4), "the strip buffer holds every tile's border strips");
constexpr int LDS_W4L = LDS_TOTAL;                                    // conv4's fp32 weights [tap][channel] (border tiles)
constexpr int LDS_TOTAL2 = LDS_W4L + LF_KK * LF_C * 4;                // 160 960 B
static_assert(LDS_TOTAL2 <= 160 * 1024, "one workgroup per CU");
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

// tanh(acc * k + b) with the factor 2 log2(e) of exp(2|x|) = 2^(2 log2(e) |x|) already folded into kc = k * 2 log2(e) and
// bc = b * 2 log2(e): u = acc * kc + bc has the sign of the argument, exp2(|u|) is exp(2|x|) -- one multiply less per value
// than fast_tanh(fma(acc, k, b)) (common.h), same formula otherwise
constexpr float TWO_LOG2E = 2.88539008177792681472f;
__device__ __forceinline__ float tanh_scaled(float acc, float kc, float bc) {
    const float u = __builtin_fmaf(acc, kc, bc);
    const float e = __builtin_amdgcn_exp2f(fabsf(u));
    const float t = __builtin_fmaf(-2.f, __builtin_amdgcn_rcpf(e + 1.f), 1.f);
    return copysignf(t, u);
}

// the same times a per-pixel factor pin (ACT_SCALE inside the image, 0 outside) folded into the last FMA: pin - 2 pin r = pin (1 - 2 r),
// bit-identical to tanh_scaled(..) * pin for a power-of-two pin, one vector instruction less per value (m2pin = -2 pin)
__device__ __forceinline__ float tanh_scaled_masked(float acc, float kc, float bc, float pin, float m2pin) {
    const float u = __builtin_fmaf(acc, kc, bc);
    const float e = __builtin_amdgcn_exp2f(fabsf(u));
    const float t = __builtin_fmaf(m2pin, __builtin_amdgcn_rcpf(e + 1.f), pin);
    return copysignf(t, u);
}

// effective tap t = dy*5+dx of an orientation -> index into the PyTorch (kh,kw) weight
__device__ __forceinline__ int srctap(int t, int orient) { return orient == 0 ? t : (t % LF_K) * LF_K + t / LF_K; }

// 1024 threads: the composed 9x9 kernels are 1 377 sums of up to 400 double-precision products each; 256 threads took 0.4 ms per
// block (it runs at every weight update, i.e. every training step)
constexpr int PACK_NT = 1024, PACK_NW = PACK_NT / 64;
__global__ __launch_bounds__(PACK_NT) void k_lift_f16_pack(const float* __restrict__ w1, const float* __restrict__ w2, const float* __restrict__ w3,
                                const float* __restrict__ w4, const float* __restrict__ b1, const float* __restrict__ b3,
                                const float* __restrict__ b4, float* __restrict__ packed, int64_t plane_stride, int f16_off, int compose) {
    // compose == 0: the composed 9x9 kernels (read by the eval path only) are left zero -- the packs of the training forward and of
    // the backward-data chain are rebuilt at every weight update, and the composition is nine tenths of this kernel's time
    const int orient = blockIdx.x, plane = blockIdx.y, tid = threadIdx.x;
    b1 += (int64_t)plane * LF_C;
    b3 += (int64_t)plane * LF_C;
    b4 += plane;
    w1 += (int64_t)plane * LF_C * LF_KK;
    w2 += (int64_t)plane * LF_C * LF_C * LF_KK;
    w3 += (int64_t)plane * LF_C * LF_C * LF_KK;
    w4 += (int64_t)plane * LF_C * LF_KK;
    __shared__ float red[4][PACK_NW];
    float m[4] = {0.f, 0.f, 0.f, 0.f};
    for (int i = tid; i < LF_C * LF_KK; i += PACK_NT) m[0] = fmaxf(m[0], fabsf(w1[i]));
    for (int i = tid; i < LF_C * LF_C * LF_KK; i += PACK_NT) {
        m[1] = fmaxf(m[1], fabsf(w2[i]));
        m[2] = fmaxf(m[2], fabsf(w3[i]));
    }
    for (int i = tid; i < LF_C * LF_KK; i += PACK_NT) m[3] = fmaxf(m[3], fabsf(w4[i]));
#pragma unroll
    for (int q = 0; q < 4; ++q) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) m[q] = fmaxf(m[q], __shfl_xor(m[q], o, 64));
        if ((tid & 63) == 0) red[q][tid >> 6] = m[q];
    }
    __syncthreads();
    float sw[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        float mm = 0.f;
        for (int i = 0; i < PACK_NW; ++i) mm = fmaxf(mm, red[q][i]);
        sw[q] = pow2_scale(mm);
    }
    float* dst = packed + (int64_t)plane * plane_stride + f16_off + (int64_t)orient * LF_ORIENT_FLOATS;
    _Float16* hp = reinterpret_cast<_Float16*>(dst);
    // ---- the composed kernels of this orientation (effective taps): P_block_v2.py:50-55 has no nonlinearity between
    // conv3 and conv4, so away from the image border conv4(conv3(t2) + b3 + r) + b4 =
    //   (w4 o w3) * t2  +  (w4 o w1) * skip  +  b4 + sum_oc (b3 + b1)[oc] * sum_p w4[oc][p]
    // wc[ic][u][v] = sum_oc sum_{p + q = (u, v)} w4[oc][p] w3[oc][ic][q]  (9x9);  wr[u][v] the same with w1 (1 -> 16)
    __shared__ float wc[LF_C * 81];
    __shared__ float wr[81];
    __shared__ float redc[PACK_NW];
    for (int i = tid; i < LF_C * 81 + 81; i += PACK_NT) {
        const bool is_r = i >= LF_C * 81;
        const int ic = is_r ? 0 : i / 81, s = is_r ? i - LF_C * 81 : i % 81;
        const int u = s / 9, vv = s % 9;
        double acc = 0.0;
        for (int oc = 0; oc < (compose ? LF_C : 0); ++oc)
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
    for (int i = tid; i < LF_C * 81; i += PACK_NT) mc = fmaxf(mc, fabsf(wc[i]));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mc = fmaxf(mc, __shfl_xor(mc, o, 64));
    if ((tid & 63) == 0) redc[tid >> 6] = mc;
    __syncthreads();
    float mcc = 0.f;
    for (int i = 0; i < PACK_NW; ++i) mcc = fmaxf(mcc, redc[i]);
    const float swc = pow2_scale(mcc);
    for (int i = tid; i < LF_H_END / 2; i += PACK_NT) {          // one (hi, lo) pair per iteration
        int rem = i;
        const int j = rem % 8; rem /= 8;
        const int lane = rem % 64; rem /= 64;
        const int step = rem;                                 // global k-step index over conv1 | conv2 | conv3 | conv4
        const int row = lane & 15, kg = lane >> 4;
        float v = 0.f;
        if (step == 0) {                                      // conv1: k = 8 kg + j; pairs i = j / 2: i < 3 -> (dy = kg, dx = 2 i + j % 2);
            const int i = j >> 1, e = j & 1;                  // i == 3 -> (dy = 4, dx = 2 kg + j % 2), nothing for kg == 3; dx == 5: zero
            const int dy = i < 3 ? kg : 4, dx = i < 3 ? 2 * i + e : 2 * kg + e;
            if (dx < LF_K && dy < LF_K && !(i == 3 && kg == 3)) v = w1[row * LF_KK + srctap(dy * LF_K + dx, orient)] * sw[0];
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
        reinterpret_cast<__bf16*>(hp + LF_H_BF)[step * 512 + lane * 8 + j] = (__bf16)v;
    }
    float* tail = dst + LF_H_END / 2;
    if (tid < 4) tail[tid] = sw[tid];
    if (tid == 4) tail[4] = swc;
    if (tid < 4) tail[8 + tid] = 1.f / sw[tid];              // exact (powers of two): the kernel multiplies instead of dividing
    if (tid == 4) tail[12] = 1.f / swc;
    if (tid < LF_C) {
        float s4 = 0.f;
        for (int t = 0; t < LF_KK; ++t) s4 += w4[tid * LF_KK + t];
        tail[16 + tid] = s4;
    }
    if (tid < 81) tail[32 + tid] = wr[tid];
    // largest row L1 norms of conv1 and conv2: the backward mode bounds its gradient images with them (|conv(x)| <= max|x| * L1)
    if (tid < 2 * LF_C) {
        const int oc = tid & (LF_C - 1);
        float l1 = 0.f;
        if (tid < LF_C) for (int t = 0; t < LF_KK; ++t) l1 += fabsf(w1[oc * LF_KK + t]);
        else for (int t = 0; t < LF_C * LF_KK; ++t) l1 += fabsf(w2[oc * LF_C * LF_KK + t]);
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) l1 = fmaxf(l1, __shfl_xor(l1, o, 64));      // over the 16 rows
        if (oc == 0) tail[13 + (tid >> 4)] = l1;
    }
    if (tid == 0) {          // the constant of the composed path, in the order the kernel used to sum it per tile
        float cst = b4[0];
        for (int oc = 0; oc < LF_C; ++oc) {
            float s4 = 0.f;
            for (int t = 0; t < LF_KK; ++t) s4 += w4[oc * LF_KK + t];
            cst += (b3[oc] + b1[oc]) * s4;
        }
        tail[5] = cst;
    }
}

struct LfArgs {
    LiftF16Views v;
    const float* src2;    // second view set (images zsplit .. of the launch): same geometry, parameters and row / column
    const float* din2;    // strides as v -- only the bases and the image strides differ
    float* dout2;
    int64_t src2_sz, din2_sz, dout2_sz;
    int64_t zsplit;       // number of images of the first set
    const float* taps;
    const float* packed;
    int64_t pstride;
    int orient_fp32;      // float offset of this orientation's fp32 section (biases)
    int b1, b2, b3, b4;   // float offsets of the biases inside an fp32 orientation section
    int w4;               // float offset of the fp32 conv4 weights [channel][tap] inside an fp32 orientation section
    int f16;              // float offset of this orientation's f16 section
    int batch, h, w, vertical;
    float sign, rw;
    int dbg;              // diagnostics only (lldwt_set_diagnostics flags): bit i set = skip the tile loop of phase P(i+1); results are then wrong
    unsigned long long* stamps;   // diagnostics only (lldwt_set_diagnostics kind 0): [tile][wave][16] s_memtime stamps
    int tiles_x, tiles_y;
    int rl, nseg;                 // run length (tiles a workgroup walks down before it moves to another column) and runs per column
    int nitems;                   // runs of the launch: Z * tiles_x * nseg
    int nborder;                  // the runs of the image's first / last column strip come first in the item order: items 0 .. nborder-1
    int qslot;                    // which of the work-queue counters this launch uses
    // TRAIN: what the backward needs of this step, dense per image of the (single) view set: the source values and the skip
    // signal (Z, h, w), t1, t2 (tanh outputs) and t3 = conv3(t2) + b3 + conv1(skip) + b1 (Z, 16, h, w)
    float* sv_src; float* sv_skip; float* sv_t1; float* sv_t2; float* sv_t3;
    // BWD (backward-data of the block on the same sequential path, packed with the transposed, mirrored weights): the saved tanh
    // outputs t1, t2 (Z, 16, h, w) whose 1 - t^2 gate the gradients; the outputs go to sv_t3 (dt3), sv_t2 (dpre2), sv_t1 (dr),
    // sv_skip (dsk); src = g = dL/dnet as a dense (Z, h, w) tensor, taps = (0, 1, 0)
    const float* gate1; const float* gate2;
    float* mx;                  // BWD: (planes, 2, 64) |.|-max slots of dt3 and dpre2 (atomic max; zeroed by the host), or null
};
// Work queue of the runs.  A workgroup starts with run blockIdx.x and takes every further one from an atomic counter, so the
// workgroups that drew border columns (their tiles cost ~25 % more) simply take fewer runs; the border runs come first in the
// item order (longest processing time first).  A static deal left the CUs 13 % idle at the level-0 shape (in-kernel stamps).
// [slot][0] = runs handed out beyond the first round, [slot][1] = workgroups that are done; the last one to finish zeroes both,
// so a slot is clean again when its launch ends (zero-initialised at module load; a launch takes the next of 64 slots, so
// launches on different streams do not share one).
__device__ int g_lf_queue[64][2];
// in-kernel clock stamps of a diagnostic run (tools/lift_stamps.py); a null pointer (always, outside that tool) skips them
#define LF_STAMP(i)                                                                                                     \
    if (a.stamps && lane == 0)                                                                                          \
        a.stamps[(stamp_tile * NWAVE + wave) * 16 + (i)] =                                                              \
            (i) >= 14 ? __builtin_amdgcn_s_memrealtime() : __builtin_amdgcn_s_memtime();

// 13 k-steps of one 16-pixel tile: B fragments from a T-image (input region width WIN, NIN pixels), A fragments in registers
template <int WIN, int NIN, int PREC = 0>
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
        half8 bl;
        if constexpr (PREC == 0) bl = *reinterpret_cast<const half8*>(img + 2 * NIN * 16 + off);
        acc = mma3<PREC>(ah[ks], al[ks], bh, bl, acc);
    }
    return acc;
}

// the same with the A fragments streamed from memory (L2) k-step by k-step: for the few strip tiles of a border tile, where
// holding all 26 fragments in registers is not worth their 104 VGPRs
template <int WIN, int NIN, int PREC>
__device__ __forceinline__ floatx4 conv16_tile_stream(const uint8_t* __restrict__ img, int basein, bool hi_tap,
                                                      const _Float16* __restrict__ frag, int step0, int lane) {
    floatx4 acc = {0.f, 0.f, 0.f, 0.f};
    // ring of 4: three k-steps of weight fragments in flight (one step ahead exposed an L2 round trip per k-step: 13 of them made
    // the strips of a border tile cost 3.4k cycles for 126 MFMAs)
    half8 ah[4], al[4];
#pragma unroll
    for (int i = 0; i < 3; ++i) ldA<PREC>(frag, step0 + i, lane, ah[i], al[i]);
#pragma unroll
    for (int ks = 0; ks < LF_KS; ++ks) {
        if (ks + 3 < LF_KS) ldA<PREC>(frag, step0 + ks + 3, lane, ah[(ks + 3) & 3], al[(ks + 3) & 3]);
        __builtin_amdgcn_sched_barrier(0);               // the loads stay three steps ahead of their use
        const int ta = 2 * ks, tb = (2 * ks + 1) < LF_KK ? 2 * ks + 1 : LF_KK - 1;
        const int offa = ((ta / LF_K) * WIN + ta % LF_K) * 16, offb = ((tb / LF_K) * WIN + tb % LF_K) * 16;
        const int off = basein + (hi_tap ? offb : offa);
        const half8 bh = *reinterpret_cast<const half8*>(img + off);
        half8 bl;
        if constexpr (PREC == 0) bl = *reinterpret_cast<const half8*>(img + 2 * NIN * 16 + off);
        acc = mma3<PREC>(ah[ks & 3], al[ks & 3], bh, bl, acc);
    }
    return acc;
}

// two tiles at once: the A fragments are shared, the two MFMA chains are independent.  The B fragments of k-step ks+1 are
// read from LDS into a second register set BEFORE the 6 MFMAs of k-step ks (the sched_group_barrier sequence pins that
// order and leaves room for 2 vector instructions of a neighbouring epilogue after every MFMA), so the matrix pipe never
// waits for an LDS round trip.
template <int WIN, int NIN, int PREC, class Piece>
__device__ __forceinline__ void conv16_tile2(const uint8_t* __restrict__ img, int base0, int base1, bool hi_tap,
                                             const half8 (&ah)[LF_KS], const half8 (&al)[LF_KS], floatx4& acc0, floatx4& acc1,
                                             Piece&& piece) {
    acc0 = floatx4{0.f, 0.f, 0.f, 0.f};
    acc1 = floatx4{0.f, 0.f, 0.f, 0.f};
    half8 bh0[2], bl0[2], bh1[2], bl1[2];
    // lanes kg >= 2 take the second tap of the pair: the next pixel (+16 B), the first pixel of the next row where the
    // first tap ends a row (k-steps 2 and 7), the same tap again in the last k-step (tap 25 does not exist: weight 0).
    // Three per-lane bases per tile, every k-step offset an immediate.
    const int dA = hi_tap ? 16 : 0, dW = hi_tap ? (WIN - 4) * 16 : 0;
    const uint8_t* q0[3] = {img + base0 + dA, img + base0 + dW, img + base0};
    const uint8_t* q1[3] = {img + base1 + dA, img + base1 + dW, img + base1};
    auto load = [&](int ks, int set) {
        const int ta = 2 * ks;
        const int offa = ((ta / LF_K) * WIN + ta % LF_K) * 16;
        const int sel = ks == LF_KS - 1 ? 2 : (ta % LF_K == LF_K - 1 ? 1 : 0);
        bh0[set] = *reinterpret_cast<const half8*>(q0[sel] + offa);
        bh1[set] = *reinterpret_cast<const half8*>(q1[sel] + offa);
        if constexpr (PREC == 0) {
            bl0[set] = *reinterpret_cast<const half8*>(q0[sel] + 2 * NIN * 16 + offa);
            bl1[set] = *reinterpret_cast<const half8*>(q1[sel] + 2 * NIN * 16 + offa);
        }
    };
    load(0, 0);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int ks = 0; ks < LF_KS; ++ks) {
        const int c = ks & 1;
        if (ks + 1 < LF_KS) load(ks + 1, c ^ 1);
        if constexpr (PREC == 0) {
            acc0 = mma1<0>(al[ks], bh0[c], acc0);
            acc1 = mma1<0>(al[ks], bh1[c], acc1);
            acc0 = mma1<0>(ah[ks], bl0[c], acc0);
            acc1 = mma1<0>(ah[ks], bl1[c], acc1);
        }
        acc0 = mma1<PREC>(ah[ks], bh0[c], acc0);
        acc1 = mma1<PREC>(ah[ks], bh1[c], acc1);
        piece(ks);       // a slice of the PREVIOUS pair's epilogue (vector ALU, LDS stores): runs in the shadow of the MFMAs
        // this k-step is a scheduling region of its own (fenced): its 4 LDS reads can only be the NEXT step's, one after
        // each of the first four MFMAs; the epilogue slice is spread between the MFMAs (two waves of a SIMD otherwise fall
        // into lockstep -- both in their MFMA stretch, then both in their vector stretch -- and the matrix pipe idles)
        if constexpr (PREC == 0) {
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                if (i < 4) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
                if (i >= 4) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
            }
        } else {               // one product: 2 MFMAs and 2 LDS reads per k-step, the same epilogue slice between them
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 9, 0);
                __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

// one tile, same pipeline (a continuing tile has 5 conv2 tiles per wave: two pairs and this one).  A single accumulation chain
// of this MFMA needs no second accumulator for throughput (MI355X guide); the slice of the previous pair's epilogue rides here
template <int WIN, int NIN, int PREC, class Piece>
__device__ __forceinline__ void conv16_tile1(const uint8_t* __restrict__ img, int base0, bool hi_tap, const half8 (&ah)[LF_KS],
                                             const half8 (&al)[LF_KS], floatx4& acc0, Piece&& piece) {
    acc0 = floatx4{0.f, 0.f, 0.f, 0.f};
    half8 bh0[2], bl0[2];
    const int dA = hi_tap ? 16 : 0, dW = hi_tap ? (WIN - 4) * 16 : 0;
    const uint8_t* q0[3] = {img + base0 + dA, img + base0 + dW, img + base0};
    auto load = [&](int ks, int set) {
        const int ta = 2 * ks;
        const int offa = ((ta / LF_K) * WIN + ta % LF_K) * 16;
        const int sel = ks == LF_KS - 1 ? 2 : (ta % LF_K == LF_K - 1 ? 1 : 0);
        bh0[set] = *reinterpret_cast<const half8*>(q0[sel] + offa);
        if constexpr (PREC == 0) bl0[set] = *reinterpret_cast<const half8*>(q0[sel] + 2 * NIN * 16 + offa);
    };
    load(0, 0);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int ks = 0; ks < LF_KS; ++ks) {
        const int c = ks & 1;
        if (ks + 1 < LF_KS) load(ks + 1, c ^ 1);
        acc0 = mma3<PREC>(ah[ks], al[ks], bh0[c], bl0[c], acc0);
        piece(ks);
        if constexpr (PREC == 0) {
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                if (i < 2) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
                if (i == 2) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
            }
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

// conv1 of one 16-pixel tile: this lane's 8 k values = 4 aligned fp16 pairs of the scaled skip patch (see the header), 3 MFMAs.
// sbase = element index of the tile pixel's top-left tap in the patch; kgoff = {kg * SW, 4 * SW + 2 * min(kg, 2)} (elements)
template <int PREC = 0>
__device__ __forceinline__ floatx4 conv1_tile(const uint8_t* __restrict__ s16, int sbase, int kgoff0, int kgoff1,
                                              const half8& a1h, const half8& a1l) {
    typedef unsigned uintx4 __attribute__((ext_vector_type(4)));
    const int par = sbase & 1;                                        // odd column: the copy shifted by one element
    const uint8_t* c = s16 + par * (2 * NS16 * 2) + (sbase - par) * 2;
    const uint8_t* r0 = c + kgoff0 * 2;
    const uint8_t* r1 = c + kgoff1 * 2;
    uintx4 uh, ul;
    uh[0] = *reinterpret_cast<const unsigned*>(r0);
    uh[1] = *reinterpret_cast<const unsigned*>(r0 + 4);
    uh[2] = *reinterpret_cast<const unsigned*>(r0 + 8);
    uh[3] = *reinterpret_cast<const unsigned*>(r1);
    if constexpr (PREC == 0) {
        ul[0] = *reinterpret_cast<const unsigned*>(r0 + NS16 * 2);
        ul[1] = *reinterpret_cast<const unsigned*>(r0 + NS16 * 2 + 4);
        ul[2] = *reinterpret_cast<const unsigned*>(r0 + NS16 * 2 + 8);
        ul[3] = *reinterpret_cast<const unsigned*>(r1 + NS16 * 2);
    }
    const half8 bh = __builtin_bit_cast(half8, uh), bl = __builtin_bit_cast(half8, ul);
    floatx4 acc = {0.f, 0.f, 0.f, 0.f};
    return mma3<PREC>(a1h, a1l, bh, bl, acc);
}

// store 4 consecutive channels (oc0 .. oc0+3) of pixel p into a T-image of N pixels (one-product modes: the hi image only,
// as fp16 or bf16)
template <int N, int PREC = 0>
__device__ __forceinline__ void timg_store(uint8_t* __restrict__ img, int p, int oc0, const float (&v)[4]) {
    uint8_t* d = img + (oc0 >> 3) * (N * 16) + p * 16 + (oc0 & 7) * 2;
    if constexpr (PREC == 0) {
        half4 hi, lo;
        split4(v, hi, lo);
        *reinterpret_cast<half4*>(d) = hi;
        *reinterpret_cast<half4*>(d + 2 * N * 16) = lo;
    } else if constexpr (PREC == 1) {
        typedef float f2 __attribute__((ext_vector_type(2)));
        typedef _Float16 h2 __attribute__((ext_vector_type(2)));
        const h2 a = __builtin_convertvector((f2){v[0], v[1]}, h2), b = __builtin_convertvector((f2){v[2], v[3]}, h2);
        *reinterpret_cast<half4*>(d) = __builtin_shufflevector(a, b, 0, 1, 2, 3);
    } else {
        typedef float f2 __attribute__((ext_vector_type(2)));
        typedef __bf16 b2 __attribute__((ext_vector_type(2)));
        typedef __bf16 b4 __attribute__((ext_vector_type(4)));
        const b2 a = __builtin_convertvector((f2){v[0], v[1]}, b2), b = __builtin_convertvector((f2){v[2], v[3]}, b2);
        *reinterpret_cast<b4*>(d) = __builtin_shufflevector(a, b, 0, 1, 2, 3);
    }
}

// raw operands of one tile's skip patch (this thread's three patch elements: centre and the two filter neighbours) and its
// dst_in value: fetched for tile n+1 while tile n is still computing
struct LfPre {
    float c[NS / NTH], m[NS / NTH], p[NS / NTH];
    float din;
};

// PERSISTENT: one workgroup per CU (155 KB of LDS) walks over tiles t = blockIdx.x, + gridDim.x, ...  The position of a
// tile inside its image is rotated by the image index, so that a workgroup does not meet the (slower) border tiles of
// every image.
// SEQ: the sequential evaluation of conv3 / conv4 for every tile (debug flag 16), the check of the composed path.
// PREC: 0 = f16x3, 1 = fp16, 2 = bf16 (see mma3)
// TRAIN (with SEQ): the training forward -- the sequential path forms t3 explicitly, and the tile's own 16 x 32 pixels of
// src, skip, t1, t2 and t3 are written out for the backward (what the three fp32 launches k_lift_a/b/c saved)
// BWD (with SEQ): backward-data of the block.  With g = dL/dnet:  dt3 = conv4^T(g),  dpre2 = (1 - t2^2) conv3^T(dt3),
// dr = (1 - t1^2) conv2^T(dpre2) + dt3,  dsk = conv1^T(dr)  -- the forward's sequential chain 1 -> 16 -> 16 -> 16 -> 1 with the
// transposed, mirrored weights (lift_f16_pack_bwd), no biases, the tanh replaced by the gates of the SAVED t2 / t1 (t2's gate is
// staged into the T2 image before conv2' overwrites it pixel by pixel; t1's is loaded per tile in P3), and the gradient images
// scaled by bounds (max|g| of the tile x the row L1 norms of the pack) instead of the fixed 2^14 of tanh outputs.
template <bool SEQ, int PREC, bool TRAIN = false, bool BWD = false>
__global__ __launch_bounds__(NTH) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_lift_fused_f16(LfArgs a) {
    static_assert(!SEQ || PREC == 0, "the sequential check path exists for the fp32-accurate arithmetic only");
    static_assert(!TRAIN || SEQ, "the training forward needs t3: sequential path");
    static_assert(!BWD || (SEQ && !TRAIN), "backward-data runs on the sequential path");
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    float* S = reinterpret_cast<float*>(lds + LDS_S);
    float* RED = reinterpret_cast<float*>(lds + LDS_RED);
    const int tid0 = threadIdx.x;
    const int h = a.h, w = a.w;
    const int tpi = a.tiles_x * a.tiles_y;                     // tiles per image

    // image z of the launch -> its view set and its image index inside that set
    auto view_of_z = [&](int64_t z, int64_t& zz) -> LiftF16Views {          // by value: scalar selects, no stack
        const bool second = z >= a.zsplit;
        zz = second ? z - a.zsplit : z;
        LiftF16Views r = a.v;
        r.src = second ? a.src2 : a.v.src;     r.src_sz = second ? a.src2_sz : a.v.src_sz;
        r.din = second ? a.din2 : a.v.din;     r.din_sz = second ? a.din2_sz : a.v.din_sz;
        r.dout = second ? a.dout2 : a.v.dout;  r.dout_sz = second ? a.dout2_sz : a.v.dout_sz;
        return r;
    };
    // work item = a RUN: rl vertically consecutive tiles of one column strip of one image (the last run of a column may be
    // shorter).  Item order: the runs of the first / last column strip of every image (border tiles), then the others.
    const int bcols = a.tiles_x >= 2 ? 2 : 1;
    auto decode = [&](int item, int j, int64_t& z, int& y0, int& x0, int& seglen) {
        int zi, sx, seg;
        if (item < a.nborder) {
            const int per = bcols * a.nseg;
            zi = item / per;
            const int r = item - zi * per, c = r / a.nseg;
            seg = r - c * a.nseg;
            sx = c ? a.tiles_x - 1 : 0;
        } else {
            const int per = (a.tiles_x - bcols) * a.nseg, it2 = item - a.nborder;
            zi = it2 / per;
            const int r = it2 - zi * per, c = r / a.nseg;
            seg = r - c * a.nseg;
            sx = 1 + c;
        }
        z = zi;
        y0 = (seg * a.rl + j) * TH;
        x0 = sx * TW;
        seglen = min(a.rl, a.tiles_y - seg * a.rl);
    };
    // branch-free: every load goes to a clamped (valid) address; P0 applies the zero padding when it consumes the values
    auto fetch = [&](int item, int j, int tid, LfPre& pr) {
        const int oy_ = tid / TW, ox_ = tid - oy_ * TW;        // this thread's output pixel inside a tile
        int64_t z;
        int y0, x0, seglen_;
        decode(item, j, z, y0, x0, seglen_);
        int64_t zz;
        const LiftF16Views vw = view_of_z(z, zz);
        const float* sp = vw.src + zz * vw.src_sz;
        const int ssy = (int)vw.src_sy, ssx = (int)vw.src_sx;   // per-image offsets fit 32 bits (checked on the host)
        const int dyv = a.vertical ? 1 : 0;
        const int fstep = dyv ? ssy : ssx, flast = dyv ? h - 1 : w - 1;
#pragma unroll
        for (int k = 0; k < NS / NTH; ++k) {
            const int i = tid + k * NTH;
            const int sy = i / SW, sx = i - sy * SW;
            const int gy = min(max(y0 - 8 + sy, 0), h - 1), gx = min(max(x0 - 8 + sx, 0), w - 1);
            // the two filter neighbours = the centre's offset -+ one step along the lifting direction, clamped at the image edge
            const int oc_ = gy * ssy + gx * ssx;
            const int along = dyv ? gy : gx;
            pr.c[k] = sp[oc_];
            pr.m[k] = sp[oc_ - (along > 0 ? fstep : 0)];
            pr.p[k] = sp[oc_ + (along < flast ? fstep : 0)];
        }
        const int gy = min(y0 + oy_, h - 1), gx = min(x0 + ox_, w - 1);
        pr.din = vw.din[zz * vw.din_sz + gy * (int)vw.din_sy + gx * (int)vw.din_sx];
    };
    // the fetched operands wait for their tile in LDS (each thread's own ten floats, written and read back by that thread
    // only: no barrier), not in registers carried around the tile loop (the register file is full in P2)
    // slot v (0..9) of thread tid.  Composed path: 20 KB laid across rows 8..15 of the four T2 arrays (rows 16..23 are handed
    // down in place, rows 0..7 receive the handed-down rows at the start of the next tile); sequential path: behind its D image
    auto stg = [&](int v, int tid) -> float* {
        // half-slots of 256 floats, five per array: which array is wave-uniform (scalar arithmetic), one add per lane
        const int hs = 2 * v + __builtin_amdgcn_readfirstlane(tid >> 8), arr = hs / 5;
        return reinterpret_cast<float*>(lds + LDS_T2 + arr * T2A + T2HALF) + (hs - arr * 5) * 256 + (tid & 255);
    };
    auto stage = [&](int tid, const LfPre& pr) {
#pragma unroll
        for (int k = 0; k < NS / NTH; ++k) {
            *stg(3 * k + 0, tid) = pr.c[k];
            *stg(3 * k + 1, tid) = pr.m[k];
            *stg(3 * k + 2, tid) = pr.p[k];
        }
        *stg(9, tid) = pr.din;
    };

    if ((int)blockIdx.x < a.nitems) {
        LfPre pr0;
        fetch((int)blockIdx.x, 0, tid0, pr0);
        stage(tid0, pr0);
    }
    int item_i = (int)blockIdx.x, run_j = 0;
    float act2_prev = ACT_SCALE;                               // BWD: the operand scale of the T2 rows the tile above handed down
    while (item_i < a.nitems) {
    // every per-lane index below derives from an OPAQUE copy of the thread id: otherwise the compiler hoists all the
    // tile-invariant per-lane address arithmetic of all phases out of the tile loop and spills it
    int tid = tid0;
    asm volatile("" : "+v"(tid));
    const int lane = tid & 63, wave = tid >> 6;
    const int kg = lane >> 4, pl = lane & 15, oc0 = 4 * kg;
    const bool hi_tap = kg >= 2;
    const int halfsel = kg & 1;
    int64_t z;
    int y0, x0, seglen;
    decode(item_i, run_j, z, y0, x0, seglen);
    const bool cont = run_j > 0;                               // the tile above handed down its last T2 rows (composed path: and T1 rows)
    const bool hand_down = run_j + 1 < seglen;
    // TRAIN / BWD: the rows of T1 / T2 this tile writes out.  A tile's own 16 rows are T1 rows 6..21 (T2: 4..19); a continuing tile
    // does not compute its first T1 rows 6, 7 (T2: 4..7) -- the tile above does, as its rows 22, 23 (T2: 20..23), and stores them
    const int st1_lo = cont ? KEEP2 : 6, st1_hi = hand_down ? 6 + TH + 2 : 6 + TH;
    const int st2_lo = cont ? KEEP2 : 4, st2_hi = hand_down ? 4 + TH + 4 : 4 + TH;
    (void)st1_lo; (void)st1_hi; (void)st2_lo; (void)st2_hi;
    int next_item = item_i;
    const int next_j = hand_down ? run_j + 1 : 0;
    int* QN = reinterpret_cast<int*>(lds + LDS_RED) + 16;       // the run this workgroup takes next (broadcast through LDS)
    if (!hand_down && tid == 0) *QN = (int)gridDim.x + atomicAdd(&g_lf_queue[a.qslot][0], 1);
    const int64_t stamp_tile = (z * a.tiles_y + y0 / TH) * a.tiles_x + x0 / TW;
    int64_t zv;
    const LiftF16Views vout = view_of_z(z, zv);
    const int plane = (int)(zv / a.batch);
    const float* pk = a.packed + (int64_t)plane * a.pstride;
    const float* bias = pk + a.orient_fp32;
    const _Float16* frag = reinterpret_cast<const _Float16*>(pk + a.f16);
    const float* scales = pk + a.f16 + LF_H_END / 2;
    const float sw1 = scales[0], sw2 = scales[1], sw3 = scales[2], sw4 = scales[3];
    const float isw1 = scales[8], isw2 = scales[9], isw3 = scales[10];           // their reciprocals (exact powers of two)
    LfPre pre;
#pragma unroll
    for (int k = 0; k < NS / NTH; ++k) {
        pre.c[k] = *stg(3 * k + 0, tid);
        pre.m[k] = *stg(3 * k + 1, tid);
        pre.p[k] = *stg(3 * k + 2, tid);
    }
    const float din_pre = *stg(9, tid);
    // the rows handed down by the tile above: T1 rows 16..27 -> 0..11, T2 rows 16..23 -> 0..7 (source and destination rows are
    // disjoint, the destinations are dead since the barrier that ended the tile above; P1 / P2 read them two barriers later)
    if (cont) {
        typedef unsigned uintx4_t __attribute__((ext_vector_type(4)));
        constexpr int NARR = PREC == 0 ? 4 : 2;                                       // one product: the two hi arrays only
        constexpr int MV1 = NARR * KEEP1 * R1W, MV2 = NARR * KEEP2 * R2W;             // 16-byte pieces: 2112, 1280
        if constexpr (!SEQ) {
#pragma unroll
            for (int k = 0; k < (MV1 + NTH - 1) / NTH; ++k) {
                const int c = tid + k * NTH, arr = c / (KEEP1 * R1W), px = c - arr * (KEEP1 * R1W);
                if (c < MV1)
                    *reinterpret_cast<uintx4_t*>(lds + LDS_T1 + arr * T1A + px * 16) =
                        *reinterpret_cast<const uintx4_t*>(lds + LDS_T1 + arr * T1A + ((N1 - KEEP1 * R1W) + px) * 16);
            }
        }
#pragma unroll
        for (int k = 0; k < (MV2 + NTH - 1) / NTH; ++k) {
            const int c = tid + k * NTH, arr = c / (KEEP2 * R2W), px = c - arr * (KEEP2 * R2W);
            if (c < MV2)
                *reinterpret_cast<uintx4_t*>(lds + LDS_T2 + arr * T2A + px * 16) =
                    *reinterpret_cast<const uintx4_t*>(lds + LDS_T2 + arr * T2A + ((N2 - KEEP2 * R2W) + px) * 16);
        }
    }
    LF_STAMP(0)
    LF_STAMP(14)
    if (a.stamps && lane == 0)          // slot 13: which CU (XCC_ID << 32 | HW_ID)
        a.stamps[(stamp_tile * NWAVE + wave) * 16 + 13] =
            ((unsigned long long)__builtin_amdgcn_s_getreg(0xF814) << 32) | (unsigned)__builtin_amdgcn_s_getreg(0xF804);
    // ---------------- P0: skip patch (32 x 48) from the prefetched operands, and its |max|
    float sv[NS / NTH];
    {
        const float t0 = a.taps[plane * 3 + 0], t1 = a.taps[plane * 3 + 1], t2 = a.taps[plane * 3 + 2];
        float amax = 0.f;
        const int dyv = a.vertical ? 1 : 0, dxv = 1 - dyv;
#pragma unroll
        for (int k = 0; k < NS / NTH; ++k) {
            const int i = tid + k * NTH;
            const int sy = i / SW, sx = i - sy * SW;
            const int gy = y0 - 8 + sy, gx = x0 - 8 + sx;
            const bool in = gy >= 0 && gy < h && gx >= 0 && gx < w;
            const float m = (gy - dyv >= 0 && gx - dxv >= 0) ? pre.m[k] : 0.f;          // zero padding of the 3-tap filter
            const float p = (gy + dyv < h && gx + dxv < w) ? pre.p[k] : 0.f;
            const float v = in ? t0 * m + t1 * pre.c[k] + t2 * p : 0.f;
            S[i] = v;
            sv[k] = v;
            if constexpr (TRAIN) {
                if (in && sy >= 8 && sy < 8 + TH && sx >= 8 && sx < 8 + TW) {
                    const int64_t e = zv * (int64_t)h * w + (int64_t)gy * w + gx;
                    a.sv_skip[e] = v;
                    a.sv_src[e] = pre.c[k];
                }
            }
            amax = fmaxf(amax, fabsf(v));
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o, 64));
        if (lane == 0) RED[wave] = amax;
    }
    LF_STAMP(1)
    __syncthreads();
    LF_STAMP(2)
    if (!hand_down) next_item = __builtin_amdgcn_readfirstlane(*QN);      // wave-uniform: everything derived from it stays scalar
    float s_skip, act1 = ACT_SCALE, act2 = ACT_SCALE;       // operand scales of the T1 / T2 images
    {
        float m = RED[0];
#pragma unroll
        for (int i = 1; i < NWAVE; ++i) m = fmaxf(m, RED[i]);
        s_skip = pow2_scale(m);
        if constexpr (BWD) {                                  // |dt3| <= max|g| L1(conv1'), |dpre2| <= that x L1(conv2') (gates <= 1)
            act1 = pow2_scale(m * scales[13]);
            act2 = pow2_scale(m * scales[13] * scales[14]);
        }
    }
    if constexpr (BWD) {
        // the handed-down dpre2 rows carry the scale of the tile above (bound-based, per tile): bring them to this tile's.  Each
        // thread rescales the pieces it moved itself (rows 0..7, nobody else touches them before P2); both scales are powers of two
        // and the values obey this tile's bound too (they depend on g rows y0 - 8 .. y0 + 7 only, all inside this tile's patch)
        if (cont && act2 != act2_prev) {
            const float ratio = act2 * __int_as_float(0x7F000000 - __float_as_int(act2_prev));
            constexpr int MV2 = 4 * KEEP2 * R2W;
#pragma unroll
            for (int k = 0; k < (MV2 + NTH - 1) / NTH; ++k) {
                const int c = tid + k * NTH, arr = c / (KEEP2 * R2W), px = c - arr * (KEEP2 * R2W);
                if (c < MV2) {
                    half8* q = reinterpret_cast<half8*>(lds + LDS_T2 + arr * T2A + px * 16);
                    half8 v = *q;
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] = (_Float16)((float)v[j] * ratio);
                    *q = v;
                }
            }
        }
        act2_prev = act2;
    }
    // s_skip = 2^k (|k| <= 120): its reciprocal by exponent arithmetic, exact, instead of a 10-instruction division
    const float inv1 = __int_as_float(0x7F000000 - __float_as_int(s_skip)) * isw1;
    {   // the scaled split-fp16 images of the patch (this thread's own three values, still in registers)
        _Float16* h0 = reinterpret_cast<_Float16*>(lds + LDS_S16);             // copy 0: hi, lo; copy 1: hi, lo
#pragma unroll
        for (int k = 0; k < NS / NTH; ++k) {
            const int i = tid + k * NTH;
            const float v = sv[k] * s_skip;
            if constexpr (PREC == 0) {
                const _Float16 hi = (_Float16)v;
                const _Float16 lo = (_Float16)(v - (float)hi);
                h0[i] = hi;
                h0[NS16 + i] = lo;
                if (i > 0) {
                    h0[2 * NS16 + i - 1] = hi;
                    h0[3 * NS16 + i - 1] = lo;
                }
            } else {                                        // one product: the hi image only, fp16 or bf16 bits
                const unsigned short hb = to16<PREC>(v);
                unsigned short* u0 = reinterpret_cast<unsigned short*>(h0);
                u0[i] = hb;
                if (i > 0) u0[2 * NS16 + i - 1] = hb;
            }
        }
        if (tid < 4 * 49) {                                                   // tails: finite values against zero weights
            const int arr = tid / 49, j = tid - arr * 49;
            if (arr >= 2 || j > 0) h0[arr * NS16 + NS - 1 + j] = (_Float16)0.f;
        }
    }
    __syncthreads();

    // conv1 operands: element offsets of this lane group's pairs inside the patch (conv1_tile)
    const uint8_t* s16 = lds + LDS_S16;
    const int kgoff0 = kg * SW, kgoff1 = 4 * SW + 2 * (kg < 2 ? kg : 2);
    half8 a1h, a1l;
    ldA<PREC>(frag, LF_H_C1 / LF_FRAG, lane, a1h, a1l);
    float b1v[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) b1v[q] = bias[a.b1 + oc0 + q];
    float b1c[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) b1c[q] = b1v[q] * TWO_LOG2E;
    const float inv1c = inv1 * TWO_LOG2E;

    // ---------------- BWD: the gate of conv2' (1 - t2^2 on the 24 x 40 region, x 2^14, split like any T-image value) into the T2
    // image: the conv2' epilogue reads the gate of a pixel from the very slot it then overwrites with dpre2.  Outside the image
    // the value is irrelevant (those pixels are masked to 0).  (pixel, group of 4 channels) items, 3 840 of them.
    if constexpr (BWD) {
        for (int i = (cont ? KEEP2 * R2W * 4 : 0) + tid; i < N2 * 4; i += NTH) {      // a continuing tile: rows 8..23 (0..7 hold dpre2)
            const int p = i >> 2, c4 = (i & 3) * 4;
            const int r = p / R2W, c = p - r * R2W;
            const int gy = min(max(y0 - 4 + r, 0), h - 1), gx = min(max(x0 - 4 + c, 0), w - 1);
            const float* tp = a.gate2 + (zv * LF_C + c4) * (int64_t)h * w + (int64_t)gy * w + gx;
            float gq[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float t = tp[(int64_t)q * h * w];
                gq[q] = __builtin_fmaf(-t, t, 1.f) * ACT_SCALE;
            }
            timg_store<N2, 0>(lds + LDS_T2, p, c4, gq);
        }
    }
    // ---------------- P1: t1 = tanh(conv1(skip) + b1) on 28 x 44.  Two tiles per iteration (independent chains for the
    // scheduler: one tile alone is a latency chain LDS -> MFMA x3 -> tanh -> split -> store); tiles past the end repeat the
    // last one (same bytes stored again)
    if (!(a.dbg & 1)) {
        // a continuing tile computes rows 12..27 only (tiles 33..76)
        // (sequential path: rows 8..27, tiles 22..76 -- T1 itself is not handed down there)
        const int nt1 = cont ? (SEQ ? NT1S : NT1C) : NT1, tb1 = NT1 - nt1;
        const int nit1 = (nt1 + 2 * NWAVE - 1) / (2 * NWAVE);           // 5, or 3 (sequential path: 4)

        float mx1 = 0.f;                        // BWD: max |dt3| x act1 over this wave's values (for the weight gradient's dY scale)
#pragma unroll 1
        for (int it = 0; it < nit1; ++it) {
            floatx4 acc[2];
            int pq[2];
            float msk[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int tile = wave + (2 * it + t) * NWAVE;
                const int p = (tb1 + (tile < nt1 ? tile : nt1 - 1)) * 16 + pl;
                const int r = p / R1W, c = p - r * R1W;
                acc[t] = conv1_tile<PREC>(s16, r * SW + c, kgoff0, kgoff1, a1h, a1l);
                const int gy = y0 - 6 + r, gx = x0 - 6 + c;
                msk[t] = (gy >= 0 && gy < h && gx >= 0 && gx < w) ? act1 : 0.f;
                pq[t] = p;
            }
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                float v[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    if constexpr (BWD) {          // dt3 = conv4^T(g): no bias, no tanh
                        v[q] = acc[t][q] * inv1 * msk[t];
                        mx1 = fmaxf(mx1, fabsf(v[q]));
                    }
                    else v[q] = tanh_scaled_masked(acc[t][q], inv1c, b1c[q], msk[t], -2.f * msk[t]);
                }
                timg_store<N1, PREC>(lds + LDS_T1, pq[t], oc0, v);
                if constexpr (BWD) {
                    const int r = pq[t] / R1W, c = pq[t] - r * R1W;
                    const int gy = y0 + r - 6, gx = x0 + c - 6;
                    if (r >= st1_lo && r < st1_hi && c >= 6 && c < 6 + TW && gy < h && gx < w) {
                        float* d = a.sv_t3 + (zv * LF_C + oc0) * (int64_t)h * w + (int64_t)gy * w + gx;
#pragma unroll
                        for (int q = 0; q < 4; ++q) d[(int64_t)q * h * w] = acc[t][q] * inv1;
                    }
                }
                if constexpr (TRAIN) {
                    const int r = pq[t] / R1W, c = pq[t] - r * R1W;
                    const int gy = y0 + r - 6, gx = x0 + c - 6;
                    if (r >= st1_lo && r < st1_hi && c >= 6 && c < 6 + TW && gy < h && gx < w) {
                        float* d = a.sv_t1 + (zv * LF_C + oc0) * (int64_t)h * w + (int64_t)gy * w + gx;
#pragma unroll
                        for (int q = 0; q < 4; ++q) d[(int64_t)q * h * w] = v[q] * (1.f / ACT_SCALE);
                    }
                }
            }
        }
        if constexpr (BWD) {
            if (a.mx) {
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) mx1 = fmaxf(mx1, __shfl_xor(mx1, o, 64));
                mx1 *= __int_as_float(0x7F000000 - __float_as_int(act1));
                if (lane == 0 && mx1 > 0.f)     // >= 0: integer order == float order
                    atomicMax(reinterpret_cast<int*>(a.mx) + plane * 128 + ((item_i + wave) & 63), __float_as_int(mx1));
            }
        }
    }
    LF_STAMP(3)
    __syncthreads();
    LF_STAMP(4)

    // ---------------- P2: t2 = tanh(conv2(t1) + b2) on 24 x 40
    {
        half8 ah[LF_KS], al[LF_KS];
#pragma unroll
        for (int ks = 0; ks < LF_KS; ++ks) ldA<PREC>(frag, LF_H_C2 / LF_FRAG + ks, lane, ah[ks], al[ks]);
        float bv[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) bv[q] = bias[a.b2 + oc0 + q];
        // the T1 image's operand scale: 2^14 for tanh outputs, the bound-based power of two of the BWD mode (exact reciprocal)
        const float inv2 = (BWD ? __int_as_float(0x7F000000 - __float_as_int(act1)) : 1.f / ACT_SCALE) * isw2;
        const float inv_act2 = BWD ? __int_as_float(0x7F000000 - __float_as_int(act2)) : 1.f / ACT_SCALE;
        float bvc[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) bvc[q] = bv[q] * TWO_LOG2E;
        const float inv2c = inv2 * TWO_LOG2E;
        // software pipeline over PAIRS of tiles (wave + 16 it, wave + 16 it + 8): the MFMA chains of pair it+1 are issued
        // next to the tanh / split / store work of pair it, so the vector ALU and the matrix pipe overlap inside one wave
        // a continuing tile computes rows 8..23 only (tiles 20..59): 5 tiles per wave = two pairs and a single one; a first
        // tile 60 = four pairs (the last one half empty for waves 4..7)
        static_assert(NT2C == 5 * NWAVE, "a continuing tile: two pairs and one single conv2 tile per wave");
        const int nt2 = cont ? NT2C : NT2, tb2 = NT2 - nt2;
        const int nit2 = cont ? 2 : (NT2 + 2 * NWAVE - 1) / (2 * NWAVE);
        auto tile_base = [&](int tile, int& p_out) {
            const int tcl = tb2 + (tile < nt2 ? tile : nt2 - 1);       // past the end: the last tile again (same bytes stored twice)
            const int p = tcl * 16 + pl;
            const int r = p / R2W, c = p - r * R2W;
            p_out = p;
            return (r * R1W + c) * 16 + halfsel * (N1 * 16);
        };
        // the epilogue of a pair (tiles at pixels pp[0], pp[1], accumulators pc[0], pc[1]) in 10 slices: one value per slice
        // (scale + bias, tanh, mask), then split + store of each tile
        floatx4 pc[2];
        int pp[2];
        float pin[2];                          // the image's operand scale inside the image, 0 outside (a factor, not a branch)
        bool pdup[2] = {false, false};         // BWD: a tile past the end (repeated last tile) must not touch the image again
        float ev[8];
        float mx2 = 0.f;                       // BWD: max |dpre2| x act2 over this wave's values
        auto in_image = [&](int p) {
            const int r = p / R2W, c = p - r * R2W;
            const int gy = y0 - 4 + r, gx = x0 - 4 + c;
            return (gy >= 0 && gy < h && gx >= 0 && gx < w) ? act2 : 0.f;
        };
        auto slice = [&](int ks) {
            if (ks < 8) {
                const int t = ks >> 2, q = ks & 3;
                if constexpr (BWD) {           // dpre2 = (1 - t2^2) conv3^T(dt3): the gate waits in this value's own T2 slot
                    const uint8_t* gp = lds + LDS_T2 + ((oc0 + q) >> 3) * (N2 * 16) + pp[t] * 16 + ((oc0 + q) & 7) * 2;
                    const float gate = ((float)*reinterpret_cast<const _Float16*>(gp) +
                                        (float)*reinterpret_cast<const _Float16*>(gp + 2 * N2 * 16)) * (1.f / ACT_SCALE);
                    ev[ks] = pc[t][q] * inv2 * gate * pin[t];
                    mx2 = fmaxf(mx2, fabsf(ev[ks]));
                } else {
                    ev[ks] = tanh_scaled_masked(pc[t][q], inv2c, bvc[q], pin[t], -2.f * pin[t]);
                }
            } else if (ks < 10) {
                const int t = ks - 8;
                const float v[4] = {ev[4 * t], ev[4 * t + 1], ev[4 * t + 2], ev[4 * t + 3]};
                if (!BWD || !pdup[t]) timg_store<N2, PREC>(lds + LDS_T2, pp[t], oc0, v);
                if constexpr (BWD) {
                    const int r = pp[t] / R2W, c = pp[t] - r * R2W;
                    const int gy = y0 + r - 4, gx = x0 + c - 4;
                    if (!pdup[t] && r >= st2_lo && r < st2_hi && c >= 4 && c < 4 + TW && gy < h && gx < w) {
                        float* d = a.sv_t2 + (zv * LF_C + oc0) * (int64_t)h * w + (int64_t)gy * w + gx;
#pragma unroll
                        for (int q = 0; q < 4; ++q) d[(int64_t)q * h * w] = v[q] * inv_act2;
                    }
                }
                if constexpr (TRAIN) {
                    const int r = pp[t] / R2W, c = pp[t] - r * R2W;
                    const int gy = y0 + r - 4, gx = x0 + c - 4;
                    if (r >= st2_lo && r < st2_hi && c >= 4 && c < 4 + TW && gy < h && gx < w) {
                        float* d = a.sv_t2 + (zv * LF_C + oc0) * (int64_t)h * w + (int64_t)gy * w + gx;
#pragma unroll
                        for (int q = 0; q < 4; ++q) d[(int64_t)q * h * w] = v[q] * (1.f / ACT_SCALE);
                    }
                }
            }
        };
        if (!(a.dbg & 2)) {
            {
                const int b0 = tile_base(wave, pp[0]), b1 = tile_base(wave + NWAVE, pp[1]);
                conv16_tile2<R1W, N1, PREC>(lds + LDS_T1, b0, b1, hi_tap, ah, al, pc[0], pc[1], [](int) {});
                pin[0] = in_image(pp[0]);
                pin[1] = in_image(pp[1]);
                pdup[0] = wave >= nt2;
                pdup[1] = wave + NWAVE >= nt2;
            }
#pragma unroll 1
            for (int it = 1; it < nit2; ++it) {
                floatx4 n0, n1;
                int q0, q1;
                const int b0 = tile_base(wave + 16 * it, q0), b1 = tile_base(wave + 16 * it + NWAVE, q1);
                conv16_tile2<R1W, N1, PREC>(lds + LDS_T1, b0, b1, hi_tap, ah, al, n0, n1, slice);
                pc[0] = n0; pc[1] = n1; pp[0] = q0; pp[1] = q1;
                pin[0] = in_image(q0);
                pin[1] = in_image(q1);
                pdup[0] = wave + 16 * it >= nt2;
                pdup[1] = wave + 16 * it + NWAVE >= nt2;
            }
            if (cont) {
                floatx4 n0;
                int q0;
                const int b0 = tile_base(wave + 4 * NWAVE, q0);
                conv16_tile1<R1W, N1, PREC>(lds + LDS_T1, b0, hi_tap, ah, al, n0, slice);
                if constexpr (SEQ) {            // TRAIN / BWD: the single tile's epilogue is a pair's first half (gates, saved tensors, maxima)
                    pc[0] = n0; pp[0] = q0; pin[0] = in_image(q0); pdup[0] = false;
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks) slice(ks);
                    slice(8);
                } else {
                    const float pin0 = in_image(q0);
                    float v[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) v[q] = tanh_scaled_masked(n0[q], inv2c, bvc[q], pin0, -2.f * pin0);
                    timg_store<N2, PREC>(lds + LDS_T2, q0, oc0, v);
                }
            } else {
#pragma unroll
                for (int ks = 0; ks < 10; ++ks) slice(ks);
            }
        }
        if constexpr (BWD) {
            if (a.mx) {
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) mx2 = fmaxf(mx2, __shfl_xor(mx2, o, 64));
                mx2 *= inv_act2;
                if (lane == 0 && mx2 > 0.f)
                    atomicMax(reinterpret_cast<int*>(a.mx) + plane * 128 + 64 + ((item_i + wave) & 63), __float_as_int(mx2));
            }
        }
    }
    LF_STAMP(5)
    __syncthreads();
    LF_STAMP(6)

    // ---------------- interior tiles: conv4(conv3(t2) + b3 + r) + b4 through the COMPOSED 9x9 kernels (packed by
    // k_lift_f16_pack): exact algebra wherever the t3 region (20 x 36 around the tile) lies inside the image, because
    // the only thing between conv3 and conv4 is the zero padding at the image border.  5 k-steps x 3 MFMAs per 16 pixels on
    // 16 x 40 pixels instead of 13 x 3 on 20 x 36 plus conv4, no t3 image, no second pass over conv1.
    // BORDER tiles take the same path plus a correction: the composed kernel also sums conv4 taps that fall OUTSIDE the
    // image, where the true t3 is conv4's zero padding but the composition sees t3v = conv3(t2) + b3 + r evaluated there
    // (t2 and the skip patch are zero outside the image, so t3v is well defined).  Those positions are at most two rows /
    // columns beyond each image edge: t3v is evaluated on these strips only (<= 14 MFMA tiles instead of the 45 of a full
    // t3 region), kept in fp32, and  sum_{taps outside} w4 . t3v  is subtracted per output pixel.  LLDWT_LF_DBG bit 16
    // selects the sequential evaluation (t3 on 20 x 36, conv4, below) for every tile instead: the check of this algebra.
    const bool interior = y0 >= 2 && y0 + TH + 2 <= h && x0 >= 2 && x0 + TW + 2 <= w;
    // strips in t3-region coordinates (20 x 36, origin (y0 - 2, x0 - 2)): rows / columns whose image coordinate is
    // -2, -1 or h, h + 1 (w, w + 1)
    int nR = 0, nC = 0, Rl0 = 0, Rl1 = 0, Rl2 = 0, Rl3 = 0, Cl0 = 0, Cl1 = 0, Cl2 = 0, Cl3 = 0;   // scalars, not arrays: no stack
    if (!SEQ && !interior) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int gy = k < 2 ? k - 2 : h + k - 2, gx = k < 2 ? k - 2 : w + k - 2;
            const int r = gy - (y0 - 2), c = gx - (x0 - 2);
            if (r >= 0 && r < TH + 4) { if (nR == 0) Rl0 = r; else if (nR == 1) Rl1 = r; else if (nR == 2) Rl2 = r; else Rl3 = r; ++nR; }
            if (c >= 0 && c < TW + 4) { if (nC == 0) Cl0 = c; else if (nC == 1) Cl1 = c; else if (nC == 2) Cl2 = c; else Cl3 = c; ++nC; }
        }
    }
    const int nF = nR * R3W + nC * (TH + 4);
    if constexpr (!SEQ) {
        constexpr int NPC = TH * R2W, NTC = NPC / 16;           // 640 pixels (16 rows x 40 T2 columns), 40 tiles
        constexpr int DP = 12;                                  // floats per pixel in the D image (9 used)
        static_assert(NPC % 16 == 0 && NPC == 4 * DPIECE_PX && DPIECE_PX % 16 == 0, "the D image is four quarters of whole tiles");
        const float* tail = pk + a.f16 + LF_H_END / 2;
        const float swc = tail[4];
        // rows 0..15 of the four T1 arrays are dead after P2 (rows 16..27 are handed down): D quarter k = output rows 4k..4k+3
        auto Dp = [&](int p) -> float* {                        // pixel p (16 rows x 40 T2 columns) of the D image
            const int k = p / DPIECE_PX;
            return reinterpret_cast<float*>(lds + LDS_T1 + k * T1A) + (p - k * DPIECE_PX) * DP;
        };
        auto T3Vp = [&](int j) -> float* {                      // [strip position][16 channels] fp32, behind the D quarters
            const int k = j / T3V_PER;
            return reinterpret_cast<float*>(lds + LDS_T1 + k * T1A + DPIECE_BYTES) + (j - k * T3V_PER) * LF_C;
        };
        float* W4L = reinterpret_cast<float*>(lds + LDS_W4L);   // conv4's fp32 weights as [tap][channel]
        const bool more = next_item < a.nitems;
        LfPre nxt;
        if (!interior) {                                        // before the composite loop: what it needs from P1 dies here
            if (tid < LF_C * LF_KK) W4L[(tid % LF_KK) * LF_C + tid / LF_KK] = bias[a.w4 + tid];
            half8 c1h, c1l;
            ldA<PREC>(frag, LF_H_C1 / LF_FRAG, lane, c1h, c1l);
            float bv[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) bv[q] = bias[a.b3 + oc0 + q] + bias[a.b1 + oc0 + q];
            const float inv3 = (1.f / ACT_SCALE) * isw3;
            for (int tile = wave; tile * 16 < nF; tile += NWAVE) {
                const int j = tile * 16 + pl, jc = j < nF ? j : nF - 1;
                int r, c;
                if (jc < nR * R3W) {
                    const int ir = jc / R3W;
                    c = jc - ir * R3W;
                    r = ir == 0 ? Rl0 : ir == 1 ? Rl1 : ir == 2 ? Rl2 : Rl3;
                } else {
                    const int q = jc - nR * R3W, ic = q / (TH + 4);
                    r = q - ic * (TH + 4);
                    c = ic == 0 ? Cl0 : ic == 1 ? Cl1 : ic == 2 ? Cl2 : Cl3;
                }
                const floatx4 acc = conv16_tile_stream<R2W, N2, PREC>(lds + LDS_T2, (r * R2W + c) * 16 + halfsel * (N2 * 16), hi_tap,
                                                                      frag, LF_H_C3 / LF_FRAG, lane);
                const floatx4 accr = conv1_tile<PREC>(s16, (r + 4) * SW + c + 4, kgoff0, kgoff1, c1h, c1l);
                floatx4 v;
#pragma unroll
                for (int q = 0; q < 4; ++q) v[q] = acc[q] * inv3 + accr[q] * inv1 + bv[q];
                if (j < nF) *reinterpret_cast<floatx4*>(T3Vp(j) + oc0) = v;
            }
        }
        float rs = 0.f;
        {
            half8 ah[LF_KSC], al[LF_KSC];
#pragma unroll
            for (int ks = 0; ks < LF_KSC; ++ks) ldA<PREC>(frag, LF_H_CC / LF_FRAG + ks, lane, ah[ks], al[ks]);
            // operands of the NEXT tile of this workgroup: issued here, in flight during the composite loop (~2 us of matrix
            // work, an HBM round trip under load), staged in LDS at the tile's end (issued behind the loop they were waited for
            // at full latency).  AFTER the weight fragments: vmcnt retires in order, so a wait for a younger L2 hit would
            // otherwise wait for these HBM loads too
            __builtin_amdgcn_sched_barrier(0);
            if (more) fetch(next_item, next_j, tid, nxt);
            __builtin_amdgcn_sched_barrier(0);
            const uint8_t* img = lds + LDS_T2;
            static_assert(NTC == 5 * NWAVE, "five composite tiles per wave");
            // (w4 o w1) * skip, 81 fp32 taps on the skip patch for this thread's output pixel: vector work with no MFMA of
            // its own -- a fifth of it rides in the shadow of each composite tile's 15 MFMAs
            const int oyc = tid / TW, oxc = tid - oyc * TW;
            const float* Sc = S + (oyc + 4) * SW + oxc + 4;
            // the B fragments of tile it + 1 (5 k-steps, hi and lo) are read from LDS while the 15 MFMAs of tile it run: a tile's own
            // reads sat right in front of its MFMAs before, and the loop was bound by LDS latency (7.4k cycles for 2.4k of matrix work)
            half8 cbh[2][LF_KSC], cbl[2][LF_KSC];
            auto cload = [&](int it, int set) {
                const int p = (wave + it * NWAVE) * 16 + pl;
                const int r = p / R2W, c = p - r * R2W;         // D(r, c) <-> output row r, T2 column c; tap dy -> T2 row r + dy
                const int basein = (r * R2W + c) * 16 + halfsel * (N2 * 16);
#pragma unroll
                for (int ks = 0; ks < LF_KSC; ++ks) {
                    const int dya = 2 * ks, dyb = 2 * ks + 1 < 9 ? 2 * ks + 1 : 8;            // dy 9 does not exist: weight 0
                    const int off = basein + (hi_tap ? dyb : dya) * (R2W * 16);
                    cbh[set][ks] = *reinterpret_cast<const half8*>(img + off);
                    if constexpr (PREC == 0) cbl[set][ks] = *reinterpret_cast<const half8*>(img + 2 * N2 * 16 + off);
                }
            };
            cload(0, 0);
#pragma unroll
            for (int it = 0; it < 5; ++it) {
                const int p = (wave + it * NWAVE) * 16 + pl;
                if (it + 1 < 5) cload(it + 1, (it + 1) & 1);
                floatx4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < LF_KSC; ++ks) acc = mma3<PREC>(ah[ks], al[ks], cbh[it & 1][ks], cbl[it & 1][ks], acc);
                // rows = dx: lanes kg 0 hold dx 0-3, kg 1 dx 4-7, kg 2 dx 8 (register 0)
                float* dq = Dp(p);
                if (kg < 2) *reinterpret_cast<floatx4*>(dq + 4 * kg) = acc;
                if (kg == 2) dq[8] = acc[0];
#pragma unroll
                for (int j = 17 * it; j < 17 * it + 17 && j < 81; ++j) rs = __builtin_fmaf(tail[32 + j], Sc[(j / 9) * SW + j % 9], rs);
#pragma unroll
                for (int i = 0; i < (PREC == 0 ? 15 : 5); ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, PREC == 0 ? 2 : 5, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, PREC == 0 ? 2 : 5, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        LF_STAMP(7)
        __syncthreads();
        LF_STAMP(8)
        // ---- border tiles: CORR[a] = sum over the conv4 taps of affected pixel a that land outside the image of
        // sum_oc w4[oc][tap] * t3v[position][oc].  Affected pixels (within two of an image edge): the tile's affected rows RA in
        // full, then the affected columns CA of the other rows.  Half a wave per pixel, one lane per tap (25 of 32 lanes), the 32
        // partial sums reduced by a fixed shuffle tree: every lane takes part, and the summation order does not depend on scheduling.
        int nRA = 0, nCA = 0, RA0 = 0, RA1 = 0, RA2 = 0, RA3 = 0, CA0 = 0, CA1 = 0, CA2 = 0, CA3 = 0, rlo = 0;
        float* CORR = reinterpret_cast<float*>(lds + LDS_T2);   // rows 0..7 of T2 array 0 are dead after the composite loop (5 120 B)
        if (!interior) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {                       // image rows / columns 0, 1, h-2, h-1 (w-2, w-1) that lie in this tile
                const int gyk = k < 2 ? k : h - 4 + k, gxk = k < 2 ? k : w - 4 + k;
                const int r = gyk - y0, c = gxk - x0;
                const bool newr = r >= 0 && r < TH && !(nRA > 0 && r == RA0) && !(nRA > 1 && r == RA1) && !(nRA > 2 && r == RA2);
                const bool newc = c >= 0 && c < TW && !(nCA > 0 && c == CA0) && !(nCA > 1 && c == CA1) && !(nCA > 2 && c == CA2);
                if (newr) { if (nRA == 0) RA0 = r; else if (nRA == 1) RA1 = r; else if (nRA == 2) RA2 = r; else RA3 = r; ++nRA; }
                if (newc) { if (nCA == 0) CA0 = c; else if (nCA == 1) CA1 = c; else if (nCA == 2) CA2 = c; else CA3 = c; ++nCA; }
            }
            // the rows not in RA are one contiguous range (RA = the tile's first and / or last rows)
            rlo = 0;
            while (rlo < TH && ((nRA > 0 && rlo == RA0) || (nRA > 1 && rlo == RA1) || (nRA > 2 && rlo == RA2) || (nRA > 3 && rlo == RA3))) ++rlo;
            const int nA = nRA * TW + (TH - nRA) * nCA;
            const int half = lane >> 5, tap = lane & 31;
            const int dy = tap / LF_K, dx = tap - dy * LF_K;
            for (int a0 = wave * 2; a0 < nA; a0 += 2 * NWAVE) {
                const int ai = a0 + half;
                float u = 0.f;
                if (ai < nA && tap < LF_KK) {
                    int oy, ox;
                    if (ai < nRA * TW) {
                        const int ir = ai / TW;
                        ox = ai - ir * TW;
                        oy = ir == 0 ? RA0 : ir == 1 ? RA1 : ir == 2 ? RA2 : RA3;
                    } else {
                        const int b = ai - nRA * TW, q = b / nCA, ic = b - q * nCA;
                        oy = rlo + q;
                        ox = ic == 0 ? CA0 : ic == 1 ? CA1 : ic == 2 ? CA2 : CA3;
                    }
                    const int r = oy + dy, c = ox + dx;                  // t3-region coordinates of this tap's position
                    const int gyy = y0 + oy - 2 + dy, gxx = x0 + ox - 2 + dx;
                    const bool rowout = gyy < 0 || gyy >= h, colout = gxx < 0 || gxx >= w;
                    if (rowout || colout) {
                        const int ir = r == Rl0 ? 0 : r == Rl1 ? 1 : r == Rl2 ? 2 : 3;
                        const int ic = c == Cl0 ? 0 : c == Cl1 ? 1 : c == Cl2 ? 2 : 3;
                        const int j = rowout ? ir * R3W + c : nR * R3W + ic * (TH + 4) + r;
                        const floatx4* tv = reinterpret_cast<const floatx4*>(T3Vp(j));
                        const floatx4* wv = reinterpret_cast<const floatx4*>(W4L + tap * LF_C);
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const floatx4 t = tv[q], ww = wv[q];
                            u += t[0] * ww[0] + t[1] * ww[1] + t[2] * ww[2] + t[3] * ww[3];
                        }
                    }
                }
#pragma unroll
                for (int o = 16; o > 0; o >>= 1) u += __shfl_xor(u, o, 64);      // inside each half of the wave
                if (tap == 0 && ai < nA) CORR[ai] = u;
            }
            __syncthreads();
        }
        {
            const int oy = tid / TW, ox = tid - oy * TW;
            const int gy = y0 + oy, gx = x0 + ox;
            const float invc = (1.f / ACT_SCALE) * tail[12];
            float net = 0.f;
            const float* drow = Dp(oy * R2W) + ox * DP;          // a row of the D image lies inside one quarter
#pragma unroll
            for (int dx = 0; dx < 9; ++dx) net += drow[dx * DP + dx];
            net *= invc;
            net += rs + tail[5];            // b4 + sum_oc (b3 + b1)[oc] * sum_taps w4[oc], summed once per weight update (pack)
            const bool valid = gy < h && gx < w;
            if (!interior) {                   // the conv4 taps of this pixel that land outside the image (CORR, computed above)
                const int ira = (nRA > 0 && oy == RA0) ? 0 : (nRA > 1 && oy == RA1) ? 1 : (nRA > 2 && oy == RA2) ? 2 : (nRA > 3 && oy == RA3) ? 3 : -1;
                const int ica = (nCA > 0 && ox == CA0) ? 0 : (nCA > 1 && ox == CA1) ? 1 : (nCA > 2 && ox == CA2) ? 2 : (nCA > 3 && ox == CA3) ? 3 : -1;
                if (ira >= 0) net -= CORR[ira * TW + ox];
                else if (ica >= 0) net -= CORR[nRA * TW + (oy - rlo) * nCA + ica];
            }
            const float skip = S[(oy + 8) * SW + ox + 8];
            const float din = din_pre;
            if (valid)      // per-image offsets fit 32 bits (checked on the host)
                (vout.dout + zv * vout.dout_sz)[gy * (int)vout.dout_sy + gx * (int)vout.dout_sx] = din + a.sign * (skip + a.rw * net);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (more) stage(tid, nxt);
        LF_STAMP(9)
        LF_STAMP(15)
    } else {

    // ---------------- P3: t3 = conv3(t2) + b3 + r,  r = conv1(skip) + b1 (recomputed), on 20 x 36; dynamic scale
    constexpr int IT3 = (NT3 + NWAVE - 1) / NWAVE;      // 6
    float t3v[IT3][4];
    float s_t3;
    {
        half8 ah[LF_KS], al[LF_KS];
#pragma unroll
        for (int ks = 0; ks < LF_KS; ++ks) ldA<0>(frag, LF_H_C3 / LF_FRAG + ks, lane, ah[ks], al[ks]);
        float bv[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) bv[q] = bias[a.b3 + oc0 + q] + b1v[q];
        const float inv3 = (BWD ? __int_as_float(0x7F000000 - __float_as_int(act2)) : 1.f / ACT_SCALE) * isw3;
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
                const floatx4 accr = conv1_tile(s16, (r + 4) * SW + c + 4, kgoff0, kgoff1, a1h, a1l);
                const int gy = y0 - 2 + r, gx = x0 - 2 + c;
                const bool in = gy >= 0 && gy < h && gx >= 0 && gx < w;
                float gate[4] = {1.f, 1.f, 1.f, 1.f};
                if constexpr (BWD) {            // dr = (1 - t1^2) conv2^T(dpre2) + dt3 (accr = conv1'(g) = dt3 recomputed)
                    const float* tp = a.gate1 + (zv * LF_C + oc0) * (int64_t)h * w + (int64_t)(in ? gy : 0) * w + (in ? gx : 0);
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float t = tp[(int64_t)q * h * w];
                        gate[q] = __builtin_fmaf(-t, t, 1.f);
                    }
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float v = in ? acc[q] * inv3 * gate[q] + accr[q] * inv1 + bv[q] : 0.f;
                    t3v[it][q] = v;
                    amax = fmaxf(amax, fabsf(v));
                }
                if constexpr (BWD) {
                    if (in && r >= 2 && r < 2 + TH && c >= 2 && c < 2 + TW) {
                        float* d = a.sv_t1 + (zv * LF_C + oc0) * (int64_t)h * w + (int64_t)gy * w + gx;
#pragma unroll
                        for (int q = 0; q < 4; ++q) d[(int64_t)q * h * w] = t3v[it][q];
                    }
                }
                if constexpr (TRAIN) {
                    if (in && r >= 2 && r < 2 + TH && c >= 2 && c < 2 + TW) {
                        float* d = a.sv_t3 + (zv * LF_C + oc0) * (int64_t)h * w + (int64_t)gy * w + gx;
#pragma unroll
                        for (int q = 0; q < 4; ++q) d[(int64_t)q * h * w] = t3v[it][q];
                    }
                }
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o, 64));
        if (lane == 0) RED[8 + wave] = amax;
    }
    LF_STAMP(7)
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
    LF_STAMP(8)
    const bool more = next_item < a.nitems;                                  // the NEXT tile's operands (see the composed path)
    LfPre nxt;
    if (more) fetch(next_item, next_j, tid, nxt);

    // ---------------- P4: D[dx][pixel] = sum over (dy, channel) of t3 * w4 on 16 x 36
    {
        half8 ah[LF_KS4], al[LF_KS4];
#pragma unroll
        for (int ks = 0; ks < LF_KS4; ++ks) ldA<0>(frag, LF_H_C4 / LF_FRAG + ks, lane, ah[ks], al[ks]);
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
                acc = mma3<0>(ah[ks], al[ks], bh, bl, acc);
            }
            // D rows = dx: lanes kg == 0 hold dx 0..3, lanes kg == 1 hold dx 4 in register 0
            if (kg == 0) *reinterpret_cast<floatx4*>(D + p * 8) = acc;
            if (kg == 1) D[p * 8 + 4] = acc[0];
        }
    }
    LF_STAMP(10)
    __syncthreads();
    LF_STAMP(11)

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
            const float din = din_pre;
            if constexpr (BWD) a.sv_skip[zv * (int64_t)h * w + (int64_t)gy * w + gx] = net;       // dsk = conv1^T(dr)
            else vout.dout[zv * vout.dout_sz + (int64_t)gy * vout.dout_sy + (int64_t)gx * vout.dout_sx] = din + a.sign * (skip + a.rw * net);
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    if (more) stage(tid, nxt);
    LF_STAMP(12)
    LF_STAMP(15)
    }                           // sequential path
    __syncthreads();            // S, the T / D images and the fp16 skip images are rewritten by the next tile
    item_i = next_item;
    run_j = next_j;
    }                           // tile loop
    if (tid0 == 0 && atomicAdd(&g_lf_queue[a.qslot][1], 1) == (int)gridDim.x - 1) {     // the last workgroup: nobody draws any more
        g_lf_queue[a.qslot][0] = 0;
        g_lf_queue[a.qslot][1] = 0;
    }
}
#undef LF_STAMP

}  // namespace

// diagnostics state (read per launch; set through the C-ABI, never from the environment on the launch path)
static int g_lf_dbg = [] { const char* e = getenv("LLDWT_LF_DBG"); return e ? atoi(e) : 0; }();
static unsigned long long* g_lf_stamps = nullptr;
static int64_t g_lf_stamps_bytes = 0;
static int g_precision = 0;
int split_precision() { return g_precision; }
void split_set_precision(int p) { g_precision = p; }
void lift_f16_set_debug(int dbg) { g_lf_dbg = dbg; }
void lift_f16_set_stamps(void* p, int64_t nbytes) { g_lf_stamps = reinterpret_cast<unsigned long long*>(p); g_lf_stamps_bytes = p ? nbytes : 0; }

// Backward-data of a P/U block = the same chain with transposed, mirrored weights (see the BWD mode of the kernel):
//   conv1' (1 -> 16) = conv4^T: w1'[oc][tap] = w4[oc][24 - tap]          conv2' = conv3^T: w2'[oc][ic][tap] = w3[ic][oc][24 - tap]
//   conv3' = conv2^T: w3'[oc][ic][tap] = w2[ic][oc][24 - tap]            conv4' (16 -> 1) = conv1^T: w4'[ic][tap] = w1[ic][24 - tap]
// written into scratch as [w1' of every plane | w2' | w3' | w4' | 64 zeros per plane (the biases)] -- the layout of stacked forward
// parameters --, then packed like a forward block by one launch
constexpr int BWD_SCRATCH = 2 * LF_C * LF_KK + 2 * LF_C * LF_C * LF_KK + 64;
__global__ void k_lift_bwd_prep(const float* __restrict__ w1, const float* __restrict__ w2, const float* __restrict__ w3,
                                const float* __restrict__ w4, float* __restrict__ scratch) {
    const int plane = blockIdx.x, planes = gridDim.x;
    w1 += (int64_t)plane * LF_C * LF_KK;
    w2 += (int64_t)plane * LF_C * LF_C * LF_KK;
    w3 += (int64_t)plane * LF_C * LF_C * LF_KK;
    w4 += (int64_t)plane * LF_C * LF_KK;
    float* o1 = scratch + (int64_t)plane * LF_C * LF_KK;
    float* o2 = scratch + (int64_t)planes * LF_C * LF_KK + (int64_t)plane * LF_C * LF_C * LF_KK;
    float* o3 = o2 + (int64_t)planes * LF_C * LF_C * LF_KK;
    float* o4 = scratch + (int64_t)planes * (LF_C * LF_KK + 2 * LF_C * LF_C * LF_KK) + (int64_t)plane * LF_C * LF_KK;
    float* z = scratch + (int64_t)planes * (2 * LF_C * LF_KK + 2 * LF_C * LF_C * LF_KK) + plane * 64;
    for (int i = threadIdx.x; i < LF_C * LF_KK; i += blockDim.x) {
        const int c = i / LF_KK, t = i - c * LF_KK;
        o1[i] = w4[c * LF_KK + (LF_KK - 1 - t)];
        o4[i] = w1[c * LF_KK + (LF_KK - 1 - t)];
    }
    for (int i = threadIdx.x; i < LF_C * LF_C * LF_KK; i += blockDim.x) {
        const int oc = i / (LF_C * LF_KK), r = i - oc * (LF_C * LF_KK), ic = r / LF_KK, t = r - ic * LF_KK;
        o2[i] = w3[(ic * LF_C + oc) * LF_KK + (LF_KK - 1 - t)];
        o3[i] = w2[(ic * LF_C + oc) * LF_KK + (LF_KK - 1 - t)];
    }
    if (threadIdx.x < 64) z[threadIdx.x] = 0.f;
}

int lift_f16_pack_bwd(const float* w1, const float* w2, const float* w3, const float* w4, float* scratch, float* packed,
                      int64_t plane_stride, int f16_off, int planes, hipStream_t st) {
    hipLaunchKernelGGL(k_lift_bwd_prep, dim3((unsigned)planes), dim3(256), 0, st, w1, w2, w3, w4, scratch);
    float* o1 = scratch;
    float* o2 = o1 + (int64_t)planes * LF_C * LF_KK;
    float* o3 = o2 + (int64_t)planes * LF_C * LF_C * LF_KK;
    float* o4 = o3 + (int64_t)planes * LF_C * LF_C * LF_KK;
    float* z = o4 + (int64_t)planes * LF_C * LF_KK;     // 64 zeros per plane: b1, b3 (16 per plane) and b4 (1 per plane) all read zeros
    hipLaunchKernelGGL(k_lift_f16_pack, dim3(2, (unsigned)planes), dim3(PACK_NT), 0, st, o1, o2, o3, o4, z, z, z, packed,
                       plane_stride, f16_off, 0);          // the backward chain runs the sequential path: no composed kernels
    return check_launch("lift_f16_pack_bwd");
}
int64_t lift_f16_bwd_scratch_floats(int planes) { return (int64_t)planes * BWD_SCRATCH; }

int lift_f16_pack(const float* w1, const float* w2, const float* w3, const float* w4, const float* b1, const float* b3,
                  const float* b4, float* packed, int64_t plane_stride, int f16_off, int planes, int compose, hipStream_t st) {
    hipLaunchKernelGGL(k_lift_f16_pack, dim3(2, (unsigned)planes), dim3(PACK_NT), 0, st, w1, w2, w3, w4, b1, b3, b4, packed, plane_stride, f16_off,
                       compose);
    return check_launch("lift_f16_pack");
}

int lift_f16_step(const LiftF16Views& v, int64_t Z, int64_t batch, int64_t h, int64_t w, const float* taps,
                  const float* packed, int64_t pstride, int fp32_orient_floats, int f16_off, int vertical, float sign, float rw,
                  hipStream_t st) {
    return lift_f16_step2(v, nullptr, Z, batch, h, w, taps, packed, pstride, fp32_orient_floats, f16_off, vertical, sign, rw, st);
}

int lift_f16_step2(const LiftF16Views& v, const LiftF16Views* v2, int64_t Z, int64_t batch, int64_t h, int64_t w,
                   const float* taps, const float* packed, int64_t pstride, int fp32_orient_floats, int f16_off, int vertical,
                   float sign, float rw, hipStream_t st) {
    return lift_f16_step_any(v, v2, nullptr, Z, batch, h, w, taps, packed, pstride, fp32_orient_floats, f16_off, vertical, sign, rw, st);
}

int lift_f16_step_train(const LiftF16Views& v, const LiftF16Saved& sv, int64_t Z, int64_t batch, int64_t h, int64_t w,
                        const float* taps, const float* packed, int64_t pstride, int fp32_orient_floats, int f16_off,
                        int vertical, float sign, float rw, hipStream_t st) {
    if (!(sv.src && sv.skip && sv.t1 && sv.t2 && sv.t3)) {
        set_error("lift_f16_step_train: null saved buffer");
        return LLDWT_EINVAL;
    }
    return lift_f16_step_any(v, nullptr, &sv, Z, batch, h, w, taps, packed, pstride, fp32_orient_floats, f16_off, vertical, sign, rw, st);
}

static const LiftF16Bwd* g_bwd_call = nullptr;      // set by lift_f16_step_bwd around its call of lift_f16_step_any (host, same thread)

int lift_f16_step_bwd(const LiftF16Bwd& b, int64_t Z, int64_t batch, int64_t h, int64_t w, const float* taps_id,
                      const float* packed_bwd, int64_t pstride, int fp32_orient_floats, int f16_off, int vertical, hipStream_t st) {
    if (!(b.g && b.t1 && b.t2 && b.dt3 && b.dpre2 && b.dr && b.dsk && taps_id && packed_bwd)) {
        set_error("lift_f16_step_bwd: null pointer");
        return LLDWT_EINVAL;
    }
    float* gm = const_cast<float*>(b.g);
    const LiftF16Views v{b.g, h * w, w, 1, b.g, h * w, w, 1, gm, h * w, w, 1};       // din / dout are not used by the BWD mode
    g_bwd_call = &b;
    const int r = lift_f16_step_any(v, nullptr, nullptr, Z, batch, h, w, taps_id, packed_bwd, pstride, fp32_orient_floats, f16_off,
                                    vertical, 1.f, 1.f, st);
    g_bwd_call = nullptr;
    return r;
}

int lift_f16_step_any(const LiftF16Views& v, const LiftF16Views* v2, const LiftF16Saved* sv, int64_t Z, int64_t batch, int64_t h,
                      int64_t w, const float* taps, const float* packed, int64_t pstride, int fp32_orient_floats, int f16_off,
                      int vertical, float sign, float rw, hipStream_t st) {
    const LiftF16Bwd* bw = g_bwd_call;
    static bool attr = false;
    if (!attr) {
        if (hipFuncSetAttribute((const void*)k_lift_fused_f16<false, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_TOTAL2) != hipSuccess ||
            hipFuncSetAttribute((const void*)k_lift_fused_f16<false, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_TOTAL2) != hipSuccess ||
            hipFuncSetAttribute((const void*)k_lift_fused_f16<false, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_TOTAL2) != hipSuccess ||
            hipFuncSetAttribute((const void*)k_lift_fused_f16<true, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_TOTAL2) != hipSuccess ||
            hipFuncSetAttribute((const void*)k_lift_fused_f16<true, 0, true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_TOTAL2) != hipSuccess ||
            hipFuncSetAttribute((const void*)k_lift_fused_f16<true, 0, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_TOTAL2) != hipSuccess) {
            set_error("lift_f16_step: cannot reserve %d bytes of LDS", LDS_TOTAL2);
            return LLDWT_EHIP;
        }
        attr = true;
    }
    LfArgs a;
    a.v = v;
    if (v2 && (v2->src_sy != v.src_sy || v2->src_sx != v.src_sx || v2->din_sy != v.din_sy || v2->din_sx != v.din_sx ||
               v2->dout_sy != v.dout_sy || v2->dout_sx != v.dout_sx)) {
        set_error("lift_f16_step2: the two view sets must share their row / column strides");
        return LLDWT_EINVAL;
    }
    a.src2 = v2 ? v2->src : v.src; a.din2 = v2 ? v2->din : v.din; a.dout2 = v2 ? v2->dout : v.dout;
    a.src2_sz = v2 ? v2->src_sz : v.src_sz; a.din2_sz = v2 ? v2->din_sz : v.din_sz; a.dout2_sz = v2 ? v2->dout_sz : v.dout_sz;
    a.zsplit = Z;
    const int64_t Zl = v2 ? 2 * Z : Z;               // images of the launch
    a.taps = taps;
    a.packed = packed;
    a.pstride = pstride;
    const int orient = vertical ? 0 : 1;
    a.orient_fp32 = orient * fp32_orient_floats;
    // bias offsets inside an fp32 orientation section (mirror of pack_off in lifting.hip for C = 16, K = 5)
    auto pad16 = [](int n) { return (n + 15) & ~15; };
    const int w1o = 0, b1o = w1o + pad16(LF_KK * LF_C), w2o = b1o + pad16(LF_C), b2o = w2o + pad16(LF_C * LF_KK * LF_C);
    const int w3o = b2o + pad16(LF_C), b3o = w3o + pad16(LF_C * LF_KK * LF_C), w4o = b3o + pad16(LF_C), b4o = w4o + pad16(LF_C * LF_KK);
    a.b1 = b1o; a.b2 = b2o; a.b3 = b3o; a.b4 = b4o; a.w4 = w4o;
    a.f16 = f16_off + orient * LF_ORIENT_FLOATS;
    a.batch = (int)batch; a.h = (int)h; a.w = (int)w; a.vertical = vertical;
    a.sign = sign; a.rw = rw;
    a.dbg = g_lf_dbg;
    a.stamps = nullptr;
    auto fits = [&](int64_t sy, int64_t sx) { return llabs(sy) * h + llabs(sx) * w < (int64_t)1 << 31; };
    if (!fits(v.src_sy, v.src_sx) || !fits(v.din_sy, v.din_sx) || !fits(v.dout_sy, v.dout_sx) ||
        (v2 && (!fits(v2->src_sy, v2->src_sx) || !fits(v2->din_sy, v2->din_sx) || !fits(v2->dout_sy, v2->dout_sx)))) {
        set_error("lift_f16_step: per-image strides beyond 32-bit offsets");
        return LLDWT_EINVAL;
    }
    a.tiles_x = (int)cdiv(w, TW);
    a.tiles_y = (int)cdiv(h, TH);
    if ((int64_t)a.tiles_x * a.tiles_y * Zl >= (int64_t)1 << 30) {
        set_error("lift_f16_step: more than 2^30 tiles in one launch");
        return LLDWT_EINVAL;
    }
    if (g_lf_stamps) {              // diagnostics (tools/lift_stamps.py): only when the registered buffer holds every tile's stamps
        const int64_t need = (int64_t)a.tiles_x * a.tiles_y * Zl * NWAVE * 16 * 8;
        if (g_lf_stamps_bytes >= need) a.stamps = g_lf_stamps;
    }
    static int ncu = 0;
    if (ncu == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess || prop.multiProcessorCount <= 0) {
            set_error("lift_f16_step: cannot read the device's CU count");
            return LLDWT_EHIP;
        }
        ncu = prop.multiProcessorCount;
    }
    // the sequential evaluation (debug check, TRAIN, BWD) hands down T2 only (its T3 image overwrites T1): a continuing tile still
    // saves a third of conv2 and 29 % of conv1
    const bool seq = (a.dbg & 16) != 0 || sv != nullptr || bw != nullptr;
    a.sv_src = sv ? sv->src : nullptr; a.sv_skip = sv ? sv->skip : nullptr;
    a.sv_t1 = sv ? sv->t1 : nullptr; a.sv_t2 = sv ? sv->t2 : nullptr; a.sv_t3 = sv ? sv->t3 : nullptr;
    a.gate1 = nullptr; a.gate2 = nullptr; a.mx = nullptr;
    if (bw) {
        a.sv_skip = bw->dsk; a.sv_t1 = bw->dr; a.sv_t2 = bw->dpre2; a.sv_t3 = bw->dt3;
        a.gate1 = bw->t1; a.gate2 = bw->t2;
        a.mx = bw->mx;
    }
    // run length: a tile that continues a run costs ~0.8 of a first tile; runs are dealt round-robin to one resident
    // workgroup per CU, so the launch takes rounds x (cost of a run) -- the longest run that still fills whole rounds
    int best_rl = 1;
    if (!(a.dbg & 32)) {
        const double cont_cost = seq ? 0.85 : 0.8;
        double best = 1e30;
        for (int rl = 1; rl <= a.tiles_y; ++rl) {
            const int64_t items = (int64_t)Zl * a.tiles_x * cdiv(a.tiles_y, rl);
            const double cost = (double)cdiv(items, ncu) * (1.0 + cont_cost * (rl - 1));
            if (cost < best - 1e-9) { best = cost; best_rl = rl; }
        }
    }
    static const int force_rl = [] { const char* e = getenv("LLDWT_LF_RL"); return e ? atoi(e) : 0; }();      // diagnostics: fixed run length
    if (force_rl > 0) best_rl = force_rl < a.tiles_y ? force_rl : a.tiles_y;
    a.rl = best_rl;
    a.nseg = (int)cdiv(a.tiles_y, a.rl);
    a.nitems = (int)(Zl * a.tiles_x * a.nseg);
    a.nborder = (int)(Zl * (a.tiles_x >= 2 ? 2 : 1) * a.nseg);
    static unsigned launch_seq = 0;
    a.qslot = (int)(launch_seq++ & 63u);
    const unsigned grid = (unsigned)(a.nitems < ncu ? a.nitems : ncu);      // one resident workgroup per CU
    const int prec = seq ? 0 : g_precision;
    if (bw) hipLaunchKernelGGL((k_lift_fused_f16<true, 0, false, true>), dim3(grid), dim3(NTH), LDS_TOTAL2, st, a);
    else if (sv) hipLaunchKernelGGL((k_lift_fused_f16<true, 0, true>), dim3(grid), dim3(NTH), LDS_TOTAL2, st, a);
    else if (seq) hipLaunchKernelGGL((k_lift_fused_f16<true, 0>), dim3(grid), dim3(NTH), LDS_TOTAL2, st, a);
    else if (prec == 1) hipLaunchKernelGGL((k_lift_fused_f16<false, 1>), dim3(grid), dim3(NTH), LDS_TOTAL2, st, a);
    else if (prec == 2) hipLaunchKernelGGL((k_lift_fused_f16<false, 2>), dim3(grid), dim3(NTH), LDS_TOTAL2, st, a);
    else hipLaunchKernelGGL((k_lift_fused_f16<false, 0>), dim3(grid), dim3(NTH), LDS_TOTAL2, st, a);
    return check_launch("lift_f16_step");
}

}  // namespace lldwt
