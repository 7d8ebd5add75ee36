// cgp_fused.hip -- the "cgp" parameter network of DWTConditioned2EntropyLayerZTsepSubbands fused with the Gaussian rate.
//
// Reference (LiftingBasedDWT_net.py:282-289,357-365): per subband g the concatenated contexts (plc_g, csc_g) go through
// grouped 1x1 convs C0 -> C1 -> C2 -> C3 -> 2 (162 -> 162 -> 54 -> 18 -> 2) with LeakyReLU between, the two outputs are
// (sigma, mu) of the conditional Gaussian whose likelihood gives the bits of the coefficient.  Unfused this moves
// ~6 KB per pixel through HBM for 36 kMAC; here one workgroup keeps a 64-pixel column of activations in LDS, runs the
// four layers on the matrix cores (v_mfma_f32_16x16x4_f32, exact fp32) and writes only the bits (4 B / coefficient).
//   * waves split OUTPUT CHANNELS: each wave streams the A operands (weights, pre-packed in lane order) of its own
//     16-channel tiles straight from L2 -- no LDS staging of weights, no duplication across waves;
//   * all waves share the B operand (activations) in one LDS buffer, overwritten IN PLACE after each layer (the
//     accumulators stay in registers until every wave has finished reading the layer's input).
#include "common.h"

namespace lldwt {

typedef float floatx4 __attribute__((ext_vector_type(4)));

constexpr int CGP_PX = 64;          // pixels per workgroup
constexpr int CGP_PS = CGP_PX + 16; // LDS row stride (dwords), == 16 mod 32 -> conflict-free B reads
constexpr int CGP_MAXT = 3;         // 16-channel tiles per wave (layer 1: 11 tiles over 4 waves)
constexpr int CGP_NPT = CGP_PX / 16;

struct CgpDims {
    int c[5];        // channels per group: C0 (in), C1, C2, C3, C4 (=2)
    int woff[4];     // float offset of layer l's packed weights inside a (plane, group) block
    int boff[4];     // float offset of layer l's bias
    int mt[4];       // 16-channel tiles packed for layer l = 4 waves x the kernel's tiles per wave (zero weights beyond M)
    int group_floats;
};

// 16-channel tiles of a layer, padded to a multiple of the 4 waves (zero weights): every wave owns the same number
static inline __host__ __device__ int cgp_tiles(int M) { return ((((M + 15) / 16) + 3) / 4) * 4; }

// wide: which layers the kernels run with CGP_MAXT tiles per wave (forward: the first; backward: the last two)
static inline CgpDims cgp_dims5(int c0, int c1, int c2, int c3, int c4, unsigned wide) {
    CgpDims d;
    d.c[0] = c0; d.c[1] = c1; d.c[2] = c2; d.c[3] = c3; d.c[4] = c4;
    int off = 0;
    for (int l = 0; l < 4; ++l) {
        d.mt[l] = 4 * (((wide >> l) & 1u) ? CGP_MAXT : 1);
        d.woff[l] = off;
        off += d.mt[l] * (int)cdiv(d.c[l] + 1, 4) * 64;      // K + 1: the bias is the weight of a constant-1 row
    }
    for (int l = 0; l < 4; ++l) {
        d.boff[l] = off;
        off += (int)round_up(d.c[l + 1], 16);
    }
    d.group_floats = off;
    return d;
}
static inline CgpDims cgp_dims(int c0, int c1, int c2, int c3) { return cgp_dims5(c0, c1, c2, c3, 2, 0x1u); }
// the backward-data stack runs the layers in reverse with transposed weights: 2 -> c3 -> c2 -> c1 -> c0
static inline CgpDims cgp_dims_bwd(int c0, int c1, int c2, int c3) { return cgp_dims5(2, c3, c2, c1, c0, 0xCu); }

// packed[plane][group] = { for each layer: [oc tile][k step][lane] weights, then biases }
__global__ void k_cgp_pack(const float* __restrict__ w0, const float* __restrict__ b0, const float* __restrict__ w1,
                           const float* __restrict__ b1, const float* __restrict__ w2, const float* __restrict__ b2,
                           const float* __restrict__ w3, const float* __restrict__ b3, float* __restrict__ packed,
                           CgpDims d, int groups, int transposed) {
    const int plane = blockIdx.z, g = blockIdx.y;
    const float* ws[4] = {w0, w1, w2, w3};
    const float* bs[4] = {b0, b1, b2, b3};
    float* dst = packed + ((int64_t)plane * groups + g) * d.group_floats;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < d.group_floats; i += gridDim.x * blockDim.x) {
        float v = 0.f;
        for (int l = 0; l < 4; ++l) {
            const int K = d.c[l], M = d.c[l + 1];
            const int ks = (K + 1 + 3) / 4, mt = d.mt[l];
            if (i >= d.woff[l] && i < d.woff[l] + mt * ks * 64) {
                const int j = i - d.woff[l];
                const int lane = j % 64, kstep = (j / 64) % ks, t = j / (64 * ks);
                const int oc = t * 16 + (lane & 15), k = 4 * kstep + (lane >> 4);
                // transposed: ws[l] is the FORWARD weight of the mirrored layer, (planes, groups*K, M): element (k, oc)
                if (oc < M && k < K)
                    v = transposed ? ws[l][((int64_t)plane * groups * K + (int64_t)g * K + k) * M + oc]
                                   : ws[l][((int64_t)plane * groups * M + (int64_t)g * M + oc) * K + k];
                // row K of the activations is the constant 1: its weight is the bias (none in the backward stack)
                if (oc < M && k == K && bs[l]) v = bs[l][(int64_t)plane * groups * M + g * M + oc];
            }
        }
        dst[i] = v;
    }
}

// one dense layer on the LDS column: acc = W . buf ; result kept in registers.  Wave w owns tiles [w*MT, (w+1)*MT).
// Branch-free hot loop: A operands stream from L2 through a double-buffered register ring (the loads of k-steps
// [s0+U, s0+2U) are in flight while the MFMAs of [s0, s0+U) issue; addresses are clamped instead of predicated).
constexpr int CGP_U = 4;            // depth of the A-operand register ring (k-steps)

// first CGP_U k-steps of wave `wave`'s A operands: issued EARLY (before the barriers / epilogue of the previous layer) so
// that the L2 latency of the ring warm-up is not paid between two barriers
template <int MT>
__device__ __forceinline__ void cgp_warm(float (&An)[CGP_U][MT], const float* __restrict__ pk, int K, int wave, int lane) {
    const int ks = (K + 3) / 4;
    const float* pa = pk + (int64_t)(wave * MT) * ks * 64 + lane;
#pragma unroll
    for (int u = 0; u < CGP_U; ++u)
#pragma unroll
        for (int j = 0; j < MT; ++j) An[u][j] = pa[(j * ks + min(u, ks - 1)) * 64];
}

template <int MT>
__device__ __forceinline__ void cgp_layer(const float* __restrict__ buf, const float* __restrict__ pk, int K, int wave,
                                          int lane, floatx4 (&acc)[MT][CGP_NPT], float (&An)[CGP_U][MT]) {
    const int ks = (K + 3) / 4;
    const int px = lane & 15, kk = lane >> 4;
#pragma unroll
    for (int j = 0; j < MT; ++j)
#pragma unroll
        for (int n = 0; n < CGP_NPT; ++n) acc[j][n] = floatx4{0.f, 0.f, 0.f, 0.f};
    const float* bb = buf + kk * CGP_PS + px;
    const float* pa = pk + (int64_t)(wave * MT) * ks * 64 + lane;
    constexpr int U = CGP_U;
    int s0 = 0;
    for (; s0 + U <= ks; s0 += U) {
        float Ac[U][MT];
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int j = 0; j < MT; ++j) {
                Ac[u][j] = An[u][j];
                An[u][j] = pa[(j * ks + min(s0 + U + u, ks - 1)) * 64];
            }
        // keep the prefetch loads ABOVE the MFMAs: without this fence hipcc sinks each load next to its use and
        // waits vmcnt(0) per k-step (seen in the .s), which serialises the loop on L2 latency
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            float B[CGP_NPT];
#pragma unroll
            for (int n = 0; n < CGP_NPT; ++n) B[n] = bb[(4 * (s0 + u)) * CGP_PS + n * 16];
#pragma unroll
            for (int j = 0; j < MT; ++j)
#pragma unroll
                for (int n = 0; n < CGP_NPT; ++n)
                    acc[j][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(Ac[u][j], B[n], acc[j][n], 0, 0, 0);
        }
    }
#pragma unroll 1
    for (int s = s0; s < ks; ++s) {      // tail: ks % U steps
        float B[CGP_NPT], A[MT];
#pragma unroll
        for (int j = 0; j < MT; ++j) A[j] = pa[(j * ks + s) * 64];
#pragma unroll
        for (int n = 0; n < CGP_NPT; ++n) B[n] = bb[(4 * s) * CGP_PS + n * 16];
#pragma unroll
        for (int j = 0; j < MT; ++j)
#pragma unroll
            for (int n = 0; n < CGP_NPT; ++n)
                acc[j][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[j], B[n], acc[j][n], 0, 0, 0);
    }
}

// hout (training): the hidden activations also go to HBM, (Z, groups*M, hw) -- same layout the unfused convs produce
template <int MT>
__device__ __forceinline__ void cgp_store(float* __restrict__ buf, int M, int wave, int lane,
                                          const floatx4 (&acc)[MT][CGP_NPT], float* __restrict__ hout = nullptr,
                                          int64_t hw = 0, int64_t p0 = 0) {
    const int px = lane & 15, kk = lane >> 4;
    const int mpad = (M + 1 + 3) & ~3;  // row M = the constant 1 that carries the next layer's bias, then zero padding of K
#pragma unroll
    for (int j = 0; j < MT; ++j) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int oc = (wave * MT + j) * 16 + 4 * kk + r;
            if (oc < mpad) {
#pragma unroll
                for (int n = 0; n < CGP_NPT; ++n) {
                    float v = acc[j][n][r];                   // bias already inside (weight of the constant-1 row)
                    v = v >= 0.f ? v : 0.01f * v;             // LeakyReLU(0.01)
                    v = oc < M ? v : (oc == M ? 1.f : 0.f);
                    buf[oc * CGP_PS + n * 16 + px] = v;
                    if (hout && oc < M && p0 + n * 16 + px < hw) hout[(int64_t)oc * hw + p0 + n * 16 + px] = v;
                }
            }
        }
    }
}

// one hidden layer; An holds its warmed-up ring, AnNext receives the next layer's (issued before the barriers)
template <int MT, int MTN>
__device__ __forceinline__ void cgp_hidden(float* __restrict__ buf, const float* __restrict__ pk, const CgpDims& d, int l,
                                           int wave, int lane, float (&An)[CGP_U][MT], float (&AnNext)[CGP_U][MTN],
                                           int wave_next, float* __restrict__ hout = nullptr, int64_t hw = 0,
                                           int64_t p0 = 0) {
    floatx4 acc[MT][CGP_NPT];
    cgp_layer<MT>(buf, pk + d.woff[l], d.c[l] + 1, wave, lane, acc, An);
    cgp_warm<MTN>(AnNext, pk + d.woff[l + 1], d.c[l + 1] + 1, wave_next, lane);
    __syncthreads();                       // every wave has read this layer's input
    cgp_store<MT>(buf, d.c[l + 1], wave, lane, acc, hout, hw, p0);
    __syncthreads();
}

// Optional folded context (eval): the masked csc conv (LiftingBasedDWT_net.py:275-277,353) feeds the first cgp layer with
// no nonlinearity in between, so W0[:, csc half] . Wcsc is folded on the host into 12 extra columns of layer 0 whose
// inputs are the live taps of the quantised subband itself -- gathered here, straight from the 1-channel image.  Rows
// 0..cplc-1 of the column then come from the plc tensor, rows cplc..cplc+npatch-1 from xq at (y + dy - R, x + dx - R).
struct CgpCtx {
    const float* xq;     // (Z, groups, h, w) quantised coefficients, or null: all c0 rows come from `cat`
    int cplc, npatch, w, R;
    int8_t tdy[16], tdx[16];
};
constexpr int CGP_ROWS_MAX = 164;                         // input channels per group supported by the register prefetch
constexpr int CGP_NIN = CGP_ROWS_MAX * CGP_PX / 256;      // input floats staged per thread (full-width column)
constexpr int CGP_NIN_SMALL = 96 * CGP_PX / 256;          // ... when the input column has <= 96 rows (folded context)
constexpr int CGP_TILES_PER_WG = 8;                       // consecutive 64-pixel columns per workgroup

// Persistent over CGP_TILES_PER_WG consecutive pixel columns: the next column's input is prefetched into registers
// while the matrix work of the current one runs (issue early / write late), so HBM latency never parks the waves.
template <bool TRAIN, int NIN>  // TRAIN: also write the hidden activations (kept out of the eval kernel: the extra pointers
                                // spill); NIN: input rows staged per thread (rows / 4 waves)
__global__ __launch_bounds__(256, 2) void k_cgp_rate(const float* __restrict__ cat, const float* __restrict__ x,
                                                  const float* __restrict__ noise, const float* __restrict__ packed,
                                                  float* __restrict__ bits, float* __restrict__ params_out,
                                                  double* __restrict__ bit_sum, CgpDims d, int groups, int batch,
                                                  int64_t hw, float* __restrict__ h1, float* __restrict__ h2,
                                                  float* __restrict__ h3, CgpCtx cx, int rows_max) {
    extern __shared__ __attribute__((aligned(16))) float buf[];     // [rows_max][CGP_PS] + sigma/mu [2][64]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = blockIdx.y;
    const int64_t z = blockIdx.z;
    const int plane = (int)(z / batch);
    const float* pk = packed + ((int64_t)plane * groups + g) * d.group_floats;
    const int C0 = d.c[0];
    const int rows0 = (C0 + 1 + 3) & ~3;                            // + the constant-1 row that carries the biases
    float* sm = buf + rows_max * CGP_PS;                            // sigma[64], mu[64]
    const int cplc = cx.xq ? cx.cplc : C0;                          // rows that come from `cat`
    const float* src = cat + (z * (int64_t)groups * cplc + (int64_t)g * cplc) * hw;
    const float* xqg = cx.xq ? cx.xq + (z * groups + g) * hw : src;
    const int h_img = cx.xq ? (int)(hw / cx.w) : 1;
    // training: hidden activations of this (image, group) in the layout of the unfused convs
    float* h1g = TRAIN ? h1 + (z * groups + g) * (int64_t)d.c[1] * hw : nullptr;
    float* h2g = TRAIN ? h2 + (z * groups + g) * (int64_t)d.c[2] * hw : nullptr;
    float* h3g = TRAIN ? h3 + (z * groups + g) * (int64_t)d.c[3] * hw : nullptr;
    const int64_t ntiles = (hw + CGP_PX - 1) / CGP_PX;
    const int64_t t0 = (int64_t)blockIdx.x * CGP_TILES_PER_WG;
    // this thread stages channels c0, c0+4, ...: c0 = wave id is wave-uniform -> row bases live in SGPRs
    const int p = tid & (CGP_PX - 1), c0 = __builtin_amdgcn_readfirstlane(tid >> 6);

    float xin[NIN];
    // patch rows: is tap (row c) of pixel pp inside the image?  (recomputed at the LDS store: the load is raw)
#define LLDWT_CGP_PATCH(pp_, c_, ok_, off_)                                                          \
    const int t_ = (c_) - cplc;                                                                      \
    const int py_ = (pp_) / cx.w, px_ = (pp_) - py_ * cx.w;                                          \
    const int qy_ = py_ + cx.tdy[t_ & 15] - cx.R, qx_ = px_ + cx.tdx[t_ & 15] - cx.R;               \
    const bool ok_ = (pp_) < hw && qy_ >= 0 && qy_ < h_img && qx_ >= 0 && qx_ < cx.w;                \
    const int off_ = qy_ * cx.w + qx_;
#define LLDWT_CGP_LOAD(TILE)                                                                         \
    {                                                                                                \
        const int pp = (int)((TILE) * CGP_PX) + p;                                                   \
        _Pragma("unroll") for (int r = 0; r < NIN; ++r) {                                            \
            const int c = c0 + 4 * r;                                                                \
            /* raw load from a safe address; the zeroing of padding happens at the LDS store, one tile later */ \
            /* (a select here makes hipcc wait for every load in turn: 29 serialised round trips in the .s)  */ \
            if (c < cplc || c >= C0) {                                       /* wave-uniform */       \
                const float* row = src + (int64_t)(c < cplc ? c : 0) * hw;   /* scalar */             \
                xin[r] = row[pp < hw ? pp : 0];                                                      \
            } else {                                                                                 \
                LLDWT_CGP_PATCH(pp, c, ok, off)                                                      \
                xin[r] = xqg[ok ? off : 0];                                                          \
            }                                                                                        \
        }                                                                                            \
    }
    if (t0 < ntiles) LLDWT_CGP_LOAD(t0)
    double local = 0;
    for (int64_t t = t0; t < t0 + CGP_TILES_PER_WG && t < ntiles; ++t) {
        const int64_t p0 = t * CGP_PX;
        __syncthreads();                       // previous column fully consumed
#pragma unroll
        for (int r = 0; r < NIN; ++r) {
            const int c = c0 + 4 * r;
            if (c < rows0) {
                float v;
                if (c < cplc) {
                    v = p0 + p < hw ? xin[r] : 0.f;
                } else if (c < C0) {
                    LLDWT_CGP_PATCH((int)p0 + p, c, ok, off)
                    (void)off;
                    v = ok ? xin[r] : 0.f;
                } else {
                    v = c == C0 ? 1.f : 0.f;
                }
                buf[c * CGP_PS + p] = v;
            }
        }
        __syncthreads();
        {
            float A0[CGP_U][CGP_MAXT], A1[CGP_U][1], A2[CGP_U][1], A3[CGP_U][1];
            cgp_warm<CGP_MAXT>(A0, pk + d.woff[0], d.c[0] + 1, wave, lane);
            cgp_hidden<CGP_MAXT, 1>(buf, pk, d, 0, wave, lane, A0, A1, wave, h1g, hw, p0);
            // the next column's input is requested only now: during layer 0 (48 accumulators + the 24-register operand ring
            // per lane) 41 more live registers spilled; layers 1-3 and the rate leave ~10 us to cover the latency
            if (t + 1 < t0 + CGP_TILES_PER_WG && t + 1 < ntiles) LLDWT_CGP_LOAD(t + 1)
            cgp_hidden<1, 1>(buf, pk, d, 1, wave, lane, A1, A2, wave, h2g, hw, p0);
            cgp_hidden<1, 1>(buf, pk, d, 2, wave, lane, A2, A3, 0, h3g, hw, p0);
        // ---- last layer (-> sigma, mu) on wave 0 (LiftingBasedDWT_net.py:360-362)
        if (wave == 0) {
            floatx4 acc[1][CGP_NPT];
            cgp_layer<1>(buf, pk + d.woff[3], d.c[3] + 1, 0, lane, acc, A3);
            if (lane < 16) {
#pragma unroll
                for (int n = 0; n < CGP_NPT; ++n) {
                    sm[n * 16 + lane] = acc[0][n][0];
                    sm[CGP_PX + n * 16 + lane] = acc[0][n][1];
                }
            }
        }
        }
        __syncthreads();
        // ---- Gaussian rate, one pixel per lane, on wave 1 (:364-365)
        if (wave == 1) {
            const int64_t pp = p0 + lane;
            if (pp < hw) {
                const float sg = sm[lane], mu = sm[CGP_PX + lane];
                const int64_t idx = (z * groups + g) * hw + pp;
                const float xv = x[idx];
                const float v = noise ? xv + noise[idx] : rintf(xv - mu) + mu;
                const float a = fabsf(v - mu);
                const float sc = fmaxf(sg, 0.11f);
                const float cst = -0.70710678118654752440f;
                const float up = 0.5f * erfcf(cst * ((0.5f - a) / sc));
                const float lo = 0.5f * erfcf(cst * ((-0.5f - a) / sc));
                const float b = -log2f(fmaxf(up - lo, 1e-9f));
                bits[idx] = b;
                local += (double)b;
                if (params_out) {
                    params_out[(z * 2 * groups + 2 * g) * hw + pp] = sg;
                    params_out[(z * 2 * groups + 2 * g + 1) * hw + pp] = mu;
                }
            }
        }
    }
#undef LLDWT_CGP_LOAD
#undef LLDWT_CGP_PATCH
    if (bit_sum && wave == 1) {
        local = wave_sum(local);
        if (lane == 0) atomicAdd(bit_sum, local);
    }
}

// ---- backward-data of the stack (training) ---------------------------------------------------------------------------
// dparams (dsigma, dmu) -> d3 -> d2 -> d1 -> dcat with the transposed weights; d_l = (W_{l+1}^T d_{l+1}) * LeakyReLU'(h_l)
// is the gradient at the PRE-activation output of layer l (what the weight-gradient GEMM of layer l needs), written to
// HBM next to the column kept in LDS.  Same machinery as the forward: waves split output channels, A operands stream
// from L2 through the register ring, one LDS column overwritten in place.
template <int MT>
__device__ __forceinline__ void cgp_gate_load(float (&gate)[MT][4][CGP_NPT], const float* __restrict__ h, int M, int wave,
                                              int lane, int64_t hw, int64_t p0) {
    const int px = lane & 15, kk = lane >> 4;
#pragma unroll
    for (int j = 0; j < MT; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int oc = (wave * MT + j) * 16 + 4 * kk + r;
#pragma unroll
            for (int n = 0; n < CGP_NPT; ++n) {
                const int64_t p = p0 + n * 16 + px;
                const bool ok = oc < M && p < hw;
                gate[j][r][n] = h[ok ? (int64_t)oc * hw + p : 0];         // RAW activation; turned into the gate at the store
            }
        }
}

template <int MT, bool GATED>
__device__ __forceinline__ void cgp_store_bwd(float* __restrict__ buf, const float (&gate)[MT][4][CGP_NPT], int M, int wave,
                                              int lane, const floatx4 (&acc)[MT][CGP_NPT], float* __restrict__ out,
                                              int64_t hw, int64_t p0, bool to_lds, float* __restrict__ out2 = nullptr,
                                              int split = 1 << 30) {
    // rows >= split go to a second tensor (out2 = its base minus split rows): the input gradient of the folded layer 0 leaves as
    // [tree-context channels | gathered taps] without ever having been one tensor
    const int px = lane & 15, kk = lane >> 4;
    const int mpad = (M + 1 + 3) & ~3;       // the packed K of the next layer is M + 1 (its bias row: zero weights here)
#pragma unroll
    for (int j = 0; j < MT; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int oc = (wave * MT + j) * 16 + 4 * kk + r;
            if (oc < mpad) {
#pragma unroll
                for (int n = 0; n < CGP_NPT; ++n) {
                    float v = oc < M ? acc[j][n][r] : 0.f;
                    if (GATED) v *= gate[j][r][n] > 0.f ? 1.f : 0.01f;    // LeakyReLU' from the sign of the stored activation
                    if (to_lds) buf[oc * CGP_PS + n * 16 + px] = v;
                    if (oc < M && p0 + n * 16 + px < hw) (oc < split ? out : out2)[(int64_t)oc * hw + p0 + n * 16 + px] = v;
                }
            }
        }
}

template <int MT, int MTN, bool GATED>
__device__ __forceinline__ void cgp_bwd_layer(float* __restrict__ buf, const float* __restrict__ pk, const CgpDims& d, int l,
                                              int wave, int lane, float (&An)[CGP_U][MT], float (&AnNext)[CGP_U][MTN],
                                              const float* __restrict__ h, float* __restrict__ out, int64_t hw, int64_t p0,
                                              bool last, float* __restrict__ out2 = nullptr, int split = 1 << 30) {
    floatx4 acc[MT][CGP_NPT];
    float gate[MT][4][CGP_NPT];
    if (GATED) cgp_gate_load<MT>(gate, h, d.c[l + 1], wave, lane, hw, p0);
    cgp_layer<MT>(buf, pk + d.woff[l], d.c[l] + 1, wave, lane, acc, An);
    if (!last) cgp_warm<MTN>(AnNext, pk + d.woff[l + 1], d.c[l + 1] + 1, wave, lane);
    __syncthreads();                       // every wave has read this layer's input
    cgp_store_bwd<MT, GATED>(buf, gate, d.c[l + 1], wave, lane, acc, out, hw, p0, !last, out2, split);
    __syncthreads();
}

__global__ __launch_bounds__(256, 2) void k_cgp_bwd(const float* __restrict__ dparams, const float* __restrict__ h1,
                                                 const float* __restrict__ h2, const float* __restrict__ h3,
                                                 const float* __restrict__ packed, float* __restrict__ d1,
                                                 float* __restrict__ d2, float* __restrict__ d3,
                                                 float* __restrict__ dcat, CgpDims d, int groups, int batch, int64_t hw,
                                                 float* __restrict__ dtaps, int split) {
    // dtaps != null: the input gradient goes to two tensors, rows < split of every group to dcat (Z, groups*split, hw), the rest to
    // dtaps (Z, groups*(c0 - split), hw)
    extern __shared__ __attribute__((aligned(16))) float buf[];     // [roundup(max width,4)][CGP_PS]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = blockIdx.y;
    const int64_t z = blockIdx.z;
    const int plane = (int)(z / batch);
    const float* pk = packed + ((int64_t)plane * groups + g) * d.group_floats;
    // d.c = {2, c3, c2, c1, c0}
    const int64_t zg = z * groups + g;
    const float* dpz = dparams + zg * 2 * hw;
    const float* h3g = h3 + zg * (int64_t)d.c[1] * hw;
    const float* h2g = h2 + zg * (int64_t)d.c[2] * hw;
    const float* h1g = h1 + zg * (int64_t)d.c[3] * hw;
    float* d3g = d3 + zg * (int64_t)d.c[1] * hw;
    float* d2g = d2 + zg * (int64_t)d.c[2] * hw;
    float* d1g = d1 + zg * (int64_t)d.c[3] * hw;
    float* dcg = dcat + zg * (int64_t)(dtaps ? split : d.c[4]) * hw;
    float* dtg = dtaps ? dtaps + zg * (int64_t)(d.c[4] - split) * hw - (int64_t)split * hw : nullptr;
    const int spl = dtaps ? split : (1 << 30);
    const int64_t ntiles = (hw + CGP_PX - 1) / CGP_PX;
    const int64_t t0 = (int64_t)blockIdx.x * CGP_TILES_PER_WG;
    for (int64_t t = t0; t < t0 + CGP_TILES_PER_WG && t < ntiles; ++t) {
        const int64_t p0 = t * CGP_PX;
        __syncthreads();                       // previous column fully consumed
        {
            // rows 0,1 = (dsigma, dmu) of this group, rows 2,3 = the zero padding of K
            const int row = tid >> 6, p = tid & 63;
            const bool ok = row < 2 && p0 + p < hw;
            const float v = dpz[ok ? (int64_t)row * hw + p0 + p : 0];
            buf[row * CGP_PS + p] = ok ? v : 0.f;
        }
        float A0[CGP_U][1], A1[CGP_U][1], A2[CGP_U][CGP_MAXT], A3[CGP_U][CGP_MAXT];
        cgp_warm<1>(A0, pk + d.woff[0], d.c[0] + 1, wave, lane);
        __syncthreads();
        cgp_bwd_layer<1, 1, true>(buf, pk, d, 0, wave, lane, A0, A1, h3g, d3g, hw, p0, false);
        cgp_bwd_layer<1, CGP_MAXT, true>(buf, pk, d, 1, wave, lane, A1, A2, h2g, d2g, hw, p0, false);
        cgp_bwd_layer<CGP_MAXT, CGP_MAXT, true>(buf, pk, d, 2, wave, lane, A2, A3, h1g, d1g, hw, p0, false);
        cgp_bwd_layer<CGP_MAXT, CGP_MAXT, false>(buf, pk, d, 3, wave, lane, A3, A3, nullptr, dcg, hw, p0, true, dtg, spl);
    }
}

}  // namespace lldwt
using namespace lldwt;

static int cgp_dims_ok(const char* who, int c0, int c1, int c2, int c3, int groups) {
    LLDWT_REQUIRE(groups > 0 && c0 > 0 && c1 > 0 && c2 > 0 && c3 > 0, "%s: bad channel counts", who);
    LLDWT_REQUIRE(c1 < 64 * CGP_MAXT && c2 < 64 && c3 < 64,
                  "%s: hidden widths (%d,%d,%d) exceed the built tile plan (%d,63,63)", who, c1, c2, c3, 64 * CGP_MAXT - 1);
    LLDWT_REQUIRE(round_up(c0 + 1, 4) <= CGP_ROWS_MAX && round_up(c1 + 1, 4) <= CGP_ROWS_MAX,
                  "%s: widths (%d,%d) exceed the %d-row LDS column", who, c0, c1, CGP_ROWS_MAX - 1);
    return 0;
}

extern "C" int64_t lldwt_cgp_packed_floats(int c0, int c1, int c2, int c3, int groups) {
    if (c0 <= 0 || c1 <= 0 || c2 <= 0 || c3 <= 0 || groups <= 0) return -1;
    return (int64_t)cgp_dims(c0, c1, c2, c3).group_floats * groups;
}

extern "C" int lldwt_cgp_pack(const float* w0, const float* b0, const float* w1, const float* b1, const float* w2,
                              const float* b2, const float* w3, const float* b3, float* packed, int64_t planes,
                              int c0, int c1, int c2, int c3, int groups, void* stream) {
    int r = cgp_dims_ok("cgp_pack", c0, c1, c2, c3, groups);
    if (r) return r;
    LLDWT_REQUIRE(w0 && b0 && w1 && b1 && w2 && b2 && w3 && b3 && packed && planes > 0, "cgp_pack: null pointer");
    const CgpDims d = cgp_dims(c0, c1, c2, c3);
    dim3 grid((unsigned)cdiv(d.group_floats, 256), (unsigned)groups, (unsigned)planes);
    hipLaunchKernelGGL(k_cgp_pack, grid, dim3(256), 0, (hipStream_t)stream, w0, b0, w1, b1, w2, b2, w3, b3, packed, d, groups, 0);
    return check_launch("cgp_pack");
}

static int cgp_rate_impl(const float* cat, const float* x, const float* noise, const float* packed, float* bits,
                         float* params_out, float* h1, float* h2, float* h3, double* bit_sum, int64_t planes, int64_t batch,
                         int64_t hw, int c0, int c1, int c2, int c3, int groups, const CgpCtx& cx, void* stream) {
    int r = cgp_dims_ok("cgp_rate", c0, c1, c2, c3, groups);
    if (r) return r;
    LLDWT_REQUIRE(cat && x && packed && bits && planes > 0 && batch > 0 && hw > 0 && planes * batch <= 65535,
                  "cgp_rate: bad arguments");
    const CgpDims d = cgp_dims(c0, c1, c2, c3);
    int rows_max = 4;
    for (int l = 0; l < 4; ++l) rows_max = rows_max > (int)round_up(d.c[l] + 1, 4) ? rows_max : (int)round_up(d.c[l] + 1, 4);
    const size_t shmem = ((size_t)rows_max * CGP_PS + 2 * CGP_PX) * sizeof(float);
    const bool train = h1 != nullptr;
    const bool small = round_up(c0 + 1, 4) <= 96;
#define LLDWT_CGP_DISPATCH(STMT_)                                      \
    if (train && small) { STMT_(true, CGP_NIN_SMALL) }                \
    else if (train) { STMT_(true, CGP_NIN) }                          \
    else if (small) { STMT_(false, CGP_NIN_SMALL) }                   \
    else { STMT_(false, CGP_NIN) }
    if (shmem > 64 * 1024) {
#define LLDWT_CGP_ATTR(T_, N_)                                                                                          \
    if (hipFuncSetAttribute((const void*)k_cgp_rate<T_, N_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem) != \
        hipSuccess) {                                                                                                   \
        set_error("cgp_rate: cannot reserve %zu bytes of LDS", shmem);                                                  \
        return LLDWT_EHIP;                                                                                              \
    }
        LLDWT_CGP_DISPATCH(LLDWT_CGP_ATTR)
#undef LLDWT_CGP_ATTR
    }
    dim3 grid((unsigned)cdiv(cdiv(hw, CGP_PX), CGP_TILES_PER_WG), (unsigned)groups, (unsigned)(planes * batch));
#define LLDWT_CGP_LAUNCH(T_, N_)                                                                                        \
    hipLaunchKernelGGL((k_cgp_rate<T_, N_>), grid, dim3(256), shmem, (hipStream_t)stream, cat, x, noise, packed, bits,  \
                       params_out, bit_sum, d, groups, (int)batch, hw, h1, h2, h3, cx, rows_max);
    LLDWT_CGP_DISPATCH(LLDWT_CGP_LAUNCH)
#undef LLDWT_CGP_LAUNCH
#undef LLDWT_CGP_DISPATCH
    return check_launch("cgp_rate");
}

static inline CgpCtx cgp_no_ctx() {
    CgpCtx cx;
    cx.xq = nullptr; cx.cplc = 0; cx.npatch = 0; cx.w = 1; cx.R = 0;
    for (int i = 0; i < 16; ++i) { cx.tdy[i] = 0; cx.tdx[i] = 0; }
    return cx;
}

extern "C" int lldwt_cgp_rate(const float* cat, const float* x, const float* noise, const float* packed, float* bits,
                              float* params_out, double* bit_sum, int64_t planes, int64_t batch, int64_t hw, int c0,
                              int c1, int c2, int c3, int groups, void* stream) {
    return cgp_rate_impl(cat, x, noise, packed, bits, params_out, nullptr, nullptr, nullptr, bit_sum, planes, batch, hw, c0,
                         c1, c2, c3, groups, cgp_no_ctx(), stream);
}

extern "C" int lldwt_cgp_rate_ctx(const float* plc, const float* xq, const float* x, const float* noise, const float* packed,
                                  float* bits, float* params_out, double* bit_sum, int64_t planes, int64_t batch, int64_t h,
                                  int64_t w_, int cplc, int K, uint32_t tap_mask, int c1, int c2, int c3, int groups,
                                  void* stream) {
    LLDWT_REQUIRE(plc && xq && h > 0 && w_ > 0 && cplc > 0 && (K == 3 || K == 5), "cgp_rate_ctx: bad arguments");
    LLDWT_REQUIRE(h * w_ < ((int64_t)1 << 31), "cgp_rate_ctx: image too large for 32-bit pixel offsets");
    CgpCtx cx = cgp_no_ctx();
    cx.xq = xq; cx.cplc = cplc; cx.w = (int)w_; cx.R = K / 2;
    int n = 0;
    for (int t = 0; t < K * K; ++t)
        if ((tap_mask >> t) & 1u) {
            LLDWT_REQUIRE(n < 16, "cgp_rate_ctx: more than 16 live taps");
            cx.tdy[n] = (int8_t)(t / K);
            cx.tdx[n] = (int8_t)(t % K);
            ++n;
        }
    LLDWT_REQUIRE(n > 0, "cgp_rate_ctx: empty tap mask");
    cx.npatch = n;
    return cgp_rate_impl(plc, x, noise, packed, bits, params_out, nullptr, nullptr, nullptr, bit_sum, planes, batch, h * w_,
                         cplc + n, c1, c2, c3, groups, cx, stream);
}

// lldwt_cgp_rate_train with the context as in lldwt_cgp_rate_ctx: rows < cplc of a group from plc (Z, groups*cplc, h, w), the other
// rows gathered from the quantised subband xq -- no concatenated [plc_g | taps_g] tensor in HBM (1.75 GB at the level-0 shape)
extern "C" int lldwt_cgp_rate_train_ctx(const float* plc, const float* xq, const float* x, const float* noise, const float* packed,
                                        float* bits, float* params_out, float* h1, float* h2, float* h3, int64_t planes,
                                        int64_t batch, int64_t h, int64_t w_, int cplc, int K, uint32_t tap_mask, int c1, int c2,
                                        int c3, int groups, void* stream) {
    LLDWT_REQUIRE(plc && xq && params_out && h1 && h2 && h3 && h > 0 && w_ > 0 && cplc > 0 && (K == 3 || K == 5),
                  "cgp_rate_train_ctx: bad arguments");
    LLDWT_REQUIRE(h * w_ < ((int64_t)1 << 31), "cgp_rate_train_ctx: image too large for 32-bit pixel offsets");
    CgpCtx cx = cgp_no_ctx();
    cx.xq = xq; cx.cplc = cplc; cx.w = (int)w_; cx.R = K / 2;
    int n = 0;
    for (int t = 0; t < K * K; ++t)
        if ((tap_mask >> t) & 1u) {
            LLDWT_REQUIRE(n < 16, "cgp_rate_train_ctx: more than 16 live taps");
            cx.tdy[n] = (int8_t)(t / K);
            cx.tdx[n] = (int8_t)(t % K);
            ++n;
        }
    LLDWT_REQUIRE(n > 0, "cgp_rate_train_ctx: empty tap mask");
    cx.npatch = n;
    return cgp_rate_impl(plc, x, noise, packed, bits, params_out, h1, h2, h3, nullptr, planes, batch, h * w_, cplc + n, c1, c2,
                         c3, groups, cx, stream);
}

extern "C" int lldwt_cgp_rate_train(const float* cat, const float* x, const float* noise, const float* packed, float* bits,
                                    float* params_out, float* h1, float* h2, float* h3, int64_t planes, int64_t batch,
                                    int64_t hw, int c0, int c1, int c2, int c3, int groups, void* stream) {
    LLDWT_REQUIRE(params_out && h1 && h2 && h3, "cgp_rate_train: null output");
    return cgp_rate_impl(cat, x, noise, packed, bits, params_out, h1, h2, h3, nullptr, planes, batch, hw, c0, c1, c2, c3,
                         groups, cgp_no_ctx(), stream);
}

extern "C" int64_t lldwt_cgp_bwd_packed_floats(int c0, int c1, int c2, int c3, int groups) {
    if (c0 <= 0 || c1 <= 0 || c2 <= 0 || c3 <= 0 || groups <= 0) return -1;
    return (int64_t)cgp_dims_bwd(c0, c1, c2, c3).group_floats * groups;
}

static int cgp_bwd_dims_ok(const char* who, int c0, int c1, int c2, int c3, int groups) {
    LLDWT_REQUIRE(groups > 0 && c0 > 0 && c1 > 0 && c2 > 0 && c3 > 0, "%s: bad channel counts", who);
    LLDWT_REQUIRE(c3 < 64 && c2 < 64 && c1 < 64 * CGP_MAXT && c0 < 64 * CGP_MAXT,
                  "%s: widths (%d,%d,%d,%d) exceed the built tile plan", who, c0, c1, c2, c3);
    return 0;
}

extern "C" int lldwt_cgp_pack_bwd(const float* w0, const float* w1, const float* w2, const float* w3, float* packed,
                                  int64_t planes, int c0, int c1, int c2, int c3, int groups, void* stream) {
    int r = cgp_bwd_dims_ok("cgp_pack_bwd", c0, c1, c2, c3, groups);
    if (r) return r;
    LLDWT_REQUIRE(w0 && w1 && w2 && w3 && packed && planes > 0, "cgp_pack_bwd: null pointer");
    const CgpDims d = cgp_dims_bwd(c0, c1, c2, c3);
    dim3 grid((unsigned)cdiv(d.group_floats, 256), (unsigned)groups, (unsigned)planes);
    // backward layer l uses the forward weight of layer 3-l, transposed
    hipLaunchKernelGGL(k_cgp_pack, grid, dim3(256), 0, (hipStream_t)stream, w3, (const float*)nullptr, w2,
                       (const float*)nullptr, w1, (const float*)nullptr, w0, (const float*)nullptr, packed, d, groups, 1);
    return check_launch("cgp_pack_bwd");
}

static int cgp_bwd_impl(const float* dparams, const float* h1, const float* h2, const float* h3, const float* packed_bwd,
                        float* d1, float* d2, float* d3, float* dcat, float* dtaps, int split, int64_t planes, int64_t batch,
                        int64_t hw, int c0, int c1, int c2, int c3, int groups, void* stream);

extern "C" int lldwt_cgp_bwd(const float* dparams, const float* h1, const float* h2, const float* h3, const float* packed_bwd,
                             float* d1, float* d2, float* d3, float* dcat, int64_t planes, int64_t batch, int64_t hw, int c0,
                             int c1, int c2, int c3, int groups, void* stream) {
    return cgp_bwd_impl(dparams, h1, h2, h3, packed_bwd, d1, d2, d3, dcat, nullptr, 0, planes, batch, hw, c0, c1, c2, c3, groups, stream);
}

// lldwt_cgp_bwd with the input gradient split as lldwt_cgp_rate_train_ctx reads the input: dplc (Z, groups*cplc, hw) = the
// tree-context channels of every group, dtaps (Z, groups*ntaps, hw) = the gathered taps (c0 = cplc + ntaps)
extern "C" int lldwt_cgp_bwd_split(const float* dparams, const float* h1, const float* h2, const float* h3, const float* packed_bwd,
                                   float* d1, float* d2, float* d3, float* dplc, float* dtaps, int64_t planes, int64_t batch,
                                   int64_t hw, int cplc, int ntaps, int c1, int c2, int c3, int groups, void* stream) {
    LLDWT_REQUIRE(dtaps && cplc > 0 && ntaps > 0, "cgp_bwd_split: bad arguments");
    return cgp_bwd_impl(dparams, h1, h2, h3, packed_bwd, d1, d2, d3, dplc, dtaps, cplc, planes, batch, hw, cplc + ntaps, c1, c2, c3,
                        groups, stream);
}

static int cgp_bwd_impl(const float* dparams, const float* h1, const float* h2, const float* h3, const float* packed_bwd,
                        float* d1, float* d2, float* d3, float* dcat, float* dtaps, int split, int64_t planes, int64_t batch,
                        int64_t hw, int c0, int c1, int c2, int c3, int groups, void* stream) {
    int r = cgp_bwd_dims_ok("cgp_bwd", c0, c1, c2, c3, groups);
    if (r) return r;
    LLDWT_REQUIRE(dparams && h1 && h2 && h3 && packed_bwd && d1 && d2 && d3 && dcat && planes > 0 && batch > 0 && hw > 0 &&
                      planes * batch <= 65535, "cgp_bwd: bad arguments");
    const CgpDims d = cgp_dims_bwd(c0, c1, c2, c3);
    int rows = 4;
    for (int l = 1; l < 5; ++l) rows = rows > (int)round_up(d.c[l] + 1, 4) ? rows : (int)round_up(d.c[l] + 1, 4);
    const size_t shmem = (size_t)rows * CGP_PS * sizeof(float);
    if (shmem > 64 * 1024) {
        if (hipFuncSetAttribute((const void*)k_cgp_bwd, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem) != hipSuccess) {
            set_error("cgp_bwd: cannot reserve %zu bytes of LDS", shmem);
            return LLDWT_EHIP;
        }
    }
    dim3 grid((unsigned)cdiv(cdiv(hw, CGP_PX), CGP_TILES_PER_WG), (unsigned)groups, (unsigned)(planes * batch));
    hipLaunchKernelGGL(k_cgp_bwd, grid, dim3(256), shmem, (hipStream_t)stream, dparams, h1, h2, h3, packed_bwd, d1, d2, d3,
                       dcat, d, groups, (int)batch, hw, dtaps, split);
    return check_launch("cgp_bwd");
}
