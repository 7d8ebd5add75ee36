// conv_mfma.hip -- fp32 implicit-GEMM convolution on the CDNA4 matrix cores (v_mfma_f32_16x16x4_f32).
//
// One engine for every CNN layer of the hot path (reference call sites: LiftingBasedDWT_net.py:271-289,299-317,793-795
// context models; lifting_dwt_nets.py:139-150 Berk auto-encoder; P_block_v2.py:45-51 the 16->16 kxk lifting convs):
//   D[oc][px] += W[oc][k] * X[k][px],   k = (input channel, tap)
// MFMA operands (MI355X guide: A[l&15][l>>4], B[l>>4][l&15], D col=l&15,row=4*(l>>4)+reg):
//   A = weights  : lane (oc = l&15, kk = l>>4) holds W[oc][ic0+4s+kk][tap]     -- read conflict-free from an LDS image
//                  that was PRE-PACKED in exactly lane order (lldwt_conv_pack), so staging is a straight 16 B/lane copy
//   B = input    : lane (px = l&15, kk = l>>4) holds X[ic0+4s+kk][y+dy][x+px+dx] -- ds_read_b32 from a planar LDS patch
//                  whose plane stride is == 16 (mod 32) dwords: the two 16-lane runs of a 32-lane group hit disjoint banks
// fp32 MFMA is bit-for-bit an fmaf chain (exact f32), so the 1e-4 parity bar of the reference's fp32 path holds.
// A workgroup owns OCT*16 output channels x (TH x TW) pixels of one image; each wave owns WM x WN 16x16 tiles
// (WM*WN*4 accumulator VGPRs), input patch + weight slab are re-staged per chunk of CK input channels.
#include "common.h"

namespace lldwt {

typedef float floatx4 __attribute__((ext_vector_type(4)));

struct ConvPlan {
    int cfg;        // tile configuration index
    int oct;        // 16-channel tiles per oc block
    int ck;         // input channels per chunk
    int ntaps;      // live taps
    int nchunk;     // chunks per group
    int nocb;       // oc blocks per group
    int64_t chunk_floats;   // packed floats per (ocb, chunk)
    int64_t plane_floats;   // packed floats per plane
};

static inline int popc(uint32_t v) { return __builtin_popcount(v); }

// tile configurations: {WM, WN, WVM, WVN, TWS}
//  cfg 0: oc block 16  (1 tile ), px 16x32     cfg 1: 32, px 8x32      cfg 2: 64, px 8x32
//  cfg 3: oc block 96  (6 tiles), px 8x16      cfg 4: 128, px 8x16
static const int kOct[5] = {1, 2, 4, 6, 8};

static inline ConvPlan make_plan(const lldwt_conv_desc& d) {
    ConvPlan p;
    const int cin_g = d.cin / d.groups, cout_g = d.cout / d.groups;
    // padded output channels / relative efficiency of the tile shape (small oc blocks re-stage the input patch more
    // often and amortise fewer MFMAs per LDS read)
    static const double eff[5] = {0.6, 0.75, 0.9, 1.0, 1.0};
    int best = 0;
    double best_cost = 1e30;
    for (int c = 0; c < 5; ++c) {
        const int blk = kOct[c] * 16;
        const double cost = (double)(cdiv(cout_g, blk) * blk) / eff[c];
        if (cost <= best_cost) {   // ties -> larger block
            best_cost = cost;
            best = c;
        }
    }
    p.cfg = best;
    p.oct = kOct[best];
    p.ck = d.K == 1 ? 32 : (cin_g <= 4 ? 4 : 8);   // must match the dispatch in lldwt_conv2d
    p.ntaps = popc(d.tap_mask & ((1u << (d.K * d.K)) - 1u));
    p.nchunk = (int)cdiv(cin_g, p.ck);
    p.nocb = (int)cdiv(cout_g, p.oct * 16);
    p.chunk_floats = (int64_t)p.ntaps * (p.ck / 4) * p.oct * 64;
    p.plane_floats = (int64_t)d.groups * p.nocb * p.nchunk * p.chunk_floats;
    return p;
}

// packed[plane][group][ocb][chunk][live tap][s][m][lane]; lane = kk*16 + ocl
__global__ void k_conv_pack(const float* __restrict__ w, float* __restrict__ packed, lldwt_conv_desc d, ConvPlan p,
                            int swap_hw) {
    const int plane = blockIdx.y;
    const int cin_g = d.cin / d.groups, cout_g = d.cout / d.groups, KK = d.K * d.K;
    const float* wp = w + (int64_t)plane * d.cout * cin_g * KK;
    float* dst = packed + (int64_t)plane * p.plane_floats;
    const int S = p.ck / 4;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < p.plane_floats; i += (int64_t)gridDim.x * blockDim.x) {
        int64_t r = i;
        const int lane = (int)(r % 64); r /= 64;
        const int m = (int)(r % p.oct); r /= p.oct;
        const int s = (int)(r % S); r /= S;
        const int tl = (int)(r % p.ntaps); r /= p.ntaps;
        const int chunk = (int)(r % p.nchunk); r /= p.nchunk;
        const int ocb = (int)(r % p.nocb); r /= p.nocb;
        const int g = (int)r;
        const int ocl = (ocb * p.oct + m) * 16 + (lane & 15);
        const int ic = chunk * p.ck + 4 * s + (lane >> 4);
        // tl-th live tap -> tap index
        int tap = -1, cnt = 0;
        for (int t = 0; t < KK; ++t)
            if ((d.tap_mask >> t) & 1u) {
                if (cnt == tl) { tap = t; break; }
                ++cnt;
            }
        float v = 0.f;
        if (ocl < cout_g && ic < cin_g && tap >= 0) {
            const int oc = g * cout_g + ocl;
            if (swap_hw) tap = (tap % d.K) * d.K + tap / d.K;      // effective tap (dy,dx) reads W[..][dx][dy]
            v = d.transposed ? wp[((int64_t)(g * cin_g + ic) * cout_g + ocl) * KK + (KK - 1 - tap)]   // (cin, cout/groups, K, K)
                             : wp[((int64_t)oc * cin_g + ic) * KK + tap];
        }
        dst[i] = v;
    }
}

struct ConvArgs {
    const float* x;
    float* y;
    const float* packed;
    const float* bias;
    const float* residual;   // same layout as y (ytot channels) or null
    const float* aux;        // same layout as y: forward output for the gradient epilogue, or null
    float* absmax;           // (planes,64) slots receiving max |y| of this launch (feeds the split-fp16 conv), or null
    _Float16* y16;           // fp16 STORAGE of the output instead of y (values multiplied by oscale[plane]), or null
    const float* oscale;     // (planes) power-of-two storage scale for y16
    lldwt_conv_desc d;
    ConvPlan p;
    int batch, h, w, tiles_x, tiles_y;
};

template <int KS, int WM, int WN, int WVM, int WVN, int TWS, int CK>
struct Geo {
    static constexpr int NW = WVM * WVN, NT = 64 * NW;
    static constexpr int OCT = WM * WVM, PXT = WN * WVN;
    static constexpr int TH = PXT / TWS, TW = 16 * TWS;
    static constexpr int R = KS / 2;
    static constexpr int IH = TH + 2 * R, IW = TW + 2 * R;
    static constexpr int PS = ((IH * IW + 15) / 32) * 32 + 16;     // plane stride, == 16 mod 32, >= IH*IW
    static constexpr int S = CK / 4;
    static constexpr int IN_FLOATS = CK * PS;
    static constexpr int NIN = (CK * IH * IW + NT - 1) / NT;             // input floats staged per thread
    static constexpr int NWV = (KS * KS * S * OCT * 16 + NT - 1) / NT;   // weight float4s staged per thread (dense)
};

// Software-pipelined main loop (guide T14, "issue early / write late"): the global loads of chunk c+1 are issued into
// registers before the MFMAs of chunk c and written to LDS after them, so HBM/L2 latency hides under the matrix work.
template <int KS, int WM, int WN, int WVM, int WVN, int TWS, int CK, bool DENSE>
__global__ __launch_bounds__(64 * WVM * WVN) void k_conv_mfma(ConvArgs a) {
    using G = Geo<KS, WM, WN, WVM, WVN, TWS, CK>;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* lin = lds;                 // [CK][PS]
    float* lw = lds + G::IN_FLOATS;   // [ntaps][S][OCT][64]
    int* s_icm = (int*)(lw + a.p.chunk_floats);   // [nchunk*CK] memory channel of input channel ic of this group, or -1
    const lldwt_conv_desc& d = a.d;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WVN, wn = wave % WVN;
    const int px = lane & 15, kk = lane >> 4;
    const int cin_g = d.cin / d.groups, cout_g = d.cout / d.groups;
    const int64_t z = blockIdx.z;
    const int plane = (int)(z / a.batch);
    // launch order: all spatial tiles of one oc block first (blockIdx.y slow), so the workgroups in flight stream the
    // SAME 1.1 MB weight slab from L2.  (An XCD-interleaved order that co-locates the oc blocks of a tile was measured
    // 5 % slower on the 243->243 layer: the weight slab, not the input patch, is the L2 working set that matters.)
    const int g = blockIdx.y / a.p.nocb, ocb = blockIdx.y % a.p.nocb;
    const int ty = blockIdx.x / a.tiles_x, tx = blockIdx.x % a.tiles_x;
    const int y0 = ty * G::TH, x0 = tx * G::TW;
    const int h = a.h, w = a.w;
    const int hi = d.upsample2 ? h >> 1 : h, wi = d.upsample2 ? w >> 1 : w;
    const int64_t hwi = (int64_t)hi * wi;
    const int xtot = d.ic_block > 0 ? d.xtot : d.cin;
    const float* xg = a.x + (z * xtot) * hwi;               // + mapped channel * hwi
    const int icb = d.ic_block > 0 ? d.ic_block : d.cin, ics = d.ic_block > 0 ? d.ic_stride : 0, ico = d.ic_block > 0 ? d.ic_off : 0;
    const float* pk = a.packed + (int64_t)plane * a.p.plane_floats +
                      ((int64_t)(g * a.p.nocb + ocb) * a.p.nchunk) * a.p.chunk_floats;
    const int wvec = (int)(a.p.chunk_floats / 4);
    const uint32_t mask = d.tap_mask;

    floatx4 acc[WM][WN];
#pragma unroll
    for (int m = 0; m < WM; ++m)
#pragma unroll
        for (int n = 0; n < WN; ++n) acc[m][n] = floatx4{0.f, 0.f, 0.f, 0.f};

    // per-thread staging coordinates (chunk independent)
    int in_off[G::NIN];      // global offset inside a channel plane, or -1 (zero padding / outside the tile list)
    int in_lds[G::NIN];      // LDS offset, or -1
    int in_c[G::NIN];
#pragma unroll
    for (int r = 0; r < G::NIN; ++r) {
        const int i = tid + r * G::NT;
        const int c = i / (G::IH * G::IW);
        const int rem = i - c * (G::IH * G::IW);
        const int ly = rem / G::IW, lx = rem - ly * G::IW;
        const int gy = y0 - G::R + ly, gx = x0 - G::R + lx;
        const bool live = i < CK * G::IH * G::IW;
        const bool inimg = gy >= 0 && gy < h && gx >= 0 && gx < w;
        const int sy = d.upsample2 ? gy >> 1 : gy, sx = d.upsample2 ? gx >> 1 : gx;
        in_c[r] = c;
        in_lds[r] = live ? c * G::PS + ly * G::IW + lx : -1;
        in_off[r] = (live && inimg) ? sy * wi + sx : -1;
    }
    float xin[G::NIN];
    float4 wv[G::NWV];
    // channel placement table: the runtime divisions happen once per workgroup instead of once per staged element
    for (int ic = tid; ic < a.p.nchunk * CK; ic += G::NT) {
        int v = -1;
        if (ic < cin_g) {
            const int icg = g * cin_g + ic;
            v = (icg / icb) * ics + ico + icg % icb;
        }
        s_icm[ic] = v;
    }
    __syncthreads();

    // Fast staging path when the input has no channel placement (every layer of the eval path): the chunk base is a
    // wave-uniform pointer (SGPRs), each thread keeps a 32-bit element offset per staged element, and the steady state
    // issues one load + one select per element with no address arithmetic on the vector ALU -- which, on this fp32 path,
    // is the same pipe the MFMAs need.  The last chunk (cin_g % CK channels) and placed inputs take the table path.
    const bool plain_in = d.ic_block == 0;
    unsigned voff[G::NIN];
#pragma unroll
    for (int r = 0; r < G::NIN; ++r)                 // BYTE offsets (< 4 GB: CK channel planes), so they fit the 32-bit voffset
        voff[r] = in_off[r] >= 0 ? 4u * (unsigned)(in_c[r] * (int)hwi + in_off[r]) : 0u;
    const int full_chunks = cin_g / CK;              // chunks whose CK channels all exist

    // (macros, not lambdas: by-reference captures of the register arrays end up in scratch)
    // LOAD only issues the loads (from a safe address where the element is padding); the zeroing happens in STORE, one
    // chunk later: a select right after the load would sit above the sched_barrier that pins the loads over the MFMAs
    // and make every wave wait out the memory latency before its matrix work (seen as vmcnt waits in the .s).
#define LLDWT_STAGE_LOAD(CHUNK)                                                                             \
    {                                                                                                       \
        if (plain_in && (CHUNK) < full_chunks) {                                                            \
            const float* cb = xg + (int64_t)(g * cin_g + (CHUNK) * CK) * hwi;      /* wave-uniform */        \
            _Pragma("unroll") for (int r = 0; r < G::NIN; ++r)                                              \
                xin[r] = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(cb) + voff[r]);      \
        } else {                                                                                            \
            _Pragma("unroll") for (int r = 0; r < G::NIN; ++r) {                                            \
                const int icm = s_icm[(CHUNK) * CK + in_c[r]];                                              \
                xin[r] = xg[(in_off[r] >= 0 && icm >= 0) ? icm * hwi + in_off[r] : 0];                      \
            }                                                                                               \
        }                                                                                                   \
        const float4* src = reinterpret_cast<const float4*>(pk + (int64_t)(CHUNK) * a.p.chunk_floats);     \
        _Pragma("unroll") for (int r = 0; r < G::NWV; ++r) {                                                \
            const int i = tid + r * G::NT;                                                                  \
            const float4 wq = src[i < wvec ? i : 0];                                                        \
            wv[r].x = wq.x; wv[r].y = wq.y; wv[r].z = wq.z; wv[r].w = wq.w;                                 \
        }                                                                                                   \
    }
#define LLDWT_STAGE_STORE(CHUNK)                                                                            \
    {                                                                                                       \
        _Pragma("unroll") for (int r = 0; r < G::NIN; ++r)                                                  \
            if (in_lds[r] >= 0)                                                                             \
                lin[in_lds[r]] = (in_off[r] >= 0 && (CHUNK) * CK + in_c[r] < cin_g) ? xin[r] : 0.f;         \
        float4* dst = reinterpret_cast<float4*>(lw);                                                        \
        _Pragma("unroll") for (int r = 0; r < G::NWV; ++r) {                                                \
            const int i = tid + r * G::NT;                                                                  \
            if (i < wvec) dst[i] = wv[r];                                                                   \
        }                                                                                                   \
    }

    const float* bbase = lin + kk * G::PS + px;                         // + (4s)*PS + row/seg/tap offsets
    const float* abase = lw + (wm * WM) * 64 + lane;                    // + ((tl*S+s)*OCT + m)*64

    LLDWT_STAGE_LOAD(0)
    // (a half-chunk start stagger between co-resident workgroups, keyed on the hardware wave slot, measured 0 +- 0.3 %
    //  on the 243->243 layer -- the workgroups do not run in lockstep -- and is not kept)
    for (int chunk = 0; chunk < a.p.nchunk; ++chunk) {
        __syncthreads();          // all waves are done reading the previous chunk
        LLDWT_STAGE_STORE(chunk)
        __syncthreads();
        if (chunk + 1 < a.p.nchunk) LLDWT_STAGE_LOAD(chunk + 1)
        if constexpr (DENSE) {
            // software-pipelined operand fetch: the LDS reads of k-step i+1 are issued before the MFMAs of k-step i
            // (sched_barrier keeps hipcc from sinking them back next to their use), so lgkmcnt waits are free
            constexpr int NK = KS * KS * G::S;
            float A0[WM], B0[WN], A1[WM], B1[WN];
            auto fetch = [&](int ks, float (&A)[WM], float (&B)[WN]) {
                const int t = ks / G::S, s = ks % G::S;
                const int dy = t / KS, dx = t % KS;
#pragma unroll
                for (int m = 0; m < WM; ++m) A[m] = abase[((t * G::S + s) * G::OCT + m) * 64];
#pragma unroll
                for (int n = 0; n < WN; ++n) {
                    const int j = wn * WN + n;
                    B[n] = bbase[(4 * s) * G::PS + (j / TWS + dy) * G::IW + (j % TWS) * 16 + dx];
                }
            };
            fetch(0, A0, B0);
#pragma unroll
            for (int ks = 0; ks < NK; ks += 2) {
                if (ks + 1 < NK) fetch(ks + 1, A1, B1);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int m = 0; m < WM; ++m)
#pragma unroll
                    for (int n = 0; n < WN; ++n)
                        acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(A0[m], B0[n], acc[m][n], 0, 0, 0);
                if (ks + 1 < NK) {
                    if (ks + 2 < NK) fetch(ks + 2, A0, B0);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int m = 0; m < WM; ++m)
#pragma unroll
                        for (int n = 0; n < WN; ++n)
                            acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(A1[m], B1[n], acc[m][n], 0, 0, 0);
                }
            }
        } else {
            int tl = 0;
#pragma unroll
            for (int t = 0; t < KS * KS; ++t) {
                if (!((mask >> t) & 1u)) continue;
                const int dy = t / KS, dx = t % KS;
                const float* bt = bbase + dy * G::IW + dx;
                const float* at = abase + (tl * G::S) * G::OCT * 64;
#pragma unroll
                for (int s = 0; s < G::S; ++s) {
                    float A[WM], B[WN];
#pragma unroll
                    for (int m = 0; m < WM; ++m) A[m] = at[(s * G::OCT + m) * 64];
#pragma unroll
                    for (int n = 0; n < WN; ++n) {
                        const int j = wn * WN + n;                 // px tile id -> (row, segment)
                        const int row = j / TWS, seg = j % TWS;
                        B[n] = bt[(4 * s) * G::PS + row * G::IW + seg * 16];
                    }
#pragma unroll
                    for (int m = 0; m < WM; ++m)
#pragma unroll
                        for (int n = 0; n < WN; ++n)
                            acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[m], B[n], acc[m][n], 0, 0, 0);
                }
                ++tl;
            }
        }
    }
#undef LLDWT_STAGE_LOAD
#undef LLDWT_STAGE_STORE
    // ---- epilogue: bias, residual, activation, channel placement
    const int64_t hw = (int64_t)h * w;
    float omax = 0.f;
#pragma unroll
    for (int m = 0; m < WM; ++m) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int ocl = (ocb * G::OCT + wm * WM + m) * 16 + kk * 4 + r;
            if (ocl >= cout_g) continue;
            const int oc = g * cout_g + ocl;
            const int ocp = (oc / d.oc_block) * d.oc_stride + d.oc_off + oc % d.oc_block;
            const float bv = a.bias ? a.bias[plane * d.cout + oc] : 0.f;
            float* yp = a.y16 ? nullptr : a.y + (z * d.ytot + ocp) * hw;
            _Float16* yp16 = a.y16 ? a.y16 + (z * d.ytot + ocp) * hw : nullptr;
            const float osc = a.y16 ? a.oscale[plane] : 1.f;
            const float* rp = a.residual ? a.residual + (z * d.ytot + ocp) * hw : nullptr;
            const float* ap = (a.aux && d.epi) ? a.aux + (z * d.ytot + ocp) * hw : nullptr;
#pragma unroll
            for (int n = 0; n < WN; ++n) {
                const int j = wn * WN + n;
                const int gy = y0 + j / TWS, gx = x0 + (j % TWS) * 16 + px;
                if (gy < h && gx < w) {
                    float v = acc[m][n][r] + bv;
                    if (ap) {
                        const float av = ap[(int64_t)gy * w + gx];
                        v *= d.epi == LLDWT_EPI_TANH_BWD ? (1.f - av * av) : (av > 0.f ? 1.f : 0.01f);
                    }
                    if (rp) v += rp[(int64_t)gy * w + gx];
                    v = act_apply(v, d.act);
                    if (yp16) yp16[(int64_t)gy * w + gx] = (_Float16)(v * osc);
                    else yp[(int64_t)gy * w + gx] = v;
                    omax = fmaxf(omax, fabsf(v));
                }
            }
        }
    }
    if (a.absmax) {            // one atomic per wave, spread over 64 slots per plane (values >= 0: int order == float order)
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) omax = fmaxf(omax, __shfl_down(omax, o, 64));
        if (lane == 0 && omax > 0.f)
            atomicMax(reinterpret_cast<int*>(a.absmax + plane * 64 + ((blockIdx.x + wave) & 63)), __float_as_int(omax));
    }
}

template <int KS, int WM, int WN, int WVM, int WVN, int TWS, int CK, bool DENSE>
static int launch_cfg(const ConvArgs& a0, int64_t Z, hipStream_t st) {
    using G = Geo<KS, WM, WN, WVM, WVN, TWS, CK>;
    ConvArgs a = a0;
    a.tiles_x = (int)cdiv(a.w, G::TW);
    a.tiles_y = (int)cdiv(a.h, G::TH);
    const size_t shmem = sizeof(float) * (G::IN_FLOATS + (size_t)a.p.chunk_floats + (size_t)a.p.nchunk * CK);
    auto kern = k_conv_mfma<KS, WM, WN, WVM, WVN, TWS, CK, DENSE>;
    if (shmem > 64 * 1024) {
        if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem) != hipSuccess) {
            set_error("conv2d: cannot reserve %zu bytes of LDS", shmem);
            return LLDWT_EHIP;
        }
    }
    dim3 grid((unsigned)(a.tiles_x * a.tiles_y), (unsigned)(a.d.groups * a.p.nocb), (unsigned)Z);
    hipLaunchKernelGGL(kern, grid, dim3(G::NT), shmem, st, a);
    return check_launch("conv2d(mfma)");
}

template <int KS, int CK, bool DENSE>
static int launch_ks(const ConvArgs& a, int64_t Z, hipStream_t st) {
    switch (a.p.cfg) {
        case 0: return launch_cfg<KS, 1, 8, 1, 4, 2, CK, DENSE>(a, Z, st);
        case 1: return launch_cfg<KS, 2, 4, 1, 4, 2, CK, DENSE>(a, Z, st);
        case 2: return launch_cfg<KS, 4, 4, 1, 4, 2, CK, DENSE>(a, Z, st);
        case 3: return launch_cfg<KS, 3, 4, 2, 2, 1, CK, DENSE>(a, Z, st);
        default:
            // 128 output channels: for 3x3 one 512-thread workgroup (8 waves, 8 rows x 32 pixels) instead of two 256-thread
            // ones (8 x 16 pixels each) -- the 36 KB weight slab of a chunk is staged once for twice the matrix work and
            // the halo overhead drops from 1.41x to 1.33x: 115 -> 120 TFLOP/s on the 243 -> 243 tree conv
            if constexpr (KS == 3) return launch_cfg<KS, 4, 4, 2, 4, 2, CK, DENSE>(a, Z, st);
            return launch_cfg<KS, 4, 4, 2, 2, 1, CK, DENSE>(a, Z, st);
    }
}

static int conv_desc_ok(const char* who, const lldwt_conv_desc* d, int64_t planes, int64_t batch, int64_t h, int64_t w) {
    LLDWT_REQUIRE(d, "%s: null descriptor", who);
    LLDWT_REQUIRE(d->K == 1 || d->K == 3 || d->K == 5, "%s: K=%d unsupported", who, d->K);
    LLDWT_REQUIRE(d->groups > 0 && d->cin > 0 && d->cout > 0 && d->cin % d->groups == 0 && d->cout % d->groups == 0,
                  "%s: bad channels/groups (%d,%d,%d)", who, d->cin, d->cout, d->groups);
    LLDWT_REQUIRE(d->ic_block == 0 || (d->ic_block > 0 && d->xtot >= d->cin), "%s: bad input placement", who);
    LLDWT_REQUIRE(planes > 0 && batch > 0 && h > 0 && w > 0 && planes * batch <= 65535, "%s: bad planes/batch/h/w", who);
    LLDWT_REQUIRE(!d->upsample2 || (h % 2 == 0 && w % 2 == 0), "%s: upsample2 needs even output dims", who);
    LLDWT_REQUIRE(d->oc_block > 0 && d->ytot >= d->cout, "%s: bad output placement", who);
    LLDWT_REQUIRE((d->tap_mask & ((1u << (d->K * d->K)) - 1u)) != 0, "%s: empty tap mask", who);
    return 0;
}

}  // namespace lldwt
using namespace lldwt;

static int conv2d_launch_impl(ConvArgs& a, const lldwt_conv_desc* d, int64_t planes, int64_t batch, int64_t h, int64_t w_,
                              void* stream);
static int conv2d_launch(ConvArgs& a, const lldwt_conv_desc* d, int64_t planes, int64_t batch, int64_t h, int64_t w_,
                         void* stream) {
    return conv2d_launch_impl(a, d, planes, batch, h, w_, stream);
}

extern "C" int64_t lldwt_conv_packed_floats(const lldwt_conv_desc* d) {
    if (!d || d->groups <= 0 || d->cin % d->groups || d->cout % d->groups) return -1;
    return make_plan(*d).plane_floats;
}

extern "C" int lldwt_conv_pack(const float* w, float* packed, const lldwt_conv_desc* d, int64_t planes, void* stream) {
    return lldwt_conv_pack_ex(w, packed, d, planes, 0, stream);
}

extern "C" int lldwt_conv_pack_ex(const float* w, float* packed, const lldwt_conv_desc* d, int64_t planes, int swap_hw,
                                  void* stream) {
    int r = conv_desc_ok("conv_pack", d, planes, 1, 2, 2);
    if (r) return r;
    LLDWT_REQUIRE(w && packed, "conv_pack: null pointer");
    const ConvPlan p = make_plan(*d);
    int64_t gx = cdiv(p.plane_floats, 256);
    if (gx > 4096) gx = 4096;
    hipLaunchKernelGGL(k_conv_pack, dim3((unsigned)gx, (unsigned)planes), dim3(256), 0, (hipStream_t)stream, w, packed, *d, p,
                       swap_hw);
    return check_launch("conv_pack");
}

extern "C" int lldwt_conv2d(const float* x, float* y, const float* packed, const float* bias, const float* residual,
                            const float* aux, const lldwt_conv_desc* d, int64_t planes, int64_t batch, int64_t h,
                            int64_t w_, void* stream) {
    return lldwt_conv2d_absmax(x, y, packed, bias, residual, aux, nullptr, d, planes, batch, h, w_, stream);
}

extern "C" int lldwt_conv2d_absmax(const float* x, float* y, const float* packed, const float* bias, const float* residual,
                                   const float* aux, float* absmax_slots, const lldwt_conv_desc* d, int64_t planes,
                                   int64_t batch, int64_t h, int64_t w_, void* stream) {
    int r = conv_desc_ok("conv2d", d, planes, batch, h, w_);
    if (r) return r;
    LLDWT_REQUIRE(x && y && packed, "conv2d: null pointer");
    if (absmax_slots && hipMemsetAsync(absmax_slots, 0, sizeof(float) * 64 * planes, (hipStream_t)stream) != hipSuccess) {
        set_error("conv2d: memset of the absmax slots failed");
        return LLDWT_EHIP;
    }
    ConvArgs a;
    a.x = x; a.y = y; a.packed = packed; a.bias = bias; a.residual = residual; a.aux = aux; a.absmax = absmax_slots;
    a.y16 = nullptr; a.oscale = nullptr;
    return conv2d_launch(a, d, planes, batch, h, w_, stream);
}

extern "C" int lldwt_conv2d_f16out(const float* x, void* y16, const float* packed, const float* bias, const float* oscale,
                                   const lldwt_conv_desc* d, int64_t planes, int64_t batch, int64_t h, int64_t w_,
                                   void* stream) {
    int r = conv_desc_ok("conv2d_f16out", d, planes, batch, h, w_);
    if (r) return r;
    LLDWT_REQUIRE(x && y16 && packed && oscale, "conv2d_f16out: null pointer");
    ConvArgs a;
    a.x = x; a.y = nullptr; a.packed = packed; a.bias = bias; a.residual = nullptr; a.aux = nullptr; a.absmax = nullptr;
    a.y16 = reinterpret_cast<_Float16*>(y16); a.oscale = oscale;
    return conv2d_launch(a, d, planes, batch, h, w_, stream);
}

static int conv2d_launch_impl(ConvArgs& a, const lldwt_conv_desc* d, int64_t planes, int64_t batch, int64_t h, int64_t w_,
                              void* stream) {
    a.d = *d;
    a.d.tap_mask &= (1u << (d->K * d->K)) - 1u;
    a.p = make_plan(a.d);
    a.batch = (int)batch; a.h = (int)h; a.w = (int)w_; a.tiles_x = 0; a.tiles_y = 0;
    const int64_t Z = planes * batch;
    hipStream_t st = (hipStream_t)stream;
    const bool dense = a.p.ntaps == d->K * d->K;
    if (d->K == 1) return launch_ks<1, 32, true>(a, Z, st);
    if (d->K == 3) {
        if (a.p.ck == 4) return dense ? launch_ks<3, 4, true>(a, Z, st) : launch_ks<3, 4, false>(a, Z, st);
        return dense ? launch_ks<3, 8, true>(a, Z, st) : launch_ks<3, 8, false>(a, Z, st);
    }
    if (a.p.ck == 4) return dense ? launch_ks<5, 4, true>(a, Z, st) : launch_ks<5, 4, false>(a, Z, st);
    return dense ? launch_ks<5, 8, true>(a, Z, st) : launch_ks<5, 8, false>(a, Z, st);
}
