// lifting.hip -- learned lifting DWT for gfx950 (CDNA4).
//
// One lifting step (reference: graphs/layers/wavelet_forward_v2.py:58-74, wavelet_inverse_v2.py:76-90,
// graphs/layers/P_block_v2.py:40-55) is three tiled launches with no recomputed halo work:
//   A: src tile (+halo) -> 3-tap skip filter -> conv1 -> tanh (LDS) -> conv2 -> tanh        -> t2 (C ch, HBM/L2)
//   B: t2 tile (+halo) (LDS) -> conv3, + conv1 pre-activation recomputed from the skip tile  -> t3 (C ch)
//   C: t3 tile (+halo) (LDS) -> conv4 ; dst_out = dst_in + sign*(skip + res_weight*net)
// The polyphase split / merge is pure addressing through lldwt_view (no transposes, no copies); the horizontal
// pass uses the (kh,kw)-transposed weights instead of transposing the data.
#include "common.h"
#include "lifting_f16.h"
#include <string.h>

namespace lldwt {

static int g_lift_mode = 1;      // 1: fused split-fp16 step kernel where it applies (eval, C=16, K=5, tanh); 0: fp32 MFMA kernels

static inline __host__ __device__ int pad16(int n) { return (n + 15) & ~15; }

struct PackOff {
    int w1, b1, w2, b2, w3, b3, w4, b4, orient, f16, total;   // f16: start of the split-fp16 section (lifting_f16.h)
};
static inline __host__ __device__ PackOff pack_off(int C, int K) {
    PackOff o;
    int KK = K * K;
    o.w1 = 0;
    o.b1 = o.w1 + pad16(KK * C);
    o.w2 = o.b1 + pad16(C);
    o.b2 = o.w2 + pad16(C * KK * C);
    o.w3 = o.b2 + pad16(C);
    o.b3 = o.w3 + pad16(C * KK * C);
    o.w4 = o.b3 + pad16(C);
    o.b4 = o.w4 + pad16(C * KK);
    o.orient = o.b4 + 16;
    o.f16 = 2 * o.orient;
    o.total = o.f16 + lift_f16_floats(C, K);
    return o;
}

// packed layouts (per orientation): W1[tap][oc], W2/W3[ic][tap][oc], W4[ic][tap]; tap = dy*K+dx in the EFFECTIVE
// orientation: vertical uses w[..][dy][dx], horizontal uses w[..][dx][dy].
__global__ void k_pack_pblock(const float* __restrict__ w1, const float* __restrict__ b1, const float* __restrict__ w2,
                              const float* __restrict__ b2, const float* __restrict__ w3, const float* __restrict__ b3,
                              const float* __restrict__ w4, const float* __restrict__ b4, float* __restrict__ packed,
                              int C, int K) {
    const PackOff o = pack_off(C, K);
    const int KK = K * K;
    const int plane = blockIdx.y;
    float* dst = packed + (int64_t)plane * o.total;
    w1 += (int64_t)plane * C * KK;
    w2 += (int64_t)plane * C * C * KK;
    w3 += (int64_t)plane * C * C * KK;
    w4 += (int64_t)plane * C * KK;
    b1 += plane * C;
    b2 += plane * C;
    b3 += plane * C;
    b4 += plane;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < o.f16; i += gridDim.x * blockDim.x) {
        int orient = i / o.orient;
        int j = i - orient * o.orient;
        float v = 0.f;
        auto srctap = [&](int tap) {
            int dy = tap / K, dx = tap % K;
            return orient == 0 ? dy * K + dx : dx * K + dy;
        };
        if (j < o.b1) {
            if (j < KK * C) { int tap = j / C, oc = j % C; v = w1[oc * KK + srctap(tap)]; }
        } else if (j < o.w2) {
            if (j - o.b1 < C) v = b1[j - o.b1];
        } else if (j < o.b2) {
            int q = j - o.w2;
            if (q < C * KK * C) { int ic = q / (KK * C), tap = (q / C) % KK, oc = q % C; v = w2[(oc * C + ic) * KK + srctap(tap)]; }
        } else if (j < o.w3) {
            if (j - o.b2 < C) v = b2[j - o.b2];
        } else if (j < o.b3) {
            int q = j - o.w3;
            if (q < C * KK * C) { int ic = q / (KK * C), tap = (q / C) % KK, oc = q % C; v = w3[(oc * C + ic) * KK + srctap(tap)]; }
        } else if (j < o.w4) {
            if (j - o.b3 < C) v = b3[j - o.b3];
        } else if (j < o.b4) {
            int q = j - o.w4;
            if (q < C * KK) { int ic = q / KK, tap = q % KK; v = w4[ic * KK + srctap(tap)]; }
        } else {
            if (j == o.b4) v = b4[0];
        }
        dst[i] = v;
    }
}

struct CView {
    const float* p;
    int64_t sz, sy, sx;
};

constexpr int TH = 16, TW = 32, NT = 256;

__device__ __forceinline__ float ld_view(const CView& v, int64_t z, int y, int x, int h, int w) {
    return (y >= 0 && y < h && x >= 0 && x < w) ? v.p[z * v.sz + (int64_t)y * v.sy + (int64_t)x * v.sx] : 0.f;
}

// C-channel KxK conv of an LDS tile for 2 pixels per thread (rows ly and ly+TH/2), all C output channels.
template <int C, int K, int PITCH, int ROWS>
__device__ __forceinline__ void conv_cc(const float (*__restrict__ t)[ROWS][PITCH], const float* __restrict__ W,
                                        int ly, int lx, float (&acc0)[C], float (&acc1)[C]) {
    constexpr int KK = K * K;
    for (int ic = 0; ic < C; ++ic) {
        const float* wi = W + ic * KK * C;
#pragma unroll
        for (int dy = 0; dy < K; ++dy) {
#pragma unroll
            for (int dx = 0; dx < K; ++dx) {
                const float v0 = t[ic][ly + dy][lx + dx];
                const float v1 = t[ic][ly + TH / 2 + dy][lx + dx];
                const float* wt = wi + (dy * K + dx) * C;
#pragma unroll
                for (int oc = 0; oc < C; ++oc) {
                    const float wv = wt[oc];   // wave-uniform address -> scalar load
                    acc0[oc] = fmaf(wv, v0, acc0[oc]);
                    acc1[oc] = fmaf(wv, v1, acc1[oc]);
                }
            }
        }
    }
}

// ---- kernel A: skip filter, conv1+act (LDS), conv2+act -> t2; also stores skip --------------------------------
template <int C, int K>
__global__ __launch_bounds__(NT) void k_lift_a(CView src, float* __restrict__ skip_out, float* __restrict__ t2_out,
                                               float* __restrict__ t1_out, float* __restrict__ src_out,
                                               int batch, int h, int w, const float* __restrict__ taps,
                                               const float* __restrict__ packed, int64_t packed_plane_stride,
                                               int vertical, int linear) {
    constexpr int R = K / 2, R2 = 2 * R, KK = K * K;
    constexpr int SH = TH + 2 * R2, SW = TW + 2 * R2;
    constexpr int T1H = TH + 2 * R, T1W = TW + 2 * R, T1P = T1W + 1;
    __shared__ float s_lds[SH][SW + 1];
    __shared__ float t1[C][T1H][T1P];
    const PackOff o = pack_off(C, K);
    const int tid = threadIdx.x;
    const int64_t z = blockIdx.z;
    const int plane = (int)(z / batch);
    const int y0 = blockIdx.y * TH, x0 = blockIdx.x * TW;
    const float* pk = packed + plane * packed_plane_stride + (vertical ? 0 : o.orient);
    const float tp0 = taps[plane * 3 + 0], tp1 = taps[plane * 3 + 1], tp2 = taps[plane * 3 + 2];
    const int act = linear ? LLDWT_ACT_NONE : LLDWT_ACT_TANH;

    for (int i = tid; i < SH * SW; i += NT) {
        const int ly = i / SW, lx = i - ly * SW;
        const int gy = y0 - R2 + ly, gx = x0 - R2 + lx;
        float v = 0.f;
        if (gy >= 0 && gy < h && gx >= 0 && gx < w) {
            const int ddy = vertical ? 1 : 0, ddx = vertical ? 0 : 1;
            const float a = ld_view(src, z, gy - ddy, gx - ddx, h, w);
            const float b = ld_view(src, z, gy, gx, h, w);
            const float c = ld_view(src, z, gy + ddy, gx + ddx, h, w);
            v = tp0 * a + tp1 * b + tp2 * c;
            if (ly >= R2 && ly < R2 + TH && lx >= R2 && lx < R2 + TW) {
                skip_out[(z * h + gy) * (int64_t)w + gx] = v;
                if (src_out) src_out[(z * h + gy) * (int64_t)w + gx] = b;     // training: kept for d(taps)
            }
        }
        s_lds[ly][lx] = v;
    }
    __syncthreads();
    // conv1 (1 -> C) + act on the (TH+2R)x(TW+2R) region; zero outside the image (conv2's zero padding)
    for (int i = tid; i < T1H * T1W; i += NT) {
        const int ly = i / T1W, lx = i - ly * T1W;
        const int gy = y0 - R + ly, gx = x0 - R + lx;
        float acc[C];
        const bool in = gy >= 0 && gy < h && gx >= 0 && gx < w;
#pragma unroll
        for (int oc = 0; oc < C; ++oc) acc[oc] = pk[o.b1 + oc];
        if (in) {
#pragma unroll
            for (int dy = 0; dy < K; ++dy)
#pragma unroll
                for (int dx = 0; dx < K; ++dx) {
                    const float v = s_lds[ly + dy][lx + dx];
                    const float* wt = pk + o.w1 + (dy * K + dx) * C;
#pragma unroll
                    for (int oc = 0; oc < C; ++oc) acc[oc] = fmaf(wt[oc], v, acc[oc]);
                }
        }
        const bool centre = t1_out && in && ly >= R && ly < R + TH && lx >= R && lx < R + TW;
#pragma unroll
        for (int oc = 0; oc < C; ++oc) {
            const float tv = in ? act_apply(acc[oc], act) : 0.f;
            t1[oc][ly][lx] = tv;
            if (centre) t1_out[(z * C + oc) * ((int64_t)h * w) + (int64_t)gy * w + gx] = tv;   // training: kept for backward
        }
    }
    __syncthreads();
    const int ly = tid / TW, lx = tid % TW;
    float acc0[C], acc1[C];
#pragma unroll
    for (int oc = 0; oc < C; ++oc) acc0[oc] = acc1[oc] = pk[o.b2 + oc];
    conv_cc<C, K, T1P, T1H>(t1, pk + o.w2, ly, lx, acc0, acc1);
    const int gx = x0 + lx;
    const int gy0 = y0 + ly, gy1 = y0 + ly + TH / 2;
    if (gx < w) {
        const int64_t cs = (int64_t)h * w;
        float* b0 = t2_out + (z * C) * cs + (int64_t)gy0 * w + gx;
        float* b1p = t2_out + (z * C) * cs + (int64_t)gy1 * w + gx;
#pragma unroll
        for (int oc = 0; oc < C; ++oc) {
            if (gy0 < h) b0[oc * cs] = act_apply(acc0[oc], act);
            if (gy1 < h) b1p[oc * cs] = act_apply(acc1[oc], act);
        }
    }
    (void)KK;
}

// ---- kernel B: conv3(t2) + conv1(skip) pre-activation -> t3 -------------------------------------------------
template <int C, int K>
__global__ __launch_bounds__(NT) void k_lift_b(const float* __restrict__ skip, const float* __restrict__ t2,
                                               float* __restrict__ t3_out, int batch, int h, int w,
                                               const float* __restrict__ packed, int64_t packed_plane_stride,
                                               int vertical) {
    constexpr int R = K / 2;
    constexpr int T1H = TH + 2 * R, T1W = TW + 2 * R, T1P = T1W + 1;
    __shared__ float s_lds[T1H][T1W + 1];
    __shared__ float t[C][T1H][T1P];
    const PackOff o = pack_off(C, K);
    const int tid = threadIdx.x;
    const int64_t z = blockIdx.z;
    const int plane = (int)(z / batch);
    const int y0 = blockIdx.y * TH, x0 = blockIdx.x * TW;
    const float* pk = packed + plane * packed_plane_stride + (vertical ? 0 : o.orient);
    const int64_t cs = (int64_t)h * w;
    for (int i = tid; i < T1H * T1W; i += NT) {
        const int ly = i / T1W, lx = i - ly * T1W;
        const int gy = y0 - R + ly, gx = x0 - R + lx;
        const bool in = gy >= 0 && gy < h && gx >= 0 && gx < w;
        const int64_t off = (int64_t)gy * w + gx;
        s_lds[ly][lx] = in ? skip[z * cs + off] : 0.f;
#pragma unroll
        for (int c = 0; c < C; ++c) t[c][ly][lx] = in ? t2[(z * C + c) * cs + off] : 0.f;
    }
    __syncthreads();
    const int ly = tid / TW, lx = tid % TW;
    float acc0[C], acc1[C];
#pragma unroll
    for (int oc = 0; oc < C; ++oc) acc0[oc] = acc1[oc] = pk[o.b3 + oc] + pk[o.b1 + oc];
#pragma unroll
    for (int dy = 0; dy < K; ++dy)
#pragma unroll
        for (int dx = 0; dx < K; ++dx) {
            const float v0 = s_lds[ly + dy][lx + dx], v1 = s_lds[ly + TH / 2 + dy][lx + dx];
            const float* wt = pk + o.w1 + (dy * K + dx) * C;
#pragma unroll
            for (int oc = 0; oc < C; ++oc) {
                acc0[oc] = fmaf(wt[oc], v0, acc0[oc]);
                acc1[oc] = fmaf(wt[oc], v1, acc1[oc]);
            }
        }
    conv_cc<C, K, T1P, T1H>(t, pk + o.w3, ly, lx, acc0, acc1);
    const int gx = x0 + lx;
    const int gy0 = y0 + ly, gy1 = y0 + ly + TH / 2;
    if (gx < w) {
        float* b0 = t3_out + (z * C) * cs + (int64_t)gy0 * w + gx;
        float* b1p = t3_out + (z * C) * cs + (int64_t)gy1 * w + gx;
#pragma unroll
        for (int oc = 0; oc < C; ++oc) {
            if (gy0 < h) b0[oc * cs] = acc0[oc];
            if (gy1 < h) b1p[oc * cs] = acc1[oc];
        }
    }
}

// ---- MFMA variants of kernels A and B for C == 16 (the reference's depth_scale*8, liftingDWT.json:22) ------------------
// The 16->16 kxk convs are implicit GEMMs on v_mfma_f32_16x16x4_f32 (exact fp32): D[oc][px] += W[oc][(ic,tap)] X[(ic,tap)][px].
//   A = weights, staged once per workgroup into LDS in lane order [tap][s][64] (lane = kk*16 + oc, ic = 4s + kk)
//   B = activation tile in LDS, planar [ic][rows][cols] with plane stride == 16 (mod 32) dwords (20*36 = 720)
// A wave owns 8 pixel tiles (4 rows x 2 segments of 16) of the 16x32 output tile: 8 accumulators, 1 A read per 8 MFMAs.
typedef float floatx4 __attribute__((ext_vector_type(4)));

template <int K>
__device__ __forceinline__ void stage_w16(float* __restrict__ wl, const float* __restrict__ Wp, int tid) {
    // Wp: packed [ic][tap][oc] (pack_off layout) -> wl[(tap*4 + s)*64 + kk*16 + oc], ic = 4s + kk
    constexpr int KK = K * K;
    for (int i = tid; i < KK * 4 * 64; i += NT) {
        const int l = i & 63, s = (i >> 6) & 3, tap = i >> 8;
        const int oc = l & 15, kk = l >> 4;
        wl[i] = Wp[((4 * s + kk) * KK + tap) * 16 + oc];
    }
}

// ---- latency-friendly staging: every global load of a thread is issued before its first LDS store ------------------------
// (a load -> store loop waits vmcnt(0) per element; the first version of kernel B spent more time in ~24 serialised
// round trips than in its 800 MFMAs).  Rows come in as dwordx4, the weights as dwordx4 in their final lane order.
struct __attribute__((packed, aligned(4))) f4u { float x, y, z, w; };

template <int K>
struct Tile16 {
    static constexpr int R = K / 2, T1H = TH + 2 * R, T1W = TW + 2 * R;
    static constexpr int NVR = (T1W + 3) / 4;                 // vectors per tile row
    static constexpr int NV = 16 * T1H * NVR;                 // vectors per 16-channel tile
    static constexpr int NVT = (NV + NT - 1) / NT;            // per thread
    static constexpr int NWV = (K * K * 64 + NT - 1) / NT;    // weight vectors per thread (KK*4*64 floats)
};

// 16-channel tile with halo R of a dense (Z,16,h,w) tensor; zero outside the image.  Two phases so that every load is in
// flight before anything consumes one: tile16_issue only issues the dwordx4 loads (from a safe address when the vector is
// not fully inside the image); tile16_fix, called once the other loads of the prologue have been issued too, zeroes the
// outside vectors and patches the ones that straddle the left / right border pixel by pixel.
template <int K>
__device__ __forceinline__ void tile16_issue(f4u (&v)[Tile16<K>::NVT], const float* __restrict__ base, int64_t cs, int y0,
                                             int x0, int h, int w, int tid) {
    using G = Tile16<K>;
#pragma unroll
    for (int r = 0; r < G::NVT; ++r) {
        const int i = tid + r * NT;
        const int c = i / (G::T1H * G::NVR), rem = i - c * (G::T1H * G::NVR);
        const int gy = y0 - G::R + rem / G::NVR, gx = x0 - G::R + 4 * (rem % G::NVR);
        const bool ok4 = i < G::NV && gy >= 0 && gy < h && gx >= 0 && gx + 3 < w;
        // 32-bit element offset from a wave-uniform base (one image's 16 channels fit): SGPR base + VGPR offset addressing,
        // no 64-bit multiplies per load
        const unsigned off = ok4 ? (unsigned)(c * (int)cs + gy * w + gx) : 0u;
        v[r] = *reinterpret_cast<const f4u*>(base + off);
    }
}
template <int K>
__device__ __forceinline__ void tile16_fix(floatx4 (&o)[Tile16<K>::NVT], const f4u (&v)[Tile16<K>::NVT],
                                           const float* __restrict__ base, int64_t cs, int y0, int x0, int h, int w, int tid) {
    using G = Tile16<K>;
#pragma unroll
    for (int r = 0; r < G::NVT; ++r) {
        const int i = tid + r * NT;
        const int c = i / (G::T1H * G::NVR), rem = i - c * (G::T1H * G::NVR);
        const int gy = y0 - G::R + rem / G::NVR, gx = x0 - G::R + 4 * (rem % G::NVR);
        const bool row = i < G::NV && gy >= 0 && gy < h;
        const bool ok4 = row && gx >= 0 && gx + 3 < w;
        o[r] = ok4 ? floatx4{v[r].x, v[r].y, v[r].z, v[r].w} : floatx4{0.f, 0.f, 0.f, 0.f};
        if (row && !ok4 && gx + 3 >= 0 && gx < w) {           // vector straddles the left / right image border
            const int off = c * (int)cs + gy * w + gx;
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (gx + e >= 0 && gx + e < w) o[r][e] = base[off + e];
        }
    }
}
#define LLDWT_TILE16(K_, tv_, base_, cs_, y0_, x0_, h_, w_, tid_)  /* declare + issue; LLDWT_TILE16_FIX completes */ \
    f4u tv_##_raw[Tile16<K_>::NVT];                                                                               \
    floatx4 tv_[Tile16<K_>::NVT];                                                                                 \
    tile16_issue<K_>(tv_##_raw, base_, cs_, y0_, x0_, h_, w_, tid_);
#define LLDWT_TILE16_FIX(K_, tv_, base_, cs_, y0_, x0_, h_, w_, tid_) \
    tile16_fix<K_>(tv_, tv_##_raw, base_, cs_, y0_, x0_, h_, w_, tid_);

// -> LDS planes of stride PS, rows of PITCH floats
template <int K, int PS, int PITCH>
__device__ __forceinline__ void tile16_store(float* __restrict__ t, const floatx4 (&v)[Tile16<K>::NVT], int tid) {
    using G = Tile16<K>;
#pragma unroll
    for (int r = 0; r < G::NVT; ++r) {
        const int i = tid + r * NT;
        if (i < G::NV) {
            const int c = i / (G::T1H * G::NVR), rem = i - c * (G::T1H * G::NVR);
            const int lx = 4 * (rem % G::NVR);
            float* d = t + c * PS + (rem / G::NVR) * PITCH + lx;
            if constexpr (PS % 2 == 0 && PITCH % 2 == 0) {
                if (lx + 1 < G::T1W) *reinterpret_cast<float2*>(d) = float2{v[r][0], v[r][1]};
                if (lx + 3 < G::T1W) *reinterpret_cast<float2*>(d + 2) = float2{v[r][2], v[r][3]};
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (lx + e < G::T1W) d[e] = v[r][e];
            }
        }
    }
}

// fix + store fused per vector (keeps only the raw registers live across the call)
template <int K, int PS, int PITCH>
__device__ __forceinline__ void tile16_fix_store(float* __restrict__ t, const f4u (&v)[Tile16<K>::NVT],
                                                 const float* __restrict__ base, int64_t cs, int y0, int x0, int h, int w,
                                                 int tid) {
    using G = Tile16<K>;
#pragma unroll
    for (int r = 0; r < G::NVT; ++r) {
        const int i = tid + r * NT;
        const int c = i / (G::T1H * G::NVR), rem = i - c * (G::T1H * G::NVR);
        const int lx = 4 * (rem % G::NVR);
        const int gy = y0 - G::R + rem / G::NVR, gx = x0 - G::R + lx;
        const bool row = i < G::NV && gy >= 0 && gy < h;
        const bool ok4 = row && gx >= 0 && gx + 3 < w;
        floatx4 o = ok4 ? floatx4{v[r].x, v[r].y, v[r].z, v[r].w} : floatx4{0.f, 0.f, 0.f, 0.f};
        if (row && !ok4 && gx + 3 >= 0 && gx < w) {
            const int off = c * (int)cs + gy * w + gx;
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (gx + e >= 0 && gx + e < w) o[e] = base[off + e];
        }
        if (i < G::NV) {
            float* d = t + c * PS + (rem / G::NVR) * PITCH + lx;
            if constexpr (PS % 2 == 0 && PITCH % 2 == 0) {
                if (lx + 1 < G::T1W) *reinterpret_cast<float2*>(d) = float2{o[0], o[1]};
                if (lx + 3 < G::T1W) *reinterpret_cast<float2*>(d + 2) = float2{o[2], o[3]};
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (lx + e < G::T1W) d[e] = o[e];
            }
        }
    }
}

// 16 -> 16 weights: Wp = forward pack [ic][tap][oc]; wl[(tap*4 + s)*64 + kk*16 + oc], ic = 4s + kk (4 oc per vector)
template <int K>
__device__ __forceinline__ void w16_load(floatx4 (&wv)[Tile16<K>::NWV], const float* __restrict__ Wp, int tid) {
    constexpr int KK = K * K;
#pragma unroll
    for (int r = 0; r < Tile16<K>::NWV; ++r) {
        const int i = tid + r * NT;
        const int oc4 = i & 3, kk = (i >> 2) & 3, s = (i >> 4) & 3, tap = i >> 6;
        const float4 q = *reinterpret_cast<const float4*>(Wp + (i < KK * 64 ? ((4 * s + kk) * KK + tap) * 16 + 4 * oc4 : 0));
        wv[r] = floatx4{q.x, q.y, q.z, q.w};
    }
}
template <int K>
__device__ __forceinline__ void w16_store(float* __restrict__ wl, const floatx4 (&wv)[Tile16<K>::NWV], int tid) {
#pragma unroll
    for (int r = 0; r < Tile16<K>::NWV; ++r) {
        const int i = tid + r * NT;
        if (i < K * K * 64) *reinterpret_cast<floatx4*>(wl + 4 * i) = wv[r];        // 4*i == (tap*4+s)*64 + kk*16 + 4*oc4
    }
}
// transposed + mirrored (backward-data): wl[(tap'*4 + s)*64 + kk*16 + oc'] = W[oc = 4s+kk][ic = oc'][KK-1-tap']
// (4 kk per vector: the source run over the forward oc)
template <int K>
__device__ __forceinline__ void w16T_load(floatx4 (&wv)[Tile16<K>::NWV], const float* __restrict__ Wp, int tid) {
    constexpr int KK = K * K;
#pragma unroll
    for (int r = 0; r < Tile16<K>::NWV; ++r) {
        const int i = tid + r * NT;
        const int oc = i & 15, s = (i >> 4) & 3, tap = i >> 6;
        const float4 q = *reinterpret_cast<const float4*>(Wp + (i < KK * 64 ? (oc * KK + (KK - 1 - tap)) * 16 + 4 * s : 0));
        wv[r] = floatx4{q.x, q.y, q.z, q.w};
    }
}
template <int K>
__device__ __forceinline__ void w16T_store(float* __restrict__ wl, const floatx4 (&wv)[Tile16<K>::NWV], int tid) {
#pragma unroll
    for (int r = 0; r < Tile16<K>::NWV; ++r) {
        const int i = tid + r * NT;
        if (i < K * K * 64) {
            const int oc = i & 15, s = (i >> 4) & 3, tap = i >> 6;
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) wl[(tap * 4 + s) * 64 + kk * 16 + oc] = wv[r][kk];
        }
    }
}

// acc[n] += conv over the 16-channel LDS tile t (planes of stride PS, rows of PITCH) for this wave's 8 pixel tiles.
// The PIXELS are the MFMA's A operand (rows) and the weights its B operand (columns): D[row = pixel][col = oc], so a lane
// ends up with 4 consecutive pixels (4*kk .. 4*kk+3 of the tile's 16) of ONE channel (oc = lane & 15) -- the epilogues
// read and write 16 bytes per lane instead of 4 scalars in 4 different channel planes.
// Software-pipelined like the conv engine: the 9 LDS operands of k-step i+1 are requested before the 8 MFMAs of k-step i
// are issued (sched_barrier keeps hipcc from sinking the reads back next to their use).  Measured alone on a CU, the
// unpipelined loop ran at 47 cycles per MFMA instead of 32.
template <int K, int PS, int PITCH>
__device__ __forceinline__ void conv16_mfma_ps(const float* __restrict__ t, const float* __restrict__ wl, int wave,
                                               int lane, floatx4 (&acc)[8]) {
    const int px = lane & 15, kk = lane >> 4;
    const float* tb = t + kk * PS + (wave * 4) * PITCH + px;
    const float* wa = wl + lane;
    constexpr int NS = K * 4;                       // k-steps per kernel row: (dx, s)
#define LLDWT_C16_FETCH(A_, B_, i_)                                                                              \
    {                                                                                                            \
        const int dx_ = (i_) / 4, s_ = (i_) % 4;                                                                 \
        A_ = wrow[(dx_ * 4 + s_) * 64];                                                                          \
        _Pragma("unroll") for (int n = 0; n < 8; ++n)                                                            \
            B_[n] = trow[(4 * s_) * PS + (n >> 1) * PITCH + (n & 1) * 16 + dx_];                                 \
    }
    __builtin_amdgcn_s_setprio(1);      // the matrix section outranks the co-resident workgroup's staging / epilogue VALU
#pragma unroll 1
    for (int dy = 0; dy < K; ++dy) {
        const float* wrow = wa + dy * K * 4 * 64;
        const float* trow = tb + dy * PITCH;
        float A0, A1, B0[8], B1[8];
        LLDWT_C16_FETCH(A0, B0, 0)
#pragma unroll
        for (int i = 0; i < NS; i += 2) {
            LLDWT_C16_FETCH(A1, B1, i + 1)
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int n = 0; n < 8; ++n) acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(B0[n], A0, acc[n], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (i + 2 < NS) LLDWT_C16_FETCH(A0, B0, i + 2)
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int n = 0; n < 8; ++n) acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(B1[n], A1, acc[n], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    __builtin_amdgcn_s_setprio(0);
#undef LLDWT_C16_FETCH
}

// conv1 (1 -> 16) as a GEMM with K = taps (padded to a multiple of 4): B gathered from the single-channel skip tile
template <int K, int SPITCH>
__device__ __forceinline__ void conv1_mfma(const float* __restrict__ sl, const float* __restrict__ w1l, int wave, int lane,
                                           int row_off, int col_off, floatx4 (&acc)[8]) {
    constexpr int KK = K * K, KS4 = (KK + 3) / 4;
    const int px = lane & 15, kk = lane >> 4;
#pragma unroll
    for (int s = 0; s < KS4; ++s) {
        const int tap = 4 * s + kk;
        const int dy = tap < KK ? tap / K : 0, dx = tap < KK ? tap % K : 0;
        const float A = w1l[s * 64 + lane];
        float B[8];
#pragma unroll
        for (int n = 0; n < 8; ++n)
            B[n] = sl[(row_off + wave * 4 + (n >> 1) + dy) * SPITCH + col_off + (n & 1) * 16 + px + dx];
#pragma unroll
        for (int n = 0; n < 8; ++n) acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(B[n], A, acc[n], 0, 0, 0);
    }
}

template <int K>
__device__ __forceinline__ void stage_w1(float* __restrict__ w1l, const float* __restrict__ W1p, int tid) {
    // W1p: packed [tap][oc] -> w1l[s*64 + kk*16 + oc] = W1[oc][tap = 4s+kk] (0 beyond the last tap)
    constexpr int KK = K * K, KS4 = (KK + 3) / 4;
    for (int i = tid; i < KS4 * 64; i += NT) {
        const int l = i & 63, s = i >> 6;
        const int tap = 4 * s + (l >> 4);
        w1l[i] = tap < KK ? W1p[tap * 16 + (l & 15)] : 0.f;
    }
}

// epilogue addressing of the layout above: tile n of wave `wave` is row wave*4 + n/2, columns (n&1)*16 + 4*kk .. +3
#define LLDWT_EPI_FOR(n_, gy_, gx_)                                                                              \
    _Pragma("unroll") for (int n_ = 0; n_ < 8; ++n_)                                                             \
        if (const int gy_ = y0 + wave * 4 + (n_ >> 1); gy_ < h)                                                  \
            if (const int gx_ = x0 + (n_ & 1) * 16 + 4 * kk; gx_ < w)
__device__ __forceinline__ floatx4 ld4_edge(const float* __restrict__ p, int gx, int w) {
    if (gx + 3 < w) {
        const f4u q = *reinterpret_cast<const f4u*>(p);
        return floatx4{q.x, q.y, q.z, q.w};
    }
    floatx4 v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int e = 0; e < 3; ++e)
        if (gx + e < w) v[e] = p[e];
    return v;
}
__device__ __forceinline__ void st4_edge(float* __restrict__ p, floatx4 v, int gx, int w) {
    if (gx + 3 < w) {
        *reinterpret_cast<f4u*>(p) = f4u{v[0], v[1], v[2], v[3]};
    } else {
#pragma unroll
        for (int e = 0; e < 3; ++e)
            if (gx + e < w) p[e] = v[e];
    }
}

template <int K>
__global__ __launch_bounds__(NT) void k_lift_a_mfma(CView src, float* __restrict__ skip_out, float* __restrict__ t2_out,
                                                    float* __restrict__ t1_out, float* __restrict__ src_out, int batch,
                                                    int h, int w, const float* __restrict__ taps,
                                                    const float* __restrict__ packed, int64_t packed_plane_stride,
                                                    int vertical, int linear) {
    constexpr int C = 16, R = K / 2, R2 = 2 * R, KK = K * K;
    constexpr int SH = TH + 2 * R2, SW = TW + 2 * R2;
    constexpr int T1H = TH + 2 * R, T1W = TW + 2 * R;      // 20 x 36 for K = 5: plane stride 720 == 16 (mod 32)
    constexpr int T1PS = ((T1H * T1W + 15) / 32) * 32 + 16;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* t1 = lds;                                       // [C] planes of stride T1PS, rows of T1W (16-byte aligned)
    float* wl = t1 + C * T1PS;                             // [KK][4][64]
    float* s_lds = wl + KK * 4 * 64;                       // [SH][SW+1]
    const PackOff o = pack_off(C, K);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t z = blockIdx.z;
    const int plane = (int)(z / batch);
    const int y0 = blockIdx.y * TH, x0 = blockIdx.x * TW;
    const float* pk = packed + plane * packed_plane_stride + (vertical ? 0 : o.orient);
    const float tp0 = taps[plane * 3 + 0], tp1 = taps[plane * 3 + 1], tp2 = taps[plane * 3 + 2];
    const int act = linear ? LLDWT_ACT_NONE : LLDWT_ACT_TANH;
    {
        // all global loads first (conv2 weights + the three skip-filter taps of every patch pixel), then the LDS stores
        floatx4 wv[Tile16<K>::NWV];
        constexpr int NS = (SH * SW + NT - 1) / NT;
        float va[NS], vb[NS], vc[NS];
        w16_load<K>(wv, pk + o.w2, tid);
        const int ddy = vertical ? 1 : 0, ddx = vertical ? 0 : 1;
        const float* sp = src.p + z * src.sz;
#pragma unroll
        for (int r = 0; r < NS; ++r) {
            const int i = tid + r * NT;
            const int ly = i / SW, lx = i - ly * SW;
            const int gy = y0 - R2 + ly, gx = x0 - R2 + lx;
            const bool in = i < SH * SW && gy >= 0 && gy < h && gx >= 0 && gx < w;
            const bool ina = in && gy - ddy >= 0 && gx - ddx >= 0, inc = in && gy + ddy < h && gx + ddx < w;
            const int64_t off = (int64_t)gy * src.sy + (int64_t)gx * src.sx;
            const int64_t step = (int64_t)ddy * src.sy + (int64_t)ddx * src.sx;
            const float qa = sp[ina ? off - step : 0], qb = sp[in ? off : 0], qc = sp[inc ? off + step : 0];
            va[r] = ina ? qa : 0.f; vb[r] = in ? qb : 0.f; vc[r] = inc ? qc : 0.f;
        }
        w16_store<K>(wl, wv, tid);
#pragma unroll
        for (int r = 0; r < NS; ++r) {
            const int i = tid + r * NT;
            if (i < SH * SW) {
                const int ly = i / SW, lx = i - ly * SW;
                const int gy = y0 - R2 + ly, gx = x0 - R2 + lx;
                const bool in = gy >= 0 && gy < h && gx >= 0 && gx < w;
                const float v = in ? tp0 * va[r] + tp1 * vb[r] + tp2 * vc[r] : 0.f;
                if (in && ly >= R2 && ly < R2 + TH && lx >= R2 && lx < R2 + TW) {
                    skip_out[(z * h + gy) * (int64_t)w + gx] = v;
                    if (src_out) src_out[(z * h + gy) * (int64_t)w + gx] = vb[r];
                }
                s_lds[ly * (SW + 1) + lx] = v;
            }
        }
    }
    __syncthreads();
    // conv1 (1 -> 16) + act on the (TH+2R)x(TW+2R) region (VALU: 6 % of the step's MACs); zero outside the image
    for (int i = tid; i < T1H * T1W; i += NT) {
        const int ly = i / T1W, lx = i - ly * T1W;
        const int gy = y0 - R + ly, gx = x0 - R + lx;
        float acc[C];
        const bool in = gy >= 0 && gy < h && gx >= 0 && gx < w;
#pragma unroll
        for (int oc = 0; oc < C; ++oc) acc[oc] = pk[o.b1 + oc];
        if (in) {
#pragma unroll
            for (int dy = 0; dy < K; ++dy)
#pragma unroll
                for (int dx = 0; dx < K; ++dx) {
                    const float v = s_lds[(ly + dy) * (SW + 1) + lx + dx];
                    const float* wt = pk + o.w1 + (dy * K + dx) * C;
#pragma unroll
                    for (int oc = 0; oc < C; ++oc) acc[oc] = fmaf(wt[oc], v, acc[oc]);
                }
        }
        const bool centre = t1_out && in && ly >= R && ly < R + TH && lx >= R && lx < R + TW;
#pragma unroll
        for (int oc = 0; oc < C; ++oc) {
            const float tv = in ? act_apply(acc[oc], act) : 0.f;
            t1[oc * T1PS + ly * T1W + lx] = tv;
            if (centre) t1_out[(z * C + oc) * ((int64_t)h * w) + (int64_t)gy * w + gx] = tv;
        }
    }
    __syncthreads();
    floatx4 acc[8];
#pragma unroll
    for (int n = 0; n < 8; ++n) acc[n] = floatx4{0.f, 0.f, 0.f, 0.f};
    conv16_mfma_ps<K, T1PS, T1W>(t1, wl, wave, lane, acc);
    // epilogue: lane holds 4 consecutive pixels of channel lane & 15 in each of its 8 pixel tiles
    const int oc = lane & 15, kk = lane >> 4;
    const int64_t cs = (int64_t)h * w;
    const float bv = pk[o.b2 + oc];
    float* op = t2_out + (z * C + oc) * cs;
    LLDWT_EPI_FOR(n, gy, gx) {
        floatx4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = act_apply(acc[n][e] + bv, act);
        st4_edge(op + (int64_t)gy * w + gx, v, gx, w);
    }
    (void)KK;
}

// Persistent over the tiles of one plane (grid = (workgroups, 1, planes)): the weights are staged once, and the loads of
// tile i+1 are in flight while the matrix cores work on tile i.  (One workgroup per tile left the memory phase exposed:
// the two workgroups of a CU start together and stay in lockstep -- measured 50 % MFMA-busy.)
template <int K>
__global__ __launch_bounds__(NT, 2) void k_lift_b_mfma(const float* __restrict__ skip, const float* __restrict__ t2,
                                                    float* __restrict__ t3_out, int batch, int h, int w,
                                                    const float* __restrict__ packed, int64_t packed_plane_stride,
                                                    int vertical) {
    constexpr int C = 16, R = K / 2, KK = K * K, KS4 = (KK + 3) / 4;
    constexpr int T1H = TH + 2 * R, T1W = TW + 2 * R;
    constexpr int T1PS = ((T1H * T1W + 15) / 32) * 32 + 16;
    constexpr int NSK = (T1H * T1W + NT - 1) / NT, NW1 = (KS4 * 64 + NT - 1) / NT;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* t = lds;                                        // [C] planes (16-byte aligned: vector stores)
    float* wl = t + C * T1PS;                              // conv3 weights [KK][4][64]
    float* w1l = wl + KK * 4 * 64;                         // conv1 weights [KS4][64]
    float* s_lds = w1l + KS4 * 64;                         // [T1H][T1W+1]
    const PackOff o = pack_off(C, K);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int plane = blockIdx.z;
    const float* pk = packed + plane * packed_plane_stride + (vertical ? 0 : o.orient);
    const int64_t cs = (int64_t)h * w;
    const int tiles_x = (w + TW - 1) / TW, tiles_y = (h + TH - 1) / TH;
    const int per_img = tiles_x * tiles_y, ntiles = batch * per_img;
    f4u raw[Tile16<K>::NVT];
    float sk[NSK];
    // tile-independent staging coordinates, computed once per workgroup: INTERIOR tiles (no border logic needed, the
    // common case) then cost one add + one load per vector at issue time and bare LDS stores afterwards -- vector ALU
    // instructions come straight out of the matrix pipe's time on this fp32 path
    constexpr bool FAST = (T1W % 4) == 0;                 // K = 5: the tile row is a whole number of vectors
    int relv[Tile16<K>::NVT], ldsv[Tile16<K>::NVT], relk[NSK], ldsk[NSK];
#pragma unroll
    for (int r = 0; r < Tile16<K>::NVT; ++r) {
        using G = Tile16<K>;
        const int i = tid + r * NT;
        const int c = i / (G::T1H * G::NVR), rem = i - c * (G::T1H * G::NVR);
        const int ly = rem / G::NVR, lxv = 4 * (rem % G::NVR);
        const bool live = i < G::NV;
        relv[r] = live ? c * (int)cs + (ly - R) * w + (lxv - R) : 0;
        ldsv[r] = live ? c * T1PS + ly * T1W + lxv : -1;
    }
#pragma unroll
    for (int r = 0; r < NSK; ++r) {
        const int i = tid + r * NT;
        const int ly = i / T1W, lx = i - ly * T1W;
        const bool live = i < T1H * T1W;
        relk[r] = live ? (ly - R) * w + (lx - R) : 0;
        ldsk[r] = live ? ly * (T1W + 1) + lx : -1;
    }
#define LLDWT_LB_INTERIOR(y0_, x0_) \
    (FAST && (y0_) >= R && (y0_) + TH + R <= h && (x0_) >= R && (x0_) - R + 4 * Tile16<K>::NVR <= w)

#define LLDWT_LB_COORDS(i_)                                                                                      \
    const int b_ = (i_) / per_img, r_ = (i_) - b_ * per_img;                                                     \
    const int64_t z = (int64_t)plane * batch + b_;                                                               \
    const int y0 = (r_ / tiles_x) * TH, x0 = (r_ % tiles_x) * TW;
#define LLDWT_LB_ISSUE(i_)                                                                                       \
    {                                                                                                            \
        LLDWT_LB_COORDS(i_)                                                                                      \
        if (LLDWT_LB_INTERIOR(y0, x0)) {                                                                         \
            const int toff = y0 * w + x0;                                                                        \
            const float* tb_ = t2 + (z * C) * cs;                                                                \
            const float* sb_ = skip + z * cs;                                                                    \
            _Pragma("unroll") for (int r = 0; r < Tile16<K>::NVT; ++r)                                           \
                raw[r] = *reinterpret_cast<const f4u*>(tb_ + (unsigned)(relv[r] + toff));                        \
            _Pragma("unroll") for (int r = 0; r < NSK; ++r) sk[r] = sb_[(unsigned)(relk[r] + toff)];             \
        } else {                                                                                                 \
            tile16_issue<K>(raw, t2 + (z * C) * cs, cs, y0, x0, h, w, tid);                                      \
            _Pragma("unroll") for (int r = 0; r < NSK; ++r) {                                                    \
                const int i = tid + r * NT;                                                                      \
                const int ly = i / T1W, lx = i - ly * T1W;                                                       \
                const int gy = y0 - R + ly, gx = x0 - R + lx;                                                    \
                const bool in = i < T1H * T1W && gy >= 0 && gy < h && gx >= 0 && gx < w;                         \
                sk[r] = (skip + z * cs)[in ? (unsigned)(gy * w + gx) : 0u];                                      \
            }                                                                                                    \
        }                                                                                                        \
    }

    int it = blockIdx.x;
    {
        // weights once per workgroup; the first tile's loads are issued with them (one round trip)
        floatx4 wv[Tile16<K>::NWV];
        float w1v[NW1];
        w16_load<K>(wv, pk + o.w3, tid);
#pragma unroll
        for (int r = 0; r < NW1; ++r) {
            const int i = tid + r * NT;
            const int tap = 4 * (i >> 6) + ((i & 63) >> 4);
            const bool ok = i < KS4 * 64 && tap < KK;
            const float tq = pk[o.w1 + (ok ? tap * 16 + (i & 15) : 0)];
            w1v[r] = ok ? tq : 0.f;
        }
        if (it < ntiles) LLDWT_LB_ISSUE(it)
        w16_store<K>(wl, wv, tid);
#pragma unroll
        for (int r = 0; r < NW1; ++r) {
            const int i = tid + r * NT;
            if (i < KS4 * 64) w1l[i] = w1v[r];
        }
    }
    const int oc = lane & 15, kk = lane >> 4;
    const float bv = pk[o.b3 + oc] + pk[o.b1 + oc];
    for (; it < ntiles; it += gridDim.x) {
        LLDWT_LB_COORDS(it)
        __syncthreads();                                   // the previous tile's LDS reads are done
        const bool interior = LLDWT_LB_INTERIOR(y0, x0);
        if (interior) {
#pragma unroll
            for (int r = 0; r < Tile16<K>::NVT; ++r)
                if (ldsv[r] >= 0) {
                    float2* d2 = reinterpret_cast<float2*>(t + ldsv[r]);
                    d2[0] = float2{raw[r].x, raw[r].y};
                    d2[1] = float2{raw[r].z, raw[r].w};
                }
#pragma unroll
            for (int r = 0; r < NSK; ++r)
                if (ldsk[r] >= 0) s_lds[ldsk[r]] = sk[r];
        } else {
            tile16_fix_store<K, T1PS, T1W>(t, raw, t2 + (z * C) * cs, cs, y0, x0, h, w, tid);
#pragma unroll
            for (int r = 0; r < NSK; ++r) {
                const int i = tid + r * NT;
                if (i < T1H * T1W) {
                    const int ly = i / T1W, lx = i - ly * T1W;
                    const int gy = y0 - R + ly, gx = x0 - R + lx;
                    s_lds[ly * (T1W + 1) + lx] = (gy >= 0 && gy < h && gx >= 0 && gx < w) ? sk[r] : 0.f;
                }
            }
        }
        __syncthreads();
        if (it + (int)gridDim.x < ntiles) LLDWT_LB_ISSUE(it + (int)gridDim.x)     // in flight during the MFMAs
        __builtin_amdgcn_sched_barrier(0);
        floatx4 acc[8];
#pragma unroll
        for (int n = 0; n < 8; ++n) acc[n] = floatx4{0.f, 0.f, 0.f, 0.f};
        conv1_mfma<K, T1W + 1>(s_lds, w1l, wave, lane, 0, 0, acc);          // residual r = conv1(skip), pre-activation
        conv16_mfma_ps<K, T1PS, T1W>(t, wl, wave, lane, acc);               // + conv3(t2)
        __builtin_amdgcn_sched_barrier(0);        // keep the epilogue's address arithmetic out of the MFMA section
        {
            float* op = t3_out + (z * C) * cs;
            if (interior) {
                // whole tile inside the image: one add per store, no bounds logic
                const unsigned e0 = (unsigned)(oc * (int)cs + (y0 + wave * 4) * w + x0 + 4 * kk);
#pragma unroll
                for (int n = 0; n < 8; ++n) {
                    const floatx4 v = acc[n] + bv;
                    *reinterpret_cast<f4u*>(op + ((n >> 1) * w + (n & 1) * 16) + e0) = f4u{v[0], v[1], v[2], v[3]};
                }
            } else {
                op += oc * cs;
                LLDWT_EPI_FOR(n, gy, gx) st4_edge(op + (unsigned)(gy * w + gx), acc[n] + bv, gx, w);
            }
        }
    }
#undef LLDWT_LB_COORDS
#undef LLDWT_LB_ISSUE
#undef LLDWT_LB_INTERIOR
}

// ---- kernel C: conv4(t3) ; dst_out = dst_in + sign*(skip + rw*net) -------------------------------------------
template <int C, int K>
__global__ __launch_bounds__(NT) void k_lift_c(const float* __restrict__ skip, const float* __restrict__ t3,
                                               CView dst_in, lldwt_view dst_out, int batch, int h, int w,
                                               const float* __restrict__ packed, int64_t packed_plane_stride,
                                               int vertical, float sign, float rw) {
    constexpr int R = K / 2;
    constexpr int T1H = TH + 2 * R, T1W = TW + 2 * R, T1P = T1W + 1;
    __shared__ float t[C][T1H][T1P];
    const PackOff o = pack_off(C, K);
    const int tid = threadIdx.x;
    const int64_t z = blockIdx.z;
    const int plane = (int)(z / batch);
    const int y0 = blockIdx.y * TH, x0 = blockIdx.x * TW;
    const float* pk = packed + plane * packed_plane_stride + (vertical ? 0 : o.orient);
    const int64_t cs = (int64_t)h * w;
    if constexpr (C == 16) {
        LLDWT_TILE16(K, tv, t3 + (z * C) * cs, cs, y0, x0, h, w, tid)
        LLDWT_TILE16_FIX(K, tv, t3 + (z * C) * cs, cs, y0, x0, h, w, tid)
        tile16_store<K, T1H * T1P, T1P>(&t[0][0][0], tv, tid);
    } else {
        for (int i = tid; i < T1H * T1W; i += NT) {
            const int ly = i / T1W, lx = i - ly * T1W;
            const int gy = y0 - R + ly, gx = x0 - R + lx;
            const bool in = gy >= 0 && gy < h && gx >= 0 && gx < w;
            const int64_t off = (int64_t)gy * w + gx;
#pragma unroll
            for (int c = 0; c < C; ++c) t[c][ly][lx] = in ? t3[(z * C + c) * cs + off] : 0.f;
        }
    }
    __syncthreads();
    // a thread owns two ADJACENT pixels of one row: the K + 1 values of a kernel row serve both (6 LDS reads per 10 FMAs
    // instead of 10; the kernel is bound by LDS reads, not by the FMAs), accumulated as one packed pair
    typedef float floatx2c __attribute__((ext_vector_type(2)));
    const int ly = tid / (TW / 2), lx = 2 * (tid % (TW / 2));
    floatx2c a01 = {pk[o.b4], pk[o.b4]};
    for (int ic = 0; ic < C; ++ic) {
        const float* wi = pk + o.w4 + ic * K * K;
#pragma unroll
        for (int dy = 0; dy < K; ++dy) {
            float v[K + 1];
#pragma unroll
            for (int j = 0; j < K + 1; ++j) v[j] = t[ic][ly + dy][lx + j];
#pragma unroll
            for (int dx = 0; dx < K; ++dx) {
                const float wv = wi[dy * K + dx];
                a01 = __builtin_elementwise_fma(floatx2c{wv, wv}, floatx2c{v[dx], v[dx + 1]}, a01);
            }
        }
    }
    const int gy = y0 + ly;
    if (gy < h) {
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int gx = x0 + lx + r;
            if (gx < w) {
                const float net = a01[r];
                const float sk = skip[z * cs + (int64_t)gy * w + gx];
                const float d = dst_in.p[z * dst_in.sz + (int64_t)gy * dst_in.sy + (int64_t)gx * dst_in.sx];
                dst_out.p[z * dst_out.sz + (int64_t)gy * dst_out.sy + (int64_t)gx * dst_out.sx] =
                    d + sign * (sk + rw * net);
            }
        }
    }
}

// out = in * s[plane]  (or / s[plane]) on views; config.scale == 1 only (wavelet_forward_v2.py:76-80)
// training: the input v goes to `save` (dense Z,h,w) -- the backward needs it for d(gain) and the op may be in place
__global__ void k_scale_view(CView in, lldwt_view out, int batch, int h, int w, const float* __restrict__ s,
                             int divide, float* __restrict__ save) {
    const int64_t z = blockIdx.z;
    const float f = s[z / batch];
    for (int y = blockIdx.y; y < h; y += gridDim.y)
        for (int x = blockIdx.x * blockDim.x + threadIdx.x; x < w; x += gridDim.x * blockDim.x) {
            const float v = in.p[z * in.sz + (int64_t)y * in.sy + (int64_t)x * in.sx];
            if (save) save[(z * h + y) * w + x] = v;
            out.p[z * out.sz + (int64_t)y * out.sy + (int64_t)x * out.sx] = divide ? v / f : v * f;
        }
}

static inline CView cv(lldwt_view v) { return CView{v.p, v.sz, v.sy, v.sx}; }

// ---- backward helpers of one lifting step -------------------------------------------------------------------
// g = G[dst_out] (view) copied to a dense (Z,h,w) tensor; G[dst_in] = g ("set": the step passes dst through unchanged)
__global__ void k_lift_bwd_pre(CView gout, lldwt_view gdin, float* __restrict__ g, int h, int w, float* __restrict__ zero_me,
                               int nzero) {
    // zero_me: the |max| slots of the fused backward launch that follows on the stream (128 floats per plane): zeroed here instead of
    // by a memset launch per step
    if (zero_me && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0)
        for (int i = threadIdx.x; i < nzero; i += blockDim.x) zero_me[i] = 0.f;
    const int64_t z = blockIdx.z;
    for (int y = blockIdx.y; y < h; y += gridDim.y)
        for (int x = blockIdx.x * blockDim.x + threadIdx.x; x < w; x += gridDim.x * blockDim.x) {
            const float v = gout.p[z * gout.sz + (int64_t)y * gout.sy + (int64_t)x * gout.sx];
            g[(z * h + y) * (int64_t)w + x] = v;
            gdin.p[z * gdin.sz + (int64_t)y * gdin.sy + (int64_t)x * gdin.sx] = v;
        }
}

// dskip = sign*(g + rw*dsk);  G[src] += taps^T (x) dskip  (transpose of the zero-padded 3-tap filter);
// dtaps[k] += sum dskip[p] * src[p + k - 1]
__global__ __launch_bounds__(256) void k_lift_bwd_fin(const float* __restrict__ g, const float* __restrict__ dsk,
                                                      const float* __restrict__ srcv, lldwt_view gsrc, int batch, int h,
                                                      int w, const float* __restrict__ taps, float* __restrict__ dtaps,
                                                      int vertical, float sign, float rw) {
    __shared__ float part[3][4];
    const int64_t z = blockIdx.z;
    const int plane = (int)(z / batch);
    const float t0 = taps[plane * 3 + 0], t1 = taps[plane * 3 + 1], t2 = taps[plane * 3 + 2];
    const int ddy = vertical ? 1 : 0, ddx = vertical ? 0 : 1;
    const int64_t base = z * (int64_t)h * w;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f;
    for (int y = blockIdx.y; y < h; y += gridDim.y)
        for (int x = blockIdx.x * blockDim.x + threadIdx.x; x < w; x += gridDim.x * blockDim.x) {
            auto dskip = [&](int yy, int xx) -> float {
                if (yy < 0 || yy >= h || xx < 0 || xx >= w) return 0.f;
                const int64_t i = base + (int64_t)yy * w + xx;
                return sign * (g[i] + rw * dsk[i]);
            };
            auto sv = [&](int yy, int xx) -> float {
                return (yy < 0 || yy >= h || xx < 0 || xx >= w) ? 0.f : srcv[base + (int64_t)yy * w + xx];
            };
            const float dm = dskip(y - ddy, x - ddx), d0 = dskip(y, x), dp = dskip(y + ddy, x + ddx);
            // skip[p] = t0*src[p-1] + t1*src[p] + t2*src[p+1]  =>  dsrc[p] = t0*dskip[p+1] + t1*dskip[p] + t2*dskip[p-1]
            gsrc.p[z * gsrc.sz + (int64_t)y * gsrc.sy + (int64_t)x * gsrc.sx] += t0 * dp + t1 * d0 + t2 * dm;
            a0 += d0 * sv(y - ddy, x - ddx);
            a1 += d0 * sv(y, x);
            a2 += d0 * sv(y + ddy, x + ddx);
        }
    float acc[3] = {a0, a1, a2};
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        float v = acc[k];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
        if ((threadIdx.x & 63) == 0) part[k][threadIdx.x >> 6] = v;
    }
    __syncthreads();
    if (threadIdx.x < 3)
        atomicAdd(dtaps + plane * 3 + threadIdx.x,
                  part[threadIdx.x][0] + part[threadIdx.x][1] + part[threadIdx.x][2] + part[threadIdx.x][3]);
}


// ---- backward-data of one P-block (C == 16) on the matrix cores ------------------------------------------------------
// The reference gets this from autograd through P_block_v2.forward (P_block_v2.py:40-55).  With g = dL/dnet:
//   dt3   = conv4^T(g)                         (1 -> 16, VALU)
//   dpre2 = tanh'(t2) * conv3^T(dt3)           (16 -> 16, MFMA)         kernel bwd_c (also copies g, sets G[dst_in])
//   dr    = tanh'(t1) * conv2^T(dpre2) + dt3   (16 -> 16, MFMA; "+ dt3" is the pre-activation residual)   kernel bwd_b
//   dsk   = conv1^T(dr)                        (16 -> 1, VALU)          kernel bwd_a
// conv^T = the same tile kernels with the weights read transposed and the taps mirrored from the FORWARD pack.
template <int K>
__device__ __forceinline__ void stage_w16T(float* __restrict__ wl, const float* __restrict__ Wp, int tid) {
    // Wp: forward pack [ic][tap][oc]; wl[(tap'*4 + s)*64 + kk*16 + oc'] = W[oc = 4s+kk][ic = oc'][KK-1-tap']
    constexpr int KK = K * K;
    for (int i = tid; i < KK * 4 * 64; i += NT) {
        const int l = i & 63, s = (i >> 6) & 3, tap = i >> 8;
        const int oc = l & 15, kk = l >> 4;
        wl[i] = Wp[(oc * KK + (KK - 1 - tap)) * 16 + 4 * s + kk];
    }
}

template <int K>
__global__ __launch_bounds__(NT) void k_lift_bwd_c_mfma(CView gout, lldwt_view gdin, float* __restrict__ g_out,
                                                        const float* __restrict__ t2, float* __restrict__ dt3_out,
                                                        float* __restrict__ dpre2_out, int batch, int h, int w,
                                                        const float* __restrict__ packed, int64_t packed_plane_stride,
                                                        int vertical, int linear) {
    constexpr int C = 16, R = K / 2, R2 = 2 * R, KK = K * K;
    constexpr int SH = TH + 2 * R2, SW = TW + 2 * R2;
    constexpr int T1H = TH + 2 * R, T1W = TW + 2 * R;
    constexpr int T1PS = ((T1H * T1W + 15) / 32) * 32 + 16;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* t1 = lds;                                       // [C] planes of dt3 (16-byte aligned)
    float* wl = t1 + C * T1PS;                             // conv3^T weights [KK][4][64]
    float* s_lds = wl + KK * 4 * 64;                       // [SH][SW+1] gradient patch
    const PackOff o = pack_off(C, K);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t z = blockIdx.z;
    const int plane = (int)(z / batch);
    const int y0 = blockIdx.y * TH, x0 = blockIdx.x * TW;
    const float* pk = packed + plane * packed_plane_stride + (vertical ? 0 : o.orient);
    const int64_t cs = (int64_t)h * w;
    {
        floatx4 wv[Tile16<K>::NWV];
        constexpr int NS = (SH * SW + NT - 1) / NT;
        float gv[NS];
        w16T_load<K>(wv, pk + o.w3, tid);
#pragma unroll
        for (int r = 0; r < NS; ++r) {
            const int i = tid + r * NT;
            const int ly = i / SW, lx = i - ly * SW;
            const int gy = y0 - R2 + ly, gx = x0 - R2 + lx;
            const bool in = i < SH * SW && gy >= 0 && gy < h && gx >= 0 && gx < w;
            const float tq = gout.p[in ? z * gout.sz + (int64_t)gy * gout.sy + (int64_t)gx * gout.sx : 0];
            gv[r] = in ? tq : 0.f;
        }
        w16T_store<K>(wl, wv, tid);
#pragma unroll
        for (int r = 0; r < NS; ++r) {
            const int i = tid + r * NT;
            if (i < SH * SW) {
                const int ly = i / SW, lx = i - ly * SW;
                const int gy = y0 - R2 + ly, gx = x0 - R2 + lx;
                const bool in = gy >= 0 && gy < h && gx >= 0 && gx < w;
                if (in && ly >= R2 && ly < R2 + TH && lx >= R2 && lx < R2 + TW) {
                    g_out[z * cs + (int64_t)gy * w + gx] = gv[r];
                    gdin.p[z * gdin.sz + (int64_t)gy * gdin.sy + (int64_t)gx * gdin.sx] = gv[r];   // the step passes dst through
                }
                s_lds[ly * (SW + 1) + lx] = gv[r];
            }
        }
    }
    __syncthreads();
    // dt3 = conv4^T(g) on the (TH+2R)x(TW+2R) region; zero outside the image
    for (int i = tid; i < T1H * T1W; i += NT) {
        const int ly = i / T1W, lx = i - ly * T1W;
        const int gy = y0 - R + ly, gx = x0 - R + lx;
        const bool in = gy >= 0 && gy < h && gx >= 0 && gx < w;
        float acc[C];
#pragma unroll
        for (int c = 0; c < C; ++c) acc[c] = 0.f;
        if (in) {
#pragma unroll
            for (int dy = 0; dy < K; ++dy)
#pragma unroll
                for (int dx = 0; dx < K; ++dx) {
                    const float v = s_lds[(ly + dy) * (SW + 1) + lx + dx];
                    const float* wt = pk + o.w4 + (KK - 1 - (dy * K + dx));     // W4[ic][mirrored tap], wave-uniform
#pragma unroll
                    for (int c = 0; c < C; ++c) acc[c] = fmaf(wt[c * KK], v, acc[c]);
                }
        }
        const bool centre = in && ly >= R && ly < R + TH && lx >= R && lx < R + TW;
#pragma unroll
        for (int c = 0; c < C; ++c) {
            t1[c * T1PS + ly * T1W + lx] = acc[c];
            if (centre) dt3_out[(z * C + c) * cs + (int64_t)gy * w + gx] = acc[c];
        }
    }
    __syncthreads();
    floatx4 acc[8];
#pragma unroll
    for (int n = 0; n < 8; ++n) acc[n] = floatx4{0.f, 0.f, 0.f, 0.f};
    conv16_mfma_ps<K, T1PS, T1W>(t1, wl, wave, lane, acc);
    const int oc = lane & 15, kk = lane >> 4;
    const int64_t cb = (z * C + oc) * cs;
    LLDWT_EPI_FOR(n, gy, gx) {
        const int64_t idx = cb + (int64_t)gy * w + gx;
        floatx4 v = acc[n];
        if (!linear) {
            const floatx4 tv = ld4_edge(t2 + idx, gx, w);
            v = v * (1.f - tv * tv);
        }
        st4_edge(dpre2_out + idx, v, gx, w);
    }
}

template <int K>
__global__ __launch_bounds__(NT) void k_lift_bwd_b_mfma(const float* __restrict__ dpre2, const float* __restrict__ t1v,
                                                        const float* __restrict__ dt3, float* __restrict__ dr_out,
                                                        int batch, int h, int w, const float* __restrict__ packed,
                                                        int64_t packed_plane_stride, int vertical, int linear) {
    constexpr int C = 16, R = K / 2, KK = K * K;
    constexpr int T1H = TH + 2 * R, T1W = TW + 2 * R;
    constexpr int T1PS = ((T1H * T1W + 15) / 32) * 32 + 16;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* t = lds;                                        // [C] planes of dpre2
    float* wl = t + C * T1PS;                              // conv2^T weights
    const PackOff o = pack_off(C, K);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t z = blockIdx.z;
    const int plane = (int)(z / batch);
    const int y0 = blockIdx.y * TH, x0 = blockIdx.x * TW;
    const float* pk = packed + plane * packed_plane_stride + (vertical ? 0 : o.orient);
    const int64_t cs = (int64_t)h * w;
    {
        floatx4 wv[Tile16<K>::NWV];
        w16T_load<K>(wv, pk + o.w2, tid);
        LLDWT_TILE16(K, tv, dpre2 + (z * C) * cs, cs, y0, x0, h, w, tid)
        w16T_store<K>(wl, wv, tid);
        LLDWT_TILE16_FIX(K, tv, dpre2 + (z * C) * cs, cs, y0, x0, h, w, tid)
        tile16_store<K, T1PS, T1W>(t, tv, tid);
    }
    __syncthreads();
    floatx4 acc[8];
#pragma unroll
    for (int n = 0; n < 8; ++n) acc[n] = floatx4{0.f, 0.f, 0.f, 0.f};
    conv16_mfma_ps<K, T1PS, T1W>(t, wl, wave, lane, acc);
    const int oc = lane & 15, kk = lane >> 4;
    const int64_t cb = (z * C + oc) * cs;
    LLDWT_EPI_FOR(n, gy, gx) {
        const int64_t idx = cb + (int64_t)gy * w + gx;
        floatx4 v = acc[n];
        if (!linear) {
            const floatx4 tv = ld4_edge(t1v + idx, gx, w);
            v = v * (1.f - tv * tv);
        }
        st4_edge(dr_out + idx, v + ld4_edge(dt3 + idx, gx, w), gx, w);
    }
    (void)KK;
}

// dsk = conv1^T(dr): 16 -> 1, VALU (mirror of kernel C)
template <int C, int K>
__global__ __launch_bounds__(NT) void k_lift_bwd_a(const float* __restrict__ dr, float* __restrict__ dsk, int batch, int h,
                                                   int w, const float* __restrict__ packed, int64_t packed_plane_stride,
                                                   int vertical) {
    constexpr int R = K / 2, KK = K * K;
    constexpr int T1H = TH + 2 * R, T1W = TW + 2 * R, T1P = T1W + 1;
    constexpr int NL = (T1H * T1W + NT - 1) / NT;
    __shared__ float t[C][T1H][T1P];
    const PackOff o = pack_off(C, K);
    const int tid = threadIdx.x;
    const int64_t z = blockIdx.z;
    const int plane = (int)(z / batch);
    const int y0 = blockIdx.y * TH, x0 = blockIdx.x * TW;
    const float* pk = packed + plane * packed_plane_stride + (vertical ? 0 : o.orient);
    const int64_t cs = (int64_t)h * w;
    if constexpr (C == 16) {
        LLDWT_TILE16(K, tv, dr + (z * C) * cs, cs, y0, x0, h, w, tid)
        LLDWT_TILE16_FIX(K, tv, dr + (z * C) * cs, cs, y0, x0, h, w, tid)
        tile16_store<K, T1H * T1P, T1P>(&t[0][0][0], tv, tid);
    } else {
#pragma unroll
        for (int r = 0; r < NL; ++r) {
            const int i = tid + r * NT;
            const int ly = i / T1W, lx = i - ly * T1W;
            const int gy = y0 - R + ly, gx = x0 - R + lx;
            const bool in = i < T1H * T1W && gy >= 0 && gy < h && gx >= 0 && gx < w;
            const int64_t off = in ? (z * C) * cs + (int64_t)gy * w + gx : 0;
            float v[C];
#pragma unroll
            for (int c = 0; c < C; ++c) v[c] = dr[off + c * cs];
            if (i < T1H * T1W) {
#pragma unroll
                for (int c = 0; c < C; ++c) t[c][ly][lx] = in ? v[c] : 0.f;
            }
        }
    }
    __syncthreads();
    // two adjacent pixels per thread, one kernel row of K + 1 LDS values serves both (see kernel C)
    typedef float floatx2c __attribute__((ext_vector_type(2)));
    const int ly = tid / (TW / 2), lx = 2 * (tid % (TW / 2));
    floatx2c a01 = {0.f, 0.f};
    for (int ic = 0; ic < C; ++ic) {
#pragma unroll
        for (int dy = 0; dy < K; ++dy) {
            float v[K + 1];
#pragma unroll
            for (int j = 0; j < K + 1; ++j) v[j] = t[ic][ly + dy][lx + j];
#pragma unroll
            for (int dx = 0; dx < K; ++dx) {
                const float wv = pk[o.w1 + (KK - 1 - (dy * K + dx)) * C + ic];     // W1[oc = ic][mirrored tap]
                a01 = __builtin_elementwise_fma(floatx2c{wv, wv}, floatx2c{v[dx], v[dx + 1]}, a01);
            }
        }
    }
    const int gy = y0 + ly;
    if (gy < h) {
        if (x0 + lx < w) dsk[z * cs + (int64_t)gy * w + x0 + lx] = a01[0];
        if (x0 + lx + 1 < w) dsk[z * cs + (int64_t)gy * w + x0 + lx + 1] = a01[1];
    }
}

// where one step keeps its intermediates (training: a per-step slice of the caller's `saved` buffer)
struct StepBufs {
    float* skip;   // (Z,h,w)
    float* t2;     // (Z,C,h,w)
    float* t3;     // (Z,C,h,w)
    float* t1;     // (Z,C,h,w) or null
    float* srcv;   // (Z,h,w)   or null
};

// LLDWT_TRAIN_LIFT=f32: the training forward of the lifting steps stays on the three fp32-MFMA launches (default: fused f16x3)
static const int g_train_lift_f16 = [] { const char* e = getenv("LLDWT_TRAIN_LIFT"); return (e && !strcmp(e, "f32")) ? 0 : 1; }();
extern "C" int lldwt_train_lift_f16(void) { return g_train_lift_f16 && g_lift_mode == 1; }

template <int C, int K>
static int launch_step(lldwt_view src, lldwt_view dst_in, lldwt_view dst_out, int64_t Z, int64_t batch, int64_t h,
                       int64_t w, const float* taps, const float* packed, int64_t pstride, int vertical, float sign,
                       float rw, int linear, const StepBufs& b, hipStream_t st) {
    if constexpr (C == LF_C && K == LF_K) {
        // eval path (no intermediates to save), tanh P-block: ONE fused launch on the fp16 matrix cores (lifting_f16.hip)
        if (g_lift_mode == 1 && !linear && b.t1 == nullptr) {
            LiftF16Views v{src.p, src.sz, src.sy, src.sx, dst_in.p, dst_in.sz, dst_in.sy, dst_in.sx,
                           dst_out.p, dst_out.sz, dst_out.sy, dst_out.sx};
            const PackOff o = pack_off(C, K);
            return lift_f16_step(v, Z, batch, h, w, taps, packed, pstride, o.orient, o.f16, vertical, sign, rw, st);
        }
        // training forward (intermediates saved), tanh P-block: the same fused kernel on its sequential path, which forms t3
        // explicitly, + stores of (src, skip, t1, t2, t3) -- the fp16 matrix cores instead of three fp32-MFMA launches.  The packed
        // buffer must carry the split-fp16 section (lldwt_pack_pblock, not _train); LLDWT_TRAIN_LIFT=f32 keeps the fp32 launches
        if (g_lift_mode == 1 && g_train_lift_f16 && !linear && b.t1 != nullptr && b.srcv != nullptr) {
            LiftF16Views v{src.p, src.sz, src.sy, src.sx, dst_in.p, dst_in.sz, dst_in.sy, dst_in.sx,
                           dst_out.p, dst_out.sz, dst_out.sy, dst_out.sx};
            const LiftF16Saved sv{b.srcv, b.skip, b.t1, b.t2, b.t3};
            const PackOff o = pack_off(C, K);
            return lift_f16_step_train(v, sv, Z, batch, h, w, taps, packed, pstride, o.orient, o.f16, vertical, sign, rw, st);
        }
    }
    dim3 grid((unsigned)cdiv(w, TW), (unsigned)cdiv(h, TH), (unsigned)Z), block(NT);
    if constexpr (C == 16) {
        constexpr int R = K / 2, R2 = 2 * R, KK = K * K;
        constexpr int T1H = TH + 2 * R, T1W = TW + 2 * R;
        constexpr int T1PS = ((T1H * T1W + 15) / 32) * 32 + 16;
        constexpr size_t sh_a = sizeof(float) * ((TH + 2 * R2) * (TW + 2 * R2 + 1) + 16 * T1PS + KK * 4 * 64);
        constexpr size_t sh_b = sizeof(float) * (T1H * (T1W + 1) + 16 * T1PS + KK * 4 * 64 + ((KK + 3) / 4) * 64);
        static bool attr_done = false;
        if (!attr_done) {
            hipFuncSetAttribute((const void*)k_lift_a_mfma<K>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh_a);
            hipFuncSetAttribute((const void*)k_lift_b_mfma<K>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh_b);
            attr_done = true;
        }
        hipLaunchKernelGGL((k_lift_a_mfma<K>), grid, block, sh_a, st, cv(src), b.skip, b.t2, b.t1, b.srcv, (int)batch,
                           (int)h, (int)w, taps, packed, pstride, vertical, linear);
        {
            const int64_t planes_ = Z / batch, per_plane = batch * cdiv(w, TW) * cdiv(h, TH);
            int64_t wg = (int64_t)lldwt_num_cus() * 2 / planes_;             // 2 workgroups per CU (LDS), one resident round
            if (wg > per_plane) wg = per_plane;
            if (wg < 1) wg = 1;
            hipLaunchKernelGGL((k_lift_b_mfma<K>), dim3((unsigned)wg, 1, (unsigned)planes_), block, sh_b, st, b.skip, b.t2,
                               b.t3, (int)batch, (int)h, (int)w, packed, pstride, vertical);
        }
    } else {
        hipLaunchKernelGGL((k_lift_a<C, K>), grid, block, 0, st, cv(src), b.skip, b.t2, b.t1, b.srcv, (int)batch, (int)h,
                           (int)w, taps, packed, pstride, vertical, linear);
        hipLaunchKernelGGL((k_lift_b<C, K>), grid, block, 0, st, b.skip, b.t2, b.t3, (int)batch, (int)h, (int)w, packed,
                           pstride, vertical);
    }
    hipLaunchKernelGGL((k_lift_c<C, K>), grid, block, 0, st, b.skip, b.t3, cv(dst_in), dst_out, (int)batch, (int)h,
                       (int)w, packed, pstride, vertical, sign, rw);
    return check_launch("lift_step");
}

static int dispatch_step(lldwt_view src, lldwt_view dst_in, lldwt_view dst_out, int64_t Z, int64_t batch, int64_t h,
                         int64_t w, const float* taps, const float* packed, int64_t pstride, int C, int K, int vertical,
                         float sign, float rw, int linear, const StepBufs& b, hipStream_t st) {
    // the tile kernels address one image's C channels with 32-bit element offsets
    LLDWT_REQUIRE((int64_t)C * h * w < (int64_t)1 << 31, "lift_step: C*h*w = %ld exceeds the 32-bit tile addressing",
                  (long)((int64_t)C * h * w));
#define LLDWT_CASE(CC, KK_)                                                                                         \
    if (C == CC && K == KK_)                                                                                        \
        return launch_step<CC, KK_>(src, dst_in, dst_out, Z, batch, h, w, taps, packed, pstride, vertical, sign, rw, \
                                    linear, b, st);
    LLDWT_CASE(16, 5)
    LLDWT_CASE(16, 3)
    LLDWT_CASE(8, 5)
    LLDWT_CASE(8, 3)
#undef LLDWT_CASE
    set_error("lift_step: unsupported (C=%d, K=%d); built: C in {8,16} x K in {3,5}", C, K);
    return LLDWT_EINVAL;
}

static inline StepBufs ws_bufs(float* ws, int64_t Z, int64_t h, int64_t w, int C) {
    StepBufs b;
    b.skip = ws;
    b.t2 = ws + Z * h * w;
    b.t3 = b.t2 + Z * C * h * w;
    b.t1 = nullptr;
    b.srcv = nullptr;
    return b;
}

// per-step slice of the training `saved` buffer: [srcv | skip | t1 | t2 | t3]
static inline int64_t saved_step_floats(int64_t Z, int64_t h, int64_t w, int C) { return Z * h * w * (2 + 3 * (int64_t)C); }
static inline StepBufs saved_bufs(float* base, int64_t Z, int64_t h, int64_t w, int C) {
    StepBufs b;
    const int64_t n = Z * h * w;
    b.srcv = base;
    b.skip = base + n;
    b.t1 = b.skip + n;
    b.t2 = b.t1 + n * C;
    b.t3 = b.t2 + n * C;
    return b;
}

// ---- the transform as a program of lifting steps over symbolic buffers ------------------------------------------------
enum { B_X = 0, B_LROW, B_HROW, B_TMPL, B_TMPH, B_LL0, B_LL1, B_LL, B_YH0 };

struct SymView {
    int buf;
    int64_t off, sz, sy, sx;
};
static inline SymView sv(int buf, int64_t off, int64_t sz, int64_t sy, int64_t sx) { return SymView{buf, off, sz, sy, sx}; }

static void push_op(lldwt_lift_op* ops, int& n, int max_ops, int kind, SymView src, SymView din, SymView dout, int64_t h,
                    int64_t w, int vertical, int tap, int block, int is_u, float sign, int64_t& saved_off, int64_t Z,
                    int C) {
    if (n < max_ops) {
        lldwt_lift_op& o = ops[n];
        o.kind = kind;
        o.buf_src = src.buf; o.off_src = src.off; o.sz_src = src.sz; o.sy_src = src.sy; o.sx_src = src.sx;
        o.buf_din = din.buf; o.off_din = din.off; o.sz_din = din.sz; o.sy_din = din.sy; o.sx_din = din.sx;
        o.buf_dout = dout.buf; o.off_dout = dout.off; o.sz_dout = dout.sz; o.sy_dout = dout.sy; o.sx_dout = dout.sx;
        o.h = (int32_t)h; o.w = (int32_t)w; o.vertical = vertical; o.tap = tap; o.block = block; o.is_u = is_u;
        o.sign = sign; o.saved_off = saved_off;
    }
    saved_off += kind == 0 ? saved_step_floats(Z, h, w, C) : Z * h * w;      // a scale op keeps its input
    ++n;
}

// 2-stage lifting on symbolic (L,H) -> (Lout,Hout) (wavelet_forward_v2.py:58-74 / wavelet_inverse_v2.py:76-90)
static void two_stage(lldwt_lift_op* ops, int& n, int max_ops, bool inverse, SymView L, SymView H, SymView Lout,
                      SymView Hout, SymView tL, SymView tH, int64_t hh, int64_t ww, int vertical, int blk,
                      int64_t& so, int64_t Z, int C) {
    if (!inverse) {
        push_op(ops, n, max_ops, 0, L, H, tH, hh, ww, vertical, 0, blk + 0, 0, 1.f, so, Z, C);        // :60-62
        push_op(ops, n, max_ops, 0, tH, L, tL, hh, ww, vertical, 1, blk + 0, 1, 1.f, so, Z, C);       // :64-66
        push_op(ops, n, max_ops, 0, tL, tH, Hout, hh, ww, vertical, 2, blk + 1, 0, 1.f, so, Z, C);    // :68-70
        push_op(ops, n, max_ops, 0, Hout, tL, Lout, hh, ww, vertical, 3, blk + 1, 1, 1.f, so, Z, C);  // :72-74
    } else {
        push_op(ops, n, max_ops, 0, H, L, tL, hh, ww, vertical, 3, blk + 1, 1, -1.f, so, Z, C);       // :76-78
        push_op(ops, n, max_ops, 0, tL, H, tH, hh, ww, vertical, 2, blk + 1, 0, -1.f, so, Z, C);      // :80-82
        push_op(ops, n, max_ops, 0, tH, tL, Lout, hh, ww, vertical, 1, blk + 0, 1, -1.f, so, Z, C);   // :84-86
        push_op(ops, n, max_ops, 0, Lout, tH, Hout, hh, ww, vertical, 0, blk + 0, 0, -1.f, so, Z, C); // :88-90
    }
}

// scale ops (config.scale == 1): kind 1 = dout = src * nh, 2 = * nl, 3 = / nh, 4 = / nl (per plane)
static void push_scale(lldwt_lift_op* ops, int& n, int max_ops, int kind, SymView src, SymView dout, int64_t h, int64_t w,
                       int64_t& so, int64_t Z) {
    push_op(ops, n, max_ops, kind, src, src, dout, h, w, 0, 0, 0, 0, 1.f, so, Z, 0);
}

static int build_program(lldwt_lift_op* ops, int max_ops, int64_t Z, int64_t H, int64_t W, int levels, int different,
                         int block_offset, int inverse, int scale, int C, int64_t* saved_total) {
    int n = 0;
    int64_t so = 0;
    if (!inverse) {
        for (int lev = 0; lev < levels; ++lev) {
            const int64_t h = H >> lev, w = W >> lev, hh = h / 2, wh = w / 2, sub = hh * wh;
            const int blk = block_offset + (different ? lev * 2 : 0);
            const int bin = lev == 0 ? B_X : (((lev - 1) & 1) ? B_LL1 : B_LL0);
            const int bll = lev == levels - 1 ? B_LL : ((lev & 1) ? B_LL1 : B_LL0);
            const int byh = B_YH0 + lev;
            SymView A = sv(bin, 0, h * w, 2 * w, 1), Bv = sv(bin, w, h * w, 2 * w, 1);         // rows (wavelet_forward_v2.py:27-29)
            SymView vL = sv(B_LROW, 0, hh * w, w, 1), vH = sv(B_HROW, 0, hh * w, w, 1);
            SymView tL = sv(B_TMPL, 0, hh * w, w, 1), tH = sv(B_TMPH, 0, hh * w, w, 1);
            two_stage(ops, n, max_ops, false, A, Bv, vL, vH, tL, tH, hh, w, 1, blk, so, Z, C);
            if (scale) { push_scale(ops, n, max_ops, 1, vH, vH, hh, w, so, Z); push_scale(ops, n, max_ops, 2, vL, vL, hh, w, so, Z); }
            SymView vLL = sv(bll, 0, sub, wh, 1);
            SymView vLH = sv(byh, 0, 3 * sub, wh, 1), vHL = sv(byh, sub, 3 * sub, wh, 1), vHH = sv(byh, 2 * sub, 3 * sub, wh, 1);
            SymView t2L = sv(B_TMPL, 0, sub, wh, 1), t2H = sv(B_TMPH, 0, sub, wh, 1);
            SymView Le = sv(B_LROW, 0, hh * w, w, 2), Lo = sv(B_LROW, 1, hh * w, w, 2);          // columns of L (:32-39)
            two_stage(ops, n, max_ops, false, Le, Lo, vLL, vHL, t2L, t2H, hh, wh, 0, blk, so, Z, C);
            if (scale) { push_scale(ops, n, max_ops, 1, vHL, vHL, hh, wh, so, Z); push_scale(ops, n, max_ops, 2, vLL, vLL, hh, wh, so, Z); }
            SymView He = sv(B_HROW, 0, hh * w, w, 2), Ho = sv(B_HROW, 1, hh * w, w, 2);          // columns of H (:43-51)
            // its own temporaries (the second halves of tmpL / tmpH, free during the column passes): the L and H column
            // passes are independent, and the eval executor runs step k of both in ONE launch (run_program)
            const int64_t toff = scale ? 0 : Z * sub;
            SymView u2L = sv(B_TMPL, toff, sub, wh, 1), u2H = sv(B_TMPH, toff, sub, wh, 1);
            two_stage(ops, n, max_ops, false, He, Ho, vLH, vHH, u2L, u2H, hh, wh, 0, blk, so, Z, C);
            if (scale) { push_scale(ops, n, max_ops, 1, vHH, vHH, hh, wh, so, Z); push_scale(ops, n, max_ops, 2, vLH, vLH, hh, wh, so, Z); }
        }
    } else {
        const int blk = block_offset;       // lifting_dwt_nets.py:718-722: every inverse level uses the same pair
        for (int lev = levels - 1; lev >= 0; --lev) {
            const int64_t h = H >> lev, w = W >> lev, hh = h / 2, wh = w / 2, sub = hh * wh;
            const int bin = lev == levels - 1 ? B_LL : ((lev & 1) ? B_LL0 : B_LL1);   // written by level lev+1 into llbuf[(lev+1)&1]
            const int bout = lev == 0 ? B_X : ((lev & 1) ? B_LL1 : B_LL0);
            const int byh = B_YH0 + lev;
            SymView vLL = sv(bin, 0, sub, wh, 1);
            SymView vLH = sv(byh, 0, 3 * sub, wh, 1), vHL = sv(byh, sub, 3 * sub, wh, 1), vHH = sv(byh, 2 * sub, 3 * sub, wh, 1);
            SymView t2L = sv(B_TMPL, 0, sub, wh, 1), t2H = sv(B_TMPH, 0, sub, wh, 1);
            SymView inL = vLL, inH = vHL;
            // scratch for the scaled copies (config.scale == 1, wavelet_inverse_v2.py:70-74): second halves of tmpL/tmpH
            SymView sL = sv(B_TMPL, Z * (H / 2) * W / 2, sub, wh, 1), sH = sv(B_TMPH, Z * (H / 2) * W / 2, sub, wh, 1);
            if (scale) { push_scale(ops, n, max_ops, 4, vLL, sL, hh, wh, so, Z); push_scale(ops, n, max_ops, 3, vHL, sH, hh, wh, so, Z); inL = sL; inH = sH; }
            SymView Le = sv(B_LROW, 0, hh * w, w, 2), Lo = sv(B_LROW, 1, hh * w, w, 2);          // (LL,HL) -> L (:21-26)
            two_stage(ops, n, max_ops, true, inL, inH, Le, Lo, t2L, t2H, hh, wh, 0, blk, so, Z, C);
            inL = vLH; inH = vHH;
            if (scale) { push_scale(ops, n, max_ops, 4, vLH, sL, hh, wh, so, Z); push_scale(ops, n, max_ops, 3, vHH, sH, hh, wh, so, Z); inL = sL; inH = sH; }
            SymView He = sv(B_HROW, 0, hh * w, w, 2), Ho = sv(B_HROW, 1, hh * w, w, 2);          // (LH,HH) -> H (:28-33)
            const int64_t toff = scale ? 0 : Z * sub;                                              // see the forward program
            SymView u2L = sv(B_TMPL, toff, sub, wh, 1), u2H = sv(B_TMPH, toff, sub, wh, 1);
            two_stage(ops, n, max_ops, true, inL, inH, He, Ho, u2L, u2H, hh, wh, 0, blk, so, Z, C);
            SymView vL = sv(B_LROW, 0, hh * w, w, 1), vH = sv(B_HROW, 0, hh * w, w, 1);
            SymView tL = sv(B_TMPL, 0, hh * w, w, 1), tH = sv(B_TMPH, 0, hh * w, w, 1);
            SymView A = sv(bout, 0, h * w, 2 * w, 1), Bv = sv(bout, w, h * w, 2 * w, 1);        // (L,H) -> rows (:35-37)
            if (scale) { push_scale(ops, n, max_ops, 4, vL, vL, hh, w, so, Z); push_scale(ops, n, max_ops, 3, vH, vH, hh, w, so, Z); }
            two_stage(ops, n, max_ops, true, vL, vH, A, Bv, tL, tH, hh, w, 1, blk, so, Z, C);
        }
    }
    if (saved_total) *saved_total = so;
    return n;
}

static inline lldwt_view resolve(float* const* bases, int buf, int64_t off, int64_t sz, int64_t sy, int64_t sx) {
    return lldwt_view{bases[buf] + off, sz, sy, sx};
}

struct RunCtx {
    int64_t Z, batch, planes;
    const float* taps;     // (4,planes,3)
    const float* packed;   // (planes,nblocks,2,total)
    int64_t pstride, total;
    int C, K, linear;
    float rw;
    const float* nh;
    const float* nl;
    float* step_ws;
    float* saved;          // training: per-step intermediates, or null
    hipStream_t st;
};

// ops[i .. i+3] and ops[i+4 .. i+7] are the two independent column passes (L and H) of one level: same geometry, same
// blocks, different buffers (build_program gives the second pass its own temporaries)
static bool column_pass_pair(const lldwt_lift_op* ops, int n, int i) {
    if (i + 8 > n) return false;
    for (int k = 0; k < 4; ++k) {
        const lldwt_lift_op &a = ops[i + k], &b = ops[i + 4 + k];
        if (a.kind != 0 || b.kind != 0 || a.vertical != 0 || b.vertical != 0 || a.h != b.h || a.w != b.w || a.tap != b.tap ||
            a.block != b.block || a.is_u != b.is_u || a.sign != b.sign)
            return false;
        if (a.h != ops[i].h || a.w != ops[i].w) return false;
        if (a.buf_dout == b.buf_dout && a.off_dout == b.off_dout) return false;
    }
    // the second pass must not touch what the first one reads or writes (and vice versa): distinct (buffer, offset) pairs
    for (int k = 0; k < 4; ++k)
        for (int m = 0; m < 4; ++m) {
            const lldwt_lift_op &a = ops[i + k], &b = ops[i + 4 + m];
            auto same = [](int ba, int64_t oa, int bb, int64_t ob) { return ba == bb && oa == ob; };
            if (same(a.buf_dout, a.off_dout, b.buf_src, b.off_src) || same(a.buf_dout, a.off_dout, b.buf_din, b.off_din) ||
                same(a.buf_dout, a.off_dout, b.buf_dout, b.off_dout) || same(b.buf_dout, b.off_dout, a.buf_src, a.off_src) ||
                same(b.buf_dout, b.off_dout, a.buf_din, a.off_din))
                return false;
        }
    return true;
}

static int run_program(const lldwt_lift_op* ops, int n, float* const* bases, const RunCtx& c) {
    const bool fused_eval = g_lift_mode == 1 && !c.linear && c.saved == nullptr && c.C == LF_C && c.K == LF_K;
    for (int i = 0; i < n; ++i) {
        const lldwt_lift_op& o = ops[i];
        if (fused_eval && o.kind == 0 && column_pass_pair(ops, n, i)) {
            // step k of the L pass and step k of the H pass in one launch (twice the tiles: the deep levels' launches are
            // too small to fill the chip on their own)
            const PackOff po = pack_off(c.C, c.K);
            for (int k = 0; k < 4; ++k) {
                const lldwt_lift_op &a = ops[i + k], &b = ops[i + 4 + k];
                auto views = [&](const lldwt_lift_op& q) {
                    const lldwt_view s_ = resolve(bases, q.buf_src, q.off_src, q.sz_src, q.sy_src, q.sx_src);
                    const lldwt_view d_ = resolve(bases, q.buf_din, q.off_din, q.sz_din, q.sy_din, q.sx_din);
                    const lldwt_view o_ = resolve(bases, q.buf_dout, q.off_dout, q.sz_dout, q.sy_dout, q.sx_dout);
                    return LiftF16Views{s_.p, s_.sz, s_.sy, s_.sx, d_.p, d_.sz, d_.sy, d_.sx, o_.p, o_.sz, o_.sy, o_.sx};
                };
                const float* pk = c.packed + ((int64_t)a.block * 2 + a.is_u) * c.total;
                const LiftF16Views va = views(a), vb = views(b);
                int r = lift_f16_step2(va, &vb, c.Z, c.batch, a.h, a.w, c.taps + (int64_t)a.tap * c.planes * 3, pk, c.pstride,
                                       po.orient, po.f16, a.vertical, a.sign, c.rw, c.st);
                if (r) return r;
            }
            i += 7;
            continue;
        }
        lldwt_view src = resolve(bases, o.buf_src, o.off_src, o.sz_src, o.sy_src, o.sx_src);
        lldwt_view dout = resolve(bases, o.buf_dout, o.off_dout, o.sz_dout, o.sy_dout, o.sx_dout);
        if (o.kind != 0) {
            const float* f = (o.kind == 1 || o.kind == 3) ? c.nh : c.nl;
            dim3 grid((unsigned)cdiv(o.w, 256), (unsigned)(o.h < 1024 ? o.h : 1024), (unsigned)c.Z);
            hipLaunchKernelGGL(k_scale_view, grid, dim3(256), 0, c.st, cv(src), dout, (int)c.batch, o.h, o.w, f, o.kind >= 3,
                               c.saved ? c.saved + o.saved_off : nullptr);
            continue;
        }
        lldwt_view din = resolve(bases, o.buf_din, o.off_din, o.sz_din, o.sy_din, o.sx_din);
        const float* pk = c.packed + ((int64_t)o.block * 2 + o.is_u) * c.total;
        const StepBufs b = c.saved ? saved_bufs(c.saved + o.saved_off, c.Z, o.h, o.w, c.C) : ws_bufs(c.step_ws, c.Z, o.h, o.w, c.C);
        int r = dispatch_step(src, din, dout, c.Z, c.batch, o.h, o.w, c.taps + (int64_t)o.tap * c.planes * 3, pk, c.pstride,
                              c.C, c.K, o.vertical, o.sign, c.rw, c.linear, b, c.st);
        if (r) return r;
    }
    return check_launch("lifting program");
}

}  // namespace lldwt

using namespace lldwt;

extern "C" int64_t lldwt_pblock_packed_floats(int C, int K) { return pack_off(C, K).total; }

extern "C" int lldwt_set_precision(int prec) {
    LLDWT_REQUIRE(prec >= 0 && prec <= 2, "set_precision: %d (0 = f16x3, 1 = fp16, 2 = bf16)", prec);
    split_set_precision(prec);
    return LLDWT_OK;
}
extern "C" int lldwt_get_precision(void) { return split_precision(); }

extern "C" int lldwt_set_diagnostics(int kind, void* stamps, int64_t nbytes, int flags) {
    LLDWT_REQUIRE(kind >= 0 && kind <= 2, "set_diagnostics: kind %d (0 = fused lifting step, 1 = tree-pair conv, 2 = cgp chain)", kind);
    LLDWT_REQUIRE(nbytes >= 0 && (stamps != nullptr || nbytes == 0), "set_diagnostics: bad buffer");
    if (kind == 0) {
        lift_f16_set_stamps(stamps, nbytes);
        lift_f16_set_debug(flags);
    } else if (kind == 1) {
        f3_set_stamps(stamps, nbytes);
    } else {
        cgp16_set_stamps(stamps, nbytes);
    }
    return LLDWT_OK;
}

extern "C" int lldwt_set_lift_mode(int mode) {
    LLDWT_REQUIRE(mode == 0 || mode == 1, "set_lift_mode: 0 (fp32 MFMA kernels) or 1 (fused split-fp16 kernel)");
    g_lift_mode = mode;
    return LLDWT_OK;
}
extern "C" int lldwt_get_lift_mode(void) { return g_lift_mode; }

static int pack_pblock_impl(const float* w1, const float* b1, const float* w2, const float* b2, const float* w3,
                            const float* b3, const float* w4, const float* b4, float* packed, int planes, int C, int K,
                            bool with_f16, void* stream, bool compose = true);

extern "C" int lldwt_pack_pblock(const float* w1, const float* b1, const float* w2, const float* b2, const float* w3,
                                 const float* b3, const float* w4, const float* b4, float* packed, int planes, int C,
                                 int K, void* stream) {
    return pack_pblock_impl(w1, b1, w2, b2, w3, b3, w4, b4, packed, planes, C, K, true, stream);
}

extern "C" int lldwt_pack_pblock_train(const float* w1, const float* b1, const float* w2, const float* b2, const float* w3,
                                       const float* b3, const float* w4, const float* b4, float* packed, int planes, int C,
                                       int K, void* stream) {
    return pack_pblock_impl(w1, b1, w2, b2, w3, b3, w4, b4, packed, planes, C, K, false, stream);
}

// the whole pack for the fused kernel's SEQUENTIAL path (training forward): everything lldwt_pack_pblock writes except the composed
// 9x9 kernels of the eval path, which are nine tenths of the pack's time and which no training kernel reads
extern "C" int lldwt_pack_pblock_seq(const float* w1, const float* b1, const float* w2, const float* b2, const float* w3,
                                     const float* b3, const float* w4, const float* b4, float* packed, int planes, int C,
                                     int K, void* stream) {
    return pack_pblock_impl(w1, b1, w2, b2, w3, b3, w4, b4, packed, planes, C, K, true, stream, false);
}

static int pack_pblock_impl(const float* w1, const float* b1, const float* w2, const float* b2, const float* w3,
                            const float* b3, const float* w4, const float* b4, float* packed, int planes, int C, int K,
                            bool with_f16, void* stream, bool compose) {
    LLDWT_REQUIRE(planes > 0 && C > 0 && (K == 3 || K == 5), "pack_pblock: bad planes/C/K (%d,%d,%d)", planes, C, K);
    LLDWT_REQUIRE(w1 && b1 && w2 && b2 && w3 && b3 && w4 && b4 && packed, "pack_pblock: null pointer");
    const PackOff o = pack_off(C, K);
    dim3 grid((unsigned)cdiv(o.total, 256), (unsigned)planes);
    hipLaunchKernelGGL(k_pack_pblock, grid, dim3(256), 0, (hipStream_t)stream, w1, b1, w2, b2, w3, b3, w4, b4, packed, C,
                       K);
    if (with_f16 && lift_f16_floats(C, K) > 0) {
        int r = lift_f16_pack(w1, w2, w3, w4, b1, b3, b4, packed, o.total, o.f16, planes, compose ? 1 : 0, (hipStream_t)stream);
        if (r) return r;
    }
    return check_launch("pack_pblock");
}

extern "C" int64_t lldwt_lift_step_ws_bytes(int64_t Z, int64_t h, int64_t w, int C) {
    return (int64_t)sizeof(float) * Z * h * w * (1 + 2 * (int64_t)C);
}

extern "C" int lldwt_lift_step(lldwt_view src, lldwt_view dst_in, lldwt_view dst_out, int64_t Z, int64_t batch,
                               int64_t h, int64_t w, const float* taps, const float* packed, int C, int K, int vertical,
                               float sign, float res_weight, int linear, void* ws, int64_t ws_bytes, void* stream) {
    LLDWT_REQUIRE(Z > 0 && batch > 0 && Z % batch == 0 && h > 0 && w > 0, "lift_step: bad dims Z=%ld batch=%ld h=%ld w=%ld",
                  (long)Z, (long)batch, (long)h, (long)w);
    LLDWT_REQUIRE(src.p && dst_in.p && dst_out.p && taps && packed && ws, "lift_step: null pointer");
    LLDWT_REQUIRE(Z <= 65535, "lift_step: Z=%ld exceeds grid.z", (long)Z);
    if (ws_bytes < lldwt_lift_step_ws_bytes(Z, h, w, C)) {
        set_error("lift_step: workspace %ld < %ld bytes", (long)ws_bytes, (long)lldwt_lift_step_ws_bytes(Z, h, w, C));
        return LLDWT_EWS;
    }
    return dispatch_step(src, dst_in, dst_out, Z, batch, h, w, taps, packed, pack_off(C, K).total, C, K, vertical, sign,
                         res_weight, linear, ws_bufs((float*)ws, Z, h, w, C), (hipStream_t)stream);
}

// backward pieces of one step, exposed so that the host can chain them with the conv engine (see autograd.py)
extern "C" int lldwt_lift_bwd_pre(lldwt_view g_dst_out, lldwt_view g_dst_in, float* g, int64_t Z, int64_t h, int64_t w,
                                  void* stream) {
    LLDWT_REQUIRE(g_dst_out.p && g_dst_in.p && g && Z > 0 && Z <= 65535 && h > 0 && w > 0, "lift_bwd_pre: bad arguments");
    dim3 grid((unsigned)cdiv(w, 256), (unsigned)(h < 1024 ? h : 1024), (unsigned)Z);
    hipLaunchKernelGGL(k_lift_bwd_pre, grid, dim3(256), 0, (hipStream_t)stream, cv(g_dst_out), g_dst_in, g, (int)h, (int)w,
                       (float*)nullptr, 0);
    return check_launch("lift_bwd_pre");
}

extern "C" int lldwt_lift_bwd_fin(const float* g, const float* dsk, const float* srcv, lldwt_view g_src, int64_t Z,
                                  int64_t batch, int64_t h, int64_t w, const float* taps, float* dtaps, int vertical,
                                  float sign, float res_weight, void* stream) {
    LLDWT_REQUIRE(g && dsk && srcv && g_src.p && taps && dtaps && Z > 0 && Z <= 65535 && batch > 0 && h > 0 && w > 0,
                  "lift_bwd_fin: bad arguments");
    // every workgroup ends with three float atomics onto its plane's tap gradients: with 64 row slices a level-0 launch sent 1 024
    // atomics to each of nine addresses (they serialise in L2 and took longer than the 63 MB the kernel moves); 16 slices
    static const int fin_rows = [] { const char* e = getenv("LLDWT_FIN_ROWS"); const int v = e ? atoi(e) : 16; return v > 0 ? v : 16; }();
    dim3 grid((unsigned)cdiv(w, 256), (unsigned)(h < fin_rows ? h : fin_rows), (unsigned)Z);
    hipLaunchKernelGGL(k_lift_bwd_fin, grid, dim3(256), 0, (hipStream_t)stream, g, dsk, srcv, g_src, (int)batch, (int)h,
                       (int)w, taps, dtaps, vertical, sign, res_weight);
    return check_launch("lift_bwd_fin");
}

extern "C" int64_t lldwt_lift_step_bwd_ws_bytes(int64_t Z, int64_t h, int64_t w, int C) {
    // g | dsk | dt3 | dpre2 | dr | 2 x 64 |dY|-max slots per plane (split-fp16 weight gradient of the 16 -> 16 convs)
    return (int64_t)sizeof(float) * (Z * h * w * (2 + 3 * (int64_t)C) + 128 * Z);
}

template <int K>
static int launch_step_bwd(lldwt_view g_dst_out, lldwt_view g_dst_in, float* g, float* dsk, float* dt3, float* dpre2,
                           float* dr, const float* t1, const float* t2, int64_t Z, int64_t batch, int64_t h, int64_t w,
                           const float* packed, int64_t pstride, int vertical, int linear, hipStream_t st) {
    constexpr int R = K / 2, R2 = 2 * R, KK = K * K;
    constexpr int T1H = TH + 2 * R, T1W = TW + 2 * R;
    constexpr int T1PS = ((T1H * T1W + 15) / 32) * 32 + 16;
    constexpr size_t sh_c = sizeof(float) * ((TH + 2 * R2) * (TW + 2 * R2 + 1) + 16 * T1PS + KK * 4 * 64);
    constexpr size_t sh_b = sizeof(float) * (16 * T1PS + KK * 4 * 64);
    static bool attr_done = false;
    if (!attr_done) {
        hipFuncSetAttribute((const void*)k_lift_bwd_c_mfma<K>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh_c);
        hipFuncSetAttribute((const void*)k_lift_bwd_b_mfma<K>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh_b);
        attr_done = true;
    }
    dim3 grid((unsigned)cdiv(w, TW), (unsigned)cdiv(h, TH), (unsigned)Z), block(NT);
    hipLaunchKernelGGL((k_lift_bwd_c_mfma<K>), grid, block, sh_c, st, cv(g_dst_out), g_dst_in, g, t2, dt3, dpre2, (int)batch,
                       (int)h, (int)w, packed, pstride, vertical, linear);
    hipLaunchKernelGGL((k_lift_bwd_b_mfma<K>), grid, block, sh_b, st, dpre2, t1, dt3, dr, (int)batch, (int)h, (int)w, packed,
                       pstride, vertical, linear);
    hipLaunchKernelGGL((k_lift_bwd_a<16, K>), grid, block, 0, st, dr, dsk, (int)batch, (int)h, (int)w, packed, pstride,
                       vertical);
    return check_launch("lift_step_bwd");
}

namespace lldwt {
int wgrad16_f16x3(const float* x, const float* dy, float* dw, float* dbias, float* slots_ws, int64_t slots_stride, bool slots_ready,
                  int64_t planes, int64_t batch, int64_t h, int64_t w_, float alpha, const int8_t* tap_of, hipStream_t st);
int wgrad_thin_pair(const float* t3, const float* g, float* dw4, float* db4, const float* skip, const float* dr, float* dw1,
                    float* db1, int64_t planes, int64_t batch, int64_t h, int64_t w_, float alpha, int swap_hw, int K, hipStream_t st);
int wgrad16_pair(const float* x3, const float* dy3, float* dw3, float* db3, const float* x2, const float* dy2, float* dw2, float* db2,
                 int64_t planes, int64_t batch, int64_t h, int64_t w_, float alpha, int swap_hw, int K, hipStream_t st);
}
// LLDWT_WGRAD16_PAIR=0 keeps two launches for the fp32 16 -> 16 weight gradients of a step (the small levels)
static const int g_w16_pair = [] { const char* e = getenv("LLDWT_WGRAD16_PAIR"); return e ? atoi(e) : 1; }();
// LLDWT_WGRAD_THIN=2 keeps the two separate launches of the thin (1 <-> 16) weight gradients of a step
static const int g_thin_pair = [] { const char* e = getenv("LLDWT_WGRAD_THIN"); return (e && !strcmp(e, "2")) ? 0 : 1; }();
// LLDWT_WGRAD16=f32 keeps the fp32-MFMA weight gradient of the 16 -> 16 lifting convs (k_wgrad16<5>); default: split-fp16
static const int g_wgrad16_f16 = [] { const char* e = getenv("LLDWT_WGRAD16"); return (e && !strcmp(e, "f32")) ? 0 : 1; }();
// smallest batch * h * w per plane at which the split-fp16 kernel takes the 16 -> 16 weight gradients (LLDWT_WGRAD16_MIN overrides)
static const int64_t g_wgrad16_min = [] { const char* e = getenv("LLDWT_WGRAD16_MIN"); return e ? (int64_t)atoll(e) : (int64_t)250000; }();

// LLDWT_BWD_LIFT=f32 keeps the three fp32-MFMA backward-data launches even when a backward pack is passed
static const int g_bwd_lift_f16 = [] { const char* e = getenv("LLDWT_BWD_LIFT"); return (e && !strcmp(e, "f32")) ? 0 : 1; }();
extern "C" int lldwt_bwd_lift_f16(void) { return g_bwd_lift_f16 && g_lift_mode == 1; }

extern "C" int64_t lldwt_pack_pblock_bwd_ws_bytes(int planes) { return (int64_t)sizeof(float) * lift_f16_bwd_scratch_floats(planes); }

// the "backward pack" of a tanh P/U block (C = 16, K = 5): the transposed, mirrored weights in the forward pack's split-fp16 layout,
// fp32 section zero (no biases in the backward chain).  packed: planes * lldwt_pblock_packed_floats(C, K) floats
extern "C" int lldwt_pack_pblock_bwd(const float* w1, const float* w2, const float* w3, const float* w4, float* packed,
                                     void* ws, int64_t ws_bytes, int planes, int C, int K, void* stream) {
    LLDWT_REQUIRE(planes > 0 && C == LF_C && K == LF_K, "pack_pblock_bwd: built for C=16, K=5 (got planes=%d C=%d K=%d)", planes, C, K);
    LLDWT_REQUIRE(w1 && w2 && w3 && w4 && packed && ws, "pack_pblock_bwd: null pointer");
    if (ws_bytes < lldwt_pack_pblock_bwd_ws_bytes(planes)) {
        set_error("pack_pblock_bwd: workspace %ld < %ld bytes", (long)ws_bytes, (long)lldwt_pack_pblock_bwd_ws_bytes(planes));
        return LLDWT_EWS;
    }
    const PackOff o = pack_off(C, K);
    if (hipMemsetAsync(packed, 0, sizeof(float) * (size_t)o.total * planes, (hipStream_t)stream) != hipSuccess) {
        set_error("pack_pblock_bwd: memset failed");
        return LLDWT_EHIP;
    }
    return lift_f16_pack_bwd(w1, w2, w3, w4, (float*)ws, packed, o.total, o.f16, planes, (hipStream_t)stream);
}

static int lift_step_bwd_impl(lldwt_view g_dst_out, lldwt_view g_dst_in, lldwt_view g_src, const float* saved_step,
                              int64_t planes, int64_t batch, int64_t h, int64_t w, const float* taps, float* dtaps,
                              const float* packed, int64_t packed_plane_stride, float* dw1, float* db1, float* dw2,
                              float* db2, float* dw3, float* db3, float* dw4, float* db4, int C, int K,
                              float res_weight, float sign, int vertical, int linear, void* ws, int64_t ws_bytes,
                              const float* packed_bwd, const float* taps_id, void* stream);

extern "C" int lldwt_lift_step_bwd(lldwt_view g_dst_out, lldwt_view g_dst_in, lldwt_view g_src, const float* saved_step,
                                   int64_t planes, int64_t batch, int64_t h, int64_t w, const float* taps, float* dtaps,
                                   const float* packed, int64_t packed_plane_stride, float* dw1, float* db1, float* dw2,
                                   float* db2, float* dw3, float* db3, float* dw4, float* db4, int C, int K,
                                   float res_weight, float sign, int vertical, int linear, void* ws, int64_t ws_bytes,
                                   void* stream) {
    return lift_step_bwd_impl(g_dst_out, g_dst_in, g_src, saved_step, planes, batch, h, w, taps, dtaps, packed, packed_plane_stride,
                              dw1, db1, dw2, db2, dw3, db3, dw4, db4, C, K, res_weight, sign, vertical, linear, ws, ws_bytes,
                              nullptr, nullptr, stream);
}

// the same with the backward-data chain (dt3, dpre2, dr, dsk) on the fused split-fp16 kernel's BWD mode: packed_bwd from
// lldwt_pack_pblock_bwd (stride = the forward pack's), taps_id = (planes, 3) floats (0, 1, 0)
extern "C" int lldwt_lift_step_bwd_f16(lldwt_view g_dst_out, lldwt_view g_dst_in, lldwt_view g_src, const float* saved_step,
                                       int64_t planes, int64_t batch, int64_t h, int64_t w, const float* taps, float* dtaps,
                                       const float* packed, int64_t packed_plane_stride, float* dw1, float* db1, float* dw2,
                                       float* db2, float* dw3, float* db3, float* dw4, float* db4, int C, int K,
                                       float res_weight, float sign, int vertical, int linear, void* ws, int64_t ws_bytes,
                                       const float* packed_bwd, const float* taps_id, void* stream) {
    LLDWT_REQUIRE(packed_bwd && taps_id, "lift_step_bwd_f16: null backward pack / identity taps");
    return lift_step_bwd_impl(g_dst_out, g_dst_in, g_src, saved_step, planes, batch, h, w, taps, dtaps, packed, packed_plane_stride,
                              dw1, db1, dw2, db2, dw3, db3, dw4, db4, C, K, res_weight, sign, vertical, linear, ws, ws_bytes,
                              packed_bwd, taps_id, stream);
}

static int lift_step_bwd_impl(lldwt_view g_dst_out, lldwt_view g_dst_in, lldwt_view g_src, const float* saved_step,
                              int64_t planes, int64_t batch, int64_t h, int64_t w, const float* taps, float* dtaps,
                              const float* packed, int64_t packed_plane_stride, float* dw1, float* db1, float* dw2,
                              float* db2, float* dw3, float* db3, float* dw4, float* db4, int C, int K,
                              float res_weight, float sign, int vertical, int linear, void* ws, int64_t ws_bytes,
                              const float* packed_bwd, const float* taps_id, void* stream) {
    LLDWT_REQUIRE(g_dst_out.p && g_dst_in.p && g_src.p && saved_step && taps && dtaps && packed && ws,
                  "lift_step_bwd: null pointer");
    LLDWT_REQUIRE(dw1 && db1 && dw2 && db2 && dw3 && db3 && dw4 && db4, "lift_step_bwd: null gradient pointer");
    LLDWT_REQUIRE(planes > 0 && batch > 0 && h > 0 && w > 0 && planes * batch <= 65535, "lift_step_bwd: bad dims");
    LLDWT_REQUIRE(C == 16 && (K == 3 || K == 5), "lift_step_bwd: built for C=16, K in {3,5} (got C=%d K=%d)", C, K);
    LLDWT_REQUIRE((int64_t)C * h * w < (int64_t)1 << 31, "lift_step_bwd: C*h*w = %ld exceeds the 32-bit tile addressing",
                  (long)((int64_t)C * h * w));
    const int64_t Z = planes * batch, n = Z * h * w;
    if (ws_bytes < lldwt_lift_step_bwd_ws_bytes(Z, h, w, C)) {
        set_error("lift_step_bwd: workspace %ld < %ld bytes", (long)ws_bytes, (long)lldwt_lift_step_bwd_ws_bytes(Z, h, w, C));
        return LLDWT_EWS;
    }
    hipStream_t st = (hipStream_t)stream;
    // saved layout of the forward step: [srcv | skip | t1 | t2 | t3]
    const float* srcv = saved_step;
    const float* skip = saved_step + n;
    const float* t1 = skip + n;
    const float* t2 = t1 + n * C;
    const float* t3 = t2 + n * C;
    float* g = (float*)ws;
    float* dsk = g + n;
    float* dt3 = dsk + n;
    float* dpre2 = dt3 + n * C;
    float* dr = dpre2 + n * C;
    int r;
    float* slots = dr + n * C;          // 128 per plane: max |dt3|, max |dpre2| (the split-fp16 weight gradient's dY scales)
    const bool wg16 = K == 5 && !linear && g_wgrad16_f16 && w % 4 == 0 && batch * h * w >= g_wgrad16_min;
    bool slots_ready = false;
    if (packed_bwd && K == LF_K && C == LF_C && !linear && g_bwd_lift_f16 && g_lift_mode == 1) {
        {   // g = G[dst_out], G[dst_in] = g; the same launch zeroes the |max| slots the fused launch fills (no memset per step)
            dim3 grid((unsigned)cdiv(w, 256), (unsigned)(h < 1024 ? h : 1024), (unsigned)Z);
            hipLaunchKernelGGL(k_lift_bwd_pre, grid, dim3(256), 0, st, cv(g_dst_out), g_dst_in, g, (int)h, (int)w,
                               wg16 ? slots : (float*)nullptr, (int)(128 * planes));
            if ((r = check_launch("lift_bwd_pre"))) return r;
        }
        slots_ready = wg16;             // the fused launch leaves both maxima in the slots: no pass over dt3 / dpre2 for them
        const LiftF16Bwd bw{g, t1, t2, dt3, dpre2, dr, dsk, slots_ready ? slots : nullptr};
        const PackOff o = pack_off(C, K);
        r = lift_f16_step_bwd(bw, Z, batch, h, w, taps_id, packed_bwd, packed_plane_stride, o.orient, o.f16, vertical, st);
    } else
    r = K == 5 ? launch_step_bwd<5>(g_dst_out, g_dst_in, g, dsk, dt3, dpre2, dr, t1, t2, Z, batch, h, w, packed,
                                        packed_plane_stride, vertical, linear, st)
                   : launch_step_bwd<3>(g_dst_out, g_dst_in, g, dsk, dt3, dpre2, dr, t1, t2, Z, batch, h, w, packed,
                                        packed_plane_stride, vertical, linear, st);
    if (r) return r;
    // weight gradients (scaled by sign*res_weight: net enters dst as sign*rw*net); row passes store (kh,kw) swapped
    const float alpha = sign * res_weight;
    const int swap = vertical ? 0 : 1;
    lldwt_conv_desc d;
    auto desc = [&](int cin, int cout) {
        d.cin = cin; d.cout = cout; d.K = K; d.groups = 1; d.act = LLDWT_ACT_NONE; d.upsample2 = 0; d.transposed = 0;
        d.tap_mask = (1u << (K * K)) - 1u; d.oc_block = cout; d.oc_stride = 0; d.oc_off = 0; d.ytot = cout;
        d.ic_block = 0; d.ic_stride = 0; d.ic_off = 0; d.xtot = 0; d.epi = 0;
    };
    if (g_thin_pair) {          // conv4's and conv1's gradients (both read what the backward-data chain left) in one launch
        if ((r = wgrad_thin_pair(t3, g, dw4, db4, skip, dr, dw1, db1, planes, batch, h, w, alpha, swap, K, st))) return r;
    } else {
        desc(C, 1);
        if ((r = lldwt_conv2d_wgrad_ex(t3, g, dw4, db4, &d, planes, batch, h, w, alpha, swap, stream))) return r;
    }
    desc(C, C);
    if (wg16) {
        // conv3 / conv2: their inputs t2 / t1 are tanh outputs (|x| <= 1): split-fp16 on the fp16 matrix cores (conv_wgrad_f16x3.hip).
        // Only where it wins (measured, 3 planes x 8 images, kernel alone: 228 vs 530 us at 256 x 512, 148 vs 272 at 256 x 256,
        // 111 vs 146 at 128 x 256, but 97 vs 71 at 128 x 128: below ~0.25 Mpixel per plane the fixed cost of a launch -- prologue,
        // tile reduction, 6 400 float atomics per workgroup -- weighs more than the matrix work saved)
        int8_t tap_of[25];
        for (int t = 0; t < 25; ++t) tap_of[t] = (int8_t)(swap ? (t % 5) * 5 + t / 5 : t);
        const int64_t ss = slots_ready ? 128 : 64;
        if ((r = wgrad16_f16x3(t2, dt3, dw3, db3, slots, ss, slots_ready, planes, batch, h, w, alpha, tap_of, st))) return r;
        if ((r = wgrad16_f16x3(t1, dpre2, dw2, db2, slots + (slots_ready ? 64 : 0), ss, slots_ready, planes, batch, h, w, alpha, tap_of,
                               st))) return r;
    } else if (g_w16_pair) {
        if ((r = wgrad16_pair(t2, dt3, dw3, db3, t1, dpre2, dw2, db2, planes, batch, h, w, alpha, swap, K, st))) return r;
    } else {
        if ((r = lldwt_conv2d_wgrad_ex(t2, dt3, dw3, db3, &d, planes, batch, h, w, alpha, swap, stream))) return r;
        if ((r = lldwt_conv2d_wgrad_ex(t1, dpre2, dw2, db2, &d, planes, batch, h, w, alpha, swap, stream))) return r;
    }
    if (!g_thin_pair) {
        desc(1, C);
        if ((r = lldwt_conv2d_wgrad_ex(skip, dr, dw1, db1, &d, planes, batch, h, w, alpha, swap, stream))) return r;
    }
    return lldwt_lift_bwd_fin(g, dsk, srcv, g_src, Z, batch, h, w, taps, dtaps, vertical, sign, res_weight, stream);
}

// workspace: [Lrow | Hrow | tmpL | tmpH] (each Z*(H/2)*W) + 2 LL ping-pong (Z*(H/2)*(W/2)) + step ws
extern "C" int64_t lldwt_lifting_ws_bytes(int64_t Z, int64_t H, int64_t W, int C) {
    const int64_t half = Z * (H / 2) * W;
    const int64_t quarter = Z * (H / 2) * (W / 2);
    return (int64_t)sizeof(float) * (4 * half + 2 * quarter) + lldwt_lift_step_ws_bytes(Z, H / 2, W, C);
}

static int lifting_args_ok(const char* who, int64_t planes, int64_t batch, int64_t H, int64_t W, int levels) {
    LLDWT_REQUIRE(planes > 0 && batch > 0 && levels > 0 && levels < 16, "%s: bad planes/batch/levels", who);
    LLDWT_REQUIRE(H > 0 && W > 0 && H % (1 << levels) == 0 && W % (1 << levels) == 0,
                  "%s: H=%ld W=%ld must be divisible by 2^levels=%d", who, (long)H, (long)W, 1 << levels);
    LLDWT_REQUIRE(planes * batch <= 65535, "%s: planes*batch exceeds grid.z", who);
    return 0;
}

extern "C" int lldwt_lifting_program(lldwt_lift_op* ops, int max_ops, int64_t Z, int64_t H, int64_t W, int levels,
                                     int different, int block_offset, int inverse, int scale, int C,
                                     int64_t* saved_floats) {
    LLDWT_REQUIRE(levels > 0 && levels < 16 && Z > 0 && H > 0 && W > 0, "lifting_program: bad arguments");
    return build_program(ops, ops ? max_ops : 0, Z, H, W, levels, different, block_offset, inverse, scale, C, saved_floats);
}

static int lifting_run(const char* who, int inverse, float* x, float* ll, float* const* yh, int64_t planes, int64_t batch,
                       int64_t H, int64_t W, int levels, const float* taps, const float* packed, int nblocks,
                       int block_offset, int different, int C, int K, float res_weight, int linear, const float* scale_nh,
                       const float* scale_nl, void* ws, int64_t ws_bytes, float* saved, void* stream) {
    int r = lifting_args_ok(who, planes, batch, H, W, levels);
    if (r) return r;
    LLDWT_REQUIRE(x && ll && yh && taps && packed && ws, "%s: null pointer", who);
    LLDWT_REQUIRE(nblocks >= 2 && block_offset >= 0 && block_offset + ((different && !inverse) ? 2 * levels : 2) <= nblocks,
                  "%s: block_offset=%d exceeds nblocks=%d", who, block_offset, nblocks);
    const int64_t Z = planes * batch;
    if (ws_bytes < lldwt_lifting_ws_bytes(Z, H, W, C)) {
        set_error("%s: workspace %ld < %ld bytes", who, (long)ws_bytes, (long)lldwt_lifting_ws_bytes(Z, H, W, C));
        return LLDWT_EWS;
    }
    const int64_t half = Z * (H / 2) * W, quarter = Z * (H / 2) * (W / 2);
    float* bases[B_YH0 + 16];
    float* base = (float*)ws;
    bases[B_X] = x;
    bases[B_LROW] = base;
    bases[B_HROW] = base + half;
    bases[B_TMPL] = base + 2 * half;
    bases[B_TMPH] = base + 3 * half;
    bases[B_LL0] = base + 4 * half;
    bases[B_LL1] = base + 4 * half + quarter;
    bases[B_LL] = ll;
    for (int i = 0; i < levels; ++i) bases[B_YH0 + i] = yh[i];
    lldwt_lift_op ops[16 * 3 * 8 + 8];
    const int n = build_program(ops, (int)(sizeof(ops) / sizeof(ops[0])), Z, H, W, levels, different, block_offset, inverse,
                                scale_nh != nullptr, C, nullptr);
    RunCtx c{Z, batch, planes, taps, packed, (int64_t)nblocks * 2 * pack_off(C, K).total, pack_off(C, K).total, C, K,
             linear, res_weight, scale_nh, scale_nl, base + 4 * half + 2 * quarter, saved, (hipStream_t)stream};
    return run_program(ops, n, bases, c);
}

extern "C" int lldwt_lifting_forward(const float* x, float* ll, float* const* yh, int64_t planes, int64_t batch,
                                     int64_t H, int64_t W, int levels, const float* taps, const float* packed,
                                     int nblocks, int block_offset, int different, int C, int K, float res_weight,
                                     int linear, const float* scale_nh, const float* scale_nl, void* ws,
                                     int64_t ws_bytes, void* stream) {
    return lifting_run("lifting_forward", 0, const_cast<float*>(x), ll, yh, planes, batch, H, W, levels, taps, packed,
                       nblocks, block_offset, different, C, K, res_weight, linear, scale_nh, scale_nl, ws, ws_bytes,
                       nullptr, stream);
}

extern "C" int lldwt_lifting_inverse(const float* ll, const float* const* yh, float* x, int64_t planes, int64_t batch,
                                     int64_t H, int64_t W, int levels, const float* taps, const float* packed,
                                     int nblocks, int block_offset, int C, int K, float res_weight, int linear,
                                     const float* scale_nh, const float* scale_nl, void* ws, int64_t ws_bytes,
                                     void* stream) {
    return lifting_run("lifting_inverse", 1, x, const_cast<float*>(ll), const_cast<float* const*>(yh), planes, batch, H, W,
                       levels, taps, packed, nblocks, block_offset, 0, C, K, res_weight, linear, scale_nh, scale_nl, ws,
                       ws_bytes, nullptr, stream);
}

// training variants: identical arithmetic, every step keeps (src, skip, t1, t2, t3) in `saved`
// (lldwt_lifting_program reports the size and the per-step offsets)
extern "C" int lldwt_lifting_forward_train(const float* x, float* ll, float* const* yh, int64_t planes, int64_t batch,
                                           int64_t H, int64_t W, int levels, const float* taps, const float* packed,
                                           int nblocks, int block_offset, int different, int C, int K, float res_weight,
                                           int linear, void* ws, int64_t ws_bytes, float* saved, void* stream) {
    return lldwt_lifting_forward_train_ex(x, ll, yh, planes, batch, H, W, levels, taps, packed, nblocks, block_offset, different,
                                          C, K, res_weight, linear, nullptr, nullptr, ws, ws_bytes, saved, stream);
}

// + the gains of config.scale == 1 (per plane; both null = no scaling): the scale ops keep their inputs in `saved` too
// (lldwt_lifting_program with scale = 1 gives the offsets)
extern "C" int lldwt_lifting_forward_train_ex(const float* x, float* ll, float* const* yh, int64_t planes, int64_t batch,
                                              int64_t H, int64_t W, int levels, const float* taps, const float* packed,
                                              int nblocks, int block_offset, int different, int C, int K, float res_weight,
                                              int linear, const float* scale_nh, const float* scale_nl, void* ws,
                                              int64_t ws_bytes, float* saved, void* stream) {
    LLDWT_REQUIRE(saved, "lifting_forward_train: null saved buffer");
    return lifting_run("lifting_forward_train", 0, const_cast<float*>(x), ll, yh, planes, batch, H, W, levels, taps, packed,
                       nblocks, block_offset, different, C, K, res_weight, linear, scale_nh, scale_nl, ws, ws_bytes, saved,
                       stream);
}

extern "C" int lldwt_lifting_inverse_train(const float* ll, const float* const* yh, float* x, int64_t planes,
                                           int64_t batch, int64_t H, int64_t W, int levels, const float* taps,
                                           const float* packed, int nblocks, int block_offset, int C, int K,
                                           float res_weight, int linear, void* ws, int64_t ws_bytes, float* saved,
                                           void* stream) {
    return lldwt_lifting_inverse_train_ex(ll, yh, x, planes, batch, H, W, levels, taps, packed, nblocks, block_offset, C, K,
                                          res_weight, linear, nullptr, nullptr, ws, ws_bytes, saved, stream);
}

extern "C" int lldwt_lifting_inverse_train_ex(const float* ll, const float* const* yh, float* x, int64_t planes,
                                              int64_t batch, int64_t H, int64_t W, int levels, const float* taps,
                                              const float* packed, int nblocks, int block_offset, int C, int K,
                                              float res_weight, int linear, const float* scale_nh, const float* scale_nl,
                                              void* ws, int64_t ws_bytes, float* saved, void* stream) {
    LLDWT_REQUIRE(saved, "lifting_inverse_train: null saved buffer");
    return lifting_run("lifting_inverse_train", 1, x, const_cast<float*>(ll), const_cast<float* const*>(yh), planes, batch,
                       H, W, levels, taps, packed, nblocks, block_offset, 0, C, K, res_weight, linear, scale_nh, scale_nl, ws,
                       ws_bytes, saved, stream);
}
