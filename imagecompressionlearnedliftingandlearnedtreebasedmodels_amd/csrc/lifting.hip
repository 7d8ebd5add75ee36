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

namespace lldwt {

static inline __host__ __device__ int pad16(int n) { return (n + 15) & ~15; }

struct PackOff {
    int w1, b1, w2, b2, w3, b3, w4, b4, orient, total;
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
    o.total = 2 * o.orient;
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
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < o.total; i += gridDim.x * blockDim.x) {
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
            if (ly >= R2 && ly < R2 + TH && lx >= R2 && lx < R2 + TW) skip_out[(z * h + gy) * (int64_t)w + gx] = v;
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
#pragma unroll
        for (int oc = 0; oc < C; ++oc) t1[oc][ly][lx] = in ? act_apply(acc[oc], act) : 0.f;
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
    for (int i = tid; i < T1H * T1W; i += NT) {
        const int ly = i / T1W, lx = i - ly * T1W;
        const int gy = y0 - R + ly, gx = x0 - R + lx;
        const bool in = gy >= 0 && gy < h && gx >= 0 && gx < w;
        const int64_t off = (int64_t)gy * w + gx;
#pragma unroll
        for (int c = 0; c < C; ++c) t[c][ly][lx] = in ? t3[(z * C + c) * cs + off] : 0.f;
    }
    __syncthreads();
    const int ly = tid / TW, lx = tid % TW;
    float a0 = pk[o.b4], a1 = pk[o.b4];
    for (int ic = 0; ic < C; ++ic) {
        const float* wi = pk + o.w4 + ic * K * K;
#pragma unroll
        for (int dy = 0; dy < K; ++dy)
#pragma unroll
            for (int dx = 0; dx < K; ++dx) {
                const float wv = wi[dy * K + dx];
                a0 = fmaf(wv, t[ic][ly + dy][lx + dx], a0);
                a1 = fmaf(wv, t[ic][ly + TH / 2 + dy][lx + dx], a1);
            }
    }
    const int gx = x0 + lx;
    if (gx < w) {
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int gy = y0 + ly + r * (TH / 2);
            if (gy < h) {
                const float net = r ? a1 : a0;
                const float sk = skip[z * cs + (int64_t)gy * w + gx];
                const float d = dst_in.p[z * dst_in.sz + (int64_t)gy * dst_in.sy + (int64_t)gx * dst_in.sx];
                dst_out.p[z * dst_out.sz + (int64_t)gy * dst_out.sy + (int64_t)gx * dst_out.sx] =
                    d + sign * (sk + rw * net);
            }
        }
    }
}

// out = in * s[plane]  (or / s[plane]) on views; config.scale == 1 only (wavelet_forward_v2.py:76-80)
__global__ void k_scale_view(CView in, lldwt_view out, int batch, int h, int w, const float* __restrict__ s,
                             int divide) {
    const int64_t z = blockIdx.z;
    const float f = s[z / batch];
    for (int y = blockIdx.y; y < h; y += gridDim.y)
        for (int x = blockIdx.x * blockDim.x + threadIdx.x; x < w; x += gridDim.x * blockDim.x) {
            const float v = in.p[z * in.sz + (int64_t)y * in.sy + (int64_t)x * in.sx];
            out.p[z * out.sz + (int64_t)y * out.sy + (int64_t)x * out.sx] = divide ? v / f : v * f;
        }
}

static inline CView cv(lldwt_view v) { return CView{v.p, v.sz, v.sy, v.sx}; }

template <int C, int K>
static int launch_step(lldwt_view src, lldwt_view dst_in, lldwt_view dst_out, int64_t Z, int64_t batch, int64_t h,
                       int64_t w, const float* taps, const float* packed, int64_t pstride, int vertical, float sign,
                       float rw, int linear, float* ws, hipStream_t st) {
    float* skip = ws;
    float* t2 = skip + Z * h * w;
    float* t3 = t2 + Z * C * h * w;
    dim3 grid((unsigned)cdiv(w, TW), (unsigned)cdiv(h, TH), (unsigned)Z), block(NT);
    hipLaunchKernelGGL((k_lift_a<C, K>), grid, block, 0, st, cv(src), skip, t2, (int)batch, (int)h, (int)w, taps, packed,
                       pstride, vertical, linear);
    hipLaunchKernelGGL((k_lift_b<C, K>), grid, block, 0, st, skip, t2, t3, (int)batch, (int)h, (int)w, packed, pstride,
                       vertical);
    hipLaunchKernelGGL((k_lift_c<C, K>), grid, block, 0, st, skip, t3, cv(dst_in), dst_out, (int)batch, (int)h, (int)w,
                       packed, pstride, vertical, sign, rw);
    return check_launch("lift_step");
}

static int dispatch_step(lldwt_view src, lldwt_view dst_in, lldwt_view dst_out, int64_t Z, int64_t batch, int64_t h,
                         int64_t w, const float* taps, const float* packed, int64_t pstride, int C, int K, int vertical,
                         float sign, float rw, int linear, float* ws, hipStream_t st) {
#define LLDWT_CASE(CC, KK_)                                                                                         \
    if (C == CC && K == KK_)                                                                                        \
        return launch_step<CC, KK_>(src, dst_in, dst_out, Z, batch, h, w, taps, packed, pstride, vertical, sign, rw, \
                                    linear, ws, st);
    LLDWT_CASE(16, 5)
    LLDWT_CASE(16, 3)
    LLDWT_CASE(8, 5)
    LLDWT_CASE(8, 3)
#undef LLDWT_CASE
    set_error("lift_step: unsupported (C=%d, K=%d); built: C in {8,16} x K in {3,5}", C, K);
    return LLDWT_EINVAL;
}

static inline lldwt_view mkview(float* p, int64_t sz, int64_t sy, int64_t sx) { return lldwt_view{p, sz, sy, sx}; }

struct LiftCtx {
    int64_t Z, batch;
    const float* taps;     // (4,planes,3)
    int64_t tstride;       // planes*3
    const float* packed;   // (planes,nblocks,2,total)
    int64_t pstride;       // floats per plane
    int64_t total;         // floats per block
    int C, K, linear;
    float rw;
    float* step_ws;
    hipStream_t st;
};

// 2-stage lifting on (L,H) half arrays given as views; writes final L to Lout, final H to Hout; tmpL/tmpH scratch
// views (contiguous, hh x ww).  blk = index of the first of the two (P,U) block pairs.
static int two_stage_forward(const LiftCtx& c, lldwt_view L, lldwt_view H, lldwt_view Lout, lldwt_view Hout,
                             lldwt_view tmpL, lldwt_view tmpH, int64_t hh, int64_t ww, int vertical, int blk) {
    auto P = [&](int s) { return c.packed + (int64_t)(blk + s) * 2 * c.total; };
    auto U = [&](int s) { return c.packed + (int64_t)(blk + s) * 2 * c.total + c.total; };
    int r;
    // wavelet_forward_v2.py:60-62  H = H + skip(L) + P0(skip)*rw
    if ((r = dispatch_step(L, H, tmpH, c.Z, c.batch, hh, ww, c.taps + 0 * c.tstride, P(0), c.pstride, c.C, c.K, vertical, 1.f, c.rw,
                           c.linear, c.step_ws, c.st)))
        return r;
    // :64-66  L = L + skip(H) + U0(skip)*rw
    if ((r = dispatch_step(tmpH, L, tmpL, c.Z, c.batch, hh, ww, c.taps + 1 * c.tstride, U(0), c.pstride, c.C, c.K, vertical, 1.f,
                           c.rw, c.linear, c.step_ws, c.st)))
        return r;
    // :68-70
    if ((r = dispatch_step(tmpL, tmpH, Hout, c.Z, c.batch, hh, ww, c.taps + 2 * c.tstride, P(1), c.pstride, c.C, c.K, vertical, 1.f,
                           c.rw, c.linear, c.step_ws, c.st)))
        return r;
    // :72-74
    return dispatch_step(Hout, tmpL, Lout, c.Z, c.batch, hh, ww, c.taps + 3 * c.tstride, U(1), c.pstride, c.C, c.K, vertical, 1.f,
                         c.rw, c.linear, c.step_ws, c.st);
}

// inverse (wavelet_inverse_v2.py:76-90): inputs L,H (hh x ww); final L' -> Lout, final H' -> Hout
static int two_stage_inverse(const LiftCtx& c, lldwt_view L, lldwt_view H, lldwt_view Lout, lldwt_view Hout,
                             lldwt_view tmpL, lldwt_view tmpH, int64_t hh, int64_t ww, int vertical, int blk) {
    auto P = [&](int s) { return c.packed + (int64_t)(blk + s) * 2 * c.total; };
    auto U = [&](int s) { return c.packed + (int64_t)(blk + s) * 2 * c.total + c.total; };
    int r;
    if ((r = dispatch_step(H, L, tmpL, c.Z, c.batch, hh, ww, c.taps + 3 * c.tstride, U(1), c.pstride, c.C, c.K, vertical, -1.f, c.rw,
                           c.linear, c.step_ws, c.st)))
        return r;
    if ((r = dispatch_step(tmpL, H, tmpH, c.Z, c.batch, hh, ww, c.taps + 2 * c.tstride, P(1), c.pstride, c.C, c.K, vertical, -1.f,
                           c.rw, c.linear, c.step_ws, c.st)))
        return r;
    if ((r = dispatch_step(tmpH, tmpL, Lout, c.Z, c.batch, hh, ww, c.taps + 1 * c.tstride, U(0), c.pstride, c.C, c.K, vertical, -1.f,
                           c.rw, c.linear, c.step_ws, c.st)))
        return r;
    return dispatch_step(Lout, tmpH, Hout, c.Z, c.batch, hh, ww, c.taps + 0 * c.tstride, P(0), c.pstride, c.C, c.K, vertical, -1.f,
                         c.rw, c.linear, c.step_ws, c.st);
}

static void scale_view(const LiftCtx& c, lldwt_view v, int64_t hh, int64_t ww, const float* s, int divide) {
    dim3 grid((unsigned)cdiv(ww, 256), (unsigned)(hh < 1024 ? hh : 1024), (unsigned)c.Z);
    hipLaunchKernelGGL(k_scale_view, grid, dim3(256), 0, c.st, cv(v), v, (int)c.batch, (int)hh, (int)ww, s, divide);
}

}  // namespace lldwt

using namespace lldwt;

extern "C" int64_t lldwt_pblock_packed_floats(int C, int K) { return pack_off(C, K).total; }

extern "C" int lldwt_pack_pblock(const float* w1, const float* b1, const float* w2, const float* b2, const float* w3,
                                 const float* b3, const float* w4, const float* b4, float* packed, int planes, int C,
                                 int K, void* stream) {
    LLDWT_REQUIRE(planes > 0 && C > 0 && (K == 3 || K == 5), "pack_pblock: bad planes/C/K (%d,%d,%d)", planes, C, K);
    LLDWT_REQUIRE(w1 && b1 && w2 && b2 && w3 && b3 && w4 && b4 && packed, "pack_pblock: null pointer");
    const PackOff o = pack_off(C, K);
    dim3 grid((unsigned)cdiv(o.total, 256), (unsigned)planes);
    hipLaunchKernelGGL(k_pack_pblock, grid, dim3(256), 0, (hipStream_t)stream, w1, b1, w2, b2, w3, b3, w4, b4, packed, C,
                       K);
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
                         res_weight, linear, (float*)ws, (hipStream_t)stream);
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

extern "C" int lldwt_lifting_forward(const float* x, float* ll, float* const* yh, int64_t planes, int64_t batch,
                                     int64_t H, int64_t W, int levels, const float* taps, const float* packed,
                                     int nblocks, int block_offset, int different, int C, int K, float res_weight,
                                     int linear, const float* scale_nh, const float* scale_nl, void* ws,
                                     int64_t ws_bytes, void* stream) {
    int r = lifting_args_ok("lifting_forward", planes, batch, H, W, levels);
    LLDWT_REQUIRE(nblocks >= 2 && block_offset >= 0 && block_offset + (different ? 2 * levels : 2) <= nblocks,
                  "lifting_forward: block_offset=%d (+%d) exceeds nblocks=%d", block_offset, different ? 2 * levels : 2, nblocks);
    if (r) return r;
    LLDWT_REQUIRE(x && ll && yh && taps && packed && ws, "lifting_forward: null pointer");
    const int64_t Z = planes * batch;
    if (ws_bytes < lldwt_lifting_ws_bytes(Z, H, W, C)) {
        set_error("lifting_forward: workspace %ld < %ld bytes", (long)ws_bytes, (long)lldwt_lifting_ws_bytes(Z, H, W, C));
        return LLDWT_EWS;
    }
    const int64_t half = Z * (H / 2) * W, quarter = Z * (H / 2) * (W / 2);
    float* Lrow = (float*)ws;
    float* Hrow = Lrow + half;
    float* tmpL = Hrow + half;
    float* tmpH = tmpL + half;
    float* llbuf[2] = {tmpH + half, tmpH + half + quarter};
    LiftCtx c{Z, batch, taps, planes * 3, packed, (int64_t)nblocks * 2 * pack_off(C, K).total, pack_off(C, K).total, C, K, linear,
              res_weight, llbuf[1] + quarter, (hipStream_t)stream};
    const float* cur = x;
    for (int lev = 0; lev < levels; ++lev) {
        const int64_t h = H >> lev, w = W >> lev, hh = h / 2, wh = w / 2;
        const int blk = block_offset + (different ? lev * 2 : 0);
        float* X = const_cast<float*>(cur);
        // rows: L = x[0::2], H = x[1::2]  (wavelet_forward_v2.py:27-29)
        lldwt_view A = mkview(X, h * w, 2 * w, 1), B = mkview(X + w, h * w, 2 * w, 1);
        lldwt_view vL = mkview(Lrow, hh * w, w, 1), vH = mkview(Hrow, hh * w, w, 1);
        lldwt_view tL = mkview(tmpL, hh * w, w, 1), tH = mkview(tmpH, hh * w, w, 1);
        if ((r = two_stage_forward(c, A, B, vL, vH, tL, tH, hh, w, 1, blk))) return r;
        if (scale_nh) { scale_view(c, vH, hh, w, scale_nh, 0); scale_view(c, vL, hh, w, scale_nl, 0); }
        // columns of L: LL = L[:, 0::2], HL = L[:, 1::2]  (:32-39)
        float* llout = (lev == levels - 1) ? ll : llbuf[lev & 1];
        float* y = yh[lev];
        const int64_t sub = hh * wh;
        lldwt_view vLL = mkview(llout, sub, wh, 1);
        lldwt_view vLH = mkview(y, 3 * sub, wh, 1), vHL = mkview(y + sub, 3 * sub, wh, 1),
                   vHH = mkview(y + 2 * sub, 3 * sub, wh, 1);
        lldwt_view t2L = mkview(tmpL, sub, wh, 1), t2H = mkview(tmpH, sub, wh, 1);
        lldwt_view Le = mkview(Lrow, hh * w, w, 2), Lo = mkview(Lrow + 1, hh * w, w, 2);
        if ((r = two_stage_forward(c, Le, Lo, vLL, vHL, t2L, t2H, hh, wh, 0, blk))) return r;
        if (scale_nh) { scale_view(c, vHL, hh, wh, scale_nh, 0); scale_view(c, vLL, hh, wh, scale_nl, 0); }
        // columns of H: LH = H[:, 0::2], HH = H[:, 1::2]  (:43-51)
        lldwt_view He = mkview(Hrow, hh * w, w, 2), Ho = mkview(Hrow + 1, hh * w, w, 2);
        if ((r = two_stage_forward(c, He, Ho, vLH, vHH, t2L, t2H, hh, wh, 0, blk))) return r;
        if (scale_nh) { scale_view(c, vHH, hh, wh, scale_nh, 0); scale_view(c, vLH, hh, wh, scale_nl, 0); }
        cur = llout;
    }
    return check_launch("lifting_forward");
}

extern "C" int lldwt_lifting_inverse(const float* ll, const float* const* yh, float* x, int64_t planes, int64_t batch,
                                     int64_t H, int64_t W, int levels, const float* taps, const float* packed,
                                     int nblocks, int block_offset, int C, int K, float res_weight, int linear,
                                     const float* scale_nh, const float* scale_nl, void* ws, int64_t ws_bytes,
                                     void* stream) {
    int r = lifting_args_ok("lifting_inverse", planes, batch, H, W, levels);
    LLDWT_REQUIRE(nblocks >= 2 && block_offset >= 0 && block_offset + 2 <= nblocks,
                  "lifting_inverse: block_offset=%d exceeds nblocks=%d", block_offset, nblocks);
    if (r) return r;
    LLDWT_REQUIRE(x && ll && yh && taps && packed && ws, "lifting_inverse: null pointer");
    const int64_t Z = planes * batch;
    if (ws_bytes < lldwt_lifting_ws_bytes(Z, H, W, C)) {
        set_error("lifting_inverse: workspace %ld < %ld bytes", (long)ws_bytes, (long)lldwt_lifting_ws_bytes(Z, H, W, C));
        return LLDWT_EWS;
    }
    const int64_t half = Z * (H / 2) * W, quarter = Z * (H / 2) * (W / 2);
    float* Lrow = (float*)ws;
    float* Hrow = Lrow + half;
    float* tmpL = Hrow + half;
    float* tmpH = tmpL + half;
    float* llbuf[2] = {tmpH + half, tmpH + half + quarter};
    LiftCtx c{Z, batch, taps, planes * 3, packed, (int64_t)nblocks * 2 * pack_off(C, K).total, pack_off(C, K).total, C, K, linear,
              res_weight, llbuf[1] + quarter, (hipStream_t)stream};
    const int blk = block_offset;   // lifting_dwt_nets.py:718-722: every inverse level uses the same pair
    const float* cur = ll;
    for (int lev = levels - 1; lev >= 0; --lev) {
        const int64_t h = H >> lev, w = W >> lev, hh = h / 2, wh = w / 2;
        const int64_t sub = hh * wh;
        float* y = const_cast<float*>(yh[lev]);
        lldwt_view vLL = mkview(const_cast<float*>(cur), sub, wh, 1);
        lldwt_view vLH = mkview(y, 3 * sub, wh, 1), vHL = mkview(y + sub, 3 * sub, wh, 1),
                   vHH = mkview(y + 2 * sub, 3 * sub, wh, 1);
        lldwt_view t2L = mkview(tmpL, sub, wh, 1), t2H = mkview(tmpH, sub, wh, 1);
        float* out = (lev == 0) ? x : llbuf[lev & 1];
        // scaled copies when config.scale == 1 (wavelet_inverse_v2.py:70-74): use the tail of Lrow/Hrow as scratch
        lldwt_view inL = vLL, inH = vHL;
        float* sc0 = llbuf[(lev & 1) ^ 1];     // free ping-pong buffer (>= sub floats per z)
        if (scale_nh) {
            lldwt_view sL = mkview(sc0, sub, wh, 1), sH = mkview(tmpH + half - Z * sub, sub, wh, 1);
            hipLaunchKernelGGL(k_scale_view, dim3((unsigned)cdiv(wh, 256), (unsigned)(hh < 1024 ? hh : 1024), (unsigned)Z),
                               dim3(256), 0, c.st, cv(vLL), sL, (int)batch, (int)hh, (int)wh, scale_nl, 1);
            hipLaunchKernelGGL(k_scale_view, dim3((unsigned)cdiv(wh, 256), (unsigned)(hh < 1024 ? hh : 1024), (unsigned)Z),
                               dim3(256), 0, c.st, cv(vHL), sH, (int)batch, (int)hh, (int)wh, scale_nh, 1);
            inL = sL; inH = sH;
        }
        // (LL,HL) -> L : even/odd columns of Lrow  (wavelet_inverse_v2.py:21-26)
        lldwt_view Le = mkview(Lrow, hh * w, w, 2), Lo = mkview(Lrow + 1, hh * w, w, 2);
        if ((r = two_stage_inverse(c, inL, inH, Le, Lo, t2L, t2H, hh, wh, 0, blk))) return r;
        inL = vLH; inH = vHH;
        if (scale_nh) {
            lldwt_view sL = mkview(sc0, sub, wh, 1), sH = mkview(tmpH + half - Z * sub, sub, wh, 1);
            hipLaunchKernelGGL(k_scale_view, dim3((unsigned)cdiv(wh, 256), (unsigned)(hh < 1024 ? hh : 1024), (unsigned)Z),
                               dim3(256), 0, c.st, cv(vLH), sL, (int)batch, (int)hh, (int)wh, scale_nl, 1);
            hipLaunchKernelGGL(k_scale_view, dim3((unsigned)cdiv(wh, 256), (unsigned)(hh < 1024 ? hh : 1024), (unsigned)Z),
                               dim3(256), 0, c.st, cv(vHH), sH, (int)batch, (int)hh, (int)wh, scale_nh, 1);
            inL = sL; inH = sH;
        }
        // (LH,HH) -> H  (:28-33)
        lldwt_view He = mkview(Hrow, hh * w, w, 2), Ho = mkview(Hrow + 1, hh * w, w, 2);
        if ((r = two_stage_inverse(c, inL, inH, He, Ho, t2L, t2H, hh, wh, 0, blk))) return r;
        // (L,H) -> rows of the output (:35-37)
        lldwt_view vL = mkview(Lrow, hh * w, w, 1), vH = mkview(Hrow, hh * w, w, 1);
        lldwt_view tL = mkview(tmpL, hh * w, w, 1), tH = mkview(tmpH, hh * w, w, 1);
        lldwt_view A = mkview(out, h * w, 2 * w, 1), B = mkview(out + w, h * w, 2 * w, 1);
        if (scale_nh) { scale_view(c, vL, hh, w, scale_nl, 1); scale_view(c, vH, hh, w, scale_nh, 1); }
        if ((r = two_stage_inverse(c, vL, vH, A, B, tL, tH, hh, w, 1, blk))) return r;
        cur = out;
    }
    return check_launch("lifting_inverse");
}
