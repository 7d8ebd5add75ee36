// cdf97.hip -- fixed CDF 9/7 (bior4.4) 2-D DWT, periodization, for gfx950.  HBM-bound (~55 MAC/px).
// Replaces pytorch_wavelets.DWTForward/DWTInverse(mode='periodization', wave='bior4.4') as used by
// DWTPytorchWaveletsLayer (graphs/layers/lifting_dwt_nets.py:228-231,250,274); filter taps are the reference's own
// table get_cdf97_filters (lifting_dwt_nets.py:415-418).
//   analysis : lo[k] = sum_m dec_lo[m] * x[(2k + 5 - m) mod N]      (same for hi)
//   synthesis: x[n]  = sum_t [ (n+4-t) mod N even ] ( lo[((n+4-t) mod N)/2] * rec_lo[t] + hi[..] * rec_hi[t] )
#include "common.h"
#include <atomic>
#include <cstdlib>
#include <mutex>

namespace lldwt {

__constant__ float c_dec_lo[10] = {0.0f, 0.037828455507264f, -0.023849465019557f, -0.110624404418437f, 0.377402855612831f,
                                   0.852698679008894f, 0.377402855612831f, -0.110624404418437f, -0.023849465019557f,
                                   0.037828455507264f};
__constant__ float c_dec_hi[10] = {0.0f, -0.064538882628697f, 0.040689417609164f, 0.418092273221617f, -0.788485616405583f,
                                   0.418092273221617f, 0.040689417609164f, -0.064538882628697f, 0.0f, 0.0f};
__constant__ float c_rec_lo[10] = {0.0f, -0.064538882628697f, -0.040689417609164f, 0.418092273221617f, 0.788485616405583f,
                                   0.418092273221617f, -0.040689417609164f, -0.064538882628697f, 0.0f, 0.0f};
__constant__ float c_rec_hi[10] = {0.0f, -0.037828455507264f, -0.023849465019557f, 0.110624404418437f, 0.377402855612831f,
                                   -0.852698679008894f, 0.377402855612831f, 0.110624404418437f, -0.023849465019557f,
                                   -0.037828455507264f};

struct V3 {   // (Z, rows, cols) strided view
    float* p;
    int64_t sz, sy, sx;
};

typedef float floatx2 __attribute__((ext_vector_type(2)));
struct __attribute__((packed, aligned(4))) f4u { float x, y, z, w; };
struct __attribute__((packed, aligned(4))) f2u { float x, y; };

// Global accesses of the tile bodies.  COH = the buffer is written and read by different workgroups of the SAME launch
// (the LL planes between the levels of the one-launch transforms): agent-scope relaxed atomics, i.e. sc1 loads / stores
// that go past the per-XCD L2, so the tile flags need no L2 write-back / invalidate around them.  8-byte pieces: the
// callers guarantee 8-byte alignment on these paths.
template <bool COH> __device__ __forceinline__ float ld1(const float* p) {
    if constexpr (COH)
        return __uint_as_float(__hip_atomic_load(reinterpret_cast<const unsigned*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    else
        return *p;
}
template <bool COH> __device__ __forceinline__ f2u ld2(const float* p) {
    if constexpr (COH) {
        const unsigned long long u = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED,
                                                       __HIP_MEMORY_SCOPE_AGENT);
        return f2u{__uint_as_float((unsigned)u), __uint_as_float((unsigned)(u >> 32))};
    } else {
        return *reinterpret_cast<const f2u*>(p);
    }
}
template <bool COH> __device__ __forceinline__ f4u ld4(const float* p) {
    if constexpr (COH) {
        const f2u a = ld2<true>(p), b = ld2<true>(p + 2);
        return f4u{a.x, a.y, b.x, b.y};
    } else {
        return *reinterpret_cast<const f4u*>(p);
    }
}
template <bool COH> __device__ __forceinline__ void st1(float* p, float v) {
    if constexpr (COH)
        __hip_atomic_store(reinterpret_cast<unsigned*>(p), __float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else
        *p = v;
}
template <bool COH> __device__ __forceinline__ void st2(float* p, float x, float y) {
    if constexpr (COH)
        __hip_atomic_store(reinterpret_cast<unsigned long long*>(p),
                           (unsigned long long)__float_as_uint(x) | (unsigned long long)__float_as_uint(y) << 32, __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
    else
        *reinterpret_cast<f2u*>(p) = f2u{x, y};
}
template <bool COH> __device__ __forceinline__ void st4(float* p, float x, float y, float z, float w) {
    if constexpr (COH) {
        st2<true>(p, x, y);
        st2<true>(p + 2, z, w);
    } else {
        *reinterpret_cast<f4u*>(p) = f4u{x, y, z, w};
    }
}

// analysis along `axis` (0 = rows/height, 1 = cols/width).  in: (Z,h,w); lo,hi: half size along axis.
// adj == 0: analysis (dec filters, x[(2k + 5 - m) mod N]);  adj == 1: ADJOINT of the synthesis k_sfb, needed by the
// backward pass of the inverse transform (rec filters, g[(2k - 4 + m) mod N])
__global__ __launch_bounds__(256) void k_afb(V3 in, V3 lo, V3 hi, int h, int w, int axis, int adj) {
    const int64_t z = blockIdx.z;
    const int oh = axis == 0 ? h / 2 : h, ow = axis == 1 ? w / 2 : w;
    const int N = axis == 0 ? h : w;
    for (int y = blockIdx.y; y < oh; y += gridDim.y)
        for (int x = blockIdx.x * blockDim.x + threadIdx.x; x < ow; x += gridDim.x * blockDim.x) {
            const int k = axis == 0 ? y : x;
            float a = 0.f, d = 0.f;
#pragma unroll
            for (int m = 0; m < 10; ++m) {
                int n = (adj ? 2 * k - 4 + m : 2 * k + 5 - m) % N;
                if (n < 0) n += N;
                const int yy = axis == 0 ? n : y, xx = axis == 1 ? n : x;
                const float v = in.p[z * in.sz + (int64_t)yy * in.sy + (int64_t)xx * in.sx];
                a = fmaf(adj ? c_rec_lo[m] : c_dec_lo[m], v, a);
                d = fmaf(adj ? c_rec_hi[m] : c_dec_hi[m], v, d);
            }
            lo.p[z * lo.sz + (int64_t)y * lo.sy + (int64_t)x * lo.sx] = a;
            hi.p[z * hi.sz + (int64_t)y * hi.sy + (int64_t)x * hi.sx] = d;
        }
}

// synthesis along `axis`.  lo,hi: half size along axis; out: (Z,h,w).
// adj == 0: synthesis (rec filters, q = (n + 4 - t) mod N);  adj == 1: ADJOINT of the analysis k_afb, needed by the backward
// pass of the forward transform (dec filters, q = (n - 5 + t) mod N)
__global__ __launch_bounds__(256) void k_sfb(V3 lo, V3 hi, V3 out, int h, int w, int axis, int adj) {
    const int64_t z = blockIdx.z;
    const int N = axis == 0 ? h : w;
    for (int y = blockIdx.y; y < h; y += gridDim.y)
        for (int x = blockIdx.x * blockDim.x + threadIdx.x; x < w; x += gridDim.x * blockDim.x) {
            const int n = axis == 0 ? y : x;
            float acc = 0.f;
#pragma unroll
            for (int t = 0; t < 10; ++t) {
                int q = (adj ? n - 5 + t : n + 4 - t) % N;
                if (q < 0) q += N;
                if ((q & 1) == 0) {
                    const int k = q >> 1;
                    const int yy = axis == 0 ? k : y, xx = axis == 1 ? k : x;
                    acc = fmaf(lo.p[z * lo.sz + (int64_t)yy * lo.sy + (int64_t)xx * lo.sx], adj ? c_dec_lo[t] : c_rec_lo[t], acc);
                    acc = fmaf(hi.p[z * hi.sz + (int64_t)yy * hi.sy + (int64_t)xx * hi.sx], adj ? c_dec_hi[t] : c_rec_hi[t], acc);
                }
            }
            out.p[z * out.sz + (int64_t)y * out.sy + (int64_t)x * out.sx] = acc;
        }
}

// ---- fused per-level kernels (adj == 0): both passes of one level through LDS, one read and one write of the data ------
// forward: a workgroup produces a CT x CT tile of each of the 4 subbands from a (2CT+8)^2 input patch (periodic wrap)
constexpr int CT = 32;                     // subband tile edge
constexpr int CIN = 2 * CT + 8;            // input patch edge (72)

constexpr int CDF_SMEM_FWD = CIN * (CIN + 1);      // floats of LDS of the generic forward body (>= 2 CIN (CT + 1): sL, sH overlay)

template <bool CI = false, bool CO = false>
__device__ __forceinline__ void fwd_level_body(float* smem, int64_t z, int by, int bx, V3 in, V3 ll, V3 lh, V3 hl, V3 vhh,
                                               int h, int w) {
    float (*sin)[CIN + 1] = reinterpret_cast<float (*)[CIN + 1]>(smem);
    float (*sL)[CT + 1] = reinterpret_cast<float (*)[CT + 1]>(smem);                         // over the patch, see below
    float (*sH)[CT + 1] = reinterpret_cast<float (*)[CT + 1]>(smem + CIN * (CT + 1));
    const int ky0 = by * CT, kx0 = bx * CT;
    const int tid = threadIdx.x;
    // input patch rows (2*ky0 - 4 + ly) mod h, cols (2*kx0 - 4 + lx) mod w
    // periodic wrap by conditional add/sub (an integer modulo per element made this phase instruction-bound); the
    // modulo is only needed when the image is smaller than the patch (deep levels, negligible work)
    const bool small = h < CIN || w < CIN;
    const float* inz = in.p + z * in.sz;
    // all loads of a thread are issued before the first LDS store (a load -> store loop waits vmcnt(0) per element and
    // serialises the HBM latency: measured 68 % of the wave cycles parked)
    constexpr int NLD = (CIN * CIN + 255) / 256;
    float v[NLD];
#pragma unroll
    for (int r = 0; r < NLD; ++r) {
        const int i = tid + r * 256;
        const int ly = i / CIN, lx = i - ly * CIN;
        int gy = 2 * ky0 - 4 + ly, gx = 2 * kx0 - 4 + lx;
        if (small) {
            gy %= h; gx %= w;
            if (gy < 0) gy += h;
            if (gx < 0) gx += w;
        } else {
            gy += gy < 0 ? h : (gy >= h ? -h : 0);
            gx += gx < 0 ? w : (gx >= w ? -w : 0);
        }
        v[r] = i < CIN * CIN ? ld1<CI>(inz + gy * (int)in.sy + gx * (int)in.sx) : 0.f;
    }
#pragma unroll
    for (int r = 0; r < NLD; ++r) {
        const int i = tid + r * 256;
        if (i < CIN * CIN) sin[i / CIN][i % CIN] = v[r];
    }
    __syncthreads();
    // width pass: lo/hi[ly][c] = sum_m dec[m] * sin[ly][2c + 9 - m]
    // (results held in registers across a barrier: sL / sH overlay the patch, which halves the LDS of a workgroup)
    constexpr int NW = CIN * CT / 256;                 // 9 outputs of each filter per thread, exactly
    static_assert(CIN * CT % 256 == 0, "width pass of the generic forward body");
    float wa[NW], wd[NW];
#pragma unroll
    for (int q = 0; q < NW; ++q) {
        const int i = tid + q * 256;
        const int ly = i / CT, c = i - ly * CT;
        float a = 0.f, d = 0.f;
#pragma unroll
        for (int m = 0; m < 10; ++m) {
            const float v = sin[ly][2 * c + 9 - m];
            a = fmaf(c_dec_lo[m], v, a);
            d = fmaf(c_dec_hi[m], v, d);
        }
        wa[q] = a;
        wd[q] = d;
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < NW; ++q) {
        const int i = tid + q * 256;
        const int ly = i / CT, c = i - ly * CT;
        sL[ly][c] = wa[q];
        sH[ly][c] = wd[q];
    }
    __syncthreads();
    // height pass on both halves
    const int hh = h / 2, wh = w / 2;
    for (int i = tid; i < CT * CT; i += 256) {
        const int r = i / CT, c = i - r * CT;
        const int ky = ky0 + r, kx = kx0 + c;
        if (ky >= hh || kx >= wh) continue;
        float a0 = 0.f, d0 = 0.f, a1 = 0.f, d1 = 0.f;
#pragma unroll
        for (int m = 0; m < 10; ++m) {
            const float vl = sL[2 * r + 9 - m][c], vh = sH[2 * r + 9 - m][c];
            a0 = fmaf(c_dec_lo[m], vl, a0);
            d0 = fmaf(c_dec_hi[m], vl, d0);
            a1 = fmaf(c_dec_lo[m], vh, a1);
            d1 = fmaf(c_dec_hi[m], vh, d1);
        }
        const int o = ky * (int)ll.sy + kx;                                   // the four outputs share row/col strides
        st1<CO>(ll.p + z * ll.sz + o, a0);     // low width, low height
        lh.p[z * lh.sz + o] = d0;     // low width, high height
        hl.p[z * hl.sz + o] = a1;
        vhh.p[z * vhh.sz + o] = d1;
    }
}

__global__ __launch_bounds__(256) void k_cdf97_fwd_level(V3 in, V3 ll, V3 lh, V3 hl, V3 vhh, int h, int w) {
    __shared__ float smem[CDF_SMEM_FWD];
    fwd_level_body(smem, blockIdx.z, blockIdx.y, blockIdx.x, in, ll, lh, hl, vhh, h, w);
}

// Fast variant of the forward level for rows of contiguous pixels, w % 4 == 0 and h, w >= CIN (every level of a 512^2
// or larger image but the deepest ones).  Same arithmetic as k_cdf97_fwd_level (the zero taps of the table are skipped),
// organised for the memory pipe:
//   * the input patch arrives as dwordx4 loads (the patch starts at a multiple of 4 pixels, so a vector never straddles
//     the periodic wrap) and is de-interleaved into even / odd columns in LDS: the stride-2 reads of the width pass
//     become unit-stride (they were 2-way bank conflicts);
//   * every thread produces two adjacent outputs, reading and writing LDS as 8-byte words.
constexpr float DEC_LO[10] = {0.0f, 0.037828455507264f, -0.023849465019557f, -0.110624404418437f, 0.377402855612831f,
                              0.852698679008894f, 0.377402855612831f, -0.110624404418437f, -0.023849465019557f,
                              0.037828455507264f};
constexpr float DEC_HI[10] = {0.0f, -0.064538882628697f, 0.040689417609164f, 0.418092273221617f, -0.788485616405583f,
                              0.418092273221617f, 0.040689417609164f, -0.064538882628697f, 0.0f, 0.0f};
constexpr int CHP = CIN / 2 + 2;           // pitch of the even / odd column planes (38: even, rows 8-byte aligned)
constexpr int CLP = CT + 2;                // pitch of the width-pass outputs (34)

constexpr int CDF_SMEM_FWDV = 2 * CIN * CHP;       // floats of LDS of the fast forward body (21 888 B; >= 2 CIN CLP: sL, sH overlay)

template <bool CI = false, bool CO = false>
__device__ __forceinline__ void fwd_level_v_body(float* smem, int64_t z, int by, int bx, V3 in, V3 ll, V3 lh, V3 hl, V3 vhh,
                                                 int h, int w) {
    float (*se)[CHP] = reinterpret_cast<float (*)[CHP]>(smem);
    float (*so)[CHP] = reinterpret_cast<float (*)[CHP]>(smem + CIN * CHP);
    float (*sL)[CLP] = reinterpret_cast<float (*)[CLP]>(smem);                                // over se / so, see below
    float (*sH)[CLP] = reinterpret_cast<float (*)[CLP]>(smem + CIN * CLP);
    const int ky0 = by * CT, kx0 = bx * CT;
    const int tid = threadIdx.x;
    const float* inz = in.p + z * in.sz;
    constexpr int VR = CIN / 4;                        // vectors per patch row (18)
    constexpr int NLV = (CIN * VR + 255) / 256;        // 6
    f4u v[NLV];
#pragma unroll
    for (int r = 0; r < NLV; ++r) {
        const int i = tid + r * 256;
        const int ly = i / VR, vx = i - ly * VR;
        int gy = 2 * ky0 - 4 + ly, gx = 2 * kx0 - 4 + 4 * vx;
        gy += gy < 0 ? h : (gy >= h ? -h : 0);
        gx += gx < 0 ? w : (gx >= w ? -w : 0);
        v[r] = ld4<CI>(inz + (i < CIN * VR ? (int64_t)gy * in.sy + gx : 0));
    }
#pragma unroll
    for (int r = 0; r < NLV; ++r) {
        const int i = tid + r * 256;
        if (i < CIN * VR) {
            const int ly = i / VR, vx = i - ly * VR;
            *reinterpret_cast<floatx2*>(&se[ly][2 * vx]) = floatx2{v[r].x, v[r].z};
            *reinterpret_cast<floatx2*>(&so[ly][2 * vx]) = floatx2{v[r].y, v[r].w};
        }
    }
    __syncthreads();
    // width pass, two adjacent outputs per thread: lo/hi[ly][c] = sum_m dec[m] * patch[ly][2c + 9 - m];
    // column 2c+9-m is even column c+4-(m-1)/2 for odd m, odd column c+4-m/2 for even m
    // (results held in registers across a barrier: sL / sH overlay the even / odd planes -- 21.9 KB of LDS per workgroup)
    constexpr int NWV = (CIN * (CT / 2) + 255) / 256;  // 5 (the last one half empty)
    floatx2 wa[NWV], wd[NWV];
#pragma unroll
    for (int q = 0; q < NWV; ++q) {
        const int i0 = tid + q * 256;
        const int i = i0 < CIN * (CT / 2) ? i0 : CIN * (CT / 2) - 1;
        const int ly = i / (CT / 2), c = 2 * (i - ly * (CT / 2));
        float e[6], o[6];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const floatx2 te = *reinterpret_cast<const floatx2*>(&se[ly][c + 2 * k]);
            const floatx2 to = *reinterpret_cast<const floatx2*>(&so[ly][c + 2 * k]);
            e[2 * k] = te[0]; e[2 * k + 1] = te[1];
            o[2 * k] = to[0]; o[2 * k + 1] = to[1];
        }
        floatx2 a = {0.f, 0.f}, d = {0.f, 0.f};
#pragma unroll
        for (int m = 1; m < 10; ++m) {
            const int k = (m & 1) ? 4 - (m - 1) / 2 : 4 - m / 2;
            const floatx2 x = (m & 1) ? floatx2{e[k], e[k + 1]} : floatx2{o[k], o[k + 1]};
            a = __builtin_elementwise_fma(floatx2{DEC_LO[m], DEC_LO[m]}, x, a);
            if (DEC_HI[m] != 0.f) d = __builtin_elementwise_fma(floatx2{DEC_HI[m], DEC_HI[m]}, x, d);
        }
        wa[q] = a;
        wd[q] = d;
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < NWV; ++q) {
        const int i = tid + q * 256;
        if (i < CIN * (CT / 2)) {
            const int ly = i / (CT / 2), c = 2 * (i - ly * (CT / 2));
            *reinterpret_cast<floatx2*>(&sL[ly][c]) = wa[q];
            *reinterpret_cast<floatx2*>(&sH[ly][c]) = wd[q];
        }
    }
    __syncthreads();
    // height pass, two adjacent columns per thread
    const int hh = h / 2, wh = w / 2;
    for (int i = tid; i < CT * (CT / 2); i += 256) {
        const int r = i / (CT / 2), c = 2 * (i - r * (CT / 2));
        const int ky = ky0 + r, kx = kx0 + c;
        if (ky >= hh || kx >= wh) continue;
        floatx2 a0 = {0.f, 0.f}, d0 = {0.f, 0.f}, a1 = {0.f, 0.f}, d1 = {0.f, 0.f};
#pragma unroll
        for (int m = 1; m < 10; ++m) {
            const floatx2 vl = *reinterpret_cast<const floatx2*>(&sL[2 * r + 9 - m][c]);
            const floatx2 vh = *reinterpret_cast<const floatx2*>(&sH[2 * r + 9 - m][c]);
            const floatx2 cl = {DEC_LO[m], DEC_LO[m]}, ch = {DEC_HI[m], DEC_HI[m]};
            a0 = __builtin_elementwise_fma(cl, vl, a0);
            a1 = __builtin_elementwise_fma(cl, vh, a1);
            if (DEC_HI[m] != 0.f) {
                d0 = __builtin_elementwise_fma(ch, vl, d0);
                d1 = __builtin_elementwise_fma(ch, vh, d1);
            }
        }
        const int64_t o = (int64_t)ky * ll.sy + kx;                          // the four outputs share row/col strides
        st2<CO>(ll.p + z * ll.sz + o, a0[0], a0[1]);                         // low width, low height
        *reinterpret_cast<f2u*>(lh.p + z * lh.sz + o) = f2u{d0[0], d0[1]};   // low width, high height
        *reinterpret_cast<f2u*>(hl.p + z * hl.sz + o) = f2u{a1[0], a1[1]};
        *reinterpret_cast<f2u*>(vhh.p + z * vhh.sz + o) = f2u{d1[0], d1[1]};
    }
}

__global__ __launch_bounds__(256) void k_cdf97_fwd_level_v(V3 in, V3 ll, V3 lh, V3 hl, V3 vhh, int h, int w) {
    __shared__ __attribute__((aligned(16))) float smem[CDF_SMEM_FWDV];
    fwd_level_v_body(smem, blockIdx.z, blockIdx.y, blockIdx.x, in, ll, lh, hl, vhh, h, w);
}

// inverse: a workgroup reconstructs a (2CT)^2 output tile from (CT+4)^2 patches of the 4 subbands
constexpr int CS = CT + 4;
constexpr int CDF_SMEM_INV = 4 * CS * (CS + 1) + 2 * 2 * CT * (CS + 1);

template <bool CI = false, bool CO = false>
__device__ __forceinline__ void inv_level_body(float* smem, int64_t z, int by, int bx, V3 ll, V3 lh, V3 hl, V3 vhh, V3 out,
                                               int h, int w) {
    float (*s4)[CS][CS + 1] = reinterpret_cast<float (*)[CS][CS + 1]>(smem);
    float (*sLw)[CS + 1] = reinterpret_cast<float (*)[CS + 1]>(smem + 4 * CS * (CS + 1));
    float (*sHw)[CS + 1] = reinterpret_cast<float (*)[CS + 1]>(smem + 4 * CS * (CS + 1) + 2 * CT * (CS + 1));
    const int y0 = by * 2 * CT, x0 = bx * 2 * CT;
    const int hh = h / 2, wh = w / 2;
    const int tid = threadIdx.x;
    V3 sb[4] = {ll, lh, hl, vhh};
    const bool small = hh < CS || wh < CS;
    constexpr int NLD = (CS * CS + 255) / 256;
    float v[4][NLD];
#pragma unroll
    for (int r = 0; r < NLD; ++r) {
        const int i = tid + r * 256;
        const int lk = i / CS, lc = i - lk * CS;
        int gy = y0 / 2 - 2 + lk, gx = x0 / 2 - 2 + lc;
        if (small) {
            gy %= hh; gx %= wh;
            if (gy < 0) gy += hh;
            if (gx < 0) gx += wh;
        } else {
            gy += gy < 0 ? hh : (gy >= hh ? -hh : 0);
            gx += gx < 0 ? wh : (gx >= wh ? -wh : 0);
        }
#pragma unroll
        for (int b = 0; b < 4; ++b)
            v[b][r] = i >= CS * CS ? 0.f
                      : b == 0     ? ld1<CI>(sb[0].p + z * sb[0].sz + gy * (int)sb[0].sy + gx * (int)sb[0].sx)
                                   : sb[b].p[z * sb[b].sz + gy * (int)sb[b].sy + gx * (int)sb[b].sx];
    }
#pragma unroll
    for (int r = 0; r < NLD; ++r) {
        const int i = tid + r * 256;
        if (i < CS * CS) {
#pragma unroll
            for (int b = 0; b < 4; ++b) s4[b][i / CS][i % CS] = v[b][r];
        }
    }
    __syncthreads();
    // height synthesis: lw[dn][lc] from (LL, LH), hw[dn][lc] from (HL, HH).  Only the 5 taps t = 2u + (dn & 1) contribute,
    // at rows (dn + 8 - t)/2 = (dn >> 1) + 4 - u: branch-free, the parity only selects the filter taps.
    for (int i = tid; i < 2 * CT * CS; i += 256) {
        const int dn = i / CS, lc = i - dn * CS;
        const int par = dn & 1, base = (dn >> 1) + 4;
        float a = 0.f, b = 0.f;
#pragma unroll
        for (int u = 0; u < 5; ++u) {
            const float rl = par ? c_rec_lo[2 * u + 1] : c_rec_lo[2 * u];
            const float rh = par ? c_rec_hi[2 * u + 1] : c_rec_hi[2 * u];
            const int lk = base - u;
            a = fmaf(s4[0][lk][lc], rl, a);
            a = fmaf(s4[1][lk][lc], rh, a);
            b = fmaf(s4[2][lk][lc], rl, b);
            b = fmaf(s4[3][lk][lc], rh, b);
        }
        sLw[dn][lc] = a;
        sHw[dn][lc] = b;
    }
    __syncthreads();
    float* oz = out.p + z * out.sz;
    for (int i = tid; i < 4 * CT * CT; i += 256) {
        const int dn = i / (2 * CT), dm = i - dn * (2 * CT);
        const int gy = y0 + dn, gx = x0 + dm;
        if (gy >= h || gx >= w) continue;
        const int par = dm & 1, base = (dm >> 1) + 4;
        float acc = 0.f;
#pragma unroll
        for (int u = 0; u < 5; ++u) {
            const float rl = par ? c_rec_lo[2 * u + 1] : c_rec_lo[2 * u];
            const float rh = par ? c_rec_hi[2 * u + 1] : c_rec_hi[2 * u];
            acc = fmaf(sLw[dn][base - u], rl, acc);
            acc = fmaf(sHw[dn][base - u], rh, acc);
        }
        st1<CO>(oz + gy * (int)out.sy + gx * (int)out.sx, acc);
    }
}

__global__ __launch_bounds__(256) void k_cdf97_inv_level(V3 ll, V3 lh, V3 hl, V3 vhh, V3 out, int h, int w) {
    __shared__ float smem[CDF_SMEM_INV];
    inv_level_body(smem, blockIdx.z, blockIdx.y, blockIdx.x, ll, lh, hl, vhh, out, h, w);
}

// Fast variant of the inverse level (contiguous rows, subband edges >= CS and even): dwordx2 loads of the four subband
// patches, 8-byte LDS words, each thread reconstructs an (even, odd) row pair of two columns in the height pass and four
// consecutive pixels in the width pass (both parities share their five input rows / columns); zero taps are skipped.
constexpr float REC_LO[10] = {0.0f, -0.064538882628697f, -0.040689417609164f, 0.418092273221617f, 0.788485616405583f,
                              0.418092273221617f, -0.040689417609164f, -0.064538882628697f, 0.0f, 0.0f};
constexpr float REC_HI[10] = {0.0f, -0.037828455507264f, -0.023849465019557f, 0.110624404418437f, 0.377402855612831f,
                              -0.852698679008894f, 0.377402855612831f, 0.110624404418437f, -0.023849465019557f,
                              -0.037828455507264f};
constexpr int CSP = CS + 2;                // 38: even pitch

constexpr int CDF_SMEM_INVV = 4 * CS * CSP + 2 * 2 * CT * CSP;

template <bool CI = false, bool CO = false>
__device__ __forceinline__ void inv_level_v_body(float* smem, int64_t z, int by, int bx, V3 ll, V3 lh, V3 hl, V3 vhh, V3 out,
                                                 int h, int w) {
    float (*s4)[CS][CSP] = reinterpret_cast<float (*)[CS][CSP]>(smem);
    float (*sLw)[CSP] = reinterpret_cast<float (*)[CSP]>(smem + 4 * CS * CSP);
    float (*sHw)[CSP] = reinterpret_cast<float (*)[CSP]>(smem + 4 * CS * CSP + 2 * CT * CSP);
    const int y0 = by * 2 * CT, x0 = bx * 2 * CT;
    const int hh = h / 2, wh = w / 2;
    const int tid = threadIdx.x;
    const float* sp[4] = {ll.p + z * ll.sz, lh.p + z * lh.sz, hl.p + z * hl.sz, vhh.p + z * vhh.sz};
    constexpr int VR = CS / 2;                          // float2 per patch row (18)
    constexpr int NLV = (CS * VR + 255) / 256;          // 3
    f2u v[4][NLV];
#pragma unroll
    for (int r = 0; r < NLV; ++r) {
        const int i = tid + r * 256;
        const int lk = i / VR, vx = i - lk * VR;
        int gy = y0 / 2 - 2 + lk, gx = x0 / 2 - 2 + 2 * vx;
        gy += gy < 0 ? hh : (gy >= hh ? -hh : 0);
        gx += gx < 0 ? wh : (gx >= wh ? -wh : 0);
        const int64_t off = i < CS * VR ? (int64_t)gy * ll.sy + gx : 0;      // the four subbands share the row stride
#pragma unroll
        for (int b = 0; b < 4; ++b) v[b][r] = b == 0 ? ld2<CI>(sp[0] + off) : *reinterpret_cast<const f2u*>(sp[b] + off);
    }
#pragma unroll
    for (int r = 0; r < NLV; ++r) {
        const int i = tid + r * 256;
        if (i < CS * VR) {
            const int lk = i / VR, vx = i - lk * VR;
#pragma unroll
            for (int b = 0; b < 4; ++b) *reinterpret_cast<floatx2*>(&s4[b][lk][2 * vx]) = floatx2{v[b][r].x, v[b][r].y};
        }
    }
    __syncthreads();
    // height synthesis: rows dn = 2j + par, base row j + 4; taps t = 2u + par at input row base - u
    for (int i = tid; i < CT * VR; i += 256) {
        const int j = i / VR, lc = 2 * (i - j * VR);
        floatx2 a0 = {0.f, 0.f}, a1 = {0.f, 0.f}, b0 = {0.f, 0.f}, b1 = {0.f, 0.f};
#pragma unroll
        for (int u = 0; u < 5; ++u) {
            const int lk = j + 4 - u;
            const floatx2 q0 = *reinterpret_cast<const floatx2*>(&s4[0][lk][lc]);
            const floatx2 q1 = *reinterpret_cast<const floatx2*>(&s4[1][lk][lc]);
            const floatx2 q2 = *reinterpret_cast<const floatx2*>(&s4[2][lk][lc]);
            const floatx2 q3 = *reinterpret_cast<const floatx2*>(&s4[3][lk][lc]);
            if (REC_LO[2 * u] != 0.f) {
                const floatx2 c = {REC_LO[2 * u], REC_LO[2 * u]};
                a0 = __builtin_elementwise_fma(q0, c, a0);
                b0 = __builtin_elementwise_fma(q2, c, b0);
            }
            if (REC_HI[2 * u] != 0.f) {
                const floatx2 c = {REC_HI[2 * u], REC_HI[2 * u]};
                a0 = __builtin_elementwise_fma(q1, c, a0);
                b0 = __builtin_elementwise_fma(q3, c, b0);
            }
            if (REC_LO[2 * u + 1] != 0.f) {
                const floatx2 c = {REC_LO[2 * u + 1], REC_LO[2 * u + 1]};
                a1 = __builtin_elementwise_fma(q0, c, a1);
                b1 = __builtin_elementwise_fma(q2, c, b1);
            }
            if (REC_HI[2 * u + 1] != 0.f) {
                const floatx2 c = {REC_HI[2 * u + 1], REC_HI[2 * u + 1]};
                a1 = __builtin_elementwise_fma(q1, c, a1);
                b1 = __builtin_elementwise_fma(q3, c, b1);
            }
        }
        *reinterpret_cast<floatx2*>(&sLw[2 * j][lc]) = a0;
        *reinterpret_cast<floatx2*>(&sLw[2 * j + 1][lc]) = a1;
        *reinterpret_cast<floatx2*>(&sHw[2 * j][lc]) = b0;
        *reinterpret_cast<floatx2*>(&sHw[2 * j + 1][lc]) = b1;
    }
    __syncthreads();
    // width synthesis: four consecutive pixels dm = 4q .. 4q+3 from columns 2q .. 2q+5 of both half-rows
    float* oz = out.p + z * out.sz;
    for (int i = tid; i < 2 * CT * (CT / 2); i += 256) {
        const int dn = i / (CT / 2), q = i - dn * (CT / 2);
        const int gy = y0 + dn, gx = x0 + 4 * q;
        if (gy >= h || gx >= w) continue;
        float L[6], H[6];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const floatx2 tl = *reinterpret_cast<const floatx2*>(&sLw[dn][2 * q + 2 * k]);
            const floatx2 th = *reinterpret_cast<const floatx2*>(&sHw[dn][2 * q + 2 * k]);
            L[2 * k] = tl[0]; L[2 * k + 1] = tl[1];
            H[2 * k] = th[0]; H[2 * k + 1] = th[1];
        }
        floatx2 ev = {0.f, 0.f}, od = {0.f, 0.f};        // (pixel 0, pixel 2) and (pixel 1, pixel 3)
#pragma unroll
        for (int u = 0; u < 5; ++u) {
            const floatx2 xl = {L[4 - u], L[5 - u]}, xh = {H[4 - u], H[5 - u]};
            if (REC_LO[2 * u] != 0.f) ev = __builtin_elementwise_fma(xl, floatx2{REC_LO[2 * u], REC_LO[2 * u]}, ev);
            if (REC_HI[2 * u] != 0.f) ev = __builtin_elementwise_fma(xh, floatx2{REC_HI[2 * u], REC_HI[2 * u]}, ev);
            if (REC_LO[2 * u + 1] != 0.f) od = __builtin_elementwise_fma(xl, floatx2{REC_LO[2 * u + 1], REC_LO[2 * u + 1]}, od);
            if (REC_HI[2 * u + 1] != 0.f) od = __builtin_elementwise_fma(xh, floatx2{REC_HI[2 * u + 1], REC_HI[2 * u + 1]}, od);
        }
        st4<CO>(oz + (int64_t)gy * out.sy + gx, ev[0], od[0], ev[1], od[1]);
    }
}

__global__ __launch_bounds__(256) void k_cdf97_inv_level_v(V3 ll, V3 lh, V3 hl, V3 vhh, V3 out, int h, int w) {
    __shared__ __attribute__((aligned(16))) float smem[CDF_SMEM_INVV];
    inv_level_v_body(smem, blockIdx.z, blockIdx.y, blockIdx.x, ll, lh, hl, vhh, out, h, w);
}

// ---- all levels of the forward transform in ONE launch ------------------------------------------------------------------
// At the BASELINE batch the per-level launches are latency-bound (four dependent launches for 50 MB).  Here every tile of
// every level is one workgroup of a single 1-D grid, coarse levels after fine ones, and a tile of level l+1 starts as soon as
// the (up to 4 x 4) level-l tiles that wrote its input patch have published their flag: a 64-bit word per producer tile set
// to this call's tag (unique per call, so the flags never need clearing and stale workspace contents cannot match).
// Forward progress: workgroups are dispatched in increasing block index and a tile only waits for tiles with smaller
// indexes, so the lowest unfinished tile can always run (the assumption every decoupled look-back scan makes); the poll loop
// is bounded all the same, and a workgroup that gives up raises *timeout (pinned host word, checked by the next call).
constexpr int CDF_MAXLEV = 16;
constexpr int cmax(int a, int b) { return a > b ? a : b; }
constexpr int CDF_SMEM_F = cmax(CDF_SMEM_FWDV, CDF_SMEM_FWD), CDF_SMEM_I = cmax(CDF_SMEM_INVV, CDF_SMEM_INV);
struct CdfFused {
    const float* x;                        // (Z, H, W)
    float* ll;                             // LL of the last level
    float* yh[CDF_MAXLEV];                 // detail subbands per level, (Z, 3, h/2, w/2)
    float* llb[CDF_MAXLEV];                // LL of level lev < levels - 1 (workspace; one buffer per level: levels overlap in time)
    unsigned long long* flags;
    unsigned long long tag;
    int* timeout;
    int off[CDF_MAXLEV + 1];               // first block of each level
    int foff[CDF_MAXLEV];                  // first flag of each level
    int H, W, levels;
    int mode;                              // 0: L2 write-back / invalidate fences around the flags; 1: coherent (sc1) LL accesses;
                                           // 2: no ordering at all (TIMING EXPERIMENTS ONLY: wrong results)
};

__device__ __forceinline__ void publish_tile(unsigned long long* flag, unsigned long long tag, int mode) {
    if (mode == 2) return;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's stores are acknowledged ...
    __syncthreads();                                      // ... and so are the other waves'
    if (threadIdx.x == 0) {
        if (mode == 0) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");        // L2 write-back: visible to the other XCDs
        __hip_atomic_store(flag, tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// wave 0 polls up to 16 flags (ny x nx producer tiles starting at (fy, fx), periodic in the pty x ptx producer grid)
__device__ __forceinline__ void await_tiles(const unsigned long long* flags, unsigned long long tag, int fy, int ny, int fx,
                                            int nx, int pty, int ptx, int* timeout, int mode) {
    if (mode == 2) return;
    if (threadIdx.x < 64) {
        const int lane = threadIdx.x, iy = lane >> 2, ix = lane & 3;
        const bool need = lane < 16 && iy < ny && ix < nx;
        const int py = ((fy + iy) % pty + pty) % pty, px = ((fx + ix) % ptx + ptx) % ptx;
        const unsigned long long* f = flags + (need ? py * ptx + px : 0);
        for (int n = 0;; ++n) {
            const unsigned long long v = need ? __hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : tag;
            if (__ballot(v != tag) == 0) break;
            if (n >= (1 << 20)) {                                                // seconds: a producer never ran
                if (lane == 0) *timeout = 1;
                break;
            }
            __builtin_amdgcn_s_sleep(4);
        }
    }
    __syncthreads();
    if (mode == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");            // drop stale L1 / L2 lines of the input
}

// Workgroups go to the 8 XCDs round-robin by block index; neighbouring tiles share halo rows and columns, so the tiles an XCD
// works on should be neighbours (its L2 then serves the halo).  r = dispatch order inside a level of n tiles: the blocks with the
// same r % 8 take a contiguous n/8 tiles (whole planes at the BASELINE shape); the n % 8 leftovers keep their index.
__device__ __forceinline__ int xcd_order(int r, int n) {
    const int per = n >> 3;
    return r < per * 8 ? (r & 7) * per + (r >> 3) : r;
}

#define CDF_BODY(BODY, ...)                            \
    if (ci && co) BODY<true, true>(__VA_ARGS__);       \
    else if (ci) BODY<true, false>(__VA_ARGS__);       \
    else if (co) BODY<false, true>(__VA_ARGS__);       \
    else BODY<false, false>(__VA_ARGS__);

__global__ __launch_bounds__(256) void k_cdf97_fwd_all(CdfFused a) {
    __shared__ __attribute__((aligned(16))) float smem[CDF_SMEM_F];
    const int b = blockIdx.x;
    int lev = 0;
    while (lev + 1 < a.levels && b >= a.off[lev + 1]) ++lev;
    const int h = a.H >> lev, w = a.W >> lev, hh = h / 2, wh = w / 2;
    const int ty = (hh + CT - 1) / CT, tx = (wh + CT - 1) / CT;
    const int r = xcd_order(b - a.off[lev], a.off[lev + 1] - a.off[lev]);
    const int z = r / (ty * tx), t = r - z * (ty * tx), by = t / tx, bx = t - by * tx;
    if (lev > 0) {
        // input = LL of level lev-1 (h x w), written by CT x CT tiles; the patch covers rows 2 by CT - 4 .. 2 by CT + CIN - 5
        const int pty = (h + CT - 1) / CT, ptx = (w + CT - 1) / CT;
        const int fy = (2 * by * CT - 4) >> 5, fx = (2 * bx * CT - 4) >> 5;          // floor division by CT = 32
        const int ly = (2 * by * CT + CIN - 5) >> 5, lx = (2 * bx * CT + CIN - 5) >> 5;
        const int ny = ly - fy + 1 < pty ? ly - fy + 1 : pty, nx = lx - fx + 1 < ptx ? lx - fx + 1 : ptx;
        await_tiles(a.flags + a.foff[lev - 1] + (int64_t)z * pty * ptx, a.tag, fy, ny, fx, nx, pty, ptx, a.timeout, a.mode);
    }
    const int64_t sub = (int64_t)hh * wh;
    V3 in{lev == 0 ? const_cast<float*>(a.x) : a.llb[lev - 1], (int64_t)h * w, w, 1};
    float* y = a.yh[lev];
    V3 vLL{lev == a.levels - 1 ? a.ll : a.llb[lev], sub, wh, 1};
    V3 vLH{y, 3 * sub, wh, 1}, vHL{y + sub, 3 * sub, wh, 1}, vHH{y + 2 * sub, 3 * sub, wh, 1};
    const bool ci = a.mode == 1 && lev > 0, co = a.mode == 1 && lev + 1 < a.levels;
    if (h >= CIN && w >= CIN && w % 4 == 0) {
        CDF_BODY(fwd_level_v_body, smem, z, by, bx, in, vLL, vLH, vHL, vHH, h, w)
    } else {
        CDF_BODY(fwd_level_body, smem, z, by, bx, in, vLL, vLH, vHL, vHH, h, w)
    }
    if (lev + 1 < a.levels) publish_tile(a.flags + a.foff[lev] + ((int64_t)z * ty + by) * tx + bx, a.tag, a.mode);
}

// the inverse the same way: coarse levels first; a 2CT x 2CT output tile of level lev waits for the (up to 2 x 2) output tiles of
// level lev+1 that wrote the LL patch it reads.  llb[lev] = output of level lev > 0; x/ll swap roles (x is the output).
__global__ __launch_bounds__(256) void k_cdf97_inv_all(CdfFused a) {
    __shared__ __attribute__((aligned(16))) float smem[CDF_SMEM_I];
    const int b = blockIdx.x;
    int j = 0;                                                    // position in launch order: level levels-1-j
    while (j + 1 < a.levels && b >= a.off[j + 1]) ++j;
    const int lev = a.levels - 1 - j;
    const int h = a.H >> lev, w = a.W >> lev, hh = h / 2, wh = w / 2;
    const int ty = (h + 2 * CT - 1) / (2 * CT), tx = (w + 2 * CT - 1) / (2 * CT);
    const int r = xcd_order(b - a.off[j], a.off[j + 1] - a.off[j]);
    const int z = r / (ty * tx), t = r - z * (ty * tx), by = t / tx, bx = t - by * tx;
    if (j > 0) {
        // LL input (hh x wh) = output of level lev+1, written in 2CT x 2CT tiles; the patch covers rows by CT - 2 .. by CT + CS - 3
        const int pty = (hh + 2 * CT - 1) / (2 * CT), ptx = (wh + 2 * CT - 1) / (2 * CT);
        const int fy = (by * CT - 2) >> 6, fx = (bx * CT - 2) >> 6;
        const int ly = (by * CT + CS - 3) >> 6, lx = (bx * CT + CS - 3) >> 6;
        const int ny = ly - fy + 1 < pty ? ly - fy + 1 : pty, nx = lx - fx + 1 < ptx ? lx - fx + 1 : ptx;
        await_tiles(a.flags + a.foff[j - 1] + (int64_t)z * pty * ptx, a.tag, fy, ny, fx, nx, pty, ptx, a.timeout, a.mode);
    }
    const int64_t sub = (int64_t)hh * wh;
    float* y = a.yh[lev];
    V3 vLL{lev == a.levels - 1 ? a.ll : a.llb[lev + 1], sub, wh, 1};
    V3 vLH{y, 3 * sub, wh, 1}, vHL{y + sub, 3 * sub, wh, 1}, vHH{y + 2 * sub, 3 * sub, wh, 1};
    V3 vo{lev == 0 ? const_cast<float*>(a.x) : a.llb[lev], (int64_t)h * w, w, 1};
    const bool ci = a.mode == 1 && j > 0, co = a.mode == 1 && lev > 0;
    if (hh >= CS && wh >= CS && wh % 2 == 0) {
        CDF_BODY(inv_level_v_body, smem, z, by, bx, vLL, vLH, vHL, vHH, vo, h, w)
    } else {
        CDF_BODY(inv_level_body, smem, z, by, bx, vLL, vLH, vHL, vHH, vo, h, w)
    }
    if (lev > 0) publish_tile(a.flags + a.foff[j] + ((int64_t)z * ty + by) * tx + bx, a.tag, a.mode);
}
static_assert(CT == 32, "the tile arithmetic above shifts by 5 and 6");

static inline dim3 grid2d(int64_t h, int64_t w, int64_t Z) {
    return dim3((unsigned)cdiv(w, 256), (unsigned)(h < 2048 ? h : 2048), (unsigned)Z);
}

}  // namespace lldwt
using namespace lldwt;

// workspace: lo_w, hi_w (Z*H*W/2 each) + two LL ping-pong buffers (Z*H*W/4 each)
// + the tile flags of the one-launch transform (8 B per tile of every level but the last; bounded by 2 x level 0 + 16 / plane)
static int64_t cdf_flag_words(int64_t Z, int64_t H, int64_t W) {
    return 2 * Z * cdiv(H / 2, CT) * cdiv(W / 2, CT) + 16 * Z;
}
extern "C" int64_t lldwt_cdf97_ws_bytes(int64_t Z, int64_t H, int64_t W) {
    return (int64_t)sizeof(float) * (Z * H * (W / 2) * 2 + Z * (H / 2) * (W / 2) * 2) + 8 * cdf_flag_words(Z, H, W);
}

// pinned host word the one-launch kernels raise when a tile gave up waiting for its producers; sticky
static int* cdf_timeout_word() {
    static std::mutex mu;
    static int* word = nullptr;
    std::lock_guard<std::mutex> g(mu);
    if (!word && hipHostMalloc(reinterpret_cast<void**>(&word), sizeof(int), hipHostMallocMapped) == hipSuccess) *word = 0;
    return word;
}
static std::atomic<unsigned long long> g_cdf_call{0};
static int cdf_mode() {
    const char* e = getenv("LLDWT_CDF97_FUSE");          // 0: per-level launches; 1 (default): one launch
    return e ? atoi(e) : 1;
}
static bool cdf_fusable(int levels, hipStream_t st) {
    if (levels < 2 || levels > CDF_MAXLEV) return false;
    if (cdf_mode() == 0) return false;
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;       // a captured launch would replay its tag: per-level launches
    if (hipStreamIsCapturing(st, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) return false;
    return true;
}

static int cdf_args(const char* who, int64_t Z, int64_t H, int64_t W, int levels, void* ws, int64_t ws_bytes) {
    LLDWT_REQUIRE(Z > 0 && Z <= 65535 && levels > 0 && levels < 16, "%s: bad Z/levels", who);
    LLDWT_REQUIRE(H > 0 && W > 0 && H % (1 << levels) == 0 && W % (1 << levels) == 0,
                  "%s: H=%ld W=%ld must be divisible by 2^levels", who, (long)H, (long)W);
    LLDWT_REQUIRE(ws, "%s: null workspace", who);
    if (ws_bytes < lldwt_cdf97_ws_bytes(Z, H, W)) {
        set_error("%s: workspace %ld < %ld bytes", who, (long)ws_bytes, (long)lldwt_cdf97_ws_bytes(Z, H, W));
        return LLDWT_EWS;
    }
    return 0;
}

extern "C" int lldwt_cdf97_forward(const float* x, float* ll, float* const* yh, int64_t Z, int64_t H, int64_t W,
                                   int levels, void* ws, int64_t ws_bytes, void* stream) {
    return lldwt_cdf97_forward_ex(x, ll, yh, Z, H, W, levels, 0, ws, ws_bytes, stream);
}

extern "C" int lldwt_cdf97_forward_ex(const float* x, float* ll, float* const* yh, int64_t Z, int64_t H, int64_t W,
                                      int levels, int adj, void* ws, int64_t ws_bytes, void* stream) {
    int r = cdf_args("cdf97_forward", Z, H, W, levels, ws, ws_bytes);
    if (r) return r;
    LLDWT_REQUIRE(x && ll && yh, "cdf97_forward: null pointer");
    hipStream_t st = (hipStream_t)stream;
    float* low = (float*)ws;
    float* hiw = low + Z * H * (W / 2);
    float* llb[2] = {hiw + Z * H * (W / 2), hiw + Z * H * (W / 2) + Z * (H / 2) * (W / 2)};
    if (!adj && cdf_fusable(levels, st)) {
        int* tw = cdf_timeout_word();
        LLDWT_REQUIRE(tw, "cdf97_forward: no pinned host word");
        LLDWT_REQUIRE(*tw == 0, "cdf97: an earlier one-launch transform timed out waiting for its producer tiles");
        CdfFused a{};
        a.x = x; a.ll = ll; a.H = (int)H; a.W = (int)W; a.levels = levels; a.timeout = tw;
        a.mode = cdf_mode() - 1;
        a.flags = reinterpret_cast<unsigned long long*>(llb[1] + Z * (H / 2) * (W / 2));
        a.tag = 0xC97F0A11ull << 32 | (++g_cdf_call & 0xffffffffull);
        int64_t blocks = 0, nflags = 0;
        float* lp = low;                                   // the LL buffers live in the (unused here) separable-pass area
        for (int lev = 0; lev < levels; ++lev) {
            const int64_t hh = (H >> lev) / 2, wh = (W >> lev) / 2, tiles = Z * cdiv(hh, CT) * cdiv(wh, CT);
            a.yh[lev] = yh[lev];
            LLDWT_REQUIRE(yh[lev], "cdf97_forward: null subband pointer");
            a.off[lev] = (int)blocks;
            blocks += tiles;
            if (lev + 1 < levels) {
                a.llb[lev] = lp;
                lp += Z * hh * wh;
                a.foff[lev] = (int)nflags;
                nflags += tiles;
            }
        }
        a.off[levels] = (int)blocks;
        LLDWT_REQUIRE(blocks < (1ll << 31) && nflags <= cdf_flag_words(Z, H, W) && lp <= hiw + Z * H * (W / 2),
                      "cdf97_forward: grid or workspace layout out of range");
        hipLaunchKernelGGL(k_cdf97_fwd_all, dim3((unsigned)blocks), dim3(256), 0, st, a);
        return check_launch("cdf97_forward");
    }
    const float* cur = x;
    for (int lev = 0; lev < levels; ++lev) {
        const int64_t h = H >> lev, w = W >> lev, hh = h / 2, wh = w / 2, sub = hh * wh;
        V3 in{const_cast<float*>(cur), h * w, w, 1};
        V3 lw{low, h * wh, wh, 1}, hw_{hiw, h * wh, wh, 1};
        float* llout = lev == levels - 1 ? ll : llb[lev & 1];
        float* y = yh[lev];
        V3 vLL{llout, sub, wh, 1}, vLH{y, 3 * sub, wh, 1}, vHL{y + sub, 3 * sub, wh, 1}, vHH{y + 2 * sub, 3 * sub, wh, 1};
        if (!adj) {       // fused level: one read of the input, one write of the four subbands
            dim3 grid((unsigned)cdiv(wh, CT), (unsigned)cdiv(hh, CT), (unsigned)Z);
            if (h >= CIN && w >= CIN && w % 4 == 0 && in.sx == 1)
                hipLaunchKernelGGL(k_cdf97_fwd_level_v, grid, dim3(256), 0, st, in, vLL, vLH, vHL, vHH, (int)h, (int)w);
            else
                hipLaunchKernelGGL(k_cdf97_fwd_level, grid, dim3(256), 0, st, in, vLL, vLH, vHL, vHH, (int)h, (int)w);
        } else {
            hipLaunchKernelGGL(k_afb, grid2d(h, wh, Z), dim3(256), 0, st, in, lw, hw_, (int)h, (int)w, 1, adj);
            hipLaunchKernelGGL(k_afb, grid2d(hh, wh, Z), dim3(256), 0, st, lw, vLL, vLH, (int)h, (int)wh, 0, adj);
            hipLaunchKernelGGL(k_afb, grid2d(hh, wh, Z), dim3(256), 0, st, hw_, vHL, vHH, (int)h, (int)wh, 0, adj);
        }
        cur = llout;
    }
    return check_launch("cdf97_forward");
}

extern "C" int lldwt_cdf97_inverse(const float* ll, const float* const* yh, float* x, int64_t Z, int64_t H, int64_t W,
                                   int levels, void* ws, int64_t ws_bytes, void* stream) {
    return lldwt_cdf97_inverse_ex(ll, yh, x, Z, H, W, levels, 0, ws, ws_bytes, stream);
}

extern "C" int lldwt_cdf97_inverse_ex(const float* ll, const float* const* yh, float* x, int64_t Z, int64_t H, int64_t W,
                                      int levels, int adj, void* ws, int64_t ws_bytes, void* stream) {
    int r = cdf_args("cdf97_inverse", Z, H, W, levels, ws, ws_bytes);
    if (r) return r;
    LLDWT_REQUIRE(x && ll && yh, "cdf97_inverse: null pointer");
    hipStream_t st = (hipStream_t)stream;
    float* low = (float*)ws;
    float* hiw = low + Z * H * (W / 2);
    float* llb[2] = {hiw + Z * H * (W / 2), hiw + Z * H * (W / 2) + Z * (H / 2) * (W / 2)};
    if (!adj && cdf_fusable(levels, st)) {
        int* tw = cdf_timeout_word();
        LLDWT_REQUIRE(tw, "cdf97_inverse: no pinned host word");
        LLDWT_REQUIRE(*tw == 0, "cdf97: an earlier one-launch transform timed out waiting for its producer tiles");
        CdfFused a{};
        a.x = x; a.ll = const_cast<float*>(ll); a.H = (int)H; a.W = (int)W; a.levels = levels; a.timeout = tw;
        a.mode = cdf_mode() - 1;
        a.flags = reinterpret_cast<unsigned long long*>(llb[1] + Z * (H / 2) * (W / 2));
        a.tag = 0xC97F0A11ull << 32 | (++g_cdf_call & 0xffffffffull);
        int64_t blocks = 0, nflags = 0;
        float* lp = low;
        for (int j = 0; j < levels; ++j) {
            const int lev = levels - 1 - j;
            const int64_t h = H >> lev, w = W >> lev, tiles = Z * cdiv(h, 2 * CT) * cdiv(w, 2 * CT);
            a.yh[lev] = const_cast<float*>(yh[lev]);
            LLDWT_REQUIRE(yh[lev], "cdf97_inverse: null subband pointer");
            a.off[j] = (int)blocks;
            blocks += tiles;
            if (lev > 0) {
                a.llb[lev] = lp;
                lp += Z * h * w;
                a.foff[j] = (int)nflags;
                nflags += tiles;
            }
        }
        a.off[levels] = (int)blocks;
        LLDWT_REQUIRE(blocks < (1ll << 31) && nflags <= cdf_flag_words(Z, H, W) && lp <= hiw + Z * H * (W / 2),
                      "cdf97_inverse: grid or workspace layout out of range");
        hipLaunchKernelGGL(k_cdf97_inv_all, dim3((unsigned)blocks), dim3(256), 0, st, a);
        return check_launch("cdf97_inverse");
    }
    const float* cur = ll;
    for (int lev = levels - 1; lev >= 0; --lev) {
        const int64_t h = H >> lev, w = W >> lev, hh = h / 2, wh = w / 2, sub = hh * wh;
        float* y = const_cast<float*>(yh[lev]);
        V3 vLL{const_cast<float*>(cur), sub, wh, 1}, vLH{y, 3 * sub, wh, 1}, vHL{y + sub, 3 * sub, wh, 1},
            vHH{y + 2 * sub, 3 * sub, wh, 1};
        V3 lw{low, h * wh, wh, 1}, hw_{hiw, h * wh, wh, 1};
        float* out = lev == 0 ? x : llb[lev & 1];
        V3 vo{out, h * w, w, 1};
        if (!adj) {
            dim3 grid((unsigned)cdiv(w, 2 * CT), (unsigned)cdiv(h, 2 * CT), (unsigned)Z);
            if (hh >= CS && wh >= CS && wh % 2 == 0)
                hipLaunchKernelGGL(k_cdf97_inv_level_v, grid, dim3(256), 0, st, vLL, vLH, vHL, vHH, vo, (int)h, (int)w);
            else
                hipLaunchKernelGGL(k_cdf97_inv_level, grid, dim3(256), 0, st, vLL, vLH, vHL, vHH, vo, (int)h, (int)w);
        } else {
            hipLaunchKernelGGL(k_sfb, grid2d(h, wh, Z), dim3(256), 0, st, vLL, vLH, lw, (int)h, (int)wh, 0, adj);
            hipLaunchKernelGGL(k_sfb, grid2d(h, wh, Z), dim3(256), 0, st, vHL, vHH, hw_, (int)h, (int)wh, 0, adj);
            hipLaunchKernelGGL(k_sfb, grid2d(h, w, Z), dim3(256), 0, st, lw, hw_, vo, (int)h, (int)w, 1, adj);
        }
        cur = out;
    }
    return check_launch("cdf97_inverse");
}
