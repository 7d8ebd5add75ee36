// conv_bwd.hip -- backward-weights of the conv layers on the fp32 matrix cores, plus the small backward helpers.
//
// Backward-data is the forward engine itself (conv_mfma.hip) applied to the gradient with the taps flipped and the
// channel roles swapped (lldwt_conv_desc.transposed); the reference gets all of this from autograd
// (agents/liftingDWT_agent.py:97 `.backward()`).  Backward-weights is a different contraction:
//     dW[oc][ic][tap] = sum over images and pixels of  dY[z][oc][p] * X[z][ic][p + tap]
// i.e. a GEMM with M = cout/groups, N = (cin/groups * live taps), K = batch * h * w.  One workgroup owns an
// (MT*16) x (NT*16) tile of dW and walks over the pixels of its images in 8x16 spatial chunks:
//   A = dY chunk   [oc][128 px]  in LDS, row stride == 2 (mod 32) dwords -> the 16 oc lanes hit distinct even banks,
//                                the +1 pixel lanes the odd ones: conflict-free ds_read_b32
//   B = X patch    [ic][10x18]   in LDS (KS=3), gathered per lane: lane n = (ic, tap) reads X[ic][p + tap]
// K is split over images (blockIdx.y) and the partial tiles are accumulated with float atomics (dW must be zeroed).
#include "common.h"

namespace lldwt {

typedef float floatx4 __attribute__((ext_vector_type(4)));

constexpr int WG_TH = 8, WG_TW = 16, WG_PX = WG_TH * WG_TW;   // 128-pixel K chunk
constexpr int WG_PSA = WG_PX + 2;                              // dY row stride: == 2 mod 32

struct WgradArgs {
    const float* x;
    const float* dy;
    float* dw;
    float* db;         // bias gradient folded in as one extra (ic,tap) column whose B operand is the constant 1, or null
    lldwt_conv_desc d;
    int batch, h, w, ntaps, zsplit;
    int n_total;       // cin_g * ntaps (+1 when db != null: the bias column)
    int n_w;           // cin_g * ntaps
    int nic_max;       // most input channels any n-block touches (sizes the X patch in LDS)
    float alpha;
    int8_t tdy[25], tdx[25];
    int8_t tap_of[25];
};

// 4-byte aligned float4: global_load_dwordx4 without the 16-byte alignment promise (rows start at any pixel)
struct __attribute__((packed, aligned(4))) f4u { float x, y, z, w; };

template <int KS, int MT, int NT>
__global__ __launch_bounds__(256) void k_conv_wgrad(WgradArgs a) {
    // wave layout: 4 waves; each wave owns (MT x NT)/4 tiles: split along N
    constexpr int WNT = NT / 4;                  // n tiles per wave
    constexpr int R = KS / 2;
    constexpr int IH = WG_TH + 2 * R, IW = WG_TW + 2 * R;
    constexpr int PSX = IH * IW + 1;
    constexpr int NV = MT * 16 * WG_PX / 4 / 256;   // dY float4s per thread and chunk
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const lldwt_conv_desc& d = a.d;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int col = lane & 15, kk = lane >> 4;
    const int cin_g = d.cin / d.groups, cout_g = d.cout / d.groups;
    const int n_tiles = (a.n_total + NT * 16 - 1) / (NT * 16);
    const int m_tiles = (cout_g + MT * 16 - 1) / (MT * 16);
    int bid = blockIdx.x;
    const int nb = bid % n_tiles; bid /= n_tiles;
    const int mb = bid % m_tiles; bid /= m_tiles;
    const int g = bid;
    const int plane = blockIdx.z;
    const int n0 = nb * NT * 16, oc0 = mb * MT * 16;
    // input channels touched by this n-block: ic_first .. ic_last
    const int ic_first = n0 / a.ntaps;
    const int n_end = min(n0 + NT * 16, a.n_w);
    const int ic_last = n_end > n0 ? (n_end - 1) / a.ntaps : ic_first;
    const int nic = ic_last - ic_first + 1;
    float* la = lds;                               // [MT*16][WG_PSA]
    float* lx = lds + MT * 16 * WG_PSA + 2;        // [nic][PSX]; lx[-1] holds the constant 1 of the bias column
    int* s_xoff = (int*)(lx + a.nic_max * PSX);    // [nic_max] memory channel of X channel c, or -1
    int* s_aoff = s_xoff + a.nic_max;              // [MT*16]   memory channel of dY row c, or -1
    const int h = a.h, w = a.w;
    const int hi = d.upsample2 ? h >> 1 : h, wi = d.upsample2 ? w >> 1 : w;
    const int64_t hw = (int64_t)h * w, hwi = (int64_t)hi * wi;
    const int xtot = d.ic_block > 0 ? d.xtot : d.cin;
    const int icb = d.ic_block > 0 ? d.ic_block : d.cin, ics = d.ic_block > 0 ? d.ic_stride : 0, ico = d.ic_block > 0 ? d.ic_off : 0;
    const int ups = d.upsample2;

    // channel placement tables (the runtime divisions happen once per workgroup, not once per loaded element)
    if (tid < MT * 16) {
        const int ocl = oc0 + tid;
        int v = -1;
        if (ocl < cout_g) {
            const int oc = g * cout_g + ocl;
            v = (oc / d.oc_block) * d.oc_stride + d.oc_off + oc % d.oc_block;
        }
        s_aoff[tid] = v;
    }
    for (int c = tid; c < a.nic_max; c += 256) {
        int v = -1;
        if (c < nic && ic_first + c < cin_g) {
            const int icg = g * cin_g + ic_first + c;
            v = (icg / icb) * ics + ico + icg % icb;
        }
        s_xoff[c] = v;
    }
    // per-lane B bases for this wave's n tiles
    int bbase[WNT];
    bool isb[WNT];
#pragma unroll
    for (int j = 0; j < WNT; ++j) {
        const int n = n0 + (wave * WNT + j) * 16 + col;
        const bool nv = n < a.n_w;
        isb[j] = a.db != nullptr && n == a.n_w;
        const int icl = nv ? n / a.ntaps : ic_first;
        const int tl = nv ? n - icl * a.ntaps : 0;
        bbase[j] = isb[j] ? -1 : (icl - ic_first) * PSX + a.tdy[tl] * IW + a.tdx[tl];
    }
    if (tid == 0) lx[-1] = 1.0f;
    floatx4 acc[MT][WNT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int j = 0; j < WNT; ++j) acc[m][j] = floatx4{0.f, 0.f, 0.f, 0.f};
    __syncthreads();

    // K (= images x spatial chunks) is split over blockIdx.y: slice s takes chunks s, s + zsplit, ...
    const int tiles_x = (w + WG_TW - 1) / WG_TW, tiles_y = (h + WG_TH - 1) / WG_TH;
    const int ntile = tiles_x * tiles_y;
    const int total = a.batch * ntile;
    const int nxe = nic * IH * IW;                 // X elements per chunk
    const bool xvec = KS == 1 && !ups;             // 1x1: the X patch is 8 rows of 16 contiguous pixels, like dY

    // one row segment of 4 pixels: a single dwordx4 load when it lies inside the image (safe address + select, no
    // branch per element); the ragged right edge (w % 4 != 0) takes the per-pixel path
    // (two phases: ISSUE4 only issues the load -- from a safe address when the segment is not fully inside -- and FIX4,
    //  run after all loads of the batch are in flight, zeroes / patches it: a branch between two loads would make the
    //  compiler wait for the first one)
#define LLDWT_WG_ISSUE4(raw_, base_, choff_, cstride_, wrow_, iv_)                                               \
    {                                                                                                            \
        const int c_ = (iv_) / (WG_PX / 4), rq_ = (iv_) % (WG_PX / 4);                                           \
        const int gy_ = y0 + rq_ / (WG_TW / 4), gx_ = x0 + 4 * (rq_ % (WG_TW / 4));                              \
        const int co_ = choff_[c_];                                                                              \
        const bool ok4_ = co_ >= 0 && gy_ < h && gx_ + 3 < w;                                                    \
        raw_ = *reinterpret_cast<const f4u*>(base_ + (ok4_ ? co_ * (cstride_) + (int64_t)gy_ * (wrow_) + gx_ : 0)); \
    }
#define LLDWT_WG_FIX4(dst_, raw_, base_, choff_, cstride_, wrow_, iv_)                                           \
    {                                                                                                            \
        const int c_ = (iv_) / (WG_PX / 4), rq_ = (iv_) % (WG_PX / 4);                                           \
        const int gy_ = y0 + rq_ / (WG_TW / 4), gx_ = x0 + 4 * (rq_ % (WG_TW / 4));                              \
        const int co_ = choff_[c_];                                                                              \
        const bool row_ = co_ >= 0 && gy_ < h;                                                                   \
        const bool ok4_ = row_ && gx_ + 3 < w;                                                                   \
        dst_ = ok4_ ? floatx4{raw_.x, raw_.y, raw_.z, raw_.w} : floatx4{0.f, 0.f, 0.f, 0.f};                     \
        if (row_ && !ok4_ && gx_ < w) {                                                                          \
            const int64_t off_ = co_ * (cstride_) + (int64_t)gy_ * (wrow_) + gx_;                                \
            _Pragma("unroll") for (int e_ = 0; e_ < 3; ++e_)                                                     \
                if (gx_ + e_ < w) dst_[e_] = base_[off_ + e_];                                                   \
        }                                                                                                        \
    }
#define LLDWT_WG_STORE4(arr_, stride_, src_, iv_)                                                                \
    {                                                                                                            \
        const int c_ = (iv_) / (WG_PX / 4), rq_ = (iv_) % (WG_PX / 4);                                           \
        float2* q_ = reinterpret_cast<float2*>(arr_ + c_ * (stride_) + 4 * rq_);                                 \
        q_[0] = float2{src_[0], src_[1]};                                                                        \
        q_[1] = float2{src_[2], src_[3]};                                                                        \
    }
#define LLDWT_WG_LOAD_X(dst_, i_)                                                                                \
    {                                                                                                            \
        const int c_ = (i_) / (IH * IW), rem_ = (i_) - c_ * (IH * IW);                                           \
        const int ly_ = rem_ / IW, lx_ = rem_ - ly_ * IW;                                                        \
        const int gy_ = y0 - R + ly_, gx_ = x0 - R + lx_;                                                        \
        const int xo_ = (i_) < nxe ? s_xoff[c_] : -1;                                                            \
        const int sy_ = ups ? gy_ >> 1 : gy_, sx_ = ups ? gx_ >> 1 : gx_;                                        \
        const bool ok_ = xo_ >= 0 && gy_ >= 0 && gy_ < h && gx_ >= 0 && gx_ < w;                                 \
        const float t_v = xz[ok_ ? xo_ * hwi + (int64_t)sy_ * wi + sx_ : 0];                                     \
        dst_ = ok_ ? t_v : 0.f;                                                                                  \
    }
#define LLDWT_WG_STORE_X(src_, i_)                                                                               \
    if ((i_) < nxe) {                                                                                            \
        const int c_ = (i_) / (IH * IW), rem_ = (i_) - c_ * (IH * IW);                                           \
        lx[c_ * PSX + rem_] = src_;                                                                              \
    }

    for (int q = blockIdx.y; q < total; q += a.zsplit) {
        const int b_ = q / ntile, t_ = q - b_ * ntile;
        const int64_t z_ = (int64_t)plane * a.batch + b_;
        const float* dyz = a.dy + z_ * d.ytot * hw;
        const float* xz = a.x + z_ * xtot * hwi;
        const int y0 = (t_ / tiles_x) * WG_TH, x0 = (t_ % tiles_x) * WG_TW;
        __syncthreads();
        // stage dY: MT*16 channels x 128 px (zero outside the image / beyond cout_g); all loads of a thread are issued
        // before its first LDS store (a load -> store loop waits vmcnt(0) per element and serialises the latency)
        f4u ra[NV];
#pragma unroll
        for (int r = 0; r < NV; ++r) LLDWT_WG_ISSUE4(ra[r], dyz, s_aoff, hw, w, tid + r * 256)
#define LLDWT_WG_FINISH_A()                                                                                      \
        _Pragma("unroll") for (int r = 0; r < NV; ++r) {                                                         \
            floatx4 v4_;                                                                                         \
            LLDWT_WG_FIX4(v4_, ra[r], dyz, s_aoff, hw, w, tid + r * 256)                                         \
            LLDWT_WG_STORE4(la, WG_PSA, v4_, tid + r * 256)                                                      \
        }
        if (xvec) {
            // 1x1: nic channels x 128 px, rows of 16 contiguous pixels (PSX = 129: scalar LDS stores)
            for (int i0 = 0; i0 < nic * (WG_PX / 4); i0 += 8 * 256) {
                f4u rx[8];
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                    const int iv = i0 + tid + r * 256;
                    LLDWT_WG_ISSUE4(rx[r], xz, s_xoff, hwi, wi, (iv < nic * (WG_PX / 4) ? iv : 0))
                }
                if (i0 == 0) { LLDWT_WG_FINISH_A() }
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                    const int iv = i0 + tid + r * 256;
                    if (iv < nic * (WG_PX / 4)) {
                        floatx4 v4_;
                        LLDWT_WG_FIX4(v4_, rx[r], xz, s_xoff, hwi, wi, iv)
                        float* q_ = lx + (iv / (WG_PX / 4)) * PSX + 4 * (iv % (WG_PX / 4));
                        q_[0] = v4_[0]; q_[1] = v4_[1]; q_[2] = v4_[2]; q_[3] = v4_[3];
                    }
                }
            }
        } else {
            // X patch: nic channels x IH x IW with the halo, 8 loads in flight per thread
            for (int i0 = 0; i0 < nxe; i0 += 8 * 256) {
                float v[8];
#pragma unroll
                for (int r = 0; r < 8; ++r) LLDWT_WG_LOAD_X(v[r], i0 + tid + r * 256)
                if (i0 == 0) { LLDWT_WG_FINISH_A() }
#pragma unroll
                for (int r = 0; r < 8; ++r) LLDWT_WG_STORE_X(v[r], i0 + tid + r * 256)
            }
        }
#undef LLDWT_WG_FINISH_A
        __syncthreads();
#pragma unroll 4
        for (int s = 0; s < WG_PX / 4; ++s) {
            const int p = 4 * s;                       // 4 consecutive pixels of one row
            const int poff = (p / WG_TW) * IW + (p % WG_TW) + kk;
            float A[MT], B[WNT];
#pragma unroll
            for (int m = 0; m < MT; ++m) A[m] = la[(m * 16 + col) * WG_PSA + p + kk];
#pragma unroll
            for (int j = 0; j < WNT; ++j) B[j] = lx[bbase[j] + (isb[j] ? 0 : poff)];
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int j = 0; j < WNT; ++j)
                    acc[m][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[m], B[j], acc[m][j], 0, 0, 0);
        }
    }
#undef LLDWT_WG_LOAD_X
#undef LLDWT_WG_STORE_X
    // epilogue: D[row = oc (4*kk + r)][col = n]
    const int KK = KS * KS;
    float* dwp = a.dw + (int64_t)plane * d.cout * cin_g * KK;
#pragma unroll
    for (int j = 0; j < WNT; ++j) {
        const int n = n0 + (wave * WNT + j) * 16 + col;
        if (n >= a.n_total) continue;
        const bool bias_col = n == a.n_w;
        const int icl = bias_col ? 0 : n / a.ntaps, tl = bias_col ? 0 : n - icl * a.ntaps;
        const int tap = a.tap_of[tl];
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int ocl = oc0 + m * 16 + 4 * kk + r;
                if (ocl < cout_g) {
                    const int oc = g * cout_g + ocl;
                    if (bias_col) atomicAdd(a.db + (int64_t)plane * d.cout + oc, a.alpha * acc[m][j][r]);
                    else atomicAdd(dwp + ((int64_t)oc * cin_g + icl) * KK + tap, a.alpha * acc[m][j][r]);
                }
            }
    }
}

// ---- lifting P-block shapes (graphs/layers/P_block_v2.py:15-33: 1 -> 16 -> 16 -> 16 -> 1, all taps live, no placement) ----
// With cout = 16 the generic tile wastes most of its work: N = 400 columns are padded to 512, and the 1 <-> 16 convs fill
// 26 of 256 columns (conv1) or 1 of 16 rows (conv4).  Two dedicated kernels:
//   k_wgrad16<K>    16 -> 16: one 16x16 tile per tap (n = ic, no padding); K waves, wave dy owns the K taps of kernel row dy
//   k_wgrad_thin<K> 1 <-> 16: D[m][tap] = sum_p A16[m][p] * B1[p +- tap]; conv1: A16 = dY, B1 = X; conv4: A16 = X, B1 = dY
//                   with the tap offsets mirrored (X[q] dY[q - t] summed over q)
// Both prefetch the next 8x16 pixel chunk into registers while the matrix cores work on the current one.
struct W16Args {
    const float* x;
    const float* dy;
    float* dw;
    float* db;
    int batch, h, w;
    float alpha;
    int8_t tap_of[25];
};

template <int K>
__device__ __forceinline__ void wgrad16_body(const W16Args& a) {
    constexpr int NTH = K * 64, KK = K * K, R = K / 2;
    constexpr int IH = WG_TH + 2 * R, IW = WG_TW + 2 * R, IHW = IH * IW;
    constexpr int PSX = IHW + 2;                       // even (8-byte aligned rows for ds_write_b64), == 18 / 22 (mod 32)
    constexpr int NVR = (IW + 3) / 4;                  // dwordx4 loads per X patch row
    constexpr int NAV = (16 * WG_PX / 4 + NTH - 1) / NTH, NXV = (16 * IH * NVR + NTH - 1) / NTH;
    constexpr int NSM = 16 * WG_PSA + 16 * PSX > 256 * KK ? 16 * WG_PSA + 16 * PSX : 256 * KK;
    __shared__ __attribute__((aligned(16))) float smem[NSM];
    float* la = smem;                 // [16][WG_PSA]
    float* lx = smem + 16 * WG_PSA;   // [16][PSX]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int col = lane & 15, kk = lane >> 4;
    const int plane = blockIdx.z;
    const int h = a.h, w = a.w;
    const int64_t hw = (int64_t)h * w;
    const int tiles_x = (w + WG_TW - 1) / WG_TW, tiles_y = (h + WG_TH - 1) / WG_TH;
    const int ntile = tiles_x * tiles_y, total = a.batch * ntile;
    floatx4 acc[K];
#pragma unroll
    for (int j = 0; j < K; ++j) acc[j] = floatx4{0.f, 0.f, 0.f, 0.f};
    float bsum = 0.f;
    const int bb = col * PSX + wave * IW + kk;
    f4u ra[NAV], rx[NXV];        // raw prefetched vectors (fixed up at store time)

    // row segments of 4 pixels: one dwordx4 load when the segment lies inside the image (safe address + select); segments
    // that straddle the image border (left/right edge tiles only) take the per-pixel path
#define LLDWT_W16_ISSUE(raw_, base_, c_, gy_, gx_, live_)                                                        \
    {                                                                                                            \
        const bool ok4_ = (live_) && (gy_) >= 0 && (gy_) < h && (gx_) >= 0 && (gx_) + 3 < w;                     \
        raw_ = *reinterpret_cast<const f4u*>(base_ + (ok4_ ? (c_) * hw + (int64_t)(gy_) * w + (gx_) : 0));       \
    }
#define LLDWT_W16_FIX(dst_, raw_, base_, c_, gy_, gx_, live_)                                                    \
    {                                                                                                            \
        const bool row_ = (live_) && (gy_) >= 0 && (gy_) < h;                                                    \
        const bool ok4_ = row_ && (gx_) >= 0 && (gx_) + 3 < w;                                                   \
        dst_ = ok4_ ? floatx4{raw_.x, raw_.y, raw_.z, raw_.w} : floatx4{0.f, 0.f, 0.f, 0.f};                     \
        if (row_ && !ok4_ && (gx_) + 3 >= 0 && (gx_) < w) {                                                      \
            const int64_t off_ = (c_) * hw + (int64_t)(gy_) * w + (gx_);                                         \
            _Pragma("unroll") for (int e_ = 0; e_ < 4; ++e_)                                                     \
                if ((gx_) + e_ >= 0 && (gx_) + e_ < w) dst_[e_] = base_[off_ + e_];                              \
        }                                                                                                        \
    }
#define LLDWT_W16_COORDS(q_)                                                                                     \
        const int b_ = (q_) / ntile, t_ = (q_) - b_ * ntile;                                                     \
        const int64_t z_ = (int64_t)plane * a.batch + b_;                                                        \
        const float* dyz = a.dy + z_ * 16 * hw;                                                                  \
        const float* xz = a.x + z_ * 16 * hw;                                                                    \
        const int y0 = (t_ / tiles_x) * WG_TH, x0 = (t_ % tiles_x) * WG_TW;
#define LLDWT_W16_A(i_) const int c = (i_) / (WG_PX / 4), rq = (i_) % (WG_PX / 4);                               \
                        const int gy = y0 + rq / (WG_TW / 4), gx = x0 + 4 * (rq % (WG_TW / 4));
#define LLDWT_W16_X(i_) const int c = (i_) / (IH * NVR), rem = (i_) - c * (IH * NVR);                            \
                        const int gy = y0 - R + rem / NVR, gx = x0 - R + 4 * (rem % NVR);
#define LLDWT_W16_LOAD(q_)                                                                                       \
    {                                                                                                            \
        LLDWT_W16_COORDS(q_)                                                                                     \
        _Pragma("unroll") for (int r = 0; r < NAV; ++r) {                                                        \
            const int i = tid + r * NTH;                                                                         \
            LLDWT_W16_A(i)                                                                                       \
            LLDWT_W16_ISSUE(ra[r], dyz, c, gy, gx, i < 16 * WG_PX / 4)                                           \
        }                                                                                                        \
        _Pragma("unroll") for (int r = 0; r < NXV; ++r) {                                                        \
            const int i = tid + r * NTH;                                                                         \
            LLDWT_W16_X(i)                                                                                       \
            LLDWT_W16_ISSUE(rx[r], xz, c, gy, gx, i < 16 * IH * NVR)                                             \
        }                                                                                                        \
    }

    int q = blockIdx.x;
    if (q < total) LLDWT_W16_LOAD(q)
    for (; q < total; q += gridDim.x) {
        __syncthreads();
        {
            LLDWT_W16_COORDS(q)
#pragma unroll
            for (int r = 0; r < NAV; ++r) {
                const int i = tid + r * NTH;
                LLDWT_W16_A(i)
                floatx4 v4;
                LLDWT_W16_FIX(v4, ra[r], dyz, c, gy, gx, i < 16 * WG_PX / 4)
                if (i < 16 * WG_PX / 4) {
                    float2* d2 = reinterpret_cast<float2*>(la + c * WG_PSA + 4 * rq);
                    d2[0] = float2{v4[0], v4[1]};
                    d2[1] = float2{v4[2], v4[3]};
                }
            }
#pragma unroll
            for (int r = 0; r < NXV; ++r) {
                const int i = tid + r * NTH;
                LLDWT_W16_X(i)
                floatx4 v4;
                LLDWT_W16_FIX(v4, rx[r], xz, c, gy, gx, i < 16 * IH * NVR)
                if (i < 16 * IH * NVR) {
                    const int lxx = 4 * (rem % NVR);
                    float2* d2 = reinterpret_cast<float2*>(lx + c * PSX + (rem / NVR) * IW + lxx);
                    if (lxx + 1 < IW) d2[0] = float2{v4[0], v4[1]};
                    if (lxx + 3 < IW) d2[1] = float2{v4[2], v4[3]};
                }
            }
        }
        __syncthreads();
        const int qn = q + gridDim.x;
        if (qn < total) LLDWT_W16_LOAD(qn)
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll 4
        for (int s = 0; s < WG_PX / 4; ++s) {
            const int p = 4 * s;
            const int poff = (p / WG_TW) * IW + (p % WG_TW);
            const float A = la[col * WG_PSA + p + kk];
            bsum += A;
            float B[K];
#pragma unroll
            for (int j = 0; j < K; ++j) B[j] = lx[bb + poff + j];
#pragma unroll
            for (int j = 0; j < K; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(A, B[j], acc[j], 0, 0, 0);
        }
    }
#undef LLDWT_W16_LOAD
#undef LLDWT_W16_COORDS
#undef LLDWT_W16_A
#undef LLDWT_W16_X
    // D_tap[row = oc (4*kk + r)][col = ic] -> LDS in dW order, then one coalesced atomic per element: every workgroup of
    // a plane adds to the same 256*KK addresses, and the L2 serialises atomics per cache line -- scattered lanes
    // (stride KK floats) touched 64 lines per instruction and cost more than the GEMM itself.
    float* dwp = a.dw + (int64_t)plane * 16 * 16 * KK;
    __syncthreads();
#pragma unroll
    for (int j = 0; j < K; ++j) {
        const int tap = a.tap_of[wave * K + j];
#pragma unroll
        for (int r = 0; r < 4; ++r) smem[((4 * kk + r) * 16 + col) * KK + tap] = acc[j][r];
    }
    __syncthreads();
    for (int i = tid; i < 256 * KK; i += NTH) atomicAdd(dwp + i, a.alpha * smem[i]);
    if (a.db && wave == 0) {
        bsum += __shfl_xor(bsum, 16, 64);
        bsum += __shfl_xor(bsum, 32, 64);
        if (kk == 0) atomicAdd(a.db + plane * 16 + col, a.alpha * bsum);
    }
}

template <int K>
__global__ __launch_bounds__(K * 64) void k_wgrad16(W16Args a) { wgrad16_body<K>(a); }

// conv3's and conv2's 16 -> 16 weight gradients of one lifting step in ONE launch (blockIdx.y picks the problem): below 0.25 Mpixel
// per plane these launches are latency-bound, and a step has two of them
struct W16Pair { W16Args p[2]; };
template <int K>
__global__ __launch_bounds__(K * 64) void k_wgrad16_2(W16Pair a2) { wgrad16_body<K>(a2.p[blockIdx.y]); }


struct WThinArgs {
    const float* a16;   // 16-channel operand (dY of conv1 / X of conv4)
    const float* b1;    // 1-channel operand  (X of conv1 / dY of conv4)
    float* dw;
    float* db;
    int batch, h, w;
    float alpha;
    int flip;           // conv4: mirrored tap offsets
    int bias_mode;      // 0 none, 1 db[m] = sum A16[m] (conv1), 2 db[0] = sum B1 (conv4)
    int8_t tap_of[25];
};

template <int K>
__device__ __forceinline__ void wgrad_thin_body(const WThinArgs& a) {
    constexpr int KK = K * K, R = K / 2, NJ = (KK + 15) / 16;
    constexpr int IH = WG_TH + 2 * R, IW = WG_TW + 2 * R, IHW = IH * IW;
    constexpr int NAV = 16 * WG_PX / 4 / 256, NB = (IHW + 255) / 256;
    __shared__ __attribute__((aligned(16))) float la[16 * WG_PSA];      // reused by the epilogue: [4 waves][16*KK] partial tiles
    __shared__ float lb[IHW + 8];
    static_assert(4 * 16 * KK <= 16 * WG_PSA, "epilogue scratch");
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int col = lane & 15, kk = lane >> 4;
    const int plane = blockIdx.z;
    const int h = a.h, w = a.w;
    const int64_t hw = (int64_t)h * w;
    const int tiles_x = (w + WG_TW - 1) / WG_TW, tiles_y = (h + WG_TH - 1) / WG_TH;
    const int ntile = tiles_x * tiles_y, total = a.batch * ntile;
    floatx4 acc[NJ];
    int boff[NJ];
    bool bval[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        acc[j] = floatx4{0.f, 0.f, 0.f, 0.f};
        const int n = j * 16 + col;
        bval[j] = n < KK;
        int tdy = bval[j] ? n / K : 0, tdx = bval[j] ? n % K : 0;
        if (a.flip) { tdy = K - 1 - tdy; tdx = K - 1 - tdx; }
        boff[j] = tdy * IW + tdx + kk;
    }
    float bsum = 0.f, bsb = 0.f;
    f4u ra[NAV];                 // raw prefetched vectors (fixed up at store time)
    float vb[NB];

#define LLDWT_WT_COORDS(q_)                                                                                      \
        const int b_ = (q_) / ntile, t_ = (q_) - b_ * ntile;                                                     \
        const int64_t z_ = (int64_t)plane * a.batch + b_;                                                        \
        const float* az = a.a16 + z_ * 16 * hw;                                                                  \
        const float* bz = a.b1 + z_ * hw;                                                                        \
        const int y0 = (t_ / tiles_x) * WG_TH, x0 = (t_ % tiles_x) * WG_TW;
#define LLDWT_WT_LOAD(q_)                                                                                        \
    {                                                                                                            \
        LLDWT_WT_COORDS(q_)                                                                                      \
        _Pragma("unroll") for (int r = 0; r < NAV; ++r) {                                                        \
            const int i = tid + r * 256;                                                                         \
            const int c = i / (WG_PX / 4), rq = i % (WG_PX / 4);                                                 \
            const int gy = y0 + rq / (WG_TW / 4), gx = x0 + 4 * (rq % (WG_TW / 4));                              \
            LLDWT_W16_ISSUE(ra[r], az, c, gy, gx, true)                                                          \
        }                                                                                                        \
        _Pragma("unroll") for (int r = 0; r < NB; ++r) {                                                         \
            const int i = tid + r * 256;                                                                         \
            const int gy = y0 - R + i / IW, gx = x0 - R + i % IW;                                                \
            const bool ok = i < IHW && gy >= 0 && gy < h && gx >= 0 && gx < w;                                   \
            vb[r] = bz[ok ? (int64_t)gy * w + gx : 0];     /* raw: zeroed at the LDS store (a select here is  */  \
                                                           /* pinned above the MFMAs with the load: a wait)   */  \
        }                                                                                                        \
    }

    int q = blockIdx.x;
    if (q < total) LLDWT_WT_LOAD(q)
    for (; q < total; q += gridDim.x) {
        __syncthreads();
        {
            LLDWT_WT_COORDS(q)
            (void)bz;
#pragma unroll
            for (int r = 0; r < NAV; ++r) {
                const int i = tid + r * 256;
                const int c = i / (WG_PX / 4), rq = i % (WG_PX / 4);
                const int gy = y0 + rq / (WG_TW / 4), gx = x0 + 4 * (rq % (WG_TW / 4));
                floatx4 v4;
                LLDWT_W16_FIX(v4, ra[r], az, c, gy, gx, true)
                float2* d2 = reinterpret_cast<float2*>(la + c * WG_PSA + 4 * rq);
                d2[0] = float2{v4[0], v4[1]};
                d2[1] = float2{v4[2], v4[3]};
            }
#pragma unroll
            for (int r = 0; r < NB; ++r) {
                const int i = tid + r * 256;
                if (i < IHW) {
                    const int ly = i / IW, lxx = i % IW;
                    const int gy = y0 - R + ly, gx = x0 - R + lxx;
                    const float v = (gy >= 0 && gy < h && gx >= 0 && gx < w) ? vb[r] : 0.f;
                    lb[i] = v;
                    if (ly >= R && ly < R + WG_TH && lxx >= R && lxx < R + WG_TW) bsb += v;
                }
            }
        }
        __syncthreads();
        const int qn = q + gridDim.x;
        if (qn < total) LLDWT_WT_LOAD(qn)
        __builtin_amdgcn_sched_barrier(0);
        // the 4 waves split the chunk's 32 k-steps
#pragma unroll
        for (int s8 = 0; s8 < WG_PX / 16; ++s8) {
            const int p = 4 * (wave * (WG_PX / 16) + s8);
            const int poff = (p / WG_TW) * IW + (p % WG_TW);
            const float A = la[col * WG_PSA + p + kk];
            bsum += A;
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const float B = bval[j] ? lb[boff[j] + poff] : 0.f;
                acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(A, B, acc[j], 0, 0, 0);
            }
        }
    }
#undef LLDWT_WT_LOAD
#undef LLDWT_WT_COORDS
#undef LLDWT_W16_ISSUE
#undef LLDWT_W16_FIX
    // the 4 waves hold partial tiles over disjoint pixels: sum them in LDS, then one coalesced atomic per element
    float* dwp = a.dw + (int64_t)plane * 16 * KK;
    __syncthreads();
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int n = j * 16 + col;
        if (n < KK) {
            const int tap = a.tap_of[n];
#pragma unroll
            for (int r = 0; r < 4; ++r) la[wave * 16 * KK + (4 * kk + r) * KK + tap] = acc[j][r];
        }
    }
    __syncthreads();
    for (int i = tid; i < 16 * KK; i += 256)
        atomicAdd(dwp + i, a.alpha * ((la[i] + la[16 * KK + i]) + (la[2 * 16 * KK + i] + la[3 * 16 * KK + i])));
    // bias: one atomic instruction per workgroup (every workgroup of a plane hits the same cache line)
    if (a.bias_mode == 1) {
        bsum += __shfl_xor(bsum, 16, 64);
        bsum += __shfl_xor(bsum, 32, 64);
        if (kk == 0) lb[wave * 16 + col] = bsum;
        __syncthreads();
        if (tid < 16) atomicAdd(a.db + plane * 16 + tid, a.alpha * ((lb[tid] + lb[16 + tid]) + (lb[32 + tid] + lb[48 + tid])));
    } else if (a.bias_mode == 2) {
        bsb = wave_sum(bsb);
        if (lane == 0) lb[wave] = bsb;
        __syncthreads();
        if (tid == 0) atomicAdd(a.db + plane, a.alpha * ((lb[0] + lb[1]) + (lb[2] + lb[3])));
    }
}

template <int K>
__global__ __launch_bounds__(256) void k_wgrad_thin(WThinArgs a) { wgrad_thin_body<K>(a); }

// the two thin weight gradients of one lifting step (conv4: X = t3, dY = g; conv1: X = skip, dY = dr) in ONE launch: blockIdx.y picks
// the problem.  Below level 1 these launches are latency-bound (a few chunks per workgroup), and a step has two of them.
struct WThinPair { WThinArgs p[2]; };
template <int K>
__global__ __launch_bounds__(256) void k_wgrad_thin2(WThinPair a2) { wgrad_thin_body<K>(a2.p[blockIdx.y]); }

// ---- 1x1 convs without channel placement (cgp stack, auto-encoder MLPs): dW[m][n] = sum_px dY[m][px] X[n][px] is a plain
// "NT" GEMM with K = pixels and no spatial structure, so the image is walked as flat 32-pixel segments (128-byte rows,
// dwordx4 loads).  A workgroup owns ALL of dW of one (plane, group) when it fits the tile -- (MT x NT) 16x16 tiles on a
// WGM x WGN wave grid -- and splits K with the other workgroups of the group; the bias gradient is the extra column
// n == cin_g whose B operand is the constant 1.  The generic kernel's 64x64 tile re-stages 64 rows of dY per 64 columns.
struct W1Args {
    const float* x;
    const float* dy;
    float* dw;
    float* db;
    int batch, cin, cout, groups;
    int64_t hw;
    float alpha;
    const float* x2;    // split input (lldwt_wgrad1x1_split): rows >= split of every group come from x2 (Z, groups*(cin_g - split), hw),
    int split;          // rows < split from x (Z, groups*split, hw); x2 == null: one tensor (Z, cin, hw)
};
constexpr int W1_KC = 32;           // pixels per chunk
constexpr int W1_PS = W1_KC + 2;    // LDS row stride: == 2 (mod 32) dwords, 8-byte aligned rows

template <int WGM, int WGN, int WTM, int WTN>
__global__ __launch_bounds__(256) void k_wgrad1x1(W1Args a) {
    static_assert(WGM * WGN == 4, "4 waves");
    constexpr int MR = WGM * WTM * 16, NR = WGN * WTN * 16;       // rows of dY / X staged per chunk
    constexpr int NV = (MR + NR) * (W1_KC / 4) / 256;             // dwordx4 per thread and chunk
    static_assert((MR + NR) * (W1_KC / 4) % 256 == 0, "staging divides evenly");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* la = lds;                  // [MR][W1_PS]
    float* lb = lds + MR * W1_PS;     // [NR][W1_PS]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int col = lane & 15, kk = lane >> 4;
    const int wm = wave / WGN, wn = wave % WGN;
    const int cin_g = a.cin / a.groups, cout_g = a.cout / a.groups;
    const int m_blocks = (cout_g + MR - 1) / MR;                 // row blocks of dW per group (blockIdx.y = g*m_blocks + mb)
    const int g = blockIdx.y / m_blocks, m0 = (blockIdx.y % m_blocks) * MR, plane = blockIdx.z;
    const int64_t hw = a.hw;
    const int cpi = (int)((hw + W1_KC - 1) / W1_KC);             // chunks per image
    const int total = a.batch * cpi;
    floatx4 acc[WTM][WTN];
#pragma unroll
    for (int i = 0; i < WTM; ++i)
#pragma unroll
        for (int j = 0; j < WTN; ++j) acc[i][j] = floatx4{0.f, 0.f, 0.f, 0.f};
    // per-thread staging rows (chunk independent): vector v of row `row`; rows < MR are dY channels, the rest X channels
    for (int q = blockIdx.x; q < total; q += gridDim.x) {
        const int b = q / cpi;
        const int64_t p0 = (int64_t)(q - b * cpi) * W1_KC;
        const int64_t z = (int64_t)plane * a.batch + b;
        const float* dyz = a.dy + (z * a.cout + (int64_t)g * cout_g + m0) * hw + p0;
        const int csp = a.x2 ? a.split : cin_g + 1;                  // rows >= csp: second source
        const float* xz = a.x2 ? a.x + (z * a.groups + g) * (int64_t)a.split * hw + p0 : a.x + (z * a.cin + (int64_t)g * cin_g) * hw + p0;
        const float* xz2 = a.x2 ? a.x2 + (z * a.groups + g) * (int64_t)(cin_g - a.split) * hw + p0 - (int64_t)a.split * hw : xz;
        const int npx = (int)(hw - p0 < W1_KC ? hw - p0 : W1_KC);     // valid pixels of this chunk
        f4u raw[NV];
#pragma unroll
        for (int r = 0; r < NV; ++r) {
            const int i = tid + r * 256;
            const int row = i / (W1_KC / 4), v4 = 4 * (i % (W1_KC / 4));
            const bool isa = row < MR;
            const int ch = isa ? row : row - MR;
            const bool ok = (isa ? m0 + ch < cout_g : ch < cin_g) && v4 + 3 < npx;
            const float* src = isa ? dyz : (ch < csp ? xz : xz2);
            raw[r] = *reinterpret_cast<const f4u*>(ok ? src + (int64_t)ch * hw + v4 : (isa ? dyz : xz));
        }
        __syncthreads();                                   // the previous chunk's LDS reads are done
#pragma unroll
        for (int r = 0; r < NV; ++r) {
            const int i = tid + r * 256;
            const int row = i / (W1_KC / 4), v4 = 4 * (i % (W1_KC / 4));
            const bool isa = row < MR;
            const int ch = isa ? row : row - MR;
            const bool chan = isa ? m0 + ch < cout_g : ch < cin_g;
            const bool ok = chan && v4 + 3 < npx;
            floatx4 v = ok ? floatx4{raw[r].x, raw[r].y, raw[r].z, raw[r].w} : floatx4{0.f, 0.f, 0.f, 0.f};
            if (chan && !ok && v4 < npx) {                 // ragged end of the image
                const float* src = (isa ? dyz : (ch < csp ? xz : xz2)) + (int64_t)ch * hw + v4;
#pragma unroll
                for (int e = 0; e < 3; ++e)
                    if (v4 + e < npx) v[e] = src[e];
            }
            if (!isa && ch == cin_g && a.db) {             // bias column: constant 1 over the valid pixels
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = v4 + e < npx ? 1.f : 0.f;
            }
            float2* d2 = reinterpret_cast<float2*>((isa ? la + ch * W1_PS : lb + ch * W1_PS) + v4);
            d2[0] = float2{v[0], v[1]};
            d2[1] = float2{v[2], v[3]};
        }
        __syncthreads();
#pragma unroll
        for (int s = 0; s < W1_KC / 4; ++s) {
            float A[WTM], B[WTN];
#pragma unroll
            for (int i = 0; i < WTM; ++i) A[i] = la[((wm * WTM + i) * 16 + col) * W1_PS + 4 * s + kk];
#pragma unroll
            for (int j = 0; j < WTN; ++j) B[j] = lb[((wn * WTN + j) * 16 + col) * W1_PS + 4 * s + kk];
#pragma unroll
            for (int i = 0; i < WTM; ++i)
#pragma unroll
                for (int j = 0; j < WTN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[i], B[j], acc[i][j], 0, 0, 0);
        }
    }
    // D[row = m (4*kk + r)][col = n]
    float* dwp = a.dw + ((int64_t)plane * a.cout + (int64_t)g * cout_g) * cin_g;
    float* dbp = a.db ? a.db + (int64_t)plane * a.cout + (int64_t)g * cout_g : nullptr;
#pragma unroll
    for (int j = 0; j < WTN; ++j) {
        const int n = (wn * WTN + j) * 16 + col;
#pragma unroll
        for (int i = 0; i < WTM; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m0 + (wm * WTM + i) * 16 + 4 * kk + r;
                if (m < cout_g) {
                    if (n < cin_g) atomicAdd(dwp + (int64_t)m * cin_g + n, a.alpha * acc[i][j][r]);
                    else if (n == cin_g && dbp) atomicAdd(dbp + m, a.alpha * acc[i][j][r]);
                }
            }
    }
}

template <int WGM, int WGN, int WTM, int WTN>
static int launch_wgrad1x1(const W1Args& a, int planes, hipStream_t st) {
    constexpr int MR = WGM * WTM * 16, NR = WGN * WTN * 16;
    const size_t shmem = sizeof(float) * (size_t)(MR + NR) * W1_PS;
    auto kern = k_wgrad1x1<WGM, WGN, WTM, WTN>;
    if (shmem > 64 * 1024 &&
        hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem) != hipSuccess) {
        set_error("conv2d_wgrad: cannot reserve %zu bytes of LDS", shmem);
        return LLDWT_EHIP;
    }
    int per_cu = 2;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)kern, 256, shmem) != hipSuccess || per_cu < 1) per_cu = 2;
    const int64_t chunks = a.batch * cdiv(a.hw, W1_KC);
    const int64_t m_blocks = cdiv(a.cout / a.groups, MR);
    int64_t slices = (int64_t)lldwt_num_cus() * per_cu / ((int64_t)a.groups * m_blocks * planes);     // one resident round
    if (slices > chunks / 8) slices = chunks / 8;
    if (slices < 1) slices = 1;
    hipLaunchKernelGGL(kern, dim3((unsigned)slices, (unsigned)(a.groups * m_blocks), (unsigned)planes), dim3(256), shmem, st, a);
    return check_launch("conv2d_wgrad");
}

__device__ __forceinline__ float act_bwd_one(float g, float yv, int act) {
    return act == LLDWT_ACT_TANH ? g * (1.f - yv * yv)
                                 : (act == LLDWT_ACT_LRELU ? (yv > 0.f ? g : 0.01f * g) : (act == LLDWT_ACT_RELU ? (yv > 0.f ? g : 0.f) : g));
}
__global__ void k_act_bwd(const float* __restrict__ dy, const float* __restrict__ y, float* __restrict__ dx, int64_t n,
                          int act) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        dx[i] = act_bwd_one(dy[i], y[i], act);
}
// 16 bytes per lane (n4 = n / 4 groups; the launcher checks the alignment), one group per lane and workgroup: many short workgroups
__global__ __launch_bounds__(256) void k_act_bwd4(const float4* __restrict__ dy, const float4* __restrict__ y,
                                                  float4* __restrict__ dx, int64_t n4, int act) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        const float4 g = dy[i], yv = y[i];
        dx[i] = float4{act_bwd_one(g.x, yv.x, act), act_bwd_one(g.y, yv.y, act), act_bwd_one(g.z, yv.z, act),
                       act_bwd_one(g.w, yv.w, act)};
    }
}

__global__ void k_downsum2(const float* __restrict__ g, float* __restrict__ out, int64_t zc, int h, int w) {
    const int ho = h / 2, wo = w / 2;
    const int64_t n = zc * ho * wo;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t c = i / (ho * wo);
        const int rem = (int)(i - c * ho * wo);
        const int y = rem / wo, x = rem - y * wo;
        const float* p = g + c * (int64_t)h * w + (int64_t)(2 * y) * w + 2 * x;
        out[i] = (p[0] + p[1]) + (p[w] + p[w + 1]);
    }
}

template <int KS, int MT, int NT>
static int launch_wgrad(WgradArgs& a, int planes, int64_t chunks, hipStream_t st) {
    const lldwt_conv_desc& d = a.d;
    const int cout_g = d.cout / d.groups;
    const int n_tiles = (a.n_total + NT * 16 - 1) / (NT * 16);
    const int m_tiles = (cout_g + MT * 16 - 1) / (MT * 16);
    constexpr int R = KS / 2;
    constexpr int IHW = (WG_TH + 2 * R) * (WG_TW + 2 * R), PSX = IHW + 1;
    // distinct input channels per n-block: ceil(NT*16 / ntaps) + 1
    const int cin_g_ = d.cin / d.groups;
    a.nic_max = min((NT * 16 + a.ntaps - 1) / a.ntaps + 1, cin_g_);
    const size_t shmem = sizeof(float) * ((size_t)MT * 16 * WG_PSA + 2 + (size_t)a.nic_max * PSX + a.nic_max + MT * 16);
    if (shmem > 160 * 1024) {
        set_error("conv2d_wgrad: tile needs %zu bytes of LDS", shmem);
        return LLDWT_EINVAL;
    }
    auto kern = k_conv_wgrad<KS, MT, NT>;
    if (shmem > 64 * 1024 &&
        hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem) != hipSuccess) {
        set_error("conv2d_wgrad: cannot reserve %zu bytes of LDS", shmem);
        return LLDWT_EHIP;
    }
    // split K (images x 8x16 chunks) over blockIdx.y: about 8 workgroups per CU-slot, and a count that fills whole
    // rounds of the chip -- workgroups take equal time, so 2052 of them on 512 slots would run a fifth round for 4
    const int64_t out_tiles = (int64_t)n_tiles * m_tiles * d.groups * planes;
    int per_cu = 2;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)kern, 256, shmem) != hipSuccess || per_cu < 1) per_cu = 2;
    const int64_t slots = (int64_t)lldwt_num_cus() * per_cu;
    int64_t hi = chunks / 4;                       // every slice accumulates >= 4 chunks before its atomics
    if (hi < 1) hi = 1;
    if (hi > 65535) hi = 65535;
    int64_t sp0 = cdiv(4 * slots, out_tiles);
    if (sp0 > hi) sp0 = hi;
    if (sp0 < 1) sp0 = 1;
    int64_t best = sp0;
    double best_eff = 0.0;
    for (int64_t sp = sp0; sp >= 1 && sp >= sp0 - sp0 / 3; --sp) {
        const int64_t tot = out_tiles * sp;
        const double eff = (double)tot / (double)(cdiv(tot, slots) * slots);
        if (eff > best_eff + 1e-9) { best_eff = eff; best = sp; }
    }
    a.zsplit = (int)best;
    dim3 grid((unsigned)(n_tiles * m_tiles * d.groups), (unsigned)a.zsplit, (unsigned)planes);
    hipLaunchKernelGGL(kern, grid, dim3(256), shmem, st, a);
    return check_launch("conv2d_wgrad");
}

static inline unsigned ew_grid2(int64_t n) {
    int64_t g = cdiv(n, 256);
    return (unsigned)(g < 1 ? 1 : (g > 4096 ? 4096 : g));
}

}  // namespace lldwt
using namespace lldwt;

extern "C" int lldwt_conv2d_wgrad(const float* x, const float* dy, float* dw, float* dbias, const lldwt_conv_desc* d,
                                  int64_t planes, int64_t batch, int64_t h, int64_t w_, void* stream) {
    return lldwt_conv2d_wgrad_ex(x, dy, dw, dbias, d, planes, batch, h, w_, 1.0f, 0, stream);
}

// LLDWT_W1_TALL=0 keeps the 96 x 192 tile for the 162 x 94 weight gradient
static const int g_w1_tall = [] { const char* e = getenv("LLDWT_W1_TALL"); return e ? atoi(e) : 1; }();

// Weight gradient of a grouped 1x1 conv whose input is two tensors side by side per group: rows < ca of group g from xa
// (planes, batch, groups*ca, hw), the other cb rows from xb (planes, batch, groups*cb, hw) -- layer 0 of the cgp stack with the folded
// context (LiftingBasedDWT_net.py:282-289,353-359: [tree-context channels | gathered taps]) without the concatenated tensor.
// dw (planes, cout, ca + cb) += sum dy (x) [xa | xb], dbias (planes, cout) += sum dy (optional).
extern "C" int lldwt_wgrad1x1_split(const float* xa, const float* xb, const float* dy, float* dw, float* dbias, int64_t planes,
                                    int64_t batch, int64_t hw, int ca, int cb, int cout, int groups, void* stream) {
    LLDWT_REQUIRE(xa && xb && dy && dw && planes > 0 && planes <= 65535 && batch > 0 && hw > 0 && ca > 0 && cb > 0 && groups > 0 &&
                      cout % groups == 0, "wgrad1x1_split: bad arguments");
    W1Args w;
    w.x = xa; w.x2 = xb; w.split = ca; w.dy = dy; w.dw = dw; w.db = dbias; w.batch = (int)batch; w.cin = (ca + cb) * groups;
    w.cout = cout; w.groups = groups; w.hw = hw; w.alpha = 1.f;
    const int cout_g = cout / groups, nb = ca + cb + (dbias ? 1 : 0);
    // 162 x 94 (layer 0 of the cgp stack): ONE 192 x 96 tile of dW per workgroup (72 accumulator registers) -- both operands are read
    // once; the 96 x 192 tile read the input rows twice (two row blocks)
    if (g_w1_tall && cout_g > 96 && cout_g <= 192 && nb > 64 && nb <= 96) return launch_wgrad1x1<2, 2, 6, 3>(w, (int)planes, (hipStream_t)stream);
    if (cout_g > 64 && nb > 64 && nb <= 192) return launch_wgrad1x1<1, 4, 6, 3>(w, (int)planes, (hipStream_t)stream);
    if (cout_g > 16 && cout_g <= 64 && nb > 64 && nb <= 192) return launch_wgrad1x1<1, 4, 4, 3>(w, (int)planes, (hipStream_t)stream);
    set_error("wgrad1x1_split: built for 64 < ca + cb (+1) <= 192 input rows and more than 16 output channels per group (got %d, %d)", nb, cout_g);
    return LLDWT_EINVAL;
}

// conv3's (x = t2, dy = dt3) and conv2's (x = t1, dy = dpre2) 16 -> 16 weight gradients of one lifting step in one launch of k_wgrad16_2
namespace lldwt {
int wgrad16_pair(const float* x3, const float* dy3, float* dw3, float* db3, const float* x2, const float* dy2, float* dw2, float* db2,
                 int64_t planes, int64_t batch, int64_t h, int64_t w_, float alpha, int swap_hw, int K, hipStream_t st) {
    LLDWT_REQUIRE(x3 && dy3 && dw3 && x2 && dy2 && dw2 && (K == 3 || K == 5), "wgrad16_pair: bad arguments");
    LLDWT_REQUIRE(planes > 0 && planes <= 65535 && batch > 0 && h > 0 && w_ > 0, "wgrad16_pair: bad dims");
    const int KK = K * K;
    W16Pair a2;
    for (int i = 0; i < 2; ++i) {
        W16Args& w = a2.p[i];
        w.x = i ? x2 : x3; w.dy = i ? dy2 : dy3; w.dw = i ? dw2 : dw3; w.db = i ? db2 : db3;
        w.batch = (int)batch; w.h = (int)h; w.w = (int)w_; w.alpha = alpha;
        for (int t = 0; t < 25; ++t) w.tap_of[t] = t < KK ? (int8_t)(swap_hw ? (t % K) * K + t / K : t) : 0;
    }
    const int64_t chunks = batch * cdiv(h, WG_TH) * cdiv(w_, WG_TW);
    int per_cu = 2;
    const void* kern = K == 5 ? (const void*)k_wgrad16_2<5> : (const void*)k_wgrad16_2<3>;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, K * 64, 0) != hipSuccess || per_cu < 1) per_cu = 2;
    if (per_cu > 8) per_cu = 8;
    int64_t sl = (int64_t)lldwt_num_cus() * per_cu / (planes * 2);       // one resident round over both problems and all planes
    if (sl > chunks / 4) sl = chunks / 4;
    if (sl < 1) sl = 1;
    dim3 grid((unsigned)sl, 2, (unsigned)planes);
    if (K == 5) hipLaunchKernelGGL(k_wgrad16_2<5>, grid, dim3(320), 0, st, a2);
    else hipLaunchKernelGGL(k_wgrad16_2<3>, grid, dim3(192), 0, st, a2);
    return check_launch("wgrad16_pair");
}
}  // namespace lldwt

// conv4's and conv1's weight gradients of one lifting step (C = 16, K = 3 or 5, all taps) in one launch of k_wgrad_thin2:
//   dw4 (planes,1,16,K,K) += alpha sum t3 (x) g,  db4 += alpha sum g;   dw1 (planes,16,1,K,K) += alpha sum dr (x) skip,  db1 += alpha sum dr
namespace lldwt {
int wgrad_thin_pair(const float* t3, const float* g, float* dw4, float* db4, const float* skip, const float* dr, float* dw1,
                    float* db1, int64_t planes, int64_t batch, int64_t h, int64_t w_, float alpha, int swap_hw, int K, hipStream_t st) {
    LLDWT_REQUIRE(t3 && g && dw4 && skip && dr && dw1 && (K == 3 || K == 5), "wgrad_thin_pair: bad arguments");
    LLDWT_REQUIRE(planes > 0 && planes <= 65535 && batch > 0 && h > 0 && w_ > 0, "wgrad_thin_pair: bad dims");
    const int KK = K * K;
    WThinPair a2;
    for (int i = 0; i < 2; ++i) {
        const bool c4 = i == 0;
        WThinArgs& w = a2.p[i];
        w.a16 = c4 ? t3 : dr; w.b1 = c4 ? g : skip; w.dw = c4 ? dw4 : dw1; w.db = c4 ? db4 : db1;
        w.batch = (int)batch; w.h = (int)h; w.w = (int)w_; w.alpha = alpha; w.flip = c4 ? 1 : 0;
        w.bias_mode = w.db ? (c4 ? 2 : 1) : 0;
        for (int t = 0; t < 25; ++t) w.tap_of[t] = t < KK ? (int8_t)(swap_hw ? (t % K) * K + t / K : t) : 0;
    }
    const int64_t chunks = batch * cdiv(h, WG_TH) * cdiv(w_, WG_TW);
    int per_cu = 2;
    const void* kern = K == 5 ? (const void*)k_wgrad_thin2<5> : (const void*)k_wgrad_thin2<3>;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, 256, 0) != hipSuccess || per_cu < 1) per_cu = 2;
    if (per_cu > 4) per_cu = 4;
    int64_t sl = (int64_t)lldwt_num_cus() * per_cu / (planes * 2);       // one resident round over both problems and all planes
    if (sl > chunks / 4) sl = chunks / 4;
    if (sl < 1) sl = 1;
    dim3 grid((unsigned)sl, 2, (unsigned)planes);
    if (K == 5) hipLaunchKernelGGL(k_wgrad_thin2<5>, grid, dim3(256), 0, st, a2);
    else hipLaunchKernelGGL(k_wgrad_thin2<3>, grid, dim3(256), 0, st, a2);
    return check_launch("wgrad_thin_pair");
}
}  // namespace lldwt

extern "C" int lldwt_conv2d_wgrad_ex(const float* x, const float* dy, float* dw, float* dbias, const lldwt_conv_desc* d,
                                     int64_t planes, int64_t batch, int64_t h, int64_t w_, float alpha, int swap_hw,
                                     void* stream) {
    LLDWT_REQUIRE(x && dy && dw && d, "conv2d_wgrad: null pointer");
    LLDWT_REQUIRE(d->K == 1 || d->K == 3 || d->K == 5, "conv2d_wgrad: K=%d unsupported", d->K);
    LLDWT_REQUIRE(d->groups > 0 && d->cin % d->groups == 0 && d->cout % d->groups == 0, "conv2d_wgrad: bad groups");
    LLDWT_REQUIRE(!d->transposed, "conv2d_wgrad: give the forward (Conv2d-layout) descriptor");
    LLDWT_REQUIRE(planes > 0 && batch > 0 && h > 0 && w_ > 0 && planes <= 65535, "conv2d_wgrad: bad dims");
    LLDWT_REQUIRE(!d->upsample2 || (h % 2 == 0 && w_ % 2 == 0), "conv2d_wgrad: upsample2 needs even dims");
    LLDWT_REQUIRE(d->oc_block > 0 && d->ytot >= d->cout, "conv2d_wgrad: bad output placement");
    hipStream_t st = (hipStream_t)stream;
    WgradArgs a;
    a.x = x; a.dy = dy; a.dw = dw; a.db = dbias; a.d = *d;
    a.batch = (int)batch; a.h = (int)h; a.w = (int)w_; a.alpha = alpha;
    const int KK = d->K * d->K, P = d->K / 2;
    int nt = 0;
    for (int t = 0; t < KK; ++t)
        if ((d->tap_mask >> t) & 1u) {
            a.tdy[nt] = (int8_t)(t / d->K);        // offset inside the halo patch (patch origin = tile origin - R)
            a.tdx[nt] = (int8_t)(t % d->K);
            a.tap_of[nt] = (int8_t)(swap_hw ? (t % d->K) * d->K + t / d->K : t);
            ++nt;
        }
    (void)P;
    LLDWT_REQUIRE(nt > 0, "conv2d_wgrad: empty tap mask");
    a.ntaps = nt;
    const int cin_g = d->cin / d->groups, cout_g = d->cout / d->groups;
    a.n_w = cin_g * nt;
    a.n_total = a.n_w + (dbias ? 1 : 0);
    // 1x1 without placement: whole-dW tiles (see k_wgrad1x1); the column cin_g carries the bias gradient
    if (d->K == 1 && !d->upsample2 && d->oc_block >= d->cout && d->oc_off == 0 && d->ytot == d->cout && d->ic_block == 0) {
        W1Args w;
        w.x = x; w.dy = dy; w.dw = dw; w.db = dbias; w.batch = (int)batch; w.cin = d->cin; w.cout = d->cout;
        w.groups = d->groups; w.hw = h * w_; w.alpha = alpha; w.x2 = nullptr; w.split = 0;
        const int nb = cin_g + (dbias ? 1 : 0);
        // 96 x 192 tile (the 162 x 162 layer takes two row blocks: a 192 x 192 tile needs 144 accumulator registers per
        // lane and leaves no room for the staging vectors at 2 waves per SIMD), 64 x 192 for the 54-row layers
        if (cout_g > 64 && nb > 64 && nb <= 192) return launch_wgrad1x1<1, 4, 6, 3>(w, (int)planes, st);
        if (cout_g > 16 && cout_g <= 64 && nb > 64 && nb <= 192) return launch_wgrad1x1<1, 4, 4, 3>(w, (int)planes, st);
    }
    // lifting P-block shapes: dedicated kernels (see k_wgrad16 / k_wgrad_thin)
    {
        const bool plain = d->groups == 1 && !d->upsample2 && nt == KK && d->oc_block >= d->cout && d->oc_off == 0 &&
                           d->ytot == d->cout && d->ic_block == 0 && (d->K == 3 || d->K == 5);
        const int64_t chunks = batch * cdiv(h, WG_TH) * cdiv(w_, WG_TW);
        // one resident round: as many workgroups as the chip holds at once (equal-time workgroups, so a partial second
        // round would idle most CUs), at least 4 chunks each; the memory-bound thin kernel stops at 4 per CU (more
        // workgroups only add contention on the few cache lines of its output)
        auto slices_for = [&](const void* kern, int threads, int cap_per_cu) {
            int per_cu = 2;
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, threads, 0) != hipSuccess || per_cu < 1) per_cu = 2;
            if (per_cu > cap_per_cu) per_cu = cap_per_cu;
            int64_t sl = (int64_t)lldwt_num_cus() * per_cu / planes;
            if (sl > chunks / 4) sl = chunks / 4;
            return sl < 1 ? (int64_t)1 : sl;
        };
        if (plain && d->cin == 16 && d->cout == 16) {
            W16Args w;
            w.x = x; w.dy = dy; w.dw = dw; w.db = dbias; w.batch = (int)batch; w.h = (int)h; w.w = (int)w_; w.alpha = alpha;
            for (int t = 0; t < 25; ++t) w.tap_of[t] = t < KK ? a.tap_of[t] : 0;
            if (d->K == 5) {
                dim3 grid((unsigned)slices_for((const void*)k_wgrad16<5>, 320, 8), 1, (unsigned)planes);
                hipLaunchKernelGGL(k_wgrad16<5>, grid, dim3(320), 0, st, w);
            } else {
                dim3 grid((unsigned)slices_for((const void*)k_wgrad16<3>, 192, 8), 1, (unsigned)planes);
                hipLaunchKernelGGL(k_wgrad16<3>, grid, dim3(192), 0, st, w);
            }
            return check_launch("conv2d_wgrad");
        }
        if (plain && ((d->cin == 1 && d->cout == 16) || (d->cin == 16 && d->cout == 1))) {
            const bool c4 = d->cout == 1;
            WThinArgs w;
            w.a16 = c4 ? x : dy; w.b1 = c4 ? dy : x; w.dw = dw; w.db = dbias;
            w.batch = (int)batch; w.h = (int)h; w.w = (int)w_; w.alpha = alpha; w.flip = c4 ? 1 : 0;
            w.bias_mode = dbias ? (c4 ? 2 : 1) : 0;
            for (int t = 0; t < 25; ++t) w.tap_of[t] = t < KK ? a.tap_of[t] : 0;
            if (d->K == 5) {
                dim3 grid((unsigned)slices_for((const void*)k_wgrad_thin<5>, 256, 4), 1, (unsigned)planes);
                hipLaunchKernelGGL(k_wgrad_thin<5>, grid, dim3(256), 0, st, w);
            } else {
                dim3 grid((unsigned)slices_for((const void*)k_wgrad_thin<3>, 256, 4), 1, (unsigned)planes);
                hipLaunchKernelGGL(k_wgrad_thin<3>, grid, dim3(256), 0, st, w);
            }
            return check_launch("conv2d_wgrad");
        }
    }
    // tile shape: 16 x 256 for the lifting convs (cout/groups <= 16), otherwise 64 rows x 256 columns (3x3 / 5x5) or
    // 64 x 128 (1x1, whose X patch is one LDS row per column).  Wide tiles matter: every n-block re-reads the dY chunk
    // and every m-block the X patch, so the 64 x 64 tile of the first version was bound by L2 -> LDS traffic (measured
    // 64 TFLOP/s on the 243 -> 243 tree conv).
    const bool narrow = cout_g <= 16;
    const int mt_ = narrow ? 1 : 4;
    const int nt_ = narrow ? 16 : (a.n_total > 64 && d->K != 1 ? 16 : 4);
    const int64_t chunks = batch * cdiv(h, WG_TH) * cdiv(w_, WG_TW);
    int r;
#define LLDWT_WG(KS_)                                                                                  \
    r = narrow ? launch_wgrad<KS_, 1, 16>(a, (int)planes, chunks, st)                                  \
               : (nt_ == 4 ? launch_wgrad<KS_, 4, 4>(a, (int)planes, chunks, st)                       \
                           : launch_wgrad<KS_, 4, 16>(a, (int)planes, chunks, st));
    if (d->K == 1) { LLDWT_WG(1) }
    else if (d->K == 3) { LLDWT_WG(3) }
    else { LLDWT_WG(5) }
#undef LLDWT_WG
    return r;
}

extern "C" int lldwt_act_bwd(const float* dy, const float* y, float* dx, int64_t n, int act, void* stream) {
    LLDWT_REQUIRE(dy && y && dx && n >= 0, "act_bwd: bad arguments");
    if (n == 0) return 0;
    static const bool scalar = getenv("LLDWT_ACT_BWD") && atoi(getenv("LLDWT_ACT_BWD")) == 1;     // A/B: the 4-byte kernel
    if (!scalar && (n & 3) == 0 && ((((uintptr_t)dy) | ((uintptr_t)y) | ((uintptr_t)dx)) & 15) == 0) {
        const int64_t n4 = n >> 2, g = cdiv(n4, 256), cap = (int64_t)lldwt_num_cus() * 128;
        hipLaunchKernelGGL(k_act_bwd4, dim3((unsigned)(g > cap ? cap : g)), dim3(256), 0, (hipStream_t)stream,
                           reinterpret_cast<const float4*>(dy), reinterpret_cast<const float4*>(y), reinterpret_cast<float4*>(dx), n4, act);
    } else {
        hipLaunchKernelGGL(k_act_bwd, dim3(ew_grid2(n)), dim3(256), 0, (hipStream_t)stream, dy, y, dx, n, act);
    }
    return check_launch("act_bwd");
}

extern "C" int lldwt_downsum2(const float* g, float* out, int64_t zc, int64_t h, int64_t w_, void* stream) {
    LLDWT_REQUIRE(g && out && zc > 0 && h > 0 && w_ > 0 && h % 2 == 0 && w_ % 2 == 0, "downsum2: bad arguments");
    hipLaunchKernelGGL(k_downsum2, dim3(ew_grid2(zc * (h / 2) * (w_ / 2))), dim3(256), 0, (hipStream_t)stream, g, out, zc,
                       (int)h, (int)w_);
    return check_launch("downsum2");
}
