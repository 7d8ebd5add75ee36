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
    float alpha;
    int8_t tdy[25], tdx[25];
    int8_t tap_of[25];
};

template <int KS, int MT, int NT>
__global__ __launch_bounds__(256) void k_conv_wgrad(WgradArgs a) {
    // wave layout: 4 waves; each wave owns (MT x NT)/4 tiles: split along N
    constexpr int WNT = NT / 4;                  // n tiles per wave
    constexpr int R = KS / 2;
    constexpr int IH = WG_TH + 2 * R, IW = WG_TW + 2 * R;
    constexpr int PSX = IH * IW + 1;
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
    float* lx = lds + MT * 16 * WG_PSA + 1;        // [nic][PSX]; lx[-1] holds the constant 1 of the bias column
    const int h = a.h, w = a.w;
    const int hi = d.upsample2 ? h >> 1 : h, wi = d.upsample2 ? w >> 1 : w;
    const int64_t hw = (int64_t)h * w, hwi = (int64_t)hi * wi;
    const int xtot = d.ic_block > 0 ? d.xtot : d.cin;
    const int icb = d.ic_block > 0 ? d.ic_block : d.cin, ics = d.ic_block > 0 ? d.ic_stride : 0, ico = d.ic_block > 0 ? d.ic_off : 0;

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

    // K (= images x spatial chunks) is split over blockIdx.y: slice s takes chunks s, s + zsplit, ...
    const int tiles_x = (w + WG_TW - 1) / WG_TW, tiles_y = (h + WG_TH - 1) / WG_TH;
    const int ntile = tiles_x * tiles_y;
    {
        for (int q = blockIdx.y; q < a.batch * ntile; q += a.zsplit) {
            const int b = q / ntile, t = q - b * ntile;
            const int64_t z = (int64_t)plane * a.batch + b;
            const float* dyz = a.dy + z * d.ytot * hw;
            const float* xz = a.x + z * xtot * hwi;
            const int y0 = (t / tiles_x) * WG_TH, x0 = (t % tiles_x) * WG_TW;
            __syncthreads();
            // stage dY: MT*16 channels x 128 px (zero outside the image / beyond cout_g).  Loads are issued in batches of
            // 8 before the LDS stores: a plain load -> store loop waits vmcnt(0) per element and serialises the latency.
            constexpr int NA = MT * 16 * WG_PX / 256;
#pragma unroll
            for (int r0 = 0; r0 < NA; r0 += 8) {
                float v[8];
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                    const int i = tid + (r0 + r) * 256;
                    const int c = i / WG_PX, p = i - c * WG_PX;
                    const int gy = y0 + p / WG_TW, gx = x0 + p % WG_TW;
                    const int ocl = oc0 + c;
                    v[r] = 0.f;
                    if (ocl < cout_g && gy < h && gx < w) {
                        const int oc = g * cout_g + ocl;
                        const int ocp = (oc / d.oc_block) * d.oc_stride + d.oc_off + oc % d.oc_block;
                        v[r] = dyz[ocp * hw + (int64_t)gy * w + gx];
                    }
                }
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                    const int i = tid + (r0 + r) * 256;
                    la[(i / WG_PX) * WG_PSA + (i % WG_PX)] = v[r];
                }
            }
            // stage X patch: nic channels x IH x IW
            for (int i0 = 0; i0 < nic * IH * IW; i0 += 8 * 256) {
                float v[8];
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                    const int i = i0 + tid + r * 256;
                    const int c = i / (IH * IW), rem = i - c * (IH * IW);
                    const int ly = rem / IW, lxx = rem - ly * IW;
                    const int gy = y0 - R + ly, gx = x0 - R + lxx;
                    v[r] = 0.f;
                    if (i < nic * IH * IW && gy >= 0 && gy < h && gx >= 0 && gx < w && ic_first + c < cin_g) {
                        const int icg = g * cin_g + ic_first + c;
                        const int icm = (icg / icb) * ics + ico + icg % icb;
                        const int sy = d.upsample2 ? gy >> 1 : gy, sx = d.upsample2 ? gx >> 1 : gx;
                        v[r] = xz[icm * hwi + (int64_t)sy * wi + sx];
                    }
                }
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                    const int i = i0 + tid + r * 256;
                    if (i < nic * IH * IW) {
                        const int c = i / (IH * IW), rem = i - c * (IH * IW);
                        lx[c * PSX + rem] = v[r];
                    }
                }
            }
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
    }
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

// dbias[plane][oc] += sum over batch, pixels of dy (read through the output placement)
__global__ __launch_bounds__(256) void k_bias_grad(const float* __restrict__ dy, float* __restrict__ db, lldwt_conv_desc d,
                                                   int batch, int64_t hw, float alpha) {
    __shared__ float part[4];
    const int oc = blockIdx.x, plane = blockIdx.z, b = blockIdx.y;
    const int ocp = (oc / d.oc_block) * d.oc_stride + d.oc_off + oc % d.oc_block;
    const float* p = dy + (((int64_t)plane * batch + b) * d.ytot + ocp) * hw;
    float s = 0.f;
    for (int64_t i = threadIdx.x; i < hw; i += 256) s += p[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(db + plane * d.cout + oc, alpha * (part[0] + part[1] + part[2] + part[3]));
}

__global__ void k_act_bwd(const float* __restrict__ dy, const float* __restrict__ y, float* __restrict__ dx, int64_t n,
                          int act) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float yv = y[i], g = dy[i];
        dx[i] = act == LLDWT_ACT_TANH ? g * (1.f - yv * yv) : (act == LLDWT_ACT_LRELU ? (yv > 0.f ? g : 0.01f * g) : g);
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
static int launch_wgrad(const WgradArgs& a, int planes, hipStream_t st) {
    const lldwt_conv_desc& d = a.d;
    const int cout_g = d.cout / d.groups;
    const int n_tiles = (a.n_total + NT * 16 - 1) / (NT * 16);
    const int m_tiles = (cout_g + MT * 16 - 1) / (MT * 16);
    constexpr int R = KS / 2;
    constexpr int PSX = (WG_TH + 2 * R) * (WG_TW + 2 * R) + 1;
    // distinct input channels per n-block: ceil(NT*16 / ntaps) + 1
    const int cin_g_ = d.cin / d.groups;
    const int nic_max = min((NT * 16 + a.ntaps - 1) / a.ntaps + 1, cin_g_);
    const size_t shmem = sizeof(float) * ((size_t)MT * 16 * WG_PSA + 1 + (size_t)nic_max * PSX);
    auto kern = k_conv_wgrad<KS, MT, NT>;
    if (shmem > 160 * 1024) {
        set_error("conv2d_wgrad: tile needs %zu bytes of LDS", shmem);
        return LLDWT_EINVAL;
    }
    if (shmem > 64 * 1024 &&
        hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem) != hipSuccess) {
        set_error("conv2d_wgrad: cannot reserve %zu bytes of LDS", shmem);
        return LLDWT_EHIP;
    }
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
    // split K (images x 8x16 chunks) so that the grid fills the chip several times over: ~2048 workgroups
    {
        const bool narrow_ = cout_g <= 16;
        const int nt_ = narrow_ ? 16 : 4, mt_ = narrow_ ? 1 : 4;
        const int64_t out_tiles = cdiv(a.n_total, nt_ * 16) * cdiv(cout_g, mt_ * 16) * d->groups * planes;
        const int64_t chunks = batch * cdiv(h, WG_TH) * cdiv(w_, WG_TW);
        int64_t sp = cdiv(2048, out_tiles);
        if (sp > chunks / 4) sp = chunks / 4;      // every slice accumulates >= 4 chunks before its atomics
        if (sp < 1) sp = 1;
        if (sp > 65535) sp = 65535;
        a.zsplit = (int)sp;
    }
    int r;
    const bool narrow = cout_g <= 16;      // 16 x 256 tile for the lifting convs, 64 x 64 otherwise
#define LLDWT_WG(KS_)                                                                                  \
    r = narrow ? launch_wgrad<KS_, 1, 16>(a, (int)planes, st) : launch_wgrad<KS_, 4, 4>(a, (int)planes, st);
    if (d->K == 1) { LLDWT_WG(1) }
    else if (d->K == 3) { LLDWT_WG(3) }
    else { LLDWT_WG(5) }
#undef LLDWT_WG
    return r;
}

extern "C" int lldwt_act_bwd(const float* dy, const float* y, float* dx, int64_t n, int act, void* stream) {
    LLDWT_REQUIRE(dy && y && dx && n >= 0, "act_bwd: bad arguments");
    if (n == 0) return 0;
    hipLaunchKernelGGL(k_act_bwd, dim3(ew_grid2(n)), dim3(256), 0, (hipStream_t)stream, dy, y, dx, n, act);
    return check_launch("act_bwd");
}

extern "C" int lldwt_downsum2(const float* g, float* out, int64_t zc, int64_t h, int64_t w_, void* stream) {
    LLDWT_REQUIRE(g && out && zc > 0 && h > 0 && w_ > 0 && h % 2 == 0 && w_ % 2 == 0, "downsum2: bad arguments");
    hipLaunchKernelGGL(k_downsum2, dim3(ew_grid2(zc * (h / 2) * (w_ / 2))), dim3(256), 0, (hipStream_t)stream, g, out, zc,
                       (int)h, (int)w_);
    return check_launch("downsum2");
}
