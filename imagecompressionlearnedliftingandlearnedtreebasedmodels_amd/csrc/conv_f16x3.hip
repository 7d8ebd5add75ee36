// conv_f16x3.hip -- dense 3x3 convolution with fp32-level accuracy on the fp16 matrix cores ("split-fp16", f16x3).
//
// Target: the tree-context conv 243 -> 243, 3x3 (LiftingBasedDWT_net.py:271-272, :793-795) -- 61 % of the headline
// step's FLOPs and, on the fp32 MFMA (v_mfma_f32_16x16x4_f32, 64 FLOP/clk/SIMD), bound at 157 TFLOP/s.  The fp16 MFMA
// (v_mfma_f32_32x32x16_f16) runs 16x that rate.  Every fp32 operand is split EXACTLY-to-2^-22 into two fp16 values
//     v = s * x  (s a power of two chosen so that max|v| is in [2^14, 2^15): no overflow, lo stays a normal number)
//     hi = fp16(v),  lo = fp16(v - hi)            |v - hi - lo| <= 2^-22 |v|
// and the product is accumulated in fp32 from three MFMAs:  x*w ~ hi_x*hi_w + hi_x*lo_w + lo_x*hi_w  (the dropped
// lo*lo term is 2^-22 relative).  fp16 x fp16 products are exact in fp32, so the result carries ~2^-21 relative error
// per product -- the same order as the fp32 fmaf chain's own accumulation error at K = 2187 (3.5e-7) -- at 16/3 = 5.3x
// fewer matrix cycles.  bf16 would not do: its 8-bit mantissa gives 2^-16 per split pair, 64x worse.
// The scales are powers of two, so applying 1/(s_x*s_w) to the accumulator is exact.
//   s_w: per plane, fixed at pack time (lldwt_conv_f16x3_pack).   s_x: per plane, from the |x| maximum of the input
//   tensor, produced on the device by lldwt_absmax_slots (no host sync) and read by the conv kernel.
//
// Kernel shape (one 256-thread workgroup per CU, ONE wave per SIMD with the whole 512-entry register file):
//   workgroup tile = 128 output channels x (8 rows x 32 pixels); wave w owns channels 32w..32w+31 x all 256 pixels
//   = 8 accumulator tiles of 32x32 (128 VGPRs).  K loop = 8 chunks of 32 input channels x 9 taps x 2 k-steps of 16.
//   B (activations): fp32 planes -> registers -> split -> LDS image [10x34 pixels][32 ch] fp16, pixel pitch 80 B
//      (20 dwords: 16 lanes of a ds_read_b128 group hit 64 distinct banks), hi and lo images, double-buffered, so the
//      global loads of chunk c+1 fly during the MFMAs of chunk c and there is one barrier per chunk.
//   A (weights): pre-split, pre-packed in lane order per (wave, chunk, tap, k-step): each wave streams ITS fragments
//      straight from L2 into registers (1 KB coalesced per instruction, no LDS), two k-steps ahead.
#include "common.h"
#include "split_f16.h"
#include "lifting_f16.h"      // split_precision(): the one-product fp16 / bf16 modes of the fused pair (lldwt_set_precision)
#include <string.h>
#include <type_traits>

namespace lldwt {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
constexpr int F3_CK = 32;                       // input channels per chunk
constexpr int F3_TH = 8, F3_TW = 32;            // output pixels per workgroup
constexpr int F3_IH = F3_TH + 2, F3_IW = F3_TW + 2, F3_NPX = F3_IH * F3_IW;   // 10 x 34 = 340 staged pixels
constexpr int F3_PITCH = F3_CK * 2 + 16;        // 80 B per pixel
constexpr int F3_PART = F3_NPX * F3_PITCH;      // 27,200 B: one image (hi or lo)
constexpr int F3_BUF = 2 * F3_PART;             // hi + lo
constexpr int F3_LDS = 2 * F3_BUF + 64;         // double buffer: 108,800 B + a 64-B dump slot for dead staging tasks
constexpr int F3_OCB = 128;                     // output channels per workgroup
constexpr int F3_STEP_BYTES = 2048;             // one (tap, k-step) of one wave: hi fragment + lo fragment, 1 KB each
constexpr int F3_CHUNK_BYTES = 9 * 2 * F3_STEP_BYTES;
constexpr int F3_HDR = 256;                     // per-plane header: [0] = s_w
constexpr int F3_NTASK = F3_NPX * (F3_CK / 8);  // staging tasks (pixel, group of 8 channels) per chunk
constexpr int F3_R = (F3_NTASK + 255) / 256;    // per thread: 6

constexpr int F1_HDR = 2048;                    // conv1 pack: [0] s_w1, [1] max row L1 norm, [2] max |b|; floats [16 .. 16+256) bias
constexpr int F1_CHUNK_BYTES = 2 * F3_STEP_BYTES;   // 2 k-steps (K = 27 -> 32) x (hi, lo)
constexpr int F1_NBLK = (F3_NPX + 31) / 32;     // 11 pixel blocks of 32 cover the 340-pixel patch (12 slots: 3 per wave)
constexpr int F1_COL = 12 * 4 * 1024;           // the split im2col of the parent patch, [block][k-step][hi|lo][lane][8 x fp16]
constexpr int F1_PP = 3 * 6 * 18;               // parent values under a tile (see the kernel)
constexpr int F1_LDS = F3_LDS + F1_COL + 1536;  // + the parent patch: 159,552 B
// behind the (hi, lo) fp16 fragments: a bf16 copy of the scaled weights, half a step's bytes per step (one-product bf16 mode)
static inline int64_t f1_bf_off(int cmid) { return F1_HDR + (int64_t)cdiv(cmid, F3_CK) * F1_CHUNK_BYTES; }
static inline int64_t f1_plane_bytes(int cmid) { return f1_bf_off(cmid) + (int64_t)cdiv(cmid, F3_CK) * (F1_CHUNK_BYTES / 2); }

static inline int f3_nch(int cin) { return (int)cdiv(cin, F3_CK); }
static inline int f3_nocb(int cout) { return (int)cdiv(cout, F3_OCB); }
static inline __host__ __device__ int64_t f3_bf_off_n(int nocb, int nch) {
    // + 5 steps of padding: the kernel prefetches weight fragments 5 steps ahead without a bounds branch
    return F3_HDR + (int64_t)nocb * 4 * nch * F3_CHUNK_BYTES + 5 * F3_STEP_BYTES;
}
static inline int64_t f3_plane_bytes(int cin, int cout) {
    // the (hi, lo) fp16 steps, then a bf16 copy at half the bytes per step (same 5 steps of padding)
    return f3_bf_off_n(f3_nocb(cout), f3_nch(cin)) + (int64_t)f3_nocb(cout) * 4 * f3_nch(cin) * (F3_CHUNK_BYTES / 2) + 5 * (F3_STEP_BYTES / 2);
}

__device__ __forceinline__ float pow2_scale_for(float amax) {
    // power of two s with amax * s in [2^14, 2^15); 1 for amax == 0 / non-finite
    if (!(amax > 0.f) || !(amax < 3.0e38f)) return 1.f;
    int e;
    (void)frexpf(amax, &e);                      // amax = m * 2^e, m in [0.5, 1)
    int k = 15 - e;
    k = k > 120 ? 120 : (k < -120 ? -120 : k);
    return ldexpf(1.f, k);
}

// ---- |x| maximum per plane into 64 slots (spreads the atomics; the consumer takes the max of the 64)
__global__ void k_absmax_slots(const float* __restrict__ x, int64_t n_per_plane, float* __restrict__ slots, int vec) {
    const int plane = blockIdx.y;
    const float* xp = x + (int64_t)plane * n_per_plane;
    float m = 0.f;
    if (vec) {                     // every plane base is 16-byte aligned and n_per_plane % 4 == 0 (checked by the host)
        const float4* p = reinterpret_cast<const float4*>(xp);
        const int64_t n4 = n_per_plane >> 2;
        for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
            const float4 v = p[i];
            m = fmaxf(m, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
        }
    } else {
        for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n_per_plane; i += (int64_t)gridDim.x * blockDim.x)
            m = fmaxf(m, fabsf(xp[i]));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_down(m, o, 64));
    if ((threadIdx.x & 63) == 0 && m > 0.f)
        atomicMax(reinterpret_cast<int*>(slots + plane * 64 + (blockIdx.x & 63)), __float_as_int(m));   // m >= 0: int order == float order
}

// ---- weight pack: (planes, cout, cin, 3, 3) fp32 -> per plane [hdr][ocb][wave][chunk][tap][ks][hi|lo][lane][8 x fp16]
__global__ void k_f3_wmax(const float* __restrict__ w, int64_t n_per_plane, float* __restrict__ hdr_base, int64_t plane_bytes) {
    const int plane = blockIdx.y;
    float m = 0.f;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n_per_plane; i += (int64_t)gridDim.x * blockDim.x)
        m = fmaxf(m, fabsf(w[(int64_t)plane * n_per_plane + i]));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_down(m, o, 64));
    float* hdr = reinterpret_cast<float*>(reinterpret_cast<char*>(hdr_base) + (int64_t)plane * plane_bytes);
    if ((threadIdx.x & 63) == 0 && m > 0.f) atomicMax(reinterpret_cast<int*>(hdr + 1), __float_as_int(m));   // hdr[1] = max|w|
}

// shape16: fragments for v_mfma_f32_16x16x32_f16 -- the two "k-steps" of a tap become the two 16-channel halves of the wave's 32
// output channels, each spanning the chunk's 32 input channels: A[row = lane&15][k = 8*(lane>>4) + j]
__global__ void k_f3_pack(const float* __restrict__ w, uint8_t* __restrict__ packed, int cin, int cout, int64_t plane_bytes,
                          int shape16) {
    const int plane = blockIdx.y;
    uint8_t* pp = packed + (int64_t)plane * plane_bytes;
    float* hdr = reinterpret_cast<float*>(pp);
    const float sw = pow2_scale_for(hdr[1]);
    const int nch = (cin + F3_CK - 1) / F3_CK, nocb = (cout + F3_OCB - 1) / F3_OCB;
    const int64_t nfrag_elems = (int64_t)nocb * 4 * nch * 9 * 2 * 64 * 8;       // (hi, lo) pairs
    const float* wp = w + (int64_t)plane * cout * cin * 9;
    _Float16* out = reinterpret_cast<_Float16*>(pp + F3_HDR);
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < nfrag_elems; i += (int64_t)gridDim.x * blockDim.x) {
        int64_t r = i;
        const int j = (int)(r % 8); r /= 8;
        const int lane = (int)(r % 64); r /= 64;
        const int ks = (int)(r % 2); r /= 2;
        const int tap = (int)(r % 9); r /= 9;
        const int chunk = (int)(r % nch); r /= nch;
        const int wv = (int)(r % 4); r /= 4;
        const int ocb = (int)r;
        const int oc = shape16 ? ocb * F3_OCB + wv * 32 + ks * 16 + (lane & 15)
                               : ocb * F3_OCB + wv * 32 + (lane & 31);          // A[row = lane&31][k = 8*(lane>>5) + j]
        const int ic = shape16 ? chunk * F3_CK + 8 * (lane >> 4) + j : chunk * F3_CK + ks * 16 + 8 * (lane >> 5) + j;
        float v = 0.f;
        if (oc < cout && ic < cin) v = wp[((int64_t)oc * cin + ic) * 9 + tap] * sw;
        const _Float16 hi = (_Float16)v;
        const _Float16 lo = (_Float16)(v - (float)hi);
        const int64_t step = ((((int64_t)(ocb * 4 + wv) * nch + chunk) * 9 + tap) * 2 + ks);
        out[step * 1024 + lane * 8 + j] = hi;                                   // 1024 halves = 2048 B per step
        out[step * 1024 + 512 + lane * 8 + j] = lo;
        reinterpret_cast<__bf16*>(pp + f3_bf_off_n(nocb, nch))[step * 512 + lane * 8 + j] = (__bf16)v;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) hdr[0] = sw;
}

// ---- FUSED mode: first tree conv (cmid, 3, 3, 3) -> per plane [hdr | chunk][ks][hi|lo][lane][8 x fp16]
// A[row = channel of the chunk][k = ci*9 + tap], k < 27
__global__ void k_f1_pack(const float* __restrict__ w1, const float* __restrict__ b1, uint8_t* __restrict__ packed, int cmid,
                          int64_t plane_bytes) {
    const int plane = blockIdx.x, tid = threadIdx.x;
    const float* wp = w1 + (int64_t)plane * cmid * 27;
    const float* bp = b1 + (int64_t)plane * cmid;
    uint8_t* pp = packed + (int64_t)plane * plane_bytes;
    float* hdr = reinterpret_cast<float*>(pp);
    __shared__ float red[3][4];
    float mw = 0.f, ml1 = 0.f, mb = 0.f;
    for (int r = tid; r < cmid; r += 256) {
        float s1 = 0.f;
        for (int k = 0; k < 27; ++k) {
            const float v = fabsf(wp[r * 27 + k]);
            s1 += v;
            mw = fmaxf(mw, v);
        }
        ml1 = fmaxf(ml1, s1);
        mb = fmaxf(mb, fabsf(bp[r]));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        mw = fmaxf(mw, __shfl_xor(mw, o, 64));
        ml1 = fmaxf(ml1, __shfl_xor(ml1, o, 64));
        mb = fmaxf(mb, __shfl_xor(mb, o, 64));
    }
    if ((tid & 63) == 0) { red[0][tid >> 6] = mw; red[1][tid >> 6] = ml1; red[2][tid >> 6] = mb; }
    __syncthreads();
    const float amw = fmaxf(fmaxf(red[0][0], red[0][1]), fmaxf(red[0][2], red[0][3]));
    const float sw1 = pow2_scale_for(amw);
    if (tid == 0) {
        hdr[0] = sw1;
        hdr[1] = fmaxf(fmaxf(red[1][0], red[1][1]), fmaxf(red[1][2], red[1][3]));
        hdr[2] = fmaxf(fmaxf(red[2][0], red[2][1]), fmaxf(red[2][2], red[2][3]));
    }
    const int nch = (cmid + F3_CK - 1) / F3_CK;
    for (int i = tid; i < nch * F3_CK; i += 256) hdr[16 + i] = i < cmid ? bp[i] : 0.f;
    _Float16* fr = reinterpret_cast<_Float16*>(pp + F1_HDR);
    for (int i = tid; i < nch * 2 * 512; i += 256) {
        const int j = i & 7, lane = (i >> 3) & 63, ks = (i >> 9) & 1, chunk = i >> 10;
        const int c = chunk * F3_CK + (lane & 31), k = 16 * ks + 8 * (lane >> 5) + j;
        float v = 0.f;
        if (c < cmid && k < 27) v = wp[c * 27 + k] * sw1;
        const _Float16 hi = (_Float16)v;
        fr[(chunk * 2 + ks) * 1024 + lane * 8 + j] = hi;
        fr[(chunk * 2 + ks) * 1024 + 512 + lane * 8 + j] = (_Float16)(v - (float)hi);
        reinterpret_cast<__bf16*>(pp + F1_HDR + (int64_t)nch * F1_CHUNK_BYTES)[(chunk * 2 + ks) * 512 + lane * 8 + j] = (__bf16)v;
    }
}

struct F3Args {
    const float* x;
    float* y;
    const uint8_t* packed;
    const float* bias;
    const float* slots;
    const _Float16* x16;     // IN16: the input tensor stored as fp16 (already multiplied by xscale[plane]); x is unused
    const float* xscale;     // IN16: (planes) power-of-two storage scale of x16
    const float* parent;     // FUSED: (planes*batch, 3, h/2, w/2) fp32, the tensor the FIRST tree conv reads (2x-upsampled)
    const uint8_t* packed1;  // FUSED: first conv's split-fp16 fragments + bias + bounds (lldwt_plc_fused_pack1)
    int64_t plane_bytes1;
    int cin, cout, act, batch, h, w, tiles_x, nch;
    int64_t plane_bytes;
    unsigned long long* stamps;   // diagnostics only (lldwt_set_diagnostics kind 1): [workgroup][wave][16] s_memtime stamps
};
// in-kernel clock stamps of a diagnostic run (tools/plc_stamps.py); a null pointer (always, outside that tool) skips them
#define F3_STAMP(i)                                                                                                     \
    if (a.stamps && lane == 0)                                                                                          \
        a.stamps[((((int64_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 4 + wave) * 16 + (i)] =   \
            (i) >= 14 ? __builtin_amdgcn_s_memrealtime() : __builtin_amdgcn_s_memtime();

// IN16 = false: fp32 input, split on the way into LDS, three MFMA products per k-step (fp32-level accuracy).
// IN16 = true : "fp16 storage" (BASELINE configs[4]): the input tensor lives in HBM as fp16 (half the bytes), it is the hi
//               part and there is no lo: two products (w_hi x + w_lo x), 2^-11 relative on the activations.
// MODE 2 (FUSED): the input tensor does not exist.  It is LeakyReLU(conv3x3(3 -> cin) of the 2x-upsampled parent), and
//               each chunk of 32 of its channels is computed ON THE FLY for the 10 x 34 patch by the matrix cores
//               (M = 32 channels, N = pixels, K = 27 -> 32; B = the im2col of the parent patch, gathered and split ONCE per
//               tile and kept in registers) and written straight into the LDS image that the second conv reads: no first
//               conv launch, no 243-channel tensor in HBM (1.5 GB written + 2 GB read per level-0 launch), no global
//               staging loads, no |x|-max pass (the activation scale comes from a per-workgroup bound).
// PREC (FUSED only; lldwt_set_precision): 0 = three products per MAC as above; 1 / 2 = ONE product on fp16 / bf16 operands -- the
//               lo images and lo fragments are neither written nor read, a third of the MFMAs.
// S16: the second conv's MFMAs in the 16x16x32 shape (same cycles per MAC; the chip holds a higher clock on it under this MFMA
//               density -- MI355X guide, DVFS item 7 -- measured here, see DESIGN): wave tile still 32 channels x 256 pixels,
//               as 2 channel halves x 16 blocks of 16 pixels; one MFMA k-step spans the chunk's 32 channels, so a tap is ONE step
//               of 2 A fragments; same LDS image, same number of fragment reads and weight bytes.  Weights packed for the shape
//               (k_f3_pack shape16): the shape is a process-wide choice (LLDWT_PLC_SHAPE=16; the default is 32x32x16, the faster one here).
template <int MODE, int PREC = 0, bool S16 = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void k_conv3_f16x3(F3Args a) {
    constexpr bool IN16 = MODE == 1, FUSED = MODE == 2;
    static_assert(PREC == 0 || FUSED, "the one-product modes exist for the fused pair (eval path)");
    constexpr int SB = PREC == 2 ? F3_STEP_BYTES / 2 : F3_STEP_BYTES;      // bytes per weight step in the section being read
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t z = blockIdx.z;
    const int plane = (int)(z / a.batch);
    const int ocb = blockIdx.y;
    const int ty = blockIdx.x / a.tiles_x, tx = blockIdx.x - ty * a.tiles_x;
    const int y0 = ty * F3_TH, x0 = tx * F3_TW;
    const int h = a.h, w = a.w;
    const int64_t hw = (int64_t)h * w;

    F3_STAMP(0)
    F3_STAMP(14)
    // ---- scales (exact powers of two)
    float sx = 1.f;
    if constexpr (IN16) {
        sx = a.xscale[plane];
    } else if constexpr (FUSED) {
        // set after the parent patch has been gathered (below)
    } else {
        float amax = a.slots[plane * 64 + lane];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o, 64));
        sx = pow2_scale_for(amax);
    }
    const uint8_t* pp = a.packed + (int64_t)plane * a.plane_bytes;
    const float sw = *reinterpret_cast<const float*>(pp);
    float out_scale = (1.f / sx) * (1.f / sw);

    // ---- staging tasks of this thread: (pixel p of the 10x34 patch, group of 8 channels icg).  Branch-free: padding
    // pixels and dead tasks load pixel 0 of a real plane and are zeroed at the split; dead tasks store to a dump slot.
    unsigned pix[F3_R];         // byte offset of the pixel inside a channel plane (0 for padding / dead tasks)
    int icg8[F3_R];             // first channel of the task's group inside the chunk
    int woff[F3_R];             // LDS byte offset inside a part
    bool inimg[F3_R];
#pragma unroll
    for (int r = 0; r < F3_R; ++r) {
        const int task = tid + r * 256;
        const bool live = task < F3_NTASK;
        const int icg = live ? task / F3_NPX : 0, p = live ? task - icg * F3_NPX : 0;
        const int ly = p / F3_IW, lx = p - ly * F3_IW;
        const int gy = y0 - 1 + ly, gx = x0 - 1 + lx;
        const bool in = live && gy >= 0 && gy < h && gx >= 0 && gx < w;
        inimg[r] = in;
        icg8[r] = icg * 8;
        woff[r] = p * F3_PITCH + icg * 16;
        pix[r] = in ? (IN16 ? 2u : 4u) * (unsigned)(gy * w + gx) : 0u;
    }
    using in_t = typename std::conditional<IN16, _Float16, float>::type;
    const in_t* xg = (IN16 ? reinterpret_cast<const in_t*>(a.x16) : reinterpret_cast<const in_t*>(a.x)) + z * (int64_t)a.cin * hw;
    const unsigned hw4 = (IN16 ? 2u : 4u) * (unsigned)hw;
    in_t xin[F3_R][8];

    // loads of staging task R of the chunk whose first channel is C1 (channels past cin-1 read plane cin-1, zeroed later)
#define F3_TASK_LOAD(R, C1)                                                                                           \
    {                                                                                                                 \
        const char* cb_ = reinterpret_cast<const char*>(xg + (int64_t)(C1) * hw);       /* wave-uniform */            \
        const unsigned rcmax_ = (unsigned)(a.cin - 1 - (C1));                                                         \
        _Pragma("unroll") for (int j = 0; j < 8; ++j) {                                                               \
            const unsigned rc_ = min((unsigned)(icg8[R] + j), rcmax_);                                                \
            xin[R][j] = *reinterpret_cast<const in_t*>(cb_ + (rc_ * hw4 + pix[R]));                                   \
        }                                                                                                             \
    }
    // split + store of channels 4*HALF .. 4*HALF+3 of staging task R into the LDS buffer at DST
#define F3_TASK_STORE(R, HALF, C1, DST)                                                                               \
    {                                                                                                                 \
        typedef _Float16 half4_ __attribute__((ext_vector_type(4)));                                                  \
        half4_ hi_, lo_;                                                                                              \
        const bool dead_ = (tid + (R) * 256) >= F3_NTASK;                                                             \
        uint8_t* d_ = dead_ ? lds + 2 * F3_BUF + (HALF) * 8 : (DST) + woff[R] + (HALF) * 8;                           \
        if constexpr (IN16) {                                                                                         \
            _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                           \
                const bool ok_ = inimg[R] && ((C1) + icg8[R] + (HALF) * 4 + j) < a.cin;                               \
                hi_[j] = ok_ ? (_Float16)xin[R][(HALF) * 4 + j] : (_Float16)0.f;                                      \
            }                                                                                                         \
            *reinterpret_cast<half4_*>(d_) = hi_;                                                                     \
        } else {                                                                                                      \
            float v4_[4];                                                                                             \
            _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                           \
                const bool ok_ = inimg[R] && ((C1) + icg8[R] + (HALF) * 4 + j) < a.cin;                               \
                v4_[j] = ok_ ? (float)xin[R][(HALF) * 4 + j] * sx : 0.f;                                              \
            }                                                                                                         \
            split4v(v4_, hi_, lo_);                                                                                   \
            *reinterpret_cast<half4_*>(d_) = hi_;                                                                     \
            *reinterpret_cast<half4_*>(d_ + (dead_ ? 16 : F3_PART)) = lo_;                                            \
        }                                                                                                             \
    }

    typedef float floatx4 __attribute__((ext_vector_type(4)));
    floatx16 acc[S16 ? 1 : 8];
    floatx4 acc16[S16 ? 2 : 1][S16 ? 16 : 1];       // S16: [channel half][pixel block = row * 2 + half row]
    if constexpr (S16) {
#pragma unroll
        for (int hv = 0; hv < 2; ++hv)
#pragma unroll
            for (int n = 0; n < 16; ++n) acc16[hv][n] = floatx4{0.f, 0.f, 0.f, 0.f};
    } else {
#pragma unroll
        for (int n = 0; n < 8; ++n)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[n][q] = 0.f;
    }

    const uint8_t* wbase = pp + (PREC == 2 ? f3_bf_off_n((a.cout + F3_OCB - 1) / F3_OCB, a.nch) : (int64_t)F3_HDR) +
                           ((int64_t)(ocb * 4 + wave) * a.nch) * (18 * SB) + lane * 16;
    // B fragment: pixel column lane&31, k half lane>>5 (S16: pixel lane&15, 8-channel group lane>>4 of the chunk's 32)
    const int boff = S16 ? (lane & 15) * F3_PITCH + (lane >> 4) * 16 : (lane & 31) * F3_PITCH + (lane >> 5) * 16;

    // ---- FUSED: the im2col of the parent patch for this wave's pixel blocks (wave, wave + 4, wave + 8), once per tile
    uint8_t* col1 = lds + F3_LDS + lane * 16;     // each lane writes and reads only its own 16-byte slots: no barrier needed
    int p1[3];
    bool pin1[3];
    float inv1 = 1.f, inv1sx = 1.f;
    const uint8_t* pk1 = nullptr;
    const float* bias1 = nullptr;
    if constexpr (FUSED) {
        pk1 = a.packed1 + (int64_t)plane * a.plane_bytes1;
        const float* h1 = reinterpret_cast<const float*>(pk1);
        bias1 = h1 + 16;
        const int hp = h >> 1, wp_ = w >> 1;
        const float* par = a.parent + z * 3 * (int64_t)hp * wp_;
        const int hh = lane >> 5;
        // the parent values under the tile: 3 channels x 6 rows x 18 columns (the 10 x 34 patch and conv1's one-pixel rim,
        // halved), zero outside the parent image = the zero padding of the upsampled image.  Into LDS once, then every
        // lane gathers its 48 im2col values from there.
        float* PP = reinterpret_cast<float*>(lds + F3_LDS + F1_COL);
        float amax = 0.f;
#pragma unroll
        for (int i = tid; i < F1_PP; i += 256) {
            const int ci = i / 108, rem = i - ci * 108, r = rem / 18, c = rem - r * 18;
            const int Yp = (y0 >> 1) - 1 + r, Xp = (x0 >> 1) - 1 + c;
            const bool in = Yp >= 0 && Yp < hp && Xp >= 0 && Xp < wp_;
            const float v = par[(int64_t)ci * hp * wp_ + min(max(Yp, 0), hp - 1) * wp_ + min(max(Xp, 0), wp_ - 1)];
            const float vz = in ? v : 0.f;
            PP[i] = vz;
            amax = fmaxf(amax, fabsf(vz));
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o, 64));
        float* red = reinterpret_cast<float*>(lds + 2 * F3_BUF + 32);      // second half of the 64-byte dump slot
        if (lane == 0) red[wave] = amax;
        __syncthreads();
        amax = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
        const float s_p = pow2_scale_for(amax);
        sx = pow2_scale_for(amax * h1[1] + h1[2]);                           // bound on |LeakyReLU(conv1)|: max|parent| * L1max + |b|max
        out_scale = (1.f / sx) * (1.f / sw);
        inv1 = (1.f / s_p) * (1.f / h1[0]);
        inv1sx = inv1 * sx;
#pragma unroll
        for (int b = 0; b < 3; ++b) {
            const int p = (wave + 4 * b) * 32 + (lane & 31);
            const bool live = p < F3_NPX;
            const int pc = live ? p : 0;
            const int ly = pc / F3_IW, lx = pc - ly * F3_IW;
            const int gy = y0 - 1 + ly, gx = x0 - 1 + lx;
            p1[b] = live ? p : -1;
            pin1[b] = live && gy >= 0 && gy < h && gx >= 0 && gx < w;
            // PP index of tap (dy, dx) of channel ci for this pixel: ci * 108 + ((ly + dy) >> 1) * 18 + ((lx + dx) >> 1)
            int ro[3], co[3];
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                ro[d] = ((ly + d) >> 1) * 18;
                co[d] = (lx + d) >> 1;
            }
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                float v8[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    // k = 16 ks + 8 hh + j -> (channel, dy, dx) = (k / 9, (k % 9) / 3, k % 3); k >= 27 meets zero weights
                    const int k0 = 16 * ks + j, k1 = k0 + 8;
                    const int q0 = k0 < 27 ? k0 : 26, q1 = k1 < 27 ? k1 : 26;
                    const int i0 = (q0 / 9) * 108 + ro[(q0 % 9) / 3] + co[q0 % 3];
                    const int i1 = (q1 / 9) * 108 + ro[(q1 % 9) / 3] + co[q1 % 3];
                    v8[j] = PP[hh ? i1 : i0] * s_p;
                }
                if constexpr (PREC == 0) {
                    half8 ch_, cl_;
                    split8v(v8, ch_, cl_);
                    *reinterpret_cast<half8*>(col1 + (((wave + 4 * b) * 2 + ks) * 2 + 0) * 1024) = ch_;
                    *reinterpret_cast<half8*>(col1 + (((wave + 4 * b) * 2 + ks) * 2 + 1) * 1024) = cl_;
                } else {
                    *reinterpret_cast<half8*>(col1 + (((wave + 4 * b) * 2 + ks) * 2 + 0) * 1024) = cvt8<PREC>(v8);
                }
            }
        }
    }
    // conv1 operands of the chunk being staged: weight fragments (2 k-steps x hi, lo) and the 16 bias values of this lane's
    // rows -- the same for the wave's three pixel blocks, loaded once per chunk a few units ahead of their first use
    half8 w1h[2], w1l[2];
    floatx4 b1r[4], b1v[4];     // raw bias as loaded; scaled by sx at its FIRST USE (block 0): a multiply at the load site is a
                                // vmcnt(0) wait there, which drains the whole weight ring (vmcnt retires in order)
#define F3_FUSED_LOAD(C1)                                                                                             \
    {                                                                                                                 \
        const uint8_t* w1_ = pk1 + F1_HDR + (PREC == 2 ? (int64_t)a.nch * F1_CHUNK_BYTES : (int64_t)0) +                \
                             (int64_t)((C1) / F3_CK) * (2 * SB) + lane * 16;                                          \
        _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) {                                                            \
            w1h[ks] = *reinterpret_cast<const half8*>(w1_ + ks * SB);                                                 \
            if constexpr (PREC == 0) w1l[ks] = *reinterpret_cast<const half8*>(w1_ + ks * SB + 1024);                 \
        }                                                                                                             \
        _Pragma("unroll") for (int gq = 0; gq < 4; ++gq)                                                              \
            b1r[gq] = *reinterpret_cast<const floatx4*>(bias1 + (C1) + 8 * gq + 4 * (lane >> 5));                     \
    }
    // channels C1 .. C1+31 of the first conv for pixel block B of this wave -> split -> the LDS image at DST
#define F3_FUSED_BLOCK(B, DST)                                                                                        \
    {                                                                                                                 \
        floatx16 t_;                                                                                                  \
        if ((B) == 0) { _Pragma("unroll") for (int gq = 0; gq < 4; ++gq) b1v[gq] = b1r[gq] * sx; }                    \
        _Pragma("unroll") for (int q = 0; q < 16; ++q) t_[q] = 0.f;                                                   \
        _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) {                                                            \
            const half8 ch_ = *reinterpret_cast<const half8*>(col1 + (((wave + 4 * (B)) * 2 + ks) * 2 + 0) * 1024);  \
            if constexpr (PREC == 0) {                                                                                \
                const half8 cl_ = *reinterpret_cast<const half8*>(col1 + (((wave + 4 * (B)) * 2 + ks) * 2 + 1) * 1024); \
                t_ = mma32<0>(w1l[ks], ch_, t_);                                                                      \
                t_ = mma32<0>(w1h[ks], cl_, t_);                                                                      \
            }                                                                                                         \
            t_ = mma32<PREC>(w1h[ks], ch_, t_);                                                                       \
        }                                                                                                             \
        typedef _Float16 half4_ __attribute__((ext_vector_type(4)));                                                  \
        const bool livep_ = p1[B] >= 0;                                                                               \
        uint8_t* d0_ = livep_ ? (DST) + p1[B] * F3_PITCH + (lane >> 5) * 8 : lds + 2 * F3_BUF;                        \
        const int lo_off_ = livep_ ? F3_PART : 16;                                                                    \
        const int gstep_ = livep_ ? 16 : 0;                                                                           \
        _Pragma("unroll") for (int gq = 0; gq < 4; ++gq) {                                                            \
            float v4_[4];                                                                                             \
            _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                           \
                const float v_ = __builtin_fmaf(t_[4 * gq + i], inv1sx, b1v[gq][i]);   /* sx (2^k) folded in */       \
                v4_[i] = pin1[B] ? fmaxf(v_, 0.01f * v_) : 0.f;                                                       \
            }                                                                                                         \
            if constexpr (PREC == 0) {                                                                                \
                half4_ hi_, lo_;                                                                                      \
                split4v(v4_, hi_, lo_);                                                                               \
                *reinterpret_cast<half4_*>(d0_ + gq * gstep_) = hi_;                                                  \
                *reinterpret_cast<half4_*>(d0_ + gq * gstep_ + lo_off_) = lo_;                                        \
            } else {                                                                                                  \
                typedef float f4_ __attribute__((ext_vector_type(4)));                                                \
                typedef __bf16 b4_ __attribute__((ext_vector_type(4)));                                               \
                const f4_ x_ = {v4_[0], v4_[1], v4_[2], v4_[3]};                                                      \
                if constexpr (PREC == 2) *reinterpret_cast<b4_*>(d0_ + gq * gstep_) = __builtin_convertvector(x_, b4_); \
                else *reinterpret_cast<half4_*>(d0_ + gq * gstep_) = __builtin_convertvector(x_, half4_);             \
            }                                                                                                         \
        }                                                                                                             \
    }

    // ---- prologue: chunk 0 into buffer 0
    if constexpr (FUSED) {
        F3_FUSED_LOAD(0)
        F3_FUSED_BLOCK(0, lds)
        F3_FUSED_BLOCK(1, lds)
        F3_FUSED_BLOCK(2, lds)
    } else {
#pragma unroll
        for (int r = 0; r < F3_R; ++r) F3_TASK_LOAD(r, 0)
#pragma unroll
        for (int r = 0; r < F3_R; ++r) {
            F3_TASK_STORE(r, 0, 0, lds)
            F3_TASK_STORE(r, 1, 0, lds)
        }
    }
    __syncthreads();

    // Per chunk: 36 units = (tap, k-step, half of the 8 pixel rows), 12 MFMAs each (384 matrix cycles).  Everything else
    // is cut into per-unit pieces and interleaved BETWEEN the MFMAs with sched_group_barrier (one wave per SIMD: nothing
    // else hides it): the 8 LDS fragment reads of unit u+1; the weight fragments of step st+5 (ring of 6, straight from
    // L2, continuous across chunks, packed buffer padded by 5 steps); the 48 global loads of the next chunk's patch
    // (units 0-5, 8 each, issued AFTER the unit's weight loads so that a later wait on a weight fragment never drags a
    // younger HBM load along); the split + LDS store of the next chunk (units 24-35, half a task each).  The unit body
    // has no branch (the last chunk re-loads itself and stores into the idle buffer), so each unit is one basic block
    // for the scheduler.  One barrier per chunk.
    if constexpr (S16) {
    // 36 units per chunk = (tap, pair of pixel rows): 4 pixel blocks x 2 channel halves x 3 products = 24 MFMAs (384 matrix
    // cycles, as in the 32x32x16 form), 8 fragment reads for the next unit; the 4 weight fragments of tap + 2 at the first
    // unit of a tap (ring of 3 taps, continuous across chunks; the packed buffer is padded by 5 steps).  The products of one
    // accumulator are 8 MFMAs apart (no back-to-back dependent pair)
    half8 ah[3][2], al[3][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int hv = 0; hv < 2; ++hv) {
            ah[i][hv] = *reinterpret_cast<const half8*>(wbase + (2 * i + hv) * SB);
            if constexpr (PREC == 0) al[i][hv] = *reinterpret_cast<const half8*>(wbase + (2 * i + hv) * SB + 1024);
        }
    F3_STAMP(1)
    for (int chunk = 0; chunk < a.nch; ++chunk) {
        if (chunk < 8) F3_STAMP(2 + chunk)
        const int buf = chunk & 1;
        const uint8_t* wp = wbase + (int64_t)chunk * (18 * SB);
        const uint8_t* bb = lds + buf * F3_BUF + boff;
        uint8_t* sdst = lds + (buf ^ 1) * F3_BUF;
        const int c1 = (chunk + 1 < a.nch ? chunk + 1 : chunk) * F3_CK;     // chunk being staged (last: itself, unused)
        half8 bh[2][4], bl[2][4];
#define F3_BLOAD16(U, SET)                                                                             \
        {                                                                                              \
            const int tap_ = (U) >> 2, rp_ = (U) & 3;                                                  \
            const int dy_ = tap_ / 3, dx_ = tap_ - dy_ * 3;                                            \
            _Pragma("unroll") for (int n = 0; n < 4; ++n) {                                            \
                const int off = ((2 * rp_ + (n >> 1) + dy_) * F3_IW + dx_ + 16 * (n & 1)) * F3_PITCH;  \
                bh[SET][n] = *reinterpret_cast<const half8*>(bb + off);                                \
                if constexpr (!IN16 && PREC == 0) bl[SET][n] = *reinterpret_cast<const half8*>(bb + F3_PART + off); \
            }                                                                                          \
        }
        F3_BLOAD16(0, 0)
#pragma unroll
        for (int u = 0; u < 36; ++u) {
            const int tap = u >> 2, rp = u & 3;
            if (rp == 0) {                                      // weight fragments of tap + 2 (may belong to the next chunk)
#pragma unroll
                for (int hv = 0; hv < 2; ++hv) {
                    ah[(tap + 2) % 3][hv] = *reinterpret_cast<const half8*>(wp + (2 * (tap + 2) + hv) * SB);
                    if constexpr (PREC == 0)
                        al[(tap + 2) % 3][hv] = *reinterpret_cast<const half8*>(wp + (2 * (tap + 2) + hv) * SB + 1024);
                }
            }
            if (u + 1 < 36) F3_BLOAD16(u + 1, (u + 1) & 1)
            if constexpr (FUSED) {
                if (u == 14) F3_FUSED_LOAD(c1)
                if (u == 22) F3_FUSED_BLOCK(0, sdst)
                if (u == 26) F3_FUSED_BLOCK(1, sdst)
                if (u == 30) F3_FUSED_BLOCK(2, sdst)
            } else {
                if (u < F3_R) F3_TASK_LOAD(u, c1)
                if (u >= 24) F3_TASK_STORE((u - 24) >> 1, (u - 24) & 1, c1, sdst)
            }
            if constexpr (PREC == 0) {
#pragma unroll
                for (int n = 0; n < 4; ++n)
#pragma unroll
                    for (int hv = 0; hv < 2; ++hv)
                        acc16[hv][4 * rp + n] = mma16<0>(al[tap % 3][hv], bh[u & 1][n], acc16[hv][4 * rp + n]);
                if constexpr (!IN16) {
#pragma unroll
                    for (int n = 0; n < 4; ++n)
#pragma unroll
                        for (int hv = 0; hv < 2; ++hv)
                            acc16[hv][4 * rp + n] = mma16<0>(ah[tap % 3][hv], bl[u & 1][n], acc16[hv][4 * rp + n]);
                }
            }
#pragma unroll
            for (int n = 0; n < 4; ++n)
#pragma unroll
                for (int hv = 0; hv < 2; ++hv)
                    acc16[hv][4 * rp + n] = mma16<PREC>(ah[tap % 3][hv], bh[u & 1][n], acc16[hv][4 * rp + n]);
#pragma unroll
            for (int i = 0; i < (PREC == 0 ? 24 : 8); ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                        // one MFMA
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                        // one LDS read
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);                        // one global load
                __builtin_amdgcn_sched_group_barrier(0x002, PREC == 0 ? 2 : 5, 0);        // a few vector ALU instructions
                __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);                        // one LDS write
            }
            __builtin_amdgcn_sched_barrier(0);
        }
#undef F3_BLOAD16
        __syncthreads();
    }
    } else {
    half8 ah[6], al[6];
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        ah[i] = *reinterpret_cast<const half8*>(wbase + i * SB);
        if constexpr (PREC == 0) al[i] = *reinterpret_cast<const half8*>(wbase + i * SB + 1024);
    }
    F3_STAMP(1)
    for (int chunk = 0; chunk < a.nch; ++chunk) {
        if (chunk < 8) F3_STAMP(2 + chunk)
        const int buf = chunk & 1;
        const uint8_t* wp = wbase + (int64_t)chunk * (18 * SB);
        const uint8_t* bb = lds + buf * F3_BUF + boff;
        uint8_t* sdst = lds + (buf ^ 1) * F3_BUF;
        const int c1 = (chunk + 1 < a.nch ? chunk + 1 : chunk) * F3_CK;     // chunk being staged (last: itself, unused)
        half8 bh[2][4], bl[2][4];
#define F3_BLOAD(U, SET)                                                                               \
        {                                                                                              \
            const int st_ = (U) >> 1, hf_ = (U) & 1;                                                   \
            const int tap_ = st_ >> 1, ks_ = st_ & 1;                                                  \
            const int dy_ = tap_ / 3, dx_ = tap_ - dy_ * 3;                                            \
            _Pragma("unroll") for (int n = 0; n < 4; ++n) {                                            \
                const int off = ((hf_ * 4 + n + dy_) * F3_IW + dx_) * F3_PITCH + ks_ * 32;             \
                bh[SET][n] = *reinterpret_cast<const half8*>(bb + off);                                \
                if constexpr (!IN16 && PREC == 0) bl[SET][n] = *reinterpret_cast<const half8*>(bb + F3_PART + off); \
            }                                                                                          \
        }
        F3_BLOAD(0, 0)
#pragma unroll
        for (int u = 0; u < 36; ++u) {
            const int st = u >> 1, hf = u & 1;
            if (hf == 0) {                                      // weight fragments of step st+5 (may belong to the next chunk)
                ah[(st + 5) % 6] = *reinterpret_cast<const half8*>(wp + (st + 5) * SB);
                if constexpr (PREC == 0) al[(st + 5) % 6] = *reinterpret_cast<const half8*>(wp + (st + 5) * SB + 1024);
            }
            if (u + 1 < 36) F3_BLOAD(u + 1, (u + 1) & 1)
            if constexpr (FUSED) {
                if (u == 14) F3_FUSED_LOAD(c1)
                if (u == 22) F3_FUSED_BLOCK(0, sdst)
                if (u == 26) F3_FUSED_BLOCK(1, sdst)
                if (u == 30) F3_FUSED_BLOCK(2, sdst)
            } else {
                if (u < F3_R) F3_TASK_LOAD(u, c1)
                if (u >= 24) F3_TASK_STORE((u - 24) >> 1, (u - 24) & 1, c1, sdst)
            }
            const half8 A_h = ah[st % 6], A_l = al[st % 6];
#pragma unroll
            for (int n = 0; n < 4; ++n) {
                if constexpr (PREC == 0) {
                    acc[hf * 4 + n] = mma32<0>(A_l, bh[u & 1][n], acc[hf * 4 + n]);
                    if constexpr (!IN16) acc[hf * 4 + n] = mma32<0>(A_h, bl[u & 1][n], acc[hf * 4 + n]);
                }
                acc[hf * 4 + n] = mma32<PREC>(A_h, bh[u & 1][n], acc[hf * 4 + n]);
            }
#pragma unroll
            for (int i = 0; i < (PREC == 0 ? 12 : 4); ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                        // one MFMA
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                        // one LDS read
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);                        // one global load
                __builtin_amdgcn_sched_group_barrier(0x002, PREC == 0 ? 4 : 10, 0);       // a few vector ALU instructions
                __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);                        // one LDS write
            }
            __builtin_amdgcn_sched_barrier(0);
        }
#undef F3_BLOAD
        __syncthreads();
    }
    }
#undef F3_TASK_LOAD
#undef F3_TASK_STORE
#undef F3_FUSED_BLOCK
#undef F3_FUSED_LOAD

    F3_STAMP(10)
    // ---- epilogue: D col = lane&31 (pixel), row = (q&3) + 8*(q>>2) + 4*(lane>>5) (channel of the wave's 32).
    // 128 stores per lane: a wave-uniform base (scalar registers) + one per-lane offset, the activation resolved outside
    // the loops, no per-element address arithmetic (the first version of this epilogue took a fifth of the kernel's time)
    if constexpr (S16) {
        // D col = lane&15 (pixel of the block), row = q + 4*(lane>>4) (channel of the half).  64 stores per lane and half:
        // wave-uniform base + one per-lane offset, 16 consecutive pixels x 4 channel rows per store instruction
        const int px = lane & 15;
        const int ocw = ocb * F3_OCB + __builtin_amdgcn_readfirstlane(wave) * 32;
        const int och = 4 * (lane >> 4);
        float* ybase = a.y + (z * a.cout + ocw) * hw + (int64_t)y0 * w;                  // wave-uniform
        const unsigned loff = (unsigned)och * (unsigned)hw + (unsigned)(x0 + px);
        const bool full = y0 + F3_TH <= h && x0 + F3_TW <= w;                            // uniform
        const float slope = a.act == LLDWT_ACT_LRELU ? 0.01f : (a.act == LLDWT_ACT_RELU ? 0.f : 1.f);   // none: max(v, v)
#pragma unroll
        for (int hv = 0; hv < 2; ++hv) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int oc = ocw + 16 * hv + och + q;
                if (oc < a.cout) {
                    const float bq = a.bias ? a.bias[plane * a.cout + oc] : 0.f;
                    float* yq = ybase + (int64_t)(16 * hv + q) * hw;
                    if (full && a.act != LLDWT_ACT_TANH) {
#pragma unroll
                        for (int n = 0; n < 16; ++n) {
                            const float v = acc16[hv][n][q] * out_scale + bq;
                            yq[(n >> 1) * w + 16 * (n & 1) + loff] = fmaxf(v, v * slope);
                        }
                    } else {
#pragma unroll
                        for (int n = 0; n < 16; ++n)
                            if (y0 + (n >> 1) < h && x0 + 16 * (n & 1) + px < w)
                                yq[(n >> 1) * w + 16 * (n & 1) + loff] = act_apply(acc16[hv][n][q] * out_scale + bq, a.act);
                    }
                }
            }
        }
    } else {
        const int gx = x0 + (lane & 31);
        const int ocw = ocb * F3_OCB + __builtin_amdgcn_readfirstlane(wave) * 32;       // first channel of this wave
        const int och = 4 * (lane >> 5);
        float* ybase = a.y + (z * a.cout + ocw) * hw + (int64_t)y0 * w;                  // wave-uniform
        const unsigned loff = (unsigned)och * (unsigned)hw + (unsigned)gx;              // this lane's element offset
        float bq[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int oc = ocw + (q & 3) + 8 * (q >> 2) + och;
            bq[q] = a.bias ? a.bias[plane * a.cout + (oc < a.cout ? oc : a.cout - 1)] : 0.f;
        }
        const bool full = y0 + F3_TH <= h && x0 + F3_TW <= w;                            // uniform
        if (full && a.act != LLDWT_ACT_TANH) {
            const float slope = a.act == LLDWT_ACT_LRELU ? 0.01f : (a.act == LLDWT_ACT_RELU ? 0.f : 1.f);   // none: max(v, v)
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                if (ocw + (q & 3) + 8 * (q >> 2) + och < a.cout) {
                    float* yq = ybase + (int64_t)((q & 3) + 8 * (q >> 2)) * hw;
#pragma unroll
                    for (int n = 0; n < 8; ++n) {
                        const float v = acc[n][q] * out_scale + bq[q];
                        yq[n * w + loff] = fmaxf(v, v * slope);
                    }
                }
            }
        } else if (full) {
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                if (ocw + (q & 3) + 8 * (q >> 2) + och < a.cout) {
                    float* yq = ybase + (int64_t)((q & 3) + 8 * (q >> 2)) * hw;
#pragma unroll
                    for (int n = 0; n < 8; ++n) yq[n * w + loff] = fast_tanh(acc[n][q] * out_scale + bq[q]);
                }
            }
        } else {
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                if (ocw + (q & 3) + 8 * (q >> 2) + och < a.cout) {
                    float* yq = ybase + (int64_t)((q & 3) + 8 * (q >> 2)) * hw;
#pragma unroll
                    for (int n = 0; n < 8; ++n)
                        if (y0 + n < h && gx < w) yq[n * w + loff] = act_apply(acc[n][q] * out_scale + bq[q], a.act);
                }
            }
        }
    }
    F3_STAMP(11)
    F3_STAMP(15)
}
#undef F3_STAMP

}  // namespace lldwt
using namespace lldwt;

// MFMA shape of the second conv's main loop, fixed for the process (weights are packed for it).  Default 32x32x16: measured
// 1.80 ms per level-0 launch against 1.91 ms for 16x16x32 (8 x 3 x 256 x 256, f16x3) -- the higher clock the chip holds on the
// small shape does not pay for the halved issue room between two MFMAs (8 free cycles instead of 24 for the fragment reads, the
// weight stream and the first conv's epilogue).  LLDWT_PLC_SHAPE=16 selects the 16x16x32 kernels (kept, tested)
static const int g_f3_shape16 = [] { const char* e = getenv("LLDWT_PLC_SHAPE"); return (e && !strcmp(e, "16")) ? 1 : 0; }();
static int f3_shape16() { return g_f3_shape16; }
extern "C" int lldwt_plc_shape16(void) { return g_f3_shape16; }

extern "C" int64_t lldwt_conv_f16x3_packed_bytes(int cin, int cout) {
    if (cin <= 0 || cout <= 0) return -1;
    return f3_plane_bytes(cin, cout);
}

extern "C" int lldwt_conv_f16x3_pack(const float* w, void* packed, int cin, int cout, int64_t planes, void* stream) {
    LLDWT_REQUIRE(w && packed && cin > 0 && cout > 0 && planes > 0 && planes <= 65535, "conv_f16x3_pack: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    const int64_t pb = f3_plane_bytes(cin, cout);
    for (int64_t p = 0; p < planes; ++p)
        if (hipMemsetAsync(reinterpret_cast<char*>(packed) + p * pb, 0, F3_HDR, st) != hipSuccess) {
            set_error("conv_f16x3_pack: memset failed");
            return LLDWT_EHIP;
        }
    const int64_t nw = (int64_t)cout * cin * 9;
    hipLaunchKernelGGL(k_f3_wmax, dim3(64, (unsigned)planes), dim3(256), 0, st, w, nw, reinterpret_cast<float*>(packed), pb);
    hipLaunchKernelGGL(k_f3_pack, dim3(1024, (unsigned)planes), dim3(256), 0, st, w, reinterpret_cast<uint8_t*>(packed), cin, cout, pb,
                       f3_shape16());
    return check_launch("conv_f16x3_pack");
}

extern "C" int lldwt_absmax_slots(const float* x, int64_t planes, int64_t n_per_plane, float* slots, void* stream) {
    LLDWT_REQUIRE(x && slots && planes > 0 && planes <= 65535 && n_per_plane > 0, "absmax_slots: bad arguments");
    const int vec = (((uintptr_t)x) & 15) == 0 && (n_per_plane & 3) == 0;
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(slots, 0, sizeof(float) * 64 * planes, st) != hipSuccess) {
        set_error("absmax_slots: memset failed");
        return LLDWT_EHIP;
    }
    int64_t gx = cdiv(n_per_plane / 4 + 1, 256 * 8);
    gx = gx < 1 ? 1 : (gx > 2048 ? 2048 : gx);
    hipLaunchKernelGGL(k_absmax_slots, dim3((unsigned)gx, (unsigned)planes), dim3(256), 0, st, x, n_per_plane, slots, vec);
    return check_launch("absmax_slots");
}

extern "C" int lldwt_conv3x3_f16x3(const float* x, float* y, const void* packed, const float* bias, const float* slots,
                                   int cin, int cout, int act, int64_t planes, int64_t batch, int64_t h, int64_t w_,
                                   void* stream) {
    LLDWT_REQUIRE(x && y && packed && slots, "conv3x3_f16x3: null pointer");
    LLDWT_REQUIRE(cin > 0 && cout > 0 && planes > 0 && batch > 0 && h > 0 && w_ > 0 && planes * batch <= 65535,
                  "conv3x3_f16x3: bad dims");
    LLDWT_REQUIRE(act == LLDWT_ACT_NONE || act == LLDWT_ACT_LRELU || act == LLDWT_ACT_TANH || act == LLDWT_ACT_RELU, "conv3x3_f16x3: bad activation");
    LLDWT_REQUIRE((int64_t)F3_CK * h * w_ * 4 < (int64_t)1 << 32, "conv3x3_f16x3: image too large for 32-bit chunk offsets");
    F3Args a;
    a.x = x; a.y = y; a.packed = reinterpret_cast<const uint8_t*>(packed); a.bias = bias; a.slots = slots;
    a.x16 = nullptr; a.xscale = nullptr; a.parent = nullptr; a.packed1 = nullptr; a.plane_bytes1 = 0; a.stamps = nullptr;
    a.cin = cin; a.cout = cout; a.act = act; a.batch = (int)batch; a.h = (int)h; a.w = (int)w_;
    a.tiles_x = (int)cdiv(w_, F3_TW);
    a.nch = f3_nch(cin);
    a.plane_bytes = f3_plane_bytes(cin, cout);
    const int tiles_y = (int)cdiv(h, F3_TH);
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute((const void*)k_conv3_f16x3<0>, hipFuncAttributeMaxDynamicSharedMemorySize, F3_LDS) != hipSuccess ||
            hipFuncSetAttribute((const void*)k_conv3_f16x3<0, 0, true>, hipFuncAttributeMaxDynamicSharedMemorySize, F3_LDS) != hipSuccess ||
            hipFuncSetAttribute((const void*)k_conv3_f16x3<1>, hipFuncAttributeMaxDynamicSharedMemorySize, F3_LDS) != hipSuccess) {
            set_error("conv3x3_f16x3: cannot reserve %d bytes of LDS", F3_LDS);
            return LLDWT_EHIP;
        }
        attr_set = true;
    }
    dim3 grid((unsigned)(a.tiles_x * tiles_y), (unsigned)f3_nocb(cout), (unsigned)(planes * batch));
    if (g_f3_shape16) hipLaunchKernelGGL((k_conv3_f16x3<0, 0, true>), grid, dim3(256), F3_LDS, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(k_conv3_f16x3<0>, grid, dim3(256), F3_LDS, (hipStream_t)stream, a);
    return check_launch("conv3x3_f16x3");
}

// fp16-storage variant: x16 (planes,batch,cin,h,w) fp16 = fp32 activations x xscale[plane] (a power of two), as written by
// lldwt_conv2d_f16out; two MFMA products per k-step.
extern "C" int lldwt_conv3x3_f16in(const void* x16, float* y, const void* packed, const float* bias, const float* xscale,
                                   int cin, int cout, int act, int64_t planes, int64_t batch, int64_t h, int64_t w_,
                                   void* stream) {
    LLDWT_REQUIRE(x16 && y && packed && xscale, "conv3x3_f16in: null pointer");
    LLDWT_REQUIRE(cin > 0 && cout > 0 && planes > 0 && batch > 0 && h > 0 && w_ > 0 && planes * batch <= 65535,
                  "conv3x3_f16in: bad dims");
    LLDWT_REQUIRE(act == LLDWT_ACT_NONE || act == LLDWT_ACT_LRELU || act == LLDWT_ACT_TANH || act == LLDWT_ACT_RELU, "conv3x3_f16in: bad activation");
    LLDWT_REQUIRE((int64_t)F3_CK * h * w_ * 2 < (int64_t)1 << 32, "conv3x3_f16in: image too large for 32-bit chunk offsets");
    F3Args a;
    a.x = nullptr; a.y = y; a.packed = reinterpret_cast<const uint8_t*>(packed); a.bias = bias; a.slots = nullptr;
    a.x16 = reinterpret_cast<const _Float16*>(x16); a.xscale = xscale;
    a.parent = nullptr; a.packed1 = nullptr; a.plane_bytes1 = 0; a.stamps = nullptr;
    a.cin = cin; a.cout = cout; a.act = act; a.batch = (int)batch; a.h = (int)h; a.w = (int)w_;
    a.tiles_x = (int)cdiv(w_, F3_TW);
    a.nch = f3_nch(cin);
    a.plane_bytes = f3_plane_bytes(cin, cout);
    const int tiles_y = (int)cdiv(h, F3_TH);
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute((const void*)k_conv3_f16x3<1>, hipFuncAttributeMaxDynamicSharedMemorySize, F3_LDS) != hipSuccess ||
            hipFuncSetAttribute((const void*)k_conv3_f16x3<1, 0, true>, hipFuncAttributeMaxDynamicSharedMemorySize, F3_LDS) != hipSuccess) {
            set_error("conv3x3_f16in: cannot reserve %d bytes of LDS", F3_LDS);
            return LLDWT_EHIP;
        }
        attr_set = true;
    }
    dim3 grid((unsigned)(a.tiles_x * tiles_y), (unsigned)f3_nocb(cout), (unsigned)(planes * batch));
    if (g_f3_shape16) hipLaunchKernelGGL((k_conv3_f16x3<1, 0, true>), grid, dim3(256), F3_LDS, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(k_conv3_f16x3<1>, grid, dim3(256), F3_LDS, (hipStream_t)stream, a);
    return check_launch("conv3x3_f16in");
}


// ---- the tree-context PAIR in one launch: y = act2(conv3x3(LeakyReLU(conv3x3(up2(parent)) + b1)) + b2)
extern "C" int64_t lldwt_plc_fused_pack1_bytes(int cmid) { return cmid > 0 && cmid <= 256 ? f1_plane_bytes(cmid) : -1; }

extern "C" int lldwt_plc_fused_pack1(const float* w1, const float* b1, void* packed1, int cmid, int64_t planes, void* stream) {
    LLDWT_REQUIRE(w1 && b1 && packed1 && cmid > 0 && cmid <= 256 && planes > 0 && planes <= 65535, "plc_fused_pack1: bad arguments");
    hipLaunchKernelGGL(k_f1_pack, dim3((unsigned)planes), dim3(256), 0, (hipStream_t)stream, w1, b1,
                       reinterpret_cast<uint8_t*>(packed1), cmid, f1_plane_bytes(cmid));
    return check_launch("plc_fused_pack1");
}

static unsigned long long* g_f3_stamps = nullptr;
static int64_t g_f3_stamps_bytes = 0;
namespace lldwt { void f3_set_stamps(void* p, int64_t nbytes) { g_f3_stamps = reinterpret_cast<unsigned long long*>(p); g_f3_stamps_bytes = p ? nbytes : 0; } }

extern "C" int lldwt_plc_fused(const float* parent, float* y, const void* packed1, const void* packed2, const float* bias2,
                               int cmid, int cout, int act, int64_t planes, int64_t batch, int64_t h, int64_t w_, void* stream) {
    LLDWT_REQUIRE(parent && y && packed1 && packed2, "plc_fused: null pointer");
    LLDWT_REQUIRE(cmid > 0 && cmid <= 256 && cout > 0 && planes > 0 && batch > 0 && h > 0 && w_ > 0 && planes * batch <= 65535,
                  "plc_fused: bad dims");
    LLDWT_REQUIRE(h % 2 == 0 && w_ % 2 == 0, "plc_fused: the output is the 2x-upsampled parent's size: even h, w");
    LLDWT_REQUIRE(act == LLDWT_ACT_NONE || act == LLDWT_ACT_LRELU || act == LLDWT_ACT_TANH || act == LLDWT_ACT_RELU, "plc_fused: bad activation");
    F3Args a;
    a.x = nullptr; a.y = y; a.packed = reinterpret_cast<const uint8_t*>(packed2); a.bias = bias2; a.slots = nullptr;
    a.x16 = nullptr; a.xscale = nullptr;
    a.parent = parent; a.packed1 = reinterpret_cast<const uint8_t*>(packed1); a.plane_bytes1 = f1_plane_bytes(cmid);
    a.cin = cmid; a.cout = cout; a.act = act; a.batch = (int)batch; a.h = (int)h; a.w = (int)w_;
    a.tiles_x = (int)cdiv(w_, F3_TW);
    a.nch = f3_nch(cmid);
    a.plane_bytes = f3_plane_bytes(cmid, cout);
    const int tiles_y = (int)cdiv(h, F3_TH);
    {   // diagnostics (tools/plc_stamps.py, lldwt_set_diagnostics): only when the registered buffer holds this grid's stamps
        const int64_t need = (int64_t)a.tiles_x * tiles_y * f3_nocb(cout) * planes * batch * 4 * 16 * 8;
        a.stamps = (g_f3_stamps && g_f3_stamps_bytes >= need) ? g_f3_stamps : nullptr;
    }
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute((const void*)k_conv3_f16x3<2, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, F1_LDS) != hipSuccess ||
            hipFuncSetAttribute((const void*)k_conv3_f16x3<2, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, F1_LDS) != hipSuccess ||
            hipFuncSetAttribute((const void*)k_conv3_f16x3<2, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, F1_LDS) != hipSuccess ||
            hipFuncSetAttribute((const void*)k_conv3_f16x3<2, 0, true>, hipFuncAttributeMaxDynamicSharedMemorySize, F1_LDS) != hipSuccess ||
            hipFuncSetAttribute((const void*)k_conv3_f16x3<2, 1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, F1_LDS) != hipSuccess ||
            hipFuncSetAttribute((const void*)k_conv3_f16x3<2, 2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, F1_LDS) != hipSuccess) {
            set_error("plc_fused: cannot reserve %d bytes of LDS", F1_LDS);
            return LLDWT_EHIP;
        }
        attr_set = true;
    }
    dim3 grid((unsigned)(a.tiles_x * tiles_y), (unsigned)f3_nocb(cout), (unsigned)(planes * batch));
    const int prec = split_precision();
    if (g_f3_shape16) {
        if (prec == 1) hipLaunchKernelGGL((k_conv3_f16x3<2, 1, true>), grid, dim3(256), F1_LDS, (hipStream_t)stream, a);
        else if (prec == 2) hipLaunchKernelGGL((k_conv3_f16x3<2, 2, true>), grid, dim3(256), F1_LDS, (hipStream_t)stream, a);
        else hipLaunchKernelGGL((k_conv3_f16x3<2, 0, true>), grid, dim3(256), F1_LDS, (hipStream_t)stream, a);
    } else if (prec == 1) hipLaunchKernelGGL((k_conv3_f16x3<2, 1>), grid, dim3(256), F1_LDS, (hipStream_t)stream, a);
    else if (prec == 2) hipLaunchKernelGGL((k_conv3_f16x3<2, 2>), grid, dim3(256), F1_LDS, (hipStream_t)stream, a);
    else hipLaunchKernelGGL((k_conv3_f16x3<2, 0>), grid, dim3(256), F1_LDS, (hipStream_t)stream, a);
    return check_launch("plc_fused");
}
