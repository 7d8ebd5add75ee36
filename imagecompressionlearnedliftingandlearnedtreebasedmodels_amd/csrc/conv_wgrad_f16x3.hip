// conv_wgrad_f16x3.hip -- backward-weights of the dense 3x3 tree-context conv (243 -> 243, LiftingBasedDWT_net.py:271-272)
// on the fp16 matrix cores with split-fp16 operands (hi + lo fp16 per fp32 value, power-of-two scaled, hi*hi + hi*lo + lo*hi,
// fp32 accumulate: fp32-level accuracy, see conv_f16x3.hip).  Replaces k_conv_wgrad<3,4,16> (fp32 MFMA, 26 ms of the
// training step at 89 TFLOP/s) for this layer; the generic kernel stays for every other shape.
//
//   dW[oc][ic][ty][tx] += sum over images and pixels of dY[oc][y][x] * X[ic][y + ty - 1][x + tx - 1]       (zero padded)
//   db[oc]             += sum dY[oc]
//
// GEMM view: M = oc, N = (tap, ic), K = pixels.  v_mfma_f32_32x32x16_f16: A = dY (32 oc x 16 pixels), B = X shifted by the
// tap (16 pixels x 32 ic), so an N tile is ONE tap for 32 input channels and its B fragment is "8 consecutive pixels of
// one channel at a fixed shift".  Both operands are indexed by pixel along K, so they live in LDS channel-major:
//   A image [128 oc][2 rows x 32 px] fp16, pitch 144 B          (9 16-byte slots: odd -> conflict-free ds_read_b128)
//   B image [3 tx][32 ic][4 rows][32 px] fp16, row pitch 80 B, channel pitch 336 B (21 slots: odd)
// -- one copy of the input rows per horizontal tap, pre-shifted by tx, because a 16-byte fragment read must be aligned and
// tx moves the start by 2 bytes; the vertical tap is a row offset.
// Workgroup = 4 waves (one per SIMD), tile 128 oc x 32 ic x 9 taps = 36 accumulator tiles, wave w owns oc 32w .. 32w+31 for
// all 9 taps (144 accumulator registers): one A fragment serves 27 MFMAs.  K is cut into chunks of 2 x 32 pixels; the
// next chunk's global loads are in flight during the current chunk's 108 MFMAs per wave; split + LDS store between two
// barriers.  K is also split over workgroups (slices of the chunk sequence), partial tiles are added with float atomics.
// Scales: one power of two per plane for dY and for X (max |.| from lldwt_absmax_slots), so every chunk accumulates at
// the same scale.
#include <cstring>
#include <type_traits>
#include "common.h"
#include "split_f16.h"

namespace lldwt {
namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef _Float16 half2 __attribute__((ext_vector_type(2)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef float floatx16 __attribute__((ext_vector_type(16)));

constexpr int WM = 128, WIC = 32;                 // oc and ic per workgroup
constexpr int CR = 2, CW = 32, CPX = CR * CW;     // K chunk: 2 rows x 32 pixels
constexpr int A_PITCH = CPX * 2 + 16;             // 144 B per oc
constexpr int A_PART = WM * A_PITCH;              // 18,432 B (hi or lo)
constexpr int B_ROW = CW * 2 + 16;                // 80 B per row
constexpr int B_IC = (CR + 2) * B_ROW + 16;       // 336 B per channel
constexpr int B_TX = WIC * B_IC;                  // 10,752 B per horizontal tap
constexpr int B_PART = 3 * B_TX;                  // 32,256 B (hi or lo)
constexpr int LDS_A = 0, LDS_B = 2 * A_PART;
constexpr int LDS_DUMP = LDS_B + 2 * B_PART;      // 64 B that out-of-range pieces of the shifted copies are stored to (no branch)
constexpr int LDS_TOTAL = LDS_DUMP + 64;          // 101,440 B
constexpr int NA4 = WM * CPX / 4 / 256;           // dY float4s per thread and chunk: 8
constexpr int XSEG = (CW + 8) / 4;                // input float4 segments per row: columns x0-4 .. x0+35 -> 10
constexpr int NX4 = WIC * (CR + 2) * XSEG / 256;  // 5
static_assert(WM * CPX / 4 % 256 == 0 && WIC * (CR + 2) * XSEG % 256 == 0, "whole staging rounds");

__device__ __forceinline__ float pow2_scale_for(float amax) {         // s = 2^k with amax * s in [2^13, 2^14)
    if (!(amax > 0.f) || !(amax < 3.0e38f)) return 1.f;
    int e;
    (void)frexpf(amax, &e);
    int k = 14 - e;
    k = k > 120 ? 120 : (k < -120 ? -120 : k);
    return ldexpf(1.f, k);
}

struct WgArgs {
    const float* x;       // (planes, batch, cin, h, w)
    const float* dy;      // (planes, batch, cout, h, w)
    float* dw;            // (planes, cout, cin, 3, 3)
    float* db;            // (planes, cout) or null
    const float* sx;      // (planes, 64) max-|x| slots
    const float* sy;      // (planes, 64) max-|dy| slots
    int cin, cout, batch, h, w, nicb, nocb, slices, chunks_x, chunks_y;
    float alpha;
};

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void k_wgrad3_f16x3(WgArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int plane = blockIdx.z;
    // blockIdx.x = (slice_hi * ncol + column) * 8 + slice_lo: the columns of one K slice sit on one XCD (id % 8), next to
    // each other in its queue, so the chunk of dY / x they all stage is read from HBM once and from that XCD's L2 after
    const int ncol = a.nicb * a.nocb;
    const int s_lo = blockIdx.x & 7, t_ = blockIdx.x >> 3;
    const int column = t_ % ncol, slice = (t_ / ncol) * 8 + s_lo;
    const int icb = column % a.nicb, ocb = column / a.nicb;
    const int ic0 = icb * WIC, oc0 = ocb * WM;
    const int h = a.h, w = a.w;
    const int64_t hw = (int64_t)h * w;
    const int nchunk_img = a.chunks_x * a.chunks_y;
    const int nchunk = a.batch * nchunk_img;

    // ---- scales (powers of two, per plane)
    float ax = a.sx[plane * 64 + lane], ay = a.sy[plane * 64 + lane];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        ax = fmaxf(ax, __shfl_xor(ax, o, 64));
        ay = fmaxf(ay, __shfl_xor(ay, o, 64));
    }
    const float sX = pow2_scale_for(ax), sY = pow2_scale_for(ay);

    floatx16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[t][q] = 0.f;

    // ---- staging roles of this thread (fixed over the chunks)
    // dY: float4 f = tid + 256 j -> oc = f / 16, row = (f % 16) / 8, x4 = 4 (f % 8)
    // x : float4 f = tid + 256 j -> ic = f / 40, row = (f % 40) / 10 (image row y0 - 1 + row), segment s = f % 10 (cols x0-4+4s ..)
    const float* xp = a.x + ((int64_t)plane * a.batch * a.cin + ic0) * hw;
    const float* yp = a.dy + ((int64_t)plane * a.batch * a.cout + oc0) * hw;
    floatx4 ra[NA4], rx[NX4];
    float dbs[NA4];
#pragma unroll
    for (int j = 0; j < NA4; ++j) dbs[j] = 0.f;

    auto issue = [&](int chunk) {                     // global loads of one chunk (clamped addresses, masked at the store)
        const int img = chunk / nchunk_img, rem = chunk - img * nchunk_img;
        const int cy = rem / a.chunks_x, cx = rem - cy * a.chunks_x;
        const int y0 = cy * CR, x0 = cx * CW;
#pragma unroll
        for (int j = 0; j < NA4; ++j) {
            const int f = tid + 256 * j;
            const int oc = f >> 4, row = (f >> 3) & 1, x4 = (f & 7) * 4;
            const int gy = min(y0 + row, h - 1), gx = min(x0 + x4, w - 4), occ = min(oc0 + oc, a.cout - 1) - oc0;
            ra[j] = *reinterpret_cast<const floatx4*>(yp + ((int64_t)img * a.cout + occ) * hw + (int64_t)gy * w + gx);
        }
#pragma unroll
        for (int j = 0; j < NX4; ++j) {
            const int f = tid + 256 * j;
            const int ic = f / ((CR + 2) * XSEG), r2 = f - ic * ((CR + 2) * XSEG), row = r2 / XSEG, s = r2 - row * XSEG;
            const int gy = min(max(y0 - 1 + row, 0), h - 1), gx = min(max(x0 - 4 + 4 * s, 0), w - 4);
            const int icc = min(ic0 + ic, a.cin - 1) - ic0;
            rx[j] = *reinterpret_cast<const floatx4*>(xp + ((int64_t)img * a.cin + icc) * hw + (int64_t)gy * w + gx);
        }
    };
    auto stage = [&](int chunk) {                     // scale, zero what lies outside, split, LDS store
        const int img = chunk / nchunk_img, rem = chunk - img * nchunk_img;
        const int cy = rem / a.chunks_x, cx = rem - cy * a.chunks_x;
        const int y0 = cy * CR, x0 = cx * CW;
        (void)img;
#pragma unroll
        for (int j = 0; j < NA4; ++j) {
            const int f = tid + 256 * j;
            const int oc = f >> 4, row = (f >> 3) & 1, x4 = (f & 7) * 4;
            const bool ok = oc0 + oc < a.cout && y0 + row < h && x0 + x4 < w;
            float v[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = ok ? ra[j][i] : 0.f;
            dbs[j] += (v[0] + v[1]) + (v[2] + v[3]);
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] *= sY;
            half4 hi, lo;
            split4v(v, hi, lo);
            uint8_t* d = lds + LDS_A + oc * A_PITCH + (row * CW + x4) * 2;
            *reinterpret_cast<half4*>(d) = hi;
            *reinterpret_cast<half4*>(d + A_PART) = lo;
        }
#pragma unroll
        for (int j = 0; j < NX4; ++j) {
            const int f = tid + 256 * j;
            const int ic = f / ((CR + 2) * XSEG), r2 = f - ic * ((CR + 2) * XSEG), row = r2 / XSEG, s = r2 - row * XSEG;
            const int gy = y0 - 1 + row, gx = x0 - 4 + 4 * s;
            const bool ok = ic0 + ic < a.cin && gy >= 0 && gy < h && gx >= 0 && gx < w;
            float v[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = ok ? rx[j][i] * sX : 0.f;
            half4 hi, lo;
            split4v(v, hi, lo);
            // copy tx holds element e = (column - x0 + 1 - tx) of the row, e in [0, 32); this segment's columns are x0-4+4s+i
            uint8_t* rowp = lds + LDS_B + ic * B_IC + row * B_ROW;
            // every piece is stored unconditionally: out-of-range pieces go to a dump slot (a branch per piece cost more than
            // the stores)
            uint8_t* dump = lds + LDS_DUMP;
            const int e1 = 4 * s - 4;                 // tx = 1: an aligned group of four
            {
                uint8_t* d = (e1 >= 0 && e1 < CW) ? rowp + B_TX + e1 * 2 : dump;
                *reinterpret_cast<half4*>(d) = hi;
                *reinterpret_cast<half4*>(d + ((e1 >= 0 && e1 < CW) ? B_PART : 8)) = lo;
            }
#pragma unroll
            for (int tx = 0; tx < 3; tx += 2) {       // tx = 0: e = 4s-3+i;  tx = 2: e = 4s-5+i  (odd start: 1 + 2 + 1)
                const int e0 = 4 * s - 3 - tx;
                uint8_t* cp = rowp + tx * B_TX;
                const bool k0 = e0 >= 0 && e0 < CW, k1 = e0 + 1 >= 0 && e0 + 1 < CW, k3 = e0 + 3 >= 0 && e0 + 3 < CW;
                uint8_t* d0 = k0 ? cp + e0 * 2 : dump + 16;
                uint8_t* d1 = k1 ? cp + (e0 + 1) * 2 : dump + 32;
                uint8_t* d3 = k3 ? cp + (e0 + 3) * 2 : dump + 48;
                *reinterpret_cast<_Float16*>(d0) = hi[0];
                *reinterpret_cast<_Float16*>(d0 + (k0 ? B_PART : 2)) = lo[0];
                *reinterpret_cast<half2*>(d1) = half2{hi[1], hi[2]};
                *reinterpret_cast<half2*>(d1 + (k1 ? B_PART : 4)) = half2{lo[1], lo[2]};
                *reinterpret_cast<_Float16*>(d3) = hi[3];
                *reinterpret_cast<_Float16*>(d3 + (k3 ? B_PART : 2)) = lo[3];
            }
        }
    };

    // ---- main loop over this slice's chunks: slice, slice + slices, ...
    const int kg = lane >> 5, l31 = lane & 31;
    const uint8_t* abase = lds + LDS_A + (wave * 32 + l31) * A_PITCH + kg * 16;
    const uint8_t* bbase = lds + LDS_B + l31 * B_IC + kg * 16;
    int chunk = slice;
    if (chunk < nchunk) {
        issue(chunk);
        stage(chunk);
    }
    __syncthreads();
    while (chunk < nchunk) {
        const int next = chunk + a.slices;
        if (next < nchunk) issue(next);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int r = ks >> 1, xh = ks & 1;
            const half8 ah = *reinterpret_cast<const half8*>(abase + (r * CW + 16 * xh) * 2);
            const half8 al = *reinterpret_cast<const half8*>(abase + (r * CW + 16 * xh) * 2 + A_PART);
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int ty = t / 3, tx = t - 3 * ty;
                const uint8_t* bp = bbase + tx * B_TX + (r + ty) * B_ROW + 16 * xh * 2;
                const half8 bh = *reinterpret_cast<const half8*>(bp);
                const half8 bl = *reinterpret_cast<const half8*>(bp + B_PART);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc[t], 0, 0, 0);
            }
        }
        __syncthreads();                              // every wave is done reading this chunk's images
        if (next < nchunk) stage(next);
        __syncthreads();
        chunk = next;
    }

    if (slice >= nchunk) return;                      // nothing staged, nothing to add (uniform: after the last barrier)
    // ---- epilogue: dW[oc][ic][tap] += alpha * acc / (sY sX);  D row = (q&3) + 8 (q>>2) + 4 (lane>>5) (oc), col = lane&31 (ic)
    const float inv = a.alpha * (1.f / sX) * (1.f / sY);
    const int ic = ic0 + l31;
    float* dwp = a.dw + (int64_t)plane * a.cout * a.cin * 9;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const int oc = oc0 + wave * 32 + (q & 3) + 8 * (q >> 2) + 4 * kg;
        if (oc < a.cout && ic < a.cin) {
            float* p = dwp + ((int64_t)oc * a.cin + ic) * 9;
#pragma unroll
            for (int t = 0; t < 9; ++t) atomicAdd(p + t, acc[t][q] * inv);
        }
    }
    // bias gradient: the input-channel block 0 of every oc block adds the dY sums it staged
    if (a.db && icb == 0) {
#pragma unroll
        for (int j = 0; j < NA4; ++j) {
            float s = dbs[j];
#pragma unroll
            for (int o = 8; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);       // the 16 threads that stage one oc row
            const int oc = oc0 + ((tid + 256 * j) >> 4);
            if ((tid & 15) == 0 && oc < a.cout) atomicAdd(a.db + (int64_t)plane * a.cout + oc, s * a.alpha);
        }
    }
}


// ================================================================================================================
// k_wgrad3_f16x3_v2 -- the same GEMM, tile and K split as k_wgrad3_f16x3 with the staging taken OFF the critical path.
// v1 kept three pre-shifted copies of the input rows (one per horizontal tap): 14 scattered 2- / 4-byte LDS stores and a dozen
// address selects per staged float4, all of it between two barriers with the matrix pipe idle (one wave per SIMD) -- the
// chunk loop ran ~10k cycles for 3.5k cycles of MFMAs.  Here:
//   * ONE copy of the input rows, [32 ic][4 rows][40 px] fp16 (columns x0-4 .. x0+35).  A lane reads the two aligned 16-byte
//     windows that hold all three horizontal shifts of its 8-pixel fragment and extracts them in registers: tx = 1 is whole
//     dwords, tx = 0 / 2 are four v_alignbit_b32 each.  Two 8-byte stores per staged float4, 21 KB instead of 63 KB per chunk;
//   * that makes room for TWO LDS buffers (2 x 58 KB): chunk n+1 is split and stored into the other buffer WHILE chunk n is
//     multiplied -- the staging is cut into 13 branch-free pieces (one float4 each) that ride in the shadow of the MFMAs
//     (a 32x32x16 MFMA leaves 24 of its 32 cycles to the vector / LDS issue), the fragment reads of block b+1 and their
//     alignbits sit under block b.  One barrier per chunk;
//   * two register sets for the global loads: chunk n+2 is in flight for a whole chunk period.
constexpr int B2_ROW = 80;                        // 40 elements: columns x0-4 .. x0+35
constexpr int B2_IC = (CR + 2) * B2_ROW + 16;     // 336 B per channel (21 16-byte slots: odd -> conflict-free ds_read_b128)
constexpr int B2_PART = WIC * B2_IC;              // 10,752 B (hi or lo)
constexpr int V2_B = 2 * A_PART;                  // the A image keeps v1's layout
constexpr int V2_BUF = V2_B + 2 * B2_PART;        // 58,368 B per buffer
constexpr int V2_TOTAL = 2 * V2_BUF;              // 116,736 B
static_assert(WIC * (CR + 2) * XSEG == NX4 * 256 && XSEG * 8 == B2_ROW, "five whole staging rounds of the input rows");
// 13 staging pieces (5 of the input rows, then 8 of dY) over the 8 MFMA blocks of a chunk: 1 2 2 1 | 1 2 2 2
__host__ __device__ constexpr int v2_first_piece(int b) {
    return b == 0 ? 0 : b == 1 ? 1 : b == 2 ? 3 : b == 3 ? 5 : b == 4 ? 6 : b == 5 ? 7 : b == 6 ? 9 : b == 7 ? 11 : 13;
}
static_assert(v2_first_piece(8) == NX4 + NA4, "every staging piece has a block");

// hi / lo split in plain C for k_wgrad3_f16x3_v2: the v_fma_mix form of split4v is inline assembly, which the scheduler cannot
// place into the VALU groups of a sched_group_barrier sequence (the staging pieces then cluster behind the MFMAs)
__device__ __forceinline__ void split4c(const float (&v)[4], half4& hi, half4& lo) {
    typedef float f2 __attribute__((ext_vector_type(2)));
    const half2 a = __builtin_convertvector((f2){v[0], v[1]}, half2), b = __builtin_convertvector((f2){v[2], v[3]}, half2);
    hi = __builtin_shufflevector(a, b, 0, 1, 2, 3);
    const half2 c = __builtin_convertvector((f2){v[0] - (float)a[0], v[1] - (float)a[1]}, half2);
    const half2 d = __builtin_convertvector((f2){v[2] - (float)b[0], v[3] - (float)b[1]}, half2);
    lo = __builtin_shufflevector(c, d, 0, 1, 2, 3);
}

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void k_wgrad3_f16x3_v2(WgArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    typedef unsigned uintx4 __attribute__((ext_vector_type(4)));
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int plane = blockIdx.z;
    const int ncol = a.nicb * a.nocb;                 // block order: see k_wgrad3_f16x3
    const int s_lo = blockIdx.x & 7, t_ = blockIdx.x >> 3;
    const int column = t_ % ncol, slice = (t_ / ncol) * 8 + s_lo;
    const int icb = column % a.nicb, ocb = column / a.nicb;
    const int ic0 = icb * WIC, oc0 = ocb * WM;
    const int h = a.h, w = a.w;
    const int hw = h * w;                             // WM * h * w < 2^31 (checked on the host): 32-bit offsets inside an image
    const int nchunk_img = a.chunks_x * a.chunks_y;
    const int nchunk = a.batch * nchunk_img;
    if (slice >= nchunk) return;                      // uniform, before any barrier

    float ax = a.sx[plane * 64 + lane], ay = a.sy[plane * 64 + lane];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        ax = fmaxf(ax, __shfl_xor(ax, o, 64));
        ay = fmaxf(ay, __shfl_xor(ay, o, 64));
    }
    const float sX = pow2_scale_for(ax), sY = pow2_scale_for(ay);

    floatx16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[t][q] = 0.f;

    const float* xp = a.x + ((int64_t)plane * a.batch * a.cin + ic0) * (int64_t)hw;
    const float* yp = a.dy + ((int64_t)plane * a.batch * a.cout + oc0) * (int64_t)hw;
    // ONE register set for the operands in flight: piece j of chunk n+1 is split and stored somewhere inside chunk n's MFMA
    // blocks and its registers are reloaded with piece j of chunk n+2 right there, so every load has a whole chunk period
    floatx4 ra[NA4], rx[NX4];
    float dbs[NA4];
#pragma unroll
    for (int j = 0; j < NA4; ++j) dbs[j] = 0.f;

    struct Pos { const float* xb; const float* yb; int y0, x0; };       // wave-uniform (scalar registers)
    auto coords = [&](int chunk) -> Pos {
        const int img = chunk / nchunk_img, rem = chunk - img * nchunk_img, cy = rem / a.chunks_x;
        return Pos{xp + (int64_t)img * a.cin * hw, yp + (int64_t)img * a.cout * hw, cy * CR, (rem - cy * a.chunks_x) * CW};
    };
    // dY piece j: float4 f = tid + 256 j -> oc = f / 16, row = (f % 16) / 8, x4 = 4 (f % 8);  clamped address, masked at the store
    auto load_a = [&](int j, const Pos& c) {
        const int f = tid + 256 * j;
        const int oc = f >> 4, row = (f >> 3) & 1, x4 = (f & 7) * 4;
        const int gy = min(c.y0 + row, h - 1), gx = min(c.x0 + x4, w - 4), occ = min(oc0 + oc, a.cout - 1) - oc0;
        ra[j] = *reinterpret_cast<const floatx4*>(c.yb + (unsigned)(occ * hw + __mul24(gy, w) + gx));      // h, w < 2^23: full-rate multiply
    };
    // x piece j: f = tid + 256 j -> ic = f / 40, row = (f % 40) / 10 (image row y0 - 1 + row), segment s = f % 10 (cols x0-4+4s ..)
    auto load_x = [&](int j, const Pos& c) {
        const int f = tid + 256 * j;
        const int ic = f / ((CR + 2) * XSEG), r2 = f - ic * ((CR + 2) * XSEG), row = r2 / XSEG, s = r2 - row * XSEG;
        const int gy = min(max(c.y0 - 1 + row, 0), h - 1), gx = min(max(c.x0 - 4 + 4 * s, 0), w - 4);
        const int icc = min(ic0 + ic, a.cin - 1) - ic0;
        rx[j] = *reinterpret_cast<const floatx4*>(c.xb + (unsigned)(icc * hw + __mul24(gy, w) + gx));
    };
    // `live` = the chunk exists (the last chunk of a slice stages a dummy: no branch in the chunk loop)
    auto stage_a = [&](int j, const Pos& c, bool live, uint8_t* buf) {
        const int f = tid + 256 * j;
        const int oc = f >> 4, row = (f >> 3) & 1, x4 = (f & 7) * 4;
        const bool ok = (int)live & (int)(oc0 + oc < a.cout) & (int)(c.y0 + row < h) & (int)(c.x0 + x4 < w);     // '&': no branch
        float v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = ok ? ra[j][i] : 0.f;
        dbs[j] += (v[0] + v[1]) + (v[2] + v[3]);
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] *= sY;
        half4 hi, lo;
        split4c(v, hi, lo);
        uint8_t* d = buf + oc * A_PITCH + (row * CW + x4) * 2;
        *reinterpret_cast<half4*>(d) = hi;
        *reinterpret_cast<half4*>(d + A_PART) = lo;
    };
    auto stage_x = [&](int j, const Pos& c, uint8_t* buf) {
        const int f = tid + 256 * j;
        const int ic = f / ((CR + 2) * XSEG), r2 = f - ic * ((CR + 2) * XSEG), row = r2 / XSEG, s = r2 - row * XSEG;
        const int gy = c.y0 - 1 + row, gx = c.x0 - 4 + 4 * s;         // w % 4 == 0: a float4 lies inside the row or outside
        const bool ok = (int)(ic0 + ic < a.cin) & (int)((unsigned)gy < (unsigned)h) & (int)((unsigned)gx < (unsigned)w);   // '&': no branch
        float v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = ok ? rx[j][i] * sX : 0.f;
        half4 hi, lo;
        split4c(v, hi, lo);
        uint8_t* d = buf + V2_B + ic * B2_IC + row * B2_ROW + s * 8;
        *reinterpret_cast<half4*>(d) = hi;
        *reinterpret_cast<half4*>(d + B2_PART) = lo;
    };

    const int kg = lane >> 5, l31 = lane & 31;
    // the three horizontal shifts of a lane's fragment from its two aligned windows (elements 8m .. 8m+15 of the row, m = 2 xh + kg;
    // output pixel j of the fragment with tap tx reads element 8m + 3 + tx + j)
    auto shifted = [&](const uintx4& w0, const uintx4& w1, int tx) -> half8 {
        const unsigned d[8] = {w0[0], w0[1], w0[2], w0[3], w1[0], w1[1], w1[2], w1[3]};
        uintx4 o;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            o[i] = tx == 1 ? d[i + 2] : tx == 0 ? __builtin_amdgcn_alignbit(d[i + 2], d[i + 1], 16) : __builtin_amdgcn_alignbit(d[i + 3], d[i + 2], 16);
        return __builtin_bit_cast(half8, o);
    };

    // ---- chunks slice, slice + slices, ...; LDS buffer `cur` is multiplied while the other one is filled
    int chunk = slice;
    {
        const Pos c = coords(chunk);
#pragma unroll
        for (int j = 0; j < NX4; ++j) load_x(j, c);
#pragma unroll
        for (int j = 0; j < NA4; ++j) load_a(j, c);
#pragma unroll
        for (int j = 0; j < NX4; ++j) stage_x(j, c, lds);
#pragma unroll
        for (int j = 0; j < NA4; ++j) stage_a(j, c, true, lds);
        const Pos c1 = coords(min(chunk + a.slices, nchunk - 1));
#pragma unroll
        for (int j = 0; j < NX4; ++j) load_x(j, c1);
#pragma unroll
        for (int j = 0; j < NA4; ++j) load_a(j, c1);
    }
    __syncthreads();
    int cur = 0;
    // positions of the chunk staged during the current one (cn; its operands are in the registers) and of the chunk loaded during it
    // (c2); the next iteration's c2 is worked out in the middle of this one (scalar divisions: ~90 instructions that sat at the top of
    // every chunk with the matrix pipe empty)
    Pos cn = coords(min(chunk + a.slices, nchunk - 1)), c2 = coords(min(chunk + 2 * a.slices, nchunk - 1)), c3 = c2;
#pragma unroll 1
    while (true) {
        const int nxt = chunk + a.slices;
        const bool live = nxt < nchunk;
        const uint8_t* cb = lds + cur * V2_BUF;
        uint8_t* sb = lds + (cur ^ 1) * V2_BUF;
        const uint8_t* ab = cb + (wave * 32 + l31) * A_PITCH + kg * 16;
        const uint8_t* bb = cb + V2_B + l31 * B2_IC + kg * 16;
        half8 fah[2][2], fal[2][2];                   // A fragments [xh][chunk row]
#pragma unroll
        for (int xh = 0; xh < 2; ++xh)
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                fah[xh][r] = *reinterpret_cast<const half8*>(ab + (r * CW + 16 * xh) * 2);
                fal[xh][r] = *reinterpret_cast<const half8*>(ab + (r * CW + 16 * xh) * 2 + A_PART);
            }
        half8 bh[2][3], bl[2][3];                     // B fragments [set][tx]
        uintx4 wh[2][2], wl[2][2];                    // the aligned windows they are cut from [set][window]
        // two-deep: block b issues the window reads of block b+2 and cuts the fragments of block b+1 from the windows read during
        // block b-1 (fragments cut right behind their reads left the wave parked for an LDS round trip in every block)
        auto windows = [&](int blk, int set) {
            const int xh = blk >> 2, row = blk & 3;
            const uint8_t* p = bb + row * B2_ROW + xh * 32;
            wh[set][0] = *reinterpret_cast<const uintx4*>(p);
            wh[set][1] = *reinterpret_cast<const uintx4*>(p + 16);
            wl[set][0] = *reinterpret_cast<const uintx4*>(p + B2_PART);
            wl[set][1] = *reinterpret_cast<const uintx4*>(p + B2_PART + 16);
        };
        auto cut = [&](int set) {
#pragma unroll
            for (int tx = 0; tx < 3; ++tx) {
                bh[set][tx] = shifted(wh[set][0], wh[set][1], tx);
                bl[set][tx] = shifted(wl[set][0], wl[set][1], tx);
            }
        };
        windows(0, 0);
        windows(1, 1);
        cut(0);
        __builtin_amdgcn_sched_barrier(0);
        // the 108 MFMAs in 8 blocks (xh, patch row); block b carries the fragment reads + alignbits of block b+1 and its share of the
        // 13 staging pieces, each followed by the reload of its registers
        auto block = [&](auto BLK) {
            constexpr int blk = decltype(BLK)::value;
            constexpr int xh = blk >> 2, row = blk & 3, set = blk & 1;
            constexpr int nm = (row == 0 || row == 3) ? 9 : 18;          // patch row 0 / 3 meets one chunk row, 1 / 2 meet both
            constexpr int p0 = v2_first_piece(blk), p1 = v2_first_piece(blk + 1), npc = p1 - p0;
            if constexpr (blk + 1 < 8) cut(set ^ 1);                      // windows of block b+1: read during block b-1
            if constexpr (blk + 2 < 8) windows(blk + 2, set);            // (its own windows are cut: the set is free)
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const int ty = row - r;
                if (ty < 0 || ty > 2) continue;
#pragma unroll
                for (int tx = 0; tx < 3; ++tx) {
                    const int t = ty * 3 + tx;
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fal[xh][r], bh[set][tx], acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fah[xh][r], bl[set][tx], acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fah[xh][r], bh[set][tx], acc[t], 0, 0, 0);
                }
            }
#pragma unroll
            for (int pc = p0; pc < p1; ++pc) {
                if (pc < NX4) {
                    stage_x(pc, cn, sb);
                    load_x(pc, c2);
                } else {
                    stage_a(pc - NX4, cn, live, sb);
                    load_a(pc - NX4, c2);
                }
            }
            if constexpr (blk == 4) c3 = coords(min(nxt + 2 * a.slices, nchunk - 1));
            // issue order inside the block: an MFMA, then a share of the block's vector / LDS / memory work
#pragma unroll
            for (int i = 0; i < nm; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                if (blk + 2 < 8 && i < 4) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, nm == 9 ? (npc == 2 ? 9 : 7) : 6, 0);
                if (blk == 4) __builtin_amdgcn_sched_group_barrier(0x004, 5, 0);        // the scalar chain of c3
                if (i >= nm - 2 * npc) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
                if (i >= nm - npc) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        };
        block(std::integral_constant<int, 0>{});
        block(std::integral_constant<int, 1>{});
        block(std::integral_constant<int, 2>{});
        block(std::integral_constant<int, 3>{});
        block(std::integral_constant<int, 4>{});
        block(std::integral_constant<int, 5>{});
        block(std::integral_constant<int, 6>{});
        block(std::integral_constant<int, 7>{});
        __syncthreads();
        chunk = nxt;
        cur ^= 1;
        cn = c2;
        c2 = c3;
        if (!live) break;
    }

    // ---- epilogue: as k_wgrad3_f16x3
    const float inv = a.alpha * (1.f / sX) * (1.f / sY);
    const int ic = ic0 + l31;
    float* dwp = a.dw + (int64_t)plane * a.cout * a.cin * 9;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const int oc = oc0 + wave * 32 + (q & 3) + 8 * (q >> 2) + 4 * kg;
        if (oc < a.cout && ic < a.cin) {
            float* p = dwp + ((int64_t)oc * a.cin + ic) * 9;
#pragma unroll
            for (int t = 0; t < 9; ++t) atomicAdd(p + t, acc[t][q] * inv);
        }
    }
    if (a.db && icb == 0) {
#pragma unroll
        for (int j = 0; j < NA4; ++j) {
            float s = dbs[j];
#pragma unroll
            for (int o = 8; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
            const int oc = oc0 + ((tid + 256 * j) >> 4);
            if ((tid & 15) == 0 && oc < a.cout) atomicAdd(a.db + (int64_t)plane * a.cout + oc, s * a.alpha);
        }
    }
}


// ================================================================================================================
// k_wgrad16_f16x3 -- backward-weights of the 16 -> 16 5x5 convs of a P/U block (conv2 and conv3 of P_block_v2.py:40-55; their
// inputs t1 / t2 are tanh outputs, |x| <= 1) on the fp16 matrix cores with split-fp16 operands.  Replaces k_wgrad16<5> (fp32
// MFMA 16x16x4: 24 ms of the headline training step at 38 TFLOP/s).
//
//   dW[oc][ic][ty][tx] += alpha * sum over images and pixels of dY[oc][y][x] * X[ic][y + ty - 2][x + tx - 2]    (zero padded)
//
// GEMM view per tap: M = 16 oc, N = 16 ic, K = pixels; v_mfma_f32_16x16x32_f16 with A = dY (16 oc x 32 pixels of one row) and
// B = X shifted by the tap (32 pixels x 16 ic).  Both fragments are "8 consecutive pixels of one channel", so both operands
// live in LDS channel-major as fp16 rows.  The horizontal tap moves the start of a B fragment by tx elements = 2 tx bytes, and a
// 16-byte LDS read must be aligned: instead of one pre-shifted copy of the patch per tx (five copies: the split + store of the
// patch would cost more than the MFMAs), every lane reads the ALIGNED 16-element window that holds all five shifts of its
// fragment (two ds_read_b128 per row and part, shared by the five tx) and extracts a shift in registers -- even shifts are
// whole dwords, odd shifts four v_alignbit_b32.
// Workgroup = 4 waves; K chunk = 8 rows x 32 pixels; wave w owns rows w and w + 4 of the chunk for all 25 taps (100 accumulator
// registers; 150 MFMAs per chunk and wave against 44 fragment reads).  No register prefetch (the accumulators leave no room
// for it): load -> split -> LDS store of the next chunk happen between two barriers, and the OTHER workgroup resident on the
// CU (two at 255 registers per lane) runs its MFMAs meanwhile.  K is also split over workgroups (one resident round); the partial 16 x 16 x 25
// tiles of the four waves are summed in LDS (the waves take turns: LDS float atomics cost 50-65 us per launch) and added to dW with
// one coalesced float atomic per element.  Measured with phases masked at 3 x 8 x 256 x 512 (356 us with the |max| pass): the MFMA
// phase is 41 us of it, the LDS reduction was 65, load + split + store of the operands 250 -- 50 KB per chunk in row pieces of
// 128-176 B, everyone loading at once between two barriers; one workgroup per CU with the next chunk prefetched in registers
// (433 of them) was no faster (357 us) -- the chunk shape, not the overlap, bounds it.
// Scales: X has a fixed 2^14 (tanh outputs); dY one power of two per plane from its |max| (lldwt_absmax_slots).
constexpr int G_CR = 8, G_CW = 32;                          // chunk: 8 rows x 32 pixels
constexpr int G_AP = G_CR * G_CW * 2 + 16;                  // dY bytes per oc: 528 (132 dwords = 4 mod 64: conflict-free b128)
constexpr int G_APART = 16 * G_AP;                          // 8 448 B (hi or lo)
constexpr int G_XR = G_CR + 4, G_BR = 80;                   // 12 patch rows; 40 elements per row: columns x0-2 .. x0+37
constexpr int G_BC = G_XR * G_BR + 16;                      // bytes per ic: 976 (244 dwords = 52 mod 64: conflict-free b128)
constexpr int G_BPART = 16 * G_BC;                          // 15 616 B (hi or lo)
constexpr int G_LDS_A = 0, G_LDS_B = 2 * G_APART, G_LDS_DUMP = G_LDS_B + 2 * G_BPART;
constexpr int G_LDS_TOTAL = G_LDS_DUMP + 64;                // 48 192 B
constexpr int G_NA4 = 16 * G_CR * G_CW / 4 / 256;           // dY float4s per thread and chunk: 4
constexpr int G_XSEG = 11;                                  // aligned float4 segments per patch row: columns x0-4 .. x0+39
constexpr int G_NX4 = (16 * G_XR * G_XSEG + 255) / 256;     // 9 (the last round: 64 threads)
static_assert(16 * 16 * 25 * 4 <= G_LDS_DUMP, "the epilogue's dW tile fits the staging images");

struct Wg16Args {
    const float* x;       // (planes, batch, 16, h, w), |x| <= 1
    const float* dy;      // (planes, batch, 16, h, w)
    float* dw;            // (planes, 16, 16, 5, 5)
    float* db;            // (planes, 16) or null
    const float* sy;      // max-|dy| slots: 64 per plane, plane stride sy_stride
    int batch, h, w, slices, chunks_x, chunks_y, sy_stride;
    float alpha;
    int8_t tap_of[25];    // tap (ty * 5 + tx) -> position inside a dW[oc][ic] block (row passes store (kh, kw) swapped)
};

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_wgrad16_f16x3(Wg16Args a) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    typedef unsigned uintx4 __attribute__((ext_vector_type(4)));
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // workgroups go to the 8 XCDs round-robin: with a multiple of 8 slices per plane, workgroup x runs on XCD x % 8.  The chunks of a
    // step are dealt so that one XCD gets a CONTIGUOUS eighth of them (horizontal and vertical neighbours): the halo columns / rows
    // two neighbouring chunks both read (x is fetched 2.06 times otherwise) then meet in that XCD's L2
    const int plane = blockIdx.z;
    const int slice = (a.slices & 7) == 0 ? ((int)blockIdx.x & 7) * (a.slices >> 3) + ((int)blockIdx.x >> 3) : (int)blockIdx.x;
    const int h = a.h, w = a.w;
    const int64_t hw = (int64_t)h * w;
    const int nchunk_img = a.chunks_x * a.chunks_y;
    const int nchunk = a.batch * nchunk_img;
    float ay = a.sy[plane * a.sy_stride + lane];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) ay = fmaxf(ay, __shfl_xor(ay, o, 64));
    const float sY = pow2_scale_for(ay);
    constexpr float sX = 16384.f;

    floatx4 acc[25];
#pragma unroll
    for (int t = 0; t < 25; ++t) acc[t] = floatx4{0.f, 0.f, 0.f, 0.f};
    const float* xp = a.x + (int64_t)plane * a.batch * 16 * hw;
    const float* yp = a.dy + (int64_t)plane * a.batch * 16 * hw;
    floatx4 ra[G_NA4], rx[G_NX4];
    float dbs[G_NA4];
#pragma unroll
    for (int j = 0; j < G_NA4; ++j) dbs[j] = 0.f;

    // dY: float4 f = tid + 256 j -> oc = f / 64, row = (f % 64) / 8, x4 = 4 (f % 8)
    // x : float4 f = tid + 256 j -> ic = f / 132, row = (f % 132) / 11 (image row y0 - 2 + row), segment s = f % 11 (columns x0-4+4s ..)
    auto issue = [&](int chunk) {
        const int img = chunk / nchunk_img, rem = chunk - img * nchunk_img;
        const int cy = rem / a.chunks_x, cx = rem - cy * a.chunks_x;
        const int y0 = cy * G_CR, x0 = cx * G_CW;
#pragma unroll
        for (int j = 0; j < G_NA4; ++j) {
            const int f = tid + 256 * j;
            const int oc = f >> 6, row = (f >> 3) & 7, x4 = (f & 7) * 4;
            const int gy = min(y0 + row, h - 1), gx = min(x0 + x4, w - 4);
            ra[j] = *reinterpret_cast<const floatx4*>(yp + ((int64_t)img * 16 + oc) * hw + (int64_t)gy * w + gx);
        }
#pragma unroll
        for (int j = 0; j < G_NX4; ++j) {
            const int f = min(tid + 256 * j, 16 * G_XR * G_XSEG - 1);
            const int ic = f / (G_XR * G_XSEG), r2 = f - ic * (G_XR * G_XSEG), row = r2 / G_XSEG, s = r2 - row * G_XSEG;
            const int gy = min(max(y0 - 2 + row, 0), h - 1), gx = min(max(x0 - 4 + 4 * s, 0), w - 4);
            rx[j] = *reinterpret_cast<const floatx4*>(xp + ((int64_t)img * 16 + ic) * hw + (int64_t)gy * w + gx);
        }
    };
    auto stage = [&](int chunk) {
        const int img = chunk / nchunk_img, rem = chunk - img * nchunk_img;
        const int cy = rem / a.chunks_x, cx = rem - cy * a.chunks_x;
        const int y0 = cy * G_CR, x0 = cx * G_CW;
        (void)img;
#pragma unroll
        for (int j = 0; j < G_NA4; ++j) {
            const int f = tid + 256 * j;
            const int oc = f >> 6, row = (f >> 3) & 7, x4 = (f & 7) * 4;
            const bool ok = y0 + row < h && x0 + x4 < w;
            float v[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = ok ? ra[j][i] : 0.f;
            dbs[j] += (v[0] + v[1]) + (v[2] + v[3]);
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] *= sY;
            half4 hi, lo;
            split4v(v, hi, lo);
            uint8_t* d = lds + G_LDS_A + oc * G_AP + (row * G_CW + x4) * 2;
            *reinterpret_cast<half4*>(d) = hi;
            *reinterpret_cast<half4*>(d + G_APART) = lo;
        }
#pragma unroll
        for (int j = 0; j < G_NX4; ++j) {
            const int f = tid + 256 * j;
            const bool live = f < 16 * G_XR * G_XSEG;
            const int fc = live ? f : 0;
            const int ic = fc / (G_XR * G_XSEG), r2 = fc - ic * (G_XR * G_XSEG), row = r2 / G_XSEG, s = r2 - row * G_XSEG;
            const int gy = y0 - 2 + row, gx = x0 - 4 + 4 * s;
            const bool ok = gy >= 0 && gy < h && gx >= 0 && gx < w;       // w % 4 == 0: a segment is inside or outside as a whole
            float v[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = ok ? rx[j][i] * sX : 0.f;
            half4 hi, lo;
            split4v(v, hi, lo);
            // element e of a patch row holds column x0 - 2 + e: this segment's columns are elements 4s-2 .. 4s+1 = two aligned
            // fp16 pairs; pairs that fall outside the 48 slots (s == 0: elements -2, -1) and dead tasks go to a dump slot
            uint8_t* rowp = lds + G_LDS_B + ic * G_BC + row * G_BR;
            uint8_t* dump = lds + G_LDS_DUMP;
            const int e0 = 4 * s - 2;
            const bool k0 = live && e0 >= 0, k1 = live && e0 + 2 < 40;
            uint8_t* d0 = k0 ? rowp + e0 * 2 : dump;
            uint8_t* d1 = k1 ? rowp + (e0 + 2) * 2 : dump + 8;
            *reinterpret_cast<half2*>(d0) = half2{hi[0], hi[1]};
            *reinterpret_cast<half2*>(d0 + (k0 ? G_BPART : 4)) = half2{lo[0], lo[1]};
            *reinterpret_cast<half2*>(d1) = half2{hi[2], hi[3]};
            *reinterpret_cast<half2*>(d1 + (k1 ? G_BPART : 4)) = half2{lo[2], lo[3]};
        }
    };

    const int kg = lane >> 4, l15 = lane & 15;
    const uint8_t* abase = lds + G_LDS_A + l15 * G_AP + kg * 16;
    const uint8_t* bbase = lds + G_LDS_B + l15 * G_BC + kg * 16;
    int chunk = slice;
    if (chunk < nchunk) {
        issue(chunk);
        stage(chunk);
    }
    __syncthreads();
    while (chunk < nchunk) {
        const int next = chunk + a.slices;
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) {
            const int r = wave + 4 * rr;                                  // chunk row of this k-step
            const half8 ah = *reinterpret_cast<const half8*>(abase + r * G_CW * 2);
            const half8 al = *reinterpret_cast<const half8*>(abase + r * G_CW * 2 + G_APART);
#pragma unroll
            for (int ty = 0; ty < 5; ++ty) {
                const uint8_t* bp = bbase + (r + ty) * G_BR;
                uintx4 wh[2], wl[2];
                wh[0] = *reinterpret_cast<const uintx4*>(bp);
                wh[1] = *reinterpret_cast<const uintx4*>(bp + 16);
                wl[0] = *reinterpret_cast<const uintx4*>(bp + G_BPART);
                wl[1] = *reinterpret_cast<const uintx4*>(bp + G_BPART + 16);
                const unsigned dh[8] = {wh[0][0], wh[0][1], wh[0][2], wh[0][3], wh[1][0], wh[1][1], wh[1][2], wh[1][3]};
                const unsigned dl[8] = {wl[0][0], wl[0][1], wl[0][2], wl[0][3], wl[1][0], wl[1][1], wl[1][2], wl[1][3]};
#pragma unroll
                for (int tx = 0; tx < 5; ++tx) {
                    uintx4 fh, fl;
                    if (tx % 2 == 0) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) { fh[i] = dh[tx / 2 + i]; fl[i] = dl[tx / 2 + i]; }
                    } else {
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            fh[i] = __builtin_amdgcn_alignbit(dh[tx / 2 + i + 1], dh[tx / 2 + i], 16);
                            fl[i] = __builtin_amdgcn_alignbit(dl[tx / 2 + i + 1], dl[tx / 2 + i], 16);
                        }
                    }
                    const half8 bh = __builtin_bit_cast(half8, fh), bl = __builtin_bit_cast(half8, fl);
                    const int t = ty * 5 + tx;
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh, acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl, acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, acc[t], 0, 0, 0);
                }
            }
        }
        __syncthreads();                              // every wave is done reading this chunk's images
        if (next < nchunk) {
            issue(next);
            stage(next);
        }
        __syncthreads();
        chunk = next;
    }

    // ---- epilogue: the four waves' partial tiles summed in LDS in dW order, then one coalesced atomic per element (every
    // workgroup of a plane adds to the same 6 400 addresses: scattered lanes would touch 64 cache lines per instruction)
    // (LDS float atomics from the four waves took 50-65 us per launch -- about one lane per 4-5 clocks; the waves take turns instead:
    // wave 0 stores its tile, each following wave adds its own with plain read-modify-writes, a barrier between two turns)
    float* tile = reinterpret_cast<float*>(lds);
    for (int turn = 0; turn < 4; ++turn) {
        if (wave == turn && slice < nchunk) {
#pragma unroll
            for (int t = 0; t < 25; ++t) {
                const int tap = a.tap_of[t];
#pragma unroll
                for (int q = 0; q < 4; ++q) {         // D row = oc = 4 kg + q, col = ic = lane & 15
                    float* d = tile + ((4 * kg + q) * 16 + l15) * 25 + tap;
                    *d = turn == 0 ? acc[t][q] : *d + acc[t][q];
                }
            }
        }
        __syncthreads();
    }
    if (slice >= nchunk) return;
    const float inv = a.alpha * (1.f / sX) * (1.f / sY);
    float* dwp = a.dw + (int64_t)plane * 16 * 16 * 25;
    for (int i = tid; i < 16 * 16 * 25; i += 256) atomicAdd(dwp + i, tile[i] * inv);
    if (a.db) {
#pragma unroll
        for (int j = 0; j < G_NA4; ++j) {
            float s_ = dbs[j];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) s_ += __shfl_xor(s_, o, 64);   // the 64 threads (one wave) that stage one oc
            const int oc = (tid + 256 * j) >> 6;
            if (lane == 0) atomicAdd(a.db + (int64_t)plane * 16 + oc, s_ * a.alpha);
        }
    }
}


// ================================================================================================================
// k_wgrad16_f16x3_v2 -- the same GEMM, chunk and LDS images as k_wgrad16_f16x3 with the recipe of k_wgrad3_f16x3_v2: TWO LDS buffers
// (2 x 48 KB, one workgroup per CU, one wave per SIMD), the 13 float4 pieces of chunk n+1 split and stored into the other buffer in
// the shadow of chunk n's 150 MFMAs, each piece's registers reloaded right behind it with chunk n+2 (a whole chunk period in flight:
// v1 issued a chunk's loads and consumed them between the same two barriers, so every chunk paid an HBM round trip that only the
// second workgroup of the CU could hide), window reads two blocks ahead, one barrier per chunk.  10 blocks (chunk row of the wave,
// vertical tap) of 15 MFMAs; the 16 x 16 x 32 MFMA leaves two vector issue slots per MFMA, so this kernel is bound by its vector
// work (fragment cuts + staging, ~850 instructions per chunk) and, behind that, by the operand bytes (50 KB per chunk).
constexpr int G2_BUF = G_LDS_DUMP;                           // 48 128 B per buffer (dY hi | lo, x hi | lo)
constexpr int G2_DUMP = 2 * G2_BUF;
constexpr int G2_TOTAL = G2_DUMP + 64;                       // 96 320 B
// 13 staging pieces (9 of x, then 4 of dY) over the 10 MFMA blocks: 1 1 2 1 1 2 1 1 2 1
__host__ __device__ constexpr int g2_first_piece(int b) { return b + b / 3; }
static_assert(g2_first_piece(0) == 0 && g2_first_piece(3) == 4 && g2_first_piece(9) == 12 && g2_first_piece(10) == 13, "13 pieces over 10 blocks");

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void k_wgrad16_f16x3_v2(Wg16Args a) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    typedef unsigned uintx4 __attribute__((ext_vector_type(4)));
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int plane = blockIdx.z;
    const int slice = (a.slices & 7) == 0 ? ((int)blockIdx.x & 7) * (a.slices >> 3) + ((int)blockIdx.x >> 3) : (int)blockIdx.x;   // see v1
    const int h = a.h, w = a.w;
    const int hw = h * w;                             // 16 * h * w < 2^31 (checked on the host): 32-bit offsets inside an image
    const int nchunk_img = a.chunks_x * a.chunks_y;
    const int nchunk = a.batch * nchunk_img;
    if (slice >= nchunk) return;                      // uniform, before any barrier
    float ay = a.sy[plane * a.sy_stride + lane];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) ay = fmaxf(ay, __shfl_xor(ay, o, 64));
    const float sY = pow2_scale_for(ay);
    constexpr float sX = 16384.f;

    floatx4 acc[25];
#pragma unroll
    for (int t = 0; t < 25; ++t) acc[t] = floatx4{0.f, 0.f, 0.f, 0.f};
    const float* xp = a.x + (int64_t)plane * a.batch * 16 * hw;
    const float* yp = a.dy + (int64_t)plane * a.batch * 16 * hw;
    floatx4 ra[G_NA4], rx[G_NX4];
    float dbs[G_NA4];
#pragma unroll
    for (int j = 0; j < G_NA4; ++j) dbs[j] = 0.f;

    struct Pos { const float* xb; const float* yb; int y0, x0; };       // wave-uniform
    auto coords = [&](int chunk) -> Pos {
        const int img = chunk / nchunk_img, rem = chunk - img * nchunk_img, cy = rem / a.chunks_x;
        return Pos{xp + (int64_t)img * 16 * hw, yp + (int64_t)img * 16 * hw, cy * G_CR, (rem - cy * a.chunks_x) * G_CW};
    };
    // dY piece j: float4 f = tid + 256 j -> oc = f / 64, row = (f % 64) / 8, x4 = 4 (f % 8)
    auto load_a = [&](int j, const Pos& c) {
        const int f = tid + 256 * j;
        const int oc = f >> 6, row = (f >> 3) & 7, x4 = (f & 7) * 4;
        const int gy = min(c.y0 + row, h - 1), gx = min(c.x0 + x4, w - 4);
        ra[j] = *reinterpret_cast<const floatx4*>(c.yb + (unsigned)(oc * hw + __mul24(gy, w) + gx));
    };
    // x piece j: f = tid + 256 j -> ic = f / 132, row = (f % 132) / 11 (image row y0 - 2 + row), segment s = f % 11 (columns x0-4+4s ..)
    auto load_x = [&](int j, const Pos& c) {
        const int f = min(tid + 256 * j, 16 * G_XR * G_XSEG - 1);
        const int ic = f / (G_XR * G_XSEG), r2 = f - ic * (G_XR * G_XSEG), row = r2 / G_XSEG, s = r2 - row * G_XSEG;
        const int gy = min(max(c.y0 - 2 + row, 0), h - 1), gx = min(max(c.x0 - 4 + 4 * s, 0), w - 4);
        rx[j] = *reinterpret_cast<const floatx4*>(c.xb + (unsigned)(ic * hw + __mul24(gy, w) + gx));
    };
    auto stage_a = [&](int j, const Pos& c, bool livec, uint8_t* buf) {
        const int f = tid + 256 * j;
        const int oc = f >> 6, row = (f >> 3) & 7, x4 = (f & 7) * 4;
        const bool ok = (int)livec & (int)(c.y0 + row < h) & (int)(c.x0 + x4 < w);
        float v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = ok ? ra[j][i] : 0.f;
        dbs[j] += (v[0] + v[1]) + (v[2] + v[3]);
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] *= sY;
        half4 hi, lo;
        split4c(v, hi, lo);
        uint8_t* d = buf + G_LDS_A + oc * G_AP + (row * G_CW + x4) * 2;
        *reinterpret_cast<half4*>(d) = hi;
        *reinterpret_cast<half4*>(d + G_APART) = lo;
    };
    auto stage_x = [&](int j, const Pos& c, uint8_t* buf) {
        const int f = tid + 256 * j;
        const bool live = f < 16 * G_XR * G_XSEG;
        const int fc = live ? f : 0;
        const int ic = fc / (G_XR * G_XSEG), r2 = fc - ic * (G_XR * G_XSEG), row = r2 / G_XSEG, s = r2 - row * G_XSEG;
        const int gy = c.y0 - 2 + row, gx = c.x0 - 4 + 4 * s;
        const bool ok = (int)((unsigned)gy < (unsigned)h) & (int)((unsigned)gx < (unsigned)w);     // w % 4 == 0: whole segments
        float v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = ok ? rx[j][i] * sX : 0.f;
        half4 hi, lo;
        split4c(v, hi, lo);
        // element e of a patch row holds column x0 - 2 + e: this segment's columns are elements 4s-2 .. 4s+1 = two aligned fp16 pairs;
        // pairs outside the 40 slots and dead tasks go to a dump slot
        uint8_t* rowp = buf + G_LDS_B + ic * G_BC + row * G_BR;
        uint8_t* dump = lds + G2_DUMP;
        const int e0 = 4 * s - 2;
        const bool k0 = (int)live & (int)(e0 >= 0), k1 = (int)live & (int)(e0 + 2 < 40);
        uint8_t* d0 = k0 ? rowp + e0 * 2 : dump;
        uint8_t* d1 = k1 ? rowp + (e0 + 2) * 2 : dump + 8;
        *reinterpret_cast<half2*>(d0) = half2{hi[0], hi[1]};
        *reinterpret_cast<half2*>(d0 + (k0 ? G_BPART : 4)) = half2{lo[0], lo[1]};
        *reinterpret_cast<half2*>(d1) = half2{hi[2], hi[3]};
        *reinterpret_cast<half2*>(d1 + (k1 ? G_BPART : 4)) = half2{lo[2], lo[3]};
    };

    const int kg = lane >> 4, l15 = lane & 15;
    int chunk = slice;
    {
        const Pos c = coords(chunk);
#pragma unroll
        for (int j = 0; j < G_NX4; ++j) load_x(j, c);
#pragma unroll
        for (int j = 0; j < G_NA4; ++j) load_a(j, c);
#pragma unroll
        for (int j = 0; j < G_NX4; ++j) stage_x(j, c, lds);
#pragma unroll
        for (int j = 0; j < G_NA4; ++j) stage_a(j, c, true, lds);
        const Pos c1 = coords(min(chunk + a.slices, nchunk - 1));
#pragma unroll
        for (int j = 0; j < G_NX4; ++j) load_x(j, c1);
#pragma unroll
        for (int j = 0; j < G_NA4; ++j) load_a(j, c1);
    }
    __syncthreads();
    int cur = 0;
    Pos cn = coords(min(chunk + a.slices, nchunk - 1)), c2 = coords(min(chunk + 2 * a.slices, nchunk - 1)), c3 = c2;
#pragma unroll 1
    while (true) {
        const int nxt = chunk + a.slices;
        const bool livec = nxt < nchunk;
        const uint8_t* cb = lds + cur * G2_BUF;
        uint8_t* sb = lds + (cur ^ 1) * G2_BUF;
        const uint8_t* abase = cb + G_LDS_A + l15 * G_AP + kg * 16;
        const uint8_t* bbase = cb + G_LDS_B + l15 * G_BC + kg * 16;
        half8 fah[2], fal[2];                         // A fragments of the wave's two chunk rows
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) {
            fah[rr] = *reinterpret_cast<const half8*>(abase + (wave + 4 * rr) * G_CW * 2);
            fal[rr] = *reinterpret_cast<const half8*>(abase + (wave + 4 * rr) * G_CW * 2 + G_APART);
        }
        half8 bh[2][5], bl[2][5];                     // B fragments [set][tx]
        uintx4 wh[2][2], wl[2][2];                    // the aligned windows they are cut from [set][window]
        auto windows = [&](int blk, int set) {        // block = (rr, ty): patch row wave + 4 rr + ty
            const int rr = blk / 5, ty = blk - 5 * rr;
            const uint8_t* bp = bbase + (wave + 4 * rr + ty) * G_BR;
            wh[set][0] = *reinterpret_cast<const uintx4*>(bp);
            wh[set][1] = *reinterpret_cast<const uintx4*>(bp + 16);
            wl[set][0] = *reinterpret_cast<const uintx4*>(bp + G_BPART);
            wl[set][1] = *reinterpret_cast<const uintx4*>(bp + G_BPART + 16);
        };
        auto cut = [&](int set) {                     // shift tx: elements tx .. tx + 7 of the 16 in the two windows
            const unsigned dh[8] = {wh[set][0][0], wh[set][0][1], wh[set][0][2], wh[set][0][3], wh[set][1][0], wh[set][1][1], wh[set][1][2], wh[set][1][3]};
            const unsigned dl[8] = {wl[set][0][0], wl[set][0][1], wl[set][0][2], wl[set][0][3], wl[set][1][0], wl[set][1][1], wl[set][1][2], wl[set][1][3]};
#pragma unroll
            for (int tx = 0; tx < 5; ++tx) {
                uintx4 fh, fl;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    fh[i] = tx % 2 == 0 ? dh[tx / 2 + i] : __builtin_amdgcn_alignbit(dh[tx / 2 + i + 1], dh[tx / 2 + i], 16);
                    fl[i] = tx % 2 == 0 ? dl[tx / 2 + i] : __builtin_amdgcn_alignbit(dl[tx / 2 + i + 1], dl[tx / 2 + i], 16);
                }
                bh[set][tx] = __builtin_bit_cast(half8, fh);
                bl[set][tx] = __builtin_bit_cast(half8, fl);
            }
        };
        windows(0, 0);
        windows(1, 1);
        cut(0);
        __builtin_amdgcn_sched_barrier(0);
        auto block = [&](auto BLK) {
            constexpr int blk = decltype(BLK)::value;
            constexpr int rr = blk / 5, ty = blk - 5 * rr, set = blk & 1;
            constexpr int p0 = g2_first_piece(blk), p1 = g2_first_piece(blk + 1), npc = p1 - p0;
            if constexpr (blk + 1 < 10) cut(set ^ 1);
            if constexpr (blk + 2 < 10) windows(blk + 2, set);
#pragma unroll
            for (int tx = 0; tx < 5; ++tx) {
                const int t = ty * 5 + tx;
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fal[rr], bh[set][tx], acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fah[rr], bl[set][tx], acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fah[rr], bh[set][tx], acc[t], 0, 0, 0);
            }
#pragma unroll
            for (int pc = p0; pc < p1; ++pc) {
                if (pc < G_NX4) {
                    stage_x(pc, cn, sb);
                    load_x(pc, c2);
                } else {
                    stage_a(pc - G_NX4, cn, livec, sb);
                    load_a(pc - G_NX4, c2);
                }
            }
            if constexpr (blk == 5) c3 = coords(min(nxt + 2 * a.slices, nchunk - 1));
#pragma unroll
            for (int i = 0; i < 15; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                if (blk + 2 < 10 && i < 4) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, npc == 2 ? 9 : 7, 0);
                if (blk == 5) __builtin_amdgcn_sched_group_barrier(0x004, 4, 0);
                if (i >= 15 - 5 * npc) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
                if (i >= 15 - npc) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        };
        block(std::integral_constant<int, 0>{});
        block(std::integral_constant<int, 1>{});
        block(std::integral_constant<int, 2>{});
        block(std::integral_constant<int, 3>{});
        block(std::integral_constant<int, 4>{});
        block(std::integral_constant<int, 5>{});
        block(std::integral_constant<int, 6>{});
        block(std::integral_constant<int, 7>{});
        block(std::integral_constant<int, 8>{});
        block(std::integral_constant<int, 9>{});
        __syncthreads();
        chunk = nxt;
        cur ^= 1;
        cn = c2;
        c2 = c3;
        if (!livec) break;
    }

    // ---- epilogue: as k_wgrad16_f16x3 (the waves take turns adding their tiles in LDS, one coalesced atomic per element)
    float* tile = reinterpret_cast<float*>(lds);
    for (int turn = 0; turn < 4; ++turn) {
        if (wave == turn) {
#pragma unroll
            for (int t = 0; t < 25; ++t) {
                const int tap = a.tap_of[t];
#pragma unroll
                for (int q = 0; q < 4; ++q) {         // D row = oc = 4 kg + q, col = ic = lane & 15
                    float* d = tile + ((4 * kg + q) * 16 + l15) * 25 + tap;
                    *d = turn == 0 ? acc[t][q] : *d + acc[t][q];
                }
            }
        }
        __syncthreads();
    }
    const float inv = a.alpha * (1.f / sX) * (1.f / sY);
    float* dwp = a.dw + (int64_t)plane * 16 * 16 * 25;
    for (int i = tid; i < 16 * 16 * 25; i += 256) atomicAdd(dwp + i, tile[i] * inv);
    if (a.db) {
#pragma unroll
        for (int j = 0; j < G_NA4; ++j) {
            float s_ = dbs[j];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) s_ += __shfl_xor(s_, o, 64);
            const int oc = (tid + 256 * j) >> 6;
            if (lane == 0) atomicAdd(a.db + (int64_t)plane * 16 + oc, s_ * a.alpha);
        }
    }
}

}  // namespace
}  // namespace lldwt
using namespace lldwt;

extern "C" int lldwt_absmax_slots(const float* x, int64_t planes, int64_t n_per_plane, float* slots, void* stream);
extern "C" int lldwt_conv3x3_wgrad_f16x3_ex(const float* x, const float* dy, float* dw, float* dbias, float* slots_ws,
                                            const float* x_slots, const float* dy_slots, int cin, int cout, int64_t planes,
                                            int64_t batch, int64_t h, int64_t w_, float alpha, void* stream);

extern "C" int lldwt_conv3x3_wgrad_f16x3(const float* x, const float* dy, float* dw, float* dbias, float* slots_ws, int cin,
                                         int cout, int64_t planes, int64_t batch, int64_t h, int64_t w_, float alpha,
                                         void* stream) {
    return lldwt_conv3x3_wgrad_f16x3_ex(x, dy, dw, dbias, slots_ws, nullptr, nullptr, cin, cout, planes, batch, h, w_, alpha, stream);
}

// + the per-plane |max| slots of x and / or dy when the caller holds them already (planes x 64 floats each, as lldwt_absmax_slots or
// lldwt_conv2d_absmax leave them): the pass over that tensor is skipped.  slots_ws may be null when both are given.
extern "C" int lldwt_conv3x3_wgrad_f16x3_ex(const float* x, const float* dy, float* dw, float* dbias, float* slots_ws,
                                            const float* x_slots, const float* dy_slots, int cin, int cout, int64_t planes,
                                            int64_t batch, int64_t h, int64_t w_, float alpha, void* stream) {
    LLDWT_REQUIRE(x && dy && dw && (slots_ws || (x_slots && dy_slots)), "conv3x3_wgrad_f16x3: null pointer");
    LLDWT_REQUIRE(cin > 0 && cout > 0 && planes > 0 && planes <= 65535 && batch > 0 && h > 0 && w_ >= 4, "conv3x3_wgrad_f16x3: bad dims");
    LLDWT_REQUIRE(w_ % 4 == 0, "conv3x3_wgrad_f16x3: the row length must be a multiple of 4 (16-byte row segments)");
    LLDWT_REQUIRE((((uintptr_t)x) & 15) == 0 && (((uintptr_t)dy) & 15) == 0, "conv3x3_wgrad_f16x3: x and dy must be 16-byte aligned");
    LLDWT_REQUIRE((int64_t)batch * (cin > cout ? cin : cout) * h * w_ < ((int64_t)1 << 40), "conv3x3_wgrad_f16x3: tensor too large");
    LLDWT_REQUIRE((int64_t)WM * h * w_ < ((int64_t)1 << 31), "conv3x3_wgrad_f16x3: 128 * h * w = %ld exceeds the 32-bit offsets inside an image",
                  (long)((int64_t)WM * h * w_));
    LLDWT_REQUIRE(h < (1 << 23) && w_ < (1 << 23), "conv3x3_wgrad_f16x3: h, w must be below 2^23 (24-bit row offsets)");
    hipStream_t st = (hipStream_t)stream;
    // per-plane max |x| and max |dy| (two passes at HBM speed; 64 slots each)
    const float* sx = x_slots ? x_slots : slots_ws;
    const float* sy = dy_slots ? dy_slots : slots_ws + planes * 64;
    int r;
    if (!x_slots && (r = lldwt_absmax_slots(x, planes, batch * cin * h * w_, slots_ws, stream))) return r;
    if (!dy_slots && (r = lldwt_absmax_slots(dy, planes, batch * cout * h * w_, slots_ws + planes * 64, stream))) return r;
    WgArgs a;
    a.x = x; a.dy = dy; a.dw = dw; a.db = dbias; a.sx = sx; a.sy = sy;
    a.cin = cin; a.cout = cout; a.batch = (int)batch; a.h = (int)h; a.w = (int)w_;
    a.nicb = (int)cdiv(cin, WIC);
    a.nocb = (int)cdiv(cout, WM);
    a.chunks_x = (int)cdiv(w_, CW);
    a.chunks_y = (int)cdiv(h, CR);
    a.alpha = alpha;
    const int64_t nchunk = batch * a.chunks_x * a.chunks_y;
    const int ncol = a.nicb * a.nocb;
    // K slices: a multiple of 8 (one per XCD), about two resident rounds of workgroups over all planes
    // (one workgroup per CU: 101 KB of LDS): the smallest such count that fills whole rounds of the chip, at least two
    const int64_t ncu = lldwt_num_cus();
    int64_t slices = 16;
    for (int64_t s = 8; s <= 64; s += 8)
        if (planes * ncol * s >= 2 * ncu && (planes * ncol * s) % ncu == 0) { slices = s; break; }
    while (slices > 8 && slices * 4 > nchunk) slices -= 8;
    a.slices = (int)slices;
    // LLDWT_WGRAD3=v1 keeps the first kernel (three pre-shifted copies of the input rows, staging between two barriers)
    static const bool v1 = [] { const char* e = getenv("LLDWT_WGRAD3"); return e && !strcmp(e, "v1"); }();
    static bool attr = false;
    if (!attr) {
        if (hipFuncSetAttribute((const void*)k_wgrad3_f16x3, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_TOTAL) != hipSuccess ||
            hipFuncSetAttribute((const void*)k_wgrad3_f16x3_v2, hipFuncAttributeMaxDynamicSharedMemorySize, V2_TOTAL) != hipSuccess) {
            set_error("conv3x3_wgrad_f16x3: cannot reserve %d bytes of LDS", V2_TOTAL);
            return LLDWT_EHIP;
        }
        attr = true;
    }
    dim3 grid((unsigned)(slices * ncol), 1, (unsigned)planes);
    if (v1) hipLaunchKernelGGL(k_wgrad3_f16x3, grid, dim3(256), LDS_TOTAL, st, a);
    else hipLaunchKernelGGL(k_wgrad3_f16x3_v2, grid, dim3(256), V2_TOTAL, st, a);
    return check_launch("conv3x3_wgrad_f16x3");
}


// 16 -> 16 5x5 weight gradient of a P/U block on the fp16 matrix cores (see k_wgrad16_f16x3).  x must be bounded by 1 in magnitude
// (the tanh outputs t1 / t2); slots_ws: planes * 64 floats.  tap_of: 25 entries, tap (ty*5+tx) -> position inside dW[oc][ic].
namespace lldwt {
int wgrad16_f16x3(const float* x, const float* dy, float* dw, float* dbias, float* slots_ws, int64_t slots_stride, bool slots_ready,
                  int64_t planes, int64_t batch, int64_t h, int64_t w_, float alpha, const int8_t* tap_of, hipStream_t st) {
    LLDWT_REQUIRE(x && dy && dw && slots_ws && tap_of, "wgrad16_f16x3: null pointer");
    LLDWT_REQUIRE(planes > 0 && planes <= 65535 && batch > 0 && h > 0 && w_ >= 4 && w_ % 4 == 0, "wgrad16_f16x3: bad dims");
    LLDWT_REQUIRE((((uintptr_t)x) & 15) == 0 && (((uintptr_t)dy) & 15) == 0, "wgrad16_f16x3: x and dy must be 16-byte aligned");
    // slots_ready: the producer of dy left its per-plane |max| in the slots already (64 per plane, plane stride slots_stride)
    if (!slots_ready) {
        LLDWT_REQUIRE(slots_stride == 64, "wgrad16_f16x3: the |max| pass writes 64 slots per plane");
        int r = lldwt_absmax_slots(dy, planes, batch * 16 * h * w_, slots_ws, st);
        if (r) return r;
    }
    Wg16Args a;
    a.x = x; a.dy = dy; a.dw = dw; a.db = dbias; a.sy = slots_ws; a.sy_stride = (int)slots_stride;
    a.batch = (int)batch; a.h = (int)h; a.w = (int)w_; a.alpha = alpha;
    a.chunks_x = (int)cdiv(w_, G_CW);
    a.chunks_y = (int)cdiv(h, G_CR);
    for (int t = 0; t < 25; ++t) a.tap_of[t] = tap_of[t];
    const int64_t nchunk = batch * a.chunks_x * a.chunks_y;
    // LLDWT_WGRAD16K=v1 keeps the first kernel (two workgroups per CU, a chunk's loads issued and consumed between the same barriers)
    static const bool v1 = [] { const char* e = getenv("LLDWT_WGRAD16K"); return e && !strcmp(e, "v1"); }();
    static bool attr = false;
    static int per_cu = 2;
    if (!attr) {
        if (hipFuncSetAttribute((const void*)k_wgrad16_f16x3, hipFuncAttributeMaxDynamicSharedMemorySize, G_LDS_TOTAL) != hipSuccess ||
            hipFuncSetAttribute((const void*)k_wgrad16_f16x3_v2, hipFuncAttributeMaxDynamicSharedMemorySize, G2_TOTAL) != hipSuccess) {
            set_error("wgrad16_f16x3: cannot reserve %d bytes of LDS", G2_TOTAL);
            return LLDWT_EHIP;
        }
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)k_wgrad16_f16x3, 256, G_LDS_TOTAL) != hipSuccess || per_cu < 1)
            per_cu = 2;
        attr = true;
    }
    LLDWT_REQUIRE((int64_t)16 * h * w_ < ((int64_t)1 << 31) && h < (1 << 23) && w_ < (1 << 23), "wgrad16_f16x3: image too large for 32-bit offsets");
    // one resident round over all planes (equal-time workgroups), at least 2 chunks per workgroup
    int64_t slices = (int64_t)lldwt_num_cus() * (v1 ? per_cu : 1) / planes;
    if (slices > nchunk / 2) slices = nchunk / 2;
    if (slices >= 16) slices &= ~(int64_t)7;            // a multiple of 8: the XCD-aware chunk order of the kernel
    if (slices < 1) slices = 1;
    a.slices = (int)slices;
    dim3 grid((unsigned)slices, 1, (unsigned)planes);
    if (v1) hipLaunchKernelGGL(k_wgrad16_f16x3, grid, dim3(256), G_LDS_TOTAL, st, a);
    else hipLaunchKernelGGL(k_wgrad16_f16x3_v2, grid, dim3(256), G2_TOTAL, st, a);
    return check_launch("wgrad16_f16x3");
}
}  // namespace lldwt

extern "C" int lldwt_wgrad16_f16x3(const float* x, const float* dy, float* dw, float* dbias, float* slots_ws, int64_t planes,
                                   int64_t batch, int64_t h, int64_t w_, float alpha, int swap_hw, void* stream) {
    int8_t tap_of[25];
    for (int t = 0; t < 25; ++t) tap_of[t] = (int8_t)(swap_hw ? (t % 5) * 5 + t / 5 : t);
    return lldwt::wgrad16_f16x3(x, dy, dw, dbias, slots_ws, 64, false, planes, batch, h, w_, alpha, tap_of, (hipStream_t)stream);
}
