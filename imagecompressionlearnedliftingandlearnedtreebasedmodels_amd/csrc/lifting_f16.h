// lifting_f16.h -- shared between lifting.hip (pack layout, step dispatch) and lifting_f16.hip (the fused split-fp16
// lifting-step kernel).  Internal to the library.
#pragma once
#include "common.h"

namespace lldwt {

// The fused kernel exists for the reference's configuration: 16 channels (depth_scale*8, liftingDWT.json:22), 5x5.
constexpr int LF_C = 16, LF_K = 5, LF_KK = 25;
constexpr int LF_KS = 13;                 // k-steps of 32 = 2 taps x 16 channels (25 taps -> 26, the last one zero)
constexpr int LF_KS4 = 3;                 // conv4 as D[dx][px]: K = 5 dy x 16 ch = 80 -> 3 k-steps of 32
constexpr int LF_FRAG = 1024;             // halves per k-step: (hi, lo) x 64 lanes x 8
// per orientation, in HALVES: conv1 | conv2 | conv3 | conv4 fragments; then 16 floats: s_w1..s_w4, b1[16]... (see .hip)
constexpr int LF_H_C1 = 0;
constexpr int LF_H_C2 = LF_H_C1 + LF_FRAG;
constexpr int LF_H_C3 = LF_H_C2 + LF_KS * LF_FRAG;
constexpr int LF_H_C4 = LF_H_C3 + LF_KS * LF_FRAG;
constexpr int LF_KSC = 5;                 // conv4 o conv3 composed (9x9, 16 -> 1) as D[dx][px]: K = 9 dy x 16 ch = 144 -> 5 k-steps
constexpr int LF_H_CC = LF_H_C4 + LF_KS4 * LF_FRAG;           // composite fragments
constexpr int LF_H_END = LF_H_CC + LF_KSC * LF_FRAG;          // 35 * 1024 halves
// floats after the fragments: [0..3] s_w1..s_w4, [4] s_wc, [5] b4 + sum_oc (b3 + b1)[oc] * sum_taps w4[oc] (the composed path's
// constant), [16..31] sum over taps of w4[oc], [32..112] Wr[81] = w4 o w1
constexpr int LF_TAIL_FLOATS = 128;
// behind the tail, in HALVES from the section start: a bf16 copy of the scaled weights, 512 per k-step (the one-product bf16 mode)
constexpr int LF_H_BF = LF_H_END + 2 * LF_TAIL_FLOATS;
constexpr int LF_NSTEP = LF_H_END / LF_FRAG;                  // 35 k-steps
constexpr int LF_ORIENT_FLOATS = (LF_H_BF + LF_NSTEP * 512) / 2;
constexpr int LF_FLOATS = 2 * LF_ORIENT_FLOATS;               // both orientations

static inline __host__ __device__ int lift_f16_floats(int C, int K) { return (C == LF_C && K == LF_K) ? LF_FLOATS : 0; }

// pack the f16 section of one P/U block (called from lldwt_pack_pblock after the fp32 section is written)
int lift_f16_pack(const float* w1, const float* w2, const float* w3, const float* w4, const float* b1, const float* b3,
                  const float* b4, float* packed, int64_t plane_stride, int f16_off, int planes, int compose, hipStream_t st);

// one fused lifting step (eval): dst_out = dst_in + sign * (skip + rw * P(skip)); returns LLDWT_OK or an error
struct LiftF16Views {
    const float* src;  int64_t src_sz, src_sy, src_sx;
    const float* din;  int64_t din_sz, din_sy, din_sx;
    float* dout;       int64_t dout_sz, dout_sy, dout_sx;
};
int lift_f16_step(const LiftF16Views& v, int64_t Z, int64_t batch, int64_t h, int64_t w, const float* taps,
                  const float* packed, int64_t pstride, int fp32_orient_floats, int f16_off, int vertical, float sign, float rw,
                  hipStream_t st);
// the same step for TWO independent view sets of the same geometry and parameters in one launch (v2 may be null): images
// 0 .. Z-1 use v, images Z .. 2Z-1 use *v2
// the training forward of one step on the fused kernel's sequential path: the same step, and the tile interiors of src, skip
// (Z, h, w) and t1, t2, t3 (Z, 16, h, w) written out for the backward (what k_lift_a/b/c save on the fp32 path)
struct LiftF16Saved { float* src; float* skip; float* t1; float* t2; float* t3; };
// backward-data of one step on the same kernel (BWD mode): g = dL/dnet dense (Z, h, w); t1, t2 = the saved tanh outputs; out:
// dt3, dpre2, dr (Z, 16, h, w) and dsk (Z, h, w).  packed_bwd: lift_f16_pack_bwd's buffer (fp32 section zero); taps_id: (planes, 3)
// floats (0, 1, 0)
// mx: null, or (planes, 2, 64) floats zeroed by the caller: the launch leaves max |dt3| (row 0) and max |dpre2| (row 1) of every
// plane spread over the 64 slots (atomic max), the form lldwt::wgrad16_f16x3 takes its dY maximum in
struct LiftF16Bwd { const float* g; const float* t1; const float* t2; float* dt3; float* dpre2; float* dr; float* dsk; float* mx; };
int lift_f16_step_bwd(const LiftF16Bwd& b, int64_t Z, int64_t batch, int64_t h, int64_t w, const float* taps_id,
                      const float* packed_bwd, int64_t pstride, int fp32_orient_floats, int f16_off, int vertical, hipStream_t st);
int lift_f16_pack_bwd(const float* w1, const float* w2, const float* w3, const float* w4, float* scratch, float* packed,
                      int64_t plane_stride, int f16_off, int planes, hipStream_t st);
int64_t lift_f16_bwd_scratch_floats(int planes);
int lift_f16_step_train(const LiftF16Views& v, const LiftF16Saved& sv, int64_t Z, int64_t batch, int64_t h, int64_t w,
                        const float* taps, const float* packed, int64_t pstride, int fp32_orient_floats, int f16_off,
                        int vertical, float sign, float rw, hipStream_t st);
int lift_f16_step_any(const LiftF16Views& v, const LiftF16Views* v2, const LiftF16Saved* sv, int64_t Z, int64_t batch, int64_t h,
                      int64_t w, const float* taps, const float* packed, int64_t pstride, int fp32_orient_floats, int f16_off,
                      int vertical, float sign, float rw, hipStream_t st);
// arithmetic of the split-fp16 kernel families (lldwt_set_precision): 0 = f16x3 (three products per MAC, fp32-level accuracy),
// 1 = fp16, 2 = bf16 (one product per MAC).  Defined in lifting_f16.hip, read at launch by conv_f16x3.hip and cgp_f16x3.hip too
int split_precision();
void split_set_precision(int p);
int lift_f16_step2(const LiftF16Views& v, const LiftF16Views* v2, int64_t Z, int64_t batch, int64_t h, int64_t w,
                   const float* taps, const float* packed, int64_t pstride, int fp32_orient_floats, int f16_off, int vertical,
                   float sign, float rw, hipStream_t st);

// diagnostics (lldwt_set_diagnostics): debug mask of the fused step (bit i = skip phase i+1, 16 = sequential conv3 / conv4,
// 32 = no vertical reuse) and the stamp buffers of the three split-fp16 kernel families
void lift_f16_set_debug(int dbg);
void lift_f16_set_stamps(void* p, int64_t nbytes);
void f3_set_stamps(void* p, int64_t nbytes);
void cgp16_set_stamps(void* p, int64_t nbytes);

}  // namespace lldwt
