/*
 * lldwt.h -- C-ABI of the MI355X-native learned-lifting DWT + CNN entropy-model hot path.
 *
 * Drop-in boundary.  The reference (uberkk/ImageCompressionLearnedLiftingandLearnedTreeBasedModels)
 * is pure PyTorch and defines no FFI; its boundary for this path is the Python module API
 * (agents/liftingDWT_agent.py, graphs/models/LiftingBasedDWT_net.py).  The entry points below are what the
 * host-side mirror of those modules binds through ctypes; each one cites the reference
 * code it replaces (paths relative to the reference root).
 *
 * Conventions
 *  - plain pointers and sizes only; every pointer is DEVICE memory owned by the caller (allocated by the host
 *    framework, e.g. PyTorch's caching allocator).  The library never allocates or frees user-visible memory;
 *    scratch space is passed in as `ws` and sized by the matching *_ws_bytes() query.
 *  - `stream` is a hipStream_t passed as void*; kernels are enqueued on it, no call synchronises.
 *  - return value: 0 on success, negative LLDWT_E* otherwise; lldwt_last_error() gives a message
 *    (thread-local, valid until the next failing call on the thread).  No exceptions cross the boundary.
 *  - tensors are fp32, dense, "plane-major NCHW": shape (Z, C, h, w) with Z = planes*batch, plane-major
 *    (z = plane*batch + b).  A *plane* is one colour channel processed by its own network (clrch == 1:
 *    LiftingBasedDWT_net.py:43-46); parameters of the `planes` networks are stacked on a leading axis and
 *    addressed as base + plane*stride.
 */
#ifndef LLDWT_H
#define LLDWT_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LLDWT_OK 0
#define LLDWT_EINVAL (-1)   /* bad argument / unsupported shape */
#define LLDWT_EHIP (-2)     /* HIP runtime error (launch failed) */
#define LLDWT_EWS (-3)      /* workspace too small */

#define LLDWT_ACT_NONE 0
#define LLDWT_ACT_TANH 1
#define LLDWT_ACT_LRELU 2   /* LeakyReLU(0.01) */
#define LLDWT_ACT_RELU 3    /* ReLU (post-processing networks, post_processing_networks.py:45,60-70) */

const char* lldwt_last_error(void);
int lldwt_version(void);
/* 1 if the calling process sees a gfx950 device, 0 otherwise (no compute is launched). */
int lldwt_device_ok(void);

/* ---------------------------------------------------------------------------------------------------------
 * Strided view of a (Z, h, w) single-channel tensor: element (z,y,x) at p[z*sz + y*sy + x*sx].
 * Used for the polyphase components: the even/odd rows (or columns) of an array are views with sy (sx)
 * doubled, so the split (wavelet_forward_v2.py:27-28,33-34,45-46) and the merge
 * (wavelet_inverse_v2.py:49-53) are pure addressing -- no copies, bit-exact by construction.            */
typedef struct lldwt_view {
    float* p;
    int64_t sz, sy, sx;
} lldwt_view;

/* ---------------------------------------------------------------------------------------------------------
 * Colour transforms of the agent (agents/liftingDWT_agent.py:86-87,90-94; compressai RGB2YCbCr/YCbCr2RGB,
 * BT.709).  rgb: (B,3,H,W) NCHW in [0,1].  ycc: plane-major (3,B,1,H,W) with 0.5 subtracted from Y only.
 * inverse: ycc -> rgb, then "- 0.5" (agents/liftingDWT_agent.py:94) and optional clamp to [-0.5,0.5] (:181). */
int lldwt_rgb_to_ycc(const float* rgb, float* ycc, int64_t B, int64_t H, int64_t W, void* stream);
int lldwt_ycc_to_rgb(const float* ycc, float* rgb, int64_t B, int64_t H, int64_t W, int clamp, void* stream);
/* backward of lldwt_ycc_to_rgb without clamp (training): grgb (B,3,H,W) -> gycc plane-major (3,B,1,H,W). */
int lldwt_ycc_to_rgb_bwd(const float* grgb, float* gycc, int64_t B, int64_t H, int64_t W, void* stream);

/* Input pipeline (dataloaders/image_dl.py:63-105: PIL decode, crop, ToTensor): the host decodes and crops to uint8
 * HWC (3 B/pixel over PCIe instead of 12); this converts a (B,H,W,3) uint8 batch to (B,3,H,W) fp32 in [0,1] with
 * ToTensor's arithmetic (x / 255 in fp32), bit-exact.                                                          */
int lldwt_u8hwc_to_f32chw(const uint8_t* src, float* dst, int64_t B, int64_t H, int64_t W, void* stream);

/* ---------------------------------------------------------------------------------------------------------
 * P/U block parameters (graphs/layers/P_block_v2.py:15-33) packed for the kernels.
 * lldwt_pack_pblock: in = the four conv weights/biases of `planes` stacked blocks in PyTorch layout
 *   w1 (planes,C,1,K,K)  w2,w3 (planes,C,C,K,K)  w4 (planes,1,C,K,K), b1,b2,b3 (planes,C), b4 (planes,1);
 *   out = packed buffer of lldwt_pblock_packed_floats(C,K) floats PER PLANE holding both orientations
 *   (vertical pass uses W, horizontal pass uses W transposed in (kh,kw): conv(x^T,W)^T == conv(x,W^T), which
 *   removes every torch.transpose of wavelet_forward_v2.py:32,38-39,43,50-51).                           */
int64_t lldwt_pblock_packed_floats(int C, int K);
/* Arithmetic of the eval-path lifting step for C = 16, K = 5 (the reference's configuration): 1 (default) = ONE fused
 * launch per step with the 16 -> 16 convolutions on the fp16 matrix cores, split-fp16 operands, intermediates in LDS
 * (csrc/lifting_f16.hip); 0 = the three fp32-MFMA launches (exact fp32 products; also what training uses).       */
int lldwt_set_lift_mode(int mode);
/* Arithmetic of the split-fp16 kernel families on the EVAL path (fused lifting step, tree-context pair, cgp chain):
 *   0 (default) = f16x3: every fp32 MAC is three fp16 MFMA products (hi hi + hi lo + lo hi), fp32 accumulate: fp32-level
 *                 accuracy; every parity bar against the fp32 reference is stated for this mode;
 *   1 = fp16, 2 = bf16: ONE MFMA product per MAC on operands rounded to fp16 / bf16 (fp32 accumulate, fp32 tensors in HBM,
 *                 biases / activations / rate arithmetic in fp32).  BASELINE configs[4] ("fp16") and configs[1] ("bf16");
 *                 their own tolerance class (<= 1e-2 relative on coefficients and summed bits), never the headline.
 * The reference itself is fp32-only (graphs/layers/wavelet_inverse_v2.py:49-51).  Training always runs f16x3 / fp32.   */
int lldwt_set_precision(int prec);
int lldwt_get_precision(void);
/* Diagnostics hook of the three split-fp16 kernel families (tools/lift_stamps.py, plc_stamps.py, cgp_stamps.py; the
 * product never calls it).  kind 0 = fused lifting step, 1 = tree-pair conv, 2 = cgp chain.  stamps / nbytes: a device
 * buffer the kernels of that kind write s_memtime stamps into (null = off); a launch whose stamps would not fit in
 * nbytes ignores the buffer.  flags (kind 0 only): debug mask -- bit 0..3 skip a phase (results are then wrong),
 * 16 = sequential conv3 / conv4 for every tile (the check of the composed 9x9 kernel), 32 = no vertical reuse
 * between the tiles of a column.  The caller keeps the buffer alive until it unregisters it.                      */
int lldwt_set_diagnostics(int kind, void* stamps, int64_t nbytes, int flags);
int lldwt_get_lift_mode(void);
/* 1 when the TRAINING forward of the lifting steps (lldwt_lifting_forward_train / _inverse_train, C = 16, K = 5, tanh blocks)
 * runs on the fused f16x3 kernel's sequential path, which writes (src, skip, t1, t2, t3) for the backward: lift mode 1 and the
 * environment variable LLDWT_TRAIN_LIFT != "f32".  The packed blocks must then come from lldwt_pack_pblock (with the split-fp16
 * section), not lldwt_pack_pblock_train. */
int lldwt_train_lift_f16(void);
int lldwt_pack_pblock(const float* w1, const float* b1, const float* w2, const float* b2, const float* w3,
                      const float* b3, const float* w4, const float* b4, float* packed, int planes, int C, int K,
                      void* stream);
/* The same buffer WITHOUT its split-fp16 section (left unwritten): for the training step, which re-packs the changed
 * weights every iteration and only runs the fp32 kernels that save their intermediates.  A buffer packed this way must
 * not be handed to the eval path (lldwt_lift_step with lift mode 1 reads the split-fp16 section).                     */
int lldwt_pack_pblock_train(const float* w1, const float* b1, const float* w2, const float* b2, const float* w3,
                            const float* b3, const float* w4, const float* b4, float* packed, int planes, int C, int K,
                            void* stream);
/* As lldwt_pack_pblock without the composed 9x9 kernels of the eval path (left zero): the pack of the fused kernel's sequential
 * path -- the training forward (lldwt_lifting_forward_train / _inverse_train), rebuilt at every weight update.  NOT for the eval
 * entry points.                                                                                                              */
int lldwt_pack_pblock_seq(const float* w1, const float* b1, const float* w2, const float* b2, const float* w3,
                          const float* b3, const float* w4, const float* b4, float* packed, int planes, int C, int K,
                          void* stream);

/* One lifting step (wavelet_forward_v2.py:60-62 and the three like it; inverse wavelet_inverse_v2.py:76-90):
 *     skip = conv3x1(src, taps)            zero padded, along rows if vertical else along columns
 *     net  = P_block(skip)                 conv1 -> tanh -> conv2 -> tanh -> conv3 (+conv1 pre-act) -> conv4
 *     dst_out = dst_in + sign * (skip + res_weight * net)
 * src/dst_in/dst_out are (Z,h,w) views; dst_out may alias dst_in.  taps: (planes,3) device floats
 * (preProcessingList.{j}.weight, lifting_dwt_nets.py:785-819).  packed: lldwt_pack_pblock output.
 * linear != 0 drops the tanh (linearity_flag != 1, P_block_v2.py:42-49).
 * ws: >= lldwt_lift_step_ws_bytes(Z,h,w,C) bytes.                                                        */
int64_t lldwt_lift_step_ws_bytes(int64_t Z, int64_t h, int64_t w, int C);
int lldwt_lift_step(lldwt_view src, lldwt_view dst_in, lldwt_view dst_out, int64_t Z, int64_t batch, int64_t h,
                    int64_t w, const float* taps, const float* packed, int C, int K, int vertical, float sign,
                    float res_weight, int linear, void* ws, int64_t ws_bytes, void* stream);

/* Whole multi-level transform of LiftingBasedNeuralWaveletv4.encode (lifting_dwt_nets.py:728-732), all planes
 * and images in one call.  x: (Z,1,H,W) plane-major; ll: (Z,1,H>>L,W>>L); yh[i]: (Z,3,H>>(i+1),W>>(i+1)) with
 * channels (LH,HL,HH), finest first (the layout of out_xo_list before the subband auto-encoder, :739-740).
 * taps: (4,planes,3) = preProcessingList.{j}.weight of every plane, j-major.
 * packed: (planes, nblocks, 2 (P,U), packed_floats); nblocks = 2 for block_property=="same", 2*2*levels for
 * "different" (lifting_dwt_nets.py:688-722).  Level `lev` of the forward uses block pairs
 * block_offset + (different ? 2*lev : 0) + {0,1}; EVERY level of the inverse uses block_offset + {0,1}
 * (lifting_dwt_nets.py:718-722 slices the inverse blocks from waveletLevel*liftingLevel for every level, so the
 * caller passes block_offset = 2*levels there when block_property=="different").
 * scale_nh/scale_nl: device (planes) floats = lifting_coeff[4/5] + n*0.1 when config.scale==1, else NULL.  */
int64_t lldwt_lifting_ws_bytes(int64_t Z, int64_t H, int64_t W, int C);
int lldwt_lifting_forward(const float* x, float* ll, float* const* yh, int64_t planes, int64_t batch, int64_t H,
                          int64_t W, int levels, const float* taps, const float* packed, int nblocks,
                          int block_offset, int different, int C, int K, float res_weight, int linear,
                          const float* scale_nh, const float* scale_nl, void* ws, int64_t ws_bytes, void* stream);
/* Inverse (LiftingBasedNeuralWaveletv4.decode, lifting_dwt_nets.py:762-781; wavelet_inverse_v2.py:20-92).  */
int lldwt_lifting_inverse(const float* ll, const float* const* yh, float* x, int64_t planes, int64_t batch,
                          int64_t H, int64_t W, int levels, const float* taps, const float* packed, int nblocks,
                          int block_offset, int C, int K, float res_weight, int linear, const float* scale_nh,
                          const float* scale_nl, void* ws, int64_t ws_bytes, void* stream);

/* Training support of the lifting transform.  The transform is a PROGRAM of lifting steps over symbolic buffers
 * (0 = x, 1 = Lrow, 2 = Hrow, 3 = tmpL, 4 = tmpH, 5/6 = LL ping-pong, 7 = ll, 8+i = yh[i]); views are
 * (buffer, element offset, z/row/col strides).  kind 0 = lifting step, 1..4 = per-plane scale (config.scale == 1).
 * lldwt_lifting_program fills `ops` (pass NULL/0 to count) and reports the floats of the per-step `saved`
 * intermediates [src | skip | t1 | t2 | t3] kept by the *_train variants; the backward pass walks the program in
 * reverse over gradient buffers of identical layout (see autograd.py: G[dst_in] = G[dst_out];
 * G[src] += J^T G[dst_out]; dW/dtaps accumulate).                                                          */
typedef struct lldwt_lift_op {
    int32_t kind, buf_src, buf_din, buf_dout;
    int64_t off_src, sz_src, sy_src, sx_src;
    int64_t off_din, sz_din, sy_din, sx_din;
    int64_t off_dout, sz_dout, sy_dout, sx_dout;
    int32_t h, w, vertical, tap, block, is_u;
    float sign;
    int32_t pad_;
    int64_t saved_off;
} lldwt_lift_op;
int lldwt_lifting_program(lldwt_lift_op* ops, int max_ops, int64_t Z, int64_t H, int64_t W, int levels, int different,
                          int block_offset, int inverse, int scale, int C, int64_t* saved_floats);
int lldwt_lifting_forward_train(const float* x, float* ll, float* const* yh, int64_t planes, int64_t batch, int64_t H,
                                int64_t W, int levels, const float* taps, const float* packed, int nblocks,
                                int block_offset, int different, int C, int K, float res_weight, int linear, void* ws,
                                int64_t ws_bytes, float* saved, void* stream);
int lldwt_lifting_inverse_train(const float* ll, const float* const* yh, float* x, int64_t planes, int64_t batch,
                                int64_t H, int64_t W, int levels, const float* taps, const float* packed, int nblocks,
                                int block_offset, int C, int K, float res_weight, int linear, void* ws, int64_t ws_bytes,
                                float* saved, void* stream);
/* the same with the per-plane gains of config.scale == 1 (wavelet_forward_v2.py:76-80, wavelet_inverse_v2.py:70-74; both
 * null = no scaling).  Build the program with scale = 1: every scale op (kind 1..4) then owns h*w*Z floats of `saved` at
 * its saved_off and keeps its INPUT there (dense Z,h,w), which the host-side backward needs for d(gain). */
int lldwt_lifting_forward_train_ex(const float* x, float* ll, float* const* yh, int64_t planes, int64_t batch, int64_t H,
                                   int64_t W, int levels, const float* taps, const float* packed, int nblocks,
                                   int block_offset, int different, int C, int K, float res_weight, int linear,
                                   const float* scale_nh, const float* scale_nl, void* ws, int64_t ws_bytes, float* saved,
                                   void* stream);
int lldwt_lifting_inverse_train_ex(const float* ll, const float* const* yh, float* x, int64_t planes, int64_t batch,
                                   int64_t H, int64_t W, int levels, const float* taps, const float* packed, int nblocks,
                                   int block_offset, int C, int K, float res_weight, int linear, const float* scale_nh,
                                   const float* scale_nl, void* ws, int64_t ws_bytes, float* saved, void* stream);
/* backward pieces of one step (chained by the host with lldwt_conv2d / lldwt_conv2d_wgrad_ex):
 *   pre: g (dense Z,h,w) = G[dst_out];  G[dst_in] = g
 *   fin: dskip = sign*(g + res_weight*dsk);  G[src] += taps^T (x) dskip;  dtaps (planes,3) += sum dskip * src shifted */
int lldwt_lift_bwd_pre(lldwt_view g_dst_out, lldwt_view g_dst_in, float* g, int64_t Z, int64_t h, int64_t w, void* stream);
int lldwt_lift_bwd_fin(const float* g, const float* dsk, const float* srcv, lldwt_view g_src, int64_t Z, int64_t batch,
                       int64_t h, int64_t w, const float* taps, float* dtaps, int vertical, float sign, float res_weight,
                       void* stream);

/* Whole backward of one lifting step with C == 16 (replaces autograd through wavelet_forward_v2.py:60-74 /
 * wavelet_inverse_v2.py:76-90 + P_block_v2.py:40-55): fused backward-data kernels on the matrix cores, the four
 * weight gradients (accumulated, scaled by sign*res_weight, row passes in (kw,kh) order like the weights' use), the
 * skip-filter transpose and its tap gradients.  saved_step: this step's slice of the forward `saved` buffer
 * [src | skip | t1 | t2 | t3]; packed/packed_plane_stride: the FORWARD pack of this step's P/U block
 * (lldwt_pack_pblock); dw1..db4: (planes, ...) gradients of that block, accumulated; taps/dtaps: (planes,3) of this
 * step's skip filter.  ws: lldwt_lift_step_bwd_ws_bytes. */
int64_t lldwt_lift_step_bwd_ws_bytes(int64_t Z, int64_t h, int64_t w, int C);
int lldwt_lift_step_bwd(lldwt_view g_dst_out, lldwt_view g_dst_in, lldwt_view g_src, const float* saved_step,
                        int64_t planes, int64_t batch, int64_t h, int64_t w, const float* taps, float* dtaps,
                        const float* packed, int64_t packed_plane_stride, float* dw1, float* db1, float* dw2, float* db2,
                        float* dw3, float* db3, float* dw4, float* db4, int C, int K, float res_weight, float sign,
                        int vertical, int linear, void* ws, int64_t ws_bytes, void* stream);

/* The same with the backward-data chain (dt3 = conv4^T g, dpre2 = tanh'(t2) conv3^T dt3, dr = tanh'(t1) conv2^T dpre2 + dt3,
 * dsk = conv1^T dr) as ONE launch of the fused split-fp16 lifting kernel in its backward mode (C == 16, K == 5, tanh block;
 * anything else, LLDWT_BWD_LIFT=f32 or lift mode f32 takes the launches of lldwt_lift_step_bwd).  packed_bwd: the "backward
 * pack" of the block from lldwt_pack_pblock_bwd -- transposed, mirrored weights in the forward pack's layout, same plane stride;
 * taps_id: (planes, 3) floats (0, 1, 0).  lldwt_bwd_lift_f16(): 1 when the fused backward is what this entry runs. */
int lldwt_bwd_lift_f16(void);
int64_t lldwt_pack_pblock_bwd_ws_bytes(int planes);
int lldwt_pack_pblock_bwd(const float* w1, const float* w2, const float* w3, const float* w4, float* packed, void* ws,
                          int64_t ws_bytes, int planes, int C, int K, void* stream);
int lldwt_lift_step_bwd_f16(lldwt_view g_dst_out, lldwt_view g_dst_in, lldwt_view g_src, const float* saved_step,
                            int64_t planes, int64_t batch, int64_t h, int64_t w, const float* taps, float* dtaps,
                            const float* packed, int64_t packed_plane_stride, float* dw1, float* db1, float* dw2, float* db2,
                            float* dw3, float* db3, float* dw4, float* db4, int C, int K, float res_weight, float sign,
                            int vertical, int linear, void* ws, int64_t ws_bytes, const float* packed_bwd,
                            const float* taps_id, void* stream);

/* ---------------------------------------------------------------------------------------------------------
 * SubbandAutoEncoder (lifting_dwt_nets.py:99-110): per-coefficient scalar MLP 1 -> Hd -> Hd -> Hd -> 1, tanh
 * between, grouped 1x1 convs (groups == channels).  x,y: (Z,C,h,w).  Parameters per plane, PyTorch layouts:
 *   encode: w0 (C*Hd,1) b0 (C*Hd)  w1,w2 (C*Hd,Hd) b1,b2 (C*Hd)  w3 (C,Hd) b3 (C)         [Conv2d]
 *   decode: ConvTranspose2d weights (in, out/groups): w0 (C,Hd) w1,w2 (C*Hd,Hd) w3 (C*Hd,1); transposed != 0. */
int lldwt_subband_mlp(const float* x, float* y, int64_t planes, int64_t batch, int C, int64_t hw, int Hd,
                      const float* w0, const float* b0, const float* w1, const float* b1, const float* w2,
                      const float* b2, const float* w3, const float* b3, int transposed, void* stream);
/* Backward of the encode-layout MLP (training; replaces autograd through lifting_dwt_nets.py:99-104 / :105-110 once the
 * decoder's ConvTranspose2d weights are viewed in Conv2d layout): recomputes the forward, returns gx (Z,C,hw) and, for
 * the four weight-gradient GEMMs (lldwt_conv2d_wgrad, groups == C), the activations h0,h1,h2 and the pre-activation
 * gradients d0,d1,d2, all (Z, C*Hd, hw). */
int lldwt_subband_mlp_bwd(const float* x, const float* gy, float* gx, float* h0, float* h1, float* h2, float* d0, float* d1,
                          float* d2, int64_t planes, int64_t batch, int C, int64_t hw, int Hd, const float* w0,
                          const float* b0, const float* w1, const float* b1, const float* w2, const float* b2,
                          const float* w3, void* stream);
/* The same backward with the eight parameter gradients formed inside the kernel (training default): only x, gy and gx touch HBM.
 * dw0, db0, db1, db2, dw3: (planes, C*Hd); dw1, dw2: (planes, C*Hd, Hd); db3: (planes, C) -- WRITTEN (not accumulated; fixed
 * summation order: the result does not depend on scheduling).  ws: lldwt_subband_mlp_bwd_w_ws_bytes (per-wave partial sums). */
int64_t lldwt_subband_mlp_bwd_w_ws_bytes(int64_t planes, int C, int64_t hw);
int lldwt_subband_mlp_bwd_w(const float* x, const float* gy, float* gx, int64_t planes, int64_t batch, int C, int64_t hw, int Hd,
                            const float* w0, const float* b0, const float* w1, const float* b1, const float* w2,
                            const float* b2, const float* w3, float* dw0, float* db0, float* dw1, float* db1, float* dw2,
                            float* db2, float* dw3, float* db3, void* ws, int64_t ws_bytes, void* stream);

/* ---------------------------------------------------------------------------------------------------------
 * General conv layer for the context models and the Berk auto-encoder
 * (LiftingBasedDWT_net.py:271-289,299-317,793-795; lifting_dwt_nets.py:139-150; masked_conv2d.py:19-21).
 *   y[z, oc', :, :] = act( bias[oc] + sum_{ic,ky,kx} w[oc, ic, ky, kx] * x[z, g*cin_g + ic, .+ky-K/2, .+kx-K/2] )
 * zero padded, stride 1.  x: (Z,cin,h,w) (if upsample2: (Z,cin,h/2,w/2), read through the nearest-neighbour 2x
 * upsampling of LiftingBasedDWT_net.py:348,367,822,835).  w: (planes,cout,cin/groups,K,K) (if transposed:
 * ConvTranspose2d layout (planes,cin,cout,K,K), groups must be 1), bias (planes,cout) or NULL.
 * tap_mask: bit (ky*K+kx) set = tap is live (MaskedConv2d type A/B; the caller has already applied
 * weight *= mask as the reference does); pass (1<<K*K)-1 for a dense conv.
 * Output channel placement (removes the chunk/cat regrouping of LiftingBasedDWT_net.py:357-359):
 *   oc' = (oc / oc_block) * oc_stride + oc_off + oc % oc_block   in a tensor of ytot channels;
 *   dense placement is oc_block = cout, oc_stride = 0, oc_off = 0, ytot = cout.                           */
typedef struct lldwt_conv_desc {
    int cin, cout, K, groups;
    int act;          /* LLDWT_ACT_* */
    int upsample2;
    int transposed;
    uint32_t tap_mask;
    int oc_block, oc_stride, oc_off, ytot;
    /* input channel placement (mirror of the output placement; used by the backward-data pass, whose input is the
     * gradient of a tensor written with oc_* placement): channel c is read at (c / ic_block)*ic_stride + ic_off +
     * c % ic_block of a tensor with xtot channels; ic_block == 0 means dense (xtot = cin).                    */
    int ic_block, ic_stride, ic_off, xtot;
    /* gradient epilogue: LLDWT_EPI_NONE, or multiply the result by act'(aux) with aux = the forward OUTPUT of the
     * layer whose pre-activation gradient is being formed (tanh: 1-aux^2, LeakyReLU: aux>0 ? 1 : 0.01).       */
    int epi;
} lldwt_conv_desc;
#define LLDWT_EPI_NONE 0
#define LLDWT_EPI_TANH_BWD 1
#define LLDWT_EPI_LRELU_BWD 2
/* Weights are pre-packed once per update into the MFMA A-operand lane order (csrc/conv_mfma.hip):
 * packed holds lldwt_conv_packed_floats(d) floats PER PLANE.  residual (optional): tensor laid out like y, added before
 * the activation (P_block_v2.py:53 "tmp + out_res").                                                       */
int64_t lldwt_conv_packed_floats(const lldwt_conv_desc* d);
int lldwt_conv_pack(const float* w, float* packed, const lldwt_conv_desc* d, int64_t planes, void* stream);
/* As lldwt_conv_pack; swap_hw != 0 packs the (kh,kw)-transposed kernel W^T (the horizontal lifting pass:
 * conv(x^T, W)^T == conv(x, W^T), replaces the torch.transpose calls of wavelet_forward_v2.py:32-51).      */
int lldwt_conv_pack_ex(const float* w, float* packed, const lldwt_conv_desc* d, int64_t planes, int swap_hw,
                       void* stream);
/* y = act( (conv(x) + bias) * epi(aux) + residual ).  transposed != 0 builds the operator from a weight in
 * ConvTranspose2d layout (planes, cin, cout/groups, K, K) with the taps flipped: that is ConvTranspose2d (stride 1)
 * itself and, applied to a forward Conv2d weight with cin/cout swapped, the BACKWARD-DATA pass of that conv.      */
int lldwt_conv2d(const float* x, float* y, const float* packed, const float* bias, const float* residual,
                 const float* aux, const lldwt_conv_desc* d, int64_t planes, int64_t batch, int64_t h, int64_t w_,
                 void* stream);
/* As lldwt_conv2d; additionally absmax_slots (planes,64), if not null, receives max |y| of each plane spread over 64
 * slots (the consumer takes their maximum): the activation scale of a following lldwt_conv3x3_f16x3, produced in the
 * epilogue instead of by a separate pass over y.                                                                  */
int lldwt_conv2d_absmax(const float* x, float* y, const float* packed, const float* bias, const float* residual,
                        const float* aux, float* absmax_slots, const lldwt_conv_desc* d, int64_t planes, int64_t batch,
                        int64_t h, int64_t w_, void* stream);
/* Backward-weights: dw (planes,cout,cin/groups,K,K) += sum over batch and pixels of dy[.,oc,p] * x[.,ic,p+tap]
 * (dead taps of tap_mask are skipped), dbias (planes,cout) += sum dy (optional).  dy is read through the OUTPUT
 * placement (oc_*), x through upsample2 / the input placement.  Accumulates with float atomics: zero dw/dbias first. */
int lldwt_conv2d_wgrad(const float* x, const float* dy, float* dw, float* dbias, const lldwt_conv_desc* d,
                       int64_t planes, int64_t batch, int64_t h, int64_t w_, void* stream);
/* As lldwt_conv2d_wgrad with dw += alpha * (...), dbias += alpha * (...); swap_hw != 0: the conv was applied with the
 * (kh,kw)-transposed kernel, so tap (ky,kx) of the contraction is accumulated into dw[..][kx][ky].        */
int lldwt_conv2d_wgrad_ex(const float* x, const float* dy, float* dw, float* dbias, const lldwt_conv_desc* d,
                          int64_t planes, int64_t batch, int64_t h, int64_t w_, float alpha, int swap_hw,
                          void* stream);
/* Backward-weights of a DENSE 3x3 conv (groups 1, no placement, no upsampling, no dead taps) on the fp16 matrix cores with
 * split-fp16 operands (fp32-level accuracy; csrc/conv_wgrad_f16x3.hip): dw (planes,cout,cin,3,3) += alpha * sum over batch
 * and pixels of dy[.,oc,p] * x[.,ic,p+tap], dbias (planes,cout) += alpha * sum dy (optional).  x (planes,batch,cin,h,w),
 * dy (planes,batch,cout,h,w), both 16-byte aligned, w % 4 == 0.  slots_ws: planes * 128 floats of scratch (the per-plane
 * max |x| and max |dy| the power-of-two operand scales come from).  Float atomics: zero dw / dbias first.  The training
 * path of the 243 -> 243 tree-context conv (LiftingBasedDWT_net.py:271-272); lldwt_conv2d_wgrad covers every other shape. */
int lldwt_conv3x3_wgrad_f16x3(const float* x, const float* dy, float* dw, float* dbias, float* slots_ws, int cin, int cout,
                              int64_t planes, int64_t batch, int64_t h, int64_t w_, float alpha, void* stream);
/* As lldwt_conv3x3_wgrad_f16x3; x_slots / dy_slots (planes,64), if not null, are the per-plane |max| slots of x / dy the caller
 * holds already (lldwt_absmax_slots, lldwt_conv2d_absmax): the pass over that tensor is skipped -- in a training step the forward
 * conv has measured x and the backward-data conv dy (autograd of LiftingBasedDWT_net.py:271-272).  slots_ws may be null when
 * both are given.                                                                                                           */
int lldwt_conv3x3_wgrad_f16x3_ex(const float* x, const float* dy, float* dw, float* dbias, float* slots_ws,
                                 const float* x_slots, const float* dy_slots, int cin, int cout, int64_t planes, int64_t batch,
                                 int64_t h, int64_t w_, float alpha, void* stream);
/* Backward-weights of the 16 -> 16 5x5 convs of a P/U block (conv2 / conv3 of graphs/layers/P_block_v2.py:40-55; autograd of
 * agents/liftingDWT_agent.py:97) on the fp16 matrix cores, split-fp16 operands (fp32-level accuracy):
 *   dw[p][oc][ic][ty][tx] += alpha * sum_{b,y,x} dy[p][b][oc][y][x] * x[p][b][ic][y+ty-2][x+tx-2],  dbias += alpha * sum dy.
 * x (planes, batch, 16, h, w) must be bounded by 1 in magnitude (the tanh outputs t1 / t2: its split uses the fixed scale
 * 2^14); dy any magnitude (one power-of-two scale per plane from its |max|).  w % 4 == 0, 16-byte aligned tensors.
 * slots_ws: planes * 64 floats of scratch.  swap_hw != 0: the (kh, kw) axes of dw are stored swapped (row passes).
 * lldwt_lift_step_bwd uses it for K == 5 tanh blocks; LLDWT_WGRAD16=f32 keeps the fp32-MFMA kernel of lldwt_conv2d_wgrad. */
int lldwt_wgrad16_f16x3(const float* x, const float* dy, float* dw, float* dbias, float* slots_ws, int64_t planes,
                        int64_t batch, int64_t h, int64_t w, float alpha, int swap_hw, void* stream);
/* dx = dy * act'(y) elementwise (y = forward output); act as in lldwt_conv_desc. */
int lldwt_act_bwd(const float* dy, const float* y, float* dx, int64_t n, int act, void* stream);
/* backward of the nearest-neighbour 2x upsampling: out (Z,C,h/2,w/2) = sum over each 2x2 block of g (Z,C,h,w). */
int lldwt_downsum2(const float* g, float* out, int64_t zc, int64_t h, int64_t w_, void* stream);
/* Dense 3x3 conv (groups 1, no placement, no upsampling) on the fp16 matrix cores with fp32-level accuracy
 * ("f16x3", csrc/conv_f16x3.hip): every fp32 operand is scaled by a power of two and split into hi + lo fp16 values
 * (|v - hi - lo| <= 2^-22 |v|), the product is hi*hi + hi*lo + lo*hi accumulated in fp32.  For the tree-context conv
 * 243 -> 243 (LiftingBasedDWT_net.py:271-272, :793-795).  The fp32 engine above stays the reference arithmetic.
 *   lldwt_conv_f16x3_pack : w (planes,cout,cin,3,3) -> packed (lldwt_conv_f16x3_packed_bytes(cin,cout) bytes per plane)
 *   lldwt_absmax_slots    : slots (planes,64) <- max |x| of each plane's n_per_plane values (the activation scale is
 *                           derived from it on the device; no host synchronisation)
 *   lldwt_conv3x3_f16x3   : y (planes,batch,cout,h,w) = act(conv3x3(x (planes,batch,cin,h,w)) + bias (planes,cout)) */
int64_t lldwt_conv_f16x3_packed_bytes(int cin, int cout);
int lldwt_conv_f16x3_pack(const float* w, void* packed, int cin, int cout, int64_t planes, void* stream);
int lldwt_absmax_slots(const float* x, int64_t planes, int64_t n_per_plane, float* slots, void* stream);
int lldwt_conv3x3_f16x3(const float* x, float* y, const void* packed, const float* bias, const float* slots, int cin,
                        int cout, int act, int64_t planes, int64_t batch, int64_t h, int64_t w_, void* stream);

/* Reduced-precision STORAGE of the tree-context tensor (BASELINE.json configs[4] "fp16"): the first tree
 * conv writes its output as fp16 (half the HBM bytes), multiplied by a power of two oscale[plane] chosen by the caller from
 * a bound (max |parent| x max row L1 norm + max |bias|) so that it cannot overflow; the second conv consumes it as the hi
 * half with no lo (two MFMA products instead of three): 2^-11 relative on the activations, fp32 accumulate.  Its own
 * tolerance (1e-2) and bench config; never the headline.
 *   lldwt_conv2d_f16out : lldwt_conv2d with y16 (planes,batch,ytot,h,w) fp16 = act(conv(x) + bias) * oscale[plane]
 *   lldwt_conv3x3_f16in : lldwt_conv3x3_f16x3 reading such a tensor (xscale = the oscale it was written with)        */
int lldwt_conv2d_f16out(const float* x, void* y16, const float* packed, const float* bias, const float* oscale,
                        const lldwt_conv_desc* d, int64_t planes, int64_t batch, int64_t h, int64_t w_, void* stream);
int lldwt_conv3x3_f16in(const void* x16, float* y, const void* packed, const float* bias, const float* xscale, int cin,
                        int cout, int act, int64_t planes, int64_t batch, int64_t h, int64_t w_, void* stream);

/* The tree-context PAIR in ONE launch (LiftingBasedDWT_net.py:271-272 / :793-795):
 *   y = act(conv3x3_{cmid->cout}(LeakyReLU(conv3x3_{3->cmid}(up2(parent)) + b1)) + b2)
 * parent (planes,batch,3,h/2,w/2) fp32, y (planes,batch,cout,h,w) fp32.  The first conv is evaluated on the fly by the
 * matrix cores for each workgroup's 10x34 halo patch (K = 27, split-fp16 like the second) and written straight into the
 * LDS image the second conv consumes: the cmid-channel tensor never exists in HBM.  Scales are per workgroup: the parent
 * patch by its own |max|, the intermediate by the bound max|patch| * max row L1 norm(w1) + max|b1| (a power of two either
 * way, so the choice does not change the result beyond the 2^-21 split error).
 *   packed1 = lldwt_plc_fused_pack1(w1 (planes,cmid,3,3,3), b1 (planes,cmid)), cmid <= 256
 *   packed2 = lldwt_conv_f16x3_pack(w2 (planes,cout,cmid,3,3));  bias2 (planes,cout) or NULL                              */
int64_t lldwt_plc_fused_pack1_bytes(int cmid);
/* 1 when this process packs and runs the split-fp16 3x3 conv kernels in the 16x16x32 MFMA shape (LLDWT_PLC_SHAPE=16, read when
 * the library loads), 0 for the default 32x32x16. */
int lldwt_plc_shape16(void);
int lldwt_plc_fused_pack1(const float* w1, const float* b1, void* packed1, int cmid, int64_t planes, void* stream);
int lldwt_plc_fused(const float* parent, float* y, const void* packed1, const void* packed2, const float* bias2, int cmid,
                    int cout, int act, int64_t planes, int64_t batch, int64_t h, int64_t w_, void* stream);

/* Same maths from the raw PyTorch-layout weights w, reference-order direct kernel (VALU); cross-checks the MFMA engine. */
int lldwt_conv2d_direct(const float* x, float* y, const float* w, const float* bias, const lldwt_conv_desc* d,
                        int64_t planes, int64_t batch, int64_t h, int64_t w_, void* stream);

/* GDN / inverse GDN (graphs/layers/gdn.py:77-92) with the NonNegativeParametrizer re-parametrisation
 * (utils/parametrizers.py:45-48) applied in-kernel to the raw beta (planes,C) / gamma (planes,C,C).         */
int lldwt_gdn(const float* x, float* y, const float* beta, const float* gamma, int64_t planes, int64_t batch,
              int C, int64_t hw, int inverse, float beta_min, void* stream);

/* Differentiable (training) composition of GDN: nrm = conv1x1(x*x, gamma', beta') on the conv engine, then
 * y = x * rsqrt(nrm) (inverse: x * sqrt(nrm)); these are its elementwise pieces and their backward.           */
int lldwt_ew_mul(const float* a, const float* b, float* out, int64_t n, float scale, void* stream);     /* out = scale*a*b */
int lldwt_gdn_apply(const float* x, const float* nrm, float* y, int64_t n, int inverse, void* stream);
int lldwt_gdn_apply_bwd(const float* x, const float* nrm, const float* g, float* dx, float* dn, int64_t n, int inverse,
                        void* stream);

/* LowerBound (utils/bound_ops.py:22-28) and NonNegativeParametrizer (utils/parametrizers.py:45-48), elementwise. */
int lldwt_lower_bound_fwd(const float* x, float* y, int64_t n, float bound, void* stream);
int lldwt_lower_bound_bwd(const float* x, const float* gy, float* gx, int64_t n, float bound, void* stream);
int lldwt_nonneg_param_fwd(const float* x, float* y, int64_t n, float minimum, void* stream);
int lldwt_nonneg_param_bwd(const float* x, const float* gy, float* gx, int64_t n, float minimum, void* stream);

/* The cgp stack with the folded context (as lldwt_cgp_rate_ctx) on the fp16 matrix cores, split-fp16 operands
 * (csrc/cgp_f16x3.hip): a register-resident chain per wave, layer l's accumulator tile is layer l+1's MFMA operand.  Built
 * for the reference's dimensions after the fold, 93 -> 162 -> 54 -> 18 -> 2 per subband (LiftingBasedDWT_net.py:282-289);
 * lldwt_cgp16_packed_bytes returns -1 for any other.  w_l: (planes, groups*c_{l+1}, c_l), b_l: (planes, groups*c_{l+1})
 * -- the folded first layer from the host (_fold_csc_into_cgp).  params: (planes, batch, 2*groups, h, w), sigma on the
 * even and mu on the odd channels, to be fed to lldwt_gauss_rate.                                                    */
int64_t lldwt_cgp16_packed_bytes(int c0, int c1, int c2, int c3, int groups);
int lldwt_cgp16_pack(const float* w0, const float* b0, const float* w1, const float* b1, const float* w2, const float* b2,
                     const float* w3, const float* b3, void* packed, int64_t planes, int c0, int c1, int c2, int c3,
                     int groups, void* stream);
int lldwt_cgp16_params(const float* plc, const float* xq, const void* packed, float* params, int64_t planes, int64_t batch,
                       int64_t h, int64_t w_, int groups, int K, uint32_t tap_mask, void* stream);
/* The training forward on the same register chain (always the fp32-accurate split-fp16 arithmetic, whatever lldwt_set_precision
 * says): lldwt_cgp16_params + the hidden activations after LeakyReLU in the layout of the unfused convs, h1 (Z, groups*162, hw),
 * h2 (Z, groups*54, hw), h3 (Z, groups*18, hw) -- what lldwt_cgp_bwd_split gates with and the 1x1 weight-gradient GEMMs read.
 * The bits follow from lldwt_gauss_rate on (x, params, noise).  Training forward of LiftingBasedDWT_net.py:282-289,357-365.   */
int lldwt_cgp16_params_train(const float* plc, const float* xq, const void* packed, float* params, float* h1, float* h2, float* h3,
                             int64_t planes, int64_t batch, int64_t h, int64_t w_, int groups, int K, uint32_t tap_mask,
                             void* stream);
/* Backward-data of the stack on the same register chain (replaces lldwt_cgp_bwd_split's fp32-MFMA kernel for the reference's
 * widths): dparams (Z, 2*groups, hw) = gradient at (sigma, mu); h1 / h2 / h3 = the activations lldwt_cgp16_params_train stored (their
 * SIGN gates LeakyReLU); -> d1 (Z, groups*162, hw), d2 (Z, groups*54, hw), d3 (Z, groups*18, hw) = gradients at the pre-activation
 * outputs (the dY of the 1x1 weight-gradient GEMMs) and the input gradient as dplc (Z, groups*81, hw) + dtaps (Z, groups*12, hw).
 * lldwt_cgp16_pack_bwd: the four forward weights (PyTorch layout, as lldwt_cgp16_pack) -> the transposed pack.                    */
int64_t lldwt_cgp16_bwd_packed_bytes(int c0, int c1, int c2, int c3, int groups);
int lldwt_cgp16_pack_bwd(const float* w0, const float* w1, const float* w2, const float* w3, void* packed, int64_t planes, int c0,
                         int c1, int c2, int c3, int groups, void* stream);
int lldwt_cgp16_bwd(const float* dparams, const float* h1, const float* h2, const float* h3, const void* packed_bwd, float* d1,
                    float* d2, float* d3, float* dplc, float* dtaps, int64_t planes, int64_t batch, int64_t hw, int groups,
                    void* stream);
/* Real entropy coding of a level with tree context + masked KxK context + cgp (the reference walks its pixels in raster
 * order with a CNN call on a crop each, graphs/models/LiftingBasedDWT_net.py:402-417,440-454,458-556): ONE anti-diagonal
 * wavefront step t = x + (K/2 + 1) * y, all planes / images / subbands / pixels of the step in one launch of the cgp
 * register chain (fp32-accurate f16x3 arithmetic whatever lldwt_set_precision says: encoder and decoder must agree bit for
 * bit).  yhat (Z, groups, h, w): the values decoded so far, 0 elsewhere (the masked taps only reach coded positions).
 * table63: the first 63 entries of the Gaussian scale table; CDF index = number of entries < max(sigma, 0.11).
 * idx / sym (Z, ntot, groups) int32 in wavefront order, this step's pixels at positions off .. off + n - 1 (rows ascending).
 *   encoder (y != null): sym = round(y - mu), yhat[pixel] = sym + mu, idx written.
 *   decoder (y == null): idx written, mu (Z, groups, n) kept for lldwt_wavefront_apply once the host has decoded sym.    */
int lldwt_cgp16_wavefront_step(const float* plc, float* yhat, const float* y, const void* packed, const float* table63,
                               int* idx, int* sym, float* mu, int64_t planes, int64_t batch, int64_t h, int64_t w,
                               int groups, int K, uint32_t tap_mask, int t, int64_t ntot, int64_t off, void* stream);
int lldwt_wavefront_apply(const int* sym, const float* mu, float* yhat, int64_t planes, int64_t batch, int64_t h, int64_t w,
                          int groups, int K, int t, int64_t ntot, int64_t off, void* stream);

/* ---------------------------------------------------------------------------------------------------------
 * Real entropy coding (reference: compress_ar / decompress_ar, LiftingBasedDWT_net.py:458-556, on
 * compressai.ans).  HOST functions (csrc/rans.hip): the range-ANS state machine is sequential byte work; the symbols
 * and CDF indexes it consumes are produced on the GPU by the wavefront scheduler of graphs/models/entropy_coding.py.
 *   lldwt_pmf_to_quantized_cdf : compressai `pmf_to_quantized_cdf` -- float pmf (n entries, tail mass last) -> n+1
 *                                cumulative counts at `precision` bits, every symbol with a non-zero frequency.
 *   lldwt_rans_encode          : BufferedRansEncoder.encode_with_indexes + flush (:466,502-505).  cdfs is an
 *                                (ncdf, cdf_stride) int32 table, cdf_sizes[i] = used entries of row i (= pmf length + 2),
 *                                offsets[i] = symbol value of slot 0.  Returns the number of bytes written to `out`
 *                                (a multiple of 4, <= out_cap) or a negative error code.  Out-of-table symbols are
 *                                escaped through the last slot and 4-bit bypass digits.
 *   lldwt_rans_decoder_new / _decode / _free : RansDecoder.set_stream / decode_stream (:516-517,540-546): an opaque
 *                                handle owning a copy of the stream; decode pops n symbols in coding order.        */
int lldwt_pmf_to_quantized_cdf(const float* pmf, int n, int precision, uint32_t* cdf);
int64_t lldwt_rans_encode(const int32_t* symbols, const int32_t* indexes, int64_t n, const int32_t* cdfs, int32_t ncdf,
                          int32_t cdf_stride, const int32_t* cdf_sizes, const int32_t* offsets, uint8_t* out,
                          int64_t out_cap);
void* lldwt_rans_decoder_new(const uint8_t* stream, int64_t nbytes);
int lldwt_rans_decode(void* dec, const int32_t* indexes, int64_t n, const int32_t* cdfs, int32_t ncdf, int32_t cdf_stride,
                      const int32_t* cdf_sizes, const int32_t* offsets, int32_t* symbols);
/* The same for nstreams decoders in one call (one wavefront step of every (plane, image) stream): stream k reads its n
 * indexes at indexes + k * stride and writes symbols + k * stride; one table set for all. */
int lldwt_rans_decode_multi(void* const* decs, int64_t nstreams, const int32_t* indexes, int64_t n, int64_t stride,
                            const int32_t* cdfs, int32_t ncdf, int32_t cdf_stride, const int32_t* cdf_sizes,
                            const int32_t* offsets, int32_t* symbols);
void lldwt_rans_decoder_free(void* dec);

/* ---------------------------------------------------------------------------------------------------------
 * Rate estimation, fused quantise + likelihood + both LowerBounds + -log2 + sum of bits.
 * Gaussian (compressai GaussianConditional.forward as called at LiftingBasedDWT_net.py:334,345,364,832):
 *   v = mode==0 ? round(x - mu) + mu : x + noise       (noise: U(-.5,.5) tensor or NULL -> eval)
 *   p = Phi((.5-|v-mu|)/max(sigma,0.11)) - Phi((-.5-|v-mu|)/max(sigma,0.11)),  bits = -log2(max(p,1e-9))
 * params: (Z, 2C, h, w) with sigma = channel 2c, mu = channel 2c+1 (:332-333).  x, bits, qout: (Z,C,h,w).
 * qout (optional) receives v (what onlyEZWT hands to the decoder, :832-834).  bits (optional).
 * bit_sum (optional): one device double, accumulated atomically (caller zeroes it).                       */
int lldwt_gauss_rate(const float* x, const float* params, const float* noise, float* bits, float* qout,
                     double* bit_sum, int64_t Z, int C, int64_t hw, void* stream);
/* Backward of lldwt_gauss_rate: gbits = dL/dbits -> dx (Z,C,hw) and dparams (Z,2C,hw) = (dsigma, dmu) interleaved;
 * closed form, both LowerBound pass-through rules (utils/bound_ops.py:26-28) applied.                      */
int lldwt_gauss_rate_bwd(const float* x, const float* params, const float* noise, const float* gbits, float* dx,
                         float* dparams, int64_t Z, int C, int64_t hw, void* stream);
/* quantize(x, mode, means=None): round(x) or x + noise (LiftingBasedDWT_net.py:330,341,352). */
int lldwt_quantize(const float* x, const float* noise, float* q, int64_t n, void* stream);

/* Fused "cgp" parameter network + Gaussian rate of DWTConditioned2EntropyLayerZTsepSubbands
 * (LiftingBasedDWT_net.py:282-289 grouped 1x1 convs c0 -> c1 -> c2 -> c3 -> 2 with LeakyReLU, :361-365 rate).
 * cat: (Z, groups*c0, hw) = the interleaved (plc_g, csc_g) tensor of :357-359; x, noise, bits: (Z, groups, hw);
 * params_out (optional): (Z, 2*groups, hw) receives (sigma, mu) as the reference's out_xo_qnt_mu_sigma.
 * lldwt_cgp_pack: the four conv weights (planes, groups*c_{l+1}, c_l) / biases (planes, groups*c_{l+1}) in PyTorch
 * layout -> packed (planes, lldwt_cgp_packed_floats) in MFMA A-operand lane order.                       */
int64_t lldwt_cgp_packed_floats(int c0, int c1, int c2, int c3, int groups);
int lldwt_cgp_pack(const float* w0, const float* b0, const float* w1, const float* b1, const float* w2,
                   const float* b2, const float* w3, const float* b3, float* packed, int64_t planes, int c0, int c1,
                   int c2, int c3, int groups, void* stream);
int lldwt_cgp_rate(const float* cat, const float* x, const float* noise, const float* packed, float* bits,
                   float* params_out, double* bit_sum, int64_t planes, int64_t batch, int64_t hw, int c0, int c1,
                   int c2, int c3, int groups, void* stream);
/* lldwt_cgp_rate with the masked context conv folded into its first layer (eval).  The csc conv (MaskedConv2d 5x5 type A,
 * LiftingBasedDWT_net.py:275-277,353) feeds cgp layer 0 with no nonlinearity in between, so the host folds
 * W0[:, csc half] . Wcsc (+ bias) into `ntaps` extra input columns of layer 0 -- packed with lldwt_cgp_pack for
 * c0 = cplc + ntaps -- and the kernel gathers those inputs itself: the live taps (tap_mask, KxK) of the quantised
 * subband xq (Z, groups, h, w), zero outside the image.  plc: (Z, groups*cplc, h, w), the tree-context conv's output. */
int lldwt_cgp_rate_ctx(const float* plc, const float* xq, const float* x, const float* noise, const float* packed,
                       float* bits, float* params_out, double* bit_sum, int64_t planes, int64_t batch, int64_t h,
                       int64_t w_, int cplc, int K, uint32_t tap_mask, int c1, int c2, int c3, int groups, void* stream);
/* Training variants of the fused stack.  lldwt_cgp_rate_train: as lldwt_cgp_rate (no bit_sum), and also writes
 * params_out (sigma, mu: (Z, 2*groups, hw)) and the hidden activations after LeakyReLU, h1 (Z, groups*c1, hw),
 * h2 (Z, groups*c2, hw), h3 (Z, groups*c3, hw) -- the layout the unfused 1x1 convs would produce.
 * lldwt_cgp_bwd: backward-data of the four layers in one launch.  dparams (Z, 2*groups, hw) = gradient at (sigma, mu)
 * (lldwt_gauss_rate_bwd) -> d3, d2, d1 = gradients at the PRE-activation outputs of layers 3, 2, 1 (shapes of
 * h3, h2, h1; what lldwt_conv2d_wgrad needs as dy) and dcat (Z, groups*c0, hw).  packed_bwd: the forward weights
 * w0..w3 (PyTorch layout) transposed into MFMA A-operand order by lldwt_cgp_pack_bwd
 * (planes, lldwt_cgp_bwd_packed_floats).  Replaces autograd through LiftingBasedDWT_net.py:282-289,360-365. */
int lldwt_cgp_rate_train(const float* cat, const float* x, const float* noise, const float* packed, float* bits,
                         float* params_out, float* h1, float* h2, float* h3, int64_t planes, int64_t batch, int64_t hw,
                         int c0, int c1, int c2, int c3, int groups, void* stream);
int64_t lldwt_cgp_bwd_packed_floats(int c0, int c1, int c2, int c3, int groups);
int lldwt_cgp_pack_bwd(const float* w0, const float* w1, const float* w2, const float* w3, float* packed_bwd,
                       int64_t planes, int c0, int c1, int c2, int c3, int groups, void* stream);
int lldwt_cgp_bwd(const float* dparams, const float* h1, const float* h2, const float* h3, const float* packed_bwd,
                  float* d1, float* d2, float* d3, float* dcat, int64_t planes, int64_t batch, int64_t hw, int c0, int c1,
                  int c2, int c3, int groups, void* stream);
/* The training stack WITHOUT the concatenated [tree-context channels | gathered taps] input tensor (1.75 GB at the level-0 shape of
 * configs[2], written by a torch.cat and read back; its gradient took the same way back):
 *   lldwt_cgp_rate_train_ctx : lldwt_cgp_rate_train reading its input as lldwt_cgp_rate_ctx does -- plc (Z, groups*cplc, h, w)
 *                              and the live taps of the quantised subband xq (Z, groups, h, w), gathered in the kernel;
 *   lldwt_cgp_bwd_split      : lldwt_cgp_bwd with the input gradient as two tensors, dplc (Z, groups*cplc, hw) and
 *                              dtaps (Z, groups*ntaps, hw) (tap order = ascending live taps of the mask);
 *   lldwt_wgrad1x1_split     : weight gradient of a grouped 1x1 conv whose input rows are [xa rows | xb rows] per group
 *                              (layer 0: xa = plc, xb = the gathered taps), dw (planes, cout, ca + cb), dbias (planes, cout).
 * Autograd of LiftingBasedDWT_net.py:282-289,353-365.                                                                       */
int lldwt_cgp_rate_train_ctx(const float* plc, const float* xq, const float* x, const float* noise, const float* packed,
                             float* bits, float* params_out, float* h1, float* h2, float* h3, int64_t planes, int64_t batch,
                             int64_t h, int64_t w_, int cplc, int K, uint32_t tap_mask, int c1, int c2, int c3, int groups,
                             void* stream);
int lldwt_cgp_bwd_split(const float* dparams, const float* h1, const float* h2, const float* h3, const float* packed_bwd,
                        float* d1, float* d2, float* d3, float* dplc, float* dtaps, int64_t planes, int64_t batch, int64_t hw,
                        int cplc, int ntaps, int c1, int c2, int c3, int groups, void* stream);
int lldwt_wgrad1x1_split(const float* xa, const float* xb, const float* dy, float* dw, float* dbias, int64_t planes,
                         int64_t batch, int64_t hw, int ca, int cb, int cout, int groups, void* stream);

/* Factorized (compressai EntropyBottleneck.forward, call sites LiftingBasedDWT_net.py:225,229,815,818):
 * per channel c of plane p: 5 tiny matrices softplus(_matrix{i}) (1x3,3x3,3x3,3x3,3x1), biases, tanh(_factor).
 * eb: packed per (plane,channel) block of LLDWT_EB_FLOATS floats =
 *   [m0(3) b0(3) f0(3) m1(9) b1(3) f1(3) m2(9) b2(3) f2(3) m3(9) b3(3) f3(3) m4(3) b4(1) median(1)] raw values. */
#define LLDWT_EB_FLOATS 59
int lldwt_factorized_rate(const float* x, const float* eb, const float* noise, float* bits, float* qout,
                          double* bit_sum, int64_t planes, int64_t batch, int C, int64_t hw, void* stream);

/* Eval with the per-offset table precomputed: lldwt_factorized_table writes, for every (plane, channel), the bits of
 * median + o for the integer offsets o = -127 .. 127 (table: planes * C rows of 256 floats) -- the values the kernel of
 * lldwt_factorized_rate builds per workgroup, by the same code; it depends on the parameters only (compressai's update() keeps
 * its quantised CDFs the same way, entropy_models.py:206-240).  lldwt_factorized_rate_tab is the eval form of
 * lldwt_factorized_rate reading that table: identical results, no table build in front of every workgroup's stream. */
int lldwt_factorized_table(const float* eb, float* table, int64_t planes, int C, void* stream);
int lldwt_factorized_rate_tab(const float* x, const float* eb, const float* table, float* bits, float* qout, double* bit_sum,
                              int64_t planes, int64_t batch, int C, int64_t hw, void* stream);

/* Backward of lldwt_factorized_rate (training, v = x + noise): dx (Z,C,hw) and deb (planes,C,59) += gradient wrt the RAW
 * packed parameters (softplus' / tanh' applied; median slot stays 0).  deb is accumulated with atomics: zero it first. */
int lldwt_factorized_rate_bwd(const float* x, const float* eb, const float* noise, const float* gbits, float* dx,
                              float* deb, int64_t planes, int64_t batch, int C, int64_t hw, void* stream);

/* sum((a-b)^2) and sum(x) into a double (graphs/losses/rate_dist.py:36-41). */
int lldwt_sq_err_sum(const float* a, const float* b, int64_t n, double* out, void* stream);
int lldwt_sum(const float* x, int64_t n, double* out, void* stream);
/* out = alpha*a + beta*b elementwise (b optional): gradient glue of the loss (d mse = 2(xhat-x)/N). */
int lldwt_axpby(const float* a, const float* b, float* out, int64_t n, float alpha, float beta, void* stream);

/* ---------------------------------------------------------------------------------------------------------
 * Fixed CDF 9/7 (bior4.4) DWT, periodization (DWTPytorchWaveletsLayer, lifting_dwt_nets.py:228-231,250,274).
 * Same tensor conventions as lldwt_lifting_forward/inverse.                                                */
/* Level inputs SHORTER than the 10-tap filter (2, 4, 6 or 8 samples, e.g. a 64 x 64 image at 4 levels).  The reference's
 * pytorch_wavelets afb1d / sfb1d fold the linear convolution back once there; these kernels compute the exact periodic
 * transform (= PyWavelets 'periodization', perfect reconstruction).  The two forms differ on such levels and agree from
 * 10 samples up (every BASELINE config: smallest level input 32).  periodic = 0 (default): lldwt_cdf97_* return
 * LLDWT_EINVAL for such a call instead of silently differing from the reference; 1: compute the periodic form.        */
int lldwt_set_cdf97_short_levels(int periodic);
int64_t lldwt_cdf97_ws_bytes(int64_t Z, int64_t H, int64_t W);
int lldwt_cdf97_forward(const float* x, float* ll, float* const* yh, int64_t Z, int64_t H, int64_t W, int levels,
                        void* ws, int64_t ws_bytes, void* stream);
int lldwt_cdf97_inverse(const float* ll, const float* const* yh, float* x, int64_t Z, int64_t H, int64_t W,
                        int levels, void* ws, int64_t ws_bytes, void* stream);
/* Adjoints for training (bior4.4 is not orthogonal, so the backward pass is NOT the other transform):
 * lldwt_cdf97_inverse_ex(adj=1) maps (g_ll, g_yh) -> g_x   = adjoint of lldwt_cdf97_forward (backward of the analysis);
 * lldwt_cdf97_forward_ex(adj=1) maps g_x -> (g_ll, g_yh)   = adjoint of lldwt_cdf97_inverse (backward of the synthesis). */
int lldwt_cdf97_forward_ex(const float* x, float* ll, float* const* yh, int64_t Z, int64_t H, int64_t W, int levels,
                           int adj, void* ws, int64_t ws_bytes, void* stream);
int lldwt_cdf97_inverse_ex(const float* ll, const float* const* yh, float* x, int64_t Z, int64_t H, int64_t W,
                           int levels, int adj, void* ws, int64_t ws_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* LLDWT_H */
