/*
 * premvos_hip.h -- C-ABI of libpremvos_hip.so: the MI355X (gfx950) kernels behind the
 * PReMVOS per-frame dense hot path (PWC-Net flow, proposal_net, refinement_net forward).
 *
 * Boundary rules:
 *   - extern "C", raw DEVICE pointers + explicit sizes/strides, caller-owned buffers;
 *   - every entry point is asynchronous on the hipStream_t handed in (void* here so the
 *     header needs no HIP include), no hidden synchronisation, no global state;
 *   - returns 0 on success, <0 on argument / launch error (message: premvos_last_error()),
 *     never exit()s -- unlike the reference FFI, which "returns 1 always" and exit(-1)s on
 *     a failed launch (correlation_package/src/corr_cuda.c:80, corr_cuda_kernel.cu:49-54).
 *
 * Layout: activations are NHWC ("pixel major") fp32 with an explicit pixel stride `ps`
 * (in elements) so a tensor may be a channel slice [coff, coff+C) of a wider concat
 * buffer: pass `base + coff` and the buffer's `ps`.  Slices used as conv INPUT need
 * (pointer % 16 B == 0), ps % 4 == 0 and roundup(C,4) readable finite floats per pixel.
 *
 * The reference interface each function replaces is cited as file:line relative to
 * /root/reference/code/.
 */
#ifndef PREMVOS_HIP_H
#define PREMVOS_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PREMVOS_OK 0
#define PREMVOS_EINVAL (-1)
#define PREMVOS_ELAUNCH (-2)
#define PREMVOS_EUNSUPPORTED (-3) /* valid input outside what the entry point covers (premvos_jpeg_*: the caller falls back) */

/* activation enum for the fused conv epilogue */
#define PREMVOS_ACT_NONE 0
#define PREMVOS_ACT_RELU 1
#define PREMVOS_ACT_LEAKY 2 /* x > 0 ? x : slope * x   (nn.LeakyReLU(0.1), PWCNet.py:28) */
#define PREMVOS_ACT_SIGMOID 3 /* 1/(1+exp(-x))   (mask head, proposal_net/train.py:305) */
/* OR-ed into `act` of premvos_dwconv3x3_f32: store the result in the resident split layout "S8" of the bf16x3 mode -- every group
 * of EIGHT channels of a pixel is the 32 bytes {hi(8 x bf16), lo(8 x bf16)}, x = hi + lo (hi = bf16(x) round-to-nearest-even,
 * lo = bf16(x - hi)), in place of its eight floats (pixel stride a multiple of 8 floats' worth, channel window 32-byte aligned)
 * -- the operand form of premvos_conv_bf16x3_s8_f32.  (Round 3's {hi4, lo4} groups, 0x100, and their consumer are gone.) */
#define PREMVOS_ACT_SPLIT8_BF16 0x200

/* arithmetic of the dense-conv MFMA pipe (activations and outputs are fp32 in HBM in every mode) */
#define PREMVOS_PREC_F32 0    /* v_mfma_f32_32x32x2_f32: exact fp32 products (the parity / default bench mode)        */
#define PREMVOS_PREC_BF16 1   /* v_mfma_f32_32x32x16_bf16 on bf16(a), bf16(b), fp32 accumulate                        */
#define PREMVOS_PREC_BF16X3 3 /* split-bf16: hi*hi + hi*lo + lo*hi, fp32 accumulate (~2^-17 relative per product)    */

/* output scatter mode of the conv epilogue */
#define PREMVOS_OUT_NHWC 0
#define PREMVOS_OUT_PIXSHUF2 1 /* column j = phase*cout_ps + co -> out[2y+phase/2][2x+phase%2][co] */

const char* premvos_last_error(void);
int premvos_abi_version(void);

/* ------------------------------------------------------------------------------------------
 * Dense convolution as implicit GEMM on the fp32 MFMA pipe (v_mfma_f32_32x32x2_f32),
 * fused bias (+ folded BatchNorm) + residual add + activation, writing into a channel
 * slice of a (concat) buffer.
 *
 * Replaces every dense conv call the reference makes through its frameworks:
 *   optical_flow_net-PWC-Net/models/PWCNet.py:24-34 (conv / predict_flow / deconv builders),
 *   proposal_net/basemodel.py:51-99 (Conv2D+BNReLU bottlenecks), proposal_net/model.py:30-51,
 *   refinement_net/network/deeplab/core/xception.py:154-178 (pointwise halves), model.py:383-433.
 *
 * Weights are pre-packed by the host (premvos_amd/ops.py: pack_conv) as a row-major matrix
 * wgt[cout_pad][k_pad], k = (kh*KW + kw)*cin_pad + c, cin_pad = roundup(cin,4),
 * k_pad = roundup(KH*KW*cin_pad, 16), cout_pad = roundup(cout, 32); zero filled padding.
 * ---------------------------------------------------------------------------------------- */
typedef struct premvos_conv_desc {
  const float* in;      /* NHWC slice base */
  const float* wgt;     /* packed weights  */
  const float* bias;    /* [cout_pad] or NULL */
  const float* res;     /* residual NHWC slice (same N,Ho,Wo,cout) or NULL; added before act */
  float* out;           /* NHWC slice base */
  int32_t n, h, w, cin; /* input dims      */
  int32_t in_ps;        /* input pixel stride (elements) */
  int32_t ho, wo, cout; /* output dims     */
  int32_t out_ps;       /* output pixel stride */
  int32_t res_ps;       /* residual pixel stride */
  int32_t kh, kw;       /* kernel          */
  int32_t sh, sw;       /* stride          */
  int32_t dh, dw;       /* dilation        */
  int32_t pt, pl;       /* pad top / left (bottom/right follow from ho,wo: asymmetric pads ok) */
  int32_t cin_pad, k_pad, cout_pad;
  int32_t act;          /* PREMVOS_ACT_*   */
  float slope;          /* leaky slope     */
  int32_t out_mode;     /* PREMVOS_OUT_*   */
  int32_t cout_ps;      /* PIXSHUF2: channels per phase (cout == 4*cout_ps); out dims are 2ho x 2wo */
  int32_t tile_hint;    /* 0 = auto; (BM<<16)|BN forces an MFMA tile config; 1 forces the direct kernel for cout <= 2;  */
                        /* 2 = Winograd F(2x2,3x3) (csrc/conv_wino_f32.hip; needs wgt_wino), 16 component slabs in the  */
                        /* workspace + an output-transform launch; 3 = the same algebra in ONE kernel without workspace */
                        /* (a workgroup walks all 16 components of its block; output transform from registers);         */
                        /* 4 = Winograd F(4x4,3x3) (needs wgt_wino4 + workspace): 4x fewer multiplies, for K-rich layers  */
                        /*     (3x3 / stride 1; atrous layers with dilation d = pad run as d x d interleaved sub-lattices)      */
                        /* 5 = short-K pointwise layers (1x1, stride 1, cin = 64 | 128 = k_pad, cout % 128 == 0, <= 512):   */
                        /* persistent workgroups, weights resident in LDS (csrc/conv_stream_f32.hip); the implicit GEMM's sums */
                        /* 6 = pointwise layers (1x1, no padding, any stride, k_pad >= 32, cout % 4 == 0, 16-byte aligned pixels): */
                        /* 128 x 128 tiles with both operands staged by LDS-DMA (csrc/conv_pwdma_f32.hip); the implicit GEMM's sums */
  int32_t split_k;      /* 0 = auto, <0 = never, >0 = force this many k-slices */
  float* workspace;     /* split-K partial slabs (may be NULL: then never split) */
  int64_t workspace_bytes;
  int32_t precision;    /* PREMVOS_PREC_*; bf16 modes: wgt = bf16 hi [cout_pad][k_pad], k_pad % 32 == 0 */
  int32_t stage_k;      /* fp32 path: k depth of an LDS stage, 16 or 32 (0 = library default); with tile_hint == 2: 64 = 64 */
                        /* instead of 128 tile rows per workgroup; with tile_hint == 3: the block (2x2-tile rows x      */
                        /* couts / waves / stage depth) of a workgroup: 0, 2..6 (table in csrc/conv_wino_f32.hip)       */
                        /* with tile_hint == 0 / 1 on a 1-2 channel head: 1 = keep the per-pixel form (csrc/conv_smalln_f32.hip) */
  const void* wgt_lo;   /* BF16X3: bf16 low parts (w - float(hi)), same shape as wgt */
  int32_t tail_m_tiles; /* fp32 path, unsplit layers: the last tail_m_tiles rows of BM-tall output tiles are computed  */
  int32_t tail_split_k; /* as tail_split_k k-slices + fixed-order reduce (fills a partly empty last wave); 0 = off    */
  const float* wgt_wino; /* optional (3x3 / stride 1 / dilation 1 / fp32 / cout % 4 == 0): the 16 Winograd F(2x2,3x3) filter  */
                         /* transforms U[c] = (G g G^T)[c/4][c%4], each packed [cout_pad][roundup(cin_pad,16)] with k = cin;   */
                         /* used when tile_hint == 2 (needs premvos_conv2d_workspace_bytes() of workspace); NULL = not packed */
  const float* wgt_wino4; /* optional, same layers: the 36 Winograd F(4x4,3x3) filter transforms U[6i+j] = (G g G^T)[i][j], packed  */
                          /* like wgt_wino; used when tile_hint == 4 (csrc/conv_wino4_f32.hip: input transform, 36 batched GEMMs,  */
                          /* output transform around workspace slabs; stage_k = GEMM block: +64 = 64 instead of 128 tile rows per  */
                          /* workgroup, +16 = 16- instead of 32-deep stages)                                                        */
} premvos_conv_desc;

int premvos_conv2d_f32(const premvos_conv_desc* d, void* stream);
/* bytes of `workspace` the descriptor needs for its (auto or forced) split-K plan; 0 = runs unsplit.  Layers with
 * too few output tiles to fill the chip (coarse PWC levels, batch-1 feature maps) are cut along K into slabs that
 * a second kernel sums in a fixed order before the fused epilogue (deterministic, unlike atomics). */
int64_t premvos_conv2d_workspace_bytes(const premvos_conv_desc* d);

/* Winograd F(4x4,3x3) with a KEPT input-transform slab, for DenseNet blocks: PWC-Net's flow estimators
 * (PWCNet.py:201-264, `x = torch.cat((self.convL_i(x), x), 1)`) feed layer i the concat of everything layers 0 ... i-1
 * produced, so premvos_conv2d_f32 with tile_hint 4 transforms the same channels again in every layer of a level.  Here the
 * caller owns ONE slab per concat buffer, V[36][tiles][v_pitch] floats (tiles = n * ceil(ho/4) * ceil(wo/4); slab channel =
 * channel of the concat buffer), and every layer adds only what is new:
 *   d        the layer's descriptor as for premvos_conv2d_f32 (wgt_wino4 packed; `in` = first channel of its input window;
 *            stage_k = GEMM block as with tile_hint 4); d->workspace holds the M slab: 36 * tiles * roundup(cout, 64 or 128) floats
 *   v_c0     slab channel of the window's first channel; the GEMM reads slab channels [v_c0, v_c0 + Kp), Kp = roundup(cin_pad, 16)
 *   t_cn     channels [0, t_cn) of the window are transformed by this call (0: everything is in the slab already); channels
 *            >= cin_pad are written as zeros, so t_cn = Kp on the first layer of a level zero-fills the K padding for all
 *            (every window of a level ends at the same channel)
 * v_pitch, v_c0, t_cn: multiples of 4, v_c0 + Kp <= v_pitch, t_cn <= Kp.  Same V values, same GEMM, same output transform as the
 * self-contained form: bit-identical results (tests/test_gpu_conv_wino.py).  PREMVOS_EINVAL if the layer is not an F(4x4) layer. */
int premvos_conv_wino4_slab_f32(const premvos_conv_desc* d, float* vslab, int64_t vslab_bytes, int32_t v_pitch, int32_t v_c0, int32_t t_cn,
                                void* stream);

/* ------------------------------------------------------------------------------------------
 * PWC-Net cost volume, the only first-party native kernel of the reference:
 *   corr_cuda_forward  correlation_package/src/corr_cuda.c:7-82
 *   blob_rearrange_kernel2 / CorrelateData  src/corr_cuda_kernel.cu:18-37, 59-127
 * for the instantiation PWC-Net uses (pad = max_displacement = md, kernel_size = 1,
 * stride1 = stride2 = 1, multiply; PWCNet.py:69):
 *   out[y][x][(dy+md)*(2md+1) + (dx+md)] = (1/C) * sum_c f1[y][x][c] * f2[y+dy][x+dx][c]
 * (zero outside), with the LeakyReLU the reference applies right after (PWCNet.py:198)
 * fused when `slope` != 1, written into channels [0, (2md+1)^2) of `out`; if `copy_f1`
 * the f1 features are also copied to channels [(2md+1)^2, +C) (the torch.cat of
 * PWCNet.py:213).  No NCHW->NHWC rearrange pass and no zeroed scratch: inputs ARE NHWC.
 * ---------------------------------------------------------------------------------------- */
int premvos_corr_fwd_f32(const float* f1, int32_t f1_ps, const float* f2, int32_t f2_ps, float* out,
                         int32_t out_ps, int32_t n, int32_t h, int32_t w, int32_t c, int32_t md,
                         float slope, int32_t copy_f1, void* stream);

/* General form of the reference op signature (pad_size, kernel_size, max_displacement, stride1,
 * stride2, corr_type_multiply) on NCHW tensors -- the exact argument list of corr_cuda_forward
 * (corr_cuda.h:1-11) minus the THCudaTensor scratch (rbot1/rbot2 are not needed).  Supports the
 * multiply type only (the reference's subtract path is never instantiated by PWCNet.py).
 * out must hold n * D*D * oh * ow floats (shape math of corr_cuda.c:23-45). */
int premvos_corr_nchw_fwd_f32(const float* in1, const float* in2, float* out, int32_t n, int32_t c,
                              int32_t h, int32_t w, int32_t pad_size, int32_t kernel_size,
                              int32_t max_displacement, int32_t stride1, int32_t stride2,
                              int32_t corr_type_multiply, void* stream);

/* ------------------------------------------------------------------------------------------
 * Backward bilinear warp of image-2 features by (flow * flow_scale) times the validity mask
 * (grid_sample(ones) >= 0.9999): PWCDCNet.warp, models/PWCNet.py:140-176, torch-0.2 grid_sample
 * (bilinear, zeros padding, align_corners=True).  flow is NHWC with 2 channels (u,v).
 * ---------------------------------------------------------------------------------------- */
int premvos_warp_fwd_f32(const float* x, int32_t x_ps, const float* flow, int32_t flow_ps, float flow_scale,
                         float* out, int32_t out_ps, int32_t n, int32_t h, int32_t w, int32_t c,
                         void* stream);

/* The two calls above fused for pyramid levels 5..2 (PWCNet.py:207-208, 220-221, ...: warp5 = self.warp(c25, up_flow5*0.625);
 * corr5 = self.corr(c15, warp5)): the image-2 features x2 are warped while the cost-volume kernel stages its f2 tile, so
 * the warped map is never written to HBM.  Bit-identical to premvos_warp_fwd_f32 followed by premvos_corr_fwd_f32.
 * md must be 4. */
int premvos_warp_corr_fwd_f32(const float* f1, int32_t f1_ps, const float* x2, int32_t x2_ps, const float* flow,
                              int32_t flow_ps, float flow_scale, float* out, int32_t out_ps, int32_t n, int32_t h,
                              int32_t w, int32_t c, int32_t md, float slope, int32_t copy_f1, void* stream);

/* layout plumbing at the stage edges (the reference nets take/return NCHW) */
int premvos_nchw_to_nhwc_f32(const float* in, float* out, int32_t out_ps, int32_t n, int32_t c, int32_t h,
                             int32_t w, void* stream);
int premvos_nhwc_to_nchw_f32(const float* in, int32_t in_ps, float* out, int32_t n, int32_t c, int32_t h,
                             int32_t w, void* stream);

/* ------------------------------------------------------------------------------------------
 * Flow-stage host pre/post processing moved on-device (script_pwc_multi.py:33-70):
 *  pre : `batch` pairs of HWC uint8 RGB frames (im1/im2: [batch][h][w][3]) -> cv2.resize(INTER_LINEAR,
 *        uint8 fixed point) to (w_,h_) -> BGR, /255 -> NHWC fp32 [2*batch][h_][w_][4] (4th channel 0;
 *        images [0,batch) = first frames, [batch,2*batch) = second frames)             (:38-56)
 *  post: flow2 NHWC [batch][h4][w4][flow_ps] (u,v in ch 0,1) -> x20 -> cv2.resize(float INTER_LINEAR) to
 *        (w,h) -> u*=w/w_, v*=h/h_ -> [batch][h][w][2] (the .flo payload layout)        (:59-68)
 * ---------------------------------------------------------------------------------------- */
int premvos_flow_preprocess_u8(const uint8_t* im1, const uint8_t* im2, int32_t batch, int32_t h, int32_t w,
                               float* out, int32_t h_, int32_t w_, void* stream);
int premvos_flow_postprocess_f32(const float* flow2, int32_t flow_ps, int32_t batch, int32_t h4, int32_t w4,
                                 float* out, int32_t h, int32_t w, int32_t h_, int32_t w_, void* stream);

/* ==========================================================================================
 * proposal_net (class-agnostic ResNet-101-C4 Faster R-CNN, `train.py --forward`); paths relative to
 * code/proposal_net/.  The dense convs (backbone, RPN head, conv5, FC heads as 1x1 convs) go through
 * premvos_conv2d_f32 with the frozen BatchNorm folded into weights+bias.
 * ========================================================================================== */

/* eval.py:76-77 CustomResize (cv2.resize INTER_LINEAR on the uint8 BGR frame) + basemodel.py:12-26
 * (x/255 - mean)/std in BGR order -> NHWC fp32 [batch][nh][nw][4] (4th channel 0).  src_is_rgb != 0: the
 * source frame is RGB (as the flow / refinement stages read it) and is channel-swapped on load. */
int premvos_proposal_preprocess_u8(const uint8_t* img_bgr, int32_t batch, int32_t h, int32_t w, float* out,
                                   int32_t nh, int32_t nw, int32_t src_is_rgb, void* stream);

/* MaxPooling('pool0', shape=3, stride=2) after tf.pad [0,1] (basemodel.py:81-82); generic k/stride/pad. */
int premvos_maxpool_f32(const float* in, int32_t in_ps, int32_t n, int32_t h, int32_t w, int32_t c, float* out,
                        int32_t out_ps, int32_t ho, int32_t wo, int32_t k, int32_t stride, int32_t pt, int32_t pl,
                        float pad_value, void* stream);

/* generate_rpn_proposals (model.py:169-217) fused with the anchor field (data.py:34-74, train.py:92-105) and
 * decode_bbox_target (model.py:113-139): per image, top-`pre_nms_topk` logits (ties -> lower index), clip to
 * the image, drop w/h <= min_size, greedy NMS (IoU > nms_thresh suppressed, TF IoU) keeping <= post_nms_topk
 * in descending-score order.  rpn is the NHWC output of the fused RPN 1x1 heads: logits at channel
 * [logit_off, +na), deltas at box_off + a*4 + {tx,ty,tw,th}.  out_idx are flat anchor indices
 * (y*fw + x)*na + a -- the 'bit-exact proposal indices' of the north star.  Unused slots: zeros / -1. */
int premvos_rpn_proposals_f32(const float* rpn, int32_t ps, int32_t n, int32_t fh, int32_t fw, int32_t na,
                              int32_t logit_off, int32_t box_off, const float* cell_anchors, float stride,
                              float img_h, float img_w, int32_t pre_nms_topk, int32_t post_nms_topk,
                              float nms_thresh, float min_size, float decode_clip, float* out_boxes,
                              float* out_scores, int32_t* out_idx, int32_t* out_count, void* stream);

/* roi_align (model.py:300-374): tf.image.crop_and_resize to (2*out_size)^2 on the remapped box, extrapolation 0,
 * fused with the 2x2 average pool.  rois: [n_img][rois_per_img][4] x1y1x2y2 in IMAGE coordinates, scaled by
 * spatial_scale (1/16); slots >= count[img] produce zeros.  out: NHWC [n_img*rois_per_img][out][out][c]. */
int premvos_roi_align_f32(const float* fmap, int32_t fmap_ps, int32_t n_img, int32_t h, int32_t w, int32_t c,
                          const float* rois, const int32_t* count, int32_t rois_per_img, float spatial_scale,
                          int32_t out_size, float* out, int32_t out_ps, void* stream);

/* GlobalAvgPooling (model.py:387,561): NHWC [n][hw][c] -> [n][c]. */
int premvos_global_avgpool_f32(const float* in, int32_t in_ps, int32_t n, int32_t hw, int32_t c, float* out,
                               int32_t out_ps, void* stream);

/* Inference tail (train.py:275-295, model.py:438-491) for NUM_CLASS=2: softmax, decode deltas / reg weights on
 * the proposals, clip, p > score_thresh, NMS(nms_thresh), <= max_out results by descending p.
 * head: [n_img][rois_per_img][head_ps] with class logits at [0,2) and box deltas at [2,6). */
int premvos_frcnn_tail_f32(const float* head, int32_t head_ps, const float* rois, const int32_t* count, int32_t n_img,
                           int32_t rois_per_img, float img_h, float img_w, float score_thresh, float nms_thresh,
                           int32_t max_out, float decode_clip, float rw_x, float rw_y, float rw_w, float rw_h,
                           float* out_boxes, float* out_probs, int32_t* out_idx, int32_t* out_count, void* stream);

/* ==========================================================================================
 * refinement_net (DeepLabv3+ / Xception-65 on 385x385 box crops); paths relative to code/refinement_net/.
 * Pointwise / dense convs go through premvos_conv2d_f32 (BatchNorm folded, ReLU and the module's
 * residual add fused in the epilogue).
 * ========================================================================================== */

/* Per-box network input, fusing datasets/util/BoundingBox.py:15-19 (guidance = 1 inside round(box)),
 * datasets/Resize.py:150-193 (crop = round(box) +- 50 px clipped; image: tf.image.resize_images bilinear,
 * guidance: resize_nearest_neighbor, both TF1-legacy coordinates), util/Normalization.py:9-21 and
 * DeepLabV3Plus.py:12-14 + core/feature_extractor.py:114-116.  frame: uint8 RGB [h][w][3];
 * boxes: [max_boxes][4] (y0,x0,y1,x1) floats; *count boxes are valid (rest -> zeros).
 * out: NHWC [max_boxes][size][size][4]; crop_boxes: [max_boxes][4] int32 (y0,x0,y1,x1). */
int premvos_refine_input_u8(const uint8_t* frame_rgb, int32_t h, int32_t w, const float* boxes_y0x0y1x1,
                            const int32_t* count, int32_t max_boxes, int32_t size, float* out, int32_t* crop_boxes,
                            void* stream);

/* Depthwise 3x3 (+stride, +atrous) with folded BatchNorm, optional ReLU on the INPUT (the xception module's
 * leading tf.nn.relu, core/xception.py:252-258) and on the output (exit flow / ASPP / decoder):
 * slim.separable_conv2d(num_outputs=None) of core/xception.py:154-167 and model.py:664-707.
 * wgt: [9][c_pad] (tap-major, BN scale folded), bias: [c_pad] or NULL. */
int premvos_dwconv3x3_f32(const float* in, int32_t in_ps, int32_t n, int32_t h, int32_t w, int32_t c,
                          const float* wgt, const float* bias, int32_t c_pad, float* out, int32_t out_ps, int32_t ho,
                          int32_t wo, int32_t stride, int32_t dilation, int32_t pt, int32_t pl, int32_t pre_relu,
                          int32_t act, void* stream);

/* Round 4: the bf16x3 mode on activations RESIDENT in the split layout "S8" -- per pixel, every group of 8 channels is the 32 bytes
 * {hi(8 x bf16), lo(8 x bf16)} (x = hi + lo; 4 bytes per element like fp32, pixel stride a multiple of 8 floats' worth of bytes,
 * channel windows 32-byte aligned).  Any conv of the three nets (1x1 / k x k, stride, dilation, asymmetric zero padding: the same
 * geometry fields of premvos_conv_desc as premvos_conv2d_f32, out_mode NHWC; replaces what the reference reaches through
 * tensorpack Conv2D basemodel.py:29-99, slim.conv2d / separable_conv2d's pointwise half xception.py:154-178, nn.Conv2d
 * PWCNet.py:24-34) as an implicit GEMM on v_mfma_f32_32x32x16_bf16: hi.hi + hi.lo + lo.hi, fp32 accumulate, operands staged by
 * LDS-DMA.  d->inp / d->wgt are IGNORED: in_s8 = the S8 input (d->in_ps its pixel stride, d->cin % 8 == 0), wgt_s8 = weights
 * packed as bf16 [cout_pad][kh*kw][ceil(cin/32)][4 groups][hi 8 | lo 8], zero padded (premvos_amd.ops.pack_conv_s8).
 * Outputs, each optional (at least one): d->out (fp32 NHWC, d->out_ps) and out_s8 (S8, pixel stride out_s8_ps floats' worth) =
 * act(sum + d->bias (+ d->res, fp32 -- or res_s8, the residual in S8 with pixel stride res_s8_ps: hi + lo, so that a bottleneck
 * chain needs no fp32 copy of its block outputs)); cout % 8 == 0.  tile: 0 = 256x256 / 8 waves / 2 buffers, 1 = 256x128 / 4 waves / 3 buffers,
 * 2 = 256x128 / 4 waves / 2 buffers, 3 = 256x128 / 8 waves / 3 buffers, 4 = 128x128 / 4 waves / 3 buffers, 5 = 128x128 / 2 buffers,
 * 6 / 7 = 256x128 and 8 = 128x256 on 16-channel stages (two workgroups per CU), 9 = 256x64, 10 = 256x256 ping-pong wave groups
 * (pointwise layers), 11 = 256x256 / 4 waves of 128x128.  Every tile adds an output's products in the same order: bit-identical
 * results -- the tile is a speed knob only (premvos_amd.ops.s8_tile_rule picks 0 or 5; the others are measured alternatives). */
int premvos_conv_bf16x3_s8_f32(const premvos_conv_desc* d, const void* in_s8, const void* wgt_s8, void* out_s8,
                               int32_t out_s8_ps, const void* res_s8, int32_t res_s8_ps, int32_t tile, void* stream);

/* fp32 NHWC [pixels][in_ps] -> S8 [pixels][out_ps floats' worth] (c channels; a partial last group is zero filled): the entry of an
 * S8 chain whose producer is an fp32 kernel. */
int premvos_split8_f32(const float* in, int32_t in_ps, void* out_s8, int32_t out_ps, int64_t pixels, int32_t c, void* stream);

/* tf.image.resize_bilinear, TF1: align_corners=1 (model.py:399-400,570-571) or 0 = legacy src = dst*in/out. */
int premvos_resize_bilinear_f32(const float* in, int32_t in_ps, int32_t n, int32_t h, int32_t w, int32_t c, float* out,
                                int32_t out_ps, int32_t ho, int32_t wo, int32_t align_corners, void* stream);

/* [n][1][1][c] -> every pixel of [n][h][w][c-slice]: the ASPP image-level feature (model.py:396-400). */
int premvos_broadcast_pixel_f32(const float* in, int32_t in_ps, int32_t n, int32_t c, float* out, int32_t out_ps,
                                int32_t h, int32_t w, void* stream);

/* SegmentationSoftmax eval branch (network/SegmentationOutputLayers.py:35-61,106-135) + the forwarder's
 * conf_score (forwarding/FewShotSegmentationForwarder.py:144-148): logits [max_boxes][lh][lw][ps>=2] ->
 * legacy-bilinear to size^2 -> softmax/argmax -> mask (nearest) and posterior (legacy bilinear) resized to the
 * crop box and zero padded to the frame.  mask: uint8 {0,1} [max_boxes][h][w]; posterior: float or NULL;
 * conf_score: [max_boxes] = mean over the frame of (2p-1 inside the mask, 1-2p outside), fixed-order reduction. */
int64_t premvos_refine_output_workspace_bytes(int32_t max_boxes, int32_t size, int32_t h, int32_t w);
int premvos_refine_output_f32(const float* logits, int32_t logits_ps, int32_t lh, int32_t lw, const int32_t* crop_boxes,
                              const int32_t* count, int32_t max_boxes, int32_t size, int32_t h, int32_t w,
                              uint8_t* mask, float* posterior, float* conf_score, void* workspace, void* stream);

/* Calibration kernel: `blocks` workgroups of 4 waves each issue iters*16 independent v_mfma_f32_32x32x2_f32 per wave
 * (FLOPs = blocks*4*iters*16*4096); timing it gives the fp32 MFMA rate the GPU sustains under its own power management
 * (bench.py: roofline.mfma_ceiling_measured). */
int premvos_mfma_f32_calibrate(int64_t iters, int32_t blocks, float* sink, void* stream);

/* The same loop on CHANGING pseudo-random operands (eight A and B registers per lane, another pair for every MFMA): the fp32 matrix
 * pipe's power follows the switching activity of its operands, and on random data the socket meets its cap and the shader clock gives
 * way (the constant operands above never do).  Timed over >= 1 s: the fp32 MFMA rate a box sustains on real data
 * (bench.py: roofline.mfma_ceiling_sustained_random_data). */
int premvos_mfma_f32_calibrate_random(int64_t iters, int32_t blocks, float* sink, void* stream);

/* Calibration kernel: copies n_float4 16-byte words src -> dst (one word per thread, flat grid);
 * 2 * 16 * n_float4 bytes / time = the streaming HBM rate the GPU sustains (bench.py: roofline.hbm_ceiling_measured).  Use
 * buffers well beyond the 256 MB Infinity Cache. */
int premvos_hbm_copy_calibrate(const void* src, void* dst, int64_t n_float4, void* stream);

/* Order-independent 64-bit digest of the 4-byte words of a pixel-major window [pixels][c] with pixel stride ps (words): the plan-time
 * tuner compares the outputs of configurations that must be bit-identical (same numerics key) before it lets a stopwatch choose
 * between them (premvos_amd/ops.py::_time_cands).  *out_u64 (device) receives the digest. */
int premvos_digest_u64(const void* buf, int64_t pixels, int32_t c, int32_t ps, void* out_u64, void* stream);

/* Host-side utility (no GPU work): CRC-32C of a host buffer -- the checksum of TensorFlow tensor-bundle checkpoints,
 * which premvos_amd/weights.py reads and writes without TensorFlow (proposal_net/train.py:655, core/Saver.py:33-48). */
uint32_t premvos_crc32c_host(const void* data, int64_t n);

/* ------------------------------------------------------------------------------------------
 * Optional GPU JPEG decode.  Replaces the image readers of the stages -- cv2.imread
 * (proposal_net/train.py:500, BGR), scipy.ndimage.imread / PIL (optical_flow_net-PWC-Net/script_pwc_multi.py:34,
 * ReID_net/prepare_input.py:38, RGB) -- i.e. libjpeg(-turbo) with its defaults: JDCT_ISLOW, fancy up-sampling, YCbCr -> RGB.
 * Baseline / extended-sequential Huffman files, 8 bit, grey or YCbCr 4:4:4 / 4:2:2 / 4:2:0, one interleaved scan, restart
 * intervals; anything else returns PREMVOS_EUNSUPPORTED and the caller keeps its default reader.  The output is the
 * library's, byte for byte.
 * ---------------------------------------------------------------------------------------- */
typedef struct premvos_jpeg_info {
  int32_t width, height, ncomp;  /* ncomp 1 (grey) or 3 (YCbCr) */
  int32_t hs, vs;                /* luma samples per chroma sample: 1x1, 2x1 or 2x2 (1x1 for grey) */
  int32_t mcux, mcuy;            /* MCU grid: ceil(width / (8 hs)) x ceil(height / (8 vs)) */
  int32_t blocks_w[3], blocks_h[3];
  int32_t reserved;
  int64_t coef_offset[3];        /* int16 elements: component c holds [blocks_h][blocks_w][64] quantised coefficients, row-major */
  int64_t coef_count;            /* total int16 elements */
  uint16_t quant[3][64];         /* quantisation table of each component, row-major (de-zig-zagged) */
} premvos_jpeg_info;

/* HOST function, no GPU work: marker parsing + Huffman decoding (ITU-T T.81 F.2).  coef == NULL: header only (fills *info).
 * Otherwise coef (host memory, any alignment; pinned for an asynchronous upload) receives info->coef_count values. */
int premvos_jpeg_entropy_decode_host(const uint8_t* data, int64_t n, premvos_jpeg_info* info, int16_t* coef,
                                     int64_t coef_capacity);
/* bytes of the device workspace (the component planes between the two kernels) */
int64_t premvos_jpeg_workspace_bytes(const premvos_jpeg_info* info);
/* coef: DEVICE copy of the coefficients (16-byte aligned); info: the host struct; out: uint8 [height][width][3],
 * RGB (bgr = 0) or BGR (bgr != 0, what cv2.imread returns).  De-quantisation + jidctint.c inverse DCT, then
 * jdsample.c fancy up-sampling fused with jdcolor.c's colour conversion. */
int premvos_jpeg_reconstruct_u8(const int16_t* coef, const premvos_jpeg_info* info, void* workspace, uint8_t* out,
                                int32_t bgr, void* stream);

/* ------------------------------------------------------------------------------------------
 * ReID embedding net (code/ReID_net).  Its convs / FC layers go through premvos_conv2d_f32, the pool
 * through premvos_maxpool_f32.
 * ---------------------------------------------------------------------------------------- */
/* datasets/Similarity/DAVIS_Forward_Feed.py:62-96 (zero_small = 1: boxes with min(w,h) <= 10 give a zero image) and
 * Similarity.py:288-297 (zero_small = 0): frame uint8 RGB [h][w][3] / 255, crop boxes_xywh[i] (int32 x, y, w, h; already
 * context-expanded, rounded and clipped by the host), tf.image.resize_images bilinear (TF1 legacy) to size x size,
 * (x - mean) / std -> NHWC [n][size][size][4] (4th channel 0). */
int premvos_reid_input_u8(const uint8_t* frame_rgb, int32_t h, int32_t w, const int32_t* boxes_xywh, int32_t n,
                          int32_t size, int32_t zero_small, float* out, void* stream);

/* Inference BatchNorm (+ ReLU when relu != 0) as a per-channel scale / shift over NHWC pixels, for pre-activation units
 * whose input is also consumed raw by the identity shortcut (network/NetworkLayers.py:171-173). */
int premvos_scale_shift_relu_f32(const float* in, int32_t in_ps, int64_t npix, int32_t c, const float* scale,
                                 const float* shift, float* out, int32_t out_ps, int32_t relu, void* stream);

/* ------------------------------------------------------------------------------------------
 * MergeTrack-side mask helpers (the consumer of the hot path).  Masks are uint8 [n][h][w] row-major,
 * nonzero = foreground, resident in HBM.
 * ---------------------------------------------------------------------------------------- */
/* MergeTrack/merge_functions.py:209-217 warp_flow: out = cv2.remap(mask, grid - flow, INTER_LINEAR) (uint8 fixed-point
 * path, zero border), then (== 1) when binarize != 0.  flow: float [h][w][2] = (u, v) as in the .flo payload. */
int premvos_mask_warp_u8(const uint8_t* masks, int32_t n, int32_t h, int32_t w, const float* flow, uint8_t* out,
                         int32_t binarize, void* stream);

/* merge_functions.py:38-45 (pycocotools iou on the proposals' RLEs): pixel counts of every pair --
 * inter[ib*na + ia] = |a_ia AND b_ib|, area_a[ia], area_b[ib]; the caller forms i/u in double (u = 1 when i == 0). */
int premvos_mask_overlap_u8(const uint8_t* a, int32_t na, const uint8_t* b, int32_t nb, int64_t hw, int64_t* inter,
                            int64_t* area_a, int64_t* area_b, void* stream);

/* merge_functions.py:224-226 (pycocotools encode of the Fortran-ordered mask): ascending column-major positions
 * q = x*h + y at which the value changes (value before q = 0 is background) -> positions[i*capacity ..], nruns[i] =
 * number of changes (may exceed capacity: then only the first `capacity` are stored).  COCO counts = successive
 * differences of [0, positions..., h*w]. */
int64_t premvos_rle_workspace_bytes(int32_t n, int32_t h, int32_t w);
int premvos_rle_boundaries_u8(const uint8_t* masks, int32_t n, int32_t h, int32_t w, int32_t* positions,
                              int32_t capacity, int32_t* nruns, void* workspace, void* stream);
/* The same boundaries for the n masks of a CHUNK of frames, POOLED: mask i's ascending positions go to
 * pool[offsets[i] .. offsets[i+1]) with offsets[0] = 0 -- one variable-length stream instead of n fixed-capacity rows, so that a
 * rank can hand the merge rank run lengths instead of masks (the RLE strings FewShotSegmentationForwarder.py:140-142 builds per
 * box are then pure host work on a few integers per run).  Mask i = the h x w top-left window of masks + i*mask_stride with
 * row_stride bytes between rows (masks inside a larger staging block).  offsets[n] = the total number of boundaries; it may exceed
 * pool_capacity: then only the entries below pool_capacity were stored and the caller falls back to the masks themselves.
 * workspace: premvos_rle_workspace_bytes(n, h, w). */
int premvos_rle_boundaries_pooled_u8(const uint8_t* masks, int32_t n, int32_t h, int32_t w, int64_t mask_stride,
                                     int32_t row_stride, int32_t* pool, int32_t pool_capacity, int32_t* offsets,
                                     void* workspace, void* stream);

/* Masks on the wire (ONE gather of fixed-size buffers per chunk to the merge rank, masks bit-packed): n mask
 * bytes (nonzero = foreground) -> ceil(n/8) bytes, bit k of byte i = masks[8i + k] != 0; and back to {0,1} bytes.  Replaces the
 * reference's hand-over through per-frame JSON files (refinement_net/forwarding/FewShotSegmentationForwarder.py:151-155 read
 * back by MergeTrack/merge_functions.py:38-76). */
int premvos_mask_pack_bits_u8(const uint8_t* masks, int64_t n, uint8_t* bits, void* stream);
int premvos_mask_unpack_bits_u8(const uint8_t* bits, int64_t n, uint8_t* masks, void* stream);

/* Host-side utility (no GPU work): COCO rleToString of `n` run lengths into `out` (capacity `cap` bytes); returns the
 * length or -1 -- the "counts" string of every mask the refinement / merge stages write
 * (forwarding/FewShotSegmentationForwarder.py:141-142). */
int64_t premvos_rle_counts_to_string_host(const int64_t* counts, int64_t n, char* out, int64_t cap);
/* Host-side utility (no GPU work): the "counts" strings of n masks from their pooled boundaries (host copies of `pool` /
 * `offsets` above; hw = h*w), written back to back into `out`; string i = out[str_offsets[i] .. str_offsets[i+1]).  Returns the
 * total length or -1 when `cap` is too small (13 bytes per run always suffice). */
int64_t premvos_rle_strings_host(const int32_t* pool, const int32_t* offsets, int32_t n, int64_t hw, char* out, int64_t cap,
                                 int64_t* str_offsets);

/* ---- host-side file writer (no GPU work; premvos_amd/csrc/host_files.hip) -------------------------------------------------
 * The files of ONE frame from the arrays its results consist of, without the Python interpreter (ctypes releases the interpreter
 * lock for the call, so N writer threads run at once): what the merge rank of a gathered multi-GPU job does ~430 times per second.
 * Same bytes as the Python writers of the stage drivers, i.e. as the reference's:
 *   flo_path      script_pwc_multi.py:16-31 writeFlowFile           (NULL: none -- the last frame of a video has no pair)
 *   json_path[0]  general proposals   proposal_net/eval.py:93-94 (boxes / scale, clip) + train.py:388-428 (xywh, round, json.dump)
 *   json_path[1]  specific proposals  (same)
 *   json_path[2]  combined            combine_general_and_specific.py:33 (general + specific)
 *   json_path[3]  refined             FewShotSegmentationForwarder.py:137-155: combined + "segmentation" {"size", "counts"} +
 *                                     "conf_score" (str of the float32); slot i of the frame = rle_offsets[i] .. [i + 1] in rle_pool
 *                                     (premvos_rle_boundaries_pooled_u8), conf[i]
 * Any json_path may be NULL (that file is not written).  boxes: [count][4] x0 y0 x1 y1 in RESIZED-image coordinates, probs: [count].
 * Returns 0, 1 when a directory of some path does not exist (the caller creates it and calls again; files already written are
 * simply written again) or a negative error. */
typedef struct premvos_frame_files {
  const char* flo_path;
  const float* flow;              /* [h][w][2], flow_row_stride floats between rows (>= 2 w) */
  int64_t flow_row_stride;
  int32_t h, w;
  const float* boxes[2];          /* general, specific */
  const float* probs[2];
  int32_t count[2];
  float scale;                    /* float32 of (newh / h + neww / w) / 2 (eval.py:78) */
  const char* json_path[4];
  const float* conf;              /* [count[0] + count[1]] */
  const int32_t* rle_pool;
  const int32_t* rle_offsets;     /* [count[0] + count[1] + 1] (a window of the chunk's offsets) */
} premvos_frame_files;
int premvos_write_frame_files_host(const premvos_frame_files* f);
/* Test hook of the number formatting above: n doubles -> one line each, as Python's repr(float) (json.dump) or, as_float32_str != 0,
 * as numpy's str(float32(value)).  Returns the length written or a negative error. */
int premvos_format_floats_host(const double* values, int64_t n, int32_t as_float32_str, char* out, int64_t cap);

#ifdef __cplusplus
}
#endif
#endif /* PREMVOS_HIP_H */
