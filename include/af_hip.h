/*
 * af_hip.h — C ABI of libafhip.so: the MI355X (gfx950) kernels behind the AltFreezing
 * `i3d_ori` clip classifier forward.
 *
 * The reference has no FFI: its hot path is stock torch.nn modules called from Python
 * (SURVEY.md section 8b).  Each entry point below therefore names the reference *operator
 * sequence* it replaces (file:line under /root/reference/altfreezing unless noted); the
 * ctypes binding a maintainer would add is shown in INTEGRATION.md.
 *
 * Conventions
 *   - plain pointers and ints only; every buffer is caller-owned DEVICE memory (16-byte
 *     aligned) and must stay alive until the stream has drained;
 *   - activations are NDHWC (channels innermost), element type `dtype` (AF_F32/AF_BF16/AF_F16);
 *     accumulation is always fp32; BatchNorm is applied as a per-channel fp32 scale/shift
 *     in the producing kernel's epilogue;
 *   - kernels are enqueued asynchronously on `stream` (a hipStream_t passed as void*,
 *     NULL = the null stream); nothing here synchronises, allocates or frees device memory (scratch a launch
 *     may need is sized by af_conv_workspace_bytes() and passed in by the caller), except the *_timed entry
 *     points which record and wait for their own events; there is no mutable global state beyond per-device
 *     "attribute already set" flags, so launches on different streams / devices of one process do not interact;
 *   - return value 0 = success, negative = error; af_last_error() (thread-local text).
 */
#ifndef AF_HIP_H
#define AF_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AF_ABI_VERSION 4

enum af_dtype { AF_F32 = 0, AF_BF16 = 1, AF_F16 = 2 };

enum af_status {
    AF_OK = 0,
    AF_ERR_ARG = -1,        /* bad pointer / dimension / unsupported configuration */
    AF_ERR_LAUNCH = -2,     /* hipLaunch / runtime error */
    AF_ERR_NO_DEVICE = -3
};

/* stem input geometry: [N][T+4][H+6][W+8][4] with zero halos (2 | 3 | 3-left,5-right) and a
 * zero 4th channel; written by af_pack_input_* and read by af_stem_conv_bn_relu. */
#define AF_STEM_PAD_T 2
#define AF_STEM_PAD_H 3
#define AF_STEM_PAD_W_LEFT 3
#define AF_STEM_PAD_W_TOTAL 8
#define AF_STEM_CPAD 4

typedef struct af_conv_desc {
    int32_t n, t, h, w, cin;        /* input  [n][t][h][w][cin]              */
    int32_t cout;
    int32_t kt, kh, kw;
    int32_t st, sh, sw;
    int32_t pt, ph, pw;
    int32_t to, ho, wo;             /* output [n][to][ho][wo][cout]; checked */
    int32_t relu;                   /* ReLU after scale/shift(+residual)     */
    int32_t dtype;                  /* af_dtype of in / weights / residual / out */
    int32_t tpool;                  /* 1: also apply MaxPool3d([2,1,1],stride [2,1,1]) (pathway0_pool,
                                       video_model_builder.py:474-480) to the result: out is then
                                       [n][to/2][ho][wo][cout]; to must be even (af_conv3d_bn_act only)
                                       2: also apply MaxPool3d((1,2,2)) (what FTCN-TT's temporal_only_conv puts
                                       after the BN of a conv that lost its stride 2): out is [n][to][ho/2][wo/2][cout];
                                       ho, wo even, no residual (the ReLU, if any, commutes with the max) */
} af_conv_desc;

typedef struct af_pool_desc {
    int32_t n, t, h, w, c;
    int32_t kt, kh, kw, st, sh, sw, pt, ph, pw;
    int32_t to, ho, wo;
    int32_t dtype;
    int32_t out_ld;                 /* af_maxpool3d: channel stride of the output rows (0 = c); lets the Slow
                                       pathway's pooled stem leave room for the lateral's channels */
} af_pool_desc;

int af_version(void);
const char* af_last_error(void);
/* number of visible HIP devices, or AF_ERR_NO_DEVICE */
int af_device_count(void);

/* ---- one-time weight transforms (model load) ------------------------------------------- */

/* nn.BatchNorm3d(eval) -> scale = gamma / sqrt(var + eps), shift = beta - mean * scale
 * (slowfast/models/batchnorm_helper.py:23-24; applied at stem_helper.py:175, resnet_helper.py:314-325,441) */
int af_fold_bn(const float* gamma, const float* beta, const float* mean, const float* var,
               float eps, int channels, float* scale, float* shift, void* stream);

/* nn.Conv3d weight OIDHW fp32 -> [cout_pad][kt*kh*kw][cin_pad] in `dtype` (K-major rows for the implicit GEMM);
 * cout is padded to a multiple of 64 rows and cin to the K-step (64 elements, fp32: 32) with zeros, so any
 * channel count that is a multiple of 8 (fp32: 4) runs (SlowFast's 8/16/32/80/320-channel layers).
 * Output bytes: af_packed_conv_weight_bytes().  The BN scale/shift arrays handed to the conv entry points must
 * have af_padded_channels(cout) entries (padding = anything finite). */
int64_t af_packed_conv_weight_bytes(int cout, int cin, int kt, int kh, int kw, int dtype);
int af_padded_channels(int cout);
int af_pack_conv_weight(const float* w_oidhw, int cout, int cin, int kt, int kh, int kw,
                        int dtype, void* packed, void* stream);
/* same, with each output-channel row multiplied by row_scale[cout] in fp32 before the rounding to `dtype`
 * (BatchNorm scale folded into the weights: needed when two convolutions share one accumulator, below) */
int af_pack_conv_weight_scaled(const float* w_oidhw, const float* row_scale, int cout, int cin, int kt, int kh, int kw,
                               int dtype, void* packed, void* stream);

/* stem weight (64,3,kt,7,7) fp32 -> [kt][7][chunk][64][16 B]: kw padded 7->8, cin 3->4 (zeros) */
int64_t af_packed_stem_weight_bytes(int cout, int kt, int kh, int dtype);
int af_pack_stem_weight(const float* w_oidhw, int cout, int kt, int kh, int kw, int dtype,
                        void* packed, void* stream);

/* ---- input prologue (replaces the callers' as_tensor/permute/sub/div, test/af_realtime.py:77-83) */

int64_t af_stem_input_bytes(int n, int t, int h, int w, int dtype);
/* (n,3,t,h,w)-logical fp32 tensor with arbitrary element strides -> padded stem input */
int af_pack_input_f32(const float* x, int n, int t, int h, int w,
                      int64_t stride_n, int64_t stride_c, int64_t stride_t, int64_t stride_h, int64_t stride_w,
                      int dtype, void* stem_in, void* stream);
/* caller-layout uint8 clips (n,t,h,w,3), 0..255 RGB -> (x - mean[c]) / std[c] -> padded stem input */
int af_pack_input_u8(const uint8_t* clips, int n, int t, int h, int w,
                     const float mean[3], const float std_[3], int dtype, void* stem_in, void* stream);

/* ---- layers ------------------------------------------------------------------------- */

/* Conv3d(3->64,[kt,7,7],s[1,2,2],p[kt/2,3,3],bias=False)+BN+ReLU  (stem_helper.py:156-177).
 * d->cin must be 3, in = padded stem input, out = [n][to][ho][wo][cout]. */
int af_stem_conv_bn_relu(const af_conv_desc* d, const void* stem_in, const void* w_packed,
                         const float* scale, const float* shift, void* out, void* stream);

/* The whole ResNetBasicStem (stem_helper.py:173-178) as one launch: conv + BN + ReLU +
 * MaxPool3d([1,3,3],stride [1,2,2],pad [0,1,1]); out = [n][to][(ho-1)/2+1][(wo-1)/2+1][64] where (to,ho,wo) are
 * the conv output dims in `d`.  16-bit dtypes only (all weight slices stay resident in LDS); the conv output
 * tensor is never materialised. */
int af_stem_conv_bn_relu_maxpool(const af_conv_desc* d, const void* stem_in, const void* w_packed,
                                 const float* scale, const float* shift, void* out, void* stream);

/* The same fused stem on the K-PACKED input ("rgb3": 3 real channels per pixel, 6 bytes; rows of
 * ((w + 8) * 6 rounded up to 16) bytes, frames / rows / left halo padded like the 4-channel layout): the 7 x 7 x 3 taps of a
 * (dt) plane fill 5.25 MFMA K-blocks instead of 7 - the stem is MFMA-bound on exactly that padding (kt = 5: 27 blocks instead
 * of 35).  af_stem_input_bytes_rgb3 includes the few rows of slack the last frame needs; the buffer is zero-filled once by
 * the caller, af_pack_input_*_rgb3 write the interior (same arguments and arithmetic as af_pack_input_*); weights:
 * af_pack_stem_weight_rgb3 -> af_packed_stem_weight_bytes_rgb3(kt, dtype) bytes.  16-bit dtypes, 64 output channels. */
int64_t af_stem_input_bytes_rgb3(int n, int t, int h, int w, int dtype);
int af_pack_input_f32_rgb3(const float* x, int n, int t, int h, int w,
                           int64_t stride_n, int64_t stride_c, int64_t stride_t, int64_t stride_h, int64_t stride_w,
                           int dtype, void* stem_in, void* stream);
int af_pack_input_u8_rgb3(const uint8_t* clips, int n, int t, int h, int w,
                          const float mean[3], const float std_[3], int dtype, void* stem_in, void* stream);
int64_t af_packed_stem_weight_bytes_rgb3(int kt, int dtype);
int af_pack_stem_weight_rgb3(const float* w_oidhw, int cout, int kt, int dtype, void* packed, void* stream);
int af_stem_conv_bn_relu_maxpool_rgb3(const af_conv_desc* d, const void* stem_in, const void* w_packed,
                                      const float* scale, const float* shift, void* out, void* stream);
/* the same with a pooled row stride of out_ld channels (>= 64): SlowFast's Slow stem writes its 64 channels in front of the
 * lateral's (FuseFastToSlow concatenates by channel, video_model_builder.py:136-143) (ABI 3) */
int af_stem_conv_bn_relu_maxpool_rgb3_ld(const af_conv_desc* d, const void* stem_in, const void* w_packed, const float* scale,
                                         const float* shift, void* out, int out_ld, void* stream);

/* Conv3d(bias=False)+BN[+residual add][+ReLU] as one implicit-GEMM launch: the a/b/c convs of
 * BottleneckTransform (resnet_helper.py:267-325), the projection shortcut and the add+ReLU of
 * ResBlock (resnet_helper.py:411-444), FuseFastToSlow's conv_f2s+bn+relu (video_model_builder.py:121-143).
 * Requires cin % 8 == 0 and cout % 8 == 0 (fp32: % 4); scale/shift have af_padded_channels(cout) entries.
 * `residual` (NULL or [n][to][ho][wo][cout]) is added before the ReLU.
 * `out_ld` = channel stride of `out` rows in elements (>= cout; lets FuseFastToSlow write into
 * the concatenated slow tensor), 0 means cout.
 * `workspace` / `workspace_bytes`: caller-owned device scratch (16-byte aligned) for the split-K path that small
 * batches take in long-K layers (a live call's 1-3 clips: a layer with a handful of tiles splits its K range over up
 * to 8 workgroups per tile, fp32 partial sums meet in the workspace).  af_conv_workspace_bytes(d) is the size this
 * layer can use (0: never splits); with NULL or fewer bytes the layer runs unsplit (same result to fp32 rounding).  The
 * workspace is free for reuse as soon as the NEXT launch on the same stream has been enqueued; launches that may run
 * concurrently (different streams) need one each. */
int64_t af_conv_workspace_bytes(const af_conv_desc* d);
int af_conv3d_bn_act(const af_conv_desc* d, const void* in, const void* w_packed,
                     const float* scale, const float* shift, const void* residual,
                     void* out, int out_ld, void* workspace, int64_t workspace_bytes, void* stream);

/* Block 0 of a stage as ONE launch: relu( bn_c(conv_c(b)) + bn_1(conv_1(x)) ), i.e. the last 1x1x1 of the
 * bottleneck plus the projection shortcut `branch1` (1x1x1, stride [1,s,s]) of ResBlock
 * (resnet_helper.py:411-444) accumulated into the same output tile: the shortcut tensor is never written
 * to or read back from HBM.  Both weights must be packed with their BN scale folded in
 * (af_pack_conv_weight_scaled); `scale` is then all ones and `shift` = shift_c + shift_1.
 * d2 describes the shortcut conv over in2 and must land on the same output positions as d. */
int af_conv3d_dual_bn_act(const af_conv_desc* d, const void* in, const void* w_packed,
                          const af_conv_desc* d2, const void* in2, const void* w2_packed,
                          const float* scale, const float* shift, void* out, int out_ld, void* stream);

/* A whole bottleneck block of a NARROW pathway as ONE launch (ABI 3):
 *     out = relu( shortcut(x) + bn_c(conv_c( relu(bn_b(conv_b( relu(bn_a(conv_a(x))) ))) )) )
 * = ResBlock.forward over BottleneckTransform (resnet_helper.py:255-326, 411-444) as SlowFast's Fast pathway instantiates it
 * (video_model_builder.py:146-387: dim_inner 8 / 16, trunk 32 / 64 channels).  da: kT x 1 x 1 (kT = 1 or 3) -> C/4, db: 1x3x3
 * C/4 -> C/4, dc: 1x1x1 C/4 -> C, all stride 1, 16-bit, C/4 = 8 or 16; the a and b tensors never leave the CU.
 * d1 == NULL: identity shortcut (x has C channels and is also the residual).  d1 != NULL (block 0 of the Fast pathway's s2:
 * 8 -> 32 channels): shortcut = bn_1(conv_1x1x1(x)), computed like af_conv3d_dual_bn_act does - wc_packed and w1_packed carry
 * their BN scale (af_pack_conv_weight_scaled), scale_c is all ones and shift_c = shift_c + shift_1.  `x` must not alias `out`.
 * af_block_abc_fusable() says whether a block takes this path. */
int af_block_abc_fusable(const af_conv_desc* da, const af_conv_desc* db, const af_conv_desc* dc, const af_conv_desc* d1);
int af_block_abc_bn_act(const af_conv_desc* da, const void* x, const void* wa_packed, const float* scale_a, const float* shift_a,
                        const af_conv_desc* db, const void* wb_packed, const float* scale_b, const float* shift_b,
                        const af_conv_desc* dc, const void* wc_packed, const float* scale_c, const float* shift_c,
                        const af_conv_desc* d1, const void* w1_packed, void* out, int out_ld, void* stream);

/* The 1x3x3 `b` conv (+BN+ReLU) and the 1x1x1 `c` conv (+BN) of a bottleneck + the ResBlock's residual add + ReLU
 * (resnet_helper.py:283-325, 438-444) as ONE launch: relu( bn_c(conv_c( relu(bn_b(conv_b(in))) )) + residual ).  The b
 * output of a frame (or band of rows) stays in LDS as the c conv's operand and never reaches HBM.  Available for the shapes
 * af_conv_bc_fusable() accepts (16-bit, stride 1, 128 or 256 mid channels, enough frames to fill the chip: the s3 / s4
 * bottlenecks at batch >= 12); `dc` describes the c conv over b's output ([n][t][h][w][db->cout]); `residual` may be NULL. */
int af_conv_bc_fusable(const af_conv_desc* db, const af_conv_desc* dc);
int af_conv3d_bc_bn_act(const af_conv_desc* db, const void* in, const void* wb_packed, const float* scale_b, const float* shift_b,
                        const af_conv_desc* dc, const void* wc_packed, const float* scale_c, const float* shift_c,
                        const void* residual, void* out, int out_ld, void* stream);

/* The END of one res block and the START of the next, across the block boundary, as ONE launch (s2, 16-bit):
 *     x     = relu( bn_c(conv1x1x1_c(b)) + residual )      resnet_helper.py:304-325, 438-444   -> out_x  [n][t][h][w][C]
 *     a_out = relu( bn_a(conv3x1x1_a(x)) )                  resnet_helper.py:267-281 (next block) -> out_a [n][t][h][w][64]
 * in the time-tiled layout of the 3x1x1 kernel (tile = all T frames of a few pixels: the temporal halo is inside the tile):
 * per 64-channel slab of the trunk the residual is fetched, the c conv's slab is computed and added in LDS, stored once and
 * multiplied by the three temporal taps there - the `a` conv no longer re-reads the trunk from HBM.  dc: 64 -> C channels
 * (C % 64 == 0), ReLU; da: C -> 64, kernel [3,1,1], pad [1,0,0]; T = 16 or 32; batches with >= 4 tiles per CU
 * (af_conv_ca_fusable says whether a pair qualifies).  Block 0 of a stage: instead of `residual`, the projection shortcut
 * (d1: 1x1x1, stride 1, 64 -> C over in1; both weight sets packed with their BN scale, scale_c = ones, shift_c = the summed
 * shifts - as for af_conv3d_dual_bn_act) is a second K segment of the c conv's accumulator; d1 / in1 / w1_packed are NULL otherwise. */
int af_conv_ca_fusable(const af_conv_desc* dc, const af_conv_desc* d1, const af_conv_desc* da);
int af_conv3d_ca_bn_act(const af_conv_desc* dc, const void* in_b, const void* wc_packed, const af_conv_desc* d1, const void* in1,
                        const void* w1_packed, const float* scale_c, const float* shift_c, const void* residual, void* out_x,
                        const af_conv_desc* da, const void* wa_packed, const float* scale_a, const float* shift_a, void* out_a,
                        void* stream);

/* The s2 -> s3 boundary as ONE launch (16-bit): the c conv of s2's last block with its residual and ReLU, pathway0_pool
 * (MaxPool3d [2,1,1], video_model_builder.py:566-569) and the 3x1x1 a conv of s3's block 0 (resnet_helper.py:267-281):
 *     x     = relu( bn_c(conv1x1x1_c(b)) + residual )                   [n][32][h][w][C]  (never stored)
 *     xp    = max(x[2t'], x[2t'+1])                                     -> out_x
 *     a_out = relu( bn_a(conv3x1x1_a(xp)) )                             -> out_a [n][16][h][w][128]
 * dc: 64 -> C (C % 64 == 0, C <= 256), tpool = 1, t = 32, h even, w % 4 == 0; da: C -> 128, kernel [3,1,1], pad [1,0,0], t = 16.
 * x_sub = 1: out_x is the whole pooled trunk [n][16][h][w][C]; x_sub = 2: only its even (h, w) positions, packed as
 * [n][16][h/2][w/2][C] - all that a 1x1x1 convolution of stride (1,2,2) (the stage's projection shortcut, resnet_helper.py:
 * 411-431) reads; its caller then runs that convolution with stride 1 over the packed tensor.  (ABI 3) */
int af_conv_cpa_fusable(const af_conv_desc* dc, const af_conv_desc* da, int x_sub);
int af_conv3d_cpa_bn_act(const af_conv_desc* dc, const void* in_b, const void* wc_packed, const float* scale_c, const float* shift_c,
                         const void* residual, void* out_x, int x_sub, const af_conv_desc* da, const void* wa_packed,
                         const float* scale_a, const float* shift_a, void* out_a, void* stream);

/* which tile variant af_conv3d_[dual_]bn_act launches for `d` (+ optional `d2`) (>= 0) and its kernel name:
 * lets a profiler attribute per-layer device time and FLOPs to a kernel instantiation (bench.py roofline). */
int af_conv_variant(const af_conv_desc* d, const af_conv_desc* d2);
const char* af_conv_variant_name(int variant);

/* nn.MaxPool3d on NDHWC (stem_helper.py:168-170 [1,3,3]/[1,2,2]/[0,1,1];
 * video_model_builder.py:474-480 [2,1,1]/[2,1,1]); padding behaves as -inf. */
int af_maxpool3d(const af_pool_desc* d, const void* in, void* out, void* stream);

/* ResNetBasicHead (head_helper.py:74-95): AvgPool3d(pool,stride 1) -> Linear(c -> num_classes).
 * pooled (optional, may be NULL): fp32 [n][to*ho*wo][c] pooled features (the input of the
 * nn.Linear that feature.py:105-114 hooks); logits: fp32 [n][to*ho*wo*num_classes]. */
int af_avgpool_fc(const af_pool_desc* d, const void* in, const float* fc_w, const float* fc_b,
                  int num_classes, float* pooled, float* logits, void* stream);
/* the same with the callers' score epilogue fused behind the Linear (ClassifierSvc.infer_scores, test/af_realtime.py:88-95;
 * = TEST2.py:186-204, demo.py:328-331): scores[row] = sigmoid(logit) for num_classes == 1, softmax(logits)[1] for
 * num_classes == 2 (fp32 [n * to*ho*wo]); scores may be NULL. */
int af_avgpool_fc_scores(const af_pool_desc* d, const void* in, const float* fc_w, const float* fc_b,
                         int num_classes, float* pooled, float* logits, float* scores, void* stream);

/* The two halves of the head separately, for multi-pathway heads (SlowFast: one AvgPool3d per pathway, pooled
 * vectors concatenated by channel - head_helper.py:79-85 - then one Linear): af_avgpool writes row r of the pooled
 * result at pooled[r * pooled_ld + 0 .. c) (pass `pooled + channel_offset` to concatenate);
 * af_linear: y[r][k] = dot(x[r][0..in_features), w[k]) + b[k] on fp32. */
int af_avgpool(const af_pool_desc* d, const void* in, float* pooled, int pooled_ld, void* stream);
int af_linear(const float* x, const float* w, const float* b, int rows, int in_features, int out_features,
              float* y, void* stream);
int af_linear_scores(const float* x, const float* w, const float* b, int rows, int in_features, int out_features,
                     float* y, float* scores, void* stream);

/* ---- FTCN-TT plugin (reference model/classifier/i3d_temporal_var_fix_dropout_tt_cfg.py, time_transformer.py) ---- */

/* Temporal stem: Conv3d(3->64,[kt,1,1],stride 1,pad [kt/2,0,0]) + BN + MaxPool3d((1,2,2)) + ReLU - what
 * `temporal_only_conv` (:207-288) makes of ResNetBasicStem's conv / bn (stem_helper.py:156-178); d describes the conv
 * with to/ho/wo = t, h/2, w/2 (the pool is fused).  stem_in: af_pack_input_* buffer; out: NDHWC [n][t][h/2][w/2][64].
 * The stem's own MaxPool3d([1,3,3],[1,2,2],[0,1,1]) follows as af_maxpool3d - or rides along:
 * af_tstem_conv_bn_pool_relu_maxpool (ABI 3, 16-bit) takes the same descriptor and writes the stem pool's output
 * [n][t][(h/2-1)/2+1][(w/2-1)/2+1][64] directly (the half-resolution tensor never exists; i3d_temporal_var_fix_dropout_tt_cfg.py
 * keeps ResNetBasicStem's pool_layer behind the converted conv / bn). */
int af_pack_tstem_weight(const float* w_oidhw, int cout, int kt, int dtype, void* out, void* stream);
long long af_packed_tstem_weight_bytes(int dtype);
int af_tstem_conv_bn_pool_relu(const af_conv_desc* d, const void* stem_in, const void* w_packed, const float* scale,
                               const float* shift, void* out, void* stream);
int af_tstem_conv_bn_pool_relu_maxpool(const af_conv_desc* d, const void* stem_in, const void* w_packed, const float* scale,
                                       const float* shift, void* out, void* stream);

/* TimeTransformer head (time_transformer.py:219-281), fp32.  The Linear layers are af_conv3d_bn_act over the token
 * rows (1x1x1, AF_F32, scale = ones, shift = bias, residual = the skip connection).
 * af_tokens_assemble: out[b][0] = cls + pos[0], out[b][1+t] = pooled[b][t] + pos[1+t]   (:270-273)
 * af_layernorm: nn.LayerNorm(dim) over `rows` rows (PreNorm :15-21; mlp_head :259), row strides in elements
 * af_attention: softmax(q k^T / sqrt(dim_head)) v per (clip, head), qkv rows = [q | k | v] (Attention.forward :52-71)
 * af_gelu: nn.GELU() (erf form) in place (FeedForward :27) */
int af_tokens_assemble(const float* pooled, const float* cls_token, const float* pos_embedding, int clips, int n_tok,
                       int dim, float* out, void* stream);
int af_layernorm(const float* x, long long x_row_stride, const float* gamma, const float* beta, int rows, int dim,
                 float eps, float* y, long long y_row_stride, void* stream);
int af_attention(const float* qkv, int clips, int n_tok, int heads, int dim_head, float* out, void* stream);
int af_gelu(float* x, long long n, void* stream);

/* ---- dualrun AU / landmark dual encoder (reference dualrun/model/dual_encoder.py) --------------------- */

/* BranchEncoder.forward (:73-107) of `branches` modalities (AU, landmarks, ...) for `clips` clips of `frames` <= 16
 * frames, fp32, ONE launch (grid = clips x branches): x[b] [clips][frames][din[b]]; lengths [clips] valid frames (NULL:
 * all; 0 counts as 1 like :162-166); pe [frames][d_model] sinusoidal table (:16-23);
 * z[clip * z_ld + b * d_model + 0..d_model) = the attention-pooled clip vector of branch b (already concatenated).
 * weights[b] is the flat fp32 image af_dual_branch_weight_floats() sizes: proj W^T, b | ln_in g, b | 3 x (depthwise
 * w [c][3], b) | pointwise W^T, b | per layer: norm1 g, b | in_proj W^T, b | out_proj W^T, b | norm2 g, b | linear1
 * W^T, b | linear2 W^T, b | pool v   (W^T = [in][out], af_transpose_f32 of the checkpoint's [out][in]). */
long long af_dual_branch_weight_floats(int din, int d_model, int depth, int ff);
int af_transpose_f32(const float* src, int rows, int cols, float* dst, void* stream);
int af_dual_branch_encoders(int branches, const float* const* x, const float* const* weights, const int* din,
                            const int* lengths, const float* pe, int clips, int frames, int d_model, int depth, int heads,
                            int ff, float pool_tau, float* z, int z_ld, void* stream);
/* DualEncoderAU_LMK.head (:115-121): LayerNorm(n) -> Linear(n,n) -> GELU -> Linear(n,1) on z [clips][n];
 * weights: gamma, beta, W1^T [n][n], b1, w2 [n], b2. */
int af_dual_head(const float* z, const float* weights, int clips, int n, float* logits, void* stream);

/* dualrun tri-modal model (reference dualrun/model/dual_rgb.py).
 * af_masked_mean_proj: AltFreezingRGBEncoder.forward with from_features (:27-44) + rgb_proj (:70, Linear without bias):
 *   v [clips][tv][vis] per-frame / per-window RGB features (the AltFreezing backbone's pooled 2048-vector, feature.py:105-114);
 *   lengths [clips] = valid frames of a key-padding mask over `tmask` frames (NULL: no mask = plain mean); tv == tmask, or
 *   tv == 1 (broadcast against the mask like torch does); wt = rgb_proj.weight^T [vis][d];
 *   z[clip * z_ld + 0..d) = (sum_t v[t] * valid[t] / max(sum valid, 1e-6)) @ wt.
 * af_mlp_head: DualEncoderRGB.head (:75-79) = LayerNorm(n) -> Linear(n, hidden) -> GELU -> Linear(hidden, 1) on z [clips][n];
 *   weights: gamma [n], beta [n], W1^T [n][hidden], b1 [hidden], w2 [hidden], b2; scores (optional) = sigmoid(logit).
 *   (af_dual_head is the hidden == n case without scores.) */
int af_masked_mean_proj(const float* v, int clips, int tv, int vis, const int* lengths, int tmask, const float* wt, int d,
                        float* z, int z_ld, void* stream);
int af_mlp_head(const float* z, const float* weights, int clips, int n, int hidden, float* logits, float* scores, void* stream);

/* GatedMoE.forward (dualrun/rgb/engine_rgb.py:369-384): fuses the RGB (AltFreezing) logit and the dual-encoder logit of n
 * clips: gate = sigmoid(W2 relu(W1 [z_rgb, z_dual, |z_rgb - z_dual|] + b1) + b2), p = gate * sigmoid(z_rgb / max(t_rgb, 1))
 * + (1 - gate) * sigmoid(z_dual / max(t_dual, 0.1)), z = logit(p) with eps 1e-6.  weights: t_rgb, t_dual, W1 [hidden][3],
 * b1, w2 [hidden], b2. */
int af_gated_moe(const float* z_rgb, const float* z_dual, const float* weights, int hidden, int n, float* z, float* gate,
                 void* stream);

/* ---- clip aligner --------------------------------------------------- */

/* Replaces the per-frame loop of FasterCropAlignXRay.__call__ / process_single
 * (altfreezing/test_tools/faster_crop_align_xray.py:62-66, 75-88):
 *     new_image = zeros((h, w, 3), uint8); new_image[y:y+ih, x:x+iw] = image; cv2.warpAffine(new_image, tfm, (size, size))
 * for the n_frames frames of one clip in one launch.  `crops`: device buffer holding the frames' HxWx3 uint8 crops
 * (tightly packed rows), frame i at byte `frames[i].offset`; `frames` (HOST memory, copied into the launch): crop size
 * (ih, iw) and paste position (x, y) on the h x w canvas - a crop that does not fit the canvas is an error, as numpy's
 * slice assignment is in the reference; `tfm`: the forward 2x3 matrix (host, row-major doubles) as passed to
 * cv2.warpAffine; `out`: device (n_frames, size, size, 3) uint8 - the layout af_pack_input_u8 takes.
 * Arithmetic: OpenCV's fixed-point INTER_LINEAR / BORDER_CONSTANT(0) warp (see csrc/af_align.hip). */
#define AF_ALIGN_MAX_FRAMES 64
typedef struct af_align_frame {
    int64_t offset;
    int32_t ih, iw;
    int32_t x, y;
} af_align_frame;
int af_warp_affine_clip_u8(const void* crops, const af_align_frame* frames, int n_frames, int canvas_h, int canvas_w,
                           const double* tfm, int size, void* out, void* stream);

/* Host-side staging for the aligner (no device work): copies n rectangles of uint8 rows (the part of each crop the warp can
 * sample) into one staging buffer, one memcpy per row.  dst_offset / rows / row_bytes describe the packed destination,
 * src_pitch the source's row pitch in bytes.  (ABI 3) */
typedef struct af_stage_rect {
    const void* src;
    int64_t dst_offset;
    int64_t src_pitch;
    int32_t rows, row_bytes;
} af_stage_rect;
int af_stage_rows_u8(void* dst, const af_stage_rect* rects, int n);

/* Host-side planning of ONE clip for the aligner (no device work; ABI 4): what FasterCropAlignXRay.__call__ does per frame between the
 * similarity fit and the warp (reference altfreezing/test_tools/faster_crop_align_xray.py:75-88: paste every crop on the common
 * canvas, warp), as one call instead of a Python loop per frame.  crops[i]: the HxWx3 uint8 crop (first byte, row pitch in bytes -
 * pixels of a row contiguous - height, width) and its paste offset (x, y) on the canvas_w x canvas_h canvas; tfm: the forward 2x3
 * matrix; size: the output edge.  Checks that every crop fits the canvas (AF_ERR_ARG with the frame in *bad_frame: numpy's slice
 * assignment raises there in the reference), cuts each crop to the rows the size x size destination can sample (bilinear taps +
 * fixed-point rounding: 3 pixels of margin; a crop cut to rows [r0, r1) is that shorter crop pasted r0 rows lower) and fills
 * rects[i] (what af_stage_rows_u8 copies into the staging buffer, 16-byte aligned pieces), frames[i] (what af_warp_affine_clip_u8
 * reads) and *total_bytes (staging bytes used). */
typedef struct af_align_crop {
    const void* src;
    int64_t pitch;
    int32_t h, w;
    int32_t x, y;
} af_align_crop;
int af_align_plan_u8(const af_align_crop* crops, int n, int canvas_h, int canvas_w, const double* tfm, int size,
                     af_stage_rect* rects, af_align_frame* frames, int64_t* total_bytes, int32_t* bad_frame);

/* ---- whole-forward op list ------------------------------------------------------------ */

enum af_op_kind { AF_OP_STEM = 0, AF_OP_CONV = 1, AF_OP_MAXPOOL = 2, AF_OP_HEAD = 3,
                  AF_OP_PACK_F32 = 4, AF_OP_PACK_U8 = 5, AF_OP_CONV_DUAL = 6, AF_OP_STEM_POOL = 7, AF_OP_AVGPOOL = 8, AF_OP_LINEAR = 9,
                  /* FTCN-TT: in/weight/scale/shift/out as commented in af_run_ops' switch (csrc/af_api.hip) */
                  AF_OP_TSTEM = 10, AF_OP_TOKENS = 11, AF_OP_LAYERNORM = 12, AF_OP_ATTENTION = 13, AF_OP_GELU = 14,
                  /* b + c of a bottleneck in one launch: conv / weight / scale / shift = the b conv, conv2 / weight2 / scale2 /
                     shift2 = the c conv, residual, out, out_ld (af_conv3d_bc_bn_act) */
                  AF_OP_CONV_BC = 15,
                  /* the K-packed stem path: same fields as PACK_F32 / PACK_U8 / STEM_POOL, rgb3 input layout */
                  AF_OP_PACK3_F32 = 16, AF_OP_PACK3_U8 = 17, AF_OP_STEM3_POOL = 18,
                  /* c of a block + a of the next in one launch: conv / weight / scale / shift / residual / out = the c conv and
                     the trunk, conv2 / weight2 / scale2 / shift2 / aux = the a conv and its output (af_conv3d_ca_bn_act) */
                  AF_OP_CONV_CA = 19,
                  /* a narrow bottleneck block in one launch: conv / weight / scale / shift = a, conv2 / weight2 / scale2 / shift2 = b,
                     conv3 / weight3 / scale3 / shift3 = c, in = the trunk (input and residual), out, out_ld; projection form:
                     pool is unused, in2 == in and (conv4, weight4) = the shortcut conv (af_block_abc_bn_act) */
                  AF_OP_BLOCK_ABC = 20,
                  /* TSTEM with the stem's 1x3x3 / stride-2 max-pool fused behind it (af_tstem_conv_bn_pool_relu_maxpool) */
                  AF_OP_TSTEM_POOL3 = 21,
                  /* c of s2's last block + temporal pool + a of s3's block 0: fields as CONV_CA (no conv3 segment), x_sub
                     (af_conv3d_cpa_bn_act) */
                  AF_OP_CONV_CPA = 22 };

typedef struct af_op {
    int32_t kind;                    /* af_op_kind */
    int32_t out_ld;
    af_conv_desc conv;               /* STEM / CONV */
    af_pool_desc pool;               /* MAXPOOL / HEAD */
    const void* in;
    const void* weight;              /* packed conv weight | fc weight */
    const float* scale;              /* BN scale | fc bias */
    const float* shift;
    const void* residual;
    void* out;                       /* activation | logits */
    void* aux;                       /* HEAD: pooled features (optional) */
    int32_t num_classes;
    int32_t tag;                     /* caller-defined (e.g. layer class) - echoed by *_timed */
    /* CONV_DUAL only: the projection-shortcut segment */
    af_conv_desc conv2;
    const void* in2;
    const void* weight2;
    /* PACK_* only */
    int64_t in_strides[5];           /* n,c,t,h,w element strides of the fp32 source */
    float mean[3], std_[3];
    /* CONV only: split-K scratch (af_conv3d_bn_act) */
    void* workspace;
    int64_t workspace_bytes;
    /* HEAD / LINEAR only: optional per-row scores (af_avgpool_fc_scores) */
    float* scores;
    /* CONV_BC / CONV_CA: BatchNorm of the second conv */
    const float* scale2;
    const float* shift2;
    /* CONV_CA of a projection block only: the shortcut segment (NULL in3: plain block with a residual) */
    af_conv_desc conv3;
    const void* in3;
    const void* weight3;
    /* BLOCK_ABC: BatchNorm of the third conv; the projection shortcut of its block-0 form (weight4 == NULL: identity) (ABI 3) */
    const float* scale3;
    const float* shift3;
    af_conv_desc conv4;
    const void* weight4;
    /* CONV_CPA: 1 = the whole pooled trunk is stored, 2 = its even (h, w) positions, packed (ABI 3) */
    int32_t x_sub;
    int32_t reserved0;
} af_op;

/* Enqueue ops[0..n) in order on `stream` (AltFreezing: ResNet.forward, video_model_builder.py:561-578). */
int af_run_ops(const af_op* ops, int n_ops, void* stream);
/* Same, bracketing every op with hipEvents on `stream`; ms[i] = device time of op i.  Synchronises. */
int af_run_ops_timed(const af_op* ops, int n_ops, void* stream, float* ms);

#ifdef __cplusplus
}
#endif
#endif /* AF_HIP_H */
