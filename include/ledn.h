/* ledn.h -- C ABI of libledn_hip.so: the MI355X (gfx950) kernels behind the
 * LED-Net forward/backward hot path.
 *
 * The reference (ly27253/LED-Net) has NO native code and no FFI: every op on
 * its hot path is an implicit ATen/cuDNN call issued from Python
 * (SURVEY.md section 2.2).  Each entry point below therefore names the
 * reference call site (file:line under /root/reference) whose ATen op(s) it
 * replaces.  INTEGRATION.md shows the ctypes binding and the mmseg-registry
 * registration a maintainer would add on the reference side.
 *
 * Conventions
 *   - every tensor is a DEVICE pointer, dense NHWC ([N][H][W][C], C fastest);
 *   - activations are f32 or bf16 (LEDN_F32 / LEDN_BF16), parameters,
 *     statistics and accumulators are always f32;
 *   - `stream` is a hipStream_t passed as void*; all work is asynchronous on it;
 *   - every function validates its arguments on the host and returns
 *     LEDN_OK / LEDN_EINVAL / LEDN_ELAUNCH; it never launches a kernel whose
 *     grid does not match the operand shapes;
 *   - no torch types, no ownership transfer: the caller owns every buffer.
 */
#ifndef LEDN_H
#define LEDN_H

#ifdef __cplusplus
extern "C" {
#endif

#define LEDN_ABI_VERSION 5

enum { LEDN_OK = 0, LEDN_EINVAL = 1, LEDN_ELAUNCH = 2, LEDN_ESKIP = 3 /* optional fast path not applicable: use the general entry */ };
enum { LEDN_F32 = 0, LEDN_BF16 = 1, LEDN_U8 = 2 };
enum { LEDN_ACT_NONE = 0, LEDN_ACT_RELU = 1, LEDN_ACT_RELU6 = 2, LEDN_ACT_PRELU = 3, LEDN_ACT_SIGMOID = 4 };
enum { LEDN_RES_NONE = 0, LEDN_RES_ADD = 1, LEDN_RES_GATE = 2 }; /* v+res | v*res+res */

int ledn_abi_version(void);

/* Optional device scratch (f32 words) owned by the caller and used by every later call on ONE
 * stream: cross-workgroup reductions (BN statistics, weight gradients) write per-workgroup
 * partials there and finish with a second tiny kernel instead of same-address atomics.
 * ptr = NULL, nfloats = 0 detaches it (kernels then fall back to atomics). */
int ledn_set_workspace(void* ptr, long long nfloats);
/* The same scratch bound to ONE stream: launches issued with `stream` use this buffer, launches on other streams
 * their own binding (or the process default above).  The library keeps no other per-call state: the workspace
 * binding by stream and the deferred-statistics hand-off (ledn_stats_defer_begin/end: per calling host thread) are
 * re-entrant across streams and threads; the launch-shape options below are process-wide tuning knobs that never
 * change results.  ptr = NULL, nfloats = 0 removes the binding. */
int ledn_bind_workspace(void* stream, void* ptr, long long nfloats);

/* Launch-shape knobs (process-wide; defaults are tuned for a 256-CU MI355X).  value <= 0 restores
 * the default.  Results never depend on them beyond f32 summation order. */
enum {
    LEDN_OPT_CONV_WORKGROUPS = 0,   /* persistent workgroups of the MFMA conv (default 512) */
    LEDN_OPT_WGRAD_WORKGROUPS = 1,  /* pixel-range workgroups of the MFMA weight gradient (default 512) */
    LEDN_OPT_STREAM_FAST = 2,       /* bit mask, default 91.  bit 0: 16-B-per-lane streaming kernels (csrc/stream_fast.hip) for
                                       the bf16 elementwise / BatchNorm passes; bit 1: LDS-tiled depthwise 3x3
                                       (csrc/dwconv.hip); bit 2: MFMA conv tiles handed out round-robin instead of
                                       as contiguous ranges (off); bit 3: 8-row MFMA conv tiles when a launch has fewer
                                       16-row tiles than workgroups; bit 4: 1x1 stride-1 convolutions without input prologue /
                                       output affine on the register-direct streaming kernel (csrc/conv1x1.hip); bit 5: 3x3 convolutions with
                                       32 < Cin <= 64 and 64-channel output tiles keep both K-chunks' weights resident in LDS (off by default: measured 13.49 vs 13.42 ms per step);
                                       bit 6: 3x3 stride-1 convolutions with 32 input channels on the register-direct wave-autonomous kernel
                                       (csrc/conv3x3.hip), and 3x3 convolutions with <= 2 input channels and 32 k output channels (the data
                                       gradient of the two-class heads) as one K = 32 fragment per pixel; bit 7 (off): the two-class heads'
                                       forward convolution on that kernel too (measured slower than conv_mfma_kernel's narrow epilogue); bit 8 (off): 3x3 stride-1
                                       convolutions with 64 input channels through a wave-private LDS ring (conv3x3_ring64_kernel: measured 10 % slower);
                                       0: the generic kernels (A/B measurements); < 0: the default mask */
    LEDN_OPT_BN_FUSED = 4,          /* 1: ledn_bn_act_bwd_fused may launch its persistent one-pass kernel (default 0: it returns
                                       LEDN_ESKIP); see that entry point */
    LEDN_OPT_DETERMINISTIC = 3      /* 1: every cross-workgroup reduction of the library (BatchNorm statistics and their
                                       backward sums, weight / bias gradients, pooled contexts and their gradients, the
                                       attention's bias gradient and the gradients of reflect-padded windows) runs in a
                                       FIXED order: per-workgroup partial rows in the bound workspace + an ordered summing
                                       launch instead of f32 atomics, whatever the grid size; an entry point that cannot
                                       (no workspace bound, non-natural dW strides) returns LEDN_EINVAL instead of falling
                                       back to atomics.  Two runs of the same step on the same device are then bit-identical,
                                       eager or replayed from a hipGraph (PyTorch's deterministic mode, which the reference
                                       stack offers as randomness=dict(seed=.., deterministic=True): configs/LED_Net/
                                       ddrnet_23_in1k-pre_2xb6-120k_cityscapes-1024x1024.py:100).  0 (default): small grids
                                       end in bounded atomics. */
};
int ledn_set_option(int option, long long value);

/* ------------------------------------------------------------------------- *
 * Dense / grouped convolution, forward and data-gradient.
 *   z[n,ho,wo,co] = sum_{kh,kw,ci} pre(x)[n, ho*s-pad+kh*dil, wo*s-pad+kw*dil, ci] * W[co][ci][kh][kw]
 *   pre(x) = act_in((x [+ xadd]) * in_scale[ci] + in_shift[ci])   (zero padding AFTER pre)
 *   v      = z * out_scale[co] + out_shift[co]                     (stats of v -> stat_sum/stat_sqsum)
 *   y      = act_out(res_mode(v, res))
 * transposed=1 computes the data gradient of the same convolution: x is dz
 * [N,H,W,Cin] with Cin = forward Cout, y is dx [N,Ho,Wo,Cout] with Cout = forward Cin.
 * Weights are addressed by element strides so PyTorch's OIHW master copy is used
 * in place:  W(co,ci,tap) = w[co*ws_co + ci*ws_ci + tap*ws_tap]  (ci is the
 * index inside the group).
 * Replaces: F.conv2d inside mmcv ConvModule (mmseg/models/utils/basic_block.py:43-57,
 * backbones/ddrnet.py:68-105,123-138, decode_heads/led_head.py:87-94,
 * decode_head.py:158), nn_layers/espnet_utils.py:22-36 (grouped 1x1),
 * backbones/UNetFormer_GETB.py:83-85,111 (1x1 qkv / Mlp), fused with the
 * BatchNorm / ReLU / PReLU / residual ops that follow them there.
 * ------------------------------------------------------------------------- */
typedef struct {
    const void* x;
    const void* xadd;      /* optional, same shape/dtype as x */
    const float* w;
    const void* w_bf16;    /* optional: bf16 weights packed by ledn_pack_conv_weights (mode 0 for
                              forward, mode 1 for transposed); enables the MFMA implicit-GEMM
                              path when x/y are bf16, Cin%32==0, Cout%16==0, groups=1 (or a grouped 1x1) */
    void* y;
    const void* res;       /* optional [N,Ho,Wo,Cout], dtype_y */
    const float* in_scale; /* optional [Cin] */
    const float* in_shift; /* optional [Cin] */
    const float* out_scale;/* optional [Cout] (NULL = 1) */
    const float* out_shift;/* optional [Cout] (NULL = 0) */
    const float* slope;    /* [Cout] when act_out == PRELU */
    float* stat_sum;       /* optional [Cout], accumulated atomically */
    float* stat_sqsum;     /* optional [Cout] */
    long long ws_co, ws_ci, ws_tap;
    int N, H, W, Cin, Ho, Wo, Cout;
    int KH, KW, stride, pad, dil, groups;
    int in_act, act_out, res_mode;
    int dtype_x, dtype_y;
    int transposed;
    const float* in_slope; /* [Cin] when in_act == PRELU (pre(x) = prelu(x*in_scale+in_shift, in_slope)) */
} ledn_conv_desc;
int ledn_conv2d(const ledn_conv_desc* d, void* stream);
/* ledn_conv2d, but when the kernel produced its channel statistics as per-workgroup partial rows
 * ([rows][2][Cout] f32 in the workspace) the summing launch is skipped and the rows are handed to the
 * caller (*rows > 0), who must pass them to ledn_bn_finalize_rows BEFORE any other ledn call on this
 * workspace; *rows == 0: stat_sum / stat_sqsum are complete as with ledn_conv2d.  Saves one launch
 * per conv + BatchNorm pair in training (BatchNorm statistics of nn.BatchNorm2d in train mode). */
int ledn_conv2d_deferred_stats(const ledn_conv_desc* d, float** part, int* rows, void* stream);
/* The same hand-off for any statistics producer (ledn_dwconv2d, ledn_channel_stats, ...): bracket ONE
 * producer call with begin/end; end reports the rows (or rows = 0: the totals are in the producer's
 * sum / sqsum outputs as usual). */
int ledn_stats_defer_begin(void);
int ledn_stats_defer_end(float** part, int* rows);
/* Pure query, no launch: 1 if ledn_conv2d would run this descriptor on conv_mfma_kernel, 2 if on
 * conv1x1_mfma_kernel, 3 if on conv3x3_reg_kernel, 4 if on conv3x3_narrowin_mfma_kernel, 5 if on conv_f32_mfma_kernel
 * (f32 activations on v_mfma_f32_32x32x2_f32, csrc/conv_f32.hip) (all matrix cores), 0 if on
 * conv_direct_kernel / conv_narrowin_kernel (VALU).  bench.py names the kernel in its
 * roofline with it. */
int ledn_conv2d_uses_mfma(const ledn_conv_desc* d);

/* bf16 weight pack for the MFMA path, from the OIHW f32 master [Cout][Cin/groups][KH][KW]
 * (Cin = full input width; a grouped 1x1 is densified: zeros outside its group):
 *   mode 0 (forward):  out[tap][co][ci]        = w[co][ci][tap]
 *   mode 1 (dgrad):    out[KK-1-tap][ci][co]   = w[co][ci][tap]   (flipped taps, roles swapped) */
int ledn_pack_conv_weights(const float* w, void* out_bf16, int Cout, int Cin, int KH, int KW, int mode,
                           int groups, void* stream);

/* The same pack for a whole table of weights in one launch (training: once per step after SGD). */
typedef struct {
    const float* w;
    void* out;
    int Cout, Cin, KK, mode, groups;
} ledn_pack_entry;
int ledn_pack_conv_weights_multi(const ledn_pack_entry* table_dev, int n, long long max_elems, void* stream);

/* im2col of the 3-channel stem (3x3, stride 2, pad 1): p[n,ho,wo,(kh*3+kw)*C + c], bf16,
 * 32 columns (9*C used, rest zero).  The stem (ddrnet.py:123-130) then runs as a K=32 1x1
 * GEMM on the MFMA path with the weight reshaped to [Cout][32][1][1]. */
int ledn_im2col_stem(const void* x, void* p, int N, int H, int W, int C, int Ho, int Wo, void* stream);
/* The same patches straight from the planar input batch (NCHW uint8 / f32 / bf16) with the
 * SegDataPreProcessor normalisation and channel map (data_preprocessor.py:98-151) applied on the
 * way: p[.., (kh*3+kw)*C + c] = bf16(x[n, map[c], hi, wi] * scale[c] + shift[c]).  Fuses
 * ledn_nchw_to_nhwc + ledn_im2col_stem (the NHWC copy of the input is never written).
 * valid_hw (optional, int32 [N][2]): image n holds data in rows < valid_hw[2n], columns < valid_hw[2n+1] only;
 * the rest of the H x W plane is batch padding and reads as pad_val IN THE NORMALISED DOMAIN (stack_batch pads
 * after the normalisation, mmseg/utils/misc.py:77-93, data_preprocessor.py:121-133; pad_val = 0 there). */
int ledn_im2col_stem_planar(const void* x, int dtype_x, void* p, int N, int H, int W, int C, int Ho, int Wo,
                            const float* scale, const float* shift, const int* map, const int* valid_hw,
                            float pad_val, void* stream);
/* The whole first stem convolution (3x3, stride 2, pad 1, 3 -> 32; ddrnet.py:123-130) from the planar input batch in
 * ONE kernel: normalisation + channel map + batch padding (as ledn_im2col_stem_planar), im2col in LDS, K = 32 GEMM on
 * the matrix cores -- the [pixels][32] patch matrix is never written.  wp: ledn_pack_conv_weights (mode 0) of the
 * [32][32][1][1] view of the filter (columns (kh*3+kw)*3 + c, zero padded).  y [N,Ho,Wo,32] bf16.
 * Either out_scale / out_shift / act_out (inference: folded BatchNorm + ReLU) or stat_sum / stat_sqsum (training:
 * raw z and its per-channel sums, deferred-statistics aware), not both. */
int ledn_stem_conv(const void* x, int dtype_x, const void* wp, void* y, int N, int H, int W, int C, int Ho, int Wo,
                   int Cout, const float* in_scale, const float* in_shift, const int* map, const int* valid_hw,
                   float pad_val, const float* out_scale, const float* out_shift, int act_out, float* stat_sum,
                   float* stat_sqsum, void* stream);
/* Its weight gradient from the same planar batch (autograd of F.conv2d in ddrnet.py:123-130):
 *   dw[co][c][kh][kw] (f32, OIHW, [32][3][3][3]) += sum over output pixels of dz[n,ho,wo,co] * pre(x)[n,c,2ho-1+kh,2wo-1+kw]
 * with pre = the normalisation / channel map / batch padding of ledn_stem_conv (zero outside the image).  dz [N,Ho,Wo,32]
 * bf16.  Uses the bound workspace (>= 864 floats per workgroup). */
int ledn_stem_conv_wgrad(const void* x, int dtype_x, const void* dz, float* dw, int N, int H, int W, int C, int Ho, int Wo,
                         int Cout, const float* in_scale, const float* in_shift, const int* map, const int* valid_hw,
                         float pad_val, void* stream);
/* The same with dz formed INSIDE the kernel from the BatchNorm + activation behind the convolution (the stem's
 * ConvModule, ddrnet.py:123-130 -> mmcv ConvModule conv -> norm -> act): bn describes that BatchNorm's backward exactly as
 * for ledn_bn_act_bwd_apply (a later declaration: z = the convolution's output, dy = the gradient of act(BN(z)), sum_g /
 * sum_gx = the totals of ledn_bn_act_bwd_reduce, SyncBN-reduced if applicable; act none / ReLU, no residual, no dz_add,
 * bf16, C = 32) but bn->dz is NOT written: the stem's input needs no gradient, so this weight gradient is the only reader
 * of dz, and the apply pass (two tensors read, one written) collapses into one more tensor read here. */
struct ledn_bnbwd_desc_s;
int ledn_stem_conv_wgrad_bn(const void* x, int dtype_x, const struct ledn_bnbwd_desc_s* bn, float* dw, int N, int H, int W, int C,
                            int Ho, int Wo, int Cout, const float* in_scale, const float* in_shift, const int* map,
                            const int* valid_hw, float pad_val, void* stream);

/* Weight (and bias) gradient of the same convolution:
 *   dw(co,ci,tap) += sum_{n,ho,wo} pre(x)[n, ho*s-pad+kh*dil, .., ci] * dz[n,ho,wo,co]
 *   db[co]        += sum dz[n,ho,wo,co]
 * dw uses the same stride addressing as w above; dw/db are ACCUMULATED (caller zeroes). */
typedef struct {
    const void* x;
    const void* xadd;
    const void* dz;        /* [N,Ho,Wo,Cout] */
    float* dw;
    float* db;             /* optional [Cout] */
    const float* in_scale;
    const float* in_shift;
    long long ws_co, ws_ci, ws_tap;
    int N, H, W, Cin, Ho, Wo, Cout;
    int KH, KW, stride, pad, dil, groups;
    int in_act;
    int dtype_x, dtype_dz;
    const float* in_slope; /* [Cin] when in_act == PRELU */
} ledn_wgrad_desc;
int ledn_conv2d_wgrad(const ledn_wgrad_desc* d, void* stream);
int ledn_conv2d_wgrad_uses_mfma(const ledn_wgrad_desc* d);   /* same query for the weight gradient: 1 conv_wgrad_mfma_kernel, 2 conv3x3_wgrad_narrow_kernel, 3 conv1x1_wgrad_reg_kernel (csrc/conv3x3.hip), 4 conv_wgrad_f32_mfma_kernel (f32 activations, csrc/conv_f32.hip), 0 the VALU kernels */
/* Deferred reduction of the weight gradient.  ledn_conv2d_wgrad runs the MFMA kernel (per-workgroup partial tiles into the
 * stream's workspace) and then a small summing launch -- ~55 of them per training step, each on the critical path of the
 * stream although nothing reads dW before the optimizer.  ledn_conv2d_wgrad_partial instead writes the partial tiles into a
 * CALLER-OWNED buffer (`part`, at least ledn_conv2d_wgrad_partial_floats(d) floats, private to this convolution until the
 * finish) and fills `entry`; ledn_conv2d_wgrad_finish_multi then sums every recorded convolution into its dW in ONE launch
 * (table_dev: the entries in device memory with chunk0 = running sum of pairs * KK * 16 over the entries before,
 * total_chunks = that sum over all).  Bias gradients (d->db) are formed at once as in ledn_conv2d_wgrad.
 * ledn_conv2d_wgrad_partial_floats returns 0 when the descriptor does not take the MFMA path (use ledn_conv2d_wgrad). */
typedef struct {
    const float* part;
    float* dw;
    long long ws_co, ws_ci, ws_tap;
    int nbx, pairs, KK, ci_tiles, Cin, Cout, groups, chunk0;
} ledn_wgrad_finish_entry;
long long ledn_conv2d_wgrad_partial_floats(const ledn_wgrad_desc* d);
int ledn_conv2d_wgrad_partial(const ledn_wgrad_desc* d, float* part, long long part_floats,
                              ledn_wgrad_finish_entry* entry, void* stream);
int ledn_conv2d_wgrad_finish_multi(const ledn_wgrad_finish_entry* table_dev, int n, int total_chunks, void* stream);

/* ------------------------------------------------------------------------- *
 * Depthwise convolution (KxK, per-channel-group dilation, pad = dil*(K-1)/2
 * unless `pad` >= 0 is given), with the same post-op as ledn_conv2d.
 *   channel c uses dilation dil[c / group_size] (up to 4 groups).
 *   ext1=1: the input is read as if reflect-extended by one row/column at the
 *   bottom/right (index H -> H-2, W -> W-2) -- GETB `pad_out`.
 *   w is [KH][KW][C] f32 (packed by the host from PyTorch's [C][1][KH][KW]).
 * Replaces: CDilated depthwise convs of SESP (nn_layers/eesp.py:60-68,92-97)
 * and SeparableConvBN's depthwise 8x8 (backbones/UNetFormer_GETB.py:59-65,118,160-162,201-204).
 * ------------------------------------------------------------------------- */
typedef struct {
    const void* x;         /* [N,H,W,C] */
    const float* w;
    void* y;               /* [N,Ho,Wo,C] */
    const float* out_scale;
    const float* out_shift;
    const float* slope;
    float* stat_sum;
    float* stat_sqsum;
    int N, H, W, C, Ho, Wo;
    int KH, KW, stride, pad;     /* pad < 0: per-group dil*(K-1)/2 */
    int dil[4], group_size;
    int act_out, ext1;
    int dtype_x, dtype_y;
} ledn_dw_desc;
int ledn_dwconv2d(const ledn_dw_desc* d, void* stream);

/* Layout bridge between PyTorch's depthwise filters [n_k][1][KH][KW] (the state_dict layout of
 * CDilated.conv.weight, nn_layers/espnet_utils.py:120-142, and SeparableConvBN's depthwise,
 * UNetFormer_GETB.py:59-65) and the channel-last filter banks the depthwise kernels read:
 *   stacked = 0:  packed[tap][c0_k + c]      (filters concatenated along channels, ledn_dwconv2d)
 *   stacked = 1:  packed[k][tap][c]          (one bank per branch, ledn_sesp_pyramid; all n_k equal)
 * ledn_dw_pack fills `packed` from w[k]; ledn_dw_unpack_grad ADDS the gradient of `packed` to
 * dw[k] in PyTorch's layout (the trainer's gradient buffer).  One launch each instead of a
 * select/permute/stack chain per filter. */
typedef struct {
    const float* w[8];
    float* dw[8];
    int n[8];
    int nsrc, taps, stacked;
} ledn_dwpack_desc;
int ledn_dw_pack(const ledn_dwpack_desc* d, float* packed, void* stream);
int ledn_dw_unpack_grad(const ledn_dwpack_desc* d, const float* dpacked, void* stream);
/* The same for a whole DEVICE table of banks in one launch (training: once at the start of a step, once at the
 * end of its backward): dir 0: packed <- filters; dir 1: filter gradients (d.dw) += dpacked, dpacked re-zeroed.
 * max_elems = the largest n[k] * taps of the table. */
typedef struct {
    ledn_dwpack_desc d;
    float* packed;
    float* dpacked;
} ledn_dwpack_entry;
int ledn_dw_repack_multi(const ledn_dwpack_entry* table_dev, int n, int max_elems, int dir, void* stream);

/* Train-time augmentation on the GPU: one launch builds the padded uint8 batch
 *   out_img [n][3][OH][OW] (BGR planes, as PackSegInputs emits them) and out_seg [n][OH][OW] (int64, may be NULL)
 * from the decoded source images: resize (cv2 INTER_LINEAR 8-bit fixed point; labels: INTER_NEAREST) -> crop ->
 * horizontal flip -> PhotoMetricDistortion, with every intermediate 8-bit rounding of the CPU pipeline; pixels
 * outside the crop extent (ch, cw) get pad_val / seg_pad_val (stack_batch).  The random parameters are drawn on the
 * host in the reference's numpy.random order (led_net_amd/transforms.py) and passed per image in a device table.
 * Replaces: RandomResize -> RandomCrop -> RandomFlip -> PhotoMetricDistortion -> PackSegInputs of
 * configs/_base_/datasets/pascal_voc12.py:6-18 (mmseg/datasets/transforms/transforms.py:208-337, 583-750, 1013-1033,
 * formatting.py:50-107; mmcv Resize, cv2.resize, cv2.cvtColor for the un-vendored parts).
 * ledn_aug_crop_hist: per-image class counts hist[n][256] (caller zeroes) of the candidate crop of the resized label
 * map: RandomCrop's cat_max_ratio test (transforms.py:287-296, np.unique on the crop). */
typedef struct {
    const unsigned char* img;   /* source image, uint8 H x W x 3 (BGR as decoded), device memory */
    const unsigned char* seg;   /* source label map, uint8 H x W, device memory (NULL: no labels) */
    int H, W;                   /* source size */
    int RH, RW;                 /* size after the resize (== H, W: no resize) */
    double sx, sy;              /* cv2's inverse scale: 1 / ((double)RW / W), 1 / ((double)RH / H) */
    int oy, ox;                 /* crop offset inside the resized image */
    int ch, cw;                 /* crop extent = valid extent of the output (<= OH, OW) */
    int flip;                   /* 1: horizontal flip of the crop */
    int bright_on;
    float bright_beta;          /* convert(img, beta=...) */
    int contrast_mode;          /* PhotoMetricDistortion's `mode`: 1 = contrast before saturation, 0 = last */
    int contrast_on;
    float contrast_alpha;
    int sat_on;
    float sat_alpha;
    int hue_on;
    int hue_delta;
} ledn_aug_entry;
int ledn_augment_batch(const ledn_aug_entry* table_dev, int n, unsigned char* out_img, long long* out_seg, int OH,
                       int OW, int pad_val, int seg_pad_val, void* stream);
int ledn_aug_crop_hist(const ledn_aug_entry* table_dev, int n, int max_pixels, int* hist, void* stream);

/* SESP split/transform stage 1 with hierarchical feature fusion:
 *   y[..., b*n + c] = sum_{b' <= b} dw3x3_{dil[b'], stride}(x)[..., c]     b = 0..3
 * x [N,H,W,n], w [4][3][3][n], y [N,Ho,Wo,4n].
 * Replaces: nn_layers/eesp.py:81-91 (4 CDilated convs + 3 adds + later torch.cat). */
typedef struct {
    const void* x;
    const float* w;
    void* y;
    int N, H, W, n, Ho, Wo, stride;
    int dil[4];
    int dtype_x, dtype_y;
} ledn_pyr_desc;
int ledn_sesp_pyramid(const ledn_pyr_desc* d, void* stream);

/* ------------------------------------------------------------------------- *
 * Per-channel statistics and elementwise affine / activation.
 * Replaces: nn.BatchNorm2d (+ReLU/PReLU/ReLU6, + residual add) wherever it is
 * not fused into a convolution above (eesp.py:99-103, espnet_utils.py:53-57,
 * UNetFormer_GETB.py:212,219, led_head.py:87-96).
 * ------------------------------------------------------------------------- */
/* sum[c] += sum_p x[p,c], sqsum[c] += sum_p x[p,c]^2   (x: [P][C]) */
int ledn_channel_stats(const void* x, const void* xadd, long long P, int C, int dtype,
                       float* sum, float* sqsum, void* stream);
/* batch-norm bookkeeping from accumulated sums (count = elements per channel):
 *   mean, biased var -> scale = gamma*invstd, shift = beta - mean*scale;
 *   running stats updated with `momentum` (unbiased var), if given. */
int ledn_bn_finalize(const float* sum, const float* sqsum, double count, const float* gamma,
                     const float* beta, float* running_mean, float* running_var, float momentum,
                     float eps, float* scale, float* shift, float* mean, float* invstd, int C,
                     void* stream);
/* y = act(res_mode((x [+ xadd]) * scale[c] + shift[c], res))   x,y: [P][C] */
typedef struct {
    const void* x;
    const void* xadd;
    void* y;
    const void* res;
    const float* scale;
    const float* shift;
    const float* slope;
    long long P;
    int C, act, res_mode;
    int dtype_x, dtype_y;
    /* optional: per-channel sums of the OUTPUT y (as stored: rounded to dtype_y), ACCUMULATED: stat_sum[c] += sum_p y[p,c],
     * stat_sqsum[c] += sum_p y[p,c]^2 -- the batch statistics a BatchNorm on y needs (LEDHead's norm -> act -> conv heads
     * on the stem maps, led_head.py:44-51), from the pass that writes y instead of a pass of their own.  Only with bf16
     * x / y, no xadd / residual, C a power of two in 8 .. 512 and >= 4096 vectors (else LEDN_EINVAL: call
     * ledn_channel_stats).  Partial rows in the bound workspace, added up in row order. */
    float* stat_sum;
    float* stat_sqsum;
} ledn_affine_desc;
/* ledn_bn_finalize from statistic rows part[rows][2][C] (see ledn_conv2d_deferred_stats): sums the rows and
 * finalizes in one launch; sum / sqsum (optional) receive the totals. */
int ledn_bn_finalize_rows(const float* part, int rows, double count, const float* gamma, const float* beta,
                          float* running_mean, float* running_var, float momentum, float eps, float* scale,
                          float* shift, float* mean, float* invstd, float* sum, float* sqsum, int C,
                          void* stream);
int ledn_affine_act(const ledn_affine_desc* d, void* stream);

/* Planar-to-interleaved input transform (the step in front of the stem):
 *   y[n,h,w,c] = x[n, map[c], h, w] * scale[c] + shift[c]
 * x: [N][C][H][W] u8 / f32 / bf16, y: [N][H][W][C] f32 / bf16; map NULL = identity.
 * Replaces: SegDataPreProcessor's BGR->RGB flip and (x-mean)/std
 * (mmseg/models/data_preprocessor.py:117-127) plus the NCHW->NHWC re-layout; valid_hw / pad_val as in
 * ledn_im2col_stem_planar (the batch padding of stack_batch, applied after the normalisation). */
int ledn_nchw_to_nhwc(const void* x, int dtype_x, void* y, int dtype_y, int N, int C, int H, int W,
                      const float* scale, const float* shift, const int* map, const int* valid_hw, float pad_val,
                      void* stream);

/* ------------------------------------------------------------------------- *
 * Resampling.
 * ------------------------------------------------------------------------- */
/* y = (add ? add : 0) + bilinear(x -> Ho x Wo), align_corners=False semantics of
 * F.interpolate (mmseg/models/utils/wrappers.py:27; call sites ddrnet.py:195-199,
 * led_head.py:106-138, decode_head.py:364-378).  out_nchw=1 writes y as
 * [N][C][Ho][Wo] f32 and, if argmax != NULL, the first-max channel index per
 * pixel (segmentors/base.py:188) as uint8. */
typedef struct {
    const void* x;         /* [N,H,W,C] */
    const void* add;       /* optional [N,Ho,Wo,C] dtype_y (NHWC) */
    void* y;
    unsigned char* argmax; /* optional [N,Ho,Wo] */
    int N, H, W, C, Ho, Wo;
    int out_nchw;
    int dtype_x, dtype_y;
} ledn_resize_desc;
int ledn_bilinear(const ledn_resize_desc* d, void* stream);

/* y[n,oy,ox,c] = mean over the adaptive window of (x [+ xadd]); y f32 [N,S,S,C].
 * Replaces nn.AdaptiveAvgPool2d in Muti_AFF (classification/model_utils.py:373-400). */
int ledn_adaptive_avgpool(const void* x, const void* xadd, float* y, int N, int H, int W, int C,
                          int S, int dtype, void* stream);
/* 3x3 stride-2 pad-1 average pool, count_include_pad (nn_layers/eesp.py:74,111). */
int ledn_avgpool3x3s2(const void* x, void* y, int N, int H, int W, int C, int Ho, int Wo,
                      int dtype, void* stream);
/* k x k average pool, stride s, zero padding p counted in the divisor (nn.AvgPool2d defaults): the pooling
 * pyramid of DAPPM / PAPPM (mmseg/models/utils/ppm.py:66-70: (5,2,2), (9,4,4), (17,8,8)).
 * Ho = (H + 2p - k)/s + 1 (floor).  _bwd: dx = adjoint (gather form). */
int ledn_avgpool2d(const void* x, void* y, int N, int H, int W, int C, int Ho, int Wo, int k, int stride, int pad,
                   int dtype, void* stream);
int ledn_avgpool2d_bwd(const void* dy, void* dx, int N, int H, int W, int C, int Ho, int Wo, int k, int stride,
                       int pad, int dtype, void* stream);

/* ------------------------------------------------------------------------- *
 * GETB (backbones/UNetFormer_GETB.py:97-206).
 * ------------------------------------------------------------------------- */
/* 8x8-window multi-head attention on a 1x1-conv qkv map.
 *   qkv [N,H,W,3*C] (channel = which*C + head*d + j), reflect-extended to
 *   multiples of ws at the bottom/right (:145-152); out [N,H,W,C] (cropped, :195).
 *   biasT [heads][ws*ws (key j)][ws*ws (query i)] f32: relative-position bias
 *   gathered by the host from the (2ws-1)^2 x heads table (:181-187). */
int ledn_window_attn(const void* qkv, const float* biasT, void* out, int N, int H, int W, int C,
                     int heads, int ws, int dtype, void* stream);
/* The relative-position bias operand of ledn_window_attn from the module's parameters
 * (UNetFormer_GETB.py:181-187): biasT[h][j][i] = table[index[i*T + j]][h]; table [R = (2ws-1)^2][heads] f32,
 * index [T*T] int64 (the module's registered buffer), T = ws*ws.  _bwd: dtable[r][h] += sum of dbiasT over the
 * token pairs that index row r (gather form, deterministic).  Replaces the host-side index / index_put chain. */
int ledn_relpos_bias(const float* table, const long long* index, float* biasT, int R, int heads, int T, void* stream);
int ledn_relpos_bias_bwd(const float* dbiasT, const long long* index, float* dtable, int R, int heads, int T,
                         void* stream);
/* out = avgpool_(ws,1)(reflect-pad-bottom(a)) + avgpool_(1,ws)(reflect-pad-right(a)) + local
 * (:197-199); a, local, out: [N,H,W,C]. */
int ledn_getb_pool(const void* a, const void* local, void* out, int N, int H, int W, int C, int ws,
                   int dtype, void* stream);

/* ------------------------------------------------------------------------- *
 * MFAF gate (classification/model_utils.py:405-429):
 *   wei = sigmoid( aff0(xl) + aff1(c1)^ + aff2(c2)^ + aff3(c3)^ + aff4(xg) ),  ^ = nearest upsample
 *   out = 2*x*wei + 2*r*(1-wei)
 * xl [N,H,W,C] (dtype), c1 [N,4,4,C], c2 [N,8,8,C], c3 [N,16,16,C], xg [N,1,1,C] (f32),
 * aff: 5 pairs (scale,shift) of [C] f32 = the trailing BatchNorm of each branch. */
typedef struct {
    const void* x;
    const void* r;
    const void* xl;
    const float* ctx[4];   /* c1, c2, c3, xg */
    int ctx_size[4];       /* 4, 8, 16, 1 */
    const float* scale[5]; /* xl, c1, c2, c3, xg */
    const float* shift[5];
    void* out;
    int N, H, W, C;
    int dtype;
    int act;               /* LEDN_ACT_NONE or LEDN_ACT_RELU applied to out (the next stage's ReLU) */
} ledn_mfaf_desc;
int ledn_mfaf_gate(const ledn_mfaf_desc* d, void* stream);

/* ------------------------------------------------------------------------- *
 * Evaluation histograms (IoUMetric.intersect_and_union,
 * mmseg/evaluation/metrics/iou_metric.py:163-199) straight from the uint8 argmax mask the fused
 * last resize writes (ledn_bilinear argmax) and the int64 label map: over pixels with
 * label != ignore_index, hist[0][c] += [pred == label == c], hist[1][c] += [pred == c],
 * hist[2][c] += [label == c] (f32 counts, ACCUMULATED: the caller zeroes; union = [1]+[2]-[0]).
 * Labels >= num_classes that are not ignore_index are counted nowhere (torch.histc drops them).
 * num_classes <= 256. */
int ledn_iou_hist(const unsigned char* pred, const long long* label, long long P, int num_classes,
                  int ignore_index, float* hist, void* stream);

/* ------------------------------------------------------------------------- *
 * SEAM edge map (prototype tools/speed/ddrnet_speed.py:24-37,282-338; percentile
 * rule: supplementary PDF section 4.2 eq.1).
 *   seg [N,h,w] f32 -> per-image min-max normalise -> Laplacian at strides 1/2/4,
 *   clamp >= 0, nearest upsample -> binarise (b > T; T = the kth smallest response
 *   (1-based; kth = ceil(q*h*w) for the q-percentile rule) per image and scale
 *   if kth > 0, else the fixed threshold `thr`) -> 0.6/0.3/0.1 fuse ->
 *   binarise (> final_thr) -> edge [N,h,w] f32 in {0,1}.
 *   scratch: 3*N*h*w floats + 16*N words.  One workgroup per image. */
int ledn_seam_edge(const float* seg, float* edge, float* scratch, int N, int h, int w, int kth,
                   float thr, float final_thr, void* stream);

/* ========================================================================= *
 * Backward kernels (training path).  The reference gets all of these from
 * torch.autograd over the ATen ops of its forward (tools/train.py:99-106 ->
 * mmengine OptimWrapper.update_params -> loss.backward()).
 * ========================================================================= */

/* Backward of  y = act(res_mode((z*scale + shift), res))  where, for bn_mode=1,
 * scale/shift are the batch-statistics affine of BatchNorm (x_hat = (z-mean)*invstd).
 *   t = res_mode(v, res), g_t = dy * act'(t);  ADD: g_v = g_t, dres = g_t;
 *   GATE (t = v*res + res): g_v = g_t*res, dres = g_t*(v+1).
 * reduce pass : sum_g[c] += sum g_v, sum_gx[c] += sum g_v*x_hat  (= dbeta, dgamma),
 *               dslope[c] += sum dy*min(t,0)                      (PReLU)
 * apply pass  : bn_mode=1: dz = scale*(g_v - sum_g/count - x_hat*sum_gx/count)
 *               bn_mode=0: dz = g_v*scale (scale NULL = 1);  dres written if given. */
typedef struct ledn_bnbwd_desc_s {
    const void* z;
    const void* res;
    const void* dy;
    const float* scale;
    const float* shift;
    const float* slope;
    const float* mean;
    const float* invstd;
    float* sum_g;
    float* sum_gx;
    float* dslope;
    void* dz;
    void* dres;
    double count;
    long long P;
    int C, act, res_mode, bn_mode;
    int dtype_z, dtype_y;
    /* apply pass, optional: partial gradients already computed for the SAME tensors by another consumer of z / res
     * (a tensor with several consumers in the forward): dz += dz_add (dtype_z), dres += dres_add (dtype_y).  Replaces
     * the separate elementwise add of autograd's gradient fan-in. */
    const void* dz_add;
    const void* dres_add;
    /* optional, both passes: ZEROED f32 scratch [LEDN_BNBWD_ROWS][3][C].  When given (and the streaming bf16 kernels
     * take the call) the reduce pass adds its per-workgroup sums into row (workgroup % LEDN_BNBWD_ROWS) with float
     * atomics and launches NO summing kernel; the apply pass adds the rows up itself (sum_g / sum_gx / dslope then
     * receive the totals from its first workgroup).  One launch less per BatchNorm backward on the critical stream;
     * the summation order over rows is fixed, within a row it is the atomics' arrival order. */
    float* rows;
} ledn_bnbwd_desc;
enum { LEDN_BNBWD_ROWS = 32 };
int ledn_bn_act_bwd_reduce(const ledn_bnbwd_desc* d, void* stream);
int ledn_bn_act_bwd_apply(const ledn_bnbwd_desc* d, void* stream);
/* Both halves in ONE persistent launch (csrc/stream_fast.hip, bn_bwd_fused_kernel): z is read from memory once and kept in
 * LDS across a grid-wide barrier (agent-scope release / acquire on an arrival counter in the bound workspace, bounded
 * spin), dy a second time from the Infinity Cache; the [3][C] totals are float atomics.  Same descriptor and results as
 * reduce + apply (sum_g / sum_gx / dslope accumulate the totals; summation order = atomics' arrival order).
 * Returns LEDN_ESKIP -- nothing launched, call reduce + apply -- unless: bf16 tensors, C a power of two in 8..128,
 * 64 K <= P*C/8 <= 2 M vectors, no `rows`, act in {none, ReLU, PReLU}, res_mode in {none, add}, a bound workspace,
 * >= 256 compute units, not deterministic mode, and LEDN_OPT_BN_FUSED switched on.  The caller must launch it from ONE
 * stream at a time (its workgroups wait for each other: two such launches running concurrently could starve each other
 * of compute units) and never between the two halves of a SyncBN all-reduce.
 * ledn_bn_act_bwd_fused_check: after a stream synchronisation, nonzero = the last fused launch gave up at its barrier
 * (results of that launch are invalid); the fused form then stays off for the rest of the process. */
int ledn_bn_act_bwd_fused(const ledn_bnbwd_desc* d, void* stream);
int ledn_bn_act_bwd_fused_check(int C, void* stream);

/* BatchNorm / activation backward of LEDHead's two-class heads -- norm -> act -> 3x3 conv 32 -> 2, stride 1, pad 1
 * (mmseg/models/decode_heads/led_head.py:44-51, applied to the 1/2- and 1/4-resolution stem maps at :93-98) -- straight
 * from the gradient of the head's LOGITS: the gradient of the activation dy = conv_transpose3x3(head_dz, w) is a function
 * of the 3 x 3 x 2 patch of head_dz and is recomputed on the matrix cores inside both passes, never written (autograd of
 * the reference materialises it: F.conv2d backward -> F.relu backward -> F.batch_norm backward).  Replaces, for these
 * layers, ledn_conv2d(transposed) + ledn_bn_act_bwd_reduce + ledn_bn_act_bwd_apply -- and ledn_conv2d_wgrad when dw is
 * given: the reduce pass then also accumulates the head's weight / bias gradient (input act(BatchNorm(x)) rounded to bf16
 * as in ledn_conv2d_wgrad's prologue; one partial row per workgroup, added up in row order: bit-reproducible).
 *   bn: the descriptor of ledn_bn_act_bwd_reduce / _apply with z = the head's input x (bf16), dy = NULL, res_mode = none,
 *       act in {none, ReLU, PReLU}; scale / shift / mean / invstd = the forward's batch-statistics affine; dz = the
 *       gradient of x (apply pass, + dz_add when given); sum_g / sum_gx (/ dslope) accumulate d beta / d gamma (/ d slope)
 *   reduce pass: sum_g, sum_gx, dslope exactly as ledn_bn_act_bwd_reduce (one partial row per workgroup in the bound
 *       workspace + the summing launch: fixed order in deterministic mode)
 *   apply pass: dz = scale * (g - sum_g / count - xhat * sum_gx / count) (+ dz_add), g = dy * act'.  Between the passes the
 *       caller may all-reduce sum_g / sum_gx (SyncBN) exactly as with the layer-wise pair.
 * w is rounded to bf16 for the matrix instruction (as in the MFMA data-gradient kernels it replaces); dy stays f32
 * (the layer-wise form rounds it to bf16 once).
 * Supported (ledn_head_bwd_supported != 0): Co = 2, C = 32, bf16 x and head_dz, >= 16384 pixels, a bound workspace;
 * otherwise LEDN_EINVAL and the caller uses the layer-wise entries. */
typedef struct {
    ledn_bnbwd_desc bn;
    const void* head_dz;    /* [N][H][W][Co] bf16 (dtype_dz): gradient of the head's logits */
    const float* w;         /* [Co][C][3][3] f32 */
    float* dw;              /* optional (reduce pass): [Co][C][3][3] f32, the head's weight gradient ADDED */
    float* db;              /* optional, with dw: [Co] f32, the bias gradient ADDED */
    int N, H, W, Co;
    int dtype_dz;
} ledn_headbwd_desc;
int ledn_head_bwd_supported(const ledn_headbwd_desc* d);
int ledn_head_bwd_reduce(const ledn_headbwd_desc* d, void* stream);
int ledn_head_bwd_apply(const ledn_headbwd_desc* d, void* stream);

/* Depthwise convolution backward (geometry as ledn_dw_desc):
 *   data  : dx[N,H,W,C]  = sum_taps dz[...] * w   (+ `add` if given; ext1 folds the
 *           reflected row/column back onto H-2 / W-2)
 *   weight: dw[KH][KW][C] += sum_pix x[pix@tap] * dz[pix]      (caller zeroes dw) */
typedef struct {
    const void* x;
    const void* dz;
    const float* w;
    const void* add;
    void* dx;
    float* dw;
    int N, H, W, C, Ho, Wo;
    int KH, KW, stride, pad;
    int dil[4], group_size;
    int ext1;
    int dtype;
} ledn_dwbwd_desc;
int ledn_dwconv2d_bwd_data(const ledn_dwbwd_desc* d, void* stream);
int ledn_dwconv2d_bwd_weight(const ledn_dwbwd_desc* d, void* stream);

/* SESP pyramid backward.  g = suffix sums of dy over the 4 branch groups
 * (g_b = sum_{b'>=b} dy_b', the adjoint of the HFF adds, eesp.py:84-91):
 *   ledn_sesp_pyramid_bwd_data  : gsum [N,Ho,Wo,4n] (scratch, written) and dx [N,H,W,n]
 *   ledn_sesp_pyramid_bwd_weight: dw [4][3][3][n] += sum x[pix@tap,dil_b] * gsum_b[pix]
 * Call bwd_data before bwd_weight with the SAME descriptor (dy must stay valid for both).  gsum is scratch owned by
 * the pair: on the spatial branch's shape class (bf16, stride 1, dilations 1, >= 16384 pixels) the data gradient reads
 * dy through an LDS-staged patch with prefix-summed filters and the weight gradient forms the suffix sums from dy
 * itself -- gsum is then left untouched. */
typedef struct {
    const void* x;
    const void* dy;
    const float* w;
    void* gsum;
    void* dx;
    float* dw;
    int N, H, W, n, Ho, Wo, stride;
    int dil[4];
    int dtype;
} ledn_pyrbwd_desc;
int ledn_sesp_pyramid_bwd_data(const ledn_pyrbwd_desc* d, void* stream);
int ledn_sesp_pyramid_bwd_weight(const ledn_pyrbwd_desc* d, void* stream);

/* dx[N,H,W,C] = adjoint of ledn_bilinear (gather form, deterministic); dy [N,Ho,Wo,C]. */
int ledn_bilinear_bwd(const void* dy, void* dx, int N, int H, int W, int C, int Ho, int Wo,
                      int dtype_dy, int dtype_dx, void* stream);
/* dx = (add ? add : 0) + adjoint of ledn_avgpool3x3s2 applied to dy [N,Ho,Wo,C]. */
int ledn_avgpool3x3s2_bwd(const void* dy, const void* add, void* dx, int N, int H, int W, int C,
                          int Ho, int Wo, int dtype, void* stream);

/* Window attention backward.  dqkv [N,H,W,3C] f32 MUST be zeroed by the caller when
 * the map is reflect-padded (H or W not a multiple of ws): padded tokens add into
 * their source pixel.  dbiasT [heads][ws^2][ws^2] f32 is accumulated (caller zeroes).
 * Head dimension C / heads = 16 (LED-Net: 128 channels, 8 heads; UNetFormer_GETB.py:104-106) or 32; other head
 * dimensions return LEDN_EINVAL (the head-dim-8 instance spilled 1.8-6 KB of registers per lane and was removed). */
int ledn_window_attn_bwd(const void* qkv, const float* biasT, const void* dout, float* dqkv,
                         float* dbiasT, int N, int H, int W, int C, int heads, int ws, int dtype,
                         void* stream);
/* da = adjoint of ledn_getb_pool wrt `a` (dlocal = dout needs no kernel). */
int ledn_getb_pool_bwd(const void* dout, void* da, int N, int H, int W, int C, int ws, int dtype,
                       void* stream);

/* MFAF gate backward (forward quantities recomputed from x, r, xl, ctx, affines):
 *   dx_b = 2*w*dout, dr_b = 2*(1-w)*dout, ds = 2*(x-r)*dout*w*(1-w)   [N,H,W,C]
 *   dctx[k][n,sy,sx,c] += sum of ds over the pixels that read that cell (caller zeroes). */
typedef struct {
    const void* x;
    const void* r;
    const void* xl;
    const float* ctx[4];
    int ctx_size[4];
    const float* scale[5];
    const float* shift[5];
    const void* dout;
    void* dx;
    void* dr;
    void* ds;
    float* dctx[4];
    int N, H, W, C;
    int dtype;
    int act;               /* forward applied ReLU to out */
} ledn_mfafbwd_desc;
int ledn_mfaf_gate_bwd(const ledn_mfafbwd_desc* d, void* stream);
/* dx = dx_b + dxa, dr = dr_b + dxa with dxa = dxl + sum_k adaptive-avg-pool adjoint of dpool[k]
 * (dpool[k]: [N,S_k,S_k,C] f32).  In place on dx/dr (dr may be NULL). */
int ledn_mfaf_bwd_combine(void* dx, void* dr, const void* dxl, const float* const* dpool,
                          const int* sizes, int npool, int N, int H, int W, int C, int dtype,
                          void* stream);

/* ------------------------------------------------------------------------- *
 * OHEM cross-entropy + accuracy (mmseg/models/losses/ohem_cross_entropy_loss.py:52-90,
 * losses/accuracy.py:6-60), fused: softmax prob of the target class, per-pixel CE,
 * exact k-th smallest prob by 3-pass radix select on the f32 bit patterns,
 * thr = max(kth, thres), masked mean.  logits [P][C] f32 (NHWC), target [P] int64.
 *   work: float[2*P] + 4096 words of scratch (prob | loss | histograms | state).
 *   out[0] = loss_weight * mean(loss[valid & prob < thr])  (0 if no valid pixel)
 *   out[1] = top-1 accuracy in percent over valid pixels;  out[2] = thr; out[3] = #selected
 * ledn_ohem_ce_bwd: dlogits[p][c] = dloss*loss_weight/#selected * (softmax_c - onehot_c)
 * for selected pixels, 0 elsewhere. */
int ledn_ohem_ce_fwd(const float* logits, const long long* target, long long P, int C, float thres,
                     long long min_kept, float loss_weight, int ignore_label, float* work,
                     float* out, void* stream);
int ledn_ohem_ce_bwd(const float* logits, const long long* target, long long P, int C,
                     int ignore_label, const float* work, const float* out, const float* dloss,
                     float loss_weight, float* dlogits, void* stream);
long long ledn_ohem_work_floats(long long P);
/* The same loss on logits that are the bilinear (align_corners=False) resize of src [N,Hs,Ws,2] (f32) to H x W, without
 * materialising them: LEDHead.loss_by_feat's resize of each fused output to the label size (led_head.py:132-138)
 * folded into the loss.  target [N,H,W]; work: ledn_ohem_work_floats(N*H*W); out as ledn_ohem_ce_fwd.
 * ledn_ohem_ce_up_bwd (exact 2x only: H = 2 Hs, W = 2 Ws): dsrc [N,Hs,Ws,2] = resize^T(dlogits), dlogits never written. */
int ledn_ohem_ce_up_fwd(const float* src, int N, int Hs, int Ws, int H, int W, const long long* target, float thres,
                        long long min_kept, float loss_weight, int ignore_label, float* work, float* out, void* stream);
int ledn_ohem_ce_up_bwd(const float* src, int N, int Hs, int Ws, int H, int W, const long long* target,
                        int ignore_label, const float* work, const float* out, const float* dloss, float loss_weight,
                        float* dsrc, void* stream);
/* BOTH losses of LEDHead.loss_by_feat (led_head.py:132-146: loss_context = Ohem_0(resize(ctx)), loss_spatial =
 * Ohem_1(resize(spa)), same labels) in one launch set: src0 / src1 [N,Hs,Ws,2] f32 are the two fused outputs at half
 * the label size, target [N,H,W] int64 (W % 4 == 0, class ids and ignore_label < 256).  The int64 labels are read
 * once (a uint8 copy inside `work` serves the later passes and the backward), no per-pixel loss array is kept, both
 * radix selects / masked means run per launch.  work: ledn_ohem2_work_floats(N*H*W) floats; out[2][4] = per loss
 * {loss, accuracy of output 0, threshold, #selected} as ledn_ohem_ce_fwd.  Selection semantics per loss exactly
 * ohem_cross_entropy_loss.py:62-90.  ledn_ohem2_up_bwd (H = 2 Hs, W = 2 Ws): dsrc_k = resize^T(dloss_k * loss_weight_k
 * / #selected_k * (softmax - onehot)) over the selected pixels of loss k; target is not needed again. */
int ledn_ohem2_up_fwd(const float* src0, const float* src1, int N, int Hs, int Ws, int H, int W,
                      const long long* target, float thres0, long long min_kept0, float loss_weight0, float thres1,
                      long long min_kept1, float loss_weight1, int ignore_label, float* work, float* out, void* stream);
int ledn_ohem2_up_bwd(const float* src0, const float* src1, int N, int Hs, int Ws, int H, int W, int ignore_label,
                      const float* work, const float* out, const float* dloss0, const float* dloss1,
                      float loss_weight0, float loss_weight1, float* dsrc0, float* dsrc1, void* stream);
long long ledn_ohem2_work_floats(long long P);

/* ------------------------------------------------------------------------- *
 * The four pooled-context MLPs of Muti_AFF (classification/model_utils.py:377-400: AdaptiveAvgPool2d(S) ->
 * Conv1x1(C->Ci)+bias -> BatchNorm -> ReLU -> Conv1x1(Ci->C)+bias -> [BatchNorm: in the gate kernel], S = 4, 8, 16, 1)
 * as ONE launch sequence for all four scales: they are chains of tiny launch-bound kernels (~50 launches per
 * Muti_AFF and train step through the per-layer entry points).  All tensors f32; pooled[k] / z2[k] / dz2[k] /
 * dpooled[k] are [P_k][C], z1[k] / g[k] are [P_k][Ci], P_k = N*S_k*S_k; weights in PyTorch's OIHW order
 * (w1 [Ci][C], w2 [C][Ci]).
 *   ledn_mfaf_ctx_fwd: z1 = W1 pooled + b1 (saved); BatchNorm on batch statistics (training != 0: running
 *     statistics updated, bn1[k] = [scale | shift | mean | invstd][Ci] saved) or on the running statistics;
 *     z2 = W2 relu(bn(z1)) + b2.  stats1: [4][2][Ci] zeroed scratch.
 *   ledn_mfaf_ctx_bwd: given dz2: dw2 / db2 / dgamma / dbeta / dw1 / db1 are ACCUMULATED (+=), dpooled written.
 *     sums: [4][2][Ci] zeroed scratch; g: scratch. */
typedef struct {
    const float* pooled[4];
    float* z1[4];
    float* z2[4];
    const float* w1[4];
    const float* b1[4];        /* may be NULL */
    const float* gamma[4];
    const float* beta[4];
    float* running_mean[4];    /* training: updated; inference: read */
    float* running_var[4];
    const float* w2[4];
    const float* b2[4];        /* may be NULL */
    float* bn1[4];             /* [4][Ci] */
    float* stats1;
    int P[4];
    int C, Ci;
    float momentum, eps;
    /* optional (training, all or none): the TRAILING BatchNorm of each scale on batch statistics -- sums of z2 in
     * stats2 ([4][2][C], zeroed), bn2[k] = [scale | shift | mean | invstd][C] written, running statistics updated */
    const float* gamma2[4];
    const float* beta2[4];
    float* running_mean2[4];
    float* running_var2[4];
    float* bn2[4];
    float* stats2;
    /* SyncBN (data-parallel): the launch sequence in PHASES with the caller's all-reduce of the statistics in between.
     * phase = bit mask of the launches to run (0 = all): 1 = conv1 + sums of z1 (-> stats1), 2 = BatchNorm + ReLU +
     * conv2 + sums of z2 (-> stats2), 4 = finalize of the trailing BatchNorms.  The caller all-reduces stats1 between
     * 1 and 2 and stats2 between 2 and 4 (in place: they are used for nothing else); count_scale = number of ranks
     * (the statistics are then over P_k * count_scale samples); 0 means 1. */
    int phase;
    float count_scale;
} ledn_mfafctx_desc;
int ledn_mfaf_ctx_fwd(const ledn_mfafctx_desc* d, int training, void* stream);
typedef struct {
    const float* pooled[4];
    const float* z1[4];
    const float* dz2[4];
    const float* w1[4];
    const float* w2[4];
    const float* bn1[4];
    float* g[4];
    float* dpooled[4];
    float* dw1[4];
    float* db1[4];             /* may be NULL */
    float* dgamma[4];
    float* dbeta[4];
    float* dw2[4];
    float* db2[4];             /* may be NULL */
    float* sums;
    int P[4];
    int C, Ci;
    /* optional (all or none): dz2[k] is then the gradient with respect to the trailing BatchNorm's OUTPUT; its
     * backward runs first (z2, bn2 from the forward; sums2 [4][2][C] zeroed; dz2s[k]: scratch that receives the
     * gradient with respect to z2; dgamma2 / dbeta2 accumulated) */
    const float* z2[4];
    const float* bn2[4];
    float* dz2s[4];
    float* dgamma2[4];
    float* dbeta2[4];
    float* sums2;
    /* SyncBN: phase = bit mask (0 = all): 1 = sums of the trailing BatchNorms' backward (-> sums2), 2 = backward through
     * conv2 / ReLU + sums of the first BatchNorm's backward (-> sums) + dW2, 4 = BatchNorm backward + dpooled + dW1.
     * The caller all-reduces sums2 between 1 and 2 and sums between 2 and 4 OUT OF PLACE: sums2 / sums then hold the
     * global sums (used for dz), sums2_local / sums_local this rank's own (= its d gamma / d beta, as
     * torch.nn.SyncBatchNorm keeps them; NULL: the same buffers).  count_scale as in the forward. */
    int phase;
    float count_scale;
    const float* sums_local;
    const float* sums2_local;
} ledn_mfafctx_bwd_desc;
int ledn_mfaf_ctx_bwd(const ledn_mfafctx_bwd_desc* d, void* stream);

/* ------------------------------------------------------------------------- *
 * SGD with momentum and weight decay over a table of tensors (torch.optim.SGD
 * semantics; config optimizer = dict(type='SGD', lr, momentum, weight_decay)):
 *   g' = grad_scale*g + wd*p;  m = momentum*m + g';  p -= lr*m;  g = 0.
 * (grad_scale = 1/world_size turns the all-reduced gradient SUM into DDP's mean.)
 * table: n_tensors x {p, g, m} device pointers followed by element counts
 * (see ledn_sgd_entry); one launch for the whole model. */
typedef struct {
    float* p;
    float* g;
    float* m;
    long long n;
} ledn_sgd_entry;
int ledn_sgd_step(const ledn_sgd_entry* table_dev, int n_tensors, long long max_n, float lr,
                  const float* lr_dev, float momentum, float weight_decay, float grad_scale,
                  void* stream);   /* lr_dev != NULL: learning rate read from device memory (hipGraph replay) */

#ifdef __cplusplus
}
#endif
#endif /* LEDN_H */
