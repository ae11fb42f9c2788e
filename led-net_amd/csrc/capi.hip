// capi.hip -- the extern "C" surface of libledn_hip.so (declared in include/ledn.h).
// Thin: validate on the host, dispatch to the kernel launchers, report status.
#include "ledn_rt.h"

namespace ledn {
int conv_validate(const ledn_conv_desc& d);
int conv_direct(const ledn_conv_desc& d, hipStream_t s);
bool conv_mfma_supported(const ledn_conv_desc& d);
bool conv_f32_mfma_supported(const ledn_conv_desc& d);
int conv_f32_mfma(const ledn_conv_desc& d, hipStream_t s);
bool conv_wgrad_f32_mfma_supported(const ledn_wgrad_desc& d);
int conv_wgrad_f32_mfma(const ledn_wgrad_desc& d, hipStream_t s);
int conv_mfma(const ledn_conv_desc& d, hipStream_t s);
int conv_wgrad_mfma_partial(const ledn_wgrad_desc& d, float* part, long long part_floats, ledn_wgrad_finish_entry* entry,
                            bool query, hipStream_t s);
int conv_wgrad_finish_multi_impl(const ledn_wgrad_finish_entry* table_dev, int n, int total_chunks, hipStream_t s);
bool conv1x1_reg_supported(const ledn_conv_desc& d);
int conv1x1_reg(const ledn_conv_desc& d, hipStream_t s);
bool conv3x3_reg_supported(const ledn_conv_desc& d);
int conv3x3_reg(const ledn_conv_desc& d, hipStream_t s);
bool conv3x3_narrowin_mfma_supported(const ledn_conv_desc& d);
int conv3x3_narrowin_mfma(const ledn_conv_desc& d, hipStream_t s);
bool wgrad_mfma_supported(const ledn_wgrad_desc& d);
bool conv_wgrad_cout2_supported(const ledn_wgrad_desc& d);
bool conv_wgrad_narrow_reg_supported(const ledn_wgrad_desc& d);
bool conv1x1_wgrad_reg_applies(const ledn_wgrad_desc& d);
int conv_wgrad_narrow_reg(const ledn_wgrad_desc& d, hipStream_t s);
int conv_wgrad_cout2(const ledn_wgrad_desc& d, hipStream_t s);
int conv_wgrad_mfma(const ledn_wgrad_desc& d, hipStream_t s);
int pack_conv_weights_multi_impl(const ledn_pack_entry* table_dev, int n, long long max_elems, hipStream_t s);
int im2col_stem_impl(const void* x, void* p, int N, int H, int W, int C, int Ho, int Wo, hipStream_t s);
int pack_conv_weights_impl(const float* w, void* out, int Cout, int Cin, int KH, int KW, int mode,
                           int groups, hipStream_t s);
int wgrad_validate(const ledn_wgrad_desc& d);
int conv_wgrad_direct(const ledn_wgrad_desc& d, hipStream_t s);
int dwconv_impl(const ledn_dw_desc& d, hipStream_t s);
int iou_hist_impl(const unsigned char* pred, const long long* label, long long P, int num_classes,
                  int ignore_index, float* hist, hipStream_t s);
int bn_finalize_rows_impl(const float* part, int rows, double count, const float* gamma, const float* beta,
                          float* running_mean, float* running_var, float momentum, float eps, float* scale,
                          float* shift, float* mean, float* invstd, float* sum, float* sqsum, int C,
                          hipStream_t s);
int im2col_stem_planar_impl(const void* x, int dtype_x, void* p, int N, int H, int W, int C, int Ho, int Wo,
                            const float* scale, const float* shift, const int* map, const int* valid_hw, float pad_val,
                            hipStream_t s);
int stem_conv_wgrad_impl(const void* x, int dtype_x, const void* dz, float* dw, int N, int H, int W, int C, int Ho, int Wo,
                         int Cout, const float* in_scale, const float* in_shift, const int* map, const int* valid_hw,
                         float pad_val, const ledn_bnbwd_desc* bn, hipStream_t s);
int stem_conv_impl(const void* x, int dtype_x, const void* wp, void* y, int N, int H, int W, int C, int Ho, int Wo,
                   int Cout, const float* in_scale, const float* in_shift, const int* map, const int* valid_hw,
                   float pad_val, const float* out_scale, const float* out_shift, int act_out, float* stat_sum,
                   float* stat_sqsum, hipStream_t s);
int mfaf_ctx_fwd_impl(const ledn_mfafctx_desc& d, int training, hipStream_t s);
int mfaf_ctx_bwd_impl(const ledn_mfafctx_bwd_desc& d, hipStream_t s);
int dw_pack_impl(const ledn_dwpack_desc& d, float* packed, hipStream_t s);
int dw_unpack_grad_impl(const ledn_dwpack_desc& d, const float* dpacked, hipStream_t s);
int dw_repack_multi_impl(const ledn_dwpack_entry* table_dev, int n, int max_elems, int dir, hipStream_t s);
int augment_batch_impl(const ledn_aug_entry* table_dev, int n, unsigned char* out_img, long long* out_seg, int OH,
                       int OW, int pad_val, int seg_pad_val, hipStream_t s);
int aug_crop_hist_impl(const ledn_aug_entry* table_dev, int n, int max_pixels, int* hist, hipStream_t s);
int sesp_pyramid_impl(const ledn_pyr_desc& d, hipStream_t s);
int channel_stats_impl(const void* x, const void* xadd, long long P, int C, int dtype, float* sum,
                       float* sqsum, hipStream_t s);
int bn_finalize_impl(const float* sum, const float* sqsum, double count, const float* gamma,
                     const float* beta, float* running_mean, float* running_var, float momentum,
                     float eps, float* scale, float* shift, float* mean, float* invstd, int C,
                     hipStream_t s);
int affine_act_impl(const ledn_affine_desc& d, hipStream_t s);
int bilinear_impl(const ledn_resize_desc& d, hipStream_t s);
int nchw_to_nhwc_impl(const void* x, int dtype_x, void* y, int dtype_y, int N, int C, int H, int W,
                      const float* scale, const float* shift, const int* map, const int* valid_hw, float pad_val,
                      hipStream_t s);
int avgpool2d_impl(const void* x, void* y, int N, int H, int W, int C, int Ho, int Wo, int k, int st, int pad,
                   int dtype, hipStream_t s);
int avgpool2d_bwd_impl(const void* dy, void* dx, int N, int H, int W, int C, int Ho, int Wo, int k, int st, int pad,
                       int dtype, hipStream_t s);
int relpos_bias_impl(const float* table, const long long* index, float* out, int R, int heads, int T, hipStream_t s);
int relpos_bias_bwd_impl(const float* dbias, const long long* index, float* dtable, int R, int heads, int T,
                         hipStream_t s);
int adaptive_avgpool_impl(const void* x, const void* xadd, float* y, int N, int H, int W, int C, int S,
                          int dtype, hipStream_t s);
int avgpool3x3s2_impl(const void* x, void* y, int N, int H, int W, int C, int Ho, int Wo, int dtype,
                      hipStream_t s);
int window_attn_impl(const void* qkv, const float* biasT, void* out, int N, int H, int W, int C,
                     int heads, int ws, int dtype, hipStream_t s);
int getb_pool_impl(const void* a, const void* local, void* out, int N, int H, int W, int C, int ws,
                   int dtype, hipStream_t s);
int mfaf_gate_impl(const ledn_mfaf_desc& d, hipStream_t s);
int seam_edge_impl(const float* seg, float* edge, float* scratch, int N, int h, int w, int kth, float thr,
                   float final_thr, hipStream_t s);
int bn_act_bwd_reduce_impl(const ledn_bnbwd_desc& d, hipStream_t s);
int bn_act_bwd_apply_impl(const ledn_bnbwd_desc& d, hipStream_t s);
int dw_bwd_data_impl(const ledn_dwbwd_desc& d, hipStream_t s);
int dw_bwd_weight_impl(const ledn_dwbwd_desc& d, hipStream_t s);
int pyr_bwd_data_impl(const ledn_pyrbwd_desc& d, hipStream_t s);
int pyr_bwd_weight_impl(const ledn_pyrbwd_desc& d, hipStream_t s);
int bilinear_bwd_impl(const void* dy, void* dx, int N, int H, int W, int C, int Ho, int Wo, int dtype_dy,
                      int dtype_dx, hipStream_t s);
int avgpool3x3s2_bwd_impl(const void* dy, const void* add, void* dx, int N, int H, int W, int C, int Ho,
                          int Wo, int dtype, hipStream_t s);
int window_attn_bwd_impl(const void* qkv, const float* biasT, const void* dout, float* dqkv, float* dbiasT,
                         int N, int H, int W, int C, int heads, int ws, int dtype, hipStream_t s);
int getb_pool_bwd_impl(const void* dout, void* da, int N, int H, int W, int C, int ws, int dtype,
                       hipStream_t s);
int mfaf_gate_bwd_impl(const ledn_mfafbwd_desc& d, hipStream_t s);
int mfaf_bwd_combine_impl(void* dx, void* dr, const void* dxl, const float* const* dpool, const int* sizes,
                          int npool, int N, int H, int W, int C, int dtype, hipStream_t s);
long long ohem_work_floats(long long P);
int ohem_ce_fwd_impl(const float* logits, const long long* target, long long P, int C, float thres,
                     long long min_kept, float loss_weight, int ignore_label, float* work, float* out,
                     hipStream_t s);
int ohem_ce_up_fwd_impl(const float* src, int N, int Hs, int Ws, int H, int W, const long long* target, float thres,
                        long long min_kept, float loss_weight, int ignore_label, float* work, float* out,
                        hipStream_t s);
int ohem2_up_fwd_impl(const float* src0, const float* src1, int N, int Hs, int Ws, int H, int W,
                      const long long* target, float thres0, long long min_kept0, float lw0, float thres1,
                      long long min_kept1, float lw1, int ignore_label, float* work, float* out, hipStream_t s);
int ohem2_up_bwd_impl(const float* src0, const float* src1, int N, int Hs, int Ws, int H, int W, int ignore_label,
                      const float* work, const float* out, const float* dloss0, const float* dloss1, float lw0,
                      float lw1, float* dsrc0, float* dsrc1, hipStream_t s);
long long ohem2_work_floats(long long P);
int ohem_ce_up_bwd_impl(const float* src, int N, int Hs, int Ws, int H, int W, const long long* target,
                        int ignore_label, const float* work, const float* out, const float* dloss, float loss_weight,
                        float* dsrc, hipStream_t s);
int ohem_ce_bwd_impl(const float* logits, const long long* target, long long P, int C, int ignore_label,
                     const float* work, const float* out, const float* dloss, float loss_weight,
                     float* dlogits, hipStream_t s);
int sgd_step_impl(const ledn_sgd_entry* table_dev, int n_tensors, long long max_n, float lr,
                  const float* lr_dev, float momentum, float weight_decay, float grad_scale, hipStream_t s);
}  // namespace ledn

#include <mutex>
#include <unordered_map>

namespace ledn {
// Scratch state is keyed by the HIP stream an entry point is called with (every entry takes one): a workspace bound
// to a stream (ledn_bind_workspace) serves the launches on that stream only, so two models / host threads / streams
// in one process do not depend on each other's call order.  ledn_set_workspace keeps the process default (used by
// streams without a binding).  The deferred-statistics hand-off lives in the calling host thread.
static Workspace g_ws = {nullptr, 0};
static std::mutex g_ws_mutex;
static std::unordered_map<void*, Workspace> g_ws_by_stream;
static thread_local void* tls_stream = nullptr;
static thread_local Workspace tls_ws = {nullptr, 0};
static hipStream_t enter_stream(void* stream) {         // every extern "C" entry passes its stream through here
    tls_stream = stream;
    std::lock_guard<std::mutex> lock(g_ws_mutex);
    auto it = g_ws_by_stream.find(stream);
    tls_ws = it != g_ws_by_stream.end() ? it->second : g_ws;
    return (hipStream_t)stream;
}
Workspace& workspace() { return tls_ws; }
static Options g_opt = {512, 512, 91, 0, 0};   // stream_fast: bit 0 BatchNorm / affine streaming kernels, bit 1 LDS-tiled depthwise 3x3, bit 2 round-robin conv tiles (off), bit 3 8-row conv tiles for under-filled grids, bit 4 register-direct 1x1 conv (conv1x1.hip), bit 5 resident 3x3 weights for 32 < Cin <= 64 (off: measured slower, r3l), bit 6 register-direct 3x3 conv for 32 input channels (conv3x3.hip)
Options& options() { return g_opt; }
static thread_local DeferredStats g_defer = {false, nullptr, 0};
DeferredStats& deferred_stats() { return g_defer; }
}  // namespace ledn

using namespace ledn;
#define S(stream) (ledn::enter_stream(stream))

extern "C" {

int ledn_abi_version(void) { return LEDN_ABI_VERSION; }

int ledn_set_workspace(void* ptr, long long nfloats) {
    if (nfloats < 0 || (ptr == nullptr) != (nfloats == 0)) return LEDN_EINVAL;
    // the process default (streams without a binding pick it up in enter_stream) and this thread's current view
    std::lock_guard<std::mutex> lock(g_ws_mutex);
    g_ws = Workspace{(float*)ptr, (long)nfloats};
    if (g_ws_by_stream.find(tls_stream) == g_ws_by_stream.end()) tls_ws = g_ws;
    return LEDN_OK;
}

int ledn_bind_workspace(void* stream, void* ptr, long long nfloats) {
    if (nfloats < 0 || (ptr == nullptr) != (nfloats == 0)) return LEDN_EINVAL;
    std::lock_guard<std::mutex> lock(g_ws_mutex);
    if (ptr) g_ws_by_stream[stream] = Workspace{(float*)ptr, (long)nfloats};
    else g_ws_by_stream.erase(stream);
    return LEDN_OK;
}

int ledn_set_option(int option, long long value) {
    if (value > (1 << 20)) return LEDN_EINVAL;
    switch (option) {
        case LEDN_OPT_CONV_WORKGROUPS: options().conv_workgroups = value > 0 ? (int)value : 512; return LEDN_OK;
        case LEDN_OPT_WGRAD_WORKGROUPS: options().wgrad_workgroups = value > 0 ? (int)value : 512; return LEDN_OK;
        case LEDN_OPT_STREAM_FAST: options().stream_fast = value < 0 ? 91 : (int)value; return LEDN_OK;   // (< 0: the default mask)
        case LEDN_OPT_DETERMINISTIC: options().deterministic = value > 0 ? 1 : 0; return LEDN_OK;
        case LEDN_OPT_BN_FUSED: options().bn_fused = value > 0 ? 1 : 0; return LEDN_OK;
        default: return LEDN_EINVAL;
    }
}

int ledn_conv2d(const ledn_conv_desc* d, void* stream) {
    if (!d) return LEDN_EINVAL;
    const int rc = conv_validate(*d);
    if (rc != LEDN_OK) return rc;
    if (head_fwd_supported(*d)) return head_fwd(*d, S(stream));      // the two-class heads (head_bwd.hip)
    if (conv_mfma_supported(*d)) {
        if (conv1x1_reg_supported(*d)) return conv1x1_reg(*d, S(stream));
        if (conv3x3_reg_supported(*d)) return conv3x3_reg(*d, S(stream));
        return conv_mfma(*d, S(stream));
    }
    if (conv3x3_narrowin_mfma_supported(*d)) return conv3x3_narrowin_mfma(*d, S(stream));
    if (conv_f32_mfma_supported(*d)) return conv_f32_mfma(*d, S(stream));       // f32 activations: v_mfma_f32_32x32x2_f32
    return conv_direct(*d, S(stream));
}

int ledn_stats_defer_begin(void) {
    DeferredStats& ds = deferred_stats();
    ds.want = true;
    ds.part = nullptr;
    ds.rows = 0;
    return LEDN_OK;
}
int ledn_stats_defer_end(float** part, int* rows) {
    if (!part || !rows) return LEDN_EINVAL;
    DeferredStats& ds = deferred_stats();
    *part = ds.part;
    *rows = ds.part ? ds.rows : 0;
    ds.want = false;
    ds.part = nullptr;
    ds.rows = 0;
    return LEDN_OK;
}

int ledn_conv2d_deferred_stats(const ledn_conv_desc* d, float** part, int* rows, void* stream) {
    if (!d || !part || !rows) return LEDN_EINVAL;
    *part = nullptr;
    *rows = 0;
    DeferredStats& ds = deferred_stats();
    ds.want = true;
    ds.part = nullptr;
    ds.rows = 0;
    const int rc = ledn_conv2d(d, stream);
    ds.want = false;
    if (rc == LEDN_OK && ds.part) {
        *part = ds.part;
        *rows = ds.rows;
    }
    return rc;
}

int ledn_bn_finalize_rows(const float* part, int rows, double count, const float* gamma, const float* beta,
                          float* running_mean, float* running_var, float momentum, float eps, float* scale,
                          float* shift, float* mean, float* invstd, float* sum, float* sqsum, int C,
                          void* stream) {
    return bn_finalize_rows_impl(part, rows, count, gamma, beta, running_mean, running_var, momentum, eps, scale,
                                 shift, mean, invstd, sum, sqsum, C, S(stream));
}

int ledn_conv2d_uses_mfma(const ledn_conv_desc* d) {
    if (d && conv_validate(*d) == LEDN_OK && head_fwd_supported(*d)) return 6;
    if (d && !conv_mfma_supported(*d) && conv3x3_narrowin_mfma_supported(*d)) return 4;
    if (d && !conv_mfma_supported(*d) && conv_validate(*d) == LEDN_OK && conv_f32_mfma_supported(*d)) return 5;
    if (!d || !conv_mfma_supported(*d)) return 0;
    return conv1x1_reg_supported(*d) ? 2 : (conv3x3_reg_supported(*d) ? 3 : 1);
}
int ledn_conv2d_wgrad_uses_mfma(const ledn_wgrad_desc* d) {
    if (d && !conv_wgrad_cout2_supported(*d) && conv_wgrad_narrow_reg_supported(*d)) return 2;
    if (d && !conv_wgrad_cout2_supported(*d) && conv1x1_wgrad_reg_applies(*d)) return 3;
    if (d && !conv_wgrad_cout2_supported(*d) && wgrad_mfma_supported(*d)) return 1;
    return d && !conv_wgrad_cout2_supported(*d) && conv_wgrad_f32_mfma_supported(*d) ? 4 : 0;
}

int ledn_pack_conv_weights_multi(const ledn_pack_entry* table_dev, int n, long long max_elems, void* stream) {
    return pack_conv_weights_multi_impl(table_dev, n, max_elems, S(stream));
}

int ledn_im2col_stem_planar(const void* x, int dtype_x, void* p, int N, int H, int W, int C, int Ho, int Wo,
                            const float* scale, const float* shift, const int* map, const int* valid_hw,
                            float pad_val, void* stream) {
    return im2col_stem_planar_impl(x, dtype_x, p, N, H, W, C, Ho, Wo, scale, shift, map, valid_hw, pad_val, S(stream));
}
int ledn_im2col_stem(const void* x, void* p, int N, int H, int W, int C, int Ho, int Wo, void* stream) {
    return im2col_stem_impl(x, p, N, H, W, C, Ho, Wo, S(stream));
}

int ledn_pack_conv_weights(const float* w, void* out_bf16, int Cout, int Cin, int KH, int KW, int mode,
                           int groups, void* stream) {
    return pack_conv_weights_impl(w, out_bf16, Cout, Cin, KH, KW, mode, groups, S(stream));
}

int ledn_conv2d_wgrad(const ledn_wgrad_desc* d, void* stream) {
    if (!d) return LEDN_EINVAL;
    const int rc = wgrad_validate(*d);
    if (rc != LEDN_OK) return rc;
    if (conv_wgrad_cout2_supported(*d)) return conv_wgrad_cout2(*d, S(stream));   // two-class heads
    if (conv_wgrad_narrow_reg_supported(*d)) {                                     // ... their 3x3 layers
        const int r2 = conv_wgrad_narrow_reg(*d, S(stream));
        if (r2 != LEDN_OK || !d->db) return r2;
        return channel_stats_impl(d->dz, nullptr, (long long)d->N * d->Ho * d->Wo, d->Cout, d->dtype_dz, d->db,
                                  nullptr, S(stream));
    }
    if (wgrad_mfma_supported(*d)) {
        const int r2 = conv_wgrad_mfma(*d, S(stream));
        if (r2 != LEDN_OK || !d->db) return r2;
        return channel_stats_impl(d->dz, nullptr, (long long)d->N * d->Ho * d->Wo, d->Cout, d->dtype_dz, d->db,
                                  nullptr, S(stream));
    }
    if (conv_wgrad_f32_mfma_supported(*d)) {          // f32 activations: v_mfma_f32_32x32x2_f32 (needs the bound workspace)
        const int r2 = conv_wgrad_f32_mfma(*d, S(stream));
        if (r2 >= 0) return r2;
    }
    return conv_wgrad_direct(*d, S(stream));
}

long long ledn_conv2d_wgrad_partial_floats(const ledn_wgrad_desc* d) {
    if (!d || wgrad_validate(*d) != LEDN_OK || conv_wgrad_cout2_supported(*d) || conv_wgrad_narrow_reg_supported(*d) ||
        !wgrad_mfma_supported(*d))
        return 0;
    ledn_wgrad_finish_entry e;
    if (conv_wgrad_mfma_partial(*d, nullptr, 0, &e, true, nullptr) != LEDN_OK || e.nbx <= 4) return 0;
    return (long long)e.nbx * e.pairs * e.KK * 1024;
}
int ledn_conv2d_wgrad_partial(const ledn_wgrad_desc* d, float* part, long long part_floats,
                              ledn_wgrad_finish_entry* entry, void* stream) {
    if (!d || !part || !entry) return LEDN_EINVAL;
    if (ledn_conv2d_wgrad_partial_floats(d) <= 0) return LEDN_EINVAL;
    const int rc = conv_wgrad_mfma_partial(*d, part, part_floats, entry, false, S(stream));
    if (rc != LEDN_OK || !d->db) return rc;
    return channel_stats_impl(d->dz, nullptr, (long long)d->N * d->Ho * d->Wo, d->Cout, d->dtype_dz, d->db, nullptr,
                              S(stream));
}
int ledn_conv2d_wgrad_finish_multi(const ledn_wgrad_finish_entry* table_dev, int n, int total_chunks, void* stream) {
    return conv_wgrad_finish_multi_impl(table_dev, n, total_chunks, S(stream));
}

int ledn_iou_hist(const unsigned char* pred, const long long* label, long long P, int num_classes,
                  int ignore_index, float* hist, void* stream) {
    return iou_hist_impl(pred, label, P, num_classes, ignore_index, hist, S(stream));
}
int ledn_dwconv2d(const ledn_dw_desc* d, void* stream) { return d ? dwconv_impl(*d, S(stream)) : LEDN_EINVAL; }
int ledn_dw_pack(const ledn_dwpack_desc* d, float* packed, void* stream) {
    return d ? dw_pack_impl(*d, packed, S(stream)) : LEDN_EINVAL;
}
int ledn_dw_unpack_grad(const ledn_dwpack_desc* d, const float* dpacked, void* stream) {
    return d ? dw_unpack_grad_impl(*d, dpacked, S(stream)) : LEDN_EINVAL;
}
int ledn_sesp_pyramid(const ledn_pyr_desc* d, void* stream) {
    return d ? sesp_pyramid_impl(*d, S(stream)) : LEDN_EINVAL;
}

int ledn_channel_stats(const void* x, const void* xadd, long long P, int C, int dtype, float* sum,
                       float* sqsum, void* stream) {
    return channel_stats_impl(x, xadd, P, C, dtype, sum, sqsum, S(stream));
}
int ledn_bn_finalize(const float* sum, const float* sqsum, double count, const float* gamma,
                     const float* beta, float* running_mean, float* running_var, float momentum,
                     float eps, float* scale, float* shift, float* mean, float* invstd, int C,
                     void* stream) {
    return bn_finalize_impl(sum, sqsum, count, gamma, beta, running_mean, running_var, momentum, eps, scale,
                            shift, mean, invstd, C, S(stream));
}
int ledn_affine_act(const ledn_affine_desc* d, void* stream) {
    return d ? affine_act_impl(*d, S(stream)) : LEDN_EINVAL;
}
int ledn_nchw_to_nhwc(const void* x, int dtype_x, void* y, int dtype_y, int N, int C, int H, int W,
                      const float* scale, const float* shift, const int* map, const int* valid_hw, float pad_val,
                      void* stream) {
    return nchw_to_nhwc_impl(x, dtype_x, y, dtype_y, N, C, H, W, scale, shift, map, valid_hw, pad_val, S(stream));
}
int ledn_avgpool2d(const void* x, void* y, int N, int H, int W, int C, int Ho, int Wo, int k, int stride, int pad,
                   int dtype, void* stream) {
    return avgpool2d_impl(x, y, N, H, W, C, Ho, Wo, k, stride, pad, dtype, S(stream));
}
int ledn_avgpool2d_bwd(const void* dy, void* dx, int N, int H, int W, int C, int Ho, int Wo, int k, int stride,
                       int pad, int dtype, void* stream) {
    return avgpool2d_bwd_impl(dy, dx, N, H, W, C, Ho, Wo, k, stride, pad, dtype, S(stream));
}
int ledn_relpos_bias(const float* table, const long long* index, float* biasT, int R, int heads, int T, void* stream) {
    return relpos_bias_impl(table, index, biasT, R, heads, T, S(stream));
}
int ledn_relpos_bias_bwd(const float* dbiasT, const long long* index, float* dtable, int R, int heads, int T,
                         void* stream) {
    return relpos_bias_bwd_impl(dbiasT, index, dtable, R, heads, T, S(stream));
}
int ledn_stem_conv(const void* x, int dtype_x, const void* wp, void* y, int N, int H, int W, int C, int Ho, int Wo,
                   int Cout, const float* in_scale, const float* in_shift, const int* map, const int* valid_hw,
                   float pad_val, const float* out_scale, const float* out_shift, int act_out, float* stat_sum,
                   float* stat_sqsum, void* stream) {
    return stem_conv_impl(x, dtype_x, wp, y, N, H, W, C, Ho, Wo, Cout, in_scale, in_shift, map, valid_hw, pad_val,
                          out_scale, out_shift, act_out, stat_sum, stat_sqsum, S(stream));
}
int ledn_stem_conv_wgrad(const void* x, int dtype_x, const void* dz, float* dw, int N, int H, int W, int C, int Ho, int Wo,
                         int Cout, const float* in_scale, const float* in_shift, const int* map, const int* valid_hw,
                         float pad_val, void* stream) {
    return stem_conv_wgrad_impl(x, dtype_x, dz, dw, N, H, W, C, Ho, Wo, Cout, in_scale, in_shift, map, valid_hw, pad_val,
                                nullptr, S(stream));
}
int ledn_stem_conv_wgrad_bn(const void* x, int dtype_x, const ledn_bnbwd_desc* bn, float* dw, int N, int H, int W, int C, int Ho,
                            int Wo, int Cout, const float* in_scale, const float* in_shift, const int* map,
                            const int* valid_hw, float pad_val, void* stream) {
    LEDN_REQUIRE(bn);
    return stem_conv_wgrad_impl(x, dtype_x, nullptr, dw, N, H, W, C, Ho, Wo, Cout, in_scale, in_shift, map, valid_hw, pad_val,
                                bn, S(stream));
}
int ledn_mfaf_ctx_fwd(const ledn_mfafctx_desc* d, int training, void* stream) {
    return d ? mfaf_ctx_fwd_impl(*d, training, S(stream)) : LEDN_EINVAL;
}
int ledn_mfaf_ctx_bwd(const ledn_mfafctx_bwd_desc* d, void* stream) {
    return d ? mfaf_ctx_bwd_impl(*d, S(stream)) : LEDN_EINVAL;
}
int ledn_dw_repack_multi(const ledn_dwpack_entry* table_dev, int n, int max_elems, int dir, void* stream) {
    return dw_repack_multi_impl(table_dev, n, max_elems, dir, S(stream));
}
int ledn_augment_batch(const ledn_aug_entry* table_dev, int n, unsigned char* out_img, long long* out_seg, int OH,
                       int OW, int pad_val, int seg_pad_val, void* stream) {
    return augment_batch_impl(table_dev, n, out_img, out_seg, OH, OW, pad_val, seg_pad_val, S(stream));
}
int ledn_aug_crop_hist(const ledn_aug_entry* table_dev, int n, int max_pixels, int* hist, void* stream) {
    return aug_crop_hist_impl(table_dev, n, max_pixels, hist, S(stream));
}
int ledn_bilinear(const ledn_resize_desc* d, void* stream) { return d ? bilinear_impl(*d, S(stream)) : LEDN_EINVAL; }
int ledn_adaptive_avgpool(const void* x, const void* xadd, float* y, int N, int H, int W, int C, int Sz,
                          int dtype, void* stream) {
    return adaptive_avgpool_impl(x, xadd, y, N, H, W, C, Sz, dtype, S(stream));
}
int ledn_avgpool3x3s2(const void* x, void* y, int N, int H, int W, int C, int Ho, int Wo, int dtype,
                      void* stream) {
    return avgpool3x3s2_impl(x, y, N, H, W, C, Ho, Wo, dtype, S(stream));
}
int ledn_window_attn(const void* qkv, const float* biasT, void* out, int N, int H, int W, int C, int heads,
                     int ws, int dtype, void* stream) {
    return window_attn_impl(qkv, biasT, out, N, H, W, C, heads, ws, dtype, S(stream));
}
int ledn_getb_pool(const void* a, const void* local, void* out, int N, int H, int W, int C, int ws,
                   int dtype, void* stream) {
    return getb_pool_impl(a, local, out, N, H, W, C, ws, dtype, S(stream));
}
int ledn_mfaf_gate(const ledn_mfaf_desc* d, void* stream) { return d ? mfaf_gate_impl(*d, S(stream)) : LEDN_EINVAL; }
int ledn_seam_edge(const float* seg, float* edge, float* scratch, int N, int h, int w, int kth, float thr,
                   float final_thr, void* stream) {
    return seam_edge_impl(seg, edge, scratch, N, h, w, kth, thr, final_thr, S(stream));
}

int ledn_bn_act_bwd_reduce(const ledn_bnbwd_desc* d, void* stream) {
    return d ? bn_act_bwd_reduce_impl(*d, S(stream)) : LEDN_EINVAL;
}
int ledn_bn_act_bwd_apply(const ledn_bnbwd_desc* d, void* stream) {
    return d ? bn_act_bwd_apply_impl(*d, S(stream)) : LEDN_EINVAL;
}
int ledn_bn_act_bwd_fused(const ledn_bnbwd_desc* d, void* stream) {
    if (!d) return LEDN_EINVAL;
    if (!options().bn_fused) return LEDN_ESKIP;
    return bn_act_bwd_fused(*d, S(stream));
}
int ledn_bn_act_bwd_fused_check(int C, void* stream) { return bn_act_bwd_fused_check(C, S(stream)); }
int ledn_head_bwd_supported(const ledn_headbwd_desc* d) { return d ? head_bwd_supported(*d) : 0; }
int ledn_head_bwd_reduce(const ledn_headbwd_desc* d, void* stream) {
    LEDN_REQUIRE(d);
    return head_bwd_reduce(*d, S(stream));
}
int ledn_head_bwd_apply(const ledn_headbwd_desc* d, void* stream) {
    LEDN_REQUIRE(d);
    return head_bwd_apply(*d, S(stream));
}
int ledn_dwconv2d_bwd_data(const ledn_dwbwd_desc* d, void* stream) {
    return d ? dw_bwd_data_impl(*d, S(stream)) : LEDN_EINVAL;
}
int ledn_dwconv2d_bwd_weight(const ledn_dwbwd_desc* d, void* stream) {
    return d ? dw_bwd_weight_impl(*d, S(stream)) : LEDN_EINVAL;
}
int ledn_sesp_pyramid_bwd_data(const ledn_pyrbwd_desc* d, void* stream) {
    return d ? pyr_bwd_data_impl(*d, S(stream)) : LEDN_EINVAL;
}
int ledn_sesp_pyramid_bwd_weight(const ledn_pyrbwd_desc* d, void* stream) {
    return d ? pyr_bwd_weight_impl(*d, S(stream)) : LEDN_EINVAL;
}
int ledn_bilinear_bwd(const void* dy, void* dx, int N, int H, int W, int C, int Ho, int Wo, int dtype_dy,
                      int dtype_dx, void* stream) {
    return bilinear_bwd_impl(dy, dx, N, H, W, C, Ho, Wo, dtype_dy, dtype_dx, S(stream));
}
int ledn_avgpool3x3s2_bwd(const void* dy, const void* add, void* dx, int N, int H, int W, int C, int Ho,
                          int Wo, int dtype, void* stream) {
    return avgpool3x3s2_bwd_impl(dy, add, dx, N, H, W, C, Ho, Wo, dtype, S(stream));
}
int ledn_window_attn_bwd(const void* qkv, const float* biasT, const void* dout, float* dqkv, float* dbiasT,
                         int N, int H, int W, int C, int heads, int ws, int dtype, void* stream) {
    return window_attn_bwd_impl(qkv, biasT, dout, dqkv, dbiasT, N, H, W, C, heads, ws, dtype, S(stream));
}
int ledn_getb_pool_bwd(const void* dout, void* da, int N, int H, int W, int C, int ws, int dtype,
                       void* stream) {
    return getb_pool_bwd_impl(dout, da, N, H, W, C, ws, dtype, S(stream));
}
int ledn_mfaf_gate_bwd(const ledn_mfafbwd_desc* d, void* stream) {
    return d ? mfaf_gate_bwd_impl(*d, S(stream)) : LEDN_EINVAL;
}
int ledn_mfaf_bwd_combine(void* dx, void* dr, const void* dxl, const float* const* dpool, const int* sizes,
                          int npool, int N, int H, int W, int C, int dtype, void* stream) {
    return mfaf_bwd_combine_impl(dx, dr, dxl, dpool, sizes, npool, N, H, W, C, dtype, S(stream));
}
long long ledn_ohem_work_floats(long long P) { return ohem_work_floats(P); }
int ledn_ohem_ce_fwd(const float* logits, const long long* target, long long P, int C, float thres,
                     long long min_kept, float loss_weight, int ignore_label, float* work, float* out,
                     void* stream) {
    return ohem_ce_fwd_impl(logits, target, P, C, thres, min_kept, loss_weight, ignore_label, work, out,
                            S(stream));
}
int ledn_ohem_ce_up_fwd(const float* src, int N, int Hs, int Ws, int H, int W, const long long* target, float thres,
                        long long min_kept, float loss_weight, int ignore_label, float* work, float* out, void* stream) {
    return ohem_ce_up_fwd_impl(src, N, Hs, Ws, H, W, target, thres, min_kept, loss_weight, ignore_label, work, out, S(stream));
}
int ledn_ohem2_up_fwd(const float* src0, const float* src1, int N, int Hs, int Ws, int H, int W,
                      const long long* target, float thres0, long long min_kept0, float loss_weight0, float thres1,
                      long long min_kept1, float loss_weight1, int ignore_label, float* work, float* out, void* stream) {
    return ohem2_up_fwd_impl(src0, src1, N, Hs, Ws, H, W, target, thres0, min_kept0, loss_weight0, thres1, min_kept1,
                             loss_weight1, ignore_label, work, out, S(stream));
}
int ledn_ohem2_up_bwd(const float* src0, const float* src1, int N, int Hs, int Ws, int H, int W, int ignore_label,
                      const float* work, const float* out, const float* dloss0, const float* dloss1,
                      float loss_weight0, float loss_weight1, float* dsrc0, float* dsrc1, void* stream) {
    return ohem2_up_bwd_impl(src0, src1, N, Hs, Ws, H, W, ignore_label, work, out, dloss0, dloss1, loss_weight0,
                             loss_weight1, dsrc0, dsrc1, S(stream));
}
long long ledn_ohem2_work_floats(long long P) { return ohem2_work_floats(P); }
int ledn_ohem_ce_up_bwd(const float* src, int N, int Hs, int Ws, int H, int W, const long long* target,
                        int ignore_label, const float* work, const float* out, const float* dloss, float loss_weight,
                        float* dsrc, void* stream) {
    return ohem_ce_up_bwd_impl(src, N, Hs, Ws, H, W, target, ignore_label, work, out, dloss, loss_weight, dsrc, S(stream));
}
int ledn_ohem_ce_bwd(const float* logits, const long long* target, long long P, int C, int ignore_label,
                     const float* work, const float* out, const float* dloss, float loss_weight,
                     float* dlogits, void* stream) {
    return ohem_ce_bwd_impl(logits, target, P, C, ignore_label, work, out, dloss, loss_weight, dlogits,
                            S(stream));
}
int ledn_sgd_step(const ledn_sgd_entry* table_dev, int n_tensors, long long max_n, float lr,
                  const float* lr_dev, float momentum, float weight_decay, float grad_scale, void* stream) {
    return sgd_step_impl(table_dev, n_tensors, max_n, lr, lr_dev, momentum, weight_decay, grad_scale,
                         S(stream));
}

}  // extern "C"
