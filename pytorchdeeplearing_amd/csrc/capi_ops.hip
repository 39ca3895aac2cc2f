// The stateless entry points of include/segengine.h: losses, metrics, optimiser, pre / post-processing and the operator-level seg_op_* calls the
// tests and tools use.  Nothing here touches an engine handle.
#include "engine_internal.h"

extern "C" {

long long seg_loss_ws_bytes(int n, int c) { return (long long)align_up(loss_sums_count(n, c) * sizeof(double) * STAT_REP); }

}  // extern "C"
namespace segi {
int fill_loss(LossArgs& a, const float* logits, const void* target, int label_type, int n, int c, long long v,
                     int loss_kind, float focal_alpha, float focal_gamma, void* ws) {
    if (!logits || !target || !ws) return fail("loss: null pointer");
    if (c < 1 || c > 16) return fail("loss: classes must be 1..16");
    if (loss_kind < 0 || loss_kind >= L_KIND_COUNT) return fail("loss: unknown loss kind");
    const bool binary_kind = loss_kind <= SEG_LOSS_BINARY_CE_DICE || (loss_kind >= L_BIN_JACCARD && loss_kind <= L_BIN_TVERSKY) || loss_kind == L_BIN_SS ||
                             loss_kind == L_BIN_MCC;
    if ((c == 1) != binary_kind) return fail("loss: binary losses need C == 1, multi-class losses C > 1");
    a.logits = logits; a.target = target; a.label_type = label_type; a.N = n; a.C = c; a.V = v; a.kind = loss_kind;
    a.focal_alpha = focal_alpha; a.focal_gamma = focal_gamma; a.class_alpha = nullptr; a.sums = (double*)ws;
    a.out = nullptr; a.dlogits = nullptr; a.grad_scale = 1.f; a.phase = 0; a.n_global = 0;
    return 0;
}
}  // namespace segi
extern "C" {

int seg_loss_forward(const float* logits, const void* target, int label_type, int n, int c, long long v, int loss_kind,
                     float focal_alpha, float focal_gamma, const float* class_alpha, void* ws, float* out3, void* stream) {
    LossArgs a;
    if (fill_loss(a, logits, target, label_type, n, c, v, loss_kind, focal_alpha, focal_gamma, ws)) return -1;
    if (!out3) return fail("seg_loss_forward: out3 is null");
    a.class_alpha = class_alpha; a.out = out3;
    launch_loss_forward(a, (hipStream_t)stream);
    return hipGetLastError() == hipSuccess ? 0 : fail("seg_loss_forward: launch failed");
}

int seg_loss_shared_doubles(void) { return loss_shared_count(); }

int seg_loss_reduce(const float* logits, const void* target, int label_type, int n, int c, long long v, int loss_kind,
                    float focal_alpha, float focal_gamma, void* ws, void* stream) {
    LossArgs a;
    if (fill_loss(a, logits, target, label_type, n, c, v, loss_kind, focal_alpha, focal_gamma, ws)) return -1;
    a.phase = 1;
    launch_loss_forward(a, (hipStream_t)stream);
    return hipGetLastError() == hipSuccess ? 0 : fail("seg_loss_reduce: launch failed");
}

int seg_loss_finalize(const float* logits, const void* target, int label_type, int n, int c, long long v, int loss_kind,
                      float focal_alpha, float focal_gamma, const float* class_alpha, int n_global, void* ws, float* out3, void* stream) {
    LossArgs a;
    if (fill_loss(a, logits, target, label_type, n, c, v, loss_kind, focal_alpha, focal_gamma, ws)) return -1;
    if (!out3) return fail("seg_loss_finalize: out3 is null");
    if (n_global != 0 && n_global < n) return fail("seg_loss_finalize: n_global must be >= the local sample count (or 0: the exchanged count)");
    a.class_alpha = class_alpha; a.out = out3; a.phase = 2; a.n_global = n_global;
    launch_loss_forward(a, (hipStream_t)stream);
    return hipGetLastError() == hipSuccess ? 0 : fail("seg_loss_finalize: launch failed");
}

int seg_loss_backward(const float* logits, const void* target, int label_type, int n, int c, long long v, int loss_kind,
                      float focal_alpha, float focal_gamma, void* ws, float grad_scale, float* dlogits, void* stream) {
    LossArgs a;
    if (fill_loss(a, logits, target, label_type, n, c, v, loss_kind, focal_alpha, focal_gamma, ws)) return -1;
    if (!dlogits) return fail("seg_loss_backward: dlogits is null");
    a.dlogits = dlogits; a.grad_scale = grad_scale;
    launch_loss_backward(a, (hipStream_t)stream);
    return hipGetLastError() == hipSuccess ? 0 : fail("seg_loss_backward: launch failed");
}

long long seg_lovasz_ws_bytes(int n, long long v) {
    if (n < 1 || v < 1) return fail("seg_lovasz_ws_bytes: empty batch");
    const long long b = lovasz_ws_bytes((long long)n * v);
    return b < 0 ? fail("seg_lovasz_ws_bytes: element count must be below 2^32 (and the sort library must be usable)") : b;
}
int seg_lovasz_forward(const float* x, const void* target, int label_type, int n, int c, long long v, void* ws, float* out1, float* dx,
                       void* stream) {
    if (!x || !target || !ws || !out1 || !dx) return fail("seg_lovasz_forward: null pointer");
    if (n < 1 || v < 1 || c < 1 || c > 16) return fail("seg_lovasz_forward: classes must be 1..16, batch and volume non-empty");
    if ((long long)n * v >= (1ll << 32)) return fail("seg_lovasz_forward: element count must be below 2^32");
    if (launch_lovasz(x, target, label_type, n, c, v, ws, out1, dx, (hipStream_t)stream)) return fail("seg_lovasz_forward: sort / scan failed");
    return hipGetLastError() == hipSuccess ? 0 : fail("seg_lovasz_forward: launch failed");
}

static int ssim_check(const char* what, const void* a, const void* b, const void* ws, int n, int c, int d, int h, int w, int nd, int window) {
    if (!a || !b || !ws) return fail(std::string(what) + ": null pointer");
    if (n < 1 || n > 64 || c < 1 || h < 1 || w < 1 || (nd != 2 && nd != 3) || (nd == 3 && d < 1)) return fail(std::string(what) + ": bad extents (batch 1..64)");
    if (window < 1 || window > 15 || !(window & 1)) return fail(std::string(what) + ": window_size must be odd and <= 15");
    return 0;
}
long long seg_ssim_ws_bytes(int n, int c, long long v) { return (n < 1 || c < 1 || v < 1) ? -1 : ssim_ws_bytes(n * c, v); }
int seg_ssim_forward(const float* img1, const float* img2, int n, int c, int d, int h, int w, int nd, int window, void* ws, float* out,
                     void* stream) {
    if (ssim_check("seg_ssim_forward", img1, img2, ws, n, c, d, h, w, nd, window) || !out) return out ? -1 : fail("seg_ssim_forward: out is null");
    if (launch_ssim_forward(img1, img2, n, c, nd == 3 ? d : 1, h, w, nd, window, ws, out, (hipStream_t)stream)) return fail("seg_ssim_forward: bad arguments");
    return hipGetLastError() == hipSuccess ? 0 : fail("seg_ssim_forward: launch failed");
}
int seg_ssim_forward_cols(const float* img1, const float* img2, int n, int c, int d, int h, int w, int nd, int window, void* ws, float* out,
                          float* out_cols, void* stream) {
    if (ssim_check("seg_ssim_forward_cols", img1, img2, ws, n, c, d, h, w, nd, window) || !out || !out_cols)
        return (out && out_cols) ? -1 : fail("seg_ssim_forward_cols: out is null");
    if (launch_ssim_forward(img1, img2, n, c, nd == 3 ? d : 1, h, w, nd, window, ws, out, (hipStream_t)stream, out_cols))
        return fail("seg_ssim_forward_cols: bad arguments");
    return hipGetLastError() == hipSuccess ? 0 : fail("seg_ssim_forward_cols: launch failed");
}
int seg_ssim_backward(const float* img1, const float* img2, int n, int c, int d, int h, int w, int nd, int window, void* ws, const float* gscale,
                      int per_sample, float* dimg1, float* dimg2, void* stream) {
    if (ssim_check("seg_ssim_backward", img1, img2, ws, n, c, d, h, w, nd, window)) return -1;
    if (!gscale || (!dimg1 && !dimg2)) return fail("seg_ssim_backward: null pointer");
    if (launch_ssim_backward(img1, img2, n, c, nd == 3 ? d : 1, h, w, nd, window, ws, gscale, per_sample, dimg1, dimg2, (hipStream_t)stream))
        return fail("seg_ssim_backward: bad arguments");
    return hipGetLastError() == hipSuccess ? 0 : fail("seg_ssim_backward: launch failed");
}

int seg_predict_mask(const float* probs, unsigned char* mask, int n, int c, long long v, float threshold, int scale, void* stream) {
    if (!probs || !mask) return fail("seg_predict_mask: null pointer");
    if (c < 1 || n < 1 || v < 1 || scale < 0 || scale > 255) return fail("seg_predict_mask: bad arguments");
    launch_mask(probs, mask, n, c, v, threshold, scale, (hipStream_t)stream);
    return hipGetLastError() == hipSuccess ? 0 : fail("seg_predict_mask: launch failed");
}
int seg_op_resample3d(const void* src, void* dst, int elem_type, int sd, int sh, int sw, int dd, int dh, int dw, double step_z, double step_y,
                      double step_x, int mode, void* stream) {
    if (!src || !dst) return fail("seg_op_resample3d: null pointer");
    if (sd < 1 || sh < 1 || sw < 1 || dd < 1 || dh < 1 || dw < 1) return fail("seg_op_resample3d: empty volume");
    if (elem_type != 0 && elem_type != 1) return fail("seg_op_resample3d: elem_type must be 0 (f32) or 1 (u8)");
    if (mode != RS_LINEAR && mode != RS_NEAREST) return fail("seg_op_resample3d: mode must be 0 (linear) or 1 (nearest)");
    if (mode == RS_LINEAR && elem_type != 0) return fail("seg_op_resample3d: linear interpolation needs f32 volumes");
    if (!(step_z > 0.0) || !(step_y > 0.0) || !(step_x > 0.0)) return fail("seg_op_resample3d: steps must be positive");
    ResampleArgs a;
    a.src = src; a.dst = dst; a.sD = sd; a.sH = sh; a.sW = sw; a.dD = dd; a.dH = dh; a.dW = dw;
    a.fz = step_z; a.fy = step_y; a.fx = step_x; a.mode = mode;
    launch_resample3d(a, elem_type, (hipStream_t)stream);
    return hipGetLastError() == hipSuccess ? 0 : fail("seg_op_resample3d: launch failed");
}
long long seg_op_normalize_ws_bytes(void) { return (long long)normalize_ws_bytes(); }
int seg_op_normalize_meanstd(const float* x, float* out, long long n, int clip, float lower, float upper, void* ws, void* stream) {
    if (!x || !out || !ws) return fail("seg_op_normalize_meanstd: null pointer");
    if (n < 1) return fail("seg_op_normalize_meanstd: empty volume");
    if (clip && !(lower <= upper)) return fail("seg_op_normalize_meanstd: lower > upper");
    launch_normalize_meanstd(x, out, n, clip, lower, upper, ws, (hipStream_t)stream);
    return hipGetLastError() == hipSuccess ? 0 : fail("seg_op_normalize_meanstd: launch failed");
}
int seg_op_normalize_percentile(const float* x, float* out, long long n, float q_lo, float q_hi, void* ws, void* stream) {
    if (!x || !out || !ws) return fail("seg_op_normalize_percentile: null pointer");
    if (n < 1) return fail("seg_op_normalize_percentile: empty volume");
    if (!(q_lo >= 0.f && q_lo <= q_hi && q_hi <= 100.f)) return fail("seg_op_normalize_percentile: need 0 <= q_lo <= q_hi <= 100");
    launch_normalize_percentile(x, out, n, q_lo, q_hi, ws, (hipStream_t)stream);
    return hipGetLastError() == hipSuccess ? 0 : fail("seg_op_normalize_percentile: launch failed");
}
static int check_windows(const char* what, int D, int H, int W, int nb, int pd, int ph, int pw) {
    if (nb < 1 || pd < 1 || ph < 1 || pw < 1) return fail(std::string(what) + ": empty patch list");
    if (pd > D || ph > H || pw > W) return fail(std::string(what) + ": patch larger than the volume");
    return 0;
}
int seg_op_gather_patches(const float* vol, int d, int h, int w, const int* origins, int nb, int pd, int ph, int pw, float* out, void* stream) {
    if (!vol || !origins || !out) return fail("seg_op_gather_patches: null pointer");
    if (check_windows("seg_op_gather_patches", d, h, w, nb, pd, ph, pw)) return -1;
    launch_gather_patches(vol, d, h, w, origins, nb, pd, ph, pw, out, (hipStream_t)stream);
    return hipGetLastError() == hipSuccess ? 0 : fail("seg_op_gather_patches: launch failed");
}
int seg_op_stitch_mask(const unsigned char* masks, const int* origins, int nb, int pd, int ph, int pw, unsigned char* out, int d, int h, int w,
                       void* stream) {
    if (!masks || !origins || !out) return fail("seg_op_stitch_mask: null pointer");
    if (check_windows("seg_op_stitch_mask", d, h, w, nb, pd, ph, pw)) return -1;
    launch_stitch_mask(masks, origins, nb, pd, ph, pw, out, d, h, w, (hipStream_t)stream);
    return hipGetLastError() == hipSuccess ? 0 : fail("seg_op_stitch_mask: launch failed");
}

int seg_metric(const float* probs, const void* target, int label_type, int n, int c, long long v, void* ws, float* out2, void* stream) {
    if (!probs || !target || !ws || !out2) return fail("seg_metric: null pointer");
    if (c < 1 || c > 16) return fail("seg_metric: classes must be 1..16");
    launch_metric(probs, target, label_type, n, c, v, (double*)ws, out2, (hipStream_t)stream);
    return hipGetLastError() == hipSuccess ? 0 : fail("seg_metric: launch failed");
}

}  // extern "C"
namespace segi {
// riders: the overflow flag was cleared and the step counter will be advanced by StepRiders of neighbouring launches (seg_train_step)
int adam_step_impl(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, long long numel, float lr, float beta1,
                          float beta2, float eps, float weight_decay, int decoupled, float inv_scale, int check_finite, int* state, void* stream,
                          bool riders) {
    if (!params || !grads || !exp_avg || !exp_avg_sq || !state) return fail("seg_adam_step: null pointer");
    hipStream_t st = (hipStream_t)stream;
    AdamArgs a;
    a.p = params; a.g = grads; a.m = exp_avg; a.v = exp_avg_sq; a.n = numel;
    a.lr = lr; a.beta1 = beta1; a.beta2 = beta2; a.eps = eps; a.weight_decay = weight_decay; a.decoupled = decoupled;
    a.inv_scale = inv_scale; a.step = state; a.found_inf = state + 1;
    if (!riders) (void)hipMemsetAsync(state + 1, 0, sizeof(int), st);
    if (check_finite) launch_grad_check(grads, numel, state + 1, st);
    launch_adam(a, st, !riders);
    return hipGetLastError() == hipSuccess ? 0 : fail("seg_adam_step: launch failed");
}
}  // namespace segi
extern "C" {
int seg_adam_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, long long numel, float lr, float beta1,
                  float beta2, float eps, float weight_decay, int decoupled, float inv_scale, int check_finite, int* state, void* stream) {
    return adam_step_impl(params, grads, exp_avg, exp_avg_sq, numel, lr, beta1, beta2, eps, weight_decay, decoupled, inv_scale, check_finite, state,
                          stream, false);
}

int seg_op_conv(const seg_conv_args* a, int dtype, void* stream) {
    if (!a || !a->in0 || !a->w || !a->out) return fail("seg_op_conv: null pointer");
    const int cin = a->C0 + a->C1;
    if (cin < 8 || (cin & (cin - 1)) || a->C0 % 8) return fail("seg_op_conv: channel counts must be powers of two >= 8");
    if (a->Cout % 16 || a->Ngemm % 16 || a->Kpad % 32 || a->Kpad < a->K) return fail("seg_op_conv: bad GEMM extents");
    if ((a->act_scale || a->act_shift) && !(a->act_scale && a->act_shift && conv_uses_stream_kernel(*a)))
        return fail("seg_op_conv: act_scale / act_shift need the streaming kernel (gather form, seg_op_conv_kernel == 1) and come as a pair");
    if (a->rq_Q && !(a->out1 && conv_uses_stream_kernel(*a))) return fail("seg_op_conv: rq_* need out1 / Cout0 on the streaming kernel and all of rq_r, rq_scale, rq_shift");
    if (a->out1 && !conv_uses_stream_kernel(*a)) return fail("seg_op_conv: out1 / Cout0 need the streaming kernel (gather form, no bias / stats, 16-channel tiles in pairs)");
    launch_conv_igemm(*a, dtype, (hipStream_t)stream);
    return hipGetLastError() == hipSuccess ? 0 : fail("seg_op_conv: launch failed");
}
int seg_op_conv_kernel(const seg_conv_args* a) { return a ? (conv_uses_stream_kernel(*a) ? 1 : 0) : -1; }
long long seg_op_wgrad_partial_bytes(const seg_wgrad_args* a) { return a ? (long long)wgrad_partial_bytes(*a) : -1; }
int seg_op_wgrad(const seg_wgrad_args* a, float* partial_scratch, int dtype, void* stream) {
    if (!a || !a->dr || !a->x0 || !a->dw || !partial_scratch) return fail("seg_op_wgrad: null pointer");
    if (a->P % 16) return fail("seg_op_wgrad: P must be a multiple of 16");
    if (a->stem ? (a->Q > 32) : (a->Q % 16 != 0)) return fail("seg_op_wgrad: bad Q");
    if ((a->act_scale || a->act_shift) && !(a->act_scale && a->act_shift && dtype != SEG_F32 && wgrad_act_supported(*a)))
        return fail("seg_op_wgrad: act_scale / act_shift need a 1^d stride-1 conv on 16-bit tensors with C0 <= 64 and come as a pair");
    launch_wgrad(*a, partial_scratch, dtype, (hipStream_t)stream);
    return hipGetLastError() == hipSuccess ? 0 : fail("seg_op_wgrad: launch failed");
}
int seg_op_pack(const seg_pack_desc* descs, int ndesc, long long max_elems, int dtype, void* stream) {
    if (!descs || ndesc < 1) return fail("seg_op_pack: no descriptors");
    launch_pack(descs, ndesc, (int)max_elems, dtype, (hipStream_t)stream);
    return hipGetLastError() == hipSuccess ? 0 : fail("seg_op_pack: launch failed");
}
int seg_op_conv3(const void* in, const void* w, const float* bias, void* out, double* stats, int n, int d, int h, int wid, int cin,
                 int cout, int ndim, int dtype, void* stream) {
    if (!in || !w || !out) return fail("seg_op_conv3: null pointer");
    if (cin < 16 || (cin & (cin - 1)) || cout % 16) return fail("seg_op_conv3: Cin must be a power of two >= 16, Cout a multiple of 16");
    launch_conv3(in, w, bias, out, stats, n, ndim == 3 ? d : 1, h, wid, cin, cout, ndim, dtype, (hipStream_t)stream);
    return hipGetLastError() == hipSuccess ? 0 : fail("seg_op_conv3: launch failed");
}
int seg_op_conv3x(int cfg, const void* in0, const void* in1, int c0, const void* w, const float* bias, void* out, double* stats, int n, int d,
                  int h, int wid, int cin, int cout, int ndim, int dtype, void* stream) {
    if (!in0 || !w || !out) return fail("seg_op_conv3x: null pointer");
    if (ndim != 2 && ndim != 3) return fail("seg_op_conv3x: ndim must be 2 or 3");
    const int dd = ndim == 3 ? d : 1;
    if (!conv3x_supported(dtype, ndim, n, dd, h, wid, cin, cout, c0, in1 != nullptr))
        return fail("seg_op_conv3x: needs a 16-bit dtype, Cin % 32 == 0 (or Cin == 16 without a concat), Cout % 16 == 0 and tensors below 2 GB per sample");
    if (cfg < 0) cfg = conv3x_pick(ndim, n, dd, h, wid, cin, cout);
    if (cfg < 0 || !launch_conv3x(cfg, in0, in1, c0, w, bias, out, stats, n, dd, h, wid, cin, cout, ndim, dtype, (hipStream_t)stream))
        return fail("seg_op_conv3x: the tiling does not fit this shape");
    return hipGetLastError() == hipSuccess ? 0 : fail("seg_op_conv3x: launch failed");
}
int seg_op_conv3x_num_cfgs(void) { return conv3x_num_cfgs(); }
int seg_op_conv3x_cfg_frag(int cfg) { return conv3x_cfg_frag(cfg); }
int seg_op_conv3x_cfg_info(int index, int* id, int* ndim, int* box3, int* bn, int* nres, char* name, int name_cap) {
    const char* nm = nullptr;
    if (conv3x_cfg_info(index, id, ndim, box3, bn, nres, &nm)) return fail("seg_op_conv3x_cfg_info: index out of range");
    if (name && name_cap > 0) snprintf(name, name_cap, "%s", nm);
    return 0;
}
int seg_op_conv3x_default_cfg(int ndim, int n, int d, int h, int wid, int cin, int cout, int dtype) {
    const int dd = ndim == 3 ? d : 1;
    if (!conv3x_supported(dtype, ndim, n, dd, h, wid, cin, cout, 0, false)) return -1;
    return conv3x_pick(ndim, n, dd, h, wid, cin, cout);
}
long long seg_op_wgrad3_partial_bytes(int ndim, int n, int d, int h, int wid, int p, int q) {
    return (long long)wgrad3_partial_bytes(ndim, n, ndim == 3 ? d : 1, h, wid, p, q);
}
int seg_op_wgrad3(const void* dr, const void* x, float* partial, float* dw, int n, int d, int h, int wid, int p, int q, int ndim,
                  int dtype, void* stream) {
    if (!dr || !x || !partial || !dw) return fail("seg_op_wgrad3: null pointer");
    if (p % 16 || q % 16 || (p > 16 && p % 32) || (q > 16 && q % 32)) return fail("seg_op_wgrad3: channel counts must be 16 or multiples of 32");
    launch_wgrad3(dr, x, partial, dw, n, ndim == 3 ? d : 1, h, wid, p, q, ndim, dtype, (hipStream_t)stream);
    return hipGetLastError() == hipSuccess ? 0 : fail("seg_op_wgrad3: launch failed");
}
int seg_op_wgrad3_cat(const void* dr, const void* x0, const void* x1, int c0, float* partial, float* dw, int n, int d, int h, int wid, int p,
                      int q, int ndim, int dtype, void* stream) {
    if (!dr || !x0 || !x1 || !partial || !dw) return fail("seg_op_wgrad3_cat: null pointer");
    if (p % 16 || q % 16 || (p > 16 && p % 32) || (q > 16 && q % 32)) return fail("seg_op_wgrad3_cat: channel counts must be 16 or multiples of 32");
    if (c0 <= 0 || c0 >= q || c0 % 16) return fail("seg_op_wgrad3_cat: c0 must be a multiple of 16 inside (0, q)");
    launch_wgrad3(dr, x0, partial, dw, n, ndim == 3 ? d : 1, h, wid, p, q, ndim, dtype, (hipStream_t)stream, x1, c0);
    return hipGetLastError() == hipSuccess ? 0 : fail("seg_op_wgrad3_cat: launch failed");
}
long long seg_op_stemx_partial_bytes(int ndim, int n, int d, int h, int wid, int cimg) {
    return (long long)stemx_partial_bytes(ndim, n, ndim == 3 ? d : 1, h, wid, cimg);
}
int seg_op_stemx(const seg_stemx_args* a, int mode, int ndim, int dtype, float* dw3, float* dw1, void* stream) {
    if (!a || !a->img || !a->w3) return fail("seg_op_stemx: null pointer");
    if (mode < 0 || mode > 3) return fail("seg_op_stemx: mode must be 0..3");
    if (ndim != 2 && ndim != 3) return fail("seg_op_stemx: ndim must be 2 or 3");
    if (a->Cimg < 1 || a->Cimg > 3 || (ndim == 3 && a->Cimg != 1)) return fail("seg_op_stemx: image channels must be 1 (3-D) or 1..3 (2-D)");
    if ((long long)(ndim == 3 ? a->D : 1) * a->H * a->W * 16 * 4 >= (1ll << 31)) return fail("seg_op_stemx: volume too large for one buffer range");
    if (mode == 0 && (!a->stats3 || (a->w1 && !a->stats1))) return fail("seg_op_stemx: statistics pointers");
    if (mode >= 1 && (!a->scale3 || !a->shift3 || (a->w1 && (!a->scale1 || !a->shift1)))) return fail("seg_op_stemx: scale / shift pointers");
    if (mode == 1 && !a->out) return fail("seg_op_stemx: out is null");
    if (mode >= 2 && (a->ndy < 1 || a->ndy > 3 || !a->dy[0])) return fail("seg_op_stemx: gradient sources");
    if (mode == 2 && (!a->Q3 || (a->w1 && !a->Q1))) return fail("seg_op_stemx: Q pointers");
    if (mode == 3 && (!a->coef3 || (a->w1 && !a->coef1) || !a->partial || !dw3 || (a->w1 && !dw1))) return fail("seg_op_stemx: weight-gradient pointers");
    launch_stemx(*a, mode, ndim, dtype, dw3, dw1, (hipStream_t)stream);
    return hipGetLastError() == hipSuccess ? 0 : fail("seg_op_stemx: launch failed");
}
int seg_abi_sizeof(int which) {
    return which == 0 ? (int)sizeof(seg_conv_args) : which == 1 ? (int)sizeof(seg_wgrad_args) : which == 2 ? (int)sizeof(seg_pack_desc)
           : which == 3 ? (int)sizeof(seg_stemx_args) : (int)sizeof(seg_train_args);
}

#define SEG_OK(what) (hipGetLastError() == hipSuccess ? 0 : fail(what ": launch failed"))
int seg_op_pool3(const float* x, float* out, int planes, int d, int h, int w, int nd, int is_min, void* stream) {
    if (!x || !out || (nd != 2 && nd != 3)) return fail("seg_op_pool3: bad arguments");
    launch_pool3(x, out, planes, d, h, w, nd, is_min, (hipStream_t)stream);
    return SEG_OK("seg_op_pool3");
}
int seg_op_skel_iter(const float* x, float* e_out, float* x_out, int planes, int d, int h, int w, int nd, void* stream) {
    if (!x || !e_out || !x_out || (nd != 2 && nd != 3)) return fail("seg_op_skel_iter: bad arguments");
    launch_skel_iter(x, e_out, x_out, planes, d, h, w, nd, (hipStream_t)stream);
    return SEG_OK("seg_op_skel_iter");
}
int seg_op_skel_iter_bwd(const float* g, const float* x, const float* e, float* dx, float* de_scratch, int planes, int d, int h, int w, int nd,
                         void* stream) {
    if (!g || !x || !e || !dx || !de_scratch || (nd != 2 && nd != 3)) return fail("seg_op_skel_iter_bwd: bad arguments");
    launch_skel_iter_bwd(g, x, e, dx, de_scratch, planes, d, h, w, nd, (hipStream_t)stream);
    return SEG_OK("seg_op_skel_iter_bwd");
}
int seg_op_skel_update(const float* x, const float* e, float* out, int planes, int d, int h, int w, int nd, void* stream) {
    if (!x || !e || !out || (nd != 2 && nd != 3)) return fail("seg_op_skel_update: bad arguments");
    launch_skel_update(x, e, out, planes, d, h, w, nd, (hipStream_t)stream);
    return SEG_OK("seg_op_skel_update");
}
int seg_op_skel_update_bwd(const float* g, const float* x, const float* e, float* dx, float* de, int planes, int d, int h, int w, int nd,
                           void* stream) {
    if (!g || !x || !e || !dx || !de || (nd != 2 && nd != 3)) return fail("seg_op_skel_update_bwd: bad arguments");
    launch_skel_update_bwd(g, x, e, dx, de, planes, d, h, w, nd, (hipStream_t)stream);
    return SEG_OK("seg_op_skel_update_bwd");
}
int seg_op_pool3_bwd(const float* src, const float* dout, float* din, int planes, int d, int h, int w, int nd, int is_min, void* stream) {
    if (!src || !dout || !din || (nd != 2 && nd != 3)) return fail("seg_op_pool3_bwd: bad arguments");
    launch_pool3_bwd(src, dout, din, planes, d, h, w, nd, is_min, (hipStream_t)stream);
    return SEG_OK("seg_op_pool3_bwd");
}
long long seg_op_plane_dot_scratch_bytes(int planes, long long v) { return (long long)plane_dot_scratch_bytes(planes, v); }
int seg_op_plane_dot(const float* a, const float* b, double* out2, double* scratch, int planes, long long v, void* stream) {
    if (!a || !b || !out2 || !scratch) return fail("seg_op_plane_dot: null pointer");
    launch_plane_dot(a, b, out2, scratch, planes, v, (hipStream_t)stream);
    return SEG_OK("seg_op_plane_dot");
}
int seg_op_plane_axpb(const float* in, const float* a, const float* b, float* out, int planes, long long v, int accumulate, void* stream) {
    if (!in || !a || !b || !out) return fail("seg_op_plane_axpb: null pointer");
    launch_plane_axpb(in, a, b, out, planes, v, accumulate, (hipStream_t)stream);
    return SEG_OK("seg_op_plane_axpb");
}

long long seg_cldice_ws_bytes(int n, int d, int h, int w, int nd, int width) {
    if (n < 1 || h < 1 || w < 1 || width < 0 || (nd != 2 && nd != 3)) return -1;
    return (long long)cldice_binary_ws_bytes(n, (long long)(nd == 3 ? d : 1) * h * w, width);
}
int seg_cldice_target(const void* target, int label_type, int n, int d, int h, int w, int nd, int width, void* ws, void* stream) {
    if (!target || !ws) return fail("seg_cldice_target: null pointer");
    if (n < 1 || h < 1 || w < 1 || width < 0 || (nd != 2 && nd != 3)) return fail("seg_cldice_target: bad extents");
    launch_cldice_target(target, label_type, n, nd == 3 ? d : 1, h, w, nd, width, ws, (hipStream_t)stream);
    return SEG_OK("seg_cldice_target");
}
int seg_cldice_binary(const float* probs, const void* target, int label_type, int n, int d, int h, int w, int nd, int width,
                      float grad_scale, void* ws, float* out1, float* dlogits, int target_ready, void* stream) {
    if (!probs || !target || !ws || !out1) return fail("seg_cldice_binary: null pointer");
    if (n < 1 || h < 1 || w < 1 || width < 0 || (nd != 2 && nd != 3)) return fail("seg_cldice_binary: bad extents");
    launch_cldice_binary(probs, target, label_type, n, nd == 3 ? d : 1, h, w, nd, width, grad_scale, ws, out1, dlogits, target_ready,
                         (hipStream_t)stream);
    return SEG_OK("seg_cldice_binary");
}

}  // extern "C"
