// Launch interfaces of the gfx950 kernels (host side).  All tensors are channels-last
// [N][D][H][W][C] in the run dtype T (f32 / f16 / bf16); statistics and gradients of parameters
// are fp32/fp64.  Every launch is asynchronous on the given stream.
#pragma once
#include "../../include/segengine.h"
#include "common.h"

namespace seg {

typedef seg_taps Taps;
// Implicit-GEMM convolution arguments: see seg_conv_args in include/segengine.h
typedef seg_conv_args ConvArgs;
// Bookkeeping stores of the train step that ride on a kernel the step launches anyway instead of being one-wave launches of their own (each
// costs ~5 us of the main queue at a place where nothing overlaps it: profiles/r04_trace_timeline.txt, step start).  Executed by the first
// thread of the carrying kernel: an in-order queue starts it after everything launched before it has completed.
//   bump: *bump += 1 unless *gate is set (then *tally += 1)   = adam_bump_kernel     clear: *clear = 0   = the fill of the overflow flag
struct StepRider { int* bump = nullptr; const int* gate = nullptr; int* tally = nullptr; int* clear = nullptr; };
__device__ __forceinline__ void step_rider_run(const StepRider& r) {
    if (r.bump) { if (!(r.gate && *r.gate)) *r.bump += 1; else if (r.tally) *r.tally += 1; }
    if (r.clear) *r.clear = 0;
}
void launch_conv_igemm(const ConvArgs& a, int dtype, hipStream_t s, int stat_rep = STAT_REP);   // stat_rep: LDS-staged kernel only
bool conv_uses_stream_kernel(const ConvArgs& a);   // true: register-resident streaming kernel, false: LDS-staged implicit GEMM

// LDS halo-tile kernels for 3^d stride-1 pad-1 convs (conv3.hip): forward / data-gradient and weight gradient
void launch_conv3(const void* in, const void* w, const float* bias, void* out, double* stats, int N, int D, int H, int W, int Cin,
                  int Cout, int ndim, int dtype, hipStream_t s, const void* in1 = nullptr, int C0 = 0);   // in1: second concat source
// register-blocked variant for 16-bit tensors, Cin % 32 == 0 (conv3x.hip); weights fragment-major (PackDesc.frag = 1)
bool conv3x_supported(int dtype, int ndim, int N, int D, int H, int W, int Cin, int Cout, int C0, bool has_in1);
int conv3x_pick(int ndim, int N, int D, int H, int W, int Cin, int Cout, bool has_in1 = false);    // tiling id for the shape, -1: none
int conv3x_cfg_frag(int cfg);      // seg_pack_desc.frag of the weights the tiling reads (0: unknown tiling)
int conv3x_num_cfgs();
int conv3x_cfg_info(int index, int* id, int* ndim, int* box3, int* bn, int* nres, const char** name);
bool launch_conv3x(int cfg, const void* in0, const void* in1, int C0, const void* w, const float* bias, void* out, double* stats, int N, int D,
                   int H, int W, int Cin, int Cout, int ndim, int dtype, hipStream_t s, int stat_rep = STAT_REP);
// replicas a statistics producer spreads its atomics over, by voxels per sample: enough to keep same-address fp64 atomics apart,
// few enough that the consumer-side fold (gn_fold_block) reads ~8 KB
inline int stat_rep_for(long long V) { return V >= 262144 ? 32 : V >= 65536 ? 16 : V >= 8192 ? 8 : 4; }
// fused input block (stemx.hip): modes 0 forward statistics, 1 forward apply, 2 backward reduce, 3 backward apply + weight gradients
int stemx_workgroups(int ndim, int N, int D, int H, int W);
size_t stemx_partial_bytes(int ndim, int N, int D, int H, int W, int Cimg);
void launch_stemx(const seg_stemx_args& a, int mode, int ndim, int dtype, float* dw3, float* dw1, hipStream_t s);
int wgrad3_blocks_per_combo(int ndim, int N, int D, int H, int W, int P, int Q, int esz = 2);
size_t wgrad3_partial_bytes(int ndim, int N, int D, int H, int W, int P, int Q);
// second stage of the halo weight gradient: dw += the ordered sum of a layer's partial tiles.  launch_wgrad3 runs it itself, or - `defer` given - hands
// back its descriptor for a later launch_wgrad3_reduce over several layers (one launch; the partial buffers must stay untouched until then)
struct Wgrad3Reduce { const float* partial; float* dw; int P, Q, CP, CQ, ntap, nb; long long sP, sQ; int qreal; };
constexpr int W3_BATCH = 4;
struct Wgrad3ReduceBatch { Wgrad3Reduce r[W3_BATCH]; };
void launch_wgrad3_reduce(const Wgrad3Reduce* list, int n, hipStream_t s);
void launch_wgrad3(const void* dr, const void* x, float* partial, float* dw, int N, int D, int H, int W, int P, int Q, int ndim,
                   int dtype, hipStream_t s, const void* x1 = nullptr, int C0 = 0, int qreal = 0, Wgrad3Reduce* defer = nullptr);      // qreal in (0, Q): x carries zero-padded channels, dw is [P][qreal][taps] (the channels beyond qreal are not written)

// MFMA image stem (K = taps*Cimg <= 32): forward and weight gradient on box tiles (conv3.hip)
void launch_stem_fwd(const void* in, const void* w, const float* bias, void* out, double* stats, int N, int D, int H, int W, int Cimg,
                     int Cout, int center, int ndim, int dtype, hipStream_t s);
size_t stem_wgrad_partial_bytes(int ndim, int N, int D, int H, int W, int Cout);
void launch_stem_wgrad(const void* dr, const void* in, float* partial, float* dw, int N, int D, int H, int W, int Cimg, int Cout,
                       int center, int ndim, int dtype, hipStream_t s);

// Direct convolution for a tiny input-channel count (stem: image_channel -> features)
struct StemArgs {
    const void* in;     // [N][V][Cimg] T
    const float* w;     // fp32 master weights, PyTorch layout [Cout][Cimg][taps]
    const float* bias;  // or null
    void* out;          // [N][V][Cout] T
    double* stats;
    int N, D, H, W, Cimg, Cout;
    Taps taps;
};
void launch_conv_stem(const StemArgs& a, int dtype, hipStream_t s);

// 1^d head: logits[n][c][v] (planar fp32) = act[n][v][:] . w[c][:] + b[c]; probs = sigmoid/softmax
struct HeadArgs {
    const void* in;     // [M][Cin] T
    const float* w;     // fp32 [C][Cin]
    const float* bias;  // [C]
    float* logits;      // [N][C][V]
    float* probs;       // [N][C][V]
    int N, V, Cin, C;
    double* zero_ptr = nullptr; long long zero_n = 0;      // (train step) the loss workspace the reduction that follows accumulates into: cleared by workgroup 0
};
void launch_head_fwd(const HeadArgs& a, int dtype, hipStream_t s);

struct HeadBwdArgs {
    const void* in;        // activated input [M][Cin] T
    const float* w;        // [C][Cin]
    const float* dlogits;  // [N][C][V] fp32 (already multiplied by the loss scale)
    void* din;             // [M][Cin] T, or null: the data-gradient stays virtual (GnBwdArgs::vdl)
    float* dw;             // [C][Cin] +=
    float* db;             // [C] +=
    int N, V, Cin, C;
};
void launch_head_bwd(const HeadBwdArgs& a, int dtype, hipStream_t s);

// GroupNorm(8) finalize: stats -> per-(n,c) scale/shift (dropout multiplier folded in) + mean/rstd
struct GnFinArgs {
    const double* stats;  // [N][C][2]
    const float* gamma;
    const float* beta;
    const float* mask;    // [N][mask_ld] dropout multipliers or null (eval)
    int mask_ld;
    float* scale;         // [N][C]
    float* shift;         // [N][C]
    float* mean;          // [N][8]
    float* rstd;          // [N][8]
    int N, C;
    long long V;
    float eps;
    int rep;              // replicas of `stats` that hold data (1..STAT_REP, power of two); 0 = STAT_REP
};
void launch_gn_finalize(const GnFinArgs& a, hipStream_t s, const GnFinArgs* b = nullptr);      // b: a second module in the same launch

// y = relu(scale1*r1+shift1) [+ relu(scale2*r2+shift2)] [+ res]
struct ActArgs {
    const void* r1; const float* scale1; const float* shift1;
    const void* r2; const float* scale2; const float* shift2;
    const void* res;
    void* out;
    int N, C;
    long long V;
    // fold != 0: scale/shift are not read; each workgroup finalizes the statistics of its sample itself (fin1, fin2 for r2) and
    // workgroup 0 of every sample publishes scale / shift / mean / rstd for the backward pass
    int fold;
    GnFinArgs fin1, fin2;
    // optional (head_w != null; C == 16, head_C in {1, 2, 4}, no second branch): the 1^d head that reads this activation (networks/VNet3d.py:83-99) runs in the same pass -
    // logits / probabilities are written next to `out` from the values just rounded to T, in head_fwd_kernel's order of operations (bit-identical), and the
    // launch that would re-read the tensor (113 MB at 4 x 96^3) is gone.  Fields as in HeadArgs.
    const float* head_w = nullptr; const float* head_b = nullptr; float* logits = nullptr; float* probs = nullptr; int head_C = 0;
    double* zero_ptr = nullptr; long long zero_n = 0;
};
void launch_gn_act(const ActArgs& a, int dtype, hipStream_t s);
inline bool gn_act_head_supported(int C, int head_C, bool two_branches) { return C == 16 && !two_branches && (head_C == 1 || head_C == 2 || head_C == 4); }

// GroupNorm+dropout+ReLU backward, pass 1: Q[n][c] = {sum dzr, sum dzr*r}, dzr = (sum_i dy_i)*[scale*r+shift>0]
struct GnBwdArgs {
    const void* dy[3]; int ndy;
    const void* r;
    const float* scale; const float* shift;    // forward scale/shift [N][C]
    double* Q;                                  // [N][C][2]
    const float* coef;                          // pass 2: [N][C][3] (A,B,Cc)
    void* dr;                                   // pass 2 output T
    int N, C;
    long long V;
    // optional SECOND branch normalised by the same GroupNorm module and fed by the same gradient sources (the two convolutions
    // of the VNet input block, networks/VNet3d.py:25-43): one pass reads the gradient sources once for both branches
    const void* r2; const float* scale2; const float* shift2; double* Q2; const float* coef2; void* dr2;
    // optional VIRTUAL gradient source: the data-gradient of the 1^d head, dy[v][c] += sum_k vdl[n][k][v] * vw[k][c], evaluated on
    // the fly from the loss gradient (planar fp32) and the head weights instead of being written as a 16-channel tensor and read
    // back by every GroupNorm-backward pass it feeds
    const float* vdl; const float* vw; int vK;    int rep_q;                                  // replicas of Q the reduce pass spreads over; 0 = STAT_REP
};
void launch_gn_bwd_reduce(const GnBwdArgs& a, int dtype, hipStream_t s);
struct GnBwdFinArgs;
// fa != null: the backward finalize of the branch(es) runs as a prologue of this launch (no gn_bwd_finalize launch before it)
void launch_gn_bwd_apply(const GnBwdArgs& a, int dtype, hipStream_t s, const GnBwdFinArgs* fa = nullptr, const GnBwdFinArgs* fb = nullptr);

struct GnBwdFinArgs {
    const double* Q;       // [N][C][2]
    const double* stats;   // forward [N][C][2]
    const float* gamma;
    const float* mask; int mask_ld;
    const float* mean; const float* rstd;   // [N][8]
    float* dgamma; float* dbeta;            // += (fp32 master-grad layout)
    float* dbias;                           // conv bias grad += or null
    float* coef;                            // [N][C][3]
    int N, C;
    long long V;
    int rep_q, rep_s;                       // replicas of Q / stats that hold data; 0 = STAT_REP
};
void launch_gn_bwd_finalize(const GnBwdFinArgs& a, hipStream_t s, const GnBwdFinArgs* b = nullptr);
// small L2-resident tensors (C >= 64): reduce + finalize + apply in one launch, one workgroup per (sample, group)
bool gn_bwd_group_eligible(int C, long long V, int esz);
void launch_gn_fwd_group(const GnFinArgs& f, const void* r, const void* res, void* out, int dtype, hipStream_t s);
void launch_gn_bwd_group(const GnBwdArgs& e, const GnBwdFinArgs& f, int dtype, hipStream_t s);
// the same in one launch on S workgroups per (sample, group) that exchange their partial sums inside the kernel (norm.hip: gn_bwd_coop_kernel); uses the
// unit's Q region (zero at launch) as the exchange area
bool gn_bwd_coop_plan(int C, long long V, int N, int esz, int* S, int* ku);
bool gn_bwd_coop_eligible(const GnBwdArgs& e, int esz);
void launch_gn_bwd_coop(const GnBwdArgs& e, const GnBwdFinArgs& f, int dtype, hipStream_t s);

// Weight gradient: see seg_wgrad_args in include/segengine.h
typedef seg_wgrad_args WgradArgs;
size_t wgrad_partial_bytes(const WgradArgs& a);   // scratch for the per-slice partial tiles
void launch_wgrad(const WgradArgs& a, float* partial, int dtype, hipStream_t s, int qreal = 0);   // qreal in (0, Q): zero-padded input channels, see launch_wgrad3
bool wgrad_act_supported(const WgradArgs& a);    // a.act_scale / act_shift (x0 activated on load) can be honoured for these extents (16-bit tensors)

struct PoolArgs {
    const void* in; void* out;       // fwd: in fine, out coarse
    const void* dout; void* din;     // bwd
    int N, D, H, W, C;               // fine dims
    int pd, ph, pw;                  // window/stride (1 or 2)
};
void launch_maxpool_fwd(const PoolArgs& a, int dtype, hipStream_t s);
void launch_maxpool_bwd(const PoolArgs& a, int dtype, hipStream_t s);

// fp32 NC[D]HW image -> channels-last T
void launch_ingest(const float* x, void* out, int N, int C, long long V, int dtype, hipStream_t s, int Csrc = 0, StepRider rd = StepRider{});      // Csrc < C: zero-padded channels

// Generic weight re-layout: see seg_pack_desc in include/segengine.h
typedef seg_pack_desc PackDesc;
void launch_pack(const PackDesc* descs_dev, int ndesc, int max_rows, int dtype, hipStream_t s, StepRider rd = StepRider{});

// Losses on planar fp32 logits [N][C][V]
enum LossKind { L_BIN_DICE = 0, L_BIN_CE = 1, L_BIN_FOCAL = 2, L_BIN_CE_DICE = 3, L_MC_CE = 4, L_MC_FOCAL = 5, L_MC_DICE = 6,
                // not selectable from the reference's wrappers: same reduction sums, different ratio
                L_BIN_JACCARD = 7, L_BIN_ELDICE = 8, L_BIN_TVERSKY = 9, L_MC_CE_DICE = 10, L_MC_ELDICE = 11, L_BIN_SS = 12,
                L_MC_TVERSKY = 13, L_MC_SS = 14,
                L_BIN_MCC = 15,            // the ONE kind whose input is a probability map, not logits (model/losses.py:200-232)
                L_KIND_COUNT = 16 };
struct LossArgs {
    const float* logits;
    const void* target; int label_type;
    int N, C; long long V;
    int kind;
    float focal_alpha, focal_gamma;
    const float* class_alpha;   // [C] (MutilDiceLoss) or null -> ones
    double* sums;               // workspace, zeroed by the launcher
    float* out;                 // [0]=loss [1]=dice metric [2]=iou metric
    float* dlogits;             // [N][C][V] or null
    float grad_scale;           // loss scale folded into dlogits
    int phase;                  // 0: reduce + finalize; 1: reduce + fold replicas only; 2: finalize only (sums exchanged by the caller)
    int n_global;               // samples over ALL ranks for the mean losses (0: N)
    int prezeroed = 0;          // (train step) `sums` was cleared by the head kernel in front of this launch: no fill
};
int loss_shared_count();        // leading doubles of `sums` that are batch-global (summed across ranks in phase 1 -> 2)
__host__ __device__ size_t loss_sums_count(int N, int C);   // doubles per replica (STAT_REP replicas)
void launch_loss_forward(const LossArgs& a, hipStream_t s);    // reduce + finalize (writes out[], coefficient block in sums)
void launch_loss_backward(const LossArgs& a, hipStream_t s);   // dlogits from the finalized coefficients

// Lovasz hinge (C == 1) / Lovasz "softmax" on raw scores (C > 1), lovasz.hip: out1[0] = loss, dx = d loss / d x [N][C][V]; returns <0 on a
// library (sort / scan) failure.  ws: lovasz_ws_bytes(N*V) bytes (<0: unsupported element count)
long long lovasz_ws_bytes(long long P);
int launch_lovasz(const float* x, const void* target, int label_type, int N, int C, long long V, void* ws, float* out1, float* dx, hipStream_t s);

// SSIM / SSIM3D (ssim.hip; model/lossesSSIM.py): planar fp32 [N][C][D][H][W]; out = {mean, per-sample means[N]}; ws keeps the derivative maps
long long ssim_ws_bytes(int planes, long long v);
int launch_ssim_forward(const float* x1, const float* x2, int N, int C, int D, int H, int W, int nd, int window, void* ws, float* out, hipStream_t s, float* out_cols = nullptr);
int launch_ssim_backward(const float* x1, const float* x2, int N, int C, int D, int H, int W, int nd, int window, void* ws, const float* gscale,
                         int per_sample_scale, float* dx1, float* dx2, hipStream_t s);

// dice / iou on probabilities (model/metric.py): out2 = {dice, iou}; sums = 3*N*C doubles (zeroed by the launcher)
void launch_metric(const float* probs, const void* target, int label_type, int N, int C, long long V, double* sums, float* out2, hipStream_t s);
// out[c] += sum_m x[m][c]   (bias gradient of a conv without GroupNorm)
void launch_colsum(const void* x, float* out, long long M, int C, int dtype, hipStream_t s);

// soft-clDice building blocks (cldice.hip): planar fp32 [planes][D][H][W]; nd = 3 pools over (D,H,W), nd = 2 over (H,W)
void launch_pool3(const float* x, float* out, int planes, int D, int H, int W, int nd, int is_min, hipStream_t s);
void launch_skel_iter(const float* x, float* e_out, float* x_out, int planes, int D, int H, int W, int nd, hipStream_t s);
void launch_skel_iter_bwd(const float* g, const float* x, const float* e, float* dx, float* de, int planes, int D, int H, int W, int nd,
                          hipStream_t s);
void launch_skel_update(const float* x, const float* e, float* out, int planes, int D, int H, int W, int nd, hipStream_t s);
void launch_skel_update_bwd(const float* g, const float* x, const float* e, float* dx, float* de, int planes, int D, int H, int W, int nd, hipStream_t s);
void launch_pool3_bwd(const float* src, const float* dout, float* din, int planes, int D, int H, int W, int nd, int is_min, hipStream_t s);
size_t plane_dot_scratch_bytes(int planes, long long V);
void launch_plane_dot(const float* a, const float* b, double* out, double* scratch, int planes, long long V, hipStream_t s);
void launch_plane_axpb(const float* in, const float* a, const float* b, float* out, int planes, long long V, int accumulate, hipStream_t s);
// binary soft-clDice, forward + backward to the logits in one call (cldice.hip)
size_t cldice_binary_ws_bytes(int planes, long long V, int width);
void launch_cldice_target(const void* target, int label_type, int planes, int D, int H, int W, int nd, int width, void* ws, hipStream_t s);
void launch_cldice_binary(const float* probs, const void* target, int label_type, int planes, int D, int H, int W, int nd, int width,
                          float gscale, void* ws, float* out1, float* dlogits, int target_ready, hipStream_t s);

// pre/post-processing around predict (prepost.hip): planar single-channel volumes [D][H][W]
constexpr int RS_LINEAR = 0, RS_NEAREST = 1;
struct ResampleArgs {
    const void* src; void* dst;
    int sD, sH, sW, dD, dH, dW;
    double fz, fy, fx;          // continuous input index of output voxel i along an axis = i * f (output spacing / input spacing)
    int mode;                   // RS_LINEAR | RS_NEAREST
};
void launch_resample3d(const ResampleArgs& a, int elem_type /* 0: f32, 1: u8 */, hipStream_t s);
size_t normalize_ws_bytes();
void launch_normalize_meanstd(const float* x, float* out, long long n, int clip, float lo, float hi, void* ws, hipStream_t s);
void launch_normalize_percentile(const float* x, float* out, long long n, float q_lo, float q_hi, void* ws, hipStream_t s);
void launch_gather_patches(const float* vol, int D, int H, int W, const int* origins, int nb, int pd, int ph, int pw, float* out, hipStream_t s);
void launch_stitch_mask(const unsigned char* masks, const int* origins, int nb, int pd, int ph, int pw, unsigned char* out, int D, int H, int W,
                        hipStream_t s);

void launch_mask(const float* probs, unsigned char* out, int N, int C, long long V, float threshold, int scale, hipStream_t s);

// Fused AdamW / Adam over the flat fp32 buffers; also clears nothing (grads are re-zeroed by the engine)
struct AdamArgs {
    float* p; const float* g; float* m; float* v;
    long long n;
    float lr, beta1, beta2, eps, weight_decay;
    int decoupled;          // 1 AdamW, 0 Adam (L2 folded in the gradient)
    float inv_scale;        // 1/loss_scale
    int* step;              // device step counter (incremented by the kernel launch)
    int* found_inf;         // device flag: when nonzero the update is skipped
};
void launch_grad_check(const float* g, long long n, int* found_inf, hipStream_t s);
void launch_adam(const AdamArgs& a, hipStream_t s, bool bump = true);      // bump = false: a StepRider of the next launch advances the counter

// channel-dropout multipliers: masks[l][n][ld] in {0, 1/(1-p)}
void launch_dropout_masks(float* masks, int L, int N, int ld, float p, unsigned long long seed, const int* step, hipStream_t s, bool bump = true);

}  // namespace seg
