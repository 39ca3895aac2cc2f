// Engine-internal host declarations shared by engine.hip (handle life cycle, forward / backward / train-step entry points), engine_plan.hip
// (network graph, workspace planner, forward and backward schedules) and capi_ops.hip (the stateless entry points of include/segengine.h).
//
// Segmentation engine: network graph (VNet / UNet, 2-D and 3-D), workspace planner, forward and
// backward schedules, C-ABI (include/segengine.h).  Host code only; kernels live in conv.hip,
// wgrad.hip, norm.hip, misc.hip.
//
// The graph is a list of steps over channels-last tensors:
//   UNIT  raw = conv(in0 [, in1 as virtual concat]) ; optional GroupNorm(8)+dropout+ReLU parameters
//   ACT   out = relu-gn(unit_a) [+ relu-gn(unit_b)] [+ residual tensor]
//   POOL  out = maxpool 2^d (UNet)          HEAD  logits/probs
// Backward is derived from the same list in reverse: every tensor collects up to three gradient
// contributions (residual fan-in, skip connections) that the GroupNorm-backward kernels sum on the
// fly, so no explicit `add` or `cat` tensor is ever materialised.
// Reference structure: networks/VNet3d.py:25-158, networks/Unet3d.py:6-86 (+ the 2-D twins).
#pragma once
#include <cstdlib>
#include <cstring>
#include <functional>
#include <string>
#include <vector>

#include "kernels.h"

using namespace seg;

namespace segi {

extern thread_local std::string g_err;          // seg_last_error(); defined in engine.hip
int fail(const std::string& m);                  // sets g_err, returns -1

enum ConvKind { CK_K3, CK_K1, CK_K2S2, CK_KT, CK_STEM3, CK_STEM1 };
enum StepType { ST_UNIT, ST_ACT, ST_POOL, ST_HEAD };

struct Param { std::string name; std::vector<int> shape; long long off; long long numel; };

struct Ten {
    int C, lvl;
    size_t off = 0;          // workspace byte offset
    bool image = false;
    bool virt = false;       // gradient of the head's input kept virtual (evaluated from dlogits and the head weights by its readers)
    std::vector<int> grads;  // gradient contribution tensors (ids)
};

struct Step {
    int type;
    // UNIT
    int ck = 0, in0 = -1, in1 = -1, raw = -1, Cin = 0, Cout = 0;
    int w = -1, b = -1, gn_w = -1, gn_b = -1;   // param indices (-1: absent)
    int cin_par = 0;                          // input channels of the weight PARAMETER when the conv reads a zero-padded image tensor (0: Cin)
    int mask_slot = -1;
    size_t stats = 0, scale = 0, shift = 0, mean = 0, rstd = 0, Q = 0, coef = 0;
    size_t wp_fwd = 0, wp_dg0 = 0, wp_dg1 = 0;
    bool fused_stem = false;                  // image stem evaluated inside the fused input block of its ACT step (stemx.hip)
    int stat_rep = 0;                         // replicas of the statistics buffers this unit's producers use (0 = STAT_REP)
    bool fold_fin = false;                    // statistics finalize folded into the consuming gn_act launch (no launch of its own)
    int x_fwd = -1, x_dg0 = -1, x_dg1 = -1;   // conv3x tiling of the forward / data-gradient launches (-1: conv3_kernel, row-major weights)
    int draw = -1;           // gradient wrt raw
    bool dual_dg = false;    // 1^d conv on a concat: both data-gradients come from one streaming launch (seg_conv_args.out1)
    int vact_unit = -1;      // >= 0: in0 is the (never written) activation of that UNIT step: this conv and its weight gradient read the unit's RAW output and apply
                             // GroupNorm + dropout + ReLU on load (conv_stream_kernel / wgrad_direct_kernel <..., ACT>)
    // ACT
    int ua = -1, ub = -1, res = -1, out = -1;
    bool vact = false;       // the output tensor is never written: its only reader applies the activation on load (Planner::plan)
    bool rq_fused = false;   // ACT (vact): the GroupNorm-backward sums of its unit ride on the data-gradient launch of the 1^d conv that reads it (no reduce launch)
    bool head_fused = false; // ACT: this pass also evaluates the 1^d head that reads its output; HEAD: evaluated by that pass (no launch of its own)
    // POOL / HEAD
    int in = -1;
};

inline size_t align_up(size_t v, size_t a = 256) { return (v + a - 1) / a * a; }
// profile class of a halo-conv launch.  "conv3" = the big-box tiling of the wide 32+-channel levels (one kernel symbol per
// network: 48^3 x 32 channels in the BASELINE VNet3d), "conv3_smallbox" = every other halo conv (16-channel top level, deep levels)
inline int conv3_class(int W, int Cin = 32) { return (W >= 32 && Cin >= 32) ? SEG_K_CONV3 : SEG_K_CONV3_SB; }

}  // namespace segi
using namespace segi;

struct seg_engine {
    int kind, ndim, in_ch, ncls, feat, dtype;
    std::vector<Param> params;
    long long nparam = 0;
    std::vector<Ten> tens;
    std::vector<Step> steps;
    std::vector<int> drop_ch;    // channels per dropout call
    int image_ten = -1;
    bool pad_img = false;        // the image tensor is zero-padded to 16 channels: 3-D inputs with > 1 channel (2-D: > 3) cannot take the fused image stem
                                 // (one MFMA K step holds taps x channels <= 32) and run through the ordinary 16-channel convs instead
    // plan
    int N = 0, D = 0, H = 0, W = 0;
    size_t ws_bytes = 0;
    size_t off_partial = 0, off_partial_stem1 = 0;
    size_t off_masks = 0, off_stats = 0, stats_bytes = 0, off_Q = 0, Q_bytes = 0, off_packdesc = 0, off_step = 0;
    bool q_clean = false;       // the forward pass's fill has cleared Q and no backward pass has used it yet
    std::vector<PackDesc> packdescs;   // dst/src stored as OFFSETS until bind
    long long pack_max = 0;
    bool planned = false;
    // bind
    float* p = nullptr; float* g = nullptr; char* ws = nullptr;
    float loss_scale = 1.f;
    int mask_mode = 0;
    int draws = 0;              // SEG_MASKS_RANDOM forwards issued so far: the device-side draw counter is restored from it at every
                                // seg_bind, so a re-plan (partial last batch, validation batch size, predict) does not restart the mask sequence
    std::vector<std::function<void(hipStream_t)>> fwd_ops, bwd_ops;
    std::vector<std::vector<int>> bwd_writes;   // parameter indices whose gradient each backward op finishes (bucketed all-reduce)
    const float* cur_x = nullptr; float* cur_logits = nullptr; float* cur_probs = nullptr;
    const float* cur_dlogits = nullptr;
    // weight gradients run on a side stream: they are off the backward critical path (only the optimiser needs them)
    hipStream_t side = nullptr;
    bool use_side = true;
    // weight re-layouts that only the backward pass reads run on the weight-gradient stream, next to the forward pass
    std::vector<char> pack_is_bwd;
    int npack_fwd = 0;
    bool pack_split = true, pack_bwd_pending = false;      // SEG_PACK_SPLIT=0: one launch on the caller's stream
    hipEvent_t pack_fork = nullptr, pack_done = nullptr;
    bool use_fold = true;       // SEG_GN_FOLD=0: finalize kernels between the GroupNorm passes (round-1 path)
    bool use_vhead = true;      // SEG_VHEAD=0: head_bwd writes its data-gradient tensor (round-1 path)
    bool head_din_needed = false;   // planning: some reader of the head's data-gradient cannot evaluate it on the fly
    int head_step = -1;
    bool use_stemx = true;      // SEG_STEMX=0: separate stem / GroupNorm / stem weight-gradient kernels (round-1 path)
    size_t off_partial_stemx = 0;
    bool use_conv3x = true;     // SEG_CONV3X=0: conv3_kernel for every halo conv (round-1 path)
    bool dual_gn_bwd = true;    // SEG_DUAL_GN=0: one GroupNorm-backward pass per branch of the VNet input block
    bool use_rq_fuse = true;    // SEG_RQ_FUSE=0: the GroupNorm-backward reduction of a VNet up-conv unit as a launch of its own
    bool use_head_fuse = true;  // SEG_HEAD_FUSE=0: the 1^d head as a launch of its own
    int use_vact = 1;           // SEG_VACT=0: the activation between a VNet up-conv and the 1^d conv on the concat is written as a tensor; 2: applied on load on small tensors too (tests)
    bool use_coop = true;       // SEG_GN_COOP=0: the deep levels' GroupNorm backward as reduce + apply launches (one-workgroup-per-group launch at 6^3)
    std::vector<hipEvent_t> ready_ev;
    hipEvent_t side_done = nullptr;
    hipEvent_t ar_ev = nullptr;          // orders the gradient-exchange stream behind / in front of the caller's stream (seg_train_step hooks)
    // in-library exchange (seg_set_rccl_comm): the communicator, the caller's ncclAllReduce and the library's own exchange stream
    typedef int (*rccl_allreduce_t)(const void*, void*, size_t, int, int, void*, hipStream_t);
    void* rccl_comm = nullptr;
    rccl_allreduce_t rccl_allreduce = nullptr;
    hipStream_t xchg = nullptr;
    size_t ready_used = 0;
    // Weight-gradient launches are queued and released to the weight-gradient stream in batches under ONE fork event: every hipEventRecord idles the
    // main stream for ~6 us, and the second stream has slack (it only has to finish before the optimiser), so a fork per weight gradient (35 per
    // step) cost more than it bought.  What rounds 2-5 measured around this scheme and did not keep - completion-flag forks (hipStreamWaitValue32),
    // two weight-gradient streams, sub-batched finest levels, a CU mask for the second stream, the last weight gradients on the main stream, late
    // release - is in profiles/HISTORY.md.
    struct Pend { std::function<void(hipStream_t)> f; };
    bool side_used = false;
    int n_event_forks = 0;              // of the current / last backward pass (seg_plan_count 2)
    std::vector<Pend> pending;
    int fork_batch = 3;      // weight gradients per fork event; measured on MI355X (VNet3d 4x96^3): 2 / 3 / 4 / 6 / 9 -> 1032 / 1031-1035 / 1025 / 1015 / 994
                             // volumes/s (profiles/r05_release_schedule_sweep.log)
    size_t cur_partial = 0;  // partial-tile scratch of the weight-gradient launch being issued
    // Halo weight gradients: the ordered reduce of the partial tiles (second stage) of every layer of a level visit runs as ONE launch behind the
    // visit's last weight-gradient kernel (launch_wgrad3_reduce; 8 reduce launches per VNet3d step instead of 20).  Each pending layer keeps a
    // partial-tile slot of its own until then.
    std::vector<Wgrad3Reduce> w3_pending;
    int w3_mode = 2;         // which launches share a reduce: 0 none (one reduce per layer), 1 a level visit, 2 the layers released to the queue together
    int w3_lvl = -1;
    hipStream_t w3_stream = nullptr;
    size_t off_partial3 = 0, partial3_stride = 0;
    void flush_w3() {
        if (w3_pending.empty()) return;
        double bytes = 0.0;
        for (auto& r : w3_pending) bytes += (double)(r.P / r.CP) * (r.Q / r.CQ) * r.nb * r.CP * r.ntap * r.CQ * 4.0 + 2.0 * (double)r.P * r.Q * r.ntap * 4.0;
        const int pi = prof_begin(w3_stream, SEG_K_WGRAD3, bytes, 0.0);
        launch_wgrad3_reduce(w3_pending.data(), (int)w3_pending.size(), w3_stream);
        prof_end(w3_stream, pi);
        w3_pending.clear();
    }
    // the partial-tile slot of the next halo weight gradient on level `lvl`, issued on stream `st`
    float* w3_slot(int lvl, hipStream_t st) {
        if (!w3_pending.empty() && (w3_lvl != lvl || w3_stream != st || (int)w3_pending.size() == W3_BATCH)) flush_w3();
        w3_lvl = lvl; w3_stream = st;
        return (float*)(ws + off_partial3 + w3_pending.size() * partial3_stride);
    }
    hipStream_t make_side() {
        // lowest priority: the weight gradients only have to finish before the optimiser, the main stream carries the critical
        // path.  At equal priority the command processor kept serving the side queue's back-to-back launches while the main
        // queue's next dispatch waited 30-125 us (profiles/r01_stream_gaps_step25.txt)
        hipStream_t st = nullptr;
        int lo = 0, hi = 0;
        (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
        if (lo != hi) (void)hipStreamCreateWithPriority(&st, hipStreamNonBlocking, lo);
        else (void)hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
        return st;
    }
    // SEG_HOLD_HEAVY_LVL=L (default off = -1): the weight gradients over >= hold_bytes tensors of the DECODER's top levels are not released while the
    // main stream still works on those bandwidth-bound levels; they are held until the backward pass reaches level L (24^3 for L = 2), where the main
    // stream's kernels are small and latency-bound and leave the HBM to the weight gradients
    int hold_lvl = -1;
    double hold_bytes = 64e6;                       // SEG_HOLD_HEAVY_MB
    bool hold_open = false;                         // the release level has been reached in this backward pass
    std::vector<Pend> held;
    // `bytes` = gradient tensor the kernel reads: a weight gradient over a big level is released at once (its inputs are final, and started
    // early it overlaps the bandwidth-bound top levels instead of the latency-bound deep chain)
    double fork_heavy_bytes = 16e6;
    void defer_wgrad(hipStream_t main, std::function<void(hipStream_t)> fn, double bytes = 0.0, int lvl = 0) {
#ifdef SEG_DIAG
        // diagnostic variant builds only (python tools/build_variant.py diag engine.hip,engine_plan.hip,capi_ops.hip -DSEG_DIAG; WRONG gradients): SEG_DIAG_NOWGRAD
        // drops weight-gradient launches to time what the step would cost without them - 1: all, 2: levels >= 2 (24^3 and deeper), 3: levels <= 1
        static const int nowgrad = knob_i("SEG_DIAG_NOWGRAD", 0);
        if (nowgrad == 1 || (nowgrad == 2 && lvl >= 2) || (nowgrad == 3 && lvl <= 1)) return;
#endif
        if (!use_side) { cur_partial = off_partial; fn(main); return; }
        Pend f{std::move(fn)};
        if (hold_lvl >= 0) {
            if (!hold_open && lvl >= hold_lvl) {
                hold_open = true;
                for (auto& h : held) pending.push_back(std::move(h));
                held.clear();
            }
            if (!hold_open && bytes >= hold_bytes) { held.push_back(std::move(f)); return; }
        }
        pending.push_back(std::move(f));
        // released at once: the weight gradients' inputs are final BEFORE the op's data-gradient kernel, so the second queue starts one
        // convolution earlier (1028-1029 vs 1016-1017 volumes/s against a release behind the op, profiles/r04_flag_forks_ab.log)
        if ((int)pending.size() >= fork_batch || bytes >= fork_heavy_bytes) flush_side(main);
    }
    void ensure_side() {
        if (side) return;
        side = make_side();
        (void)hipEventCreateWithFlags(&side_done, hipEventDisableTiming);
    }
    void flush_side(hipStream_t main) {
        if (pending.empty()) return;
        ensure_side();
        if (ready_used == ready_ev.size()) { hipEvent_t e; (void)hipEventCreateWithFlags(&e, hipEventDisableTiming); ready_ev.push_back(e); }
        ++n_event_forks;
        hipEvent_t e = ready_ev[ready_used++];
        (void)hipEventRecord(e, main);          // everything the queued weight gradients read has been produced on `main`
        (void)hipStreamWaitEvent(side, e, 0);
        for (auto& f : pending) { cur_partial = off_partial; f.f(side); }
        if (w3_mode == 2) flush_w3();
        side_used = true;
        pending.clear();
    }
    // every weight gradient issued so far is complete behind this point of its stream (joins, bucket boundaries)
    void finish_wgrads() { flush_w3(); }
    void join_side(hipStream_t main) {
        for (auto& h : held) pending.push_back(std::move(h));      // (a network without deep levels never reached the release level)
        held.clear();
        flush_side(main);
        finish_wgrads();
        if (use_side && side && (ready_used || side_used)) { (void)hipEventRecord(side_done, side); (void)hipStreamWaitEvent(main, side_done, 0); }
        ready_used = 0;
        side_used = false;
    }
    const float* mask_base(int slot) const {          // dropout multipliers of unit `slot`: table [slot][N][ld]
        return (const float*)(ws + off_masks) + (size_t)slot * N * ld_mask();
    }
    void run_ops(std::vector<std::function<void(hipStream_t)>>& ops, int b, int e, hipStream_t st) {
        for (int i = b; i < e; ++i) ops[i](st);
    }
    // seg_train_step: bookkeeping stores that ride on the step's own kernels (StepRider, kernels.h) - the dropout draw counter and the clear of
    // the overflow flag on the image ingest, the optimiser's step counter on the weight re-pack - and the loss workspace cleared by the head kernel
    bool ride_on = false;
    StepRider ride_ingest, ride_pack;
    double* ride_zero = nullptr; long long ride_zero_n = 0;
    bool head_zeroed = false;
    // one optimisation step captured as a HIP graph (seg_train_graph_*): the host side of a replay is ONE hipGraphLaunch
    hipGraph_t tgraph = nullptr;
    hipGraphExec_t tgraph_exec = nullptr;
    bool capturing = false;
    int tgraph_mask_mode = 0;
    hipStream_t tgraph_stream = nullptr;        // the stream of the last replay: a replay may still be running when the graph is dropped
    void drop_graph() {
        if (tgraph_exec) {
            if (tgraph_stream) (void)hipStreamSynchronize(tgraph_stream);
            (void)hipGraphExecDestroy(tgraph_exec); tgraph_exec = nullptr;
        }
        tgraph_stream = nullptr;
        if (tgraph) { (void)hipGraphDestroy(tgraph); tgraph = nullptr; }
    }
    // measurement (seg_profile_*)
    struct ProfRec { hipEvent_t a, b; int cls; double bytes, flops; };
    unsigned prof_mask = 0;
    std::vector<ProfRec> prof_pool;
    size_t prof_used = 0;
    int prof_begin(hipStream_t st, int cls, double bytes, double flops) {
        if (!(prof_mask >> cls & 1u)) return -1;
        if (prof_used == prof_pool.size()) {
            ProfRec r{};
            (void)hipEventCreate(&r.a); (void)hipEventCreate(&r.b);
            prof_pool.push_back(r);
        }
        ProfRec& r = prof_pool[prof_used];
        r.cls = cls; r.bytes = bytes; r.flops = flops;
        (void)hipEventRecord(r.a, st);
        return (int)prof_used++;
    }
    void prof_end(hipStream_t st, int idx) { if (idx >= 0) (void)hipEventRecord(prof_pool[idx].b, st); }
    double tbytes(int ten) const { return (double)N * vol(tens[ten].lvl) * tens[ten].C * esz(); }
    size_t esz() const { return dtype == DT_F32 ? 4 : 2; }
    int ld_mask() const { return 16 * feat; }
    int dim_d(int l) const { return ndim == 3 ? (D >> l) : 1; }
    int dim_h(int l) const { return H >> l; }
    int dim_w(int l) const { return W >> l; }
    long long vol(int l) const { return (long long)dim_d(l) * dim_h(l) * dim_w(l); }
};

namespace segi {
int check_handle(seg_handle h);
void build_network(seg_engine& e, int net_kind);          // engine_plan.hip: the step list of a VNet / UNet (Builder)
void plan_engine(seg_engine& e);                          // engine_plan.hip: workspace layout + forward / backward schedules for the planned shape (Planner)
int fill_loss(LossArgs& a, const float* logits, const void* target, int label_type, int n, int c, long long v, int loss_kind, float focal_alpha,
              float focal_gamma, void* ws);               // capi_ops.hip
// riders: the overflow flag was cleared and the step counter will be advanced by StepRiders of neighbouring launches (seg_train_step)
int adam_step_impl(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, long long numel, float lr, float beta1, float beta2, float eps,
                   float weight_decay, int decoupled, float inv_scale, int check_finite, int* state, void* stream, bool riders);   // capi_ops.hip
}  // namespace segi
