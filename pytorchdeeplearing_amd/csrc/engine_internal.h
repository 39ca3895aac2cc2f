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


// default of SEG_SUB_MB (group size of the sub-batched finest level, seg_engine::run_chain); 0 = whole-batch launches
#ifndef SEG_SUB_MB_DEFAULT
#define SEG_SUB_MB_DEFAULT 0.0
#endif
enum ConvKind { CK_K3, CK_K1, CK_K2S2, CK_KT, CK_STEM3, CK_STEM1 };
enum StepType { ST_UNIT, ST_ACT, ST_POOL, ST_HEAD };

struct Param { std::string name; std::vector<int> shape; long long off; long long numel; };

struct Ten {
    int C, lvl;
    size_t off = 0;          // workspace byte offset
    bool image = false;
    bool virt = false;       // gradient of the head's input kept virtual (evaluated from dlogits and the head weights by its readers)
    std::vector<int> grads;  // gradient contribution tensors (ids)
    int prod_step = -1, prod_which = 0;   // gradient tensors: the UNIT step whose conv3x data-gradient launch (in0 / in1 part) writes it
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
    int rq_unit[2] = {-1, -1};                // UNIT: its data-gradient launch (in0 / in1 part) also runs the GroupNorm-backward reduce of this unit
    bool rfused = false;                      // UNIT: its GroupNorm-backward reduce is done by the epilogue of the conv that produces its only gradient
    int stat_rep = 0;                         // replicas of the statistics buffers this unit's producers use (0 = STAT_REP)
    bool fold_fin = false;                    // statistics finalize folded into the consuming gn_act launch (no launch of its own)
    int x_fwd = -1, x_dg0 = -1, x_dg1 = -1;   // conv3x tiling of the forward / data-gradient launches (-1: conv3_kernel, row-major weights)
    int draw = -1;           // gradient wrt raw
    int vact_prod = -1;      // UNIT (3^d conv on conv3x): its input tensor is VIRTUAL - the launch reads the raw output of unit `vact_prod` and
                             // applies that unit's GroupNorm + dropout + ReLU while staging (forward conv and weight gradient alike)
    bool vact = false;       // ACT: the activated tensor is never written (its single consumer is a vact_prod conv)
    // ACT
    int ua = -1, ub = -1, res = -1, out = -1;
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
    bool use_vact = false;      // SEG_GN_VACT=1: GroupNorm + dropout + ReLU of the LUConv chains applied by the consuming halo conv and its weight
                                // gradient (12 launches and 12 activated tensors less in VNet3d).  Bit-identical, but measured 2.3 % SLOWER
                                // (863 vs 883 volumes/s, profiles/r03_vact_cumask_ab.log): the per-workgroup statistics fold costs the two L2
                                // round trips the 4.8 us launch cost, and the FUSE kernels spill scalars into the tap loop.  Opt-in.
    // (rounds 2-3 could fold the GroupNorm-backward reduce of a unit into the epilogue of the data-gradient conv producing its only gradient;
    // neutral in round 2, 0.4 % slower in round 3 - profiles/r03_epilogue_ab.log - and gone since the conv epilogue stores straight from the
    // accumulators)
    bool use_rfuse = false;     // experiments build, SEG_GN_RFUSE=1: the GroupNorm-backward reduce of a unit with a single gradient source runs in the epilogue of the
                                // conv3x data-gradient launch that writes that gradient (measured in round 5: slower than the separate pass, see Conv3xArgs::rq_*)
    bool use_fold = true;       // SEG_GN_FOLD=0: finalize kernels between the GroupNorm passes (round-1 path)
    bool use_vhead = true;      // SEG_VHEAD=0: head_bwd writes its data-gradient tensor (round-1 path)
    bool head_din_needed = false;   // planning: some reader of the head's data-gradient cannot evaluate it on the fly
    int head_step = -1;
    bool use_stemx = true;      // SEG_STEMX=0: separate stem / GroupNorm / stem weight-gradient kernels (round-1 path)
    size_t off_partial_stemx = 0;
    bool use_conv3x = true;     // SEG_CONV3X=0: conv3_kernel for every halo conv (round-1 path)
    bool dual_gn_bwd = true;    // SEG_DUAL_GN=0: one GroupNorm-backward pass per branch of the VNet input block
    bool stem_on_main = true;   // SEG_STEM_MAIN=0: 3^d stem weight gradient on the side stream (round-1 layout)
    int side_prio = 1;          // SEG_SIDE_PRIO=0: side stream at the default priority
    std::vector<hipEvent_t> ready_ev;
    hipEvent_t side_done = nullptr;
    hipEvent_t ar_ev = nullptr;          // orders the gradient-exchange stream behind / in front of the caller's stream (seg_train_step hooks)
    size_t ready_used = 0;
    // Weight-gradient launches are queued and released to the side stream in batches under ONE fork event: every
    // hipEventRecord idles the main stream for ~6 us, and the side stream has slack (it only has to finish before the
    // optimiser), so a fork per weight gradient (35 per step) cost more than it bought.
    // (round 3 built completion-flag forks - gn_bwd_apply publishing a per-unit sequence number, a one-wave kernel on the weight-gradient stream
    // spinning on it - to save the event record; on hardware the step ran at 451 vs 988 volumes/s with wrong gradients, profiles/r04_fork_flag_stress.json:
    // removed in round 4)
    struct Pend { std::function<void(hipStream_t)> f; };
    bool side_used = false;
    int n_event_forks = 0;              // of the current / last backward pass (seg_plan_count 2)
    std::vector<Pend> pending;
    int fork_batch = 3;      // measured on MI355X (VNet3d 4x96^3), round 1: 1 -> 641, 3 -> 645, 6 -> 649 volumes/s; round 2 with the
                             // heavy levels released at once: 6 -> 826, 3 -> 838
    // Up to two weight-gradient streams, each with its own partial-tile scratch: the kernels behind them run with 3-512 workgroups,
    // so two of them side by side fill CUs that one alone leaves idle (SEG_WGRAD_STREAMS, default in seg_create)
    int n_side = 1;
    hipStream_t side2 = nullptr;
    hipEvent_t side2_done = nullptr;
    size_t off_partial2 = 0, cur_partial = 0;
    int rr = 0;                 // round-robin cursor over the side streams
    hipStream_t make_side() {
        // lowest priority: the weight gradients only have to finish before the optimiser, the main stream carries the critical
        // path.  At equal priority the command processor kept serving the side queue's back-to-back launches while the main
        // queue's next dispatch waited 30-125 us (profiles/r01_stream_gaps_step25.txt)
        hipStream_t st = nullptr;
        // (hipExtStreamCreateWithCUMask was tried for this stream in round 3: ANY mask - 64 ... 192 CUs, contiguous or strided - halves the
        // step throughput, profiles/r03_vact_cumask_ab.log; not kept)
        int lo = 0, hi = 0;
        (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
        if (side_prio && lo != hi) (void)hipStreamCreateWithPriority(&st, hipStreamNonBlocking, lo);
        else (void)hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
        return st;
    }
    // `bytes` = gradient tensor the kernel reads: a weight gradient over a big level is released at once (its inputs are
    // final, and started early it overlaps the bandwidth-bound top levels instead of the latency-bound deep chain)
    // The last `tail_wgrads` deferred weight gradients of a backward pass stay on the main stream and run after its last op:
    // r02 trace — the main queue finished 140-210 us before the side queue and idled behind it.
    // Measured (profiles/r02_small_kernels_ab.log): 0 -> 849, 1 -> 842, 2 -> 836, 3 -> 832 volumes/s — the main stream is the
    // critical path once its idle time is gone, so the default keeps every weight gradient on the side stream.
    int n_deferred = 0, wgrad_seq = 0, tail_wgrads = 0;       // SEG_TAIL_WGRADS
    size_t off_partial_main = 0;
    std::vector<Pend> tail_pending;
    // SEG_HOLD_HEAVY_LVL=L (experiment, default off = -1): the weight gradients over >= hold_bytes tensors of the DECODER's top levels are
    // not released while the main stream still works on those bandwidth-bound levels; they are held until the backward pass reaches level L
    // (24^3 for L = 2), where the main stream's kernels are small and latency-bound and leave the HBM to the weight gradients
    int hold_lvl = -1;
    double hold_bytes = 64e6;                       // SEG_HOLD_HEAVY_MB
    bool hold_open = false;                         // the release level has been reached in this backward pass
    std::vector<Pend> held;
    // next_takes: the caller launches a kernel right behind this call that stores a released batch's number itself (take_sig)
    void defer_wgrad(hipStream_t main, std::function<void(hipStream_t)> fn, double bytes = 0.0, int lvl = 0, int sig_unit = -1, bool next_takes = false) {
        if (sub_active) {                                  // a chain runs group by group: the weight gradient is a whole-batch launch, queued once
            if (!sub_last) return;
            bytes *= (double)Nplan / (double)N;
        }
        if (!use_side) { const bool was = sub_suspend(); cur_partial = off_partial; fn(main); sub_resume(was); return; }
        (void)sig_unit;
        Pend f{std::move(fn)};
        if (wgrad_seq++ >= n_deferred - tail_wgrads) { tail_pending.push_back(std::move(f)); return; }
        if (hold_lvl >= 0) {
            if (!hold_open && lvl >= hold_lvl) {
                hold_open = true;
                for (auto& h : held) pending.push_back(std::move(h));
                held.clear();
                flush_due = true;
            }
            if (!hold_open && bytes >= hold_bytes) { held.push_back(std::move(f)); return; }
        }
        pending.push_back(std::move(f));
        // a full batch is released AFTER the op that queued it has enqueued its own main-stream kernels (maybe_flush): the dozen
        // launches + events of a batch take the host ~45 us, during which the main queue used to run dry (r02 trace: 138 us idle)
        if ((int)pending.size() >= fork_batch || bytes >= fork_heavy_bytes) { if (flush_late) flush_due = true; else flush_side(main, next_takes); }
    }
    double fork_heavy_bytes = 16e6;                 // SEG_FORK_HEAVY_MB
    // SEG_FLUSH_LATE=1: a full batch is released after the op that queued it has enqueued its own main-stream kernels (rounds 2-3, when the
    // host needed ~45 us for a batch and the main queue ran dry meanwhile).  Round 4: released at once - the weight gradients' inputs are final
    // BEFORE the op's data-gradient kernel, so the second queue starts one convolution earlier: 1028-1029 vs 1016-1017 volumes/s
    // (profiles/r04_flag_forks_ab.log; 1029 vs 1007 with event forks)
    bool flush_due = false, flush_late = false;
    void maybe_flush(hipStream_t main, bool next_takes = false) {
        if (flush_due) { flush_due = false; flush_side(main, next_takes); }
    }
    void ensure_side() {
        if (side) return;
        side = make_side();
        (void)hipEventCreateWithFlags(&side_done, hipEventDisableTiming);
        if (n_side > 1) { side2 = make_side(); (void)hipEventCreateWithFlags(&side2_done, hipEventDisableTiming); }
    }
    // ---- flag forks (round 4, opt-in: SEG_FORK=flag; tools/microbench/fork_cost.hip, profiles/r04_flag_forks_ab.log).  A hipEventRecord idles the
    // main queue ~6.4 us (19 forks per step = 3 % of it, profiles/r04_trace_timeline.txt).  With SEG_FORK=flag the weight-gradient queue instead
    // waits on a word in signal memory (hipStreamWaitValue32) and the word is stored by the first thread of the NEXT kernel the main queue runs
    // anyway: an in-order queue starts that kernel only after everything launched before it has completed and released its writes.  Sequence
    // numbers only grow, so a store also releases every older wait.  The kernels that follow a release - the data-gradient convolutions
    // (ForkSig), the GroupNorm-backward reduce / one-launch passes (GnBwdArgs::sig_flag) - take the number along in their arguments; anywhere else
    // a one-wave kernel stores it (~3 us).  A captured step (HIP graph) keeps event forks.
    // Measured: the main queue's fork gaps disappear (median gap 6.4 -> 0.2 us), and the step does not get faster - 1020-1021 vs 1015-1024
    // volumes/s with event forks on one lease, 1029 vs 1030 on another: with the early release below the weight-gradient queue is busy 92 % of the
    // backward window, so the main queue's saved 100 us are spent waiting at the join.  With the runtime's DEFAULT hipStreamWaitValue32 (a
    // one-thread polling kernel, __amd_rocclr_streamOpsWait, on the waiting queue) it is 2 % SLOWER (978-984): the process has to start with
    // GPU_STREAMOPS_CP_WAIT=1 (barrier-value packet: the command processor waits).  Kept opt-in.  (An own sleeping one-lane poll kernel on the waiting
    // queue was measured too: 996-997; removed.)
    int fork_mode = -1;                 // -1: decided on first use; 0: events; 1: flag (hipStreamWaitValue32)
    unsigned* fork_flag = nullptr;      // 8 bytes of signal memory
    unsigned fork_seq = 0;              // last number a weight-gradient queue was told to wait for
    unsigned sig_pending = 0;           // ... and not yet stored / handed to a kernel: nobody may wait on the side queues before it is
    int n_flag_forks = 0, n_sig_kernels = 0, n_sig_taken = 0;       // of the current / last backward pass (seg_plan_count 3, 7, 8)
    std::vector<char> bwd_sig;          // planning: per backward op, whether its first kernel takes a pending number along
    bool flag_forks() {
        if (fork_mode < 0) {
            fork_mode = 0;
            const char* e = xenv("SEG_FORK");
            const bool want = e && !strcmp(e, "flag");                // opt-in (see above)
            int can = 0;
            int devid = 0;
            (void)hipGetDevice(&devid);                                // the CURRENT device, not device 0
            if (want && hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, devid) == hipSuccess && can) {
                if (hipExtMallocWithFlags((void**)&fork_flag, 8, hipMallocSignalMemory) == hipSuccess && fork_flag) {
                    launch_fork_signal(fork_flag, 0u, nullptr);
                    fork_mode = hipDeviceSynchronize() == hipSuccess ? 1 : 0;
                }
                (void)hipGetLastError();
            }
        }
        return fork_mode >= 1 && !capturing;
    }
    unsigned* take_sig(unsigned& seq) {                    // called by the op whose first kernel stores the number itself
        if (!sig_pending) return nullptr;
        seq = sig_pending; sig_pending = 0; ++n_sig_taken;
        return fork_flag;
    }
    void emit_sig(hipStream_t main) {                      // nobody took it: a one-wave kernel on the main stream
        if (!sig_pending) return;
        launch_fork_signal(fork_flag, sig_pending, main);
        sig_pending = 0; ++n_sig_kernels;
    }
    void release_waiters() {                               // before the host waits for a weight-gradient queue outside a step (plan / bind / destroy)
        if (fork_mode >= 1 && fork_flag && fork_seq) { launch_fork_signal(fork_flag, fork_seq, nullptr); sig_pending = 0; }
    }
    void flush_side(hipStream_t main, bool next_takes = false) {
        if (pending.empty()) return;
        const bool was_sub = sub_suspend();                // the queued launches are whole-batch
        flush_side_full(main, next_takes);
        sub_resume(was_sub);
    }
    void flush_side_full(hipStream_t main, bool next_takes) {
        ensure_side();
        if (flag_forks()) {
            if (fork_seq >= (1u << 30)) {                  // (once per ~5e7 steps) start the numbers over with both queues drained
                emit_sig(main);
                (void)hipStreamSynchronize(main); (void)hipStreamSynchronize(side); if (side2) (void)hipStreamSynchronize(side2);
                launch_fork_signal(fork_flag, 0u, main); (void)hipStreamSynchronize(main);
                fork_seq = 0;
            }
            const unsigned seq = ++fork_seq;
            (void)hipStreamWaitValue32(side, fork_flag, seq, hipStreamWaitValueGte, 0xffffffffu);
            if (side2) (void)hipStreamWaitValue32(side2, fork_flag, seq, hipStreamWaitValueGte, 0xffffffffu);
            sig_pending = seq;                             // (a number still pending from an earlier release is covered by this larger one)
            ++n_flag_forks;
        } else {
        if (ready_used == ready_ev.size()) { hipEvent_t e; (void)hipEventCreateWithFlags(&e, hipEventDisableTiming); ready_ev.push_back(e); }
        ++n_event_forks;
        hipEvent_t e = ready_ev[ready_used++];
        (void)hipEventRecord(e, main);          // everything the queued weight gradients read has been produced on `main`
        (void)hipStreamWaitEvent(side, e, 0);
        if (side2) (void)hipStreamWaitEvent(side2, e, 0);
        }
        for (auto& f : pending) {
            const bool second = side2 && (rr++ & 1);
            cur_partial = second ? off_partial2 : off_partial;
            f.f(second ? side2 : side);
        }
        side_used = true;
        pending.clear();
        if (!next_takes) emit_sig(main);
    }
    void join_side(hipStream_t main) {
        for (auto& h : held) pending.push_back(std::move(h));      // (a network without deep levels never reached the release level)
        held.clear();
        flush_side(main);
        emit_sig(main);                                            // the main stream is about to wait for the weight-gradient queues
        for (auto& f : tail_pending) { cur_partial = off_partial_main; f.f(main); }
        tail_pending.clear();
        if (use_side && side && (ready_used || side_used)) {
            (void)hipEventRecord(side_done, side); (void)hipStreamWaitEvent(main, side_done, 0);
            if (side2) { (void)hipEventRecord(side2_done, side2); (void)hipStreamWaitEvent(main, side2_done, 0); }
        }
        ready_used = 0;
        side_used = false;
    }
    // ---- sub-batch execution of the finest level(s) (SEG_SUB_MB).  At 4 x 96^3 one 16-channel tensor is 113 MB, so a
    // consumer never finds what its producer just wrote in the 256 MB memory-side cache.  GroupNorm statistics and dropout masks are per sample
    // (networks/VNet3d.py:9), so maximal runs of consecutive ops whose units all live on the finest level ("chains": the decoder top
    // convT -> act -> 1^d conv -> act -> LUConv -> act -> head, and its backward twin) are executed group of samples by group of samples: all ops
    // of the chain for samples [n0, n0 + nb), then the next group.  No kernel knows: the op lambdas read tensor offsets / N / per-unit buffers at
    // call time, and run_chain shifts exactly those for the duration of a group (a unit's statistics replicas are then laid out
    // [group][rep][nb][C][2] instead of [rep][N][C][2]; every toucher of a finest-level unit's statistics is a chain op in BOTH passes, so the
    // layout is consistent).  Weight gradients stay whole-batch: defer_wgrad queues them on the LAST group only, and anything released to the
    // weight-gradient stream while a group is active runs with the shifts suspended (sub_suspend / sub_resume).
    int Nplan = 0;                       // the planned batch (N is the group size while a chain runs)
    int sub_nb = 0;                      // samples per group; 0 = off
    int sub_lvl = 0;                     // finest levels that are sub-batched (SEG_SUB_LVL)
    double sub_mb = -1.0;                // SEG_SUB_MB: group size = as many samples as keep a 16-channel finest-level tensor under this many MB (0 = off)
    std::vector<std::pair<int, int>> fwd_chains, bwd_chains;      // [begin, end) op ranges
    std::vector<char> bwd_sub;           // planning: per backward op, whether it may run per group
    bool sub_active = false, sub_last = false;
    int sub_n0 = 0, sub_cur = 0;
    void sub_shift(int n0, long long sign) {
        auto mv = [&](size_t& off, long long per) { off = (size_t)((long long)off + sign * n0 * per); };
        for (auto& t : tens) mv(t.off, (long long)vol(t.lvl) * t.C * (long long)esz());
        for (auto& s : steps) {
            if (s.type != ST_UNIT || s.gn_w < 0) continue;
            mv(s.stats, (long long)STAT_REP * s.Cout * 2 * 8);
            mv(s.Q, (long long)STAT_REP * s.Cout * 2 * 8);
            mv(s.scale, (long long)s.Cout * 4);
            mv(s.shift, (long long)s.Cout * 4);
            mv(s.mean, (long long)GN_GROUPS * 4);
            mv(s.rstd, (long long)GN_GROUPS * 4);
            mv(s.coef, (long long)s.Cout * 3 * 4);
        }
        const long long v0 = vol(0);
        if (cur_x) cur_x += sign * n0 * in_ch * v0;
        if (cur_logits) cur_logits += sign * n0 * ncls * v0;
        if (cur_probs) cur_probs += sign * n0 * ncls * v0;
        if (cur_dlogits) cur_dlogits += sign * n0 * ncls * v0;
    }
    void sub_enter(int n0, int nb) { sub_shift(n0, +1); N = nb; sub_n0 = n0; sub_cur = nb; sub_active = true; }
    void sub_leave() { sub_shift(sub_n0, -1); N = Nplan; sub_n0 = 0; sub_active = false; }
    // whole-batch work issued from inside a group (a release of queued weight gradients): shifts off, run, shifts back on
    bool sub_suspend() { if (!sub_active) return false; const int n0 = sub_n0; sub_leave(); sub_n0 = n0; return true; }
    void sub_resume(bool was) { if (was) { const int n0 = sub_n0; sub_enter(n0, sub_cur); } }
    const float* mask_base(int slot) const {          // dropout multipliers of unit `slot`: table [slot][Nplan][ld]; a group starts at its first sample
        return (const float*)(ws + off_masks) + ((size_t)slot * Nplan + (sub_active ? sub_n0 : 0)) * ld_mask();
    }
    void run_chain(std::vector<std::function<void(hipStream_t)>>& ops, int b, int e, hipStream_t st, bool bwd) {
        const int nb = sub_nb;
        for (int n0 = 0; n0 < Nplan; n0 += nb) {
            sub_last = n0 + nb >= Nplan;
            sub_enter(n0, nb);
            for (int i = b; i < e; ++i) { ops[i](st); if (bwd) emit_sig(st); if (bwd && sub_last) maybe_flush(st); }
            sub_leave();
        }
        sub_last = false;
    }
    void run_ops(std::vector<std::function<void(hipStream_t)>>& ops, const std::vector<std::pair<int, int>>& chains, int b, int e, hipStream_t st, bool bwd) {
        size_t ci = 0;
        for (int i = b; i < e;) {
            while (ci < chains.size() && chains[ci].second <= i) ++ci;
            if (sub_nb > 0 && sub_nb < Nplan && ci < chains.size() && chains[ci].first <= i) {
                const int ce = chains[ci].second < e ? chains[ci].second : e;
                run_chain(ops, i, ce, st, bwd);
                i = ce;
            } else {
                ops[i](st);
                if (bwd) {
                    emit_sig(st);                          // a number the op did not take along (its first kernel was not the expected one)
                    maybe_flush(st, i + 1 < e && i + 1 < (int)bwd_sig.size() && bwd_sig[i + 1]);
                }
                ++i;
            }
        }
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
    // arguments of a GroupNorm-backward reduce folded into a data-gradient launch (unit index, or -1: none)
    Conv3xReduce reduce_args(int ui) const {
        Conv3xReduce r{nullptr, nullptr, nullptr, nullptr, STAT_REP};
        if (ui < 0) return r;
        const Step& u = steps[ui];
        r.y = ws + tens[u.raw].off;
        r.scale = (const float*)(ws + u.scale); r.shift = (const float*)(ws + u.shift);
        r.Q = (double*)(ws + u.Q);
        r.rep = use_fold ? stat_rep_for(vol(tens[u.raw].lvl)) : STAT_REP;
        return r;
    }
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
