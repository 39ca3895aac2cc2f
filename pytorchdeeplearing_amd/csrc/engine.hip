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
#include <cstdlib>
#include <cstring>
#include <functional>
#include <string>
#include <vector>

#include "kernels.h"

using namespace seg;

namespace {

thread_local std::string g_err;
int fail(const std::string& m) { g_err = m; return -1; }

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

}  // namespace

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
    // GPU_STREAMOPS_CP_WAIT=1 (barrier-value packet: the command processor waits).  Kept opt-in; SEG_FORK=spin: own sleeping poll kernel (996).
    int fork_mode = -1;                 // -1: decided on first use; 0: events; 1: flag, hipStreamWaitValue32; 2: flag, own one-lane polling kernel
    unsigned* fork_flag = nullptr;      // 8 bytes of signal memory
    unsigned fork_seq = 0;              // last number a weight-gradient queue was told to wait for
    unsigned sig_pending = 0;           // ... and not yet stored / handed to a kernel: nobody may wait on the side queues before it is
    int n_flag_forks = 0, n_sig_kernels = 0, n_sig_taken = 0;       // of the current / last backward pass (seg_plan_count 3, 7, 8)
    std::vector<char> bwd_sig;          // planning: per backward op, whether its first kernel takes a pending number along
    bool flag_forks() {
        if (fork_mode < 0) {
            fork_mode = 0;
            const char* e = getenv("SEG_FORK");
            const bool want = e && strcmp(e, "event") != 0;           // opt-in (see above)
            int can = 0;
            if (want && hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, 0) == hipSuccess && can) {
                if (hipExtMallocWithFlags((void**)&fork_flag, 8, hipMallocSignalMemory) == hipSuccess && fork_flag) {
                    launch_fork_signal(fork_flag, 0u, nullptr);
                    fork_mode = hipDeviceSynchronize() == hipSuccess ? ((e && !strcmp(e, "spin")) ? 2 : 1) : 0;
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
            if (fork_mode == 2) { launch_fork_wait(fork_flag, seq, side); if (side2) launch_fork_wait(fork_flag, seq, side2); }
            else {
                (void)hipStreamWaitValue32(side, fork_flag, seq, hipStreamWaitValueGte, 0xffffffffu);
                if (side2) (void)hipStreamWaitValue32(side2, fork_flag, seq, hipStreamWaitValueGte, 0xffffffffu);
            }
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
    double tbytes(int ten) const { return (double)N * vol(tens[ten].lvl) * tens[ten].C * esz(); }
    size_t esz() const { return dtype == DT_F32 ? 4 : 2; }
    int ld_mask() const { return 16 * feat; }
    int dim_d(int l) const { return ndim == 3 ? (D >> l) : 1; }
    int dim_h(int l) const { return H >> l; }
    int dim_w(int l) const { return W >> l; }
    long long vol(int l) const { return (long long)dim_d(l) * dim_h(l) * dim_w(l); }
};

namespace {

// ------------------------------------------------------------------------------------------------
// graph construction
// ------------------------------------------------------------------------------------------------
struct Builder {
    seg_engine& e;
    int nd;
    explicit Builder(seg_engine& e_) : e(e_), nd(e_.ndim) {}

    int param(const std::string& name, std::vector<int> shape) {
        Param p; p.name = name; p.shape = shape; p.off = e.nparam; p.numel = 1;
        for (int s : shape) p.numel *= s;
        // keep every tensor 64-float aligned inside the flat buffer (vector loads, 256-B alignment)
        e.nparam += (p.numel + 63) / 64 * 64;
        e.params.push_back(p);
        return (int)e.params.size() - 1;
    }
    std::vector<int> kshape(int a, int b, int k) {
        std::vector<int> s{a, b};
        for (int i = 0; i < nd; ++i) s.push_back(k);
        return s;
    }
    int tensor(int C, int lvl, bool image = false) {
        Ten t; t.C = C; t.lvl = lvl; t.image = image;
        e.tens.push_back(t);
        return (int)e.tens.size() - 1;
    }
    // conv (+ optional GroupNorm params gw/gb: -2 => create "<gn>.weight/.bias")
    int unit(int ck, const std::string& cname, bool bias, int in0, int in1, int Cout, int lvl_out,
             const std::string& gname, int gw = -2, int gb = -2, bool has_gn = true) {
        Step s; s.type = ST_UNIT; s.ck = ck; s.in0 = in0; s.in1 = in1;
        s.Cin = e.tens[in0].C + (in1 >= 0 ? e.tens[in1].C : 0);
        s.Cout = Cout;
        if (e.tens[in0].image && e.pad_img) s.cin_par = e.in_ch;        // the parameter keeps the reference's shape [Cout][image channels][k^d]
        const int cpar = s.cin_par ? s.cin_par : s.Cin;
        const int k = (ck == CK_K3 || ck == CK_STEM3) ? 3 : (ck == CK_K2S2 || ck == CK_KT) ? 2 : 1;
        s.w = param(cname + ".weight", ck == CK_KT ? kshape(s.Cin, Cout, k) : kshape(Cout, cpar, k));
        if (bias) s.b = param(cname + ".bias", {Cout});
        if (has_gn) {
            if (gw == -2) { gw = param(gname + ".weight", {Cout}); gb = param(gname + ".bias", {Cout}); }
            s.gn_w = gw; s.gn_b = gb;
            s.mask_slot = (int)e.drop_ch.size();
            e.drop_ch.push_back(Cout);
        }
        s.raw = tensor(Cout, lvl_out);
        e.steps.push_back(s);
        return (int)e.steps.size() - 1;
    }
    int act(int ua, int ub, int res) {
        Step s; s.type = ST_ACT; s.ua = ua; s.ub = ub; s.res = res;
        const Ten& r = e.tens[e.steps[ua].raw];
        s.out = tensor(r.C, r.lvl);
        e.steps.push_back(s);
        return s.out;
    }
    int pool(int in) {
        Step s; s.type = ST_POOL; s.in = in;
        s.out = tensor(e.tens[in].C, e.tens[in].lvl + 1);
        e.steps.push_back(s);
        return s.out;
    }
    void head(int in, const std::string& cname) {
        Step s; s.type = ST_HEAD; s.in = in; s.Cin = e.tens[in].C; s.Cout = e.ncls;
        s.w = param(cname + ".weight", kshape(e.ncls, s.Cin, 1));
        s.b = param(cname + ".bias", {e.ncls});
        e.steps.push_back(s);
    }

    void build_vnet() {   // networks/VNet3d.py:102-158
        const int F = e.feat;
        const int x = tensor(e.pad_img ? 16 : e.in_ch, 0, true);
        e.image_ten = x;
        // InputTransition (VNet3d.py:25-43): parameter order conv1, conv2, bn1; ONE GroupNorm for both branches
        const int ua = unit(e.pad_img ? CK_K3 : CK_STEM3, "in_tr.conv1", true, x, -1, F, 0, "", -1, -1, false);
        const int ub = unit(e.pad_img ? CK_K1 : CK_STEM1, "in_tr.conv2", true, x, -1, F, 0, "", -1, -1, false);
        const int gw = param("in_tr.bn1.weight", {F}), gb = param("in_tr.bn1.bias", {F});
        for (int u : {ua, ub}) {
            e.steps[u].gn_w = gw; e.steps[u].gn_b = gb;
            e.steps[u].mask_slot = (int)e.drop_ch.size();
            e.drop_ch.push_back(F);
        }
        int prev = act(ua, ub, -1);
        std::vector<int> skips{prev};
        const int nconv_down[4] = {2, 3, 3, 3};
        for (int l = 1; l <= 4; ++l) {   // DownTransition (VNet3d.py:46-59)
            const int C = F << l;
            const std::string pre = "down_tr" + std::to_string(32 << (l - 1));
            const int ud = unit(CK_K2S2, pre + ".down_conv", true, prev, -1, C, l, pre + ".bn1");
            const int down = act(ud, -1, -1);
            int t = down;
            for (int i = 0; i < nconv_down[l - 1]; ++i) {
                const std::string op = pre + ".ops." + std::to_string(i);
                const int u = unit(CK_K3, op + ".conv1", true, t, -1, C, l, op + ".bn1");
                t = act(u, -1, i == nconv_down[l - 1] - 1 ? down : -1);
            }
            prev = t;
            skips.push_back(prev);
        }
        skips.pop_back();
        const int nconv_up[4] = {3, 3, 2, 1};
        for (int k = 0; k < 4; ++k) {    // UpTransition (VNet3d.py:62-80): parameter order up_conv, bn, ops, conv
            const int l = 3 - k, C = F << l;
            const std::string pre = "up_tr" + std::to_string(256 >> k);
            const int skip = skips.back(); skips.pop_back();
            const int uu = unit(CK_KT, pre + ".up_conv", true, prev, -1, C, l, pre + ".bn");
            const int gwu = e.steps[uu].gn_w, gbu = e.steps[uu].gn_b;
            const int up = act(uu, -1, -1);
            // the LUConv parameters are registered BEFORE `conv` in the reference module; keep state_dict order
            // by creating the ops' parameters first and the 1^d conv's afterwards.
            std::vector<int> opw, opb, opgw, opgb;
            for (int i = 0; i < nconv_up[k]; ++i) {
                const std::string op = pre + ".ops." + std::to_string(i);
                opw.push_back(param(op + ".conv1.weight", kshape(C, C, 3)));
                opb.push_back(param(op + ".conv1.bias", {C}));
                opgw.push_back(param(op + ".bn1.weight", {C}));
                opgb.push_back(param(op + ".bn1.bias", {C}));
            }
            const int cw = param(pre + ".conv.weight", kshape(C, 2 * C, 1));
            const int cb = param(pre + ".conv.bias", {C});
            const int uc = unit_preparam(CK_K1, cw, cb, up, skip, C, l, gwu, gbu);
            const int xcat = act(uc, -1, -1);
            int t = xcat;
            for (int i = 0; i < nconv_up[k]; ++i) {
                const int u = unit_preparam(CK_K3, opw[i], opb[i], t, -1, C, l, opgw[i], opgb[i]);
                t = act(u, -1, i == nconv_up[k] - 1 ? xcat : -1);
            }
            prev = t;
        }
        head(prev, "out_tr.conv");
    }
    int unit_preparam(int ck, int w, int b, int in0, int in1, int Cout, int lvl, int gw, int gb) {
        Step s; s.type = ST_UNIT; s.ck = ck; s.in0 = in0; s.in1 = in1;
        s.Cin = e.tens[in0].C + (in1 >= 0 ? e.tens[in1].C : 0);
        s.Cout = Cout; s.w = w; s.b = b; s.gn_w = gw; s.gn_b = gb;
        s.mask_slot = (int)e.drop_ch.size();
        e.drop_ch.push_back(Cout);
        s.raw = tensor(Cout, lvl);
        e.steps.push_back(s);
        return (int)e.steps.size() - 1;
    }

    int unet_block(const std::string& mod, const std::string& name, int in0, int in1, int C, int lvl, bool first) {
        // Unet3d.py:64-86: conv3(no bias) GN drop relu, twice
        const int u1 = unit((first && !e.pad_img) ? CK_STEM3 : CK_K3, mod + "." + name + "conv1", false, in0, in1, C, lvl, mod + "." + name + "norm1");
        const int a1 = act(u1, -1, -1);
        const int u2 = unit(CK_K3, mod + "." + name + "conv2", false, a1, -1, C, lvl, mod + "." + name + "norm2");
        return act(u2, -1, -1);
    }
    void build_unet() {   // networks/Unet3d.py:6-62
        const int F = e.feat;
        const int x = tensor(e.pad_img ? 16 : e.in_ch, 0, true);
        e.image_ten = x;
        int t = x;
        std::vector<int> enc;
        for (int l = 0; l < 4; ++l) {
            const std::string nm = "enc" + std::to_string(l + 1);
            const int en = unet_block("encoder" + std::to_string(l + 1), nm, t, -1, F << l, l, l == 0);
            enc.push_back(en);
            t = pool(en);
        }
        t = unet_block("bottleneck", "bottleneck", t, -1, F << 4, 4, false);
        for (int l = 3; l >= 0; --l) {
            const std::string up = "upconv" + std::to_string(l + 1);
            const int uu = unit(CK_KT, up, true, t, -1, F << l, l, "", -1, -1, false);
            // plain ConvTranspose: its raw output IS the activation fed to the concat
            t = unet_block("decoder" + std::to_string(l + 1), "dec" + std::to_string(l + 1), e.steps[uu].raw, enc[l], F << l, l, false);
        }
        head(t, "conv");
    }
};

Taps make_taps(int ndim, int k, int pad) {
    Taps t; t.n = 0;
    const int kd = ndim == 3 ? k : 1;
    for (int a = 0; a < kd; ++a)
        for (int b = 0; b < k; ++b)
            for (int c = 0; c < k; ++c) {
                t.d[t.n] = (int8_t)(ndim == 3 ? a - pad : 0);
                t.h[t.n] = (int8_t)(b - pad);
                t.w[t.n] = (int8_t)(c - pad);
                ++t.n;
            }
    return t;
}

// weight-gradient launch arguments of a UNIT (pointers are null until the engine is bound)
WgradArgs make_wgrad_args(const seg_engine& E, const Step& s, int draw) {
    const Ten& i0 = E.tens[s.in0];
    const Ten& ro = E.tens[s.raw];
    const int li = i0.lvl, lo = ro.lvl;
    const int T = (s.ck == CK_K3 || s.ck == CK_STEM3) ? (E.ndim == 3 ? 27 : 9)
                  : (s.ck == CK_K2S2 || s.ck == CK_KT) ? (E.ndim == 3 ? 8 : 4) : 1;
    char* ws = E.ws;
    auto P = [&](size_t off) -> const void* { return ws ? ws + off : nullptr; };
    WgradArgs w{};
    w.dw = E.g ? E.g + E.params[s.w].off : nullptr; w.N = E.N; w.sT = 1; w.sQ = T;
    if (s.ck == CK_KT) {
        // dW[ci][co][a] = sum_coarse X[m][ci] * dY[2m+a][co]
        w.dr = P(i0.off); w.P = s.Cin;
        w.x0 = draw >= 0 ? P(E.tens[draw].off) : nullptr; w.C0 = s.Cout; w.x1 = nullptr; w.C1 = 0; w.Q = s.Cout;
        w.ID = E.dim_d(lo); w.IH = E.dim_h(lo); w.IW = E.dim_w(lo);
        w.OD = E.dim_d(li); w.OH = E.dim_h(li); w.OW = E.dim_w(li);
        w.sd = E.ndim == 3 ? 2 : 1; w.sh = 2; w.sw = 2;
        w.taps = make_taps(E.ndim, 2, 0);
        w.sP = (long long)s.Cout * T;
    } else {
        w.dr = draw >= 0 ? P(E.tens[draw].off) : nullptr; w.P = s.Cout;
        w.x0 = P(i0.off); w.C0 = i0.C;
        w.x1 = s.in1 >= 0 ? P(E.tens[s.in1].off) : nullptr;
        w.C1 = s.in1 >= 0 ? E.tens[s.in1].C : 0;
        w.Q = s.Cin;
        w.ID = E.dim_d(li); w.IH = E.dim_h(li); w.IW = E.dim_w(li);
        w.OD = E.dim_d(lo); w.OH = E.dim_h(lo); w.OW = E.dim_w(lo);
        const int k = (s.ck == CK_K3 || s.ck == CK_STEM3) ? 3 : s.ck == CK_K2S2 ? 2 : 1;
        const int str = s.ck == CK_K2S2 ? 2 : 1;
        w.sd = E.ndim == 3 ? str : 1; w.sh = str; w.sw = str;
        w.taps = make_taps(E.ndim, k, k == 3 ? 1 : 0);
        w.sP = (long long)(s.cin_par ? s.cin_par : s.Cin) * T;
        if (s.ck == CK_STEM3 || s.ck == CK_STEM1) { w.stem = 1; w.Q = T * s.Cin; }
    }
    return w;
}

// arguments of the fused input block behind ACT step `s` (pointers valid once the engine is bound)
seg_stemx_args stemx_args(const seg_engine& E, const Step& s) {
    const Step& ua = E.steps[s.ua];
    seg_stemx_args x{};
    x.img = E.ws + E.tens[ua.in0].off;
    x.w3 = E.ws + ua.wp_fwd; x.bias3 = ua.b >= 0 ? E.p + E.params[ua.b].off : nullptr;
    x.stats3 = (double*)(E.ws + ua.stats); x.scale3 = (float*)(E.ws + ua.scale); x.shift3 = (float*)(E.ws + ua.shift);
    x.Q3 = (double*)(E.ws + ua.Q); x.coef3 = (float*)(E.ws + ua.coef);
    if (s.ub >= 0) {
        const Step& ub = E.steps[s.ub];
        x.w1 = E.ws + ub.wp_fwd; x.bias1 = ub.b >= 0 ? E.p + E.params[ub.b].off : nullptr;
        x.stats1 = (double*)(E.ws + ub.stats); x.scale1 = (float*)(E.ws + ub.scale); x.shift1 = (float*)(E.ws + ub.shift);
        x.Q1 = (double*)(E.ws + ub.Q); x.coef1 = (float*)(E.ws + ub.coef);
    }
    x.out = E.ws + E.tens[s.out].off;
    x.partial = (float*)(E.ws + E.off_partial_stemx);
    x.N = E.N; x.D = E.dim_d(0); x.H = E.dim_h(0); x.W = E.dim_w(0); x.Cimg = E.tens[ua.in0].C;
    return x;
}

// ------------------------------------------------------------------------------------------------
// planning: workspace layout + forward / backward schedules
// ------------------------------------------------------------------------------------------------
struct Planner {
    seg_engine& e;
    size_t cur = 0;
    explicit Planner(seg_engine& e_) : e(e_) {}
    size_t alloc(size_t bytes) { size_t o = cur; cur = align_up(cur + bytes); return o; }
    size_t ten_bytes(const Ten& t) const { return (size_t)e.N * e.vol(t.lvl) * t.C * e.esz(); }
    int new_grad(int like) {
        Ten t; t.C = e.tens[like].C; t.lvl = e.tens[like].lvl;
        t.off = alloc(ten_bytes(t));
        e.tens.push_back(t);
        return (int)e.tens.size() - 1;
    }
    template <class T = void> T* P(size_t off) const { return (T*)(e.ws + off); }

    int ntaps(int ck) const {
        const int k = (ck == CK_K3 || ck == CK_STEM3) ? 3 : (ck == CK_K2S2 || ck == CK_KT) ? 2 : 1;
        return e.ndim == 3 ? k * k * k : k * k;
    }
    bool pack_bwd = false;     // the descriptors added while set feed the backward pass only (data-gradient layouts)
    void add_pack(size_t dst, long long src_off, int R1, int R2, int T, int Cc, long long s1, long long s2, long long sT, long long sC, int flip,
                  int frag = 0, int csrc = 0) {
        PackDesc d;
        d.frag = frag;
        d.csrc = csrc;
        d.src = (const float*)(uintptr_t)src_off;   // offsets; resolved in seg_bind
        d.dst = (void*)(uintptr_t)dst;
        d.R1 = R1; d.R2 = R2; d.T = T; d.Cc = Cc;
        d.Kpad = (T * Cc + 31) / 32 * 32;
        d.s1 = s1; d.s2 = s2; d.sT = sT; d.sC = sC; d.flipT = flip;
        e.packdescs.push_back(d);
        e.pack_is_bwd.push_back(pack_bwd ? 1 : 0);
        const long long tot = (long long)R1 * R2 * d.Kpad;
        if (tot > e.pack_max) e.pack_max = tot;
    }
    size_t alloc_pack(int rows, int K) { return alloc((size_t)rows * ((K + 31) / 32 * 32) * e.esz()); }

    void plan() {
        seg_engine& E = e;
        const int N = E.N, dt = E.dtype;
        E.fwd_ops.clear(); E.bwd_ops.clear(); E.bwd_writes.clear(); E.packdescs.clear(); E.pack_is_bwd.clear(); E.pack_max = 0; E.n_deferred = 0;
        E.bwd_sub.clear(); E.bwd_sig.clear(); E.fwd_chains.clear(); E.bwd_chains.clear();
        // drop gradient tensors of a previous plan
        size_t nfw = 0;
        for (auto& s : E.steps) {
            nfw = std::max<size_t>(nfw, std::max(s.raw, s.out) + 1);
            s.draw = -1; s.vact = false; s.vact_prod = -1;
        }
        E.tens.resize(std::max<size_t>(nfw, (size_t)E.image_ten + 1));
        for (auto& t : E.tens) t.grads.clear();

        // ---- fused input block: an ACT whose unit(s) are image stems (3^d [+ 1^d]) without a residual
        for (auto& st_ : E.steps) st_.fused_stem = false;
        if (E.use_stemx && E.feat == 16 && (long long)E.vol(0) * 16 * 4 < (1ll << 31))
            for (auto& st_ : E.steps)
                if (st_.type == ST_ACT && st_.res < 0 && E.steps[st_.ua].ck == CK_STEM3 && E.steps[st_.ua].gn_w >= 0 &&
                    (st_.ub < 0 || (E.steps[st_.ub].ck == CK_STEM1 && E.steps[st_.ub].gn_w >= 0))) {
                    E.steps[st_.ua].fused_stem = true;
                    if (st_.ub >= 0) E.steps[st_.ub].fused_stem = true;
                }
        // ---- statistics finalize folded into the elementwise consumer (not for the fused input block / one-launch small tensors)
        for (auto& st_ : E.steps) st_.fold_fin = false;
        if (E.use_fold)
            for (auto& st_ : E.steps) {
                if (st_.type != ST_ACT || E.steps[st_.ua].fused_stem || E.steps[st_.ua].gn_w < 0) continue;
                const Step& ua_ = E.steps[st_.ua];
                if (st_.ub < 0 && gn_bwd_group_eligible(ua_.Cout, E.vol(E.tens[ua_.raw].lvl), (int)E.esz())) continue;
                if (ua_.Cout > 256) continue;
                E.steps[st_.ua].fold_fin = true;
                if (st_.ub >= 0) E.steps[st_.ub].fold_fin = true;
            }
        // ---- small persistent regions
        E.off_step = alloc(256);
        E.off_masks = alloc((size_t)E.drop_ch.size() * N * E.ld_mask() * 4);
        // forward tensors
        for (auto& t : E.tens) t.off = alloc(ten_bytes(t));
        // statistics (fp64) contiguous so one memset clears them; same for Q
        const size_t s0 = cur;
        for (auto& s : E.steps)
            if (s.type == ST_UNIT && s.gn_w >= 0) s.stats = alloc((size_t)STAT_REP * N * s.Cout * 2 * 8);
        E.off_stats = s0; E.stats_bytes = cur - s0;
        const size_t q0 = cur;
        for (auto& s : E.steps)
            if (s.type == ST_UNIT && s.gn_w >= 0) s.Q = alloc((size_t)STAT_REP * N * s.Cout * 2 * 8);
        E.off_Q = q0; E.Q_bytes = cur - q0;
        for (auto& s : E.steps)
            if (s.type == ST_UNIT && s.gn_w >= 0) {
                s.scale = alloc((size_t)N * s.Cout * 4);
                s.shift = alloc((size_t)N * s.Cout * 4);
                s.mean = alloc((size_t)N * GN_GROUPS * 4);
                s.rstd = alloc((size_t)N * GN_GROUPS * 4);
                s.coef = alloc((size_t)N * s.Cout * 3 * 4);
            }
        // ---- packed weights
        for (auto& s : E.steps) {
            if (s.type != ST_UNIT) continue;
            const int T = ntaps(s.ck), Ci = s.Cin, Co = s.Cout;
            const long long woff = E.params[s.w].off;
            const int C0 = E.tens[s.in0].C, C1 = s.in1 >= 0 ? E.tens[s.in1].C : 0;
            switch (s.ck) {
                case CK_K3: case CK_K1: case CK_K2S2:
                    s.x_fwd = s.x_dg0 = s.x_dg1 = -1;
                    if (s.ck == CK_K3 && E.use_conv3x) {
                        // register-blocked halo kernel (conv3x.hip) wherever the shape allows: fragment-major weights
                        const int l = E.tens[s.raw].lvl, d_ = E.dim_d(l), h_ = E.dim_h(l), w_ = E.dim_w(l);
                        if (conv3x_supported(dt, E.ndim, N, d_, h_, w_, Ci, Co, C0, C1 > 0)) s.x_fwd = conv3x_pick(E.ndim, N, d_, h_, w_, Ci, Co, C1 > 0);
                        if (!E.tens[s.in0].image && conv3x_supported(dt, E.ndim, N, d_, h_, w_, Co, C0, 0, false))
                            s.x_dg0 = conv3x_pick(E.ndim, N, d_, h_, w_, Co, C0);
                        if (C1 && conv3x_supported(dt, E.ndim, N, d_, h_, w_, Co, C1, 0, false)) s.x_dg1 = conv3x_pick(E.ndim, N, d_, h_, w_, Co, C1);
                    }
                    s.wp_fwd = alloc_pack(Co, T * Ci);
                    add_pack(s.wp_fwd, woff, Co, 1, T, Ci, (long long)(s.cin_par ? s.cin_par : Ci) * T, 0, 1, T, 0, s.x_fwd >= 0 ? (Ci == 16 ? 2 : 1) : 0,
                             s.cin_par);                         // image convs on a zero-padded image tensor: the parameter has cin_par channels
                    if (s.ck == CK_K2S2) {       // data-gradient = scatter GEMM, rows (a, ci), K = Cout
                        s.wp_dg0 = alloc_pack(T * Ci, Co);
                        pack_bwd = true;
                        add_pack(s.wp_dg0, woff, T, Ci, 1, Co, 1, T, 0, (long long)Ci * T, 0);
                        pack_bwd = false;
                    } else {                     // data-gradient = gather conv with flipped taps, rows ci, k = (tap, co)
                        if (!E.tens[s.in0].image) {
                            s.wp_dg0 = alloc_pack(C0, T * Co);
                            pack_bwd = true;
                            add_pack(s.wp_dg0, woff, C0, 1, T, Co, T, 0, 1, (long long)Ci * T, 1, s.x_dg0 >= 0 ? (Co == 16 ? 2 : 1) : 0);
                            pack_bwd = false;
                        }
                        if (C1) {
                            s.wp_dg1 = alloc_pack(C1, T * Co);
                            pack_bwd = true;
                            add_pack(s.wp_dg1, woff + (long long)C0 * T, C1, 1, T, Co, T, 0, 1, (long long)Ci * T, 1, s.x_dg1 >= 0 ? (Co == 16 ? 2 : 1) : 0);
                            pack_bwd = false;
                        }
                    }
                    break;
                case CK_KT:                      // forward = scatter GEMM rows (a, co), K = Cin
                    s.wp_fwd = alloc_pack(T * Co, Ci);
                    add_pack(s.wp_fwd, woff, T, Co, 1, Ci, 1, T, 0, (long long)Co * T, 0);
                    s.wp_dg0 = alloc_pack(Ci, T * Co);   // data-gradient = gather stride 2, rows ci, k = (a, co)
                    pack_bwd = true;
                    add_pack(s.wp_dg0, woff, Ci, 1, T, Co, (long long)Co * T, 0, 1, T, 0);
                    pack_bwd = false;
                    break;
                default:                         // image stems: [Cout][32] with k = tap*Cimg + ci (1^d stem: k = ci)
                    s.wp_fwd = alloc_pack(Co, T * Ci);
                    add_pack(s.wp_fwd, woff, Co, 1, T, Ci, (long long)Ci * T, 0, 1, T, 0);
                    break;
            }
        }
        // ---- virtual activations: an ACT (one branch, no residual) whose output feeds exactly ONE 3^d conv that runs on conv3x is
        // never written: the consumer applies relu(scale * raw + shift) while it stages its halo, and so does the consumer's weight
        // gradient (LUConv chains of networks/VNet3d.py:5-23, the two convs of networks/Unet3d.py:64-86 _block)
        if (E.use_vact && dt != DT_F32 && !(getenv("SEG_WGRAD3X") && atoi(getenv("SEG_WGRAD3X")) != 0))
            for (size_t ai = 0; ai < E.steps.size(); ++ai) {
                Step& act = E.steps[ai];
                if (act.type != ST_ACT || act.ub >= 0 || act.res >= 0) continue;
                const Step& prod = E.steps[act.ua];
                if (prod.fused_stem || prod.gn_w < 0 || prod.Cout > 256) continue;
                int users = 0, cons = -1;
                for (size_t ci = 0; ci < E.steps.size(); ++ci) {
                    const Step& c = E.steps[ci];
                    if (c.type == ST_UNIT && (c.in0 == act.out || c.in1 == act.out)) { ++users; cons = (int)ci; }
                    if (c.type == ST_ACT && c.res == act.out) ++users;
                    if ((c.type == ST_POOL || c.type == ST_HEAD) && c.in == act.out) ++users;
                }
                if (users != 1 || cons < 0) continue;
                Step& c = E.steps[cons];
                if (c.ck != CK_K3 || c.in0 != act.out || c.in1 >= 0 || c.x_fwd < 0 || !conv3x_gn_supported(c.Cin, false)) continue;
                act.vact = true;
                c.vact_prod = act.ua;
                E.steps[act.ua].fold_fin = true;       // no finalize launch either: the consumer folds the statistics itself
            }
        {   // forward layouts first, backward-only layouts behind them: the second range is packed on the weight-gradient stream
            std::vector<PackDesc> fw, bw;
            for (size_t i = 0; i < E.packdescs.size(); ++i) (E.pack_is_bwd[i] ? bw : fw).push_back(E.packdescs[i]);
            E.npack_fwd = (int)fw.size();
            E.packdescs = fw;
            E.packdescs.insert(E.packdescs.end(), bw.begin(), bw.end());
            E.pack_is_bwd.assign(E.packdescs.size(), 0);
            for (size_t i = fw.size(); i < E.packdescs.size(); ++i) E.pack_is_bwd[i] = 1;
        }
        E.off_packdesc = alloc(E.packdescs.size() * sizeof(PackDesc));
        // partial-tile buffer of the halo weight-gradient kernel (largest K3 layer)
        size_t pmax = 0;
        for (auto& s : E.steps)
            if (s.type == ST_UNIT) {
                if (s.ck == CK_K3) {
                    const int l = E.tens[s.raw].lvl;
                    pmax = std::max(pmax, wgrad3_partial_bytes(E.ndim, N, E.dim_d(l), E.dim_h(l), E.dim_w(l), s.Cout, s.Cin));
                } else if (s.ck == CK_STEM3 || s.ck == CK_STEM1) {
                    pmax = std::max(pmax, stem_wgrad_partial_bytes(E.ndim, N, E.dim_d(0), E.dim_h(0), E.dim_w(0), s.Cout));
                } else {
                    char* keep = E.ws; E.ws = nullptr;
                    pmax = std::max(pmax, wgrad_partial_bytes(make_wgrad_args(E, s, -1)));
                    E.ws = keep;
                }
            }
        E.off_partial = alloc(pmax);
        E.off_partial2 = E.n_side > 1 ? alloc(pmax) : E.off_partial;
        E.off_partial_main = alloc(pmax);
        E.off_partial_stemx = alloc(stemx_partial_bytes(E.ndim, N, E.dim_d(0), E.dim_h(0), E.dim_w(0), E.in_ch));
        E.off_partial_stem1 = alloc(stem_wgrad_partial_bytes(E.ndim, N, E.dim_d(0), E.dim_h(0), E.dim_w(0), 16 * ((E.feat + 15) / 16)));

        // ------------------------------------------------------------------ forward schedule
        E.fwd_ops.push_back([this_ = &E](hipStream_t st) {
            seg_engine& E = *this_;
            // the backward sums (Q) sit right behind the forward statistics: ONE fill clears both (a fill is a ~6 us launch on the main
            // stream); a backward pass that does not follow a forward pass directly clears Q itself
            const Ten& x = E.tens[E.image_ten];
            const size_t fill = E.stats_bytes + (E.off_Q == E.off_stats + E.stats_bytes ? E.Q_bytes : 0);
            const int pi = E.prof_begin(st, SEG_K_MISC, (double)fill + (double)E.N * E.vol(0) * (4.0 * E.in_ch + (double)x.C * E.esz()), 0.0);
            (void)hipMemsetAsync(E.ws + E.off_stats, 0, fill, st);
            E.q_clean = E.off_Q == E.off_stats + E.stats_bytes;
            launch_ingest(E.cur_x, E.ws + x.off, E.N, x.C, E.vol(0), E.dtype, st, E.in_ch, E.ride_on ? E.ride_ingest : StepRider{});
            E.prof_end(st, pi);
        });
        for (size_t si = 0; si < E.steps.size(); ++si) {
            Step& s = E.steps[si];
            if (s.type == ST_UNIT) {
                E.fwd_ops.push_back([this_ = &E, si](hipStream_t st) {
                    seg_engine& E = *this_;
                    const Step& s = E.steps[si];
                    if (s.fused_stem) return;              // evaluated by the fused input block of its ACT step
                    const Ten& i0 = E.tens[s.in0];
                    const Ten& ro = E.tens[s.raw];
                    double* stats = s.gn_w >= 0 ? (double*)(E.ws + s.stats) : nullptr;
                    const float* bias = s.b >= 0 ? E.p + E.params[s.b].off : nullptr;
                    if (s.ck == CK_STEM3 || s.ck == CK_STEM1) {
                        const int T = s.ck == CK_STEM3 ? (E.ndim == 3 ? 27 : 9) : 1;
                        const int pi = E.prof_begin(st, SEG_K_STEM, E.tbytes(s.in0) + E.tbytes(s.raw), 2.0 * E.N * E.vol(0) * T * i0.C * s.Cout);
                        launch_stem_fwd(E.ws + i0.off, E.ws + s.wp_fwd, bias, E.ws + ro.off, stats, E.N, E.dim_d(0), E.dim_h(0), E.dim_w(0),
                                        i0.C, s.Cout, s.ck == CK_STEM1, E.ndim, E.dtype, st);
                        E.prof_end(st, pi);
                    } else if (s.ck == CK_K3) {
                        const int l = ro.lvl;
                        const int pi = E.prof_begin(st, conv3_class(E.dim_w(l), s.Cin), E.tbytes(s.in0) + E.tbytes(s.raw),
                                                    2.0 * E.N * E.vol(l) * (E.ndim == 3 ? 27 : 9) * s.Cin * s.Cout);
                        // replicas this producer spreads the statistics over (read back by the folded finalize of the consumers)
                        E.steps[si].stat_rep = (s.x_fwd >= 0 && E.use_fold) ? stat_rep_for(E.vol(l)) : STAT_REP;
                        if (s.x_fwd >= 0 && s.vact_prod >= 0) {
                            const Step& u = E.steps[s.vact_prod];        // the producer: its raw output is this launch's input
                            GnFinArgs f{};
                            f.stats = (double*)(E.ws + u.stats); f.gamma = E.p + E.params[u.gn_w].off; f.beta = E.p + E.params[u.gn_b].off;
                            f.mask = E.mask_mode == SEG_MASKS_EVAL ? nullptr
                                     : E.mask_base(u.mask_slot);
                            f.mask_ld = E.ld_mask();
                            f.scale = (float*)(E.ws + u.scale); f.shift = (float*)(E.ws + u.shift);
                            f.mean = (float*)(E.ws + u.mean); f.rstd = (float*)(E.ws + u.rstd);
                            f.N = E.N; f.C = u.Cout; f.V = E.vol(E.tens[u.raw].lvl); f.eps = 1e-5f; f.rep = u.stat_rep;
                            launch_conv3x(s.x_fwd, E.ws + E.tens[u.raw].off, nullptr, i0.C, E.ws + s.wp_fwd, bias, E.ws + ro.off, stats, E.N,
                                          E.dim_d(l), E.dim_h(l), E.dim_w(l), s.Cin, s.Cout, E.ndim, E.dtype, st, s.stat_rep, &f);
                        } else if (s.x_fwd >= 0)
                            launch_conv3x(s.x_fwd, E.ws + i0.off, s.in1 >= 0 ? E.ws + E.tens[s.in1].off : nullptr, i0.C, E.ws + s.wp_fwd, bias,
                                          E.ws + ro.off, stats, E.N, E.dim_d(l), E.dim_h(l), E.dim_w(l), s.Cin, s.Cout, E.ndim, E.dtype, st,
                                          s.stat_rep);
                        else
                        launch_conv3(E.ws + i0.off, E.ws + s.wp_fwd, bias, E.ws + ro.off, stats, E.N, E.dim_d(l), E.dim_h(l), E.dim_w(l),
                                     s.Cin, s.Cout, E.ndim, E.dtype, st, s.in1 >= 0 ? E.ws + E.tens[s.in1].off : nullptr, i0.C);
                        E.prof_end(st, pi);
                    } else {
                        ConvArgs a{};
                        a.in0 = E.ws + i0.off; a.C0 = i0.C;
                        a.in1 = s.in1 >= 0 ? E.ws + E.tens[s.in1].off : nullptr;
                        a.C1 = s.in1 >= 0 ? E.tens[s.in1].C : 0;
                        a.w = E.ws + s.wp_fwd; a.bias = bias; a.out = E.ws + ro.off; a.stats = stats;
                        a.N = E.N; a.Cout = s.Cout;
                        const int li = i0.lvl, lo = ro.lvl;
                        a.ID = E.dim_d(li); a.IH = E.dim_h(li); a.IW = E.dim_w(li);
                        if (s.ck == CK_KT) {
                            a.scatter = 1;
                            a.OD = a.ID; a.OH = a.IH; a.OW = a.IW;
                            a.FD = E.dim_d(lo); a.FH = E.dim_h(lo); a.FW = E.dim_w(lo);
                            a.sd = E.ndim == 3 ? 2 : 1; a.sh = 2; a.sw = 2;
                            a.taps = make_taps(E.ndim, 2, 0);
                            a.K = s.Cin; a.Ngemm = a.taps.n * s.Cout;
                        } else {
                            a.scatter = 0;
                            a.OD = E.dim_d(lo); a.OH = E.dim_h(lo); a.OW = E.dim_w(lo);
                            const int k = s.ck == CK_K3 ? 3 : s.ck == CK_K2S2 ? 2 : 1;
                            a.taps = make_taps(E.ndim, k, s.ck == CK_K3 ? 1 : 0);
                            const int str = s.ck == CK_K2S2 ? 2 : 1;
                            a.sd = E.ndim == 3 ? str : 1; a.sh = str; a.sw = str;
                            a.K = a.taps.n * s.Cin; a.Ngemm = s.Cout;
                        }
                        a.Kpad = (a.K + 31) / 32 * 32;
                        const int pi = E.prof_begin(st, SEG_K_CONV_GENERIC,
                                                    E.tbytes(s.in0) + (s.in1 >= 0 ? E.tbytes(s.in1) : 0.0) + E.tbytes(s.raw),
                                                    2.0 * E.N * E.vol(s.ck == CK_KT ? li : lo) * (double)a.K * a.Ngemm);
                        E.steps[si].stat_rep = (E.use_fold && !conv_uses_stream_kernel(a)) ? stat_rep_for(E.vol(lo)) : STAT_REP;
                        launch_conv_igemm(a, E.dtype, st, s.stat_rep);
                        E.prof_end(st, pi);
                    }
                    if (s.gn_w >= 0 && !s.fold_fin && !gn_bwd_group_eligible(s.Cout, E.vol(ro.lvl), (int)E.esz())) {
                        GnFinArgs f{};
                        f.stats = stats; f.gamma = E.p + E.params[s.gn_w].off; f.beta = E.p + E.params[s.gn_b].off;
                        f.mask = E.mask_mode == SEG_MASKS_EVAL ? nullptr
                                 : E.mask_base(s.mask_slot);
                        f.mask_ld = E.ld_mask();
                        f.scale = (float*)(E.ws + s.scale); f.shift = (float*)(E.ws + s.shift);
                        f.mean = (float*)(E.ws + s.mean); f.rstd = (float*)(E.ws + s.rstd);
                        f.N = E.N; f.C = s.Cout; f.V = E.vol(ro.lvl); f.eps = 1e-5f;
                        launch_gn_finalize(f, st);
                    }
                });
            } else if (s.type == ST_ACT) {
                E.fwd_ops.push_back([this_ = &E, si](hipStream_t st) {
                    seg_engine& E = *this_;
                    const Step& s = E.steps[si];
                    const Step& ua = E.steps[s.ua];
                    if (s.vact) return;                    // applied by the consuming conv while it stages its halo (Step::vact_prod)
                    if (ua.fused_stem) {
                        // fused input block: statistics of both branches from the image, finalize, then recompute + normalise + add
                        seg_stemx_args x = stemx_args(E, s);
                        const int pi = E.prof_begin(st, SEG_K_STEM, E.tbytes(ua.in0) * 2 + E.tbytes(s.out), 0.0);
                        launch_stemx(x, 0, E.ndim, E.dtype, nullptr, nullptr, st);
                        GnFinArgs fin[2];
                        int nfin = 0;
                        for (int ui : {s.ua, s.ub}) {
                            if (ui < 0) continue;
                            const Step& u = E.steps[ui];
                            GnFinArgs f{};
                            f.stats = (double*)(E.ws + u.stats); f.gamma = E.p + E.params[u.gn_w].off; f.beta = E.p + E.params[u.gn_b].off;
                            f.mask = E.mask_mode == SEG_MASKS_EVAL ? nullptr
                                     : E.mask_base(u.mask_slot);
                            f.mask_ld = E.ld_mask();
                            f.scale = (float*)(E.ws + u.scale); f.shift = (float*)(E.ws + u.shift);
                            f.mean = (float*)(E.ws + u.mean); f.rstd = (float*)(E.ws + u.rstd);
                            f.N = E.N; f.C = u.Cout; f.V = E.vol(0); f.eps = 1e-5f;
                            fin[nfin++] = f;
                        }
                        launch_gn_finalize(fin[0], st, nfin > 1 ? &fin[1] : nullptr);      // both branches: one launch
                        launch_stemx(x, 1, E.ndim, E.dtype, nullptr, nullptr, st);
                        E.prof_end(st, pi);
                        return;
                    }
                    {
                        const Ten& ro = E.tens[ua.raw];
                        if (s.ub < 0 && gn_bwd_group_eligible(ua.Cout, E.vol(ro.lvl), (int)E.esz())) {
                            // small L2-resident tensor: statistics finalize + activation in one launch
                            GnFinArgs f{};
                            f.stats = (double*)(E.ws + ua.stats);
                            f.gamma = E.p + E.params[ua.gn_w].off; f.beta = E.p + E.params[ua.gn_b].off;
                            f.mask = E.mask_mode == SEG_MASKS_EVAL ? nullptr
                                     : E.mask_base(ua.mask_slot);
                            f.mask_ld = E.ld_mask();
                            f.scale = (float*)(E.ws + ua.scale); f.shift = (float*)(E.ws + ua.shift);
                            f.mean = (float*)(E.ws + ua.mean); f.rstd = (float*)(E.ws + ua.rstd);
                            f.N = E.N; f.C = ua.Cout; f.V = E.vol(ro.lvl); f.eps = 1e-5f; f.rep = ua.stat_rep;
                            const int pi = E.prof_begin(st, SEG_K_GN_GROUP, E.tbytes(s.out) * (2 + (s.res >= 0)), 0.0);
                            launch_gn_fwd_group(f, E.ws + ro.off, s.res >= 0 ? E.ws + E.tens[s.res].off : nullptr, E.ws + E.tens[s.out].off,
                                                E.dtype, st);
                            E.prof_end(st, pi);
                            return;
                        }
                    }
                    ActArgs a{};
                    a.r1 = E.ws + E.tens[ua.raw].off; a.scale1 = (float*)(E.ws + ua.scale); a.shift1 = (float*)(E.ws + ua.shift);
                    if (s.ub >= 0) {
                        const Step& ub = E.steps[s.ub];
                        a.r2 = E.ws + E.tens[ub.raw].off; a.scale2 = (float*)(E.ws + ub.scale); a.shift2 = (float*)(E.ws + ub.shift);
                    }
                    a.res = s.res >= 0 ? E.ws + E.tens[s.res].off : nullptr;
                    a.out = E.ws + E.tens[s.out].off;
                    a.N = E.N; a.C = E.tens[s.out].C; a.V = E.vol(E.tens[s.out].lvl);
                    if (ua.fold_fin) {
                        auto fin = [&E](const Step& u, GnFinArgs& f) {
                            f = GnFinArgs{};
                            f.stats = (double*)(E.ws + u.stats); f.gamma = E.p + E.params[u.gn_w].off; f.beta = E.p + E.params[u.gn_b].off;
                            f.mask = E.mask_mode == SEG_MASKS_EVAL ? nullptr
                                     : E.mask_base(u.mask_slot);
                            f.mask_ld = E.ld_mask();
                            f.scale = (float*)(E.ws + u.scale); f.shift = (float*)(E.ws + u.shift);
                            f.mean = (float*)(E.ws + u.mean); f.rstd = (float*)(E.ws + u.rstd);
                            f.N = E.N; f.C = u.Cout; f.V = E.vol(E.tens[u.raw].lvl); f.eps = 1e-5f; f.rep = u.stat_rep;
                        };
                        a.fold = 1;
                        fin(ua, a.fin1);
                        if (s.ub >= 0) fin(E.steps[s.ub], a.fin2);
                    }
                    const int pi = E.prof_begin(st, SEG_K_GN_ACT, E.tbytes(s.out) * (2 + (s.ub >= 0) + (s.res >= 0)), 0.0);
                    launch_gn_act(a, E.dtype, st);
                    E.prof_end(st, pi);
                });
            } else if (s.type == ST_POOL) {
                E.fwd_ops.push_back([this_ = &E, si](hipStream_t st) {
                    seg_engine& E = *this_;
                    const Step& s = E.steps[si];
                    const Ten& ti = E.tens[s.in];
                    PoolArgs a{};
                    a.in = E.ws + ti.off; a.out = E.ws + E.tens[s.out].off;
                    a.N = E.N; a.D = E.dim_d(ti.lvl); a.H = E.dim_h(ti.lvl); a.W = E.dim_w(ti.lvl); a.C = ti.C;
                    a.pd = E.ndim == 3 ? 2 : 1; a.ph = 2; a.pw = 2;
                    launch_maxpool_fwd(a, E.dtype, st);
                });
            } else {   // HEAD
                E.fwd_ops.push_back([this_ = &E, si](hipStream_t st) {
                    seg_engine& E = *this_;
                    const Step& s = E.steps[si];
                    HeadArgs a;
                    a.in = E.ws + E.tens[s.in].off; a.w = E.p + E.params[s.w].off; a.bias = E.p + E.params[s.b].off;
                    a.logits = E.cur_logits; a.probs = E.cur_probs;
                    a.N = E.N; a.V = (int)E.vol(0); a.Cin = s.Cin; a.C = s.Cout;
                    if (E.ride_on && E.ride_zero) { a.zero_ptr = E.ride_zero; a.zero_n = E.ride_zero_n; E.head_zeroed = true; }
                    const int pi = E.prof_begin(st, SEG_K_HEAD, E.tbytes(s.in) + 2.0 * 4.0 * E.N * E.vol(0) * s.Cout, 0.0);
                    launch_head_fwd(a, E.dtype, st);
                    E.prof_end(st, pi);
                });
            }
        }

        // a unit whose statistics live in the per-group layout while chains run (seg_engine::run_chain): finest level(s), real kernels
        auto unit_sub = [&E](int ui) {
            if (ui < 0) return false;
            const Step& u = E.steps[ui];
            if (u.type != ST_UNIT || E.tens[u.raw].lvl > E.sub_lvl || u.vact_prod >= 0) return false;
            return (u.ck != CK_STEM3 && u.ck != CK_STEM1) || u.fused_stem;
        };
        // ------------------------------------------------------------------ backward schedule
        E.bwd_writes.push_back({});
        E.bwd_sub.push_back(0);
        E.bwd_sig.push_back(0);
        E.bwd_ops.push_back([this_ = &E](hipStream_t st) {
            seg_engine& E = *this_;
            if (!E.q_clean) (void)hipMemsetAsync(E.ws + E.off_Q, 0, E.Q_bytes, st);
            E.q_clean = false;
        });
        for (int si = (int)E.steps.size() - 1; si >= 0; --si) {
            Step& s = E.steps[si];
            if (s.type == ST_HEAD) {
                const int gin = new_grad(s.in);
                E.tens[gin].virt = E.use_vhead;
                E.head_din_needed = !E.use_vhead;
                E.head_step = si;
                E.tens[s.in].grads.push_back(gin);
                E.bwd_writes.push_back({s.w, s.b});
                E.bwd_sub.push_back(1);
                E.bwd_sig.push_back(0);
                E.bwd_ops.push_back([this_ = &E, si, gin](hipStream_t st) {
                    seg_engine& E = *this_;
                    const Step& s = E.steps[si];
                    HeadBwdArgs a;
                    a.in = E.ws + E.tens[s.in].off; a.w = E.p + E.params[s.w].off; a.dlogits = E.cur_dlogits;
                    // rank-K gradient: its readers (GroupNorm-backward passes) rebuild it from dlogits unless one of them cannot
                    a.din = E.head_din_needed ? E.ws + E.tens[gin].off : nullptr;
                    a.dw = E.g + E.params[s.w].off; a.db = E.g + E.params[s.b].off;
                    a.N = E.N; a.V = (int)E.vol(0); a.Cin = s.Cin; a.C = s.Cout;
                    const int pi = E.prof_begin(st, SEG_K_HEAD, E.tbytes(s.in) * (a.din ? 2.0 : 1.0) + 4.0 * E.N * E.vol(0) * s.Cout, 0.0);
                    launch_head_bwd(a, E.dtype, st);
                    E.prof_end(st, pi);
                });
            } else if (s.type == ST_POOL) {
                std::vector<int> gl = E.tens[s.out].grads;
                if (gl.size() != 1) { g_err = "internal: pool output needs exactly one gradient"; return; }
                if (E.tens[gl[0]].virt) E.head_din_needed = true;
                const int gin = new_grad(s.in);
                E.tens[s.in].grads.push_back(gin);
                const int gout = gl[0];
                E.bwd_writes.push_back({});
                E.bwd_sub.push_back(E.tens[s.in].lvl <= E.sub_lvl ? 1 : 0);
                E.bwd_sig.push_back(0);
                E.bwd_ops.push_back([this_ = &E, si, gin, gout](hipStream_t st) {
                    seg_engine& E = *this_;
                    const Step& s = E.steps[si];
                    const Ten& ti = E.tens[s.in];
                    PoolArgs a{};
                    a.in = E.ws + ti.off; a.dout = E.ws + E.tens[gout].off; a.din = E.ws + E.tens[gin].off;
                    a.N = E.N; a.D = E.dim_d(ti.lvl); a.H = E.dim_h(ti.lvl); a.W = E.dim_w(ti.lvl); a.C = ti.C;
                    a.pd = E.ndim == 3 ? 2 : 1; a.ph = 2; a.pw = 2;
                    launch_maxpool_bwd(a, E.dtype, st);
                });
            } else if (s.type == ST_ACT) {
                std::vector<int> gl = E.tens[s.out].grads;
                if (gl.empty() || gl.size() > 3) { g_err = "internal: unsupported gradient fan-in"; return; }
                if (s.res >= 0) for (int gi : gl) E.tens[s.res].grads.push_back(gi);
                {
                    // the fused input block, the dual-branch and the one-launch small-tensor passes read real tensors only
                    const Step& ua_ = E.steps[s.ua];
                    const bool generic = !ua_.fused_stem && s.ub < 0 &&
                                         !gn_bwd_group_eligible(E.tens[ua_.raw].C, E.vol(E.tens[ua_.raw].lvl), (int)E.esz());
                    for (int gi : gl) if (E.tens[gi].virt && !generic) E.head_din_needed = true;
                }
                // per-branch argument builders (shared by the single- and the dual-branch op)
                auto fill = [](seg_engine& E, int ui, const std::vector<int>& gl, GnBwdArgs& a, GnBwdFinArgs& f) {
                    const Step& u = E.steps[ui];
                    const Ten& r = E.tens[u.raw];
                    a = GnBwdArgs{};
                    a.ndy = 0;
                    for (int gi : gl) {
                        if (E.tens[gi].virt && !E.head_din_needed) {
                            const Step& hs = E.steps[E.head_step];
                            a.vdl = E.cur_dlogits; a.vw = E.p + E.params[hs.w].off; a.vK = hs.Cout;
                        } else a.dy[a.ndy++] = E.ws + E.tens[gi].off;
                    }
                    a.r = E.ws + r.off;
                    a.scale = (float*)(E.ws + u.scale); a.shift = (float*)(E.ws + u.shift);
                    a.Q = (double*)(E.ws + u.Q); a.coef = (float*)(E.ws + u.coef);
                    a.dr = E.ws + E.tens[u.draw].off;
                    a.N = E.N; a.C = r.C; a.V = E.vol(r.lvl);
                    f = GnBwdFinArgs{};
                    f.Q = a.Q; f.stats = (double*)(E.ws + u.stats);
                    f.gamma = E.p + E.params[u.gn_w].off;
                    f.mask = E.mask_mode == SEG_MASKS_EVAL ? nullptr
                             : E.mask_base(u.mask_slot);
                    f.mask_ld = E.ld_mask();
                    f.mean = (float*)(E.ws + u.mean); f.rstd = (float*)(E.ws + u.rstd);
                    f.dgamma = E.g + E.params[u.gn_w].off; f.dbeta = E.g + E.params[u.gn_b].off;
                    f.dbias = u.b >= 0 ? E.g + E.params[u.b].off : nullptr;
                    f.coef = (float*)(E.ws + u.coef);
                    f.N = E.N; f.C = r.C; f.V = a.V;
                    a.rep_q = f.rep_q = E.use_fold ? stat_rep_for(a.V) : 0;
                    f.rep_s = u.stat_rep;
                };
                if (E.steps[s.ua].fused_stem) {
                    // fused input block: reduce (recomputing r), finalize per branch, then d(raw) in registers -> stem weight gradients
                    std::vector<int> wr;
                    for (int ui : {s.ua, s.ub})
                        if (ui >= 0) { const Step& u = E.steps[ui]; wr.push_back(u.gn_w); wr.push_back(u.gn_b); wr.push_back(u.b); wr.push_back(u.w); }
                    E.bwd_writes.push_back(wr);
                    E.bwd_sub.push_back((unit_sub(s.ua) && (s.ub < 0 || unit_sub(s.ub))) ? 1 : 0);
                    E.bwd_sig.push_back(0);
                    E.bwd_ops.push_back([this_ = &E, si, gl](hipStream_t st) {
                        seg_engine& E = *this_;
                        const Step& s = E.steps[si];
                        seg_stemx_args x = stemx_args(E, s);
                        x.ndy = (int)gl.size();
                        for (int i = 0; i < x.ndy; ++i) x.dy[i] = E.ws + E.tens[gl[i]].off;
                        E.flush_side(st);
                        const double tb = E.tbytes(s.out);
                        int pi = E.prof_begin(st, SEG_K_STEM, tb * x.ndy, 0.0);
                        launch_stemx(x, 2, E.ndim, E.dtype, nullptr, nullptr, st);
                        E.prof_end(st, pi);
                        GnBwdFinArgs fin[2];
                        int nfin = 0;
                        for (int ui : {s.ua, s.ub}) {
                            if (ui < 0) continue;
                            const Step& u = E.steps[ui];
                            GnBwdFinArgs f{};
                            f.Q = (double*)(E.ws + u.Q); f.stats = (double*)(E.ws + u.stats);
                            f.gamma = E.p + E.params[u.gn_w].off;
                            f.mask = E.mask_mode == SEG_MASKS_EVAL ? nullptr
                                     : E.mask_base(u.mask_slot);
                            f.mask_ld = E.ld_mask();
                            f.mean = (float*)(E.ws + u.mean); f.rstd = (float*)(E.ws + u.rstd);
                            f.dgamma = E.g + E.params[u.gn_w].off; f.dbeta = E.g + E.params[u.gn_b].off;
                            f.dbias = u.b >= 0 ? E.g + E.params[u.b].off : nullptr;
                            f.coef = (float*)(E.ws + u.coef);
                            f.N = E.N; f.C = u.Cout; f.V = E.vol(0);
                            fin[nfin++] = f;
                        }
                        launch_gn_bwd_finalize(fin[0], st, nfin > 1 ? &fin[1] : nullptr);  // both branches: one launch
                        pi = E.prof_begin(st, SEG_K_STEM, tb * x.ndy, 0.0);
                        launch_stemx(x, 3, E.ndim, E.dtype, E.g + E.params[E.steps[s.ua].w].off,
                                     s.ub >= 0 ? E.g + E.params[E.steps[s.ub].w].off : nullptr, st);
                        E.prof_end(st, pi);
                    });
                    continue;
                }
                const bool dual = s.ua >= 0 && s.ub >= 0 && E.dual_gn_bwd &&
                                  !gn_bwd_group_eligible(E.tens[E.steps[s.ua].raw].C, E.vol(E.tens[E.steps[s.ua].raw].lvl), (int)E.esz());
                if (dual) {
                    // both branches of the VNet input block (one GroupNorm module applied twice, networks/VNet3d.py:36-41) receive
                    // the SAME gradient sources: one reduce and one apply pass read them once for both (14 -> 10 tensor passes)
                    Step& ua = E.steps[s.ua];
                    Step& ub = E.steps[s.ub];
                    ua.draw = new_grad(ua.raw);
                    ub.draw = new_grad(ub.raw);
                    E.bwd_writes.push_back({ua.gn_w, ua.gn_b, ua.b, ub.gn_w, ub.gn_b, ub.b});
                    E.bwd_sub.push_back((unit_sub(s.ua) && unit_sub(s.ub)) ? 1 : 0);
                    E.bwd_sig.push_back(1);
                    E.bwd_ops.push_back([this_ = &E, uia = s.ua, uib = s.ub, gl, fill](hipStream_t st) {
                        seg_engine& E = *this_;
                        GnBwdArgs a, b;
                        GnBwdFinArgs fa{}, fb{};
                        fill(E, uia, gl, a, fa);
                        fill(E, uib, gl, b, fb);
                        a.sig_flag = E.take_sig(a.sig_seq);        // the reduce pass is the first kernel behind a released batch of weight gradients
                        a.r2 = b.r; a.scale2 = b.scale; a.shift2 = b.shift; a.Q2 = b.Q; a.coef2 = b.coef; a.dr2 = b.dr;
                        const double tb = E.tbytes(E.steps[uia].raw);
                        int pi = E.prof_begin(st, SEG_K_GN_BWD_REDUCE, tb * (a.ndy + 2), 0.0);
                        launch_gn_bwd_reduce(a, E.dtype, st);
                        E.prof_end(st, pi);
                        const bool fold = E.use_fold && a.C <= 256;
                        if (!fold) { launch_gn_bwd_finalize(fa, st); launch_gn_bwd_finalize(fb, st); }
                        pi = E.prof_begin(st, SEG_K_GN_BWD_APPLY, tb * (a.ndy + 4), 0.0);
                        launch_gn_bwd_apply(a, E.dtype, st, fold ? &fa : nullptr, fold ? &fb : nullptr);
                        E.prof_end(st, pi);
                    });
                } else
                for (int ui : {s.ua, s.ub}) {
                    if (ui < 0) continue;
                    Step& u = E.steps[ui];
                    u.draw = new_grad(u.raw);
                    E.bwd_writes.push_back({u.gn_w, u.gn_b, u.b});      // gamma/beta and (analytically) the conv bias
                    E.bwd_sub.push_back(unit_sub(ui) ? 1 : 0);
                    E.bwd_sig.push_back(1);
                    E.bwd_ops.push_back([this_ = &E, ui, gl, fill](hipStream_t st) {
                        seg_engine& E = *this_;
                        const Step& u = E.steps[ui];
                        const Ten& r = E.tens[u.raw];
                        GnBwdArgs a;
                        GnBwdFinArgs f{};
                        fill(E, ui, gl, a, f);
                        a.sig_flag = E.take_sig(a.sig_seq);        // (see the dual-branch op)
                        if (gn_bwd_group_eligible(r.C, a.V, (int)E.esz())) {
                            const int pg = E.prof_begin(st, SEG_K_GN_GROUP, E.tbytes(u.raw) * (2 * a.ndy + 3), 0.0);
                            launch_gn_bwd_group(a, f, E.dtype, st);
                            E.prof_end(st, pg);
                            return;
                        }
                        int pi = E.prof_begin(st, SEG_K_GN_BWD_REDUCE, E.tbytes(u.raw) * (a.ndy + 1), 0.0);
                        launch_gn_bwd_reduce(a, E.dtype, st);
                        E.prof_end(st, pi);
                        const bool fold = E.use_fold && a.C <= 256;
                        if (!fold) launch_gn_bwd_finalize(f, st);
                        pi = E.prof_begin(st, SEG_K_GN_BWD_APPLY, E.tbytes(u.raw) * (a.ndy + 2), 0.0);
                        launch_gn_bwd_apply(a, E.dtype, st, fold ? &f : nullptr, nullptr);
                        E.prof_end(st, pi);
                    });
                }
            } else {   // UNIT: weight gradient + data gradient given d(raw)
                if (s.fused_stem) continue;      // weight gradients come out of the fused input block (the ACT op above)
                int draw = s.draw;
                if (s.gn_w < 0) {
                    // plain ConvTranspose (UNet up-conv): d(raw) is the (single) gradient of its output tensor
                    std::vector<int> gl = E.tens[s.raw].grads;
                    if (gl.size() != 1) { g_err = "internal: plain conv output needs exactly one gradient"; return; }
                    draw = gl[0];
                    if (E.tens[draw].virt) E.head_din_needed = true;
                }
                if (draw < 0) { g_err = "internal: unit without output gradient"; return; }
                const bool need_dg0 = !E.tens[s.in0].image;
                int g0 = -1, g1 = -1;
                if (need_dg0) { g0 = new_grad(s.in0); E.tens[s.in0].grads.push_back(g0); }
                if (s.in1 >= 0) { g1 = new_grad(s.in1); E.tens[s.in1].grads.push_back(g1); }
                E.bwd_writes.push_back({s.w, s.gn_w < 0 ? s.b : -1});
                E.bwd_sub.push_back(unit_sub((int)si) ? 1 : 0);
                E.bwd_sig.push_back(0);
                if (s.ck != CK_STEM3 && s.ck != CK_STEM1) ++E.n_deferred;
                E.bwd_ops.push_back([this_ = &E, si, draw, g0, g1](hipStream_t st) {
                    seg_engine& E = *this_;
                    const Step& s = E.steps[si];
                    const Ten& i0 = E.tens[s.in0];
                    const Ten& ro = E.tens[s.raw];
                    const int li = i0.lvl, lo = ro.lvl;
                    const int T = (s.ck == CK_K3 || s.ck == CK_STEM3) ? (E.ndim == 3 ? 27 : 9)
                                  : (s.ck == CK_K2S2 || s.ck == CK_KT) ? (E.ndim == 3 ? 8 : 4) : 1;
                    // ---- bias gradient of convs without GroupNorm
                    if (s.gn_w < 0 && s.b >= 0)
                        launch_colsum(E.ws + E.tens[draw].off, E.g + E.params[s.b].off, (long long)E.N * E.vol(lo), s.Cout, E.dtype, st);
                    if (s.ck == CK_K3) {
                        // halo-tile kernels: weight gradient (deterministic two-stage reduction) + data gradient(s)
                        const double fl = 2.0 * E.N * E.vol(lo) * (E.ndim == 3 ? 27 : 9) * s.Cin * s.Cout;
                        E.defer_wgrad(st, [this_, si, draw, fl, lo](hipStream_t ws_) {
                            seg_engine& E = *this_;
                            const Step& s = E.steps[si];
                            const Ten& i0 = E.tens[s.in0];
                            const int pi = E.prof_begin(ws_, SEG_K_WGRAD3, E.tbytes(draw) + E.tbytes(s.in0) + (s.in1 >= 0 ? E.tbytes(s.in1) : 0.0), fl);
                            if (s.vact_prod >= 0) {     // the input tensor was never written: raw producer output + its published scale / shift
                                const Step& u = E.steps[s.vact_prod];
                                launch_wgrad3(E.ws + E.tens[draw].off, E.ws + E.tens[u.raw].off, (float*)(E.ws + E.cur_partial),
                                              E.g + E.params[s.w].off, E.N, E.dim_d(lo), E.dim_h(lo), E.dim_w(lo), s.Cout, s.Cin, E.ndim, E.dtype, ws_,
                                              nullptr, i0.C, (const float*)(E.ws + u.scale), (const float*)(E.ws + u.shift));
                            } else
                            launch_wgrad3(E.ws + E.tens[draw].off, E.ws + i0.off, (float*)(E.ws + E.cur_partial), E.g + E.params[s.w].off,
                                          E.N, E.dim_d(lo), E.dim_h(lo), E.dim_w(lo), s.Cout, s.Cin, E.ndim, E.dtype, ws_,
                                          s.in1 >= 0 ? E.ws + E.tens[s.in1].off : nullptr, i0.C, nullptr, nullptr, s.cin_par);
                            E.prof_end(ws_, pi);
                        }, E.tbytes(draw), lo, s.gn_w >= 0 ? si : -1, g0 >= 0 ? s.x_dg0 >= 0 : (g1 >= 0 && s.x_dg1 >= 0));
                        int pi;
                        ForkSig sg;                               // a batch released just now: the first data-gradient kernel stores its number
                        if (g0 >= 0) {
                            pi = E.prof_begin(st, conv3_class(E.dim_w(lo), s.Cout), E.tbytes(draw) + E.tbytes(g0), fl * i0.C / s.Cin);
                            if (s.x_dg0 >= 0) {
                                sg.flag = E.take_sig(sg.seq);
                                launch_conv3x(s.x_dg0, E.ws + E.tens[draw].off, nullptr, 0, E.ws + s.wp_dg0, nullptr, E.ws + E.tens[g0].off, nullptr,
                                              E.N, E.dim_d(lo), E.dim_h(lo), E.dim_w(lo), s.Cout, i0.C, E.ndim, E.dtype, st, STAT_REP, nullptr, sg);
                            } else
                            launch_conv3(E.ws + E.tens[draw].off, E.ws + s.wp_dg0, nullptr, E.ws + E.tens[g0].off, nullptr, E.N,
                                         E.dim_d(lo), E.dim_h(lo), E.dim_w(lo), s.Cout, i0.C, E.ndim, E.dtype, st);
                            E.prof_end(st, pi);
                        }
                        if (g1 >= 0) {
                            const int C1 = E.tens[s.in1].C;
                            pi = E.prof_begin(st, conv3_class(E.dim_w(lo), s.Cout), E.tbytes(draw) + E.tbytes(g1), fl * C1 / s.Cin);
                            if (s.x_dg1 >= 0) {
                                sg = ForkSig{};
                                sg.flag = E.take_sig(sg.seq);     // (null when the first data-gradient took it)
                                launch_conv3x(s.x_dg1, E.ws + E.tens[draw].off, nullptr, 0, E.ws + s.wp_dg1, nullptr, E.ws + E.tens[g1].off, nullptr,
                                              E.N, E.dim_d(lo), E.dim_h(lo), E.dim_w(lo), s.Cout, C1, E.ndim, E.dtype, st, STAT_REP, nullptr, sg);
                            } else
                            launch_conv3(E.ws + E.tens[draw].off, E.ws + s.wp_dg1, nullptr, E.ws + E.tens[g1].off, nullptr, E.N,
                                         E.dim_d(lo), E.dim_h(lo), E.dim_w(lo), s.Cout, C1, E.ndim, E.dtype, st);
                            E.prof_end(st, pi);
                        }
                        return;
                    }
                    if (s.ck == CK_STEM3 || s.ck == CK_STEM1) {
                        // the image stems close the backward pass: nothing is left on the main stream to overlap with, so the
                        // 1^d stem (own scratch) runs on the main stream next to the 3^d stem on the side stream
                        auto run = [this_, si, draw](hipStream_t ws_) {
                            seg_engine& E = *this_;
                            const Step& s = E.steps[si];
                            const Ten& i0 = E.tens[s.in0];
                            // both stems use the stem scratch when they run on the main stream (in order there); the shared
                            // partial buffer belongs to whatever the side stream is still reducing
                            const size_t scratch = (s.ck == CK_STEM1 || E.stem_on_main) ? E.off_partial_stem1 : E.cur_partial;
                            const int pi = E.prof_begin(ws_, SEG_K_STEM, E.tbytes(draw) + E.tbytes(s.in0), 0.0);
                            launch_stem_wgrad(E.ws + E.tens[draw].off, E.ws + i0.off, (float*)(E.ws + scratch), E.g + E.params[s.w].off,
                                              E.N, E.dim_d(0), E.dim_h(0), E.dim_w(0), i0.C, s.Cout, s.ck == CK_STEM1, E.ndim, E.dtype, ws_);
                            E.prof_end(ws_, pi);
                        };
                        // step-24 trace: with the 3^d stem on the side stream the main stream idled 256 us at the end of every step
                        // behind wgrad3(16ch@96^3) + the 1^d concat wgrad + this kernel; both stems now run on the main stream
                        if (s.ck == CK_STEM1 || E.stem_on_main) { E.flush_side(st); run(st); }
                        else { E.defer_wgrad(st, run); E.flush_side(st); }
                        return;
                    }
                    // ---- weight gradient
                    E.defer_wgrad(st, [this_, si, draw](hipStream_t ws_) {
                        seg_engine& E = *this_;
                        const Step& s = E.steps[si];
                        WgradArgs w = make_wgrad_args(E, s, draw);
                        const int pi = E.prof_begin(ws_, SEG_K_WGRAD_GENERIC,
                                                    E.tbytes(draw) + E.tbytes(s.in0) + (s.in1 >= 0 ? E.tbytes(s.in1) : 0.0), 0.0);
                        launch_wgrad(w, (float*)(E.ws + E.cur_partial), E.dtype, ws_, s.cin_par);
                        E.prof_end(ws_, pi);
                    }, E.tbytes(draw), lo < li ? lo : li, s.gn_w >= 0 ? si : -1, g0 >= 0 || g1 >= 0);
                    // ---- data gradient(s)
                    if (g0 < 0 && g1 < 0) return;
                    ForkSig sg;                                   // a batch released just now: the first data-gradient kernel stores its number
                    sg.flag = E.take_sig(sg.seq);
                    ConvArgs a{};
                    a.in0 = E.ws + E.tens[draw].off; a.C0 = s.Cout; a.in1 = nullptr; a.C1 = 0;
                    a.bias = nullptr; a.stats = nullptr; a.N = E.N;
                    if (s.ck == CK_K2S2) {
                        // d_in[2o+a][ci] = sum_co draw[o][co] W[co][ci][a] : scatter GEMM over coarse rows
                        a.scatter = 1; a.w = E.ws + s.wp_dg0; a.out = E.ws + E.tens[g0].off;
                        a.ID = a.OD = E.dim_d(lo); a.IH = a.OH = E.dim_h(lo); a.IW = a.OW = E.dim_w(lo);
                        a.FD = E.dim_d(li); a.FH = E.dim_h(li); a.FW = E.dim_w(li);
                        a.sd = E.ndim == 3 ? 2 : 1; a.sh = 2; a.sw = 2;
                        a.taps = make_taps(E.ndim, 2, 0);
                        a.Cout = s.Cin; a.K = s.Cout; a.Ngemm = a.taps.n * s.Cin; a.Kpad = (a.K + 31) / 32 * 32;
                        { launch_conv_igemm(a, E.dtype, st, STAT_REP, sg); sg = ForkSig{}; }
                    } else if (s.ck == CK_KT) {
                        // d_X[i][ci] = sum_{a,co} dY[2i+a][co] Wt[ci][co][a] : gather, stride 2 over the fine gradient
                        a.scatter = 0; a.w = E.ws + s.wp_dg0; a.out = E.ws + E.tens[g0].off;
                        a.ID = E.dim_d(lo); a.IH = E.dim_h(lo); a.IW = E.dim_w(lo);
                        a.OD = E.dim_d(li); a.OH = E.dim_h(li); a.OW = E.dim_w(li);
                        a.sd = E.ndim == 3 ? 2 : 1; a.sh = 2; a.sw = 2;
                        a.taps = make_taps(E.ndim, 2, 0);
                        a.Cout = s.Cin; a.Ngemm = s.Cin; a.K = a.taps.n * s.Cout; a.Kpad = (a.K + 31) / 32 * 32;
                        { launch_conv_igemm(a, E.dtype, st, STAT_REP, sg); sg = ForkSig{}; }
                    } else {
                        // conv 3^d / 1^d: gather conv of d(raw) with flipped taps, once per concat source
                        a.scatter = 0;
                        a.ID = a.OD = E.dim_d(lo); a.IH = a.OH = E.dim_h(lo); a.IW = a.OW = E.dim_w(lo);
                        a.sd = a.sh = a.sw = 1;
                        const int k = s.ck == CK_K3 ? 3 : 1;
                        a.taps = make_taps(E.ndim, k, k == 3 ? 1 : 0);
                        a.K = a.taps.n * s.Cout; a.Kpad = (a.K + 31) / 32 * 32;
                        if (g0 >= 0) {
                            a.w = E.ws + s.wp_dg0; a.out = E.ws + E.tens[g0].off; a.Cout = a.Ngemm = E.tens[s.in0].C;
                            { launch_conv_igemm(a, E.dtype, st, STAT_REP, sg); sg = ForkSig{}; }
                        }
                        if (g1 >= 0) {
                            a.w = E.ws + s.wp_dg1; a.out = E.ws + E.tens[g1].off; a.Cout = a.Ngemm = E.tens[s.in1].C;
                            { launch_conv_igemm(a, E.dtype, st, STAT_REP, sg); sg = ForkSig{}; }
                        }
                    }
                });
            }
        }
        // ---- chains of finest-level ops that run group of samples by group of samples (seg_engine::run_chain)
        {
            double mb = E.sub_mb;
            if (mb < 0.0) mb = SEG_SUB_MB_DEFAULT;
            E.sub_nb = 0;
            const double per_sample_mb = (double)E.vol(0) * 16.0 * (double)E.esz() / 1e6;          // one 16-channel finest-level tensor
            if (mb > 0.0 && !E.use_vact && N > 1) {
                int nb = (int)(mb / per_sample_mb);
                if (nb < 1) nb = 1;
                while (nb > 1 && N % nb) --nb;                    // equal groups only (a unit's replica count is remembered per launch)
                if (nb < N) E.sub_nb = nb;
            }
            auto runs = [](const std::vector<char>& ok, std::vector<std::pair<int, int>>& out) {
                for (int i = 0; i < (int)ok.size();) {
                    if (!ok[i]) { ++i; continue; }
                    int j = i;
                    while (j < (int)ok.size() && ok[j]) ++j;
                    out.push_back({i, j});
                    i = j;
                }
            };
            std::vector<char> fok(E.fwd_ops.size(), 0);           // fwd_ops[0] = fill + ingest, fwd_ops[1 + si] = step si
            for (size_t si = 0; si < E.steps.size(); ++si) {
                const Step& s = E.steps[si];
                bool ok;
                if (s.type == ST_UNIT) ok = unit_sub((int)si);
                else if (s.type == ST_ACT) ok = unit_sub(s.ua) && (s.ub < 0 || unit_sub(s.ub));
                else if (s.type == ST_POOL) ok = E.tens[s.in].lvl <= E.sub_lvl;
                else ok = true;
                fok[1 + si] = ok ? 1 : 0;
            }
            runs(fok, E.fwd_chains);
            runs(E.bwd_sub, E.bwd_chains);
        }
        E.ws_bytes = align_up(cur, 4096);
        E.planned = true;
        (void)dt;
    }
};

int check_handle(seg_handle h) { return h ? 0 : fail("null handle"); }

}  // namespace

// =================================================================================================
// C-ABI
// =================================================================================================
extern "C" {

int seg_create(int net_kind, int ndim, int in_channels, int num_class, int init_features, int dtype, seg_handle* out) {
    if (!out) return fail("seg_create: out is null");
    if (net_kind != SEG_NET_VNET && net_kind != SEG_NET_UNET) return fail("seg_create: unknown net kind");
    if (ndim != 2 && ndim != 3) return fail("seg_create: ndim must be 2 or 3");
    if (dtype < 0 || dtype > 2) return fail("seg_create: dtype must be SEG_F32/F16/BF16");
    if (init_features != 16) return fail("seg_create: init_features must be 16 (GroupNorm(8) tiles; the reference never overrides the default)");
    if (num_class < 1 || num_class > 16) return fail("seg_create: num_class must be in 1..16");
    if (in_channels < 1 || in_channels > 16) return fail("seg_create: in_channels must be in 1..16");
    seg_engine* e = new seg_engine();
    e->kind = net_kind; e->ndim = ndim; e->in_ch = in_channels; e->ncls = num_class; e->feat = init_features; e->dtype = dtype;
    e->pad_img = in_channels > 3 || (ndim == 3 && in_channels > 1);
    e->loss_scale = dtype == DT_F16 ? 16384.f : 1.f;
    e->use_side = !(getenv("SEG_WGRAD_STREAM") && atoi(getenv("SEG_WGRAD_STREAM")) == 0);
    if (getenv("SEG_CONV3X")) e->use_conv3x = atoi(getenv("SEG_CONV3X")) != 0;
    if (getenv("SEG_STEMX")) e->use_stemx = atoi(getenv("SEG_STEMX")) != 0;
    if (getenv("SEG_PACK_SPLIT")) e->pack_split = atoi(getenv("SEG_PACK_SPLIT")) != 0;
    if (getenv("SEG_GN_VACT")) e->use_vact = atoi(getenv("SEG_GN_VACT")) != 0;
    if (getenv("SEG_GN_FOLD")) e->use_fold = atoi(getenv("SEG_GN_FOLD")) != 0;
    if (getenv("SEG_VHEAD")) e->use_vhead = atoi(getenv("SEG_VHEAD")) != 0;
    if (getenv("SEG_TAIL_WGRADS")) e->tail_wgrads = atoi(getenv("SEG_TAIL_WGRADS"));
    if (getenv("SEG_FORK_HEAVY_MB")) e->fork_heavy_bytes = atof(getenv("SEG_FORK_HEAVY_MB")) * 1e6;
    if (getenv("SEG_FLUSH_LATE")) e->flush_late = atoi(getenv("SEG_FLUSH_LATE")) != 0;
    if (getenv("SEG_HOLD_HEAVY_LVL")) e->hold_lvl = atoi(getenv("SEG_HOLD_HEAVY_LVL"));
    if (getenv("SEG_HOLD_HEAVY_MB")) e->hold_bytes = atof(getenv("SEG_HOLD_HEAVY_MB")) * 1e6;
    if (getenv("SEG_DUAL_GN")) e->dual_gn_bwd = atoi(getenv("SEG_DUAL_GN")) != 0;
    if (getenv("SEG_STEM_MAIN")) e->stem_on_main = atoi(getenv("SEG_STEM_MAIN")) != 0;
    if (getenv("SEG_SIDE_PRIO")) e->side_prio = atoi(getenv("SEG_SIDE_PRIO"));
    if (getenv("SEG_WGRAD_STREAMS")) e->n_side = atoi(getenv("SEG_WGRAD_STREAMS")) >= 2 ? 2 : 1;
    if (getenv("SEG_FORK_BATCH") && atoi(getenv("SEG_FORK_BATCH")) > 0) e->fork_batch = atoi(getenv("SEG_FORK_BATCH"));
    if (getenv("SEG_SUB_MB")) e->sub_mb = atof(getenv("SEG_SUB_MB"));
    if (getenv("SEG_SUB_LVL")) e->sub_lvl = atoi(getenv("SEG_SUB_LVL"));
    Builder b(*e);
    if (net_kind == SEG_NET_VNET) b.build_vnet(); else b.build_unet();
    *out = e;
    return 0;
}

void seg_destroy(seg_handle h) {
    if (!h) return;
    h->drop_graph();
    for (auto& r : h->prof_pool) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
    for (auto e : h->ready_ev) (void)hipEventDestroy(e);
    if (h->pack_fork) (void)hipEventDestroy(h->pack_fork);
    if (h->pack_done) (void)hipEventDestroy(h->pack_done);
    if (h->side_done) (void)hipEventDestroy(h->side_done);
    if (h->ar_ev) (void)hipEventDestroy(h->ar_ev);
    h->release_waiters();
    if (h->side) (void)hipStreamDestroy(h->side);
    if (h->side2_done) (void)hipEventDestroy(h->side2_done);
    if (h->side2) (void)hipStreamDestroy(h->side2);
    if (h->fork_flag) { (void)hipDeviceSynchronize(); (void)hipFree(h->fork_flag); }
    delete h;
}

int seg_param_count(seg_handle h) { return h ? (int)h->params.size() : -1; }
long long seg_param_numel(seg_handle h) { return h ? h->nparam : -1; }
int seg_param_info(seg_handle h, int index, char* name, int name_cap, int* shape8, int* ndim, long long* offset) {
    if (check_handle(h)) return -1;
    if (index < 0 || index >= (int)h->params.size()) return fail("seg_param_info: index out of range");
    const Param& p = h->params[index];
    if (name && name_cap > 0) { snprintf(name, name_cap, "%s", p.name.c_str()); }
    if (shape8) for (size_t i = 0; i < 8; ++i) shape8[i] = i < p.shape.size() ? p.shape[i] : 0;
    if (ndim) *ndim = (int)p.shape.size();
    if (offset) *offset = p.off;
    return 0;
}
int seg_dropout_calls(seg_handle h) { return h ? (int)h->drop_ch.size() : -1; }
int seg_dropout_ld(seg_handle h) { return h ? h->ld_mask() : -1; }
int seg_dropout_channels(seg_handle h, int call) {
    if (!h || call < 0 || call >= (int)h->drop_ch.size()) return -1;
    return h->drop_ch[call];
}

long long seg_dropout_draws(seg_handle h) { return h ? (long long)h->draws : -1; }
int seg_set_dropout_draws(seg_handle h, long long draws) {
    if (check_handle(h)) return -1;
    if (draws < 0 || draws > 0x7fffffffll) return fail("seg_set_dropout_draws: counter out of range");
    h->draws = (int)draws;
    if (h->ws) {          // bound: the device-side counter follows (seg_bind restores it from the host copy otherwise)
        h->release_waiters();
        if (h->side) (void)hipStreamSynchronize(h->side);
        if (hipDeviceSynchronize() != hipSuccess || hipMemcpy(h->ws + h->off_step, &h->draws, sizeof(int), hipMemcpyHostToDevice) != hipSuccess)
            return fail("seg_set_dropout_draws: counter upload failed");
    }
    return 0;
}

int seg_plan(seg_handle h, int n, int d, int hgt, int wid) {
    if (check_handle(h)) return -1;
    if (n < 1) return fail("seg_plan: batch must be >= 1");
    if (h->ndim == 2) d = 1;
    if ((h->ndim == 3 && (d % 16 || d < 16)) || hgt % 16 || wid % 16 || hgt < 16 || wid < 16)
        return fail("seg_plan: spatial dims must be multiples of 16 (four 2x down-samplings)");
    // a backward-only weight pack of the previous step may still be running on the (non-blocking) side stream into the workspace the
    // caller is about to replace
    h->release_waiters();
    if (h->side) { (void)hipStreamSynchronize(h->side); h->pack_bwd_pending = false; }
    if (h->side2) (void)hipStreamSynchronize(h->side2);
    h->drop_graph();
    h->N = n; h->Nplan = n; h->D = d; h->H = hgt; h->W = wid;
    g_err.clear();
    Planner pl(*h);
    pl.plan();
    if (!g_err.empty()) return -1;
    h->p = nullptr; h->g = nullptr; h->ws = nullptr;
    return 0;
}
int seg_plan_count(seg_handle h, int what) {
    if (!h || !h->planned) return -1;
    if (what == 2) return h->n_event_forks;             // last backward pass: fork events recorded on the main stream
    if (what == 3) return h->n_flag_forks;              // ... flag forks (the weight-gradient queue's command processor waits on a word the main queue's next kernel stores)
    if (what == 7) return h->n_sig_taken;               // ... of which the next kernel of the main stream stored the number itself
    if (what == 8) return h->n_sig_kernels;             // ... and one-wave kernels that stored it
    if (what == 9) {                                    // 1: no released batch is left waiting (the host checker also reads the flag word itself)
#ifdef SEG_EMU
        if (h->fork_flag && *h->fork_flag != h->fork_seq) return 0;
#endif
        return h->sig_pending == 0;
    }
    if (what == 4) return h->sub_nb;                    // samples per group of the sub-batched finest level (0: whole-batch launches)
    if (what == 5 || what == 6) {                       // forward / backward ops that run group by group
        int n = 0;
        for (auto& c : (what == 5 ? h->fwd_chains : h->bwd_chains)) n += c.second - c.first;
        return n;
    }
    int n = 0;
    for (auto& s : h->steps) {
        if (what == 0) n += s.type == ST_ACT && s.vact;                                   // activations applied by their consumer (never written)
        else if (what == 1) n += s.type == ST_UNIT;                                       // convolution units
        else return -1;
    }
    return n;
}
long long seg_workspace_bytes(seg_handle h) { return (h && h->planned) ? (long long)h->ws_bytes : -1; }

int seg_bind(seg_handle h, float* params, float* grads, void* workspace) {
    if (check_handle(h)) return -1;
    if (!h->planned) return fail("seg_bind: call seg_plan first");
    if (!params || !workspace) return fail("seg_bind: params/workspace must not be null");
    if (((uintptr_t)params | (uintptr_t)grads | (uintptr_t)workspace) & 255) return fail("seg_bind: buffers must be 256-byte aligned");
    h->release_waiters();
    if (h->side) { (void)hipStreamSynchronize(h->side); h->pack_bwd_pending = false; }     // see seg_plan
    if (h->side2) (void)hipStreamSynchronize(h->side2);
    h->drop_graph();
    h->p = params; h->g = grads; h->ws = (char*)workspace;
    // resolve and upload the weight re-layout descriptors; reset the device-side step counter
    std::vector<PackDesc> d = h->packdescs;
    for (auto& x : d) {
        x.src = h->p + (long long)(uintptr_t)x.src;
        x.dst = h->ws + (size_t)(uintptr_t)x.dst;
    }
    if (hipMemcpy(h->ws + h->off_packdesc, d.data(), d.size() * sizeof(PackDesc), hipMemcpyHostToDevice) != hipSuccess)
        return fail("seg_bind: descriptor upload failed");
    if (hipMemset(h->ws + h->off_step, 0, 256) != hipSuccess) return fail("seg_bind: memset failed");
    if (h->draws && hipMemcpy(h->ws + h->off_step, &h->draws, sizeof(int), hipMemcpyHostToDevice) != hipSuccess)
        return fail("seg_bind: counter upload failed");
    return 0;
}

int seg_pack_weights(seg_handle h, void* stream) {
    if (check_handle(h)) return -1;
    if (!h->ws) return fail("seg_pack_weights: not bound");
    hipStream_t st = (hipStream_t)stream;
    const PackDesc* descs = (const PackDesc*)(h->ws + h->off_packdesc);
    const int nall = (int)h->packdescs.size(), nbwd = nall - h->npack_fwd;
    if (h->use_side && h->pack_split && !h->capturing && nbwd > 0 && h->npack_fwd > 0) {
        h->ensure_side();
        if (!h->pack_fork) {
            (void)hipEventCreateWithFlags(&h->pack_fork, hipEventDisableTiming);
            (void)hipEventCreateWithFlags(&h->pack_done, hipEventDisableTiming);
        }
        (void)hipEventRecord(h->pack_fork, st);            // the parameters are final on `st` here
        (void)hipStreamWaitEvent(h->side, h->pack_fork, 0);
        launch_pack(descs + h->npack_fwd, nbwd, (int)h->pack_max, h->dtype, h->side);
        (void)hipEventRecord(h->pack_done, h->side);
        h->pack_bwd_pending = true;                        // seg_backward_range waits for it
        launch_pack(descs, h->npack_fwd, (int)h->pack_max, h->dtype, st, h->ride_on ? h->ride_pack : StepRider{});
    } else {
        launch_pack(descs, nall, (int)h->pack_max, h->dtype, st, h->ride_on ? h->ride_pack : StepRider{});
    }
    return hipGetLastError() == hipSuccess ? 0 : fail("seg_pack_weights: launch failed");
}

int seg_forward(seg_handle h, const float* x, int mask_mode, const float* masks, unsigned long long seed, float* logits,
                float* probs, void* stream) {
    if (check_handle(h)) return -1;
    if (!h->ws) return fail("seg_forward: not bound");
    if (!x || !logits || !probs) return fail("seg_forward: null tensor");
    hipStream_t st = (hipStream_t)stream;
    h->mask_mode = mask_mode;
    const size_t mbytes = (size_t)h->drop_ch.size() * h->N * h->ld_mask() * 4;
    if (mask_mode == SEG_MASKS_GIVEN) {
        if (!masks) return fail("seg_forward: SEG_MASKS_GIVEN needs a mask table");
        (void)hipMemcpyAsync(h->ws + h->off_masks, masks, mbytes, hipMemcpyDeviceToDevice, st);
    } else if (mask_mode == SEG_MASKS_RANDOM) {
        launch_dropout_masks((float*)(h->ws + h->off_masks), (int)h->drop_ch.size(), h->N, h->ld_mask(), 0.2f, seed,
                             (const int*)(h->ws + h->off_step), st, !h->ride_on);      // (train step: the ingest kernel advances the counter)
        ++h->draws;
    }
    h->cur_x = x; h->cur_logits = logits; h->cur_probs = probs;
    h->run_ops(h->fwd_ops, h->fwd_chains, 0, (int)h->fwd_ops.size(), st, false);
    return hipGetLastError() == hipSuccess ? 0 : fail(std::string("seg_forward: ") + hipGetErrorString(hipGetLastError()));
}

static int backward_slice(seg_handle h, const float* dlogits, int zero_grads, int op_begin, int op_end, int join, void* stream);
int seg_backward_range(seg_handle h, const float* dlogits, int zero_grads, int op_begin, int op_end, void* stream) {
    return backward_slice(h, dlogits, zero_grads, op_begin, op_end, 1, stream);
}
int seg_backward_slice(seg_handle h, const float* dlogits, int zero_grads, int op_begin, int op_end, int join, void* stream) {
    return backward_slice(h, dlogits, zero_grads, op_begin, op_end, join, stream);
}
int seg_side_wait(seg_handle h, void* stream) {
    if (check_handle(h)) return -1;
    hipStream_t st = (hipStream_t)stream;
    h->emit_sig(st);
    if (h->use_side && h->side) {
        (void)hipEventRecord(h->side_done, h->side); (void)hipStreamWaitEvent(st, h->side_done, 0);
        if (h->side2) { (void)hipEventRecord(h->side2_done, h->side2); (void)hipStreamWaitEvent(st, h->side2_done, 0); }
    }
    return hipGetLastError() == hipSuccess ? 0 : fail("seg_side_wait: event error");
}
static int backward_slice(seg_handle h, const float* dlogits, int zero_grads, int op_begin, int op_end, int join, void* stream) {
    if (check_handle(h)) return -1;
    if (!h->ws || !h->g) return fail("seg_backward: gradients not bound");
    if (!dlogits) return fail("seg_backward: null dlogits");
    const int nops = (int)h->bwd_ops.size();
    if (op_begin < 0 || op_end > nops || op_begin > op_end) return fail("seg_backward_range: bad op range");
    hipStream_t st = (hipStream_t)stream;
    if (zero_grads && op_begin == 0) (void)hipMemsetAsync(h->g, 0, (size_t)h->nparam * 4, st);
    h->cur_dlogits = dlogits;
    if (op_begin == 0) { h->wgrad_seq = 0; h->ready_used = 0; h->hold_open = false; h->n_event_forks = 0; h->n_flag_forks = 0; h->n_sig_kernels = 0; h->n_sig_taken = 0; }
    if (h->pack_bwd_pending) { (void)hipStreamWaitEvent(st, h->pack_done, 0); h->pack_bwd_pending = false; }
    h->run_ops(h->bwd_ops, h->bwd_chains, op_begin, op_end, st, true);
    if (join) h->join_side(st);
    else { h->flush_side(st); h->emit_sig(st); }         // the queued weight gradients of this slice are released; `stream` does not wait for them
    return hipGetLastError() == hipSuccess ? 0 : fail(std::string("seg_backward: ") + hipGetErrorString(hipGetLastError()));
}
int seg_backward(seg_handle h, const float* dlogits, int zero_grads, void* stream) {
    if (check_handle(h)) return -1;
    return seg_backward_range(h, dlogits, zero_grads, 0, (int)h->bwd_ops.size(), stream);
}
int seg_backward_ops(seg_handle h) { return (h && h->planned) ? (int)h->bwd_ops.size() : -1; }
int seg_backward_bucket(seg_handle h, double tail_fraction, int* op_split, long long* param_offset) {
    if (check_handle(h)) return -1;
    if (!h->planned) return fail("seg_backward_bucket: call seg_plan first");
    if (!op_split || !param_offset) return fail("seg_backward_bucket: null output");
    // writers left per parameter; after op k the finished gradients form a suffix [S(k), nparam) of the flat buffer
    std::vector<int> left(h->params.size(), 0);
    for (auto& w : h->bwd_writes) for (int p : w) if (p >= 0) ++left[p];
    const int nops = (int)h->bwd_ops.size();
    *op_split = nops; *param_offset = 0;
    for (int k = 0; k < nops; ++k) {
        for (int p : h->bwd_writes[k]) if (p >= 0) --left[p];
        int first_done = (int)h->params.size();
        while (first_done > 0 && left[first_done - 1] == 0) --first_done;
        const long long S = first_done < (int)h->params.size() ? h->params[first_done].off : h->nparam;
        if ((double)(h->nparam - S) >= tail_fraction * (double)h->nparam) { *op_split = k + 1; *param_offset = S; return 0; }
    }
    return 0;
}

int seg_set_loss_scale(seg_handle h, float scale) {
    if (check_handle(h)) return -1;
    if (!(scale > 0.f)) return fail("seg_set_loss_scale: scale must be positive");
    if (scale != h->loss_scale) h->drop_graph();          // the scale is baked into the captured launches
    h->loss_scale = scale;
    return 0;
}
float seg_get_loss_scale(seg_handle h) { return h ? h->loss_scale : 0.f; }

long long seg_loss_ws_bytes(int n, int c) { return (long long)align_up(loss_sums_count(n, c) * sizeof(double) * STAT_REP); }

static int fill_loss(LossArgs& a, const float* logits, const void* target, int label_type, int n, int c, long long v,
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

// riders: the overflow flag was cleared and the step counter will be advanced by StepRiders of neighbouring launches (seg_train_step)
static int adam_step_impl(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, long long numel, float lr, float beta1,
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
int seg_adam_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, long long numel, float lr, float beta1,
                  float beta2, float eps, float weight_decay, int decoupled, float inv_scale, int check_finite, int* state, void* stream) {
    return adam_step_impl(params, grads, exp_avg, exp_avg_sq, numel, lr, beta1, beta2, eps, weight_decay, decoupled, inv_scale, check_finite, state,
                          stream, false);
}

// One optimisation step of the reference loop (model/modelVNet.py:570-596) enqueued by ONE call: the host side of a step is then a
// single FFI crossing plus this function's launches (round 2: >= 6 crossings, each with Python argument marshalling and torch
// stream look-ups, 0.7-4 ms of host time per 4.5 ms step depending on the box).
int seg_train_step(seg_handle h, const seg_train_args* a, void* stream) {
    if (check_handle(h)) return -1;
    if (!a) return fail("seg_train_step: args is null");
    if (!h->ws || !h->g) return fail("seg_train_step: not bound");
    if (!a->x || !a->target || !a->logits || !a->probs || !a->dlogits || !a->loss_ws || !a->out3)
        return fail("seg_train_step: null tensor");
    if (!a->exp_avg || !a->exp_avg_sq || !a->opt_state) return fail("seg_train_step: optimiser state is null");
    hipStream_t st = (hipStream_t)stream;
    if (!a->packed && seg_pack_weights(h, stream)) return -1;
    // Serial section at the step boundary (profiles/r04_trace_timeline.txt: fill, overflow check, Adam, counter, re-pack, masks, counter, fill,
    // ingest - nothing overlaps them): the one-wave bookkeeping launches ride on their neighbours.  SEG_STEP_RIDERS=0: separate launches.
    const char* riders_e = getenv("SEG_STEP_RIDERS");
    const bool riders_env = !(riders_e && atoi(riders_e) == 0);
    const bool riders = riders_env && h->sub_nb == 0;
    const long long v = h->vol(0);
    struct RideGuard { seg_engine* e; ~RideGuard() { e->ride_on = false; e->ride_zero = nullptr; } } ride_guard{h};      // every way out of the step
    h->ride_on = riders; h->head_zeroed = false;
    h->ride_ingest = StepRider{}; h->ride_pack = StepRider{};
    if (riders) {
        if (a->mask_mode == SEG_MASKS_RANDOM) h->ride_ingest.bump = (int*)(h->ws + h->off_step);      // the dropout draw counter (after the mask kernel read it)
        h->ride_ingest.clear = a->opt_state + 1;                                                   // this step's overflow flag
        h->ride_pack.bump = a->opt_state; h->ride_pack.gate = a->opt_state + 1; h->ride_pack.tally = a->opt_state + 2;     // = adam_bump_kernel
        if (!a->loss_cb) { h->ride_zero = (double*)a->loss_ws; h->ride_zero_n = (long long)loss_sums_count(h->N, h->ncls) * STAT_REP; }
        else h->ride_zero = nullptr;
    }
    const int frc = seg_forward(h, a->x, a->mask_mode, a->masks, a->seed, a->logits, a->probs, stream);
    const bool zeroed = riders && h->head_zeroed;
    h->ride_zero = nullptr;
    if (frc) { h->ride_on = false; return -1; }
    const double lbytes = (double)h->N * v * (4.0 * h->ncls + ((a->label_type & 15) == SEG_LABEL_U8 ? 1.0 : (a->label_type & 15) == SEG_LABEL_I64 ? 8.0 : 4.0));
    int pi = h->prof_begin(st, SEG_K_MISC, 2.0 * lbytes + 4.0 * h->N * v * h->ncls, 0.0);
    if (a->loss_cb) {
        // exact global-batch loss: the rank's batch-global sums are exchanged between the reduction and the finalize (parallel.GlobalBatchLoss)
        if (seg_loss_reduce(a->logits, a->target, a->label_type, h->N, h->ncls, v, a->loss_kind, a->focal_alpha, a->focal_gamma, a->loss_ws, stream)) return -1;
        const long long ng = a->loss_cb(a->cb_user, (double*)a->loss_ws, seg_loss_shared_doubles());
        if (ng < 0) return fail("seg_train_step: the loss exchange hook failed");
        if (seg_loss_finalize(a->logits, a->target, a->label_type, h->N, h->ncls, v, a->loss_kind, a->focal_alpha, a->focal_gamma, a->class_alpha,
                              (int)ng, a->loss_ws, a->out3, stream)) return -1;
    } else {
        LossArgs la;
        if (fill_loss(la, a->logits, a->target, a->label_type, h->N, h->ncls, v, a->loss_kind, a->focal_alpha, a->focal_gamma, a->loss_ws)) return -1;
        la.class_alpha = a->class_alpha; la.out = a->out3; la.prezeroed = zeroed ? 1 : 0;      // (the head kernel cleared the workspace)
        launch_loss_forward(la, st);
        if (hipGetLastError() != hipSuccess) return fail("seg_train_step: loss launch failed");
    }
    if (seg_loss_backward(a->logits, a->target, a->label_type, h->N, h->ncls, v, a->loss_kind, a->focal_alpha, a->focal_gamma, a->loss_ws,
                          h->loss_scale, a->dlogits, stream)) return -1;
    h->prof_end(st, pi);
    if (a->bucket_cb) {
        // bucketed gradient exchange: every finished suffix of the flat gradient buffer is handed to the caller's hook while the finer levels
        // still run.  On the GPU the hook runs behind an auxiliary stream that waits for the caller's stream and the
        // weight-gradient stream, so the backward pass itself never stalls at a bucket boundary.
        if (a->nfrac < 0 || a->nfrac > 4) return fail("seg_train_step: nfrac must be 0..4");
        const int nops = (int)h->bwd_ops.size();
        hipStream_t aux = (hipStream_t)a->aux_stream;
        if (aux && !h->ar_ev) (void)hipEventCreateWithFlags(&h->ar_ev, hipEventDisableTiming);
        int prev_k = 0, idx = 0;
        long long prev_off = h->nparam;
        for (int f = 0; f < a->nfrac; ++f) {
            int k = 0; long long off = 0;
            if (seg_backward_bucket(h, a->fractions[f], &k, &off)) return -1;
            if (k <= prev_k || k >= nops || off >= prev_off) continue;         // boundaries that coincide are skipped
            if (backward_slice(h, a->dlogits, 1, prev_k, k, aux ? 0 : 1, stream)) return -1;
            if (aux) {
                (void)hipEventRecord(h->ar_ev, st); (void)hipStreamWaitEvent(aux, h->ar_ev, 0);
                if (seg_side_wait(h, aux)) return -1;
            }
            if (a->bucket_cb(a->cb_user, idx++, off, prev_off - off)) return fail("seg_train_step: the gradient exchange hook failed");
            prev_k = k; prev_off = off;
        }
        if (backward_slice(h, a->dlogits, 1, prev_k, nops, 1, stream)) return -1;
        if (aux && idx) { (void)hipEventRecord(h->ar_ev, aux); (void)hipStreamWaitEvent(st, h->ar_ev, 0); }     // whatever the hooks queued on the auxiliary stream itself
        if (a->bucket_cb(a->cb_user, idx, 0, prev_off) || a->bucket_cb(a->cb_user, -1, 0, 0)) return fail("seg_train_step: the gradient exchange hook failed");
    } else
    if (seg_backward(h, a->dlogits, 1, stream)) return -1;
    // fused optimiser: p, m, v read + written, g read (+ once more by the overflow check); re-pack: fp32 masters read, run-dtype layouts written
    pi = h->prof_begin(st, SEG_K_MISC, (double)h->nparam * (28.0 + (a->check_finite ? 4.0 : 0.0) + 4.0 + 3.0 * (double)h->esz()), 0.0);
    if (adam_step_impl(h->p, h->g, a->exp_avg, a->exp_avg_sq, h->nparam, a->lr, a->beta1, a->beta2, a->eps, a->weight_decay, a->decoupled,
                       1.0f / (h->loss_scale * (a->grad_div > 0.f ? a->grad_div : 1.f)), a->check_finite, a->opt_state, stream, riders)) {
        h->ride_on = false;
        return -1;
    }
    const int rc = seg_pack_weights(h, stream);          // (its first thread advances the optimiser's step counter: ride_pack)
    h->ride_on = false;
    h->prof_end(st, pi);
    return rc;
}

// The same step captured once as a HIP graph and replayed: ~250 launches + ~60 event operations become one hipGraphLaunch on the host.  For
// hosts that cannot enqueue a step as fast as the GPU runs it (BENCH_r02: 4.2 ms of host time per 5.5 ms step on the driver's box against
// 0.9 ms on the builder's); on a fast host the stream launches are as fast or faster (the graph orders the weight-gradient branch less
// favourably), so the caller measures both and picks (SegEngine.train_step(launch="auto"), bench.py --launch auto).
// Every pointer and scalar of `a` (and the loss scale) is baked in; the device-side dropout / Adam step counters keep advancing.
int seg_train_graph_capture(seg_handle h, const seg_train_args* a, void* stream) {
    if (check_handle(h)) return -1;
    if (!a) return fail("seg_train_graph_capture: args is null");
    if (!a->packed) return fail("seg_train_graph_capture: run one ordinary step first (the captured step starts from packed weights)");
    if (a->bucket_cb || a->loss_cb) return fail("seg_train_graph_capture: a step with exchange hooks cannot be captured (host callbacks)");
    if (h->prof_mask) return fail("seg_train_graph_capture: switch seg_profile_enable off first");
    hipStream_t st = (hipStream_t)stream;
    h->drop_graph();
    // nothing un-captured may be pending on the streams the capture forks to
    (void)hipStreamSynchronize(st);
    h->release_waiters();
    if (h->side) (void)hipStreamSynchronize(h->side);
    if (h->side2) (void)hipStreamSynchronize(h->side2);
    h->pack_bwd_pending = false;
    h->ensure_side();
    if (hipStreamBeginCapture(st, hipStreamCaptureModeRelaxed) != hipSuccess) { (void)hipGetLastError(); return fail("seg_train_graph_capture: hipStreamBeginCapture failed"); }
    h->capturing = true;
    const int draws0 = h->draws;
    const int rc = seg_train_step(h, a, stream);
    h->capturing = false;
    h->draws = draws0;                                   // the capture itself executes nothing
    hipGraph_t g = nullptr;
    const hipError_t ec = hipStreamEndCapture(st, &g);
    if (rc || ec != hipSuccess || !g) {
        if (g) (void)hipGraphDestroy(g);
        (void)hipGetLastError();
        return rc ? -1 : fail(std::string("seg_train_graph_capture: hipStreamEndCapture failed: ") + hipGetErrorString(ec));
    }
    hipGraphExec_t ge = nullptr;
    if (hipGraphInstantiate(&ge, g, nullptr, nullptr, 0) != hipSuccess || !ge) {
        (void)hipGraphDestroy(g); (void)hipGetLastError();
        return fail("seg_train_graph_capture: hipGraphInstantiate failed");
    }
    h->tgraph = g; h->tgraph_exec = ge; h->tgraph_mask_mode = a->mask_mode;
    return 0;
}
int seg_train_graph_launch(seg_handle h, void* stream) {
    if (check_handle(h)) return -1;
    if (!h->tgraph_exec) return fail("seg_train_graph_launch: no captured step (seg_train_graph_capture; a re-plan, re-bind or loss-scale change drops it)");
    if (hipGraphLaunch(h->tgraph_exec, (hipStream_t)stream) != hipSuccess) { (void)hipGetLastError(); return fail("seg_train_graph_launch: hipGraphLaunch failed"); }
    if (h->tgraph_mask_mode == SEG_MASKS_RANDOM) ++h->draws;
    h->tgraph_stream = (hipStream_t)stream;
    h->q_clean = false;          // the replayed backward pass used the GroupNorm-backward sums: an eager backward that follows must clear them
    return 0;
}
int seg_train_graph_ready(seg_handle h) { return (h && h->tgraph_exec) ? 1 : 0; }

int seg_op_conv(const seg_conv_args* a, int dtype, void* stream) {
    if (!a || !a->in0 || !a->w || !a->out) return fail("seg_op_conv: null pointer");
    const int cin = a->C0 + a->C1;
    if (cin < 8 || (cin & (cin - 1)) || a->C0 % 8) return fail("seg_op_conv: channel counts must be powers of two >= 8");
    if (a->Cout % 16 || a->Ngemm % 16 || a->Kpad % 32 || a->Kpad < a->K) return fail("seg_op_conv: bad GEMM extents");
    launch_conv_igemm(*a, dtype, (hipStream_t)stream);
    return hipGetLastError() == hipSuccess ? 0 : fail("seg_op_conv: launch failed");
}
int seg_op_conv_kernel(const seg_conv_args* a) { return a ? (conv_uses_stream_kernel(*a) ? 1 : 0) : -1; }
long long seg_op_wgrad_partial_bytes(const seg_wgrad_args* a) { return a ? (long long)wgrad_partial_bytes(*a) : -1; }
int seg_op_wgrad(const seg_wgrad_args* a, float* partial_scratch, int dtype, void* stream) {
    if (!a || !a->dr || !a->x0 || !a->dw || !partial_scratch) return fail("seg_op_wgrad: null pointer");
    if (a->P % 16) return fail("seg_op_wgrad: P must be a multiple of 16");
    if (a->stem ? (a->Q > 32) : (a->Q % 16 != 0)) return fail("seg_op_wgrad: bad Q");
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

int seg_profile_enable(seg_handle h, unsigned mask) {
    if (check_handle(h)) return -1;
    h->prof_mask = mask;
    return 0;
}
int seg_profile_read(seg_handle h, int* calls, float* ms, double* bytes, double* flops) {
    if (check_handle(h)) return -1;
    for (int c = 0; c < SEG_K_COUNT; ++c) {
        if (calls) calls[c] = 0;
        if (ms) ms[c] = 0.f;
        if (bytes) bytes[c] = 0.0;
        if (flops) flops[c] = 0.0;
    }
    for (size_t i = 0; i < h->prof_used; ++i) {
        auto& r = h->prof_pool[i];
        (void)hipEventSynchronize(r.b);
        float t = 0.f;
        (void)hipEventElapsedTime(&t, r.a, r.b);
        if (calls) calls[r.cls] += 1;
        if (ms) ms[r.cls] += t;
        if (bytes) bytes[r.cls] += r.bytes;
        if (flops) flops[r.cls] += r.flops;
    }
    h->prof_used = 0;
    return 0;
}

const char* seg_last_error(void) { return g_err.c_str(); }

const char* seg_build_info(void) {
#ifdef SEG_EMU
    return "segengine host-checker build (tests only)";
#else
    return "segengine gfx950";
#endif
}

}  // extern "C"
