// Segmentation engine, C-ABI part with a handle (include/segengine.h): life cycle, plan / bind, forward, backward, the one-call train step and its
// captured form, profiling brackets.  The graph / planner live in engine_plan.hip, the stateless entry points in capi_ops.hip, the shared
// declarations (and the overview of the design) in engine_internal.h.
#include "engine_internal.h"

namespace segi {
thread_local std::string g_err;
int fail(const std::string& m) { g_err = m; return -1; }
int check_handle(seg_handle h) { return h ? 0 : fail("null handle"); }
}  // namespace segi


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
    // product switches (documented in include/segengine.h; each selects a complete, tested path of the library)
    e->use_side = knob_i("SEG_WGRAD_STREAM", 1) != 0;
    e->use_conv3x = knob_i("SEG_CONV3X", e->use_conv3x) != 0;
    e->use_stemx = knob_i("SEG_STEMX", e->use_stemx) != 0;
    e->pack_split = knob_i("SEG_PACK_SPLIT", e->pack_split) != 0;
    e->use_fold = knob_i("SEG_GN_FOLD", e->use_fold) != 0;
    e->use_vhead = knob_i("SEG_VHEAD", e->use_vhead) != 0;
    e->dual_gn_bwd = knob_i("SEG_DUAL_GN", e->dual_gn_bwd) != 0;
    e->use_coop = knob_i("SEG_GN_COOP", e->use_coop) != 0;
    e->use_vact = knob_i("SEG_VACT", e->use_vact);
    e->use_head_fuse = knob_i("SEG_HEAD_FUSE", e->use_head_fuse) != 0;
    e->use_rq_fuse = knob_i("SEG_RQ_FUSE", e->use_rq_fuse) != 0;
#ifdef SEG_DIAG
    e->w3_mode = knob_i("SEG_DIAG_W3_MODE", e->w3_mode);
#endif
    build_network(*e, net_kind);
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
    if (h->xchg) (void)hipStreamDestroy(h->xchg);
    if (h->side) (void)hipStreamDestroy(h->side);
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
    if (h->side) { (void)hipStreamSynchronize(h->side); h->pack_bwd_pending = false; }
    h->drop_graph();
    h->N = n; h->D = d; h->H = hgt; h->W = wid;
    g_err.clear();
    plan_engine(*h);
    if (!g_err.empty()) return -1;
    h->p = nullptr; h->g = nullptr; h->ws = nullptr;
    return 0;
}
int seg_plan_count(seg_handle h, int what) {
    if (!h || !h->planned) return -1;
    if (what == 2) return h->n_event_forks;             // last backward pass: fork events recorded on the main stream
    int n = 0;
    for (auto& s : h->steps) {
        if (what == 1) n += s.type == ST_UNIT;                                            // convolution units
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
    if (h->side) { (void)hipStreamSynchronize(h->side); h->pack_bwd_pending = false; }     // see seg_plan
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
    h->run_ops(h->fwd_ops, 0, (int)h->fwd_ops.size(), st);
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
    h->finish_wgrads();
    if (h->use_side && h->side) { (void)hipEventRecord(h->side_done, h->side); (void)hipStreamWaitEvent(st, h->side_done, 0); }
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
    if (op_begin == 0) { h->ready_used = 0; h->hold_open = false; h->n_event_forks = 0; }
    if (h->pack_bwd_pending) { (void)hipStreamWaitEvent(st, h->pack_done, 0); h->pack_bwd_pending = false; }
    h->run_ops(h->bwd_ops, op_begin, op_end, st);
    if (join) h->join_side(st);
    else h->flush_side(st);                              // the queued weight gradients of this slice are released; `stream` does not wait for them
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
    const bool riders = knob_i("SEG_STEP_RIDERS", 1) != 0;
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
    const bool native = !a->bucket_cb && h->rccl_comm && h->rccl_allreduce;
    if (a->bucket_cb || native) {
        // bucketed gradient exchange: every finished suffix of the flat gradient buffer is exchanged while the finer levels still run (DESIGN.md
        // section 6) - by the caller's hook, or by the library itself (seg_set_rccl_comm: ncclAllReduce on the exchange stream, no host callback).
        // On the GPU the exchange of every bucket but the last is ordered behind an auxiliary stream that waits for the caller's stream and the
        // weight-gradient stream, so the backward pass itself never stalls at a bucket boundary.
        if (a->nfrac < 0 || a->nfrac > 4) return fail("seg_train_step: nfrac must be 0..4");
        const int nops = (int)h->bwd_ops.size();
        hipStream_t aux = (hipStream_t)a->aux_stream;
#ifndef SEG_EMU
        if (native && !aux) {
            if (!h->xchg && hipStreamCreateWithFlags(&h->xchg, hipStreamNonBlocking) != hipSuccess) return fail("seg_train_step: cannot create the exchange stream");
            aux = h->xchg;
        }
#endif
        if (aux && !h->ar_ev) (void)hipEventCreateWithFlags(&h->ar_ev, hipEventDisableTiming);
        auto exchange = [&](int idx, long long off, long long cnt, hipStream_t on) -> int {
            if (!native) return a->bucket_cb(a->cb_user, idx, off, cnt);
            return cnt > 0 ? h->rccl_allreduce(h->g + off, h->g + off, (size_t)cnt, 7 /* ncclFloat */, 0 /* ncclSum */, h->rccl_comm, on) : 0;
        };
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
            if (exchange(idx++, off, prev_off - off, aux ? aux : st)) return fail("seg_train_step: the gradient exchange failed");
            prev_k = k; prev_off = off;
        }
        if (backward_slice(h, a->dlogits, 1, prev_k, nops, 1, stream)) return -1;
        if (aux && idx) { (void)hipEventRecord(h->ar_ev, aux); (void)hipStreamWaitEvent(st, h->ar_ev, 0); }     // whatever was queued on the auxiliary stream itself
        if (exchange(idx, 0, prev_off, st)) return fail("seg_train_step: the gradient exchange failed");
        if (!native && a->bucket_cb(a->cb_user, -1, 0, 0)) return fail("seg_train_step: the gradient exchange hook failed");
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

int seg_set_rccl_comm(seg_handle h, void* comm, void* allreduce_fn) {
    if (check_handle(h)) return -1;
    if (comm && !allreduce_fn) return fail("seg_set_rccl_comm: the ncclAllReduce of the library that created the communicator must be given");
    h->drop_graph();
    h->rccl_comm = comm;
    h->rccl_allreduce = comm ? (seg_engine::rccl_allreduce_t)allreduce_fn : nullptr;
    return 0;
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
    if (h->rccl_comm) return fail("seg_train_graph_capture: a step with an RCCL communicator (seg_set_rccl_comm) is not captured");
    if (h->prof_mask) return fail("seg_train_graph_capture: switch seg_profile_enable off first");
    hipStream_t st = (hipStream_t)stream;
    h->drop_graph();
    // nothing un-captured may be pending on the streams the capture forks to
    (void)hipStreamSynchronize(st);
    if (h->side) (void)hipStreamSynchronize(h->side);
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
#ifdef SEG_BUILD_ID
    return "segengine gfx950 " SEG_BUILD_ID;          // build.py: sha256 over the sources + flags (which binary a profile was taken from)
#else
    return "segengine gfx950";
#endif
#endif
}

}  // extern "C"
