/* libsegengine — C-ABI of the MI355X (gfx950) segmentation engine.
 *
 * Drop-in boundary for the hot path of junqiangchen/PytorchDeepLearing.  The
 * reference has no FFI of its own — its boundary is Python duck typing — so every entry point
 * cites the reference interface it stands behind.  Plain pointers and sizes only; no torch types.
 *
 * Ownership: the caller allocates and owns every buffer (parameters, gradients, optimiser state,
 * workspace, inputs, outputs); the engine keeps views plus small host-side descriptors.
 * Errors: every function returns 0 on success, <0 on failure; seg_last_error() gives the message.
 * Threading: calls are stream-ordered and asynchronous; a handle is not thread-safe.
 * All device pointers must be 256-byte aligned.  `stream` is a hipStream_t.
 *
 * Environment switches read by the PRODUCT library (each selects a complete path that the parity tests cover; none is needed to run; read once per
 * process unless noted) - the whole list:
 *   SEG_WGRAD_STREAM=0   weight gradients on the caller's stream instead of the library's second (low-priority) stream
 *   SEG_CONV3X=0         16-bit halo convolutions through conv3_kernel (round 1) instead of c3x::conv3x_kernel - bit-identical results
 *   SEG_STEMX=0          separate image-stem / GroupNorm / stem weight-gradient kernels instead of the fused input block
 *   SEG_VHEAD=0          the head writes its data-gradient tensor instead of the on-the-fly form
 *   SEG_GN_FOLD=0        GroupNorm finalize as launches of its own;  SEG_DUAL_GN=0: one GroupNorm-backward pass per branch of the VNet input block
 *   SEG_GN_COOP=0        (read per seg_create) GroupNorm backward of the >= 64-channel levels as reduce + apply launches instead of the one-launch
 *                        kernel whose workgroups exchange their partial sums inside the launch
 *   SEG_HEAD_FUSE=0      (read per seg_create) the 1^d head (networks/VNet3d.py:83-99) as a launch of its own instead of inside the activation pass that produces its
 *                        input (bit-identical results)
 *   SEG_RQ_FUSE=0        (read per seg_create) the GroupNorm-backward reduction of a VNet up-conv unit as a launch of its own instead of riding on the
 *                        data-gradient launch of the 1^d conv that produces its gradient
 *   SEG_VACT=0|2         (read per seg_create) the activation between a VNet up-conv and the 1^d conv on the concat (networks/VNet3d.py:72-77): 0 = written as a
 *                        tensor at every level, 2 = applied by its two readers on load at every level the kernels allow (default: on tensors >= 16 MB) -
 *                        bit-identical results
 *   SEG_STEP_RIDERS=0    (read per step) step counters / flag clears as launches of their own instead of riding on neighbouring kernels
 *   SEG_PACK_SPLIT=0     the weight re-layout as one launch on the caller's stream
 *   SEG_CONV_STREAM=0    the generic implicit-GEMM kernel also where the register-resident streaming conv applies
 *   SEG_WG_DIRECT=0      1^d-conv weight gradients through wgrad_kernel instead of the multi-step streaming kernel
 *   SEG_W3_BOX16=0|2     the 4 x 8 x 16-box weight gradient of the 16-channel level: off / forced onto small volumes (operator tests)
 *   SEG_C3X_MAP="cin:cout:w=id,..."   per-shape halo-conv tiling override (tools/tune_conv3x.py)
 *   SEG_C3X16_REUSE=0    the 16 -> 16 channel 3-D halo convs through conv3x16_kernel (one LDS fragment read per MFMA, weight layout 2) instead of
 *                        conv3x16r_kernel (fragments reused across the kh taps, weight layout 3) - bit-identical on integer data
 * The tuning knobs and measured-slower paths of rounds 1-5 (double-buffered weight gradient, GroupNorm in the consumer conv, persistent halo convs, flag
 * forks, sub-batched levels, second weight-gradient stream) were removed from the sources in round 6; what each measured is in profiles/HISTORY.md.
 */
#ifndef SEGENGINE_H
#define SEGENGINE_H
#ifdef __cplusplus
extern "C" {
#endif

typedef struct seg_engine* seg_handle;

enum { SEG_NET_VNET = 0, SEG_NET_UNET = 1 };
enum { SEG_F32 = 0, SEG_F16 = 1, SEG_BF16 = 2 };
enum { SEG_LABEL_U8 = 0, SEG_LABEL_I32 = 1, SEG_LABEL_I64 = 2, SEG_LABEL_F32 = 3,
       /* flag, or-ed into a label type: every kernel reads the label as (value != 0): `y[y != 0] = 1` of the binary training loops
        * (model/modelVNet.py:576) on the device, so e.g. 0/255 mask images go to the loss / metric / clDice kernels as stored */
       SEG_LABEL_BINARIZE = 16 };
/* loss_name strings of model/modelVNet.py:68-76,513-521,750-756 */
enum {
    SEG_LOSS_BINARY_DICE = 0,     /* model/losses.py:33-53   BinaryDiceLoss */
    SEG_LOSS_BINARY_CE = 1,       /* model/losses.py:129-147 BinaryCrossEntropyLoss */
    SEG_LOSS_BINARY_FOCAL = 2,    /* model/losses.py:150-181 BinaryFocalLoss */
    SEG_LOSS_BINARY_CE_DICE = 3,  /* model/losses.py:184-197 BinaryCrossEntropyDiceLoss */
    SEG_LOSS_MULTI_CE = 4,        /* model/losses.py:247-260 MutilCrossEntropyLoss */
    SEG_LOSS_MULTI_FOCAL = 5,     /* model/losses.py:263-285 MutilFocalLoss */
    SEG_LOSS_MULTI_DICE = 6,      /* model/losses.py:288-325 MutilDiceLoss */
    /* not selectable through the reference wrappers' loss_name; same reduction, other ratios */
    SEG_LOSS_BINARY_JACCARD = 7,  /* model/losses.py:9-30    BinaryJaccardLoss */
    SEG_LOSS_BINARY_ELDICE = 8,   /* model/losses.py:56-74   BinaryELDiceLoss */
    SEG_LOSS_BINARY_TVERSKY = 9,  /* model/losses.py:102-126 BinaryTverskyLoss (alpha 0.3, beta 0.7) */
    SEG_LOSS_MULTI_CE_DICE = 10,  /* model/losses.py:328-342 MutilCrossEntropyDiceLoss */
    SEG_LOSS_MULTI_ELDICE = 11,   /* model/losses.py:345-382 MutilELDiceLoss */
    SEG_LOSS_BINARY_SS = 12,      /* model/losses.py:77-99   BinarySSLoss (sensitivity-specificity, r = 0.1) */
    SEG_LOSS_MULTI_TVERSKY = 13,  /* model/losses.py:421-459 MutilTverskyLoss: class_alpha = its alpha tensor (class weight AND false-positive weight);
                                     beta (never defined by the class; set by the caller) is passed in the focal_gamma argument */
    SEG_LOSS_MULTI_SS = 14,       /* model/losses.py:385-418 MutilSSLoss: r (never defined by the class) is passed in the focal_gamma argument */
    SEG_LOSS_BINARY_MCC = 15      /* model/losses.py:200-232 MCC_Loss: the `logits` argument holds PROBABILITIES (its `inputs`), dlogits = d loss / d inputs;
                                     torch.add(a, 1, b) read as a + 1*b (the torch 1.x signature the reference was written for) */
};
enum { SEG_MASKS_EVAL = 0, SEG_MASKS_GIVEN = 1, SEG_MASKS_RANDOM = 2 };

/* networks/VNet3d.py:109 VNet3d(image_channel, numclass, init_features=16), networks/VNet2d.py:109,
 * networks/Unet3d.py:11 UNet3d(in_channels, out_channels, init_features=16), networks/Unet2d.py:11.
 * ndim = 2 or 3.  in_channels 1..16 (1 in 3-D / 1..3 in 2-D take the fused image stem; more channels are zero-padded to a 16-channel image tensor
 * and run through the ordinary 16-channel convs), num_class 1..16, init_features 16. */
int seg_create(int net_kind, int ndim, int in_channels, int num_class, int init_features, int dtype,
               seg_handle* out);
void seg_destroy(seg_handle h);

/* Parameter table in reference state_dict order: name, PyTorch shape, and the
 * offset (in floats) of the tensor inside the flat fp32 parameter / gradient buffers. */
int seg_param_count(seg_handle h);
int seg_param_info(seg_handle h, int index, char* name, int name_cap, int* shape8, int* ndim,
                   long long* offset);
long long seg_param_numel(seg_handle h);

/* Number of channel-dropout calls per forward (34 VNet / 18 UNet) and the row stride of the
 * multiplier table [calls][N][ld] used by SEG_MASKS_GIVEN (networks/VNet3d.py:11,31,51,67). */
int seg_dropout_calls(seg_handle h);
int seg_dropout_ld(seg_handle h);
int seg_dropout_channels(seg_handle h, int call);

/* SEG_MASKS_RANDOM forwards issued so far (the mask of draw k is a function of (seed, k)); a resumed run restores it so that the mask
 * sequence continues instead of restarting (torch's generator state in the reference, model/modelVNet.py:570-596 under nn.Dropout3d). */
long long seg_dropout_draws(seg_handle h);
int seg_set_dropout_draws(seg_handle h, long long draws);

/* Fix the batch shape (N, D, H, W; D ignored for ndim 2) and size the workspace. */
int seg_plan(seg_handle h, int n, int d, int hgt, int wid);
long long seg_workspace_bytes(seg_handle h);
/* what the planner decided for the current shape (tests / diagnostics): what = 1: convolution units, 2: fork events the last backward pass recorded
 * on the caller's stream.  <0: not planned / unknown `what`. */
int seg_plan_count(seg_handle h, int what);

/* Bind caller-owned buffers: flat fp32 params / grads (seg_param_numel floats each) + workspace. */
int seg_bind(seg_handle h, float* params, float* grads, void* workspace);

/* Re-layout the fp32 master weights into the run-dtype GEMM layouts (call after every update). */
int seg_pack_weights(seg_handle h, void* stream);

/* forward(x) -> (logits, probs): x is fp32 NC[D]HW like the reference datasets deliver
 * (model/dataset.py:107), logits/probs are fp32 NC[D]HW (networks/VNet3d.py:90-99,129-158).
 * mask_mode: SEG_MASKS_EVAL = model.eval(); SEG_MASKS_GIVEN = `masks` holds the per-call (n,c)
 * dropout multipliers; SEG_MASKS_RANDOM = engine draws them from (seed, internal step). */
int seg_forward(seg_handle h, const float* x, int mask_mode, const float* masks,
                unsigned long long seed, float* logits, float* probs, void* stream);

/* backward of the last forward given d(loss)/d(logits) (fp32 NC[D]HW, already multiplied by
 * seg_get_loss_scale); parameter gradients (times the loss scale) are ACCUMULATED into the bound
 * flat gradient buffer (zero_grads != 0 clears it first — `opt.zero_grad()` of
 * model/modelVNet.py:593). */
int seg_backward(seg_handle h, const float* dlogits, int zero_grads, void* stream);
/* Bucketed gradient exchange (one process per GPU): the backward pass is a fixed list of seg_backward_ops() operations.
 * seg_backward_bucket finds the earliest op count `op_split` after which a suffix [param_offset, numel) of the flat gradient
 * buffer holding at least `tail_fraction` of the elements is final (gradients finish in reverse registration order);
 * seg_backward_range runs ops [op_begin, op_end) and joins the weight-gradient side stream, so the caller can start the
 * collective on that suffix while the remaining ops run.  seg_backward == seg_backward_range(0, seg_backward_ops()). */
int seg_backward_ops(seg_handle h);
int seg_backward_bucket(seg_handle h, double tail_fraction, int* op_split, long long* param_offset);
int seg_backward_range(seg_handle h, const float* dlogits, int zero_grads, int op_begin, int op_end, void* stream);
/* the same with join = 0: the slice's weight gradients are released to the engine's side stream but `stream` does not wait for them;
 * seg_side_wait(other_stream) makes ANOTHER stream (the one a collective is ordered after) wait for every weight gradient issued so far, so the
 * backward pass itself never stalls at a bucket boundary.  The last slice of a pass must be run with join = 1 (or through seg_backward_range). */
int seg_backward_slice(seg_handle h, const float* dlogits, int zero_grads, int op_begin, int op_end, int join, void* stream);
int seg_side_wait(seg_handle h, void* stream);

int seg_set_loss_scale(seg_handle h, float scale);
float seg_get_loss_scale(seg_handle h);

/* Losses and metrics on planar fp32 logits [N][C][V] (model/losses.py, model/metric.py:146-215).
 * out3 = {loss, dice metric, iou metric}.  `ws` needs seg_loss_ws_bytes(N, C) bytes; it carries
 * the reduction results from seg_loss_forward to seg_loss_backward. */
long long seg_loss_ws_bytes(int n, int c);
int seg_loss_forward(const float* logits, const void* target, int label_type, int n, int c,
                     long long v, int loss_kind, float focal_alpha, float focal_gamma,
                     const float* class_alpha, void* ws, float* out3, void* stream);
int seg_loss_backward(const float* logits, const void* target, int label_type, int n, int c,
                      long long v, int loss_kind, float focal_alpha, float focal_gamma, void* ws,
                      float grad_scale, float* dlogits, void* stream);
/* Exact global-batch losses across data-parallel ranks (SURVEY.md section 8e mode ii).  The reference losses are
 * batch-global ratios (model/losses.py:50-51 BinaryDiceLoss sums over the whole batch; :315-325 MutilDiceLoss reduces
 * over dim (0,2) and counts the classes present in the batch; :259 the CE / focal means divide by the batch voxel
 * count), so a rank-local loss is NOT the loss of the global batch.  seg_loss_forward == seg_loss_reduce followed by
 * seg_loss_finalize(n_global = n).  Across ranks: seg_loss_reduce, SUM-all-reduce the first seg_loss_shared_doubles()
 * doubles of `ws` (I, sum p, sum y, sum bce/nll, sum focal, per-class I_c / P_c / Y_c), seg_loss_finalize with
 * n_global = samples over all ranks (or 0: the count is read on the device - seg_loss_reduce leaves the local sample count in double
 * [5] of `ws`, so the all-reduce of the shared doubles delivers the global count and unequal shards need no host read),
 * seg_loss_backward; parameter gradients are then SUMMED over ranks (not averaged).
 * The metrics in out3[1..2] stay per-rank (they are per-sample means, model/metric.py:146-155). */
int seg_loss_shared_doubles(void);
int seg_loss_reduce(const float* logits, const void* target, int label_type, int n, int c, long long v,
                    int loss_kind, float focal_alpha, float focal_gamma, void* ws, void* stream);
int seg_loss_finalize(const float* logits, const void* target, int label_type, int n, int c, long long v,
                      int loss_kind, float focal_alpha, float focal_gamma, const float* class_alpha,
                      int n_global, void* ws, float* out3, void* stream);
/* Lovasz losses (model/lovasz.py:20-141 through model/losses.py:235-242 BinaryLovaszLoss and :462-473 LovaszLoss; per_image = False, no
 * ignore index).  x: fp32 [n][c][v].  c == 1: Lovasz hinge on logits, target in {0,1}.  c > 1: the reference's LovaszLoss, which hands
 * its `logits` argument to _lovasz_softmax as class probabilities WITHOUT a soft-max - x is used as given (pass probabilities for the
 * published Lovasz-Softmax); classes = 'present', mean over the present classes.  One radix sort of n*v (error, index) pairs per class.
 * out1[0] = loss; dx [n][c][v] = d loss / d x (written by the forward pass - the backward is a scaling by the incoming gradient).
 * ws: seg_lovasz_ws_bytes(n, v) bytes.  n*v < 2^32. */
long long seg_lovasz_ws_bytes(int n, long long v);
int seg_lovasz_forward(const float* x, const void* target, int label_type, int n, int c, long long v, void* ws, float* out1, float* dx,
                       void* stream);
/* SSIM / SSIM3D (model/lossesSSIM.py:47-99, 102-167): img1, img2 fp32 [n][c][d][h][w] (nd = 2: d = 1), Gaussian window `window` (11 in the
 * reference, sigma 1.5), zero padding, the same window for every channel.  out[0] = mean of the SSIM map (size_average=True),
 * out[1 .. n] = the per-sample means.  seg_ssim_forward leaves the derivative maps in ws (seg_ssim_ws_bytes(n, c, d*h*w) bytes) for
 * seg_ssim_backward: dimg = gscale * d(sum of the map)/d img, gscale[0] (per_sample = 0), gscale[sample] (per_sample = 1) or gscale[sample][x] (per_sample = row length w >= 2) carrying the
 * incoming gradient times 1/count; dimg1 or dimg2 may be NULL.  The backward pass consumes ws (one backward per forward).
 * Limits: n <= 64, window odd and <= 15. */
long long seg_ssim_ws_bytes(int n, int c, long long v);
int seg_ssim_forward(const float* img1, const float* img2, int n, int c, int d, int h, int w, int nd, int window, void* ws, float* out,
                     void* stream);
/* the same forward pass, additionally out_cols[n][w] = the map's means over (c, d, h): what `ssim3D(size_average=False)` returns in the reference (its
 * `.mean(1).mean(1).mean(1)` of a 5-D map, model/lossesSSIM.py:92-97, leaves (N, W)); seg_ssim_backward takes the matching gscale[n][w] with per_sample = w */
int seg_ssim_forward_cols(const float* img1, const float* img2, int n, int c, int d, int h, int w, int nd, int window, void* ws, float* out,
                          float* out_cols, void* stream);
int seg_ssim_backward(const float* img1, const float* img2, int n, int c, int d, int h, int w, int nd, int window, void* ws, const float* gscale,
                      int per_sample, float* dimg1, float* dimg2, void* stream);
/* predict() post-processing on the device (modelVNet.py:670-676): probs [N][C][V] fp32 -> uint8 mask [N][V];
 * C == 1: (p > threshold) * scale (scale 255 or 1); C > 1: first arg-max over the class axis. */
int seg_predict_mask(const float* probs, unsigned char* mask, int n, int c, long long v, float threshold, int scale, void* stream);
/* ---- pre/post-processing either side of predict, planar single-channel volumes [D][H][W].
 * seg_op_resample3d replaces the SimpleITK ResampleImageFilter calls of dataprocess/utils.py:99-145 (identity transform,
 * same origin/direction): output voxel i along an axis samples the input at continuous index i * step, step = output
 * spacing / input spacing (= originSize/newSize for resize_image_itkwithsize, newSpacing/originSpacing for
 * resize_image_itk).  mode 0 = sitkLinear (f32 volumes), 1 = sitkNearestNeighbor (f32 or u8); outside the input
 * buffer (index < -0.5 or >= size - 0.5) the value is 0, as ITK's default pixel.  elem_type 0 = f32, 1 = u8. */
int seg_op_resample3d(const void* src, void* dst, int elem_type, int sd, int sh, int sw, int dd, int dh, int dw,
                      double step_z, double step_y, double step_x, int mode, void* stream);
/* workspace for the two normalisations below (bytes) */
long long seg_op_normalize_ws_bytes(void);
/* ConvertitkTrunctedValue(image, upper, lower, 'meanstd') (dataprocess/utils.py:148-179): optional clip to
 * [lower, upper], then itk::NormalizeImageFilter: (x - mean) / sigma, sigma with the N-1 denominator. */
int seg_op_normalize_meanstd(const float* x, float* out, long long n, int clip, float lower, float upper, void* ws, void* stream);
/* normalize(slice, bottom=95, down=5) (dataprocess/utils.py:182-204): t, b = np.percentile(x, q_lo), np.percentile(x, q_hi)
 * (float32 'linear' method, exact order statistics); x = clip(x, t, b); z-score with the mean / population std of the
 * NON-ZERO clipped voxels; the clipped volume is returned unchanged when either std is 0. */
int seg_op_normalize_percentile(const float* x, float* out, long long n, float q_lo, float q_hi, void* ws, void* stream);
/* inference_patch (model/modelUnet.py:707-763): crop nb windows (origins = nb x {z,y,x} int32, device memory) of
 * pd x ph x pw voxels into a batch [nb][pd][ph][pw]; and the reverse: out[window] = 1 wherever the window's mask is
 * non-zero (out_mask += patch; out_mask[out_mask != 0] = 1).  `out` must be zero-initialised by the caller. */
int seg_op_gather_patches(const float* vol, int d, int h, int w, const int* origins, int nb, int pd, int ph, int pw, float* out, void* stream);
int seg_op_stitch_mask(const unsigned char* masks, const int* origins, int nb, int pd, int ph, int pw, unsigned char* out,
                       int d, int h, int w, void* stream);
/* dice_coeff / iou_coeff / multiclass_* on probabilities (model/metric.py:146-215): out2 = {dice, iou} */
int seg_metric(const float* probs, const void* target, int label_type, int n, int c, long long v,
               void* ws, float* out2, void* stream);

/* torch.optim.AdamW (model/modelVNet.py:548) / Adam (model/modelUnet.py:849) over flat buffers.
 * `state` = int[3] on the device: {step, found_inf, skipped}.  Gradients are multiplied by inv_scale first;
 * with check_finite the update is skipped (state[1] set for this call, state[2] += 1) when any gradient is inf/nan;
 * the caller reads state[2] now and then to back its loss scale off (no per-step host sync). */
int seg_adam_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq,
                  long long numel, float lr, float beta1, float beta2, float eps, float weight_decay,
                  int decoupled, float inv_scale, int check_finite, int* state, void* stream);

/* One whole optimisation step of the reference training loop (model/modelVNet.py:570-596: pred = model(x); loss = lossFunc(pred, y);
 * opt.zero_grad(); loss.backward(); opt.step(), plus the accuracy line :587) enqueued by ONE call:
 * [seg_pack_weights if !packed] -> seg_forward -> seg_loss_forward (out3 = {loss, dice, iou}) -> seg_loss_backward (times the loss
 * scale) -> seg_backward(zero_grads = 1) -> seg_adam_step on the bound params / grads -> seg_pack_weights.  Same launches as the
 * separate calls; the point is one FFI crossing per step.  The rank-local path only: gradient exchange, global-batch losses and the
 * clDice term go through the separate entry points.  grad_div: extra gradient divisor (1 on a single rank). */
typedef struct seg_train_args {
    const float* x; const void* target; int label_type;
    int loss_kind; float focal_alpha, focal_gamma; const float* class_alpha;
    float* logits; float* probs; float* dlogits; void* loss_ws; float* out3;
    int mask_mode; const float* masks; unsigned long long seed;
    float* exp_avg; float* exp_avg_sq; int* opt_state;
    float lr, beta1, beta2, eps, weight_decay; int decoupled; float grad_div; int check_finite;
    int packed;      /* != 0: the run-dtype weight layouts are current (the previous seg_train_step left them so) */
    /* ---- data-parallel exchange hooks (world > 1; all null / 0 for a rank-local step).  The library owns the sequencing of the step, the
     * caller owns the collectives (torch.distributed over RCCL / gloo): each hook is called synchronously from inside seg_train_step at the point
     * of the schedule where its exchange belongs and ENQUEUES the collective (it must not block the host on the device).
     * bucket_cb(user, index, offset, count): the gradient slice grads[offset, offset + count) is final on the device - enqueue its SUM
     *   all-reduce.  The finished slices are suffixes of the flat buffer (gradients finish in reverse registration order); `nfrac` boundaries
     *   `fractions[]` (finished fraction of the buffer, ascending) cut the backward pass into nfrac + 1 buckets (nfrac = 0: one exchange after the
     *   whole backward pass).  With `aux_stream` != null (GPU) the hook for every bucket but the last runs after `aux_stream` has been ordered
     *   behind the caller's stream AND the weight-gradient stream, and is expected to enqueue on `aux_stream`; the caller's stream never waits at
     *   a bucket boundary and meets `aux_stream` again before the last bucket.  Finally bucket_cb(user, -1, 0, 0): order the caller's stream
     *   behind every exchange (the optimiser follows).  Return != 0 aborts the step.
     * loss_cb(user, shared_sums, n_doubles): called between the loss reduction and its finalize with the rank's batch-global fp64 sums (device
     *   pointer): enqueue their SUM all-reduce on the caller's stream; returns the global sample count (0: take it from the exchanged sums,
     *   seg_loss_finalize) or < 0 to abort. */
    int (*bucket_cb)(void* user, int index, long long offset, long long count);
    long long (*loss_cb)(void* user, double* shared_sums, int n_doubles);
    void* cb_user;
    int nfrac; double fractions[4];
    void* aux_stream;
} seg_train_args;
int seg_train_step(seg_handle h, const seg_train_args* a, void* stream);
/* In-library gradient exchange - no host callback between the slices of the backward pass (the hook form above stays as the fallback, and is what
 * the CPU tests with gloo use).  `comm` = an ncclComm_t of RCCL (one rank per GPU; e.g. torch.distributed's: ProcessGroupNCCL._comm_ptr()),
 * `allreduce_fn` = the address of `ncclAllReduce` of the SAME RCCL library that created the communicator:
 *     ncclResult_t (*)(const void* sendbuf, void* recvbuf, size_t count, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t)
 * (the engine does not link RCCL: the caller hands it the function).  While a communicator is set, a seg_train_step with bucket_cb == NULL issues
 * ncclAllReduce(ncclFloat, ncclSum), in place, over every finished suffix of the flat gradient buffer (nfrac / fractions[] as above; nfrac = 0: one
 * all-reduce after the backward pass) on an exchange stream of the library that waits for the caller's stream and the weight-gradient stream (or on
 * a->aux_stream when given); the caller's stream meets it again in front of the optimiser.  a->grad_div carries the 1/world average.  comm == NULL
 * removes the communicator.  A step with a communicator cannot be captured as a HIP graph.  seg_backward_* / seg_adam_step are unaffected. */
int seg_set_rccl_comm(seg_handle h, void* comm, void* allreduce_fn);
/* The same step captured ONCE as a HIP graph (hipStreamBeginCapture around seg_train_step on `stream`, the weight-gradient stream forked and
 * joined inside) and replayed with one hipGraphLaunch per step: for hosts that cannot enqueue ~250 launches per step as fast as the GPU runs
 * them.  Every pointer / scalar of `a` and the current loss scale are baked into the graph (re-capture to change them; seg_plan, seg_bind and
 * seg_set_loss_scale drop the graph); the device-side dropout-draw and Adam step counters keep advancing, so replays are successive steps.
 * a->packed must be 1 (run one ordinary step first).  seg_train_graph_ready: 1 while a captured step exists. */
int seg_train_graph_capture(seg_handle h, const seg_train_args* a, void* stream);
int seg_train_graph_launch(seg_handle h, void* stream);
int seg_train_graph_ready(seg_handle h);

/* ---- operator-level entry points (what torch.nn.functional.conv3d / conv_transpose3d and their
 * autograd weight-gradients are to the reference: networks/VNet3d.py:8,28,29,49,65,70,88).  The
 * network-level calls above are built from exactly these launches; they are exported so each
 * kernel can be checked in isolation.  All tensors channels-last [N][D][H][W][C] in `dtype`. */
typedef struct seg_taps {
    int n;
    signed char d[27], h[27], w[27];
} seg_taps;

/* implicit-GEMM convolution: out[m][co] = sum_{tap,ci} in[vox(m,tap)][ci] * w[co][tap*Cin+ci] (+bias).
 * gather mode (scatter=0): rows m = output voxels (N,OD,OH,OW), input voxel = o*stride + tap.
 * scatter mode (scatter=1): rows m = input voxels, K = Cin, GEMM columns = (tap, co), column block
 * of tap a goes to fine voxel i*up + a (ConvTranspose k2 s2 / data-gradient of conv k2 s2).
 * The reduction input may be a virtual channel concat of in0 (C0 ch) and in1 (C1 ch). */
typedef struct seg_conv_args {
    const void* in0;
    const void* in1;
    int C0, C1;
    const void* w;      /* packed [Ngemm][Kpad] in dtype, zero padded */
    const float* bias;  /* [Cout] or null */
    void* out;
    double* stats;      /* [32][N][Cout][2] sum / sum of squares (+=), 32 replicas to be summed, or null */
    int N, ID, IH, IW;  /* gather source dims */
    int OD, OH, OW;     /* row-space dims */
    int FD, FH, FW;     /* scatter: fine output dims */
    int Cout, Ngemm, K, Kpad;
    int sd, sh, sw;     /* gather: input stride; scatter: up-sampling factor */
    int scatter;
    seg_taps taps;
    /* optional (null: off), gather form on the streaming kernel only (seg_op_conv_kernel == 1): in0 holds the RAW output r of a conv + GroupNorm unit and is read as
     * a = relu(act_scale[n][c] * r + act_shift[n][c]) rounded to dtype - GroupNorm + channel dropout + ReLU of the producer (networks/VNet3d.py:72-74) applied on
     * load, so that the activated tensor is never written; [N][C0] fp32 each */
    const float* act_scale;
    const float* act_shift;
    /* optional (null: off), gather form on the streaming kernel only, no bias / stats: GEMM columns >= Cout0 (a multiple of 16) are written to out1 as rows of
     * Ngemm - Cout0 channels, columns < Cout0 to out as rows of Cout0 channels - the data-gradients of BOTH sources of a virtual concat from one pass over d(raw) */
    void* out1;
    int Cout0;
    /* optional (null: off), with out1 only: `out` (the first Cout0 columns) is the gradient dz of an activation a = relu(rq_scale * r + rq_shift) whose raw tensor
     * r is rq_r ([rows][Cout0] in dtype); the launch adds that unit's GroupNorm-backward sums - sum dz*[a > 0] and sum dz*[a > 0]*r per (sample, channel) - to
     * rq_Q ([32][N][Cout0][2] fp64, replicas to be summed): the reduction pass over (dz, r) that would follow is not needed */
    const void* rq_r;
    const float* rq_scale;
    const float* rq_shift;
    double* rq_Q;
} seg_conv_args;
int seg_op_conv(const seg_conv_args* a, int dtype, void* stream);
/* which kernel seg_op_conv picks for these extents: 1 = register-resident streaming kernel (short reductions on large
 * volumes), 0 = LDS-staged implicit GEMM */
int seg_op_conv_kernel(const seg_conv_args* a);

/* weight gradient dW[p][tap][q] += sum_m dR[m][p] * X[vox(m,tap)][q], written to
 * dw[p*sP + q*sQ + tap*sT] (fp32) through per-slice partial tiles in `partial_scratch`
 * (seg_op_wgrad_partial_bytes bytes; deterministic reduction).  stem=1: X is [N][V][C0] with tiny C0 and q enumerates (tap,ci). */
typedef struct seg_wgrad_args {
    const void* dr;
    const void* x0;
    const void* x1;
    int C0, C1;
    float* dw;
    int P, Q;
    int N, ID, IH, IW, OD, OH, OW;
    int sd, sh, sw;
    seg_taps taps;
    long long sP, sQ, sT;
    int stem;
    /* optional (null: off), 1^d stride-1 convs on 16-bit tensors only (seg_op_wgrad fails otherwise): x0 holds the raw output of a conv + GroupNorm unit and is
     * read as relu(act_scale[n][c] * r + act_shift[n][c]) rounded to dtype, like seg_conv_args.act_scale; [N][C0] fp32 each, C0 <= 64 */
    const float* act_scale;
    const float* act_shift;
} seg_wgrad_args;
long long seg_op_wgrad_partial_bytes(const seg_wgrad_args* a);
int seg_op_wgrad(const seg_wgrad_args* a, float* partial_scratch, int dtype, void* stream);

/* weight re-layout: dst[r1][r2][t][c] (dtype, row length Kpad, zero padded) = src[r1*s1+r2*s2+t'*sT+c*sC],
 * t' = flipT ? T-1-t : t.  `descs` is a DEVICE array. */
typedef struct seg_pack_desc {
    const float* src;
    void* dst;
    int R1, R2, T, Cc;
    int Kpad;
    long long s1, s2, sT, sC;
    int flipT;
    int csrc;   /* channels the SOURCE holds per tap (0: Cc).  csrc < Cc: channels csrc .. Cc-1 of the packed layout are zeros - the image convs of a
                   multi-channel 3-D input, whose image tensor is zero-padded to 16 channels so that they run as ordinary 16-channel halo convs */
    int frag;   /* 0: rows [R1*R2][Kpad]; 1: MFMA-fragment-major [Cc/32][T][rows/16][64 lanes][8] (Cc % 32 == 0, rows % 16 == 0):
                   lane = 16*((c%32)/8) + row%16 holds k = c%8 .. — one 16x32 B fragment is one contiguous 1 KB line;
                   2: the same over the flat k = t*Cc + c axis, [Kpad/32][rows/16][64 lanes][8] (Cc == 16: two taps per step);
                   3: Cc == 16, T == 27, Kpad == 480: [15 steps = kh*5 + pair][rows/16][64 lanes][8], the two taps of a step differ in kd / kw only -
                      pairs 0..2: (kd = pair, kw = 0 | 1), 3: (kd = 0 | 1, kw = 2), 4: (kd = 2, kw = 2 | zeros) (seg_op_conv3x_cfg_frag) */
} seg_pack_desc;
int seg_op_pack(const seg_pack_desc* descs, int ndesc, long long max_elems, int dtype, void* stream);
/* LDS halo-tile kernels for the 3^d (ndim 3) / 3^2 (ndim 2, D = 1) stride-1 pad-1 convolutions.
 * (stats, when given, is [32][N][Cout][2] like seg_conv_args.stats.)
 * seg_op_conv3: out[N][D][H][W][Cout] = conv(in[N][D][H][W][Cin], w packed [Cout][Kpad], k = (tap, ci)) + bias,
 * optional GroupNorm partial sums; with the "conv_dgrad" weight layout it is the data-gradient.
 * seg_op_wgrad3: dw[p][q][tap] += sum_m dr[m][p] * x[m + tap][q]; `partial` needs
 * seg_op_wgrad3_partial_bytes(...) bytes of scratch (deterministic two-stage reduction). */
int seg_op_conv3(const void* in, const void* w, const float* bias, void* out, double* stats, int n, int d, int h,
                 int wid, int cin, int cout, int ndim, int dtype, void* stream);
long long seg_op_wgrad3_partial_bytes(int ndim, int n, int d, int h, int wid, int p, int q);
int seg_op_wgrad3(const void* dr, const void* x, float* partial, float* dw, int n, int d, int h, int wid, int p,
                  int q, int ndim, int dtype, void* stream);
/* seg_op_wgrad3 with x a virtual channel concat: channels [0, c0) of x come from x0 ([..][c0]), the rest from x1 ([..][q - c0])
 * (the UNet decoder blocks read cat(up, skip), networks/Unet3d.py:50-62, without materialising it). */
int seg_op_wgrad3_cat(const void* dr, const void* x0, const void* x1, int c0, float* partial, float* dw, int n, int d, int h,
                      int wid, int p, int q, int ndim, int dtype, void* stream);
/* Fused input block (csrc/stemx.hip): y = relu(drop(GN(conv3(x)))) [+ relu(drop(GN(conv1(x))))] for a 1..3-channel image and 16
 * output channels (InputTransition of networks/VNet3d.py:25-43 with its shared GroupNorm module applied twice; first conv of
 * networks/Unet3d.py:64-86 with w1 = NULL).  The raw conv outputs are recomputed in every pass instead of being stored.
 * mode 0: GroupNorm partial sums of both branches -> stats3 / stats1 ([32][N][16][2] fp64, accumulated);
 * mode 1: out[N][D][H][W][16] from scale / shift ([N][16], dropout multiplier folded in);
 * mode 2: Q3 / Q1 ([32][N][16][2]: sum dz, sum dz*r with dz = (sum of the ndy gradient sources) * [scale*r + shift > 0]);
 * mode 3: weight gradients: d(raw) = coef[0]*dz + coef[1]*r + coef[2] (coef [N][16][3]) times the im2col rows, accumulated
 *         into dw3 (16, Cimg, 3^d) / dw1 (16, Cimg, 1^d) through `partial` (seg_op_stemx_partial_bytes of scratch).
 * w3 / w1: run-dtype [16][32] rows, k = tap*Cimg + ci / k = ci (seg_op_pack "conv_fwd" layout).  img: run dtype, channels-last. */
typedef struct seg_stemx_args {
    const void* img; const void* w3; const void* w1;
    const float* bias3; const float* bias1;
    double* stats3; double* stats1;
    const float* scale3; const float* shift3; const float* scale1; const float* shift1;
    void* out;
    const void* dy[3]; int ndy;
    double* Q3; double* Q1;
    const float* coef3; const float* coef1;
    float* partial;
    int N, D, H, W, Cimg;
} seg_stemx_args;
long long seg_op_stemx_partial_bytes(int ndim, int n, int d, int h, int wid, int cimg);
int seg_op_stemx(const seg_stemx_args* a, int mode, int ndim, int dtype, float* dw3, float* dw1, void* stream);
/* Register-blocked halo conv for 16-bit tensors with Cin % 32 == 0 or Cin == 16 (csrc/conv3x.hip): same operator as seg_op_conv3, the
 * weights packed with seg_pack_desc.frag = seg_op_conv3x_cfg_frag(cfg) (1; Cin == 16: 2 or 3) ("conv_fwd" / "conv_dgrad" element order), `in1` an optional second source of a
 * virtual channel concat (channels c0..cin-1).  cfg selects a tiling (seg_op_conv3x_cfg_info enumerates them; -1 = the
 * engine's default for the shape).  Returns <0 when the tiling does not fit the shape. */
int seg_op_conv3x(int cfg, const void* in0, const void* in1, int c0, const void* w, const float* bias, void* out, double* stats,
                  int n, int d, int h, int wid, int cin, int cout, int ndim, int dtype, void* stream);
int seg_op_conv3x_num_cfgs(void);
/* index in [0, num_cfgs): tiling id, ndim, box {d,h,w}, output channels per workgroup, resident 32-channel chunks, description */
int seg_op_conv3x_cfg_info(int index, int* id, int* ndim, int* box3, int* bn, int* nres, char* name, int name_cap);
int seg_op_conv3x_default_cfg(int ndim, int n, int d, int h, int wid, int cin, int cout, int dtype);
/* seg_pack_desc.frag of the weights tiling `cfg` reads: 1, 2 (Cin == 16) or 3 (Cin == 16, 3-D tilings 28 .. 31: the halo fragments are reused across
 * the kh taps, csrc/conv3x_impl.h conv3x16r_kernel); 0 for an unknown id */
int seg_op_conv3x_cfg_frag(int cfg);
/* sizeof of the structs of this header as compiled into the library: 0 conv, 1 wgrad, 2 pack, 3 stemx, 4 train */
int seg_abi_sizeof(int which);

/* ---- soft-clDice building blocks (model/lossescldice.py:5-59; corrected restatement).
 * Planar fp32 tensors [planes][D][H][W]; nd = 3 pools 3x3x3 over (D,H,W), nd = 2 pools 3x3 over (H,W); stride 1, pad 1,
 * padding ignored.  seg_op_skel_update: out = relu(x - relu(maxpool(e) - e)).  The *_bwd entry points route gradients to the
 * first extremum of each window (ATen max_pool backward); `de` / `din` are accumulated into (zero them first). */
int seg_op_pool3(const float* x, float* out, int planes, int d, int h, int w, int nd, int is_min, void* stream);
/* one whole skeleton iteration (e_out = minpool(x), x_out = relu(x - relu(maxpool(e_out) - e_out))) in one pass over LDS tiles;
 * bit-identical to seg_op_pool3(is_min) + seg_op_skel_update */
int seg_op_skel_iter(const float* x, float* e_out, float* x_out, int planes, int d, int h, int w, int nd, void* stream);
/* backward of seg_op_skel_iter: dx = d loss / d x given g = d loss / d x_out (x, e as saved by the forward); two gather passes
 * over LDS tiles - deterministic, no atomics; `de_scratch` is a tensor-sized fp32 work buffer (need not be cleared) */
int seg_op_skel_iter_bwd(const float* g, const float* x, const float* e, float* dx, float* de_scratch, int planes, int d, int h, int w,
                         int nd, void* stream);
int seg_op_skel_update(const float* x, const float* e, float* out, int planes, int d, int h, int w, int nd, void* stream);
int seg_op_skel_update_bwd(const float* g, const float* x, const float* e, float* dx, float* de, int planes, int d, int h, int w,
                           int nd, void* stream);
int seg_op_pool3_bwd(const float* src, const float* dout, float* din, int planes, int d, int h, int w, int nd, int is_min,
                     void* stream);
/* out2[p] = {sum a*b, sum a} per plane (fp64; `scratch` = seg_op_plane_dot_scratch_bytes bytes of per-workgroup partials);
 * out = a[p]*in + b[p] (accumulate != 0: +=) */
long long seg_op_plane_dot_scratch_bytes(int planes, long long v);
int seg_op_plane_dot(const float* a, const float* b, double* out2, double* scratch, int planes, long long v, void* stream);
int seg_op_plane_axpb(const float* in, const float* a, const float* b, float* out, int planes, long long v, int accumulate,
                      void* stream);

/* Binary soft-clDice as a loss of the engine (reference: model/lossescldice.py:37-59 Binary_Soft_cldice_loss, with the repairs listed in
 * oracle/make_golden.py:CLDICE_REPAIRS; `width` = skeleton iterations, 10 in the reference).  probs = the head's probabilities
 * [n][1][d][h][w] fp32, target = labels of the same extent.  out1[0] = the clDice loss; when dlogits != NULL,
 * grad_scale * d loss / d logit (sigmoid Jacobian included) is ADDED to dlogits, i.e. call it after seg_loss_backward of the
 * companion loss (Dice) with grad_scale = weight * loss scale.  ws: seg_cldice_ws_bytes() bytes planned by the caller; no
 * allocation, no host synchronisation.  nd = 2 (d = 1) or 3.
 * The target is ALWAYS read as the binary mask (label != 0), whatever `label_type` says (SEG_LABEL_BINARIZE is implied): that is what the
 * reference's train loop hands to every loss (model/modelVNet.py:576 binarises the labels first), and it lets the target's skeleton be computed on a
 * bit image.  A caller with soft (fractional) targets - which Binary_Soft_cldice_loss.forward itself would skeletonise as given - must use the
 * building blocks above (seg_op_skel_iter, seg_op_plane_dot, ...); the companion loss keeps reading the labels as `label_type` says. */
long long seg_cldice_ws_bytes(int n, int d, int h, int w, int nd, int width);
/* optional: the label-only part (float labels + the target skeleton) into ws ahead of time, e.g. on another stream while the forward pass
 * runs; seg_cldice_binary(target_ready = 1) then skips it (the caller orders the two calls with an event). */
int seg_cldice_target(const void* target, int label_type, int n, int d, int h, int w, int nd, int width, void* ws, void* stream);
int seg_cldice_binary(const float* probs, const void* target, int label_type, int n, int d, int h, int w, int nd, int width,
                      float grad_scale, void* ws, float* out1, float* dlogits, int target_ready, void* stream);

/* ---- measurement: HIP-event timing of kernel classes inside a running forward/backward.
 * seg_profile_enable(h, mask): from now on every launch whose class bit is set in `mask` is
 * bracketed by hipEventRecord on the launch stream (0 disables).  seg_profile_read(h, ...)
 * synchronises the recorded events, returns per class {launch count, total ms, algorithmic bytes,
 * algorithmic flops} accumulated since the last read, and clears the records. */
enum {
    SEG_K_CONV3 = 0,          /* halo-tile 3^d conv, forward + data-gradient, big-box tiling of the wide levels with >= 32 channels */
    SEG_K_WGRAD3 = 1,         /* halo-tile weight gradient (main kernel + partial reduce) */
    SEG_K_CONV_GENERIC = 2,   /* gather / scatter implicit GEMM */
    SEG_K_WGRAD_GENERIC = 3,
    SEG_K_STEM = 4,
    SEG_K_GN_ACT = 5,
    SEG_K_GN_BWD_REDUCE = 6,
    SEG_K_GN_BWD_APPLY = 7,
    SEG_K_HEAD = 8,
    SEG_K_CONV3_SB = 9,       /* every other halo-tile conv (16-channel top level, deep levels) */
    SEG_K_GN_GROUP = 10,      /* one-launch GroupNorm passes of the small tensors (forward and backward) */
    SEG_K_MISC = 11,          /* everything else of a train step: fill + ingest, loss reduce / finalize / backward, fused optimiser, weight re-pack */
    SEG_K_COUNT = 12
};
int seg_profile_enable(seg_handle h, unsigned mask);
int seg_profile_read(seg_handle h, int* calls, float* ms, double* bytes, double* flops);

const char* seg_last_error(void);
const char* seg_build_info(void);

#ifdef __cplusplus
}
#endif
#endif
