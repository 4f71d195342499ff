/*
 * ctl_hip.h -- C-ABI of libctl_hip.so: the MI355X (gfx950) kernels behind the cooperative-training hot path.
 *
 * The reference (cherise215/Cooperative_Training_and_Latent_Space_Data_Augmentation) is pure PyTorch-eager and has
 * no FFI of its own; every entry point below replaces the ATen ops that the cited reference lines dispatch.
 * "model.py" = medseg/models/advanced_triplet_recon_segmentation_model.py,
 * "encdec.py" = medseg/models/ebm/encoder_decoder.py, "util.py" = medseg/models/model_util.py.
 *
 * Conventions
 *   - all tensors are fp32 NHWC ("channels_last") in device memory owned by the caller (bf16 where a ctl_conv.dt / op flag says so:
 *     the `float*` in those signatures is then the base address of bf16 data); labels are int64 NHW
 *   - every call only enqueues work on `stream`; no allocation, no host sync, no retained pointers
 *   - return 0 on success, negative ctl_status on error; ctl_last_error() gives a thread-local message
 *   - scratch / partial-sum buffers are sized by the *_ws_* helpers and passed in by the caller
 */
#ifndef CTL_HIP_H
#define CTL_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* ctl_stream;            /* hipStream_t */

enum ctl_status { CTL_OK = 0, CTL_EINVAL = -1, CTL_EUNSUPPORTED = -2, CTL_ELAUNCH = -3 };

/* ABI version: bumped whenever a struct layout, an argument list or a plan-op slot assignment changes.  Bindings must compare
 * ctl_version() with the CTL_ABI_VERSION they were written against (and ctl_sizeof_conv / ctl_sizeof_op with their struct sizes)
 * at load time and refuse to run on a mismatch.  3 = round 3: fused-finalize entry points and side lanes removed, `ds` argument of
 * ctl_bwd_reduce_dt, ctl_red_blocks().  4 = CTL_EPI_TAILBWD / ctl_conv_forward_ex (plan op CONV slot 10).
 * 5 = ctl_bn_finalize_ex (save_uvar, plan op BN_FINALIZE slot 10), ctl_bn_replay_running / CTL_OP_BN_REPLAY.
 * 6 = the BatchNorm-backward prologue: ctl_conv.pro_affine == 2 + the x2 argument of ctl_conv_forward_ex (plan op CONV slot 11),
 *     ctl_conv_wgrad_ex (plan op WGRAD slots 6, 7); CTL_EPI_TAILBWD in the bf16 family.
 * 7 = the `pool` argument of ctl_conv_forward_ex (plan op CONV slot 12; CTL_OP_MAX_T 12 -> 14: sizeof(ctl_op) 304 -> 328), ctl_conv_pool_ok.
 * 8 = the `xout` argument of ctl_conv_forward_ex (plan op CONV slot 13).
 * 9 = CTL_DT_X3 / CTL_PACK_X3, ctl_conv_wpack_floats_x3, ctl_pack_weights_x3_batched; plan op PACK_BATCH i[1] is a bit mask.
 * 10 = grouped weight gradients: ctl_wgrad_group_class / ctl_wgrad_group_plan / ctl_conv_wgrad_group, plan op CTL_OP_WGRAD_GROUP.
 * 11 = ctl_bn_bwd_finalize_ex (per-group gamma / beta gradient switch; plan op BN_BWD_FINALIZE i[4]); a WGRAD record with i[24] != 0 is refused
 *      outside its WGRAD_GROUP. */
#define CTL_ABI_VERSION 11
int         ctl_version(void);
const char* ctl_last_error(void);

/* ------------------------------------------------------------------------------------------------ convolution
 * One implicit-GEMM kernel family (MFMA f32 16x16x4, LDS-staged NHWC input tiles) serves
 *   nn.Conv2d 3x3 s1/s2 p1 and 1x1          encdec.py:40-55, 323-335, 371-376, 390-391, 439-440, 469-474
 *   nn.ConvTranspose2d k2 s2                 encdec.py:302            (4 scattered 1x1 problems, `nsub` = 4)
 *   nn.UpsamplingNearest2d folded into the consumer conv's input indexing   encdec.py:294-296 (in_mode 1)
 *   every dgrad (conv over dy with transposed/flipped weights; stride-2 dgrad via zero-insertion, in_mode 2)
 * with BatchNorm-apply + LeakyReLU fused into the input staging ("prologue") and bias / residual / activation /
 * BatchNorm statistics fused into the epilogue.
 */
enum { CTL_IN_PLAIN = 0, CTL_IN_UP2 = 1, CTL_IN_ZINS2 = 2, CTL_IN_C4 = 3 /* plain input with <= 4 channels, 3x3 stride 1: the taps are
       K-packed (weights from ctl_pack_weights_batched mode 4); 12 MFMAs per pixel tile instead of 36 */ };
enum { CTL_ACT_NONE = 0, CTL_ACT_LEAKY = 1, CTL_ACT_SIGMOID = 2 };
enum { CTL_EPI_BIAS = 1, CTL_EPI_ACCUM = 2, CTL_EPI_RES = 4, CTL_EPI_STATS = 8, CTL_EPI_BNBWD = 16, CTL_EPI_TAILBWD = 32 };

typedef struct ctl_conv {
    int32_t n, hin, win, cin;        /* stored input tensor [n,hin,win,cin]                                   */
    int32_t hout, wout, cout;        /* output pixel grid of this problem and its channel count               */
    int32_t ks, stride, pad;         /* 3/1|2/1, 1/1/0, 2/2/0                                                  */
    int32_t in_mode;                 /* CTL_IN_*: virtual input = stored | nearest-up x2 | zero-insert x2     */
    int32_t pro_affine;              /* 1: x <- leaky(x*pro_scale[c]+pro_shift[c], pro_slope) while staging;
                                        needs max(groups,1)*cin <= 256 (the coefficients sit in an LDS table)   */
    float   pro_slope;
    int32_t epi_flags;               /* CTL_EPI_*                                                             */
    int32_t epi_act;                 /* CTL_ACT_* applied after bias/residual                                  */
    float   epi_slope;
    int32_t out_h, out_w;            /* output tensor [n,out_h,out_w,cout]; pixel (ho,wo) of sub-problem z     */
    int32_t out_sy, out_sx;          /*   lands at (ho*out_sy + z/2*out_sub, wo*out_sx + z%2*out_sub)          */
    int32_t nsub, out_sub;           /* nsub = 1 (plain) or 4 (ConvTranspose k2s2: out_sy=out_sx=2,out_sub=1)  */
    int32_t groups;                  /* BatchNorm groups along n (0/1 = one): images [g*n/groups, (g+1)*n/groups) use row g of
                                        pro_scale/pro_shift/res_scale/res_shift ([groups][c]) and get their own statistics
                                        partials -- several independent passes of one network batched into one launch   */
    int32_t dt;                      /* CTL_DT_* : 0 = everything fp32 (BASELINE config 2).  CTL_DT_BF16 selects the bf16 kernel family
                                        (v_mfma_f32_16x16x32_bf16: operands rounded to bf16 AFTER the fp32 prologue, fp32 accumulate,
                                        fp32 bias / BatchNorm statistics / epilogue; weights packed as bf16 by the *_batched pack with
                                        the same flag); CTL_DT_X16 / _Y16 / _RES16: that tensor is STORED as bf16 (activation storage
                                        of BASELINE config 3) -- network inputs / outputs stay fp32.
                                        CTL_DT_X3 (alone; fp32 tensors; cin a multiple of 16, cout a multiple of 16 or 4 / 8 / 12 (the
                                        weight gradient: both multiples of 16); 2x2 / 3x3 / 4x4 kernels): the SAME
                                        fp32 computation with the contraction on the bf16 matrix pipe -- every operand is split
                                        exactly into three bf16 numbers while it is staged, six v_mfma_f32_16x16x32_bf16 per
                                        contraction step (hi*hi + hi*mid + mid*hi + mid*mid + hi*lo + lo*hi, fp32 accumulate): error
                                        per product <= 2^-24 (1 + 2^-8) worst case (the rounding of one fp32 multiply), typically 2^-25; 16/6 of the fp32 MFMA rate.
                                        Weights come from ctl_pack_weights_x3_batched (records with CTL_PACK_X3)                    */
} ctl_conv;
enum { CTL_DT_BF16 = 1, CTL_DT_X16 = 2, CTL_DT_Y16 = 4, CTL_DT_RES16 = 8, CTL_DT_X3 = 16 };
enum { CTL_PACK_X3 = 16 };          /* or-ed into the mode word of a pack record: three-plane bf16 fragments for CTL_DT_X3 launches */

/* number of floats of the packed weight buffer for one sub-problem, and of the statistics partial buffer */
size_t ctl_conv_wpack_floats(int32_t cin, int32_t cout, int32_t ks);
size_t ctl_conv_stats_floats(const ctl_conv* d);     /* [groups][blocks][2][cout] */
int    ctl_conv_stats_blocks(const ctl_conv* d);

/* Pack weights into MFMA-fragment order.  Element (co,ci,kh,kw) of the *effective* conv is read from
 * src[co*s_co + ci*s_ci + kh'*s_kh + kw'*s_kw] with (kh',kw') = flip ? (ks-1-kh, ks-1-kw) : (kh,kw).  Covers OIHW
 * forward weights, their dgrad transposes and both ConvTranspose2d uses. */
int ctl_pack_weights(const float* src, float* dst, int32_t cout, int32_t cin, int32_t ks,
                     int64_t s_co, int64_t s_ci, int64_t s_kh, int64_t s_kw, int32_t flip, ctl_stream stream);

/* y = epi( conv(pro(x)) ).  res/res_scale/res_shift: CTL_EPI_RES adds res*res_scale[c]+res_shift[c] (the
 * BatchNorm'ed main branch of res_convdown / res_up_family, encdec.py:64,344).  stats_partial: CTL_EPI_STATS.
 * CTL_EPI_BNBWD (with CTL_EPI_STATS; a data-gradient conv whose result is dL/da of a = leaky(BN(u), epi_slope)): res = u,
 * res_scale/res_shift = the BatchNorm coefficients; y = g = conv * leaky'(u*scale+shift) and stats_partial receives
 * (sum g, sum g*u) -- the reduction pass of the BatchNorm backward (ctl_bwd_reduce mode 1) folded into the producer. */
int ctl_conv_forward(const ctl_conv* d, const float* x, const float* wpack, const float* bias,
                     const float* pro_scale, const float* pro_shift,
                     const float* res, const float* res_scale, const float* res_shift,
                     float* y, float* stats_partial, ctl_stream stream);
/* ... with a second epilogue tensor.  CTL_EPI_TAILBWD (with CTL_EPI_STATS, optionally CTL_EPI_ACCUM; cout % 16 == 0; fp32, or the bf16
 * family with bf16-stored y / res / res2 -- there the sums come from the unrounded g and dOut is never rounded on its own; the 1x1 / 2x2 /
 * zero-insert 3x3 launches that write the output gradient of a residual block, encdec.py:64,344): the conv result (+ y with
 * CTL_EPI_ACCUM) is dL/dOut of out = leaky(S + BN(v), epi_slope); res = out, res2 = v.  y receives g = dOut * leaky'(out) and
 * stats_partial (sum g, sum g*v) per BatchNorm group: the reduction pass of the residual tail (ctl_bwd_reduce mode 0) folded into the
 * producer of dOut, which is then never materialised.  res2 == NULL: plain ctl_conv_forward. */
/* x2 (with ctl_conv.pro_affine == 2; BOTH families -- fp32 tensors, or the bf16 family with bf16-stored x / x2 / y; cin % 16 == 0,
 * plain (CTL_IN_PLAIN) 3x3 stride-1 or 4x4 stride-2 conv, groups * cin <= 256, not together with CTL_EPI_TAILBWD): the
 * BatchNorm-backward prologue.  The conv input is the VIRTUAL tensor
 *     A[c] * x + B[c] * x2 + C[c]     (zero outside the image; bf16 family: rounded to bf16 as the stored tensor would have been)
 * with pro_scale = the [group][A | B | C][cin] coefficients ctl_bn_bwd_finalize writes (pro_shift is not read): x = g = dL/da * leaky',
 * x2 = the BatchNorm input.  This is ctl_bwd_apply (mode 2) run inside the staging of its consumer: the data-gradient convs of a
 * residual block read (g, u) instead of dU, which is never written.  x2 == NULL and pro_affine <= 1: as before. */
int ctl_conv_forward_ex(const ctl_conv* d, const float* x, const float* wpack, const float* bias,
                        const float* pro_scale, const float* pro_shift,
                        const float* res, const float* res_scale, const float* res_shift, const float* res2, const float* x2,
                        float* y, float* stats_partial, float* pool, float* xout, ctl_stream stream);
/* pool (with CTL_EPI_TAILBWD on a 1x1 conv with even output sizes; ctl_conv_pool_ok(d) says whether d's tile configuration can do it):
 * the epilogue also writes the 2x2 sum-pool of g, [n, out_h/2, out_w/2, cout] -- the input of the consuming block's 1x1 weight / data
 * gradients behind a nearest-neighbour up-sampling (encdec.py:344): the stand-alone ctl_sumpool2 pass disappears.  fp32: the same
 * association as ctl_sumpool2, bit-identical; bf16: pooled from the unrounded g, rounded once.  NULL: not written. */
int ctl_conv_pool_ok(const ctl_conv* d);
/* xout (with pro_affine == 2; a tensor of x's geometry and storage type): the conv also WRITES the virtual input A*x + B*x2 + C it stages --
 * every input pixel by the tile that owns it, from the blocks of the first output-channel group -- so that the weight-gradient kernel of
 * the same layer reads it as a plain output gradient (ctl_conv_wgrad) instead of evaluating it again in each of its cin-chunk blocks
 * (ctl_conv_wgrad_ex: +10-18 % fp32, +14-44 % bf16 in isolation).  Issue this conv BEFORE that weight gradient.  NULL: not written. */

/* Weight gradient of the conv described by d (x [n,hin,win,cin] -> dy [n,hout,wout,cout], nsub must be 1):
 * partial[split][tap][cin16][cout16] (+ bias partial[split][cout16]); then ctl_wgrad_reduce sums the splits and
 * (accumulate ? += : =) into a gradient tensor with the same generic strides as ctl_pack_weights. */
int    ctl_wgrad_splits(const ctl_conv* d);
size_t ctl_wgrad_partial_floats(const ctl_conv* d);          /* weights part  */
size_t ctl_wgrad_bias_partial_floats(const ctl_conv* d);     /* bias part     */
int ctl_conv_wgrad(const ctl_conv* d, const float* x, const float* pro_scale, const float* pro_shift,
                   const float* dy, float* w_partial, float* b_partial, ctl_stream stream);
/* ... whose output gradient is the virtual BatchNorm-backward result  A[c] * dy + B[c] * dy2 + C[c]  (dy_coef = [group][A | B | C][cout]
 * from ctl_bn_bwd_finalize; BOTH families -- fp32 dy / dy2, or the bf16 family with bf16-stored dy / dy2; 3x3 stride-1 convs,
 * cout % 16 == 0, groups * cout <= 256): the other
 * consumer of a block's dU / dV (see ctl_conv_forward_ex).  The bias gradient is the sum of the virtual tensor.  dy2 == NULL: ctl_conv_wgrad. */
int ctl_conv_wgrad_ex(const ctl_conv* d, const float* x, const float* pro_scale, const float* pro_shift,
                      const float* dy, const float* dy2, const float* dy_coef, float* w_partial, float* b_partial, ctl_stream stream);

/* Grouped weight gradients (ABI 10).  The reference computes every layer's dW inside autograd's backward sweep, one cuDNN/MKL call per layer
 * (encoder_decoder.py:19-68, 285-348 under loss.backward(), train_adv_supervised_segmentation_triplet.py:225).  Nothing downstream of a backward
 * pass reads dW before the optimizer (or the gradient all-reduce), so the plan compiler defers the X3 weight gradients of a pass and serves up to
 * 8 of one class with ONE launch: the CUs are dealt to the members in proportion to their work.  Per launch ~15 us are fixed (launch, exposed first
 * loads, reduction tail) against ~25 us of work for an n = 16 layer; a member of a group gets fewer CUs, more tiles per block, fewer split-K partials.
 *   ctl_wgrad_group_class: >= 0 (the class: members of one launch must agree) if `d` (CTL_DT_X3, 3x3 stride 1, plain / nearest-up-sampled input,
 *                          cin and cout multiples of 32, hout >= 8) can ride in a group, -1 otherwise.  has_dy2: the two-tensor output gradient.
 *   ctl_wgrad_group_plan : the members' pixel splits; member i then needs splits[i] * 9 * cin * cout floats of w_partial (and splits[i] * cout of
 *                          b_partial), laid out and reduced exactly like ctl_conv_wgrad's ([split][tap][cin][cout]; ctl_wgrad_reduce_batched).
 *   ctl_conv_wgrad_group : the launch.  Arrays of n pointers; pro_scale / pro_shift / dy2 / dy_coef / b_partial entries (or the arrays) may be NULL
 *                          where a member has none.
 * The bf16 family (CTL_DT_BF16) rides the same three entry points: class = 0x100 | the kernel instantiation its dispatch ends in (any bf16 weight
 * gradient has one), the launch stacks the members' own grids along blockIdx.x, ctl_wgrad_group_plan deals the resident blocks in proportion to
 * the work (never more splits than a launch of its own); with splits[i] = ctl_wgrad_splits(d_i) a member's partial sums are bit for bit those of
 * ctl_conv_wgrad_ex (tests/test_bf16_gpu.py). */
int ctl_wgrad_group_class(const ctl_conv* d, int32_t has_dy2);
int ctl_wgrad_group_plan(const ctl_conv* descs, int32_t n, int32_t* splits);
int ctl_conv_wgrad_group(int32_t n, const ctl_conv* descs, const int32_t* splits, const float* const* x, const float* const* pro_scale,
                         const float* const* pro_shift, const float* const* dy, const float* const* dy2, const float* const* dy_coef,
                         float* const* w_partial, float* const* b_partial, ctl_stream stream);
int ctl_wgrad_reduce(const ctl_conv* d, const float* w_partial, const float* b_partial,
                     float* dw, int64_t s_co, int64_t s_ci, int64_t s_kh, int64_t s_kw,
                     float* dbias, int32_t accumulate, ctl_stream stream);

/* Batched forms (one launch for a whole network): `table` is a device array of int64 records.
 * pack record   (12 words): src_off, dst_off (floats into params / wpack), cout, cin, ks, flip, s_co, s_ci, s_kh, s_kw, total, 0
 * reduce record (16 words): w_off, b_off (floats into scratch; b_off < 0: none), dw_off, db_off (floats into grad; db_off < 0:
 *                           none), splits, taps|ks<<8, cin, cout, cin_p, cout_p, s_co, s_ci, s_kh, s_kw, accumulate, 0;
 *                           max_blocks = max over records of ceil((elements + cout) / (splits <= 64 ? 64 : 8)) */
int ctl_pack_weights_batched(const float* params, float* wpack, const int64_t* table, int32_t n_rec, int64_t max_total,
                             ctl_stream stream);
int ctl_wgrad_reduce_batched(const float* scratch, float* grad, const int64_t* table, int32_t n_rec, int64_t max_blocks,
                             ctl_stream stream);
/* bf16 MFMA fragments for the CTL_DT_BF16 kernels from the same pack records (modes 0-3): per 16-channel chunk a fragment carries a PAIR
 * of taps (v_mfma_f32_16x16x32_bf16 has 32 k-slots), values rounded to bf16 (RNE) from the fp32 master weights; written at the same
 * float offsets as the fp32 layout (it is smaller: 5 of 9 fragments for a 3x3 kernel).  max_total as for ctl_pack_weights_batched. */
int ctl_pack_weights_bf16_batched(const float* params, float* wpack, const int64_t* table, int32_t n_rec, int64_t max_total,
                                  ctl_stream stream);
/* Fragments for the CTL_DT_X3 launches from the pack records whose mode word carries CTL_PACK_X3 (the other records are skipped; the
 * fp32 / bf16 pack entries skip these): [cout tile][tap pair][chunk][split hi | mid | lo][64 lanes][8 bf16], the three bf16 numbers of
 * a split summing EXACTLY to the fp32 weight.  ctl_conv_wpack_floats_x3 = the float count of one sub-problem's buffer (15 KB per
 * (cout tile, chunk) of a 3x3 kernel against 9 KB of the fp32 layout); max_total = the largest such count among the records. */
size_t ctl_conv_wpack_floats_x3(int32_t cin, int32_t cout, int32_t ks);
int ctl_pack_weights_x3_batched(const float* params, float* wpack, const int64_t* table, int32_t n_rec, int64_t max_total,
                                ctl_stream stream);

/* ------------------------------------------------------------------------------------------------ BatchNorm2d
 * encdec.py: every `norm(out_ch)`; three modes of SURVEY 8a row 4 (util.py:414-451).
 * `groups` (>= 1): independent passes of one network batched along n (ctl_conv.groups).  Statistics partials are
 * [groups][blocks][2][c], every coefficient vector is [groups][c] (coef: [groups][3][c]), `count` is the pixel count of
 * ONE group, `pixels` the total; running statistics and dgamma/dbeta see the groups in order, as consecutive calls would.
 * finalize: partial [blocks][2][c] (sum, sum of squares over `count` pixels) -> scale=gamma*invstd,
 * shift=beta-mean*scale, save_mean, save_invstd; if update_running: running stats (momentum, unbiased var) and
 * num_batches_tracked (int64) are updated in place. */
int ctl_bn_finalize(const float* partial, int32_t blocks, int32_t c, int64_t count, const float* gamma,
                    const float* beta, float eps, float momentum, int32_t update_running, float* running_mean,
                    float* running_var, int64_t* num_batches_tracked, float* scale, float* shift, float* save_mean,
                    float* save_invstd, int32_t groups, ctl_stream stream);
/* ... additionally saving the unbiased batch variance as the float the running update used (save_uvar, [groups][c], may be NULL) */
int ctl_bn_finalize_ex(const float* partial, int32_t blocks, int32_t c, int64_t count, const float* gamma,
                       const float* beta, float eps, float momentum, int32_t update_running, float* running_mean,
                       float* running_var, int64_t* num_batches_tracked, float* scale, float* shift, float* save_mean,
                       float* save_invstd, float* save_uvar, int32_t groups, ctl_stream stream);
/* The running-statistics update of n_rec BatchNorm layers replayed from saved batch statistics (save_mean / save_uvar of a forward pass
 * whose activations are re-used instead of recomputed: the saliency pass of the targeted latent masks decodes, in training mode, the very
 * code the standard pass has just decoded -- util.py:214 after model.py:444 / 436-440).  table: n_rec x 6 int64 {save_mean byte offset in
 * `act`, save_uvar byte offset in `act`, running_mean / running_var float offsets in `buffers`, num_batches_tracked index, c}.
 * Bit-identical to running the pass again: the same two floats enter the same momentum update. */
int ctl_bn_replay_running(const void* act, float* buffers, int64_t* num_batches_tracked, const int64_t* table, int32_t n_rec,
                          float momentum, ctl_stream stream);
/* eval mode: scale = gamma/sqrt(running_var+eps), shift = beta - running_mean*scale */
int ctl_bn_eval_coeffs(int32_t c, const float* gamma, const float* beta, const float* running_mean,
                       const float* running_var, float eps, float* scale, float* shift, int32_t groups, ctl_stream stream);
/* y = leaky(x*scale[c]+shift[c], slope)  (slope 0 = ReLU, slope 1 = identity) */
int ctl_bn_act(const float* x, const float* scale, const float* shift, float slope, float* y, int64_t pixels,
               int32_t c, int32_t groups, ctl_stream stream);

/* backward helpers; `partial` buffers are [groups][rows][2][c] floats, rows = ctl_bwd_reduce_rows() <= CTL_RED_BLOCKS */
#ifndef CTL_RED_BLOCKS
#define CTL_RED_BLOCKS 512
#endif
/* mode 0 (residual tail, encdec.py:64,344): g = dout * leaky'(out);        sums: sum g, sum g*v
 * mode 1 (BN->act tail):                    g = da * leaky'(u*scale+shift); sums: sum g, sum g*u
 * mode 2 (plain):                           g = da;                         sums: sum g, (unused)          */
int ctl_bwd_reduce(int32_t mode, const float* dy, const float* act_src, const float* bn_src, const float* scale,
                   const float* shift, float slope, int64_t pixels, int32_t c, float* partial, int32_t groups,
                   ctl_stream stream);
/* rows per group that ctl_bwd_reduce* writes for this problem: min(CTL_RED_BLOCKS, max(16, ceil(quads per group / 2048))) for modes
 * 0 / 1, CTL_RED_BLOCKS for mode 2; ctl_red_blocks() returns the compiled CTL_RED_BLOCKS (bindings size their scratch with it) */
int ctl_bwd_reduce_rows(int32_t mode, int64_t pixels_per_group, int32_t c);
int ctl_red_blocks(void);
/* partial -> coefficients A,B,C with dx = A*g + B*bn_src + C (training-mode BN backward), and, if dgamma/dbeta
 * are non-NULL, dgamma += sum g*xhat, dbeta += sum g (accumulate ? += : =). */
/* `blocks` = rows per group of `partial` (0 = CTL_RED_BLOCKS, i.e. written by ctl_bwd_reduce; a conv with CTL_EPI_BNBWD
 * writes ctl_conv_stats_blocks rows) */
int ctl_bn_bwd_finalize(const float* partial, int32_t c, int64_t count, const float* gamma, const float* save_mean,
                        const float* save_invstd, float* coef, float* dgamma, float* dbeta, int32_t accumulate,
                        int32_t groups, int32_t blocks, ctl_stream stream);
/* the same with a per-group switch for the gamma / beta gradients: bit g of `affine_groups` set = group g adds its sums (0 = every group).
 * A launch that stacks passes of different BatchNorm modes along n (round 6: the standard pass, mode A, and the hard-example pass, mode B =
 * gamma / beta frozen for that pass, model_util.py:414-451) clears the bits of the frozen passes; every group still gets its coefficients. */
int ctl_bn_bwd_finalize_ex(const float* partial, int32_t c, int64_t count, const float* gamma, const float* save_mean,
                           const float* save_invstd, float* coef, float* dgamma, float* dbeta, int32_t accumulate,
                           int32_t groups, int32_t blocks, uint32_t affine_groups, ctl_stream stream);
/* mode 0: ds = dout*leaky'(out) (written if ds != NULL), dv = A*ds + B*v + C;  mode 1: du = A*g + B*u + C with g = dy*leaky'(..);
 * mode 2: dy is already g (CTL_EPI_BNBWD): du = A*dy + B*u + C */
int ctl_bwd_apply(int32_t mode, const float* dy, const float* act_src, const float* bn_src, const float* scale,
                  const float* shift, float slope, const float* coef, int64_t pixels, int32_t c, float* ds,
                  float* dx, int32_t groups, ctl_stream stream);
/* partial[blocks][2][c] -> out[c] (+)= sum over blocks of row 0 (bias gradient of ConvTranspose2d) */
int ctl_chan_sum_finalize(const float* partial, int32_t c, float* out, int32_t accumulate, ctl_stream stream);
/* nearest-upsample backward: dx[n,h,w,c] = sum of the 2x2 block of dup[n,2h,2w,c]; accumulate ? += : = */
int ctl_sumpool2(const float* dup, float* dx, int32_t n, int32_t h, int32_t w, int32_t c, int32_t accumulate,
                 ctl_stream stream);
/* dlogit = dy * y * (1-y)  (nn.Sigmoid of image_decoder, model.py:100) */
int ctl_sigmoid_bwd(const float* dy, const float* y, float* dx, int64_t count, ctl_stream stream);

/* ------------------------------------------------------------------------------------------------ STN input, losses
 * construct_input (medseg/common_utils/basic_operations.py:110-158): softmax(x/T, dim=C) or one-hot(label) */
int ctl_softmax_t_fwd(const float* x, float inv_t, float* p, int64_t pixels, int32_t c, ctl_stream stream);
int ctl_softmax_t_bwd(const float* p, const float* dp, float inv_t, float* dx, int64_t pixels, int32_t c,
                      ctl_stream stream);
int ctl_onehot(const int64_t* label, float* y, int64_t pixels, int32_t c, ctl_stream stream);
/* cross_entropy_2D (medseg/models/custom_loss.py:706-740, util.py:104-115): loss = mean_pixels -log_softmax[label].
 * fwd writes loss[0]; partial is [CTL_RED_BLOCKS] doubles.  bwd: dlogit = gout[0] * (softmax - onehot) / pixels */
int ctl_ce2d_fwd(const float* logit, const int64_t* label, int64_t pixels, int32_t c, double* partial, float* loss,
                 ctl_stream stream);
int ctl_ce2d_bwd(const float* logit, const int64_t* label, const float* gout, int64_t pixels, int32_t c,
                 float* dlogit, ctl_stream stream);
/* loss = scale * mean((a-b)^2) (model.py:445-447: scale 0.5; util.py:216: scale 1); bwd: da = gout*2*scale*(a-b)/count */
int ctl_mse_fwd(const float* a, const float* b, int64_t count, float scale, double* partial, float* loss,
                ctl_stream stream);
int ctl_mse_bwd(const float* a, const float* b, const float* gout, int64_t count, float scale, float* da,
                ctl_stream stream);
/* pred.max(1)[1] (model.py:657): first maximal channel, uint8 out */
int ctl_argmax_c(const float* logit, uint8_t* out, int64_t pixels, int32_t c, ctl_stream stream);

/* Storage-type aware forms of the element-wise kernels that touch network-internal tensors (BASELINE config 3 stores those as bf16):
 * `bf16_mask` bit k = tensor argument k (in the order x,y | dy,act_src,bn_src | dy,act_src,bn_src,ds,dx | dup,dx) is bf16; the
 * arithmetic is fp32, a store rounds once (RNE).  mask 0 == the plain entry points above. */
int ctl_bn_act_dt(const float* x, const float* scale, const float* shift, float slope, float* y, int64_t pixels, int32_t c,
                  int32_t groups, uint32_t bf16_mask, ctl_stream stream);
/* ds (mode 0 only, may be NULL): also writes ds = dy * leaky'(act_src) (bf16_mask bit 3 = its storage), so that the apply pass can run
 * in mode 2 on ds instead of recomputing it from dy and act_src */
int ctl_bwd_reduce_dt(int32_t mode, const float* dy, const float* act_src, const float* bn_src, const float* scale,
                      const float* shift, float slope, int64_t pixels, int32_t c, float* partial, int32_t groups,
                      uint32_t bf16_mask, float* ds, ctl_stream stream);
int ctl_bwd_apply_dt(int32_t mode, const float* dy, const float* act_src, const float* bn_src, const float* scale,
                     const float* shift, float slope, const float* coef, int64_t pixels, int32_t c, float* ds, float* dx,
                     int32_t groups, uint32_t bf16_mask, ctl_stream stream);
int ctl_sumpool2_dt(const float* dup, float* dx, int32_t n, int32_t h, int32_t w, int32_t c, int32_t accumulate,
                    uint32_t bf16_mask, ctl_stream stream);

/* ------------------------------------------------------------------------------------------------ latent masking
 * util.py:224-249 (channel) / 285-312 (spatial).  mode 0: score[n,c] = mean_hw grad; mode 1: score[n,hw] = mean_c grad.
 * `scratch` holds ctl_latent_score_ws_floats() floats (deterministic two-stage sum, no float atomics).
 * ctl_latent_mask_apply: entry i of row n is masked iff
 * #{j : score[n,j] >= score[n,i]} <= k  (== "score > sort(desc)[k]", strict, util.py:231-244);
 * mask value = soft_noise ? 0.5*soft_noise[n,i] : 0; kept = 1.  k is read from k_dev[0] if k_dev != NULL (graph replay)
 * else from k_host.  masked = code * mask (broadcast), mask_out [n,L].  Rows up to 1024 entries are ranked inside the apply
 * kernel; longer rows (spatial mode on large latents) take the threshold from a per-image bitonic sort (scratch). */
size_t ctl_latent_score_ws_floats(int32_t mode, int32_t n, int32_t hw, int32_t c);
int ctl_latent_score(int32_t mode, const float* grad, float* score, float* scratch, int32_t n, int32_t hw, int32_t c,
                     ctl_stream stream);
size_t ctl_latent_mask_apply_ws_floats(int32_t mode, int32_t n, int32_t hw, int32_t c);   /* 0 unless the row is > 1024 long */
int ctl_latent_mask_apply(int32_t mode, const float* code, const float* score, const float* soft_noise,
                          int32_t k_host, const int32_t* k_dev, float* masked, float* mask_out, float* scratch, int32_t n,
                          int32_t hw, int32_t c, ctl_stream stream);
/* The whole generator tail behind one call (reference: util.py:224-249 / 285-312).  Latent codes up to 64 Ki elements per image with
 * rows <= 1024 (the configured 128 x 16 x 16 included) run as ONE launch: one 1024-thread block per image holds the image's grad and
 * code in registers, builds the score row in LDS, ranks, and stores code * mask -- no workspace (ctl_latent_mask_fused_ws_floats == 0).
 * Larger problems are HBM streams and run as the score + apply launches above on `workspace`.  Scores (and therefore masks) are
 * bit-identical between the two forms.  score_out (nullable) receives the [n,L] scores.  Rows up to 8192 entries. */
size_t ctl_latent_mask_fused_ws_floats(int32_t mode, int32_t n, int32_t hw, int32_t c);
int ctl_latent_mask_fused(int32_t mode, const float* grad, const float* code, const float* soft_noise, int32_t k_host,
                          const int32_t* k_dev, float* masked, float* mask_out, float* score_out, float* workspace, int32_t n,
                          int32_t hw, int32_t c, ctl_stream stream);
/* F.dropout2d(z,p) (model.py:333): out = z * keep[n,c] / (1-p).  keep != NULL: injected {0,1} floats; else drawn on
 * device from a counter hash of (seed, n*c index) and written to keep_out. */
int ctl_dropout2d(const float* z, const float* keep, uint64_t seed, float p, float* out, float* keep_out, int32_t n,
                  int32_t hw, int32_t c, ctl_stream stream);
/* 0.5*U[0,1) style uniform fill from the same counter hash (soft-mask noise, util.py:239) */
int ctl_uniform(float* out, int64_t count, uint64_t seed, ctl_stream stream);
/* HIP-graph-safe forms: nothing that changes from step to step is a launch ARGUMENT.  `state` is a device int64[3]:
 * [0] RNG seed, [1] RNG step counter, [2] Adam step count; ctl_step_tick (one launch at the head of a training step, replaces the
 * host-side `step += 1` of torch.optim.Adam and the per-call host seed draw) advances [1] and [2].  With state != NULL the first
 * integer argument is a per-call-site salt.  ctl_dropout2d_ex additionally writes (mask_full != NULL) upstream's dropout `mask`
 * (model.py:334-336: 1 where the dropped-out tensor equals the input element, else 0; [n,hw,c] like out). */
int ctl_step_tick(int64_t* state, ctl_stream stream);
/* One idle wave for `microseconds` on `stream`: the host-side probe for "do these two streams overlap?" (streams share a few hardware
 * queues; the two launch chains of a training step must not sit on the same one, see solver.py) */
int ctl_spin(int32_t microseconds, ctl_stream stream);
int ctl_dropout2d_ex(const float* z, const float* keep, uint64_t seed_or_salt, const int64_t* state, float p, float* out,
                     float* keep_out, float* mask_full, int32_t n, int32_t hw, int32_t c, ctl_stream stream);
int ctl_uniform_dev(float* out, int64_t count, uint64_t salt, const int64_t* state, ctl_stream stream);
/* nn.Dropout2d behind every residual block (encoder_dropout / decoder_dropout, model.py:27-28, 92-106; encoder_decoder.py:58-66, 338-347):
 * ctl_dropout2d_ex on network-internal tensors, which BASELINE config 3 stores as bf16.  bf16_mask: bit 0 = z, bit 1 = out are bf16
 * ([n,hw,c] either way); the product z * keep / (1-p) is formed in fp32 and rounded once by the store. */
int ctl_dropout2d_dt(const void* z, const float* keep, uint64_t seed_or_salt, const int64_t* state, float p, void* out,
                     float* keep_out, int32_t n, int32_t hw, int32_t c, uint32_t bf16_mask, ctl_stream stream);

/* ------------------------------------------------------------------------------------------------ SURVEY 8(f) rows 1, 3
 * Validation metrics and the input pipeline on device (no host round trip per batch).
 * ctl_confusion_hist: `runningScore._fast_hist` (medseg/common_utils/metrics.py:18-23): hist[n_class*t + p] += 1 for every
 *   element with 0 <= t < n_class (p = predicted label, uint8 as written by ctl_argmax_c); hist is int64 [n_class*n_class] and
 *   ACCUMULATES (zero it to start a new evaluation); n_class <= 16.  Integer atomics only: order-independent, bit-exact.
 * ctl_rescale_intensity: per plane (n*c planes of plane_elems floats) (x-min)/(max-min+eps)*(new_max-new_min)+new_min
 *   (medseg/common_utils/basic_operations.py:232-245), torch's operation order, one rounding per operation.
 * ctl_noise_clamp: out = clamp(x + noise, lo, hi) (train_adv_supervised_segmentation_triplet.py:185-187); noise == NULL:
 *   sigma * N(0,1) drawn on device from a counter hash of (seed, index) (Box-Muller).
 * ctl_crop_or_pad: centre crop / zero pad of [n,h,w] arrays to [n,new_h,new_w] (medseg/common_utils/basic_operations.py:
 *   173-220): dst[y][x] = src[y + floor((h-new_h)/2)][x + floor((w-new_w)/2)] or 0 outside; elem_bytes 1, 4 or 8. */
int ctl_confusion_hist(const int64_t* label_true, const uint8_t* label_pred, int64_t count, int32_t n_class, int64_t* hist,
                       ctl_stream stream);
size_t ctl_rescale_intensity_ws_floats(int32_t planes);
int ctl_rescale_intensity(const float* x, float* out, float* workspace, int32_t planes, int64_t plane_elems, float new_min,
                          float new_max, float eps, ctl_stream stream);
int ctl_noise_clamp(const float* x, const float* noise, uint64_t seed, float sigma, float lo, float hi, float* out,
                    int64_t count, ctl_stream stream);
int ctl_crop_or_pad(const void* src, void* dst, int32_t elem_bytes, int32_t n, int32_t h, int32_t w, int32_t new_h,
                    int32_t new_w, ctl_stream stream);

/* ------------------------------------------------------------------------------------------------ optimizer
 * torch.optim.Adam defaults (model.py:774-785), one flat buffer: p,g,m,v [count].  step = 1-based step index.
 * grad_scale folds the 1/world_size of the data-parallel all-reduce. */
int ctl_adam(float* p, const float* g, float* m, float* v, int64_t count, float lr, float beta1, float beta2,
             float eps, int32_t step, float grad_scale, ctl_stream stream);
/* dst[i] += srcs[0][i] + ... + srcs[k-1][i] in that order (1 <= k <= 8; `srcs` is a HOST array of device pointers): the flat parameter
 * gradients that the passes of one network produced in a step, added into the network's gradient buffer by ONE launch after
 * loss.backward() instead of one autograd accumulation per pass (which, with two launch chains, is a cross-stream dependency each). */
int ctl_accumulate(float* dst, const float* const* srcs, int32_t k, int64_t count, ctl_stream stream);
/* same update with the step count read from state[2] on the device (bias corrections computed in double there) */
int ctl_adam_dev(float* p, const float* g, float* m, float* v, int64_t count, float lr, float beta1, float beta2,
                 float eps, const int64_t* state, float grad_scale, ctl_stream stream);

/* ------------------------------------------------------------------------------------------------ plans
 * A plan is an array of ctl_op executed in order on one stream: one C call per network pass (the Python host builds
 * it once per (network, shape, mode)).  Tensor arguments are (slot, byte offset) pairs resolved against `bases`. */
enum ctl_op_kind {
    CTL_OP_CONV = 1, CTL_OP_WGRAD = 2, CTL_OP_WGRAD_REDUCE = 3, CTL_OP_PACK = 4, CTL_OP_BN_FINALIZE = 5,
    CTL_OP_BN_EVAL = 6, CTL_OP_BN_ACT = 7, CTL_OP_BWD_REDUCE = 8, CTL_OP_BN_BWD_FINALIZE = 9, CTL_OP_BWD_APPLY = 10,
    CTL_OP_CHAN_SUM_FINALIZE = 11, CTL_OP_SUMPOOL2 = 12, CTL_OP_SIGMOID_BWD = 13, CTL_OP_ZERO = 14, CTL_OP_COPY = 15, CTL_OP_PACK_BATCH = 16,
    CTL_OP_WGRAD_REDUCE_BATCH = 17, CTL_OP_DROPOUT2D = 18, CTL_OP_BN_REPLAY = 19,
    CTL_OP_WGRAD_GROUP = 20           /* i[0] = n members (<= 8): the next n records are WGRAD records served by ONE launch (ctl_conv_wgrad_group), i[24] of each
                                         member = its pixel splits (ctl_wgrad_group_plan).  The members' partial buffers and reduction records are sized for
                                         THOSE split counts: a member record (i[24] != 0) must never be launched on its own -- ctl_plan_run refuses one that is
                                         not preceded by its GROUP record */
};
#define CTL_OP_MAX_T 14
typedef struct ctl_op {
    int32_t kind;
    int32_t i[27];                    /* CONV/WGRAD/WGRAD_REDUCE: i[0..23] = ctl_conv as int32 words, i[24] = accumulate; i[25] = bf16
                                         storage mask of the element-wise ops; others: see ctl_plan.cpp */
    float   f[4];
    int32_t slot[CTL_OP_MAX_T];       /* -1 = NULL */
    int64_t off[CTL_OP_MAX_T];
    int64_t l[4];                     /* 64-bit scalars (strides, counts) */
} ctl_op;
int ctl_plan_run(const ctl_op* ops, int32_t n_ops, void* const* bases, int32_t n_bases, ctl_stream stream);

/* ------------------------------------------------------------------------------------------------ in-process profiling
 * Brackets every conv-family launch whose kernel id contains `filter` ("" = all) with hipEvents recorded on the launch
 * stream, and sums the ALGORITHMIC work of those launches: flops = 2*pixels*cout*cin*ks*ks (real channels), bytes =
 * input + output (+ residual / accumulate reads) tensors once each.  Kernel ids look like
 * "conv_igemm<ks3,s1,in0,mt4,tw32,nt1>" / "conv_wgrad<...>" (the template instantiation rocprofv3 reports).
 * ctl_prof_stop synchronises the recorded events and writes one text line per kernel id:
 *   "<id> launches=<n> ms=<total> flops=<sum> bytes=<sum>\n".  Not for use under graph capture. */
int ctl_prof_start(const char* filter);
/* the same, bracketing only every `every`-th matching launch (bench.py samples inside its timed region: two event records per launch
 * are not free on a launch-bound step) */
int ctl_prof_start_sampled(const char* filter, int32_t every);
int ctl_prof_stop(char* out, size_t cap);
/* launch census: kernels / stream memsets / copies enqueued by this library since it was loaded (bench.py reports launches per step) */
unsigned long long ctl_launch_count(void);
size_t ctl_sizeof_op(void);
size_t ctl_sizeof_conv(void);

#ifdef __cplusplus
}
#endif
#endif
