// Plan executor: runs an array of ctl_op on one HIP stream (one C call per network pass), plus the library's
// version / error plumbing.  No allocation, no synchronisation: safe under hipStreamBeginCapture.
//
// Tensor slots per op kind (index into op.slot/op.off):
//   CONV              0 x  1 wpack  2 bias  3 pro_scale  4 pro_shift  5 res  6 res_scale  7 res_shift  8 y  9 stats_partial  10 res2 (CTL_EPI_TAILBWD)
//   WGRAD             0 x  1 pro_scale  2 pro_shift  3 dy  4 w_partial  5 b_partial
//   WGRAD_REDUCE      0 w_partial  1 b_partial  2 dw  3 dbias            i[24]=accumulate  l[0..3]=s_co,s_ci,s_kh,s_kw
//   PACK              0 src  1 dst                                       i[0..3]=cout,cin,ks,flip  l[0..3]=strides
//   BN_FINALIZE       0 partial 1 gamma 2 beta 3 running_mean 4 running_var 5 nbt 6 scale 7 shift 8 save_mean 9 save_invstd 10 save_uvar
//   BN_REPLAY         0 act arena 1 buffers 2 nbt 3 table                i[0]=n_rec f[0]=momentum
//                                                                        i[0..2]=blocks,c,update_running l[0]=count f[0]=eps f[1]=momentum
//   BN_EVAL           0 gamma 1 beta 2 running_mean 3 running_var 4 scale 5 shift      i[0]=c f[0]=eps
//   BN_ACT            0 x 1 scale 2 shift 3 y                            i[0]=c l[0]=pixels f[0]=slope
//   BWD_REDUCE        0 dy 1 act_src 2 bn_src 3 scale 4 shift 5 partial 6 ds (mode 0, optional)  i[0]=mode i[1]=c l[0]=pixels f[0]=slope
//   BN_BWD_FINALIZE   0 partial 1 gamma 2 save_mean 3 save_invstd 4 coef 5 dgamma 6 dbeta   i[0]=c i[1]=accumulate i[2]=groups i[3]=rows i[4]=affine group mask (0 = all) l[0]=count
//   BWD_APPLY         0 dy 1 act_src 2 bn_src 3 scale 4 shift 5 coef 6 ds 7 dx   i[0]=mode i[1]=c l[0]=pixels f[0]=slope
//   CHAN_SUM_FINALIZE 0 partial 1 out                                    i[0]=c i[1]=accumulate
//   SUMPOOL2          0 dup 1 dx                                         i[0..4]=n,h,w,c,accumulate
//   SIGMOID_BWD       0 dy 1 y 2 dx                                      l[0]=count
//   ZERO              0 ptr                                              l[0]=bytes
//   COPY              0 src 1 dst                                        l[0]=bytes
//   PACK_BATCH        0 params 1 wpack 2 table                           i[0]=n_rec i[1]=bit 0 bf16 fragments, bit 1 CTL_PACK_X3 records l[0]=max_total
//   WGRAD_REDUCE_BATCH 0 scratch 1 grad 2 table                          i[0]=n_rec l[0]=max_elems
//   DROPOUT2D         0 z 1 keep (given pattern, optional) 2 state (device RNG state, optional) 3 out 4 keep_out (optional)
//                                                                        i[0..2]=n,hw,c f[0]=p l[0]=seed (no state) | call-site salt (state)
#include <stdarg.h>
#include <stdlib.h>
#include <string.h>

#include "ctl_common.h"

static thread_local char g_err[512] = "";

void ctl_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

// ------------------------------------------------------------------------------------------------ profiling
#include <atomic>
#include <map>
#include <string>
#include <vector>
namespace {
struct ProfRec { std::string id; hipEvent_t a, b; double flops, bytes; void* stream; };
bool g_prof_on = false;
std::string g_prof_filter;
std::vector<ProfRec> g_prof;
std::vector<hipEvent_t> g_prof_pool;      // events are recycled: two hipEventCreate per bracketed launch were most of the profiler's cost
int g_prof_every = 1;                     // bracket every n-th matching launch (sampling: the timed region of bench.py)
long g_prof_seen = 0;
hipEvent_t prof_event() {
    if (!g_prof_pool.empty()) { hipEvent_t e = g_prof_pool.back(); g_prof_pool.pop_back(); return e; }
    hipEvent_t e = nullptr;
    return hipEventCreate(&e) == hipSuccess ? e : nullptr;
}
}  // namespace

int ctl_prof_begin(const char* kind, const ctl_conv* d, const ctl_conv_cfg* c, int nt, hipStream_t stream, bool dy2) {
    if (!g_prof_on) return -1;
    char id[160], sfx[40];
    // one id per kernel instantiation family, as rocprofv3 lists them: the epilogue class (fp32: the EPI template value, 1 = residual /
    // accumulate / BatchNorm-backward operand, 2 = tail reduction, 3 = BatchNorm-backward reduction alone; bf16: the FAST value) and the two-tensor prologue are part of it
    const int f = d->epi_flags;
    const bool bnb_only = (f & (CTL_EPI_RES | CTL_EPI_ACCUM | CTL_EPI_BNBWD)) == CTL_EPI_BNBWD && d->ks == 3 && d->stride == 1 && d->in_mode == CTL_IN_PLAIN && d->cout % 16 == 0;
    const int e32 = (f & CTL_EPI_TAILBWD) ? 2 : (bnb_only ? 3 : ((f & (CTL_EPI_RES | CTL_EPI_ACCUM | CTL_EPI_BNBWD)) ? 1 : 0));
    const int e16 = (f & CTL_EPI_TAILBWD) ? 5 : ((f & CTL_EPI_BNBWD) ? 4 : ((f & CTL_EPI_RES) ? 2 : ((f & CTL_EPI_ACCUM) ? 3 : 0)));
    const int e = (d->dt & CTL_DT_BF16) ? e16 : e32;
    const bool two = d->pro_affine == 2 || dy2;
    char nar[12] = "";      // fewer than 16 output channels: most of the MFMA's N columns are padding -- an HBM-side launch of a matrix-side template
    if (d->cout < 16) snprintf(nar, sizeof(nar), ",co%d", d->cout);
    if (e) snprintf(sfx, sizeof(sfx), ",e%d%s%s", e, two ? ",x2" : "", nar);
    else snprintf(sfx, sizeof(sfx), "%s%s", two ? ",x2" : "", nar);
    static const bool shapes = ctl_tune_str("CTL_PROF_SHAPES") != nullptr;      // (-DCTL_TUNING builds: one id per layer shape)
    if (shapes)
        snprintf(id, sizeof(id), "%s<ks%d,s%d,in%d,mt%d,tw%d,nt%d%s>[n%d,h%d,ci%d,co%d,e%d]", kind, d->ks, d->stride, d->in_mode, c->mt, c->tw, nt, sfx, d->n,
                 d->hout, d->cin, d->cout, d->epi_flags);
    else
        snprintf(id, sizeof(id), "%s<ks%d,s%d,in%d,mt%d,tw%d,nt%d%s>", kind, d->ks, d->stride, d->in_mode, c->mt, c->tw, nt, sfx);
    if (!g_prof_filter.empty() && std::string(id).find(g_prof_filter) == std::string::npos) return -1;
    if (g_prof_every > 1 && (g_prof_seen++ % g_prof_every) != 0) return -1;
    ProfRec r;
    r.id = id;
    r.stream = (void*)stream;
    r.a = prof_event();
    r.b = prof_event();
    if (!r.a || !r.b) return -1;
    const double pix = (double)d->n * d->hout * d->wout * d->nsub;
    r.flops = 2.0 * pix * d->cout * d->cin * d->ks * d->ks;
    const double in_b = ((d->dt & CTL_DT_X16) ? 2.0 : 4.0) * d->n * d->hin * d->win * d->cin, out_b = ((d->dt & CTL_DT_Y16) ? 2.0 : 4.0) * pix * d->cout;
    r.bytes = in_b + out_b;
    if (d->pro_affine == 2) r.bytes += in_b;             // the second tensor of the BatchNorm-backward prologue
    if (dy2) r.bytes += out_b;                           // ... of the weight gradient's virtual output gradient
    if (d->epi_flags & (CTL_EPI_RES | CTL_EPI_BNBWD)) r.bytes += ((d->dt & CTL_DT_RES16) ? 2.0 : 4.0) * pix * d->cout;
    if (d->epi_flags & CTL_EPI_TAILBWD) r.bytes += 2.0 * 4.0 * pix * d->cout;            // the block output and the BatchNorm input of the tail
    if (d->epi_flags & CTL_EPI_ACCUM) r.bytes += out_b;
    (void)hipEventRecord(r.a, stream);
    g_prof.push_back(r);
    return (int)g_prof.size() - 1;
}
// a launch that serves several problems (grouped weight gradients): the caller sums the members' algorithmic work
int ctl_prof_begin_raw(const char* id, double flops, double bytes, hipStream_t stream) {
    if (!g_prof_on) return -1;
    if (!g_prof_filter.empty() && std::string(id).find(g_prof_filter) == std::string::npos) return -1;
    if (g_prof_every > 1 && (g_prof_seen++ % g_prof_every) != 0) return -1;
    ProfRec r;
    r.id = id;
    r.stream = (void*)stream;
    r.a = prof_event();
    r.b = prof_event();
    if (!r.a || !r.b) return -1;
    r.flops = flops; r.bytes = bytes;
    (void)hipEventRecord(r.a, stream);
    g_prof.push_back(r);
    return (int)g_prof.size() - 1;
}
// HBM-bound plan ops (BatchNorm backward passes, bn_act, sum-pool): bracketed with their ALGORITHMIC bytes (every tensor argument once,
// at its storage width); the other non-conv ops (finalizes, copies) only for the timeline dump of a -DCTL_TUNING build
static int prof_begin_op(const ctl_op& op, hipStream_t stream) {
    if (!g_prof_on) return -1;
    static const bool timeline = ctl_tune_str("CTL_PROF_TIMELINE") != nullptr;
    char id[96];
    double bytes = 0.0;
    const unsigned m = (unsigned)op.i[25];
    auto w = [&](int bit) { return (m >> bit) & 1u ? 2.0 : 4.0; };
    switch (op.kind) {
        case CTL_OP_BWD_REDUCE: {
            const double e = (double)op.l[0] * op.i[1];
            snprintf(id, sizeof(id), "bwd_reduce<%d>", op.i[0]);
            bytes = e * w(0) + (op.i[0] == 0 ? e * w(1) : 0.0) + (op.i[0] != 2 ? e * w(2) : 0.0) + (op.slot[6] >= 0 ? e * w(3) : 0.0);
            break;
        }
        case CTL_OP_BWD_APPLY: {
            const double e = (double)op.l[0] * op.i[1];
            snprintf(id, sizeof(id), "bwd_apply<%d>", op.i[0]);
            bytes = e * w(0) + (op.i[0] == 0 ? e * w(1) : 0.0) + e * w(2) + (op.slot[6] >= 0 ? e * w(3) : 0.0) + e * w(4);
            break;
        }
        case CTL_OP_BN_ACT: {
            const double e = (double)op.l[0] * op.i[0];
            snprintf(id, sizeof(id), "bn_act");
            bytes = e * w(0) + e * w(1);
            break;
        }
        case CTL_OP_SUMPOOL2: {
            const double e = (double)op.i[0] * op.i[1] * op.i[2] * op.i[3];
            snprintf(id, sizeof(id), "sumpool2");
            bytes = 4.0 * e * w(0) + e * w(1) * (op.i[4] ? 2.0 : 1.0);
            break;
        }
        default:
            if (!timeline) return -1;
            snprintf(id, sizeof(id), "op%d", op.kind);
    }
    if (!g_prof_filter.empty() && std::string(id).find(g_prof_filter) == std::string::npos) return -1;
    if (g_prof_every > 1 && (g_prof_seen++ % g_prof_every) != 0) return -1;
    ProfRec r;
    r.id = id;
    r.stream = (void*)stream;
    r.flops = 0.0;
    r.bytes = bytes;
    r.a = prof_event();
    r.b = prof_event();
    if (!r.a || !r.b) return -1;
    (void)hipEventRecord(r.a, stream);
    g_prof.push_back(r);
    return (int)g_prof.size() - 1;
}
void ctl_prof_end(int token, hipStream_t stream) {
    if (token >= 0 && token < (int)g_prof.size()) (void)hipEventRecord(g_prof[token].b, stream);
}
extern "C" int ctl_prof_start_sampled(const char* filter, int32_t every) {
    for (auto& r : g_prof) { g_prof_pool.push_back(r.a); g_prof_pool.push_back(r.b); }
    g_prof.clear();
    g_prof_filter = filter ? filter : "";
    g_prof_every = every > 1 ? every : 1;
    g_prof_seen = 0;
    g_prof_on = true;
    return CTL_OK;
}
extern "C" int ctl_prof_start(const char* filter) { return ctl_prof_start_sampled(filter, 1); }
extern "C" int ctl_prof_stop(char* out, size_t cap) {
    g_prof_on = false;
    struct Agg { long n = 0; double ms = 0, flops = 0, bytes = 0; };
    std::map<std::string, Agg> agg;
    // CTL_PROF_TIMELINE=<file>: raw per-launch intervals (ms since the first bracketed launch) with their stream -- the real
    // overlap of concurrent launch chains, which rocprofv3's kernel trace hides by serialising the dispatches
    FILE* tl = nullptr;
    if (const char* path = ctl_tune_str("CTL_PROF_TIMELINE")) tl = fopen(path, "w");
    for (auto& r : g_prof) {
        if (tl && !g_prof.empty() && hipEventSynchronize(r.b) == hipSuccess) {
            float t0 = 0.f, t1 = 0.f;
            if (hipEventElapsedTime(&t0, g_prof[0].a, r.a) == hipSuccess && hipEventElapsedTime(&t1, g_prof[0].a, r.b) == hipSuccess)
                fprintf(tl, "%s %p %.6f %.6f\n", r.id.c_str(), r.stream, t0, t1);
        }
        float ms = 0.f;
        if (hipEventSynchronize(r.b) == hipSuccess && hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) {
            Agg& a = agg[r.id];
            a.n++; a.ms += ms; a.flops += r.flops; a.bytes += r.bytes;
        }
    }
    for (auto& r : g_prof) { g_prof_pool.push_back(r.a); g_prof_pool.push_back(r.b); }
    g_prof.clear();
    if (tl) fclose(tl);
    std::string text;
    for (auto& kv : agg) {
        char line[320];
        snprintf(line, sizeof(line), "%s launches=%ld ms=%.6f flops=%.6e bytes=%.6e\n", kv.first.c_str(), kv.second.n,
                 kv.second.ms, kv.second.flops, kv.second.bytes);
        text += line;
    }
    if (out && cap) {
        CTL_REQUIRE(text.size() + 1 <= cap, "prof_stop: output buffer too small (%zu needed)", text.size() + 1);
        memcpy(out, text.c_str(), text.size() + 1);
    }
    return CTL_OK;
}

static std::atomic<unsigned long long> g_launches{0};
void ctl_count_launches(int n) { g_launches.fetch_add((unsigned long long)n, std::memory_order_relaxed); }
extern "C" unsigned long long ctl_launch_count(void) { return g_launches.load(std::memory_order_relaxed); }
extern "C" int ctl_version(void) { return CTL_ABI_VERSION; }
extern "C" const char* ctl_last_error(void) { return g_err; }
extern "C" size_t ctl_sizeof_op(void) { return sizeof(ctl_op); }
extern "C" size_t ctl_sizeof_conv(void) { return sizeof(ctl_conv); }

static_assert(sizeof(ctl_conv) == 24 * 4, "ctl_conv must be 24 32-bit words (it is embedded in ctl_op.i[0..23]; i[24] and i[25] are taken)");

extern "C" int ctl_plan_run(const ctl_op* ops, int32_t n_ops, void* const* bases, int32_t n_bases, ctl_stream stream_) {
    CTL_REQUIRE(ops && bases && n_ops >= 0, "plan_run: null arguments");
    const ctl_stream stream = stream_;
#ifdef CTL_TUNING
    // ceiling probes (wrong numbers, right launch structure): what would the step cost without these launches?
    static const int skip_mask = ctl_tune_int("CTL_SKIP_OPS", 0);      // bit 0: BN_FINALIZE, 1: BN_BWD_FINALIZE, 2: BWD_REDUCE mode 0, 3: BWD_REDUCE mode 1, 4: BWD_APPLY
#endif
    for (int32_t k = 0; k < n_ops; ++k) {
        const ctl_op& op = ops[k];
#ifdef CTL_TUNING
        if (((skip_mask & 1) && op.kind == CTL_OP_BN_FINALIZE) || ((skip_mask & 2) && op.kind == CTL_OP_BN_BWD_FINALIZE) ||
            ((skip_mask & 4) && op.kind == CTL_OP_BWD_REDUCE && op.i[0] == 0) || ((skip_mask & 8) && op.kind == CTL_OP_BWD_REDUCE && op.i[0] == 1) ||
            ((skip_mask & 16) && op.kind == CTL_OP_BWD_APPLY))
            continue;
#endif
        void* t[CTL_OP_MAX_T];
        for (int a = 0; a < CTL_OP_MAX_T; ++a) {
            const int s = op.slot[a];
            if (s < 0) { t[a] = nullptr; continue; }
            CTL_REQUIRE(s < n_bases && bases[s], "plan_run: op %d arg %d uses empty slot %d", k, a, s);
            t[a] = (char*)bases[s] + op.off[a];
        }
#define F(a) ((float*)t[a])
#define NG(v) ((v) > 0 ? (v) : 1)      /* BatchNorm groups: 0 in a record means one group */
#define CF(a) ((const float*)t[a])
        int rc = CTL_OK;
        ctl_conv d;
        const int ptok = (op.kind == CTL_OP_CONV || op.kind == CTL_OP_WGRAD) ? -1 : prof_begin_op(op, (hipStream_t)stream);
        switch (op.kind) {
            case CTL_OP_CONV:
                memcpy(&d, op.i, sizeof(d));
                rc = ctl_conv_forward_ex(&d, CF(0), CF(1), CF(2), CF(3), CF(4), CF(5), CF(6), CF(7), CF(10), CF(11), F(8), F(9), F(12), F(13), stream);
                break;
            case CTL_OP_WGRAD:
                // a member of a grouped launch carries the group's split count in i[24] (its partial buffers are sized for it): reached here, the
                // GROUP record in front of it is missing (a sliced or edited plan) and a launch of its own would overrun those buffers
                CTL_REQUIRE(op.i[24] == 0, "plan_run: op %d is a member record of a WGRAD_GROUP (i[24] = %d splits) without its GROUP record", k, op.i[24]);
                memcpy(&d, op.i, sizeof(d));
                rc = ctl_conv_wgrad_ex(&d, CF(0), CF(1), CF(2), CF(3), CF(6), CF(7), F(4), F(5), stream);
                break;
            case CTL_OP_WGRAD_GROUP: {      // i[0] = members: the next i[0] records are WGRAD records (not launched on their own), i[24] of each = its pixel splits
                const int nm = op.i[0];
                CTL_REQUIRE(nm >= 1 && nm <= 8 && k + nm < n_ops, "plan_run: op %d WGRAD_GROUP with %d members", k, nm);
                ctl_conv ds[8];
                int32_t sp[8];
                const float *mx[8], *mps[8], *mpb[8], *mdy[8], *mdy2[8], *mco[8];
                float *mw[8], *mb[8];
                for (int j = 0; j < nm; ++j) {
                    const ctl_op& mo = ops[k + 1 + j];
                    CTL_REQUIRE(mo.kind == CTL_OP_WGRAD, "plan_run: op %d: member %d of a WGRAD_GROUP is not a WGRAD record", k, j);
                    void* mt[8];
                    for (int a = 0; a < 8; ++a) {
                        const int s = mo.slot[a];
                        if (s < 0) { mt[a] = nullptr; continue; }
                        CTL_REQUIRE(s < n_bases && bases[s], "plan_run: op %d member %d arg %d uses empty slot %d", k, j, a, s);
                        mt[a] = (char*)bases[s] + mo.off[a];
                    }
                    memcpy(&ds[j], mo.i, sizeof(ctl_conv));
                    sp[j] = mo.i[24];
                    mx[j] = (const float*)mt[0]; mps[j] = (const float*)mt[1]; mpb[j] = (const float*)mt[2]; mdy[j] = (const float*)mt[3];
                    mw[j] = (float*)mt[4]; mb[j] = (float*)mt[5]; mdy2[j] = (const float*)mt[6]; mco[j] = (const float*)mt[7];
                }
                rc = ctl_conv_wgrad_group(nm, ds, sp, mx, mps, mpb, mdy, mdy2, mco, mw, mb, stream);
                k += nm;
                break;
            }
            case CTL_OP_WGRAD_REDUCE:
                memcpy(&d, op.i, sizeof(d));
                rc = ctl_wgrad_reduce(&d, CF(0), CF(1), F(2), op.l[0], op.l[1], op.l[2], op.l[3], F(3), op.i[24], stream);
                break;
            case CTL_OP_PACK:
                rc = ctl_pack_weights(CF(0), F(1), op.i[0], op.i[1], op.i[2], op.l[0], op.l[1], op.l[2], op.l[3], op.i[3], stream);
                break;
            case CTL_OP_BN_FINALIZE:
                rc = ctl_bn_finalize_ex(CF(0), op.i[0], op.i[1], op.l[0], CF(1), CF(2), op.f[0], op.f[1], op.i[2], F(3), F(4),
                                        (int64_t*)t[5], F(6), F(7), F(8), F(9), F(10), NG(op.i[3]), stream);
                break;
            case CTL_OP_BN_REPLAY:
                rc = ctl_bn_replay_running(t[0], F(1), (int64_t*)t[2], (const int64_t*)t[3], op.i[0], op.f[0], stream);
                break;
            case CTL_OP_BN_EVAL:
                rc = ctl_bn_eval_coeffs(op.i[0], CF(0), CF(1), CF(2), CF(3), op.f[0], F(4), F(5), NG(op.i[1]), stream);
                break;
            case CTL_OP_BN_ACT:
                rc = ctl_bn_act_dt(CF(0), CF(1), CF(2), op.f[0], F(3), op.l[0], op.i[0], NG(op.i[1]), (uint32_t)op.i[25], stream);
                break;
            case CTL_OP_BWD_REDUCE:
                rc = ctl_bwd_reduce_dt(op.i[0], CF(0), CF(1), CF(2), CF(3), CF(4), op.f[0], op.l[0], op.i[1], F(5), NG(op.i[2]), (uint32_t)op.i[25], F(6), stream);
                break;
            case CTL_OP_BN_BWD_FINALIZE:
                rc = ctl_bn_bwd_finalize_ex(CF(0), op.i[0], op.l[0], CF(1), CF(2), CF(3), F(4), F(5), F(6), op.i[1], NG(op.i[2]), op.i[3], (uint32_t)op.i[4], stream);
                break;
            case CTL_OP_BWD_APPLY:
                rc = ctl_bwd_apply_dt(op.i[0], CF(0), CF(1), CF(2), CF(3), CF(4), op.f[0], CF(5), op.l[0], op.i[1], F(6), F(7), NG(op.i[2]), (uint32_t)op.i[25], stream);
                break;
            case CTL_OP_CHAN_SUM_FINALIZE:
                rc = ctl_chan_sum_finalize(CF(0), op.i[0], F(1), op.i[1], stream);
                break;
            case CTL_OP_SUMPOOL2:
                rc = ctl_sumpool2_dt(CF(0), F(1), op.i[0], op.i[1], op.i[2], op.i[3], op.i[4], (uint32_t)op.i[25], stream);
                break;
            case CTL_OP_SIGMOID_BWD:
                rc = ctl_sigmoid_bwd(CF(0), CF(1), F(2), op.l[0], stream);
                break;
            case CTL_OP_ZERO: {
                CTL_REQUIRE(t[0] && op.l[0] > 0, "plan_run: op %d ZERO needs a pointer and a size", k);
                ctl_count_launches(1);
                hipError_t e = hipMemsetAsync(t[0], 0, (size_t)op.l[0], (hipStream_t)stream);
                if (e != hipSuccess) CTL_FAIL(CTL_ELAUNCH, "plan_run: memset: %s", hipGetErrorString(e));
                break;
            }
            case CTL_OP_PACK_BATCH:       // i[1] bit 0: bf16 fragments (CTL_DT_BF16 kernels); bit 1: the table has CTL_PACK_X3 records
                rc = (op.i[1] & 1) ? ctl_pack_weights_bf16_batched(CF(0), F(1), (const int64_t*)t[2], op.i[0], op.l[0], stream)
                                   : ctl_pack_weights_batched(CF(0), F(1), (const int64_t*)t[2], op.i[0], op.l[0], stream);
                if (rc == CTL_OK && (op.i[1] & 2)) rc = ctl_pack_weights_x3_batched(CF(0), F(1), (const int64_t*)t[2], op.i[0], op.l[0], stream);
                break;
            case CTL_OP_WGRAD_REDUCE_BATCH:
                rc = ctl_wgrad_reduce_batched(CF(0), F(1), (const int64_t*)t[2], op.i[0], op.l[0], stream);
                break;
            case CTL_OP_DROPOUT2D:      // nn.Dropout2d behind a residual block (encoder_decoder.py:58-66); backward = the same op on dy with the saved pattern
                rc = ctl_dropout2d_dt(t[0], CF(1), (uint64_t)op.l[0], (const int64_t*)t[2], op.f[0], t[3], F(4), op.i[0], op.i[1], op.i[2], (uint32_t)op.i[25], stream);
                break;
            case CTL_OP_COPY: {
                CTL_REQUIRE(t[0] && t[1] && op.l[0] > 0, "plan_run: op %d COPY needs two pointers and a size", k);
                ctl_count_launches(1);
                hipError_t e = hipMemcpyAsync(t[1], t[0], (size_t)op.l[0], hipMemcpyDeviceToDevice, (hipStream_t)stream);
                if (e != hipSuccess) CTL_FAIL(CTL_ELAUNCH, "plan_run: memcpy: %s", hipGetErrorString(e));
                break;
            }
            default:
                CTL_FAIL(CTL_EINVAL, "plan_run: op %d has unknown kind %d", k, op.kind);
        }
#undef F
#undef CF
#undef NG
        if (ptok >= 0) ctl_prof_end(ptok, (hipStream_t)stream);
        if (rc != CTL_OK) {
            char msg[400];
            snprintf(msg, sizeof(msg), "%s", g_err);
            ctl_set_error("plan_run: op %d (kind %d) failed: %s", k, op.kind, msg);
            return rc;
        }
    }
    return CTL_OK;
}
