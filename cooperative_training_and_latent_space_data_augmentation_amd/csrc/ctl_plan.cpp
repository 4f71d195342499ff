// Plan executor: runs an array of ctl_op on one HIP stream (one C call per network pass), plus the library's
// version / error plumbing.  No allocation, no synchronisation: safe under hipStreamBeginCapture.
//
// Tensor slots per op kind (index into op.slot/op.off):
//   CONV              0 x  1 wpack  2 bias  3 pro_scale  4 pro_shift  5 res  6 res_scale  7 res_shift  8 y  9 stats_partial
//   WGRAD             0 x  1 pro_scale  2 pro_shift  3 dy  4 w_partial  5 b_partial
//   WGRAD_REDUCE      0 w_partial  1 b_partial  2 dw  3 dbias            i[24]=accumulate  l[0..3]=s_co,s_ci,s_kh,s_kw
//   PACK              0 src  1 dst                                       i[0..3]=cout,cin,ks,flip  l[0..3]=strides
//   BN_FINALIZE       0 partial 1 gamma 2 beta 3 running_mean 4 running_var 5 nbt 6 scale 7 shift 8 save_mean 9 save_invstd
//                                                                        i[0..2]=blocks,c,update_running l[0]=count f[0]=eps f[1]=momentum
//   BN_EVAL           0 gamma 1 beta 2 running_mean 3 running_var 4 scale 5 shift      i[0]=c f[0]=eps
//   BN_ACT            0 x 1 scale 2 shift 3 y                            i[0]=c l[0]=pixels f[0]=slope
//   BWD_REDUCE        0 dy 1 act_src 2 bn_src 3 scale 4 shift 5 partial 6 ds (mode 0, optional)  i[0]=mode i[1]=c l[0]=pixels f[0]=slope
//   BN_BWD_FINALIZE   0 partial 1 gamma 2 save_mean 3 save_invstd 4 coef 5 dgamma 6 dbeta   i[0]=c i[1]=accumulate l[0]=count
//   BWD_APPLY         0 dy 1 act_src 2 bn_src 3 scale 4 shift 5 coef 6 ds 7 dx   i[0]=mode i[1]=c l[0]=pixels f[0]=slope
//   CHAN_SUM_FINALIZE 0 partial 1 out                                    i[0]=c i[1]=accumulate
//   SUMPOOL2          0 dup 1 dx                                         i[0..4]=n,h,w,c,accumulate
//   SIGMOID_BWD       0 dy 1 y 2 dx                                      l[0]=count
//   ZERO              0 ptr                                              l[0]=bytes
//   COPY              0 src 1 dst                                        l[0]=bytes
//   PACK_BATCH        0 params 1 wpack 2 table                           i[0]=n_rec i[1]=bf16 fragments l[0]=max_total
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

int ctl_prof_begin(const char* kind, const ctl_conv* d, const ctl_conv_cfg* c, int nt, hipStream_t stream) {
    if (!g_prof_on) return -1;
    char id[128];
    snprintf(id, sizeof(id), "%s<ks%d,s%d,in%d,mt%d,tw%d,nt%d>", kind, d->ks, d->stride, d->in_mode, c->mt, c->tw, nt);
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
    if (d->epi_flags & CTL_EPI_RES) r.bytes += ((d->dt & CTL_DT_RES16) ? 2.0 : 4.0) * pix * d->cout;
    if (d->epi_flags & CTL_EPI_ACCUM) r.bytes += out_b;
    (void)hipEventRecord(r.a, stream);
    g_prof.push_back(r);
    return (int)g_prof.size() - 1;
}
// non-conv plan ops: bracketed only for the timeline dump (CTL_PROF_TIMELINE), no algorithmic work attached
static int prof_begin_op(int kind, hipStream_t stream) {
    static const bool want = getenv("CTL_PROF_TIMELINE") != nullptr;
    if (!g_prof_on || !want || !g_prof_filter.empty()) return -1;
    ProfRec r;
    char id[32];
    snprintf(id, sizeof(id), "op%d", kind);
    r.id = id;
    r.stream = (void*)stream;
    r.flops = r.bytes = 0.0;
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
    if (const char* path = getenv("CTL_PROF_TIMELINE")) tl = fopen(path, "w");
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
extern "C" int ctl_version(void) { return 1; }
extern "C" const char* ctl_last_error(void) { return g_err; }
extern "C" size_t ctl_sizeof_op(void) { return sizeof(ctl_op); }
extern "C" size_t ctl_sizeof_conv(void) { return sizeof(ctl_conv); }

static_assert(sizeof(ctl_conv) == 24 * 4, "ctl_conv must be 24 32-bit words (it is embedded in ctl_op.i[0..23]; i[24] and i[26] are taken)");

// Side lane: ops with i[26] == 1 (weight gradients and their batched reduction: off the critical dgrad chain) run on a
// library-owned second stream so that their launches fill the ramp-up / tail bubbles of the main chain.  Fork = event
// recorded on the main stream right before the side op (it then sees everything the main stream produced so far);
// join = the main stream waits for the side stream once, at the end of the plan.  Opt-in: CTL_SIDE_STREAM=1 (eager) / 2 (+ captured).
// Every main stream has its own side stream (the two launch chains of a training step do not serialise each other's side work).
// Under stream capture (hipGraph mode) the fork / join events become graph dependencies; nothing may be CREATED while a capture is
// running, so a lane is only used there if an eager plan on the same stream created it (and enough fork events) before -- the graph
// module's eager warm-up step does -- and the op runs inline otherwise.
namespace {
struct side_lane {
    hipStream_t side = nullptr;
    hipEvent_t join = nullptr;
    std::vector<hipEvent_t> forks;
};
std::map<hipStream_t, side_lane> g_lanes;
int g_side_enabled = -1;
}  // namespace

static side_lane* lane_of(hipStream_t main, bool may_create) {
    auto it = g_lanes.find(main);
    if (it != g_lanes.end()) return &it->second;
    if (!may_create) return nullptr;
    side_lane l;
    if (hipStreamCreateWithFlags(&l.side, hipStreamNonBlocking) != hipSuccess) return nullptr;
    if (hipEventCreateWithFlags(&l.join, hipEventDisableTiming) != hipSuccess) return nullptr;
    return &(g_lanes[main] = l);
}
static hipEvent_t fork_event(side_lane* l, size_t k, bool may_create) {
    while (l->forks.size() <= k) {
        hipEvent_t e = nullptr;
        if (!may_create || hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return nullptr;
        l->forks.push_back(e);
    }
    return l->forks[k];
}

static void side_mode_init() {
    if (g_side_enabled >= 0) return;
    // 0 (default) = off; 1 = eager plans; 2 = also inside a stream capture.  Round-2 measurements (profiles/README.md): in eager mode the
    // lanes were worth +1-5 % on some boxes and -1-4 % on others, and merely HAVING the two extra streams alive cost the hipGraph replays
    // of the same process 1.5 % (fp32 18.08 -> 18.35 ms): streams share a few hardware queues, and which streams share one decides how
    // well the two launch chains overlap (GPU_MAX_HW_QUEUES=8 without lanes put both chains of the replay on ONE queue: 31.6 ms).
    // Captured lanes are correct but slow (~165 extra cross-stream edges per replay: fp32 17.9 -> 22.3 ms).
    const char* e = getenv("CTL_SIDE_STREAM");
    g_side_enabled = e ? atoi(e) : 0;
    if (g_side_enabled < 0 || g_side_enabled > 2) g_side_enabled = 0;
}
extern "C" int ctl_plan_side_lanes(int32_t mode) {
    side_mode_init();
    const int prev = g_side_enabled;
    if (mode >= 0 && mode <= 2) g_side_enabled = mode;
    return prev;
}

extern "C" int ctl_plan_run(const ctl_op* ops, int32_t n_ops, void* const* bases, int32_t n_bases, ctl_stream stream_) {
    CTL_REQUIRE(ops && bases && n_ops >= 0, "plan_run: null arguments");
    side_mode_init();
    size_t forks = 0;
    bool side_used = false;
    bool side_ok = g_side_enabled >= 1;
    bool capturing = false;
    side_lane* lane = nullptr;
    if (side_ok) {
        hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing((hipStream_t)stream_, &st) != hipSuccess) side_ok = false;
        capturing = st != hipStreamCaptureStatusNone;
        if (capturing && g_side_enabled < 2) side_ok = false;
        if (side_ok) lane = lane_of((hipStream_t)stream_, !capturing);
        if (!lane) side_ok = false;
    }
    // ---- fused finalizes (opt-in per op: BN_FINALIZE / BN_BWD_FINALIZE with i[4] == 1 directly behind the kernel that writes their
    // partial rows; the buffer of the partial slot starts with CTL_FIN_HEADER_BYTES of zero-initialised header = the record table).
    // The finalize op is folded into its producer: rec_of[k] = table slot used by op k, skip[k + 1] drops the stand-alone launch.
    // MEASURED on MI355X (profiles/README.md, round 2): OFF by default.  The arrival + agent-scope acquire + re-read of the rows by the
    // last block costs ~10 us per fused launch (fp32 step 18.3 -> 21.6 ms) -- more than the ~7 us of a stand-alone finalize kernel and
    // its boundary; cross-block hand-offs inside a launch are as expensive as a kernel boundary on this part.  CTL_FUSE_FINALIZE=1 opts in.
    static const bool fuse_enabled = [] { const char* e = getenv("CTL_FUSE_FINALIZE"); return e && atoi(e) != 0; }();
    // CONSUMER-side variant (CTL_FUSE_CONSUMER, see ctl_bn_consume in ctl_common.h): conv k (statistics) -> BN_FINALIZE k+1 -> conv k+2
    // whose prologue (role 1) or residual affine (role 2) are exactly that finalize's scale / shift: the finalize moves into the first
    // blocks of conv k+2.  No hand-off behind the producer's tiles; the wait sits where conv k+2's blocks wait for their first loads anyway.
    // MEASURED on MI355X (round 2): OFF by default.  With an agent-scope acquire fence in every waiting block the step lost 2 ms (the fence
    // invalidates the XCD's L2: +20 us per launch); with agent-scope LOADS of the coefficients instead it is 94 launches fewer per step
    // and still 0.2-0.35 ms slower (fp32 18.3 -> 18.5 ms, bf16 12.0 -> 12.3 ms): the serial writer -> flag -> poll -> load chain at the head
    // of the consumer costs what the stand-alone launch and its boundary cost.  CTL_FUSE_CONSUMER=1 opts in.
    static const bool consumer_enabled = [] { const char* e = getenv("CTL_FUSE_CONSUMER"); return CTL_CONSUMER_FINALIZE && e && atoi(e) != 0; }();
    thread_local std::vector<int> rec_of;
    thread_local std::vector<char> skip, role_of;
    rec_of.assign((size_t)n_ops, -1);
    skip.assign((size_t)n_ops + 1, 0);
    role_of.assign((size_t)n_ops, 0);
    ctl_bn_fin recs[CTL_FIN_MAX_RECS];
    int n_rec = 0;
    char* table = nullptr;
    auto resolve = [&](const ctl_op& o, int a) -> void* {
        const int sl = o.slot[a];
        return (sl < 0 || sl >= n_bases || !bases[sl]) ? nullptr : (void*)((char*)bases[sl] + o.off[a]);
    };
    auto same = [](const ctl_op& x, int ax, const ctl_op& y, int ay) { return x.slot[ax] >= 0 && x.slot[ax] == y.slot[ay] && x.off[ax] == y.off[ay]; };
    for (int32_t k = 0; consumer_enabled && k + 2 < n_ops && n_rec < CTL_FIN_MAX_RECS; ++k) {
        const ctl_op& a = ops[k];
        const ctl_op& b = ops[k + 1];
        const ctl_op& c = ops[k + 2];
        if (a.kind != CTL_OP_CONV || b.kind != CTL_OP_BN_FINALIZE || c.kind != CTL_OP_CONV || b.i[4] != 1 || !same(a, 9, b, 0)) continue;
        ctl_conv dc;
        memcpy(&dc, c.i, sizeof(dc));
        if (dc.epi_flags & CTL_EPI_BNBWD) continue;
        int role = 0;
        if (dc.pro_affine && same(c, 3, b, 6) && same(c, 4, b, 7) && dc.cin == b.i[1]) role = 1;
        else if ((dc.epi_flags & CTL_EPI_RES) && !dc.pro_affine && same(c, 6, b, 6) && same(c, 7, b, 7) && dc.cout == b.i[1] && dc.cout >= 4 &&
                 (dc.groups > 1 ? dc.groups : 1) * dc.cout <= 256 /* CTL_PRO_MAX */) role = 2;
        if (!role || (dc.groups > 1 ? dc.groups : 1) != (b.i[3] > 0 ? b.i[3] : 1)) continue;
        ctl_bn_fin& r = recs[n_rec];
        memset(&r, 0, sizeof(r));
        r.gamma = (const float*)resolve(b, 1); r.beta = (const float*)resolve(b, 2);
        r.running_mean = (float*)resolve(b, 3); r.running_var = (float*)resolve(b, 4);
        r.num_batches_tracked = (int64_t*)resolve(b, 5);
        r.scale = (float*)resolve(b, 6); r.shift = (float*)resolve(b, 7); r.save_mean = (float*)resolve(b, 8); r.save_invstd = (float*)resolve(b, 9);
        r.count = b.l[0]; r.eps = b.f[0]; r.momentum = b.f[1]; r.update_running = b.i[2];
        r.role = role; r.rows = b.i[0]; r.partial = (const float*)resolve(b, 0);
        const int pslot = a.slot[9];
        CTL_REQUIRE(pslot < n_bases && bases[pslot], "plan_run: op %d: empty partial slot", k);
        CTL_REQUIRE(table == nullptr || table == (char*)bases[pslot], "plan_run: fused finalizes must share one partial slot");
        table = (char*)bases[pslot];
        rec_of[k + 2] = n_rec++;
        role_of[k + 2] = (char)role;
        skip[k + 1] = 1;
    }
    for (int32_t k = 0; fuse_enabled && k + 1 < n_ops && n_rec < CTL_FIN_MAX_RECS; ++k) {
        const ctl_op& a = ops[k];
        const ctl_op& b = ops[k + 1];
        if (b.i[4] != 1 || skip[k + 1] || rec_of[k] >= 0) continue;
        int pslot = -1;
        if (a.kind == CTL_OP_CONV && b.kind == CTL_OP_BN_FINALIZE && a.slot[9] >= 0 && a.slot[9] == b.slot[0] && a.off[9] == b.off[0]) {
            ctl_conv dd;
            memcpy(&dd, a.i, sizeof(dd));
            ctl_conv_cfg cc;
            if (!(dd.epi_flags & CTL_EPI_STATS) || (dd.epi_flags & CTL_EPI_BNBWD) || ctl_conv_pick_cfg(&dd, &cc, 0) != CTL_OK ||
                cc.cot / cc.nt > CTL_FIN_MAX_Y)
                continue;                                   // (more block rows of output-channel tiles than a record has counter sets)
            ctl_bn_fin& r = recs[n_rec];
            memset(&r, 0, sizeof(r));
            r.gamma = (const float*)resolve(b, 1); r.beta = (const float*)resolve(b, 2);
            r.running_mean = (float*)resolve(b, 3); r.running_var = (float*)resolve(b, 4);
            r.num_batches_tracked = (int64_t*)resolve(b, 5);
            r.scale = (float*)resolve(b, 6); r.shift = (float*)resolve(b, 7); r.save_mean = (float*)resolve(b, 8); r.save_invstd = (float*)resolve(b, 9);
            r.count = b.l[0]; r.eps = b.f[0]; r.momentum = b.f[1]; r.update_running = b.i[2];
            pslot = a.slot[9];
        } else if (a.kind == CTL_OP_BWD_REDUCE && b.kind == CTL_OP_BN_BWD_FINALIZE && a.i[0] != 2 && b.i[3] == 0 && a.slot[5] >= 0 &&
                   a.slot[5] == b.slot[0] && a.off[5] == b.off[0]) {
            memset(&recs[n_rec], 0, sizeof(recs[n_rec]));          // (backward: arguments go by value, only the slot's counter is used)
            recs[n_rec].gamma = recs[n_rec].beta = (const float*)resolve(b, 1);
            recs[n_rec].scale = recs[n_rec].shift = (float*)resolve(b, 4);
            recs[n_rec].count = 1;
            pslot = a.slot[5];
        } else {
            continue;
        }
        CTL_REQUIRE(pslot < n_bases && bases[pslot], "plan_run: op %d: empty partial slot", k);
        CTL_REQUIRE(table == nullptr || table == (char*)bases[pslot], "plan_run: fused finalizes must share one partial slot");
        table = (char*)bases[pslot];
        rec_of[k] = n_rec++;
        skip[k + 1] = 1;
    }
    if (n_rec > 0) {
        int rc0 = ctl_bn_fin_table_write(table, recs, n_rec, stream_);
        if (rc0 != CTL_OK) return rc0;
    }
    for (int32_t k = 0; k < n_ops; ++k) {
        if (skip[k]) continue;
        const ctl_op& op = ops[k];
        ctl_stream stream = stream_;
        if (side_ok && op.i[26] == 1) {
            hipEvent_t ev = fork_event(lane, forks, !capturing);
            if (ev) {                                    // (no event left under capture: the op stays on the main stream)
                ++forks;
                CTL_REQUIRE(hipEventRecord(ev, (hipStream_t)stream_) == hipSuccess &&
                            hipStreamWaitEvent(lane->side, ev, 0) == hipSuccess, "plan_run: fork failed");
                stream = (ctl_stream)lane->side;
                side_used = true;
            }
        }
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
        const int ptok = (op.kind == CTL_OP_CONV || op.kind == CTL_OP_WGRAD) ? -1 : prof_begin_op(op.kind, (hipStream_t)stream);
        switch (op.kind) {
            case CTL_OP_CONV:
                memcpy(&d, op.i, sizeof(d));
                rc = ctl_conv_forward_fin(&d, CF(0), CF(1), CF(2), CF(3), CF(4), CF(5), CF(6), CF(7), F(8), F(9),
                                          rec_of[k] >= 0 ? (void*)(table + (size_t)rec_of[k] * CTL_FIN_REC_BYTES) : nullptr, role_of[k], stream);
                break;
            case CTL_OP_WGRAD:
                memcpy(&d, op.i, sizeof(d));
                rc = ctl_conv_wgrad(&d, CF(0), CF(1), CF(2), CF(3), F(4), F(5), stream);
                break;
            case CTL_OP_WGRAD_REDUCE:
                memcpy(&d, op.i, sizeof(d));
                rc = ctl_wgrad_reduce(&d, CF(0), CF(1), F(2), op.l[0], op.l[1], op.l[2], op.l[3], F(3), op.i[24], stream);
                break;
            case CTL_OP_PACK:
                rc = ctl_pack_weights(CF(0), F(1), op.i[0], op.i[1], op.i[2], op.l[0], op.l[1], op.l[2], op.l[3], op.i[3], stream);
                break;
            case CTL_OP_BN_FINALIZE:
                rc = ctl_bn_finalize(CF(0), op.i[0], op.i[1], op.l[0], CF(1), CF(2), op.f[0], op.f[1], op.i[2], F(3), F(4),
                                     (int64_t*)t[5], F(6), F(7), F(8), F(9), NG(op.i[3]), stream);
                break;
            case CTL_OP_BN_EVAL:
                rc = ctl_bn_eval_coeffs(op.i[0], CF(0), CF(1), CF(2), CF(3), op.f[0], F(4), F(5), NG(op.i[1]), stream);
                break;
            case CTL_OP_BN_ACT:
                rc = ctl_bn_act_dt(CF(0), CF(1), CF(2), op.f[0], F(3), op.l[0], op.i[0], NG(op.i[1]), (uint32_t)op.i[25], stream);
                break;
            case CTL_OP_BWD_REDUCE:
                if (rec_of[k] >= 0) {       // + the BN_BWD_FINALIZE op behind it (its arguments by value, the slot's first counter)
                    const ctl_op& fo = ops[k + 1];
                    ctl_bnb_fin bf;
                    memset(&bf, 0, sizeof(bf));
                    bf.gamma = (const float*)resolve(fo, 1); bf.save_mean = (const float*)resolve(fo, 2); bf.save_invstd = (const float*)resolve(fo, 3);
                    bf.coef = (float*)resolve(fo, 4); bf.dgamma = (float*)resolve(fo, 5); bf.dbeta = (float*)resolve(fo, 6);
                    bf.counter = (uint32_t*)(table + (size_t)rec_of[k] * CTL_FIN_REC_BYTES + 128);      // the slot's first counter set
                    bf.count = fo.l[0]; bf.accumulate = fo.i[1];
                    rc = ctl_bwd_reduce_fin(op.i[0], CF(0), CF(1), CF(2), CF(3), CF(4), op.f[0], op.l[0], op.i[1], F(5), NG(op.i[2]), (uint32_t)op.i[25], &bf, F(6), stream);
                } else {
                    rc = ctl_bwd_reduce_fin(op.i[0], CF(0), CF(1), CF(2), CF(3), CF(4), op.f[0], op.l[0], op.i[1], F(5), NG(op.i[2]), (uint32_t)op.i[25], nullptr, F(6), stream);
                }
                break;
            case CTL_OP_BN_BWD_FINALIZE:
                rc = ctl_bn_bwd_finalize(CF(0), op.i[0], op.l[0], CF(1), CF(2), CF(3), F(4), F(5), F(6), op.i[1], NG(op.i[2]), op.i[3], stream);
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
            case CTL_OP_PACK_BATCH:       // i[1] = 1: bf16 fragments (CTL_DT_BF16 kernels)
                rc = op.i[1] ? ctl_pack_weights_bf16_batched(CF(0), F(1), (const int64_t*)t[2], op.i[0], op.l[0], stream)
                             : ctl_pack_weights_batched(CF(0), F(1), (const int64_t*)t[2], op.i[0], op.l[0], stream);
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
    if (side_used && !capturing) {
        CTL_REQUIRE(hipEventRecord(lane->join, lane->side) == hipSuccess &&
                    hipStreamWaitEvent((hipStream_t)stream_, lane->join, 0) == hipSuccess, "plan_run: join failed");
    } else if (side_used) {
        // Under capture the join is an explicit graph dependency (the side lane's last nodes become predecessors of the main stream's
        // next node), NOT an event wait: a non-origin stream that waits on an event of a stream it forked is entered into that
        // stream's list of parallel capture streams as well, and the runtime's EndCapture then recurses A -> side -> A -> ... until the
        // stack ends (ROCm 7.0, hip::Stream::EndCapture; found with rocgdb).
        hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
        unsigned long long id = 0;
        hipGraph_t graph = nullptr;
        const hipGraphNode_t* deps = nullptr;
        size_t ndeps = 0;
        CTL_REQUIRE(hipStreamGetCaptureInfo_v2(lane->side, &st, &id, &graph, &deps, &ndeps) == hipSuccess &&
                    st == hipStreamCaptureStatusActive, "plan_run: the side lane left the capture");
        if (ndeps > 0)
            CTL_REQUIRE(hipStreamUpdateCaptureDependencies((hipStream_t)stream_, const_cast<hipGraphNode_t*>(deps), ndeps,
                                                           hipStreamAddCaptureDependencies) == hipSuccess, "plan_run: join (capture) failed");
    }
    return CTL_OK;
}
