// X3 half of the fp32-storage convolution family for gfx950 (MI355X): the same tensors, descriptors and epilogues as ctl_conv.hip, the
// contraction on v_mfma_f32_16x16x32_bf16 over an exact three-way bf16 split of both operands (ctl_conv_x3_stage.h has the arithmetic
// contract and the LDS layout; ctl_conv_igemm.h the kernel, shared with the fp32-MFMA instantiations).  Selected per launch by
// CTL_DT_X3 in ctl_conv.dt; eligible: 3x3 / 4x4 / 2x2 kernels with cin and cout multiples of 16 (the 1x1 convs and the <= 4-channel
// first layers are HBM-bound and stay on the fp32 pipe).
#include "ctl_common.h"
// Phase timers (variant builds only: tools/build_variant.sh tm3 "-DCTL_TIMING_X3" ctl_conv_x3.hip; read with ctl_debug_timing_x3): s_memtime deltas
// summed over every wave: [0] prefetch issue, [1] MFMA phase, [2] barrier after the reads, [3] staging (vmcnt wait + prologue + split + ds_write),
// [4] barrier after the writes, [5] epilogue, [6] steps, [7] setup, [8] wave span, [9] realtime span (100 MHz)
#ifdef CTL_TIMING_X3
#define CTL_TM_WAVES 65536
__device__ unsigned long long ctl_tm3[CTL_TM_WAVES][10];
#define TM_DECL unsigned long long tm_prev = __builtin_amdgcn_s_memtime(), tm_acc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; \
                const unsigned long long tm_t0 = tm_prev, tm_r0 = __builtin_amdgcn_s_memrealtime();
#define TM(i) { const unsigned long long tm_now = __builtin_amdgcn_s_memtime(); tm_acc[i] += tm_now - tm_prev; tm_prev = tm_now; }
#define TM_COUNT(i) { tm_acc[i] += 1; }
#ifdef CTL_TIMING_X3_VMCNT      // slot 9 then holds the vmcnt wait of the staging phase instead of the realtime span
#define CTL_TM3_REAL
#else
#define CTL_TM3_REAL tm_acc[9] = __builtin_amdgcn_s_memrealtime() - tm_r0;
#endif
#define TM_FLUSH { const unsigned w_ = ((blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z)) * 8 + (threadIdx.x >> 6)) % CTL_TM_WAVES; \
                   tm_acc[8] = __builtin_amdgcn_s_memtime() - tm_t0; CTL_TM3_REAL \
                   if ((threadIdx.x & 63) == 0) { _Pragma("unroll") for (int i_ = 0; i_ < 10; ++i_) ctl_tm3[w_][i_] += tm_acc[i_]; } }
extern "C" int ctl_debug_timing_x3(unsigned long long* out12) {
    static unsigned long long host[CTL_TM_WAVES][10];
    if (hipMemcpyFromSymbol(host, HIP_SYMBOL(ctl_tm3), sizeof(host)) != hipSuccess) return -1;
    for (int i = 0; i < 12; ++i) out12[i] = 0;       // [10] max span over waves, [11] number of waves that ran
    for (int w = 0; w < CTL_TM_WAVES; ++w) {
        for (int i = 0; i < 10; ++i) out12[i] += host[w][i];
        if (host[w][8] > out12[10]) out12[10] = host[w][8];
        if (host[w][8]) out12[11] += 1;
    }
    for (int w = 0; w < CTL_TM_WAVES; ++w) for (int i = 0; i < 10; ++i) host[w][i] = 0;
    return hipMemcpyToSymbol(HIP_SYMBOL(ctl_tm3), host, sizeof(host)) == hipSuccess ? 0 : -1;
}
#endif
#include "ctl_conv_igemm.h"

// ------------------------------------------------------------------------------------------------ weight packing
// Same table records as pack_weights_batched_kernel (modes 0-3 | CTL_PACK_X3); records without CTL_PACK_X3 are skipped here, records
// with it are skipped by the fp32 / bf16 pack kernels.  One thread = one uint32 = two bf16 of
// dst[cot][fragment][chunk g][split][lane][8]: lane (co = lane & 15, q = lane >> 4), element j -> tap 2f + (q >> 1), ci = 16 g + 8 (q & 1) + j.
__device__ static float x3_pack_value(const float* __restrict__ src, const int64_t* __restrict__ r, int mode, int co, int ci, int kh, int kw) {
    const int cout = (int)r[2], cin = (int)r[3], ks = (int)r[4], flip = (int)r[5];
    if (co >= cout || ci >= cin) return 0.f;
    float v = 0.f;
    if (mode == 1) {            // 4x4 stride-2 kernel of sumpool2(conv3x3^T(.)): sums of 3x3 taps (see pack_weights_batched_kernel)
        for (int a = 0; a < 2; ++a) {
            const int sh = a + 2 - kh;
            if (sh < 0 || sh > 2) continue;
            for (int b = 0; b < 2; ++b) {
                const int sw = b + 2 - kw;
                if (sw < 0 || sw > 2) continue;
                v += src[co * r[6] + ci * r[7] + sh * r[8] + sw * r[9]];
            }
        }
    } else if (mode == 2) {     // phase of a 3x3 conv on a nearest-upsampled input as a 2x2 conv on the stored input
        const int a = flip >> 1, b = flip & 1;
        const int h0 = (kh == 0) ? 0 : (a ? 2 : 1), h1 = (kh == 0) ? (a ? 1 : 0) : 2;
        const int w0 = (kw == 0) ? 0 : (b ? 2 : 1), w1 = (kw == 0) ? (b ? 1 : 0) : 2;
        for (int sh = h0; sh <= h1; ++sh)
            for (int sw = w0; sw <= w1; ++sw) v += src[co * r[6] + ci * r[7] + sh * r[8] + sw * r[9]];
    } else if (mode == 3) {     // phase of the data gradient of a stride-2 3x3 conv
        const int a = flip >> 1, b = flip & 1;
        const int sh = a ? (kh == 0 ? 2 : 0) : (kh == 0 ? 1 : -1);
        const int sw = b ? (kw == 0 ? 2 : 0) : (kw == 0 ? 1 : -1);
        if (sh >= 0 && sw >= 0) v = src[co * r[6] + ci * r[7] + sh * r[8] + sw * r[9]];
    } else {
        if (flip) { kh = ks - 1 - kh; kw = ks - 1 - kw; }
        v = src[co * r[6] + ci * r[7] + kh * r[8] + kw * r[9]];
    }
    return v;
}
__global__ void pack_weights_x3_batched_kernel(const float* __restrict__ params, float* __restrict__ wpack, const int64_t* __restrict__ table) {
    const int64_t* r = table + (int64_t)blockIdx.y * 12;
    if (!(r[11] & CTL_PACK_X3)) return;
    const int mode = (int)(r[11] & ~(int64_t)CTL_PACK_X3);
    const int cout = (int)r[2], cin = (int)r[3], ks = (int)r[4];
    const int g_chunks = (cin + 15) / 16, taps = ks * ks, nfrag = (taps + 1) / 2;
    const int64_t total = (int64_t)((cout + 15) / 16) * nfrag * g_chunks * 64 * 4;      // uint32 words per split
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const float* src = params + r[0];
    unsigned* dst = reinterpret_cast<unsigned*>(wpack + r[1]);
    const int jp = idx & 3, lane = (idx >> 2) & 63;
    int64_t rest = idx >> 8;
    const int g = rest % g_chunks;
    rest /= g_chunks;
    const int f = rest % nfrag;
    const int cot = rest / nfrag;
    const int co = cot * 16 + (lane & 15), q = lane >> 4;
    const int tap = 2 * f + (q >> 1);
    float a = 0.f, b = 0.f;
    if (tap < taps) {
        const int ci = g * 16 + (q & 1) * 8 + jp * 2;
        a = x3_pack_value(src, r, mode, co, ci, tap / ks, tap % ks);
        b = x3_pack_value(src, r, mode, co, ci + 1, tap / ks, tap % ks);
    }
    const int64_t base = ((((int64_t)cot * nfrag + f) * g_chunks + g) * 3) * 256 + lane * 4 + jp;
#pragma unroll
    for (int sp = 0; sp < 3; ++sp) dst[base + sp * 256] = x3_pack2(x3_part(a, sp), x3_part(b, sp));
}
extern "C" size_t ctl_conv_wpack_floats_x3(int32_t cin, int32_t cout, int32_t ks) {
    return (size_t)ctl_cdiv(cout, 16) * ((ks * ks + 1) / 2) * ctl_cdiv(cin, 16) * 3 * 256;
}
extern "C" int ctl_pack_weights_x3_batched(const float* params, float* wpack, const int64_t* table, int32_t n_rec, int64_t max_total, ctl_stream stream) {
    CTL_REQUIRE(params && wpack && table && n_rec > 0 && max_total > 0, "pack_weights_x3_batched: bad arguments");
    // `max_total` = the largest float count of a record's destination (3 planes of uint32 words): a third of it covers every record's words per split
    pack_weights_x3_batched_kernel<<<dim3((unsigned)ctl_cdiv64(max_total, 3 * 256) + 1, (unsigned)n_rec), dim3(256), 0, (hipStream_t)stream>>>(params, wpack, table);
    CTL_LAUNCH_CHECK("pack_weights_x3_batched");
    return CTL_OK;
}

// ------------------------------------------------------------------------------------------------ host side
struct conv3_call {
    const ctl_conv* d; ctl_conv_cfg c;
    const float *x, *wpack, *bias, *pro_scale, *pro_shift, *res, *res_scale, *res_shift, *res2, *x2;
    float *y, *stats_partial, *pool, *xout;
    hipStream_t stream;
    bool query;
    int grid_x;
    int stat_rows_per_block;      // 4 for the producer / consumer form (one statistics row per consumer wave)
};

// The producer / consumer form (ctl_conv_igemm.h, PC): one 512-thread block per CU.  Taken when a block gets at least CTL_X3_PC_MIN_STEPS
// (tile, chunk) steps: its pipeline costs one exposed load + staging at the head of a block.
#ifndef CTL_X3_PC_DEFAULT
#define CTL_X3_PC_DEFAULT 1
#endif
#ifndef CTL_X3_PC_MIN_STEPS
#define CTL_X3_PC_MIN_STEPS 4
#endif
// (which launches: the 3x3 stride-1 convs and data gradients on plain inputs -- 4.9 of the 8.6 ms this family takes per step; the other
//  kernel shapes keep the single-role form until they are instantiated here)
static bool conv3_pc_shape_ok(const ctl_conv* d) {
    const int f = d->epi_flags & (CTL_EPI_RES | CTL_EPI_ACCUM | CTL_EPI_BNBWD | CTL_EPI_TAILBWD);
    const bool epi_ok = f == 0 || (f == CTL_EPI_BNBWD && d->cout % 16 == 0 && d->epi_act == CTL_ACT_NONE);
    return d->ks == 3 && d->stride == 1 && d->in_mode == CTL_IN_PLAIN && d->nsub == 1 && epi_ok;
}
template <int KS, int S, int MODE, int MT, int TW, int NT, int EPI, bool X2>
constexpr bool conv3_pc_ok() { return S == 1 && KS == 3 && MODE == CTL_IN_PLAIN && (EPI == 0 || EPI == 3); }
template <int KS, int S, int MODE, int MT, int TW, int NT, int EPI, bool X2>
static bool conv3_go_pc(conv3_call& a) {
    if constexpr (conv3_pc_ok<KS, S, MODE, MT, TW, NT, EPI, X2>()) {
        if (!a.c.pc) return false;
        const ctl_conv* d = a.d;
        const int ntiles = d->n * a.c.tiles_h * a.c.tiles_w;
        const int grid_x = ctl_conv_grid_x(ntiles, (a.c.cot / NT) * d->nsub, 1);
        a.grid_x = grid_x;
        a.stat_rows_per_block = 4;
        if (a.query) return true;
        const dim3 grid((unsigned)grid_x, (unsigned)(a.c.cot / NT), (unsigned)d->nsub);
        conv_igemm_kernel<KS, S, MODE, MT, TW, NT, EPI, X2, true, true><<<grid, dim3(512), 0, a.stream>>>(
            *d, a.x, a.wpack, a.bias, a.pro_scale, a.pro_shift, a.res, a.res_scale, a.res_shift, a.y, a.stats_partial, a.c.tiles_h,
            a.c.tiles_w, a.c.g, (int64_t)ctl_conv_wpack_floats_x3(d->cin, d->cout, d->ks), ntiles, a.res2, a.x2, a.pool, a.xout);
        return true;
    }
    return false;
}
template <int KS, int S, int MODE, int MT, int TW, int NT, int EPI, bool X2 = false>
static void conv3_go(conv3_call& a) {
    if constexpr (!X2 && ((KS == 3 && S == 1 && MODE == CTL_IN_PLAIN) || (KS == 4 && S == 2)) && EPI != 2) {
        if (a.d->pro_affine == 2) { conv3_go<KS, S, MODE, MT, TW, NT, EPI, true>(a); return; }
    }
    a.stat_rows_per_block = 1;
    if (conv3_go_pc<KS, S, MODE, MT, TW, NT, EPI, X2>(a)) return;
    static int occ = 0;
    if (!occ) {
        int n = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, conv_igemm_kernel<KS, S, MODE, MT, TW, NT, EPI, X2, true>, 256, 0) != hipSuccess || n < 1) {
            (void)hipGetLastError();
            n = 2;
        }
        occ = n;
    }
    const ctl_conv* d = a.d;
    const int ntiles = d->n * a.c.tiles_h * a.c.tiles_w;
    a.grid_x = ctl_conv_grid_x(ntiles, (a.c.cot / NT) * d->nsub, occ);
    if (a.query) return;
    const dim3 grid((unsigned)a.grid_x, (unsigned)(a.c.cot / NT), (unsigned)d->nsub);
    conv_igemm_kernel<KS, S, MODE, MT, TW, NT, EPI, X2, true><<<grid, dim3(256), 0, a.stream>>>(
        *d, a.x, a.wpack, a.bias, a.pro_scale, a.pro_shift, a.res, a.res_scale, a.res_shift, a.y, a.stats_partial, a.c.tiles_h,
        a.c.tiles_w, a.c.g, (int64_t)ctl_conv_wpack_floats_x3(d->cin, d->cout, d->ks), ntiles, a.res2, a.x2, a.pool, a.xout);
}
template <int KS, int S, int MODE>
constexpr bool conv3_tail_ok() { return KS == 2 || (KS == 3 && S == 1 && MODE == CTL_IN_ZINS2); }
template <int KS, int S, int MODE, int MT, int TW>
static void conv3_go_nt(conv3_call& a) {
    constexpr bool NT2 = !(S == 2 && KS >= 3);          // (conv3_pick_cfg: the stride-2 3x3 / 4x4 forms run one cout tile per block)
    const bool epi = (a.d->epi_flags & (CTL_EPI_RES | CTL_EPI_ACCUM | CTL_EPI_BNBWD)) != 0;
    const bool two = NT2 && a.c.nt == 2;
    auto go = [&](auto E) {
        constexpr int EPI = decltype(E)::value;
        if constexpr (NT2) { if (two) { conv3_go<KS, S, MODE, MT, TW, 2, EPI>(a); return; } }
        conv3_go<KS, S, MODE, MT, TW, 1, EPI>(a);
    };
    if constexpr (conv3_tail_ok<KS, S, MODE>()) {
        if (a.d->epi_flags & CTL_EPI_TAILBWD) { go(std::integral_constant<int, 2>{}); return; }
    }
    if constexpr (KS == 3 && S == 1 && MODE == CTL_IN_PLAIN) {
        if ((a.d->epi_flags & (CTL_EPI_RES | CTL_EPI_ACCUM | CTL_EPI_BNBWD)) == CTL_EPI_BNBWD && a.d->cout % 16 == 0 && a.d->epi_act == CTL_ACT_NONE) {
            go(std::integral_constant<int, 3>{});
            return;
        }
    }
    if (epi) go(std::integral_constant<int, 1>{}); else go(std::integral_constant<int, 0>{});
}
template <int KS, int S, int MODE>
static void conv3_go_tile(conv3_call& a) {
    if (S == 1 && a.c.mt == 4 && a.c.tw == 32) conv3_go_nt<KS, S, MODE, (S == 1 ? 4 : 2), (S == 1 ? 32 : 16)>(a);
    else if (a.c.mt == 2) conv3_go_nt<KS, S, MODE, 2, 16>(a);
    else conv3_go_nt<KS, S, MODE, 1, 16>(a);
}
static int conv3_dispatch(conv3_call& a) {
    const int k = a.d->ks, s = a.d->stride, m = a.d->in_mode;
    if (k == 3 && s == 1 && m == CTL_IN_PLAIN) conv3_go_tile<3, 1, CTL_IN_PLAIN>(a);
    else if (k == 3 && s == 1 && m == CTL_IN_UP2) conv3_go_tile<3, 1, CTL_IN_UP2>(a);
    else if (k == 3 && s == 1 && m == CTL_IN_ZINS2) conv3_go_tile<3, 1, CTL_IN_ZINS2>(a);
    else if (k == 3 && s == 2) conv3_go_tile<3, 2, CTL_IN_PLAIN>(a);
    else if (k == 2 && s == 2) conv3_go_tile<2, 2, CTL_IN_PLAIN>(a);
    else if (k == 2 && s == 1) conv3_go_tile<2, 1, CTL_IN_PLAIN>(a);
    else if (k == 4 && s == 2) conv3_go_nt<4, 2, CTL_IN_PLAIN, 1, 16>(a);
    else CTL_FAIL(CTL_EUNSUPPORTED, "conv_forward(x3): no kernel for ks/stride/in_mode %d/%d/%d", k, s, m);
    return CTL_OK;
}

// The X3 images are 1.5x the fp32 ones and the weight fragments come in three planes: the stride-2 forms (17- / 18-row input tiles, 8 tap
// pairs of a 4x4 kernel) keep two resident blocks per CU with one cout tile per block, the 4x4 kernel with the 4x16-pixel tile on top
static int conv3_pick_cfg(const ctl_conv* d, ctl_conv_cfg* c) {
    const int rc = ctl_conv_pick_cfg(d, c, 0);
    if (rc != CTL_OK) return rc;
    // Producer / consumer form: ONE block per CU, so the tile is the largest one that still gives every CU a block (fewer halo bytes and
    // fewer barriers per MFMA), and a block must get CTL_X3_PC_MIN_STEPS (tile, chunk) steps to pay for its pipeline's head.
    static const int pc_on = ctl_tune_int("CTL_X3_PC", CTL_X3_PC_DEFAULT), min_steps = ctl_tune_int("CTL_X3_PC_MIN_STEPS", CTL_X3_PC_MIN_STEPS);
    // (measured per layer, profiles/r5_pc_conv_layers.txt: 128 -> 128 @32^2 31.3 vs 35.9 us, @16^2 14.5 vs 17.7, 64 -> 64 @64^2 n32 59.4 vs 70.0;
    //  16 -> 16 @256^2 43.8 vs 38.7, 32 -> 16 @128^2 24.3 vs 22.7: with few channels the staging's vector instructions and the matrix
    //  instructions of ONE wave pair per SIMD do not fill the issue port as well as three single-role waves do -> cin >= 64 only)
    static const int pc_min_g = ctl_tune_int("CTL_X3_PC_MIN_G", 4);
    if (pc_on && conv3_pc_shape_ok(d) && c->g >= pc_min_g && c->nt == 2) {
        static const int tiles[3][2] = {{4, 32}, {2, 16}, {1, 16}};
        const int other = (c->cot / c->nt) * d->nsub;
        for (int i = 0; i < 3; ++i) {
            const int mt = tiles[i][0], tw = tiles[i][1], th = 4 * mt * 16 / tw;
            if (tw == 32 && d->wout < 32) continue;
            const int64_t ntiles = (int64_t)d->n * ctl_cdiv(d->hout, th) * ctl_cdiv(d->wout, tw);
            if (ntiles * other < ctl_num_cus() && i < 2) continue;       // (a smaller tile fills more CUs)
            const int grid_x = ctl_conv_grid_x((int)ntiles, other, 1);
            // Round 6 (tools/r6_pc_gate.py, profiles/r6_pc_conv_gate.txt: 14 layer shapes x n = 10 / 16 / 32 / 40, both forms): the one-block-per-CU
            // pipeline needs EQUAL shares -- every block the same number of tiles, or at most one.  With 1.25 ... 3.75 tiles per block (the
            // 40-slice inference volumes: 240 / 480 / 80 tiles over 64 / 128 blocks) the launch ends in a round that part of the chip sits out,
            // and three resident single-role blocks per CU level that out better: 1.03-1.15x, 128 -> 128 @12^2 n40 1.50x (VERDICT r5 #3:
            // config 5 18.9 -> 18.0 k slices/s).  Every layer of the bs16 256^2 training step has equal shares: its launches are unchanged.
            // (Also rejecting equal shares of 2-3 tiles, which lose 2-5 % per launch in isolation, made the step 0.7 % SLOWER: 14.41 -> 14.52 ms.)
            static const int gate = ctl_tune_int("CTL_X3_PC_GATE", 1);
            const bool shares_ok = !gate || ntiles <= grid_x || ntiles % grid_x == 0;
            if (shares_ok && ntiles * c->g >= (int64_t)min_steps * grid_x) {
                c->pc = 1; c->mt = mt; c->tw = tw; c->th = th;
                c->tiles_h = ctl_cdiv(d->hout, th); c->tiles_w = ctl_cdiv(d->wout, tw);
            }
            break;
        }
    }
    // (the fp32 family's tile choice is kept for stride 1: forcing the 8x32-pixel tile wherever it gives 512 blocks is 7-9 % faster per launch on the
    //  n = 32 layers alone -- tools/bench_conv.py, CTL_BENCH_X3=1 -- and worth nothing in the step: 15.69 vs 15.65 ms same-box)
    if (d->stride == 2 && d->ks >= 3) {
        c->nt = 1;
        if (d->ks == 4 && c->mt != 1) {
            c->mt = 1; c->tw = 16; c->th = 4;
            c->tiles_h = ctl_cdiv(d->hout, c->th);
            c->tiles_w = ctl_cdiv(d->wout, c->tw);
        }
    }
    return CTL_OK;
}

int ctl_conv_x3_ok(const ctl_conv* d) {
    // (cout 4 / 8 / 12: one padded cout tile -- the fp32 pipe pays the padding 16x dearer; the STN's first-layer data gradient 16 -> 4)
    return d->cin % 16 == 0 && (d->cout % 16 == 0 || (d->cout < 16 && d->cout % 4 == 0)) && d->ks >= 2 && d->in_mode != CTL_IN_C4 && !(d->dt & (CTL_DT_BF16 | CTL_DT_X16 | CTL_DT_Y16 | CTL_DT_RES16));
}
int ctl_conv_x3_stats_blocks(const ctl_conv* d) {
    conv3_call a = {};
    a.d = d;
    if (!ctl_conv_x3_ok(d) || conv3_pick_cfg(d, &a.c) != CTL_OK) return -1;
    a.query = true;
    if (conv3_dispatch(a) != CTL_OK) return -1;
    return a.grid_x * d->nsub * a.stat_rows_per_block;
}
// (argument checks: ctl_conv_forward_ex, which hands over here when ctl_conv.dt has CTL_DT_X3)
int ctl_conv_forward_x3(const ctl_conv* d, const float* x, const float* wpack, const float* bias, const float* pro_scale, const float* pro_shift,
                        const float* res, const float* res_scale, const float* res_shift, const float* res2, const float* x2, float* y,
                        float* stats_partial, float* pool, float* xout, ctl_stream stream) {
    CTL_REQUIRE(ctl_conv_x3_ok(d), "conv_forward(x3): CTL_DT_X3 needs fp32-stored tensors, cin %% 16 == 0, cout %% 16 == 0 (or 4 / 8 / 12) and a 2x2 / 3x3 / 4x4 kernel "
                                   "(got cin %d, cout %d, ks %d, in_mode %d, dt %d)", d->cin, d->cout, d->ks, d->in_mode, d->dt);
    CTL_REQUIRE(!pool, "conv_forward(x3): `pool` belongs to the 1x1 hosts, which stay on the fp32 pipe");
    conv3_call a = {};
    a.d = d;
    int rc = conv3_pick_cfg(d, &a.c);
    if (rc != CTL_OK) return rc;
    a.x = x; a.wpack = wpack; a.bias = bias; a.pro_scale = pro_scale; a.pro_shift = pro_shift; a.res = res; a.res_scale = res_scale;
    a.res_shift = res_shift; a.res2 = res2; a.x2 = x2; a.y = y; a.stats_partial = stats_partial; a.pool = pool; a.xout = xout;
    a.stream = (hipStream_t)stream;
    a.query = true;                  // (which form will run: the producer / consumer launches carry their own profiling id)
    rc = conv3_dispatch(a);
    if (rc != CTL_OK) return rc;
    a.query = false;
    const int ptok = ctl_prof_begin(a.stat_rows_per_block == 4 ? "conv_igemm_x3pc" : "conv_igemm_x3", d, &a.c, a.c.nt, a.stream);
    rc = conv3_dispatch(a);
    if (ptok >= 0) ctl_prof_end(ptok, a.stream);
    if (rc != CTL_OK) return rc;
    CTL_LAUNCH_CHECK("conv_forward(x3)");
    return CTL_OK;
}
