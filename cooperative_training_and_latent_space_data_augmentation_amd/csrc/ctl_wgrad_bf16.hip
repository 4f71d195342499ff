// bf16 weight-gradient kernels for gfx950 (MI355X), the second half of the bf16 family (see ctl_conv_bf16.hip for the arithmetic contract).
#include "ctl_conv_bf16_common.h"

// ------------------------------------------------------------------------------------------------ weight gradient (bf16 operands)
// dW[tap][ci][co] = sum_pixels x_virtual[pixel*S + tap - pad][ci] * dy[pixel][co]:  D[ci][co] += A[ci][k = pixel] B[pixel][co], K = 32 pixels.
// Both operands want 8 PIXELS of one channel per lane, i.e. the transpose of the [pixel][16 channel] LDS tiles -- which is what
// gfx950's ds_read_b64_tr_b16 delivers for free: per 16-lane group it reads 4 rows (pixels) x 16 columns (channels) of 16-bit elements and
// hands lane i column i.  Lane 4q'+p' of a group supplies the address of row q', columns 4p'..4p'+3; every lane supplies its own row
// address, so a tap shift is just another pixel address.  k-block kb of a tile = tile rows 2kb, 2kb+1 (TW = 16): lane group gk takes row
// 2kb + (gk >> 1), columns 8(gk & 1) .. +7 (two transposed reads of 4 pixels each).  Wave w owns k-block w; the four waves are summed
// through LDS at the end exactly like the fp32 kernel, same partial layout, same deterministic split reduction.
typedef short s16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ bf16x8 tr_read8(const unsigned char* a0, const unsigned char* a1) {
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(a0));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(a1));
    typedef short s16x8 __attribute__((ext_vector_type(8)));
    return __builtin_bit_cast(bf16x8, s16x8{lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w});
}

// DY2: the output gradient is the virtual BatchNorm-backward result  A * dy + B * dy2 + C  (coefficients [group][3][cout] as the
// finalize writes them; dy = g, dy2 = the BatchNorm input): the `apply` pass runs in this staging (see XStage16 X2)
// BB: both operands are known to be STORED as bf16 with whole 16-channel chunks (every layer but the network boundaries): the staging is
// straight-line code.  With the storage types as run-time flags the compiler keeps the staged units in SCRATCH memory and waits for every
// global load right where it is issued (seen in the ISA: buffer_load, s_waitcnt vmcnt(0), scratch_store) -- no prefetch left at all.
// XK / DYK: the same compile-time storage kinds as XStage16's X16C for x and for dy (1 bf16 with whole chunks, 2 fp32 quads, 3 fp32 single
// channel, 0 run time); BB = both bf16.  The network-boundary layers are (fp32 x, bf16 dy) first layers and (bf16 x, fp32 dy) output layers.
// The kernel body takes its place in the launch as arguments (BX of GX pixel splits, cin chunk BY, cout tile pair BZ): the single launch passes
// blockIdx / gridDim, the grouped launch (below) the member's own coordinates.
template <int KS, int S, int MODE, int MT, int NTW, bool DY2, int XK, int DYK>
__device__ __forceinline__ void wgrad16_body(const ctl_conv& d, const void* __restrict__ x,
                                             const float* __restrict__ pro_scale, const float* __restrict__ pro_shift,
                                             const void* __restrict__ dy, const void* __restrict__ dy2,
                                             const float* __restrict__ dy_coef, float* __restrict__ w_partial,
                                             float* __restrict__ b_partial, int tiles_h, int tiles_w, int ntiles,
                                             int cin_p, int cout_p, const int BX, const int BY, const int BZ, const int GX) {
    constexpr int TW = 16;
    using G = Geom<KS, S, MT, TW>;
    using XS = XStage16<KS, S, MODE, MT, TW, XK>;
    constexpr int TAPS = KS * KS;
    constexpr int KB = G::TP / 32;                       // k-blocks per tile: 4 (8x16 tile) or 2 (4x16)
    constexpr int XT_ALLOC = XS::XT_BYTES + 16;
    constexpr int DYT_BYTES = NTW * G::TP * 32;
    constexpr int RED_BYTES = 4 * NTW * 256 * 4;
    constexpr int MAIN_BYTES = (XT_ALLOC + DYT_BYTES > RED_BYTES) ? (XT_ALLOC + DYT_BYTES) : RED_BYTES;
    __shared__ __attribute__((aligned(16))) unsigned char smem[MAIN_BYTES + (DY2 ? 5 : 2) * CTL_PRO_MAX * 4];
    unsigned char* xt = smem;
    unsigned char* dyt = smem + XT_ALLOC;
    float* cf_scale = reinterpret_cast<float*>(smem + MAIN_BYTES);
    float* cf_shift = cf_scale + CTL_PRO_MAX;
    float* cd = cf_shift + CTL_PRO_MAX;                  // DY2: A | B | C, [group][cout] each

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int p = lane & 15, q = lane >> 4;
    const int g = BY;
    const int cot0 = BZ * NTW;
    const int group_n = d.n / (d.groups > 1 ? d.groups : 1);
    const bool dy16 = DYK == 1 || (DYK == 0 && (d.dt & CTL_DT_Y16) != 0);
    const int des = dy16 ? 2 : 4;

    f32x4 acc[TAPS][NTW];
#pragma unroll
    for (int a = 0; a < TAPS; ++a)
#pragma unroll
        for (int t = 0; t < NTW; ++t) acc[a][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    float bsum[NTW];
#pragma unroll
    for (int t = 0; t < NTW; ++t) bsum[t] = 0.f;

    const __amdgpu_buffer_rsrc_t rx = ctl_rsrc(x, (int64_t)d.n * d.hin * d.win * d.cin * ((XK == 1 || (XK == 0 && (d.dt & CTL_DT_X16))) ? 2 : 4));
    const __amdgpu_buffer_rsrc_t rdy = ctl_rsrc(dy, (int64_t)d.n * d.hout * d.wout * d.cout * des);
    const __amdgpu_buffer_rsrc_t rdy2 = DY2 ? ctl_rsrc(dy2, (int64_t)d.n * d.hout * d.wout * d.cout * 2) : rdy;
    XS xs;
    xs.init(d);
    // dy tile [cout tile t][pixel][16 ch] bf16: units of 8 channels
    constexpr int DU = G::TP * NTW * 2, ND = (DU + 255) / 256;
    u32x4 dv0[ND], dv1[ND];
    int drel[ND], drc[ND], dlds[ND];
    bool dq1[ND];
    unsigned dmask = 0;                                  // DY2: units of the tile in flight that lie inside the image
    int dco = 0;                                         // DY2: first channel of this thread's units (the same for all of them: 256 % (2 NTW) == 0)
#pragma unroll
    for (int i = 0; i < ND; ++i) {
        const int u = tid + i * 256;
        const int pix = u / (NTW * 2), rest = u - pix * (NTW * 2);
        const int t = rest >> 1, h = rest & 1;
        const int pr = pix / TW, pc = pix % TW;
        const int co = (cot0 + t) * 16 + h * 8;
        const bool in = u < DU && co < d.cout;
        drc[i] = in ? (pr | (pc << 16)) : 0x7fff7fff;
        drel[i] = in ? ((pr * d.wout + pc) * d.cout + co) * des : CTL_OOB;
        dlds[i] = (u < DU) ? ((t * G::TP + pix) * 32 + h * 16) : DYT_BYTES - 16;      // (units past the tile cannot exist: DU is a multiple of 256 or ND covers it)
        dq1[i] = co + 4 < d.cout;
        if (i == 0) dco = co < d.cout ? co : 0;
    }
    auto dyload = [&](int n, int ho0, int wo0) {
        const int tb = ((n * d.hout + ho0) * d.wout + wo0) * d.cout * des;
        const u32x4 z = {0u, 0u, 0u, 0u};
        dmask = 0;
#pragma unroll
        for (int i = 0; i < ND; ++i) {
            const bool ok = (unsigned)(ho0 + (drc[i] & 0xffff)) < (unsigned)d.hout && (unsigned)(wo0 + (drc[i] >> 16)) < (unsigned)d.wout;
            const int vo = ok ? (tb + drel[i]) : CTL_OOB;
            if constexpr (DY2) {
                dv0[i] = ctl_bload4u(rdy, vo, 0); dv1[i] = ctl_bload4u(rdy2, vo, 0);
                dmask |= (ok && drel[i] != CTL_OOB) ? (1u << i) : 0u;
                continue;
            }
            if constexpr (DYK == 1) { dv0[i] = ctl_bload4u(rdy, vo, 0); continue; }
            if constexpr (DYK == 2) { dv0[i] = ctl_bload4u(rdy, vo, 0); dv1[i] = dq1[i] ? ctl_bload4u(rdy, vo + 16, 0) : z; continue; }
            if constexpr (DYK == 3) { dv0[i] = u32x4{__builtin_amdgcn_raw_buffer_load_b32(rdy, vo, 0, 0), 0u, 0u, 0u}; dv1[i] = z; continue; }
            if (dy16) dv0[i] = ctl_bload4u(rdy, vo, 0);
            else if (d.cout >= 4) { dv0[i] = ctl_bload4u(rdy, vo, 0); dv1[i] = dq1[i] ? ctl_bload4u(rdy, vo + 16, 0) : z; }
            else { dv0[i] = u32x4{__builtin_amdgcn_raw_buffer_load_b32(rdy, vo, 0, 0), 0u, 0u, 0u}; dv1[i] = z; }
        }
    };
    auto dystore = [&](int goff) {
        if constexpr (DY2) {
            const float* cc = cd + goff + dco;
            const f32x4 a0 = *reinterpret_cast<const f32x4*>(cc), a1 = *reinterpret_cast<const f32x4*>(cc + 4);
            const f32x4 b0 = *reinterpret_cast<const f32x4*>(cc + CTL_PRO_MAX), b1 = *reinterpret_cast<const f32x4*>(cc + CTL_PRO_MAX + 4);
            const f32x4 c0 = *reinterpret_cast<const f32x4*>(cc + 2 * CTL_PRO_MAX), c1 = *reinterpret_cast<const f32x4*>(cc + 2 * CTL_PRO_MAX + 4);
            const u32x4 zero = {0u, 0u, 0u, 0u};
#pragma unroll
            for (int i = 0; i < ND; ++i) {
                if (tid + i * 256 < DU) {
                    const f32x4 glo = unpack_bf16x4(dv0[i].x, dv0[i].y), ghi = unpack_bf16x4(dv0[i].z, dv0[i].w);
                    const f32x4 ulo = unpack_bf16x4(dv1[i].x, dv1[i].y), uhi = unpack_bf16x4(dv1[i].z, dv1[i].w);
                    u32x4 pk = pack_bf16x8(a0 * glo + b0 * ulo + c0, a1 * ghi + b1 * uhi + c1);
                    if (!((dmask >> i) & 1u)) pk = zero;          // pixels past the image contribute nothing (C alone would)
                    *reinterpret_cast<u32x4*>(dyt + dlds[i]) = pk;
                }
            }
            return;
        }
        if constexpr (DYK == 1) {
#pragma unroll
            for (int i = 0; i < ND; ++i) {
                if (tid + i * 256 < DU) *reinterpret_cast<u32x4*>(dyt + dlds[i]) = dv0[i];
            }
            return;
        }
        if constexpr (DYK == 2 || DYK == 3) {
#pragma unroll
            for (int i = 0; i < ND; ++i) {
                if (tid + i * 256 < DU)
                    *reinterpret_cast<u32x4*>(dyt + dlds[i]) = pack_bf16x8(__builtin_bit_cast(f32x4, dv0[i]), __builtin_bit_cast(f32x4, dv1[i]));
            }
            return;
        }
#pragma unroll
        for (int i = 0; i < ND; ++i) {
            if (tid + i * 256 < DU)
                *reinterpret_cast<u32x4*>(dyt + dlds[i]) =
                    dy16 ? dv0[i] : pack_bf16x8(__builtin_bit_cast(f32x4, dv0[i]), __builtin_bit_cast(f32x4, dv1[i]));
        }
    };
    // transposed-read addresses of this lane: block row q' = (lane & 15) >> 2 (pixel), 8-byte column chunk p' = lane & 3
    const int qp = (lane & 15) >> 2, pp = lane & 3;
    const int krow = (q >> 1), kcol0 = 8 * (q & 1);     // tile row (within the k-block's two rows) and first column of this lane group
    TileWalk cur;
    cur.init(BX, GX, tiles_h, tiles_w);
    if (BX < ntiles) {
        xs.load(rx, d, cur.n, cur.th * G::TH, cur.tw * TW, g);
        dyload(cur.n, cur.th * G::TH, cur.tw * TW);
    }
    if (d.pro_affine || DY2) {
        if (d.pro_affine)
            for (int i = tid; i < (d.groups > 1 ? d.groups : 1) * d.cin; i += 256) { cf_scale[i] = pro_scale[i]; cf_shift[i] = pro_shift[i]; }
        if constexpr (DY2) {
            for (int i = tid; i < (d.groups > 1 ? d.groups : 1) * d.cout; i += 256) {
                const int gi = i / d.cout, ch = i - gi * d.cout;
                cd[i] = dy_coef[(gi * 3 + 0) * d.cout + ch]; cd[CTL_PRO_MAX + i] = dy_coef[(gi * 3 + 1) * d.cout + ch];
                cd[2 * CTL_PRO_MAX + i] = dy_coef[(gi * 3 + 2) * d.cout + ch];
            }
        }
        __syncthreads();
    }
    if (BX < ntiles) {
        xs.store(xt, d, g, cf_scale, cf_shift, (cur.n / group_n) * d.cin);
        dystore((cur.n / group_n) * d.cout);
    }
    __syncthreads();
    for (int tile = BX; tile < ntiles; tile += GX) {
        const bool has_next = tile + GX < ntiles;
        if (has_next) {
            cur.next();
            xs.load(rx, d, cur.n, cur.th * G::TH, cur.tw * TW, g);
            dyload(cur.n, cur.th * G::TH, cur.tw * TW);
        }
        for (int kb = wave; kb < KB; kb += 4) {          // (4x16 tiles: two k-blocks, waves 2 and 3 only stage; 16x16 tiles: two k-blocks per wave)
            const int tr = kb * 2 + krow;                // tile row of this lane group's 8 pixels
            bf16x8 bf[NTW];
#pragma unroll
            for (int t = 0; t < NTW; ++t) {
                const unsigned char* b0 = dyt + ((t * G::TP + tr * TW + kcol0 + qp) * 32) + pp * 8;
                bf[t] = tr_read8(b0, b0 + 4 * 32);
                float s = 0.f;
#pragma unroll
                for (int j = 0; j < 8; ++j) s += (float)bf[t][j];
                bsum[t] += s;
            }
            // the A operand of tap+1 is requested before the MFMAs of tap; the empty asm pins that order (the scheduler otherwise sinks
            // every transposed read to its use: read, wait, MFMA, nine times per tile -- see the forward kernel)
            auto a_operand = [&](int tap) -> bf16x8 {
                const int kh = tap / KS, kw = tap % KS;
                const int c0 = G::ldscol((kcol0 + qp) * S + kw);          // consecutive output columns are consecutive LDS columns (stride 2: de-interleaved)
                const unsigned char* a0 = xt + (((tr * S + kh) * G::IWP + c0) * 32) + pp * 8;
                return tr_read8(a0, a0 + 4 * 32);
            };
            bf16x8 af[2];
            af[0] = a_operand(0);
#pragma unroll
            for (int tap = 0; tap < TAPS; ++tap) {
                if (tap + 1 < TAPS) af[(tap + 1) & 1] = a_operand(tap + 1);
                asm volatile("" : "+v"(af[tap & 1]) : : "memory");
#pragma unroll
                for (int t = 0; t < NTW; ++t) acc[tap][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[tap & 1], bf[t], acc[tap][t], 0, 0, 0);
            }
        }
        ctl_barrier_lds_reads_done();
        if (has_next) {
            xs.store(xt, d, g, cf_scale, cf_shift, (cur.n / group_n) * d.cin);
            dystore((cur.n / group_n) * d.cout);
        }
        ctl_barrier_lds_writes_done();
    }
    __syncthreads();

    // ---------------- sum the four waves through LDS and write this split's partial (layout of the fp32 kernel)
    float* red = reinterpret_cast<float*>(smem);
    constexpr int TAP_FLOATS = 4 * NTW * 256;
    constexpr int TPR = (MAIN_BYTES / 4 / TAP_FLOATS) < TAPS ? (MAIN_BYTES / 4 / TAP_FLOATS) : TAPS;
    static_assert(TPR >= 1, "reduction scratch");
    const int64_t split_base = (int64_t)BX * TAPS * cin_p * cout_p;
#pragma unroll
    for (int tap0 = 0; tap0 < TAPS; tap0 += TPR) {
        if (tap0 > 0) ctl_barrier_lds_reads_done();
#pragma unroll
        for (int tp = 0; tp < TPR; ++tp) {
            const int tap = tap0 + tp;
            if (tap < TAPS) {
#pragma unroll
                for (int t = 0; t < NTW; ++t) {
                    float* r0 = red + tp * TAP_FLOATS + ((wave * NTW + t) * 4) * 64 + lane;
                    r0[0] = acc[tap][t].x; r0[64] = acc[tap][t].y; r0[128] = acc[tap][t].z; r0[192] = acc[tap][t].w;
                }
            }
        }
        ctl_barrier_lds_writes_done();
#pragma unroll
        for (int tp = 0; tp < TPR; ++tp) {
            const int tap = tap0 + tp;
            if (tap < TAPS) {
#pragma unroll
                for (int e0 = 0; e0 < NTW * 256; e0 += 256) {
                    const int e = e0 + tid;
                    const int t = e >> 8, r = (e >> 6) & 3, l = e & 63;
                    float v = 0.f;
#pragma unroll
                    for (int w = 0; w < 4; ++w) v += red[tp * TAP_FLOATS + ((w * NTW + t) * 4 + r) * 64 + l];
                    const int co = (cot0 + t) * 16 + (l & 15);
                    const int ci = g * 16 + (l >> 4) * 4 + r;
                    if (co < cout_p) w_partial[split_base + ((int64_t)tap * cin_p + ci) * cout_p + co] = v;
                }
            }
        }
    }
    if (g == 0 && b_partial != nullptr) {
        ctl_barrier_lds_reads_done();
#pragma unroll
        for (int t = 0; t < NTW; ++t) {
            float v = bsum[t];
            v += __shfl_xor(v, 16);
            v += __shfl_xor(v, 32);
            if (q == 0) red[(wave * NTW + t) * 16 + p] = v;
        }
        ctl_barrier_lds_writes_done();
        if (tid < NTW * 16) {
            float v = 0.f;
#pragma unroll
            for (int w = 0; w < 4; ++w) v += red[w * NTW * 16 + tid];
            const int co = cot0 * 16 + tid;
            if (co < cout_p) b_partial[(int64_t)BX * cout_p + co] = v;
        }
    }
}

template <int KS, int S, int MODE, int MT, int NTW, bool DY2 = false, int XK = 0, int DYK = 0>
__global__ __launch_bounds__(256) void conv_wgrad_bf16_kernel(const ctl_conv d, const void* __restrict__ x,
                                                               const float* __restrict__ pro_scale, const float* __restrict__ pro_shift,
                                                               const void* __restrict__ dy, const void* __restrict__ dy2,
                                                               const float* __restrict__ dy_coef, float* __restrict__ w_partial,
                                                               float* __restrict__ b_partial, int tiles_h, int tiles_w, int ntiles,
                                                               int cin_p, int cout_p) {
    wgrad16_body<KS, S, MODE, MT, NTW, DY2, XK, DYK>(d, x, pro_scale, pro_shift, dy, dy2, dy_coef, w_partial, b_partial, tiles_h, tiles_w, ntiles, cin_p, cout_p,
                                                    (int)blockIdx.x, (int)blockIdx.y, (int)blockIdx.z, (int)gridDim.x);
}

// ---- grouped launch (round 5): up to CTL_WG16_MAX weight gradients of ONE kernel instantiation stacked along blockIdx.x, each with its own
// pixel-split count (ctl_wgrad_group_plan deals the chip's resident blocks to the members in proportion to their work: a member of a group
// walks more tiles per block, writes fewer partial sums, and its blocks start while its predecessor's drain).  With the split count of a launch
// of its own a member's partial sums are bit for bit those of ctl_conv_wgrad_ex.  The bf16 step is launch-bound (~900 launches in 9.5 ms), its
// weight gradients take 6-20 us each; stacked with their own full grids they measured only 10 % less serial time (the fixed cost is per BLOCK:
// coefficient tables, pipeline ramp, cross-wave reduction, partial write), hence the proportional splits.
#define CTL_WG16_MAX 8
struct wg16_member {
    ctl_conv d;
    const void *x, *dy, *dy2;
    const float *pro_scale, *pro_shift, *dy_coef;
    float *w_partial, *b_partial;
    int tiles_h, tiles_w, ntiles, cin_p, cout_p;
    int gx, gy, gz, block0;          // the member's grid and its first block in the stacked launch
};
struct wg16_group { int n, pad; wg16_member m[CTL_WG16_MAX]; };
template <int KS, int S, int MODE, int MT, int NTW, bool DY2, int XK, int DYK>
__global__ __launch_bounds__(256) void conv_wgrad_bf16_group_kernel(const wg16_group grp) {
    int mi = 0;
#pragma unroll
    for (int i = 1; i < CTL_WG16_MAX; ++i)
        if (i < grp.n && (int)blockIdx.x >= grp.m[i].block0) mi = i;
    const wg16_member& M = grp.m[mi];
    const int local = (int)blockIdx.x - M.block0;
    const int bx = local % M.gx, t = local / M.gx;
    wgrad16_body<KS, S, MODE, MT, NTW, DY2, XK, DYK>(M.d, M.x, M.pro_scale, M.pro_shift, M.dy, M.dy2, M.dy_coef, M.w_partial, M.b_partial, M.tiles_h, M.tiles_w,
                                                    M.ntiles, M.cin_p, M.cout_p, bx, t % M.gy, t / M.gy, M.gx);
}

struct wgrad16_call {
    const ctl_conv* d; ctl_conv_cfg c; int ntw, splits, ntiles, cin_p, cout_p;
    const void *x, *dy, *dy2; const float *pro_scale, *pro_shift, *dy_coef; float *w_partial, *b_partial;
    hipStream_t stream; bool query;
    int key;                       // the instantiation the dispatch ends in (members of a grouped launch must agree)
    int cap;                       // blocks of this instantiation the chip holds at once
    const wg16_group* grp;         // launch this stacked group instead of the single problem
    int grp_blocks;
};
static inline int wg16_key(int ks, int s, int mode, int mt, int ntw, int dy2, int xk, int dyk) {
    return ks | (s << 3) | (mode << 5) | (mt << 7) | (ntw << 10) | (dy2 << 12) | (xk << 13) | (dyk << 15);
}
template <int KS, int S, int MODE, int MT, int NTW, bool DY2, int XK, int DYK>
static void wgrad16_go_f(wgrad16_call& a) {
    static int occ = 0;
    if (!occ) {
        int n = 0;
        // (of the plain instantiation, also for DY2: the split count is queried at plan time from the descriptor alone and sizes the partials)
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, conv_wgrad_bf16_kernel<KS, S, MODE, MT, NTW, false, 0, 0>, 256, 0) != hipSuccess || n < 1) {
            (void)hipGetLastError();
            n = 2;
        }
        occ = n;
    }
    const int par = a.c.g * (a.c.cot / NTW);
    static const int cap = ctl_tune_int("CTL16_WGRAD_SPLITS", 1024);      // tuning hook (measured: 512 -> 768/1024 splits = 28.7 -> 24.5 us on the 16->16 3x3 layer at 256^2)
    int splits = (ctl_num_cus() * (occ < 4 ? occ : 4)) / par;
    if (splits > cap) splits = cap;
    if (splits > a.ntiles) splits = a.ntiles;
    if (splits < 1) splits = 1;
    a.splits = splits;
    a.cap = ctl_num_cus() * (occ < 4 ? occ : 4);
    a.key = wg16_key(KS, S, MODE, MT, NTW, DY2, XK, DYK);
    if (a.query) return;
    if (a.grp) {
        conv_wgrad_bf16_group_kernel<KS, S, MODE, MT, NTW, DY2, XK, DYK><<<dim3((unsigned)a.grp_blocks), dim3(256), 0, a.stream>>>(*a.grp);
        return;
    }
    const dim3 grid((unsigned)splits, (unsigned)a.c.g, (unsigned)(a.c.cot / NTW));
    conv_wgrad_bf16_kernel<KS, S, MODE, MT, NTW, DY2, XK, DYK><<<grid, dim3(256), 0, a.stream>>>(*a.d, a.x, a.pro_scale, a.pro_shift, a.dy, a.dy2, a.dy_coef,
                                                                                        a.w_partial, a.b_partial, a.c.tiles_h, a.c.tiles_w, a.ntiles,
                                                                                        a.cin_p, a.cout_p);
}
template <int KS, int S, int MODE, int MT, int NTW>
static void wgrad16_go(wgrad16_call& a) {
    if constexpr (KS == 3 && S == 1) {      // the two-tensor output gradient: the 3x3 convs of the residual blocks
        if (a.dy2) {      // (bf16-stored dy / dy2 by contract; x is fp32 with 1 or 4 channels in the encoders' first layer)
            if ((a.d->dt & CTL_DT_X16) && a.d->cin % 16 == 0) wgrad16_go_f<KS, S, MODE, MT, NTW, true, 1, 1>(a);
            else if (!(a.d->dt & CTL_DT_X16) && a.d->cin == 1) wgrad16_go_f<KS, S, MODE, MT, NTW, true, 3, 1>(a);
            else if (!(a.d->dt & CTL_DT_X16) && a.d->cin >= 4) wgrad16_go_f<KS, S, MODE, MT, NTW, true, 2, 1>(a);
            else wgrad16_go_f<KS, S, MODE, MT, NTW, true, 0, 1>(a);
            return;
        }
    }
    const ctl_conv* d = a.d;
    const int xk = (d->dt & CTL_DT_X16) ? (d->cin % 16 == 0 ? 1 : 0) : (d->cin == 1 ? 3 : 2);
    const int dk = (d->dt & CTL_DT_Y16) ? (d->cout % 16 == 0 ? 1 : 0) : (d->cout == 1 ? 3 : 2);
    if (xk == 1 && dk == 1) { wgrad16_go_f<KS, S, MODE, MT, NTW, false, 1, 1>(a); return; }
    if constexpr (KS == 3 && S == 1 && MODE == CTL_IN_PLAIN) {      // first layers: fp32 network input, bf16 output gradient
        if (xk == 2 && dk == 1) { wgrad16_go_f<KS, S, MODE, MT, NTW, false, 2, 1>(a); return; }
        if (xk == 3 && dk == 1) { wgrad16_go_f<KS, S, MODE, MT, NTW, false, 3, 1>(a); return; }
    }
    if constexpr (KS == 1 && MODE == CTL_IN_PLAIN) {                // output layers: bf16 input, fp32 gradient of the network output (4 or 1 channels)
        if (xk == 1 && dk == 2) { wgrad16_go_f<KS, S, MODE, MT, NTW, false, 1, 2>(a); return; }
        if (xk == 1 && dk == 3) { wgrad16_go_f<KS, S, MODE, MT, NTW, false, 1, 3>(a); return; }
    }
    wgrad16_go_f<KS, S, MODE, MT, NTW, false, 0, 0>(a);
}
template <int KS, int S, int MODE>
static void wgrad16_go_tile(wgrad16_call& a) {
    if constexpr (S == 1 && (KS == 3 || KS == 1)) {
        if (a.c.mt == 4) { if (a.ntw == 2) wgrad16_go<KS, S, MODE, 4, 2>(a); else wgrad16_go<KS, S, MODE, 4, 1>(a); return; }
    }
    if (a.c.mt == 2) { if (a.ntw == 2) wgrad16_go<KS, S, MODE, 2, 2>(a); else wgrad16_go<KS, S, MODE, 2, 1>(a); }
    else { if (a.ntw == 2) wgrad16_go<KS, S, MODE, 1, 2>(a); else wgrad16_go<KS, S, MODE, 1, 1>(a); }
}
static int wgrad16_dispatch(wgrad16_call& a) {
    const int k = a.d->ks, s = a.d->stride, m = a.d->in_mode == CTL_IN_C4 ? CTL_IN_PLAIN : a.d->in_mode;
    if (k == 3 && s == 1 && m == CTL_IN_PLAIN) wgrad16_go_tile<3, 1, CTL_IN_PLAIN>(a);
    else if (k == 3 && s == 1 && m == CTL_IN_UP2) wgrad16_go_tile<3, 1, CTL_IN_UP2>(a);
    else if (k == 3 && s == 2) wgrad16_go_tile<3, 2, CTL_IN_PLAIN>(a);
    else if (k == 1 && m == CTL_IN_PLAIN) wgrad16_go_tile<1, 1, CTL_IN_PLAIN>(a);
    else if (k == 1 && m == CTL_IN_UP2) wgrad16_go_tile<1, 1, CTL_IN_UP2>(a);
    else if (k == 2 && s == 2) wgrad16_go_tile<2, 2, CTL_IN_PLAIN>(a);
    else CTL_FAIL(CTL_EUNSUPPORTED, "conv_wgrad(bf16): no kernel for this combination");
    return CTL_OK;
}
static int wgrad16_pick(const ctl_conv* d, wgrad16_call* a) {
    CTL_REQUIRE(d->nsub == 1 && d->in_mode != CTL_IN_ZINS2, "wgrad(bf16): nsub must be 1, no zero-insert input");
    CTL_REQUIRE(!(d->dt & CTL_DT_X16) || d->cin % 16 == 0, "wgrad(bf16): bf16-stored x needs cin %% 16 == 0");
    CTL_REQUIRE(!(d->dt & CTL_DT_Y16) || d->cout % 16 == 0, "wgrad(bf16): bf16-stored dy needs cout %% 16 == 0");
    a->d = d;
    int rc = ctl_conv_pick_cfg(d, &a->c, 1);
    if (rc != CTL_OK) return rc;
    // With 16-cycle bf16 MFMAs a tile's matrix work (9 per wave for a 3x3 kernel) is far shorter than the load latency of the next tile,
    // which the single-buffered loop then exposes every time: the large layers take 16x16-pixel tiles (twice the bytes in flight per
    // block, half the barriers per byte).  Measured 16->16 at 256^2: 41.9 us with 8x16 tiles.
    static const int big_ok = ctl_tune_int("CTL16_WGRAD_MT4", 1);
    if (big_ok && d->stride == 1 && d->ks == 3 && d->hout >= 64 && d->wout >= 16) {      // (1x1: 8x16 tiles measured faster)
        a->c.mt = 4; a->c.th = 16;
        a->c.tiles_h = ctl_cdiv(d->hout, 16);
    }
    a->ntw = (a->c.cot >= 2 && a->c.cot % 2 == 0) ? 2 : 1;
    a->ntiles = d->n * a->c.tiles_h * a->c.tiles_w;
    a->cin_p = a->c.g * 16;
    a->cout_p = a->c.cot * 16;
    a->query = true;
    rc = wgrad16_dispatch(*a);
    a->query = false;
    return rc;
}
int ctl_wgrad_bf16_splits(const ctl_conv* d) {
    wgrad16_call a = {};
    return wgrad16_pick(d, &a) == CTL_OK ? a.splits : -1;
}
int ctl_conv_wgrad_bf16(const ctl_conv* d, const void* x, const float* pro_scale, const float* pro_shift, const void* dy, const void* dy2,
                        const float* dy_coef, float* w_partial, float* b_partial, ctl_stream stream) {
    CTL_REQUIRE(!dy2 || (dy_coef && d->ks == 3 && d->stride == 1 && (d->dt & CTL_DT_Y16) && d->cout % 16 == 0 &&
                         (d->groups > 1 ? d->groups : 1) * d->cout <= CTL_PRO_MAX),
                "wgrad(bf16): the two-tensor output gradient needs coefficients, a 3x3 stride-1 conv, bf16-stored dy / dy2 with cout %% 16 == 0 and groups * cout <= %d", CTL_PRO_MAX);
    wgrad16_call a = {};
    int rc = wgrad16_pick(d, &a);
    if (rc != CTL_OK) return rc;
    a.x = x; a.dy = dy; a.dy2 = dy2; a.dy_coef = dy_coef; a.pro_scale = pro_scale; a.pro_shift = pro_shift; a.w_partial = w_partial; a.b_partial = b_partial;
    a.stream = (hipStream_t)stream;
    rc = wgrad16_dispatch(a);
    if (rc != CTL_OK) return rc;
    CTL_LAUNCH_CHECK("conv_wgrad(bf16)");
    return CTL_OK;
}

// ---- grouped launches: class = 0x100 | the instantiation key (ctl_wgrad_group_class), splits = the member's own (ctl_wgrad_group_plan)
int ctl_wgrad_bf16_group_class(const ctl_conv* d, int has_dy2) {
    static const int on = ctl_tune_int("CTL16_WGRAD_GROUP", 1);
    if (!on) return -1;
    if (has_dy2 && !(d->ks == 3 && d->stride == 1 && (d->dt & CTL_DT_Y16) && d->cout % 16 == 0)) return -1;
    wgrad16_call a = {};
    if (wgrad16_pick(d, &a) != CTL_OK) return -1;
    int dummy = 0;
    a.dy2 = has_dy2 ? &dummy : nullptr;       // (only its presence selects the instantiation)
    a.query = true;
    if (wgrad16_dispatch(a) != CTL_OK) return -1;
    return 0x100 | a.key;
}
int ctl_conv_wgrad_bf16_group(int n, const ctl_conv* descs, const int32_t* splits, const void* const* x, const float* const* pro_scale,
                              const float* const* pro_shift, const void* const* dy, const void* const* dy2, const float* const* dy_coef,
                              float* const* w_partial, float* const* b_partial, ctl_stream stream) {
    CTL_REQUIRE(n >= 1 && n <= CTL_WG16_MAX, "conv_wgrad_group(bf16): 1..%d members", CTL_WG16_MAX);
    wg16_group g = {};
    g.n = n;
    wgrad16_call first = {};
    int blocks = 0, key = -1;
    double flops = 0.0, bytes = 0.0;
    for (int i = 0; i < n; ++i) {
        const ctl_conv* d = &descs[i];
        const bool two = dy2 && dy2[i];
        CTL_REQUIRE(x[i] && dy[i] && w_partial[i] && (!d->pro_affine || (pro_scale && pro_shift && pro_scale[i] && pro_shift[i])) && (!two || (dy_coef && dy_coef[i])),
                    "conv_wgrad_group(bf16): member %d misses a tensor", i);
        CTL_REQUIRE(!two || (d->ks == 3 && d->stride == 1 && (d->dt & CTL_DT_Y16) && d->cout % 16 == 0 && (d->groups > 1 ? d->groups : 1) * d->cout <= CTL_PRO_MAX),
                    "conv_wgrad_group(bf16): member %d: the two-tensor output gradient needs a 3x3 stride-1 conv, bf16-stored dy / dy2, groups * cout <= %d", i, CTL_PRO_MAX);
        wgrad16_call a = {};
        int rc = wgrad16_pick(d, &a);
        if (rc != CTL_OK) return rc;
        a.dy2 = two ? dy2[i] : nullptr;
        a.query = true;
        rc = wgrad16_dispatch(a);
        if (rc != CTL_OK) return rc;
        if (i == 0) key = a.key;
        CTL_REQUIRE(a.key == key, "conv_wgrad_group(bf16): member %d is of another class than member 0", i);
        CTL_REQUIRE(splits[i] >= 1 && splits[i] <= a.ntiles, "conv_wgrad_group(bf16): member %d: %d splits of %d tiles", i, splits[i], a.ntiles);
        a.splits = splits[i];
        wg16_member& m = g.m[i];
        m.d = *d;
        m.x = x[i]; m.dy = dy[i]; m.dy2 = a.dy2; m.pro_scale = pro_scale ? pro_scale[i] : nullptr; m.pro_shift = pro_shift ? pro_shift[i] : nullptr;
        m.dy_coef = dy_coef ? dy_coef[i] : nullptr; m.w_partial = w_partial[i]; m.b_partial = b_partial ? b_partial[i] : nullptr;
        m.tiles_h = a.c.tiles_h; m.tiles_w = a.c.tiles_w; m.ntiles = a.ntiles; m.cin_p = a.cin_p; m.cout_p = a.cout_p;
        m.gx = a.splits; m.gy = a.c.g; m.gz = a.c.cot / a.ntw; m.block0 = blocks;
        blocks += m.gx * m.gy * m.gz;
        if (i == 0) first = a;
        const double pix = (double)d->n * d->hout * d->wout;
        flops += 2.0 * pix * d->cout * d->cin * d->ks * d->ks;
        bytes += ((d->dt & CTL_DT_X16) ? 2.0 : 4.0) * d->n * d->hin * d->win * d->cin + ((d->dt & CTL_DT_Y16) ? 2.0 : 4.0) * pix * d->cout * (two ? 2 : 1);
    }
    char kind[64];
    snprintf(kind, sizeof(kind), "conv_wgrad_bf16grp<ks%d,s%d,in%d,mt%d,nt%d%s>", key & 7, (key >> 3) & 3, (key >> 5) & 3, (key >> 7) & 7, (key >> 10) & 3, ((key >> 12) & 1) ? ",x2" : "");
    const int ptok = ctl_prof_begin_raw(kind, flops, bytes, (hipStream_t)stream);
    first.query = false;
    first.grp = &g;
    first.grp_blocks = blocks;
    first.stream = (hipStream_t)stream;
    int rc = wgrad16_dispatch(first);
    if (ptok >= 0) ctl_prof_end(ptok, (hipStream_t)stream);
    if (rc != CTL_OK) return rc;
    CTL_LAUNCH_CHECK("conv_wgrad_group(bf16)");
    return CTL_OK;
}
// pixel splits of the members of one stacked launch: the resident blocks of the instantiation, dealt in proportion to the members' work
int ctl_wgrad_bf16_group_plan(const ctl_conv* descs, int n, int32_t* splits) {
    CTL_REQUIRE(n >= 1 && n <= CTL_WG16_MAX, "wgrad_group_plan(bf16): 1..%d members", CTL_WG16_MAX);
    int64_t work[CTL_WG16_MAX], total = 0;
    int par[CTL_WG16_MAX], ntiles[CTL_WG16_MAX], own[CTL_WG16_MAX], cap = 0;
    for (int i = 0; i < n; ++i) {
        wgrad16_call a = {};
        int rc = wgrad16_pick(&descs[i], &a);
        if (rc != CTL_OK) return rc;
        par[i] = a.c.g * (a.c.cot / a.ntw);
        ntiles[i] = a.ntiles;
        own[i] = a.splits;
        if (a.cap > cap) cap = a.cap;
        work[i] = (int64_t)a.ntiles * par[i];
        total += work[i];
    }
    static const int mode = ctl_tune_int("CTL16_WGRAD_GROUP_SPLITS", 1);      // tuning hook: 0 = every member keeps the splits of a launch of its own
    static const int cap_pct = ctl_tune_int("CTL16_WGRAD_GROUP_CAP", 100);    // tuning hook: blocks of a stacked launch as a percentage of the resident set
    cap = cap * cap_pct / 100;
    for (int i = 0; i < n; ++i) {
        if (!mode) { splits[i] = own[i]; continue; }
        // (rounded UP: a member rounded down would walk up to twice the tiles per block of the others.  Measured per class against the members'
        //  own grids stacked: 3x3 16x16-tile classes 41 -> 37 / 67 -> 61 us, 1x1 24 -> 22, up-sampled input 33 -> 29, the batched
        //  reduction 47 -> 34 us; the stride-2 class goes 40 -> 67 us, keeping its own grids costs the others more than it saves)
        int sp = (int)(((int64_t)cap * work[i] + total * par[i] - 1) / (total * par[i]));
        if (sp < 1) sp = 1;
        if (sp > ntiles[i]) sp = ntiles[i];
        if (sp > own[i]) sp = own[i];
        splits[i] = sp;
    }
    return CTL_OK;
}
