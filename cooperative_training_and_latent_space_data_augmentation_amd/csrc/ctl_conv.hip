// Implicit-GEMM convolution family for gfx950 (MI355X): fp32 NHWC, MFMA f32 16x16x4, LDS-staged input tiles.
//
// GEMM view (computed transposed so that every lane ends up with 4 consecutive output channels of ONE pixel and the
// epilogue is a single coalesced 16-byte store per lane):
//     D^T[co][pixel] += W^T[co][k] * X^T[k][pixel],   k = (tap, ci)
//   MFMA A operand = weights  : lane l -> A[row co = l&15][k-slot l>>4]
//   MFMA B operand = input    : lane l -> B[k-slot l>>4][col pixel = l&15]
//   D                          : lane l holds co = 4*(l>>4)+r (r=0..3) of pixel l&15
// The order of k inside a 16-channel chunk is free as long as A and B agree, so one ds_read_b128 per lane
// (4 consecutive channels of "its" pixel, quarter l>>4 of the chunk) feeds FOUR MFMAs: MFMA j uses component j, i.e.
// k-slot q of MFMA j is channel 16*g + 4*q + j.  Weights are pre-packed in exactly that fragment order
// (ctl_pack_weights), so the A operand is one coalesced 16-byte global load per lane, served from L2.
//
// LDS input tile: [row][col][16 channels] floats for the current 16-channel chunk; 16 consecutive output pixels of a
// row read 1 KiB contiguous -> conflict-free ds_read_b128.  For stride 2 the columns are de-interleaved (even | odd)
// so the same holds.  BatchNorm-apply + LeakyReLU of the producer layer is applied once per element while staging.
#include <stdlib.h>

#include <type_traits>

#include "ctl_common.h"

#include "ctl_conv_common.h"

// Phase timers of the forward kernel (variant builds only: tools/build_variant.sh tm "-DCTL_TIMING"; read with
// ctl_debug_timing).  Sums s_memtime deltas over every wave: [0] prefetch issue, [1] MFMA loop, [2] barrier after the
// reads, [3] staging (vmcnt wait + prologue + ds_write), [4] barrier after the writes, [5] epilogue, [6] steps, [7] setup.
#ifdef CTL_TIMING
#define CTL_TM_WAVES 65536
__device__ unsigned long long ctl_tm[CTL_TM_WAVES][10];      // one slot per wave: no atomics, the host sums
#define TM_DECL unsigned long long tm_prev = __builtin_amdgcn_s_memtime(), tm_acc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; \
                const unsigned long long tm_t0 = tm_prev, tm_r0 = __builtin_amdgcn_s_memrealtime();
#define TM(i) { const unsigned long long tm_now = __builtin_amdgcn_s_memtime(); tm_acc[i] += tm_now - tm_prev; tm_prev = tm_now; }
#define TM_COUNT(i) { tm_acc[i] += 1; }
#define TM_FLUSH { const unsigned w_ = ((blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z)) * 4 + (threadIdx.x >> 6)) % CTL_TM_WAVES; \
                   tm_acc[8] = __builtin_amdgcn_s_memtime() - tm_t0; tm_acc[9] = __builtin_amdgcn_s_memrealtime() - tm_r0; \
                   if ((threadIdx.x & 63) == 0) { _Pragma("unroll") for (int i_ = 0; i_ < 10; ++i_) ctl_tm[w_][i_] += tm_acc[i_]; } }
extern "C" int ctl_debug_timing(unsigned long long* out8) {
    static unsigned long long host[CTL_TM_WAVES][10];       // out8 holds 10 values: [8] s_memtime span, [9] s_memrealtime span (100 MHz)
    if (hipMemcpyFromSymbol(host, HIP_SYMBOL(ctl_tm), sizeof(host)) != hipSuccess) return -1;
    for (int i = 0; i < 12; ++i) out8[i] = 0;       // [10] max span over waves, [11] number of waves that ran
    for (int w = 0; w < CTL_TM_WAVES; ++w) {
        for (int i = 0; i < 10; ++i) out8[i] += host[w][i];
        if (host[w][8] > out8[10]) out8[10] = host[w][8];
        if (host[w][8]) out8[11] += 1;
    }
    for (int w = 0; w < CTL_TM_WAVES; ++w) for (int i = 0; i < 10; ++i) host[w][i] = 0;
    return hipMemcpyToSymbol(HIP_SYMBOL(ctl_tm), host, sizeof(host)) == hipSuccess ? 0 : -1;
}
#else
#define TM_DECL
#define TM(i)
#define TM_COUNT(i)
#define TM_FLUSH
#endif
#include "ctl_conv_igemm.h"


#ifndef CTL_WGRAD_PIPE_ASM_MFMA
#define CTL_WGRAD_PIPE_ASM_MFMA 0      // 1: the MFMA pairs as volatile asm (in place, no ties): measured 1.5-3 % slower, see the loop
#endif
#ifndef CTL_WGRAD_PIPE_ABLATE
#define CTL_WGRAD_PIPE_ABLATE 0
#endif
#ifndef CTL_WGRAD_PIPE_LAG
#define CTL_WGRAD_PIPE_LAG 2
#endif
// compile-time loop: f(std::integral_constant<int, K>) for K in [K0, N)
template <int K, int N, class F>
__device__ __forceinline__ void ctl_unroll(F&& f) {
    if constexpr (K < N) {
        f(std::integral_constant<int, K>{});
        ctl_unroll<K + 1, N>(f);
    }
}

// ------------------------------------------------------------------------------------------------ weight gradient
// dW[tap][ci][co] = sum_pixels x_virtual[pixel*S + tap - pad][ci] * dy[pixel][co]:  D[ci][co] += A[ci][k=pixel] B[pixel][co]
//   A: lane l -> x[pixel 4s + (l>>4) shifted by tap][ci = l&15]   (ds_read_b32, 256 B contiguous per wave)
//   B: lane l -> dy[pixel 4s + (l>>4)][co = l&15]
//   D: lane l holds dW[ci = 4*(l>>4)+r][co = l&15]
// Each block owns one 16-channel cin chunk (blockIdx.y) and NTW cout tiles (blockIdx.z), walks tiles
// blockIdx.x, +gridDim.x, ... and keeps all KS*KS*NTW accumulator tiles in registers; its four waves split the
// tile's pixels and are summed through LDS at the end.  Partial results per split are reduced by wgrad_reduce_kernel
// (deterministic: no float atomics).
// DY2: the output gradient is the virtual BatchNorm-backward result  A * dy + B * dy2 + C  (coefficients [group][3][cout] as the finalize
// writes them; dy = g, dy2 = the BatchNorm input): the `apply` pass runs in this staging (see XStage X2)
// PIPE (3x3 stride-1 layers with cin, cout multiples of 4): two LDS images.  While the MFMAs of tile t read one, the units of tile t+1
// go from their staging registers into the other and the loads of tile t+2 refill those registers, one unit per k-slot, all of it
// branch-free inside the MFMA stream: one barrier per tile, no exposed load-issue / staging phase (they were 13 % + 9 % of a tile's
// time in the single-image loop with one block per CU, profiles/r3_wgrad_phase_timers.txt).
template <int KS, int S, int MODE, int MT, int TW, int NTW, bool DY2 = false, bool PIPE = false>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(const ctl_conv d, const float* __restrict__ x,
                                                          const float* __restrict__ pro_scale,
                                                          const float* __restrict__ pro_shift,
                                                          const float* __restrict__ dy, float* __restrict__ w_partial,
                                                          float* __restrict__ b_partial, int tiles_h, int tiles_w,
                                                          int ntiles, int cin_p, int cout_p, const float* __restrict__ dy2,
                                                          const float* __restrict__ dy_coef) {
    using G = Geom<KS, S, MT, TW>;
    constexpr int TAPS = KS * KS;
    constexpr int DYT_FLOATS = NTW * G::TP * 16;
    constexpr int RED_FLOATS = 4 * NTW * 256;
    constexpr int BUF_FLOATS = G::XT_FLOATS + DYT_FLOATS;          // one LDS image: the input tile + the output-gradient tile
    constexpr int IMG_FLOATS = (PIPE ? 2 : 1) * BUF_FLOATS;
    constexpr int LDS_FLOATS = (IMG_FLOATS > RED_FLOATS) ? IMG_FLOATS : RED_FLOATS;
    static_assert(!PIPE || (KS == 3 && S == 1 && MODE != CTL_IN_C4), "pipelined weight gradient: 3x3 stride-1 layers");
    __shared__ __attribute__((aligned(16))) float lds[LDS_FLOATS + (DY2 ? 5 : 2) * CTL_PRO_MAX];
    float* xt = lds;
    float* dyt = lds + G::XT_FLOATS;
    float* cf_scale = lds + LDS_FLOATS;  // prologue coefficients [groups][cin], see XStage::store
    float* cf_shift = cf_scale + CTL_PRO_MAX;
    float* cd = cf_shift + CTL_PRO_MAX;  // DY2: A | B | C, [group][cout] each

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int p = lane & 15, q = lane >> 4;
    const int g = blockIdx.y;
    const int cot0 = blockIdx.z * NTW;
    const int group_n = d.n / (d.groups > 1 ? d.groups : 1);        // images per BatchNorm group (prologue coefficients)

    // C4 (input with <= 4 channels): the 16 rows of an MFMA result hold (tap 4j + i/4, channel i%4) instead of 16 channels of one
    // tap: 3 MFMA groups cover the 3x3 taps instead of 9; the partial layout [tap][ci][co] and the reduction are unchanged.
    constexpr bool C4 = (MODE == CTL_IN_C4);
    static_assert(!C4 || (KS == 3 && S == 1), "row-packed taps: 3x3 stride-1 only");
    constexpr int NACC = C4 ? 3 : TAPS;
    f32x4 acc[NACC][NTW];
#pragma unroll
    for (int a = 0; a < NACC; ++a)
#pragma unroll
        for (int t = 0; t < NTW; ++t) acc[a][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    int c4off[3];                   // C4: LDS float offset of this lane's (tap, channel) relative to the k-slot pixel, per MFMA group
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int tp = (4 * j + (p >> 2) < 9) ? 4 * j + (p >> 2) : 8;      // rows past tap 8 are computed on tap 8 and dropped
        c4off[j] = ((tp / 3) * G::IWP + tp % 3) * 4 + (p & 3);      // (dense image: 4 floats per pixel, see XStage)
    }
    float bsum[NTW];
#pragma unroll
    for (int t = 0; t < NTW; ++t) bsum[t] = 0.f;

    constexpr int DU = G::TP * NTW * 4, ND = DU / 256;       // TP is a multiple of 64 -> exact
    const __amdgpu_buffer_rsrc_t rx = ctl_rsrc(x, (int64_t)d.n * d.hin * d.win * d.cin * 4);
    const __amdgpu_buffer_rsrc_t rdy = ctl_rsrc(dy, (int64_t)d.n * d.hout * d.wout * d.cout * 4);
    const __amdgpu_buffer_rsrc_t rdy2 = DY2 ? ctl_rsrc(dy2, (int64_t)d.n * d.hout * d.wout * d.cout * 4) : rdy;
    XStage<KS, S, MODE, MT, TW> xs;
    xs.init(d);
    // dy tile: per-thread constants (pixel row/col inside the tile, byte offset relative to the tile origin, LDS offset)
    f32x4 dv[ND];
    f32x4 dv2[DY2 ? ND : 1];
    unsigned dmask = 0;             // DY2: units of the tile in flight that lie inside the image
    bool dwhole = false;            // DY2: ... all of them
    int dco = 0;                    // DY2: first channel of this thread's units (the same for all of them: 256 % (4 NTW) == 0)
    int drel[ND], drc[ND], dlds[ND];
#pragma unroll
    for (int i = 0; i < ND; ++i) {
        const int u = tid + i * 256;
        const int pix = u / (NTW * 4);
        const int q4 = u - pix * (NTW * 4);
        const int t = q4 >> 2, cq = q4 & 3;
        const int pr = pix / TW, pc = pix % TW;
        const int co = (cot0 + t) * 16 + cq * 4;
        drc[i] = (co < d.cout) ? (pr | (pc << 16)) : 0x7fff7fff;
        drel[i] = (co < d.cout) ? ((pr * d.wout + pc) * d.cout + co) * 4 : CTL_OOB;
        dlds[i] = (t * G::TP + pix) * 16 + cq * 4;
        if (i == 0) dco = co < d.cout ? co : 0;
    }
    auto dyload = [&](int n, int ho0, int wo0) {
        const int tb = ((n * d.hout + ho0) * d.wout + wo0) * d.cout * 4;
        dwhole = ho0 + G::TH <= d.hout && wo0 + TW <= d.wout && d.cout >= 4;
        if (dwhole) {      // whole tile: scalar tile offset, no per-unit VALU
#pragma unroll
            for (int i = 0; i < ND; ++i) dv[i] = ctl_bload4s(rdy, drel[i], tb);
            if constexpr (DY2) {
#pragma unroll
                for (int i = 0; i < ND; ++i) dv2[i] = ctl_bload4s(rdy2, drel[i], tb);
            }
            return;
        }
        dmask = 0;
#pragma unroll
        for (int i = 0; i < ND; ++i) {
            const bool ok = (unsigned)(ho0 + (drc[i] & 0xffff)) < (unsigned)d.hout && (unsigned)(wo0 + (drc[i] >> 16)) < (unsigned)d.wout;
            const int vo = ok ? (tb + drel[i]) : CTL_OOB;
            if constexpr (DY2) {
                dv[i] = ctl_bload4(rdy, vo); dv2[i] = ctl_bload4(rdy2, vo);
                dmask |= (ok && drel[i] != CTL_OOB) ? (1u << i) : 0u;
                continue;
            }
            if (d.cout >= 4) dv[i] = ctl_bload4(rdy, vo);
            else dv[i] = f32x4{ctl_bload1(rdy, vo), 0.f, 0.f, 0.f};
        }
    };
    auto dystore = [&](int goff) {
        if constexpr (DY2) {
            const float* cc = cd + goff + dco;
            const f32x4 ca = *reinterpret_cast<const f32x4*>(cc), cb = *reinterpret_cast<const f32x4*>(cc + CTL_PRO_MAX);
            const f32x4 c3 = *reinterpret_cast<const f32x4*>(cc + 2 * CTL_PRO_MAX);
            const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int i = 0; i < ND; ++i) {
                const f32x4 r = ca * dv[i] + cb * dv2[i] + c3;
                // pixels past the image contribute nothing (C alone would)
                *reinterpret_cast<f32x4*>(dyt + dlds[i]) = (dwhole || ((dmask >> i) & 1u)) ? r : zero;
            }
            return;
        }
#pragma unroll
        for (int i = 0; i < ND; ++i) *reinterpret_cast<f32x4*>(dyt + dlds[i]) = dv[i];
    };
    TileWalk cur;
    cur.init(blockIdx.x, gridDim.x, tiles_h, tiles_w);
    TM_DECL
    if constexpr (PIPE) {
        using XS = XStage<KS, S, MODE, MT, TW>;
        constexpr int NU = XS::NU, NITEM = NU + ND, NSLOT = MT * 4;
        constexpr bool PLAIN = CTL_MODE_IS_PLAIN(MODE);
        static_assert(TW == 16, "pipelined weight gradient: 16-pixel-wide tiles");
        // The loads of the pipeline go through resources built from laundered pointers: from `const __restrict__` arguments they are
        // loads of constant memory, which the optimizer may move anywhere -- it sank every load of a tile body across the barrier,
        // right in front of their use.
        const float *xl = x, *dyl = dy, *dy2l = DY2 ? dy2 : dy;
        asm volatile("" : "+s"(xl), "+s"(dyl), "+s"(dy2l));
        const __amdgpu_buffer_rsrc_t rx = ctl_rsrc(xl, (int64_t)d.n * d.hin * d.win * d.cin * 4);
        const __amdgpu_buffer_rsrc_t rdy = ctl_rsrc(dyl, (int64_t)d.n * d.hout * d.wout * d.cout * 4);
        const __amdgpu_buffer_rsrc_t rdy2 = ctl_rsrc(dy2l, (int64_t)d.n * d.hout * d.wout * d.cout * 4);
        const int ngrp = d.groups > 1 ? d.groups : 1;
        if (d.pro_affine)
            for (int i = tid; i < ngrp * d.cin; i += 256) { cf_scale[i] = pro_scale[i]; cf_shift[i] = pro_shift[i]; }
        if constexpr (DY2) {
            for (int i = tid; i < ngrp * d.cout; i += 256) {
                const int gi = i / d.cout, ch = i - gi * d.cout;
                cd[i] = dy_coef[(gi * 3 + 0) * d.cout + ch]; cd[CTL_PRO_MAX + i] = dy_coef[(gi * 3 + 1) * d.cout + ch];
                cd[2 * CTL_PRO_MAX + i] = dy_coef[(gi * 3 + 2) * d.cout + ch];
            }
        }
        const bool pro = d.pro_affine != 0;
        const float slope = d.pro_slope;
        const unsigned hv = PLAIN ? d.hin : 2 * d.hin, wv = PLAIN ? d.win : 2 * d.win;
        const bool chan_ok = g * 16 + (tid & 3) * 4 < d.cin;
        const bool chan_all = g * 16 + 16 <= d.cin;
        const int xcb = chan_ok ? g * 16 + (tid & 3) * 4 : 0;
        // L: the tile whose units are being LOADED into the staging registers; R: the tile held in them (wave-uniform scalars).
        // `xall` / `dall`: every unit of the tile lies inside the tensor -> the tile origin rides in the scalar offset of the loads and
        // the staging needs no masks (the fp32 MFMA shares the VALU issue port: VALU work is never hidden behind a wave's own MFMAs)
        int L_tb = 0, L_vh0 = 0, L_vw0 = 0, L_dtb = 0, L_ho0 = 0, L_wo0 = 0, L_n = 0, R_n = 0;
        bool L_live = false, L_xall = false, L_dall = false, R_xall = false, R_dall = false;
        int L_oh = 0, L_ow = 0;
        auto L_set_a = [&](bool live) {
            const int ho0 = cur.th * G::TH, wo0 = cur.tw * TW;
            L_live = live; L_n = live ? cur.n : 0; L_ho0 = ho0; L_wo0 = wo0;
            L_vh0 = ho0 - G::PAD; L_vw0 = wo0 - G::PAD;
            L_oh = PLAIN ? L_vh0 : ((ho0 >> 1) - XS::PADH); L_ow = PLAIN ? L_vw0 : ((wo0 >> 1) - XS::PADH);
        };
        auto L_set_b = [&]() {
            L_tb = (((cur.n * d.hin + L_oh) * d.win + L_ow) * d.cin + g * 16) * 4;
            L_dtb = ((cur.n * d.hout + L_ho0) * d.wout + L_wo0) * d.cout * 4;
        };
        auto L_set_c = [&]() {
            L_xall = L_live && chan_all && L_vh0 >= 0 && L_vw0 >= 0 && L_vh0 + G::IH <= (int)hv && L_vw0 + G::IW <= (int)wv;
            L_dall = L_live && L_ho0 + G::TH <= d.hout && L_wo0 + TW <= d.wout;
        };
        auto L_set = [&](bool live) { L_set_a(live); L_set_b(); L_set_c(); };
        // (every memory instruction sits outside the uniform branches: with loads on both sides of a branch the compiler loses count of
        // the loads in flight and waits for ALL of them -- vmcnt(0) in the first k-slot, i.e. for loads issued one slot earlier)
        // Each unit's work comes in four pieces, so that every piece fits behind one pair of MFMAs: value, LDS write, address, load.
        auto x_addr = [&](auto I) -> int {
            constexpr int i = decltype(I)::value;
            int vo = L_tb + xs.rel[i];             // (a unit past the tile: CTL_OOB + tile offset is still out of range)
            if (__builtin_expect(!L_xall, 0)) {
                const int vh = L_vh0 + (xs.rc[i] & 0xffff), vw = L_vw0 + (xs.rc[i] >> 16);
                const bool ok = (L_live & chan_ok) & ((unsigned)vh < hv) & ((unsigned)vw < wv);
                vo = ok ? vo : CTL_OOB;
                xs.vmask = (xs.vmask & ~(1u << i)) | (ok ? (1u << i) : 0u);
            }
            return vo;
        };
        auto x_value = [&](auto I, f32x4 sc, f32x4 sh) -> f32x4 {
            constexpr int i = decltype(I)::value;
            f32x4 t = xs.v[i];                     // no prologue: out-of-range units were loaded as hardware zeros
            if (pro) {
                t = ctl_leaky01(t * sc + sh, slope);
                if (__builtin_expect(!R_xall, 0)) {
                    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
                    t = ((xs.vmask >> i) & 1u) ? t : zero;      // padding stays zero
                }
            }
            return t;
        };
        auto dy_addr = [&](auto I) -> int {
            constexpr int i = decltype(I)::value;
            int vo = L_dtb + drel[i];
            if (__builtin_expect(!L_dall, 0)) {
                const bool ok = L_live & ((unsigned)(L_ho0 + (drc[i] & 0xffff)) < (unsigned)d.hout) & ((unsigned)(L_wo0 + (drc[i] >> 16)) < (unsigned)d.wout);
                vo = ok ? vo : CTL_OOB;
                if constexpr (DY2) dmask = (dmask & ~(1u << i)) | ((ok & (drel[i] != CTL_OOB)) ? (1u << i) : 0u);
            }
            return vo;
        };
        auto dy_value = [&](auto I, f32x4 ca, f32x4 cb, f32x4 c3) -> f32x4 {
            constexpr int i = decltype(I)::value;
            f32x4 r = dv[i];
            if constexpr (DY2) {
                r = ca * dv[i] + cb * dv2[i] + c3;
                if (__builtin_expect(!(R_dall && d.cout % 16 == 0), 0)) {
                    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
                    r = ((dmask >> i) & 1u) ? r : zero;     // pixels past the image contribute nothing (C alone would)
                }
            }
            return r;
        };
        // unit k of the tile (k < NU: input unit k, else output-gradient unit k - NU)
        auto u_value = [&](auto K, f32x4 sc, f32x4 sh, f32x4 ca, f32x4 cb, f32x4 c3) -> f32x4 {
            constexpr int k = decltype(K)::value;
            if constexpr (k < NU) return x_value(std::integral_constant<int, k>{}, sc, sh);
            else return dy_value(std::integral_constant<int, k - NU>{}, ca, cb, c3);
        };
        auto u_write = [&](auto K, float* img_w, f32x4 v) {
            constexpr int k = decltype(K)::value;
            if constexpr (k < NU) *reinterpret_cast<f32x4*>(img_w + xs.lds[k]) = v;
            else *reinterpret_cast<f32x4*>(img_w + G::XT_FLOATS + dlds[k - NU]) = v;
        };
        auto u_addr = [&](auto K) -> int {
            constexpr int k = decltype(K)::value;
            if constexpr (k < NU) return x_addr(std::integral_constant<int, k>{});
            else return dy_addr(std::integral_constant<int, k - NU>{});
        };
        auto u_load = [&](auto K, int vo) {
            constexpr int k = decltype(K)::value;
            if constexpr (k < NU) xs.v[k] = ctl_bload4(rx, vo);
            else {
                dv[k - NU] = ctl_bload4(rdy, vo);
                if constexpr (DY2) dv2[k - NU] = ctl_bload4(rdy2, vo);
            }
        };
        f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f}, ca = sh, cb = sh, c3 = sh;
        // coefficients of the tile in the staging registers (BatchNorm group of image n)
        auto coefs = [&](int n) {
            const int gi = ngrp > 1 ? n / group_n : 0;
            if (pro) {
                sc = *reinterpret_cast<const f32x4*>(cf_scale + gi * d.cin + xcb); sh = *reinterpret_cast<const f32x4*>(cf_shift + gi * d.cin + xcb);
            }
            if constexpr (DY2) {
                const float* cc = cd + gi * d.cout + dco;
                ca = *reinterpret_cast<const f32x4*>(cc); cb = *reinterpret_cast<const f32x4*>(cc + CTL_PRO_MAX);
                c3 = *reinterpret_cast<const f32x4*>(cc + 2 * CTL_PRO_MAX);
            }
        };
        auto all_units = [&](auto&& f) { ctl_unroll<0, NITEM>(f); };
        // tile 0 -> image 0, tile 1 -> the staging registers, L -> tile 2
        L_set((int)blockIdx.x < ntiles);
        all_units([&](auto K) { u_load(K, u_addr(K)); });
        __syncthreads();                                   // the coefficient tables
        R_xall = L_xall; R_dall = L_dall;
        coefs(L_n);
        all_units([&](auto K) { u_write(K, lds, u_value(K, sc, sh, ca, cb, c3)); });
        cur.next();
        L_set((int)blockIdx.x + (int)gridDim.x < ntiles);
        all_units([&](auto K) { u_load(K, u_addr(K)); });
        R_xall = L_xall; R_dall = L_dall;
        coefs(L_n);
        cur.next();
        L_set((int)blockIdx.x + 2 * (int)gridDim.x < ntiles);
        ctl_barrier_lds_writes_done();
        // one lane address per M-tile and operand (image 0); every tap / k-slot / cout tile is an immediate offset from it
        const float *xrb0[MT], *drb0[MT];
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            xrb0[m] = lds + (((wave * MT + m) * G::IWP + q) * 16 + p);
            drb0[m] = lds + G::XT_FLOATS + (((wave * MT + m) * TW + q) * 16 + p);
        }
        // EARLY: the last k-slot stages no unit, and its operands are read one slot ahead -- the tile's last LDS access is the unit write
        // of the slot before it.  The barrier goes there, and the last slot's spare pieces read the NEXT tile's first operands from the
        // other image: no LDS latency and no barrier skew between the last MFMA of a tile and the first of the next.
        constexpr bool EARLY = NSLOT >= 2 && NSLOT % 2 == 0 && (NITEM - 1) * NSLOT / NITEM < NSLOT - 1;
        float af[2][TAPS], bf[2][NTW];
        auto opread_from = [&](const float* const* xb, const float* const* db, auto J, auto T0, auto T1, bool with_b) {
            constexpr int slot = decltype(J)::value, m = slot / 4, s = slot % 4;
            if (with_b) {
#pragma unroll
                for (int t = 0; t < NTW; ++t) bf[slot & 1][t] = db[m][(t * G::TP + 4 * s) * 16];
            }
#pragma unroll
            for (int tap = decltype(T0)::value; tap < decltype(T1)::value; ++tap)
                af[slot & 1][tap] = xb[m][((tap / KS) * G::IWP + 4 * s + tap % KS) * 16];
        };
        if constexpr (EARLY) opread_from(xrb0, drb0, std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, std::integral_constant<int, TAPS>{}, true);
        int img = 0;
        for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
            TM_COUNT(6)
            const float *xrb[MT], *drb[MT], *xrn[MT], *drn[MT];          // this tile's image, the next tile's
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                xrb[m] = xrb0[m] + img * BUF_FLOATS; drb[m] = drb0[m] + img * BUF_FLOATS;
                xrn[m] = xrb0[m] + (img ^ 1) * BUF_FLOATS; drn[m] = drb0[m] + (img ^ 1) * BUF_FLOATS;
            }
            float* img_w = lds + (img ^ 1) * BUF_FLOATS;
            // The operands of a k-slot are read one slot ahead (two register sets).  A slot is nine MFMA pairs (one per tap); behind
            // each pair goes one small piece of the other work -- the next slot's operand reads (taps 0-2), this slot's unit of the
            // staging (value, LDS write, address, load: taps 3-6) and, in the last slot, the scalars of the next tile (taps 7-8) --
            // so that it issues while the pair executes.  MFMAs have no side effects: the two empty asm statements tie each pair to
            // its place (instruction selection would sink them below every fence), the fences keep the machine scheduler from
            // regrouping what the asm statements ordered.
            f32x4 stg = {0.f, 0.f, 0.f, 0.f};
            int ldvo = CTL_OOB;
            auto opread = [&](auto J, auto T0, auto T1, bool with_b) { opread_from(xrb, drb, J, T0, T1, with_b); };
            if constexpr (!EARLY) opread(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, std::integral_constant<int, TAPS>{}, true);
            TM(0)
            ctl_unroll<0, NSLOT>([&](auto J) {
                constexpr int slot = decltype(J)::value;
                constexpr int NXT = (slot + 1 < NSLOT) ? slot + 1 : 0;
                // the unit staged / loaded in this slot (at most one: NITEM <= NSLOT)
                constexpr int UK = [] { for (int k = 0; k < NITEM; ++k) if (k * NSLOT / NITEM == slot) return k; return -1; }();
                ctl_unroll<0, TAPS>([&](auto T) {
                    constexpr int tap = decltype(T)::value;
#if CTL_WGRAD_PIPE_ASM_MFMA
                    // The pair as ONE volatile asm statement: in place (the builtin form lets the register allocator rotate accumulator
                    // tiles, ~80 v_accvgpr moves per tile at the loop's back edge) and ordered like every other side effect, so no
                    // ties are needed; 148 + 72 registers instead of 164 + 104.  Safe without the compiler's hazard tracking: an
                    // accumulator tile is touched again 17 MFMAs later at the earliest, and the code behind the loop reads the tiles
                    // behind a barrier.  Measured (profiles/r3_wgrad_pipe.txt): 54.3 vs 53.5 us (64->64 at 64^2), 99.6 vs 96.7 with the
                    // second tensor -- the moves were not what the loop loses its time to; off.
                    if constexpr (NTW == 2)
                        asm volatile("v_mfma_f32_16x16x4_f32 %0, %2, %3, %0\n\tv_mfma_f32_16x16x4_f32 %1, %2, %4, %1"
                                     : "+a"(acc[tap][0]), "+a"(acc[tap][1]) : "v"(af[slot & 1][tap]), "v"(bf[slot & 1][0]), "v"(bf[slot & 1][NTW - 1]) : "memory");
                    else
                        asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(acc[tap][0]) : "v"(af[slot & 1][tap]), "v"(bf[slot & 1][0]) : "memory");
#else
                    if constexpr (CTL_WGRAD_PIPE_ABLATE < 3) asm volatile("" : "+v"(af[slot & 1][tap]));      // (3 = no ties)
#pragma unroll
                    for (int t = 0; t < NTW; ++t)
                        acc[tap][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[slot & 1][tap], bf[slot & 1][t], acc[tap][t], 0, 0, 0);
                    // (the output tie names the pair issued CTL_WGRAD_PIPE_LAG pairs earlier, which has completed: a tie on the pair
                    // just issued makes the compiler wait for its result before the piece below, out of the pair's shadow)
                    constexpr int tp = (tap + TAPS - CTL_WGRAD_PIPE_LAG) % TAPS;
                    if constexpr (CTL_WGRAD_PIPE_ABLATE >= 3) {}
                    else if constexpr (NTW == 2) asm volatile("" : "+a"(acc[tp][0]), "+a"(acc[tp][NTW - 1]) :: "memory");
                    else asm volatile("" : "+a"(acc[tp][0]) :: "memory");
#endif
                    if constexpr (tap < 3 && CTL_WGRAD_PIPE_ABLATE < 2) {          // (2 = no operand reads either)
                        if constexpr (slot + 1 < NSLOT)
                            opread(std::integral_constant<int, NXT>{}, std::integral_constant<int, 3 * tap>{}, std::integral_constant<int, 3 * tap + 3>{}, tap == 0);
                        else if constexpr (EARLY)         // (behind the barrier of the slot before: the next tile's first operands)
                            opread_from(xrn, drn, std::integral_constant<int, 0>{}, std::integral_constant<int, 3 * tap>{}, std::integral_constant<int, 3 * tap + 3>{}, tap == 0);
                    } else if constexpr (UK >= 0 && CTL_WGRAD_PIPE_ABLATE < 1) {      // (timing ablations: 1 = no staging, wrong results)
                        constexpr int UKK = UK >= 0 ? UK : 0;
                        constexpr bool ST = CTL_WGRAD_PIPE_ABLATE != -1, LD = CTL_WGRAD_PIPE_ABLATE != -2;      // (-1: no LDS writes, -2: no loads)
                        if constexpr (tap == 3 && ST) stg = u_value(std::integral_constant<int, UKK>{}, sc, sh, ca, cb, c3);
                        else if constexpr (tap == 4 && ST) u_write(std::integral_constant<int, UKK>{}, img_w, stg);
                        else if constexpr (tap == 5 && LD) ldvo = u_addr(std::integral_constant<int, UKK>{});
                        else if constexpr (tap == 6 && LD) u_load(std::integral_constant<int, UKK>{}, ldvo);
                    }
                    // the last slot, behind its unit if it has one: every unit of the tile in the registers is staged -> R <- L, L <- next
                    if constexpr (slot == NSLOT - 1) {
                        constexpr int P0 = UK >= 0 ? 7 : 3;          // first free piece
                        if constexpr (tap == P0) { R_xall = L_xall; R_dall = L_dall; coefs(L_n); cur.next(); }
                        if constexpr (UK >= 0) {
                            if constexpr (tap == 8) L_set(tile + 3 * (int)gridDim.x < ntiles);
                        } else {
                            if constexpr (tap == 5) L_set_a(tile + 3 * (int)gridDim.x < ntiles);
                            if constexpr (tap == 6) L_set_b();
                            if constexpr (tap == 7) L_set_c();
                        }
                    }
                    if constexpr (tap == 8) {
#pragma unroll
                        for (int t = 0; t < NTW; ++t) bsum[t] += bf[slot & 1][t];
                    }
                    if constexpr (EARLY && slot == NSLOT - 2 && tap == 4) {
                        TM(1)
                        ctl_barrier_lds_writes_done();       // everybody is done reading this image and has written the other
                        TM(2)
                    }
                    __builtin_amdgcn_sched_barrier(0);
                });
            });
            if constexpr (!EARLY) {
                TM(1)
                ctl_barrier_lds_writes_done();       // everybody is done reading this image and has written the other
                TM(2)
            }
            img ^= 1;
        }
#if CTL_WGRAD_PIPE_ASM_MFMA
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");      // the last MFMAs complete before anything reads their tiles
#endif
    } else {
    if ((int)blockIdx.x < ntiles) {
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
    if ((int)blockIdx.x < ntiles) {
        xs.store(xt, d, g, cf_scale, cf_shift, (cur.n / group_n) * d.cin);
        dystore((cur.n / group_n) * d.cout);
    }
    __syncthreads();
    TM(7)
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        TM_COUNT(6)
        const bool has_next = tile + (int)gridDim.x < ntiles;
        if (has_next) {                   // next tile's loads fly while this tile's MFMAs run
            cur.next();
            xs.load(rx, d, cur.n, cur.th * G::TH, cur.tw * TW, g);
            dyload(cur.n, cur.th * G::TH, cur.tw * TW);
        }
        TM(0)
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const int mt = wave * MT + m;
            const int tr = mt / (TW / 16), tc = (mt % (TW / 16)) * 16;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const int pc = tc + 4 * s + q;  // this lane's k-slot pixel column inside the tile
                float bf[NTW];
#pragma unroll
                for (int t = 0; t < NTW; ++t) {
                    bf[t] = dyt[(t * G::TP + tr * TW + pc) * 16 + p];
                    bsum[t] += bf[t];
                }
#pragma unroll
                for (int tap = 0; tap < NACC; ++tap) {
                    const int kh = tap / KS, kw = tap % KS;
                    const float af = C4 ? xt[(tr * G::IWP + pc) * 4 + c4off[tap]]
                                        : xt[((tr * S + kh) * G::IWP + G::ldscol(pc * S + kw)) * 16 + p];
#pragma unroll
                    for (int t = 0; t < NTW; ++t)
                        acc[tap][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(af, bf[t], acc[tap][t], 0, 0, 0);
                }
            }
        }
        TM(1)
        ctl_barrier_lds_reads_done();
        TM(2)
        if (has_next) {
            xs.store(xt, d, g, cf_scale, cf_shift, (cur.n / group_n) * d.cin);
            dystore((cur.n / group_n) * d.cout);
        }
        TM(3)
        ctl_barrier_lds_writes_done();
        TM(4)
    }
    }
    __syncthreads();
#ifdef CTL_TIMING_WGRAD
    const unsigned long long tm_loop_end = __builtin_amdgcn_s_memtime();
#endif

    // ---------------- sum the four waves through LDS, several taps per round, and write this split's partial.  Raw LDS
    // barriers: a __syncthreads() here would drain the partial stores of the previous round (vmcnt(0)) 18 times per block.
    float* red = lds;
    constexpr int TAP_FLOATS = 4 * NTW * 256;
    constexpr int TPR = (LDS_FLOATS / TAP_FLOATS) < NACC ? (LDS_FLOATS / TAP_FLOATS) : NACC;     // taps per round (>= 1)
    const int64_t split_base = (int64_t)blockIdx.x * TAPS * cin_p * cout_p;
#pragma unroll
    for (int tap0 = 0; tap0 < NACC; tap0 += TPR) {
        if (tap0 > 0) ctl_barrier_lds_reads_done();     // the sums of the previous round were consumed by their stores
#pragma unroll
        for (int tp = 0; tp < TPR; ++tp) {
            const int tap = tap0 + tp;
            if (tap < NACC) {
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
            if (tap < NACC) {
#pragma unroll
                for (int e0 = 0; e0 < NTW * 256; e0 += 256) {
                    const int e = e0 + tid;
                    const int t = e >> 8, r = (e >> 6) & 3, l = e & 63;
                    float v = 0.f;
#pragma unroll
                    for (int w = 0; w < 4; ++w) v += red[tp * TAP_FLOATS + ((w * NTW + t) * 4 + r) * 64 + l];
                    const int co = (cot0 + t) * 16 + (l & 15);
                    if (C4) {      // result row 4*(l>>4) + r = (tap 4*group + (l>>4), channel r)
                        const int wtap = 4 * tap + (l >> 4);
                        if (wtap < TAPS && co < cout_p) w_partial[split_base + ((int64_t)wtap * cin_p + r) * cout_p + co] = v;
                    } else {
                        const int ci = g * 16 + (l >> 4) * 4 + r;
                        if (co < cout_p) w_partial[split_base + ((int64_t)tap * cin_p + ci) * cout_p + co] = v;
                    }
                }
            }
        }
    }
    if (g == 0 && b_partial != nullptr) {  // bias gradient: column sums of dy
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
            if (co < cout_p) b_partial[(int64_t)blockIdx.x * cout_p + co] = v;
        }
    }
#ifdef CTL_TIMING_WGRAD
    tm_acc[5] += __builtin_amdgcn_s_memtime() - tm_loop_end;      // [5] = the cross-wave reduction + partial write
    TM_FLUSH
#endif
}

// Sums the per-split partials: a block owns 32 consecutive weight elements (contiguous in the partial layout) and
// spreads the splits over 8 thread groups; deterministic (fixed order), no atomics.
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ w_partial,
                                                            const float* __restrict__ b_partial, int splits, int taps,
                                                            int ks, int cin, int cout, int cin_p, int cout_p,
                                                            float* __restrict__ dw, int64_t s_co, int64_t s_ci,
                                                            int64_t s_kh, int64_t s_kw, float* __restrict__ dbias,
                                                            int accumulate) {
    __shared__ float sm[8][32];
    const int e = threadIdx.x & 31, sl = threadIdx.x >> 5;
    const int64_t total = (int64_t)taps * cin * cout;
    const int64_t idx = (int64_t)blockIdx.x * 32 + e;
    const bool is_w = idx < total;
    const bool is_b = !is_w && dbias != nullptr && idx < total + cout;
    float v = 0.f;
    int co = 0, ci = 0, tap = 0;
    if (is_w) {
        co = idx % cout;
        ci = (idx / cout) % cin;
        tap = idx / ((int64_t)cout * cin);
        const int64_t stride = (int64_t)taps * cin_p * cout_p;
        const float* src = w_partial + ((int64_t)tap * cin_p + ci) * cout_p + co;
        for (int s = sl; s < splits; s += 8) v += src[s * stride];
    } else if (is_b) {
        co = idx - total;
        for (int s = sl; s < splits; s += 8) v += b_partial[(int64_t)s * cout_p + co];
    }
    sm[sl][e] = v;
    __syncthreads();
    if (sl == 0 && (is_w || is_b)) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) t += sm[k][e];
        float* dst = is_w ? dw + co * s_co + ci * s_ci + (tap / ks) * s_kh + (tap % ks) * s_kw : dbias + co;
        *dst = accumulate ? (*dst + t) : t;
    }
}

__global__ void pack_weights_kernel(const float* __restrict__ src, float* __restrict__ dst, int cout, int cin, int ks,
                                    int g_chunks, int64_t total, int64_t s_co, int64_t s_ci, int64_t s_kh,
                                    int64_t s_kw, int flip) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int j = idx & 3, lane = (idx >> 2) & 63;
    int64_t rest = idx >> 8;
    const int g = rest % g_chunks;
    rest /= g_chunks;
    const int taps = ks * ks;
    const int tap = rest % taps;
    const int cot = rest / taps;
    const int co = cot * 16 + (lane & 15);
    const int ci = g * 16 + (lane >> 4) * 4 + j;
    int kh = tap / ks, kw = tap % ks;
    if (flip) { kh = ks - 1 - kh; kw = ks - 1 - kw; }
    float v = 0.f;
    if (co < cout && ci < cin) v = src[co * s_co + ci * s_ci + kh * s_kh + kw * s_kw];
    dst[idx] = v;
}

// ---- table-driven batched forms: blockIdx.y selects the record
__global__ void pack_weights_batched_kernel(const float* __restrict__ params, float* __restrict__ wpack,
                                            const int64_t* __restrict__ table) {
    const int64_t* r = table + (int64_t)blockIdx.y * 12;
    const int64_t total = r[10];
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total || (r[11] & CTL_PACK_X3)) return;      // (CTL_PACK_X3 records: ctl_pack_weights_x3_batched)
    const float* src = params + r[0];
    float* dst = wpack + r[1];
    const int cout = (int)r[2], cin = (int)r[3], ks = (int)r[4], flip = (int)r[5];
    const int g_chunks = (cin + 15) / 16;
    const int j = idx & 3, lane = (idx >> 2) & 63;
    int64_t rest = idx >> 8;
    const int g = rest % g_chunks;
    rest /= g_chunks;
    const int taps = ks * ks;
    const int tap = rest % taps;
    const int cot = rest / taps;
    const int co = cot * 16 + (lane & 15);
    const int ci = g * 16 + (lane >> 4) * 4 + j;
    int kh = tap / ks, kw = tap % ks;
    float v = 0.f;
    if (r[11] == 4) {
        // K-packed 3x3 weights for inputs with <= 4 channels: [cot][quad j][lane][c] = W[co][c][tap 4j + (lane>>4)] (0 past tap 8 / cin)
        const int c = j, quad = (int)((idx >> 8) % 3), cot4 = (int)((idx >> 8) / 3);
        const int co4 = cot4 * 16 + (lane & 15), tp = 4 * quad + (lane >> 4);
        if (co4 < cout && c < cin && tp < 9) v = src[co4 * r[6] + c * r[7] + (tp / 3) * r[8] + (tp % 3) * r[9]];
        dst[idx] = v;
        return;
    }
    if (r[11] == 1) {
        // 4x4 stride-2 pad-1 kernel K of  dx = sumpool2(conv3x3^T(dy))  (data gradient of a 3x3 conv on a nearest-upsampled input):
        // dx[i] = sum_a dx'[2i+a] = sum_{a,kh'} W[kh']^T dy[2i + a + 1 - kh']  ==>  K[u] = sum_{a in {0,1}, kh' = a+2-u in [0,2]} W[kh']
        if (co < cout && ci < cin) {
            for (int a = 0; a < 2; ++a) {
                const int sh = a + 2 - kh;
                if (sh < 0 || sh > 2) continue;
                for (int b = 0; b < 2; ++b) {
                    const int sw = b + 2 - kw;
                    if (sw < 0 || sw > 2) continue;
                    v += src[co * r[6] + ci * r[7] + sh * r[8] + sw * r[9]];
                }
            }
        }
    } else if (r[11] == 2) {
        // phase z = 2a+b (in r[5]) of a 3x3 conv on a nearest-upsampled input, as a 2x2 conv on the stored input:
        // y[2i+a] = sum_kh W[kh] x[(2i+a+kh-1)>>1]:  a=0: tap 0 <- W[0], tap 1 <- W[1]+W[2];  a=1: tap 0 <- W[0]+W[1], tap 1 <- W[2]
        if (co < cout && ci < cin) {
            const int a = flip >> 1, b = flip & 1;
            const int h0 = (kh == 0) ? 0 : (a ? 2 : 1), h1 = (kh == 0) ? (a ? 1 : 0) : 2;
            const int w0 = (kw == 0) ? 0 : (b ? 2 : 1), w1 = (kw == 0) ? (b ? 1 : 0) : 2;
            for (int sh = h0; sh <= h1; ++sh)
                for (int sw = w0; sw <= w1; ++sw) v += src[co * r[6] + ci * r[7] + sh * r[8] + sw * r[9]];
        }
    } else if (r[11] == 3) {
        // phase z = 2a+b of the data gradient of a stride-2 pad-1 3x3 conv: dx[2i+a] = sum over kh with (a+1-kh) even of
        // W[kh]^T dy[i + (a+1-kh)/2]:  a=0: tap 0 <- W[1];  a=1: tap 0 <- W[2], tap 1 <- W[0]   (the other taps are zero)
        if (co < cout && ci < cin) {
            const int a = flip >> 1, b = flip & 1;
            const int sh = a ? (kh == 0 ? 2 : 0) : (kh == 0 ? 1 : -1);
            const int sw = b ? (kw == 0 ? 2 : 0) : (kw == 0 ? 1 : -1);
            if (sh >= 0 && sw >= 0) v = src[co * r[6] + ci * r[7] + sh * r[8] + sw * r[9]];
        }
    } else {
        if (flip) { kh = ks - 1 - kh; kw = ks - 1 - kw; }
        if (co < cout && ci < cin) v = src[co * r[6] + ci * r[7] + kh * r[8] + kw * r[9]];
    }
    dst[idx] = v;
}

// Block shape adapts to the record: many splits of a small weight tensor -> 8 elements x 32 split lanes; few splits of a
// large one -> 64 elements x 4 split lanes (256-byte coalesced rows).  Splits are summed in a fixed order (deterministic).
__global__ __launch_bounds__(256) void wgrad_reduce_batched_kernel(const float* __restrict__ scratch, float* __restrict__ grad,
                                                                    const int64_t* __restrict__ table) {
    __shared__ float sm[32][65];
    const int64_t* r = table + (int64_t)blockIdx.y * 16;
    const int splits = (int)r[4], taps = (int)(r[5] & 0xff), ks = (int)(r[5] >> 8), cin = (int)r[6], cout = (int)r[7];
    const int cin_p = (int)r[8], cout_p = (int)r[9];
    const int64_t total = (int64_t)taps * cin * cout;
    const bool has_b = r[1] >= 0 && r[3] >= 0;
    const bool wide = splits <= 64;
    const int EW = wide ? 64 : 8, SLN = wide ? 4 : 32;
    if ((int64_t)blockIdx.x * EW >= total + (has_b ? cout : 0)) return;
    const int e = wide ? (threadIdx.x & 63) : (threadIdx.x & 7), sl = wide ? (threadIdx.x >> 6) : (threadIdx.x >> 3);
    const int64_t idx = (int64_t)blockIdx.x * EW + e;
    const bool is_w = idx < total, is_b = !is_w && has_b && idx < total + cout;
    float v = 0.f;
    int co = 0, ci = 0, tap = 0;
    if (is_w) {
        co = idx % cout;
        ci = (idx / cout) % cin;
        tap = idx / ((int64_t)cout * cin);
        const int64_t stride = (int64_t)taps * cin_p * cout_p;
        const float* src = scratch + r[0] + ((int64_t)tap * cin_p + ci) * cout_p + co;
        // 8 split rows in flight per thread (one per trip leaves this launch, the serial tail of every backward plan, latency-bound);
        // the adds keep the order of the rolled loop
        for (int s0 = sl; s0 < splits; s0 += 8 * SLN) {
            float t8[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) { const int s = s0 + u * SLN; t8[u] = s < splits ? src[s * stride] : 0.f; }
#pragma unroll
            for (int u = 0; u < 8; ++u) v += t8[u];
        }
    } else if (is_b) {
        co = idx - total;
        const float* src = scratch + r[1] + co;
        for (int s = sl; s < splits; s += SLN) v += src[(int64_t)s * cout_p];
    }
    sm[sl][e] = v;
    __syncthreads();
    if (sl == 0 && (is_w || is_b)) {
        float t = 0.f;
        for (int k = 0; k < SLN; ++k) t += sm[k][e];
        float* dst = is_w ? grad + r[2] + co * r[10] + ci * r[11] + (tap / ks) * r[12] + (tap % ks) * r[13] : grad + r[3] + co;
        *dst = r[14] ? (*dst + t) : t;
    }
}

// ------------------------------------------------------------------------------------------------ host side
extern "C" int ctl_pack_weights_batched(const float* params, float* wpack, const int64_t* table, int32_t n_rec,
                                        int64_t max_total, ctl_stream stream) {
    CTL_REQUIRE(params && wpack && table && n_rec > 0 && max_total > 0, "pack_weights_batched: bad arguments");
    pack_weights_batched_kernel<<<dim3((unsigned)ctl_cdiv64(max_total, 256), (unsigned)n_rec), dim3(256), 0, (hipStream_t)stream>>>(
        params, wpack, table);
    CTL_LAUNCH_CHECK("pack_weights_batched");
    return CTL_OK;
}
extern "C" int ctl_wgrad_reduce_batched(const float* scratch, float* grad, const int64_t* table, int32_t n_rec,
                                        int64_t max_elems, ctl_stream stream) {
    CTL_REQUIRE(scratch && grad && table && n_rec > 0 && max_elems > 0, "wgrad_reduce_batched: bad arguments");
    // `max_elems` is the largest per-record block count (elements / 64 for <= 64 splits, / 8 otherwise), see nets.py
    wgrad_reduce_batched_kernel<<<dim3((unsigned)max_elems, (unsigned)n_rec), dim3(256), 0, (hipStream_t)stream>>>(
        scratch, grad, table);
    CTL_LAUNCH_CHECK("wgrad_reduce_batched");
    return CTL_OK;
}

extern "C" size_t ctl_conv_wpack_floats(int32_t cin, int32_t cout, int32_t ks) {
    return (size_t)ctl_cdiv(cout, 16) * ks * ks * ctl_cdiv(cin, 16) * 256;
}

static bool conv_combo_ok(const ctl_conv* d) {
    const int k = d->ks, s = d->stride, m = d->in_mode, pd = d->pad;
    if (k == 3 && s == 1 && pd == 1 && m != CTL_IN_C4) return true;
    if (k == 3 && s == 1 && pd == 1 && m == CTL_IN_C4 && d->cin <= 4 && d->nsub == 1) return true;     // K-packed taps (first layers)
    if (k == 3 && s == 2 && pd == 1 && m == CTL_IN_PLAIN) return true;
    if (k == 1 && s == 1 && pd == 0 && m != CTL_IN_ZINS2) return true;
    if (k == 2 && s == 2 && pd == 0 && m == CTL_IN_PLAIN) return true;
    if (k == 4 && s == 2 && pd == 1 && m == CTL_IN_PLAIN) return true;     // pooled 3x3 data gradient (nearest-upsample blocks)
    if (k == 2 && s == 1 && (pd == 0 || pd == 2) && m == CTL_IN_PLAIN && d->nsub == 4) return true;   // four scattered phase problems
    return false;
}

int ctl_conv_pick_cfg(const ctl_conv* d, ctl_conv_cfg* c, int for_wgrad) {
    CTL_REQUIRE(conv_combo_ok(d), "conv: unsupported ks/stride/pad/in_mode %d/%d/%d/%d", d->ks, d->stride, d->pad,
                d->in_mode);
    CTL_REQUIRE(d->cin == 1 || d->cin % 4 == 0, "conv: cin must be 1 or a multiple of 4 (got %d)", d->cin);
    CTL_REQUIRE(d->cout == 1 || d->cout % 4 == 0, "conv: cout must be 1 or a multiple of 4 (got %d)", d->cout);
    CTL_REQUIRE(d->cin < 16 ? (d->cin == 1 || d->cin == 4 || d->cin == 8 || d->cin == 12) : d->cin % 16 == 0,
                "conv: cin >= 16 must be a multiple of 16 (got %d)", d->cin);
    CTL_REQUIRE(d->nsub == 1 || d->nsub == 4, "conv: nsub must be 1 or 4");
    c->cot = ctl_cdiv(d->cout, 16);
    c->g = ctl_cdiv(d->cin, 16);
    // Measured on MI355X (tools/bench_conv.py, bs16 layers): 32 output channels per block (NT=2, weights through LDS)
    // beats 64 everywhere (occupancy); 8x32 tiles win while they still give >= 512 blocks, then 8x16, then 4x16.
    c->nt = (c->cot % 2 == 0) ? 2 : 1;
    auto blocks = [&](int mt, int tw) {
        const int th = 4 * mt * 16 / tw;
        return (int64_t)d->n * ctl_cdiv(d->hout, th) * ctl_cdiv(d->wout, tw) * (c->cot / c->nt) * d->nsub;
    };
    int mt, tw;
    if (for_wgrad) {
        // the wgrad kernel keeps KS*KS*NTW accumulator tiles in registers: 8x16 tiles measured best on every layer
        if (d->hout >= 8) { mt = 2; tw = 16; }
        else { mt = 1; tw = 16; }
    } else if (d->stride == 2) {
        if (blocks(2, 16) >= 384) { mt = 2; tw = 16; } else { mt = 1; tw = 16; }
        if (d->ks == 4 && c->nt == 2) { mt = 1; tw = 16; }       // 16-tap weight image + 18x34 input tile would exceed 64 KiB of LDS
    } else if (d->wout >= 32 && blocks(4, 32) >= 512 && !(c->nt == 2 && blocks(4, 32) > 768)) {
        mt = 4; tw = 32;          // (with 32 output channels per block the 8x32 kernel sits at 2 blocks per CU: once there are more
                                  //  tiles than that the 8x16 kernel at 3 per CU is faster, e.g. 32->32 at 128^2: 53 vs 57 us)
    } else if (blocks(2, 16) >= 384) {
        mt = 2; tw = 16;
    } else {
        mt = 1; tw = 16;
    }
    {   // tuning hook (tools/bench_conv.py): CTL_FORCE_CFG="mt,tw,nt" overrides the heuristic when the combination is valid
        static int fm = 0, ft = 0, fn = 0;                 // read ONCE per process (this runs on every conv launch)
        static const bool forced = [] {
            const char* f = ctl_tune_str("CTL_FORCE_CFG");
            return f && sscanf(f, "%d,%d,%d", &fm, &ft, &fn) == 3;
        }();
        if (forced) {
            const bool tile_ok = (fm == 4 && ft == 32 && d->stride == 1 && !for_wgrad) || (fm == 2 && ft == 16) || (fm == 1 && ft == 16);
            if (tile_ok) { mt = fm; tw = ft; }
            if ((fn == 1 || fn == 2) && c->cot % fn == 0) c->nt = fn;
            if (d->ks == 4 && c->nt == 2) { mt = 1; tw = 16; }
        }
    }
    c->pc = 0;
    c->mt = mt; c->tw = tw; c->th = 4 * mt * 16 / tw;
    c->tiles_h = ctl_cdiv(d->hout, c->th);
    c->tiles_w = ctl_cdiv(d->wout, c->tw);
    return CTL_OK;
}

// Persistent grid: what is RESIDENT at once (256 CUs x the kernel's occupancy, at most CTL_PERSIST = 4 blocks per CU) -- a
// larger grid runs in rounds, pays the block setup again and ends in a thin tail; a smaller one leaves CUs with fewer blocks
// than others (the dispatcher spreads blocks evenly over the CUs, tools/micro/dispatch_probe.hip).  With gridDim.y/z > 1 the
// x extent is kept a multiple of 8 so that blockIdx.x % 8 stays the XCD of a block.
int ctl_conv_grid_x(int ntiles, int other, int occ) {
    static int per_cu = -1;
    if (per_cu < 0) {
        per_cu = ctl_tune_int("CTL_PERSIST", 4);
        if (per_cu < 1) per_cu = 1;
    }
    const int resident = occ < per_cu ? occ : per_cu;
    int cap = (ctl_num_cus() * resident) / other;
    if (other > 1 && cap >= 8) cap -= cap % 8;
    if (cap < 1) cap = 1;
    return ntiles < cap ? ntiles : cap;
}

extern "C" int ctl_pack_weights(const float* src, float* dst, int32_t cout, int32_t cin, int32_t ks, int64_t s_co,
                                int64_t s_ci, int64_t s_kh, int64_t s_kw, int32_t flip, ctl_stream stream) {
    CTL_REQUIRE(src && dst && cout > 0 && cin > 0 && ks >= 1 && ks <= 4, "pack_weights: bad arguments");
    const int64_t total = (int64_t)ctl_conv_wpack_floats(cin, cout, ks);
    pack_weights_kernel<<<dim3((unsigned)ctl_cdiv64(total, 256)), dim3(256), 0, (hipStream_t)stream>>>(
        src, dst, cout, cin, ks, ctl_cdiv(cin, 16), total, s_co, s_ci, s_kh, s_kw, flip);
    CTL_LAUNCH_CHECK("pack_weights");
    return CTL_OK;
}

struct conv_call {
    const ctl_conv* d; ctl_conv_cfg c;
    const float *x, *wpack, *bias, *pro_scale, *pro_shift, *res, *res_scale, *res_shift, *res2, *x2;
    float *y, *stats_partial, *pool, *xout;
    hipStream_t stream;
    bool query;      // only report the grid (ctl_conv_stats_blocks), launch nothing
    int grid_x;
};

template <int KS, int S, int MODE, int MT, int TW, int NT, int EPI, bool X2 = false>
static void conv_go(conv_call& a) {
    if constexpr (!X2 && ((KS == 3 && S == 1 && MODE == CTL_IN_PLAIN) || (KS == 4 && S == 2)) && EPI != 2) {
        if (a.d->pro_affine == 2) { conv_go<KS, S, MODE, MT, TW, NT, EPI, true>(a); return; }      // the BatchNorm-backward prologue (checked in ctl_conv_forward_ex)
    }
    static int occ = 0;          // resident blocks per CU of this instantiation (asked once)
    if (!occ) {
        int n = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, conv_igemm_kernel<KS, S, MODE, MT, TW, NT, EPI, X2>, 256, 0) != hipSuccess || n < 1) {
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
    conv_igemm_kernel<KS, S, MODE, MT, TW, NT, EPI, X2><<<grid, dim3(256), 0, a.stream>>>(
        *d, a.x, a.wpack, a.bias, a.pro_scale, a.pro_shift, a.res, a.res_scale, a.res_shift, a.y, a.stats_partial, a.c.tiles_h,
        a.c.tiles_w, a.c.g, (int64_t)ctl_conv_wpack_floats(d->cin, d->cout, d->ks), ntiles, a.res2, a.x2, a.pool, a.xout);
}
// the launches that write dL/dOut of a residual block (and can carry CTL_EPI_TAILBWD): the 1x1 data gradients, the 2x2 stride-2 conv
// behind a ConvTranspose2d, the four phase problems / the zero-insert form of a stride-2 3x3 data gradient
template <int KS, int S, int MODE>
constexpr bool conv_tail_ok() { return (KS == 1 && MODE == CTL_IN_PLAIN) || KS == 2 || (KS == 3 && S == 1 && MODE == CTL_IN_ZINS2); }
template <int KS, int S, int MODE, int MT, int TW>
static void conv_go_nt(conv_call& a) {
    const bool epi = (a.d->epi_flags & (CTL_EPI_RES | CTL_EPI_ACCUM | CTL_EPI_BNBWD)) != 0;
    if constexpr (conv_tail_ok<KS, S, MODE>()) {
        if (a.d->epi_flags & CTL_EPI_TAILBWD) {
            if (a.c.nt == 2) conv_go<KS, S, MODE, MT, TW, 2, 2>(a); else conv_go<KS, S, MODE, MT, TW, 1, 2>(a);
            return;
        }
    }
    if constexpr (KS == 3 && S == 1 && MODE == CTL_IN_PLAIN) {      // CTL_EPI_BNBWD alone on whole 16-channel tiles: its own instantiation
        if ((a.d->epi_flags & (CTL_EPI_RES | CTL_EPI_ACCUM | CTL_EPI_BNBWD)) == CTL_EPI_BNBWD && a.d->cout % 16 == 0 && a.d->epi_act == CTL_ACT_NONE) {
            if (a.c.nt == 2) conv_go<KS, S, MODE, MT, TW, 2, 3>(a); else conv_go<KS, S, MODE, MT, TW, 1, 3>(a);
            return;
        }
    }
    if (a.c.nt == 2) { if (epi) conv_go<KS, S, MODE, MT, TW, 2, 1>(a); else conv_go<KS, S, MODE, MT, TW, 2, 0>(a); }
    else { if (epi) conv_go<KS, S, MODE, MT, TW, 1, 1>(a); else conv_go<KS, S, MODE, MT, TW, 1, 0>(a); }
}
template <int KS, int S, int MODE>
static void conv_go_tile(conv_call& a) {
    if (S == 1 && a.c.mt == 4 && a.c.tw == 32) conv_go_nt<KS, S, MODE, (S == 1 ? 4 : 2), (S == 1 ? 32 : 16)>(a);
    else if (a.c.mt == 2) conv_go_nt<KS, S, MODE, 2, 16>(a);
    else conv_go_nt<KS, S, MODE, 1, 16>(a);
}
static int conv_dispatch(conv_call& a) {
    const int k = a.d->ks, s = a.d->stride, m = a.d->in_mode;
    if (k == 3 && s == 1 && m == CTL_IN_PLAIN) conv_go_tile<3, 1, CTL_IN_PLAIN>(a);
    else if (k == 3 && s == 1 && m == CTL_IN_UP2) conv_go_tile<3, 1, CTL_IN_UP2>(a);
    else if (k == 3 && s == 1 && m == CTL_IN_ZINS2) conv_go_tile<3, 1, CTL_IN_ZINS2>(a);
    else if (k == 3 && s == 1 && m == CTL_IN_C4) conv_go_tile<3, 1, CTL_IN_C4>(a);
    else if (k == 3 && s == 2) conv_go_tile<3, 2, CTL_IN_PLAIN>(a);
    else if (k == 1 && m == CTL_IN_PLAIN) conv_go_tile<1, 1, CTL_IN_PLAIN>(a);
    else if (k == 1 && m == CTL_IN_UP2) conv_go_tile<1, 1, CTL_IN_UP2>(a);
    else if (k == 2 && s == 2) conv_go_tile<2, 2, CTL_IN_PLAIN>(a);
    else if (k == 2 && s == 1) conv_go_tile<2, 1, CTL_IN_PLAIN>(a);
    else if (k == 4 && s == 2) {          // only the LDS-feasible shapes are instantiated
        const bool epi = (a.d->epi_flags & (CTL_EPI_RES | CTL_EPI_ACCUM | CTL_EPI_BNBWD)) != 0;
        if (a.c.mt == 2 && a.c.nt == 1) { if (epi) conv_go<4, 2, CTL_IN_PLAIN, 2, 16, 1, 1>(a); else conv_go<4, 2, CTL_IN_PLAIN, 2, 16, 1, 0>(a); }
        else if (a.c.nt == 2) { if (epi) conv_go<4, 2, CTL_IN_PLAIN, 1, 16, 2, 1>(a); else conv_go<4, 2, CTL_IN_PLAIN, 1, 16, 2, 0>(a); }
        else { if (epi) conv_go<4, 2, CTL_IN_PLAIN, 1, 16, 1, 1>(a); else conv_go<4, 2, CTL_IN_PLAIN, 1, 16, 1, 0>(a); }
    }
    else CTL_FAIL(CTL_EUNSUPPORTED, "conv_forward: no kernel for this combination");
    return CTL_OK;
}

extern "C" int ctl_conv_stats_blocks(const ctl_conv* d) {
    if (d->dt & CTL_DT_BF16) return ctl_conv_bf16_stats_blocks(d);
    if (d->dt & CTL_DT_X3) return ctl_conv_x3_stats_blocks(d);
    conv_call a = {};
    a.d = d;
    if (ctl_conv_pick_cfg(d, &a.c, 0) != CTL_OK) return -1;
    a.query = true;
    if (conv_dispatch(a) != CTL_OK) return -1;
    return a.grid_x * d->nsub;          // rows per group: [sub-problem][block]
}
extern "C" size_t ctl_conv_stats_floats(const ctl_conv* d) {
    const int b = ctl_conv_stats_blocks(d);
    return b < 0 ? 0 : (size_t)(d->groups > 1 ? d->groups : 1) * b * 2 * d->cout;
}

extern "C" int ctl_conv_forward(const ctl_conv* d, const float* x, const float* wpack, const float* bias,
                                const float* pro_scale, const float* pro_shift, const float* res,
                                const float* res_scale, const float* res_shift, float* y, float* stats_partial,
                                ctl_stream stream) {
    return ctl_conv_forward_ex(d, x, wpack, bias, pro_scale, pro_shift, res, res_scale, res_shift, nullptr, nullptr, y, stats_partial, nullptr, nullptr, stream);
}
// the tile configuration of d gives every wave a row pair: its CTL_EPI_TAILBWD epilogue can write the 2x2 sum-pool of g as well
extern "C" int ctl_conv_pool_ok(const ctl_conv* d) {
    ctl_conv_cfg c;
    if (!d) return 0;
    if (!(d->epi_flags & CTL_EPI_TAILBWD) || d->ks != 1 || d->stride != 1 || d->in_mode != CTL_IN_PLAIN || d->nsub != 1 || (d->out_h & 1) || (d->out_w & 1) ||
        d->out_h != d->hout || d->out_w != d->wout || d->cout % 16 != 0)
        return 0;
    if (ctl_conv_pick_cfg(d, &c, 0) != CTL_OK) return 0;
    return (c.mt == 4 && c.tw == 32) || (c.mt == 2 && c.tw == 16);
}
extern "C" int ctl_conv_forward_ex(const ctl_conv* d, const float* x, const float* wpack, const float* bias,
                                   const float* pro_scale, const float* pro_shift, const float* res,
                                   const float* res_scale, const float* res_shift, const float* res2, const float* x2, float* y,
                                   float* stats_partial, float* pool, float* xout, ctl_stream stream) {
    CTL_REQUIRE(d && x && wpack && y, "conv_forward: null argument");
    CTL_REQUIRE(!xout || d->pro_affine == 2, "conv_forward: `xout` (the virtual input written out) goes with the BatchNorm-backward prologue (pro_affine 2)");
    CTL_REQUIRE(!pool || ctl_conv_pool_ok(d), "conv_forward: `pool` needs a CTL_EPI_TAILBWD 1x1 conv with even output sizes whose tile configuration gives every wave a row pair (ctl_conv_pool_ok)");
    CTL_REQUIRE((d->dt & CTL_DT_BF16) || !(d->dt & (CTL_DT_X16 | CTL_DT_Y16 | CTL_DT_RES16)), "conv_forward: bf16-stored tensors need CTL_DT_BF16");
    conv_call a = {};
    a.d = d;
    int rc = ctl_conv_pick_cfg(d, &a.c, 0);
    if (rc != CTL_OK) return rc;
    CTL_REQUIRE(!(d->epi_flags & CTL_EPI_BIAS) || bias, "conv_forward: CTL_EPI_BIAS without bias");
    CTL_REQUIRE(!(d->epi_flags & CTL_EPI_RES) || (res && res_scale && res_shift), "conv_forward: CTL_EPI_RES without res");
    CTL_REQUIRE(!(d->epi_flags & CTL_EPI_STATS) || stats_partial, "conv_forward: CTL_EPI_STATS without a partial buffer");
    CTL_REQUIRE(!(d->epi_flags & CTL_EPI_BNBWD) || ((d->epi_flags & CTL_EPI_STATS) && res && res_scale && res_shift &&
                                                    !(d->epi_flags & (CTL_EPI_RES | CTL_EPI_ACCUM | CTL_EPI_BIAS)) && d->epi_act == CTL_ACT_NONE),
                "conv_forward: CTL_EPI_BNBWD needs CTL_EPI_STATS + res (= u) + res_scale/res_shift (BatchNorm coefficients) and nothing else");
    if (d->epi_flags & CTL_EPI_TAILBWD) {
        const int k = d->ks, s = d->stride, m = d->in_mode;
        CTL_REQUIRE((d->epi_flags & CTL_EPI_STATS) && res && res2 && !(d->epi_flags & (CTL_EPI_RES | CTL_EPI_BNBWD | CTL_EPI_BIAS)) &&
                    d->epi_act == CTL_ACT_NONE && d->cout % 16 == 0,
                    "conv_forward: CTL_EPI_TAILBWD needs CTL_EPI_STATS + res (= the block output) + res2 (= the BatchNorm input), cout %% 16 == 0, nothing else but CTL_EPI_ACCUM");
        CTL_REQUIRE((k == 1 && m == CTL_IN_PLAIN) || k == 2 || (k == 3 && s == 1 && m == CTL_IN_ZINS2),
                    "conv_forward: CTL_EPI_TAILBWD is built for the launches that write a block's output gradient (1x1, 2x2, zero-insert 3x3)");
        CTL_REQUIRE(d->epi_slope >= 0.f && d->epi_slope <= 1.f, "conv_forward: LeakyReLU slope must be in [0, 1]");
    }
    CTL_REQUIRE(d->pro_affine >= 0 && d->pro_affine <= 2, "conv_forward: pro_affine must be 0, 1 or 2");
    CTL_REQUIRE(d->pro_affine != 1 || (pro_scale && pro_shift), "conv_forward: prologue without scale/shift");
    CTL_REQUIRE(d->pro_affine != 2 || (x2 && pro_scale && d->cin % 16 == 0 && d->in_mode == CTL_IN_PLAIN &&
                                       ((d->ks == 3 && d->stride == 1) || (d->ks == 4 && d->stride == 2)) && !(d->epi_flags & CTL_EPI_TAILBWD)),
                "conv_forward: the BatchNorm-backward prologue (pro_affine 2) needs x2 + coefficients, cin %% 16 == 0 and a plain 3x3 stride-1 or 4x4 stride-2 conv");
    CTL_REQUIRE(!d->pro_affine || (d->groups > 1 ? d->groups : 1) * d->cin <= CTL_PRO_MAX, "conv_forward: groups * cin = %d prologue coefficients exceed %d", (d->groups > 1 ? d->groups : 1) * d->cin, CTL_PRO_MAX);
    CTL_REQUIRE(d->pro_affine != 1 || (d->pro_slope >= 0.f && d->pro_slope <= 1.f), "conv_forward: prologue slope must be in [0, 1]");
    CTL_REQUIRE(d->epi_act != CTL_ACT_LEAKY || (d->epi_slope >= 0.f && d->epi_slope <= 1.f), "conv_forward: LeakyReLU slope must be in [0, 1]");
    CTL_REQUIRE(d->n > 0 && d->hout > 0 && d->wout > 0, "conv_forward: empty problem");
    CTL_REQUIRE(d->groups >= 0 && (d->groups <= 1 || d->n % d->groups == 0), "conv_forward: n=%d is not divisible into %d groups", d->n, d->groups);
    CTL_REQUIRE((int64_t)d->n * d->hin * d->win * d->cin * 4 < (1ll << 31) &&
                (int64_t)d->n * d->out_h * d->out_w * d->cout * 4 < (1ll << 31),
                "conv_forward: tensors must stay below 2 GiB (32-bit buffer offsets)");
    if (d->dt & CTL_DT_X3)
        return ctl_conv_forward_x3(d, x, wpack, bias, pro_scale, pro_shift, res, res_scale, res_shift, res2, x2, y, stats_partial, pool, xout, stream);
    if (d->dt & CTL_DT_BF16) {
        const int ptok16 = ctl_prof_begin("conv_igemm_bf16", d, &a.c, a.c.nt, (hipStream_t)stream);
        rc = ctl_conv_forward_bf16(d, x, x2, wpack, bias, pro_scale, pro_shift, res, res_scale, res_shift, res2, y, stats_partial, pool, xout, stream);
        ctl_prof_end(ptok16, (hipStream_t)stream);
        return rc;
    }
    a.x = x; a.wpack = wpack; a.bias = bias; a.pro_scale = pro_scale; a.pro_shift = pro_shift; a.res = res;
    a.res_scale = res_scale; a.res_shift = res_shift; a.res2 = res2; a.x2 = x2; a.y = y; a.stats_partial = stats_partial; a.pool = pool; a.xout = xout;
    a.stream = (hipStream_t)stream;
    const int ptok = ctl_prof_begin("conv_igemm", d, &a.c, a.c.nt, a.stream);
    rc = conv_dispatch(a);
    if (rc != CTL_OK) return rc;
    ctl_prof_end(ptok, a.stream);
    CTL_LAUNCH_CHECK("conv_forward");
    return CTL_OK;
}

// ---- wgrad host side
#ifndef CTL_WGRAD_PIPE
#define CTL_WGRAD_PIPE 1
#endif
struct wgrad_cfg { ctl_conv_cfg c; int ntw, splits, ntiles, cin_p, cout_p; };

struct wgrad_call {
    const ctl_conv* d; wgrad_cfg* w;
    const float *x, *pro_scale, *pro_shift, *dy, *dy2, *dy_coef;
    float *w_partial, *b_partial;
    hipStream_t stream;
    bool query;          // only compute w->splits
};

// splits = blocks along x: the grid (splits x cin chunks x cout tile groups) is what is resident at once (256 CUs x the
// kernel's occupancy, at most CTL_PERSIST = 4 per CU).  More blocks would run in rounds and pay the per-block setup and the
// cross-wave reduction + partial write again (each ~1.5 tiles worth of time), and write / re-read more partials.
template <int KS, int S, int MODE, int MT, int TW, int NTW>
static void wgrad_go(wgrad_call& a) {
    static int occ = 0;
    if (!occ) {
        int n = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, conv_wgrad_kernel<KS, S, MODE, MT, TW, NTW>, 256, 0) != hipSuccess || n < 1) {
            (void)hipGetLastError();
            n = 2;
        }
        occ = n;
    }
    wgrad_cfg& w = *a.w;
    static int per_cu = -1;
    if (per_cu < 0) {
        // ONE block per CU: the weight gradients co-run with the other launch chain, and every block costs a partial tensor that the
        // reduction at the end of the plan reads back (measured, whole step: 256 blocks 18.23 ms, 512-768 (occupancy) 18.40, 768 19.05,
        // 128 20.98).  CTL_WGRAD_PERSIST / CTL_WGRAD_SLOTS are the tuning hooks.
        per_cu = ctl_tune_int("CTL_WGRAD_PERSIST", 1);
        if (per_cu < 1) per_cu = 1;
    }
    const int par = w.c.g * (w.c.cot / NTW);
    static const int slots = ctl_tune_int("CTL_WGRAD_SLOTS", 0);      // tuning hook: total blocks
    // (the HBM-side members of the family -- 1x1 convs, the <= 4-channel first layers -- run at 0.28 of 8 TB/s with one block per CU and
    // 0.37 with two or four.  While the backward still had its element-wise passes the STEP was no faster for it (17.37-17.53 vs
    // 17.43-17.59 ms); without them two blocks give 16.80 -> 16.72 ms, four 16.79: tuning hook CTL_WGRAD_NARROW_PERSIST)
    static const int narrow_cu = ctl_tune_int("CTL_WGRAD_NARROW_PERSIST", 2);
    const int cu_blocks = (KS == 1 || MODE == CTL_IN_C4) ? narrow_cu : per_cu;
    int splits = (slots > 0 ? slots : ctl_num_cus() * (occ < cu_blocks ? occ : cu_blocks)) / par;
    if (splits > 512) splits = 512;
    if (splits > w.ntiles) splits = w.ntiles;
    if (splits < 1) splits = 1;
    w.splits = splits;
    if (a.query) return;
    const dim3 grid((unsigned)splits, (unsigned)w.c.g, (unsigned)(w.c.cot / NTW));
    if constexpr (KS == 3 && S == 1 && MODE != CTL_IN_C4) {
        // Two LDS images, staging inside the MFMA stream.  Measured alone (profiles/r3_wgrad_pipe.txt): with the two-tensor output
        // gradient 5-8 % faster than the single-image loop on layers with >= 16 tiles per block (its extra loads and FMAs ride behind
        // the MFMAs), equal without it, 10 % slower at 4 tiles per block (one more exposed load latency per block) -> 1 = the
        // two-tensor launches with >= 8 tiles per block, 2 = every eligible launch, 0 = off
        static const int pipe = ctl_tune_int("CTL_WGRAD_PIPE", CTL_WGRAD_PIPE);
        const bool pipe_on = pipe >= 2 || (pipe == 1 && a.dy2 && w.ntiles >= 8 * splits);
        if (pipe_on && a.d->cin % 4 == 0 && a.d->cout % 4 == 0 && a.d->cin <= CTL_PRO_MAX) {
            if (a.dy2)
                conv_wgrad_kernel<KS, S, MODE, MT, TW, NTW, true, true><<<grid, dim3(256), 0, a.stream>>>(
                    *a.d, a.x, a.pro_scale, a.pro_shift, a.dy, a.w_partial, a.b_partial, w.c.tiles_h, w.c.tiles_w, w.ntiles, w.cin_p, w.cout_p, a.dy2, a.dy_coef);
            else
                conv_wgrad_kernel<KS, S, MODE, MT, TW, NTW, false, true><<<grid, dim3(256), 0, a.stream>>>(
                    *a.d, a.x, a.pro_scale, a.pro_shift, a.dy, a.w_partial, a.b_partial, w.c.tiles_h, w.c.tiles_w, w.ntiles, w.cin_p, w.cout_p, nullptr, nullptr);
            return;
        }
    }
    if constexpr (KS == 3 && S == 1) {      // the two-tensor output gradient: the 3x3 convs of the residual blocks and of the encoder heads
        if (a.dy2) {
            conv_wgrad_kernel<KS, S, MODE, MT, TW, NTW, true><<<grid, dim3(256), 0, a.stream>>>(
                *a.d, a.x, a.pro_scale, a.pro_shift, a.dy, a.w_partial, a.b_partial, w.c.tiles_h, w.c.tiles_w, w.ntiles, w.cin_p, w.cout_p, a.dy2, a.dy_coef);
            return;
        }
    }
    conv_wgrad_kernel<KS, S, MODE, MT, TW, NTW><<<grid, dim3(256), 0, a.stream>>>(
        *a.d, a.x, a.pro_scale, a.pro_shift, a.dy, a.w_partial, a.b_partial, w.c.tiles_h, w.c.tiles_w, w.ntiles, w.cin_p, w.cout_p, nullptr, nullptr);
}
template <int KS, int S, int MODE>
static void wgrad_go_tile(wgrad_call& a) {
    const wgrad_cfg& w = *a.w;
    if (w.c.mt == 2) { if (w.ntw == 2) wgrad_go<KS, S, MODE, 2, 16, 2>(a); else wgrad_go<KS, S, MODE, 2, 16, 1>(a); }
    else { if (w.ntw == 2) wgrad_go<KS, S, MODE, 1, 16, 2>(a); else wgrad_go<KS, S, MODE, 1, 16, 1>(a); }
}
static int wgrad_dispatch(wgrad_call& a) {
    const int k = a.d->ks, s = a.d->stride, m = a.d->in_mode;
    if (k == 3 && s == 1 && m == CTL_IN_PLAIN) wgrad_go_tile<3, 1, CTL_IN_PLAIN>(a);
    else if (k == 3 && s == 1 && m == CTL_IN_UP2) wgrad_go_tile<3, 1, CTL_IN_UP2>(a);
    else if (k == 3 && s == 1 && m == CTL_IN_C4) wgrad_go_tile<3, 1, CTL_IN_C4>(a);
    else if (k == 3 && s == 2) wgrad_go_tile<3, 2, CTL_IN_PLAIN>(a);
    else if (k == 1 && m == CTL_IN_PLAIN) wgrad_go_tile<1, 1, CTL_IN_PLAIN>(a);
    else if (k == 1 && m == CTL_IN_UP2) wgrad_go_tile<1, 1, CTL_IN_UP2>(a);
    else if (k == 2 && s == 2) wgrad_go_tile<2, 2, CTL_IN_PLAIN>(a);
    else CTL_FAIL(CTL_EUNSUPPORTED, "conv_wgrad: no kernel for this combination");
    return CTL_OK;
}

static int wgrad_pick(const ctl_conv* d, wgrad_cfg* w) {
    CTL_REQUIRE(d->nsub == 1, "wgrad: nsub must be 1");
    CTL_REQUIRE(d->in_mode != CTL_IN_ZINS2, "wgrad: zero-insert input is not a forward mode");
    int rc = ctl_conv_pick_cfg(d, &w->c, 1);
    if (rc != CTL_OK) return rc;
    w->ntw = (w->c.cot >= 2 && w->c.cot % 2 == 0) ? 2 : 1;
    w->ntiles = d->n * w->c.tiles_h * w->c.tiles_w;
    w->cin_p = w->c.g * 16;
    w->cout_p = w->c.cot * 16;
    if (d->dt & CTL_DT_BF16) {            // the bf16 kernels have their own occupancy, hence their own split count
        w->splits = ctl_wgrad_bf16_splits(d);
        return w->splits > 0 ? CTL_OK : CTL_EUNSUPPORTED;
    }
    if (d->dt & CTL_DT_X3) {              // ... and so has the X3 kernel (ctl_wgrad_x3.hip)
        w->splits = ctl_wgrad_x3_splits(d);
        return w->splits > 0 ? CTL_OK : CTL_EUNSUPPORTED;
    }
    wgrad_call a = {};
    a.d = d; a.w = w; a.query = true;
    return wgrad_dispatch(a);
}

extern "C" int ctl_wgrad_splits(const ctl_conv* d) {
    wgrad_cfg w;
    return wgrad_pick(d, &w) == CTL_OK ? w.splits : -1;
}
extern "C" size_t ctl_wgrad_partial_floats(const ctl_conv* d) {
    wgrad_cfg w;
    if (wgrad_pick(d, &w) != CTL_OK) return 0;
    return (size_t)w.splits * d->ks * d->ks * w.cin_p * w.cout_p;
}
extern "C" size_t ctl_wgrad_bias_partial_floats(const ctl_conv* d) {
    wgrad_cfg w;
    if (wgrad_pick(d, &w) != CTL_OK) return 0;
    return (size_t)w.splits * w.cout_p;
}

extern "C" int ctl_conv_wgrad(const ctl_conv* d, const float* x, const float* pro_scale, const float* pro_shift,
                              const float* dy, float* w_partial, float* b_partial, ctl_stream stream) {
    return ctl_conv_wgrad_ex(d, x, pro_scale, pro_shift, dy, nullptr, nullptr, w_partial, b_partial, stream);
}
extern "C" int ctl_conv_wgrad_ex(const ctl_conv* d, const float* x, const float* pro_scale, const float* pro_shift,
                                 const float* dy, const float* dy2, const float* dy_coef, float* w_partial, float* b_partial,
                                 ctl_stream stream) {
    CTL_REQUIRE(d && x && dy && w_partial, "conv_wgrad: null argument");
    CTL_REQUIRE(d->pro_affine == 0 || d->pro_affine == 1, "conv_wgrad: pro_affine must be 0 or 1");
    CTL_REQUIRE(!dy2 || (dy_coef && d->ks == 3 && d->stride == 1 && d->cout % 16 == 0 &&
                         (d->groups > 1 ? d->groups : 1) * d->cout <= CTL_PRO_MAX),
                "conv_wgrad: the two-tensor output gradient needs coefficients, a 3x3 stride-1 conv, cout %% 16 == 0 and groups * cout <= %d", CTL_PRO_MAX);
    CTL_REQUIRE(!d->pro_affine || (pro_scale && pro_shift), "conv_wgrad: prologue without scale/shift");
    CTL_REQUIRE(!d->pro_affine || (d->groups > 1 ? d->groups : 1) * d->cin <= CTL_PRO_MAX, "conv_wgrad: groups * cin = %d prologue coefficients exceed %d", (d->groups > 1 ? d->groups : 1) * d->cin, CTL_PRO_MAX);
    CTL_REQUIRE(!d->pro_affine || (d->pro_slope >= 0.f && d->pro_slope <= 1.f), "conv_wgrad: prologue slope must be in [0, 1]");
    CTL_REQUIRE((int64_t)d->n * d->hin * d->win * d->cin * 4 < (1ll << 31) &&
                (int64_t)d->n * d->hout * d->wout * d->cout * 4 < (1ll << 31),
                "conv_wgrad: tensors must stay below 2 GiB (32-bit buffer offsets)");
    wgrad_cfg w;
    int rc = wgrad_pick(d, &w);
    if (rc != CTL_OK) return rc;
    if (d->dt & CTL_DT_X3) return ctl_conv_wgrad_x3(d, x, pro_scale, pro_shift, dy, dy2, dy_coef, w_partial, b_partial, stream);
    if (d->dt & CTL_DT_BF16) {
        const int ptok16 = ctl_prof_begin("conv_wgrad_bf16", d, &w.c, w.ntw, (hipStream_t)stream, dy2 != nullptr);
        rc = ctl_conv_wgrad_bf16(d, x, pro_scale, pro_shift, dy, dy2, dy_coef, w_partial, b_partial, stream);
        ctl_prof_end(ptok16, (hipStream_t)stream);
        return rc;
    }
    wgrad_call a = {};
    a.d = d; a.w = &w; a.x = x; a.pro_scale = pro_scale; a.pro_shift = pro_shift; a.dy = dy; a.dy2 = dy2; a.dy_coef = dy_coef; a.w_partial = w_partial;
    a.b_partial = b_partial; a.stream = (hipStream_t)stream;
    const int ptok = ctl_prof_begin("conv_wgrad", d, &w.c, w.ntw, a.stream, dy2 != nullptr);
    rc = wgrad_dispatch(a);
    if (rc != CTL_OK) return rc;
    ctl_prof_end(ptok, a.stream);
    CTL_LAUNCH_CHECK("conv_wgrad");
    return CTL_OK;
}

extern "C" int ctl_wgrad_reduce(const ctl_conv* d, const float* w_partial, const float* b_partial, float* dw,
                                int64_t s_co, int64_t s_ci, int64_t s_kh, int64_t s_kw, float* dbias,
                                int32_t accumulate, ctl_stream stream) {
    CTL_REQUIRE(d && w_partial && dw, "wgrad_reduce: null argument");
    CTL_REQUIRE(!dbias || b_partial, "wgrad_reduce: dbias without b_partial");
    wgrad_cfg w;
    int rc = wgrad_pick(d, &w);
    if (rc != CTL_OK) return rc;
    const int taps = d->ks * d->ks;
    const int64_t total = (int64_t)taps * d->cin * d->cout + (dbias ? d->cout : 0);
    wgrad_reduce_kernel<<<dim3((unsigned)ctl_cdiv64(total, 32)), dim3(256), 0, (hipStream_t)stream>>>(
        w_partial, b_partial, w.splits, taps, d->ks, d->cin, d->cout, w.cin_p, w.cout_p, dw, s_co, s_ci, s_kh, s_kw,
        dbias, accumulate);
    CTL_LAUNCH_CHECK("wgrad_reduce");
    return CTL_OK;
}
