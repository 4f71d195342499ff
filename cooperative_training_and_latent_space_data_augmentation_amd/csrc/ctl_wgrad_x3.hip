// X3 weight gradient for gfx950 (MI355X): the fp32 weight gradient (same tensors, same partial layout and deterministic split reduction
// as conv_wgrad_kernel in ctl_conv.hip) contracted on v_mfma_f32_16x16x32_bf16 over the exact three-way bf16 split of both operands
// (ctl_conv_x3_stage.h).  Structure of the bf16 family's kernel (ctl_wgrad_bf16.hip):
//   dW[tap][ci][co] = sum_pixels x_virtual[pixel*S + tap - pad][ci] * dy[pixel][co]:  D[ci][co] += A[ci][k = pixel] B[pixel][co], K = 32 pixels.
// Both operands want 8 PIXELS of one channel per lane -- the transpose of the [pixel][16 channel] LDS images -- which ds_read_b64_tr_b16
// delivers: per 16-lane group it reads 4 rows (pixels) x 16 columns (channels) of 16-bit elements and hands lane i column i.  Here every
// operand exists three times (hi | mid | lo image), and a (tap, cout tile) step is six MFMAs
//     x_hi*dy_hi + x_hi*dy_mid + x_mid*dy_hi + x_mid*dy_mid + x_hi*dy_lo + x_lo*dy_hi
// -- 108 MFMAs of 16 cycles per 32 pixels, 16 cin and 32 cout where the fp32 kernel issues 144 of 32 cycles.
// The bias gradient (column sums of dy) is taken from the staged fp32 values, before the split.
#include <utility>

#include <type_traits>

#include "ctl_conv_x3_stage.h"

typedef short x3_s16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ x3_bf16x8 x3_tr_read8(const unsigned char* a0, const unsigned char* a1) {
    const x3_s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((x3_s16x4 __attribute__((address_space(3)))*)(a0));
    const x3_s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((x3_s16x4 __attribute__((address_space(3)))*)(a1));
    typedef short s16x8 __attribute__((ext_vector_type(8)));
    return __builtin_bit_cast(x3_bf16x8, s16x8{lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w});
}

#ifndef CTL_X3W_LB
#define CTL_X3W_LB 2
#endif
#ifndef CTL_X3W_ABLATE
#define CTL_X3W_ABLATE 0      // timing ablations (variant builds only, WRONG results): 1 no split of dy (with -DCTL_X3_ABLATE=1: nor of x),
#endif                        // 2 no global loads after the first tile, 4 no MFMAs, 8 no A-operand reads after tap 0, 16 no staging stores,
                              // 32 no cross-wave reduction (one wave's sums are written), 64 return at once, 128 no tile loop at all
// DY2: the output gradient is the virtual BatchNorm-backward result  A * dy + B * dy2 + C  (coefficients [group][3][cout] as the finalize
// writes them; dy = g, dy2 = the BatchNorm input), evaluated in fp32 in this staging, then split
template <int KS, int S, int MODE, int MT, int NTW, bool DY2>
__global__ __launch_bounds__(256, CTL_X3W_LB) void conv_wgrad_x3_kernel(const ctl_conv d, const float* __restrict__ x,
                                                                       const float* __restrict__ pro_scale, const float* __restrict__ pro_shift,
                                                                       const float* __restrict__ dy, const float* __restrict__ dy2,
                                                                       const float* __restrict__ dy_coef, float* __restrict__ w_partial,
                                                                       float* __restrict__ b_partial, int tiles_h, int tiles_w, int ntiles,
                                                                       int cin_p, int cout_p) {
    if constexpr (CTL_X3W_ABLATE & 64) return;
    constexpr int TW = 16;
    using G = Geom<KS, S, MT, TW>;
    using XS = XStage3<KS, S, MODE, MT, TW, false, false>;
    constexpr int TAPS = KS * KS;
    constexpr int KB = G::TP / 32;                       // k-blocks per tile: 4 (8x16) or 2 (4x16)
    constexpr int DYI = NTW * G::TP * 32;                // one split image of the dy tile: [cout tile][pixel][16 ch] bf16
    constexpr int DYT_BYTES = 3 * DYI;
    constexpr int RED_BYTES = 4 * NTW * 256 * 4;
    constexpr int MAIN_BYTES = (XS::XT_BYTES + DYT_BYTES > RED_BYTES) ? (XS::XT_BYTES + DYT_BYTES) : RED_BYTES;
    __shared__ __attribute__((aligned(16))) unsigned char smem[MAIN_BYTES + (DY2 ? 5 : 2) * CTL_PRO_MAX * 4];
    unsigned char* xt = smem;
    unsigned char* dyt = smem + XS::XT_BYTES;
    float* cf_scale = reinterpret_cast<float*>(smem + MAIN_BYTES);
    float* cf_shift = cf_scale + CTL_PRO_MAX;
    float* cd = cf_shift + CTL_PRO_MAX;                  // DY2: A | B | C, [group][cout] each

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int p = lane & 15, q = lane >> 4;
    const int g = blockIdx.y;
    const int cot0 = blockIdx.z * NTW;
    const int group_n = d.n / (d.groups > 1 ? d.groups : 1);

    f32x4 acc[TAPS][NTW];
#pragma unroll
    for (int a = 0; a < TAPS; ++a)
#pragma unroll
        for (int t = 0; t < NTW; ++t) acc[a][t] = f32x4{0.f, 0.f, 0.f, 0.f};

    const __amdgpu_buffer_rsrc_t rx = ctl_rsrc(x, (int64_t)d.n * d.hin * d.win * d.cin * 4);
    const __amdgpu_buffer_rsrc_t rdy = ctl_rsrc(dy, (int64_t)d.n * d.hout * d.wout * d.cout * 4);
    const __amdgpu_buffer_rsrc_t rdy2 = DY2 ? ctl_rsrc(dy2, (int64_t)d.n * d.hout * d.wout * d.cout * 4) : rdy;
    XS xs;
    xs.init(d);
    // dy tile: units of 8 channels (two 16-byte fp32 loads); all units of a thread share (cout tile, half): 256 % (2 NTW) == 0
    constexpr int DU = G::TP * NTW * 2, ND = (DU + 255) / 256;
    f32x4 dv0[ND], dv1[ND], dw0[DY2 ? ND : 1], dw1[DY2 ? ND : 1];
    int drel[ND], drc[ND], dlds[ND];
    unsigned dmask = 0;
    bool dall = false;
    f32x4 bs0 = {0.f, 0.f, 0.f, 0.f}, bs1 = bs0;        // bias gradient: this thread's 8 channels summed over its pixels (fp32, before the split)
    const int drest = tid % (NTW * 2), dt_ = drest >> 1, dh = drest & 1;
    const int dco = (cot0 + dt_) * 16 + dh * 8;          // first channel of this thread's units (cout is a multiple of 16: in range)
#pragma unroll
    for (int i = 0; i < ND; ++i) {
        const int u = tid + i * 256;
        const int pix = u / (NTW * 2);
        const int pr = pix / TW, pc = pix % TW;
        drc[i] = (u < DU) ? (pr | (pc << 16)) : 0x7fff7fff;            // (4x16 tiles with one cout tile: 128 units, half of the threads have none)
        drel[i] = (u < DU) ? ((pr * d.wout + pc) * d.cout + dco) * 4 : CTL_OOB;
        dlds[i] = (dt_ * G::TP + pix) * 32 + dh * 16;
    }
    auto dyload = [&](int n, int ho0, int wo0) {
        const int tb = ((n * d.hout + ho0) * d.wout + wo0) * d.cout * 4;
        dall = ho0 + G::TH <= d.hout && wo0 + TW <= d.wout;
        if (dall) {
#pragma unroll
            for (int i = 0; i < ND; ++i) { dv0[i] = ctl_bload4s(rdy, drel[i], tb); dv1[i] = ctl_bload4s(rdy, drel[i] + 16, tb); }
            if constexpr (DY2) {
#pragma unroll
                for (int i = 0; i < ND; ++i) { dw0[i] = ctl_bload4s(rdy2, drel[i], tb); dw1[i] = ctl_bload4s(rdy2, drel[i] + 16, tb); }
            }
            return;
        }
        dmask = 0;
#pragma unroll
        for (int i = 0; i < ND; ++i) {
            const bool ok = (unsigned)(ho0 + (drc[i] & 0xffff)) < (unsigned)d.hout && (unsigned)(wo0 + (drc[i] >> 16)) < (unsigned)d.wout;
            const int vo = ok ? (tb + drel[i]) : CTL_OOB, vo1 = ok ? (tb + drel[i] + 16) : CTL_OOB;
            dv0[i] = ctl_bload4(rdy, vo); dv1[i] = ctl_bload4(rdy, vo1);
            if constexpr (DY2) { dw0[i] = ctl_bload4(rdy2, vo); dw1[i] = ctl_bload4(rdy2, vo1); }
            dmask |= ok ? (1u << i) : 0u;
        }
    };
    auto dystore = [&](int goff) {
        f32x4 a0, a1, b0, b1, c0, c1;
        if constexpr (DY2) {
            const float* cc = cd + goff + dco;
            a0 = *reinterpret_cast<const f32x4*>(cc); a1 = *reinterpret_cast<const f32x4*>(cc + 4);
            b0 = *reinterpret_cast<const f32x4*>(cc + CTL_PRO_MAX); b1 = *reinterpret_cast<const f32x4*>(cc + CTL_PRO_MAX + 4);
            c0 = *reinterpret_cast<const f32x4*>(cc + 2 * CTL_PRO_MAX); c1 = *reinterpret_cast<const f32x4*>(cc + 2 * CTL_PRO_MAX + 4);
        }
        const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < ND; ++i) {
            f32x4 lo = dv0[i], hi = dv1[i];
            if constexpr (DY2) {      // pixels past the image contribute nothing (C alone would)
                const bool in = dall || ((dmask >> i) & 1u);
                lo = in ? (a0 * lo + b0 * dw0[i] + c0) : zero;
                hi = in ? (a1 * hi + b1 * dw1[i] + c1) : zero;
            }
            if (DU % 256 != 0 && tid + i * 256 >= DU) continue;
            bs0 += lo; bs1 += hi;
            u32x4 ph, pm, pl;
            if constexpr (CTL_X3W_ABLATE & 1) { ph = x3_pack8(lo, hi); pm = ph; pl = ph; }
            else x3_split8(lo, hi, ph, pm, pl);
            *reinterpret_cast<u32x4*>(dyt + dlds[i]) = ph;
            *reinterpret_cast<u32x4*>(dyt + dlds[i] + DYI) = pm;
            *reinterpret_cast<u32x4*>(dyt + dlds[i] + 2 * DYI) = pl;
        }
    };
    // transposed-read addresses of this lane: block row q' = (lane & 15) >> 2 (pixel), 8-byte column chunk p' = lane & 3
    const int qp = (lane & 15) >> 2, pp = lane & 3;
    const int krow = (q >> 1), kcol0 = 8 * (q & 1);     // tile row (within the k-block's two rows) and first column of this lane group
    const __amdgpu_buffer_rsrc_t rnone = ctl_rsrc((const void*)nullptr, 0);
    TileWalk cur;
    cur.init(blockIdx.x, gridDim.x, tiles_h, tiles_w);
    if ((int)blockIdx.x < ntiles) {
        xs.load(rx, rx, d, cur.n, cur.th * G::TH, cur.tw * TW, g);
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
        xs.store(reinterpret_cast<float*>(xt), d, g, cf_scale, cf_shift, (cur.n / group_n) * d.cin, nullptr, rnone, false);
        dystore((cur.n / group_n) * d.cout);
    }
    __syncthreads();
    for (int tile = blockIdx.x; tile < ((CTL_X3W_ABLATE & 128) ? 0 : ntiles); tile += gridDim.x) {
        const bool has_next = tile + (int)gridDim.x < ntiles;
        if (has_next) {
            cur.next();
            if constexpr (!(CTL_X3W_ABLATE & 2)) {
                xs.load(rx, rx, d, cur.n, cur.th * G::TH, cur.tw * TW, g);
                dyload(cur.n, cur.th * G::TH, cur.tw * TW);
            }
        }
        for (int kb = wave; kb < KB; kb += 4) {          // (4x16 tiles: two k-blocks, waves 2 and 3 only stage; 16x16 tiles: two k-blocks per wave)
            const int tr = kb * 2 + krow;                // tile row of this lane group's 8 pixels
            x3_bf16x8 bf[3][NTW];
#pragma unroll
            for (int sp = 0; sp < 3; ++sp)
#pragma unroll
                for (int t = 0; t < NTW; ++t) {
                    const unsigned char* b0 = dyt + sp * DYI + ((t * G::TP + tr * TW + kcol0 + qp) * 32) + pp * 8;
                    bf[sp][t] = x3_tr_read8(b0, b0 + 4 * 32);
                }
            // the A operands of tap+1 are requested before the MFMAs of tap; the empty asm pins that order (see ctl_wgrad_bf16.hip)
            auto a_operand = [&](int tap, x3_bf16x8* af) {
                const int kh = tap / KS, kw = tap % KS;
                const int c0 = G::ldscol((kcol0 + qp) * S + kw);
                const unsigned char* a0 = xt + (((tr * S + kh) * G::IWP + c0) * 32) + pp * 8;
#pragma unroll
                for (int sp = 0; sp < 3; ++sp) af[sp] = x3_tr_read8(a0 + sp * XS::SPLIT, a0 + sp * XS::SPLIT + 4 * 32);
            };
            x3_bf16x8 af[2][3];
            a_operand(0, af[0]);
#pragma unroll
            for (int tap = 0; tap < TAPS; ++tap) {
                if constexpr (!(CTL_X3W_ABLATE & 8)) { if (tap + 1 < TAPS) a_operand(tap + 1, af[(tap + 1) & 1]); }
                x3_bf16x8* a = af[(CTL_X3W_ABLATE & 8) ? 0 : (tap & 1)];
                asm volatile("" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]) : : "memory");
#define CTL_X3W_MFMA(SA, SB) _Pragma("unroll") for (int t = 0; t < NTW; ++t) { \
                    if constexpr (CTL_X3W_ABLATE & 4) { if (SA == 0 && SB == 0) acc[tap][t].x += (float)a[SA][0] * (float)bf[SB][t][0]; } \
                    else acc[tap][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[SA], bf[SB][t], acc[tap][t], 0, 0, 0); }
                CTL_X3W_MFMA(2, 0)
                CTL_X3W_MFMA(0, 2)
                CTL_X3W_MFMA(1, 1)
                CTL_X3W_MFMA(1, 0)
                CTL_X3W_MFMA(0, 1)
                CTL_X3W_MFMA(0, 0)
#undef CTL_X3W_MFMA
            }
        }
        ctl_barrier_lds_reads_done();
        if (has_next && !(CTL_X3W_ABLATE & 16)) {
            xs.store(reinterpret_cast<float*>(xt), d, g, cf_scale, cf_shift, (cur.n / group_n) * d.cin, nullptr, rnone, false);
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
    const int64_t split_base = (int64_t)blockIdx.x * TAPS * cin_p * cout_p;
    if constexpr (CTL_X3W_ABLATE & 32) {
        if (wave == 0)
#pragma unroll
            for (int tap = 0; tap < TAPS; ++tap)
#pragma unroll
                for (int t = 0; t < NTW; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int co = (cot0 + t) * 16 + (lane & 15), ci = g * 16 + (lane >> 4) * 4 + r;
                        if (co < cout_p) w_partial[split_base + ((int64_t)tap * cin_p + ci) * cout_p + co] = acc[tap][t][r];
                    }
        return;
    }
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
    if (g == 0 && b_partial != nullptr) {      // bias gradient: the threads' channel sums, thread (pixel slot, cout tile, half) -> [256][8] floats
        ctl_barrier_lds_reads_done();
        *reinterpret_cast<f32x4*>(red + tid * 8) = bs0;
        *reinterpret_cast<f32x4*>(red + tid * 8 + 4) = bs1;
        ctl_barrier_lds_writes_done();
        if (tid < NTW * 16) {
            const int t = tid >> 4, c = tid & 15, h = c >> 3, j = c & 7;
            float v = 0.f;
            for (int s = t * 2 + h; s < 256; s += NTW * 2) v += red[s * 8 + j];      // fixed order: deterministic
            const int co = cot0 * 16 + tid;
            if (co < cout_p) b_partial[(int64_t)blockIdx.x * cout_p + co] = v;
        }
    }
}

// ------------------------------------------------------------------------------------------------ producer / consumer form (round 5)
// conv_wgrad_x3pc16_kernel: 32 cin x 32 cout per block on v_mfma_f32_32x32x16_bf16, ONE 1024-thread block per CU.
//   * the 32x32x16 form issues HALF the matrix instructions of the 16x16x32 form for the same products, and a 32 x 32 output block stages dy
//     once per 32 input channels instead of once per 16 (and x once per 32 output channels);
//   * producer waves keep two tiles of loads in flight in registers, evaluate the prologue / the virtual output gradient, split and write the
//     three bf16 images of tile `it` into LDS image it & 1; consumer waves hold the tap accumulators and do nothing but transposed operand
//     reads and MFMAs: one s_barrier per tile (see conv_igemm_kernel, PC, for the protocol and for why every load in the producer loop is
//     unconditional and single-path).
//   D[ci][co] += A[ci][k = pixel] B[pixel][co], K = 16 pixels = one tile row; the four row waves' sums meet in LDS at the end of the block.
// Lane maps (v_mfma_f32_32x32x16_bf16): A[row = lane & 31][k = 8 (lane >> 5) + j], B[k = 8 (lane >> 5) + j][col = lane & 31], D[row = (r & 3)
// + 8 (r >> 2) + 4 (lane >> 5)][col = lane & 31]: lane group q = lane >> 4 reads the 16-channel chunk q & 1 and the pixel half q >> 1 of
// its operand with the transposed reads of the kernel above.
// Bank phase: a transposed read is served in two halves of 32 lanes; lanes 0-15 (chunk 0) and 16-31 (chunk 1) each cover 128 contiguous bytes
// = 32 of the 64 banks, so the second chunk of every operand sits 128 B (mod 256) behind the first (chunk strides that are multiples of 256 B:
// SQ_LDS_BANK_CONFLICT 0.37-0.45 of the LDS-active cycles in the single-role kernel, whose two lane groups read pixels 256 B apart).
// An eight-wave form (one producer + one consumer wave per SIMD, all nine taps on the consumer) and a single-stream software-pipelined form
// were built, measured and removed: profiles/r5_wgrad_pc_experiments.txt has their phase timers and the micro-benchmarks behind them.
typedef float f32x16 __attribute__((ext_vector_type(16)));
// A launch serves a GROUP of problems (deferred weight gradients of one backward plan, nets.PlanBuilder.flush_wgrad_groups): the 256 CUs are
// dealt to the members in proportion to their work, one block = one job = (member, pixel split, 32-cin block, 32-cout block).  Why groups:
// per launch ~15 us of a 40 us n = 16 layer are fixed (launch, first loads + first staging exposed, reduction tail, partial write) while the
// steady state is ~3.1 us per 32 x 32 x 128-pixel tile for every kernel form tried (profiles/r5_wgrad_pc_experiments.txt) -- a member of an
// 8-member group gets an eighth of the CUs, eight times the tiles per block and an eighth of the split-K partials.
#define CTL_WG_MAX 8
struct wg_member {
    ctl_conv d;
    const float *x, *pro_scale, *pro_shift, *dy, *dy2, *dy_coef;
    float *w_partial, *b_partial;
    int tiles_h, tiles_w, ntiles, cin_p, cout_p, splits, job0, pad_;
};
struct wg_group { int n, pad_[3]; wg_member m[CTL_WG_MAX]; };
// Sixteen waves: per SIMD TWO producer waves and TWO consumer waves.  In the eight-wave form (one + one) the producer wave was the critical
// path -- 3000 cycles of staging per tile alone, 5750 beside the MFMA wave of its SIMD (with or without that wave's operand reads) -- while the
// MFMA wave needed 2900 and waited for the rest.  Here every producer wave stages half as much (x: one 16-channel chunk per 256-thread group;
// dy: one unit per thread), and the nine tap accumulators are split over the SIMD's two consumer waves (taps 0-4 | 5-8: 80 accumulator
// registers, which is what fits four waves of 128 registers on a SIMD).  Measured against the eight-wave form: grouped launch 124 -> 113 us,
// step 14.84 -> 14.63 ms same-box.
template <int KS, int S, int MODE, bool DY2>
__global__ __launch_bounds__(1024, 4) void conv_wgrad_x3pc16_kernel(const wg_group grp_) {
    int mi = 0;
#pragma unroll
    for (int i = 1; i < CTL_WG_MAX; ++i)
        if (i < grp_.n && (int)blockIdx.x >= grp_.m[i].job0) mi = i;
    const wg_member& M = grp_.m[mi];
    const ctl_conv& d = M.d;
    const float* __restrict__ x = M.x; const float* __restrict__ pro_scale = M.pro_scale; const float* __restrict__ pro_shift = M.pro_shift;
    const float* __restrict__ dy = M.dy; const float* __restrict__ dy2 = M.dy2; const float* __restrict__ dy_coef = M.dy_coef;
    float* __restrict__ w_partial = M.w_partial; float* __restrict__ b_partial = M.b_partial;
    const int tiles_h = M.tiles_h, tiles_w = M.tiles_w, ntiles = M.ntiles, cin_p = M.cin_p, cout_p = M.cout_p;
    const int job = (int)blockIdx.x - M.job0, job_ns = M.splits;
    const int job_s = job % job_ns, job_r = job / job_ns;
    const int gb = job_r % (cin_p / 32), cb = job_r / (cin_p / 32);
    static_assert(S == 1 && KS == 3, "3x3 stride 1");
    constexpr int MT = 2, TW = 16;
    using G = Geom<KS, S, MT, TW>;
    using XS = XStage3<KS, S, MODE, MT, TW, false, false>;
    constexpr int TAPS = KS * KS, TH0 = 5;                   // taps 0 .. TH0-1 on the first consumer wave of a SIMD, the rest on the second
    constexpr int XCH = XS::XT_BYTES + 128, DCH = G::TP * 32 + 128;
    constexpr int XBYTES = 2 * XCH, DYI = 2 * DCH, IMG = XBYTES + 3 * DYI;
    constexpr int TAP_BYTES = 4 * 16 * 64 * 4;
    constexpr int RT = (2 * IMG) / TAP_BYTES < TAPS ? (2 * IMG) / TAP_BYTES : TAPS;
    constexpr int COEF = (DY2 ? 5 : 2) * CTL_PRO_MAX * 4;
    static_assert(2 * IMG + COEF + 512 * 8 * 4 <= 160 * 1024, "LDS budget of one CU");
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * IMG + COEF + 512 * 8 * 4];
    float* cf_scale = reinterpret_cast<float*>(smem + 2 * IMG);
    float* cf_shift = cf_scale + CTL_PRO_MAX;
    float* cd = cf_shift + CTL_PRO_MAX;
    float* bsred = reinterpret_cast<float*>(smem + 2 * IMG + COEF);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave_all = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool consumer = wave_all >= 8;
    const int group_n = d.n / (d.groups > 1 ? d.groups : 1);
    const int my_tiles = (job_s < ntiles) ? (ntiles - job_s + job_ns - 1) / job_ns : 0;
    if (d.pro_affine)
        for (int i = tid; i < (d.groups > 1 ? d.groups : 1) * d.cin; i += 1024) { cf_scale[i] = pro_scale[i]; cf_shift[i] = pro_shift[i]; }
    if constexpr (DY2) {
        for (int i = tid; i < (d.groups > 1 ? d.groups : 1) * d.cout; i += 1024) {
            const int gi = i / d.cout, ch = i - gi * d.cout;
            cd[i] = dy_coef[(gi * 3 + 0) * d.cout + ch]; cd[CTL_PRO_MAX + i] = dy_coef[(gi * 3 + 1) * d.cout + ch];
            cd[2 * CTL_PRO_MAX + i] = dy_coef[(gi * 3 + 2) * d.cout + ch];
        }
    }
    __syncthreads();

    if (!consumer) {
        // ================================================================ producers: group pg = tid >> 8 stages x chunk 2 gb + pg and dy units pg * 256 ...
        constexpr int R = 2;
        const int pg = tid >> 8, ltid = tid & 255;
        const __amdgpu_buffer_rsrc_t rx = ctl_rsrc(x, (int64_t)d.n * d.hin * d.win * d.cin * 4);
        const __amdgpu_buffer_rsrc_t rdy = ctl_rsrc(dy, (int64_t)d.n * d.hout * d.wout * d.cout * 4);
        const __amdgpu_buffer_rsrc_t rdy2 = DY2 ? ctl_rsrc(dy2, (int64_t)d.n * d.hout * d.wout * d.cout * 4) : rdy;
        const __amdgpu_buffer_rsrc_t rnone = ctl_rsrc((const void*)nullptr, 0);
        XS xs;
        xs.init(d, ltid);
        static_assert(G::TP * 4 == 512, "one dy unit (8 channels of a pixel) per producer thread");
        struct DyPay { f32x4 v0, v1, w0, w1; bool in; };
        const int drest = tid & 3, dt_ = drest >> 1, dh = drest & 1;
        const int dco = cb * 32 + dt_ * 16 + dh * 8;
        const int dpix = tid >> 2, dpr = dpix / TW, dpc = dpix % TW;
        const int drel = ((dpr * d.wout + dpc) * d.cout + dco) * 4;
        const int dlds = dt_ * DCH + dpix * 32 + dh * 16;
        f32x4 bs0 = {0.f, 0.f, 0.f, 0.f}, bs1 = bs0;
        auto dyload = [&](DyPay& P, int n, int ho0, int wo0, bool live) {
            const int tb = ((n * d.hout + ho0) * d.wout + wo0) * d.cout * 4;
            const bool ok = live && (unsigned)(ho0 + dpr) < (unsigned)d.hout && (unsigned)(wo0 + dpc) < (unsigned)d.wout;
            const int vo = ok ? (tb + drel) : CTL_OOB, vo1 = ok ? (tb + drel + 16) : CTL_OOB;
            P.v0 = ctl_bload4(rdy, vo); P.v1 = ctl_bload4(rdy, vo1);
            if constexpr (DY2) { P.w0 = ctl_bload4(rdy2, vo); P.w1 = ctl_bload4(rdy2, vo1); }
            P.in = ok;
        };
        auto dystore = [&](const DyPay& P, unsigned char* dyt, int goff) {
            f32x4 lo = P.v0, hi = P.v1;
            if constexpr (DY2) {      // pixels past the image contribute nothing (C alone would)
                const float* cc = cd + goff + dco;
                const f32x4 a0 = *reinterpret_cast<const f32x4*>(cc), a1 = *reinterpret_cast<const f32x4*>(cc + 4);
                const f32x4 b0 = *reinterpret_cast<const f32x4*>(cc + CTL_PRO_MAX), b1 = *reinterpret_cast<const f32x4*>(cc + CTL_PRO_MAX + 4);
                const f32x4 c0 = *reinterpret_cast<const f32x4*>(cc + 2 * CTL_PRO_MAX), c1 = *reinterpret_cast<const f32x4*>(cc + 2 * CTL_PRO_MAX + 4);
                const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
                lo = P.in ? (a0 * lo + b0 * P.w0 + c0) : zero;
                hi = P.in ? (a1 * hi + b1 * P.w1 + c1) : zero;
            }
            bs0 += lo; bs1 += hi;
            u32x4 ph, pm, pl;
            x3_split8(lo, hi, ph, pm, pl);
            *reinterpret_cast<u32x4*>(dyt + dlds) = ph;
            *reinterpret_cast<u32x4*>(dyt + dlds + DYI) = pm;
            *reinterpret_cast<u32x4*>(dyt + dlds + 2 * DYI) = pl;
        };
        struct Set { typename XS::Pay xa; DyPay dyp; };
        Set P[R];
        int s_n[R];
        TileWalk lw;
        lw.init(job_s, job_ns, tiles_h, tiles_w);
        int lit = 0;
        auto load_step = [&](Set& st, int& sn) {
            const bool live = lit < my_tiles;
            xs.template load<true>(st.xa, rx, rx, d, lw.n, lw.th * G::TH, lw.tw * TW, 2 * gb + pg, live);
            dyload(st.dyp, lw.n, lw.th * G::TH, lw.tw * TW, live);
            sn = lw.n;
            ++lit;
            lw.next();
        };
#pragma unroll
        for (int r = 0; r < R; ++r) load_step(P[r], s_n[r]);
        auto step = [&](int it, Set& st, int& sn, bool more) {
            unsigned char* img = smem + (it & 1) * IMG;
            const int grp = sn / group_n;
            xs.template store<true>(st.xa, reinterpret_cast<float*>(img + pg * XCH), d, 2 * gb + pg, cf_scale, cf_shift, grp * d.cin, nullptr, rnone, false);
            dystore(st.dyp, img + XBYTES, grp * d.cout);
            if (more) load_step(st, sn);
            ctl_barrier_lds_writes_done();
        };
        const int full = (my_tiles / R) * R;
        for (int it0 = 0; it0 < full; it0 += R) {
#pragma unroll
            for (int r = 0; r < R; ++r) step(it0 + r, P[r], s_n[r], true);
        }
#pragma unroll
        for (int r = 0; r < R - 1; ++r)
            if (full + r < my_tiles) step(full + r, P[r], s_n[r], false);
        *reinterpret_cast<f32x4*>(bsred + tid * 8) = bs0;
        *reinterpret_cast<f32x4*>(bsred + tid * 8 + 4) = bs1;
        ctl_barrier_lds_writes_done();                       // E1
        if (gb == 0 && b_partial != nullptr && tid < 32) {
            const int t = tid >> 4, c = tid & 15, h = c >> 3, j = c & 7;
            float v = 0.f;
            for (int sl = t * 2 + h; sl < 512; sl += 4) v += bsred[sl * 8 + j];
            const int co = cb * 32 + tid;
            if (co < cout_p) b_partial[(int64_t)job_s * cout_p + co] = v;
        }
#pragma unroll
        for (int t0 = 0; t0 < TAPS; t0 += RT) {
            ctl_barrier_lds_reads_done();
            if (t0 + RT < TAPS) ctl_barrier_lds_reads_done();
        }
        return;
    }
    // ================================================================ consumers: wave cw = (rows 2 (cw & 3), 2 (cw & 3) + 1; tap half cw >> 2)
    const int cw = wave_all - 8, wave = cw & 3, half = cw >> 2;
    const int q = lane >> 4, chunk = q & 1, phalf = q >> 1;
    const int qp = (lane & 15) >> 2, pp = lane & 3;
    const int a_off = chunk * XCH + (8 * phalf + qp) * 32 + pp * 8;
    const int b_off = XBYTES + chunk * DCH + (8 * phalf + qp) * 32 + pp * 8;
    float* red = reinterpret_cast<float*>(smem);
    const int ctid = tid - 512;
    const int64_t split_base = (int64_t)job_s * TAPS * cin_p * cout_p;
    auto consume = [&](auto half_tag) {
        constexpr int T0 = decltype(half_tag)::value ? TH0 : 0, NTAP = decltype(half_tag)::value ? TAPS - TH0 : TH0;
        f32x16 acc[NTAP];
#pragma unroll
        for (int a = 0; a < NTAP; ++a)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
        for (int it = 0; it < my_tiles; ++it) {
            ctl_barrier_lds_reads_done();
            const unsigned char* img = smem + (it & 1) * IMG;
#pragma unroll
            for (int rr = 0; rr < 2; ++rr) {
                const int tr = 2 * wave + rr;
                const unsigned char* arow = img + a_off + tr * (G::IWP * 32);
                const unsigned char* brow = img + b_off + tr * (TW * 32);
                x3_bf16x8 bf[3];
#pragma unroll
                for (int sp = 0; sp < 3; ++sp) bf[sp] = x3_tr_read8(brow + sp * DYI, brow + sp * DYI + 4 * 32);
                auto a_operand = [&](int tap, x3_bf16x8* af) {
                    const int kh = tap / KS, kw = tap % KS;
                    const unsigned char* a0 = arow + (kh * G::IWP + kw) * 32;
#pragma unroll
                    for (int sp = 0; sp < 3; ++sp) af[sp] = x3_tr_read8(a0 + sp * XS::SPLIT, a0 + sp * XS::SPLIT + 4 * 32);
                };
                x3_bf16x8 af[2][3];
                a_operand(T0, af[0]);
#pragma unroll
                for (int t = 0; t < NTAP; ++t) {
                    if (t + 1 < NTAP) a_operand(T0 + t + 1, af[(t + 1) & 1]);
                    x3_bf16x8* a = af[t & 1];
                    asm volatile("" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]) : : "memory");
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], bf[0], acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], bf[2], acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], bf[1], acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], bf[0], acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], bf[1], acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], bf[0], acc[t], 0, 0, 0);
                }
            }
        }
        ctl_barrier_lds_reads_done();                        // E1
        // ---- the sums of the four row waves of every tap through LDS, RT taps per round (each tap is owned by one tap half)
#pragma unroll
        for (int t0 = 0; t0 < TAPS; t0 += RT) {
#pragma unroll
            for (int tp = 0; tp < RT; ++tp) {
                const int tap = t0 + tp;
                if (tap < TAPS && tap >= T0 && tap < T0 + NTAP) {
                    float* r0 = red + (tp * 4 + wave) * 16 * 64 + lane;
#pragma unroll
                    for (int r = 0; r < 16; ++r) r0[r * 64] = acc[(tap - T0) < NTAP ? (tap - T0) : 0][r];
                }
            }
            ctl_barrier_lds_writes_done();
#pragma unroll
            for (int tp = 0; tp < RT; ++tp) {
                const int tap = t0 + tp;
                if (tap < TAPS) {
#pragma unroll
                    for (int e0 = 0; e0 < 16 * 64; e0 += 512) {
                        const int e = e0 + ctid, r = e >> 6, l = e & 63;
                        float v = 0.f;
#pragma unroll
                        for (int w = 0; w < 4; ++w) v += red[((tp * 4 + w) * 16 + r) * 64 + l];
                        const int co = cb * 32 + (l & 31);
                        const int ci = gb * 32 + 8 * (r >> 2) + 4 * (l >> 5) + (r & 3);
                        if (co < cout_p && ci < cin_p) w_partial[split_base + ((int64_t)tap * cin_p + ci) * cout_p + co] = v;
                    }
                }
            }
            if (t0 + RT < TAPS) ctl_barrier_lds_reads_done();
        }
    };
    if (half) consume(std::true_type{}); else consume(std::false_type{});
}

// ------------------------------------------------------------------------------------------------ host side
struct wgrad3_call {
    const ctl_conv* d; ctl_conv_cfg c; int ntw, splits, ntiles, cin_p, cout_p;
    const float *x, *dy, *dy2, *pro_scale, *pro_shift, *dy_coef; float *w_partial, *b_partial;
    hipStream_t stream; bool query;
    int pc;      // the producer / consumer kernel (32 x 32 blocks) was chosen
};
template <int KS, int S, int MODE, int MT, int NTW, bool DY2>
static void wgrad3_go_f(wgrad3_call& a) {
    static int occ = 0;
    if (!occ) {
        int n = 0;
        // (of the plain instantiation, also for DY2: the split count is queried at plan time from the descriptor alone and sizes the partials)
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, conv_wgrad_x3_kernel<KS, S, MODE, MT, NTW, false>, 256, 0) != hipSuccess || n < 1) {
            (void)hipGetLastError();
            n = 1;
        }
        occ = n;
    }
    const int par = a.c.g * (a.c.cot / NTW);
    static const int per_cu = ctl_tune_int("CTL_X3W_PERSIST", 2);      // tuning hook: resident blocks per CU
    int splits = (ctl_num_cus() * (occ < per_cu ? occ : per_cu)) / par;
    if (splits > 512) splits = 512;
    if (splits > a.ntiles) splits = a.ntiles;
    if (splits < 1) splits = 1;
    a.splits = splits;
    if (a.query) return;
    const dim3 grid((unsigned)splits, (unsigned)a.c.g, (unsigned)(a.c.cot / NTW));
    conv_wgrad_x3_kernel<KS, S, MODE, MT, NTW, DY2><<<grid, dim3(256), 0, a.stream>>>(*a.d, a.x, a.pro_scale, a.pro_shift, a.dy, a.dy2, a.dy_coef, a.w_partial,
                                                                                    a.b_partial, a.c.tiles_h, a.c.tiles_w, a.ntiles, a.cin_p, a.cout_p);
}
template <int KS, int S, int MODE, int MT, int NTW>
static void wgrad3_go(wgrad3_call& a) {
    if constexpr (KS == 3 && S == 1) {
        if (a.dy2) { wgrad3_go_f<KS, S, MODE, MT, NTW, true>(a); return; }
    }
    wgrad3_go_f<KS, S, MODE, MT, NTW, false>(a);
}
template <int KS, int S, int MODE>
static void wgrad3_go_tile(wgrad3_call& a) {
    if (a.c.mt == 2) { if (a.ntw == 2) wgrad3_go<KS, S, MODE, 2, 2>(a); else wgrad3_go<KS, S, MODE, 2, 1>(a); }
    else { if (a.ntw == 2) wgrad3_go<KS, S, MODE, 1, 2>(a); else wgrad3_go<KS, S, MODE, 1, 1>(a); }
}
#ifndef CTL_X3W_PC_DEFAULT
#define CTL_X3W_PC_DEFAULT 0      // a SINGLE problem on the producer / consumer kernel: per layer equal to the single-role kernel, in the step 0.8 % slower
#endif                            // (one 130 KB block per CU next to the other launch chain); the kernel is what the GROUPED launches run on
#ifndef CTL_X3W_PC_MIN_TILES
#define CTL_X3W_PC_MIN_TILES 4
#endif
static void wgrad3_fill_member(wg_member& m, const wgrad3_call& a, int job0) {
    m.d = *a.d;
    m.x = a.x; m.pro_scale = a.pro_scale; m.pro_shift = a.pro_shift; m.dy = a.dy; m.dy2 = a.dy2; m.dy_coef = a.dy_coef;
    m.w_partial = a.w_partial; m.b_partial = a.b_partial;
    m.tiles_h = a.c.tiles_h; m.tiles_w = a.c.tiles_w; m.ntiles = a.ntiles; m.cin_p = a.cin_p; m.cout_p = a.cout_p;
    m.splits = a.splits; m.job0 = job0; m.pad_ = 0;
}
static void wgrad3_launch_group(const wg_group& g, int jobs, int mode, bool dy2, hipStream_t stream) {
    const dim3 grid((unsigned)jobs);
    if (mode == CTL_IN_PLAIN) {
        if (dy2) conv_wgrad_x3pc16_kernel<3, 1, CTL_IN_PLAIN, true><<<grid, dim3(1024), 0, stream>>>(g);
        else conv_wgrad_x3pc16_kernel<3, 1, CTL_IN_PLAIN, false><<<grid, dim3(1024), 0, stream>>>(g);
    } else {
        if (dy2) conv_wgrad_x3pc16_kernel<3, 1, CTL_IN_UP2, true><<<grid, dim3(1024), 0, stream>>>(g);
        else conv_wgrad_x3pc16_kernel<3, 1, CTL_IN_UP2, false><<<grid, dim3(1024), 0, stream>>>(g);
    }
}
template <int MODE>
static void wgrad3_go_pc(wgrad3_call& a) {      // a single problem = a group of one
    wg_group g = {};
    g.n = 1;
    wgrad3_fill_member(g.m[0], a, 0);
    wgrad3_launch_group(g, a.splits * (a.cin_p / 32) * (a.cout_p / 32), MODE, a.dy2 != nullptr, a.stream);
}
// one block per CU, (cin / 32) x (cout / 32) output blocks: the pixel tiles are split over what is left of the 256 CUs
static bool wgrad3_pc_shape_ok(const ctl_conv* d) {
    return ctl_wgrad_x3_ok(d) && d->ks == 3 && d->stride == 1 && (d->in_mode == CTL_IN_PLAIN || d->in_mode == CTL_IN_UP2) && d->cin % 32 == 0 && d->cout % 32 == 0 &&
           d->hout >= 8;
}
static bool wgrad3_pc_pick(const ctl_conv* d, wgrad3_call* a) {
    static const int pc_on = ctl_tune_int("CTL_X3W_PC", CTL_X3W_PC_DEFAULT), min_tiles = ctl_tune_int("CTL_X3W_PC_MIN_TILES", CTL_X3W_PC_MIN_TILES);
    if (!pc_on || !wgrad3_pc_shape_ok(d) || a->c.mt != 2) return false;
    const int par = (d->cin / 32) * (d->cout / 32);
    int splits = ctl_num_cus() / par;
    if (splits > a->ntiles) splits = a->ntiles;
    if (splits < 1) splits = 1;
    if (a->ntiles < (int64_t)min_tiles * splits) return false;
    a->splits = splits;
    return true;
}
static int wgrad3_dispatch(wgrad3_call& a) {
    const int k = a.d->ks, s = a.d->stride, m = a.d->in_mode;
    if (a.pc) {
        if (a.query) return CTL_OK;
        if (m == CTL_IN_PLAIN) wgrad3_go_pc<CTL_IN_PLAIN>(a); else wgrad3_go_pc<CTL_IN_UP2>(a);
        return CTL_OK;
    }
    if (k == 3 && s == 1 && m == CTL_IN_PLAIN) wgrad3_go_tile<3, 1, CTL_IN_PLAIN>(a);
    else if (k == 3 && s == 1 && m == CTL_IN_UP2) wgrad3_go_tile<3, 1, CTL_IN_UP2>(a);
    else if (k == 3 && s == 2 && m == CTL_IN_PLAIN) wgrad3_go_tile<3, 2, CTL_IN_PLAIN>(a);
    else if (k == 2 && s == 2 && m == CTL_IN_PLAIN) wgrad3_go_tile<2, 2, CTL_IN_PLAIN>(a);
    else CTL_FAIL(CTL_EUNSUPPORTED, "conv_wgrad(x3): no kernel for ks/stride/in_mode %d/%d/%d", k, s, m);
    return CTL_OK;
}
int ctl_wgrad_x3_ok(const ctl_conv* d) {
    const int k = d->ks, s = d->stride, m = d->in_mode;
    const bool combo = (k == 3 && s == 1 && (m == CTL_IN_PLAIN || m == CTL_IN_UP2)) || (k == 3 && s == 2 && m == CTL_IN_PLAIN) || (k == 2 && s == 2 && m == CTL_IN_PLAIN);
    return combo && d->cin % 16 == 0 && d->cout % 16 == 0 && d->nsub == 1 && !(d->dt & (CTL_DT_BF16 | CTL_DT_X16 | CTL_DT_Y16 | CTL_DT_RES16));
}
static int wgrad3_pick(const ctl_conv* d, wgrad3_call* a) {
    CTL_REQUIRE(ctl_wgrad_x3_ok(d), "wgrad(x3): CTL_DT_X3 needs fp32-stored tensors, cin %% 16 == 0, cout %% 16 == 0 and a 3x3 (stride 1 / 2) or 2x2 stride-2 "
                                    "kernel (got cin %d, cout %d, ks %d, stride %d, in_mode %d, dt %d)", d->cin, d->cout, d->ks, d->stride, d->in_mode, d->dt);
    a->d = d;
    int rc = ctl_conv_pick_cfg(d, &a->c, 1);
    if (rc != CTL_OK) return rc;
    // (16x16-pixel tiles for the full-resolution 16-channel layers, the bf16 family's choice, measured slower here at two resident blocks per
    //  CU: 16->16 @256^2 45.0 vs 47.6 us, up-sampled 16->16 @128^2 42.4 vs 44.7 us, tools/sweep_wgrad_x3.sh -- the 8x16 tile stays)
    a->ntw = (a->c.cot >= 2 && a->c.cot % 2 == 0) ? 2 : 1;
    // stride-2 3x3 with two cout tiles: the 17x33-pixel input tile of an 8x16 output tile costs 238 + 126 registers -> ONE resident block per
    // CU; the 4x16 tile (194 registers, two blocks) is 7-15 % faster (tools/sweep_wgrad_x3_s2.sh: 16->32 @256^2 35.5 -> 33.2 us,
    // 32->64 @128^2 37.6 -> 32.1, 64->128 @64^2 35.0 -> 31.5); with one cout tile the 8x16 tile fits twice and stays (25.6 vs 27.6 us)
    static const int s2_mt = ctl_tune_int("CTL_X3W_S2_MT", 0), s2_ntw = ctl_tune_int("CTL_X3W_S2_NTW", 0);      // tuning hooks
    if (d->ks == 3 && d->stride == 2) {
        if (s2_ntw) a->ntw = s2_ntw;
        const int mt = s2_mt ? s2_mt : (a->ntw == 2 ? 1 : a->c.mt);
        if (mt != a->c.mt) { a->c.mt = mt; a->c.th = 4 * mt; a->c.tiles_h = ctl_cdiv(d->hout, a->c.th); }
    }
    a->ntiles = d->n * a->c.tiles_h * a->c.tiles_w;
    a->cin_p = a->c.g * 16;
    a->cout_p = a->c.cot * 16;
    a->pc = wgrad3_pc_pick(d, a) ? 1 : 0;
    a->query = true;
    rc = wgrad3_dispatch(*a);
    a->query = false;
    return rc;
}
int ctl_wgrad_x3_splits(const ctl_conv* d) {
    wgrad3_call a = {};
    return wgrad3_pick(d, &a) == CTL_OK ? a.splits : -1;
}
int ctl_conv_wgrad_x3(const ctl_conv* d, const float* x, const float* pro_scale, const float* pro_shift, const float* dy, const float* dy2,
                      const float* dy_coef, float* w_partial, float* b_partial, ctl_stream stream) {
    wgrad3_call a = {};
    int rc = wgrad3_pick(d, &a);
    if (rc != CTL_OK) return rc;
    a.x = x; a.dy = dy; a.dy2 = dy2; a.dy_coef = dy_coef; a.pro_scale = pro_scale; a.pro_shift = pro_shift; a.w_partial = w_partial; a.b_partial = b_partial;
    a.stream = (hipStream_t)stream;
    const int ptok = ctl_prof_begin(a.pc ? "conv_wgrad_x3pc" : "conv_wgrad_x3", d, &a.c, a.ntw, a.stream, dy2 != nullptr);
    rc = wgrad3_dispatch(a);
    if (ptok >= 0) ctl_prof_end(ptok, a.stream);
    if (rc != CTL_OK) return rc;
    CTL_LAUNCH_CHECK("conv_wgrad(x3)");
    return CTL_OK;
}

// ------------------------------------------------------------------------------------------------ grouped launches (deferred weight gradients)
// ctl_wgrad_group_class: >= 0 if the weight gradient of this descriptor can ride in a grouped launch (members of one launch share the class =
// kernel instantiation: input mode | two-tensor output gradient << 1), -1 otherwise.
extern "C" int ctl_wgrad_group_class(const ctl_conv* d, int32_t has_dy2) {
    static const int on = ctl_tune_int("CTL_X3W_GROUP", 1);
    if (d && (d->dt & CTL_DT_BF16)) return ctl_wgrad_bf16_group_class(d, has_dy2);      // the bf16 family: stacked launches, class 0x100 | instantiation
    if (!on || !d || !(d->dt & CTL_DT_X3) || !wgrad3_pc_shape_ok(d)) return -1;
    return (d->in_mode == CTL_IN_UP2 ? 1 : 0) | (has_dy2 ? 2 : 0);
}
// ctl_wgrad_group_plan: pixel splits of the n members of one launch: the CUs are dealt in proportion to the members' work (pixel tiles x
// 32 x 32 output blocks), every member gets at least one split, no split is empty.  The plan compiler sizes the partial buffers from this.
extern "C" int ctl_wgrad_group_plan(const ctl_conv* descs, int32_t n, int32_t* splits) {
    CTL_REQUIRE(descs && splits && n >= 1 && n <= CTL_WG_MAX, "wgrad_group_plan: 1..%d members", CTL_WG_MAX);
    if (descs[0].dt & CTL_DT_BF16) return ctl_wgrad_bf16_group_plan(descs, n, splits);
    int64_t work[CTL_WG_MAX], total = 0;
    int par[CTL_WG_MAX], ntiles[CTL_WG_MAX];
    for (int i = 0; i < n; ++i) {
        CTL_REQUIRE(wgrad3_pc_shape_ok(&descs[i]), "wgrad_group_plan: member %d is not a grouped-launch shape", i);
        par[i] = (descs[i].cin / 32) * (descs[i].cout / 32);
        ntiles[i] = descs[i].n * ctl_cdiv(descs[i].hout, 8) * ctl_cdiv(descs[i].wout, 16);
        work[i] = (int64_t)ntiles[i] * par[i];
        total += work[i];
    }
    const int cus = ctl_num_cus();
    for (int i = 0; i < n; ++i) {
        int sp = (int)((cus * work[i]) / (total * par[i]));          // floor: the sum of jobs stays <= the CU count unless a member needs its minimum
        if (sp < 1) sp = 1;
        if (sp > ntiles[i]) sp = ntiles[i];
        splits[i] = sp;
    }
    return CTL_OK;
}
extern "C" int ctl_conv_wgrad_group(int32_t n, const ctl_conv* descs, const int32_t* splits, const float* const* x, const float* const* pro_scale,
                                    const float* const* pro_shift, const float* const* dy, const float* const* dy2, const float* const* dy_coef,
                                    float* const* w_partial, float* const* b_partial, ctl_stream stream) {
    CTL_REQUIRE(n >= 1 && n <= CTL_WG_MAX && descs && splits && x && dy && w_partial, "conv_wgrad_group: 1..%d members and their tensors", CTL_WG_MAX);
    if (descs[0].dt & CTL_DT_BF16)
        return ctl_conv_wgrad_bf16_group(n, descs, splits, (const void* const*)x, pro_scale, pro_shift, (const void* const*)dy, (const void* const*)dy2, dy_coef,
                                         w_partial, b_partial, stream);
    wg_group g = {};
    g.n = n;
    int jobs = 0;
    const int cls = ctl_wgrad_group_class(&descs[0], dy2 && dy2[0]);
    CTL_REQUIRE(cls >= 0, "conv_wgrad_group: member 0 is not a grouped-launch shape");
    double flops = 0.0, bytes = 0.0;
    for (int i = 0; i < n; ++i) {
        const ctl_conv* d = &descs[i];
        CTL_REQUIRE(ctl_wgrad_group_class(d, dy2 && dy2[i]) == cls, "conv_wgrad_group: member %d is of another class than member 0", i);
        CTL_REQUIRE(x[i] && dy[i] && w_partial[i] && (!d->pro_affine || (pro_scale && pro_shift && pro_scale[i] && pro_shift[i])) && (!(cls & 2) || (dy_coef && dy_coef[i])),
                    "conv_wgrad_group: member %d misses a tensor", i);
        // the coefficient tables a member actually stages (as in ctl_conv_wgrad_ex): the prologue's over groups * cin, the virtual output gradient's over groups * cout
        {
            const int ng = d->groups > 1 ? d->groups : 1;
            CTL_REQUIRE((!d->pro_affine || ng * d->cin <= CTL_PRO_MAX) && (!(cls & 2) || ng * d->cout <= CTL_PRO_MAX),
                        "conv_wgrad_group: member %d: groups * channels of a staged coefficient table > %d", i, CTL_PRO_MAX);
        }
        CTL_REQUIRE((int64_t)d->n * d->hin * d->win * d->cin * 4 < (1ll << 31) && (int64_t)d->n * d->hout * d->wout * d->cout * 4 < (1ll << 31),
                    "conv_wgrad_group: tensors must stay below 2 GiB (32-bit buffer offsets)");
        wgrad3_call a = {};
        a.d = d;
        int rc = ctl_conv_pick_cfg(d, &a.c, 1);
        if (rc != CTL_OK) return rc;
        a.ntiles = d->n * a.c.tiles_h * a.c.tiles_w;
        a.cin_p = a.c.g * 16; a.cout_p = a.c.cot * 16;
        CTL_REQUIRE(splits[i] >= 1 && splits[i] <= a.ntiles && a.c.mt == 2, "conv_wgrad_group: member %d: splits %d of %d tiles", i, splits[i], a.ntiles);
        a.splits = splits[i];
        a.x = x[i]; a.pro_scale = pro_scale ? pro_scale[i] : nullptr; a.pro_shift = pro_shift ? pro_shift[i] : nullptr; a.dy = dy[i];
        a.dy2 = dy2 ? dy2[i] : nullptr; a.dy_coef = dy_coef ? dy_coef[i] : nullptr; a.w_partial = w_partial[i]; a.b_partial = b_partial ? b_partial[i] : nullptr;
        wgrad3_fill_member(g.m[i], a, jobs);
        jobs += a.splits * (a.cin_p / 32) * (a.cout_p / 32);
        const double pix = (double)d->n * d->hout * d->wout;
        flops += 2.0 * pix * d->cout * d->cin * 9;
        bytes += 4.0 * d->n * d->hin * d->win * d->cin + 4.0 * pix * d->cout * ((cls & 2) ? 2 : 1);
    }
    char kind[48];
    snprintf(kind, sizeof(kind), "conv_wgrad_x3grp<in%d%s>", cls & 1, (cls & 2) ? ",x2" : "");
    const int ptok = ctl_prof_begin_raw(kind, flops, bytes, (hipStream_t)stream);
    wgrad3_launch_group(g, jobs, (cls & 1) ? CTL_IN_UP2 : CTL_IN_PLAIN, (cls & 2) != 0, (hipStream_t)stream);
    if (ptok >= 0) ctl_prof_end(ptok, (hipStream_t)stream);
    CTL_LAUNCH_CHECK("conv_wgrad_group");
    return CTL_OK;
}
