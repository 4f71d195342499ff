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
// conv_wgrad_x3pc_kernel: 32 cin x 32 cout per block on v_mfma_f32_32x32x16_bf16, ONE 512-thread block per CU.
//   * the 32x32x16 form issues HALF the matrix instructions of the 16x16x32 form for the same products (an MFMA holds the SIMD's vector
//     issue port for 8 cycles whatever its shape: 8 of 32 instead of 8 of 16), and a 32 x 32 output block stages dy once per 32 input
//     channels instead of once per 16 (and x once per 32 output channels): what bounds the single-role kernel above is the instruction
//     issue of a SIMD -- matrix + split / prologue vector instructions + LDS traffic -- not the matrix pipe;
//   * waves 0-3 (producers) keep CTL_X3W_PC_SETS tiles of loads in flight in registers, evaluate the prologue / the virtual output
//     gradient, split and write the three bf16 images of tile `it` into LDS image it & 1; waves 4-7 (consumers) hold the 9 x 32 x 32
//     accumulators (144 registers) and do nothing but transposed operand reads and MFMAs: one s_barrier per tile (see conv_igemm_kernel,
//     PC, for the protocol and for why every load in the producer loop is unconditional and single-path).
//   D[ci][co] += A[ci][k = pixel] B[pixel][co], K = 16 pixels = one tile row; consumer wave w owns tile rows 2w, 2w + 1 and all nine taps;
//   the four waves' sums meet in LDS at the end of the block (two rounds of taps through the then free images).
// Lane maps (v_mfma_f32_32x32x16_bf16): A[row = lane & 31][k = 8 (lane >> 5) + j], B[k = 8 (lane >> 5) + j][col = lane & 31], D[row = (r & 3)
// + 8 (r >> 2) + 4 (lane >> 5)][col = lane & 31]: lane group q = lane >> 4 reads the 16-channel chunk q & 1 and the pixel half q >> 1 of
// its operand with the transposed reads of the kernel above.
#ifndef CTL_X3W_PC_SETS
#define CTL_X3W_PC_SETS 3
#endif
#ifndef CTL_X3W_PC_SETS_DY2
#define CTL_X3W_PC_SETS_DY2 2
#endif
typedef float f32x16 __attribute__((ext_vector_type(16)));
// compile-time loop: f(std::integral_constant<int, 0>{}) ... f(std::integral_constant<int, N - 1>{}) in order
template <class F, int... I>
__device__ __forceinline__ void ctl_static_for_impl(F&& f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void ctl_static_for(F&& f) { ctl_static_for_impl(f, std::make_integer_sequence<int, N>{}); }
#ifndef CTL_X3WPC_CPRIO      // wave priorities of the two roles (measured: no effect either way, profiles/r5_wgrad_pc_experiments.txt)
#define CTL_X3WPC_CPRIO 0
#endif
#ifndef CTL_X3WPC_PPRIO
#define CTL_X3WPC_PPRIO 0
#endif
// Phase timers (variant builds only: tools/build_variant.sh tmw "-DCTL_TIMING_X3W" ctl_wgrad_x3.hip; read and reset with ctl_debug_timing_x3w):
// s_memtime deltas summed over waves: [0] producer staging (load wait + prologue + split + LDS writes), [1] producer load issue, [2] producer
// barrier wait, [3] consumer barrier wait, [4] consumer matrix phase, [5] tiles (consumer waves), [6] consumer head (to the first barrier's
// release), [7] consumer tail (reduction), [8] producer head (first loads issued), [9] block span (consumer waves)
#ifdef CTL_TIMING_X3W
__device__ unsigned long long ctl_tmw[16];
#define TW_DECL unsigned long long tw_prev = __builtin_amdgcn_s_memtime(), tw_acc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; const unsigned long long tw_t0 = tw_prev;
#define TW(i) { const unsigned long long tw_now = __builtin_amdgcn_s_memtime(); tw_acc[i] += tw_now - tw_prev; tw_prev = tw_now; }
#define TW_COUNT(i) { tw_acc[i] += 1; }
#define TW_FLUSH(span) { if (span) tw_acc[9] = __builtin_amdgcn_s_memtime() - tw_t0; \
                         if ((threadIdx.x & 63) == 0) { _Pragma("unroll") for (int i_ = 0; i_ < 10; ++i_) if (tw_acc[i_]) atomicAdd(&ctl_tmw[i_], tw_acc[i_]); } }
extern "C" int ctl_debug_timing_x3w(unsigned long long* out10) {
    unsigned long long host[16] = {0};
    if (hipMemcpyFromSymbol(host, HIP_SYMBOL(ctl_tmw), sizeof(host)) != hipSuccess) return -1;
    for (int i = 0; i < 10; ++i) out10[i] = host[i];
    for (int i = 0; i < 16; ++i) host[i] = 0;
    return hipMemcpyToSymbol(HIP_SYMBOL(ctl_tmw), host, sizeof(host)) == hipSuccess ? 0 : -1;
}
#else
#define TW_DECL
#define TW(i)
#define TW_COUNT(i)
#define TW_FLUSH(span)
#endif
template <int KS, int S, int MODE, bool DY2>
__global__ __launch_bounds__(512, 2) void conv_wgrad_x3pc_kernel(const ctl_conv d, const float* __restrict__ x, const float* __restrict__ pro_scale,
                                                                 const float* __restrict__ pro_shift, const float* __restrict__ dy,
                                                                 const float* __restrict__ dy2, const float* __restrict__ dy_coef,
                                                                 float* __restrict__ w_partial, float* __restrict__ b_partial, int tiles_h, int tiles_w,
                                                                 int ntiles, int cin_p, int cout_p) {
    static_assert(S == 1, "stride 1 (the transposed reads take 8 consecutive pixels of a row)");
    constexpr int MT = 2, TW = 16;
    using G = Geom<KS, S, MT, TW>;
    using XS = XStage3<KS, S, MODE, MT, TW, false, false>;
    constexpr int TAPS = KS * KS;
    // Bank phase: a transposed read is served in two halves of 32 lanes; lanes 0-15 (chunk 0) and 16-31 (chunk 1) each cover 128 contiguous
    // bytes = 32 of the 64 banks, so the second chunk of every operand sits 128 B (mod 256) behind the first: the two spans take the
    // two halves of the bank row (chunk strides that are multiples of 256 B: SQ_LDS_BANK_CONFLICT 0.37-0.45 of the LDS-active cycles in
    // the single-role kernel, where the two lane groups read pixels 0-7 / 8-15 of one image, 256 B apart)
    constexpr int XCH = XS::XT_BYTES + 128;                // byte distance of the two 16-channel chunks of x (XT_BYTES is a multiple of 256)
    constexpr int DCH = G::TP * 32 + 128;                  // ... and of the two 16-channel chunks of a dy split image
    static_assert(XS::XT_BYTES % 256 == 0 && (G::TP * 32) % 256 == 0, "bank phase of the chunks");
    constexpr int XBYTES = 2 * XCH;                        // two 16-channel chunks of x, three split images each
    constexpr int DYI = 2 * DCH;                           // one split image of the dy tile: [cout chunk][pixel][16 ch] bf16
    constexpr int IMG = XBYTES + 3 * DYI;                  // everything one tile needs
    constexpr int TAP_BYTES = 4 * 16 * 64 * 4;             // reduction scratch of one tap: [wave][register][lane] floats
    constexpr int RT = (2 * IMG) / TAP_BYTES < TAPS ? (2 * IMG) / TAP_BYTES : TAPS;      // taps per reduction round
    static_assert(RT >= 1, "reduction scratch");
    constexpr int COEF = (DY2 ? 5 : 2) * CTL_PRO_MAX * 4;
    static_assert(2 * IMG + COEF + 256 * 8 * 4 <= 160 * 1024, "LDS budget of one CU");
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * IMG + COEF + 256 * 8 * 4];
    float* cf_scale = reinterpret_cast<float*>(smem + 2 * IMG);
    float* cf_shift = cf_scale + CTL_PRO_MAX;
    float* cd = cf_shift + CTL_PRO_MAX;                    // DY2: A | B | C, [group][cout] each
    float* bsred = reinterpret_cast<float*>(smem + 2 * IMG + COEF);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave_all = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool consumer = wave_all >= 4;
    const int wave = wave_all & 3;
    const int gb = blockIdx.y, cb = blockIdx.z;            // 32-channel blocks of cin / cout
    const int group_n = d.n / (d.groups > 1 ? d.groups : 1);
    // tiles of this block: blockIdx.x, + gridDim.x, ...
    const int my_tiles = ((int)blockIdx.x < ntiles) ? (ntiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x : 0;

    if (tid < 256 || true) {                               // coefficient tables (every thread helps; one barrier for all)
        if (d.pro_affine)
            for (int i = tid; i < (d.groups > 1 ? d.groups : 1) * d.cin; i += 512) { cf_scale[i] = pro_scale[i]; cf_shift[i] = pro_shift[i]; }
        if constexpr (DY2) {
            for (int i = tid; i < (d.groups > 1 ? d.groups : 1) * d.cout; i += 512) {
                const int gi = i / d.cout, ch = i - gi * d.cout;
                cd[i] = dy_coef[(gi * 3 + 0) * d.cout + ch]; cd[CTL_PRO_MAX + i] = dy_coef[(gi * 3 + 1) * d.cout + ch];
                cd[2 * CTL_PRO_MAX + i] = dy_coef[(gi * 3 + 2) * d.cout + ch];
            }
        }
    }
    __syncthreads();
    TW_DECL

    if (!consumer) {
        // ================================================================ producers
        // (priority: the staging is a long chain of short vector instructions with one wave to hide its own latencies, the matrix wave
        //  needs one issue slot in 32 cycles: whenever the producer can issue it should, the MFMAs fill what is left)
        __builtin_amdgcn_s_setprio(CTL_X3WPC_PPRIO);
        constexpr int R = DY2 ? CTL_X3W_PC_SETS_DY2 : CTL_X3W_PC_SETS;
        const __amdgpu_buffer_rsrc_t rx = ctl_rsrc(x, (int64_t)d.n * d.hin * d.win * d.cin * 4);
        const __amdgpu_buffer_rsrc_t rdy = ctl_rsrc(dy, (int64_t)d.n * d.hout * d.wout * d.cout * 4);
        const __amdgpu_buffer_rsrc_t rdy2 = DY2 ? ctl_rsrc(dy2, (int64_t)d.n * d.hout * d.wout * d.cout * 4) : rdy;
        const __amdgpu_buffer_rsrc_t rnone = ctl_rsrc((const void*)nullptr, 0);
        XS xs;
        xs.init(d);
        // dy tile: units of 8 channels (two 16-byte fp32 loads); a thread's units share (cout chunk, half): 256 % 4 == 0
        constexpr int DU = G::TP * 4, ND = DU / 256;
        static_assert(DU % 256 == 0, "dy units");
        struct DyPay { f32x4 v0[ND], v1[ND], w0[DY2 ? ND : 1], w1[DY2 ? ND : 1]; unsigned mask; };
        int drel[ND], drc[ND], dlds[ND];
        const int drest = tid & 3, dt_ = drest >> 1, dh = drest & 1;
        const int dco = cb * 32 + dt_ * 16 + dh * 8;        // first channel of this thread's units
#pragma unroll
        for (int i = 0; i < ND; ++i) {
            const int u = tid + i * 256;
            const int pix = u >> 2;
            const int pr = pix / TW, pc = pix % TW;
            drc[i] = pr | (pc << 16);
            drel[i] = ((pr * d.wout + pc) * d.cout + dco) * 4;
            dlds[i] = dt_ * DCH + pix * 32 + dh * 16;
        }
        f32x4 bs0 = {0.f, 0.f, 0.f, 0.f}, bs1 = bs0;        // bias gradient: this thread's 8 channels over its pixels (fp32, before the split)
        auto dyload = [&](DyPay& P, int n, int ho0, int wo0, bool live) {
            const int tb = ((n * d.hout + ho0) * d.wout + wo0) * d.cout * 4;
            unsigned m = 0;
#pragma unroll
            for (int i = 0; i < ND; ++i) {
                const bool ok = live && (unsigned)(ho0 + (drc[i] & 0xffff)) < (unsigned)d.hout && (unsigned)(wo0 + (drc[i] >> 16)) < (unsigned)d.wout;
                const int vo = ok ? (tb + drel[i]) : CTL_OOB, vo1 = ok ? (tb + drel[i] + 16) : CTL_OOB;
                P.v0[i] = ctl_bload4(rdy, vo); P.v1[i] = ctl_bload4(rdy, vo1);
                if constexpr (DY2) { P.w0[i] = ctl_bload4(rdy2, vo); P.w1[i] = ctl_bload4(rdy2, vo1); }
                m |= ok ? (1u << i) : 0u;
            }
            P.mask = m;
        };
        auto dystore = [&](const DyPay& P, unsigned char* dyt, int goff) {
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
                f32x4 lo = P.v0[i], hi = P.v1[i];
                if constexpr (DY2) {      // pixels past the image contribute nothing (C alone would)
                    const bool in = (P.mask >> i) & 1u;
                    lo = in ? (a0 * lo + b0 * P.w0[i] + c0) : zero;
                    hi = in ? (a1 * hi + b1 * P.w1[i] + c1) : zero;
                }
                bs0 += lo; bs1 += hi;
                u32x4 ph, pm, pl;
                x3_split8(lo, hi, ph, pm, pl);
                *reinterpret_cast<u32x4*>(dyt + dlds[i]) = ph;
                *reinterpret_cast<u32x4*>(dyt + dlds[i] + DYI) = pm;
                *reinterpret_cast<u32x4*>(dyt + dlds[i] + 2 * DYI) = pl;
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        struct Set { typename XS::Pay xa, xb; DyPay dyp; };
        Set P[R];
        int s_n[R];
        TileWalk lw;
        lw.init(blockIdx.x, gridDim.x, tiles_h, tiles_w);
        int lit = 0;
        auto load_step = [&](Set& st, int& sn) {
            const bool live = lit < my_tiles;
            xs.template load<true>(st.xa, rx, rx, d, lw.n, lw.th * G::TH, lw.tw * TW, 2 * gb, live);
            xs.template load<true>(st.xb, rx, rx, d, lw.n, lw.th * G::TH, lw.tw * TW, 2 * gb + 1, live);
            dyload(st.dyp, lw.n, lw.th * G::TH, lw.tw * TW, live);
            sn = lw.n;
            ++lit;
            lw.next();
        };
#pragma unroll
        for (int r = 0; r < R; ++r) load_step(P[r], s_n[r]);
        TW(8)
        auto step = [&](int it, Set& st, int& sn, bool more) {
            unsigned char* img = smem + (it & 1) * IMG;
            const int grp = sn / group_n;
            xs.template store<true>(st.xa, reinterpret_cast<float*>(img), d, 2 * gb, cf_scale, cf_shift, grp * d.cin, nullptr, rnone, false);
            xs.template store<true>(st.xb, reinterpret_cast<float*>(img + XCH), d, 2 * gb + 1, cf_scale, cf_shift, grp * d.cin, nullptr, rnone, false);
            dystore(st.dyp, img + XBYTES, grp * d.cout);
            TW(0)
            if (more) load_step(st, sn);
            TW(1)
            ctl_barrier_lds_writes_done();
            TW(2)
        };
        const int full = (my_tiles / R) * R;
        for (int it0 = 0; it0 < full; it0 += R) {
#pragma unroll
            for (int r = 0; r < R; ++r) step(it0 + r, P[r], s_n[r], true);
        }
#pragma unroll
        for (int r = 0; r < R - 1; ++r)
            if (full + r < my_tiles) step(full + r, P[r], s_n[r], false);
        // ---- the end of the block: bias sums into their own LDS region, then the consumers' reduction rounds (barriers only)
        *reinterpret_cast<f32x4*>(bsred + tid * 8) = bs0;
        *reinterpret_cast<f32x4*>(bsred + tid * 8 + 4) = bs1;
        ctl_barrier_lds_writes_done();                       // E1
        if (gb == 0 && b_partial != nullptr && tid < 32) {  // thread -> channel cb * 32 + tid: its (cout chunk, half) slots, fixed order
            const int t = tid >> 4, c = tid & 15, h = c >> 3, j = c & 7;
            float v = 0.f;
            for (int sl = t * 2 + h; sl < 256; sl += 4) v += bsred[sl * 8 + j];
            const int co = cb * 32 + tid;
            if (co < cout_p) b_partial[(int64_t)blockIdx.x * cout_p + co] = v;
        }
#pragma unroll
        for (int t0 = 0; t0 < TAPS; t0 += RT) {
            ctl_barrier_lds_reads_done();                    // the round's sums are written
            if (t0 + RT < TAPS) ctl_barrier_lds_reads_done(); // ... and read
        }
        TW_FLUSH(false)
        return;
    }
    // ================================================================ consumers
    __builtin_amdgcn_s_setprio(CTL_X3WPC_CPRIO);
    f32x16 acc[TAPS];
#pragma unroll
    for (int a = 0; a < TAPS; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    const int q = lane >> 4, chunk = q & 1, phalf = q >> 1;
    const int qp = (lane & 15) >> 2, pp = lane & 3;
    // per-lane byte offsets inside a step image (split 0, tile row 0, tap (0, 0)); rows, taps and splits are compile-time immediates
    const int a_off = chunk * XCH + (8 * phalf + qp) * 32 + pp * 8;
    const int b_off = XBYTES + chunk * DCH + (8 * phalf + qp) * 32 + pp * 8;
    for (int it = 0; it < my_tiles; ++it) {
        ctl_barrier_lds_reads_done();                        // tile `it` is staged; this wave's reads of tile it - 1 were consumed
#ifdef CTL_TIMING_X3W
        if (it == 0) TW(6) else TW(3)
        TW_COUNT(5)
#endif
        const unsigned char* img = smem + (it & 1) * IMG;
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) {
            const int tr = 2 * wave + rr;                    // tile row of this k-step (runtime: one address add per row)
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
            a_operand(0, af[0]);
#pragma unroll
            for (int tap = 0; tap < TAPS; ++tap) {
                if (tap + 1 < TAPS) a_operand(tap + 1, af[(tap + 1) & 1]);
                x3_bf16x8* a = af[tap & 1];
                asm volatile("" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]) : : "memory");
#if defined(CTL_X3WPC_ABLATE) && (CTL_X3WPC_ABLATE & 1)      // (timing ablation, WRONG results: no matrix instructions)
                acc[tap][0] += (float)a[0][0] * (float)bf[0][0] + (float)a[1][0] * (float)bf[1][0] + (float)a[2][0] * (float)bf[2][0];
#else
                acc[tap] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], bf[0], acc[tap], 0, 0, 0);
                acc[tap] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], bf[2], acc[tap], 0, 0, 0);
                acc[tap] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], bf[1], acc[tap], 0, 0, 0);
                acc[tap] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], bf[0], acc[tap], 0, 0, 0);
                acc[tap] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], bf[1], acc[tap], 0, 0, 0);
                acc[tap] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], bf[0], acc[tap], 0, 0, 0);
#endif
            }
        }
        TW(4)
    }
    ctl_barrier_lds_reads_done();                            // E1: every wave has left the last image
    // ---------------- the four waves' sums through LDS, RT taps per round; thread -> (register r, lane l) of a tap: ci = 8 (r >> 2) + 4 (l >> 5) + (r & 3)
    float* red = reinterpret_cast<float*>(smem);
    const int ctid = tid - 256;
    const int64_t split_base = (int64_t)blockIdx.x * TAPS * cin_p * cout_p;
#pragma unroll
    for (int t0 = 0; t0 < TAPS; t0 += RT) {
#pragma unroll
        for (int tp = 0; tp < RT; ++tp) {
            const int tap = t0 + tp;
            if (tap < TAPS) {
                float* r0 = red + (tp * 4 + wave) * 16 * 64 + lane;
#pragma unroll
                for (int r = 0; r < 16; ++r) r0[r * 64] = acc[tap][r];
            }
        }
        ctl_barrier_lds_writes_done();
#pragma unroll
        for (int tp = 0; tp < RT; ++tp) {
            const int tap = t0 + tp;
            if (tap < TAPS) {
#pragma unroll
                for (int e0 = 0; e0 < 16 * 64; e0 += 256) {
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
    TW(7)
    TW_FLUSH(true)
}

// ------------------------------------------------------------------------------------------------ software-pipelined form (round 5)
// conv_wgrad_x3pipe_kernel: the same 32 cin x 32 cout blocks, images and lane maps as the producer / consumer kernel above, but ONE
// instruction stream per SIMD (256-thread blocks, one per CU): while a wave's MFMAs of tile t read LDS image t & 1, the SAME wave's
// staging of tile t + 1 (prologue, split, LDS writes into image (t + 1) & 1) and the loads of tile t + 1 + R are interleaved with them.
// Why: the phase timers of the producer / consumer kernel (profiles/r5_wgrad_pc_experiments.txt) show that on a SIMD the matrix stream
// of one wave and the vector stream of ANOTHER wave do not overlap -- their times add (staging alone 3400 cycles per tile, beside the
// MFMA wave 5800; MFMA phase 2900; tile period 6300), whatever the wave priorities; what the hardware does overlap is an MFMA with the
// FOLLOWING independent instructions of its own wave (MI355X_MICROARCH.md: <= 5 single-issue instructions hidden per 32-cycle MFMA).
// The tile body is one basic block (every load unconditional and single-path, prologue choice a template argument); the interleave is
// requested from the scheduler with sched_group_barrier (1 MFMA : 1-2 LDS : 4-5 VALU : occasional VMEM).  One barrier per tile.
#ifndef CTL_X3W_PIPE_SETS
#define CTL_X3W_PIPE_SETS 2
#endif
#ifndef CTL_X3W_PIPE_VALU
#define CTL_X3W_PIPE_VALU 4      // VALU instructions requested behind every MFMA
#endif
template <int KS, int S, int MODE, bool DY2, int PROK>
__global__ __launch_bounds__(256, 1) void conv_wgrad_x3pipe_kernel(const ctl_conv d, const float* __restrict__ x, const float* __restrict__ pro_scale,
                                                                   const float* __restrict__ pro_shift, const float* __restrict__ dy,
                                                                   const float* __restrict__ dy2, const float* __restrict__ dy_coef,
                                                                   float* __restrict__ w_partial, float* __restrict__ b_partial, int tiles_h, int tiles_w,
                                                                   int ntiles, int cin_p, int cout_p) {
    static_assert(S == 1, "stride 1");
    constexpr int MT = 2, TW = 16;
    using G = Geom<KS, S, MT, TW>;
    using XS = XStage3<KS, S, MODE, MT, TW, false, false>;
    constexpr int TAPS = KS * KS;
    constexpr int XCH = XS::XT_BYTES + 128, DCH = G::TP * 32 + 128;      // chunk strides: 128 B (mod 256) apart, see the producer / consumer kernel
    constexpr int XBYTES = 2 * XCH, DYI = 2 * DCH, IMG = XBYTES + 3 * DYI;
    constexpr int TAP_BYTES = 4 * 16 * 64 * 4;
    constexpr int RT = (2 * IMG) / TAP_BYTES < TAPS ? (2 * IMG) / TAP_BYTES : TAPS;
    constexpr int COEF = (DY2 ? 5 : 2) * CTL_PRO_MAX * 4;
    static_assert(2 * IMG + COEF + 256 * 8 * 4 <= 160 * 1024, "LDS budget of one CU");
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * IMG + COEF + 256 * 8 * 4];
    float* cf_scale = reinterpret_cast<float*>(smem + 2 * IMG);
    float* cf_shift = cf_scale + CTL_PRO_MAX;
    float* cd = cf_shift + CTL_PRO_MAX;
    float* bsred = reinterpret_cast<float*>(smem + 2 * IMG + COEF);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int gb = blockIdx.y, cb = blockIdx.z;
    const int group_n = d.n / (d.groups > 1 ? d.groups : 1);
    const int my_tiles = ((int)blockIdx.x < ntiles) ? (ntiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x : 0;
    if (PROK)
        for (int i = tid; i < (d.groups > 1 ? d.groups : 1) * d.cin; i += 256) { cf_scale[i] = pro_scale[i]; cf_shift[i] = pro_shift[i]; }
    if constexpr (DY2) {
        for (int i = tid; i < (d.groups > 1 ? d.groups : 1) * d.cout; i += 256) {
            const int gi = i / d.cout, ch = i - gi * d.cout;
            cd[i] = dy_coef[(gi * 3 + 0) * d.cout + ch]; cd[CTL_PRO_MAX + i] = dy_coef[(gi * 3 + 1) * d.cout + ch];
            cd[2 * CTL_PRO_MAX + i] = dy_coef[(gi * 3 + 2) * d.cout + ch];
        }
    }
    __syncthreads();

    constexpr int R = CTL_X3W_PIPE_SETS;
    const __amdgpu_buffer_rsrc_t rx = ctl_rsrc(x, (int64_t)d.n * d.hin * d.win * d.cin * 4);
    const __amdgpu_buffer_rsrc_t rdy = ctl_rsrc(dy, (int64_t)d.n * d.hout * d.wout * d.cout * 4);
    const __amdgpu_buffer_rsrc_t rdy2 = DY2 ? ctl_rsrc(dy2, (int64_t)d.n * d.hout * d.wout * d.cout * 4) : rdy;
    const __amdgpu_buffer_rsrc_t rnone = ctl_rsrc((const void*)nullptr, 0);
    XS xs;
    xs.init(d);
    constexpr int DU = G::TP * 4, ND = DU / 256;
    static_assert(DU % 256 == 0, "dy units");
    struct DyPay { f32x4 v0[ND], v1[ND], w0[DY2 ? ND : 1], w1[DY2 ? ND : 1]; unsigned mask; };
    int drel[ND], drc[ND], dlds[ND];
    const int drest = tid & 3, dt_ = drest >> 1, dh = drest & 1;
    const int dco = cb * 32 + dt_ * 16 + dh * 8;
#pragma unroll
    for (int i = 0; i < ND; ++i) {
        const int u = tid + i * 256;
        const int pix = u >> 2;
        const int pr = pix / TW, pc = pix % TW;
        drc[i] = pr | (pc << 16);
        drel[i] = ((pr * d.wout + pc) * d.cout + dco) * 4;
        dlds[i] = dt_ * DCH + pix * 32 + dh * 16;
    }
    f32x4 bs0 = {0.f, 0.f, 0.f, 0.f}, bs1 = bs0;
    auto dyload = [&](DyPay& P, int n, int ho0, int wo0, bool live) {
        const int tb = ((n * d.hout + ho0) * d.wout + wo0) * d.cout * 4;
        unsigned m = 0;
#pragma unroll
        for (int i = 0; i < ND; ++i) {
            const bool ok = live && (unsigned)(ho0 + (drc[i] & 0xffff)) < (unsigned)d.hout && (unsigned)(wo0 + (drc[i] >> 16)) < (unsigned)d.wout;
            const int vo = ok ? (tb + drel[i]) : CTL_OOB, vo1 = ok ? (tb + drel[i] + 16) : CTL_OOB;
            P.v0[i] = ctl_bload4(rdy, vo); P.v1[i] = ctl_bload4(rdy, vo1);
            if constexpr (DY2) { P.w0[i] = ctl_bload4(rdy2, vo); P.w1[i] = ctl_bload4(rdy2, vo1); }
            m |= ok ? (1u << i) : 0u;
        }
        P.mask = m;
    };
    auto dystore = [&](const DyPay& P, unsigned char* dyt, int goff) {
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
            f32x4 lo = P.v0[i], hi = P.v1[i];
            if constexpr (DY2) {      // (branch-free form of `in ? ... : 0`: the tile body is one basic block)
                const bool in = (P.mask >> i) & 1u;
                lo = x3_keep4(a0 * lo + b0 * P.w0[i] + c0, in);
                hi = x3_keep4(a1 * hi + b1 * P.w1[i] + c1, in);
            }
            bs0 += lo; bs1 += hi;
            u32x4 ph, pm, pl;
            x3_split8(lo, hi, ph, pm, pl);
            *reinterpret_cast<u32x4*>(dyt + dlds[i]) = ph;
            *reinterpret_cast<u32x4*>(dyt + dlds[i] + DYI) = pm;
            *reinterpret_cast<u32x4*>(dyt + dlds[i] + 2 * DYI) = pl;
        }
    };
    struct Set { typename XS::Pay xa, xb; DyPay dyp; };
    Set P[R];
    int s_n[R];
    TileWalk lw;
    lw.init(blockIdx.x, gridDim.x, tiles_h, tiles_w);
    int lit = 0;
    auto load_step = [&](Set& st, int& sn) {
        const bool live = lit < my_tiles;
        xs.template load<true>(st.xa, rx, rx, d, lw.n, lw.th * G::TH, lw.tw * TW, 2 * gb, live);
        xs.template load<true>(st.xb, rx, rx, d, lw.n, lw.th * G::TH, lw.tw * TW, 2 * gb + 1, live);
        dyload(st.dyp, lw.n, lw.th * G::TH, lw.tw * TW, live);
        sn = lw.n;
        ++lit;
        lw.next();
    };
    auto stage = [&](int it, Set& st, int sn) {                  // tile `it` -> image it & 1
        unsigned char* img = smem + (it & 1) * IMG;
        const int grp = sn / group_n;
        xs.template store<false, PROK>(st.xa, reinterpret_cast<float*>(img), d, 2 * gb, cf_scale, cf_shift, grp * d.cin, nullptr, rnone, false);
        xs.template store<false, PROK>(st.xb, reinterpret_cast<float*>(img + XCH), d, 2 * gb + 1, cf_scale, cf_shift, grp * d.cin, nullptr, rnone, false);
        dystore(st.dyp, img + XBYTES, grp * d.cout);
    };
    f32x16 acc[TAPS];
#pragma unroll
    for (int a = 0; a < TAPS; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    const int q = lane >> 4, chunk = q & 1, phalf = q >> 1;
    const int qp = (lane & 15) >> 2, pp = lane & 3;
    const int a_off = chunk * XCH + (8 * phalf + qp) * 32 + pp * 8;
    const int b_off = XBYTES + chunk * DCH + (8 * phalf + qp) * 32 + pp * 8;
    // ---- the staging of one tile cut into PIECES (a dozen instructions each) that the tile body places behind pairs of MFMAs.  Units of
    // the tile being staged: x chunk a (XS::NU units), x chunk b, dy (ND units); per unit: begin (prologue / virtual gradient), four
    // pair splits, the three LDS writes, and the reload of the unit's registers with the tile R tiles on (its buffer loads).
    constexpr int NUX = XS::NU, NUNITS = 2 * NUX + ND, PPU = 7, NPIECES = NUNITS * PPU + 1;
    static_assert(NPIECES <= TAPS * 6, "one piece behind every pair of MFMAs");
    f32x4 ulo, uhi;
    unsigned uh[4], um[4], ul[4];
    // scalars of the tile that is being LOADED (tile lit): set once per body by `next_tile`
    int l_n = 0, l_tbx = 0, l_tbd = 0, l_vh0 = 0, l_vw0 = 0, l_ho0 = 0, l_wo0 = 0;
    bool l_live = false;
    constexpr bool PLAIN = (MODE == CTL_IN_PLAIN);
    const unsigned hv = PLAIN ? d.hin : 2 * d.hin, wv = PLAIN ? d.win : 2 * d.win;
    auto next_tile = [&]() {
        l_live = lit < my_tiles;
        l_n = lw.n; l_ho0 = lw.th * G::TH; l_wo0 = lw.tw * TW;
        l_vh0 = l_ho0 * S - xs.pad_h; l_vw0 = l_wo0 * S - xs.pad_w;
        const int oh = PLAIN ? l_vh0 : ((l_ho0 >> 1) - XS::PADH), ow = PLAIN ? l_vw0 : ((l_wo0 >> 1) - XS::PADH);
        l_tbx = (((l_n * d.hin + oh) * d.win + ow) * d.cin + 2 * gb * 16) * 4;
        l_tbd = ((l_n * d.hout + l_ho0) * d.wout + l_wo0) * d.cout * 4;
        ++lit;
        lw.next();
    };
    auto x_begin = [&](const typename XS::Pay& pay, int i, int chunk, int grp) {
        f32x4 lo = pay.v0[i], hi = pay.v1[i];
        if constexpr (PROK != 0) {
            const float* ps = cf_scale + grp * d.cin + (2 * gb + chunk) * 16 + (tid & 1) * 8;
            const float* pb = cf_shift + grp * d.cin + (2 * gb + chunk) * 16 + (tid & 1) * 8;
            const f32x4 a0 = *reinterpret_cast<const f32x4*>(ps), a1 = *reinterpret_cast<const f32x4*>(ps + 4);
            const f32x4 b0 = *reinterpret_cast<const f32x4*>(pb), b1 = *reinterpret_cast<const f32x4*>(pb + 4);
            const bool in = (pay.vmask >> i) & 1u;      // (padding stays zero under the affine prologue; branch-free)
            lo = x3_keep4(ctl_leaky01(lo * a0 + b0, d.pro_slope), in);
            hi = x3_keep4(ctl_leaky01(hi * a1 + b1, d.pro_slope), in);
        }
        ulo = lo; uhi = hi;
    };
    auto d_begin = [&](const DyPay& P, int i, int grp) {
        f32x4 lo = P.v0[i], hi = P.v1[i];
        if constexpr (DY2) {
            const float* cc = cd + grp * d.cout + dco;
            const f32x4 a0 = *reinterpret_cast<const f32x4*>(cc), a1 = *reinterpret_cast<const f32x4*>(cc + 4);
            const f32x4 b0 = *reinterpret_cast<const f32x4*>(cc + CTL_PRO_MAX), b1 = *reinterpret_cast<const f32x4*>(cc + CTL_PRO_MAX + 4);
            const f32x4 c0 = *reinterpret_cast<const f32x4*>(cc + 2 * CTL_PRO_MAX), c1 = *reinterpret_cast<const f32x4*>(cc + 2 * CTL_PRO_MAX + 4);
            const bool in = (P.mask >> i) & 1u;
            lo = x3_keep4(a0 * lo + b0 * P.w0[i] + c0, in);
            hi = x3_keep4(a1 * hi + b1 * P.w1[i] + c1, in);
        }
        bs0 += lo; bs1 += hi;
        ulo = lo; uhi = hi;
    };
    auto u_pair = [&](int pr) {
        const x3_f32x2 e = pr == 0 ? x3_f32x2{ulo.x, ulo.y} : pr == 1 ? x3_f32x2{ulo.z, ulo.w} : pr == 2 ? x3_f32x2{uhi.x, uhi.y} : x3_f32x2{uhi.z, uhi.w};
        x3_split2(e, uh[pr], um[pr], ul[pr]);
    };
    auto u_write = [&](unsigned char* base, int split_stride) {
        *reinterpret_cast<u32x4*>(base) = u32x4{uh[0], uh[1], uh[2], uh[3]};
        *reinterpret_cast<u32x4*>(base + split_stride) = u32x4{um[0], um[1], um[2], um[3]};
        *reinterpret_cast<u32x4*>(base + 2 * split_stride) = u32x4{ul[0], ul[1], ul[2], ul[3]};
    };
    auto x_reload = [&](typename XS::Pay& pay, int i, int chunk) {
        const int vh = l_vh0 + (xs.rc[i] & 0xffff), vw = l_vw0 + (xs.rc[i] >> 16);
        const bool ok = l_live && (unsigned)vh < hv && (unsigned)vw < wv;
        const int vo = ok ? (l_tbx + chunk * 64 + xs.rel[i]) : CTL_OOB, vo1 = ok ? (l_tbx + chunk * 64 + xs.rel[i] + 16) : CTL_OOB;
        pay.v0[i] = ctl_bload4(rx, vo); pay.v1[i] = ctl_bload4(rx, vo1);
        pay.vmask = (pay.vmask & ~(1u << i)) | (ok ? (1u << i) : 0u);
    };
    auto d_reload = [&](DyPay& P, int i) {
        const bool ok = l_live && (unsigned)(l_ho0 + (drc[i] & 0xffff)) < (unsigned)d.hout && (unsigned)(l_wo0 + (drc[i] >> 16)) < (unsigned)d.wout;
        const int vo = ok ? (l_tbd + drel[i]) : CTL_OOB, vo1 = ok ? (l_tbd + drel[i] + 16) : CTL_OOB;
        P.v0[i] = ctl_bload4(rdy, vo); P.v1[i] = ctl_bload4(rdy, vo1);
        if constexpr (DY2) { P.w0[i] = ctl_bload4(rdy2, vo); P.w1[i] = ctl_bload4(rdy2, vo1); }
        P.mask = (P.mask & ~(1u << i)) | (ok ? (1u << i) : 0u);
    };
    // piece K of the tile body: stages tile (it + 1) from set `st` into image `img`, reloads the set with tile lit
    auto piece = [&](auto ktag, Set& st, int sn, unsigned char* img) {
        constexpr int K = decltype(ktag)::value;
        if constexpr (K == 0) {
            next_tile();
        } else if constexpr (K - 1 < NUNITS * PPU) {
            constexpr int U = (K - 1) / PPU, M = (K - 1) % PPU;
            const int grp = sn / group_n;
            if constexpr (U < 2 * NUX) {
                constexpr int CH = U / NUX, I = U % NUX;
                typename XS::Pay& pay = CH ? st.xb : st.xa;
                if constexpr (M == 0) x_begin(pay, I, CH, grp);
                else if constexpr (M <= 4) u_pair(M - 1);
                else if constexpr (M == 5) u_write(img + CH * XCH + xs.lds[I], XS::SPLIT);
                else x_reload(pay, I, CH);
            } else {
                constexpr int I = U - 2 * NUX;
                if constexpr (M == 0) d_begin(st.dyp, I, grp);
                else if constexpr (M <= 4) u_pair(M - 1);
                else if constexpr (M == 5) u_write(img + XBYTES + dlds[I], DYI);
                else d_reload(st.dyp, I);
            }
        }
    };
    // ---- head: R tiles of loads in flight, tile 0 staged
#pragma unroll
    for (int r = 0; r < R; ++r) load_step(P[r], s_n[r]);
    stage(0, P[0], s_n[0]);
    load_step(P[0], s_n[0]);                                     // set 0 now holds tile R
    ctl_barrier_lds_writes_done();
    // ---- tile body: the MFMAs of tile `it` (this wave's two tile rows, all taps) with the pieces of tile it + 1 between them.  Every
    // (two MFMAs, one piece) group is fenced (sched_barrier): left to itself the scheduler puts all 108 MFMAs first and the staging
    // behind them (ISA checked), and sched_group_barrier requests did not move it.
    auto body = [&](int it, Set& nxt, int& nxt_n) {
        const unsigned char* img = smem + (it & 1) * IMG;
        unsigned char* wimg = smem + ((it + 1) & 1) * IMG;
        const int staged_n = nxt_n;                              // image index of the tile being staged (piece 0 moves on to the tile to load)
        const unsigned char* arow[2];
        const unsigned char* brow[2];
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) {
            arow[rr] = img + a_off + (2 * wave + rr) * (G::IWP * 32);
            brow[rr] = img + b_off + (2 * wave + rr) * (TW * 32);
        }
        x3_bf16x8 bf[3], af[2][3];
        auto a_operand = [&](int rr, int tap, x3_bf16x8* a) {
            const int kh = tap / KS, kw = tap % KS;
            const unsigned char* a0 = arow[rr] + (kh * G::IWP + kw) * 32;
#pragma unroll
            for (int sp = 0; sp < 3; ++sp) a[sp] = x3_tr_read8(a0 + sp * XS::SPLIT, a0 + sp * XS::SPLIT + 4 * 32);
        };
        ctl_static_for<2 * TAPS * 3>([&](auto gtag) {
            constexpr int GI = decltype(gtag)::value, RR = GI / (TAPS * 3), TAP = (GI % (TAPS * 3)) / 3, J = GI % 3;
            if constexpr (TAP == 0 && J == 0) {
#pragma unroll
                for (int sp = 0; sp < 3; ++sp) bf[sp] = x3_tr_read8(brow[RR] + sp * DYI, brow[RR] + sp * DYI + 4 * 32);
                a_operand(RR, 0, af[0]);
            }
            if constexpr (J == 0 && TAP + 1 < TAPS) a_operand(RR, TAP + 1, af[(TAP + 1) & 1]);
            x3_bf16x8* a = af[TAP & 1];
            constexpr int SA0 = J == 0 ? 2 : J == 1 ? 1 : 0, SB0 = J == 0 ? 0 : 1;
            constexpr int SA1 = J == 0 ? 0 : J == 1 ? 1 : 0, SB1 = J == 0 ? 2 : 0;
            acc[TAP] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[SA0], bf[SB0], acc[TAP], 0, 0, 0);
            acc[TAP] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[SA1], bf[SB1], acc[TAP], 0, 0, 0);
            piece(std::integral_constant<int, GI>{}, nxt, staged_n, wimg);
            __builtin_amdgcn_sched_barrier(0);
        });
        nxt_n = l_n;
        ctl_barrier_lds_writes_done();
    };
    static_assert(R == 2, "the body below alternates two payload sets");
    int it = 0;
    for (; it + 2 <= my_tiles; it += 2) {
        body(it, P[1], s_n[1]);                                  // tile it + 1 lives in set (it + 1) % 2 = 1 (it is even)
        body(it + 1, P[0], s_n[0]);
    }
    if (it < my_tiles) body(it, P[1], s_n[1]);
    // ---- the end of the block: bias sums, then the four waves' accumulators through LDS (every image is idle behind the last barrier)
    *reinterpret_cast<f32x4*>(bsred + tid * 8) = bs0;
    *reinterpret_cast<f32x4*>(bsred + tid * 8 + 4) = bs1;
    float* red = reinterpret_cast<float*>(smem);
    const int64_t split_base = (int64_t)blockIdx.x * TAPS * cin_p * cout_p;
#pragma unroll
    for (int t0 = 0; t0 < TAPS; t0 += RT) {
#pragma unroll
        for (int tp = 0; tp < RT; ++tp) {
            const int tap = t0 + tp;
            if (tap < TAPS) {
                float* r0 = red + (tp * 4 + wave) * 16 * 64 + lane;
#pragma unroll
                for (int r = 0; r < 16; ++r) r0[r * 64] = acc[tap][r];
            }
        }
        ctl_barrier_lds_writes_done();
        if (t0 == 0 && gb == 0 && b_partial != nullptr && tid < 32) {
            const int t = tid >> 4, c = tid & 15, h = c >> 3, j = c & 7;
            float v = 0.f;
            for (int sl = t * 2 + h; sl < 256; sl += 4) v += bsred[sl * 8 + j];
            const int co = cb * 32 + tid;
            if (co < cout_p) b_partial[(int64_t)blockIdx.x * cout_p + co] = v;
        }
#pragma unroll
        for (int tp = 0; tp < RT; ++tp) {
            const int tap = t0 + tp;
            if (tap < TAPS) {
#pragma unroll
                for (int e0 = 0; e0 < 16 * 64; e0 += 256) {
                    const int e = e0 + tid, r = e >> 6, l = e & 63;
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
#define CTL_X3W_PC_DEFAULT 1
#endif
#ifndef CTL_X3W_PC_MIN_TILES
#define CTL_X3W_PC_MIN_TILES 4
#endif
#ifndef CTL_X3W_PIPE_DEFAULT
#define CTL_X3W_PIPE_DEFAULT 1
#endif
template <int MODE, bool DY2, int PROK>
static void wgrad3_launch_pipe(wgrad3_call& a, dim3 grid) {
    conv_wgrad_x3pipe_kernel<3, 1, MODE, DY2, PROK><<<grid, dim3(256), 0, a.stream>>>(*a.d, a.x, a.pro_scale, a.pro_shift, a.dy, a.dy2, a.dy_coef, a.w_partial, a.b_partial,
                                                                                   a.c.tiles_h, a.c.tiles_w, a.ntiles, a.cin_p, a.cout_p);
}
template <int MODE>
static void wgrad3_go_pc(wgrad3_call& a) {
    const dim3 grid((unsigned)a.splits, (unsigned)(a.cin_p / 32), (unsigned)(a.cout_p / 32));
    static const int pipe_on = ctl_tune_int("CTL_X3W_PIPE", CTL_X3W_PIPE_DEFAULT);
    if (pipe_on) {      // the software-pipelined single-stream form
        const bool pro = a.d->pro_affine != 0;
        if (a.dy2) { if (pro) wgrad3_launch_pipe<MODE, true, 1>(a, grid); else wgrad3_launch_pipe<MODE, true, 0>(a, grid); }
        else { if (pro) wgrad3_launch_pipe<MODE, false, 1>(a, grid); else wgrad3_launch_pipe<MODE, false, 0>(a, grid); }
        return;
    }
    if (a.dy2) conv_wgrad_x3pc_kernel<3, 1, MODE, true><<<grid, dim3(512), 0, a.stream>>>(*a.d, a.x, a.pro_scale, a.pro_shift, a.dy, a.dy2, a.dy_coef, a.w_partial, a.b_partial,
                                                                                        a.c.tiles_h, a.c.tiles_w, a.ntiles, a.cin_p, a.cout_p);
    else conv_wgrad_x3pc_kernel<3, 1, MODE, false><<<grid, dim3(512), 0, a.stream>>>(*a.d, a.x, a.pro_scale, a.pro_shift, a.dy, a.dy2, a.dy_coef, a.w_partial, a.b_partial,
                                                                                     a.c.tiles_h, a.c.tiles_w, a.ntiles, a.cin_p, a.cout_p);
}
// one block per CU, (cin / 32) x (cout / 32) output blocks: the pixel tiles are split over what is left of the 256 CUs
static bool wgrad3_pc_pick(const ctl_conv* d, wgrad3_call* a) {
    static const int pc_on = ctl_tune_int("CTL_X3W_PC", CTL_X3W_PC_DEFAULT), min_tiles = ctl_tune_int("CTL_X3W_PC_MIN_TILES", CTL_X3W_PC_MIN_TILES);
    if (!pc_on || d->ks != 3 || d->stride != 1 || (d->in_mode != CTL_IN_PLAIN && d->in_mode != CTL_IN_UP2) || d->cin % 32 || d->cout % 32 || a->c.mt != 2) return false;
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
