// X3: fp32 tensors contracted on the bf16 matrix pipe.
//
// gfx950 runs v_mfma_f32_16x16x4_f32 at 1/16 of the rate of v_mfma_f32_16x16x32_bf16 (MI355X_MICROARCH.md: 157 vs 2500 TFLOP/s dense).
// Every fp32 number is EXACTLY the sum of three bf16 numbers: hi = rne_bf16(x), mid = rne_bf16(x - hi), lo = x - hi - mid (each step
// takes 8 significant bits plus the sign of the remainder; the two subtractions are exact in fp32, and what is left after two steps
// has at most 24 - 16 = 8 significant bits, so `lo` needs no rounding).  A product x*w is then the sum of nine bf16 x bf16 products,
// each of which the matrix pipe forms exactly (16 significant bits) and adds into an fp32 accumulator.  The three smallest -- mid*lo,
// lo*mid, lo*lo, together at most 2^-24 (1 + 2^-8) |x*w| (|mid| <= 2^-8 |x|, |lo| <= 2^-17 |x|; worst case x = w = 1 + 2^-8 - 2^-16 + 2^-17:
// tests/test_x3_split_cpu.py), typically 2^-25 and less -- are dropped: six MFMAs
//     hi*hi + hi*mid + mid*hi + mid*mid + hi*lo + lo*hi
// per fp32 contraction step, each with K = 32, against eight fp32 MFMAs with K = 4 for the same K: 16/6 = 2.7x the fp32 matrix rate at an
// error per product at the rounding of one fp32 multiply (2^-24).  Tensors in HBM stay fp32 (same bytes as the fp32 family, same
// epilogues); what changes is what the staging writes to LDS (three bf16 planes per half-chunk instead of one fp32 image: 1.5x the
// bytes), the weight fragments (packed once per optimizer step as three bf16 planes, ctl_conv_x3.hip) and the matrix instruction.
//
// LDS image of one 16-channel chunk of the input tile: six planes [split s][half h] of [row][col][8 channels] bf16 = 16 B per pixel.
// A B-operand read (ds_read_b128) is serviced in four groups of 16 lanes, each made of 8 lanes of lane-row q and 8 of row q ^ 1 with
// COMPLEMENTARY pixel sets (MI355X_MICROARCH.md, LDS table: {0-3, 12-15, 20-27}, ...): rows q and q ^ 1 read the two halves h of one
// tap, so with every plane starting at a multiple of 256 B the 16 lanes of a group cover 256 contiguous bytes modulo the plane -- every
// bank once.  A staging write (ds_write_b128: 8 groups of 8 consecutive lanes) is conflict-free when 8 consecutive lanes write 8
// consecutive pixels of ONE plane, hence the unit -> thread mapping below (lanes 0-7: half 0 of pixels 0-7, lanes 8-15: half 1).
#pragma once
#include "ctl_conv_common.h"

typedef __bf16 x3_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 x3_bf16x2 __attribute__((ext_vector_type(2)));
typedef float x3_f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned x3_pack2(float a, float b) {
    return __builtin_bit_cast(unsigned, __builtin_convertvector(x3_f32x2{a, b}, x3_bf16x2));      // v_cvt_pk_bf16_f32, round to nearest even
}
__device__ __forceinline__ u32x4 x3_pack8(f32x4 a, f32x4 b) {
    return u32x4{x3_pack2(a.x, a.y), x3_pack2(a.z, a.w), x3_pack2(b.x, b.y), x3_pack2(b.z, b.w)};
}
__device__ __forceinline__ f32x4 x3_unpack4(unsigned a, unsigned b) {
    return f32x4{__builtin_bit_cast(float, a << 16), __builtin_bit_cast(float, a & 0xffff0000u), __builtin_bit_cast(float, b << 16),
                 __builtin_bit_cast(float, b & 0xffff0000u)};
}
// the exact three-way split of 8 fp32 values (a = channels 0-3, b = 4-7 of a unit) into three packed bf16x8.  Written on PAIRS: a pair of
// floats -> one packed dword (v_cvt_pk_bf16_f32) -> the pair it stands for (shift / mask into a register pair) -> one packed subtraction
// (v_pk_add_f32 with negated operand): 9 VALU instructions per pair, no register moves (as f32x4 arithmetic the compiler scalarised the
// subtractions and shuffled the halves through ~30 v_mov per unit)
__device__ __forceinline__ x3_f32x2 x3_up2(unsigned p) {
    return x3_f32x2{__builtin_bit_cast(float, p << 16), __builtin_bit_cast(float, p & 0xffff0000u)};
}
#ifndef CTL_X3_DOT2_SPLIT
#define CTL_X3_DOT2_SPLIT 1
#endif
// x - bf16 element `k` of the packed pair p, in ONE instruction: v_dot2c_f32_bf16  d = p[0] * b[0] + p[1] * b[1] + x  with b = (-1, 0) or
// (0, -1).  The products are exact and the result is exactly representable, so it is the same number the unpack + subtract form gives
// (tests/test_x3_gpu.py::test_split_is_exact_on_adversarial_values runs this code); 7 instead of 9 VALU instructions per pair.
// The selector travels in a scalar register the compiler cannot see through: written as a constant it is encoded as the INLINE constant
// -1.0, which this instruction reads as the 32-bit pattern 0xBF800000 = (0, -1) whichever element was meant (conv results off by O(1)).
template <int K>
__device__ __forceinline__ float x3_sub_part(float x, unsigned p) {
    unsigned sel;
    if constexpr (K == 0) asm("s_mov_b32 %0, 0x0000bf80" : "=s"(sel));
    else asm("s_mov_b32 %0, 0xbf800000" : "=s"(sel));
    return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(x3_bf16x2, p), __builtin_bit_cast(x3_bf16x2, sel), x, false);
}
#ifndef CTL_X3_SPLIT_MODE
#define CTL_X3_SPLIT_MODE 0      // 0: v_dot2c form (7 instructions per pair); 1: plain fp32 instructions only (11 per pair: shifts / masks / v_sub_f32 --
#endif                           // no packed or dot instruction, which the matrix pipe's neighbours pay for); 2: timing ablation, WRONG results (no split)
__device__ __forceinline__ float x3_opaque(float v) { asm volatile("" : "+v"(v)); return v; }      // (keeps the SLP vectoriser from re-packing the scalar form)
__device__ __forceinline__ void x3_split2(x3_f32x2 e, unsigned& h, unsigned& m, unsigned& l) {
    h = x3_pack2(e.x, e.y);
#if CTL_X3_SPLIT_MODE == 2
    m = h; l = h;
#elif CTL_X3_SPLIT_MODE == 1
    const float r0 = x3_opaque(e.x - __builtin_bit_cast(float, h << 16)), r1 = x3_opaque(e.y - __builtin_bit_cast(float, h & 0xffff0000u));      // exact
    m = x3_pack2(r0, r1);
    const float l0 = x3_opaque(r0 - __builtin_bit_cast(float, m << 16)), l1 = x3_opaque(r1 - __builtin_bit_cast(float, m & 0xffff0000u));        // exact, representable
    l = x3_pack2(l0, l1);
#elif CTL_X3_DOT2_SPLIT
    const float r0 = x3_sub_part<0>(e.x, h), r1 = x3_sub_part<1>(e.y, h);      // exact
    m = x3_pack2(r0, r1);
    l = x3_pack2(x3_sub_part<0>(r0, m), x3_sub_part<1>(r1, m));                 // exact, and exactly representable
#else
    const x3_f32x2 r = e - x3_up2(h);            // exact
    m = x3_pack2(r.x, r.y);
    const x3_f32x2 r2 = r - x3_up2(m);           // exact, and exactly representable
    l = x3_pack2(r2.x, r2.y);
#endif
}
__device__ __forceinline__ void x3_split8(f32x4 a, f32x4 b, u32x4& ph, u32x4& pm, u32x4& pl) {
    unsigned h[4], m[4], l[4];
    x3_split2(x3_f32x2{a.x, a.y}, h[0], m[0], l[0]);
    x3_split2(x3_f32x2{a.z, a.w}, h[1], m[1], l[1]);
    x3_split2(x3_f32x2{b.x, b.y}, h[2], m[2], l[2]);
    x3_split2(x3_f32x2{b.z, b.w}, h[3], m[3], l[3]);
    ph = u32x4{h[0], h[1], h[2], h[3]}; pm = u32x4{m[0], m[1], m[2], m[3]}; pl = u32x4{l[0], l[1], l[2], l[3]};
}
// v where `on`, +0 elsewhere, branch-free and NaN-safe (a bit mask, not a multiplication by 0 / 1: a dead tile of the software-pipelined
// kernels reads coefficient rows past its table, and 0 * garbage may be NaN)
__device__ __forceinline__ f32x4 x3_keep4(f32x4 v, bool on) {
    const unsigned m = on ? 0xffffffffu : 0u;
    return f32x4{__builtin_bit_cast(float, __builtin_bit_cast(unsigned, v.x) & m), __builtin_bit_cast(float, __builtin_bit_cast(unsigned, v.y) & m),
                 __builtin_bit_cast(float, __builtin_bit_cast(unsigned, v.z) & m), __builtin_bit_cast(float, __builtin_bit_cast(unsigned, v.w) & m)};
}
// scalar form (weight packing): split s of v
__device__ __forceinline__ float x3_part(float v, int s) {
    auto rne = [](float f) { return __builtin_bit_cast(float, x3_pack2(f, 0.f) << 16); };
    const float h = rne(v);
    if (s == 0) return h;
    const float m = rne(v - h);
    return s == 1 ? m : (v - h) - m;
}

// Staging of one 16-channel chunk of the (virtual) input tile; a unit = 8 channels of a pixel = two 16-byte fp32 loads -> three 16-byte
// bf16 LDS writes.  Same tile geometry, bounds handling, prologues (BatchNorm apply + LeakyReLU; X2: the BatchNorm-backward prologue on
// two fp32 tensors, with the optional side output of the virtual tensor) as XStage.  cin is a multiple of 16 (checked on the host).
// PLANAR = false (the weight-gradient kernel, whose transposed reads want whole pixel rows): three interleaved images [row][col][16 channels]
// bf16 = 32 B per pixel; 8 consecutive lanes then write both halves of 4 consecutive pixels (lane = 2 pixel + half): 128 contiguous bytes.
template <int KS, int S, int MODE, int MT, int TW, bool X2 = false, bool PLANAR = true>
struct XStage3 {
    using G = Geom<KS, S, MT, TW>;
    static constexpr int NPIX = G::IH * G::IW;
    static constexpr int SLOTS = PLANAR ? ((NPIX + 7) / 8) * 16 : NPIX * 2;
    static constexpr int NU = (SLOTS + 255) / 256;
    static constexpr int PADH = (G::PAD + 1) >> 1;
    static constexpr int PLANE = ((G::IH * G::IWP * 16 + 16 + 255) / 256) * 256;      // >= 16 B of slack behind every plane: the dump slot
    static constexpr int IMAGE = ((G::IH * G::IWP * 32 + 16 + 255) / 256) * 256;      // (interleaved form)
    static constexpr int SPLIT = PLANAR ? 2 * PLANE : IMAGE;                           // byte distance between the images of two splits
    static constexpr int XT_BYTES = 3 * SPLIT;
    static constexpr int DUMP = (PLANAR ? PLANE : IMAGE) - 16;
    static constexpr bool PLAIN = (MODE == CTL_IN_PLAIN);
    int rel[NU];        // byte offset of the unit's first source element relative to the tile's source origin
    int rc[NU];         // r | c << 16 (tile-relative virtual coordinates); 0x7fff7fff past the tile
    int lds[NU];        // LDS byte offset inside plane (0, h); split s adds 2 s PLANE; units past the tile write the dump slot
    // What a tile's loads leave behind until its store: the units themselves and the three words that say how to treat them.  The
    // single-set kernels use the member `pay`; the producer waves of the producer / consumer kernels (ctl_conv_igemm.h, PC) keep several
    // sets -- several tiles of loads in flight -- and pass them explicitly.
    struct Pay {
        f32x4 v0[NU], v1[NU];
        f32x4 w0[X2 ? NU : 1], w1[X2 ? NU : 1];      // X2: the second tensor's units
        unsigned vmask;
        int tb_last;
        bool all_in;
    };
    Pay pay;
    int pad_h, pad_w;

    // (tid_: the thread's index among the 256 that stage this chunk; the multi-group producers of the 1024-thread kernels pass threadIdx.x & 255)
    __device__ __forceinline__ void init(const ctl_conv& d, int tid_ = -1) {
        const int tid = tid_ < 0 ? (int)threadIdx.x : tid_;
#pragma unroll
        for (int i = 0; i < NU; ++i) {
            const int u = tid + i * 256;
            const int pix = PLANAR ? (((u >> 4) << 3) | (u & 7)) : (u >> 1), h = PLANAR ? ((u >> 3) & 1) : (u & 1);
            const int r = pix / G::IW;
            const int c = pix - r * G::IW;
            const bool in = pix < NPIX;
            const int rr = PLAIN ? r : (((r - G::PAD) >> 1) + PADH);
            const int cc = PLAIN ? c : (((c - G::PAD) >> 1) + PADH);
            rel[i] = in ? ((rr * d.win + cc) * d.cin + h * 8) * 4 : CTL_OOB;
            rc[i] = in ? (r | (c << 16)) : 0x7fff7fff;
            lds[i] = in ? (PLANAR ? (h * PLANE + (r * G::IWP + G::ldscol(c)) * 16) : ((r * G::IWP + G::ldscol(c)) * 32 + h * 16)) : DUMP;
        }
        pay.vmask = 0;
        pay.all_in = false;
        pad_h = pad_w = G::PAD;
    }

    __device__ __forceinline__ void load(__amdgpu_buffer_rsrc_t rx, __amdgpu_buffer_rsrc_t rx2, const ctl_conv& d, int n, int ho0, int wo0, int g) {
        load(pay, rx, rx2, d, n, ho0, wo0, g);
    }
    // `live` = false: a step past the block's last one -- the same number of loads, every offset out of range (zeros come back, nothing is
    // fetched): the producer loops issue their loads unconditionally, so that the compiler's wait counts are those of the steady state
    // ONEPATH: no interior-tile shortcut -- one straight-line sequence of loads with per-unit offsets (a dozen vector instructions per
    // unit, which the producer waves have to spare): the two-path form leaves a load-free edge in the structurised control flow, and at a
    // join the wait-count pass then assumes that none of the step's loads is in flight behind the older ones
    template <bool ONEPATH = false>
    __device__ __forceinline__ void load(Pay& P, __amdgpu_buffer_rsrc_t rx, __amdgpu_buffer_rsrc_t rx2, const ctl_conv& d, int n, int ho0, int wo0, int g,
                                         bool live = true) {
        const int vh0 = ho0 * S - pad_h, vw0 = wo0 * S - pad_w;
        const unsigned hv = PLAIN ? d.hin : 2 * d.hin;
        const unsigned wv = PLAIN ? d.win : 2 * d.win;
        const int oh = PLAIN ? vh0 : ((ho0 >> 1) - PADH);
        const int ow = PLAIN ? vw0 : ((wo0 >> 1) - PADH);
        const int tb = (((n * d.hin + oh) * d.win + ow) * d.cin + g * 16) * 4;
        P.tb_last = tb;
        P.all_in = !ONEPATH && live && MODE != CTL_IN_ZINS2 && vh0 >= 0 && vw0 >= 0 && vh0 + G::IH <= (int)hv && vw0 + G::IW <= (int)wv;
        if (!ONEPATH && P.all_in) {       // interior tile: the origin rides in the scalar offset, the per-thread offsets are loop-invariant
#pragma unroll
            for (int i = 0; i < NU; ++i) { P.v0[i] = ctl_bload4s(rx, rel[i], tb); P.v1[i] = ctl_bload4s(rx, rel[i] + 16, tb); }
            if constexpr (X2) {
#pragma unroll
                for (int i = 0; i < NU; ++i) { P.w0[i] = ctl_bload4s(rx2, rel[i], tb); P.w1[i] = ctl_bload4s(rx2, rel[i] + 16, tb); }
            }
            return;
        }
        unsigned m = 0;
        int vo[NU];
#pragma unroll
        for (int i = 0; i < NU; ++i) {
            const int vh = vh0 + (rc[i] & 0xffff), vw = vw0 + (rc[i] >> 16);
            bool ok = live && (unsigned)vh < hv && (unsigned)vw < wv;
            if (MODE == CTL_IN_ZINS2) ok = ok && (((vh | vw) & 1) == 0);
            vo[i] = ok ? (tb + rel[i]) : CTL_OOB;
            m |= ok ? (1u << i) : 0u;
        }
        P.vmask = m;
#pragma unroll
        for (int i = 0; i < NU; ++i) { P.v0[i] = ctl_bload4(rx, vo[i]); P.v1[i] = ctl_bload4(rx, vo[i] == CTL_OOB ? CTL_OOB : vo[i] + 16); }
        if constexpr (X2) {
#pragma unroll
            for (int i = 0; i < NU; ++i) { P.w0[i] = ctl_bload4(rx2, vo[i]); P.w1[i] = ctl_bload4(rx2, vo[i] == CTL_OOB ? CTL_OOB : vo[i] + 16); }
        }
    }

    // pro_scale / pro_shift (/ pro_c with X2) are the block's LDS copies of the coefficients, [group][cin] each; `goff` = group * cin
    __device__ __forceinline__ void store(float* __restrict__ xtf, const ctl_conv& d, int g, const float* pro_scale, const float* pro_shift,
                                          int goff, const float* pro_c, __amdgpu_buffer_rsrc_t rxout, bool xout_on) {
        store(pay, xtf, d, g, pro_scale, pro_shift, goff, pro_c, rxout, xout_on);
    }
    // SEQ: one unit after the other (a scheduling barrier behind each unit's LDS writes): the producer waves keep several payload sets
    // alive, and interleaving the units' prologue / split chains on top of that costs more registers than a wave has
    // PROK: the prologue known at compile time (0 none, 1 BatchNorm + LeakyReLU; -1: ctl_conv.pro_affine decides at run time) -- the
    // software-pipelined kernels need the staging branch-free (one basic block per tile for the instruction scheduler)
    template <bool SEQ = false, int PROK = -1>
    __device__ __forceinline__ void store(const Pay& P, float* __restrict__ xtf, const ctl_conv& d, int g, const float* pro_scale, const float* pro_shift,
                                          int goff, const float* pro_c, __amdgpu_buffer_rsrc_t rxout, bool xout_on) {
        const bool pro_on = PROK < 0 ? (d.pro_affine != 0) : (PROK != 0);
        unsigned char* xt = reinterpret_cast<unsigned char*>(xtf);
        const int cb = g * 16 + (PLANAR ? ((threadIdx.x >> 3) & 1) : (threadIdx.x & 1)) * 8;
        const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
        f32x4 a0 = {1.f, 1.f, 1.f, 1.f}, a1 = a0, b0 = zero, b1 = zero, c0 = zero, c1 = zero;
        if (X2 || pro_on) {
            a0 = *reinterpret_cast<const f32x4*>(pro_scale + goff + cb); a1 = *reinterpret_cast<const f32x4*>(pro_scale + goff + cb + 4);
            b0 = *reinterpret_cast<const f32x4*>(pro_shift + goff + cb); b1 = *reinterpret_cast<const f32x4*>(pro_shift + goff + cb + 4);
        }
        if constexpr (X2) { c0 = *reinterpret_cast<const f32x4*>(pro_c + goff + cb); c1 = *reinterpret_cast<const f32x4*>(pro_c + goff + cb + 4); }
        const float slope = d.pro_slope;
#pragma unroll
        for (int i = 0; i < NU; ++i) {
            f32x4 lo = P.v0[i], hi = P.v1[i];
            const bool in = P.all_in || ((P.vmask >> i) & 1u);
            if constexpr (X2) {      // the virtual tensor A * x + B * x2 + C; padding stays zero (C alone would leak into it)
                lo = in ? (a0 * lo + b0 * P.w0[i] + c0) : zero;
                hi = in ? (a1 * hi + b1 * P.w1[i] + c1) : zero;
                if (xout_on) {
                    const unsigned tr = (unsigned)((rc[i] & 0xffff) - pad_h), tc = (unsigned)((rc[i] >> 16) - pad_w);
                    const bool own = in && tr < (unsigned)(G::TH * S) && tc < (unsigned)(TW * S);
                    ctl_bstore4(rxout, own ? (P.tb_last + rel[i]) : CTL_OOB, lo);
                    ctl_bstore4(rxout, own ? (P.tb_last + rel[i] + 16) : CTL_OOB, hi);
                }
            } else if (pro_on) {      // out-of-range units hold hardware zeros and must stay zero under the affine prologue
                if constexpr (PROK >= 0) {      // (branch-free: as a select the compiler wraps the prologue of every unit into an exec-masked block)
                    lo = x3_keep4(ctl_leaky01(lo * a0 + b0, slope), in);
                    hi = x3_keep4(ctl_leaky01(hi * a1 + b1, slope), in);
                } else {
                    lo = in ? ctl_leaky01(lo * a0 + b0, slope) : zero;
                    hi = in ? ctl_leaky01(hi * a1 + b1, slope) : zero;
                }
            }
            u32x4 ph, pm, pl;
#if defined(CTL_X3_ABLATE) && (CTL_X3_ABLATE & 1)      // (timing ablation, WRONG results: no split arithmetic)
            ph = pm = pl = x3_pack8(lo, hi);
#else
            x3_split8(lo, hi, ph, pm, pl);
#endif
            *reinterpret_cast<u32x4*>(xt + lds[i]) = ph;
            *reinterpret_cast<u32x4*>(xt + lds[i] + SPLIT) = pm;
            *reinterpret_cast<u32x4*>(xt + lds[i] + 2 * SPLIT) = pl;
            if constexpr (SEQ) __builtin_amdgcn_sched_barrier(0);
        }
    }
};
