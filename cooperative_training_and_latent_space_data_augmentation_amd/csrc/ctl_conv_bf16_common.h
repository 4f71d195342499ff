// Shared pieces of the bf16 kernel family (ctl_conv_bf16.hip: convolutions; ctl_wgrad_bf16.hip: weight gradients): vector types, bf16 packing,
// and the LDS staging of the (virtual) input tile.  Two translation units so that the two halves compile in parallel.
#pragma once
#include <stdlib.h>

#include <type_traits>

#include "ctl_conv_common.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned pack_bf16x2(float a, float b) {
    return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{a, b}, bf16x2));
}
__device__ __forceinline__ u32x4 pack_bf16x8(f32x4 lo, f32x4 hi) {
    return u32x4{pack_bf16x2(lo.x, lo.y), pack_bf16x2(lo.z, lo.w), pack_bf16x2(hi.x, hi.y), pack_bf16x2(hi.z, hi.w)};
}
__device__ __forceinline__ f32x4 unpack_bf16x4(unsigned a, unsigned b) {
    return f32x4{__builtin_bit_cast(float, a << 16), __builtin_bit_cast(float, a & 0xffff0000u), __builtin_bit_cast(float, b << 16),
                 __builtin_bit_cast(float, b & 0xffff0000u)};
}
__device__ __forceinline__ u32x4 ctl_bload4u(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
    return __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0);
}
__device__ __forceinline__ u32x2 ctl_bload2u(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
    return __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0);
}

// Staging of one 16-channel chunk of the (virtual) input tile: a unit = 8 channels of a pixel = 16 B of bf16 in LDS.
// PLANAR: the LDS image is two planes [channels 0-7 | channels 8-15] of [row][col][8 ch] = 16 B per pixel and plane.  A B-operand
// read (16 lanes = 16 consecutive pixels x 16 B) then covers 256 contiguous bytes = every bank once; in the interleaved [pixel][16 ch]
// image the same read strides 32 B and hits half the banks twice (2-way conflict on every operand read: measured ~10 of the 17.5 us of
// the 16->16 layer at 256^2).  Both planes start at multiples of 256 B (see PLANE); a staging write is conflict-free because 8 consecutive
// lanes write 8 consecutive pixels of one plane.  The weight-gradient kernel keeps the interleaved image (its transposed reads want pixel rows).
// X2 (pro_affine == 2, the BatchNorm-backward prologue): the operand is the VIRTUAL tensor  A[c] * x + B[c] * x2 + C[c]  of two bf16 tensors
// of one geometry (x = g = dL/da * leaky', x2 = the BatchNorm input u): the `apply` pass of the BatchNorm backward runs here, in the
// staging of its consumers, and its output tensor never exists.  Rounded to bf16 once, exactly where the stored tensor was rounded.
// X16C: what is known at compile time about the source -- 1: stored as bf16 (whole 16-channel chunks); 2: stored as fp32 with a multiple of 4
// channels; 3: fp32 with ONE channel; 0: decided at run time.  Run-time storage flags cost more than branches: with them the compiler keeps
// v0 / v1 in scratch memory and waits for each global load where it is issued (no prefetch left), so every layer shape of the path has
// its compile-time kind and 0 only serves shapes outside it.
template <int KS, int S, int MODE, int MT, int TW, int X16C = 0, bool PLANAR = false, bool X2 = false>
struct XStage16 {
    static_assert(!X2 || X16C == 1, "the two-tensor prologue works on bf16-stored tensors");
    using G = Geom<KS, S, MT, TW>;
    static constexpr int NPIX = G::IH * G::IW;
    static constexpr int UNITS = PLANAR ? ((NPIX + 7) / 8) * 16 : NPIX * 2;      // (planar: unit slots, see init)
    static constexpr int NU = (UNITS + 255) / 256;
    static constexpr int PADH = (G::PAD + 1) >> 1;
    // Round 4: every plane starts at a multiple of 256 B.  A ds_read_b128 is serviced in groups of 8 lanes of lane-row q + 8 lanes of row q ^ 1
    // with complementary pixel sets (MI355X_MICROARCH.md, LDS table); rows q and q ^ 1 read the two planes of one tap, so equal plane phases
    // make a group cover 256 contiguous bytes -- with the former +128 B phase the two halves overlapped: SQ_LDS_BANK_CONFLICT was 0.23-0.40 of
    // SQ_LDS_IDX_ACTIVE on every bf16 conv (profiles/r4_sq_counters_bf16.json), and 0 on the X3 images laid out this way.  The staging writes
    // stay conflict-free through the unit -> thread mapping instead: 8 consecutive lanes write 8 consecutive pixels of ONE plane.
    static constexpr int PLANE = ((G::IH * G::IWP * 16 + 16 + 255) / 256) * 256;
    static constexpr int XT_BYTES = PLANAR ? 2 * PLANE - 16 : G::IH * G::IWP * 32;      // (planar: the dump slot is the last 16 B of the second plane's slack)
    int rel[NU];        // byte offset of the unit's first source element relative to the tile's source origin
    int rc[NU];         // r | c << 16 (tile-relative virtual coordinates); 0x7fff7fff past the tile
    int lds[NU];        // LDS byte offset; units past the tile write a dump slot behind the image
    u32x4 v0[NU], v1[NU];     // source bf16: v0 = 8 channels (X2: v1 = 8 channels of the second tensor); source fp32: v0 = channels 0-3, v1 = 4-7 of the unit
    unsigned vmask;
    int pad_h, pad_w;
    int tb_last;        // X2: byte offset of the tile held in v0 / v1 (side output of the virtual tensor, see store)
    bool all_in, x16;

    __device__ __forceinline__ void init(const ctl_conv& d) {
        const int tid = threadIdx.x, h = PLANAR ? ((tid >> 3) & 1) : (tid & 1);
        x16 = X16C == 1 || (X16C == 0 && (d.dt & CTL_DT_X16) != 0);
        const int esz = x16 ? 2 : 4;
#pragma unroll
        for (int i = 0; i < NU; ++i) {
            const int u = tid + i * 256;
            const int pix = PLANAR ? (((u >> 4) << 3) | (u & 7)) : (u >> 1);
            const int r = pix / G::IW;
            const int c = pix - r * G::IW;
            const bool in = pix < NPIX;
            const int rr = (MODE == CTL_IN_PLAIN) ? r : (((r - G::PAD) >> 1) + PADH);
            const int cc = (MODE == CTL_IN_PLAIN) ? c : (((c - G::PAD) >> 1) + PADH);
            rel[i] = in ? ((rr * d.win + cc) * d.cin + h * 8) * esz : CTL_OOB;
            rc[i] = in ? (r | (c << 16)) : 0x7fff7fff;
            lds[i] = in ? (PLANAR ? (h * PLANE + (r * G::IWP + G::ldscol(c)) * 16) : ((r * G::IWP + G::ldscol(c)) * 32 + h * 16)) : XT_BYTES;
        }
        vmask = 0;
        all_in = false;
        pad_h = pad_w = G::PAD;
    }

    __device__ __forceinline__ void load(__amdgpu_buffer_rsrc_t rx, const ctl_conv& d, int n, int ho0, int wo0, int g) { load(rx, rx, d, n, ho0, wo0, g); }
    __device__ __forceinline__ void load(__amdgpu_buffer_rsrc_t rx, __amdgpu_buffer_rsrc_t rx2, const ctl_conv& d, int n, int ho0, int wo0, int g) {
        const int vh0 = ho0 * S - pad_h, vw0 = wo0 * S - pad_w;
        const unsigned hv = (MODE == CTL_IN_PLAIN) ? d.hin : 2 * d.hin;
        const unsigned wv = (MODE == CTL_IN_PLAIN) ? d.win : 2 * d.win;
        const int oh = (MODE == CTL_IN_PLAIN) ? vh0 : ((ho0 >> 1) - PADH);
        const int ow = (MODE == CTL_IN_PLAIN) ? vw0 : ((wo0 >> 1) - PADH);
        const int esz = x16 ? 2 : 4;
        const int tb = (((n * d.hin + oh) * d.win + ow) * d.cin + g * 16) * esz;
        tb_last = tb;
        all_in = MODE != CTL_IN_ZINS2 && vh0 >= 0 && vw0 >= 0 && vh0 + G::IH <= (int)hv && vw0 + G::IW <= (int)wv &&
                 g * 16 + 16 <= d.cin;
        if (all_in) {
            if (X2) {
#pragma unroll
                for (int i = 0; i < NU; ++i) { v0[i] = ctl_bload4u(rx, rel[i], tb); v1[i] = ctl_bload4u(rx2, rel[i], tb); }
            } else if (X16C == 1 || (X16C == 0 && x16)) {
#pragma unroll
                for (int i = 0; i < NU; ++i) v0[i] = ctl_bload4u(rx, rel[i], tb);
            } else {
#pragma unroll
                for (int i = 0; i < NU; ++i) { v0[i] = ctl_bload4u(rx, rel[i], tb); v1[i] = ctl_bload4u(rx, rel[i] + 16, tb); }
            }
            return;
        }
        const int cb = g * 16 + (PLANAR ? ((threadIdx.x >> 3) & 1) : (threadIdx.x & 1)) * 8;          // first channel of this thread's units
        unsigned m = 0;
        int vo[NU];
#pragma unroll
        for (int i = 0; i < NU; ++i) {
            const int vh = vh0 + (rc[i] & 0xffff), vw = vw0 + (rc[i] >> 16);
            bool ok = cb < d.cin && (unsigned)vh < hv && (unsigned)vw < wv;
            if (MODE == CTL_IN_ZINS2) ok = ok && (((vh | vw) & 1) == 0);
            vo[i] = ok ? (tb + rel[i]) : CTL_OOB;
            m |= ok ? (1u << i) : 0u;
        }
        vmask = m;
        const u32x4 z = {0u, 0u, 0u, 0u};
        if (X2) {
#pragma unroll
            for (int i = 0; i < NU; ++i) { v0[i] = ctl_bload4u(rx, vo[i], 0); v1[i] = ctl_bload4u(rx2, vo[i], 0); }
        } else if (X16C == 1 || (X16C == 0 && x16)) {      // internal tensors: cin is a multiple of 16 (checked on the host)
#pragma unroll
            for (int i = 0; i < NU; ++i) v0[i] = ctl_bload4u(rx, vo[i], 0);
        } else if (X16C == 2 || (X16C == 0 && d.cin >= 4)) {      // fp32 source: quads of 4 channels, the second one may lie past cin (4 or 12 channels)
            const bool q1 = cb + 4 < d.cin;
#pragma unroll
            for (int i = 0; i < NU; ++i) { v0[i] = ctl_bload4u(rx, vo[i], 0); v1[i] = q1 ? ctl_bload4u(rx, vo[i] + 16, 0) : z; }
        } else {                                    // one input channel
#pragma unroll
            for (int i = 0; i < NU; ++i) {
                v0[i] = u32x4{__builtin_amdgcn_raw_buffer_load_b32(rx, vo[i], 0, 0), 0u, 0u, 0u};
                v1[i] = z;
            }
        }
    }

    // X2: cf_scale / cf_shift / cf_c hold A / B / C of the block's groups ([group][cin] each)
    // rxout / xout_on (X2): the tile's interior units are also written to a tensor of x's geometry -- the virtual tensor materialises as a
    // by-product of this staging for the weight-gradient kernel of the same layer (see ctl_conv.hip)
    __device__ __forceinline__ void store(unsigned char* __restrict__ xt, const ctl_conv& d, int g, const float* cf_scale,
                                          const float* cf_shift, int goff, const float* cf_c = nullptr) {
        store(xt, d, g, cf_scale, cf_shift, goff, cf_c, ctl_rsrc((const void*)nullptr, 0), false);
    }
    __device__ __forceinline__ void store(unsigned char* __restrict__ xt, const ctl_conv& d, int g, const float* cf_scale,
                                          const float* cf_shift, int goff, const float* cf_c,
                                          __amdgpu_buffer_rsrc_t rxout, bool xout_on) {
        if constexpr (X2) {
            const int cb = g * 16 + (PLANAR ? ((threadIdx.x >> 3) & 1) : (threadIdx.x & 1)) * 8;
            const f32x4 a0 = *reinterpret_cast<const f32x4*>(cf_scale + goff + cb), a1 = *reinterpret_cast<const f32x4*>(cf_scale + goff + cb + 4);
            const f32x4 b0 = *reinterpret_cast<const f32x4*>(cf_shift + goff + cb), b1 = *reinterpret_cast<const f32x4*>(cf_shift + goff + cb + 4);
            const f32x4 c0 = *reinterpret_cast<const f32x4*>(cf_c + goff + cb), c1 = *reinterpret_cast<const f32x4*>(cf_c + goff + cb + 4);
            const u32x4 zero = {0u, 0u, 0u, 0u};
#pragma unroll
            for (int i = 0; i < NU; ++i) {
                const f32x4 glo = unpack_bf16x4(v0[i].x, v0[i].y), ghi = unpack_bf16x4(v0[i].z, v0[i].w);
                const f32x4 ulo = unpack_bf16x4(v1[i].x, v1[i].y), uhi = unpack_bf16x4(v1[i].z, v1[i].w);
                u32x4 pk = pack_bf16x8(a0 * glo + b0 * ulo + c0, a1 * ghi + b1 * uhi + c1);
                const bool in = all_in || ((vmask >> i) & 1u);
                if (!in) pk = zero;      // padding stays zero (C alone would leak into it)
                *reinterpret_cast<u32x4*>(xt + lds[i]) = pk;
                if (xout_on) {
                    const unsigned tr = (unsigned)((rc[i] & 0xffff) - pad_h), tc = (unsigned)((rc[i] >> 16) - pad_w);
                    const bool own = in && tr < (unsigned)(G::TH * S) && tc < (unsigned)(TW * S);
                    __builtin_amdgcn_raw_buffer_store_b128(pk, rxout, own ? (tb_last + rel[i]) : CTL_OOB, 0, 0);
                }
            }
            return;
        }
        if (!d.pro_affine && (X16C == 1 || (X16C == 0 && x16))) {      // bf16 in, nothing to compute: out-of-range units were loaded as hardware zeros
#pragma unroll
            for (int i = 0; i < NU; ++i) *reinterpret_cast<u32x4*>(xt + lds[i]) = v0[i];
            return;
        }
        const int cb = g * 16 + (PLANAR ? ((threadIdx.x >> 3) & 1) : (threadIdx.x & 1)) * 8;
        f32x4 sc0 = {1.f, 1.f, 1.f, 1.f}, sh0 = {0.f, 0.f, 0.f, 0.f}, sc1 = sc0, sh1 = sh0;
        if (d.pro_affine && cb < d.cin) {
            if (d.cin >= 4) {
                sc0 = *reinterpret_cast<const f32x4*>(cf_scale + goff + cb);
                sh0 = *reinterpret_cast<const f32x4*>(cf_shift + goff + cb);
                if (cb + 4 < d.cin) {
                    sc1 = *reinterpret_cast<const f32x4*>(cf_scale + goff + cb + 4);
                    sh1 = *reinterpret_cast<const f32x4*>(cf_shift + goff + cb + 4);
                }
            } else { sc0.x = cf_scale[goff]; sh0.x = cf_shift[goff]; }
        }
        const float slope = d.pro_slope;
        const u32x4 zero = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int i = 0; i < NU; ++i) {
            f32x4 lo, hi;
            if (X16C == 1 || (X16C == 0 && x16)) { lo = unpack_bf16x4(v0[i].x, v0[i].y); hi = unpack_bf16x4(v0[i].z, v0[i].w); }
            else { lo = __builtin_bit_cast(f32x4, v0[i]); hi = __builtin_bit_cast(f32x4, v1[i]); }
            if (d.pro_affine) { lo = ctl_leaky01(lo * sc0 + sh0, slope); hi = ctl_leaky01(hi * sc1 + sh1, slope); }
            u32x4 pk = pack_bf16x8(lo, hi);
            // padding / channel-pad lanes hold hardware zeros and must stay zero under the affine prologue
            if (d.pro_affine && !all_in && !((vmask >> i) & 1u)) pk = zero;
            *reinterpret_cast<u32x4*>(xt + lds[i]) = pk;
        }
    }
};
