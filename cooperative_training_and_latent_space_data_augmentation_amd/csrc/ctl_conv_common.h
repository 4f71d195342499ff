// Device helpers shared by the fp32 (ctl_conv.hip) and bf16 (ctl_conv_bf16.hip) implicit-GEMM convolution kernels: tile geometry,
// raw buffer loads / stores with hardware bounds checking, the persistent blocks' tile walker.
#pragma once
#include "ctl_common.h"

template <int KS, int S, int MT, int TW>
struct Geom {
    static constexpr int MTILES = 4 * MT;            // 16-pixel M-tiles per 256-thread block
    static constexpr int TH = MTILES * 16 / TW;      // output tile height
    static constexpr int TP = TH * TW;               // output pixels per tile
    static constexpr int IH = (TH - 1) * S + KS;     // input tile (virtual coordinates)
    static constexpr int IW = (TW - 1) * S + KS;
    static constexpr int IWH = (IW + 1) / 2;
    static constexpr int IWP = (S == 2) ? 2 * IWH : IW;
    static constexpr int XT_IMAGE = IH * IWP * 16;
    static constexpr int XT_FLOATS = XT_IMAGE + 4;   // + one 16-byte dump slot for the staging units past the tile
    static constexpr int PAD = (KS >= 3) ? 1 : 0;     // 3x3 and the 4x4 stride-2 form of a pooled 3x3 data gradient: pad 1
    __device__ static __forceinline__ int ldscol(int c) { return (S == 2) ? ((c & 1) * IWH + (c >> 1)) : c; }
};

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
// Byte offset that is out of range for every tensor (all are < 2 GiB, checked on the host): a buffer load from it returns 0
// and a buffer store to it is dropped by the hardware bounds check -> zero padding / ragged edges cost no branch and no select.
#define CTL_OOB ((int)0x80000000)
// prologue coefficients (BatchNorm scale / shift per [group][cin]) are copied to LDS once per block: groups * cin <= CTL_PRO_MAX
#define CTL_PRO_MAX 256

__device__ __forceinline__ __amdgpu_buffer_rsrc_t ctl_rsrc(const void* p, int64_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ f32x4 ctl_bload4(__amdgpu_buffer_rsrc_t r, int voff) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, 0, 0));
}
__device__ __forceinline__ float ctl_bload1(__amdgpu_buffer_rsrc_t r, int voff) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, 0, 0));
}
#ifndef CTL_STORE_AUX
#define CTL_STORE_AUX 0      // cache-policy bits of the epilogue stores (experiment hook: sc0 = 1, nt = 2, sc1 = 16 on gfx94x/95x)
#endif
__device__ __forceinline__ void ctl_bstore4(__amdgpu_buffer_rsrc_t r, int voff, f32x4 v) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, voff, 0, CTL_STORE_AUX);
}
__device__ __forceinline__ void ctl_bstore1(__amdgpu_buffer_rsrc_t r, int voff, float v) {
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned int, v), r, voff, 0, 0);
}
// Load with a wave-uniform byte offset in an SGPR: per-thread offsets stay loop-invariant VGPRs, the tile origin costs no VALU.
// LOADS ONLY.  A buffer_store_dwordx4 with an SGPR soffset followed directly by a VALU write of its data VGPRs stores
// garbage in the late-read lanes on gfx950 (measured: lanes 12-15 of every 16, second dword), and the compiler inserts the
// required wait state only when soffset is NOT a register -> stores always carry the full offset in the VGPR.
__device__ __forceinline__ f32x4 ctl_bload4s(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}
// LeakyReLU for 0 <= slope <= 1 (checked on the host) as max(v, v*slope): two VALU ops, no compare+select
__device__ __forceinline__ f32x4 ctl_leaky01(f32x4 v, float slope) {
    const f32x4 m = v * slope;
    return f32x4{fmaxf(v.x, m.x), fmaxf(v.y, m.y), fmaxf(v.z, m.z), fmaxf(v.w, m.w)};
}

// Tile walker of a persistent block: tile index bid0, bid0+nblk, ... decoded incrementally (no per-tile divisions).
struct TileWalk {
    int n, th, tw;            // current tile coordinates
    int dn, dth, dtw;         // decomposition of the stride nblk
    int tiles_h, tiles_w;
    __device__ __forceinline__ void init(int bid0, int nblk, int tiles_h_, int tiles_w_) {
        tiles_h = tiles_h_; tiles_w = tiles_w_;
        tw = bid0 % tiles_w; int b = bid0 / tiles_w; th = b % tiles_h; n = b / tiles_h;
        dtw = nblk % tiles_w; b = nblk / tiles_w; dth = b % tiles_h; dn = b / tiles_h;
    }
    __device__ __forceinline__ void next() {
        tw += dtw;
        if (tw >= tiles_w) { tw -= tiles_w; ++th; }
        th += dth;
        if (th >= tiles_h) { th -= tiles_h; ++n; }
        n += dn;
    }
};

