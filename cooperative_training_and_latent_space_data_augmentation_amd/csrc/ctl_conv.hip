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
    static constexpr int XT_FLOATS = IH * IWP * 16;
    static constexpr int PAD = (KS == 3) ? 1 : 0;
    __device__ static __forceinline__ int ldscol(int c) { return (S == 2) ? ((c & 1) * IWH + (c >> 1)) : c; }
};

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
// Byte offset that is out of range for every tensor (all are < 2 GiB, checked on the host): a buffer load from it returns 0
// and a buffer store to it is dropped by the hardware bounds check -> zero padding / ragged edges cost no branch and no select.
#define CTL_OOB ((int)0x80000000)

__device__ __forceinline__ __amdgpu_buffer_rsrc_t ctl_rsrc(const void* p, int64_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ f32x4 ctl_bload4(__amdgpu_buffer_rsrc_t r, int voff) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, 0, 0));
}
__device__ __forceinline__ float ctl_bload1(__amdgpu_buffer_rsrc_t r, int voff) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, 0, 0));
}
__device__ __forceinline__ void ctl_bstore4(__amdgpu_buffer_rsrc_t r, int voff, f32x4 v) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, voff, 0, 0);
}
__device__ __forceinline__ void ctl_bstore1(__amdgpu_buffer_rsrc_t r, int voff, float v) {
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned int, v), r, voff, 0, 0);
}

// Staging of one 16-channel chunk of the (virtual) input tile into LDS, with the BN+LeakyReLU prologue.  Everything that
// depends only on the thread (tile-relative coordinates, source byte offset, LDS offset) is computed ONCE (init); per tile a
// unit costs two adds + two unsigned compares (bounds) + one select for the buffer offset.  LOAD issues all buffer loads of
// the tile back to back (nothing consumes them), STORE (after the MFMA phase) applies the prologue and writes LDS.
template <int KS, int S, int MODE, int MT, int TW>
struct XStage {
    using G = Geom<KS, S, MT, TW>;
    static constexpr int UNITS = G::IH * G::IW * 4;
    static constexpr int NU = (UNITS + 255) / 256;
    int rel[NU];        // byte offset of the unit relative to the tile's source origin
    int rc[NU];         // r | c << 16 (tile-relative virtual coordinates); 0x7fff7fff for the units past the tile
    int lds[NU];        // LDS float offset, -1 for the units past the tile
    f32x4 v[NU];
    unsigned vmask;     // bit i: unit i of the tile held in v[] lies inside the image (gets the prologue)

    __device__ __forceinline__ void init(const ctl_conv& d) {
        const int tid = threadIdx.x, cq = tid & 3;
#pragma unroll
        for (int i = 0; i < NU; ++i) {
            const int u = tid + i * 256;
            const int pix = u >> 2;
            const int r = pix / G::IW;
            const int c = pix - r * G::IW;
            const bool in = u < UNITS;
            // source = virtual for plain inputs; for x2 nearest / zero-insert inputs the tile origin is even, so
            // (origin - PAD + r) >> 1 = origin/2 + ((r - PAD) >> 1)
            const int rr = (MODE == CTL_IN_PLAIN) ? r : ((r - G::PAD) >> 1);
            const int cc = (MODE == CTL_IN_PLAIN) ? c : ((c - G::PAD) >> 1);
            rel[i] = ((rr * d.win + cc) * d.cin + cq * 4) * 4;
            rc[i] = in ? (r | (c << 16)) : 0x7fff7fff;
            lds[i] = in ? ((r * G::IWP + G::ldscol(c)) * 16 + cq * 4) : -1;
        }
        vmask = 0;
    }

    __device__ __forceinline__ void load(__amdgpu_buffer_rsrc_t rx, const ctl_conv& d, int n, int ho0, int wo0, int g) {
        const int vh0 = ho0 * S - G::PAD, vw0 = wo0 * S - G::PAD;
        const unsigned hv = (MODE == CTL_IN_PLAIN) ? d.hin : 2 * d.hin;
        const unsigned wv = (MODE == CTL_IN_PLAIN) ? d.win : 2 * d.win;
        const int oh = (MODE == CTL_IN_PLAIN) ? vh0 : (ho0 >> 1);
        const int ow = (MODE == CTL_IN_PLAIN) ? vw0 : (wo0 >> 1);
        const int tb = (((n * d.hin + oh) * d.win + ow) * d.cin + g * 16) * 4;      // uniform; may be negative at the border
        const bool chan_ok = g * 16 + (threadIdx.x & 3) * 4 < d.cin;
        unsigned m = 0;
        int vo[NU];
#pragma unroll
        for (int i = 0; i < NU; ++i) {
            const int vh = vh0 + (rc[i] & 0xffff), vw = vw0 + (rc[i] >> 16);
            bool ok = chan_ok && (unsigned)vh < hv && (unsigned)vw < wv;
            if (MODE == CTL_IN_ZINS2) ok = ok && (((vh | vw) & 1) == 0);
            vo[i] = ok ? (tb + rel[i]) : CTL_OOB;
            m |= ok ? (1u << i) : 0u;
        }
        vmask = m;
        if (d.cin >= 4) {
#pragma unroll
            for (int i = 0; i < NU; ++i) v[i] = ctl_bload4(rx, vo[i]);
        } else {
#pragma unroll
            for (int i = 0; i < NU; ++i) v[i] = f32x4{ctl_bload1(rx, vo[i]), 0.f, 0.f, 0.f};
        }
    }

    __device__ __forceinline__ void store(float* __restrict__ xt, const ctl_conv& d, int g,
                                          const float* __restrict__ pro_scale, const float* __restrict__ pro_shift) {
        const int cb = g * 16 + (threadIdx.x & 3) * 4;
        const bool pro = d.pro_affine != 0;
        f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
        if (pro && cb < d.cin) {
            if (d.cin >= 4) {
                sc = *reinterpret_cast<const f32x4*>(pro_scale + cb);
                sh = *reinterpret_cast<const f32x4*>(pro_shift + cb);
            } else {
                sc.x = pro_scale[0];
                sh.x = pro_shift[0];
            }
        }
        const float slope = d.pro_slope;
#pragma unroll
        for (int i = 0; i < NU; ++i) {
            f32x4 t = v[i];
            if (pro && ((vmask >> i) & 1u)) {          // padding / channel-pad lanes hold hardware zeros: no prologue
                t.x = ctl_leaky(t.x * sc.x + sh.x, slope);
                if (d.cin >= 4) {
                    t.y = ctl_leaky(t.y * sc.y + sh.y, slope);
                    t.z = ctl_leaky(t.z * sc.z + sh.z, slope);
                    t.w = ctl_leaky(t.w * sc.w + sh.w, slope);
                }
            }
            if (lds[i] >= 0) *reinterpret_cast<f32x4*>(xt + lds[i]) = t;
        }
    }
};

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

// ------------------------------------------------------------------------------------------------ forward-type kernel
// Persistent, software-pipelined: a block walks tiles bid, bid+grid, ... and for every (tile, 16-channel chunk) step
//   1. issues the NEXT step's buffer loads (input tile + weight chunk) into registers        -- HBM/L2 latency in flight
//   2. runs the MFMA loop of the CURRENT step out of LDS
//   3. bare s_barrier; ds_write the prefetched registers; lgkmcnt(0) + s_barrier               -- no vmcnt drain
//   4. (last chunk of a tile) epilogue: bias / residual / activation, buffer stores that are never waited for in the loop;
//      BatchNorm statistics stay in registers until the block is done.
// Weight chunks [tap][nt][64 lanes][4] go through LDS (shared by the four waves; staged once when Cin <= 16).
template <int KS, int S, int MODE, int MT, int TW, int NT, int EPI>
__global__ __launch_bounds__(256, (MT * NT >= 8) ? 2 : ((MT * NT >= 4) ? 3 : 4)) void conv_igemm_kernel(const ctl_conv d, const float* __restrict__ x,
                                                          const float* __restrict__ wpack,
                                                          const float* __restrict__ bias,
                                                          const float* __restrict__ pro_scale,
                                                          const float* __restrict__ pro_shift,
                                                          const float* __restrict__ res,
                                                          const float* __restrict__ res_scale,
                                                          const float* __restrict__ res_shift, float* __restrict__ y,
                                                          float* __restrict__ stats_partial, int tiles_h, int tiles_w,
                                                          int G_chunks, int64_t wpack_sub_stride, int ntiles, int dbg) {
    using G = Geom<KS, S, MT, TW>;
    constexpr int TAPS = KS * KS;
    constexpr int RED_FLOATS = 4 * NT * 16 * 2;
    constexpr int WT_FLOATS = TAPS * NT * 256;
    constexpr int XT_ALLOC = (G::XT_FLOATS > RED_FLOATS) ? G::XT_FLOATS : RED_FLOATS;
    __shared__ __attribute__((aligned(16))) float xt[XT_ALLOC + WT_FLOATS];
    float* wt = xt + XT_ALLOC;
    constexpr int WU = TAPS * NT * 64, NW = (WU + 255) / 256;

    if (dbg & 8) return;                       // ablation: pure launch + dispatch cost
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int p = lane & 15, q = lane >> 4;
    const int nblk = gridDim.x;
    const int bid0 = ctl_xcd_remap(blockIdx.x, nblk);
    const int z = blockIdx.z;
    const int cot0 = blockIdx.y * NT;
    const float* wp = wpack + (int64_t)z * wpack_sub_stride;
    const int my_tiles = (bid0 < ntiles) ? (ntiles - bid0 + nblk - 1) / nblk : 0;
    const int total_it = my_tiles * G_chunks;
    const int flags = d.epi_flags;
    const int oy0 = (z >> 1) * d.out_sub, ox0 = (z & 1) * d.out_sub;
    const __amdgpu_buffer_rsrc_t rx = ctl_rsrc(x, (int64_t)d.n * d.hin * d.win * d.cin * 4);
    const int64_t ybytes = (int64_t)d.n * d.out_h * d.out_w * d.cout * 4;
    const __amdgpu_buffer_rsrc_t ry = ctl_rsrc(y, ybytes);
    const __amdgpu_buffer_rsrc_t rres = ctl_rsrc(EPI ? (const void*)res : (const void*)y, ybytes);

    f32x4 ssum[NT], ssq[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) ssum[t] = ssq[t] = f32x4{0.f, 0.f, 0.f, 0.f};

    // per-thread constants of the epilogue: byte offset of this lane's 4 channels of M-tile m relative to the tile's output origin
    int yrel[MT], wcol[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        const int mt = wave * MT + m;
        const int tr = mt / (TW / 16), tc = (mt % (TW / 16)) * 16;
        wcol[m] = tc + p;
        yrel[m] = ((tr * d.out_sy * d.out_w + (tc + p) * d.out_sx) * d.cout + q * 4) * 4;
    }

    XStage<KS, S, MODE, MT, TW> xs;
    xs.init(d);
    f32x4 wv[NW];
    auto wload = [&](int g) {
#pragma unroll
        for (int i = 0; i < NW; ++i) {
            const int u = tid + i * 256;
            const int tt = u >> 6, l = u & 63;           // tt = tap * NT + t
            const int tap = tt / NT, t = tt - tap * NT;
            if (u < WU)
                wv[i] = *reinterpret_cast<const f32x4*>(
                    wp + ((((int64_t)(cot0 + t) * TAPS + tap) * G_chunks + g) * 64 + l) * 4);
        }
    };
    auto wstore = [&]() {
#pragma unroll
        for (int i = 0; i < NW; ++i) {
            const int u = tid + i * 256;
            if (u < WU) *reinterpret_cast<f32x4*>(wt + u * 4) = wv[i];
        }
    };

    TileWalk cur, nxt;
    cur.init(bid0, nblk, tiles_h, tiles_w);
    nxt = cur;
    ctl_stagger_sleep(dbg >> 8);
    if (total_it > 0) {
        xs.load(rx, d, cur.n, cur.th * G::TH, cur.tw * TW, 0);
        wload(0);
        xs.store(xt, d, 0, pro_scale, pro_shift);
        wstore();
    }
    __syncthreads();
    if (dbg & 16) return;                      // ablation: + init + first-tile staging
    if (dbg & 32) {                            // ablation: + the loop skeleton (barriers, tile walk), no stats epilogue
        for (int it = 0; it < total_it; ++it) { ctl_barrier_lds_reads_done(); ctl_barrier_lds_writes_done(); }
        return;
    }

    f32x4 acc[MT][NT];
    for (int it = 0, g = 0; it < total_it; ++it) {
        const int n = cur.n, ho0 = cur.th * G::TH, wo0 = cur.tw * TW;
        const bool has_next = it + 1 < total_it;
        const int g2 = (g + 1 == G_chunks) ? 0 : g + 1;
        const bool new_w = has_next && G_chunks > 1;
        if (g2 == 0) nxt.next();
        if (has_next && !(dbg & 2)) {
            xs.load(rx, d, nxt.n, nxt.th * G::TH, nxt.tw * TW, g2);
            if (new_w) wload(g2);
        }
        if (g == 0) {
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int t = 0; t < NT; ++t) acc[m][t] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        if (!(dbg & 1))
#pragma unroll
        for (int tap = 0; tap < TAPS; ++tap) {
            const int kh = tap / KS, kw = tap % KS;
            f32x4 wf[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) wf[t] = *reinterpret_cast<const f32x4*>(wt + ((tap * NT + t) * 64 + lane) * 4);
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                const int mt = wave * MT + m;
                const int tr = mt / (TW / 16), tc = (mt % (TW / 16)) * 16;
                const int r = tr * S + kh;
                const int c = (tc + p) * S + kw;
                f32x4 xf;
                if ((dbg & 64) && m > 0) xf = wf[0];      // ablation: no LDS read for M-tiles 1.. (wrong results, timing only)
                else xf = *reinterpret_cast<const f32x4*>(xt + (r * G::IWP + G::ldscol(c)) * 16 + q * 4);
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    acc[m][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[t].x, xf.x, acc[m][t], 0, 0, 0);
                    acc[m][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[t].y, xf.y, acc[m][t], 0, 0, 0);
                    acc[m][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[t].z, xf.z, acc[m][t], 0, 0, 0);
                    acc[m][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[t].w, xf.w, acc[m][t], 0, 0, 0);
                }
            }
        }

        if (!(dbg & 128)) ctl_barrier_lds_reads_done();    // every wave is done reading this step's LDS images
        if (has_next && !(dbg & 2)) {    // refill LDS from the prefetched registers
            xs.store(xt, d, g2, pro_scale, pro_shift);
            if (new_w) wstore();
        }
        if (!(dbg & 128)) ctl_barrier_lds_writes_done();

        if (g == G_chunks - 1 && !(dbg & 4)) {
            // ---------------- epilogue: lane (p,q) holds channels co0..co0+3 of pixel p of each M-tile.  Buffer stores with
            // hardware bounds checks: ragged pixels / padded channels get CTL_OOB and are dropped; nothing here is waited for.
            const int ybase = (((n * d.out_h + ho0 * d.out_sy + oy0) * d.out_w + wo0 * d.out_sx + ox0) * d.cout + cot0 * 16) * 4;
            bool pv[MT];
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                const int tr = (wave * MT + m) / (TW / 16);
                pv[m] = (ho0 + tr < d.hout) && (wo0 + wcol[m] < d.wout);
            }
            f32x4 rv[EPI ? MT : 1][EPI ? NT : 1], ov[EPI ? MT : 1][EPI ? NT : 1];
            if (EPI) {
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const bool cok = (cot0 + t) * 16 + q * 4 < d.cout;
#pragma unroll
                    for (int m = 0; m < MT; ++m) {
                        const int vo = (pv[m] && cok) ? (ybase + yrel[m] + t * 64) : CTL_OOB;
                        rv[m][t] = ov[m][t] = f32x4{0.f, 0.f, 0.f, 0.f};
                        if (d.cout >= 4) {
                            if (flags & CTL_EPI_RES) rv[m][t] = ctl_bload4(rres, vo);
                            if (flags & CTL_EPI_ACCUM) ov[m][t] = ctl_bload4(ry, vo);
                        } else {
                            if (flags & CTL_EPI_RES) rv[m][t].x = ctl_bload1(rres, vo);
                            if (flags & CTL_EPI_ACCUM) ov[m][t].x = ctl_bload1(ry, vo);
                        }
                    }
                }
            }
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const int co0 = (cot0 + t) * 16 + q * 4;
                const bool cok = co0 < d.cout;
                const int cc = cok ? co0 : 0;
                f32x4 b = {0.f, 0.f, 0.f, 0.f}, rs = {0.f, 0.f, 0.f, 0.f}, rh = {0.f, 0.f, 0.f, 0.f};
                if (d.cout >= 4) {
                    if (flags & CTL_EPI_BIAS) b = *reinterpret_cast<const f32x4*>(bias + cc);
                    if (EPI && (flags & CTL_EPI_RES)) {
                        rs = *reinterpret_cast<const f32x4*>(res_scale + cc);
                        rh = *reinterpret_cast<const f32x4*>(res_shift + cc);
                    }
                } else {
                    if (flags & CTL_EPI_BIAS) b.x = bias[0];
                    if (EPI && (flags & CTL_EPI_RES)) { rs.x = res_scale[0]; rh.x = res_shift[0]; }
                }
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                    f32x4 v = acc[m][t];
                    v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w;
                    if (EPI) {
                        const f32x4 r_ = rv[m][t];
                        v.x += r_.x * rs.x + rh.x; v.y += r_.y * rs.y + rh.y;
                        v.z += r_.z * rs.z + rh.z; v.w += r_.w * rs.w + rh.w;
                    }
                    if ((flags & CTL_EPI_STATS) && pv[m]) {
                        ssum[t].x += v.x; ssum[t].y += v.y; ssum[t].z += v.z; ssum[t].w += v.w;
                        ssq[t].x += v.x * v.x; ssq[t].y += v.y * v.y; ssq[t].z += v.z * v.z; ssq[t].w += v.w * v.w;
                    }
                    if (d.epi_act == CTL_ACT_LEAKY) {
                        v.x = ctl_leaky(v.x, d.epi_slope); v.y = ctl_leaky(v.y, d.epi_slope);
                        v.z = ctl_leaky(v.z, d.epi_slope); v.w = ctl_leaky(v.w, d.epi_slope);
                    } else if (d.epi_act == CTL_ACT_SIGMOID) {
                        v.x = 1.f / (1.f + expf(-v.x)); v.y = 1.f / (1.f + expf(-v.y));
                        v.z = 1.f / (1.f + expf(-v.z)); v.w = 1.f / (1.f + expf(-v.w));
                    }
                    if (EPI) {
                        const f32x4 o = ov[m][t];
                        v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
                    }
                    const int vo = (pv[m] && cok) ? (ybase + yrel[m] + t * 64) : CTL_OOB;
                    if (d.cout >= 4) ctl_bstore4(ry, vo, v);
                    else ctl_bstore1(ry, vo, v.x);       // cout == 1: only q == 0 passes `cok`, component x is the channel
                }
            }
        }
        if (g2 == 0) cur = nxt;
        g = g2;
    }

    __syncthreads();
    if (flags & CTL_EPI_STATS) {  // per-channel sum / sum of squares of this block's tiles -> stats_partial[block][2][cout]
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            float a[8] = {ssum[t].x, ssum[t].y, ssum[t].z, ssum[t].w, ssq[t].x, ssq[t].y, ssq[t].z, ssq[t].w};
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                float v = a[i];
                v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); v += __shfl_xor(v, 4); v += __shfl_xor(v, 8);
                a[i] = v;
            }
            if (p == 0) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    xt[((wave * NT + t) * 16 + q * 4 + r) * 2 + 0] = a[r];
                    xt[((wave * NT + t) * 16 + q * 4 + r) * 2 + 1] = a[4 + r];
                }
            }
        }
        __syncthreads();
        if (tid < NT * 16 * 2) {
            const int stat = tid / (NT * 16), cl = tid % (NT * 16);
            float v = 0.f;
#pragma unroll
            for (int w = 0; w < 4; ++w) v += xt[((w * NT * 16) + cl) * 2 + stat];
            const int co = cot0 * 16 + cl;
            if (co < d.cout) stats_partial[((int64_t)blockIdx.x * 2 + stat) * d.cout + co] = v;
        }
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
template <int KS, int S, int MODE, int MT, int TW, int NTW>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(const ctl_conv d, const float* __restrict__ x,
                                                          const float* __restrict__ pro_scale,
                                                          const float* __restrict__ pro_shift,
                                                          const float* __restrict__ dy, float* __restrict__ w_partial,
                                                          float* __restrict__ b_partial, int tiles_h, int tiles_w,
                                                          int ntiles, int cin_p, int cout_p) {
    using G = Geom<KS, S, MT, TW>;
    constexpr int TAPS = KS * KS;
    constexpr int DYT_FLOATS = NTW * G::TP * 16;
    constexpr int RED_FLOATS = 4 * NTW * 256;
    constexpr int LDS_FLOATS = (G::XT_FLOATS + DYT_FLOATS > RED_FLOATS) ? (G::XT_FLOATS + DYT_FLOATS) : RED_FLOATS;
    __shared__ __attribute__((aligned(16))) float lds[LDS_FLOATS];
    float* xt = lds;
    float* dyt = lds + G::XT_FLOATS;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int p = lane & 15, q = lane >> 4;
    const int g = blockIdx.y;
    const int cot0 = blockIdx.z * NTW;

    f32x4 acc[TAPS][NTW];
#pragma unroll
    for (int a = 0; a < TAPS; ++a)
#pragma unroll
        for (int t = 0; t < NTW; ++t) acc[a][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    float bsum[NTW];
#pragma unroll
    for (int t = 0; t < NTW; ++t) bsum[t] = 0.f;

    constexpr int DU = G::TP * NTW * 4, ND = DU / 256;       // TP is a multiple of 64 -> exact
    const __amdgpu_buffer_rsrc_t rx = ctl_rsrc(x, (int64_t)d.n * d.hin * d.win * d.cin * 4);
    const __amdgpu_buffer_rsrc_t rdy = ctl_rsrc(dy, (int64_t)d.n * d.hout * d.wout * d.cout * 4);
    XStage<KS, S, MODE, MT, TW> xs;
    xs.init(d);
    // dy tile: per-thread constants (pixel row/col inside the tile, byte offset relative to the tile origin, LDS offset)
    f32x4 dv[ND];
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
        drel[i] = ((pr * d.wout + pc) * d.cout + co) * 4;
        dlds[i] = (t * G::TP + pix) * 16 + cq * 4;
    }
    auto dyload = [&](int n, int ho0, int wo0) {
        const int tb = ((n * d.hout + ho0) * d.wout + wo0) * d.cout * 4;
#pragma unroll
        for (int i = 0; i < ND; ++i) {
            const bool ok = (unsigned)(ho0 + (drc[i] & 0xffff)) < (unsigned)d.hout && (unsigned)(wo0 + (drc[i] >> 16)) < (unsigned)d.wout;
            const int vo = ok ? (tb + drel[i]) : CTL_OOB;
            if (d.cout >= 4) dv[i] = ctl_bload4(rdy, vo);
            else dv[i] = f32x4{ctl_bload1(rdy, vo), 0.f, 0.f, 0.f};
        }
    };
    auto dystore = [&]() {
#pragma unroll
        for (int i = 0; i < ND; ++i) *reinterpret_cast<f32x4*>(dyt + dlds[i]) = dv[i];
    };
    TileWalk cur;
    cur.init(blockIdx.x, gridDim.x, tiles_h, tiles_w);
    if ((int)blockIdx.x < ntiles) {
        xs.load(rx, d, cur.n, cur.th * G::TH, cur.tw * TW, g);
        dyload(cur.n, cur.th * G::TH, cur.tw * TW);
        xs.store(xt, d, g, pro_scale, pro_shift);
        dystore();
    }
    __syncthreads();
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const bool has_next = tile + (int)gridDim.x < ntiles;
        if (has_next) {                   // next tile's loads fly while this tile's MFMAs run
            cur.next();
            xs.load(rx, d, cur.n, cur.th * G::TH, cur.tw * TW, g);
            dyload(cur.n, cur.th * G::TH, cur.tw * TW);
        }
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
                for (int tap = 0; tap < TAPS; ++tap) {
                    const int kh = tap / KS, kw = tap % KS;
                    const float af = xt[((tr * S + kh) * G::IWP + G::ldscol(pc * S + kw)) * 16 + p];
#pragma unroll
                    for (int t = 0; t < NTW; ++t)
                        acc[tap][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(af, bf[t], acc[tap][t], 0, 0, 0);
                }
            }
        }
        ctl_barrier_lds_reads_done();
        if (has_next) {
            xs.store(xt, d, g, pro_scale, pro_shift);
            dystore();
        }
        ctl_barrier_lds_writes_done();
    }
    __syncthreads();

    // ---------------- sum the four waves through LDS, tap by tap, and write this split's partial
    float* red = lds;
    const int64_t split_base = (int64_t)blockIdx.x * TAPS * cin_p * cout_p;
#pragma unroll
    for (int tap = 0; tap < TAPS; ++tap) {
        __syncthreads();
#pragma unroll
        for (int t = 0; t < NTW; ++t) {
            red[((wave * NTW + t) * 4 + 0) * 64 + lane] = acc[tap][t].x;
            red[((wave * NTW + t) * 4 + 1) * 64 + lane] = acc[tap][t].y;
            red[((wave * NTW + t) * 4 + 2) * 64 + lane] = acc[tap][t].z;
            red[((wave * NTW + t) * 4 + 3) * 64 + lane] = acc[tap][t].w;
        }
        __syncthreads();
        for (int e = tid; e < NTW * 256; e += 256) {
            const int t = e >> 8, r = (e >> 6) & 3, l = e & 63;
            float v = 0.f;
#pragma unroll
            for (int w = 0; w < 4; ++w) v += red[((w * NTW + t) * 4 + r) * 64 + l];
            const int ci = g * 16 + (l >> 4) * 4 + r;
            const int co = (cot0 + t) * 16 + (l & 15);
            if (co < cout_p) w_partial[split_base + ((int64_t)tap * cin_p + ci) * cout_p + co] = v;
        }
    }
    if (g == 0 && b_partial != nullptr) {  // bias gradient: column sums of dy
        __syncthreads();
#pragma unroll
        for (int t = 0; t < NTW; ++t) {
            float v = bsum[t];
            v += __shfl_xor(v, 16);
            v += __shfl_xor(v, 32);
            if (q == 0) red[(wave * NTW + t) * 16 + p] = v;
        }
        __syncthreads();
        if (tid < NTW * 16) {
            float v = 0.f;
#pragma unroll
            for (int w = 0; w < 4; ++w) v += red[w * NTW * 16 + tid];
            const int co = cot0 * 16 + tid;
            if (co < cout_p) b_partial[(int64_t)blockIdx.x * cout_p + co] = v;
        }
    }
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
    if (idx >= total) return;
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
    if (flip) { kh = ks - 1 - kh; kw = ks - 1 - kw; }
    float v = 0.f;
    if (co < cout && ci < cin) v = src[co * r[6] + ci * r[7] + kh * r[8] + kw * r[9]];
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
        for (int s = sl; s < splits; s += SLN) v += src[s * stride];
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
    if (k == 3 && s == 1 && pd == 1) return true;
    if (k == 3 && s == 2 && pd == 1 && m == CTL_IN_PLAIN) return true;
    if (k == 1 && s == 1 && pd == 0 && m != CTL_IN_ZINS2) return true;
    if (k == 2 && s == 2 && pd == 0 && m == CTL_IN_PLAIN) return true;
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
    } else if (d->wout >= 32 && blocks(4, 32) >= 512) {
        mt = 4; tw = 32;
    } else if (blocks(2, 16) >= 384) {
        mt = 2; tw = 16;
    } else {
        mt = 1; tw = 16;
    }
    {   // tuning hook (tools/bench_conv.py): CTL_FORCE_CFG="mt,tw,nt" overrides the heuristic when the combination is valid
        const char* f = getenv("CTL_FORCE_CFG");
        int fm, ft, fn;
        if (f && sscanf(f, "%d,%d,%d", &fm, &ft, &fn) == 3) {
            const bool tile_ok = (fm == 4 && ft == 32 && d->stride == 1) || (fm == 2 && ft == 16) || (fm == 1 && ft == 16);
            if (tile_ok) { mt = fm; tw = ft; }
            if ((fn == 1 || fn == 2 || fn == 4) && c->cot % fn == 0) c->nt = fn;
        }
    }
    if (d->stride == 2 && c->nt == 4) c->nt = 2;            // stride-2 input tiles are 4x larger: keep LDS < 64 KiB
    c->mt = mt; c->tw = tw; c->th = 4 * mt * 16 / tw;
    c->tiles_h = ctl_cdiv(d->hout, c->th);
    c->tiles_w = ctl_cdiv(d->wout, c->tw);
    return CTL_OK;
}

// persistent grid: ~CTL_PERSIST blocks per CU in total (default 4; the blocks of one launch share the chip with
// nothing else), never more than there are tiles
static int conv_grid_x(const ctl_conv* d, const ctl_conv_cfg* c) {
    static int per_cu = -1;
    if (per_cu < 0) {
        const char* e = getenv("CTL_PERSIST");
        per_cu = e ? atoi(e) : 4;
        if (per_cu < 1) per_cu = 1;
    }
    const int ntiles = d->n * c->tiles_h * c->tiles_w;
    const int other = (c->cot / c->nt) * d->nsub;
    int cap = ctl_cdiv(256 * per_cu, other);
    if (cap < 1) cap = 1;
    return ntiles < cap ? ntiles : cap;
}

extern "C" int ctl_conv_stats_blocks(const ctl_conv* d) {
    ctl_conv_cfg c;
    if (ctl_conv_pick_cfg(d, &c, 0) != CTL_OK) return -1;
    return conv_grid_x(d, &c);
}
extern "C" size_t ctl_conv_stats_floats(const ctl_conv* d) {
    const int b = ctl_conv_stats_blocks(d);
    return b < 0 ? 0 : (size_t)b * 2 * d->cout;
}

extern "C" int ctl_pack_weights(const float* src, float* dst, int32_t cout, int32_t cin, int32_t ks, int64_t s_co,
                                int64_t s_ci, int64_t s_kh, int64_t s_kw, int32_t flip, ctl_stream stream) {
    CTL_REQUIRE(src && dst && cout > 0 && cin > 0 && ks >= 1 && ks <= 3, "pack_weights: bad arguments");
    const int64_t total = (int64_t)ctl_conv_wpack_floats(cin, cout, ks);
    pack_weights_kernel<<<dim3((unsigned)ctl_cdiv64(total, 256)), dim3(256), 0, (hipStream_t)stream>>>(
        src, dst, cout, cin, ks, ctl_cdiv(cin, 16), total, s_co, s_ci, s_kh, s_kw, flip);
    CTL_LAUNCH_CHECK("pack_weights");
    return CTL_OK;
}

#define CONV_ARGS *d, x, wpack, bias, pro_scale, pro_shift, res, res_scale, res_shift, y, stats_partial, c.tiles_h, \
                  c.tiles_w, c.g, sub_stride, ntiles, dbg
#define LAUNCH_CONV(KS, S, MODE, MT, TW, NT)                                                                     \
    do {                                                                                                         \
        if (epi) conv_igemm_kernel<KS, S, MODE, MT, TW, NT, 1><<<grid, dim3(256), 0, (hipStream_t)stream>>>(CONV_ARGS); \
        else conv_igemm_kernel<KS, S, MODE, MT, TW, NT, 0><<<grid, dim3(256), 0, (hipStream_t)stream>>>(CONV_ARGS);     \
    } while (0)
#define DISPATCH_NT(KS, S, MODE, MT, TW)                     \
    do {                                                     \
        if (c.nt == 4) LAUNCH_CONV(KS, S, MODE, MT, TW, 4);  \
        else if (c.nt == 2) LAUNCH_CONV(KS, S, MODE, MT, TW, 2); \
        else LAUNCH_CONV(KS, S, MODE, MT, TW, 1);            \
    } while (0)
#define DISPATCH_TILE(KS, S, MODE)                                   \
    do {                                                             \
        if (c.mt == 4 && c.tw == 32) DISPATCH_NT(KS, S, MODE, 4, 32); \
        else if (c.mt == 2) DISPATCH_NT(KS, S, MODE, 2, 16);         \
        else DISPATCH_NT(KS, S, MODE, 1, 16);                        \
    } while (0)
#define DISPATCH_NT_S2(KS, S, MODE, MT, TW)                  \
    do {                                                     \
        if (c.nt == 2) LAUNCH_CONV(KS, S, MODE, MT, TW, 2);  \
        else LAUNCH_CONV(KS, S, MODE, MT, TW, 1);            \
    } while (0)
#define DISPATCH_TILE_S2(KS, S, MODE)                        \
    do {                                                     \
        if (c.mt == 2) DISPATCH_NT_S2(KS, S, MODE, 2, 16);   \
        else DISPATCH_NT_S2(KS, S, MODE, 1, 16);             \
    } while (0)

extern "C" int ctl_conv_forward(const ctl_conv* d, const float* x, const float* wpack, const float* bias,
                                const float* pro_scale, const float* pro_shift, const float* res,
                                const float* res_scale, const float* res_shift, float* y, float* stats_partial,
                                ctl_stream stream) {
    CTL_REQUIRE(d && x && wpack && y, "conv_forward: null argument");
    ctl_conv_cfg c;
    int rc = ctl_conv_pick_cfg(d, &c, 0);
    if (rc != CTL_OK) return rc;
    CTL_REQUIRE(!(d->epi_flags & CTL_EPI_BIAS) || bias, "conv_forward: CTL_EPI_BIAS without bias");
    CTL_REQUIRE(!(d->epi_flags & CTL_EPI_RES) || (res && res_scale && res_shift), "conv_forward: CTL_EPI_RES without res");
    CTL_REQUIRE(!(d->epi_flags & CTL_EPI_STATS) || (stats_partial && d->nsub == 1), "conv_forward: bad CTL_EPI_STATS use");
    CTL_REQUIRE(!d->pro_affine || (pro_scale && pro_shift), "conv_forward: prologue without scale/shift");
    CTL_REQUIRE(d->n > 0 && d->hout > 0 && d->wout > 0, "conv_forward: empty problem");
    CTL_REQUIRE((int64_t)d->n * d->hin * d->win * d->cin * 4 < (1ll << 31) &&
                (int64_t)d->n * d->out_h * d->out_w * d->cout * 4 < (1ll << 31),
                "conv_forward: tensors must stay below 2 GiB (32-bit buffer offsets)");
    const int64_t sub_stride = (int64_t)ctl_conv_wpack_floats(d->cin, d->cout, d->ks);
    const int ntiles = d->n * c.tiles_h * c.tiles_w;
    const bool epi = (d->epi_flags & (CTL_EPI_RES | CTL_EPI_ACCUM)) != 0;
    static int dbg = -1;                       // CTL_DBG ablation mask (tools/bench_conv.py): 1 no MFMA, 2 no prefetch, 4 no epilogue
    if (dbg < 0) { const char* e = getenv("CTL_DBG"); dbg = e ? atoi(e) : 0; }
    const dim3 grid((unsigned)conv_grid_x(d, &c), (unsigned)(c.cot / c.nt), (unsigned)d->nsub);
    const int k = d->ks, s = d->stride, m = d->in_mode;
    const int ptok = ctl_prof_begin("conv_igemm", d, &c, c.nt, (hipStream_t)stream);
    if (k == 3 && s == 1 && m == CTL_IN_PLAIN) DISPATCH_TILE(3, 1, CTL_IN_PLAIN);
    else if (k == 3 && s == 1 && m == CTL_IN_UP2) DISPATCH_TILE(3, 1, CTL_IN_UP2);
    else if (k == 3 && s == 1 && m == CTL_IN_ZINS2) DISPATCH_TILE(3, 1, CTL_IN_ZINS2);
    else if (k == 3 && s == 2) DISPATCH_TILE_S2(3, 2, CTL_IN_PLAIN);
    else if (k == 1 && m == CTL_IN_PLAIN) DISPATCH_TILE(1, 1, CTL_IN_PLAIN);
    else if (k == 1 && m == CTL_IN_UP2) DISPATCH_TILE(1, 1, CTL_IN_UP2);
    else if (k == 2 && s == 2) DISPATCH_TILE_S2(2, 2, CTL_IN_PLAIN);
    else CTL_FAIL(CTL_EUNSUPPORTED, "conv_forward: no kernel for this combination");
    ctl_prof_end(ptok, (hipStream_t)stream);
    CTL_LAUNCH_CHECK("conv_forward");
    return CTL_OK;
}

// ---- wgrad host side
struct wgrad_cfg { ctl_conv_cfg c; int ntw, splits, ntiles, cin_p, cout_p; };

static int wgrad_pick(const ctl_conv* d, wgrad_cfg* w) {
    CTL_REQUIRE(d->nsub == 1, "wgrad: nsub must be 1");
    CTL_REQUIRE(d->in_mode != CTL_IN_ZINS2, "wgrad: zero-insert input is not a forward mode");
    int rc = ctl_conv_pick_cfg(d, &w->c, 1);
    if (rc != CTL_OK) return rc;
    // the wgrad kernel reuses the forward tile shapes; cap LDS by keeping (4,32) only for stride 1
    w->ntw = (w->c.cot >= 2 && w->c.cot % 2 == 0) ? 2 : 1;
    w->ntiles = d->n * w->c.tiles_h * w->c.tiles_w;
    w->cin_p = w->c.g * 16;
    w->cout_p = w->c.cot * 16;
    const int par = w->c.g * (w->c.cot / w->ntw);
    int splits = ctl_cdiv(1024, par);          // ~4 blocks per CU in total; every block then walks >= a few tiles
    if (splits > 512) splits = 512;
    if (splits > w->ntiles) splits = w->ntiles;
    if (splits < 1) splits = 1;
    w->splits = splits;
    return CTL_OK;
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

#define WGRAD_ARGS *d, x, pro_scale, pro_shift, dy, w_partial, b_partial, w.c.tiles_h, w.c.tiles_w, w.ntiles, w.cin_p, w.cout_p
#define LAUNCH_WG(KS, S, MODE, MT, TW, NTW)                                                              \
    conv_wgrad_kernel<KS, S, MODE, MT, TW, NTW><<<grid, dim3(256), 0, (hipStream_t)stream>>>(WGRAD_ARGS)
#define WG_NT(KS, S, MODE, MT, TW)                       \
    do {                                                 \
        if (w.ntw == 2) LAUNCH_WG(KS, S, MODE, MT, TW, 2); \
        else LAUNCH_WG(KS, S, MODE, MT, TW, 1);          \
    } while (0)
#define WG_TILE(KS, S, MODE)                                      \
    do {                                                          \
        if (w.c.mt == 4 && w.c.tw == 32) WG_NT(KS, S, MODE, 4, 32); \
        else if (w.c.mt == 2) WG_NT(KS, S, MODE, 2, 16);          \
        else WG_NT(KS, S, MODE, 1, 16);                           \
    } while (0)
#define WG_TILE_S2(KS, S, MODE)                         \
    do {                                                \
        if (w.c.mt == 2) WG_NT(KS, S, MODE, 2, 16);     \
        else WG_NT(KS, S, MODE, 1, 16);                 \
    } while (0)

extern "C" int ctl_conv_wgrad(const ctl_conv* d, const float* x, const float* pro_scale, const float* pro_shift,
                              const float* dy, float* w_partial, float* b_partial, ctl_stream stream) {
    CTL_REQUIRE(d && x && dy && w_partial, "conv_wgrad: null argument");
    CTL_REQUIRE(!d->pro_affine || (pro_scale && pro_shift), "conv_wgrad: prologue without scale/shift");
    CTL_REQUIRE((int64_t)d->n * d->hin * d->win * d->cin * 4 < (1ll << 31) &&
                (int64_t)d->n * d->hout * d->wout * d->cout * 4 < (1ll << 31),
                "conv_wgrad: tensors must stay below 2 GiB (32-bit buffer offsets)");
    wgrad_cfg w;
    int rc = wgrad_pick(d, &w);
    if (rc != CTL_OK) return rc;
    const dim3 grid((unsigned)w.splits, (unsigned)w.c.g, (unsigned)(w.c.cot / w.ntw));
    const int k = d->ks, s = d->stride, m = d->in_mode;
    const int ptok = ctl_prof_begin("conv_wgrad", d, &w.c, w.ntw, (hipStream_t)stream);
    if (k == 3 && s == 1 && m == CTL_IN_PLAIN) WG_TILE(3, 1, CTL_IN_PLAIN);
    else if (k == 3 && s == 1 && m == CTL_IN_UP2) WG_TILE(3, 1, CTL_IN_UP2);
    else if (k == 3 && s == 2) WG_TILE_S2(3, 2, CTL_IN_PLAIN);
    else if (k == 1 && m == CTL_IN_PLAIN) WG_TILE(1, 1, CTL_IN_PLAIN);
    else if (k == 1 && m == CTL_IN_UP2) WG_TILE(1, 1, CTL_IN_UP2);
    else if (k == 2 && s == 2) WG_TILE_S2(2, 2, CTL_IN_PLAIN);
    else CTL_FAIL(CTL_EUNSUPPORTED, "conv_wgrad: no kernel for this combination");
    ctl_prof_end(ptok, (hipStream_t)stream);
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
