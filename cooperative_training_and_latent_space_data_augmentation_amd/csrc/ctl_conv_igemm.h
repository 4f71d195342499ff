// The forward-type implicit-GEMM convolution kernel of the fp32-storage family and the staging of its input tile: shared by
// ctl_conv.hip (fp32 MFMA, v_mfma_f32_16x16x4_f32) and ctl_conv_x3.hip (the same fp32 tensors contracted on the bf16 matrix pipe by an
// exact three-way split of every operand, X3 below).  Two translation units so that the two halves compile in parallel.
#pragma once
#include <stdlib.h>

#include <type_traits>

#include "ctl_common.h"

#include "ctl_conv_common.h"
#include "ctl_conv_x3_stage.h"

#ifndef TM_DECL      // (phase timers: variant builds of ctl_conv.hip only)
#define TM_DECL
#define TM(i)
#define TM_COUNT(i)
#define TM_FLUSH
#endif

// Staging of one 16-channel chunk of the (virtual) input tile into LDS, with the BN+LeakyReLU prologue.  Everything that
// depends only on the thread (tile-relative coordinates, source byte offset, LDS offset) is computed ONCE (init); per tile a
// unit costs two adds + two unsigned compares (bounds) + one select for the buffer offset.  LOAD issues all buffer loads of
// the tile back to back (nothing consumes them), STORE (after the MFMA phase) applies the prologue and writes LDS.
// CTL_IN_C4: plain stored input with <= 4 channels whose 3x3 taps are K-packed (see conv_igemm_kernel)
#define CTL_MODE_IS_PLAIN(M) ((M) == CTL_IN_PLAIN || (M) == CTL_IN_C4)

// X2 (ctl_conv.pro_affine == 2, the BatchNorm-backward prologue): the operand is the VIRTUAL tensor  A[c] * x + B[c] * x2 + C[c]  of two
// tensors of one geometry (x = g = dL/da * leaky', x2 = the BatchNorm input): the `apply` pass of the BatchNorm backward runs here, in
// the staging of its consumers, and its output tensor never exists (two packed fmas per element on top of the second load).
template <int KS, int S, int MODE, int MT, int TW, bool X2 = false>
struct XStage {
    using G = Geom<KS, S, MT, TW>;
    // C4M (round 4): inputs with <= 4 channels keep a DENSE image, [row][col][4 floats] = 16 B per pixel, one staging unit per pixel.  In
    // the 64-B-per-pixel image of the other modes only every fourth quad was used: 3 of 4 staging threads idled and the 16 lanes of an
    // operand read hit four banks' worth of addresses four times over (SQ_LDS_BANK_CONFLICT = 0.53 of the LDS-active cycles on the
    // first-layer convs, profiles/r4_sq_counters_fp32.json).
    static constexpr bool C4M = (MODE == CTL_IN_C4);
    static constexpr int UNITS = G::IH * G::IW * (C4M ? 1 : 4);
    static constexpr int NU = (UNITS + 255) / 256;
    static constexpr int PADH = (G::PAD + 1) >> 1;   // source-space padding of the x2 modes
    int rel[NU];        // byte offset of the unit relative to the tile's source origin
    int rc[NU];         // r | c << 16 (tile-relative virtual coordinates); 0x7fff7fff for the units past the tile
    int lds[NU];        // LDS float offset; the units past the tile write a dump slot behind the image
    f32x4 v[NU];
    f32x4 v2[X2 ? NU : 1];      // X2: the second tensor's units
    unsigned vmask;     // bit i: unit i of the tile held in v[] lies inside the image (gets the prologue)
    int pad_h, pad_w;   // top / left padding of this block's problem (G::PAD except for the phase problems of the 2x2 kernels)
    bool all_in;        // wave-uniform: every unit of the tile held in v[] is inside the image and the channel range
    int tb_last;        // X2: byte offset of the tile held in v[] (for the side output of the virtual tensor, see store)

    __device__ __forceinline__ void init(const ctl_conv& d) {
        const int tid = threadIdx.x, cq = C4M ? 0 : (tid & 3);
#pragma unroll
        for (int i = 0; i < NU; ++i) {
            const int u = tid + i * 256;
            const int pix = C4M ? u : (u >> 2);
            const int r = pix / G::IW;
            const int c = pix - r * G::IW;
            const bool in = u < UNITS;
            // source = virtual for plain inputs; for x2 nearest / zero-insert inputs the tile origin is even, so
            // (origin - PAD + r) >> 1 = origin/2 + ((r - PAD) >> 1); the source origin is moved up/left by PADH so that
            // the per-thread offsets are never negative (they are unsigned voffsets next to a scalar tile offset)
            const int rr = CTL_MODE_IS_PLAIN(MODE) ? r : (((r - G::PAD) >> 1) + PADH);
            const int cc = CTL_MODE_IS_PLAIN(MODE) ? c : (((c - G::PAD) >> 1) + PADH);
            rel[i] = in ? ((rr * d.win + cc) * d.cin + cq * 4) * 4 : CTL_OOB;
            rc[i] = in ? (r | (c << 16)) : 0x7fff7fff;
            lds[i] = in ? (C4M ? (r * G::IWP + c) * 4 : ((r * G::IWP + G::ldscol(c)) * 16 + cq * 4)) : G::XT_IMAGE;
        }
        vmask = 0;
        all_in = false;
        pad_h = pad_w = G::PAD;
    }

    __device__ __forceinline__ void load(__amdgpu_buffer_rsrc_t rx, const ctl_conv& d, int n, int ho0, int wo0, int g) { load(rx, rx, d, n, ho0, wo0, g); }
    __device__ __forceinline__ void load(__amdgpu_buffer_rsrc_t rx, __amdgpu_buffer_rsrc_t rx2, const ctl_conv& d, int n, int ho0, int wo0, int g) {
        const int vh0 = ho0 * S - pad_h, vw0 = wo0 * S - pad_w;
        const unsigned hv = CTL_MODE_IS_PLAIN(MODE) ? d.hin : 2 * d.hin;
        const unsigned wv = CTL_MODE_IS_PLAIN(MODE) ? d.win : 2 * d.win;
        const int oh = CTL_MODE_IS_PLAIN(MODE) ? vh0 : ((ho0 >> 1) - PADH);
        const int ow = CTL_MODE_IS_PLAIN(MODE) ? vw0 : ((wo0 >> 1) - PADH);
        const int tb = (((n * d.hin + oh) * d.win + ow) * d.cin + g * 16) * 4;      // uniform; may be negative at the border
        tb_last = tb;
        // interior tile (most of them): every unit is in range -> the tile origin goes into the scalar offset of the buffer
        // loads and the per-thread offsets are the loop-invariant rel[]: no VALU at all (fp32 MFMA shares the VALU issue
        // port on this part, so every VALU instruction in the loop is time taken from the matrix work)
#ifdef CTL_NO_INTERIOR
        all_in = false;
#else
        all_in = MODE != CTL_IN_ZINS2 && vh0 >= 0 && vw0 >= 0 && vh0 + G::IH <= (int)hv && vw0 + G::IW <= (int)wv &&
                 g * 16 + 16 <= d.cin;
#endif
        if (all_in) {
#pragma unroll
            for (int i = 0; i < NU; ++i) v[i] = ctl_bload4s(rx, rel[i], tb);
            if constexpr (X2) {
#pragma unroll
                for (int i = 0; i < NU; ++i) v2[i] = ctl_bload4s(rx2, rel[i], tb);
            }
            return;
        }
        const bool chan_ok = C4M || g * 16 + (threadIdx.x & 3) * 4 < d.cin;
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
        if constexpr (X2) {
#pragma unroll
            for (int i = 0; i < NU; ++i) { v[i] = ctl_bload4(rx, vo[i]); v2[i] = ctl_bload4(rx2, vo[i]); }
        } else if (d.cin >= 4) {
#pragma unroll
            for (int i = 0; i < NU; ++i) v[i] = ctl_bload4(rx, vo[i]);
        } else {
#pragma unroll
            for (int i = 0; i < NU; ++i) v[i] = f32x4{ctl_bload1(rx, vo[i]), 0.f, 0.f, 0.f};
        }
    }

    // `goff` = BatchNorm group of the tile held in v[] times cin (row of the [groups][cin] prologue coefficients).  The
    // coefficients are read from the block's LDS copy: a global load here sits between the two barriers of a step with nothing
    // to hide its latency behind
    // X2: pro_scale / pro_shift / pro_c are the LDS copies of A / B / C ([group][cin] each).  xout (a buffer resource over a tensor of x's
    // geometry, `xout_on` on the blocks of the first cout group only): the units of the tile's INTERIOR -- the input pixels no other tile
    // owns -- are also written to global memory: the virtual tensor materialises as a by-product of this conv's staging, for the weight-
    // gradient kernel of the same layer (which would otherwise evaluate it again in each of its cin-chunk blocks).
    __device__ __forceinline__ void store(float* __restrict__ xt, const ctl_conv& d, int g,
                                          const float* pro_scale, const float* pro_shift, int goff, const float* pro_c = nullptr) {
        store(xt, d, g, pro_scale, pro_shift, goff, pro_c, ctl_rsrc((const void*)nullptr, 0), false);
    }
    __device__ __forceinline__ void store(float* __restrict__ xt, const ctl_conv& d, int g,
                                          const float* pro_scale, const float* pro_shift, int goff, const float* pro_c,
                                          __amdgpu_buffer_rsrc_t rxout, bool xout_on) {
        if constexpr (X2) {
            const int cb = g * 16 + (threadIdx.x & 3) * 4;
            const f32x4 ca = *reinterpret_cast<const f32x4*>(pro_scale + goff + cb), cbb = *reinterpret_cast<const f32x4*>(pro_shift + goff + cb);
            const f32x4 cc = *reinterpret_cast<const f32x4*>(pro_c + goff + cb);
            const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int i = 0; i < NU; ++i) {
                const bool in = all_in || ((vmask >> i) & 1u);
                const f32x4 r = in ? (ca * v[i] + cbb * v2[i] + cc) : zero;      // padding stays zero (C alone would leak into it)
                *reinterpret_cast<f32x4*>(xt + lds[i]) = r;
                if (xout_on) {
                    const unsigned tr = (unsigned)((rc[i] & 0xffff) - pad_h), tc = (unsigned)((rc[i] >> 16) - pad_w);
                    const bool own = in && tr < (unsigned)(G::TH * S) && tc < (unsigned)(TW * S);
                    ctl_bstore4(rxout, own ? (tb_last + rel[i]) : CTL_OOB, r);
                }
            }
            return;
        }
        if (!d.pro_affine) {      // out-of-range units were loaded as hardware zeros: nothing to compute
#pragma unroll
            for (int i = 0; i < NU; ++i) *reinterpret_cast<f32x4*>(xt + lds[i]) = v[i];
            return;
        }
        const int cb = C4M ? 0 : g * 16 + (threadIdx.x & 3) * 4;
        f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
        if (cb < d.cin) {
            if (d.cin >= 4) {
                sc = *reinterpret_cast<const f32x4*>(pro_scale + goff + cb);
                sh = *reinterpret_cast<const f32x4*>(pro_shift + goff + cb);
            } else {
                sc.x = pro_scale[goff];
                sh.x = pro_shift[goff];
            }
        }
        const float slope = d.pro_slope;
        if (all_in) {
#pragma unroll
            for (int i = 0; i < NU; ++i) *reinterpret_cast<f32x4*>(xt + lds[i]) = ctl_leaky01(v[i] * sc + sh, slope);
            return;
        }
        const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < NU; ++i) {
            const f32x4 t = ctl_leaky01(v[i] * sc + sh, slope);
            // padding / channel-pad lanes hold hardware zeros and must stay zero; units past the tile go to the dump slot
            *reinterpret_cast<f32x4*>(xt + lds[i]) = ((vmask >> i) & 1u) ? t : zero;
        }
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

#ifndef CTL_LB_MID
#define CTL_LB_MID 3
#endif
#ifndef CTL_LB_SMALL
#define CTL_LB_SMALL 3      // (4: 28-76 B of scratch in the epilogue-operand forms of this class; 16.61 -> 16.57 ms)
#endif
// X3: the contraction runs on v_mfma_f32_16x16x32_bf16 over an exact three-way bf16 split of both fp32 operands (ctl_conv_x3_stage.h):
// the LDS images, the weight fragments and the matrix phase change, everything around them (tile walk, prefetch pipeline, epilogues) is shared
// PC (round 5, X3 only): the PRODUCER / CONSUMER form.  One 512-thread block per CU: waves 0-3 (one per SIMD) are producers -- they keep
// CTL_PC_SETS tiles of buffer loads in flight in registers (they hold no accumulators), apply the prologue, split and write the bf16
// planes of step it into LDS image it & 1 (and the step's weight fragments into weight image it & 1); waves 4-7 (the SIMDs' second
// waves) are consumers -- operand reads, MFMAs and the epilogue (whose operands they request before the step's barrier).  ONE s_barrier
// per (tile, chunk) step: producers arrive with image it & 1 written, consumers with every read of image (it - 1) & 1 consumed, so the
// staging of step it + 1 runs beside the matrix phase of step it on every SIMD (the matrix pipe and the vector ALU issue from
// different waves), and a load has CTL_PC_SETS - 1 steps to come back.  Statistics rows are per consumer WAVE (no block-level sum: a
// barrier among the consumers alone does not exist on gfx950), so ctl_conv_x3_stats_blocks reports 4 rows per block.
#ifndef CTL_PC_SETS
#define CTL_PC_SETS 3
#endif
#ifndef CTL_PC_SETS_X2
#define CTL_PC_SETS_X2 2
#endif
template <int KS, int S, int MODE, int MT, int TW, int NT, int EPI, bool X2 = false, bool X3 = false, bool PC = false>      // X2: see XStage (pro_scale = the [group][3][cin] coefficients)
// resident blocks of the X2 instantiations: two staged tensors (and with EPI an epilogue tensor) in registers -- at 3 blocks per CU the
// 8x32-pixel form needs 92 B of scratch per lane, at 2 none: 17.15 -> 16.98 ms per step (tools/debug/fp32_x2_occ_ab.sh)
#ifndef CTL_LB_X2EPI
#define CTL_LB_X2EPI 2
#endif
#ifndef CTL_LB_X2
#define CTL_LB_X2 2      // (X2 without an epilogue operand: 36 B of scratch at 3; 17.17 -> 17.12 ms)
#endif
#ifndef CTL_LB_SMALL_X2EPI
#define CTL_LB_SMALL_X2EPI CTL_LB_SMALL
#endif
// X3: the issue port, not the matrix pipe, is what a SIMD runs out of (per tile and wave ~120 MFMAs x 8 issue cycles + ~300 vector
// instructions of staging / split / epilogue): the more waves a SIMD can choose from, the better the two streams interleave.  One cout
// tile per block (the 16-channel layers at full resolution) fits three resident blocks per CU at 168 registers with one M-tile per
// operand step; the two-cout-tile forms need 2 x 3 x 4 more weight-operand registers and stay at two.
#ifndef CTL_LB_X3
#define CTL_LB_X3(MT, NT, X2) (((NT) == 1 && (MT) >= 2 && !(X2)) ? 3 : 2)
#endif
__global__ __launch_bounds__(PC ? 512 : 256, PC ? 2 : X3 ? CTL_LB_X3(MT, NT, X2) : (MT * NT >= 8 || KS == 4) ? 2 : ((MT * NT >= 4) ? (X2 ? (EPI ? CTL_LB_X2EPI : CTL_LB_X2) : CTL_LB_MID)
                                                                                    : ((X2 && EPI) ? CTL_LB_SMALL_X2EPI : CTL_LB_SMALL))) void conv_igemm_kernel(const ctl_conv d, const float* __restrict__ x,
                                                          const float* __restrict__ wpack,
                                                          const float* __restrict__ bias,
                                                          const float* __restrict__ pro_scale,
                                                          const float* __restrict__ pro_shift,
                                                          const float* __restrict__ res,
                                                          const float* __restrict__ res_scale,
                                                          const float* __restrict__ res_shift, float* __restrict__ y,
                                                          float* __restrict__ stats_partial, int tiles_h, int tiles_w,
                                                          int G_chunks, int64_t wpack_sub_stride, int ntiles,
                                                          const float* __restrict__ res2, const float* __restrict__ x2,
                                                          float* __restrict__ pool, float* __restrict__ xout) {
    using G = Geom<KS, S, MT, TW>;
    constexpr int TAPS = KS * KS;
    // C4: input with <= 4 channels.  The four lane groups of an MFMA (its k index) carry four different TAPS (channels 0-3 each)
    // instead of four channel groups of one tap: 3 fragments ("quads" of taps 0-3, 4-7, 8) cover the 3x3 kernel, so a pixel tile
    // costs 12 MFMAs instead of 36.  Only the per-lane LDS read base and the weight fragment order differ from the plain path.
    constexpr bool C4 = (MODE == CTL_IN_C4);
    static_assert(!C4 || (KS == 3 && S == 1), "K-packed taps: 3x3 stride-1 only");
    static_assert(!X3 || !C4, "X3: whole 16-channel chunks");
    static_assert(!PC || X3, "the producer / consumer form exists for the X3 launches");
    constexpr int NIMG = PC ? 2 : 1;                                         // LDS images of the input tile and of the weight chunk
    using XS3 = XStage3<KS, S, C4 ? CTL_IN_PLAIN : MODE, MT, TW, X2>;
    // X3: a fragment carries a PAIR of taps (K = 32 = 2 taps x 16 channels), in three split planes: [fragment][t][split][64 lanes][16 B]
    constexpr int NFRAG = X3 ? (TAPS + 1) / 2 : (C4 ? 3 : TAPS);             // weight fragments per cout tile and chunk
    constexpr int WSPL = X3 ? 3 : 1;
    constexpr int RED_FLOATS = 4 * NT * 16 * 2;
    constexpr int WT_FLOATS = NFRAG * NT * WSPL * 256;
    constexpr int XT_ALLOC = X3 ? XS3::XT_BYTES / 4 : G::XT_FLOATS;
    static_assert((NIMG * (XT_ALLOC + WT_FLOATS) + RED_FLOATS + (X2 ? 3 : 2) * CTL_PRO_MAX) * 4 <= 160 * 1024, "LDS budget of one CU");
    __shared__ __attribute__((aligned(16))) float xt[NIMG * (XT_ALLOC + WT_FLOATS) + RED_FLOATS + (X2 ? 3 : 2) * CTL_PRO_MAX];
    float* wt = xt + NIMG * XT_ALLOC;
    float* sred = wt + NIMG * WT_FLOATS; // statistics reduction scratch (a flush can happen while xt holds the next tile)
    float* cf_scale = sred + RED_FLOATS; // prologue coefficients [groups][cin]
    float* cf_shift = cf_scale + CTL_PRO_MAX;
    float* cf_c = cf_shift + (X2 ? CTL_PRO_MAX : 0);
    constexpr int WU = NFRAG * NT * WSPL * 64, NW = (WU + 255) / 256;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave_all = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wave = PC ? (wave_all & 3) : wave_all;                 // PC: index among the producers (waves 0-3) / the consumers (waves 4-7)
    const bool consumer = PC && wave_all >= 4;
    const int p = lane & 15, q = lane >> 4;
    // Tile ownership.  Blocks b, b+8, ... share an XCD (and its L2) and the dispatcher spreads blocks breadth-first: the
    // first 256 land on distinct CUs (tools/micro/dispatch_probe.hip).  So the tile list is cut into one contiguous range
    // per XCD (neighbouring tiles share halos in one L2) and the j-th block of an XCD walks tiles j, j+nb, ... of that
    // range: the blocks that get one tile more are the first of each XCD, i.e. sit on different CUs.
    const int P = gridDim.x < 8 ? (int)gridDim.x : 8;
    const int xcd = blockIdx.x % P, jblk = blockIdx.x / P;
    const int nb = ((int)gridDim.x - xcd + P - 1) / P;                 // blocks of this XCD
    const int t_lo = (int)(((int64_t)ntiles * xcd) / P), t_hi = (int)(((int64_t)ntiles * (xcd + 1)) / P);
    const int bid0 = t_lo + jblk;
    const int z = blockIdx.z;
    const int cot0 = blockIdx.y * NT;
    const float* wp = wpack + (int64_t)z * wpack_sub_stride;
    const int my_tiles = (bid0 < t_hi) ? (t_hi - bid0 + nb - 1) / nb : 0;
    const int total_it = my_tiles * G_chunks;
    const int flags = d.epi_flags;
    const int ngroups = d.groups > 1 ? d.groups : 1;
    const int group_n = d.n / ngroups;                               // images per BatchNorm group
    const int oy0 = (z >> 1) * d.out_sub, ox0 = (z & 1) * d.out_sub;
    const __amdgpu_buffer_rsrc_t rx = ctl_rsrc(x, (int64_t)d.n * d.hin * d.win * d.cin * 4);
    const __amdgpu_buffer_rsrc_t rx2 = X2 ? ctl_rsrc(x2, (int64_t)d.n * d.hin * d.win * d.cin * 4) : rx;
    const bool xout_on = X2 && xout != nullptr && blockIdx.y == 0 && blockIdx.z == 0;
    const __amdgpu_buffer_rsrc_t rxout = xout_on ? ctl_rsrc(xout, (int64_t)d.n * d.hin * d.win * d.cin * 4) : rx;
    const int64_t ybytes = (int64_t)d.n * d.out_h * d.out_w * d.cout * 4;
    const __amdgpu_buffer_rsrc_t ry = ctl_rsrc(y, ybytes);
    const __amdgpu_buffer_rsrc_t rres = ctl_rsrc(EPI ? (const void*)res : (const void*)y, ybytes);
    // EPI == 2 (CTL_EPI_TAILBWD): this launch produces dL/dOut of a residual block; the epilogue turns it into g = dOut * leaky'(out)
    // (res = the block's stored output), stores g and takes the BatchNorm-backward sums (sum g, sum g*v; res2 = v, the BatchNorm input of the
    // tail) -- the stand-alone reduction pass over dOut, out and v (ctl_bwd_reduce mode 0) and the dOut tensor itself disappear
    constexpr bool TAIL = (EPI == 2);
    const __amdgpu_buffer_rsrc_t rres2 = ctl_rsrc(TAIL ? (const void*)res2 : (const void*)y, ybytes);
    // TAIL with `pool` (1x1 hosts whose waves own row pairs): the epilogue also writes sumpool2(g) -- the 2x2 sum-pool of the stored g, which
    // the consuming block's 1x1 weight / data gradients read (nearest-upsample blocks): the stand-alone pooling pass disappears.  Same
    // association as sumpool2_kernel, (a0 + a1) + (a2 + a3): bit-identical.
    constexpr bool CAN_POOL = TAIL && KS == 1 && S == 1 && MODE == CTL_IN_PLAIN && (MT / (TW / 16)) == 2;
    const __amdgpu_buffer_rsrc_t rpool = ctl_rsrc(CAN_POOL && pool ? (const void*)pool : (const void*)y, CAN_POOL && pool ? ybytes / 4 : ybytes);

    f32x4 ssum[NT], ssq[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) ssum[t] = ssq[t] = f32x4{0.f, 0.f, 0.f, 0.f};

    // M-tile m of this wave: tile row tr = wave*(MT/TWT) + m/TWT, first column tc = (m % TWT) * 16
    constexpr int TWT = TW / 16;
    static_assert(MT % TWT == 0, "a wave's M-tiles must cover whole tile rows");
    const int wrow = wave * (MT / TWT);
    // per-thread constants of the epilogue: byte offset of this lane's 4 channels of M-tile m relative to the tile's output origin
    int yrel[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
        yrel[m] = (((wrow + m / TWT) * d.out_sy * d.out_w + ((m % TWT) * 16 + p) * d.out_sx) * d.cout + q * 4) * 4;
    // LDS operand addresses: one per-thread base; (M-tile, tap) offsets are compile-time immediates of the ds_read
    const float* xrd = xt + ((wrow * S) * G::IWP + p) * 16 + q * 4;
    const float* wrd = wt + lane * 4;
    const float* xrd4[3];            // C4: lane group q reads channels 0-3 of the pixel shifted by tap 4j+q (tap 8 only for q = 0)
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int tp = (4 * j + q < 9) ? 4 * j + q : 8;
        xrd4[j] = xt + ((wrow + tp / 3) * G::IWP + p + tp % 3) * 4;      // (dense image: 4 floats per pixel)
    }
    // X3: byte offsets of the B operand inside plane (split 0, half q & 1): lane group q carries tap 2f + (q >> 1) of fragment f (clamped
    // to the last tap: its pair partner has zero weights); M-tile and split offsets are compile-time immediates
    int xoff3[X3 ? NFRAG : 1];
    if constexpr (X3) {
#pragma unroll
        for (int f = 0; f < NFRAG; ++f) {
            int tap = 2 * f + (q >> 1);
            tap = tap < TAPS ? tap : TAPS - 1;
            const int kh = tap / KS, kw = tap % KS;
            const int kcol = (S == 2) ? ((kw & 1) * G::IWH + (kw >> 1)) : kw;
            xoff3[f] = (((wrow * S + kh) * G::IWP + p + kcol) * 16) + (q & 1) * XS3::PLANE;
        }
    }
    const unsigned char* xt8 = reinterpret_cast<const unsigned char*>(xt);
    const unsigned char* wrd8 = reinterpret_cast<const unsigned char*>(wt) + lane * 16;

    std::conditional_t<X3, XS3, XStage<KS, S, MODE, MT, TW, X2>> xs;
    xs.init(d);
    if (KS == 2 && S == 1) {
        // phase problems (z = 2a + b, outputs at (2i+a, 2j+b)): d.pad == 2 -> 3x3 conv on a nearest-upsampled input, phase (a,b)
        // reads rows i+a-1, i+a (pad 1-a); d.pad == 0 -> data gradient of a stride-2 3x3 conv, every phase reads rows i, i+1
        xs.pad_h = d.pad == 2 ? 1 - (z >> 1) : 0;
        xs.pad_w = d.pad == 2 ? 1 - (z & 1) : 0;
    }
    // weight chunk g: [tap][t][64 lanes][4] floats; per-thread byte offsets are loop-invariant, the chunk goes in the scalar offset
    const __amdgpu_buffer_rsrc_t rw = ctl_rsrc(wp, (int64_t)ctl_cdiv(d.cout, 16) * NFRAG * G_chunks * WSPL * 1024);
    f32x4 wv[NW];
    int wrel[NW];
#pragma unroll
    for (int i = 0; i < NW; ++i) {
        const int u = tid + i * 256;
        const int tt = u >> 6, l = u & 63;           // tt = tap * NT + t   (X3: (fragment * NT + t) * 3 + split)
        const int sp = X3 ? tt % WSPL : 0, tq = tt / WSPL;
        const int tap = tq / NT, t = tq - tap * NT;
        // global layout [cout tile][fragment][chunk][split][64 lanes][16 B]; LDS layout [fragment][t][split][64 lanes][16 B]
        wrel[i] = (u < WU) ? (((((cot0 + t) * NFRAG + tap) * G_chunks) * WSPL + sp) * 64 + l) * 16 : CTL_OOB;
    }
    auto wload = [&](int g) {
#pragma unroll
        for (int i = 0; i < NW; ++i) wv[i] = ctl_bload4s(rw, wrel[i], g * WSPL * 1024);
    };
    auto wstore = [&](int img = 0) {
#pragma unroll
        for (int i = 0; i < NW; ++i) {
            const int u = tid + i * 256;
            if (u < WU) *reinterpret_cast<f32x4*>(wt + img * WT_FLOATS + u * 4) = wv[i];
        }
    };

    TileWalk cur, nxt;
    cur.init(bid0, nb, tiles_h, tiles_w);
    nxt = cur;
    TM_DECL
    if (!PC && total_it > 0) {
        xs.load(rx, rx2, d, cur.n, cur.th * G::TH, cur.tw * TW, 0);
        wload(0);
    }
    // Everything else the block needs from memory before its first tile is requested BEHIND the first tile's loads and waited for ONCE: the
    // bias (every forward conv has one) used to be loaded and waited for ahead of ~300 instructions of set-up and the tile's loads -- one
    // memory round trip of its own at the start of 130 launches per step (ISA; a loaded round trip is 2-4 us, profiles/r5_finalize_probe.txt)
    // -- and the prologue coefficients another behind it.
    // bias of this lane's 4 output channels: the accumulators start from it (no add in the epilogue)
    // (NO load under a branch: the wait-count pass puts an s_waitcnt vmcnt(0) at the join; absent operands read a valid dummy address)
    f32x4 bias4[NT];
    float bv[NT][4];                 // the raw loads: first USED behind the coefficient requests below
    {
        const bool has_bias = (flags & CTL_EPI_BIAS) != 0;
        const float* bp = has_bias ? bias : wpack;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int co0 = (cot0 + t) * 16 + q * 4;
#pragma unroll
            for (int k = 0; k < 4; ++k) bv[t][k] = bp[has_bias ? (co0 + k < d.cout ? co0 + k : d.cout - 1) : k];
        }
    }
    {   // prologue coefficients [groups][cin] (groups * cin <= CTL_PRO_MAX = 256 <= the block's threads): one entry per thread, requested
        // unconditionally (clamped) so that no load sits under a branch, stored to LDS behind the wait
        const int ncf = ngroups * d.cin;
        const bool pro_on = X2 || d.pro_affine != 0;
        const bool con = pro_on && tid < ncf;
        const int ci = con ? tid : 0;
        float c0, c1, c2 = 0.f;
        if constexpr (X2) {      // coefficients as the BatchNorm-backward finalize writes them: [group][A | B | C][cin]
            const int gi = ci / d.cin, ch = ci - gi * d.cin;
            c0 = pro_scale[(gi * 3 + 0) * d.cin + ch]; c1 = pro_scale[(gi * 3 + 1) * d.cin + ch]; c2 = pro_scale[(gi * 3 + 2) * d.cin + ch];
        } else {
            const float* ps = pro_on ? pro_scale : wpack;
            const float* ph = pro_on ? pro_shift : wpack;
            c0 = ps[ci]; c1 = ph[ci];
        }
        // the bias is complete HERE (pinned): left to itself the wait-count pass puts an s_waitcnt vmcnt(0) in front of the first read of
        // bias4 INSIDE the tile loop (the accumulator init), right behind the next tile's prefetch loads
        // (ONE statement consumes the last-requested values and hands the bias on: the scheduler cannot put a use of the bias, and with it
        //  a wait, in front of the coefficient requests)
        static_assert(NT == 1 || NT == 2, "bias hand-over written for one or two cout tiles per block");
        if constexpr (NT == 1)
            asm volatile("" : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(bv[0][0]), "+v"(bv[0][1]), "+v"(bv[0][2]), "+v"(bv[0][3]));
        else
            asm volatile("" : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(bv[0][0]), "+v"(bv[0][1]), "+v"(bv[0][2]), "+v"(bv[0][3]),
                              "+v"(bv[NT - 1][0]), "+v"(bv[NT - 1][1]), "+v"(bv[NT - 1][2]), "+v"(bv[NT - 1][3]));
        const bool hb = (flags & CTL_EPI_BIAS) != 0;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int co0 = (cot0 + t) * 16 + q * 4;
            bias4[t] = f32x4{(hb && co0 + 0 < d.cout) ? bv[t][0] : 0.f, (hb && co0 + 1 < d.cout) ? bv[t][1] : 0.f,
                             (hb && co0 + 2 < d.cout) ? bv[t][2] : 0.f, (hb && co0 + 3 < d.cout) ? bv[t][3] : 0.f};
            asm volatile("" ::"v"(bias4[t]));
        }
        if (pro_on) {
            if (con) {
                cf_scale[tid] = c0; cf_shift[tid] = c1;
                if constexpr (X2) cf_c[tid] = c2;
            }
            __syncthreads();
        }
    }
    if constexpr (!PC) {
        if (total_it > 0) {
            xs.store(xt, d, 0, cf_scale, cf_shift, (cur.n / group_n) * d.cin, cf_c, rxout, xout_on);
            wstore();
        }
        __syncthreads();
    }

    // X3, one 16-channel chunk (cin == 16) and the smallest tile: the block's weight fragments never change -> read them from LDS ONCE
    // into registers (5 fragments x 3 splits x 4 VGPRs for a 3x3 kernel with one cout tile) instead of once per tile
    constexpr bool WREG3 = X3 && !PC && MT == 1 && (NFRAG * NT * 3 <= 15);      // (larger tiles: 60 registers the operand read-ahead needs more)
    const bool wreg_on = WREG3 && G_chunks == 1;
    x3_bf16x8 wreg[WREG3 ? NFRAG : 1][WREG3 ? 3 : 1][WREG3 ? NT : 1];
    if constexpr (WREG3) {
        if (wreg_on) {
#pragma unroll
            for (int f = 0; f < NFRAG; ++f)
#pragma unroll
                for (int sp = 0; sp < 3; ++sp)
#pragma unroll
                    for (int t = 0; t < NT; ++t) wreg[f][sp][t] = *reinterpret_cast<const x3_bf16x8*>(wrd8 + ((f * NT + t) * 3 + sp) * 1024);
        }
    }

    // Statistics of one BatchNorm group: block-level sum through LDS -> stats_partial[group][block][2][cout].  Called when the
    // walk enters the next group (tiles are visited in increasing order) and once at the end; every wave takes part.
    // (PC: one row per consumer wave, [sub-problem][block][wave]: the wave's own sums go straight to global memory)
    const int srows = gridDim.x * gridDim.z * (PC ? 4 : 1), srow = (z * gridDim.x + blockIdx.x) * (PC ? 4 : 1) + (PC ? wave : 0);
    auto flush_stats = [&](int grp) {
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            float a[8] = {ssum[t].x, ssum[t].y, ssum[t].z, ssum[t].w, ssq[t].x, ssq[t].y, ssq[t].z, ssq[t].w};
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                float v = a[i];
                v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); v += __shfl_xor(v, 4); v += __shfl_xor(v, 8);
                a[i] = v;
            }
            if constexpr (PC) {
                if (p == 0) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int co = (cot0 + t) * 16 + q * 4 + r;
                        if (co < d.cout) {
                            stats_partial[(((int64_t)grp * srows + srow) * 2 + 0) * d.cout + co] = a[r];
                            stats_partial[(((int64_t)grp * srows + srow) * 2 + 1) * d.cout + co] = a[4 + r];
                        }
                    }
                }
                ssum[t] = ssq[t] = f32x4{0.f, 0.f, 0.f, 0.f};
                continue;
            }
            if (p == 0) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    sred[((wave * NT + t) * 16 + q * 4 + r) * 2 + 0] = a[r];
                    sred[((wave * NT + t) * 16 + q * 4 + r) * 2 + 1] = a[4 + r];
                }
            }
            ssum[t] = ssq[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        if constexpr (PC) return;
        __syncthreads();
        if (tid < NT * 16 * 2) {
            const int stat = tid / (NT * 16), cl = tid % (NT * 16);
            float v = 0.f;
#pragma unroll
            for (int w = 0; w < 4; ++w) v += sred[((w * NT * 16) + cl) * 2 + stat];
            const int co = cot0 * 16 + cl;
            if (co < d.cout) stats_partial[(((int64_t)grp * srows + srow) * 2 + stat) * d.cout + co] = v;
        }
        __syncthreads();
    };
    int cur_grp = (my_tiles > 0) ? cur.n / group_n : 0;
    if constexpr (PC) {      // groups this wave never visits contribute zeros (written by the lanes that flush: program order per address)
        if ((flags & CTL_EPI_STATS) && ngroups > 1 && consumer && p == 0) {
            for (int gi = 0; gi < ngroups; ++gi)
#pragma unroll
                for (int t = 0; t < NT; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int co = (cot0 + t) * 16 + q * 4 + r;
                        if (co < d.cout) {
                            stats_partial[(((int64_t)gi * srows + srow) * 2 + 0) * d.cout + co] = 0.f;
                            stats_partial[(((int64_t)gi * srows + srow) * 2 + 1) * d.cout + co] = 0.f;
                        }
                    }
        }
    } else if ((flags & CTL_EPI_STATS) && ngroups > 1 && tid < NT * 16 * 2) {      // groups this block never visits contribute zeros
        const int stat = tid / (NT * 16), co = cot0 * 16 + tid % (NT * 16);
        if (co < d.cout)
            for (int gi = 0; gi < ngroups; ++gi) stats_partial[(((int64_t)gi * srows + srow) * 2 + stat) * d.cout + co] = 0.f;
    }

    TM(7)
    f32x4 acc[MT][NT];
#ifndef CTL_X3_ABLATE
#define CTL_X3_ABLATE 0      // timing ablations of the X3 instantiations (variant builds only, WRONG results): 1 no split arithmetic, 2 no global
#endif                       // loads in the loop, 4 no matrix phase, 8 no epilogue (profiles/r4_x3_ablation.txt)

    // ---------------- X3 matrix phase of one (tile, chunk) step out of the LDS images at xb (input planes) / wb (weight fragments)
    // Six MFMAs per (fragment, M-tile, cout tile): hi*hi + hi*mid + mid*hi + mid*mid + hi*lo + lo*hi (ctl_conv_x3_stage.h).  A step is one
    // fragment of MS M-tiles; the three split operands of the next step are requested before this step's MFMAs (the empty asm
    // pins that order: the scheduler otherwise sinks every read to its use -- read, wait, MFMA, see ctl_conv_bf16.hip)
    auto mfma_phase3 = [&](auto wreg_tag, const unsigned char* xb, const unsigned char* wb) {
        constexpr bool WR = decltype(wreg_tag)::value;
#ifndef CTL_X3_MS
#define CTL_X3_MS(MT, NT, X2) ((CTL_LB_X3(MT, NT, X2) == 3 || (MT) < 2) ? 1 : 2)
#endif
        constexpr int MS = PC ? (MT >= 2 ? 2 : 1) : CTL_X3_MS(MT, NT, X2);       // M-tiles per operand step: their MFMA chains interleave
        constexpr int NSTEP = NFRAG * (MT / MS);
        x3_bf16x8 xf[2][MS][3], wf[2][WR ? 1 : 3][WR ? 1 : NT];
        auto xread = [&](int st, int b) {
            const int f = st / (MT / MS), m0 = (st % (MT / MS)) * MS;
#pragma unroll
            for (int mi = 0; mi < MS; ++mi)
#pragma unroll
                for (int sp = 0; sp < 3; ++sp)
                    xf[b][mi][sp] = *reinterpret_cast<const x3_bf16x8*>(xb + xoff3[f] + ((((m0 + mi) / TWT) * S) * G::IWP + ((m0 + mi) % TWT) * 16) * 16 + 2 * sp * XS3::PLANE);
        };
        auto wread = [&](int f, int b) {
            if constexpr (!WR) {
#pragma unroll
                for (int sp = 0; sp < 3; ++sp)
#pragma unroll
                    for (int t = 0; t < NT; ++t) wf[b][sp][t] = *reinterpret_cast<const x3_bf16x8*>(wb + ((f * NT + t) * 3 + sp) * 1024);
            }
        };
        xread(0, 0);
        wread(0, 0);
#pragma unroll
        for (int st = 0; st < NSTEP; ++st) {
            const int f = st / (MT / MS), m0 = (st % (MT / MS)) * MS, b = st & 1;
            if (st + 1 < NSTEP) {
                xread(st + 1, b ^ 1);
                if ((st + 1) % (MT / MS) == 0) wread(f + 1, (f + 1) & 1);
            }
#pragma unroll
            for (int mi = 0; mi < MS; ++mi) asm volatile("" : "+v"(xf[b][mi][0]), "+v"(xf[b][mi][1]), "+v"(xf[b][mi][2]) : : "memory");
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                x3_bf16x8 ah, am, al;
                if constexpr (WR) { ah = wreg[WREG3 ? f : 0][0][WREG3 ? t : 0]; am = wreg[WREG3 ? f : 0][WREG3 ? 1 : 0][WREG3 ? t : 0]; al = wreg[WREG3 ? f : 0][WREG3 ? 2 : 0][WREG3 ? t : 0]; }
                else { ah = wf[f & 1][0][t]; am = wf[f & 1][WR ? 0 : 1][t]; al = wf[f & 1][WR ? 0 : 2][t]; }
                // (small terms first; the chains of the step's M-tiles alternate)
#define CTL_X3_MFMA(A, SP) _Pragma("unroll") for (int mi = 0; mi < MS; ++mi) \
                    acc[m0 + mi][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A, xf[b][mi][SP], acc[m0 + mi][t], 0, 0, 0);
                CTL_X3_MFMA(al, 0)
                CTL_X3_MFMA(ah, 2)
                CTL_X3_MFMA(am, 1)
                CTL_X3_MFMA(am, 0)
                CTL_X3_MFMA(ah, 1)
                CTL_X3_MFMA(ah, 0)
#undef CTL_X3_MFMA
            }
        }
    };

    // ---------------- epilogue of one tile: lane (p,q) holds channels co0..co0+3 of pixel p of each M-tile.  Buffer stores with
    // hardware bounds checks; nothing here is waited for in the loop.  Whole tiles with whole channel tiles (FULL) put
    // the tile origin into the scalar offset of the loads and skip every mask; ragged ones redirect dropped lanes to
    // CTL_OOB.  Two halves: epi_load requests the epilogue's operands (residual / accumulate / tail tensors, the group's residual
    // coefficients), epi_finish consumes them.  The single-role kernels call them back to back (requesting the operands before the
    // barriers was measured there: the extra live registers cost more than the hidden latency gains); the consumer waves of the PC
    // form call epi_load BEFORE the step's barrier and matrix phase -- with one block per CU nothing else would cover that round trip.
    // EPI == 3: CTL_EPI_BNBWD alone, known at compile time (the 3x3 data gradients of the residual blocks): no accumulate
    // operand, no per-fragment flag tests -- 16 VGPRs less than the generic EPI == 1 form, which matters next to the second
    // staged tensor of the X2 prologue (168 VGPRs + 120 B of scratch otherwise)
    constexpr bool BNB = (EPI == 3), HAS_OV = (EPI == 1 || EPI == 2);
    f32x4 rv[EPI ? MT : 1][EPI ? NT : 1], ov[HAS_OV ? MT : 1][HAS_OV ? NT : 1], r2[TAIL ? MT : 1][TAIL ? NT : 1];
    f32x4 e_rs[EPI ? NT : 1] = {}, e_rh[EPI ? NT : 1] = {};
    auto tile_full = [&](int ho0, int wo0) {
#ifdef CTL_NO_FULL_EPI
        return false;
#else
        return ho0 + G::TH <= d.hout && wo0 + TW <= d.wout && (cot0 + NT) * 16 <= d.cout;
#endif
    };
    auto tile_ybase = [&](int n, int ho0, int wo0) {
        return (((n * d.out_h + ho0 * d.out_sy + oy0) * d.out_w + wo0 * d.out_sx + ox0) * d.cout + cot0 * 16) * 4;
    };
    auto epi_load = [&](auto full_tag, int n, int ho0, int wo0) {
        constexpr bool FULL = decltype(full_tag)::value;
        if constexpr (EPI != 0) {
            const int grp = n / group_n;
            const int ybase = tile_ybase(n, ho0, wo0);
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const bool cok = FULL || (cot0 + t) * 16 + q * 4 < d.cout;
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                    const bool pvm = FULL || ((ho0 + wrow + m / TWT < d.hout) && (wo0 + (m % TWT) * 16 + p < d.wout));
                    const int vo = FULL ? (yrel[m] + t * 64) : ((pvm && cok) ? (ybase + yrel[m] + t * 64) : CTL_OOB);
                    const int so = FULL ? ybase : 0;
                    rv[m][t] = f32x4{0.f, 0.f, 0.f, 0.f};
                    if constexpr (HAS_OV) ov[m][t] = f32x4{0.f, 0.f, 0.f, 0.f};
                    if constexpr (BNB) {            // (cout is a multiple of 16 here, checked on the host)
                        rv[m][t] = ctl_bload4s(rres, vo, so);
                    } else if constexpr (TAIL) {    // (likewise)
                        rv[m][t] = ctl_bload4s(rres, vo, so);
                        r2[m][t] = ctl_bload4s(rres2, vo, so);
                        if (flags & CTL_EPI_ACCUM) ov[m][t] = ctl_bload4s(ry, vo, so);
                    } else if (d.cout >= 4) {
                        if (flags & (CTL_EPI_RES | CTL_EPI_BNBWD)) rv[m][t] = ctl_bload4s(rres, vo, so);
                        if (flags & CTL_EPI_ACCUM) ov[m][t] = ctl_bload4s(ry, vo, so);
                    } else {
                        if (flags & (CTL_EPI_RES | CTL_EPI_BNBWD)) rv[m][t].x = ctl_bload1(rres, vo);
                        if (flags & CTL_EPI_ACCUM) ov[m][t].x = ctl_bload1(ry, vo);
                    }
                }
                const int co0 = (cot0 + t) * 16 + q * 4;
                const int cc = cok ? co0 : 0;
                f32x4 rs = {0.f, 0.f, 0.f, 0.f}, rh = {0.f, 0.f, 0.f, 0.f};
                if (!TAIL && (BNB || (flags & (CTL_EPI_RES | CTL_EPI_BNBWD)))) {
                    if (d.cout >= 4) {
                        rs = *reinterpret_cast<const f32x4*>(res_scale + grp * d.cout + cc);
                        rh = *reinterpret_cast<const f32x4*>(res_shift + grp * d.cout + cc);
                    } else { rs.x = res_scale[grp]; rh.x = res_shift[grp]; }
                }
                e_rs[t] = rs; e_rh[t] = rh;
            }
        }
    };
    auto epi_finish = [&](auto full_tag, int n, int ho0, int wo0) {
        constexpr bool FULL = decltype(full_tag)::value;
        const int ybase = tile_ybase(n, ho0, wo0);
#ifndef CTL_NO_FAST_EPI
        if constexpr (!EPI && FULL) {
            // the common case (whole tile, no residual / accumulate operand, no activation): the stores read the accumulator
            // registers directly and the statistics sit behind a real branch.  Written as its own path because the generic
            // code below funnels every variant through one set of store registers (4 v_mov per fragment) and turns the
            // statistics flag into 8 v_cndmask per tile -- VALU issue slots taken from the matrix pipe.
            if (d.epi_act == CTL_ACT_NONE) {
                if (flags & CTL_EPI_STATS) {
                    asm volatile("" ::: "memory");                  // keeps the branch (not a select)
#pragma unroll
                    for (int t = 0; t < NT; ++t)
#pragma unroll
                        for (int m = 0; m < MT; ++m) { ssum[t] += acc[m][t]; ssq[t] += acc[m][t] * acc[m][t]; }
                }
#pragma unroll
                for (int t = 0; t < NT; ++t)
#pragma unroll
                    for (int m = 0; m < MT; ++m) ctl_bstore4(ry, ybase + yrel[m] + t * 64, acc[m][t]);
                return;
            }
        }
#endif
        bool pv[MT];
#pragma unroll
        for (int m = 0; m < MT; ++m)
            pv[m] = FULL || ((ho0 + wrow + m / TWT < d.hout) && (wo0 + (m % TWT) * 16 + p < d.wout));
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int co0 = (cot0 + t) * 16 + q * 4;
            const bool cok = FULL || co0 < d.cout;
            const f32x4 rs = e_rs[EPI ? t : 0], rh = e_rh[EPI ? t : 0];
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                f32x4 v = acc[m][t];
                if constexpr (TAIL) {
                    v += ov[m][t];                                   // (CTL_EPI_ACCUM: the other half of dOut, already in y)
                    const f32x4 o = rv[m][t];
                    const float sl = d.epi_slope;
                    v.x *= o.x > 0.f ? 1.f : sl; v.y *= o.y > 0.f ? 1.f : sl;
                    v.z *= o.z > 0.f ? 1.f : sl; v.w *= o.w > 0.f ? 1.f : sl;
                    if (FULL || pv[m]) { ssum[t] += v; ssq[t] += v * r2[m][t]; }
                    if (FULL) ctl_bstore4(ry, ybase + yrel[m] + t * 64, v);
                    else ctl_bstore4(ry, (pv[m] && cok) ? (ybase + yrel[m] + t * 64) : CTL_OOB, v);
                    if constexpr (CAN_POOL) {
                        if (pool) {
                            // column pairs sit in neighbouring lanes (p, p ^ 1), the wave's two rows in M-tiles m and m + TWT
                            f32x4 hs;
                            hs.x = v.x + __shfl_xor(v.x, 1); hs.y = v.y + __shfl_xor(v.y, 1);
                            hs.z = v.z + __shfl_xor(v.z, 1); hs.w = v.w + __shfl_xor(v.w, 1);
                            if (m < TWT) ov[m][t] = hs;              // (the accumulate operand of this fragment is consumed: its registers hold the top row's pair sums)
                            else {
                                const f32x4 top = ov[m - TWT][t];
                                const f32x4 pl = {top.x + hs.x, top.y + hs.y, top.z + hs.z, top.w + hs.w};
                                // low-resolution pixel ((ho0 + wrow) / 2, (wo0 + (m % TWT) * 16 + p) / 2); written by the even lanes
                                const int lo = ((((n * (d.out_h >> 1) + ((ho0 + wrow) >> 1)) * (d.out_w >> 1) + ((wo0 + (m % TWT) * 16 + p) >> 1)) * d.cout) + co0) * 4;
                                ctl_bstore4(rpool, ((p & 1) == 0 && (FULL || (pv[m] && cok))) ? lo : CTL_OOB, pl);
                            }
                        }
                    }
                    continue;
                }
                if (BNB || (EPI == 1 && (flags & CTL_EPI_BNBWD))) {
                    // this conv produced dL/da of a = leaky(BN(u)): turn it into g = dL/da * leaky'(BN(u)) and take the two
                    // sums of the BatchNorm backward (sum g, sum g*u) here instead of in a separate pass over da and u
                    const f32x4 u = rv[m][t], sa = u * rs + rh;
                    const float sl = d.epi_slope;
                    v.x *= sa.x > 0.f ? 1.f : sl; v.y *= sa.y > 0.f ? 1.f : sl;
                    v.z *= sa.z > 0.f ? 1.f : sl; v.w *= sa.w > 0.f ? 1.f : sl;
                    if (FULL || pv[m]) { ssum[t] += v; ssq[t] += v * u; }
                } else {
                    if (EPI) v += rv[m][t] * rs + rh;
                    if (flags & CTL_EPI_STATS) {
                        if (FULL) { ssum[t] += v; ssq[t] += v * v; }
                        else if (pv[m]) { ssum[t] += v; ssq[t] += v * v; }
                    }
                }
                if (d.epi_act == CTL_ACT_LEAKY) {
                    v = ctl_leaky01(v, d.epi_slope);
                } else if (d.epi_act == CTL_ACT_SIGMOID) {
                    v.x = 1.f / (1.f + expf(-v.x)); v.y = 1.f / (1.f + expf(-v.y));
                    v.z = 1.f / (1.f + expf(-v.z)); v.w = 1.f / (1.f + expf(-v.w));
                }
                if constexpr (HAS_OV) v += ov[m][t];
                if (FULL) {
                    ctl_bstore4(ry, ybase + yrel[m] + t * 64, v);     // no SGPR soffset on stores, see ctl_bload4s
                } else {
                    const int vo = (pv[m] && cok) ? (ybase + yrel[m] + t * 64) : CTL_OOB;
                    if (d.cout >= 4) ctl_bstore4(ry, vo, v);
                    else ctl_bstore1(ry, vo, v.x);       // cout == 1: only q == 0 passes `cok`, component x is the channel
                }
            }
        }
    };

    if constexpr (PC) {
        // ================================================================ producer / consumer form (see the template's header comment)
        const bool wdouble = G_chunks > 1;                       // one weight chunk per step -> two weight images; cin == 16: staged once
        if (!consumer) {
            // ---------------- producers: CTL_PC_SETS payload sets in flight; step `it` lives in set it % R and goes to image it & 1.
            // Every load of the steady-state loop is UNCONDITIONAL (steps past the last one load from out-of-range offsets): at a
            // control-flow join the compiler's wait-count pass keeps the SMALLER number of younger loads, so one conditional load in
            // the loop body turns every counted wait into a near-drain (seen in the ISA: vmcnt(5) where 17 were in flight)
            constexpr int R = X2 ? CTL_PC_SETS_X2 : CTL_PC_SETS;      // (two staged tensors: 48 registers per set of an 8x32 tile)
            typename XS3::Pay P[R];
            int s_n[R], s_g[R];
            TileWalk lw = cur;
            int lg = 0, lit = 0;
            auto load_step = [&](typename XS3::Pay& pay, int& sn, int& sg) {
                xs.template load<true>(pay, rx, rx2, d, lw.n, lw.th * G::TH, lw.tw * TW, lg, lit < total_it);
                sn = lw.n; sg = lg;
                ++lit;
                if (++lg == G_chunks) { lg = 0; lw.next(); }
            };
            // (issue order of the steady state -- ..., weights of step it + 1, inputs of step it + R -- from the first load on: the wait
            //  counts at the loop header are the join of the prologue's and the back edge's)
#pragma unroll
            for (int r = 0; r < R - 1; ++r) load_step(P[r], s_n[r], s_g[r]);
            wload(0);
            load_step(P[R - 1], s_n[R - 1], s_g[R - 1]);
            auto run = [&](auto wd_tag) {
                constexpr bool WD = decltype(wd_tag)::value;     // one weight chunk per step -> two weight images (cin == 16: staged once)
                int wg = 0;                                      // chunk of the weights held in wv[]
                auto step = [&](int it, typename XS3::Pay& pay, int& sn, int& sg, bool more) {
                    const int img = it & 1;
#ifdef CTL_TIMING_X3_VMCNT
                    asm volatile("s_waitcnt vmcnt(%0)" :: "n"((R - 1) * XS3::NU * (X2 ? 4 : 2)) : "memory");
                    TM(9)                                        // (timing variant: the wait for this step's loads alone; WD launches over-wait here)
#endif
                    xs.template store<true>(pay, xt + img * XT_ALLOC, d, sg, cf_scale, cf_shift, (sn / group_n) * d.cin, cf_c, rxout, xout_on);
                    if constexpr (WD) wstore(img);
                    else if (it == 0) wstore(0);
                    TM(3)
                    if (more) {      // (the next step's weights FIRST: loads return in order, so whatever is issued behind them
                        if constexpr (WD) { wg = (wg + 1 == G_chunks) ? 0 : wg + 1; wload(wg); }      // may still be in flight at their use)
                        load_step(pay, sn, sg);
                    }
                    TM(0)
                    ctl_barrier_lds_writes_done();               // image `img` is complete (and every consumer has left step it - 1's image)
                    TM(4)
                };
                const int full = (total_it / R) * R;
                for (int it0 = 0; it0 < full; it0 += R) {
#pragma unroll
                    for (int r = 0; r < R; ++r) step(it0 + r, P[r], s_n[r], s_g[r], true);
                }
#pragma unroll
                for (int r = 0; r < R - 1; ++r)
                    if (full + r < total_it) {
                        step(full + r, P[r], s_n[r], s_g[r], false);
                        if constexpr (WD) { if (full + r + 1 < total_it) { wg = (wg + 1 == G_chunks) ? 0 : wg + 1; wload(wg); } }
                    }
            };
            if (wdouble) run(std::true_type{}); else run(std::false_type{});
            TM_FLUSH
            return;
        }
        // ---------------- consumers: the critical path.  Raised priority: the vector instructions of the epilogue (and the operand
        // addresses of the matrix phase) otherwise queue behind the producer wave of the same SIMD, the OLDER wave, whose staging is one
        // long vector stream (phase timers before: epilogue 900 cycles per tile for 16 packed adds / fmas and four stores)
#ifndef CTL_PC_PRIO
#define CTL_PC_PRIO 2
#endif
        __builtin_amdgcn_s_setprio(CTL_PC_PRIO);
        TileWalk cw = cur;
        int g = 0;
        for (int it = 0; it < total_it; ++it) {
            const int n = cw.n, ho0 = cw.th * G::TH, wo0 = cw.tw * TW;
            const bool last = (g == G_chunks - 1);
            const bool full = tile_full(ho0, wo0);
            if (last) {
                const int grp = n / group_n;
                if ((flags & CTL_EPI_STATS) && grp != cur_grp) { flush_stats(cur_grp); cur_grp = grp; }
                if (full) epi_load(std::true_type{}, n, ho0, wo0); else epi_load(std::false_type{}, n, ho0, wo0);
            }
            TM(7)
            ctl_barrier_lds_reads_done();                        // step `it`'s images are written; this wave's reads of step it - 1 were consumed
            TM(2)
            TM_COUNT(6)
            if (g == 0) {
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int t = 0; t < NT; ++t) acc[m][t] = bias4[t];
            }
            const int img = it & 1;
            mfma_phase3(std::false_type{}, xt8 + img * (XT_ALLOC * 4), wrd8 + (wdouble ? img : 0) * (WT_FLOATS * 4));
            TM(1)
            if (last) {
                if (full) epi_finish(std::true_type{}, n, ho0, wo0); else epi_finish(std::false_type{}, n, ho0, wo0);
            }
            TM(5)
            if (++g == G_chunks) { g = 0; cw.next(); }
        }
        if (flags & CTL_EPI_STATS) flush_stats(cur_grp);
        TM_FLUSH
        return;
    } else {
    for (int it = 0, g = 0; it < total_it; ++it) {
        TM_COUNT(6)
        const int n = cur.n, ho0 = cur.th * G::TH, wo0 = cur.tw * TW;
        const bool has_next = it + 1 < total_it;
        const int g2 = (g + 1 == G_chunks) ? 0 : g + 1;
        const bool new_w = has_next && G_chunks > 1;
        if (g2 == 0) nxt.next();
        if (has_next && !(X3 && (CTL_X3_ABLATE & 2))) {
            xs.load(rx, rx2, d, nxt.n, nxt.th * G::TH, nxt.tw * TW, g2);
            if (new_w) wload(g2);
        }
        TM(0)
        if (g == 0) {        // (taking the bias as the C operand of each accumulator's first MFMA instead costs 25-55 VGPRs)
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int t = 0; t < NT; ++t) acc[m][t] = bias4[t];
        }
        if constexpr (X3 && (CTL_X3_ABLATE & 4)) {
        } else if constexpr (X3) {
            if constexpr (WREG3) {
                if (wreg_on) mfma_phase3(std::true_type{}, xt8, wrd8);
                else mfma_phase3(std::false_type{}, xt8, wrd8);
            } else {
                mfma_phase3(std::false_type{}, xt8, wrd8);
            }
        } else
        {   // operands of tap+1 are requested before the MFMAs of tap.  The machine scheduler sinks the reads back to their uses
            // (register pressure); pinning them with sched_barriers (-DCTL_PIN_READ_AHEAD) costs 12 VGPRs and measures the
            // same: with fp32 MFMA on the VALU port the LDS latency of one wave is covered by the other waves of the SIMD.
            f32x4 wf[2][NT], xf[2][MT];
            auto lds_operands = [&](int tap, int b) {
                const int kh = tap / KS, kw = tap % KS;
                const int kcol = (S == 2) ? ((kw & 1) * G::IWH + (kw >> 1)) : kw;
#pragma unroll
                for (int t = 0; t < NT; ++t) wf[b][t] = *reinterpret_cast<const f32x4*>(wrd + (tap * NT + t) * 256);
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                    if (C4) xf[b][m] = *reinterpret_cast<const f32x4*>(xrd4[tap] + ((m / TWT) * G::IWP + (m % TWT) * 16) * 4);   // tap = quad
                    else xf[b][m] = *reinterpret_cast<const f32x4*>(xrd + (((m / TWT) * S + kh) * G::IWP + (m % TWT) * 16 + kcol) * 16);
                }
            };
            lds_operands(0, 0);
#pragma unroll
            for (int tap = 0; tap < NFRAG; ++tap) {
                const int b = tap & 1;
                if (tap + 1 < NFRAG) lds_operands(tap + 1, b ^ 1);
#ifdef CTL_PIN_READ_AHEAD
                __builtin_amdgcn_sched_barrier(0);
#endif
#pragma unroll
                for (int m = 0; m < MT; ++m) {
#pragma unroll
                    for (int t = 0; t < NT; ++t) {
                        acc[m][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[b][t].x, xf[b][m].x, acc[m][t], 0, 0, 0);
                        acc[m][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[b][t].y, xf[b][m].y, acc[m][t], 0, 0, 0);
                        acc[m][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[b][t].z, xf[b][m].z, acc[m][t], 0, 0, 0);
                        acc[m][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[b][t].w, xf[b][m].w, acc[m][t], 0, 0, 0);
                    }
                }
#ifdef CTL_PIN_READ_AHEAD
                __builtin_amdgcn_sched_barrier(0);
#endif
            }
        }

        TM(1)
        ctl_barrier_lds_reads_done();    // every wave is done reading this step's LDS images
        TM(2)
#ifdef CTL_TIMING_X3_VMCNT      // (timing variant: split the staging phase into the wait for the prefetched loads [2 -> 9] and the rest [3])
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        TM(9)
#endif
        if (has_next) {    // refill LDS from the prefetched registers
            xs.store(xt, d, g2, cf_scale, cf_shift, (nxt.n / group_n) * d.cin, cf_c, rxout, xout_on);   // v[] holds chunk g2 of tile nxt (== cur unless g2 == 0)
            if (new_w) wstore();
        }
        TM(3)
        ctl_barrier_lds_writes_done();
        TM(4)

        if (g == G_chunks - 1 && !(X3 && (CTL_X3_ABLATE & 8))) {
            const int grp = n / group_n;
            if ((flags & CTL_EPI_STATS) && grp != cur_grp) {
                flush_stats(cur_grp);
                cur_grp = grp;
            }
            if (tile_full(ho0, wo0)) { epi_load(std::true_type{}, n, ho0, wo0); epi_finish(std::true_type{}, n, ho0, wo0); }
            else { epi_load(std::false_type{}, n, ho0, wo0); epi_finish(std::false_type{}, n, ho0, wo0); }
        }
        TM(5)
        if (g2 == 0) cur = nxt;
        g = g2;
    }
    }
#ifndef CTL_TIMING_WGRAD
#ifdef CTL_TIMING_DOMINANT_ONLY
    if (KS == 3 && S == 1 && MODE == 0 && MT == 4 && TW == 32 && NT == 1 && EPI == 0)
#endif
    TM_FLUSH
#endif

    if (flags & CTL_EPI_STATS) flush_stats(cur_grp);
}
