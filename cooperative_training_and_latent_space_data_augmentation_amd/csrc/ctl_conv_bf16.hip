// bf16 implicit-GEMM convolution family for gfx950 (MI355X): BASELINE config 3 (bf16 activation storage, fp32 accumulate, fp32
// BatchNorm statistics, fp32 master weights).  Same GEMM view, tile geometry, persistent XCD-aware tile walk and fused prologue /
// epilogue as the fp32 family (ctl_conv.hip); what changes is the matrix instruction and everything that feeds it:
//
//   v_mfma_f32_16x16x32_bf16:  D^T[co][pixel] += W^T[co][k] * X^T[k][pixel],  K = 32 per instruction, 8 bf16 per lane and operand
//     A (weights): lane l -> row co = l & 15, k-slots 8*(l>>4) .. +7        B (input): lane l -> k-slots 8*(l>>4) .. +7, col pixel = l & 15
//   A 16-channel chunk of ONE tap fills only 16 of the 32 k-slots, so a fragment carries a PAIR of taps: lane groups 0,1 hold channels
//   0-7 / 8-15 of tap 2f, groups 2,3 those of tap 2f+1 (a 3x3 conv: 5 fragments per 16-channel chunk instead of 36 fp32 MFMAs; the
//   odd tap of the last fragment has zero weights).  One ds_read_b128 per lane (8 channels of "its" pixel, shifted by "its" tap) is one
//   B operand; weights are pre-packed in exactly that order (pack_weights_bf16_batched_kernel), 16 bytes per lane and fragment.
//
// LDS input tile: [row][col][16 channels] bf16 = 32 B per pixel.  The input is staged from fp32 (network inputs) or bf16 (internal
// activations) global memory; BatchNorm-apply + LeakyReLU of the producer runs in fp32 on the staged values, which are then rounded to
// bf16 once (v_cvt_pk_bf16_f32, round-to-nearest-even) -- the rounding points of the path are: MFMA operands (activations after the
// prologue, weights) and stored tensors; accumulation, bias, statistics, residual and activation arithmetic stay fp32.
// With 14x fewer matrix cycles than fp32 these kernels are HBM-bound: what matters is bytes in flight, not VALU.
#include "ctl_conv_bf16_common.h"

#define NFRAG_OF(KS) (((KS) * (KS) + 1) / 2)

// Phase timers (variant builds only: tools/build_variant.sh tm16 "-DCTL_TIMING16" ctl_conv_bf16.hip; read with ctl_debug_timing16):
// s_memtime deltas summed over every wave: [0] prefetch issue, [1] MFMA loop, [2] barrier after the reads, [3] staging (vmcnt wait +
// prologue + ds_write), [4] barrier after the writes, [5] epilogue, [6] steps, [7] setup, [8] wave span, [9] realtime span (100 MHz)
#ifdef CTL_TIMING16
#define CTL_TM_WAVES 65536
__device__ unsigned long long ctl_tm16[CTL_TM_WAVES][10];
#define TM_DECL unsigned long long tm_prev = __builtin_amdgcn_s_memtime(), tm_acc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; \
                const unsigned long long tm_t0 = tm_prev, tm_r0 = __builtin_amdgcn_s_memrealtime();
#define TM(i) { const unsigned long long tm_now = __builtin_amdgcn_s_memtime(); tm_acc[i] += tm_now - tm_prev; tm_prev = tm_now; }
#define TM_COUNT(i) { tm_acc[i] += 1; }
#define TM_FLUSH { const unsigned w_ = ((blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z)) * 4 + (threadIdx.x >> 6)) % CTL_TM_WAVES; \
                   tm_acc[8] = __builtin_amdgcn_s_memtime() - tm_t0; tm_acc[9] = __builtin_amdgcn_s_memrealtime() - tm_r0; \
                   if ((threadIdx.x & 63) == 0) { _Pragma("unroll") for (int i_ = 0; i_ < 10; ++i_) ctl_tm16[w_][i_] += tm_acc[i_]; } }
extern "C" int ctl_debug_timing16(unsigned long long* out12) {
    static unsigned long long host[CTL_TM_WAVES][10];
    if (hipMemcpyFromSymbol(host, HIP_SYMBOL(ctl_tm16), sizeof(host)) != hipSuccess) return -1;
    for (int i = 0; i < 12; ++i) out12[i] = 0;       // [10] max span over waves, [11] number of waves that ran
    for (int w = 0; w < CTL_TM_WAVES; ++w) {
        for (int i = 0; i < 10; ++i) out12[i] += host[w][i];
        if (host[w][8] > out12[10]) out12[10] = host[w][8];
        if (host[w][8]) out12[11] += 1;
    }
    for (int w = 0; w < CTL_TM_WAVES; ++w) for (int i = 0; i < 10; ++i) host[w][i] = 0;
    return hipMemcpyToSymbol(HIP_SYMBOL(ctl_tm16), host, sizeof(host)) == hipSuccess ? 0 : -1;
}
#else
#define TM_DECL
#define TM(i)
#define TM_COUNT(i)
#define TM_FLUSH
#endif

// resident blocks per CU the register allocation aims for (HBM-bound kernels: tiles in flight per CU is what hides the latency)
#ifndef CTL16_OCC_BIG
#define CTL16_OCC_BIG 2
#endif
#ifndef CTL16_OCC
#define CTL16_OCC 3
#endif
// timing ablations (variant builds only, WRONG results): which resource bounds the kernel?  1 = no epilogue stores, 2 = no global loads
// in the loop, 4 = no MFMA phase (bit mask)
#ifndef CTL16_ABLATE
#define CTL16_ABLATE 0
#endif
#ifndef CTL16_STORE16
#define CTL16_STORE16 0      // 1: 16-byte epilogue stores through v_permlane16_swap (parity-green; measured: 18.0 vs 18.1 us on the 16->16 layer, +24 spilled VGPRs in the 32-channel instantiation -> off)
#endif
// FAST: the instantiation for the layers that carry the step -- bf16 in, bf16 out, whole 16-channel tiles on both sides, no residual /
// accumulate operand, no activation (every 3x3 / 2x2 / 4x4 forward conv with statistics and every plain data gradient).  With those known
// at compile time the epilogue is straight-line code (the generic one decides per fragment between fp32 / bf16 / 1-channel forms of
// three optional operands: ~3700 cycles per tile and wave of scalar branches, measured with the phase timers) and the staging loses
// the fp32-source half of its registers.
// XB: the input is stored as bf16 with whole 16-channel chunks (compile-time staging); the network-boundary layers (fp32 input with 1 or 4
// channels) take the FAST epilogues with the generic staging.
// X2: the input is the virtual BatchNorm-backward result  A * x + B * x2 + C  (XStage16; pro_scale = the [group][3][cin] coefficients)
template <int KS, int S, int MODE, int MT, int TW, int NT, int FAST, int XB, bool X2 = false>      // (XB: XStage16's X16C) FAST: 0 generic; 1 plain; 2 + bf16 residual * scale + shift (+ LeakyReLU); 3 accumulate into y
#ifndef CTL16_OCC_X2
#define CTL16_OCC_X2 2      // resident blocks of the two-tensor instantiations: at 3 they need 20-44 B of scratch per lane (bf16 step 10.64 -> 10.48 ms)
#endif
// (the same for the fp32-input kinds -- two registers per staged unit -- and the stride-2 3x3 form at 4 fragments per wave: 36-100 B of scratch at 3)
__global__ __launch_bounds__(256, (MT * NT >= 8 || KS == 4) ? CTL16_OCC_BIG : ((X2 || XB >= 2 || (KS == 3 && S == 2 && MT * NT >= 4)) ? CTL16_OCC_X2 : CTL16_OCC)) void conv_igemm_bf16_kernel(
    const ctl_conv d, const void* __restrict__ x, const void* __restrict__ x2, const void* __restrict__ wpack, const float* __restrict__ bias,
    const float* __restrict__ pro_scale, const float* __restrict__ pro_shift, const void* __restrict__ res,
    const float* __restrict__ res_scale, const float* __restrict__ res_shift, const void* __restrict__ res2, void* __restrict__ y,
    float* __restrict__ stats_partial, int tiles_h, int tiles_w, int G_chunks, int64_t wpack_sub_bytes, int ntiles, void* __restrict__ pool,
    void* __restrict__ xout) {
    using G = Geom<KS, S, MT, TW>;
    using XS = XStage16<KS, S, MODE, MT, TW, XB, true, X2>;
    constexpr int TAPS = KS * KS;
    constexpr int NFRAG = NFRAG_OF(KS);
    constexpr int XT_ALLOC = XS::XT_BYTES + 16;            // + dump slot
    constexpr int WT_BYTES = NFRAG * NT * 1024;
    constexpr int RED_FLOATS = 4 * NT * 16 * 2;
    __shared__ __attribute__((aligned(16))) unsigned char smem[XT_ALLOC + WT_BYTES + (RED_FLOATS + (X2 ? 3 : 2) * CTL_PRO_MAX) * 4];
    unsigned char* xt = smem;
    unsigned char* wt = smem + XT_ALLOC;
    float* sred = reinterpret_cast<float*>(wt + WT_BYTES);
    float* cf_scale = sred + RED_FLOATS;
    float* cf_shift = cf_scale + CTL_PRO_MAX;
    float* cf_c = cf_shift + (X2 ? CTL_PRO_MAX : 0);
    constexpr int WU = NFRAG * NT * 64, NW = (WU + 255) / 256;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int p = lane & 15, q = lane >> 4;
    const int P = gridDim.x < 8 ? (int)gridDim.x : 8;
    const int xcd = blockIdx.x % P, jblk = blockIdx.x / P;
    const int nb = ((int)gridDim.x - xcd + P - 1) / P;
    const int t_lo = (int)(((int64_t)ntiles * xcd) / P), t_hi = (int)(((int64_t)ntiles * (xcd + 1)) / P);
    const int bid0 = t_lo + jblk;
    const int z = blockIdx.z;
    const int cot0 = blockIdx.y * NT;
    const unsigned char* wp = reinterpret_cast<const unsigned char*>(wpack) + (int64_t)z * wpack_sub_bytes;
    const int my_tiles = (bid0 < t_hi) ? (t_hi - bid0 + nb - 1) / nb : 0;
    const int total_it = my_tiles * G_chunks;
    const int flags = FAST == 1 ? (d.epi_flags & (CTL_EPI_BIAS | CTL_EPI_STATS)) : ((FAST == 4 || FAST == 5) ? CTL_EPI_STATS : (FAST ? (d.epi_flags & CTL_EPI_BIAS) : d.epi_flags));
    const bool y16 = FAST || (d.dt & CTL_DT_Y16) != 0, r16 = (d.dt & CTL_DT_RES16) != 0;
    const int yes = y16 ? 2 : 4, res_es = r16 ? 2 : 4;
    const int ngroups = d.groups > 1 ? d.groups : 1;
    const int group_n = d.n / ngroups;
    const int oy0 = (z >> 1) * d.out_sub, ox0 = (z & 1) * d.out_sub;
    const __amdgpu_buffer_rsrc_t rx = ctl_rsrc(x, (int64_t)d.n * d.hin * d.win * d.cin * ((XB == 1 || (XB == 0 && (d.dt & CTL_DT_X16))) ? 2 : 4));
    const __amdgpu_buffer_rsrc_t rx2 = X2 ? ctl_rsrc(x2, (int64_t)d.n * d.hin * d.win * d.cin * 2) : rx;
    const bool xout_on = X2 && xout != nullptr && blockIdx.y == 0 && blockIdx.z == 0;
    const __amdgpu_buffer_rsrc_t rxout = xout_on ? ctl_rsrc(xout, (int64_t)d.n * d.hin * d.win * d.cin * 2) : rx;
    const int64_t ypix = (int64_t)d.n * d.out_h * d.out_w * d.cout;
    const __amdgpu_buffer_rsrc_t ry = ctl_rsrc(y, ypix * yes);
    const __amdgpu_buffer_rsrc_t rres = ctl_rsrc(res ? res : y, ypix * (res ? res_es : yes));
    const __amdgpu_buffer_rsrc_t rres2 = FAST == 5 ? ctl_rsrc(res2, ypix * 2) : ry;
    // FAST 5 with `pool` (1x1 hosts whose waves own row pairs): the epilogue also writes sumpool2(g), from the UNROUNDED g (see ctl_conv.hip)
    constexpr bool CAN_POOL = FAST == 5 && KS == 1 && S == 1 && MODE == CTL_IN_PLAIN && (MT / (TW / 16)) == 2;
    const __amdgpu_buffer_rsrc_t rpool = (CAN_POOL && pool) ? ctl_rsrc(pool, ypix * 2 / 4) : ry;

    f32x4 ssum[NT], ssq[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) ssum[t] = ssq[t] = f32x4{0.f, 0.f, 0.f, 0.f};

    constexpr int TWT = TW / 16;
    static_assert(MT % TWT == 0, "a wave's M-tiles must cover whole tile rows");
    const int wrow = wave * (MT / TWT);
    int yrel[MT];                                       // element offset of this lane's 4 channels of M-tile m relative to the tile's output origin
#pragma unroll
    for (int m = 0; m < MT; ++m)
        yrel[m] = ((wrow + m / TWT) * d.out_sy * d.out_w + ((m % TWT) * 16 + p) * d.out_sx) * d.cout + q * 4;
    // 16-byte stores of bf16 results (FAST 1, whole tiles): lane rows q and q^1 exchange halves of two M-tiles (v_permlane16_swap), after
    // which lane (p, q) holds 8 consecutive channels 8*(q>>1).. of pixel p of M-tile 2j + (q & 1): one b128 store instead of two b64
    int yrel2[(MT + 1) / 2];
#pragma unroll
    for (int j = 0; j < (MT + 1) / 2; ++j) {
        const int mt2 = 2 * j + (q & 1);
        yrel2[j] = ((wrow + mt2 / TWT) * d.out_sy * d.out_w + ((mt2 % TWT) * 16 + p) * d.out_sx) * d.cout + (q >> 1) * 8;
    }
    // B-operand addresses: lane group q carries tap 2f + (q >> 1) (clamped to the last tap: its weights are zero there), channels
    // 8*(q & 1) .. +7 of it; (fragment, M-tile) offsets: one VGPR per fragment + compile-time immediates per M-tile
    int xoff[NFRAG];
#pragma unroll
    for (int f = 0; f < NFRAG; ++f) {
        int tap = 2 * f + (q >> 1);
        tap = tap < TAPS ? tap : TAPS - 1;
        const int kh = tap / KS, kw = tap % KS;
        const int kcol = (S == 2) ? ((kw & 1) * G::IWH + (kw >> 1)) : kw;
        xoff[f] = (((wrow * S + kh) * G::IWP + p + kcol) * 16) + (q & 1) * XS::PLANE;
    }
    const unsigned char* wrd = wt + lane * 16;

    XS xs;
    xs.init(d);
    if (KS == 2 && S == 1) {
        xs.pad_h = d.pad == 2 ? 1 - (z >> 1) : 0;
        xs.pad_w = d.pad == 2 ? 1 - (z & 1) : 0;
    }
    // weight chunk g: [fragment][t][64 lanes][16 B]
    const __amdgpu_buffer_rsrc_t rw = ctl_rsrc(wp, (int64_t)ctl_cdiv(d.cout, 16) * NFRAG * G_chunks * 1024);
    u32x4 wv[NW];
    int wrel[NW];
#pragma unroll
    for (int i = 0; i < NW; ++i) {
        const int u = tid + i * 256;
        const int tt = u >> 6, l = u & 63;
        const int f = tt / NT, t = tt - f * NT;
        wrel[i] = (u < WU) ? ((((cot0 + t) * NFRAG + f) * G_chunks) * 64 + l) * 16 : CTL_OOB;
    }
    auto wload = [&](int g) {
#pragma unroll
        for (int i = 0; i < NW; ++i) wv[i] = ctl_bload4u(rw, wrel[i], g * 1024);
    };
    auto wstore = [&]() {
#pragma unroll
        for (int i = 0; i < NW; ++i) {
            const int u = tid + i * 256;
            if (u < WU) *reinterpret_cast<u32x4*>(wt + u * 16) = wv[i];
        }
    };

    TileWalk cur, nxt;
    cur.init(bid0, nb, tiles_h, tiles_w);
    nxt = cur;
    TM_DECL
    if (total_it > 0) {
        xs.load(rx, rx2, d, cur.n, cur.th * G::TH, cur.tw * TW, 0);
        wload(0);
    }
    // The bias and the prologue coefficients are requested BEHIND the first tile's loads, without a branch around any load, and waited for
    // once (see ctl_conv_igemm.h: the bias used to cost a memory round trip of its own at the start of every forward launch).
    f32x4 bias4[NT];
    float bv[NT][4];                 // the raw loads: first USED behind the coefficient requests below
    {
        const bool has_bias = (flags & CTL_EPI_BIAS) != 0;
        const float* bp = has_bias ? bias : reinterpret_cast<const float*>(wpack);
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int co0 = (cot0 + t) * 16 + q * 4;
#pragma unroll
            for (int k = 0; k < 4; ++k) bv[t][k] = bp[has_bias ? (co0 + k < d.cout ? co0 + k : d.cout - 1) : k];
        }
        const int ncf = ngroups * d.cin;       // groups * cin <= CTL_PRO_MAX = 256 = the block's threads: one coefficient entry per thread
        const bool pro_on = X2 || d.pro_affine != 0;
        const bool con = pro_on && tid < ncf;
        const int ci = con ? tid : 0;
        float c0, c1, c2 = 0.f;
        if constexpr (X2) {          // coefficients as the BatchNorm-backward finalize writes them: [group][A | B | C][cin]
            const int gi = ci / d.cin, ch = ci - gi * d.cin;
            c0 = pro_scale[(gi * 3 + 0) * d.cin + ch]; c1 = pro_scale[(gi * 3 + 1) * d.cin + ch]; c2 = pro_scale[(gi * 3 + 2) * d.cin + ch];
        } else {
            const float* ps = pro_on ? pro_scale : reinterpret_cast<const float*>(wpack);
            const float* ph = pro_on ? pro_shift : reinterpret_cast<const float*>(wpack);
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
    if (total_it > 0) {
        xs.store(xt, d, 0, cf_scale, cf_shift, (cur.n / group_n) * d.cin, cf_c, rxout, xout_on);
        wstore();
    }
    __syncthreads();
    // Pipeline (two steps deep): while step `it` computes out of LDS, the registers hold step it+1 (requested one step earlier, staged
    // after the MFMA phase) and step it+2 is requested right behind that staging -- BEFORE the epilogue's stores.  (Requested at the top
    // of the loop, the loads waited ~800 cycles per tile for the previous epilogue's stores: same registers, in-order vmcnt.)
    TileWalk nn = cur;                 // walker of the step whose loads are in flight next
    int gn = 0;                        // its chunk
    auto advance_nn = [&]() { gn = (gn + 1 == G_chunks) ? 0 : gn + 1; if (gn == 0) nn.next(); };
    if (total_it > 1) {
        advance_nn();
        xs.load(rx, rx2, d, nn.n, nn.th * G::TH, nn.tw * TW, gn);
        if (G_chunks > 1) wload(gn);
    }

    // One 16-channel chunk (cin <= 16): the block's weight fragments never change -> read them from LDS ONCE into registers
    // (5 fragments x 4 VGPRs per cout tile for a 3x3 kernel) instead of once per tile.
    constexpr bool WREG = (NFRAG * NT <= 10);
    const bool wreg_on = WREG && G_chunks == 1;
    bf16x8 wreg[WREG ? NFRAG : 1][WREG ? NT : 1];
    if (wreg_on) {
#pragma unroll
        for (int f = 0; f < (WREG ? NFRAG : 1); ++f)
#pragma unroll
            for (int t = 0; t < (WREG ? NT : 1); ++t) wreg[f][t] = *reinterpret_cast<const bf16x8*>(wrd + (f * NT + t) * 1024);
    }
    const int srows = gridDim.x * gridDim.z, srow = z * gridDim.x + blockIdx.x;
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
            if (p == 0) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    sred[((wave * NT + t) * 16 + q * 4 + r) * 2 + 0] = a[r];
                    sred[((wave * NT + t) * 16 + q * 4 + r) * 2 + 1] = a[4 + r];
                }
            }
            ssum[t] = ssq[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
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
    if ((flags & CTL_EPI_STATS) && ngroups > 1 && tid < NT * 16 * 2) {
        const int stat = tid / (NT * 16), co = cot0 * 16 + tid % (NT * 16);
        if (co < d.cout)
            for (int gi = 0; gi < ngroups; ++gi) stats_partial[(((int64_t)gi * srows + srow) * 2 + stat) * d.cout + co] = 0.f;
    }

    TM(7)
    f32x4 acc[MT][NT];
    for (int it = 0, g = 0; it < total_it; ++it) {
        TM_COUNT(6)
        const int n = cur.n, ho0 = cur.th * G::TH, wo0 = cur.tw * TW;
        const bool has_next = it + 1 < total_it;
        const int g2 = (g + 1 == G_chunks) ? 0 : g + 1;
        const bool new_w = has_next && G_chunks > 1;
        if (g2 == 0) nxt.next();
        TM(0)
        if (g == 0) {
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int t = 0; t < NT; ++t) acc[m][t] = bias4[t];
        }
        // operands of fragment f+1 are requested before the MFMAs of fragment f; two instances of the phase: weights from the
        // block's registers (single-chunk problems) or from LDS -- a branch once per tile, not a select per operand
        auto mfma_phase = [&](auto wreg_tag) {
            constexpr bool WR = decltype(wreg_tag)::value;
            // PF fragments of B operands are requested ahead of the MFMAs that consume them: with 16-cycle bf16 MFMAs one fragment of
            // work (MT x NT x 16 cycles) is shorter than an LDS round trip, so a read-ahead of one (the fp32 kernels' scheme) leaves
            // the latency exposed at every fragment (phase timers: 1430 cycles for 20 MFMAs = 320 cycles of matrix pipe)
            // (the scheduler sinks every read back to its use -- ds_read, s_waitcnt lgkmcnt(0), v_mfma, 20 times over, seen in the ISA --
            // unless scheduling barriers pin the read-ahead)
#ifndef CTL16_PF
#define CTL16_PF 2
#endif
            constexpr int PF = (CTL16_PF > NFRAG) ? NFRAG : CTL16_PF;
            constexpr bool PIN = S == 1 && KS != 4 && (NT == 1 || MT == 1);      // (the other instantiations have no registers to spare: 30-110 spills)
            bf16x8 wf[PF][NT], xf[PF][MT];
            auto lds_operands = [&](int f) {
                const int b = f % PF;
                if constexpr (!WR) {
#pragma unroll
                    for (int t = 0; t < NT; ++t) wf[b][t] = *reinterpret_cast<const bf16x8*>(wrd + (f * NT + t) * 1024);
                }
#pragma unroll
                for (int m = 0; m < MT; ++m)
                    xf[b][m] = *reinterpret_cast<const bf16x8*>(xt + xoff[f] + (((m / TWT) * S) * G::IWP + (m % TWT) * 16) * 16);
            };
#pragma unroll
            for (int f = 0; f < PF - 1 && f < NFRAG; ++f) lds_operands(f);
#pragma unroll
            for (int f = 0; f < NFRAG; ++f) {
                const int b = f % PF;
                if (f + PF - 1 < NFRAG) lds_operands(f + PF - 1);
                // pin: the wait for fragment f's operands comes AFTER the requests above (left alone, the scheduler sinks every read to
                // its use: ds_read, s_waitcnt lgkmcnt(0), v_mfma, 20 times per tile = 1430 cycles for 320 cycles of matrix work).  The
                // empty asm "rewrites" the operands, so the MFMAs depend on it, and its memory clobber keeps the reads above it.
#pragma unroll
                for (int m = 0; m < MT; ++m) { if constexpr (PIN) asm volatile("" : "+v"(xf[b][m]) : : "memory"); }
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int t = 0; t < NT; ++t) {
                        if constexpr (WR) acc[m][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wreg[WREG ? f : 0][WREG ? t : 0], xf[b][m], acc[m][t], 0, 0, 0);
                        else acc[m][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[b][t], xf[b][m], acc[m][t], 0, 0, 0);
                    }
            }
        };
        if (!(CTL16_ABLATE & 4)) {
        if constexpr (WREG) {
            if (wreg_on) mfma_phase(std::true_type{});
            else mfma_phase(std::false_type{});
        } else {
            mfma_phase(std::false_type{});
        }
        }
        TM(1)
        ctl_barrier_lds_reads_done();
        TM(2)
        if (has_next) {
            xs.store(xt, d, g2, cf_scale, cf_shift, (nxt.n / group_n) * d.cin, cf_c, rxout, xout_on);
            if (new_w) wstore();
        }
        TM(3)
        ctl_barrier_lds_writes_done();
        TM(4)
        if (it + 2 < total_it) {       // step it+2: its loads have the whole next step to land
            advance_nn();
            if (!(CTL16_ABLATE & 2)) xs.load(rx, rx2, d, nn.n, nn.th * G::TH, nn.tw * TW, gn);
            if (G_chunks > 1) wload(gn);
        }

        if (g == G_chunks - 1) {
            const int grp = n / group_n;
            if ((flags & CTL_EPI_STATS) && grp != cur_grp) {
                flush_stats(cur_grp);
                cur_grp = grp;
            }
            const int ybase = ((n * d.out_h + ho0 * d.out_sy + oy0) * d.out_w + wo0 * d.out_sx + ox0) * d.cout + cot0 * 16;      // elements
            const bool full = ho0 + G::TH <= d.hout && wo0 + TW <= d.wout && (cot0 + NT) * 16 <= d.cout;
            if constexpr (FAST == 4) {
                // CTL_EPI_BNBWD on bf16 tensors: this conv produced dL/da of a = leaky(BN(u)).  Write g = dL/da * leaky'(BN(u)) and take the
                // two BatchNorm-backward sums (sum g, sum g*u) here -- the separate reduction pass over da and u disappears (its launch
                // and one of its two tensor reads; the fp32 family has the same epilogue, where it only broke even: MFMA-bound there)
                f32x4 rs[NT], rh[NT];
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const int co0 = (cot0 + t) * 16 + q * 4;
                    rs[t] = *reinterpret_cast<const f32x4*>(res_scale + grp * d.cout + co0);
                    rh[t] = *reinterpret_cast<const f32x4*>(res_shift + grp * d.cout + co0);
                }
                int bo[MT];
                u32x2 uq[MT][NT];
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                    const bool pv = full || ((ho0 + wrow + m / TWT < d.hout) && (wo0 + (m % TWT) * 16 + p < d.wout));
                    bo[m] = pv ? (ybase + yrel[m]) * 2 : CTL_OOB;
#pragma unroll
                    for (int t = 0; t < NT; ++t) uq[m][t] = ctl_bload2u(rres, bo[m] == CTL_OOB ? CTL_OOB : bo[m] + t * 32, 0);
                }
                const float sl = d.epi_slope;
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int t = 0; t < NT; ++t) {
                        f32x4 v = acc[m][t];
                        const f32x4 u = unpack_bf16x4(uq[m][t].x, uq[m][t].y);
                        const f32x4 sa = u * rs[t] + rh[t];
                        v.x *= sa.x > 0.f ? 1.f : sl; v.y *= sa.y > 0.f ? 1.f : sl;
                        v.z *= sa.z > 0.f ? 1.f : sl; v.w *= sa.w > 0.f ? 1.f : sl;
                        if (bo[m] != CTL_OOB) { ssum[t] += v; ssq[t] += v * u; }
                        __builtin_amdgcn_raw_buffer_store_b64(u32x2{pack_bf16x2(v.x, v.y), pack_bf16x2(v.z, v.w)}, ry,
                                                              bo[m] == CTL_OOB ? CTL_OOB : bo[m] + t * 32, 0, CTL_STORE_AUX);
                    }
            } else if constexpr (FAST == 5) {
                // CTL_EPI_TAILBWD on bf16 tensors: this launch produces dL/dOut of a residual block (+ the half already in y with
                // CTL_EPI_ACCUM).  Write g = dOut * leaky'(out) (res = the stored block output) and take the tail's BatchNorm-backward
                // sums (sum g, sum g*v; res2 = v) from the unrounded g: the reduction pass over dOut, out and v disappears, and dOut
                // is never rounded on its own
                const bool accum = (d.epi_flags & CTL_EPI_ACCUM) != 0;
                int bo[MT];
                u32x2 oq[MT][NT], vq[MT][NT], yq[MT][NT];
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                    const bool pv = full || ((ho0 + wrow + m / TWT < d.hout) && (wo0 + (m % TWT) * 16 + p < d.wout));
                    bo[m] = pv ? (ybase + yrel[m]) * 2 : CTL_OOB;
#pragma unroll
                    for (int t = 0; t < NT; ++t) {
                        const int o = bo[m] == CTL_OOB ? CTL_OOB : bo[m] + t * 32;
                        oq[m][t] = ctl_bload2u(rres, o, 0);
                        vq[m][t] = ctl_bload2u(rres2, o, 0);
                        yq[m][t] = accum ? ctl_bload2u(ry, o, 0) : u32x2{0u, 0u};
                    }
                }
                const float sl = d.epi_slope;
                f32x4 hsum[CAN_POOL ? TWT : 1][CAN_POOL ? NT : 1];       // pair sums of the wave's top row, kept for the bottom row
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int t = 0; t < NT; ++t) {
                        f32x4 v = acc[m][t] + unpack_bf16x4(yq[m][t].x, yq[m][t].y);
                        const f32x4 o = unpack_bf16x4(oq[m][t].x, oq[m][t].y), r2 = unpack_bf16x4(vq[m][t].x, vq[m][t].y);
                        v.x *= o.x > 0.f ? 1.f : sl; v.y *= o.y > 0.f ? 1.f : sl;
                        v.z *= o.z > 0.f ? 1.f : sl; v.w *= o.w > 0.f ? 1.f : sl;
                        if (bo[m] != CTL_OOB) { ssum[t] += v; ssq[t] += v * r2; }
                        __builtin_amdgcn_raw_buffer_store_b64(u32x2{pack_bf16x2(v.x, v.y), pack_bf16x2(v.z, v.w)}, ry,
                                                              bo[m] == CTL_OOB ? CTL_OOB : bo[m] + t * 32, 0, CTL_STORE_AUX);
                        if constexpr (CAN_POOL) {
                            if (pool) {
                                f32x4 hs;
                                hs.x = v.x + __shfl_xor(v.x, 1); hs.y = v.y + __shfl_xor(v.y, 1);
                                hs.z = v.z + __shfl_xor(v.z, 1); hs.w = v.w + __shfl_xor(v.w, 1);
                                if (m < TWT) hsum[m][t] = hs;
                                else {
                                    const f32x4 top = hsum[m - TWT][t];
                                    const f32x4 pl = {top.x + hs.x, top.y + hs.y, top.z + hs.z, top.w + hs.w};
                                    const int lo = ((((n * (d.out_h >> 1) + ((ho0 + wrow) >> 1)) * (d.out_w >> 1) + ((wo0 + (m % TWT) * 16 + p) >> 1)) * d.cout) + (cot0 + t) * 16 + q * 4) * 2;
                                    __builtin_amdgcn_raw_buffer_store_b64(u32x2{pack_bf16x2(pl.x, pl.y), pack_bf16x2(pl.z, pl.w)}, rpool,
                                                                          ((p & 1) == 0 && bo[m] != CTL_OOB) ? lo : CTL_OOB, 0, CTL_STORE_AUX);
                                }
                            }
                        }
                    }
            } else if constexpr (FAST >= 2) {
                // the residual tail  out = LeakyReLU(conv + v * scale + shift)  (FAST 2) and  y += conv  (FAST 3) on bf16 operands: the
                // operand loads of every fragment go out first, then the arithmetic; ragged tiles use CTL_OOB offsets (loads return 0,
                // stores are dropped)
                f32x4 rs[NT], rh[NT];
                if constexpr (FAST == 2) {
#pragma unroll
                    for (int t = 0; t < NT; ++t) {
                        const int co0 = (cot0 + t) * 16 + q * 4;
                        rs[t] = *reinterpret_cast<const f32x4*>(res_scale + grp * d.cout + co0);
                        rh[t] = *reinterpret_cast<const f32x4*>(res_shift + grp * d.cout + co0);
                    }
                }
                int bo[MT];
                u32x2 rq[MT][NT];
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                    const bool pv = full || ((ho0 + wrow + m / TWT < d.hout) && (wo0 + (m % TWT) * 16 + p < d.wout));
                    bo[m] = pv ? (ybase + yrel[m]) * 2 : CTL_OOB;
#pragma unroll
                    for (int t = 0; t < NT; ++t) rq[m][t] = ctl_bload2u(FAST == 2 ? rres : ry, bo[m] == CTL_OOB ? CTL_OOB : bo[m] + t * 32, 0);
                }
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int t = 0; t < NT; ++t) {
                        f32x4 v = acc[m][t];
                        const f32x4 r = unpack_bf16x4(rq[m][t].x, rq[m][t].y);
                        if constexpr (FAST == 2) {
                            v += r * rs[t] + rh[t];
                            if (d.epi_act == CTL_ACT_LEAKY) v = ctl_leaky01(v, d.epi_slope);
                        } else {
                            v += r;
                        }
                        __builtin_amdgcn_raw_buffer_store_b64(u32x2{pack_bf16x2(v.x, v.y), pack_bf16x2(v.z, v.w)}, ry,
                                                              bo[m] == CTL_OOB ? CTL_OOB : bo[m] + t * 32, 0, CTL_STORE_AUX);
                    }
            } else if constexpr (FAST == 1) {
                // lane (p, q) holds channels 4q .. 4q+3 of pixel p of each M-tile: 8 bytes of bf16 per fragment; nothing is read back,
                // nothing branches per fragment; ragged tiles send the dropped lanes to CTL_OOB (hardware bounds check)
                if (full) {
                    if (flags & CTL_EPI_STATS) {
                        asm volatile("" ::: "memory");                  // keeps the branch (not a select per element)
#pragma unroll
                        for (int t = 0; t < NT; ++t)
#pragma unroll
                            for (int m = 0; m < MT; ++m) { ssum[t] += acc[m][t]; ssq[t] += acc[m][t] * acc[m][t]; }
                    }
                    if constexpr ((CTL16_ABLATE & 1) != 0) {
#pragma unroll
                        for (int t = 0; t < NT; ++t)
#pragma unroll
                            for (int m = 0; m < MT; ++m) ssum[t] += acc[m][t];
                    } else if constexpr (MT % 2 == 0 && CTL16_STORE16) {
#pragma unroll
                        for (int t = 0; t < NT; ++t)
#pragma unroll
                            for (int j = 0; j < MT / 2; ++j) {
                                const f32x4 a = acc[2 * j][t], b = acc[2 * j + 1][t];
                                const u32x2 r0 = __builtin_amdgcn_permlane16_swap(pack_bf16x2(a.x, a.y), pack_bf16x2(b.x, b.y), false, false);
                                const u32x2 r1 = __builtin_amdgcn_permlane16_swap(pack_bf16x2(a.z, a.w), pack_bf16x2(b.z, b.w), false, false);
                                __builtin_amdgcn_raw_buffer_store_b128(u32x4{r0.x, r1.x, r0.y, r1.y}, ry, (ybase + yrel2[j] + t * 16) * 2, 0, CTL_STORE_AUX);
                            }
                    } else {
#pragma unroll
                        for (int t = 0; t < NT; ++t)
#pragma unroll
                            for (int m = 0; m < MT; ++m)
                                __builtin_amdgcn_raw_buffer_store_b64(u32x2{pack_bf16x2(acc[m][t].x, acc[m][t].y), pack_bf16x2(acc[m][t].z, acc[m][t].w)},
                                                                      ry, (ybase + yrel[m] + t * 16) * 2, 0, CTL_STORE_AUX);
                    }
                } else {
#pragma unroll
                    for (int t = 0; t < NT; ++t) {
                        const bool cok = (cot0 + t) * 16 + q * 4 < d.cout;
#pragma unroll
                        for (int m = 0; m < MT; ++m) {
                            const bool pv = (ho0 + wrow + m / TWT < d.hout) && (wo0 + (m % TWT) * 16 + p < d.wout);
                            const f32x4 v = acc[m][t];
                            if ((flags & CTL_EPI_STATS) && pv) { ssum[t] += v; ssq[t] += v * v; }
                            const int bo = (pv && cok) ? (ybase + yrel[m] + t * 16) * 2 : CTL_OOB;
                            __builtin_amdgcn_raw_buffer_store_b64(u32x2{pack_bf16x2(v.x, v.y), pack_bf16x2(v.z, v.w)}, ry, bo, 0, CTL_STORE_AUX);
                        }
                    }
                }
            } else {
            auto load4 = [&](__amdgpu_buffer_rsrc_t r, int eoff, bool b16) -> f32x4 {      // 4 channels at element offset eoff (or OOB)
                if (eoff == CTL_OOB) return f32x4{0.f, 0.f, 0.f, 0.f};
                if (b16) { const u32x2 u = ctl_bload2u(r, eoff * 2, 0); return unpack_bf16x4(u.x, u.y); }
                return __builtin_bit_cast(f32x4, ctl_bload4u(r, eoff * 4, 0));
            };
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const int co0 = (cot0 + t) * 16 + q * 4;
                const bool cok = full || co0 < d.cout;
                const int cc = cok ? co0 : 0;
                f32x4 rs = {0.f, 0.f, 0.f, 0.f}, rh = {0.f, 0.f, 0.f, 0.f};
                if (flags & CTL_EPI_RES) {
                    if (d.cout >= 4) {
                        rs = *reinterpret_cast<const f32x4*>(res_scale + grp * d.cout + cc);
                        rh = *reinterpret_cast<const f32x4*>(res_shift + grp * d.cout + cc);
                    } else { rs.x = res_scale[grp]; rh.x = res_shift[grp]; }
                }
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                    const bool pv = full || ((ho0 + wrow + m / TWT < d.hout) && (wo0 + (m % TWT) * 16 + p < d.wout));
                    const int eo = (pv && cok) ? (ybase + yrel[m] + t * 16) : CTL_OOB;
                    f32x4 v = acc[m][t];
                    if (flags & CTL_EPI_RES) {
                        f32x4 rv;
                        if (d.cout >= 4) rv = load4(rres, eo, r16);
                        else rv = f32x4{eo == CTL_OOB ? 0.f : __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rres, eo * 4, 0, 0)), 0.f, 0.f, 0.f};
                        v += rv * rs + rh;
                    }
                    if ((flags & CTL_EPI_STATS) && pv) { ssum[t] += v; ssq[t] += v * v; }
                    if (d.epi_act == CTL_ACT_LEAKY) {
                        v = ctl_leaky01(v, d.epi_slope);
                    } else if (d.epi_act == CTL_ACT_SIGMOID) {
                        v.x = 1.f / (1.f + expf(-v.x)); v.y = 1.f / (1.f + expf(-v.y));
                        v.z = 1.f / (1.f + expf(-v.z)); v.w = 1.f / (1.f + expf(-v.w));
                    }
                    if (flags & CTL_EPI_ACCUM) {
                        if (d.cout >= 4) v += load4(ry, eo, y16);
                        else v.x += eo == CTL_OOB ? 0.f : __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ry, eo * 4, 0, 0));
                    }
                    if (eo != CTL_OOB) {
                        if (d.cout >= 4) {
                            if (y16) __builtin_amdgcn_raw_buffer_store_b64(u32x2{pack_bf16x2(v.x, v.y), pack_bf16x2(v.z, v.w)}, ry, eo * 2, 0, CTL_STORE_AUX);
                            else ctl_bstore4(ry, eo * 4, v);
                        } else {
                            ctl_bstore1(ry, eo * 4, v.x);             // cout == 1 (fp32 network output)
                        }
                    }
                }
            }
            }      // generic epilogue
        }
        TM(5)
        if (g2 == 0) cur = nxt;
        g = g2;
    }
    TM_FLUSH
    if (flags & CTL_EPI_STATS) flush_stats(cur_grp);
}

// ------------------------------------------------------------------------------------------------ weight packing (bf16 fragments)
// Same table records as pack_weights_batched_kernel (modes 0-3; mode 4 = K-packed taps is an fp32-only layout: the bf16 family runs the
// <= 4-channel first layers through the plain path, the tile is HBM-bound either way).  One thread = one uint32 = two bf16 of
// dst[cot][fragment][chunk g][lane][8]: lane (co = lane & 15, q = lane >> 4), element j -> tap 2f + (q >> 1), ci = 16 g + 8 (q & 1) + j.
__device__ float ctl_pack_value(const float* __restrict__ src, const int64_t* __restrict__ r, int co, int ci, int kh, int kw) {
    const int cout = (int)r[2], cin = (int)r[3], ks = (int)r[4], flip = (int)r[5];
    if (co >= cout || ci >= cin) return 0.f;
    float v = 0.f;
    if (r[11] == 1) {
        for (int a = 0; a < 2; ++a) {
            const int sh = a + 2 - kh;
            if (sh < 0 || sh > 2) continue;
            for (int b = 0; b < 2; ++b) {
                const int sw = b + 2 - kw;
                if (sw < 0 || sw > 2) continue;
                v += src[co * r[6] + ci * r[7] + sh * r[8] + sw * r[9]];
            }
        }
    } else if (r[11] == 2) {
        const int a = flip >> 1, b = flip & 1;
        const int h0 = (kh == 0) ? 0 : (a ? 2 : 1), h1 = (kh == 0) ? (a ? 1 : 0) : 2;
        const int w0 = (kw == 0) ? 0 : (b ? 2 : 1), w1 = (kw == 0) ? (b ? 1 : 0) : 2;
        for (int sh = h0; sh <= h1; ++sh)
            for (int sw = w0; sw <= w1; ++sw) v += src[co * r[6] + ci * r[7] + sh * r[8] + sw * r[9]];
    } else if (r[11] == 3) {
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
__global__ void pack_weights_bf16_batched_kernel(const float* __restrict__ params, float* __restrict__ wpack,
                                                 const int64_t* __restrict__ table) {
    const int64_t* r = table + (int64_t)blockIdx.y * 12;
    const int cout = (int)r[2], cin = (int)r[3], ks = (int)r[4];
    if (r[11] == 4 || (r[11] & CTL_PACK_X3)) return;         // (not a bf16 layout)
    const int g_chunks = (cin + 15) / 16, taps = ks * ks, nfrag = (taps + 1) / 2;
    const int64_t total = (int64_t)((cout + 15) / 16) * nfrag * g_chunks * 64 * 4;      // uint32 words
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
        a = ctl_pack_value(src, r, co, ci, tap / ks, tap % ks);
        b = ctl_pack_value(src, r, co, ci + 1, tap / ks, tap % ks);
    }
    dst[idx] = pack_bf16x2(a, b);
}
extern "C" int ctl_pack_weights_bf16_batched(const float* params, float* wpack, const int64_t* table, int32_t n_rec, int64_t max_total,
                                             ctl_stream stream) {
    CTL_REQUIRE(params && wpack && table && n_rec > 0 && max_total > 0, "pack_weights_bf16_batched: bad arguments");
    // `max_total` is the fp32 layout's float count per record, an upper bound of the bf16 layout's uint32 count (5 of 9 taps)
    pack_weights_bf16_batched_kernel<<<dim3((unsigned)ctl_cdiv64(max_total, 256), (unsigned)n_rec), dim3(256), 0, (hipStream_t)stream>>>(
        params, wpack, table);
    CTL_LAUNCH_CHECK("pack_weights_bf16_batched");
    return CTL_OK;
}

// ------------------------------------------------------------------------------------------------ host side
struct conv16_call {
    const ctl_conv* d; ctl_conv_cfg c;
    const void *x, *x2, *wpack, *res, *res2; void *y, *pool, *xout;
    const float *bias, *pro_scale, *pro_shift, *res_scale, *res_shift; float* stats_partial;
    hipStream_t stream; bool query; int grid_x;
};
template <int KS, int S, int MODE, int MT, int TW, int NT, int FAST, int XB, bool X2 = false>
static void conv16_go_f(conv16_call& a) {
    static int occ = 0;
    if (!occ) {
        int n = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, conv_igemm_bf16_kernel<KS, S, MODE, MT, TW, NT, FAST, XB, X2>, 256, 0) != hipSuccess || n < 1) {
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
    conv_igemm_bf16_kernel<KS, S, MODE, MT, TW, NT, FAST, XB, X2><<<grid, dim3(256), 0, a.stream>>>(
        *d, a.x, a.x2, a.wpack, a.bias, a.pro_scale, a.pro_shift, a.res, a.res_scale, a.res_shift, a.res2, a.y, a.stats_partial, a.c.tiles_h, a.c.tiles_w,
        a.c.g, (int64_t)ctl_conv_wpack_floats(d->cin, d->cout, d->ks) * 4, ntiles, a.pool, a.xout);
}
template <int KS, int S, int MODE, int MT, int TW, int NT>
static void conv16_go(conv16_call& a) {
    const ctl_conv* d = a.d;
    // the FAST instantiations (see the kernel): bf16 on both sides, whole channel tiles; 1 = bias / statistics only, 2 = bf16 residual
    // with affine (+ LeakyReLU): the tail of every residual block, 3 = accumulate into the bf16 output (the 1x1 data gradients)
    const bool xb = (d->dt & CTL_DT_X16) && d->cin % 16 == 0;
    const bool o16 = (d->dt & CTL_DT_Y16) && d->cout % 16 == 0;
    const int e = d->epi_flags & (CTL_EPI_RES | CTL_EPI_ACCUM | CTL_EPI_BNBWD | CTL_EPI_STATS);
    int fast = 0;
    if (o16 && !(e & ~CTL_EPI_STATS) && d->epi_act == CTL_ACT_NONE) fast = 1;
    else if (o16 && e == CTL_EPI_RES && (d->dt & CTL_DT_RES16) && (d->epi_act == CTL_ACT_NONE || d->epi_act == CTL_ACT_LEAKY)) fast = 2;
    else if (o16 && e == CTL_EPI_ACCUM && d->epi_act == CTL_ACT_NONE) fast = 3;
    if (d->pro_affine == 2) {      // the BatchNorm-backward prologue: the data-gradient convs that consume a block's dU / dV (checked in ctl_conv_forward_bf16)
        if constexpr ((KS == 3 && S == 1 && MODE == CTL_IN_PLAIN) || (KS == 4 && S == 2)) {
            if (d->epi_flags & CTL_EPI_BNBWD) conv16_go_f<KS, S, MODE, MT, TW, NT, 4, true, true>(a);
            else conv16_go_f<KS, S, MODE, MT, TW, NT, 1, true, true>(a);
        }
        return;
    }
    if (d->epi_flags & CTL_EPI_BNBWD) { conv16_go_f<KS, S, MODE, MT, TW, NT, 4, true>(a); return; }      // (checked in ctl_conv_forward_bf16)
    if (d->epi_flags & CTL_EPI_TAILBWD) {       // the launches that write a block's output gradient (checked in ctl_conv_forward_ex)
        if constexpr ((KS == 1 && MODE == CTL_IN_PLAIN) || KS == 2 || (KS == 3 && S == 1 && MODE == CTL_IN_ZINS2)) {
            if (xb) conv16_go_f<KS, S, MODE, MT, TW, NT, 5, 1>(a);
            else if constexpr (KS == 1) {          // (the 1x1 data gradient of a network's fp32 output layer: 4 or 1 channels)
                if (!(d->dt & CTL_DT_X16) && d->cin >= 4) conv16_go_f<KS, S, MODE, MT, TW, NT, 5, 2>(a);
                else if (!(d->dt & CTL_DT_X16) && d->cin == 1) conv16_go_f<KS, S, MODE, MT, TW, NT, 5, 3>(a);
                else conv16_go_f<KS, S, MODE, MT, TW, NT, 5, 0>(a);
            }
        }
        return;
    }
    if (xb) {
        if (fast == 1) conv16_go_f<KS, S, MODE, MT, TW, NT, 1, true>(a);
        else if (fast == 2) conv16_go_f<KS, S, MODE, MT, TW, NT, 2, true>(a);
        else if (fast == 3) conv16_go_f<KS, S, MODE, MT, TW, NT, 3, true>(a);
        else conv16_go_f<KS, S, MODE, MT, TW, NT, 0, 1>(a);      // (bf16 in, generic epilogue: the fp32 output layers)
    } else if constexpr (KS == 3 && S == 1 && MODE == CTL_IN_PLAIN) {          // fp32 input: first layers of the encoders (4 or 1 channels)
        const int xk = (d->dt & CTL_DT_X16) ? 0 : (d->cin == 1 ? 3 : 2);
        if (fast == 1 && xk == 2) conv16_go_f<KS, S, MODE, MT, TW, NT, 1, 2>(a);
        else if (fast == 1 && xk == 3) conv16_go_f<KS, S, MODE, MT, TW, NT, 1, 3>(a);
        else if (fast == 1) conv16_go_f<KS, S, MODE, MT, TW, NT, 1, 0>(a);
        else if (fast == 2) conv16_go_f<KS, S, MODE, MT, TW, NT, 2, 0>(a);
        else conv16_go_f<KS, S, MODE, MT, TW, NT, 0, 0>(a);
    } else {          // (shapes outside the path)
        if (fast == 1) conv16_go_f<KS, S, MODE, MT, TW, NT, 1, 0>(a);
        else if (fast == 2) conv16_go_f<KS, S, MODE, MT, TW, NT, 2, 0>(a);
        else conv16_go_f<KS, S, MODE, MT, TW, NT, 0, 0>(a);
    }
}
template <int KS, int S, int MODE>
static void conv16_go_tile(conv16_call& a) {
    const bool big = S == 1 && a.c.mt == 4 && a.c.tw == 32;
    if (a.c.nt == 2) {
        if (big) conv16_go<KS, S, MODE, (S == 1 ? 4 : 2), (S == 1 ? 32 : 16), 2>(a);
        else if (a.c.mt == 2) conv16_go<KS, S, MODE, 2, 16, 2>(a);
        else conv16_go<KS, S, MODE, 1, 16, 2>(a);
    } else {
        if (big) conv16_go<KS, S, MODE, (S == 1 ? 4 : 2), (S == 1 ? 32 : 16), 1>(a);
        else if (a.c.mt == 2) conv16_go<KS, S, MODE, 2, 16, 1>(a);
        else conv16_go<KS, S, MODE, 1, 16, 1>(a);
    }
}
static int conv16_dispatch(conv16_call& a) {
    const int k = a.d->ks, s = a.d->stride, m = a.d->in_mode == CTL_IN_C4 ? CTL_IN_PLAIN : a.d->in_mode;
    if (k == 3 && s == 1 && m == CTL_IN_PLAIN) conv16_go_tile<3, 1, CTL_IN_PLAIN>(a);
    else if (k == 3 && s == 1 && m == CTL_IN_UP2) conv16_go_tile<3, 1, CTL_IN_UP2>(a);
    else if (k == 3 && s == 1 && m == CTL_IN_ZINS2) conv16_go_tile<3, 1, CTL_IN_ZINS2>(a);
    else if (k == 3 && s == 2) conv16_go_tile<3, 2, CTL_IN_PLAIN>(a);
    else if (k == 1 && m == CTL_IN_PLAIN) conv16_go_tile<1, 1, CTL_IN_PLAIN>(a);
    else if (k == 1 && m == CTL_IN_UP2) conv16_go_tile<1, 1, CTL_IN_UP2>(a);
    else if (k == 2 && s == 2) conv16_go_tile<2, 2, CTL_IN_PLAIN>(a);
    else if (k == 2 && s == 1) conv16_go_tile<2, 1, CTL_IN_PLAIN>(a);
    else if (k == 4 && s == 2) {
        if (a.c.mt == 2 && a.c.nt == 1) conv16_go<4, 2, CTL_IN_PLAIN, 2, 16, 1>(a);
        else if (a.c.nt == 2) conv16_go<4, 2, CTL_IN_PLAIN, 1, 16, 2>(a);
        else conv16_go<4, 2, CTL_IN_PLAIN, 1, 16, 1>(a);
    } else CTL_FAIL(CTL_EUNSUPPORTED, "conv_forward(bf16): no kernel for this combination");
    return CTL_OK;
}

int ctl_conv_bf16_stats_blocks(const ctl_conv* d) {
    conv16_call a = {};
    a.d = d;
    if (ctl_conv_pick_cfg(d, &a.c, 0) != CTL_OK) return -1;
    a.query = true;
    if (conv16_dispatch(a) != CTL_OK) return -1;
    return a.grid_x * d->nsub;
}

int ctl_conv_forward_bf16(const ctl_conv* d, const void* x, const void* x2, const void* wpack, const float* bias, const float* pro_scale,
                          const float* pro_shift, const void* res, const float* res_scale, const float* res_shift, const void* res2, void* y,
                          float* stats_partial, void* pool, void* xout, ctl_stream stream) {
    if (d->epi_flags & CTL_EPI_TAILBWD) {
        CTL_REQUIRE((d->dt & CTL_DT_Y16) && (d->dt & CTL_DT_RES16) && d->cout % 16 == 0 && ((d->dt & CTL_DT_X16) ? d->cin % 16 == 0 : d->ks == 1),
                    "conv_forward(bf16): CTL_EPI_TAILBWD needs bf16-stored y / res / res2 with whole 16-channel tiles (and a bf16-stored x, except for 1x1 convs)");
    }
    if (d->pro_affine == 2) {
        const int e = d->epi_flags & (CTL_EPI_RES | CTL_EPI_ACCUM | CTL_EPI_BIAS);
        CTL_REQUIRE(x2 && pro_scale && (d->dt & CTL_DT_X16) && (d->dt & CTL_DT_Y16) && d->cin % 16 == 0 && d->cout % 16 == 0 && !e &&
                    d->epi_act == CTL_ACT_NONE && d->in_mode == CTL_IN_PLAIN && ((d->ks == 3 && d->stride == 1) || (d->ks == 4 && d->stride == 2)),
                    "conv_forward(bf16): the BatchNorm-backward prologue (pro_affine 2) needs x2 + coefficients, bf16-stored x / x2 / y with whole "
                    "16-channel tiles, a plain 3x3 stride-1 or 4x4 stride-2 conv and no epilogue operand other than CTL_EPI_STATS / CTL_EPI_BNBWD");
    }
    CTL_REQUIRE(!(d->epi_flags & CTL_EPI_BNBWD) || ((d->dt & CTL_DT_X16) && (d->dt & CTL_DT_Y16) && (d->dt & CTL_DT_RES16) && d->cin % 16 == 0 && d->cout % 16 == 0),
                "conv_forward(bf16): CTL_EPI_BNBWD needs bf16-stored x, y and u with whole 16-channel tiles");
    CTL_REQUIRE(!(d->dt & CTL_DT_X16) || d->cin % 16 == 0, "conv_forward(bf16): bf16-stored inputs need cin %% 16 == 0 (got %d)", d->cin);
    CTL_REQUIRE(!(d->dt & (CTL_DT_Y16 | CTL_DT_RES16)) || d->cout % 4 == 0, "conv_forward(bf16): bf16-stored outputs need cout %% 4 == 0");
    conv16_call a = {};
    a.d = d;
    int rc = ctl_conv_pick_cfg(d, &a.c, 0);
    if (rc != CTL_OK) return rc;
    a.x = x; a.x2 = x2; a.wpack = wpack; a.bias = bias; a.pro_scale = pro_scale; a.pro_shift = pro_shift; a.res = res; a.res_scale = res_scale;
    a.res_shift = res_shift; a.res2 = res2; a.y = y; a.stats_partial = stats_partial; a.pool = pool; a.xout = xout; a.stream = (hipStream_t)stream;
    rc = conv16_dispatch(a);
    if (rc != CTL_OK) return rc;
    CTL_LAUNCH_CHECK("conv_forward(bf16)");
    return CTL_OK;
}

