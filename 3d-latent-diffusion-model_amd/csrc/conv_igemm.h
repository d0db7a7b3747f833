// 3D convolution as an implicit GEMM on the gfx950 matrix cores (bf16 MFMA 16x16x32, fp32 accumulate).
//
// Replaces every nn.Conv3d / nn.Linear the reference reaches through MONAI on the denoising path
// (conv3d 3^3 s1/s2, 1x1 skip convs, attention projections, VAE convs) - the
// reference itself has no kernel code; its call sites are 3d_ldm/train_diffusion.py:197-205 and
// 3d_ldm/inference.py:94-99 (UNet forward inside LatentDiffusionInferer).
//
// Layout: activations NDHWC bf16 (C % 32 == 0), weights [tap][CoutPad][Cin] bf16 (K contiguous).
// One workgroup (4 waves, 256 threads) owns a BM-voxel x BN-cout output tile; each wave a 64x64 sub-tile
// (4x4 MFMA tiles, 64 accumulator VGPRs).  The K loop walks (tap, Cin chunk of BK) steps; per step the
// voxel rows (shifted by the tap, zero page where padded) and the weight rows are copied global->LDS with
// LDS-DMA (global_load_lds_dwordx4, 1 KiB per wave instruction) into a double-buffered, XOR-swizzled
// image that ds_read_b128 reads conflict-free.  Never materialised: zero padding, nearest x2 upsampling
// (source index >> 1), channel concatenation (two source pointers) and the 1x1 skip convolution of a
// ResBlock (appended as extra K steps, "group 1").  MFMA roles: A = weights (rows = cout), B = voxels
// (cols), so one lane ends up with 16 consecutive output channels of one voxel -> 2 x 16-byte stores.
//
// Split-K: blockIdx also enumerates K slices; slices write fp32 slabs that splitk_finalize_kernel sums
// (deterministic, no atomics) before applying the epilogue.
#pragma once
#include "common.h"

struct ConvParams {
    // group 0: k^3 convolution over (x0a | x0b) channel-concatenated sources
    const bf16_t* x0a; const bf16_t* x0b; int c0a, c0b;
    const bf16_t* w0;                 // [taps][CoutPad][c0a + c0b]
    // group 1 (optional, steps1 > 0): 1x1 convolution over (x1a | x1b) at output resolution
    const bf16_t* x1a; const bf16_t* x1b; int c1a, c1b;
    const bf16_t* w1;                 // [CoutPad][c1a + c1b]
    const bf16_t* zero_page;          // zeros, at least one full input row (C * 2 bytes) long
    int N, Din, Hin, Win;             // group-0 source dims (before the optional x2 upsample)
    int Dout, Hout, Wout;
    int ksize, stride, pad, ups;      // ups = 0/1 : nearest-neighbour x2 upsample folded into the loader
    int exact;                        // with ups = 1: tap positions must be EVEN (source = pos / 2), odd ones read zero:
                                      // the transposed (data-gradient) form of a stride-2 convolution
    int M;                            // N * Dout * Hout * Wout
    int CoutS;                        // stored output channels (multiple of 32, >= real Cout)
    int CoutPad;                      // weight rows (multiple of 64 and of BN)
    int CoutReal;                     // channels written in fp32-NCDHW mode
    int nchunk0, nchunk1;             // Cin / BK per group
    int steps0, steps1;               // K steps per group (taps * nchunk0, nchunk1)
    int splitk, steps_per_split;
    int mtiles, ntiles;
    int halo_mtps, q_per_split;       // conv3_halo_kernel only: 126-row tiles per sample, (kd,kh,chunk) macro steps per K split
    int tile_order;                   // 0 = M tiles fastest over the (XCD-contiguous) block order: an XCD streams one weight panel; 1 = cout
                                      // tiles fastest: an XCD owns a contiguous range of output rows (all couts), like the GroupNorm blocks
                                      // that read them next (GnFusedParams::xcd_rows)
    // phase mode (nearest x2 upsample + 3^3 conv, pad 1, as eight 2^3 convolutions on the LOW-resolution grid, one per output
    // parity (pd, ph, pw): 8 taps instead of 27.  ksize = 2, w0 = [parity][tap][CoutPad][cin] with the 3^3 taps that land on
    // the same source voxel pre-summed (phase_weights_kernel).  Tiles are [sample][parity][mtiles_pp] over the source voxels;
    // M, partial slabs, stats and the output rows stay in output order.
    int phase_mode, mtiles_pp;
    // conv3_halo_kernel as the fp32 precision mode's 3 x bf16 product (x3_n > 0 = 64-channel chunks of the REAL Cin): the voxel operand
    // is [rows][hi | lo] bf16 (2 x3_n chunks per row), the weights [tap][cout][hi | lo | hi] (3 x3_n chunks); K chunk c reads voxel
    // chunk c (c < x3_n: hi), c - x3_n (hi again) or c - x3_n (c >= 2 x3_n: lo) against weight chunk c:  hi*Whi + hi*Wlo + lo*Whi.
    // conv_igemm_kernel takes the same product through its two concatenated sources: x0a = the split tensor (hi | lo, c0a = 2 C), x0b = the
    // same pointer again (c0b = C: hi), both with the split tensor's row stride, weights [.. ][hi | hi | lo]:  hi*Whi + lo*Whi + hi*Wlo.
    int x3_n;
    int raw_partial;                  // write the fp32 accumulators to the partial slab even with splitk == 1 (the fp32 finalize follows)
    // exact unsigned division by Hout * Wout and by Wout as multiply-high + shifts (FastDiv; filled by the launchers): the prologues decompose
    // ~14 voxel indices per lane, and a 32-bit division by a run-time value is a ~40-instruction sequence on this ISA
    unsigned fd_hw_m, fd_hw_s, fd_w_m, fd_w_s;
    int slab_lg;                      // split-K slabs PLANAR for the group-owning finalize + GroupNorm (fin_gn.h): 0 = [split][M][CoutPad]; else
                                      // [split][CoutPad >> slab_lg][M][1 << slab_lg]: the channels of one GroupNorm group contiguous over all rows
    float* out32; const float* residual32;   // conv3_halo_kernel, fp32 precision, splitk == 1: fp32 NDHWC output [M][CoutS] (+ fp32 residual) from the fused epilogue
    // epilogue (splitk == 1) ------------------------------------------------------------
    const float* bias;                // [CoutPad] or null
    const float* bias2;               // [CoutPad] or null (bias of the fused 1x1 skip)
    const float* temb; int temb_stride;   // per-sample channel bias [N][temb_stride] or null
    const bf16_t* residual;           // [M][CoutS] or null
    bf16_t* out;                      // [M][CoutS] bf16 NDHWC            (mode 0)
    float* out_f32;                   // [N][CoutReal][Dout*Hout*Wout]    (mode 1)
    float* partial;                   // [splitk][M][CoutPad] fp32 slabs  (splitk > 1)
    float* stats;                     // GroupNorm partials of the OUTPUT: [mtile][CoutS][2] (sum, sum sq over the tile's rows) or null
    unsigned long long* stamps;       // diagnostic (dbg & 512): per-workgroup s_memrealtime stamps [nwg][8]
    int dbg;                          // timing experiments only (LDM_CONV_DBG): 1 = all voxel rows from the zero page, 2 = all weight rows = row 0
};

// address of 4 consecutive couts [c, c + 4) of output row m in split `split` of the fp32 slabs (ConvParams::slab_lg)
__device__ __forceinline__ float* slab_ptr(float* partial, int split, int M, int CoutPad, int lg, int m, int c) {
    float* base = partial + (size_t)split * M * CoutPad;
    if (lg == 0) return base + (size_t)m * CoutPad + c;
    return base + (((size_t)(c >> lg) * M + m) << lg) + (c & ((1 << lg) - 1));
}


// n / d for 32-bit unsigned n and a divisor fixed at launch time (round-up multiplier with an add-shift fix-up, exact for every n):
//   host: s = ceil(log2 d) - 1, m = floor(2^32 (2^(s+1) - d) / d) + 1;  device: q = mulhi(n, m); (((n - q) >> 1) + q) >> s.   d = 1: s = 255.
static inline void fastdiv_make(unsigned d, unsigned* m, unsigned* s) {
    if (d <= 1) { *m = 0; *s = 255u; return; }
    unsigned l = 0; while ((1ull << l) < d) ++l;                  // ceil(log2 d)
    *m = (unsigned)((((1ull << l) - d) << 32) / d + 1); *s = l - 1;
}
__device__ __forceinline__ unsigned fastdiv(unsigned n, unsigned m, unsigned s) {
    if (s == 255u) return n;
    const unsigned q = __umulhi(n, m);
    return (((n - q) >> 1) + q) >> s;
}

// Bijective XCD-aware remap: blocks b, b+8, b+16.. share an XCD (observed round-robin dispatch); give each
// XCD a contiguous range of logical tiles so neighbouring tiles share weights/halo rows in one L2.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    int q = nwg >> 3, r = nwg & 7, x = bid & 7, i = bid >> 3;
    int base = (x < r) ? x * (q + 1) : r * (q + 1) + (x - r) * q;
    return base + i;
}

__device__ __forceinline__ void glds16(const void* gsrc, void* lds_dst) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_dst, 16, 0, 0);
}

template <int WGM, int WGN, int BK, int NS, int NG>
__global__ __launch_bounds__(256 * NG, 2) void conv_igemm_kernel(const ConvParams p) {
#if defined(__HIP_DEVICE_COMPILE__)   // the body uses gfx950-only types (buffer resources); the host pass only needs the stub
    constexpr int BM = 64 * WGM, BN = 64 * WGN;
    constexpr int RB = BK * 2;                 // bytes per LDS row
    constexpr int CPR = RB / 16;               // 16-byte chunks per row (4 | 8)
    constexpr int RPP = 1024 / RB;             // rows per LDS-DMA piece (16 | 8)
    constexpr int NW = 4 * NG;                 // waves per workgroup
    constexpr int PA = BM / RPP / NW;          // voxel pieces per wave
    constexpr int PB = BN / RPP / NW;          // weight pieces per wave
    constexpr int STAGE = (BM + BN) * RB;      // bytes per ring slot
    constexpr int SWZ_SHIFT = (CPR == 8) ? 1 : 2;
    constexpr int KS = BK / 32;                // MFMA k-substeps per K step
    constexpr int PF = NS - 1;                 // K steps in flight beyond the one being computed
    constexpr int LPS = PA + PB;               // LDS-DMA instructions per wave per K step
    static_assert(PA >= 1 && PB >= 1, "tile too small for this many waves");
    static_assert(KS == NG, "each wave group computes exactly one 32-deep k-substep per K step");
    static_assert((PF - 1) * LPS <= 63 && PF * LPS <= 63, "vmcnt is a 6-bit counter");
    static_assert(NG == 1 || NS * STAGE >= 65536 + 4 * BN * 8, "the accumulator exchange needs 64 KiB of LDS (+ the GroupNorm fold)");
    static_assert(NS * STAGE + 28 * BM * 4 <= 160 * 1024, "ring + tap table must fit the 160 KiB LDS");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef __attribute__((address_space(3))) void* lds_ptr_t;

    KSTAMP_BEGIN(7);
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2;                 // wave group: 0 = waves 0-3, 1 = waves 4-7 (second wave of each SIMD)
    const int wq = wave & 3;
    const int wm = wq % WGM, wn = wq / WGM;

    // ---- block -> (split, ntile, mtile); mtile fastest so one XCD streams one weight panel
    const int nwg = gridDim.x;
    int lid = xcd_remap(blockIdx.x, nwg);
    int mtile, ntile, split;
    if (p.tile_order == 1) {                           // cout tiles fastest: an XCD gets a contiguous range of M tiles with all their cout tiles
        ntile = lid % p.ntiles; lid /= p.ntiles; mtile = lid % p.mtiles; split = lid / p.mtiles;
    } else {
        mtile = lid % p.mtiles; lid /= p.mtiles; ntile = lid % p.ntiles; split = lid / p.ntiles;
    }
    int m0 = mtile * BM; const int n0 = ntile * BN;
    int Mloc = p.M;                                    // rows addressed by m0 + row
    int ph_n = 0, ph_d = 0, ph_h = 0, ph_w = 0;        // phase mode: sample and output parity of this tile
    const bf16_t* w0p = p.w0;
    if (p.phase_mode) {
        const int t8 = mtile / p.mtiles_pp, phase = t8 & 7;
        m0 = (mtile - t8 * p.mtiles_pp) * BM; ph_n = t8 >> 3;
        ph_d = phase >> 2; ph_h = (phase >> 1) & 1; ph_w = phase & 1;
        Mloc = p.Din * p.Hin * p.Win;
        w0p += (size_t)phase * 8 * p.CoutPad * (p.c0a + p.c0b);
    }
    const int steps_total = p.steps0 + p.steps1;
    const int s_begin = split * p.steps_per_split;
    int s_end = s_begin + p.steps_per_split; if (s_end > steps_total) s_end = steps_total;
    const int nsteps = s_end - s_begin;
    const int dbg = p.dbg;
    if (dbg & 128) return;                             // timing experiment: launch cost only
#define LDM_STAMP(I) do { if ((dbg & 512) && tid == 0) p.stamps[(size_t)blockIdx.x * 8 + (I)] = __builtin_amdgcn_s_memrealtime(); } while (0)
    LDM_STAMP(0);

    // ---- per-lane loader constants ------------------------------------------------------------
    const int prow = lane / CPR;                       // row inside a piece
    const int pchunk = lane % CPR;                     // physical 16-B chunk this lane fills
    const int DHWo = p.Dout * p.Hout * p.Wout;
    const int HWo = p.Hout * p.Wout;
    const int DinU = p.Din << p.ups, HinU = p.Hin << p.ups, WinU = p.Win << p.ups;
    const int rows_in0 = p.N * p.Din * p.Hin * p.Win;  // voxel rows of the group-0 source

    // (tap, row) -> source voxel table in LDS, built once by the whole workgroup: entry = voxel index of the shifted tap
    // (covers stride, asymmetric padding and the folded x2 upsample: src = (o*stride + k - pad) >> ups) or -1 where
    // the tap falls into the zero padding / the row is beyond M.  A tap change in the K loop is then one ds_read_b32
    // per copied row instead of ~100 serialized instructions on every wave.
    int* const tapv = reinterpret_cast<int*>(smem + NS * STAGE);
    {
        constexpr int TPR = (256 * NG) / BM;           // threads per row
        const int row = tid / TPR, part = tid % TPR;
        const int m = m0 + row;
        const int taps = p.ksize * p.ksize * p.ksize;
        if (row < BM) {
            int n = 0, od = 0, oh = 0, ow = 0;
            if (p.phase_mode) {
                n = ph_n;
                if (m < Mloc) { od = m / (p.Hin * p.Win); int r = m - od * p.Hin * p.Win; oh = r / p.Win; ow = r - oh * p.Win; }
            } else if (m < p.M) { n = m / DHWo; int r = m - n * DHWo; od = r / HWo; r -= od * HWo; oh = r / p.Wout; ow = r - oh * p.Wout; }
            const int nb = n * p.Din * p.Hin * p.Win;
            if (part == 0) tapv[taps * BM + row] = -1;             // sentinel row read by the one-tap-ahead prefetch
            if (p.phase_mode) {
                for (int tap = part; tap < 8; tap += TPR) {        // source voxel = low + tap bit - 1 + parity, per dimension
                    const int id = od + (tap >> 2) - 1 + ph_d, ih = oh + ((tap >> 1) & 1) - 1 + ph_h, iw = ow + (tap & 1) - 1 + ph_w;
                    const bool ok = (m < Mloc) & ((unsigned)id < (unsigned)p.Din) & ((unsigned)ih < (unsigned)p.Hin) &
                                    ((unsigned)iw < (unsigned)p.Win) & !(dbg & 1);
                    tapv[tap * BM + row] = ok ? nb + (id * p.Hin + ih) * p.Win + iw : -1;
                }
            } else
            for (int tap = part; tap < taps; tap += TPR) {
                int kd = 0, kh = 0, kw = 0;
                if (p.ksize == 3) { kd = tap / 9; kh = (tap - kd * 9) / 3; kw = tap - kd * 9 - kh * 3; }
                const int id = od * p.stride + kd - p.pad, ih = oh * p.stride + kh - p.pad, iw = ow * p.stride + kw - p.pad;
                const bool ok = (m < Mloc) & ((unsigned)id < (unsigned)DinU) & ((unsigned)ih < (unsigned)HinU) &
                                ((unsigned)iw < (unsigned)WinU) & !(dbg & 1) & !(p.exact & (id | ih | iw) & 1);
                tapv[tap * BM + row] = ok ? nb + ((id >> p.ups) * p.Hin + (ih >> p.ups)) * p.Win + (iw >> p.ups) : -1;
            }
        }
    }
    int a_row[PA], a_m[PA], a_kb[PA];                  // per copied voxel row: row in tile, output row (group 1), chunk byte
#pragma unroll
    for (int j = 0; j < PA; ++j) {
        const int row = (wave * PA + j) * RPP + prow;
        const int swz = (row >> SWZ_SHIFT) & (CPR - 1);
        a_kb[j] = (pchunk ^ swz) * 16;                 // byte offset of this lane's logical chunk inside the BK chunk
        a_row[j] = row;
        a_m[j] = (m0 + row < Mloc && !(dbg & 1)) ? m0 + row : -1;
    }
    int b_row[PB], b_kb[PB];
#pragma unroll
    for (int j = 0; j < PB; ++j) {
        const int R = (wave * PB + j) * RPP + prow;    // LDS row inside the weight tile
        const int swz = (R >> SWZ_SHIFT) & (CPR - 1);
        b_kb[j] = (pchunk ^ swz) * 16;
        // LDS row (64q + 16nt + i) holds cout 64q + 16(i>>2) + 4nt + (i&3): after the MFMA a lane owns 16
        // consecutive couts (accumulator row 4g+r of tile nt  <->  cout 16g + 4nt + r).
        const int q = R >> 6, nt = (R >> 4) & 3, i = R & 15;
        b_row[j] = ((dbg & 2) ? 0 : n0 + 64 * q + 16 * (i >> 2) + 4 * nt + (i & 3));
    }

    // ---- loader: the K loop is a sequence of SEGMENTS = (group, tap, concat source) runs of cs/BK steps.  Inside a
    // segment every copy is  buffer_load_dwordx4 ... lds  with a per-lane byte offset that stays fixed (voxel row of
    // the shifted tap; 0xFFFFFFFF for zero padding: the buffer range check then writes zeros) and a SCALAR offset
    // that advances by BK*2 bytes per step, so a step costs one m0 write + one instruction per piece.  Three tiers
    // keep the rare work rare: per GROUP (descriptors, weight row offsets), per TAP (voxel index from the packed
    // candidates: ~10 VALU per row), per SOURCE (row byte offset = voxel * channels).
    const unsigned x3s2 = p.x3_n ? (unsigned)p.x3_n * 256u : 0u;   // x3 form: both sources are views of one [rows][2 C] bf16 tensor (row stride 4 C bytes)
    int sg_grp, sg_tap, sg_src, sg_left;                           // scalar segment state
    int g_ca = 0, g_cb = 0;                                        // channels of the current group's two sources
    unsigned g_wtap = 0;                                           // bytes between two taps of the weight tensor
    unsigned soff_a = 0, soff_b = 0;
    int a_v[PA];                                                   // source voxel of the current tap (-1: padding)
    unsigned a_vo[PA], b_vo[PB];
    __amdgpu_buffer_rsrc_t rs_a, rs_a0, rs_a1, rs_b;
    int ld_s = s_begin;
    int first_cis;                                                 // chunk inside the source where this K range starts
    {
        int chunk;
        if (ld_s < p.steps0) {
            sg_grp = 0; sg_tap = ld_s / p.nchunk0; chunk = ld_s - sg_tap * p.nchunk0;
        } else { sg_grp = 1; sg_tap = 0; chunk = ld_s - p.steps0; }
        const int ca = sg_grp ? p.c1a : p.c0a;
        sg_src = (chunk * BK >= ca) ? 1 : 0;
        first_cis = sg_src ? chunk - ca / BK : chunk;
        sg_left = -1;                                              // "state not set up yet"
    }

#define LDM_GROUP_SETUP() do {                                                                                \
        g_ca = sg_grp ? p.c1a : p.c0a; g_cb = sg_grp ? p.c1b : p.c0b;                                         \
        const int cin_ = g_ca + g_cb;                                                                         \
        const unsigned rows_ = sg_grp ? (unsigned)p.M : (unsigned)rows_in0;                                   \
        rs_a0 = __builtin_amdgcn_make_buffer_rsrc((void*)(sg_grp ? p.x1a : p.x0a), 0, (int)(rows_ * (x3s2 ? x3s2 : (unsigned)g_ca * 2u)), 0x00020000); \
        rs_a1 = __builtin_amdgcn_make_buffer_rsrc((void*)(sg_grp ? p.x1b : p.x0b), 0, (int)(rows_ * (x3s2 ? x3s2 : (unsigned)g_cb * 2u)), 0x00020000); \
        g_wtap = (unsigned)p.CoutPad * (unsigned)cin_ * 2u;                                                   \
        rs_b = __builtin_amdgcn_make_buffer_rsrc((void*)(sg_grp ? p.w1 : w0p), 0,                            \
                                                 (int)((unsigned)(sg_grp ? 1 : p.ksize * p.ksize * p.ksize) * g_wtap), 0x00020000); \
        _Pragma("unroll") for (int j = 0; j < PB; ++j)                                                        \
            b_vo[j] = (unsigned)b_row[j] * (unsigned)(cin_ * 2) + (unsigned)b_kb[j];                          \
    } while (0)
#define LDM_TAP_SETUP() do {                                                                                  \
        _Pragma("unroll") for (int j = 0; j < PA; ++j)                                                        \
            a_v[j] = (sg_grp == 0) ? tapv[sg_tap * BM + a_row[j]] : a_m[j];                                   \
    } while (0)
#define LDM_SRC_SETUP() do {                                                                                  \
        const unsigned cs2_ = x3s2 ? x3s2 : (unsigned)(sg_src ? g_cb : g_ca) * 2u;   /* bytes per source row */ \
        rs_a = sg_src ? rs_a1 : rs_a0;                                                                        \
        _Pragma("unroll") for (int j = 0; j < PA; ++j)                                                        \
            a_vo[j] = (a_v[j] >= 0) ? (unsigned)a_v[j] * cs2_ + (unsigned)a_kb[j] : 0xFFFFFFFFu;              \
    } while (0)
    // move to the segment that contains step ld_s (called when the current one is exhausted, or not yet set up)
#define LDM_SEG_ADVANCE() do {                                                                                \
        if (sg_left <= 0) {                                                                                   \
            if (sg_left < 0) {                                         /* first call: enter mid-segment */     \
                LDM_GROUP_SETUP(); LDM_TAP_SETUP(); LDM_SRC_SETUP();                                          \
                soff_a = (unsigned)first_cis * (BK * 2);                                                      \
                soff_b = (unsigned)sg_tap * g_wtap + (unsigned)((sg_src ? g_ca : 0) + first_cis * BK) * 2u;   \
                sg_left = (sg_src ? g_cb : g_ca) / BK - first_cis;                                            \
            } else if (sg_src == 0 && g_cb > 0) {                      /* second concat source, same tap */    \
                sg_src = 1; LDM_SRC_SETUP();                                                                  \
                soff_a = 0; sg_left = g_cb / BK;                       /* soff_b simply keeps running */       \
            } else {                                                   /* next tap (or the fused 1x1 group) */ \
                sg_src = 0;                                                                                   \
                if (sg_grp == 0 && ld_s >= p.steps0) { sg_grp = 1; sg_tap = 0; LDM_GROUP_SETUP(); }           \
                else ++sg_tap;                                                                                  \
                LDM_TAP_SETUP(); LDM_SRC_SETUP();                                                             \
                soff_a = 0; soff_b = (unsigned)sg_tap * g_wtap; sg_left = g_ca / BK;                          \
            }                                                                                                 \
        }                                                                                                     \
    } while (0)
    // the LPS copies of step ld_s (no control flow: this is part of the hot block)
#define LDM_DMA_RAW() do {                                                                                    \
        {                                                                                                     \
            char* st_ = smem + ((ld_s - s_begin) % NS) * STAGE;                                               \
            _Pragma("unroll") for (int j = 0; j < PA; ++j)                                                    \
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_a, (lds_ptr_t)(st_ + (wave * PA + j) * 1024), 16, \
                                                         a_vo[j], soff_a, 0, 0);                              \
            _Pragma("unroll") for (int j = 0; j < PB; ++j)                                                    \
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_b, (lds_ptr_t)(st_ + BM * RB + (wave * PB + j) * 1024), \
                                                         16, b_vo[j], soff_b, 0, 0);                          \
        }                                                                                                     \
        soff_a += BK * 2; soff_b += BK * 2; --sg_left; ++ld_s;                                                \
    } while (0)
#define LDM_DMA() do {                                                                                        \
        if (!(dbg & 4)) LDM_DMA_RAW();                                                                        \
        else { soff_a += BK * 2; soff_b += BK * 2; --sg_left; ++ld_s; }                                       \
    } while (0)
    // issue the copies of step ld_s into ring slot (ld_s - s_begin) % NS and advance the scalar state
#define LDM_ISSUE() do { LDM_SEG_ADVANCE(); LDM_DMA(); } while (0)

    // ---- fragment reads: one 32-deep k-substep per wave group.  Row swizzle bits do not depend on the 16-row tile
    //      index t, so tile t sits at a constant +t*16*RB from the wave's base address (immediate offsets). ------
    const int fr = lane & 15, fg = lane >> 4;
    const int ra0 = wm * 64 + fr, rb0 = wn * 64 + fr;
    const int cfrag = grp * 4 + fg;                     // 16-byte chunk of this lane's 8 k-values (ks = grp)
    const int a_rd0 = ra0 * RB + ((cfrag ^ ((ra0 >> SWZ_SHIFT) & (CPR - 1))) << 4);
    const int b_rd0 = BM * RB + rb0 * RB + ((cfrag ^ ((rb0 >> SWZ_SHIFT) & (CPR - 1))) << 4);

    f32x4 acc[4][4];                                   // [cout tile nt][voxel tile mt]
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

    bf16x8 wfA[4], afA[4], wfB[4], afB[4];             // two fragment sets: one feeds the MFMAs while the other loads

#define LDM_READ_FRAGS_RAW(WF, AF, SLOT) do {                                                       \
        {                                                                                           \
            const char* sb_ = smem + (SLOT) * STAGE;                                                \
            _Pragma("unroll") for (int t = 0; t < 4; ++t) {                                         \
                WF[t] = *reinterpret_cast<const bf16x8*>(sb_ + b_rd0 + t * 16 * RB);                \
                AF[t] = *reinterpret_cast<const bf16x8*>(sb_ + a_rd0 + t * 16 * RB);                \
            }                                                                                       \
        }                                                                                           \
    } while (0)
#define LDM_READ_FRAGS(WF, AF, SLOT) do { if (!(dbg & 16)) LDM_READ_FRAGS_RAW(WF, AF, SLOT); } while (0)
#define LDM_MFMA16_RAW(WF, AF) do {                                                                 \
        {                                                                                           \
            _Pragma("unroll") for (int nt = 0; nt < 4; ++nt)                                        \
                _Pragma("unroll") for (int mt = 0; mt < 4; ++mt)                                    \
                    acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(WF[nt], AF[mt], acc[nt][mt], 0, 0, 0); \
        }                                                                                           \
    } while (0)
#define LDM_MFMA16(WF, AF) do { if (!(dbg & 8)) LDM_MFMA16_RAW(WF, AF); } while (0)
    // Step S+1 ready: counted vmcnt (hipcc does not track LDS-DMA, every wait on it is hand written; a plain
    // __syncthreads() would drain vmcnt(0) and with it the whole prefetch ring), then a RAW barrier: every wave's
    // copies of step S+1 are visible and every wave holds its fragments of step S in registers, so ring slot S % NS
    // may be refilled.  The empty asm statements pin LDS reads behind the barrier.
#define LDM_NEXT_STEP_READY(S) do {                                                                 \
        if ((S) + PF < s_end) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((PF - 1) * LPS) : "memory"); \
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                       \
        __builtin_amdgcn_s_barrier();                                                               \
        asm volatile("" ::: "memory");                                                              \
    } while (0)
    // One K step: CUR fragments are (being) loaded, NXT get loaded behind the barrier while CUR feeds the MFMAs.
    // lgkmcnt(0) goes through the builtin (0xC07F = lgkmcnt 0 only) so that hipcc's wait insertion KNOWS set CUR
    // has arrived and does not re-wait behind the reads of NXT.  Group 0 issues its copies before its MFMAs, group 1
    // after: on every SIMD one wave is in its matrix phase while its partner is in its copy phase.
#define LDM_HALF(S, WC, AC, WN, AN) do {                                                            \
        LDM_ST2(S, 0);                                                                              \
        __builtin_amdgcn_s_waitcnt(0xC07F);                                                         \
        if ((S) + 1 < s_end) {                                                                      \
            LDM_NEXT_STEP_READY(S);                                                                 \
            if (grp == 0 && ld_s < s_end) LDM_ISSUE();                                              \
            LDM_READ_FRAGS(WN, AN, ((S) + 1 - s_begin) % NS);                                       \
        }                                                                                           \
        LDM_MFMA16(WC, AC);                                                                         \
        if (grp == 1 && (S) + 1 < s_end && ld_s < s_end) LDM_ISSUE();                              \
    } while (0)

    LDM_STAMP(1);
    KSTAMP(1);
    if (dbg & 256) return;                             // timing experiment: setup only
    // ---- prologue: fill the ring (NS steps in flight), wait for the first step ----------------------------------
    __syncthreads();                                   // tap table complete
    if (nsteps > 0) {
#pragma unroll
        for (int i = 0; i < NS; ++i) if (i < nsteps) LDM_ISSUE();
        if (nsteps > PF) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PF * LPS) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        LDM_READ_FRAGS(wfA, afA, 0);
    }
    LDM_STAMP(2);
    KSTAMP(2);
    // Steady state: one hot basic block per K step.  The segment change (rare, VALU heavy) runs BEFORE the wait; the
    // copies of step s+NS, the fragment reads of step s+1 and the 16 MFMAs of step s then sit in one scheduling
    // region and sched_group_barrier interleaves them (an MFMA occupies the matrix pipe for 16 cycles but the issue
    // port for ~8: one copy / LDS read rides in the shadow of each MFMA instead of in front of all of them).
#define LDM_ST2(S, K) do { if ((dbg & 1024) && tid == 0 && (S) - s_begin < 120) p.stamps[(size_t)gridDim.x * 8 + (size_t)blockIdx.x * 512 + ((S) - s_begin) * 4 + (K)] = __builtin_amdgcn_s_memrealtime(); } while (0)
    // Branch-free: valid while the loader stays inside group 0 with a single source (every 3^3 conv of the UNet: the
    // concat is materialised by GroupNorm; the fused 1x1 skip steps at the end run through the generic loop below).
    // Tap changes are handled by selects: a_nx[] always holds the row offsets of tap f_tap+1 (read from the LDS table
    // one step ahead), the scalar offsets wrap with s_cselect.
#define LDM_FAST_TAIL(WC, AC, WN, AN, S) \
        LDM_READ_FRAGS_RAW(WN, AN, ((S) + 1 - s_begin) % NS);                                       \
        LDM_MFMA16_RAW(WC, AC);                                                                     \
        _Pragma("unroll") for (int i_ = 0; i_ < LPS; ++i_) {                                        \
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   /* 1 MFMA */                       \
            __builtin_amdgcn_sched_group_barrier(0x004, 2, 0);   /* m0 / offset SALU */             \
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);   /* 1 LDS-DMA copy (VMEM read) */   \
        }                                                                                           \
        _Pragma("unroll") for (int i_ = 0; i_ < 8; ++i_) {                                          \
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   /* 1 MFMA */                       \
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   /* 1 ds_read_b128 */               \
        }                                                                                           \
        __builtin_amdgcn_sched_group_barrier(0x008, 16 - LPS - 8 > 0 ? 16 - LPS - 8 : 0, 0);
#define LDM_FAST_SYNC(S) \
        __builtin_amdgcn_s_waitcnt(0xC07F);                                                         \
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((PF - 1) * LPS) : "memory");                       \
        LDM_ST2(S, 2);                                                                              \
        __builtin_amdgcn_s_barrier();                                                               \
        asm volatile("" ::: "memory");                                                              \
        LDM_ST2(S, 3);
#define LDM_FAST_COPIES() \
            char* st_ = smem + ((ld_s - s_begin) % NS) * STAGE;                                     \
            _Pragma("unroll") for (int j = 0; j < PA; ++j)                                          \
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_a, (lds_ptr_t)(st_ + (wave * PA + j) * 1024), 16, \
                                                         a_vo[j], soff_a, 0, 0);                    \
            _Pragma("unroll") for (int j = 0; j < PB; ++j)                                          \
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_b, (lds_ptr_t)(st_ + BM * RB + (wave * PB + j) * 1024), \
                                                         16, b_vo[j], soff_b, 0, 0);
    // Variant A (few chunks per tap, e.g. the 64/128-channel VAE convs): tap changes by selects inside the hot block.
#define LDM_FAST_HALF(S, WC, AC, WN, AN) do {                                                       \
        LDM_ST2(S, 0);                                                                              \
        LDM_FAST_SYNC(S)                                                                            \
        {                                                                                           \
            LDM_FAST_COPIES()                                                                       \
            const bool last_ = (sg_left == 1);                                                      \
            soff_a = last_ ? 0u : soff_a + BK * 2;                                                  \
            soff_b += BK * 2 + (last_ ? f_wjump : 0u);                                              \
            sg_left = last_ ? f_nch : sg_left - 1;                                                  \
            sg_tap += last_ ? 1 : 0;                                                                \
            ++ld_s;                                                                                 \
            _Pragma("unroll") for (int j = 0; j < PA; ++j) {                                        \
                /* a_vr[] was read from the table one step ago (no LDS wait inside the hot block) */ \
                const unsigned nx_ = (a_vr[j] >= 0) ? (unsigned)a_vr[j] * f_cs2 + (unsigned)a_kb[j] : 0xFFFFFFFFu; \
                a_vo[j] = last_ ? nx_ : a_vo[j];                                                    \
                a_vr[j] = tapv[(sg_tap + 1) * BM + a_row[j]];                                       \
            }                                                                                       \
        }                                                                                           \
        LDM_FAST_TAIL(WC, AC, WN, AN, S)                                                            \
    } while (0)
    // Variant B (many chunks per tap, the UNet convs): the hot block carries nothing extra; a tap change is one short
    // uniform branch in front of the waits (table read + multiply-add per copied row).
#define LDM_FAST_HALF_B(S, WC, AC, WN, AN) do {                                                     \
        LDM_ST2(S, 0);                                                                              \
        if (sg_left == 0) {                                                                         \
            ++sg_tap; soff_a = 0; soff_b = (unsigned)sg_tap * g_wtap; sg_left = f_nch;              \
            _Pragma("unroll") for (int j = 0; j < PA; ++j) {                                        \
                const int v_ = tapv[sg_tap * BM + a_row[j]];                                        \
                a_vo[j] = (v_ >= 0) ? (unsigned)v_ * f_cs2 + (unsigned)a_kb[j] : 0xFFFFFFFFu;       \
            }                                                                                       \
        }                                                                                           \
        LDM_FAST_SYNC(S)                                                                            \
        {                                                                                           \
            LDM_FAST_COPIES()                                                                       \
            soff_a += BK * 2; soff_b += BK * 2; --sg_left; ++ld_s;                                  \
        }                                                                                           \
        LDM_FAST_TAIL(WC, AC, WN, AN, S)                                                            \
    } while (0)

    int s = s_begin;
    const int f_lim = (s_end < p.steps0) ? s_end : p.steps0;      // fast loop: only group-0 steps are issued
    if ((dbg & ~(512 | 1024 | 3)) == 0 && p.c0b == 0 && s + 1 + NS < f_lim) {
        LDM_SEG_ADVANCE();                                          // make sure a segment is set up (sg_left >= 1)
        const unsigned f_cs2 = (unsigned)g_ca * 2u;
        const unsigned f_wjump = g_wtap - (unsigned)g_ca * 2u;     // from the end of a tap's channel run to the next tap
        const int f_nch = g_ca / BK;
        if (f_nch <= 3) {
            int a_vr[PA];                                           // voxel of each copied row under tap sg_tap + 1
#pragma unroll
            for (int j = 0; j < PA; ++j) a_vr[j] = tapv[(sg_tap + 1) * BM + a_row[j]];
            while (s + 1 + NS < f_lim) {
                LDM_FAST_HALF(s, wfA, afA, wfB, afB);
                LDM_FAST_HALF(s + 1, wfB, afB, wfA, afA);
                s += 2;
            }
        } else {
            while (s + 1 + NS < f_lim) {
                LDM_FAST_HALF_B(s, wfA, afA, wfB, afB);
                LDM_FAST_HALF_B(s + 1, wfB, afB, wfA, afA);
                s += 2;
            }
        }
        if (ld_s >= p.steps0) { sg_left = 0; sg_tap = p.ksize * p.ksize * p.ksize - 1; }   // group 0 exhausted: let the generic advance switch groups
    }
    for (; s < s_end; s += 2) {                       // generic loop: prologue-sized K ranges, dual sources, fused skip, tail
        LDM_HALF(s, wfA, afA, wfB, afB);
        if (s + 1 >= s_end) break;
        LDM_HALF(s + 1, wfB, afB, wfA, afA);
    }
#undef LDM_FAST_HALF
#undef LDM_FAST_HALF_B
#undef LDM_FAST_TAIL
#undef LDM_FAST_SYNC
#undef LDM_FAST_COPIES
#undef LDM_HALF
#undef LDM_NEXT_STEP_READY
#undef LDM_MFMA16
#undef LDM_READ_FRAGS
#undef LDM_ISSUE
#undef LDM_DMA
#undef LDM_DMA_RAW
#undef LDM_READ_FRAGS_RAW
#undef LDM_MFMA16_RAW
#undef LDM_SEG_ADVANCE
#undef LDM_GROUP_SETUP
#undef LDM_TAP_SETUP
#undef LDM_SRC_SETUP

    LDM_STAMP(3);
    KSTAMP(3);
    if (dbg & 64) return;                              // timing experiment: no reduction / epilogue
    // ---- intra-workgroup K reduction: group g keeps voxel tiles mt in {2g, 2g+1} and receives its partner's
    //      partial sums for them through LDS (the ring is dead by now). --------------------------------------------
    constexpr int MTN = (NG == 2) ? 2 : 4;           // 16-row tiles this wave finishes
    const int mt_base = (NG == 2) ? 2 * grp : 0;
    if constexpr (NG == 2) {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __syncthreads();
        float* xch = reinterpret_cast<float*>(smem);
        const int dst = 1 - grp;                       // partner group owns mt in {2*dst, 2*dst+1}
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int ml = 0; ml < 2; ++ml)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    xch[(((dst * 4 + wq) * 32) + (nt * 2 + ml) * 4 + r) * 64 + lane] =
                        (grp == 0) ? acc[nt][2 + ml][r] : acc[nt][ml][r];
        __syncthreads();
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int ml = 0; ml < 2; ++ml)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float v = xch[(((grp * 4 + wq) * 32) + (nt * 2 + ml) * 4 + r) * 64 + lane];
                    if (grp == 0) acc[nt][ml][r] += v; else acc[nt][2 + ml][r] += v;
                }
    }

    LDM_STAMP(4);
    KSTAMP(4);
    // ---- epilogue -------------------------------------------------------------------------------
    // Per lane: 16 consecutive couts of one voxel per 16-row tile.  Optionally also the GroupNorm partial sums of
    // the (bf16-rounded) output over each 32-row block, written to a slab (no atomics -> bitwise reproducible).
    const int cbase = n0 + wn * 64 + 16 * fg;          // this lane's 16 consecutive couts
    const bool to_slab = p.splitk > 1 || p.raw_partial;
    const bool do_stats = (p.stats != nullptr) && !to_slab && (p.out != nullptr);
    float ssum[16], ssq[16];                           // GroupNorm partials of this wave's rows (whole tile after the LDS fold)
#pragma unroll
    for (int q = 0; q < 16; ++q) { ssum[q] = 0.f; ssq[q] = 0.f; }
#pragma unroll
    for (int pr = 0; pr < MTN / 2; ++pr) {             // pairs of 16-row tiles
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
            const int ml = pr * 2 + hh;
            int m = m0 + wm * 64 + (mt_base + ml) * 16 + fr;
            if (m >= Mloc) continue;
            int n_ph = 0;
            if (p.phase_mode) {                        // source voxel of this tile's parity -> output row
                const int dl = m / (p.Hin * p.Win); int r2 = m - dl * p.Hin * p.Win; const int hl = r2 / p.Win, wl = r2 - hl * p.Win;
                m = ((ph_n * p.Dout + 2 * dl + ph_d) * p.Hout + 2 * hl + ph_h) * p.Wout + 2 * wl + ph_w;
                n_ph = ph_n;
            }
            float v[16];
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if constexpr (NG == 2) v[nt * 4 + r] = (grp == 0) ? acc[nt][ml][r] : acc[nt][2 + ml][r];
                    else v[nt * 4 + r] = acc[nt][ml][r];
                }
            if (to_slab) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 t4 = make_float4(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
                    *reinterpret_cast<float4*>(slab_ptr(p.partial, split, p.M, p.CoutPad, p.slab_lg, m, cbase + 4 * q)) = t4;
                }
                continue;
            }
            const int n = p.phase_mode ? n_ph : m / DHWo;
            if (p.bias) {
#pragma unroll
                for (int q = 0; q < 16; ++q) v[q] += p.bias[cbase + q];
            }
            if (p.bias2) {
#pragma unroll
                for (int q = 0; q < 16; ++q) v[q] += p.bias2[cbase + q];
            }
            if (p.temb) {                              // rows are padded to CoutPad by the host: no bounds check
                const float* te = p.temb + (size_t)n * p.temb_stride + cbase;
#pragma unroll
                for (int q = 0; q < 16; ++q) v[q] += te[q];
            }
            if (p.out_f32) {
                const int sp = m - n * DHWo;
#pragma unroll
                for (int q = 0; q < 16; ++q)
                    if (cbase + q < p.CoutReal) p.out_f32[((size_t)n * p.CoutReal + cbase + q) * DHWo + sp] = v[q];
                continue;
            }
            if (cbase >= p.CoutS) continue;            // weight-row padding beyond the stored channels
            if (p.residual) {
                const u32x4* rp = reinterpret_cast<const u32x4*>(p.residual + (size_t)m * p.CoutS + cbase);
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const u32x4 rv = rp[h];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        v[h * 8 + 2 * q] += __uint_as_float(rv[q] << 16);
                        v[h * 8 + 2 * q + 1] += __uint_as_float(rv[q] & 0xffff0000u);
                    }
                }
            }
            u32x4* op = reinterpret_cast<u32x4*>(p.out + (size_t)m * p.CoutS + cbase);
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                u32x4 o;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    o[q] = pack2bf(v[h * 8 + 2 * q], v[h * 8 + 2 * q + 1]);
                    const float lo = __uint_as_float(o[q] << 16), hi = __uint_as_float(o[q] & 0xffff0000u);
                    ssum[h * 8 + 2 * q] += lo; ssq[h * 8 + 2 * q] += lo * lo;
                    ssum[h * 8 + 2 * q + 1] += hi; ssq[h * 8 + 2 * q + 1] += hi * hi;
                }
                op[h] = o;
            }
        }
    }
    if (do_stats) {
        // sum over the 16 voxel lanes of each DPP row (lanes sharing fg): rotate-and-add within the row
#define LDM_ROW_ADD(X, CTRL) X += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(X), CTRL, 0xf, 0xf, true))
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            LDM_ROW_ADD(ssum[q], 0x128); LDM_ROW_ADD(ssum[q], 0x124); LDM_ROW_ADD(ssum[q], 0x122); LDM_ROW_ADD(ssum[q], 0x121);
            LDM_ROW_ADD(ssq[q], 0x128); LDM_ROW_ADD(ssq[q], 0x124); LDM_ROW_ADD(ssq[q], 0x122); LDM_ROW_ADD(ssq[q], 0x121);
        }
#undef LDM_ROW_ADD
        // fold the row blocks of the tile (WGM wave rows x NG wave groups) through LDS -> ONE slab row per tile
        constexpr int NRB = WGM * NG;
        float* red = reinterpret_cast<float*>(smem + ((NG == 2) ? 65536 : 0));         // [NRB][BN][2]
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __syncthreads();                                   // the ring (and its last fragment reads) is dead in every wave
        if (fr == 0) {
            float* d = red + (((wm * NG + grp) * BN) + wn * 64 + 16 * fg) * 2;
#pragma unroll
            for (int q = 0; q < 16; ++q) { d[2 * q] = ssum[q]; d[2 * q + 1] = ssq[q]; }
        }
        __syncthreads();
        if (tid < BN && n0 + tid < p.CoutS) {
            float s0 = 0.f, s1 = 0.f;
#pragma unroll
            for (int b = 0; b < NRB; ++b) { s0 += red[(b * BN + tid) * 2]; s1 += red[(b * BN + tid) * 2 + 1]; }
            *reinterpret_cast<float2*>(p.stats + ((size_t)mtile * p.CoutS + n0 + tid) * 2) = make_float2(s0, s1);
        }
    }
    LDM_STAMP(5);
    KSTAMP(5);
    KSTAMP_DRAIN(6);
#undef LDM_STAMP
#endif  // __HIP_DEVICE_COMPILE__
}

// Sums split-K slabs and applies the same epilogue as the fused path.  One block per 32 output rows (the GroupNorm
// statistics granule); thread = (row lane, 8 channels).
struct FinalizeParams {
    const float* partial; int splitk; int M; int CoutPad; int CoutS; int CoutReal; int DHWo;
    const float* bias; const float* bias2; const float* temb; int temb_stride; const bf16_t* residual;
    bf16_t* out; float* out_f32; float* stats;
};

// bx / by = the block's 32-row granule and
// 64-channel slice; o_keep returns the thread's packed bf16 output (row bx * 32 + tid / 8, channels by * 64 + (tid & 7) * 8 ...).
template <bool WT>
__device__ __forceinline__ void splitk_finalize_body(const FinalizeParams& p, const int bx, const int by, float (*red)[8][16], u32x4& o_keep KSTAMP_PARAM) {
    // block = 32 rows (bx) x 64 channels (by); thread = one row x 8 channels
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int cv = tid & 7, rl = tid >> 3;
    const int c = by * 64 + cv * 8;
    const int m = bx * 32 + rl;
    const bool do_stats = p.stats != nullptr && p.out != nullptr;
    float ss[8], sq[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) { ss[q] = 0.f; sq[q] = 0.f; }
    if (m < p.M && c < p.CoutS) {
        float v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = 0.f;
        // Round 3 measured the alternatives on the whole step (same box): all operands requested up front with 16 or 32 slabs in flight
        // per thread is 1 - 1.3 % SLOWER, and unconditional clamped batches of 8 (no 4-batch, no serial tail) 1.6 % slower (a split
        // factor of 9 then loads 16 slabs): the launch is bound by the slab bytes crossing XCDs, not by its round trips.
        // slabs are summed in order 0, 1, 2, ... (bitwise reproducible); four slabs' loads are in flight at a time so the
        // sum is not one L2 round trip per slab
        const size_t slab = (size_t)p.M * p.CoutPad;
        const float* src0 = p.partial + (size_t)m * p.CoutPad + c;
        int s = 0;
        for (; s + 8 <= p.splitk; s += 8) {              // eight slabs' loads in flight (the launch is latency bound)
            float4 a[8], b[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const float4* src = reinterpret_cast<const float4*>(src0 + (size_t)(s + u) * slab);
                a[u] = src[0]; b[u] = src[1];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                v[0] += a[u].x; v[1] += a[u].y; v[2] += a[u].z; v[3] += a[u].w;
                v[4] += b[u].x; v[5] += b[u].y; v[6] += b[u].z; v[7] += b[u].w;
            }
        }
        for (; s + 4 <= p.splitk; s += 4) {
            float4 a[4], b[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float4* src = reinterpret_cast<const float4*>(src0 + (size_t)(s + u) * slab);
                a[u] = src[0]; b[u] = src[1];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                v[0] += a[u].x; v[1] += a[u].y; v[2] += a[u].z; v[3] += a[u].w;
                v[4] += b[u].x; v[5] += b[u].y; v[6] += b[u].z; v[7] += b[u].w;
            }
        }
        for (; s < p.splitk; ++s) {
            const float4* src = reinterpret_cast<const float4*>(src0 + (size_t)s * slab);
            const float4 a = src[0], b = src[1];
            v[0] += a.x; v[1] += a.y; v[2] += a.z; v[3] += a.w;
            v[4] += b.x; v[5] += b.y; v[6] += b.z; v[7] += b.w;
        }
        const int n = m / p.DHWo;
        KSTAMP(1);
        if (p.bias) {
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] += p.bias[c + q];
        }
        if (p.bias2) {
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] += p.bias2[c + q];
        }
        if (p.temb) {
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] += p.temb[(size_t)n * p.temb_stride + c + q];
        }
        if (p.out_f32) {
            const int sp = m - n * p.DHWo;
#pragma unroll
            for (int q = 0; q < 8; ++q)
                if (c + q < p.CoutReal) p.out_f32[((size_t)n * p.CoutReal + c + q) * p.DHWo + sp] = v[q];
        } else {
            if (p.residual) {
                const u32x4 rv = *reinterpret_cast<const u32x4*>(p.residual + (size_t)m * p.CoutS + c);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    v[2 * q] += __uint_as_float(rv[q] << 16);
                    v[2 * q + 1] += __uint_as_float(rv[q] & 0xffff0000u);
                }
            }
            u32x4 o;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                o[q] = pack2bf(v[2 * q], v[2 * q + 1]);
                const float lo = __uint_as_float(o[q] << 16), hi = __uint_as_float(o[q] & 0xffff0000u);
                ss[2 * q] = lo; sq[2 * q] = lo * lo; ss[2 * q + 1] = hi; sq[2 * q + 1] = hi * hi;
            }
            KSTAMP(2);
            store16<WT>(p.out + (size_t)m * p.CoutS + c, o);
            o_keep = o;
        }
    }
    if (do_stats) {     // fold the 32 rows: 8 row lanes per wave by shuffles (lane = row*8 + cv), 4 waves through LDS
#pragma unroll
        for (int q = 0; q < 8; ++q) {
#pragma unroll
            for (int o = 8; o < 64; o <<= 1) { ss[q] += __shfl_xor(ss[q], o, 64); sq[q] += __shfl_xor(sq[q], o, 64); }
        }
        if (lane < 8) {
#pragma unroll
            for (int q = 0; q < 8; ++q) { red[wave][lane][q] = ss[q]; red[wave][lane][8 + q] = sq[q]; }
        }
        __syncthreads();
        if (tid < 8 && c < p.CoutS) {
            float* dst = p.stats + ((size_t)bx * p.CoutS + c) * 2;
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                dst[2 * q] = red[0][tid][q] + red[1][tid][q] + red[2][tid][q] + red[3][tid][q];
                dst[2 * q + 1] = red[0][tid][8 + q] + red[1][tid][8 + q] + red[2][tid][8 + q] + red[3][tid][8 + q];
            }
        }
    }
}

template <bool WT>
__global__ __launch_bounds__(256) void splitk_finalize_kernel(const FinalizeParams p) {
    KSTAMP_BEGIN(4);
    __shared__ float red[4][8][16];
    u32x4 o;
    splitk_finalize_body<WT>(p, blockIdx.x, blockIdx.y, red, o KSTAMP_ARG);
    KSTAMP(3);
    KSTAMP_DRAIN(4);
}
