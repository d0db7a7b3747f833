// 3x3x3 stride-1 "same" convolution, W-halo reuse, ALTERNATING K STEPS PER WAVE GROUP (gfx950, bf16 MFMA 16x16x32).
// EXPERIMENTS BUILDS ONLY (make EXTRA=-DLDM_EXPERIMENTS, LDM_HALO_PP=1): measured 38 % slower per K step than conv3_halo_kernel, see the end of this comment.
//
// Same job, data layout, tile (126 output voxels x 128 couts) and epilogue as conv3_halo_kernel<6> (conv_halo.h; the reference's call sites
// are the nn.Conv3d modules MONAI builds for 3d_ldm/train_diffusion.py:197-205 / 3d_ldm/inference.py:94-99).  Second answer to what
// profiles/r05_halo_ablations.txt measured (lock-step behind a per-step barrier; conv_halo_rw.h was the first and lost to half cache lines
// through the vector L1 and to the copies clustered behind its barrier):
//   * a SUPER step = two (kd, kh, Cin chunk) macro steps = two 16 KiB voxel tiles = six 64-deep K steps.  Wave group 0 (waves 0 - 3) takes K
//     steps (tile 0, kw 0), (tile 0, kw 2), (tile 1, kw 1); group 1 takes (0, 1), (1, 0), (1, 2).  A wave owns ALL 128 tile rows x 32 couts
//     of its steps' full 64-deep K, so its weight fragments are WHOLE 128-byte rows of the weight tensor (4 x buffer_load_dwordx4 per wave
//     and step, three register sets, two own steps ahead) -- no LDS, nobody to share them with;
//   * ONE workgroup barrier per super step (per six K steps); between barriers the two waves of a SIMD (one of each group) run free;
//   * the two tiles of the NEXT super step travel by LDS-DMA into the other slot, issued in the MIDDLE of the super step (position 1), not
//     behind the barrier, 4 pieces per wave and super step;
//   * no special cases: a macro step outside this workgroup's K range is a tile of zeros (copies issued out of range zero-fill) multiplied
//     by weights from a clamped, valid address, so odd K ranges and the last super step run the same instruction stream;
//   * W-border masks cost no VALU on the data: a masked (lane, 16-row tile) reads a ZERO ROW appended to every tile instead of its voxel row;
//     the 3 x 8 fragment addresses per lane are worked out once.
// LDS: slot s, tile t at s * 64 KiB + t * 16512 (16 KiB + the zero row); tap table and statistics fold in the gap; the K-group exchange
// aliases the slots after the loop.  98560 bytes, one workgroup per CU, 2 waves per SIMD, 256 VGPRs each.
// Covers: Cin % 64 == 0, CoutPad % 128 == 0, bf16 NDHWC output or split-K slabs, GroupNorm partials.  NOT covered: the fused 1x1 skip,
// the tall tile, the tile loop, fp32 outputs, the 3 x bf16 product of the fp32 precision mode (conv3_halo_kernel keeps them).
//
// MEASURED (round 5, parity-green on the 93 conv operator tests; profiles/r05_halo_pp_ablations.txt; sustained 256 -> 256 at 24^3, cycles per K
// step, 512 = MFMA issue, conv3_halo_kernel on the same box 823): 1133.  What the ablations say:
//   MFMAs alone 522, + the barrier 533 (one barrier per six K steps costs nothing by itself); + the 4 tile copies per wave and super step 586;
//   fragment reads + barrier, no weights, no copies: 625 (725 before the reads were moved under the previous half step's MFMAs -- the two
//   waves of a SIMD run the same stream and stall together, so a wave must hide its own reads); everything but the weight loads 677;
//   weight loads issued out of range (instructions, no bytes) 846; with their bytes 1133, and 1045 - 1137 whatever else is switched off.
// So the kernel is bound by 16 KiB of weights per K step arriving through the vector L1: 16 KiB / 475 ns = 34.5 GB/s per CU, the same wall
// conv_halo_rw.h hit with half cache lines -- whole lines did not move it.  LDS-DMA takes 21.3 KiB per 361 ns = 59 GB/s per CU through the same
// L2 in conv3_halo_kernel; and even with free weight bytes this form (846) would not beat it (823).  Register-fed weights are closed.
#pragma once
#include "conv_igemm.h"

constexpr int HPP_TILE = 16384 + 128;                             // a voxel tile + its zero row
constexpr int HPP_TOFF = 2 * HPP_TILE;                            // tap table (9 x 128 ints), then the statistics fold (2 KiB)
constexpr int HPP_ROFF = HPP_TOFF + 9 * 128 * 4;
constexpr int HPP_LDS = 65536 + 2 * HPP_TILE;
static_assert(HPP_ROFF + 2048 <= 65536, "LDS layout");

template <int ABL = 0>
__global__ __launch_bounds__((ABL & 1024) ? 768 : 512, (ABL & 1024) ? 3 : 2) void conv3_halo_pp_kernel(const ConvParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int BM = 128, TM = BM - 2, BN = 128, BK = 64, RB = 128;
    constexpr int AT = HPP_TILE;
    constexpr int PA = BM / 64;                                // 1 KiB LDS-DMA pieces per wave and voxel tile
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef __attribute__((address_space(3))) void* lds_ptr_t;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // ABL & 1024 (timing experiment, 768 threads): waves 8 - 11 are PRODUCERS, one per SIMD -- they issue every tile copy of the workgroup
    // (a quarter each) plus, per K step, 4 more 1 KiB copies each out of the weight tensor into a scratch area (the 16 KiB of weights per
    // step conv3_halo_kernel moves by LDS-DMA: 21.3 pieces per K step in all) and no MFMA; in the epilogue they shadow waves 0 - 3
    constexpr bool PROD = (ABL & 1024) != 0;
    const bool producer = PROD && wave >= 8;
    const int cwave = producer ? wave - 8 : wave;
    const int grp = cwave >> 2, wn = cwave & 3;                // K-step group; 32-cout quarter of the tile
    const int nwg = p.mtiles * p.ntiles * p.splitk;
    const int DHW = p.Dout * p.Hout * p.Wout, HW = p.Hout * p.Wout;
    const int nch = p.nchunk0, Q = 9 * nch;                    // macro steps: (kd, kh) x Cin chunk
    const unsigned cin2 = (unsigned)p.c0a * 2u;                // bytes per voxel row = bytes per weight row
    const int dbgflag = p.dbg;

    // ---- tile
    int mtile, ntile, split;
    {
        int lid = xcd_remap((int)blockIdx.x, nwg);
        if (p.tile_order == 1) { ntile = lid % p.ntiles; lid /= p.ntiles; mtile = lid % p.mtiles; split = lid / p.mtiles; }
        else { mtile = lid % p.mtiles; lid /= p.mtiles; ntile = lid % p.ntiles; split = lid / p.ntiles; }
    }
    const int n0 = ntile * BN;
    const int smp = mtile / p.halo_mtps;
    const int l0 = (mtile - smp * p.halo_mtps) * TM;           // first output voxel of the tile inside its sample
    const int m_base = smp * DHW + l0;
    const int q_begin = split * p.q_per_split;
    int q_end = q_begin + p.q_per_split; if (q_end > Q) q_end = Q;
    q_end = __builtin_amdgcn_readfirstlane(q_end);
    const int nsuper = (q_end - q_begin + 1) >> 1;

    int* const tab = reinterpret_cast<int*>(smem + HPP_TOFF);  // (pair, LDS row) -> source voxel, built in the prologue

    // ---- the group's three K steps of a super step: (tile, kw) at positions 0, 1, 2
    const int T1 = grp;                                        // tiles: {0, grp, 1}
    const int KW0 = grp, KW1 = grp ? 0 : 2, KW2 = grp ? 2 : 1;

    // ---- voxel tile loader (LDS-DMA): every wave copies PA pieces (8 rows x 128 B) of a tile; 16-byte chunks XOR-swizzled by (row & 7)
    const int prow = lane >> 3, pchunk = lane & 7;
    __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc((void*)p.x0a, 0, (int)((unsigned)(p.N * DHW) * cin2), 0x00020000);
    const unsigned wtap = (unsigned)p.CoutPad * cin2;          // bytes between two taps of the weight tensor
    __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)p.w0, 0, (int)(27u * wtap), 0x00020000);
    int P0 = __builtin_amdgcn_readfirstlane(q_begin / nch);    // (pair, chunk) of the first macro step of the current super step
    int C0 = __builtin_amdgcn_readfirstlane(q_begin) - P0 * nch;
    int q0 = __builtin_amdgcn_readfirstlane(q_begin);
    // macro step q0 + QREL -> (pair p_, chunk c_); QREL <= 3
#define PP_PC(QREL)                                                                                 \
        int c_ = C0 + (QREL), p_ = P0;                                                              \
        _Pragma("unroll") for (int i_ = 0; i_ < 3; ++i_) if (c_ >= nch) { c_ -= nch; ++p_; }
    // the tile of macro step q0 + QREL -> LDS at byte offset DST (zeros when the step is outside the K range)
#define PP_ISSUE_TILE(QREL, DST) do {                                                               \
        if (ABL & 256) break;                                                                       \
        if (PROD) {                                                                                 \
            if (producer) {                                                                         \
                PP_PC(QREL)                                                                         \
                const bool live_ = (q0 + (QREL) < q_end) && !(ABL & 4);                             \
                const int pc_ = p_ < 9 ? p_ : 8;                                                    \
                int v4_[4];                                                                         \
                _Pragma("unroll") for (int j = 0; j < 4; ++j) v4_[j] = tab[pc_ * BM + (cwave * 4 + j) * 8 + prow]; \
                __builtin_amdgcn_sched_barrier(0);                                                  \
                _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                     \
                    const int row_ = (cwave * 4 + j) * 8 + prow;                                    \
                    const unsigned vo_ = (live_ && v4_[j] >= 0) ? (unsigned)v4_[j] * cin2 + (unsigned)((pchunk ^ (row_ & 7)) * 16) : 0xFFFFFFFFu; \
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_a, (lds_ptr_t)(smem + (DST) + (cwave * 4 + j) * 1024), 16, vo_, (unsigned)__builtin_amdgcn_readfirstlane(c_ * (BK * 2)), 0, 0); \
                }                                                                                   \
            }                                                                                       \
            break;                                                                                  \
        }                                                                                           \
        PP_PC(QREL)                                                                                 \
        const bool live_ = (q0 + (QREL) < q_end) && !(ABL & 4);                                     \
        const int pc_ = p_ < 9 ? p_ : 8;                                                            \
        _Pragma("unroll") for (int j = 0; j < PA; ++j) {                                            \
            const int row_ = (wave * PA + j) * 8 + prow;                                            \
            const int v_ = tab[pc_ * BM + row_];                                                    \
            const unsigned vo_ = (live_ && v_ >= 0) ? (unsigned)v_ * cin2 + (unsigned)((pchunk ^ (row_ & 7)) * 16) : 0xFFFFFFFFu; \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_a, (lds_ptr_t)(smem + (DST) + (wave * PA + j) * 1024), 16, vo_, (unsigned)__builtin_amdgcn_readfirstlane(c_ * (BK * 2)), 0, 0); \
        }                                                                                           \
    } while (0)

    // ---- weight fragments (A operand) from global memory: row fr of cout tile nt <-> cout n0 + 32 wn + 8 (fr >> 2) + 4 nt + (fr & 3), so that
    //      after the MFMA a lane owns 8 CONSECUTIVE couts of one voxel; the lane's 16 bytes of K half ks sit at fg * 16 + ks * 64 of the row
    const int fr = lane & 15, fg = lane >> 4;
    unsigned w_vo[2];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        const int co = n0 + 32 * wn + 8 * (fr >> 2) + 4 * nt + (fr & 3);
        w_vo[nt] = ((ABL & 32) || (dbgflag & 2)) ? 0xFFFFFFFFu : (unsigned)co * cin2 + (unsigned)(fg * 16);
    }
    bf16x8 wf[3][2][2];
    if (ABL & 128) {
#pragma unroll
        for (int a = 0; a < 12; ++a) wf[a / 4][(a >> 1) & 1][a & 1] = (bf16x8){0, 0, 0, 0, 0, 0, 0, 0};
    }
#define PP_WLOAD(SET, QREL, KW) do {                                                                \
        if (PROD) {                                            /* called 3 x per super step: 8 scratch copies each = 4 per K step */ \
            if (producer && !(ABL & 2)) {                                                           \
                PP_PC(QREL)                                                                         \
                if (p_ > 8) p_ = 8;                                                                 \
                _Pragma("unroll") for (int j = 0; j < 8; ++j) {                                     \
                    const unsigned so_ = (unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned)(p_ * 3 + (j % 3)) * wtap + (unsigned)c_ * (BK * 2))); \
                    const unsigned vo_ = (ABL & 32) ? 0xFFFFFFFFu : (unsigned)(n0 + (cwave * 8 + j) * 8 + (lane >> 3)) * cin2 + (unsigned)((lane & 7) * 16); \
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lds_ptr_t)(smem + 100 * 1024 + (cwave * 8 + j) * 1024), 16, vo_, so_, 0, 0); \
                }                                                                                   \
            }                                                                                       \
            break;                                                                                  \
        }                                                                                           \
        if (ABL & 128) break;                                                                       \
        PP_PC(QREL)                                                                                 \
        if (p_ > 8) p_ = 8;                                    /* past the end: a valid row, multiplied by a tile of zeros */ \
        const unsigned so_ = (unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned)(p_ * 3 + (KW)) * wtap + (unsigned)c_ * (BK * 2))); \
        _Pragma("unroll") for (int nt = 0; nt < 2; ++nt)                                            \
            _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                        \
                wf[SET][nt][ks] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rs_w, (int)(w_vo[nt] + ks * 64), (int)so_, 0)); \
    } while (0)

    f32x4 acc[2][8];
    bf16x8 xfA[8], xfB[8];
#define PP_STAMP(I) do { if ((dbgflag & 512) && tid == 0) {                                         \
        p.stamps[(size_t)blockIdx.x * 8 + 2 * (I)] = __builtin_amdgcn_s_memrealtime();             \
        p.stamps[(size_t)blockIdx.x * 8 + 2 * (I) + 1] = __builtin_amdgcn_s_memtime(); } } while (0)

    // ---- prologue: weights of positions 0 and 1, the tap table, the zero rows, the fragment addresses, then the first two tiles
    PP_WLOAD(0, 0, KW0);
    PP_WLOAD(1, T1, KW1);
    {
        constexpr int NPART = 512 / BM;
        const int row = tid % BM, part = tid / BM;
        const int l = l0 - 1 + row;                            // LDS row <-> output voxel l0 - 1 + row (kw = 1 tap)
        const bool ok = (l >= 0 && l < DHW);
        int od = 0, oh = 0, ow = 0;
        if (ok) {
            od = (int)fastdiv((unsigned)l, p.fd_hw_m, p.fd_hw_s); const int r = l - od * HW;
            oh = (int)fastdiv((unsigned)r, p.fd_w_m, p.fd_w_s); ow = r - oh * p.Wout;
        }
        for (int pr = part; pr < 9; pr += NPART) {
            const int id = od + pr / 3 - 1, ih = oh + pr % 3 - 1;
            int v = -1;
            if (ok && (unsigned)id < (unsigned)p.Din && (unsigned)ih < (unsigned)p.Hin) {
                v = smp * DHW + (id * p.Hin + ih) * p.Win + ow;
                if (dbgflag & 1) v &= 1023;
            }
            tab[pr * BM + row] = v;
        }
        if (tid < 128) *reinterpret_cast<int*>(smem + (tid >> 6) * 65536 + ((tid >> 5) & 1) * AT + 16384 + (tid & 31) * 4) = 0;
    }
    // fragment addresses (B operand): lane = (row fr of 16-row tile t shifted by kw, 16-byte k chunk ks * 4 + fg); relative to the slot, K half
    // 0; a border-masked (lane, t) points at the tile's zero row.  XOR 64 selects K half 1, XOR 0x10000 slot 1.
    int pre[3][8];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int kw = i == 0 ? KW0 : (i == 1 ? KW1 : KW2);
        const int ti = i == 0 ? 0 : (i == 1 ? T1 : 1);
        const int rd = (fr + kw) * RB + ((fg ^ ((fr + kw) & 7)) << 4);
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const int l = l0 + t * 16 + fr;
            const int ow = l - (int)fastdiv((unsigned)l, p.fd_w_m, p.fd_w_s) * p.Wout;
            const bool masked = (kw == 0 && ow == 0) || (kw == 2 && ow == p.Wout - 1);
            pre[i][t] = ti * AT + (masked ? 16384 + fg * 16 : rd + t * 16 * RB);
        }
    }
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 8; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // table and zero rows written
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    PP_ISSUE_TILE(0, 0);
    PP_ISSUE_TILE(1, AT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    PP_STAMP(0);

    int xslot = 0;
    // A HALF step = 16 MFMAs over one 32-deep K half of an own step, with the 8 fragment reads of the NEXT half riding between them (two
    // fragment buffers: xfA holds K half 0, xfB K half 1).  pre[][] carries the slot bit; K half 1 = address XOR 64.
    typedef const __attribute__((address_space(3))) bf16x8* lds_frag_t;
#define PP_READ8(XF, I, KS) do {                                                                    \
        if (!(ABL & 16)) _Pragma("unroll") for (int t = 0; t < 8; ++t) XF[t] = *(lds_frag_t)(unsigned)(pre[I][t] ^ ((KS) * 64)); \
    } while (0)
#define PP_MFMA16(I, KS, XF) do {                                                                   \
        if (PROD && producer) break;                                                                \
        if (!(ABL & 8)) _Pragma("unroll") for (int mt = 0; mt < 8; ++mt)                            \
            _Pragma("unroll") for (int nt = 0; nt < 2; ++nt)                                        \
                acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[I][nt][KS], XF[mt], acc[nt][mt], 0, 0, 0); \
    } while (0)
    // scheduling pattern of a half step: NV memory instructions behind the first MFMAs, then (address XOR), read, 2 MFMAs
#define PP_ILV(NV, KSN) do {                                                                        \
        _Pragma("unroll") for (int i_ = 0; i_ < (NV); ++i_) {                                       \
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                      \
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);                                      \
        }                                                                                           \
        _Pragma("unroll") for (int i_ = 0; i_ < 8 - (NV); ++i_) {                                   \
            if (KSN) __builtin_amdgcn_sched_group_barrier(0x002, 1, 0);                             \
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                      \
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);                                      \
        }                                                                                           \
        _Pragma("unroll") for (int i_ = 0; i_ < (NV); ++i_) {                                       \
            if (KSN) __builtin_amdgcn_sched_group_barrier(0x002, 1, 0);                             \
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                      \
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                      \
        }                                                                                           \
    } while (0)

    PP_READ8(xfA, 0, 0);
    for (int S = 0; S < nsuper; ++S) {
        // position 0: weights of position 2 (this super step).  The scheduling fences keep the loads where they are written: two own steps
        // ahead of their use (left alone, the scheduler sinks them next to their MFMAs to save registers)
        __builtin_amdgcn_sched_barrier(0);
        PP_WLOAD(2, 1, KW2);
        PP_READ8(xfB, 0, 1);
        PP_MFMA16(0, 0, xfA);
        PP_ILV(4, 1);
        __builtin_amdgcn_sched_barrier(0);
        PP_READ8(xfA, 1, 0);
        PP_MFMA16(0, 1, xfB);
        PP_ILV(0, 0);
        __builtin_amdgcn_sched_barrier(0);
        // position 1: the next super step's tiles into the other slot, weights of position 0 of the next super step
        PP_ISSUE_TILE(2, (xslot ^ 0x10000));
        PP_ISSUE_TILE(3, (xslot ^ 0x10000) + AT);
        __builtin_amdgcn_sched_barrier(0);
        PP_WLOAD(0, 2, KW0);
        PP_READ8(xfB, 1, 1);
        PP_MFMA16(1, 0, xfA);
        PP_ILV(4, 1);
        __builtin_amdgcn_sched_barrier(0);
        PP_READ8(xfA, 2, 0);
        PP_MFMA16(1, 1, xfB);
        PP_ILV(0, 0);
        __builtin_amdgcn_sched_barrier(0);
        // position 2: weights of position 1 of the next super step
        PP_WLOAD(1, 2 + T1, KW1);
        PP_READ8(xfB, 2, 1);
        PP_MFMA16(2, 0, xfA);
        PP_ILV(4, 1);
        __builtin_amdgcn_sched_barrier(0);
        // in front of the LAST half step of the super step: this wave's fragment reads of the slot are complete and its copies of the next
        // tiles have landed (8 younger weight loads may stay in flight); behind the barrier that holds for every wave, so the first
        // fragments of the next super step are read from the other slot under the last 16 MFMAs
        if (PROD) asm volatile("s_waitcnt vmcnt(16) lgkmcnt(0)" ::: "memory");    /* producers: the tile copies landed, the 16 younger scratch copies fly */
        else asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
        if (!(ABL & 64)) __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        xslot ^= 0x10000;
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int t = 0; t < 8; ++t) pre[i][t] ^= 0x10000;
        __builtin_amdgcn_sched_barrier(0);
        PP_READ8(xfA, 0, 0);
        PP_MFMA16(2, 1, xfB);
        PP_ILV(0, 0);
        __builtin_amdgcn_sched_barrier(0);
        q0 += 2; C0 += 2;
#pragma unroll
        for (int i_ = 0; i_ < 2; ++i_) if (C0 >= nch) { C0 -= nch; ++P0; }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // trailing (zero-fill) copies and weight loads: the exchange below reuses the slots
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    PP_STAMP(1);

    // ---- K-group reduction: group g keeps 16-row tiles 4 g .. 4 g + 3 and receives its partner's partial sums for them
    const int cbase = n0 + 32 * wn + 8 * fg;                   // this lane's 8 consecutive couts
    const bool to_slab = p.splitk > 1 || p.raw_partial;
    float4 ebias[2], etemb[2]; u32x4 eres[4];
#pragma unroll
    for (int q = 0; q < 2; ++q) { ebias[q] = make_float4(0.f, 0.f, 0.f, 0.f); etemb[q] = ebias[q]; }
#pragma unroll
    for (int ml = 0; ml < 4; ++ml) eres[ml] = (u32x4){0u, 0u, 0u, 0u};
    const int mt_base = 4 * grp;
    if (!to_slab) {                                            // the epilogue's operands: one more round trip, hidden by the exchange
        if (p.bias) {
#pragma unroll
            for (int q = 0; q < 2; ++q) ebias[q] = *reinterpret_cast<const float4*>(p.bias + cbase + 4 * q);
        }
        if (p.bias2) {
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const float4 b2 = *reinterpret_cast<const float4*>(p.bias2 + cbase + 4 * q);
                ebias[q].x += b2.x; ebias[q].y += b2.y; ebias[q].z += b2.z; ebias[q].w += b2.w;
            }
        }
        if (p.temb) {
            const float* te = p.temb + (size_t)smp * p.temb_stride + cbase;
#pragma unroll
            for (int q = 0; q < 2; ++q) etemb[q] = *reinterpret_cast<const float4*>(te + 4 * q);
        }
        if (p.residual && cbase < p.CoutS) {
#pragma unroll
            for (int ml = 0; ml < 4; ++ml) {
                const int r_t = (mt_base + ml) * 16 + fr;
                if (r_t < TM && l0 + r_t < DHW) eres[ml] = *reinterpret_cast<const u32x4*>(p.residual + (size_t)(m_base + r_t) * p.CoutS + cbase);
            }
        }
    }
    {
        float* const xw = reinterpret_cast<float*>(smem + (1 - grp) * 65536);           // the half the OTHER group reads
        const float* const xr = reinterpret_cast<const float*>(smem + grp * 65536);
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int ml = 0; ml < 4; ++ml)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    xw[((wn * 32) + (nt * 4 + ml) * 4 + r) * 64 + lane] = (grp == 0) ? acc[nt][4 + ml][r] : acc[nt][ml][r];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int ml = 0; ml < 4; ++ml)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float v = xr[((wn * 32) + (nt * 4 + ml) * 4 + r) * 64 + lane];
                    if (grp == 0) acc[nt][ml][r] += v; else acc[nt][4 + ml][r] += v;
                }
    }

    // ---- epilogue: this wave owns rows [64 grp, 64 grp + 64) of the tile x 32 couts; a lane: voxel fr of each 16-row tile, 8 consecutive couts
    const bool do_stats = (p.stats != nullptr) && !to_slab && (p.out != nullptr);
    float ssum[8], ssq[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) { ssum[q] = 0.f; ssq[q] = 0.f; }
#pragma unroll
    for (int ml = 0; ml < 4; ++ml) {
        const int r_t = (mt_base + ml) * 16 + fr;
        if (r_t >= TM || l0 + r_t >= DHW) continue;
        const int m = m_base + r_t;
        float v[8];
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) v[nt * 4 + r] = (grp == 0) ? acc[nt][ml][r] : acc[nt][4 + ml][r];
        if (to_slab) {
#pragma unroll
            for (int q = 0; q < 2; ++q)
                *reinterpret_cast<float4*>(slab_ptr(p.partial, split, p.M, p.CoutPad, p.slab_lg, m, cbase + 4 * q)) =
                    make_float4(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
            continue;
        }
        if (p.bias || p.bias2) {
#pragma unroll
            for (int q = 0; q < 2; ++q) { v[4 * q] += ebias[q].x; v[4 * q + 1] += ebias[q].y; v[4 * q + 2] += ebias[q].z; v[4 * q + 3] += ebias[q].w; }
        }
        if (p.temb) {
#pragma unroll
            for (int q = 0; q < 2; ++q) { v[4 * q] += etemb[q].x; v[4 * q + 1] += etemb[q].y; v[4 * q + 2] += etemb[q].z; v[4 * q + 3] += etemb[q].w; }
        }
        if (cbase >= p.CoutS) continue;
        if (p.residual) {
            const u32x4 rv = eres[ml];
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
            ssum[2 * q] += lo; ssq[2 * q] += lo * lo;
            ssum[2 * q + 1] += hi; ssq[2 * q + 1] += hi * hi;
        }
        *reinterpret_cast<u32x4*>(p.out + (size_t)m * p.CoutS + cbase) = o;
    }
    if (do_stats) {
#define PP_ROW_ADD(X, CTRL) X += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(X), CTRL, 0xf, 0xf, true))
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            PP_ROW_ADD(ssum[q], 0x128); PP_ROW_ADD(ssum[q], 0x124); PP_ROW_ADD(ssum[q], 0x122); PP_ROW_ADD(ssum[q], 0x121);
            PP_ROW_ADD(ssq[q], 0x128); PP_ROW_ADD(ssq[q], 0x124); PP_ROW_ADD(ssq[q], 0x122); PP_ROW_ADD(ssq[q], 0x121);
        }
#undef PP_ROW_ADD
        float* red = reinterpret_cast<float*>(smem + HPP_ROFF);    // [2 row halves][BN couts][2]
        if (fr == 0) {
            float* d = red + ((grp * BN) + wn * 32 + 8 * fg) * 2;
#pragma unroll
            for (int q = 0; q < 8; ++q) { d[2 * q] = ssum[q]; d[2 * q + 1] = ssq[q]; }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (tid < BN && n0 + tid < p.CoutS) {
            const float s0 = red[tid * 2] + red[(BN + tid) * 2], s1 = red[tid * 2 + 1] + red[(BN + tid) * 2 + 1];
            *reinterpret_cast<float2*>(p.stats + ((size_t)mtile * p.CoutS + n0 + tid) * 2) = make_float2(s0, s1);
        }
    }
#undef PP_ILV
#undef PP_MFMA16
#undef PP_READ8
#undef PP_WLOAD
#undef PP_ISSUE_TILE
#undef PP_PC
#undef PP_STAMP
#endif  // __HIP_DEVICE_COMPILE__
}
