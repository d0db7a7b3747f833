// 3x3x3 stride-1 "same" convolution, W-halo reuse of the voxel operand, WEIGHTS FED FROM REGISTERS (gfx950, bf16 MFMA 16x16x32).
// EXPERIMENTS BUILDS ONLY (make EXTRA=-DLDM_EXPERIMENTS, LDM_HALO_RW=1): measured 22 % slower than conv3_halo_kernel, see the end of this comment.
//
// Same job, data layout, tile (126 output voxels x 128 couts), numerics and epilogue as conv3_halo_kernel<6> (conv_halo.h; the reference's
// call sites are the nn.Conv3d modules MONAI builds for 3d_ldm/train_diffusion.py:197-205 / 3d_ldm/inference.py:94-99).  What changed,
// and why (round 5, profiles/r05_halo_ablations.txt -- sustained launches of 256 -> 256 at 24^3, in-kernel stamps):
//
//   conv3_halo_kernel's K step takes 844 shader cycles against 512 of MFMA issue, at 2.27 - 2.38 GHz (the clock is NOT the loss).  Its
//   ablations do not overlap, they ADD: MFMAs + barrier 578, + fragment reads 121, + the issue of 21 LDS-DMA pieces 164 (the same with every
//   copy out of range: it is the instructions, not the bytes) = 863.  With the per-step s_barrier removed (wrong results, same instruction
//   stream) the step takes 516 cycles.  So the loop is bound by LOCK-STEP: all eight waves issue their copies, then their reads, then their
//   MFMAs together, and a wave stuck in a 60 - 180 cycle LDS-DMA issue leaves its SIMD's matrix pipe idle because its partner wave is stuck
//   in the same place; the barrier re-aligns them every 0.37 us.
//
// Here the per-step synchronisation and 16 of the 21 LDS-DMA pieces per step are gone:
//   * wave layout 1 x 4 x 2: wave = (K half grp, 32-cout quarter wn) and owns ALL 128 tile rows x 32 couts, so nobody shares its weight
//     fragments: they go straight from global memory / L2 into the A-side registers (buffer_load_dwordx4, 2 per wave and step, six
//     register sets, five steps ahead) -- no LDS, no visibility barrier;
//   * only the voxel tile (16 KiB per (kd, kh, Cin chunk) macro step = 3 K steps) still travels by LDS-DMA, into a 3-slot ring, and the
//     workgroup meets at ONE barrier per macro step (in front of the first fragment read of the next tile); inside a macro step the
//     waves run free and drift apart, which is what lets one wave's MFMAs cover another's loads;
//   * LDS: 48 KiB ring + tap table + a 64 KiB exchange area of its own (the K-half reduction no longer has to wait for the ring).
// Per K step and workgroup: 16 KiB of weights through the vector L1 (2 KiB per wave), 5.3 KiB by LDS-DMA, 64 KiB of fragment reads.
// Covers: Cin % 64 == 0, CoutPad % 128 == 0, bf16 NDHWC output or split-K slabs (row-major or planar), the fused 1x1 skip (second K loop,
// shared among the splits), GroupNorm partials.  NOT covered (conv3_halo_kernel keeps them): the 254 x 64 tall tile, the tile loop,
// fp32-NCDHW / fp32-NDHWC outputs, the 3 x bf16 product of the fp32 precision mode.
//
// MEASURED (round 5, same box, sustained, 256 -> 256 at 24^3; profiles/r05_halo_ablations.txt): 1107 cycles per K step against
// conv3_halo_kernel's 844 (59.1 vs 47.2 us per launch).  Its own ablations: weight loads out of range (no bytes) 756; no barrier 825; both
// 503 (= the MFMA issue).  So (1) 16 KiB of weight fragments per step through the vector L1 cost ~350 cycles: a lane group reads 64 of a
// line's 128 bytes (the K half of its wave), the sibling wave the other half a little later, and the CU asks L2 for ~32 KiB per step --
// beyond the ~70 - 90 GB/s a CU takes in (MI355X_MICROARCH.md, gather rows); (2) ONE barrier per macro step still costs ~250 cycles per
// step: behind it every wave issues its LDS-DMA pieces (60 - 180 cycles each) and its loads at the same moment and no SIMD multiplies.
// What would remove both (alternate K steps per wave group = whole cache lines; register-staged voxel tiles = nothing heavy behind the
// barrier) is bounded by the same ingest ceiling: 21.3 KiB per step at 80 GB/s = 0.27 us per step, 25 % under today's 0.36.
#pragma once
#include "conv_igemm.h"

constexpr int HRW_LDS = 3 * 16384 + 9 * 128 * 4 + 65536 + 2048;      // ring | tap table | K-half exchange | statistics fold

template <int ABL = 0>
__global__ __launch_bounds__(512, 2) void conv3_halo_rw_kernel(const ConvParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int BM = 128, TM = BM - 2, BN = 128, BK = 64, RB = 128;
    constexpr int AT = BM * RB;                                // bytes per voxel tile (16 KiB)
    constexpr int PA = BM / 64;                                // 1 KiB LDS-DMA pieces per wave and voxel tile
    constexpr int NSA = 3, NW = 6, WD = 5;                     // voxel ring slots; weight register sets; steps the weight loads run ahead
    constexpr int TOFF = NSA * AT, XOFF = TOFF + 9 * BM * 4, ROFF = XOFF + 65536;
    static_assert(ROFF + 2048 == HRW_LDS, "LDS layout");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef __attribute__((address_space(3))) void* lds_ptr_t;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2, wn = wave & 3;                  // K half of every 64-deep step; 32-cout quarter of the tile
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
    const int nsteps = 3 * (q_end - q_begin);

    int* const tab = reinterpret_cast<int*>(smem + TOFF);      // (pair, LDS row) -> source voxel, built in the prologue

    // ---- voxel tile loader (LDS-DMA): every wave copies PA pieces (8 rows x 128 B) of a tile; 16-byte chunks XOR-swizzled by (row & 7)
    const int prow = lane >> 3, pchunk = lane & 7;
    int a_row[PA], t_next[PA]; unsigned a_kb[PA], a_vo[PA];
#pragma unroll
    for (int j = 0; j < PA; ++j) {
        const int row = (wave * PA + j) * 8 + prow;
        a_row[j] = row;
        a_kb[j] = (unsigned)((pchunk ^ (row & 7)) * 16);
    }
    __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc((void*)p.x0a, 0, (int)((unsigned)(p.N * DHW) * cin2), 0x00020000);
    const unsigned wtap = (unsigned)p.CoutPad * cin2;          // bytes between two taps of the weight tensor
    __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)p.w0, 0, (int)(27u * wtap), 0x00020000);
    int d_pair = q_begin / nch, d_chunk = q_begin - d_pair * nch;     // DMA stream: the macro step whose tile is issued next
    int d_q = q_begin;
    unsigned d_slot = 0;
#define RW_LOAD_TAB() do {                                                                          \
        _Pragma("unroll") for (int j = 0; j < PA; ++j) {                                            \
            const int v_ = t_next[j];                                                               \
            a_vo[j] = (v_ >= 0 && !(ABL & 32)) ? (unsigned)v_ * cin2 + a_kb[j] : 0xFFFFFFFFu;       \
        }                                                                                           \
        if (d_pair + 1 < 9) { _Pragma("unroll") for (int j = 0; j < PA; ++j) t_next[j] = tab[(d_pair + 1) * BM + a_row[j]]; } \
    } while (0)
    // one voxel tile: macro step d_q into ring slot d_slot; nothing is issued past the K range (the counted waits below rest on the WEIGHT
    // loads that are younger than the tile they guard, so a missing trailing copy only makes them stricter)
#define RW_ISSUE_A() do {                                                                           \
        if (d_q < q_end && !(ABL & 4)) _Pragma("unroll") for (int j = 0; j < PA; ++j)               \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_a, (lds_ptr_t)(smem + d_slot + (wave * PA + j) * 1024), 16,                \
                                                     a_vo[j], (unsigned)d_chunk * (BK * 2), 0, 0);  \
        d_slot = (d_slot == 2 * AT) ? 0u : d_slot + AT; ++d_q;                                      \
        if (++d_chunk == nch) { d_chunk = 0; ++d_pair; if (d_pair < 9) RW_LOAD_TAB(); }             \
    } while (0)

    // ---- weight fragments (A operand) from global memory: row fr of cout tile nt <-> cout n0 + 32 wn + 8 (fr >> 2) + 4 nt + (fr & 3), so that
    //      after the MFMA a lane owns 8 CONSECUTIVE couts of one voxel; k chunk = this wave's K half, 16 bytes per lane
    const int fr = lane & 15, fg = lane >> 4;
    unsigned w_vo[2];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        const int co = n0 + 32 * wn + 8 * (fr >> 2) + 4 * nt + (fr & 3);
        w_vo[nt] = ((ABL & 32) || (dbgflag & 2)) ? 0xFFFFFFFFu : (unsigned)co * cin2 + (unsigned)(grp * 64 + fg * 16);
    }
    bf16x8 wf[NW][2];
    int w_pair = d_pair, w_chunk = d_chunk;                    // weight stream: macro step of the K step loaded next (its kw is static)
#define RW_WLOAD(SET, KW) do {                                                                      \
        const int wp_ = w_pair < 9 ? w_pair : 8;               /* past the end: a valid, unused tile */ \
        const unsigned so_ = (unsigned)(wp_ * 3 + (KW)) * wtap + (unsigned)w_chunk * (BK * 2);      \
        _Pragma("unroll") for (int nt = 0; nt < 2; ++nt)                                            \
            wf[SET][nt] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rs_w, (int)w_vo[nt], (int)so_, 0)); \
        if ((KW) == 2) { if (++w_chunk == nch) { w_chunk = 0; ++w_pair; } }                         \
    } while (0)

    // ---- voxel fragment addressing (B operand): lane = (tile row fr of 16-row tile t, 16-byte k chunk grp * 4 + fg); rows shifted by kw
    const int cfrag = grp * 4 + fg;
    int a_rdk[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) a_rdk[k] = (fr + k) * RB + ((cfrag ^ ((fr + k) & 7)) << 4);
    unsigned c_slot = 0;                                       // ring slot (byte offset) of the macro step whose fragments are read next
    f32x4 acc[2][8];
    bf16x8 xfA[8], xfB[8];
#define RW_READ(XF, KW) do {                                                                        \
        if (!(ABL & 16)) {                                                                          \
            const char* sa_ = smem + c_slot + a_rdk[KW];                                            \
            _Pragma("unroll") for (int t = 0; t < 8; ++t) XF[t] = *reinterpret_cast<const bf16x8*>(sa_ + t * 16 * RB); \
        }                                                                                           \
    } while (0)
#define RW_MASK(XF, KW) do {                                                                        \
        if ((KW) != 1) {                                                                            \
            _Pragma("unroll") for (int t = 0; t < 8; ++t)                                           \
                if ((wmask >> (((KW) == 0 ? 0 : 8) + t)) & 1u) XF[t] = (bf16x8){0, 0, 0, 0, 0, 0, 0, 0}; \
        }                                                                                           \
    } while (0)
#define RW_MFMA(SET, XF) do {                                                                       \
        if (!(ABL & 8)) _Pragma("unroll") for (int mt = 0; mt < 8; ++mt)                            \
            _Pragma("unroll") for (int nt = 0; nt < 2; ++nt)                                        \
                acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[SET][nt], XF[mt], acc[nt][mt], 0, 0, 0); \
    } while (0)
#define RW_STAMP(I) do { if ((dbgflag & 512) && tid == 0) {                                         \
        p.stamps[(size_t)blockIdx.x * 8 + 2 * (I)] = __builtin_amdgcn_s_memrealtime();             \
        p.stamps[(size_t)blockIdx.x * 8 + 2 * (I) + 1] = __builtin_amdgcn_s_memtime(); } } while (0)

    // ---- prologue: first voxel tile (a lane works out the source voxels of its own rows), weights of steps 0 .. WD - 1, then the tap table
    auto decompose = [&](const int j, int& od, int& oh, int& ow) -> bool {   // LDS row j <-> output voxel l0 - 1 + j (kw = 1 tap)
        const int l = l0 - 1 + j;
        if (l < 0 || l >= DHW) return false;
        od = (int)fastdiv((unsigned)l, p.fd_hw_m, p.fd_hw_s); const int r = l - od * HW;
        oh = (int)fastdiv((unsigned)r, p.fd_w_m, p.fd_w_s); ow = r - oh * p.Wout;
        return true;
    };
    auto src_of = [&](const bool ok, const int od, const int oh, const int ow, const int pr) -> int {
        const int id = od + pr / 3 - 1, ih = oh + pr % 3 - 1;
        int v = -1;
        if (ok && (unsigned)id < (unsigned)p.Din && (unsigned)ih < (unsigned)p.Hin) {
            v = smp * DHW + (id * p.Hin + ih) * p.Win + ow;
            if (dbgflag & 1) v &= 1023;
        }
        return v;
    };
    RW_WLOAD(0, 0); RW_WLOAD(1, 1); RW_WLOAD(2, 2); RW_WLOAD(3, 0); RW_WLOAD(4, 1);
#pragma unroll
    for (int j = 0; j < PA; ++j) {
        int od = 0, oh = 0, ow = 0;
        const bool ok = decompose(a_row[j], od, oh, ow);
        const int v_ = src_of(ok, od, oh, ow, d_pair);
        a_vo[j] = (v_ >= 0 && !(ABL & 32)) ? (unsigned)v_ * cin2 + a_kb[j] : 0xFFFFFFFFu;
    }
    {   // tile of macro step q_begin -> slot 0 (the stream state moves on below, once the table exists)
        if (!(ABL & 4))
#pragma unroll
            for (int j = 0; j < PA; ++j)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_a, (lds_ptr_t)(smem + (wave * PA + j) * 1024), 16, a_vo[j], (unsigned)d_chunk * (BK * 2), 0, 0);
    }
    {
        constexpr int NPART = 512 / BM;
        const int row = tid % BM, part = tid / BM;
        int od = 0, oh = 0, ow = 0;
        const bool ok = decompose(row, od, oh, ow);
        for (int pr = part; pr < 9; pr += NPART) tab[pr * BM + row] = src_of(ok, od, oh, ow, pr);
    }
    unsigned wmask = 0;                                        // bit t: w == 0, bit 8 + t: w == W - 1 for this lane's voxel of 16-row tile t
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        const int l = l0 + t * 16 + fr;
        const int ow = l - (int)fastdiv((unsigned)l, p.fd_w_m, p.fd_w_s) * p.Wout;
        wmask |= (ow == 0 ? 1u : 0u) << t;
        wmask |= (ow == p.Wout - 1 ? 1u : 0u) << (8 + t);
    }
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 8; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // table written
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    {   // the DMA stream moves past the first tile (as RW_ISSUE_A does), then the tiles of macro steps 1 and 2 -> slots 1, 2
#pragma unroll
        for (int j = 0; j < PA; ++j) t_next[j] = (d_pair + 1 < 9) ? tab[(d_pair + 1) * BM + a_row[j]] : -1;
        d_slot = AT; ++d_q;
        if (++d_chunk == nch) { d_chunk = 0; ++d_pair; if (d_pair < 9) RW_LOAD_TAB(); }
        RW_ISSUE_A();
        RW_ISSUE_A();
    }
    RW_STAMP(0);
    // first tile landed everywhere: it is older than the copies of tiles 1 and 2 (where the K range has them), which may stay in flight; the
    // weight loads were issued before it.  vmcnt(N) = "all but the N youngest have completed": N must not exceed what was issued after the tile
    if (q_end - q_begin >= 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PA) : "memory");
    else if (q_end - q_begin == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PA) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    RW_READ(xfA, 0);

    // One K step at static position J (relative step % 6): weights in register set J, kw = J % 3.
    //   kw == 2 is the macro step's last K step: in front of the first fragment read of the NEXT tile the workgroup meets, and the slot of
    //   THIS macro step -- whose last fragments every wave has just received -- is refilled three macro steps ahead.  The next tile must
    //   have landed in every wave before the barrier: vmcnt retires in order and vmcnt(N) leaves only the N youngest operations in flight,
    //   so N has to stay below the number of operations issued AFTER that tile.  Steady state: the tile of macro step M + 1 went out at the
    //   top of step (M - 2, 2); since then 6 steps x 2 weight loads (+ PA copies of tile M + 2 where the K range has it) = 12 or 14 -> N = 8.
    //   Second macro step of a tile (tile 2 came from the prologue): 5 x 2 (+ PA) = 10 or 12 -> 8 holds.  First: the copies of tile 2
    //   (or none) + 2 x 2 weight loads = 4 or 6 -> N = 4.
#define RW_STEP(J, XC, XN) do {                                                                     \
        constexpr int kw_ = (J) % 3, kn_ = ((J) + 1) % 3;                                           \
        __builtin_amdgcn_sched_barrier(0);                                                          \
        __builtin_amdgcn_s_waitcnt(0xC07F);                    /* lgkmcnt(0): the fragments of this step have arrived */ \
        if (kw_ == 2) {                                                                             \
            if (s + (J) < 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4) : "memory");               \
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(8) : "memory");                           \
            if (!(ABL & 64)) __builtin_amdgcn_s_barrier();                                          \
            asm volatile("" ::: "memory");                                                          \
            RW_ISSUE_A();                                                                           \
        }                                                                                           \
        RW_WLOAD(((J) + WD) % NW, ((J) + WD) % 3);                                                  \
        if (kn_ == 0) c_slot = (c_slot == 2 * AT) ? 0u : c_slot + AT;                               \
        RW_MASK(XC, kw_);                                                                           \
        RW_READ(XN, kn_);                                                                           \
        RW_MFMA(J, XC);                                                                             \
        /* one scheduling region: the loads ride in the shadow of the first MFMAs, then one fragment read per MFMA */ \
        _Pragma("unroll") for (int i_ = 0; i_ < (kw_ == 2 ? 2 + PA : 2); ++i_) {                    \
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                      \
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);                                      \
        }                                                                                           \
        _Pragma("unroll") for (int i_ = 0; i_ < 8; ++i_) {                                          \
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                      \
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                      \
        }                                                                                           \
        __builtin_amdgcn_sched_group_barrier(0x008, 16 - 8 - (kw_ == 2 ? 2 + PA : 2), 0);           \
    } while (0)

    int s = 0;
    for (; s + 6 <= nsteps; s += 6) {
        RW_STEP(0, xfA, xfB);
        RW_STEP(1, xfB, xfA);
        RW_STEP(2, xfA, xfB);
        RW_STEP(3, xfB, xfA);
        RW_STEP(4, xfA, xfB);
        RW_STEP(5, xfB, xfA);
    }
    if (s < nsteps) {                                          // 3 steps left (nsteps is a multiple of 3)
        RW_STEP(0, xfA, xfB);
        RW_STEP(1, xfB, xfA);
        RW_STEP(2, xfA, xfB);
    }

    // ---- fused 1x1 skip convolution of a ResBlock (MONAI's skip_connection when Cin != Cout): this split's share [s1b, s1e) of the
    //      p.steps1 K steps over the channel-concatenated sources (x1a | x1b) at the CENTRE tap (LDS row r <-> output voxel l0 - 1 + r: the
    //      (kd, kh) = (1, 1) row of the tap table read at kw = 1, no border masks).  One voxel tile per step through the ring (three slots,
    //      two steps ahead, one barrier per step: 8 - 16 steps against 108 - 216), weights from registers again.
    int s1b = 0, s1e = p.steps1;
    if (p.steps1 > 0 && p.splitk > 1) {
        const int per = (p.steps1 + p.splitk - 1) / p.splitk;
        s1b = split * per; s1e = s1b + per; if (s1e > p.steps1) s1e = p.steps1;
    }
    if (s1e > s1b) {
        const int n1 = s1e - s1b, nca = p.c1a / BK;
        const unsigned c1a2 = (unsigned)p.c1a * 2u, c1b2 = (unsigned)p.c1b * 2u, w1row2 = (unsigned)(p.c1a + p.c1b) * 2u;
        __amdgpu_buffer_rsrc_t rs_1a = __builtin_amdgcn_make_buffer_rsrc((void*)p.x1a, 0, (int)((unsigned)(p.N * DHW) * c1a2), 0x00020000);
        __amdgpu_buffer_rsrc_t rs_1b = __builtin_amdgcn_make_buffer_rsrc((void*)(p.x1b ? p.x1b : p.x1a), 0, (int)((unsigned)(p.N * DHW) * (p.x1b ? c1b2 : c1a2)), 0x00020000);
        __amdgpu_buffer_rsrc_t rs_1w = __builtin_amdgcn_make_buffer_rsrc((void*)p.w1, 0, (int)((unsigned)p.CoutPad * w1row2), 0x00020000);
        unsigned va1[PA], vb1[PA], wo1[2];
#pragma unroll
        for (int j = 0; j < PA; ++j) {
            const int v_ = tab[4 * BM + a_row[j]];
            va1[j] = v_ >= 0 ? (unsigned)v_ * c1a2 + a_kb[j] : 0xFFFFFFFFu;
            vb1[j] = v_ >= 0 ? (unsigned)v_ * c1b2 + a_kb[j] : 0xFFFFFFFFu;
        }
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const int co = n0 + 32 * wn + 8 * (fr >> 2) + 4 * nt + (fr & 3);
            wo1[nt] = (unsigned)co * w1row2 + (unsigned)(grp * 64 + fg * 16);
        }
        bf16x8 wk[3][2];
        auto issue1 = [&](const int c, const int slot) {       // voxel tile of skip chunk c -> ring slot `slot`
            if (c < nca) {
#pragma unroll
                for (int j = 0; j < PA; ++j)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_1a, (lds_ptr_t)(smem + slot * AT + (wave * PA + j) * 1024), 16, va1[j], (unsigned)c * (BK * 2), 0, 0);
            } else {
#pragma unroll
                for (int j = 0; j < PA; ++j)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_1b, (lds_ptr_t)(smem + slot * AT + (wave * PA + j) * 1024), 16, vb1[j], (unsigned)(c - nca) * (BK * 2), 0, 0);
            }
        };
#define RW_W1LOAD(SET, C) do {                                                                      \
        _Pragma("unroll") for (int nt = 0; nt < 2; ++nt)                                            \
            wk[SET][nt] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rs_1w, (int)wo1[nt], (int)((unsigned)(C) * (BK * 2)), 0)); \
    } while (0)
        // every wave is past its last fragment read of the 3^3 loop before the ring is refilled (the loop's trailing copies, issued out of
        // range or not, are waited for by the counts below: they are older)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        issue1(s1b, 0); RW_W1LOAD(0, s1b);
        if (n1 > 1) { issue1(s1b + 1, 1); RW_W1LOAD(1, s1b + 1); }
        // step S1 (ring slot and register set SET = S1 % 3): its tile + weights landed (the PA + 2 operations of step S1 + 1 may fly),
        // barrier, refill the slot step S1 - 1 released with step S1 + 2, read, multiply
#define RW_SKIP_STEP(SET, S1) do {                                                                  \
        if ((S1) + 1 < n1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PA + 2) : "memory");            \
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                       \
        __builtin_amdgcn_s_barrier();                                                               \
        asm volatile("" ::: "memory");                                                              \
        if ((S1) + 2 < n1) { issue1(s1b + (S1) + 2, ((SET) + 2) % 3); RW_W1LOAD(((SET) + 2) % 3, s1b + (S1) + 2); } \
        c_slot = (unsigned)(SET) * AT;                                                              \
        RW_READ(xfA, 1);                                                                            \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                          \
        if (!(ABL & 8)) _Pragma("unroll") for (int mt = 0; mt < 8; ++mt)                            \
            _Pragma("unroll") for (int nt = 0; nt < 2; ++nt)                                        \
                acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wk[SET][nt], xfA[mt], acc[nt][mt], 0, 0, 0); \
    } while (0)
        for (int s1 = 0; s1 < n1; s1 += 3) {
            RW_SKIP_STEP(0, s1);
            if (s1 + 1 < n1) RW_SKIP_STEP(1, s1 + 1);
            if (s1 + 2 < n1) RW_SKIP_STEP(2, s1 + 2);
        }
#undef RW_SKIP_STEP
#undef RW_W1LOAD
    }
    RW_STAMP(1);

    // ---- K-half reduction: group g keeps 16-row tiles 4 g .. 4 g + 3 and receives its partner's partial sums for them (an LDS area of its own)
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
        float* const xw = reinterpret_cast<float*>(smem + XOFF + (1 - grp) * 32768);    // the half the OTHER group reads
        const float* const xr = reinterpret_cast<const float*>(smem + XOFF + grp * 32768);
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
#define RW_ROW_ADD(X, CTRL) X += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(X), CTRL, 0xf, 0xf, true))
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            RW_ROW_ADD(ssum[q], 0x128); RW_ROW_ADD(ssum[q], 0x124); RW_ROW_ADD(ssum[q], 0x122); RW_ROW_ADD(ssum[q], 0x121);
            RW_ROW_ADD(ssq[q], 0x128); RW_ROW_ADD(ssq[q], 0x124); RW_ROW_ADD(ssq[q], 0x122); RW_ROW_ADD(ssq[q], 0x121);
        }
#undef RW_ROW_ADD
        float* red = reinterpret_cast<float*>(smem + ROFF);    // [2 row halves][BN couts][2]
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
#undef RW_STEP
#undef RW_MFMA
#undef RW_MASK
#undef RW_READ
#undef RW_WLOAD
#undef RW_ISSUE_A
#undef RW_LOAD_TAB
#undef RW_STAMP
#endif  // __HIP_DEVICE_COMPILE__
}
