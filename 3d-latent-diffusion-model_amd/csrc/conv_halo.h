// 3x3x3 stride-1 "same" convolution with W-HALO REUSE of the voxel operand (gfx950, bf16 MFMA 16x16x32).
//
// Same job, data layout, MFMA roles, epilogue and numerics as conv_igemm_kernel<2,2,64> (conv_igemm.h; the reference's
// call sites are the nn.Conv3d modules MONAI builds for 3d_ldm/train_diffusion.py:197-205 / 3d_ldm/inference.py:94-99),
// for the case that carries 95 % of the UNet's FLOPs: k = 3, stride 1, pad 1, one source tensor, Cin % 64 == 0.
//
// Why a second kernel: conv_igemm copies one voxel tile AND one weight tile per K step (32 KiB / 2.1 MFLOP) and is bound
// by the per-CU global->LDS ingest rate (measured ~71 GB/s/CU), not by the matrix pipe.  The three
// kw taps of one (kd, kh) pair read the SAME voxel rows shifted by one voxel along W, because consecutive tile rows are
// consecutive voxels in NDHWC memory.  So the voxel tile is copied ONCE per (kd, kh, Cin chunk) and the three kw steps
// read it from LDS at row offsets 0 / +1 / +2; a lane zeroes its fragment where the shifted voxel would cross the W
// border (w == 0 for kw = 0, w == W-1 for kw = 2: two precomputed bit masks, 16 v_cndmask per masked step).  Bytes
// copied per K step drop from 32 KiB to 21.3 KiB (voxels 16 KiB / 3 + weights 16 KiB).
//
// Tile geometry: the LDS voxel tile has 128 rows = output rows -1 .. 126 of the tile, so a workgroup produces 126
// output voxels (rows 126/127 of the MFMA tile are computed on junk and never stored: 1.6 % waste) and no separate
// halo copy is needed.  Tiles never straddle samples (tile index -> (sample, tile in sample)).
// Rings: weights 6 x 16 KiB, voxels 3 x 16 KiB, (pair, row) -> voxel table 4.5 KiB = 148.5 KiB of the 160 KiB LDS.
// The step issued in K step S is S + 6, whose kw equals S's kw: the instruction mix of every step is static.
#pragma once
#include "conv_igemm.h"

// ABL: compile-time ablations for timing experiments (LDM_CONV_DBG through the operator-level API only; results are wrong):
//      4 = no global->LDS copies, 8 = no MFMAs, 16 = no LDS fragment reads, 32 = copies issued but all out of range, 64 = no per-step barrier.
// TALL: the same kernel on a 254-voxel x 64-cout tile (waves 4 x 1 instead of 2 x 2, every wave still 64 x 64): the layers with
// Cout = 64 (the AutoencoderKL's 96^3 level) get the halo reuse too, and the copied bytes per K step drop to 8 KiB of weights +
// 32 / 3 KiB of voxels = 18.7 KiB.
// PERSIST: the tile loop below exists (launches with more tiles than CUs); false = one tile per workgroup, straight-line code (every conv
// of the B = 1 UNet step: the loop form keeps per-lane constants live across the K loop and costs registers that launch does not have)
template <int NSB, int ABL = 0, bool TALL = false, bool PERSIST = false, bool MASK_INLINE = false>
__global__ __launch_bounds__(512, 2) void conv3_halo_kernel(const ConvParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int BM = TALL ? 256 : 128, TM = BM - 2, BN = TALL ? 64 : 128, BK = 64, RB = 128;
    constexpr int BT = BN * RB, AT = BM * RB;                  // bytes per weight / voxel tile
    constexpr int PA = BM / 64, PB = BN / 64;                  // 1 KiB copy pieces per wave: voxel tile, weight tile
    constexpr int WGM = TALL ? 4 : 2;                          // wave rows (x 2 K groups = row blocks of the GroupNorm fold)
    constexpr int NSA = 3;
    constexpr bool SPLIT_A = !TALL;                            // voxel tile issued in two halves (kw = 0, kw = 1 steps)
    constexpr int AOFF = NSB * BT;                             // voxel ring behind the weight ring
    constexpr int TOFF = AOFF + NSA * AT;                      // (pair, row) -> voxel table
    static_assert(NSB == 6, "the static step schedule assumes a 6-deep weight ring (a multiple of the 3 kw steps)");
    static_assert(TOFF + 9 * BM * 4 <= 160 * 1024, "LDS budget");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef __attribute__((address_space(3))) void* lds_ptr_t;

    KSTAMP_BEGIN(6);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2, wq = wave & 3;
    const int wm = TALL ? wq : (wq & 1), wn = TALL ? 0 : (wq >> 1);

    // Persistent form: the grid is min(tiles, CUs) workgroups (one fits a CU: 148.5 KiB of LDS) and a workgroup walks the tiles
    // vbid = blockIdx.x, + gridDim.x, ... (gridDim.x is a multiple of 8 whenever it is smaller than the tile count, so a workgroup's
    // tiles keep its XCD residue).  Saves the ~1.2 us between the exit of one workgroup and the first instruction of the next on the
    // same CU (dispatch, kernel-argument loads) for every tile but the first: the 96^3 convolutions of the AutoencoderKL run 13.6
    // tiles per CU.  Launches with <= CUs tiles (every conv of the B = 1 UNet step) are unchanged: one trip through the loop.
    // Inside that loop a tile's K-group exchange and epilogue overlap the NEXT tile's set-up: right after the K loop (behind the barrier
    // that ends its fragment reads) the workgroup works out the next tile, requests its first macro step (weight slots 0 - 2, voxel
    // slot 0) and builds its tap table; the exchange and the statistics fold of the current tile live in the ring slots that step does
    // not touch (XCH0 / XCH1 / RED below), and the epilogue's stores run while those copies are in flight.
    const int nwg = p.mtiles * p.ntiles * p.splitk;
    const int DHW = p.Dout * p.Hout * p.Wout, HW = p.Hout * p.Wout;
    const int nch = p.nchunk0;
    const int Q = 9 * nch;                                     // macro steps: (kd, kh) x Cin chunk
    // tile state (HL_TILE_SETUP): output tile, K range, issue stream
    int mtile = 0, ntile = 0, split = 0, n0 = 0, smp = 0, l0 = 0, m_base = 0, q_begin = 0, q_end = 0, nsteps = 0;
    const unsigned cin2 = (unsigned)p.c0a * 2u;                  // bytes per voxel row
    const int x3n = p.x3_n;
    const unsigned wrow2 = x3n ? (unsigned)x3n * 3u * (BK * 2) : cin2;   // bytes per weight row
    const int dbgflag = p.dbg;

    int* const tab = reinterpret_cast<int*>(smem + TOFF);          // (pair, LDS row) -> source voxel, built in the prologue

    // ---- loader lanes: every wave copies PA pieces (8 rows x 128 B) of a voxel tile and PB of a weight tile
    const int prow = lane >> 3, pchunk = lane & 7;
    int a_row[PA], t_next[PA]; unsigned a_kb[PA], a_vo[PA], b_vo[PB];
#pragma unroll
    for (int j = 0; j < PA; ++j) {
        const int row = (wave * PA + j) * 8 + prow;
        a_row[j] = row;
        // 16-byte chunk swizzle keyed by (row & 7): the low swizzle bit equals the row parity, so the 16 rows a
        // ds_read_b128 lane group touches hit 16 different bank quads for ANY row shift (the kw-shifted reads below);
        // the (row >> 1) key of conv_igemm.h is conflict free only for unshifted tiles (2-way conflicts at kw = 1, 2).
        a_kb[j] = (unsigned)((pchunk ^ (row & 7)) * 16);
    }
    const unsigned wtap = (unsigned)p.CoutPad * wrow2;          // bytes between two taps of the weight tensor
    __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc((void*)p.x0a, 0, (int)((unsigned)(p.N * DHW) * cin2), 0x00020000);
    __amdgpu_buffer_rsrc_t rs_b = __builtin_amdgcn_make_buffer_rsrc((void*)p.w0, 0, (int)(27u * wtap), 0x00020000);

    // issue-stream state (scalar): next step to copy = macro (i_pair, i_chunk), kw i_kw; i_s = its relative index
    int i_pair = 0, i_chunk = 0, i_s = 0;
    unsigned i_aslot = 0;                                      // byte offset of the voxel ring slot of the macro being issued
#define HL_TILE_SETUP(VB) do {                                                                      \
        int lid_ = xcd_remap((VB), nwg);                                                            \
        if (p.tile_order == 1) {       /* cout tiles fastest: every XCD gets a contiguous range of M tiles with ALL their cout tiles */ \
            ntile = lid_ % p.ntiles; lid_ /= p.ntiles; mtile = lid_ % p.mtiles; split = lid_ / p.mtiles;                   \
        } else {                       /* M tiles fastest: every XCD streams one weight panel (one cout tile) */          \
            mtile = lid_ % p.mtiles; lid_ /= p.mtiles; ntile = lid_ % p.ntiles; split = lid_ / p.ntiles;                   \
        }                                                                                           \
        n0 = ntile * BN;                                                                            \
        smp = mtile / p.halo_mtps;                                                                  \
        l0 = (mtile - smp * p.halo_mtps) * TM;       /* first output voxel of the tile inside its sample */               \
        m_base = smp * DHW + l0;                                                                    \
        q_begin = split * p.q_per_split;                                                            \
        q_end = q_begin + p.q_per_split; if (q_end > Q) q_end = Q;                                  \
        nsteps = 3 * (q_end - q_begin);              /* K steps of this tile, relative index 0 .. nsteps-1 */              \
        _Pragma("unroll") for (int j = 0; j < PB; ++j) {                                            \
            const int R = (wave * PB + j) * 8 + prow;                      /* row of the weight tile */                   \
            const unsigned b_kb = (unsigned)((pchunk ^ ((R >> 1) & 7)) * 16);                       \
            const int q = R >> 6, nt = (R >> 4) & 3, i = R & 15;                                    \
            const int co = n0 + 64 * q + 16 * (i >> 2) + 4 * nt + (i & 3);  /* see conv_igemm.h: lane ends up with 16 consecutive couts */ \
            b_vo[j] = ((ABL & 32) || (dbgflag & 2)) ? 0xFFFFFFFFu : (unsigned)co * wrow2 + b_kb;   /* ABL 32: every copy out of range */ \
        }                                                                                           \
        i_pair = q_begin / nch; i_chunk = q_begin - i_pair * nch; i_s = 0; i_aslot = 0;             \
    } while (0)
    // the table entries of a (kd, kh) pair are read ONE PAIR AHEAD (t_next): with Cin = 64 (one chunk per pair: the AutoencoderKL's
    // 96^3 level) a new pair starts every third K step and its LDS read + address arithmetic sat in front of that step's waits
#define HL_LOAD_TAB() do {                                                                          \
        _Pragma("unroll") for (int j = 0; j < PA; ++j) {                                            \
            const int v_ = t_next[j];                                                               \
            a_vo[j] = (v_ >= 0 && !(ABL & 32)) ? (unsigned)v_ * cin2 + a_kb[j] : 0xFFFFFFFFu;       \
        }                                                                                           \
        if (i_pair + 1 < 9) { _Pragma("unroll") for (int j = 0; j < PA; ++j) t_next[j] = tab[(i_pair + 1) * BM + a_row[j]]; } \
    } while (0)
    // copies of one step; KW is the step's kw (static).  Slot of the weight tile = relative step % NSB (static: BSLOT).
#define HL_ISSUE_W(KW, BSLOT) do {                                                                  \
        const unsigned sb_ = (unsigned)(i_pair * 3 + (KW)) * wtap + (unsigned)i_chunk * (BK * 2);                          \
        if (!(ABL & 4)) _Pragma("unroll") for (int j = 0; j < PB; ++j)                              \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_b, (lds_ptr_t)(smem + (BSLOT) * BT + (wave * PB + j) * 1024), 16, b_vo[j], sb_, 0, 0); \
    } while (0)
#define HL_ISSUE_A() do {                                                                           \
        if (!(ABL & 4)) {                                                                           \
            const int ac_ = (x3n && i_chunk >= x3n) ? i_chunk - x3n : i_chunk;     /* x3: hi, hi again, lo */ \
            _Pragma("unroll") for (int j = 0; j < PA; ++j)                                          \
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_a, (lds_ptr_t)(smem + AOFF + i_aslot + (wave * PA + j) * 1024), 16, a_vo[j], \
                                                         (unsigned)ac_ * (BK * 2), 0, 0);          \
        }                                                                                           \
    } while (0)
    // 126 x 128 tile (SPLIT_A): the voxel tile of a macro step travels in two halves, with the kw = 0 and the kw = 1 step of the macro
    // step issued six steps earlier (3 / 3 / 2 copies per step instead of 4 / 2 / 2): +0.5 % on the UNet step, same box.  The tall tile
    // keeps all PA = 4 pieces in the kw = 0 step: split 3 / 3 / 1 it measured 0.7 % SLOWER on the 96^3 convolutions (profiles/r04_ab_split_issue.txt)
#define HL_ISSUE_A_HALF(H) do {                                                                     \
        if (!(ABL & 4)) {                                                                           \
            const int ac_ = (x3n && i_chunk >= x3n) ? i_chunk - x3n : i_chunk;     /* x3: hi, hi again, lo */ \
            _Pragma("unroll") for (int j = (H) * (PA / 2); j < ((H) + 1) * (PA / 2); ++j)           \
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_a, (lds_ptr_t)(smem + AOFF + i_aslot + (wave * PA + j) * 1024), 16, a_vo[j], \
                                                         (unsigned)ac_ * (BK * 2), 0, 0);          \
        }                                                                                           \
    } while (0)
#define HL_ISSUE(KW, BSLOT) do {                                                                    \
        HL_ISSUE_W(KW, BSLOT);                                                                      \
        if (SPLIT_A) { if ((KW) == 0) HL_ISSUE_A_HALF(0); if ((KW) == 1) HL_ISSUE_A_HALF(1); }      \
        else if ((KW) == 0) HL_ISSUE_A();                                                           \
        ++i_s;                                                                                      \
    } while (0)
    // issue stream moves on to the next macro step: next Cin chunk, or next (kd, kh) pair (table read + multiply-add per
    // copied row).  Runs in front of the waits of a kw == 0 step, outside the hot block.
#define HL_ADVANCE() do {                                                                           \
        i_aslot = (i_aslot == 2 * AT) ? 0u : i_aslot + AT;                                          \
        if (++i_chunk == nch) { i_chunk = 0; ++i_pair; if (i_pair < 9) HL_LOAD_TAB(); }             \
    } while (0)

    // ---- fragment addressing
    const int fr = lane & 15, fg = lane >> 4;
    const int ra0 = wm * 64 + fr, rb0 = wn * 64 + fr;
    const int cfrag = grp * 4 + fg;
    const int b_rd0 = rb0 * RB + ((cfrag ^ ((rb0 >> 1) & 7)) << 4);
    int a_rdk[3];                                              // voxel rows shifted by kw: LDS row = tile row + kw
#pragma unroll
    for (int k = 0; k < 3; ++k) a_rdk[k] = AOFF + (ra0 + k) * RB + ((cfrag ^ ((ra0 + k) & 7)) << 4);
    unsigned wmask = 0;                                        // W-border masks of this lane's 4 voxel rows (per tile, below)
    f32x4 acc[4][4];
    bf16x8 wfA[4], afA[4], wfB[4], afB[4];
    unsigned c_aslot = 0;                                      // voxel ring slot (byte offset) of the macro step being read NEXT
    // LDS the K-group exchange (two 32 KiB halves, one per receiving group) and the statistics fold use: ring slots the next tile's first
    // macro step (weight slots 0 - 2, voxel slot 0) does not touch
    constexpr int XCH0 = TALL ? AOFF + AT : 3 * BT, XCH1 = TALL ? AOFF + AT + 32768 : AOFF + AT, RED = TALL ? 3 * BT : 3 * BT + 32768;
    static_assert(XCH0 + 32768 <= (TALL ? XCH1 : AOFF) && XCH1 + 32768 <= TOFF && RED + 4096 <= (TALL ? AOFF : AOFF), "exchange areas must avoid the first macro step's slots");

#define HL_READ(WF, AF, BSLOT, KW) do {                                                             \
        if (!(ABL & 16)) {                                                                          \
            const char* sb_ = smem + (BSLOT) * BT + b_rd0;                                          \
            const char* sa_ = smem + c_aslot + a_rdk[KW];                                           \
            _Pragma("unroll") for (int t = 0; t < 4; ++t) {                                         \
                WF[t] = *reinterpret_cast<const bf16x8*>(sb_ + t * 16 * RB);                        \
                AF[t] = *reinterpret_cast<const bf16x8*>(sa_ + t * 16 * RB);                        \
            }                                                                                       \
        }                                                                                           \
    } while (0)
#define HL_MASK(AF, KW) do {                                                                        \
        if ((KW) != 1) {                                                                            \
            _Pragma("unroll") for (int t = 0; t < 4; ++t)                                           \
                if ((wmask >> (((KW) == 0 ? 0 : 4) + t)) & 1u) AF[t] = (bf16x8){0, 0, 0, 0, 0, 0, 0, 0}; \
        }                                                                                           \
    } while (0)
#define HL_MFMA(WF, AF) do {                                                                        \
        if (!(ABL & 8)) _Pragma("unroll") for (int nt = 0; nt < 4; ++nt)                            \
            _Pragma("unroll") for (int mt = 0; mt < 4; ++mt)                                        \
                acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(WF[nt], AF[mt], acc[nt][mt], 0, 0, 0); \
    } while (0)
    // One K step at static position J (relative step % 6, kw = J % 3).  VM = copies that may stay in flight while the
    // data of step S+1 must have landed: steps S+2 .. S+5 -> 2 weight copies each + 2 voxel copies per kw == 0 step.
    // ISSUE: 1 = steady state (step S+6 exists), 0 = tail (nothing left to issue: drain).
#define HL_STEP(J, WC, AC, WN, AN) do {                                                             \
        constexpr int kw_ = (J) % 3, kn_ = ((J) + 1) % 3;                                           \
        /* copies that may stay in flight while the operands of step S+1 must have landed (in-order retirement).  kn_ == 0: S+1 opens a  \
           macro step: both halves of its voxel tile (issue slots S-5, S-4) are needed, slots S-3 .. S-1 (kw 2, 0, 1) may fly; kn_ == 1 / 2:  \
           only the weight tile of S+1 (slot S-5) is new, slots S-4 .. S-1 may fly */                                                        \
        constexpr int vm_ = SPLIT_A ? ((kn_ == 0) ? 3 * PB + PA : (kn_ == 1) ? 4 * PB + PA : 4 * PB + 3 * (PA / 2))                          \
            /* whole tile with the kw = 0 step: steps S+2 .. S+5 -> their weight tiles + one voxel tile per kw == 0 step among them */       \
            : 4 * PB + PA * ((((J) + 2) % 3 == 0) + (((J) + 3) % 3 == 0) + (((J) + 4) % 3 == 0) + (((J) + 5) % 3 == 0));                     \
        constexpr int nc_ = SPLIT_A ? ((kw_ < 2) ? PB + PA / 2 : PB) : ((kw_ == 0) ? PB + PA : PB);   /* copies issued by this step */ \
        if (kw_ == 0) HL_ADVANCE();                                                                 \
        if (kn_ == 0) c_aslot = (c_aslot == 2 * AT) ? 0u : c_aslot + AT;                            \
        /* hard boundary: s_barrier alone does not stop register-only MFMAs from drifting into the neighbouring step  \
           (instruction selection orders them freely inside a basic block), so every step gets its own block: an       \
           opaque never-taken branch, as the diagnostic stamps do in conv_igemm_kernel */                               \
        if (dbgflag & 2048) asm volatile("s_nop 0");                                                \
        __builtin_amdgcn_sched_barrier(0);                                                          \
        __builtin_amdgcn_s_waitcnt(0xC07F);                    /* fragments of step S have arrived */ \
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(vm_) : "memory");                                  \
        if ((ABL & 384) == 384) {                              /* timing: a barrier with SLACK -- go on once every wave has reached step S - slack */ \
            volatile int* const cnt_ = reinterpret_cast<volatile int*>(smem + TOFF + 9 * BM * 4);   \
            if (lane == 0) cnt_[wave] = s + (J) + 1;                                                \
            const int need_ = s + (J) + 1 - ((ABL & 1) ? 2 : 1);                                    \
            for (int it_ = 0; it_ < 100000; ++it_) {                                                \
                const int v_ = cnt_[lane & 7];                                                      \
                if (__builtin_amdgcn_ballot_w64(v_ < need_) == 0) break;                            \
                __builtin_amdgcn_s_sleep(1);                                                        \
            }                                                                                       \
        } else                                                                                      \
        if (!(ABL & 64) && !((ABL & 128) && ((J) & 1)) && !((ABL & 256) && (J) % 3 != 0)) __builtin_amdgcn_s_barrier();   /* ABL 128 / 256: every 2nd / 3rd step only (timing) */ \
        asm volatile("" ::: "memory");                                                              \
        HL_ISSUE(kw_, (J));                                    /* step S+6 refills the weight slot of step S */ \
        HL_READ(WN, AN, ((J) + 1) % NSB, kn_);                                                      \
        if (MASK_INLINE) {                                                                          \
            /* round 5: the W-border masks of a voxel fragment sit right in front of the four MFMAs that read it (voxel tile outer, cout  \
               tile inner) instead of all 16 v_cndmask in front of the step's first MFMA, where both waves of a SIMD executed them at the  \
               same moment behind the barrier with the matrix pipe idle */                                                              \
            _Pragma("unroll") for (int mt = 0; mt < 4; ++mt) {                                      \
                if (kw_ != 1 && ((wmask >> ((kw_ == 0 ? 0 : 4) + mt)) & 1u)) AC[mt] = (bf16x8){0, 0, 0, 0, 0, 0, 0, 0}; \
                if (!(ABL & 8)) _Pragma("unroll") for (int nt = 0; nt < 4; ++nt)                    \
                    acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(WC[nt], AC[mt], acc[nt][mt], 0, 0, 0); \
            }                                                                                       \
            _Pragma("unroll") for (int i_ = 0; i_ < 16; ++i_) {                                     \
                if (kw_ != 1 && (i_ & 3) == 0) __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);   \
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                  \
                if (i_ < nc_) { __builtin_amdgcn_sched_group_barrier(0x004, 2, 0); __builtin_amdgcn_sched_group_barrier(0x020, 1, 0); } \
                else if (i_ < nc_ + 8) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);           \
            }                                                                                       \
        } else {                                                                                    \
        HL_MASK(AC, kw_);                                                                           \
        HL_MFMA(WC, AC);                                                                            \
        /* one scheduling region: every copy / LDS read rides in the shadow of an MFMA (border masks float freely) */ \
        _Pragma("unroll") for (int i_ = 0; i_ < nc_; ++i_) {                                        \
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                      \
            __builtin_amdgcn_sched_group_barrier(0x004, 2, 0);                                      \
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);                                      \
        }                                                                                           \
        _Pragma("unroll") for (int i_ = 0; i_ < 8; ++i_) {                                          \
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                      \
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                      \
        }                                                                                           \
        __builtin_amdgcn_sched_group_barrier(0x008, 16 - nc_ - 8, 0);                               \
        }                                                                                           \
    } while (0)

    // diagnostic (dbg & 512, operator-level API): shader-clock and 100 MHz stamps around the K loop -> effective clock
#define HL_STAMP(I) do { if ((dbgflag & 512) && tid == 0) {                                         \
        p.stamps[(size_t)vbid * 8 + 2 * (I)] = __builtin_amdgcn_s_memrealtime();                   \
        p.stamps[(size_t)vbid * 8 + 2 * (I) + 1] = __builtin_amdgcn_s_memtime(); } } while (0)
    // ---- prologue.  The first macro step's copies are requested BEFORE the tap table exists (a lane works out the source voxels of
    //      its own PA rows for that one (kd, kh) pair by itself), so that their round trip to memory (cold: the weights come from HBM,
    //      the voxels from another XCD's write-back) overlaps the table's integer divisions and its barrier.
    //      Integer divisions are the expensive part (~40 instructions each): a lane decomposes an output voxel ONCE and derives the
    //      source voxel of any (kd, kh) pair from it with adds and compares.
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
            if (dbgflag & 1) v &= 1023;                            // timing experiment: every voxel copy hits the same 1024 rows (L2 resident)
        }
        return v;
    };
    // first macro step of the tile HL_TILE_SETUP has just selected + its tap table (no barrier: the caller's next barrier completes it)
#define HL_TILE_EARLY() do {                                                                        \
        HL_ISSUE_W(0, 0); HL_ISSUE_W(1, 1); HL_ISSUE_W(2, 2);  /* the weight tiles need no voxel arithmetic: on their way first */ \
        _Pragma("unroll") for (int j = 0; j < PA; ++j) {                                            \
            int od = 0, oh = 0, ow = 0;                                                             \
            const bool ok = decompose(a_row[j], od, oh, ow);                                        \
            const int v_ = src_of(ok, od, oh, ow, i_pair);                                          \
            a_vo[j] = (v_ >= 0 && !(ABL & 32)) ? (unsigned)v_ * cin2 + a_kb[j] : 0xFFFFFFFFu;       \
        }                                                                                           \
        HL_ISSUE_A(); i_s += 3;        /* issue order of the first macro step: w0 w1 w2 a0; from the next on w a w w */   \
        {                                                                                           \
            constexpr int NPART = 512 / BM;          /* lanes per LDS row: each takes every NPART-th (kd, kh) pair */      \
            const int row = tid % BM, part = tid / BM;                                              \
            int od = 0, oh = 0, ow = 0;                                                             \
            const bool ok = decompose(row, od, oh, ow);                                             \
            for (int pr = part; pr < 9; pr += NPART) tab[pr * BM + row] = src_of(ok, od, oh, ow, pr); \
        }                                                                                           \
    } while (0)
    HL_TILE_SETUP((int)blockIdx.x);
    HL_TILE_EARLY();
    int vbid = blockIdx.x;
    do {
    wmask = 0;                                                 // bit t = w == 0, bit 4 + t = w == W - 1 for the lane's voxel row of 16-row tile t
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int l = l0 + wm * 64 + t * 16 + fr;
        const int ow = l - (int)fastdiv((unsigned)l, p.fd_w_m, p.fd_w_s) * p.Wout;
        wmask |= (ow == 0 ? 1u : 0u) << t;
        wmask |= (ow == p.Wout - 1 ? 1u : 0u) << (4 + t);
    }
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
    c_aslot = 0;
    // table complete (and, from the second tile on, the previous tile's exchange / statistics reads of the LDS that the next copies
    // overwrite are over).  A raw barrier behind an LDS-only wait: __syncthreads() would drain vmcnt(0) and with it the copies in flight
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
#pragma unroll
    for (int j = 0; j < PA; ++j) t_next[j] = (i_pair + 1 < 9) ? tab[(i_pair + 1) * BM + a_row[j]] : -1;   // the pair after the first one
    KSTAMP(1);
    HL_STAMP(0);
    if (nsteps >= 6) {
        HL_ADVANCE(); HL_ISSUE(0, 3); HL_ISSUE(1, 4); HL_ISSUE(2, 5);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * PB + PA) : "memory");      // steps 0..2 landed (w0 w1 w2 a0 went first); steps 3..5 = 3 weight tiles + one voxel tile in flight
    } else {                                                   // a single macro step in this K range
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    KSTAMP(2);
    HL_READ(wfA, afA, 0, 0);

    int s = 0;
    for (; s + 12 <= nsteps; s += 6) {                         // steady state: every step of the block issues step s+6+j
        HL_STEP(0, wfA, afA, wfB, afB);
        HL_STEP(1, wfB, afB, wfA, afA);
        HL_STEP(2, wfA, afA, wfB, afB);
        HL_STEP(3, wfB, afB, wfA, afA);
        HL_STEP(4, wfA, afA, wfB, afB);
        HL_STEP(5, wfB, afB, wfA, afA);
    }
    // tail: 3, 6 or 9 steps left (nsteps is a multiple of 3 and s of 6); copies for relative steps >= i_s still to issue
    for (; s < nsteps; s += 6) {
#define HL_TAIL_STEP(J, WC, AC, WN, AN) do {                                                        \
        constexpr int kw_ = (J) % 3, kn_ = ((J) + 1) % 3;                                           \
        /* copies that may stay in flight while the operands of step S+1 must have landed (in-order retirement).  kn_ == 0: S+1 opens a  \
           macro step: both halves of its voxel tile (issue slots S-5, S-4) are needed, slots S-3 .. S-1 (kw 2, 0, 1) may fly; kn_ == 1 / 2:  \
           only the weight tile of S+1 (slot S-5) is new, slots S-4 .. S-1 may fly */                                                        \
        constexpr int vm_ = SPLIT_A ? ((kn_ == 0) ? 3 * PB + PA : (kn_ == 1) ? 4 * PB + PA : 4 * PB + 3 * (PA / 2))                          \
            /* whole tile with the kw = 0 step: steps S+2 .. S+5 -> their weight tiles + one voxel tile per kw == 0 step among them */       \
            : 4 * PB + PA * ((((J) + 2) % 3 == 0) + (((J) + 3) % 3 == 0) + (((J) + 4) % 3 == 0) + (((J) + 5) % 3 == 0));                     \
        if (kw_ == 0 && i_s < nsteps) HL_ADVANCE();                                                 \
        __builtin_amdgcn_s_waitcnt(0xC07F);                                                         \
        HL_MASK(AC, kw_);                                                                           \
        if (s + (J) + 6 <= nsteps) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(vm_) : "memory");       \
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                       \
        if (!(ABL & 64)) __builtin_amdgcn_s_barrier();                                              \
        asm volatile("" ::: "memory");                                                              \
        if (i_s < nsteps) HL_ISSUE(kw_, (J));                                                       \
        if (s + (J) + 1 < nsteps) {                                                                 \
            if (kn_ == 0) c_aslot = (c_aslot == 2 * AT) ? 0u : c_aslot + AT;                        \
            HL_READ(WN, AN, ((J) + 1) % NSB, kn_);                                                  \
        }                                                                                           \
        HL_MFMA(WC, AC);                                                                            \
    } while (0)
        HL_TAIL_STEP(0, wfA, afA, wfB, afB);
        HL_TAIL_STEP(1, wfB, afB, wfA, afA);
        HL_TAIL_STEP(2, wfA, afA, wfB, afB);
        if (s + 3 >= nsteps) break;
        HL_TAIL_STEP(3, wfB, afB, wfA, afA);
        HL_TAIL_STEP(4, wfA, afA, wfB, afB);
        HL_TAIL_STEP(5, wfB, afB, wfA, afA);
#undef HL_TAIL_STEP
    }
    // ---- fused 1x1 skip convolution of a ResBlock (MONAI's skip_connection when Cin != Cout): p.steps1 extra K steps over the
    //      channel-concatenated sources (x1a | x1b) at the CENTRE tap, i.e. LDS row = tile row + 1 of the (kd, kh) = (1, 1) table,
    //      no border masks.  A short second loop behind the 3^3 one (8 steps for 512 channels against 108): one voxel tile + one
    //      weight tile per step, three ring slots, copies two steps ahead.
    // Split-K convs (round 5): the skip's K steps are dealt to the splits in equal shares (split s takes steps [s * per, (s + 1) * per)), so the
    // 12^3 / 6^3 ResBlocks with a channel change leave conv_igemm_kernel too; the finalize adds bias2 once.
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
        unsigned va1[PA], vb1[PA], wo1[PB];
#pragma unroll
        for (int j = 0; j < PA; ++j) {
            const int v_ = tab[4 * BM + a_row[j]];           // centre (kd, kh) pair: LDS row r <-> output voxel l0 - 1 + r itself
            va1[j] = v_ >= 0 ? (unsigned)v_ * c1a2 + a_kb[j] : 0xFFFFFFFFu;
            vb1[j] = v_ >= 0 ? (unsigned)v_ * c1b2 + a_kb[j] : 0xFFFFFFFFu;
        }
#pragma unroll
        for (int j = 0; j < PB; ++j) {
            const int R = (wave * PB + j) * 8 + prow;
            const unsigned b_kb = (unsigned)((pchunk ^ ((R >> 1) & 7)) * 16);
            const int q = R >> 6, nt = (R >> 4) & 3, i = R & 15;
            const int co = n0 + 64 * q + 16 * (i >> 2) + 4 * nt + (i & 3);
            wo1[j] = (unsigned)co * w1row2 + b_kb;
        }
        auto issue1 = [&](const int c, const int slot) {
#pragma unroll
            for (int j = 0; j < PB; ++j)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_1w, (lds_ptr_t)(smem + slot * BT + (wave * PB + j) * 1024), 16, wo1[j], (unsigned)c * (BK * 2), 0, 0);
            if (c < nca) {
#pragma unroll
                for (int j = 0; j < PA; ++j)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_1a, (lds_ptr_t)(smem + AOFF + slot * AT + (wave * PA + j) * 1024), 16, va1[j], (unsigned)c * (BK * 2), 0, 0);
            } else {
#pragma unroll
                for (int j = 0; j < PA; ++j)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_1b, (lds_ptr_t)(smem + AOFF + slot * AT + (wave * PA + j) * 1024), 16, vb1[j], (unsigned)(c - nca) * (BK * 2), 0, 0);
            }
        };
        // every copy of the 3^3 loop has landed (vmcnt(0) in its last steps); its last fragment reads must be over before slots are refilled
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        issue1(s1b, 0);
        if (n1 > 1) issue1(s1b + 1, 1);
        int slot = 0;
        for (int s1 = 0; s1 < n1; ++s1) {
            if (s1 + 1 < n1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PA + PB) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (s1 + 2 < n1) issue1(s1b + s1 + 2, slot == 0 ? 2 : slot - 1);      // (s1 + 2) % 3: the slot step s1 - 1 has just released
            c_aslot = (unsigned)slot * AT;
            HL_READ(wfA, afA, slot, 1);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            HL_MFMA(wfA, afA);
            slot = slot == 2 ? 0 : slot + 1;
        }
    }
    HL_STAMP(1);
    KSTAMP(3);
    // the epilogue below belongs to THIS tile; the tile state is about to move on to the next one
    const int e_mtile = mtile, e_split = split, e_smp = smp, e_l0 = l0, e_mbase = m_base, e_n0 = n0;
    // (loop form: the lane constants the epilogue derives its addresses from are made opaque here, so that the compiler recomputes the
    //  epilogue's per-lane values per tile instead of hoisting them out of the tile loop and keeping them live across the K loop)
    int fr_e = fr, fg_e = fg, lane_e = lane, tid_e = tid;
    if (PERSIST) asm volatile("" : "+v"(fr_e), "+v"(fg_e), "+v"(lane_e), "+v"(tid_e));
    // ---- intra-workgroup K reduction (the two wave groups took the two 32-deep halves of every K step)
    const int mt_base = 2 * grp;
    // The epilogue's operands (bias, time-embedding row, residual rows) are requested HERE, in front of the exchange: they are one more
    // round trip to memory nobody has touched in this launch, which the K-group exchange and its two barriers then hide
    const int cbase = e_n0 + wn * 64 + 16 * fg_e;
    const bool to_slab = p.splitk > 1 || p.raw_partial;
    const bool fused_ep = !to_slab;
    float4 ebias[4], etemb[4]; u32x4 eres[2][2]; float4 eres32[2][4];     // ebias = bias + bias2 (the fused skip's)
    const bool f32nd = p.out32 != nullptr;                     // fp32 precision mode: fp32 NDHWC output, fp32 residual
#pragma unroll
    for (int q = 0; q < 4; ++q) { ebias[q] = make_float4(0.f, 0.f, 0.f, 0.f); etemb[q] = ebias[q]; }
#pragma unroll
    for (int ml = 0; ml < 2; ++ml) {
        eres[ml][0] = (u32x4){0u, 0u, 0u, 0u}; eres[ml][1] = eres[ml][0];
#pragma unroll
        for (int q = 0; q < 4; ++q) eres32[ml][q] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // every LDS-DMA copy was waited for by the last K steps (vmcnt(0) there)
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        // every wave is past its last fragment read: the ring and the tap table are free.  Next tile of this workgroup: first macro
        // step + tap table now, so that the copies travel under the exchange and the epilogue of the current one
        if (PERSIST && vbid + (int)gridDim.x < nwg) { HL_TILE_SETUP(vbid + (int)gridDim.x); HL_TILE_EARLY(); }
        // the epilogue's operands (bias, time-embedding row, residual rows): one more round trip to memory nobody has touched in this
        // launch, requested here so that the K-group exchange and its barrier hide it (after the next tile's set-up: fewer live registers there)
        if (fused_ep) {
            if (p.bias) {
    #pragma unroll
                for (int q = 0; q < 4; ++q) ebias[q] = *reinterpret_cast<const float4*>(p.bias + cbase + 4 * q);
            }
            if (p.bias2) {
    #pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 b2 = *reinterpret_cast<const float4*>(p.bias2 + cbase + 4 * q);
                    ebias[q].x += b2.x; ebias[q].y += b2.y; ebias[q].z += b2.z; ebias[q].w += b2.w;
                }
            }
            if (p.temb) {
                const float* te = p.temb + (size_t)e_smp * p.temb_stride + cbase;
    #pragma unroll
                for (int q = 0; q < 4; ++q) etemb[q] = *reinterpret_cast<const float4*>(te + 4 * q);
            }
            if (f32nd && p.residual32 && cbase < p.CoutS) {
    #pragma unroll
                for (int ml = 0; ml < 2; ++ml) {
                    const int r_t = wm * 64 + (mt_base + ml) * 16 + fr_e;
                    if (r_t < TM && e_l0 + r_t < DHW) {
                        const float4* rp = reinterpret_cast<const float4*>(p.residual32 + (size_t)(e_mbase + r_t) * p.CoutS + cbase);
    #pragma unroll
                        for (int q = 0; q < 4; ++q) eres32[ml][q] = rp[q];
                    }
                }
            }
            if (p.residual && !p.out_f32 && !f32nd && cbase < p.CoutS) {
    #pragma unroll
                for (int ml = 0; ml < 2; ++ml) {
                    const int r_t = wm * 64 + (mt_base + ml) * 16 + fr_e;
                    if (r_t < TM && e_l0 + r_t < DHW) {
                        const u32x4* rp = reinterpret_cast<const u32x4*>(p.residual + (size_t)(e_mbase + r_t) * p.CoutS + cbase);
                        eres[ml][0] = rp[0]; eres[ml][1] = rp[1];
                    }
                }
            }
        }
        const int dst = 1 - grp;
        float* const xw = reinterpret_cast<float*>(smem + (dst ? XCH1 : XCH0));       // the half the OTHER group reads
        const float* const xr = reinterpret_cast<const float*>(smem + (grp ? XCH1 : XCH0));
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int ml = 0; ml < 2; ++ml)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    xw[((wq * 32) + (nt * 2 + ml) * 4 + r) * 64 + lane_e] = (grp == 0) ? acc[nt][2 + ml][r] : acc[nt][ml][r];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // LDS-only wait + raw barrier: the epilogue operands stay in flight
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int ml = 0; ml < 2; ++ml)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float v = xr[((wq * 32) + (nt * 2 + ml) * 4 + r) * 64 + lane_e];
                    if (grp == 0) acc[nt][ml][r] += v; else acc[nt][2 + ml][r] += v;
                }
    }

    KSTAMP(4);
    // ---- epilogue (conv_igemm.h's, with the 126-row tile mapping).  After the exchange this wave owns the 32-row block
    //      rows [64 wm + 32 grp, +32) of the tile = 16-row tiles mt_base + {0, 1}; GroupNorm partials of the whole tile go to slab
    //      row `e_mtile` (bitwise reproducible: no atomics).
    const bool do_stats = (p.stats != nullptr) && !to_slab && (p.out != nullptr || f32nd);
    float ssum[16], ssq[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) { ssum[q] = 0.f; ssq[q] = 0.f; }
#pragma unroll
    for (int ml = 0; ml < 2; ++ml) {
        const int r_t = wm * 64 + (mt_base + ml) * 16 + fr_e;                // row inside the tile
        if (r_t >= TM || e_l0 + r_t >= DHW) continue;
        const int m = e_mbase + r_t;
        float v[16];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) v[nt * 4 + r] = (grp == 0) ? acc[nt][ml][r] : acc[nt][2 + ml][r];
        if (to_slab) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 t4 = make_float4(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
                *reinterpret_cast<float4*>(slab_ptr(p.partial, e_split, p.M, p.CoutPad, p.slab_lg, m, cbase + 4 * q)) = t4;
            }
            continue;
        }
        if (p.bias || p.bias2) {
#pragma unroll
            for (int q = 0; q < 4; ++q) { v[4 * q] += ebias[q].x; v[4 * q + 1] += ebias[q].y; v[4 * q + 2] += ebias[q].z; v[4 * q + 3] += ebias[q].w; }
        }
        if (p.temb) {
#pragma unroll
            for (int q = 0; q < 4; ++q) { v[4 * q] += etemb[q].x; v[4 * q + 1] += etemb[q].y; v[4 * q + 2] += etemb[q].z; v[4 * q + 3] += etemb[q].w; }
        }
        if (p.out_f32) {
            const int sp = e_l0 + r_t;
#pragma unroll
            for (int q = 0; q < 16; ++q)
                if (cbase + q < p.CoutReal) p.out_f32[((size_t)e_smp * p.CoutReal + cbase + q) * DHW + sp] = v[q];
            continue;
        }
        if (cbase >= p.CoutS) continue;
        if (f32nd) {                                           // statistics of the exact fp32 values the GroupNorm will read
            float4* op32 = reinterpret_cast<float4*>(p.out32 + (size_t)m * p.CoutS + cbase);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 rv = eres32[ml][q];
                v[4 * q] += rv.x; v[4 * q + 1] += rv.y; v[4 * q + 2] += rv.z; v[4 * q + 3] += rv.w;
                op32[q] = make_float4(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
            }
#pragma unroll
            for (int q = 0; q < 16; ++q) { ssum[q] += v[q]; ssq[q] += v[q] * v[q]; }
            continue;
        }
        if (p.residual) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const u32x4 rv = eres[ml][h];
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
    if (do_stats) {
#define HL_ROW_ADD(X, CTRL) X += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(X), CTRL, 0xf, 0xf, true))
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            HL_ROW_ADD(ssum[q], 0x128); HL_ROW_ADD(ssum[q], 0x124); HL_ROW_ADD(ssum[q], 0x122); HL_ROW_ADD(ssum[q], 0x121);
            HL_ROW_ADD(ssq[q], 0x128); HL_ROW_ADD(ssq[q], 0x124); HL_ROW_ADD(ssq[q], 0x122); HL_ROW_ADD(ssq[q], 0x121);
        }
#undef HL_ROW_ADD
        // fold the tile's 32-row blocks through LDS (RED: a ring slot the next tile's first copies do not touch) -> ONE slab row per tile.
        // Raw barriers behind LDS-only waits: __syncthreads() would also wait for the next tile's copies and this tile's stores
        float* red = reinterpret_cast<float*>(smem + RED);                 // [2 WGM][BN couts][2]
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (fr_e == 0) {
            float* d = red + (((wm * 2 + grp) * BN) + wn * 64 + 16 * fg_e) * 2;
#pragma unroll
            for (int q = 0; q < 16; ++q) { d[2 * q] = ssum[q]; d[2 * q + 1] = ssq[q]; }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (tid_e < BN && e_n0 + tid_e < p.CoutS) {
            float s0 = 0.f, s1 = 0.f;
#pragma unroll
            for (int b = 0; b < 2 * WGM; ++b) { s0 += red[(b * BN + tid_e) * 2]; s1 += red[(b * BN + tid_e) * 2 + 1]; }
            *reinterpret_cast<float2*>(p.stats + ((size_t)e_mtile * p.CoutS + e_n0 + tid_e) * 2) = make_float2(s0, s1);
        }
    }
    KSTAMP(5);
    } while (PERSIST && (vbid += (int)gridDim.x) < nwg);   // the next trip's table barrier also orders this epilogue's LDS reads before the copies that overwrite them
#undef HL_STEP
#undef HL_MFMA
#undef HL_MASK
#undef HL_READ
#undef HL_ISSUE
#undef HL_ISSUE_W
#undef HL_ISSUE_A
#undef HL_ISSUE_A_HALF
#undef HL_ADVANCE
#undef HL_LOAD_TAB
#undef HL_TILE_SETUP
#undef HL_TILE_EARLY
#undef HL_STAMP
    KSTAMP_DRAIN(6);
#endif  // __HIP_DEVICE_COMPILE__
}
