// Weight gradient of the 3D convolution, SIXTEEN-WAVE form of conv_wgrad_kernel (conv_wgrad.h: same job, data layout, LDS ring, source-offset
// table, numerics; reference call site: loss.backward() at 3d_ldm/train_diffusion.py:214).  For the convs where neither channel count fits a
// half tile (WgradParams::pair == 0: the UNet's 256 / 512-channel levels).
//
// Why: conv_wgrad_kernel's K step is a serial sum per wave -- 4 LDS-DMA pieces to issue (~145 cycles each, profiles/r05_producer_wave_experiment.txt),
// 16 transposed fragment reads, 16 MFMAs -- and all eight waves run the same phase at the same moment (profiles/r05_wgrad_kw3.txt), so the
// step takes ~1850 cycles against 512 of MFMA issue.  With sixteen waves (four per SIMD, 128 VGPRs each) a wave issues 2 pieces, reads 12
// fragments and multiplies 8 tiles per step: the per-wave non-MFMA time halves while the SIMD's MFMA work stays the same.
//   wave = (K group grp of 2) x (cout quarter wa of 4: 32 couts) x (cin half wb of 2: 64 cins); accumulators 2 x 4 tiles = 32 registers.
#pragma once
#include "conv_wgrad.h"

template <int ABL1 = 0>
__global__ __launch_bounds__(1024, 4) void conv_wgrad_w16_kernel(const WgradParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int KV = 64;                     // voxels per K step
    constexpr int TR = 256;                    // bytes per LDS row (128 channels)
    constexpr int STAGE = 2 * KV * TR;         // dY tile + X tile = 32 KiB
    constexpr int NS = 4, PF = NS - 1, LPS = 2;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef __attribute__((address_space(3))) void* lds_ptr_t;
    typedef __attribute__((ext_vector_type(4))) short s16x4;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 3, wq = wave & 7;
    const int wa = wq & 3, wb = wq >> 2;       // wave tile: couts [32 wa, +32) x cins [64 wb, +64)
    int bid = blockIdx.x;
    const int ci_t = bid % p.ci_tiles; bid /= p.ci_tiles;
    const int co_t = bid % p.co_tiles; bid /= p.co_tiles;
    const int taps = p.ksize * p.ksize * p.ksize;
    const int tap = bid % taps; const int split = bid / taps;          // one tap per workgroup (the several-taps forms stay with conv_wgrad_kernel)
    const int kk = p.ksize * p.ksize;
    const int kd = tap / kk, kh = (tap - kd * kk) / p.ksize, kw = tap - kd * kk - kh * p.ksize;
    constexpr int TABW = 2 * KV;                                                       // table entries per buffer: dY rows, X rows
    const int DinU = p.Din << p.ups, HinU = p.Hin << p.ups, WinU = p.Win << p.ups;
    const int HWo = p.Hout * p.Wout, DHWo = p.Dout * HWo;
    const int steps_all = (p.M + KV - 1) / KV;
    const int sps = (steps_all + p.ksplit - 1) / p.ksplit;
    const int s_begin = split * sps;
    const int nsteps = (s_begin + sps < steps_all ? s_begin + sps : steps_all) - s_begin;     // may be <= 0 for a trailing split

    // ---- source-offset table (double buffered, behind the ring): thread r < 128 owns row slot r of a ring stage (64 dY rows
    //      then 64 X rows), keeps the output coordinates of the voxel that slot holds at the step being prepared, advances
    //      them by 64 voxels per step and publishes the row's byte offset (0xFFFFFFFF: beyond M / tap in the zero padding ->
    //      the copy writes zeros).  The copy path of every lane is then a table read + add per piece, loop free, so the whole
    //      K step is one basic block in which copies and fragment reads ride between the MFMAs.  Three table buffers: a fast
    //      wave may publish for step k + 1 while a slow one still reads the table of step k - 1.
    unsigned* const tab = reinterpret_cast<unsigned*>(smem + NS * STAGE);
    const bool owner = tid < TABW, own_x = tid >= KV;
    const int own_row = own_x ? tid - KV : tid;
    int vw = 0, vh = 0, vd = 0, vn = 0, vm = s_begin * KV + own_row;
    if (owner) {
        int m = vm;
        vn = m / DHWo; m -= vn * DHWo; vd = m / HWo; m -= vd * HWo; vh = m / p.Wout; vw = m - vh * p.Wout;
    }
    // 64 voxels ahead, decomposed once (q_w < Wout and q_h < Hout, so those carries wrap at most once; volumes smaller
    // than 64 voxels make q_d >= Dout, hence the loop on the depth carry)
    const int q_d = KV / HWo, q_h = (KV - q_d * HWo) / p.Wout, q_w = KV - q_d * HWo - q_h * p.Wout;
#define WG_PUBLISH(PAR) do {                                                                                  \
        if (owner) {                                                                                          \
            unsigned off_ = 0xFFFFFFFFu;                                                                      \
            if (!own_x) { if (vm < p.M) off_ = (unsigned)vm * (unsigned)(p.cdy * 2); }                   \
            else {                                                                                            \
                const int id = vd * p.stride + kd - p.pad, ih = vh * p.stride + kh - p.pad, iw = vw * p.stride + kw - p.pad; \
                const bool ok_ = (vm < p.M) & ((unsigned)id < (unsigned)DinU) & ((unsigned)ih < (unsigned)HinU) & ((unsigned)iw < (unsigned)WinU); \
                const int src_ = ((vn * p.Din + (id >> p.ups)) * p.Hin + (ih >> p.ups)) * p.Win + (iw >> p.ups); \
                if (ok_) off_ = (unsigned)src_ * (unsigned)(p.cx * 2);                                        \
            }                                                                                                 \
            tab[(PAR) * TABW + tid] = off_;                                                                   \
            vm += KV;                                                                                         \
            vw += q_w; if (vw >= p.Wout) { vw -= p.Wout; ++vh; }                                              \
            vh += q_h; if (vh >= p.Hout) { vh -= p.Hout; ++vd; }                                              \
            vd += q_d; while (vd >= p.Dout) { vd -= p.Dout; ++vn; }                                           \
        }                                                                                                     \
    } while (0)

    // ---- loader lanes: each wave copies ONE piece (4 voxel rows x 256 B) of the dY tile and one of the X tile per step.
    // lane -> (row = lane / 16 inside the piece, physical 16-B chunk = lane % 16); 32-byte blocks are XOR-swizzled by
    // f(row) = (row & 3) + 4 * ((row >> 3) & 1) so the transposed reads below are bank-conflict free.
    const int prow = lane >> 4, pch = lane & 15;
    const int l_row = wave * 4 + prow;                                   // voxel row inside the step
    unsigned l_ady, l_ax;                                                // channel byte added to the row offset (or out of range)
    {
        const int f = (l_row & 3) + 4 * ((l_row >> 3) & 1);
        const unsigned kb = (unsigned)(((((pch >> 1) ^ f) << 1) | (pch & 1)) * 16);
        l_ady = ((unsigned)co_t * 256u + kb < (unsigned)p.cdy * 2u) ? (unsigned)co_t * 256u + kb : 0xFFFFFFFFu;
        l_ax = ((unsigned)ci_t * 256u + kb < (unsigned)p.cx * 2u) ? (unsigned)ci_t * 256u + kb : 0xFFFFFFFFu;
    }
    __amdgpu_buffer_rsrc_t rs_dy = __builtin_amdgcn_make_buffer_rsrc((void*)p.dy, 0, (int)((unsigned)p.M * (unsigned)p.cdy * 2u), 0x00020000);
    __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(
        (void*)p.x, 0, (int)((unsigned)(p.N * p.Din * p.Hin * p.Win) * (unsigned)p.cx * 2u), 0x00020000);
    int ld_s = 0;

    // table reads of one step's two copies (issued ahead of the waits), then the copies themselves
#define WG_TAB(T, PAR) do { T[0] = tab[(PAR) * TABW + l_row]; T[1] = tab[(PAR) * TABW + KV + l_row]; } while (0)
#define WG_COPIES(T) do {                                                                                     \
        char* st_ = smem + (ld_s % NS) * STAGE;                                                               \
        const unsigned vo_dy = ((T[0] == 0xFFFFFFFFu) | (l_ady == 0xFFFFFFFFu)) ? 0xFFFFFFFFu : T[0] + l_ady; \
        const unsigned vo_x = ((T[1] == 0xFFFFFFFFu) | (l_ax == 0xFFFFFFFFu)) ? 0xFFFFFFFFu : T[1] + l_ax;    \
        if (!(ABL1 & 4)) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_dy, (lds_ptr_t)(st_ + wave * 1024), 16, vo_dy, 0, 0, 0); \
        if (!(ABL1 & 4)) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, (lds_ptr_t)(st_ + KV * TR + wave * 1024), 16, vo_x, 0, 0, 0); \
        ++ld_s;                                                                                               \
    } while (0)

    // ---- fragments: this wave group's 32 voxels of the step are rows [32 grp, 32 grp + 32).  Lane (i = lane & 15,
    //      g = lane >> 4) gets k = 8g .. 8g+7 of column i: two transposed reads (rows 8g..8g+3 and 8g+4..8g+7).
    const int fi = lane & 15, fg = lane >> 4, tq = fi >> 2, tp = fi & 3;
    const int r_lo = 32 * grp + 8 * fg + tq, r_hi = r_lo + 4;            // voxel rows this lane ADDRESSES
    const int f_lo = (r_lo & 3) + 4 * ((r_lo >> 3) & 1), f_hi = (r_hi & 3) + 4 * ((r_hi >> 3) & 1);
    int a_lo[2], a_hi[2], b_lo[4], b_hi[4];                              // byte offsets inside a ring slot
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int ca = wa * 2 + (t & 1), cb = wb * 4 + t;                // 32-byte column block (16 channels) in the tile
        if (t < 2) {
            a_lo[t] = r_lo * TR + ((ca ^ f_lo) << 5) + tp * 8;
            a_hi[t] = r_hi * TR + ((ca ^ f_hi) << 5) + tp * 8;
        }
        b_lo[t] = KV * TR + r_lo * TR + ((cb ^ f_lo) << 5) + tp * 8;
        b_hi[t] = KV * TR + r_hi * TR + ((cb ^ f_hi) << 5) + tp * 8;
    }
    f32x4 acc[2][4];                                                     // [cout tile][cin tile]
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
    bf16x8 afA[2], bfA[4], afB[2], bfB[4];
#define WG_READ(AF, BF, SLOT) do {                                                                  \
        if (ABL1 & 16) break;                                                                       \
        const char* sb_ = smem + (SLOT) * STAGE;                                                    \
        _Pragma("unroll") for (int t = 0; t < 2; ++t) {                                             \
            const s16x4 al_ = ds_read_tr16_b64_raw(sb_ + a_lo[t]); \
            const s16x4 ah_ = ds_read_tr16_b64_raw(sb_ + a_hi[t]); \
            AF[t] = (bf16x8){al_[0], al_[1], al_[2], al_[3], ah_[0], ah_[1], ah_[2], ah_[3]};       \
        }                                                                                           \
        _Pragma("unroll") for (int t = 0; t < 4; ++t) {                                             \
            const s16x4 bl_ = ds_read_tr16_b64_raw(sb_ + b_lo[t]); \
            const s16x4 bh_ = ds_read_tr16_b64_raw(sb_ + b_hi[t]); \
            BF[t] = (bf16x8){bl_[0], bl_[1], bl_[2], bl_[3], bh_[0], bh_[1], bh_[2], bh_[3]};       \
        }                                                                                           \
    } while (0)
#define WG_MFMA(AF, BF) do {                                                                        \
        if (ABL1 & 8) break;                                                                        \
        _Pragma("unroll") for (int a = 0; a < 2; ++a)                                               \
            _Pragma("unroll") for (int b = 0; b < 4; ++b)                                           \
                acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(AF[a], BF[b], acc[a][b], 0, 0, 0); \
    } while (0)
    // table protocol: publish #k (offsets of step k) goes to buffer k % 3; the copies of step k (in step k - NS, behind that
    // step's barrier) read it; publish #k happens at the top of step k - NS - 1 and overwrites #k-3, read two barriers earlier.
    // Steady-state step: owners publish (branchy, in front of the waits), then ONE basic block: table reads, waits,
    // barrier, 4 copies + 16 transposed fragment reads interleaved with the 16 MFMAs of the step.
#define WG_FAST(S, AC, BC, AN, BN) do {                                                             \
        if (dbgf & 2048) asm volatile("s_nop 0");              /* opaque branch: one basic block per step */ \
        WG_PUBLISH(((S) + NS + 1) % 3);                                                             \
        __builtin_amdgcn_sched_barrier(0);                                                          \
        unsigned t_[2];                                                                             \
        WG_TAB(t_, ((S) + NS) % 3);                            /* published one barrier ago: readable ahead of the waits */ \
        __builtin_amdgcn_s_waitcnt(0xC07F);                    /* fragments of step S, table values, own table write retired */ \
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((PF - 1) * LPS) : "memory");                       \
        __builtin_amdgcn_s_barrier();                                                               \
        asm volatile("" ::: "memory");                                                              \
        __builtin_amdgcn_sched_barrier(0);                     /* nothing (MFMAs on asm-read fragments) moves above the waits */ \
        WG_COPIES(t_);                                                                              \
        WG_READ(AN, BN, ((S) + 1) % NS);                                                            \
        WG_MFMA(AC, BC);                                                                            \
        _Pragma("unroll") for (int i_ = 0; i_ < 2; ++i_) {                                          \
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                      \
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);                                      \
        }                                                                                           \
        _Pragma("unroll") for (int i_ = 0; i_ < 6; ++i_) {                                          \
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                      \
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);                                      \
        }                                                                                           \
    } while (0)
#define WG_HALF(S, AC, BC, AN, BN) do {                                                             \
        __builtin_amdgcn_sched_barrier(0);                                                          \
        __builtin_amdgcn_s_waitcnt(0xC07F);                                                         \
        __builtin_amdgcn_sched_barrier(0);                                                          \
        if ((S) + 1 < nsteps) {                                                                     \
            if ((S) + PF < nsteps) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((PF - 1) * LPS) : "memory"); \
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                   \
            __builtin_amdgcn_s_barrier();                                                           \
            asm volatile("" ::: "memory");                                                          \
            if (ld_s < nsteps) { unsigned t_[2]; WG_TAB(t_, ((S) + NS) % 3); WG_COPIES(t_); }       \
            WG_READ(AN, BN, ((S) + 1) % NS);                                                        \
            WG_PUBLISH(((S) + NS + 1) % 3);                                                         \
        }                                                                                           \
        WG_MFMA(AC, BC);                                                                            \
    } while (0)

    const int dbgf = p.dbg;
    WG_PUBLISH(0);
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NS; ++i) {
        if (i < nsteps) { unsigned t_[2]; WG_TAB(t_, i % 3); WG_COPIES(t_); }
        __syncthreads();                                       // table #i read by every wave
        WG_PUBLISH((i + 1) % 3);                               // #1 .. #NS
        __syncthreads();
    }
    if (nsteps > PF) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PF * LPS) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    WG_READ(afA, bfA, 0);
    int s = 0;
    // steady state: step s + NS exists (copies issued every step), two steps per iteration for the static fragment sets
    for (; s + NS + 2 <= nsteps; s += 2) {
        WG_FAST(s, afA, bfA, afB, bfB);
        WG_FAST(s + 1, afB, bfB, afA, bfA);
    }
    for (; s < nsteps; s += 2) {
        WG_HALF(s, afA, bfA, afB, bfB);
        if (s + 1 >= nsteps) break;
        WG_HALF(s + 1, afB, bfB, afA, bfA);
    }
#undef WG_FAST
#undef WG_HALF
#undef WG_MFMA
#undef WG_READ
#undef WG_COPIES
#undef WG_TAB
#undef WG_PUBLISH

    // ---- reduce the two wave groups through LDS (group 1 -> group 0), then group 0 stores ------------------------
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();
    float* xch = reinterpret_cast<float*>(smem);
    if (grp == 1) {
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b)
#pragma unroll
                for (int r = 0; r < 4; ++r) xch[((wq * 32) + (a * 4 + b) * 4 + r) * 64 + lane] = acc[a][b][r];
    }
    __syncthreads();
    if (grp == 0) {
        // accumulator: col = lane & 15 -> cin, row = 4 fg + r -> cout
        const size_t tap_off = (size_t)split * p.slab_stride + (size_t)tap * p.Cout * p.dw_ld + p.dw_ci_off;
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const int ci = ci_t * 128 + wb * 64 + b * 16 + fi;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int co = co_t * 128 + wa * 32 + a * 16 + 4 * fg + r;
                    const float v = acc[a][b][r] + xch[((wq * 32) + (a * 4 + b) * 4 + r) * 64 + lane];
                    if (co < p.Cout && ci < p.Cin) p.dw[tap_off + (size_t)co * p.dw_ld + ci] = v;
                }
            }
    }
#endif
}
