// Weight gradient of the 3x3x3 stride-1 "same" convolution, THREE kw TAPS PER WORKGROUP over one shared X tile (gfx950, bf16 MFMA
// 16x16x32).  Training; reference call site: loss.backward() at 3d_ldm/train_diffusion.py:214 reaching
// every nn.Conv3d of the UNet.
//
//   dW[(kd, kh, kw)][co][ci] = sum over output voxels m of  dY[m][co] * X[m + (kd - 1) H W + (kh - 1) W + (kw - 1)][ci]      (zero outside)
//
// conv_wgrad_kernel (conv_wgrad.h) gives every tap its own workgroup: per 64-voxel K step it copies a 16 KiB dY tile and a 16 KiB X tile
// for 128 MFMAs, and a CU takes in 45 - 59 GB/s by LDS-DMA (profiles/r05_halo_pp_ablations.txt), so the step takes
// 1800 cycles against 512 of MFMA issue at 24^3 (round 5 profile of the training step: 82 us per 256 -> 256 launch, 0.24 of peak, 47 % of
// the weight-gradient time of a step).  Here the W-halo trick of the forward kernel is applied to the contraction over voxels:
//   * workgroup = one (kd, kh) x 128 couts x 64 cins x ALL THREE kw; per K step the 64 dY rows (16 KiB) and 66 X rows (voxels m0 - 1 ..
//     m0 + 64 at the centre kw, 8.25 KiB) feed 192 MFMAs: 24.3 KiB per 768 MFMA cycles instead of 32 KiB per 512;
//   * the kw = 0 / 2 fragments are the same LDS rows read one voxel row down / up (the transposed read ds_read_b64_tr_b16 addresses rows per
//     lane, so a row shift is an address); a voxel whose neighbour lies across a W border is dropped by zeroing that k element of the X
//     fragment (one element per lane, kw side and step at most: W >= 8), D / H borders and the ends of the voxel range by the copy's zero fill;
//   * same pipeline as conv_wgrad_kernel: 4-slot LDS ring filled by buffer_load ... lds, source-offset table published one step ahead,
//     counted vmcnt, raw barrier, fragments of step s + 1 read under the MFMAs of step s; 8 waves = 2 K groups (32 voxels of every step
//     each) x (64 couts x 32 cins x 3 kw) wave tiles, 96 accumulator registers.
// EXPERIMENTS BUILDS ONLY (make EXTRA=-DLDM_EXPERIMENTS, LDM_WGRAD_KW3=1).  MEASURED (round 5, profiles/r05_wgrad_kw3.txt): parity-green on the
// 11 operator cases of tests/test_gpu_ops.py on the first run, 251 VGPRs, no scratch -- and 81.4 us against conv_wgrad_kernel's 83.8 at
// 256 -> 256, 24^3 (ksplit 3 / 2): no gain.  The bytes were not the bound.  Ablations of the K loop (72 steps, ~14 us of the launch are
// prologue, the 21 MB of partial matrices and the launch itself): no copies -15 us, no fragment reads -15, no masks -10, no barrier -7.5, no
// MFMAs -28; MFMAs alone 41 us.  The step is a serial SUM: copies ~500 cycles (3 - 4 pieces per wave at the ~145 cycles a wave needs to
// issue one 1 KiB LDS-DMA piece, profiles/r05_producer_wave_experiment.txt) + transposed reads ~500 + masks ~340 + barrier ~250 + MFMAs
// ~900, because all eight waves run the same phase at the same time.  conv_wgrad_kernel's step is the same sum with 4 pieces and 16 reads
// per 16 MFMAs of a wave.
// Output: fp32 [ksplit][27][Cout][dw_ld] like conv_wgrad_kernel (deterministic, no atomics).  Numerics: the same products in the same
// per-workgroup order over voxels as conv_wgrad_kernel with the same ksplit (fp32 accumulation in the MFMA; the two K groups are added last).
#pragma once
#include "conv_wgrad.h"

constexpr int WG3_KV = 64, WG3_YR = 256, WG3_XR = 128, WG3_XROWS = 72;
constexpr int WG3_STAGE = WG3_KV * WG3_YR + WG3_XROWS * WG3_XR;            // 25600 B per ring stage
constexpr int WG3_NS = 4;
constexpr int WG3_TABW = WG3_KV + WG3_XROWS;                               // 136 table entries per buffer
constexpr int WG3_LDS = WG3_NS * WG3_STAGE + 3 * WG3_TABW * 4 + 64;

template <int ABL1 = 0>
__global__ __launch_bounds__(512, 2) void conv_wgrad_kw3_kernel(const WgradParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int KV = WG3_KV, YR = WG3_YR, XR = WG3_XR, XB = KV * YR, STAGE = WG3_STAGE, NS = WG3_NS, PF = NS - 1, TABW = WG3_TABW;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef __attribute__((address_space(3))) void* lds_ptr_t;
    typedef __attribute__((ext_vector_type(4))) short s16x4;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2, wq = wave & 3;
    const int wa = wq & 1, wb = wq >> 1;       // wave tile: couts [64 wa, +64) x cins [32 wb, +32) x 3 kw
    int bid = blockIdx.x;
    const int ci_t = bid % p.ci_tiles; bid /= p.ci_tiles;                  // 64-cin tiles here
    const int co_t = bid % p.co_tiles; bid /= p.co_tiles;
    const int pr = bid % 9; const int split = bid / 9;                    // (kd, kh) pair; voxel-range split
    const int kd = pr / 3, kh = pr - 3 * kd;
    const int HWo = p.Hout * p.Wout, DHWo = p.Dout * HWo;
    const int dsrc = (kd - 1) * HWo + (kh - 1) * p.Wout;                  // source voxel = m' + dsrc at the centre kw (same-size conv)
    const int steps_all = (p.M + KV - 1) / KV;
    const int sps = (steps_all + p.ksplit - 1) / p.ksplit;
    const int s_begin = split * sps;
    const int nsteps = (s_begin + sps < steps_all ? s_begin + sps : steps_all) - s_begin;     // may be <= 0 for a trailing split

    // ---- source-offset table (triple buffered, behind the ring): thread r < 64 owns dY row r, thread 64 + r < 136 X row r of a stage
    //      (X row r <-> voxel m0 + r - 1; rows 66 .. 71 pad the ninth 8-row piece); each keeps the coordinates of its voxel, advances them
    //      by 64 voxels per step and publishes the row's byte offset (0xFFFFFFFF -> the copy writes zeros).
    unsigned* const tab = reinterpret_cast<unsigned*>(smem + NS * STAGE);
    const bool owner = tid < TABW, own_x = tid >= KV;
    const int own_row = own_x ? tid - KV : tid;
    int vw = 0, vh = 0, vd = 0, vm = s_begin * KV + own_row - (own_x ? 1 : 0);
    bool lead = false;                                                     // X row 0 of the first step of the volume: voxel -1
    if (owner) {
        int m = vm;
        if (m < 0) { m += KV; lead = true; }                               // coordinates of voxel 63: its next step
        const int vn = m / DHWo; m -= vn * DHWo; vd = m / HWo; m -= vd * HWo; vh = m / p.Wout; vw = m - vh * p.Wout;
    }
    const int q_d = KV / HWo, q_h = (KV - q_d * HWo) / p.Wout, q_w = KV - q_d * HWo - q_h * p.Wout;
#define W3_PUBLISH(PAR) do {                                                                                  \
        if (owner) {                                                                                          \
            unsigned off_ = 0xFFFFFFFFu;                                                                      \
            if (lead) { lead = false; vm += KV; }                                                             \
            else {                                                                                            \
                if (!own_x) { if (vm < p.M) off_ = (unsigned)vm * (unsigned)(p.cdy * 2); }                    \
                else if (own_row < KV + 2 && vm < p.M) {                                                      \
                    const int id = vd + kd - 1, ih = vh + kh - 1;                                             \
                    if (((unsigned)id < (unsigned)p.Din) & ((unsigned)ih < (unsigned)p.Hin)) off_ = (unsigned)(vm + dsrc) * (unsigned)(p.cx * 2); \
                }                                                                                             \
                vm += KV;                                                                                     \
                vw += q_w; if (vw >= p.Wout) { vw -= p.Wout; ++vh; }                                          \
                vh += q_h; if (vh >= p.Hout) { vh -= p.Hout; ++vd; }                                          \
                vd += q_d; while (vd >= p.Dout) vd -= p.Dout;                                                 \
            }                                                                                                 \
            tab[(PAR) * TABW + tid] = off_;                                                                   \
        }                                                                                                     \
    } while (0)

    // ---- loader lanes.  dY: every wave copies 2 pieces (4 voxel rows x 256 B), 32-byte blocks XOR-swizzled by f(row) = (row & 3) +
    //      4 ((row >> 3) & 1) as in conv_wgrad_kernel.  X: wave w copies piece w (8 rows x 128 B), wave 0 also piece 8; 32-byte blocks
    //      XOR-swizzled by g(row) = ((row >> 1) & 1) + 2 ((row >> 3) & 1): the 8 rows a half wave's transposed read touches (4 consecutive rows
    //      from any start, twice, 8 rows apart) land on 8 different (row parity, block) bank groups.
    int ly_row[2]; unsigned ly_a[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int row = (wave * 2 + j) * 4 + (lane >> 4), pch = lane & 15;
        const int f = (row & 3) + 4 * ((row >> 3) & 1);
        const unsigned kb = (unsigned)(((((pch >> 1) ^ f) << 1) | (pch & 1)) * 16);
        ly_row[j] = row;
        ly_a[j] = ((unsigned)co_t * 256u + kb < (unsigned)p.cdy * 2u) ? (unsigned)co_t * 256u + kb : 0xFFFFFFFFu;
    }
    int lx_row[2]; unsigned lx_a[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int row = (j == 0 ? wave : 8) * 8 + (lane >> 3), pch = lane & 7;
        const int g = ((row >> 1) & 1) + 2 * ((row >> 3) & 1);
        const unsigned kb = (unsigned)(((((pch >> 1) ^ g) << 1) | (pch & 1)) * 16);
        lx_row[j] = KV + row;
        lx_a[j] = ((unsigned)ci_t * 128u + kb < (unsigned)p.cx * 2u) ? (unsigned)ci_t * 128u + kb : 0xFFFFFFFFu;
    }
    __amdgpu_buffer_rsrc_t rs_dy = __builtin_amdgcn_make_buffer_rsrc((void*)p.dy, 0, (int)((unsigned)p.M * (unsigned)p.cdy * 2u), 0x00020000);
    __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, (int)((unsigned)p.M * (unsigned)p.cx * 2u), 0x00020000);
    int ld_s = 0;
#define W3_TAB(T, PAR) do {                                                                                   \
        T[0] = tab[(PAR) * TABW + ly_row[0]]; T[1] = tab[(PAR) * TABW + ly_row[1]];                           \
        T[2] = tab[(PAR) * TABW + lx_row[0]]; T[3] = tab[(PAR) * TABW + lx_row[1]];                           \
    } while (0)
#define W3_COPIES(T) do {                                                                                     \
        char* st_ = smem + (ld_s % NS) * STAGE;                                                               \
        _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                                       \
            const unsigned vo_ = ((T[j] == 0xFFFFFFFFu) | (ly_a[j] == 0xFFFFFFFFu)) ? 0xFFFFFFFFu : T[j] + ly_a[j]; \
            if (!(ABL1 & 4)) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_dy, (lds_ptr_t)(st_ + (wave * 2 + j) * 1024), 16, vo_, 0, 0, 0); \
        }                                                                                                     \
        {                                                                                                     \
            const unsigned vo_ = ((T[2] == 0xFFFFFFFFu) | (lx_a[0] == 0xFFFFFFFFu)) ? 0xFFFFFFFFu : T[2] + lx_a[0]; \
            if (!(ABL1 & 4)) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, (lds_ptr_t)(st_ + XB + wave * 1024), 16, vo_, 0, 0, 0); \
        }                                                                                                     \
        if (wave == 0) {                                                                                      \
            const unsigned vo_ = ((T[3] == 0xFFFFFFFFu) | (lx_a[1] == 0xFFFFFFFFu)) ? 0xFFFFFFFFu : T[3] + lx_a[1]; \
            if (!(ABL1 & 4)) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, (lds_ptr_t)(st_ + XB + 8 * 1024), 16, vo_, 0, 0, 0); \
        }                                                                                                     \
        ++ld_s;                                                                                               \
    } while (0)
    // all but the copies of the (PF - 1) youngest steps have landed: wave 0 issues 4 per step, the others 3
#define W3_WAIT_RING() do {                                                                                   \
        if (wave == 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((PF - 1) * 4) : "memory");                    \
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"((PF - 1) * 3) : "memory");                              \
    } while (0)

    // ---- fragments: this wave group's 32 voxels of the step are dY rows [32 grp, +32) and X rows [32 grp + kw, +32).  Lane (i = lane & 15,
    //      g = lane >> 4) gets k = 8 g .. 8 g + 7 of column i: two transposed reads (rows 8 g .. + 3 and 8 g + 4 .. + 7).
    const int fi = lane & 15, fg = lane >> 4, tq = fi >> 2, tp = fi & 3;
    const int r_lo = 32 * grp + 8 * fg + tq, r_hi = r_lo + 4;
    int a_lo[4], a_hi[4], b_lo[3][2], b_hi[3][2];
    {
        const int f_lo = (r_lo & 3) + 4 * ((r_lo >> 3) & 1), f_hi = (r_hi & 3) + 4 * ((r_hi >> 3) & 1);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int ca = wa * 4 + t;
            a_lo[t] = r_lo * YR + ((ca ^ f_lo) << 5) + tp * 8;
            a_hi[t] = r_hi * YR + ((ca ^ f_hi) << 5) + tp * 8;
        }
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
            const int rl = r_lo + kw, rh = r_hi + kw;
            const int g_lo = ((rl >> 1) & 1) + 2 * ((rl >> 3) & 1), g_hi = ((rh >> 1) & 1) + 2 * ((rh >> 3) & 1);
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int cb = wb * 2 + t;
                b_lo[kw][t] = XB + rl * XR + ((cb ^ g_lo) << 5) + tp * 8;
                b_hi[kw][t] = XB + rh * XR + ((cb ^ g_hi) << 5) + tp * 8;
            }
        }
    }
    // W-border masks: element j of this lane's X fragments is X row 32 grp + 8 fg + j + kw <-> voxel m0 - 1 + 32 grp + 8 fg + j + kw.  kw = 0 pairs
    // it with output voxel m = that voxel + 1: dropped when w(m) == 0, i.e. when the X voxel has w == W - 1; kw = 2: dropped when it has w == 0.
    // cw = w of the voxel of element 0 at kw = 0; j0 = W - 1 - cw is the element to drop at kw = 0, j0 - 1 (mod W) the one at kw = 2.
    const int W = p.Wout, q64 = KV % W;
    int cw = (s_begin * KV + 32 * grp + 8 * fg + W - 1) % W;

    f32x4 acc[3][4][2];                                                     // [kw][cout tile][cin tile]
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) acc[k][a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
    bf16x8 afA[4], afB[4], bfA[3][2], bfB[3][2];
#define W3_READ(AF, BF, SLOT) do {                                                                  \
        if (ABL1 & 16) break;                                                                       \
        const char* sb_ = smem + (SLOT) * STAGE;                                                    \
        _Pragma("unroll") for (int t = 0; t < 4; ++t) {                                             \
            const s16x4 al_ = ds_read_tr16_b64_raw(sb_ + a_lo[t]);                                  \
            const s16x4 ah_ = ds_read_tr16_b64_raw(sb_ + a_hi[t]);                                  \
            AF[t] = (bf16x8){al_[0], al_[1], al_[2], al_[3], ah_[0], ah_[1], ah_[2], ah_[3]};       \
        }                                                                                           \
        _Pragma("unroll") for (int kw = 0; kw < 3; ++kw)                                            \
            _Pragma("unroll") for (int t = 0; t < 2; ++t) {                                         \
                const s16x4 bl_ = ds_read_tr16_b64_raw(sb_ + b_lo[kw][t]);                          \
                const s16x4 bh_ = ds_read_tr16_b64_raw(sb_ + b_hi[kw][t]);                          \
                BF[kw][t] = (bf16x8){bl_[0], bl_[1], bl_[2], bl_[3], bh_[0], bh_[1], bh_[2], bh_[3]}; \
            }                                                                                       \
    } while (0)
    // masks of the step whose fragments are about to be multiplied, applied in registers; then cw moves on 64 voxels
#define W3_MASK(BF) do {                                                                            \
        if (ABL1 & 32) break;                                                                       \
        const int j0_ = W - 1 - cw, j2_ = j0_ == 0 ? W - 1 : j0_ - 1;                               \
        _Pragma("unroll") for (int d = 0; d < 4; ++d) {                                             \
            const unsigned m0_ = (j0_ >> 1) == d ? ((j0_ & 1) ? 0x0000FFFFu : 0xFFFF0000u) : 0xFFFFFFFFu; \
            const unsigned m2_ = (j2_ >> 1) == d ? ((j2_ & 1) ? 0x0000FFFFu : 0xFFFF0000u) : 0xFFFFFFFFu; \
            _Pragma("unroll") for (int t = 0; t < 2; ++t) {                                         \
                u32x4 v0_ = __builtin_bit_cast(u32x4, BF[0][t]); v0_[d] &= m0_; BF[0][t] = __builtin_bit_cast(bf16x8, v0_); \
                u32x4 v2_ = __builtin_bit_cast(u32x4, BF[2][t]); v2_[d] &= m2_; BF[2][t] = __builtin_bit_cast(bf16x8, v2_); \
            }                                                                                       \
        }                                                                                           \
        cw += q64; if (cw >= W) cw -= W;                                                            \
    } while (0)
#define W3_MFMA(AF, BF) do {                                                                        \
        if (ABL1 & 8) break;                                                                        \
        _Pragma("unroll") for (int k = 0; k < 3; ++k) {                                             \
            const int kw = (k + 1) % 3;                        /* the unmasked centre tap first */   \
            _Pragma("unroll") for (int a = 0; a < 4; ++a)                                           \
                _Pragma("unroll") for (int b = 0; b < 2; ++b)                                       \
                    acc[kw][a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(AF[a], BF[kw][b], acc[kw][a][b], 0, 0, 0); \
        }                                                                                           \
    } while (0)
    // table protocol (as conv_wgrad_kernel): publish #k goes to buffer k % 3; the copies of step k (issued in step k - NS, behind that step's
    // barrier) read it; publish #k happens at the top of step k - NS - 1 and overwrites #k - 3, read two barriers earlier.
#define W3_FAST(S, AC, BC, AN, BN) do {                                                             \
        if (dbgf & 2048) asm volatile("s_nop 0");              /* opaque branch: one basic block per step */ \
        W3_PUBLISH(((S) + NS + 1) % 3);                                                             \
        __builtin_amdgcn_sched_barrier(0);                                                          \
        unsigned t_[4];                                                                             \
        W3_TAB(t_, ((S) + NS) % 3);                                                                 \
        __builtin_amdgcn_s_waitcnt(0xC07F);                    /* fragments of step S, table values, own table write retired */ \
        W3_WAIT_RING();                                                                             \
        if (!(ABL1 & 64)) __builtin_amdgcn_s_barrier();                                             \
        asm volatile("" ::: "memory");                                                              \
        __builtin_amdgcn_sched_barrier(0);                                                          \
        W3_COPIES(t_);                                                                              \
        W3_MASK(BC);                                                                                \
        W3_READ(AN, BN, ((S) + 1) % NS);                                                            \
        W3_MFMA(AC, BC);                                                                            \
    } while (0)
#define W3_HALF(S, AC, BC, AN, BN) do {                                                             \
        __builtin_amdgcn_sched_barrier(0);                                                          \
        __builtin_amdgcn_s_waitcnt(0xC07F);                                                         \
        __builtin_amdgcn_sched_barrier(0);                                                          \
        if ((S) + 1 < nsteps) {                                                                     \
            if ((S) + PF < nsteps) W3_WAIT_RING();                                                  \
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                   \
            __builtin_amdgcn_s_barrier();                                                           \
            asm volatile("" ::: "memory");                                                          \
            if (ld_s < nsteps) { unsigned t_[4]; W3_TAB(t_, ((S) + NS) % 3); W3_COPIES(t_); }       \
            W3_MASK(BC);                                                                            \
            W3_READ(AN, BN, ((S) + 1) % NS);                                                        \
            W3_PUBLISH(((S) + NS + 1) % 3);                                                         \
        } else W3_MASK(BC);                                                                         \
        __builtin_amdgcn_sched_barrier(0);                                                          \
        W3_MFMA(AC, BC);                                                                            \
    } while (0)

    const int dbgf = p.dbg;
    W3_PUBLISH(0);
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NS; ++i) {
        if (i < nsteps) { unsigned t_[4]; W3_TAB(t_, i % 3); W3_COPIES(t_); }
        __syncthreads();                                       // table #i read by every wave
        W3_PUBLISH((i + 1) % 3);                               // #1 .. #NS
        __syncthreads();
    }
    if (nsteps > PF) {
        if (wave == 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PF * 4) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PF * 3) : "memory");
    } else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (nsteps > 0) W3_READ(afA, bfA, 0);
    int s = 0;
    for (; s + NS + 2 <= nsteps; s += 2) {
        W3_FAST(s, afA, bfA, afB, bfB);
        W3_FAST(s + 1, afB, bfB, afA, bfA);
    }
    for (; s < nsteps; s += 2) {
        W3_HALF(s, afA, bfA, afB, bfB);
        if (s + 1 >= nsteps) break;
        W3_HALF(s + 1, afB, bfB, afA, bfA);
    }
#undef W3_FAST
#undef W3_HALF
#undef W3_MFMA
#undef W3_MASK
#undef W3_READ
#undef W3_WAIT_RING
#undef W3_COPIES
#undef W3_TAB
#undef W3_PUBLISH

    // ---- reduce the two wave groups through LDS (group 1 -> group 0) one kw at a time, then group 0 stores -----------
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    float* xch = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) {
        __syncthreads();
        if (grp == 1) {
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b)
#pragma unroll
                    for (int r = 0; r < 4; ++r) xch[((wq * 32) + (a * 2 + b) * 4 + r) * 64 + lane] = acc[kw][a][b][r];
        }
        __syncthreads();
        if (grp == 0) {
            // accumulator: col = lane & 15 -> cin, row = 4 fg + r -> cout
            const size_t tap_off = (size_t)split * p.slab_stride + (size_t)(pr * 3 + kw) * p.Cout * p.dw_ld + p.dw_ci_off;
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    const int ci = ci_t * 64 + wb * 32 + b * 16 + fi;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int co = co_t * 128 + wa * 64 + a * 16 + 4 * fg + r;
                        const float v = acc[kw][a][b][r] + xch[((wq * 32) + (a * 2 + b) * 4 + r) * 64 + lane];
                        if (co < p.Cout && ci < p.Cin) p.dw[tap_off + (size_t)co * p.dw_ld + ci] = v;
                    }
                }
        }
    }
#endif
}

// ---- sixteen-wave form: wave = (K group of 2) x (cout quarter of 4: 32 couts) x (cin half of 2: 32 cins) x 3 kw, 48 accumulator registers,
//      four waves per SIMD: a wave issues 1 - 2 copies, 20 fragment reads and 12 MFMAs per K step (conv_wgrad_w16.h says why)
//      ONE fragment set (the reads of step s + 1 follow the MFMAs of step s; two sets spill 155 registers at the 128 a wave has here).
//      MEASURED (profiles/r05_wgrad_w16.txt): parity-green, 90.6 us against the eight-wave form's 80.5 at 256 -> 256, 24^3: slower.  LDM_WGRAD_KW3=2.
template <int ABL1 = 0>
__global__ __launch_bounds__(1024, 4) void conv_wgrad_kw3w16_kernel(const WgradParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int KV = WG3_KV, YR = WG3_YR, XR = WG3_XR, XB = KV * YR, STAGE = WG3_STAGE, NS = WG3_NS, PF = NS - 1, TABW = WG3_TABW;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef __attribute__((address_space(3))) void* lds_ptr_t;
    typedef __attribute__((ext_vector_type(4))) short s16x4;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 3, wq = wave & 7;
    const int wa = wq & 3, wb = wq >> 2;       // wave tile: couts [32 wa, +32) x cins [32 wb, +32) x 3 kw
    int bid = blockIdx.x;
    const int ci_t = bid % p.ci_tiles; bid /= p.ci_tiles;                  // 64-cin tiles here
    const int co_t = bid % p.co_tiles; bid /= p.co_tiles;
    const int pr = bid % 9; const int split = bid / 9;                    // (kd, kh) pair; voxel-range split
    const int kd = pr / 3, kh = pr - 3 * kd;
    const int HWo = p.Hout * p.Wout, DHWo = p.Dout * HWo;
    const int dsrc = (kd - 1) * HWo + (kh - 1) * p.Wout;                  // source voxel = m' + dsrc at the centre kw (same-size conv)
    const int steps_all = (p.M + KV - 1) / KV;
    const int sps = (steps_all + p.ksplit - 1) / p.ksplit;
    const int s_begin = split * sps;
    const int nsteps = (s_begin + sps < steps_all ? s_begin + sps : steps_all) - s_begin;     // may be <= 0 for a trailing split

    // ---- source-offset table (triple buffered, behind the ring): thread r < 64 owns dY row r, thread 64 + r < 136 X row r of a stage
    //      (X row r <-> voxel m0 + r - 1; rows 66 .. 71 pad the ninth 8-row piece); each keeps the coordinates of its voxel, advances them
    //      by 64 voxels per step and publishes the row's byte offset (0xFFFFFFFF -> the copy writes zeros).
    unsigned* const tab = reinterpret_cast<unsigned*>(smem + NS * STAGE);
    const bool owner = tid < TABW, own_x = tid >= KV;
    const int own_row = own_x ? tid - KV : tid;
    int vw = 0, vh = 0, vd = 0, vm = s_begin * KV + own_row - (own_x ? 1 : 0);
    bool lead = false;                                                     // X row 0 of the first step of the volume: voxel -1
    if (owner) {
        int m = vm;
        if (m < 0) { m += KV; lead = true; }                               // coordinates of voxel 63: its next step
        const int vn = m / DHWo; m -= vn * DHWo; vd = m / HWo; m -= vd * HWo; vh = m / p.Wout; vw = m - vh * p.Wout;
    }
    const int q_d = KV / HWo, q_h = (KV - q_d * HWo) / p.Wout, q_w = KV - q_d * HWo - q_h * p.Wout;
#define W3_PUBLISH(PAR) do {                                                                                  \
        if (owner) {                                                                                          \
            unsigned off_ = 0xFFFFFFFFu;                                                                      \
            if (lead) { lead = false; vm += KV; }                                                             \
            else {                                                                                            \
                if (!own_x) { if (vm < p.M) off_ = (unsigned)vm * (unsigned)(p.cdy * 2); }                    \
                else if (own_row < KV + 2 && vm < p.M) {                                                      \
                    const int id = vd + kd - 1, ih = vh + kh - 1;                                             \
                    if (((unsigned)id < (unsigned)p.Din) & ((unsigned)ih < (unsigned)p.Hin)) off_ = (unsigned)(vm + dsrc) * (unsigned)(p.cx * 2); \
                }                                                                                             \
                vm += KV;                                                                                     \
                vw += q_w; if (vw >= p.Wout) { vw -= p.Wout; ++vh; }                                          \
                vh += q_h; if (vh >= p.Hout) { vh -= p.Hout; ++vd; }                                          \
                vd += q_d; while (vd >= p.Dout) vd -= p.Dout;                                                 \
            }                                                                                                 \
            tab[(PAR) * TABW + tid] = off_;                                                                   \
        }                                                                                                     \
    } while (0)

    // ---- loader lanes.  dY: every wave copies 2 pieces (4 voxel rows x 256 B), 32-byte blocks XOR-swizzled by f(row) = (row & 3) +
    //      4 ((row >> 3) & 1) as in conv_wgrad_kernel.  X: wave w copies piece w (8 rows x 128 B), wave 0 also piece 8; 32-byte blocks
    //      XOR-swizzled by g(row) = ((row >> 1) & 1) + 2 ((row >> 3) & 1): the 8 rows a half wave's transposed read touches (4 consecutive rows
    //      from any start, twice, 8 rows apart) land on 8 different (row parity, block) bank groups.
    int ly_row[1]; unsigned ly_a[1];
#pragma unroll
    for (int j = 0; j < 1; ++j) {
        const int row = (wave + j) * 4 + (lane >> 4), pch = lane & 15;
        const int f = (row & 3) + 4 * ((row >> 3) & 1);
        const unsigned kb = (unsigned)(((((pch >> 1) ^ f) << 1) | (pch & 1)) * 16);
        ly_row[j] = row;
        ly_a[j] = ((unsigned)co_t * 256u + kb < (unsigned)p.cdy * 2u) ? (unsigned)co_t * 256u + kb : 0xFFFFFFFFu;
    }
    int lx_row[1]; unsigned lx_a[1];
#pragma unroll
    for (int j = 0; j < 1; ++j) {
        const int row = (wave < 9 ? wave : 8) * 8 + (lane >> 3), pch = lane & 7;
        const int g = ((row >> 1) & 1) + 2 * ((row >> 3) & 1);
        const unsigned kb = (unsigned)(((((pch >> 1) ^ g) << 1) | (pch & 1)) * 16);
        lx_row[j] = KV + row;
        lx_a[j] = ((unsigned)ci_t * 128u + kb < (unsigned)p.cx * 2u) ? (unsigned)ci_t * 128u + kb : 0xFFFFFFFFu;
    }
    __amdgpu_buffer_rsrc_t rs_dy = __builtin_amdgcn_make_buffer_rsrc((void*)p.dy, 0, (int)((unsigned)p.M * (unsigned)p.cdy * 2u), 0x00020000);
    __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, (int)((unsigned)p.M * (unsigned)p.cx * 2u), 0x00020000);
    int ld_s = 0;
#define W3_TAB(T, PAR) do {                                                                                   \
        T[0] = tab[(PAR) * TABW + ly_row[0]]; T[2] = tab[(PAR) * TABW + lx_row[0]];                           \
    } while (0)
#define W3_COPIES(T) do {                                                                                     \
        char* st_ = smem + (ld_s % NS) * STAGE;                                                               \
        {                                                                                                     \
            const unsigned vo_ = ((T[0] == 0xFFFFFFFFu) | (ly_a[0] == 0xFFFFFFFFu)) ? 0xFFFFFFFFu : T[0] + ly_a[0]; \
            if (!(ABL1 & 4)) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_dy, (lds_ptr_t)(st_ + wave * 1024), 16, vo_, 0, 0, 0); \
        }                                                                                                     \
        if (wave < 9) {                                                                                       \
            const unsigned vo_ = ((T[2] == 0xFFFFFFFFu) | (lx_a[0] == 0xFFFFFFFFu)) ? 0xFFFFFFFFu : T[2] + lx_a[0]; \
            if (!(ABL1 & 4)) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, (lds_ptr_t)(st_ + XB + wave * 1024), 16, vo_, 0, 0, 0); \
        }                                                                                                     \
        ++ld_s;                                                                                               \
    } while (0)
    // all but the copies of the (PF - 1) youngest steps have landed: waves 0 - 8 issue 2 per step, the others 1
#define W3_WAIT_RING() do {                                                                                   \
        if (wave < 9) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((PF - 1) * 2) : "memory");                     \
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"((PF - 1) * 1) : "memory");                              \
    } while (0)

    // ---- fragments: this wave group's 32 voxels of the step are dY rows [32 grp, +32) and X rows [32 grp + kw, +32).  Lane (i = lane & 15,
    //      g = lane >> 4) gets k = 8 g .. 8 g + 7 of column i: two transposed reads (rows 8 g .. + 3 and 8 g + 4 .. + 7).
    const int fi = lane & 15, fg = lane >> 4, tq = fi >> 2, tp = fi & 3;
    const int r_lo = 32 * grp + 8 * fg + tq, r_hi = r_lo + 4;
    int a_lo[2], a_hi[2], b_lo[3][2], b_hi[3][2];
    {
        const int f_lo = (r_lo & 3) + 4 * ((r_lo >> 3) & 1), f_hi = (r_hi & 3) + 4 * ((r_hi >> 3) & 1);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int ca = wa * 2 + t;
            a_lo[t] = r_lo * YR + ((ca ^ f_lo) << 5) + tp * 8;
            a_hi[t] = r_hi * YR + ((ca ^ f_hi) << 5) + tp * 8;
        }
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
            const int rl = r_lo + kw, rh = r_hi + kw;
            const int g_lo = ((rl >> 1) & 1) + 2 * ((rl >> 3) & 1), g_hi = ((rh >> 1) & 1) + 2 * ((rh >> 3) & 1);
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int cb = wb * 2 + t;
                b_lo[kw][t] = XB + rl * XR + ((cb ^ g_lo) << 5) + tp * 8;
                b_hi[kw][t] = XB + rh * XR + ((cb ^ g_hi) << 5) + tp * 8;
            }
        }
    }
    // W-border masks: element j of this lane's X fragments is X row 32 grp + 8 fg + j + kw <-> voxel m0 - 1 + 32 grp + 8 fg + j + kw.  kw = 0 pairs
    // it with output voxel m = that voxel + 1: dropped when w(m) == 0, i.e. when the X voxel has w == W - 1; kw = 2: dropped when it has w == 0.
    // cw = w of the voxel of element 0 at kw = 0; j0 = W - 1 - cw is the element to drop at kw = 0, j0 - 1 (mod W) the one at kw = 2.
    const int W = p.Wout, q64 = KV % W;
    int cw = (s_begin * KV + 32 * grp + 8 * fg + W - 1) % W;

    f32x4 acc[3][2][2];                                                     // [kw][cout tile][cin tile]
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) acc[k][a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
    bf16x8 afA[2], bfA[3][2];                   // ONE fragment set: the reads of step s + 1 follow the MFMAs of step s (four waves per SIMD cover their latency)
#define W3_READ(AF, BF, SLOT) do {                                                                  \
        if (ABL1 & 16) break;                                                                       \
        const char* sb_ = smem + (SLOT) * STAGE;                                                    \
        _Pragma("unroll") for (int t = 0; t < 2; ++t) {                                             \
            const s16x4 al_ = ds_read_tr16_b64_raw(sb_ + a_lo[t]);                                  \
            const s16x4 ah_ = ds_read_tr16_b64_raw(sb_ + a_hi[t]);                                  \
            AF[t] = (bf16x8){al_[0], al_[1], al_[2], al_[3], ah_[0], ah_[1], ah_[2], ah_[3]};       \
        }                                                                                           \
        _Pragma("unroll") for (int kw = 0; kw < 3; ++kw)                                            \
            _Pragma("unroll") for (int t = 0; t < 2; ++t) {                                         \
                const s16x4 bl_ = ds_read_tr16_b64_raw(sb_ + b_lo[kw][t]);                          \
                const s16x4 bh_ = ds_read_tr16_b64_raw(sb_ + b_hi[kw][t]);                          \
                BF[kw][t] = (bf16x8){bl_[0], bl_[1], bl_[2], bl_[3], bh_[0], bh_[1], bh_[2], bh_[3]}; \
            }                                                                                       \
    } while (0)
    // masks of the step whose fragments are about to be multiplied, applied in registers; then cw moves on 64 voxels
#define W3_MASK(BF) do {                                                                            \
        if (ABL1 & 32) break;                                                                       \
        const int j0_ = W - 1 - cw, j2_ = j0_ == 0 ? W - 1 : j0_ - 1;                               \
        _Pragma("unroll") for (int d = 0; d < 4; ++d) {                                             \
            const unsigned m0_ = (j0_ >> 1) == d ? ((j0_ & 1) ? 0x0000FFFFu : 0xFFFF0000u) : 0xFFFFFFFFu; \
            const unsigned m2_ = (j2_ >> 1) == d ? ((j2_ & 1) ? 0x0000FFFFu : 0xFFFF0000u) : 0xFFFFFFFFu; \
            _Pragma("unroll") for (int t = 0; t < 2; ++t) {                                         \
                u32x4 v0_ = __builtin_bit_cast(u32x4, BF[0][t]); v0_[d] &= m0_; BF[0][t] = __builtin_bit_cast(bf16x8, v0_); \
                u32x4 v2_ = __builtin_bit_cast(u32x4, BF[2][t]); v2_[d] &= m2_; BF[2][t] = __builtin_bit_cast(bf16x8, v2_); \
            }                                                                                       \
        }                                                                                           \
        cw += q64; if (cw >= W) cw -= W;                                                            \
    } while (0)
#define W3_MFMA(AF, BF) do {                                                                        \
        if (ABL1 & 8) break;                                                                        \
        _Pragma("unroll") for (int k = 0; k < 3; ++k) {                                             \
            const int kw = (k + 1) % 3;                        /* the unmasked centre tap first */   \
            _Pragma("unroll") for (int a = 0; a < 2; ++a)                                           \
                _Pragma("unroll") for (int b = 0; b < 2; ++b)                                       \
                    acc[kw][a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(AF[a], BF[kw][b], acc[kw][a][b], 0, 0, 0); \
        }                                                                                           \
    } while (0)
    // table protocol (as conv_wgrad_kernel): publish #k goes to buffer k % 3; the copies of step k (issued in step k - NS, behind that step's
    // barrier) read it; publish #k happens at the top of step k - NS - 1 and overwrites #k - 3, read two barriers earlier.
#define W3_FAST(S, AC, BC, AN, BN) do {                                                             \
        if (dbgf & 2048) asm volatile("s_nop 0");              /* opaque branch: one basic block per step */ \
        W3_PUBLISH(((S) + NS + 1) % 3);                                                             \
        __builtin_amdgcn_sched_barrier(0);                                                          \
        unsigned t_[3];                                                                             \
        W3_TAB(t_, ((S) + NS) % 3);                                                                 \
        __builtin_amdgcn_s_waitcnt(0xC07F);                    /* fragments of step S, table values, own table write retired */ \
        W3_WAIT_RING();                                                                             \
        if (!(ABL1 & 64)) __builtin_amdgcn_s_barrier();                                             \
        asm volatile("" ::: "memory");                                                              \
        __builtin_amdgcn_sched_barrier(0);                                                          \
        W3_COPIES(t_);                                                                              \
        W3_MASK(BC);                                                                                \
        W3_MFMA(AC, BC);                                                                            \
        __builtin_amdgcn_sched_barrier(0);                                                          \
        W3_READ(AN, BN, ((S) + 1) % NS);                                                            \
    } while (0)
#define W3_HALF(S, AC, BC, AN, BN) do {                                                             \
        __builtin_amdgcn_sched_barrier(0);                                                          \
        __builtin_amdgcn_s_waitcnt(0xC07F);                                                         \
        __builtin_amdgcn_sched_barrier(0);                                                          \
        W3_MASK(BC);                                                                                \
        W3_MFMA(AC, BC);                                                                            \
        __builtin_amdgcn_sched_barrier(0);                                                          \
        if ((S) + 1 < nsteps) {                                                                     \
            if ((S) + PF < nsteps) W3_WAIT_RING();                                                  \
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                   \
            __builtin_amdgcn_s_barrier();                                                           \
            asm volatile("" ::: "memory");                                                          \
            if (ld_s < nsteps) { unsigned t_[3]; W3_TAB(t_, ((S) + NS) % 3); W3_COPIES(t_); }       \
            W3_READ(AN, BN, ((S) + 1) % NS);                                                        \
            W3_PUBLISH(((S) + NS + 1) % 3);                                                         \
        }                                                                                           \
    } while (0)

    const int dbgf = p.dbg;
    W3_PUBLISH(0);
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NS; ++i) {
        if (i < nsteps) { unsigned t_[3]; W3_TAB(t_, i % 3); W3_COPIES(t_); }
        __syncthreads();                                       // table #i read by every wave
        W3_PUBLISH((i + 1) % 3);                               // #1 .. #NS
        __syncthreads();
    }
    if (nsteps > PF) {
        if (wave < 9) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PF * 2) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PF * 1) : "memory");
    } else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (nsteps > 0) W3_READ(afA, bfA, 0);
    int s = 0;
    for (; s + NS + 1 <= nsteps; ++s) W3_FAST(s, afA, bfA, afA, bfA);
    for (; s < nsteps; ++s) W3_HALF(s, afA, bfA, afA, bfA);
#undef W3_FAST
#undef W3_HALF
#undef W3_MFMA
#undef W3_MASK
#undef W3_READ
#undef W3_WAIT_RING
#undef W3_COPIES
#undef W3_TAB
#undef W3_PUBLISH

    // ---- reduce the two wave groups through LDS (group 1 -> group 0) one kw at a time, then group 0 stores -----------
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    float* xch = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) {
        __syncthreads();
        if (grp == 1) {
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b)
#pragma unroll
                    for (int r = 0; r < 4; ++r) xch[((wq * 16) + (a * 2 + b) * 4 + r) * 64 + lane] = acc[kw][a][b][r];
        }
        __syncthreads();
        if (grp == 0) {
            // accumulator: col = lane & 15 -> cin, row = 4 fg + r -> cout
            const size_t tap_off = (size_t)split * p.slab_stride + (size_t)(pr * 3 + kw) * p.Cout * p.dw_ld + p.dw_ci_off;
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    const int ci = ci_t * 64 + wb * 32 + b * 16 + fi;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int co = co_t * 128 + wa * 32 + a * 16 + 4 * fg + r;
                        const float v = acc[kw][a][b][r] + xch[((wq * 16) + (a * 2 + b) * 4 + r) * 64 + lane];
                        if (co < p.Cout && ci < p.Cin) p.dw[tap_off + (size_t)co * p.dw_ld + ci] = v;
                    }
                }
        }
    }
#endif
}
