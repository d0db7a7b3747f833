// Split-K slab fold + GroupNorm(+SiLU) of the folded tensor in ONE launch -- an EXPERIMENT that is off by default (LDM_FIN_GN=1 plans it).
//
// At 12^3 / 6^3 every conv of the UNet is a split-K launch followed by splitk_finalize_kernel (fold the slabs, bias / time embedding /
// residual, bf16 store, per-32-row GroupNorm partials) followed by gn_fused_apply_kernel (fold the partials, normalise): two launches of
// ~6 us each that move ~1 MB.  Both have the same thread mapping (32 rows x 64 channels per block, thread = one row x 8 channels), so this
// kernel runs the finalize body, keeps the folded row in registers, meets every other block at a grid-wide barrier (the GroupNorm
// statistics need every row), then runs the partial fold and normalises from registers.  Arithmetic and summation order are those of the
// two kernels it replaces: results are bit-identical (tests/test_gpu_launch_fusion.py), and the headline plan drops from 152 to 125
// launches.
//
// Measured (round 2, MI355X, same box): 424 steps/s with it vs 473 without -- each of the 27 fused launches is ~9 us SLOWER than the two
// launches it replaces.  Per-op events (tools/plan_trace.py, LDM_FIN_GN_DBG ablations): pair 17.1 us -> fused 22.8 us, of which the wait
// at the barrier is 4.0 us at 216 blocks (the arrivals are same-address device-scope atomics, ~20 ns each, serialised at the memory side
// because the 8 XCDs' L2s are not coherent), the agent-scope release fence (buffer_wbl2) 2.6 us, the acquire (buffer_inv) 0.7 us; and even
// with all three removed the merged kernel takes 15.9 us: it cannot start the GroupNorm's loads before the fold has been stored, while the
// gap between two dependent kernels of a replayed graph is only ~2 us.  Conclusion recorded in DESIGN.md section 5: at these sizes the step is
// bound by the chain of cross-XCD memory round trips inside each small kernel, not by the number of launches, and a software grid
// barrier costs more than a launch boundary.
//
// Grid barrier.  All blocks of the launch must be co-resident; the planner only emits this op for grids <= FIN_GN_MAX_BLOCKS (256-thread
// blocks, ~19 KB of LDS: the chip holds 2048 of them), and a block that waits longer than FIN_GN_TIMEOUT_TICKS of the 100 MHz wall clock
// gives up, raises the fault word and returns, so a mis-sized launch ends as a reported error (ldm_model_sync_faults), not as a hung GPU.
// The barrier is sense-reversing and self-resetting (state = {arrivals, generation, faults}), so graph replays need no host reset.
// Visibility across the 8 XCDs: the partials travel through agent-scope release / acquire fences around the atomics, which write back /
// invalidate L2 as a kernel boundary would.
#pragma once
#include "conv_igemm.h"
#include "norm_elem.h"

constexpr int FIN_GN_MAX_BLOCKS = 1024;
constexpr long long FIN_GN_TIMEOUT_TICKS = 20000000LL;          // 0.2 s of the 100 MHz constant clock

struct FinGnParams {
    FinalizeParams f;                                  // f.out = the folded tensor (still written: later ops read it), f.stats = its partial rows
    GnFusedParams g;                                   // g.xa == f.out, g.sa == f.stats, g.nrb_a = 32-row granules per sample
    unsigned* sync;                                    // {arrivals, generation, faults} (model-owned, zero at creation)
    unsigned nblocks;
    int dbg;                                           // timing ablations (LDM_FIN_GN_DBG; results are wrong): 1 no release fence, 2 no acquire fence, 4 no wait
};

// returns false after a timeout (the caller must not touch shared results then)
__device__ __forceinline__ bool grid_barrier(unsigned* st, const unsigned nblocks, const int dbg = 0) {
    __shared__ int ok_s;
    __syncthreads();                                   // every wave's stores are issued (workgroup-scope release)
    if (threadIdx.x == 0) {
        const unsigned g0 = __hip_atomic_load(&st[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (!(dbg & 1)) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");   // this block's partials reach memory before the arrival counts
        int ok = 1;
        const unsigned old = __hip_atomic_fetch_add(&st[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (old == nblocks - 1) {
            __hip_atomic_store(&st[0], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // reset for the next launch, then open the gate
            __hip_atomic_fetch_add(&st[1], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        } else if (!(dbg & 4)) {
            const long long t0 = wall_clock64();
            while (__hip_atomic_load(&st[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == g0) {
                __builtin_amdgcn_s_sleep(2);
                if (wall_clock64() - t0 > FIN_GN_TIMEOUT_TICKS) {
                    __hip_atomic_fetch_add(&st[2], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    ok = 0; break;
                }
            }
        }
        if (!(dbg & 2)) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");   // drop stale lines before the partials of other XCDs are read
        ok_s = ok;
    }
    __syncthreads();
    return ok_s != 0;
}

// grid = (32-row granules of all samples, 64-channel slices of ca + cb), 256 threads
template <bool WT>
__global__ __launch_bounds__(256) void fin_gn_kernel(const FinGnParams p) {
    __shared__ float red[4][8][16];
    __shared__ __attribute__((aligned(16))) float part[8][96][4];
    __shared__ double csum[192][2];
    __shared__ float gstat[64][2];
    const GnFusedParams& g = p.g;
    const int tid = threadIdx.x;
    const int C = g.ca + g.cb, cpg = C / g.groups;
    const int c0 = blockIdx.y * 64;
    const int vec = tid & 7, rl = tid >> 3;
    const int c = c0 + vec * 8;
    const int m = blockIdx.x * 32 + rl;                // row over all samples
    const int M = g.N * g.DHW;
    const int n = (blockIdx.x * 32) / g.DHW;           // granules never straddle samples (host: N == 1 or DHW % 32 == 0)
    const bool active = c < C && m < M;
    u32x4 o = {0u, 0u, 0u, 0u};
    float4 gam[2] = {make_float4(0.f, 0.f, 0.f, 0.f), make_float4(0.f, 0.f, 0.f, 0.f)}, bet[2] = {gam[0], gam[0]};
    if (c0 < g.ca) {                                   // host: ca % 64 == 0, so a slice never straddles the two sources
        splitk_finalize_body<WT>(p.f, blockIdx.x, blockIdx.y, red, o);
    } else if (active) {
        o = *reinterpret_cast<const u32x4*>(g.xb + (size_t)m * g.cb + (c - g.ca));
    }
    if (c < C) {
        gam[0] = *reinterpret_cast<const float4*>(g.gamma + c); gam[1] = *reinterpret_cast<const float4*>(g.gamma + c + 4);
        bet[0] = *reinterpret_cast<const float4*>(g.beta + c); bet[1] = *reinterpret_cast<const float4*>(g.beta + c + 4);
    }
    if (!grid_barrier(p.sync, p.nblocks, p.dbg)) return;
    const int blocks_per_sample = (g.DHW + 31) / 32;
    gn_fused_fold(g, n, blockIdx.y, (int)blockIdx.x == n * blocks_per_sample, part, csum, gstat);
    if (c >= C) return;
    const int g_lo = c0 / cpg;
    float a[8], b[8];
    const float gk[8] = {gam[0].x, gam[0].y, gam[0].z, gam[0].w, gam[1].x, gam[1].y, gam[1].z, gam[1].w};
    const float bk[8] = {bet[0].x, bet[0].y, bet[0].z, bet[0].w, bet[1].x, bet[1].y, bet[1].z, bet[1].w};
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int gi = (c + k) / cpg - g_lo;
        a[k] = gk[k] * gstat[gi][1];
        b[k] = bk[k] - gstat[gi][0] * a[k];
    }
    if (g.ab && (int)blockIdx.x == n * blocks_per_sample && rl == 0) {
#pragma unroll
        for (int k = 0; k < 8; ++k) { g.ab[((size_t)n * C + c + k) * 2] = a[k]; g.ab[((size_t)n * C + c + k) * 2 + 1] = b[k]; }
    }
    if (!active) return;
    u32x4 y;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        float lo = __uint_as_float(o[j] << 16) * a[2 * j] + b[2 * j];
        float hi = __uint_as_float(o[j] & 0xffff0000u) * a[2 * j + 1] + b[2 * j + 1];
        if (g.silu == 1) { lo = silu_f(lo); hi = silu_f(hi); }
        else if (g.silu == 2) { lo = lo > 0.f ? lo : 0.2f * lo; hi = hi > 0.f ? hi : 0.2f * hi; }
        y[j] = pack2bf(lo, hi);
    }
    store16<WT>(g.out + (size_t)m * C + c, y);
}
