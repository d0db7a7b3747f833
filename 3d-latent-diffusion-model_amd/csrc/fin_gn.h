// Split-K finalize AND the GroupNorm(+SiLU) that consumes its result, in ONE launch (gfx950).
//
// A ResBlock of the UNet is GN -> SiLU -> conv1 (+ time embedding) -> GN -> SiLU -> conv2 (+ skip) (MONAI DiffusionUNetResnetBlock,
// reached from 3d_ldm/train_diffusion.py:197-205 / 3d_ldm/inference.py:94-99).  At the 12^3 / 6^3 levels of the B = 1 step every
// conv is split over K, so "conv -> GroupNorm" was three dependent launches: conv (fp32 slabs), splitk_finalize_kernel (sum of the
// slabs + epilogue -> bf16 tensor + per-32-row statistics), gn_fused_apply_kernel (fold of the statistics, normalise).  Here the
// finalize keeps its 32 x 64 block of summed, bf16-ROUNDED values in registers, the blocks of one (sample, 64-channel slice)
// exchange their per-group partial sums through memory behind an arrival counter, and every block normalises what it holds:
//   - one launch and one round trip of the activation (0.9 MB at 12^3) less per pair;
//   - the un-normalised tensor is only written when somebody else reads it (conv2's output is the residual stream; conv1's is not).
//
// The exchange is NOT a grid barrier: a block waits only for the <= 54 blocks of its own (sample, slice), on a counter of their
// own, and the payload (16 floats per block) travels as write-through (sc1) stores that the ONE storing wave drains
// (s_waitcnt vmcnt(0)) before its own lane 0 bumps the counter (agent-scope atomic); the reader polls with sc1 loads and reads the
// payload with sc1 loads (MI355X_MICROARCH.md, "Valid forms", first row of the hand-off table).  Every block of the grid must be
// resident for the spin to terminate: the planner only takes this path for grids of at most 512 blocks of 256 threads (two per CU;
// the kernel is built for >= 2 blocks per CU), and the spin is bounded: after ~2^20 polls a block gives up, raises the error word
// of the sync buffer and finishes with the statistics it has (tests read the word; it never fires on a healthy device).
// Counters return to zero inside the launch (the last block to LEAVE a (sample, slice) clears it), so graph replays need no host work.
#pragma once
#include "conv_igemm.h"

struct FinGnParams {
    FinalizeParams f;                  // slab sum + epilogue; f.out = un-normalised bf16 tensor or null (nobody else reads it); f.stats unused
    const float* gamma; const float* beta;
    bf16_t* y;                         // GroupNorm(+SiLU) of the finalized tensor, [M][CoutS] bf16
    int groups, silu, chunks, dhw;     // chunks = 32-row blocks per sample (gridDim.x); dhw = rows per sample
    float eps;
    float* xpart;                      // [N][slices][chunks][16 groups max][2] partial (sum, sum of squares) of the block's rows, per group
    unsigned* cnt;                     // [N][slices][2]: arrivals, departures (zero between launches)
    unsigned* err;                     // one word: set to 1 when a spin gave up
};

// agent-scope relaxed accesses = global_load / global_store ... sc1 on gfx950 (they bypass the CU's L1; the compiler tracks their waits,
// so several can be in flight)
__device__ __forceinline__ void st_agent_f32(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ unsigned long long ld_agent_u64(const unsigned long long* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ unsigned ld_agent_u32(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

template <bool WT>
__global__ __launch_bounds__(256, 2) void fin_gn_kernel(const FinGnParams p) {
    __shared__ float red[4][8][16];                    // per wave: 8 channel vectors x (8 sums, 8 sums of squares)
    __shared__ float chs[64][2];                       // per channel of the slice: sum, sum of squares over the block's 32 rows
    __shared__ float gstat[16][2];                     // per group of the slice: mean, rstd
    KSTAMP_BEGIN(10);
    const FinalizeParams& f = p.f;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int bx = blockIdx.x, by = blockIdx.y, n = blockIdx.z;
    const int cv = tid & 7, rl = tid >> 3;
    const int c = by * 64 + cv * 8;
    const int row = bx * 32 + rl;                      // row inside the sample
    const int m = n * p.dhw + row;
    const bool live = row < p.dhw && c < f.CoutS;
    const int cpg = f.CoutS / p.groups;                // host guarantees 64 % cpg == 0 and CoutS % 64 == 0 or the last slice is cut at CoutS
    const int gs = 64 / cpg;                           // groups per slice (<= 16)
    // gamma / beta of the thread's 8 channels: requested first, used last
    float4 gam[2] = {make_float4(0.f, 0.f, 0.f, 0.f), make_float4(0.f, 0.f, 0.f, 0.f)}, bet[2] = {gam[0], gam[0]};
    if (c < f.CoutS) {
        gam[0] = *reinterpret_cast<const float4*>(p.gamma + c); gam[1] = *reinterpret_cast<const float4*>(p.gamma + c + 4);
        bet[0] = *reinterpret_cast<const float4*>(p.beta + c); bet[1] = *reinterpret_cast<const float4*>(p.beta + c + 4);
    }
    // ---- phase 1: the finalize (same arithmetic, same order as splitk_finalize_body) -> 8 bf16-rounded values per thread
    float x[8], ss[8], sq[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) { x[q] = 0.f; ss[q] = 0.f; sq[q] = 0.f; }
    if (live) {
        float v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = 0.f;
        const size_t slab = (size_t)f.M * f.CoutPad;
        const float* src0 = f.partial + (size_t)m * f.CoutPad + c;
        int s = 0;
        for (; s + 8 <= f.splitk; s += 8) {
            float4 a[8], b[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) { const float4* src = reinterpret_cast<const float4*>(src0 + (size_t)(s + u) * slab); a[u] = src[0]; b[u] = src[1]; }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                v[0] += a[u].x; v[1] += a[u].y; v[2] += a[u].z; v[3] += a[u].w; v[4] += b[u].x; v[5] += b[u].y; v[6] += b[u].z; v[7] += b[u].w;
            }
        }
        for (; s + 4 <= f.splitk; s += 4) {
            float4 a[4], b[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { const float4* src = reinterpret_cast<const float4*>(src0 + (size_t)(s + u) * slab); a[u] = src[0]; b[u] = src[1]; }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                v[0] += a[u].x; v[1] += a[u].y; v[2] += a[u].z; v[3] += a[u].w; v[4] += b[u].x; v[5] += b[u].y; v[6] += b[u].z; v[7] += b[u].w;
            }
        }
        for (; s < f.splitk; ++s) {
            const float4* src = reinterpret_cast<const float4*>(src0 + (size_t)s * slab);
            const float4 a = src[0], b = src[1];
            v[0] += a.x; v[1] += a.y; v[2] += a.z; v[3] += a.w; v[4] += b.x; v[5] += b.y; v[6] += b.z; v[7] += b.w;
        }
        KSTAMP(1);
        if (f.bias) {
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] += f.bias[c + q];
        }
        if (f.bias2) {
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] += f.bias2[c + q];
        }
        if (f.temb) {
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] += f.temb[(size_t)n * f.temb_stride + c + q];
        }
        if (f.residual) {
            const u32x4 rv = *reinterpret_cast<const u32x4*>(f.residual + (size_t)m * f.CoutS + c);
#pragma unroll
            for (int q = 0; q < 4; ++q) { v[2 * q] += __uint_as_float(rv[q] << 16); v[2 * q + 1] += __uint_as_float(rv[q] & 0xffff0000u); }
        }
        u32x4 o;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            o[q] = pack2bf(v[2 * q], v[2 * q + 1]);
            const float lo = __uint_as_float(o[q] << 16), hi = __uint_as_float(o[q] & 0xffff0000u);
            x[2 * q] = lo; x[2 * q + 1] = hi;
            ss[2 * q] = lo; sq[2 * q] = lo * lo; ss[2 * q + 1] = hi; sq[2 * q + 1] = hi * hi;
        }
        if (f.out) store16<WT>(f.out + (size_t)m * f.CoutS + c, o);   // the un-normalised tensor, only where another op reads it
    }
    // ---- per-channel sums over the block's 32 rows: 8 row lanes per wave by shuffles (lane = row * 8 + cv), 4 waves through LDS
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
    if (tid < 64) {
        const int v8 = tid >> 3, q = tid & 7;
        chs[tid][0] = red[0][v8][q] + red[1][v8][q] + red[2][v8][q] + red[3][v8][q];
        chs[tid][1] = red[0][v8][8 + q] + red[1][v8][8 + q] + red[2][v8][8 + q] + red[3][v8][8 + q];
    }
    __syncthreads();
    KSTAMP(2);
    // ---- phase 2 (wave 0 only): publish the block's per-group partials, arrive, wait for the slice's other blocks, fold in a fixed order
    const int unit = n * gridDim.y + by;               // (sample, slice)
    unsigned* const arrive = p.cnt + 2 * unit;
    float* const mine = p.xpart + ((size_t)unit * p.chunks + bx) * 32;
    if (wave == 0) {
        if (lane < 2 * gs) {                           // lane = group * 2 + {sum, sum of squares}
            const int g = lane >> 1, w = lane & 1;
            double acc = 0.0;
            for (int k = 0; k < cpg; ++k) acc += (double)chs[g * cpg + k][w];
            st_agent_f32(mine + lane, (float)acc);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                   // the payload has left this wave before the counter moves
        if (lane == 0) __hip_atomic_fetch_add(arrive, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned want = (unsigned)p.chunks;
        bool ok = true;
        {
            int spins = 0;
            while (ld_agent_u32(arrive) < want) {
                __builtin_amdgcn_s_sleep(4);
                if (++spins > (1 << 20)) { ok = false; break; }
            }
        }
        if (!ok && lane == 0) *p.err = 1u;
        // fold: lane = (group g, chunk lane cl); chunks cl, cl + nl, ... summed in order, then the nl lanes of a group in a fixed tree
        const int nl = 64 / gs;                        // lanes per group (4 .. 64)
        const int g = lane / nl, cl = lane - g * nl;
        double s0 = 0.0, s1 = 0.0;
        const unsigned long long* base = reinterpret_cast<const unsigned long long*>(p.xpart + (size_t)unit * p.chunks * 32 + 2 * g);
        for (int ch0 = cl; ch0 < p.chunks; ch0 += 8 * nl) {             // eight (sum, sum of squares) pairs in flight per lane
            unsigned long long t[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                int ch = ch0 + k * nl; if (ch >= p.chunks) ch = p.chunks - 1;      // clamped: unconditional loads stay in flight together
                t[k] = ld_agent_u64(base + (size_t)ch * 16);
            }
#pragma unroll
            for (int k = 0; k < 8; ++k)
                if (ch0 + k * nl < p.chunks) { s0 += (double)__uint_as_float((unsigned)t[k]); s1 += (double)__uint_as_float((unsigned)(t[k] >> 32)); }
        }
        for (int o = 1; o < nl; o <<= 1) { s0 += __shfl_xor(s0, o, 64); s1 += __shfl_xor(s1, o, 64); }
        if (cl == 0) {
            const double cnt = (double)cpg * (double)p.dhw;
            const double mean = s0 / cnt;
            double var = s1 / cnt - mean * mean; if (var < 0.0) var = 0.0;
            gstat[g][0] = (float)mean; gstat[g][1] = (float)(1.0 / sqrt(var + (double)p.eps));
        }
        // leave: the last block to leave clears both counters (nobody polls any more: every block has seen the full count)
        if (lane == 0) {
            const unsigned d = __hip_atomic_fetch_add(arrive + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (d + 1 == want || !ok) {
                __hip_atomic_store(arrive, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(arrive + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
    __syncthreads();
    KSTAMP(3);
    // ---- phase 3: normalise the values held since phase 1
    if (live) {
        const float gk[8] = {gam[0].x, gam[0].y, gam[0].z, gam[0].w, gam[1].x, gam[1].y, gam[1].z, gam[1].w};
        const float bk[8] = {bet[0].x, bet[0].y, bet[0].z, bet[0].w, bet[1].x, bet[1].y, bet[1].z, bet[1].w};
        float yv[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int g = (cv * 8 + k) / cpg;
            const float a = gk[k] * gstat[g][1];
            const float b = bk[k] - gstat[g][0] * a;
            float t = x[k] * a + b;
            if (p.silu == 1) t = silu_f(t);
            yv[k] = t;
        }
        u32x4 o;
#pragma unroll
        for (int q = 0; q < 4; ++q) o[q] = pack2bf(yv[2 * q], yv[2 * q + 1]);
        store16<WT>(p.y + (size_t)m * f.CoutS + c, o);
    }
    KSTAMP(4);
    KSTAMP_DRAIN(5);
}
