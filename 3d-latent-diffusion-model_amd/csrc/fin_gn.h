// Split-K finalize AND the GroupNorm(+SiLU) that consumes its result, in ONE launch with NO exchange between workgroups (gfx950).
//
// A ResBlock of the UNet is GN -> SiLU -> conv1 (+ time embedding) -> GN -> SiLU -> conv2 (+ skip), an attention block starts with a
// GN (MONAI DiffusionUNetResnetBlock / SpatialAttentionBlock, reached from 3d_ldm/train_diffusion.py:197-205 / 3d_ldm/inference.py:94-99).
// At the 12^3 / 6^3 levels of the B = 1 step every conv is split over K, so "conv -> GroupNorm" was three dependent launches: conv
// (fp32 slabs), splitk_finalize_kernel (sum of the slabs + epilogue -> bf16 tensor + per-32-row statistics), gn_fused_apply_kernel
// (fold of the statistics, normalise).  Rounds 2 and 4 fused the last two behind a grid barrier / an arrival counter and lost: any
// hand-off between workgroups on this 8-XCD part costs 4 - 8 us, more than the kernel boundary it replaces.
//
// Here the workgroup IS the statistics domain: one workgroup owns one (sample, group) -- all D*H*W rows of the group's channels --
// so mean and variance never leave it.  For that the conv writes its split-K slabs PLANAR (ConvParams::slab_lg): [split][group plane]
// [row][channels of the group], i.e. everything a workgroup sums is one contiguous range per split (13.8 KB at 6^3 x 16 channels,
// 55 KB at 12^3 x 8 channels) and no cache line is shared with another workgroup.  The workgroup
//   1. sums the slabs in split order (the order of splitk_finalize_body: the un-normalised tensor is bit-identical to the two-launch
//      path), adds bias / time embedding / residual, rounds to bf16 -- values stay in registers;
//   2. reduces sum and sum of squares of the ROUNDED values (what a separate GroupNorm would read) over its 1024 threads in fp64;
//   3. normalises (+ SiLU) what it holds and stores it; the un-normalised tensor is written only where someone else reads it
//      (conv2's output is the residual stream; conv1's is not).
// 32 groups x N samples = 32 workgroups at B = 1: the slab bytes that 216 finalize blocks used to share now arrive at 32 CUs
// (332 KB each at 6^3, 498 KB at 12^3), which costs ~1 - 2 us over the plain finalize and saves a whole launch (4 - 4.5 us in-kernel
// + 1.2 - 1.6 us boundary) and one round trip of the activation.
#pragma once
#include "conv_igemm.h"

struct FinGnParams {
    FinalizeParams f;                  // planar slabs + epilogue operands; f.out = un-normalised bf16 tensor or null; f.stats unused
    const float* gamma; const float* beta;
    bf16_t* y;                         // GroupNorm(+SiLU) of the finalized tensor, [M][CoutS] bf16
    int groups, silu, lg;              // lg = log2(channels per group) = log2(plane width), 2 ... 6
    float eps;
};

constexpr int FIN_GN_THREADS = 1024;
constexpr int FIN_GN_MAX_ITEMS = 4 * FIN_GN_THREADS;     // (row, 4-channel quad) items of one (sample, group): D*H*W * cpg / 4

// 8-byte store of four bf16 (two packed pairs); WT: write-through (sc1), as store16<true> (a 64-bit store needs no wait state)
template <bool WT>
__device__ __forceinline__ void store8(void* ptr, unsigned lo, unsigned hi) {
    typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
    const u32x2_t v = {lo, hi};
    if constexpr (WT) asm volatile("global_store_dwordx2 %0, %1, off sc1" ::"v"(ptr), "v"(v) : "memory");
    else *reinterpret_cast<u32x2_t*>(ptr) = v;
}

// NIT = items per thread (1, 2 or 4); SB = splits whose loads are in flight together (NIT * SB = 16 float4 per thread)
template <bool WT, int NIT>
__global__ __launch_bounds__(FIN_GN_THREADS) void fin_gn_kernel(const FinGnParams p) {
    constexpr int SB = 16 / NIT;
    __shared__ double red[FIN_GN_THREADS / 64][2];
    __shared__ float gstat[2];
    KSTAMP_BEGIN(10);
    const FinalizeParams& f = p.f;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = blockIdx.x, n = blockIdx.y;
    const int lg = p.lg, pw = 1 << lg, qlg = lg - 2;                 // quads per row = 1 << qlg
    const int items = f.DHWo << qlg;
    const size_t slab = (size_t)f.M * f.CoutPad;
    // planar slab: plane g = rows [0, M) x pw channels; this sample's rows start at n * DHWo
    const float* plane = f.partial + ((size_t)g * f.M + (size_t)n * f.DHWo) * pw;
    int it[NIT]; bool live[NIT];
#pragma unroll
    for (int k = 0; k < NIT; ++k) { const int i = tid + k * FIN_GN_THREADS; live[k] = i < items; it[k] = live[k] ? i : items - 1; }
    float4 v[NIT];
#pragma unroll
    for (int k = 0; k < NIT; ++k) v[k] = make_float4(0.f, 0.f, 0.f, 0.f);
    // ---- 1. slab sum, splits in order 0, 1, 2, ... per element; SB splits x NIT items of loads in flight
    int s = 0;
    for (; s + SB <= f.splitk; s += SB) {
        float4 t[SB][NIT];
#pragma unroll
        for (int u = 0; u < SB; ++u)
#pragma unroll
            for (int k = 0; k < NIT; ++k) t[u][k] = *reinterpret_cast<const float4*>(plane + (size_t)(s + u) * slab + 4 * (size_t)it[k]);
#pragma unroll
        for (int u = 0; u < SB; ++u)
#pragma unroll
            for (int k = 0; k < NIT; ++k) { v[k].x += t[u][k].x; v[k].y += t[u][k].y; v[k].z += t[u][k].z; v[k].w += t[u][k].w; }
    }
    if (s < f.splitk) {                                              // the remaining < SB splits, again all in flight (clamped, masked adds)
        float4 t[SB][NIT];
#pragma unroll
        for (int u = 0; u < SB; ++u) {
            int su = s + u; if (su >= f.splitk) su = f.splitk - 1;
            if (u == 0 || s + u < f.splitk) {
#pragma unroll
                for (int k = 0; k < NIT; ++k) t[u][k] = *reinterpret_cast<const float4*>(plane + (size_t)su * slab + 4 * (size_t)it[k]);
            }
        }
#pragma unroll
        for (int u = 0; u < SB; ++u)
            if (s + u < f.splitk) {
#pragma unroll
                for (int k = 0; k < NIT; ++k) { v[k].x += t[u][k].x; v[k].y += t[u][k].y; v[k].z += t[u][k].z; v[k].w += t[u][k].w; }
            }
    }
    KSTAMP(1);
    // ---- epilogue (order of splitk_finalize_body: bias, bias2, time embedding, residual; one bf16 rounding)
    const int c0 = g << lg;
    float r4[NIT][4];                                                // the ROUNDED values (what a separate GroupNorm would read)
    unsigned pk[NIT][2];
    float ls = 0.f, lq = 0.f;
#pragma unroll
    for (int k = 0; k < NIT; ++k) {
        const int row = it[k] >> qlg, c = c0 + 4 * (it[k] & ((1 << qlg) - 1));
        const size_t m = (size_t)n * f.DHWo + row;
        float a[4] = {v[k].x, v[k].y, v[k].z, v[k].w};
        if (f.bias) { const float4 b = *reinterpret_cast<const float4*>(f.bias + c); a[0] += b.x; a[1] += b.y; a[2] += b.z; a[3] += b.w; }
        if (f.bias2) { const float4 b = *reinterpret_cast<const float4*>(f.bias2 + c); a[0] += b.x; a[1] += b.y; a[2] += b.z; a[3] += b.w; }
        if (f.temb) { const float4 b = *reinterpret_cast<const float4*>(f.temb + (size_t)n * f.temb_stride + c); a[0] += b.x; a[1] += b.y; a[2] += b.z; a[3] += b.w; }
        if (f.residual) {
            const uint2 rv = *reinterpret_cast<const uint2*>(f.residual + m * f.CoutS + c);
            a[0] += __uint_as_float(rv.x << 16); a[1] += __uint_as_float(rv.x & 0xffff0000u);
            a[2] += __uint_as_float(rv.y << 16); a[3] += __uint_as_float(rv.y & 0xffff0000u);
        }
        pk[k][0] = pack2bf(a[0], a[1]); pk[k][1] = pack2bf(a[2], a[3]);
        r4[k][0] = __uint_as_float(pk[k][0] << 16); r4[k][1] = __uint_as_float(pk[k][0] & 0xffff0000u);
        r4[k][2] = __uint_as_float(pk[k][1] << 16); r4[k][3] = __uint_as_float(pk[k][1] & 0xffff0000u);
        if (live[k]) {
            if (f.out) store8<WT>(f.out + m * f.CoutS + c, pk[k][0], pk[k][1]);
#pragma unroll
            for (int e = 0; e < 4; ++e) { ls += r4[k][e]; lq += r4[k][e] * r4[k][e]; }
        }
    }
    // ---- 2. statistics of the group: fp64 from the thread partials on (fixed order: lane tree, then waves 0 .. 15)
    double ds = (double)ls, dq = (double)lq;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { ds += __shfl_xor(ds, o, 64); dq += __shfl_xor(dq, o, 64); }
    if (lane == 0) { red[wave][0] = ds; red[wave][1] = dq; }
    __syncthreads();
    if (tid == 0) {
        double a = 0.0, b = 0.0;
        for (int w = 0; w < FIN_GN_THREADS / 64; ++w) { a += red[w][0]; b += red[w][1]; }
        const double cnt = (double)pw * (double)f.DHWo;
        const double mean = a / cnt;
        double var = b / cnt - mean * mean; if (var < 0.0) var = 0.0;
        gstat[0] = (float)mean; gstat[1] = (float)(1.0 / sqrt(var + (double)p.eps));
    }
    __syncthreads();
    KSTAMP(2);
    // ---- 3. normalise (+ activation) what this thread holds
    const float mean = gstat[0], rstd = gstat[1];
#pragma unroll
    for (int k = 0; k < NIT; ++k) {
        if (!live[k]) continue;
        const int row = it[k] >> qlg, c = c0 + 4 * (it[k] & ((1 << qlg) - 1));
        const size_t m = (size_t)n * f.DHWo + row;
        const float4 gm = *reinterpret_cast<const float4*>(p.gamma + c), bt = *reinterpret_cast<const float4*>(p.beta + c);
        const float gk[4] = {gm.x, gm.y, gm.z, gm.w}, bk[4] = {bt.x, bt.y, bt.z, bt.w};
        float y[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float a = gk[e] * rstd, b = bk[e] - mean * a;                  // the scale / shift form of gn_fused_apply_kernel
            y[e] = r4[k][e] * a + b;
            if (p.silu == 1) y[e] = silu_f(y[e]);
            else if (p.silu == 2) y[e] = y[e] > 0.f ? y[e] : 0.2f * y[e];
        }
        store8<WT>(p.y + m * f.CoutS + c, pack2bf(y[0], y[1]), pack2bf(y[2], y[3]));
    }
    KSTAMP(3);
    KSTAMP_DRAIN(4);
}

// host side: can one workgroup own a (sample, group) of this tensor?  cpg = channels per group, dhw = rows per sample
static inline bool fin_gn_ok(int C, int groups, int dhw) {
    if (groups < 1 || C % groups) return false;
    const int cpg = C / groups;
    if (cpg < 4 || cpg > 64 || (cpg & (cpg - 1))) return false;      // planes of 4 ... 64 channels, a power of two
    return (long)dhw * (cpg / 4) <= FIN_GN_MAX_ITEMS;
}
static inline int fin_gn_lg(int C, int groups) { int cpg = C / groups, lg = 0; while ((1 << lg) < cpg) ++lg; return lg; }
template <bool WT>
static inline hipError_t launch_fin_gn_t(const FinGnParams& q, int N, hipStream_t s) {
    const long items = (long)q.f.DHWo << (q.lg - 2);
    const dim3 grid(q.groups, N);
    if (items <= FIN_GN_THREADS) hipLaunchKernelGGL((fin_gn_kernel<WT, 1>), grid, dim3(FIN_GN_THREADS), 0, s, q);
    else if (items <= 2 * FIN_GN_THREADS) hipLaunchKernelGGL((fin_gn_kernel<WT, 2>), grid, dim3(FIN_GN_THREADS), 0, s, q);
    else hipLaunchKernelGGL((fin_gn_kernel<WT, 4>), grid, dim3(FIN_GN_THREADS), 0, s, q);
    return hipGetLastError();
}
static inline hipError_t launch_fin_gn(const FinGnParams& q, int N, bool wt, hipStream_t s) {
    return wt ? launch_fin_gn_t<true>(q, N, s) : launch_fin_gn_t<false>(q, N, s);
}
