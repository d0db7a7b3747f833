// GroupNorm (+SiLU), layout packing, timestep-embedding MLP and scheduler step kernels (gfx950).
//
// These replace torch.nn.GroupNorm / SiLU / Linear and monai DDPMScheduler/DDIMScheduler element-wise math
// that the reference reaches through MONAI (call sites 3d_ldm/inference.py:94-99,
// 3d_ldm/train_diffusion.py:197-205).  All are HBM/L2-bound: 16-byte vector accesses, fp32 statistics,
// wavefront shuffles + one LDS hop for the reductions.
#pragma once
#include "common.h"

// ------------------------------------------------------------------------------------------------
// GroupNorm statistics, pass 1: per-(sample, channel) partial sum / sum-of-squares over a slab of voxels.
// Input is the channel-concatenation (xa | xb) in NDHWC bf16.  Thread = 8 channels x strided voxels.
// partial layout: [N][nslab][C][2] fp32.
struct GnStatsParams {
    const bf16_t* xa; const bf16_t* xb; int ca, cb;
    int DHW; int nslab; int rows_per_slab;
    float* partial;
};

__global__ __launch_bounds__(256) void gn_stats_kernel(const GnStatsParams p) {
    __shared__ float red[256 * 16];
    const int C = p.ca + p.cb;
    const int cvec = C / 8;                       // 8-channel vectors per voxel
    const int n = blockIdx.y, slab = blockIdx.x;
    const int r0 = slab * p.rows_per_slab;
    int r1 = r0 + p.rows_per_slab; if (r1 > p.DHW) r1 = p.DHW;
    float s[8], q[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) { s[k] = 0.f; q[k] = 0.f; }
    // vector slots: cv = tid % cvp, row lane = tid / cvp, where cvp = cvec rounded up to a divisor-friendly stride
    const int tid = threadIdx.x;
    const int rows_par = 256 / cvec > 0 ? 256 / cvec : 1;   // voxels processed concurrently by the block
    if (cvec <= 256) {
        const int cv = tid % cvec, rl = tid / cvec;
        if (rl < rows_par) {
            const int c = cv * 8;
            const bool second = c >= p.ca;
            const bf16_t* base = second ? p.xb : p.xa;
            const int cs = second ? p.cb : p.ca;
            const int cc = second ? c - p.ca : c;
            for (int r = r0 + rl; r < r1; r += rows_par) {
                const u32x4 v = *reinterpret_cast<const u32x4*>(base + ((size_t)n * p.DHW + r) * cs + cc);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float lo = __uint_as_float(v[k] << 16), hi = __uint_as_float(v[k] & 0xffff0000u);
                    s[2 * k] += lo; q[2 * k] += lo * lo; s[2 * k + 1] += hi; q[2 * k + 1] += hi * hi;
                }
            }
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) { red[tid * 16 + k] = s[k]; red[tid * 16 + 8 + k] = q[k]; }
        __syncthreads();
        // threads 0..cvec-1 fold the row lanes of their channel vector
        if (tid < cvec) {
            for (int rl2 = 1; rl2 < rows_par; ++rl2) {
                const int t2 = rl2 * cvec + tid;
#pragma unroll
                for (int k = 0; k < 8; ++k) { s[k] += red[t2 * 16 + k]; q[k] += red[t2 * 16 + 8 + k]; }
            }
            float* dst = p.partial + (((size_t)n * p.nslab + slab) * C + tid * 8) * 2;
#pragma unroll
            for (int k = 0; k < 8; ++k) { dst[2 * k] = s[k]; dst[2 * k + 1] = q[k]; }
        }
    } else {
        // very wide tensors (C > 2048): each thread walks several channel vectors, all rows of the slab
        for (int cv = tid; cv < cvec; cv += 256) {
            const int c = cv * 8;
            const bool second = c >= p.ca;
            const bf16_t* base = second ? p.xb : p.xa;
            const int cs = second ? p.cb : p.ca;
            const int cc = second ? c - p.ca : c;
#pragma unroll
            for (int k = 0; k < 8; ++k) { s[k] = 0.f; q[k] = 0.f; }
            for (int r = r0; r < r1; ++r) {
                const u32x4 v = *reinterpret_cast<const u32x4*>(base + ((size_t)n * p.DHW + r) * cs + cc);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float lo = __uint_as_float(v[k] << 16), hi = __uint_as_float(v[k] & 0xffff0000u);
                    s[2 * k] += lo; q[2 * k] += lo * lo; s[2 * k + 1] += hi; q[2 * k + 1] += hi * hi;
                }
            }
            float* dst = p.partial + (((size_t)n * p.nslab + slab) * C + c) * 2;
#pragma unroll
            for (int k = 0; k < 8; ++k) { dst[2 * k] = s[k]; dst[2 * k + 1] = q[k]; }
        }
    }
}

// Pass 2: one block per (sample, group): fold slabs and the group's channels in fp64, emit per-channel
// scale/shift  a_c = gamma_c * rstd_g,  b_c = beta_c - mean_g * a_c   ->  ab[N][C][2].
struct GnFinalizeParams {
    const float* partial; int nslab; int C; int Creal; int groups; int DHW; float eps;
    const float* gamma; const float* beta; float* ab;
    float* mr;                                     // optional [N][groups][2] mean, rstd (saved for the backward pass)
};

__global__ __launch_bounds__(256) void gn_finalize_kernel(const GnFinalizeParams p) {   // fallback path and the fp32 precision mode
    __shared__ double red[2][4];
    const int n = blockIdx.y, g = blockIdx.x;
    const int cpg = p.Creal / p.groups;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nthr = blockDim.x;
    double s = 0.0, q = 0.0;
    const int items = p.nslab * cpg;
    for (int i = tid; i < items; i += nthr) {              // fixed assignment and fixed fold order below: reproducible
        const int slab = i / cpg, c = g * cpg + (i - slab * cpg);
        const float* src = p.partial + (((size_t)n * p.nslab + slab) * p.C + c) * 2;
        s += (double)src[0]; q += (double)src[1];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { s += __shfl_xor(s, o, 64); q += __shfl_xor(q, o, 64); }
    if (lane == 0) { red[0][wave] = s; red[1][wave] = q; }
    __syncthreads();
    s = 0.0; q = 0.0;
    for (int w = 0; w < (nthr + 63) / 64; ++w) { s += red[0][w]; q += red[1][w]; }
    const double cnt = (double)cpg * (double)p.DHW;
    const double mean = s / cnt;
    double var = q / cnt - mean * mean; if (var < 0.0) var = 0.0;
    const float rstd = (float)(1.0 / sqrt(var + (double)p.eps));
    if (p.mr && tid == 0) { p.mr[((size_t)n * p.groups + g) * 2] = (float)mean; p.mr[((size_t)n * p.groups + g) * 2 + 1] = rstd; }
    for (int c = g * cpg + tid; c < (g + 1) * cpg; c += nthr) {
        const float a = p.gamma[c] * rstd;
        p.ab[((size_t)n * p.C + c) * 2] = a;
        p.ab[((size_t)n * p.C + c) * 2 + 1] = p.beta[c] - (float)mean * a;
    }
}

// Pass 2': the same result from the per-32-row partials the producing kernels wrote (conv / split-K finalize
// epilogues): slab [rowblock][C][2] per source tensor.  grid = (groups, N), 256 threads.
struct GnPrepParams {
    const float* sa; const float* sb; int ca, cb;      // slabs of the two concatenated sources (sb may be null)
    int nrb_a, nrb_b;                                  // statistics blocks per sample of each source's slab
    int groups; int DHW; float eps;
    const float* gamma; const float* beta; float* ab;
    float* mr;                                     // optional [N][groups][2] mean, rstd (saved for the backward pass)
};

__global__ __launch_bounds__(256) void gn_prep_kernel(const GnPrepParams p) {
    __shared__ double rs[256], rq[256];
    const int n = blockIdx.y, g = blockIdx.x, tid = threadIdx.x;
    const int C = p.ca + p.cb;
    const int cpg = C / p.groups;
    const int c_lo = g * cpg, c_hi = c_lo + cpg;
    double s = 0.0, q = 0.0;
    // one source's share of the group: nc channels from channel c0 of a slab with cs channels per row and nrb rows per sample.
    // thread = (channel, row lane); eight slab rows in flight per thread (the fold of 7000 rows is otherwise one L2 round
    // trip per row and thread: 50 us at 96^3), fixed order -> reproducible.
    auto fold = [&](const float* slab, int cs, int nrb, int c0, int nc) {
        if (nc <= 0) return;
        if (256 % nc == 0) {
            const int cc = tid % nc, rstep = 256 / nc;
            const float* base = slab + ((size_t)n * nrb * cs + c0 + cc) * 2;
            int rbl = tid / nc;
            for (; rbl + 7 * rstep < nrb; rbl += 8 * rstep) {
                float2 v[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] = *reinterpret_cast<const float2*>(base + (size_t)(rbl + k * rstep) * cs * 2);
#pragma unroll
                for (int k = 0; k < 8; ++k) { s += (double)v[k].x; q += (double)v[k].y; }
            }
            for (; rbl < nrb; rbl += rstep) {
                const float2 v = *reinterpret_cast<const float2*>(base + (size_t)rbl * cs * 2);
                s += (double)v.x; q += (double)v.y;
            }
        } else {
            for (int i = tid; i < nrb * nc; i += 256) {
                const int rbl = i / nc, c = c0 + (i - rbl * nc);
                const float2 v = *reinterpret_cast<const float2*>(slab + ((size_t)(n * nrb + rbl) * cs + c) * 2);
                s += (double)v.x; q += (double)v.y;
            }
        }
    };
    {   // channels of the group that live in source a, then those in source b (a group may straddle the concat boundary)
        const int a_hi = c_hi < p.ca ? c_hi : p.ca;
        fold(p.sa, p.ca, p.nrb_a, c_lo, a_hi - c_lo);
        const int b_lo = c_lo > p.ca ? c_lo : p.ca;
        if (p.sb) fold(p.sb, p.cb, p.nrb_b, b_lo - p.ca, c_hi - b_lo);
    }
    rs[tid] = s; rq[tid] = q;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (tid < o) { rs[tid] += rs[tid + o]; rq[tid] += rq[tid + o]; }
        __syncthreads();
    }
    const double cnt = (double)cpg * (double)p.DHW;
    const double mean = rs[0] / cnt;
    double var = rq[0] / cnt - mean * mean; if (var < 0.0) var = 0.0;
    const float rstd = (float)(1.0 / sqrt(var + (double)p.eps));
    if (p.mr && tid == 0) { p.mr[((size_t)n * p.groups + g) * 2] = (float)mean; p.mr[((size_t)n * p.groups + g) * 2 + 1] = rstd; }
    for (int c = g * cpg + tid; c < (g + 1) * cpg; c += 256) {
        const float a = p.gamma[c] * rstd;
        p.ab[((size_t)n * C + c) * 2] = a;
        p.ab[((size_t)n * C + c) * 2 + 1] = p.beta[c] - (float)mean * a;
    }
}

// Pass 3: y = silu?(x * a_c + b_c) -> contiguous bf16 NDHWC (materialises the concat for the next conv).
struct GnApplyParams {
    const bf16_t* xa; const bf16_t* xb; int ca, cb; int DHW; int N; int silu;
    const float* ab; bf16_t* out;
};

__global__ __launch_bounds__(256) void gn_apply_kernel(const GnApplyParams p) {
    const int C = p.ca + p.cb;
    const int cvec = C / 8;
    const long total = (long)p.N * p.DHW * cvec;
    const long stride = (long)gridDim.x * blockDim.x;
    // four 16-byte vectors in flight per thread (clamped, unconditional loads): at 96^3 the pass moves 226 MB and one load per
    // iteration left it at 3.6 TB/s
    for (long i0 = (long)blockIdx.x * blockDim.x + threadIdx.x; i0 < total; i0 += 4 * stride) {
        u32x4 v[4]; long row[4]; int c[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            long i = i0 + u * stride; if (i >= total) i = total - 1;
            row[u] = i / cvec;
            c[u] = (int)(i - row[u] * cvec) * 8;
            const bool second = c[u] >= p.ca;
            const bf16_t* src = second ? p.xb + row[u] * p.cb + (c[u] - p.ca) : p.xa + row[u] * p.ca + c[u];
            v[u] = *reinterpret_cast<const u32x4*>(src);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (i0 + u * stride >= total) break;
            const int n = (int)(row[u] / p.DHW);
            const float4* abp = reinterpret_cast<const float4*>(p.ab + ((size_t)n * C + c[u]) * 2);
            float y[8];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float4 ab = abp[k];
                float lo = __uint_as_float(v[u][k] << 16) * ab.x + ab.y;
                float hi = __uint_as_float(v[u][k] & 0xffff0000u) * ab.z + ab.w;
                if (p.silu == 1) { lo = silu_f(lo); hi = silu_f(hi); }
                else if (p.silu == 2) { lo = lo > 0.f ? lo : 0.2f * lo; hi = hi > 0.f ? hi : 0.2f * hi; }
                y[2 * k] = lo; y[2 * k + 1] = hi;
            }
            u32x4 o;
#pragma unroll
            for (int k = 0; k < 4; ++k) o[k] = pack2bf(y[2 * k], y[2 * k + 1]);
            *reinterpret_cast<u32x4*>(p.out + row[u] * C + c[u]) = o;
        }
    }
}

// One-launch GroupNorm(+SiLU) for tensors whose producers left few slab rows (one per conv tile): every block first
// folds the slabs of the channels it needs (its 64-channel slice widened to whole groups), derives mean / rstd of those
// groups, then normalises its rows.  Replaces gn_prep + gn_apply (two launch floors) wherever slab rows <= 256 per
// sample; the redundant slab reads stay in L2.  grid = (row chunks, ceil(C / 64), N), 256 threads.
struct GnFusedParams {
    const bf16_t* xa; const bf16_t* xb; int ca, cb;
    const float* sa; const float* sb; int nrb_a, nrb_b;      // slabs [N * nrb][c][2] of the two sources
    int groups, DHW, N, silu, rows_per_block; float eps;
    const float* gamma; const float* beta; bf16_t* out;
    float* ab; float* mr;                              // optional (training): [N][C][2] scale/shift, [N][G][2] mean/rstd
    int xcd_rows;                                      // block order: 1 = every XCD gets a contiguous range of row chunks (all channel slices)
};

// Channel fold shared by the GroupNorm kernels that read per-row-block partial slabs [N * nrb][c][2] of two channel-concatenated sources:
// leaves the totals of channel cov_lo + k (k < ncov <= 192) in csum[k][0..1].  Fixed summation order (reproducible).
struct GnFoldSrc { const float* sa; const float* sb; int ca, cb, nrb_a, nrb_b; };
__device__ __forceinline__ void gn_fold_cover(const GnFoldSrc p, const int n, const int cpg, const int cov_lo, const int ncov,
                                              float (*part)[96][4], double (*csum)[2]) {
    const int tid = threadIdx.x;
    if (((cpg | p.ca) & 1) == 0) {
        // thread = (channel pair of the cover, one of 8 slab lanes); 16 slab rows in flight per thread
        const int bl = tid >> 5;
        for (int pp = tid & 31; pp < (ncov >> 1); pp += 32) {
            const int cch = cov_lo + 2 * pp;
            const bool second = cch >= p.ca;
            const float* sl = second ? p.sb : p.sa;
            const int cs = second ? p.cb : p.ca, cl = second ? cch - p.ca : cch, nrb = second ? p.nrb_b : p.nrb_a;
            const float* base = sl + ((size_t)n * nrb * cs + cl) * 2;
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int b0 = bl; b0 < nrb; b0 += 128) {
                float4 t[16];
#pragma unroll
                for (int k = 0; k < 16; ++k) {
                    int bb = b0 + 8 * k; const bool ok = bb < nrb; if (!ok) bb = nrb - 1;
                    t[k] = *reinterpret_cast<const float4*>(base + (size_t)bb * cs * 2);
                    if (!ok) t[k] = make_float4(0.f, 0.f, 0.f, 0.f);
                }
#pragma unroll
                for (int k = 0; k < 16; ++k) { acc.x += t[k].x; acc.y += t[k].y; acc.z += t[k].z; acc.w += t[k].w; }
            }
            *reinterpret_cast<float4*>(&part[bl][pp][0]) = acc;
        }
        __syncthreads();
        if (tid < ncov) {
            const int pp = tid >> 1, o = (tid & 1) * 2;
            double s = 0.0, q = 0.0;
#pragma unroll
            for (int k = 0; k < 8; ++k) { s += (double)part[k][pp][o]; q += (double)part[k][pp][o + 1]; }
            csum[tid][0] = s; csum[tid][1] = q;
        }
    } else {
        // odd channels per group: thread = (channel of the cover, one of 4 slab lanes)
        const int bl = tid >> 6;
        float* part1 = &part[0][0][0];                 // viewed as [4][192][2]
        for (int cc = tid & 63; cc < ncov; cc += 64) {
            const int cch = cov_lo + cc;
            const bool second = cch >= p.ca;
            const float* sl = second ? p.sb : p.sa;
            const int cs = second ? p.cb : p.ca, cl = second ? cch - p.ca : cch, nrb = second ? p.nrb_b : p.nrb_a;
            float s0 = 0.f, s1 = 0.f;
            const float* base = sl + ((size_t)n * nrb * cs + cl) * 2;
            for (int bb = bl; bb < nrb; bb += 4) {
                const float2 t = *reinterpret_cast<const float2*>(base + (size_t)bb * cs * 2);
                s0 += t.x; s1 += t.y;
            }
            part1[(bl * 192 + cc) * 2] = s0; part1[(bl * 192 + cc) * 2 + 1] = s1;
        }
        __syncthreads();
        if (tid < ncov) {
            double s = 0.0, q = 0.0;
#pragma unroll
            for (int k = 0; k < 4; ++k) { s += (double)part1[(k * 192 + tid) * 2]; q += (double)part1[(k * 192 + tid) * 2 + 1]; }
            csum[tid][0] = s; csum[tid][1] = q;
        }
    }
}

// Slab fold of gn_fused_apply_kernel: sums the partial (sum, sum of squares) rows of the channels that cover the block's
// 64-channel slice `by` of sample n and leaves mean / rstd of those groups in gstat[g - g_lo].
__device__ __forceinline__ void gn_fused_fold(const GnFusedParams& p, const int n, const int by, const bool first_block,
                                              float (*part)[96][4], double (*csum)[2], float (*gstat)[2]) {
    const int tid = threadIdx.x;
    const int C = p.ca + p.cb, cpg = C / p.groups;
    const int c0 = by * 64;
    int c1 = c0 + 64; if (c1 > C) c1 = C;
    const int g_lo = c0 / cpg, g_hi = (c1 + cpg - 1) / cpg;
    const int cov_lo = g_lo * cpg, ncov = g_hi * cpg - cov_lo;           // <= 64 + 2 * 63
    gn_fold_cover(GnFoldSrc{p.sa, p.sb, p.ca, p.cb, p.nrb_a, p.nrb_b}, n, cpg, cov_lo, ncov, part, csum);
    __syncthreads();
    if (tid < g_hi - g_lo) {
        double s = 0.0, q = 0.0;
        for (int k = 0; k < cpg; ++k) { s += csum[tid * cpg + k][0]; q += csum[tid * cpg + k][1]; }
        const double cnt = (double)cpg * (double)p.DHW;
        const double mean = s / cnt;
        double var = q / cnt - mean * mean; if (var < 0.0) var = 0.0;
        const float rstd = (float)(1.0 / sqrt(var + (double)p.eps));
        gstat[tid][0] = (float)mean; gstat[tid][1] = rstd;
        if (p.mr && first_block) {
            p.mr[((size_t)n * p.groups + g_lo + tid) * 2] = (float)mean; p.mr[((size_t)n * p.groups + g_lo + tid) * 2 + 1] = rstd;
        }
    }
    __syncthreads();
}

template <bool WT>
__global__ __launch_bounds__(256) void gn_fused_apply_kernel(const GnFusedParams p) {
    __shared__ __attribute__((aligned(16))) float part[8][96][4];
    __shared__ double csum[192][2];
    __shared__ float gstat[64][2];
    KSTAMP_BEGIN(3);
    const int tid = threadIdx.x;
    // Block order.  xcd_rows: blocks b, b + 8, ... share an XCD (observed round-robin dispatch; speed only, never correctness), so the
    // bijective remap of the convolutions hands every XCD one contiguous range of (sample, row chunk) units with all their channel
    // slices: the rows a conv's workgroups left in that XCD's L2 (ConvParams::tile_order 1) are normalised by blocks of the same XCD,
    // and the conv that follows finds most of its voxel rows there too.  A hand-off between XCDs runs at about a third of that rate.
    int bx = blockIdx.x, by = blockIdx.y, n = blockIdx.z;
    if (p.xcd_rows) {
        const int per_n = gridDim.x * gridDim.y;
        int lid = xcd_remap((int)(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z)), per_n * (int)gridDim.z);
        n = lid / per_n; lid -= n * per_n;
        bx = lid / (int)gridDim.y; by = lid - bx * (int)gridDim.y;
    }
    const int C = p.ca + p.cb, cpg = C / p.groups;     // host guarantees cpg <= 64
    const int c0 = by * 64;
    const int g_lo = c0 / cpg;
    // ---- the block's first rows are requested before the fold, so that their latency overlaps it
    //      apply mapping: thread = (8-channel vector of the slice, row lane)
    constexpr int PF = 8;
    const int vec = tid & 7, rl = tid >> 3;
    const int c = c0 + vec * 8;
    const bool active = c < C;
    const bool second_x = c >= p.ca;
    const bf16_t* src = second_x ? p.xb : p.xa;
    const int xcs = second_x ? p.cb : p.ca, xcl = second_x ? c - p.ca : c;
    const int r0 = bx * p.rows_per_block;
    int r1 = r0 + p.rows_per_block; if (r1 > p.DHW) r1 = p.DHW;
    u32x4 v[PF];
    float4 gam[2] = {make_float4(0.f, 0.f, 0.f, 0.f), make_float4(0.f, 0.f, 0.f, 0.f)}, bet[2] = {gam[0], gam[0]};
    if (active) {
#pragma unroll
        for (int k = 0; k < PF; ++k) {
            int r = r0 + rl + 32 * k; if (r >= r1) r = r1 - 1;          // clamped: unconditional loads stay in flight together
            v[k] = *reinterpret_cast<const u32x4*>(src + ((size_t)n * p.DHW + r) * xcs + xcl);
        }
        gam[0] = *reinterpret_cast<const float4*>(p.gamma + c); gam[1] = *reinterpret_cast<const float4*>(p.gamma + c + 4);
        bet[0] = *reinterpret_cast<const float4*>(p.beta + c); bet[1] = *reinterpret_cast<const float4*>(p.beta + c + 4);
    }
    KSTAMP(1);
    gn_fused_fold(p, n, by, bx == 0, part, csum, gstat);
    KSTAMP(2);
    // ---- apply
    if (!active) return;
    float a[8], b[8];
    const float gk[8] = {gam[0].x, gam[0].y, gam[0].z, gam[0].w, gam[1].x, gam[1].y, gam[1].z, gam[1].w};
    const float bk[8] = {bet[0].x, bet[0].y, bet[0].z, bet[0].w, bet[1].x, bet[1].y, bet[1].z, bet[1].w};
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int g = (c + k) / cpg - g_lo;
        a[k] = gk[k] * gstat[g][1];
        b[k] = bk[k] - gstat[g][0] * a[k];
    }
    if (p.ab && bx == 0 && rl == 0) {
#pragma unroll
        for (int k = 0; k < 8; ++k) { p.ab[((size_t)n * C + c + k) * 2] = a[k]; p.ab[((size_t)n * C + c + k) * 2 + 1] = b[k]; }
    }
    for (int rb = r0 + rl; rb < r1; rb += 32 * PF) {
        if (rb != r0 + rl) {
#pragma unroll
            for (int k = 0; k < PF; ++k) {
                int r = rb + 32 * k; if (r >= r1) r = r1 - 1;
                v[k] = *reinterpret_cast<const u32x4*>(src + ((size_t)n * p.DHW + r) * xcs + xcl);
            }
        }
#pragma unroll
        for (int k = 0; k < PF; ++k) {
            const int r = rb + 32 * k;
            if (r < r1) {
                u32x4 o;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float lo = __uint_as_float(v[k][j] << 16) * a[2 * j] + b[2 * j];
                    float hi = __uint_as_float(v[k][j] & 0xffff0000u) * a[2 * j + 1] + b[2 * j + 1];
                    if (p.silu == 1) { lo = silu_f(lo); hi = silu_f(hi); }
                    else if (p.silu == 2) { lo = lo > 0.f ? lo : 0.2f * lo; hi = hi > 0.f ? hi : 0.2f * hi; }
                    o[j] = pack2bf(lo, hi);
                }
                store16<WT>(p.out + ((size_t)n * p.DHW + r) * C + c, o);
            }
        }
    }
    KSTAMP(3);
    KSTAMP_DRAIN(4);
}

// ------------------------------------------------------------------------------------------------
// fp32 NCDHW (x | cond channel-concatenated) -> bf16 NDHWC with zero channel padding to Cs.  Once per forward.
__global__ __launch_bounds__(256) void pack2_ncdhw_kernel(const float* __restrict__ x, int cx, const float* __restrict__ cond,
                                                          int cc, bf16_t* __restrict__ out, int N, int Cs, int DHW) {
    const long total = (long)N * DHW * Cs;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % Cs);
        const long row = i / Cs;
        const int n = (int)(row / DHW);
        const int sp = (int)(row - (long)n * DHW);
        float v = 0.f;
        if (c < cx) v = x[((size_t)n * cx + c) * DHW + sp];
        else if (c < cx + cc) v = cond[((size_t)n * cc + (c - cx)) * DHW + sp];
        out[i] = f2bf(v);
    }
}

// ------------------------------------------------------------------------------------------------
// Timestep embedding + MLP:  emb = W2 silu(W1 [cos(t f), sin(t f)] + b1) + b2   (cos first).
// GEMV with bf16 weights [out][in], fp32 activations; one wave per output row.
__global__ __launch_bounds__(256) void temb_sinusoid_kernel(const float* __restrict__ t, float* __restrict__ out,
                                                            int B, int dim) {
    KSTAMP_BEGIN(1);
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int half = dim / 2;
    if (i >= B * dim) return;
    const int b = i / dim, k = i - b * dim;
    const int j = k < half ? k : k - half;
    const float f = expf((-logf(10000.0f) * (float)j) / (float)half);     // same op order as get_timestep_embedding
    const float a = t[b] * f;
    KSTAMP(1);
    out[i] = (k < half) ? cosf(a) : ((k < 2 * half) ? sinf(a) : 0.f);
    KSTAMP_DRAIN(2);
}

// y[b][o] = sum_i W[o][i] * act(x[b][i]) + bias[o];  act = SiLU when silu_in.  grid = (ceil(O/4), B).
__global__ __launch_bounds__(256) void gemv_bf16_kernel(const bf16_t* __restrict__ W, const float* __restrict__ bias,
                                                        const float* __restrict__ x, float* __restrict__ y,
                                                        int I, int O, int x_stride, int y_stride, int silu_in) {
    KSTAMP_BEGIN(2);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int o = blockIdx.x * 4 + wave, b = blockIdx.y;
    if (o >= O) return;
    const bf16_t* wr = W + (size_t)o * I;
    const float* xr = x + (size_t)b * x_stride;
    float acc = 0.f;
    for (int i = lane * 8; i < I; i += 64 * 8) {
        const u32x4 w = *reinterpret_cast<const u32x4*>(wr + i);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float x0 = xr[i + 2 * k], x1 = xr[i + 2 * k + 1];
            if (silu_in) { x0 = silu_f(x0); x1 = silu_f(x1); }
            acc += __uint_as_float(w[k] << 16) * x0 + __uint_as_float(w[k] & 0xffff0000u) * x1;
        }
    }
    acc = wave_sum(acc);
    KSTAMP(1);
    if (lane == 0) y[(size_t)b * y_stride + o] = acc + (bias ? bias[o] : 0.f);
    KSTAMP_DRAIN(2);
}

// ------------------------------------------------------------------------------------------------
// Scheduler steps (fp32, NCDHW flat).  Coefficients are computed on the host in fp32 exactly as MONAI does
//  and passed by value, so the device side is a pure fused multiply-add pass.
struct StepCoef { float inv_sqrt_a, sqrt_b, c0, c1, sigma, dir; int clip; };

__global__ __launch_bounds__(256) void ddpm_step_kernel(const float* __restrict__ eps, const float* __restrict__ x,
                                                        const float* __restrict__ z, float* __restrict__ prev,
                                                        float* __restrict__ x0_out, long n, StepCoef k) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        float x0 = (x[i] - k.sqrt_b * eps[i]) * k.inv_sqrt_a;
        if (k.clip) x0 = fminf(fmaxf(x0, -1.f), 1.f);
        float pv = k.c0 * x0 + k.c1 * x[i];
        if (z) pv += k.sigma * z[i];
        prev[i] = pv;
        if (x0_out) x0_out[i] = x0;
    }
}

__global__ __launch_bounds__(256) void ddim_step_kernel(const float* __restrict__ eps, const float* __restrict__ x,
                                                        const float* __restrict__ z, float* __restrict__ prev,
                                                        float* __restrict__ x0_out, long n, StepCoef k) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        float x0 = (x[i] - k.sqrt_b * eps[i]) * k.inv_sqrt_a;
        if (k.clip) x0 = fminf(fmaxf(x0, -1.f), 1.f);
        float pv = k.c0 * x0 + k.dir * eps[i];          // c0 = sqrt(abar_prev), dir = sqrt(1 - abar_prev - sigma^2)
        if (z) pv += k.sigma * z[i];
        prev[i] = pv;
        if (x0_out) x0_out[i] = x0;
    }
}

// noisy = sa[b] * x0 + sb[b] * eps, per-sample coefficients read from device arrays.
__global__ __launch_bounds__(256) void add_noise_kernel(const float* __restrict__ x0, const float* __restrict__ eps,
                                                        const float* __restrict__ sa, const float* __restrict__ sb,
                                                        float* __restrict__ out, long per_sample, int B) {
    const long n = per_sample * B;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int b = (int)(i / per_sample);
        out[i] = sa[b] * x0[i] + sb[b] * eps[i];
    }
}

// VAE heads: input fp32 NCDHW [N][2L][DHW] (mu | log_var) -> z_mu, z_sigma (clamp [-30, 20], exp(./2)) and
// optionally z = mu + sigma * eps.
__global__ __launch_bounds__(256) void vae_heads_kernel(const float* __restrict__ ml, const float* __restrict__ eps,
                                                        float* __restrict__ mu, float* __restrict__ sigma,
                                                        float* __restrict__ z, int N, int L, int DHW) {
    const long total = (long)N * L * DHW;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int n = (int)(i / ((long)L * DHW));
        const long r = i - (long)n * L * DHW;
        const float m = ml[(size_t)n * 2 * L * DHW + r];
        float lv = ml[(size_t)n * 2 * L * DHW + (long)L * DHW + r];
        lv = fminf(fmaxf(lv, -30.f), 20.f);
        const float sg = expf(0.5f * lv);
        if (mu) mu[i] = m;
        if (sigma) sigma[i] = sg;
        if (z) z[i] = m + sg * (eps ? eps[i] : 0.f);
    }
}

// ------------------------------------------------------------------------------------------------
// Device-resident sampler (ldm_sampler_*): the scheduler step with its noise drawn INSIDE the kernel (Philox4x32-10, counter =
// (element quad, step index), key = seed) and its per-step coefficients read from a device table indexed by a device step counter,
// so that one denoising step (UNet forward + this kernel) is a fixed sequence of launches with fixed arguments: one HIP graph.
// Replaces torch.randn + fill_ + the host-side coefficient lookup of DDPMScheduler.step / DDIMScheduler.step
// (3d_ldm/inference.py:94-99 loop body).  coef row: {1/sqrt(abar_t), sqrt(1-abar_t), c0, c1 (DDPM) | dir (DDIM), sigma, t, 0, 0}.
struct SamplerState { int k; unsigned done; };
struct SamplerParams {
    const float* coef; SamplerState* st; int n_steps; int kind;      // kind 0 = DDPM, 1 = DDIM
    int clip; unsigned seed_lo, seed_hi;
    const float* eps; float* x; float* x0_out; long n;               // x is updated in place
    float* tbuf; int B;                                              // the UNet's timestep input: receives t of the NEXT step
};
__device__ __forceinline__ void philox4x32_10(unsigned c0, unsigned c1, unsigned c2, unsigned c3, unsigned k0, unsigned k1, unsigned out[4]) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const unsigned hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        const unsigned hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        const unsigned n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
// four independent N(0, 1) draws for element quad `quad` of step `step` (Box-Muller on two pairs of uniforms in (0, 1])
__device__ __forceinline__ float4 sampler_normal4(unsigned long long quad, unsigned step, unsigned seed_lo, unsigned seed_hi) {
    unsigned r[4];
    philox4x32_10((unsigned)quad, (unsigned)(quad >> 32), step, 0x5eedu, seed_lo, seed_hi, r);
    const float u0 = ((float)(r[0] >> 8) + 1.0f) * (1.0f / 16777216.0f), u1 = (float)(r[1] >> 8) * (1.0f / 16777216.0f);
    const float u2 = ((float)(r[2] >> 8) + 1.0f) * (1.0f / 16777216.0f), u3 = (float)(r[3] >> 8) * (1.0f / 16777216.0f);
    const float ra = sqrtf(-2.0f * logf(u0)), rb = sqrtf(-2.0f * logf(u2));
    float sa, ca, sb, cb;
    sincosf(6.283185307179586f * u1, &sa, &ca); sincosf(6.283185307179586f * u3, &sb, &cb);
    return make_float4(ra * ca, ra * sa, rb * cb, rb * sb);
}
__global__ __launch_bounds__(256) void sampler_noise_kernel(float* __restrict__ out, long n, int step, unsigned seed_lo, unsigned seed_hi) {
    const long nq = (n + 3) / 4;
    for (long q = (long)blockIdx.x * blockDim.x + threadIdx.x; q < nq; q += (long)gridDim.x * blockDim.x) {
        const float4 z = sampler_normal4((unsigned long long)q, (unsigned)step, seed_lo, seed_hi);
        const float zz[4] = {z.x, z.y, z.z, z.w};
        for (int e = 0; e < 4; ++e) if (4 * q + e < n) out[4 * q + e] = zz[e];
    }
}
__global__ __launch_bounds__(256) void sampler_step_kernel(const SamplerParams p) {
    KSTAMP_BEGIN(9);
    __shared__ int s_last;
    const int k = p.st->k;                                  // every block reads the counter before it can bump `done`
    const bool live = k < p.n_steps;
    const float* c = p.coef + (size_t)(live ? k : p.n_steps - 1) * 8;
    const float inv_sqrt_a = c[0], sqrt_b = c[1], c0 = c[2], c1 = c[3], sigma = c[4];
    const long nq = (p.n + 3) / 4;
    if (live)
    for (long q = (long)blockIdx.x * blockDim.x + threadIdx.x; q < nq; q += (long)gridDim.x * blockDim.x) {
        float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
        if (sigma != 0.f) z = sampler_normal4((unsigned long long)q, (unsigned)k, p.seed_lo, p.seed_hi);
        const float zz[4] = {z.x, z.y, z.z, z.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const long i = 4 * q + e;
            if (i >= p.n) break;
            const float xe = p.x[i], ee = p.eps[i];
            float x0 = (xe - sqrt_b * ee) * inv_sqrt_a;
            if (p.clip) x0 = fminf(fmaxf(x0, -1.f), 1.f);
            float pv = (p.kind == 0) ? c0 * x0 + c1 * xe : c0 * x0 + c1 * ee;       // DDIM: c1 = sqrt(1 - abar_prev - sigma^2)
            if (sigma != 0.f) pv += sigma * zz[e];
            p.x[i] = pv;
            if (p.x0_out) p.x0_out[i] = x0;
        }
    }
    // the block that finishes last advances the step counter and publishes the next timestep (all blocks have read k by then)
    __syncthreads();
    if (threadIdx.x == 0) s_last = (atomicAdd(&p.st->done, 1u) == gridDim.x - 1) ? 1 : 0;
    __syncthreads();
    if (s_last && threadIdx.x == 0) {
        p.st->done = 0;
        const int kn = live ? k + 1 : k;
        p.st->k = kn;
        const float tn = p.coef[(size_t)(kn < p.n_steps ? kn : p.n_steps - 1) * 8 + 5];
        for (int b = 0; b < p.B; ++b) p.tbuf[b] = tn;
    }
}
__global__ void sampler_reset_kernel(SamplerState* st, const float* coef, float* tbuf, int B) {
    if (threadIdx.x == 0 && blockIdx.x == 0) { st->k = 0; st->done = 0; for (int b = 0; b < B; ++b) tbuf[b] = coef[5]; }
}
// Tabulated time embedding ("depends only on t => tabulate"): the 17 stacked time_emb_proj outputs of every
// timestep of a sampler's schedule are computed once per parameter upload by the plan's own sinusoid + GEMV kernels (bit-identical
// rows); a denoising step then copies row k (k = the sampler's device-resident step counter) instead of running those four launches.
// out[b][0..rows) = tab[min(k, n_steps - 1)][0..rows), rows % 4 == 0, out rows `stride` floats apart.  grid = (ceil(rows / 1024), B).
__global__ __launch_bounds__(256) void temb_row_kernel(const float* __restrict__ tab, const SamplerState* __restrict__ st,
                                                       float* __restrict__ out, int rows, int stride, int n_steps) {
    KSTAMP_BEGIN(1);
    int k = st->k; if (k >= n_steps) k = n_steps - 1; if (k < 0) k = 0;
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    KSTAMP(1);
    if (4 * q < rows)
        *reinterpret_cast<float4*>(out + (size_t)blockIdx.y * stride + 4 * q) = *reinterpret_cast<const float4*>(tab + (size_t)k * rows + 4 * q);
    KSTAMP_DRAIN(2);
}

// ------------------------------------------------------------------------------------------------
// PatchDiscriminator support (stage-1 GAN tail, 3d_ldm/train_autoencoder.py:150-158,407-424,454-494): its 4^3 convolutions run as
// im2col + the 1x1 GEMM kernels.  col[m][tap * C + c] = x[voxel(m, tap)][c] (zero where the tap falls into the padding; columns
// beyond taps * C up to Kp are zero), taps ordered (kd, kh, kw); x NDHWC bf16 with Cs stored channels of which C are used.
// T = bf16_t (raw bits) or float: the fp32 forms serve the discriminator under --precision fp32 (ldm_op_*_f32)
template <class T> __device__ __forceinline__ float el_ld(const T* p);
template <> __device__ __forceinline__ float el_ld<bf16_t>(const bf16_t* p) { return bf2f(*p); }
template <> __device__ __forceinline__ float el_ld<float>(const float* p) { return *p; }
template <class T> __device__ __forceinline__ void el_st(T* p, float v);
template <> __device__ __forceinline__ void el_st<bf16_t>(bf16_t* p, float v) { *p = f2bf(v); }
template <> __device__ __forceinline__ void el_st<float>(float* p, float v) { *p = v; }
template <class T>
__global__ __launch_bounds__(256) void im2col_generic_kernel(const T* __restrict__ x, T* __restrict__ col, int N, int D, int H, int W,
                                                             int Cs, int C, int k, int stride, int pad, int Do, int Ho, int Wo, int Kp) {
    const long total = (long)N * Do * Ho * Wo * Kp;
    const int taps = k * k * k;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int kk = (int)(i % Kp);
        const long m = i / Kp;
        T v = 0;
        if (kk < taps * C) {
            const int tap = kk / C, c = kk - tap * C;
            const int kd = tap / (k * k), kh = (tap / k) % k, kw = tap % k;
            const int ow = (int)(m % Wo); long r = m / Wo; const int oh = (int)(r % Ho); r /= Ho; const int od = (int)(r % Do); const int n = (int)(r / Do);
            const int id = od * stride + kd - pad, ih = oh * stride + kh - pad, iw = ow * stride + kw - pad;
            if ((unsigned)id < (unsigned)D && (unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W)
                v = x[((((size_t)n * D + id) * H + ih) * W + iw) * Cs + c];
        }
        col[i] = v;
    }
}
// adjoint: dx[voxel][c] = sum over the (output position, tap) pairs that read that voxel of dcol[m][tap * C + c] (gather form: no
// atomics, fixed summation order); channels C..Cs of dx are written as zeros.
template <class T>
__global__ __launch_bounds__(256) void col2im_generic_kernel(const T* __restrict__ dcol, T* __restrict__ dx, int N, int D, int H, int W,
                                                             int Cs, int C, int k, int stride, int pad, int Do, int Ho, int Wo, int Kp) {
    const long total = (long)N * D * H * W * Cs;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % Cs);
        long r = i / Cs;
        const int iw = (int)(r % W); r /= W; const int ih = (int)(r % H); r /= H; const int id = (int)(r % D); const int n = (int)(r / D);
        float s = 0.f;
        if (c < C)
            for (int kd = 0; kd < k; ++kd) {
                const int td = id + pad - kd; if (td < 0 || td % stride) continue; const int od = td / stride; if (od >= Do) continue;
                for (int kh = 0; kh < k; ++kh) {
                    const int th = ih + pad - kh; if (th < 0 || th % stride) continue; const int oh = th / stride; if (oh >= Ho) continue;
                    for (int kw = 0; kw < k; ++kw) {
                        const int tw = iw + pad - kw; if (tw < 0 || tw % stride) continue; const int ow = tw / stride; if (ow >= Wo) continue;
                        const size_t m = (((size_t)n * Do + od) * Ho + oh) * Wo + ow;
                        s += el_ld<T>(dcol + m * Kp + ((kd * k + kh) * k + kw) * C + c);
                    }
                }
            }
        el_st<T>(dx + i, s);
    }
}
template <class T>
__global__ __launch_bounds__(256) void leaky_relu_kernel(const T* __restrict__ x, T* __restrict__ y, long n, float slope) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float v = el_ld<T>(x + i);
        el_st<T>(y + i, v > 0.f ? v : slope * v);
    }
}
template <class T>
__global__ __launch_bounds__(256) void leaky_relu_bwd_kernel(const T* __restrict__ x, const T* __restrict__ dy, T* __restrict__ dx,
                                                             long n, float slope) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        el_st<T>(dx + i, el_ld<T>(x + i) > 0.f ? el_ld<T>(dy + i) : slope * el_ld<T>(dy + i));
}
// bf16 NDHWC (Cs stored channels) -> fp32 NCDHW (first C channels): the op-level counterpart of pack2_ncdhw_kernel
template <class T>
__global__ __launch_bounds__(256) void unpack_ndhwc_kernel(const T* __restrict__ act, float* __restrict__ out, int N, int C, int Cs, int DHW) {
    const long total = (long)N * C * DHW;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int sp = (int)(i % DHW);
        const long r = i / DHW;
        const int c = (int)(r % C), n = (int)(r / C);
        out[i] = el_ld<T>(act + ((size_t)n * DHW + sp) * Cs + c);
    }
}

// ------------------------------------------------------------------------------------------------
// Data path: MONAI's ScaleIntensityRangePercentiles(lower, upper, b_min, b_max) of the reference's loader
// (3d_ldm/utils.py:94-107) on the device.  The two percentiles need exact order statistics of each volume: a three-pass radix
// select (12 + 12 + 8 bits) on the order-preserving integer image of the fp32 values, for up to four ranks at once (floor / ceil
// index of both percentiles), integer histograms in LDS + global integer atomics (exact and order independent), then one apply pass.
struct PctState { unsigned prefix[4]; unsigned rank[4]; float value[4]; float t[2]; int n_ranks; int pad_; };
__device__ __forceinline__ unsigned pct_key(float f) { const unsigned u = __float_as_uint(f); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); }
__device__ __forceinline__ float pct_unkey(unsigned k) { return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k); }
// pass 0: bits 31..20, pass 1: bits 19..8 (among keys whose top 12 bits equal the rank's prefix), pass 2: bits 7..0
__global__ __launch_bounds__(256) void pct_hist_kernel(const float* __restrict__ x, long n, const PctState* __restrict__ st, unsigned* __restrict__ hist, int pass) {
    __shared__ unsigned h[4 * 4096];
    const int b = blockIdx.y;
    const PctState s = st[b];
    const int R = pass == 0 ? 1 : s.n_ranks;             // before any prefix is known all ranks share one histogram
    const int bins = pass == 2 ? 256 : 4096;
    for (int i = threadIdx.x; i < R * bins; i += 256) h[i] = 0;
    __syncthreads();
    const float* xb = x + (size_t)b * n;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const unsigned k = pct_key(xb[i]);
        if (pass == 0) atomicAdd(&h[k >> 20], 1u);
        else
            for (int r = 0; r < R; ++r) {
                if (pass == 1) { if ((k >> 20) == s.prefix[r]) atomicAdd(&h[r * 4096 + ((k >> 8) & 4095u)], 1u); }
                else if ((k >> 8) == s.prefix[r]) atomicAdd(&h[r * 256 + (k & 255u)], 1u);
            }
    }
    __syncthreads();
    unsigned* hb = hist + (size_t)b * 4 * 4096;
    for (int i = threadIdx.x; i < R * bins; i += 256) { const unsigned v = h[i]; if (v) atomicAdd(&hb[(i / bins) * 4096 + (i % bins)], v); }
}
// one block per volume: locate each rank's bin, extend its prefix, reduce its rank to the rank inside the bin; zero the histograms
__global__ __launch_bounds__(256) void pct_scan_kernel(PctState* __restrict__ st, unsigned* __restrict__ hist, int pass) {
    __shared__ unsigned part[256];
    __shared__ unsigned found_bin, found_before;
    const int b = blockIdx.x, tid = threadIdx.x;
    PctState* s = st + b;
    unsigned* hb = hist + (size_t)b * 4 * 4096;
    const int bins = pass == 2 ? 256 : 4096, per = bins / 256;
    const int R = s->n_ranks;
    for (int r = 0; r < R; ++r) {
        const unsigned* hr = hb + (pass == 0 ? 0 : r) * 4096;
        unsigned sum = 0;
        for (int i = 0; i < per; ++i) sum += hr[tid * per + i];
        part[tid] = sum;
        __syncthreads();
        if (tid == 0) {
            const unsigned rank = s->rank[r];
            unsigned cum = 0; int t = 0;
            while (t < 255 && cum + part[t] <= rank) { cum += part[t]; ++t; }
            int bin = t * per;
            while (bin < t * per + per - 1 && cum + hr[bin] <= rank) { cum += hr[bin]; ++bin; }
            found_bin = (unsigned)bin; found_before = cum;
        }
        __syncthreads();
        if (tid == 0) {
            s->prefix[r] = pass == 0 ? found_bin : ((s->prefix[r] << (pass == 2 ? 8 : 12)) | found_bin);
            s->rank[r] -= found_before;
            if (pass == 2) s->value[r] = pct_unkey(s->prefix[r]);
        }
        __syncthreads();
    }
    for (int i = tid; i < 4 * 4096; i += 256) hb[i] = 0;
}
__global__ __launch_bounds__(256) void pct_apply_kernel(const float* __restrict__ x, float* __restrict__ out, long n, const PctState* __restrict__ st,
                                                        float b_min, float b_max) {
    const int b = blockIdx.y;
    const PctState s = st[b];
    // value[0..1] = floor / ceil order statistics of the lower percentile, value[2..3] of the upper one (n_ranks == 4 always)
    const float a_min = (float)((double)s.value[0] + ((double)s.value[1] - (double)s.value[0]) * (double)s.t[0]);
    const float a_max = (float)((double)s.value[2] + ((double)s.value[3] - (double)s.value[2]) * (double)s.t[1]);
    const float d = a_max - a_min;
    const float* xb = x + (size_t)b * n; float* ob = out + (size_t)b * n;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256)
        ob[i] = d == 0.f ? b_min : (xb[i] - a_min) / d * (b_max - b_min) + b_min;      // MONAI ScaleIntensityRange: constant image -> b_min
}
__global__ void pct_init_kernel(PctState* st, int B, long n, double lower, double upper) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    PctState s{};
    const double pl = (double)(n - 1) * lower / 100.0, pu = (double)(n - 1) * upper / 100.0;
    const long l0 = (long)floor(pl), u0 = (long)floor(pu);
    s.rank[0] = (unsigned)l0; s.rank[1] = (unsigned)(l0 + 1 < n ? l0 + 1 : n - 1);
    s.rank[2] = (unsigned)u0; s.rank[3] = (unsigned)(u0 + 1 < n ? u0 + 1 : n - 1);
    s.t[0] = (float)(pl - (double)l0); s.t[1] = (float)(pu - (double)u0);
    s.n_ranks = 4;
    st[b] = s;
}

__global__ __launch_bounds__(256) void scale_kernel(const float* __restrict__ x, float* __restrict__ y, long n, float s) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) y[i] = x[i] * s;
}

// ------------------------------------------------------------------------------------------------
// Training support.
// Weights for the data-gradient convolution: Wt[tap'][ci][co] = W[taps-1-tap'][co][ci]  (flip + transpose), both in the
// arena layout [tap][rows padded][cols]; zero rows / columns in the padding.
__global__ __launch_bounds__(256) void weight_flip_transpose_kernel(const bf16_t* __restrict__ w, bf16_t* __restrict__ wt,
                                                                    int taps, int cout, int cout_pad, int cin, int cin_pad_rows,
                                                                    int ci_off, int ci_cnt) {
    // w: [taps][cout_pad][cin];  wt: [taps][cin_pad_rows][round32(cout)] holding input channels [ci_off, ci_off + ci_cnt)
    // (one matrix per source tensor of a channel-concatenated conv input).  64 x 64 tiles through LDS so that both the
    // reads (ci contiguous) and the writes (co contiguous) are coalesced.  grid = (col tiles, row tiles, taps).
    __shared__ bf16_t tile[64][66];
    const int cols = (cout + 31) / 32 * 32;
    const int tp = blockIdx.z, co0 = blockIdx.x * 64, ci0 = blockIdx.y * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const bf16_t* src = w + (size_t)(taps - 1 - tp) * cout_pad * cin;
#pragma unroll 4
    for (int r = ty; r < 64; r += 4) {                   // r = cout row of the tile, tx = cin column
        const int co = co0 + r, ci = ci0 + tx;
        tile[r][tx] = (co < cout && ci < ci_cnt) ? src[(size_t)co * cin + ci_off + ci] : (bf16_t)0;
    }
    __syncthreads();
    bf16_t* dst = wt + (size_t)tp * cin_pad_rows * cols;
#pragma unroll 4
    for (int r = ty; r < 64; r += 4) {                   // r = cin row of the output, tx = cout column
        const int ci = ci0 + r, co = co0 + tx;
        if (ci < cin_pad_rows && co < cols) dst[(size_t)ci * cols + co] = tile[tx][r];
    }
}

// ------------------------------------------------------------------------------------------------
// GroupNorm (+SiLU) backward.  Forward: y = act(u), u = gamma_c * xhat + beta_c, xhat = (x - mean_g) * rstd_g.
// With g = dy * act'(u):  dgamma_c = sum g*xhat,  dbeta_c = sum g,
//   dx = rstd_g * (gamma_c * g - mean_grp(gamma*g) - xhat * mean_grp(gamma*g*xhat))   (means over the group's cpg*DHW elements).
// Pass 1: per-(sample, channel) partial sums of g and g*xhat over voxel slabs (same structure as gn_stats_kernel).
struct GnBwdParams {
    const bf16_t* dy;                              // [N*DHW][C] gradient w.r.t. the GroupNorm(+SiLU) output
    const bf16_t* xa; const bf16_t* xb; int ca, cb; // saved forward input (channel-concatenated sources)
    const float* ab;                               // [N][C][2] forward scale/shift (u = a*x + b)
    const float* mr;                               // [N][G][2] forward mean, rstd
    const float* gamma;
    int groups, DHW, N, silu, nslab, rows_per_slab;
    float* partial;                                // [N][nslab][C][2]
    float* gsum;                                   // [N][G][2]  mean_grp(gamma*g), mean_grp(gamma*g*xhat)
    float* dgamma_n; float* dbeta_n;               // [N][C] per-sample parameter gradients (summed over N by the caller)
    const bf16_t* acc_a; const bf16_t* acc_b;      // optional gradients to add to dx (residual / skip paths), same split
    bf16_t* dxa; bf16_t* dxb;                      // outputs [N*DHW][ca], [N*DHW][cb]
    int rows_per_block;                            // gn_bwd_fold_apply_kernel: rows per block of the apply grid (a multiple of 32)
    float* cs;                                     // optional: column sums of dx per row chunk, [N][chunks][ca][2] then [N][chunks][cb][2] (pair = (sum, 0)):
                                                   // the bias / time-embedding gradient of the conv that produced x (Builder::emit_colsum)
};

// activation code of the GroupNorm kernels (fields named `silu`): 0 = none, 1 = SiLU, 2 = LeakyReLU(0.2) (PatchDiscriminator)
__device__ __forceinline__ float gn_act(float u, int act) { return act == 1 ? silu_f(u) : (act == 2 ? (u > 0.f ? u : 0.2f * u) : u); }
__device__ __forceinline__ float gn_bwd_g(float dy, float u, int silu) {
    if (!silu) return dy;
    if (silu == 2) return u > 0.f ? dy : 0.2f * dy;
    const float sg = 1.0f / (1.0f + __expf(-u));
    return dy * sg * (1.0f + u * (1.0f - sg));
}

__global__ __launch_bounds__(256) void gn_bwd_stats_kernel(const GnBwdParams p) {
    __shared__ float red[256 * 16];
    const int C = p.ca + p.cb, cvec = C / 8, cpg = C / p.groups;
    const int n = blockIdx.y, slab = blockIdx.x, tid = threadIdx.x;
    const int r0 = slab * p.rows_per_slab;
    int r1 = r0 + p.rows_per_slab; if (r1 > p.DHW) r1 = p.DHW;
    const int rows_par = 256 / cvec > 0 ? 256 / cvec : 1;
    for (int cv0 = 0; cv0 < cvec; cv0 += 256) {
        const int cw = cvec - cv0 < 256 ? cvec - cv0 : 256;
        const int rp = 256 / cw;
        const int cv = tid % cw, rl = tid / cw;
        float s1[8], s2[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) { s1[k] = 0.f; s2[k] = 0.f; }
        if (rl < rp) {
            const int c = (cv0 + cv) * 8;
            const bool second = c >= p.ca;
            const bf16_t* xb_ = second ? p.xb : p.xa;
            const int cs = second ? p.cb : p.ca, cc = second ? c - p.ca : c;
            float a[8], b[8], mean[8], rstd[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                a[k] = p.ab[((size_t)n * C + c + k) * 2]; b[k] = p.ab[((size_t)n * C + c + k) * 2 + 1];
                const int g = (c + k) / cpg;
                mean[k] = p.mr[((size_t)n * p.groups + g) * 2]; rstd[k] = p.mr[((size_t)n * p.groups + g) * 2 + 1];
            }
            for (int r = r0 + rl; r < r1; r += rp) {
                const size_t row = (size_t)n * p.DHW + r;
                const u32x4 xv = *reinterpret_cast<const u32x4*>(xb_ + row * cs + cc);
                const u32x4 dv = *reinterpret_cast<const u32x4*>(p.dy + row * C + c);
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const float x = __uint_as_float((k & 1) ? (xv[k >> 1] & 0xffff0000u) : (xv[k >> 1] << 16));
                    const float dy = __uint_as_float((k & 1) ? (dv[k >> 1] & 0xffff0000u) : (dv[k >> 1] << 16));
                    const float g = gn_bwd_g(dy, a[k] * x + b[k], p.silu);
                    s1[k] += g; s2[k] += g * (x - mean[k]) * rstd[k];
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 8; ++k) { red[tid * 16 + k] = s1[k]; red[tid * 16 + 8 + k] = s2[k]; }
        __syncthreads();
        if (tid < cw) {
            for (int r2 = 1; r2 < rp; ++r2) {
                const int t2 = r2 * cw + tid;
#pragma unroll
                for (int k = 0; k < 8; ++k) { s1[k] += red[t2 * 16 + k]; s2[k] += red[t2 * 16 + 8 + k]; }
            }
            float* dst = p.partial + (((size_t)n * p.nslab + slab) * C + (cv0 + tid) * 8) * 2;
#pragma unroll
            for (int k = 0; k < 8; ++k) { dst[2 * k] = s1[k]; dst[2 * k + 1] = s2[k]; }
        }
    }
    (void)rows_par;
}

// Pass 2: one block per (sample, group): per-channel totals -> dgamma/dbeta (per sample), group means of gamma*g(.xhat).
// 256 threads = (channel of the group) x (slab lanes) when channels-per-group divides 256, else channel by channel.
__global__ __launch_bounds__(256) void gn_bwd_finalize_kernel(const GnBwdParams p) {
    __shared__ float r1[256], r2[256];
    const int n = blockIdx.y, g = blockIdx.x, tid = threadIdx.x;
    const int C = p.ca + p.cb, cpg = C / p.groups;
    float S1 = 0.f, S2 = 0.f;                            // valid on tid 0 after the loop
    if (cpg <= 256 && 256 % cpg == 0) {
        const int cl = tid % cpg, sl = tid / cpg, lanes = 256 / cpg, c = g * cpg + cl;
        float t1 = 0.f, t2 = 0.f;
        for (int s = sl; s < p.nslab; s += lanes) {
            const float2 v = *reinterpret_cast<const float2*>(p.partial + (((size_t)n * p.nslab + s) * C + c) * 2);
            t1 += v.x; t2 += v.y;
        }
        r1[tid] = t1; r2[tid] = t2;
        __syncthreads();
        float a1 = 0.f, a2 = 0.f;
        if (tid < cpg) {
            for (int k = 0; k < lanes; ++k) { a1 += r1[k * cpg + tid]; a2 += r2[k * cpg + tid]; }
            p.dbeta_n[(size_t)n * C + c] = a1;
            p.dgamma_n[(size_t)n * C + c] = a2;
            a1 *= p.gamma[c]; a2 *= p.gamma[c];
        }
        __syncthreads();
        r1[tid] = a1; r2[tid] = a2;
        __syncthreads();
        if (tid == 0) for (int k = 0; k < cpg; ++k) { S1 += r1[k]; S2 += r2[k]; }
    } else {
        for (int c = g * cpg; c < (g + 1) * cpg; ++c) {
            float t1 = 0.f, t2 = 0.f;
            for (int s = tid; s < p.nslab; s += 256) {
                const float* src = p.partial + (((size_t)n * p.nslab + s) * C + c) * 2;
                t1 += src[0]; t2 += src[1];
            }
            r1[tid] = t1; r2[tid] = t2;
            __syncthreads();
            for (int o = 128; o > 0; o >>= 1) { if (tid < o) { r1[tid] += r1[tid + o]; r2[tid] += r2[tid + o]; } __syncthreads(); }
            if (tid == 0) {
                p.dbeta_n[(size_t)n * C + c] = r1[0]; p.dgamma_n[(size_t)n * C + c] = r2[0];
                S1 += p.gamma[c] * r1[0]; S2 += p.gamma[c] * r2[0];
            }
            __syncthreads();
        }
    }
    if (tid == 0) {
        const float cnt = (float)cpg * (float)p.DHW;
        p.gsum[((size_t)n * p.groups + g) * 2] = S1 / cnt;
        p.gsum[((size_t)n * p.groups + g) * 2 + 1] = S2 / cnt;
    }
}

// Pass 3: dx (+ accumulated residual gradient) -> bf16, split back into the two concatenated sources.
__global__ __launch_bounds__(256) void gn_bwd_apply_kernel(const GnBwdParams p) {
    const int C = p.ca + p.cb, cvec = C / 8, cpg = C / p.groups;
    const long total = (long)p.N * p.DHW * cvec;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long row = i / cvec;
        const int c = (int)(i - row * cvec) * 8;
        const int n = (int)(row / p.DHW);
        const bool second = c >= p.ca;
        const int cs = second ? p.cb : p.ca, cc = second ? c - p.ca : c;
        const u32x4 xv = *reinterpret_cast<const u32x4*>((second ? p.xb : p.xa) + row * cs + cc);
        const u32x4 dv = *reinterpret_cast<const u32x4*>(p.dy + row * C + c);
        const bf16_t* accp = second ? p.acc_b : p.acc_a;
        u32x4 av = {0u, 0u, 0u, 0u};
        if (accp) av = *reinterpret_cast<const u32x4*>(accp + row * cs + cc);
        float out[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const float x = __uint_as_float((k & 1) ? (xv[k >> 1] & 0xffff0000u) : (xv[k >> 1] << 16));
            const float dy = __uint_as_float((k & 1) ? (dv[k >> 1] & 0xffff0000u) : (dv[k >> 1] << 16));
            const float ac = __uint_as_float((k & 1) ? (av[k >> 1] & 0xffff0000u) : (av[k >> 1] << 16));
            const float a = p.ab[((size_t)n * C + c + k) * 2], b = p.ab[((size_t)n * C + c + k) * 2 + 1];
            const int g = (c + k) / cpg;
            const float mean = p.mr[((size_t)n * p.groups + g) * 2], rstd = p.mr[((size_t)n * p.groups + g) * 2 + 1];
            const float m1 = p.gsum[((size_t)n * p.groups + g) * 2], m2 = p.gsum[((size_t)n * p.groups + g) * 2 + 1];
            const float gg = gn_bwd_g(dy, a * x + b, p.silu);
            const float xh = (x - mean) * rstd;
            out[k] = rstd * (p.gamma[c + k] * gg - m1 - xh * m2) + ac;
        }
        u32x4 o;
#pragma unroll
        for (int k = 0; k < 4; ++k) o[k] = pack2bf(out[2 * k], out[2 * k + 1]);
        *reinterpret_cast<u32x4*>((second ? p.dxb : p.dxa) + row * cs + cc) = o;
    }
}

// Passes 2 + 3 in one launch (training plans, bf16): every block folds the partial rows of the channels that cover its 64-channel slice
// (gn_fold_cover, as the forward's one-launch GroupNorm does), derives the two group means, and applies to its row chunk; the blocks
// of row chunk 0 write dgamma / dbeta of their slice.  Replaces gn_bwd_finalize + gn_bwd_apply wherever channels per group <= 64 and
// the partial rows per sample are few (<= 512): one launch floor less per GroupNorm.  grid = (row chunks, ceil(C / 64), N), 256 threads.
__global__ __launch_bounds__(256) void gn_bwd_fold_apply_kernel(const GnBwdParams p) {
    __shared__ __attribute__((aligned(16))) float part[8][96][4];
    __shared__ double csum[192][2];
    __shared__ float gstat[64][2];
    const int tid = threadIdx.x, n = blockIdx.z;
    const int C = p.ca + p.cb, cpg = C / p.groups;
    const int c0 = blockIdx.y * 64;
    int c1 = c0 + 64; if (c1 > C) c1 = C;
    const int g_lo = c0 / cpg, g_hi = (c1 + cpg - 1) / cpg;
    const int cov_lo = g_lo * cpg, ncov = g_hi * cpg - cov_lo;
    gn_fold_cover(GnFoldSrc{p.partial, nullptr, C, 0, p.nslab, 0}, n, cpg, cov_lo, ncov, part, csum);
    __syncthreads();
    if (blockIdx.x == 0 && tid < c1 - c0) {
        p.dbeta_n[(size_t)n * C + c0 + tid] = (float)csum[c0 - cov_lo + tid][0];
        p.dgamma_n[(size_t)n * C + c0 + tid] = (float)csum[c0 - cov_lo + tid][1];
    }
    if (tid < g_hi - g_lo) {
        double s1 = 0.0, s2 = 0.0;
        for (int k = 0; k < cpg; ++k) {
            const double gm = (double)p.gamma[cov_lo + tid * cpg + k];
            s1 += gm * csum[tid * cpg + k][0]; s2 += gm * csum[tid * cpg + k][1];
        }
        const double cnt = (double)cpg * (double)p.DHW;
        gstat[tid][0] = (float)(s1 / cnt); gstat[tid][1] = (float)(s2 / cnt);
    }
    __syncthreads();
    const int vec = tid & 7, rl = tid >> 3;
    const int c = c0 + vec * 8;
    const bool active = c < C;
    const bool second = c >= p.ca;
    const int cs = second ? p.cb : p.ca, cc = second ? c - p.ca : c;
    const bf16_t* xs = second ? p.xb : p.xa;
    const bf16_t* accp = second ? p.acc_b : p.acc_a;
    bf16_t* dxs = second ? p.dxb : p.dxa;
    float colsum[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) colsum[k] = 0.f;
    if (active) {
        float a[8], b[8], mean[8], rstd[8], m1[8], m2[8], gam[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            a[k] = p.ab[((size_t)n * C + c + k) * 2]; b[k] = p.ab[((size_t)n * C + c + k) * 2 + 1];
            const int g = (c + k) / cpg;
            mean[k] = p.mr[((size_t)n * p.groups + g) * 2]; rstd[k] = p.mr[((size_t)n * p.groups + g) * 2 + 1];
            m1[k] = gstat[g - g_lo][0]; m2[k] = gstat[g - g_lo][1];
            gam[k] = p.gamma[c + k];
        }
        const int r0 = blockIdx.x * p.rows_per_block;          // row chunk of this block (host: a multiple of 32 rows)
        int r1 = r0 + p.rows_per_block; if (r1 > p.DHW) r1 = p.DHW;
        for (int r = r0 + rl; r < r1; r += 32) {
            const size_t row = (size_t)n * p.DHW + r;
            const u32x4 xv = *reinterpret_cast<const u32x4*>(xs + row * cs + cc);
            const u32x4 dv = *reinterpret_cast<const u32x4*>(p.dy + row * C + c);
            u32x4 av = {0u, 0u, 0u, 0u};
            if (accp) av = *reinterpret_cast<const u32x4*>(accp + row * cs + cc);
            float out[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const float x = __uint_as_float((k & 1) ? (xv[k >> 1] & 0xffff0000u) : (xv[k >> 1] << 16));
                const float dy = __uint_as_float((k & 1) ? (dv[k >> 1] & 0xffff0000u) : (dv[k >> 1] << 16));
                const float ac = __uint_as_float((k & 1) ? (av[k >> 1] & 0xffff0000u) : (av[k >> 1] << 16));
                const float gg = gn_bwd_g(dy, a[k] * x + b[k], p.silu);
                const float xh = (x - mean[k]) * rstd[k];
                out[k] = rstd[k] * (gam[k] * gg - m1[k] - xh * m2[k]) + ac;
            }
            u32x4 o;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                o[k] = pack2bf(out[2 * k], out[2 * k + 1]);
                colsum[2 * k] += __uint_as_float(o[k] << 16); colsum[2 * k + 1] += __uint_as_float(o[k] & 0xffff0000u);   // of the STORED values
            }
            *reinterpret_cast<u32x4*>(dxs + row * cs + cc) = o;
        }
    }
    if (p.cs == nullptr) return;
    // column sums of this block's rows: 32 row lanes through LDS (fixed order), one (sum, 0) pair per channel
    float* red = &part[0][0][0];                           // 32 x 64 floats (the fold is done with it)
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 8; ++k) red[rl * 64 + vec * 8 + k] = colsum[k];
    __syncthreads();
    if (tid < 64 && c0 + tid < C) {
        float t = 0.f;
#pragma unroll 8
        for (int k = 0; k < 32; ++k) t += red[k * 64 + tid];
        const int ch = c0 + tid;
        const bool sec = ch >= p.ca;
        const int chunks = gridDim.x;
        float* dst = sec ? p.cs + (size_t)p.N * chunks * p.ca * 2 + (((size_t)n * chunks + blockIdx.x) * p.cb + (ch - p.ca)) * 2
                         : p.cs + (((size_t)n * chunks + blockIdx.x) * p.ca + ch) * 2;
        dst[0] = t; dst[1] = 0.f;
    }
}

// ------------------------------------------------------------------------------------------------
// Small training kernels.
// MSE loss gradient: d/dpred mean((pred - target)^2) = 2 (pred - target) / numel, fp32 NCDHW -> bf16 NDHWC (padded);
// also accumulates the loss (one atomicAdd per block; the value is only reported, never fed back).
__global__ __launch_bounds__(256) void mse_grad_pack_kernel(const float* __restrict__ pred, const float* __restrict__ target,
                                                            bf16_t* __restrict__ dy, float* __restrict__ loss,
                                                            int N, int C, int Cs, int DHW) {
    const long total = (long)N * DHW * Cs;
    const float inv = 1.0f / (float)((long)N * C * DHW);
    float lsum = 0.f;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % Cs);
        const long row = i / Cs;
        const int n = (int)(row / DHW);
        const int sp = (int)(row - (long)n * DHW);
        float g = 0.f;
        if (c < C) {
            const size_t j = ((size_t)n * C + c) * DHW + sp;
            const float d = pred[j] - target[j];
            g = 2.0f * d * inv; lsum += d * d * inv;
        }
        dy[i] = f2bf(g);
    }
    lsum = wave_sum(lsum);
    if (loss && (threadIdx.x & 63) == 0) atomicAdd(loss, lsum);
}

// Per-(sample, channel) column sums of an NDHWC bf16 tensor from gn_stats-style partials [N][nslab][C][2]:
//   accumulate_over_n: out[c] = sum_n sum_slab (bias gradient);  else out[n * out_stride + c] = sum_slab (time-embedding
//   gradient).  Only the first `count` channels are written.
__global__ __launch_bounds__(256) void colsum_finalize_kernel(const float* __restrict__ partial, float* __restrict__ out,
                                                              int N, int nslab, int C, int accumulate_over_n, int count, int out_stride) {
    // grid = (ceil(count / 16), accumulate_over_n ? 1 : N); block = 16 channels x 16 slab lanes
    __shared__ float red[256];
    const int cl = threadIdx.x & 15, sl = threadIdx.x >> 4;
    const int c = blockIdx.x * 16 + cl;
    const int n0 = accumulate_over_n ? 0 : blockIdx.y, n1 = accumulate_over_n ? N : blockIdx.y + 1;
    float t = 0.f;
    if (c < count)
        for (int n = n0; n < n1; ++n)
            for (int s = sl; s < nslab; s += 16) t += partial[(((size_t)n * nslab + s) * C + c) * 2];
    red[threadIdx.x] = t;
    __syncthreads();
    if (sl == 0 && c < count) {
        float a = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) a += red[k * 16 + cl];
        out[accumulate_over_n ? (size_t)c : (size_t)blockIdx.y * out_stride + c] = a;
    }
}

// All column-sum finalizes of a backward plan in ONE launch (79 bias gradients + 17 time-embedding rows for the benchmark UNet: each is
// a launch at its floor otherwise).  blockmap[b] = (descriptor, block inside it); block layout as colsum_finalize_kernel.
struct ColsumDesc { long partial_off; long out_off; int N, nslab, C, accumulate_over_n, count, out_stride, gx, pad_; };
__global__ __launch_bounds__(256) void colsum_finalize_batched_kernel(const ColsumDesc* __restrict__ descs, const int2* __restrict__ blockmap,
                                                                      char* __restrict__ ws) {
    __shared__ float red[256];
    const int2 bm = blockmap[blockIdx.x];
    const ColsumDesc e = descs[bm.x];
    const float* partial = reinterpret_cast<const float*>(ws + e.partial_off);
    float* out = reinterpret_cast<float*>(ws + e.out_off);
    const int bx = bm.y % e.gx, by = bm.y / e.gx;
    const int cl = threadIdx.x & 15, sl = threadIdx.x >> 4;
    const int c = bx * 16 + cl;
    const int n0 = e.accumulate_over_n ? 0 : by, n1 = e.accumulate_over_n ? e.N : by + 1;
    float t = 0.f;
    if (c < e.count)
        for (int n = n0; n < n1; ++n)
            for (int s = sl; s < e.nslab; s += 16) t += partial[(((size_t)n * e.nslab + s) * e.C + c) * 2];
    red[threadIdx.x] = t;
    __syncthreads();
    if (sl == 0 && c < e.count) {
        float a = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) a += red[k * 16 + cl];
        out[e.accumulate_over_n ? (size_t)c : (size_t)by * e.out_stride + c] = a;
    }
}

// 2x2x2 sum pooling (adjoint of the nearest x2 upsample): out[n][d][h][w][c] = sum of the 8 fine voxels, bf16 NDHWC.
__global__ __launch_bounds__(256) void sumpool2_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ out,
                                                       int N, int D, int H, int W, int C) {   // D,H,W = COARSE dims
    const int cvec = C / 8;
    const long total = (long)N * D * H * W * cvec;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int cv = (int)(i % cvec);
        long r = i / cvec;
        const int w = (int)(r % W); r /= W; const int h = (int)(r % H); r /= H; const int d = (int)(r % D); const int n = (int)(r / D);
        float s[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) s[k] = 0.f;
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const size_t row = (((size_t)n * 2 * D + 2 * d + (t >> 2)) * 2 * H + 2 * h + ((t >> 1) & 1)) * 2 * W + 2 * w + (t & 1);
            const u32x4 v = *reinterpret_cast<const u32x4*>(x + row * C + cv * 8);
#pragma unroll
            for (int k = 0; k < 4; ++k) { s[2 * k] += __uint_as_float(v[k] << 16); s[2 * k + 1] += __uint_as_float(v[k] & 0xffff0000u); }
        }
        u32x4 o;
#pragma unroll
        for (int k = 0; k < 4; ++k) o[k] = pack2bf(s[2 * k], s[2 * k + 1]);
        *reinterpret_cast<u32x4*>(out + (size_t)i * 8) = o;
    }
}

// Linear layers of the time-embedding path (tiny: batch x <= 1024 features), fp32 activations, bf16-rounded weights.
//   dx[b][i] = act'(x_pre[b][i]) * sum_o W[o][i] dy[b][o]      (act = SiLU applied to the layer INPUT when silu_in)
//   dW[o][i] = sum_b dy[b][o] * act(x_pre[b][i]);  db[o] = sum_b dy[b][o]
__global__ __launch_bounds__(256) void linear_bwd_dx_kernel(const bf16_t* __restrict__ W, const float* __restrict__ dy,
                                                            const float* __restrict__ x_pre, float* __restrict__ dx,
                                                            int I, int O, int dy_stride, int x_stride, int silu_in) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x, b = blockIdx.y;
    if (i >= I) return;
    float acc = 0.f;
    for (int o = 0; o < O; ++o) acc += bf2f(W[(size_t)o * I + i]) * dy[(size_t)b * dy_stride + o];
    if (silu_in) {
        const float u = x_pre[(size_t)b * x_stride + i];
        const float sg = 1.0f / (1.0f + __expf(-u));
        acc *= sg * (1.0f + u * (1.0f - sg));
    }
    dx[(size_t)b * x_stride + i] = acc;
}

// The same in two stages for tall matrices (the stacked time_emb_proj: O ~ 6400): stage 1 sums a slice of the output rows
// (blockIdx.z) into part[z][b][i]; stage 2 folds the slices in a fixed order and applies act'.
__global__ __launch_bounds__(256) void linear_bwd_dx_part_kernel(const bf16_t* __restrict__ W, const float* __restrict__ dy,
                                                                 float* __restrict__ part, int I, int O, int dy_stride, int B) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x, b = blockIdx.y, z = blockIdx.z, nz = gridDim.z;
    if (i >= I) return;
    const int per = (O + nz - 1) / nz, o0 = z * per;
    int o1 = o0 + per; if (o1 > O) o1 = O;
    float acc = 0.f;
#pragma unroll 4
    for (int o = o0; o < o1; ++o) acc += bf2f(W[(size_t)o * I + i]) * dy[(size_t)b * dy_stride + o];
    part[((size_t)z * B + b) * I + i] = acc;
}
__global__ __launch_bounds__(256) void linear_bwd_dx_fold_kernel(const float* __restrict__ part, const float* __restrict__ x_pre,
                                                                 float* __restrict__ dx, int I, int nz, int x_stride, int B, int silu_in) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x, b = blockIdx.y;
    if (i >= I) return;
    float acc = 0.f;
    for (int z = 0; z < nz; ++z) acc += part[((size_t)z * B + b) * I + i];
    if (silu_in) {
        const float u = x_pre[(size_t)b * x_stride + i];
        const float sg = 1.0f / (1.0f + __expf(-u));
        acc *= sg * (1.0f + u * (1.0f - sg));
    }
    dx[(size_t)b * x_stride + i] = acc;
}

__global__ __launch_bounds__(256) void linear_bwd_dw_kernel(const float* __restrict__ dy, const float* __restrict__ x_pre,
                                                            float* __restrict__ dW, float* __restrict__ db,
                                                            int B, int I, int O, int dy_stride, int x_stride, int silu_in) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)O * I) return;
    const int o = (int)(idx / I), i = (int)(idx - (long)o * I);
    float acc = 0.f, bsum = 0.f;
    for (int b = 0; b < B; ++b) {
        float xv = x_pre[(size_t)b * x_stride + i];
        if (silu_in) xv = silu_f(xv);
        const float d = dy[(size_t)b * dy_stride + o];
        acc += d * xv; bsum += d;
    }
    dW[idx] = acc;
    if (i == 0 && db) db[o] = bsum;
}

// out[c] = sum_n in[n][c]
__global__ __launch_bounds__(256) void rowsum_n_kernel(const float* __restrict__ in, float* __restrict__ out, int N, int C) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    float t = 0.f;
    for (int n = 0; n < N; ++n) t += in[(size_t)n * C + c];
    out[c] = t;
}

// out = a + b (bf16 NDHWC gradients meeting at a tensor with two consumers), 8 elements per thread.
__global__ __launch_bounds__(256) void add_bf16_kernel(const bf16_t* __restrict__ a, const bf16_t* __restrict__ b,
                                                       bf16_t* __restrict__ out, long nvec) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (long)gridDim.x * blockDim.x) {
        const u32x4 va = reinterpret_cast<const u32x4*>(a)[i], vb = reinterpret_cast<const u32x4*>(b)[i];
        u32x4 o;
#pragma unroll
        for (int k = 0; k < 4; ++k)
            o[k] = pack2bf(__uint_as_float(va[k] << 16) + __uint_as_float(vb[k] << 16),
                           __uint_as_float(va[k] & 0xffff0000u) + __uint_as_float(vb[k] & 0xffff0000u));
        reinterpret_cast<u32x4*>(out)[i] = o;
    }
}

// ------------------------------------------------------------------------------------------------
// Parameter traffic between the caller's fp32 tensors (MONAI state_dict layout) and the library's device layouts.
// Pack:   src [cout][cin][taps] fp32  ->  dst [taps][cout_pad][cin_s] bf16, rows row_off .. row_off + cout, zero channel padding.
// Export: src [taps][rows_total][ld] fp32 (wgrad output), rows row_off.., columns col_off..  ->  dst [cout][cin][taps] fp32.
// One block per (64-channel chunk, cout row): the [64][taps] tile goes through LDS so both sides are coalesced.
__global__ __launch_bounds__(256) void param_pack_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst,
                                                         int taps, int cout, int cin, int cin_s, int cout_pad, int row_off) {
    __shared__ float tile[64 * 27];
    const int co = blockIdx.y, ci0 = blockIdx.x * 64, tid = threadIdx.x;
    int nci = cin - ci0; if (nci > 64) nci = 64; if (nci < 0) nci = 0;
    const float* s = src + ((size_t)co * cin + ci0) * taps;
    for (int i = tid; i < nci * taps; i += 256) tile[i] = s[i];
    __syncthreads();
    for (int i = tid; i < taps * 64; i += 256) {
        const int t = i >> 6, c = i & 63, ci = ci0 + c;
        if (ci < cin_s) dst[((size_t)t * cout_pad + row_off + co) * cin_s + ci] = (c < nci) ? f2bf(tile[c * taps + t]) : (bf16_t)0;
    }
}
__global__ __launch_bounds__(256) void grad_export_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                                          int taps, int rows_total, int ld, int row_off, int col_off, int cout, int cin,
                                                          int nsplit, long slab_stride) {
    // src may hold `nsplit` partial matrices (the weight-gradient kernel splits the voxel range): folded here, in order
    __shared__ float tile[64 * 27];
    const int co = blockIdx.y, ci0 = blockIdx.x * 64, tid = threadIdx.x;
    int nci = cin - ci0; if (nci > 64) nci = 64;
    for (int i = tid; i < taps * 64; i += 256) {
        const int t = i >> 6, c = i & 63;
        if (c < nci) {
            const float* sp = src + ((size_t)t * rows_total + row_off + co) * ld + col_off + ci0 + c;
            float v = sp[0];
            for (int k = 1; k < nsplit; ++k) v += sp[(size_t)k * slab_stride];
            tile[c * taps + t] = v;
        }
    }
    __syncthreads();
    float* d = dst + ((size_t)co * cin + ci0) * taps;
    for (int i = tid; i < nci * taps; i += 256) d[i] = tile[i];
}

// ------------------------------------------------------------------------------------------------
// Optimizer tail on flat fp32 buffers: global gradient norm and Adam with the clip
// factor folded in.  Both HBM-bound, 16-byte accesses.  The norm is a two-stage deterministic reduction.
__global__ __launch_bounds__(256) void sq_norm_part_kernel(const float* __restrict__ g, long n, float* __restrict__ part) {
    __shared__ float red[4];
    float acc = 0.f;
    const long nv = n / 4;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += (long)gridDim.x * blockDim.x) {
        const float4 v = reinterpret_cast<const float4*>(g)[i];
        acc += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
    }
    if (blockIdx.x == 0 && threadIdx.x < (int)(n - nv * 4)) { const float v = g[nv * 4 + threadIdx.x]; acc += v * v; }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}
__global__ __launch_bounds__(256) void sq_norm_fold_kernel(const float* __restrict__ part, int nparts, float* __restrict__ out) {
    __shared__ double red[256];
    double acc = 0.0;
    for (int i = threadIdx.x; i < nparts; i += 256) acc += (double)part[i];
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
    if (threadIdx.x == 0) *out = (float)red[0];
}
// bf16 wire format of the bucketed gradient exchange (ldm_model_set_grad_wire): fp32 slice -> bf16 staging -> (all-reduce) -> fp32 slice.
// The slice starts at an arbitrary element of the flat buffer, so the fp32 side is only 4-byte aligned: scalar accesses, 8 per thread.
__global__ __launch_bounds__(256) void grad_wire_pack_kernel(const float* __restrict__ g, bf16_t* __restrict__ stage, long n) {
    for (long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 8; i < n; i += (long)gridDim.x * blockDim.x * 8) {
        if (i + 8 <= n) {
            u32x4 v;
            v[0] = pack2bf(g[i], g[i + 1]); v[1] = pack2bf(g[i + 2], g[i + 3]); v[2] = pack2bf(g[i + 4], g[i + 5]); v[3] = pack2bf(g[i + 6], g[i + 7]);
            *reinterpret_cast<u32x4*>(stage + i) = v;
        } else
            for (long k = i; k < n; ++k) stage[k] = f2bf(g[k]);
    }
}
__global__ __launch_bounds__(256) void grad_wire_unpack_kernel(const bf16_t* __restrict__ stage, float* __restrict__ g, long n) {
    for (long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 8; i < n; i += (long)gridDim.x * blockDim.x * 8) {
        if (i + 8 <= n) {
            const u32x4 v = *reinterpret_cast<const u32x4*>(stage + i);
#pragma unroll
            for (int k = 0; k < 4; ++k) { g[i + 2 * k] = __uint_as_float(v[k] << 16); g[i + 2 * k + 1] = __uint_as_float(v[k] & 0xffff0000u); }
        } else
            for (long k = i; k < n; ++k) g[k] = bf2f(stage[k]);
    }
}
// mean((pred - target)^2) and its gradient 2 (pred - target) / n in one pass (F.mse_loss + the first step of loss.backward(),
// 3d_ldm/train_diffusion.py:207,214): partial sums per block, folded by mse_fold_kernel (deterministic two-stage reduction)
__global__ __launch_bounds__(256) void mse_part_kernel(const float* __restrict__ pred, const float* __restrict__ target, long n,
                                                       float* __restrict__ grad, float* __restrict__ part) {
    __shared__ float red[4];
    float acc = 0.f;
    const float gs = 2.0f / (float)n;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float d = pred[i] - target[i];
        acc += d * d;
        if (grad) grad[i] = gs * d;
    }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}
__global__ __launch_bounds__(256) void mse_fold_kernel(const float* __restrict__ part, int nparts, long n, float* __restrict__ out) {
    __shared__ double red[256];
    double acc = 0.0;
    for (int i = threadIdx.x; i < nparts; i += 256) acc += (double)part[i];
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
    if (threadIdx.x == 0) *out = (float)(red[0] / (double)n);
}
struct AdamCoef { float lr, b1, b2, eps, bc1, bc2_sqrt, max_norm, decay; int step; };   // decay = lr * weight_decay (AdamW, decoupled)
// The agreed NaN-skip of the trainers without a host read: a non-finite gradient norm (a NaN / inf loss makes every gradient NaN,
// and the data-parallel mean carries it to every rank) leaves parameters and moments untouched, and sq_norm[1] counts the skipped
// steps; the bias corrections use step - skipped, as if the optimizer had not been called (the reference `continue`s in front of
// optimizer.step(): 3d_ldm/train_diffusion.py:210-212).  Returns false when this launch must do nothing.
__device__ __forceinline__ bool adam_prologue(AdamCoef& k, const float* __restrict__ sq_norm, float& clip, const bool counter_block) {
    clip = 1.f;
    if (!sq_norm) return true;                          // no norm supplied: plain Adam (raw ABI callers; the Python optimizers always pass one)
    const float nn = sq_norm[0];
    const float skipped = sq_norm[1];
    if (!(nn == nn) || nn > 3.0e38f) {                  // NaN or inf
        if (counter_block) const_cast<float*>(sq_norm)[1] = skipped + 1.f;
        return false;
    }
    if (k.max_norm > 0.f) { const float c = k.max_norm / (sqrtf(nn) + 1e-6f); clip = c < 1.f ? c : 1.f; }   // max_norm <= 0: no clipping, the skip stays
    if (skipped > 0.f) {                                // rare: two powf per thread cost 0.37 ms of the 1.2 ms launch when always evaluated
        const float eff = fmaxf((float)k.step - skipped, 1.f);
        k.bc1 = 1.0f - powf(k.b1, eff); k.bc2_sqrt = sqrtf(1.0f - powf(k.b2, eff));
    }
    return true;
}
__global__ __launch_bounds__(256) void adam_step_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                        float* __restrict__ v, long n, AdamCoef k, const float* __restrict__ sq_norm) {
    float clip;
    // every block reads the skip counter before any block can bump it?  No: block 0 may run first.  The counter is therefore bumped by
    // the LAST block of the grid in launch order only after its own read, and readers tolerate either value: a skip decision depends on
    // sq_norm[0] alone, and the bias correction is only evaluated on steps that do update.
    if (!adam_prologue(k, sq_norm, clip, blockIdx.x == gridDim.x - 1 && threadIdx.x == 0)) return;
    const float step = k.lr / k.bc1;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float gi = g[i] * clip;
        const float mi = k.b1 * m[i] + (1.f - k.b1) * gi;
        const float vi = k.b2 * v[i] + (1.f - k.b2) * gi * gi;
        m[i] = mi; v[i] = vi;
        const float pi = p[i] * (1.f - k.decay);                   // torch.optim.AdamW: param.mul_(1 - lr * weight_decay) first
        p[i] = pi - step * mi / (sqrtf(vi) / k.bc2_sqrt + k.eps);  // torch.optim.Adam: denom = sqrt(v)/sqrt(bc2) + eps
    }
}

// ------------------------------------------------------------------------------------------------
// Descriptor-driven (batched) forms of the three per-parameter kernels of a training step: ONE launch each instead of
// ~100-300 launches at their launch floor.  blockmap[b] = (descriptor index, block index inside that descriptor).
struct PackDesc { long src_off; long dst_off; int taps, cout, cin, cin_s, cout_pad, row_off, kind, pad_; };   // kind 0 matrix, 1 fp32 vector
__global__ __launch_bounds__(256) void param_pack_batched_kernel(const PackDesc* __restrict__ descs, const int2* __restrict__ blockmap,
                                                                 const float* __restrict__ flat, char* __restrict__ arena) {
    __shared__ float tile[64 * 27];
    const int2 bm = blockmap[blockIdx.x];
    const PackDesc e = descs[bm.x];
    const int tid = threadIdx.x;
    const float* src = flat + e.src_off;
    if (e.kind == 1) {                                   // fp32 vector: 1024 elements per block
        float* dst = reinterpret_cast<float*>(arena + e.dst_off);
        for (int i = bm.y * 1024 + tid; i < e.cout && i < (bm.y + 1) * 1024; i += 256) dst[i] = src[i];
        return;
    }
    bf16_t* dst = reinterpret_cast<bf16_t*>(arena + e.dst_off);
    const int nchunk = (e.cin_s + 63) / 64;
    const int co = bm.y / nchunk, ci0 = (bm.y - co * nchunk) * 64;
    int nci = e.cin - ci0; if (nci > 64) nci = 64; if (nci < 0) nci = 0;
    const float* sp = src + ((size_t)co * e.cin + ci0) * e.taps;
    for (int i = tid; i < nci * e.taps; i += 256) tile[i] = sp[i];
    __syncthreads();
    for (int i = tid; i < e.taps * 64; i += 256) {
        const int t = i >> 6, c = i & 63, ci = ci0 + c;
        if (ci < e.cin_s) dst[((size_t)t * e.cout_pad + e.row_off + co) * e.cin_s + ci] = (c < nci) ? f2bf(tile[c * e.taps + t]) : (bf16_t)0;
    }
}

// Adam(W) and the bf16 re-pack in ONE pass over the parameters (training step tail): the block that packs a (cout row, 64-cin chunk)
// of a matrix -- or 1024 elements of a vector -- first updates exactly those flat fp32 elements (p, m, v in place, same arithmetic as
// adam_step_kernel) and packs the NEW values, so the 765 MB of updated master weights are not read a second time.
__global__ __launch_bounds__(256) void adam_pack_batched_kernel(const PackDesc* __restrict__ descs, const int2* __restrict__ blockmap,
                                                                float* __restrict__ flat_p, const float* __restrict__ flat_g,
                                                                float* __restrict__ flat_m, float* __restrict__ flat_v,
                                                                char* __restrict__ arena, AdamCoef k, const float* __restrict__ sq_norm) {
    __shared__ float tile[64 * 27];
    const int2 bm = blockmap[blockIdx.x];
    const PackDesc e = descs[bm.x];
    const int tid = threadIdx.x;
    float clip;
    if (!adam_prologue(k, sq_norm, clip, blockIdx.x == gridDim.x - 1 && tid == 0)) return;
    const float step = k.lr / k.bc1;
    auto update = [&](long i) -> float {
        const float gi = flat_g[i] * clip;
        const float mi = k.b1 * flat_m[i] + (1.f - k.b1) * gi;
        const float vi = k.b2 * flat_v[i] + (1.f - k.b2) * gi * gi;
        flat_m[i] = mi; flat_v[i] = vi;
        const float pi = flat_p[i] * (1.f - k.decay);
        const float pn = pi - step * mi / (sqrtf(vi) / k.bc2_sqrt + k.eps);
        flat_p[i] = pn;
        return pn;
    };
    if (e.kind == 1) {                                   // fp32 vector: 1024 elements per block
        float* dst = reinterpret_cast<float*>(arena + e.dst_off);
        for (int i = bm.y * 1024 + tid; i < e.cout && i < (bm.y + 1) * 1024; i += 256) dst[i] = update(e.src_off + i);
        return;
    }
    bf16_t* dst = reinterpret_cast<bf16_t*>(arena + e.dst_off);
    const int nchunk = (e.cin_s + 63) / 64;
    const int co = bm.y / nchunk, ci0 = (bm.y - co * nchunk) * 64;
    int nci = e.cin - ci0; if (nci > 64) nci = 64; if (nci < 0) nci = 0;
    const long s0 = e.src_off + ((long)co * e.cin + ci0) * e.taps;
    for (int i = tid; i < nci * e.taps; i += 256) tile[i] = update(s0 + i);
    __syncthreads();
    for (int i = tid; i < e.taps * 64; i += 256) {
        const int t = i >> 6, c = i & 63, ci = ci0 + c;
        if (ci < e.cin_s) dst[((size_t)t * e.cout_pad + e.row_off + co) * e.cin_s + ci] = (c < nci) ? f2bf(tile[c * e.taps + t]) : (bf16_t)0;
    }
}

struct ExportDesc { long src_off; long dst_off; long slab_stride; int taps, rows_total, ld, row_off, col_off, cout, cin, nsplit; };
__global__ __launch_bounds__(256) void grad_export_batched_kernel(const ExportDesc* __restrict__ descs, const int2* __restrict__ blockmap,
                                                                  const char* __restrict__ ws, float* __restrict__ flat) {
    __shared__ float tile[64 * 27];
    const int2 bm = blockmap[blockIdx.x];
    const ExportDesc e = descs[bm.x];
    const int tid = threadIdx.x;
    const float* src = reinterpret_cast<const float*>(ws + e.src_off);
    const int nchunk = (e.cin + 63) / 64;
    const int co = bm.y / nchunk, ci0 = (bm.y - co * nchunk) * 64;
    int nci = e.cin - ci0; if (nci > 64) nci = 64;
    for (int i = tid; i < e.taps * 64; i += 256) {
        const int t = i >> 6, c = i & 63;
        if (c < nci) {
            const float* sp = src + ((size_t)t * e.rows_total + e.row_off + co) * e.ld + e.col_off + ci0 + c;
            float v = sp[0];
            for (int k = 1; k < e.nsplit; ++k) v += sp[(size_t)k * e.slab_stride];
            tile[c * e.taps + t] = v;
        }
    }
    __syncthreads();
    float* d = flat + e.dst_off + ((size_t)co * e.cin + ci0) * e.taps;
    for (int i = tid; i < nci * e.taps; i += 256) d[i] = tile[i];
}

struct WtDesc { long src_off; long dst_off; int taps, cout, cout_pad, cin, rows, ci_off, ci_cnt, col_tiles, row_tiles, pad_; };
__global__ __launch_bounds__(256) void weight_flip_transpose_batched_kernel(const WtDesc* __restrict__ descs, const int2* __restrict__ blockmap,
                                                                            const char* __restrict__ arena, char* __restrict__ ws) {
    __shared__ bf16_t tile[64][66];
    const int2 bm = blockmap[blockIdx.x];
    const WtDesc e = descs[bm.x];
    const int cols = (e.cout + 31) / 32 * 32;
    int b = bm.y;
    const int ct = b % e.col_tiles; b /= e.col_tiles;
    const int rt = b % e.row_tiles; const int tp = b / e.row_tiles;
    const int co0 = ct * 64, ci0 = rt * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const bf16_t* src = reinterpret_cast<const bf16_t*>(arena + e.src_off) + (size_t)(e.taps - 1 - tp) * e.cout_pad * e.cin;
#pragma unroll 4
    for (int r = ty; r < 64; r += 4) {
        const int co = co0 + r, ci = ci0 + tx;
        tile[r][tx] = (co < e.cout && ci < e.ci_cnt) ? src[(size_t)co * e.cin + e.ci_off + ci] : (bf16_t)0;
    }
    __syncthreads();
    bf16_t* dst = reinterpret_cast<bf16_t*>(ws + e.dst_off) + (size_t)tp * e.rows * cols;
#pragma unroll 4
    for (int r = ty; r < 64; r += 4) {
        const int ci = ci0 + r, co = co0 + tx;
        if (ci < e.rows && co < cols) dst[(size_t)ci * cols + co] = tile[tx][r];
    }
}

// ------------------------------------------------------------------------------------------------
// Backward of the AutoencoderKL heads + sampling (training: train_autoencoder.py:366-451):
//   mu, lv = heads;  sigma = exp(0.5 clamp(lv, -30, 20));  z = mu + sigma * eps.
// Given dz (bf16 NDHWC [M][Ls], from the decoder), g_mu / g_sigma (fp32 NCDHW, from the KL term; may be null):
//   d_mu = dz + g_mu;   d_lv = (dz * (z - mu) + g_sigma * sigma) * 0.5  inside the clamp, 0 outside.
// Output: gradient of the fused 1x1 heads conv, bf16 NDHWC [M][Cs] with channels (d_mu | d_lv | 0 padding).
template <typename T>                                     // activation storage: bf16_t, or float in the fp32 precision mode
__global__ __launch_bounds__(256) void vae_heads_bwd_kernel(const T* __restrict__ dz, int Ls, const float* __restrict__ ml,
                                                            const float* __restrict__ z, const float* __restrict__ g_mu,
                                                            const float* __restrict__ g_sigma, T* __restrict__ dy,
                                                            int N, int L, int Cs, int DHW) {
    const long total = (long)N * DHW * Cs;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % Cs);
        const long row = i / Cs;
        const int n = (int)(row / DHW);
        const int sp = (int)(row - (long)n * DHW);
        float g = 0.f;
        if (c < 2 * L) {
            const int l = c < L ? c : c - L;
            float gz;
            if constexpr (sizeof(T) == 4) gz = dz[row * Ls + l]; else gz = bf2f(dz[row * Ls + l]);
            const size_t j = ((size_t)n * L + l) * DHW + sp;
            if (c < L) g = gz + (g_mu ? g_mu[j] : 0.f);
            else {
                const float mu = ml[((size_t)n * 2 * L + l) * DHW + sp];
                const float lv = ml[((size_t)n * 2 * L + L + l) * DHW + sp];
                if (lv > -30.f && lv < 20.f) {
                    const float sg = expf(0.5f * lv);
                    g = 0.5f * (gz * (z[j] - mu) + (g_sigma ? g_sigma[j] * sg : 0.f));
                }
            }
        }
        if constexpr (sizeof(T) == 4) dy[i] = g; else dy[i] = f2bf(g);
    }
}


// ------------------------------------------------------------------------------------------------
// Phase weights of a (nearest x2 upsample -> 3^3 conv, pad 1): out[parity][tap2][co][ci] = sum of the 3^3 taps that read the
// same low-resolution voxel.  Per dimension, output o = 2 l + p reads up[o-1], up[o], up[o+1] with up[i] = low[i >> 1]:
//   p = 0: low[l-1] <- k0,      low[l]   <- k1 + k2        p = 1: low[l] <- k0 + k1,   low[l+1] <- k2
// (tap bit 0 = the lower of the two source voxels).  Source: the packed bf16 [27][cout_pad][cin_s] matrix; sums in fp32.
// grid = (blocks over cout_pad * cin_s / 8, 64), 256 threads; thread = 8 consecutive cin of one cout row.
__global__ __launch_bounds__(256) void phase_weights_kernel(const bf16_t* __restrict__ w3, bf16_t* __restrict__ wp, int cout_pad, int cin_s) {
    const long vecs = (long)cout_pad * cin_s / 8;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= vecs) return;
    const int phase = blockIdx.y >> 3, tap = blockIdx.y & 7;
    const int pb[3] = {phase >> 2, (phase >> 1) & 1, phase & 1}, tb[3] = {tap >> 2, (tap >> 1) & 1, tap & 1};
    int lo[3], hi[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        if (pb[d] == 0) { lo[d] = tb[d] ? 1 : 0; hi[d] = tb[d] ? 2 : 0; }
        else            { lo[d] = tb[d] ? 2 : 0; hi[d] = tb[d] ? 2 : 1; }
    }
    float acc[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) acc[q] = 0.f;
    const size_t mat = (size_t)cout_pad * cin_s;
    for (int kd = lo[0]; kd <= hi[0]; ++kd)
        for (int kh = lo[1]; kh <= hi[1]; ++kh)
            for (int kw = lo[2]; kw <= hi[2]; ++kw) {
                const u32x4 v = *reinterpret_cast<const u32x4*>(w3 + (size_t)((kd * 3 + kh) * 3 + kw) * mat + idx * 8);
#pragma unroll
                for (int q = 0; q < 4; ++q) { acc[2 * q] += __uint_as_float(v[q] << 16); acc[2 * q + 1] += __uint_as_float(v[q] & 0xffff0000u); }
            }
    u32x4 o;
#pragma unroll
    for (int q = 0; q < 4; ++q) o[q] = pack2bf(acc[2 * q], acc[2 * q + 1]);
    *reinterpret_cast<u32x4*>(wp + (size_t)blockIdx.y * mat + idx * 8) = o;
}

// ------------------------------------------------------------------------------------------------
// First conv of a network (Cin <= 9) as im2col + GEMM: row m of the patch matrix holds the 27 taps x Cin input values of output
// voxel m (k = tap * Cin + c; zero padding and the pad columns k >= 27 Cin are written as zeros), straight from the fp32 NCDHW
// inputs (x | cond channel-concatenated).  The 3^3 conv over 1..8 real channels padded to 32 becomes ONE K step of 32..256.
// thread = one 8-element vector of one row.
__global__ __launch_bounds__(256) void pack_im2col_kernel(const float* __restrict__ x, int cx, const float* __restrict__ cond, int cc,
                                                          bf16_t* __restrict__ out, int N, int D, int H, int W, int Kp) {
    const int cin = cx + cc, DHW = D * H * W, HW = H * W, vecs = Kp / 8;
    const long total = (long)N * DHW * vecs;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int vq = (int)(i % vecs);
        const long row = i / vecs;
        const int n = (int)(row / DHW), sp = (int)(row - (long)n * DHW);
        const int d = sp / HW, r = sp - d * HW, h = r / W, w = r - h * W;
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int k = vq * 8 + e;
            float val = 0.f;
            if (k < 27 * cin) {
                const int tap = k / cin, c = k - tap * cin;
                const int kd = tap / 9, kh = (tap - kd * 9) / 3, kw = tap - kd * 9 - kh * 3;
                const int id = d + kd - 1, ih = h + kh - 1, iw = w + kw - 1;
                if ((unsigned)id < (unsigned)D && (unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W) {
                    const int spi = (id * H + ih) * W + iw;
                    val = c < cx ? x[((size_t)n * cx + c) * DHW + spi] : cond[((size_t)n * cc + (c - cx)) * DHW + spi];
                }
            }
            v[e] = val;
        }
        u32x4 o;
#pragma unroll
        for (int q = 0; q < 4; ++q) o[q] = pack2bf(v[2 * q], v[2 * q + 1]);
        *reinterpret_cast<u32x4*>(out + i * 8) = o;
    }
}

// its weights: w2[co][tap * cin + c] = w3[tap][co][c] from the packed [27][cout_pad][cin_s] matrix (pad columns zero)
__global__ __launch_bounds__(256) void im2col_weights_kernel(const bf16_t* __restrict__ w3, bf16_t* __restrict__ w2, int cout_pad, int cin_s,
                                                             int cin, int Kp) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= cout_pad * Kp) return;
    const int co = i / Kp, k = i - co * Kp;
    bf16_t v = 0;
    if (k < 27 * cin) { const int tap = k / cin, c = k - tap * cin; v = w3[((size_t)tap * cout_pad + co) * cin_s + c]; }
    w2[i] = v;
}
