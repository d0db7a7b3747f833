// fp32 precision mode ("LDM_PREC_FP32"): the same launch plans on fp32 NDHWC activations and fp32 weights, every
// product on the gfx950 fp32 matrix instruction v_mfma_f32_32x32x2_f32 (exact f32 FMA chain, 64 FLOP/clk/SIMD =
// 157 TFLOP/s chip peak, 1/16 of the bf16 rate).
//
// Why it exists: the reference computes in fp32 (autocast is off: 3d_ldm/train_diffusion.py:177,237; inference.py:91-99
// has no autocast at all) and BASELINE.json's parity bar is 1e-3 rel-L2 against that CPU path.  A bf16 network of this
// depth sits at 3e-2 (~110 bf16 rounding points, chaotic amplification); this mode meets the bar
// (measured ~1e-5) at roughly a quarter of the bf16 throughput, still above the 50 steps/s target.
//
// The kernels are deliberately plain: at 64 cycles per MFMA the loop is matrix-pipe bound by a factor of ~10 over its
// operand traffic (16 KiB of operands per 2048 MFMA cycles per CU), so register-staged double buffering with one
// barrier per K step already keeps the pipe >90 % busy; none of the LDS-DMA ring machinery of conv_igemm.h is needed.
#pragma once
#include "common.h"
#include "conv_igemm.h"      // xcd_remap

typedef __attribute__((ext_vector_type(16))) float f32x16;

struct Conv32Params {
    const float* xa; const float* xb; int ca, cb;     // channel-concatenated sources, NDHWC fp32, channels % 16 == 0
    const float* w;                                   // [taps][CoutPad][ca + cb] fp32
    int N, Din, Hin, Win, Dout, Hout, Wout;
    int ksize, stride, pad, ups, exact;               // same addressing modes as ConvParams (conv_igemm.h)
    int M, CoutS, CoutPad, CoutReal;
    int nchunk, steps;                                // (ca + cb) / 16, taps * nchunk
    int splitk, steps_per_split, mtiles, ntiles;
    const float* bias; const float* temb; int temb_stride; const float* residual;
    float* out;                                       // [M][CoutS] fp32 NDHWC        (mode 0)
    float* out_ncdhw;                                 // [N][CoutReal][DHWo] fp32     (mode 1)
    float* partial;                                   // [splitk][M][CoutPad]         (splitk > 1)
    float* stats; int stats_nrb, stats_rows;          // finalize_stats_f32_kernel: [N * stats_nrb][CoutS][2] (sum, sum of squares) over stats_rows rows per block
};

// Workgroup = 128 voxels x BN couts (128, or 64 for the layers with Cout % 128 != 0: no wasted MFMA rows), 4 waves (2 x 2), wave tile
// 64 voxels x BN / 2 couts = 2 x NI MFMA tiles of 32 x 32 (A = weights, B = voxels: a lane ends with 4 consecutive couts per register
// quad of one voxel).  K step = 16 input channels of one tap.
template <int BN>
__global__ __launch_bounds__(256, 3) void conv_f32_kernel(const Conv32Params p) {
    constexpr int BM = 128, BK = 16, LDR = BK + 4, NI = BN / 64;   // LDS row stride 80 B: conflict-free ds_read_b128
    __shared__ __attribute__((aligned(16))) float sA[2][BN * LDR];  // weights (rows 64 .. 127 unused when BN == 64)
    __shared__ __attribute__((aligned(16))) float sB[2][BM * LDR];  // voxels
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave & 1, wn = wave >> 1;
    int lid = xcd_remap(blockIdx.x, gridDim.x);
    const int mtile = lid % p.mtiles; lid /= p.mtiles;
    const int ntile = lid % p.ntiles;
    const int split = lid / p.ntiles;
    const int m0 = mtile * BM, n0 = ntile * BN;
    const int s_begin = split * p.steps_per_split;
    int s_end = s_begin + p.steps_per_split; if (s_end > p.steps) s_end = p.steps;
    const int DHWo = p.Dout * p.Hout * p.Wout, HWo = p.Hout * p.Wout;
    const int cin = p.ca + p.cb;
    const int DinU = p.Din << p.ups, HinU = p.Hin << p.ups, WinU = p.Win << p.ups;

    const int lc = tid & 3, lr = tid >> 2;              // loader: 16-byte chunk of the 16-channel row, rows lr and lr + 64
    // output coordinates of the loader's two voxel rows (the (tap, row) -> source voxel map is computed per step from these: a table
    // in LDS would cost 13.5 KiB and the third workgroup per CU)
    int r_n[2], r_d[2], r_h[2], r_w[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int m = m0 + lr + 64 * j;
        r_n[j] = -1; r_d[j] = r_h[j] = r_w[j] = 0;
        if (m < p.M) {
            const int n = m / DHWo; int r = m - n * DHWo; const int od = r / HWo; r -= od * HWo; const int oh = r / p.Wout;
            r_n[j] = n; r_d[j] = od * p.stride - p.pad; r_h[j] = oh * p.stride - p.pad; r_w[j] = (r - oh * p.Wout) * p.stride - p.pad;
        }
    }
    float4 ra[NI], rb[2];
    auto load_step = [&](int s) {
        const int tap = s / p.nchunk, ch = (s - tap * p.nchunk) * BK;
        int kd = 0, kh = 0, kw = 0;
        if (p.ksize == 3) { kd = tap / 9; kh = (tap - kd * 9) / 3; kw = tap - kd * 9 - kh * 3; }
        const bool second = ch >= p.ca;
        const float* src = second ? p.xb : p.xa;
        const int cs = second ? p.cb : p.ca, cc = (second ? ch - p.ca : ch) + lc * 4;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int row = lr + 64 * j;
            const int id = r_d[j] + kd, ih = r_h[j] + kh, iw = r_w[j] + kw;
            const bool ok = (r_n[j] >= 0) & ((unsigned)id < (unsigned)DinU) & ((unsigned)ih < (unsigned)HinU) & ((unsigned)iw < (unsigned)WinU) &
                            !(p.exact & (id | ih | iw) & 1);
            const int v = ok ? ((r_n[j] * p.Din + (id >> p.ups)) * p.Hin + (ih >> p.ups)) * p.Win + (iw >> p.ups) : -1;
            rb[j] = (v >= 0) ? *reinterpret_cast<const float4*>(src + (size_t)v * cs + cc) : make_float4(0.f, 0.f, 0.f, 0.f);
            if (j < NI) {
                const int co = n0 + row;
                ra[j] = (co < p.CoutPad) ? *reinterpret_cast<const float4*>(p.w + ((size_t)tap * p.CoutPad + co) * cin + ch + lc * 4)
                                         : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
    };
    auto store_step = [&](int buf) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int row = lr + 64 * j;
            if (j < NI) *reinterpret_cast<float4*>(&sA[buf][row * LDR + lc * 4]) = ra[j];
            *reinterpret_cast<float4*>(&sB[buf][row * LDR + lc * 4]) = rb[j];
        }
    };

    f32x16 acc[NI][2];                                  // [cout tile][voxel tile]
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int fr = lane & 31, fh = lane >> 5;
    if (s_begin < s_end) { load_step(s_begin); store_step(0); }
    __syncthreads();
    for (int s = s_begin; s < s_end; ++s) {
        const int buf = (s - s_begin) & 1;
        if (s + 1 < s_end) load_step(s + 1);
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            // lane half fh reads 4 consecutive k (chunk 2 * half + fh); MFMA e pairs k = 8 half + e (lanes 0-31) with k = 8 half + 4 + e
            // (lanes 32-63) on BOTH operands, so the four MFMAs together cover the 8 k's of this half step
            float4 a[NI], b[2];
#pragma unroll
            for (int i = 0; i < NI; ++i) a[i] = *reinterpret_cast<const float4*>(&sA[buf][(wn * (BN / 2) + i * 32 + fr) * LDR + (2 * half + fh) * 4]);
#pragma unroll
            for (int j = 0; j < 2; ++j) b[j] = *reinterpret_cast<const float4*>(&sB[buf][(wm * 64 + j * 32 + fr) * LDR + (2 * half + fh) * 4]);
#pragma unroll
            for (int i = 0; i < NI; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].x, b[j].x, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].y, b[j].y, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].z, b[j].z, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].w, b[j].w, acc[i][j], 0, 0, 0);
                }
        }
        if (s + 1 < s_end) store_step(buf ^ 1);
        __syncthreads();
    }

    // ---- epilogue: accumulator register 4g + r of tile (i, j) = cout 32 i + 8 g + 4 fh + r of voxel 32 j + fr
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int m = m0 + wm * 64 + j * 32 + fr;
        if (m >= p.M) continue;
        const int n = m / DHWo, sp = m - n * DHWo;
#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int c = n0 + wn * (BN / 2) + i * 32 + 8 * g + 4 * fh;
                float4 v = make_float4(acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]);
                if (p.splitk > 1) {
                    if (c < p.CoutPad) *reinterpret_cast<float4*>(p.partial + ((size_t)split * p.M + m) * p.CoutPad + c) = v;
                    continue;
                }
                if (c >= p.CoutPad) continue;
                if (p.bias) { const float4 b = *reinterpret_cast<const float4*>(p.bias + c); v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w; }
                if (p.temb) { const float4 b = *reinterpret_cast<const float4*>(p.temb + (size_t)n * p.temb_stride + c); v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w; }
                if (p.out_ncdhw) {
                    const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                    for (int r = 0; r < 4; ++r) if (c + r < p.CoutReal) p.out_ncdhw[((size_t)n * p.CoutReal + c + r) * DHWo + sp] = vv[r];
                    continue;
                }
                if (c >= p.CoutS) continue;
                if (p.residual) { const float4 b = *reinterpret_cast<const float4*>(p.residual + (size_t)m * p.CoutS + c); v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w; }
                *reinterpret_cast<float4*>(p.out + (size_t)m * p.CoutS + c) = v;
            }
    }
}

// ---- the same convolution on the bf16 matrix cores with fp32-class accuracy ("3 x bf16"): every fp32 operand is split on its way into
// LDS into hi = bf16(x) and lo = bf16(x - hi), so x = hi + lo to 2^-18 relative, and every product is three MFMAs,
// hi*hi + hi*lo + lo*hi (the dropped lo*lo term is <= 2^-18 of the product), accumulated in fp32.  v_mfma_f32_16x16x32_bf16 runs at
// 16x the rate of v_mfma_f32_32x32x2_f32 (2.5 vs 0.157 PFLOP/s), so three of them per product are still ~5x faster than one fp32 MFMA; per-product
// error ~1e-5 relative, i.e. the same order as fp32 accumulation-order noise over K = 27 * Cin.  Same tile, grid, split-K and epilogue as
// conv_f32_kernel; K step = 32 input channels of one tap (Cin % 32 == 0, checked by the planner).  LDM_F32_X3=0 plans the plain fp32 kernel.
__device__ __forceinline__ void split_bf16x4(const float4 v, uint2& hi, uint2& lo) {
    const float x[4] = {v.x, v.y, v.z, v.w};
    unsigned h[4], l[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        h[k] = f2bf(x[k]);
        l[k] = f2bf(x[k] - __uint_as_float(h[k] << 16));
    }
    hi = make_uint2(h[0] | (h[1] << 16), h[2] | (h[3] << 16));
    lo = make_uint2(l[0] | (l[1] << 16), l[2] | (l[3] << 16));
}

template <int BN>
__global__ __launch_bounds__(256, 2) void conv_x3_kernel(const Conv32Params p) {
    constexpr int BM = 128, BK = 32, LDB = (BK + 8) * 2, NI = BN / 32;   // LDS row stride 80 B (bf16): conflict-free ds_read_b128; NI cout tiles of 16 per wave
    // [buffer][hi | lo][rows][80 B]
    __shared__ __attribute__((aligned(16))) char sA[2][2][BN * LDB];   // weights
    __shared__ __attribute__((aligned(16))) char sB[2][2][BM * LDB];   // voxels
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave & 1, wn = wave >> 1;
    int lid = xcd_remap(blockIdx.x, gridDim.x);
    const int mtile = lid % p.mtiles; lid /= p.mtiles;
    const int ntile = lid % p.ntiles;
    const int split = lid / p.ntiles;
    const int m0 = mtile * BM, n0 = ntile * BN;
    const int s_begin = split * p.steps_per_split;
    int s_end = s_begin + p.steps_per_split; if (s_end > p.steps) s_end = p.steps;
    const int DHWo = p.Dout * p.Hout * p.Wout, HWo = p.Hout * p.Wout;
    const int cin = p.ca + p.cb;
    const int DinU = p.Din << p.ups, HinU = p.Hin << p.ups, WinU = p.Win << p.ups;

    // loader: thread = (16-byte fp32 chunk lc of the 32-channel row, rows lr + 32 j)
    const int lc = tid & 7, lr = tid >> 3;
    constexpr int RJ = BM / 32, WJ = BN / 32;
    int r_n[RJ], r_d[RJ], r_h[RJ], r_w[RJ];
#pragma unroll
    for (int j = 0; j < RJ; ++j) {
        const int m = m0 + lr + 32 * j;
        r_n[j] = -1; r_d[j] = r_h[j] = r_w[j] = 0;
        if (m < p.M) {
            const int n = m / DHWo; int r = m - n * DHWo; const int od = r / HWo; r -= od * HWo; const int oh = r / p.Wout;
            r_n[j] = n; r_d[j] = od * p.stride - p.pad; r_h[j] = oh * p.stride - p.pad; r_w[j] = (r - oh * p.Wout) * p.stride - p.pad;
        }
    }
    float4 ra[WJ], rb[RJ];
    auto load_step = [&](int s) {
        const int tap = s / p.nchunk, ch = (s - tap * p.nchunk) * BK;
        int kd = 0, kh = 0, kw = 0;
        if (p.ksize == 3) { kd = tap / 9; kh = (tap - kd * 9) / 3; kw = tap - kd * 9 - kh * 3; }
        const bool second = ch >= p.ca;
        const float* src = second ? p.xb : p.xa;
        const int cs = second ? p.cb : p.ca, cc = (second ? ch - p.ca : ch) + lc * 4;
#pragma unroll
        for (int j = 0; j < RJ; ++j) {
            const int id = r_d[j] + kd, ih = r_h[j] + kh, iw = r_w[j] + kw;
            const bool ok = (r_n[j] >= 0) & ((unsigned)id < (unsigned)DinU) & ((unsigned)ih < (unsigned)HinU) & ((unsigned)iw < (unsigned)WinU) &
                            !(p.exact & (id | ih | iw) & 1);
            const int v = ok ? ((r_n[j] * p.Din + (id >> p.ups)) * p.Hin + (ih >> p.ups)) * p.Win + (iw >> p.ups) : -1;
            rb[j] = (v >= 0) ? *reinterpret_cast<const float4*>(src + (size_t)v * cs + cc) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int j = 0; j < WJ; ++j) {
            const int co = n0 + lr + 32 * j;
            ra[j] = (co < p.CoutPad) ? *reinterpret_cast<const float4*>(p.w + ((size_t)tap * p.CoutPad + co) * cin + ch + lc * 4)
                                     : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto store_step = [&](int buf) {
#pragma unroll
        for (int j = 0; j < RJ; ++j) {
            uint2 hi, lo; split_bf16x4(rb[j], hi, lo);
            *reinterpret_cast<uint2*>(&sB[buf][0][(lr + 32 * j) * LDB + lc * 8]) = hi;
            *reinterpret_cast<uint2*>(&sB[buf][1][(lr + 32 * j) * LDB + lc * 8]) = lo;
        }
#pragma unroll
        for (int j = 0; j < WJ; ++j) {
            uint2 hi, lo; split_bf16x4(ra[j], hi, lo);
            *reinterpret_cast<uint2*>(&sA[buf][0][(lr + 32 * j) * LDB + lc * 8]) = hi;
            *reinterpret_cast<uint2*>(&sA[buf][1][(lr + 32 * j) * LDB + lc * 8]) = lo;
        }
    };

    f32x4 acc[NI][4];                                   // [cout tile of 16][voxel tile of 16]: wave tile = (BN / 2) couts x 64 voxels
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int fr = lane & 15, fg = lane >> 4;           // fragment row, 8-deep k slice
    if (s_begin < s_end) { load_step(s_begin); store_step(0); }
    __syncthreads();
    for (int s = s_begin; s < s_end; ++s) {
        const int buf = (s - s_begin) & 1;
        if (s + 1 < s_end) load_step(s + 1);
        bf16x8 ah[NI], al[NI], bh[4], bl[4];
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int off = (wn * (BN / 2) + i * 16 + fr) * LDB + fg * 16;
            ah[i] = *reinterpret_cast<const bf16x8*>(&sA[buf][0][off]); al[i] = *reinterpret_cast<const bf16x8*>(&sA[buf][1][off]);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int off = (wm * 64 + j * 16 + fr) * LDB + fg * 16;
            bh[j] = *reinterpret_cast<const bf16x8*>(&sB[buf][0][off]); bl[j] = *reinterpret_cast<const bf16x8*>(&sB[buf][1][off]);
        }
#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);   // the small terms first
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
            }
        if (s + 1 < s_end) store_step(buf ^ 1);
        __syncthreads();
    }

    // ---- epilogue: accumulator register r of tile (i, j) = cout 16 i + 4 fg + r of voxel 16 j + fr
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int m = m0 + wm * 64 + j * 16 + fr;
        if (m >= p.M) continue;
        const int n = m / DHWo, sp = m - n * DHWo;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int c = n0 + wn * (BN / 2) + i * 16 + 4 * fg;
            float4 v = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
            if (p.splitk > 1) {
                if (c < p.CoutPad) *reinterpret_cast<float4*>(p.partial + ((size_t)split * p.M + m) * p.CoutPad + c) = v;
                continue;
            }
            if (c >= p.CoutPad) continue;
            if (p.bias) { const float4 b = *reinterpret_cast<const float4*>(p.bias + c); v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w; }
            if (p.temb) { const float4 b = *reinterpret_cast<const float4*>(p.temb + (size_t)n * p.temb_stride + c); v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w; }
            if (p.out_ncdhw) {
                const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int r = 0; r < 4; ++r) if (c + r < p.CoutReal) p.out_ncdhw[((size_t)n * p.CoutReal + c + r) * DHWo + sp] = vv[r];
                continue;
            }
            if (c >= p.CoutS) continue;
            if (p.residual) { const float4 b = *reinterpret_cast<const float4*>(p.residual + (size_t)m * p.CoutS + c); v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w; }
            *reinterpret_cast<float4*>(p.out + (size_t)m * p.CoutS + c) = v;
        }
    }
}

// split-K slabs -> epilogue (slabs summed in order: bitwise reproducible).  thread = one row x 4 channels.
__global__ __launch_bounds__(256) void finalize_f32_kernel(const Conv32Params p) {
    const int DHWo = p.Dout * p.Hout * p.Wout;
    const int cvec = p.CoutPad / 4;
    const long total = (long)p.M * cvec;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const int m = (int)(e / cvec), c = (int)(e - (long)m * cvec) * 4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        const size_t slab = (size_t)p.M * p.CoutPad;
        const float* src = p.partial + (size_t)m * p.CoutPad + c;
        int s = 0;
        for (; s + 8 <= p.splitk; s += 8) {              // eight slab loads in flight; summed in slab order (bitwise reproducible)
            float4 a[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) a[k] = *reinterpret_cast<const float4*>(src + (size_t)(s + k) * slab);
#pragma unroll
            for (int k = 0; k < 8; ++k) { v.x += a[k].x; v.y += a[k].y; v.z += a[k].z; v.w += a[k].w; }
        }
        for (; s < p.splitk; ++s) { const float4 a = *reinterpret_cast<const float4*>(src + (size_t)s * slab); v.x += a.x; v.y += a.y; v.z += a.z; v.w += a.w; }
        const int n = m / DHWo, sp = m - n * DHWo;
        if (p.bias) { const float4 b = *reinterpret_cast<const float4*>(p.bias + c); v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w; }
        if (p.temb) { const float4 b = *reinterpret_cast<const float4*>(p.temb + (size_t)n * p.temb_stride + c); v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w; }
        if (p.out_ncdhw) {
            const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int r = 0; r < 4; ++r) if (c + r < p.CoutReal) p.out_ncdhw[((size_t)n * p.CoutReal + c + r) * DHWo + sp] = vv[r];
            continue;
        }
        if (c >= p.CoutS) continue;
        if (p.residual) { const float4 b = *reinterpret_cast<const float4*>(p.residual + (size_t)m * p.CoutS + c); v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w; }
        *reinterpret_cast<float4*>(p.out + (size_t)m * p.CoutS + c) = v;
    }
}

// The same finalize for an output a GroupNorm reads next (fp32 inference plans): block = stats_rows consecutive rows of one sample x
// all CoutS <= 1024 channels, thread = 4 channels x every rows_par-th row; beside the store it leaves the block's per-channel
// (sum, sum of squares) in the layout gn_stats_f32_kernel writes, so gn32_fold_apply_kernel runs without a statistics pass of its own.
__global__ __launch_bounds__(256) void finalize_stats_f32_kernel(const Conv32Params p) {
    __shared__ float red[256 * 8];
    const int DHWo = p.Dout * p.Hout * p.Wout;
    const int cvec = p.CoutS / 4, tid = threadIdx.x;
    const int n = blockIdx.y, blk = blockIdx.x;
    const int rows_par = 256 / cvec > 0 ? 256 / cvec : 1;
    const int cv = tid % cvec, rl = tid / cvec;
    const int r0 = blk * p.stats_rows;
    int r1 = r0 + p.stats_rows; if (r1 > DHWo) r1 = DHWo;
    float s[4] = {0.f, 0.f, 0.f, 0.f}, q[4] = {0.f, 0.f, 0.f, 0.f};
    if (rl < rows_par) {
        const int c = cv * 4;
        float4 bi = make_float4(0.f, 0.f, 0.f, 0.f), te = bi;
        if (p.bias) bi = *reinterpret_cast<const float4*>(p.bias + c);
        if (p.temb) te = *reinterpret_cast<const float4*>(p.temb + (size_t)n * p.temb_stride + c);
        const size_t slab = (size_t)p.M * p.CoutPad;
        for (int r = r0 + rl; r < r1; r += rows_par) {
            const size_t m = (size_t)n * DHWo + r;
            const float* src = p.partial + m * p.CoutPad + c;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            int k0 = 0;
            for (; k0 + 8 <= p.splitk; k0 += 8) {        // slabs summed in order, as finalize_f32_kernel does
                float4 a[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) a[k] = *reinterpret_cast<const float4*>(src + (size_t)(k0 + k) * slab);
#pragma unroll
                for (int k = 0; k < 8; ++k) { v.x += a[k].x; v.y += a[k].y; v.z += a[k].z; v.w += a[k].w; }
            }
            for (; k0 < p.splitk; ++k0) { const float4 a = *reinterpret_cast<const float4*>(src + (size_t)k0 * slab); v.x += a.x; v.y += a.y; v.z += a.z; v.w += a.w; }
            if (p.bias) { v.x += bi.x; v.y += bi.y; v.z += bi.z; v.w += bi.w; }    // bias, then the time embedding, then the residual: finalize_f32_kernel's order, bit for bit
            if (p.temb) { v.x += te.x; v.y += te.y; v.z += te.z; v.w += te.w; }
            if (p.residual) { const float4 b = *reinterpret_cast<const float4*>(p.residual + m * p.CoutS + c); v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w; }
            *reinterpret_cast<float4*>(p.out + m * p.CoutS + c) = v;
            s[0] += v.x; q[0] += v.x * v.x; s[1] += v.y; q[1] += v.y * v.y; s[2] += v.z; q[2] += v.z * v.z; s[3] += v.w; q[3] += v.w * v.w;
        }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) { red[tid * 8 + k] = s[k]; red[tid * 8 + 4 + k] = q[k]; }
    __syncthreads();
    if (tid < cvec) {
        for (int rl2 = 1; rl2 < rows_par; ++rl2) {
            const int t2 = rl2 * cvec + tid;
#pragma unroll
            for (int k = 0; k < 4; ++k) { s[k] += red[t2 * 8 + k]; q[k] += red[t2 * 8 + 4 + k]; }
        }
        float* dst = p.stats + (((size_t)n * p.stats_nrb + blk) * p.CoutS + tid * 4) * 2;
#pragma unroll
        for (int k = 0; k < 4; ++k) { dst[2 * k] = s[k]; dst[2 * k + 1] = q[k]; }
    }
}

// ---- GroupNorm on fp32 tensors: per-slab partial sums (fp32 over <= a few rows per thread, folded in fp64 by gn_finalize_kernel), then apply
struct Gn32Params {
    const float* xa; const float* xb; int ca, cb; int DHW, N, nslab, rows_per_slab, silu;
    float* partial; const float* ab; float* out;
};
__global__ __launch_bounds__(256) void gn_stats_f32_kernel(const Gn32Params p) {
    __shared__ float red[256 * 8];
    const int C = p.ca + p.cb, cvec = C / 4;
    const int n = blockIdx.y, slab = blockIdx.x, tid = threadIdx.x;
    const int r0 = slab * p.rows_per_slab;
    int r1 = r0 + p.rows_per_slab; if (r1 > p.DHW) r1 = p.DHW;
    for (int cv0 = 0; cv0 < cvec; cv0 += 256) {            // C <= 1024: one pass
        const int nv = cvec - cv0 < 256 ? cvec - cv0 : 256;
        const int rows_par = 256 / nv > 0 ? 256 / nv : 1;
        const int cv = tid % nv, rl = tid / nv;
        float s[4] = {0.f, 0.f, 0.f, 0.f}, q[4] = {0.f, 0.f, 0.f, 0.f};
        if (rl < rows_par) {
            const int c = (cv0 + cv) * 4;
            const bool second = c >= p.ca;
            const float* base = second ? p.xb : p.xa;
            const int cs = second ? p.cb : p.ca, cc = second ? c - p.ca : c;
            for (int r = r0 + rl; r < r1; r += rows_par) {
                const float4 v = *reinterpret_cast<const float4*>(base + ((size_t)n * p.DHW + r) * cs + cc);
                s[0] += v.x; q[0] += v.x * v.x; s[1] += v.y; q[1] += v.y * v.y; s[2] += v.z; q[2] += v.z * v.z; s[3] += v.w; q[3] += v.w * v.w;
            }
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 4; ++k) { red[tid * 8 + k] = s[k]; red[tid * 8 + 4 + k] = q[k]; }
        __syncthreads();
        if (tid < nv) {
            for (int rl2 = 1; rl2 < rows_par; ++rl2) {
                const int t2 = rl2 * nv + tid;
#pragma unroll
                for (int k = 0; k < 4; ++k) { s[k] += red[t2 * 8 + k]; q[k] += red[t2 * 8 + 4 + k]; }
            }
            float* dst = p.partial + (((size_t)n * p.nslab + slab) * C + (cv0 + tid) * 4) * 2;
#pragma unroll
            for (int k = 0; k < 4; ++k) { dst[2 * k] = s[k]; dst[2 * k + 1] = q[k]; }
        }
    }
}
__global__ __launch_bounds__(256) void gn_apply_f32_kernel(const Gn32Params p) {
    const int C = p.ca + p.cb, cvec = C / 4;
    const long total = (long)p.N * p.DHW * cvec;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long row = i / cvec;
        const int c = (int)(i - row * cvec) * 4;
        const int n = (int)(row / p.DHW);
        const bool second = c >= p.ca;
        const float4 v = *reinterpret_cast<const float4*>(second ? p.xb + row * p.cb + (c - p.ca) : p.xa + row * p.ca + c);
        const float4* abp = reinterpret_cast<const float4*>(p.ab + ((size_t)n * C + c) * 2);
        const float4 ab0 = abp[0], ab1 = abp[1];
        float4 y = make_float4(v.x * ab0.x + ab0.y, v.y * ab0.z + ab0.w, v.z * ab1.x + ab1.y, v.w * ab1.z + ab1.w);
        if (p.silu == 1) { y.x = y.x / (1.0f + expf(-y.x)); y.y = y.y / (1.0f + expf(-y.y)); y.z = y.z / (1.0f + expf(-y.z)); y.w = y.w / (1.0f + expf(-y.w)); }
        else if (p.silu == 2) {                           // LeakyReLU(0.2) (InstanceNorm + LeakyReLU of the PatchDiscriminator)
            y.x = y.x > 0.f ? y.x : 0.2f * y.x; y.y = y.y > 0.f ? y.y : 0.2f * y.y; y.z = y.z > 0.f ? y.z : 0.2f * y.z; y.w = y.w > 0.f ? y.w : 0.2f * y.w;
        }
        *reinterpret_cast<float4*>(p.out + row * C + c) = y;
    }
}

// Statistics fold + apply in one launch (inference plans): every block folds the partial rows of the channels that cover its 64-channel
// slice (gn_fold_cover, norm_elem.h: the forward one-launch GroupNorm's fold, fp64 totals), derives mean / rstd of those groups and
// normalises its row chunk.  Replaces gn_finalize_kernel + gn_apply_f32_kernel.  grid = (row chunks, ceil(C / 64), N), 256 threads.
struct Gn32FusedParams {
    const float* xa; const float* xb; int ca, cb, DHW, N, nslab, groups, silu, rows_per_block; float eps;
    const float* partial; const float* gamma; const float* beta; float* out;
    const float* sa; const float* sb; int nrb_a, nrb_b;   // producer statistics (conv3_halo_kernel's fused fp32 epilogue) instead of `partial`: one slab per source tensor
    bf16_t* out_hl;                    // instead of `out`: [rows][hi(C) | lo(C)] bf16, y = hi + lo to ~2^-17 (the voxel operand of the 3 x bf16 halo conv)
};
__global__ __launch_bounds__(256) void gn32_fold_apply_kernel(const Gn32FusedParams p) {
    __shared__ __attribute__((aligned(16))) float part[8][96][4];
    __shared__ double csum[192][2];
    __shared__ float gstat[64][2];
    const int tid = threadIdx.x, n = blockIdx.z;
    const int C = p.ca + p.cb, cpg = C / p.groups;
    const int c0 = blockIdx.y * 64;
    int c1 = c0 + 64; if (c1 > C) c1 = C;
    const int g_lo = c0 / cpg, g_hi = (c1 + cpg - 1) / cpg;
    const int cov_lo = g_lo * cpg, ncov = g_hi * cpg - cov_lo;
    if (p.sa) gn_fold_cover(GnFoldSrc{p.sa, p.sb, p.ca, p.cb, p.nrb_a, p.nrb_b}, n, cpg, cov_lo, ncov, part, csum);
    else gn_fold_cover(GnFoldSrc{p.partial, nullptr, C, 0, p.nslab, 0}, n, cpg, cov_lo, ncov, part, csum);
    __syncthreads();
    if (tid < g_hi - g_lo) {
        double s = 0.0, q = 0.0;
        for (int k = 0; k < cpg; ++k) { s += csum[tid * cpg + k][0]; q += csum[tid * cpg + k][1]; }
        const double cnt = (double)cpg * (double)p.DHW;
        const double mean = s / cnt;
        double var = q / cnt - mean * mean; if (var < 0.0) var = 0.0;
        gstat[tid][0] = (float)mean; gstat[tid][1] = (float)(1.0 / sqrt(var + (double)p.eps));
    }
    __syncthreads();
    const int vec = tid & 15, rl = tid >> 4;               // 16 float4 per 64-channel slice x 16 row lanes
    const int c = c0 + vec * 4;
    if (c >= C) return;
    const bool second = c >= p.ca;
    const float* src = second ? p.xb : p.xa;
    const int cs = second ? p.cb : p.ca, cc = second ? c - p.ca : c;
    float a[4], b[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int g = (c + k) / cpg - g_lo;
        a[k] = p.gamma[c + k] * gstat[g][1];
        b[k] = p.beta[c + k] - gstat[g][0] * a[k];
    }
    const int r0 = blockIdx.x * p.rows_per_block;
    int r1 = r0 + p.rows_per_block; if (r1 > p.DHW) r1 = p.DHW;
    for (int r = r0 + rl; r < r1; r += 16) {
        const size_t row = (size_t)n * p.DHW + r;
        const float4 v = *reinterpret_cast<const float4*>(src + row * cs + cc);
        float4 y = make_float4(v.x * a[0] + b[0], v.y * a[1] + b[1], v.z * a[2] + b[2], v.w * a[3] + b[3]);
        if (p.silu == 1) { y.x = y.x / (1.0f + expf(-y.x)); y.y = y.y / (1.0f + expf(-y.y)); y.z = y.z / (1.0f + expf(-y.z)); y.w = y.w / (1.0f + expf(-y.w)); }
        else if (p.silu == 2) {                           // LeakyReLU(0.2) (InstanceNorm + LeakyReLU of the PatchDiscriminator)
            y.x = y.x > 0.f ? y.x : 0.2f * y.x; y.y = y.y > 0.f ? y.y : 0.2f * y.y; y.z = y.z > 0.f ? y.z : 0.2f * y.z; y.w = y.w > 0.f ? y.w : 0.2f * y.w;
        }
        if (p.out_hl) {
            uint2 hi, lo; split_bf16x4(y, hi, lo);
            *reinterpret_cast<uint2*>(p.out_hl + row * (2 * C) + c) = hi;
            *reinterpret_cast<uint2*>(p.out_hl + row * (2 * C) + C + c) = lo;
        } else
        *reinterpret_cast<float4*>(p.out + row * C + c) = y;
    }
}

// Nearest x2 upsample of an fp32 NDHWC tensor straight into the (hi | lo) bf16 split: [N][2D][2H][2W][hi(C) | lo(C)].  The voxel operand
// of the 3 x bf16 halo conv behind an Upsample block (the general fp32 kernels fold the upsample into their loader instead).
// up = 0: the split alone, same size (the voxel operand of the phase form of the Upsample conv, which walks the low-resolution grid).
__global__ __launch_bounds__(256) void upsample_split_f32_kernel(const float* __restrict__ x, bf16_t* __restrict__ out, int N, int C, int D, int H, int W, int up) {
    const int cvec = C / 4, Do = D << up, Ho = H << up, Wo = W << up;
    const long total = (long)N * Do * Ho * Wo * cvec;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const long row = e / cvec; const int c = (int)(e - row * cvec) * 4;
        long r = row; const int ow = (int)(r % Wo); r /= Wo; const int oh = (int)(r % Ho); r /= Ho; const int od = (int)(r % Do); const long n = r / Do;
        const long src = ((n * D + (od >> up)) * H + (oh >> up)) * W + (ow >> up);
        uint2 hi, lo; split_bf16x4(*reinterpret_cast<const float4*>(x + src * C + c), hi, lo);
        *reinterpret_cast<uint2*>(out + row * (2 * C) + c) = hi;
        *reinterpret_cast<uint2*>(out + row * (2 * C) + C + c) = lo;
    }
}

// Weights of the 3 x bf16 halo conv (ConvParams::x3_n): fp32 [taps * cout_pad][cin] -> bf16 [taps * cout_pad][hi(cin) | lo(cin) | hi(cin)]
__global__ __launch_bounds__(256) void x3_weights_kernel(const float* __restrict__ w, bf16_t* __restrict__ out, long rows, int cin) {
    const long total = rows * (cin / 4);
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const long r = e / (cin / 4); const int c = (int)(e - r * (cin / 4)) * 4;
        uint2 hi, lo; split_bf16x4(*reinterpret_cast<const float4*>(w + r * cin + c), hi, lo);
        bf16_t* dst = out + r * (3 * cin) + c;
        *reinterpret_cast<uint2*>(dst) = hi; *reinterpret_cast<uint2*>(dst + cin) = lo; *reinterpret_cast<uint2*>(dst + 2 * cin) = hi;
    }
}

// Phase weights (norm_elem.h phase_weights_kernel: the 3^3 taps that read the same low-resolution voxel, summed) of the 3 x bf16 form on
// conv_igemm_kernel: fp32 [27][cout_pad][cin] -> bf16 [parity 8][tap 8][cout_pad][hi | hi | lo] (sums in fp32, then the split).
// grid = (blocks over cout_pad * cin / 4, 64), 256 threads; thread = 4 consecutive cin of one cout row.
__global__ __launch_bounds__(256) void x3_phase_weights_kernel(const float* __restrict__ w3, bf16_t* __restrict__ wp, int cout_pad, int cin) {
    const long vecs = (long)cout_pad * cin / 4;
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
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    const size_t mat = (size_t)cout_pad * cin;
    for (int kd = lo[0]; kd <= hi[0]; ++kd)
        for (int kh = lo[1]; kh <= hi[1]; ++kh)
            for (int kw = lo[2]; kw <= hi[2]; ++kw) {
                const float4 v = *reinterpret_cast<const float4*>(w3 + (size_t)((kd * 3 + kh) * 3 + kw) * mat + idx * 4);
                acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
            }
    const long row = idx / (cin / 4); const int c = (int)(idx - row * (cin / 4)) * 4;
    uint2 h, l; split_bf16x4(acc, h, l);
    bf16_t* dst = wp + ((size_t)blockIdx.y * cout_pad + row) * (3 * cin) + c;
    *reinterpret_cast<uint2*>(dst) = h; *reinterpret_cast<uint2*>(dst + cin) = h; *reinterpret_cast<uint2*>(dst + 2 * cin) = l;
}

// ---- self-attention on fp32 q|k|v rows [B*N][3C] (q | k | v, heads of d channels each), flash style, fp32 MFMA.
// A wave owns 32 queries; per 32-key tile  S^T = K Q^T  (32x32x2 MFMA over d) lands with the query on the lane and 16 of the 32 keys
// in the lane's registers (the other 16 in lane ^ 32), so the softmax needs one cross-lane exchange per row statistic and P feeds
// O^T += V^T P^T  as the B operand straight from those registers (k order permuted identically on the V^T operand).
struct Attn32Params { const float* qkv; float* out; int B, N, C, heads, d; float scale; float* lse; int x3; };   // lse (optional): [B][heads][N] natural log-sum-exp of the scaled scores; x3: the 3 x bf16 form of both products (inference plans)

// x = hi + lo with hi = bf16(x), lo = bf16(x - hi): |x - hi - lo| <= 2^-18 |x|.  Eight consecutive k of one MFMA operand row.
__device__ __forceinline__ void split8_bf16(const float (&x)[8], bf16x8& hi, bf16x8& lo) {
    u32x4 h, l;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        h[q] = pack2bf(x[2 * q], x[2 * q + 1]);
        l[q] = pack2bf(x[2 * q] - __uint_as_float(h[q] << 16), x[2 * q + 1] - __uint_as_float(h[q] & 0xffff0000u));
    }
    hi = __builtin_bit_cast(bf16x8, h); lo = __builtin_bit_cast(bf16x8, l);
}

template <int DT>                                        // DT = d / 32
__global__ __launch_bounds__(256) void attn_f32_kernel(const Attn32Params p) {
    constexpr int D = DT * 32, LDQ = D + 1;              // odd row stride: the strided K / Q reads hit 32 different banks
    constexpr int NW = (DT <= 4) ? 4 : 2;                // waves (32 queries each) per workgroup; d = 256: LDS holds only 64 query rows
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* sQ = reinterpret_cast<float*>(smem);                                    // [NW * 32][LDQ]
    float* sK = sQ + NW * 32 * LDQ;                      // [32][LDQ]
    float* sV = sK + 32 * LDQ;                           // [32][LDQ]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 31, fh = lane >> 5;
    const int b = blockIdx.z, head = blockIdx.y, q0 = blockIdx.x * (NW * 32);
    const int C3 = 3 * p.C;
    const float* base = p.qkv + (size_t)b * p.N * C3 + head * D;
    const int nthr = NW * 64;
    if (tid < nthr) {
        for (int e = tid; e < NW * 32 * (D / 4); e += nthr) {
            const int row = e / (D / 4), c4 = (e - row * (D / 4)) * 4;
            int q = q0 + row; if (q >= p.N) q = p.N - 1;
            const float4 v = *reinterpret_cast<const float4*>(base + (size_t)q * C3 + c4);
            float* dst = sQ + row * LDQ + c4; dst[0] = v.x; dst[1] = v.y; dst[2] = v.z; dst[3] = v.w;
        }
    }
    f32x16 o[DT];
#pragma unroll
    for (int t = 0; t < DT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[t][r] = 0.f;
    float mrow = -INFINITY, lrow = 0.f;                  // running max / sum of this lane's query (both lane halves hold the same values)
    const float sl2 = p.scale * 1.4426950408889634f;     // softmax in base 2
    const bool active = wave < NW;
    for (int k0 = 0; k0 < p.N; k0 += 32) {
        __syncthreads();                                 // previous tile's reads are done (first pass: Q is being written, harmless)
        if (tid < nthr) {
            for (int e = tid; e < 32 * (D / 4); e += nthr) {
                const int row = e / (D / 4), c4 = (e - row * (D / 4)) * 4;
                int k = k0 + row; if (k >= p.N) k = p.N - 1;
                const float4 kv = *reinterpret_cast<const float4*>(base + (size_t)k * C3 + p.C + c4);
                const float4 vv = *reinterpret_cast<const float4*>(base + (size_t)k * C3 + 2 * p.C + c4);
                float* dk = sK + row * LDQ + c4; dk[0] = kv.x; dk[1] = kv.y; dk[2] = kv.z; dk[3] = kv.w;
                float* dv = sV + row * LDQ + c4; dv[0] = vv.x; dv[1] = vv.y; dv[2] = vv.z; dv[3] = vv.w;
            }
        }
        __syncthreads();
        if (!active) continue;
        f32x16 s;
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = 0.f;
        const float* kq = sK + fr * LDQ + fh;            // A[key fr][dd = 2 kk + fh]
        const float* qq = sQ + (wave * 32 + fr) * LDQ + fh;
#pragma unroll 8
        for (int kk = 0; kk < D / 2; ++kk) s = __builtin_amdgcn_mfma_f32_32x32x2f32(kq[2 * kk], qq[2 * kk], s, 0, 0, 0);
        // register r of lane (query fr, half fh) = key (r & 3) + 8 (r >> 2) + 4 fh
        float mx = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = k0 + (r & 3) + 8 * (r >> 2) + 4 * fh;
            s[r] = (key < p.N) ? s[r] * sl2 : -INFINITY;
            mx = fmaxf(mx, s[r]);
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float mnew = fmaxf(mrow, mx);
        const float alpha = exp2f(mrow - mnew);          // first tile: exp2(-inf) = 0
        float ps = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) { s[r] = exp2f(s[r] - mnew); ps += s[r]; }
        ps += __shfl_xor(ps, 32, 64);
        lrow = lrow * alpha + ps; mrow = mnew;
#pragma unroll
        for (int t = 0; t < DT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[t][r] *= alpha;
        // O^T[dv][query] += V^T[dv][key] P^T[key][query]: step j pairs key (j & 3) + 8 (j >> 2) (lanes 0-31) with that key + 4 (lanes 32-63)
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const float* vrow = sV + ((j & 3) + 8 * (j >> 2) + 4 * fh) * LDQ + fr;
#pragma unroll
            for (int t = 0; t < DT; ++t) o[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(vrow[32 * t], s[j], o[t], 0, 0, 0);
        }
    }
    if (!active) return;
    const int q = q0 + wave * 32 + fr;
    if (q >= p.N) return;
    if (p.lse && fh == 0) p.lse[((size_t)b * p.heads + head) * p.N + q] = mrow * 0.6931471805599453f + logf(lrow);
    const float inv = 1.0f / lrow;
    float* dst = p.out + ((size_t)b * p.N + q) * p.C + head * D;
#pragma unroll
    for (int t = 0; t < DT; ++t)
#pragma unroll
        for (int g = 0; g < 4; ++g)
            *reinterpret_cast<float4*>(dst + 32 * t + 8 * g + 4 * fh) =
                make_float4(o[t][4 * g] * inv, o[t][4 * g + 1] * inv, o[t][4 * g + 2] * inv, o[t][4 * g + 3] * inv);
}

// The same attention with the KEY range split over the workgroup's four waves (d <= 64): the plain kernel gives one wave per 32 queries,
// i.e. 216 waves for 4 heads x 1728 tokens on a chip with 1024 SIMDs; here the four waves of a workgroup share the 32 queries, wave w
// walks key tiles w, w + 4, ... with its own K / V images, and the (max, sum, O) partials are merged through LDS in a fixed order.
// X3 (round 5, inference plans): both products as three bf16 MFMAs (32x32x16) on hi / lo splits of the fp32 operands -- hi*lo + lo*hi + hi*hi,
// fp32 accumulate, the dropped lo*lo term is <= 2^-18 of the product -- instead of the 32x32x2 fp32 MFMA (1/16 of the bf16 rate): per
// 32-key tile 12 + 12 MFMAs of 32 cycles instead of 32 + 32 of 64.  The softmax stays fp32; P is split in registers (the lane that holds a
// query's 16 scores feeds them as the B operand with the k order (r & 3) + 8 (r >> 2) + 4 fh applied to the V^T operand as well).
template <int DT, bool X3 = false>
__global__ __launch_bounds__(256) void attn_f32_split_kernel(const Attn32Params p) {
    constexpr int D = DT * 32, LDQ = D + 1;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* sQ = reinterpret_cast<float*>(smem);          // [32][LDQ]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float* sK = sQ + 32 * LDQ + wave * (64 * LDQ);       // this wave's K image [32][LDQ], then its V image
    float* sV = sK + 32 * LDQ;
    const int fr = lane & 31, fh = lane >> 5;
    const int b = blockIdx.z, head = blockIdx.y, q0 = blockIdx.x * 32;
    const int C3 = 3 * p.C;
    const float* base = p.qkv + (size_t)b * p.N * C3 + head * D;
    for (int e = tid; e < 32 * (D / 4); e += 256) {
        const int row = e / (D / 4), c4 = (e - row * (D / 4)) * 4;
        int q = q0 + row; if (q >= p.N) q = p.N - 1;
        const float4 v = *reinterpret_cast<const float4*>(base + (size_t)q * C3 + c4);
        float* dst = sQ + row * LDQ + c4; dst[0] = v.x; dst[1] = v.y; dst[2] = v.z; dst[3] = v.w;
    }
    f32x16 o[DT];
#pragma unroll
    for (int t = 0; t < DT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[t][r] = 0.f;
    float mrow = -INFINITY, lrow = 0.f;
    const float sl2 = p.scale * 1.4426950408889634f;
    const int ntile = (p.N + 31) / 32, nround = (ntile + 3) / 4;
    // the wave's K / V images are private to it: the next tile is fetched into registers while this one is multiplied (the loop was one
    // exposed global round trip per tile: 85 -> 5x us at N = 1728)
    constexpr int NLD = 32 * (D / 4) / 64;               // float4 per lane and operand
    float4 kreg[NLD], vreg[NLD];
    auto fetch = [&](const int t) {
        const int k0 = t * 32;
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int e = lane + 64 * i, row = e / (D / 4), c4 = (e - row * (D / 4)) * 4;
            int k = k0 + row; if (k >= p.N) k = p.N - 1;
            kreg[i] = *reinterpret_cast<const float4*>(base + (size_t)k * C3 + p.C + c4);
            vreg[i] = *reinterpret_cast<const float4*>(base + (size_t)k * C3 + 2 * p.C + c4);
        }
    };
    if (wave < ntile) fetch(wave);
    bf16x8 qh[X3 ? D / 16 : 1], ql[X3 ? D / 16 : 1];       // X3: this lane's Q fragments (query fr, k = 16 ks + 8 fh ..), split once
    if constexpr (X3) {
        __syncthreads();                                 // Q is in LDS
#pragma unroll
        for (int ks = 0; ks < D / 16; ++ks) {
            float qx[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) qx[j] = sQ[fr * LDQ + 16 * ks + 8 * fh + j];
            split8_bf16(qx, qh[ks], ql[ks]);
        }
    }
    for (int it = 0; it < nround; ++it) {
        const int t = it * 4 + wave, k0 = t * 32;
        __syncthreads();                                 // Q written (first round) / the previous tile's reads are done
        if (t < ntile) {
#pragma unroll
            for (int i = 0; i < NLD; ++i) {
                const int e = lane + 64 * i, row = e / (D / 4), c4 = (e - row * (D / 4)) * 4;
                float* dk = sK + row * LDQ + c4; dk[0] = kreg[i].x; dk[1] = kreg[i].y; dk[2] = kreg[i].z; dk[3] = kreg[i].w;
                float* dv = sV + row * LDQ + c4; dv[0] = vreg[i].x; dv[1] = vreg[i].y; dv[2] = vreg[i].z; dv[3] = vreg[i].w;
            }
            if (t + 4 < ntile) fetch(t + 4);
        }
        __syncthreads();
        if (t >= ntile) continue;
        f32x16 s;
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = 0.f;
        if constexpr (X3) {
#pragma unroll
            for (int ks = 0; ks < D / 16; ++ks) {
                float kx[8]; bf16x8 kh, kl;
#pragma unroll
                for (int j = 0; j < 8; ++j) kx[j] = sK[fr * LDQ + 16 * ks + 8 * fh + j];
                split8_bf16(kx, kh, kl);
                s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kh, ql[ks], s, 0, 0, 0);       // small terms first
                s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kl, qh[ks], s, 0, 0, 0);
                s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kh, qh[ks], s, 0, 0, 0);
            }
        } else {
        const float* kq = sK + fr * LDQ + fh;
        const float* qq = sQ + fr * LDQ + fh;
#pragma unroll 8
        for (int kk = 0; kk < D / 2; ++kk) s = __builtin_amdgcn_mfma_f32_32x32x2f32(kq[2 * kk], qq[2 * kk], s, 0, 0, 0);
        }
        float mx = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = k0 + (r & 3) + 8 * (r >> 2) + 4 * fh;
            s[r] = (key < p.N) ? s[r] * sl2 : -INFINITY;
            mx = fmaxf(mx, s[r]);
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float mnew = fmaxf(mrow, mx);
        const float alpha = exp2f(mrow - mnew);
        float ps = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) { s[r] = exp2f(s[r] - mnew); ps += s[r]; }
        ps += __shfl_xor(ps, 32, 64);
        lrow = lrow * alpha + ps; mrow = mnew;
#pragma unroll
        for (int tt = 0; tt < DT; ++tt)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[tt][r] *= alpha;
        if constexpr (X3) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {             // k slot (ks, fh, j) <-> key (r & 3) + 8 (r >> 2) + 4 fh with r = 8 ks + j, on BOTH operands
                float px[8]; bf16x8 ph, pl;
#pragma unroll
                for (int j = 0; j < 8; ++j) px[j] = s[8 * ks + j];
                split8_bf16(px, ph, pl);
#pragma unroll
                for (int tt = 0; tt < DT; ++tt) {
                    float vx[8]; bf16x8 vh, vl;
#pragma unroll
                    for (int j = 0; j < 8; ++j) { const int r = 8 * ks + j; vx[j] = sV[((r & 3) + 8 * (r >> 2) + 4 * fh) * LDQ + 32 * tt + fr]; }
                    split8_bf16(vx, vh, vl);
                    o[tt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vh, pl, o[tt], 0, 0, 0);
                    o[tt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vl, ph, o[tt], 0, 0, 0);
                    o[tt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vh, ph, o[tt], 0, 0, 0);
                }
            }
        } else {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const float* vrow = sV + ((j & 3) + 8 * (j >> 2) + 4 * fh) * LDQ + fr;
#pragma unroll
            for (int tt = 0; tt < DT; ++tt) o[tt] = __builtin_amdgcn_mfma_f32_32x32x2f32(vrow[32 * tt], s[j], o[tt], 0, 0, 0);
        }
        }
    }
    // ---- merge the four waves' partials (waves 1..3 publish, wave 0 folds them in wave order: reproducible)
    __syncthreads();
    float* xo = sQ + 32 * LDQ + wave * (64 * LDQ);        // the wave's own (dead) tile area: DT * 16 * 64 floats of O, then m and l per lane
    if (wave > 0) {
#pragma unroll
        for (int tt = 0; tt < DT; ++tt)
#pragma unroll
            for (int r = 0; r < 16; ++r) xo[(tt * 16 + r) * 64 + lane] = o[tt][r];
        xo[DT * 16 * 64 + lane] = mrow; xo[DT * 16 * 64 + 64 + lane] = lrow;
    }
    __syncthreads();
    if (wave > 0) return;
    for (int w = 1; w < 4; ++w) {
        const float* xw = sQ + 32 * LDQ + w * (64 * LDQ);
        const float mg = xw[DT * 16 * 64 + lane], lg = xw[DT * 16 * 64 + 64 + lane];
        const float mnew = fmaxf(mrow, mg);               // wave 0 owns tile 0: mrow is finite
        const float a0 = exp2f(mrow - mnew), a1 = exp2f(mg - mnew);      // a wave without a tile has mg = -inf -> a1 = 0
        lrow = lrow * a0 + lg * a1; mrow = mnew;
#pragma unroll
        for (int tt = 0; tt < DT; ++tt)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[tt][r] = o[tt][r] * a0 + xw[(tt * 16 + r) * 64 + lane] * a1;
    }
    const int q = q0 + fr;
    if (q >= p.N) return;
    if (p.lse && fh == 0) p.lse[((size_t)b * p.heads + head) * p.N + q] = mrow * 0.6931471805599453f + logf(lrow);
    const float inv = 1.0f / lrow;
    float* dst = p.out + ((size_t)b * p.N + q) * p.C + head * D;
#pragma unroll
    for (int tt = 0; tt < DT; ++tt)
#pragma unroll
        for (int g = 0; g < 4; ++g)
            *reinterpret_cast<float4*>(dst + 32 * tt + 8 * g + 4 * fh) =
                make_float4(o[tt][4 * g] * inv, o[tt][4 * g + 1] * inv, o[tt][4 * g + 2] * inv, o[tt][4 * g + 3] * inv);
}

static hipError_t launch_attn_f32(const Attn32Params& p, hipStream_t s) {
    if (p.d <= 64 && p.N >= 128) {                        // key range split over the four waves of a workgroup
        const int lds = (32 + 4 * 64) * (p.d + 1) * 4;
        const dim3 grid((p.N + 31) / 32, p.heads, p.B);
        if (p.d == 64 && p.x3) {
            static bool set64x = false;
            if (!set64x) { hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_f32_split_kernel<2, true>), hipFuncAttributeMaxDynamicSharedMemorySize, lds); if (e != hipSuccess) return e; set64x = true; }
            hipLaunchKernelGGL((attn_f32_split_kernel<2, true>), grid, dim3(256), lds, s, p);
        } else if (p.d == 64) {
            static bool set64 = false;
            if (!set64) { hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_f32_split_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, lds); if (e != hipSuccess) return e; set64 = true; }
            hipLaunchKernelGGL(attn_f32_split_kernel<2>, grid, dim3(256), lds, s, p);
        } else {
            static bool set32 = false;
            if (!set32) { hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_f32_split_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, lds); if (e != hipSuccess) return e; set32 = true; }
            hipLaunchKernelGGL(attn_f32_split_kernel<1>, grid, dim3(256), lds, s, p);
        }
        return hipSuccess;
    }

    const int dt = p.d / 32;
    const int nw = dt <= 4 ? 4 : 2;
    const int lds = ((nw * 32 + 64) * (p.d + 1)) * 4;
    const dim3 grid((p.N + nw * 32 - 1) / (nw * 32), p.heads, p.B);
#define LDM_A32(DT_) case DT_: { \
        static bool set_ = false; \
        if (!set_) { hipError_t e_ = hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_f32_kernel<DT_>), hipFuncAttributeMaxDynamicSharedMemorySize, lds); \
                     if (e_ != hipSuccess) return e_; set_ = true; } \
        hipLaunchKernelGGL(attn_f32_kernel<DT_>, grid, dim3(256), lds, s, p); break; }
    switch (dt) { LDM_A32(1) LDM_A32(2) LDM_A32(4) LDM_A32(8) default: return hipErrorInvalidValue; }
#undef LDM_A32
    return hipSuccess;
}

// ---- small ones
__global__ __launch_bounds__(256) void pack2_ncdhw_f32_kernel(const float* __restrict__ x, int cx, const float* __restrict__ cond, int cc,
                                                              float* __restrict__ out, int N, int Cs, int DHW) {
    const long total = (long)N * DHW * Cs;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % Cs);
        const long row = i / Cs;
        const int n = (int)(row / DHW);
        const int sp = (int)(row - (long)n * DHW);
        float v = 0.f;
        if (c < cx) v = x[((size_t)n * cx + c) * DHW + sp];
        else if (c < cx + cc) v = cond[((size_t)n * cc + (c - cx)) * DHW + sp];
        out[i] = v;
    }
}
__global__ __launch_bounds__(256) void gemv_f32_kernel(const float* __restrict__ W, const float* __restrict__ bias, const float* __restrict__ x,
                                                       float* __restrict__ y, int I, int O, int x_stride, int y_stride, int silu_in) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int o = blockIdx.x * 4 + wave, b = blockIdx.y;
    if (o >= O) return;
    const float* wr = W + (size_t)o * I;
    const float* xr = x + (size_t)b * x_stride;
    float acc = 0.f;
    for (int i = lane * 4; i < I; i += 256) {
        const float4 w = *reinterpret_cast<const float4*>(wr + i);
        float4 xv = *reinterpret_cast<const float4*>(xr + i);
        if (silu_in) { xv.x = xv.x / (1.0f + expf(-xv.x)); xv.y = xv.y / (1.0f + expf(-xv.y)); xv.z = xv.z / (1.0f + expf(-xv.z)); xv.w = xv.w / (1.0f + expf(-xv.w)); }
        acc += w.x * xv.x + w.y * xv.y + w.z * xv.z + w.w * xv.w;
    }
    acc = wave_sum(acc);
    if (lane == 0) y[(size_t)b * y_stride + o] = acc + (bias ? bias[o] : 0.f);
}
// fp32 [cout][cin][taps] (MONAI layout) -> fp32 [tap][cout_pad][cin_s] rows row_off.. (the bf16 arena's matrix layout, unrounded)
__global__ __launch_bounds__(256) void param_pack_f32_kernel(const float* __restrict__ src, float* __restrict__ dst, int taps, int cout, int cin,
                                                             int cin_s, int cout_pad, int row_off) {
    const long total = (long)taps * cout * cin_s;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int ci = (int)(i % cin_s);
        const long r = i / cin_s;
        const int co = (int)(r % cout), t = (int)(r / cout);
        dst[((size_t)t * cout_pad + row_off + co) * cin_s + ci] = ci < cin ? src[((size_t)co * cin + ci) * taps + t] : 0.f;
    }
}

// ---- debug taps (both precisions): NDHWC activation <-> fp32 NCDHW
template <typename T> __device__ __forceinline__ float tap_ld(const T* p);
template <> __device__ __forceinline__ float tap_ld<float>(const float* p) { return *p; }
template <> __device__ __forceinline__ float tap_ld<bf16_t>(const bf16_t* p) { return bf2f(*p); }
template <typename T>
__global__ __launch_bounds__(256) void tap_export_kernel(const T* __restrict__ act, float* __restrict__ out, int N, int C, int Cs, int DHW) {
    const long total = (long)N * C * DHW;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int sp = (int)(i % DHW);
        const long r = i / DHW;
        const int c = (int)(r % C), n = (int)(r / C);
        out[i] = tap_ld<T>(act + ((size_t)n * DHW + sp) * Cs + c);
    }
}
