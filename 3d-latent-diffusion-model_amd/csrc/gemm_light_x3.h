// fp32 precision mode: the 1x1x1 convolutions (attention q|k|v and output projections, the ResBlocks' channel-changing skips) as the
// light GEMM of gemm_light.h on fp32 operands.  Every lane loads its MFMA fragments straight from global memory as fp32 (8 consecutive
// K elements = two float4), splits them in registers into hi = bf16(x), lo = bf16(x - hi) and issues the three products
// hi*hi + hi*lo + lo*hi of the 3 x bf16 form (f32_path.h) into one fp32 accumulator: no operand staging in LDS, no split-K slabs and
// no finalize launch (conv_x3_kernel + finalize_f32_kernel: ~11 + 5 us per layer in the graph at 12^3 / 6^3, most of it set-up and the
// slab round trip).  Measured (fp32 mode of the headline step, same box): 226.6 -> 232.5 steps/s with it on the 29 launches of
// M <= 4096 rows; the three 24^3 skips (13824 x 512 -> 256) stay on conv_x3_kernel, whose LDS tile splits each operand once per
// tile instead of once per wave (61 vs 68 us).  The split is the cost here: ~350 VALU per K step per wave against 48 MFMAs.
// Same role as gemm_light_kernel for the bf16 plans (the nn.Linear layers of SABlock; the 1x1
// nin_shortcut of the reference's ResBlock).
#pragma once
#include "common.h"

struct LightX3Params {
    const float* x; const float* xb; int ca;   // [M][ca] (| xb [M][K - ca], channel-concatenated; ca = K when xb is null)
    const float* w;                            // [CoutPad][K] fp32
    const float* bias;                         // [CoutPad] or null
    const float* residual;                     // [M][CoutS] or null
    float* out;                                // [M][CoutS]
    float* stats;                              // GroupNorm partials of the output: [mtile][CoutS][2] (sum, sum of squares) or null
    int M, K, CoutS, mtiles;
};

// 8 fp32 -> (hi, lo) bf16x8 fragments
__device__ __forceinline__ void lx3_split(const float4 a, const float4 b, bf16x8& hi, bf16x8& lo) {
    const float x[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    u32x4 h, l;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const unsigned h0 = f2bf(x[2 * q]), h1 = f2bf(x[2 * q + 1]);
        h[q] = h0 | (h1 << 16);
        l[q] = pack2bf(x[2 * q] - __uint_as_float(h0 << 16), x[2 * q + 1] - __uint_as_float(h1 << 16));
    }
    hi = __builtin_bit_cast(bf16x8, h); lo = __builtin_bit_cast(bf16x8, l);
}

// Workgroup = KW waves that split the K range of one (16 MT) rows x (16 NT) couts tile between them (these launches are a few hundred
// tiles of 8 - 32 K steps: a single wave per tile walks its steps as a chain of L2 / HBM round trips, ~1 us per step measured; KW waves
// with DEPTH steps of raw fp32 fragments in flight each cover the whole range in one or two round trips).  Waves 1 .. KW-1 leave their
// accumulators in LDS, wave 0 adds them in wave order (fixed order: reproducible) and runs the epilogue.
template <int MT, int NT, int KW, int DEPTH>
__global__ __launch_bounds__(64 * KW) void gemm_light_x3_kernel(const LightX3Params p) {
    __shared__ float xch[KW > 1 ? (KW - 1) * NT * MT * 4 * 64 : 1];
    const int lane = threadIdx.x & 63, kg = threadIdx.x >> 6, fr = lane & 15, fg = lane >> 4;
    const int mtile = blockIdx.x % p.mtiles, ntile = blockIdx.x / p.mtiles;       // mtile fastest: neighbours share the weight rows
    const int m0 = mtile * 16 * MT, n0 = ntile * 16 * NT;
    // MFMA row i of cout tile nt <-> cout n0 + 4 NT (i >> 2) + 4 nt + (i & 3): after the MFMA a lane owns 4 NT consecutive couts
    const float* wrow[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) wrow[nt] = p.w + (size_t)(n0 + 4 * NT * (fr >> 2) + 4 * nt + (fr & 3)) * p.K + 8 * fg;
    const float* xrow[MT]; const float* xbrow[MT];
    const int cb2 = p.K - p.ca;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        int m = m0 + 16 * mt + fr; if (m >= p.M) m = p.M - 1;                     // clamped: rows past M are computed and dropped
        xrow[mt] = p.x + (size_t)m * p.ca + 8 * fg;
        xbrow[mt] = p.xb ? p.xb + (size_t)m * cb2 + 8 * fg - p.ca : xrow[mt];    // indexed with the global k
    }
    f32x4 acc[NT][MT];
#pragma unroll
    for (int a = 0; a < NT; ++a)
#pragma unroll
        for (int b = 0; b < MT; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int nsteps = p.K / 32, per = (nsteps + KW - 1) / KW;
    const int s0 = kg * per;
    int s1 = s0 + per; if (s1 > nsteps) s1 = nsteps;
    float4 wf[DEPTH][NT][2], xf[DEPTH][MT][2];                                    // DEPTH K steps of raw fragments in flight
#define LX_LOAD(D, S) do {                                                                            \
        int s_ = (S); if (s_ >= nsteps) s_ = nsteps - 1;              /* unconditional (clamped): no wait merges */ \
        const int k_ = s_ * 32;                                                                       \
        _Pragma("unroll") for (int nt = 0; nt < NT; ++nt) {                                           \
            wf[D][nt][0] = *reinterpret_cast<const float4*>(wrow[nt] + k_); wf[D][nt][1] = *reinterpret_cast<const float4*>(wrow[nt] + k_ + 4); } \
        const bool sec_ = k_ >= p.ca;                                 /* wave-uniform: the step lies in the second source */ \
        _Pragma("unroll") for (int mt = 0; mt < MT; ++mt) {                                           \
            const float* r_ = (sec_ ? xbrow[mt] : xrow[mt]) + k_;                                     \
            xf[D][mt][0] = *reinterpret_cast<const float4*>(r_); xf[D][mt][1] = *reinterpret_cast<const float4*>(r_ + 4); } \
    } while (0)
#define LX_MFMA(D) do {                                                                               \
        bf16x8 wh_[NT], wl_[NT];                                                                      \
        _Pragma("unroll") for (int nt = 0; nt < NT; ++nt) lx3_split(wf[D][nt][0], wf[D][nt][1], wh_[nt], wl_[nt]); \
        _Pragma("unroll") for (int mt = 0; mt < MT; ++mt) {                                           \
            bf16x8 xh_, xl_; lx3_split(xf[D][mt][0], xf[D][mt][1], xh_, xl_);                         \
            _Pragma("unroll") for (int nt = 0; nt < NT; ++nt) acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl_[nt], xh_, acc[nt][mt], 0, 0, 0); \
            _Pragma("unroll") for (int nt = 0; nt < NT; ++nt) acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh_[nt], xl_, acc[nt][mt], 0, 0, 0); \
            _Pragma("unroll") for (int nt = 0; nt < NT; ++nt) acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh_[nt], xh_, acc[nt][mt], 0, 0, 0); \
        }                                                                                             \
    } while (0)
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) LX_LOAD(d, s0 + d);
    for (int s = s0; s < s1; s += DEPTH) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            if (s + d < s1) LX_MFMA(d);
            LX_LOAD(d, s + DEPTH + d);
        }
    }
#undef LX_LOAD
#undef LX_MFMA
    if constexpr (KW > 1) {
        if (kg > 0) {
#pragma unroll
            for (int a = 0; a < NT; ++a)
#pragma unroll
                for (int b = 0; b < MT; ++b)
#pragma unroll
                    for (int r = 0; r < 4; ++r) xch[(((kg - 1) * NT * MT + a * MT + b) * 4 + r) * 64 + lane] = acc[a][b][r];
        }
        __syncthreads();
        if (kg > 0) return;
#pragma unroll
        for (int g = 0; g < KW - 1; ++g)
#pragma unroll
            for (int a = 0; a < NT; ++a)
#pragma unroll
                for (int b = 0; b < MT; ++b)
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[a][b][r] += xch[((g * NT * MT + a * MT + b) * 4 + r) * 64 + lane];
    }

    // ---- epilogue: lane = voxel fr of each 16-row tile, couts cb .. cb + 4 NT - 1 --------------------------------
    constexpr int NC = 4 * NT;
    const int cb = n0 + NC * fg;
    const bool do_stats = p.stats != nullptr;
    float ssum[NC], ssq[NC];
#pragma unroll
    for (int q = 0; q < NC; ++q) { ssum[q] = 0.f; ssq[q] = 0.f; }
    float bv[NC];
#pragma unroll
    for (int q = 0; q < NC; ++q) bv[q] = p.bias ? p.bias[cb + q] : 0.f;
    if (cb < p.CoutS) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int m = m0 + 16 * mt + fr;
            if (m >= p.M) continue;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                float4 v = make_float4(acc[nt][mt][0] + bv[4 * nt], acc[nt][mt][1] + bv[4 * nt + 1], acc[nt][mt][2] + bv[4 * nt + 2], acc[nt][mt][3] + bv[4 * nt + 3]);
                if (p.residual) {
                    const float4 r = *reinterpret_cast<const float4*>(p.residual + (size_t)m * p.CoutS + cb + 4 * nt);
                    v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
                }
                *reinterpret_cast<float4*>(p.out + (size_t)m * p.CoutS + cb + 4 * nt) = v;
                ssum[4 * nt] += v.x; ssq[4 * nt] += v.x * v.x; ssum[4 * nt + 1] += v.y; ssq[4 * nt + 1] += v.y * v.y;
                ssum[4 * nt + 2] += v.z; ssq[4 * nt + 2] += v.z * v.z; ssum[4 * nt + 3] += v.w; ssq[4 * nt + 3] += v.w * v.w;
            }
        }
    }
    if (do_stats) {
        // sum over the 16 voxel lanes of each DPP row (lanes sharing fg): rotate-and-add within the row
#define LX_ROW_ADD(X, CTRL) X += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(X), CTRL, 0xf, 0xf, true))
#pragma unroll
        for (int q = 0; q < NC; ++q) {
            LX_ROW_ADD(ssum[q], 0x128); LX_ROW_ADD(ssum[q], 0x124); LX_ROW_ADD(ssum[q], 0x122); LX_ROW_ADD(ssum[q], 0x121);
            LX_ROW_ADD(ssq[q], 0x128); LX_ROW_ADD(ssq[q], 0x124); LX_ROW_ADD(ssq[q], 0x122); LX_ROW_ADD(ssq[q], 0x121);
        }
#undef LX_ROW_ADD
        if (fr == 0 && cb < p.CoutS) {
            float* d = p.stats + ((size_t)mtile * p.CoutS + cb) * 2;
#pragma unroll
            for (int q = 0; q < NC / 2; ++q)
                *reinterpret_cast<float4*>(d + 4 * q) = make_float4(ssum[2 * q], ssq[2 * q], ssum[2 * q + 1], ssq[2 * q + 1]);
        }
    }
}

// tile choice: 64 x 64 per wave once that still gives every CU a wave (LDM_LIGHT_X3_BIG_MIN tiles), else 32 x 32
static inline int gemm_light_x3_big(long M, int cout_pad) {
    static const int thr = ldm_xknob("LDM_LIGHT_X3_BIG_MIN", 256);
    return cout_pad % 64 == 0 && ((M + 63) / 64) * (cout_pad / 64) >= thr;
}
static inline hipError_t launch_gemm_light_x3(const LightX3Params& p0, int cout_pad, int big, hipStream_t s) {
    LightX3Params p = p0;
    if (big) {
        p.mtiles = (p.M + 63) / 64;
        hipLaunchKernelGGL((gemm_light_x3_kernel<4, 4, 4, 2>), dim3(p.mtiles * (cout_pad / 64)), dim3(256), 0, s, p);
    } else {
        p.mtiles = (p.M + 31) / 32;
        hipLaunchKernelGGL((gemm_light_x3_kernel<2, 2, 4, 4>), dim3(p.mtiles * (cout_pad / 32)), dim3(256), 0, s, p);
    }
    return hipGetLastError();
}
