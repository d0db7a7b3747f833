// 1x1x1 convolutions with short K (the attention q|k|v and output projections: M = D*H*W tokens, K = C <= 1024): a GEMM
//   out[m][co] = sum_k x[m][k] * w[co][k] + bias[co] (+ residual[m][co])
// whose whole K range is a handful of MFMA steps.  The implicit-GEMM conv kernel spends most of such a launch in set-up
// (tap table, LDS ring fill, accumulator exchange, LDS-folded epilogue: 7-9 us in-kernel for 4-8 K steps, tools/stamp_conv1.py);
// here both operands are K-contiguous rows (NDHWC activations, [cout][cin] weights), which IS the MFMA 16x16x32 operand
// layout, so every lane loads its fragments straight from global memory (L2) into registers: no LDS, no barrier, one wave
// per workgroup so that even M = 216 spreads over ~100 CUs.  Replaces the reference's q/k/v/out_proj nn.Linear layers of
// SABlock.
#pragma once
#include "common.h"

struct LightParams {
    const bf16_t* x; const bf16_t* w;     // [M][ca] (| xb [M][K - ca], channel-concatenated), [CoutPad][K]
    const bf16_t* xb; int ca;             // second source or null (ca = K)
    const float* bias;                    // [CoutPad] or null
    const bf16_t* residual;               // [M][CoutS] or null
    bf16_t* out;                          // [M][CoutS]
    float* stats;                         // GroupNorm partials of the output: [mtile][CoutS][2] or null
    int M, K, CoutS, mtiles;
};

// wave tile = (16 MT) rows x (16 NT) couts; CH = 32-deep K steps per register chunk (two chunks in flight).
template <int MT, int NT, int CH>
__global__ __launch_bounds__(64) void gemm_light_kernel(const LightParams p) {
    KSTAMP_BEGIN(5);
    const int lane = threadIdx.x, fr = lane & 15, fg = lane >> 4;
    const int mtile = blockIdx.x % p.mtiles, ntile = blockIdx.x / p.mtiles;       // mtile fastest: neighbours share the weight rows
    const int m0 = mtile * 16 * MT, n0 = ntile * 16 * NT;
    // MFMA row i of cout tile nt <-> cout n0 + 4 NT (i >> 2) + 4 nt + (i & 3): after the MFMA a lane owns 4 NT consecutive couts
    const bf16_t* wrow[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) wrow[nt] = p.w + (size_t)(n0 + 4 * NT * (fr >> 2) + 4 * nt + (fr & 3)) * p.K + 8 * fg;
    const bf16_t* xrow[MT]; const bf16_t* xbrow[MT];
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

    bf16x8 wa[CH][NT], xa[CH][MT], wb[CH][NT], xb[CH][MT];                       // two chunks of fragments
    const int nchunks = p.K / (32 * CH);
#define GL_LOAD(WF, XF, C) do {                                                                       \
        { int c_ = (C); if (c_ >= nchunks) c_ = nchunks - 1;          /* unconditional (clamped): no wait merges */ \
          _Pragma("unroll") for (int s = 0; s < CH; ++s) {                                            \
              const int k_ = (c_ * CH + s) * 32;                                                      \
              _Pragma("unroll") for (int nt = 0; nt < NT; ++nt) WF[s][nt] = *reinterpret_cast<const bf16x8*>(wrow[nt] + k_); \
              const bool sec_ = k_ >= p.ca;                           /* wave-uniform: the step lies in the second source */ \
              _Pragma("unroll") for (int mt = 0; mt < MT; ++mt) XF[s][mt] = *reinterpret_cast<const bf16x8*>((sec_ ? xbrow[mt] : xrow[mt]) + k_); \
          } }                                                                                         \
    } while (0)
#define GL_MFMA(WF, XF) do {                                                                          \
        _Pragma("unroll") for (int s = 0; s < CH; ++s)                                                \
            _Pragma("unroll") for (int nt = 0; nt < NT; ++nt)                                         \
                _Pragma("unroll") for (int mt = 0; mt < MT; ++mt)                                     \
                    acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(WF[s][nt], XF[s][mt], acc[nt][mt], 0, 0, 0); \
    } while (0)
    GL_LOAD(wa, xa, 0);
    GL_LOAD(wb, xb, 1);
    for (int c = 0; c < nchunks; c += 2) {
        GL_MFMA(wa, xa);
        GL_LOAD(wa, xa, c + 2);
        if (c + 1 < nchunks) GL_MFMA(wb, xb);
        GL_LOAD(wb, xb, c + 3);
    }
#undef GL_LOAD
#undef GL_MFMA
    KSTAMP(1);

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
            float v[NC];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int r = 0; r < 4; ++r) v[4 * nt + r] = acc[nt][mt][r] + bv[4 * nt + r];
            if (p.residual) {
#pragma unroll
                for (int h = 0; h < NC / 8; ++h) {
                    const u32x4 rv = *reinterpret_cast<const u32x4*>(p.residual + (size_t)m * p.CoutS + cb + 8 * h);
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        v[h * 8 + 2 * q] += __uint_as_float(rv[q] << 16);
                        v[h * 8 + 2 * q + 1] += __uint_as_float(rv[q] & 0xffff0000u);
                    }
                }
            }
#pragma unroll
            for (int h = 0; h < NC / 8; ++h) {
                u32x4 o;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    o[q] = pack2bf(v[h * 8 + 2 * q], v[h * 8 + 2 * q + 1]);
                    const float lo = __uint_as_float(o[q] << 16), hi = __uint_as_float(o[q] & 0xffff0000u);
                    ssum[h * 8 + 2 * q] += lo; ssq[h * 8 + 2 * q] += lo * lo;
                    ssum[h * 8 + 2 * q + 1] += hi; ssq[h * 8 + 2 * q + 1] += hi * hi;
                }
                *reinterpret_cast<u32x4*>(p.out + (size_t)m * p.CoutS + cb + 8 * h) = o;
            }
        }
    }
    if (do_stats) {
        // sum over the 16 voxel lanes of each DPP row (lanes sharing fg): rotate-and-add within the row
#define GL_ROW_ADD(X, CTRL) X += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(X), CTRL, 0xf, 0xf, true))
#pragma unroll
        for (int q = 0; q < NC; ++q) {
            GL_ROW_ADD(ssum[q], 0x128); GL_ROW_ADD(ssum[q], 0x124); GL_ROW_ADD(ssum[q], 0x122); GL_ROW_ADD(ssum[q], 0x121);
            GL_ROW_ADD(ssq[q], 0x128); GL_ROW_ADD(ssq[q], 0x124); GL_ROW_ADD(ssq[q], 0x122); GL_ROW_ADD(ssq[q], 0x121);
        }
#undef GL_ROW_ADD
        if (fr == 0 && cb < p.CoutS) {
            float* d = p.stats + ((size_t)mtile * p.CoutS + cb) * 2;
#pragma unroll
            for (int q = 0; q < NC / 2; ++q)
                *reinterpret_cast<float4*>(d + 4 * q) = make_float4(ssum[2 * q], ssq[2 * q], ssum[2 * q + 1], ssq[2 * q + 1]);
        }
    }
    KSTAMP(2);
    KSTAMP_DRAIN(3);
}

// tile choice: 64 x 64 per wave once that still gives every CU a wave, else 32 x 32
static inline hipError_t launch_gemm_light(const LightParams& p0, int cout_pad, int big, hipStream_t s) {
    LightParams p = p0;
    const bool k128 = p.K % 128 == 0;               // else K = 32 / 64 / 96 (im2col first convs): one K step per register chunk
    if (big) {
        p.mtiles = (p.M + 63) / 64;
        if (k128) hipLaunchKernelGGL((gemm_light_kernel<4, 4, 2>), dim3(p.mtiles * (cout_pad / 64)), dim3(64), 0, s, p);
        else hipLaunchKernelGGL((gemm_light_kernel<4, 4, 1>), dim3(p.mtiles * (cout_pad / 64)), dim3(64), 0, s, p);
    } else {
        p.mtiles = (p.M + 31) / 32;
        if (k128) hipLaunchKernelGGL((gemm_light_kernel<2, 2, 4>), dim3(p.mtiles * (cout_pad / 32)), dim3(64), 0, s, p);
        else hipLaunchKernelGGL((gemm_light_kernel<2, 2, 1>), dim3(p.mtiles * (cout_pad / 32)), dim3(64), 0, s, p);
    }
    return hipGetLastError();
}
static inline int gemm_light_big(long M, int cout_pad) {
    static const int thr = ldm_xknob("LDM_LIGHT_BIG_MIN", 256);   // tuning knob (tools/stamp_conv1.py)
    return ((M + 63) / 64) * (cout_pad / 64) >= thr;
}
static inline bool gemm_light_ok(int k, int stride, int ups, int cin, bool single_source, bool plain_epilogue) {
    return k == 1 && stride == 1 && ups == 0 && single_source && plain_epilogue && cin <= 2048 &&
           (cin % 128 == 0 || (cin % 32 == 0 && cin < 128));
}
