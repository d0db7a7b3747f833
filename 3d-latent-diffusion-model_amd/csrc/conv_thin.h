// 3x3x3 stride-1 "same" convolution with a THIN output (Cout <= 4): the networks' last layers (AutoencoderKL decoder 64 -> 1 at full
// resolution, 3d_ldm/inference.py:94-99 via decode; DiffusionModelUNet out conv 256 -> 4, 3d_ldm/train_diffusion.py:197-205).
//
// conv3_halo_kernel computes 64 couts per tile whatever Cout is: 254 x 64 tiles for ONE real channel are 98 % wasted matrix work and, worse,
// wasted LDS traffic (the K loop is LDS bound).  Here a workgroup owns a 4 x 4 x 16 block of output voxels, copies its 6 x 6 x 18 input halo
// (32 channels at a time: 48 KB of LDS, three workgroups per CU so that one's copy phase sits under another's multiply) into LDS ONCE and reads every voxel row 27 times from there; the weights of all 27 taps (4 rows of 64 channels) sit
// next to it.  MFMA 16x16x32 with the couts on the A side (rows 4.. of the 16 are zero registers, never read from LDS), a wave = one d-slice
// of the block = 4 voxel tiles of 16 consecutive w.  Per 32 channels and wave: 27 x (1 weight + 4 voxel fragment reads, 4 MFMAs).
// Output: fp32 NCDHW (+ bias), the layout the callers take.  bf16 operands, fp32 accumulation: the arithmetic of the kernels it replaces
// (summation order over K differs).  Cin % 32 == 0, any D / H / W (ragged blocks are bounds-checked).
#pragma once
#include "common.h"

struct ThinParams {
    const bf16_t* x; const bf16_t* w;      // x [N][D][H][W][Cin]; w packed [27][CoutPad][Cin] (rows 0 .. CoutReal-1 are read)
    const float* bias; float* out;         // bias [>= CoutReal] or null; out fp32 [N][CoutReal][D*H*W]
    int N, D, H, W, Cin, CoutPad, CoutReal;
    int td, th, tw;                        // blocks per dimension
    // fp32 precision mode (3 x bf16 product): x = the (hi | lo) split [..][2 C] bf16, w = [27][CoutPad][hi | lo | hi] (Cin = 3 C here);
    // channel chunk c of the weights meets chunk c (hi), c - n (hi again) or c - n (lo) of a voxel row, n = x3_c / 32 (conv_halo.h's mapping)
    int x3_c;                              // 0, or the real channel count C
};

constexpr int THIN_TD = 4, THIN_TH = 4, THIN_TW = 16;
constexpr int THIN_HD = THIN_TD + 2, THIN_HH = THIN_TH + 2, THIN_HW = THIN_TW + 2;
constexpr int THIN_HV = THIN_HD * THIN_HH * THIN_HW;                 // 648 halo voxels
constexpr int THIN_CK = 32, THIN_RB = THIN_CK * 2, THIN_NC = THIN_RB / 16;   // channels per LDS chunk (one MFMA k step), bytes per row, 16-byte chunks per row
constexpr int THIN_LDS_X = THIN_HV * THIN_RB, THIN_LDS_W = 27 * 4 * THIN_RB;   // one chunk of the halo block (41 KB); of the weights (7 KB): three workgroups per CU
constexpr int THIN_LDS = THIN_LDS_X + THIN_LDS_W;

__global__ __launch_bounds__(256) void conv3_thin_kernel(const ThinParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* sx = smem; char* sw = smem + THIN_LDS_X;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 15, fg = lane >> 4;
    int b = blockIdx.x;
    const int bw = b % p.tw; b /= p.tw; const int bh = b % p.th; b /= p.th; const int bd = b % p.td; const int n = b / p.td;
    const int d0 = bd * THIN_TD, h0 = bh * THIN_TH, w0 = bw * THIN_TW;
    const size_t xn = (size_t)n * p.D * p.H * p.W;
    const int xstride = p.x3_c ? 2 * p.x3_c : p.Cin;    // channels per voxel row of x
    // MFMA column fr <-> voxel w0 + wm: even voxels on columns 0-3 / 12-15, odd ones on 4-11, so that the lane groups of ds_read_b128
    // ({0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, +32) read conflict-free under the (hv >> 2) & 3 slot swizzle (conv_block.h)
    const int wm = fr < 4 ? 2 * fr : fr < 12 ? 2 * (fr - 4) + 1 : 2 * (fr - 8);
    f32x4 acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const u32x4 zero = {0u, 0u, 0u, 0u};
    for (int c0 = 0; c0 < p.Cin; c0 += THIN_CK) {
        __syncthreads();                                  // the previous chunk's reads are done
        // halo block: voxel hv, 16-byte chunk c16 (XOR-swizzled by the voxel index).  All of a lane's loads are issued before the first LDS
        // store (a load -> store -> load chain costs one global round trip per item: 21 of them)
        const int xc0 = (p.x3_c && c0 >= p.x3_c) ? c0 - p.x3_c : c0;       // channel offset of this chunk inside a voxel row
        constexpr int NIT = (THIN_HV * THIN_NC + 255) / 256;
        u32x4 hv_v[NIT];
#pragma unroll
        for (int k = 0; k < NIT; ++k) {
            const int e = tid + k * 256;
            const int hv = e / THIN_NC, c16 = e % THIN_NC;
            const int wx = hv % THIN_HW, r = hv / THIN_HW, hy = r % THIN_HH, dz = r / THIN_HH;
            const int gd = d0 - 1 + dz, gh = h0 - 1 + hy, gw = w0 - 1 + wx;
            hv_v[k] = zero;
            if (e < THIN_HV * THIN_NC && (unsigned)gd < (unsigned)p.D && (unsigned)gh < (unsigned)p.H && (unsigned)gw < (unsigned)p.W)
                hv_v[k] = *reinterpret_cast<const u32x4*>(p.x + (xn + ((size_t)gd * p.H + gh) * p.W + gw) * xstride + xc0 + c16 * 8);
        }
#pragma unroll
        for (int k = 0; k < NIT; ++k) {
            const int e = tid + k * 256;
            const int hv = e / THIN_NC, c16 = e % THIN_NC;
            if (e < THIN_HV * THIN_NC) *reinterpret_cast<u32x4*>(sx + hv * THIN_RB + ((c16 ^ ((hv >> 2) & (THIN_NC - 1))) << 4)) = hv_v[k];
        }
        for (int e = tid; e < 27 * 4 * THIN_NC; e += 256) {   // weights of the chunk: [tap][row 0..3][THIN_CK channels], rows >= CoutReal zero
            const int c16 = e % THIN_NC, row = (e / THIN_NC) & 3, tap = e / (4 * THIN_NC);
            u32x4 v = zero;
            if (row < p.CoutReal) v = *reinterpret_cast<const u32x4*>(p.w + ((size_t)tap * p.CoutPad + row) * p.Cin + c0 + c16 * 8);
            *reinterpret_cast<u32x4*>(sw + (tap * 4 + row) * THIN_RB + ((c16 ^ row) << 4)) = v;
        }
        __syncthreads();
#pragma unroll 1
        for (int tap = 0; tap < 27; ++tap) {
            const int kd = tap / 9, kh = (tap / 3) % 3, kw = tap % 3;
            u32x4 wv = zero;
            if (fr < 4) wv = *reinterpret_cast<const u32x4*>(sw + (tap * 4 + fr) * THIN_RB + ((fg ^ fr) << 4));
            const bf16x8 wf = *reinterpret_cast<const bf16x8*>(&wv);
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int hv = ((wave + kd) * THIN_HH + (t + kh)) * THIN_HW + wm + kw;
                const bf16x8 xf = *reinterpret_cast<const bf16x8*>(sx + hv * THIN_RB + ((fg ^ ((hv >> 2) & (THIN_NC - 1))) << 4));
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, xf, acc[t], 0, 0, 0);
            }
        }
    }
    // accumulator row 4 fg + r = cout, column fr = voxel: the real couts live in the fg == 0 lanes
    if (fg != 0) return;
    const int gd = d0 + wave, gw = w0 + wm;
    if (gd >= p.D || gw >= p.W) return;
    const size_t dhw = (size_t)p.D * p.H * p.W;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int gh = h0 + t;
        if (gh >= p.H) continue;
        const size_t sp = ((size_t)gd * p.H + gh) * p.W + gw;
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (r < p.CoutReal) p.out[((size_t)n * p.CoutReal + r) * dhw + sp] = acc[t][r] + (p.bias ? p.bias[r] : 0.f);
    }
#endif
}
