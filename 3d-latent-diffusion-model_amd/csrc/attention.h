// Fused 3D self-attention over D*H*W tokens (flash-style, bf16 MFMA 16x16x32, online softmax).
//
// Replaces monai SABlock's unfused  softmax(q k^T d^-1/2) v  (reached from
// 3d_ldm/inference.py:94-99 / 3d_ldm/train_diffusion.py:197-205 through DiffusionModelUNet) without ever
// materialising the h x N x N score matrix.  head_dim is fixed at 64 (num_head_channels of every shipped
// config: 3d_ldm/config/config_train_16g.json:46).
//
// Input  qkv : [B*N][3C] bf16 (q | k | v per token, channel = head*64 + d)  - output of the fused 1x1 projection
// Output o   : [B*N][C]  bf16
// One workgroup = 4 waves = 64 query rows of one (batch, head); each wave owns 16 query rows.
//   S^T = K Q^T        : MFMA A = K tile rows (LDS, swizzled ds_read_b128), B = Q fragments (registers)
//                        -> a lane holds 16 scores of ONE query (col = lane&15): row max/sum need 2 shuffles.
//   O^T += V^T P^T     : P stays in registers as the B operand (k order permuted identically on both operands),
//                        A = V^T rows read from a transposed LDS image (ds_read_b64, padded rows).
#pragma once
#include "common.h"

struct AttnParams {
    const bf16_t* qkv; bf16_t* out;
    int B, N, C, heads; float scale;
};

__global__ __launch_bounds__(256) void attn_fwd_kernel(const AttnParams p) {
    constexpr int D = 64, KT = 64;
    constexpr int VROW = 136;                         // bytes per V^T row (64 keys * 2 B + 8 B pad: conflict-free b64 reads)
    __shared__ __attribute__((aligned(16))) char smem[KT * 128 + D * VROW];
    char* ks = smem;                                  // K tile  [64 keys][64 d] bf16, 16-B chunks XOR-swizzled
    char* vs = smem + KT * 128;                       // V^T tile [64 d][64 keys] bf16

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, fg = lane >> 4;
    const int b = blockIdx.z, head = blockIdx.y;
    const int q0 = blockIdx.x * 64 + wave * 16;
    const int ld = 3 * p.C;                           // token stride (elements)
    const bf16_t* base = p.qkv + (size_t)b * p.N * ld;
    const int hoff = head * D;

    // Q fragments (B operand): lane holds Q[q0 + fr][ks*32 + 8*fg .. +7]
    bf16x8 qf[2];
    {
        int qr = q0 + fr; if (qr >= p.N) qr = p.N - 1;
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2)
            qf[k2] = *reinterpret_cast<const bf16x8*>(base + (size_t)qr * ld + hoff + k2 * 32 + 8 * fg);
    }

    f32x4 ot[4];                                      // O^T: ot[dt][r] = O[q = fr][d = dt*16 + 4*fg + r]
#pragma unroll
    for (int i = 0; i < 4; ++i) ot[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float mrun = -INFINITY, lrun = 0.f;               // running max / per-lane partial row sum

    const int ntile = (p.N + KT - 1) / KT;
    for (int t = 0; t < ntile; ++t) {
        const int k0 = t * KT;
        __syncthreads();                              // previous tile fully consumed
        // ---- stage K (row-major, swizzled) and V^T ------------------------------------------------
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int idx = tid + it * 256;           // 512 (row, chunk) pairs
            const int row = idx >> 3, ch = idx & 7;
            int kr = k0 + row; if (kr >= p.N) kr = p.N - 1;
            const bf16_t* tok = base + (size_t)kr * ld + hoff + ch * 8;
            const u32x4 kv = *reinterpret_cast<const u32x4*>(tok + p.C);
            const u32x4 vv = *reinterpret_cast<const u32x4*>(tok + 2 * p.C);
            *reinterpret_cast<u32x4*>(ks + row * 128 + ((ch ^ ((row >> 1) & 7)) << 4)) = kv;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int d = ch * 8 + 2 * e;
                *reinterpret_cast<bf16_t*>(vs + d * VROW + row * 2) = (bf16_t)(vv[e] & 0xffff);
                *reinterpret_cast<bf16_t*>(vs + (d + 1) * VROW + row * 2) = (bf16_t)(vv[e] >> 16);
            }
        }
        __syncthreads();

        // ---- S^T = K Q^T : st[j][r] = S[q = fr][key = k0 + 16 j + 4 fg + r] ------------------------
        f32x4 st[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            st[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
            const int row = j * 16 + fr;
#pragma unroll
            for (int k2 = 0; k2 < 2; ++k2) {
                const int c = k2 * 4 + fg;
                const bf16x8 kf = *reinterpret_cast<const bf16x8*>(ks + row * 128 + ((c ^ ((row >> 1) & 7)) << 4));
                st[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[k2], st[j], 0, 0, 0);
            }
        }
        // ---- online softmax ----------------------------------------------------------------------
        float tmax = -INFINITY;
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int key = k0 + 16 * j + 4 * fg + r;
                const float s = (key < p.N) ? st[j][r] * p.scale : -INFINITY;
                st[j][r] = s;
                tmax = fmaxf(tmax, s);
            }
        tmax = fmaxf(tmax, __shfl_xor(tmax, 16, 64));
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
        const float mnew = fmaxf(mrun, tmax);          // finite: every tile holds >= 1 valid key
        const float alpha = __expf(mrun - mnew);
        mrun = mnew;
        float psum = 0.f;
        bf16x8 pf[2];                                  // P^T fragments: k-slot (fg, e): e<4 -> sub-tile 2h, e>=4 -> 2h+1
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            float pv[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float pe = __expf(st[2 * h + (e >> 2)][e & 3] - mnew);
                psum += pe;
                pv[e] = pe;
            }
            u32x4 pk;
#pragma unroll
            for (int e = 0; e < 4; ++e) pk[e] = pack2bf(pv[2 * e], pv[2 * e + 1]);
            pf[h] = *reinterpret_cast<bf16x8*>(&pk);
        }
        lrun = lrun * alpha + psum;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
#pragma unroll
            for (int r = 0; r < 4; ++r) ot[dt][r] *= alpha;
        }
        // ---- O^T += V^T P^T ------------------------------------------------------------------------
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            const char* vrow = vs + (dt * 16 + fr) * VROW;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const u32x2 lo = *reinterpret_cast<const u32x2*>(vrow + ((2 * h) * 16 + 4 * fg) * 2);
                const u32x2 hi = *reinterpret_cast<const u32x2*>(vrow + ((2 * h + 1) * 16 + 4 * fg) * 2);
                u32x4 vv = {lo[0], lo[1], hi[0], hi[1]};
                const bf16x8 vf = *reinterpret_cast<bf16x8*>(&vv);
                ot[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf[h], ot[dt], 0, 0, 0);
            }
        }
    }
    // ---- normalise and store ------------------------------------------------------------------------
    lrun += __shfl_xor(lrun, 16, 64);
    lrun += __shfl_xor(lrun, 32, 64);
    const float inv = 1.0f / lrun;
    const int qr = q0 + fr;
    if (qr < p.N) {
        bf16_t* orow = p.out + ((size_t)b * p.N + qr) * p.C + hoff;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            u32x2 o;
            o[0] = pack2bf(ot[dt][0] * inv, ot[dt][1] * inv);
            o[1] = pack2bf(ot[dt][2] * inv, ot[dt][3] * inv);
            *reinterpret_cast<u32x2*>(orow + dt * 16 + 4 * fg) = o;
        }
    }
}
