// Fused 3D self-attention over D*H*W tokens (flash-style, bf16 MFMA 16x16x32, online softmax).
//
// Replaces monai SABlock's unfused  softmax(q k^T d^-1/2) v  (reached from
// 3d_ldm/inference.py:94-99 / 3d_ldm/train_diffusion.py:197-205 through DiffusionModelUNet) without ever
// materialising the h x N x N score matrix.  head_dim D in {32, 64, 128, 256}: 64 is num_head_channels of the benchmark UNet
// (3d_ldm/config/config_train_16g.json:46), 32 that of config_train_stable.json:45-46, and the AutoencoderKL attention blocks are
// single-head with D = C in {64, 128, 256} (config_train_32g.json:21-25).  A head wider than 64 is walked as D / 64 SLABS of 64
// channels: one [64 keys][64 d] LDS image per slab, scores summed over the slabs, one O^T accumulator set per slab.
//
// Input  qkv : [B*N][3C] bf16 (q | k | v per token, channel = head*64 + d)  - output of the fused 1x1 projection
// Output o   : [B*N][C]  bf16
// One workgroup = 4 waves = 64 query rows of one (batch, head); each wave owns 16 query rows.
//   S^T = K Q^T        : MFMA A = K tile rows (LDS, swizzled ds_read_b128), B = Q fragments (registers)
//                        -> a lane holds 16 scores of ONE query (col = lane&15): row max/sum need 2 shuffles.
//   O^T += V^T P^T     : P stays in registers as the B operand (k order permuted identically on both operands),
//                        A = V^T fragments read from the row-major V image with ds_read_b64_tr_b16 (hardware transpose).
#pragma once
#include "common.h"

struct AttnParams {
    const bf16_t* qkv; bf16_t* out;
    int B, N, C, heads; float scale;
    float* lse;                       // optional [B][heads][N]: log-sum-exp of the scaled scores (saved for the backward pass)
    int d;                            // head dimension (C / heads): 32, 64, 128 or 256
};

// S > 1: the workgroup carries S groups of 4 waves; group g walks the key tiles g, g+S, g+2S, ... with its own LDS double
// buffer and the groups' (max, sum, O) partials are merged through LDS at the end (in-workgroup split of the key range):
// at N = 1728 the grid is only 27 x heads workgroups, so the extra waves are what hides the load -> LDS -> MFMA latency chain.
// QW = waves (16 query rows each) per group; DB = double-buffered K/V images (one barrier per tile) or a single image with
// two barriers per tile (S = 8 groups of 2 waves: 32 query rows per workgroup, twice the workgroups at N = 1728).
template <int S, int QW, bool DB, int D = 64>
__global__ __launch_bounds__(64 * QW * S) void attn_fwd_kernel(const AttnParams p) {
    constexpr int KT = 64;
    constexpr int NSL = (D + 63) / 64, DS = D < 64 ? D : 64;   // slabs of DS channels
    constexpr int K2 = DS / 32, NDT = DS / 16, CH = DS / 8;    // per slab: MFMA k-steps of Q K^T, 16-row tiles of O^T, 16-byte chunks per row
    constexpr int IMG = KT * 128;                     // one [64 keys][64 d] image (rows stay 128 bytes apart for D = 32 too)
    constexpr int TILE = 2 * NSL * IMG;               // NSL K images, then NSL V images (row-major [key][d], swizzled)
    constexpr int GT = 64 * QW;                       // threads per group
    constexpr int NIT = (KT * CH * NSL) / GT;         // staged (slab, row, chunk) triples per thread
    static_assert(S == 1 || D == 64, "the in-workgroup key split is built for D = 64");
    static_assert((KT * CH * NSL) % GT == 0, "staging must divide evenly");
    extern __shared__ __attribute__((aligned(16))) char smem_all[];   // S x (double) buffer

    const int tid = threadIdx.x % GT, lane = tid & 63, wave = tid >> 6, grp = threadIdx.x / GT;
    char* smem = smem_all + grp * (DB ? 2 : 1) * TILE;
    const int fr = lane & 15, fg = lane >> 4;
    const int b = blockIdx.z, head = blockIdx.y;
    const int q0 = blockIdx.x * (16 * QW) + wave * 16;
    const int ld = 3 * p.C;                           // token stride (elements)
    const bf16_t* base = p.qkv + (size_t)b * p.N * ld;
    const int hoff = head * D;

    // Q fragments (B operand): lane holds Q[q0 + fr][sl*64 + ks*32 + 8*fg .. +7]
    bf16x8 qf[NSL][K2];
    {
        int qr = q0 + fr; if (qr >= p.N) qr = p.N - 1;
#pragma unroll
        for (int sl = 0; sl < NSL; ++sl)
#pragma unroll
            for (int k2 = 0; k2 < K2; ++k2)
                qf[sl][k2] = *reinterpret_cast<const bf16x8*>(base + (size_t)qr * ld + hoff + sl * 64 + k2 * 32 + 8 * fg);
    }

    f32x4 ot[NSL][NDT];                               // O^T: ot[sl][dt][r] = O[q = fr][d = sl*64 + dt*16 + 4*fg + r]
#pragma unroll
    for (int sl = 0; sl < NSL; ++sl)
#pragma unroll
        for (int i = 0; i < NDT; ++i) ot[sl][i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float mrun = -INFINITY, lrun = 0.f;               // running max (raw score units) / per-lane partial row sum
    const float cexp = p.scale * 1.44269504088896340736f;     // scale * log2(e)

    // staging split (issue early / write late): the global loads of the group's next tile are issued before the MFMAs of
    // the current one and written to the other LDS image after them, so their latency hides under the compute; one barrier
    // per tile.  The prefetch is unconditional (rows clamped) so that no wait for it is needed before the tile's own MFMAs.
    u32x4 kreg[NIT], vreg[NIT];
    auto stage_load = [&](int k0) {
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int idx = tid + it * GT;            // (slab, row, chunk) triples
            const int sl = idx / (KT * CH), rem = idx - sl * (KT * CH);
            const int row = rem / CH, ch = rem - row * CH;
            int kr = k0 + row; if (kr >= p.N) kr = p.N - 1;
            const bf16_t* tok = base + (size_t)kr * ld + hoff + sl * 64 + ch * 8;
            kreg[it] = *reinterpret_cast<const u32x4*>(tok + p.C);
            vreg[it] = *reinterpret_cast<const u32x4*>(tok + 2 * p.C);
        }
    };
    auto stage_write = [&](char* ks, char* vs) {
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int idx = tid + it * GT;
            const int sl = idx / (KT * CH), rem = idx - sl * (KT * CH);
            const int row = rem / CH, ch = rem - row * CH;
            *reinterpret_cast<u32x4*>(ks + sl * IMG + row * 128 + ((ch ^ ((row >> 1) & 7)) << 4)) = kreg[it];
            // V stays row-major; its 32-byte blocks are XOR-swizzled by (row>>1)&3 so that the transposed reads
            // (4 rows x 16 columns per 16-lane group) are bank-conflict free
            *reinterpret_cast<u32x4*>(vs + sl * IMG + row * 128 + ((ch ^ (((row >> 1) & 3) << 1)) << 4)) = vreg[it];
        }
    };

    const int ntile = (p.N + KT - 1) / KT;
    const int nrounds = (ntile + S - 1) / S;          // every group runs the same number of rounds (barriers are workgroup-wide)
    stage_load(grp * KT);
#pragma unroll
    for (int sl = 0; sl < NSL; ++sl)
#pragma unroll
        for (int k2 = 0; k2 < K2; ++k2) asm volatile("" :: "v"(qf[sl][k2]));   // Q has landed before the loop: no load wait is left inside it
    stage_write(smem, smem + NSL * IMG);
    __syncthreads();
    for (int it = 0; it < nrounds; ++it) {
        const int t = it * S + grp;
        const int k0 = t * KT;
        const char* ks = smem + (DB ? (it & 1) * TILE : 0);   // K tiles [slab][64 keys][64 d] bf16, 16-B chunks XOR-swizzled
        const char* vs = ks + NSL * IMG;              // V tiles [slab][64 keys][64 d] bf16, read transposed (ds_read_b64_tr_b16)
        stage_load(k0 + S * KT);
        if (t < ntile) {                              // wave-uniform
        // ---- S^T = K Q^T : st[j][r] = S[q = fr][key = k0 + 16 j + 4 fg + r] ------------------------
        f32x4 st[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            st[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
            const int row = j * 16 + fr;
#pragma unroll
            for (int sl = 0; sl < NSL; ++sl)
#pragma unroll
                for (int k2 = 0; k2 < K2; ++k2) {
                    const int c = k2 * 4 + fg;
                    const bf16x8 kf = *reinterpret_cast<const bf16x8*>(ks + sl * IMG + row * 128 + ((c ^ ((row >> 1) & 7)) << 4));
                    st[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[sl][k2], st[j], 0, 0, 0);
                }
        }
        // ---- online softmax: the running max is kept in RAW score units (scale > 0 commutes with max) and the scale is folded
        //      into the exponent: p = exp2(s * c - m * c), c = scale * log2(e) -> one FMA + one v_exp per score.  Only the last
        //      (partial) tile masks keys beyond N. --------------------------------------------------------------------------
        if (k0 + KT > p.N) {                           // wave-uniform
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (k0 + 16 * j + 4 * fg + r >= p.N) st[j][r] = -INFINITY;
        }
        float tmax = fmaxf(fmaxf(fmaxf(st[0][0], st[0][1]), fmaxf(st[0][2], st[0][3])), fmaxf(fmaxf(st[1][0], st[1][1]), fmaxf(st[1][2], st[1][3])));
        tmax = fmaxf(tmax, fmaxf(fmaxf(fmaxf(st[2][0], st[2][1]), fmaxf(st[2][2], st[2][3])), fmaxf(fmaxf(st[3][0], st[3][1]), fmaxf(st[3][2], st[3][3]))));
        tmax = fmaxf(tmax, __shfl_xor(tmax, 16, 64));
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
        const float mnew = fmaxf(mrun, tmax);          // finite: every tile holds >= 1 valid key
        const float alpha = __builtin_amdgcn_exp2f((mrun - mnew) * cexp);
        const float mc = -mnew * cexp;
        mrun = mnew;
        float psum = 0.f;
        bf16x8 pf[2];                                  // P^T fragments: k-slot (fg, e): e<4 -> sub-tile 2h, e>=4 -> 2h+1
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            float pv[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float pe = __builtin_amdgcn_exp2f(fmaf(st[2 * h + (e >> 2)][e & 3], cexp, mc));
                psum += pe;
                pv[e] = pe;
            }
            u32x4 pk;
#pragma unroll
            for (int e = 0; e < 4; ++e) pk[e] = pack2bf(pv[2 * e], pv[2 * e + 1]);
            pf[h] = *reinterpret_cast<bf16x8*>(&pk);
        }
        lrun = lrun * alpha + psum;
        if (__builtin_amdgcn_ballot_w64(alpha != 1.0f)) {      // the max rarely moves after the first tiles: skip the rescale then
#pragma unroll
            for (int sl = 0; sl < NSL; ++sl)
#pragma unroll
                for (int dt = 0; dt < NDT; ++dt) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) ot[sl][dt][r] *= alpha;
                }
        }
        // ---- O^T += V^T P^T : the A operand V^T[d][key] comes from the row-major V image through the hardware
        //      transpose read: lane 4q+p of a 16-lane group addresses row (key) q, columns (d) 4p..4p+3 of a 4 x 16 block
        //      and lane i receives column i of the 4 rows (probed: tools/probe/ds_read_tr_probe.hip). ------------------
        typedef __attribute__((ext_vector_type(4))) short s16x4;
        typedef __attribute__((address_space(3))) s16x4* lds_s16x4_t;
        const int tq = fr >> 2, tp = fr & 3;
#pragma unroll
        for (int sl = 0; sl < NSL; ++sl)
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int rlo = (2 * h) * 16 + 4 * fg + tq, rhi = rlo + 16;       // key rows this lane addresses
                const char* alo = vs + sl * IMG + rlo * 128 + ((dt ^ ((rlo >> 1) & 3)) << 5) + tp * 8;
                const char* ahi = vs + sl * IMG + rhi * 128 + ((dt ^ ((rhi >> 1) & 3)) << 5) + tp * 8;
                const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t)alo);
                const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t)ahi);
                const bf16x8 vf = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                ot[sl][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf[h], ot[sl][dt], 0, 0, 0);
            }
        }
        }
        // ---- late write of the next tile into the other image (its last readers finished before the previous barrier) ----
        if constexpr (!DB) __syncthreads();            // single image: every reader of this tile is done
        {
            char* nk = smem + (DB ? ((it + 1) & 1) * TILE : 0);
            stage_write(nk, nk + NSL * IMG);
        }
        __syncthreads();
    }
    // ---- merge the S groups' partials (all staging images are dead after the last barrier) ----------------------
    lrun += __shfl_xor(lrun, 16, 64);
    lrun += __shfl_xor(lrun, 32, 64);
    if constexpr (S > 1) {
        // per (group, wave): O^T partial [4 dt][64 lanes] f32x4 = 4 KiB, then m and l per query (16 floats each)
        float* xo = reinterpret_cast<float*>(smem_all) + (size_t)(grp * QW + wave) * (1024 + 32);
        if (grp > 0) {
#pragma unroll
            for (int dt = 0; dt < NDT; ++dt) *reinterpret_cast<f32x4*>(xo + (dt * 64 + lane) * 4) = ot[0][dt];
            if (fg == 0) { xo[1024 + fr] = mrun; xo[1040 + fr] = lrun; }
        }
        __syncthreads();
        if (grp > 0) return;
#pragma unroll
        for (int g = 1; g < S; ++g) {                  // fixed order: reproducible
            const float* xg = reinterpret_cast<const float*>(smem_all) + (size_t)(g * QW + wave) * (1024 + 32);
            const float mg = xg[1024 + fr], lg = xg[1040 + fr];
            const float mnew = fmaxf(mrun, mg);        // group 0 always owns tile 0: finite
            const float a0 = __builtin_amdgcn_exp2f((mrun - mnew) * cexp), a1 = __builtin_amdgcn_exp2f((mg - mnew) * cexp);   // mg = -inf (group without a tile) -> 0
            lrun = lrun * a0 + lg * a1;
            mrun = mnew;
#pragma unroll
            for (int dt = 0; dt < NDT; ++dt) {
                const f32x4 og = *reinterpret_cast<const f32x4*>(xg + (dt * 64 + lane) * 4);
#pragma unroll
                for (int r = 0; r < 4; ++r) ot[0][dt][r] = ot[0][dt][r] * a0 + og[r] * a1;
            }
        }
    }
    // ---- normalise and store ------------------------------------------------------------------------
    const float inv = 1.0f / lrun;
    const int qr = q0 + fr;
    if (p.lse && qr < p.N && fg == 0) p.lse[((size_t)b * p.heads + head) * p.N + qr] = mrun * p.scale + __logf(lrun);
    if (qr < p.N) {
        bf16_t* orow = p.out + ((size_t)b * p.N + qr) * p.C + hoff;
#pragma unroll
        for (int sl = 0; sl < NSL; ++sl)
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt) {
            u32x2 o;
            o[0] = pack2bf(ot[sl][dt][0] * inv, ot[sl][dt][1] * inv);
            o[1] = pack2bf(ot[sl][dt][2] * inv, ot[sl][dt][3] * inv);
            *reinterpret_cast<u32x2*>(orow + sl * 64 + dt * 16 + 4 * fg) = o;
        }
    }
}

// 8 groups of 2 waves (32 query rows per workgroup) for long sequences, 4 groups of 4 waves once a (batch, head) has at least
// 4 key tiles, else the plain 4-wave kernel
template <int D>
static inline hipError_t launch_attn_fwd_d(const AttnParams& p, hipStream_t s) {
    constexpr int LDS = 2 * 2 * ((D + 63) / 64) * 64 * 128;          // double-buffered K + V images of every slab
    static bool once = false;
    if (!once) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_fwd_kernel<1, 4, true, D>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        if (e != hipSuccess) return e;
        once = true;
    }
    hipLaunchKernelGGL((attn_fwd_kernel<1, 4, true, D>), dim3((p.N + 63) / 64, p.heads, p.B), dim3(256), LDS, s, p);
    return hipGetLastError();
}
static inline hipError_t launch_attn_fwd(const AttnParams& p, hipStream_t s) {
    if (p.d == 32) return launch_attn_fwd_d<32>(p, s);
    if (p.d == 128) return launch_attn_fwd_d<128>(p, s);
    if (p.d == 256) return launch_attn_fwd_d<256>(p, s);
    if (p.d != 64) return hipErrorInvalidValue;
    static bool once = false;
    if (!once) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_fwd_kernel<4, 4, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 32768);
        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_fwd_kernel<8, 2, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 8 * 16384);
        if (e != hipSuccess) return e;
        once = true;
    }
    if (p.N >= 1024 && (long)((p.N + 63) / 64) * p.heads * p.B < 128)      // small grid: twice the workgroups beats the extra K/V traffic
        hipLaunchKernelGGL((attn_fwd_kernel<8, 2, false>), dim3((p.N + 31) / 32, p.heads, p.B), dim3(1024), 8 * 16384, s, p);
    else if (p.N > 3 * 64) hipLaunchKernelGGL((attn_fwd_kernel<4, 4, true>), dim3((p.N + 63) / 64, p.heads, p.B), dim3(1024), 4 * 32768, s, p);
    else hipLaunchKernelGGL((attn_fwd_kernel<1, 4, true>), dim3((p.N + 63) / 64, p.heads, p.B), dim3(256), 32768, s, p);
    return hipGetLastError();
}


// ================================================================================================ backward
// Given dO:  delta_i = sum_d dO_id O_id ;  P_ij = exp(scale q_i.k_j - LSE_i) ;  dV_j = sum_i P_ij dO_i ;
//            dP_ij = dO_i . V_j ;  dS_ij = P_ij (dP_ij - delta_i) ;  dQ_i = scale sum_j dS_ij K_j ;  dK_j = scale sum_i dS_ij Q_i.
// Two kernels so that every output row is owned by exactly one wave (no atomics, bitwise reproducible):
//   attn_bwd_dq_kernel  : workgroup = 64 query rows of one (batch, head), loops over key tiles   -> dQ
//   attn_bwd_dkv_kernel : workgroup = 64 key rows of one (batch, head), loops over query tiles   -> dK, dV
// Same MFMA idioms as the forward: scores are produced with the reduction-free index on the lane, P / dS go straight
// back in as B operands with the k order permuted identically on both operands, transposed operands come from
// row-major LDS images through ds_read_b64_tr_b16.
struct AttnBwdParams {
    const bf16_t* qkv; const bf16_t* o; const bf16_t* d_o; const float* lse;
    float* delta;                     // [B][heads][N] scratch
    bf16_t* dqkv;                     // [B*N][3C]  (dq | dk | dv)
    int B, N, C, heads; float scale;
    int d;                            // head dimension: 32, 64, 128 or 256
};

// delta[b][h][i] = sum_d dO[i][h*64+d] * O[i][h*64+d]; one wave per (token, head) pair of 64 channels... 16 lanes x 4 elems
__global__ __launch_bounds__(256) void attn_delta_kernel(const AttnBwdParams p) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;             // one thread per (token, head, 16-byte chunk of 8 d)
    const int cpd = p.d >> 3;                                          // chunks per head: 4, 8, 16 or 32 consecutive lanes
    const long total = (long)p.B * p.N * p.heads * cpd;
    float s = 0.f;
    long tok = 0; int head = 0;
    if (idx < total) {
        const int ch = (int)(idx % cpd);
        const long th = idx / cpd;
        head = (int)(th % p.heads); tok = th / p.heads;
        const u32x4 a = *reinterpret_cast<const u32x4*>(p.o + tok * p.C + head * p.d + ch * 8);
        const u32x4 b = *reinterpret_cast<const u32x4*>(p.d_o + tok * p.C + head * p.d + ch * 8);
#pragma unroll
        for (int k = 0; k < 4; ++k)
            s += __uint_as_float(a[k] << 16) * __uint_as_float(b[k] << 16) +
                 __uint_as_float(a[k] & 0xffff0000u) * __uint_as_float(b[k] & 0xffff0000u);
    }
    for (int o = 1; o < cpd; o <<= 1) s += __shfl_xor(s, o, 64);
    if (idx < total && (idx % cpd) == 0) {
        const long bb = tok / p.N, i = tok - bb * p.N;
        p.delta[((size_t)bb * p.heads + head) * p.N + i] = s;
    }
}

typedef __attribute__((ext_vector_type(4))) short attn_s16x4;
typedef __attribute__((address_space(3))) attn_s16x4* attn_lds_s16x4_t;

// LDS image helpers: 64 rows x 64 bf16 (128-byte rows).  ROW image: 16-byte chunks XOR (row>>1)&7 (ds_read_b128 row reads);
// TR image: 32-byte blocks XOR (row>>1)&3 (transposed reads).
__device__ __forceinline__ int attn_row_off(int row, int ch) { return row * 128 + ((ch ^ ((row >> 1) & 7)) << 4); }
__device__ __forceinline__ int attn_tr_off(int row, int ch) { return row * 128 + ((ch ^ (((row >> 1) & 3) << 1)) << 4); }
// fragment of the TRANSPOSED tile for MFMA rows dt*16.., k-slots of half h: lane (fr, fg) -> 8 values
__device__ __forceinline__ bf16x8 attn_tr_frag(const char* img, int dt, int h, int fr, int fg) {
    const int tq = fr >> 2, tp = fr & 3;
    const int rlo = (2 * h) * 16 + 4 * fg + tq, rhi = rlo + 16;
    const attn_s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((attn_lds_s16x4_t)(img + rlo * 128 + ((dt ^ ((rlo >> 1) & 3)) << 5) + tp * 8));
    const attn_s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((attn_lds_s16x4_t)(img + rhi * 128 + ((dt ^ ((rhi >> 1) & 3)) << 5) + tp * 8));
    return (bf16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}

template <int D>
__global__ __launch_bounds__(256) void attn_bwd_dq_kernel(const AttnBwdParams p) {
    constexpr int KT = 64;
    constexpr int NSL = (D + 63) / 64, DS = D < 64 ? D : 64, K2 = DS / 32, NDT = DS / 16, CH = DS / 8, IMG = KT * 128;
    constexpr int NIT = (KT * CH * NSL) / 256;
    extern __shared__ __attribute__((aligned(16))) char smem[];         // per slab: K row image, K tr image, V row image
    char* k_row = smem; char* k_tr = smem + NSL * IMG; char* v_row = smem + 2 * NSL * IMG;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, fg = lane >> 4;
    const int b = blockIdx.z, head = blockIdx.y;
    const int q0 = blockIdx.x * 64 + wave * 16;
    const int ld = 3 * p.C, hoff = head * D;
    const bf16_t* base = p.qkv + (size_t)b * p.N * ld;
    int qr = q0 + fr; const bool qok = qr < p.N; if (!qok) qr = p.N - 1;
    bf16x8 qf[NSL][K2], dof[NSL][K2];                  // B operands: Q[q][d], dO[q][d] with d = sl*64 + k2*32 + 8 fg ..
#pragma unroll
    for (int sl = 0; sl < NSL; ++sl)
#pragma unroll
    for (int k2 = 0; k2 < K2; ++k2) {
        qf[sl][k2] = *reinterpret_cast<const bf16x8*>(base + (size_t)qr * ld + hoff + sl * 64 + k2 * 32 + 8 * fg);
        dof[sl][k2] = *reinterpret_cast<const bf16x8*>(p.d_o + ((size_t)b * p.N + qr) * p.C + hoff + sl * 64 + k2 * 32 + 8 * fg);
    }
    const float lse = p.lse[((size_t)b * p.heads + head) * p.N + qr];
    const float dlt = p.delta[((size_t)b * p.heads + head) * p.N + qr];
    f32x4 dq[NSL][NDT];                                // dQ^T: dq[sl][dt][r] = dQ[q = fr][d = sl*64 + dt*16 + 4 fg + r]
#pragma unroll
    for (int sl = 0; sl < NSL; ++sl)
#pragma unroll
    for (int i = 0; i < NDT; ++i) dq[sl][i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int ntile = (p.N + KT - 1) / KT;
    // the next tile's rows are loaded into registers (unconditionally, rows clamped) before the current tile's MFMAs and
    // written to LDS after them: their latency hides under the compute
    u32x4 kreg[NIT], vreg[NIT];
    auto stage_load = [&](int k0) {
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int idx = tid + it * 256, sl = idx / (KT * CH), rem = idx - sl * (KT * CH), row = rem / CH, ch = rem - row * CH;
            int kr = k0 + row; if (kr >= p.N) kr = p.N - 1;
            const bf16_t* tok = base + (size_t)kr * ld + hoff + sl * 64 + ch * 8;
            kreg[it] = *reinterpret_cast<const u32x4*>(tok + p.C);
            vreg[it] = *reinterpret_cast<const u32x4*>(tok + 2 * p.C);
        }
    };
    stage_load(0);
#pragma unroll
    for (int sl = 0; sl < NSL; ++sl)
#pragma unroll
        for (int k2 = 0; k2 < K2; ++k2) asm volatile("" :: "v"(qf[sl][k2]), "v"(dof[sl][k2]));
    asm volatile("" :: "v"(lse), "v"(dlt));            // landed before the loop
    for (int t = 0; t < ntile; ++t) {
        const int k0 = t * KT;
        __syncthreads();
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int idx = tid + it * 256, sl = idx / (KT * CH), rem = idx - sl * (KT * CH), row = rem / CH, ch = rem - row * CH;
            *reinterpret_cast<u32x4*>(k_row + sl * IMG + attn_row_off(row, ch)) = kreg[it];
            *reinterpret_cast<u32x4*>(k_tr + sl * IMG + attn_tr_off(row, ch)) = kreg[it];
            *reinterpret_cast<u32x4*>(v_row + sl * IMG + attn_row_off(row, ch)) = vreg[it];
        }
        __syncthreads();
        stage_load(k0 + KT);
        bf16x8 dsf[2];                                 // dS^T fragments (B operand of the dQ product)
        f32x4 st[4], dp[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            st[j] = (f32x4){0.f, 0.f, 0.f, 0.f}; dp[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
            const int row = j * 16 + fr;
#pragma unroll
            for (int sl = 0; sl < NSL; ++sl)
#pragma unroll
            for (int k2 = 0; k2 < K2; ++k2) {
                const bf16x8 kf = *reinterpret_cast<const bf16x8*>(k_row + sl * IMG + attn_row_off(row, k2 * 4 + fg));
                const bf16x8 vf = *reinterpret_cast<const bf16x8*>(v_row + sl * IMG + attn_row_off(row, k2 * 4 + fg));
                st[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[sl][k2], st[j], 0, 0, 0);    // S^T[key][q]
                dp[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, dof[sl][k2], dp[j], 0, 0, 0);   // dP^T[key][q]
            }
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            float dsv[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int j = 2 * h + (e >> 2), r = e & 3;
                const int key = k0 + 16 * j + 4 * fg + r;
                const float pe = (key < p.N) ? __expf(st[j][r] * p.scale - lse) : 0.f;
                dsv[e] = pe * (dp[j][r] - dlt);
            }
            u32x4 pk;
#pragma unroll
            for (int e = 0; e < 4; ++e) pk[e] = pack2bf(dsv[2 * e], dsv[2 * e + 1]);
            dsf[h] = *reinterpret_cast<bf16x8*>(&pk);
        }
        // dQ^T[d][q] += K^T[d][key] dS^T[key][q]
#pragma unroll
        for (int sl = 0; sl < NSL; ++sl)
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
            for (int h = 0; h < 2; ++h)
                dq[sl][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(attn_tr_frag(k_tr + sl * IMG, dt, h, fr, fg), dsf[h], dq[sl][dt], 0, 0, 0);
    }
    if (qok) {
        bf16_t* orow = p.dqkv + ((size_t)b * p.N + qr) * ld + hoff;
#pragma unroll
        for (int sl = 0; sl < NSL; ++sl)
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt) {
            u32x2 o;
            o[0] = pack2bf(dq[sl][dt][0] * p.scale, dq[sl][dt][1] * p.scale);
            o[1] = pack2bf(dq[sl][dt][2] * p.scale, dq[sl][dt][3] * p.scale);
            *reinterpret_cast<u32x2*>(orow + sl * 64 + dt * 16 + 4 * fg) = o;
        }
    }
}

template <int D>
__global__ __launch_bounds__(256) void attn_bwd_dkv_kernel(const AttnBwdParams p) {
    constexpr int QT = 64;
    constexpr int NSL = (D + 63) / 64, DS = D < 64 ? D : 64, K2 = DS / 32, NDT = DS / 16, CH = DS / 8, IMG = QT * 128;
    constexpr int NIT = (QT * CH * NSL) / 256;
    extern __shared__ __attribute__((aligned(16))) char smem[];         // per slab: Q row/tr, dO row/tr images; then LSE, delta
    char* q_row = smem; char* q_tr = smem + NSL * IMG; char* do_row = smem + 2 * NSL * IMG; char* do_tr = smem + 3 * NSL * IMG;
    float* s_lse = reinterpret_cast<float*>(smem + 4 * NSL * IMG); float* s_dlt = s_lse + QT;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, fg = lane >> 4;
    const int b = blockIdx.z, head = blockIdx.y;
    const int kv0 = blockIdx.x * 64 + wave * 16;
    const int ld = 3 * p.C, hoff = head * D;
    const bf16_t* base = p.qkv + (size_t)b * p.N * ld;
    int kr = kv0 + fr; const bool kok = kr < p.N; if (!kok) kr = p.N - 1;
    bf16x8 kf[NSL][K2], vf[NSL][K2];                   // B operands: K[key][d], V[key][d] of this wave's 16 keys
#pragma unroll
    for (int sl = 0; sl < NSL; ++sl)
#pragma unroll
    for (int k2 = 0; k2 < K2; ++k2) {
        kf[sl][k2] = *reinterpret_cast<const bf16x8*>(base + (size_t)kr * ld + p.C + hoff + sl * 64 + k2 * 32 + 8 * fg);
        vf[sl][k2] = *reinterpret_cast<const bf16x8*>(base + (size_t)kr * ld + 2 * p.C + hoff + sl * 64 + k2 * 32 + 8 * fg);
    }
    f32x4 dk[NSL][NDT], dv[NSL][NDT];                  // dK^T, dV^T: [sl][dt][r] = d?[key = fr][d = sl*64 + dt*16 + 4 fg + r]
#pragma unroll
    for (int sl = 0; sl < NSL; ++sl)
#pragma unroll
    for (int i = 0; i < NDT; ++i) { dk[sl][i] = (f32x4){0.f, 0.f, 0.f, 0.f}; dv[sl][i] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
    const int ntile = (p.N + QT - 1) / QT;
    // register prefetch of the next query tile (unconditional, rows clamped), as in the dQ kernel
    u32x4 qreg[NIT], oreg[NIT]; float lreg = 0.f, dreg = 0.f;
    auto stage_load = [&](int q0) {
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int idx = tid + it * 256, sl = idx / (QT * CH), rem = idx - sl * (QT * CH), row = rem / CH, ch = rem - row * CH;
            int qr = q0 + row; if (qr >= p.N) qr = p.N - 1;
            qreg[it] = *reinterpret_cast<const u32x4*>(base + (size_t)qr * ld + hoff + sl * 64 + ch * 8);
            oreg[it] = *reinterpret_cast<const u32x4*>(p.d_o + ((size_t)b * p.N + qr) * p.C + hoff + sl * 64 + ch * 8);
        }
        {
            int qr = q0 + (tid & (QT - 1)); const bool ok = qr < p.N; if (!ok) qr = p.N - 1;
            const float l = p.lse[((size_t)b * p.heads + head) * p.N + qr];
            lreg = ok ? l : INFINITY;                                                      // exp(s - inf) = 0 masks the row
            dreg = p.delta[((size_t)b * p.heads + head) * p.N + qr];
        }
    };
    stage_load(0);
#pragma unroll
    for (int sl = 0; sl < NSL; ++sl)
#pragma unroll
        for (int k2 = 0; k2 < K2; ++k2) asm volatile("" :: "v"(kf[sl][k2]), "v"(vf[sl][k2]));   // landed before the loop
    for (int t = 0; t < ntile; ++t) {
        const int q0 = t * QT;
        __syncthreads();
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int idx = tid + it * 256, sl = idx / (QT * CH), rem = idx - sl * (QT * CH), row = rem / CH, ch = rem - row * CH;
            *reinterpret_cast<u32x4*>(q_row + sl * IMG + attn_row_off(row, ch)) = qreg[it];
            *reinterpret_cast<u32x4*>(q_tr + sl * IMG + attn_tr_off(row, ch)) = qreg[it];
            *reinterpret_cast<u32x4*>(do_row + sl * IMG + attn_row_off(row, ch)) = oreg[it];
            *reinterpret_cast<u32x4*>(do_tr + sl * IMG + attn_tr_off(row, ch)) = oreg[it];
        }
        if (tid < QT) { s_lse[tid] = lreg; s_dlt[tid] = dreg; }
        __syncthreads();
        stage_load(q0 + QT);
        // S[q][key] = Q K^T and dP[q][key] = dO V^T with the key on the lane (col = fr), q = 16 j + 4 fg + r
        f32x4 st[4], dp[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            st[j] = (f32x4){0.f, 0.f, 0.f, 0.f}; dp[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
            const int row = j * 16 + fr;
#pragma unroll
            for (int sl = 0; sl < NSL; ++sl)
#pragma unroll
            for (int k2 = 0; k2 < K2; ++k2) {
                const bf16x8 qf = *reinterpret_cast<const bf16x8*>(q_row + sl * IMG + attn_row_off(row, k2 * 4 + fg));
                const bf16x8 of = *reinterpret_cast<const bf16x8*>(do_row + sl * IMG + attn_row_off(row, k2 * 4 + fg));
                st[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qf, kf[sl][k2], st[j], 0, 0, 0);
                dp[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(of, vf[sl][k2], dp[j], 0, 0, 0);
            }
        }
        bf16x8 pf[2], dsf[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            float pv[8], dsv[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int j = 2 * h + (e >> 2), r = e & 3;
                const int ql = 16 * j + 4 * fg + r;
                const float pe = __expf(st[j][r] * p.scale - s_lse[ql]);
                pv[e] = pe; dsv[e] = pe * (dp[j][r] - s_dlt[ql]);
            }
            u32x4 pk, dk_;
#pragma unroll
            for (int e = 0; e < 4; ++e) { pk[e] = pack2bf(pv[2 * e], pv[2 * e + 1]); dk_[e] = pack2bf(dsv[2 * e], dsv[2 * e + 1]); }
            pf[h] = *reinterpret_cast<bf16x8*>(&pk); dsf[h] = *reinterpret_cast<bf16x8*>(&dk_);
        }
        // dV^T[d][key] += dO^T[d][q] P[q][key] ;  dK^T[d][key] += Q^T[d][q] dS[q][key]
#pragma unroll
        for (int sl = 0; sl < NSL; ++sl)
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                dv[sl][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(attn_tr_frag(do_tr + sl * IMG, dt, h, fr, fg), pf[h], dv[sl][dt], 0, 0, 0);
                dk[sl][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(attn_tr_frag(q_tr + sl * IMG, dt, h, fr, fg), dsf[h], dk[sl][dt], 0, 0, 0);
            }
    }
    if (kok) {
        bf16_t* orow = p.dqkv + ((size_t)b * p.N + kr) * ld + hoff;
#pragma unroll
        for (int sl = 0; sl < NSL; ++sl)
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt) {
            u32x2 o;
            o[0] = pack2bf(dk[sl][dt][0] * p.scale, dk[sl][dt][1] * p.scale); o[1] = pack2bf(dk[sl][dt][2] * p.scale, dk[sl][dt][3] * p.scale);
            *reinterpret_cast<u32x2*>(orow + p.C + sl * 64 + dt * 16 + 4 * fg) = o;
            o[0] = pack2bf(dv[sl][dt][0], dv[sl][dt][1]); o[1] = pack2bf(dv[sl][dt][2], dv[sl][dt][3]);
            *reinterpret_cast<u32x2*>(orow + 2 * p.C + sl * 64 + dt * 16 + 4 * fg) = o;
        }
    }
}

// delta + dQ + dK/dV for any supported head dimension
template <int D>
static inline hipError_t launch_attn_bwd_d(const AttnBwdParams& p, hipStream_t s) {
    constexpr int NSL = (D + 63) / 64;
    constexpr int LDS_DQ = 3 * NSL * 64 * 128, LDS_DKV = 4 * NSL * 64 * 128 + 2 * 64 * 4;
    static bool once = false;
    if (!once) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bwd_dq_kernel<D>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_DQ);
        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bwd_dkv_kernel<D>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_DKV);
        if (e != hipSuccess) return e;
        once = true;
    }
    const long dthreads = (long)p.B * p.N * p.heads * (D / 8);
    hipLaunchKernelGGL(attn_delta_kernel, dim3((unsigned)((dthreads + 255) / 256)), dim3(256), 0, s, p);
    hipLaunchKernelGGL(attn_bwd_dq_kernel<D>, dim3((p.N + 63) / 64, p.heads, p.B), dim3(256), LDS_DQ, s, p);
    hipLaunchKernelGGL(attn_bwd_dkv_kernel<D>, dim3((p.N + 63) / 64, p.heads, p.B), dim3(256), LDS_DKV, s, p);
    return hipGetLastError();
}
static inline hipError_t launch_attn_bwd(const AttnBwdParams& p, hipStream_t s) {
    switch (p.d) {
        case 32: return launch_attn_bwd_d<32>(p, s);
        case 64: return launch_attn_bwd_d<64>(p, s);
        case 128: return launch_attn_bwd_d<128>(p, s);
        case 256: return launch_attn_bwd_d<256>(p, s);
    }
    return hipErrorInvalidValue;
}
