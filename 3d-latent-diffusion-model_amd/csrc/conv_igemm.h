// 3D convolution as an implicit GEMM on the gfx950 matrix cores (bf16 MFMA 16x16x32, fp32 accumulate).
//
// Replaces every nn.Conv3d / nn.Linear the reference reaches through MONAI on the denoising path
// (conv3d 3^3 s1/s2, 1x1 skip convs, attention projections, VAE convs) - the
// reference itself has no kernel code; its call sites are 3d_ldm/train_diffusion.py:197-205 and
// 3d_ldm/inference.py:94-99 (UNet forward inside LatentDiffusionInferer).
//
// Layout: activations NDHWC bf16 (C % 32 == 0), weights [tap][CoutPad][Cin] bf16 (K contiguous).
// One workgroup (4 waves, 256 threads) owns a BM-voxel x BN-cout output tile; each wave a 64x64 sub-tile
// (4x4 MFMA tiles, 64 accumulator VGPRs).  The K loop walks (tap, Cin chunk of BK) steps; per step the
// voxel rows (shifted by the tap, zero page where padded) and the weight rows are copied global->LDS with
// LDS-DMA (global_load_lds_dwordx4, 1 KiB per wave instruction) into a double-buffered, XOR-swizzled
// image that ds_read_b128 reads conflict-free.  Never materialised: zero padding, nearest x2 upsampling
// (source index >> 1), channel concatenation (two source pointers) and the 1x1 skip convolution of a
// ResBlock (appended as extra K steps, "group 1").  MFMA roles: A = weights (rows = cout), B = voxels
// (cols), so one lane ends up with 16 consecutive output channels of one voxel -> 2 x 16-byte stores.
//
// Split-K: blockIdx also enumerates K slices; slices write fp32 slabs that splitk_finalize_kernel sums
// (deterministic, no atomics) before applying the epilogue.
#pragma once
#include "common.h"

struct ConvParams {
    // group 0: k^3 convolution over (x0a | x0b) channel-concatenated sources
    const bf16_t* x0a; const bf16_t* x0b; int c0a, c0b;
    const bf16_t* w0;                 // [taps][CoutPad][c0a + c0b]
    // group 1 (optional, steps1 > 0): 1x1 convolution over (x1a | x1b) at output resolution
    const bf16_t* x1a; const bf16_t* x1b; int c1a, c1b;
    const bf16_t* w1;                 // [CoutPad][c1a + c1b]
    const bf16_t* zero_page;          // zeros, at least one full input row (C * 2 bytes) long
    int N, Din, Hin, Win;             // group-0 source dims (before the optional x2 upsample)
    int Dout, Hout, Wout;
    int ksize, stride, pad, ups;      // ups = 0/1 : nearest-neighbour x2 upsample folded into the loader
    int M;                            // N * Dout * Hout * Wout
    int CoutS;                        // stored output channels (multiple of 32, >= real Cout)
    int CoutPad;                      // weight rows (multiple of 64 and of BN)
    int CoutReal;                     // channels written in fp32-NCDHW mode
    int nchunk0, nchunk1;             // Cin / BK per group
    int steps0, steps1;               // K steps per group (taps * nchunk0, nchunk1)
    int splitk, steps_per_split;
    int mtiles, ntiles;
    // epilogue (splitk == 1) ------------------------------------------------------------
    const float* bias;                // [CoutPad] or null
    const float* bias2;               // [CoutPad] or null (bias of the fused 1x1 skip)
    const float* temb; int temb_stride;   // per-sample channel bias [N][temb_stride] or null
    const bf16_t* residual;           // [M][CoutS] or null
    bf16_t* out;                      // [M][CoutS] bf16 NDHWC            (mode 0)
    float* out_f32;                   // [N][CoutReal][Dout*Hout*Wout]    (mode 1)
    float* partial;                   // [splitk][M][CoutPad] fp32 slabs  (splitk > 1)
};

// Bijective XCD-aware remap: blocks b, b+8, b+16.. share an XCD (observed round-robin dispatch); give each
// XCD a contiguous range of logical tiles so neighbouring tiles share weights/halo rows in one L2.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    int q = nwg >> 3, r = nwg & 7, x = bid & 7, i = bid >> 3;
    int base = (x < r) ? x * (q + 1) : r * (q + 1) + (x - r) * q;
    return base + i;
}

__device__ __forceinline__ void glds16(const void* gsrc, void* lds_dst) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_dst, 16, 0, 0);
}

template <int WGM, int WGN, int BK>
__global__ __launch_bounds__(256, 2) void conv_igemm_kernel(const ConvParams p) {
    constexpr int BM = 64 * WGM, BN = 64 * WGN;
    constexpr int RB = BK * 2;                 // bytes per LDS row
    constexpr int CPR = RB / 16;               // 16-byte chunks per row (4 | 8)
    constexpr int RPP = 1024 / RB;             // rows per LDS-DMA piece (16 | 8)
    constexpr int PA = BM / RPP / 4;           // voxel pieces per wave
    constexpr int PB = BN / RPP / 4;           // weight pieces per wave
    constexpr int STAGE = (BM + BN) * RB;      // bytes per pipeline stage
    constexpr int SWZ_SHIFT = (CPR == 8) ? 1 : 2;
    constexpr int KS = BK / 32;                // MFMA k-substeps per stage
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave % WGM, wn = wave / WGM;

    // ---- block -> (split, ntile, mtile); mtile fastest so one XCD streams one weight panel
    const int nwg = gridDim.x;
    int lid = xcd_remap(blockIdx.x, nwg);
    const int mtile = lid % p.mtiles; lid /= p.mtiles;
    const int ntile = lid % p.ntiles;
    const int split = lid / p.ntiles;
    const int m0 = mtile * BM, n0 = ntile * BN;
    const int steps_total = p.steps0 + p.steps1;
    const int s_begin = split * p.steps_per_split;
    int s_end = s_begin + p.steps_per_split; if (s_end > steps_total) s_end = steps_total;

    // ---- per-lane loader state -----------------------------------------------------------------
    const int prow = lane / CPR;                       // row inside a piece
    const int pchunk = lane % CPR;                     // physical 16-B chunk this lane fills
    const int DHWo = p.Dout * p.Hout * p.Wout;
    const int HWo = p.Hout * p.Wout;
    const int DinU = p.Din << p.ups, HinU = p.Hin << p.ups, WinU = p.Win << p.ups;

    int a_id0[PA], a_ih0[PA], a_iw0[PA], a_nbase[PA], a_m[PA], a_koff[PA];
#pragma unroll
    for (int j = 0; j < PA; ++j) {
        const int row = (wave * PA + j) * RPP + prow;  // voxel row inside the tile
        const int m = m0 + row;
        const int swz = (row >> SWZ_SHIFT) & (CPR - 1);
        a_koff[j] = ((pchunk ^ swz) * 8);              // logical channel offset inside the BK chunk
        if (m < p.M) {
            const int n = m / DHWo; int r = m - n * DHWo;
            const int od = r / HWo; r -= od * HWo;
            const int oh = r / p.Wout; const int ow = r - oh * p.Wout;
            a_id0[j] = od * p.stride - p.pad; a_ih0[j] = oh * p.stride - p.pad; a_iw0[j] = ow * p.stride - p.pad;
            a_nbase[j] = n * p.Din * p.Hin * p.Win;
            a_m[j] = m;
        } else {
            a_id0[j] = -(1 << 20); a_ih0[j] = 0; a_iw0[j] = 0; a_nbase[j] = 0; a_m[j] = -1;
        }
    }
    int b_row[PB], b_koff[PB];
#pragma unroll
    for (int j = 0; j < PB; ++j) {
        const int R = (wave * PB + j) * RPP + prow;    // LDS row inside the weight tile
        const int swz = (R >> SWZ_SHIFT) & (CPR - 1);
        b_koff[j] = ((pchunk ^ swz) * 8);
        // LDS row (64q + 16nt + i) holds cout 64q + 16(i>>2) + 4nt + (i&3): after the MFMA a lane owns 16
        // consecutive couts (accumulator row 4g+r of tile nt  <->  cout 16g + 4nt + r).
        const int q = R >> 6, nt = (R >> 4) & 3, i = R & 15;
        b_row[j] = n0 + 64 * q + 16 * (i >> 2) + 4 * nt + (i & 3);
    }

    // Loader state machine.  The K loop visits steps s = (group, tap, chunk) in order; the loader runs one step
    // ahead of the MFMAs.  Row pointers are recomputed only when the tap or the concat source changes; inside
    // a (tap, source) run each step is one 64-bit add per row (padded rows point into the zero page, which is
    // at least one full row long, so the same add is harmless there).
    const int taps_k = p.ksize, c0a = p.c0a, c0b = p.c0b, c1a = p.c1a, c1b = p.c1b;
    const int steps0 = p.steps0, nchunk0 = p.nchunk0;
    const bf16_t* const x0a = p.x0a; const bf16_t* const x0b = p.x0b;
    const bf16_t* const x1a = p.x1a; const bf16_t* const x1b = p.x1b;
    const bf16_t* const zp = p.zero_page;
    const int upsh = p.ups, Hin = p.Hin, Win = p.Win, CoutPad = p.CoutPad, nchunk1 = p.nchunk1;
    const bf16_t* const w0p = p.w0; const bf16_t* const w1p = p.w1;

    int ld_s = s_begin;                                // next step to load
    int ld_grp, ld_tap, ld_chunk, ld_kd, ld_kh, ld_kw;
    if (ld_s < steps0) {
        ld_grp = 0; ld_tap = ld_s / nchunk0; ld_chunk = ld_s - ld_tap * nchunk0;
        const int kk = taps_k * taps_k;
        ld_kd = ld_tap / kk; ld_kh = (ld_tap - ld_kd * kk) / taps_k; ld_kw = ld_tap - ld_kd * kk - ld_kh * taps_k;
    } else { ld_grp = 1; ld_tap = 0; ld_chunk = ld_s - steps0; ld_kd = ld_kh = ld_kw = 0; }
    int ld_src = -1;                                   // concat source the cached pointers belong to (-1: stale)
    int a_voff[PA];
    const char* a_ptr[PA];
    const char* b_ptr[PB];
#pragma unroll
    for (int j = 0; j < PA; ++j) { a_voff[j] = -1; a_ptr[j] = nullptr; }
#pragma unroll
    for (int j = 0; j < PB; ++j) b_ptr[j] = nullptr;
    bool voff_stale = true;

    // ---- fragment read addresses (swizzled) -----------------------------------------------------
    const int fr = lane & 15, fg = lane >> 4;
    int a_rd[4][KS], b_rd[4][KS];                     // byte offsets inside a stage
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int ra = wm * 64 + t * 16 + fr;
        const int rb = wn * 64 + t * 16 + fr;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int c = ks * 4 + fg;
            a_rd[t][ks] = ra * RB + ((c ^ ((ra >> SWZ_SHIFT) & (CPR - 1))) << 4);
            b_rd[t][ks] = BM * RB + rb * RB + ((c ^ ((rb >> SWZ_SHIFT) & (CPR - 1))) << 4);
        }
    }

    f32x4 acc[4][4];                                   // [cout tile nt][voxel tile mt]
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // Software pipeline, one barrier per K step: iteration s issues the loads of step s+1 into the other
    // stage and runs the MFMAs of step s (iteration s_begin-1 only primes the pipe).
    for (int s = s_begin - 1; s < s_end; ++s) {
        const int buf = (s - s_begin) & 1;
        // hipcc does not count LDS-DMA in the waits it emits for __syncthreads(): drain this wave's copies by hand,
        // then the barrier makes every wave's copies of stage `buf` visible and frees stage buf^1.
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (s + 1 < s_end) {
            const int ibuf = buf ^ 1;
            const int ca = ld_grp ? c1a : c0a, cb = ld_grp ? c1b : c0b;
            const int ch = ld_chunk * BK;                  // first channel of this chunk in the concatenated input
            const int src = (ch >= ca) ? 1 : 0;
            if (voff_stale) {                              // wave-uniform: new tap (or group) -> new voxel offsets
                voff_stale = false; ld_src = -1;
                if (ld_grp == 0) {
#pragma unroll
                    for (int j = 0; j < PA; ++j) {
                        const int id = a_id0[j] + ld_kd, ih = a_ih0[j] + ld_kh, iw = a_iw0[j] + ld_kw;
                        const bool ok = ((unsigned)id < (unsigned)DinU) & ((unsigned)ih < (unsigned)HinU) &
                                        ((unsigned)iw < (unsigned)WinU);
                        const int v = a_nbase[j] + ((id >> upsh) * Hin + (ih >> upsh)) * Win + (iw >> upsh);
                        a_voff[j] = ok ? v : -1;
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < PA; ++j) a_voff[j] = a_m[j];
                }
                const int cin = ca + cb;
                const bf16_t* wb = ld_grp ? w1p : w0p + (size_t)ld_tap * CoutPad * cin;
#pragma unroll
                for (int j = 0; j < PB; ++j)
                    b_ptr[j] = reinterpret_cast<const char*>(wb + (size_t)b_row[j] * cin + b_koff[j]);
            }
            if (src != ld_src) {                           // wave-uniform: (re)base the row pointers on this source
                ld_src = src;
                const bf16_t* xs = ld_grp ? (src ? x1b : x1a) : (src ? x0b : x0a);
                const int cs = src ? cb : ca;
#pragma unroll
                for (int j = 0; j < PA; ++j) {
                    const bool ok = a_voff[j] >= 0;
                    const bf16_t* rowp = xs + (size_t)(ok ? a_voff[j] : 0) * cs;
                    a_ptr[j] = reinterpret_cast<const char*>((ok ? rowp : zp) + a_koff[j]);
                }
            }
            const int chs = src ? ch - ca : ch;
            char* stage = smem + ibuf * STAGE;
#pragma unroll
            for (int j = 0; j < PA; ++j) glds16(a_ptr[j] + chs * 2, stage + (wave * PA + j) * 1024);
#pragma unroll
            for (int j = 0; j < PB; ++j) glds16(b_ptr[j] + ch * 2, stage + BM * RB + (wave * PB + j) * 1024);
            // advance to the next step
            ++ld_s; ++ld_chunk;
            const int nch = ld_grp ? nchunk1 : nchunk0;
            if (ld_chunk == nch) {
                ld_chunk = 0; voff_stale = true;
                if (ld_grp == 0) {
                    ++ld_tap;
                    if (++ld_kw == taps_k) { ld_kw = 0; if (++ld_kh == taps_k) { ld_kh = 0; ++ld_kd; } }
                    if (ld_s >= steps0) { ld_grp = 1; ld_tap = 0; }
                }
            }
        }
        if (s < s_begin) continue;
        const char* stage = smem + buf * STAGE;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            bf16x8 wf[4], af[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                wf[t] = *reinterpret_cast<const bf16x8*>(stage + b_rd[t][ks]);
                af[t] = *reinterpret_cast<const bf16x8*>(stage + a_rd[t][ks]);
            }
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                for (int mt = 0; mt < 4; ++mt)
                    acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[nt], af[mt], acc[nt][mt], 0, 0, 0);
        }
    }

    // ---- epilogue -------------------------------------------------------------------------------
    const int cbase = n0 + wn * 64 + 16 * fg;          // this lane's 16 consecutive couts
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
        const int m = m0 + wm * 64 + mt * 16 + fr;
        if (m >= p.M) continue;
        float v[16];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) v[nt * 4 + r] = acc[nt][mt][r];
        if (p.splitk > 1) {
            float* dst = p.partial + ((size_t)split * p.M + m) * p.CoutPad + cbase;
#pragma unroll
            for (int q = 0; q < 4; ++q)
                *reinterpret_cast<float4*>(dst + 4 * q) = make_float4(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
            continue;
        }
        const int n = m / DHWo;
        if (p.bias) {
#pragma unroll
            for (int q = 0; q < 16; ++q) v[q] += p.bias[cbase + q];
        }
        if (p.bias2) {
#pragma unroll
            for (int q = 0; q < 16; ++q) v[q] += p.bias2[cbase + q];
        }
        if (p.temb) {                                  // rows are padded to CoutPad by the host: no bounds check
            const float* te = p.temb + (size_t)n * p.temb_stride + cbase;
#pragma unroll
            for (int q = 0; q < 16; ++q) v[q] += te[q];
        }
        if (p.out_f32) {
            const int sp = m - n * DHWo;
#pragma unroll
            for (int q = 0; q < 16; ++q)
                if (cbase + q < p.CoutReal) p.out_f32[((size_t)n * p.CoutReal + cbase + q) * DHWo + sp] = v[q];
            continue;
        }
        if (cbase >= p.CoutS) continue;                // weight-row padding beyond the stored channels
        if (p.residual) {
            const u32x4* rp = reinterpret_cast<const u32x4*>(p.residual + (size_t)m * p.CoutS + cbase);
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const u32x4 rv = rp[h];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    v[h * 8 + 2 * q] += __uint_as_float(rv[q] << 16);
                    v[h * 8 + 2 * q + 1] += __uint_as_float(rv[q] & 0xffff0000u);
                }
            }
        }
        u32x4* op = reinterpret_cast<u32x4*>(p.out + (size_t)m * p.CoutS + cbase);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            u32x4 o;
#pragma unroll
            for (int q = 0; q < 4; ++q) o[q] = pack2bf(v[h * 8 + 2 * q], v[h * 8 + 2 * q + 1]);
            op[h] = o;
        }
    }
}

// Sums split-K slabs and applies the same epilogue as the fused path.  One thread per (voxel, 8 channels).
struct FinalizeParams {
    const float* partial; int splitk; int M; int CoutPad; int CoutS; int CoutReal; int DHWo;
    const float* bias; const float* bias2; const float* temb; int temb_stride; const bf16_t* residual;
    bf16_t* out; float* out_f32;
};

__global__ __launch_bounds__(256) void splitk_finalize_kernel(const FinalizeParams p) {
    const int cvec = p.CoutS / 8;
    const long total = (long)p.M * cvec;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int m = (int)(i / cvec);
        const int c = (int)(i - (long)m * cvec) * 8;
        float v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = 0.f;
        for (int s = 0; s < p.splitk; ++s) {
            const float4* src = reinterpret_cast<const float4*>(p.partial + ((size_t)s * p.M + m) * p.CoutPad + c);
            const float4 a = src[0], b = src[1];
            v[0] += a.x; v[1] += a.y; v[2] += a.z; v[3] += a.w;
            v[4] += b.x; v[5] += b.y; v[6] += b.z; v[7] += b.w;
        }
        const int n = m / p.DHWo;
        if (p.bias) {
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] += p.bias[c + q];
        }
        if (p.bias2) {
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] += p.bias2[c + q];
        }
        if (p.temb) {
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] += p.temb[(size_t)n * p.temb_stride + c + q];
        }
        if (p.out_f32) {
            const int sp = m - n * p.DHWo;
#pragma unroll
            for (int q = 0; q < 8; ++q)
                if (c + q < p.CoutReal) p.out_f32[((size_t)n * p.CoutReal + c + q) * p.DHWo + sp] = v[q];
            continue;
        }
        if (p.residual) {
            const u32x4 rv = *reinterpret_cast<const u32x4*>(p.residual + (size_t)m * p.CoutS + c);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                v[2 * q] += __uint_as_float(rv[q] << 16);
                v[2 * q + 1] += __uint_as_float(rv[q] & 0xffff0000u);
            }
        }
        u32x4 o;
#pragma unroll
        for (int q = 0; q < 4; ++q) o[q] = pack2bf(v[2 * q], v[2 * q + 1]);
        *reinterpret_cast<u32x4*>(p.out + (size_t)m * p.CoutS + c) = o;
    }
}
