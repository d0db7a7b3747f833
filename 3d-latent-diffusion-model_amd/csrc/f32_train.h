// fp32 precision mode, training half: the backward kernels of the fp32 plans (forward kernels: f32_path.h).
//
// The reference trains in fp32 (autocast is switched off: 3d_ldm/train_diffusion.py:177,237), so with
// ldm_model_set_precision(LDM_PREC_FP32) the training plans reproduce its arithmetic: fp32 activations, fp32 weights, fp32 MFMA
// (v_mfma_f32_32x32x2_f32), gradients within ~1e-5 of torch autograd through the CPU oracle at UNIT weight gain (the bf16 plans
// sit on their own rounding floor there: 0.4 rel-L2, tests/test_gpu_train.py).  Same plan structure as the bf16 backward
// (DESIGN.md section 3.3b): data gradient = the forward conv kernel on dY with flipped + transposed weights, weight gradient =
// a GEMM that contracts over voxels, GroupNorm / attention backward in three launches each, everything deterministic (no atomics).
#pragma once
#include "f32_path.h"
#include "norm_elem.h"

// ---- weight gradient: dW[tap][co][ci] = sum_m dY[m][co] * X[voxel(m, tap)][ci] -------------------------------------------------
struct Wgrad32Params {
    const float* dy; int cdy;                       // [M][cdy]
    const float* x; int cx;                         // source tensor [rows][cx] (one of the channel-concatenated conv inputs)
    float* dw;                                      // [ksplit][taps][Cout][dw_ld] partial matrices; this source's columns start at dw_ci_off
    int Cout, Cin, dw_ld, dw_ci_off;
    int N, Din, Hin, Win, Dout, Hout, Wout, ksize, stride, pad, ups, M;
    int co_tiles, ci_tiles, ksplit; long slab_stride;
};
// workgroup = one tap x 128 couts x 128 cins x one slice of the voxel range; 4 waves (2 x 2), wave tile 64 x 64 as 2 x 2 MFMA tiles
// of 32 x 32 (A = dY^T: row co, B = X: column ci), K step = 16 voxels.  Both operands are stored [voxel][channel] in LDS exactly as
// they are loaded (rows of consecutive channels) and fed to the MFMAs with one ds_read_b32 per lane: a lane's A element for
// k-slot h is dY[voxel 2 kk + h][co], 32 consecutive floats per lane half = conflict free.
__global__ __launch_bounds__(256, 2) void wgrad_f32_kernel(const Wgrad32Params p) {
    constexpr int BK = 16, LD = 128 + 4;
    __shared__ __attribute__((aligned(16))) float sA[2][BK * LD];      // dY rows [voxel][co]
    __shared__ __attribute__((aligned(16))) float sB[2][BK * LD];      // X rows  [voxel][ci]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave & 1, wn = wave >> 1;                           // wm: co half, wn: ci half
    int lid = blockIdx.x;
    const int split = lid % p.ksplit; lid /= p.ksplit;
    const int ci_t = lid % p.ci_tiles; lid /= p.ci_tiles;
    const int co_t = lid % p.co_tiles; const int tap = lid / p.co_tiles;
    const int co0 = co_t * 128, ci0 = ci_t * 128;
    const int steps_total = (p.M + BK - 1) / BK;
    const int sps = (steps_total + p.ksplit - 1) / p.ksplit;
    const int s_begin = split * sps;
    int s_end = s_begin + sps; if (s_end > steps_total) s_end = steps_total;
    const int DHWo = p.Dout * p.Hout * p.Wout, HWo = p.Hout * p.Wout;
    const int DinU = p.Din << p.ups, HinU = p.Hin << p.ups, WinU = p.Win << p.ups;
    int kd = 0, kh = 0, kw = 0;
    if (p.ksize == 3) { kd = tap / 9; kh = (tap - kd * 9) / 3; kw = tap - kd * 9 - kh * 3; }
    // loader: thread = (row of the 16-voxel step, 32-float4 column); 512 float4 per operand tile -> 2 per thread
    const int lrow = tid >> 4, lcol = (tid & 15) * 4;                  // rows lrow (0..15), float4 columns lcol and lcol + 64
    float4 ra[2], rb[2];
    auto load_step = [&](int s) {
        const int m = s * BK + lrow;
        int v = -1;
        if (m < p.M) {
            const int n = m / DHWo; int r = m - n * DHWo; const int od = r / HWo; r -= od * HWo; const int oh = r / p.Wout, ow = r - oh * p.Wout;
            const int id = od * p.stride + kd - p.pad, ih = oh * p.stride + kh - p.pad, iw = ow * p.stride + kw - p.pad;
            if (((unsigned)id < (unsigned)DinU) & ((unsigned)ih < (unsigned)HinU) & ((unsigned)iw < (unsigned)WinU))
                v = n * p.Din * p.Hin * p.Win + ((id >> p.ups) * p.Hin + (ih >> p.ups)) * p.Win + (iw >> p.ups);
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int c = lcol + 64 * j;
            ra[j] = (m < p.M && co0 + c < p.cdy) ? *reinterpret_cast<const float4*>(p.dy + (size_t)m * p.cdy + co0 + c) : make_float4(0.f, 0.f, 0.f, 0.f);
            rb[j] = (v >= 0 && ci0 + c < p.cx) ? *reinterpret_cast<const float4*>(p.x + (size_t)v * p.cx + ci0 + c) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto store_step = [&](int buf) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            *reinterpret_cast<float4*>(&sA[buf][lrow * LD + lcol + 64 * j]) = ra[j];
            *reinterpret_cast<float4*>(&sB[buf][lrow * LD + lcol + 64 * j]) = rb[j];
        }
    };
    f32x16 acc[2][2];                                                  // [co tile][ci tile]
#pragma unroll
    for (int i = 0; i < 2; ++i)
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
        for (int kk = 0; kk < BK / 2; ++kk) {
            float a[2], b[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) a[i] = sA[buf][(2 * kk + fh) * LD + wm * 64 + i * 32 + fr];
#pragma unroll
            for (int j = 0; j < 2; ++j) b[j] = sB[buf][(2 * kk + fh) * LD + wn * 64 + j * 32 + fr];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        if (s + 1 < s_end) store_step(buf ^ 1);
        __syncthreads();
    }
    // accumulator register 4g + r of tile (i, j): row (co) = 32 i + 8 g + 4 fh + r, column (ci) = 32 j + fr
    float* dst = p.dw + (size_t)split * p.slab_stride + (size_t)tap * p.Cout * p.dw_ld;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int ci = ci0 + wn * 64 + j * 32 + fr;
            if (ci >= p.Cin) continue;
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int co = co0 + wm * 64 + i * 32 + 8 * g + 4 * fh + r;
                    if (co < p.Cout) dst[(size_t)co * p.dw_ld + p.dw_ci_off + ci] = acc[i][j][4 * g + r];
                }
        }
}

// (A 3 x bf16 form of this kernel -- operands split into hi + lo bf16 by a transposing loader, as conv_x3_kernel does for the forward --
// was built and measured in round 2: gradients stayed at 2e-6 ... 7e-5 from fp32 autograd, since an error in dW is not amplified by
// anything downstream, but it ran 20 % SLOWER than this kernel (250 vs 208 us per launch on average; 820 vs 690 us for the 24^3 256-channel
// shapes) at every split setting: with one or two workgroups per CU and a one-step register prefetch the loop is bound by the latency
// of its 32 KiB of operand loads per step, not by the matrix pipe, so making the MFMAs 5x cheaper buys nothing.  Removed.)

// ---- GroupNorm (+SiLU) backward on fp32 tensors: partial sums per slab, fold (gn_bwd_finalize_kernel, type agnostic), apply ----
struct Gnb32Params {
    const float* dy; const float* xa; const float* xb; int ca, cb;
    const float* ab; const float* mr; const float* gamma;
    int groups, DHW, N, silu, nslab, rows_per_slab;
    float* partial; const float* gsum;
    const float* acc_a; const float* acc_b; float* dxa; float* dxb;
};
__device__ __forceinline__ float gn32_bwd_g(float dy, float u, int act) {
    if (!act) return dy;
    if (act == 2) return u > 0.f ? dy : 0.2f * dy;
    const float sg = 1.0f / (1.0f + expf(-u));
    return dy * sg * (1.0f + u * (1.0f - sg));
}
__global__ __launch_bounds__(256) void gnb32_stats_kernel(const Gnb32Params p) {
    __shared__ float red[256 * 8];
    const int C = p.ca + p.cb, cvec = C / 4, cpg = C / p.groups;
    const int n = blockIdx.y, slab = blockIdx.x, tid = threadIdx.x;
    const int r0 = slab * p.rows_per_slab;
    int r1 = r0 + p.rows_per_slab; if (r1 > p.DHW) r1 = p.DHW;
    for (int cv0 = 0; cv0 < cvec; cv0 += 256) {
        const int nv = cvec - cv0 < 256 ? cvec - cv0 : 256;
        const int rp = 256 / nv > 0 ? 256 / nv : 1;
        const int cv = tid % nv, rl = tid / nv;
        float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
        if (rl < rp) {
            const int c = (cv0 + cv) * 4;
            const bool second = c >= p.ca;
            const float* xs = second ? p.xb : p.xa;
            const int cs = second ? p.cb : p.ca, cc = second ? c - p.ca : c;
            float a[4], b[4], mean[4], rstd[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                a[k] = p.ab[((size_t)n * C + c + k) * 2]; b[k] = p.ab[((size_t)n * C + c + k) * 2 + 1];
                const int g = (c + k) / cpg;
                mean[k] = p.mr[((size_t)n * p.groups + g) * 2]; rstd[k] = p.mr[((size_t)n * p.groups + g) * 2 + 1];
            }
            for (int r = r0 + rl; r < r1; r += rp) {
                const size_t row = (size_t)n * p.DHW + r;
                const float4 xv = *reinterpret_cast<const float4*>(xs + row * cs + cc);
                const float4 dv = *reinterpret_cast<const float4*>(p.dy + row * C + c);
                const float xx[4] = {xv.x, xv.y, xv.z, xv.w}, dd[4] = {dv.x, dv.y, dv.z, dv.w};
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float g = gn32_bwd_g(dd[k], a[k] * xx[k] + b[k], p.silu);
                    s1[k] += g; s2[k] += g * (xx[k] - mean[k]) * rstd[k];
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 4; ++k) { red[tid * 8 + k] = s1[k]; red[tid * 8 + 4 + k] = s2[k]; }
        __syncthreads();
        if (tid < nv) {
            for (int r2 = 1; r2 < rp; ++r2) {
                const int t2 = r2 * nv + tid;
#pragma unroll
                for (int k = 0; k < 4; ++k) { s1[k] += red[t2 * 8 + k]; s2[k] += red[t2 * 8 + 4 + k]; }
            }
            float* dst = p.partial + (((size_t)n * p.nslab + slab) * C + (cv0 + tid) * 4) * 2;
#pragma unroll
            for (int k = 0; k < 4; ++k) { dst[2 * k] = s1[k]; dst[2 * k + 1] = s2[k]; }
        }
    }
}
__global__ __launch_bounds__(256) void gnb32_apply_kernel(const Gnb32Params p) {
    const int C = p.ca + p.cb, cvec = C / 4, cpg = C / p.groups;
    const long total = (long)p.N * p.DHW * cvec;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long row = i / cvec;
        const int c = (int)(i - row * cvec) * 4;
        const int n = (int)(row / p.DHW);
        const bool second = c >= p.ca;
        const int cs = second ? p.cb : p.ca, cc = second ? c - p.ca : c;
        const float4 xv = *reinterpret_cast<const float4*>((second ? p.xb : p.xa) + row * cs + cc);
        const float4 dv = *reinterpret_cast<const float4*>(p.dy + row * C + c);
        const float* accp = second ? p.acc_b : p.acc_a;
        float4 av = make_float4(0.f, 0.f, 0.f, 0.f);
        if (accp) av = *reinterpret_cast<const float4*>(accp + row * cs + cc);
        const float xx[4] = {xv.x, xv.y, xv.z, xv.w}, dd[4] = {dv.x, dv.y, dv.z, dv.w}, aa[4] = {av.x, av.y, av.z, av.w};
        float out[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float a = p.ab[((size_t)n * C + c + k) * 2], b = p.ab[((size_t)n * C + c + k) * 2 + 1];
            const int g = (c + k) / cpg;
            const float mean = p.mr[((size_t)n * p.groups + g) * 2], rstd = p.mr[((size_t)n * p.groups + g) * 2 + 1];
            const float m1 = p.gsum[((size_t)n * p.groups + g) * 2], m2 = p.gsum[((size_t)n * p.groups + g) * 2 + 1];
            const float gg = gn32_bwd_g(dd[k], a * xx[k] + b, p.silu);
            const float xh = (xx[k] - mean) * rstd;
            out[k] = rstd * (p.gamma[c + k] * gg - m1 - xh * m2) + aa[k];
        }
        *reinterpret_cast<float4*>((second ? p.dxb : p.dxa) + row * cs + cc) = make_float4(out[0], out[1], out[2], out[3]);
    }
}

// ---- attention backward (fp32): delta, dQ (wave = 32 queries, loops over key tiles), dK / dV (wave = 32 keys, loops over query tiles)
struct Attn32BwdParams {
    const float* qkv; const float* o; const float* d_o; const float* lse; float* delta; float* dqkv;
    int B, N, C, heads, d; float scale;
};
__global__ __launch_bounds__(256) void attn32_delta_kernel(const Attn32BwdParams p) {
    const int cpd = p.d >> 2;                              // float4 chunks per head: 8 .. 64 consecutive lanes (a power of two <= 64)
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    const long total = (long)p.B * p.N * p.heads * cpd;
    float s = 0.f; long tok = 0; int head = 0;
    if (idx < total) {
        const int ch = (int)(idx % cpd);
        const long th = idx / cpd;
        head = (int)(th % p.heads); tok = th / p.heads;
        const float4 a = *reinterpret_cast<const float4*>(p.o + tok * p.C + head * p.d + ch * 4);
        const float4 b = *reinterpret_cast<const float4*>(p.d_o + tok * p.C + head * p.d + ch * 4);
        s = a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w;
    }
    for (int o = 1; o < cpd && o < 64; o <<= 1) s += __shfl_xor(s, o, 64);
    if (idx < total && (idx % cpd) == 0) {
        const long bb = tok / p.N, i = tok - bb * p.N;
        p.delta[((size_t)bb * p.heads + head) * p.N + i] = s;
    }
}
// tile loader shared by the two kernels: rows [r0, r0 + 32) of one of q / k / v / dO into an LDS image with row stride D + 1
template <int D>
__device__ __forceinline__ void attn32_load_tile(float* dst, const float* src, size_t row_stride, int r0, int N, int tid, int nthr) {
    for (int e = tid; e < 32 * (D / 4); e += nthr) {
        const int row = e / (D / 4), c4 = (e - row * (D / 4)) * 4;
        int r = r0 + row; if (r >= N) r = N - 1;
        const float4 v = *reinterpret_cast<const float4*>(src + (size_t)r * row_stride + c4);
        float* d = dst + row * (D + 1) + c4; d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
    }
}
// NW = waves per workgroup (each owns 32 queries / keys): 4 for d <= 64, 2 for d = 128, 1 for d = 256 (the AutoencoderKL's single-head blocks),
// so that the row images stay inside the 160 KiB LDS
template <int DT, int NW>
__global__ __launch_bounds__(64 * NW) void attn32_bwd_dq_kernel(const Attn32BwdParams p) {
    constexpr int D = DT * 32, LDQ = D + 1, NT = 64 * NW, ROWS = 32 * NW;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* sQ = reinterpret_cast<float*>(smem);            // [NW waves * 32][LDQ]  Q rows of the workgroup's queries
    float* sO = sQ + ROWS * LDQ;                           // dO rows
    float* sK = sO + ROWS * LDQ;                           // [32][LDQ] key tile
    float* sV = sK + 32 * LDQ;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 31, fh = lane >> 5;
    const int b = blockIdx.z, head = blockIdx.y, q0 = blockIdx.x * ROWS;
    const int C3 = 3 * p.C;
    const float* base = p.qkv + (size_t)b * p.N * C3 + head * D;
    const float* dob = p.d_o + (size_t)b * p.N * p.C + head * D;
    for (int w = 0; w < NW; ++w) {
        attn32_load_tile<D>(sQ + w * 32 * LDQ, base, C3, q0 + w * 32, p.N, tid, NT);
        attn32_load_tile<D>(sO + w * 32 * LDQ, dob, p.C, q0 + w * 32, p.N, tid, NT);
    }
    int qi = q0 + wave * 32 + fr; const bool qok = qi < p.N; if (!qok) qi = p.N - 1;
    const float lse = p.lse[((size_t)b * p.heads + head) * p.N + qi];
    const float dlt = p.delta[((size_t)b * p.heads + head) * p.N + qi];
    f32x16 dq[DT];
#pragma unroll
    for (int t = 0; t < DT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) dq[t][r] = 0.f;
    for (int k0 = 0; k0 < p.N; k0 += 32) {
        __syncthreads();
        attn32_load_tile<D>(sK, base + p.C, C3, k0, p.N, tid, NT);
        attn32_load_tile<D>(sV, base + 2 * p.C, C3, k0, p.N, tid, NT);
        __syncthreads();
        f32x16 s, dp;
#pragma unroll
        for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
        const float* kq = sK + fr * LDQ + fh; const float* vq = sV + fr * LDQ + fh;
        const float* qq = sQ + (wave * 32 + fr) * LDQ + fh; const float* oq = sO + (wave * 32 + fr) * LDQ + fh;
#pragma unroll 8
        for (int kk = 0; kk < D / 2; ++kk) {
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(kq[2 * kk], qq[2 * kk], s, 0, 0, 0);       // S^T[key][q]
            dp = __builtin_amdgcn_mfma_f32_32x32x2f32(vq[2 * kk], oq[2 * kk], dp, 0, 0, 0);     // dP^T[key][q]
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = k0 + (r & 3) + 8 * (r >> 2) + 4 * fh;
            const float pe = (key < p.N) ? expf(s[r] * p.scale - lse) : 0.f;
            s[r] = pe * (dp[r] - dlt);                                                            // dS^T[key][q]
        }
        // dQ^T[dd][q] += K^T[dd][key] dS^T[key][q]; MFMA step j pairs key (j & 3) + 8 (j >> 2) (lanes 0-31) with that key + 4
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const float* krow = sK + ((j & 3) + 8 * (j >> 2) + 4 * fh) * LDQ + fr;
#pragma unroll
            for (int t = 0; t < DT; ++t) dq[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(krow[32 * t], s[j], dq[t], 0, 0, 0);
        }
    }
    if (!qok) return;
    float* dst = p.dqkv + ((size_t)b * p.N + qi) * C3 + head * D;
#pragma unroll
    for (int t = 0; t < DT; ++t)
#pragma unroll
        for (int g = 0; g < 4; ++g)
            *reinterpret_cast<float4*>(dst + 32 * t + 8 * g + 4 * fh) =
                make_float4(dq[t][4 * g] * p.scale, dq[t][4 * g + 1] * p.scale, dq[t][4 * g + 2] * p.scale, dq[t][4 * g + 3] * p.scale);
}
template <int DT, int NW>
__global__ __launch_bounds__(64 * NW) void attn32_bwd_dkv_kernel(const Attn32BwdParams p) {
    constexpr int D = DT * 32, LDQ = D + 1, NT = 64 * NW, ROWS = 32 * NW;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* sK = reinterpret_cast<float*>(smem);            // [NW * 32][LDQ] K rows of the workgroup's keys
    float* sV = sK + ROWS * LDQ;
    float* sQ = sV + ROWS * LDQ;                            // [32][LDQ] query tile
    float* sO = sQ + 32 * LDQ;                             // dO tile
    float* sL = sO + 32 * LDQ;                             // [32] lse, [32] delta
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 31, fh = lane >> 5;
    const int b = blockIdx.z, head = blockIdx.y, kv0 = blockIdx.x * ROWS;
    const int C3 = 3 * p.C;
    const float* base = p.qkv + (size_t)b * p.N * C3 + head * D;
    const float* dob = p.d_o + (size_t)b * p.N * p.C + head * D;
    for (int w = 0; w < NW; ++w) {
        attn32_load_tile<D>(sK + w * 32 * LDQ, base + p.C, C3, kv0 + w * 32, p.N, tid, NT);
        attn32_load_tile<D>(sV + w * 32 * LDQ, base + 2 * p.C, C3, kv0 + w * 32, p.N, tid, NT);
    }
    const int ki = kv0 + wave * 32 + fr; const bool kok = ki < p.N;
    f32x16 dk[DT], dv[DT];
#pragma unroll
    for (int t = 0; t < DT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) { dk[t][r] = 0.f; dv[t][r] = 0.f; }
    for (int q0 = 0; q0 < p.N; q0 += 32) {
        __syncthreads();
        attn32_load_tile<D>(sQ, base, C3, q0, p.N, tid, NT);
        attn32_load_tile<D>(sO, dob, p.C, q0, p.N, tid, NT);
        if (tid < 32) {
            int qi = q0 + tid; const bool ok = qi < p.N; if (!ok) qi = p.N - 1;
            sL[tid] = ok ? p.lse[((size_t)b * p.heads + head) * p.N + qi] : INFINITY;           // exp(s - inf) = 0 masks the row
            sL[32 + tid] = p.delta[((size_t)b * p.heads + head) * p.N + qi];
        }
        __syncthreads();
        // S[q][key] = Q K^T and dP[q][key] = dO V^T with the key on the lane; register r <-> query (r & 3) + 8 (r >> 2) + 4 fh
        f32x16 s, dp;
#pragma unroll
        for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
        const float* qa = sQ + fr * LDQ + fh; const float* oa = sO + fr * LDQ + fh;
        const float* kb = sK + (wave * 32 + fr) * LDQ + fh; const float* vb = sV + (wave * 32 + fr) * LDQ + fh;
#pragma unroll 8
        for (int kk = 0; kk < D / 2; ++kk) {
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(qa[2 * kk], kb[2 * kk], s, 0, 0, 0);
            dp = __builtin_amdgcn_mfma_f32_32x32x2f32(oa[2 * kk], vb[2 * kk], dp, 0, 0, 0);
        }
        f32x16 pp;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int ql = (r & 3) + 8 * (r >> 2) + 4 * fh;
            const float pe = kok ? expf(s[r] * p.scale - sL[ql]) : 0.f;
            pp[r] = pe; s[r] = pe * (dp[r] - sL[32 + ql]);
        }
        // dV^T[dd][key] += dO^T[dd][q] P[q][key] ;  dK^T[dd][key] += Q^T[dd][q] dS[q][key]   (k = q, permuted as the registers hold it)
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int ql = (j & 3) + 8 * (j >> 2) + 4 * fh;
            const float* orow = sO + ql * LDQ + fr; const float* qrow = sQ + ql * LDQ + fr;
#pragma unroll
            for (int t = 0; t < DT; ++t) {
                dv[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(orow[32 * t], pp[j], dv[t], 0, 0, 0);
                dk[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(qrow[32 * t], s[j], dk[t], 0, 0, 0);
            }
        }
    }
    if (!kok) return;
    float* dst = p.dqkv + ((size_t)b * p.N + ki) * C3 + head * D;
#pragma unroll
    for (int t = 0; t < DT; ++t)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            *reinterpret_cast<float4*>(dst + p.C + 32 * t + 8 * g + 4 * fh) =
                make_float4(dk[t][4 * g] * p.scale, dk[t][4 * g + 1] * p.scale, dk[t][4 * g + 2] * p.scale, dk[t][4 * g + 3] * p.scale);
            *reinterpret_cast<float4*>(dst + 2 * p.C + 32 * t + 8 * g + 4 * fh) =
                make_float4(dv[t][4 * g], dv[t][4 * g + 1], dv[t][4 * g + 2], dv[t][4 * g + 3]);
        }
}
template <int DT, int NW>
static hipError_t launch_attn32_bwd_d(const Attn32BwdParams& p, hipStream_t s) {
    constexpr int D = DT * 32, ROWS = 32 * NW;
    constexpr int LDS_DQ = (2 * ROWS + 2 * 32) * (D + 1) * 4, LDS_DKV = (2 * ROWS + 2 * 32) * (D + 1) * 4 + 64 * 4;
    static_assert(LDS_DKV <= 160 * 1024, "LDS budget");
    static bool once = false;
    if (!once) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&attn32_bwd_dq_kernel<DT, NW>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_DQ);
        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&attn32_bwd_dkv_kernel<DT, NW>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_DKV);
        if (e != hipSuccess) return e;
        once = true;
    }
    const long dthreads = (long)p.B * p.N * p.heads * (D / 4);
    hipLaunchKernelGGL(attn32_delta_kernel, dim3((unsigned)((dthreads + 255) / 256)), dim3(256), 0, s, p);
    hipLaunchKernelGGL((attn32_bwd_dq_kernel<DT, NW>), dim3((p.N + ROWS - 1) / ROWS, p.heads, p.B), dim3(64 * NW), LDS_DQ, s, p);
    hipLaunchKernelGGL((attn32_bwd_dkv_kernel<DT, NW>), dim3((p.N + ROWS - 1) / ROWS, p.heads, p.B), dim3(64 * NW), LDS_DKV, s, p);
    return hipGetLastError();
}
static hipError_t launch_attn32_bwd(const Attn32BwdParams& p, hipStream_t s) {
    switch (p.d) {
        case 32: return launch_attn32_bwd_d<1, 4>(p, s);
        case 64: return launch_attn32_bwd_d<2, 4>(p, s);
        case 128: return launch_attn32_bwd_d<4, 2>(p, s);        // the AutoencoderKL's single-head blocks (d = C)
        case 256: return launch_attn32_bwd_d<8, 1>(p, s);
    }
    return hipErrorInvalidValue;
}

// ---- small ones -----------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void add_f32_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out, long nvec) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (long)gridDim.x * blockDim.x) {
        const float4 va = reinterpret_cast<const float4*>(a)[i], vb = reinterpret_cast<const float4*>(b)[i];
        reinterpret_cast<float4*>(out)[i] = make_float4(va.x + vb.x, va.y + vb.y, va.z + vb.z, va.w + vb.w);
    }
}
__global__ __launch_bounds__(256) void sumpool2_f32_kernel(const float* __restrict__ x, float* __restrict__ out, int N, int D, int H, int W, int C) {
    const int cvec = C / 4;
    const long total = (long)N * D * H * W * cvec;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int cv = (int)(i % cvec);
        long r = i / cvec;
        const int w = (int)(r % W); r /= W; const int h = (int)(r % H); r /= H; const int d = (int)(r % D); const int n = (int)(r / D);
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const size_t row = (((size_t)n * 2 * D + 2 * d + (t >> 2)) * 2 * H + 2 * h + ((t >> 1) & 1)) * 2 * W + 2 * w + (t & 1);
            const float4 v = *reinterpret_cast<const float4*>(x + row * C + cv * 4);
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
        *reinterpret_cast<float4*>(out + (size_t)i * 4) = s;
    }
}
// flipped + transposed weights of the data-gradient convs from the fp32 arena (WtDesc offsets are those of the bf16 arena: x 2 here)
__global__ __launch_bounds__(256) void weight_flip_transpose_batched_f32_kernel(const WtDesc* __restrict__ descs, const int2* __restrict__ blockmap,
                                                                                const char* __restrict__ arena32, char* __restrict__ ws) {
    __shared__ float tile[64][65];
    const int2 bm = blockmap[blockIdx.x];
    const WtDesc e = descs[bm.x];
    const int cols = (e.cout + 31) / 32 * 32;
    int b = bm.y;
    const int ct = b % e.col_tiles; b /= e.col_tiles;
    const int rt = b % e.row_tiles; const int tp = b / e.row_tiles;
    const int co0 = ct * 64, ci0 = rt * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const float* src = reinterpret_cast<const float*>(arena32 + 2 * e.src_off) + (size_t)(e.taps - 1 - tp) * e.cout_pad * e.cin;
#pragma unroll 4
    for (int r = ty; r < 64; r += 4) {
        const int co = co0 + r, ci = ci0 + tx;
        tile[r][tx] = (co < e.cout && ci < e.ci_cnt) ? src[(size_t)co * e.cin + e.ci_off + ci] : 0.f;
    }
    __syncthreads();
    float* dst = reinterpret_cast<float*>(ws + e.dst_off) + (size_t)tp * e.rows * cols;
#pragma unroll 4
    for (int r = ty; r < 64; r += 4) {
        const int ci = ci0 + r, co = co0 + tx;
        if (ci < e.rows && co < cols) dst[(size_t)ci * cols + co] = tile[tx][r];
    }
}
// time-embedding MLP backward with fp32 weights (stage 1 of linear_bwd_dx: partial sums over a slice of the output rows)
__global__ __launch_bounds__(256) void linear_bwd_dx_part_f32w_kernel(const float* __restrict__ W, const float* __restrict__ dy,
                                                                      float* __restrict__ part, int I, int O, int dy_stride, int B) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x, b = blockIdx.y, z = blockIdx.z, nz = gridDim.z;
    if (i >= I) return;
    const int per = (O + nz - 1) / nz, o0 = z * per;
    int o1 = o0 + per; if (o1 > O) o1 = O;
    float acc = 0.f;
#pragma unroll 4
    for (int o = o0; o < o1; ++o) acc += W[(size_t)o * I + i] * dy[(size_t)b * dy_stride + o];
    part[((size_t)z * B + b) * I + i] = acc;
}
// exact-expf variants of the time-embedding MLP backward helpers (the bf16 plans use __expf: fine at 1e-2, visible at 1e-5)
__global__ __launch_bounds__(256) void linear_bwd_dx_fold_f32_kernel(const float* __restrict__ part, const float* __restrict__ x_pre,
                                                                     float* __restrict__ dx, int I, int nz, int x_stride, int B, int silu_in) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x, b = blockIdx.y;
    if (i >= I) return;
    float acc = 0.f;
    for (int z = 0; z < nz; ++z) acc += part[((size_t)z * B + b) * I + i];
    if (silu_in) {
        const float u = x_pre[(size_t)b * x_stride + i];
        const float sg = 1.0f / (1.0f + expf(-u));
        acc *= sg * (1.0f + u * (1.0f - sg));
    }
    dx[(size_t)b * x_stride + i] = acc;
}
__global__ __launch_bounds__(256) void linear_bwd_dw_f32_kernel(const float* __restrict__ dy, const float* __restrict__ x_pre,
                                                                float* __restrict__ dW, float* __restrict__ db,
                                                                int B, int I, int O, int dy_stride, int x_stride, int silu_in) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)O * I) return;
    const int o = (int)(idx / I), i = (int)(idx - (long)o * I);
    float acc = 0.f, bsum = 0.f;
    for (int b = 0; b < B; ++b) {
        float xv = x_pre[(size_t)b * x_stride + i];
        if (silu_in) xv = xv / (1.0f + expf(-xv));
        const float d = dy[(size_t)b * dy_stride + o];
        acc += d * xv; bsum += d;
    }
    dW[idx] = acc;
    if (i == 0 && db) db[o] = bsum;
}
