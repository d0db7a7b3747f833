// 3x3x3 stride-1 "same" convolution with 64 output channels over a LARGE grid: the full-resolution level of the AutoencoderKL (64 -> 64 at
// 96^3 in the benchmark configuration, 144 x 176 x 112 in the configs[3] training step: 45 % of an encode, 39 % of a decode; the
// reference's Encoder / Decoder ResBlocks).
//
// conv3_halo_kernel's 254 x 64 tile runs these at 0.27 - 0.29 of the MFMA peak: with 64 couts every voxel fragment read from LDS feeds
// only two MFMAs per wave (0.5 fragment read per MFMA: the K loop is LDS-read bound), and the tile's rows are copied again for every
// (kd, kh): ~2 KB of LDS-DMA per output voxel.  Here a workgroup owns a TD x TH x 16 BLOCK of output voxels and copies its
// (TD + 2)(TH + 2) 18 input halo ONCE per 32 input channels (LDS-DMA, 64-byte voxel rows, XOR-swizzled so that the 16 voxel lanes of a
// fragment read hit 16 distinct 16-byte slots); every wave owns one d-slice = TH voxel tiles of 16 consecutive w and ALL 64 couts, so a
// voxel fragment feeds four MFMAs (0.25 read per MFMA) and the 27 taps re-read the block from LDS, not from L2: 2.1 x (TH = 8) the
// input bytes per output voxel instead of 9 x.  The weights (27 x 64 x Cin, L2-resident, shared by every workgroup) go straight from
// global memory into the A-side registers, one tap ahead.  Two (TH = 8: 68 KB of LDS) or three (TH = 4: 41 KB) workgroups per CU: one's
// copy phase sits under the others' multiplies.
//
// MFMA 16x16x32, A = weights with the cout rows permuted so that a lane ends with 16 CONSECUTIVE couts of one voxel (row i of cout tile
// ct <-> cout 16 (i >> 2) + 4 ct + (i & 3)), B = voxels.  Epilogue: bias, per-sample channel bias, residual, bf16 NDHWC store, and the
// GroupNorm partials of the stored (rounded) values, one row per block: [N * blocks][64][2].  Cin % 32 == 0, any D / H / W (ragged
// blocks: out-of-volume halo voxels come back as zeros from the buffer load, out-of-volume outputs are masked).
#pragma once
#include "common.h"
#include "conv_igemm.h"

struct BlockParams {
    const bf16_t* x; const bf16_t* w;      // x [N][D][H][W][Cin] bf16; w packed [27][64][Cin]
    const float* bias; const float* temb; int temb_stride;
    const bf16_t* residual;                // [N*D*H*W][64] or null
    bf16_t* out;                           // [N*D*H*W][64]
    float* stats;                          // [N * td * th * tw][64][2] or null
    int N, D, H, W, Cin;
    int td, th, tw;                        // blocks per dimension
};

constexpr int BLK_TD = 4, BLK_TW = 16;
template <int TH> struct BlkGeom {
    static constexpr int HD = BLK_TD + 2, HH = TH + 2, HW = BLK_TW + 2;
    static constexpr int HV = HD * HH * HW;                    // halo voxels (1080 | 648)
    static constexpr int PIECES = (HV + 15) / 16;              // 1 KiB LDS-DMA pieces of 16 voxel rows x 64 B
    static constexpr int PPW = (PIECES + 3) / 4;               // pieces per wave
    static constexpr int LDS = PPW * 4 * 1024;                 // 69632 | 45056 bytes (the halo block; the kernel adds 2 KiB for the GroupNorm fold)
    static constexpr int LDS_ALL = LDS + 2048;
};

// PERSIST: the grid is two workgroups per CU and each walks tiles b, b + gridDim.x, ...: the next tile's first halo copy is issued BEFORE the
// epilogue of the current one (the block area is free once every wave has read its last fragment; the GroupNorm fold has an area of its own), so
// that copy and the dispatch of a fresh workgroup sit under the stores.  PERSIST = false is the one-tile form: the default, because it is
// FASTER (255 vs 280 us at 96^3): the hardware dispatcher balances 1728 tiles over 512 slots better than a static 3-or-4 share does.
template <int TH, int DBG = 0, bool PERSIST = false>   // DBG: timing experiments only (LDM_BLOCK_DBG): 1 = weights of tap 0 for every tap, 2 = no fragment re-reads, 4 = no halo copies, 8 = a quarter of the MFMAs
__global__ __launch_bounds__(256, TH == 8 ? 2 : 3) void conv3_block_kernel(const BlockParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
    using G = BlkGeom<TH>;
    extern __shared__ __attribute__((aligned(16))) char smem[];      // [G::LDS halo block][2 KiB GroupNorm fold]
    typedef __attribute__((address_space(3))) void* lds_ptr_t;
    const int tid = threadIdx.x, lane = tid & 63, fr = lane & 15, fg = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ntiles = p.N * p.td * p.th * p.tw;
    const unsigned row_bytes = (unsigned)p.Cin * 2u;
    const unsigned x_bytes = (unsigned)p.N * p.D * p.H * p.W * row_bytes;
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, (int)x_bytes, 0x00020000);
    // A side: row fr of cout tile ct <-> cout 16 (fr >> 2) + 4 ct + (fr & 3); k chunk fg.  Buffer loads: the lane's byte offset stays fixed,
    // (tap, cout tile, channel chunk) ride in the scalar offset: no vector address arithmetic in the loop
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, (int)(27u * 64u * row_bytes), 0x00020000);
    const int w_vo = (int)((unsigned)(16 * (fr >> 2) + (fr & 3)) * row_bytes) + 16 * fg;
    const int wtap = 64 * (int)row_bytes;                       // bytes between two taps
    const int wct = 4 * (int)row_bytes;                         // bytes between two cout tiles
    // B side: halo voxel of (tile vt, tap): hv = hv_base + ((kd HH + kh + vt) HW + kw).  MFMA column fr <-> voxel w0 + wm of the tile:
    // ds_read_b128 is serviced in lane groups {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} (+32), i.e. eight lanes of chunk fg and eight of
    // chunk fg ^ 1; with the even voxels on columns 0-3 / 12-15 and the odd ones on 4-11, the four voxels of a group that share a bank
    // quarter (hv & 3) ask for the same logical chunk, which the (hv >> 2) & 3 swizzle spreads over four slots: conflict-free (4 LDS
    // cycles per read) at every alignment; columns in voxel order cost 8 (MI355X_MICROARCH.md, LDS)
    const int wm = fr < 4 ? 2 * fr : fr < 12 ? 2 * (fr - 4) + 1 : 2 * (fr - 8);
    const int hv_base = wave * G::HH * G::HW + wm;

    // tile state (wave-uniform).  A halo copy works out this lane's voxel of each of the wave's DMA pieces as it goes (nothing kept across
    // the K loop: the kernel sits at the 256-register line): piece q covers halo voxels 16 q .. 16 q + 15, lane = (voxel L >> 2, slot L & 3);
    // slot s of voxel hv holds the 16-byte channel chunk s ^ ((hv >> 2) & 3)
    int blk, n, d0, h0, w0;
#define BK_TILE_SETUP(B) do {                                                                         \
        int b_ = xcd_remap((B), ntiles); blk = b_;                                                    \
        const int bw_ = b_ % p.tw; b_ /= p.tw; const int bh_ = b_ % p.th; b_ /= p.th; const int bd_ = b_ % p.td; n = b_ / p.td; \
        d0 = bd_ * BLK_TD; h0 = bh_ * TH; w0 = bw_ * BLK_TW;                                          \
    } while (0)
#define BK_COPY(C0) do {                                                                              \
        int lane_c = lane; asm volatile("" : "+v"(lane_c));  /* opaque: or the per-piece decompositions are hoisted and held in registers */ \
        if (!(DBG & 4)) { _Pragma("unroll") for (int j = 0; j < G::PPW; ++j) {                        \
            const int hv = (wave * G::PPW + j) * 16 + (lane_c >> 2);                                  \
            const int wx = hv % G::HW, r = hv / G::HW, hy = r % G::HH, dz = r / G::HH;                \
            const int gd = d0 - 1 + dz, gh = h0 - 1 + hy, gw = w0 - 1 + wx;                           \
            const bool ok = hv < G::HV && (unsigned)gd < (unsigned)p.D && (unsigned)gh < (unsigned)p.H && (unsigned)gw < (unsigned)p.W; \
            const unsigned vox = (unsigned)(((n * p.D + gd) * p.H + gh) * p.W + gw);                  \
            const unsigned vo = ok ? vox * row_bytes + (unsigned)(((lane_c & 3) ^ ((hv >> 2) & 3)) << 4) : 0xFFFFFFFFu;   /* out of range: the load returns zeros */ \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, (lds_ptr_t)(smem + (wave * G::PPW + j) * 1024), 16, vo, (C0) * 2, 0, 0); \
        } }                                                                                           \
    } while (0)

    int tile = blockIdx.x;
    BK_TILE_SETUP(tile);
    BK_COPY(0);
    do {
        f32x4 acc[TH][4];
#pragma unroll
        for (int vt = 0; vt < TH; ++vt)
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) acc[vt][ct] = (f32x4){0.f, 0.f, 0.f, 0.f};

        for (int c0 = 0; c0 < p.Cin; c0 += 32) {
            if (c0 > 0) {
                __syncthreads();                              // the previous chunk's fragment reads are done
                BK_COPY(c0);
            }
            // weights: three register sets, tap t in set t % 3, loaded two taps ahead (one tap of 32 MFMAs is ~0.25 us, an L2 hit ~0.5 us)
            bf16x8 wf[3][4];
#define BK_WLOAD(SET, TAP) do {                                                                      \
        const int t_ = (DBG & 1) ? 0 : (TAP) < 27 ? (TAP) : 26;   /* clamped: no branch in the hot block */ \
        const int so_ = t_ * wtap + c0 * 2;                                                          \
        _Pragma("unroll") for (int ct = 0; ct < 4; ++ct)                                             \
            wf[SET][ct] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rs_w, w_vo, so_ + ct * wct, 0)); \
    } while (0)
            BK_WLOAD(0, 0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            BK_WLOAD(1, 1);
            // voxel fragments: a rolling pipeline over (tap, tile): right after the four MFMAs of tile vt are issued, xf[vt] is re-read for the
            // NEXT tap, so every LDS read has the 28 MFMAs of the other seven tiles to land in
#define BK_READ(HV) (*reinterpret_cast<const bf16x8*>(smem + (HV) * 64 + ((fg ^ (((HV) >> 2) & 3)) << 4)))
            bf16x8 xf[TH];
#pragma unroll
            for (int vt = 0; vt < TH; ++vt) { const int hv = hv_base + vt * G::HW; xf[vt] = BK_READ(hv); }
#pragma unroll 1
            for (int it = 0; it < 9; ++it) {                      // it = 3 kd + kh (wave-uniform)
                const int kd = it / 3, kh = it - 3 * kd;
                const int itn = it < 8 ? it + 1 : 8, kdn = itn / 3, khn = itn - 3 * kdn;
                const int hv_it = hv_base + (kd * G::HH + kh) * G::HW, hv_nx = hv_base + (kdn * G::HH + khn) * G::HW;
#pragma unroll
                for (int kw = 0; kw < 3; ++kw) {
                    BK_WLOAD((kw + 2) % 3, 3 * it + kw + 2);
                    const int hv_n = kw < 2 ? hv_it + kw + 1 : hv_nx;     // first tile of the next tap (the last tap re-reads its own rows)
#pragma unroll
                    for (int vt = 0; vt < TH; ++vt) {
#pragma unroll
                        for (int ct = 0; ct < 4; ++ct) if (!(DBG & 8) || ct == 0) acc[vt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[kw][ct], xf[vt], acc[vt][ct], 0, 0, 0);
                        const int hv = hv_n + vt * G::HW;
                        if (!(DBG & 2)) xf[vt] = BK_READ(hv);
                    }
                    // one scheduling region per tap: the four weight loads up front, then (4 MFMA, 1 LDS read) per tile
                    __builtin_amdgcn_sched_group_barrier(0x020, 4, 0);
#pragma unroll
                    for (int vt = 0; vt < TH; ++vt) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
                        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    }
                }
            }
#undef BK_READ
#undef BK_WLOAD
        }

        // ---- epilogue: lane = voxel (d0 + wave, h0 + vt, w0 + wm), couts 16 fg .. 16 fg + 15 ---------------------------------------
        // the epilogue's lane constants pass through an opaque move in the loop form: otherwise the compiler hoists everything that depends only on
        // them (the 16 bias values, store offsets) out of the tile loop and holds it in registers across the K loop (283 spilled registers)
        int fg_e = fg, wm_e = wm, fr_e = fr;
        if (PERSIST) asm volatile("" : "+v"(fg_e), "+v"(wm_e), "+v"(fr_e));
        const int cb_e = 16 * fg_e;
        const int e_blk = blk, e_n = n, e_h0 = h0, gd = d0 + wave, gw = w0 + wm_e;
        bool more = false;
        if (PERSIST) {
            const int nxt = tile + (int)gridDim.x;
            more = nxt < ntiles;
            if (more) {
                tile = nxt;
                BK_TILE_SETUP(tile);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();                 // every wave has read its last fragment of this tile: the block area is free
                BK_COPY(0);
            }
        }
        float add[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) add[q] = (p.bias ? p.bias[cb_e + q] : 0.f) + (p.temb ? p.temb[(size_t)e_n * p.temb_stride + cb_e + q] : 0.f);
        float ssum[16], ssq[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) { ssum[q] = 0.f; ssq[q] = 0.f; }
        const bool col_ok = gd < p.D && gw < p.W;
#pragma unroll
        for (int vt = 0; vt < TH; ++vt) {
            const int gh = e_h0 + vt;
            if (!col_ok || gh >= p.H) continue;
            const size_t m = (size_t)((e_n * p.D + gd) * p.H + gh) * p.W + gw;
            float v[16];
#pragma unroll
            for (int ct = 0; ct < 4; ++ct)
#pragma unroll
                for (int r = 0; r < 4; ++r) v[4 * ct + r] = acc[vt][ct][r] + add[4 * ct + r];
            if (p.residual) {
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const u32x4 rv = *reinterpret_cast<const u32x4*>(p.residual + m * 64 + cb_e + 8 * h);
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        v[h * 8 + 2 * q] += __uint_as_float(rv[q] << 16);
                        v[h * 8 + 2 * q + 1] += __uint_as_float(rv[q] & 0xffff0000u);
                    }
                }
            }
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                u32x4 o;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    o[q] = pack2bf(v[h * 8 + 2 * q], v[h * 8 + 2 * q + 1]);
                    const float lo = __uint_as_float(o[q] << 16), hi = __uint_as_float(o[q] & 0xffff0000u);
                    ssum[h * 8 + 2 * q] += lo; ssq[h * 8 + 2 * q] += lo * lo;
                    ssum[h * 8 + 2 * q + 1] += hi; ssq[h * 8 + 2 * q + 1] += hi * hi;
                }
                *reinterpret_cast<u32x4*>(p.out + m * 64 + cb_e + 8 * h) = o;
            }
        }
        if (p.stats) {
            // sum over the 16 voxel lanes of each DPP row (lanes sharing fg), then over the four waves through LDS (fixed order: reproducible).
            // Raw barriers with an LDS-only wait: a __syncthreads here would also wait for the next tile's halo copy
#define BK_ROW_ADD(X, CTRL) X += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(X), CTRL, 0xf, 0xf, true))
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                BK_ROW_ADD(ssum[q], 0x128); BK_ROW_ADD(ssum[q], 0x124); BK_ROW_ADD(ssum[q], 0x122); BK_ROW_ADD(ssum[q], 0x121);
                BK_ROW_ADD(ssq[q], 0x128); BK_ROW_ADD(ssq[q], 0x124); BK_ROW_ADD(ssq[q], 0x122); BK_ROW_ADD(ssq[q], 0x121);
            }
#undef BK_ROW_ADD
            float* red = reinterpret_cast<float*>(smem + G::LDS);      // [wave][64][2], its own 2 KiB
            if (fr_e == 0) {
#pragma unroll
                for (int q = 0; q < 16; ++q) { red[(wave * 64 + cb_e + q) * 2] = ssum[q]; red[(wave * 64 + cb_e + q) * 2 + 1] = ssq[q]; }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (tid < 128) {
                const float t = (red[tid] + red[128 + tid]) + (red[256 + tid] + red[384 + tid]);
                p.stats[(size_t)e_blk * 128 + tid] = t;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                     // the fold area may be written again (next tile)
            asm volatile("" ::: "memory");
        }
        if (!PERSIST || !more) break;
    } while (true);
#undef BK_TILE_SETUP
#undef BK_COPY
#endif
}

// ---- the same block scheme for 128 OUTPUT channels (the AutoencoderKL's half-resolution level: 128 -> 128 at 48^3, 19 % of an encode + decode on
// conv3_halo_kernel's 126 x 128 tile at 0.28 - 0.31 of the MFMA peak).  One workgroup of EIGHT waves per CU: wave = (d-slice w & 3, cout half
// w >> 2), so two waves read every voxel fragment (still 0.25 LDS reads per MFMA per wave) and each loads the weights of its 64 couts.  With one
// workgroup per CU nobody else covers a copy phase, so the halo chunk is DOUBLE-BUFFERED (2 x 72 KiB): the copy of chunk c + 1 is issued at the
// top of the K loop of chunk c (nine 1 KiB pieces per wave; spreading them one per (kd, kh) over the loop measured the same: 130.5 us either
// way, the copies are not what this kernel waits for).  Only a tile's first chunk is exposed.
// Measured at 48^3 (tools/bench_conv_block.py, random data): 64 -> 128: 70 vs 80 us on the halo tile; 128 -> 128: 131 vs 125; 256 -> 128: 243 vs
// 212 -- per MFMA it runs at the 64-cout kernel's rate (0.30 of peak), the halo tile gains with K.  Inside the AutoencoderKL (same-box A/B,
// Cin <= 128 here): encode 2.37 -> 2.34 ms, decode 3.30 -> 3.27 ms (fewer GroupNorm partial rows behind it); at the configs[3] patch (693 tiles =
// 2.7 rounds of one workgroup per CU) it LOSES 3.5 %, so the plans use it only where the tiles fit one round (ldm3d.hip).
#ifndef BLK128_BURST_AT
#define BLK128_BURST_AT 0
#endif
struct Blk128 {
    static constexpr int TH = 8;
    using G = BlkGeom<8>;
    static constexpr int PPW = 9;                              // pieces per wave per chunk: 8 x 9 = 72 >= 68
    static constexpr int BUF = 8 * PPW * 1024;                 // 73728 bytes per buffer (pieces 68 .. 71 hold zeros)
    static constexpr int LDS_ALL = 2 * BUF + 8 * 64 * 2 * 4;   // + the GroupNorm fold [wave][64][2]: 151552 bytes
};

__global__ __launch_bounds__(512, 1) void conv3_block128_kernel(const BlockParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
    using G = Blk128::G;
    constexpr int TH = 8;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef __attribute__((address_space(3))) void* lds_ptr_t;
    const int tid = threadIdx.x, lane = tid & 63, fr = lane & 15, fg = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ds = wave & 3, ch = wave >> 2;
    const int ntiles = p.N * p.td * p.th * p.tw;
    const unsigned row_bytes = (unsigned)p.Cin * 2u;
    const unsigned x_bytes = (unsigned)p.N * p.D * p.H * p.W * row_bytes;
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, (int)x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, (int)(27u * 128u * row_bytes), 0x00020000);
    const int w_vo = (int)((unsigned)(64 * ch + 16 * (fr >> 2) + (fr & 3)) * row_bytes) + 16 * fg;
    const int wtap = 128 * (int)row_bytes;
    const int wct = 4 * (int)row_bytes;
    const int wm = fr < 4 ? 2 * fr : fr < 12 ? 2 * (fr - 4) + 1 : 2 * (fr - 8);       // column map of conv3_block_kernel
    const int hv_base = ds * G::HH * G::HW + wm;

    int b_ = xcd_remap(blockIdx.x, ntiles);
    const int blk = b_;
    const int bw_ = b_ % p.tw; b_ /= p.tw; const int bh_ = b_ % p.th; b_ /= p.th; const int bd_ = b_ % p.td; const int n = b_ / p.td;
    const int d0 = bd_ * BLK_TD, h0 = bh_ * TH, w0 = bw_ * BLK_TW;
    // piece J of this wave (halo voxels 16 q .. 16 q + 15, q = wave + 8 J) of channel chunk C0 into buffer at byte offset BO
#define B8_PIECE(BO, C0, J) do {                                                                      \
        int lane_c = lane; asm volatile("" : "+v"(lane_c));                                           \
        const int q_ = wave + 8 * (J);                                                                \
        const int hv = q_ * 16 + (lane_c >> 2);                                                       \
        const int wx = hv % G::HW, r = hv / G::HW, hy = r % G::HH, dz = r / G::HH;                    \
        const int gd = d0 - 1 + dz, gh = h0 - 1 + hy, gw = w0 - 1 + wx;                               \
        const bool ok = hv < G::HV && (unsigned)gd < (unsigned)p.D && (unsigned)gh < (unsigned)p.H && (unsigned)gw < (unsigned)p.W; \
        const unsigned vox = (unsigned)(((n * p.D + gd) * p.H + gh) * p.W + gw);                      \
        const unsigned vo = ok ? vox * row_bytes + (unsigned)(((lane_c & 3) ^ ((hv >> 2) & 3)) << 4) : 0xFFFFFFFFu; \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, (lds_ptr_t)(smem + (BO) + q_ * 1024), 16, vo, (C0) * 2, 0, 0); \
    } while (0)

    f32x4 acc[TH][4];
#pragma unroll
    for (int vt = 0; vt < TH; ++vt)
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) acc[vt][ct] = (f32x4){0.f, 0.f, 0.f, 0.f};

#pragma unroll
    for (int j = 0; j < Blk128::PPW; ++j) B8_PIECE(0, 0, j);
    int cur = 0;                                              // byte offset of the buffer being read
    for (int c0 = 0; c0 < p.Cin; c0 += 32) {
        const bool more = c0 + 32 < p.Cin;                    // wave-uniform
        const int nxt = Blk128::BUF - cur;
        bf16x8 wf[3][4];
#define B8_WLOAD(SET, TAP) do {                                                                       \
        const int t_ = (TAP) < 27 ? (TAP) : 26;                                                       \
        const int so_ = t_ * wtap + c0 * 2;                                                           \
        _Pragma("unroll") for (int ct = 0; ct < 4; ++ct)                                              \
            wf[SET][ct] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rs_w, w_vo, so_ + ct * wct, 0)); \
    } while (0)
        B8_WLOAD(0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this chunk's pieces (all issued during the previous chunk) and the first weights
        __syncthreads();                                      // ... of every wave; and every wave is done with the other buffer
        B8_WLOAD(1, 1);
#define B8_READ(HV) (*reinterpret_cast<const bf16x8*>(smem + cur + (HV) * 64 + ((fg ^ (((HV) >> 2) & 3)) << 4)))
        bf16x8 xf[TH];
#pragma unroll
        for (int vt = 0; vt < TH; ++vt) { const int hv = hv_base + vt * G::HW; xf[vt] = B8_READ(hv); }
#pragma unroll 1
        for (int it = 0; it < 9; ++it) {
            const int kd = it / 3, kh = it - 3 * kd;
            const int itn = it < 8 ? it + 1 : 8, kdn = itn / 3, khn = itn - 3 * kdn;
            const int hv_it = hv_base + (kd * G::HH + kh) * G::HW, hv_nx = hv_base + (kdn * G::HH + khn) * G::HW;
            if (more && it == BLK128_BURST_AT) {               // the next chunk's copy, all nine pieces of this wave at once (see above)
#pragma unroll
                for (int j = 0; j < Blk128::PPW; ++j) B8_PIECE(nxt, c0 + 32, j);
            }
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                B8_WLOAD((kw + 2) % 3, 3 * it + kw + 2);
                const int hv_n = kw < 2 ? hv_it + kw + 1 : hv_nx;
#pragma unroll
                for (int vt = 0; vt < TH; ++vt) {
#pragma unroll
                    for (int ct = 0; ct < 4; ++ct) acc[vt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[kw][ct], xf[vt], acc[vt][ct], 0, 0, 0);
                    const int hv = hv_n + vt * G::HW;
                    xf[vt] = B8_READ(hv);
                }
                __builtin_amdgcn_sched_group_barrier(0x020, 4, 0);
#pragma unroll
                for (int vt = 0; vt < TH; ++vt) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
            }
        }
#undef B8_READ
#undef B8_WLOAD
        cur = nxt;
    }
#undef B8_PIECE

    // ---- epilogue: lane = voxel (d0 + ds, h0 + vt, w0 + wm), couts 64 ch + 16 fg .. + 15 of 128 ---------------------------------
    const int cb = 64 * ch + 16 * fg;
    float add[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) add[q] = (p.bias ? p.bias[cb + q] : 0.f) + (p.temb ? p.temb[(size_t)n * p.temb_stride + cb + q] : 0.f);
    float ssum[16], ssq[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) { ssum[q] = 0.f; ssq[q] = 0.f; }
    const int gd = d0 + ds, gw = w0 + wm;
    const bool col_ok = gd < p.D && gw < p.W;
#pragma unroll
    for (int vt = 0; vt < TH; ++vt) {
        const int gh = h0 + vt;
        if (!col_ok || gh >= p.H) continue;
        const size_t m = (size_t)((n * p.D + gd) * p.H + gh) * p.W + gw;
        float v[16];
#pragma unroll
        for (int ct = 0; ct < 4; ++ct)
#pragma unroll
            for (int r = 0; r < 4; ++r) v[4 * ct + r] = acc[vt][ct][r] + add[4 * ct + r];
        if (p.residual) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const u32x4 rv = *reinterpret_cast<const u32x4*>(p.residual + m * 128 + cb + 8 * h);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    v[h * 8 + 2 * q] += __uint_as_float(rv[q] << 16);
                    v[h * 8 + 2 * q + 1] += __uint_as_float(rv[q] & 0xffff0000u);
                }
            }
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            u32x4 o;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                o[q] = pack2bf(v[h * 8 + 2 * q], v[h * 8 + 2 * q + 1]);
                const float lo = __uint_as_float(o[q] << 16), hi = __uint_as_float(o[q] & 0xffff0000u);
                ssum[h * 8 + 2 * q] += lo; ssq[h * 8 + 2 * q] += lo * lo;
                ssum[h * 8 + 2 * q + 1] += hi; ssq[h * 8 + 2 * q + 1] += hi * hi;
            }
            *reinterpret_cast<u32x4*>(p.out + m * 128 + cb + 8 * h) = o;
        }
    }
    if (p.stats) {
#define BK_ROW_ADD(X, CTRL) X += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(X), CTRL, 0xf, 0xf, true))
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            BK_ROW_ADD(ssum[q], 0x128); BK_ROW_ADD(ssum[q], 0x124); BK_ROW_ADD(ssum[q], 0x122); BK_ROW_ADD(ssum[q], 0x121);
            BK_ROW_ADD(ssq[q], 0x128); BK_ROW_ADD(ssq[q], 0x124); BK_ROW_ADD(ssq[q], 0x122); BK_ROW_ADD(ssq[q], 0x121);
        }
#undef BK_ROW_ADD
        float* red = reinterpret_cast<float*>(smem + 2 * Blk128::BUF);      // [wave][64][2]
        if (fr == 0) {
#pragma unroll
            for (int q = 0; q < 16; ++q) { red[(wave * 64 + 16 * fg + q) * 2] = ssum[q]; red[(wave * 64 + 16 * fg + q) * 2 + 1] = ssq[q]; }
        }
        __syncthreads();
        if (tid < 256) {                                      // tid = (cout 0 .. 127, sum | sum of squares): fold the four d-slice waves of the cout half
            const int c = tid >> 1, k = tid & 1, h2 = c >> 6, cc = c & 63;
            const float* r0 = red + ((h2 * 4) * 64 + cc) * 2 + k;
            const float t = (r0[0] + r0[128]) + (r0[256] + r0[384]);
            p.stats[(size_t)blk * 256 + tid] = t;
        }
    }
#endif
}

static inline int conv_block128_max_cin() { static const int c = ldm_xknob("LDM_CONV_BLOCK128_MAX_CIN", 128); return c; }
static inline bool conv_block128_enabled() { static const int on = ldm_knob("LDM_CONV_BLOCK128", 1); return on != 0; }
static inline bool conv_block_enabled() { static const int on = ldm_knob("LDM_CONV_BLOCK", 1); return on != 0; }
static inline int conv_block_th() { static const int th = ldm_xknob("LDM_CONV_BLOCK_TH", 8); return th == 4 ? 4 : 8; }
// 128 -> 64 channels at 96^3 alone: 464 us here against 451 on the 254 x 64 halo tile (four channel chunks = four exposed copy phases), but inside
// the AutoencoderKL decode the plan with it is FASTER (3.32 vs 3.39 ms, same box: half as many GroupNorm partial rows for the norm that follows,
// and the halo tile's persistent grid loses its balance next to it), so the plans send Cin <= 128 here.  LDM_CONV_BLOCK_MAX_CIN overrides.
static inline int conv_block_max_cin() { static const int c = ldm_xknob("LDM_CONV_BLOCK_MAX_CIN", 128); return c; }
