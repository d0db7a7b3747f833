// libldm3d.so - host side: launch plans, weight arena and the C ABI declared in include/ldm3d.h.
//
// The reference (sanazkaviani/3d-latent-diffusion-model) builds its networks by instantiating MONAI classes from
// a JSON "_target_" (3d_ldm/utils.py:243-246) and runs them as a tree of nn.Modules.  Here a model is a flat,
// static LAUNCH PLAN over NDHWC bf16 tensors in one caller-owned workspace: channel concat, nearest upsampling,
// zero padding and 1x1 skip convolutions are addressing modes / extra K steps of the conv kernel, the time-
// embedding bias, conv bias and residual add are conv epilogues, and every ResBlock's time projection is one
// batched GEMV.  Plans are cached per input shape; nothing is allocated or synchronised on the step path.
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "../../include/ldm3d.h"
#include "attention.h"
#include "conv_igemm.h"
#include "conv_halo.h"
#ifdef LDM_EXPERIMENTS
#include "conv_halo_pp.h"                // alternating K steps per wave group, one barrier per six K steps (experiments builds)
#include "conv_halo_rw.h"                // register-fed weights: built, parity-green, slower
#endif
#include "conv_thin.h"
#include "gemm_light.h"
#include "gemm_light_x3.h"
#include "conv_block.h"
#include "conv_wgrad.h"
#include "conv_wgrad_w16.h"            // sixteen-wave form of the weight gradient for the wide-channel convs
#include "conv_wgrad_kw3.h"            // 3^3 stride-1 weight gradient, three kw taps per workgroup over one shared X tile
#include "norm_elem.h"
#include "fin_gn.h"
#include "f32_path.h"
#include "f32_train.h"

#ifndef CONV_STAGES
#define CONV_STAGES 4     // depth of the conv kernel's LDS ring (prefetch distance = stages - 1 K steps)
#endif

// ================================================================================================ errors
static thread_local char g_err[1024] = "";
static int fail(int code, const char* fmt, ...) {
    va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof g_err, fmt, ap); va_end(ap);
    return code;
}
#define HIP_TRY(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) \
    return fail(LDM_ERR_HIP, "%s failed: %s (%s:%d)", #x, hipGetErrorString(e_), __FILE__, __LINE__); } while (0)
#define LDM_TRY(x) do { int r_ = (x); if (r_ != 0) return r_; } while (0)

// "has hipFuncSetAttribute been called for this kernel" is a per-DEVICE fact: launchers keep a flag per device ordinal
static inline bool& attr_flag(bool (&tab)[32]) { int d = 0; if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= 32) d = 0; return tab[d]; }

#ifdef LDM_EXPERIMENTS
static int g_block_slots = 0;                                  // ldm_debug_conv_block_slots: tests walk the tile loop on small volumes
#endif
static hipError_t launch_conv_block(const BlockParams& q0, int TH, hipStream_t s) {
    BlockParams q = q0;
    q.td = (q.D + BLK_TD - 1) / BLK_TD; q.th = (q.H + TH - 1) / TH; q.tw = (q.W + BLK_TW - 1) / BLK_TW;
    const long tiles = (long)q.N * q.td * q.th * q.tw;
    static const int dbg = ldm_xknob("LDM_BLOCK_DBG", 0);
    // the tile-loop form (two workgroups per CU walk the tiles, the next tile's first halo copy issued before the epilogue) lives in experiments
    // builds only (EXTRA=-DLDM_EXPERIMENTS, LDM_CONV_BLOCK_PERSIST=1): 280 vs 255 us at 96^3 (AutoencoderKL encode 2.49 vs 2.35 ms).  A static
    // share of 3 or 4 tiles per workgroup loses what the hardware dispatcher gives the one-tile form for free: the next tile goes to whichever CU
    // frees a slot first.
    static const int persist = ldm_xknob("LDM_CONV_BLOCK_PERSIST", 0);
    static int cus_tab[32] = {};
    int dev_ = 0; if (hipGetDevice(&dev_) != hipSuccess || dev_ < 0 || dev_ >= 32) dev_ = 0;
    if (!cus_tab[dev_]) { hipDeviceProp_t pr; if (hipGetDeviceProperties(&pr, dev_) != hipSuccess) return hipErrorUnknown; cus_tab[dev_] = pr.multiProcessorCount / 8 * 8; if (cus_tab[dev_] < 8) cus_tab[dev_] = 8; }
    static bool attr_tab[32] = {}; bool& attr_set = attr_flag(attr_tab);
    if (!attr_set) {
#define BLK_ATTR(...) { hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3_block_kernel<__VA_ARGS__>), hipFuncAttributeMaxDynamicSharedMemorySize, BlkGeom<8>::LDS_ALL); if (e != hipSuccess) return e; }
        BLK_ATTR(8, 0, false) BLK_ATTR(4, 0, false)
#ifdef LDM_EXPERIMENTS
        BLK_ATTR(8, 0, true)
#endif
#ifdef LDM_BLOCK_EXPERIMENTS
        BLK_ATTR(8, 1, false) BLK_ATTR(8, 2, false) BLK_ATTR(8, 4, false) BLK_ATTR(8, 3, false) BLK_ATTR(8, 7, false) BLK_ATTR(8, 8, false) BLK_ATTR(8, 15, false)
#endif
#undef BLK_ATTR
        attr_set = true;
    }
    const unsigned grid = (unsigned)tiles;
#ifdef LDM_BLOCK_EXPERIMENTS
#define BLK_DBG(D) if (TH == 8 && dbg == D) { hipLaunchKernelGGL((conv3_block_kernel<8, D, false>), dim3(grid), dim3(256), BlkGeom<8>::LDS_ALL, s, q); return hipGetLastError(); }
    BLK_DBG(1) BLK_DBG(2) BLK_DBG(4) BLK_DBG(3) BLK_DBG(7) BLK_DBG(8) BLK_DBG(15)
#undef BLK_DBG
#endif
    (void)dbg; (void)persist;
#ifdef LDM_EXPERIMENTS
    const long slots = g_block_slots > 0 ? g_block_slots : 2L * cus_tab[dev_];     // two workgroups per CU
    if (TH == 8 && persist && tiles > slots) { hipLaunchKernelGGL((conv3_block_kernel<8, 0, true>), dim3((unsigned)slots), dim3(256), BlkGeom<8>::LDS_ALL, s, q); return hipGetLastError(); }
#endif
    if (TH == 8) hipLaunchKernelGGL((conv3_block_kernel<8, 0, false>), dim3(grid), dim3(256), BlkGeom<8>::LDS_ALL, s, q);
    else hipLaunchKernelGGL((conv3_block_kernel<4, 0, false>), dim3(grid), dim3(256), BlkGeom<4>::LDS_ALL, s, q);
    return hipGetLastError();
}
// CUs of the current device (256 when there is none: the planner also runs on hosts without a GPU, tests/test_abi_and_host.py)
static int device_cus() {
    static int tab[32] = {};
    int d = 0; if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= 32) return 256;
    if (!tab[d]) { hipDeviceProp_t pr; tab[d] = (hipGetDeviceProperties(&pr, d) == hipSuccess && pr.multiProcessorCount >= 8) ? pr.multiProcessorCount : 256; }
    return tab[d];
}
static hipError_t launch_conv_block128(const BlockParams& q0, hipStream_t s) {
    BlockParams q = q0;
    q.td = (q.D + BLK_TD - 1) / BLK_TD; q.th = (q.H + 7) / 8; q.tw = (q.W + BLK_TW - 1) / BLK_TW;
    static bool attr_tab[32] = {}; bool& attr_set = attr_flag(attr_tab);
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3_block128_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, Blk128::LDS_ALL);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    hipLaunchKernelGGL(conv3_block128_kernel, dim3((unsigned)((long)q.N * q.td * q.th * q.tw)), dim3(512), Blk128::LDS_ALL, s, q);
    return hipGetLastError();
}
static inline int rup(int v, int m) { return (v + m - 1) / m * m; }
static inline size_t rup_sz(size_t v, size_t m) { return (v + m - 1) / m * m; }

static inline uint16_t host_f2bf(float f) {            // round-to-nearest-even, NaN stays NaN
    uint32_t u; memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}

// ================================================================================================ plan IR
enum BaseId { BASE_NULL = 0, BASE_WS, BASE_W, BASE_IO0, BASE_IO1, BASE_IO2, BASE_IO3, BASE_IO4, BASE_IO5, BASE_W32, BASE_TAPO, BASE_TAPI, BASE_TTAB, BASE_SST, BASE_COUNT };   // TTAB: tabulated time-embedding rows (model-owned), SST: the sampler's device state
struct Ref { int base = BASE_NULL; size_t off = 0; };
static inline Ref ws_ref(size_t off) { Ref r; r.base = BASE_WS; r.off = off; return r; }
static inline Ref w_ref(size_t off) { Ref r; r.base = BASE_W; r.off = off; return r; }
static inline Ref io_ref(int i) { Ref r; r.base = BASE_IO0 + i; r.off = 0; return r; }
// fp32 precision mode: the matrix at byte offset `off` of the bf16 arena lives unrounded at 2 * off of the fp32 arena
static inline Ref w32_ref(size_t off) { Ref r; r.base = BASE_W32; r.off = 2 * off; return r; }

struct Act {                                         // NDHWC bf16 activation living in the workspace
    size_t off = 0; int C = 0; int N = 0, D = 0, H = 0, W = 0; bool valid = false;
    size_t stats_off = 0; bool has_stats = false;   // GroupNorm partials [blocks][C][2] written by the producer
    int stats_nrb = 0;                               // blocks per sample (0: one per 32 rows of the whole tensor)
    int esz = 2;                                     // bytes per element: 2 = bf16, 4 = fp32 (LDM_PREC_FP32 plans)
    bool hl = false;                                 // fp32 plans: the tensor is stored as [rows][hi(C) | lo(C)] bf16 (same bytes), for the 3 x bf16 halo conv
    size_t cs_off = 0; int cs_rows = 0;              // gradient tensors: column-sum partials [N][cs_rows][C][2] left by the GroupNorm backward that wrote it
    size_t bytes() const { return (size_t)N * D * H * W * C * esz; }
    long rows() const { return (long)N * D * H * W; }
};

enum OpKind { OP_PACK, OP_CONV, OP_FINALIZE, OP_GN_STATS, OP_GN_FINALIZE, OP_GN_PREP, OP_GN_APPLY, OP_ATTN, OP_SINUSOID,
              OP_GEMV, OP_VAE_HEADS, OP_GN_FUSED,
              // backward (training plans only)
              OP_WT, OP_WT_BATCH, OP_WGRAD, OP_EXPORT, OP_EXPORT_BATCH, OP_COLSUM, OP_GNB, OP_ATTN_BWD, OP_ADD, OP_SUMPOOL, OP_LIN_DX, OP_LIN_DW, OP_VAE_HEADS_BWD,
              OP_GEMM_LIGHT,                       // 1x1 convolution with short K (gemm_light.h)
              OP_COLSUM_BATCH,                     // every column-sum finalize of a backward plan in one launch
              OP_IM2COL,                           // fp32 NCDHW inputs -> bf16 patch matrix of the first conv (pack_im2col_kernel)
              // fp32 precision mode (f32_path.h)
              OP_PACK32, OP_CONV32, OP_FIN32, OP_GN_STATS32, OP_GN_APPLY32, OP_ATTN32, OP_GEMV32,
              OP_TAP,                              // debug tap: export an activation as fp32 NCDHW and / or overwrite it (teacher forcing)
              OP_BUCKET, OP_BUCKET_JOIN,           // training plans: a tail range of the flat gradient buffer is final (all-reduce it now) / wait for every bucket
              OP_UPS_SPLIT32,                      // fp32 precision: nearest x2 upsample into the (hi | lo) bf16 split (upsample_split_f32_kernel)
              OP_CONV_THIN,                        // 3^3 conv with Cout <= 4 and fp32 NCDHW output: the networks' last layer (conv_thin.h)
              OP_FIN_GN,                           // split-K finalize + the GroupNorm(+SiLU) that consumes it, one launch (fin_gn.h)
              OP_GEMM_LIGHT32,
              OP_CONV_BLOCK,                       // 3^3 conv with 64 output channels over a large grid, one halo block in LDS per workgroup (conv_block.h)
              OP_TEMB_ROW };                       // denoise-step plans: the time-embedding projections of the sampler's current step, copied from the table (temb_row_kernel)
             //                   // fp32 precision: 1x1 convolution as the light GEMM on fp32 operands split in registers (gemm_light_x3.h)

struct ConvCfg { int wgm, wgn, bk, splitk; int halo = 0, mtps = 0, qps = 0; int slab_lg = 0; };   // halo: conv3_halo_kernel (126-row tiles); slab_lg: planar split-K slabs (fin_gn.h)

struct Op {
    OpKind kind;
    // generic refs; meaning depends on kind
    Ref r[14];
    int i[24];
    float f[2];
    ConvCfg cc;
};

struct Pool {                                        // plan-time workspace allocator (first fit + coalescing)
    struct Blk { size_t off, size; };
    std::vector<Blk> free_list;
    std::map<size_t, size_t> live;
    size_t top = 0, high = 0;
    size_t alloc(size_t bytes) {
        bytes = rup_sz(bytes, 256);
        for (size_t k = 0; k < free_list.size(); ++k)
            if (free_list[k].size >= bytes) {
                size_t off = free_list[k].off;
                if (free_list[k].size == bytes) free_list.erase(free_list.begin() + k);
                else { free_list[k].off += bytes; free_list[k].size -= bytes; }
                live[off] = bytes;
                return off;
            }
        size_t off = top; top += bytes; high = std::max(high, top); live[off] = bytes;
        return off;
    }
    void release(size_t off) {
        auto it = live.find(off); if (it == live.end()) return;
        Blk b{off, it->second}; live.erase(it);
        free_list.push_back(b);
        std::sort(free_list.begin(), free_list.end(), [](const Blk& a, const Blk& c) { return a.off < c.off; });
        for (size_t k = 0; k + 1 < free_list.size();)
            if (free_list[k].off + free_list[k].size == free_list[k + 1].off) {
                free_list[k].size += free_list[k + 1].size; free_list.erase(free_list.begin() + k + 1);
            } else ++k;
        if (!free_list.empty() && free_list.back().off + free_list.back().size == top) {
            top = free_list.back().off; free_list.pop_back();
        }
    }
};

// Descriptor table of a batched kernel: device copy of the descriptors + (descriptor, local block) per launched block.
struct DevTable {
    void* descs = nullptr; int2* map = nullptr; int nblocks = 0;
    std::vector<char> h_descs, h_map;                // staged at plan-build time (no device needed), uploaded on first use (ready)
    ~DevTable() { if (descs) (void)hipFree(descs); if (map) (void)hipFree(map); }
    template <class D> int upload(const std::vector<D>& d, const std::vector<int2>& m) {
        if (d.empty()) return 0;
        h_descs.assign((const char*)d.data(), (const char*)(d.data() + d.size()));
        h_map.assign((const char*)m.data(), (const char*)(m.data() + m.size()));
        nblocks = (int)m.size();
        return 0;
    }
    int ready() {
        if (descs || h_descs.empty()) return 0;
        if (hipMalloc(&descs, h_descs.size()) != hipSuccess || hipMalloc((void**)&map, h_map.size()) != hipSuccess) return -1;
        if (hipMemcpy(descs, h_descs.data(), h_descs.size(), hipMemcpyHostToDevice) != hipSuccess) return -1;
        if (hipMemcpy(map, h_map.data(), h_map.size(), hipMemcpyHostToDevice) != hipSuccess) return -1;
        return 0;
    }
};

struct TapInfo { std::string name; int dims[5]; size_t off; };   // dims = B, C, D, H, W; off = element offset in the tap buffers

struct Plan {
    std::vector<Op> ops;
    std::vector<TapInfo> taps; size_t tap_elems = 0;
    size_t ws_bytes = 0;
    size_t bwd_begin = 0;                            // training plans: ops [0, bwd_begin) = forward, the rest = backward
    bool train = false;
    mutable DevTable wt_tab, exp_tab, cs_tab;                // training plans: all weight transposes / gradient exports / column-sum finalizes in one launch each
};

// ================================================================================================ parameters
enum PackKind { PK_CONV_W, PK_VEC_F32, PK_LINEAR_W };
struct ParamDesc {
    std::string name; std::vector<int64_t> shape;
    PackKind kind; size_t dst_off;                   // arena byte offset of the destination matrix / vector
    int k = 1, cout = 0, cin = 0, cout_pad = 0, cin_s = 0, row_off = 0;
    int64_t flat_off = 0;                            // element offset in the flat fp32 gradient buffer (parameter order)
    bool loaded = false;
};

struct ConvW { size_t w_off = 0; size_t b_off = 0; int cout = 0, cout_pad = 0, cin_s = 0, k = 1; bool has = false;
               size_t wp_off = 0;                    // upsample convs: the derived phase weights [8][8][cout_pad][cin_s] (0 = none)
               size_t x3_off = (size_t)-1;           // fp32 precision: [27][cout_pad][hi | lo | hi] bf16 behind the fp32 arena ((size_t)-1 = none)
               size_t x3p_off = (size_t)-1; };       // ... and of the upsample convs: phase weights [8][8][cout_pad][hi | hi | lo]
struct GnW { size_t g_off = 0, b_off = 0; int C = 0; };
struct LinW { size_t w_off = 0, b_off = 0; int in = 0, out = 0; };

// data-parallel gradient exchange overlapped with backward (ldm_model_set_grad_sync): every OP_BUCKET records an event on the launch
// stream, makes the comm stream wait for it and queues the bucket's all-reduce (mean) there; OP_BUCKET_JOIN makes the launch stream
// wait for the last one.  Timing events (issue point on the launch stream, completion on the comm stream) feed ldm_model_grad_sync_trace.
struct ldm_comm;
struct GradSyncState {
    ldm_comm* comm = nullptr; hipStream_t stream = nullptr;
    std::vector<hipEvent_t> ev;                      // per bucket: issue (launch stream), done (comm stream); [0] = backward start
    size_t used = 0; std::vector<int64_t> elems;
    int wire = 0;                                    // 0 = fp32 on the wire (the reference's DDP), 1 = bf16 (ldm_model_set_grad_wire)
    void* stage = nullptr; size_t stage_bytes = 0;   // bf16 staging slice of the largest bucket seen (comm stream is in order: one suffices)
    int pending = 0;                                 // buckets handed to the comm stream since the last join
};
static int grad_sync_begin(GradSyncState& g, hipStream_t s);
static int grad_sync_bucket(GradSyncState& g, float* buf, int64_t count, hipStream_t s);
static int grad_sync_join(GradSyncState& g, hipStream_t s);
struct ldm_model {
    int type = 0;                                    // 0 = UNet, 1 = VAE
    ldm_unet_cfg ucfg{}; ldm_vae_cfg vcfg{};
    std::vector<ParamDesc> params;
    std::map<std::string, int> pindex;
    size_t arena_bytes = 8192 + 256;                 // first 8 KiB: the zero page (one padded input row, C <= 4096)
    char* arena = nullptr;
    // fp32 precision mode (ldm_model_set_precision): unrounded copies of every matrix, allocated on first use
    int precision = 0; char* arena32 = nullptr;
    int loaded_count = 0;
    std::map<std::string, ConvW> convs; std::map<std::string, GnW> gns; std::map<std::string, LinW> lins;
    std::map<std::string, std::shared_ptr<Plan>> plans;
    DevTable pack_tab;                               // one-launch re-pack of a flat fp32 parameter buffer
    // HIP-graph replay of the forward plan (ldm_model_set_graph_mode): one hipGraphLaunch instead of ~150 kernel launches
    // per step on the host.  A graph is instantiated per (plan, pointer set) the second time that set is seen.
    int graph_mode = 0;
    // sampler_uid: the captured sampler kernel bakes the sampler's seed / step table / state pointers in by value, and a freed
    // ldm_sampler's address is readily handed out again: the key carries the sampler's never-reused id, not only its address
    struct GraphEntry { const Plan* plan; const void* ptr[8]; int rt[2]; uint64_t sampler_uid; int seen; hipGraphExec_t exec; };
    std::vector<GraphEntry> graphs;
    hipStream_t cap_stream = nullptr;        // capture happens on a private stream (the caller's may be the null stream, which cannot capture)
    GradSyncState gsync;                     // ldm_model_set_grad_sync
    // UNet: stacked time_emb_proj GEMV
    size_t tproj_w_off = 0, tproj_b_off = 0; int tproj_rows = 0; std::map<std::string, int> tproj_row;
    // UNet: the stacked projections tabulated for every step of a sampler's schedule ([n_steps][tproj_rows] fp32; ldm_unet_denoise_step):
    // valid for the schedule `temb_tab_ts` and the precision it was built in, until the next parameter upload
    float* temb_tab = nullptr; size_t temb_tab_cap = 0; std::vector<float> temb_tab_ts; int temb_tab_prec = -1; bool temb_tab_valid = false;

    size_t arena_alloc(size_t bytes) { size_t o = arena_bytes; arena_bytes += rup_sz(bytes, 256); return o; }
    // weights derived from the packed ones (phase weights of the upsample convs): rebuilt on the stream of the next inference
    // call after any parameter upload
    struct PhaseW { size_t w_off, wp_off; int cout_pad, cin_s; };
    std::vector<PhaseW> phase_ws; bool derived_dirty = true;
    // fp32 precision: bf16 (hi | lo | hi) weights of the 3^3 convs for the 3 x bf16 halo conv, derived from the fp32 arena; they live
    // behind it (arena32 = [2 * arena_bytes fp32 twins][x3_bytes])
    struct X3W { size_t w_off, x3_off; long rows; int cin_s; };
    std::vector<X3W> x3_ws; size_t x3_bytes = 0;
    struct X3PW { size_t w_off, x3p_off; int cout_pad, cin_s; };
    std::vector<X3PW> x3p_ws;
    struct Im2colW { size_t w_off, wi_off; int cout_pad, cin_s, cin, Kp; };     // first convs run as im2col + GEMM (inference plans)
    std::vector<Im2colW> im2col_ws;
    static int im2col_k(int cin) { const int k = 27 * cin; return k <= 96 ? rup(k, 32) : rup(k, 128); }
    // registers <name>.im2col: a 1x1 "conv" over the patch matrix whose weights are derived from <name>'s packed 3^3 weights
    void reg_im2col(const std::string& name, int cin) {
        if (cin > 9) return;
        const ConvW& c3 = convs.at(name);
        ConvW c = c3; c.k = 1; c.cin_s = im2col_k(cin); c.wp_off = 0;
        c.w_off = arena_alloc((size_t)c.cout_pad * c.cin_s * 2);
        im2col_ws.push_back(Im2colW{c3.w_off, c.w_off, c3.cout_pad, c3.cin_s, cin, c.cin_s});
        convs[name + ".im2col"] = c;
    }
    int64_t flat_total = 0;
    void add_param(const ParamDesc& d) {
        pindex[d.name] = (int)params.size(); params.push_back(d);
        ParamDesc& q = params.back(); q.flat_off = flat_total;
        int64_t n = 1; for (auto v : q.shape) n *= v;
        flat_total += n;
    }

    // conv registered under MONAI's Convolution wrapper: <name>.conv.{weight,bias}; several logical convs may
    // share one destination matrix (fused q|k|v, mu|log_sigma) through row_off.
    ConvW reg_conv_into(const std::string& pname, ConvW dst, int cin, int cout, int k, int row_off, bool lin_names) {
        ParamDesc w; w.name = pname + (lin_names ? ".weight" : ".conv.weight");
        if (k == 1 && lin_names) w.shape = {cout, cin}; else w.shape = {cout, cin, k, k, k};
        w.kind = PK_CONV_W; w.dst_off = dst.w_off; w.k = k; w.cout = cout; w.cin = cin;
        w.cout_pad = dst.cout_pad; w.cin_s = dst.cin_s; w.row_off = row_off;
        add_param(w);
        ParamDesc b; b.name = pname + (lin_names ? ".bias" : ".conv.bias"); b.shape = {cout};
        b.kind = PK_VEC_F32; b.dst_off = dst.b_off + (size_t)row_off * 4; b.cout = cout;
        add_param(b);
        return dst;
    }
    ConvW new_conv_slot(int cin_s, int cout_total, int k) {
        ConvW c; c.has = true; c.k = k; c.cout = cout_total; c.cout_pad = rup(cout_total, 64); c.cin_s = cin_s;
        c.w_off = arena_alloc((size_t)k * k * k * c.cout_pad * cin_s * 2);
        c.b_off = arena_alloc((size_t)c.cout_pad * 4);
        return c;
    }
    void reg_x3(ConvW& c) {                          // a 3^3 conv the fp32 inference plans may run on the halo kernel (x3_ws)
        if (c.k != 3 || c.cin_s % 64) return;
        c.x3_off = x3_bytes; x3_bytes += rup_sz((size_t)27 * c.cout_pad * 3 * c.cin_s * 2, 256);
        x3_ws.push_back(X3W{c.w_off, c.x3_off, 27L * c.cout_pad, c.cin_s});
    }
    void reg_conv(const std::string& name, int cin, int cin_s, int cout, int k, bool phase = false) {
        ConvW c = new_conv_slot(cin_s, cout, k);
        if (phase && k == 3) {
            c.wp_off = arena_alloc((size_t)64 * c.cout_pad * cin_s * 2);
            phase_ws.push_back(PhaseW{c.w_off, c.wp_off, c.cout_pad, cin_s});
            if (cin_s % 64 == 0) {
                c.x3p_off = x3_bytes; x3_bytes += rup_sz((size_t)64 * c.cout_pad * 3 * cin_s * 2, 256);
                x3p_ws.push_back(X3PW{c.w_off, c.x3p_off, c.cout_pad, cin_s});
            }
        }
        reg_x3(c);
        reg_conv_into(name, c, cin, cout, k, 0, false);
        convs[name] = c;
    }
    void reg_gn(const std::string& name, int C) {
        GnW g; g.C = C; g.g_off = arena_alloc((size_t)C * 4); g.b_off = arena_alloc((size_t)C * 4);
        ParamDesc w; w.name = name + ".weight"; w.shape = {C}; w.kind = PK_VEC_F32; w.dst_off = g.g_off; w.cout = C; add_param(w);
        ParamDesc b; b.name = name + ".bias"; b.shape = {C}; b.kind = PK_VEC_F32; b.dst_off = g.b_off; b.cout = C; add_param(b);
        gns[name] = g;
    }
    void reg_linear_at(const std::string& name, int in, int out, size_t w_off, size_t b_off) {
        ParamDesc w; w.name = name + ".weight"; w.shape = {out, in}; w.kind = PK_LINEAR_W; w.dst_off = w_off; w.cout = out; w.cin = in; add_param(w);
        ParamDesc b; b.name = name + ".bias"; b.shape = {out}; b.kind = PK_VEC_F32; b.dst_off = b_off; b.cout = out; add_param(b);
    }
    void reg_linear(const std::string& name, int in, int out) {
        LinW l; l.in = in; l.out = out; l.w_off = arena_alloc((size_t)in * out * 2); l.b_off = arena_alloc((size_t)out * 4);
        reg_linear_at(name, in, out, l.w_off, l.b_off);
        lins[name] = l;
    }
    void reg_attn(const std::string& p, int C) {       // SpatialAttentionBlock: norm + fused q|k|v + out_proj
        reg_gn(p + ".norm", C);
        ConvW qkv = new_conv_slot(C, 3 * C, 1);
        reg_conv_into(p + ".attn.to_q", qkv, C, C, 1, 0, true);
        reg_conv_into(p + ".attn.to_k", qkv, C, C, 1, C, true);
        reg_conv_into(p + ".attn.to_v", qkv, C, C, 1, 2 * C, true);
        convs[p + ".attn.qkv"] = qkv;
        ConvW o = new_conv_slot(C, C, 1);
        reg_conv_into(p + ".attn.out_proj", o, C, C, 1, 0, true);
        convs[p + ".attn.out_proj"] = o;
    }
};

// ================================================================================================ builder
static int wgrad_ksplit(long M, int taps, int cout, int cin, bool hp = false, int stride = 1, int ups = 0);

struct Builder {
    ldm_model* m; Plan* plan; Pool pool;
    size_t partial_off = 0, partial_bytes = 0;       // shared split-K slab scratch (sized at the end)
    size_t gnpart_off = 0, gnpart_bytes = 0;         // GroupNorm partial sums
    size_t gnab_off = 0, gnab_bytes = 0;             // GroupNorm per-channel scale/shift
    std::vector<size_t> partial_fixups, gnpart_fixups, gnab_fixups;   // op indices whose refs need final offsets
    std::string err;

    Act new_act(int N, int D, int H, int W, int C) {
        Act a; a.N = N; a.D = D; a.H = H; a.W = W; a.C = C; a.valid = true; a.esz = hp ? 4 : 2; a.off = pool.alloc(a.bytes()); return a;
    }
    bool hp = false;                                  // fp32 precision plan (f32_path.h kernels, fp32 activations)
    // debug taps (ldm_*_taps entry points): 0 = none, 1 = export every block output, 2 = export, then overwrite it with the
    // caller's tensor (teacher forcing: the next block sees the reference's input)
    int tap_mode = 0;
    void tap(const std::string& name, Act& h, int c_real) {
        if (!tap_mode) return;
        TapInfo t; t.name = name; t.dims[0] = h.N; t.dims[1] = c_real; t.dims[2] = h.D; t.dims[3] = h.H; t.dims[4] = h.W; t.off = plan->tap_elems;
        plan->tap_elems += (size_t)h.N * c_real * h.D * h.H * h.W;
        Op o{}; o.kind = OP_TAP; o.r[0] = ws_ref(h.off);
        o.r[1].base = BASE_TAPO; o.r[1].off = t.off * 4; o.r[2].base = BASE_TAPI; o.r[2].off = t.off * 4;
        o.i[0] = h.N; o.i[1] = c_real; o.i[2] = h.C; o.i[3] = h.D * h.H * h.W; o.i[4] = h.esz; o.i[5] = tap_mode;
        plan->ops.push_back(o);
        plan->taps.push_back(t);
        if (tap_mode == 2 && h.has_stats) { pool.release(h.stats_off); h.has_stats = false; }   // the producer's GroupNorm partials describe the overwritten tensor
    }
    void free_act(Act& a) {
        if (train) return;                            // training plans keep every activation for the backward pass
        if (a.valid) { pool.release(a.off); if (a.has_stats) pool.release(a.stats_off); }
        a.valid = false; a.has_stats = false;
    }
    bool train = false, recording = false;

    // ---- conv ---------------------------------------------------------------------------------------
    struct ConvArgs {
        Act xa, xb;                                   // group-0 sources (xb optional)
        const ConvW* w = nullptr;
        int k = 3, stride = 1, pad = 1, ups = 0;
        int Do = 0, Ho = 0, Wo = 0;
        Act g1a, g1b; const ConvW* w1 = nullptr;     // fused 1x1 skip (optional)
        Ref temb; int temb_stride = 0;               // optional per-sample channel bias
        Act residual;                                // optional
        bool f32_out = false; Ref out_ref; int cout_real = 0;
        bool want_stats = true;                       // also emit the output's GroupNorm partials
        // backward-pass uses of the same kernel (data gradients)
        Ref w_over; bool no_bias = false; int exact = 0;   // weights from the workspace; zero-insertion upsample
        int temb_row = -1;                            // first row of this ResBlock in the stacked time projection
        int f32_tag = 0;                              // which externally supplied gradient an fp32-output conv receives (0 final, 1 VAE heads)
    };
    struct Tape {                                    // one differentiable forward op, recorded in training plans
        int kind = 0;                                 // 0 conv, 1 GroupNorm(+SiLU), 2 attention
        ConvArgs c; Act out; bool leaf_input = false;
        const GnW* g = nullptr; Act xa, xb; size_t ab_off = 0, mr_off = 0; int groups = 0; bool silu = false;
        Act qkv, o; size_t lse_off = 0; int head_ch = 64;
    };
    std::vector<Tape> tape;

    static bool halo_enabled() { return ldm_knob("LDM_CONV_HALO", 1) != 0; }
    // LDM_HALO_TALL: 0 = never, 1 = wherever the cost model prefers it, 2 (default) = only for Cout % 128 != 0
    static int tall_mode() { return ldm_xknob("LDM_HALO_TALL", 2); }
    static bool light_enabled() { return ldm_knob("LDM_GEMM_LIGHT", 1) != 0; }
    static bool two_wg_enabled() { return ldm_knob("LDM_IGEMM_2WG", 1) != 0; }
    static bool phase_enabled() { return ldm_knob("LDM_CONV_PHASE", 1) != 0; }
    // halo_n > 0: the conv is eligible for conv3_halo_kernel (3^3, stride 1, pad 1, single source, BK 64); halo_n = N
    // and halo_dhw = voxels per sample (its 126-row tiles never straddle samples).
    // steps = K steps of the 3^3 (or k^3) part; steps1 = extra K steps of a fused 1x1 skip (the halo kernel runs them as a second loop
    // behind the 3^3 one, unsplit convs only)
    static bool halo_skip_enabled() { static const int v = ldm_knob("LDM_HALO_SKIP", 1); return v != 0; }
    static bool halo_skip_split_enabled() { static const int v = ldm_knob("LDM_HALO_SKIP_SPLIT", 1); return v != 0; }   // ... in split-K convs too
    static ConvCfg choose_cfg(long M, int cout_pad, int steps0, int bk, int halo_n = 0, long halo_dhw = 0, bool halo_only = false, int steps1 = 0) {
        ConvCfg best{2, 2, bk, 1}; double best_t = 1e30;
        const int steps = steps0 + steps1;
        const int rb = bk * 2;
        static const int splits[] = {1, 2, 3, 4, 6, 8, 9, 12, 16, 18, 24, 27, 32, 48, 64};
        for (int wgn = 1; wgn <= 4 && !halo_only; wgn *= 2) {
            const int bn = 64 * wgn, bm = 64 * (4 / wgn);
            if (cout_pad % bn) continue;
            const long tiles = ((M + bm - 1) / bm) * (cout_pad / bn);
            const double c_mfma = (double)bm * bn * bk * 2.0 / 4096.0;           // cycles at the per-CU MFMA peak
            const double c_load = (double)(bm + bn) * rb / 28.0;                 // ~28 B/clk/CU global->LDS
            const double c_step = std::max(c_mfma, c_load) + 60.0;
            for (int sk : splits) {
                if (sk > 1 && steps / sk < 6) break;
                const long nwg = tiles * sk;
                const int sps = (steps + sk - 1) / sk;
                const double rounds = ceil((double)nwg / 256.0);
                double t = rounds * (sps * c_step + 2500.0);
                if (sk > 1) t += 6000.0 + (double)M * cout_pad * 4.0 * sk * 2.0 / 2500.0;   // slab write+read, ~2.5 KB/clk chip
                if (t < best_t) { best_t = t; best = ConvCfg{4 / wgn, wgn, bk, sk}; }
            }
        }
        if (halo_n > 0 && bk == 64 && cout_pad % 128 == 0 && halo_enabled()) {
            const int mtps = (int)((halo_dhw + 125) / 126), Q = steps0 / 3;      // steps0 = 27 * nchunk
            const long tiles = (long)halo_n * mtps * (cout_pad / 128);
            const double c_mfma = 128.0 * 128.0 * 64.0 * 2.0 / 4096.0;
            const double c_load = (128.0 / 3.0 + 128.0) * 128.0 / 28.0;          // the voxel tile is copied once per 3 steps
            const double c_step = std::max(c_mfma, c_load) + 30.0;
            for (int sk : splits) {
                if (sk > 1 && Q / sk < 3) break;
                const int qps = (Q + sk - 1) / sk, skr = (Q + qps - 1) / qps;    // every split non-empty
                const long nwg = tiles * skr;
                const double rounds = ceil((double)nwg / 256.0);
                double t = rounds * (3.0 * qps * c_step + 2500.0 + ((steps1 + skr - 1) / skr) * 1.6 * c_step + (steps1 && skr > 1 ? 800.0 : 0.0));
                if (skr > 1) t += 6000.0 + (double)M * cout_pad * 4.0 * skr * 2.0 / 2500.0;
                if (steps1 && skr > 1 && !halo_skip_split_enabled()) continue;   // the splits share the fused skip's K steps (round 5)
                if (t < best_t) { best_t = t; best = ConvCfg{2, 2, 64, skr}; best.halo = 1; best.mtps = mtps; best.qps = qps; }
            }
        }
        // the tall halo tile (254 voxels x 64 couts, halo = 2): always for Cout % 128 != 0, else where the model says so
        if (halo_n > 0 && bk == 64 && cout_pad % 64 == 0 && halo_enabled() && tall_mode() != 0) {
            const int mtps = (int)((halo_dhw + 253) / 254), Q = steps0 / 3;
            const long tiles = (long)halo_n * mtps * (cout_pad / 64);
            const double c_mfma = 256.0 * 64.0 * 64.0 * 2.0 / 4096.0;
            const double c_load = (256.0 / 3.0 + 64.0) * 128.0 / 28.0;
            const double c_step = std::max(c_mfma, c_load) + 30.0;
            for (int sk : splits) {
                if (sk > 1 && Q / sk < 3) break;
                const int qps = (Q + sk - 1) / sk, skr = (Q + qps - 1) / qps;
                const long nwg = tiles * skr;
                const double rounds = ceil((double)nwg / 256.0);
                double t = rounds * (3.0 * qps * c_step + 2500.0 + ((steps1 + skr - 1) / skr) * 1.6 * c_step + (steps1 && skr > 1 ? 800.0 : 0.0));
                if (skr > 1) t += 6000.0 + (double)M * cout_pad * 4.0 * skr * 2.0 / 2500.0;
                if (steps1 && skr > 1 && !halo_skip_split_enabled()) continue;
                if (tall_mode() == 2 && cout_pad % 128 == 0) t = 1e31;           // mode 2: only where the 128-cout tile cannot run
                if (tall_mode() == 3 && skr == 1 && best.halo == 1 && best.splitk == 1) t = 0.0;   // mode 3 (experiments): wherever the 128-cout tile runs unsplit
                if (t < best_t) { best_t = t; best = ConvCfg{4, 1, 64, skr}; best.halo = 2; best.mtps = mtps; best.qps = qps; }
            }
        }
        return best;
    }

    // fp32 precision: every conv form the inference plans use on conv_f32_kernel (the 1x1 skip runs as its own conv -> residual)
    // fp32 inference plans: the split-K finalize also leaves the output's GroupNorm partials (finalize_stats_f32_kernel), so the
    // GroupNorm that reads it folds them in its own launch (gn_apply's hp && fused branch) instead of a statistics pass.  LDM_FIN32_STATS=0: off.
    void fin32_stats(Op& f, Act& out, const ConvArgs& a, int N, int dhwo, int couts) {
        static const int on = ldm_xknob("LDM_FIN32_STATS", 1);
        if (!on || train || !a.want_stats || a.f32_out || couts % 4 || couts > 1024) return;
        const int cvec = couts / 4, rows_par = std::max(1, 256 / cvec);
        int nrb = std::min((dhwo + rows_par - 1) / rows_par, std::max(1, 256 / N));
        const int rows = (dhwo + nrb - 1) / nrb;
        nrb = (dhwo + rows - 1) / rows;
        out.stats_off = pool.alloc((size_t)N * nrb * couts * 2 * 4); out.has_stats = true; out.stats_nrb = nrb;
        f.r[12] = ws_ref(out.stats_off); f.i[2] = nrb; f.i[3] = rows;
    }
    Act conv32(const ConvArgs& a, const std::string& tag) {
        const ConvW& w = *a.w;
        if (a.w1) { err = "fp32 precision: unsupported conv form (" + tag + ")"; return Act(); }
        const int cin0 = a.xa.C + (a.xb.valid ? a.xb.C : 0);
        if (cin0 != w.cin_s || cin0 % 16 || a.xa.C % 16) { err = "conv " + tag + ": channel bookkeeping mismatch"; return Act(); }
        const int N = a.xa.N;
        const long M = (long)N * a.Do * a.Ho * a.Wo;
        if (M >= (1L << 31)) { err = "conv " + tag + ": M too large"; return Act(); }
        // Upsample block in an inference plan, phase form (8 taps on the source grid instead of 27 on the upsampled one) as the
        // 3 x bf16 product on conv_igemm_kernel: the source is split once (hi | lo, same size), the kernel reads it as two concatenated
        // sources (hi | lo, then hi again) against phase weights [hi | hi | lo], fp32 slabs, fp32 finalize.
        {
            static const int x3_phase = ldm_xknob("LDM_X3_PHASE", 1);
            const int C = a.xa.C;
            if (x3_phase && !train && a.ups == 1 && !a.exact && a.k == 3 && a.stride == 1 && a.pad == 1 && !a.xb.valid && !a.w1 && !a.f32_out &&
                a.w_over.base == BASE_NULL && !a.xa.hl && a.Do == 2 * a.xa.D && a.Ho == 2 * a.xa.H && a.Wo == 2 * a.xa.W && phase_enabled() &&
                w.x3p_off != (size_t)-1 && x3_halo_ok(w, M, C) && a.temb.base == BASE_NULL && !a.residual.valid) {
                const long rows_src = (long)N * a.xa.D * a.xa.H * a.xa.W;
                Act sp = new_act(N, a.xa.D, a.xa.H, a.xa.W, C); sp.hl = true;
                Op u{}; u.kind = OP_UPS_SPLIT32; u.r[0] = ws_ref(a.xa.off); u.r[1] = ws_ref(sp.off);
                u.i[0] = N; u.i[1] = C; u.i[2] = a.xa.D; u.i[3] = a.xa.H; u.i[4] = a.xa.W; u.i[5] = 0;
                plan->ops.push_back(u);
                int bk = 64, nchunk0 = 3 * C / 64, steps0 = 8 * nchunk0;
                ConvCfg cc = choose_cfg(M, w.cout_pad, steps0, 64);
                if (cc.wgm > 2 || cc.splitk != 1) cc = ConvCfg{2, 2, 64, 1};
                if (w.cout_pad % (64 * cc.wgn)) cc = ConvCfg{4, 1, 64, 1};
                const int bm = 64 * cc.wgm, bn = 64 * cc.wgn;
                const int mtiles_pp = (int)(((long)a.xa.D * a.xa.H * a.xa.W + bm - 1) / bm);
                if (cc.wgm <= 2 && two_wg_enabled() && steps0 <= 64 && (long)N * 8 * mtiles_pp * (w.cout_pad / bn) >= 512) { cc.bk = bk = 32; nchunk0 = 3 * C / 32; steps0 = 8 * nchunk0; }
                (void)rows_src;
                const int couts = rup(w.cout, 32);
                Act out = new_act(N, a.Do, a.Ho, a.Wo, couts);
                Op op{}; op.kind = OP_CONV; op.cc = cc;
                op.r[0] = ws_ref(sp.off); op.r[1] = ws_ref(sp.off); op.r[2] = Ref{BASE_W32, 2 * m->arena_bytes + w.x3p_off};
                int* i = op.i;
                i[0] = 2 * C; i[1] = C; i[4] = N; i[5] = a.xa.D; i[6] = a.xa.H; i[7] = a.xa.W; i[8] = a.Do; i[9] = a.Ho; i[10] = a.Wo;
                i[11] = 2; i[12] = 1; i[13] = 1; i[14] = 4 | 8; i[15] = (int)M; i[16] = couts; i[17] = w.cout_pad; i[18] = w.cout;
                i[19] = nchunk0; i[23] = N * 8 * mtiles_pp;
                partial_bytes = std::max(partial_bytes, (size_t)M * w.cout_pad * 4);
                partial_fixups.push_back(plan->ops.size()); plan->ops.push_back(op);
                Op f{}; f.kind = OP_FIN32; f.cc = ConvCfg{2, 2, 32, 1};
                f.r[2] = w32_ref(w.w_off); f.r[6] = a.no_bias ? Ref() : w_ref(w.b_off); f.r[10] = ws_ref(out.off);
                int* j = f.i;
                j[0] = C; j[4] = N; j[5] = a.xa.D; j[6] = a.xa.H; j[7] = a.xa.W; j[8] = a.Do; j[9] = a.Ho; j[10] = a.Wo;
                j[11] = 3; j[12] = 1; j[13] = 1; j[15] = (int)M; j[16] = couts; j[17] = w.cout_pad; j[18] = w.cout;
                j[19] = C / 32; j[20] = 1; j[23] = (int)((M + 127) / 128);
                fin32_stats(f, out, a, N, a.Do * a.Ho * a.Wo, couts);
                partial_fixups.push_back(plan->ops.size()); plan->ops.push_back(f);
                free_act(sp);
                return out;
            }
        }
        // Upsample block (nearest x2, then 3^3 conv) in an inference plan: one pass writes the upsampled tensor as the (hi | lo) split, the
        // conv then runs on the halo kernel like the ResBlock convs (12^3 -> 24^3, 256 channels: 258 -> ~175 us)
        if (!train && a.ups == 1 && !a.exact && a.k == 3 && a.stride == 1 && a.pad == 1 && !a.xb.valid && !a.w1 && !a.f32_out &&
            a.w_over.base == BASE_NULL && !a.xa.hl && a.Do == 2 * a.xa.D && a.Ho == 2 * a.xa.H && a.Wo == 2 * a.xa.W && x3_halo_ok(w, M, a.xa.C)) {
            Act up = new_act(N, a.Do, a.Ho, a.Wo, a.xa.C); up.hl = true;
            Op u{}; u.kind = OP_UPS_SPLIT32; u.r[0] = ws_ref(a.xa.off); u.r[1] = ws_ref(up.off);
            u.i[0] = N; u.i[1] = a.xa.C; u.i[2] = a.xa.D; u.i[3] = a.xa.H; u.i[4] = a.xa.W; u.i[5] = 1;
            plan->ops.push_back(u);
            ConvArgs a2 = a; a2.xa = up; a2.ups = 0;
            Act out = conv32(a2, tag);
            free_act(up);
            return out;
        }
        if (a.xa.hl) {                                   // 3 x bf16 product on conv3_halo_kernel (x3_halo_ok decided it when the GroupNorm was planned)
            const int C = a.xa.C, n3 = C / 64;
            if (a.k != 3 || a.stride != 1 || a.pad != 1 || a.ups || a.exact || a.xb.valid || a.w_over.base != BASE_NULL ||
                w.x3_off == (size_t)-1 || C != w.cin_s || a.xa.D != a.Do || a.xa.H != a.Ho || a.xa.W != a.Wo || (long)M * C * 4 >= (1L << 32)) {
                err = "conv " + tag + ": hi/lo input reached a conv that cannot take it"; return Act();
            }
            ConvCfg cc = choose_cfg(M, w.cout_pad, 27 * 3 * n3, 64, N, (long)a.Do * a.Ho * a.Wo, true);
            if (!cc.halo) { err = "conv " + tag + ": no halo configuration"; return Act(); }
            const int couts = rup(w.cout, 32);
            Act out;                                         // f32_out (the network's last conv): fp32 NCDHW straight to the caller's buffer
            if (!a.f32_out) out = new_act(N, a.Do, a.Ho, a.Wo, couts);
            Op op{}; op.kind = OP_CONV; op.cc = cc;
            op.r[0] = ws_ref(a.xa.off); op.r[2] = Ref{BASE_W32, 2 * m->arena_bytes + w.x3_off};
            int* i = op.i;
            i[0] = 2 * C; i[4] = N; i[5] = a.xa.D; i[6] = a.xa.H; i[7] = a.xa.W; i[8] = a.Do; i[9] = a.Ho; i[10] = a.Wo;
            i[11] = 3; i[12] = 1; i[13] = 1; i[14] = 8; i[15] = (int)M; i[16] = couts; i[17] = w.cout_pad; i[18] = w.cout;
            i[19] = 3 * n3; i[23] = N * cc.mtps;
            static const int x3_fused = ldm_xknob("LDM_X3_FUSED_EP", 1);
            {   // the last layer with <= 4 output channels: conv3_thin_kernel over the split (conv_thin.h, ThinParams::x3_c)
                static const int thin = ldm_knob("LDM_CONV_THIN", 1);
                const int cr = a.cout_real ? a.cout_real : w.cout;
                if (thin && a.f32_out && cr <= 4 && C <= 128 && a.temb.base == BASE_NULL && !a.residual.valid &&
                    (long)N * ((a.Do + THIN_TD - 1) / THIN_TD) * ((a.Ho + THIN_TH - 1) / THIN_TH) * ((a.Wo + THIN_TW - 1) / THIN_TW) >= 512) {
                    Op t{}; t.kind = OP_CONV_THIN;
                    t.r[0] = ws_ref(a.xa.off); t.r[2] = Ref{BASE_W32, 2 * m->arena_bytes + w.x3_off}; t.r[6] = a.no_bias ? Ref() : w_ref(w.b_off); t.r[10] = a.out_ref;
                    t.i[0] = N; t.i[1] = a.xa.D; t.i[2] = a.xa.H; t.i[3] = a.xa.W; t.i[4] = 3 * C; t.i[5] = w.cout_pad; t.i[6] = cr; t.i[7] = C;
                    plan->ops.push_back(t);
                    return Act();
                }
            }
            if (cc.splitk == 1 && x3_fused && a.f32_out) {   // no split, last conv: the kernel's own fp32 NCDHW epilogue (bias only)
                i[14] |= 32; i[22] = 1; i[18] = a.cout_real ? a.cout_real : w.cout;
                op.r[6] = a.no_bias ? Ref() : w_ref(w.b_off); op.r[10] = a.out_ref;
                plan->ops.push_back(op);
                return out;
            }
            if (cc.splitk == 1 && x3_fused) {                // no split: bias / time embedding / residual, the fp32 store and the GroupNorm partials in the conv's epilogue
                i[14] |= 16; i[21] = a.temb_stride;
                op.r[6] = a.no_bias ? Ref() : w_ref(w.b_off); op.r[8] = a.temb;
                op.r[9] = a.residual.valid ? ws_ref(a.residual.off) : Ref(); op.r[10] = ws_ref(out.off);
                if (a.want_stats) {
                    out.stats_off = pool.alloc((size_t)N * cc.mtps * couts * 2 * 4); out.has_stats = true; out.stats_nrb = cc.mtps;
                    op.r[12] = ws_ref(out.stats_off);
                }
                plan->ops.push_back(op);
                return out;
            }
            partial_bytes = std::max(partial_bytes, (size_t)cc.splitk * M * w.cout_pad * 4);
            partial_fixups.push_back(plan->ops.size()); plan->ops.push_back(op);
            Op f{}; f.kind = OP_FIN32; f.cc = ConvCfg{2, 2, 32, cc.splitk};
            f.r[2] = w32_ref(w.w_off); f.r[6] = a.no_bias ? Ref() : w_ref(w.b_off);
            f.r[8] = a.temb; f.r[9] = a.residual.valid ? ws_ref(a.residual.off) : Ref(); f.r[10] = a.f32_out ? a.out_ref : ws_ref(out.off);
            int* j = f.i;
            j[0] = C; j[4] = N; j[5] = a.xa.D; j[6] = a.xa.H; j[7] = a.xa.W; j[8] = a.Do; j[9] = a.Ho; j[10] = a.Wo;
            j[11] = 3; j[12] = 1; j[13] = 1; j[15] = (int)M; j[16] = couts; j[17] = w.cout_pad; j[18] = w.cout;
            j[19] = C / 32; j[20] = w.cout_pad / 128 ? w.cout_pad / 128 : 1; j[21] = a.temb_stride; j[23] = (int)((M + 127) / 128);
            if (a.f32_out) { j[22] = 1; j[18] = a.cout_real ? a.cout_real : w.cout; }
            fin32_stats(f, out, a, N, a.Do * a.Ho * a.Wo, couts);
            partial_fixups.push_back(plan->ops.size()); plan->ops.push_back(f);
            return out;
        }
        // 3 x bf16 form of the product (conv_x3_kernel, K steps of 32 channels) in the INFERENCE plans wherever the channel counts allow:
        // ~5e-6 per block, 5e-5 end to end (gate 1e-3).  Training plans keep the exact fp32 MFMA: at unit weight gain the backward of
        // these networks amplifies a 1e-5 perturbation to 1e-3 of the gradient (measured), which is the whole parity budget.
        // LDM_F32_X3=0: fp32 MFMA everywhere; 2: 3 x bf16 in training plans too.
        static const int f32_x3 = ldm_knob("LDM_F32_X3", 1);
        const bool x3 = (f32_x3 == 2 || (f32_x3 == 1 && !train)) && cin0 % 32 == 0 && a.xa.C % 32 == 0;
        {   // 1x1x1 convolutions of the 3 x bf16 plans: light GEMM on the fp32 operands, no slabs and no finalize (gemm_light_x3.h).  LDM_LIGHT_X3=0: off.
            static const int light_x3 = ldm_xknob("LDM_LIGHT_X3", 1);
            static const long light_x3_max_m = ldm_xknob("LDM_LIGHT_X3_MAX_M", 4096L);   // above: conv_x3_kernel's LDS tiles (one split per tile) win: 24^3 x 512 -> 256: 61 vs 68 us
            if (light_x3 && x3 && M <= light_x3_max_m && a.k == 1 && a.stride == 1 && a.pad == 0 && !a.ups && !a.exact && !a.f32_out && !a.xa.hl && a.temb.base == BASE_NULL &&
                (!a.xb.valid || a.xb.C % 32 == 0) && w.cout_pad % 32 == 0 && a.xa.D == a.Do && a.xa.H == a.Ho && a.xa.W == a.Wo &&
                M * (long)std::max(cin0, w.cout_pad) * 4 < (1L << 31)) {
                const int big = gemm_light_x3_big(M, w.cout_pad), rows = big ? 64 : 32;
                const int couts_l = rup(w.cout, 32);
                Act out = new_act(N, a.Do, a.Ho, a.Wo, couts_l);
                const long dhwo = (long)a.Do * a.Ho * a.Wo;
                if (a.want_stats && !train && (N == 1 || dhwo % rows == 0)) {
                    out.stats_off = pool.alloc((size_t)((M + rows - 1) / rows) * couts_l * 2 * 4); out.has_stats = true;
                    out.stats_nrb = (int)(N == 1 ? (M + rows - 1) / rows : dhwo / rows);
                }
                Op op{}; op.kind = OP_GEMM_LIGHT32;
                op.r[0] = ws_ref(a.xa.off); op.r[1] = a.xb.valid ? ws_ref(a.xb.off) : Ref();
                op.r[2] = a.w_over.base != BASE_NULL ? a.w_over : w32_ref(w.w_off);
                op.r[6] = a.no_bias ? Ref() : w_ref(w.b_off); op.r[9] = a.residual.valid ? ws_ref(a.residual.off) : Ref();
                op.r[10] = ws_ref(out.off); op.r[12] = out.has_stats ? ws_ref(out.stats_off) : Ref();
                op.i[0] = (int)M; op.i[1] = cin0; op.i[2] = couts_l; op.i[3] = w.cout_pad; op.i[4] = big; op.i[5] = a.xa.C;
                plan->ops.push_back(op);
                if (recording) { Tape t; t.kind = 0; t.c = a; t.out = out; tape.push_back(t); }
                return out;
            }
        }
        const int kb = x3 ? 32 : 16;
        const int taps = a.k * a.k * a.k, nchunk = cin0 / kb, steps = taps * nchunk;
        static const int f32_bn = ldm_knob("LDM_F32_BN", 0);        // tuning knobs
        static const int f32_wgs = ldm_knob("LDM_F32_WGS", 512);   // two workgroups per CU: 66 -> 73 TFLOP/s over the step
        const int mtiles = (int)((M + 127) / 128);
        int bn = (w.cout_pad % 128) ? 64 : 128;          // 64-wide tiles where 128 would idle half of the MFMA rows
        if (f32_bn == 64 || (f32_bn == 1 && (long)mtiles * (w.cout_pad / 128) < 256)) bn = 64;
        const int ntiles = (w.cout_pad + bn - 1) / bn;
        const long tiles = (long)mtiles * ntiles;
        int sk = 1;
        if (tiles < f32_wgs * 3 / 4) {                   // fill the CUs: K split into deterministic fp32 slabs
            sk = (int)std::min<long>(std::max<long>(1, f32_wgs / tiles), std::max(1, steps / (x3 ? 4 : 8)));
            const int sps = (steps + sk - 1) / sk; sk = (steps + sps - 1) / sps;
        }
        const int couts = a.f32_out ? 0 : rup(w.cout, 32);
        Act out;
        if (!a.f32_out) out = new_act(N, a.Do, a.Ho, a.Wo, couts);
        Op op{}; op.kind = OP_CONV32; op.cc = ConvCfg{2, bn / 64, kb, sk};
        op.r[0] = ws_ref(a.xa.off); op.r[1] = a.xb.valid ? ws_ref(a.xb.off) : Ref();
        op.r[2] = a.w_over.base != BASE_NULL ? a.w_over : w32_ref(w.w_off);
        op.r[6] = a.no_bias ? Ref() : w_ref(w.b_off);
        op.r[8] = a.temb; op.r[9] = a.residual.valid ? ws_ref(a.residual.off) : Ref();
        op.r[10] = a.f32_out ? a.out_ref : ws_ref(out.off);
        int* i = op.i;
        i[0] = a.xa.C; i[1] = a.xb.valid ? a.xb.C : 0;
        i[4] = N; i[5] = a.xa.D; i[6] = a.xa.H; i[7] = a.xa.W; i[8] = a.Do; i[9] = a.Ho; i[10] = a.Wo;
        i[11] = a.k; i[12] = a.stride; i[13] = a.pad; i[14] = a.ups | (a.exact << 1); i[15] = (int)M;
        i[16] = a.f32_out ? rup(w.cout, 32) : couts; i[17] = w.cout_pad; i[18] = a.cout_real ? a.cout_real : w.cout;
        i[19] = nchunk; i[21] = a.temb_stride; i[22] = a.f32_out ? 1 : 0; i[23] = mtiles; i[20] = ntiles;
        if (sk > 1) { partial_bytes = std::max(partial_bytes, (size_t)sk * M * w.cout_pad * 4); partial_fixups.push_back(plan->ops.size()); }
        plan->ops.push_back(op);
        if (sk > 1) { Op f = op; f.kind = OP_FIN32; fin32_stats(f, out, a, N, a.Do * a.Ho * a.Wo, couts); partial_fixups.push_back(plan->ops.size()); plan->ops.push_back(f); }
        if (recording) { Tape t; t.kind = 0; t.c = a; t.out = out; tape.push_back(t); }
        return out;
    }

    Act conv(const ConvArgs& a, std::string tag = "") {
        if (hp) return conv32(a, tag);
        const ConvW& w = *a.w;
        const int cin0 = a.xa.C + (a.xb.valid ? a.xb.C : 0);
        int bk = 64;
        if (a.xa.C % 64 || (a.xb.valid && a.xb.C % 64)) bk = 32;
        int cin1 = 0;
        if (a.w1) {
            cin1 = a.g1a.C + (a.g1b.valid ? a.g1b.C : 0);
            if (a.g1a.C % 64 || (a.g1b.valid && a.g1b.C % 64)) bk = 32;
        }
        if (cin0 != w.cin_s || (a.w1 && cin1 != a.w1->cin_s) || cin0 % 32 || cin1 % 32) {
            err = "conv " + tag + ": channel bookkeeping mismatch"; return Act();
        }
        const int N = a.xa.N;
        const long M = (long)N * a.Do * a.Ho * a.Wo;
        {   // the loaders address each source through a buffer descriptor with 32-bit byte offsets
            const long rows0 = (long)N * a.xa.D * a.xa.H * a.xa.W;
            const long big = std::max(std::max(rows0 * a.xa.C, a.xb.valid ? rows0 * a.xb.C : 0L),
                                      a.w1 ? std::max(M * a.g1a.C, a.g1b.valid ? M * a.g1b.C : 0L) : 0L) * 2;
            if (big >= (1L << 32)) { err = "conv " + tag + ": a source tensor exceeds 4 GiB (split the batch)"; return Act(); }
        }
        {   // the networks' last layer (Cout <= 4, fp32 NCDHW output): conv_thin.h; inference plans and (round 4) the training forward too
            static const int thin = ldm_knob("LDM_CONV_THIN", 1);
            static const long thin_min = ldm_knob("LDM_CONV_THIN_MIN", 512L);
            const int cr = a.cout_real ? a.cout_real : w.cout;
            if (thin && a.f32_out && a.k == 3 && a.stride == 1 && a.pad == 1 && !a.ups && !a.exact && !a.xb.valid && !a.w1 &&
                a.temb.base == BASE_NULL && !a.residual.valid && a.w_over.base == BASE_NULL && cr <= 4 && cin0 % 32 == 0 && cin0 <= 128 &&
                a.xa.D == a.Do && a.xa.H == a.Ho && a.xa.W == a.Wo &&
                // enough blocks to fill the chip and few channel chunks: measured 64 -> 1 at 96^3 247 -> ~100 us; 256 -> 4 at 24^3 (72 blocks, 8 chunks) 22 -> 44 us
                (long)N * ((a.Do + THIN_TD - 1) / THIN_TD) * ((a.Ho + THIN_TH - 1) / THIN_TH) * ((a.Wo + THIN_TW - 1) / THIN_TW) >= thin_min) {
                Op op{}; op.kind = OP_CONV_THIN;
                op.r[0] = ws_ref(a.xa.off); op.r[2] = w_ref(w.w_off); op.r[6] = a.no_bias ? Ref() : w_ref(w.b_off); op.r[10] = a.out_ref;
                op.i[0] = N; op.i[1] = a.xa.D; op.i[2] = a.xa.H; op.i[3] = a.xa.W; op.i[4] = cin0; op.i[5] = w.cout_pad; op.i[6] = cr;
                plan->ops.push_back(op);
                if (recording) { Tape t; t.kind = 0; t.c = a; t.out = Act(); tape.push_back(t); }
                return Act();
            }
        }
        // inference plans run (nearest x2 upsample -> 3^3 conv) as eight 2^3 convs on the source grid (conv_igemm.h, phase mode)
        const bool phase = a.ups == 1 && !a.exact && a.k == 3 && a.stride == 1 && a.pad == 1 && !train && !a.xb.valid && !a.w1 &&
                           a.w_over.base == BASE_NULL && w.wp_off != 0 && phase_enabled() &&
                           a.Do == 2 * a.xa.D && a.Ho == 2 * a.xa.H && a.Wo == 2 * a.xa.W;
        if (gemm_light_ok(a.k, a.stride, a.ups, cin0, !a.w1, !a.f32_out && a.temb.base == BASE_NULL) && light_enabled() &&
            a.xa.D == a.Do && a.xa.H == a.Ho && a.xa.W == a.Wo && M * (long)cin0 * 2 < (1L << 31)) {
            const int big = gemm_light_big(M, w.cout_pad), rows = big ? 64 : 32;
            const int couts_l = rup(w.cout, 32);
            Act out = new_act(N, a.Do, a.Ho, a.Wo, couts_l);
            const long dhwo = (long)a.Do * a.Ho * a.Wo;
            if (a.want_stats && (N == 1 || dhwo % rows == 0)) {
                out.stats_off = pool.alloc((size_t)((M + rows - 1) / rows) * couts_l * 2 * 4); out.has_stats = true;
                out.stats_nrb = (int)(N == 1 ? (M + rows - 1) / rows : dhwo / rows);
            }
            Op op{}; op.kind = OP_GEMM_LIGHT;
            op.r[0] = ws_ref(a.xa.off); op.r[1] = a.xb.valid ? ws_ref(a.xb.off) : Ref();
            op.r[2] = a.w_over.base != BASE_NULL ? a.w_over : w_ref(w.w_off);
            op.r[6] = a.no_bias ? Ref() : w_ref(w.b_off); op.r[9] = a.residual.valid ? ws_ref(a.residual.off) : Ref();
            op.i[5] = a.xa.C;
            op.r[10] = ws_ref(out.off); op.r[12] = out.has_stats ? ws_ref(out.stats_off) : Ref();
            op.i[0] = (int)M; op.i[1] = cin0; op.i[2] = couts_l; op.i[3] = w.cout_pad; op.i[4] = big;
            plan->ops.push_back(op);
            if (recording) { Tape t; t.kind = 0; t.c = a; t.out = out; tape.push_back(t); }
            return out;
        }
        // 64 output channels over a large grid (the AutoencoderKL's full-resolution level): one halo block in LDS per workgroup (conv_block.h)
        if (conv_block_enabled() && a.k == 3 && a.stride == 1 && a.pad == 1 && a.ups == 0 && !a.exact && !a.xb.valid && !a.w1 && !a.f32_out &&
            w.cout_pad == 64 && rup(w.cout, 32) == 64 && cin0 <= conv_block_max_cin() &&      // (w_over: the data-gradient convs' flipped weights, same layout)
            a.xa.D == a.Do && a.xa.H == a.Ho && a.xa.W == a.Wo) {
            const int TH = conv_block_th();
            const int td = (a.Do + BLK_TD - 1) / BLK_TD, th = (a.Ho + TH - 1) / TH, tw = (a.Wo + BLK_TW - 1) / BLK_TW;
            static const long min_blocks = ldm_knob("LDM_CONV_BLOCK_MIN", 512L);
            if ((long)N * td * th * tw >= min_blocks) {
                Act out = new_act(N, a.Do, a.Ho, a.Wo, 64);
                if (a.want_stats) { out.stats_off = pool.alloc((size_t)N * td * th * tw * 64 * 2 * 4); out.has_stats = true; out.stats_nrb = td * th * tw; }
                Op op{}; op.kind = OP_CONV_BLOCK;
                op.r[0] = ws_ref(a.xa.off); op.r[2] = a.w_over.base != BASE_NULL ? a.w_over : w_ref(w.w_off); op.r[6] = a.no_bias ? Ref() : w_ref(w.b_off); op.r[8] = a.temb;
                op.r[9] = a.residual.valid ? ws_ref(a.residual.off) : Ref(); op.r[10] = ws_ref(out.off); op.r[12] = out.has_stats ? ws_ref(out.stats_off) : Ref();
                op.i[0] = N; op.i[1] = a.Do; op.i[2] = a.Ho; op.i[3] = a.Wo; op.i[4] = cin0; op.i[5] = TH; op.i[6] = a.temb_stride;
                plan->ops.push_back(op);
                if (recording) { Tape t; t.kind = 0; t.c = a; t.out = out; tape.push_back(t); }
                return out;
            }
        }
        // 128 output channels (the AutoencoderKL's half-resolution level): eight-wave block kernel with double-buffered halo chunks (conv_block.h)
        if (conv_block128_enabled() && a.k == 3 && a.stride == 1 && a.pad == 1 && a.ups == 0 && !a.exact && !a.xb.valid && !a.w1 && !a.f32_out &&
            w.cout_pad == 128 && rup(w.cout, 32) == 128 && cin0 <= conv_block128_max_cin() && a.xa.D == a.Do && a.xa.H == a.Ho && a.xa.W == a.Wo) {
            const int td = (a.Do + BLK_TD - 1) / BLK_TD, th = (a.Ho + 7) / 8, tw = (a.Wo + BLK_TW - 1) / BLK_TW;
            static const long min_blocks = ldm_knob("LDM_CONV_BLOCK128_MIN", 192L);   // 128 tiles (32^3 x batch 2) leave half the CUs idle: AutoencoderKL training step 18.4 vs 18.6 ms
            // one eight-wave workgroup per CU: it pays where the tiles fit ONE round (48^3: 216 tiles: AutoencoderKL 96^3 encode 2.37 -> 2.34 ms) and
            // loses where they need several (72 x 88 x 56: 693 tiles = 2.7 rounds that take 3: configs[3] encode 6.65 -> 6.89 ms), so: <= CUs tiles
            static const long max_blocks = ldm_xknob("LDM_CONV_BLOCK128_MAX", 0L);
            const long tiles128 = (long)N * td * th * tw;
            if (tiles128 >= min_blocks && tiles128 <= (max_blocks > 0 ? max_blocks : (long)device_cus())) {
                Act out = new_act(N, a.Do, a.Ho, a.Wo, 128);
                if (a.want_stats) { out.stats_off = pool.alloc((size_t)N * td * th * tw * 128 * 2 * 4); out.has_stats = true; out.stats_nrb = td * th * tw; }
                Op op{}; op.kind = OP_CONV_BLOCK;
                op.r[0] = ws_ref(a.xa.off); op.r[2] = a.w_over.base != BASE_NULL ? a.w_over : w_ref(w.w_off); op.r[6] = a.no_bias ? Ref() : w_ref(w.b_off); op.r[8] = a.temb;
                op.r[9] = a.residual.valid ? ws_ref(a.residual.off) : Ref(); op.r[10] = ws_ref(out.off); op.r[12] = out.has_stats ? ws_ref(out.stats_off) : Ref();
                op.i[0] = N; op.i[1] = a.Do; op.i[2] = a.Ho; op.i[3] = a.Wo; op.i[4] = cin0; op.i[5] = 8; op.i[6] = a.temb_stride; op.i[7] = 128;
                plan->ops.push_back(op);
                if (recording) { Tape t; t.kind = 0; t.c = a; t.out = out; tape.push_back(t); }
                return out;
            }
        }
        const int taps = phase ? 8 : a.k * a.k * a.k;
        const bool halo_ok = a.k == 3 && a.stride == 1 && a.pad == 1 && a.ups == 0 && !a.exact && !a.xb.valid && (!a.w1 || (halo_skip_enabled() && bk == 64)) &&
                             a.xa.D == a.Do && a.xa.H == a.Ho && a.xa.W == a.Wo;
        int nchunk0 = cin0 / bk, nchunk1 = cin1 / bk;
        int steps0 = taps * nchunk0, steps1 = nchunk1;
        ConvCfg cc = choose_cfg(M, w.cout_pad, steps0, bk, halo_ok ? N : 0, (long)a.Do * a.Ho * a.Wo, false, steps1);
        const int bm = 64 * cc.wgm, bn = 64 * cc.wgn;
        const int couts = a.f32_out ? 0 : rup(w.cout, 32);
        const int mtiles_pp = (int)(((long)a.xa.D * a.xa.H * a.xa.W + bm - 1) / bm);      // phase mode: tiles per (sample, parity)
        // Two workgroups per CU on the general kernel: with 32-channel K steps a 128-row tile runs on four waves and 78 KiB of LDS, so
        // a CU holds two and one's prologue / epilogue sits under the other's K loop.  Measured on the 96^3 phase-upsample conv of the
        // AutoencoderKL decoder (16 K steps per tile: fixed cost = half of a tile): 474 -> 328 us; 4 x 1 tiles (108 KiB) do not fit twice.
        // Short K loops only (<= 64 steps of 64 channels): the 116-step convs of the configs[3] training step lost 1 % with it.
        if (!cc.halo && cc.bk == 64 && cc.wgm <= 2 && cc.splitk == 1 && two_wg_enabled() && steps0 + steps1 <= 64 &&
            (phase ? (long)N * 8 * mtiles_pp : (M + bm - 1) / bm) * (w.cout_pad / bn) >= 512) {
            cc.bk = 32; nchunk0 = cin0 / 32; nchunk1 = cin1 / 32; steps0 = taps * nchunk0; steps1 = nchunk1;
        }
        Act out;
        if (!a.f32_out) {
            out = new_act(N, a.Do, a.Ho, a.Wo, couts);
            if (a.want_stats) {
                // GroupNorm partial slabs: one row per conv tile (split-K: one per 32 output rows, written by the finalize)
                const long dhwo = (long)a.Do * a.Ho * a.Wo;
                if (cc.splitk > 1) {
                    out.stats_off = pool.alloc((size_t)((M + 31) / 32) * couts * 2 * 4); out.has_stats = true; out.stats_nrb = 0;
                } else if (cc.halo) {
                    out.stats_off = pool.alloc((size_t)N * cc.mtps * couts * 2 * 4); out.has_stats = true; out.stats_nrb = cc.mtps;
                } else if (phase) {                                                  // tiles are per (sample, parity)
                    out.stats_off = pool.alloc((size_t)N * 8 * mtiles_pp * couts * 2 * 4); out.has_stats = true; out.stats_nrb = 8 * mtiles_pp;
                } else if (N == 1 || dhwo % bm == 0) {                               // tiles must not straddle samples
                    out.stats_off = pool.alloc((size_t)((M + bm - 1) / bm) * couts * 2 * 4); out.has_stats = true;
                    out.stats_nrb = (int)(N == 1 ? (M + bm - 1) / bm : dhwo / bm);
                }
            }
        }
        Op op{}; op.kind = OP_CONV; op.cc = cc;
        op.r[0] = ws_ref(a.xa.off); op.r[1] = a.xb.valid ? ws_ref(a.xb.off) : Ref();
        op.r[2] = a.w_over.base != BASE_NULL ? a.w_over : w_ref(phase ? w.wp_off : w.w_off);
        op.r[3] = a.w1 ? ws_ref(a.g1a.off) : Ref(); op.r[4] = (a.w1 && a.g1b.valid) ? ws_ref(a.g1b.off) : Ref();
        op.r[5] = a.w1 ? w_ref(a.w1->w_off) : Ref();
        op.r[6] = a.no_bias ? Ref() : w_ref(w.b_off); op.r[7] = a.w1 ? w_ref(a.w1->b_off) : Ref();
        op.r[8] = a.temb; op.r[9] = a.residual.valid ? ws_ref(a.residual.off) : Ref();
        op.r[10] = a.f32_out ? a.out_ref : ws_ref(out.off);
        op.r[11] = Ref();                                                     // partial slab (fixed up later)
        op.r[12] = out.has_stats ? ws_ref(out.stats_off) : Ref();
        int* i = op.i;
        i[0] = a.xa.C; i[1] = a.xb.valid ? a.xb.C : 0; i[2] = a.w1 ? a.g1a.C : 0; i[3] = (a.w1 && a.g1b.valid) ? a.g1b.C : 0;
        i[4] = N; i[5] = a.xa.D; i[6] = a.xa.H; i[7] = a.xa.W; i[8] = a.Do; i[9] = a.Ho; i[10] = a.Wo;
        i[11] = phase ? 2 : a.k; i[12] = a.stride; i[13] = a.pad; i[14] = phase ? 4 : (a.ups | (a.exact << 1)); i[15] = (int)M;
        i[16] = a.f32_out ? rup(w.cout, 32) : couts; i[17] = w.cout_pad; i[18] = a.cout_real ? a.cout_real : w.cout;
        i[19] = nchunk0; i[20] = nchunk1; i[21] = a.temb_stride; i[22] = a.f32_out ? 1 : 0;
        i[23] = cc.halo ? N * cc.mtps : phase ? N * 8 * mtiles_pp : (int)((M + bm - 1) / bm);
        if (M >= (1L << 31)) { err = "conv " + tag + ": M too large"; return Act(); }
        if (cc.splitk > 1) {
            partial_bytes = std::max(partial_bytes, (size_t)cc.splitk * M * w.cout_pad * 4);
            partial_fixups.push_back(plan->ops.size());
        }
        (void)bn;
        plan->ops.push_back(op);
        if (cc.splitk > 1) {
            Op f = op; f.kind = OP_FINALIZE;
            partial_fixups.push_back(plan->ops.size());
            plan->ops.push_back(f);
        }
        if (recording) { Tape t; t.kind = 0; t.c = a; t.out = out; tape.push_back(t); }
        return out;
    }

    // ---- split-K finalize + the GroupNorm that consumes it, in one launch (fin_gn.h) ------------------------------------------
    // Called right after conv() returned `raw`: if the plan's last op is that conv's OP_FINALIZE and one workgroup can own a whole
    // (sample, group) of the tensor, the finalize becomes an OP_FIN_GN that also writes GroupNorm(+SiLU)(raw) into *y, and the conv in
    // front of it writes its slabs planar (ConvCfg::slab_lg).  keep_raw = false: nobody else reads the un-normalised tensor (a
    // ResBlock's conv1 output): it is not written at all and its workspace block returns to the pool.
    // History: rounds 2 / 4 built this with a grid barrier / an arrival counter between the launch's workgroups and lost 10 % / 2.2 %
    // (DESIGN.md 3.2b); the group-owning form has no exchange.  LDM_FIN_GN=0 switches it off.
    static bool fin_gn_enabled() { static const int v = ldm_knob("LDM_FIN_GN", 1); return v != 0; }
    std::map<size_t, Act> prenorm;                   // un-normalised activation (by offset) -> its GroupNorm output, produced by an OP_FIN_GN
    bool fuse_fin_gn(Act& raw, bool keep_raw, const GnW& g, int groups, float eps, bool silu, Act* y) {
        if (hp || train || !fin_gn_enabled() || plan->ops.size() < 2 || !raw.valid) return false;
        Op& f = plan->ops.back();
        Op& cv = plan->ops[plan->ops.size() - 2];
        if (f.kind != OP_FINALIZE || cv.kind != OP_CONV || f.i[22] || f.r[10].base != BASE_WS || f.r[10].off != raw.off) return false;
        const int C = raw.C, N = raw.N, DHW = raw.D * raw.H * raw.W;
        if (C != g.C || !fin_gn_ok(C, groups, DHW) || (f.i[14] & 4)) return false;      // (phase-mode convs scatter their rows: not planar)
        const int lg = fin_gn_lg(C, groups);
        // one CU streams a whole group's slabs (~40 - 50 GB/s per CU measured): it pays where that is less than what the separate finalize
        // + GroupNorm cost.  6^3 x 16 channels x 24 splits = 332 KB per group: 12.3 vs 9 + 8 us (per-op trace); 12^3 x 8 channels x 9 splits =
        // 498 KB: 21 - 25 vs 8.5 + 8 us: the 12^3 level keeps its three launches (profiles/r05_ab_fin_gn.txt)
        static const long max_kb = ldm_knob("LDM_FIN_GN_MAX_KB", 400);
        if ((long)cv.cc.splitk * DHW * (C / groups) * 4 > max_kb * 1024) return false;
        if (f.i[17] % (1 << lg)) return false;
        Act out = new_act(N, raw.D, raw.H, raw.W, C);
        f.kind = OP_FIN_GN;
        f.r[3] = w_ref(g.g_off); f.r[4] = w_ref(g.b_off); f.r[5] = ws_ref(out.off);
        f.r[12] = Ref();                                                        // no statistics slab: the statistics never leave the launch
        f.i[0] = groups; f.i[1] = silu ? 1 : 0; f.i[2] = lg; f.f[0] = eps;
        f.cc.slab_lg = lg; cv.cc.slab_lg = lg; cv.r[12] = Ref();
        if (raw.has_stats) { pool.release(raw.stats_off); raw.has_stats = false; }
        if (!keep_raw) { f.r[10] = Ref(); pool.release(raw.off); raw.valid = false; }
        *y = out;
        return true;
    }

    // ---- GroupNorm (+SiLU) over (xa | xb) -> contiguous bf16 --------------------------------------------
    // fp32 inference plans: LDM_X3_HALO (default 1) runs the 3^3 stride-1 convs behind a GroupNorm with >= LDM_X3_HALO_ROWS (default 1) output rows
    // as conv3_halo_kernel on the bf16 (hi | lo) split of the normalised tensor (ConvParams::x3_n): the GroupNorm writes the split
    // instead of fp32 values (same bytes), the conv leaves fp32 slabs, finalize_f32_kernel applies the epilogue.  Measured at 24^3:
    // 256 -> 256 channels 250 -> 160 us, 512 -> 256 493 -> 297 us (conv_x3_kernel splits every operand on its way into LDS, in every tile).
    static bool x3_halo_ok(const ConvW& w, long rows, int C) {
        static const int on = ldm_xknob("LDM_X3_HALO", 1);
        static const long min_rows = ldm_xknob("LDM_X3_HALO_ROWS", 1L);
        static const int f32_x3 = ldm_knob("LDM_F32_X3", 1);
        return on && f32_x3 != 0 && halo_enabled() && w.k == 3 && w.x3_off != (size_t)-1 && C == w.cin_s && C % 64 == 0 && rows >= min_rows &&
               rows * C * 4 < (1L << 32);                  // the halo kernel addresses its voxel operand with 32-bit byte offsets
    }
    // next3: the 3^3 stride-1 pad-1 convolution that is the ONLY reader of the result (resblock), or null
    Act gn_apply(const GnW& g, const Act& xa, const Act& xb, int groups, float eps, bool silu, const ConvW* next3 = nullptr) {
        const int C = xa.C + (xb.valid ? xb.C : 0);
        if (C != g.C || C % 8 || (C / groups) * groups != C || xa.C % 8) { err = "groupnorm: channel mismatch"; return Act(); }
        const int DHW = xa.D * xa.H * xa.W, N = xa.N;
        size_t ab_off = 0, mr_off = 0;               // training: scale/shift and mean/rstd are saved per instance
        if (train) { ab_off = pool.alloc((size_t)N * C * 2 * 4); mr_off = pool.alloc((size_t)N * groups * 2 * 4); }
        else gnab_bytes = std::max(gnab_bytes, (size_t)N * C * 2 * 4);
        auto ab_ref = [&](Op& o_) {
            if (train) { o_.r[5] = ws_ref(ab_off); if (o_.kind != OP_GN_APPLY && o_.kind != OP_GN_APPLY32) o_.r[6] = ws_ref(mr_off); }
            else gnab_fixups.push_back(plan->ops.size());
        };
        auto blocks_ok = [&](const Act& t) { return t.has_stats && (t.stats_nrb > 0 || N == 1 || DHW % 32 == 0); };
        auto blocks_of = [&](const Act& t) { return t.stats_nrb > 0 ? t.stats_nrb : (N == 1 ? (DHW + 31) / 32 : DHW / 32); };
        bool fused = blocks_ok(xa) && (!xb.valid || blocks_ok(xb));
        const int nrb_tot = fused ? blocks_of(xa) + (xb.valid ? blocks_of(xb) : 0) : 0;
        if (hp && (train || C / groups > 64 || nrb_tot > 1024)) fused = false;
        if (hp && fused) {                             // fp32 inference: the producing convs' epilogues left the partials; fold + apply in one launch
            Act out = new_act(N, xa.D, xa.H, xa.W, C);
            out.hl = next3 && x3_halo_ok(*next3, (long)N * DHW, C);
            const int slices = (C + 63) / 64;
            int chunks = std::max(1, std::min(256 / (slices * N), (DHW + 15) / 16));
            const int rpb = rup((DHW + chunks - 1) / chunks, 16);
            chunks = (DHW + rpb - 1) / rpb;
            Op ap{}; ap.kind = OP_GN_APPLY32;
            ap.r[0] = ws_ref(xa.off); ap.r[1] = xb.valid ? ws_ref(xb.off) : Ref(); ap.r[3] = ws_ref(out.off);
            ap.r[6] = w_ref(g.g_off); ap.r[7] = w_ref(g.b_off);
            ap.r[8] = ws_ref(xa.stats_off); ap.r[9] = xb.valid ? ws_ref(xb.stats_off) : Ref();
            ap.i[0] = xa.C; ap.i[1] = xb.valid ? xb.C : 0; ap.i[2] = DHW; ap.i[3] = N; ap.i[4] = silu ? 1 : 0;
            ap.i[5] = 1; ap.i[6] = groups; ap.i[7] = rpb; ap.i[8] = chunks; ap.i[9] = out.hl ? 1 : 0;
            ap.i[10] = blocks_of(xa); ap.i[11] = xb.valid ? blocks_of(xb) : 0; ap.f[0] = eps;
            plan->ops.push_back(ap);
            return out;
        }
        if (fused && C / groups <= 64 && nrb_tot <= 512) {   // few slab rows: ONE launch folds them per block and applies
            Act out = new_act(N, xa.D, xa.H, xa.W, C);
            const int slices = (C + 63) / 64;
            static const int gn_blocks = ldm_xknob("LDM_GN_BLOCKS", 512);   // tuning knob: 512 = two blocks per CU at 24^3 (the apply is VALU-latency bound at one wave per SIMD: +0.3 % over the step; 1024: -1 %, every block folds the slabs again)
            int chunks = std::max(1, std::min(gn_blocks / (slices * N), (DHW + 31) / 32));   // one round of the 256 CUs: the slab fold is per block
            int rpb = rup((DHW + chunks - 1) / chunks, 32);
            chunks = (DHW + rpb - 1) / rpb;
            Op f{}; f.kind = OP_GN_FUSED;
            f.r[0] = ws_ref(xa.off); f.r[1] = xb.valid ? ws_ref(xb.off) : Ref(); f.r[2] = w_ref(g.g_off); f.r[3] = w_ref(g.b_off);
            f.r[7] = ws_ref(xa.stats_off); f.r[8] = xb.valid ? ws_ref(xb.stats_off) : Ref(); f.r[9] = ws_ref(out.off);
            if (train) { f.r[5] = ws_ref(ab_off); f.r[6] = ws_ref(mr_off); }
            f.i[0] = xa.C; f.i[1] = xb.valid ? xb.C : 0; f.i[2] = blocks_of(xa); f.i[3] = xb.valid ? blocks_of(xb) : 0;
            f.i[4] = groups; f.i[5] = DHW; f.i[6] = N; f.i[7] = silu ? 1 : 0; f.i[8] = rpb; f.i[9] = chunks; f.f[0] = eps;
            plan->ops.push_back(f);
            if (recording) {
                Tape t; t.kind = 1; t.g = &g; t.xa = xa; t.xb = xb; t.out = out; t.ab_off = ab_off; t.mr_off = mr_off;
                t.groups = groups; t.silu = silu; tape.push_back(t);
            }
            return out;
        }
        if (fused) {                                   // partials came with the tensors: one small reduce
            Op f{}; f.kind = OP_GN_PREP;
            f.r[0] = ws_ref(xa.stats_off); f.r[1] = xb.valid ? ws_ref(xb.stats_off) : Ref();
            f.r[2] = w_ref(g.g_off); f.r[3] = w_ref(g.b_off);
            f.i[0] = xa.C; f.i[1] = xb.valid ? xb.C : 0; f.i[2] = blocks_of(xa); f.i[3] = groups;
            f.i[4] = DHW; f.i[5] = N; f.i[6] = xb.valid ? blocks_of(xb) : 0; f.f[0] = eps;
            ab_ref(f); plan->ops.push_back(f);
        } else {
            const int cvec = C / 8;
            const int rows_par = std::max(1, 256 / cvec);
            // fp32 inference plans: statistics fold + apply in one launch (gn32_fold_apply_kernel); one partial row per CU keeps the fold short
            static const bool gn32_fold = (ldm_knob("LDM_GN32_FOLD", 1) != 0);
            const bool fold32 = hp && !train && gn32_fold && C / groups <= 64;
            int nslab = std::min((DHW + rows_par - 1) / rows_par, std::max(1, (fold32 ? 256 : 512) / N));
            int rps = (DHW + nslab - 1) / nslab;
            nslab = (DHW + rps - 1) / rps;
            gnpart_bytes = std::max(gnpart_bytes, (size_t)N * nslab * C * 2 * 4);
            Op st{}; st.kind = hp ? OP_GN_STATS32 : OP_GN_STATS;
            st.r[0] = ws_ref(xa.off); st.r[1] = xb.valid ? ws_ref(xb.off) : Ref();
            st.i[0] = xa.C; st.i[1] = xb.valid ? xb.C : 0; st.i[2] = DHW; st.i[3] = nslab; st.i[4] = rps; st.i[5] = N;
            gnpart_fixups.push_back(plan->ops.size()); plan->ops.push_back(st);
            if (fold32) {
                Act out = new_act(N, xa.D, xa.H, xa.W, C);
                out.hl = next3 && x3_halo_ok(*next3, (long)N * DHW, C);
                const int slices = (C + 63) / 64;
                int chunks = std::max(1, std::min(256 / (slices * N), (DHW + 15) / 16));
                const int rpb = rup((DHW + chunks - 1) / chunks, 16);
                chunks = (DHW + rpb - 1) / rpb;
                Op ap{}; ap.kind = OP_GN_APPLY32;
                ap.r[0] = ws_ref(xa.off); ap.r[1] = xb.valid ? ws_ref(xb.off) : Ref(); ap.r[3] = ws_ref(out.off);
                ap.r[6] = w_ref(g.g_off); ap.r[7] = w_ref(g.b_off);
                ap.i[0] = xa.C; ap.i[1] = xb.valid ? xb.C : 0; ap.i[2] = DHW; ap.i[3] = N; ap.i[4] = silu ? 1 : 0;
                ap.i[5] = nslab; ap.i[6] = groups; ap.i[7] = rpb; ap.i[8] = chunks; ap.i[9] = out.hl ? 1 : 0; ap.f[0] = eps;
                gnpart_fixups.push_back(plan->ops.size()); plan->ops.push_back(ap);
                return out;
            }
            Op f{}; f.kind = OP_GN_FINALIZE;
            f.r[1] = w_ref(g.g_off); f.r[2] = w_ref(g.b_off);
            f.i[0] = nslab; f.i[1] = C; f.i[2] = groups; f.i[3] = DHW; f.i[4] = N; f.f[0] = eps;
            gnpart_fixups.push_back(plan->ops.size()); ab_ref(f); plan->ops.push_back(f);
        }
        Act out = new_act(N, xa.D, xa.H, xa.W, C);
        Op ap{}; ap.kind = hp ? OP_GN_APPLY32 : OP_GN_APPLY;
        ap.r[0] = ws_ref(xa.off); ap.r[1] = xb.valid ? ws_ref(xb.off) : Ref(); ap.r[3] = ws_ref(out.off);
        ap.i[0] = xa.C; ap.i[1] = xb.valid ? xb.C : 0; ap.i[2] = DHW; ap.i[3] = N; ap.i[4] = silu ? 1 : 0;
        ab_ref(ap); plan->ops.push_back(ap);
        if (recording) {
            Tape t; t.kind = 1; t.g = &g; t.xa = xa; t.xb = xb; t.out = out; t.ab_off = ab_off; t.mr_off = mr_off;
            t.groups = groups; t.silu = silu; tape.push_back(t);
        }
        return out;
    }

    // first conv of a network as im2col + light GEMM (inference plans; the training tape keeps the 3^3 form): r0 / r1 = the fp32
    // NCDHW inputs (x | cond).  Returns an invalid Act when the model has no derived weights for it.
    static bool im2col_enabled() { return ldm_knob("LDM_CONV_IM2COL", 1) != 0; }
    Act conv_in_im2col(const std::string& name, Ref r0, Ref r1, int N, int D, int H, int W, int cin, bool internal) {
        auto it = m->convs.find(name + ".im2col");
        if (train || hp || it == m->convs.end() || !im2col_enabled() || !light_enabled()) return Act();
        const ConvW& wi = it->second;
        Act pm = new_act(N, D, H, W, wi.cin_s);
        Op o{}; o.kind = OP_IM2COL; o.r[0] = r0; o.r[1] = r1; o.r[2] = ws_ref(pm.off);
        o.i[0] = N; o.i[1] = cin; o.i[2] = wi.cin_s; o.i[3] = D; o.i[4] = H; o.i[5] = W; o.i[6] = internal ? 1 : 0;
        plan->ops.push_back(o);
        ConvArgs a; a.xa = pm; a.w = &wi; a.k = 1; a.pad = 0; a.Do = D; a.Ho = H; a.Wo = W;
        Act out = conv(a, name + ".im2col");
        free_act(pm);
        return out;
    }

    // ---- blocks ---------------------------------------------------------------------------------------
    Act conv3(const std::string& name, const Act& x, int stride = 1, int pad = 1, int ups = 0) {
        ConvArgs a; a.xa = x; a.w = &m->convs.at(name); a.k = 3; a.stride = stride; a.pad = pad; a.ups = ups;
        const int Du = x.D << ups, Hu = x.H << ups, Wu = x.W << ups;
        if (stride == 1) { a.Do = Du; a.Ho = Hu; a.Wo = Wu; }
        else if (pad == 1) { a.Do = (Du + 2 - 3) / 2 + 1; a.Ho = (Hu + 2 - 3) / 2 + 1; a.Wo = (Wu + 2 - 3) / 2 + 1; }
        else { a.Do = (Du + 1 - 3) / 2 + 1; a.Ho = (Hu + 1 - 3) / 2 + 1; a.Wo = (Wu + 1 - 3) / 2 + 1; }   // F.pad(0,1) + s2 p0
        return conv(a, name);
    }

    // ResBlock (UNet: with temb; VAE: without).  xb = skip tensor concatenated after xa (up path) or invalid.
    // next_gn: the GroupNorm (no SiLU) of the attention block that reads this ResBlock's output next, or null: where conv2 is split
    // over K its finalize then also produces that GroupNorm's output (prenorm), which attention() picks up
    Act resblock(const std::string& p, const Act& xa, const Act& xb, int cout, int groups, float eps,
                 const std::string& skip_name, bool with_temb, const GnW* next_gn = nullptr) {
        const int cin = xa.C + (xb.valid ? xb.C : 0);
        // fp32 precision: the 1x1 skip projection runs as its own conv and enters conv2 as its residual (the bf16 plans fuse it
        // into conv2 as extra K steps).  A second stream for it (and for the time-embedding chain) was measured and removed: one
        // fork / join pair per step costs 5 % of the step under graph replay, fourteen cost 8 %
        Act sk;
        if (cin != cout && hp) {
            ConvArgs cs; cs.xa = xa; cs.xb = xb; cs.w = &m->convs.at(p + skip_name); cs.k = 1; cs.pad = 0;
            cs.Do = xa.D; cs.Ho = xa.H; cs.Wo = xa.W; cs.want_stats = false;
            sk = conv(cs, p + skip_name);
            if (!sk.valid) return Act();
        }
        Act h0 = gn_apply(m->gns.at(p + ".norm1"), xa, xb, groups, eps, true, &m->convs.at(p + ".conv1"));
        if (!h0.valid) return Act();
        ConvArgs c1; c1.xa = h0; c1.w = &m->convs.at(p + ".conv1"); c1.Do = xa.D; c1.Ho = xa.H; c1.Wo = xa.W;
        if (with_temb) {
            c1.temb = ws_ref(temb_all_off + (size_t)m->tproj_row.at(p) * 4); c1.temb_stride = tproj_stride;
            c1.temb_row = m->tproj_row.at(p);
        }
        Act h1 = conv(c1, p + ".conv1");
        free_act(h0);
        if (!h1.valid) return Act();
        Act h2;
        if (!fuse_fin_gn(h1, false, m->gns.at(p + ".norm2"), groups, eps, true, &h2)) {
            h2 = gn_apply(m->gns.at(p + ".norm2"), h1, Act(), groups, eps, true, &m->convs.at(p + ".conv2"));
            free_act(h1);
        }
        if (!h2.valid) return Act();
        ConvArgs c2; c2.xa = h2; c2.w = &m->convs.at(p + ".conv2"); c2.Do = xa.D; c2.Ho = xa.H; c2.Wo = xa.W;
        if (sk.valid) c2.residual = sk;
        else if (cin != cout) { c2.g1a = xa; c2.g1b = xb; c2.w1 = &m->convs.at(p + skip_name); }
        else { if (xb.valid) { err = "resblock: identity skip with concat input"; return Act(); } c2.residual = xa; }
        Act out = conv(c2, p + ".conv2");
        free_act(h2);
        if (sk.valid) free_act(sk);
        if (next_gn && out.valid && tap_mode != 2) {
            Act yn;
            if (fuse_fin_gn(out, true, *next_gn, groups, eps, false, &yn)) prenorm[out.off] = yn;
        }
        return out;
    }

    Act attention32(const std::string& p, const Act& x, int head_ch, int groups, float eps) {
        const int C = x.C;
        if (head_ch <= 0) head_ch = C;                  // single head (AutoencoderKL attention blocks)
        if (C % head_ch || (head_ch != 32 && head_ch != 64 && head_ch != 128 && head_ch != 256)) {
            err = "attention: head dimension must be 32, 64, 128 or 256 (" + p + ")"; return Act();
        }
        Act hn = gn_apply(m->gns.at(p + ".norm"), x, Act(), groups, eps, false);
        if (!hn.valid) return Act();
        ConvArgs q; q.xa = hn; q.w = &m->convs.at(p + ".attn.qkv"); q.k = 1; q.pad = 0; q.Do = x.D; q.Ho = x.H; q.Wo = x.W;
        Act qkv = conv(q, p + ".qkv");
        free_act(hn);
        if (!qkv.valid) return Act();
        Act o = new_act(x.N, x.D, x.H, x.W, C);
        Op at{}; at.kind = OP_ATTN32; at.r[0] = ws_ref(qkv.off); at.r[1] = ws_ref(o.off);
        at.i[0] = x.N; at.i[1] = x.D * x.H * x.W; at.i[2] = C; at.i[3] = C / head_ch; at.i[4] = head_ch; at.f[0] = 1.0f / sqrtf((float)head_ch);
        // inference plans: QK^T and PV as 3 x bf16 MFMAs on hi / lo splits (f32_path.h, X3), like their convolutions; LDM_ATTN_X3=0: the exact fp32 MFMA
        static const int attn_x3 = ldm_knob("LDM_ATTN_X3", 1);
        at.i[5] = (!train && attn_x3 && ldm_knob("LDM_F32_X3", 1) != 0) ? 1 : 0;
        size_t lse_off = 0;
        if (train) {
            lse_off = pool.alloc((size_t)x.N * (C / head_ch) * at.i[1] * 4); at.r[2] = ws_ref(lse_off);
        }
        plan->ops.push_back(at);
        if (recording) { Tape t; t.kind = 2; t.qkv = qkv; t.o = o; t.lse_off = lse_off; t.head_ch = head_ch; tape.push_back(t); }
        free_act(qkv);
        ConvArgs pr; pr.xa = o; pr.w = &m->convs.at(p + ".attn.out_proj"); pr.k = 1; pr.pad = 0;
        pr.Do = x.D; pr.Ho = x.H; pr.Wo = x.W; pr.residual = x;
        Act out = conv(pr, p + ".out_proj");
        free_act(o);
        return out;
    }

    Act attention(const std::string& p, const Act& x, int head_ch, int groups, float eps) {
        const int C = x.C;
        if (hp) return attention32(p, x, head_ch, groups, eps);
        if (head_ch <= 0) head_ch = C;                  // single head (AutoencoderKL attention blocks)
        if (C % head_ch || (head_ch != 32 && head_ch != 64 && head_ch != 128 && head_ch != 256)) {
            err = "attention: head dimension must be 32, 64, 128 or 256 (" + p + ": " + std::to_string(head_ch) + ")"; return Act();
        }
        Act hn;
        { auto it = prenorm.find(x.off); if (it != prenorm.end()) { hn = it->second; prenorm.erase(it); } }
        if (!hn.valid) hn = gn_apply(m->gns.at(p + ".norm"), x, Act(), groups, eps, false);
        if (!hn.valid) return Act();
        ConvArgs q; q.xa = hn; q.w = &m->convs.at(p + ".attn.qkv"); q.k = 1; q.pad = 0; q.Do = x.D; q.Ho = x.H; q.Wo = x.W;
        q.want_stats = false;
        Act qkv = conv(q, p + ".qkv");
        free_act(hn);
        if (!qkv.valid) return Act();
        Act o = new_act(x.N, x.D, x.H, x.W, C);
        Op at{}; at.kind = OP_ATTN; at.r[0] = ws_ref(qkv.off); at.r[1] = ws_ref(o.off);
        at.i[0] = x.N; at.i[1] = x.D * x.H * x.W; at.i[2] = C; at.i[3] = C / head_ch; at.i[4] = head_ch; at.f[0] = 1.0f / sqrtf((float)head_ch);
        size_t lse_off = 0;
        if (train) { lse_off = pool.alloc((size_t)x.N * (C / head_ch) * at.i[1] * 4); at.r[2] = ws_ref(lse_off); }
        plan->ops.push_back(at);
        if (recording) { Tape t; t.kind = 2; t.qkv = qkv; t.o = o; t.lse_off = lse_off; t.head_ch = head_ch; tape.push_back(t); }
        free_act(qkv);
        ConvArgs pr; pr.xa = o; pr.w = &m->convs.at(p + ".attn.out_proj"); pr.k = 1; pr.pad = 0;
        pr.Do = x.D; pr.Ho = x.H; pr.Wo = x.W; pr.residual = x;
        Act out = conv(pr, p + ".out_proj");
        free_act(o);
        return out;
    }


    // ---- backward pass (training plans) -------------------------------------------------------------------
    // Gradients w.r.t. activations are bf16 NDHWC tensors in the same workspace; gslot maps a forward activation
    // (by offset) to the tensor holding the sum of the contributions emitted so far.  Kernels that can add an
    // incoming gradient while they write (conv epilogue `residual`, GroupNorm backward `acc`) always write to a
    // fresh buffer; plain pass-through contributions (identity skips) alias the producer's buffer.
    // Parameter gradients go to one flat fp32 buffer (BASE_IO4) in parameter order, MONAI tensor layouts.
    std::map<size_t, Act> gslot;
    size_t dw_off = 0, vec_off = 0, dtemb_off = 0;     // staging of the conv being differentiated (fresh per conv: exports run at the end)
    std::vector<ExportDesc> exp_descs; std::vector<int2> exp_map;
    std::vector<WtDesc> wt_descs; std::vector<int2> wt_map;
    size_t sin_off = 0, e1_off = 0, e2_off = 0;      // saved time-embedding MLP activations
    static Ref grad_ref(int64_t elem_off) { Ref r; r.base = BASE_IO4; r.off = (size_t)elem_off * 4; return r; }

    Act take_grad(const Act& a) const { auto it = gslot.find(a.off); return it == gslot.end() ? Act() : it->second; }
    void add_grad_alias(const Act& target, const Act& g) {
        Act cur = take_grad(target);
        if (!cur.valid) { gslot[target.off] = g; return; }
        Act sum = new_act(g.N, g.D, g.H, g.W, g.C);
        Op o{}; o.kind = OP_ADD; o.r[0] = ws_ref(cur.off); o.r[1] = ws_ref(g.off); o.r[2] = ws_ref(sum.off);
        o.i[0] = (int)(g.rows() * g.C / (hp ? 4 : 8)); o.i[1] = hp ? 1 : 0;
        plan->ops.push_back(o);
        gslot[target.off] = sum;
    }
    void emit_export(size_t src_off, int taps, int rows_total, int ld, int row_off, int col_off, int cout, int cin, int64_t flat_off,
                     int nsplit = 1) {
        ExportDesc e{}; e.src_off = (long)src_off; e.dst_off = (long)flat_off; e.slab_stride = (long)taps * rows_total * ld;
        e.taps = taps; e.rows_total = rows_total; e.ld = ld; e.row_off = row_off; e.col_off = col_off; e.cout = cout; e.cin = cin; e.nsplit = nsplit;
        const int nb = ((cin + 63) / 64) * cout;
        for (int b = 0; b < nb; ++b) exp_map.push_back(make_int2((int)exp_descs.size(), b));
        exp_descs.push_back(e);
    }
    void export_conv_weights(const ConvW& w, int ld) {           // staging [taps][w.cout][ld] -> every parameter of the slot
        for (const ParamDesc& d : m->params)
            if ((d.kind == PK_CONV_W) && d.dst_off == w.w_off)
                emit_export(dw_off, d.k * d.k * d.k, w.cout, ld, d.row_off, 0, d.cout, d.cin, d.flat_off, cur_ksplit);
    }
    void export_bias(const ConvW& w) {                           // staging vector [couts] -> bias parameter(s) of the slot
        for (const ParamDesc& d : m->params)
            if (d.kind == PK_VEC_F32 && d.dst_off >= w.b_off && d.dst_off < w.b_off + (size_t)w.cout_pad * 4)
                emit_export(vec_off, 1, 1, 0, 0, (int)((d.dst_off - w.b_off) / 4), 1, d.cout, d.flat_off);
    }
    // column sums of a bf16 gradient tensor: slab partials (shared scratch) + fold
    void emit_colsum(const Act& g, bool per_sample, Ref out, int count, int out_stride) {
        const int C = g.C, DHW = g.D * g.H * g.W, N = g.N;
        const int cvec = C / 8, rows_par = std::max(1, 256 / cvec);
        int nslab = std::min((DHW + rows_par - 1) / rows_par, std::max(1, 512 / N));
        const int rps = (DHW + nslab - 1) / nslab; nslab = (DHW + rps - 1) / rps;
        if (g.cs_off && colsum_batched() && out.base == BASE_WS) {       // the GroupNorm backward that wrote g left its column sums
            ColsumDesc d{}; d.partial_off = (long)g.cs_off; d.out_off = (long)out.off; d.N = N; d.nslab = g.cs_rows; d.C = C;
            d.accumulate_over_n = per_sample ? 0 : 1; d.count = count; d.out_stride = out_stride; d.gx = (count + 15) / 16;
            const int nb = d.gx * (per_sample ? N : 1);
            for (int b = 0; b < nb; ++b) cs_map.push_back(make_int2((int)cs_descs.size(), b));
            cs_descs.push_back(d);
            return;
        }
        Op st{}; st.kind = hp ? OP_GN_STATS32 : OP_GN_STATS; st.r[0] = ws_ref(g.off);
        st.i[0] = C; st.i[1] = 0; st.i[2] = DHW; st.i[3] = nslab; st.i[4] = rps; st.i[5] = N;
        if (colsum_batched() && out.base == BASE_WS) {
            // the partials get their own block (kept to the end of the plan) and the finalize joins ONE batched launch
            // (flush_colsums) in front of the first consumer of any of these sums
            const size_t poff = pool.alloc((size_t)N * nslab * C * 2 * 4);
            st.r[4] = ws_ref(poff); plan->ops.push_back(st);
            ColsumDesc d{}; d.partial_off = (long)poff; d.out_off = (long)out.off; d.N = N; d.nslab = nslab; d.C = C;
            d.accumulate_over_n = per_sample ? 0 : 1; d.count = count; d.out_stride = out_stride; d.gx = (count + 15) / 16;
            const int nb = d.gx * (per_sample ? N : 1);
            for (int b = 0; b < nb; ++b) cs_map.push_back(make_int2((int)cs_descs.size(), b));
            cs_descs.push_back(d);
            return;
        }
        gnpart_bytes = std::max(gnpart_bytes, (size_t)N * nslab * C * 2 * 4);
        gnpart_fixups.push_back(plan->ops.size()); plan->ops.push_back(st);
        Op cs{}; cs.kind = OP_COLSUM; cs.r[0] = out;
        cs.i[0] = N; cs.i[1] = nslab; cs.i[2] = C; cs.i[3] = per_sample ? 0 : 1; cs.i[4] = count; cs.i[5] = out_stride;
        gnpart_fixups.push_back(plan->ops.size()); plan->ops.push_back(cs);
    }
    std::vector<ColsumDesc> cs_descs; std::vector<int2> cs_map;
    static bool gnb_fold_enabled() { return ldm_xknob("LDM_GNB_FOLD", 1) != 0; }
    static bool colsum_batched() { return ldm_xknob("LDM_COLSUM_BATCH", 1) != 0; }
    size_t cs_flushed = 0, exp_flushed = 0;              // blocks of cs_map / exp_map already launched by an earlier flush
    void flush_colsums() {
        if (cs_map.size() > cs_flushed) {
            Op o{}; o.kind = OP_COLSUM_BATCH; o.i[0] = (int)cs_flushed; o.i[1] = (int)cs_map.size(); plan->ops.push_back(o);
            cs_flushed = cs_map.size();
        }
    }
    void flush_exports() {
        if (exp_map.size() > exp_flushed) {
            Op o{}; o.kind = OP_EXPORT_BATCH; o.i[0] = (int)exp_flushed; o.i[1] = (int)exp_map.size(); plan->ops.push_back(o);
            exp_flushed = exp_map.size();
        }
    }
    // Gradient buckets: parameters are registered in execution order, so the backward walk finishes the flat gradient buffer from
    // its end towards its front.  Whenever bucket_elems more elements are final, their staged gradients are exported and an OP_BUCKET
    // marks the range: with a communicator attached (ldm_model_set_grad_sync) its all-reduce starts there, on the comm stream, while
    // the launch stream carries on with backward (what DistributedDataParallel's bucketed hooks do: 3d_ldm/train_diffusion.py:147-149).
    int64_t bucket_hi = 0, done_from = 0;
    int64_t tail_reserved = 0;                           // parameters below this offset are produced after the tape walk (time-embedding MLP)
    static int64_t bucket_elems() { const char* e = getenv("LDM_GRAD_BUCKET_MB"); const long mb = e ? atol(e) : 48; return (int64_t)(mb < 1 ? 1 : mb) * 262144; }
    void note_done(int64_t) {}                            // superseded by the pending_end bookkeeping of backward_all
    void close_bucket(bool force) {
        if (done_from >= bucket_hi) return;
        if (!force && bucket_hi - done_from < bucket_elems()) return;
        flush_colsums(); flush_exports();
        Op o{}; o.kind = OP_BUCKET; o.i[0] = (int)done_from; o.i[1] = (int)(bucket_hi - done_from); plan->ops.push_back(o);
        bucket_hi = done_from;
    }
    void emit_wgrad(const Act& dy, const Act& x, int cout, int cin, int ld, int ci_off, int k, int stride, int pad, int ups) {
        Op o{}; o.kind = OP_WGRAD; o.r[0] = ws_ref(dy.off); o.r[1] = ws_ref(x.off); o.r[2] = ws_ref(dw_off);
        int* i = o.i;
        i[0] = dy.C; i[1] = x.C; i[2] = cout; i[3] = cin; i[4] = ld; i[5] = ci_off; i[6] = x.N; i[7] = x.D; i[8] = x.H; i[9] = x.W;
        i[10] = dy.D; i[11] = dy.H; i[12] = dy.W; i[13] = k; i[14] = stride; i[15] = pad; i[16] = ups; i[17] = (int)dy.rows();
        i[18] = cur_ksplit; i[19] = cur_rows_total; i[21] = hp ? 1 : 0;
        plan->ops.push_back(o);
    }
    int cur_ksplit = 1, cur_rows_total = 0;   // voxel split / slab rows of the gradient being staged
    // dX of one source tensor `src` (channels [ci_off, ci_off + src.C) of the conv input) given dY.
    bool emit_dgrad(const Act& dy, const Act& src, const ConvW& w, int ci_off, int k, int stride, int pad, int ups) {
        if (stride == 2 && !(k == 3 && (pad == 1 || pad == 0))) { err = "backward of a stride-2 conv needs k = 3, pad 0 | 1"; return false; }
        const int taps = k * k * k;
        const int rows = rup(src.C, 64), cols = rup(w.cout, 32);
        const size_t wt_off = pool.alloc((size_t)taps * rows * cols * (hp ? 4 : 2));
        {   // transposed once per backward by the batched kernel at the head of the backward plan
            WtDesc e{}; e.src_off = (long)w.w_off; e.dst_off = (long)wt_off; e.taps = taps; e.cout = w.cout; e.cout_pad = w.cout_pad;
            e.cin = w.cin_s; e.rows = rows; e.ci_off = ci_off; e.ci_cnt = src.C; e.col_tiles = (cols + 63) / 64; e.row_tiles = rows / 64;
            const int nb = e.col_tiles * e.row_tiles * taps;
            for (int b = 0; b < nb; ++b) wt_map.push_back(make_int2((int)wt_descs.size(), b));
            wt_descs.push_back(e);
        }
        ConvW syn; syn.has = true; syn.k = k; syn.cout = src.C; syn.cout_pad = rup(src.C, 64); syn.cin_s = dy.C;
        ConvArgs d; d.xa = dy; d.w = &syn; d.w_over = ws_ref(wt_off); d.no_bias = true; d.k = k; d.stride = 1; d.pad = k - 1 - pad;
        if (stride == 2) { d.ups = 1; d.exact = 1; }
        d.Do = src.D << ups; d.Ho = src.H << ups; d.Wo = src.W << ups;
        d.want_stats = false;
        if (!ups) d.residual = take_grad(src);
        Act g = conv(d, "dgrad");
        if (!g.valid) return false;
        if (ups) {                                      // adjoint of the fused nearest x2 upsample
            Act gc = new_act(src.N, src.D, src.H, src.W, src.C);
            Op sp{}; sp.kind = OP_SUMPOOL; sp.r[0] = ws_ref(g.off); sp.r[1] = ws_ref(gc.off);
            sp.i[0] = src.N; sp.i[1] = src.D; sp.i[2] = src.H; sp.i[3] = src.W; sp.i[4] = src.C; sp.i[5] = hp ? 1 : 0;
            plan->ops.push_back(sp);
            add_grad_alias(src, gc);
        } else gslot[src.off] = g;
        return true;
    }
    bool backward_conv(const Tape& t, const Act& dout) {
        const ConvArgs& a = t.c; const ConvW& w = *a.w;
        int cin_real = 0;
        for (const ParamDesc& d : m->params) if (d.kind == PK_CONV_W && d.dst_off == w.w_off) cin_real = d.cin;
        for (const ParamDesc& d : m->params) {               // every parameter of the slot(s) this conv differentiates
            const bool mine = (d.kind == PK_CONV_W && (d.dst_off == w.w_off || (a.w1 && d.dst_off == a.w1->w_off))) ||
                              (d.kind == PK_VEC_F32 && ((d.dst_off >= w.b_off && d.dst_off < w.b_off + (size_t)w.cout_pad * 4) ||
                                                        (a.w1 && d.dst_off >= a.w1->b_off && d.dst_off < a.w1->b_off + (size_t)a.w1->cout_pad * 4)));
            if (mine) note_done(d.flat_off);
        }
        if (!cin_real) { err = "backward: conv slot without parameters"; return false; }
        if (dout.C != rup(w.cout, 32)) { err = "backward: gradient channel mismatch"; return false; }
        // bias (both biases of a conv with a fused 1x1 skip see the same column sums) and the time-embedding rows
        vec_off = pool.alloc((size_t)dout.C * 4);
        emit_colsum(dout, false, ws_ref(vec_off), dout.C, 0);
        export_bias(w);
        if (a.w1) export_bias(*a.w1);
        if (a.temb_row >= 0) emit_colsum(dout, true, ws_ref(dtemb_off + (size_t)a.temb_row * 4), w.cout, tproj_stride);
        // weights
        cur_ksplit = wgrad_ksplit(dout.rows(), a.k * a.k * a.k, w.cout, cin_real, hp, a.stride, a.ups);
        cur_rows_total = w.cout;
        dw_off = pool.alloc((size_t)cur_ksplit * a.k * a.k * a.k * w.cout * cin_real * 4);
        if (a.xb.valid) {
            emit_wgrad(dout, a.xa, w.cout, a.xa.C, cin_real, 0, a.k, a.stride, a.pad, a.ups);
            emit_wgrad(dout, a.xb, w.cout, a.xb.C, cin_real, a.xa.C, a.k, a.stride, a.pad, a.ups);
        } else emit_wgrad(dout, a.xa, w.cout, cin_real, cin_real, 0, a.k, a.stride, a.pad, a.ups);
        export_conv_weights(w, cin_real);
        if (a.w1) {
            const int c1 = a.g1a.C + (a.g1b.valid ? a.g1b.C : 0);
            cur_ksplit = wgrad_ksplit(dout.rows(), 1, a.w1->cout, c1, hp); cur_rows_total = a.w1->cout;
            dw_off = pool.alloc((size_t)cur_ksplit * a.w1->cout * c1 * 4);
            emit_wgrad(dout, a.g1a, a.w1->cout, a.g1a.C, c1, 0, 1, 1, 0, 0);
            if (a.g1b.valid) emit_wgrad(dout, a.g1b, a.w1->cout, a.g1b.C, c1, a.g1a.C, 1, 1, 0, 0);
            export_conv_weights(*a.w1, c1);
        }
        // data
        if (!t.leaf_input) {
            if (!emit_dgrad(dout, a.xa, w, 0, a.k, a.stride, a.pad, a.ups)) return false;
            if (a.xb.valid && !emit_dgrad(dout, a.xb, w, a.xa.C, a.k, a.stride, a.pad, a.ups)) return false;
        }
        if (a.w1) {
            if (!emit_dgrad(dout, a.g1a, *a.w1, 0, 1, 1, 0, 0)) return false;
            if (a.g1b.valid && !emit_dgrad(dout, a.g1b, *a.w1, a.g1a.C, 1, 1, 0, 0)) return false;
        }
        if (a.residual.valid) add_grad_alias(a.residual, dout);
        return true;
    }
    bool backward_gn(const Tape& t) {
        Act dy = take_grad(t.out);
        if (!dy.valid) { err = "backward: GroupNorm output without a gradient"; return false; }
        const Act& xa = t.xa; const Act& xb = t.xb;
        const int C = t.g->C, DHW = xa.D * xa.H * xa.W, N = xa.N;
        const int cvec = C / 8, rows_par = std::max(1, 256 / cvec);
        int nslab = std::min((DHW + rows_par - 1) / rows_par, std::max(1, 512 / N));
        const int rps = (DHW + nslab - 1) / nslab; nslab = (DHW + rps - 1) / rps;
        gnpart_bytes = std::max(gnpart_bytes, (size_t)N * nslab * C * 2 * 4);
        const size_t gsum = pool.alloc((size_t)N * t.groups * 2 * 4), dgn = pool.alloc((size_t)N * C * 4), dbn = pool.alloc((size_t)N * C * 4);
        Act acc_a = take_grad(xa), acc_b = xb.valid ? take_grad(xb) : Act();
        Act dxa = new_act(xa.N, xa.D, xa.H, xa.W, xa.C), dxb;
        if (xb.valid) dxb = new_act(xb.N, xb.D, xb.H, xb.W, xb.C);
        int64_t go = -1, bo = -1;
        for (const ParamDesc& d : m->params) {
            if (d.kind == PK_VEC_F32 && d.dst_off == t.g->g_off) go = d.flat_off;
            if (d.kind == PK_VEC_F32 && d.dst_off == t.g->b_off) bo = d.flat_off;
        }
        if (go < 0 || bo < 0) { err = "backward: GroupNorm parameters not found"; return false; }
        note_done(std::min(go, bo));
        Op o{}; o.kind = OP_GNB;
        o.r[0] = ws_ref(dy.off); o.r[1] = ws_ref(xa.off); o.r[2] = xb.valid ? ws_ref(xb.off) : Ref(); o.r[3] = ws_ref(t.ab_off);
        o.r[5] = ws_ref(t.mr_off); o.r[6] = w_ref(t.g->g_off); o.r[7] = ws_ref(gsum); o.r[8] = ws_ref(dgn); o.r[9] = ws_ref(dbn);
        // passes 2 + 3 in one launch (gn_bwd_fold_apply_kernel) where the forward's one-launch form applies too; its blocks also leave the
        // column sums of dx per row chunk, which are the bias / time-embedding gradients of the convs that produced xa / xb (emit_colsum)
        o.i[11] = 0; o.i[12] = 0;
        if (!hp && gnb_fold_enabled() && C / t.groups <= 64 && nslab <= 512) {
            const int slices = (C + 63) / 64;
            int chunks = std::max(1, std::min(256 / (slices * N), (DHW + 31) / 32));
            const int rpb = rup((DHW + chunks - 1) / chunks, 32);
            chunks = (DHW + rpb - 1) / rpb;
            o.i[11] = rpb; o.i[12] = chunks;
            if (colsum_batched()) {
                const size_t cs = pool.alloc((size_t)N * chunks * C * 2 * 4);       // kept to the end of the plan, like every batched column-sum partial
                o.r[7] = ws_ref(cs);                                               // (the group-sum scratch is not used by this form)
                dxa.cs_off = cs; dxa.cs_rows = chunks;
                if (xb.valid) { dxb.cs_off = cs + (size_t)N * chunks * xa.C * 2 * 4; dxb.cs_rows = chunks; }
                o.i[13] = 1;
            }
        }
        o.r[10] = acc_a.valid ? ws_ref(acc_a.off) : Ref(); o.r[11] = acc_b.valid ? ws_ref(acc_b.off) : Ref();
        o.r[12] = ws_ref(dxa.off); o.r[13] = xb.valid ? ws_ref(dxb.off) : Ref();
        int* i = o.i;
        i[0] = xa.C; i[1] = xb.valid ? xb.C : 0; i[2] = t.groups; i[3] = DHW; i[4] = N; i[5] = t.silu ? 1 : 0; i[6] = nslab; i[7] = rps;
        i[8] = (int)go; i[9] = (int)bo; i[10] = hp ? 1 : 0;
        gnpart_fixups.push_back(plan->ops.size()); plan->ops.push_back(o);
        gslot[xa.off] = dxa;
        if (xb.valid) gslot[xb.off] = dxb;
        return true;
    }
    bool backward_attn(const Tape& t) {
        Act d_o = take_grad(t.o);
        if (!d_o.valid) { err = "backward: attention output without a gradient"; return false; }
        const Act& q = t.qkv;
        const int C = t.o.C, N = q.D * q.H * q.W;
        Act dqkv = new_act(q.N, q.D, q.H, q.W, q.C);
        const size_t delta = pool.alloc((size_t)q.N * (C / t.head_ch) * N * 4);
        Op o{}; o.kind = OP_ATTN_BWD;
        o.r[0] = ws_ref(q.off); o.r[1] = ws_ref(t.o.off); o.r[2] = ws_ref(d_o.off); o.r[3] = ws_ref(t.lse_off); o.r[4] = ws_ref(delta);
        o.r[5] = ws_ref(dqkv.off);
        o.i[0] = q.N; o.i[1] = N; o.i[2] = C; o.i[3] = t.head_ch; o.i[4] = hp ? 1 : 0; o.f[0] = 1.0f / sqrtf((float)t.head_ch);
        plan->ops.push_back(o);
        gslot[q.off] = dqkv;
        return true;
    }
    void emit_lin_dw(Ref dy, Ref x_pre, Ref dW, Ref db, int B, int I, int O, int dy_stride, int x_stride, int silu) {
        Op o{}; o.kind = OP_LIN_DW; o.r[0] = dy; o.r[1] = x_pre; o.r[2] = dW; o.r[3] = db;
        o.i[0] = B; o.i[1] = I; o.i[2] = O; o.i[3] = dy_stride; o.i[4] = x_stride; o.i[5] = silu; o.i[6] = hp ? 1 : 0;
        plan->ops.push_back(o);
    }
    void emit_lin_dx(Ref W, Ref dy, Ref x_pre, Ref dx, int B, int I, int O, int dy_stride, int x_stride, int silu) {
        const int nz = std::max(1, std::min(64, O / 64));
        const size_t part = pool.alloc((size_t)nz * B * I * 4);
        Op o{}; o.kind = OP_LIN_DX; o.r[0] = W; o.r[1] = dy; o.r[2] = x_pre; o.r[3] = dx; o.r[4] = ws_ref(part);
        o.i[0] = B; o.i[1] = I; o.i[2] = O; o.i[3] = dy_stride; o.i[4] = x_stride; o.i[5] = silu; o.i[6] = nz; o.i[7] = hp ? 1 : 0;
        plan->ops.push_back(o);
    }
    // AutoencoderKL: gradient of the fused heads conv from dz (decoder side) and the KL-term gradients (I/O 1, 2)
    Act dout_heads, vae_zin; size_t vae_ml_off = 0, vae_z_off = 0; int vae_L = 0;
    bool emit_heads_bwd() {
        Act dz = take_grad(vae_zin);
        if (!dz.valid) { err = "backward: latent without a gradient"; return false; }
        const int cs = rup(2 * vae_L, 32);
        dout_heads = new_act(dz.N, dz.D, dz.H, dz.W, cs);
        Op o{}; o.kind = OP_VAE_HEADS_BWD;
        o.r[0] = ws_ref(dz.off); o.r[1] = ws_ref(vae_ml_off); o.r[2] = ws_ref(vae_z_off); o.r[3] = io_ref(1); o.r[6] = io_ref(2);
        o.r[7] = ws_ref(dout_heads.off);
        o.i[0] = dz.N; o.i[1] = vae_L; o.i[2] = cs; o.i[3] = dz.D * dz.H * dz.W; o.i[4] = dz.C; o.i[5] = hp ? 1 : 0;
        plan->ops.push_back(o);
        return true;
    }
    // Walk the tape backwards.  `dout_final` = the packed gradient of the network output (the last conv writes fp32
    // NCDHW straight to the caller, so its gradient arrives through the I/O table).
    int64_t entry_param_end(const Tape& t) const {       // one past the last flat offset of the parameters this tape entry differentiates
        int64_t end = 0;
        auto upd = [&](const ParamDesc& d) { int64_t n = 1; for (auto v : d.shape) n *= v; end = std::max(end, d.flat_off + n); };
        if (t.kind == 0) {
            const ConvW& w = *t.c.w; const ConvW* w1 = t.c.w1;
            for (const ParamDesc& d : m->params) {
                const bool mine = (d.kind == PK_CONV_W && (d.dst_off == w.w_off || (w1 && d.dst_off == w1->w_off))) ||
                                  (d.kind == PK_VEC_F32 && ((d.dst_off >= w.b_off && d.dst_off < w.b_off + (size_t)w.cout_pad * 4) ||
                                                            (w1 && d.dst_off >= w1->b_off && d.dst_off < w1->b_off + (size_t)w1->cout_pad * 4)));
                if (mine) upd(d);
            }
        } else if (t.kind == 1) {
            for (const ParamDesc& d : m->params) if (d.kind == PK_VEC_F32 && (d.dst_off == t.g->g_off || d.dst_off == t.g->b_off)) upd(d);
        }
        return end;
    }
    bool backward_all(const Act& dout_final) {
        // pending_end[k] = the highest parameter end among tape entries BEFORE k: once entry k has been differentiated, the flat
        // gradient range [pending_end[k], total) is final whatever order the entries were recorded in
        std::vector<int64_t> pending_end(tape.size() + 1, 0);
        for (size_t k = 0; k < tape.size(); ++k) pending_end[k + 1] = std::max(pending_end[k], entry_param_end(tape[k]));
        for (size_t k = tape.size(); k-- > 0;) {
            const Tape& t = tape[k];
            bool ok = true;
            if (t.kind == 0) {
                if (t.c.f32_out && t.c.f32_tag == 1 && !emit_heads_bwd()) return false;
                Act dout = t.c.f32_out ? (t.c.f32_tag == 1 ? dout_heads : dout_final) : take_grad(t.out);
                if (!dout.valid) { err = "backward: conv output without a gradient"; return false; }
                ok = backward_conv(t, dout);
            } else if (t.kind == 1) ok = backward_gn(t);
            else ok = backward_attn(t);
            if (!ok) return false;
            done_from = std::min(done_from, std::max(pending_end[k], tail_reserved));
            close_bucket(false);
        }
        return true;
    }

    size_t temb_all_off = 0; int tproj_stride = 0;

    void finish() {
        // shared scratch lives ABOVE the activation pool's high-water mark: it is used throughout the plan, so it
        // must never alias a (temporarily freed) activation block
        size_t top = rup_sz(pool.high, 256);
        partial_off = top; top += rup_sz(partial_bytes, 256);
        gnpart_off = top; top += rup_sz(gnpart_bytes, 256);
        gnab_off = top; top += rup_sz(gnab_bytes, 256);
        pool.high = top;
        for (size_t k : partial_fixups) plan->ops[k].r[11] = ws_ref(partial_off);
        for (size_t k : gnpart_fixups) plan->ops[k].r[4] = ws_ref(gnpart_off);
        for (size_t k : gnab_fixups) plan->ops[k].r[5] = ws_ref(gnab_off);
        plan->ws_bytes = pool.high + 256;
    }
};

// ================================================================================================ UNet
static int unet_register(ldm_model* m) {
    const ldm_unet_cfg& c = m->ucfg;
    const int L = c.num_levels; const int* ch = c.channels;
    const int temb = ch[0] * 4;
    for (int i = 0; i < L; ++i) if (ch[i] % 32 || ch[i] % c.norm_num_groups)
        return fail(LDM_ERR_UNSUPPORTED, "UNet channels must be multiples of 32 and of norm_num_groups (level %d: %d)", i, ch[i]);
    // Parameter (= flat gradient buffer) order follows the order in which the BACKWARD pass finishes gradients, reversed: the time
    // embedding MLP and every ResBlock's time_emb_proj come first (their gradients are the last ones produced), then conv_in, the
    // down blocks, the middle block, the up blocks and out.  The flat gradient buffer therefore fills strictly from its end to its
    // front, and the data-parallel exchange can reduce completed tail ranges while backward is still running (ldm_model_set_grad_sync).
    std::vector<std::pair<std::string, int>> tprojs;        // (resblock prefix, cout) in execution order
    bool collect = true;                                     // first pass: only list the ResBlocks (their projections are registered up front)
    auto reg_res = [&](const std::string& p, int cin, int cout) {
        if (collect) { tprojs.push_back({p, cout}); return; }
        m->reg_gn(p + ".norm1", cin);
        m->reg_conv(p + ".conv1", cin, cin, cout, 3);
        m->reg_gn(p + ".norm2", cout);
        m->reg_conv(p + ".conv2", cout, cout, cout, 3);
        if (cin != cout) m->reg_conv(p + ".skip_connection", cin, cin, cout, 1);
    };
    auto walk = [&]() {
    if (!collect) { m->reg_conv("conv_in", c.in_channels, rup(c.in_channels, 32), ch[0], 3); m->reg_im2col("conv_in", c.in_channels); }
    int oc = ch[0];
    for (int i = 0; i < L; ++i) {
        int ic = oc; oc = ch[i];
        for (int j = 0; j < c.num_res_blocks[i]; ++j) {
            char p[96]; snprintf(p, sizeof p, "down_blocks.%d.resnets.%d", i, j);
            reg_res(p, j == 0 ? ic : oc, oc);
            if (c.attention_levels[i] && !collect) { snprintf(p, sizeof p, "down_blocks.%d.attentions.%d", i, j); m->reg_attn(p, oc); }
        }
        if (i != L - 1 && !collect) { char p[96]; snprintf(p, sizeof p, "down_blocks.%d.downsampler.op", i); m->reg_conv(p, oc, oc, oc, 3); }
    }
    reg_res("middle_block.resnet_1", ch[L - 1], ch[L - 1]);
    if (!collect) m->reg_attn("middle_block.attention", ch[L - 1]);
    reg_res("middle_block.resnet_2", ch[L - 1], ch[L - 1]);
    oc = ch[L - 1];
    for (int i = 0; i < L; ++i) {
        const int lvl = L - 1 - i;
        int prev = oc; oc = ch[lvl];
        const int ic = ch[std::max(lvl - 1, 0)];
        const int nres = c.num_res_blocks[lvl] + 1;
        for (int j = 0; j < nres; ++j) {
            const int skip_c = (j == nres - 1) ? ic : oc;
            const int rin = (j == 0) ? prev : oc;
            char p[96]; snprintf(p, sizeof p, "up_blocks.%d.resnets.%d", i, j);
            reg_res(p, rin + skip_c, oc);
            if (c.attention_levels[lvl] && !collect) { snprintf(p, sizeof p, "up_blocks.%d.attentions.%d", i, j); m->reg_attn(p, oc); }
        }
        if (i != L - 1 && !collect) { char p[96]; snprintf(p, sizeof p, "up_blocks.%d.upsampler.postconv", i); m->reg_conv(p, oc, oc, oc, 3, true); }
    }
    if (!collect) { m->reg_gn("out.0", ch[0]); m->reg_conv("out.2", ch[0], ch[0], c.out_channels, 3); }
    };
    walk();                                                  // pass 1: the ResBlock list
    m->reg_linear("time_embed.0", ch[0], temb);
    m->reg_linear("time_embed.2", temb, temb);
    // stacked time_emb_proj: one GEMV for every ResBlock ([sum cout][temb] bf16)
    int rows = 0; for (auto& t : tprojs) rows += t.second;
    m->tproj_rows = rows;
    m->tproj_w_off = m->arena_alloc((size_t)rows * temb * 2);
    m->tproj_b_off = m->arena_alloc((size_t)rows * 4);
    int r = 0;
    for (auto& t : tprojs) {
        m->tproj_row[t.first] = r;
        m->reg_linear_at(t.first + ".time_emb_proj", temb, t.second, m->tproj_w_off + (size_t)r * temb * 2, m->tproj_b_off + (size_t)r * 4);
        r += t.second;
    }
    collect = false;
    walk();                                                  // pass 2: everything else, in execution order
    return 0;
}

// temb_table: the plan of ldm_unet_denoise_step when the model holds the tabulated projections of the sampler's schedule: one row copy
// (OP_TEMB_ROW) instead of sinusoid + three GEMVs; everything else (workspace layout included) is the plain forward plan
static int unet_build(ldm_model* m, int B, int D, int H, int W, Plan* plan, bool train, bool hp = false, int tap_mode = 0, bool temb_table = false) {
    const ldm_unet_cfg& c = m->ucfg;
    const int L = c.num_levels; const int* ch = c.channels;
    const int temb = ch[0] * 4, G = c.norm_num_groups; const float eps = c.norm_eps;
    if (train && tap_mode) return fail(LDM_ERR_UNSUPPORTED, "debug taps are inference-only");
    Builder b; b.m = m; b.plan = plan; b.train = train; b.recording = train; b.hp = hp; b.tap_mode = tap_mode;
    // ---- time embedding: sinusoid -> Linear -> SiLU -> Linear -> (SiLU -> stacked projections)
    const size_t sin_off = b.pool.alloc((size_t)B * ch[0] * 4);
    const size_t e1_off = b.pool.alloc((size_t)B * temb * 4);
    const size_t e2_off = b.pool.alloc((size_t)B * temb * 4);
    b.tproj_stride = m->tproj_rows;
    b.temb_all_off = b.pool.alloc(((size_t)B * m->tproj_rows + 256) * 4);
    const LinW& l0 = m->lins.at("time_embed.0"); const LinW& l2 = m->lins.at("time_embed.2");
    if (temb_table && !train) {
        Op o{}; o.kind = OP_TEMB_ROW; o.r[0].base = BASE_TTAB; o.r[1].base = BASE_SST; o.r[2] = ws_ref(b.temb_all_off);
        o.i[0] = m->tproj_rows; o.i[1] = B; plan->ops.push_back(o);
    } else {
    { Op o{}; o.kind = OP_SINUSOID; o.r[0] = io_ref(2); o.r[1] = ws_ref(sin_off); o.i[0] = B; o.i[1] = ch[0];
      plan->ops.push_back(o); }
    auto gemv = [&](size_t w_off, size_t b_off, size_t x_off, size_t y_off, int I, int O, int xs, int ys, int silu) {
        Op o{}; o.kind = hp ? OP_GEMV32 : OP_GEMV; o.r[0] = hp ? w32_ref(w_off) : w_ref(w_off); o.r[1] = w_ref(b_off); o.r[2] = ws_ref(x_off); o.r[3] = ws_ref(y_off);
        o.i[0] = I; o.i[1] = O; o.i[2] = xs; o.i[3] = ys; o.i[4] = silu; o.i[5] = B; plan->ops.push_back(o);
    };
    gemv(l0.w_off, l0.b_off, sin_off, e1_off, ch[0], temb, ch[0], temb, 0);
    gemv(l2.w_off, l2.b_off, e1_off, e2_off, temb, temb, temb, temb, 1);
    gemv(m->tproj_w_off, m->tproj_b_off, e2_off, b.temb_all_off, temb, m->tproj_rows, temb, m->tproj_rows, 1);
    }

    // ---- pack input (x | cond) -> NDHWC bf16, channels padded to 32
    const int cin_s = rup(c.in_channels, 32);
    Act h = b.conv_in_im2col("conv_in", io_ref(0), io_ref(1), B, D, H, W, c.in_channels, false);
    if (!h.valid) {
        if (!b.err.empty()) return fail(LDM_ERR_UNSUPPORTED, "%s", b.err.c_str());
        Act xin = b.new_act(B, D, H, W, cin_s);
        { Op o{}; o.kind = hp ? OP_PACK32 : OP_PACK; o.r[0] = io_ref(0); o.r[1] = io_ref(1); o.r[2] = ws_ref(xin.off);
          o.i[0] = B; o.i[1] = c.in_channels; o.i[2] = cin_s; o.i[3] = D * H * W; plan->ops.push_back(o); }
        h = b.conv3("conv_in", xin);
        if (train && !b.tape.empty()) b.tape.back().leaf_input = true;        // no gradient w.r.t. the network input
        b.free_act(xin);
    }
    if (!h.valid) return fail(LDM_ERR_UNSUPPORTED, "%s", b.err.c_str());
    b.tap("conv_in", h, ch[0]);
    std::vector<Act> skips; skips.push_back(h);
    char p[96];
    for (int i = 0; i < L; ++i) {
        for (int j = 0; j < c.num_res_blocks[i]; ++j) {
            snprintf(p, sizeof p, "down_blocks.%d.resnets.%d", i, j);
            char pa[96]; snprintf(pa, sizeof pa, "down_blocks.%d.attentions.%d.norm", i, j);
            Act hn = b.resblock(p, h, Act(), ch[i], G, eps, ".skip_connection", true, c.attention_levels[i] ? &m->gns.at(pa) : nullptr);
            if (!hn.valid) return fail(LDM_ERR_UNSUPPORTED, "%s", b.err.c_str());
            b.tap(p, hn, ch[i]);
            if (c.attention_levels[i]) {
                snprintf(p, sizeof p, "down_blocks.%d.attentions.%d", i, j);
                Act ha = b.attention(p, hn, c.num_head_channels[i], G, eps);
                b.free_act(hn);
                if (!ha.valid) return fail(LDM_ERR_UNSUPPORTED, "%s", b.err.c_str());
                hn = ha;
                b.tap(p, hn, ch[i]);
            }
            skips.push_back(hn); h = hn;
        }
        if (i != L - 1) {
            if ((h.D | h.H | h.W) & 1) return fail(LDM_ERR_UNSUPPORTED, "odd spatial size %dx%dx%d at a UNet downsample", h.D, h.H, h.W);
            snprintf(p, sizeof p, "down_blocks.%d.downsampler.op", i);
            Act hd = b.conv3(p, h, 2, 1);
            if (!hd.valid) return fail(LDM_ERR_UNSUPPORTED, "%s", b.err.c_str());
            b.tap(p, hd, ch[i]);
            skips.push_back(hd); h = hd;
        }
    }
    {   // middle: Res, Attn, Res.  `h` is also the last skip and stays alive.
        Act h1 = b.resblock("middle_block.resnet_1", h, Act(), ch[L - 1], G, eps, ".skip_connection", true, &m->gns.at("middle_block.attention.norm"));
        if (!h1.valid) return fail(LDM_ERR_UNSUPPORTED, "%s", b.err.c_str());
        b.tap("middle_block.resnet_1", h1, ch[L - 1]);
        Act h2 = b.attention("middle_block.attention", h1, c.num_head_channels[L - 1], G, eps);
        b.free_act(h1);
        if (!h2.valid) return fail(LDM_ERR_UNSUPPORTED, "%s", b.err.c_str());
        b.tap("middle_block.attention", h2, ch[L - 1]);
        Act h3 = b.resblock("middle_block.resnet_2", h2, Act(), ch[L - 1], G, eps, ".skip_connection", true);
        b.free_act(h2);
        if (!h3.valid) return fail(LDM_ERR_UNSUPPORTED, "%s", b.err.c_str());
        b.tap("middle_block.resnet_2", h3, ch[L - 1]);
        h = h3;
    }
    for (int i = 0; i < L; ++i) {
        const int lvl = L - 1 - i;
        for (int j = 0; j < c.num_res_blocks[lvl] + 1; ++j) {
            Act s = skips.back(); skips.pop_back();
            snprintf(p, sizeof p, "up_blocks.%d.resnets.%d", i, j);
            char pa[96]; snprintf(pa, sizeof pa, "up_blocks.%d.attentions.%d.norm", i, j);
            Act hn = b.resblock(p, h, s, ch[lvl], G, eps, ".skip_connection", true, c.attention_levels[lvl] ? &m->gns.at(pa) : nullptr);
            b.free_act(h); b.free_act(s);
            if (!hn.valid) return fail(LDM_ERR_UNSUPPORTED, "%s", b.err.c_str());
            b.tap(p, hn, ch[lvl]);
            if (c.attention_levels[lvl]) {
                snprintf(p, sizeof p, "up_blocks.%d.attentions.%d", i, j);
                Act ha = b.attention(p, hn, c.num_head_channels[lvl], G, eps);
                b.free_act(hn);
                if (!ha.valid) return fail(LDM_ERR_UNSUPPORTED, "%s", b.err.c_str());
                hn = ha;
                b.tap(p, hn, ch[lvl]);
            }
            h = hn;
        }
        if (i != L - 1) {
            snprintf(p, sizeof p, "up_blocks.%d.upsampler.postconv", i);
            Act hu = b.conv3(p, h, 1, 1, 1);
            b.free_act(h);
            if (!hu.valid) return fail(LDM_ERR_UNSUPPORTED, "%s", b.err.c_str());
            b.tap(p, hu, ch[lvl]);
            h = hu;
        }
    }
    Act hn = b.gn_apply(m->gns.at("out.0"), h, Act(), G, eps, true, &m->convs.at("out.2"));
    b.free_act(h);
    if (!hn.valid) return fail(LDM_ERR_UNSUPPORTED, "%s", b.err.c_str());
    Builder::ConvArgs oa; oa.xa = hn; oa.w = &m->convs.at("out.2"); oa.Do = D; oa.Ho = H; oa.Wo = W;
    oa.f32_out = true; oa.out_ref = io_ref(3); oa.cout_real = c.out_channels;
    b.conv(oa, "out.2");
    if (!b.err.empty()) return fail(LDM_ERR_UNSUPPORTED, "%s", b.err.c_str());
    if (train) {
        // ================= backward: reverse walk over the tape, then the time-embedding MLP =================
        plan->train = true; plan->bwd_begin = plan->ops.size();
        b.recording = false;
        const int rows = m->tproj_rows;
        b.dtemb_off = b.pool.alloc(((size_t)B * rows + 256) * 4);
        { Op o{}; o.kind = OP_WT_BATCH; o.i[0] = hp ? 1 : 0; plan->ops.push_back(o); }          // every flipped / transposed weight matrix, one launch
        const int cos_ = rup(c.out_channels, 32);
        Act dout = b.new_act(B, D, H, W, cos_);
        { Op o{}; o.kind = hp ? OP_PACK32 : OP_PACK; o.r[0] = io_ref(0); o.r[1] = Ref(); o.r[2] = ws_ref(dout.off);
          o.i[0] = B; o.i[1] = c.out_channels; o.i[2] = cos_; o.i[3] = D * H * W; plan->ops.push_back(o); }
        b.bucket_hi = b.done_from = m->flat_total;
        b.tail_reserved = m->params[m->pindex.at("conv_in.conv.weight")].flat_off;     // the time-embedding parameters in front of it come last
        if (!b.backward_all(dout)) return fail(LDM_ERR_UNSUPPORTED, "%s", b.err.c_str());
        b.flush_colsums();                              // bias gradients and the time-embedding rows the MLP backward reads next
        // stacked projections  temb_all = Wt silu(e2) + bt
        const Ref dtemb = ws_ref(b.dtemb_off);
        b.dw_off = b.pool.alloc((size_t)rows * temb * 4); b.vec_off = b.pool.alloc((size_t)rows * 4);
        b.emit_lin_dw(dtemb, ws_ref(e2_off), ws_ref(b.dw_off), ws_ref(b.vec_off), B, temb, rows, rows, temb, 1);
        for (const ParamDesc& d : m->params) {
            if (d.kind == PK_LINEAR_W && d.dst_off >= m->tproj_w_off && d.dst_off < m->tproj_w_off + (size_t)rows * temb * 2)
                b.emit_export(b.dw_off, 1, rows, temb, (int)((d.dst_off - m->tproj_w_off) / ((size_t)temb * 2)), 0, d.cout, d.cin, d.flat_off);
            if (d.kind == PK_VEC_F32 && d.dst_off >= m->tproj_b_off && d.dst_off < m->tproj_b_off + (size_t)rows * 4)
                b.emit_export(b.vec_off, 1, 1, 0, 0, (int)((d.dst_off - m->tproj_b_off) / 4), 1, d.cout, d.flat_off);
        }
        const size_t de2 = b.pool.alloc((size_t)B * temb * 4), de1 = b.pool.alloc((size_t)B * temb * 4);
        b.emit_lin_dx(hp ? w32_ref(m->tproj_w_off) : w_ref(m->tproj_w_off), dtemb, ws_ref(e2_off), ws_ref(de2), B, temb, rows, rows, temb, 1);
        auto P = [&](const char* n) { return m->params[m->pindex.at(n)].flat_off; };
        // time_embed.2: e2 = L2 silu(e1) + b2;  time_embed.0: e1 = L0 sinusoid + b0
        b.emit_lin_dw(ws_ref(de2), ws_ref(e1_off), Builder::grad_ref(P("time_embed.2.weight")), Builder::grad_ref(P("time_embed.2.bias")),
                      B, temb, temb, temb, temb, 1);
        b.emit_lin_dx(hp ? w32_ref(l2.w_off) : w_ref(l2.w_off), ws_ref(de2), ws_ref(e1_off), ws_ref(de1), B, temb, temb, temb, temb, 1);
        b.emit_lin_dw(ws_ref(de1), ws_ref(sin_off), Builder::grad_ref(P("time_embed.0.weight")), Builder::grad_ref(P("time_embed.0.bias")),
                      B, ch[0], temb, temb, ch[0], 0);
        if (!b.err.empty()) return fail(LDM_ERR_UNSUPPORTED, "%s", b.err.c_str());
        b.done_from = 0; b.close_bucket(true);          // the front of the buffer: time embedding MLP, projections, whatever is left
        { Op o{}; o.kind = OP_BUCKET_JOIN; plan->ops.push_back(o); }
        if (plan->wt_tab.upload(b.wt_descs, b.wt_map) || plan->exp_tab.upload(b.exp_descs, b.exp_map) ||
            (!b.cs_descs.empty() && plan->cs_tab.upload(b.cs_descs, b.cs_map)))
            return fail(LDM_ERR_HIP, "descriptor table upload failed");
    }
    b.finish();
    return 0;
}

// ================================================================================================ VAE
struct AeBlock { int kind; int a, b; };                 // 0 conv, 1 res, 2 attn, 3 down, 4 up, 5 gn
static std::vector<AeBlock> ae_encoder_layout(const ldm_vae_cfg& c) {
    std::vector<AeBlock> v; const int L = c.num_levels; const int* ch = c.channels;
    v.push_back({0, c.in_channels, ch[0]});
    int oc = ch[0];
    for (int i = 0; i < L; ++i) {
        int ic = oc; oc = ch[i];
        for (int j = 0; j < c.num_res_blocks[i]; ++j) { v.push_back({1, ic, oc}); ic = oc; if (c.attention_levels[i]) v.push_back({2, ic, ic}); }
        if (i != L - 1) v.push_back({3, ic, ic});
    }
    if (c.with_encoder_nonlocal_attn) { v.push_back({1, ch[L - 1], ch[L - 1]}); v.push_back({2, ch[L - 1], ch[L - 1]}); v.push_back({1, ch[L - 1], ch[L - 1]}); }
    v.push_back({5, ch[L - 1], ch[L - 1]});
    v.push_back({0, ch[L - 1], c.latent_channels});
    return v;
}
static std::vector<AeBlock> ae_decoder_layout(const ldm_vae_cfg& c) {
    std::vector<AeBlock> v; const int L = c.num_levels;
    std::vector<int> rev(c.channels, c.channels + L); std::reverse(rev.begin(), rev.end());
    v.push_back({0, c.latent_channels, rev[0]});
    if (c.with_decoder_nonlocal_attn) { v.push_back({1, rev[0], rev[0]}); v.push_back({2, rev[0], rev[0]}); v.push_back({1, rev[0], rev[0]}); }
    int oc = rev[0];
    for (int i = 0; i < L; ++i) {
        int ic = oc; oc = rev[i];
        const int lvl = L - 1 - i;
        for (int j = 0; j < c.num_res_blocks[lvl]; ++j) { v.push_back({1, ic, oc}); ic = oc; if (c.attention_levels[lvl]) v.push_back({2, ic, ic}); }
        if (i != L - 1) v.push_back({4, ic, ic});
    }
    v.push_back({5, oc, oc});
    v.push_back({0, oc, c.out_channels});
    return v;
}

static int vae_register(ldm_model* m) {
    const ldm_vae_cfg& c = m->vcfg;
    for (int i = 0; i < c.num_levels; ++i) if (c.channels[i] % 32 || c.channels[i] % c.norm_num_groups)
        return fail(LDM_ERR_UNSUPPORTED, "AutoencoderKL channels must be multiples of 32 and of norm_num_groups");
    auto reg = [&](const char* prefix, const std::vector<AeBlock>& lay) -> int {
        for (size_t k = 0; k < lay.size(); ++k) {
            char p[96]; snprintf(p, sizeof p, "%s.blocks.%zu", prefix, k);
            const AeBlock& bl = lay[k];
            switch (bl.kind) {
                case 0: {   // plain Convolution: keys <p>.conv.*
                    ConvW cw = m->new_conv_slot(rup(bl.a, 32), bl.b, 3);
                    m->reg_x3(cw);
                    m->reg_conv_into(p, cw, bl.a, bl.b, 3, 0, false); m->convs[p] = cw;
                    if (k == 0 && std::string(prefix) == "encoder") m->reg_im2col(p, bl.a);       // the encoder's first conv
                    break; }
                case 1:
                    m->reg_gn(std::string(p) + ".norm1", bl.a); m->reg_conv(std::string(p) + ".conv1", bl.a, bl.a, bl.b, 3);
                    m->reg_gn(std::string(p) + ".norm2", bl.b); m->reg_conv(std::string(p) + ".conv2", bl.b, bl.b, bl.b, 3);
                    if (bl.a != bl.b) m->reg_conv(std::string(p) + ".nin_shortcut", bl.a, bl.a, bl.b, 1);
                    break;
                case 2: m->reg_attn(p, bl.a); break;      // SpatialAttentionBlock, single head (d = C)
                case 3: m->reg_conv(std::string(p) + ".conv", bl.a, bl.a, bl.a, 3); break;
                case 4: m->reg_conv(std::string(p) + ".postconv", bl.a, bl.a, bl.a, 3, true); break;
                case 5: m->reg_gn(p, bl.a); break;
            }
        }
        return 0;
    };
    // parameter order = execution order (encoder, heads, post_quant_conv, decoder): the flat gradient buffer fills back to front
    LDM_TRY(reg("encoder", ae_encoder_layout(c)));
    const int Lc = c.latent_channels, Ls = rup(Lc, 32);
    ConvW heads = m->new_conv_slot(Ls, 2 * Lc, 1);                       // mu | log_sigma fused
    m->reg_conv_into("quant_conv_mu", heads, Lc, Lc, 1, 0, false);
    m->reg_conv_into("quant_conv_log_sigma", heads, Lc, Lc, 1, Lc, false);
    m->convs["quant_heads"] = heads;
    ConvW pq = m->new_conv_slot(Ls, Lc, 1);
    m->reg_conv_into("post_quant_conv", pq, Lc, Lc, 1, 0, false);
    m->convs["post_quant_conv"] = pq;
    LDM_TRY(reg("decoder", ae_decoder_layout(c)));
    return 0;
}

static int vae_run_layout(Builder& b, const char* prefix, const std::vector<AeBlock>& lay, Act h, int G, float eps,
                          bool last_f32, int io_out, Act* out, size_t k0 = 0) {
    for (size_t k = k0; k < lay.size(); ++k) {
        char p[96]; snprintf(p, sizeof p, "%s.blocks.%zu", prefix, k);
        const AeBlock& bl = lay[k];
        Act hn;
        if (bl.kind == 0) {
            const bool last = (k == lay.size() - 1);
            if (last && last_f32) {
                Builder::ConvArgs a; a.xa = h; a.w = &b.m->convs.at(p); a.Do = h.D; a.Ho = h.H; a.Wo = h.W;
                a.f32_out = true; a.out_ref = io_ref(io_out); a.cout_real = bl.b;
                b.conv(a, p); b.free_act(h);
                if (!b.err.empty()) return fail(LDM_ERR_UNSUPPORTED, "%s", b.err.c_str());
                *out = Act(); return 0;
            }
            hn = b.conv3(p, h);
        } else if (bl.kind == 1) {
            hn = b.resblock(p, h, Act(), bl.b, G, eps, ".nin_shortcut", false);
        } else if (bl.kind == 2) {
            hn = b.attention(p, h, 0, G, eps);              // head_ch 0 = one head over all channels
        } else if (bl.kind == 3) {
            if ((h.D | h.H | h.W) & 1) return fail(LDM_ERR_UNSUPPORTED, "odd spatial size at an AutoencoderKL downsample");
            hn = b.conv3(std::string(p) + ".conv", h, 2, 0);
        } else if (bl.kind == 4) {
            hn = b.conv3(std::string(p) + ".postconv", h, 1, 1, 1);
        } else if (bl.kind == 5) {
            const ConvW* next3 = nullptr;                    // the network's last conv reads this norm (not in tapped plans: the tap would export the split)
            if (k + 1 < lay.size() && lay[k + 1].kind == 0 && b.tap_mode == 0) {
                char pn[96]; snprintf(pn, sizeof pn, "%s.blocks.%zu", prefix, k + 1);
                next3 = &b.m->convs.at(pn);
            }
            hn = b.gn_apply(b.m->gns.at(p), h, Act(), G, eps, false, next3);
        } else return fail(LDM_ERR_UNSUPPORTED, "unsupported AutoencoderKL block");
        b.free_act(h);
        if (!hn.valid) return fail(LDM_ERR_UNSUPPORTED, "%s", b.err.c_str());
        b.tap(p, hn, bl.b);
        h = hn;
    }
    *out = h;
    return 0;
}

static int vae_build_encode(ldm_model* m, int B, int D, int H, int W, Plan* plan, bool hp = false, int tap_mode = 0) {
    const ldm_vae_cfg& c = m->vcfg;
    Builder b; b.m = m; b.plan = plan; b.hp = hp; b.tap_mode = tap_mode;
    const int cs = rup(c.in_channels, 32);
    Act h;
    Act first = b.conv_in_im2col("encoder.blocks.0", io_ref(0), Ref(), B, D, H, W, c.in_channels, true);
    if (first.valid) {
        b.tap("encoder.blocks.0", first, c.channels[0]);
        LDM_TRY(vae_run_layout(b, "encoder", ae_encoder_layout(c), first, c.norm_num_groups, c.norm_eps, false, 0, &h, 1));
    } else {
        if (!b.err.empty()) return fail(LDM_ERR_UNSUPPORTED, "%s", b.err.c_str());
        Act xin = b.new_act(B, D, H, W, cs);
        { Op o{}; o.kind = hp ? OP_PACK32 : OP_PACK; o.r[0] = io_ref(0); o.r[1] = Ref(); o.r[2] = ws_ref(xin.off);
          o.i[0] = B; o.i[1] = c.in_channels; o.i[2] = cs; o.i[3] = D * H * W; plan->ops.push_back(o); }
        LDM_TRY(vae_run_layout(b, "encoder", ae_encoder_layout(c), xin, c.norm_num_groups, c.norm_eps, false, 0, &h));
    }
    // fused 1x1 heads -> fp32 [B][2L][dhw] scratch, then clamp/exp/sample
    const int dhw = h.D * h.H * h.W;
    const size_t ml_off = b.pool.alloc((size_t)B * 2 * c.latent_channels * dhw * 4);
    Builder::ConvArgs a; a.xa = h; a.w = &m->convs.at("quant_heads"); a.k = 1; a.pad = 0; a.Do = h.D; a.Ho = h.H; a.Wo = h.W;
    a.f32_out = true; a.out_ref = ws_ref(ml_off); a.cout_real = 2 * c.latent_channels;
    b.conv(a, "quant_heads");
    if (!b.err.empty()) return fail(LDM_ERR_UNSUPPORTED, "%s", b.err.c_str());
    b.free_act(h);
    { Op o{}; o.kind = OP_VAE_HEADS; o.r[0] = ws_ref(ml_off); o.r[1] = io_ref(1); o.r[2] = io_ref(2); o.r[3] = io_ref(3); o.r[4] = io_ref(4);
      o.i[0] = B; o.i[1] = c.latent_channels; o.i[2] = dhw; plan->ops.push_back(o); }
    b.finish();
    return 0;
}

static int vae_build_decode(ldm_model* m, int B, int d, int h_, int w, Plan* plan, bool hp = false, int tap_mode = 0) {
    const ldm_vae_cfg& c = m->vcfg;
    Builder b; b.m = m; b.plan = plan; b.hp = hp; b.tap_mode = tap_mode;
    const int ls = rup(c.latent_channels, 32);
    Act zin = b.new_act(B, d, h_, w, ls);
    { Op o{}; o.kind = hp ? OP_PACK32 : OP_PACK; o.r[0] = io_ref(0); o.r[1] = Ref(); o.r[2] = ws_ref(zin.off);
      o.i[0] = B; o.i[1] = c.latent_channels; o.i[2] = ls; o.i[3] = d * h_ * w; plan->ops.push_back(o); }
    Builder::ConvArgs a; a.xa = zin; a.w = &m->convs.at("post_quant_conv"); a.k = 1; a.pad = 0; a.Do = d; a.Ho = h_; a.Wo = w;
    Act z2 = b.conv(a, "post_quant_conv");
    b.free_act(zin);
    if (!z2.valid) return fail(LDM_ERR_UNSUPPORTED, "%s", b.err.c_str());
    b.tap("post_quant_conv", z2, c.latent_channels);
    Act out;
    LDM_TRY(vae_run_layout(b, "decoder", ae_decoder_layout(c), z2, c.norm_num_groups, c.norm_eps, true, 1, &out));
    b.finish();
    return 0;
}

// AutoencoderKL.forward with the tape + backward (stage-1 trainer, 3d_ldm/train_autoencoder.py:366-451):
//   forward  I/O: 0 = x, 1 = eps, 2 = z_mu, 3 = z_sigma, 5 = reconstruction
//   backward I/O: 0 = d recon, 1 = d z_mu (or null), 2 = d z_sigma (or null), 4 = flat parameter gradients
static int vae_build_train(ldm_model* m, int B, int D, int H, int W, Plan* plan, bool hp = false) {
    const ldm_vae_cfg& c = m->vcfg;
    Builder b; b.m = m; b.plan = plan; b.train = true; b.recording = true; b.hp = hp;
    const int cs = rup(c.in_channels, 32), L = c.latent_channels, ls = rup(L, 32);
    Act xin = b.new_act(B, D, H, W, cs);
    { Op o{}; o.kind = hp ? OP_PACK32 : OP_PACK; o.r[0] = io_ref(0); o.r[1] = Ref(); o.r[2] = ws_ref(xin.off);
      o.i[0] = B; o.i[1] = c.in_channels; o.i[2] = cs; o.i[3] = D * H * W; o.i[4] = 1; plan->ops.push_back(o); }
    const size_t t0 = b.tape.size();
    Act h;
    LDM_TRY(vae_run_layout(b, "encoder", ae_encoder_layout(c), xin, c.norm_num_groups, c.norm_eps, false, 0, &h));
    if (b.tape.size() > t0) b.tape[t0].leaf_input = true;                 // no gradient w.r.t. the image
    const int dhw = h.D * h.H * h.W;
    const size_t ml_off = b.pool.alloc((size_t)B * 2 * L * dhw * 4), z_off = b.pool.alloc((size_t)B * L * dhw * 4);
    { Builder::ConvArgs a; a.xa = h; a.w = &m->convs.at("quant_heads"); a.k = 1; a.pad = 0; a.Do = h.D; a.Ho = h.H; a.Wo = h.W;
      a.f32_out = true; a.out_ref = ws_ref(ml_off); a.cout_real = 2 * L; a.f32_tag = 1;
      b.conv(a, "quant_heads"); }
    if (!b.err.empty()) return fail(LDM_ERR_UNSUPPORTED, "%s", b.err.c_str());
    { Op o{}; o.kind = OP_VAE_HEADS; o.r[0] = ws_ref(ml_off); o.r[1] = io_ref(1); o.r[2] = io_ref(2); o.r[3] = io_ref(3); o.r[4] = ws_ref(z_off);
      o.i[0] = B; o.i[1] = L; o.i[2] = dhw; plan->ops.push_back(o); }
    Act zin = b.new_act(B, h.D, h.H, h.W, ls);
    { Op o{}; o.kind = hp ? OP_PACK32 : OP_PACK; o.r[0] = ws_ref(z_off); o.r[1] = Ref(); o.r[2] = ws_ref(zin.off);
      o.i[0] = B; o.i[1] = L; o.i[2] = ls; o.i[3] = dhw; o.i[4] = 1; plan->ops.push_back(o); }
    Builder::ConvArgs pq; pq.xa = zin; pq.w = &m->convs.at("post_quant_conv"); pq.k = 1; pq.pad = 0; pq.Do = h.D; pq.Ho = h.H; pq.Wo = h.W;
    Act z2 = b.conv(pq, "post_quant_conv");
    if (!z2.valid) return fail(LDM_ERR_UNSUPPORTED, "%s", b.err.c_str());
    Act out;
    LDM_TRY(vae_run_layout(b, "decoder", ae_decoder_layout(c), z2, c.norm_num_groups, c.norm_eps, true, 5, &out));
    // ---- backward
    plan->train = true; plan->bwd_begin = plan->ops.size();
    b.recording = false;
    b.vae_zin = zin; b.vae_ml_off = ml_off; b.vae_z_off = z_off; b.vae_L = L;
    { Op o{}; o.kind = OP_WT_BATCH; o.i[0] = hp ? 1 : 0; plan->ops.push_back(o); }
    const int cos_ = rup(c.out_channels, 32);
    Act dout = b.new_act(B, D, H, W, cos_);
    { Op o{}; o.kind = hp ? OP_PACK32 : OP_PACK; o.r[0] = io_ref(0); o.r[1] = Ref(); o.r[2] = ws_ref(dout.off);
      o.i[0] = B; o.i[1] = c.out_channels; o.i[2] = cos_; o.i[3] = D * H * W; o.i[4] = 1; plan->ops.push_back(o); }
    b.bucket_hi = b.done_from = m->flat_total;
    if (!b.backward_all(dout)) return fail(LDM_ERR_UNSUPPORTED, "%s", b.err.c_str());
    b.done_from = 0; b.close_bucket(true);
    { Op o{}; o.kind = OP_BUCKET_JOIN; plan->ops.push_back(o); }
    if (plan->wt_tab.upload(b.wt_descs, b.wt_map) || plan->exp_tab.upload(b.exp_descs, b.exp_map) ||
        (!b.cs_descs.empty() && plan->cs_tab.upload(b.cs_descs, b.cs_map)))
        return fail(LDM_ERR_HIP, "descriptor table upload failed");
    b.finish();
    return 0;
}

// ================================================================================================ launch
struct Bases { char* p[BASE_COUNT]; int sampler_steps; };   // sampler_steps: rows of the table behind BASE_TTAB (OP_TEMB_ROW)
static inline char* rp(const Bases& b, const Ref& r) { return r.base == BASE_NULL ? nullptr : (b.p[r.base] ? b.p[r.base] + r.off : nullptr); }

// LDM_XCD_ROWS (tuning knob, default 0: measured 0.4 % SLOWER over the step, with and without write-through stores: DESIGN.md section 5): unsplit convolutions and the GroupNorm launches deal contiguous ROW ranges to the XCDs
// (ConvParams::tile_order 1, GnFusedParams::xcd_rows), so that a tensor is produced and consumed by the same XCD where the order of
// the work allows it; 0 = the round-2 orders (an XCD streams one weight panel; GroupNorm blocks in launch order)
static int xcd_rows_mode() { static const int v = ldm_xknob("LDM_XCD_ROWS", 0); return v; }
static inline int conv_tile_order(const ConvParams& p) { return (xcd_rows_mode() && p.splitk == 1 && !p.phase_mode) ? 1 : 0; }

template <int WGM, int WGN, int BK>
static int launch_conv_t(const ConvParams& p_in, hipStream_t s) {
    ConvParams p = p_in; p.tile_order = conv_tile_order(p);
    // 128 KiB-class LDS ring; BK = 64 tiles run 8 waves (two per SIMD, intra-workgroup K split)
    constexpr int STAGE = (64 * WGM + 64 * WGN) * BK * 2;
    constexpr int NG = (BK == 64) ? 2 : 1;
    constexpr int NS = (STAGE * 4 <= 131072) ? 4 : 3;
    constexpr int LDS = NS * STAGE + 28 * 64 * WGM * 4;       // ring + (tap, row) -> voxel table (+1 sentinel tap)
    static bool attr_tab[32] = {}; bool& attr_set = attr_flag(attr_tab);   // per device
    if (!attr_set) {
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_igemm_kernel<WGM, WGN, BK, NS, NG>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
        attr_set = true;
    }
    const int grid = p.mtiles * p.ntiles * p.splitk;
    hipLaunchKernelGGL((conv_igemm_kernel<WGM, WGN, BK, NS, NG>), dim3(grid), dim3(256 * NG), LDS, s, p);
    return 0;
}

// ---- optional HIP-event instrumentation of one conv tile configuration (bench.py's roofline leg) ----------
struct ProfState {
    bool on = false; int wgm = 0, wgn = 0, bk = 0, halo = 0;
    std::vector<hipEvent_t> ev; size_t used = 0;          // pairs (start, stop)
    std::vector<double> flops;                            // algorithmic FLOPs of each instrumented launch
    double flops_all = 0.0; long launches_all = 0;        // every conv launch while profiling is on
};
static ProfState g_prof;

static double conv_algorithmic_flops(const ConvParams& p) {
    // 2 * M * Cout_real * (k^3 * Cin0 + Cin1); input channels as stored (only the two stem convs are padded, 4 -> 32)
    const double taps = (double)p.ksize * p.ksize * p.ksize;
    return 2.0 * (double)p.M * (double)p.CoutReal * (taps * (double)(p.c0a + p.c0b) + (double)(p.c1a + p.c1b));
}

static void launch_finalize(const FinalizeParams& f, bool wt, hipStream_t s) {
    const dim3 grid((f.M + 31) / 32, (f.CoutS + 63) / 64);
    if (wt) hipLaunchKernelGGL(splitk_finalize_kernel<true>, grid, dim3(256), 0, s, f);
    else hipLaunchKernelGGL(splitk_finalize_kernel<false>, grid, dim3(256), 0, s, f);
}
static int launch_conv_impl(const ConvParams& p, const ConvCfg& cc, hipStream_t s);
static int launch_conv(const ConvParams& p, const ConvCfg& cc, hipStream_t s) {
    if (!g_prof.on) return launch_conv_impl(p, cc, s);
    g_prof.flops_all += conv_algorithmic_flops(p); g_prof.launches_all++;
    const bool hit = cc.wgm == g_prof.wgm && cc.wgn == g_prof.wgn && cc.bk == g_prof.bk && cc.halo == g_prof.halo &&
                     g_prof.used + 2 <= g_prof.ev.size();
    if (hit) HIP_TRY(hipEventRecord(g_prof.ev[g_prof.used], s));
    LDM_TRY(launch_conv_impl(p, cc, s));
    if (hit) { HIP_TRY(hipEventRecord(g_prof.ev[g_prof.used + 1], s)); g_prof.used += 2; g_prof.flops.push_back(conv_algorithmic_flops(p)); }
    return 0;
}

static int launch_conv_halo(const ConvParams& p_in, hipStream_t s, bool tall = false) {
    ConvParams p = p_in; p.tile_order = conv_tile_order(p);
    fastdiv_make((unsigned)(p.Hout * p.Wout), &p.fd_hw_m, &p.fd_hw_s); fastdiv_make((unsigned)p.Wout, &p.fd_w_m, &p.fd_w_s);
    constexpr int LDS = 6 * 16384 + 3 * 16384 + 9 * 128 * 4;
    constexpr int LDS_TALL = 6 * 8192 + 3 * 32768 + 9 * 256 * 4;
    static bool attr_tab[32] = {}; bool& attr_set = attr_flag(attr_tab);   // per device
    if (!attr_set) {
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3_halo_kernel<6>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3_halo_kernel<6, 0, true>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_TALL));
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3_halo_kernel<6, 0, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3_halo_kernel<6, 0, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_TALL));
        attr_set = true;
    }
    // persistent grid (conv_halo.h): at most one workgroup per CU, each walks its tiles; LDM_HALO_PERSIST=0: one workgroup per tile
    static const int persist = ldm_xknob("LDM_HALO_PERSIST", 1);
    static int cus_tab[32] = {};
    int dev_ = 0; if (hipGetDevice(&dev_) != hipSuccess || dev_ < 0 || dev_ >= 32) dev_ = 0;
    int& cus = cus_tab[dev_];
    if (!cus) { hipDeviceProp_t pr; if (hipGetDeviceProperties(&pr, dev_) == hipSuccess) cus = pr.multiProcessorCount; if (cus < 8) cus = 256; cus -= cus % 8; }
    const int tiles = p.mtiles * p.ntiles * p.splitk;
    const int grid = (persist && tiles > cus) ? cus : tiles;
    const bool loop = grid < tiles;                  // more tiles than workgroups: the persistent instantiation walks them
    static const int mi = ldm_knob("LDM_HALO_MASK_INLINE", 1);    // W-border masks between the MFMAs (conv_halo.h, MASK_INLINE)
    if (mi) {
        static bool mi_attr_tab[32] = {}; bool& mi_attr = attr_flag(mi_attr_tab);
        if (!mi_attr) {
            HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3_halo_kernel<6, 0, false, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
            HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3_halo_kernel<6, 0, true, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_TALL));
            mi_attr = true;
        }
    }
    if (tall) {
        // (the tile-loop instantiations keep the masks in front: inline they push 35 - 53 registers into scratch)
        if (loop) hipLaunchKernelGGL((conv3_halo_kernel<6, 0, true, true>), dim3(grid), dim3(512), LDS_TALL, s, p);
        else if (mi) hipLaunchKernelGGL((conv3_halo_kernel<6, 0, true, false, true>), dim3(grid), dim3(512), LDS_TALL, s, p);
        else hipLaunchKernelGGL((conv3_halo_kernel<6, 0, true>), dim3(grid), dim3(512), LDS_TALL, s, p);
        return 0;
    }
#ifdef LDM_EXPERIMENTS
    // EXPERIMENTS BUILD ONLY (LDM_HALO_RW=1): the 126 x 128 tile with register-fed weights and one barrier per macro step (conv_halo_rw.h).
    // Parity-green on every conv operator test (profiles/r05_halo_rw_ops.txt) and 22 % SLOWER than conv3_halo_kernel at 24^3 (59.1 vs 47.2 us;
    // headline 443.8 vs 484.5 steps/s, profiles/r05_ab_halo_rw.txt): 16 KiB of weight fragments per K step through the vector L1 as half
    // cache lines cost ~350 cycles per step, and ONE barrier per macro step still costs ~250 cycles per step (profiles/r05_halo_ablations.txt).
    // LDM_HALO_PP=1: alternating K steps per wave group, one barrier per six K steps, weights as whole cache lines into registers
    // (conv_halo_pp.h): parity-green, 1133 cycles per K step against 823 -- bound by the vector L1's ingest of the weights
    // (profiles/r05_halo_pp_ablations.txt).  LDM_CONV_DBG bits 4 .. 256 pick its timing ablations.
    static const int pp = ldm_xknob("LDM_HALO_PP", 0);
    if (pp && !p.x3_n && !p.out_f32 && !p.out32 && !p.raw_partial && p.steps1 == 0 && p.CoutPad % 128 == 0 && (p.out || p.splitk > 1)) {
#define PP_CASE(A) if ((p.dbg & 508) == A) { \
            HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3_halo_pp_kernel<A>), hipFuncAttributeMaxDynamicSharedMemorySize, HPP_LDS)); \
            hipLaunchKernelGGL((conv3_halo_pp_kernel<A>), dim3(tiles), dim3(512), HPP_LDS, s, p); return 0; }
        if (p.dbg & 1024) {                                  // producer-wave timing experiment: 12 waves, waves 8 - 11 issue every copy (results are wrong)
            constexpr int LDSP = 100 * 1024 + 32 * 1024;
            if (p.dbg & 32) {
                HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3_halo_pp_kernel<1200>), hipFuncAttributeMaxDynamicSharedMemorySize, LDSP));
                hipLaunchKernelGGL((conv3_halo_pp_kernel<1200>), dim3(tiles), dim3(768), LDSP, s, p); return 0;
            }
            if (p.dbg & 2) {
                HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3_halo_pp_kernel<1170>), hipFuncAttributeMaxDynamicSharedMemorySize, LDSP));
                hipLaunchKernelGGL((conv3_halo_pp_kernel<1170>), dim3(tiles), dim3(768), LDSP, s, p); return 0;
            }
            HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3_halo_pp_kernel<1168>), hipFuncAttributeMaxDynamicSharedMemorySize, LDSP));
            hipLaunchKernelGGL((conv3_halo_pp_kernel<1168>), dim3(tiles), dim3(768), LDSP, s, p); return 0;
        }
        PP_CASE(144) PP_CASE(4) PP_CASE(16) PP_CASE(64) PP_CASE(20) PP_CASE(68) PP_CASE(84) PP_CASE(80) PP_CASE(128) PP_CASE(256) PP_CASE(384) PP_CASE(464) PP_CASE(400) PP_CASE(208)
#undef PP_CASE
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3_halo_pp_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, HPP_LDS));
        hipLaunchKernelGGL((conv3_halo_pp_kernel<0>), dim3(tiles), dim3(512), HPP_LDS, s, p);
        return 0;
    }
    static const int rw = ldm_xknob("LDM_HALO_RW", 0);
    if (rw && !p.x3_n && !p.out_f32 && !p.out32 && !p.raw_partial && p.CoutPad % 128 == 0 && (p.out || p.splitk > 1)) {
        static bool rw_attr_tab[32] = {}; bool& rw_attr = attr_flag(rw_attr_tab);
        if (!rw_attr) {
            HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3_halo_rw_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, HRW_LDS));
            HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3_halo_rw_kernel<64>), hipFuncAttributeMaxDynamicSharedMemorySize, HRW_LDS));
            HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3_halo_rw_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, HRW_LDS));
            rw_attr = true;
        }
        if ((p.dbg & 124) == 64) { hipLaunchKernelGGL((conv3_halo_rw_kernel<64>), dim3(tiles), dim3(512), HRW_LDS, s, p); return 0; }
        if ((p.dbg & 124) == 4) { hipLaunchKernelGGL((conv3_halo_rw_kernel<4>), dim3(tiles), dim3(512), HRW_LDS, s, p); return 0; }
        hipLaunchKernelGGL((conv3_halo_rw_kernel<0>), dim3(tiles), dim3(512), HRW_LDS, s, p);
        return 0;
    }
#endif
    if (p.dbg & (4 | 8 | 16 | 32 | 64 | 128 | 256)) {      // timing ablations (operator-level API + LDM_CONV_DBG only)
#define ABL_CASE(A) if ((p.dbg & 508) == A) { \
            HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3_halo_kernel<6, A>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS)); \
            hipLaunchKernelGGL((conv3_halo_kernel<6, A>), dim3(p.mtiles * p.ntiles * p.splitk), dim3(512), LDS, s, p); return 0; }
        if ((p.dbg & 508) == 384) {                            // barrier with slack 1 (dbg & 2048: slack 2): timing only, the ring protocol is NOT adapted
            constexpr int LDSX = LDS + 64;
            if (p.dbg & 2048) {
                HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3_halo_kernel<6, 385>), hipFuncAttributeMaxDynamicSharedMemorySize, LDSX));
                hipLaunchKernelGGL((conv3_halo_kernel<6, 385>), dim3(p.mtiles * p.ntiles * p.splitk), dim3(512), LDSX, s, p); return 0;
            }
            HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3_halo_kernel<6, 384>), hipFuncAttributeMaxDynamicSharedMemorySize, LDSX));
            hipLaunchKernelGGL((conv3_halo_kernel<6, 384>), dim3(p.mtiles * p.ntiles * p.splitk), dim3(512), LDSX, s, p); return 0;
        }
        ABL_CASE(4) ABL_CASE(8) ABL_CASE(16) ABL_CASE(12) ABL_CASE(20) ABL_CASE(24) ABL_CASE(32) ABL_CASE(40) ABL_CASE(84) ABL_CASE(68) ABL_CASE(64) ABL_CASE(128) ABL_CASE(256)
#undef ABL_CASE
    }
    if (loop) hipLaunchKernelGGL((conv3_halo_kernel<6, 0, false, true>), dim3(grid), dim3(512), LDS, s, p);
    else if (mi) hipLaunchKernelGGL((conv3_halo_kernel<6, 0, false, false, true>), dim3(grid), dim3(512), LDS, s, p);
    else hipLaunchKernelGGL((conv3_halo_kernel<6>), dim3(grid), dim3(512), LDS, s, p);
    return 0;
}

static int launch_conv_impl(const ConvParams& p, const ConvCfg& cc, hipStream_t s) {
    if (cc.halo) return launch_conv_halo(p, s, cc.halo == 2);
#define CASE(M_, N_, K_) if (cc.wgm == M_ && cc.wgn == N_ && cc.bk == K_) return launch_conv_t<M_, N_, K_>(p, s);
    CASE(2, 2, 64) CASE(4, 1, 64) CASE(1, 4, 64) CASE(2, 2, 32) CASE(4, 1, 32) CASE(1, 4, 32)
#undef CASE
    return fail(LDM_ERR_BAD_ARG, "no conv kernel for tile config %dx%d bk %d", cc.wgm, cc.wgn, cc.bk);
}

static inline int grid_for(long total, int per_block = 256, int cap = 4096) {
    long g = (total + per_block - 1) / per_block; if (g > cap) g = cap; if (g < 1) g = 1; return (int)g;
}

// voxel-range split of the weight-gradient GEMM: enough workgroups for 2 waves of 256 CUs, at least 16 K steps each
static int wgrad_pair(int taps, int cout, int cin, int stride, int ups, bool hp) {   // conv_wgrad_kernel's forms with several taps per workgroup (WgradParams::pair): 0 | 1 | 2 (three taps)
    static const int on = ldm_xknob("LDM_WGRAD_PAIR", 3);
    if (!on || hp || taps != 27) return 0;
    const bool y = on >= 2 && cout <= 64 && stride == 1 && ups == 0;
    if (cin > 64) return (y && on >= 3) ? 3 : 0;
    return y ? 2 : 1;
}
// conv_wgrad_kw3_kernel (three kw taps of a (kd, kh) per workgroup, 128 couts x 64 cins, one shared X tile): 3^3, stride 1, both channel
// counts above 64 (below, the pair forms of conv_wgrad_kernel fill the tile better).  launch_wgrad also asks for pad 1 and W >= 8.
// EXPERIMENTS BUILDS ONLY (LDM_WGRAD_KW3=1): parity-green on 11 operator cases and no faster (81.4 vs 83.8 us at 256 -> 256, 24^3): 24 % fewer
// copied bytes per MFMA bought nothing because the step is a serial sum -- copies 500 + fragment reads 500 + masks 340 + barrier 250 +
// MFMAs 900 cycles (profiles/r05_wgrad_kw3.txt), the copies at the ~145 cycles a wave needs to issue one 1 KiB LDS-DMA piece
// (profiles/r05_producer_wave_experiment.txt).
static bool wgrad_kw3(int taps, int cout, int cin, int stride, int ups, bool hp) {
#ifdef LDM_EXPERIMENTS
    static const int on = ldm_xknob("LDM_WGRAD_KW3", 0);
    return on && !hp && taps == 27 && stride == 1 && ups == 0 && cin > 64 && cout > 64;
#else
    return false;
#endif
}
static int wgrad_ksplit(long M, int taps, int cout, int cin, bool hp, int stride, int ups) {
    const int pm = wgrad_pair(taps, cout, cin, stride, ups, hp);
    const bool k3 = wgrad_kw3(taps, cout, cin, stride, ups, hp);
    const long wgs = k3 ? 9L * ((cout + 127) / 128) * ((cin + 63) / 64)
                        : (long)(pm == 3 ? 2 * (taps / 3) : pm == 2 ? taps / 3 : pm == 1 ? (taps + 1) / 2 : taps) * ((cout + 127) / 128) * ((cin + 127) / 128);
    const long steps = (M + 63) / 64;
    // tuning knob: workgroups aimed at (default: one round of 256 CUs; the fp32 kernel, latency bound on its operand loads, wants three per CU: 49.3 -> 46.6 ms per step)
    const long target = ldm_xknob("LDM_WGRAD_WGS", hp ? 768 : 256);
    long k = (target + wgs / 2) / wgs;                       // nearest count of whole rounds
    // at least 16 K steps per workgroup for a 3^3 conv (prologue and ring fill amortised); a 1x1 conv at 12^3 is 27 steps on 4 - 12 workgroups
    // in all (24.7 us of serial K loop, 20 such launches per UNet training step): there 4 steps per workgroup are enough
    const long min_steps = taps == 1 ? 4 : 16;
    if (k > steps / min_steps) k = steps / min_steps;
    // at most 16 copies of a 3^3 weight-sized matrix for the export to fold; a 1x1 conv's matrix is 27 x smaller and its grid is ksplit
    // workgroups in all (nin_shortcut 128 -> 64 over 64^3 voxels: 16 workgroups ran 147 us on 6 % of the chip), so: up to 64 there
    const long cap = taps == 1 ? 64 : pm == 2 ? 32 : 16;     // the three-tap form has 9 workgroups per voxel range (and a 64 x 64 matrix per tap): 28 ranges fill the chip
    if (k > cap) k = cap;
    return (int)(k < 1 ? 1 : k);
}
static int launch_wgrad(const WgradParams& p0, hipStream_t s) {
    WgradParams p = p0;
    const int taps_ = p.ksize * p.ksize * p.ksize;
    p.pair = wgrad_pair(taps_, p.Cout, p.Cin, p.stride, p.ups, false);
    if ((p.pair == 1 || p.pair == 2) && p.cx > 64) p.pair = 0;      // stored channels decide what fits a half row
    if (p.pair == 2 && p.cdy > 64) p.pair = 1;
    if (p.pair == 3 && p.cdy > 64) p.pair = 0;
#ifdef LDM_EXPERIMENTS
    if (p.pair == 0 && wgrad_kw3(taps_, p.Cout, p.Cin, p.stride, p.ups, false) && p.pad == 1 && p.Wout >= 8 && p.Dout == p.Din && p.Hout == p.Hin && p.Wout == p.Win) {
        p.ci_tiles = (p.Cin + 63) / 64;
        static bool attr3_tab[32] = {}; bool& attr3 = attr_flag(attr3_tab);
        if (!attr3) { HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad_kw3_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, WG3_LDS)); attr3 = true; }
        { const int dbg = (int)ldm_xknob("LDM_CONV_DBG", 0);  // timing ablations (results are wrong): 4 no copies, 8 no MFMAs, 16 no fragment reads, 32 no masks, 64 no per-step barrier
#define W3_ABL(A) if (dbg == A) { HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad_kw3_kernel<A>), hipFuncAttributeMaxDynamicSharedMemorySize, WG3_LDS)); \
          hipLaunchKernelGGL(conv_wgrad_kw3_kernel<A>, dim3(p.co_tiles * p.ci_tiles * 9 * p.ksplit), dim3(512), WG3_LDS, s, p); return 0; }
          W3_ABL(4) W3_ABL(8) W3_ABL(16) W3_ABL(32) W3_ABL(64) W3_ABL(20) W3_ABL(52) W3_ABL(116) W3_ABL(48) W3_ABL(112) W3_ABL(96)
#undef W3_ABL
        }
        if (ldm_xknob("LDM_WGRAD_KW3", 0) == 2) {           // sixteen-wave form
            static bool attr3w_tab[32] = {}; bool& attr3w = attr_flag(attr3w_tab);
            if (!attr3w) { HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad_kw3w16_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, WG3_LDS)); attr3w = true; }
            hipLaunchKernelGGL(conv_wgrad_kw3w16_kernel<0>, dim3(p.co_tiles * p.ci_tiles * 9 * p.ksplit), dim3(1024), WG3_LDS, s, p);
            return 0;
        }
        hipLaunchKernelGGL(conv_wgrad_kw3_kernel<0>, dim3(p.co_tiles * p.ci_tiles * 9 * p.ksplit), dim3(512), WG3_LDS, s, p);
        return 0;
    }
#endif
    const int tg_ = p.pair == 3 ? 2 * (taps_ / 3) : p.pair == 2 ? taps_ / 3 : p.pair ? (taps_ + 1) / 2 : taps_;
    constexpr int LDS = 4 * 2 * 64 * 256 + 3 * 256 * 4;        // ring + triple-buffered source-offset table (up to four sections)
    static bool attr_tab[32] = {}; bool& attr_set = attr_flag(attr_tab);   // per device
    if (!attr_set) { HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS)); attr_set = true; }
    { const int dbg = (int)ldm_xknob("LDM_CONV_DBG", 0);      // timing ablations (results are wrong; experiments builds only)
#define W1_ABL(A) if (dbg == A) { HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad_kernel<A>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS)); \
          hipLaunchKernelGGL(conv_wgrad_kernel<A>, dim3(p.co_tiles * p.ci_tiles * tg_ * p.ksplit), dim3(512), LDS, s, p); return 0; }
      W1_ABL(4) W1_ABL(8) W1_ABL(16) W1_ABL(12) W1_ABL(20) W1_ABL(24)
#undef W1_ABL
    }
    static const int w16 = ldm_knob("LDM_WGRAD_W16", 1);     // sixteen waves per workgroup where no several-taps form applies (conv_wgrad_w16.h)
    if (w16 && p.pair == 0) {
        static bool attr16_tab[32] = {}; bool& attr16 = attr_flag(attr16_tab);
        if (!attr16) { HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad_w16_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS)); attr16 = true; }
        hipLaunchKernelGGL(conv_wgrad_w16_kernel<0>, dim3(p.co_tiles * p.ci_tiles * tg_ * p.ksplit), dim3(1024), LDS, s, p);
        return 0;
    }
    hipLaunchKernelGGL(conv_wgrad_kernel<0>, dim3(p.co_tiles * p.ci_tiles * tg_ * p.ksplit), dim3(512), LDS, s, p);
    return 0;
}

static bool wt_stores() { static const int v = ldm_xknob("LDM_WT_STORES", 1); return v != 0; }   // GroupNorm / finalize outputs written through (sc1): -24 us per step

// per-op timeline of every launch plan that runs while it is on (ldm_set_plan_trace; initial state from LDM_PLAN_TRACE)
struct PlanTrace { bool on = false; std::string path; PlanTrace() { const char* e = getenv("LDM_PLAN_TRACE"); if (e && *e) { on = true; path = e; } } };
static PlanTrace g_plan_trace;
struct LaneCtx { GradSyncState* sync = nullptr; };   // the gradient exchange of the backward plans (comm stream, events)

static int run_plan(const Plan& plan, const Bases& bs, const int* rt, hipStream_t s, size_t begin = 0, size_t end = (size_t)-1,
                    LaneCtx lanes = LaneCtx()) {
    if (end > plan.ops.size()) end = plan.ops.size();
    // LDM_PLAN_TRACE=<file>: measurement aid (tools/plan_trace.py) -- a HIP event before every op, one CSV row per op appended
    const char* trace_path = g_plan_trace.on ? g_plan_trace.path.c_str() : nullptr;
    std::vector<hipEvent_t> tev;
    if (trace_path) { tev.resize(end - begin + 1); for (auto& e : tev) HIP_TRY(hipEventCreate(&e)); }
    struct TraceDone {
        const Plan& plan; size_t begin, end; std::vector<hipEvent_t>& ev; hipStream_t s; const char* path;
        ~TraceDone() {
            if (ev.empty()) return;
            (void)hipEventRecord(ev.back(), s); (void)hipEventSynchronize(ev.back());
            FILE* f = fopen(path, "a");
            for (size_t oi = begin; oi < end && f; ++oi) {
                const Op& o = plan.ops[oi]; float ms = 0.f; (void)hipEventElapsedTime(&ms, ev[oi - begin], ev[oi - begin + 1]);
                fprintf(f, "%zu,%zu,%d,%.3f", plan.ops.size(), oi, (int)o.kind, ms * 1e3f);
                if (o.kind == OP_CONV || o.kind == OP_FINALIZE)
                    fprintf(f, ",M=%d k=%d s=%d ups=%d cin=%d+%d(x%d) cin1=%d couts=%d cfg=%dx%dx%d splitk=%d halo=%d mtps=%d qps=%d", o.i[15], o.i[11], o.i[12],
                            o.i[14], o.i[0], o.i[1], o.i[19], o.i[2] + o.i[3], o.i[16], o.cc.wgm, o.cc.wgn, o.cc.bk, o.cc.splitk, o.cc.halo, o.cc.mtps, o.cc.qps);
                else fprintf(f, ",i=%d %d %d %d %d %d", o.i[0], o.i[1], o.i[2], o.i[3], o.i[4], o.i[5]);
                fprintf(f, "\n");
            }
            if (f) fclose(f);
            for (auto e : ev) (void)hipEventDestroy(e);
        }
    } trace_done{plan, begin, end, tev, s, trace_path};
    for (size_t oi = begin; oi < end; ++oi) {
        const Op& o = plan.ops[oi];
        const int* i = o.i;
        if (trace_path) HIP_TRY(hipEventRecord(tev[oi - begin], s));
        switch (o.kind) {
            case OP_PACK: {
                // two fp32 NCDHW sources (x | cond) -> one zero-padded NDHWC bf16 tensor
                const long total = (long)i[0] * i[3] * i[2];
                int cx = rt[0], cc = rt[1];
                if (i[4]) { cx = i[1]; cc = 0; }         // internal pack (fixed channel count, single source)
                if (cx + cc != i[1]) return fail(LDM_ERR_BAD_ARG, "x_channels + cond_channels = %d, model expects %d", cx + cc, i[1]);
                hipLaunchKernelGGL(pack2_ncdhw_kernel, dim3(grid_for(total)), dim3(256), 0, s, (const float*)rp(bs, o.r[0]), cx,
                                   (const float*)rp(bs, o.r[1]), cc, (bf16_t*)rp(bs, o.r[2]), i[0], i[2], i[3]);
                break; }
            case OP_IM2COL: {           // i: N, cin, Kp, D, H, W, internal
                int cx = rt[0], cc = rt[1];
                if (i[6]) { cx = i[1]; cc = 0; }
                if (cx + cc != i[1]) return fail(LDM_ERR_BAD_ARG, "x_channels + cond_channels = %d, model expects %d", cx + cc, i[1]);
                const long total = (long)i[0] * i[3] * i[4] * i[5] * (i[2] / 8);
                hipLaunchKernelGGL(pack_im2col_kernel, dim3(grid_for(total, 256, 16384)), dim3(256), 0, s, (const float*)rp(bs, o.r[0]), cx,
                                   (const float*)rp(bs, o.r[1]), cc, (bf16_t*)rp(bs, o.r[2]), i[0], i[3], i[4], i[5], i[2]);
                break; }
            case OP_CONV_THIN: {        // i: N, D, H, W, Cin, CoutPad, CoutReal
                ThinParams q{}; q.x = (const bf16_t*)rp(bs, o.r[0]); q.w = (const bf16_t*)rp(bs, o.r[2]); q.bias = (const float*)rp(bs, o.r[6]);
                q.out = (float*)rp(bs, o.r[10]); q.N = i[0]; q.D = i[1]; q.H = i[2]; q.W = i[3]; q.Cin = i[4]; q.CoutPad = i[5]; q.CoutReal = i[6];
                q.x3_c = i[7];
                if (!q.w) return fail(LDM_ERR_NOT_LOADED, "the weight arena is empty");
                q.td = (q.D + THIN_TD - 1) / THIN_TD; q.th = (q.H + THIN_TH - 1) / THIN_TH; q.tw = (q.W + THIN_TW - 1) / THIN_TW;
                static bool attr_tab[32] = {}; bool& attr_set = attr_flag(attr_tab);   // per device
                if (!attr_set) { HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3_thin_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, THIN_LDS)); attr_set = true; }
                hipLaunchKernelGGL(conv3_thin_kernel, dim3((unsigned)((long)q.N * q.td * q.th * q.tw)), dim3(256), THIN_LDS, s, q);
                break; }
            case OP_CONV_BLOCK: {       // i: N, D, H, W, Cin, TH, temb stride, couts (0 = 64 | 128: conv3_block128_kernel)
                BlockParams q{}; q.x = (const bf16_t*)rp(bs, o.r[0]); q.w = (const bf16_t*)rp(bs, o.r[2]); q.bias = (const float*)rp(bs, o.r[6]);
                q.temb = (const float*)rp(bs, o.r[8]); q.temb_stride = i[6]; q.residual = (const bf16_t*)rp(bs, o.r[9]); q.out = (bf16_t*)rp(bs, o.r[10]);
                q.stats = (float*)rp(bs, o.r[12]); q.N = i[0]; q.D = i[1]; q.H = i[2]; q.W = i[3]; q.Cin = i[4];
                if (!q.w) return fail(LDM_ERR_NOT_LOADED, "the weight arena is empty");
                if (i[7] == 128) HIP_TRY(launch_conv_block128(q, s)); else HIP_TRY(launch_conv_block(q, i[5], s));
                break; }
            case OP_UPS_SPLIT32: {      // i: N, C, D, H, W of the source, upsample (1) or same size (0)
                hipLaunchKernelGGL(upsample_split_f32_kernel, dim3(grid_for(((long)i[0] * i[2] * i[3] * i[4] << (3 * i[5])) * (i[1] / 4), 256, 4096)), dim3(256), 0, s,
                                   (const float*)rp(bs, o.r[0]), (bf16_t*)rp(bs, o.r[1]), i[0], i[1], i[2], i[3], i[4], i[5]);
                break; }
            case OP_PACK32: {
                const long total = (long)i[0] * i[3] * i[2];
                int cx = rt[0], cc = rt[1];
                if (i[4]) { cx = i[1]; cc = 0; }
                if (cx + cc != i[1]) return fail(LDM_ERR_BAD_ARG, "x_channels + cond_channels = %d, model expects %d", cx + cc, i[1]);
                hipLaunchKernelGGL(pack2_ncdhw_f32_kernel, dim3(grid_for(total)), dim3(256), 0, s, (const float*)rp(bs, o.r[0]), cx,
                                   (const float*)rp(bs, o.r[1]), cc, (float*)rp(bs, o.r[2]), i[0], i[2], i[3]);
                break; }
            case OP_CONV32: case OP_FIN32: {
                Conv32Params p{};
                p.xa = (const float*)rp(bs, o.r[0]); p.xb = (const float*)rp(bs, o.r[1]); p.ca = i[0]; p.cb = i[1];
                p.w = (const float*)rp(bs, o.r[2]);
                if (!p.w) return fail(LDM_ERR_NOT_LOADED, "fp32 precision: the fp32 weight arena is empty (re-upload the parameters after ldm_model_set_precision)");
                p.N = i[4]; p.Din = i[5]; p.Hin = i[6]; p.Win = i[7]; p.Dout = i[8]; p.Hout = i[9]; p.Wout = i[10];
                p.ksize = i[11]; p.stride = i[12]; p.pad = i[13]; p.ups = i[14] & 1; p.exact = (i[14] >> 1) & 1; p.M = i[15];
                p.CoutS = i[16]; p.CoutPad = i[17]; p.CoutReal = i[18]; p.nchunk = i[19]; p.steps = i[11] * i[11] * i[11] * i[19];
                p.splitk = o.cc.splitk; p.steps_per_split = (p.steps + p.splitk - 1) / p.splitk; p.mtiles = i[23]; p.ntiles = i[20];
                p.bias = (const float*)rp(bs, o.r[6]); p.temb = (const float*)rp(bs, o.r[8]); p.temb_stride = i[21];
                p.residual = (const float*)rp(bs, o.r[9]);
                if (i[22]) p.out_ncdhw = (float*)rp(bs, o.r[10]); else p.out = (float*)rp(bs, o.r[10]);
                p.partial = (float*)rp(bs, o.r[11]);
                if (o.kind == OP_CONV32 && o.cc.bk == 32 && o.cc.wgn == 2) hipLaunchKernelGGL(conv_x3_kernel<128>, dim3(p.mtiles * p.ntiles * p.splitk), dim3(256), 0, s, p);
                else if (o.kind == OP_CONV32 && o.cc.bk == 32) hipLaunchKernelGGL(conv_x3_kernel<64>, dim3(p.mtiles * p.ntiles * p.splitk), dim3(256), 0, s, p);
                else if (o.kind == OP_CONV32 && o.cc.wgn == 2) hipLaunchKernelGGL(conv_f32_kernel<128>, dim3(p.mtiles * p.ntiles * p.splitk), dim3(256), 0, s, p);
                else if (o.kind == OP_CONV32) hipLaunchKernelGGL(conv_f32_kernel<64>, dim3(p.mtiles * p.ntiles * p.splitk), dim3(256), 0, s, p);
                else if (o.kind == OP_FIN32 && i[2] > 0) {
                    p.stats = (float*)rp(bs, o.r[12]); p.stats_nrb = i[2]; p.stats_rows = i[3];
                    hipLaunchKernelGGL(finalize_stats_f32_kernel, dim3(i[2], p.N), dim3(256), 0, s, p);
                } else hipLaunchKernelGGL(finalize_f32_kernel, dim3(grid_for((long)p.M * (p.CoutPad / 4), 256, 4096)), dim3(256), 0, s, p);
                break; }
            case OP_GEMM_LIGHT32: {
                LightX3Params p{};
                p.x = (const float*)rp(bs, o.r[0]); p.xb = (const float*)rp(bs, o.r[1]); p.ca = i[5];
                p.w = (const float*)rp(bs, o.r[2]);
                if (!p.w) return fail(LDM_ERR_NOT_LOADED, "fp32 precision: the fp32 weight arena is empty (re-upload the parameters after ldm_model_set_precision)");
                p.bias = (const float*)rp(bs, o.r[6]); p.residual = (const float*)rp(bs, o.r[9]);
                p.out = (float*)rp(bs, o.r[10]); p.stats = (float*)rp(bs, o.r[12]);
                p.M = i[0]; p.K = i[1]; p.CoutS = i[2];
                HIP_TRY(launch_gemm_light_x3(p, i[3], i[4], s));
                break; }
            case OP_GN_STATS32: case OP_GN_APPLY32: {
                Gn32Params p{}; p.xa = (const float*)rp(bs, o.r[0]); p.xb = (const float*)rp(bs, o.r[1]); p.ca = i[0]; p.cb = i[1];
                if (o.kind == OP_GN_STATS32) {
                    p.DHW = i[2]; p.nslab = i[3]; p.rows_per_slab = i[4]; p.N = i[5]; p.partial = (float*)rp(bs, o.r[4]);
                    hipLaunchKernelGGL(gn_stats_f32_kernel, dim3(i[3], i[5]), dim3(256), 0, s, p);
                } else if (i[5] > 0) {       // statistics fold + apply in one launch (inference plans)
                    Gn32FusedParams q{}; q.xa = p.xa; q.xb = p.xb; q.ca = p.ca; q.cb = p.cb; q.DHW = i[2]; q.N = i[3]; q.silu = i[4];
                    q.nslab = i[5]; q.groups = i[6]; q.rows_per_block = i[7]; q.eps = o.f[0]; q.partial = (const float*)rp(bs, o.r[4]);
                    q.gamma = (const float*)rp(bs, o.r[6]); q.beta = (const float*)rp(bs, o.r[7]);
                    if (i[10]) { q.sa = (const float*)rp(bs, o.r[8]); q.sb = (const float*)rp(bs, o.r[9]); q.nrb_a = i[10]; q.nrb_b = i[11]; q.partial = nullptr; }
                    if (i[9]) q.out_hl = (bf16_t*)rp(bs, o.r[3]); else q.out = (float*)rp(bs, o.r[3]);
                    hipLaunchKernelGGL(gn32_fold_apply_kernel, dim3(i[8], (p.ca + p.cb + 63) / 64, i[3]), dim3(256), 0, s, q);
                } else {
                    p.DHW = i[2]; p.N = i[3]; p.silu = i[4]; p.ab = (const float*)rp(bs, o.r[5]); p.out = (float*)rp(bs, o.r[3]);
                    hipLaunchKernelGGL(gn_apply_f32_kernel, dim3(grid_for((long)i[3] * i[2] * ((i[0] + i[1]) / 4), 256, 4096)), dim3(256), 0, s, p);
                }
                break; }
            case OP_ATTN32: {
                Attn32Params p{}; p.qkv = (const float*)rp(bs, o.r[0]); p.out = (float*)rp(bs, o.r[1]);
                p.B = i[0]; p.N = i[1]; p.C = i[2]; p.heads = i[3]; p.d = i[4]; p.scale = o.f[0]; p.lse = (float*)rp(bs, o.r[2]); p.x3 = i[5];
                HIP_TRY(launch_attn_f32(p, s));
                break; }
            case OP_GEMV32: {
                const float* w = (const float*)rp(bs, o.r[0]);
                if (!w) return fail(LDM_ERR_NOT_LOADED, "fp32 precision: the fp32 weight arena is empty");
                hipLaunchKernelGGL(gemv_f32_kernel, dim3((i[1] + 3) / 4, i[5]), dim3(256), 0, s, w, (const float*)rp(bs, o.r[1]),
                                   (const float*)rp(bs, o.r[2]), (float*)rp(bs, o.r[3]), i[0], i[1], i[2], i[3], i[4]);
                break; }
            case OP_TAP: {               // i: N, C real, C stored, DHW, element size, mode
                float* dst = (float*)rp(bs, o.r[1]); const float* src = (const float*)rp(bs, o.r[2]);
                const long total = (long)i[0] * i[1] * i[3];
                if (dst) {
                    if (i[4] == 4) hipLaunchKernelGGL(tap_export_kernel<float>, dim3(grid_for(total)), dim3(256), 0, s, (const float*)rp(bs, o.r[0]), dst, i[0], i[1], i[2], i[3]);
                    else hipLaunchKernelGGL(tap_export_kernel<bf16_t>, dim3(grid_for(total)), dim3(256), 0, s, (const bf16_t*)rp(bs, o.r[0]), dst, i[0], i[1], i[2], i[3]);
                }
                if (src && i[5] == 2) {
                    const long tot2 = (long)i[0] * i[3] * i[2];
                    if (i[4] == 4) hipLaunchKernelGGL(pack2_ncdhw_f32_kernel, dim3(grid_for(tot2)), dim3(256), 0, s, src, i[1], (const float*)nullptr, 0, (float*)rp(bs, o.r[0]), i[0], i[2], i[3]);
                    else hipLaunchKernelGGL(pack2_ncdhw_kernel, dim3(grid_for(tot2)), dim3(256), 0, s, src, i[1], (const float*)nullptr, 0, (bf16_t*)rp(bs, o.r[0]), i[0], i[2], i[3]);
                }
                break; }
            case OP_FIN_GN: {            // an OP_FINALIZE whose r[3..5] = gamma, beta, normalised output; i[0..2] = groups, silu, log2(channels per group)
                FinGnParams q{}; FinalizeParams& f = q.f;
                f.partial = (const float*)rp(bs, o.r[11]); f.splitk = o.cc.splitk; f.M = i[15]; f.CoutPad = i[17]; f.CoutS = i[16]; f.CoutReal = i[18];
                f.DHWo = i[8] * i[9] * i[10];
                f.bias = (const float*)rp(bs, o.r[6]); f.bias2 = (const float*)rp(bs, o.r[7]); f.temb = (const float*)rp(bs, o.r[8]); f.temb_stride = i[21];
                f.residual = (const bf16_t*)rp(bs, o.r[9]); f.out = (bf16_t*)rp(bs, o.r[10]);
                q.gamma = (const float*)rp(bs, o.r[3]); q.beta = (const float*)rp(bs, o.r[4]); q.y = (bf16_t*)rp(bs, o.r[5]);
                q.groups = i[0]; q.silu = i[1]; q.lg = i[2]; q.eps = o.f[0];
                HIP_TRY(launch_fin_gn(q, i[4], wt_stores(), s));
                break; }
            case OP_CONV: case OP_FINALIZE: {
                ConvParams p{};
                p.x0a = (const bf16_t*)rp(bs, o.r[0]); p.x0b = (const bf16_t*)rp(bs, o.r[1]); p.c0a = i[0]; p.c0b = i[1];
                p.w0 = (const bf16_t*)rp(bs, o.r[2]);
                p.x1a = (const bf16_t*)rp(bs, o.r[3]); p.x1b = (const bf16_t*)rp(bs, o.r[4]); p.c1a = i[2]; p.c1b = i[3];
                p.w1 = (const bf16_t*)rp(bs, o.r[5]);
                p.zero_page = (const bf16_t*)bs.p[BASE_W];
                p.N = i[4]; p.Din = i[5]; p.Hin = i[6]; p.Win = i[7]; p.Dout = i[8]; p.Hout = i[9]; p.Wout = i[10];
                p.ksize = i[11]; p.stride = i[12]; p.pad = i[13]; p.ups = i[14] & 1; p.exact = (i[14] >> 1) & 1; p.M = i[15];
                p.phase_mode = (i[14] >> 2) & 1; p.mtiles_pp = p.phase_mode ? i[23] / (8 * i[4]) : 0;
                if (i[14] & 8) { p.x3_n = (i[0] / 2) / 64; p.raw_partial = (i[14] & (16 | 32)) ? 0 : 1; }   // fp32 precision: 3 x bf16 product on the halo kernel
                p.CoutS = i[16]; p.CoutPad = i[17]; p.CoutReal = i[18]; p.nchunk0 = i[19]; p.nchunk1 = i[20];
                p.steps0 = i[11] * i[11] * i[11] * i[19]; p.steps1 = i[20];
                p.splitk = o.cc.splitk; p.steps_per_split = (p.steps0 + p.steps1 + p.splitk - 1) / p.splitk;
                p.mtiles = i[23]; p.ntiles = p.CoutPad / (64 * o.cc.wgn);
                p.halo_mtps = o.cc.mtps; p.q_per_split = o.cc.qps;
                if (o.cc.halo) p.steps_per_split = 3 * o.cc.qps;
                p.bias = (const float*)rp(bs, o.r[6]); p.bias2 = (const float*)rp(bs, o.r[7]);
                p.temb = (const float*)rp(bs, o.r[8]); p.temb_stride = i[21];
                p.residual = (const bf16_t*)rp(bs, o.r[9]);
                if (i[22]) { p.out_f32 = (float*)rp(bs, o.r[10]); p.out = nullptr; }
                else { p.out = (bf16_t*)rp(bs, o.r[10]); p.out_f32 = nullptr; }
                p.partial = (float*)rp(bs, o.r[11]);
                p.stats = (float*)rp(bs, o.r[12]);
                if (i[14] & 16) {                          // ... with the epilogue fused (splitk 1): fp32 NDHWC output, fp32 residual
                    p.out32 = (float*)rp(bs, o.r[10]); p.residual32 = (const float*)rp(bs, o.r[9]); p.out = nullptr; p.residual = nullptr;
                }
                p.slab_lg = o.cc.slab_lg;
                if (o.kind == OP_CONV) { LDM_TRY(launch_conv(p, o.cc, s)); }
                else {
                    FinalizeParams f{}; f.partial = p.partial; f.splitk = p.splitk; f.M = p.M; f.CoutPad = p.CoutPad;
                    f.CoutS = p.CoutS; f.CoutReal = p.CoutReal; f.DHWo = p.Dout * p.Hout * p.Wout;
                    f.bias = p.bias; f.bias2 = p.bias2; f.temb = p.temb; f.temb_stride = p.temb_stride; f.residual = p.residual;
                    f.out = p.out; f.out_f32 = p.out_f32; f.stats = p.stats;
                    launch_finalize(f, wt_stores(), s);
                }
                break; }
            case OP_GEMM_LIGHT: {       // i: M, K, CoutS, CoutPad, big
                LightParams p{}; p.x = (const bf16_t*)rp(bs, o.r[0]); p.w = (const bf16_t*)rp(bs, o.r[2]); p.bias = (const float*)rp(bs, o.r[6]);
                p.residual = (const bf16_t*)rp(bs, o.r[9]); p.out = (bf16_t*)rp(bs, o.r[10]); p.stats = (float*)rp(bs, o.r[12]);
                p.M = i[0]; p.K = i[1]; p.CoutS = i[2]; p.xb = (const bf16_t*)rp(bs, o.r[1]); p.ca = p.xb ? i[5] : i[1];
                HIP_TRY(launch_gemm_light(p, i[3], i[4], s));
                break; }
            case OP_GN_STATS: {
                GnStatsParams p{}; p.xa = (const bf16_t*)rp(bs, o.r[0]); p.xb = (const bf16_t*)rp(bs, o.r[1]); p.ca = i[0]; p.cb = i[1];
                p.DHW = i[2]; p.nslab = i[3]; p.rows_per_slab = i[4]; p.partial = (float*)rp(bs, o.r[4]);
                hipLaunchKernelGGL(gn_stats_kernel, dim3(i[3], i[5]), dim3(256), 0, s, p);
                break; }
            case OP_GN_FINALIZE: {
                GnFinalizeParams p{}; p.partial = (const float*)rp(bs, o.r[4]); p.nslab = i[0]; p.C = i[1]; p.Creal = i[1]; p.groups = i[2];
                p.DHW = i[3]; p.eps = o.f[0]; p.gamma = (const float*)rp(bs, o.r[1]); p.beta = (const float*)rp(bs, o.r[2]);
                p.ab = (float*)rp(bs, o.r[5]); p.mr = (float*)rp(bs, o.r[6]);
                hipLaunchKernelGGL(gn_finalize_kernel, dim3(i[2], i[4]), dim3(256), 0, s, p);
                break; }
            case OP_GN_PREP: {
                GnPrepParams p{}; p.sa = (const float*)rp(bs, o.r[0]); p.sb = (const float*)rp(bs, o.r[1]); p.ca = i[0]; p.cb = i[1];
                p.nrb_a = i[2]; p.nrb_b = i[6]; p.groups = i[3]; p.DHW = i[4]; p.eps = o.f[0];
                p.gamma = (const float*)rp(bs, o.r[2]); p.beta = (const float*)rp(bs, o.r[3]); p.ab = (float*)rp(bs, o.r[5]); p.mr = (float*)rp(bs, o.r[6]);
                hipLaunchKernelGGL(gn_prep_kernel, dim3(i[3], i[5]), dim3(256), 0, s, p);
                break; }
            case OP_GN_FUSED: {
                GnFusedParams p{}; p.xa = (const bf16_t*)rp(bs, o.r[0]); p.xb = (const bf16_t*)rp(bs, o.r[1]); p.ca = i[0]; p.cb = i[1];
                p.sa = (const float*)rp(bs, o.r[7]); p.sb = (const float*)rp(bs, o.r[8]); p.nrb_a = i[2]; p.nrb_b = i[3];
                p.groups = i[4]; p.DHW = i[5]; p.N = i[6]; p.silu = i[7]; p.rows_per_block = i[8]; p.eps = o.f[0];
                p.gamma = (const float*)rp(bs, o.r[2]); p.beta = (const float*)rp(bs, o.r[3]); p.out = (bf16_t*)rp(bs, o.r[9]);
                p.ab = (float*)rp(bs, o.r[5]); p.mr = (float*)rp(bs, o.r[6]); p.xcd_rows = xcd_rows_mode();
                if (wt_stores()) hipLaunchKernelGGL(gn_fused_apply_kernel<true>, dim3(i[9], (i[0] + i[1] + 63) / 64, i[6]), dim3(256), 0, s, p);
                else hipLaunchKernelGGL(gn_fused_apply_kernel<false>, dim3(i[9], (i[0] + i[1] + 63) / 64, i[6]), dim3(256), 0, s, p);
                break; }
            case OP_GN_APPLY: {
                GnApplyParams p{}; p.xa = (const bf16_t*)rp(bs, o.r[0]); p.xb = (const bf16_t*)rp(bs, o.r[1]); p.ca = i[0]; p.cb = i[1];
                p.DHW = i[2]; p.N = i[3]; p.silu = i[4]; p.ab = (const float*)rp(bs, o.r[5]); p.out = (bf16_t*)rp(bs, o.r[3]);
                const long total = (long)i[3] * i[2] * ((i[0] + i[1]) / 8);
                hipLaunchKernelGGL(gn_apply_kernel, dim3(grid_for(total, 256, 2048)), dim3(256), 0, s, p);
                break; }
            case OP_ATTN: {
                AttnParams p{}; p.qkv = (const bf16_t*)rp(bs, o.r[0]); p.out = (bf16_t*)rp(bs, o.r[1]);
                p.B = i[0]; p.N = i[1]; p.C = i[2]; p.heads = i[3]; p.d = i[4]; p.scale = o.f[0]; p.lse = (float*)rp(bs, o.r[2]);
                HIP_TRY(launch_attn_fwd(p, s));
                break; }
            case OP_TEMB_ROW: {         // i: rows, B; bs[TTAB] = table, bs[SST] = sampler state
                const float* tab = (const float*)bs.p[BASE_TTAB]; const SamplerState* st = (const SamplerState*)bs.p[BASE_SST];
                if (!tab || !st || bs.sampler_steps < 1) return fail(LDM_ERR_BAD_ARG, "denoise-step plan without a sampler / time-embedding table");
                hipLaunchKernelGGL(temb_row_kernel, dim3((i[0] / 4 + 255) / 256, i[1]), dim3(256), 0, s, tab, st, (float*)rp(bs, o.r[2]), i[0], i[0], bs.sampler_steps);
                break; }
            case OP_SINUSOID:
                hipLaunchKernelGGL(temb_sinusoid_kernel, dim3(grid_for((long)i[0] * i[1])), dim3(256), 0, s,
                                   (const float*)rp(bs, o.r[0]), (float*)rp(bs, o.r[1]), i[0], i[1]);
                break;
            case OP_GEMV:
                hipLaunchKernelGGL(gemv_bf16_kernel, dim3((i[1] + 3) / 4, i[5]), dim3(256), 0, s,
                                   (const bf16_t*)rp(bs, o.r[0]), (const float*)rp(bs, o.r[1]), (const float*)rp(bs, o.r[2]),
                                   (float*)rp(bs, o.r[3]), i[0], i[1], i[2], i[3], i[4]);
                break;
            case OP_VAE_HEADS: {
                const long total = (long)i[0] * i[1] * i[2];
                hipLaunchKernelGGL(vae_heads_kernel, dim3(grid_for(total)), dim3(256), 0, s, (const float*)rp(bs, o.r[0]),
                                   (const float*)rp(bs, o.r[1]), (float*)rp(bs, o.r[2]), (float*)rp(bs, o.r[3]), (float*)rp(bs, o.r[4]),
                                   i[0], i[1], i[2]);
                break; }
            // ------------------------------------------------------------------ backward ops
            case OP_WT: {
                const int cols = rup(i[1], 32);
                hipLaunchKernelGGL(weight_flip_transpose_kernel, dim3((cols + 63) / 64, (i[4] + 63) / 64, i[0]), dim3(256), 0, s,
                                   (const bf16_t*)rp(bs, o.r[0]), (bf16_t*)rp(bs, o.r[1]), i[0], i[1], i[2], i[3], i[4], i[5], i[6]);
                break; }
            case OP_WT_BATCH:
                if (plan.wt_tab.nblocks && i[0]) {
                    if (!bs.p[BASE_W32]) return fail(LDM_ERR_NOT_LOADED, "fp32 precision: the fp32 weight arena is empty");
                    hipLaunchKernelGGL(weight_flip_transpose_batched_f32_kernel, dim3(plan.wt_tab.nblocks), dim3(256), 0, s,
                                       (const WtDesc*)plan.wt_tab.descs, (const int2*)plan.wt_tab.map, (const char*)bs.p[BASE_W32], bs.p[BASE_WS]);
                } else if (plan.wt_tab.nblocks)
                    hipLaunchKernelGGL(weight_flip_transpose_batched_kernel, dim3(plan.wt_tab.nblocks), dim3(256), 0, s,
                                       (const WtDesc*)plan.wt_tab.descs, (const int2*)plan.wt_tab.map, (const char*)bs.p[BASE_W], bs.p[BASE_WS]);
                break;
            case OP_COLSUM_BATCH:       // i: first block, end block of the table's block map
                if (i[1] > i[0])
                    hipLaunchKernelGGL(colsum_finalize_batched_kernel, dim3(i[1] - i[0]), dim3(256), 0, s,
                                       (const ColsumDesc*)plan.cs_tab.descs, (const int2*)plan.cs_tab.map + i[0], (char*)bs.p[BASE_WS]);
                break;
            case OP_EXPORT_BATCH:
                if (!bs.p[BASE_IO4]) return fail(LDM_ERR_BAD_ARG, "backward without a gradient buffer");
                if (i[1] > i[0])
                    hipLaunchKernelGGL(grad_export_batched_kernel, dim3(i[1] - i[0]), dim3(256), 0, s,
                                       (const ExportDesc*)plan.exp_tab.descs, (const int2*)plan.exp_tab.map + i[0], (const char*)bs.p[BASE_WS],
                                       (float*)bs.p[BASE_IO4]);
                break;
            case OP_BUCKET:             // i: first element, element count of a final tail range of the flat gradient buffer
                if (lanes.sync) LDM_TRY(grad_sync_bucket(*lanes.sync, (float*)bs.p[BASE_IO4] + i[0], i[1], s));
                break;
            case OP_BUCKET_JOIN:
                if (lanes.sync) LDM_TRY(grad_sync_join(*lanes.sync, s));
                break;
            case OP_WGRAD: {
                if (i[21]) {                 // fp32 precision
                    Wgrad32Params q{}; q.dy = (const float*)rp(bs, o.r[0]); q.cdy = i[0]; q.x = (const float*)rp(bs, o.r[1]); q.cx = i[1];
                    q.dw = (float*)rp(bs, o.r[2]); q.Cout = i[2]; q.Cin = i[3]; q.dw_ld = i[4]; q.dw_ci_off = i[5];
                    q.N = i[6]; q.Din = i[7]; q.Hin = i[8]; q.Win = i[9]; q.Dout = i[10]; q.Hout = i[11]; q.Wout = i[12];
                    q.ksize = i[13]; q.stride = i[14]; q.pad = i[15]; q.ups = i[16]; q.M = i[17];
                    q.co_tiles = (q.Cout + 127) / 128; q.ci_tiles = (q.Cin + 127) / 128; q.ksplit = i[18];
                    q.slab_stride = (long)i[13] * i[13] * i[13] * i[19] * i[4];
                    hipLaunchKernelGGL(wgrad_f32_kernel, dim3(q.co_tiles * q.ci_tiles * i[13] * i[13] * i[13] * q.ksplit), dim3(256), 0, s, q);
                    break;
                }
                WgradParams p{}; p.dy = (const bf16_t*)rp(bs, o.r[0]); p.cdy = i[0]; p.x = (const bf16_t*)rp(bs, o.r[1]); p.cx = i[1];
                p.dw = (float*)rp(bs, o.r[2]); p.Cout = i[2]; p.Cin = i[3]; p.dw_ld = i[4]; p.dw_ci_off = i[5];
                p.N = i[6]; p.Din = i[7]; p.Hin = i[8]; p.Win = i[9]; p.Dout = i[10]; p.Hout = i[11]; p.Wout = i[12];
                p.ksize = i[13]; p.stride = i[14]; p.pad = i[15]; p.ups = i[16]; p.M = i[17];
                p.co_tiles = (p.Cout + 127) / 128; p.ci_tiles = (p.Cin + 127) / 128;
                p.ksplit = i[18]; p.slab_stride = (long)i[13] * i[13] * i[13] * i[19] * i[4];
                if ((long)p.M * p.cdy * 2 >= (1L << 32) || (long)p.N * p.Din * p.Hin * p.Win * p.cx * 2 >= (1L << 32))
                    return fail(LDM_ERR_UNSUPPORTED, "weight gradient: tensor exceeds 4 GiB");
                LDM_TRY(launch_wgrad(p, s));
                break; }
            case OP_EXPORT:
                hipLaunchKernelGGL(grad_export_kernel, dim3((i[6] + 63) / 64, i[5]), dim3(256), 0, s, (const float*)rp(bs, o.r[0]),
                                   (float*)rp(bs, o.r[1]), i[0], i[1], i[2], i[3], i[4], i[5], i[6], i[7] < 1 ? 1 : i[7],
                                   (long)i[0] * i[1] * i[2]);
                break;
            case OP_COLSUM: {
                hipLaunchKernelGGL(colsum_finalize_kernel, dim3((i[4] + 15) / 16, i[3] ? 1 : i[0]), dim3(256), 0, s, (const float*)rp(bs, o.r[4]),
                                   (float*)rp(bs, o.r[0]), i[0], i[1], i[2], i[3], i[4], i[5]);
                break; }
            case OP_GNB: {
                if (i[10]) {                 // fp32 precision: stats and apply on fp32 tensors, the fold kernel is type agnostic
                    float* flat = (float*)bs.p[BASE_IO4];
                    if (!flat) return fail(LDM_ERR_BAD_ARG, "backward without a gradient buffer");
                    const int C = i[0] + i[1];
                    Gnb32Params q{}; q.dy = (const float*)rp(bs, o.r[0]); q.xa = (const float*)rp(bs, o.r[1]); q.xb = (const float*)rp(bs, o.r[2]);
                    q.ca = i[0]; q.cb = i[1]; q.ab = (const float*)rp(bs, o.r[3]); q.mr = (const float*)rp(bs, o.r[5]); q.gamma = (const float*)rp(bs, o.r[6]);
                    q.groups = i[2]; q.DHW = i[3]; q.N = i[4]; q.silu = i[5]; q.nslab = i[6]; q.rows_per_slab = i[7];
                    q.partial = (float*)rp(bs, o.r[4]); q.gsum = (const float*)rp(bs, o.r[7]);
                    q.acc_a = (const float*)rp(bs, o.r[10]); q.acc_b = (const float*)rp(bs, o.r[11]); q.dxa = (float*)rp(bs, o.r[12]); q.dxb = (float*)rp(bs, o.r[13]);
                    GnBwdParams f{}; f.ca = i[0]; f.cb = i[1]; f.gamma = q.gamma; f.groups = i[2]; f.DHW = i[3]; f.N = i[4]; f.nslab = i[6];
                    f.partial = q.partial; f.gsum = (float*)rp(bs, o.r[7]); f.dgamma_n = (float*)rp(bs, o.r[8]); f.dbeta_n = (float*)rp(bs, o.r[9]);
                    if (i[4] == 1) { f.dgamma_n = flat + i[8]; f.dbeta_n = flat + i[9]; }
                    hipLaunchKernelGGL(gnb32_stats_kernel, dim3(i[6], i[4]), dim3(256), 0, s, q);
                    hipLaunchKernelGGL(gn_bwd_finalize_kernel, dim3(i[2], i[4]), dim3(256), 0, s, f);
                    hipLaunchKernelGGL(gnb32_apply_kernel, dim3(grid_for((long)i[4] * i[3] * (C / 4), 256, 4096)), dim3(256), 0, s, q);
                    if (i[4] > 1) {
                        hipLaunchKernelGGL(rowsum_n_kernel, dim3((C + 255) / 256), dim3(256), 0, s, (const float*)f.dgamma_n, flat + i[8], i[4], C);
                        hipLaunchKernelGGL(rowsum_n_kernel, dim3((C + 255) / 256), dim3(256), 0, s, (const float*)f.dbeta_n, flat + i[9], i[4], C);
                    }
                    break;
                }
                GnBwdParams p{}; p.dy = (const bf16_t*)rp(bs, o.r[0]); p.xa = (const bf16_t*)rp(bs, o.r[1]); p.xb = (const bf16_t*)rp(bs, o.r[2]);
                p.ca = i[0]; p.cb = i[1]; p.ab = (const float*)rp(bs, o.r[3]); p.mr = (const float*)rp(bs, o.r[5]);
                p.gamma = (const float*)rp(bs, o.r[6]); p.groups = i[2]; p.DHW = i[3]; p.N = i[4]; p.silu = i[5]; p.nslab = i[6];
                p.rows_per_slab = i[7]; p.partial = (float*)rp(bs, o.r[4]); p.gsum = (float*)rp(bs, o.r[7]);
                p.dgamma_n = (float*)rp(bs, o.r[8]); p.dbeta_n = (float*)rp(bs, o.r[9]);
                if (i[4] == 1 && bs.p[BASE_IO4]) { p.dgamma_n = (float*)bs.p[BASE_IO4] + i[8]; p.dbeta_n = (float*)bs.p[BASE_IO4] + i[9]; }
                p.acc_a = (const bf16_t*)rp(bs, o.r[10]); p.acc_b = (const bf16_t*)rp(bs, o.r[11]);
                p.dxa = (bf16_t*)rp(bs, o.r[12]); p.dxb = (bf16_t*)rp(bs, o.r[13]);
                const int C = i[0] + i[1];
                float* flat = (float*)bs.p[BASE_IO4];
                if (!flat) return fail(LDM_ERR_BAD_ARG, "backward without a gradient buffer");
                hipLaunchKernelGGL(gn_bwd_stats_kernel, dim3(i[6], i[4]), dim3(256), 0, s, p);
                if (i[11]) {                             // passes 2 + 3 in one launch (Builder::backward_gn decided)
                    p.rows_per_block = i[11];
                    p.cs = i[13] ? (float*)rp(bs, o.r[7]) : nullptr;
                    hipLaunchKernelGGL(gn_bwd_fold_apply_kernel, dim3(i[12], (C + 63) / 64, i[4]), dim3(256), 0, s, p);
                } else {
                    hipLaunchKernelGGL(gn_bwd_finalize_kernel, dim3(i[2], i[4]), dim3(256), 0, s, p);
                    hipLaunchKernelGGL(gn_bwd_apply_kernel, dim3(grid_for((long)i[4] * i[3] * (C / 8), 256, 2048)), dim3(256), 0, s, p);
                }
                if (i[4] > 1) {
                    hipLaunchKernelGGL(rowsum_n_kernel, dim3((C + 255) / 256), dim3(256), 0, s, (const float*)p.dgamma_n, flat + i[8], i[4], C);
                    hipLaunchKernelGGL(rowsum_n_kernel, dim3((C + 255) / 256), dim3(256), 0, s, (const float*)p.dbeta_n, flat + i[9], i[4], C);
                }
                break; }
            case OP_ATTN_BWD: {
                if (i[4]) {                  // fp32 precision
                    Attn32BwdParams q{}; q.qkv = (const float*)rp(bs, o.r[0]); q.o = (const float*)rp(bs, o.r[1]); q.d_o = (const float*)rp(bs, o.r[2]);
                    q.lse = (const float*)rp(bs, o.r[3]); q.delta = (float*)rp(bs, o.r[4]); q.dqkv = (float*)rp(bs, o.r[5]);
                    q.B = i[0]; q.N = i[1]; q.C = i[2]; q.d = i[3]; q.heads = i[2] / i[3]; q.scale = o.f[0];
                    HIP_TRY(launch_attn32_bwd(q, s));
                    break;
                }
                AttnBwdParams p{}; p.qkv = (const bf16_t*)rp(bs, o.r[0]); p.o = (const bf16_t*)rp(bs, o.r[1]); p.d_o = (const bf16_t*)rp(bs, o.r[2]);
                p.lse = (const float*)rp(bs, o.r[3]); p.delta = (float*)rp(bs, o.r[4]); p.dqkv = (bf16_t*)rp(bs, o.r[5]);
                p.B = i[0]; p.N = i[1]; p.C = i[2]; p.d = i[3]; p.heads = i[2] / i[3]; p.scale = o.f[0];
                HIP_TRY(launch_attn_bwd(p, s));
                break; }
            case OP_ADD:
                if (i[1]) { hipLaunchKernelGGL(add_f32_kernel, dim3(grid_for(i[0], 256, 4096)), dim3(256), 0, s, (const float*)rp(bs, o.r[0]),
                                               (const float*)rp(bs, o.r[1]), (float*)rp(bs, o.r[2]), (long)i[0]); break; }
                hipLaunchKernelGGL(add_bf16_kernel, dim3(grid_for(i[0], 256, 2048)), dim3(256), 0, s, (const bf16_t*)rp(bs, o.r[0]),
                                   (const bf16_t*)rp(bs, o.r[1]), (bf16_t*)rp(bs, o.r[2]), (long)i[0]);
                break;
            case OP_SUMPOOL:
                if (i[5]) { hipLaunchKernelGGL(sumpool2_f32_kernel, dim3(grid_for((long)i[0] * i[1] * i[2] * i[3] * (i[4] / 4), 256, 4096)), dim3(256), 0, s,
                                               (const float*)rp(bs, o.r[0]), (float*)rp(bs, o.r[1]), i[0], i[1], i[2], i[3], i[4]); break; }
                hipLaunchKernelGGL(sumpool2_kernel, dim3(grid_for((long)i[0] * i[1] * i[2] * i[3] * (i[4] / 8), 256, 2048)), dim3(256), 0, s,
                                   (const bf16_t*)rp(bs, o.r[0]), (bf16_t*)rp(bs, o.r[1]), i[0], i[1], i[2], i[3], i[4]);
                break;
            case OP_LIN_DX: {       // i: B, I, O, dy_stride, x_stride, silu, nz, fp32 weights
                if (i[7]) {
                    const float* w = (const float*)rp(bs, o.r[0]);
                    if (!w) return fail(LDM_ERR_NOT_LOADED, "fp32 precision: the fp32 weight arena is empty");
                    hipLaunchKernelGGL(linear_bwd_dx_part_f32w_kernel, dim3((i[1] + 255) / 256, i[0], i[6]), dim3(256), 0, s, w,
                                       (const float*)rp(bs, o.r[1]), (float*)rp(bs, o.r[4]), i[1], i[2], i[3], i[0]);
                    hipLaunchKernelGGL(linear_bwd_dx_fold_f32_kernel, dim3((i[1] + 255) / 256, i[0]), dim3(256), 0, s, (const float*)rp(bs, o.r[4]),
                                       (const float*)rp(bs, o.r[2]), (float*)rp(bs, o.r[3]), i[1], i[6], i[4], i[0], i[5]);
                    break;
                }
                hipLaunchKernelGGL(linear_bwd_dx_part_kernel, dim3((i[1] + 255) / 256, i[0], i[6]), dim3(256), 0, s, (const bf16_t*)rp(bs, o.r[0]),
                                   (const float*)rp(bs, o.r[1]), (float*)rp(bs, o.r[4]), i[1], i[2], i[3], i[0]);
                hipLaunchKernelGGL(linear_bwd_dx_fold_kernel, dim3((i[1] + 255) / 256, i[0]), dim3(256), 0, s, (const float*)rp(bs, o.r[4]),
                                   (const float*)rp(bs, o.r[2]), (float*)rp(bs, o.r[3]), i[1], i[6], i[4], i[0], i[5]);
                break; }
            case OP_VAE_HEADS_BWD: {
                const long total = (long)i[0] * i[3] * i[2];
                if (i[5]) hipLaunchKernelGGL(vae_heads_bwd_kernel<float>, dim3(grid_for(total)), dim3(256), 0, s, (const float*)rp(bs, o.r[0]), i[4],
                                             (const float*)rp(bs, o.r[1]), (const float*)rp(bs, o.r[2]), (const float*)rp(bs, o.r[3]),
                                             (const float*)rp(bs, o.r[6]), (float*)rp(bs, o.r[7]), i[0], i[1], i[2], i[3]);
                else hipLaunchKernelGGL(vae_heads_bwd_kernel<bf16_t>, dim3(grid_for(total)), dim3(256), 0, s, (const bf16_t*)rp(bs, o.r[0]), i[4],
                                        (const float*)rp(bs, o.r[1]), (const float*)rp(bs, o.r[2]), (const float*)rp(bs, o.r[3]),
                                        (const float*)rp(bs, o.r[6]), (bf16_t*)rp(bs, o.r[7]), i[0], i[1], i[2], i[3]);
                break; }
            case OP_LIN_DW:         // i: B, I, O, dy_stride, x_stride, silu, fp32 precision
                if (i[6]) { hipLaunchKernelGGL(linear_bwd_dw_f32_kernel, dim3(grid_for((long)i[2] * i[1], 256, 1 << 24)), dim3(256), 0, s,
                                               (const float*)rp(bs, o.r[0]), (const float*)rp(bs, o.r[1]), (float*)rp(bs, o.r[2]), (float*)rp(bs, o.r[3]),
                                               i[0], i[1], i[2], i[3], i[4], i[5]); break; }
                hipLaunchKernelGGL(linear_bwd_dw_kernel, dim3(grid_for((long)i[2] * i[1], 256, 1 << 24)), dim3(256), 0, s,
                                   (const float*)rp(bs, o.r[0]), (const float*)rp(bs, o.r[1]), (float*)rp(bs, o.r[2]), (float*)rp(bs, o.r[3]),
                                   i[0], i[1], i[2], i[3], i[4], i[5]);
                break;
        }
    }
    HIP_TRY(hipGetLastError());
    return 0;
}

// ================================================================================================ C ABI
extern "C" {

int ldm_version(void) { return LDM_ABI_VERSION; }
const char* ldm_last_error(void) { return g_err; }

int ldm_unet_create(const ldm_unet_cfg* cfg, ldm_model** out) {
    if (!cfg || !out) return fail(LDM_ERR_BAD_ARG, "null argument");
    if (cfg->spatial_dims != 3) return fail(LDM_ERR_UNSUPPORTED, "only spatial_dims == 3 is implemented");
    if (cfg->num_levels < 1 || cfg->num_levels > LDM_MAX_LEVELS) return fail(LDM_ERR_BAD_ARG, "num_levels out of range");
    if (cfg->in_channels < 1 || cfg->out_channels < 1 || cfg->norm_num_groups < 1) return fail(LDM_ERR_BAD_ARG, "bad channel counts");
    for (int i = 0; i < cfg->num_levels; ++i) {
        if (cfg->num_res_blocks[i] < 1) return fail(LDM_ERR_BAD_ARG, "num_res_blocks[%d] < 1", i);
        const int hc = cfg->num_head_channels[i];
        const bool used = cfg->attention_levels[i] || i == cfg->num_levels - 1;          // the middle block always attends (MONAI)
        if (used && ((hc != 32 && hc != 64 && hc != 128 && hc != 256) || cfg->channels[i] % hc))
            return fail(LDM_ERR_UNSUPPORTED, "attention level %d: num_head_channels must be 32, 64, 128 or 256 and divide the level's "
                        "%d channels (got %d)", i, cfg->channels[i], hc);
    }
    std::unique_ptr<ldm_model> m(new ldm_model());
    m->type = 0; m->ucfg = *cfg;
    LDM_TRY(unet_register(m.get()));
    *out = m.release();
    return 0;
}

int ldm_vae_create(const ldm_vae_cfg* cfg, ldm_model** out) {
    if (!cfg || !out) return fail(LDM_ERR_BAD_ARG, "null argument");
    if (cfg->spatial_dims != 3) return fail(LDM_ERR_UNSUPPORTED, "only spatial_dims == 3 is implemented");
    if (cfg->num_levels < 1 || cfg->num_levels > LDM_MAX_LEVELS) return fail(LDM_ERR_BAD_ARG, "num_levels out of range");
    if (cfg->in_channels < 1 || cfg->out_channels < 1 || cfg->latent_channels < 1 || cfg->latent_channels > 16)
        return fail(LDM_ERR_BAD_ARG, "bad channel counts");
    std::unique_ptr<ldm_model> m(new ldm_model());
    m->type = 1; m->vcfg = *cfg;
    LDM_TRY(vae_register(m.get()));
    *out = m.release();
    return 0;
}

void ldm_model_destroy(ldm_model* m) {
    if (!m) return;
    for (auto& g : m->graphs) if (g.exec) (void)hipGraphExecDestroy(g.exec);
    if (m->cap_stream) (void)hipStreamDestroy(m->cap_stream);
    if (m->gsync.stream) (void)hipStreamDestroy(m->gsync.stream);
    if (m->gsync.stage) (void)hipFree(m->gsync.stage);
    for (auto e : m->gsync.ev) (void)hipEventDestroy(e);
    if (m->arena) (void)hipFree(m->arena);
    if (m->arena32) (void)hipFree(m->arena32);
    if (m->temb_tab) (void)hipFree(m->temb_tab);
    delete m;
}

int ldm_model_num_params(const ldm_model* m) { return m ? (int)m->params.size() : 0; }
const char* ldm_model_param_name(const ldm_model* m, int i) {
    return (m && i >= 0 && i < (int)m->params.size()) ? m->params[i].name.c_str() : nullptr;
}
int ldm_model_param_ndim(const ldm_model* m, int i) {
    return (m && i >= 0 && i < (int)m->params.size()) ? (int)m->params[i].shape.size() : -1;
}
const int64_t* ldm_model_param_shape(const ldm_model* m, int i) {
    return (m && i >= 0 && i < (int)m->params.size()) ? m->params[i].shape.data() : nullptr;
}
int64_t ldm_model_param_numel_total(const ldm_model* m) {
    int64_t t = 0; if (!m) return 0;
    for (auto& p : m->params) { int64_t n = 1; for (auto s : p.shape) n *= s; t += n; }
    return t;
}

static int ensure_arena(ldm_model* m) {
    if (m->arena) return 0;
    HIP_TRY(hipMalloc((void**)&m->arena, m->arena_bytes));
    HIP_TRY(hipMemset(m->arena, 0, m->arena_bytes));
    // the memset runs on the null stream and may still be in flight when this returns; the uploads that follow go to the
    // caller's stream, which need not synchronise with the null stream (non-blocking streams): wait here, once per model
    HIP_TRY(hipDeviceSynchronize());
    return 0;
}

static int ensure_arena32(ldm_model* m) {
    if (m->arena32) return 0;
    HIP_TRY(hipMalloc((void**)&m->arena32, 2 * m->arena_bytes + m->x3_bytes));
    HIP_TRY(hipMemset(m->arena32, 0, 2 * m->arena_bytes + m->x3_bytes));
    HIP_TRY(hipDeviceSynchronize());
    return 0;
}
// device-side re-pack of one parameter (fp32 MONAI layout) into the fp32 arena
static void pack32_device(ldm_model* m, const ParamDesc& d, const float* src, hipStream_t s) {
    if (d.kind == PK_VEC_F32) return;
    float* dst = (float*)(m->arena32 + 2 * d.dst_off);
    const bool lin = d.kind == PK_LINEAR_W;
    const int taps = lin ? 1 : d.k * d.k * d.k, cin_s = lin ? d.cin : d.cin_s, cout_pad = lin ? d.cout : d.cout_pad, row_off = lin ? 0 : d.row_off;
    const long total = (long)taps * d.cout * cin_s;
    hipLaunchKernelGGL(param_pack_f32_kernel, dim3(grid_for(total, 256, 8192)), dim3(256), 0, s, src, dst, taps, d.cout, d.cin, cin_s, cout_pad, row_off);
}

int ldm_model_load_param(ldm_model* m, const char* name, const float* src, size_t numel) {
    if (!m || !name || !src) return fail(LDM_ERR_BAD_ARG, "null argument");
    auto it = m->pindex.find(name);
    if (it == m->pindex.end()) return fail(LDM_ERR_BAD_ARG, "unknown parameter '%s'", name);
    ParamDesc& d = m->params[it->second];
    size_t expect = 1; for (auto s : d.shape) expect *= (size_t)s;
    if (expect != numel) return fail(LDM_ERR_BAD_ARG, "parameter '%s': expected %zu elements, got %zu", name, expect, numel);
    LDM_TRY(ensure_arena(m));
    if (d.kind == PK_VEC_F32) {
        HIP_TRY(hipMemcpy(m->arena + d.dst_off, src, numel * 4, hipMemcpyHostToDevice));
    } else if (d.kind == PK_LINEAR_W) {
        std::vector<uint16_t> tmp(numel);
        for (size_t k = 0; k < numel; ++k) tmp[k] = host_f2bf(src[k]);
        HIP_TRY(hipMemcpy(m->arena + d.dst_off, tmp.data(), numel * 2, hipMemcpyHostToDevice));
    } else {   // PK_CONV_W: [cout][cin][kd][kh][kw] fp32 -> [tap][cout_pad][cin_s] bf16 rows row_off..row_off+cout
        const int taps = d.k * d.k * d.k;
        std::vector<uint16_t> tmp((size_t)d.cout * d.cin_s);
        for (int t = 0; t < taps; ++t) {
            std::fill(tmp.begin(), tmp.end(), (uint16_t)0);
            for (int co = 0; co < d.cout; ++co)
                for (int ci = 0; ci < d.cin; ++ci)
                    tmp[(size_t)co * d.cin_s + ci] = host_f2bf(src[((size_t)co * d.cin + ci) * taps + t]);
            char* dst = m->arena + d.dst_off + ((size_t)t * d.cout_pad + d.row_off) * d.cin_s * 2;
            HIP_TRY(hipMemcpy(dst, tmp.data(), tmp.size() * 2, hipMemcpyHostToDevice));
        }
    }
    if (m->precision == 1 && d.kind != PK_VEC_F32) {     // the unrounded copy for the fp32 precision mode
        LDM_TRY(ensure_arena32(m));
        if (d.kind == PK_LINEAR_W) {
            HIP_TRY(hipMemcpy(m->arena32 + 2 * d.dst_off, src, numel * 4, hipMemcpyHostToDevice));
        } else {
            const int taps = d.k * d.k * d.k;
            std::vector<float> tmp((size_t)d.cout * d.cin_s);
            for (int t = 0; t < taps; ++t) {
                std::fill(tmp.begin(), tmp.end(), 0.f);
                for (int co = 0; co < d.cout; ++co)
                    for (int ci = 0; ci < d.cin; ++ci)
                        tmp[(size_t)co * d.cin_s + ci] = src[((size_t)co * d.cin + ci) * taps + t];
                char* dst = m->arena32 + 2 * (d.dst_off + ((size_t)t * d.cout_pad + d.row_off) * d.cin_s * 2);
                HIP_TRY(hipMemcpy(dst, tmp.data(), tmp.size() * 4, hipMemcpyHostToDevice));
            }
        }
    }
    if (!d.loaded) { d.loaded = true; m->loaded_count++; }
    m->derived_dirty = true; m->temb_tab_valid = false;
    return 0;
}

// phase weights of the upsample convs, rebuilt (stream-ordered) after a parameter upload: called by the inference entries
static int ensure_derived(ldm_model* m, hipStream_t s) {
    if (!m->derived_dirty) return 0;
    for (const auto& iw : m->im2col_ws)
        hipLaunchKernelGGL(im2col_weights_kernel, dim3((unsigned)((iw.cout_pad * iw.Kp + 255) / 256)), dim3(256), 0, s,
                           (const bf16_t*)(m->arena + iw.w_off), (bf16_t*)(m->arena + iw.wi_off), iw.cout_pad, iw.cin_s, iw.cin, iw.Kp);
    for (const auto& pw : m->phase_ws) {
        const long vecs = (long)pw.cout_pad * pw.cin_s / 8;
        hipLaunchKernelGGL(phase_weights_kernel, dim3((unsigned)((vecs + 255) / 256), 64), dim3(256), 0, s,
                           (const bf16_t*)(m->arena + pw.w_off), (bf16_t*)(m->arena + pw.wp_off), pw.cout_pad, pw.cin_s);
    }
    if (m->precision == 1 && m->arena32)
        for (const auto& xw : m->x3p_ws)
            hipLaunchKernelGGL(x3_phase_weights_kernel, dim3((unsigned)(((long)xw.cout_pad * xw.cin_s / 4 + 255) / 256), 64), dim3(256), 0, s,
                               (const float*)(m->arena32 + 2 * xw.w_off), (bf16_t*)(m->arena32 + 2 * m->arena_bytes + xw.x3p_off), xw.cout_pad, xw.cin_s);
    if (m->precision == 1 && m->arena32)
        for (const auto& xw : m->x3_ws)
            hipLaunchKernelGGL(x3_weights_kernel, dim3(grid_for(xw.rows * (xw.cin_s / 4), 256, 2048)), dim3(256), 0, s,
                               (const float*)(m->arena32 + 2 * xw.w_off), (bf16_t*)(m->arena32 + 2 * m->arena_bytes + xw.x3_off), xw.rows, xw.cin_s);
    HIP_TRY(hipGetLastError());
    m->derived_dirty = false;
    return 0;
}

static int get_plan(ldm_model* m, const char* kind, int B, int D, int H, int W, std::shared_ptr<Plan>* out, int tap_mode = 0) {
    if (B < 1 || D < 1 || H < 1 || W < 1 || D > 255 * 8 || H > 255 * 8 || W > 255 * 8) return fail(LDM_ERR_BAD_ARG, "bad shape");
    const bool train = kind[0] == 't';
    const bool hp = m->precision == 1;
    char key[96]; snprintf(key, sizeof key, "%s:%d:%d:%d:%d:p%d:t%d", kind, B, D, H, W, hp ? 1 : 0, tap_mode);
    auto it = m->plans.find(key);
    if (it != m->plans.end()) { *out = it->second; return 0; }
    std::shared_ptr<Plan> p(new Plan());
    if (m->type == 0) LDM_TRY(unet_build(m, B, D, H, W, p.get(), train, hp, tap_mode, strcmp(kind, "unet_tab") == 0));
    else if (train) LDM_TRY(vae_build_train(m, B, D, H, W, p.get(), hp));
    else if (kind[0] == 'e') LDM_TRY(vae_build_encode(m, B, D, H, W, p.get(), hp, tap_mode));
    else LDM_TRY(vae_build_decode(m, B, D, H, W, p.get(), hp, tap_mode));
    m->plans[key] = p; *out = p;
    return 0;
}

static int check_ready(ldm_model* m, const void* ws, size_t ws_bytes, const Plan& p) {
    if (m->loaded_count != (int)m->params.size()) {
        for (auto& d : m->params) if (!d.loaded)
            return fail(LDM_ERR_NOT_LOADED, "parameter '%s' (and %d others) not uploaded", d.name.c_str(),
                        (int)m->params.size() - m->loaded_count - 1);
    }
    if (!ws || ws_bytes < p.ws_bytes) return fail(LDM_ERR_WORKSPACE, "workspace too small: need %zu bytes, got %zu", p.ws_bytes, ws_bytes);
    if (((uintptr_t)ws) & 255) return fail(LDM_ERR_BAD_ARG, "workspace must be 256-byte aligned");
    if (p.wt_tab.ready() || p.exp_tab.ready() || p.cs_tab.ready()) return fail(LDM_ERR_HIP, "descriptor table upload failed");
    return 0;
}

size_t ldm_unet_workspace_bytes(ldm_model* m, int B, int D, int H, int W) {
    if (!m || m->type != 0) { fail(LDM_ERR_BAD_ARG, "not a UNet handle"); return 0; }
    std::shared_ptr<Plan> p; if (get_plan(m, "unet", B, D, H, W, &p)) return 0;
    return p->ws_bytes;
}

// ---- device-resident sampler (fused scheduler step with in-kernel Philox noise) ----------------------------------------------
static std::atomic<uint64_t> g_sampler_uid{0};
struct ldm_sampler {
    float* coef = nullptr; SamplerState* st = nullptr; int n_steps = 0, kind = 0, clip = 1; unsigned seed_lo = 0, seed_hi = 0;
    uint64_t uid = ++g_sampler_uid;                  // never reused (graph-replay cache key)
    std::vector<float> ts;                           // host copy of the schedule's timesteps in sampling order (key of the model's time-embedding table)
};
static int sampler_launch(ldm_sampler* sp, const float* eps, float* x, float* x0_out, int64_t n, float* tbuf, int B, hipStream_t s) {
    SamplerParams p{}; p.coef = sp->coef; p.st = sp->st; p.n_steps = sp->n_steps; p.kind = sp->kind; p.clip = sp->clip;
    p.seed_lo = sp->seed_lo; p.seed_hi = sp->seed_hi; p.eps = eps; p.x = x; p.x0_out = x0_out; p.n = (long)n; p.tbuf = tbuf; p.B = B;
    hipLaunchKernelGGL(sampler_step_kernel, dim3(grid_for((n + 3) / 4, 256, 1024)), dim3(256), 0, s, p);
    return 0;
}

// The stacked time_emb_proj outputs for every step of `sp`'s schedule, [n_steps][tproj_rows] fp32, built by the forward plan's own
// kernels (sinusoid -> Linear -> SiLU -> Linear -> SiLU -> stacked projections) at batch n_steps, so row k is bit-identical to what
// the plan computes for t = ts[k].  Rebuilt after a parameter upload / optimizer step, a precision switch or for another schedule;
// a table that moves in memory invalidates the model's cached graphs (they hold its address).  LDM_TEMB_TABLE=0: never used.
static bool temb_table_enabled() { static const int v = ldm_knob("LDM_TEMB_TABLE", 1); return v != 0; }
static int ensure_temb_table(ldm_model* m, const ldm_sampler* sp, hipStream_t s) {
    if (m->temb_tab_valid && m->temb_tab_prec == m->precision && m->temb_tab_ts == sp->ts) return 0;
    const int n = sp->n_steps, rows = m->tproj_rows, c0 = m->ucfg.channels[0], temb = 4 * c0;
    const bool hp = m->precision == 1;
    if (hp && !m->arena32) return fail(LDM_ERR_NOT_LOADED, "fp32 precision: the fp32 weight arena is empty");
    const size_t need = (size_t)n * rows * 4;
    if (need > m->temb_tab_cap) {
        HIP_TRY(hipDeviceSynchronize());                                   // a replaying graph may still read the old table
        for (auto& g : m->graphs) if (g.exec) (void)hipGraphExecDestroy(g.exec);
        m->graphs.clear();
        if (m->temb_tab) (void)hipFree(m->temb_tab);
        m->temb_tab = nullptr; m->temb_tab_cap = 0;
        HIP_TRY(hipMalloc((void**)&m->temb_tab, need));
        m->temb_tab_cap = need;
    }
    float* tmp = nullptr;                                                  // t [n] | sinusoid [n][c0] | e1 [n][temb] | e2 [n][temb]
    HIP_TRY(hipMalloc((void**)&tmp, (size_t)n * (1 + c0 + 2 * temb) * 4));
    float* d_t = tmp; float* d_sin = tmp + n; float* d_e1 = d_sin + (size_t)n * c0; float* d_e2 = d_e1 + (size_t)n * temb;
    hipError_t e = hipMemcpyAsync(d_t, sp->ts.data(), (size_t)n * 4, hipMemcpyHostToDevice, s);
    if (e == hipSuccess) {
        const LinW& l0 = m->lins.at("time_embed.0"); const LinW& l2 = m->lins.at("time_embed.2");
        hipLaunchKernelGGL(temb_sinusoid_kernel, dim3(grid_for((long)n * c0)), dim3(256), 0, s, (const float*)d_t, d_sin, n, c0);
        auto gemv = [&](size_t w_off, size_t b_off, const float* xin, float* y, int I, int O, int silu) {
            if (hp) hipLaunchKernelGGL(gemv_f32_kernel, dim3((O + 3) / 4, n), dim3(256), 0, s, (const float*)(m->arena32 + 2 * w_off),
                                       (const float*)(m->arena + b_off), xin, y, I, O, I, O, silu);
            else hipLaunchKernelGGL(gemv_bf16_kernel, dim3((O + 3) / 4, n), dim3(256), 0, s, (const bf16_t*)(m->arena + w_off),
                                    (const float*)(m->arena + b_off), xin, y, I, O, I, O, silu);
        };
        gemv(l0.w_off, l0.b_off, d_sin, d_e1, c0, temb, 0);
        gemv(l2.w_off, l2.b_off, d_e1, d_e2, temb, temb, 1);
        gemv(m->tproj_w_off, m->tproj_b_off, d_e2, m->temb_tab, temb, rows, 1);
        e = hipGetLastError();
        if (e == hipSuccess) e = hipStreamSynchronize(s);                  // the temporaries go away below; one-off per upload
    }
    (void)hipFree(tmp);
    if (e != hipSuccess) return fail(LDM_ERR_HIP, "time-embedding table: %s", hipGetErrorString(e));
    m->temb_tab_ts = sp->ts; m->temb_tab_prec = m->precision; m->temb_tab_valid = true;
    return 0;
}

static int unet_forward_impl(ldm_model* m, const float* x, int x_channels, const float* cond, int cond_channels,
                             const float* timesteps, float* out, int B, int D, int H, int W,
                             void* workspace, size_t workspace_bytes, void* stream, ldm_sampler* sp, float* x_inout) {
    if (!m || m->type != 0) return fail(LDM_ERR_BAD_ARG, "not a UNet handle");
    if (!x || !timesteps || !out) return fail(LDM_ERR_BAD_ARG, "null tensor argument");
    if (!cond) cond_channels = 0;
    // a denoising step driven by a sampler knows its timestep as a step INDEX on the device: the time-embedding chain becomes one row copy
    const bool tab = sp != nullptr && temb_table_enabled() && m->tproj_rows % 4 == 0;
    std::shared_ptr<Plan> p; LDM_TRY(get_plan(m, tab ? "unet_tab" : "unet", B, D, H, W, &p));
    LDM_TRY(check_ready(m, workspace, workspace_bytes, *p));
    if (tab) LDM_TRY(ensure_temb_table(m, sp, (hipStream_t)stream));
    Bases bs{}; bs.p[BASE_WS] = (char*)workspace; bs.p[BASE_W] = m->arena; bs.p[BASE_W32] = m->arena32;
    bs.p[BASE_IO0] = (char*)x; bs.p[BASE_IO1] = (char*)cond; bs.p[BASE_IO2] = (char*)timesteps; bs.p[BASE_IO3] = (char*)out;
    if (tab) { bs.p[BASE_TTAB] = (char*)m->temb_tab; bs.p[BASE_SST] = (char*)sp->st; bs.sampler_steps = sp->n_steps; }
    const int rt[2] = {x_channels, cond_channels};
    LDM_TRY(ensure_derived(m, (hipStream_t)stream));
    LaneCtx lanes;
    const int64_t n_out = (int64_t)B * m->ucfg.out_channels * D * H * W;
    // one denoising step = the forward plan, then (with a sampler) the fused scheduler step on the same stream
    auto run_all = [&](hipStream_t s) -> int {
        LDM_TRY(run_plan(*p, bs, rt, s, 0, (size_t)-1, lanes));
        if (sp) LDM_TRY(sampler_launch(sp, out, x_inout, nullptr, n_out, (float*)timesteps, B, s));
        return 0;
    };
    if (!m->graph_mode || g_prof.on || g_plan_trace.on) return run_all((hipStream_t)stream);
    // ---- graph replay: same launches, recorded once per pointer set
    const void* key[8] = {x, cond, timesteps, out, workspace, stream, sp, x_inout};
    const uint64_t suid = sp ? sp->uid : 0;
    ldm_model::GraphEntry* ge = nullptr;
    for (size_t k = 0; k < m->graphs.size();) {          // entries recorded for a sampler that has since been destroyed at this address
        auto& g = m->graphs[k];
        if (sp && g.ptr[6] == (const void*)sp && g.sampler_uid != suid) {
            if (g.exec) (void)hipGraphExecDestroy(g.exec);
            m->graphs.erase(m->graphs.begin() + k);
        } else ++k;
    }
    for (auto& g : m->graphs)
        if (g.plan == p.get() && !memcmp(g.ptr, key, sizeof key) && g.rt[0] == rt[0] && g.rt[1] == rt[1] && g.sampler_uid == suid) { ge = &g; break; }
    if (!ge) {
        if (m->graphs.size() >= 16) {                    // bounded cache: drop the oldest entry
            if (m->graphs.front().exec) (void)hipGraphExecDestroy(m->graphs.front().exec);
            m->graphs.erase(m->graphs.begin());
        }
        ldm_model::GraphEntry g{}; g.plan = p.get(); memcpy(g.ptr, key, sizeof key); g.rt[0] = rt[0]; g.rt[1] = rt[1]; g.sampler_uid = suid;
        m->graphs.push_back(g); ge = &m->graphs.back();
    }
    if (ge->exec) { HIP_TRY(hipGraphLaunch(ge->exec, (hipStream_t)stream)); return 0; }
    if (ge->seen++ == 0) return run_all((hipStream_t)stream);   // first sight: eager (also warms one-time set-up)
    hipGraph_t graph = nullptr;
    if (!m->cap_stream) HIP_TRY(hipStreamCreateWithFlags(&m->cap_stream, hipStreamNonBlocking));
    HIP_TRY(hipStreamBeginCapture(m->cap_stream, hipStreamCaptureModeThreadLocal));
    const int rc = run_all(m->cap_stream);
    const hipError_t ec = hipStreamEndCapture(m->cap_stream, &graph);
    if (rc) { if (graph) (void)hipGraphDestroy(graph); return rc; }
    if (ec != hipSuccess || !graph) return fail(LDM_ERR_HIP, "stream capture failed: %s", hipGetErrorString(ec));
    HIP_TRY(hipGraphInstantiate(&ge->exec, graph, nullptr, nullptr, 0));
    (void)hipGraphDestroy(graph);
    HIP_TRY(hipGraphLaunch(ge->exec, (hipStream_t)stream));
    return 0;
}

int ldm_unet_forward(ldm_model* m, const float* x, int x_channels, const float* cond, int cond_channels,
                     const float* timesteps, float* out, int B, int D, int H, int W,
                     void* workspace, size_t workspace_bytes, void* stream) {
    return unet_forward_impl(m, x, x_channels, cond, cond_channels, timesteps, out, B, D, H, W, workspace, workspace_bytes, stream, nullptr, nullptr);
}

/* Sampler: coef_host = [n_steps][6] fp32 rows {1/sqrt(abar_t), sqrt(1 - abar_t), c0, c1 (DDPM: coefficient of x_t | DDIM: direction
 * coefficient of eps), sigma, t} in sampling order (the host mirror computes them exactly as MONAI does and as DDPMScheduler.step /
 * DDIMScheduler.step pass them by value); kind 0 = DDPM, 1 = DDIM.  Noise: Philox4x32-10(counter = (element quad, step), key = seed). */
int ldm_sampler_create(const float* coef_host, int n_steps, int kind, int clip, uint64_t seed, ldm_sampler** out) {
    if (!coef_host || n_steps < 1 || kind < 0 || kind > 1 || !out) return fail(LDM_ERR_BAD_ARG, "bad argument");
    std::unique_ptr<ldm_sampler> sp(new ldm_sampler());
    std::vector<float> rows((size_t)n_steps * 8, 0.f);
    for (int k = 0; k < n_steps; ++k) for (int j = 0; j < 6; ++j) rows[(size_t)k * 8 + j] = coef_host[(size_t)k * 6 + j];
    HIP_TRY(hipMalloc((void**)&sp->coef, rows.size() * 4));
    HIP_TRY(hipMemcpy(sp->coef, rows.data(), rows.size() * 4, hipMemcpyHostToDevice));
    HIP_TRY(hipMalloc((void**)&sp->st, 256));
    HIP_TRY(hipMemset(sp->st, 0, 256));
    HIP_TRY(hipDeviceSynchronize());
    sp->n_steps = n_steps; sp->kind = kind; sp->clip = clip ? 1 : 0; sp->seed_lo = (unsigned)seed; sp->seed_hi = (unsigned)(seed >> 32);
    sp->ts.resize(n_steps); for (int k = 0; k < n_steps; ++k) sp->ts[k] = coef_host[(size_t)k * 6 + 5];
    *out = sp.release();
    return 0;
}
void ldm_sampler_destroy(ldm_sampler* sp) {
    if (!sp) return;
    if (sp->coef) (void)hipFree(sp->coef);
    if (sp->st) (void)hipFree(sp->st);
    delete sp;
}
/* step counter := 0, tbuf[0..B) := t of the first step */
int ldm_sampler_reset(ldm_sampler* sp, float* tbuf, int B, void* stream) {
    if (!sp || !tbuf || B < 1) return fail(LDM_ERR_BAD_ARG, "bad argument");
    hipLaunchKernelGGL(sampler_reset_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, sp->st, (const float*)sp->coef, tbuf, B);
    HIP_TRY(hipGetLastError());
    return 0;
}
/* x := scheduler step (x, eps) for the step the device counter points at (in place), x0_out (optional) := x0_hat; then the counter
 * advances and tbuf[0..B) receives the next step's t.  Calls beyond n_steps leave x unchanged. */
int ldm_sampler_step(ldm_sampler* sp, const float* eps, float* x, float* x0_out, int64_t n, float* tbuf, int B, void* stream) {
    if (!sp || !eps || !x || !tbuf || n < 0 || B < 1) return fail(LDM_ERR_BAD_ARG, "bad argument");
    SamplerParams p{}; p.coef = sp->coef; p.st = sp->st; p.n_steps = sp->n_steps; p.kind = sp->kind; p.clip = sp->clip;
    p.seed_lo = sp->seed_lo; p.seed_hi = sp->seed_hi; p.eps = eps; p.x = x; p.x0_out = x0_out; p.n = (long)n; p.tbuf = tbuf; p.B = B;
    hipLaunchKernelGGL(sampler_step_kernel, dim3(grid_for((n + 3) / 4, 256, 1024)), dim3(256), 0, (hipStream_t)stream, p);
    HIP_TRY(hipGetLastError());
    return 0;
}
/* the N(0, 1) draws step `step` uses for a tensor of n elements (tests: statistics, reproducibility, fused == unfused) */
int ldm_sampler_noise(const ldm_sampler* sp, int step, float* out, int64_t n, void* stream) {
    if (!sp || !out || n < 0 || step < 0) return fail(LDM_ERR_BAD_ARG, "bad argument");
    hipLaunchKernelGGL(sampler_noise_kernel, dim3(grid_for((n + 3) / 4, 256, 1024)), dim3(256), 0, (hipStream_t)stream, out, (long)n, step, sp->seed_lo, sp->seed_hi);
    HIP_TRY(hipGetLastError());
    return 0;
}
/* One whole denoising step: eps_hat = UNet(x, tbuf) into eps_scratch, then ldm_sampler_step(x in place).  With
 * ldm_model_set_graph_mode the forward plan AND the scheduler step replay as ONE hipGraphLaunch: no host-side per-step work at all
 * (3d_ldm/inference.py:94-99's loop body). */
int ldm_unet_denoise_step(ldm_model* m, ldm_sampler* sp, float* x, int x_channels, const float* cond, int cond_channels,
                          float* tbuf, float* eps_scratch, int B, int D, int H, int W,
                          void* workspace, size_t workspace_bytes, void* stream) {
    if (!sp) return fail(LDM_ERR_BAD_ARG, "null sampler");
    if (m && m->type == 0 && x_channels != m->ucfg.out_channels) return fail(LDM_ERR_BAD_ARG, "x must have the UNet's out_channels (%d)", m->ucfg.out_channels);
    return unet_forward_impl(m, x, x_channels, cond, cond_channels, tbuf, eps_scratch, B, D, H, W, workspace, workspace_bytes, stream, sp, x);
}

/* Diagnostic builds only (EXTRA=-DLDM_KSTAMPS, tools/kstamps.py): the in-kernel stamps of the instrumented kernels, [entries][8] =
 * {kernel id, entry, stamp 1 ... stamp 6} in 10 ns ticks, in launch order; reset != 0 clears the log afterwards.  Returns the number
 * of entries (0 in the product library, which carries no stamps). */
int ldm_debug_kstamps(unsigned long long* out, int max_entries, int reset) {
#ifdef LDM_KSTAMPS
    if (!out || max_entries < 0) return fail(LDM_ERR_BAD_ARG, "bad argument");
    HIP_TRY(hipDeviceSynchronize());
    unsigned n = 0;
    HIP_TRY(hipMemcpyFromSymbol(&n, HIP_SYMBOL(g_kstamp_seq), sizeof n));
    const int cnt = (int)std::min<unsigned>(n, 8192u);
    const int take = std::min(cnt, max_entries);
    if (take > 0) HIP_TRY(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_kstamp), (size_t)take * 8 * sizeof(unsigned long long)));
    if (reset) { const unsigned z = 0; HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_kstamp_seq), &z, sizeof z)); }
    return take;
#else
    (void)out; (void)max_entries; (void)reset;
    return 0;
#endif
}

/* Per-op timeline of every launch plan that runs from now on: a HIP event in front of every op, one CSV row per op appended to
 * `path` when the call returns (ops of the plan, op index, op kind, microseconds, shape); NULL or "" switches it off.  Tracing
 * synchronises at the end of every call and bypasses graph replay: a measurement aid, the hook behind the harness's --profile
 * (3d_ldm/train_autoencoder.py:81,312-329 wraps a few steps in torch.profiler).  Initial state: the LDM_PLAN_TRACE environment variable. */
int ldm_set_plan_trace(const char* path) {
    g_plan_trace.on = path && *path;
    g_plan_trace.path = g_plan_trace.on ? path : "";
    return 0;
}

/* on != 0: ldm_unet_forward replays a HIP graph of its launch plan whenever it sees the same (x, cond, timesteps, out,
 * workspace, stream) pointers again (callers keep those buffers fixed: the Python shell stages through persistent tensors).
 * Same kernels, same results; only the host cost per step changes (one graph launch instead of ~150 launches). */
int ldm_model_set_graph_mode(ldm_model* m, int on) {
    if (!m) return fail(LDM_ERR_BAD_ARG, "null model");
    m->graph_mode = on ? 1 : 0;
    if (!on) { for (auto& g : m->graphs) if (g.exec) (void)hipGraphExecDestroy(g.exec); m->graphs.clear(); }
    return 0;
}

/* Arithmetic of the INFERENCE plans (ldm_unet_forward, ldm_vae_encode, ldm_vae_decode and their *_taps forms):
 *   LDM_PREC_BF16 (0, default): bf16 storage, bf16 MFMA, fp32 accumulation (the headline path);
 *   LDM_PREC_FP32 (1): fp32 activations and weights on the fp32 matrix instruction (f32_path.h): the reference's own
 *   arithmetic (3d_ldm/train_diffusion.py:177: autocast off), within ~1e-5 rel-L2 of the CPU path instead of ~3e-2.
 * Switching to fp32 needs the parameters uploaded again (the unrounded copies are made at upload time): every ldm_model_load_*
 * call after the switch fills both arenas; forward calls fail with LDM_ERR_NOT_LOADED until then. */
int ldm_model_set_precision(ldm_model* m, int precision) {
    if (!m) return fail(LDM_ERR_BAD_ARG, "null model");
    if (precision != 0 && precision != 1) return fail(LDM_ERR_BAD_ARG, "precision must be LDM_PREC_BF16 (0) or LDM_PREC_FP32 (1)");
    if (precision == 1 && m->precision != 1) {          // every matrix must be uploaded again
        for (ParamDesc& d : m->params) d.loaded = false;
        m->loaded_count = 0;
    }
    m->precision = precision;
    return 0;
}
int ldm_model_get_precision(const ldm_model* m) { return m ? m->precision : -1; }

/* Debug taps: kind = "unet" | "enc" | "dec".  ldm_model_tap_count builds (and caches) the tapped plan of that shape and returns
 * the number of taps (or a negative status); ldm_model_tap_info describes tap i: name (block prefix in the MONAI state_dict),
 * dims = {B, C, D, H, W} and the element offset of its fp32 NCDHW tensor inside the taps_out / taps_in buffers of the
 * *_taps entry points (total elements = ldm_model_tap_elems). */
static const char* tap_kind(ldm_model* m, const char* kind) {
    if (!kind) return nullptr;
    if (m->type == 0) return strcmp(kind, "unet") == 0 ? "unet" : nullptr;
    return strcmp(kind, "enc") == 0 ? "enc" : strcmp(kind, "dec") == 0 ? "dec" : nullptr;
}
int ldm_model_tap_count(ldm_model* m, const char* kind, int B, int D, int H, int W) {
    if (!m || !tap_kind(m, kind)) return fail(LDM_ERR_BAD_ARG, "bad model / plan kind");
    std::shared_ptr<Plan> p; LDM_TRY(get_plan(m, tap_kind(m, kind), B, D, H, W, &p, 1));
    return (int)p->taps.size();
}
int64_t ldm_model_tap_elems(ldm_model* m, const char* kind, int B, int D, int H, int W) {
    if (!m || !tap_kind(m, kind)) return fail(LDM_ERR_BAD_ARG, "bad model / plan kind");
    std::shared_ptr<Plan> p; LDM_TRY(get_plan(m, tap_kind(m, kind), B, D, H, W, &p, 1));
    return (int64_t)p->tap_elems;
}
int ldm_model_tap_info(ldm_model* m, const char* kind, int B, int D, int H, int W, int i, char* name, int name_cap, int dims[5], int64_t* offset) {
    if (!m || !tap_kind(m, kind) || !name || name_cap < 1 || !dims || !offset) return fail(LDM_ERR_BAD_ARG, "bad argument");
    std::shared_ptr<Plan> p; LDM_TRY(get_plan(m, tap_kind(m, kind), B, D, H, W, &p, 1));
    if (i < 0 || i >= (int)p->taps.size()) return fail(LDM_ERR_BAD_ARG, "tap index out of range");
    const TapInfo& t = p->taps[i];
    snprintf(name, (size_t)name_cap, "%s", t.name.c_str());
    for (int k = 0; k < 5; ++k) dims[k] = t.dims[k];
    *offset = (int64_t)t.off;
    return 0;
}
/* ldm_unet_forward that also writes every block output to taps_out (fp32 NCDHW, layout from ldm_model_tap_info) and, when
 * taps_in is given, then OVERWRITES each block output with the caller's tensor, so that every block is computed from the
 * reference's input (teacher forcing: per-block parity without compounding rounding noise).  Eager launches only. */
int ldm_unet_forward_taps(ldm_model* m, const float* x, int x_channels, const float* cond, int cond_channels,
                          const float* timesteps, float* out, int B, int D, int H, int W, float* taps_out, const float* taps_in,
                          void* workspace, size_t workspace_bytes, void* stream) {
    if (!m || m->type != 0) return fail(LDM_ERR_BAD_ARG, "not a UNet handle");
    if (!x || !timesteps || !out || !taps_out) return fail(LDM_ERR_BAD_ARG, "null tensor argument");
    if (!cond) cond_channels = 0;
    std::shared_ptr<Plan> p; LDM_TRY(get_plan(m, "unet", B, D, H, W, &p, taps_in ? 2 : 1));
    LDM_TRY(check_ready(m, workspace, workspace_bytes, *p));
    Bases bs{}; bs.p[BASE_WS] = (char*)workspace; bs.p[BASE_W] = m->arena; bs.p[BASE_W32] = m->arena32;
    bs.p[BASE_IO0] = (char*)x; bs.p[BASE_IO1] = (char*)cond; bs.p[BASE_IO2] = (char*)timesteps; bs.p[BASE_IO3] = (char*)out;
    bs.p[BASE_TAPO] = (char*)taps_out; bs.p[BASE_TAPI] = (char*)taps_in;
    const int rt[2] = {x_channels, cond_channels};
    LDM_TRY(ensure_derived(m, (hipStream_t)stream));
    return run_plan(*p, bs, rt, (hipStream_t)stream);
}
size_t ldm_model_taps_workspace_bytes(ldm_model* m, const char* kind, int B, int D, int H, int W, int force) {
    if (!m || !tap_kind(m, kind)) { fail(LDM_ERR_BAD_ARG, "bad model / plan kind"); return 0; }
    std::shared_ptr<Plan> p; if (get_plan(m, tap_kind(m, kind), B, D, H, W, &p, force ? 2 : 1)) return 0;
    return p->ws_bytes;
}

// ---- training: forward that keeps the tape, backward into one flat fp32 gradient buffer --------------------------
int64_t ldm_model_param_offset(const ldm_model* m, int i) {
    return (m && i >= 0 && i < (int)m->params.size()) ? m->params[i].flat_off : -1;
}

/* Re-pack every parameter from device fp32 tensors (MONAI layouts, library parameter order) into the bf16 arena:
 * what an optimizer step is followed by.  ptrs is a HOST array of n device pointers. */
int ldm_model_load_params_device(ldm_model* m, const float* const* ptrs, int n, void* stream) {
    if (!m || !ptrs) return fail(LDM_ERR_BAD_ARG, "null argument");
    if (n != (int)m->params.size()) return fail(LDM_ERR_BAD_ARG, "expected %d parameter pointers, got %d", (int)m->params.size(), n);
    LDM_TRY(ensure_arena(m));
    hipStream_t s = (hipStream_t)stream;
    for (int k = 0; k < n; ++k) {
        ParamDesc& d = m->params[k];
        if (!ptrs[k]) return fail(LDM_ERR_BAD_ARG, "parameter '%s': null pointer", d.name.c_str());
        if (d.kind == PK_VEC_F32) {
            HIP_TRY(hipMemcpyAsync(m->arena + d.dst_off, ptrs[k], (size_t)d.cout * 4, hipMemcpyDeviceToDevice, s));
        } else if (d.kind == PK_LINEAR_W) {
            hipLaunchKernelGGL(param_pack_kernel, dim3((d.cin + 63) / 64, d.cout), dim3(256), 0, s, ptrs[k], (bf16_t*)(m->arena + d.dst_off),
                               1, d.cout, d.cin, d.cin, d.cout, 0);
        } else {
            hipLaunchKernelGGL(param_pack_kernel, dim3((d.cin_s + 63) / 64, d.cout), dim3(256), 0, s, ptrs[k], (bf16_t*)(m->arena + d.dst_off),
                               d.k * d.k * d.k, d.cout, d.cin, d.cin_s, d.cout_pad, d.row_off);
        }
        if (m->precision == 1) { LDM_TRY(ensure_arena32(m)); pack32_device(m, d, ptrs[k], s); }
        if (!d.loaded) { d.loaded = true; m->loaded_count++; }
    }
    HIP_TRY(hipGetLastError());
    m->derived_dirty = true; m->temb_tab_valid = false;
    return 0;
}

/* The same from ONE flat fp32 device buffer holding every parameter at ldm_model_param_offset(i) (the layout of the flat
 * gradient buffer): a single descriptor-driven launch. */
static int ensure_pack_tab(ldm_model* m) {
    if (!m->pack_tab.nblocks) {
        std::vector<PackDesc> descs; std::vector<int2> map;
        for (const ParamDesc& d : m->params) {
            PackDesc e{}; e.src_off = (long)d.flat_off; e.dst_off = (long)d.dst_off;
            int nb;
            if (d.kind == PK_VEC_F32) { e.kind = 1; e.cout = d.cout; nb = (d.cout + 1023) / 1024; }
            else if (d.kind == PK_LINEAR_W) { e.kind = 0; e.taps = 1; e.cout = d.cout; e.cin = d.cin; e.cin_s = d.cin; e.cout_pad = d.cout; e.row_off = 0; nb = ((d.cin + 63) / 64) * d.cout; }
            else { e.kind = 0; e.taps = d.k * d.k * d.k; e.cout = d.cout; e.cin = d.cin; e.cin_s = d.cin_s; e.cout_pad = d.cout_pad; e.row_off = d.row_off; nb = ((d.cin_s + 63) / 64) * d.cout; }
            for (int b = 0; b < nb; ++b) map.push_back(make_int2((int)descs.size(), b));
            descs.push_back(e);
        }
        if (m->pack_tab.upload(descs, map) || m->pack_tab.ready()) return fail(LDM_ERR_HIP, "descriptor table upload failed");
    }
    return 0;
}

int ldm_model_load_params_flat(ldm_model* m, const float* flat, void* stream) {
    if (!m || !flat) return fail(LDM_ERR_BAD_ARG, "null argument");
    LDM_TRY(ensure_arena(m));
    LDM_TRY(ensure_pack_tab(m));
    hipLaunchKernelGGL(param_pack_batched_kernel, dim3(m->pack_tab.nblocks), dim3(256), 0, (hipStream_t)stream,
                       (const PackDesc*)m->pack_tab.descs, (const int2*)m->pack_tab.map, flat, m->arena);
    if (m->precision == 1) {
        LDM_TRY(ensure_arena32(m));
        for (const ParamDesc& d : m->params) pack32_device(m, d, flat + d.flat_off, (hipStream_t)stream);
    }
    HIP_TRY(hipGetLastError());
    for (ParamDesc& d : m->params) if (!d.loaded) { d.loaded = true; m->loaded_count++; }
    m->derived_dirty = true; m->temb_tab_valid = false;
    return 0;
}

size_t ldm_unet_train_workspace_bytes(ldm_model* m, int B, int D, int H, int W) {
    if (!m || m->type != 0) { fail(LDM_ERR_BAD_ARG, "not a UNet handle"); return 0; }
    std::shared_ptr<Plan> p; if (get_plan(m, "train", B, D, H, W, &p)) return 0;
    return p->ws_bytes;
}

/* Same result as ldm_unet_forward; additionally leaves every activation, GroupNorm statistic and attention
 * log-sum-exp row in `workspace`, which must reach ldm_unet_train_backward untouched. */
int ldm_unet_train_forward(ldm_model* m, const float* x, int x_channels, const float* cond, int cond_channels,
                           const float* timesteps, float* out, int B, int D, int H, int W,
                           void* workspace, size_t workspace_bytes, void* stream) {
    if (!m || m->type != 0) return fail(LDM_ERR_BAD_ARG, "not a UNet handle");
    if (!x || !timesteps || !out) return fail(LDM_ERR_BAD_ARG, "null tensor argument");
    if (!cond) cond_channels = 0;
    std::shared_ptr<Plan> p; LDM_TRY(get_plan(m, "train", B, D, H, W, &p));
    LDM_TRY(check_ready(m, workspace, workspace_bytes, *p));
    Bases bs{}; bs.p[BASE_WS] = (char*)workspace; bs.p[BASE_W] = m->arena; bs.p[BASE_W32] = m->arena32;
    bs.p[BASE_IO0] = (char*)x; bs.p[BASE_IO1] = (char*)cond; bs.p[BASE_IO2] = (char*)timesteps; bs.p[BASE_IO3] = (char*)out;
    const int rt[2] = {x_channels, cond_channels};
    return run_plan(*p, bs, rt, (hipStream_t)stream, 0, p->bwd_begin);
}

/* grad_out: fp32 [B][out_channels][D][H][W] (dLoss/d eps_hat).  flat_grads: fp32 [ldm_model_param_numel_total], every
 * element is overwritten with dLoss/dparam (parameter i at ldm_model_param_offset(i), its MONAI tensor layout). */
int ldm_unet_train_backward(ldm_model* m, const float* grad_out, float* flat_grads, int B, int D, int H, int W,
                            void* workspace, size_t workspace_bytes, void* stream) {
    if (!m || m->type != 0) return fail(LDM_ERR_BAD_ARG, "not a UNet handle");
    if (!grad_out || !flat_grads) return fail(LDM_ERR_BAD_ARG, "null tensor argument");
    std::shared_ptr<Plan> p; LDM_TRY(get_plan(m, "train", B, D, H, W, &p));
    LDM_TRY(check_ready(m, workspace, workspace_bytes, *p));
    Bases bs{}; bs.p[BASE_WS] = (char*)workspace; bs.p[BASE_W] = m->arena; bs.p[BASE_W32] = m->arena32;
    bs.p[BASE_IO0] = (char*)grad_out; bs.p[BASE_IO4] = (char*)flat_grads;
    const int rt[2] = {m->ucfg.out_channels, 0};
    LaneCtx lanes;
    if (m->gsync.comm) { lanes.sync = &m->gsync; LDM_TRY(grad_sync_begin(m->gsync, (hipStream_t)stream)); }
    return run_plan(*p, bs, rt, (hipStream_t)stream, p->bwd_begin, p->ops.size(), lanes);
}

size_t ldm_vae_train_workspace_bytes(ldm_model* m, int B, int D, int H, int W) {
    if (!m || m->type != 1) { fail(LDM_ERR_BAD_ARG, "not an AutoencoderKL handle"); return 0; }
    std::shared_ptr<Plan> p; if (get_plan(m, "train", B, D, H, W, &p)) return 0;
    return p->ws_bytes;
}
/* AutoencoderKL.forward(x) -> (reconstruction, z_mu, z_sigma) with z = z_mu + z_sigma * eps, keeping the tape in `workspace`. */
int ldm_vae_train_forward(ldm_model* m, const float* x, const float* eps, float* recon, float* z_mu, float* z_sigma,
                          int B, int D, int H, int W, void* workspace, size_t workspace_bytes, void* stream) {
    if (!m || m->type != 1) return fail(LDM_ERR_BAD_ARG, "not an AutoencoderKL handle");
    if (!x || !eps || !recon || !z_mu || !z_sigma) return fail(LDM_ERR_BAD_ARG, "null tensor argument");
    const int f = 1 << (m->vcfg.num_levels - 1);
    if (D % f || H % f || W % f) return fail(LDM_ERR_UNSUPPORTED, "image size must be a multiple of %d", f);
    std::shared_ptr<Plan> p; LDM_TRY(get_plan(m, "train", B, D, H, W, &p));
    LDM_TRY(check_ready(m, workspace, workspace_bytes, *p));
    Bases bs{}; bs.p[BASE_WS] = (char*)workspace; bs.p[BASE_W] = m->arena; bs.p[BASE_W32] = m->arena32;
    bs.p[BASE_IO0] = (char*)x; bs.p[BASE_IO1] = (char*)eps; bs.p[BASE_IO2] = (char*)z_mu; bs.p[BASE_IO3] = (char*)z_sigma; bs.p[BASE_IO5] = (char*)recon;
    const int rt[2] = {m->vcfg.in_channels, 0};
    return run_plan(*p, bs, rt, (hipStream_t)stream, 0, p->bwd_begin);
}
/* d_recon: [B,Cout,D,H,W]; d_mu / d_sigma: [B,L,d,h,w] or NULL (the KL term's gradients); flat_grads as for the UNet. */
int ldm_vae_train_backward(ldm_model* m, const float* d_recon, const float* d_mu, const float* d_sigma, float* flat_grads,
                           int B, int D, int H, int W, void* workspace, size_t workspace_bytes, void* stream) {
    if (!m || m->type != 1) return fail(LDM_ERR_BAD_ARG, "not an AutoencoderKL handle");
    if (!d_recon || !flat_grads) return fail(LDM_ERR_BAD_ARG, "null tensor argument");
    std::shared_ptr<Plan> p; LDM_TRY(get_plan(m, "train", B, D, H, W, &p));
    LDM_TRY(check_ready(m, workspace, workspace_bytes, *p));
    Bases bs{}; bs.p[BASE_WS] = (char*)workspace; bs.p[BASE_W] = m->arena; bs.p[BASE_W32] = m->arena32;
    bs.p[BASE_IO0] = (char*)d_recon; bs.p[BASE_IO1] = (char*)d_mu; bs.p[BASE_IO2] = (char*)d_sigma; bs.p[BASE_IO4] = (char*)flat_grads;
    const int rt[2] = {m->vcfg.out_channels, 0};
    LaneCtx lanes;
    if (m->gsync.comm) { lanes.sync = &m->gsync; LDM_TRY(grad_sync_begin(m->gsync, (hipStream_t)stream)); }
    return run_plan(*p, bs, rt, (hipStream_t)stream, p->bwd_begin, p->ops.size(), lanes);
}

/* loss_out[0] = mean((pred - target)^2) over n fp32 elements; grad_out (optional, n elements) = 2 (pred - target) / n: the loss of
 * 3d_ldm/train_diffusion.py:207 and the gradient autograd hands to the network's backward (:214), in two launches. */
// reduction scratch of ldm_op_mse_loss / ldm_grad_sq_norm: one buffer per (device, entry point), allocated at the first call on
// that device (one process drives one GPU, but a caller with several devices current in turn must not get a foreign pointer);
// calls on the SAME device are expected on one stream at a time (include/ldm3d.h)
static float* device_scratch(int slot, size_t bytes) {
    static float* tab[32][2] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 32) return nullptr;
    if (!tab[dev][slot] && hipMalloc((void**)&tab[dev][slot], bytes) != hipSuccess) tab[dev][slot] = nullptr;
    return tab[dev][slot];
}
int ldm_op_mse_loss(const float* pred, const float* target, int64_t n, float* loss_out, float* grad_out, void* stream) {
    if (!pred || !target || !loss_out || n < 1) return fail(LDM_ERR_BAD_ARG, "bad argument");
    float* g_mse_parts = device_scratch(0, 1024 * 4);
    if (!g_mse_parts) return fail(LDM_ERR_HIP, "scratch allocation failed");
    const int nb = grid_for(n, 256 * 4, 1024);
    hipLaunchKernelGGL(mse_part_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, pred, target, (long)n, grad_out, g_mse_parts);
    hipLaunchKernelGGL(mse_fold_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, (const float*)g_mse_parts, nb, (long)n, loss_out);
    HIP_TRY(hipGetLastError());
    return 0;
}

int ldm_grad_sq_norm(const float* flat_grads, int64_t n, float* out, void* stream) {
    if (!flat_grads || !out || n < 0) return fail(LDM_ERR_BAD_ARG, "bad argument");
    if (((uintptr_t)flat_grads) & 15) return fail(LDM_ERR_BAD_ARG, "gradient buffer must be 16-byte aligned");
    float* g_norm_parts = device_scratch(1, 2048 * 4);
    if (!g_norm_parts) return fail(LDM_ERR_HIP, "scratch allocation failed");
    const int nb = grid_for((n + 3) / 4, 256, 2048);
    hipLaunchKernelGGL(sq_norm_part_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, flat_grads, (long)n, g_norm_parts);
    hipLaunchKernelGGL(sq_norm_fold_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, (const float*)g_norm_parts, nb, out);
    HIP_TRY(hipGetLastError());
    return 0;
}
int ldm_adam_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t n, float lr, float beta1,
                  float beta2, float eps, float weight_decay, int step, const float* sq_norm, float max_norm, void* stream) {
    if (!params || !grads || !exp_avg || !exp_avg_sq || n < 0 || step < 1) return fail(LDM_ERR_BAD_ARG, "bad argument");
    AdamCoef k{lr, beta1, beta2, eps, 1.0f - powf(beta1, (float)step), sqrtf(1.0f - powf(beta2, (float)step)), max_norm, lr * weight_decay, step};
    hipLaunchKernelGGL(adam_step_kernel, dim3(grid_for(n, 256, 8192)), dim3(256), 0, (hipStream_t)stream, params, grads, exp_avg, exp_avg_sq,
                       (long)n, k, sq_norm);
    HIP_TRY(hipGetLastError());
    return 0;
}

/* The same optimizer step over the model's flat parameter buffer (layout of ldm_model_param_offset) that ALSO re-packs the bf16
 * arena from the updated values in the same pass: replaces ldm_adam_step + ldm_model_load_params_flat after it. */
int ldm_model_adam_step(ldm_model* m, float* params_flat, const float* grads_flat, float* exp_avg, float* exp_avg_sq, float lr,
                        float beta1, float beta2, float eps, float weight_decay, int step, const float* sq_norm, float max_norm,
                        void* stream) {
    if (!m || !params_flat || !grads_flat || !exp_avg || !exp_avg_sq || step < 1) return fail(LDM_ERR_BAD_ARG, "bad argument");
    LDM_TRY(ensure_arena(m));
    LDM_TRY(ensure_pack_tab(m));
    AdamCoef k{lr, beta1, beta2, eps, 1.0f - powf(beta1, (float)step), sqrtf(1.0f - powf(beta2, (float)step)), max_norm, lr * weight_decay, step};
    hipLaunchKernelGGL(adam_pack_batched_kernel, dim3(m->pack_tab.nblocks), dim3(256), 0, (hipStream_t)stream,
                       (const PackDesc*)m->pack_tab.descs, (const int2*)m->pack_tab.map, params_flat, grads_flat, exp_avg, exp_avg_sq,
                       m->arena, k, sq_norm);
    if (m->precision == 1) {
        LDM_TRY(ensure_arena32(m));
        for (const ParamDesc& d : m->params) pack32_device(m, d, params_flat + d.flat_off, (hipStream_t)stream);
    }
    HIP_TRY(hipGetLastError());
    for (ParamDesc& d : m->params) if (!d.loaded) { d.loaded = true; m->loaded_count++; }
    m->derived_dirty = true; m->temb_tab_valid = false;
    return 0;
}

static int vae_factor(const ldm_model* m) { return 1 << (m->vcfg.num_levels - 1); }

// The samples of an inference batch are independent (GroupNorm and attention are per sample), and the kernels address a tensor through
// buffer descriptors with 32-bit byte offsets: a batch whose largest activation would pass 4 GiB (4 x 64 ch x 160x224x160 in the fp32
// mode: 5.9 GB) runs as several sub-batches of the largest size that divides B and fits, one after the other on the same stream and
// in the same workspace.  Returns that sub-batch size and its plan.
static int vae_fit_batch(ldm_model* m, const char* kind, int B, int D, int H, int W, std::shared_ptr<Plan>* p, int* chunk) {
    int rc = 0;
    for (int c = B; c >= 1; --c) {
        if (B % c) continue;
        rc = get_plan(m, kind, c, D, H, W, p);
        if (rc == 0) { *chunk = c; return 0; }
        if (rc != LDM_ERR_UNSUPPORTED) break;
    }
    return rc;
}
size_t ldm_vae_encode_workspace_bytes(ldm_model* m, int B, int D, int H, int W) {
    if (!m || m->type != 1) { fail(LDM_ERR_BAD_ARG, "not an AutoencoderKL handle"); return 0; }
    std::shared_ptr<Plan> p; int chunk = 0; if (vae_fit_batch(m, "enc", B, D, H, W, &p, &chunk)) return 0;
    return p->ws_bytes;
}
size_t ldm_vae_decode_workspace_bytes(ldm_model* m, int B, int d, int h, int w) {
    if (!m || m->type != 1) { fail(LDM_ERR_BAD_ARG, "not an AutoencoderKL handle"); return 0; }
    std::shared_ptr<Plan> p; int chunk = 0; if (vae_fit_batch(m, "dec", B, d, h, w, &p, &chunk)) return 0;
    return p->ws_bytes;
}

static int vae_encode_impl(ldm_model* m, const float* x, const float* eps, float* z_mu, float* z_sigma, float* z,
                           int B, int D, int H, int W, float* taps_out, const float* taps_in, int tap_mode,
                           void* workspace, size_t workspace_bytes, void* stream) {
    if (!m || m->type != 1) return fail(LDM_ERR_BAD_ARG, "not an AutoencoderKL handle");
    if (!x) return fail(LDM_ERR_BAD_ARG, "null tensor argument");
    const int f = vae_factor(m);
    if (D % f || H % f || W % f) return fail(LDM_ERR_UNSUPPORTED, "image size must be a multiple of %d", f);
    std::shared_ptr<Plan> p; int chunk = B;
    if (tap_mode) LDM_TRY(get_plan(m, "enc", B, D, H, W, &p, tap_mode)); else LDM_TRY(vae_fit_batch(m, "enc", B, D, H, W, &p, &chunk));
    LDM_TRY(check_ready(m, workspace, workspace_bytes, *p));
    LDM_TRY(ensure_derived(m, (hipStream_t)stream));
    const size_t in_n = (size_t)m->vcfg.in_channels * D * H * W, lat_n = (size_t)m->vcfg.latent_channels * (D / f) * (H / f) * (W / f);
    for (int b0 = 0; b0 < B; b0 += chunk) {
        Bases bs{}; bs.p[BASE_WS] = (char*)workspace; bs.p[BASE_W] = m->arena; bs.p[BASE_W32] = m->arena32;
        auto at = [&](const float* q, size_t per) { return q ? (char*)(q + (size_t)b0 * per) : nullptr; };
        bs.p[BASE_IO0] = at(x, in_n); bs.p[BASE_IO1] = at(eps, lat_n); bs.p[BASE_IO2] = at(z_mu, lat_n); bs.p[BASE_IO3] = at(z_sigma, lat_n);
        bs.p[BASE_IO4] = at(z, lat_n);
        bs.p[BASE_TAPO] = (char*)taps_out; bs.p[BASE_TAPI] = (char*)taps_in;
        const int rt[2] = {m->vcfg.in_channels, 0};
        LDM_TRY(run_plan(*p, bs, rt, (hipStream_t)stream));
    }
    return 0;
}
int ldm_vae_encode(ldm_model* m, const float* x, const float* eps, float* z_mu, float* z_sigma, float* z,
                   int B, int D, int H, int W, void* workspace, size_t workspace_bytes, void* stream) {
    return vae_encode_impl(m, x, eps, z_mu, z_sigma, z, B, D, H, W, nullptr, nullptr, 0, workspace, workspace_bytes, stream);
}
/* ldm_vae_encode / ldm_vae_decode with debug taps (see ldm_unet_forward_taps) */
int ldm_vae_encode_taps(ldm_model* m, const float* x, const float* eps, float* z_mu, float* z_sigma, float* z,
                        int B, int D, int H, int W, float* taps_out, const float* taps_in,
                        void* workspace, size_t workspace_bytes, void* stream) {
    if (!taps_out) return fail(LDM_ERR_BAD_ARG, "null tensor argument");
    return vae_encode_impl(m, x, eps, z_mu, z_sigma, z, B, D, H, W, taps_out, taps_in, taps_in ? 2 : 1, workspace, workspace_bytes, stream);
}

static int vae_decode_impl(ldm_model* m, const float* z, float* out, int B, int d, int h, int w, float* taps_out, const float* taps_in,
                           int tap_mode, void* workspace, size_t workspace_bytes, void* stream) {
    if (!m || m->type != 1) return fail(LDM_ERR_BAD_ARG, "not an AutoencoderKL handle");
    if (!z || !out) return fail(LDM_ERR_BAD_ARG, "null tensor argument");
    std::shared_ptr<Plan> p; int chunk = B;
    if (tap_mode) LDM_TRY(get_plan(m, "dec", B, d, h, w, &p, tap_mode)); else LDM_TRY(vae_fit_batch(m, "dec", B, d, h, w, &p, &chunk));
    LDM_TRY(check_ready(m, workspace, workspace_bytes, *p));
    LDM_TRY(ensure_derived(m, (hipStream_t)stream));
    const int f = vae_factor(m);
    const size_t lat_n = (size_t)m->vcfg.latent_channels * d * h * w, out_n = (size_t)m->vcfg.out_channels * d * h * w * f * f * f;
    for (int b0 = 0; b0 < B; b0 += chunk) {
        Bases bs{}; bs.p[BASE_WS] = (char*)workspace; bs.p[BASE_W] = m->arena; bs.p[BASE_W32] = m->arena32;
        bs.p[BASE_IO0] = (char*)(z + (size_t)b0 * lat_n); bs.p[BASE_IO1] = (char*)(out + (size_t)b0 * out_n);
        bs.p[BASE_TAPO] = (char*)taps_out; bs.p[BASE_TAPI] = (char*)taps_in;
        const int rt[2] = {m->vcfg.latent_channels, 0};
        LDM_TRY(run_plan(*p, bs, rt, (hipStream_t)stream));
    }
    return 0;
}
int ldm_vae_decode(ldm_model* m, const float* z, float* out, int B, int d, int h, int w,
                   void* workspace, size_t workspace_bytes, void* stream) {
    return vae_decode_impl(m, z, out, B, d, h, w, nullptr, nullptr, 0, workspace, workspace_bytes, stream);
}
int ldm_vae_decode_taps(ldm_model* m, const float* z, float* out, int B, int d, int h, int w, float* taps_out, const float* taps_in,
                        void* workspace, size_t workspace_bytes, void* stream) {
    if (!taps_out) return fail(LDM_ERR_BAD_ARG, "null tensor argument");
    return vae_decode_impl(m, z, out, B, d, h, w, taps_out, taps_in, taps_in ? 2 : 1, workspace, workspace_bytes, stream);
}

// ---- scheduler element-wise launches --------------------------------------------------------------------
int ldm_ddpm_step(const float* eps, const float* x, const float* noise, float* prev, float* x0_out, int64_t n,
                  float inv_sqrt_a, float sqrt_b, float c0, float c1, float sigma, int clip, void* stream) {
    if (!eps || !x || !prev || n < 0) return fail(LDM_ERR_BAD_ARG, "bad argument");
    StepCoef k{inv_sqrt_a, sqrt_b, c0, c1, sigma, 0.f, clip};
    hipLaunchKernelGGL(ddpm_step_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, eps, x, noise, prev, x0_out, (long)n, k);
    HIP_TRY(hipGetLastError());
    return 0;
}
int ldm_ddim_step(const float* eps, const float* x, const float* noise, float* prev, float* x0_out, int64_t n,
                  float inv_sqrt_a, float sqrt_b, float c0, float dir, float sigma, int clip, void* stream) {
    if (!eps || !x || !prev || n < 0) return fail(LDM_ERR_BAD_ARG, "bad argument");
    StepCoef k{inv_sqrt_a, sqrt_b, c0, 0.f, sigma, dir, clip};
    hipLaunchKernelGGL(ddim_step_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, eps, x, noise, prev, x0_out, (long)n, k);
    HIP_TRY(hipGetLastError());
    return 0;
}
int ldm_add_noise(const float* x0, const float* eps, const float* sqrt_a, const float* sqrt_b, float* out,
                  int B, int64_t per_sample, void* stream) {
    if (!x0 || !eps || !sqrt_a || !sqrt_b || !out || B < 1 || per_sample < 0) return fail(LDM_ERR_BAD_ARG, "bad argument");
    hipLaunchKernelGGL(add_noise_kernel, dim3(grid_for(per_sample * B)), dim3(256), 0, (hipStream_t)stream, x0, eps, sqrt_a, sqrt_b, out,
                       (long)per_sample, B);
    HIP_TRY(hipGetLastError());
    return 0;
}
int ldm_scale(const float* x, float* y, int64_t n, float s, void* stream) {
    if (!x || !y || n < 0) return fail(LDM_ERR_BAD_ARG, "bad argument");
    hipLaunchKernelGGL(scale_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, x, y, (long)n, s);
    HIP_TRY(hipGetLastError());
    return 0;
}

// ---- profiling: HIP events around every launch of one conv tile configuration ------------------------------
int ldm_profile_start(int wgm, int wgn, int bk, int max_launches) {
    if (max_launches < 1) return fail(LDM_ERR_BAD_ARG, "max_launches < 1");
    for (auto e : g_prof.ev) (void)hipEventDestroy(e);
    g_prof = ProfState();
    g_prof.ev.resize((size_t)max_launches * 2);
    for (auto& e : g_prof.ev) HIP_TRY(hipEventCreate(&e));
    g_prof.wgm = wgm; g_prof.wgn = wgn; g_prof.bk = bk & 0xff; g_prof.halo = (bk >> 8) & 1; g_prof.on = true;   // bk | 256 selects conv3_halo_kernel
    return 0;
}
/* Per-launch view of the instrumented launches, to be called BEFORE ldm_profile_stop (which frees the events): fills
 * flops[k] / ms[k] for k < min(n, max) and returns n (the number of instrumented launches) or a negative status. */
int ldm_profile_detail(double* flops, double* ms, int max) {
    if (!flops || !ms || max < 0) return fail(LDM_ERR_BAD_ARG, "null argument");
    const int n = (int)(g_prof.used / 2);
    for (int k = 0; k < n && k < max; ++k) {
        HIP_TRY(hipEventSynchronize(g_prof.ev[2 * k + 1]));
        float t = 0.f; HIP_TRY(hipEventElapsedTime(&t, g_prof.ev[2 * k], g_prof.ev[2 * k + 1]));
        flops[k] = g_prof.flops[k]; ms[k] = t;
    }
    return n;
}
/* out[0] = instrumented launches, out[1] = their total ms, out[2] = their total algorithmic FLOPs,
 * out[3] = all conv launches seen, out[4] = algorithmic FLOPs of all conv launches seen */
int ldm_profile_stop(double out[5]) {
    if (!out) return fail(LDM_ERR_BAD_ARG, "null out");
    g_prof.on = false;
    double ms = 0.0, fl = 0.0;
    for (size_t k = 0; k + 1 < g_prof.used + 1 && k < g_prof.used; k += 2) {
        HIP_TRY(hipEventSynchronize(g_prof.ev[k + 1]));
        float t = 0.f; HIP_TRY(hipEventElapsedTime(&t, g_prof.ev[k], g_prof.ev[k + 1]));
        ms += t; fl += g_prof.flops[k / 2];
    }
    out[0] = (double)(g_prof.used / 2); out[1] = ms; out[2] = fl; out[3] = (double)g_prof.launches_all; out[4] = g_prof.flops_all;
    for (auto e : g_prof.ev) (void)hipEventDestroy(e);
    g_prof.ev.clear(); g_prof.used = 0; g_prof.flops.clear();
    return 0;
}
/* Tile configuration the planner chose for each conv of a cached plan: fills cfgs[4*i..] = {wgm, wgn, bk | halo << 8, splitk} */
int ldm_model_plan_conv_cfgs(ldm_model* m, const char* kind, int B, int D, int H, int W, int* cfgs, int max_convs) {
    if (!m || !kind) return fail(LDM_ERR_BAD_ARG, "null argument");
    std::shared_ptr<Plan> p; LDM_TRY(get_plan(m, kind, B, D, H, W, &p));
    int n = 0;
    for (const Op& o : p->ops) if (o.kind == OP_CONV) {
        if (cfgs && n < max_convs) { cfgs[4 * n] = o.cc.wgm; cfgs[4 * n + 1] = o.cc.wgn; cfgs[4 * n + 2] = o.cc.bk | (o.cc.halo << 8); cfgs[4 * n + 3] = o.cc.splitk; }
        ++n;
    } else if (o.kind == OP_CONV_BLOCK) {            // conv3_block_kernel: reported as a 4 x 1 tile, 32-channel chunks, halo = 3
        if (cfgs && n < max_convs) { cfgs[4 * n] = 4; cfgs[4 * n + 1] = o.i[7] == 128 ? 2 : 1; cfgs[4 * n + 2] = 32 | ((o.i[7] == 128 ? 4 : 3) << 8); cfgs[4 * n + 3] = 1; }
        ++n;
    }
    return n;
}

/* Launches of a cached inference plan ("unet" | "enc" | "dec"; builds it if needed): the number of ops. */
int ldm_model_plan_launches(ldm_model* m, const char* kind, int B, int D, int H, int W) {
    if (!m || !kind) return fail(LDM_ERR_BAD_ARG, "null argument");
    std::shared_ptr<Plan> p; LDM_TRY(get_plan(m, kind, B, D, H, W, &p));
    int n = 0;
    for (const Op& o : p->ops) if (o.kind != OP_TAP) ++n;
    return n;
}

// ---- operator-level entry points (the same kernels the plans launch; used by per-kernel parity tests and
//      micro-benchmarks).  Tensors are NDHWC bf16 device memory with C % 32 == 0. -------------------------------
static char* g_zero_page = nullptr;
static int ensure_zero_page() {
    if (g_zero_page) return 0;
    HIP_TRY(hipMalloc((void**)&g_zero_page, 8192));
    HIP_TRY(hipMemset(g_zero_page, 0, 8192));
    HIP_TRY(hipDeviceSynchronize());                 // see ensure_arena
    return 0;
}

// stats (optional): GroupNorm partial slabs of the bf16 output, exactly as the plans request them from the producing kernel;
// *stats_nrb receives the slab rows per sample.  With stats the split-K finalize is the write-through variant the plans launch.
/* 3x3x3 stride-1 pad-1 convolution with 64 output channels as conv3_block_kernel (conv_block.h; the plans pick it for the AutoencoderKL's
 * full-resolution level): x [N][D][H][W][cin] bf16, w packed [27][64][cin] bf16, out [N*D*H*W][64] bf16; bias [64], temb [N][temb_stride],
 * residual [N*D*H*W][64] optional.  stats (optional): [N * rows][64][2] per-block (sum, sum of squares) of the stored values,
 * rows = ldm_op_conv3d_block_stats_rows(D, H, W, th); th = 8 (4 x 8 x 16 blocks) or 4 (4 x 4 x 16). */
/* The same for 128 output channels (conv3_block128_kernel): w packed [27][128][cin], out / residual [N*D*H*W][128], bias [128],
 * stats [N * ldm_op_conv3d_block_stats_rows(D, H, W, 8)][128][2]. */
int ldm_op_conv3d_block128(const void* x, int cin, const void* w, const float* bias, const float* temb, int temb_stride, const void* residual,
                           void* out, float* stats, int N, int D, int H, int W, void* stream) {
    if (!x || !w || !out || N < 1 || D < 1 || H < 1 || W < 1 || cin < 32 || cin % 32) return fail(LDM_ERR_BAD_ARG, "bad argument");
    if ((long)N * D * H * W * cin * 2 >= (1L << 32)) return fail(LDM_ERR_BAD_ARG, "the input tensor exceeds 4 GiB (split the batch)");
    BlockParams q{}; q.x = (const bf16_t*)x; q.w = (const bf16_t*)w; q.bias = bias; q.temb = temb; q.temb_stride = temb_stride;
    q.residual = (const bf16_t*)residual; q.out = (bf16_t*)out; q.stats = stats; q.N = N; q.D = D; q.H = H; q.W = W; q.Cin = cin;
    HIP_TRY(launch_conv_block128(q, (hipStream_t)stream));
    return 0;
}
/* tests only: the grid of conv3_block_kernel's tile loop (0 = two workgroups per CU); returns the previous value */
int ldm_debug_conv_block_slots(int slots) {
#ifdef LDM_EXPERIMENTS
    const int old = g_block_slots; g_block_slots = slots < 0 ? 0 : slots; return old;
#else
    (void)slots; return -1;                              // the tile-loop form of conv3_block_kernel is not in the product library
#endif
}
int ldm_op_conv3d_block_stats_rows(int D, int H, int W, int th) {
    if (th != 4 && th != 8) return 0;
    return ((D + BLK_TD - 1) / BLK_TD) * ((H + th - 1) / th) * ((W + BLK_TW - 1) / BLK_TW);
}
int ldm_op_conv3d_block(const void* x, int cin, const void* w, const float* bias, const float* temb, int temb_stride, const void* residual,
                        void* out, float* stats, int N, int D, int H, int W, int th, void* stream) {
    if (!x || !w || !out || N < 1 || D < 1 || H < 1 || W < 1 || cin < 32 || cin % 32 || (th != 4 && th != 8)) return fail(LDM_ERR_BAD_ARG, "bad argument");
    if ((long)N * D * H * W * cin * 2 >= (1L << 32)) return fail(LDM_ERR_BAD_ARG, "the input tensor exceeds 4 GiB (split the batch)");
    BlockParams q{}; q.x = (const bf16_t*)x; q.w = (const bf16_t*)w; q.bias = bias; q.temb = temb; q.temb_stride = temb_stride;
    q.residual = (const bf16_t*)residual; q.out = (bf16_t*)out; q.stats = stats; q.N = N; q.D = D; q.H = H; q.W = W; q.Cin = cin;
    HIP_TRY(launch_conv_block(q, th, (hipStream_t)stream));
    return 0;
}

static int op_conv3d_impl(const void* xa, int ca, const void* xb, int cb, const void* w, const float* bias,
                          const void* x1a, int c1a, const void* x1b, int c1b, const void* w1, const float* bias2,
                          const float* temb, int temb_stride, const void* residual, void* out_bf16, float* out_f32,
                          int N, int Din, int Hin, int Win, int ksize, int stride, int pad, int ups,
                          int cout, int cout_pad, int wgn, int splitk, void* scratch, size_t scratch_bytes, void* stream,
                          float* stats, int* stats_nrb, FinGnParams* fg = nullptr) {
    if (!xa || !w || (!out_bf16 && !out_f32 && !fg)) return fail(LDM_ERR_BAD_ARG, "null tensor argument");
    if (!xb) cb = 0;
    if (!x1a) { c1a = 0; c1b = 0; }
    if (!x1b) c1b = 0;
    const int cin0 = ca + cb, cin1 = c1a + c1b;
    // zero padding reads come from the 8 KiB zero page: at most 4096 channels per source row for a padded conv (1x1 GEMMs have none)
    const int cmax = (ksize == 1 && stride == 1 && pad == 0 && ups == 0) ? 65536 : 4096;
    if (ca % 32 || cb % 32 || c1a % 32 || c1b % 32 || cout_pad % 64 || cout > cout_pad || cin0 > cmax || cin1 > 4096)
        return fail(LDM_ERR_BAD_ARG, "channel counts must be multiples of 32 (cout_pad of 64), at most %d per row", cmax);
    if (ksize != 1 && ksize != 3) return fail(LDM_ERR_UNSUPPORTED, "ksize must be 1 or 3");
    int exact = 0;
    if (ups == 2) { ups = 1; exact = 1; }            // ups = 2: zero-insertion upsample (transposed stride-2 conv)
    if (ups < 0 || ups > 1 || stride < 1 || stride > 2) return fail(LDM_ERR_UNSUPPORTED, "stride 1|2, ups 0|1|2");
    LDM_TRY(ensure_zero_page());
    int bk = 64;
    if (ca % 64 || cb % 64 || c1a % 64 || c1b % 64) bk = 32;
    const int Du = Din << ups, Hu = Hin << ups, Wu = Win << ups;
    const int pad_total = (stride == 2 && pad == 0 && ksize == 3) ? 1 : 2 * pad;      // F.pad(0,1) form for s2 p0
    const int Do = (Du + pad_total - ksize) / stride + 1, Ho = (Hu + pad_total - ksize) / stride + 1,
              Wo = (Wu + pad_total - ksize) / stride + 1;
    const long M = (long)N * Do * Ho * Wo;
    if (M >= (1L << 31) || M < 1) return fail(LDM_ERR_BAD_ARG, "bad output size");
    if (!stats && !wgn && !splitk && out_bf16 && !out_f32 && gemm_light_ok(ksize, stride, ups, cin0, cin1 == 0, !temb && !bias2) &&
        Builder::light_enabled() && M * (long)cin0 * 2 < (1L << 31)) {
        LightParams lp{}; lp.x = (const bf16_t*)xa; lp.w = (const bf16_t*)w; lp.bias = bias; lp.residual = (const bf16_t*)residual;
        lp.out = (bf16_t*)out_bf16; lp.stats = nullptr; lp.M = (int)M; lp.K = cin0; lp.CoutS = rup(cout, 32);
        lp.xb = (const bf16_t*)xb; lp.ca = ca;
        HIP_TRY(launch_gemm_light(lp, cout_pad, gemm_light_big(M, cout_pad), (hipStream_t)stream));
        return 0;
    }
    const int taps = ksize * ksize * ksize;
    ConvParams p{};
    p.x0a = (const bf16_t*)xa; p.x0b = (const bf16_t*)xb; p.c0a = ca; p.c0b = cb; p.w0 = (const bf16_t*)w;
    p.x1a = (const bf16_t*)x1a; p.x1b = (const bf16_t*)x1b; p.c1a = c1a; p.c1b = c1b; p.w1 = (const bf16_t*)w1;
    p.zero_page = (const bf16_t*)g_zero_page;
    p.N = N; p.Din = Din; p.Hin = Hin; p.Win = Win; p.Dout = Do; p.Hout = Ho; p.Wout = Wo;
    p.ksize = ksize; p.stride = stride; p.pad = pad; p.ups = ups; p.exact = exact; p.M = (int)M;
    p.CoutS = rup(cout, 32); p.CoutPad = cout_pad; p.CoutReal = cout;
    p.nchunk0 = cin0 / bk; p.nchunk1 = cin1 / bk; p.steps0 = taps * p.nchunk0; p.steps1 = p.nchunk1;
    const bool halo_ok = ksize == 3 && stride == 1 && pad == 1 && ups == 0 && !exact && cb == 0 && bk == 64 &&
                         (cin1 == 0 || (Builder::halo_skip_enabled() && (splitk <= 1 || Builder::halo_skip_split_enabled()))) && Builder::halo_enabled();
    ConvCfg cc = Builder::choose_cfg(M, cout_pad, p.steps0, bk, halo_ok ? N : 0, (long)Do * Ho * Wo, false, p.steps1);
    if (wgn) {                                       // forced tile shape: wgn = 2 keeps the halo kernel where it applies
        if ((wgn != 1 && wgn != 2 && wgn != 4) || cout_pad % (64 * wgn)) return fail(LDM_ERR_BAD_ARG, "bad wgn");
        cc.wgn = wgn; cc.wgm = 4 / wgn; cc.halo = (wgn == 2 && halo_ok && cout_pad % 128 == 0) ? 1 : (wgn == 1 && halo_ok && Builder::tall_mode() != 0) ? 2 : 0;
    }
    if (splitk) cc.splitk = splitk;
    if (cc.splitk < 1 || cc.splitk > p.steps0 + p.steps1) return fail(LDM_ERR_BAD_ARG, "bad splitk");
    if (cc.halo) {                                   // K splits of whole (kd, kh, chunk) macro steps, none empty
        const int Q = 9 * p.nchunk0;
        if (cc.splitk > Q) cc.splitk = Q;
        cc.qps = (Q + cc.splitk - 1) / cc.splitk; cc.splitk = (Q + cc.qps - 1) / cc.qps;
        cc.mtps = (int)(((long)Do * Ho * Wo + (cc.halo == 2 ? 253 : 125)) / (cc.halo == 2 ? 254 : 126));
    }
    p.splitk = cc.splitk; p.steps_per_split = cc.halo ? 3 * cc.qps : (p.steps0 + p.steps1 + cc.splitk - 1) / cc.splitk;
    p.halo_mtps = cc.mtps; p.q_per_split = cc.qps;
    p.mtiles = cc.halo ? N * cc.mtps : (int)((M + 64 * cc.wgm - 1) / (64 * cc.wgm)); p.ntiles = cout_pad / (64 * cc.wgn);
    p.bias = bias; p.bias2 = bias2; p.temb = temb; p.temb_stride = temb_stride; p.residual = (const bf16_t*)residual;
    p.out = out_f32 ? nullptr : (bf16_t*)out_bf16; p.out_f32 = out_f32;
    if (cc.splitk > 1) {
        const size_t need = (size_t)cc.splitk * M * cout_pad * 4;
        if (!scratch || scratch_bytes < need) return fail(LDM_ERR_WORKSPACE, "split-K scratch too small: need %zu bytes", need);
        p.partial = (float*)scratch;
    }
    if (stats) {                                     // slab geometry as Builder::conv lays it out
        const long dhwo = (long)Do * Ho * Wo; const int bm = 64 * cc.wgm;
        if (out_f32 || !stats_nrb) return fail(LDM_ERR_BAD_ARG, "statistics need the bf16 output form");
        if (cc.splitk > 1) { if (N > 1 && dhwo % 32) return fail(LDM_ERR_UNSUPPORTED, "statistics: DHW %% 32 != 0 with a batch"); *stats_nrb = (int)(N == 1 ? (M + 31) / 32 : dhwo / 32); }
        else if (cc.halo) *stats_nrb = cc.mtps;
        else { if (N > 1 && dhwo % bm) return fail(LDM_ERR_UNSUPPORTED, "statistics: tiles straddle samples"); *stats_nrb = (int)(N == 1 ? (M + bm - 1) / bm : dhwo / bm); }
        p.stats = stats;
    }
    p.dbg = (int)ldm_xknob("LDM_CONV_DBG", 0);
    if (p.dbg & 512) {                          // diagnostic stamps go to the caller's scratch (needs nwg * 64 bytes)
        if (cc.splitk > 1 || !scratch || scratch_bytes < (size_t)p.mtiles * p.ntiles * (64 + 4096)) return fail(LDM_ERR_BAD_ARG, "stamps need scratch and splitk 1");
        p.stamps = (unsigned long long*)scratch;
    }
    if (fg && cc.splitk > 1) p.slab_lg = fg->lg;     // planar slabs for the group-owning finalize
    LDM_TRY(launch_conv(p, cc, (hipStream_t)stream));
    if (cc.splitk > 1) {
        FinalizeParams f{}; f.partial = p.partial; f.splitk = p.splitk; f.M = p.M; f.CoutPad = p.CoutPad; f.CoutS = p.CoutS;
        f.CoutReal = p.CoutReal; f.DHWo = Do * Ho * Wo; f.bias = bias; f.bias2 = bias2; f.temb = temb; f.temb_stride = temb_stride;
        f.residual = p.residual; f.out = p.out; f.out_f32 = p.out_f32; f.stats = stats;
        if (fg) {                                    // finalize + GroupNorm in one launch (fin_gn.h), as the plans' OP_FIN_GN
            fg->f = f; fg->f.stats = nullptr;
            HIP_TRY(launch_fin_gn(*fg, N, wt_stores(), (hipStream_t)stream));
        } else
            launch_finalize(f, stats && wt_stores(), (hipStream_t)stream);
    } else if (fg) return fail(LDM_ERR_UNSUPPORTED, "the fused finalize + GroupNorm needs a conv that is split over K");
    HIP_TRY(hipGetLastError());
    return 0;
}

int ldm_op_conv3d(const void* xa, int ca, const void* xb, int cb, const void* w, const float* bias,
                  const void* x1a, int c1a, const void* x1b, int c1b, const void* w1, const float* bias2,
                  const float* temb, int temb_stride, const void* residual, void* out_bf16, float* out_f32,
                  int N, int Din, int Hin, int Win, int ksize, int stride, int pad, int ups,
                  int cout, int cout_pad, int wgn, int splitk, void* scratch, size_t scratch_bytes, void* stream) {
    return op_conv3d_impl(xa, ca, xb, cb, w, bias, x1a, c1a, x1b, c1b, w1, bias2, temb, temb_stride, residual, out_bf16, out_f32, N, Din, Hin, Win,
                          ksize, stride, pad, ups, cout, cout_pad, wgn, splitk, scratch, scratch_bytes, stream, nullptr, nullptr);
}

/* The producer -> GroupNorm pair exactly as the inference plans launch it: a 3^3 stride-1 conv (bias only) whose epilogue - or whose
 * write-through split-K finalize - leaves the GroupNorm partial slabs of its bf16 output, then the ONE-launch GroupNorm(+SiLU)
 * (gn_fused_apply_kernel, write-through stores) that folds those slabs.  conv_out [M][round32(cout)] and gn_out (same shape) are bf16
 * NDHWC.  scratch: split-K slabs + statistics slabs (ldm_op_conv3d_gn_scratch_bytes).  Returns LDM_ERR_UNSUPPORTED where the plans
 * would not take the one-launch path (channels per group > 64, more than 512 slab rows per sample). */
size_t ldm_op_conv3d_gn_scratch_bytes(int N, int D, int H, int W, int cout_pad, int splitk) {
    const size_t M = (size_t)N * D * H * W;
    return (size_t)(splitk > 1 ? splitk : 0) * M * cout_pad * 4 + ((M + 31) / 32 + 64) * cout_pad * 2 * 4 + 1024;
}
int ldm_op_conv3d_gn(const void* x, int cin, const void* w, const float* bias, const float* gamma, const float* beta, int groups, float eps,
                     int silu, void* conv_out, void* gn_out, int N, int D, int H, int W, int cout, int cout_pad, int wgn, int splitk,
                     void* scratch, size_t scratch_bytes, void* stream) {
    if (!x || !w || !gamma || !beta || !conv_out || !gn_out || !scratch) return fail(LDM_ERR_BAD_ARG, "null tensor argument");
    const int C = rup(cout, 32);
    if (groups < 1 || C % groups || cout != C) return fail(LDM_ERR_BAD_ARG, "cout must be a multiple of 32 and of groups");
    if (scratch_bytes < ldm_op_conv3d_gn_scratch_bytes(N, D, H, W, cout_pad, splitk)) return fail(LDM_ERR_WORKSPACE, "scratch too small");
    const size_t M = (size_t)N * D * H * W;
    const size_t slab_bytes = (size_t)(splitk > 1 ? splitk : 0) * M * cout_pad * 4;
    float* stats = (float*)((char*)scratch + rup_sz(slab_bytes, 256));
    int nrb = 0;
    LDM_TRY(op_conv3d_impl(x, cin, nullptr, 0, w, bias, nullptr, 0, nullptr, 0, nullptr, nullptr, nullptr, 0, nullptr, conv_out, nullptr,
                           N, D, H, W, 3, 1, 1, 0, cout, cout_pad, wgn, splitk, scratch, slab_bytes, stream, stats, &nrb));
    const int DHW = D * H * W;
    if (C / groups > 64 || nrb > 512) return fail(LDM_ERR_UNSUPPORTED, "the plans use the two-launch GroupNorm here (%d channels per group, %d slab rows)", C / groups, nrb);
    const int slices = (C + 63) / 64;
    int chunks = std::max(1, std::min(256 / (slices * N), (DHW + 31) / 32));
    const int rpb = rup((DHW + chunks - 1) / chunks, 32);
    chunks = (DHW + rpb - 1) / rpb;
    GnFusedParams g{}; g.xa = (const bf16_t*)conv_out; g.xb = nullptr; g.ca = C; g.cb = 0; g.sa = stats; g.sb = nullptr; g.nrb_a = nrb; g.nrb_b = 0;
    g.groups = groups; g.DHW = DHW; g.N = N; g.silu = silu; g.rows_per_block = rpb; g.eps = eps; g.gamma = gamma; g.beta = beta; g.out = (bf16_t*)gn_out;
    if (wt_stores()) hipLaunchKernelGGL(gn_fused_apply_kernel<true>, dim3(chunks, slices, N), dim3(256), 0, (hipStream_t)stream, g);
    else hipLaunchKernelGGL(gn_fused_apply_kernel<false>, dim3(chunks, slices, N), dim3(256), 0, (hipStream_t)stream, g);
    HIP_TRY(hipGetLastError());
    return 0;
}

/* The split-K conv -> GroupNorm pair as the inference plans launch it at the 12^3 / 6^3 levels (OP_CONV + OP_FIN_GN, csrc/fin_gn.h): a 3^3
 * stride-1 conv split over K (splitk >= 2) that writes its fp32 slabs planar (one plane per GroupNorm group), then ONE launch in which
 * every workgroup owns a whole (sample, group): it sums the slabs, applies bias / per-sample channel bias `temb` / `residual`, rounds to
 * bf16, reduces the group's statistics and writes GroupNorm(+SiLU) of the rounded tensor to gn_out; conv_out (optional) receives the
 * un-normalised bf16 tensor.  scratch: the split-K slabs (ldm_op_conv3d_fin_gn_scratch_bytes).
 * LDM_ERR_UNSUPPORTED where the plans keep the two launches (channels per group not a power of two in 4 ... 64, or more than 4096
 * (row, 4-channel) items per group). */
size_t ldm_op_conv3d_fin_gn_scratch_bytes(int N, int D, int H, int W, int cout_pad, int splitk) {
    return rup_sz((size_t)splitk * N * D * H * W * cout_pad * 4, 256);
}
int ldm_op_conv3d_fin_gn(const void* x, int cin, const void* w, const float* bias, const float* temb, int temb_stride, const void* residual,
                         const float* gamma, const float* beta, int groups, float eps, int silu, void* conv_out, void* gn_out,
                         int N, int D, int H, int W, int cout, int cout_pad, int wgn, int splitk, void* scratch, size_t scratch_bytes,
                         void* stream) {
    if (!x || !w || !gamma || !beta || !gn_out || !scratch) return fail(LDM_ERR_BAD_ARG, "null tensor argument");
    const int C = rup(cout, 32);
    if (groups < 1 || C % groups || cout != C || splitk < 2) return fail(LDM_ERR_BAD_ARG, "cout must be a multiple of 32 and of groups, splitk >= 2");
    if (!fin_gn_ok(C, groups, D * H * W) || cout_pad % (C / groups))
        return fail(LDM_ERR_UNSUPPORTED, "the plans keep finalize and GroupNorm apart here (%d channels per group, %d rows)", C / groups, D * H * W);
    if (scratch_bytes < ldm_op_conv3d_fin_gn_scratch_bytes(N, D, H, W, cout_pad, splitk)) return fail(LDM_ERR_WORKSPACE, "scratch too small");
    const size_t slab_bytes = rup_sz((size_t)splitk * N * D * H * W * cout_pad * 4, 256);
    FinGnParams q{}; q.gamma = gamma; q.beta = beta; q.y = (bf16_t*)gn_out; q.groups = groups; q.silu = silu; q.eps = eps; q.lg = fin_gn_lg(C, groups);
    LDM_TRY(op_conv3d_impl(x, cin, nullptr, 0, w, bias, nullptr, 0, nullptr, 0, nullptr, nullptr, temb, temb_stride, residual, conv_out, nullptr,
                           N, D, H, W, 3, 1, 1, 0, cout, cout_pad, wgn, splitk, scratch, slab_bytes, stream, nullptr, nullptr, &q));
    return 0;
}

size_t ldm_op_group_norm_scratch_bytes(int N, int C, int DHW) {
    const int cvec = C / 8, rows_par = std::max(1, 256 / std::max(1, cvec));
    int nslab = std::min((DHW + rows_par - 1) / rows_par, std::max(1, 512 / N));
    return ((size_t)N * nslab * C * 2 + (size_t)N * C * 2) * 4 + 512;
}

int ldm_op_group_norm(const void* xa, int ca, const void* xb, int cb, const float* gamma, const float* beta,
                      int groups, float eps, int silu, void* out, int N, int DHW, void* scratch, size_t scratch_bytes,
                      void* stream) {
    if (!xa || !gamma || !beta || !out || !scratch) return fail(LDM_ERR_BAD_ARG, "null tensor argument");
    if (!xb) cb = 0;
    const int C = ca + cb;
    if (C % 8 || ca % 8 || groups < 1 || C % groups) return fail(LDM_ERR_BAD_ARG, "bad channel / group counts");
    if (scratch_bytes < ldm_op_group_norm_scratch_bytes(N, C, DHW)) return fail(LDM_ERR_WORKSPACE, "scratch too small");
    const int cvec = C / 8, rows_par = std::max(1, 256 / cvec);
    int nslab = std::min((DHW + rows_par - 1) / rows_par, std::max(1, 512 / N));
    const int rps = (DHW + nslab - 1) / nslab; nslab = (DHW + rps - 1) / rps;
    float* partial = (float*)scratch;
    float* ab = partial + (size_t)N * nslab * C * 2;
    hipStream_t s = (hipStream_t)stream;
    GnStatsParams sp{}; sp.xa = (const bf16_t*)xa; sp.xb = (const bf16_t*)xb; sp.ca = ca; sp.cb = cb; sp.DHW = DHW; sp.nslab = nslab;
    sp.rows_per_slab = rps; sp.partial = partial;
    hipLaunchKernelGGL(gn_stats_kernel, dim3(nslab, N), dim3(256), 0, s, sp);
    GnFinalizeParams fp{}; fp.partial = partial; fp.nslab = nslab; fp.C = C; fp.Creal = C; fp.groups = groups; fp.DHW = DHW; fp.eps = eps;
    fp.gamma = gamma; fp.beta = beta; fp.ab = ab;
    hipLaunchKernelGGL(gn_finalize_kernel, dim3(groups, N), dim3(256), 0, s, fp);
    GnApplyParams ap{}; ap.xa = sp.xa; ap.xb = sp.xb; ap.ca = ca; ap.cb = cb; ap.DHW = DHW; ap.N = N; ap.silu = silu; ap.ab = ab; ap.out = (bf16_t*)out;
    hipLaunchKernelGGL(gn_apply_kernel, dim3(grid_for((long)N * DHW * cvec, 256, 2048)), dim3(256), 0, s, ap);
    HIP_TRY(hipGetLastError());
    return 0;
}

/* ---- PatchDiscriminator building blocks (stage-1 GAN tail): generic-kernel-size convs as im2col + the 1x1 GEMM kernels ---- */
}  // extern "C" (templates need C++ linkage)
template <class T>
static int op_im2col(const void* x, void* col, int N, int D, int H, int W, int Cs, int C, int k, int stride, int pad, int Kp, int kmul, void* stream) {
    if (!x || !col || N < 1 || C < 1 || C > Cs || k < 1 || stride < 1 || pad < 0 || Kp < k * k * k * C || Kp % kmul) return fail(LDM_ERR_BAD_ARG, "bad argument");
    const int Do = (D + 2 * pad - k) / stride + 1, Ho = (H + 2 * pad - k) / stride + 1, Wo = (W + 2 * pad - k) / stride + 1;
    if (Do < 1 || Ho < 1 || Wo < 1) return fail(LDM_ERR_BAD_ARG, "empty output");
    hipLaunchKernelGGL(im2col_generic_kernel<T>, dim3(grid_for((long)N * Do * Ho * Wo * Kp, 256, 16384)), dim3(256), 0, (hipStream_t)stream,
                       (const T*)x, (T*)col, N, D, H, W, Cs, C, k, stride, pad, Do, Ho, Wo, Kp);
    HIP_TRY(hipGetLastError());
    return 0;
}
template <class T>
static int op_col2im(const void* dcol, void* dx, int N, int D, int H, int W, int Cs, int C, int k, int stride, int pad, int Kp, int kmul, void* stream) {
    if (!dcol || !dx || N < 1 || C < 1 || C > Cs || k < 1 || stride < 1 || pad < 0 || Kp < k * k * k * C || Kp % kmul) return fail(LDM_ERR_BAD_ARG, "bad argument");
    const int Do = (D + 2 * pad - k) / stride + 1, Ho = (H + 2 * pad - k) / stride + 1, Wo = (W + 2 * pad - k) / stride + 1;
    if (Do < 1 || Ho < 1 || Wo < 1) return fail(LDM_ERR_BAD_ARG, "empty output");
    hipLaunchKernelGGL(col2im_generic_kernel<T>, dim3(grid_for((long)N * D * H * W * Cs, 256, 16384)), dim3(256), 0, (hipStream_t)stream,
                       (const T*)dcol, (T*)dx, N, D, H, W, Cs, C, k, stride, pad, Do, Ho, Wo, Kp);
    HIP_TRY(hipGetLastError());
    return 0;
}
extern "C" {
int ldm_op_im2col(const void* x, void* col, int N, int D, int H, int W, int Cs, int C, int k, int stride, int pad, int Kp, void* stream) {
    return op_im2col<bf16_t>(x, col, N, D, H, W, Cs, C, k, stride, pad, Kp, 32, stream);
}
int ldm_op_col2im(const void* dcol, void* dx, int N, int D, int H, int W, int Cs, int C, int k, int stride, int pad, int Kp, void* stream) {
    return op_col2im<bf16_t>(dcol, dx, N, D, H, W, Cs, C, k, stride, pad, Kp, 32, stream);
}
int ldm_op_leaky_relu(const void* x, void* y, int64_t n, float slope, void* stream) {
    if (!x || !y || n < 0) return fail(LDM_ERR_BAD_ARG, "bad argument");
    hipLaunchKernelGGL(leaky_relu_kernel<bf16_t>, dim3(grid_for(n, 256, 8192)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, (bf16_t*)y, (long)n, slope);
    HIP_TRY(hipGetLastError());
    return 0;
}
int ldm_op_leaky_relu_bwd(const void* x, const void* dy, void* dx, int64_t n, float slope, void* stream) {
    if (!x || !dy || !dx || n < 0) return fail(LDM_ERR_BAD_ARG, "bad argument");
    hipLaunchKernelGGL(leaky_relu_bwd_kernel<bf16_t>, dim3(grid_for(n, 256, 8192)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, (const bf16_t*)dy, (bf16_t*)dx, (long)n, slope);
    HIP_TRY(hipGetLastError());
    return 0;
}
/* fp32 NCDHW [N][C][DHW] <-> bf16 NDHWC [N][DHW][Cs] (channels zero-padded to Cs): the layout conversions the plans do internally */
int ldm_op_pack_ncdhw(const float* x, void* out, int N, int C, int Cs, int64_t DHW, void* stream) {
    if (!x || !out || N < 1 || C < 1 || C > Cs || DHW < 1 || DHW >= (1L << 31)) return fail(LDM_ERR_BAD_ARG, "bad argument");
    hipLaunchKernelGGL(pack2_ncdhw_kernel, dim3(grid_for((long)N * DHW * Cs)), dim3(256), 0, (hipStream_t)stream, x, C, (const float*)nullptr, 0, (bf16_t*)out, N, Cs, (int)DHW);
    HIP_TRY(hipGetLastError());
    return 0;
}
int ldm_op_unpack_ndhwc(const void* act, float* out, int N, int C, int Cs, int64_t DHW, void* stream) {
    if (!act || !out || N < 1 || C < 1 || C > Cs || DHW < 1 || DHW >= (1L << 31)) return fail(LDM_ERR_BAD_ARG, "bad argument");
    hipLaunchKernelGGL(unpack_ndhwc_kernel<bf16_t>, dim3(grid_for((long)N * C * DHW)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)act, out, N, C, Cs, (int)DHW);
    HIP_TRY(hipGetLastError());
    return 0;
}

/* ---- the same building blocks on fp32 NDHWC tensors (the reference trains the PatchDiscriminator in fp32 when AMP is off,
 *      3d_ldm/train_autoencoder.py:150-158,454-494; `--precision fp32`): exact fp32 MFMA (csrc/f32_path.h, f32_train.h).
 *      Channels / K are multiples of 16, Cout padded to 64 weight rows. ------------------------------------------------------ */
int ldm_op_im2col_f32(const float* x, float* col, int N, int D, int H, int W, int Cs, int C, int k, int stride, int pad, int Kp, void* stream) {
    return op_im2col<float>(x, col, N, D, H, W, Cs, C, k, stride, pad, Kp, 16, stream);
}
int ldm_op_col2im_f32(const float* dcol, float* dx, int N, int D, int H, int W, int Cs, int C, int k, int stride, int pad, int Kp, void* stream) {
    return op_col2im<float>(dcol, dx, N, D, H, W, Cs, C, k, stride, pad, Kp, 16, stream);
}
int ldm_op_leaky_relu_f32(const float* x, float* y, int64_t n, float slope, void* stream) {
    if (!x || !y || n < 0) return fail(LDM_ERR_BAD_ARG, "bad argument");
    hipLaunchKernelGGL(leaky_relu_kernel<float>, dim3(grid_for(n, 256, 8192)), dim3(256), 0, (hipStream_t)stream, x, y, (long)n, slope);
    HIP_TRY(hipGetLastError());
    return 0;
}
int ldm_op_leaky_relu_bwd_f32(const float* x, const float* dy, float* dx, int64_t n, float slope, void* stream) {
    if (!x || !dy || !dx || n < 0) return fail(LDM_ERR_BAD_ARG, "bad argument");
    hipLaunchKernelGGL(leaky_relu_bwd_kernel<float>, dim3(grid_for(n, 256, 8192)), dim3(256), 0, (hipStream_t)stream, x, dy, dx, (long)n, slope);
    HIP_TRY(hipGetLastError());
    return 0;
}
int ldm_op_pack_ncdhw_f32(const float* x, float* out, int N, int C, int Cs, int64_t DHW, void* stream) {
    if (!x || !out || N < 1 || C < 1 || C > Cs || DHW < 1 || DHW >= (1L << 31)) return fail(LDM_ERR_BAD_ARG, "bad argument");
    hipLaunchKernelGGL(pack2_ncdhw_f32_kernel, dim3(grid_for((long)N * DHW * Cs)), dim3(256), 0, (hipStream_t)stream, x, C, (const float*)nullptr, 0, out, N, Cs, (int)DHW);
    HIP_TRY(hipGetLastError());
    return 0;
}
int ldm_op_unpack_ndhwc_f32(const float* act, float* out, int N, int C, int Cs, int64_t DHW, void* stream) {
    if (!act || !out || N < 1 || C < 1 || C > Cs || DHW < 1 || DHW >= (1L << 31)) return fail(LDM_ERR_BAD_ARG, "bad argument");
    hipLaunchKernelGGL(unpack_ndhwc_kernel<float>, dim3(grid_for((long)N * C * DHW)), dim3(256), 0, (hipStream_t)stream, act, out, N, C, Cs, (int)DHW);
    HIP_TRY(hipGetLastError());
    return 0;
}
/* out[M][couts] = x[M][K] w[cout_pad][K]^T + bias (fp32, exact fp32 MFMA): K % 16 == 0, cout_pad % 64 == 0 weight rows (rows >= cout zero),
 * couts = stored output channels (% 4 == 0, >= cout; columns >= cout receive the products of the zero rows). */
int ldm_op_gemm_f32(const float* x, int K, const float* w, const float* bias, float* out, int64_t M, int cout, int cout_pad, int couts, void* stream) {
    if (!x || !w || !out || M < 1 || M >= (1L << 31) || K < 16 || K % 16 || cout < 1 || cout_pad % 64 || cout_pad < cout || couts % 4 || couts < cout || couts > cout_pad)
        return fail(LDM_ERR_BAD_ARG, "bad argument");
    Conv32Params p{}; p.xa = x; p.ca = K; p.w = w; p.N = 1; p.Din = p.Dout = (int)M; p.Hin = p.Win = p.Hout = p.Wout = 1;
    p.ksize = 1; p.stride = 1; p.pad = 0; p.M = (int)M; p.CoutS = couts; p.CoutPad = cout_pad; p.CoutReal = cout;
    p.nchunk = K / 16; p.steps = p.nchunk; p.splitk = 1; p.steps_per_split = p.steps; p.mtiles = (int)((M + 127) / 128);
    p.bias = bias; p.out = out;
    if (cout_pad % 128 == 0) { p.ntiles = cout_pad / 128; hipLaunchKernelGGL(conv_f32_kernel<128>, dim3(p.mtiles * p.ntiles), dim3(256), 0, (hipStream_t)stream, p); }
    else { p.ntiles = cout_pad / 64; hipLaunchKernelGGL(conv_f32_kernel<64>, dim3(p.mtiles * p.ntiles), dim3(256), 0, (hipStream_t)stream, p); }
    HIP_TRY(hipGetLastError());
    return 0;
}
/* out[M][couts] = (xa | xb)[M][ca + cb] w[cout_pad][ca + cb]^T + bias (+ residual[M][couts]) on fp32 operands as three bf16 MFMAs per product
 * (gemm_light_x3.h: the 1x1x1 convolutions of the fp32 inference plans).  ca, cb % 32 == 0 (cb = 0: one source), cout_pad % 32 == 0,
 * couts % 32 == 0.  stats (optional): [ceil(M / rows)][couts][2] per-tile (sum, sum of squares) of the stored values, rows = 64 when
 * big else 32.  big: 64 x 64 tiles (cout_pad % 64 == 0) instead of 32 x 32; -1 = the planner's choice. */
int ldm_op_linear_f32x3(const float* xa, int ca, const float* xb, int cb, const float* w, const float* bias, const float* residual, float* out,
                        float* stats, int64_t M, int cout_pad, int couts, int big, void* stream) {
    if (!xa || !w || !out || M < 1 || ca < 32 || ca % 32 || cb < 0 || cb % 32 || (cb > 0 && !xb) || cout_pad < 32 || cout_pad % 32 || couts < 32 || couts % 32 ||
        couts > cout_pad || M * (int64_t)std::max(ca + cb, cout_pad) * 4 >= (1L << 31) || (big == 1 && cout_pad % 64))
        return fail(LDM_ERR_BAD_ARG, "bad argument");
    LightX3Params p{}; p.x = xa; p.xb = cb > 0 ? xb : nullptr; p.ca = ca; p.w = w; p.bias = bias; p.residual = residual; p.out = out; p.stats = stats;
    p.M = (int)M; p.K = ca + cb; p.CoutS = couts;
    HIP_TRY(launch_gemm_light_x3(p, cout_pad, big < 0 ? gemm_light_x3_big(M, cout_pad) : big, (hipStream_t)stream));
    return 0;
}
/* dw[ksplit][cout][K] = sum over the rows of dy[M][cdy]^T x[M][K] (partial matrices of `ksplit` row ranges; the caller sums them) */
int ldm_op_gemm_wgrad_f32(const float* dy, int cdy, const float* x, int K, float* dw, int cout, int64_t M, int ksplit, void* stream) {
    if (!dy || !x || !dw || M < 1 || M >= (1L << 31) || K < 4 || K % 4 || cdy % 4 || cout < 1 || cout > cdy || ksplit < 1 || ksplit > 64) return fail(LDM_ERR_BAD_ARG, "bad argument");
    Wgrad32Params q{}; q.dy = dy; q.cdy = cdy; q.x = x; q.cx = K; q.dw = dw; q.Cout = cout; q.Cin = K; q.dw_ld = K; q.dw_ci_off = 0;
    q.N = 1; q.Din = q.Dout = (int)M; q.Hin = q.Win = q.Hout = q.Wout = 1; q.ksize = 1; q.stride = 1; q.pad = 0; q.ups = 0; q.M = (int)M;
    q.co_tiles = (cout + 127) / 128; q.ci_tiles = (K + 127) / 128; q.ksplit = ksplit; q.slab_stride = (long)cout * K;
    hipLaunchKernelGGL(wgrad_f32_kernel, dim3(q.co_tiles * q.ci_tiles * q.ksplit), dim3(256), 0, (hipStream_t)stream, q);
    HIP_TRY(hipGetLastError());
    return 0;
}
static void gn32_slabs(int N, int C, int DHW, int* nslab, int* rps) {
    const int cvec = C / 4, rows_par = std::max(1, 256 / std::max(1, cvec));
    int ns = std::min((DHW + rows_par - 1) / rows_par, std::max(1, 512 / N));
    *rps = (DHW + ns - 1) / ns; *nslab = (DHW + *rps - 1) / *rps;
}
size_t ldm_op_group_norm_f32_scratch_bytes(int N, int C, int DHW, int groups) {
    int nslab = 1, rps = 1; gn32_slabs(std::max(1, N), std::max(4, C), std::max(1, DHW), &nslab, &rps);
    return ((size_t)N * nslab * C * 2 * 2 + (size_t)N * C * 2 + (size_t)N * groups * 4 + (size_t)N * C * 2) * 4 + 1024;
}
/* y = act(GroupNorm(x)) on fp32 NDHWC [N][DHW][C] (act 0 none, 1 SiLU, 2 LeakyReLU(0.2)); C % 4 == 0, C <= 1024 */
int ldm_op_group_norm_f32(const float* x, int C, const float* gamma, const float* beta, int groups, float eps, int act, float* out,
                          int N, int DHW, void* scratch, size_t scratch_bytes, void* stream) {
    if (!x || !gamma || !beta || !out || !scratch || N < 1 || DHW < 1 || C < 4 || C % 4 || C > 1024 || groups < 1 || C % groups || act < 0 || act > 2)
        return fail(LDM_ERR_BAD_ARG, "bad argument");
    if (scratch_bytes < ldm_op_group_norm_f32_scratch_bytes(N, C, DHW, groups)) return fail(LDM_ERR_WORKSPACE, "scratch too small");
    int nslab, rps; gn32_slabs(N, C, DHW, &nslab, &rps);
    float* partial = (float*)scratch; float* ab = partial + (size_t)N * nslab * C * 2;
    hipStream_t s = (hipStream_t)stream;
    Gn32Params p{}; p.xa = x; p.ca = C; p.DHW = DHW; p.N = N; p.nslab = nslab; p.rows_per_slab = rps; p.silu = act; p.partial = partial; p.ab = ab; p.out = out;
    hipLaunchKernelGGL(gn_stats_f32_kernel, dim3(nslab, N), dim3(256), 0, s, p);
    GnFinalizeParams fp{}; fp.partial = partial; fp.nslab = nslab; fp.C = C; fp.Creal = C; fp.groups = groups; fp.DHW = DHW; fp.eps = eps;
    fp.gamma = gamma; fp.beta = beta; fp.ab = ab;
    hipLaunchKernelGGL(gn_finalize_kernel, dim3(groups, N), dim3(256), 0, s, fp);
    hipLaunchKernelGGL(gn_apply_f32_kernel, dim3(grid_for((long)N * DHW * (C / 4), 256, 4096)), dim3(256), 0, s, p);
    HIP_TRY(hipGetLastError());
    return 0;
}
/* backward of the above: dx (fp32), dgamma / dbeta (fp32 [C], summed over the batch); recomputes the forward statistics */
int ldm_op_group_norm_bwd_f32(const float* dy, const float* x, int C, const float* gamma, const float* beta, int groups, float eps, int act,
                              float* dx, float* dgamma, float* dbeta, int N, int DHW, void* scratch, size_t scratch_bytes, void* stream) {
    if (!dy || !x || !gamma || !beta || !dx || !dgamma || !dbeta || !scratch || N < 1 || DHW < 1 || C < 4 || C % 4 || C > 1024 || groups < 1 || C % groups)
        return fail(LDM_ERR_BAD_ARG, "bad argument");
    if (scratch_bytes < ldm_op_group_norm_f32_scratch_bytes(N, C, DHW, groups)) return fail(LDM_ERR_WORKSPACE, "scratch too small");
    int nslab, rps; gn32_slabs(N, C, DHW, &nslab, &rps);
    float* partial = (float*)scratch;                       // [N][nslab][C][2] (forward stats), then reused by the backward sums
    float* partial2 = partial + (size_t)N * nslab * C * 2;  // [N][nslab][C][2]
    float* ab = partial2 + (size_t)N * nslab * C * 2;       // [N][C][2]
    float* mr = ab + (size_t)N * C * 2;                     // [N][G][2]
    float* gsum = mr + (size_t)N * groups * 2;              // [N][G][2]
    float* dgn = gsum + (size_t)N * groups * 2;             // [N][C]
    float* dbn = dgn + (size_t)N * C;                       // [N][C]
    hipStream_t s = (hipStream_t)stream;
    Gn32Params p{}; p.xa = x; p.ca = C; p.DHW = DHW; p.N = N; p.nslab = nslab; p.rows_per_slab = rps; p.partial = partial;
    hipLaunchKernelGGL(gn_stats_f32_kernel, dim3(nslab, N), dim3(256), 0, s, p);
    GnFinalizeParams fp{}; fp.partial = partial; fp.nslab = nslab; fp.C = C; fp.Creal = C; fp.groups = groups; fp.DHW = DHW; fp.eps = eps;
    fp.gamma = gamma; fp.beta = beta; fp.ab = ab; fp.mr = mr;
    hipLaunchKernelGGL(gn_finalize_kernel, dim3(groups, N), dim3(256), 0, s, fp);
    Gnb32Params q{}; q.dy = dy; q.xa = x; q.ca = C; q.ab = ab; q.mr = mr; q.gamma = gamma; q.groups = groups; q.DHW = DHW; q.N = N; q.silu = act;
    q.nslab = nslab; q.rows_per_slab = rps; q.partial = partial2; q.gsum = gsum; q.dxa = dx;
    GnBwdParams f{}; f.ca = C; f.cb = 0; f.gamma = gamma; f.groups = groups; f.DHW = DHW; f.N = N; f.nslab = nslab;
    f.partial = partial2; f.gsum = gsum; f.dgamma_n = dgn; f.dbeta_n = dbn;
    hipLaunchKernelGGL(gnb32_stats_kernel, dim3(nslab, N), dim3(256), 0, s, q);
    hipLaunchKernelGGL(gn_bwd_finalize_kernel, dim3(groups, N), dim3(256), 0, s, f);
    hipLaunchKernelGGL(gnb32_apply_kernel, dim3(grid_for((long)N * DHW * (C / 4), 256, 4096)), dim3(256), 0, s, q);
    hipLaunchKernelGGL(rowsum_n_kernel, dim3((C + 255) / 256), dim3(256), 0, s, (const float*)dgn, dgamma, N, C);
    hipLaunchKernelGGL(rowsum_n_kernel, dim3((C + 255) / 256), dim3(256), 0, s, (const float*)dbn, dbeta, N, C);
    HIP_TRY(hipGetLastError());
    return 0;
}

size_t ldm_op_scale_intensity_percentiles_scratch_bytes(int B) { return (size_t)(B < 1 ? 1 : B) * (4 * 4096 * 4 + sizeof(PctState)) + 256; }
int ldm_op_scale_intensity_percentiles(const float* x, float* out, int B, int64_t n, float lower, float upper, float b_min, float b_max,
                                       void* scratch, size_t scratch_bytes, void* stream) {
    if (!x || !out || !scratch || B < 1 || n < 1 || n >= (1LL << 32) || lower < 0.f || upper > 100.f || lower > upper) return fail(LDM_ERR_BAD_ARG, "bad argument");
    if (scratch_bytes < ldm_op_scale_intensity_percentiles_scratch_bytes(B)) return fail(LDM_ERR_WORKSPACE, "scratch too small");
    hipStream_t s = (hipStream_t)stream;
    unsigned* hist = (unsigned*)scratch;
    PctState* st = (PctState*)((char*)scratch + (size_t)B * 4 * 4096 * 4);
    HIP_TRY(hipMemsetAsync(hist, 0, (size_t)B * 4 * 4096 * 4, s));
    hipLaunchKernelGGL(pct_init_kernel, dim3((B + 63) / 64), dim3(64), 0, s, st, B, (long)n, (double)lower, (double)upper);
    const int gx = grid_for(n, 256 * 16, 512);
    for (int pass = 0; pass < 3; ++pass) {
        hipLaunchKernelGGL(pct_hist_kernel, dim3(gx, B), dim3(256), 0, s, x, (long)n, (const PctState*)st, hist, pass);
        hipLaunchKernelGGL(pct_scan_kernel, dim3(B), dim3(256), 0, s, st, hist, pass);
    }
    hipLaunchKernelGGL(pct_apply_kernel, dim3(grid_for(n, 256 * 4, 2048), B), dim3(256), 0, s, x, out, (long)n, (const PctState*)st, b_min, b_max);
    HIP_TRY(hipGetLastError());
    return 0;
}

/* Weights of the data-gradient conv: wt[tap'][ci][co] = w[taps-1-tap'][co][ci].  w: [taps][cout_pad][cin] bf16,
 * wt: [taps][round64(cin)][round32(cout)] bf16 (device memory, caller allocated). */
int ldm_op_weight_flip_transpose(const void* w, void* wt, int ksize, int cout, int cout_pad, int cin, void* stream) {
    if (!w || !wt || (ksize != 1 && ksize != 3) || cout < 1 || cin < 1 || cout_pad < cout) return fail(LDM_ERR_BAD_ARG, "bad argument");
    const int taps = ksize * ksize * ksize, rows = rup(cin, 64), cols = rup(cout, 32);
    hipLaunchKernelGGL(weight_flip_transpose_kernel, dim3((cols + 63) / 64, (rows + 63) / 64, taps), dim3(256), 0, (hipStream_t)stream,
                       (const bf16_t*)w, (bf16_t*)wt, taps, cout, cout_pad, cin, rows, 0, cin);
    HIP_TRY(hipGetLastError());
    return 0;
}

/* dW[tap][co][ci] (fp32, [k^3][cout][cin]) of y = conv3d(x, w): dy [M][cdy] and x [rows][cx] are NDHWC bf16.
 * ksplit > 1 splits the voxel range: dw then holds ksplit partial matrices [ksplit][k^3][cout][cin] to be summed. */
int ldm_op_conv3d_wgrad(const void* dy, int cdy, const void* x, int cx, float* dw, int cout, int cin,
                        int N, int Din, int Hin, int Win, int ksize, int stride, int pad, int ups, int ksplit, void* stream) {
    if (!dy || !x || !dw) return fail(LDM_ERR_BAD_ARG, "null tensor argument");
    if (cdy % 32 || cx % 32 || cout < 1 || cin < 1 || cout > cdy || cin > cx) return fail(LDM_ERR_BAD_ARG, "bad channel counts");
    if ((ksize != 1 && ksize != 3) || stride < 1 || stride > 2 || ups < 0 || ups > 1) return fail(LDM_ERR_UNSUPPORTED, "ksize 1|3, stride 1|2, ups 0|1");
    const int Du = Din << ups, Hu = Hin << ups, Wu = Win << ups;
    const int pad_total = (stride == 2 && pad == 0 && ksize == 3) ? 1 : 2 * pad;
    const int Do = (Du + pad_total - ksize) / stride + 1, Ho = (Hu + pad_total - ksize) / stride + 1, Wo = (Wu + pad_total - ksize) / stride + 1;
    const long M = (long)N * Do * Ho * Wo;
    if (M < 1 || M >= (1L << 31)) return fail(LDM_ERR_BAD_ARG, "bad output size");
    if ((long)M * cdy * 2 >= (1L << 32) || (long)N * Din * Hin * Win * cx * 2 >= (1L << 32)) return fail(LDM_ERR_UNSUPPORTED, "tensor exceeds 4 GiB");
    WgradParams p{}; p.dy = (const bf16_t*)dy; p.cdy = cdy; p.x = (const bf16_t*)x; p.cx = cx; p.dw = dw; p.Cout = cout; p.Cin = cin; p.dw_ld = cin; p.dw_ci_off = 0;
    p.N = N; p.Din = Din; p.Hin = Hin; p.Win = Win; p.Dout = Do; p.Hout = Ho; p.Wout = Wo; p.ksize = ksize; p.stride = stride; p.pad = pad; p.ups = ups;
    p.M = (int)M; p.co_tiles = (cout + 127) / 128; p.ci_tiles = (cin + 127) / 128;
    if (ksplit < 1 || ksplit > 64) return fail(LDM_ERR_BAD_ARG, "ksplit must be in 1..64");
    p.ksplit = ksplit; p.slab_stride = (long)ksize * ksize * ksize * cout * cin;
    LDM_TRY(launch_wgrad(p, (hipStream_t)stream));
    HIP_TRY(hipGetLastError());
    return 0;
}

size_t ldm_op_group_norm_bwd_scratch_bytes(int N, int C, int DHW, int groups) {
    return ldm_op_group_norm_scratch_bytes(N, C, DHW) * 2 + ((size_t)N * groups * 4 + (size_t)N * C * 2) * 4 + 1024;
}

/* Backward of y = act(GroupNorm(cat(xa, xb))): dxa/dxb (bf16, += acc_a/acc_b when given), dgamma/dbeta (fp32 [C], summed
 * over the batch).  Recomputes the forward statistics (the training plan saves them instead). */
int ldm_op_group_norm_bwd(const void* dy, const void* xa, int ca, const void* xb, int cb, const float* gamma, const float* beta,
                          int groups, float eps, int silu, const void* acc_a, const void* acc_b, void* dxa, void* dxb,
                          float* dgamma, float* dbeta, int N, int DHW, void* scratch, size_t scratch_bytes, void* stream) {
    if (!dy || !xa || !gamma || !beta || !dxa || !dgamma || !dbeta || !scratch) return fail(LDM_ERR_BAD_ARG, "null tensor argument");
    if (!xb) cb = 0;
    const int C = ca + cb;
    if (C % 8 || ca % 8 || groups < 1 || C % groups || (cb && !dxb)) return fail(LDM_ERR_BAD_ARG, "bad channel / group counts");
    if (scratch_bytes < ldm_op_group_norm_bwd_scratch_bytes(N, C, DHW, groups)) return fail(LDM_ERR_WORKSPACE, "scratch too small");
    const int cvec = C / 8, rows_par = std::max(1, 256 / cvec);
    int nslab = std::min((DHW + rows_par - 1) / rows_par, std::max(1, 512 / N));
    const int rps = (DHW + nslab - 1) / nslab; nslab = (DHW + rps - 1) / rps;
    float* partial = (float*)scratch;                       // [N][nslab][C][2]
    float* ab = partial + (size_t)N * nslab * C * 2;        // [N][C][2]
    float* mr = ab + (size_t)N * C * 2;                     // [N][G][2]
    float* gsum = mr + (size_t)N * groups * 2;              // [N][G][2]
    float* dgn = gsum + (size_t)N * groups * 2;             // [N][C]
    float* dbn = dgn + (size_t)N * C;                       // [N][C]
    hipStream_t s = (hipStream_t)stream;
    GnStatsParams sp{}; sp.xa = (const bf16_t*)xa; sp.xb = (const bf16_t*)xb; sp.ca = ca; sp.cb = cb; sp.DHW = DHW; sp.nslab = nslab;
    sp.rows_per_slab = rps; sp.partial = partial;
    hipLaunchKernelGGL(gn_stats_kernel, dim3(nslab, N), dim3(256), 0, s, sp);
    GnFinalizeParams fp{}; fp.partial = partial; fp.nslab = nslab; fp.C = C; fp.Creal = C; fp.groups = groups; fp.DHW = DHW; fp.eps = eps;
    fp.gamma = gamma; fp.beta = beta; fp.ab = ab; fp.mr = mr;
    hipLaunchKernelGGL(gn_finalize_kernel, dim3(groups, N), dim3(256), 0, s, fp);
    GnBwdParams bp{}; bp.dy = (const bf16_t*)dy; bp.xa = sp.xa; bp.xb = sp.xb; bp.ca = ca; bp.cb = cb; bp.ab = ab; bp.mr = mr; bp.gamma = gamma;
    bp.groups = groups; bp.DHW = DHW; bp.N = N; bp.silu = silu; bp.nslab = nslab; bp.rows_per_slab = rps; bp.partial = partial; bp.gsum = gsum;
    bp.dgamma_n = dgn; bp.dbeta_n = dbn; bp.acc_a = (const bf16_t*)acc_a; bp.acc_b = (const bf16_t*)acc_b; bp.dxa = (bf16_t*)dxa; bp.dxb = (bf16_t*)dxb;
    hipLaunchKernelGGL(gn_bwd_stats_kernel, dim3(nslab, N), dim3(256), 0, s, bp);
    hipLaunchKernelGGL(gn_bwd_finalize_kernel, dim3(groups, N), dim3(256), 0, s, bp);
    hipLaunchKernelGGL(gn_bwd_apply_kernel, dim3(grid_for((long)N * DHW * cvec, 256, 2048)), dim3(256), 0, s, bp);
    // parameter gradients: sum the per-sample rows (reuses the column-sum finalize with nslab = 1 layout [N][1][C][2]? no: plain loop)
    hipLaunchKernelGGL(rowsum_n_kernel, dim3((C + 255) / 256), dim3(256), 0, s, (const float*)dgn, dgamma, N, C);
    hipLaunchKernelGGL(rowsum_n_kernel, dim3((C + 255) / 256), dim3(256), 0, s, (const float*)dbn, dbeta, N, C);
    HIP_TRY(hipGetLastError());
    return 0;
}

static bool head_dim_ok(int C, int d) { return (d == 32 || d == 64 || d == 128 || d == 256) && C >= d && C % d == 0; }
/* head_dim = 32 | 64 | 128 | 256 (C % head_dim == 0); lse may be NULL */
int ldm_op_attention_hd(const void* qkv, void* out, float* lse, int B, int N, int C, int head_dim, void* stream) {
    if (!qkv || !out || B < 1 || N < 1 || !head_dim_ok(C, head_dim)) return fail(LDM_ERR_BAD_ARG, "bad argument (head_dim 32|64|128|256, C % head_dim == 0)");
    AttnParams p{}; p.qkv = (const bf16_t*)qkv; p.out = (bf16_t*)out; p.B = B; p.N = N; p.C = C; p.d = head_dim; p.heads = C / head_dim;
    p.scale = 1.0f / sqrtf((float)head_dim); p.lse = lse;
    HIP_TRY(launch_attn_fwd(p, (hipStream_t)stream));
    HIP_TRY(hipGetLastError());
    return 0;
}
int ldm_op_attention_bwd_hd(const void* qkv, const void* o, const void* d_o, const float* lse, float* delta_scratch, void* dqkv,
                            int B, int N, int C, int head_dim, void* stream) {
    if (!qkv || !o || !d_o || !lse || !delta_scratch || !dqkv || B < 1 || N < 1 || !head_dim_ok(C, head_dim)) return fail(LDM_ERR_BAD_ARG, "bad argument");
    AttnBwdParams p{}; p.qkv = (const bf16_t*)qkv; p.o = (const bf16_t*)o; p.d_o = (const bf16_t*)d_o; p.lse = lse; p.delta = delta_scratch;
    p.dqkv = (bf16_t*)dqkv; p.B = B; p.N = N; p.C = C; p.d = head_dim; p.heads = C / head_dim; p.scale = 1.0f / sqrtf((float)head_dim);
    HIP_TRY(launch_attn_bwd(p, (hipStream_t)stream));
    return 0;
}
int ldm_op_attention(const void* qkv, void* out, int B, int N, int C, void* stream) {
    return ldm_op_attention_hd(qkv, out, nullptr, B, N, C, 64, stream);
}

/* attention forward that also saves the log-sum-exp rows, and its backward: dqkv [B*N][3C] from dO [B*N][C] (head_dim 64). */
int ldm_op_attention_train(const void* qkv, void* out, float* lse, int B, int N, int C, void* stream) {
    if (!lse) return fail(LDM_ERR_BAD_ARG, "bad argument");
    return ldm_op_attention_hd(qkv, out, lse, B, N, C, 64, stream);
}
int ldm_op_attention_bwd(const void* qkv, const void* o, const void* d_o, const float* lse, float* delta_scratch, void* dqkv,
                         int B, int N, int C, void* stream) {
    return ldm_op_attention_bwd_hd(qkv, o, d_o, lse, delta_scratch, dqkv, B, N, C, 64, stream);
}

// ---- RCCL (loaded lazily so the library itself has no hard dependency on librccl) -----------------------
typedef struct { char internal[128]; } rccl_uid;
typedef void* rccl_comm_t;
struct RcclApi {
    void* lib = nullptr;
    int (*GetUniqueId)(rccl_uid*) = nullptr;
    int (*CommInitRank)(rccl_comm_t*, int, rccl_uid, int) = nullptr;
    int (*AllReduce)(const void*, void*, size_t, int, int, rccl_comm_t, hipStream_t) = nullptr;
    int (*Broadcast)(const void*, void*, size_t, int, int, rccl_comm_t, hipStream_t) = nullptr;
    int (*CommDestroy)(rccl_comm_t) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    int (*GetVersion)(int*) = nullptr;
};
static RcclApi g_rccl;
static int rccl_load() {
    if (g_rccl.lib) return 0;
    // the copy the process has already loaded (torch links its own librccl) comes first: two different RCCL builds in one process
    // would each bring their own kernels and IPC state
    void* h = dlopen("librccl.so", RTLD_NOW | RTLD_NOLOAD);
    if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD);
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("/opt/rocm/lib/librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) return fail(LDM_ERR_RCCL, "cannot load librccl.so: %s", dlerror());
    g_rccl.GetUniqueId = (int (*)(rccl_uid*))dlsym(h, "ncclGetUniqueId");
    g_rccl.CommInitRank = (int (*)(rccl_comm_t*, int, rccl_uid, int))dlsym(h, "ncclCommInitRank");
    g_rccl.AllReduce = (int (*)(const void*, void*, size_t, int, int, rccl_comm_t, hipStream_t))dlsym(h, "ncclAllReduce");
    g_rccl.Broadcast = (int (*)(const void*, void*, size_t, int, int, rccl_comm_t, hipStream_t))dlsym(h, "ncclBroadcast");
    g_rccl.CommDestroy = (int (*)(rccl_comm_t))dlsym(h, "ncclCommDestroy");
    g_rccl.GetErrorString = (const char* (*)(int))dlsym(h, "ncclGetErrorString");
    g_rccl.GetVersion = (int (*)(int*))dlsym(h, "ncclGetVersion");
    if (!g_rccl.GetUniqueId || !g_rccl.CommInitRank || !g_rccl.AllReduce || !g_rccl.Broadcast || !g_rccl.CommDestroy)
        return fail(LDM_ERR_RCCL, "librccl.so lacks a required symbol");
    g_rccl.lib = h;
    return 0;
}
#define RCCL_TRY(x) do { int r_ = (x); if (r_ != 0) return fail(LDM_ERR_RCCL, "%s failed: %s", #x, \
    g_rccl.GetErrorString ? g_rccl.GetErrorString(r_) : "?"); } while (0)

struct ldm_comm { rccl_comm_t comm = nullptr; int rank = 0, world = 1; float* token = nullptr;
                  ldm_allreduce_fn fn = nullptr; void* user = nullptr;      // fn: caller-supplied transport (ldm_comm_init_custom)
                  int64_t n_allreduce = 0, bytes_allreduce = 0, n_broadcast = 0, bytes_broadcast = 0; };   // as handed to the transport

int ldm_comm_unique_id(char id[128]) {
    if (!id) return fail(LDM_ERR_BAD_ARG, "null id");
    LDM_TRY(rccl_load());
    rccl_uid u; RCCL_TRY(g_rccl.GetUniqueId(&u)); memcpy(id, u.internal, 128);
    return 0;
}
int ldm_comm_init(int rank, int world, const char id[128], ldm_comm** out) {
    if (!id || !out || world < 1 || rank < 0 || rank >= world) return fail(LDM_ERR_BAD_ARG, "bad argument");
    LDM_TRY(rccl_load());
    std::unique_ptr<ldm_comm> c(new ldm_comm()); c->rank = rank; c->world = world;
    rccl_uid u; memcpy(u.internal, id, 128);
    RCCL_TRY(g_rccl.CommInitRank(&c->comm, world, u, rank));
    HIP_TRY(hipMalloc((void**)&c->token, 256));
    HIP_TRY(hipMemset(c->token, 0, 256));
    HIP_TRY(hipDeviceSynchronize());
    *out = c.release();
    return 0;
}
/* A communicator whose all-reduce is the caller's function instead of RCCL: fn(user, buf, count, dtype, op, stream) must leave the
 * reduction over the `world` ranks in buf, ordered on `stream` (dtype / op as for ldm_comm_allreduce; non-zero return = failure).
 * Everything above the transport -- which ranges of the gradient buffer are handed over, when, on which stream, with which op, and
 * the join in front of the optimizer -- is the same code as with RCCL, so a test can stand in for the peers of a world > 1 job on one
 * GPU.  ldm_comm_broadcast is not available on such a communicator. */
int ldm_comm_init_custom(int rank, int world, ldm_allreduce_fn fn, void* user, ldm_comm** out) {
    if (!fn || !out || world < 1 || rank < 0 || rank >= world) return fail(LDM_ERR_BAD_ARG, "bad argument");
    std::unique_ptr<ldm_comm> c(new ldm_comm()); c->rank = rank; c->world = world; c->fn = fn; c->user = user;
    *out = c.release();
    return 0;
}
// ncclDataType: float32 = 7, bfloat16 = 9; ncclRedOp: sum = 0, avg = 4
int ldm_comm_allreduce(ldm_comm* c, void* buf, int64_t count, int dtype, int op, void* stream) {
    if (!c || !buf || count < 0 || dtype < 0 || dtype > 1 || op < 0 || op > 1) return fail(LDM_ERR_BAD_ARG, "bad argument");
    c->n_allreduce += 1; c->bytes_allreduce += count * (dtype == 0 ? 4 : 2);
    if (c->fn) {
        const int r = c->fn(c->user, buf, count, dtype, op, stream);
        return r ? fail(LDM_ERR_RCCL, "custom all-reduce returned %d", r) : 0;
    }
    RCCL_TRY(g_rccl.AllReduce(buf, buf, (size_t)count, dtype == 0 ? 7 : 9, op == 0 ? 0 : 4, c->comm, (hipStream_t)stream));
    return 0;
}
/* What this communicator has carried so far, counted where the bytes are handed to the transport (RCCL or the custom function):
 * out = {all-reduce calls, all-reduce bytes in the wire dtype, broadcast calls, broadcast bytes}. */
int ldm_comm_stats(const ldm_comm* c, int64_t out[4]) {
    if (!c || !out) return fail(LDM_ERR_BAD_ARG, "bad argument");
    out[0] = c->n_allreduce; out[1] = c->bytes_allreduce; out[2] = c->n_broadcast; out[3] = c->bytes_broadcast;
    return 0;
}
/* 1 if the communicator's transport is RCCL (ncclAllReduce), 0 for a caller-supplied function */
int ldm_comm_is_rccl(const ldm_comm* c) { return c ? (c->fn ? 0 : 1) : -1; }
/* ncclGetVersion of the librccl this process uses (e.g. 22606), or a negative status when it cannot be loaded */
int ldm_comm_rccl_version(void) {
    LDM_TRY(rccl_load());
    int v = 0;
    if (!g_rccl.GetVersion) return fail(LDM_ERR_RCCL, "librccl.so lacks ncclGetVersion");
    RCCL_TRY(g_rccl.GetVersion(&v));
    return v;
}
int ldm_comm_broadcast(ldm_comm* c, void* buf, int64_t count, int dtype, int root, void* stream) {
    if (!c || !buf || count < 0 || dtype < 0 || dtype > 1 || root < 0 || root >= c->world) return fail(LDM_ERR_BAD_ARG, "bad argument");
    if (c->fn) return fail(LDM_ERR_UNSUPPORTED, "broadcast on a custom-transport communicator");
    c->n_broadcast += 1; c->bytes_broadcast += count * (dtype == 0 ? 4 : 2);
    RCCL_TRY(g_rccl.Broadcast(buf, buf, (size_t)count, dtype == 0 ? 7 : 9, root, c->comm, (hipStream_t)stream));
    return 0;
}
int ldm_comm_barrier(ldm_comm* c, void* stream) {
    if (!c) return fail(LDM_ERR_BAD_ARG, "null comm");
    if (c->fn) { HIP_TRY(hipStreamSynchronize((hipStream_t)stream)); return 0; }
    RCCL_TRY(g_rccl.AllReduce(c->token, c->token, 1, 7, 0, c->comm, (hipStream_t)stream));
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    return 0;
}
void ldm_comm_destroy(ldm_comm* c) {
    if (!c) return;
    if (c->comm && g_rccl.CommDestroy) g_rccl.CommDestroy(c->comm);
    if (c->token) (void)hipFree(c->token);
    delete c;
}
int ldm_comm_rank(const ldm_comm* c) { return c ? c->rank : -1; }
int ldm_comm_world(const ldm_comm* c) { return c ? c->world : -1; }

/* Attach (comm != NULL) or detach (NULL) the data-parallel gradient exchange of a model's training plans.  While attached,
 * ldm_unet_train_backward / ldm_vae_train_backward all-reduce (mean over ranks, fp32) the flat gradient buffer in buckets of
 * LDM_GRAD_BUCKET_MB (default 48) MB: a bucket's collective is queued on the communicator's own stream as soon as the backward pass
 * has finished that tail range of the buffer, overlapped with the rest of backward, and `stream` waits for the last one before the call's
 * work is complete in stream order.  Replaces DistributedDataParallel's bucketed hooks (3d_ldm/train_diffusion.py:147-149).
 * The communicator must outlive the attachment. */
int ldm_model_set_grad_sync(ldm_model* m, ldm_comm* comm) {
    if (!m) return fail(LDM_ERR_BAD_ARG, "null model");
    if (comm && !m->gsync.stream) HIP_TRY(hipStreamCreateWithFlags(&m->gsync.stream, hipStreamNonBlocking));
    m->gsync.comm = comm;
    return 0;
}
/* Wire format of the bucketed exchange: 0 = fp32 (default: what DistributedDataParallel reduces, 3d_ldm/train_diffusion.py:147-149),
 * 1 = bf16: on the communicator's stream every bucket is cast into a bf16 staging slice, all-reduced (avg) as bf16 and cast back into
 * the fp32 gradient buffer -- half the bytes per step for a ring that is xGMI-link bound (765 -> 382 MB for the benchmark UNet).
 * The staging slice (2 bytes per element of the largest bucket) is allocated at the first backward that needs it. */
int ldm_model_set_grad_wire(ldm_model* m, int dtype) {
    if (!m || dtype < 0 || dtype > 1) return fail(LDM_ERR_BAD_ARG, "wire dtype must be 0 (fp32) or 1 (bf16)");
    m->gsync.wire = dtype;
    return 0;
}
/* Buckets handed to the communicator's stream that no join has covered yet (0 = the exchange is quiescent: every collective of the
 * library's communicator is ordered in front of whatever the launch stream does next).  Host-side bookkeeping, no synchronisation. */
int ldm_model_grad_sync_pending(const ldm_model* m) { return m ? m->gsync.pending : -1; }
/* Timeline of the buckets of the LAST backward call (synchronises): issue_ms[k] = when bucket k was handed to the comm stream,
 * done_ms[k] = when its all-reduce finished, both relative to the start of that backward call; elems[k] = its size; then, at index
 * n (if max > n), the end of the backward call itself in issue_ms[n] (done_ms[n] = the same, elems[n] = 0).  Returns n. */
int ldm_model_grad_sync_trace(ldm_model* m, double* issue_ms, double* done_ms, int64_t* elems, int max) {
    if (!m || !issue_ms || !done_ms || !elems || max < 0) return fail(LDM_ERR_BAD_ARG, "bad argument");
    GradSyncState& g = m->gsync;
    const int n = (int)g.elems.size();
    if (g.used < 2 + 2 * (size_t)n) return 0;
    HIP_TRY(hipEventSynchronize(g.ev[1]));
    for (int k = 0; k < n && k < max; ++k) {
        float a = 0.f, b = 0.f;
        HIP_TRY(hipEventSynchronize(g.ev[2 + 2 * k + 1]));
        HIP_TRY(hipEventElapsedTime(&a, g.ev[0], g.ev[2 + 2 * k])); HIP_TRY(hipEventElapsedTime(&b, g.ev[0], g.ev[2 + 2 * k + 1]));
        issue_ms[k] = a; done_ms[k] = b; elems[k] = g.elems[k];
    }
    if (max > n) { float e = 0.f; HIP_TRY(hipEventElapsedTime(&e, g.ev[0], g.ev[1])); issue_ms[n] = done_ms[n] = e; elems[n] = 0; }
    return n;
}

/* The gradient-exchange schedule of a training plan, readable without a GPU (plans are built on the host): one event per entry, in
 * launch order.  kind[k]: 0 = an op writes flat_grads[lo, lo + n) (final values), 1 = bucket: [lo, lo + n) is handed to the
 * communicator here, 2 = join (the launch stream waits for every bucket; the optimizer may follow), 3 = a write the dump does not
 * understand (a test failure).  op[k] = index of the launch-plan op.  Returns the number of events (may exceed max; only max are
 * written).  tests/test_grad_schedule_cpu.py checks: buckets tile [0, total) exactly once, no write into a range after its bucket
 * was issued, every element written before its bucket, join last. */
int ldm_model_grad_schedule(ldm_model* m, int B, int D, int H, int W, int* kind, int64_t* lo, int64_t* n, int* op, int max) {
    if (!m || max < 0 || (max > 0 && (!kind || !lo || !n || !op))) return fail(LDM_ERR_BAD_ARG, "bad argument");
    std::shared_ptr<Plan> p; LDM_TRY(get_plan(m, "train", B, D, H, W, &p));
    int cnt = 0;
    auto put = [&](int k, int64_t a, int64_t c, size_t oi) { if (cnt < max) { kind[cnt] = k; lo[cnt] = a; n[cnt] = c; op[cnt] = (int)oi; } ++cnt; };
    const ExportDesc* ed = (const ExportDesc*)p->exp_tab.h_descs.data(); const int2* em = (const int2*)p->exp_tab.h_map.data();
    for (size_t oi = p->bwd_begin; oi < p->ops.size(); ++oi) {
        const Op& o = p->ops[oi];
        switch (o.kind) {
            case OP_EXPORT_BATCH: {
                int last = -1;
                for (int b = o.i[0]; b < o.i[1]; ++b) if (em[b].x != last) {
                    last = em[b].x; const ExportDesc& e = ed[last];
                    put(0, e.dst_off, (int64_t)e.taps * e.cout * e.cin, oi);
                }
                break;
            }
            case OP_GNB: put(0, o.i[8], o.i[0] + o.i[1], oi); put(0, o.i[9], o.i[0] + o.i[1], oi); break;
            case OP_LIN_DW:
                if (o.r[2].base == BASE_IO4) put(0, (int64_t)(o.r[2].off / 4), (int64_t)o.i[1] * o.i[2], oi);
                if (o.r[3].base == BASE_IO4) put(0, (int64_t)(o.r[3].off / 4), o.i[2], oi);
                break;
            case OP_BUCKET: put(1, o.i[0], o.i[1], oi); break;
            case OP_BUCKET_JOIN: put(2, 0, 0, oi); break;
            default:
                for (const Ref& r : o.r) if (r.base == BASE_IO4) put(3, (int64_t)(r.off / 4), 0, oi);
        }
    }
    return cnt;
}

}  // extern "C"

// ---- gradient buckets (run_plan: OP_BUCKET / OP_BUCKET_JOIN) ----------------------------------------------------------------------
static hipEvent_t* gs_event(GradSyncState& g, size_t k) {
    while (g.ev.size() <= k) { hipEvent_t e; if (hipEventCreate(&e) != hipSuccess) return nullptr; g.ev.push_back(e); }
    return &g.ev[k];
}
static int grad_sync_begin(GradSyncState& g, hipStream_t s) {
    // a previous backward that returned an error between a bucket and its join left buckets un-joined: this backward is ordered behind
    // them (the comm stream is in order: its last done-event covers all) and the bookkeeping starts clean
    if (g.pending > 0 && g.used > 2) HIP_TRY(hipStreamWaitEvent(s, g.ev[g.used - 1], 0));
    g.pending = 0;
    g.used = 2; g.elems.clear();                     // ev[0] = start of this backward, ev[1] = its end (recorded by the join)
    hipEvent_t* e0 = gs_event(g, 1); if (!e0) return fail(LDM_ERR_HIP, "hipEventCreate failed");
    HIP_TRY(hipEventRecord(g.ev[0], s));
    return 0;
}
static int grad_sync_bucket(GradSyncState& g, float* buf, int64_t count, hipStream_t s) {
    if (!buf) return fail(LDM_ERR_BAD_ARG, "backward without a gradient buffer");
    if (!gs_event(g, g.used + 1)) return fail(LDM_ERR_HIP, "hipEventCreate failed");
    hipEvent_t issue = g.ev[g.used], done = g.ev[g.used + 1];
    HIP_TRY(hipEventRecord(issue, s));
    HIP_TRY(hipStreamWaitEvent(g.stream, issue, 0));
    if (g.wire == 1) {                                                                // bf16 on the wire: cast -> all-reduce(avg) -> cast back
        const size_t need = (size_t)count * 2;
        if (g.stage_bytes < need) {                                                   // first backward (or a larger bucket): grow once
            HIP_TRY(hipStreamSynchronize(g.stream));
            if (g.stage) (void)hipFree(g.stage);
            g.stage = nullptr; g.stage_bytes = 0;
            HIP_TRY(hipMalloc(&g.stage, rup_sz(need, 256)));
            g.stage_bytes = rup_sz(need, 256);
        }
        const int nb = grid_for((count + 7) / 8, 256, 2048);
        hipLaunchKernelGGL(grad_wire_pack_kernel, dim3(nb), dim3(256), 0, g.stream, (const float*)buf, (bf16_t*)g.stage, (long)count);
        LDM_TRY(ldm_comm_allreduce(g.comm, g.stage, count, 1, 1, g.stream));
        hipLaunchKernelGGL(grad_wire_unpack_kernel, dim3(nb), dim3(256), 0, g.stream, (const bf16_t*)g.stage, buf, (long)count);
        HIP_TRY(hipGetLastError());
    } else
        LDM_TRY(ldm_comm_allreduce(g.comm, buf, count, 0, 1, g.stream));             // fp32, mean over ranks
    HIP_TRY(hipEventRecord(done, g.stream));
    g.used += 2; g.elems.push_back(count); g.pending += 1;
    return 0;
}
static int grad_sync_join(GradSyncState& g, hipStream_t s) {
    if (g.used > 2) HIP_TRY(hipStreamWaitEvent(s, g.ev[g.used - 1], 0));             // the comm stream is in order: the last bucket covers all
    HIP_TRY(hipEventRecord(g.ev[1], s));
    g.pending = 0;
    return 0;
}
