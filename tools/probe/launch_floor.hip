// What one dependent kernel boundary costs on this box, in the forms the launch plans use: a chain of K dependent kernels on one
// stream, eager and as one captured graph, for (a) a kernel that does nothing, (b) one that reads and writes a small tensor
// (every block loads 16 B per thread, adds one, stores it back: a GroupNorm-sized hand-off at 6^3: 216 x 512 bf16 = 221 KB),
// (c) the same over 7 MB (24^3 x 256 bf16), with plain and with write-through (sc1) stores.
//   hipcc --offload-arch=gfx950 -O3 tools/probe/launch_floor.hip -o tools/probe/launch_floor && tools/probe/launch_floor
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void k_empty() {}
__global__ void k_touch(const uint4* __restrict__ in, uint4* __restrict__ out, long n) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        uint4 v = in[i]; v.x += 1; out[i] = v;
    }
}
__global__ void k_touch_wt(const uint4* __restrict__ in, uint4* __restrict__ out, long n) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
        const uint4 t = in[i]; u32x4 v = {t.x + 1, t.y, t.z, t.w};
        asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"((void*)(out + i)), "v"(v) : "memory");
    }
}

// block b reads the range block (b + shift) wrote in the previous launch: shift % 8 != 0 = produced on ANOTHER XCD (round-robin
// placement), shift % 8 == 0 = same XCD, another CU.  wt: write-through stores.
template <bool WT>
__global__ void k_shift(const uint4* __restrict__ in, uint4* __restrict__ out, int per_block, int shift) {
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    const int src = (blockIdx.x + shift) % gridDim.x;
    for (int i = threadIdx.x; i < per_block; i += blockDim.x) {
        const uint4 t = in[(long)src * per_block + i];
        u32x4 v = {t.x + 1, t.y, t.z, t.w};
        if (WT) asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"((void*)(out + (long)blockIdx.x * per_block + i)), "v"(v) : "memory");
        else out[(long)blockIdx.x * per_block + i] = make_uint4(v.x, v.y, v.z, v.w);
    }
}
// two dependent round trips inside the kernel: an index read, then the row it names (a slab fold followed by an apply has this shape)
__global__ void k_chain2(const uint4* __restrict__ in, uint4* __restrict__ out, int per_block, int shift) {
    const int src = (blockIdx.x + shift) % gridDim.x;
    const uint4 first = in[(long)src * per_block];
    const int hop = (int)(first.y & 7u);                       // data dependent (always < 8)
    for (int i = threadIdx.x; i < per_block; i += blockDim.x) {
        uint4 t = in[(long)((src + hop) % gridDim.x) * per_block + i];
        t.x += 1; t.y = 0;
        out[(long)blockIdx.x * per_block + i] = t;
    }
}

template <class F> static double run(hipStream_t s, int K, int reps, bool graph, F launch) {
    hipGraphExec_t exec = nullptr;
    if (graph) {
        hipGraph_t g;
        hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
        for (int k = 0; k < K; ++k) launch(k);
        hipStreamEndCapture(s, &g);
        hipGraphInstantiate(&exec, g, nullptr, nullptr, 0);
        hipGraphDestroy(g);
    }
    auto once = [&] { if (graph) hipGraphLaunch(exec, s); else for (int k = 0; k < K; ++k) launch(k); };
    for (int r = 0; r < 3; ++r) once();
    hipStreamSynchronize(s);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipEventRecord(a, s);
    for (int r = 0; r < reps; ++r) once();
    hipEventRecord(b, s);
    hipEventSynchronize(b);
    float ms = 0.f; hipEventElapsedTime(&ms, a, b);
    if (exec) hipGraphExecDestroy(exec);
    return ms * 1e3 / (reps * K);
}

int main() {
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    const long big = 7L << 20, small = 221184;
    char *a, *b; CK(hipMalloc(&a, big)); CK(hipMalloc(&b, big)); CK(hipMemset(a, 1, big)); CK(hipMemset(b, 1, big));
    const int K = 100, reps = 20;
    for (int graph = 0; graph < 2; ++graph) {
        const char* tag = graph ? "graph" : "eager";
        printf("%s  empty kernel, 1 block          : %6.2f us per launch\n", tag, run(s, K, reps, graph, [&](int) { hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, s); }));
        printf("%s  empty kernel, 256 blocks       : %6.2f us per launch\n", tag, run(s, K, reps, graph, [&](int) { hipLaunchKernelGGL(k_empty, dim3(256), dim3(256), 0, s); }));
        printf("%s  221 KB ping-pong, plain stores : %6.2f us per launch\n", tag, run(s, K, reps, graph, [&](int k) {
            hipLaunchKernelGGL(k_touch, dim3(54), dim3(256), 0, s, (const uint4*)(k & 1 ? b : a), (uint4*)(k & 1 ? a : b), small / 16); }));
        printf("%s  221 KB ping-pong, sc1 stores   : %6.2f us per launch\n", tag, run(s, K, reps, graph, [&](int k) {
            hipLaunchKernelGGL(k_touch_wt, dim3(54), dim3(256), 0, s, (const uint4*)(k & 1 ? b : a), (uint4*)(k & 1 ? a : b), small / 16); }));
        printf("%s  7 MB ping-pong, plain stores   : %6.2f us per launch\n", tag, run(s, K, reps, graph, [&](int k) {
            hipLaunchKernelGGL(k_touch, dim3(1024), dim3(256), 0, s, (const uint4*)(k & 1 ? b : a), (uint4*)(k & 1 ? a : b), big / 16); }));
        printf("%s  7 MB ping-pong, sc1 stores     : %6.2f us per launch\n", tag, run(s, K, reps, graph, [&](int k) {
            hipLaunchKernelGGL(k_touch_wt, dim3(1024), dim3(256), 0, s, (const uint4*)(k & 1 ? b : a), (uint4*)(k & 1 ? a : b), big / 16); }));
        printf("%s  7 MB ping-pong, sc1, 2048 blk  : %6.2f us per launch\n", tag, run(s, K, reps, graph, [&](int k) {
            hipLaunchKernelGGL(k_touch_wt, dim3(2048), dim3(256), 0, s, (const uint4*)(k & 1 ? b : a), (uint4*)(k & 1 ? a : b), big / 16); }));
        for (int shift : {0, 8, 1, 3}) {
            printf("%s  221 KB, block b reads what block b+%d wrote, plain : %6.2f us per launch\n", tag, shift, run(s, K, reps, graph, [&](int k) {
                hipLaunchKernelGGL(k_shift<false>, dim3(54), dim3(256), 0, s, (const uint4*)(k & 1 ? b : a), (uint4*)(k & 1 ? a : b), 256, shift); }));
            printf("%s  221 KB, block b reads what block b+%d wrote, sc1   : %6.2f us per launch\n", tag, shift, run(s, K, reps, graph, [&](int k) {
                hipLaunchKernelGGL(k_shift<true>, dim3(54), dim3(256), 0, s, (const uint4*)(k & 1 ? b : a), (uint4*)(k & 1 ? a : b), 256, shift); }));
        }
        for (int shift : {0, 8, 256, 1}) {
            printf("%s  7 MB (1728 blocks), reads block b+%d, plain        : %6.2f us per launch\n", tag, shift, run(s, K, reps, graph, [&](int k) {
                hipLaunchKernelGGL(k_shift<false>, dim3(1728), dim3(256), 0, s, (const uint4*)(k & 1 ? b : a), (uint4*)(k & 1 ? a : b), 256, shift); }));
            printf("%s  7 MB (1728 blocks), reads block b+%d, sc1          : %6.2f us per launch\n", tag, shift, run(s, K, reps, graph, [&](int k) {
                hipLaunchKernelGGL(k_shift<true>, dim3(1728), dim3(256), 0, s, (const uint4*)(k & 1 ? b : a), (uint4*)(k & 1 ? a : b), 256, shift); }));
        }
        for (int shift : {0, 1})
            printf("%s  221 KB, two dependent reads, shift %d               : %6.2f us per launch\n", tag, shift, run(s, K, reps, graph, [&](int k) {
                hipLaunchKernelGGL(k_chain2, dim3(54), dim3(256), 0, s, (const uint4*)(k & 1 ? b : a), (uint4*)(k & 1 ? a : b), 256, shift); }));
    }
    return 0;
}
