// Shared device/host helpers for the gfx950 (CDNA4) 3D latent-diffusion kernels.
#pragma once
#include <stdlib.h>
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef uint16_t bf16_t;                                        // raw bf16 bits
typedef __attribute__((ext_vector_type(8))) short bf16x8;       // one MFMA 16x16x32 A/B fragment
typedef __attribute__((ext_vector_type(4))) float f32x4;        // one MFMA 16x16 accumulator
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

#define LDM_WAVE 64

__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }

// fp32 -> bf16, round-to-nearest-even.  A plain cast lowers to v_cvt_pk_bf16_f32 on gfx950 and keeps NaN a NaN.
__device__ __forceinline__ bf16_t f2bf(float f) {
    __bf16 b = (__bf16)f;
    return *reinterpret_cast<bf16_t*>(&b);
}
__device__ __forceinline__ uint32_t pack2bf(float lo, float hi) {
    return (uint32_t)f2bf(lo) | ((uint32_t)f2bf(hi) << 16);
}
__device__ __forceinline__ float silu_f(float x) { return x / (1.0f + __expf(-x)); }

// ---- environment knobs (host).  ldm_knob: switches of the SHIPPING library -- A/B of a kept feature, flipped by the parity tests and
// documented in INTEGRATION.md.  ldm_xknob: tuning / diagnostic knobs of measured-and-settled choices (DESIGN.md keeps the numbers): they
// read the environment only in an experiments build (make EXTRA=-DLDM_EXPERIMENTS) and are their defaults in the product library.
static inline long ldm_knob(const char* name, long dflt) { const char* e = getenv(name); return (e && *e) ? atol(e) : dflt; }
#ifdef LDM_EXPERIMENTS
static inline long ldm_xknob(const char* name, long dflt) { return ldm_knob(name, dflt); }
#else
static inline long ldm_xknob(const char*, long dflt) { return dflt; }
#endif

// 16-byte store; WT = true: write-through (sc1): the bytes leave the XCD's L2 while the kernel still runs instead of in the
// write-back at its end (short launches whose consumers run on every XCD anyway).  Knob: LDM_WT_STORES (GroupNorm / finalize).
// Measured in the conv / light-GEMM epilogues too (with the wait state): no gain there (475 vs 478 steps/s), so they store plainly.
template <bool WT>
__device__ __forceinline__ void store16(void* ptr, u32x4 v) {
    // s_nop 1 BEHIND the store: a VMEM store of more than 64 bits reads its data VGPRs over several cycles, and a VALU write to
    // them in the next wait state corrupts the lanes read last (hipcc's hazard recognizer inserts the wait state for its own stores but
    // cannot see into this asm: without it lanes 12-15 of every 16-lane group stored the NEXT value: found in the conv epilogue)
    if constexpr (WT) asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(ptr), "v"(v) : "memory");
    else *reinterpret_cast<u32x4*>(ptr) = v;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// ---- diagnostic build only (make EXTRA=-DLDM_KSTAMPS OUT=...): per-launch in-kernel timeline.  Thread 0 of block (0,0,0) of every
// instrumented kernel draws a ticket (= launch order) and records 100 MHz s_memrealtime stamps: [0] = kernel id, [1] = entry,
// [1 + k] = KSTAMP(k).  Read back with ldm_debug_kstamps (tools/kstamps.py).  Compiled out of the product library.
#ifdef LDM_KSTAMPS
#ifndef LDM_KSTAMP_BLOCK
#define LDM_KSTAMP_BLOCK 0        // which blockIdx.x records (a late block of a many-round grid shows the steady state, not the cold start)
#endif
__device__ unsigned long long g_kstamp[8192 * 8];
__device__ unsigned g_kstamp_seq;
struct KStamp {
    unsigned slot;
    __device__ __forceinline__ KStamp(int id) {
        slot = 0xffffffffu;
        if (blockIdx.x == LDM_KSTAMP_BLOCK && (blockIdx.y | blockIdx.z) == 0 && threadIdx.x == 0) {
            const unsigned long long t = __builtin_amdgcn_s_memrealtime();
            slot = atomicAdd(&g_kstamp_seq, 1u) & 8191u;
            g_kstamp[slot * 8] = (unsigned long long)id; g_kstamp[slot * 8 + 1] = t;
            for (int k = 2; k < 8; ++k) g_kstamp[slot * 8 + k] = 0;
        }
    }
    __device__ __forceinline__ void at(int k) { if (slot != 0xffffffffu) g_kstamp[slot * 8 + 1 + k] = __builtin_amdgcn_s_memrealtime(); }
};
#define KSTAMP_BEGIN(id) KStamp kst_(id)
#define KSTAMP(k) kst_.at(k)
#define KSTAMP_DRAIN(k) do { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); kst_.at(k); } while (0)
#define KSTAMP_PARAM , KStamp& kst_
#define KSTAMP_ARG , kst_
#else
#define KSTAMP_PARAM
#define KSTAMP_ARG
#define KSTAMP_BEGIN(id) do { } while (0)
#define KSTAMP(k) do { } while (0)
#define KSTAMP_DRAIN(k) do { } while (0)
#endif
