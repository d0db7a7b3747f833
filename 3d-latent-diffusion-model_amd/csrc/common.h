// Shared device/host helpers for the gfx950 (CDNA4) 3D latent-diffusion kernels.
#pragma once
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
