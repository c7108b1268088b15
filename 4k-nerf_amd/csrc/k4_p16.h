// Shared by the producers / consumers of PRE-SPLIT ("p16") activations: k4_sr_p16.hip (3x3 convolutions), k4_sr.hip (SFT layers).
// Format: see the header comment of k4_sr_p16.hip.
#pragma once
#include "k4_common.h"

typedef float p16_f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 p16_f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned p16_u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned p16_u32x2 __attribute__((ext_vector_type(2)));
typedef float p16_f32x4 __attribute__((ext_vector_type(4)));

#define P16_OOB 0x80000000u

// a * b + c with TWO roundings (the reference's separate ops, lib/sr_esrnet.py:158,182; see k4_sr.hip)
__device__ __forceinline__ float p16_mul_add(float a, float b, float c) { return __fadd_rn(__fmul_rn(a, b), c); }

__device__ __forceinline__ __amdgpu_buffer_rsrc_t p16_rsrc(const void* p, unsigned bytes) {
    const unsigned long long a = (unsigned long long)p;
    return __builtin_amdgcn_make_buffer_rsrc(
        reinterpret_cast<void*>(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(a >> 32)) << 32) |
                                (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)a)),
        0, __builtin_amdgcn_readfirstlane((int)bytes), 0x00020000);
}

// Four values v[0..3] (channels c .. c+3 of one pixel) -> fp16 hi pairs H[2] and lo pairs L[2] of v * sc (sc a power of two):
// hi = RNE(v sc) (adding -0.0 keeps the sign of a negative zero), lo = RNE(v sc - hi): 8 x v_fma_mix{lo,hi}_f16, each one rounding.
__device__ __forceinline__ void p16_split4(const float (&v)[4], float sc, unsigned (&H)[2], unsigned (&L)[2]) {
    const float nz = -0.f;
    asm("v_fma_mixlo_f16 %0, %1, %2, %3" : "=v"(H[0]) : "v"(v[0]), "v"(sc), "v"(nz));
    asm("v_fma_mixhi_f16 %0, %1, %2, %3" : "+v"(H[0]) : "v"(v[1]), "v"(sc), "v"(nz));
    asm("v_fma_mixlo_f16 %0, %1, %2, %3" : "=v"(H[1]) : "v"(v[2]), "v"(sc), "v"(nz));
    asm("v_fma_mixhi_f16 %0, %1, %2, %3" : "+v"(H[1]) : "v"(v[3]), "v"(sc), "v"(nz));
    asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel:[0,0,0] op_sel_hi:[0,0,1]" : "=v"(L[0]) : "v"(v[0]), "v"(sc), "v"(H[0]));
    asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(L[0]) : "v"(v[1]), "v"(sc), "v"(H[0]));
    asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel:[0,0,0] op_sel_hi:[0,0,1]" : "=v"(L[1]) : "v"(v[2]), "v"(sc), "v"(H[1]));
    asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(L[1]) : "v"(v[3]), "v"(sc), "v"(H[1]));
}

// The p16 unit this lane stores for its pixel: lanes l and l + 32 hold channels 8q + 0..3 and 8q + 4..7 of the same pixel; after two
// v_permlane32_swap the LOWER lane holds the hi unit [H(l), H(l+32)] and the UPPER lane the lo unit [L(l), L(l+32)] of channels 8q .. 8q+7.
__device__ __forceinline__ p16_u32x4 p16_unit(const unsigned (&H)[2], const unsigned (&L)[2]) {
    const p16_u32x2 s0 = __builtin_amdgcn_permlane32_swap(H[0], L[0], false, false);      // .x: lower lanes H own | upper lanes L of the lower partner
    const p16_u32x2 s1 = __builtin_amdgcn_permlane32_swap(H[1], L[1], false, false);      // .y: lower lanes H of the upper partner | upper lanes L own
    return p16_u32x4{s0.x, s1.x, s0.y, s1.y};
}
// byte offset, inside a pixel's 32-channel block, of the unit `p16_unit` returns for channel group q (0..3) on this half-wave
#define P16_UNIT_OFF(Q, HALF) ((unsigned)((((Q) >> 1) * 64) + ((2 * (HALF) + ((Q) & 1)) * 16)))

