// Shared device helpers for the osu-dreamer denoiser kernels (gfx950 / CDNA4).
//
// Internal activation layout is FRAME-MAJOR: a (batch, channel, frame) tensor of the
// reference is held as rows = frames (m = b*L + l), columns = channels, channel
// contiguous.  Per-frame channel reductions (RMS norm) are then contiguous, every
// 1x1 Conv1d is one NT GEMM over M = B*L rows, and attention heads are 64 contiguous
// features per frame.  Only the boundary tensors (xt, audio, v) stay channel-major.
#pragma once
#if defined(OD_EMU)
#include "emu_hip.h"
#else
#include <hip/hip_runtime.h>
#define OD_LAUNCH(kern, grid, block, smem, stream, ...) \
    hipLaunchKernelGGL(kern, (grid), (block), (smem), (stream), __VA_ARGS__)
// kernels whose dynamic LDS may exceed the 64 KiB default (gfx950 has 160 KiB per CU)
#define OD_LAUNCH_DYN(kern, grid, block, smem, stream, ...)                                                    \
    do {                                                                                                       \
        /* once per process (one process drives one GPU): a refused attribute is an error code, not a silent 64 KiB limit */ \
        static const hipError_t attr_rc__ =                                                                    \
            hipFuncSetAttribute((const void*)(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);  \
        if (attr_rc__ != hipSuccess) return -(int)attr_rc__ - 1000;                                            \
        hipLaunchKernelGGL(kern, (grid), (block), (smem), (stream), __VA_ARGS__);                              \
    } while (0)
#define OD_DYN_SMEM(name) extern __shared__ __attribute__((aligned(16))) unsigned char name[]
#endif
#include <stdint.h>

typedef unsigned short bf16_t;  // raw bfloat16 bits
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

#define OD_WAVE 64

__device__ __forceinline__ float od_bf2f(bf16_t h) {
    union { uint32_t u; float f; } c; c.u = ((uint32_t)h) << 16; return c.f;
}
#if defined(OD_EMU)
// round-to-nearest-even, NaN preserved (software; the emulator has no v_cvt_pk_bf16_f32)
__device__ __forceinline__ bf16_t od_f2bf(float f) {
    union { uint32_t u; float f; } c; c.f = f;
    uint32_t u = c.u;
    if ((u & 0x7fffffffu) > 0x7f800000u) return (bf16_t)((u >> 16) | 0x40);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (bf16_t)(u >> 16);
}
__device__ __forceinline__ uint32_t od_pack_bf2(float lo, float hi) { return (uint32_t)od_f2bf(lo) | ((uint32_t)od_f2bf(hi) << 16); }
__device__ __forceinline__ float od_exp2(float x) { return exp2f(x); }
#else
// gfx950: float -> bf16 is one v_cvt_pk_bf16_f32 (RNE); exp2 is the raw v_exp_f32
typedef __bf16 od_bf2_t __attribute__((ext_vector_type(2)));
typedef float od_f2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ bf16_t od_f2bf(float f) { return __builtin_bit_cast(unsigned short, (__bf16)f); }
__device__ __forceinline__ uint32_t od_pack_bf2(float lo, float hi) {
    od_f2_t v; v[0] = lo; v[1] = hi;
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, od_bf2_t));
}
__device__ __forceinline__ float od_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
#endif

// IEEE half (OD_F16): the operand type of the attention MFMAs when the host asks for "attention in fp16" (BASELINE configs[4]; Lightning's
// precision: 16-mixed of the reference's trainer, model.yml:12).  A distinct type (bf16_t is a bare unsigned short): q, k, v, the staged dO
// and the in-kernel P / dS are half, everything around the attention core stays bf16.  Round-to-nearest-even both ways.
struct f16_t { unsigned short bits; };
#if defined(OD_EMU)
__device__ __forceinline__ float od_h2f(f16_t h) { return emu::f16_to_f32(h.bits); }
__device__ __forceinline__ f16_t od_f2h(float f) { return f16_t{emu::f32_to_f16(f)}; }
#else
__device__ __forceinline__ float od_h2f(f16_t h) { return (float)__builtin_bit_cast(_Float16, h.bits); }
__device__ __forceinline__ f16_t od_f2h(float f) { return f16_t{__builtin_bit_cast(unsigned short, (_Float16)f)}; }
#endif
#if defined(OD_EMU)
__device__ __forceinline__ uint32_t od_pack_h2(float lo, float hi) { return (uint32_t)od_f2h(lo).bits | ((uint32_t)od_f2h(hi).bits << 16); }
#else
// one packed conversion (round to nearest even) where the target has it
typedef _Float16 od_h2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t od_pack_h2(float lo, float hi) {
    float __attribute__((ext_vector_type(2))) v; v[0] = lo; v[1] = hi;
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, od_h2_t));
}
#endif
// two floats -> one dword of the 16-bit operand type T
template <class T> __device__ __forceinline__ uint32_t od_pack2(float lo, float hi);
template <> __device__ __forceinline__ uint32_t od_pack2<bf16_t>(float lo, float hi) { return od_pack_bf2(lo, hi); }
template <> __device__ __forceinline__ uint32_t od_pack2<f16_t>(float lo, float hi) { return od_pack_h2(lo, hi); }

// compute-type tag: fp32 storage whose MFMA products run as 3 bf16 MFMAs (OD_F32X3, inference only)
struct f32x3_t { float x; };
// same, for GEMMs whose WEIGHT operand was split into (hi, lo) bf16 halves once at pack time (od_pack_weight with OD_F32X3W): the
// activations are still split per fragment load, the weight fragments come out of LDS ready-made.  Pointers convert to f32x3_t*.
struct f32x3w_t : f32x3_t {};

template <class T> struct od_t;
template <> struct od_t<f32x3_t> {
    static __device__ __forceinline__ float ld(const f32x3_t* p) { return p->x; }
    static __device__ __forceinline__ void st(f32x3_t* p, float v) { p->x = v; }
};
template <> struct od_t<f32x3w_t> : od_t<f32x3_t> {};
template <> struct od_t<float> {
    static __device__ __forceinline__ float ld(const float* p) { return *p; }
    static __device__ __forceinline__ void st(float* p, float v) { *p = v; }
};
template <> struct od_t<bf16_t> {
    static __device__ __forceinline__ float ld(const bf16_t* p) { return od_bf2f(*p); }
    static __device__ __forceinline__ void st(bf16_t* p, float v) { *p = od_f2bf(v); }
};
template <> struct od_t<f16_t> {
    static __device__ __forceinline__ float ld(const f16_t* p) { return od_h2f(*p); }
    static __device__ __forceinline__ void st(f16_t* p, float v) { *p = od_f2h(v); }
};

// x as it reads back after a store to T
template <class T> __device__ __forceinline__ float od_round_to(float x);
template <> __device__ __forceinline__ float od_round_to<float>(float x) { return x; }
template <> __device__ __forceinline__ float od_round_to<bf16_t>(float x) { return od_bf2f(od_f2bf(x)); }
template <> __device__ __forceinline__ float od_round_to<f16_t>(float x) { return od_h2f(od_f2h(x)); }
template <> __device__ __forceinline__ float od_round_to<f32x3_t>(float x) { return x; }
template <> __device__ __forceinline__ float od_round_to<f32x3w_t>(float x) { return x; }

// 8 consecutive elements <-> 8 floats (p must be 16-byte aligned for bf16, 32 for f32)
__device__ __forceinline__ void od_ld8(const float* p, float (&v)[8]) {
    f32x4 a = *(const f32x4*)p, b = *(const f32x4*)(p + 4);
    v[0] = a[0]; v[1] = a[1]; v[2] = a[2]; v[3] = a[3];
    v[4] = b[0]; v[5] = b[1]; v[6] = b[2]; v[7] = b[3];
}
__device__ __forceinline__ void od_ld8(const bf16_t* p, float (&v)[8]) {
    u32x4 r = *(const u32x4*)p;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        union { uint32_t u; float f; } lo, hi;
        lo.u = r[i] << 16; hi.u = r[i] & 0xffff0000u;
        v[2 * i] = lo.f; v[2 * i + 1] = hi.f;
    }
}
__device__ __forceinline__ void od_st8(float* p, const float (&v)[8]) {
    f32x4 a, b;
    a[0] = v[0]; a[1] = v[1]; a[2] = v[2]; a[3] = v[3];
    b[0] = v[4]; b[1] = v[5]; b[2] = v[6]; b[3] = v[7];
    *(f32x4*)p = a; *(f32x4*)(p + 4) = b;
}
__device__ __forceinline__ void od_ld8(const f32x3_t* p, float (&v)[8]) { od_ld8((const float*)p, v); }
__device__ __forceinline__ void od_st8(f32x3_t* p, const float (&v)[8]) { od_st8((float*)p, v); }
__device__ __forceinline__ void od_st8(bf16_t* p, const float (&v)[8]) {
    u32x4 r;
#pragma unroll
    for (int i = 0; i < 4; i++) r[i] = od_pack_bf2(v[2 * i], v[2 * i + 1]);
    *(u32x4*)p = r;
}

__device__ __forceinline__ void od_ld8(const f16_t* p, float (&v)[8]) {
    u32x4 r = *(const u32x4*)p;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        v[2 * i] = od_h2f(f16_t{(unsigned short)(r[i] & 0xffffu)});
        v[2 * i + 1] = od_h2f(f16_t{(unsigned short)(r[i] >> 16)});
    }
}
__device__ __forceinline__ void od_st8(f16_t* p, const float (&v)[8]) {
    u32x4 r;
#pragma unroll
    for (int i = 0; i < 4; i++) r[i] = od_pack_h2(v[2 * i], v[2 * i + 1]);
    *(u32x4*)p = r;
}

// streaming (non-temporal) form: an output that is not read again soon should not evict re-read operands from the L2
template <class T> __device__ __forceinline__ void od_st8_nt(T* p, const float (&v)[8]) { od_st8(p, v); }
#if !defined(OD_EMU)
template <> __device__ __forceinline__ void od_st8_nt<bf16_t>(bf16_t* p, const float (&v)[8]) {
    u32x4 r;
#pragma unroll
    for (int i = 0; i < 4; i++) r[i] = od_pack_bf2(v[2 * i], v[2 * i + 1]);
    // inline asm, not __builtin_nontemporal_store: when a runtime flag selects between this and the plain store, hipcc merges the two
    // arms into ONE plain store and the hint is lost (seen in the ISA of gemm_nt_big_kernel: no `nt` on any store)
    // The s_nop is the wait state the ISA asks for between a store of more than 64 bits and a VALU write of its data registers (the store
    // reads them a cycle after issue).  hipcc's hazard pass inserts it for stores it can see; an asm statement is opaque to it, and two of
    // these stores back to back — the second one's packing reusing the first one's registers — stored garbage (profiles/r05d: the
    // "qk stored non-temporally" experiment wrote NaNs).
    asm volatile("global_store_dwordx4 %0, %1, off nt\n\ts_nop 0" ::"v"(p), "v"(r) : "memory");
}
#endif

// 16 packed bytes, streaming store (see od_st8_nt)
#if defined(OD_EMU)
__device__ __forceinline__ void od_st16_nt(void* p, u32x4 r) { *(u32x4*)p = r; }
#else
// OD_ST16_POLICY (A/B, round 6: profiles/r06g_nt_store_policy.txt): cache policy of the GEMMs' streaming output stores — 0 = nt (the default), 1 = sc1,
// 2 = sc0 sc1, 3 = sc1 nt, 4 = sc0 sc1 nt
#ifndef OD_ST16_POLICY
#define OD_ST16_POLICY 0
#endif
__device__ __forceinline__ void od_st16_nt(void* p, u32x4 r) {
#if OD_ST16_POLICY == 1
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 0" ::"v"(p), "v"(r) : "memory");
#elif OD_ST16_POLICY == 2
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 0" ::"v"(p), "v"(r) : "memory");
#elif OD_ST16_POLICY == 3
    asm volatile("global_store_dwordx4 %0, %1, off sc1 nt\n\ts_nop 0" ::"v"(p), "v"(r) : "memory");
#elif OD_ST16_POLICY == 4
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1 nt\n\ts_nop 0" ::"v"(p), "v"(r) : "memory");
#else
    asm volatile("global_store_dwordx4 %0, %1, off nt\n\ts_nop 0" ::"v"(p), "v"(r) : "memory");
#endif
}
#endif

__device__ __forceinline__ float od_wave_sum(float v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
    return v;
}
__device__ __forceinline__ float od_silu(float x) { return x / (1.0f + __expf(-x)); }
__device__ __forceinline__ float od_sigmoid(float x) { return 1.0f / (1.0f + __expf(-x)); }
// d/dx [x*sigmoid(x)]
__device__ __forceinline__ float od_silu_grad(float x) {
    float s = od_sigmoid(x);
    return s * (1.0f + x * (1.0f - s));
}

// Element-wise product WITHOUT packed-f32 math: beside MFMAs a v_pk_mul_f32 costs the matrix pipe ~16 cycles (tools/ubench/mix_power.hip:
// 64 MFMAs 446 ns; + 16 v_pk_mul_f32 551 ns; + 32 v_fma_f32 / v_exp_f32 / v_cvt_pk_bf16_f32: 443 - 451 ns), and hipcc turns every
// f32x4 * f32x4 (and f32x4 * scalar) into two of them.
#if defined(OD_EMU)
__device__ __forceinline__ f32x4 od_mul4(f32x4 a, f32x4 b) { return a * b; }
__device__ __forceinline__ f32x4 od_mul4s(f32x4 a, float b) { return a * b; }
#else
// Each product is laundered through an empty asm (no instruction; it only hides from the vector combiner that the four products are lanes
// of one vector).  The multiply itself stays a compiler-visible v_mul_f32: the wait states between an MFMA and a VALU read of its result are
// software-managed, and a multiply written AS asm read the MFMA result early (dk, dq wrong).
__device__ __forceinline__ float od_mul1(float a, float b) {
    float d = a * b;
    asm("" : "+v"(d));
    return d;
}
__device__ __forceinline__ f32x4 od_mul4(f32x4 a, f32x4 b) {
    f32x4 d;
#pragma unroll
    for (int r = 0; r < 4; r++) d[r] = od_mul1(a[r], b[r]);
    return d;
}
__device__ __forceinline__ f32x4 od_mul4s(f32x4 a, float b) {
    f32x4 d;
#pragma unroll
    for (int r = 0; r < 4; r++) d[r] = od_mul1(a[r], b);
    return d;
}
#endif

// ---------------------------------------------------------------------------------
// MFMA fragment abstraction, one shape for both compute types: a 16x16 output tile
// and a 32-deep K slab.  Each lane holds 8 K-consecutive elements of its row:
//   A frag: row (lane&15), k = 8*(lane>>4) + j      B frag: col (lane&15), same k
// bf16: one v_mfma_f32_16x16x32_bf16.  f32: eight v_mfma_f32_16x16x4_f32, the j-th
// taking element j of every lane group — a K permutation applied identically to A
// and B, hence the same dot product (exact f32 FMA chain).
// Accumulator (both): col = lane&15, row = 4*(lane>>4) + r.
// ---------------------------------------------------------------------------------
template <class T> struct od_frag;
template <> struct od_frag<bf16_t> { s16x8 v; };
template <> struct od_frag<f16_t> { s16x8 v; };
template <> struct od_frag<float> { float v[8]; };
// f32x3: fp32 in memory, three bf16 MFMAs per product.  x = hi + lo with hi = bf16(x), lo = bf16(x - hi)
// (|x - hi - lo| <= 2^-17 |x|);  a.b ~ ahi.bhi + ahi.blo + alo.bhi, accumulated in fp32.
template <> struct od_frag<f32x3_t> { s16x8 hi, lo; };
template <> struct od_frag<f32x3w_t> : od_frag<f32x3_t> {};
__device__ __forceinline__ void od_frag_pack(od_frag<float>& f, const float (&x)[8]) {
#pragma unroll
    for (int j = 0; j < 8; j++) f.v[j] = x[j];
}
__device__ __forceinline__ void od_split2(float a, float b, uint32_t& hi, uint32_t& lo) {
    hi = od_pack_bf2(a, b);
    union { uint32_t u; float f; } x, y;
    x.u = hi << 16; y.u = hi & 0xffff0000u;
    lo = od_pack_bf2(a - x.f, b - y.f);
}
__device__ __forceinline__ void od_frag_pack(od_frag<f32x3_t>& f, const float (&x)[8]) {
    u32x4 h, l;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        uint32_t a, b;
        od_split2(x[2 * i], x[2 * i + 1], a, b);
        h[i] = a; l[i] = b;
    }
    f.hi = __builtin_bit_cast(s16x8, h); f.lo = __builtin_bit_cast(s16x8, l);
}

__device__ __forceinline__ void od_frag_load(od_frag<bf16_t>& f, const bf16_t* p) { f.v = *(const s16x8*)p; }
__device__ __forceinline__ void od_frag_load(od_frag<f16_t>& f, const f16_t* p) { f.v = *(const s16x8*)p; }
__device__ __forceinline__ void od_frag_load(od_frag<float>& f, const float* p) { od_ld8(p, f.v); }
__device__ __forceinline__ void od_frag_load(od_frag<f32x3_t>& f, const f32x3_t* p) { float x[8]; od_ld8(p, x); od_frag_pack(f, x); }
__device__ __forceinline__ void od_frag_zero(od_frag<bf16_t>& f) { f.v = (s16x8)(0); }
__device__ __forceinline__ void od_frag_zero(od_frag<float>& f) {
#pragma unroll
    for (int j = 0; j < 8; j++) f.v[j] = 0.f;
}
__device__ __forceinline__ void od_frag_zero(od_frag<f32x3_t>& f) { f.hi = (s16x8)(0); f.lo = (s16x8)(0); }
// elements 4*half .. 4*half+3 of a fragment (half = 0 or 1)
__device__ __forceinline__ void od_frag_set4(od_frag<bf16_t>& f, int half, float a, float b, float c, float d) {
    u32x4 w = __builtin_bit_cast(u32x4, f.v);
    w[2 * half] = od_pack_bf2(a, b);
    w[2 * half + 1] = od_pack_bf2(c, d);
    f.v = __builtin_bit_cast(s16x8, w);
}
__device__ __forceinline__ void od_frag_set4(od_frag<f16_t>& f, int half, float a, float b, float c, float d) {
    u32x4 w = __builtin_bit_cast(u32x4, f.v);
    w[2 * half] = od_pack_h2(a, b);
    w[2 * half + 1] = od_pack_h2(c, d);
    f.v = __builtin_bit_cast(s16x8, w);
}
__device__ __forceinline__ void od_frag_set4(od_frag<float>& f, int half, float a, float b, float c, float d) {
    f.v[4 * half] = a; f.v[4 * half + 1] = b; f.v[4 * half + 2] = c; f.v[4 * half + 3] = d;
}
__device__ __forceinline__ void od_frag_set4(od_frag<f32x3_t>& f, int half, float a, float b, float c, float d) {
    u32x4 h = __builtin_bit_cast(u32x4, f.hi), l = __builtin_bit_cast(u32x4, f.lo);
    uint32_t h0, l0, h1, l1;
    od_split2(a, b, h0, l0);
    od_split2(c, d, h1, l1);
    h[2 * half] = h0; l[2 * half] = l0; h[2 * half + 1] = h1; l[2 * half + 1] = l1;
    f.hi = __builtin_bit_cast(s16x8, h); f.lo = __builtin_bit_cast(s16x8, l);
}

// LDS transpose read (gfx950 ds_read_b64_tr_b16): within each 16-lane group, lane c supplies the
// address of a 4-element chunk; lane i receives element (i&3) of chunks (i>>2) + 4j, j = 0..3
// (semantics measured on hardware with tools/probes/tr_probe.hip).  With lane c pointing at
// tile[r0 + (c>>2)][c0 + 4*(c&3)], lane i gets tile[r0 + j][c0 + i], j = 0..3: a column of 4 rows.
#if defined(OD_EMU)
__device__ __forceinline__ s16x4 od_lds_tr_read(const bf16_t* p) { return emu::ds_read_tr16_b64(p); }
#else
__device__ __forceinline__ s16x4 od_lds_tr_read(const bf16_t* p) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p);
}
#endif
// the same from a raw LDS byte address (od_lds_addr of the array + offset): `addr + constant` folds into the instruction's offset field, so one
// lane-dependent register serves every row / stage variant of a fragment's chunk (declared after od_lds_addr, below)
__device__ __forceinline__ s16x4 od_lds_tr_read_at(unsigned addr);

__device__ __forceinline__ f32x4 od_mma(const od_frag<bf16_t>& a, const od_frag<bf16_t>& b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.v, b.v, c, 0, 0, 0);
}
#if defined(OD_EMU)
__device__ __forceinline__ f32x4 od_mma(const od_frag<f16_t>& a, const od_frag<f16_t>& b, f32x4 c) { return emu::mfma_16x16x32_f16(a.v, b.v, c); }
#else
typedef _Float16 od_h8_t __attribute__((ext_vector_type(8)));
__device__ __forceinline__ f32x4 od_mma(const od_frag<f16_t>& a, const od_frag<f16_t>& b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(od_h8_t, a.v), __builtin_bit_cast(od_h8_t, b.v), c, 0, 0, 0);
}
#endif
__device__ __forceinline__ f32x4 od_mma(const od_frag<float>& a, const od_frag<float>& b, f32x4 c) {
#pragma unroll
    for (int j = 0; j < 8; j++) c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.v[j], b.v[j], c, 0, 0, 0);
    return c;
}
__device__ __forceinline__ f32x4 od_mma(const od_frag<f32x3_t>& a, const od_frag<f32x3_t>& b, f32x4 c) {
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.lo, b.hi, c, 0, 0, 0);  // small terms first
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.hi, b.lo, c, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.hi, b.hi, c, 0, 0, 0);
}

// 16 zero bytes in global memory: the source of LDS-DMA lanes that fall outside a matrix
static __device__ __attribute__((aligned(16))) unsigned char od_zero16[16];

// LDS-DMA: 16 bytes per lane straight from global memory into LDS, no VGPR round trip and no
// ds_write.  The hardware writes lane i's 16 bytes at (wave-uniform LDS base) + 16*i, so `lds` must be
// lane 0's destination + 16*lane; the SOURCE address is free per lane (swizzles go there).
#if defined(OD_EMU)
__device__ __forceinline__ void od_glds16(const void* g, void* lds) { emu::global_load_lds16(g, lds); }
#else
__device__ __forceinline__ void od_glds16(const void* g, void* lds) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)lds, 16, 0, 0);
}
#endif
// The same from inline asm (see the note at od_buffer_lds16_at: the builtin form is waited for, with vmcnt(0), in front of the NEXT LDS read of
// any address — nothing can be fetched more than one tile ahead).  `lds_base`: the wave's destination base (lane i lands at lds_base + 16 i),
// wave-uniform.  The caller counts: OD_WAIT_VMCNT before the barrier that publishes a tile.
#if defined(OD_EMU)
__device__ __forceinline__ void od_glds16_async(const void* g, void* lds_base) { emu::global_load_lds16(g, (unsigned char*)lds_base + 16 * emu::lane_id()); }
#else
__device__ __forceinline__ void od_glds16_async(const void* g, void* lds_base) {
    const unsigned a = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)lds_base);
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(g), "s"(a) : "memory", "m0");
}
#endif

// Buffer-addressed LDS-DMA (buffer_load_dwordx4 ... lds): descriptor (base, byte size) + per-lane byte offset (VGPR) +
// wave-uniform byte offset (SGPR).  The per-tile address arithmetic is then ONE scalar add instead of 64-bit vector math per
// piece, and bytes past the end of the buffer read as zero (no row clamping for ragged tiles).  `lds` is the wave's
// destination base (lane i lands at lds + 16 i); it must be wave-uniform.
#if defined(OD_EMU)
struct od_srd_t { const unsigned char* base; unsigned bytes; };
__device__ __forceinline__ od_srd_t od_make_srd(const void* base, unsigned bytes) { return od_srd_t{(const unsigned char*)base, bytes}; }
__device__ __forceinline__ void od_srd_set_bytes(od_srd_t& r, unsigned bytes) { r.bytes = bytes; }
__device__ __forceinline__ void od_buffer_lds16(od_srd_t r, unsigned voff, unsigned soff, void* lds) {
    // the hardware range-checks per DWORD: the in-range dwords of a chunk that straddles the end of the buffer are loaded
    alignas(16) unsigned char tmp[16] = {0};
    const unsigned off = voff + soff;
    if (off < r.bytes) { const unsigned n = (r.bytes - off) & ~3u; memcpy(tmp, r.base + off, n < 16 ? n : 16); }
    emu::global_load_lds16(tmp, (unsigned char*)lds + 16 * emu::lane_id());
}
__device__ __forceinline__ int od_uniform(int x) { return x; }
// the emulator's "LDS address" is an offset from the dynamic shared array
__device__ __forceinline__ unsigned od_lds_addr(const void* p) { return (unsigned)((const unsigned char*)p - emu::dyn_smem()); }
__device__ __forceinline__ void od_buffer_lds16_at(od_srd_t r, unsigned voff, unsigned soff, unsigned lds_addr) {
    od_buffer_lds16(r, voff, soff, emu::dyn_smem() + (long)(int)lds_addr);     // signed: a static __shared__ array may lie below the dynamic one
}
__device__ __forceinline__ void od_buffer_lds16_at_nt(od_srd_t r, unsigned voff, unsigned soff, unsigned lds_addr) { od_buffer_lds16_at(r, voff, soff, lds_addr); }
// M0 is a per-wave register: the emulator's fibers of one workgroup share an OS thread, so the stand-in is indexed by the wave
__device__ __forceinline__ unsigned& od_emu_m0() { static thread_local unsigned m0[32] = {0}; return m0[emu::wave_id() & 31]; }
__device__ __forceinline__ void od_dma_set_dst(unsigned lds_addr) { od_emu_m0() = lds_addr; }
__device__ __forceinline__ void od_buffer_lds16_m0(od_srd_t r, unsigned voff, unsigned soff) { od_buffer_lds16_at(r, voff, soff, od_emu_m0()); }
// one dword per lane: lane i's 4 bytes land at lds_addr + 4*i (zero past the end of the buffer)
__device__ __forceinline__ void od_buffer_lds4_at(od_srd_t r, unsigned voff, unsigned soff, unsigned lds_addr) {
    unsigned v = 0;
    const unsigned off = voff + soff;
    if (off + 4 <= r.bytes) memcpy(&v, r.base + off, 4);
    memcpy(emu::dyn_smem() + (long)(int)lds_addr + 4 * emu::lane_id(), &v, 4);
}
#else
// The DMA is issued from inline asm, on purpose: hipcc treats a builtin LDS-DMA as a pending LDS write and puts
// `s_waitcnt vmcnt(0)` in front of the NEXT ds_read of any address (seen in every attention loop: the tile prefetched at the top
// of an iteration was waited for in the middle of it).  An asm statement is opaque to that pass; the kernels place their own
// counted waits (OD_WAIT_VMCNT) before the barrier that publishes a tile.  m0 = LDS byte address of lane 0's destination.
typedef u32x4 od_srd_t;
__device__ __forceinline__ od_srd_t od_make_srd(const void* base, unsigned bytes) {
    const unsigned long long a = (unsigned long long)base;
    od_srd_t r;
    r[0] = __builtin_amdgcn_readfirstlane((unsigned)a);
    r[1] = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
    r[2] = __builtin_amdgcn_readfirstlane(bytes);
    r[3] = 0x00020000u;
    return r;
}
__device__ __forceinline__ void od_srd_set_bytes(od_srd_t& r, unsigned bytes) { r[2] = __builtin_amdgcn_readfirstlane(bytes); }
__device__ __forceinline__ void od_buffer_lds16(od_srd_t r, unsigned voff, unsigned soff, void* lds) {
    const unsigned m0v = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)lds);
    asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %2 offen lds" ::"v"(voff), "s"(r), "s"(soff), "s"(m0v) : "memory", "m0");
}
// same, the destination given as an LDS byte address (od_lds_addr of the array base + offset): no generic-pointer
// conversion (null check, aperture compare) per call
__device__ __forceinline__ unsigned od_lds_addr(const void* p) { return (unsigned)(size_t)(__attribute__((address_space(3))) const unsigned char*)p; }
__device__ __forceinline__ void od_buffer_lds16_at(od_srd_t r, unsigned voff, unsigned soff, unsigned lds_addr) {
    asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %2 offen lds" ::"v"(voff), "s"(r), "s"(soff), "s"(lds_addr) : "memory", "m0");
}
// the same in two halves, for loops that put an MFMA between them (the M0 write needs one wait state before the DMA reads it, and a wave that
// is alone on its SIMD has no issue slot to spare for an s_nop)
__device__ __forceinline__ void od_dma_set_dst(unsigned lds_addr) { asm volatile("s_mov_b32 m0, %0" ::"s"(lds_addr) : "memory", "m0"); }
__device__ __forceinline__ void od_buffer_lds16_m0(od_srd_t r, unsigned voff, unsigned soff) {
    asm volatile("buffer_load_dwordx4 %0, %1, %2 offen lds" ::"v"(voff), "s"(r), "s"(soff) : "memory", "m0");
}
// one dword per lane (64 x 4 B = 256 B per wave): lane i's dword lands at lds_addr + 4*i, zero past the end of the buffer
__device__ __forceinline__ void od_buffer_lds4_at(od_srd_t r, unsigned voff, unsigned soff, unsigned lds_addr) {
    asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dword %0, %1, %2 offen lds" ::"v"(voff), "s"(r), "s"(soff), "s"(lds_addr) : "memory", "m0");
}
// the same with the non-temporal hint: a stream that is read once (or twice, close together) and should not push a re-read
// operand out of the XCD's L2
__device__ __forceinline__ void od_buffer_lds16_at_nt(od_srd_t r, unsigned voff, unsigned soff, unsigned lds_addr) {
    asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %2 offen nt lds" ::"v"(voff), "s"(r), "s"(soff), "s"(lds_addr) : "memory", "m0");
}
// a value the program knows to be wave-uniform, told to the compiler (keeps it in an SGPR)
__device__ __forceinline__ int od_uniform(int x) { return __builtin_amdgcn_readfirstlane(x); }
#endif

#if defined(OD_EMU)
__device__ __forceinline__ s16x4 od_lds_tr_read_at(unsigned addr) { return emu::ds_read_tr16_b64((const unsigned short*)(emu::dyn_smem() + (long)(int)addr)); }
#else
__device__ __forceinline__ s16x4 od_lds_tr_read_at(unsigned addr) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(size_t)addr);
}
#endif

// Row exchange inside the 16-lane rows of a wave (DPP row_shr:8 / row_shl:8): lanes 8..15 of each row take `src` of the lane 8 below them
// (resp. lanes 0..7 that of the lane 8 above) and the other half keeps `old`.  The GEMM epilogues use it to turn two half-line (64-byte) row
// segments per store instruction into whole 128-byte lines.
#if defined(OD_EMU)
__device__ __forceinline__ uint32_t od_dpp_up8(uint32_t old, uint32_t src) { const uint32_t t = emu::shfl(src, emu::lane_id() - 8 >= 0 ? emu::lane_id() - 8 : 0); return (emu::lane_id() & 8) ? t : old; }
__device__ __forceinline__ uint32_t od_dpp_down8(uint32_t old, uint32_t src) { const uint32_t t = emu::shfl(src, emu::lane_id() + 8 < 64 ? emu::lane_id() + 8 : 63); return (emu::lane_id() & 8) ? old : t; }
#else
__device__ __forceinline__ uint32_t od_dpp_up8(uint32_t old, uint32_t src) {       // lanes x >= 8: src[lane - 8]
    return (uint32_t)__builtin_amdgcn_update_dpp((int)old, (int)src, 0x118 /* row_shr:8 */, 0xF, 0xC, false);
}
__device__ __forceinline__ uint32_t od_dpp_down8(uint32_t old, uint32_t src) {     // lanes x < 8: src[lane + 8]
    return (uint32_t)__builtin_amdgcn_update_dpp((int)old, (int)src, 0x108 /* row_shl:8 */, 0xF, 0x3, false);
}
#endif

// Two 16-byte pieces per lane — columns c0 + 8 g .. + 7 (a) and c0 + 32 + 8 g .. + 7 (b) of row row16 + x, the layout the GEMM epilogues hold
// — stored as WHOLE 128-byte lines: the two halves of the 16-lane rows trade pieces (DPP), instruction one writes rows 0..7 of the 16,
// instruction two rows 8..15, each lane row covering columns c0 .. c0 + 63.  Every lane must call it (the exchange is wave-wide); `cols_ok` and
// the row bounds only guard the stores.  TT = bf16_t or f16_t; NT: streaming stores.
template <class TT, bool NT>
__device__ __forceinline__ void od_store_line_pair(TT* base, size_t ld, int row16, int x, int g, int M, bool cols_ok, const float (&a)[8], const float (&b)[8]) {
    u32x4 ra, rb, lo, hi;
#pragma unroll
    for (int i = 0; i < 4; i++) { ra[i] = od_pack2<TT>(a[2 * i], a[2 * i + 1]); rb[i] = od_pack2<TT>(b[2 * i], b[2 * i + 1]); }
#pragma unroll
    for (int i = 0; i < 4; i++) { lo[i] = od_dpp_up8(ra[i], rb[i]); hi[i] = od_dpp_down8(ra[i], rb[i]); }
    const int xr = x & 7, xh = x >> 3;
    const int row_lo = row16 + xr, row_hi = row_lo + 8;
    if (cols_ok && row_lo < M) {
        TT* dst = base + (size_t)row_lo * ld + 32 * xh + 8 * g;
        if (NT) od_st16_nt(dst, lo); else *(u32x4*)dst = lo;
    }
    if (cols_ok && row_hi < M) {
        TT* dst = base + (size_t)row_hi * ld + 32 * (1 - xh) + 8 * g;
        if (NT) od_st16_nt(dst, hi); else *(u32x4*)dst = hi;
    }
}

// 8 bf16 from a raw LDS byte address -> 8 floats
#if defined(OD_EMU)
__device__ __forceinline__ void od_lds_ld8_at(unsigned addr, float (&v)[8]) { od_ld8((const bf16_t*)(emu::dyn_smem() + (long)(int)addr), v); }
#else
__device__ __forceinline__ void od_lds_ld8_at(unsigned addr, float (&v)[8]) {
    const u32x4 r = *(const __attribute__((address_space(3))) u32x4*)(size_t)addr;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        union { uint32_t u; float f; } lo, hi;
        lo.u = r[i] << 16; hi.u = r[i] & 0xffff0000u;
        v[2 * i] = lo.f; v[2 * i + 1] = hi.f;
    }
}
#endif

// Hand-placed synchronisation for kernels that keep LDS-DMA in flight across workgroup barriers: __syncthreads() would drain
// the VMEM counter (an LDS-DMA is a pending LDS write), so those kernels use a bare s_barrier and counted vmcnt waits.
#if defined(OD_EMU)
#define OD_WAIT_VMCNT(n) ((void)0)                       // the emulator's DMA completes at issue
#define OD_WAIT_LGKMCNT(n) ((void)0)
#define OD_DRAIN_VMEM() ((void)0)
__device__ __forceinline__ void od_barrier_raw() { __syncthreads(); }
#else
#define OD_WAIT_VMCNT(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
// vmcnt(0) through the builtin (simm16: vmcnt = 0, expcnt = 7, lgkmcnt = 15): hipcc's wait insertion SEES it, so compiler-visible
// global loads issued before a hand-counted DMA loop are not waited for again (with vmcnt(0)) at their first use INSIDE that loop
#define OD_DRAIN_VMEM() __builtin_amdgcn_s_waitcnt(0x0F70)
// lgkmcnt(0) through the builtin (simm16: vmcnt = 63, expcnt = 7, lgkmcnt = 0), so that hipcc's own wait insertion SEES it and
// does not wait again for the same ds_reads at their first use
#define OD_WAIT_LGKMCNT(n) __builtin_amdgcn_s_waitcnt(0xC07F | ((n) << 8))
__device__ __forceinline__ void od_barrier_raw() { asm volatile("s_barrier" ::: "memory"); }
#endif

// ---------------------------------------------------------------------------------
// Deterministic accumulation (OD_DETERMINISTIC; od_det_* in det.hip).  Every sum the step forms with fp32 atomics — weight-gradient
// split-K partials, per-(batch, channel) modulation gradients, bias / norm-weight gradients, the loss and norm scalars — depends on the order the
// blocks arrive in, so two runs of one step differ in the last bits (5.9e-5 rel-L2 at the bench shape).  With a table of registered destination
// ranges the same kernels add 2^40-scaled INTEGERS into a 64-bit shadow of the destination instead: integer addition is associative, the
// result is the same whatever the order (and wrap-around in between is harmless: only the final sum has to fit, |sum| < 2^23), and
// od_det_flush folds the shadow into the fp32 destination before its first reader.  Values the shadow cannot hold (NaN, Inf, |v| >= 2^23)
// take the float atomic, so a non-finite gradient stays visible.  det == NULL: the plain float atomic.
struct OdDetRange { const float* base; long long count; long long* shadow; };
struct OdDetTable { int n; int pad; OdDetRange r[20]; };      // 488 bytes: arena + two live plans' six buffers each + spare
__device__ __forceinline__ long long od_fix(float v) { return llrint((double)v * 1099511627776.0); }       // 2^40; exact product, rounded to nearest (truncation biased every sum towards zero)
__device__ __forceinline__ float od_unfix(long long s) { return (float)((double)s * (1.0 / 1099511627776.0)); }
__device__ __forceinline__ void od_red_add(const OdDetTable* det, float* p, float v) {
    if (det && fabsf(v) < 8388608.f) {
        const int n = det->n;
        for (int i = 0; i < n; i++) {
            const long long off = p - det->r[i].base;
            if (off >= 0 && off < det->r[i].count) {
                atomicAdd((unsigned long long*)(det->r[i].shadow + off), (unsigned long long)od_fix(v));
                return;
            }
        }
    }
    atomicAdd(p, v);
}
// The same for an epilogue that adds MANY elements of one matrix (the weight-gradient GEMMs: 256 accumulator registers per lane): the range
// is looked up once, the per-element code is a uniform branch and one atomic (a table walk per element would not unroll, and an accumulator
// array that is indexed by a loop variable lives in scratch memory).
__device__ __forceinline__ long long* od_det_find(const OdDetTable* det, const float* base) {
    if (det) {
        const int n = det->n;
        for (int i = 0; i < n; i++) {
            const long long off = base - det->r[i].base;
            if (off >= 0 && off < det->r[i].count) return det->r[i].shadow + off;
        }
    }
    return nullptr;
}
__device__ __forceinline__ void od_red_add_at(long long* shadow, float* base, size_t off, float v) {
    if (shadow && fabsf(v) < 8388608.f) atomicAdd((unsigned long long*)(shadow + off), (unsigned long long)od_fix(v));
    else atomicAdd(base + off, v);
}
// Block-level sums in LDS: always fixed point (ds_add_u64), so the order the waves of a block arrive in never shows either.  A value the
// format cannot hold sets the block's `bad` word instead; od_lds_unfix then hands back NaN for every slot of that block.
__device__ __forceinline__ void od_lds_fix_add(long long* p, float v, int* bad) {
    if (fabsf(v) < 8388608.f) atomicAdd((unsigned long long*)p, (unsigned long long)od_fix(v));
    else *bad = 1;
}
__device__ __forceinline__ float od_lds_unfix(long long s, int bad) { return bad ? __builtin_nanf("") : od_unfix(s); }

#define OD_CHECK_LAUNCH()                                \
    do {                                                 \
        hipError_t e__ = hipGetLastError();              \
        if (e__ != hipSuccess) return -(int)e__ - 1000;  \
    } while (0)
