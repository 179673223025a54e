// Fused backward of the denoiser's self-attention (common/attn.py:82, F.scaled_dot_product_attention) — bf16, head_dim 64.
//
// ONE kernel, the five algorithmic MFMA passes (S, dP, dV, dK, dQ) instead of the seven of the dK/dV + dQ kernel pair in attn.hip
// (which recomputes S and dP for dQ).  The obstacle to a 5-pass backward is dQ: a workgroup that owns a block of keys forms, for every
// query tile, only ITS keys' share of dQ, and the chip's atomics (340 G fp32 / s) cannot sum those shares.  Here they are summed by a
// CHAIN: the key blocks of one (batch, head) run on the compute units of one XCD at the same time, each adds its share to a running fp32
// tile that travels block -> block through that XCD's L2 (plain stores, L1-bypassing loads; no flags: every 16-byte piece carries the number
// of the write it belongs to), and the last key block writes the finished bf16 dQ.  No atomics on data, a fixed summation order
// (deterministic), no extra pass over memory.
//
// Workgroup = 8 waves on one CU, two ROLES (one wave of each per SIMD), each with its own job loop:
//   * waves 0-3 ("key waves") own 48 keys each (192 per workgroup) with K, V fragments and the dK, dV accumulators in registers, exactly as
//     flash_bwd_dkv_kernel does: per 64-query tile  S = Q K^T ; P = 2^S ; dP = dO V^T ; dS = P (dP - delta) ; dV^T += dO^T P ; dK^T += Q^T dS.
//     They touch no global memory inside the loop, and publish dS (bf16) to LDS as [key][query].  They are the pace of the workgroup.
//   * waves 4-7 ("query waves") stream the Q / dO tiles (LDS-DMA, issued piece by piece between their MFMAs), and turn the published dS of
//     the PREVIOUS tile into dQ: wave h owns 16 of the tile's 64 queries, dQ^T[64 features x 16] = K^T[64 x 192 keys] dS[192 x 16], the block's
//     K^T in registers (read once per job from an LDS image of its K rows), its quarter of dS by LDS transpose reads — the contraction over
//     the workgroup's 192 keys happens inside the MFMA chain, so no cross-wave reduction exists.  Then they run the chain step for a tile
//     whose running sum has arrived (an elastic ring of four shares: a late predecessor costs an iteration of lag, not a stall).
// The grid is persistent (one workgroup per CU); workgroups take (batch-head, key-block) jobs from a queue PER XCD (by the hardware's XCC id),
// in order, so a job's predecessor in the chain was always taken earlier: it is running or done, never waiting for a free CU.
#include "od_common.h"
#include "od_tiles.h"
#include "od_api_internal.h"
#include <type_traits>
#include <stddef.h>
#include <stdlib.h>

namespace {

constexpr float FB_LOG2E = 1.4426950408889634f;
constexpr float FB_LN2 = 0.6931471805599453f;

constexpr int FB_NK = 3;                          // 16-key tiles per key wave
constexpr int FB_KB = 4 * FB_NK * 16;             // keys per workgroup (192)
constexpr int FB_TILE = 64 * 128;                 // one 64-row x 64-feature bf16 tile
constexpr int FB_STAGE = 2 * FB_TILE + 512;       // Q, dO, (-lse', -delta)
constexpr int FB_DS = FB_KB * 128;                // dS of one tile: [192 keys][64 queries] bf16
#ifndef FB_NSLOTS
#define FB_NSLOTS 4
#endif
constexpr int FB_SLOTS = FB_NSLOTS;                          // (batch, head)s per XCD whose running tiles exist at a time (two are in flight; the buffer is reused)
#ifndef FB_PACK
#define FB_PACK 1      // 1: the running tile travels as 22-bit values (sign, exponent, 13 mantissa bits), 16 of them + the write number in THREE 16-byte
                       // pieces per lane instead of four: 0.75 x the chain's bytes and vector-memory instructions (round 6, see fb_pack): 21.01 -> 20.69 ms
                       // per call, 320.4 -> 317.9 ms per step on one box (profiles/r06e_chain_pack.txt); 0 = four fp32 pieces with 3 tag bits per dword
#endif
constexpr int FB_NP = FB_PACK ? 3 : 4;               // 16-byte pieces per lane and tile
constexpr int FB_RUN_TILE = 4 * FB_NP * 64 * 4;      // floats of one (batch-head, query tile) of the running buffer: 4 waves x FB_NP pieces x 64 lanes x 4
constexpr int FB_KST = 2 * FB_STAGE + 2 * FB_DS;     // the key block's K rows, staged once per job for the query waves (same image as a dS tile)
constexpr int FB_SMEM = FB_KST + FB_DS + 16;

#ifndef FB_ORDER
#define FB_ORDER 3     // how a query wave orders its vector-memory work inside an iteration (1, 2: measured alternatives, profiles/r04ac)
#endif
#ifndef FB_REQ
#define FB_REQ 2       // the slab of the dQ product after which the running tile is requested
#endif
#ifndef FB_SPLIT
#define FB_SPLIT 1     // how the four query waves split the dQ product of a tile (see there; 0, 2: measured alternatives, profiles/r04ae, r04al)
#endif
#ifndef FB_X
#define FB_X 0        // timing experiments only (wrong dq): bit 0 = no chain traffic, bit 1 = no dQ products, bit 2 = no dS publication, bit 8 = no exp2, bit 9 = no transposed Q / dO reads after the first, bit 10 = no row-major Q / dO reads after the first,
                      // bit 3 = every running-tile load / store goes to tile 0 of its slot (the traffic never leaves the L2), bit 5 = tags not checked (nobody waits for a predecessor), bit 4 = no running-tile stores, bit 6 = no running-tile loads (both with bit 5)
#endif

// every lambda of the kernel must be inlined: one that is not keeps its by-reference captures (the accumulator rings!) in scratch memory
#if defined(OD_EMU)
#define FB_INLINE
#else
#define FB_INLINE __attribute__((always_inline))
#endif
#ifndef FB_PROF
#define FB_PROF 0
#endif
#if FB_PROF && !defined(OD_EMU)
#define FB_CLK() __builtin_readcyclecounter()
#else
#define FB_CLK() 0ull
#endif

// control block at the head of the workspace (zero before the FIRST launch; every launch leaves it zero)
struct FbSync {
    int head[8];          // next job of each XCD's queue
    int finished;         // workgroups that have left the job loop
    int jobs_done;        // jobs processed (all XCDs)
    int err;              // sticky: 1 = a launch ended with unprocessed jobs (an XCD without workgroups), 2 = shape mismatch (refused),
                          // 3 = a chain wait ran out of time (the predecessor's write never came): dq of that launch is NaN
    int calls;            // completed launches on this workspace: its number is in every running-tile tag
    int shape[3];         // B, H, L of the launches this workspace has served (0 = none yet): slot numbering and layout depend on them
    int pad;
    unsigned long long prof[16];   // FB_PROF builds: cycle counts (see fb_prof_add)
};

// 16-byte slot swizzle of the dS image (row = key, 128-byte rows): both the key waves' ds_write_b128 (8 consecutive keys per service group)
// and the query waves' transpose reads (rows r..r+3 and r+8..r+11 per 32 lanes) are bank-conflict-free (enumerated offline)
// by value: __builtin_bit_cast applied directly to a vector ELEMENT reads element 0 whatever the index (clang 20)
__device__ __forceinline__ unsigned fb_f2u(float x) { return __builtin_bit_cast(unsigned, x); }
__device__ __forceinline__ float fb_u2f(unsigned x) { return __builtin_bit_cast(float, x); }

// a quiet NaN the compiler cannot see through (this file is built with -ffinite-math-only: a NaN it knew of would be undefined behaviour)
__device__ __forceinline__ float fb_poison() {
#if defined(OD_EMU)
    volatile unsigned bits = 0x7fc00000u;
    return fb_u2f(bits);
#else
    float x;
    asm volatile("v_mov_b32 %0, 0x7fc00000" : "=v"(x));
    return x;
#endif
}

__device__ __forceinline__ int fb_swz(int row) { return (row & 7) ^ (((row >> 3) & 1) << 2); }

#if defined(OD_EMU)
__device__ __forceinline__ int fb_xcc_id() { return (int)(blockIdx.x & 7); }
__device__ __forceinline__ int fb_flag_load(const int* p) { return *p; }
__device__ __forceinline__ void fb_flag_store(int* p, int v) { *p = v; }
struct fb_rsrc_t { const unsigned char* base; };
__device__ __forceinline__ fb_rsrc_t fb_make_rsrc(const float* base, unsigned) { return fb_rsrc_t{(const unsigned char*)base}; }
__device__ __forceinline__ f32x4 fb_ld_l2(fb_rsrc_t r, unsigned voff, unsigned soff) { return *(const f32x4*)(r.base + voff + soff); }
__device__ __forceinline__ void fb_st_run(float* p, f32x4 t) { *(f32x4*)p = t; }
__device__ __forceinline__ void fb_sleep() {}
// the emulator runs the workgroups one after the other: a predecessor's write is either there or will never come — every poll costs one "tick"
__device__ __forceinline__ unsigned long long fb_now(unsigned polls) { return polls; }
__device__ __forceinline__ int fb_atomic_inc(int* p) { return atomicAdd(p, 1); }
#define FB_WAIT_ALL() ((void)0)
#define FB_COMPILER_FENCE() ((void)0)
#define FB_WAIT_BUT(n) ((void)0)
#else
__device__ __forceinline__ int fb_xcc_id() {
    int v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 7;
}
// relaxed agent-scope accesses: sc1 = served by the XCD's L2, never by this CU's L1
__device__ __forceinline__ int fb_flag_load(const int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void fb_flag_store(int* p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// 16 bytes of the running sum, L1-bypassing and COMPILER-VISIBLE (the value is used an iteration later: the compiler must know the registers
// are pending): buffer_load_dwordx4 ... offen sc1 through a descriptor over this (batch, head)'s running tiles
// Cache policy of the running-tile loads: any L1-bypassing form is correct (the tags reject what is not the predecessor's write).  Round 6 A/B
// (profiles/r06b_chain_price_policy.txt, same box): sc1 (16; rounds 4-5) 21.74 ms, sc0 sc1 (17) +-0, nt (2) and sc1 nt (18) 21.18 ms = -2.6 %,
// -5.9 ms per training step — a running tile is read ONCE by its one consumer, and the streaming hint keeps it from displacing the Q / dO
// tiles the XCD's other 31 key blocks re-read from the L2.
#ifndef FB_LD_AUX
#define FB_LD_AUX 2       /* nt */
#endif
typedef __amdgpu_buffer_rsrc_t fb_rsrc_t;
__device__ __forceinline__ fb_rsrc_t fb_make_rsrc(const float* base, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ f32x4 fb_ld_l2(fb_rsrc_t r, unsigned voff, unsigned soff) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, FB_LD_AUX));
}
// a running-tile store.  FB_ST_POLICY: 0 = plain (the line stays in the XCD's L2: the successor's sc1 load finds it there), 1 = nt, 2 = sc1,
// 3 = sc0 sc1 (write-through forms: the line is dropped / bypasses) — round-6 A/B, profiles/r06a_chain_policy.txt
#ifndef FB_ST_POLICY
#define FB_ST_POLICY 0
#endif
__device__ __forceinline__ void fb_st_run(float* p, f32x4 t) {
#if FB_X & 16
    asm volatile("" :: "v"(t));
#elif FB_ST_POLICY == 0
    *(f32x4*)p = t;
#elif FB_ST_POLICY == 1
    asm volatile("global_store_dwordx4 %0, %1, off nt\n\ts_nop 0" :: "v"(p), "v"(t) : "memory");      // (s_nop: see od_st8_nt)
#elif FB_ST_POLICY == 2
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 0" :: "v"(p), "v"(t) : "memory");
#else
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 0" :: "v"(p), "v"(t) : "memory");
#endif
}
__device__ __forceinline__ void fb_sleep() { __builtin_amdgcn_s_sleep(4); }
// the constant 100 MHz counter (s_memrealtime): the budget of a chain wait is wall time, whatever the shader clock does
__device__ __forceinline__ unsigned long long fb_now(unsigned) { return __builtin_amdgcn_s_memrealtime(); }
__device__ __forceinline__ int fb_atomic_inc(int* p) { return __hip_atomic_fetch_add(p, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// everything this wave has issued to memory is complete (loads landed, stores acknowledged by the L2)
#define FB_WAIT_ALL() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#define FB_COMPILER_FENCE() asm volatile("" ::: "memory")
// ... except the n youngest operations (vector memory operations complete in issue order)
#define FB_WAIT_BUT(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
#endif

// nl = -lse * inv_scale (the score accumulators' start value), nd = -delta = -sum_d dO*O — one wave per frame row, lanes over (h, d) chunks of 8
__global__ __launch_bounds__(256) void fb_prep_kernel(const bf16_t* __restrict__ o, int ldo, const bf16_t* __restrict__ dout, int lddo,
                                                      const float* __restrict__ lse, float* __restrict__ nl, float* __restrict__ nd,
                                                      int B, int H, int L, float inv_scale) {
    const int lane = threadIdx.x & 63;
    const long m = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (m >= (long)B * L) return;
    const int b = (int)(m / L), l = (int)(m % L);
    const int nch = H * 8;
    for (int it = 0; it * 64 < nch; it++) {
        const int qd = it * 64 + lane;
        const bool act = qd < nch;
        float s = 0.f;
        if (act) {
            float a[8], d[8];
            od_ld8(o + m * ldo + qd * 8, a); od_ld8(dout + m * lddo + qd * 8, d);
#pragma unroll
            for (int e = 0; e < 8; e++) s += a[e] * d[e];
        }
        s += __shfl_xor(s, 1); s += __shfl_xor(s, 2); s += __shfl_xor(s, 4);
        if (act && (qd & 7) == 0) {
            const size_t i = ((size_t)b * H + (qd >> 3)) * L + l;
            nd[i] = -s;
            nl[i] = -lse[i] * inv_scale;
        }
    }
}

// ---- "attention in fp16": dout arrives as bf16 like every other activation gradient and is STAGED as half, scaled by a power of two
// largest |dout| of the tensor (bit pattern of a non-negative float: unsigned compare = float compare); *amax_bits is zero before the launch
__global__ __launch_bounds__(256) void fb_amax_kernel(const bf16_t* __restrict__ dout, int lddo, long M, int dh, unsigned* __restrict__ amax_bits) {
    const int nch = dh / 8;
    float m = 0.f;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < M * nch; i += (long)gridDim.x * 256) {
        float d[8];
        od_ld8(dout + (i / nch) * lddo + (i % nch) * 8, d);
#pragma unroll
        for (int e = 0; e < 8; e++) m = fmaxf(m, fabsf(d[e]));
    }
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) m = fmaxf(m, __shfl_xor(m, s));
    if ((threadIdx.x & 63) == 0 && m > 0.f) atomicMax(amax_bits, fb_f2u(m));
}
// the power of two that brings amax into (2, 4]: dP = dO V^T sums 64 products, so |dS| <= 64 * 4 * max|v| stays below half's 65504 up to
// max|v| = 255 (round 5 aimed at (32, 64], which left room for |v| < 16 only: ADVICE r5), and 2^-24 (half's smallest subnormal) is still
// 2^-26 of the largest element.  1 for an all-zero or non-finite tensor.
__device__ __forceinline__ float fb_dscale_of(unsigned amax_bits) {
    const unsigned e = (amax_bits >> 23) & 0xffu;
    if (amax_bits == 0u || e == 0u || e == 0xffu) return 1.0f;
    const int ex = (int)e - 127 + ((amax_bits & 0x7fffffu) ? 1 : 0);      // ceil(log2(amax))
    int se = 2 - ex; se = se < -100 ? -100 : se > 100 ? 100 : se;
    return fb_u2f((unsigned)(se + 127) << 23);
}
// fb_prep_kernel + the staged copy: nl = -lse', nd = -dscale * delta, do16 = half(dscale * dout)
__global__ __launch_bounds__(256) void fb_prep_f16_kernel(const bf16_t* __restrict__ o, int ldo, const bf16_t* __restrict__ dout, int lddo,
                                                          const float* __restrict__ lse, float* __restrict__ nl, float* __restrict__ nd,
                                                          f16_t* __restrict__ do16, int ldd16, const unsigned* __restrict__ amax_bits,
                                                          float* __restrict__ dscale_out, int B, int H, int L, float inv_scale) {
    const float ds = fb_dscale_of(amax_bits[0]);
    if (blockIdx.x == 0 && threadIdx.x == 0) dscale_out[0] = ds;
    const int lane = threadIdx.x & 63;
    const long m = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (m >= (long)B * L) return;
    const int b = (int)(m / L), l = (int)(m % L);
    const int nch = H * 8;
    for (int it = 0; it * 64 < nch; it++) {
        const int qd = it * 64 + lane;
        const bool act = qd < nch;
        float s = 0.f;
        if (act) {
            float a[8], d[8];
            od_ld8(o + m * ldo + qd * 8, a); od_ld8(dout + m * lddo + qd * 8, d);
#pragma unroll
            for (int e = 0; e < 8; e++) { s += a[e] * d[e]; d[e] *= ds; }
            od_st8(do16 + m * ldd16 + qd * 8, d);
        }
        s += __shfl_xor(s, 1); s += __shfl_xor(s, 2); s += __shfl_xor(s, 4);
        if (act && (qd & 7) == 0) {
            const size_t i = ((size_t)b * H + (qd >> 3)) * L + l;
            nd[i] = -s * ds;
            nl[i] = -lse[i] * inv_scale;
        }
    }
}

// TA = the MFMA operand type: bf16_t, or f16_t ("attention in fp16": q, k, v, dout point at IEEE-half data, P and dS are rounded to half, the
// five products are v_mfma_f32_16x16x32_f16; dq, dk, dv are written as bf16 either way — they are rounded accumulators).  In the half form dout
// is the copy fb_prep_kernel staged, multiplied by the power of two `dscale` that brings the tensor's largest magnitude into (2, 4] —
// gradients of a mean-reduced loss are ~1e-6 and would vanish in half; the outputs are divided by it again.
template <bool PRE, class TA>
__global__ __launch_bounds__(512, 2) void flash_bwd_fused_kernel(const bf16_t* __restrict__ q, int ldq, const bf16_t* __restrict__ k, int ldk,
                                                                 const bf16_t* __restrict__ v, int ldv, const bf16_t* __restrict__ dout, int lddo,
                                                                 const float* __restrict__ nl, const float* __restrict__ nd,
                                                                 bf16_t* __restrict__ dq, int lddq, bf16_t* __restrict__ dk, int lddk,
                                                                 bf16_t* __restrict__ dv, int lddv, float* __restrict__ run,
                                                                 FbSync* __restrict__ sync, int B, int H, int L, float scale,
                                                                 unsigned wait_ticks, const float* __restrict__ dscale_p) {
    constexpr int NK = FB_NK, KB = FB_KB;
    OD_DYN_SMEM(smem);
    int* const s_job = (int*)(smem + FB_KST + FB_DS);
    const int BH = B * H, nkb = (L + KB - 1) / KB, nqt = (L + 63) / 64;
    const int jobs_per_xcd = ((BH + 7) / 8) * nkb;
    const int lane = threadIdx.x & 63, wave = od_uniform(threadIdx.x >> 6), x = lane & 15, g = lane >> 4;
    const bool key_wave = wave < 4;
    const int xcc = fb_xcc_id();
    const int calls = sync->calls;                     // launches completed on this workspace (the last workgroup out advances it)
    // A workspace belongs to ONE (B, H, L): the write numbers in its running tiles continue from launch to launch, and a launch of another shape
    // would wait for numbers that never come.  Every workgroup sees the same words here (they are written by the last workgroup to LEAVE a launch).
    {
        const int s0 = sync->shape[0], s1 = sync->shape[1], s2 = sync->shape[2];
        if ((s0 | s1 | s2) != 0 && (s0 != B || s1 != H || s2 != L)) {
            if (blockIdx.x == 0 && threadIdx.x == 0) fb_flag_store(&sync->err, 2);
            return;
        }
    }
    const float inv_ds = std::is_same<TA, f16_t>::value ? 1.0f / dscale_p[0] : 1.0f;      // (a power of two: exact)
    const float c = scale * FB_LOG2E, out_scale = (PRE ? FB_LN2 : scale) * inv_ds;
    int my_jobs = 0;

    // Jobs.  The next job's number is drawn by thread 0 at the START of a job's last query tile and left in LDS in front of that tile's barrier:
    // when the key waves come out of their loop they know where to go, and fetch the next key block's K / V fragments while the query waves are
    // still finishing the chain of this one (~2 tiles) — no queue round trip, no extra barrier and no idle key waves between jobs.
    if (threadIdx.x == 0) s_job[0] = fb_atomic_inc(&sync->head[xcc]);
    __syncthreads();
    int job = od_uniform(s_job[0]), jpar = 0;
    // One job loop PER ROLE (the same code, instantiated twice, the role branch outside): a role's loop-invariant values — LDS offsets, lane
    // constants: a dozen registers each — are then live in its own loop only.  With the branch inside a common loop they were hoisted out of it
    // and stayed live through BOTH roles' inner loops, which are each at the register limit.
    auto run_jobs = [&](auto role_) FB_INLINE {
    constexpr bool KEY_ROLE = decltype(role_)::value;
    for (;; jpar ^= 1) {
        if (job >= jobs_per_xcd) break;
        my_jobs++;
        const int bh = (job / nkb) * 8 + xcc, kblk = job % nkb;
        if (bh >= BH) {                                    // a queue slot past the last (batch, head): draw again, everybody together
            __syncthreads();
            if (threadIdx.x == 0) s_job[jpar ^ 1] = fb_atomic_inc(&sync->head[xcc]);
            __syncthreads();
            job = od_uniform(s_job[jpar ^ 1]);
            continue;
        }
        const int b = bh / H, h = bh % H;
        const bf16_t* qb = q + (size_t)b * L * ldq + h * 64;
        const bf16_t* kb_ = k + (size_t)b * L * ldk + h * 64;
        const bf16_t* vb = v + (size_t)b * L * ldv + h * 64;
        const bf16_t* dob = dout + (size_t)b * L * lddo + h * 64;
        const bool kragged = kblk * KB + KB > L;
#ifndef FB_PRIO
#define FB_PRIO 0         // s_setprio of the key waves (the query waves stay at 0)
#endif
        if constexpr (KEY_ROLE) {
            // =================================================================== key waves
            if (FB_PRIO) __builtin_amdgcn_s_setprio(FB_PRIO);
            const int key0 = kblk * KB + wave * NK * 16;
            od_frag<TA> fk[NK][2], fv[NK][2];
#pragma unroll
            for (int ki = 0; ki < NK; ki++) {
                int row = key0 + ki * 16 + x; row = row < L ? row : L - 1;
#pragma unroll
                for (int s = 0; s < 2; s++) {
                    od_frag_load(fk[ki][s], (const TA*)(kb_ + (size_t)row * ldk + s * 32 + g * 8));
                    od_frag_load(fv[ki][s], (const TA*)(vb + (size_t)row * ldv + s * 32 + g * 8));
                }
            }
            f32x4 dkacc[NK][4], dvacc[NK][4];
#pragma unroll
            for (int ki = 0; ki < NK; ki++)
#pragma unroll
                for (int dt = 0; dt < 4; dt++) { dkacc[ki][dt] = (f32x4)(0.f); dvacc[ki][dt] = (f32x4)(0.f); }
            // lane-constant LDS offset of the dS slots this lane writes: row wave*48 + ki*16 + x (the swizzle key depends on x alone: the rest of
            // the row index is a multiple of 16), slot (4u + g) ^ key — u = 1 flips bit 2 of the slot, i.e. byte 64 of the offset
            const int ds_off0 = (wave * 48 + x) * 128 + ((g ^ fb_swz(x)) << 4);
            __syncthreads();                               // tile 0 has landed (query waves waited for their DMA)
            unsigned long long pk_bar = 0;
            int next_job = 0;
            auto tile = [&](int qt, auto masked_t) FB_INLINE {
                constexpr bool MASKED = decltype(masked_t)::value;
                const bool last_tile = qt == nqt - 1;
                if (last_tile && threadIdx.x == 0) next_job = fb_atomic_inc(&sync->head[xcc]);     // used a tile later
                const unsigned char* st = smem + (qt & 1) * FB_STAGE;
                unsigned char* dsb = smem + 2 * FB_STAGE + (qt & 1) * FB_DS;
                const unsigned char* tQ = st;
                const unsigned char* tO = st + FB_TILE;
                const float* s_nl = (const float*)(st + 2 * FB_TILE);
                const float* s_nd = s_nl + 64;
                const int qbase = qt * 64;
                // one 32-query half (two 16-row tiles, the k-slab u of the dV / dK products) at a time: its P / dS fragments are consumed by
                // their MFMAs before the next half's are formed — half the live fragment registers of a whole-tile pass
#pragma unroll
                for (int u = 0; u < 2; u++) {
                    od_frag<TA> fp[NK], fds[NK];
#pragma unroll
                    for (int th = 0; th < 2; th++) {
                        const int t4 = 2 * u + th;
                        od_frag<TA> fqr[2], fdo[2];
#pragma unroll
                        for (int s = 0; s < 2; s++) {
                            if ((FB_X & 1024) && t4) { fqr[s] = fk[0][s]; fdo[s] = fv[0][s]; continue; }
                            frag_contig<128>(fqr[s], tQ, t4 * 16 + x, s * 32 + g * 8);
                            frag_contig<128>(fdo[s], tO, t4 * 16 + x, s * 32 + g * 8);
                        }
                        const f32x4 l4 = *(const f32x4*)(s_nl + t4 * 16 + 4 * g);
                        const f32x4 d4 = *(const f32x4*)(s_nd + t4 * 16 + 4 * g);
#pragma unroll
                        for (int ki = 0; ki < NK; ki++) {
                            // (issuing the six chains of a 16-query tile step by step — no MFMA right behind the one it depends on — shortens this
                            // loop by 0.7 % and lengthens the call by 1.5 %: the clumped MFMAs hold up the query wave; profiles/r04aj)
                            f32x4 sa_ = l4, pa_ = d4;
#pragma unroll
                            for (int s = 0; s < 2; s++) { sa_ = od_mma(fqr[s], fk[ki][s], sa_); pa_ = od_mma(fdo[s], fv[ki][s], pa_); }
                            f32x4 e = sa_;
                            if constexpr (!PRE) e = od_mul4s(sa_, c);
                            f32x4 p;
#pragma unroll
                            for (int r = 0; r < 4; r++) p[r] = (FB_X & 256) ? e[r] : od_exp2(e[r]);
                            if constexpr (MASKED) {
                                const bool kvalid = key0 + ki * 16 + x < L;
#pragma unroll
                                for (int r = 0; r < 4; r++)
                                    if (!(kvalid && (qbase + t4 * 16 + 4 * g + r < L))) p[r] = 0.f;
                            }
                            const f32x4 ds = od_mul4(p, pa_);
                            od_frag_set4(fp[ki], th, p[0], p[1], p[2], p[3]);
                            od_frag_set4(fds[ki], th, ds[0], ds[1], ds[2], ds[3]);
                        }
                    }
                    // dV^T += dO^T P ; dK^T += Q^T dS     (A rows = features, k = permuted queries of this half, cols = keys)
                    // and, one per feature tile, the publication of dS: fragment u of key tile ki = queries {32u + 4g + r, 32u + 16 + 4g + r} of
                    // key row (wave, ki, x), one 16-byte slot.  A ds_write_b128 holds the LDS store path for ~13 cycles; three in a row in front
                    // of the MFMAs were fully exposed (1.5 ms per call), one between each group of six MFMAs hides behind them.
                    // The transposed fragments of feature tile dt + 1 are read BEFORE the six MFMAs of tile dt: read -> wait -> MFMA per tile left the
                    // LDS round trip of the transpose reads exposed eight times per query tile (this wave is alone with its MFMAs on the SIMD).
                    od_frag<TA> fot[2], fqt[2];
                    frag_cols<128, 128>(fot[0], tO, tO, 0, x, u, g);
                    frag_cols<128, 128>(fqt[0], tQ, tQ, 0, x, u, g);
#pragma unroll
                    for (int dt = 0; dt < 4; dt++) {
                        if (dt < 3 && !(FB_X & 512)) {
                            frag_cols<128, 128>(fot[(dt + 1) & 1], tO, tO, (dt + 1) * 16, x, u, g);
                            frag_cols<128, 128>(fqt[(dt + 1) & 1], tQ, tQ, (dt + 1) * 16, x, u, g);
                        }
                        if (dt < NK && !(FB_X & 4)) *(s16x8*)(dsb + ((ds_off0 ^ (u * 64)) + dt * 2048)) = fds[dt].v;
                        __builtin_amdgcn_sched_barrier(0);               // the reads stay above this tile's MFMAs
#pragma unroll
                        for (int ki = 0; ki < NK; ki++) {
                            dvacc[ki][dt] = od_mma(fot[dt & 1], fp[ki], dvacc[ki][dt]);
                            dkacc[ki][dt] = od_mma(fqt[dt & 1], fds[ki], dkacc[ki][dt]);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
                if (last_tile && threadIdx.x == 0) s_job[jpar ^ 1] = next_job;
                const unsigned long long tb0 = FB_CLK();
                __syncthreads();
                pk_bar += FB_CLK() - tb0;
            };
            const unsigned long long tk0 = FB_CLK();
            const int nfull = kragged ? 0 : L / 64;
            for (int qt = 0; qt < nfull; qt++) tile(qt, std::false_type{});
            for (int qt = nfull; qt < nqt; qt++) tile(qt, std::true_type{});
            if (FB_PROF && threadIdx.x == 0) { atomicAdd(&sync->prof[0], pk_bar); atomicAdd(&sync->prof[1], FB_CLK() - tk0); }
#pragma unroll
            for (int ki = 0; ki < NK; ki++) {
                const int row = key0 + ki * 16 + x;
                if (row < L) {
                    bf16_t* dkr = dk + ((size_t)b * L + row) * lddk + h * 64;
                    bf16_t* dvr = dv + ((size_t)b * L + row) * lddv + h * 64;
#pragma unroll
                    for (int dt = 0; dt < 4; dt++) {
                        const f32x4 a = dkacc[ki][dt] * out_scale, e = dvacc[ki][dt] * inv_ds;
                        u32x2 w0, w1;
                        w0[0] = od_pack_bf2(a[0], a[1]); w0[1] = od_pack_bf2(a[2], a[3]);
                        w1[0] = od_pack_bf2(e[0], e[1]); w1[1] = od_pack_bf2(e[2], e[3]);
                        *(u32x2*)(dkr + dt * 16 + 4 * g) = w0;
                        *(u32x2*)(dvr + dt * 16 + 4 * g) = w1;
                    }
                }
            }
        } else {
            // =================================================================== query waves
            const int hh = wave - 4;
            // The dQ product of a tile, dQ^T[64 features x 64 queries] = K^T[64 x 192 keys] dS[192 x 64], is split by QUERIES: wave h owns query group h
            // (16 queries) for all four feature tiles — it reads a quarter of the dS image (12 transpose reads per tile; a split by features reads
            // all of it, 48) and holds the whole K^T of the block in 96 registers.  A operand of feature tile j, slab s (32 keys): row = feature
            // 16 j + x, k-slot (g, i) = key 32 s + 8 g + i of the block: the transposed image of the block's K rows, which the four query waves stage
            // in LDS once per job (DMA, in the dS image's swizzle) and read back with the same transpose reads as a dS tile.  (Gathered from global
            // memory element by element — 192 two-byte loads per lane — the job's prologue cost more than the split saved: 23.2 ms per call against
            // 21.6 for the split by features; staged, 20.8: profiles/r04al.)
            // (FB_SPLIT: 0 = by features — wave h owns feature tile h for all four query groups —, 1 = by queries, 2 = two feature tiles x two groups)
            constexpr int NJ = FB_SPLIT == 0 ? 1 : FB_SPLIT == 1 ? 4 : 2, NG = 4 / NJ;     // feature tiles / query groups of a wave; accumulator a = (ji, gi)
            auto ji_of = [](int a) FB_INLINE { return FB_SPLIT == 0 ? 0 : FB_SPLIT == 1 ? a : a >> 1; };
            auto gi_of = [](int a) FB_INLINE { return FB_SPLIT == 0 ? a : FB_SPLIT == 1 ? 0 : a & 1; };
            const int j0 = FB_SPLIT == 0 ? hh : FB_SPLIT == 1 ? 0 : 2 * (hh & 1), G0 = FB_SPLIT == 0 ? 0 : FB_SPLIT == 1 ? hh : 2 * (hh >> 1);
            od_frag<TA> fkt[NJ][6];                                  // filled from the staged K rows after the job's first barrier (below)
            // Q / dO tile streaming: piece = 8 rows x 128 B; this wave moves pieces hh and hh + 4 of both tiles; waves 0 / 1 also move the
            // 64 start values (-lse', -delta).  Rows past L lie beyond the descriptors and read as zero.
            const od_srd_t rq = od_make_srd(qb, (unsigned)(((size_t)(L - 1) * ldq + 64) * 2));
            const od_srd_t ro = od_make_srd(dob, (unsigned)(((size_t)(L - 1) * lddo + 64) * 2));
            const od_srd_t rs = od_make_srd((hh == 0 ? nl : nd) + (size_t)bh * L, (unsigned)L * 4u);
            const int prow = lane >> 3, pslot = (lane & 7) ^ prow;          // tile_off<128>: slot ^ (row & 7), row & 7 = lane >> 3
            const unsigned vq = (unsigned)((hh * 8 + prow) * ldq * 2 + pslot * 16), vo = (unsigned)((hh * 8 + prow) * lddo * 2 + pslot * 16);
            const unsigned smem_a = od_lds_addr(smem);
            // K rows of the block -> LDS: 24 pieces of 8 rows x 128 B, six per wave; LDS slot c of row r holds chunk c ^ fb_swz(r) (the swizzle is
            // applied on the SOURCE address: the DMA writes linearly).  Rows past L lie beyond the descriptor and read as zero.
            const od_srd_t rk = od_make_srd(kb_, (unsigned)(((size_t)(L - 1) * ldk + 64) * 2));
            auto stage_k = [&]() FB_INLINE {
#pragma unroll
                for (int i = 0; i < 6; i++) {
                    const int pc = hh + 4 * i;                          // piece: rows 8 pc .. 8 pc + 7 of the block
                    const int r = 8 * pc + (lane >> 3);
                    const unsigned vk = (unsigned)((lane >> 3) * ldk * 2 + (((lane & 7) ^ fb_swz(r)) << 4));
                    od_buffer_lds16_at(rk, vk, (unsigned)od_uniform((kblk * KB + 8 * pc) * ldk * 2), (unsigned)od_uniform((int)(smem_a + (unsigned)FB_KST + (unsigned)pc * 1024u)));
                }
            };
            int trk[NJ][2];                                              // transpose-read offsets of feature tile j0 + ji in the staged image (as tr_off below)
#pragma unroll
            for (int ji = 0; ji < NJ; ji++)
#pragma unroll
                for (int e = 0; e < 2; e++) {
                    const int row = 8 * g + (x >> 2) + 4 * e;
                    trk[ji][e] = row * 128 + (((2 * (j0 + ji) + ((x & 3) >> 1)) ^ fb_swz(row)) << 4) + (x & 1) * 8;
                }
            // the tile DMA in five single-instruction pieces (0, 1: Q rows 0..31 / 32..63; 2, 3: dO; 4: the tile's -lse / -delta rows): a wave's vector-memory
            // instruction issues only when the CU's address pipeline (64 bytes per clock for the DMA, the running-tile loads AND stores: ~1000 clocks
            // per tile) takes it, so the pieces go out one at a time between the slabs of the dQ product instead of back to back in front of it
            auto dma_piece = [&](int qt, int j) FB_INLINE {
                const unsigned st = (unsigned)od_uniform((int)(smem_a + (unsigned)((qt & 1) * FB_STAGE) + (unsigned)hh * 1024u));     // this wave's piece
                const unsigned qi = (unsigned)od_uniform(qt);
                const unsigned sq = qi * 64u * (unsigned)ldq * 2u, so = qi * 64u * (unsigned)lddo * 2u;
                if (j == 0) od_buffer_lds16_at(rq, vq, sq, st);
                if (j == 1) od_buffer_lds16_at(rq, vq, sq + 32u * (unsigned)ldq * 2u, st + 4096);
                if (j == 2) od_buffer_lds16_at(ro, vo, so, st + FB_TILE);
                if (j == 3) od_buffer_lds16_at(ro, vo, so + 32u * (unsigned)lddo * 2u, st + FB_TILE + 4096);
                if (j == 4 && hh < 2) od_buffer_lds4_at(rs, (unsigned)lane * 4u, qi * 256u, (unsigned)od_uniform((int)(st + 2 * FB_TILE - (unsigned)hh * 768u)));
            };
            auto dma = [&](int qt) FB_INLINE {
                qt = od_uniform(qt);
#pragma unroll
                for (int j = 0; j < 5; j++) dma_piece(qt, j);
            };
            // lane-constant LDS offsets of the dS^T transpose-read chunks: query group G (16 columns = slots 2G, 2G+1), slab rows 8 g + (x >> 2) (+4)
            int tr_off[NG][2], qcol[NG];                                 // qcol: the query this lane's accumulator column stands for (the order the key waves packed them in)
#pragma unroll
            for (int gi = 0; gi < NG; gi++) {
                const int G = G0 + gi;
#pragma unroll
                for (int e = 0; e < 2; e++) {
                    const int row = 8 * g + (x >> 2) + 4 * e;           // + 32 s: bits 3.. of the row unchanged mod 16 -> same swizzle
                    tr_off[gi][e] = row * 128 + (((2 * G + ((x & 3) >> 1)) ^ fb_swz(row)) << 4) + (x & 1) * 8;
                }
                qcol[gi] = 16 * (2 * (G >> 1) + ((x >> 2) & 1)) + 4 * (2 * (G & 1) + (x >> 3)) + (x & 3);
            }

            // ---- chain pipeline.  Every dword of a running tile carries a 2-bit tag in its lowest mantissa bits that identifies its writer (below).  The consumer simply loads the tile and looks at the tags — no flag, no acknowledgement to wait for: a tile that is
            // not there yet (or half there) shows a wrong tag somewhere and is loaded again.  Per tile tau:
            //   it = tau + 1: share -> acc[tau & 1]
            //   it = tau + 2: (top) load the running tile; (end) tags good -> running + share -> tagged stores (last key block: the bf16 dQ)
            // A key block therefore trails its predecessor by ~1.5 iterations; consecutive key blocks of a (batch, head) START nqt / (CUs per XCD)
            // iterations apart (they are consecutive jobs of one queue), which has to stay above that: the host routes short sequences elsewhere.
            f32x4 acc[4][4], rb[FB_NP];
            bool poisoned = false;                                        // the tile in rb could not be had (chain_try): its values are NaN
            unsigned long long ph[6] = {0, 0, 0, 0, 0, 0};       // FB_PROF cycle counters
            const size_t tile_f = FB_RUN_TILE;                           // floats of one (bh, tile): [hh][G][lane][4]
            const bool last = kblk == nkb - 1;
            // Tags.  The running tiles of a (batch, head) live in one of FB_SLOTS slots of its XCD — slot (bh / 8) % FB_SLOTS, reused by every
            // FB_SLOTS-th (batch, head) of that XCD's queue: 64 MB instead of a tile per (batch, head, query tile), small enough to stay in the
            // Infinity Cache.  EVERY key block reads the tile and writes it back (the first ignores what it reads, the last writes a bare marker
            // beside dQ), always in the same order, so the writes of a slot can be numbered through the life of the workspace:
            //     c = 1 + (launches * G + generation) * nkb + key block        (G = generations of this slot per launch; a zero-filled slot is "write 0")
            // and a key block proceeds only when the tile carries c - 1: its predecessor's write — or, for the first key block, the LAST block's marker
            // of the slot's previous user, i.e. that user has read what is about to be overwritten.  Every 16-byte piece (one lane of one load /
            // store instruction: the unit the memory system moves) carries c mod 4096, three bits in the lowest mantissa bits of each of its four
            // dwords (2^-20 relative per hop: far below the bf16 result's own rounding).  A piece is accepted only with exactly that number: the
            // consumer polls a line that is being rewritten, and what comes back is not always the newest version (measured: under a burst of L2
            // evictions a poll can return the line as it was two or four writes ago, whole 128-byte lines at a time) — every piece has to prove
            // which write it belongs to.  (Three values + a 32-bit number per piece was built too: exact values, a third of the vector-ALU work,
            // but six pieces instead of four per tile and wave — and this path is bound by the bytes it moves: 25.2 ms per call against 23.9,
            // profiles/r04o_tagword.txt.)
            const int bhq = bh >> 3, slot = bhq % FB_SLOTS, gen = bhq / FB_SLOTS;
            const int nbhq_x = (BH - xcc + 7) >> 3;                      // (batch, head)s of this XCD's queue
            const int gens = (nbhq_x - slot + FB_SLOTS - 1) / FB_SLOTS;  // ... that use this slot
            const unsigned id_mine = (unsigned)od_uniform((int)((1u + ((unsigned)calls * (unsigned)gens + (unsigned)gen) * (unsigned)nkb + (unsigned)kblk) & 0xfffu)), id_pred = (id_mine - 1u) & 0xfffu;
            auto tag_of = [](unsigned id, int r) FB_INLINE -> unsigned { return (id >> (3 * r)) & 7u; };
            // dQ^T share of tile tau from the dS image it was published in.  The transpose reads of slab s + 1 are issued before the MFMAs of slab s:
            // read -> wait -> MFMA one fragment at a time (what the plain loop compiles to) costs an LDS round trip per MFMA, ~3000 cycles per
            // tile, which made this wave — not the key wave beside it — the pace of the workgroup.
            auto phase_b = [&](int tau, f32x4 (&a)[4], auto&& between) FB_INLINE {          // between(s): vector-memory work to issue after slab s
                const unsigned char* dsb = smem + 2 * FB_STAGE + (tau & 1) * FB_DS;
                s16x4 f[2][NG][2];                                       // [buffer][group][half]
                auto rd = [&](int s, s16x4 (&dst)[NG][2]) FB_INLINE {              // the 2 NG transpose reads of slab s
#pragma unroll
                    for (int gi = 0; gi < NG; gi++)
#pragma unroll
                        for (int e = 0; e < 2; e++) dst[gi][e] = od_lds_tr_read((const bf16_t*)(dsb + s * 4096 + tr_off[gi][e]));
                };
                rd(0, f[0]);
#pragma unroll
                for (int s = 0; s < 6; s++) {
                    if (s + 1 < 6) rd(s + 1, f[(s + 1) & 1]);
                    __builtin_amdgcn_sched_barrier(0);                   // keep the reads of the next slab ABOVE this slab's MFMAs
                    od_frag<TA> fb[NG];
#pragma unroll
                    for (int gi = 0; gi < NG; gi++) {
                        const s16x4 b0 = f[s & 1][gi][0], b1 = f[s & 1][gi][1];
                        fb[gi].v[0] = b0[0]; fb[gi].v[1] = b0[1]; fb[gi].v[2] = b0[2]; fb[gi].v[3] = b0[3];
                        fb[gi].v[4] = b1[0]; fb[gi].v[5] = b1[1]; fb[gi].v[6] = b1[2]; fb[gi].v[7] = b1[3];
                    }
#pragma unroll
                    for (int G = 0; G < 4; G++) a[G] = od_mma(fkt[ji_of(G)][s], fb[gi_of(G)], s == 0 ? (f32x4)(0.f) : a[G]);
                    __builtin_amdgcn_sched_barrier(0);
                    between(s);
                }
            };
            const size_t rbh = (size_t)od_uniform(xcc * FB_SLOTS + slot);
            const fb_rsrc_t run_rs = fb_make_rsrc(run + rbh * nqt * tile_f, (unsigned)((size_t)nqt * tile_f * 4));
            const unsigned run_vo = (unsigned)((hh * (FB_NP * 256) + lane * 4) * 4);
            float* const run_st = run + rbh * nqt * tile_f + (size_t)hh * (FB_NP * 256) + (size_t)lane * 4;
            auto chain_load = [&](int tau, f32x4 (&r4)[FB_NP]) FB_INLINE {
#pragma unroll
                for (int G = 0; G < FB_NP; G++) r4[G] = fb_ld_l2(run_rs, run_vo + G * 1024, (unsigned)((FB_X & 8) ? 0 : tau) * (unsigned)(FB_RUN_TILE * 4));
            };
            auto tags_good = [&](f32x4 (&r4)[FB_NP]) FB_INLINE -> bool {   // every piece of the wave's slice carries the predecessor's number
                unsigned bad = 0;
#if FB_PACK
#pragma unroll
                for (int P = 0; P < FB_NP; P++) bad |= ((fb_f2u(r4[P][3]) >> 14) ^ id_pred) & 0x3ffu;      // ten bits of the number in every piece
                if (FB_X & 32) return true;
                return !__any(bad != 0);
#else
#pragma unroll
                for (int G = 0; G < 4; G++)
#pragma unroll
                    for (int r = 0; r < 4; r++) bad |= fb_f2u(r4[G][r]) ^ tag_of(id_pred, r);
                if (FB_X & 32) return true;
                return !__any((bad & 7u) != 0);
#endif
            };
#if FB_PACK
            // The packed tile.  A lane's 16 values (feature tile G = i / 4, row r = i % 4) travel as 22-bit floats u = (bits + 0x200) >> 10 —
            // sign, exponent, 13 mantissa bits, rounded to nearest: 2^-14 relative per hop, ~2^-12 over the 43 hops of a chain, against the
            // 2^-9 of the bf16 the sum is rounded to at the end.  Piece P (16 bytes, the unit the memory system moves) holds values 5 P .. 5 P + 4
            // in bits [0, 110), ten bits of the write number in [110, 120) and 8 / 7 / 7 bits of value 15 in [120, 128): every piece still proves
            // which write it belongs to, and a tile is three vector-memory instructions per wave and direction instead of four.
            auto fb_unpack = [&](const f32x4 (&r4)[FB_NP], float (&v)[16]) FB_INLINE {
                unsigned last = 0;
#pragma unroll
                for (int P = 0; P < 3; P++) {
                    const unsigned d0 = fb_f2u(r4[P][0]), d1 = fb_f2u(r4[P][1]), d2 = fb_f2u(r4[P][2]), d3 = fb_f2u(r4[P][3]);
                    v[5 * P + 0] = fb_u2f(d0 << 10);
                    v[5 * P + 1] = fb_u2f(((d0 >> 22) | (d1 << 10)) << 10);
                    v[5 * P + 2] = fb_u2f(((d1 >> 12) | (d2 << 20)) << 10);
                    v[5 * P + 3] = fb_u2f((d2 << 8) & 0xfffffc00u);
                    v[5 * P + 4] = fb_u2f(((d2 >> 24) | (d3 << 8)) << 10);
                    last |= (d3 >> 24) << (P == 0 ? 0 : P == 1 ? 8 : 15);
                }
                v[15] = fb_u2f(last << 10);
            };
            auto fb_pack = [&](const float (&v)[16], unsigned id, f32x4 (&out)[FB_NP]) FB_INLINE {
                unsigned u[16];
#pragma unroll
                for (int i = 0; i < 16; i++) u[i] = (fb_f2u(v[i]) + 0x200u) >> 10;
#pragma unroll
                for (int P = 0; P < 3; P++) {
                    const unsigned a0 = u[5 * P], a1 = u[5 * P + 1], a2 = u[5 * P + 2], a3 = u[5 * P + 3], a4 = u[5 * P + 4];
                    const unsigned ex = P == 0 ? (u[15] & 0xffu) : P == 1 ? ((u[15] >> 8) & 0x7fu) : (u[15] >> 15);
                    out[P][0] = fb_u2f(a0 | (a1 << 22));
                    out[P][1] = fb_u2f((a1 >> 10) | (a2 << 12));
                    out[P][2] = fb_u2f((a2 >> 20) | (a3 << 2) | (a4 << 24));
                    out[P][3] = fb_u2f((a4 >> 8) | ((id & 0x3ffu) << 14) | (ex << 24));
                }
            };
#endif
            auto chain_commit = [&](int tau, f32x4 (&a)[4]) FB_INLINE {            // running sum (rb, tags verified) + this block's share -> next block, or the finished dQ
#if FB_PACK
                {
                    float rv[16];
                    fb_unpack(rb, rv);
                    if (poisoned) {
#pragma unroll
                        for (int i = 0; i < 16; i++) rv[i] = fb_poison();
                    }
#pragma unroll
                    for (int G = 0; G < 4; G++)
#pragma unroll
                        for (int r = 0; r < 4; r++) a[G][r] += rv[4 * G + r];                       // (the first key block adds the marker tile's zeros)
                }
                if (!last) {
                    float sv[16];
#pragma unroll
                    for (int G = 0; G < 4; G++)
#pragma unroll
                        for (int r = 0; r < 4; r++) sv[4 * G + r] = a[G][r];
                    f32x4 t3[FB_NP];
                    fb_pack(sv, id_mine, t3);
#pragma unroll
                    for (int P = 0; P < FB_NP; P++) fb_st_run(run_st + (size_t)((FB_X & 8) ? 0 : tau) * tile_f + P * 256, t3[P]);
                } else {
#else
#pragma unroll
                for (int G = 0; G < 4; G++)
#pragma unroll
                    for (int r = 0; r < 4; r++) a[G][r] += fb_u2f(fb_f2u(rb[G][r]) & ~7u);      // (the first key block adds the marker tile's zeros)
                if (!last) {
#pragma unroll
                    for (int G = 0; G < 4; G++) {
                        f32x4 t;
#pragma unroll
                        for (int r = 0; r < 4; r++) t[r] = fb_u2f((fb_f2u(a[G][r]) & ~7u) | tag_of(id_mine, r));
                        fb_st_run(run_st + (size_t)((FB_X & 8) ? 0 : tau) * tile_f + G * 256, t);
                    }
                } else {
#endif
#pragma unroll
                    for (int G = 0; G < 4; G++) {
                        const int qrow = tau * 64 + qcol[gi_of(G)];
                        if (qrow < L) {
                            u32x2 w;
                            w[0] = od_pack_bf2(a[G][0] * out_scale, a[G][1] * out_scale);
                            w[1] = od_pack_bf2(a[G][2] * out_scale, a[G][3] * out_scale);
                            *(u32x2*)(dq + ((size_t)b * L + qrow) * lddq + h * 64 + 16 * (j0 + ji_of(G)) + 4 * g) = w;
                        }
                    }
                    f32x4 t;                                             // the marker "this tile has been read" (tags on zeros): the slot's next user waits for it
#if FB_PACK
                    t[0] = t[1] = t[2] = 0.f; t[3] = fb_u2f((id_mine & 0x3ffu) << 14);
#else
#pragma unroll
                    for (int r = 0; r < 4; r++) t[r] = fb_u2f(tag_of(id_mine, r));
#endif
#pragma unroll
                    for (int G = 0; G < FB_NP; G++) fb_st_run(run_st + (size_t)((FB_X & 8) ? 0 : tau) * tile_f + G * 256, t);
                }
            };
            auto commit_dyn = [&](int tau) FB_INLINE {                             // the share of tile tau sits in ring slot tau & 3 (a wave-uniform switch)
                switch (tau & 3) {
                    case 0: chain_commit(tau, acc[0]); break;
                    case 1: chain_commit(tau, acc[1]); break;
                    case 2: chain_commit(tau, acc[2]); break;
                    default: chain_commit(tau, acc[3]); break;
                }
            };
            // The chain is ELASTIC: `done` tiles have been passed on, tiles done .. produced-1 wait in the ring.  Each iteration tries the oldest
            // waiting tile ONCE — its running tile was requested at the top of the iteration; if the predecessor's write has not arrived (a tag
            // is wrong) the tile simply stays queued and is requested again next iteration.  Only a full ring blocks.  A wave that has caught up
            // with its predecessor thus falls one iteration behind it WITHOUT holding up its own workgroup, and from then on every request
            // finds its tile: neither the L2 round trip nor the predecessor's jitter is ever inside the workgroup's barrier interval.
            int done = 0;
            bool inflight = false, aborted = false;
            auto chain_request = [&](int produced) FB_INLINE {                     // top of an iteration: tiles < produced have their share in the ring
                if ((FB_X & 1) || inflight || done >= produced) return;
                if (!(FB_X & 64)) chain_load(done, rb);
                inflight = true;
            };
            // returns true when the vector-memory operations it issued are exactly one tile's four running-tile stores
            auto chain_try = [&](int produced, int must_reach) FB_INLINE -> bool { // end of an iteration (the requested tile has landed); must_reach: ring capacity
                if (FB_X & 1) { done = produced; return false; }
                bool four = false;
                if (done < produced) {
                    if (inflight) {
                        inflight = false;
                        if (tags_good(rb)) { commit_dyn(done); done++; four = !last; }
                        else if (FB_PROF) ph[5]++;
                    }
                }
                while (done < must_reach) {                              // the ring is full (or the job ends): now it has to be waited for
                    four = false;
                    // BOUNDED: a predecessor that never publishes (a workspace that was not zeroed, or was left mid-launch by an aborted process; a
                    // launch whose tags do not continue the workspace's) must not hang the GPU — and, under data parallelism, every rank behind
                    // it.  After `wait_ticks` of the 100 MHz counter (~1 s by default) the wave gives up: sticky err = 3, the tile it could not
                    // get is replaced by NaNs (they travel down the chain into dq, and the tags it writes let its successors pass), and every
                    // other wait of the launch ends as soon as it sees the word.  The launch then terminates in its normal time.
                    if (!aborted) {
                        unsigned long long t_start = 0;
                        for (unsigned polls = 0;; polls++) {
                            FB_COMPILER_FENCE();                         // every reload is a NEW read of memory another CU is writing
                            chain_load(done, rb);
                            if (tags_good(rb)) break;
                            if ((polls & 15u) == 15u) {
                                const unsigned long long now = fb_now(polls);
                                if (polls == 15u) t_start = now;
                                else if (now - t_start > (unsigned long long)wait_ticks) aborted = true;
                                if (od_uniform(fb_flag_load(&sync->err)) == 3) aborted = true;
                                if (aborted) break;
                            }
                            fb_sleep();
                        }
                    }
                    if (aborted) {
                        if (lane == 0) fb_flag_store(&sync->err, 3);
#pragma unroll
                        for (int G = 0; G < FB_NP; G++) rb[G] = (f32x4)(fb_poison());
                        poisoned = true;
                    }
                    commit_dyn(done);
                    poisoned = false;
                    done++;
                }
                return four;
            };
            auto step = [&](int it, auto k_) FB_INLINE {
                constexpr int K = decltype(k_)::value;                   // it & 3: the ring slot phase_b writes is compile-time
                const unsigned long long t0 = FB_CLK();
                const int nxt = od_uniform(it + 1);
                const unsigned long long t1 = FB_CLK();
                if (!(FB_X & 2)) phase_b(it - 1, acc[(K + 3) & 3], [&](int s) FB_INLINE {
#if FB_ORDER == 2
                    if (s < 5) { if (nxt < nqt) dma_piece(nxt, s); }
                    else chain_request(it - 1);
#elif FB_ORDER == 1
                    if (s == 0) chain_request(it - 1);
                    else if (nxt < nqt) dma_piece(nxt, s - 1);
#else
                    // the DMA first (two pieces after slab 0, two after slab 1), then the running tile: the loads are compiler-visible, and the compiler
                    // waits for a load with a count of the operations IT knows of — the asm DMA behind it would be waited for as well
                    if (s < 2) { if (nxt < nqt) { dma_piece(nxt, 2 * s); dma_piece(nxt, 2 * s + 1); } }
                    else if (s == 2) { if (nxt < nqt) dma_piece(nxt, 4); }
                    if (s == FB_REQ) chain_request(it - 1);              // tiles 0 .. it-2 had their share in the ring when the last iteration ended
#endif
                });
                else { chain_request(it - 1); if (nxt < nqt) dma(nxt); }
#if FB_PROF && !defined(OD_EMU)
                asm volatile("" : "+v"(acc[(K + 3) & 3][0]), "+v"(acc[(K + 3) & 3][1]), "+v"(acc[(K + 3) & 3][2]), "+v"(acc[(K + 3) & 3][3]));
#endif
                const unsigned long long t2 = FB_CLK();
                FB_WAIT_ALL();                                           // the tile DMA and the requested running tile have landed
                const unsigned long long t3 = FB_CLK();
                const bool four = chain_try(it, it - 3);                 // the next share goes to slot it & 3, tile it - 4's: tiles < it - 3 must be gone
                const unsigned long long t4 = FB_CLK();
                (void)four;
                od_barrier_raw();                                        // bare: a fence here would wait for the stores just issued
                if (FB_PROF) { const unsigned long long t5 = FB_CLK(); ph[0] += t1 - t0; ph[1] += t2 - t1; ph[2] += t3 - t2; ph[3] += t4 - t3; ph[4] += t5 - t4; }
            };
            OD_DRAIN_VMEM();
            stage_k();
            dma(0);
            FB_WAIT_ALL();
            od_barrier_raw();                              // tile 0 visible to the key waves, the staged K rows to the query waves
#pragma unroll
            for (int s = 0; s < 6; s++)
#pragma unroll
                for (int ji = 0; ji < NJ; ji++) {
                    const s16x4 b0 = od_lds_tr_read((const bf16_t*)(smem + FB_KST + s * 4096 + trk[ji][0]));
                    const s16x4 b1 = od_lds_tr_read((const bf16_t*)(smem + FB_KST + s * 4096 + trk[ji][1]));
                    fkt[ji][s].v[0] = b0[0]; fkt[ji][s].v[1] = b0[1]; fkt[ji][s].v[2] = b0[2]; fkt[ji][s].v[3] = b0[3];
                    fkt[ji][s].v[4] = b1[0]; fkt[ji][s].v[5] = b1[1]; fkt[ji][s].v[6] = b1[2]; fkt[ji][s].v[7] = b1[3];
                }
            // it = 0: nothing to turn into dQ yet
            if (1 < nqt) dma(1);
            FB_WAIT_ALL();
            od_barrier_raw();
            int it = 1;
            for (; it + 3 < nqt; it += 4) {
                step(it, std::integral_constant<int, 1>{}); step(it + 1, std::integral_constant<int, 2>{});
                step(it + 2, std::integral_constant<int, 3>{}); step(it + 3, std::integral_constant<int, 0>{});
            }
            if (it < nqt) { step(it, std::integral_constant<int, 1>{}); it++; }
            if (it < nqt) { step(it, std::integral_constant<int, 2>{}); it++; }
            if (it < nqt) { step(it, std::integral_constant<int, 3>{}); it++; }
            // the last tile's share (its dS was published before the final barrier), then whatever is still queued
            if (!(FB_X & 2)) {
                auto nothing = [](int) FB_INLINE {};
                switch ((nqt - 1) & 3) {
                    case 0: phase_b(nqt - 1, acc[0], nothing); break;
                    case 1: phase_b(nqt - 1, acc[1], nothing); break;
                    case 2: phase_b(nqt - 1, acc[2], nothing); break;
                    default: phase_b(nqt - 1, acc[3], nothing); break;
                }
            }
            inflight = false;
            chain_try(nqt, nqt);
            if (FB_PROF && threadIdx.x == 256) { for (int i = 0; i < 6; i++) atomicAdd(&sync->prof[2 + i], ph[i]); }
            FB_WAIT_ALL();
        }
        job = od_uniform(s_job[jpar ^ 1]);               // left there in front of the last tile's barrier
    }
    };
    if (key_wave) run_jobs(std::true_type{}); else run_jobs(std::false_type{});
    // ---- leave: the last workgroup out checks that every job ran and re-arms the control block for the next launch
    if (threadIdx.x == 0) {
        if (my_jobs) {
#if defined(OD_EMU)
            sync->jobs_done += my_jobs;
#else
            __hip_atomic_fetch_add(&sync->jobs_done, my_jobs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
        }
#if !defined(OD_EMU)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
        const int t = fb_atomic_inc(&sync->finished);
        if (t == (int)gridDim.x - 1) {
#if !defined(OD_EMU)
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
#endif
            const int done = fb_flag_load(&sync->jobs_done);
            if (done != 8 * jobs_per_xcd) fb_flag_store(&sync->err, 1);
            for (int i = 0; i < 8; i++) fb_flag_store(&sync->head[i], 0);
            fb_flag_store(&sync->jobs_done, 0);
            fb_flag_store(&sync->calls, sync->calls + 1);
            fb_flag_store(&sync->shape[0], B); fb_flag_store(&sync->shape[1], H); fb_flag_store(&sync->shape[2], L);
            fb_flag_store(&sync->finished, 0);
        }
    }
}

inline size_t fb_align(size_t n) { return (n + 255) & ~(size_t)255; }
struct FbLayout { size_t sync, nl, nd, run, aux, do16, total; };
inline FbLayout fb_layout(int B, int H, int L, bool f16 = false) {
    const size_t BH = (size_t)B * H, nqt = (size_t)(L + 63) / 64;
    FbLayout l;
    l.sync = 0;
    l.run = fb_align(sizeof(FbSync));
    l.nl = l.run + fb_align((size_t)8 * FB_SLOTS * nqt * FB_RUN_TILE * sizeof(float));      // [0, nl): what has to be zero before the first launch
    l.nd = l.nl + fb_align(BH * L * sizeof(float));
    l.aux = l.nd + fb_align(BH * L * sizeof(float));                // half form: [0] = amax bits of dout, [1] = dscale
    l.do16 = l.aux + 256;                                            // half form: the staged dout, [B L][H 64]
    l.total = f16 ? l.do16 + fb_align(BH * L * 64 * sizeof(unsigned short)) : l.aux;
    return l;
}

// Budget of one blocking chain wait in ticks of the 100 MHz counter: OD_FB_CHAIN_TIMEOUT_MS (default 1000; read once).  A healthy launch never
// waits longer than a few microseconds (the predecessor is running or done by construction); the bound exists so that a corrupt workspace ends
// as status 3 + NaN gradients instead of a hung GPU.
inline unsigned fb_wait_ticks() {
    static const unsigned ticks = [] {
        const char* e = getenv("OD_FB_CHAIN_TIMEOUT_MS");
        long ms = e ? atol(e) : 1000;
        if (ms < 1) ms = 1;
        if (ms > 40000) ms = 40000;
#if defined(OD_EMU)
        return (unsigned)1024;                 // emulator: polls, not time
#else
        return (unsigned)(ms * 100000);
#endif
    }();
    return ticks;
}

}  // namespace

// Bytes of the caller-owned workspace of od_flash_attn_bwd_fused (*total_out) and of its head (*zero_out: the control block and the running
// tiles), which must be ZERO before the first launch (the start values behind it are written by every launch before they are read).
extern "C" int od_flash_attn_bwd_fused_ws_bytes(int dtype, int B, int H, int L, long* total_out, long* zero_out) {
    if (B <= 0 || H <= 0 || L <= 0 || !total_out || !zero_out || (dtype != OD_BF16 && dtype != OD_F16)) return OD_ERR_ARG;
    const FbLayout lay = fb_layout(B, H, L, dtype == OD_F16);
    *total_out = (long)lay.total;
    *zero_out = (long)lay.nl;
    return 0;
}
// byte offset of the sticky error word (an int) inside a workspace: od_sqnorm takes its address and folds it into the gradient norm ON THE DEVICE
extern "C" int od_flash_attn_bwd_fused_err_offset(void) { return (int)offsetof(FbSync, err); }
extern "C" int od_flash_attn_bwd_fused_passes(void) { return 5; }
// FB_PROF builds: copies the 16 cycle counters to the host and clears them
extern "C" int od_flash_attn_bwd_fused_prof(void* ws, long* out16) {
    if (!ws || !out16) return OD_ERR_ARG;
    FbSync s;
#if defined(OD_EMU)
    s = *(FbSync*)ws;
#else
    if (hipDeviceSynchronize() != hipSuccess) return OD_ERR_ARG;      // (a profiling aid: the blocking copy alone orders against the null stream only)
    if (hipMemcpy(&s, ws, sizeof(s), hipMemcpyDeviceToHost) != hipSuccess) return OD_ERR_ARG;
    if (hipMemset((unsigned char*)ws + offsetof(FbSync, prof), 0, sizeof(s.prof)) != hipSuccess) return OD_ERR_ARG;
#endif
    for (int i = 0; i < 16; i++) out16[i] = (long)s.prof[i];
    return 0;
}

extern "C" int od_flash_attn_bwd_fused(int dtype, const void* q, int ldq, const void* k, int ldk, const void* v, int ldv, const void* o, int ldo,
                                       const void* dout, int lddo, const float* lse, void* dq, int lddq, void* dk, int lddk, void* dv, int lddv,
                                       int B, int H, int L, int hd, float scale, int q_prescaled, void* ws, long ws_bytes, void* stream) {
    if ((dtype != OD_BF16 && dtype != OD_F16) || hd != 64) return OD_ERR_UNSUPPORTED;
    if (!ws || B <= 0 || H <= 0 || L <= 0) return OD_ERR_ARG;
    if (ldq % 8 || ldk % 8 || ldv % 8 || ldo % 8 || lddo % 8 || lddq % 8 || lddk % 8 || lddv % 8) return OD_ERR_ALIGN;
    const bool f16 = dtype == OD_F16;
    const FbLayout lay = fb_layout(B, H, L, f16);
    if (ws_bytes < (long)lay.total) return OD_ERR_ARG;
    if (((size_t)L * (size_t)(ldq > lddo ? (ldq > ldk ? ldq : ldk) : (lddo > ldk ? lddo : ldk))) * 2 >= 0xffffffffull) return OD_ERR_UNSUPPORTED;      // 32-bit buffer offsets
    hipStream_t st = (hipStream_t)stream;
    unsigned char* w = (unsigned char*)ws;
    float* nl = (float*)(w + lay.nl);
    float* nd = (float*)(w + lay.nd);
    const long M = (long)B * L;
    const float inv_scale = q_prescaled ? FB_LOG2E : 1.0f / scale;
    const float* dscale_p = nullptr;
    if (f16) {
        // half form: amax of dout -> power-of-two scale -> staged half copy (dout itself is bf16, like every other activation gradient)
        unsigned* amax = (unsigned*)(w + lay.aux);
        float* dsc = (float*)(w + lay.aux) + 1;
        f16_t* do16 = (f16_t*)(w + lay.do16);
#if defined(OD_EMU)
        *amax = 0u;
#else
        if (hipMemsetAsync(amax, 0, sizeof(unsigned), st) != hipSuccess) return OD_ERR_ARG;
#endif
        OD_LAUNCH(fb_amax_kernel, dim3(2048), dim3(256), 0, st, (const bf16_t*)dout, lddo, M, H * 64, amax);
        OD_LAUNCH(fb_prep_f16_kernel, dim3((unsigned)((M + 3) / 4)), dim3(256), 0, st, (const bf16_t*)o, ldo, (const bf16_t*)dout, lddo, lse, nl, nd,
                  do16, H * 64, (const unsigned*)amax, dsc, B, H, L, inv_scale);
        dout = do16; lddo = H * 64; dscale_p = dsc;
    } else
        OD_LAUNCH(fb_prep_kernel, dim3((unsigned)((M + 3) / 4)), dim3(256), 0, st, (const bf16_t*)o, ldo, (const bf16_t*)dout, lddo, lse, nl, nd, B, H, L, inv_scale);
    const int grid = od_num_cus();
#define FB_ARGS dim3(grid), dim3(512), FB_SMEM, st, (const bf16_t*)q, ldq, (const bf16_t*)k, ldk, (const bf16_t*)v, ldv, (const bf16_t*)dout, lddo, \
        (const float*)nl, (const float*)nd, (bf16_t*)dq, lddq, (bf16_t*)dk, lddk, (bf16_t*)dv, lddv, (float*)(w + lay.run), (FbSync*)(w + lay.sync), B, H, L, scale, \
        fb_wait_ticks(), dscale_p
#define FB_LAUNCH_X(...) OD_LAUNCH_DYN(__VA_ARGS__)      /* lets the argument-list macro expand first */
    if (f16) {
        if (q_prescaled) FB_LAUNCH_X((flash_bwd_fused_kernel<true, f16_t>), FB_ARGS);
        else FB_LAUNCH_X((flash_bwd_fused_kernel<false, f16_t>), FB_ARGS);
    } else {
        if (q_prescaled) FB_LAUNCH_X((flash_bwd_fused_kernel<true, bf16_t>), FB_ARGS);
        else FB_LAUNCH_X((flash_bwd_fused_kernel<false, bf16_t>), FB_ARGS);
    }
#undef FB_LAUNCH_X
#undef FB_ARGS
    OD_CHECK_LAUNCH();
    return 0;
}

// the sticky error word of a workspace (0 = every launch processed every job); a host-side read: the copy is enqueued on `stream` — the
// stream the launches went to — and that stream is waited for (a blocking hipMemcpy would order against the null stream only, and the
// streams torch hands out do not synchronise with it)
extern "C" int od_flash_attn_bwd_fused_status(const void* ws, int* err_out, void* stream) {
    if (!ws || !err_out) return OD_ERR_ARG;
#if defined(OD_EMU)
    (void)stream;
    *err_out = ((const FbSync*)ws)->err;
#else
    int v = -1;
    hipError_t e = hipMemcpyAsync(&v, (const unsigned char*)ws + offsetof(FbSync, err), sizeof(int), hipMemcpyDeviceToHost, (hipStream_t)stream);
    if (e == hipSuccess) e = hipStreamSynchronize((hipStream_t)stream);
    if (e != hipSuccess) return -(int)e - 1000;
    *err_out = v;
#endif
    return 0;
}
