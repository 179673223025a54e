// Flash attention for the denoiser's non-causal self-attention (common/attn.py:82,
// F.scaled_dot_product_attention) — forward, and backward as two kernels (dK/dV owned by
// key tiles, dQ owned by query tiles; no atomics, deterministic).
//
// Layout: q/k/v/o are frame-major rows with heads as 64-feature column groups:
//   element (b, h, l, d) at base[(b*L + l)*ld + h*HD + d].
//
// MFMA formulation (16x16 tiles, 32-deep K slabs, od_frag<T>): every product is arranged
// so that the softmax row index is the accumulator COLUMN (lane&15).  Then the running max,
// the running sum, LSE and delta are lane-local scalars, and a C-layout score tile is
// directly the B (or A) operand of the next MFMA with the key permutation
//     k-slot j of lane-group g  <->  index 32u + 16*(j>>2) + 4g + (j&3),
// which the other operand reproduces by reading a TRANSPOSED LDS tile with two half-width
// reads.  fwd:  S^T = K Q^T ;  O^T += V^T P^T.
#include "od_common.h"
#include "od_tiles.h"
#include "od_api_internal.h"
#include <type_traits>

namespace {

constexpr float LOG2E = 1.4426950408889634f;
constexpr float LN2 = 0.6931471805599453f;
constexpr float NEG_BIG = -1.0e30f;

// ---- staging a [64 rows][HD] global tile through registers -------------------------------
template <class T, int HD>
struct Stage {
    static constexpr int ROWB = HD * (int)sizeof(T);        // row-major tile row bytes
    static constexpr int TROWB = 64 * (int)sizeof(T);       // transposed tile row bytes
    static constexpr int CPR = ROWB / 16;                   // 16-byte chunks per row
    static constexpr int NCH = 64 * CPR / 256;              // chunks per thread
    static constexpr int EPC = 16 / (int)sizeof(T);         // elements per chunk
    static constexpr int BYTES = 64 * ROWB;                 // == HD * TROWB
    static constexpr bool TR = sizeof(T) == 2;              // bf16: transpose at read time (no T tile)
    static constexpr int NT = TR ? 0 : 1;                   // transposed tiles kept per operand
    u32x4 r[NCH];

    // rows row0..row0+63 of a (rows x HD) strip; rows >= nrows are clamped (caller masks)
    __device__ __forceinline__ void load(const T* base, int ld, int row0, int nrows) {
#pragma unroll
        for (int i = 0; i < NCH; i++) {
            const int c = threadIdx.x + 256 * i, row = c / CPR, ch = c % CPR;
            int gr = row0 + row; gr = gr < nrows ? gr : nrows - 1;
            r[i] = *(const u32x4*)(base + (size_t)gr * ld + ch * EPC);
        }
    }
    __device__ __forceinline__ void store_rowmajor(unsigned char* t) const {
#pragma unroll
        for (int i = 0; i < NCH; i++) {
            const int c = threadIdx.x + 256 * i, row = c / CPR, ch = c % CPR;
            *(u32x4*)(t + tile_off<ROWB>(row, ch * 16)) = r[i];
        }
    }
    // LDS-DMA of the same row-major swizzled image (no VGPRs, no ds_write): the block's 4 waves each
    // stream 1 KiB pieces = (1024 / ROWB) tile rows; the swizzle is applied to the source column.
    template <int NW = 4>
    static __device__ __forceinline__ void dma_rowmajor(const T* base, int ld, int row0, int nrows, unsigned char* t) {
        constexpr int RPP = 1024 / ROWB;                   // rows per piece
        constexpr int PIECES = BYTES / 1024;
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
        for (int i = 0; i < (PIECES + NW - 1) / NW; i++) {
            const int piece = wave + NW * i;
            if (PIECES % NW != 0 && piece >= PIECES) break;
            const int row = piece * RPP + lane / CPR;
            const int slot = (lane % CPR) ^ (row & (CPR - 1));
            int gr = row0 + row; gr = gr < nrows ? gr : nrows - 1;
            od_glds16(base + (size_t)gr * ld + slot * EPC, t + piece * 1024 + lane * 16);
        }
    }
    // element (row, col) -> transposed tile row `col`, position `row`
    __device__ __forceinline__ void store_transposed(unsigned char* t) const {
#pragma unroll
        for (int i = 0; i < NCH; i++) {
            const int c = threadIdx.x + 256 * i, row = c / CPR, ch = c % CPR;
            const T* p = (const T*)&r[i];
#pragma unroll
            for (int e = 0; e < EPC; e++) *(T*)(t + tile_off<TROWB>(ch * EPC + e, row * (int)sizeof(T))) = p[e];
        }
    }
};

// XCD-aware (b,h)/tile order: all tiles of one (b,h) run on one XCD so K/V (or Q/dO) stay in its L2
__device__ __forceinline__ bool attn_block_coords(int ntiles, int BH, int& tile, int& bh) {
    const int id = blockIdx.x, xcd = id & 7, slot = id >> 3;
    bh = (slot / ntiles) * 8 + xcd;
    tile = slot % ntiles;
    return bh < BH;
}
inline int attn_grid(int ntiles, int BH) { return ((BH + 7) / 8) * 8 * ntiles; }

// store 4 consecutive features
__device__ __forceinline__ void st4(bf16_t* p, float a, float b, float c, float d) {
    u32x2 v;
    v[0] = od_pack_bf2(a, b);
    v[1] = od_pack_bf2(c, d);
    *(u32x2*)p = v;
}
__device__ __forceinline__ void st4(float* p, float a, float b, float c, float d) {
    f32x4 v; v[0] = a; v[1] = b; v[2] = c; v[3] = d;
    *(f32x4*)p = v;
}
__device__ __forceinline__ void st4(f32x3_t* p, float a, float b, float c, float d) { st4((float*)p, a, b, c, d); }

// ======================================================================== forward
// block: 4 waves x NQT 16-query tiles (32 or 16 query rows per wave); loop over 64-key tiles, double-buffered in LDS.
// NQT = 1 halves a wave's work and its registers: at sampler sizes (B*H*L/32 = 2240 waves of work against 2048 wave slots at two waves per
// SIMD) the 32-query form runs two rounds for 1.09 rounds of work.
// PRE: q arrives multiplied by scale*log2(e) (od_qk_norm_rope's q_scale), so q.k is already the base-2 exponent and
// the per-element multiply disappears from the loop.
template <class T, int HD, int NW, bool PRE, int NQT = 2>
__global__ __launch_bounds__(64 * NW, (NW >= 6 || NQT == 1 ? 3 : 2)) void flash_fwd_kernel(const T* __restrict__ q, int ldq, const T* __restrict__ k, int ldk,
                                                           const T* __restrict__ v, int ldv, T* __restrict__ o, int ldo,
                                                           float* __restrict__ lse, int B, int H, int L, float scale) {
    using St = Stage<T, HD>;
    constexpr int NS = HD / 32;   // 32-deep slabs over the head dim
    constexpr int ND = HD / 16;   // 16-row tiles over the head dim
    OD_DYN_SMEM(smem);   // 2 stages x (K row-major, V row-major [bf16] or V^T [f32])
    constexpr int QB = NW * 16 * NQT;
    const int nqt = (L + QB - 1) / QB;
    int qt, bh;
    if (!attn_block_coords(nqt, B * H, qt, bh)) return;
    const int b = bh / H, h = bh % H;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, x = lane & 15, g = lane >> 4;
    const T* qb = q + (size_t)b * L * ldq + h * HD;
    const T* kb = k + (size_t)b * L * ldk + h * HD;
    const T* vb = v + (size_t)b * L * ldv + h * HD;
    const int q0 = qt * QB + wave * 16 * NQT;
    const float c = PRE ? 1.f : scale * LOG2E;

    od_frag<T> fq[NQT][NS];
#pragma unroll
    for (int qi = 0; qi < NQT; qi++) {
        int row = q0 + qi * 16 + x; row = row < L ? row : L - 1;
#pragma unroll
        for (int s = 0; s < NS; s++) od_frag_load(fq[qi][s], qb + (size_t)row * ldq + s * 32 + g * 8);
    }
    f32x4 oacc[NQT][ND];
    // Softmax against a LAZY reference (VALU diet: the loop is bound by the vector ALU, not the MFMA — 32 exps at
    // quarter rate per lane per tile already cost as many cycles as the 32 MFMAs).  mref[qi] is a per-query
    // reference score, set to the row max of the first key tile and raised only when a later tile exceeds it by
    // more than 2^OD_FWD_SLACK; the score accumulators START at -mref (the MFMA's C operand), so p = 2^(c * acc)
    // takes one packed multiply per pair — no fma, no cross-lane max, no rescale of O on the common path.
    // p <= 2^OD_FWD_SLACK keeps fp32 sums and bf16 P far from overflow; the result is the same softmax
    // (shift invariance), lse = mref * scale + ln(l).
    float mref[NQT], lrun[NQT];
#pragma unroll
    for (int qi = 0; qi < NQT; qi++) {
        mref[qi] = 0.f; lrun[qi] = 0.f;
#pragma unroll
        for (int dt = 0; dt < ND; dt++) oacc[qi][dt] = (f32x4)(0.f);
    }
#ifndef OD_FWD_SLACK
#define OD_FWD_SLACK 8.0f
#endif
    const float inv_c = 1.0f / c;

    const int nkt = (L + 63) / 64;
    St sk, sv;
    auto lstore = [&](unsigned char* base) {
        sk.store_rowmajor(base);
        if constexpr (St::TR) sv.store_rowmajor(base + St::BYTES); else sv.store_transposed(base + St::BYTES);
    };
    // bf16: K/V tiles go global -> LDS by LDS-DMA; f32 (V must be transposed) stages through registers
    auto dma = [&](int kt, unsigned char* base) {
        St::template dma_rowmajor<NW>(kb, ldk, kt * 64, L, base);
        St::template dma_rowmajor<NW>(vb, ldv, kt * 64, L, base + St::BYTES);
    };
    if constexpr (St::TR) dma(0, smem);
    else { sk.load(kb, ldk, 0, L); sv.load(vb, ldv, 0, L); lstore(smem); }
    __syncthreads();
    auto tile = [&](int kt, auto masked_t, auto first_t) {
        constexpr bool MASKED = decltype(masked_t)::value;
        constexpr bool FIRST = decltype(first_t)::value;
        const unsigned char* tK = smem + (kt & 1) * 2 * St::BYTES;
        const unsigned char* tV = tK + St::BYTES;
        if (kt + 1 < nkt) {
            if constexpr (St::TR) dma(kt + 1, smem + ((kt + 1) & 1) * 2 * St::BYTES);
            else { sk.load(kb, ldk, (kt + 1) * 64, L); sv.load(vb, ldv, (kt + 1) * 64, L); }
        }

        // S^T tiles: rows = keys (4 tiles of 16), cols = queries; accumulators start at -mref
        f32x4 e[NQT][4];
#pragma unroll
        for (int t4 = 0; t4 < 4; t4++) {
            od_frag<T> fk[NS];
#pragma unroll
            for (int s = 0; s < NS; s++) frag_contig<St::ROWB>(fk[s], tK, t4 * 16 + x, s * 32 + g * 8);
#pragma unroll
            for (int qi = 0; qi < NQT; qi++) {
                f32x4 a = (f32x4)(-mref[qi]);
#pragma unroll
                for (int s = 0; s < NS; s++) a = od_mma(fk[s], fq[qi][s], a);
                if constexpr (PRE) e[qi][t4] = a; else e[qi][t4] = a * c;     // log2 units relative to the reference
            }
        }
        const int kbase = kt * 64;
        if constexpr (MASKED) {    // ragged last tile only: mask keys >= L
#pragma unroll
            for (int qi = 0; qi < NQT; qi++)
#pragma unroll
                for (int t4 = 0; t4 < 4; t4++)
#pragma unroll
                    for (int r = 0; r < 4; r++)
                        if (kbase + t4 * 16 + 4 * g + r >= L) e[qi][t4][r] = NEG_BIG;
        }
        // lane-local maxima (lane owns query column x, keys 16*t4 + 4g + r)
        float mx[NQT];
#pragma unroll
        for (int qi = 0; qi < NQT; qi++) {
            mx[qi] = fmaxf(fmaxf(e[qi][0][0], e[qi][0][1]), fmaxf(e[qi][0][2], e[qi][0][3]));
#pragma unroll
            for (int t4 = 1; t4 < 4; t4++)
                mx[qi] = fmaxf(mx[qi], fmaxf(fmaxf(e[qi][t4][0], e[qi][t4][1]), fmaxf(e[qi][t4][2], e[qi][t4][3])));
        }
        // rare path (always on the first tile): move the reference.  Wave-uniform branch.
        if (FIRST || __any(fmaxf(mx[0], mx[NQT - 1]) > OD_FWD_SLACK)) {
#pragma unroll
            for (int qi = 0; qi < NQT; qi++) {
                float m = mx[qi];
                m = fmaxf(m, __shfl_xor(m, 16));
                m = fmaxf(m, __shfl_xor(m, 32));
                const float d = FIRST ? m : fmaxf(m, 0.f);
                if constexpr (!FIRST) {
                    const float alpha = od_exp2(-d);
                    lrun[qi] *= alpha;
#pragma unroll
                    for (int dt = 0; dt < ND; dt++) oacc[qi][dt] *= alpha;
                }
                mref[qi] += d * inv_c;
#pragma unroll
                for (int t4 = 0; t4 < 4; t4++) e[qi][t4] -= d;
            }
        }
        od_frag<T> fp[NQT][2];
#pragma unroll
        for (int qi = 0; qi < NQT; qi++) {
            f32x4 ps = (f32x4)(0.f);
#pragma unroll
            for (int t4 = 0; t4 < 4; t4++) {
                f32x4 p;
#pragma unroll
                for (int r = 0; r < 4; r++) p[r] = od_exp2(e[qi][t4][r]);
                ps += p;
                od_frag_set4(fp[qi][t4 >> 1], t4 & 1, p[0], p[1], p[2], p[3]);
            }
            lrun[qi] += (ps[0] + ps[1]) + (ps[2] + ps[3]);
        }
        // O^T += V^T P^T
#pragma unroll
        for (int dt = 0; dt < ND; dt++)
#pragma unroll
            for (int u = 0; u < 2; u++) {
                od_frag<T> fv;
                frag_cols<St::ROWB, St::TROWB>(fv, tV, tV, dt * 16, x, u, g);
#pragma unroll
                for (int qi = 0; qi < NQT; qi++) oacc[qi][dt] = od_mma(fv, fp[qi][u], oacc[qi][dt]);
            }
        if constexpr (!St::TR) { if (kt + 1 < nkt) lstore(smem + ((kt + 1) & 1) * 2 * St::BYTES); }
        __syncthreads();
    };
    const int nfull = L / 64;
    if (nfull > 0) tile(0, std::false_type{}, std::true_type{});
    else tile(0, std::true_type{}, std::true_type{});
    for (int kt = 1; kt < nfull; kt++) tile(kt, std::false_type{}, std::false_type{});
    if (nfull > 0 && nfull < nkt) tile(nfull, std::true_type{}, std::false_type{});
#pragma unroll
    for (int qi = 0; qi < NQT; qi++) {
        float l = lrun[qi];
        l += __shfl_xor(l, 16);
        l += __shfl_xor(l, 32);
        const float inv = 1.f / l;
        const int row = q0 + qi * 16 + x;
        if (row < L) {
            T* orow = o + ((size_t)b * L + row) * ldo + h * HD;
#pragma unroll
            for (int dt = 0; dt < ND; dt++)
                st4(orow + dt * 16 + 4 * g, oacc[qi][dt][0] * inv, oacc[qi][dt][1] * inv, oacc[qi][dt][2] * inv, oacc[qi][dt][3] * inv);
            if (g == 0) lse[((size_t)b * H + h) * L + row] = (mref[qi] * c + log2f(l)) * LN2;
        }
    }
}

// ======================================================================== forward, fp32 tensors, 3 x bf16 MFMAs per product, head_dim 64
// The sampler's compliant mode (f32_matmul = "bf16x3"; round 4).  Same formulation as flash_fwd_kernel, but the (hi, lo) bf16 split of K and V
// happens ONCE per tile, when the workgroup stages it: a thread turns its 16-byte fp32 chunk into two 8-byte bf16 chunks and stores them into a
// hi and a lo plane of the LDS stage (row-major bf16 images, 16 KB per operand like the fp32 tile they replace).  The generic kernel staged fp32
// (plus a transposed fp32 copy of V, four 4-byte stores per chunk) and split every fragment at read time — in each of the four waves, ~450
// vector-ALU operations per wave and key tile beside 96 MFMAs: the kernel was bound by that (119 us per launch at the sampler's size, 0.51 PF/s
// executed).  Now K fragments are 16-byte reads of the planes, V^T fragments transpose reads of the row-major planes (no transposed copy), the
// probabilities are split as before (they are formed in registers), and the staging itself is the only split work left: a quarter of it.
template <int NW, bool PRE, int NQT>
__global__ __launch_bounds__(64 * NW, 2) void flash_fwd_x3p_kernel(const float* __restrict__ q, int ldq, const float* __restrict__ k, int ldk,
                                                                   const float* __restrict__ v, int ldv, float* __restrict__ o, int ldo,
                                                                   float* __restrict__ lse, int B, int H, int L, float scale) {
    constexpr int HD = 64, PL = 64 * 128;          // one bf16 plane of a 64-row tile
    constexpr int STAGE = 4 * PL;                  // K hi, K lo, V hi, V lo
    static_assert(NW == 4, "the staging maps 1024 chunks of a tile onto 256 threads");
    OD_DYN_SMEM(smem);
    constexpr int QB = NW * 16 * NQT;
    const int nqt = (L + QB - 1) / QB;
    int qt, bh;
    if (!attn_block_coords(nqt, B * H, qt, bh)) return;
    const int b = bh / H, h = bh % H;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, x = lane & 15, g = lane >> 4;
    const float* qb = q + (size_t)b * L * ldq + h * HD;
    const float* kb = k + (size_t)b * L * ldk + h * HD;
    const float* vb = v + (size_t)b * L * ldv + h * HD;
    const int q0 = qt * QB + wave * 16 * NQT;
    const float c = PRE ? 1.f : scale * LOG2E;

    od_frag<f32x3_t> fq[NQT][2];
#pragma unroll
    for (int qi = 0; qi < NQT; qi++) {
        int row = q0 + qi * 16 + x; row = row < L ? row : L - 1;
#pragma unroll
        for (int s = 0; s < 2; s++) od_frag_load(fq[qi][s], (const f32x3_t*)(qb + (size_t)row * ldq + s * 32 + g * 8));
    }
    f32x4 oacc[NQT][4];
    float mref[NQT], lrun[NQT];
#pragma unroll
    for (int qi = 0; qi < NQT; qi++) {
        mref[qi] = 0.f; lrun[qi] = 0.f;
#pragma unroll
        for (int dt = 0; dt < 4; dt++) oacc[qi][dt] = (f32x4)(0.f);
    }
    const float inv_c = 1.0f / c;
    const int nkt = (L + 63) / 64;
    // staging: chunk cc = threadIdx.x + 256 i (i = 0..3) of the 64 x 16 fp32 chunks of a tile: row cc / 16, floats 4 (cc % 16) .. + 3
    u32x4 rk[4], rv[4];
    auto gload = [&](int kt) {
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int cc = threadIdx.x + 256 * i, row = cc >> 4, ch = cc & 15;
            int gr = kt * 64 + row; gr = gr < L ? gr : L - 1;
            rk[i] = *(const u32x4*)(kb + (size_t)gr * ldk + ch * 4);
            rv[i] = *(const u32x4*)(vb + (size_t)gr * ldv + ch * 4);
        }
    };
    auto lstore = [&](unsigned char* st) {
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int cc = threadIdx.x + 256 * i, row = cc >> 4, ch = cc & 15;
            const int off = tile_off<128>(row, ch * 8);
            u32x2 hi, lo;
            const f32x4 fk4 = __builtin_bit_cast(f32x4, rk[i]), fv4 = __builtin_bit_cast(f32x4, rv[i]);
            uint32_t h0, l0, h1, l1;
            od_split2(fk4[0], fk4[1], h0, l0); od_split2(fk4[2], fk4[3], h1, l1);
            hi[0] = h0; hi[1] = h1; lo[0] = l0; lo[1] = l1;
            *(u32x2*)(st + off) = hi; *(u32x2*)(st + PL + off) = lo;
            od_split2(fv4[0], fv4[1], h0, l0); od_split2(fv4[2], fv4[3], h1, l1);
            hi[0] = h0; hi[1] = h1; lo[0] = l0; lo[1] = l1;
            *(u32x2*)(st + 2 * PL + off) = hi; *(u32x2*)(st + 3 * PL + off) = lo;
        }
    };
    gload(0); lstore(smem);
    __syncthreads();
    auto tile = [&](int kt, auto masked_t, auto first_t) {
        constexpr bool MASKED = decltype(masked_t)::value;
        constexpr bool FIRST = decltype(first_t)::value;
        const unsigned char* st = smem + (kt & 1) * STAGE;
        if (kt + 1 < nkt) gload(kt + 1);
        f32x4 e[NQT][4];
#pragma unroll
        for (int t4 = 0; t4 < 4; t4++) {
            od_frag<f32x3_t> fk[2];
#pragma unroll
            for (int s = 0; s < 2; s++) {
                fk[s].hi = *(const s16x8*)(st + tile_off<128>(t4 * 16 + x, (s * 32 + g * 8) * 2));
                fk[s].lo = *(const s16x8*)(st + PL + tile_off<128>(t4 * 16 + x, (s * 32 + g * 8) * 2));
            }
#pragma unroll
            for (int qi = 0; qi < NQT; qi++) {
                f32x4 a = (f32x4)(-mref[qi]);
#pragma unroll
                for (int s = 0; s < 2; s++) a = od_mma(fk[s], fq[qi][s], a);
                if constexpr (PRE) e[qi][t4] = a; else e[qi][t4] = a * c;
            }
        }
        const int kbase = kt * 64;
        if constexpr (MASKED) {
#pragma unroll
            for (int qi = 0; qi < NQT; qi++)
#pragma unroll
                for (int t4 = 0; t4 < 4; t4++)
#pragma unroll
                    for (int r = 0; r < 4; r++)
                        if (kbase + t4 * 16 + 4 * g + r >= L) e[qi][t4][r] = NEG_BIG;
        }
        float mx[NQT];
#pragma unroll
        for (int qi = 0; qi < NQT; qi++) {
            mx[qi] = fmaxf(fmaxf(e[qi][0][0], e[qi][0][1]), fmaxf(e[qi][0][2], e[qi][0][3]));
#pragma unroll
            for (int t4 = 1; t4 < 4; t4++)
                mx[qi] = fmaxf(mx[qi], fmaxf(fmaxf(e[qi][t4][0], e[qi][t4][1]), fmaxf(e[qi][t4][2], e[qi][t4][3])));
        }
        if (FIRST || __any(fmaxf(mx[0], mx[NQT - 1]) > OD_FWD_SLACK)) {      // the lazy reference moves (always on the first tile)
#pragma unroll
            for (int qi = 0; qi < NQT; qi++) {
                float m = mx[qi];
                m = fmaxf(m, __shfl_xor(m, 16));
                m = fmaxf(m, __shfl_xor(m, 32));
                const float d = FIRST ? m : fmaxf(m, 0.f);
                if constexpr (!FIRST) {
                    const float alpha = od_exp2(-d);
                    lrun[qi] *= alpha;
#pragma unroll
                    for (int dt = 0; dt < 4; dt++) oacc[qi][dt] *= alpha;
                }
                mref[qi] += d * inv_c;
#pragma unroll
                for (int t4 = 0; t4 < 4; t4++) e[qi][t4] -= d;
            }
        }
        od_frag<f32x3_t> fp[NQT][2];
#pragma unroll
        for (int qi = 0; qi < NQT; qi++) {
            f32x4 ps = (f32x4)(0.f);
#pragma unroll
            for (int t4 = 0; t4 < 4; t4++) {
                f32x4 p;
#pragma unroll
                for (int r = 0; r < 4; r++) p[r] = od_exp2(e[qi][t4][r]);
                ps += p;
                od_frag_set4(fp[qi][t4 >> 1], t4 & 1, p[0], p[1], p[2], p[3]);
            }
            lrun[qi] += (ps[0] + ps[1]) + (ps[2] + ps[3]);
        }
        // O^T += V^T P^T : V^T fragments by transpose reads of the row-major hi / lo planes
#pragma unroll
        for (int dt = 0; dt < 4; dt++)
#pragma unroll
            for (int u = 0; u < 2; u++) {
                od_frag<bf16_t> vh, vl;
                frag_cols<128, 128>(vh, st + 2 * PL, st + 2 * PL, dt * 16, x, u, g);
                frag_cols<128, 128>(vl, st + 3 * PL, st + 3 * PL, dt * 16, x, u, g);
                od_frag<f32x3_t> fv;
                fv.hi = vh.v; fv.lo = vl.v;
#pragma unroll
                for (int qi = 0; qi < NQT; qi++) oacc[qi][dt] = od_mma(fv, fp[qi][u], oacc[qi][dt]);
            }
        if (kt + 1 < nkt) lstore(smem + ((kt + 1) & 1) * STAGE);
        __syncthreads();
    };
    const int nfull = L / 64;
    if (nfull > 0) tile(0, std::false_type{}, std::true_type{});
    else tile(0, std::true_type{}, std::true_type{});
    for (int kt = 1; kt < nfull; kt++) tile(kt, std::false_type{}, std::false_type{});
    if (nfull > 0 && nfull < nkt) tile(nfull, std::true_type{}, std::false_type{});
#pragma unroll
    for (int qi = 0; qi < NQT; qi++) {
        float l = lrun[qi];
        l += __shfl_xor(l, 16);
        l += __shfl_xor(l, 32);
        const float inv = 1.f / l;
        const int row = q0 + qi * 16 + x;
        if (row < L) {
            float* orow = o + ((size_t)b * L + row) * ldo + h * HD;
#pragma unroll
            for (int dt = 0; dt < 4; dt++)
                st4(orow + dt * 16 + 4 * g, oacc[qi][dt][0] * inv, oacc[qi][dt][1] * inv, oacc[qi][dt][2] * inv, oacc[qi][dt][3] * inv);
            if (g == 0) lse[((size_t)b * H + h) * L + row] = (mref[qi] * c + log2f(l)) * LN2;
        }
    }
}

// ======================================================================== forward, bf16, head_dim 64, 32x32x16 MFMA
// Round-2 rewrite of the bf16 hot path, driven by the issue-model probe (tools/probes/issue_model.hip ->
// profiles/r02a_issue_model.txt): a gfx950 SIMD issues ONE instruction per ~4.2-4.5 cycles whatever its kind and however
// many waves it holds (v_exp_f32 ~8.5, an MFMA ~9.5 of issue while its pipe runs 16 / 32 cycles in the background), so a
// softmax-carrying loop is bound by its instruction COUNT.  v_mfma_f32_32x32x16_bf16 does the work of two 16x16x32 for one
// issue slot: 16 MFMAs per (32 queries x 64 keys) instead of 32.  Same formulation as above (S^T = K Q^T with the softmax
// row as the accumulator COLUMN, P^T accumulators feeding O^T += V^T P^T through a key permutation that the V operand
// reproduces with transpose reads), on 32-wide tiles:
//   accumulator element r of lane (c = lane & 31, hi = lane >> 5): row (r & 3) + 8 (r >> 2) + 4 hi, column c
//   k-slot 8 hi + j of key slab sl (16 keys) of key block kb (32 keys)  <->  key 32 kb + 16 sl + 8 (j >> 2) + 4 hi + (j & 3)
// The running max is gone from the common path: p = 2^(s - ref) against the lazy per-query reference, and the lane-local
// partial row sum (needed anyway) doubles as the overflow guard — only if a partial sum leaves [0, 2^OD_FWD32_GUARD) does the
// wave take the exact path (true row max, reference raised, O and l rescaled, p recomputed).
typedef float f32x16_t __attribute__((ext_vector_type(16)));
// TA = operand type tag: bf16_t, or f16_t (IEEE half operands: "attention in fp16", v_mfma_f32_32x32x16_f16)
template <class TA> __device__ __forceinline__ f32x16_t od_mma32(s16x8 a, s16x8 b, f32x16_t c);
template <> __device__ __forceinline__ f32x16_t od_mma32<bf16_t>(s16x8 a, s16x8 b, f32x16_t c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
template <> __device__ __forceinline__ f32x16_t od_mma32<f16_t>(s16x8 a, s16x8 b, f32x16_t c) {
#if defined(OD_EMU)
    return emu::mfma_32x32x16_f16(a, b, c);
#else
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(od_h8_t, a), __builtin_bit_cast(od_h8_t, b), c, 0, 0, 0);
#endif
}
#ifndef OD_FWD32_GUARD
#define OD_FWD32_GUARD 16384.0f
#endif

#ifndef OD_FWD32_OCC2
#define OD_FWD32_OCC2 2     // waves per SIMD asked of the register allocator at 64 queries per wave
#endif
template <int NW, int NQB, bool PRE, class TA = bf16_t>
__global__ __launch_bounds__(64 * NW, (NQB == 1 ? 2 : OD_FWD32_OCC2)) void flash_fwd32_kernel(const bf16_t* __restrict__ q, int ldq, const bf16_t* __restrict__ k, int ldk,
                                                                 const bf16_t* __restrict__ v, int ldv, bf16_t* __restrict__ o, int ldo,
                                                                 float* __restrict__ lse, int B, int H, int L, float scale) {
    using St = Stage<bf16_t, 64>;
    constexpr int HD = 64, QB = NW * NQB * 32, STAGE = 2 * St::BYTES;
    static_assert(NW == 4 || NW == 8, "the K/V tiles are streamed as 2 + 2 (four waves) or 1 + 1 (eight waves) one-KiB pieces per wave");
    OD_DYN_SMEM(smem);   // 2 stages x (K row-major, V row-major), 16-byte slots XOR-swizzled by row
    const int nqt = (L + QB - 1) / QB;
    int qt, bh;
    if (!attn_block_coords(nqt, B * H, qt, bh)) return;
    const int b = bh / H, h = bh % H;
    const int lane = threadIdx.x & 63, wave = od_uniform(threadIdx.x >> 6), c32 = lane & 31, hi = lane >> 5, x = lane & 15, g4 = lane >> 4;
    const bf16_t* qb_ = q + (size_t)b * L * ldq + h * HD;
    const int q0 = qt * QB + wave * NQB * 32;
    const float c = PRE ? 1.f : scale * LOG2E, inv_c = 1.0f / c;

    // K / V tiles by buffer-addressed LDS-DMA: per-lane offset fixed for the whole kernel, one scalar add per tile;
    // rows >= L lie past the end of the descriptor and read as zero
    const od_srd_t rk = od_make_srd(k + (size_t)b * L * ldk + h * HD, (unsigned)(((size_t)(L - 1) * ldk + HD) * 2));
    const od_srd_t rv = od_make_srd(v + (size_t)b * L * ldv + h * HD, (unsigned)(((size_t)(L - 1) * ldv + HD) * 2));
    // piece = 8 rows x 128 B; the swizzle (swz32, od_tiles.h) is applied on the SOURCE column; a wave's two pieces (w, w + 4)
    // have the same row bits 3..0, so one per-lane offset serves both
    // (eight waves: wave w moves piece w of each tile alone; the per-lane offsets are the same expression with the piece index w)
    const int prow = lane >> 3, pslot = (lane & 7) ^ swz32(wave * 8 + prow);
    const unsigned vk = (unsigned)((wave * 8 + prow) * ldk * 2 + pslot * 16), vv = (unsigned)((wave * 8 + prow) * ldv * 2 + pslot * 16);
    auto dma = [&](int kt, unsigned char* st) {
        const unsigned sk = (unsigned)kt * 64u * (unsigned)ldk * 2u, sv = (unsigned)kt * 64u * (unsigned)ldv * 2u;
        od_buffer_lds16(rk, vk, sk, st + wave * 1024);
        if (NW == 4) od_buffer_lds16(rk, vk, sk + 32u * (unsigned)ldk * 2u, st + (wave + 4) * 1024);
        od_buffer_lds16(rv, vv, sv, st + St::BYTES + wave * 1024);
        if (NW == 4) od_buffer_lds16(rv, vv, sv + 32u * (unsigned)ldv * 2u, st + St::BYTES + (wave + 4) * 1024);
    };

    // Q^T fragments (B operand): column = query, k = 16 s + 8 hi + j
    s16x8 fq[NQB][4];
#pragma unroll
    for (int qi = 0; qi < NQB; qi++) {
        int row = q0 + qi * 32 + c32; row = row < L ? row : L - 1;
#pragma unroll
        for (int s4 = 0; s4 < 4; s4++) fq[qi][s4] = *(const s16x8*)(qb_ + (size_t)row * ldq + s4 * 16 + hi * 8);
    }
    f32x16_t oacc[NQB][2], minit[NQB];      // minit = splat(-reference): the score MFMAs' C operand, touched on the exact path only
    float mref[NQB], lrun[NQB];
#pragma unroll
    for (int qi = 0; qi < NQB; qi++) {
        mref[qi] = 0.f; lrun[qi] = 0.f;
        minit[qi] = (f32x16_t)(0.f);
        oacc[qi][0] = (f32x16_t)(0.f); oacc[qi][1] = (f32x16_t)(0.f);
    }
    const int nkt = (L + 63) / 64;
    // lane-constant LDS byte offsets inside a stage: K fragment (key row c32 (+32), 16-byte slot 2 s + hi) and V transpose-read
    // chunk (rows 4 (g4 >> 1) + (x >> 2) (+8, +16, ...) of the tile, feature column 16 (g4 & 1) + 4 (x & 3) (+32))
    int offK[4], offV[2][2];
#pragma unroll
    for (int s4 = 0; s4 < 4; s4++) offK[s4] = tile32_off(c32, (s4 * 16 + hi * 8) * 2);                // + kb * 32 rows = + 4096 B
#pragma unroll
    for (int db = 0; db < 2; db++)
#pragma unroll
        for (int e = 0; e < 2; e++)                                                                  // + 16 rows per slab = + 2048 B
            offV[db][e] = St::BYTES + tile32_off(4 * (g4 >> 1) + (x >> 2) + 8 * e, (db * 32 + 16 * (g4 & 1) + 4 * (x & 3)) * 2);
    // The Q fragments above are compiler-visible loads: drain them with a wait hipcc SEES.  Otherwise it waits for them at their
    // first use INSIDE the loop (vmcnt(3) .. vmcnt(0) before the first score MFMAs of every other tile), and since the tile
    // DMAs below are asm it cannot count, those waits also drained the four pieces of the NEXT tile just issued.
    OD_DRAIN_VMEM();
    dma(0, smem);
    OD_WAIT_VMCNT(0);
    __syncthreads();

    auto tile = [&](int kt, const unsigned char* st, unsigned char* st_next, auto masked_t, auto first_t) {
        constexpr bool MASKED = decltype(masked_t)::value;
        constexpr bool FIRST = decltype(first_t)::value;
        if (kt + 1 < nkt) dma(kt + 1, st_next);

        // One query block at a time between the K fragments (read once per tile, shared by the wave's query blocks) and the P fragments
        // (kept for all of them: every V fragment is read once and applied to each): only one block's 32 score registers are live.
        s16x8 fk[2][4];
#pragma unroll
        for (int kb = 0; kb < 2; kb++)
#pragma unroll
            for (int s4 = 0; s4 < 4; s4++) fk[kb][s4] = *(const s16x8*)(st + offK[s4] + kb * 4096);
        s16x8 fp[NQB][2][2];
#pragma unroll
        for (int qi = 0; qi < NQB; qi++) {
            f32x16_t sa[2];
            // S^T blocks (rows = keys); the accumulators start at -reference
#pragma unroll
            for (int kb = 0; kb < 2; kb++) {
                sa[kb] = od_mma32<TA>(fk[kb][0], fq[qi][0], minit[qi]);
#pragma unroll
                for (int s4 = 1; s4 < 4; s4++) sa[kb] = od_mma32<TA>(fk[kb][s4], fq[qi][s4], sa[kb]);
                if constexpr (!PRE) sa[kb] = sa[kb] * c + minit[qi] * (1.f - c);    // (q.k) c - reference
                if constexpr (MASKED) {    // ragged last tile only: keys >= L
#pragma unroll
                    for (int r = 0; r < 16; r++)
                        if (kt * 64 + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi >= L) sa[kb][r] = NEG_BIG;
                }
            }
            // exact path: move the reference to the row maximum (always on the first tile; otherwise only when the guard trips)
            auto exact = [&]() {
                float m = NEG_BIG;
#pragma unroll
                for (int kb = 0; kb < 2; kb++)
#pragma unroll
                    for (int r = 0; r < 16; r++) m = fmaxf(m, sa[kb][r]);
                m = fmaxf(m, __shfl_xor(m, 32));               // the two half-waves hold complementary keys of the same query
                const float d = FIRST ? m : fmaxf(m, 0.f);     // scores are relative to the current reference
                if constexpr (!FIRST) {
                    const float alpha = od_exp2(-d);
                    lrun[qi] *= alpha;
                    oacc[qi][0] *= alpha; oacc[qi][1] *= alpha;
                }
                mref[qi] += d;                                 // log2 units
                minit[qi] = (f32x16_t)(-mref[qi]);
#pragma unroll
                for (int kb = 0; kb < 2; kb++) sa[kb] -= d;
            };
            float ps;
            auto probs = [&]() {
                float acc0 = 0.f, acc1 = 0.f;
#pragma unroll
                for (int kb = 0; kb < 2; kb++)
#pragma unroll
                    for (int sl = 0; sl < 2; sl++) {
                        u32x4 w;
#pragma unroll
                        for (int jj = 0; jj < 4; jj++) {
                            const float p0 = od_exp2(sa[kb][8 * sl + 2 * jj]), p1 = od_exp2(sa[kb][8 * sl + 2 * jj + 1]);
                            acc0 += p0; acc1 += p1;
                            w[jj] = od_pack2<TA>(p0, p1);
                        }
                        fp[qi][kb][sl] = __builtin_bit_cast(s16x8, w);
                    }
                ps = acc0 + acc1;
            };
            if constexpr (FIRST) { exact(); probs(); }
            else {
                probs();
                if (__any(!(ps < OD_FWD32_GUARD))) { exact(); probs(); }       // wave-uniform, rare
            }
            lrun[qi] += ps;
        }
        // O^T += V^T P^T : A = V^T fragment (rows = features of block db), two transpose reads per 16-key slab
#pragma unroll
        for (int db = 0; db < 2; db++)
#pragma unroll
            for (int kb = 0; kb < 2; kb++)
#pragma unroll
                for (int sl = 0; sl < 2; sl++) {
                    const s16x4 a0 = od_lds_tr_read((const bf16_t*)(st + offV[db][0] + (kb * 2 + sl) * 2048));
                    const s16x4 a1 = od_lds_tr_read((const bf16_t*)(st + offV[db][1] + (kb * 2 + sl) * 2048));
                    s16x8 fv;
                    fv[0] = a0[0]; fv[1] = a0[1]; fv[2] = a0[2]; fv[3] = a0[3];
                    fv[4] = a1[0]; fv[5] = a1[1]; fv[6] = a1[2]; fv[7] = a1[3];
#pragma unroll
                    for (int qi = 0; qi < NQB; qi++) oacc[qi][db] = od_mma32<TA>(fv, fp[qi][kb][sl], oacc[qi][db]);
                }
        OD_WAIT_VMCNT(0);           // this wave's pieces of the next tile have landed (the asm DMA is invisible to hipcc's waits)
        __syncthreads();
    };
    // the two LDS stages alternate with compile-time addresses: the loop body is unrolled by two
    unsigned char* const s0 = smem;
    unsigned char* const s1 = smem + STAGE;
    const int nfull = L / 64;
    if (nfull > 0) tile(0, s0, s1, std::false_type{}, std::true_type{});
    else tile(0, s0, s1, std::true_type{}, std::true_type{});
    int kt = 1;
    for (; kt + 1 < nfull; kt += 2) {
        tile(kt, s1, s0, std::false_type{}, std::false_type{});
        tile(kt + 1, s0, s1, std::false_type{}, std::false_type{});
    }
    if (kt < nfull) { tile(kt, s1, s0, std::false_type{}, std::false_type{}); kt++; }
    if (nfull > 0 && nfull < nkt) { if (kt & 1) tile(kt, s1, s0, std::true_type{}, std::false_type{}); else tile(kt, s0, s1, std::true_type{}, std::false_type{}); }
#pragma unroll
    for (int qi = 0; qi < NQB; qi++) {
        float l = lrun[qi];
        l += __shfl_xor(l, 32);
        const float inv = 1.f / l;
        const int row = q0 + qi * 32 + c32;
        if (row < L) {
            bf16_t* orow = o + ((size_t)b * L + row) * ldo + h * HD;
#pragma unroll
            for (int db = 0; db < 2; db++)
#pragma unroll
                for (int t4 = 0; t4 < 4; t4++)
                    st4(orow + db * 32 + 8 * t4 + 4 * hi, oacc[qi][db][4 * t4] * inv, oacc[qi][db][4 * t4 + 1] * inv,
                        oacc[qi][db][4 * t4 + 2] * inv, oacc[qi][db][4 * t4 + 3] * inv);
            if (hi == 0) lse[((size_t)b * H + h) * L + row] = (mref[qi] + log2f(l)) * LN2;
        }
    }
}

#ifndef OD_FWD16X
#define OD_FWD16X 1       // bf16 / IEEE-half operands, hd 64, pre-multiplied q: flash_fwd16x_kernel — flash_fwd32_kernel's structure on 16x16x32 tiles (0: flash_fwd32_kernel)
#endif
#ifndef OD_FWD16X_NQT
#define OD_FWD16X_NQT 2   // 16-query tiles per wave (4: a tie, 7.39-7.48 ms; 1: 8.87; 3: 7.85 — profiles/r06h_ab_fwd16x_nqt_sampler.txt)
#endif
#ifndef OD_FWD16X_NW
#define OD_FWD16X_NW 4    // waves per workgroup (8: the K / V tiles staged once for 256 queries; A/B)
#endif
#ifndef OD_FWD16X_MIN_L
#define OD_FWD16X_MIN_L 2048   // sequences from this length on run flash_fwd16x_kernel, shorter ones flash_fwd32_kernel (the emulator build lowers it: both are tested)
#endif
#if OD_FWD16X
// flash_fwd32_kernel's STRUCTURE — buffer-addressed asm LDS-DMA (rows past L read as zero), compile-time stage addresses (loop unrolled by two),
// the lazy log2-domain reference with the running row SUM as the overflow guard (no per-tile row maximum), scalar row sums — on
// v_mfma_f32_16x16x32 tiles in flash_fwd_kernel's formulation (S^T = K Q^T: a softmax row is an accumulator column; P^T feeds O^T += V^T P^T
// through a k-permutation the V transpose reads reproduce).  Per wave and 64-key tile: 32 MFMAs of 16 cycles where the 32x32x16 form issues
// 16 of 32, the same 8 + 16 LDS fragment reads, the same exponentials — and 128 registers (four waves per SIMD; the 32x32 form: 158, three).
// Round 6 ("build the thing, or retire the question with a measurement"): 7.64 against 8.01 ms per call at B = 32 x L = 8192 on
// one box, 311.6 against 314.6 ms per step (profiles/r06g_ab_fwd16x.txt) — the 16x16x32 MFMA draws ~7 % less per FLOP on a board whose power cap
// sets the clock (profiles/r03u_mfma_power.txt), which the issue model of round 2 (twice the MFMA issue slots: -12 %) did not know about.
// bf16 or IEEE-half operands (TA), head_dim 64, q pre-multiplied by scale * log2(e).
template <int NW, int NQT, class TA>
__global__ __launch_bounds__(64 * NW, (NW == 8 ? 1 : 2)) void flash_fwd16x_kernel(const bf16_t* __restrict__ q, int ldq, const bf16_t* __restrict__ k, int ldk,
                                                                  const bf16_t* __restrict__ v, int ldv, bf16_t* __restrict__ o, int ldo,
                                                                  float* __restrict__ lse, int B, int H, int L) {
    using T = TA;                                   // the MFMA operand type: q, k, v hold bf16 or IEEE half (same 16-bit containers); o is bf16
    using St = Stage<bf16_t, 64>;
    constexpr int HD = 64, QB = NW * 16 * NQT, STAGE = 2 * St::BYTES;
    static_assert(NW == 4 || NW == 8, "the K / V tiles are streamed as 2 + 2 (four waves) or 1 + 1 (eight waves) one-KiB pieces per wave");
    OD_DYN_SMEM(smem);
    const int nqt = (L + QB - 1) / QB;
    int qt, bh;
    if (!attn_block_coords(nqt, B * H, qt, bh)) return;
    const int b = bh / H, h = bh % H;
    const int lane = threadIdx.x & 63, wave = od_uniform(threadIdx.x >> 6), x = lane & 15, g = lane >> 4;
    const bf16_t* qb = q + (size_t)b * L * ldq + h * HD;
    const int q0 = qt * QB + wave * 16 * NQT;
    const od_srd_t rk = od_make_srd(k + (size_t)b * L * ldk + h * HD, (unsigned)(((size_t)(L - 1) * ldk + HD) * 2));
    const od_srd_t rv = od_make_srd(v + (size_t)b * L * ldv + h * HD, (unsigned)(((size_t)(L - 1) * ldv + HD) * 2));
    // piece = 8 rows x 128 B, tile_off<128>'s swizzle (slot ^ (row & 7), row & 7 = lane >> 3) applied on the SOURCE column; a wave moves pieces w, w + 4
    const int prow = lane >> 3, pslot = (lane & 7) ^ prow;
    const unsigned vk = (unsigned)((wave * 8 + prow) * ldk * 2 + pslot * 16), vv = (unsigned)((wave * 8 + prow) * ldv * 2 + pslot * 16);
    auto dma = [&](int kt, unsigned char* st) {
        const unsigned sk = (unsigned)kt * 64u * (unsigned)ldk * 2u, sv = (unsigned)kt * 64u * (unsigned)ldv * 2u;
        od_buffer_lds16(rk, vk, sk, st + wave * 1024);
        if (NW == 4) od_buffer_lds16(rk, vk, sk + 32u * (unsigned)ldk * 2u, st + (wave + 4) * 1024);
        od_buffer_lds16(rv, vv, sv, st + St::BYTES + wave * 1024);
        if (NW == 4) od_buffer_lds16(rv, vv, sv + 32u * (unsigned)ldv * 2u, st + St::BYTES + (wave + 4) * 1024);
    };
    od_frag<T> fq[NQT][2];
#pragma unroll
    for (int qi = 0; qi < NQT; qi++) {
        int row = q0 + qi * 16 + x; row = row < L ? row : L - 1;
#pragma unroll
        for (int s2 = 0; s2 < 2; s2++) od_frag_load(fq[qi][s2], (const TA*)(qb + (size_t)row * ldq + s2 * 32 + g * 8));
    }
    f32x4 oacc[NQT][4];
    float mref[NQT], lrun[NQT];                     // mref in log2 units (q is pre-multiplied)
#pragma unroll
    for (int qi = 0; qi < NQT; qi++) {
        mref[qi] = 0.f; lrun[qi] = 0.f;
#pragma unroll
        for (int dt = 0; dt < 4; dt++) oacc[qi][dt] = (f32x4)(0.f);
    }
    const int nkt = (L + 63) / 64;
    OD_DRAIN_VMEM();                                // the Q fragments: a wait hipcc sees (see flash_fwd32_kernel)
    dma(0, smem);
    OD_WAIT_VMCNT(0);
    __syncthreads();
    auto tile = [&](int kt, const unsigned char* st, unsigned char* st_next, auto masked_t, auto first_t) {
        constexpr bool MASKED = decltype(masked_t)::value;
        constexpr bool FIRST = decltype(first_t)::value;
        if (kt + 1 < nkt) dma(kt + 1, st_next);
        const unsigned char* tK = st;
        const unsigned char* tV = st + St::BYTES;
        // S^T tiles: rows = keys (4 tiles of 16), cols = queries; the accumulators start at -reference
        f32x4 e[NQT][4];
#pragma unroll
        for (int t4 = 0; t4 < 4; t4++) {
            od_frag<T> fk[2];
#pragma unroll
            for (int s2 = 0; s2 < 2; s2++) frag_contig<128>(fk[s2], tK, t4 * 16 + x, s2 * 32 + g * 8);
#pragma unroll
            for (int qi = 0; qi < NQT; qi++) {
                f32x4 a = (f32x4)(-mref[qi]);
#pragma unroll
                for (int s2 = 0; s2 < 2; s2++) a = od_mma(fk[s2], fq[qi][s2], a);
                e[qi][t4] = a;
            }
        }
        if constexpr (MASKED) {    // ragged last tile only: keys >= L
#pragma unroll
            for (int qi = 0; qi < NQT; qi++)
#pragma unroll
                for (int t4 = 0; t4 < 4; t4++)
#pragma unroll
                    for (int r = 0; r < 4; r++)
                        if (kt * 64 + t4 * 16 + 4 * g + r >= L) e[qi][t4][r] = NEG_BIG;
        }
        od_frag<T> fp[NQT][2];
#pragma unroll
        for (int qi = 0; qi < NQT; qi++) {
            auto exact = [&]() {                    // move the reference to the row maximum (first tile; later only when the guard trips)
                float m = NEG_BIG;
#pragma unroll
                for (int t4 = 0; t4 < 4; t4++)
#pragma unroll
                    for (int r = 0; r < 4; r++) m = fmaxf(m, e[qi][t4][r]);
                m = fmaxf(m, __shfl_xor(m, 16));
                m = fmaxf(m, __shfl_xor(m, 32));
                const float d = FIRST ? m : fmaxf(m, 0.f);
                if constexpr (!FIRST) {
                    const float alpha = od_exp2(-d);
                    lrun[qi] *= alpha;
#pragma unroll
                    for (int dt = 0; dt < 4; dt++) oacc[qi][dt] *= alpha;
                }
                mref[qi] += d;
#pragma unroll
                for (int t4 = 0; t4 < 4; t4++) e[qi][t4] -= d;
            };
            float ps;
            auto probs = [&]() {
                float acc0 = 0.f, acc1 = 0.f;
#pragma unroll
                for (int t4 = 0; t4 < 4; t4++) {
                    const float p0 = od_exp2(e[qi][t4][0]), p1 = od_exp2(e[qi][t4][1]), p2 = od_exp2(e[qi][t4][2]), p3 = od_exp2(e[qi][t4][3]);
                    acc0 += p0; acc1 += p1; acc0 += p2; acc1 += p3;
                    od_frag_set4(fp[qi][t4 >> 1], t4 & 1, p0, p1, p2, p3);
                }
                ps = acc0 + acc1;
            };
            if constexpr (FIRST) { exact(); probs(); }
            else {
                probs();
                if (__any(!(ps < OD_FWD32_GUARD))) { exact(); probs(); }       // wave-uniform, rare
            }
            lrun[qi] += ps;
        }
        // O^T += V^T P^T
#pragma unroll
        for (int dt = 0; dt < 4; dt++)
#pragma unroll
            for (int u = 0; u < 2; u++) {
                od_frag<T> fv;
                frag_cols<128, 128>(fv, tV, tV, dt * 16, x, u, g);
#pragma unroll
                for (int qi = 0; qi < NQT; qi++) oacc[qi][dt] = od_mma(fv, fp[qi][u], oacc[qi][dt]);
            }
        OD_WAIT_VMCNT(0);
        __syncthreads();
    };
    unsigned char* const s0 = smem;
    unsigned char* const s1 = smem + STAGE;
    const int nfull = L / 64;
    if (nfull > 0) tile(0, s0, s1, std::false_type{}, std::true_type{});
    else tile(0, s0, s1, std::true_type{}, std::true_type{});
    int kt = 1;
    for (; kt + 1 < nfull; kt += 2) {
        tile(kt, s1, s0, std::false_type{}, std::false_type{});
        tile(kt + 1, s0, s1, std::false_type{}, std::false_type{});
    }
    if (kt < nfull) { tile(kt, s1, s0, std::false_type{}, std::false_type{}); kt++; }
    if (nfull > 0 && nfull < nkt) { if (kt & 1) tile(kt, s1, s0, std::true_type{}, std::false_type{}); else tile(kt, s0, s1, std::true_type{}, std::false_type{}); }
#pragma unroll
    for (int qi = 0; qi < NQT; qi++) {
        float l = lrun[qi];
        l += __shfl_xor(l, 16);
        l += __shfl_xor(l, 32);
        const float inv = 1.f / l;
        const int row = q0 + qi * 16 + x;
        if (row < L) {
            bf16_t* orow = o + ((size_t)b * L + row) * ldo + h * HD;
#pragma unroll
            for (int dt = 0; dt < 4; dt++)
                st4(orow + dt * 16 + 4 * g, oacc[qi][dt][0] * inv, oacc[qi][dt][1] * inv, oacc[qi][dt][2] * inv, oacc[qi][dt][3] * inv);
            if (g == 0) lse[((size_t)b * H + h) * L + row] = (mref[qi] + log2f(l)) * LN2;
        }
    }
}
#endif

// (An 8-wave ping-pong arrangement of this kernel — two wave groups alternating MFMA and softmax phases behind bare barriers, K/V in a
// 4-deep LDS ring — measured 8.36 ms against 7.9-8.1 and was removed: profiles/r02g_ab_pingpong.txt.)

// delta[b][h][l] = sum_d dO*O  — one wave per frame row, lanes over (h, d) chunks of 8
template <class T>
__global__ __launch_bounds__(256) void attn_delta_kernel(const T* __restrict__ o, int ldo, const T* __restrict__ dout, int lddo,
                                                         float* __restrict__ delta, int B, int H, int L, int HD) {
    const int lane = threadIdx.x & 63;
    const long m = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (m >= (long)B * L) return;
    const int b = (int)(m / L), l = (int)(m % L);
    const int lph = HD / 8, nch = H * lph;
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
        for (int msk = 1; msk < lph; msk <<= 1) s += __shfl_xor(s, msk);
        if (act && (qd % lph) == 0) delta[((size_t)b * H + qd / lph) * L + l] = s;
    }
}

#define OD_BWD_LOOP_SYNC() __syncthreads()
// dK, dV: block owns 4 waves x NK*16 keys; loop over 64-query tiles.
//   S = Q K^T (cols = keys) ; dV^T += dO^T P ; dP = dO V^T ; dS = P*(dP - delta)*scale ; dK^T += Q^T dS
// LDS per stage: Q, dO row-major (+ Q^T, dO^T for f32); bf16 double-buffers the stage.
template <class T, int HD, int NK, int NWK, bool PRE>
__global__ __launch_bounds__(64 * NWK, (NWK == 8 ? 1 : 2)) void flash_bwd_dkv_kernel(const T* __restrict__ q, int ldq, const T* __restrict__ k, int ldk,
                                                               const T* __restrict__ v, int ldv, const T* __restrict__ dout, int lddo,
                                                               const float* __restrict__ lse, const float* __restrict__ delta,
                                                               T* __restrict__ dk, int lddk, T* __restrict__ dv, int lddv,
                                                               int B, int H, int L, float scale) {
    using St = Stage<T, HD>;
    constexpr int NS = HD / 32, ND = HD / 16, KB = NWK * NK * 16;
    static_assert(NWK == 4 || St::TR, "the register-staged (f32) path assumes 256 threads");
    constexpr int NSTAGE = St::TR ? 2 : 1;
    constexpr int STAGE = (2 + 2 * St::NT) * St::BYTES + 512;     // tiles + (lse, delta)
    OD_DYN_SMEM(smem);
    const int nkt = (L + KB - 1) / KB;
    int ktile, bh;
    if (!attn_block_coords(nkt, B * H, ktile, bh)) return;
    const int b = bh / H, h = bh % H;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, x = lane & 15, g = lane >> 4;
    const T* qb = q + (size_t)b * L * ldq + h * HD;
    const T* kb = k + (size_t)b * L * ldk + h * HD;
    const T* vb = v + (size_t)b * L * ldv + h * HD;
    const T* dob = dout + (size_t)b * L * lddo + h * HD;
    const float* lseb = lse + ((size_t)b * H + h) * L;
    const float* delb = delta + ((size_t)b * H + h) * L;
    const int key0 = ktile * KB + wave * NK * 16;
    // VALU diet: the score and dP accumulators START at -lse/scale and -delta (the MFMA's C operand), so
    // P = exp2(c * acc) and dS' = P * acc' need one packed multiply each — no fma/sub per element; the factor
    // `scale` of dS is applied once to dK at the end.
    // PRE (q pre-multiplied by scale*log2e): the accumulator starts at -lse*log2e and IS the exponent; dK then carries
    // ln2 instead of scale (dL/dk = scale * sum dS q = ln2 * sum dS q').
    const float c = scale * LOG2E, inv_scale = PRE ? LOG2E : 1.0f / scale, out_scale = PRE ? LN2 : scale;

    od_frag<T> fk[NK][NS], fv[NK][NS];
#pragma unroll
    for (int ki = 0; ki < NK; ki++) {
        int row = key0 + ki * 16 + x; row = row < L ? row : L - 1;
#pragma unroll
        for (int s = 0; s < NS; s++) {
            od_frag_load(fk[ki][s], kb + (size_t)row * ldk + s * 32 + g * 8);
            od_frag_load(fv[ki][s], vb + (size_t)row * ldv + s * 32 + g * 8);
        }
    }
    f32x4 dkacc[NK][ND], dvacc[NK][ND];
#pragma unroll
    for (int ki = 0; ki < NK; ki++)
#pragma unroll
        for (int dt = 0; dt < ND; dt++) { dkacc[ki][dt] = (f32x4)(0.f); dvacc[ki][dt] = (f32x4)(0.f); }

    const int nqt = (L + 63) / 64;
    St sq, so;
    float r_lse = 0.f, r_del = 0.f;
    auto gload = [&](int qt) {
        sq.load(qb, ldq, qt * 64, L); so.load(dob, lddo, qt * 64, L);
        if (threadIdx.x < 64) {
            const int row = qt * 64 + threadIdx.x;
            r_lse = row < L ? lseb[row] : 0.f;
            r_del = row < L ? delb[row] : 0.f;
        }
    };
    // stage layout: [Q][dO]([Q^T][dO^T])[lse(64) delta(64)]
    auto lstore = [&](unsigned char* st) {
        sq.store_rowmajor(st); so.store_rowmajor(st + St::BYTES);
        if constexpr (!St::TR) { sq.store_transposed(st + 2 * St::BYTES); so.store_transposed(st + 3 * St::BYTES); }
        float* sl = (float*)(st + (2 + 2 * St::NT) * St::BYTES);
        if (threadIdx.x < 64) { sl[threadIdx.x] = -r_lse * inv_scale; sl[64 + threadIdx.x] = -r_del; }
    };
    // bf16: Q/dO tiles by LDS-DMA into the other stage at the top of the iteration; lse/delta by registers
    auto gload_small = [&](int qt) {
        if (threadIdx.x < 64) {
            const int row = qt * 64 + threadIdx.x;
            r_lse = row < L ? lseb[row] : 0.f;
            r_del = row < L ? delb[row] : 0.f;
        }
    };
    auto lstore_small = [&](unsigned char* st) {
        float* sl = (float*)(st + (2 + 2 * St::NT) * St::BYTES);
        if (threadIdx.x < 64) { sl[threadIdx.x] = -r_lse * inv_scale; sl[64 + threadIdx.x] = -r_del; }
    };
    auto dma = [&](int qt, unsigned char* st) {
        St::template dma_rowmajor<NWK>(qb, ldq, qt * 64, L, st);
        St::template dma_rowmajor<NWK>(dob, lddo, qt * 64, L, st + St::BYTES);
    };
    if constexpr (St::TR) { dma(0, smem); gload_small(0); lstore_small(smem); }
    else { gload(0); lstore(smem); }
    __syncthreads();
    int cur = 0;
    const bool kragged = ktile * KB + KB > L;
    auto tile = [&](int qt, auto masked_t) {
        constexpr bool MASKED = decltype(masked_t)::value;
        if (qt + 1 < nqt) {
            if constexpr (St::TR) { dma(qt + 1, smem + (cur ^ 1) * STAGE); gload_small(qt + 1); }
            else gload(qt + 1);
        }
        const unsigned char* st = smem + cur * STAGE;
        const unsigned char* tQ = st;
        const unsigned char* tO = st + St::BYTES;
        const unsigned char* tQT = st + 2 * St::BYTES;     // f32 only
        const unsigned char* tOT = st + 3 * St::BYTES;     // f32 only
        const float* s_lse = (const float*)(st + (2 + 2 * St::NT) * St::BYTES);
        const float* s_delta = s_lse + 64;
        const int qbase = qt * 64;
        // per 32-query slab u (two 16-row tiles): scores, probabilities, dP, dS
        od_frag<T> fp[NK][2], fds[NK][2];
#pragma unroll
        for (int t4 = 0; t4 < 4; t4++) {
            od_frag<T> fqr[NS], fdo[NS];
#pragma unroll
            for (int s = 0; s < NS; s++) {
                frag_contig<St::ROWB>(fqr[s], tQ, t4 * 16 + x, s * 32 + g * 8);
                frag_contig<St::ROWB>(fdo[s], tO, t4 * 16 + x, s * 32 + g * 8);
            }
            const f32x4 l4 = *(const f32x4*)(s_lse + t4 * 16 + 4 * g);
            const f32x4 d4 = *(const f32x4*)(s_delta + t4 * 16 + 4 * g);

#pragma unroll
            for (int ki = 0; ki < NK; ki++) {
                f32x4 sa = l4, pa = d4;
#pragma unroll
                for (int s = 0; s < NS; s++) { sa = od_mma(fqr[s], fk[ki][s], sa); pa = od_mma(fdo[s], fv[ki][s], pa); }
                f32x4 e = sa;
                if constexpr (!PRE) e = od_mul4s(sa, c);
                f32x4 p;
#pragma unroll
                for (int r = 0; r < 4; r++) p[r] = od_exp2(e[r]);
                if constexpr (MASKED) {
                    const bool kvalid = key0 + ki * 16 + x < L;
#pragma unroll
                    for (int r = 0; r < 4; r++)
                        if (!(kvalid && (qbase + t4 * 16 + 4 * g + r < L))) p[r] = 0.f;
                }
                const f32x4 ds = od_mul4(p, pa);
                od_frag_set4(fp[ki][t4 >> 1], t4 & 1, p[0], p[1], p[2], p[3]);
                od_frag_set4(fds[ki][t4 >> 1], t4 & 1, ds[0], ds[1], ds[2], ds[3]);
            }
        }
        // dV^T += dO^T P ; dK^T += Q^T dS     (A rows = features, k = permuted queries, cols = keys)
#pragma unroll
        for (int dt = 0; dt < ND; dt++)
#pragma unroll
            for (int u = 0; u < 2; u++) {
                od_frag<T> fot, fqt;
                frag_cols<St::ROWB, St::TROWB>(fot, tO, tOT, dt * 16, x, u, g);
                frag_cols<St::ROWB, St::TROWB>(fqt, tQ, tQT, dt * 16, x, u, g);
#pragma unroll
                for (int ki = 0; ki < NK; ki++) {
                    dvacc[ki][dt] = od_mma(fot, fp[ki][u], dvacc[ki][dt]);
                    dkacc[ki][dt] = od_mma(fqt, fds[ki][u], dkacc[ki][dt]);
                }
            }
        if (NSTAGE == 1) __syncthreads();
        if (qt + 1 < nqt) {
            if constexpr (St::TR) lstore_small(smem + (cur ^ 1) * STAGE);
            else lstore(smem + (NSTAGE == 2 ? (cur ^ 1) * STAGE : 0));
        }
        OD_BWD_LOOP_SYNC();
        if (NSTAGE == 2) cur ^= 1;
    };
    const int nfull = kragged ? 0 : L / 64;
    for (int qt = 0; qt < nfull; qt++) tile(qt, std::false_type{});
    for (int qt = nfull; qt < nqt; qt++) tile(qt, std::true_type{});
#pragma unroll
    for (int ki = 0; ki < NK; ki++) {
        const int row = key0 + ki * 16 + x;
        if (row < L) {
            T* dkr = dk + ((size_t)b * L + row) * lddk + h * HD;
            T* dvr = dv + ((size_t)b * L + row) * lddv + h * HD;
#pragma unroll
            for (int dt = 0; dt < ND; dt++) {
                dkacc[ki][dt] *= out_scale;
                st4(dkr + dt * 16 + 4 * g, dkacc[ki][dt][0], dkacc[ki][dt][1], dkacc[ki][dt][2], dkacc[ki][dt][3]);
                st4(dvr + dt * 16 + 4 * g, dvacc[ki][dt][0], dvacc[ki][dt][1], dvacc[ki][dt][2], dvacc[ki][dt][3]);
            }
        }
    }
}

// (Round 3 also built a software-pipelined form of the dK/dV loop — the score / softmax half "P1" of one 32-query half-tile interleaved with the
// MFMA-only dV / dK half "P2" of the PREVIOUS half-tile, a ring of three LDS stages, buffer-addressed asm DMA, lane-constant LDS offsets, the
// MFMA / filler interleave pinned with sched_group_barrier — so that every stretch of the loop carries the same ~2 fillers per MFMA gap instead of 3
// then 0.7.  With both halves' P / dS fragments live beside P1's and P2's temporaries it fits only 32 keys per wave (48: 416 B of scratch, 94 ms), and at
// 32 keys it measured 27.0 ms (scheduler-placed) / 28.6 ms (pinned) per layer against 25.0 for this kernel: profiles/r03l_ab_dkv_pipe.txt, r03m.  Removed.)

// dQ: block owns 4 waves x NQ*16 queries; loop over 64-key tiles.
//   S^T = K Q^T ; dP^T = V dO^T ; dS^T = P^T*(dP^T - delta)*scale ; dQ^T += K^T dS^T
template <class T, int HD, int NQ, int NW, bool PRE>
__global__ __launch_bounds__(64 * NW, (NW >= 6 ? 3 : 2)) void flash_bwd_dq_kernel(const T* __restrict__ q, int ldq, const T* __restrict__ k, int ldk,
                                                              const T* __restrict__ v, int ldv, const T* __restrict__ dout, int lddo,
                                                              const float* __restrict__ lse, const float* __restrict__ delta,
                                                              T* __restrict__ dq, int lddq, int B, int H, int L, float scale) {
    using St = Stage<T, HD>;
    constexpr int NS = HD / 32, ND = HD / 16, QB = NW * NQ * 16;
    constexpr int NSTAGE = St::TR ? 2 : 1;
    constexpr int STAGE = (2 + St::NT) * St::BYTES;      // K, V (+ K^T for f32)
    OD_DYN_SMEM(smem);
    const int nqt = (L + QB - 1) / QB;
    int qtile, bh;
    if (!attn_block_coords(nqt, B * H, qtile, bh)) return;
    const int b = bh / H, h = bh % H;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, x = lane & 15, g = lane >> 4;
    const T* qb = q + (size_t)b * L * ldq + h * HD;
    const T* kb = k + (size_t)b * L * ldk + h * HD;
    const T* vb = v + (size_t)b * L * ldv + h * HD;
    const T* dob = dout + (size_t)b * L * lddo + h * HD;
    const int q0 = qtile * QB + wave * NQ * 16;
    // accumulator-init trick and PRE as in the dK/dV kernel; with PRE the result is the gradient of the pre-multiplied q
    const float c = scale * LOG2E, inv_scale = PRE ? LOG2E : 1.0f / scale, out_scale = PRE ? LN2 : scale;

    od_frag<T> fq[NQ][NS], fdo[NQ][NS];
    float r_lse[NQ], r_del[NQ];
#pragma unroll
    for (int qi = 0; qi < NQ; qi++) {
        int row = q0 + qi * 16 + x; row = row < L ? row : L - 1;
#pragma unroll
        for (int s = 0; s < NS; s++) {
            od_frag_load(fq[qi][s], qb + (size_t)row * ldq + s * 32 + g * 8);
            od_frag_load(fdo[qi][s], dob + (size_t)row * lddo + s * 32 + g * 8);
        }
        r_lse[qi] = -lse[((size_t)b * H + h) * L + row] * inv_scale;
        r_del[qi] = -delta[((size_t)b * H + h) * L + row];
    }
    f32x4 dqacc[NQ][ND];
#pragma unroll
    for (int qi = 0; qi < NQ; qi++)
#pragma unroll
        for (int dt = 0; dt < ND; dt++) dqacc[qi][dt] = (f32x4)(0.f);

    const int nkt = (L + 63) / 64;
    St sk, sv;
    auto lstore = [&](unsigned char* st) {
        sk.store_rowmajor(st); sv.store_rowmajor(st + St::BYTES);
        if constexpr (!St::TR) sk.store_transposed(st + 2 * St::BYTES);
    };
    auto dma = [&](int kt, unsigned char* st) {
        St::template dma_rowmajor<NW>(kb, ldk, kt * 64, L, st);
        St::template dma_rowmajor<NW>(vb, ldv, kt * 64, L, st + St::BYTES);
    };
    if constexpr (St::TR) dma(0, smem);
    else { sk.load(kb, ldk, 0, L); sv.load(vb, ldv, 0, L); lstore(smem); }
    __syncthreads();
    int cur = 0;
    auto tile = [&](int kt, auto masked_t) {
        constexpr bool MASKED = decltype(masked_t)::value;
        if (kt + 1 < nkt) {
            if constexpr (St::TR) dma(kt + 1, smem + (cur ^ 1) * STAGE);
            else { sk.load(kb, ldk, (kt + 1) * 64, L); sv.load(vb, ldv, (kt + 1) * 64, L); }
        }
        const unsigned char* st = smem + cur * STAGE;
        const unsigned char* tK = st;
        const unsigned char* tV = st + St::BYTES;
        const unsigned char* tKT = st + 2 * St::BYTES;      // f32 only
        const int kbase = kt * 64;
        od_frag<T> fds[NQ][2];
#pragma unroll
        for (int t4 = 0; t4 < 4; t4++) {
            od_frag<T> fkr[NS], fvr[NS];
#pragma unroll
            for (int s = 0; s < NS; s++) {
                frag_contig<St::ROWB>(fkr[s], tK, t4 * 16 + x, s * 32 + g * 8);
                frag_contig<St::ROWB>(fvr[s], tV, t4 * 16 + x, s * 32 + g * 8);
            }
#pragma unroll
            for (int qi = 0; qi < NQ; qi++) {
                f32x4 sa = (f32x4)(r_lse[qi]), pa = (f32x4)(r_del[qi]);
#pragma unroll
                for (int s = 0; s < NS; s++) { sa = od_mma(fkr[s], fq[qi][s], sa); pa = od_mma(fvr[s], fdo[qi][s], pa); }
                f32x4 e = sa;
                if constexpr (!PRE) e = od_mul4s(sa, c);
                f32x4 p;
#pragma unroll
                for (int r = 0; r < 4; r++) p[r] = od_exp2(e[r]);
                if constexpr (MASKED) {
#pragma unroll
                    for (int r = 0; r < 4; r++)
                        if (kbase + t4 * 16 + 4 * g + r >= L) p[r] = 0.f;
                }
                const f32x4 ds = od_mul4(p, pa);
                od_frag_set4(fds[qi][t4 >> 1], t4 & 1, ds[0], ds[1], ds[2], ds[3]);
            }
        }
#pragma unroll
        for (int dt = 0; dt < ND; dt++)
#pragma unroll
            for (int u = 0; u < 2; u++) {
                od_frag<T> fkt;
                frag_cols<St::ROWB, St::TROWB>(fkt, tK, tKT, dt * 16, x, u, g);
#pragma unroll
                for (int qi = 0; qi < NQ; qi++) dqacc[qi][dt] = od_mma(fkt, fds[qi][u], dqacc[qi][dt]);
            }
        if (NSTAGE == 1) __syncthreads();
        if constexpr (!St::TR) { if (kt + 1 < nkt) lstore(smem + (NSTAGE == 2 ? (cur ^ 1) * STAGE : 0)); }
        OD_BWD_LOOP_SYNC();
        if (NSTAGE == 2) cur ^= 1;
    };
    const int nfull = L / 64;
    for (int kt = 0; kt < nfull; kt++) tile(kt, std::false_type{});
    if (nfull < nkt) tile(nfull, std::true_type{});
#pragma unroll
    for (int qi = 0; qi < NQ; qi++) {
        const int row = q0 + qi * 16 + x;
        if (row < L) {
            T* dqr = dq + ((size_t)b * L + row) * lddq + h * HD;
#pragma unroll
            for (int dt = 0; dt < ND; dt++) {
                dqacc[qi][dt] *= out_scale;
                st4(dqr + dt * 16 + 4 * g, dqacc[qi][dt][0], dqacc[qi][dt][1], dqacc[qi][dt][2], dqacc[qi][dt][3]);
            }
        }
    }
}

// (The 32x32x16-MFMA form of the dQ kernel, the norm + RoPE backward in the dQ / dK epilogues, the interleaved-chain and 8-wave dK/dV variants and the
// timing-only modes of rounds 2-3 live in variants/attn_r03_variants.hip: tools/build_variant.sh compiles that file in place of this one on demand.)

#ifndef OD_ATTN_NW
#define OD_ATTN_NW 4      // waves per workgroup of the bf16 forward / dQ kernels.  6 (K/V streamed once per 192 queries) measured 0.71x: a 6-wave group lands 2,2,1,1 on the SIMDs and a second group no longer fits at 3 waves/SIMD
#endif
#ifndef OD_FWD_NQT1_BELOW
#define OD_FWD_NQT1_BELOW 2048  // fp32 forward, head_dim 64: 16 queries per wave when the 32-query grid has fewer workgroups than this (0 = never).  At the sampler's size
                                // (576 workgroups = 1.09 rounds of the chip run as two) the fp32 call goes 293.5 -> 287.3 ms; the fp32-bf16x3 product, whose hi / lo fragment
                                // split is amortised over half the MFMAs then, 155.5 -> 170.8 ms: fp32 only
#endif
#ifndef OD_FWD32
#define OD_FWD32 1        // bf16, head_dim 64: the 32x32x16 kernel (0 = the 16x16x32 kernel, kept for A/B and for hd 32 / fp32)
#endif
#ifndef OD_FWD32_NQB
#define OD_FWD32_NQB 1    // 32-query blocks per wave
#endif
// host-side helper objects of the two-stream backward, created explicitly by the caller (od_attn_aux_create): the library itself holds no state
struct AttnAux {
#if !defined(OD_EMU)
    hipStream_t side; hipEvent_t fork, join;
#else
    int unused;
#endif
};

template <class T, int HD, bool PRE>
int launch_fwd(const void* q, int ldq, const void* k, int ldk, const void* v, int ldv, void* o, int ldo, float* lse, int B,
               int H, int L, float scale, hipStream_t st) {
    if constexpr (std::is_same<T, f16_t>::value) {          // half operands: q, k, v are IEEE half, o is written as bf16 (head_dim 64 only)
        static_assert(HD == 64, "the half-operand forward exists for head_dim 64");
#if OD_FWD16X
        static const int fwd16x_min_l_h = od_env_int("OD_FWD16X_MIN_L", OD_FWD16X_MIN_L);
        if (PRE && L >= fwd16x_min_l_h) {
            constexpr int NQT = OD_FWD16X_NQT;
            const int grid = attn_grid((L + 4 * 16 * NQT - 1) / (4 * 16 * NQT), B * H);
            OD_LAUNCH_DYN((flash_fwd16x_kernel<4, NQT, f16_t>), dim3(grid), dim3(256), (4 * Stage<bf16_t, HD>::BYTES), st, (const bf16_t*)q, ldq,
                          (const bf16_t*)k, ldk, (const bf16_t*)v, ldv, (bf16_t*)o, ldo, lse, B, H, L);
            OD_CHECK_LAUNCH();
            return 0;
        }
#endif
        constexpr int NW = 4, NQB = OD_FWD32_NQB;
        const int grid = attn_grid((L + NW * NQB * 32 - 1) / (NW * NQB * 32), B * H);
        OD_LAUNCH_DYN((flash_fwd32_kernel<NW, NQB, PRE, f16_t>), dim3(grid), dim3(64 * NW), (4 * Stage<bf16_t, HD>::BYTES), st, (const bf16_t*)q, ldq,
                      (const bf16_t*)k, ldk, (const bf16_t*)v, ldv, (bf16_t*)o, ldo, lse, B, H, L, scale);
        OD_CHECK_LAUNCH();
        return 0;
    } else {
#ifndef OD_FWD32_NW
#define OD_FWD32_NW 4     // waves per workgroup of the bf16 / hd 64 forward (8: one workgroup per CU, the K / V tiles staged once for 256 queries; A/B)
#endif
#if OD_FWD16X
    // long sequences: the 16x16x32 form (7.48 against 7.80 ms at B = 32 x L = 8192); at the sampler's L = 1115 the 32x32x16 form is ahead by 1.5 %
    // (66.7 against 65.7 ms per 50-step call: profiles/r06h_ab_fwd16x_nqt_sampler.txt).  OD_FWD16X_MIN_L moves the switch.
    static const int fwd16x_min_l = od_env_int("OD_FWD16X_MIN_L", OD_FWD16X_MIN_L);
    if (std::is_same<T, bf16_t>::value && HD == 64 && PRE && L >= fwd16x_min_l) {
      if constexpr (std::is_same<T, bf16_t>::value && HD == 64 && PRE) {
        constexpr int NQT = OD_FWD16X_NQT, NWX = OD_FWD16X_NW;
        const int grid = attn_grid((L + NWX * 16 * NQT - 1) / (NWX * 16 * NQT), B * H);
        OD_LAUNCH_DYN((flash_fwd16x_kernel<NWX, NQT, bf16_t>), dim3(grid), dim3(64 * NWX), (4 * Stage<T, HD>::BYTES), st, (const bf16_t*)q, ldq,
                      (const bf16_t*)k, ldk, (const bf16_t*)v, ldv, (bf16_t*)o, ldo, lse, B, H, L);
        OD_CHECK_LAUNCH();
        return 0;
      }
    }
#endif
    if constexpr (OD_FWD32 && std::is_same<T, bf16_t>::value && HD == 64) {
        constexpr int NW = OD_FWD32_NW, NQB = OD_FWD32_NQB;
        const int grid = attn_grid((L + NW * NQB * 32 - 1) / (NW * NQB * 32), B * H);
        OD_LAUNCH_DYN((flash_fwd32_kernel<NW, NQB, PRE>), dim3(grid), dim3(64 * NW), (4 * Stage<T, HD>::BYTES), st, (const bf16_t*)q, ldq,
                      (const bf16_t*)k, ldk, (const bf16_t*)v, ldv, (bf16_t*)o, ldo, lse, B, H, L, scale);
        OD_CHECK_LAUNCH();
        return 0;
    }
#ifndef OD_FWD_X3P
#define OD_FWD_X3P 1      // fp32-as-3-x-bf16, head_dim 64: the kernel that splits K / V once per tile at staging time (0 = the generic kernel, for A/B)
#endif
#ifndef OD_X3P_NQT
#define OD_X3P_NQT 2      // 16-query tiles per wave of the fp32-as-3-x-bf16 forward (1: A/B, profiles/r06l_ab_x3p_nqt.txt)
#endif
    if constexpr (OD_FWD_X3P && std::is_same<T, f32x3_t>::value && HD == 64) {
        const int grid = attn_grid((L + 4 * 16 * OD_X3P_NQT - 1) / (4 * 16 * OD_X3P_NQT), B * H);
        OD_LAUNCH_DYN((flash_fwd_x3p_kernel<4, PRE, OD_X3P_NQT>), dim3(grid), dim3(256), (2 * 4 * 64 * 128), st, (const float*)q, ldq, (const float*)k, ldk,
                      (const float*)v, ldv, (float*)o, ldo, lse, B, H, L, scale);
        OD_CHECK_LAUNCH();
        return 0;
    }
    constexpr int NW = Stage<T, HD>::TR ? OD_ATTN_NW : 4;
    // fp32: 16 queries per wave when the 32-query grid is only a few rounds of the chip (sampler sizes): half-size waves quantise better
    static const int nqt1_below = od_env_int("OD_FWD_NQT1_BELOW", OD_FWD_NQT1_BELOW);       // workgroups of the 32-query form
    const int blocks2 = ((L + NW * 32 - 1) / (NW * 32)) * B * H;
    if (std::is_same<T, float>::value && HD == 64 && blocks2 < nqt1_below) {
        const int grid = attn_grid((L + NW * 16 - 1) / (NW * 16), B * H);
        OD_LAUNCH_DYN((flash_fwd_kernel<T, HD, NW, PRE, 1>), dim3(grid), dim3(64 * NW), (4 * Stage<T, HD>::BYTES), st, (const T*)q, ldq, (const T*)k, ldk, (const T*)v, ldv,
                  (T*)o, ldo, lse, B, H, L, scale);
        OD_CHECK_LAUNCH();
        return 0;
    }
    const int grid = attn_grid((L + NW * 32 - 1) / (NW * 32), B * H);
    OD_LAUNCH_DYN((flash_fwd_kernel<T, HD, NW, PRE>), dim3(grid), dim3(64 * NW), (4 * Stage<T, HD>::BYTES), st, (const T*)q, ldq, (const T*)k, ldk, (const T*)v, ldv,
              (T*)o, ldo, lse, B, H, L, scale);
    OD_CHECK_LAUNCH();
    return 0;
    }
}

template <class T, int HD, int NK, int NQ, bool PRE>
int launch_bwd(const void* q, int ldq, const void* k, int ldk, const void* v, int ldv, const void* o, int ldo, const void* dout,
               int lddo, const float* lse, float* delta, void* dq, int lddq, void* dk, int lddk, void* dv, int lddv, int B, int H,
               int L, float scale, hipStream_t st, const AttnAux* aux = nullptr) {
    const long M = (long)B * L;
    OD_LAUNCH((attn_delta_kernel<T>), dim3((unsigned)((M + 3) / 4)), dim3(256), 0, st, (const T*)o, ldo, (const T*)dout, lddo, delta,
              B, H, L, HD);
    // Two-stream form: with a side stream (od_attn_aux_create) the dQ kernel runs beside the dK/dV kernel — both depend on delta alone — and its
    // workgroups fill the CUs the other kernel's last block round leaves idle: 24.69 -> 24.43 ms per layer (profiles/r03h_ab_bwd_two_streams.txt).
    hipStream_t st_q = st;
#if !defined(OD_EMU)
    static const int two_streams = od_env_int("OD_BWD_2STREAM", 1);      // 0: ignore the side stream (A/B)
    if (aux && two_streams) {
        // a side stream that cannot be forked into (another device's, a destroyed one) is not an error of this call: run single-stream
        if (hipEventRecord(aux->fork, st) == hipSuccess && hipStreamWaitEvent(aux->side, aux->fork, 0) == hipSuccess) st_q = aux->side;
        else (void)hipGetLastError();
    }
#endif
    constexpr int NWK = 4;
    const int gk = attn_grid((L + 16 * NWK * NK - 1) / (16 * NWK * NK), B * H);
    OD_LAUNCH_DYN((flash_bwd_dkv_kernel<T, HD, NK, NWK, PRE>), dim3(gk), dim3(64 * NWK),
                  ((Stage<T, HD>::TR ? 2 : 1) * ((2 + 2 * Stage<T, HD>::NT) * Stage<T, HD>::BYTES + 512)), st, (const T*)q, ldq, (const T*)k, ldk,
                  (const T*)v, ldv, (const T*)dout, lddo, lse, (const float*)delta, (T*)dk, lddk, (T*)dv, lddv, B, H, L, scale);
    constexpr int NWQ = Stage<T, HD>::TR ? OD_ATTN_NW : 4;
    const int gq = attn_grid((L + 16 * NQ * NWQ - 1) / (16 * NQ * NWQ), B * H);
    OD_LAUNCH_DYN((flash_bwd_dq_kernel<T, HD, NQ, NWQ, PRE>), dim3(gq), dim3(64 * NWQ),
                  ((Stage<T, HD>::TR ? 2 : 1) * ((2 + Stage<T, HD>::NT) * Stage<T, HD>::BYTES)), st_q, (const T*)q, ldq, (const T*)k, ldk,
                  (const T*)v, ldv, (const T*)dout, lddo, lse, (const float*)delta, (T*)dq, lddq, B, H, L, scale);
#if !defined(OD_EMU)
    if (st_q != st) { if (hipEventRecord(aux->join, st_q) != hipSuccess || hipStreamWaitEvent(st, aux->join, 0) != hipSuccess) return OD_ERR_ARG; }
#endif
    OD_CHECK_LAUNCH();
    return 0;
}

}  // namespace

extern "C" int od_flash_attn_fwd(int dtype, const void* q, int ldq, const void* k, int ldk, const void* v, int ldv, void* o,
                                 int ldo, float* lse, int B, int H, int L, int hd, float scale, int q_prescaled, void* stream) {
    if (ldq % 8 || ldk % 8 || ldv % 8 || ldo % 8) return OD_ERR_ALIGN;
    hipStream_t st = (hipStream_t)stream;
#define FWD(TT, HDV) (q_prescaled ? launch_fwd<TT, HDV, true>(q, ldq, k, ldk, v, ldv, o, ldo, lse, B, H, L, scale, st) \
                                  : launch_fwd<TT, HDV, false>(q, ldq, k, ldk, v, ldv, o, ldo, lse, B, H, L, scale, st))
    if (dtype == OD_BF16 && hd == 64) return FWD(bf16_t, 64);
    if (dtype == OD_F16 && hd == 64) return FWD(f16_t, 64);
    if (dtype == OD_BF16 && hd == 32) return FWD(bf16_t, 32);
    if (dtype == OD_F32 && hd == 64) return FWD(float, 64);
    if (dtype == OD_F32 && hd == 32) return FWD(float, 32);
    if (dtype == OD_F32X3 && hd == 64) return FWD(f32x3_t, 64);
    if (dtype == OD_F32X3 && hd == 32) return FWD(f32x3_t, 32);
#undef FWD
    return OD_ERR_UNSUPPORTED;
}

#ifndef OD_BWD_NK
#define OD_BWD_NK 3
#endif
#ifndef OD_BWD_NQ
#define OD_BWD_NQ 4
#endif
extern "C" int od_flash_attn_bwd_passes(void) { return 7; }   // dK/dV kernel: S, dP, dV, dK; dQ kernel: S, dP, dQ

extern "C" int od_attn_aux_create(void** aux_out) {
    if (!aux_out) return OD_ERR_ARG;
    AttnAux* a = new AttnAux();
#if !defined(OD_EMU)
    hipError_t e = hipStreamCreateWithFlags(&a->side, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&a->fork, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&a->join, hipEventDisableTiming);
    if (e != hipSuccess) { delete a; return -(int)e - 1000; }
#endif
    *aux_out = a;
    return 0;
}
extern "C" int od_attn_aux_destroy(void* aux) {
    if (!aux) return OD_ERR_ARG;
    AttnAux* a = (AttnAux*)aux;
#if !defined(OD_EMU)
    (void)hipStreamSynchronize(a->side);
    (void)hipEventDestroy(a->fork); (void)hipEventDestroy(a->join); (void)hipStreamDestroy(a->side);
#endif
    delete a;
    return 0;
}

extern "C" int od_flash_attn_bwd_aux(int dtype, const void* q, int ldq, const void* k, int ldk, const void* v, int ldv, const void* o,
                                     int ldo, const void* dout, int lddo, const float* lse, float* delta, void* dq, int lddq,
                                     void* dk, int lddk, void* dv, int lddv, int B, int H, int L, int hd, float scale, int q_prescaled,
                                     void* aux, void* stream);

extern "C" int od_flash_attn_bwd(int dtype, const void* q, int ldq, const void* k, int ldk, const void* v, int ldv, const void* o,
                                 int ldo, const void* dout, int lddo, const float* lse, float* delta, void* dq, int lddq,
                                 void* dk, int lddk, void* dv, int lddv, int B, int H, int L, int hd, float scale, int q_prescaled,
                                 void* stream) {
    return od_flash_attn_bwd_aux(dtype, q, ldq, k, ldk, v, ldv, o, ldo, dout, lddo, lse, delta, dq, lddq, dk, lddk, dv, lddv, B, H, L, hd, scale,
                                 q_prescaled, nullptr, stream);
}

extern "C" int od_flash_attn_bwd_aux(int dtype, const void* q, int ldq, const void* k, int ldk, const void* v, int ldv, const void* o,
                                     int ldo, const void* dout, int lddo, const float* lse, float* delta, void* dq, int lddq,
                                     void* dk, int lddk, void* dv, int lddv, int B, int H, int L, int hd, float scale, int q_prescaled,
                                     void* aux, void* stream) {
    if (ldq % 8 || ldk % 8 || ldv % 8 || ldo % 8 || lddo % 8 || lddq % 8 || lddk % 8 || lddv % 8) return OD_ERR_ALIGN;
    hipStream_t st = (hipStream_t)stream;
#define ARGS q, ldq, k, ldk, v, ldv, o, ldo, dout, lddo, lse, delta, dq, lddq, dk, lddk, dv, lddv, B, H, L, scale, st, (const AttnAux*)aux
// Register tiles of the bf16 / hd 64 backward: 16-row tiles per wave.  Per streamed tile every wave pays a fixed budget — its LDS-DMA
// pieces, the row-wise AND transposed fragment reads of the whole tile pair, the barrier — whatever it owns (removing the in-loop DMA takes
// 25.5 -> 20.2 ms; deeper tile rings, trimmed loops and 8-wave workgroups change nothing or lose: profiles/r02l_ab_bwd_tile_budget.txt), so
// owning more rows per wave is the lever.  48 keys (dK/dV, 256 VGPRs) and 64 queries (dQ, 246 VGPRs) are what two waves per SIMD allow.
#ifndef OD_BWD_NK
#define OD_BWD_NK 3
#endif
#ifndef OD_BWD_NQ
#define OD_BWD_NQ 4
#endif
#define BWD(TT, HDV, NKV, NQV) (q_prescaled ? launch_bwd<TT, HDV, NKV, NQV, true>(ARGS) : launch_bwd<TT, HDV, NKV, NQV, false>(ARGS))
    if (dtype == OD_BF16 && hd == 64) return BWD(bf16_t, 64, OD_BWD_NK, OD_BWD_NQ);
    if (dtype == OD_BF16 && hd == 32) return BWD(bf16_t, 32, 2, 2);
    if (dtype == OD_F32 && hd == 64) return BWD(float, 64, 1, 1);
    if (dtype == OD_F32 && hd == 32) return BWD(float, 32, 1, 1);
#undef BWD
#undef ARGS
    return OD_ERR_UNSUPPORTED;
}
