// MFMA GEMMs for the denoiser's 1x1 Conv1d / Linear layers (frame-major rows).
//
//   NT : C[M,N]  = epi(A[M,K] · W[N,K]^T + bias)       forward and backward-data
//   TN : dW[N,K] += G[M,N]^T · A[M,K]                  backward-weight (fp32, split-M)
//
// Replaces the torch ops at the reference's nn.Conv1d(k=1)/nn.Linear call sites
// (common/attn.py:68-69, common/swiglu.py:21,25, models/diffusion/backbone.py:63,
// models/diffusion/model.py:45-50).  128x128 block tile, 4 waves each owning 64x64
// (4x4 MFMA 16x16 tiles), 128-byte K slabs double-buffered in LDS with a 16-byte-slot
// XOR swizzle, register-staged prefetch of the next slab under the MFMAs, and an
// epilogue staged through LDS so global stores are 16/32-byte rows.
#include "od_common.h"
#include "od_tiles.h"
#include "od_api_internal.h"

namespace {

#ifndef OD_GEMM_NT_STORE_MIN_N
#define OD_GEMM_NT_STORE_MIN_N 1024   // large-M NT kernel: outputs at least this wide are stored non-temporally
#endif
#ifndef OD_TN_BIG_MIN_TILES
#define OD_TN_BIG_MIN_TILES 8   // weight-gradient GEMM: 256x256 output tiles from this many tiles on (fewer: the M-splits' fp32 atomics dominate)
#endif
#ifndef OD_GEMM_BIG_MIN_M
#define OD_GEMM_BIG_MIN_M 32768   // rows from which the 256x256 kernels are used
#endif
#ifndef OD_NT_BIG_MIN_TILES
#define OD_NT_BIG_MIN_TILES 0     // (see launch_nt)
#endif

constexpr int BM = 128, BN = 128;
constexpr int STAGE_BYTES = 32768;  // A 16 KiB + B 16 KiB

__device__ __forceinline__ int swz(int row, int slot) { return row * 128 + (((slot) ^ (row & 7)) << 4); }

template <class T>
__device__ __forceinline__ void frag_from_lds(od_frag<T>& f, const unsigned char* tile, int row, int slab, int g);
template <>
__device__ __forceinline__ void frag_from_lds<bf16_t>(od_frag<bf16_t>& f, const unsigned char* tile, int row, int slab, int g) {
    f.v = *(const s16x8*)(tile + swz(row, slab * 4 + g));
}
template <class F>
__device__ __forceinline__ void frag_from_lds_f32(od_frag<F>& f, const unsigned char* tile, int row, int slab, int g) {
    f32x4 a = *(const f32x4*)(tile + swz(row, slab * 8 + 2 * g));
    f32x4 b = *(const f32x4*)(tile + swz(row, slab * 8 + 2 * g + 1));
    const float x8[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    od_frag_pack(f, x8);
}
template <>
__device__ __forceinline__ void frag_from_lds<float>(od_frag<float>& f, const unsigned char* tile, int row, int slab, int g) {
    frag_from_lds_f32(f, tile, row, slab, g);
}
template <>
__device__ __forceinline__ void frag_from_lds<f32x3_t>(od_frag<f32x3_t>& f, const unsigned char* tile, int row, int slab, int g) {
    frag_from_lds_f32(f, tile, row, slab, g);
}

template <>
__device__ __forceinline__ void frag_from_lds<f32x3w_t>(od_frag<f32x3w_t>& f, const unsigned char* tile, int row, int slab, int g) {
    frag_from_lds_f32(f, tile, row, slab, g);
}
// the WEIGHT operand's fragment.  f32x3w_t: the staged row holds 32 bf16 high halves (16-byte slots 0..3) and 32 low halves (slots 4..7)
// of the slab, written once by od_pack_weight — no conversion here.
template <class T>
__device__ __forceinline__ void frag_w_from_lds(od_frag<T>& f, const unsigned char* tile, int row, int slab, int g) { frag_from_lds<T>(f, tile, row, slab, g); }
template <>
__device__ __forceinline__ void frag_w_from_lds<f32x3w_t>(od_frag<f32x3w_t>& f, const unsigned char* tile, int row, int, int g) {
    f.hi = *(const s16x8*)(tile + swz(row, g));
    f.lo = *(const s16x8*)(tile + swz(row, 4 + g));
}

// One 128-byte-deep slab of MFMAs from a staged (A,B) pair.
template <class T, int WMT>
__device__ __forceinline__ void compute_stage(const unsigned char* sA, const unsigned char* sB, int wm, int wn, int lane,
                                              f32x4 (&acc)[WMT][4]) {
    constexpr int SLABS = (128 / (int)sizeof(T)) / 32;
    const int r16 = lane & 15, g = lane >> 4;
#pragma unroll
    for (int s = 0; s < SLABS; s++) {
        od_frag<T> fa[WMT], fb[4];
#pragma unroll
        for (int i = 0; i < WMT; i++) frag_from_lds<T>(fa[i], sA, wm * 16 * WMT + i * 16 + r16, s, g);
#pragma unroll
        for (int j = 0; j < 4; j++) frag_w_from_lds<T>(fb[j], sB, wn * 64 + j * 16 + r16, s, g);
#pragma unroll
        for (int i = 0; i < WMT; i++)
#pragma unroll
            for (int j = 0; j < 4; j++) acc[i][j] = od_mma(fa[i], fb[j], acc[i][j]);
    }
}

// XCD-aware tile order: block b runs on XCD b%8 (observed dispatch); each XCD walks
// its own row-tiles with the column tile fastest so an A row-panel is fetched from
// HBM once and re-read from that XCD's L2 by the other column tiles.
__device__ __forceinline__ bool tile_of_block(int tiles_m, int tiles_n, int& tm, int& tn) {
    const int b = blockIdx.x;
    const int xcd = b & 7, slot = b >> 3;
    tm = (slot / tiles_n) * 8 + xcd;
    tn = slot % tiles_n;
    return tm < tiles_m;
}

// OD_EPI_QKROPE (internal): the q/k RMSNorm + RoPE of attn.py:74-80 applied to the first n_rope columns of the output
// tile while it sits in LDS (a 128-column tile holds whole heads), the remaining columns (v) stored as they are.
struct RopeEpi {
    const float* wq; const float* wk; const float* table;   // norm weights [hd], (cos, sin) table [L][hd/2][2]
    int L, dh, hd, n_rope; float eps, q_scale;               // dh = H*hd (q columns), n_rope = 2*dh; q outputs * q_scale
    void* qk_out; int ldqk;                                  // large-M kernel only: normed + rotated q, k go HERE (C keeps the pre-norm
                                                             // values the backward needs); NULL: they replace C[:, :n_rope]
    int f16;                                                 // with qk_out, 4-wave kernel only: qk_out and the v columns of C (>= n_rope) are
                                                             // written as IEEE half ("attention in fp16"); C's q / k columns stay bf16
};
constexpr int OD_EPI_QKROPE = 2;

// od_gemm_tn_blocks: the N columns of G come in blocks of `block` columns of which the first `valid` are live (the SwiGLU hidden width 1365
// padded to 1408, twice: v then g); output row of column n = (n / block) * valid + n % block, the padding columns are dropped.  block = 0: identity.
struct TnRowMap { int block, valid; };
__device__ __forceinline__ bool tn_map_row(const TnRowMap rm, int n, int N, int& out) {
    out = n;
    if (n >= N) return false;
    if (rm.block) {
        const int q = n / rm.block, r = n - q * rm.block;
        out = q * rm.valid + r;
        return r < rm.valid;
    }
    return true;
}

// WMT = 16-row MFMA tiles per wave along m: 4 -> 128 x 128 block tile, 2 -> 64 x 128 (twice the workgroups, for
// launches whose 128-row tiling would leave CUs idle: the sampler's M = B*L = 4460 against N = 512)
// LDS stages of gemm_nt_kernel and its dynamic LDS size.  At the sampler's sizes (M = 4460: one workgroup per CU, 16-44 k-tiles of ~0.15 us of
// MFMAs each) an iteration of the two-stage loop lasts one fetch latency (~1 us); with three stages two fetches are in flight.
#ifndef OD_GEMM_NT_STAGES3
#define OD_GEMM_NT_STAGES3 2       // 1: three stages for every type at WMT <= 2; 2: not for plain fp32; 0: never
#endif
template <class T, bool DMA, int WMT>
constexpr int gemm_nt_stages() {
    return (DMA && (WMT <= 2 || OD_GEMM_NT_STAGES3 == 3) && (OD_GEMM_NT_STAGES3 == 1 || (OD_GEMM_NT_STAGES3 >= 2 && !std::is_same<T, float>::value))) ? 3 : 2;
}
template <class T, bool DMA, int WMT>
constexpr int gemm_nt_smem_bytes() {
    constexpr int stg = 32 * WMT * 128 + 16384, csz = 32 * WMT * 512, n = gemm_nt_stages<T, DMA, WMT>();
    return n * stg > csz ? n * stg : csz;
}

template <class T, int EPI, bool DMA, int WMT>
__global__ __launch_bounds__(256) void gemm_nt_kernel(const T* __restrict__ A, int lda, const T* __restrict__ W, int ldw,
                                                      const float* __restrict__ bias, T* __restrict__ C, int ldc,
                                                      int M, int N, int K, int accumulate, RopeEpi rp) {
    constexpr int BK = 128 / (int)sizeof(T);  // elements per slab row
    constexpr int CH = 16 / (int)sizeof(T);   // elements per 16-byte chunk
    constexpr int ASZ = 32 * WMT * 128;                    // A tile bytes per stage (BMT rows x 128 B)
    constexpr int STG = ASZ + 16384;                       // + W tile
    constexpr int CSZ = 32 * WMT * 512;                    // epilogue image, f32 [BMT][128]
    constexpr int NSTG = gemm_nt_stages<T, DMA, WMT>();       // DMA, WMT <= 2: three stages (two tiles in flight), 60 / 72 KiB — two workgroups per CU still fit
    OD_DYN_SMEM(smem);                                     // gemm_nt_smem_bytes<T, DMA, WMT>()
    static_assert(NSTG * STG >= CSZ || !DMA || WMT == 4, "");

    constexpr int BMT = 32 * WMT;
    const int tiles_m = (M + BMT - 1) / BMT, tiles_n = (N + BN - 1) / BN;
    int tm, tn;
    if (!tile_of_block(tiles_m, tiles_n, tm, tn)) return;
    const int m0 = tm * BMT, n0 = tn * BN;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;

    f32x4 acc[WMT][4];
#pragma unroll
    for (int i = 0; i < WMT; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) acc[i][j] = (f32x4)(0.f);

    u32x4 ra[4], rb[4];
    const int nk = (K + BK - 1) / BK;

    auto gload = [&](int kt) {
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int c = tid + 256 * i, row = c >> 3, slot = c & 7;
            const int k = kt * BK + slot * CH;
            int ar = m0 + row; ar = ar < M ? ar : M - 1;
            int br = n0 + row; br = br < N ? br : N - 1;
            if (k < K) {
                if (i < WMT) ra[i] = *(const u32x4*)(A + (size_t)ar * lda + k);
                rb[i] = *(const u32x4*)(W + (size_t)br * ldw + k);
            } else {
                ra[i] = (u32x4)(0u); rb[i] = (u32x4)(0u);
            }
        }
    };
    auto lstore = [&](int buf) {
        unsigned char* sA = smem + buf * STG;
        unsigned char* sB = sA + ASZ;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int c = tid + 256 * i, row = c >> 3, slot = c & 7;
            if (i < WMT) *(u32x4*)(sA + swz(row, slot)) = ra[i];
            *(u32x4*)(sB + swz(row, slot)) = rb[i];
        }
    };

    if constexpr (DMA) {
        // LDS-DMA staging (requires K % BK == 0): wave w streams 1 KiB pieces = 8 tile rows x 128 B; the
        // XOR swizzle is applied to the SOURCE column so the LDS image is the same one the reads expect.
        auto dma = [&](int kt, int buf) {
            unsigned char* sA = smem + buf * STG;
            unsigned char* sB = sA + ASZ;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int row = (wave * 4 + i) * 8 + (lane >> 3);
                const int slot = (lane & 7) ^ (row & 7);
                const int k = kt * BK + slot * CH;
                int br = n0 + row; br = br < N ? br : N - 1;
                if constexpr (NSTG == 3) od_glds16_async(W + (size_t)br * ldw + k, sB + (wave * 4 + i) * 1024);
                else od_glds16(W + (size_t)br * ldw + k, sB + (wave * 4 + i) * 1024 + lane * 16);
                if (i < WMT) {
                    const int rowa = (wave * WMT + i) * 8 + (lane >> 3);     // same (row & 7), hence the same slot
                    int ar = m0 + rowa; ar = ar < M ? ar : M - 1;
                    if constexpr (NSTG == 3) od_glds16_async(A + (size_t)ar * lda + k, sA + (wave * WMT + i) * 1024);
                    else od_glds16(A + (size_t)ar * lda + k, sA + (wave * WMT + i) * 1024 + lane * 16);
                }
            }
        };
        // asm DMA with hand-counted waits (the builtin form is waited for in front of the next LDS read): a wave has 4 + WMT pieces per tile
        if constexpr (NSTG == 3) {
            dma(0, 0);
            if (1 < nk) dma(1, 1);
            if (1 < nk) { if (WMT == 1) OD_WAIT_VMCNT(5); else if (WMT == 2) OD_WAIT_VMCNT(6); else OD_WAIT_VMCNT(8); } else OD_WAIT_VMCNT(0);       // tile 0 has landed
            od_barrier_raw();
            int buf = 0;
            for (int kt = 0; kt < nk; kt++) {
                const int nb = buf == 2 ? 0 : buf + 1, fb = nb == 2 ? 0 : nb + 1;      // stages of tiles kt + 1, kt + 2 (the latter was tile kt - 1's: free since the last barrier)
                if (kt + 2 < nk) dma(kt + 2, fb);
                compute_stage<T, WMT>(smem + buf * STG, smem + buf * STG + ASZ, wm, wn, lane, acc);
                if (kt + 2 < nk) { if (WMT == 1) OD_WAIT_VMCNT(5); else if (WMT == 2) OD_WAIT_VMCNT(6); else OD_WAIT_VMCNT(8); } else OD_WAIT_VMCNT(0);   // tile kt + 1 has landed
                od_barrier_raw();
                buf = nb;
            }
        } else {
            dma(0, 0);                                     // (the builtin DMA: an asm one with the wait behind the compute measured +1 % at two stages)
            __syncthreads();
            for (int kt = 0; kt < nk; kt++) {
                const int buf = kt & 1;
                if (kt + 1 < nk) dma(kt + 1, buf ^ 1);
                compute_stage<T, WMT>(smem + buf * STG, smem + buf * STG + ASZ, wm, wn, lane, acc);
                __syncthreads();
            }
        }
    } else {
    gload(0);
    lstore(0);
    __syncthreads();
    for (int kt = 0; kt < nk; kt++) {
        const int buf = kt & 1;
        if (kt + 1 < nk) gload(kt + 1);
        compute_stage<T, WMT>(smem + buf * STG, smem + buf * STG + ASZ, wm, wn, lane, acc);
        if (kt + 1 < nk) lstore(buf ^ 1);
        __syncthreads();
    }
    }

    // epilogue: accumulators -> LDS (f32 [BMT][128]) -> coalesced row stores
    float* sC = (float*)smem;
    {
        const int col = lane & 15, g = lane >> 4;
#pragma unroll
        for (int i = 0; i < WMT; i++)
#pragma unroll
            for (int j = 0; j < 4; j++)
#pragma unroll
                for (int r = 0; r < 4; r++)
                    sC[(wm * 16 * WMT + i * 16 + g * 4 + r) * 128 + wn * 64 + j * 16 + col] = acc[i][j][r];
    }
    __syncthreads();
    const bool vec_ok = (N % 8 == 0) && (ldc % 8 == 0);
    if constexpr (EPI == OD_EPI_QKROPE) {
        // (launcher guarantees N % 8 == 0, ldc % 8 == 0, n_rope % 128 == 0, hd in {32, 64}, bias != null)
        const int half = rp.hd >> 1, lph = rp.hd >> 3;          // lanes (8-column chunks) per head
        // a thread keeps its 8-column chunk (tid & 15) over the rows it walks: bias, norm weights and chunk geometry
        // are loaded once, only the frame's (cos, sin) row changes
        const int ch = tid & 15, gn = n0 + ch * 8;
        const bool roped = n0 < rp.n_rope;                      // block-uniform
        const int pc = ch ^ (lph >> 1);                         // chunk holding the rotary partners (d +- hd/2)
        const int d0 = (ch * 8) & (rp.hd - 1), dp = (pc * 8) & (rp.hd - 1);
        const bool lo = d0 < half;
        float bv[8], pb[8], wv[8], wp[8];
        od_ld8(bias + (gn < N ? gn : 0), bv);
        od_ld8(bias + (gn < N ? n0 + pc * 8 : 0), pb);
        {
            const float* w = gn < rp.dh ? rp.wq : rp.wk;
            od_ld8(w + d0, wv); od_ld8(w + dp, wp);
        }
        const float qs = gn < rp.dh ? rp.q_scale : 1.f;
#pragma unroll
        for (int i = 0; i < 2 * WMT; i++) {
            const int row = (tid + 256 * i) >> 4;
            const int gm = m0 + row;
            const bool valid = gm < M && gn < N;
            float v[8];
            od_ld8(sC + row * 128 + ch * 8, v);
#pragma unroll
            for (int e = 0; e < 8; e++) v[e] = od_round_to<T>(v[e] + bv[e]);   // what the unfused path reads back from qkv
            if (roped) {
                float t0[8], t1[8], pv[8];                        // the frame's 8 (cos, sin) pairs, partner values
                const float* tb = rp.table + ((size_t)((gm < M ? gm : 0) % rp.L) * half + (d0 & (half - 1))) * 2;
                od_ld8(tb, t0); od_ld8(tb + 8, t1);
                od_ld8(sC + row * 128 + pc * 8, pv);
                float ss = 0.f;
#pragma unroll
                for (int e = 0; e < 8; e++) ss += v[e] * v[e];
                for (int msk = 1; msk < lph; msk <<= 1) ss += __shfl_xor(ss, msk);
                const float invs = rsqrtf(ss / (float)rp.hd + rp.eps) * qs;
#pragma unroll
                for (int e = 0; e < 8; e++) {
                    const float y = v[e] * invs * wv[e];
                    const float yp = od_round_to<T>(pv[e] + pb[e]) * invs * wp[e];
                    const float cs = e < 4 ? t0[2 * e] : t1[2 * e - 8], sn = e < 4 ? t0[2 * e + 1] : t1[2 * e - 7];
                    v[e] = lo ? y * cs - yp * sn : yp * sn + y * cs;
                }
            }
            if (valid) od_st8(C + (size_t)gm * ldc + gn, v);
        }
        return;
    }
#pragma unroll
    for (int i = 0; i < 2 * WMT; i++) {
        const int c = tid + 256 * i, row = c >> 4, ch = c & 15;
        const int gm = m0 + row, gn = n0 + ch * 8;
        if (gm >= M || gn >= N) continue;
        float v[8];
        od_ld8(sC + row * 128 + ch * 8, v);
        T* dst = C + (size_t)gm * ldc + gn;
        if (vec_ok) {
            if (bias) {
                float bv[8]; od_ld8(bias + gn, bv);
#pragma unroll
                for (int e = 0; e < 8; e++) v[e] += bv[e];
            }
            if (EPI == OD_EPI_SILU) {
#pragma unroll
                for (int e = 0; e < 8; e++) v[e] = od_silu(v[e]);
            }
            if (accumulate) {
                float o[8]; od_ld8(dst, o);
#pragma unroll
                for (int e = 0; e < 8; e++) v[e] += o[e];
            }
            od_st8(dst, v);
        } else {
            for (int e = 0; e < 8 && gn + e < N; e++) {
                float x = v[e] + (bias ? bias[gn + e] : 0.f);
                if (EPI == OD_EPI_SILU) x = od_silu(x);
                if (accumulate) x += od_t<T>::ld(dst + e);
                od_t<T>::st(dst + e, x);
            }
        }
    }
}

// ---- TN: dW[n][k] += sum_m G[m][n] * A[m][k] over this block's M range ----------
// The reduction index m is the slow (row) dimension of both operands.  bf16: the slab is staged
// ROW-MAJOR ([m][128 cols], coalesced 16-byte writes) and the MFMA fragments — which need 8
// different m per lane — come from LDS transpose reads (ds_read_b64_tr_b16).  f32 (no 32-bit
// transpose read): the slab is transposed while it is written to LDS.
// Blocks of k-tile 0 also accumulate the column sums of G (the bias gradient) from the registers
// they stage, so G is not read a second time.
// ---- NT, large-M variant: 256x256 block tile, 8 waves (2 x 4) each owning 128 (m) x 64 (n) ----------
// Why: a 1-KiB LDS-DMA piece costs the issuing wave ~100+ cycles; at 128x128 a wave issues 8 pieces per
// 32 MFMAs.  Doubling both tile edges halves pieces per MFMA (8 per 64) and LDS fragment reads per MFMA
// (24 per 64 instead of 16 per 32).  128 KiB of dynamic LDS (2 stages x (A 32 KiB + W 32 KiB)), one
// workgroup per CU.  The MFMA is issued as (W rows) x (A rows)^T with the W-row permutation
//   pair p, half h, tile-row rho -> n = 32p + 8(rho>>2) + 4h + (rho&3)
// so a lane ends up holding 8 consecutive output columns of one output row: the epilogue is one 16-byte
// store per tile pair straight from the accumulators (no LDS round trip).
template <class T, int EPI>
__global__ __launch_bounds__(512, 1) void gemm_nt_big_kernel(const T* __restrict__ A, int lda, const T* __restrict__ W, int ldw,
                                                             const float* __restrict__ bias, T* __restrict__ C, int ldc,
                                                             int M, int N, int K, int accumulate, int nt_store, RopeEpi rp) {
    constexpr int TM = 256, TN = 256;
    constexpr int BK = 128 / (int)sizeof(T);
    constexpr int CH = 16 / (int)sizeof(T);
    constexpr int SLABS = BK / 32;
    constexpr int STG = 65536;                 // bytes per stage: A 32 KiB then W 32 KiB
    OD_DYN_SMEM(smem);
    const int tiles_m = (M + TM - 1) / TM, tiles_n = (N + TN - 1) / TN;
    int tm, tn;
    if (!tile_of_block(tiles_m, tiles_n, tm, tn)) return;
    const int m0 = tm * TM, n0 = tn * TN;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 2, wn = wave & 3;
    const int x = lane & 15, g = lane >> 4;

    f32x4 acc[4][8];           // [n tile i (pair i>>1, half i&1)][m tile j]
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 8; j++) acc[i][j] = (f32x4)(0.f);
    const int nk = K / BK;

    auto dma = [&](int kt, int buf) {
        unsigned char* st = smem + buf * STG;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const int piece = wave * 8 + i;                 // 0..31 -> A rows, 32..63 -> W rows
            const bool isw = piece >= 32;
            const int row = (piece & 31) * 8 + (lane >> 3);
            const int slot = (lane & 7) ^ (row & 7);
            const int k = kt * BK + slot * CH;
            int gr = (isw ? n0 : m0) + row;
            const int lim = isw ? N : M;
            gr = gr < lim ? gr : lim - 1;
            const T* src = isw ? (W + (size_t)gr * ldw + k) : (A + (size_t)gr * lda + k);
            od_glds16(src, st + piece * 1024 + lane * 16);
        }
    };
    auto wrow = [&](int i) { return wn * 64 + 32 * (i >> 1) + 8 * (x >> 2) + 4 * (i & 1) + (x & 3); };

    dma(0, 0);
    __syncthreads();
    for (int kt = 0; kt < nk; kt++) {
        const int buf = kt & 1;
        if (kt + 1 < nk) dma(kt + 1, buf ^ 1);
        const unsigned char* sA = smem + buf * STG;
        const unsigned char* sW = sA + 32768;
#pragma unroll
        for (int sl = 0; sl < SLABS; sl++) {
            od_frag<T> fw[4], fa[8];
#pragma unroll
            for (int i = 0; i < 4; i++) frag_w_from_lds<T>(fw[i], sW, wrow(i), sl, g);
#pragma unroll
            for (int j = 0; j < 8; j++) frag_from_lds<T>(fa[j], sA, wm * 128 + j * 16 + x, sl, g);
#pragma unroll
            for (int i = 0; i < 4; i++)
#pragma unroll
                for (int j = 0; j < 8; j++) acc[i][j] = od_mma(fw[i], fa[j], acc[i][j]);
        }
        __syncthreads();
    }
    if constexpr (EPI == OD_EPI_QKROPE) {
        // q/k RMSNorm + RoPE (attn.py:74-80) on the accumulators.  A wave's 64 columns are ONE head (hd = 64), a lane holds columns
        // 8g..8g+7 and 32+8g..32+8g+7 of it for row x: the rotary partner (d, d + 32) is in the same lane and the head's sum of
        // squares needs two shuffles (lanes x, x+16, x+32, x+48).  The pre-norm values are rounded to the tensor type first (what the
        // unfused path reads back) and, when a second output is given, stored to C for the backward.
        const int hc0 = n0 + wn * 64;
        const bool roped = hc0 < rp.n_rope, isq = hc0 < rp.dh;
        T* qk = (T*)rp.qk_out;
        float wv[2][8], bv[2][8];
        {
            const float* w = isq ? rp.wq : rp.wk;
            od_ld8(w + 8 * g, wv[0]); od_ld8(w + 32 + 8 * g, wv[1]);
            const int c0 = hc0 + 8 * g < N ? hc0 + 8 * g : 0, c1 = hc0 + 32 + 8 * g < N ? hc0 + 32 + 8 * g : 0;
            od_ld8(bias + c0, bv[0]); od_ld8(bias + c1, bv[1]);
        }
        const float qs = isq ? rp.q_scale : 1.f;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const int gm = m0 + wm * 128 + j * 16 + x;
            const bool valid = gm < M && hc0 < N;
            float v[2][8];
#pragma unroll
            for (int p = 0; p < 2; p++)
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    v[p][r] = od_round_to<T>(acc[2 * p][j][r] + bv[p][r]);
                    v[p][4 + r] = od_round_to<T>(acc[2 * p + 1][j][r] + bv[p][4 + r]);
                }
            T* crow = C + (size_t)(valid ? gm : 0) * ldc + hc0 + 8 * g;
            if (!roped || qk) {
                if (valid) {
                    if (nt_store) { od_st8_nt(crow, v[0]); od_st8_nt(crow + 32, v[1]); }
                    else { od_st8(crow, v[0]); od_st8(crow + 32, v[1]); }
                }
                if (!roped) continue;
            }
            float ss = 0.f;
#pragma unroll
            for (int e = 0; e < 8; e++) ss += v[0][e] * v[0][e] + v[1][e] * v[1][e];
            ss += __shfl_xor(ss, 16); ss += __shfl_xor(ss, 32);
            const float invs = rsqrtf(ss / 64.f + rp.eps) * qs;
            float t0[8], t1[8];                               // (cos, sin) of features 8g .. 8g+7 at this frame's position
            const float* tb = rp.table + ((size_t)((valid ? gm : 0) % rp.L) * 32 + 8 * g) * 2;
            od_ld8(tb, t0); od_ld8(tb + 8, t1);
            float o0[8], o1[8];
#pragma unroll
            for (int e = 0; e < 8; e++) {
                const float y0 = v[0][e] * invs * wv[0][e], y1 = v[1][e] * invs * wv[1][e];
                const float cs = e < 4 ? t0[2 * e] : t1[2 * e - 8], sn = e < 4 ? t0[2 * e + 1] : t1[2 * e - 7];
                o0[e] = y0 * cs - y1 * sn;
                o1[e] = y1 * cs + y0 * sn;
            }
            if (valid) {
                T* dst = qk ? qk + (size_t)gm * rp.ldqk + hc0 + 8 * g : crow;
                od_st8(dst, o0); od_st8(dst + 32, o1);
            }
        }
        return;
    }
    // lane (x, g): rows m0 + wm*128 + 16j + x, columns n0 + wn*64 + 32p + 8g .. +7
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const int gm = m0 + wm * 128 + j * 16 + x;
        if (gm >= M) continue;
#pragma unroll
        for (int p = 0; p < 2; p++) {
            const int gn = n0 + wn * 64 + 32 * p + 8 * g;
            if (gn >= N) continue;                          // N % 8 == 0 is required by the launcher
            float v[8];
#pragma unroll
            for (int r = 0; r < 4; r++) { v[r] = acc[2 * p][j][r]; v[4 + r] = acc[2 * p + 1][j][r]; }
            T* dst = C + (size_t)gm * ldc + gn;
            if (bias) {
                float bv[8]; od_ld8(bias + gn, bv);
#pragma unroll
                for (int e = 0; e < 8; e++) v[e] += bv[e];
            }
            if (EPI == OD_EPI_SILU) {
#pragma unroll
                for (int e = 0; e < 8; e++) v[e] = od_silu(v[e]);
            }
            if (accumulate) {
                float o[8]; od_ld8(dst, o);
#pragma unroll
                for (int e = 0; e < 8; e++) v[e] += o[e];
            }
            if (nt_store) od_st8_nt(dst, v); else od_st8(dst, v);
        }
    }
}

// ---- NT, large-M variant with FOUR waves (round 3): 256x256 block tile, one wave per SIMD, each wave a 128 x 128 sub-tile -----------
// The board runs these GEMMs at its power limit (1400 W, tools/power_probe.py): what decides the rate is how few clock cycles the work takes
// (the governor trades the cycles saved for a lower clock and voltage), i.e. how close the MFMA pipe is to always busy.  The vendor
// library's kernel for the long-K shapes — same macro tile, same fetch volume, 20-27 % faster than gemm_nt_big_kernel — gets there with a
// loop rather than a tile (its disassembly: 4 waves, 256 accumulator registers, fragments of two slabs in VGPRs, LDS-DMA, counted waits):
//   * 128 x 128 per wave: 16 fragment reads per 64 MFMAs instead of 12 per 32;
//   * one wave per SIMD with 64 independent accumulators issues MFMAs back to back; an MFMA leaves ~12 cycles of issue shadow, enough for ONE
//     cheap instruction — so every fragment read and every DMA instruction sits alone between two MFMAs (tools/ubench/barrier_cost.hip:
//     64 MFMAs 1044 cycles; + 16 ds_read_b128 1220; + 8 DMA pieces as s_mov m0 / MFMA / buffer_load 1244, as s_mov, s_nop, load 1340);
//   * the fetch must move whole 128-byte lines per DMA piece (tools/ubench/stream_cost.hip: pieces of 16 rows x 64 B, the natural shape
//     for a ring of 32-deep slab stages, run the same skeleton at 2095 cycles per slab instead of 1612 — the L1 fetches the line twice).
// This kernel is that loop in HIP.  Two 64-KiB stages (a 64-deep K tile each, 128-byte rows); per tile of 128 MFMAs and per wave:
//   MFMA   0.. 31   the tile's slab-1 fragments go to register set 1 (a read in front of every 2nd MFMA: R_AT / 16);
//   RELEASE barrier (the stage is in everybody's registers) at 32;
//   MFMA  33..123   the wave's 16 DMA pieces of tile kt + 2 into the released stage, one around every 6th MFMA;
//   LANDED  barrier at 88: vmcnt(10) — everything but this tile's first ten pieces, i.e. all of tile kt + 1 — then
//   MFMA  88..118   slab-0 fragments of tile kt + 1 from the other stage to register set 0 (every 2nd MFMA).
// A fetch has 0.7 - 1.4 tile times to land.  asm MFMAs with "a" constraints keep the 256 accumulators in AGPRs (the builtin form compiled
// to ~6 v_accvgpr copies per MFMA); LDS-DMA through a buffer descriptor (rows past M / N read as zero; past the last tile the
// descriptor has length 0: no fetch, same wait counts).  bf16, K % 128 == 0, N % 8 == 0.
#ifndef OD_W4_PIN
#define OD_W4_PIN 1
#endif
#ifndef OD_W4_R_AT
#define OD_W4_R_AT 32          // RELEASE barrier in front of this MFMA; the slab-1 reads sit in front of MFMAs 0, 2, .. below it
#endif
#ifndef OD_W4_RD1_BY
#define OD_W4_RD1_BY 24        // the slab-1 reads sit in front of MFMAs 0, 1, 3, 4, ... below this
#endif
#ifndef OD_W4_L_AT
#define OD_W4_L_AT 88          // LANDED barrier in front of this MFMA
#endif
#ifndef OD_W4_DMA_EVERY
#define OD_W4_DMA_EVERY 6
#endif
#ifndef OD_W4_X
#define OD_W4_X 0          // timing experiments only (wrong results): 2 no loop fragment reads, 4 no loop barriers, 8 no loop waits, 16 no fetch
#endif
#ifndef OD_W4_STAGGER
#define OD_W4_STAGGER 0
#endif
template <int EPI>
__global__ __launch_bounds__(256, 1) void gemm_nt_w4_kernel(const bf16_t* __restrict__ A, int lda, const bf16_t* __restrict__ W, int ldw,
                                                            const float* __restrict__ bias, bf16_t* __restrict__ C, int ldc,
                                                            int M, int N, int K, int nt_store, RopeEpi rp) {
    using T = bf16_t;
    constexpr int TM = 256, TN = 256, STG = 65536;       // stage: A 32 KiB (256 rows x 128 B) then W 32 KiB
    OD_DYN_SMEM(smem);
    // PERSISTENT: one workgroup per CU (grid = od_num_cus(), 256 on MI355X) walks output tiles v = 0, 1, ... — the tile a grid of gridDim.x * (v + 1) blocks would
    // give block blockIdx.x + gridDim.x * v under tile_of_block's XCD-aware order — and the operand pipeline runs across the tile
    // boundary: the last two K tiles of an output tile fetch the first two of the next one, whose first fragments are in registers when the
    // epilogue starts.  A cold start (fetch latency, ~2 us) and a drained pipeline per output tile cost the non-persistent form ~8 % at
    // K = 3072 and far more at K = 512.
    const int tiles_m = (M + TM - 1) / TM, tiles_n = (N + TN - 1) / TN;
    const int xcd = blockIdx.x & 7, slot0 = blockIdx.x >> 3, slots = gridDim.x >> 3;
    auto tile_at = [&](int v, int& m0_, int& n0_) {
        const int sl = slot0 + slots * v;
        const int tm = (sl / tiles_n) * 8 + xcd;
        m0_ = tm * TM; n0_ = (sl % tiles_n) * TN;
        return tm < tiles_m;
    };
    int m0, n0;
    if (!tile_at(0, m0, n0)) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = od_uniform(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int x = lane & 15, g = lane >> 4;
    const int nk = K / 64;
#if OD_W4_STAGGER && !defined(OD_EMU)
    // A/B (round 6): every second workgroup of an XCD starts OD_W4_STAGGER x ~4 us late, so that the store bursts of the epilogues (all
    // workgroups walk tiles of equal length) do not fall on top of each other
    if ((blockIdx.x >> 3) & 1) {
#pragma unroll 1
        for (int i = 0; i < OD_W4_STAGGER; i++) __builtin_amdgcn_s_sleep(127);
    }
#endif

    // The accumulators START at the bias: lane (x, g) holds, for output row m0 + wm*128 + 16 j + x, the columns n0 + wn*128 + 32 p + 8 g .. + 7
    // (acc[2p][j][0..3], acc[2p+1][j][0..3]) — the same eight bias values for every j.  (A bias load in the epilogue would sit behind the
    // epilogue's own stores in the VMEM counter and wait for every one of them.)
    f32x4 acc[8][8];           // [n tile i (pair i>>1, half i&1)][m tile j]
    auto load_bias = [&](int n0_, f32x4 (&bv)[8]) {
#pragma unroll
        for (int p = 0; p < 4; p++) {
            const int gn = n0_ + wn * 128 + 32 * p + 8 * g;
            float t[8];
#pragma unroll
            for (int e = 0; e < 8; e++) t[e] = 0.f;
            if (bias && gn < N) od_ld8(bias + gn, t);
#pragma unroll
            for (int r = 0; r < 4; r++) { bv[2 * p][r] = t[r]; bv[2 * p + 1][r] = t[4 + r]; }
        }
    };
    {
        f32x4 bv[8];
        load_bias(n0, bv);
#pragma unroll
        for (int i = 0; i < 8; i++)
#pragma unroll
            for (int j = 0; j < 8; j++) acc[i][j] = bv[i];
    }

    // staging: waves 0, 1 stream the A tile (32 pieces of 8 rows x 128 B), waves 2, 3 the W tile; 16 pieces per wave and tile.
    // LDS rows are 128 B with the 16-byte slots XOR-swizzled by a per-row key.  A rows: key = row & 7 (a fragment read covers 16 consecutive
    // rows).  W rows: a fragment covers rows 8 (x >> 2) + 4 h + (x & 3) — under row & 7 the lanes x and x + 12 (and x + 4, x + 8) of one
    // ds_read_b128 lane group ({0-3, 12-15, 20-27}, ...) meet in the same banks, a 2-way conflict on every W read — so the key takes row bits
    // 1, 3, 4 instead: ((row >> 1) & 1) | (((row >> 3) & 3) << 1), which makes each group's 16 lanes cover all 64 banks once.
    const bool isw = wave >= 2;
    const int ld = isw ? ldw : lda, lim = isw ? N : M;
    auto srd_of = [&](int m0_, int n0_, bool valid) {                       // this wave's operand rows of the tile at (m0_, n0_)
        const int r0 = isw ? n0_ : m0_;
        const int rows = lim - r0 < 256 ? lim - r0 : 256;                   // the descriptor covers this tile's rows only: its 32-bit length
        const long avail = (long)(rows - 1) * ld + K;                       // never sees the size of the whole matrix; it ends with the last valid row
        return od_make_srd((isw ? W : A) + (size_t)r0 * ld, valid && rows > 0 ? (unsigned)(avail * 2) : 0u);
    };
    od_srd_t srd_cur = srd_of(m0, n0, true), srd_nxt = srd_cur, srd = srd_cur;
    const int prow = lane >> 3;
    unsigned voff4[4];                                                       // by piece & 3 (the W key depends on it)
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const int key = isw ? (((prow >> 1) & 1) | (q << 1)) : prow;
        voff4[q] = (unsigned)((((wave & 1) * 16 + q) * 8 + prow) * ld * 2 + (((lane & 7) ^ key) << 4));
    }
    const unsigned lds_mine = od_lds_addr(smem) + (isw ? 32768u : 0u) + (unsigned)(wave & 1) * 16384u;
    const unsigned piece_stride = (unsigned)(8 * ld * 2);
    // piece i of this wave's 16: rows ((wave & 1) * 16 + i) * 8 ..., i = 4 t + q -> voff4[q], t * 4 pieces further down
    int offA[2], offW[2];
    {
        const int wkey = ((x >> 1) & 1) | ((x >> 2) << 1);
        const int wrow = wn * 128 + 8 * (x >> 2) + (x & 3);
#pragma unroll
        for (int sl = 0; sl < 2; sl++) {
            offA[sl] = (wm * 128 + x) * 128 + (((sl * 4 + g) ^ (x & 7)) << 4);          // + j * 2048
            offW[sl] = 32768 + wrow * 128 + (((sl * 4 + g) ^ wkey) << 4);              // + (i & 1) * 512 + (i >> 1) * 4096
        }
    }
    od_frag<T> fa[2][8], fw[2][8];                        // [slab = register set][tile]
    auto rdA = [&](const unsigned char* st, int sl, int j) { fa[sl][j].v = *(const s16x8*)(st + offA[sl] + j * 2048); };
    auto rdW = [&](const unsigned char* st, int sl, int i) { fw[sl][i].v = *(const s16x8*)(st + offW[sl] + (i & 1) * 512 + (i >> 1) * 4096); };
    // MFMA n of a slab: eight consecutive ones share the W fragment and walk the A-matrix fragments; the reads come in the order of first use
    auto mma_one = [&](int sl, int n) {
        const int i = n >> 3, j = n & 7;
#if defined(OD_EMU)
        acc[i][j] = od_mma(fw[sl][i], fa[sl][j], acc[i][j]);
#else
        // an accumulator is reused 64 MFMAs later: no back-to-back dependency
        asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[i][j]) : "v"(fw[sl][i].v), "v"(fa[sl][j].v));
#endif
    };
    auto rd_seq = [&](const unsigned char* st, int sl, int r) {
        if (r == 0) rdW(st, sl, 0); else if (r < 9) rdA(st, sl, r - 1); else rdW(st, sl, r - 8);
    };
#if OD_W4_PIN
#define W4_FENCE() __builtin_amdgcn_sched_barrier(0)
#else
#define W4_FENCE() ((void)0)
#endif

    // prologue: tiles 0 and 1 in flight, tile 0 landed, its slab-0 fragments in register set 0
#pragma unroll
    for (int t = 0; t < 2; t++) {                          // nk >= 2
#pragma unroll
        for (int i = 0; i < 16; i++)
            od_buffer_lds16_at(srd_cur, voff4[i & 3], (unsigned)t * 128u + (unsigned)(i >> 2) * 4u * piece_stride, lds_mine + (unsigned)t * STG + (unsigned)i * 1024u);
    }
    OD_WAIT_VMCNT(16);
    od_barrier_raw();
#pragma unroll
    for (int r = 0; r < 16; r++) rd_seq(smem, 0, r);

    constexpr int DMA0 = OD_W4_R_AT + 1;                                        // first DMA slot
    constexpr int BEFORE_L = (OD_W4_L_AT - DMA0 + OD_W4_DMA_EVERY - 1) / OD_W4_DMA_EVERY;   // pieces issued in front of the LANDED barrier
    static_assert(OD_W4_RD1_BY >= 24 && OD_W4_RD1_BY <= OD_W4_R_AT && DMA0 + 15 * OD_W4_DMA_EVERY < 128 && OD_W4_L_AT + 30 < 128, "schedule does not fit the tile");
    auto tile = [&](int kt, const int xs) {
        const unsigned char* X = smem + xs * STG;
        const unsigned char* Y = smem + (xs ^ 1) * STG;
        const unsigned dst = lds_mine + (unsigned)xs * STG;
        const bool wrap = kt + 2 >= nk;                    // the fetch belongs to the next output tile (or to nothing: a descriptor of length 0)
        const unsigned so = (unsigned)(wrap ? kt + 2 - nk : kt + 2) * 128u;
#pragma clang loop unroll(full)
        for (int n = 0; n < 128; n++) {
            if (n == OD_W4_R_AT) {
                if (!(OD_W4_X & 8)) OD_WAIT_LGKMCNT(0);
                if (!(OD_W4_X & 4)) od_barrier_raw();
                srd = wrap ? srd_nxt : srd_cur;
                if (OD_W4_X & 16) od_srd_set_bytes(srd, 0u);
            }
            if (n == OD_W4_L_AT) {
                if (!(OD_W4_X & 8)) {
#define W4_VM(c) else if (BEFORE_L == c) OD_WAIT_VMCNT(c)
                    if (BEFORE_L >= 16) OD_WAIT_VMCNT(16);
                    W4_VM(15); W4_VM(14); W4_VM(13); W4_VM(12); W4_VM(11); W4_VM(10); W4_VM(9); W4_VM(8); W4_VM(7); W4_VM(6); W4_VM(5); W4_VM(4);
                    else OD_WAIT_VMCNT(0);
#undef W4_VM
                }
                if (!(OD_W4_X & 4)) od_barrier_raw();
            }
            const bool d = n >= DMA0 && (n - DMA0) % OD_W4_DMA_EVERY == 0 && (n - DMA0) / OD_W4_DMA_EVERY < 16;
            const int q = (n - DMA0) / OD_W4_DMA_EVERY;
            if (d) od_dma_set_dst(dst + (unsigned)q * 1024u);
            if (!(OD_W4_X & 2)) {
                // 16 reads in front of MFMAs 0, 1, 3, 4, 6, ... (two per three) of the first OD_W4_RD1_BY: the last one is well ahead of the barrier
                if (n < OD_W4_RD1_BY && (n % 3 != 2) && (n / 3) * 2 + n % 3 < 16) rd_seq(X, 1, (n / 3) * 2 + n % 3);
                if (n >= OD_W4_L_AT && (n - OD_W4_L_AT) % 2 == 0 && (n - OD_W4_L_AT) / 2 < 16) rd_seq(Y, 0, (n - OD_W4_L_AT) / 2);
            }
            mma_one(n >> 6, n & 63);
            if (d) od_buffer_lds16_m0(srd, voff4[q & 3], so + (unsigned)(q >> 2) * 4u * piece_stride);
            W4_FENCE();
        }
    };
    for (int v = 0;; v++) {
        int m1, n1;
        const bool more = tile_at(v + 1, m1, n1);
        srd_nxt = srd_of(more ? m1 : 0, more ? n1 : 0, more);
        for (int kt = 0; kt < nk; kt += 2) {               // K % 128 == 0 (launcher)
            tile(kt, 0);
            tile(kt + 1, 1);
        }
#if !defined(OD_EMU)
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");   // the last MFMAs' results before the epilogue reads the accumulators (asm MFMAs are invisible to the hazard pass)
#endif
        // epilogue: the next tile's bias first (a load issued after the stores would wait for them), then one 16-byte store per tile pair
        // straight from the accumulators, which restart at that bias
        f32x4 bv[8];
        load_bias(more ? n1 : n0, bv);
        if constexpr (EPI == OD_EPI_QKROPE) {
            // q/k RMSNorm + RoPE (attn.py:74-80) on the accumulators, as in gemm_nt_big_kernel's epilogue: the wave's 128 columns are TWO heads
            // (hd = 64); of head hq a lane holds columns 8g..8g+7 (acc[4hq], acc[4hq+1]) and 32+8g..+7 (acc[4hq+2], acc[4hq+3]) for row
            // 16 j + x: the rotary partner (d, d + 32) is in the same lane, the head's sum of squares takes two shuffles.  The values are
            // rounded to the tensor type first (what the unfused path reads back); with a second output (training) C keeps them.
            T* qk = (T*)rp.qk_out;
            const bool c_nt = nt_store;
            // rows outer, the wave's two heads inner: the (cos, sin) row of a frame is loaded once for both
            float wv[2][2][8];
            bool roped2[2], isq2[2];
#pragma unroll
            for (int hq = 0; hq < 2; hq++) {
                const int hc0 = n0 + wn * 128 + 64 * hq;
                roped2[hq] = hc0 < rp.n_rope; isq2[hq] = hc0 < rp.dh;
                const float* w = isq2[hq] ? rp.wq : rp.wk;
                od_ld8(w + 8 * g, wv[hq][0]); od_ld8(w + 32 + 8 * g, wv[hq][1]);
            }
            const bool any_roped = roped2[0] || roped2[1];
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const int gm = m0 + wm * 128 + j * 16 + x;
                float t0[8], t1[8];                           // (cos, sin) of features 8g .. 8g+7 at this frame's position
                if (any_roped) {
                    const float* tb = rp.table + ((size_t)((gm < M ? gm : 0) % rp.L) * 32 + 8 * g) * 2;
                    od_ld8(tb, t0); od_ld8(tb + 8, t1);
                }
#pragma unroll
                for (int hq = 0; hq < 2; hq++) {
                    const int hc0 = n0 + wn * 128 + 64 * hq;
                    const bool roped = roped2[hq];
                    const bool valid = gm < M && hc0 < N;
                    float v[2][8], raw[2][8];
#pragma unroll
                    for (int p = 0; p < 2; p++) {
#pragma unroll
                        for (int r = 0; r < 4; r++) {
                            raw[p][r] = acc[4 * hq + 2 * p][j][r]; raw[p][4 + r] = acc[4 * hq + 2 * p + 1][j][r];
                            v[p][r] = od_round_to<T>(raw[p][r]);
                            v[p][4 + r] = od_round_to<T>(raw[p][4 + r]);
                        }
                        acc[4 * hq + 2 * p][j] = bv[4 * hq + 2 * p]; acc[4 * hq + 2 * p + 1][j] = bv[4 * hq + 2 * p + 1];
                    }
#ifndef OD_W4Q_LINE_STORES
#define OD_W4Q_LINE_STORES 0     // 1: whole 128-byte lines here too (od_store_line_pair) — measured SLOWER in this VALU-heavy epilogue (1.33 against 1.25 ms)
#endif
                    const int row16 = m0 + wm * 128 + j * 16;
                    const bool cols_ok = hc0 < N;
                    T* crow = C + (size_t)(valid ? gm : 0) * ldc + hc0 + 8 * g;
                    if (!roped && rp.f16) {                   // v as IEEE half, straight from the accumulators (no bf16 rounding in between)
                        if (OD_W4Q_LINE_STORES) od_store_line_pair<f16_t, false>((f16_t*)C + hc0, (size_t)ldc, row16, x, g, M, cols_ok, raw[0], raw[1]);
                        else if (valid) { od_st8((f16_t*)crow, raw[0]); od_st8((f16_t*)crow + 32, raw[1]); }
                        continue;
                    }
                    if (!roped || qk) {
                        if (OD_W4Q_LINE_STORES) {
                            if (c_nt) od_store_line_pair<T, true>(C + hc0, (size_t)ldc, row16, x, g, M, cols_ok, v[0], v[1]);
                            else od_store_line_pair<T, false>(C + hc0, (size_t)ldc, row16, x, g, M, cols_ok, v[0], v[1]);
                        } else if (valid) {
                            if (c_nt) { od_st8_nt(crow, v[0]); od_st8_nt(crow + 32, v[1]); }
                            else { od_st8(crow, v[0]); od_st8(crow + 32, v[1]); }
                        }
                        if (!roped) continue;
                    }
                    float ss = 0.f;
#pragma unroll
                    for (int e = 0; e < 8; e++) ss += v[0][e] * v[0][e] + v[1][e] * v[1][e];
                    ss += __shfl_xor(ss, 16); ss += __shfl_xor(ss, 32);
                    const float invs = rsqrtf(ss / 64.f + rp.eps) * (isq2[hq] ? rp.q_scale : 1.f);
                    float o0[8], o1[8];
#pragma unroll
                    for (int e = 0; e < 8; e++) {
                        const float y0 = v[0][e] * invs * wv[hq][0][e], y1 = v[1][e] * invs * wv[hq][1][e];
                        const float cs = e < 4 ? t0[2 * e] : t1[2 * e - 8], sn = e < 4 ? t0[2 * e + 1] : t1[2 * e - 7];
                        o0[e] = y0 * cs - y1 * sn;
                        o1[e] = y1 * cs + y0 * sn;
                    }
                    // (stored non-temporally the q, k stream takes a third off the kernel's fabric reads and nothing off the step:
                    // profiles/r05b_qkrope_fetch.txt)
                    if (OD_W4Q_LINE_STORES) {
                        if (rp.f16 && qk) od_store_line_pair<f16_t, false>((f16_t*)qk + hc0, (size_t)rp.ldqk, row16, x, g, M, cols_ok, o0, o1);
                        else if (qk) od_store_line_pair<T, false>(qk + hc0, (size_t)rp.ldqk, row16, x, g, M, cols_ok, o0, o1);
                        else od_store_line_pair<T, false>(C + hc0, (size_t)ldc, row16, x, g, M, cols_ok, o0, o1);
                    } else if (valid) {
                        T* dst = qk ? qk + (size_t)gm * rp.ldqk + hc0 + 8 * g : crow;
                        if (rp.f16 && qk) { od_st8((f16_t*)dst, o0); od_st8((f16_t*)dst + 32, o1); }
                        else { od_st8(dst, o0); od_st8(dst + 32, o1); }
                    }
                }
            }
        } else {
#ifndef OD_W4_LINE_STORES
#define OD_W4_LINE_STORES 1     // 0: the round-3 epilogue (two 64-byte row segments per line, from two store instructions); A/B
#endif
#if OD_W4_LINE_STORES
        // Whole 128-byte lines per store instruction.  A lane holds, of row 16 j + x, the columns 32 p + 8 g .. + 7 (16 bytes): the four g lanes
        // of a row cover 64 bytes, and the second half of that line belongs to p + 1 — another instruction, written some microseconds later, so
        // the memory system saw two partial-line writes per line (every K = 512 product wrote at the same 1.85 TB/s: profiles/r05_ab_records.txt).
        // Here the two halves of the lane rows trade places (DPP row_shr / row_shl by 8): instruction one writes rows 0..7 of the 16, columns
        // 64 pp .. + 63 — lanes x < 8 their own p = 2 pp piece, lanes x >= 8 the p = 2 pp + 1 piece of the lane 8 below —, instruction two rows 8..15.
        const int xr = x & 7, xh = x >> 3;
#pragma unroll
        for (int j = 0; j < 8; j++) {
#pragma unroll
            for (int pp = 0; pp < 2; pp++) {
                u32x4 ra, rb;                                  // this lane's p = 2 pp and p = 2 pp + 1 pieces, packed
#pragma unroll
                for (int h2 = 0; h2 < 2; h2++) {
                    const int p = 2 * pp + h2;
                    float v8[8];
#pragma unroll
                    for (int r = 0; r < 4; r++) { v8[r] = acc[2 * p][j][r]; v8[4 + r] = acc[2 * p + 1][j][r]; }
                    acc[2 * p][j] = bv[2 * p]; acc[2 * p + 1][j] = bv[2 * p + 1];
                    if (EPI == OD_EPI_SILU) {
#pragma unroll
                        for (int e = 0; e < 8; e++) v8[e] = od_silu(v8[e]);
                    }
                    u32x4& rr = h2 ? rb : ra;
#pragma unroll
                    for (int i = 0; i < 4; i++) rr[i] = od_pack_bf2(v8[2 * i], v8[2 * i + 1]);
                }
                u32x4 lo, hi;                                  // rows 0..7 / rows 8..15 of this 16-row tile
#pragma unroll
                for (int i = 0; i < 4; i++) { lo[i] = od_dpp_up8(ra[i], rb[i]); hi[i] = od_dpp_down8(ra[i], rb[i]); }
                const int gn = n0 + wn * 128 + 64 * pp + 32 * xh + 8 * g;         // lanes x >= 8 sit in the second half of the line
                const int gm_lo = m0 + wm * 128 + j * 16 + xr, gm_hi = gm_lo + 8;
                // rows 8..15: lanes x >= 8 write their OWN p = 2 pp piece at the line's first half, lanes x < 8 the partner's p = 2 pp + 1 piece
                const int gn_hi = n0 + wn * 128 + 64 * pp + 32 * (1 - xh) + 8 * g;
                // (OD_W4_X & 32, timing only: every second workgroup of an XCD keeps its results in registers — does a CU's store phase get
                // shorter when half the chip is silent?  profiles/r06d_nt_store_contention.txt)
                const bool silent = (OD_W4_X & 32) && ((blockIdx.x >> 3) & 1);
                if (silent) { asm volatile("" :: "v"(lo), "v"(hi)); continue; }
                if (gn < N && gm_lo < M) {
                    T* dst = C + (size_t)gm_lo * ldc + gn;
                    if (nt_store) od_st16_nt(dst, lo); else *(u32x4*)dst = lo;
                }
                if (gn_hi < N && gm_hi < M) {
                    T* dst = C + (size_t)gm_hi * ldc + gn_hi;
                    if (nt_store) od_st16_nt(dst, hi); else *(u32x4*)dst = hi;
                }
            }
        }
#else
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const int gm = m0 + wm * 128 + j * 16 + x;
#pragma unroll
            for (int p = 0; p < 4; p++) {
                const int gn = n0 + wn * 128 + 32 * p + 8 * g;
                float v8[8];
#pragma unroll
                for (int r = 0; r < 4; r++) { v8[r] = acc[2 * p][j][r]; v8[4 + r] = acc[2 * p + 1][j][r]; }
                acc[2 * p][j] = bv[2 * p]; acc[2 * p + 1][j] = bv[2 * p + 1];
                if (gm >= M || gn >= N) continue;
                T* dst = C + (size_t)gm * ldc + gn;
                if (EPI == OD_EPI_SILU) {
#pragma unroll
                    for (int e = 0; e < 8; e++) v8[e] = od_silu(v8[e]);
                }
                if (nt_store) od_st8_nt(dst, v8); else od_st8(dst, v8);
            }
        }
#endif
        }
        if (!more) break;
#if !defined(OD_EMU)
        asm volatile("s_nop 7" ::: "memory");               // accumulator writes (zeroing) before the next asm MFMA reads them
#endif
        m0 = m1; n0 = n1; srd_cur = srd_nxt;
    }
#undef W4_FENCE
    OD_WAIT_VMCNT(0);
}

// 256-byte-row tile addressing for the TN slabs: XOR at 32-byte granularity (slot PAIRS), because a
// transpose read touches two adjacent 16-byte slots of 8 different rows per 32-lane group.
__device__ __forceinline__ int tn_off(int row, int byte) {
    return row * 256 + ((((byte >> 5)) ^ (row & 7)) << 5) + (byte & 31);
}
__device__ __forceinline__ void tn_frag(od_frag<bf16_t>& f, const unsigned char* t, int c0, int x, int u, int g) {
    const int cb = (c0 + 4 * (x & 3)) * 2, rr = 32 * u + 4 * g + (x >> 2);
    const s16x4 a = od_lds_tr_read((const bf16_t*)(t + tn_off(rr, cb)));
    const s16x4 b = od_lds_tr_read((const bf16_t*)(t + tn_off(rr + 16, cb)));
    f.v[0] = a[0]; f.v[1] = a[1]; f.v[2] = a[2]; f.v[3] = a[3];
    f.v[4] = b[0]; f.v[5] = b[1]; f.v[6] = b[2]; f.v[7] = b[3];
}
__device__ __forceinline__ void tn_frag(od_frag<float>&, const unsigned char*, int, int, int, int) {}

template <class T>
__global__ __launch_bounds__(256) void gemm_tn_kernel(const T* __restrict__ G, int ldg, const T* __restrict__ A, int lda,
                                                      float* __restrict__ dW, int lddw, float* __restrict__ dbias,
                                                      int M, int N, int K, int m_per_block, const OdDetTable* __restrict__ det, TnRowMap rm) {
    constexpr bool TR = sizeof(T) == 2;
    constexpr int BR = 128 / (int)sizeof(T);  // reduction rows per slab (64 bf16 / 32 f32)
    constexpr int CH = 16 / (int)sizeof(T);
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * STAGE_BYTES];
    __shared__ float sred[128];
    __shared__ long long sfix[128];        // the f32 path's bias sums: many threads per column -> fixed point (order-free)
    __shared__ int s_bad;
    const int tiles_n = (N + BN - 1) / BN, tiles_k = (K + BM - 1) / BM;
    const int tile = blockIdx.x % (tiles_n * tiles_k), split = blockIdx.x / (tiles_n * tiles_k);
    const int n0 = (tile / tiles_k) * BN, k0 = (tile % tiles_k) * BM;
    const int mb = split * m_per_block;
    int me = mb + m_per_block; me = me < M ? me : M;
    if (mb >= M) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const bool do_bias = dbias != nullptr && (tile % tiles_k) == 0;
    if (tid < 128) { sred[tid] = 0.f; sfix[tid] = 0; }
    if (tid == 0) s_bad = 0;

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) acc[i][j] = (f32x4)(0.f);

    constexpr int CPR = 128 / CH;              // 16-byte chunks per slab row (16 bf16 / 32 f32)
    constexpr int NCH = BR * CPR / 256;        // chunks per thread per operand (4)
    const int nslab = (me - mb + BR - 1) / BR;

    if constexpr (TR) {
        // ---- bf16: LDS-DMA of row-major slabs (out-of-range rows / columns read a global zero), MFMA
        //      fragments by transpose reads.  The bias gradient is summed from the staged G slab in LDS.
        // Buffer-addressed LDS-DMA from inline asm (the builtin form makes hipcc wait vmcnt(0) before the first transpose read of
        // every slab, serialising the prefetch with the MFMAs — see gemm_tn_big_kernel).  Each wave stages 4 pieces (4 rows x 256 B)
        // of the G slab and 4 of the A slab; rows past this block's M range read as zero (the descriptors end at row `me`), columns
        // past N / K inside a row bring in neighbouring data that only reaches outputs the epilogue drops.
        const int uw = od_uniform(wave);
        // the descriptors end after the last valid column of row me-1, rounded UP to a whole 16-byte chunk (ld is a multiple of 8 elements, so
        // the chunk lies inside the row): the hardware checks the range per DWORD, and an odd width (Hf = 1365) would otherwise lose the last
        // column of that row
        const long availg = (long)(me - mb - 1) * ldg + ((N - n0 + 7) & ~7), availa = (long)(me - mb - 1) * lda + ((K - k0 + 7) & ~7);
        const od_srd_t srdg = od_make_srd(G + (size_t)mb * ldg + n0, (unsigned)((availg > 0 ? availg : 0) * 2));
        const od_srd_t srda = od_make_srd(A + (size_t)mb * lda + k0, (unsigned)((availa > 0 ? availa : 0) * 2));
        unsigned vg[4], va[4];
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int r = (uw * 4 + i) * 4 + (lane >> 4);     // row of the 64-row slab
            const int pos = lane & 15;                         // 16-byte position within the LDS row
            const int slot = ((((pos >> 1) ^ (r & 7)) << 1) | (pos & 1));   // logical column chunk stored there
            vg[i] = (unsigned)(r * ldg * 2 + slot * 16);
            va[i] = (unsigned)(r * lda * 2 + slot * 16);
        }
        const unsigned lds_mine = od_lds_addr(smem) + (unsigned)uw * 4096u;
        auto dma = [&](int st, int buf) {
            const unsigned dst = lds_mine + (unsigned)buf * STAGE_BYTES;
            const unsigned sg = (unsigned)st * BR * (unsigned)ldg * 2u, sa = (unsigned)st * BR * (unsigned)lda * 2u;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                od_buffer_lds16_at(srdg, vg[i], sg, dst + i * 1024u);
                od_buffer_lds16_at(srda, va[i], sa, dst + 16384u + i * 1024u);
            }
        };
        float bsum = 0.f;                                    // thread t < 128 owns column n0 + t
        dma(0, 0);
        OD_WAIT_VMCNT(0);
        __syncthreads();
        const int x = lane & 15, g = lane >> 4;
        for (int st = 0; st < nslab; st++) {
            const int buf = st & 1;
            if (st + 1 < nslab) dma(st + 1, buf ^ 1);
            const unsigned char* sA = smem + buf * STAGE_BYTES;
            const unsigned char* sB = sA + 16384;
#pragma unroll
            for (int u = 0; u < 2; u++) {
                od_frag<T> fa[4], fb[4];
#pragma unroll
                for (int i = 0; i < 4; i++) tn_frag(fa[i], sA, wm * 64 + i * 16, x, u, g);
#pragma unroll
                for (int j = 0; j < 4; j++) tn_frag(fb[j], sB, wn * 64 + j * 16, x, u, g);
#pragma unroll
                for (int i = 0; i < 4; i++)
#pragma unroll
                    for (int j = 0; j < 4; j++) acc[i][j] = od_mma(fa[i], fb[j], acc[i][j]);
            }
            if (do_bias) {
                const int col = tid & 127, half = tid >> 7;
#pragma unroll 8
                for (int r = 0; r < 32; r++)
                    bsum += od_bf2f(*(const bf16_t*)(sA + tn_off(half * 32 + r, col * 2)));
            }
            OD_WAIT_VMCNT(0);
            __syncthreads();
        }
        if (do_bias) atomicAdd(&sred[tid & 127], bsum);
    } else {
        // ---- f32: register staging, transposing while writing to LDS ([col][m], 128-byte rows)
        u32x4 rg[NCH], ra[NCH];
        float bsum[CH];
#pragma unroll
        for (int e = 0; e < CH; e++) bsum[e] = 0.f;
        auto gload = [&](int st) {
#pragma unroll
            for (int i = 0; i < NCH; i++) {
                const int c = tid + 256 * i, r = c / CPR, cc = (c % CPR) * CH;
                const int m = mb + st * BR + r;
                const bool mv = m < me;
                rg[i] = (mv && n0 + cc < N) ? *(const u32x4*)(G + (size_t)m * ldg + n0 + cc) : (u32x4)(0u);
                ra[i] = (mv && k0 + cc < K) ? *(const u32x4*)(A + (size_t)m * lda + k0 + cc) : (u32x4)(0u);
            }
            if (do_bias) {
#pragma unroll
                for (int i = 0; i < NCH; i++) {
                    const T* pg = (const T*)&rg[i];
#pragma unroll
                    for (int e = 0; e < CH; e++) bsum[e] += od_t<T>::ld(pg + e);
                }
            }
        };
        auto lstore = [&](int buf) {
            unsigned char* sA = smem + buf * STAGE_BYTES;
            unsigned char* sB = sA + 16384;
#pragma unroll
            for (int i = 0; i < NCH; i++) {
                const int c = tid + 256 * i, r = c / CPR, cc = (c % CPR) * CH;
                const T* pg = (const T*)&rg[i];
                const T* pa = (const T*)&ra[i];
#pragma unroll
                for (int e = 0; e < CH; e++) {
                    *(T*)(sA + tile_off<128>(cc + e, r * (int)sizeof(T))) = pg[e];
                    *(T*)(sB + tile_off<128>(cc + e, r * (int)sizeof(T))) = pa[e];
                }
            }
        };
        gload(0);
        lstore(0);
        __syncthreads();
        for (int st = 0; st < nslab; st++) {
            const int buf = st & 1;
            if (st + 1 < nslab) gload(st + 1);
            compute_stage<T>(smem + buf * STAGE_BYTES, smem + buf * STAGE_BYTES + 16384, wm, wn, lane, acc);
            if (st + 1 < nslab) lstore(buf ^ 1);
            __syncthreads();
        }
        if (do_bias) {
            const int cc = (tid % CPR) * CH;
#pragma unroll
            for (int e = 0; e < CH; e++) od_lds_fix_add(&sfix[cc + e], bsum[e], &s_bad);
            __syncthreads();
            if (tid < 128) sred[tid] = od_lds_unfix(sfix[tid], s_bad);
        }
    }
    const int col = lane & 15, g2 = lane >> 4;
    long long* const dw_shadow = od_det_find(det, dW);
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int n = n0 + wm * 64 + i * 16 + g2 * 4 + r, k = k0 + wn * 64 + j * 16 + col;
                int no;
                if (tn_map_row(rm, n, N, no) && k < K) od_red_add_at(dw_shadow, dW, (size_t)no * lddw + k, acc[i][j][r]);
            }
    if (do_bias) {
        __syncthreads();
        int no;
        if (tid < 128 && tn_map_row(rm, n0 + tid, N, no)) od_red_add(det, dbias + no, sred[tid]);       // (the bf16 path: two contributions per column — a + b = b + a)
    }
}

// ---- TN, large variant (bf16): 256 (n) x 256 (k) output tile, reduction slabs of 64 rows staged row-major by LDS-DMA (512-byte rows,
// XOR swizzle at 32-byte granularity over the row's 16 slot pairs), fragments by transpose reads.  (Rounds 1-4 ran it with 8 waves of
// 128 x 64 — gemm_tn_big_kernel, 0.76-0.89 PF/s, removed in round 5; its tile -> (n0, k0, M-split) maps live on in the kernel below.)
__device__ __forceinline__ int tn512_off(int row, int byte) {
    return row * 512 + ((((byte >> 5)) ^ (row & 15)) << 5) + (byte & 31);
}
__device__ __forceinline__ void tn512_frag(od_frag<bf16_t>& f, const unsigned char* t, int c0, int x, int u, int g) {
    const int cb = (c0 + 4 * (x & 3)) * 2, rr = 32 * u + 4 * g + (x >> 2);
    const s16x4 a = od_lds_tr_read((const bf16_t*)(t + tn512_off(rr, cb)));
    const s16x4 b = od_lds_tr_read((const bf16_t*)(t + tn512_off(rr + 16, cb)));
    f.v[0] = a[0]; f.v[1] = a[1]; f.v[2] = a[2]; f.v[3] = a[3];
    f.v[4] = b[0]; f.v[5] = b[1]; f.v[6] = b[2]; f.v[7] = b[3];
}
// ---- TN, large variant, FOUR waves (round 5): gemm_nt_w4_kernel's organisation for the weight-gradient product.  One wave per SIMD, 128 (n) x 128 (k)
// per wave with the 256 accumulators in AGPRs behind asm MFMAs, the fragments of the two 32-row halves of a 64-row slab in two register
// sets, every transpose read / DMA piece alone between two MFMAs, two barriers per slab:
//   MFMA   0.. 31   half 1's 32 transpose reads (one in front of every MFMA) -> register set 1;  [bias tiles: the slab's column sums]
//   RELEASE barrier at 32 (the stage is in everybody's registers)
//   MFMA  33.. 85   the wave's 16 DMA pieces of slab st + 2 into the released stage
//   LANDED  barrier at 88: vmcnt(16) = everything but those 16 pieces, i.e. all of slab st + 1
//   MFMA  88..119   half 0 of slab st + 1 from the other stage -> register set 0
// against the retired 8-wave kernel's read-everything / 32-MFMAs / barrier per half (0.75 transpose reads per MFMA; here 0.5): +12-17 % on the
// step's four shapes (profiles/r05_ab_records.txt).  Tile -> (n0, k0, M-split) maps (block b runs on XCD b % 8):
//   packed (xcd_order 2): the (split, tile) items, split-major, are cut into 8 consecutive runs, one per XCD, so the tiles of one M-split — which
//     stream the same G / A rows — sit on ONE XCD and share them through its L2, with the FEWEST M-splits that fill the chip (every split costs
//     N x K fp32 atomics in the epilogue: at 340 G atomics/s chip-wide a workgroup's 65,536 take ~49 us).  The default on every shape.
//   XCD-aware (1): all output tiles of one M-split side by side on one XCD, M-splits = 8 k (round 2; OD_TN_PACK=1 brings it back for >= 16 tiles).
//   plain (0): tile index fastest.
// every lambda of the kernel must be inlined: one that is not keeps its by-reference captures (the 256 accumulators!) in scratch memory
#if defined(OD_EMU)
#define TNW4_INLINE
#else
#define TNW4_INLINE __attribute__((always_inline))
#endif
#ifndef OD_TNW4_DMA_EVERY
#define OD_TNW4_DMA_EVERY 3       // a DMA piece behind every n-th MFMA from 33 on
#endif
#ifndef OD_TNW4_L_AT
#define OD_TNW4_L_AT 88           // LANDED barrier in front of this MFMA; the next slab's half-0 reads follow it, one per MFMA
#endif
#ifndef OD_TNW4_X
#define OD_TNW4_X 0        // timing experiments only (wrong results): 2 no loop fragment reads, 16 no fetch
#endif
__global__ __launch_bounds__(256, 1) void gemm_tn_w4_kernel(const bf16_t* __restrict__ G, int ldg, const bf16_t* __restrict__ A, int lda,
                                                            float* __restrict__ dW, int lddw, float* __restrict__ dbias,
                                                            int M, int N, int K, int m_per_block, int xcd_full,
                                                            const OdDetTable* __restrict__ det, TnRowMap rm) {
    const int xcd_order = xcd_full & 3;
    constexpr int STG = 65536;                 // G slab [64][256] 32 KiB + A slab [64][256] 32 KiB
    OD_DYN_SMEM(smem);
    float* sred = (float*)(smem + 2 * STG);    // 256 floats
    const int tiles_n = (N + 255) / 256, tiles_k = (K + 255) / 256;
    int tile, split;
    if (xcd_order == 2) {
        const int per_xcd = xcd_full >> 2;
        const int gi = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
        if ((blockIdx.x >> 3) >= per_xcd) return;
        split = gi / (tiles_n * tiles_k);
        tile = gi % (tiles_n * tiles_k);
    } else if (xcd_order) {
        const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
        split = (slot / (tiles_n * tiles_k)) * 8 + xcd;
        tile = slot % (tiles_n * tiles_k);
    } else {
        tile = blockIdx.x % (tiles_n * tiles_k); split = blockIdx.x / (tiles_n * tiles_k);
    }
    const int n0 = (tile / tiles_k) * 256, k0 = (tile % tiles_k) * 256;
    const int mb = split * m_per_block;
    int me = mb + m_per_block; me = me < M ? me : M;
    if (mb >= M) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = od_uniform(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int x = lane & 15, g = lane >> 4;
    const bool do_bias = dbias != nullptr && (tile % tiles_k) == 0;
    sred[tid] = 0.f;

    f32x4 acc[8][8];           // [n tile i][k tile j]
#pragma unroll
    for (int i = 0; i < 8; i++)
#pragma unroll
        for (int j = 0; j < 8; j++) acc[i][j] = (f32x4)(0.f);
    const int nslab = (me - mb + 63) / 64;

    // staging: waves 0, 1 stream the G slab (rows (wave & 1) * 32 ...), waves 2, 3 the A slab; 16 pieces of 2 rows x 512 B per wave and slab.
    // Rows past this block's M range and bytes past the operand's end read as zero (the descriptor ends at row `me`, the last row's width
    // rounded up to a 16-byte chunk: the hardware range-checks per dword).
    const bool isa = wave >= 2;
    const int ld = isa ? lda : ldg, c0 = isa ? k0 : n0, width = isa ? K : N;
    const bf16_t* opnd = (isa ? A : G) + (size_t)mb * ld + c0;
    const long avail = (long)(me - mb - 1) * ld + ((width - c0 + 7) & ~7);
    od_srd_t srd = od_make_srd(opnd, (unsigned)((avail > 0 ? avail : 0) * 2));
    if (OD_TNW4_X & 16) od_srd_set_bytes(srd, 0u);
    unsigned voff8[8];                                                       // by piece & 7 (the swizzle key is row & 15 = (2 i + lane / 32) & 15)
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const int r = ((wave & 1) * 16 + i) * 2 + (lane >> 5);              // row of the 64-row slab (pieces 8..15: + 16 rows, same key)
        const int pos = lane & 31;                                          // 16-byte position within the 512-byte LDS row
        const int slot = ((((pos >> 1) ^ (r & 15)) << 1) | (pos & 1));
        voff8[i] = (unsigned)(r * ld * 2 + slot * 16);
    }
    const unsigned lds_mine = od_lds_addr(smem) + (isa ? 32768u : 0u) + (unsigned)(wave & 1) * 16384u;
    const unsigned half_stride = (unsigned)(16 * ld * 2);                    // pieces 8..15 sit 16 rows further down

    od_frag<bf16_t> fa[2][8], fb[2][8];                    // [half = register set][tile]: fa from the G slab (n), fb from the A slab (k)
    // read r of a half's 32: r = 2 t + e -> transpose read e of fragment t (t < 8: fa, else fb), in the order the MFMAs first use them
    // (MFMA n of a half = (i = n >> 3, j = n & 7): fa[0] and all eight fb first)
    // LDS byte addresses of the transpose-read chunks, one register per 16-column fragment tile (0..7: the wave's G columns, 8..15: its A
    // columns): row 4 g + (x >> 2) of the stage — + 32 u + 16 e rows are multiples of 16, which leave tn512_off's swizzle key (row & 15) alone
    // and fold into the instruction's offset field —, 32-byte slot (tile index) ^ key, 8 (x & 3) bytes in.  They point at the stage being
    // READ and flip to the other one (^ 65536) in the middle of every slab, between the last read of this stage and the first of the next.
    unsigned offs[16];
    {
        const int rb = 4 * g + (x >> 2);
        const unsigned base = od_lds_addr(smem);
#pragma unroll
        for (int t = 0; t < 8; t++) {
            offs[t] = base + (unsigned)(rb * 512 + (((wm * 8 + t) ^ rb) << 5) + 8 * (x & 3));
            offs[8 + t] = base + (unsigned)(32768 + rb * 512 + (((wn * 8 + t) ^ rb) << 5) + 8 * (x & 3));
        }
    }
    auto rd_one = [&](int u, int r) TNW4_INLINE {
        const int t = r >> 1, e = r & 1;
        const int seq = t == 0 ? 0 : t < 9 ? t + 7 : t - 8;            // fragment order: fa[0], fb[0..7], fa[1..7]  -> index into (fa: 0..7, fb: 8..15)
        const bool is_b = seq >= 8;
        const int idx = seq & 7;
        const s16x4 v4 = od_lds_tr_read_at(offs[seq] + (unsigned)((32 * u + 16 * e) * 512));
        od_frag<bf16_t>& f = is_b ? fb[u][idx] : fa[u][idx];
        f.v[4 * e] = v4[0]; f.v[4 * e + 1] = v4[1]; f.v[4 * e + 2] = v4[2]; f.v[4 * e + 3] = v4[3];
    };
    auto mma_one = [&](int u, int n) TNW4_INLINE {
        const int i = n >> 3, j = n & 7;
#if defined(OD_EMU)
        acc[i][j] = od_mma(fa[u][i], fb[u][j], acc[i][j]);
#else
        asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[i][j]) : "v"(fa[u][i].v), "v"(fb[u][j].v));
#endif
    };
    // bias tiles: column sums of the G slab in 16-byte row pieces.  A thread owns 8 columns and the rows (tid >> 5) + 8 p + 16 q (p = 0, 1;
    // q = 0..3): rows 16 apart share the swizzle key, so q enters as an immediate and two address registers serve the eight reads of a slab.
    // (They follow the fragment addresses to the other stage in the middle of every slab.)
    float bs[8];
#pragma unroll
    for (int e = 0; e < 8; e++) bs[e] = 0.f;
    const int bcol8 = (tid & 31) * 8;
    unsigned boff[2];
#pragma unroll
    for (int p_ = 0; p_ < 2; p_++) {
        const int rk = (tid >> 5) + 8 * p_;
        boff[p_] = od_lds_addr(smem) + (unsigned)(rk * 512 + (((((tid & 31) >> 1)) ^ rk) << 5) + (tid & 1) * 16);
    }

    // prologue: slabs 0 and 1 in flight, slab 0 landed, its half-0 fragments in register set 0
#pragma unroll
    for (int t = 0; t < 2; t++) {
        const unsigned so = (unsigned)t * 64u * (unsigned)ld * 2u;
#pragma unroll
        for (int i = 0; i < 16; i++)
            od_buffer_lds16_at(srd, voff8[i & 7], so + (unsigned)(i >> 3) * half_stride, lds_mine + (unsigned)t * STG + (unsigned)i * 1024u);
    }
    OD_WAIT_VMCNT(16);
    od_barrier_raw();
#pragma unroll
    for (int r = 0; r < 32; r++) rd_one(0, r);

    auto slab = [&](int st, const int xs) TNW4_INLINE {
        const unsigned dst = lds_mine + (unsigned)xs * STG;
        const unsigned so = (unsigned)(st + 2) * 64u * (unsigned)ld * 2u;      // past the last slab: beyond the descriptor, zeros
#pragma clang loop unroll(full)
        for (int n = 0; n < 128; n++) {
            if (n == 32) {
                if (do_bias) {
#pragma unroll
                    for (int rr = 0; rr < 8; rr++) {
                        float v8[8];
                        od_lds_ld8_at(boff[rr & 1] + (unsigned)((rr >> 1) * 16 * 512), v8);
#pragma unroll
                        for (int e = 0; e < 8; e++) bs[e] += v8[e];
                    }
                }
                OD_WAIT_LGKMCNT(0);
                od_barrier_raw();
            }
            static_assert(33 + 15 * OD_TNW4_DMA_EVERY < OD_TNW4_L_AT && OD_TNW4_L_AT + 32 <= 128 && OD_TNW4_L_AT >= 56, "schedule does not fit the slab");
            if (n == OD_TNW4_L_AT) {
                OD_WAIT_VMCNT(16);
                od_barrier_raw();
            }
            const bool d = n >= 33 && n < 33 + 16 * OD_TNW4_DMA_EVERY && (n - 33) % OD_TNW4_DMA_EVERY == 0;         // 16 pieces at MFMAs 33, 36, ..., 78
            const int q = (n - 33) / OD_TNW4_DMA_EVERY;
            if (d) od_dma_set_dst(dst + (unsigned)q * 1024u);
            if (!(OD_TNW4_X & 2)) {
                if (n < 32) rd_one(1, n);
                if (n >= OD_TNW4_L_AT && n < OD_TNW4_L_AT + 32) rd_one(0, n - OD_TNW4_L_AT);
            }
            if (n >= 40 && n < 56) offs[n - 40] ^= 65536u;                     // the read addresses move to the other stage
            if (n == 56) { boff[0] ^= 65536u; boff[1] ^= 65536u; }
            mma_one(n >> 6, n & 63);
            if (d) od_buffer_lds16_m0(srd, voff8[q & 7], so + (unsigned)(q >> 3) * half_stride);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    // slabs in pairs, unconditionally (an odd count computes one slab of zeros: rows past `me` lie beyond the descriptor) — with a branch
    // between the two instances the register allocator kept the accumulators in VGPRs and copied them to AGPRs in front of every MFMA
    for (int st = 0; st < nslab; st += 2) {
        slab(st, 0);
        slab(st + 1, 1);
    }
#if !defined(OD_EMU)
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");   // the last MFMAs' results before the epilogue reads the accumulators
#endif
    OD_WAIT_VMCNT(0);
    if (do_bias) {
        // the 8 row groups of a column meet in a FIXED order (deterministic), one group per barrier
        for (int rg = 0; rg < 8; rg++) {
            __syncthreads();
            if ((tid >> 5) == rg) {
#pragma unroll
                for (int e = 0; e < 8; e++) sred[bcol8 + e] += bs[e];
            }
        }
    }
    long long* const dw_shadow = od_det_find(det, dW);
#pragma unroll
    for (int i = 0; i < 8; i++)
#pragma unroll
        for (int j = 0; j < 8; j++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int n = n0 + wm * 128 + i * 16 + g * 4 + r, k = k0 + wn * 128 + j * 16 + x;
                int no;
                if (tn_map_row(rm, n, N, no) && k < K) od_red_add_at(dw_shadow, dW, (size_t)no * lddw + k, acc[i][j][r]);
            }
    if (do_bias) {
        __syncthreads();
        int no;
        if (tn_map_row(rm, n0 + tid, N, no)) od_red_add(det, dbias + no, sred[tid]);
    }
}

// (Round 3 also measured a 4-deep ring of 32-row slabs — three stages = 96 KiB in flight per CU, counted vmcnt, one bare barrier per stage —
// in place of the two 64-row stages: identical within +-1 % on every shape, profiles/r03d_ab_gemm_tn.txt.  Like the NT kernel, this loop is
// not waiting for one slab's latency; its DMA skeleton and its MFMA + transpose-read skeleton each run at ~1.05 PF/s on their own.  Removed.)

// column sums (bias gradients): out[n] += sum_m G[m][n]
template <class T>
__global__ __launch_bounds__(256) void colsum_kernel(const T* __restrict__ G, int ldg, float* __restrict__ out, int M, int N,
                                                     int rows_per_block, const OdDetTable* __restrict__ det) {
    const int n = blockIdx.x * 256 + threadIdx.x;
    const int mb = blockIdx.y * rows_per_block;
    int me = mb + rows_per_block; me = me < M ? me : M;
    if (n >= N) return;
    float s = 0.f;
    for (int m = mb; m < me; m++) s += od_t<T>::ld(G + (size_t)m * ldg + n);
    od_red_add(det, out + n, s);
}

template <class T>
int launch_nt(const T* A, int lda, const T* W, int ldw, const float* bias, T* C, int ldc, int M, int N, int K, int epi,
              int accumulate, hipStream_t st, RopeEpi rp = RopeEpi{}) {
    const int tiles_n = (N + BN - 1) / BN;
#ifndef OD_GEMM_SMALL_TILES
#define OD_GEMM_SMALL_TILES 512     // fewer 128-row tiles than 2 per CU: use 64-row tiles (the emulator build lowers it)
#endif
    const bool half = ((M + 127) / 128) * tiles_n < OD_GEMM_SMALL_TILES;
#ifndef OD_GEMM_QUARTER_TILES
#define OD_GEMM_QUARTER_TILES 0      // bf16: 32-row tiles never paid (out 13.0 -> 12.2 us, proj_o 14.6 -> 15.5 us at M = 4460)
#endif
#ifndef OD_GEMM_QUARTER_TILES_F32
#define OD_GEMM_QUARTER_TILES_F32 600   // fp32 products (fp32-as-3xbf16 measured +-0: left on 64-row tiles) (32-float k slabs: twice the iterations of bf16, one wave per SIMD at 64-row
                                        // tiles): 32-row tiles put two workgroups on a CU — out 74 -> 60 us, proj_o 100 -> 80 us at M = 4460
#endif
    const bool quarter = ((M + 63) / 64) * tiles_n < (std::is_same<T, float>::value ? OD_GEMM_QUARTER_TILES_F32 : OD_GEMM_QUARTER_TILES);
    const int tiles_m = quarter ? (M + 31) / 32 : half ? (M + 63) / 64 : (M + 127) / 128;
    const int grid = ((tiles_m + 7) / 8) * 8 * tiles_n;
    const bool dma = (K % (128 / (int)sizeof(T))) == 0;
    // q/k norm + RoPE in the large-M kernel's epilogue: bf16, head_dim 64 (a wave's 64 columns are one head), whole 256-column tiles
    const bool big_rope = epi == OD_EPI_QKROPE && std::is_same<T, bf16_t>::value && rp.hd == 64 && rp.n_rope % 64 == 0 && N % 64 == 0 &&
                          (!rp.qk_out || rp.ldqk % 8 == 0);
    // Below OD_GEMM_BIG_MIN_M rows (the sampler: M = 4460) a product whose 256 x 256 tiles fill most of the chip in ONE round — qkv (N = 3072:
    // 216 tiles) and proj_vg (N = 2816: 198) — still takes the 4-wave persistent kernel: its tile time beats the 128-row kernel's three rounds
    // (round 6: profiles/r06b_ab_sampler_big_tiles.txt).  OD_NT_BIG_MIN_TILES: the fewest tiles for that (0 = never).
    static const int big_min_tiles = od_env_int("OD_NT_BIG_MIN_TILES", OD_NT_BIG_MIN_TILES);
    const int tiles256 = ((M + 255) / 256) * ((N + 255) / 256);
    const bool big_by_tiles = std::is_same<T, bf16_t>::value && big_min_tiles > 0 && M >= 1024 && tiles256 >= big_min_tiles && tiles256 <= od_num_cus() &&
                              !accumulate && K % 128 == 0;
    if ((epi != OD_EPI_QKROPE || big_rope) && dma && (M >= OD_GEMM_BIG_MIN_M || big_by_tiles) && N % 8 == 0 && ldc % 8 == 0 && N >= 256) {
        const int tm2 = (M + 255) / 256, tn2 = (N + 255) / 256;
        const int grid2 = ((tm2 + 7) / 8) * 8 * tn2;
        // Wide outputs are written with non-temporal stores: the 128 KiB tile bursts of 256 CUs (32 MiB, the size of all L2s) otherwise
        // evict the W / A lines the next tiles re-read, and this loop is bound by fetch latency x outstanding misses
        // (profiles/r02l_gemm_fetch_bound.txt): +7..12 % at N = 1024..2816, +2.5 % at 3072; at N = 512 (2 column tiles) it costs 3 %.
        const int nt_store = !accumulate && N >= OD_GEMM_NT_STORE_MIN_N;
        if constexpr (std::is_same<T, bf16_t>::value) {
            static const int w4 = od_env_int("OD_NT_W4", 1);
            static const int w4_min_k = od_env_int("OD_NT_W4_MIN_K", 0);
            static const int w4_rope = od_env_int("OD_NT_W4_QKROPE", 1);      // (0: the 8-wave kernel's norm + RoPE epilogue; A/B)
            if ((rp.f16 || (w4 && (epi != OD_EPI_QKROPE || w4_rope) && K >= w4_min_k)) && !accumulate && K % 128 == 0) {
                int pgrid = od_num_cus() & ~7;                 // persistent: one workgroup per CU, a multiple of 8 (block b runs on XCD b % 8)
                pgrid = pgrid < 8 ? 8 : pgrid;
                if (epi == OD_EPI_QKROPE)
                    OD_LAUNCH_DYN((gemm_nt_w4_kernel<OD_EPI_QKROPE>), dim3(grid2 < pgrid ? grid2 : pgrid), dim3(256), 131072, st, A, lda, W, ldw, bias, C, ldc, M, N, K, (rp.qk_out ? nt_store : 0), rp);
                else if (epi == OD_EPI_SILU)
                    OD_LAUNCH_DYN((gemm_nt_w4_kernel<OD_EPI_SILU>), dim3(grid2 < pgrid ? grid2 : pgrid), dim3(256), 131072, st, A, lda, W, ldw, bias, C, ldc, M, N, K, nt_store, rp);
                else
                    OD_LAUNCH_DYN((gemm_nt_w4_kernel<OD_EPI_NONE>), dim3(grid2 < pgrid ? grid2 : pgrid), dim3(256), 131072, st, A, lda, W, ldw, bias, C, ldc, M, N, K, nt_store, rp);
                OD_CHECK_LAUNCH();
                return 0;
            }
        }
        if (epi == OD_EPI_QKROPE) {
            if constexpr (std::is_same<T, bf16_t>::value)
                OD_LAUNCH_DYN((gemm_nt_big_kernel<T, OD_EPI_QKROPE>), dim3(grid2), dim3(512), 131072, st, A, lda, W, ldw, bias, C, ldc, M, N, K, 0, (rp.qk_out ? nt_store : 0), rp);
        } else if (epi == OD_EPI_SILU)
            OD_LAUNCH_DYN((gemm_nt_big_kernel<T, OD_EPI_SILU>), dim3(grid2), dim3(512), 131072, st, A, lda, W, ldw, bias, C, ldc, M, N, K, accumulate, nt_store, rp);
        else
            OD_LAUNCH_DYN((gemm_nt_big_kernel<T, OD_EPI_NONE>), dim3(grid2), dim3(512), 131072, st, A, lda, W, ldw, bias, C, ldc, M, N, K, accumulate, nt_store, rp);
        OD_CHECK_LAUNCH();
        return 0;
    }
    if (epi == OD_EPI_QKROPE && rp.qk_out) return OD_ERR_UNSUPPORTED;      // the split form exists in the large-M kernel only (callers check)
#define NT_GO(EPI_, DMA_, WMT_) OD_LAUNCH_DYN((gemm_nt_kernel<T, EPI_, DMA_, WMT_>), dim3(grid), dim3(256), (gemm_nt_smem_bytes<T, DMA_, WMT_>()), st, A, lda, W, ldw, bias, C, ldc, M, N, K, accumulate, rp)
#define NT_GO2(EPI_, DMA_) do { if (quarter) NT_GO(EPI_, DMA_, 1); else if (half) NT_GO(EPI_, DMA_, 2); else NT_GO(EPI_, DMA_, 4); } while (0)
    if (epi == OD_EPI_QKROPE) {
        if (dma) NT_GO2(OD_EPI_QKROPE, true); else NT_GO2(OD_EPI_QKROPE, false);
    } else if (epi == OD_EPI_SILU) {
        if (dma) NT_GO2(OD_EPI_SILU, true); else NT_GO2(OD_EPI_SILU, false);
    } else {
        if (dma) NT_GO2(OD_EPI_NONE, true); else NT_GO2(OD_EPI_NONE, false);
    }
#undef NT_GO2
#undef NT_GO
    OD_CHECK_LAUNCH();
    return 0;
}

template <class T>
int launch_tn(const T* G, int ldg, const T* A, int lda, float* dW, int lddw, float* dbias, int M, int N, int K, hipStream_t st, TnRowMap rm = TnRowMap{0, 0}) {
    constexpr int BR = 128 / (int)sizeof(T);
    if constexpr (sizeof(T) == 2) {
        const int tiles2 = ((N + 255) / 256) * ((K + 255) / 256);
        // few output tiles => many M-splits => the fp32 atomics of the epilogue dominate: stay on 128x128 there
        if (M >= OD_GEMM_BIG_MIN_M && N >= 256 && K >= 256 && (tiles2 >= OD_TN_BIG_MIN_TILES || OD_GEMM_BIG_MIN_M < 32768)) {
#ifndef OD_TN_BLOCKS
#define OD_TN_BLOCKS 256     // one workgroup per CU: M-splits = 256 / output tiles (fewest fp32 atomics, no second block wave)
#endif
#ifndef OD_TN_XCD_MIN_TILES
#define OD_TN_XCD_MIN_TILES 16
#endif
            static const int xcd_min_tiles = od_env_int("OD_TN_XCD_MIN_TILES", OD_TN_XCD_MIN_TILES);
            static const int eff_pct = od_env_int("OD_TN_EFF_PCT", 100);      // take the SMALLEST k whose fill efficiency reaches this
            // packed order (round 3) below `xcd_min_tiles` output tiles; from there on the round-2 order (8 k splits, whole splits per XCD, several
            // block rounds), which still wins on the 24-tile qkv shape (777 vs 750 TF/s).  OD_TN_PACK=0 / 2: never / always packed (A/B).
            // Round 5: with the 4-wave kernel the packed order wins on every shape of the step — qkv (24 tiles) 803 against 811 us, the merged
            // 22-tile proj_vg gradient 746 against 924 (the round-2 order takes 56 M-splits there: 80 M epilogue atomics) — and is the default.
            static const int pack_mode = od_env_int("OD_TN_PACK", 2);
            int xcd_order = tiles2 >= xcd_min_tiles;
            const bool pack = pack_mode == 2 || (pack_mode == 1 && !xcd_order);
            int sp, grid_tn;
            if (pack) {
                // fewest M-splits that fill the chip, whole splits side by side on an XCD (see the kernel): tiles2 x sp <= 256 workgroups
                sp = OD_TN_BLOCKS / tiles2 > 0 ? OD_TN_BLOCKS / tiles2 : 1;
            } else if (xcd_order) {
                // M-splits = 8 k: each XCD (32 CUs, one workgroup each) holds k splits x tiles2 tiles; pick the k whose k * tiles2 fills
                // whole waves of 32 workgroups best (qkv: 24 tiles -> k = 4 -> 96 = 3 x 32)
                int best_k = 1; double best_eff = 0.0;
                for (int kk = 1; kk <= 8; kk++) {
                    const int bpx = kk * tiles2, waves = (bpx + 31) / 32;
                    const double eff = (double)bpx / (waves * 32);
                    if (eff > best_eff + 1e-9) { best_eff = eff; best_k = kk; }
                    if (eff * 100.0 >= eff_pct - 1e-9) { best_k = kk; break; }
                }
                static const int force_k = od_env_int("OD_TN_KK", 0);                // A/B: M-splits per XCD given outright
                if (force_k > 0) best_k = force_k;
                sp = 8 * best_k;
            } else {
                sp = OD_TN_BLOCKS >= 512 ? (OD_TN_BLOCKS + tiles2 - 1) / tiles2 : (OD_TN_BLOCKS / tiles2 > 0 ? OD_TN_BLOCKS / tiles2 : 1);
            }
            int mpb2 = (M + sp - 1) / sp;
            mpb2 = ((mpb2 + 63) / 64) * 64;
            sp = (M + mpb2 - 1) / mpb2;
            if (pack) {
                const int per_xcd = (tiles2 * sp + 7) / 8;
                xcd_order = (per_xcd << 2) | 2;
                grid_tn = per_xcd * 8;
            } else
                grid_tn = xcd_order ? ((sp + 7) / 8) * 8 * tiles2 : tiles2 * sp;
            OD_LAUNCH_DYN(gemm_tn_w4_kernel, dim3(grid_tn), dim3(256), (131072 + 1024), st, G, ldg, A, lda, dW, lddw, dbias, M, N, K, mpb2, xcd_order, od_det_active(), rm);
            OD_CHECK_LAUNCH();
            return 0;
        }
    }
    const int tiles = ((N + BN - 1) / BN) * ((K + BM - 1) / BM);
    // ~2048 workgroups (8 per CU) — but every M-split costs N x K fp32 atomics, and with few output tiles that is what the launch waits for
    // (proj_cl's dW, 4 tiles: 512 splits = 33.5 M atomics ~ 0.1 ms of a 0.155 ms launch): large-M launches with <= 8 tiles aim for 512
    static const int few_tile_wgs = od_env_int("OD_TN_SMALL_WGS", 512);
    const int target_wgs = (tiles <= 8 && M >= OD_GEMM_BIG_MIN_M) ? few_tile_wgs : 2048;
    int splits = (target_wgs + tiles - 1) / tiles;
    int mpb = (M + splits - 1) / splits;
    mpb = ((mpb + BR - 1) / BR) * BR;
    if (mpb < 4 * BR) mpb = 4 * BR;
    splits = (M + mpb - 1) / mpb;
    OD_LAUNCH((gemm_tn_kernel<T>), dim3(tiles * splits), dim3(256), 0, st, G, ldg, A, lda, dW, lddw, dbias, M, N, K, mpb, od_det_active(), rm);
    OD_CHECK_LAUNCH();
    return 0;
}

}  // namespace

extern "C" int od_gemm_nt(int dtype, const void* A, int lda, const void* W, int ldw, const float* bias, void* C, int ldc,
                          int M, int N, int K, int epilogue, int accumulate, void* stream) {
    if (M <= 0 || N <= 0 || K <= 0) return OD_ERR_ARG;
    const int ch = dtype == OD_BF16 ? 8 : 4;
    if (lda % ch || ldw % ch || K % ch) return OD_ERR_ALIGN;
    if (dtype == OD_BF16)
        return launch_nt<bf16_t>((const bf16_t*)A, lda, (const bf16_t*)W, ldw, bias, (bf16_t*)C, ldc, M, N, K, epilogue, accumulate, (hipStream_t)stream);
    if (dtype == OD_F32)
        return launch_nt<float>((const float*)A, lda, (const float*)W, ldw, bias, (float*)C, ldc, M, N, K, epilogue, accumulate, (hipStream_t)stream);
    if (dtype == OD_F32X3)
        return launch_nt<f32x3_t>((const f32x3_t*)A, lda, (const f32x3_t*)W, ldw, bias, (f32x3_t*)C, ldc, M, N, K, epilogue, accumulate, (hipStream_t)stream);
    if (dtype == OD_F32X3W) {
        if (K % 32) return OD_ERR_ALIGN;
        return launch_nt<f32x3w_t>((const f32x3w_t*)A, lda, (const f32x3w_t*)W, ldw, bias, (f32x3w_t*)C, ldc, M, N, K, epilogue, accumulate, (hipStream_t)stream);
    }
    return OD_ERR_ARG;
}

extern "C" int od_gemm_nt_qkrope(int dtype, const void* A, int lda, const void* W, int ldw, const float* bias, void* C, int ldc,
                                 int M, int N, int K, const float* wq, const float* wk, const float* table, int L, int H, int hd,
                                 float eps, float q_scale, void* stream) {
    if (M <= 0 || N <= 0 || K <= 0 || !bias || !wq || !wk || !table || L <= 0) return OD_ERR_ARG;
    const int ch = dtype == OD_BF16 ? 8 : 4;
    if (lda % ch || ldw % ch || K % ch || N % 8 || ldc % 8) return OD_ERR_ALIGN;
    const int n_rope = 2 * H * hd;
    if ((hd != 32 && hd != 64) || n_rope % 128 || n_rope > N) return OD_ERR_UNSUPPORTED;
    const RopeEpi rp{wq, wk, table, L, H * hd, hd, n_rope, eps, q_scale, nullptr, 0, 0};
    if (dtype == OD_BF16)
        return launch_nt<bf16_t>((const bf16_t*)A, lda, (const bf16_t*)W, ldw, bias, (bf16_t*)C, ldc, M, N, K, OD_EPI_QKROPE, 0, (hipStream_t)stream, rp);
    if (dtype == OD_F32)
        return launch_nt<float>((const float*)A, lda, (const float*)W, ldw, bias, (float*)C, ldc, M, N, K, OD_EPI_QKROPE, 0, (hipStream_t)stream, rp);
    if (dtype == OD_F32X3)
        return launch_nt<f32x3_t>((const f32x3_t*)A, lda, (const f32x3_t*)W, ldw, bias, (f32x3_t*)C, ldc, M, N, K, OD_EPI_QKROPE, 0, (hipStream_t)stream, rp);
    if (dtype == OD_F32X3W) {
        if (K % 32) return OD_ERR_ALIGN;
        return launch_nt<f32x3w_t>((const f32x3w_t*)A, lda, (const f32x3w_t*)W, ldw, bias, (f32x3w_t*)C, ldc, M, N, K, OD_EPI_QKROPE, 0, (hipStream_t)stream, rp);
    }
    return OD_ERR_ARG;
}

namespace {
// columns [c0, c0 + ncols) of a bf16 matrix re-encoded as IEEE half in place (the small-shape path of "attention in fp16": v)
__global__ __launch_bounds__(256) void cast_bf16_to_f16_kernel(bf16_t* __restrict__ p, int ld, long M, int ncols) {
    const int nch = ncols / 8;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < M * nch; i += (long)gridDim.x * 256) {
        bf16_t* a = p + (i / nch) * ld + (i % nch) * 8;
        float v[8];
        od_ld8(a, v);
        od_st8((f16_t*)a, v);
    }
}
}  // namespace

extern "C" int od_gemm_nt_qkrope_split(int dtype, const void* A, int lda, const void* W, int ldw, const float* bias, void* C, int ldc,
                                       void* qk_out, int ldqk, int qk_dtype, int M, int N, int K, const float* wq, const float* wk,
                                       const float* table, int L, int H, int hd, float eps, float q_scale, void* stream) {
    if (M <= 0 || N <= 0 || K <= 0 || !bias || !wq || !wk || !table || L <= 0 || !qk_out || M % L) return OD_ERR_ARG;
    const int ch = dtype == OD_BF16 ? 8 : 4;
    if (lda % ch || ldw % ch || K % ch || N % 8 || ldc % 8 || ldqk % 8) return OD_ERR_ALIGN;
    const int n_rope = 2 * H * hd;
    if (n_rope > N) return OD_ERR_UNSUPPORTED;
    // qk_dtype = OD_F16 ("attention in fp16"): qk_out and the v columns of C (from 2*H*hd on) hold IEEE half; C's q / k columns — the
    // pre-norm values the norm's backward reads — stay bf16.  bf16 GEMM, head_dim 64 only.
    const bool f16 = qk_dtype == OD_F16;
    if (!f16 && qk_dtype != dtype && !((dtype == OD_F32X3 || dtype == OD_F32X3W) && qk_dtype == OD_F32)) return OD_ERR_ARG;
    if (f16 && (dtype != OD_BF16 || hd != 64 || (N - n_rope) % 8)) return OD_ERR_UNSUPPORTED;
    // one launch where the large-M kernel's epilogue applies (bf16, head_dim 64, M >= OD_GEMM_BIG_MIN_M, K a multiple of 64); otherwise the two
    // kernels it replaces
    if (dtype == OD_BF16 && hd == 64 && M >= OD_GEMM_BIG_MIN_M && K % 64 == 0 && N % 64 == 0 && N >= 256 && (!f16 || K % 128 == 0)) {
        const RopeEpi rp{wq, wk, table, L, H * hd, hd, n_rope, eps, q_scale, qk_out, ldqk, f16 ? 1 : 0};
        return launch_nt<bf16_t>((const bf16_t*)A, lda, (const bf16_t*)W, ldw, bias, (bf16_t*)C, ldc, M, N, K, OD_EPI_QKROPE, 0, (hipStream_t)stream, rp);
    }
    if (int rc = od_gemm_nt(dtype, A, lda, W, ldw, bias, C, ldc, M, N, K, OD_EPI_NONE, 0, stream)) return rc;
    if (int rc = od_qk_norm_rope(f16 ? OD_F16 : (dtype == OD_F32X3 || dtype == OD_F32X3W) ? OD_F32 : dtype, C, ldc, wq, wk, table, qk_out, ldqk, M / L, L, H, hd, eps,
                                 q_scale, stream)) return rc;
    if (f16 && N > n_rope) {
        OD_LAUNCH(cast_bf16_to_f16_kernel, dim3(1024), dim3(256), 0, (hipStream_t)stream, (bf16_t*)C + n_rope, ldc, (long)M, N - n_rope);
        OD_CHECK_LAUNCH();
    }
    return 0;
}

extern "C" int od_gemm_tn_blocks(int dtype, const void* G, int ldg, const void* A, int lda, float* dW, int lddw, float* dbias, int M,
                                 int N, int K, int n_block, int n_valid, void* stream) {
    if (M <= 0 || N <= 0 || K <= 0 || n_block < 0 || n_valid < 0 || n_valid > n_block) return OD_ERR_ARG;
    const int ch = dtype == OD_BF16 ? 8 : 4;
    if (ldg % ch || lda % ch) return OD_ERR_ALIGN;
    const TnRowMap rm{n_block, n_block ? n_valid : 0};
    if (dtype == OD_BF16) return launch_tn<bf16_t>((const bf16_t*)G, ldg, (const bf16_t*)A, lda, dW, lddw, dbias, M, N, K, (hipStream_t)stream, rm);
    if (dtype == OD_F32) return launch_tn<float>((const float*)G, ldg, (const float*)A, lda, dW, lddw, dbias, M, N, K, (hipStream_t)stream, rm);
    return OD_ERR_ARG;
}

extern "C" int od_gemm_tn(int dtype, const void* G, int ldg, const void* A, int lda, float* dW, int lddw, float* dbias, int M,
                          int N, int K, void* stream) {
    return od_gemm_tn_blocks(dtype, G, ldg, A, lda, dW, lddw, dbias, M, N, K, 0, 0, stream);
}

extern "C" int od_colsum(int dtype, const void* G, int ldg, float* out, int M, int N, void* stream) {
    if (M <= 0 || N <= 0) return OD_ERR_ARG;
    int rpb = (M + 255) / 256; if (rpb < 64) rpb = 64;
    dim3 grid((N + 255) / 256, (M + rpb - 1) / rpb);
    if (dtype == OD_BF16) OD_LAUNCH((colsum_kernel<bf16_t>), grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)G, ldg, out, M, N, rpb, od_det_active());
    else OD_LAUNCH((colsum_kernel<float>), grid, dim3(256), 0, (hipStream_t)stream, (const float*)G, ldg, out, M, N, rpb, od_det_active());
    OD_CHECK_LAUNCH();
    return 0;
}
