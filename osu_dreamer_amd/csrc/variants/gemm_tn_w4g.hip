// RETIRED EXPERIMENT (round 6; not part of the product build).  gemm_tn_w4g_kernel: the weight-gradient GEMM on 128 (n) x 512 (k) tiles, so that every
// column of G — the large operand of the qkv (3072 x 512) and proj_vg (2816 x 512) gradients — is fetched by ONE tile.  Built, parity-green
// (emulator and GPU: tests/test_kernels.py + tests/test_full_size.py gemm_tn tests with OD_TN_GONCE_MIN_N=1024), measured (profiles/r06p_tn_gonce.txt):
// w_qkv 812 -> 797 us, w_vg_v 402 -> 388, w_vg_merged 746 -> 740: 1-3 %, ~0.2 ms per training step.  Halving G's fabric reads buys almost nothing
// because what the fetch costs (25-34 % of the kernel, profiles/r06o_tn_skeletons.txt) is mostly the L2 -> LDS fill, which this tile raises from 64 to 80 KB
// per 128 MFMAs.  To build it again: paste the kernel behind gemm_tn_w4_kernel in gemm.hip and put this in launch_tn in front of the packed-order code:
//     static const int gonce_min_n = od_env_int("OD_TN_GONCE_MIN_N", 0);
//     if (gonce_min_n > 0 && N >= gonce_min_n && K <= 512 && K > 256) {
//         const int tiles3 = (N + 127) / 128;
//         int sp3 = OD_TN_BLOCKS / tiles3 > 0 ? OD_TN_BLOCKS / tiles3 : 1;
//         int mpb3 = (M + sp3 - 1) / sp3; mpb3 = ((mpb3 + 63) / 64) * 64; sp3 = (M + mpb3 - 1) / mpb3;
//         const int per_xcd3 = (tiles3 * sp3 + 7) / 8;
//         OD_LAUNCH_DYN(gemm_tn_w4g_kernel, dim3(per_xcd3 * 8), dim3(256), 163840, st, G, ldg, A, lda, dW, lddw, dbias, M, N, K, mpb3, (per_xcd3 << 2) | 2, od_det_active(), rm);
//         OD_CHECK_LAUNCH(); return 0;
//     }
#if 0
// ---- TN, "G once" (round 6).  Where the output is much taller than wide (qkv: 3072 x 512, proj_vg: 2816 x 512) the 256 x 256 tiles
// of gemm_tn_w4_kernel pair up along K: both read the same 256 columns of G — the LARGE operand — and each fetches them (3.65 GB at the fabric for
// 1.88 GB of operands on the qkv shape; the fetch is worth 25-34 % of the kernel: profiles/r06o_tn_skeletons.txt).  Here a workgroup owns 128 (n) x 512 (k):
// every G column is fetched by ONE tile.  Per 64-row slab: G [64][128] in 256-byte rows (tn_off's layout, 16 KB) and A as two [64][256] regions in the
// 512-byte-row layout (2 x 32 KB): 80 KB per stage, two stages = the CU's whole LDS (the bias sums meet in a stage after the loop).  Every wave reads
// ALL of G's fragments and its own 128 A columns (wave w: region w >> 1, half w & 1), streams 16 A pieces + 4 G pieces per slab.  Same MFMA /
// transpose-read schedule as gemm_tn_w4_kernel; LANDED moves to MFMA 96 (20 pieces behind every third MFMA from 33).  K <= 512.
#ifndef OD_TNG_X
#define OD_TNG_X 0
#endif
__global__ __launch_bounds__(256, 1) void gemm_tn_w4g_kernel(const bf16_t* __restrict__ G, int ldg, const bf16_t* __restrict__ A, int lda,
                                                             float* __restrict__ dW, int lddw, float* __restrict__ dbias,
                                                             int M, int N, int K, int m_per_block, int xcd_full,
                                                             const OdDetTable* __restrict__ det, TnRowMap rm) {
    const int xcd_order = xcd_full & 3;
    constexpr int STG = 81920;                 // G slab [64][128] 16 KiB + A regions 2 x [64][256] 64 KiB
    constexpr int L_AT = 96;
    OD_DYN_SMEM(smem);
    const int tiles_n = (N + 127) / 128, tiles_k = (K + 511) / 512;
    int tile, split;
    if (xcd_order == 2) {
        const int per_xcd = xcd_full >> 2;
        const int gi = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
        if ((blockIdx.x >> 3) >= per_xcd) return;
        split = gi / (tiles_n * tiles_k);
        tile = gi % (tiles_n * tiles_k);
    } else {
        tile = blockIdx.x % (tiles_n * tiles_k); split = blockIdx.x / (tiles_n * tiles_k);
    }
    const int n0 = (tile / tiles_k) * 128, k0 = (tile % tiles_k) * 512;
    const int mb = split * m_per_block;
    int me = mb + m_per_block; me = me < M ? me : M;
    if (mb >= M) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = od_uniform(tid >> 6);
    const int reg = wave >> 1, wn = wave & 1;                    // this wave's A region and its 128-column half
    const int x = lane & 15, g = lane >> 4;
    const bool do_bias = dbias != nullptr && (tile % tiles_k) == 0;

    f32x4 acc[8][8];           // [n tile i][k tile j]
#pragma unroll
    for (int i = 0; i < 8; i++)
#pragma unroll
        for (int j = 0; j < 8; j++) acc[i][j] = (f32x4)(0.f);
    const int nslab = (me - mb + 63) / 64;

    // staging.  A: wave w streams rows (w & 1) * 32 .. + 31 of region w >> 1 (columns k0 + 256 (w >> 1) ..), 16 pieces of 2 rows x 512 B, exactly as
    // gemm_tn_w4_kernel's A waves do.  G: pieces 4 w .. 4 w + 3 of 4 rows x 256 B.  Rows past `me` / bytes past the operand's end read as zero.
    const int ka = k0 + reg * 256;
    const long availa = (long)(me - mb - 1) * lda + (K > ka ? ((K - ka + 7) & ~7) : 0);
    od_srd_t srda = od_make_srd(A + (size_t)mb * lda + ka, (unsigned)((K > ka && availa > 0 ? availa : 0) * 2));
    const long availg = (long)(me - mb - 1) * ldg + ((N - n0 + 7) & ~7);
    od_srd_t srdg = od_make_srd(G + (size_t)mb * ldg + n0, (unsigned)((availg > 0 ? availg : 0) * 2));
    if (OD_TNG_X & 16) { od_srd_set_bytes(srda, 0u); od_srd_set_bytes(srdg, 0u); }
    unsigned voff8[8], vg4[4];
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const int r = (wn * 16 + i) * 2 + (lane >> 5);
        const int pos = lane & 31;
        const int slot = ((((pos >> 1) ^ (r & 15)) << 1) | (pos & 1));
        voff8[i] = (unsigned)(r * lda * 2 + slot * 16);
    }
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int r = (wave * 4 + i) * 4 + (lane >> 4);
        const int pos = lane & 15;
        const int slot = ((((pos >> 1) ^ (r & 7)) << 1) | (pos & 1));
        vg4[i] = (unsigned)(r * ldg * 2 + slot * 16);
    }
    const unsigned ldsA = od_lds_addr(smem) + 16384u + (unsigned)reg * 32768u + (unsigned)wn * 16384u;
    const unsigned ldsG = od_lds_addr(smem) + (unsigned)wave * 4096u;
    const unsigned half_stride = (unsigned)(16 * lda * 2);

    od_frag<bf16_t> fa[2][8], fb[2][8];                    // [half = register set][tile]: fa from the G slab (n), fb from the A region (k)
    unsigned offs[16];                                     // 0..7: G tiles (256-byte rows), 8..15: this wave's A tiles (512-byte rows); of the stage being READ
    {
        const int rb = 4 * g + (x >> 2);
        const unsigned base = od_lds_addr(smem);
#pragma unroll
        for (int t = 0; t < 8; t++) {
            offs[t] = base + (unsigned)(rb * 256 + ((t ^ (rb & 7)) << 5) + 8 * (x & 3));
            offs[8 + t] = base + (unsigned)(16384 + reg * 32768 + rb * 512 + (((wn * 8 + t) ^ rb) << 5) + 8 * (x & 3));
        }
    }
    auto rd_one = [&](int u, int r) TNW4_INLINE {
        const int t = r >> 1, e = r & 1;
        const int seq = t == 0 ? 0 : t < 9 ? t + 7 : t - 8;            // fragment order: fa[0], fb[0..7], fa[1..7]
        const bool is_b = seq >= 8;
        const int idx = seq & 7;
        const s16x4 v4 = od_lds_tr_read_at(offs[seq] + (unsigned)((32 * u + 16 * e) * (is_b ? 512 : 256)));
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
    // bias: column sums of the G slab.  Thread t owns columns 8 (t & 15) .. + 7 and rows (t >> 4) + 16 q (q = 0..3): rows 16 apart share tn_off's key
    float bs[8];
#pragma unroll
    for (int e = 0; e < 8; e++) bs[e] = 0.f;
    unsigned boff;
    {
        const int rk = tid >> 4, pos = tid & 15;
        boff = od_lds_addr(smem) + (unsigned)(rk * 256 + (((pos >> 1) ^ (rk & 7)) << 5) + (pos & 1) * 16);
    }

    // prologue: slabs 0 and 1 in flight, slab 0 landed, its half-0 fragments in register set 0
#pragma unroll
    for (int t = 0; t < 2; t++) {
        const unsigned soa = (unsigned)t * 64u * (unsigned)lda * 2u, sog = (unsigned)t * 64u * (unsigned)ldg * 2u;
#pragma unroll
        for (int i = 0; i < 16; i++)
            od_buffer_lds16_at(srda, voff8[i & 7], soa + (unsigned)(i >> 3) * half_stride, ldsA + (unsigned)t * STG + (unsigned)i * 1024u);
#pragma unroll
        for (int i = 0; i < 4; i++) od_buffer_lds16_at(srdg, vg4[i], sog, ldsG + (unsigned)t * STG + (unsigned)i * 1024u);
    }
    OD_WAIT_VMCNT(20);
    od_barrier_raw();
#pragma unroll
    for (int r = 0; r < 32; r++) rd_one(0, r);

    auto slab = [&](int st, auto xs_) TNW4_INLINE {
        constexpr int XS = decltype(xs_)::value;
        constexpr unsigned FLIP = XS == 0 ? (unsigned)STG : (unsigned)(0u - (unsigned)STG);   // the read addresses move to the other stage in the middle of the slab
        const unsigned soa = (unsigned)(st + 2) * 64u * (unsigned)lda * 2u, sog = (unsigned)(st + 2) * 64u * (unsigned)ldg * 2u;      // past the last slab: zeros
#pragma clang loop unroll(full)
        for (int n = 0; n < 128; n++) {
            if (n == 32) {
                if (do_bias) {
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        float v8[8];
                        od_lds_ld8_at(boff + (unsigned)(q * 16 * 256), v8);
#pragma unroll
                        for (int e = 0; e < 8; e++) bs[e] += v8[e];
                    }
                }
                OD_WAIT_LGKMCNT(0);
                od_barrier_raw();
            }
            if (n == L_AT) {
                OD_WAIT_VMCNT(20);
                od_barrier_raw();
            }
            const bool d = n >= 33 && n < 33 + 20 * 3 && (n - 33) % 3 == 0;         // 20 pieces at MFMAs 33, 36, ..., 90
            const int q = (n - 33) / 3;
            if (d) od_dma_set_dst((q < 16 ? ldsA + (unsigned)q * 1024u : ldsG + (unsigned)(q - 16) * 1024u) + (unsigned)XS * STG);
            if (!(OD_TNG_X & 2)) {
                if (n < 32) rd_one(1, n);
                if (n >= L_AT && n < L_AT + 32) rd_one(0, n - L_AT);
            }
            if (n >= 40 && n < 56) offs[n - 40] += FLIP;
            if (n == 56) boff += FLIP;
            mma_one(n >> 6, n & 63);
            if (d) {
                if (q < 16) od_buffer_lds16_m0(srda, voff8[q & 7], soa + (unsigned)(q >> 3) * half_stride);
                else od_buffer_lds16_m0(srdg, vg4[q - 16], sog);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    for (int st = 0; st < nslab; st += 2) {
        slab(st, std::integral_constant<int, 0>{});
        slab(st + 1, std::integral_constant<int, 1>{});
    }
#if !defined(OD_EMU)
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");   // the last MFMAs' results before the epilogue reads the accumulators
#endif
    OD_WAIT_VMCNT(0);
    long long* const dw_shadow = od_det_find(det, dW);
#pragma unroll
    for (int i = 0; i < 8; i++)
#pragma unroll
        for (int j = 0; j < 8; j++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int n = n0 + i * 16 + g * 4 + r, k = k0 + reg * 256 + wn * 128 + j * 16 + x;
                int no;
                if (tn_map_row(rm, n, N, no) && k < K) od_red_add_at(dw_shadow, dW, (size_t)no * lddw + k, acc[i][j][r]);
            }
    if (do_bias) {
        // the 16 row groups of a column meet in a stage (free now) and are added in index order: deterministic
        float* sp = (float*)smem;
        __syncthreads();
#pragma unroll
        for (int e = 0; e < 8; e++) sp[(tid >> 4) * 128 + (tid & 15) * 8 + e] = bs[e];
        __syncthreads();
        if (tid < 128) {
            float t = 0.f;
            for (int rg = 0; rg < 16; rg++) t += sp[rg * 128 + tid];
            int no;
            if (tn_map_row(rm, n0 + tid, N, no)) od_red_add(det, dbias + no, t);
        }
    }
}

#endif
