// RETIRED EXPERIMENT (round 6; not part of the product build).  gemm_nt_w4d_kernel: the 4-wave persistent NT GEMM with TWO accumulator sets, i.e.
// VERDICT r5 item 2 ("hide the NT GEMM epilogue behind the next tile's MFMAs"), built and measured: profiles/r06f_ab_nt_two_sets.txt.
//   * with the drain's stores removed the kernel runs 1475 TF/s on the K = 512 products (gemm_nt_w4_kernel: 1060-1090): the interleaved
//     drain (reads, conversions, line exchange) does hide behind the MFMAs, and the 256 x 128 tile's extra fetches do not hurt;
//   * WITH the stores it runs 835-850 TF/s: a wave's store stalls at issue while the chip's write path is busy, and with every CU storing all
//     the time the chip writes 1.6 TB/s — the 256 x 256 kernel's bursts of 512-byte row segments reach 2.05 TB/s.  The K = 512 products are
//     bound by the chip's write rate for this pattern, not by an exposed epilogue: there is nothing to hide it behind.
//   * this build also computes WRONG results on the GPU (NaNs; right on the emulator): a hazard around the asm MFMAs / accumulator
//     re-initialisation that was not chased once the timing had answered the question.
// To build it again: paste the kernel in front of tn_off() in gemm.hip and put this dispatch in launch_nt, in front of the gemm_nt_w4_kernel launches:
//     static const int w4d_max_k = od_env_int("OD_NT_W4D_MAX_K", 0);
//     if (epi == OD_EPI_NONE && K >= 512 && K <= w4d_max_k && !rp.f16) {
//         const int grid3 = ((tm2 + 7) / 8) * 8 * ((N + 127) / 128);
//         OD_LAUNCH_DYN((gemm_nt_w4d_kernel<0>), dim3(grid3 < pgrid ? grid3 : pgrid), dim3(256), 98304, st, A, lda, W, ldw, bias, C, ldc, M, N, K, nt_store);
//         OD_CHECK_LAUNCH();
//         return 0;
//     }
#if 0
// ---------------------------------------------------------------------------------------------------------------------------------
// gemm_nt_w4d_kernel: the 4-wave persistent organisation with TWO accumulator sets (round 6).
// gemm_nt_w4_kernel's epilogue — 256 accumulators read, converted and stored, 128 KB per workgroup, with one workgroup per CU and nothing beside
// it — is 7.3 us of an 18.4-us tile at K = 512 (profiles/r05_ab_records.txt), and a CU's stores only move at ~26 GB/s while a hundred other CUs
// are in their store phase (profiles/r06d_nt_store_contention.txt).  Here the output tile is 256 x 128: a wave owns 128 x 64 in 128 AGPRs per SET,
// and while the MFMAs of output tile v accumulate into set v & 1, the OTHER set — tile v - 1's results — is drained one 16-row unit per K tile:
// its reads, conversions, line exchange and two stores sit alone between the MFMAs like the fragment reads and DMA pieces do, and its registers
// restart at tile v + 1's bias.  Nothing of the epilogue is exposed except the last tile's.  Price: 24 fragment reads and 12 DMA pieces per 64
// MFMAs (the 256 x 256 tile: 16 and 8), 48 KB per stage.  K % 128 == 0, K >= 512 (eight K tiles carry the eight drain units), plain epilogue.
#ifndef OD_W4D_X
#define OD_W4D_X 0        // timing experiments only: 1 = no drain stores
#endif
#if defined(OD_EMU)
#define TNW4_INLINE
#else
#define TNW4_INLINE __attribute__((always_inline))
#endif
template <int DUMMY>
__global__ __launch_bounds__(256, 1) void gemm_nt_w4d_kernel(const bf16_t* __restrict__ A, int lda, const bf16_t* __restrict__ W, int ldw,
                                                             const float* __restrict__ bias, bf16_t* __restrict__ C, int ldc,
                                                             int M, int N, int K, int nt_store) {
    using T = bf16_t;
    constexpr int TM = 256, TN = 128, STG = 49152;       // stage: A 32 KiB (256 rows x 128 B) then W 16 KiB (128 rows)
    OD_DYN_SMEM(smem);
    const int tiles_m = (M + TM - 1) / TM, tiles_n = (N + TN - 1) / TN;
    const int xcd = blockIdx.x & 7, slot0 = blockIdx.x >> 3, slots = gridDim.x >> 3;
    auto tile_at = [&](int v, int& m0_, int& n0_) TNW4_INLINE {
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

    // lane (x, g) holds, for output row wm*128 + 16 j + x, the columns wn*64 + 32 p + 8 g .. + 7 in acc[s][2p][j][0..3], acc[s][2p+1][j][0..3]
    f32x4 acc[2][4][8];
    auto load_bias = [&](int n0_, f32x4 (&bv)[4]) TNW4_INLINE {
#pragma unroll
        for (int p = 0; p < 2; p++) {
            const int gn = n0_ + wn * 64 + 32 * p + 8 * g;
            float t[8];
#pragma unroll
            for (int e = 0; e < 8; e++) t[e] = 0.f;
            if (bias && gn < N) od_ld8(bias + gn, t);
#pragma unroll
            for (int r = 0; r < 4; r++) { bv[2 * p][r] = t[r]; bv[2 * p + 1][r] = t[4 + r]; }
        }
    };
    {
        f32x4 bv[4];
        load_bias(n0, bv);
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
            for (int j = 0; j < 8; j++) { acc[0][i][j] = bv[i]; acc[1][i][j] = (f32x4)(0.f); }
    }

    // staging: every wave streams 8 A pieces (rows wave*64 + 8 t ..) and 4 W pieces (rows wave*32 + 8 t ..) of 8 rows x 128 B per K tile; swizzle keys
    // as in gemm_nt_w4_kernel (A: row & 7; W: row bits 1, 3, 4)
    auto srd_of = [&](bool isw, int r0, bool valid) TNW4_INLINE {
        const int lim = isw ? N : M, ld = isw ? ldw : lda, full = isw ? TN : TM;
        const int rows = lim - r0 < full ? lim - r0 : full;
        const long avail = (long)(rows - 1) * ld + K;
        return od_make_srd((isw ? W : A) + (size_t)r0 * ld, valid && rows > 0 ? (unsigned)(avail * 2) : 0u);
    };
    od_srd_t sa_cur = srd_of(false, m0, true), sw_cur = srd_of(true, n0, true), sa_nxt = sa_cur, sw_nxt = sw_cur, sa = sa_cur, sw = sw_cur;
    const int prow = lane >> 3;
    const unsigned voffA = (unsigned)((wave * 64 + prow) * lda * 2 + (((lane & 7) ^ prow) << 4));
    unsigned voffW[4];
#pragma unroll
    for (int t = 0; t < 4; t++) {
        const int key = ((prow >> 1) & 1) | (t << 1);
        voffW[t] = (unsigned)((wave * 32 + t * 8 + prow) * ldw * 2 + (((lane & 7) ^ key) << 4));
    }
    const unsigned ldsA = od_lds_addr(smem) + (unsigned)wave * 8192u, ldsW = od_lds_addr(smem) + 32768u + (unsigned)wave * 4096u;
    const unsigned piece_stride = (unsigned)(8 * lda * 2);
    int offA[2], offW[2];
    {
        const int wkey = ((x >> 1) & 1) | ((x >> 2) << 1);
        const int wrow = wn * 64 + 8 * (x >> 2) + (x & 3);
#pragma unroll
        for (int sl = 0; sl < 2; sl++) {
            offA[sl] = (wm * 128 + x) * 128 + (((sl * 4 + g) ^ (x & 7)) << 4);          // + j * 2048
            offW[sl] = 32768 + wrow * 128 + (((sl * 4 + g) ^ wkey) << 4);              // + (i & 1) * 512 + (i >> 1) * 4096
        }
    }
    od_frag<T> fa[2][8], fw[2][4];                        // [slab = register set][tile]
    auto rdA = [&](const unsigned char* st, int sl, int j) TNW4_INLINE { fa[sl][j].v = *(const s16x8*)(st + offA[sl] + j * 2048); };
    auto rdW = [&](const unsigned char* st, int sl, int i) TNW4_INLINE { fw[sl][i].v = *(const s16x8*)(st + offW[sl] + (i & 1) * 512 + (i >> 1) * 4096); };
    auto rd_seq = [&](const unsigned char* st, int sl, int r) TNW4_INLINE {               // in the order of first use: W0, A0..A7, W1, W2, W3
        if (r == 0) rdW(st, sl, 0); else if (r < 9) rdA(st, sl, r - 1); else rdW(st, sl, r - 8);
    };
    auto dma_piece = [&](od_srd_t ra_, od_srd_t rw_, int q, unsigned so) TNW4_INLINE {     // the second half of a piece (its M0 was set in front of the MFMA)
        if (q < 8) od_buffer_lds16_m0(ra_, voffA, so + (unsigned)q * piece_stride);
        else od_buffer_lds16_m0(rw_, voffW[q - 8], so);
    };
    auto dma_dst = [&](unsigned stage_off, int q) TNW4_INLINE {
        od_dma_set_dst((q < 8 ? ldsA + (unsigned)q * 1024u : ldsW + (unsigned)(q - 8) * 1024u) + stage_off);
    };

    // prologue: K tiles 0 and 1 in flight, tile 0 landed, its slab-0 fragments in register set 0
#pragma unroll
    for (int t = 0; t < 2; t++)
#pragma unroll
        for (int q = 0; q < 12; q++) {
            if (q < 8) od_buffer_lds16_at(sa_cur, voffA, (unsigned)t * 128u + (unsigned)q * piece_stride, ldsA + (unsigned)t * STG + (unsigned)q * 1024u);
            else od_buffer_lds16_at(sw_cur, voffW[q - 8], (unsigned)t * 128u, ldsW + (unsigned)t * STG + (unsigned)(q - 8) * 1024u);
        }
    OD_WAIT_VMCNT(12);
    od_barrier_raw();
#pragma unroll
    for (int r = 0; r < 12; r++) rd_seq(smem, 0, r);

    // drain state: the previous output tile (whose results sit in the set that is NOT accumulating), and the bias the drained registers restart at
    int mp = 0, np = 0;
    bool have_prev = false;
    f32x4 bvn[4];
    const int xr = x & 7, xh = x >> 3;
    // one K tile: 64 MFMAs into set S from stages X (this tile) / Y (the next); DRAIN: unit J (rows 16 J .. + 15 of the wave's 128) of set S ^ 1 leaves
    auto ktile = [&](int kt, auto s_, auto xs_, auto drain_, auto j_) TNW4_INLINE {
        constexpr int S = decltype(s_)::value, XS = decltype(xs_)::value, J = decltype(j_)::value;
        constexpr bool DRAIN = decltype(drain_)::value;
        const unsigned char* X = smem + XS * STG;
        const unsigned char* Y = smem + (XS ^ 1) * STG;
        const bool wrap = kt + 2 >= nk;                    // the fetch belongs to the next output tile (or to nothing: a descriptor of length 0)
        const unsigned so = (unsigned)(wrap ? kt + 2 - nk : kt + 2) * 128u;
        f32x4 a0, a1, a2, a3;
        u32x4 ra, rb, lo, hi;
        bf16_t *dlo = nullptr, *dhi = nullptr;
        bool plo = false, phi = false;
#pragma clang loop unroll(full)
        for (int n = 0; n < 64; n++) {
            if (n == 16) {
                OD_WAIT_LGKMCNT(0);
                od_barrier_raw();                          // RELEASE: stage X is in everybody's registers
                sa = wrap ? sa_nxt : sa_cur; sw = wrap ? sw_nxt : sw_cur;
            }
            if (n == 44) {
                OD_WAIT_VMCNT(9);                          // everything but this tile's first nine pieces: all of K tile kt + 1 (and older stores)
                od_barrier_raw();                          // LANDED
            }
            const bool d = n >= 17 && (n - 17) % 3 == 0 && (n - 17) / 3 < 12;
            const int q = (n - 17) / 3;
            if (d) dma_dst((unsigned)XS * STG, q);
            if (n < 12) rd_seq(X, 1, n);
            if (n >= 44 && n < 56) rd_seq(Y, 0, n - 44);
            if constexpr (DRAIN) {
                // the drain of unit J, a few vector-ALU operations per MFMA gap, away from the gaps that carry a DMA piece (n = 2 mod 3)
                if (n == 18) a0 = acc[S ^ 1][0][J];
                if (n == 19) acc[S ^ 1][0][J] = bvn[0];
                if (n == 21) a1 = acc[S ^ 1][1][J];
                if (n == 22) acc[S ^ 1][1][J] = bvn[1];
                if (n == 24) { ra[0] = od_pack_bf2(a0[0], a0[1]); ra[1] = od_pack_bf2(a0[2], a0[3]); ra[2] = od_pack_bf2(a1[0], a1[1]); ra[3] = od_pack_bf2(a1[2], a1[3]); }
                if (n == 25) a2 = acc[S ^ 1][2][J];
                if (n == 27) acc[S ^ 1][2][J] = bvn[2];
                if (n == 28) a3 = acc[S ^ 1][3][J];
                if (n == 30) acc[S ^ 1][3][J] = bvn[3];
                if (n == 31) { rb[0] = od_pack_bf2(a2[0], a2[1]); rb[1] = od_pack_bf2(a2[2], a2[3]); rb[2] = od_pack_bf2(a3[0], a3[1]); rb[3] = od_pack_bf2(a3[2], a3[3]); }
                if (n == 33) {
#pragma unroll
                    for (int i = 0; i < 4; i++) lo[i] = od_dpp_up8(ra[i], rb[i]);
                }
                if (n == 34) {
#pragma unroll
                    for (int i = 0; i < 4; i++) hi[i] = od_dpp_down8(ra[i], rb[i]);
                }
                if (n == 36) {
                    // whole 128-byte lines (see gemm_nt_w4_kernel): instruction one writes rows 0..7 of the 16, instruction two rows 8..15
                    const int gm_lo = mp + wm * 128 + J * 16 + xr, gm_hi = gm_lo + 8;
                    const int gn_lo = np + wn * 64 + 32 * xh + 8 * g, gn_hi = np + wn * 64 + 32 * (1 - xh) + 8 * g;
                    plo = have_prev && gn_lo < N && gm_lo < M; phi = have_prev && gn_hi < N && gm_hi < M;
                    dlo = C + (size_t)(plo ? gm_lo : 0) * ldc + (plo ? gn_lo : 0);
                    dhi = C + (size_t)(phi ? gm_hi : 0) * ldc + (phi ? gn_hi : 0);
                }
                if (n == 57 && plo && !(OD_W4D_X & 1)) { if (nt_store) od_st16_nt(dlo, lo); else *(u32x4*)dlo = lo; }
                if (n == 60 && phi && !(OD_W4D_X & 1)) { if (nt_store) od_st16_nt(dhi, hi); else *(u32x4*)dhi = hi; }
            }
            {
                const int sl = n >> 5, i = (n & 31) >> 3, j = n & 7;
#if defined(OD_EMU)
                acc[S][i][j] = od_mma(fw[sl][i], fa[sl][j], acc[S][i][j]);
#else
                asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[S][i][j]) : "v"(fw[sl][i].v), "v"(fa[sl][j].v));
#endif
            }
            if (d) dma_piece(sa, sw, q, so);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    using std::integral_constant;
    using std::true_type;
    using std::false_type;
    // one output tile into set S: the first eight K tiles drain the other set, the rest run bare
    auto out_tile = [&](auto s_) TNW4_INLINE {
        ktile(0, s_, integral_constant<int, 0>{}, true_type{}, integral_constant<int, 0>{});
        ktile(1, s_, integral_constant<int, 1>{}, true_type{}, integral_constant<int, 1>{});
        ktile(2, s_, integral_constant<int, 0>{}, true_type{}, integral_constant<int, 2>{});
        ktile(3, s_, integral_constant<int, 1>{}, true_type{}, integral_constant<int, 3>{});
        ktile(4, s_, integral_constant<int, 0>{}, true_type{}, integral_constant<int, 4>{});
        ktile(5, s_, integral_constant<int, 1>{}, true_type{}, integral_constant<int, 5>{});
        ktile(6, s_, integral_constant<int, 0>{}, true_type{}, integral_constant<int, 6>{});
        ktile(7, s_, integral_constant<int, 1>{}, true_type{}, integral_constant<int, 7>{});
        for (int kt = 8; kt < nk; kt += 2) {               // K % 128 == 0 (launcher)
            ktile(kt, s_, integral_constant<int, 0>{}, false_type{}, integral_constant<int, 0>{});
            ktile(kt + 1, s_, integral_constant<int, 1>{}, false_type{}, integral_constant<int, 0>{});
        }
    };
    // the set of the LAST output tile, drained with nothing beside it
    auto drain_all = [&](auto s_) TNW4_INLINE {
        constexpr int S = decltype(s_)::value;
#if !defined(OD_EMU)
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");   // the last MFMAs' results before the accumulators are read (asm MFMAs are invisible to the hazard pass)
#endif
#pragma unroll
        for (int j = 0; j < 8; j++) {
            u32x4 ra, rb, lo, hi;
#pragma unroll
            for (int h2 = 0; h2 < 2; h2++) {
                const f32x4 e0 = acc[S][2 * h2][j], e1 = acc[S][2 * h2 + 1][j];
                u32x4& rr = h2 ? rb : ra;
                rr[0] = od_pack_bf2(e0[0], e0[1]); rr[1] = od_pack_bf2(e0[2], e0[3]); rr[2] = od_pack_bf2(e1[0], e1[1]); rr[3] = od_pack_bf2(e1[2], e1[3]);
            }
#pragma unroll
            for (int i = 0; i < 4; i++) { lo[i] = od_dpp_up8(ra[i], rb[i]); hi[i] = od_dpp_down8(ra[i], rb[i]); }
            const int gm_lo = mp + wm * 128 + j * 16 + xr, gm_hi = gm_lo + 8;
            const int gn_lo = np + wn * 64 + 32 * xh + 8 * g, gn_hi = np + wn * 64 + 32 * (1 - xh) + 8 * g;
            if (gn_lo < N && gm_lo < M) { T* dst = C + (size_t)gm_lo * ldc + gn_lo; if (nt_store) od_st16_nt(dst, lo); else *(u32x4*)dst = lo; }
            if (gn_hi < N && gm_hi < M) { T* dst = C + (size_t)gm_hi * ldc + gn_hi; if (nt_store) od_st16_nt(dst, hi); else *(u32x4*)dst = hi; }
        }
    };
    for (int v = 0;; v += 2) {
        // even tile -> set 0 (draining set 1 = tile v - 1), odd tile -> set 1
        int m1, n1;
        bool more = tile_at(v + 1, m1, n1);
        sa_nxt = srd_of(false, more ? m1 : 0, more); sw_nxt = srd_of(true, more ? n1 : 0, more);
        load_bias(more ? n1 : n0, bvn);                      // set 1 restarts at tile v + 1's bias
        out_tile(integral_constant<int, 0>{});
        mp = m0; np = n0; have_prev = true;
        if (!more) { drain_all(integral_constant<int, 0>{}); break; }
        m0 = m1; n0 = n1; sa_cur = sa_nxt; sw_cur = sw_nxt;
        more = tile_at(v + 2, m1, n1);
        sa_nxt = srd_of(false, more ? m1 : 0, more); sw_nxt = srd_of(true, more ? n1 : 0, more);
        load_bias(more ? n1 : n0, bvn);                      // set 0 restarts at tile v + 2's bias
        out_tile(integral_constant<int, 1>{});
        mp = m0; np = n0;
        if (!more) { drain_all(integral_constant<int, 1>{}); break; }
        m0 = m1; n0 = n1; sa_cur = sa_nxt; sw_cur = sw_nxt;
    }
    OD_WAIT_VMCNT(0);
}

#endif
