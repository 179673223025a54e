// Boundary/layout kernels, depthwise Conv1d, per-sample linears, weight packing.
#include "od_common.h"
#include "od_api_internal.h"
#include <type_traits>

namespace {

// ------------------------------------------------------------ (B,C,L) f32 -> frame-major rows
template <class T>
__global__ __launch_bounds__(256) void cl_to_frames_kernel(const float* __restrict__ src, T* __restrict__ dst, int ldd, int C, int L) {
    __shared__ float tile[64][65];
    const int b = blockIdx.z, l0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int i = ty; i < 64; i += 4) {               // read: lanes along frames (contiguous)
        const int c = c0 + i, l = l0 + tx;
        tile[i][tx] = (c < C && l < L) ? src[((size_t)b * C + c) * L + l] : 0.f;
    }
    __syncthreads();
    for (int i = ty; i < 64; i += 4) {               // write: lanes along channels (contiguous)
        const int l = l0 + i, c = c0 + tx;
        if (l < L && c < C) od_t<T>::st(dst + ((size_t)b * L + l) * ldd + c, tile[tx][i]);
    }
}

// ------------------------------------------------------------ proj_in: E (<=8) channels -> D
template <class T>
__global__ __launch_bounds__(256) void proj_in_kernel(const float* __restrict__ xt, const float* __restrict__ W,
                                                      const float* __restrict__ bias, T* __restrict__ x, int ldx,
                                                      int B, int E, int L, int D) {
    const int lane = threadIdx.x & 63;
    const long M = (long)B * L;
    const long nw = (long)gridDim.x * 4, w0 = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    for (int c = lane * 8; c < D; c += 512) {
        float wv[8][8], bv[8];
        od_ld8(bias + c, bv);
        if (E == 6) {
            // the lane's 8 x 6 weights are 48 consecutive floats (192 B, 16-byte aligned): six 32-byte loads instead of 48 scalar ones
            // (at the sampler's size the kernel is this prologue: 17 -> 8 us)
            float flat[48];
#pragma unroll
            for (int q = 0; q < 6; q++) od_ld8(W + (size_t)c * 6 + q * 8, *(float(*)[8])&flat[q * 8]);
#pragma unroll
            for (int k = 0; k < 8; k++)
#pragma unroll
                for (int e = 0; e < 8; e++) wv[k][e] = e < 6 ? flat[k * 6 + e] : 0.f;
        } else {
#pragma unroll
            for (int k = 0; k < 8; k++)
#pragma unroll
                for (int e = 0; e < 8; e++) wv[k][e] = e < E ? W[(size_t)(c + k) * E + e] : 0.f;
        }
        for (long m = w0; m < M; m += nw) {
            const int b = (int)(m / L), l = (int)(m % L);
            float xe[8];
#pragma unroll
            for (int e = 0; e < 8; e++) xe[e] = e < E ? xt[((size_t)b * E + e) * L + l] : 0.f;
            float o[8];
#pragma unroll
            for (int k = 0; k < 8; k++) {
                float s = bv[k];
#pragma unroll
                for (int e = 0; e < 8; e++) s += wv[k][e] * xe[e];
                o[k] = s;
            }
            od_st8(x + m * ldx + c, o);
        }
    }
}

// dW[c][e] += sum_m dx[m][c] xt[b][e][l], db[c] += sum_m dx[m][c].  A wave walks frames with 8 channels per lane
// (D <= 512 per pass); the four waves of a block are summed in LDS and each block issues ONE atomic per output
// element (all blocks hit the same D*(E+1) addresses, so the atomic count per address is what costs).
template <class T>
__global__ __launch_bounds__(256) void proj_in_bwd_kernel(const float* __restrict__ xt, const T* __restrict__ dx, int ldx,
                                                          float* __restrict__ dW, float* __restrict__ db, int B, int E, int L, int D,
                                                          const OdDetTable* __restrict__ det) {
    __shared__ float red[64][73];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int M = B * L;
    const int nw = gridDim.x * 4, w0 = blockIdx.x * 4 + wave;
    for (int c0 = 0; c0 < D; c0 += 512) {
        const int c = c0 + lane * 8;
        float aw[8][8], ab[8];
#pragma unroll
        for (int k = 0; k < 8; k++) {
            ab[k] = 0.f;
#pragma unroll
            for (int e = 0; e < 8; e++) aw[k][e] = 0.f;
        }
        if (c < D)
            for (int m = w0; m < M; m += nw) {
                const int b = m / L, l = m - b * L;
                float xe[8], g[8];
#pragma unroll
                for (int e = 0; e < 8; e++) xe[e] = e < E ? xt[((size_t)b * E + e) * L + l] : 0.f;
                od_ld8(dx + (size_t)m * ldx + c, g);
#pragma unroll
                for (int k = 0; k < 8; k++) {
                    ab[k] += g[k];
#pragma unroll
                    for (int e = 0; e < 8; e++) aw[k][e] += g[k] * xe[e];
                }
            }
        for (int w = 0; w < 4; w++) {
            if (wave == w) {
#pragma unroll
                for (int k = 0; k < 8; k++) {
                    red[lane][k * 9 + 8] = (w ? red[lane][k * 9 + 8] : 0.f) + ab[k];
#pragma unroll
                    for (int e = 0; e < 8; e++) red[lane][k * 9 + e] = (w ? red[lane][k * 9 + e] : 0.f) + aw[k][e];
                }
            }
            __syncthreads();
        }
        if (wave == 0 && c < D) {
#pragma unroll
            for (int k = 0; k < 8; k++) {
                od_red_add(det, db + c + k, red[lane][k * 9 + 8]);
#pragma unroll
                for (int e = 0; e < 8; e++)
                    if (e < E) od_red_add(det, dW + (size_t)(c + k) * E + e, red[lane][k * 9 + e]);
            }
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------ elementwise SiLU
template <class T>
__global__ __launch_bounds__(256) void silu_kernel(const T* __restrict__ x, T* __restrict__ y, long n8) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n8; i += (long)gridDim.x * 256) {
        float v[8]; od_ld8(x + i * 8, v);
#pragma unroll
        for (int e = 0; e < 8; e++) v[e] = od_silu(v[e]);
        od_st8(y + i * 8, v);
    }
}
template <class T>
__global__ __launch_bounds__(256) void silu_bwd_kernel(const T* __restrict__ x, const T* __restrict__ dy, T* __restrict__ dx, long n8) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n8; i += (long)gridDim.x * 256) {
        float v[8], g[8]; od_ld8(x + i * 8, v); od_ld8(dy + i * 8, g);
#pragma unroll
        for (int e = 0; e < 8; e++) g[e] *= od_silu_grad(v[e]);
        od_st8(dx + i * 8, g);
    }
}

// ------------------------------------------------------------ depthwise conv along frames
// thread = 8 channels x a run of RUN frames, sliding a K-tap register window down the frames.
// RUN = 32 for long sequences (halo re-read 1.125x); short workloads (the sampler's B*L = 4460 frames) use
// RUN = 4 so the launch still covers the chip (the halo re-reads hit L2).
template <class T, int KS, int DW_RUN>
__global__ __launch_bounds__(256) void dwconv_kernel(const T* __restrict__ x, int ldx, const float* __restrict__ w,
                                                     const float* __restrict__ bias, T* __restrict__ y, int ldy,
                                                     int L, int C) {
    constexpr int R = KS / 2;
    const int cg = C / 8;
    const int b = blockIdx.y;
    const long t = (long)blockIdx.x * 256 + threadIdx.x;
    const int cgi = (int)(t % cg);
    const long run = t / cg;
    const int l0 = (int)(run * DW_RUN);
    if (l0 >= L) return;
    const int c = cgi * 8;
    float wv[8][KS], bv[8];
#pragma unroll
    for (int k = 0; k < 8; k++) {
        bv[k] = bias[c + k];
#pragma unroll
        for (int j = 0; j < KS; j++) wv[k][j] = w[(size_t)(c + k) * KS + j];
    }
    float win[KS][8];
    const T* xb = x + (size_t)b * L * ldx + c;
#pragma unroll
    for (int j = 0; j < KS - 1; j++) {
        const int l = l0 - R + j;
        if (l >= 0 && l < L) od_ld8(xb + (size_t)l * ldx, win[j]);
        else {
#pragma unroll
            for (int k = 0; k < 8; k++) win[j][k] = 0.f;
        }
    }
    for (int i = 0; i < DW_RUN; i++) {
        const int l = l0 + i;
        if (l >= L) break;
        const int ln = l + R;
        if (ln < L) od_ld8(xb + (size_t)ln * ldx, win[KS - 1]);
        else {
#pragma unroll
            for (int k = 0; k < 8; k++) win[KS - 1][k] = 0.f;
        }
        float o[8];
#pragma unroll
        for (int k = 0; k < 8; k++) {
            float s = bv[k];
#pragma unroll
            for (int j = 0; j < KS; j++) s += wv[k][j] * win[j][k];
            o[k] = s;
        }
        od_st8(y + ((size_t)b * L + l) * ldy + c, o);
#pragma unroll
        for (int j = 0; j < KS - 1; j++)
#pragma unroll
            for (int k = 0; k < 8; k++) win[j][k] = win[j + 1][k];
    }
}

// backward: dx[l] = sum_j w[j] dy[l - j + R];  dw[j] += sum_l dy[l] x[l + j - R];  db += sum_l dy[l]
// The per-channel weight/bias sums are reduced across the block in LDS before they touch global
// atomics (one atomic per (channel, tap) per block instead of per thread).
constexpr int DW_RUN_BWD = 64;
template <class T, int KS>
__global__ __launch_bounds__(256) void dwconv_bwd_kernel(const T* __restrict__ x, int ldx, const float* __restrict__ w,
                                                         const T* __restrict__ dy, int lddy, T* __restrict__ dx, int lddx,
                                                         float* __restrict__ dw, float* __restrict__ db, int L, int C,
                                                         const OdDetTable* __restrict__ det) {
    constexpr int R = KS / 2;
    // fixed point (od_lds_fix_add): the block's sums do not depend on the order its threads arrive in.  C <= 1024 at k <= 5, 512 above (launcher)
    __shared__ long long red[(KS <= 5 ? 1024 : 512) * (KS + 1)];
    __shared__ int s_bad;
    const int cg = C / 8;
    const int b = blockIdx.y;
    const long t = (long)blockIdx.x * 256 + threadIdx.x;
    const int cgi = (int)(t % cg);
    const long run = t / cg;
    const int l0 = (int)(run * DW_RUN_BWD);
    const bool active = l0 < L;
    const int c = cgi * 8;
    for (int i = threadIdx.x; i < C * (KS + 1); i += 256) red[i] = 0;
    if (threadIdx.x == 0) s_bad = 0;
    __syncthreads();
    float wv[8][KS], adw[8][KS], adb[8];
#pragma unroll
    for (int k = 0; k < 8; k++) {
        adb[k] = 0.f;
#pragma unroll
        for (int j = 0; j < KS; j++) { wv[k][j] = w[(size_t)(c + k) * KS + j]; adw[k][j] = 0.f; }
    }
    if (active) {
        // windows over frames l-R .. l+R of both dy (for dx) and x (for dw)
        float wy[KS][8], wx[KS][8];
        const T* xb = x + (size_t)b * L * ldx + c;
        const T* yb = dy + (size_t)b * L * lddy + c;
#pragma unroll
        for (int j = 0; j < KS - 1; j++) {
            const int l = l0 - R + j;
            if (l >= 0 && l < L) { od_ld8(xb + (size_t)l * ldx, wx[j]); od_ld8(yb + (size_t)l * lddy, wy[j]); }
            else {
#pragma unroll
                for (int k = 0; k < 8; k++) { wx[j][k] = 0.f; wy[j][k] = 0.f; }
            }
        }
        // The rows entering the window are requested DW_PF iterations before they are used (raw 16-byte rows in a small register ring, the
        // loop unrolled by the ring size so that its slots are compile-time): the one-row-per-iteration form consumed each load in the
        // iteration that issued it — two waves per SIMD with two loads in flight each, 43 % of the HBM rate (profiles/r03_pmc_step.txt).
        constexpr int DW_PF = 2;
        typedef typename std::conditional<sizeof(T) == 2, u32x4, f32x4>::type row_t;      // 8 channels as loaded (bf16: one 16-byte word)
        row_t px[DW_PF][sizeof(T) == 2 ? 1 : 2], py[DW_PF][sizeof(T) == 2 ? 1 : 2];
        auto fetch = [&](int ln, int slot) {
            if (ln < L) {
#pragma unroll
                for (int h2 = 0; h2 < (sizeof(T) == 2 ? 1 : 2); h2++) {
                    px[slot][h2] = *(const row_t*)(xb + (size_t)ln * ldx + h2 * 4);
                    py[slot][h2] = *(const row_t*)(yb + (size_t)ln * lddy + h2 * 4);
                }
            } else {
#pragma unroll
                for (int h2 = 0; h2 < (sizeof(T) == 2 ? 1 : 2); h2++) { px[slot][h2] = (row_t)(0); py[slot][h2] = (row_t)(0); }
            }
        };
        auto unpack = [&](const row_t (&r)[sizeof(T) == 2 ? 1 : 2], float (&o8)[8]) {
            if constexpr (sizeof(T) == 2) {
#pragma unroll
                for (int i2 = 0; i2 < 4; i2++) {
                    union { uint32_t u; float f; } lo, hi;
                    lo.u = r[0][i2] << 16; hi.u = r[0][i2] & 0xffff0000u;
                    o8[2 * i2] = lo.f; o8[2 * i2 + 1] = hi.f;
                }
            } else {
#pragma unroll
                for (int i2 = 0; i2 < 4; i2++) { o8[i2] = r[0][i2]; o8[4 + i2] = r[1][i2]; }
            }
        };
#pragma unroll
        for (int pf = 0; pf < DW_PF; pf++) fetch(l0 + R + pf, pf);
        for (int i0 = 0; i0 < DW_RUN_BWD; i0 += DW_PF) {
          if (l0 + i0 >= L) break;
#pragma unroll
          for (int pf = 0; pf < DW_PF; pf++) {
            const int i = i0 + pf;
            const int l = l0 + i;
            if (l >= L) break;
            unpack(px[pf], wx[KS - 1]);
            unpack(py[pf], wy[KS - 1]);
            fetch(l + R + DW_PF, pf);
            float o[8];
#pragma unroll
            for (int k = 0; k < 8; k++) {
                float sacc = 0.f;
                // window slot j holds frame l - R + j; dx[l] += w[jj] * dy[l - jj + R]  ->  slot = 2R - jj
#pragma unroll
                for (int jj = 0; jj < KS; jj++) sacc += wv[k][jj] * wy[KS - 1 - jj][k];
                o[k] = sacc;
                const float gy = wy[R][k];     // dy[l]
                adb[k] += gy;
#pragma unroll
                for (int jj = 0; jj < KS; jj++) adw[k][jj] += gy * wx[jj][k];   // x[l + jj - R]
            }
            od_st8(dx + ((size_t)b * L + l) * lddx + c, o);
#pragma unroll
            for (int j = 0; j < KS - 1; j++)
#pragma unroll
                for (int k = 0; k < 8; k++) { wx[j][k] = wx[j + 1][k]; wy[j][k] = wy[j + 1][k]; }
          }
        }
#pragma unroll
        for (int k = 0; k < 8; k++) {
            od_lds_fix_add(&red[(c + k) * (KS + 1) + KS], adb[k], &s_bad);
#pragma unroll
            for (int j = 0; j < KS; j++) od_lds_fix_add(&red[(c + k) * (KS + 1) + j], adw[k][j], &s_bad);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < C * (KS + 1); i += 256) {
        const int ch = i / (KS + 1), j = i % (KS + 1);
        const float v = od_lds_unfix(red[i], s_bad);
        if (v != 0.f) {
            if (j == KS) od_red_add(det, db + ch, v); else od_red_add(det, dw + (size_t)ch * KS + j, v);
        }
    }
}

// ------------------------------------------------------------ per-sample linears (ssg1 / ssg2 / u_mod / proj_style: B <= a few dozen rows)
// Round 3: these were one wave per output feature walking the batch serially (a dependent load -> 6-shuffle reduction chain per sample:
// 72 us for the 32 x 512 -> 1536 ssg linear, 84 + 27 us for its backward; 3.3 ms of every training step and 1.2 ms of every sampler call).
// Now: forward = a 32 (samples) x 8 (features) output tile per block, both operands staged through LDS in 256-deep K chunks, one thread
// per output (no cross-lane reduction, W read once); dW = one thread per weight element, the batch in registers (dpre[b][n] is block-uniform:
// scalar loads); dx = 32 x 256 output tiles over 32-row slices of N (W read once, coalesced), fp32 atomics into dx.
constexpr int LS_KC = 256, LS_NB = 8;
__global__ __launch_bounds__(256) void linear_small_kernel(const float* __restrict__ x, const float* __restrict__ W,
                                                           const float* __restrict__ bias, float* __restrict__ out,
                                                           float* __restrict__ pre, int B, int N, int K, int act) {
    __shared__ __attribute__((aligned(16))) float xs[32][LS_KC + 4];
    __shared__ __attribute__((aligned(16))) float ws[LS_NB][LS_KC + 4];
    const int b0 = blockIdx.y * 32, n0 = blockIdx.x * LS_NB;
    const int bl = threadIdx.x & 31, nl = threadIdx.x >> 5;
    float acc = 0.f;
    for (int k0 = 0; k0 < K; k0 += LS_KC) {
        const int kc = K - k0 < LS_KC ? K - k0 : LS_KC;
        for (int i = threadIdx.x; i < 32 * LS_KC; i += 256) {
            const int r = i / LS_KC, c = i % LS_KC;
            xs[r][c] = (b0 + r < B && c < kc) ? x[(size_t)(b0 + r) * K + k0 + c] : 0.f;
        }
        for (int i = threadIdx.x; i < LS_NB * LS_KC; i += 256) {
            const int r = i / LS_KC, c = i % LS_KC;
            ws[r][c] = (n0 + r < N && c < kc) ? W[(size_t)(n0 + r) * K + k0 + c] : 0.f;
        }
        __syncthreads();
#pragma unroll 8
        for (int c = 0; c < LS_KC; c += 4) {
            const f32x4 a = *(const f32x4*)&xs[bl][c], w = *(const f32x4*)&ws[nl][c];
            acc += a[0] * w[0]; acc += a[1] * w[1]; acc += a[2] * w[2]; acc += a[3] * w[3];
        }
        __syncthreads();
    }
    const int b = b0 + bl, n = n0 + nl;
    if (b < B && n < N) {
        const float s = acc + (bias ? bias[n] : 0.f);
        if (pre) pre[(size_t)b * N + n] = s;
        out[(size_t)b * N + n] = act == OD_ACT_SILU ? od_silu(s) : s;
    }
}
__global__ __launch_bounds__(256) void linear_small_dpre_kernel(const float* __restrict__ pre, const float* __restrict__ dout,
                                                                float* __restrict__ dpre, long n, int act) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    dpre[i] = act == OD_ACT_SILU ? dout[i] * od_silu_grad(pre[i]) : dout[i];
}
// dW[n][k] += sum_b dpre[b][n] x[b][k]; db[n] += sum_b dpre[b][n]     (block = 256 consecutive k of ONE n: dpre[b][n] is block-uniform)
__global__ __launch_bounds__(256) void linear_small_dw_kernel(const float* __restrict__ x, const float* __restrict__ dpre,
                                                              float* __restrict__ dW, float* __restrict__ db, int B, int N, int K) {
    const int n = blockIdx.y, k = blockIdx.x * 256 + threadIdx.x;
    if (dW && k < K) {
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        int b = 0;
        for (; b + 3 < B; b += 4) {          // four independent loads in flight
            s0 += dpre[(size_t)b * N + n] * x[(size_t)b * K + k];
            s1 += dpre[(size_t)(b + 1) * N + n] * x[(size_t)(b + 1) * K + k];
            s2 += dpre[(size_t)(b + 2) * N + n] * x[(size_t)(b + 2) * K + k];
            s3 += dpre[(size_t)(b + 3) * N + n] * x[(size_t)(b + 3) * K + k];
        }
        for (; b < B; b++) s0 += dpre[(size_t)b * N + n] * x[(size_t)b * K + k];
        dW[(size_t)n * K + k] += (s0 + s1) + (s2 + s3);
    }
    if (db && blockIdx.x == 0 && threadIdx.x == 0) {
        float s = 0.f;
        for (int b = 0; b < B; b++) s += dpre[(size_t)b * N + n];
        db[n] += s;
    }
}
// dx[b][k] += sum_n dpre[b][n] W[n][k]: block = 32 samples x 256 columns over a 32-row slice of N, one thread per column with the 32
// sample accumulators in registers; the dpre tile is staged transposed ([n][b]) so a thread reads four samples per (broadcast) LDS read;
// the slices meet in dx through fp32 atomics (dx is zeroed by the launcher when it does not accumulate): the style / global-conditioning
// gradients (ssg1 / ssg2 / u_mod / proj_style backward) are therefore reproducible to fp32 rounding, not bit for bit, run to run — like the
// weight-gradient GEMMs' split-M sums (tests/test_ddp_rccl.py holds a DDP step to 1e-4 of a plain step for that reason).
constexpr int LS_NR = 32;
__global__ __launch_bounds__(256) void linear_small_dx_kernel(const float* __restrict__ W, const float* __restrict__ dpre,
                                                              float* __restrict__ dx, int B, int N, int K, const OdDetTable* __restrict__ det) {
    __shared__ __attribute__((aligned(16))) float ds[LS_NR][32];
    const int n0 = blockIdx.x * LS_NR, b0 = blockIdx.y * 32, k = blockIdx.z * 256 + threadIdx.x;
    for (int i = threadIdx.x; i < LS_NR * 32; i += 256) {
        const int bl = i / LS_NR, nl = i % LS_NR;           // consecutive threads: consecutive n of one sample (coalesced)
        ds[nl][bl] = (b0 + bl < B && n0 + nl < N) ? dpre[(size_t)(b0 + bl) * N + n0 + nl] : 0.f;
    }
    __syncthreads();
    if (k >= K) return;
    float acc[32];
#pragma unroll
    for (int i = 0; i < 32; i++) acc[i] = 0.f;
    const int nr = N - n0 < LS_NR ? N - n0 : LS_NR;
    for (int nl = 0; nl < nr; nl++) {
        const float w = W[(size_t)(n0 + nl) * K + k];
#pragma unroll
        for (int q4 = 0; q4 < 8; q4++) {
            const f32x4 d = *(const f32x4*)&ds[nl][4 * q4];
            acc[4 * q4] += d[0] * w; acc[4 * q4 + 1] += d[1] * w; acc[4 * q4 + 2] += d[2] * w; acc[4 * q4 + 3] += d[3] * w;
        }
    }
#pragma unroll
    for (int i = 0; i < 32; i++)
        if (b0 + i < B) od_red_add(det, dx + (size_t)(b0 + i) * K + k, acc[i]);
}

__global__ __launch_bounds__(256) void zero_f32_kernel(float* __restrict__ p, long n) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) p[i] = 0.f;
}

// ------------------------------------------------------------ fp32 master weight -> packed compute copy
template <class T>
__global__ __launch_bounds__(256) void pack_weight_kernel(const float* __restrict__ src, int N, int K, T* __restrict__ dst,
                                                          int Np, int Kp, int transpose, const int* __restrict__ row_map) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)Np * Kp) return;
    int n, k;
    if (!transpose) { n = (int)(i / Kp); k = (int)(i % Kp); }     // dst[n][k]
    else { k = (int)(i / Np); n = (int)(i % Np); }                // dst[k][n]
    const int sn = row_map ? row_map[n] : (n < N ? n : -1);
    const float v = (sn >= 0 && k < K) ? src[(size_t)sn * K + k] : 0.f;
    od_t<T>::st(dst + i, v);
}

// fp32 master -> (hi, lo) bf16 halves, interleaved per 32-element K slab: dst[n][32 s + j] as storage holds, for slab s of row n, the 32 high
// halves in its first 64 bytes and the 32 low halves in the next 64 (what frag_w_from_lds<f32x3w_t> in gemm.hip reads)
__global__ __launch_bounds__(256) void pack_weight_split_kernel(const float* __restrict__ src, int N, int K, bf16_t* __restrict__ dst,
                                                                int Np, int Kp, const int* __restrict__ row_map) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)Np * Kp) return;
    const int n = (int)(i / Kp), k = (int)(i % Kp);
    const int sn = row_map ? row_map[n] : (n < N ? n : -1);
    const float v = (sn >= 0 && k < K) ? src[(size_t)sn * K + k] : 0.f;
    const bf16_t hi = od_f2bf(v);
    const bf16_t lo = od_f2bf(v - od_bf2f(hi));
    bf16_t* slab = dst + ((size_t)n * Kp + (k & ~31)) * 2;      // 32 floats = 64 bf16 slots
    slab[k & 31] = hi;
    slab[32 + (k & 31)] = lo;
}

}  // namespace

#define DISPATCH_T(DT, CALL)                                       \
    do {                                                           \
        if ((DT) == OD_BF16) { typedef bf16_t T_; CALL; }          \
        else if ((DT) == OD_F32) { typedef float T_; CALL; }       \
        else return OD_ERR_ARG;                                    \
    } while (0)

extern "C" int od_cl_to_frames(int dtype, const float* src, void* dst, int ldd, int B, int C, int L, void* stream) {
    dim3 grid((L + 63) / 64, (C + 63) / 64, B);
    DISPATCH_T(dtype, OD_LAUNCH((cl_to_frames_kernel<T_>), grid, dim3(256), 0, (hipStream_t)stream, src, (T_*)dst, ldd, C, L));
    OD_CHECK_LAUNCH();
    return 0;
}

extern "C" int od_proj_in(int dtype, const float* xt, const float* W, const float* bias, void* x, int ldx, int B, int E, int L,
                          int D, void* stream) {
    if (E > 8) return OD_ERR_UNSUPPORTED;
    if (D % 8 || ldx % 8) return OD_ERR_ALIGN;
    const long M = (long)B * L;
    // a thread loads its 8 x E weights once and then streams frames: give every wave >= 4 frames
    int blocks = (int)((M + 15) / 16); if (blocks > 4096) blocks = 4096;
    DISPATCH_T(dtype, OD_LAUNCH((proj_in_kernel<T_>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, xt, W, bias, (T_*)x, ldx, B, E, L, D));
    OD_CHECK_LAUNCH();
    return 0;
}

extern "C" int od_proj_in_bwd(int dtype, const float* xt, const void* dx, int ldx, float* dW, float* db, int B, int E, int L, int D,
                              void* stream) {
    if (E > 8) return OD_ERR_UNSUPPORTED;
    if (D % 8 || ldx % 8) return OD_ERR_ALIGN;
    const long M = (long)B * L;
    if (M > 0x7fffffffL) return OD_ERR_UNSUPPORTED;
    int blocks = (int)((M + 63) / 64); if (blocks > 512) blocks = 512; if (blocks < 1) blocks = 1;
    DISPATCH_T(dtype, OD_LAUNCH((proj_in_bwd_kernel<T_>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, xt, (const T_*)dx, ldx, dW, db, B, E, L, D, od_det_active()));
    OD_CHECK_LAUNCH();
    return 0;
}

extern "C" int od_silu(int dtype, const void* x, void* y, long n, void* stream) {
    if (n % 8) return OD_ERR_ALIGN;
    long n8 = n / 8; int blocks = (int)((n8 + 255) / 256); if (blocks > 4096) blocks = 4096; if (blocks < 1) blocks = 1;
    DISPATCH_T(dtype, OD_LAUNCH((silu_kernel<T_>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const T_*)x, (T_*)y, n8));
    OD_CHECK_LAUNCH();
    return 0;
}
extern "C" int od_silu_bwd(int dtype, const void* x, const void* dy, void* dx, long n, void* stream) {
    if (n % 8) return OD_ERR_ALIGN;
    long n8 = n / 8; int blocks = (int)((n8 + 255) / 256); if (blocks > 4096) blocks = 4096; if (blocks < 1) blocks = 1;
    DISPATCH_T(dtype, OD_LAUNCH((silu_bwd_kernel<T_>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const T_*)x, (const T_*)dy, (T_*)dx, n8));
    OD_CHECK_LAUNCH();
    return 0;
}

// x[(b, l)][c] *= scale[b][c] in place — nn.Dropout1d in training mode (common/swiglu.py:23,30): whole channels of a sample are
// zeroed with probability p and the survivors scaled by 1 / (1 - p); the host draws the (B, C) factors.  Its own backward.
template <class T>
__global__ __launch_bounds__(256) void scale_channels_kernel(T* __restrict__ x, int ldx, const float* __restrict__ scale, long M, int L, int C) {
    const int cg = C / 8;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= M * cg) return;
    const long m = i / cg;
    const int c = (int)(i % cg) * 8;
    float v[8], f[8];
    od_ld8(x + m * ldx + c, v);
    od_ld8(scale + (m / L) * C + c, f);
#pragma unroll
    for (int e = 0; e < 8; e++) v[e] *= f[e];
    od_st8(x + m * ldx + c, v);
}

extern "C" int od_scale_channels(int dtype, void* x, int ldx, const float* scale, int B, int L, int C, void* stream) {
    if (C % 8 || ldx % 8) return OD_ERR_ALIGN;
    const long M = (long)B * L, n = M * (C / 8);
    if (n == 0) return 0;
    DISPATCH_T(dtype, OD_LAUNCH((scale_channels_kernel<T_>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (T_*)x, ldx, scale, M, L, C));
    OD_CHECK_LAUNCH();
    return 0;
}

extern "C" int od_dwconv(int dtype, const void* x, int ldx, const float* w, const float* bias, void* y, int ldy, int B, int L,
                         int C, int ksize, void* stream) {
    if (C % 8 || ldx % 8 || ldy % 8) return OD_ERR_ALIGN;
    if (ksize != 3 && ksize != 5 && ksize != 7 && ksize != 9) return OD_ERR_UNSUPPORTED;
#ifndef OD_DW_SMALL_THREADS
#define OD_DW_SMALL_THREADS 131072      // < 2 workgroups per CU at RUN = 32 (the emulator build lowers it to reach both paths)
#endif
    const bool small = (long)B * (C / 8) * ((L + 31) / 32) < OD_DW_SMALL_THREADS;
    const int run = small ? 4 : 32;
    const long threads = (long)(C / 8) * ((L + run - 1) / run);
    dim3 grid((unsigned)((threads + 255) / 256), B);
#define DW_GO(KS_, RUN_) DISPATCH_T(dtype, OD_LAUNCH((dwconv_kernel<T_, KS_, RUN_>), grid, dim3(256), 0, (hipStream_t)stream, (const T_*)x, ldx, w, bias, (T_*)y, ldy, L, C))
    if (ksize == 5) { if (small) DW_GO(5, 4); else DW_GO(5, 32); }
    else if (ksize == 3) { if (small) DW_GO(3, 4); else DW_GO(3, 32); }
    else if (ksize == 7) { if (small) DW_GO(7, 4); else DW_GO(7, 32); }      // radius 3, 4: not a shipped config, same kernel
    else { if (small) DW_GO(9, 4); else DW_GO(9, 32); }
#undef DW_GO
    OD_CHECK_LAUNCH();
    return 0;
}

extern "C" int od_dwconv_bwd(int dtype, const void* x, int ldx, const float* w, const void* dy, int lddy, void* dx, int lddx,
                             float* dw, float* db, int B, int L, int C, int ksize, void* stream) {
    if (C % 8 || ldx % 8 || lddy % 8 || lddx % 8) return OD_ERR_ALIGN;
    if (C > 1024 || (ksize > 5 && C > 512)) return OD_ERR_UNSUPPORTED;
    const long threads = (long)(C / 8) * ((L + DW_RUN_BWD - 1) / DW_RUN_BWD);
    dim3 grid((unsigned)((threads + 255) / 256), B);
    if (ksize == 5) DISPATCH_T(dtype, OD_LAUNCH((dwconv_bwd_kernel<T_, 5>), grid, dim3(256), 0, (hipStream_t)stream, (const T_*)x, ldx, w, (const T_*)dy, lddy, (T_*)dx, lddx, dw, db, L, C, od_det_active()));
    else if (ksize == 3) DISPATCH_T(dtype, OD_LAUNCH((dwconv_bwd_kernel<T_, 3>), grid, dim3(256), 0, (hipStream_t)stream, (const T_*)x, ldx, w, (const T_*)dy, lddy, (T_*)dx, lddx, dw, db, L, C, od_det_active()));
    else if (ksize == 7 && C <= 512) DISPATCH_T(dtype, OD_LAUNCH((dwconv_bwd_kernel<T_, 7>), grid, dim3(256), 0, (hipStream_t)stream, (const T_*)x, ldx, w, (const T_*)dy, lddy, (T_*)dx, lddx, dw, db, L, C, od_det_active()));
    else if (ksize == 9 && C <= 512) DISPATCH_T(dtype, OD_LAUNCH((dwconv_bwd_kernel<T_, 9>), grid, dim3(256), 0, (hipStream_t)stream, (const T_*)x, ldx, w, (const T_*)dy, lddy, (T_*)dx, lddx, dw, db, L, C, od_det_active()));
    else return OD_ERR_UNSUPPORTED;
    OD_CHECK_LAUNCH();
    return 0;
}

extern "C" int od_linear_small(const float* x, const float* W, const float* b, float* out, float* pre, int B, int N, int K, int act,
                               void* stream) {
    OD_LAUNCH(linear_small_kernel, dim3((N + LS_NB - 1) / LS_NB, (B + 31) / 32), dim3(256), 0, (hipStream_t)stream, x, W, b, out, pre, B, N, K, act);
    OD_CHECK_LAUNCH();
    return 0;
}

extern "C" int od_linear_small_bwd(const float* x, const float* W, const float* pre, const float* dout, float* dpre, float* dW,
                                   float* db, float* dx, int accumulate_dx, int B, int N, int K, int act, void* stream) {
    if (act != OD_ACT_NONE && !pre) return OD_ERR_ARG;
    const long n = (long)B * N;
    OD_LAUNCH(linear_small_dpre_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, pre, dout, dpre, n, act);
    if (dW || db)
        OD_LAUNCH(linear_small_dw_kernel, dim3((K + 255) / 256, N), dim3(256), 0, (hipStream_t)stream, x, (const float*)dpre, dW, db, B, N, K);
    if (dx) {
        if (!accumulate_dx)
            OD_LAUNCH(zero_f32_kernel, dim3((unsigned)(((long)B * K + 255) / 256)), dim3(256), 0, (hipStream_t)stream, dx, (long)B * K);
        OD_LAUNCH(linear_small_dx_kernel, dim3((N + LS_NR - 1) / LS_NR, (B + 31) / 32, (K + 255) / 256), dim3(256), 0, (hipStream_t)stream, W,
                  (const float*)dpre, dx, B, N, K, od_det_active());
    }
    OD_CHECK_LAUNCH();
    return 0;
}

extern "C" int od_pack_weight(int dtype, const float* src, int N, int K, void* dst, int Np, int Kp, int transpose,
                              const int* row_map, void* stream) {
    const long n = (long)Np * Kp;
    if (dtype == OD_F32X3W) {
        if (transpose || Kp % 32) return OD_ERR_UNSUPPORTED;
        OD_LAUNCH(pack_weight_split_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, src, N, K, (bf16_t*)dst, Np, Kp, row_map);
        OD_CHECK_LAUNCH();
        return 0;
    }
    DISPATCH_T(dtype, OD_LAUNCH((pack_weight_kernel<T_>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, src, N, K, (T_*)dst, Np, Kp, transpose, row_map));
    OD_CHECK_LAUNCH();
    return 0;
}
