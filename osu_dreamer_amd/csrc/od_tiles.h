// Swizzled LDS tile addressing and MFMA fragment readers shared by the attention and GEMM kernels.
#pragma once
#include "od_common.h"

// ---- swizzled LDS tile addressing: 16-byte slots XORed with the row index ---------------
template <int ROWB>
__device__ __forceinline__ int tile_off(int row, int byte) {
    constexpr int NS = ROWB / 16;
    return row * ROWB + ((((byte >> 4)) ^ (row & (NS - 1))) << 4) + (byte & 15);
}

// Swizzle for 128-byte-row bf16 tiles of the 32x32x16 attention kernels, which read one image BOTH as 32-row ds_read_b128
// fragments (a 16-lane service group touches 16 rows that are distinct mod 16) and as 32-lane-wide ds_read_b64_tr_b16 chunks
// (4 consecutive rows x 64 bytes).  The 16-byte slot index is XORed with the bit-reversed row bits 3..1:
//   * rows equal mod 16 only share (bank half = row & 1, slot)                      -> the b128 reads are conflict-free
//   * rows r and r + 2 (same bank half) land in different groups of four slots      -> so are the transpose reads
// (the 16-row kernels' key `row & 7` puts rows r and r + 8, resp. the two row pairs of a 64-byte-wide chunk, on the same banks:
// SQ_LDS_BANK_CONFLICT was a third of the LDS cycles of the first 32x32 forward, profiles/r02d_pmc_attn.txt)
__device__ __forceinline__ int swz32(int row) { return (((row >> 1) & 1) << 2) | (((row >> 2) & 1) << 1) | ((row >> 3) & 1); }
__device__ __forceinline__ int tile32_off(int row, int byte) { return row * 128 + ((((byte >> 4)) ^ swz32(row)) << 4) + (byte & 15); }

// 8 k-contiguous elements of `row` starting at element k0 (multiple of 8)
template <int ROWB>
__device__ __forceinline__ void frag_contig(od_frag<bf16_t>& f, const unsigned char* t, int row, int k0) {
    f.v = *(const s16x8*)(t + tile_off<ROWB>(row, k0 * 2));
}
template <int ROWB>
__device__ __forceinline__ void frag_contig(od_frag<f16_t>& f, const unsigned char* t, int row, int k0) {
    f.v = *(const s16x8*)(t + tile_off<ROWB>(row, k0 * 2));
}
template <int ROWB, class F>
__device__ __forceinline__ void frag_contig_f32(od_frag<F>& f, const unsigned char* t, int row, int k0) {
    const f32x4 a = *(const f32x4*)(t + tile_off<ROWB>(row, k0 * 4));
    const f32x4 b = *(const f32x4*)(t + tile_off<ROWB>(row, k0 * 4 + 16));
    const float x8[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    od_frag_pack(f, x8);
}
template <int ROWB>
__device__ __forceinline__ void frag_contig(od_frag<float>& f, const unsigned char* t, int row, int k0) { frag_contig_f32<ROWB>(f, t, row, k0); }
template <int ROWB>
__device__ __forceinline__ void frag_contig(od_frag<f32x3_t>& f, const unsigned char* t, int row, int k0) { frag_contig_f32<ROWB>(f, t, row, k0); }
// permuted slab u: elements {32u+4g .. +3} and {32u+16+4g .. +3} of `row`
template <int ROWB>
__device__ __forceinline__ void frag_perm(od_frag<bf16_t>& f, const unsigned char* t, int row, int u, int g) {
    const s16x4 a = *(const s16x4*)(t + tile_off<ROWB>(row, (32 * u + 4 * g) * 2));
    const s16x4 b = *(const s16x4*)(t + tile_off<ROWB>(row, (32 * u + 16 + 4 * g) * 2));
    f.v[0] = a[0]; f.v[1] = a[1]; f.v[2] = a[2]; f.v[3] = a[3];
    f.v[4] = b[0]; f.v[5] = b[1]; f.v[6] = b[2]; f.v[7] = b[3];
}
template <int ROWB, class F>
__device__ __forceinline__ void frag_perm_f32(od_frag<F>& f, const unsigned char* t, int row, int u, int g) {
    const f32x4 a = *(const f32x4*)(t + tile_off<ROWB>(row, (32 * u + 4 * g) * 4));
    const f32x4 b = *(const f32x4*)(t + tile_off<ROWB>(row, (32 * u + 16 + 4 * g) * 4));
    const float x8[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    od_frag_pack(f, x8);
}
template <int ROWB>
__device__ __forceinline__ void frag_perm(od_frag<float>& f, const unsigned char* t, int row, int u, int g) { frag_perm_f32<ROWB>(f, t, row, u, g); }
template <int ROWB>
__device__ __forceinline__ void frag_perm(od_frag<f32x3_t>& f, const unsigned char* t, int row, int u, int g) { frag_perm_f32<ROWB>(f, t, row, u, g); }

// "column" fragment: element j of lane (x, g) = tile[row 32u + 16*(j>>2) + 4g + (j&3)][col c0 + x].
// bf16: two LDS transpose reads (ds_read_b64_tr_b16) from the ROW-MAJOR tile — no transposed copy
// of the tile exists.  f32: two 16-byte reads from a transposed tile written at staging time.
template <int ROWB, int TROWB>
__device__ __forceinline__ void frag_cols(od_frag<bf16_t>& f, const unsigned char* t_rm, const unsigned char*, int c0, int x, int u, int g) {
    const int cb = (c0 + 4 * (x & 3)) * 2, rr = 32 * u + 4 * g + (x >> 2);
    const s16x4 a = od_lds_tr_read((const bf16_t*)(t_rm + tile_off<ROWB>(rr, cb)));
    const s16x4 b = od_lds_tr_read((const bf16_t*)(t_rm + tile_off<ROWB>(rr + 16, cb)));
    f.v[0] = a[0]; f.v[1] = a[1]; f.v[2] = a[2]; f.v[3] = a[3];
    f.v[4] = b[0]; f.v[5] = b[1]; f.v[6] = b[2]; f.v[7] = b[3];
}
template <int ROWB, int TROWB>
__device__ __forceinline__ void frag_cols(od_frag<f16_t>& f, const unsigned char* t_rm, const unsigned char*, int c0, int x, int u, int g) {
    od_frag<bf16_t> t;                                   // a 16-bit transpose moves bits: the element type does not matter
    frag_cols<ROWB, TROWB>(t, t_rm, t_rm, c0, x, u, g);
    f.v = t.v;
}
template <int ROWB, int TROWB>
__device__ __forceinline__ void frag_cols(od_frag<float>& f, const unsigned char*, const unsigned char* t_tr, int c0, int x, int u, int g) {
    frag_perm<TROWB>(f, t_tr, c0 + x, u, g);
}
template <int ROWB, int TROWB>
__device__ __forceinline__ void frag_cols(od_frag<f32x3_t>& f, const unsigned char*, const unsigned char* t_tr, int c0, int x, int u, int g) {
    frag_perm<TROWB>(f, t_tr, c0 + x, u, g);
}

