/* osu_dreamer_hip.h — C ABI of libosudreamer_hip.so (MI355X / gfx950).
 *
 * The reference (jaswon/osu-dreamer) has no FFI of its own: its denoiser hot path is
 * torch op call sites inside osu_dreamer/models/diffusion/{backbone,model,train}.py and
 * osu_dreamer/common/{attn,swiglu,rms_norm}.py.  Each entry point below replaces one
 * (group of) those call sites; the citation after "replaces:" is the reference
 * file:line.  The Python host (osu_dreamer_amd/) binds these with ctypes; see
 * INTEGRATION.md for the stub a reference maintainer would add.
 *
 * Conventions
 *  - plain pointers (device memory unless stated), ints, floats; the last argument is
 *    a hipStream_t passed as void*.  Nothing allocates, nothing synchronises.
 *  - return 0 on success; OD_ERR_* (negative) for argument errors;
 *    -(hipError_t) - 1000 when the launch itself failed.
 *  - dtype: OD_F32 or OD_BF16 = element type of activations / packed weights.
 *    Reductions and accumulators are always fp32; parameter gradients are fp32.
 *  - "rows" = frames: activations are frame-major [M = B*L][C] with leading dimension
 *    ld (elements).  Boundary tensors of the reference keep (B, C, L) fp32.
 */
#ifndef OSU_DREAMER_HIP_H
#define OSU_DREAMER_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

enum { OD_F32 = 0, OD_BF16 = 1,
       /* fp32 tensors, products as 3 bf16 MFMAs (hi*hi + hi*lo + lo*hi, fp32 accumulate): ~4e-6 relative
        * error per GEMM instead of 4e-7, 1.7x the fp32 rate.  Accepted by od_gemm_nt and od_flash_attn_fwd
        * only (the no-grad forward / sampler); every other entry point takes OD_F32 for such tensors. */
       OD_F32X3 = 2,
       /* OD_F32X3 with the WEIGHT operand pre-split: W was written by od_pack_weight(OD_F32X3W, ...) as, per row and per 32-element K slab,
        * 32 bf16 high halves followed by 32 bf16 low halves (same bytes and leading dimension as the fp32 matrix; K % 32 == 0).  The weights
        * of a sampler call are constant over its 51 evaluations: splitting them once takes half of the split work out of every GEMM.
        * Accepted by od_gemm_nt / od_gemm_nt_qkrope and od_pack_weight. */
       OD_F32X3W = 3,
       /* "attention in fp16" (BASELINE configs[4]; the reference's trainer option precision: 16-mixed, model.yml:12): the operands of the attention
        * MFMAs are IEEE half — q, k, v in memory, P / dS and the staged dO inside the kernels (v_mfma_f32_*_f16, fp32 accumulation) — while
        * everything around the attention core stays bf16: o, dO, dq, dk, dv are bf16 tensors.  Accepted by od_flash_attn_fwd,
        * od_flash_attn_bwd_fused(+_ws_bytes), od_qk_norm_rope and, as `qk_dtype`, by od_gemm_nt_qkrope_split (head_dim 64 only). */
       OD_F16 = 4 };
enum { OD_EPI_NONE = 0, OD_EPI_SILU = 1 };
enum { OD_ACT_NONE = 0, OD_ACT_SILU = 1 };
enum { OD_ERR_ARG = -1, OD_ERR_ALIGN = -2, OD_ERR_UNSUPPORTED = -3, OD_ERR_COMM = -4 };

int od_version(void);
/* 16 hex digits: hash of the kernel sources (every .hip and .h under csrc, this header, extra compiler flags) the library was built from.
 * Measurement records under profiles/ quote it; bench.py refuses a record taken on another build. */
const char* od_build_source_sha(void);
const char* od_error_string(int code);

/* ---- GEMMs: every nn.Conv1d(k=1) / nn.Linear on the path ------------------------ */
/* C[M,N] = epi(A[M,K] W[N,K]^T + bias[N]) (+= old C if accumulate).
 * replaces: common/attn.py:68-69,75,84; common/swiglu.py:21,25,28,32; backbone.py:63,78;
 *           model.py:45 (proj_audio), and their autograd backward-data. */
int od_gemm_nt(int dtype, const void* A, int lda, const void* W, int ldw, const float* bias, void* C, int ldc,
               int M, int N, int K, int epilogue, int accumulate, void* stream);
/* the qkv projection with the q/k RMSNorm + RoPE applied in its epilogue (no-grad forward / sampler: no separate
 * od_qk_norm_rope pass): C[:, :2*H*hd] = rope(rms_norm(A W^T + bias) * w), C[:, 2*H*hd:] = A W^T + bias.
 * hd in {32, 64}; the pre-norm values are rounded to dtype first, so the result equals od_gemm_nt + od_qk_norm_rope.
 * replaces: attn.py:74-80. */
int od_gemm_nt_qkrope(int dtype, const void* A, int lda, const void* W, int ldw, const float* bias, void* C, int ldc, int M,
                      int N, int K, const float* wq, const float* wk, const float* table, int L, int H, int hd, float eps,
                      float q_scale, void* stream);
/* the training-time form of the above: C[M,N] keeps the pre-norm projection (the backward of the norm needs it) and the normed +
 * rotated q, k go to qk_out[M, 2*H*hd].  One launch at training sizes (bf16, hd 64, M >= 32768: the norm + RoPE run in the large-M
 * GEMM's epilogue); otherwise od_gemm_nt followed by od_qk_norm_rope.  replaces: attn.py:74-80.
 * qk_dtype: the element type of qk_out — dtype itself, or OD_F16 ("attention in fp16", bf16 GEMM, hd 64): qk_out AND the v columns of C
 * (from 2*H*hd on) then hold IEEE half, rounded once from the fp32 accumulators; C's q / k columns stay bf16 (the norm's backward reads them). */
int od_gemm_nt_qkrope_split(int dtype, const void* A, int lda, const void* W, int ldw, const float* bias, void* C, int ldc,
                            void* qk_out, int ldqk, int qk_dtype, int M, int N, int K, const float* wq, const float* wk, const float* table,
                            int L, int H, int hd, float eps, float q_scale, void* stream);
/* dW[N,K] (fp32, ld lddw) += G[M,N]^T A[M,K]; if dbias != NULL also dbias[N] += column sums of G
 * — autograd weight and bias gradients of the above, G read once. */
int od_gemm_tn(int dtype, const void* G, int ldg, const void* A, int lda, float* dW, int lddw, float* dbias, int M, int N,
               int K, void* stream);
/* the same for a G whose N columns come in blocks of n_block columns with only the first n_valid of each live (the packed SwiGLU projection:
 * Hf = 1365 padded to 1408, v then g): output row of column n = (n / n_block) * n_valid + n % n_block, padding columns dropped — one launch
 * over the padded width writes the un-padded (2 Hf, K) gradient (n_block = 0: od_gemm_tn).  replaces: autograd of swiglu.py:21. */
int od_gemm_tn_blocks(int dtype, const void* G, int ldg, const void* A, int lda, float* dW, int lddw, float* dbias, int M, int N,
                      int K, int n_block, int n_valid, void* stream);
/* out[N] (fp32) += column sums of G[M,N]        — autograd bias gradient. */
int od_colsum(int dtype, const void* G, int ldg, float* out, int M, int N, void* stream);
/* dst[Np,Kp] (dtype, zero padded) = src[N,K] fp32 (transpose=0) or dst[Kp,Np] = src^T (transpose=1);
 * row_map: optional int[Np] giving the source row of each destination row (-1 = zero row). */
int od_pack_weight(int dtype, const float* src, int N, int K, void* dst, int Np, int Kp, int transpose,
                   const int* row_map, void* stream);

/* ---- small fp32 linears on per-sample vectors (B rows) ---------------------------- */
/* out[B,N] = act(x[B,K] W[N,K]^T + b).  replaces: model.py:46 (proj_style), backbone.py:76,82
 * (ssg1/ssg2), model.py:100 (u_mod). */
int od_linear_small(const float* x, const float* W, const float* b, float* out, float* pre, int B, int N, int K, int act,
                    void* stream);
/* given dout[B,N] (gradient wrt the post-activation output) and the forward's pre-activation
 * `pre` (may be NULL when act == OD_ACT_NONE): dW[N,K] += , db[N] += , dx[B,K] (+)= .
 * Any of dW/db/dx may be NULL.  dpre[B,N] is fp32 workspace. */
int od_linear_small_bwd(const float* x, const float* W, const float* pre, const float* dout, float* dpre, float* dW,
                        float* db, float* dx, int accumulate_dx, int B, int N, int K, int act, void* stream);

/* ---- layout / boundary kernels ------------------------------------------------------ */
/* (B,C,L) fp32 channel-major -> [B*L][C] frame-major of dtype.  replaces the implicit
 * layout of audio at model.py:120. */
int od_cl_to_frames(int dtype, const float* src, void* dst, int ldd, int B, int C, int L, void* stream);
/* x[m][c] = sum_e W[c][e] xt[b][e][l] + bias[c].  replaces: model.py:95 (proj_in). */
int od_proj_in(int dtype, const float* xt, const float* W, const float* bias, void* x, int ldx, int B, int E, int L,
               int D, void* stream);
/* dW[D,E] += , db[D] += from dx[M][D] and xt. */
int od_proj_in_bwd(int dtype, const float* xt, const void* dx, int ldx, float* dW, float* db, int B, int E, int L, int D,
                   void* stream);
/* y = silu(x) elementwise over [M][C]; dx = dy * silu'(x). */
int od_silu(int dtype, const void* x, void* y, long n, void* stream);
int od_silu_bwd(int dtype, const void* x, const void* dy, void* dx, long n, void* stream);

/* ---- channel RMS norm + adaLN modulation (rms_norm at common/rms_norm.py:7-16) ------- */
/* h = rms_norm(x)*(1+scale[b]) + shift[b] (+ cl[row or row%L]);  inv_rms[m] saved.
 * ssg: fp32 [B][3C] = scale|shift|gate.  cl may be NULL; cl_bcast=1 when cl has batch 1.
 * replaces: backbone.py:76-78,82-83. */
int od_rmsnorm_film(int dtype, const void* x, int ldx, const float* ssg, const void* cl, int ldcl, int cl_bcast,
                    void* h, int ldh, float* inv_rms, int B, int L, int C, float eps, void* stream);
/* dres[m] += rms_norm_bwd(dh*(1+scale)); dssg[b][0:C] += sum dh*xhat; dssg[b][C:2C] += sum dh. */
int od_rmsnorm_film_bwd(int dtype, const void* x, int ldx, const float* inv_rms, const float* ssg, const void* dh,
                        int lddh, void* dres, int lddres, float* dssg, int B, int L, int C, void* stream);
/* xo = x + rms_norm(h)*gate[b];  inv_rms[m] saved.  replaces: backbone.py:79-80,85-86. */
int od_rmsnorm_gate_residual(int dtype, const void* x, int ldx, const void* h, int ldh, const float* ssg, void* xo,
                             int ldxo, float* inv_rms, int B, int L, int C, float eps, void* stream);
/* the pair above fused for the forward pass: xo = x + rms_norm(h) * gate_a, then
 * h2 = rms_norm(xo) * (1 + scale_b) + shift_b (+ cl) — one pass over the frame instead of two
 * (identical results: xo is rounded to dtype before the second norm).  replaces: backbone.py:78-86. */
int od_rmsnorm_gate_residual_film(int dtype, const void* x, int ldx, const void* h, int ldh, const float* ssg_a, void* xo, int ldxo,
                                  float* inv_a, const float* ssg_b, const void* cl, int ldcl, int cl_bcast, void* h2, int ldh2,
                                  float* inv_b, int B, int L, int C, float eps, void* stream);
/* the same AND the depthwise Conv1d that opens the SwiGLU branch, in one pass (round 6): y = dwconv_k(h2) + conv_b with h2 as above (no cl),
 * h2 rounded to dtype before the taps — bit-identical to od_rmsnorm_gate_residual_film followed by od_dwconv.  h2 may be NULL (not kept);
 * xo must not alias x or h (a wave recomputes the ksize/2 frames either side of its run from them).  C <= 512, ksize 3 / 5 / 7 / 9.
 * replaces: backbone.py:78-86 + common/swiglu.py:20 (proj_vg's depthwise conv). */
int od_rmsnorm_gate_residual_film_dwconv(int dtype, const void* x, int ldx, const void* h, int ldh, const float* ssg_a, void* xo, int ldxo,
                                         float* inv_a, const float* ssg_b, void* h2, int ldh2, float* inv_b, const float* conv_w,
                                         const float* conv_b, void* y, int ldy, int B, int L, int C, int ksize, float eps, void* stream);
/* dh = rms_norm_bwd(dy*gate); dssg[b][2C:3C] += sum dy*hhat.  (dy itself is the residual gradient.) */
int od_rmsnorm_gate_residual_bwd(int dtype, const void* h, int ldh, const float* inv_rms, const float* ssg,
                                 const void* dy, int lddy, void* dh, int lddh, float* dssg, int B, int L, int C,
                                 void* stream);

/* ---- attention (common/attn.py:62-84) ------------------------------------------------ */
/* table[l][j] = (cos, sin)(l * 10000^(-2j/hd)), fp32 [L][hd/2][2].  replaces: attn.py:18-24. */
int od_rope_table(float* table, int L, int hd, void* stream);
/* qk_out[m][0:2*dh] = rope(RMSNorm_hd(qkv[m][0:2*dh]) * w) per head, the q heads multiplied by q_scale (1 = the
 * reference's tensor; scale*log2(e) feeds od_flash_attn_* with q_prescaled = 1 and removes a multiply per score from
 * their loops).  replaces: attn.py:77-81. */
int od_qk_norm_rope(int dtype, const void* qkv, int ldqkv, const float* wq, const float* wk, const float* table,
                    void* qk_out, int ldo, int B, int L, int H, int hd, float eps, float q_scale, void* stream);
/* dqkv[m][0:2*dh] from dqk (gradient wrt the roped q * q_scale and the roped k); dwq/dwk[hd] += . */
int od_qk_norm_rope_bwd(int dtype, const void* qkv, int ldqkv, const float* wq, const float* wk, const float* table,
                        const void* dqk, int lddqk, void* dqkv, int lddqkv, float* dwq, float* dwk, int B, int L,
                        int H, int hd, float eps, float q_scale, void* stream);
/* q_prescaled != 0: q holds q * scale * log2(e) (od_qk_norm_rope's q_scale), the softmax is the same function of the
 * unscaled q, and od_flash_attn_bwd returns dq as the gradient of that pre-multiplied tensor.
 * o[m][h*hd+d] = softmax(q k^T * scale) v, non-causal, per (b,h); lse fp32 [B][H][L].
 * replaces: attn.py:82 (F.scaled_dot_product_attention). */
int od_flash_attn_fwd(int dtype, const void* q, int ldq, const void* k, int ldk, const void* v, int ldv, void* o, int ldo,
                      float* lse, int B, int H, int L, int hd, float scale, int q_prescaled, void* stream);
/* dq,dk,dv from do; delta fp32 [B][H][L] is workspace. */
int od_flash_attn_bwd(int dtype, const void* q, int ldq, const void* k, int ldk, const void* v, int ldv, const void* o,
                      int ldo, const void* dout, int lddo, const float* lse, float* delta, void* dq, int lddq, void* dk,
                      int lddk, void* dv, int lddv, int B, int H, int L, int hd, float scale, int q_prescaled, void* stream);
/* Two-stream form of the above.  `aux` (od_attn_aux_create: one non-blocking side stream and two events, owned by the caller; NULL = exactly
 * od_flash_attn_bwd) lets the dQ kernel run beside the dK/dV kernel — both depend on delta alone — so its workgroups fill the CUs the other
 * kernel's last block round leaves idle (-1 % per layer at the bench shape).  Everything is ordered after `stream`'s earlier work and `stream`
 * waits for the side stream before the call's results are used: the caller sees one stream.  Not capturable into a hipGraph while aux != NULL. */
int od_attn_aux_create(void** aux_out);
int od_attn_aux_destroy(void* aux);
int od_flash_attn_bwd_aux(int dtype, const void* q, int ldq, const void* k, int ldk, const void* v, int ldv, const void* o,
                          int ldo, const void* dout, int lddo, const float* lse, float* delta, void* dq, int lddq, void* dk,
                          int lddk, void* dv, int lddv, int B, int H, int L, int hd, float scale, int q_prescaled, void* aux, void* stream);
/* The same gradients from ONE kernel that executes the 5 algorithmic MFMA passes (od_flash_attn_bwd executes 7: its dQ kernel recomputes S and dP).
 * bf16, head_dim 64 only (OD_ERR_UNSUPPORTED otherwise: call od_flash_attn_bwd).  Each workgroup owns 192 keys; its share of every 64-query
 * dQ tile is added to a running fp32 tile that travels key block -> key block through the XCD's L2 in a fixed order (deterministic, no
 * atomics on data), and the last key block writes dq.  `ws` is caller-owned device memory of od_flash_attn_bwd_fused_ws_bytes(...) bytes whose first
 * *zero_out bytes are zero before the first call; one workspace serves any number of calls of ONE (B, H, L) on one stream (zero it again
 * before a launch of another shape: the running tiles carry write numbers that continue from launch to launch).
 * od_flash_attn_bwd_fused_status copies the workspace's sticky error word to the host (the copy is enqueued on `stream` — the stream of the
 * launches — and waited for): 0 = every launch processed all of its jobs, 1 = a launch left jobs unprocessed, 2 = a launch with another
 * (B, H, L) than the workspace's earlier launches was refused (nothing was computed), 3 = a chain wait ran out of time (the predecessor's
 * write never came — a workspace that was not zero, or was left mid-launch by an aborted process): that launch TERMINATED and wrote NaN into dq;
 * zero the workspace's head again before reusing it.  The wait's budget is OD_FB_CHAIN_TIMEOUT_MS (environment, default 1000).
 * The same word can be folded into the step on the device: od_sqnorm / od_adamw_ema take its address (workspace + ..._err_offset()).
 * replaces: autograd of attn.py:82. */
int od_flash_attn_bwd_fused_ws_bytes(int dtype, int B, int H, int L, long* total_out, long* zero_out);
int od_flash_attn_bwd_fused(int dtype, const void* q, int ldq, const void* k, int ldk, const void* v, int ldv, const void* o, int ldo,
                            const void* dout, int lddo, const float* lse, void* dq, int lddq, void* dk, int lddk, void* dv, int lddv,
                            int B, int H, int L, int hd, float scale, int q_prescaled, void* ws, long ws_bytes, void* stream);
int od_flash_attn_bwd_fused_status(const void* ws, int* err_out, void* stream);
int od_flash_attn_bwd_fused_err_offset(void);
int od_flash_attn_bwd_fused_passes(void);
/* profiling builds (-DFB_PROF=1) only: 16 cycle counters of the workspace, copied out and cleared; zeros otherwise. */
int od_flash_attn_bwd_fused_prof(void* ws, long* out16);
/* L x L x hd MFMA passes one od_flash_attn_bwd call issues for bf16 (the algorithmic minimum with the score recompute is 5);
 * bench.py reports it next to the algorithmic roofline figure. */
int od_flash_attn_bwd_passes(void);

/* ---- SwiGLU feed-forward (common/swiglu.py:9-32) ------------------------------------- */
/* y[b][l][c] = bias[c] + sum_j w[c][j] x[b][l+j-r][c], zero padded; ksize in {3, 5, 7, 9}.  replaces: swiglu.py:20, model.py:59,62. */
int od_dwconv(int dtype, const void* x, int ldx, const float* w, const float* bias, void* y, int ldy, int B, int L,
              int C, int ksize, void* stream);
int od_dwconv_bwd(int dtype, const void* x, int ldx, const float* w, const void* dy, int lddy, void* dx, int lddx,
                  float* dw, float* db, int B, int L, int C, int ksize, void* stream);
/* x[(b,l)][c] *= scale[b][c] in place (scale fp32 [B][C]): nn.Dropout1d in training mode — the host draws 0 or 1/(1-p) per
 * (sample, channel); the same call is its backward.  replaces: swiglu.py:23,30. */
int od_scale_channels(int dtype, void* x, int ldx, const float* scale, int B, int L, int C, void* stream);
/* hh[m][0:Hp] = rms_norm_Hf(v*silu(g)), v = vg[m][0:Hp], g = vg[m][Hp:2Hp] (columns >= Hf are zero padding).
 * replaces: swiglu.py:28-30. */
int od_swiglu_rmsnorm(int dtype, const void* vg, int ldvg, void* hh, int ldhh, float* inv_rms, int M, int Hf, int Hp,
                      float eps, void* stream);
int od_swiglu_rmsnorm_bwd(int dtype, const void* vg, int ldvg, const float* inv_rms, const void* dhh, int lddhh,
                          void* dvg, int lddvg, int M, int Hf, int Hp, void* stream);

/* ---- heads (models/diffusion/model.py:86-103) ---------------------------------------- */
/* v[b][e][l] = bias[e] + sum_c W[e][c] rms_norm(x[m])[c]  (fp32, channel-major out).
 * replaces: backbone.py:50 + model.py:97. */
int od_final_norm_proj_out(int dtype, const void* x, int ldx, const float* W, const float* bias, float* v,
                           float* inv_rms, int B, int L, int C, int E, float eps, void* stream);
int od_final_norm_proj_out_bwd(int dtype, const void* x, int ldx, const float* inv_rms, const float* W, const float* dv,
                               void* dx, int lddx, float* dW, float* db, int B, int L, int C, int E, void* stream);
/* fsum[b][c] += sum_l act2[b][c][l] where act2 is the u_head stack applied to xt.  replaces: model.py:58-65,99. */
int od_uhead_fwd(const float* xt, const float* w0, const float* b0, const float* w1, const float* b1, const float* w3,
                 const float* b3, const float* w4, const float* b4, float* fsum, int B, int E, int L, int U,
                 void* stream);
/* parameter gradients of the stack given dfm[b][c] = dLoss/d(mean_l act2[b][c]). */
int od_uhead_bwd(const float* xt, const float* w0, const float* b0, const float* w1, const float* b1, const float* w3,
                 const float* b3, const float* w4, const float* b4, const float* dfm, float* dw0, float* db0, float* dw1,
                 float* db1, float* dw3, float* db3, float* dw4, float* db4, int B, int E, int L, int U, void* stream);
/* u[b] = u_scale*softplus(w_out . (f*(1+mod[0:U]) + mod[U:2U]) + b_out), f = fsum/L.  replaces: model.py:100-102. */
int od_uhead_tail(const float* fsum, const float* mod, const float* w_out, const float* b_out, float* u, int B, int U,
                  int L, float u_scale, void* stream);
/* backward of the tail: dfm[b][c], dmod[b][2U], dw_out[U] +=, db_out[1] += from du[b]. */
int od_uhead_tail_bwd(const float* fsum, const float* mod, const float* w_out, const float* b_out, const float* du,
                      float* dfm, float* dmod, float* dw_out, float* db_out, int B, int U, int L, float u_scale,
                      void* stream);

/* ---- diffusion loss + sampler (models/diffusion/train.py:69-108, model.py:117-138) -- */
/* xt = lerp(x0,x1,t[b]); dsq[b] = frame_dist_sq(xt,x1).  (dsq must be zeroed by the caller.) */
int od_make_xt(const float* x0, const float* x1, const float* t, float* xt, float* dsq, int B, int E, int L, void* stream);
/* sums[b][0..2] += (S1, S2, dS1/du) ; dv = dLoss/dv_pred.  Needs dsq complete. */
int od_loss_grad(const float* xt, const float* x1, const float* u, const float* v, const float* dsq, float* dv,
                 float* sums, int B, int E, int L, float c0, float osl_w, float del_w, void* stream);
/* out[0..3] = loss, osl, del, u_mape; du[b] = dLoss/du_pred. */
int od_loss_finalize(const float* sums, const float* dsq, const float* u, float* out, float* du, int B, float c0,
                     float osl_w, float del_w, void* stream);
/* x -= eta[0] * u[b] * v  (eta is a device scalar so the step is graph-capturable).  replaces: model.py:136. */
int od_sampler_step(float* x, const float* u, const float* v, const float* eta, int B, int E, int L, void* stream);
/* eta[0] = 1 - (sqrt(c0)/max(mean(u), sqrt(c0)+1e-6))^(1/num_steps); also eta[1] = mean(u).  replaces: model.py:131-132. */
int od_sampler_eta(const float* u, float* eta, int B, float c0, int num_steps, void* stream);

/* ---- deterministic accumulation (OD_DETERMINISTIC=1 in the Python host; not part of the reference's path) ------------------
 * The step forms its weight / bias / modulation gradients and its loss scalars with fp32 atomics, whose order — and therefore last bits —
 * changes from run to run.  With a table of registered destination ranges the same kernels add 2^40-scaled INTEGERS into a caller-owned
 * 64-bit shadow of each destination instead (integer addition is associative: any order, same bits; |sum| < 2^23, resolution 2^-40), and
 * od_det_flush folds a shadow range into its fp32 destination (dst += shadow * 2^-40; shadow = 0) before the destination's first reader.
 * Values the format cannot hold (NaN, Inf, |v| >= 2^23) take the float atomic, so a non-finite gradient stays visible.
 *   od_det_clear()                      forget every range and switch the mode off
 *   od_det_register(base, count, sh)    add the range base[0..count) with its shadow sh (count x 8 bytes of device memory, ZERO); at most 14
 *   od_det_enable(table_dev, stream)    copy the table into table_dev (od_det_table_bytes() bytes of device memory) and switch the mode on
 *                                       for every later launch of this process; table_dev = NULL switches it off
 *   od_det_flush(dst, count, stream)    dst[0..count) must lie inside one registered range
 * Process-global state (one process drives one GPU). */
int od_det_clear(void);
int od_det_register(const float* base, long count, void* shadow_i64);
int od_det_enable(void* table_dev, void* stream);
int od_det_table_bytes(void);
int od_det_flush(float* dst, long count, void* stream);

/* ---- optimizer (models/diffusion/train.py:110-126; model.yml:39) --------------------- */
/* out[0] += sum g^2.  `status` (may be NULL): device address of an error word of an earlier kernel of the step — the fused attention
 * backward's, workspace + od_flash_attn_bwd_fused_err_offset().  Non-zero there turns out[0] into NaN on the device: every step is checked
 * without a host synchronisation (replaces nothing in the reference; the clip it feeds is model.yml:39). */
int od_sqnorm(const float* g, long n, float* out, const int* status, void* stream);
/* clip by global norm (gnorm_sq device scalar, max_norm<=0 disables) -> AdamW -> EMA, one pass.
 * ema_mode: 0 none, 1 copy (first update), 2 lerp with (1-ema_decay).
 * `status` (may be NULL): as for od_sqnorm; non-zero there SKIPS the update (parameters, moments and the average are left untouched). */
int od_adamw_ema(float* p, const float* g, float* m, float* v, float* ema, long n, float lr, float beta1, float beta2,
                 float eps, float weight_decay, int step, float ema_decay, int ema_mode, const float* gnorm_sq,
                 float max_norm, const int* status, void* stream);

/* EMA only: mode 1 copy, mode 2 ema += (1-decay)*(p-ema).  replaces: train.py:125-126 when the
 * average is updated outside the fused pass. */
int od_ema_update(float* ema, const float* p, long n, float ema_decay, int ema_mode, void* stream);

/* ---- style-model sampler (models/style/model.py:73-119; the rest of it reuses od_linear_small,
 *      od_rmsnorm_film / od_rmsnorm_gate_residual with L = 1, od_uhead_tail, od_sampler_eta, od_sampler_step) */
/* c[B,H] = sum over the NL labels of (label < 0 ? null[n] : cond_b[n] + rff(label/10) . cond_w[n]),
 * rff(x)[f] = sqrt(2/F) cos(x*rff_w[f] + rff_b[f]).  replaces: style/model.py:73-80 + common/fourier_features.py:15. */
int od_style_conditioning(const float* labels, const float* rff_w, const float* rff_b, const float* cond_w,
                          const float* cond_b, const float* null_labels, float* c, int B, int NL, int F, int H,
                          void* stream);
/* y[M,C] = x * rsqrt(mean_c x^2 + eps) (* gamma[C] if non-NULL), fp32.  replaces: nn.RMSNorm at style/model.py:50
 * and rms_norm at :99. */
int od_rmsnorm_rows(const float* x, const float* gamma, float* y, int M, int C, float eps, void* stream);

/* ---- latent model, inference path (models/latent/{spec_features,unet,model}.py; LDM.sample's steps either
 *      side of diffusion.sample, inference/model.py:48,51).  Frame-major [B*L][C] activations; C = h_dim must be a
 *      power of two in 8..512.  The SwiGLU body of each block is od_dwconv + od_gemm_nt + od_swiglu_rmsnorm. ---- */
/* out[(b,l)][c2*3 + a] = SiLU(rms(conv2(SiLU(rms(conv1(audio)) * g1))) * g2): the two strided Conv2d stages of
 * SpecFeatures with their channel RMS norms, audio (B,72,L) fp32 channel-major.  replaces: spec_features.py:17-26. */
int od_spec_features_conv(int dtype, const float* audio, const float* w1, const float* b1, const float* g1, const float* w2,
                          const float* b2, const float* g2, void* out, int ldo, int B, int F, int L, float eps, void* stream);
/* y = act(rms_norm(x) * gamma * (1 + scale[b]) + shift[b]); ssg = [B][3C] (scale, shift, gate) or NULL.
 * replaces: unet.py:50 (norm + FiLM), :53 (out_norm), spec_features.py:27-28. */
int od_rmsnorm_affine_film(int dtype, const void* x, int ldx, const float* gamma, const float* ssg, void* y, int ldy, int B,
                           int L, int C, float eps, int act, void* stream);
/* xo = x + rms_norm(h) * gamma * (1 + gate[b]) (gate = ssg[b][2C:3C], 0 when ssg is NULL).  replaces: unet.py:28,51. */
int od_rmsnorm_affine_gate_residual(int dtype, const void* x, int ldx, const void* h, int ldh, const float* gamma,
                                    const float* ssg, void* xo, int ldxo, int B, int L, int C, float eps, void* stream);
/* xo = x + rms_norm(p) * gamma * gx, p = proj(skip) rows (L rows shared by all b when p_bcast), gx = gate(x) rows.
 * replaces: unet.py:117-126 (mixer). */
int od_unet_mixer(int dtype, const void* x, int ldx, const void* p, int ldp, int p_bcast, const void* gx, int ldg,
                  const float* gamma, void* xo, int ldxo, int B, int L, int C, float eps, void* stream);
/* depthwise Conv1d(k = 1 + 2*(stride/2), zero pad) then AvgPool1d(stride): x [B*Lo*stride][C] -> y [B*Lo][C].
 * replaces: unet.py:58-63. */
int od_unet_down(int dtype, const void* x, int ldx, const float* w, const float* bias, void* y, int ldy, int B, int Lo, int C,
                 int stride, void* stream);
/* nearest Upsample(stride) then the same depthwise Conv1d: x [B*Li][C] -> y [B*Li*stride][C].  replaces: unet.py:80-85. */
int od_unet_up(int dtype, const void* x, int ldx, const float* w, const float* bias, void* y, int ldy, int B, int Li, int C,
               int stride, void* stream);
/* out (B,N,L) fp32 = f_n(bias[n] + W[n] . x[(b,l)]), f_n = sigmoid for n < n_sigmoid else identity (N <= 16); with
 * rms != 0 the N outputs of each frame are RMS-normalised (no gain).
 * replaces: latent/model.py:114 (proj_out) + :127-131 (hit-signal sigmoid); :65-68 (temporal_head). */
int od_chart_head(int dtype, const void* x, int ldx, const float* W, const float* bias, float* out, int B, int L, int C, int N,
                  int n_sigmoid, int rms, float eps, void* stream);
/* out[b][h*hd + d] (fp32) = sum_l softmax_l(scores[(b,l)][h]) * values[(b,l)][h*hd + d]   (hd <= 256).
 * replaces: AttnPool.forward, latent/model.py:33-36 (the two 1x1 convs before it are od_gemm_nt). */
int od_attn_pool(int dtype, const void* scores, int lds, const void* values, int ldv, float* out, int B, int L, int Hh, int hd,
                 void* stream);

/* ---- data-parallel exchange: RCCL over xGMI (no counterpart in the reference, which is
 *      single-device, models/diffusion/model.yml:11 `devices: 1`) ------------------------------------------------- */
/* Bind RCCL at run time: dlopen(path), or "librccl.so.1" when path is NULL/empty.  A host process that already holds an
 * RCCL instance (PyTorch's torch/lib/librccl.so) passes that path so both share one library.  Called implicitly (with the
 * default) by the entry points below if it was not called before. */
int od_comm_load(const char* path);
/* ncclGetVersion() of the bound library (e.g. 22606), 0 if none could be bound. */
int od_comm_version(void);
/* rank 0: fill out[nbytes = 128] with a fresh ncclUniqueId (host memory); the host ships it to the other ranks. */
int od_comm_unique_id(void* out, int nbytes);
/* every rank: *comm_out = communicator of `nranks` ranks built from the shared 128-byte id (ncclCommInitRank on the
 * current device). */
int od_comm_init(void** comm_out, int nranks, int rank, const void* unique_id, int nbytes);
int od_comm_destroy(void* comm);
/* failure path (a peer died): ncclCommAbort — frees the communicator without waiting for collectives that can no longer complete */
int od_comm_abort(void* comm);
/* number of ranks RCCL itself counts in the communicator (ncclCommCount), or a negative error: the bench line's `rccl_ranks_seen` */
int od_comm_count(void* comm);
/* Test aid for boxes with one GPU (a single-rank ncclAllReduce in place launches nothing): the footprint of a ring all-reduce on the
 * kernels beside it — `channels` workgroups of `threads` (256 / 512 / 1024) threads, resident for the whole call, each streaming its slice of
 * buf[count] (read, written back unchanged) `rounds` times.  Not part of the reference's path. */
int od_comm_ring_standin(float* buf, long count, int channels, int threads, int rounds, void* stream);
/* grads[0:count] (fp32, device) <- sum over ranks (average != 0: mean) in place, enqueued on `stream`: the one exchange
 * of a data-parallel training step (the gradient of train.py:120-123's loss over the global batch). */
int od_allreduce_grads(void* comm, float* grads, long count, int average, void* stream);
/* buf[0:count] (fp32, device) <- rank `root`'s copy: parameters, AdamW moments and the EMA at start-up / resume. */
int od_broadcast_f32(void* comm, float* buf, long count, int root, void* stream);

/* ---- hipGraph helpers for the captured sampler loop ---------------------------------- */
int od_graph_begin(void* stream);
int od_graph_end(void* stream, void** graph_exec_out);
int od_graph_launch(void* graph_exec, void* stream);
int od_graph_destroy(void* graph_exec);

/* ---- measurement aid ------------------------------------------------------------------ */
/* One launch of a bare-MFMA kernel (v_mfma_f32_16x16x32_bf16 on pseudo-random operands, one wave per SIMD on every CU, nothing else in the
 * loop): `iters` x 64 MFMAs per wave; *flops_out = the FLOPs of the launch.  bench.py times ~2 s of it beside every measurement: the dense bf16
 * rate this board sustains under its power cap (the pool's boxes differ by +-5 %), against which a step time of that box can be read.
 * scratch: device memory, >= 1 KiB per CU.  Not part of the reference's path. */
int od_mfma_calibrate(float* scratch, int iters, double* flops_out, void* stream);

#ifdef __cplusplus
}
#endif
#endif
