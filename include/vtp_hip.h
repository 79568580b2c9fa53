/* vtp_hip.h -- C ABI of libvtp_hip.so: the MI355X (gfx950 / CDNA4) kernels of the VTP training hot path.
 *
 * The reference (MiniMax-AI/VTP) is pure PyTorch: it has no FFI / plugin registry, every device
 * "kernel" is an ATen call issued from an nn.Module.forward.  Each entry point below
 * therefore names the reference ATen call site(s) (file:line under /root/reference) it replaces.
 *
 * Conventions
 *   - plain pointers + sizes only; all pointers are DEVICE pointers unless stated; bf16 = raw uint16
 *     storage (IEEE bfloat16, round-to-nearest-even); "f32" = float.
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream).  Kernels are enqueued
 *     asynchronously; nothing synchronises the device.
 *   - return 0 on success, VTP_ERR_ARG (-1) on a rejected argument (message via vtp_last_error()),
 *     or a positive hipError_t if the launch failed.  The library never aborts the process.
 *   - no hidden global state apart from the last-error string (thread-local) and kernel attributes;
 *     callers own every buffer (ownership never transfers).
 *   - token matrices are row-major [rows, features] with an explicit leading dimension where given.
 */
#ifndef VTP_HIP_H
#define VTP_HIP_H
#ifdef __cplusplus
extern "C" {
#endif

#define VTP_ABI_VERSION 1

int vtp_abi_version(void);
const char* vtp_last_error(void);

/* ---- GEMM ------------------------------------------------------------------------------------
 * C[M,N] = epilogue(alpha * A[M,K] * B[N,K]^T).  A, B bf16 K-contiguous.  Replaces every nn.Linear /
 * 1x1-conv / patch-embed conv `addmm`: attention.py:92,94 (qkv, proj), ffn.py:78-81 (w1,w2,w3),
 * vision_transformer_bottleneck.py:68-74, pixel_decoder.py:138,157, embeddings.py:64 (after im2col),
 * block.py:412-414 + nn.MultiheadAttention in/out proj (text tower), modeling_vtp.py:274,306, and (with
 * transposed operands) their autograd dgrad/wgrad.
 *   a_grp/a_pre, c_grp/c_pre: optional row remap row(m) = m + (m / grp + 1) * pre (grp = 0: none) used to
 *   read/write only the patch rows of a [B, 1+hw, D] token stream.  c_grp = -1 selects the SwiGLU de-interleave
 *   row(m) = ((m>>4)<<3) + (m&7) + ((m&8) ? c_pre : 0) (wgrad of the interleaved [w1|w2] matrix, c_pre = H).
 * Epilogues:
 *   VTP_EPI_BF16        C bf16 = acc + bias
 *   VTP_EPI_F32         C f32  = resid + gamma * (acc + bias)            (bias/gamma/resid optional; block.py:293-294)
 *   VTP_EPI_SWIGLU      B rows interleaved [8 x w1 | 8 x w2] per 16; C bf16 [M, N/2] = silu(x1) * x2, C2 bf16 [M,N] = (x1|x2)
 *   VTP_EPI_GELU        C bf16 = gelu_erf(acc + bias), C2 bf16 = acc + bias (optional)
 *   VTP_EPI_QUICK_GELU  the same with QuickGELU x * sigmoid(1.702 x) (text_quick_gelu; layers/activation.py:5-12)
 *   VTP_EPI_F32_ATOMIC  C f32 += alpha * acc, split-K over `splits` slices with fp32 atomics
 *   VTP_EPI_F32_SLAB    split-K slice z writes alpha * acc to C + z * (4*ldc2) floats with plain stores (wgrad; sum the
 *                       slabs with vtp_reduce_slabs).  The slice count actually used is vtp_gemm_splits(K, splits).
 */
enum { VTP_EPI_BF16 = 0, VTP_EPI_F32 = 1, VTP_EPI_SWIGLU = 2, VTP_EPI_GELU = 3, VTP_EPI_F32_ATOMIC = 4, VTP_EPI_F32_SLAB = 5,
       VTP_EPI_QUICK_GELU = 8 };
int vtp_gemm_splits(int K, int splits);
/* split-K factor the weight-gradient GEMM C[M,N] = A[K,M]^T B[K,N] should be launched with (tile configuration aware:
 * 256x256 8-phase kernel -> tiles x splits = one round of the CUs; ring kernel -> just under 512 workgroups) */
int vtp_gemm_tn_splits(int M, int N, int K);
/* C[M,N] f32 = A[K,M]^T * B[K,N]: A, B bf16 row-major with the reduction dimension (tokens) as ROWS -- the weight-gradient
 * GEMM dW = dY^T X of every linear, read straight from the activation layouts (fragments are formed with the gfx950 LDS
 * transpose read; no transposed copies).  epilogue VTP_EPI_F32 (C = resid + acc, pass resid = C to accumulate),
 * a_colsum (optional, f32 [M], ACCUMULATED, rows mapped like C's): column sums of A over the tokens = the bias gradient of the
 * same linear layer (nn.Linear backward: db = dY.sum(0)); fused into the GEMM where the tile configuration allows, else a
 * column-sum pass on the same stream.
 * VTP_EPI_F32_SLAB (split-K slabs, see above) or VTP_EPI_F32_ATOMIC (split-K slices add into C with fp32 atomics: no slab
 * round trip through HBM; the summation order, hence the last bits, vary from run to run).  a_/b_ remaps act on the token rows of A / B, c_ on the rows of C. */
int vtp_gemm_tn(const void* A, int lda, const void* B, int ldb, void* C, int ldc, int ldc2, const float* resid, int M, int N,
                int K, int epilogue, int a_grp, int a_pre, int b_grp, int b_pre, int c_grp, int c_pre, int splits,
                float* a_colsum, void* stream);
/* The weight gradients of one transformer block as ONE launch: up to 8 problems C_g[M_g, N_g] (+)= A_g[K, M_g]^T B_g[K, N_g] that
 * reduce over the same K token rows (nn.Linear backward dW = dY^T X of attn.qkv / attn.proj / mlp.w1|w2 / mlp.w3, block.py:290-296),
 * (sum of 256 x 256 tiles) x splits workgroups; the K slices of a tile are combined inside the launch by the last-arriving
 * workgroup, which applies the epilogue (accumulate or overwrite, SwiGLU row de-interleave c_grp = -1, bias-gradient column sums).
 * probs: device array of nprob records of 16 int64 {A, B, C, colsum | lda, ldb, ldc | M, N | c_grp, c_pre | tile0 | accumulate | 0 0 0};
 * part: ntiles * splits_eff * 65536 floats, ticket: ntiles ints, zero before the first launch (the kernel leaves them zero). */
int vtp_gemm_tn_grouped(const void* probs, int nprob, int ntiles, int K, int splits, void* part, void* ticket, void* stream);
/* the same launch with the kernel named: 0 = the 8-phase kernel (= vtp_gemm_tn_grouped), 1 = the one-wave-per-SIMD kernel with the
 * hand-scheduled k loop (needs K % 8 == 0; bit-identical results per K slice) */
int vtp_gemm_tn_grouped_k(const void* probs, int nprob, int ntiles, int K, int splits, void* part, void* ticket, int kernel, void* stream);
/* the same launch (one-wave-per-SIMD kernel, K % 8 == 0) from an explicit work-item list, so that the tiles need not be cut alike: items =
 * device array of nitems records of 8 int32 {tile, kbeg, kcount, nparts, part, 0, 0, 0}, one workgroup each; the items of a tile partition
 * [0, K) (kbeg multiples of 64); slots = the largest nparts; part: ntiles * slots * 65536 floats.  The tiles that also form a bias gradient
 * (column sums of dY beside the MFMAs: 27 % more time per k-tile) get one slice more than the rest -- same reference call sites as above. */
int vtp_gemm_tn_grouped_items(const void* probs, int nprob, int ntiles, int K, const void* items, int nitems, int slots, void* part,
                              void* ticket, void* stream);
/* tuning knob (benchmarks / experiments): force a tile configuration id (-1 = heuristic) and toggle the XCD-aware
 * workgroup remap.  Process-global; not part of the reference-facing surface. */
int vtp_set_gemm_tuning(int force_cfg, int xcd_swizzle);
/* persistent 256 x 256 NT launches draw their tiles from per-XCD queues in device memory (1) instead of owning a static tile list (0,
 * default): a launch that cannot get every CU at once -- RCCL channels hold some for the whole backward -- then ends when the tiles
 * are done, not when the last late-starting workgroup's list is.  Process-global; VTP_GEMM_DYN=0 / 1 in the environment overrides.
 * VTP_GEMM_CUS=n caps the workgroup slots of every one-workgroup-per-CU GEMM launch (INTEGRATION.md "Running beside RCCL"). */
int vtp_set_gemm_dynamic(int on);
/* diagnostics (tools/gemm8p_timeline.py): `timing` = device buffer of [workgroups][16 tiles][4] 64-bit s_memrealtime stamps
 * {tile start, k loop done, epilogue issued} written by the 256x256 kernel (null = off); grid_limit caps its persistent grid
 * (0 = every CU); delay_ticks > 0 starts every second workgroup of an XCD that many 10-ns ticks late (lock-step experiments).
 * Process-global; not part of the reference-facing surface. */
int vtp_gemm_debug(void* timing, int grid_limit, int delay_ticks);
/* diagnostics (tools/attn_bwd_timeline.py): `timing` = device buffer of 2 x [workgroups][16 waves][4] 64-bit s_memrealtime stamps
 * {start, operands staged, loop done, gradients stored} written by the resident attention backward kernels (dQ kernel, then the
 * dK/dV kernel), null = off.  Process-global; not part of the reference-facing surface. */
/* diagnostics (tools/cu_thief.py; VTP_DIAG=1): nwg workgroups that hold one CU each (160 KiB LDS) for ticks x 10 ns on `stream` -- the
 * footprint of a communication kernel beside the step; out: null or nwg u64 (ticks held) */
int vtp_cu_thief(int nwg, int ticks, void* out, void* stream);
int vtp_attn_debug(void* timing, int lds_pad, int waves_per_wg, int stagger_ticks);  /* lds_pad: extra dynamic LDS bytes per workgroup; waves_per_wg: 0 = heuristic */
/* host-only: the kernel configuration vtp_gemm_nt picks for a shape (8 = 256x256 8-phase, 7 = 128x64 ring tiles, other ids = ring
 * configurations; bits 8.. = in-launch split-K slices when > 1); -1 for a bad shape.  No launch, no device needed. */
int vtp_gemm_nt_config(int M, int N, int K, int epilogue);
int vtp_gemm_nt(const void* A, int lda, const void* B, int ldb, void* C, int ldc, void* C2, int ldc2,
                const float* bias, const float* gamma, const float* resid, int M, int N, int K, int epilogue,
                int a_grp, int a_pre, int c_grp, int c_pre, int splits, float alpha, void* stream);

/* qkv projection with apply_rope fused into the epilogue: replaces attention.py:115 (qkv Linear) + attention.py:70-89
 * (cast to the rope dtype, rotate the patch rows of q and k, cast back; bit-identical to vtp_gemm_nt + vtp_rope_qk).
 * C bf16 [M, N = 3*D] = A[M,K] W[N,K]^T + bias; columns < rope_cols (= 2*D) of row m are rotated with row rope_pos[m] of the
 * bf16 tables sin / cos [*, 64] (rope_pos[m] < 0: prefix (cls) rows, untouched).  rope_pos lets several images / resolutions
 * share one launch (the list forward, vision_transformer.py:221-258): it indexes a concatenation of the per-resolution tables. */
int vtp_gemm_qkv_rope(const void* A, int lda, const void* W, int ldb, const float* bias, void* C, int ldc, int M, int N, int K,
                      const int* rope_pos, const void* rope_sin, const void* rope_cos, int rope_cols, void* stream);

/* ---- residual-wiring extras of SelfAttentionBlock (block.py:20-118,207-289; misc.py:7-26) ---------------------------------
 * Stochastic depth ("sample drop"): the residual branch runs on a random subset of the images of a segment (an image = N
 * consecutive token rows) and is added back with alpha = batch / kept (torch.index_add(x, 0, residual, idx, alpha), block.py:216-232).
 *   vtp_gather_image_rows : dst[i*N + t, :] = src[idx[i]*N + t, :] as f32 (dst, optional) and/or as bf16(scale * .) (dst_bf16, optional)
 *   vtp_scatter_image_rows: dst[idx[i]*N + t, :] = (accumulate ? dst : 0) + alpha * src[i*N + t, :]      (idx entries are distinct)
 * LayerScale (y = x + gamma * f(x), misc.py:24-25) backward without storing f: with G = dy^T x_in (the unscaled weight-gradient
 * GEMM) and cs = colsum(dy):  dW += gamma (.) G,  db += gamma (.) cs,  dgamma += rowsum(W (.) G) + b (.) cs;
 * vtp_scaled_transpose writes bf16 (gamma (.) W)^T, the dgrad operand. */
int vtp_gather_image_rows(const float* src, const int* img_idx, float* dst, void* dst_bf16, int n_img, long N, int D, float scale,
                          void* stream);
int vtp_scatter_image_rows(const float* src, const int* img_idx, float* dst, int n_img, long N, int D, float alpha, int accumulate,
                           void* stream);
int vtp_layerscale_wgrad(const float* G, const float* W, const float* bias, const float* colsum, const float* gamma, float* dW,
                         float* db, float* dgamma, int N, int K, void* stream);
int vtp_scaled_transpose(const float* W, const float* gamma, void* dstT, int N, int K, void* stream);
/* QK normalisation (attention.py:67-68,119-120): q = RMSNorm(64)(q), k likewise (weights wq / wk [64], shared by the heads), before
 * RoPE.  qkv bf16 [M, 3D] -> out (v copied), inv f32 [M, 2 D / 64] = rsqrt(mean x^2 + eps) per (row, part, head);
 * backward in place on the q / k parts of dqkv, dwq / dwk accumulated. */
int vtp_qk_norm_fwd(const void* qkv, const float* wq, const float* wk, void* out, float* inv, long M, int D, float eps, void* stream);
int vtp_qk_norm_bwd(void* dqkv, const void* qkv, const float* inv, const float* wq, const float* wk, float* dwq, float* dwk, long M, int D,
                    void* stream);

/* ---- normalisation ---------------------------------------------------------------------------
 * kind 0 = RMSNorm (normalization.py:17-22, eps 1e-5, no bias), 1 = LayerNorm (vision_transformer.py:30-34
 * eps 1e-6 decoder; normalization.py:25-31 text).  x f32 [M, D] -> y bf16 [M, D]; stats f32 [M,2] = (mean, rstd).
 * bwd: dx f32 [M,D] = (dres ? dres : 0) + norm_bwd(dy bf16), optionally also written as bf16 (dx_bf16, the A operand
 * of the next dgrad GEMM); dw/db f32 [D] are ACCUMULATED (+=) atomically.  dx_colsum (optional, f32 [D], accumulated):
 * column sums of dx_bf16 = the bias gradient of the linear layer that receives dx_bf16 as its dy (no separate pass). */
int vtp_norm_fwd(const float* x, const float* w, const float* b, void* y, float* stats, int M, int D, float eps,
                 int kind, void* stream);
int vtp_norm_bwd(const void* dy, const float* x, const float* w, const float* stats, const float* dres, float* dx,
                 void* dx_bf16, float* dw, float* db, float* dx_colsum, int M, int D, int kind, void* stream);

/* ---- RoPE (attention.py:12-23,70-89) ----------------------------------------------------------
 * In place on the q and k thirds of a packed qkv bf16 [B*N, 3*D] buffer (head h at column h*64 of each third).
 * Rows n < prefix (cls) are untouched; sin/cos are the reference's bf16 tables [N-prefix, 64].  Rounding follows
 * eager bf16: bf16(bf16(x*cos) + bf16(rot(x)*sin)).  inverse != 0 applies the transpose (backward). */
int vtp_rope_qk(void* qkv, const void* sin, const void* cos, int B, int N, int heads, int prefix, int inverse,
                void* stream);

/* ---- attention (attention.py:124 F.scaled_dot_product_attention; text: nn.MultiheadAttention causal) ----
 * q/k/v/o are bf16 with element strides: batch stride `sb`, token stride `sn`, head h at +h*64 (head_dim is 64).
 * lse f32 [B, heads, N] (natural-log-sum-exp of scale*q.k).  causal != 0 masks key > query.
 * bwd: delta f32 [B, heads, N] is scratch (rowsum(dO*O)).  rope_sin / rope_cos (bf16 [N - rope_prefix, 64], both or neither):
 * dq and dk are returned as gradients w.r.t. the UN-rotated q, k (the inverse of vtp_rope_qk applied with the same eager-bf16
 * rounding) -- fused into the kernels' stores for short sequences, a follow-up pass otherwise (needs the packed qkv layout). */
int vtp_attn_fwd(const void* q, const void* k, const void* v, void* o, float* lse, int B, int N, int heads,
                 long sb_qkv, long sn_qkv, long sb_o, long sn_o, float scale, int causal, void* stream);
int vtp_attn_bwd(const void* q, const void* k, const void* v, const void* o, const void* d_o, const float* lse,
                 float* delta, void* dq, void* dk, void* dv, const void* rope_sin, const void* rope_cos, int rope_prefix,
                 int B, int N, int heads, long sb_qkv, long sn_qkv, long sb_o, long sn_o, float scale, int causal,
                 void* stream);

/* ---- data movement / elementwise --------------------------------------------------------------- */
/* im2col for the 16x16/s16 patch-embed conv (embeddings.py:58,64-69): img f32 [B,3,H,W] -> patches bf16 [B*h*w, 768],
 * K order (c, ky, kx), token order (y, x). */
int vtp_im2col16(const float* img, void* patches, int B, int H, int W, void* stream);
/* the same into a row layout with `prefix` rows in front of every image's patches (prefix = 1: the trunk's token rows, row 0 = cls, cf.
 * prepare_tokens_with_masks, vision_transformer.py:189-219): patch p of image b -> row b*(h*w+prefix) + prefix + p of rows bf16
 * [B*(h*w+prefix), 768]; the prefix rows are not written */
int vtp_im2col16_rows(const float* img, void* rows, int B, int H, int W, int prefix, void* stream);
/* gradient w.r.t. the input image of PatchEmbed (embeddings.py:61-70 backward; the reference's autograd returns it when the image
 * requires grad): d_patches f32 [B*hw, 768] (= d_tokens[patch rows] W_pe, K order (c, ky, kx)) folded back to d_img f32 [B,3,H,W] */
int vtp_col2im16(const float* dpatches, float* dimg, int B, int H, int W, void* stream);

/* rows [B, N, D] f32: write row 0 of every batch element = cls[D] (vision_transformer.py:198,210-217);
 * optionally substitute mask_token on masked patch rows (vision_transformer.py:195). masks: uint8 [B, N-1] or NULL. */
int vtp_assemble_tokens(float* x, const float* cls, const float* mask_token, const unsigned char* masks, int B, int N,
                        int D, void* stream);

/* out[c, r] = in[r, c] for a bf16 [R, C] matrix (ld_in) -> [C, ld_out]; columns r in [R, ld_out) are zero-filled.
 * If colsum != NULL, colsum[c'] += sum_r in[r,c] (bias gradients); c' = c, or, when colsum_swiglu_h = H > 0, the
 * de-interleaved SwiGLU index ((c>>4)<<3) + (c&7) + ((c&8) ? H : 0)  (b1 then b2, each [H]).
 * in_grp/in_pre: input row remap row(r) = r + (r / in_grp + 1) * in_pre (in_grp = 0: none), as in vtp_gemm_nt. */
int vtp_transpose_bf16(const void* in, int ld_in, void* out, int ld_out, float* colsum, int colsum_swiglu_h, int in_grp,
                       int in_pre, int R, int C, void* stream);
/* out[c'] += sum_r in[r, c] for a bf16 [R, C] matrix (bias gradient); same column / row remaps as vtp_transpose_bf16. */
int vtp_colsum_bf16(const void* in, int ld, float* out, int colsum_swiglu_h, int in_grp, int in_pre, int R, int C, void* stream);
/* the same with the row count in DEVICE memory (rows >= n_rows_dev[0] are skipped; R_max sizes the launch): the masked-token
 * count of an iBOT batch changes every step while the padded token buffers (vtp.py:432-439 `upperbound`) and a captured
 * hipGraph do not */
int vtp_colsum_bf16_rows(const void* in, int ld, float* out, const int* n_rows_dev, int R_max, int C, void* stream);
/* backward of vtp_assemble_tokens' mask substitution: for masked patch rows d_mask_token += dx[row] and dx_bf16[row] = 0
 * (so the patch-embed wgrad / bias-grad skip them).  dx f32 / dx_bf16 [B*N, D], masks uint8 [B, N-1]. */
int vtp_mask_rows_bwd(const float* dx, void* dx_bf16, const unsigned char* masks, float* d_mask_token, int B, int N, int D,
                      void* stream);
/* out[d] += sum_b in[b*stride + d], f32 (gradient of the broadcast cls token, vision_transformer.py:210-217). */
/* backward of vtp_assemble_tokens as a whole (vision_transformer.py:189-219 backward): the mask substitution as above (masks and
 * d_mask_token may both be null: an item without masks) AND the cls row: d_cls_token += dx[row 0 of every image], dx_bf16[row 0] = 0 --
 * afterwards dx_bf16 is the gradient of the patch-embed output in the token-row layout (zero in every row that is not a visible patch) */
int vtp_token_rows_bwd(const float* dx, void* dx_bf16, const unsigned char* masks, float* d_mask_token, float* d_cls_token, int B, int N,
                       int D, void* stream);
int vtp_strided_rowsum(const float* in, long stride, float* out, int B, int D, void* stream);

/* f32 -> bf16 cast (n elements); f32 [R,C] -> bf16 transposed [C,R] (weight caches W, W^T). */
int vtp_cast_f32_bf16(const float* in, void* out, long n, void* stream);
int vtp_cast_transpose_f32_bf16(const float* in, void* out, int R, int C, void* stream);
/* Batched refresh of the bf16 compute copies of all fp32 master weights in ONE launch.  `descs` is a device array of
 * n records of 8 x int64: {src f32*, src2 f32*, dst, dstT, R, C, mode, tile_start}; mode 0: dst bf16 [R,C] = src,
 * dstT bf16 [C,R] = src^T (either may be NULL); mode 1: the logical matrix is the SwiGLU interleave of src=w1 and
 * src2=w2 (16-row groups = [8 rows w1 | 8 rows w2], R = 2H); mode 2: dst f32 [R] = interleave of two bias vectors.
 * tile_start = exclusive prefix sum of per-record tile counts (64x64 tiles; mode 2: 256 elements per tile). */
int vtp_prep_weights(const void* descs, int n, int total_tiles, void* stream);
/* the same over a RUN of the table (the layers of one gradient bucket, refreshed right behind that bucket's optimizer update while the
 * backward of the other layers is still running): descs = the run's first record, n records, tile_base = that record's tile_start,
 * n_tiles = tiles of the run. */
int vtp_prep_weights_range(const void* descs, int n, int tile_base, int n_tiles, void* stream);

/* SwiGLU backward (ffn.py:80): given dh bf16 [M,H] and saved x12 bf16 [M,2H] (interleaved 8|8), writes dx12 bf16 [M,2H].
 * db12 (optional, f32 [2H] = [b1 | b2], accumulated): column sums of dx12 = the bias gradients of w1 / w2. */
int vtp_swiglu_bwd(const void* dh, const void* x12, void* dx12, float* db12, int M, int H, void* stream);
/* the same backward fused into the w3 dgrad GEMM: dx12[M, 2H] from dy[M, K] W3^T[H, K]^T and the saved x12 -- dh never reaches HBM */
int vtp_gemm_dgrad_swiglu(const void* A, int lda, const void* WT, int ldb, const void* x12, int ldx, void* dx12, int ldc, int M, int H,
                          int K, void* stream);
/* GELU backward (text MLP, block.py:399): dx = dy * gelu'(pre). */
int vtp_gelu_bwd(const void* dy, const void* pre, void* dx, long n, void* stream);
/* dx = dy * QuickGELU'(pre)  (x sigmoid(1.702 x), layers/activation.py:5-12); bf16, n % 8 == 0 */
int vtp_quick_gelu_bwd(const void* dy, const void* pre, void* dx, long n, void* stream);

/* PixelShuffle(16) (pixel_decoder.py:160): t bf16 [B*h*w, 768] token-major -> img f32 [B,3,16h,16w]. */
int vtp_pixel_shuffle16(const void* t, float* img, int B, int h, int w, void* stream);
/* backward of F.pixel_shuffle(., 16) (pixel_decoder.py:160): image gradient f32 [B,3,16h,16w] -> token-major bf16 [B*h*w, 768] */
int vtp_pixel_unshuffle16(const float* d_img, void* dt, int B, int h, int w, void* stream);
/* L1 reconstruction loss on the token-major decoder output: loss_sum[0] += sum |shuffle(t) - target|;
 * dt bf16 [B*h*w,768] = sign(shuffle(t) - target) * gscale  (gscale = loss_weight / numel). */
int vtp_l1_loss_fwd_bwd(const void* t, const float* target, void* dt, float* loss_sum, int B, int h, int w,
                        float gscale, void* stream);

/* fused AdamW over one flat f32 parameter buffer (torch.optim.AdamW semantics); also refreshes the bf16 copy. */
int vtp_adamw(float* p, const float* g, float* m, float* v, void* p_bf16, long n, float lr, float beta1, float beta2,
              float eps, float weight_decay, int step, float grad_scale, void* stream);
/* same, hyper-parameters read from DEVICE memory: hyper[8] = {lr, beta1, beta2, eps, weight_decay, 1-beta1^t,
 * sqrt(1-beta2^t), grad_scale} -- lets a captured hipGraph of the training step replay with per-step values. */
int vtp_adamw_dev(float* p, const float* g, float* m, float* v, void* p_bf16, long n, const float* hyper, void* stream);
/* the same with a weight-decay exemption table: nodecay4[i] != 0 exempts elements [4i, 4i + 4) (one flag per float4 -- parameters are
 * padded to 4 elements in the flat buffer): biases / norm gains / tokens / logit scale follow the usual AdamW recipe */
int vtp_adamw_dev_masked(float* p, const float* g, float* m, float* v, void* p_bf16, const void* nodecay4, long n, const float* hyper,
                         void* stream);
/* dst[i] (+)= sum_{s<S} slabs[s*stride + i], f32 (split-K partials -> gradient buffer). */
int vtp_reduce_slabs(const float* slabs, long stride, int S, float* dst, long n, int accumulate, void* stream);
/* EMA teacher update t = m*t + (1-m)*s over a flat buffer (vtp.py:388-401). */
int vtp_ema(float* t, const float* s, long n, float momentum, void* stream);
int vtp_ema_dev(float* t, const float* s, long n, const float* momentum /* device scalar */, void* stream);
/* AdamW (vtp_adamw_dev_masked; nodecay4 may be NULL) with the EMA update of the teacher's copy of the same elements fused in
 * (teacher = mom * teacher + (1 - mom) * p_new, vtp.py:388-401; teacher may be NULL; mom = hyper[9]): ONE pass over a gradient
 * bucket for the per-bucket optimizer lane of the training step.  Short-lived blocks (4096 elements each). */
int vtp_adamw_ema_dev(float* p, const float* g, float* m, float* v, float* teacher, const void* nodecay4, long n, const float* hyper,
                      void* stream);

/* ---- CLIP text-tower glue + contrastive head (fp32; clip.hip) -------------------------------------------------
 * embed: x f32 [B*T, D] = table[ids] + pos (modeling_vtp.py:296-297); eot[b] = argmax_t ids[b,t] (text_global_pool
 * 'argmax', text_transformer.py:222-224).  ids are int64 [B, T].  bwd: d_table[ids] += dx (atomics), d_pos += sum_b dx. */
int vtp_embed_tokens(const long* ids, const float* table, const float* pos, float* x, int* eot, int B, int T, int D,
                     void* stream);
int vtp_embed_tokens_bwd(const long* ids, const float* dx, float* d_table, float* d_pos, int B, int T, int D, void* stream);
/* out f32 [B, D] = x[b*T + idx[b]] ; scatter: dx f32 [B*T, D] (and optional bf16 copy) = 0 except those rows = dy. */
int vtp_gather_rows(const float* x, const int* idx, float* out, int B, int T, int D, void* stream);
int vtp_scatter_rows(const float* dy, const int* idx, float* dx, void* dx_bf16, int B, int T, int D, void* stream);
/* F.normalize(x, dim=-1, eps) (modeling_vtp.py:276,310): y = x * inv_norm, inv_norm = 1 / max(||x||, eps); and its backward. */
int vtp_l2norm_fwd(const float* x, float* y, float* inv_norm, int B, int D, float eps, void* stream);
int vtp_l2norm_bwd(const float* dy, const float* y, const float* inv_norm, float* dx, int B, int D, void* stream);
/* Contrastive loss (OpenCLIP ClipLoss, local_loss + gather_with_grad convention; NOT in the reference -> parity unpinned):
 *   logits_i2t = exp(ls) * img_local @ txt_all^T, logits_t2i = exp(ls) * txt_local @ img_all^T  (modeling_vtp.py:329),
 *   labels = label_offset + arange(B_local); loss_sum += 0.5 * (mean CE_i2t + mean CE_t2i).
 * Outputs (overwritten): d_img_local, d_txt_local [B_local, D]; d_img_all, d_txt_all [B_all, D] = the gradient this rank
 * contributes to EVERY rank's features (reduce-scatter them across ranks and add the local slice); d_logit_scale[0] +=
 * dL/d(logit_scale).  scratch: 2 * B_local * B_all floats.  All features f32, L2-normalised by the caller. */
int vtp_clip_loss(const float* img_local, const float* txt_local, const float* img_all, const float* txt_all,
                  const float* logit_scale, int B_local, int B_all, int D, int label_offset, float* loss_sum,
                  float* d_img_local, float* d_txt_local, float* d_img_all, float* d_txt_all, float* d_logit_scale,
                  float* scratch, void* stream);

/* ---- loss-head variants the reference names but does not ship (parity unpinned, KATs = fp64 restatements) ----
 * SigLIP (vtp.py:180,185-188 creates `logit_bias` when init_logit_bias is set; OpenCLIP SigLipLoss):
 *   z = exp(logit_scale) <I_m, T_n> + logit_bias, y = +1 on matching pairs else -1, loss = weight * sum softplus(-y z).
 *   vtp_clip_logits writes exp(ls) <A_m, B_n>; vtp_siglip_pairs turns them IN PLACE into G = dloss/dz and accumulates loss,
 *   d log-scale and d bias; vtp_clip_grad_rows / _cols map G back to the feature gradients (local rows / gathered columns). */
int vtp_clip_logits(const float* A, const float* B, const float* logit_scale, float* logits, int M, int N, int D, void* stream);
int vtp_clip_grad_rows(const float* G, const float* B, const float* logit_scale, float* out, int M, int N, int D, int accumulate,
                       void* stream);
int vtp_clip_grad_cols(const float* G, const float* A, const float* logit_scale, float* out, int M, int N, int D, int accumulate,
                       void* stream);
int vtp_siglip_pairs(float* logits, const float* logit_bias, int B_local, int B_all, int label_offset, float weight,
                     float* loss_sum, float* d_logit_scale, float* d_logit_bias, void* stream);
/* KoLeo (DINOv2 KoLeoLoss) on L2-normalised rows xn [B, D]: loss += -weight * sum_i log(|xn_i - xn_nn(i) + 1e-8| + eps), nn(i) = the
 * other row with the largest dot product; d_xn (ACCUMULATED) receives the gradient w.r.t. xn (both ends of every pair). */
int vtp_koleo(const float* xn, int* nn_scratch, float* d_xn, float* loss_sum, int B, int D, float weight, float eps, void* stream);
/* Sinkhorn-Knopp centring (DINOv2 sinkhorn_knopp_teacher): probs bf16 [T, K] from teacher logits bf16 [T, K] at temperature
 * 1 / inv_temp, n_iters alternating normalisations over prototypes and samples (count = total samples; count_dev / n_rows_dev:
 * the same from device memory for padded buffers).  u [T], v [K], scratch [8 + K + T] f32.  phase -1 = everything (one process);
 * phases 0..3 expose the steps between which a data-parallel caller all-reduces scratch[0] (max), scratch[8..8+K) (sums). */
int vtp_sinkhorn_knopp(const void* logits, float inv_temp, void* probs, float* u, float* v, float* scratch, int T, int K, float count,
                       const float* count_dev, const int* n_rows_dev, int n_iters, int phase, void* stream);

/* ---- fp8 forward path (BASELINE config 5: VTP-L encode -> decode with fp8 MFMA; fp8.hip, gemm8p.hip) --------
 * e4m3 = the OCP fp8 format of gfx950, per-tensor scales:  q = e4m3(clamp(x * scale, -448, 448)).
 * vtp_quantize_e4m3: src bf16 (src_is_f32 = 0) or f32 [n] -> dst bytes [n]; scale from device memory (scale_dev) or the argument.
 * vtp_amax: amax[0] = max(amax[0], max |src|)  (calibration of the scales).   vtp_dequantize_e4m3: bytes -> f32 * inv_scale.
 * vtp_gemm_nt_fp8: C[M,N] = alpha * A8[M,K] B8[N,K]^T (+ bias, + residual | SwiGLU), epilogues VTP_EPI_BF16 / F32 / SWIGLU as
 *   in vtp_gemm_nt, alpha = 1 / (scale_A * scale_B); K, lda, ldb in fp8 elements, multiples of 16; rope_pos / rope_sin / rope_cos /
 *   rope_cols (or nulls / 0): apply_rope fused into the bf16 epilogue exactly as in vtp_gemm_qkv_rope. */
/* norm_fwd with the quantisation fused: y8[M, D] = e4m3(bf16(norm(x)) * q_scale[0]) (q_scale in device memory) */
int vtp_norm_fwd_e4m3(const float* x, const float* w, const float* b, void* y8, const float* q_scale, float* stats, int M, int D,
                      float eps, int kind, void* stream);
int vtp_quantize_e4m3(const void* src, int src_is_f32, void* dst, long n, const float* scale_dev, float scale, void* stream);
int vtp_amax(const void* src, int src_is_f32, long n, float* amax, void* stream);
int vtp_dequantize_e4m3(const void* src, float* dst, long n, float inv_scale, void* stream);
int vtp_gemm_nt_fp8(const void* A8, int lda, const void* B8, int ldb, void* C, int ldc, void* C2, int ldc2, const float* bias,
                    const float* resid, int M, int N, int K, int epilogue, float alpha, const int* rope_pos, const void* rope_sin,
                    const void* rope_cos, int rope_cols, void* stream);

/* ---- tokenizer boundary (tokenizer.hip): the byte-image ends of generation/tokenizer/vtp_tokenizer.py --------
 * vtp_u8_to_images: ToTensor + Normalize(mean, std) (+ horizontal flip) of img_transform (vtp_tokenizer.py:74-81):
 *   img[b,c,y,x] = (float(u8[b,y,xs,c]) / 255 - mean[c]) / std[c], xs = flip ? W-1-x : x.   u8 NHWC -> f32 NCHW, W % 4 == 0.
 * vtp_images_to_u8: the tail of decode_to_images (vtp_tokenizer.py:105-111): Normalize(sub, div) -> * 255 -> clamp(0,255) ->
 *   uint8 (truncation) -> NHWC.   mean3 / std3 / sub3 / div3 are HOST pointers to 3 floats.
 * vtp_latent_channel_stats: sums[0..C) += sum, sums[C..2C) += sum of squares of latents f32 [B, C, hw] per channel, in fp64
 *   (the latents_stats.pt mean / std of the LightningDiT latent dataset, extract_features_vtp.py:124-126). */
int vtp_u8_to_images(const void* u8_nhwc, float* img_nchw, long B, int H, int W, const float* mean3, const float* std3, int flip,
                     void* stream);
int vtp_images_to_u8(const float* img_nchw, void* u8_nhwc, long B, int H, int W, const float* sub3, const float* div3, void* stream);
int vtp_latent_channel_stats(const float* latents, double* sums, long B, int C, int hw, void* stream);

/* ---- self-supervised (DINO / iBOT) head (ssl.hip) -----------------------------------------------------------
 * Token buffers of VTP.get_teacher_forward_outputs / get_student_ssl_outputs (vtp.py:432-439,470-473):
 * dst bf16 [T, D] row t = src row idx[t] (idx[t] < 0: zero row); scatter is the backward (indices are unique). */
int vtp_gather_token_rows(const void* src, const int* idx, void* dst, int T, int D, void* stream);
int vtp_scatter_token_rows(const void* d_dst, const int* idx, void* d_src, int T, int D, void* stream);
/* weight_norm(Linear(C -> K, bias=False)) of DINOHead.last_layer (dino_head.py:47-49): weff bf16 [K,C] = g * v / ||v||_row,
 * weffT bf16 [C,K] (optional), inv_norm f32 [K] = 1/||v||.  bwd: given dW_eff f32 [K,C]: dv += ..., dg += <dW, v/||v||>. */
int vtp_weight_norm_prep(const float* v, const float* g, void* weff, void* weffT, float* inv_norm, int K, int C, void* stream);
int vtp_weight_norm_bwd(const float* dW, const float* v, const float* g, const float* inv_norm, float* dv, float* dg, int K, int C,
                        void* stream);
/* teacher targets: probs bf16 [T,K] = softmax((logits - center) * inv_temp) row-wise (center f32 [K] or NULL). */
int vtp_softmax_center(const void* logits, const float* center, float inv_temp, void* probs, int T, int K, void* stream);
/* the same with the inverse temperature read from device memory (teacher-temperature schedules under hipGraph replay) */
int vtp_softmax_center_dev(const void* logits, const float* center, const float* inv_temp, void* probs, int T, int K, void* stream);
/* student cross-entropy (DINO cls / iBOT patch loss; OUR spec, DINOv2 convention): for student row r with teacher target rows
 * t_idx0[r], t_idx1[r] (-1 = none; t_idx0 < 0 or row_weight 0 = padding row):
 *   loss_sum += w_r * sum_targets( -sum_k p_t[k] * log_softmax(s_r * inv_temp)[k] ),
 *   d_student_logits[r] = w_r * inv_temp * (n_targets * softmax(s_r * inv_temp) - sum_targets p_t)     (bf16). */
int vtp_dino_ce(const void* student_logits, const void* teacher_probs, const int* t_idx0, const int* t_idx1,
                const float* row_weight, float inv_temp, float* loss_sum, void* d_student_logits, int T, int K, void* stream);
/* center = momentum * center + (1 - momentum) * col_sum * inv_count   (teacher-output centering, f32 [K]). */
int vtp_center_ema(float* center, const float* col_sum, float inv_count, const float* count_ptr, float momentum, int K,
                   void* stream);  /* count_ptr != NULL: inv_count = 1 / max(*count_ptr, 1) read on the device */

/* ---- LPIPS perceptual distance (reference: vtp/utils/lpips.py:61-175; VGG16 taps relu1_2 .. relu5_3) ----------------------
 * Activations are zero-bordered NHWC stacks, bf16 [NB, H+2, W+2, C] flattened to pixel rows, preceded and followed by
 * >= W+3 guard rows of zeros; every pointer below addresses row 0 of the stack (after the guard).
 *
 * vtp_conv3x3: y = epilogue(conv3x3_pad1(x)) as an implicit GEMM on the MFMA GEMM kernel.  w bf16 [Cout, taps*Cin] with the
 *   reduction index ordered (ky, kx, ci); taps = 9, or 1 for rows that are already unfolded (first VGG layer: Cin = 32 =
 *   27 unfolded values + 5 zeros, replaces nn.Conv2d(3, 64, 3, padding=1)).  mode 0: y = relu(acc + bias), border rows = 0
 *   (Conv2d + ReLU, lpips.py:131-146).  mode 1 (input gradient; w = flipped-tap transposed weights): y = acc masked by
 *   relu_mask > 0 (bf16 [rows, Cout], the activation this gradient flows into; NULL = no ReLU in front), border rows = 0. */
int vtp_conv3x3(const void* x, const void* w, const float* bias, void* y, const void* relu_mask, int NB, int H, int W, int Cin,
                int Cout, int taps, int mode, void* stream);
/* first-layer unfold with ScalingLayer (lpips.py:103-114) fused: source = reconstruction tokens bf16 [n*hw, 768] (tok; the
 * PixelShuffle(16) of pixel_decoder.py:158-161 is folded into the addressing) or an f32 NCHW image [n,3,H,W] (img); exactly
 * one is non-NULL.  out bf16 [n*(H+2)*(W+2), 32].  shift / scale: host float[3]. */
int vtp_lpips_unfold3(const void* tok, const float* img, void* out, int n, int H, int W, const float* shift, const float* scale,
                      void* stream);
/* dt bf16 [n*hw, 768] += fold(dA bf16 [n*(H+2)*(W+2), 32]) / scale_c: gradient of vtp_lpips_unfold3 w.r.t. tok. */
int vtp_lpips_fold3_bwd(const void* dA, void* dt, int n, int H, int W, const float* scale, void* stream);
/* nn.MaxPool2d(2, 2) on the bordered layout; bwd fuses the ReLU mask of Y and an optional tap gradient:
 * dY = (Y > 0) * (route(dP) + tap), route = first maximum of each 2x2 window in row-major order. */
int vtp_maxpool2_fwd(const void* in, void* out, int NB, int H, int W, int C, void* stream);
int vtp_maxpool2_bwd(const void* Y, const void* dP, const void* tap, void* dY, int NB, int H, int W, int C, void* stream);
/* one LPIPS tap: val[b] += mean_pixels sum_c w_c (f0/|f0| - f1/|f1|)^2_c  (normalize_tensor + NetLinLayer + spatial_average,
 * lpips.py:84-100,169-175); df0 (bf16, may be NULL) = gscale * d val / d f0, masked by f0 > 0.  f0, f1 bf16 [n*(H+2)*(W+2), C]. */
int vtp_lpips_tap(const void* f0, const void* f1, const float* w, float* val, void* df0, int n, int H, int W, int C,
                  float gscale, void* stream);

#ifdef __cplusplus
}
#endif
#endif
