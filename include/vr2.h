/* vr2.h -- C ABI of libvr2.so, the MI355X (gfx950) kernels of the Video-R2 GRPO rollout-and-update hot path.
 *
 * Boundary contract (last row).  The reference (mbzuai-oryx/Video-R2) is 100 % Python and has no
 * FFI layer of its own: every arithmetic op on its hot path is reached through HF transformers / TRL calls.  Each
 * entry point below therefore names the reference CALL SITE (train/...:line under /root/reference) and the
 * third-party implementation (HF: transformers/models/qwen2_5_vl/modeling_qwen2_5_vl.py, pinned 4.51.1) it replaces.
 *
 * Conventions
 *   - plain C: device pointers as void*, sizes as int64_t, strides in ELEMENTS, `stream` = hipStream_t as void*
 *   - the caller owns every buffer; no entry point allocates, frees or synchronises (graph-capturable)
 *   - returns 0 on success, <0 on error; message via vr2_last_error() (thread-local)
 *   - bf16 = 16-bit brain float storage; f32 = IEEE binary32; i32 = int32_t
 *   - one host thread per GPU; no global state
 */
#ifndef VR2_H
#define VR2_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

const char* vr2_last_error(void);
int vr2_version(void);
/* launch-time kernel-variant knobs for A/B measurements (defaults = the measured best).  key 0: decode GEMV for 16 < M <= 128
 * rows -- above 128 rows, up to 256, the register-ring kernel always runs -- (0 = LDS-staged kernel of round 1, 1 = one-workgroup-per-CU register-ring kernel with 3 weight chunks in flight per wave,
 * 2 = the same with 1: the default).  key 1: ablations of that kernel for profiles/ (results wrong).  key 2: x is chunk-major.
 * keys 3 / 4: K/V tiles a decode-attention suffix wave / shared-prefix workgroup keeps in flight in registers (1-4 / 1-3; defaults
 * 2 / 1, profiles/r04_attn_decode_depth.md). */
int vr2_tune_set(int key, int value);

/* ---- GEMMs ------------------------------------------------------------------------------------------------- */
/* epilogue codes of vr2_gemm_bf16 */
#define VR2_EPI_BF16 0      /* C bf16  = A W^T + bias                                         */
#define VR2_EPI_F32 1       /* C f32   = A W^T + bias                                         */
#define VR2_EPI_ADD_F32 2   /* C f32  += A W^T + bias   (residual stream, HF :754, :321-322)  */
#define VR2_EPI_GELU_BF16 3 /* C bf16  = gelu_erf(A W^T + bias)   (merger, HF :143-146)       */
#define VR2_EPI_SWIGLU_BF16 4 /* W = interleaved gate/up 16-row tiles (N = 2*I): C[M][I] bf16 = silu(g)*u (HF :96, :553) */

/* WEIGHT LAYOUT ("tile-packed", built once at load time by video_r2_amd/weights.py:pack_tiles): W[N][K] is stored as
 * 16-row x 32-k tiles of 1 KiB, tile (n/16, k/32) at byte ((n/16)*(K/32) + k/32) * 1024, and inside a tile element
 * (n, k) at 16-byte slot 16*((k%32)/8) + n%16, position k%8 -- the v_mfma_f32_16x16x32_bf16 A-fragment order, so one
 * wave-level 16-byte load fetches one contiguous KiB that is already an MFMA operand.  N is padded to 16, K to 64. */

/* C[M,N] (op)= A[M,K] * W[N,K]^T + bias.  Replaces every nn.Linear on the prefill / teacher-forced path:
 * ViT patch-embed Conv3d-as-GEMM HF:99-122 (K1), ViT qkv/proj/MLP HF:211-291,85-96 (K3), merger HF:137-150 (K7),
 * LM q/k/v/o/gate/up/down HF:634-637,541-554 (K11), lm_head HF:1386-1387 (K15); reached from
 * train/src/trainer/grpo_trainer.py:725-727 and :783-786.  K % 64 == 0 (pad operands), N % 4 == 0. */
int vr2_gemm_bf16(const void* A_bf16, int64_t lda, const void* W_packed_bf16, const void* bias_f32, void* C, int64_t ldc,
                  int64_t M, int64_t N, int64_t K, int epilogue, void* stream);
/* fp32 scratch for the GEMMs' split launches: the <= 128 tail rows of a 256-row tiling with few output tiles (N = 3584 at
 * T = 9280: the o / down projections and their dgrads) run as k-parts that each store a partial tile here and are summed in part
 * order -- deterministic, unlike atomic adds, so two passes over the same weights give the same bits
 * (grpo_trainer.py:723-741 called for the policy :1094 and the reference :826).  8 MiB covers every shape; the caller owns the
 * memory and runs the GEMMs that use it on one stream.  Without a workspace those tail rows run as full-K tiles (slower, same
 * determinism).  ptr = NULL, bytes = 0 unregisters. */
int vr2_set_gemm_workspace(void* ptr, int64_t bytes);

/* Decode-step weight streaming for M <= 256 rows (above 128: K % 64 == 0 and >= 4 chunks of 64 k per part): out[s][m][n] = sum over K-slice s of x[m][k] W[n][k] (fp32 partial
 * slabs, summed by the consumer).  Replaces the 1-token forward of HF generate, generation/utils.py:2876-2937 (K11
 * decode rows, K14 lm_head). */
int vr2_gemv_bf16(const void* x_bf16, int64_t ldx, int64_t M, const void* W_packed_bf16, int64_t N, int64_t K,
                  void* out_f32, int64_t split_stride, int64_t ldo, int nsplit, void* stream);
/* act[m][i] = silu(x W_gate[i]) * (x W_up[i]); W = gate/up interleaved by 16-row tiles  (Qwen2MLP HF:541-554) */
int vr2_gemv_gateup_silu_bf16(const void* x_bf16, int64_t ldx, int64_t M, const void* W_packed_bf16, int64_t I, int64_t K,
                              void* act_bf16, int64_t ldo, void* stream);

/* the same two decode projections with per-channel fp8 (e4m3, OCP) WEIGHTS (BASELINE config 5): W_packed_fp8 in the 16-row x
 * 64-k tile layout of weights.py pack_tiles_fp8, scale_f32[n] per PACKED row (w ~= fp8 * scale[n]); activations, MFMA and
 * outputs as above (weight-only quantisation: the fp8 bytes are expanded to bf16 in registers).  K % 64 == 0. */
int vr2_gemv_w8(const void* x_bf16, int64_t ldx, int64_t M, const void* W_packed_fp8, const void* scale_f32, int64_t N, int64_t K,
                void* out_f32, int64_t split_stride, int64_t ldo, int nsplit, void* stream);
int vr2_gemv_gateup_silu_w8(const void* x_bf16, int64_t ldx, int64_t M, const void* W_packed_fp8, const void* scale_f32, int64_t I,
                            int64_t K, void* act_bf16, int64_t ldo, void* stream);

/* BASELINE config 5 on the fp8 MFMA (v_mfma_f32_16x16x32_fp8_fp8): fp8 ACTIVATIONS x fp8 weights for 16 < M <= 128 decode rows.
 * vr2_quant_rows_fp8: x bf16 [M][K] -> e4m3 bytes [M][ldq] (K zero-padded to 64) + scale[m] = amax / 448; inside every 64-k block
 * the 16-byte slot s holds k = 8s..8s+7 then k = 32+8s..32+8s+7 (two MFMA k-steps per 16-byte fragment read).  The two GEMVs
 * take that buffer + scales and the weight copies of vr2_gemv_w8; out = (sum_k q_x q_w) * scale_x[m] * scale_w[n].  K % 128 == 0. */
int vr2_quant_rows_fp8(const void* x_bf16, int64_t ldx, int64_t M, int64_t K, void* q_fp8, int64_t ldq, void* scale_f32, void* stream);
int vr2_gemv_a8w8(const void* x_fp8, int64_t ldx8, const void* xscale_f32, int64_t M, const void* W_packed_fp8, const void* wscale_f32,
                  int64_t N, int64_t K, void* out_f32, int64_t split_stride, int64_t ldo, int nsplit, void* stream);
int vr2_gemv_gateup_silu_a8w8(const void* x_fp8, int64_t ldx8, const void* xscale_f32, int64_t M, const void* W_packed_fp8,
                              const void* wscale_f32, int64_t I, int64_t K, void* act_bf16, int64_t ldo, void* stream);
/* The same operands at prefill / scoring row counts (config 5, the no-grad passes: rollout prefill, reference prefill +
 * log-probs -- the nn.Linear calls vr2_gemm_bf16 replaces, with quantised operands): C (op)= (q_x q_w^T) * scale_x[m] * scale_w[n]
 * + bias, epilogues VR2_EPI_BF16 / _F32 / _ADD_F32 / _SWIGLU_BF16.  Tilings, staging and LDS images of vr2_gemm_bf16 on half the
 * bytes per k; the multiply is v_mfma_scale_f32_16x16x128_f8f6f4 with unit block scales (2x the bf16 MFMA rate).  K % 128 == 0. */
int vr2_gemm_a8w8(const void* x_fp8, int64_t ldx8, const void* xscale_f32, const void* W_packed_fp8, const void* wscale_f32,
                  const void* bias_f32, void* C, int64_t ldc, int64_t M, int64_t N, int64_t K, int epilogue, void* stream);

/* ---- row kernels ------------------------------------------------------------------------------------------- */
/* x (+)= bias + sum_s partial[s] ; y = bf16(rmsnorm(x) * w).  Qwen2_5_VLRMSNorm HF:65-79 (K2, K10) fused with the
 * residual add of the preceding projection.  partial / bias / y may be NULL. */
int vr2_add_rmsnorm(void* x_f32, int64_t ldx, const void* partial_f32, int64_t split_stride, int64_t ldp, int nsplit,
                    const void* bias_f32, const void* w_f32, void* y_bf16, int64_t ldy, int64_t M, int64_t H, float eps,
                    void* stream);
/* the same with vr2_quant_rows_fp8 fused in: besides y, the row's e4m3 bytes (fragment order) and scale.  H % 64 == 0, H <= 4096 */
int vr2_add_rmsnorm_q8(void* x_f32, int64_t ldx, const void* partial_f32, int64_t split_stride, int64_t ldp, int nsplit,
                       const void* bias_f32, const void* w_f32, void* y_bf16, int64_t ldy, int64_t M, int64_t H, float eps,
                       void* q_fp8, int64_t ldq, void* qscale_f32, void* stream);
/* dst[m][0:Kpad] = bf16(src[row_idx ? row_idx[m] : m][0:K]) zero padded: pixel_values cast + window permutation
 * gather (HF :436-439, K6) in one pass. */
int vr2_cast_rows_f32_bf16(const void* src_f32, int64_t lds, const void* row_idx_i32, void* dst_bf16, int64_t ldd,
                           int64_t M, int64_t K, int64_t Kpad, void* stream);
/* one axis of the antialiased bicubic resize of uint8 planes [planes][in_rows][in_cols] (vertical = 0: along columns,
 * 1: along rows), integer taps from host/frames.py resize_taps: first tap, tap count, int16 weights [out][kmax], fractional
 * bits.  Two calls (width, then height over the uint8 intermediate) reproduce bit for bit torch's uint8 antialias kernel,
 * i.e. torchvision.transforms.functional.resize(BICUBIC, antialias=True) of the reference, vision_process.py:802-807. */
int vr2_resize_axis_u8(const void* src_u8, void* dst_u8, const void* first_i32, const void* count_i32, const void* weights_i16,
                       int64_t kmax, int64_t bits, int64_t planes, int64_t in_rows, int64_t in_cols, int64_t out_rows,
                       int64_t out_cols, int vertical, void* stream);
/* uint8 frames [n][3][H][W] (device) -> the patch-embed GEMM input: row m = patch row_idx[m] (processor order (t, h/merge,
 * w/merge, merge_h, merge_w); NULL = identity), columns (c, tp, ph, pw), value bf16((x - mean[c]) / std[c]) with the
 * rescaled constants (mean3 / std3: 3 host floats each), zero padded to Kpad; an odd frame count repeats the last frame.
 * Fuses HF qwen2_vl/video_processing_qwen2_vl.py:236-274 (rescale + normalise + patchify permute) with the input cast of K1. */
int vr2_patchify_u8(const void* frames_u8, int64_t n_frames, int64_t H, int64_t W, const void* row_idx_i32,
                    const float* mean3_host, const float* std3_host, void* dst_bf16, int64_t ldd, int64_t P,
                    int64_t patch_size, int64_t temporal_patch_size, int64_t merge_size, int64_t Kpad, void* stream);
/* cos/sin tables: ViT 2-D rope HF:125-134,441-446 (K4); M-RoPE HF:525-538,579-585 (K9) */
int vr2_vision_rope_table(const void* pos_hw_i32, const void* inv_freq_f32, void* cos_f32, void* sin_f32, int64_t P,
                          int64_t head_dim, void* stream);
int vr2_mrope_table(const void* pos3_i32, int64_t T, const void* inv_freq_f32, const void* section_i32, void* cos_f32,
                    void* sin_f32, int64_t head_dim, void* stream);
/* rotate-half rope on fp32 [q heads | k heads | v heads] rows straight from the QKV GEMM (HF:160-171 vision, :557-599
 * LM): q -> q_out rows t, roped k and v -> k_out / v_out rows kv_row0 + t (KV-cache append = DynamicCache.update,
 * HF:667-668, K13; for the ViT the k / v column blocks of a packed bf16 buffer). */
int vr2_rope_qkv(const void* qkv_f32, int64_t ld, const void* cos_f32, const void* sin_f32, int64_t T, int64_t nq,
                 int64_t nkv, int64_t head_dim, void* q_out_bf16, int64_t ldq, void* k_out_bf16, void* v_out_bf16,
                 int64_t ldkv, int64_t kv_row0, void* stream);
/* the same for a BATCH of prompts prefilled in one row space (row t = prompt t / seg_rows, token t % seg_rows): k / v rows go
 * to per-prompt cache slots seg_stride rows apart, row (t / seg_rows) * seg_stride + t % seg_rows */
int vr2_rope_qkv_seg(const void* qkv_f32, int64_t ld, const void* cos_f32, const void* sin_f32, int64_t T, int64_t nq,
                     int64_t nkv, int64_t head_dim, void* q_out_bf16, int64_t ldq, void* k_out_bf16, void* v_out_bf16,
                     int64_t ldkv, int64_t seg_rows, int64_t seg_stride, void* stream);
/* decode: qkv = bias + sum partial; rope at position pos[g] + step[0]; q -> q_out, k/v -> cache row g*rows_per_seq+step */
int vr2_rope_append_decode(const void* partial_f32, int64_t split_stride, int64_t ldp, int nsplit, const void* bias_f32,
                           const void* pos_i32, const void* step_i32, const void* inv_freq_f32, int64_t G, int64_t nq,
                           int64_t nkv, int64_t head_dim, void* q_out_bf16, int64_t ldq, void* kcache_bf16,
                           void* vcache_bf16, int64_t ldkv, int64_t rows_per_seq, void* stream);
/* embed_tokens lookup (HF:807-808) and masked_scatter of the merged video tokens (HF:1206-1223, K8) */
int vr2_embed_rows(const void* ids_i32, const void* table_bf16, int64_t ldt, void* out_f32, int64_t ldo, int64_t T,
                   int64_t H, void* stream);
int vr2_scatter_rows_f32(const void* src_f32, int64_t lds, const void* idx_i32, void* dst_f32, int64_t ldd, int64_t T,
                         int64_t H, void* stream);
int vr2_add_int(void* p_i32, int v, void* stream);

/* ---- attention --------------------------------------------------------------------------------------------- */
/* work item (8 x int32): q_row0, nq, ka_row0, ka_len, kb_row0, kb_len, q_causal0, pad.  Query i of the tile attends
 * all ka_len rows of source A and row j of source B iff j <= q_causal0 + i.  Covers ViT windowed/full varlen attention
 * (HF:244-291, K5; A empty, B = segment, non-causal), LM causal prefill (HF:641-689, K12) and teacher-forced scoring
 * against a shared prompt-prefix cache (grpo_trainer.py:725-727).  head_dim 80 or 128.  lse_out_f32 (nullable,
 * [rows][n_heads]) receives the log-sum-exp of the scaled scores for the backward pass. */
int vr2_attn_tok(const void* work_i32x8, int64_t n_work, const void* Q_bf16, int64_t ldq, const void* KA_bf16,
                 const void* VA_bf16, const void* KB_bf16, const void* VB_bf16, int64_t ldk, void* O_bf16, int64_t ldo,
                 int64_t n_heads, int64_t group, int64_t head_dim, float scale, void* lse_out_f32, void* stream);
/* decode attention of ONE new token per sequence against [its group's shared prefix | own suffix rows 0..step], split over
 * key chunks + combine (HF:667-689 with DynamicCache, generation/utils.py:2876-2937, K13).  Sequence g belongs to prompt
 * group g / seqs_per_group; group p's prefix KV starts at Kprefix + p*prefix_group_stride, length prefix_len_i32[p]
 * (<= prefix_chunk * n_prefix_chunks); the prefix chunks are computed once per GROUP for all its sequences, the suffix chunks
 * per sequence (step + 1 <= chunk * n_suffix_chunks); both chunk sizes are multiples of 64.  part_o: [G][n_heads][n_prefix_chunks + n_suffix_chunks][head_dim] f32,
 * part_ml: [..][2] f32.  finished_i32 (nullable, [G]): a sequence whose flag is set has produced its EOS -- every later token is
 * the pad id whatever the model computes (generation/utils.py:2928-2929) -- and its own keys are not walked any more. */
int vr2_attn_decode(const void* Q_bf16, int64_t ldq, const void* Kprefix, const void* Vprefix, const void* Ksuffix,
                    const void* Vsuffix, int64_t ldk, const void* prefix_len_i32, int64_t seqs_per_group,
                    int64_t prefix_group_stride, int64_t rows_per_seq, const void* step_i32,
                    int64_t chunk, int64_t prefix_chunk, int64_t n_prefix_chunks, int64_t n_suffix_chunks, int64_t G,
                    int64_t n_heads, int64_t group,
                    int64_t head_dim, float scale, void* part_o_f32, void* part_ml_f32, void* out_bf16, int64_t ldo,
                    const void* finished_i32, void* stream);

/* ---- token level ------------------------------------------------------------------------------------------- */
/* temperature / top-k / multinomial + EOS-pad bookkeeping (HF generation/logits_process.py:238,542;
 * generation/utils.py:2894-2937; knobs train/src/params.py:193-195; K14).  seed_dev_u64 (device, nullable) overrides
 * `seed` so a captured decode graph can be re-seeded without re-capture. */
int vr2_sample_topk(const void* logits_f32, int64_t ldl, int64_t G, int64_t V, float temperature, int top_k,
                    uint64_t seed, const void* seed_dev_u64, const void* step_i32, void* finished_i32, int eos_id,
                    int pad_id, void* next_ids_i32, void* tokens_i32, int64_t ldt, void* cand_ws, void* stream);
/* the same with HF's nucleus (top_p) and min-p warpers applied after top-k, in HF's order (TopPLogitsWarper, MinPLogitsWarper of
 * generation/logits_process.py; the reference forwards both flags to generate, grpo_trainer.py:559-570).  top_p = 1 and min_p = 0
 * are vr2_sample_topk; otherwise 1 <= top_k <= 64 and the candidate workspace are required (the warpers run on the top-k set). */
int vr2_sample_topk_p(const void* logits_f32, int64_t ldl, int64_t G, int64_t V, float temperature, int top_k, float top_p,
                      float min_p, uint64_t seed, const void* seed_dev_u64, const void* step_i32, void* finished_i32, int eos_id,
                      int pad_id, void* next_ids_i32, void* tokens_i32, int64_t ldt, void* cand_ws, void* stream);
/* HF RepetitionPenaltyLogitsProcessor on the raw logits, before the warpers: ids seen so far (prompt + generated) get
 * logit < 0 ? logit * penalty : logit / penalty.  seen_bits: [G][words_per_row] uint32 bitmap kept by the caller across steps
 * (prompt ids set beforehand); new_ids (nullable): the token each row appended since the last call, set here. */
int vr2_repetition_penalty(void* logits_f32, int64_t ldl, int64_t G, int64_t V, float penalty, void* seen_bits_u32,
                           int64_t words_per_row, const void* new_ids_i32, void* stream);
/* bytes of the optional two-stage workspace `cand_ws` (NULL = single-pass kernel) */
int64_t vr2_sample_ws_bytes(int64_t G);
/* logp[r] = logits[r][tok[r]]/T - logsumexp(logits[r]/T)  (trl.selective_log_softmax via grpo_trainer.py:738-739, K15) */
int vr2_lse_gather(const void* logits_f32, int64_t ldl, int64_t R, int64_t V, const void* tok_i32, float temperature,
                   void* logp_f32, void* lse_f32, void* stream);
/* GRPO loss + d loss / d logp + metrics (grpo_trainer.py:1099-1177, K16).  loss_type: 0 grpo 1 bnpo 2 dr_grpo 3 gmpo
 * 4 mapo 5 mapo_bnpo.  norm_rows: rows per normalisation block of the token-mean aggregations (bnpo, mapo_bnpo) and of the
 * kl / clip metrics = the reference's per-rank micro-batch (per_device_train_batch_size completions; blocks are averaged as
 * its ranks and accumulation steps are); <= 0: the whole batch is one block.
 * out8 = {loss, kl, clip_low, clip_high, clip_region, n_tokens, -, -} */
int vr2_grpo_loss(const void* logp_f32, const void* ref_logp_f32, const void* old_logp_f32, const void* mask_i32,
                  const void* adv_f32, int64_t B, int64_t C, float beta, float eps_low, float eps_high, int loss_type,
                  float max_completion_length, int64_t norm_rows, void* out8_f32, void* dlogp_f32, void* stream);

/* ---- update path (backward + optimiser; autograd of the HF forward reached from grpo_trainer.py:1094-1096, K17/K18) ---- */
/* weight gradient: C[M][N] f32 (+)= sum_k A[k][m] B[k][n]; A = dY [tokens][out], B = X [tokens][in] (both row-major bf16) */
int vr2_gemm_tn_bf16(const void* A_bf16, int64_t lda, const void* B_bf16, int64_t ldb, void* C_f32, int64_t ldc, int64_t M,
                     int64_t N, int64_t K, int accumulate, void* stream);
/* training-forward MLP input GEMM: act = silu(g)*u (one rounding, as VR2_EPI_SWIGLU_BF16) AND the bf16 pre-activations gu */
int vr2_gemm_swiglu_save(const void* A_bf16, int64_t lda, const void* W_packed_bf16, const void* bias_f32, void* act_bf16,
                         int64_t ldact, void* gu_bf16, int64_t ldgu, int64_t M, int64_t N, int64_t K, void* stream);
/* SwiGLU on saved interleaved pre-activations and its backward */
int vr2_swiglu_fwd(const void* gu_bf16, int64_t ldg, void* act_bf16, int64_t lda, int64_t T, int64_t I16, void* stream);
int vr2_swiglu_bwd(const void* gu_bf16, int64_t ldg, const void* dact_bf16, int64_t ldd, void* dgu_bf16, int64_t ldo, int64_t T,
                   int64_t I16, void* stream);
/* erf-GELU on saved fp32 pre-activations z (h = bf16(gelu(z)): the values the fused VR2_EPI_GELU_BF16 epilogue produces) and its
 * backward dz = dh * (Phi(z) + z phi(z)) -- the merger's fc1 (HF modeling_qwen2_5_vl.py:143-146) when --freeze_merger False */
int vr2_gelu_fwd(const void* z_f32, int64_t ldz, void* h_bf16, int64_t ldh, int64_t T, int64_t N, void* stream);
int vr2_gelu_bwd(const void* z_f32, int64_t ldz, const void* dh_bf16, int64_t ldd, void* dz_bf16, int64_t ldo, int64_t T, int64_t N,
                 void* stream);
/* Dropout of the LoRA branch's input (peft lora.Linear.forward: lora_dropout(x), torch.nn.Dropout semantics: kept elements scaled by
 * 1 / (1 - p)); keep(t, k) is a pure function of (seed, salt, t, k), so vr2_dropout_add redraws the mask in the backward pass:
 * dst[t][k] += keep ? g[t][k] / (1 - p) : 0, dst fp32 or bf16.  Replaces torch.nn.Dropout inside peft's LoRA layers as configured
 * by /root/reference/train/src/train/train_sft.py:155-161 (lora_dropout) */
int vr2_dropout_bf16(const void* x_bf16, int64_t ldx, void* out_bf16, int64_t ldo, int64_t T, int64_t K, float p, uint32_t seed,
                     uint32_t salt, void* stream);
int vr2_dropout_add(const void* g_f32, int64_t ldg, void* dst, int64_t ldd, int dst_is_bf16, int64_t T, int64_t K, float p,
                    uint32_t seed, uint32_t salt, void* stream);
/* RMSNorm backward: dx (+)= d/dx of rmsnorm(x)*w against dy; dw += column sums (rinv = per-row scratch [M]); dx_bf16: optional
 * bf16 copy of the final dx rows (what the next dgrad / wgrad GEMMs read) */
int vr2_rmsnorm_bwd(const void* x_f32, int64_t ldx, const void* w_f32, const void* dy_f32, int64_t ldy, void* dx_f32,
                    int64_t lddx, void* rinv_f32, void* dw_f32, int64_t M, int64_t H, float eps, int accumulate,
                    void* dx_bf16_or_null, int64_t lddb, void* stream);
int vr2_colsum_bf16(const void* a_bf16, int64_t lda, void* out_f32, int64_t M, int64_t N, void* stream);
/* inverse rotation of dq / dk, v passthrough -> packed [q|k|v] gradient rows */
int vr2_rope_bwd(const void* dq_bf16, int64_t ldq, const void* dk_f32, const void* dv_f32, int64_t ldkv, const void* cos_f32,
                 const void* sin_f32, int64_t T, int64_t nq, int64_t nkv, int64_t head_dim, void* dqkv_bf16, int64_t ld,
                 void* stream);
/* flash-style attention backward (head_dim 128).  vr2_attn_delta: rowc_f32 [2][n_heads][T] = the per-row constants, transposed:
 * rowc[0][h][t] = -lse[t][h] * log2(e), rowc[1][h][t] = delta = rowsum(dO o O).  dQ query-centric (work items as vr2_attn_tok,
 * up to 128 query rows each); dK/dV key-centric into ZEROED fp32 buffers (work items add their share).  key work item
 * (8 x int32): k_row0, nk (<= 128), q1_row0, q1_len (causal: key row <= query row), q2_row0, q2_len (all visible), head0,
 * nheads (the q heads head0 .. head0 + nheads - 1 of the KV head's GQA group; nheads = 0: the whole group) -- a long sweep is
 * split over several items by the host, by query range and / or by heads */
int vr2_attn_delta(const void* O_bf16, const void* dO_bf16, int64_t ld, int64_t T, int64_t n_heads, int64_t head_dim,
                   const void* lse_f32, void* rowc_f32, void* stream);
int vr2_attn_bwd_dq(const void* work_i32x8, int64_t n_work, const void* Q, int64_t ldq, const void* KA, const void* VA,
                    const void* KB, const void* VB, int64_t ldk, const void* dO, const void* rowc_f32, int64_t T,
                    void* dQ, int64_t n_heads, int64_t group, int64_t head_dim, float scale, void* stream);
int vr2_attn_bwd_dkv(const void* keywork_i32x8, int64_t n_work, const void* Q, int64_t ldq, const void* K, const void* V,
                     int64_t ldk, const void* dO, const void* rowc_f32, int64_t T, void* dK_f32, void* dV_f32,
                     int64_t n_heads, int64_t group, int64_t head_dim, float scale, void* stream);
/* dz = dlogp/T * (onehot(tok) - softmax(z/T)) as bf16 rows (log-prob head backward, grpo_trainer.py:738-739) */
int vr2_dlogits(const void* logits_f32, int64_t ldl, int64_t R, int64_t V, const void* tok_i32, const void* lse_f32,
                const void* dlogp_f32, float temperature, void* out_bf16, int64_t ldo, void* stream);
int vr2_embed_bwd(const void* ids_i32, const void* skip_id_i32, const void* dx_f32, int64_t ldx, void* dtable_f32, int64_t ldt,
                  int64_t T, int64_t H, void* stream);
/* global-norm clipping + fused AdamW on fp32 masters that re-emits the bf16 compute copies (packed W, packed W^T, rows) */
/* global gradient norm without atomics (bitwise reproducible, identical on every data-parallel replica): tensor number `slot`
 * writes vr2_sumsq_slot_floats() per-block partial sums of squares into partials_f32[slot * that .. ]; vr2_clip_coef adds the
 * partials of n_slots tensors in a fixed order: coef = min(1, max_norm / (norm * pre_scale + 1e-6)) * pre_scale. */
int64_t vr2_sumsq_slot_floats(void);
int vr2_sumsq_f32(const void* g_f32, int64_t n, void* partials_f32, int64_t slot, void* stream);
int vr2_clip_coef(const void* partials_f32, int64_t n_slots, float max_norm, float pre_scale, void* coef_f32, void* norm_f32,
                  void* stream);
int vr2_adamw(void* p_f32, const void* g_f32, void* m_f32, void* v_f32, int64_t N, int64_t K, float lr, float beta1, float beta2,
              float eps, float weight_decay, int64_t step, const void* gscale_f32, void* w_packed_bf16, int64_t kpad,
              void* wt_packed_bf16, int64_t npad, void* w_rows_bf16, int64_t ld_rows, void* stream);

/* ---- timing / graphs (plumbing) ---------------------------------------------------------------------------- */
int vr2_event_create(void** ev);
int vr2_event_record(void* ev, void* stream);
int vr2_event_elapsed_ms(void* start, void* stop, float* ms);
int vr2_event_destroy(void* ev);
int vr2_graph_begin(void* stream);
int vr2_graph_end(void* stream, void** exec_out);
int vr2_graph_launch(void* exec, void* stream);
int vr2_graph_destroy(void* exec);

#ifdef __cplusplus
}
#endif
#endif
