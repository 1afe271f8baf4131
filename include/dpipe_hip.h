/* dpipe_hip.h -- C ABI of libdpipe_hip.so, the MI355X (gfx950) kernel layer of the pipeline-parallel training step.
 *
 * Boundary B3 of SURVEY.md section 8(b).  The reference (tdrussell/diffusion-pipe) has no native layer: every op
 * below replaces a PyTorch / flash-attn / DeepSpeed call made from Python on the train_batch path.  Each entry
 * point cites the reference call site it stands in for.  Conventions:
 *   - plain pointers and sizes only; PyTorch (or any host) owns every buffer; no ownership transfer;
 *   - `stream` is a hipStream_t passed as void*; launches are asynchronous on that stream;
 *   - return 0 on success, negative on argument / support errors, positive = hipError_t of the failed launch;
 *     dpipe_last_error() returns a thread-local description;
 *   - dtype codes: 0 = bf16, 1 = fp32; all reductions / statistics are fp32;
 *   - "16-byte vector" requirements: pointers 16-byte aligned, innermost extents multiples of 8 (bf16) / 4 (fp32)
 *     unless stated otherwise.  Host wrappers check and fail loudly rather than fall back.
 */
#ifndef DPIPE_HIP_H
#define DPIPE_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

#define DPIPE_DTYPE_BF16 0
#define DPIPE_DTYPE_F32 1

#define DPIPE_ACT_NONE 0
#define DPIPE_ACT_GELU_TANH 1
#define DPIPE_ACT_GELU_ERF 2
#define DPIPE_ACT_SILU 3
#define DPIPE_ACT_QUICK_GELU 4
/* dpipe_gemm_ex / dpipe_gemm_group only (ABI 9): `act` = DPIPE_ACT_GEGLU_BWD | activation turns the epilogue into the backward of a GEGLU that FEEDS the Linear whose
 * dgrad this GEMM is (diffusers FeedForward behind models/sdxl.py:797-865: net = [GEGLU, Dropout, Linear]):  acc = dy [M, N];  `residual` = the GEGLU's input
 * h [M, 2 N] (value | gate halves, pitch ldr), C = dh [M, 2 N] (pitch ldc):  dh[m, n] = dy * act(gate),  dh[m, N + n] = dy * value * act'(gate).  bf16, N % 4 == 0,
 * 8-byte aligned h / dh, no bias, no accumulation; otherwise DPIPE_ERR_UNSUPPORTED (the caller runs dpipe_geglu_bwd after a plain dgrad). */
#define DPIPE_ACT_GEGLU_BWD 16

#define DPIPE_LOSS_MSE 0
#define DPIPE_LOSS_HUBER 1
#define DPIPE_LOSS_SMOOTH_L1 2

/* ABI version: bumped whenever a signature of this header changes; the host binding refuses a library of another version. */
#define DPIPE_ABI_VERSION 10
int dpipe_version(void);
const char* dpipe_last_error(void);
/* Kernel-selection options (process-wide; for A/B timing and for testing the fallback kernels -- every default is the measured-faster choice).
 * value -1 = unset: the environment variable of the same name (DPIPE_ATTN_FWD_DMA ...) if present, else the default. */
#define DPIPE_OPT_ATTN_FWD_DMA 0    /* 1 (default): LDS-DMA flash-attention forward; 0: register-staged kernel */
#define DPIPE_OPT_ATTN_BWD_DMA 1    /* 1 (default): LDS-DMA dQ / dK / dV kernels (delta fused into dQ); 0: register-staged kernels + attn_delta */
#define DPIPE_OPT_ATTN_DQ8 2        /* register-staged path, head dim 128: 1 (default) 8-wave dQ kernel for long sequences */
#define DPIPE_OPT_ATTN_DKV_SPLIT 3  /* head dim 128, long key sequences: 1 (default) dV and dK as two 8-wave kernels; 0: one pass */
#define DPIPE_OPT_GEMM_SHALLOW 4    /* ring depth of the plain GEMM's 128^2 tile.  0 (default): the 3-deep 96 KiB ring -- the fastest launch in isolation; non-zero: the 2-deep
                                      64 KiB ring (two workgroups per CU) -- slower alone, faster when concurrent streams share the chip: the engine selects it for
                                      >= 2 micro-batch lanes */
#define DPIPE_OPT_GEMM_BIG_TILES 5  /* fewest 128^2 output tiles for which the plain GEMM takes the 128^2 tile instead of 64^2 (default 128 = best isolated launch; lower under
                                      concurrent lanes, where CU time per FLOP is what counts: the engine's choice) */
/* (ABI 6, round 5: options that lost their measurements twice are gone -- DPIPE_OPT_GEMM_SKINNY (128 x 64 tile for M <= 128: 56 vs 39 ms per step) and
 *  DPIPE_OPT_ATTN_BIG_WG (256-row attention workgroups from fewer workgroups on: neutral under lanes); the rule values they overrode are fixed in the dispatchers) */
#define DPIPE_OPTION_COUNT 6
int dpipe_set_option(int option, int value);
int dpipe_get_option(int option);    /* the effective explicit / environment value, -1 if neither is set */
/* Number of compute units / name of device `dev`; used by the host to sanity-check it runs on gfx950. */
int dpipe_device_info(int dev, int* cu_count, char* arch_name, int arch_name_len);

/* ---- C1 / C2 stage-to-stage point-to-point over RCCL / xGMI -- the per-micro-batch SendActivation / RecvActivation / SendGrad / RecvGrad of the
 * reference's schedule (utils/patches.py:126-160 -> DeepSpeed p2p.send / recv -> torch.distributed isend / irecv -> NCCL).  One communicator
 * per neighbour pair (world = 2) or per pipe group; `id128` = the 128-byte ncclUniqueId made by one rank (dpipe_comm_unique_id) and handed to
 * the others out of band.  dpipe_send / dpipe_recv move `nbytes` untyped bytes to / from rank `peer` of the communicator, asynchronously on
 * `stream`; the sends and receives of one stage-boundary tuple are bracketed by dpipe_group_start / dpipe_group_end so RCCL issues them as one
 * operation (required when a rank both sends to and receives from the same peer).  Return codes >= 1000 are 1000 + ncclResult_t. */
int dpipe_comm_unique_id(void* id128);
int dpipe_comm_init(void** comm, int world, int rank, const void* id128);
int dpipe_comm_destroy(void* comm);
int dpipe_group_start(void);
int dpipe_group_end(void);
int dpipe_send(void* comm, const void* buf, long nbytes, int peer, void* stream);
int dpipe_recv(void* comm, void* buf, long nbytes, int peer, void* stream);

/* ---- C5 (ABI 10) progress marks INSIDE a captured graph -- the data-parallel all-reduce of a gradient bucket under the tail of the backward (the reference:
 * DeepSpeed's ReduceGrads behind utils/patches.py:153-156, train.py:843-844, which starts after the last backward).  A lane's micro-batch is ONE hipGraph, so the
 * point "the late layers' gradients are final" lies inside it.  A mark is a 4-byte device word:
 *   dpipe_mark_post(mark, gen, stream)   a one-thread KERNEL (a plain kernel node when `stream` is capturing): *mark = *gen with agent-scope release -- `gen` is a device
 *                                        word the host sets to a fresh value on the lane's stream before every replay, so a mark always names the replay that wrote it;
 *   dpipe_mark_wait(mark, value, err, timeout_ms, stream)   a one-wave kernel on the communication stream that sleeps until (int)(*mark - value) >= 0, i.e. until the
 *                                        replay numbered `value` has passed the mark; whatever is enqueued behind it on `stream` (lane sums, RCCL) then runs while the
 *                                        graph is still in the backward of the earlier layers.  After `timeout_ms` of waiting it stores 1 to *err (host-visible memory, may
 *                                        be NULL) and returns: a lost mark shows up as an error of the NEXT step instead of a hung queue.
 * Why not an event-record node: hipEventRecordWithFlags(hipEventRecordExternal) under capture does what is wanted with ROCm 7.2's runtime
 * (tools/probes/external_event_probe.hip) but returns hipErrorInvalidValue inside a PyTorch-ROCm 2.10 process (its own libamdhip64, HIP 7.0:
 * profiles/r6zh_mark_capture_probe.log), and torch.cuda.Event(external=True) is refused under capture on ROCm.  A kernel node has no such dependency. */
int dpipe_mark_post(void* mark, const void* gen, void* stream);
int dpipe_mark_wait(const void* mark, unsigned value, void* err, int timeout_ms, void* stream);

/* ---- K9 loss -------------------------------------------------------------------------------------------------
 * loss = (1/rows) * sum_r row_weight[r] * (1/cols) * sum_c elem(out[r,c] - target[r,c]) * mask[r,c]
 * Replaces F.mse_loss/huber_loss/smooth_l1_loss(reduction='none') * mask -> mean of models/base.py:418-436 (rows=1)
 * and the per-sample mean over (1,2,3) x SNR weight -> mean of models/sdxl.py:632-651 (rows=batch).
 * mask / row_weight / row_loss may be NULL.  workspace: dpipe_loss_workspace_floats(rows, cols) floats. */
int dpipe_loss_workspace_floats(long rows, long cols);
int dpipe_loss_fwd(const void* out, int dtype, const float* target, const float* mask, const float* row_weight,
                   long rows, long cols, int kind, float param, float* workspace, float* loss, float* row_loss,
                   void* stream);
/* grad_out[r,c] = grad_loss[0] * row_weight[r] * elem'(out - target) * mask / (rows * cols) ; autograd of the above. */
int dpipe_loss_bwd(const void* out, int dtype, const float* target, const float* mask, const float* row_weight,
                   const float* grad_loss, long rows, long cols, int kind, float param, void* grad_out, void* stream);

/* ---- K6 activations ------------------------------------------------------------------------------------------
 * nn.GELU(approximate='tanh') of models/wan/model.py:270-272, nn.SiLU of wan.py time_embedding, diffusers GEGLU. */
int dpipe_act_fwd(const void* x, void* y, long n, int dtype, int act, void* stream);
/* (ABI 7) adjoint of the nearest 2x up-sampling folded into a convolution (diffusers Upsample2D behind models/sdxl.py:846,865): dst [B, H, W, C] = the 2 x 2 block sums of
 * src [B, 2H, 2W, C], channels-last, fp32 accumulate; C a multiple of 8 (bf16) / 4 (fp32) */
int dpipe_upsample2x_adjoint(const void* src, void* dst, int B, int H, int W, int C, int dtype, void* stream);
int dpipe_act_bwd(const void* x, const void* gy, void* gx, long n, int dtype, int act, void* stream);
/* (ABI 8) Precise row linear: a few rows of fp32 values that become PER-CHANNEL ADDENDS -- the SDXL time embedding through diffusers' TimestepEmbedding and
 * ResnetBlock2D.time_emb_proj, `hidden_states + temb[:, :, None, None]` behind models/sdxl.py:797-865 -- keep fp32 accuracy through the bf16 MFMA GEMM: the
 * (activated) operand rows are split into bf16 hi / lo pairs, the GEMM (dpipe_gemm_ex, fp32 result) runs over 2R rows, the pair of result rows is added in fp32.
 *   rowsplit_fwd   : hl [2][R*K] bf16 <- act(x [R*K] fp32): hl[0] = bf16(v), hl[1] = bf16(v - hl[0])                    (n = R*K)
 *   rowsplit_bwd   : dx [n] fp32 = ds [n] bf16 * act'(x [n])
 *   rowcombine_fwd : y[r][c] = g[r][c] + g[R + r][c] + bias[c] + extra[c] (g fp32 [2R][N]; bias / extra bf16 [N] or NULL) -> out32 [R][N] fp32 and / or
 *                    out_hl [2][R][N] bf16 (hi / lo pair: the bias operand of dpipe_conv2d_fwd under DPIPE_CONV_BIAS_HILO); either may be NULL
 *   rowcombine_bwd : gy [2][R][N] bf16 <- gout [R][N] (fp32 or bf16, both halves equal); dbias[c] (+)= sum_r gout[r][c], dextra likewise (bf16 [N] or NULL) */
int dpipe_rowsplit_fwd(const float* x, void* hl, long n, int act, void* stream);
int dpipe_rowsplit_bwd(const void* ds, const float* x, float* dx, long n, int act, void* stream);
int dpipe_rowcombine_fwd(const float* g, const void* bias, const void* extra, float* out32, void* out_hl, int R, int N, void* stream);
int dpipe_rowcombine_bwd(const void* gout, int gout_dtype, void* gy, void* dbias, int dbias_accumulate, void* dextra, int dextra_accumulate, int R, int N, void* stream);
/* x: [rows, 2H] -> y[rows, H] = x[:, :H] * act(x[:, H:]) */
int dpipe_geglu_fwd(const void* x, void* y, long rows, long H, int dtype, int act, void* stream);
int dpipe_geglu_bwd(const void* x, const void* gy, void* gx, long rows, long H, int dtype, int act, void* stream);

/* ---- K5 gated residual ---------------------------------------------------------------------------------------
 * out[r,:] = x[r,:] + y[r,:] * gate[r / rows_per_gate, :]   ("x + y * e[2]" of models/wan/model.py:301,308;
 * apply_gate of models/hunyuan_image_modeling.py:222-237).  gate == NULL => plain residual add. */
int dpipe_gated_residual_fwd(const void* x, const void* y, const void* gate, void* out, long rows, long D,
                             long rows_per_gate, int dtype, int gate_dtype, void* stream);
int dpipe_gated_residual_slabs(long rows_per_gate);
/* gy = gout * gate ; dgate[b,:] = sum_rows gout * y.  workspace: batches * slabs * D floats. */
int dpipe_gated_residual_bwd(const void* gout, const void* y, const void* gate, void* gy, void* dgate,
                             float* workspace, long batches, long rows_per_gate, long D, int dtype, int gate_dtype,
                             void* stream);

/* ---- K7 timestep embedding: sinusoidal_embedding_1d (models/wan/model.py:15-25; diffusers Timesteps) ---------- */
int dpipe_sinusoidal_embed(const float* t, float* out, long n, int dim, float max_period, int sin_first,
                           float downscale_shift, float scale, void* stream);

/* ---- K8 flow-matching prep: x_t = (1-t) x1 + t x0, target = x0 - x1 (models/flux.py:368-372, wan.py:363-367) --- */
int dpipe_flow_match_prep(const float* x1, const float* x0, const float* t, float* xt, float* target, long batch,
                          long per_sample, void* stream);

/* ---- K10 gradient norm + clip (utils/patches.py:175-246) ------------------------------------------------------
 * Chunk table (device memory): chunk c = elements [chunk_off[c], chunk_off[c]+chunk_len[c]) of tensor
 * ptrs[chunk_tensor[c]].  sumsq: out_sumsq[0] = sum over all chunks of x^2 (fp32, deterministic two-stage).
 * clip_scale: every element *= min(1, max_norm / (sqrt(total_sumsq[0]) + 1e-6)); no host synchronisation. */
int dpipe_multi_sumsq(const void* const* ptrs, const int* chunk_tensor, const long* chunk_off, const int* chunk_len,
                      int nchunks, int dtype, float* partials, float* out_sumsq, void* stream);
int dpipe_multi_clip_scale(void* const* ptrs, const int* chunk_tensor, const long* chunk_off, const int* chunk_len,
                           int nchunks, int dtype, const float* total_sumsq, float max_norm, void* stream);

/* ---- step end: lane sum + clip + AdamW + zero in two passes (replaces the reference's separate clip_grad_norm_
 * utils/patches.py:175-246 and torch.optim.AdamW step train.py:672-678) ----------------------
 * Same chunk table; g_ptrs holds `lanes` gradient pointers per tensor, laid out [tensor][lane] (lanes = the engine's
 * concurrent micro-batch accumulators, 1..8); all tensors of one call share `dtype` (bf16 or fp32; p, m, v, g alike).
 * sumsq: out_sumsq[0] (+)= sum over elements of (sum over lanes g)^2, fp32, deterministic two-stage.
 * step:  g = min(1, max_norm / (sqrt(total_sumsq[0]) + 1e-6)) * sum_lanes g   (total_sumsq NULL or max_norm <= 0: no clip)
 *        p *= 1 - lr wd; m = lerp(m, g, 1 - beta1); v = beta2 v + (1 - beta2) g^2;
 *        p -= lr / bias_correction1 * m / (sqrt(v) / sqrt(bias_correction2) + eps)      (fp32 arithmetic, torch's fused AdamW)
 *        zero_grads != 0: every lane's gradient is overwritten with 0 in the same pass. */
int dpipe_adamw_sumsq(const void* const* g_ptrs, int lanes, const int* chunk_tensor, const long* chunk_off,
                      const int* chunk_len, int nchunks, int dtype, float* partials, float* out_sumsq, int accumulate,
                      void* stream);
int dpipe_adamw_step(void* const* p_ptrs, void* const* m_ptrs, void* const* v_ptrs, void* const* g_ptrs, int lanes,
                     const int* chunk_tensor, const long* chunk_off, const int* chunk_len, int nchunks, int dtype, float lr,
                     float beta1, float beta2, float eps, float weight_decay, float bias_correction1,
                     float bias_correction2, const float* total_sumsq, float max_norm, int zero_grads, void* stream);
/* The same step with Kahan (compensated) application of the update to low-precision parameters -- what the reference's optimizers do for bf16
 * parameters (optimizers/generic_optim.py:486-497, optimizers/automagic.py:309-320, optimizers/adamw_8bit.py): shift_ptrs[t] = the `shift` state
 * tensor of parameter t (parameter dtype): shift += update; old = p; p += shift; shift += old - p, every intermediate rounded to the dtype. */
int dpipe_adamw_step_kahan(void* const* p_ptrs, void* const* m_ptrs, void* const* v_ptrs, void* const* shift_ptrs, void* const* g_ptrs, int lanes,
                           const int* chunk_tensor, const long* chunk_off, const int* chunk_len, int nchunks, int dtype, float lr, float beta1,
                           float beta2, float eps, float weight_decay, float bias_correction1, float bias_correction2, const float* total_sumsq,
                           float max_norm, int zero_grads, void* stream);

/* 8-bit block-wise AdamW -- the reference's `adamw8bit` / `adamw8bitkahan` (train.py:673-686: bitsandbytes.optim.AdamW8bit, optimizers/adamw_8bit.py:6-124;
 * the library's optimizer_update_8bit_blockwise).  One parameter tensor per call: `state1` / `state2` = n uint8 codes of the two 256-entry maps `qmap1`
 * (signed dynamic) / `qmap2` (unsigned dynamic), `absmax1` / `absmax2` = one fp32 per block of 256 elements; `step` = this update's 1-based count (bias
 * corrections are computed here); `shift` != NULL selects the reference's Kahan variant (the update lands in `shift`, then p' = p + shift,
 * shift' = shift + (p - p') in the parameter dtype).  Gradient clipping is the caller's (`gnorm_scale` multiplies the gradient, 1.0 = none). */
int dpipe_adamw8bit_step(void* p, const void* g, void* state1, void* state2, float* absmax1, float* absmax2, const float* qmap1, const float* qmap2,
                         void* shift, long n, float lr, float beta1, float beta2, float eps, float weight_decay, int step, float gnorm_scale, int dtype,
                         void* stream);
/* The same update for MANY tensors in one launch (the step of a whole parameter group): device tables of per-tensor pointers (`shift_ptrs` NULL = no Kahan
 * buffers) and element counts `sizes`, and a chunk table (chunk c covers elements [chunk_off[c], chunk_off[c] + 2048) of tensor chunk_tensor[c]; offsets are
 * multiples of 2 048 = 8 quantisation blocks).  8 elements per thread with 16-byte accesses; HBM-bound: 10 B per parameter (14 with Kahan). */
int dpipe_adamw8bit_multi(void* const* p_ptrs, void* const* g_ptrs, void* const* state1_ptrs, void* const* state2_ptrs, void* const* absmax1_ptrs,
                          void* const* absmax2_ptrs, void* const* shift_ptrs, const long* sizes, const int* chunk_tensor, const long* chunk_off, int nchunks,
                          const float* qmap1, const float* qmap2, float lr, float beta1, float beta2, float eps, float weight_decay, int step, float gnorm_scale,
                          int dtype, void* stream);

/* ---- K2 RMSNorm (models/wan/model.py:70-86; per-head form models/hunyuan_image_modeling.py:98-103) ------------
 * y = cast(x * rsqrt(mean(x^2) + eps)) * w ; w may be NULL; rstd [rows] saved for backward (may be NULL). */
int dpipe_rmsnorm_fwd(const void* x, const void* w, void* y, float* rstd, long rows, int cols, float eps, int dtype,
                      int wdtype, void* stream);
int dpipe_norm_slabs(long rows_per_group);
/* workspace: dpipe_norm_slabs(rows) * cols floats, needed only when dw != NULL.  accumulate_params != 0: dw += (the
 * gradient-accumulation step of the micro-batch loop fused into the reduction; dw then is the persistent .grad). */
int dpipe_rmsnorm_bwd(const void* x, const void* w, const void* gy, const float* rstd, void* gx, void* dw,
                      float* workspace, long rows, int cols, int dtype, int wdtype, int accumulate_params, void* stream);
/* K2 + K3 fused (RMSNorm -> RoPE -> Q/K write-out in one pass; models/wan/model.py:124-125,139-140, hunyuan_image_modeling.py:181-190):
 * y = rope(type_as(x * rstd) * w).  A row is one normalisation group: the whole token (groups_per_token = 1, cols = H * head_dim: Wan's norm_q / norm_k) or
 * one head (groups_per_token = H, cols = head_dim: Flux / HunyuanVideo); rows = B * S * groups_per_token.  The rotation (interleaved pairs, cos_t / sin_t:
 * fp32 [>= token_offset + S, head_dim / 2]) applies to tokens s < rope_tokens (-1 = all).  x may be a strided view of a fused QKV projection: token pitch
 * x_token_stride elements, the token's groups `cols` apart; y, gy, gx are dense [rows, cols].  workspace / dw / accumulate_params as dpipe_rmsnorm_bwd. */
int dpipe_rmsnorm_rope_fwd(const void* x, const void* w, const float* cos_t, const float* sin_t, void* y, float* rstd, long rows, int cols, int head_dim, long S,
                           int groups_per_token, long token_offset, long rope_tokens, long x_token_stride, float eps, int dtype, int wdtype, void* stream);
int dpipe_rmsnorm_rope_bwd(const void* x, const void* w, const void* gy, const float* rstd, const float* cos_t, const float* sin_t, void* gx, void* dw, float* workspace,
                           long rows, int cols, int head_dim, long S, int groups_per_token, long token_offset, long rope_tokens, long x_token_stride, int dtype,
                           int wdtype, int accumulate_params, void* stream);
/* out[c] (+)= sum_r x[r, c], x: [rows, cols] with row stride ld -- the bias gradient of nn.Linear (column sums of dy).
 * dtype: x; out_dtype: out (bf16 x -> bf16 or fp32 out; fp32 x -> fp32 out).  workspace as above. */
int dpipe_colsum(const void* x, long rows, int cols, long ld, void* out, float* workspace, int dtype, int out_dtype,
                 int accumulate, void* stream);

/* ---- K5 LayerNorm (+affine) + AdaLN modulate (models/wan/model.py:89-99,295-309,332-343) ----------------------
 * n = (x - mean) * rstd [* gamma + beta] ; y = n * (1 + scale[r / rows_per_mod]) + shift[r / rows_per_mod].
 * gamma/beta/scale/shift may each be NULL.  wdtype: dtype of gamma/beta; mdtype: dtype of scale/shift. */
int dpipe_lnmod_fwd(const void* x, const void* gamma, const void* beta, const void* scale, const void* shift,
                    void* y, float* mean, float* rstd, long rows, int cols, long rows_per_mod, float eps, int dtype,
                    int wdtype, int mdtype, void* stream);
int dpipe_lnmod_workspace_floats(long rows, int cols, long rows_per_mod);
int dpipe_lnmod_bwd(const void* x, const void* gy, const void* gamma, const void* beta, const void* scale,
                    const float* mean, const float* rstd, void* gx, void* dgamma, void* dbeta, void* dscale,
                    void* dshift, float* workspace, long rows, int cols, long rows_per_mod, int dtype, int wdtype,
                    int mdtype, int accumulate_params, const void* gx_add, void* stream);
/* accumulate_params: dgamma / dbeta +=.  gx_add (may be NULL, [rows, cols] like gx): gx = d(norm)/dx + gx_add -- the gradient that
 * reached x through the residual branch around the norm, folded into this pass. */

/* ---- 2-D convolution as implicit GEMM (NHWC bf16): the nn.Conv2d of diffusers' ResnetBlock2D / Downsample2D / Upsample2D
 * behind the UNet call sites models/sdxl.py:797-865 (the reference reaches MIOpen / cuDNN through torch.nn.functional.conv2d).
 * x: [B, H, W, Cin] with pixel pitch ldx elements; w: [Cout, kh, kw, Cin] contiguous (the channels_last storage of a
 * [Cout, Cin, kh, kw] parameter); y / dy: [B, Ho, Wo, Cout]; stride, upsample in {1, 2}; `upsample` = nearest 2x up-sampling of x
 * fused into the gather (Upsample2D), H / W are then the size BEFORE up-sampling; Ho = ((H * upsample) + 2 pad - kh) / stride + 1.
 * Zero padding never materialises (out-of-image taps read through the buffer bounds check).  Same LDS-DMA MFMA tile machinery,
 * split-K workspace (`splitk_ws`, see dpipe_gemm_ex) and `tile_hint` as dpipe_gemm_ex.
 *   fwd  : y = act(conv(x, w) + bias) (+ residual [B, Ho, Wo, Cout] with pitch ldr);            needs Cin % 64 == 0
 *   dgrad: dx [B, H, W, Cin] = conv_transpose(dy, w)  (no fused up-sampling: H, W = the conv's input size);  needs Cout % 64 == 0, Cin % 8 == 0
 *   wgrad: dw [Cout, kh, kw, Cin] (+)= dy^T * gathered x, dbias [Cout] (+)= column sums of dy (NULL = skip); needs Cin % 8 == 0, Cout % 8 == 0
 * `flags` (fwd / dgrad) / `out_f32` (wgrad): the result (and the forward's residual) is fp32 and, with DPIPE_CONV_ACCUMULATE, added to what the
 * output already holds.  That is what the exact-parity mode builds an fp32 convolution from: x = x_hi + x_lo, w = w_hi + w_lo (bf16 pairs),
 * conv(x, w) = conv(x_hi, w_hi) + conv(x_lo, w_hi) + conv(x_hi, w_lo) accumulated in fp32 (the dropped lo x lo term and the rounding of the lo
 * parts are <= 2^-16 relative per product) -- the same MFMA kernels, no library convolution. */
#define DPIPE_CONV_OUT_F32 1
#define DPIPE_CONV_ACCUMULATE 2
#define DPIPE_CONV_BIAS_PER_SAMPLE 4   /* dpipe_conv2d_fwd: `bias` is [B][Cout] (bf16, row pitch Cout, Cout % 4 == 0): every output pixel of sample b adds row b */
#define DPIPE_CONV_BIAS_HILO 8         /* (ABI 8) dpipe_conv2d_fwd: `bias` holds TWO such sets back to back, [2][Cout] or (with PER_SAMPLE) [2][B][Cout]: a bf16 hi / lo pair of an
                                        * fp32 addend (dpipe_rowcombine_fwd), both added to the fp32 accumulator before the output is rounded; Cout % 4 == 0 */
int dpipe_conv2d_fwd(const void* x, long ldx, const void* w, const void* bias, const void* residual, long ldr, void* y, long ldy,
                     int B, int H, int W, int Cin, int Cout, int kh, int kw, int stride, int pad, int upsample, int act, int flags,
                     void* splitk_ws, long splitk_ws_bytes, int tile_hint, void* stream);
int dpipe_conv2d_dgrad(const void* dy, long lddy, const void* w, void* dx, long lddx,
                       int B, int H, int W, int Cin, int Cout, int kh, int kw, int stride, int pad, int flags,
                       void* splitk_ws, long splitk_ws_bytes, int tile_hint, void* stream);
int dpipe_conv2d_wgrad(const void* dy, long lddy, const void* x, long ldx, void* dw, void* dbias,
                       int B, int H, int W, int Cin, int Cout, int kh, int kw, int stride, int pad, int upsample,
                       int accumulate, int bias_accumulate, int out_f32, void* splitk_ws, long splitk_ws_bytes, int tile_hint, void* stream);

/* ---- GroupNorm (+ fused SiLU) on NHWC activations: the same nn.GroupNorm(G, C) call sites as below for the channels-last UNet.
 * x, y, dy, dx: [N, HW, C] contiguous, C a multiple of the 16-byte vector and <= 8192; other arguments as dpipe_groupnorm_fwd / _bwd;
 * workspace: dpipe_groupnorm_nhwc_workspace_floats() floats. */
long dpipe_groupnorm_nhwc_workspace_floats(long N, int C, long HW, int G);
int dpipe_groupnorm_nhwc_fwd(const void* x, const void* gamma, const void* beta, void* y, float* mean, float* rstd, float* workspace,
                             long N, int C, long HW, int G, float eps, int act, int dtype, int wdtype, void* stream);
int dpipe_groupnorm_nhwc_bwd(const void* x, const void* dy, const void* gamma, const void* beta, const float* mean, const float* rstd,
                             void* dx, void* dgamma, void* dbeta, float* workspace, long N, int C, long HW, int G, int act, int dtype,
                             int wdtype, int accumulate_params, const void* dx_add, void* stream);   /* dx_add: as dpipe_lnmod_bwd's gx_add */

/* ---- GroupNorm (+ fused SiLU) on NCHW activations: nn.GroupNorm(G, C) of the diffusers ResnetBlock2D / Transformer2DModel
 * behind models/sdxl.py:797-865 (and the SiLU that follows it in the resnets).  x, y, dy, dx: [N, C, HW] contiguous, HW a
 * multiple of the 16-byte vector; gamma / beta [C] (NULL = no affine); mean / rstd [N * G] fp32 saved for backward;
 * act: DPIPE_ACT_NONE or DPIPE_ACT_SILU; workspace: dpipe_groupnorm_workspace_floats() floats.
 * bwd: dgamma / dbeta may be NULL; accumulate_params != 0 adds into them (fused gradient accumulation). */
long dpipe_groupnorm_workspace_floats(long N, int C, long HW, int G);
int dpipe_groupnorm_fwd(const void* x, const void* gamma, const void* beta, void* y, float* mean, float* rstd,
                        float* workspace, long N, int C, long HW, int G, float eps, int act, int dtype, int wdtype,
                        void* stream);
int dpipe_groupnorm_bwd(const void* x, const void* dy, const void* gamma, const void* beta, const float* mean,
                        const float* rstd, void* dx, void* dgamma, void* dbeta, float* workspace, long N, int C, long HW,
                        int G, int act, int dtype, int wdtype, int accumulate_params, const void* dx_add, void* stream);   /* dx_add: as gx_add above */

/* ---- K3 RoPE (models/wan/model.py:40-67 rope_apply; Flux/HunyuanVideo cos/sin tables) -------------------------
 * x, y: [B, S, H, D] contiguous; cos/sin: [S, D/2] fp32.  interleaved=1 rotates pairs (2i, 2i+1) (view_as_complex),
 * interleaved=0 rotates (i, i + D/2).  conj=1 applies the inverse rotation (the backward pass). */
int dpipe_rope(const void* x, const float* cos_t, const float* sin_t, void* y, long B, long S, long H, int D,
               int interleaved, int conj, int dtype, void* stream);

/* ---- row softmax (only the fp32 unfused attention parity path uses it) ---------------------------------------- */
/* causal_rows > 0: rows form matrices of that many rows; row r only sees columns <= r (masked outputs are 0). */
int dpipe_softmax_fwd(const void* x, void* y, long rows, int cols, long ld, float scale, int causal_rows, int dtype,
                      void* stream);
int dpipe_softmax_bwd(const void* y, const void* gy, void* gx, long rows, int cols, long ld, float scale, int dtype,
                      void* stream);

/* ---- batched 2-D transpose: in [batch][R][C] -> out [batch][C][R] ---------------------------------------------- */
int dpipe_transpose(const void* in, void* out, int R, int C, long ldi, long ldo, long batch_stride_in,
                    long batch_stride_out, int batch, int dtype, void* stream);

/* ---- K1/K6 GEMM on MFMA (nn.Linear fwd / dgrad / wgrad: models/wan/model.py:120-122,138-142,270-272) ----------
 * C[z] = act(alpha * op(A[z]) . op(B[z]) + bias) (+ C[z] if accumulate); z = (zo, zi), zi < batch_inner.
 * transA=0: A is [M][K] row-major (lda); transA=1: A is [K][M].  transB=1: B is [N][K]; transB=0: B is [K][N].
 * bias: [N] in the operand dtype or NULL.  out_f32: write fp32 C from bf16 operands.  tile_hint: 0 auto, 64, 128. */
int dpipe_gemm(int dtype, int transA, int transB, int M, int N, int K, const void* A, long lda, const void* B,
               long ldb, void* C, long ldc, int batch_outer, int batch_inner, long strideA_outer, long strideA_inner,
               long strideB_outer, long strideB_inner, long strideC_outer, long strideC_inner, const void* bias,
               int act, float alpha, int accumulate, int out_f32, int tile_hint, void* stream);
/* Same contract plus the split-K workspace of the pipelined bf16 kernel (csrc/gemm_pipe.hip): `splitk_ws` =
 * [4 KiB ticket counters][64 KiB fp32 slabs ...], zero-filled ONCE by the host and then private to one stream (the
 * kernel re-arms the counters itself); >= 4 KiB + 256 slabs makes every split decision available; NULL = never split.
 * tile_hint: 0 = auto, 64 / 128 = generic register-staged kernel with that tile, 1000 + S = pipelined kernel with S
 * K-slices forced (S = 0: its own choice), 2000 + S / 3000 + S = pipelined kernel with the 64 x 64 / 128 x 128 tile
 * forced (4000 / 5000 / 6000 + S: 128 x 128 with a 2-deep ring, 256 x 128, 64 x 64 with a 3-deep ring); forced forms
 * fail when the problem is not eligible: bf16, 16-byte aligned operands, leading dims multiples of 8, K % 64 == 0 unless
 * both operands are K-major.
 * Fused epilogue extras: `residual` (C's dtype, row pitch ldr, C's batch strides; NULL = none): C = act(...) + residual
 * -- the "x + attn(x)" / "x + ff(x)" adds of a transformer block folded into the output projection.  `colsum` (bf16 [M]
 * per batch; NULL = none): colsum[m] (+)= sum_k A[k][m] for transA = 1 -- the bias gradient of nn.Linear computed inside
 * its wgrad GEMM (dW = dy^T x, db = column sums of dy); returns DPIPE_ERR_UNSUPPORTED (-2) when the problem cannot take
 * the pipelined kernel, the caller then uses dpipe_colsum. */
int dpipe_gemm_ex(int dtype, int transA, int transB, int M, int N, int K, const void* A, long lda, const void* B,
                  long ldb, void* C, long ldc, int batch_outer, int batch_inner, long strideA_outer,
                  long strideA_inner, long strideB_outer, long strideB_inner, long strideC_outer, long strideC_inner,
                  const void* bias, int act, float alpha, int accumulate, int out_f32, int tile_hint, void* splitk_ws,
                  long splitk_ws_bytes, const void* residual, long ldr, void* colsum, int colsum_accumulate, void* stream);
/* Grouped launch: n INDEPENDENT plain GEMMs (batch 1; no problem reads what another writes) as few kernel launches as possible -- the dgrad
 * (dx = dy W) and wgrad (dW += dy^T x, db += column sums of dy) of one nn.Linear backward (models/wan/model.py:120-122 under autograd), or
 * the same-shaped linears of two sibling branches.  Every problem is computed exactly as dpipe_gemm_ex(tile_hint = 0) would compute it alone
 * (same tile, same split-K: bit-identical results); bf16 problems of one tile geometry leave as ONE launch of the LDS-DMA kernel whose
 * workgroups are divided between them (csrc/gemm_pipe_kernel.h: gemm_pipe_group_kernel), the rest as single launches.  `splitk_ws` as
 * dpipe_gemm_ex (the problems of a group get disjoint shares).  Returns DPIPE_ERR_UNSUPPORTED (-2) with NOTHING launched when a problem that
 * asks for `colsum` cannot take the pipelined kernel (the caller then uses dpipe_colsum, as with dpipe_gemm_ex).  *launches_out (may be NULL)
 * = kernel launches issued.  n <= 16. */
typedef struct dpipe_gemm_desc {
    int dtype, transA, transB, M, N, K;
    const void* A; long lda;
    const void* B; long ldb;
    void* C; long ldc;
    const void* bias; int act; float alpha; int accumulate; int out_f32;
    const void* residual; long ldr;
    void* colsum; int colsum_accumulate;
} dpipe_gemm_desc;
int dpipe_gemm_group(const dpipe_gemm_desc* descs, int n, void* splitk_ws, long splitk_ws_bytes, int* launches_out, void* stream);
/* (ABI 8) The plan of such a call WITHOUT launching anything (host code only: no device, no dereference of the operand pointers -- they are looked at for alignment):
 * tiles_out[i] = the tile code problem i would run on (64 = 64^2 4-deep LDS-DMA ring, 128 = 128^2 3-deep, 129 = 128^2 2-deep, 132 = 128^2 register-staged, 257 / 258 =
 * 256^2), splitk_out[i] (optional) its K-slice count, *launches_out the kernel launches the call would issue -- under the process's current dpipe_set_option values.
 * What the dispatcher tests hold the tile policy with (tests/test_gemm_plan_cpu.py); n = 1 is the plan of a plain dpipe_gemm_ex call. */
int dpipe_gemm_group_plan(const dpipe_gemm_desc* descs, int n, long splitk_ws_bytes, int* tiles_out, int* splitk_out, int* launches_out);
/* Test probe: runs ds_read_b64_tr_b16 over a 256-element i16 LDS image so the GPU tests can pin the lane mapping
 * the transposed-operand paths rely on. */
int dpipe_tr16_probe(const void* in256_i16, void* out256_i16, void* stream);

/* ---- K4 scaled-dot-product attention, flash style (models/wan/attention.py:91-122 flash_attn_varlen_func;
 *      diffusers attention processors behind models/sdxl.py:797-865; joint attention of utils/patches.py:325-340)
 * q: [B, Sq, H, D], k/v: [B, Sk, H, D], o: [B, Sq, H, D]; bf16; strides in elements (innermost D contiguous).
 * kv_len: optional int32 [B] valid key counts (NULL => Sk); lse: [B, H, Sq] fp32 (log-sum-exp, natural log).
 * D in {64, 128}.  causal != 0 masks keys with index > query index (CLIP text encoders inside SDXL stage 0,
 * models/sdxl.py:738-784). */
int dpipe_attn_fwd(const void* q, const void* k, const void* v, void* o, float* lse, const int* kv_len, int B, int H,
                   int Sq, int Sk, int D, long q_sb, long q_ss, long q_sh, long k_sb, long k_ss, long k_sh, long v_sb,
                   long v_ss, long v_sh, long o_sb, long o_ss, long o_sh, float scale, int causal, float* o_f32, void* stream);
/* o_f32 (optional, NULL = none): O once more in fp32, [B, Sq, H, D] contiguous, before its rounding to bf16 -- handed to dpipe_attn_bwd, whose
 * delta = rowsum(dO . O) then does not carry the bf16 rounding of O into dS = P (dP - delta) (it matters when the value rows share a large
 * common component: dP and delta are then both ~ dO . c and their difference is the signal). */
/* delta: [B, H, Sq] fp32 workspace.  dq/dk/dv have the layouts (and strides) of q/k/v.
 * dkv_partial: fp32 workspace of dpipe_attn_bwd_partial_floats() elements (0 = not needed) -- when few key blocks x
 * heads cannot fill the chip (cross attention to 77 text tokens) the dK/dV kernel splits the queries over workgroups
 * and a reduce kernel sums the slices in order; NULL keeps the unsplit single-kernel form. */
long dpipe_attn_bwd_partial_floats(int B, int H, int Sq, int Sk, int D);
int dpipe_attn_bwd(const void* q, const void* k, const void* v, const void* o, const void* dout, const float* lse,
                   float* delta, void* dq, void* dk, void* dv, const int* kv_len, int B, int H, int Sq, int Sk, int D,
                   long q_sb, long q_ss, long q_sh, long k_sb, long k_ss, long k_sh, long v_sb, long v_ss, long v_sh,
                   long o_sb, long o_ss, long o_sh, long do_sb, long do_ss, long do_sh, long dq_sb, long dq_ss,
                   long dq_sh, long dk_sb, long dk_ss, long dk_sh, long dv_sb, long dv_ss, long dv_sh, float scale,
                   int causal, float* dkv_partial, long dkv_partial_floats, const float* o_f32, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DPIPE_HIP_H */
