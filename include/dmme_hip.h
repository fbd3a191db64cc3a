/*
 * dmme_hip.h -- C ABI of libdmme_hip.so: the MI355X (gfx950) denoiser hot path.
 *
 * The reference (urw7rs/diffusion-models-made-easy v0.5.2) is pure Python and has no
 * FFI of its own; every entry point below replaces a *Python* interface of the hot
 * path and cites it (paths relative to the reference root).  A maintainer binds these
 * with ctypes.
 *
 * Conventions
 *   - plain pointers + sizes only; no torch / C++ types cross the boundary;
 *   - every device pointer is caller-owned (the plan never owns tensor memory; it
 *     owns host metadata plus one small device table used by the weight re-packer);
 *   - every call is asynchronous on the caller's HIP stream (`stream` is a
 *     hipStream_t passed as void*); no hidden synchronisation or allocation in the
 *     launch path, so calls can be captured into a hipGraph;
 *   - return value: 0 (DMME_OK) or a negative dmme_status; the message of the last
 *     failure on the calling thread is available from dmme_last_error();
 *   - no exceptions cross the ABI.
 *
 * Tensor layouts at the boundary
 *   - images x / y / eps / z: NCHW fp32 (what the reference's UNet.forward takes,
 *     src/dmme/models/ddpm.py:281-316);
 *   - timesteps: int64 device array of length 1 or B (reference passes `all_t[t]`
 *     of shape (1,) when sampling, (B,) when training; diffusion_models/ddpm.py:65-77,124-131);
 *   - parameters: one flat fp32 buffer in reference state_dict order and reference
 *     layouts (conv OIHW, linear (out,in)); dmme_unet_pack_params() converts it to
 *     the library's packed layout ([Cout][kh][kw][Cin], compute dtype).
 */
#ifndef DMME_HIP_H
#define DMME_HIP_H

#include <stddef.h>
#include <stdint.h>

#if defined(DMME_BUILD)
#define DMME_API __attribute__((visibility("default")))
#else
#define DMME_API
#endif

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
    DMME_OK = 0,
    DMME_ERR_INVALID = -1,     /* bad argument / shape the reference would also reject */
    DMME_ERR_UNSUPPORTED = -2, /* valid in the reference, not supported by this build */
    DMME_ERR_HIP = -3,         /* a HIP runtime call failed */
    DMME_ERR_NOMEM = -4
} dmme_status;

/* DMME_BF16X3: the accurate mode.  Tensors and weights stay fp32 (same layouts, workspace and packed sizes as DMME_F32); every
 * convolution product runs as three bf16 MFMA passes over hi/lo splits of its fp32 operands (a_hi*b_hi + a_hi*b_lo + a_lo*b_hi,
 * fp32 accumulation), ~2^-16 per product instead of 2^-9: within 1e-3 of the fp32 reference at the bf16 matrix rate / 3.
 * Accepted wherever a dtype is (plans and the single-op entry points); buffers are the fp32 ones. */
/* DMME_F16R32: the reduced-precision mode that stays within 1e-3 of the fp32 reference (precision="fp16r32", inference).  A
 * DMME_F16 plan whose FULL-RESOLUTION level - where a rounding error reaches the output undamped: the first and last ResBlocks,
 * the skip tensors between them, input and output conv (models/ddpm.py:293-295, 308-315) - keeps its tensors in fp32 and runs
 * every product as three fp16 MFMA passes over hi / lo halves of both operands (csrc/conv_pipe.hip: conv3x3_ws2_kernel<.., SPLIT>);
 * every level below it is plain DMME_F16.  Workspace and packed sizes differ from DMME_F16's (query them). */
typedef enum { DMME_F32 = 0, DMME_BF16 = 1, DMME_BF16X3 = 2, DMME_F16 = 3, DMME_F16R32 = 4 } dmme_dtype;

/* Which of the reference's two UNets the plan builds. */
typedef enum {
    DMME_ARCH_DDPM = 0,  /* dmme.models.ddpm.UNet (src/dmme/models/ddpm.py:176-316): eps only */
    DMME_ARCH_IDDPM = 1  /* dmme.models.iddpm.UNet (src/dmme/models/iddpm.py:125-265): scale-shift ResBlocks,
                            multi-head attention, 2*in_channels outputs (eps, v) */
} dmme_unet_arch;

/* Constructor arguments of the reference UNet (src/dmme/models/ddpm.py:190-200, models/iddpm.py:139-149). */
typedef struct {
    int in_channels;
    int pos_dim;
    int emb_dim;
    int num_groups;
    float dropout;
    int num_depths;
    int channels_per_depth[8];
    int num_blocks;
    int num_attention_depths;
    int attention_depths[8];
    int arch;      /* dmme_unet_arch */
    int num_heads; /* DMME_ARCH_IDDPM: heads of MultiHeadAttention (the reference hard-codes 4, models/iddpm.py:82);
                      ignored (1) for DMME_ARCH_DDPM */
} dmme_unet_cfg;

typedef struct dmme_plan dmme_plan;

DMME_API const char* dmme_last_error(void);
DMME_API int dmme_version(void);
/* number of HIP devices visible to the library (0 => the library loaded but cannot compute) */
DMME_API int dmme_device_count(void);

/* ---- UNet plan: replaces UNet.__init__ (models/ddpm.py:190-279) --------------------
 * Builds the layer graph (including the reference's always-false `if` at :242), the
 * parameter table, the packed-weight layout and the activation workspace layout for a
 * fixed (B, H, W, dtype).  Host-only; `device` selects which GPU owns the small
 * re-pack table (-1: do not touch any device, for CPU-side inspection of the table). */
DMME_API int dmme_unet_plan_create(const dmme_unet_cfg* cfg, int B, int H, int W, int dtype, int device, dmme_plan** out);
DMME_API void dmme_unet_plan_destroy(dmme_plan* plan);

/* Parameter table in reference state_dict order (305 entries for the default config;
 * SURVEY 8b).  `ref_offset` is the element offset inside the flat fp32 buffer. */
DMME_API int dmme_unet_plan_num_params(const dmme_plan* plan);
DMME_API int dmme_unet_plan_param_info(const dmme_plan* plan, int index, char* name, int name_cap, int* ndim,
                              int64_t shape[4], int64_t* ref_offset, int* is_buffer);
DMME_API int64_t dmme_unet_plan_ref_numel(const dmme_plan* plan);
DMME_API int64_t dmme_unet_plan_packed_bytes(const dmme_plan* plan);
DMME_API int64_t dmme_unet_plan_workspace_bytes(const dmme_plan* plan);
/* floats in a full set of Dropout2d multipliers: sum over ResBlocks of B*Cout, laid out
 * block after block (graph order: down, middle, up), each [B][Cout]. */
DMME_API int64_t dmme_unet_plan_dropmask_numel(const dmme_plan* plan);
/* channels of the network output: in_channels (DDPM) or 2*in_channels (IDDPM: eps and v stacked) */
DMME_API int dmme_unet_plan_out_channels(const dmme_plan* plan);
/* number of kernel launches one forward issues (for launch-overhead accounting) */
DMME_API int dmme_unet_plan_num_launches(const dmme_plan* plan);

/* fp32 reference-layout flat buffer -> packed buffer (replaces nothing in the
 * reference; it is the load_state_dict side of the boundary). */
DMME_API int dmme_unet_pack_params(const dmme_plan* plan, const float* ref_flat, void* packed, void* stream);

/* ---- UNet forward: replaces UNet.forward (models/ddpm.py:281-316) -------------------
 * y = eps_theta(x, t).  x: (B, C, H, W), y: (B, dmme_unet_plan_out_channels(), H, W), fp32 NCHW.
 * t: int64[t_len], t_len in {1, B}.  DMME_ARCH_IDDPM: iddpm.UNet.forward (models/iddpm.py:232-265).
 * drop_masks: NULL for eval(); else the Dropout2d multipliers (0 or 1/(1-p)) laid out
 * as dmme_unet_plan_dropmask_numel() describes (nn.Dropout2d, models/ddpm.py:29). */
DMME_API int dmme_unet_forward(const dmme_plan* plan, const void* packed, const float* x, const int64_t* t, int t_len,
                      float* y, void* workspace, const float* drop_masks, void* stream);
/* The same forward where no dmme_unet_backward will follow (the reference's sampling loops run under torch.no_grad(),
 * diffusion_models/ddpm.py:113-133): tensors that only the backward pass reads - the context of an attention block whose proj
 * conv runs inside the attention launch (models/ddpm.py:66-75), raw conv outputs of the level engine that only their norm's
 * pre-activated copy stands for - are not written.  Same arguments, same result y.  dmme_unet_backward on that workspace fails
 * with DMME_ERR_INVALID until a dmme_unet_forward has filled it again.
 * dmme_chain_step and dmme_unet_forward_profiled run this form. */
DMME_API int dmme_unet_forward_nograd(const dmme_plan* plan, const void* packed, const float* x, const int64_t* t, int t_len,
                      float* y, void* workspace, const float* drop_masks, void* stream);

/* ---- per-op accounting + event-bracketed profiling (bench.py's roofline leg) --------
 * op_info: kernel label (the kernel symbol the op launches, e.g.
 * "conv_mfma_kernel<bf16,9,128,128>"), its algorithmic FLOPs and algorithmic HBM bytes
 * (every operand read once, result written once) for the plan's (B,H,W).
 * forward_profiled: same launches as dmme_unet_forward, but each op is bracketed by HIP
 * events on `stream`; synchronises, then writes one elapsed time per op (ms). */
DMME_API int dmme_unet_plan_num_ops(const dmme_plan* plan);
/* Level-engine launches of this plan (csrc/lvl_engine.hip: one persistent launch per stretch of layers on a 4x4 / 8x8 map, replacing
 * the per-layer launches of ResBlock / Attention there, models/ddpm.py:118-133, 38-75) as text: "runs=N [map=4x4 plan_ops=a-b
 * engine_ops=.. groups=.. per_iteration=.. workgroups=.. epoch=.. err=..] ...".  err != 0: a bounded in-kernel wait timed out (results of
 * that launch are invalid).  Synchronises with the device (reads the control words). */
DMME_API int dmme_unet_plan_level_info(const dmme_plan* plan, char* buf, int cap);
/* Status of the level engine's bounded hand-off waits.  The engine's workgroups exchange tensors through spin-waits, so all of a
 * launch must be resident at once (the plan sizes its grids by what the device holds and keeps per-op launches otherwise); a wait
 * that still times out - compute units held by another stream, process or CU mask - lets the launch drain with invalid numbers and
 * raises a host-visible status word.  dmme_unet_forward / _forward_profiled / _backward / dmme_chain_step look at that word on
 * entry and fail with DMME_ERR_HIP (clearing it); this call is for hosts that REPLAY a captured graph of those launches (no entry
 * point runs then): call it after synchronising with the stream, before using the results.  DMME_OK: every engine launch of this
 * plan completed since the last check.  (Own invariant of the MI355X path; the reference's loop it guards:
 * src/dmme/diffusion_models/ddpm.py:113-133.) */
DMME_API int dmme_unet_plan_check(const dmme_plan* plan);
DMME_API int dmme_unet_plan_op_info(const dmme_plan* plan, int index, char* label, int label_cap, double* flops,
                           double* bytes);
DMME_API int dmme_unet_forward_profiled(const dmme_plan* plan, const void* packed, const float* x, const int64_t* t,
                               int t_len, float* y, void* workspace, const float* drop_masks, void* stream,
                               float* op_ms);

/* ---- training: replaces loss.backward() through the UNet + the optimiser side of the step ----
 * dmme_unet_backward: given d_y = dL/d(eps) (NCHW fp32) and the workspace left by the matching
 * dmme_unet_forward call (same x, t, drop_masks), ACCUMULATES dL/d(parameters) into grad_flat
 * (fp32, reference state_dict order and layouts, like the flat parameter buffer).
 * packed_bwd holds transposed, tap-flipped conv weights (dmme_unet_pack_params_bwd) for the
 * data-gradient convolutions. Replaces torch autograd of models/ddpm.py:281-316.
 * d_x (nullable): when given, receives dL/dx (NCHW fp32, shape of x) - the gradient autograd hands back for an input that
 * requires grad (guidance-style callers); NULL skips that convolution. */
DMME_API int64_t dmme_unet_plan_packed_bwd_bytes(const dmme_plan* plan);
DMME_API int64_t dmme_unet_plan_bwd_workspace_bytes(const dmme_plan* plan);
DMME_API int dmme_unet_pack_params_bwd(const dmme_plan* plan, const float* ref_flat, void* packed_bwd, void* stream);
DMME_API int dmme_unet_backward(const dmme_plan* plan, const void* packed, const void* packed_bwd, const float* x,
                       const int64_t* t, int t_len, const float* d_y, void* workspace, void* bwd_workspace,
                       const float* drop_masks, float* grad_flat, float* d_x, void* stream);
/* Same backward in GRADIENT BUCKETS, so a data-parallel caller can exchange a bucket while the rest of backward is still being
 * computed (north star: "RCCL all-reduce of UNet grads over xGMI overlapped with backward"; the reference gets this from
 * Lightning's DDP wrapper around `loss.backward()`, 25 MB buckets).  The op list is cut at ResBlock boundaries into stretches of
 * at most ~1/6 of the parameters, in the order backward finishes them: output conv + the last up blocks first, ..., the first down
 * blocks + input conv + time MLP last (<= 15 % of the bytes: the only exchange nothing can hide).  `ready(user, bucket, offset,
 * numel)` is called on the calling thread - once per contiguous range of a bucket (a bucket may cover two: middle_layers and
 * output_conv sit behind up_layers in the flat buffer) - as soon as every launch that writes grad_flat[offset, offset + numel) has
 * been ENQUEUED on `stream` (record an event there; nothing has necessarily executed yet).  Same results as dmme_unet_backward.
 * dmme_unet_plan_grad_buckets: the (offset, numel, bucket) triples in hand-over order, returns their count (may exceed `cap`);
 * 1 = the configuration has no clean cut, `ready` is then never called. */
typedef void (*dmme_bucket_fn)(void* user, int bucket, int64_t offset, int64_t numel);
DMME_API int dmme_unet_backward_buckets(const dmme_plan* plan, const void* packed, const void* packed_bwd, const float* x,
                               const int64_t* t, int t_len, const float* d_y, void* workspace, void* bwd_workspace,
                               const float* drop_masks, float* grad_flat, float* d_x, void* stream, dmme_bucket_fn ready, void* user);
DMME_API int dmme_unet_plan_grad_buckets(const dmme_plan* plan, int64_t* offsets, int64_t* numels, int* bucket_of, int cap);
/* test / diagnostic: the kernels a backward of this plan launches, as space-separated key=value pairs
 * ("wgrad_group3x3_jobs=..", "colsum_group_jobs=..", "dgrad[conv3x3_ws2_kernel<11>]=..") */
DMME_API int dmme_unet_plan_bwd_summary(const dmme_plan* plan, char* buf, int cap);
/* diagnostic (tools/l2_stream.py): `blocks` workgroups of 256 threads each stream the same `bytes` (a multiple of 64 KiB) `iters`
 * times with `depth` (1, 2, 4, 8, 16) 16-byte loads in flight per thread; mode 0 loads into registers, mode 1 uses the global -> LDS DMA,
 * mode m >= 2 the DMA in the convolutions' access shape (a wave instruction gathers eight 128-byte rows m 16-byte vectors apart, e.g.
 * 288 = a 3x3 filter row of 256 input channels).  sink: `blocks` uint32 of scratch.  Measures what a CU can draw from L2. */
/* diagnostic (tools/mfma_valu.py): do the MFMAs of one wave and the VALU work of another wave on the same SIMD overlap?  `blocks`
 * workgroups of 512 threads; mode bit 0: MFMA waves run, bit 1: VALU waves run, bit 2: MFMA accumulators in AccVGPRs.  sink: `blocks` floats. */
/* diagnostic (tools/stamp_lvl.py): workgroup `workgroup` of level run `run` writes 100 MHz wall-clock stamps per op iteration into
 * buf ([120][8] int64; NULL: off): 0 start, 1 hand-off seen, 2 input gathered + first filter unit landed, 3 main loop done,
 * 4 epilogue done, 5 stores acknowledged + barrier, 6 flag stored and the next filter stream started. */
DMME_API int dmme_debug_level_stamps(void* buf, int run, int workgroup);
DMME_API int dmme_debug_mfma_valu(int mode, int iters, int blocks, void* sink, void* stream);
/* diagnostic (tools/issue_probe.py): what `n_inner` vector instructions of ONE kind per MFMA slot, issued by the partner wave of an
 * MFMA wave on the same SIMD, cost either wave - all inline asm, per-wave cycle counts (s_memtime) into sink[2 b] (MFMA wave 0) and
 * sink[2 b + 1] (vector wave 4), `blocks` x 2 int64.  kind: 0 v_fma_f32, 1 v_pk_fma_f32, 2 v_exp_f32, 3 v_cvt_pk_bf16_f32,
 * 4 v_pk_mul_f32, 5 v_pk_add_f32, 6 v_add_f32, 7 v_rcp_f32, 8 / 9 the GroupNorm + SiLU prologue of one halo dword in plain / packed
 * instructions, 10 ds_write_b128, 11 ds_read_b128, 12 bf16 unpack (shift / mask), 13 / 14 a 1-KB request in a filter tap's access shape
 * (eight 128-byte rows 2304 bytes apart out of `src`, >= 1.2 MB, L2-resident) by LDS-DMA / into registers, eight in flight per wave.
 * flags: bit 0 MFMA waves run, bit 1 vector waves run, bit 2 / 3 s_setprio 3 on the MFMA / vector waves, bit 4 16x16x32 instead of
 * 32x32x16 MFMAs, bit 5 the MFMA waves also read six 16-byte LDS fragments per eight MFMAs. */
DMME_API int dmme_debug_issue_probe(int kind, int n_inner, int iters, int flags, int blocks, void* sink, const void* src, void* stream);
DMME_API int dmme_debug_l2_stream(const void* buf, int64_t bytes, int iters, int mode, int depth, int blocks, void* sink, void* stream);
/* global L2 norm of a flat fp32 gradient buffer (clip_grad_norm_; scratch: 1024 floats) */
DMME_API int dmme_grad_norm(const float* grad, int64_t numel, float* norm_out, float* scratch, void* stream);
/* one fused pass: clip by global norm (max_norm <= 0: off) -> Adam (torch.optim.Adam, no weight
 * decay; reference lit_modules/ddpm.py:130) -> optional EMA (reference callbacks/ema.py:169-176;
 * ema may be NULL).  `step` is 1-based; grad_norm is the device scalar of dmme_grad_norm.
 * grad_scale multiplies the gradient (and its norm) first: 1 / world when the exchange left rank SUMS in the buffer, else 1. */
DMME_API int dmme_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, float* ema, int64_t numel,
                            float lr, float beta1, float beta2, float eps, int step, const float* grad_norm, float max_norm,
                            float ema_decay, float grad_scale, void* stream);
/* Dynamic loss scaling for half-precision training (precision="fp16"), device-resident - what torch.cuda.amp.GradScaler does around
 * the reference's `precision: 16`, `amp_backend: native` step (configs/ddpm/cifar10.yaml:53,66; scripts/main.py:44), without a
 * read-back.  amp_state: 8 floats in device memory - [0] scale S, [1] finite steps since the last change of S, [2] optimiser steps
 * taken, [3] found_inf of the last step, [4] steps skipped.
 *   dmme_amp_init:  S = init_scale (GradScaler: 65536), the rest 0.
 *   dmme_amp_scale: grad *= S - on the loss gradient handed to dmme_unet_backward, i.e. backward of S * loss.
 *   dmme_adam_step_amp: dmme_adam_step on the SCALED gradient: divides S out (with grad_scale), clips the unscaled norm (grad_norm is
 *     dmme_grad_norm of the scaled buffer - always required here: it is also the inf / NaN detector), takes the bias-correction step
 *     count from amp_state[2]; a non-finite norm leaves parameters, moments and EMA untouched and multiplies S by backoff_factor
 *     (0.5), growth_interval (2000) finite steps in a row multiply it by growth_factor (2). */
DMME_API int dmme_amp_init(float* amp_state, float init_scale, void* stream);
DMME_API int dmme_amp_scale(float* grad, int64_t numel, const float* amp_state, void* stream);
DMME_API int dmme_adam_step_amp(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, float* ema, int64_t numel, float lr, float beta1,
                                float beta2, float eps, const float* grad_norm, float max_grad_norm, float ema_decay, float grad_scale,
                                float* amp_state, float growth_factor, float backoff_factor, int growth_interval, void* stream);
/* ---- data-parallel gradient exchange with bf16 on the wire (the reference relies on Lightning DDP for this step:
 * configs/ddpm/cifar10.yaml:28,51,62 `devices`, `strategy`, `replace_sampler_ddp`; NCCL's fp32 ring all-reduce there).  One bucket =
 * pack (fp32 -> bf16, zero padded to world * per_rank) -> all-to-all of the per-rank shards (torch.distributed / RCCL) ->
 * shard_reduce (fp32 accumulation in rank order, x scale = 1 / world, ONE rounding) -> all-gather of the bf16 means -> unpack into
 * the fp32 gradient buffer: every rank ends with identical bits; each GPU sends and receives 2 x numel / world x 2 bytes per peer. */
DMME_API int dmme_grad_pack_bf16(const float* grad, int64_t numel, void* dst_bf16, int64_t numel_padded, void* stream);
DMME_API int dmme_shard_reduce_bf16(const void* recv_bf16, int world, int64_t per_rank, float scale, void* out_bf16, void* stream);
DMME_API int dmme_grad_unpack_bf16(const void* src_bf16, int64_t numel, float* grad, void* stream);

/* Copy an intermediate activation (the output of module `name`, e.g. "down_layers.3",
 * "input_conv", "condition") out of the workspace as fp32 NCHW for parity tests.
 * numel_cap guards the destination size. */
DMME_API int dmme_unet_debug_read(const dmme_plan* plan, const void* workspace, const char* name, float* dst,
                         int64_t numel_cap, int64_t* numel_out, void* stream);

/* Dropout2d multipliers from the library's own Philox stream (train mode without
 * injected masks): keep with probability 1-p, value 1/(1-p). */
DMME_API int dmme_dropout_masks(const dmme_plan* plan, uint64_t seed, uint64_t offset, float* masks, void* stream);

/* ---- diffusion process (elementwise; all NCHW fp32) --------------------------------- */

/* standard normal fill from the library's Philox4x32-10 stream: replaces dmme.gaussian
 * (src/dmme/common/noise.py:4-6) / the draw inside Normal.sample(). */
DMME_API int dmme_randn(float* out, int64_t numel, uint64_t seed, uint64_t offset, void* stream);

/* forward noising: replaces forward_process(...).sample() and the target re-derivation
 * of DDPM.training_step (equations/ddpm/ddpm.py:36-41, diffusion_models/ddpm.py:72-79):
 *   mean = sqrt_abar[t_n] x0 ; std = sqrt_1m_abar[t_n] ; x_t = mean + std z ;
 *   target = (x_t - mean)/std   (optional).
 * sqrt_abar / sqrt_1m_abar: device fp32 tables indexed by t holding sqrt(abar_t) and
 * sqrt(1 - abar_t), evaluated on the host with the reference's fp32 torch ops. */
DMME_API int dmme_q_sample(const float* x0, const float* z, const float* sqrt_abar, const float* sqrt_1m_abar,
                  const int64_t* t, int B, int64_t chw, float* x_t, float* target, void* stream);

/* one DDPM reverse update, in place on x: replaces DDPM.sampling_step after the model
 * call (diffusion_models/ddpm.py:99-110, equations/ddpm/ddpm.py:65-71):
 *   mean = inv_sqrt_alpha * (x - eps_coef * eps);  x = add_noise ? mean + sigma * z : mean
 * The three coefficients are the reference's fp32 values 1/sqrt(alpha_t),
 * beta_t/sqrt(1-abar_t), sqrt(beta_t); add_noise = (t != 1). */
DMME_API int dmme_ddpm_step(float* x, const float* eps, const float* z, float inv_sqrt_alpha, float eps_coef,
                   float sigma, int add_noise, int64_t numel, void* stream);

/* one DDIM update as shipped: replaces DDIM.sampling_step after the model call
 * (diffusion_models/ddim.py:73-77, equations/ddim/ddim.py:52-57):
 *   x = sqrt_abar_prev * ((x - sqrt_one_minus_abar * eps) / sqrt_abar_prev). */
DMME_API int dmme_ddim_step(float* x, const float* eps, float sqrt_one_minus_abar, float sqrt_abar_prev, int64_t numel,
                   void* stream);

/* simple_loss (equations/ddpm/losses.py:13): loss[0] = mean((target-eps)^2); optional
 * d_eps = 2 (eps - target) / numel * grad_scale.  `scratch` needs 1024 floats. */
DMME_API int dmme_mse_loss(const float* eps, const float* target, int64_t numel, float* loss, float* d_eps,
                  float grad_scale, float* scratch, void* stream);

/* ---- input pipeline: training batch from an HBM-resident uint8 image set ---------------------------------------------
 * Replaces the per-sample transform chain of the reference's data module (data_modules/cifar10.py:39-44:
 * [RandomHorizontalFlip] -> torchvision ToTensor (uint8 / 255) -> norm, common/norm.py:4-6) plus the DataLoader's collate:
 *   out[b] = (flip[b] ? hflip : id)(data[idx[b]]) / 255, then (x - 0.5) * 2, fp32 NCHW.
 * data: [n_images][C][H][W] uint8 (the CIFAR10 python-pickle layout); idx: int64[B] (caller guarantees 0 <= idx < n_images);
 * flip: uint8[B] or NULL (no augmentation, the reference's test set); W must be a multiple of 4. */
DMME_API int dmme_image_batch(const uint8_t* data, int64_t n_images, const int64_t* idx, const uint8_t* flip, int B, int C, int H, int W,
                     float* out, void* stream);

/* ---- one replayable denoising step ---------------------------------------------------------------------
 * The reference's sampling loops run on the host: `for t in range(T, 0, -1)` around `sampling_step`
 * (diffusion_models/ddpm.py:130, ddim.py:96, iddpm.py), `LitDDPM.forward` builds `torch.tensor([t])` - a host-to-device copy -
 * every step (lit_modules/ddpm.py:77; called per step by callbacks/generate.py:82).  Here everything that varies from step to
 * step is device-resident, so ONE launch sequence (time MLP + UNet + noise draw + update + loop-state advance) serves every
 * step and can be captured in a hipGraph and replayed:
 *   state     64 bytes of device memory (eight 64-bit words): int64 i (loop index), int64 t (= t_table[i], what the network is
 *             evaluated at), uint64 Philox offset in quads, uint64 Philox seed, uint32 ticket + pad, three reserved words.
 *             dmme_chain_set initialises it (a one-thread kernel: the values travel as kernel arguments, no host buffer to
 *             keep alive, and a graph captured once serves any later seed).
 *   t_table   int64[n+1]: DDPM / IDDPM: t_table[i] = i; DDIM: the tau table (diffusion_models/ddim.py:41-53)
 *   step_coef float[n+1][4], per loop index:
 *             DMME_CHAIN_DDPM  {1/sqrt(alpha_t), beta_t/sqrt(1-abar_t), sqrt(beta_t), -}     (equations/ddpm/ddpm.py:65-71)
 *             DMME_CHAIN_DDIM  {sqrt(1-abar_tau_i), sqrt(abar_tau_{i-1}), -, -}              (equations/ddim/ddim.py:52-57)
 *             DMME_CHAIN_IDDPM {1/sqrt(alpha_t), beta_t/sqrt(1-abar_t), log beta_t, log max(beta~_t, 1e-12)}
 * dmme_chain_update = the sampler update alone (noise drawn in the kernel from Philox(state.seed, state.offset + quad index): the values
 * dmme_randn would produce at the same offset, so a chain equals the eager loop bit for bit); no noise is added at t == 1 but
 * the offset advances all the same (the reference draws and discards, diffusion_models/ddpm.py:107-110).  After the update the
 * state moves on: i -= 1, t = t_table[i], offset += B*chw/4.  dmme_chain_step = dmme_unet_forward at t = state.t followed by
 * dmme_chain_update; x is updated in place, model_out receives the network output.  chw must be a multiple of 4. */
enum { DMME_CHAIN_DDPM = 0, DMME_CHAIN_DDIM = 1, DMME_CHAIN_IDDPM = 2 };
DMME_API int dmme_chain_set(void* state, int64_t i, const int64_t* t_table, uint64_t philox_seed, uint64_t philox_offset, void* stream);
DMME_API int dmme_chain_update(int kind, float* x, const float* model_out, const float* step_coef, const int64_t* t_table, void* state,
                      int B, int64_t chw, void* stream);
DMME_API int dmme_chain_step(const dmme_plan* plan, const void* packed, float* x, float* model_out, void* workspace, int kind,
                    const float* step_coef, const int64_t* t_table, void* state, void* stream);

/* ---- Improved DDPM (learned variance): model_out is (B, 2C, H, W), channels [0, C) = eps, [C, 2C) = v
 * (IDDPM.forward_model, diffusion_models/iddpm.py:152-164); chw = C*H*W of ONE image of x. */

/* one reverse update, in place on x: replaces IDDPM.sampling_step after the model call
 * (diffusion_models/iddpm.py:118-150, equations/ddpm/ddpm.py:65-71, equations/iddpm/losses.py:34-37):
 *   mean = inv_sqrt_alpha * (x - eps_coef * eps);  std = sqrt(exp(v log_beta + (1 - v) log_beta_tilde));
 *   x = add_noise ? mean + std * z : mean.   log_beta_tilde = log(max(beta~_t, 1e-12)); add_noise = (t != 1). */
DMME_API int dmme_iddpm_step(float* x, const float* model_out, const float* z, float inv_sqrt_alpha, float eps_coef, float log_beta,
                    float log_beta_tilde, int add_noise, int B, int64_t chw, void* stream);

/* hybrid / VLB training loss and its gradient w.r.t. model_out: replaces the loss side of IDDPM.training_step
 * (diffusion_models/iddpm.py:92-116) = eq.ddpm.simple_loss + gamma * eq.iddpm.loss_vlb (equations/iddpm/losses.py:40-98,
 * discrete NLL rows for t == 1, KL rows otherwise, stop-gradient on eps inside L_vlb).
 *   loss[0] = w_simple * L_simple + w_vlb * L_vlb, loss[1] = L_simple, loss[2] = L_vlb   (3 floats);
 *   d_out (optional, B x 2C x H x W) = grad_scale * d loss[0] / d model_out.
 * t: int64[B] per-sample timesteps; coef: device fp32 table of 8 floats per timestep evaluated on the host with the
 * reference's fp32 torch ops: {1/sqrt(alpha_t), beta_t/sqrt(1-abar_t), log beta_t, log max(beta~_t, 1e-12),
 * sqrt(abar_{t-1}) beta_t/(1-abar_t), sqrt(alpha_t)(1-abar_{t-1})/(1-abar_t), sqrt(beta~_t), 0}.
 * target: the noise re-derived by dmme_q_sample.  `scratch` needs 1024 floats. */
DMME_API int dmme_iddpm_loss(const float* model_out, const float* x_t, const float* x_0, const float* target, const int64_t* t,
                    const float* coef, int B, int64_t chw, float w_simple, float w_vlb, float* loss, float* d_out,
                    float grad_scale, float* scratch, void* stream);

/* ---- single-op entry points used by the unit parity tests ---------------------------
 * Activations here are NHWC in the compute dtype; weights in the packed layout
 * [Cout][taps][Cin]; they exercise exactly the kernels the plan launches.
 * force_generic = 1 selects the shape-generic kernel instead of the MFMA kernel. */
typedef struct {
    int dtype;         /* dmme_dtype of activations / weights */
    int N, Hin, Win;   /* source dims (before the optional nearest 2x upsample) */
    int C1, C2;        /* channels of source 1 and (concat) source 2; C2 = 0: none */
    int upsample;      /* 1: nearest 2x upsample fused in front of the conv */
    int stride;        /* 1 or 2 */
    int taps;          /* 9: 3x3 pad 1; 1: 1x1 pad 0 */
    int Cout;
    int pro_silu;      /* SiLU after the affine prologue */
    int out_silu;      /* SiLU on the output */
    int nt;            /* rows of tproj (1 or N); 0: no time-embedding add */
    int tproj_ld;      /* leading dimension of tproj */
    int in_nchw;       /* src1 is fp32 NCHW (network input) */
    int out_nchw;      /* dst is fp32 NCHW (network output) */
    int force_generic;
} dmme_conv_desc;

DMME_API int dmme_conv2d(const dmme_conv_desc* d, const void* src1, const void* src2, const void* weight,
                const float* bias, const float* scale, const float* shift, const float* dmask,
                const float* tproj, const void* res1, const void* res2, int R1, void* dst, void* stream);

/* The second half of a channel-changing ResBlock as ONE launch (models/ddpm.py:108-111,131: `conv2(act(h)) + residual(x)`): the 3x3 conv
 * of dmme_conv2d over src1 (++ src2) with its prologue, plus the block's 1x1 residual conv over the RAW block input r_src1 [.., r_C1]
 * (++ r_src2 [.., r_C2]) with r_weight packed [Cout][r_C1 + r_C2] and r_bias [Cout], accumulated into the same output tile - no
 * residual tensor.  16-bit dtypes, stride 1, the shapes the wave-specialised kernel's 256-pixel form takes (whole 128-cout tiles,
 * r_C1 a multiple of 64, r_C1 + r_C2 a multiple of 128 and <= 512); DMME_ERR_UNSUPPORTED otherwise - it never falls back. */
DMME_API int dmme_conv2d_res(const dmme_conv_desc* d, const void* src1, const void* src2, const void* weight, const float* bias,
                const float* scale, const float* shift, const float* dmask, const void* r_src1, const void* r_src2, int r_C1,
                int r_C2, const void* r_weight, const float* r_bias, void* dst, void* stream);

/* GroupNorm statistics folded with the affine: scale[n][c] = rstd*gamma[c],
 * shift[n][c] = beta[c] - mean*rstd*gamma[c]  (nn.GroupNorm(eps=1e-5), models/ddpm.py:17-18). */
DMME_API int dmme_groupnorm_scale_shift(int dtype, const void* src1, const void* src2, int N, int HW, int C1, int C2,
                               int groups, const float* gamma, const float* beta, float eps, float* scale,
                               float* shift, float* partial_scratch, int force_generic, void* stream);

/* single-head self-attention over S tokens: qkv [N][S][3C] -> out [N][S][C]
 * softmax(q (k*C^-0.5)^T) v   (Attention.forward_attention, models/ddpm.py:54-63). */
DMME_API int dmme_attention(int dtype, const void* qkv, int N, int S, int C, void* out, int force_generic, void* stream);
/* The rest of the block in the same launch: dst = res + proj(attention(qkv)) (Attention.forward, models/ddpm.py:66-75, behind the
 * norm + qkv conv): w [C][C] (16-bit, the packed 1x1 layout: row = cout), bias [C] fp32, res / dst [N][S][C].  ctx (nullable):
 * also receives the attention output [N][S][C] (the proj conv's input, which its weight gradient needs).  gn_part (nullable):
 * (mean, M2) of dst per (image, 32-token tile, group of gn_cg = 4 or 8 channels), [N][S/32][C/gn_cg][2] fp32 - the partials the
 * next GroupNorm is finished from.  16-bit dtypes, S = 256, C = 128 or 256, N >= 128; DMME_ERR_UNSUPPORTED otherwise. */
DMME_API int dmme_attention_proj(int dtype, const void* qkv, int N, int S, int C, const void* w, const float* bias, const void* res,
                                 void* dst, void* ctx, float* gn_part, int gn_cg, void* stream);

/* multi-head self-attention exactly as the reference ships it (MultiHeadAttention.forward_attention,
 * models/iddpm.py:35-47): qkv [N][S][3C]; head h of image n reads channels [h*3d, (h+1)*3d) as (q | k | v), d = C/heads;
 * K is scaled by C^-0.5; result row n*heads + h is written to out[(n*heads + h) % N][S][((n*heads + h) / N)*d ...]
 * (the reference's "(b head)" split merged back as "(head b)").  heads = 1 equals dmme_attention. */
DMME_API int dmme_attention_heads(int dtype, const void* qkv, int N, int S, int C, int heads, void* out, int force_generic, void* stream);

/* NCHW fp32 <-> NHWC compute-dtype converters (test plumbing for the single-op calls) */
DMME_API int dmme_nchw_to_nhwc(int dtype, const float* src, int N, int C, int HW, void* dst, void* stream);
DMME_API int dmme_nhwc_to_nchw(int dtype, const void* src, int N, int C, int HW, float* dst, void* stream);
/* reference-layout fp32 weight (Cout, Cin, k, k) -> packed [Cout][k*k][Cin] in dtype */
DMME_API int dmme_pack_weight(int dtype, const float* src, int Cout, int Cin, int taps, void* dst, void* stream);

/* diagnostic: device buffer of >= 128 int64 that the wave-specialised conv kernel launched through dmme_conv2d fills
 * with shader-clock stamps of consumer wave 0 of workgroup 0: [0..63] arrive / leave of every stage barrier,
 * [64..87] the epilogue passes (tools/stamp_ws.py); NULL switches it off. */
DMME_API int dmme_debug_set_stamps(void* buf);

/* timing helper: records a HIP event pair around nothing; used by bench.py to time
 * on the launch stream.  Returns elapsed ms between two events created by the lib. */
DMME_API int dmme_event_create(void** ev);
DMME_API int dmme_event_record(void* ev, void* stream);
DMME_API int dmme_event_elapsed_ms(void* start, void* stop, float* ms);
DMME_API int dmme_event_destroy(void* ev);

#ifdef __cplusplus
}
#endif
#endif /* DMME_HIP_H */
