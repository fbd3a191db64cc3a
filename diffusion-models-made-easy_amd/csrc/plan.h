// Shared definitions of the plan translation units (plan.hip: layer graph, layouts, forward; plan_lvl.hip: level-engine planning
// and status; plan_bwd.hip: backward pass, gradient buckets): the plan's data model and the helpers that cross those files.
#pragma once
#include <stdarg.h>
#include <stdlib.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <string>
#include <unordered_map>
#include <vector>

#include "common.h"
#include "lvl.h"

namespace dmme {

static inline int64_t align_up(int64_t v, int64_t a) { return (v + a - 1) / a * a; }

struct Param {
    std::string name;
    int ndim = 0;
    int64_t shape[4] = {1, 1, 1, 1};
    int64_t ref_off = 0;     // elements, fp32 reference-layout flat buffer
    int64_t packed_off = 0;  // bytes
    int64_t packed_bwd_off = -1;  // bytes in the data-gradient weight buffer (conv weights only)
    int64_t wp_off = -1;          // float offset of this conv weight in the packed-layout gradient image
    bool is_buffer = false;
    bool as_f32 = true;      // stays fp32 in the packed buffer (bias / gamma / beta / freqs)
    int pack_code = -1;      // mixed plans: PackItem::as_f32 code of this conv weight (3: fp32 [co][tap][ci]; 4: split fp16 halves), -1: by as_f32
    int cout = 1, cin = 1, taps = 1;
    int64_t numel() const { return shape[0] * shape[1] * shape[2] * shape[3]; }
};

struct Tensor {  // an activation in the workspace, NHWC in the compute dtype
    int64_t off = 0;
    int C = 0, H = 0, W = 0;
    int f32 = 0;  // precision="fp16r32": this tensor of a 16-bit plan is stored in fp32 (the full-resolution level)
    // GroupNorm partials emitted by the producing conv's epilogue (-1: none): [B][tiles][G][2] floats
    int64_t stats_off = -1;
    int stats_tiles = 0, stats_cnt = 0;
};

enum OpKind { OP_SINUS, OP_LINEAR, OP_GN, OP_CONV, OP_ATTN, OP_CAST };

struct Op {
    OpKind kind;
    // OP_LINEAR: in (fp32 ws offset) -> out (fp32 ws offset)
    int64_t lin_in = 0, lin_out = 0;
    int lin_K = 0, lin_N = 0, lin_w = -1, lin_b = -1, lin_silu = 0;
    int64_t lin_pre = -1;  // workspace offset of the layer's pre-activation copy (time MLP, training batch), -1: none
    // OP_GN
    int gn_src1 = -1, gn_src2 = -1, gn_gamma = -1, gn_beta = -1;
    int64_t b_rowsum = 0, b_ab = 0;  // backward scratch (bytes in the zeroed region): column sums of dY, GroupNorm channel sums
    int64_t b_gnrows = -1;           // [2][N][C] per-image sums for dbeta / dgamma of this conv's GroupNorm (written whole)
    int gn_rows_deferred = 0;        // ... reduced over the batch by the grouped bias launch instead of same-address atomics
    int64_t gn_scale = 0, gn_shift = 0, gn_mr = 0;  // workspace offsets: scale/shift [N][C], {mean, rstd} [N][G][2]
    // scale-shift conditioning (iddpm.ResBlock, models/iddpm.py:117-118): columns of tproj holding (shift | scale), -1: none.
    // The GroupNorm output becomes GN(h) * (scale + 1) + shift, folded into the per-(n, c) scale / shift the consumer applies.
    int gn_mod_col = -1, gn_mod_C = 0;
    // small maps: this GroupNorm also writes its consumer's pre-activated input (gn_small_kernel); -1: the conv applies the affine itself
    int64_t gn_act = -1;   // workspace offset of act [N][HW][C] in the compute dtype
    int gn_force_small = 0;  // statistics from the one-workgroup-per-image kernel even where the producers left partials (it writes act)
    int gn_consumer = -1;  // the conv op that reads it (its pro_silu / Dropout2d mask define the activation)
    int gn_direct = 0;     // every source's producing conv finishes this norm in its epilogue (ConvArgs::gno): no launch here
    int gn_in_consumer = 0;  // the consuming conv merges the producers' partials itself (ConvArgs::gni): no launch here
    // OP_CONV
    int src1 = -1, src2 = -1;  // tensor ids; -2: network input (NCHW fp32)
    int w = -1, b = -1;
    int gn = -1;               // op index of the GN providing scale/shift
    int pro_silu = 0, out_silu = 0;
    int64_t dmask_off = -1;    // float offset into the drop-mask buffer
    int tproj_col = -1;        // column offset into tproj
    int res1 = -1, res2 = -1;
    int dst = -1;              // tensor id; -2: network output (NCHW fp32)
    int up = 0, stride = 1, taps = 9;
    int use_act = 0;           // FORWARD reads the pre-activated tensor of its GroupNorm (backward still works from src1 / src2 + scale / shift)
    int gd_n = 0, gd_gn[2] = {-1, -1}, gd_coff[2] = {0, 0};  // norms this conv's forward epilogue finishes (op index, channel offset in the norm)
    int gd_act = -1;           // which of them also gets the consumer's pre-activated input written (-1: none)
    int res_alias = 0;         // backward: the residual input's gradient buffer is this conv's output gradient buffer (no copy)
    // FORWARD fusion of a ResBlock's 1x1 residual conv into conv2 (ConvArgs::r_w, the wave-specialised kernel's residual segment):
    // conv2.rseg = op index of the residual conv, whose own launch is skipped (fused_away) and whose output tensor is never written.
    // The backward pass is untouched: it reads neither that tensor nor these fields.
    int rseg = -1, fused_away = 0;
    int wg_layer = -1;         // index into the grouped weight-gradient table of its kernel size (-1: per-layer kernels)
    int64_t wg_act = -1;       // backward workspace offset of its pre-activated input, written by its GroupNorm's backward for the
                               // deferred weight gradient (-1: none)
    int bias_deferred = 0;     // its bias / time-projection reduction runs in the grouped launch
    // OP_ATTN
    int at_qkv = -1, at_out = -1, at_heads = 1;
    int64_t at_lse = 0;  // workspace offset of the forward's log-sum-exp [N][S]
    // FORWARD fusion of the block's proj conv + residual add into the attention launch (attn_mfma.hip, AttnProj): at_proj = op index of
    // that conv, whose own launch is skipped (fused_away = 2).  The context tensor at_out is then written only by forwards a backward
    // pass may follow (run_op's keep_ctx); the backward pass itself is untouched.
    int at_proj = -1;
    // precision="fp16r32" (dmme_plan::mix): how this conv of the fp32 level runs.  mix: ConvArgs::mix (1 / 2: split-pass 3x3 kernel, 3: split-pass
    // thin output conv); route_f32: on the fp32-tensor kernels with three-pass bf16 products (input conv, the blocks' 1x1 residual convs)
    int mix = 0, route_f32 = 0;
    int mix2 = 0;  // split-pass 3x3 kernel: TWO passes (hi.hi + lo.hi: the filter's lo half dropped) instead of three - ConvArgs::mix2
    // OP_CAST: fp32 tensor -> 16-bit copy (the stride-2 conv that leaves the fp32 level reads it)
    int cast_src = -1, cast_dst = -1;
    // level engine (lvl_engine.hip): index of the run that executes this op (-1: its own launch); the run's first op launches it
    int lvl = -1, lvl_first = 0;
};

// one persistent launch for a stretch of the op list on a 4x4 / 8x8 map (lvl.h)
struct LvlRun {
    int op_first = 0, op_last = 0;  // plan ops [op_first, op_last]
    int sh = 0, NG = 0, NGS = 0, GB = 1, NJ = 1;
    std::vector<LvlOp> ops;
    std::vector<std::pair<int, int>> made;  // (tensor id, index of the op that produces / normalises it): later runs attach norms there
    LvlOp* ops_dev = nullptr;
    // the op table of forwards no backward pass follows (run_op's keep_ctx == false): convs whose RAW output no forward op reads - a
    // ResBlock's conv1 (only its norm's pre-activated copy is read), the qkv slices kept in LDS - do not store it (assign_lvl_nograd)
    std::vector<LvlOp> ops_nograd;
    LvlOp* ops_nograd_dev = nullptr;
    int raw_skipped = 0;
    unsigned* sync_dev = nullptr;   // [16] control words (epoch, done, error), then the flag rows [n_ops * 2][NG][LVL_NS]
    double flops = 0, bytes = 0;
};

}  // namespace dmme

using namespace dmme;

struct dmme_plan {
    dmme_unet_cfg cfg;
    int B, H, W, dtype, device;
    int x3 = 0;            // DMME_BF16X3: dtype is DMME_F32 (storage), the convolutions take the three-pass bf16 MFMA path
    int mix = 0;           // DMME_F16R32: dtype is DMME_F16; the tensors of the full-resolution level are fp32 and its convolutions run
                           // three fp16 MFMA passes on hi / lo halves (or, for the few small ones, the fp32-tensor kernels above)
    int out_channels = 0;  // in_channels (DDPM) or 2 * in_channels (IDDPM: eps, v)
    std::vector<Param> params;
    std::vector<Tensor> tensors;
    std::vector<Op> ops;
    std::unordered_map<std::string, int> named;  // module name -> tensor id
    int64_t ref_numel = 0, packed_bytes = 0, ws_bytes = 0, dropmask_numel = 0;
    int64_t ws_tsin = 0, ws_th1 = 0, ws_temb = 0, ws_tproj = 0, ws_gnpart = 0;
    int64_t ws_tz1 = -1, ws_tz2 = -1;  // pre-activations of the two time-MLP layers (written at training batch; the backward's SiLU')
    int64_t ws_splitk = 0, splitk_floats = 0;  // split-K partial sums of the small-map convolutions (forward and data gradient)
    int tproj_cols = 0;
    int64_t tproj_w_off = 0, tproj_b_off = 0;  // packed byte offsets of the concatenated projection
    int freqs_param = -1;
    PackItem* items_dev = nullptr;
    int n_items = 0;
    int n_launches = 0;
    // ---- training (backward) ----
    struct TBlock { int tw, tb, col, cout; };
    std::vector<TBlock> tblocks;           // per-ResBlock time projection parameters
    int p_l1w = -1, p_l1b = -1, p_l2w = -1, p_l2b = -1;
    int64_t packed_bwd_bytes = 0, bws_bytes = 0;
    std::vector<int64_t> gt_off;           // gradient buffer of every forward tensor
    int64_t bws_zero = 0, bws_zero_bytes = 0, bws_wimage = 0, bws_gnS = 0, bws_zpage = 0;  // region cleared once per backward
    PackItem* items_unpack_dev = nullptr;
    int n_items_unpack = 0;
    int64_t bws_tmp = 0, bws_dy = 0, bws_rowsum = 0, bws_dtproj = 0, bws_dtemb = 0, bws_dh1 = 0, bws_z = 0, bws_wT = 0, bws_attP = 0,
            bws_attdS = 0;
    PackItem* items_bwd_dev = nullptr;
    int n_items_bwd = 0;
    // grouped weight gradients (one launch per backward)
    struct WgGroup {
        int taps = 0;
        std::vector<WgLayer> layers;
        std::vector<WgJob> jobs;
        WgLayer* layers_dev = nullptr;
        WgJob* jobs_dev = nullptr;
        int dma = 0;  // every layer's second operand is one prologue-free tensor: the LDS-DMA kernel runs the table
        int stride = 1;
    } wg[3];  // 3x3, 1x1, 3x3 stride 2 (LDS-DMA kernel only)
    // Bucketed backward (gradient exchange overlapped with backward): the op list is cut at ResBlock boundaries into stretches that
    // backward finishes one after the other (bucket 0 = output conv + the last up blocks, ... the last bucket = the first down
    // blocks, input conv and time MLP); every deferred table is split along the same cuts at plan time, so a bucket's parameter
    // gradients are complete - and handed to the exchange - as soon as the reverse walk leaves its stretch.  Cut so that no bucket
    // holds more than ~1/6 of the parameters: the LAST one, whose exchange nothing hides, is <= 15 % of the bytes.
    struct GradBucket {
        int op_lo = 0, op_hi = 0;                          // plan ops [op_lo, op_hi)
        WgGroup wg[3];
        int col0 = 0, col1 = 0, bias0 = 0, bias1 = 0;      // ranges of col_jobs / bias_jobs
        std::vector<std::pair<int, int>> unpack, tcols;    // ranges of items_unpack / of time-projection columns
        std::vector<std::pair<int64_t, int64_t>> ranges;   // (flat offset, numel) of its parameters, merged
    };
    std::vector<GradBucket> gb;                            // empty: no clean cut for this configuration (one piece)
    // batched time-projection gradients: destination (float offset into grad_flat) of every 64-row tile of
    // dtproj^T temb, then of every 32-column tile of the bias sums
    int64_t* tp_tiles_dev = nullptr;
    int tp_n64 = 0;
    // deferred bias / time-projection reductions (one launch per backward)
    std::vector<BiasJob> bias_jobs;
    BiasJob* bias_jobs_dev = nullptr;
    std::vector<ColJob> col_jobs;     // column sums of dY of every bias-deferred conv: one grouped launch per flush
    ColJob* col_jobs_dev = nullptr;
    std::vector<LvlRun> lvl_runs;     // level-engine launches (small maps)
    // every workgroup of an engine launch must be resident at once: grids are sized by what the device holds (assign_levels)
    int lvl_max_wg = LVL_MAX_WG;
    // host-visible status word of the engine's bounded hand-off waits (pinned, device-mapped; null: no engine run in this plan):
    // non-zero = 1 + index of a run in which a wait timed out, i.e. the outputs since are invalid (lvl_check)
    unsigned* err_host = nullptr;
    // the workspace the last forward wrote WITHOUT the tensors only a backward pass reads (dmme_unet_forward_nograd, dmme_chain_step):
    // dmme_unet_backward refuses it instead of differentiating stale activations (null: the last forward kept everything)
    mutable const void* nograd_ws = nullptr;
};

namespace dmme {

// precision="fp16r32": the split-pass 3x3 convs (op order: down block 0 conv1 / conv2, down block 1 conv1 / conv2, the up-sampling conv
// that enters the level, up blocks conv1 / conv2 x 3) that run two passes instead of three (tests/test_gpu_fp16.py holds the result at 1e-3)
constexpr int kR32TwoPassDefault = 0;
// the dtype code a conv's kernels are selected by: the plan's, except the fp32-routed convs of a mixed plan
static inline int conv_dt(const dmme_plan* P, const Op& o) { return o.route_f32 ? DMME_F32 : P->dtype; }
static inline int wg_index(const Op& o) { return o.taps == 1 ? 1 : o.stride == 2 ? 2 : 0; }

// plan.hip
int run_any_conv(int dtype, const ConvArgs& a, hipStream_t s);
void fill_conv(const dmme_plan* P, const Op& o, const char* packed, const float* x, float* y, char* ws, const float* drop_masks, int nt, ConvArgs& a,
               bool fwd = false);
bool gn_from_parts(const dmme_plan* P, const Op& o);
// plan_lvl.hip: which stretches of the op list become level-engine runs; their launch; the bounded waits' status word
void assign_levels(dmme_plan* P);
// plan.hip: which residual 1x1 convs ride in their block's conv2 (after the level runs are known)
void assign_rseg(dmme_plan* P);
int run_level(const dmme_plan* P, const LvlRun& R, const char* pk, char* ws, int nt, const float* drop_masks, hipStream_t s, bool keep_ctx = true);
void assign_lvl_nograd(dmme_plan* P);
int lvl_check(const dmme_plan* P, const char* where, hipStream_t stream = nullptr, bool have_stream = false);
// plan_bwd.hip: the grouped (deferred) weight-gradient tables of ops [op_lo, op_hi)
void build_wgrad_group(dmme_plan* P, dmme_plan::WgGroup& G, int gi, int op_lo = 0, int op_hi = 1 << 30);

}  // namespace dmme
