// Level engine (lvl_engine.hip): descriptors shared between the plan builder (plan.hip) and the kernel.
//
// A "level run" is a maximal stretch of the UNet's op list whose tensors all live on one small map (4x4 or 8x8: at most 64 pixels
// per image): the ResBlocks of down_layers / middle_layers / up_layers of that resolution with their GroupNorms, time-embedding
// rows, residual 1x1 convs and (4x4 maps) the single-head attention block (reference: models/ddpm.py:118-133, 38-75, 297-313).
// One persistent launch runs the whole stretch.  Work unit = (pixel group g, cout slice s): 64 consecutive pixels of the NHWC tensors
// (4 whole 4x4 images / one 8x8 image) x 32 output channels; the LVL_NS = 8 workgroups of a group exchange their output slices
// through global memory with write-through stores and one flag word per (op, group, slice), see lvl_engine.hip.
#pragma once
#include <stdint.h>

namespace dmme {

constexpr int LVL_NS = 8;   // cout slices per pixel group (Cout = 256 = 8 x 32 for every op the engine takes)
constexpr int LVL_BN = 32;  // couts per slice
constexpr int LVL_BM = 64;  // pixels per group
constexpr int LVL_MAX_WG = 256;

enum { LVL_CONV = 0, LVL_NORM = 1, LVL_ATTN = 2 };

// A GroupNorm finished by the op that PRODUCES (one of) its source tensor(s): a 32-channel slice holds whole groups of every norm
// that reads the tensor (group sizes 8 and 16), and the 64-pixel group holds whole images, so the producing workgroup has complete
// per-(image, group) statistics.  It writes the norm's scale / shift / {mean, rstd} rows (kept for the backward pass) and the
// consumer conv's pre-activated input act = T(silu?(x * scale + shift) * mask) - each element normalised once, by its producer.
struct LvlNorm {
    int64_t gamma_off, beta_off;  // bytes into the packed parameter buffer: fp32 [Cn]
    int64_t scale_off, shift_off; // bytes into the workspace: fp32 [N][Cn]
    int64_t mr_off;               // bytes into the workspace: fp32 [N][Cn / cg][2] {mean, rstd}
    int64_t act_off;              // bytes into the workspace: T [N][HW][Cn]
    int64_t dmask_off;            // floats into the Dropout2d mask buffer: [N][Cn] (-1: none)
    int Cn, cg, c_off, act_silu;  // the norm's width and group size; this tensor's channel c is channel c_off + c of the norm
};

struct LvlOp {
    int kind;                 // LVL_CONV / LVL_NORM / LVL_ATTN
    int taps;                 // 9 or 1
    int C1, C2;               // channels of the A operand's sources (C2 = 0: one source); C1 + C2 a multiple of 256, C1 of 64
    int64_t a1_off, a2_off;   // bytes into the workspace: T [N][HW][C1], [N][HW][C2]
    int wait0, wait1;         // flag rows (op indices * 2 + which) that cover the sources; -1: complete before the launch
    int reuse_a;              // the A image of the previous op is this op's too (q / k / v share the normalised input)
    int64_t w_off, b_off;     // bytes into the packed buffer: T [rows][taps][C1 + C2], fp32 [rows]
    int w_row0;               // first weight row of this op's 256 couts (qkv: 0 / 256 / 512)
    int64_t dst_off;          // bytes into the workspace: raw output T [N][HW][dst_C] (LVL_NORM: the tensor that is read)
    int dst_C, dst_c0;        // its width, and the channel the op's 256 outputs start at
    int64_t res_off;          // residual input, same slicing as the output (-1: none)
    int res_C, res_c0;
    int tproj_col;            // column of the batched time projection added per image (-1: none)
    int n_norm;
    LvlNorm norm[2];
    int keep;                 // 0 / 1 / 2: the output slice also stays in LDS as q / k / v of the attention that follows (-1: no)
    int signal;               // publish a flag row (op index * 2) once the slice is in memory
    int64_t sc_off;           // LVL_ATTN: bytes into the workspace, fp32 partial scores [NG][LVL_NS][1024]
    float kscale;             // LVL_ATTN: C^-0.5, applied to K before the product (models/ddpm.py:50,58)
    // second K segment of a conv op: the ResBlock's 1x1 residual conv (models/ddpm.py:108-111,131) over the block's RAW input,
    // accumulated into the same tile as conv2's nine taps - `h + residual(x)` without the residual tensor, its op, its hand-off
    int C3, C4;               // channels of its sources (0: no second segment); C3 + C4 a multiple of 256
    int64_t a3_off, a4_off;   // bytes into the workspace: T [N][HW][C3], [N][HW][C4]
    int wait2, wait3;         // flag rows covering them (-1: complete before the launch)
    int64_t w2_off, b2_off;   // packed 1x1 weights T [256][C3 + C4] and bias fp32 [256]
    int pad_;
};

struct LvlArgs {
    const LvlOp* ops;
    int n_ops;
    char* ws;
    const char* packed;
    const float* drop_masks;  // null: eval
    const float* tproj;       // [nt][tproj_ld]
    int tproj_ld, nt;
    int N, sh;                // batch; log2 of the map's width (= height): 2 or 3
    int NG, NGS, GB;          // pixel groups; iterations resident at once (grid = NGS * LVL_NS / NJ); groups per op iteration of a workgroup (1 / 2)
    int NJ;                   // 32-cout blocks per cout slice: 1 (8 slices per group) or 2 (4 slices of 64 couts; GB = 2 only)
    unsigned* flags;          // [n_ops * 2][NG][LVL_NS]
    unsigned* ctl;            // [0] epoch of the last completed launch, [1] workgroups of this launch that are done, [2] error word
    unsigned* err_sys;        // the plan's host-visible status word (pinned host memory; null: none): set to run_tag by a wait that timed out
    int run_tag;              // 1 + index of this run in the plan
    int spin_limit;           // polls before a hand-off wait gives up (0: LVL_SPIN_LIMIT, ~a second); DMME_DEBUG_ROUTE lvl_spin=
    int xcd_group;            // 1: the slices of a pixel group run on ONE XCD (a permutation of the workgroup index inside windows of 64; speed only)
    int withhold;             // test knob (DMME_DEBUG_ROUTE lvl_withhold=K): workgroup 0 stops signalling from the run's K-th launch on, so its consumers time out
    int max_wg;               // workgroups the device can hold at once (all of a launch must be co-resident)
    long long* stamps;        // diagnostic (null: off): 100 MHz wall-clock stamps of workgroup `stamp_wg`, [op iteration][8] (dmme_debug_set_stamps)
    int stamp_wg;
};

int launch_lvl_engine(int dtype, const LvlArgs& a, hipStream_t s);
// how many workgroups of the engine kernel the device `device` holds at once (compute units x workgroups per unit at the kernel's
// LDS / register footprint), capped at LVL_MAX_WG; -1 on a HIP error (message set)
int lvl_engine_max_resident(int dtype, int device);
size_t lvl_engine_lds_bytes();

}  // namespace dmme
