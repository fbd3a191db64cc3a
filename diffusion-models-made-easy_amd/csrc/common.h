// Shared device/host helpers for libdmme_hip (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "dmme_hip.h"

namespace dmme {

void set_error(const char* fmt, ...) __attribute__((format(printf, 1, 2)));

// Experiment / test-only switches live behind ONE environment variable: DMME_DEBUG_ROUTE="key[=int],key[=int],..." (a bare key
// reads as 1).  Returns `dflt` when the key is absent.  The product's own A/B switches keep their DMME_NO_* names.
int debug_route(const char* key, int dflt = 0);

#define DMME_CHECK_HIP(expr)                                                                          \
    do {                                                                                              \
        hipError_t e_ = (expr);                                                                       \
        if (e_ != hipSuccess) {                                                                       \
            ::dmme::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
            return DMME_ERR_HIP;                                                                      \
        }                                                                                             \
    } while (0)

#define DMME_REQUIRE(cond, code, ...)        \
    do {                                     \
        if (!(cond)) {                       \
            ::dmme::set_error(__VA_ARGS__);  \
            return (code);                   \
        }                                    \
    } while (0)

// launch check: catches configuration errors without synchronising
#define DMME_CHECK_LAUNCH() DMME_CHECK_HIP(hipGetLastError())

typedef __bf16 bf16;
typedef _Float16 f16;  // IEEE half: the reference's own AMP dtype (configs/ddpm/cifar10.yaml:53 `precision: 16`), 8x finer than bf16

template <typename T>
struct dtype_of;
template <>
struct dtype_of<float> {
    static constexpr int value = DMME_F32;
};
template <>
struct dtype_of<bf16> {
    static constexpr int value = DMME_BF16;
};
template <>
struct dtype_of<f16> {
    static constexpr int value = DMME_F16;
};

__host__ __device__ inline size_t dtype_size(int dt) { return dt == DMME_BF16 || dt == DMME_F16 ? 2 : 4; }
__host__ __device__ inline bool is16(int dt) { return dt == DMME_BF16 || dt == DMME_F16; }  // the two 16-bit operand types share every kernel

__device__ __forceinline__ float to_f(float v) { return v; }
__device__ __forceinline__ float to_f(bf16 v) { return (float)v; }
__device__ __forceinline__ float to_f(f16 v) { return (float)v; }
template <typename T>
__device__ __forceinline__ T from_f(float v);
template <>
__device__ __forceinline__ float from_f<float>(float v) {
    return v;
}
template <>
__device__ __forceinline__ bf16 from_f<bf16>(float v) {
    return (bf16)v;  // v_cvt_pk_bf16_f32: round-to-nearest-even, NaN preserving
}
template <>
__device__ __forceinline__ f16 from_f<f16>(float v) {
    return (f16)v;  // v_cvt_f16_f32: round-to-nearest-even
}

__device__ __forceinline__ float silu_f(float x) { return x / (1.0f + expf(-x)); }
// hardware-transcendental SiLU for the bf16 path: v_exp_f32 + v_rcp_f32 (~1 ulp each), far
// inside the 2^-9 rounding the value gets when it is stored as bf16
__device__ __forceinline__ float silu_fast(float x) {
    return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x));
}

// Wave-wide reductions without LDS traffic.  `__shfl_xor` is a ds_bpermute (an LDS-pipe round trip per step: six dependent ones per
// reduction, ~100 cycles each); here the four steps inside a 16-lane row are DPP operands of the add itself (quad_perm swaps, then
// the mirrors: the lanes they pair already hold identical partial results, so they act as lane ^ 4 / lane ^ 8), and the two steps
// across rows are the gfx950 row / half-wave register swaps.  Every lane ends up with the same bits.
// (The swaps are inline asm: this hipcc folds several calls of __builtin_amdgcn_permlane32_swap with different operands into one;
// `s_nop 1`: a VALU result needs two wait states before a swap reads it, and the hazard recogniser does not see inside asm.)
__device__ __forceinline__ void permlane16_swap(float& x, float& y) {  // odd 16-lane rows of x <-> even rows of y
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(x), "+v"(y));
#endif
}
__device__ __forceinline__ void permlane32_swap(float& x, float& y) {  // lanes 32-63 of x <-> lanes 0-31 of y
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(x), "+v"(y));
#endif
}
#define DMME_DPP_F(v, ctrl) __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl, 0xf, 0xf, true))
__device__ __forceinline__ float row16_sum(float v) {  // over the lane's 16-lane row
    v += DMME_DPP_F(v, 0xB1);   // quad_perm [1,0,3,2]
    v += DMME_DPP_F(v, 0x4E);   // quad_perm [2,3,0,1]
    v += DMME_DPP_F(v, 0x141);  // row_half_mirror
    v += DMME_DPP_F(v, 0x140);  // row_mirror
    return v;
}
__device__ __forceinline__ float half_sum(float v) {  // over the lane's 32-lane half wave
    v = row16_sum(v);
    float a = v, b = v;
    permlane16_swap(a, b);  // a: the even row's value in both rows of a pair, b: the odd row's
    return a + b;
}
// (the general-purpose reductions keep the xor butterfly of ds_bpermute steps: their callers are HBM-bound kernels, and the bf16
// network's max-error statistics are calibrated on this summation order)
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// block-wide sum for blockDim.x <= 1024 (multiple of 64); `red` needs 16 floats of LDS
__device__ __forceinline__ float block_sum(float v, float* red) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    v = wave_sum(v);
    __syncthreads();
    if (lane == 0) red[w] = v;
    __syncthreads();
    float r = 0.f;
    for (int i = 0; i < nw; ++i) r += red[i];
    return r;
}
__device__ __forceinline__ float block_max(float v, float* red) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    v = wave_max(v);
    __syncthreads();
    if (lane == 0) red[w] = v;
    __syncthreads();
    float r = red[0];
    for (int i = 1; i < nw; ++i) r = fmaxf(r, red[i]);
    return r;
}

// Device-side description of one convolution launch (generic and MFMA kernels share it).
// A GroupNorm whose statistics the PRODUCING conv completes in its own epilogue (tiles of whole images: 8x8, 4x4 maps): the conv
// writes the norm's per-(n, c) scale / shift rows and {mean, rstd} itself - no finalize launch, no separate norm kernel.
struct GnOut {
    const float* gamma;  // [C] of the consuming norm
    const float* beta;
    float* scale;        // [N][C]
    float* shift;
    float* mean_rstd;    // [N][groups][2]
    int C, cg, c_off;    // the norm's width and group size; channel c of this conv's output is channel c_off + c of the norm
};

// A GroupNorm whose statistics arrive as the producers' per-tile partials and are merged by the CONSUMING conv itself (its parameter
// fill; wave-specialised 3x3 kernel) instead of by a finalize launch.  Same layout as launch_gn_finalize_parts' arguments: source i has
// t_i partials per image of cnt_i elements each, [N][t_i][groups][2] {mean, M2}; a consumer group is nf whole producer groups of one source.
struct GnIn {
    const float* p1;
    const float* p2;
    int t1, cnt1, C1, t2, cnt2, C2, groups;
    const float* gamma;
    const float* beta;
    float eps;
    float* mean_rstd;  // [N][groups][2], written (with scale / shift) by the workgroup holding the image's first tile
    // scale-shift conditioning folded in (iddpm.ResBlock; null: none): scale' = scale (1 + t_scale), shift' = shift (1 + t_scale) + t_shift,
    // rows [nt][t_ld] of the batched time projection (nt == 1: one row for the batch)
    const float* t_shift;
    const float* t_scale;
    int t_ld, nt;
};

struct ConvArgs {
    const void* src1;
    const void* src2;
    const void* w;      // [Cout][taps][Cin] T
    const float* bias;  // [Cout] or null
    const float* scale; // [N][Cin] or null
    const float* shift;
    const float* dmask; // [N][Cin] or null
    const float* tproj; // [nt][tproj_ld] or null
    const void* res1;   // NHWC T [.., R1]
    const void* res2;   // NHWC T [.., Cout-R1] or null
    void* dst;
    int N, Hin, Win, C1, C2;
    int up, stride, taps;  // up: 0 none, 1 nearest 2x (nn.Upsample), 2 zero-insertion 2x (data gradient of a stride-2 conv)
    int Hout, Wout, Cout;
    int pro_silu, out_silu;
    int nt, tproj_ld, R1;
    int in_nchw, out_nchw;
    // fused GroupNorm statistics of the OUTPUT tensor (null: off): per (image, spatial tile, group) {mean, M2}
    // partials, [N][gn_tiles][Cout/gn_cg][2], written by the LDS-staged epilogue; merged by gn_finalize_parts
    float* gn_part;
    int gn_cg, gn_tiles;
    // diagnostic (null: off): per-phase cycle stamps of a few workgroups, [slot][64] (dmme_debug_set_stamps)
    long long* stamps;
    // split-K scratch (null: off): fp32 partial outputs [ksplit][N*Hout*Wout][Cout] of layers with too few output tiles to fill
    // the chip (4x4 maps: 128 workgroups, each streaming every filter of its 64 couts); summed by conv_splitk_finish_kernel,
    // which also applies bias / time row / residual.  splitk_cap: capacity in floats.
    float* splitk;
    int64_t splitk_cap;
    // accurate mode (fp32 tensors only): every product as three bf16 MFMA passes on hi/lo splits instead of the fp32 MFMA
    int x3;
    // precision="fp16r32" (a 16-bit plan whose full-resolution level lives in fp32): this conv reads / writes fp32 tensors and runs every
    // product as three fp16 MFMA passes over hi / lo splits of both operands (filter packed [Cout][taps][Cin / 32][hi 32 | lo 32] halves).
    // 0: no; 1: the wave-specialised 3x3 kernel's split form; 2: the same with a 16-bit SOURCE tensor (fp32 residual / output);
    // 3: the thin output conv's split form (fp32 source, NCHW fp32 out)
    // 4: the 1x1 convs of that level (conv1x1_pipe.hip: conv1x1_split_kernel)
    int mix;
    // mix 1 / 2 only: run two passes (hi.hi + lo.hi - activations exact to 2^-22, the FILTER rounded to half) instead of three.  The
    // filter's rounding of a conv costs what it costs in precision="fp16" (which layers can afford it)
    int mix2;
    // 16-bit tensors are IEEE half (precision="fp16") instead of bf16: for the launchers that take no dtype argument
    int f16;
    // pipelined 3x3 kernel, 64-cout bf16 tiles: filter tiles by LDS-DMA into a second buffer instead of through registers
    int dma_b;
    // norms finished by this conv's epilogue (n_gno = 0: none) and, for norm gno[act_k], the consumer's pre-activated input
    // act[n][p][act_C] = T(silu?(y * scale + shift) * mask) (null: not written)
    GnOut gno[2];
    int n_gno;
    float gn_eps;
    void* act;
    int act_k, act_silu;
    const float* act_dmask;  // [N][gno[act_k].C] or null
    GnIn gni;                 // has_gni: scale / shift come from gni's partials (the arrays are outputs of this conv, not inputs)
    int has_gni;
    // second K segment (r_w null: none; the wave-specialised 3x3 kernel's 256-pixel form only): the ResBlock's 1x1 residual conv
    // (models/ddpm.py:108-111,131) over the block's RAW input [.., r_C1] ++ [.., r_C2], filter [Cout][r_C1 + r_C2], bias [Cout] -
    // accumulated into the same output tile, so `h + residual(x)` needs no residual tensor (res1 is null then)
    const void* r_src1;
    const void* r_src2;
    const void* r_w;
    const float* r_bias;
    int r_C1, r_C2;
};

// ---- kernel launchers (defined in the .hip files) ------------------------------------
int launch_conv_generic(int dtype, const ConvArgs& a, hipStream_t s);
int launch_mfma_valu(int mode, int iters, int blocks, float* sink, hipStream_t s);
int launch_issue_probe(int kind, int n_inner, int iters, int flags, int blocks, long long* sink, const void* src, hipStream_t s);
int launch_l2_stream(const void* buf, int64_t bytes, int iters, int mode, int depth, int blocks, unsigned* sink, hipStream_t s);
const char* conv_generic_kernel_name(const ConvArgs& a);
bool conv_in_stats_query(int dtype, const ConvArgs& a, int cg, int* tiles, int* px);
// returns DMME_ERR_UNSUPPORTED (without setting the error) when the shape is outside
// the MFMA kernel's domain, so callers can fall back to the generic kernel.
bool conv_mfma_supported(int dtype, const ConvArgs& a);
int launch_conv_mfma(int dtype, const ConvArgs& a, hipStream_t s);
// "conv_mfma_kernel<bf16,9,128,128>" for the variant launch_conv_mfma would pick
void conv_mfma_label(int dtype, const ConvArgs& a, char* buf, int cap);
// software-pipelined 3x3 stride-1 variant (conv_pipe.hip); preferred when it applies
bool conv_pipe_supported(int dtype, const ConvArgs& a);
// can the wave-specialised kernel take this conv WITH the residual segment described in a.r_* (ConvArgs::r_w)?
bool conv_pipe_rseg_supported(int dtype, const ConvArgs& a);
// would this conv run on the wave-specialised kernel, and can that kernel merge its norm's partials itself (ConvArgs::gni)?
bool conv_gn_in_query(int dtype, const ConvArgs& a);
int launch_conv_pipe(int dtype, const ConvArgs& a, hipStream_t s);
void conv_pipe_label(int dtype, const ConvArgs& a, char* buf, int cap);
// the 3- / 6-channel output conv as one 27- / 54-column GEMM + a 9-term gather (conv_thin.hip)
bool conv_out_thin_supported(int dtype, const ConvArgs& a);
int launch_conv_out_thin(const ConvArgs& a, hipStream_t s);
// K-split-over-waves 3x3 kernel for layers with few output pixels (conv_kw.hip); launch_conv_pipe dispatches to it
struct ConvTile;
bool conv_kw_pick(int dtype, const ConvArgs& a, ConvTile& g, int* ni, int* ring, int* bm);
int launch_conv_kw(int dtype, const ConvArgs& a, const ConvTile& g, int NI, int ring, int BM, int ksplit, hipStream_t s);
// will the kernel that runs this conv finish the norms consuming its output (ConvArgs::n_gno set; cg[k]: their group sizes)?
bool conv_gn_direct_query(int dtype, const ConvArgs& a, const int* cg, int n);
bool conv_gn_direct_ws_query(int dtype, const ConvArgs& a, const int* cg, int n);  // a.gn_cg: the output tensor's own group size
// software-pipelined 1x1 variant (conv1x1_pipe.hip); preferred for taps == 1
bool conv1x1_pipe_supported(int dtype, const ConvArgs& a);
int launch_conv1x1_pipe(int dtype, const ConvArgs& a, hipStream_t s);
void conv1x1_pipe_label(int dtype, const ConvArgs& a, char* buf, int cap);
bool conv1x1_stats_query(int dtype, const ConvArgs& a, int cg, int* tiles, int* px);
bool conv1x1_pipe_gn_in_ok(int dtype, const ConvArgs& a);
// activation-stationary 1x1 variant for K <= 256 (conv1x1_as.hip); launch_conv1x1_pipe dispatches to it
bool conv1x1_as_supported(int dtype, const ConvArgs& a);
bool conv1x1_as_stats_query(int dtype, const ConvArgs& a, int cg, int* tiles, int* px);
int launch_conv1x1_as(const ConvArgs& a, hipStream_t s);
// can the kernel that would run this conv also emit GroupNorm partials of its output (group size cg)?
// on success: tiles = spatial tiles per image, px = pixels per tile (the partial's element count is px*cg)
bool conv_stats_query(int dtype, const ConvArgs& a, int cg, int* tiles, int* px);
// merge producer partials of one or two (concatenated) tensors into scale/shift (+ mean/rstd)
int launch_gn_finalize_parts(const float* part1, int tiles1, int cnt1, int C1, const float* part2, int tiles2, int cnt2, int C2, int N, int groups,
                             const float* gamma, const float* beta, float eps, float* scale, float* shift, float* mean_rstd, const float* t_shift,
                             const float* t_scale, int t_ld, int nt, hipStream_t s);

// mean_rstd (nullable): [N][groups][2] = {mean, rstd}, kept for the backward pass
int launch_gn_generic(int dtype, const void* src1, const void* src2, int N, int HW, int C1, int C2, int groups,
                      const float* gamma, const float* beta, float eps, float* scale, float* shift, float* mean_rstd,
                      hipStream_t s);
bool gn_fast_supported(int dtype, int N, int HW, int C1, int C2, int groups);
// partial: scratch of gn_fast_scratch_floats(...) floats
size_t gn_fast_scratch_floats(int N, int HW, int C, int groups);
int launch_gn_fast(int dtype, const void* src1, const void* src2, int N, int HW, int C1, int C2, int groups,
                   const float* gamma, const float* beta, float eps, float* scale, float* shift, float* mean_rstd,
                   float* partial, hipStream_t s, void* act = nullptr, int act_silu = 0, const float* dmask = nullptr);
// the one-workgroup-per-image GroupNorm (maps of at most 64 pixels) can also write the consumer's pre-activated input
bool gn_small_act_supported(int dtype, int HW, int C1, int C2, int groups);

int launch_attn_generic(int dtype, const void* qkv, int N, int S, int C, void* out, hipStream_t s);
// multi-head attention as the reference ships it (models/iddpm.py:35-47): head h owns qkv channels [h*3d, (h+1)*3d) split
// (q | k | v), K scaled by C^-0.5, and row b*heads + h of the result lands at batch (b*heads + h) % N, head (b*heads + h) / N
int launch_attn_heads(int dtype, const void* qkv, int N, int S, int C, int heads, void* out, hipStream_t s);
int launch_attn_heads_bwd(int dtype, const void* qkv, const void* dO, int N, int S, int C, int heads, float* P, float* dS, void* dqkv, hipStream_t s);
// scale-shift conditioning folded into a GroupNorm's per-(n, c) scale / shift: sc *= 1 + t_scale, sh = sh * (1 + t_scale) + t_shift
int launch_gn_modulate(float* scale, float* shift, const float* t_shift, const float* t_scale, int ld, int nt, int N, int C, hipStream_t s);
// accurate mode (fp32 tensors, three-pass bf16 MFMA products): fused attention, head widths 64 / 128 / 256
bool attn_x3_supported(int N, int S, int C, int heads);
int launch_attn_x3(const void* qkv, int N, int S, int C, int heads, void* out, hipStream_t s);
bool attn_mfma_supported(int dtype, int N, int S, int C);
bool attn_full_takes(int S, int D);  // the whole-row softmax kernel serves this (keys, head width): launch_attn_heads_mfma's dispatch
// lse (nullable): [N][S] log2-domain log-sum-exp of the scaled scores, kept for the backward pass
int launch_attn_mfma(int dtype, const void* qkv, int N, int S, int C, void* out, float* lse, hipStream_t s);
// the block's proj 1x1 conv + residual inside the whole-row attention launch (attn_mfma.hip, AttnProj); ctx: the context tensor, nullptr
// where no backward pass will read it; gn_part / gn_tiles / gn_cg: ConvArgs' fields of the proj conv (one partial per 32 pixels)
bool attn_proj_fusable(int dtype, int N, int S, int C, int gn_cg, int gn_tiles);
int launch_attn_proj(int dtype, const void* qkv, int N, int S, int C, void* ctx, float* lse, const void* w, const float* bias, const void* res, void* dst,
                     float* gn_part, int gn_tiles, int gn_cg, hipStream_t s, long long* stamps = nullptr);
bool attn_heads_mfma_supported(int dtype, int N, int S, int C, int heads);
int launch_attn_heads_mfma(int dtype, const void* qkv, int N, int S, int C, int heads, void* out, float* lse, hipStream_t s);
int launch_attn_heads_bwd_mfma(int dtype, const void* qkv, const void* O, const void* dO, const float* lse, int N, int S, int C, int heads, void* P,
                               void* dS, void* dqkv, hipStream_t s);
bool attn_bwd_mfma_supported(int dtype, int N, int S, int C);
int launch_attn_bwd_mfma(int dtype, const void* qkv, const void* O, const void* dO, const float* lse, int N, int S, int C, void* P, void* dS,
                         void* dqkv, hipStream_t s);

int launch_time_sinusoid(const int64_t* t, int nt, const float* freqs, int half, float* out, hipStream_t s);
// out[nt][Nout] = act(in[nt][K] . W[Nout][K]^T + b); in/out fp32, W in dtype
int launch_linear_wave(int dtype, const float* in, int nt, int K, const void* W, const float* bias, int Nout,
                       int out_silu, float* out, hipStream_t s);

int launch_nchw_to_nhwc(int dtype, const float* src, int N, int C, int HW, void* dst, hipStream_t s);
int launch_nhwc_to_nchw(int dtype, const void* src, int N, int C, int HW, float* dst, hipStream_t s);
int launch_pack_weight(int dtype, const float* src, int Cout, int Cin, int taps, void* dst, hipStream_t s);
// fp32 -> 16-bit copy of a tensor (numel % 8 == 0): where a conv of a mixed plan's 16-bit levels reads a tensor of its fp32 level
int launch_cast_f32_to_16(int dtype, const float* src, int64_t numel, void* dst, hipStream_t s);

struct PackItem {  // one chunk of the table-driven parameter re-pack
    int64_t src_off;   // element offset in the fp32 reference-layout flat buffer
    int64_t dst_off;   // BYTE offset in the packed buffer
    int32_t cout, cin, taps;  // tensor geometry (cin*taps = row length)
    int32_t row0, rows;       // rows (couts) [row0, row0+rows) of this tensor handled by this item
    int32_t ci0, nci;         // cin range [ci0, ci0+nci) handled by this item (re-pack items; rows*nci*taps <= 8192)
    int32_t as_f32;           // 1: keep fp32 (biases, gammas, freqs); 0: convert to dtype;
                              // 2: transposed + tap-flipped copy [cin][taps-1-tap][cout] in dtype (data-gradient weights)
                              // 3: [cout][tap][cin] like 0, but kept fp32 whatever the plan's dtype (fp32-routed convs of a mixed plan)
                              // 4: [cout][tap][cin / 32][hi 32 | lo 32] IEEE halves, hi = f16(w), lo = f16(w - hi) (ConvArgs::mix)
};
int launch_pack_table(int dtype, const PackItem* items_dev, int n_items, const float* ref_flat, void* packed,
                      hipStream_t s);

int launch_randn(float* out, int64_t numel, uint64_t seed, uint64_t offset, hipStream_t s);
int launch_dropmask(float* out, int64_t numel, float p, uint64_t seed, uint64_t offset, hipStream_t s);
int launch_q_sample(const float* x0, const float* z, const float* sqrt_abar, const float* sqrt_1m_abar,
                    const int64_t* t, int B, int64_t chw, float* x_t, float* target, hipStream_t s);
int launch_ddpm_step(float* x, const float* eps, const float* z, float c1, float c2, float sigma, int add_noise,
                     int64_t numel, hipStream_t s);
int launch_ddim_step(float* x, const float* eps, float s1, float s2, int64_t numel, hipStream_t s);
int launch_chain_set(void* state, int64_t i, const int64_t* t_table, uint64_t seed, uint64_t offset, hipStream_t s);
int launch_chain_update(int kind, float* x, const float* out, const float* coef, const int64_t* t_table, void* state, int B, int64_t chw,
                        hipStream_t s);
int launch_image_batch(const uint8_t* data, const int64_t* idx, const uint8_t* flip, int B, int C, int H, int W, float* out, hipStream_t s);
int launch_iddpm_step(float* x, const float* out, const float* z, float c1, float c2, float log_beta, float log_beta_tilde, int add_noise,
                      int B, int64_t chw, hipStream_t s);
int launch_iddpm_loss(const float* out, const float* x_t, const float* x_0, const float* target, const int64_t* t, const float* coef, int B,
                      int64_t chw, float w_simple, float w_vlb, float* loss, float* d_out, float gscale, float* scratch, hipStream_t s);
int launch_mse(const float* eps, const float* target, int64_t numel, float* loss, float* d_eps, float gscale,
               float* scratch, hipStream_t s);

// ---- backward (kernels_bwd.hip) ----
int launch_wgrad_generic(int dtype, const ConvArgs& a, const void* dY, float* dW, hipStream_t s);
// first / last layer weight gradients (few input or output channels)
bool wgrad_small_supported(int dtype, const ConvArgs& a);
int launch_wgrad_small(int dtype, const ConvArgs& a, const void* dY, float* dW, hipStream_t s);
// MFMA weight gradient (wgrad_mfma.hip): atomically accumulates into a zero-initialised packed-layout image
bool wgrad_mfma_supported(int dtype, const ConvArgs& a);
int launch_wgrad_mfma(int dtype, const ConvArgs& a, const void* dY, float* dWp, hipStream_t s);
struct PackItem;
struct ConvTile {  // host-computed geometry, passed by value
    int TW, TH, TN;       // output tile: TN images x TH x TW pixels (product = BM)
    int HH, HWd;          // halo extent in (virtual) input space
    int tiles_x, tiles_y; // tiles per image
    int tiles_m, tiles_n;
    int a_rows;           // TN*HH*HWd
    unsigned magic_px, magic_w;  // ceil(2^32 / (HH*HWd)), ceil(2^32 / HWd): exact x/d by __umulhi for x*d < 2^32
};

// grouped (deferred) 3x3 weight gradients: plan-time tables, one launch per backward (wgrad_mfma.hip)
struct WgLayer {
    int64_t src1_off, src2_off;    // bytes into the forward workspace (src2: -1 none)
    int64_t scale_off, shift_off;  // bytes into the forward workspace (-1: no GroupNorm prologue)
    int64_t dmask_off;             // floats into the drop-mask buffer (-1: none)
    int64_t dy_off;                // bytes into the backward workspace: gradient of the conv output
    int64_t dw_off;                // floats into the packed weight-gradient image
    int64_t act_off;               // the conv's pre-activated input [N][Hin][Win][C1 + C2] (-1: none - src1 / src2 + prologue), bytes into
    int act_bws;                   //   the backward workspace (written by the GroupNorm backward) or the forward one (written by the forward)
    int N, Hin, Win, C1, C2, up, Hout, Wout, Cout, pro_silu;
    int shTW, shTH;
    int ks_off[4], half_off;       // LDS byte offsets of tile pixels 16*ks and 4 relative to pixel 0 (input tile rows)
    int ks_row[4], half_row;       // the same in halo rows (the DMA-fed kernel's 128-byte swizzled rows)
    ConvTile g;
};
struct WgJob {
    int layer, cot, cit, tile0, ntiles;
};
// fills the geometry fields of L when the conv qualifies (offsets are the caller's)
// co_tile / ci_tile: the job tile of the kernel that will run it (64x64 for 3x3, 128x128 for 1x1)
bool wgrad_group_layer(int dtype, const ConvArgs& a, WgLayer& L, int* co_tile, int* ci_tile);
// dma: every layer of the table has a single prologue-free operand tensor (WgLayer::act_off, or one source and no norm): both tiles
// arrive by LDS-DMA; zero_page: >= 16 zero bytes in device memory (the source of padding rows)
int launch_wgrad_group(int dtype, int taps, const WgLayer* layers_dev, const WgJob* jobs_dev, int njobs, const void* ws, const void* bws,
                       const float* drop_masks, float* wimage, hipStream_t s, int dma = 0, const void* zero_page = nullptr);
int launch_wgrad_unpack(const PackItem* items_dev, int n_items, const float* image, float* grad_flat, hipStream_t s);
int launch_colsum(int dtype, const void* dY, int N, int HW, int C, float* rowsum, float* dbias, float* dtproj, int ld, int nt,
                  hipStream_t s);
// Scale-shift conditioning of a GroupNorm in the backward pass (iddpm.ResBlock, models/iddpm.py:117-118):
// y = GN(x) * (1 + t_scale[r][c]) + t_shift[r][c], r = n (nt == N) or 0 (nt == 1).  The GroupNorm backward then runs with the
// effective gamma_c * (1 + t_scale) and also emits d t_shift = A, d t_scale = gamma_c B + beta_c A (A = sum du, B = sum du xhat).
// t_scale == nullptr: plain GroupNorm.  d_* rows are assigned when nt == N and atomically accumulated (zeroed buffer) when nt == 1.
struct GnMod {
    const float* t_scale = nullptr;
    const float* beta = nullptr;
    float* d_shift = nullptr;
    float* d_scale = nullptr;
    int ld = 0, nt = 0;
#if defined(__HIPCC__)
    __device__ __forceinline__ float mul(int n, int c) const { return t_scale ? 1.0f + t_scale[(int64_t)(nt == 1 ? 0 : n) * ld + c] : 1.0f; }
    __device__ __forceinline__ void emit(int n, int c, float A, float B, float gamma_c) const {
        if (!t_scale) return;
        const float ds = fmaf(gamma_c, B, beta[c] * A);
        if (nt == 1) {
            atomicAdd(&d_shift[c], A);
            atomicAdd(&d_scale[c], ds);
        } else {
            d_shift[(int64_t)n * ld + c] = A;
            d_scale[(int64_t)n * ld + c] = ds;
        }
    }
#endif
};
int launch_gn_bwd_generic(int dtype, const void* dv, const void* x1, const void* x2, int N, int HW, int C1, int C2, int groups,
                          const float* gamma, const float* mean_rstd, const float* scale, const float* shift, const float* dmask,
                          int pro_silu, void* dx1, void* dx2, int acc1, int acc2, float* dgamma, float* dbeta, GnMod mod, hipStream_t s);
// coalesced vector versions (bwd_fast.hip)
bool colsum_fast_supported(int dtype, int HW, int C);
// rowsum must be zero on entry (the backward pass clears all its accumulation scratch with one memset)
// deferred bias / time-projection reductions of many convs (plan-time table, one launch per backward)
struct BiasJob {
    int64_t rowsum_off;  // bytes into the backward workspace: this conv's [N][C] column sums
    int64_t dbias_off;   // floats into grad_flat
    int C, cblock, tcol; // channels, 32-channel block of this job, column of the conv's time-projection rows (-1: none)
};
int launch_bias_tproj_group(const BiasJob* jobs_dev, int njobs, const void* bws, float* grad_flat, float* dtproj, int N, int ld, int nt, hipStream_t s);
// the [N][C] column sums of dY of MANY convs in one launch, deferred to the end of backward like the weight gradients (every dY is
// still in its gradient buffer): a job is one pixel chunk of one conv, grid.y walks the images
struct ColJob {
    int64_t dy_off;      // bytes into the backward workspace
    int64_t rowsum_off;  // bytes into the backward workspace (zeroed region)
    int HW, C, chunk, chunk_px, ppw, pad;
};
// fills the job fields that depend on the geometry; returns the number of chunks (jobs) of this conv, 0: unsupported
int colsum_group_chunks(int dtype, int HW, int C, int* chunk_px, int* ppw);
int launch_colsum_group(int dtype, const ColJob* jobs_dev, int njobs, void* bws, int N, hipStream_t s);
// dbias == dtproj == nullptr: only the [N][C] column sums (the caller reduces them later with launch_bias_tproj_group)
int launch_colsum_fast(int dtype, const void* dY, int N, int HW, int C, float* rowsum, float* dbias, float* dtproj, int ld, int nt,
                       hipStream_t s);
bool gn_bwd_fast_supported(int dtype, int HW, int C1, int C2);
int gn_bwd_fast_chunks(int dtype, int HW, int C);
int launch_gn_bwd_fast(int dtype, const void* dv, const void* x1, const void* x2, int N, int HW, int C1, int C2, int groups,
                       const float* gamma, const float* mean_rstd, const float* scale, const float* shift, const float* dmask,
                       int pro_silu, void* dx1, void* dx2, int acc1, int acc2, float* dgamma, float* dbeta, float* AB_zeroed,
                       float* S_scratch, GnMod mod, hipStream_t s, void* act = nullptr, float* rows = nullptr, const void* extra = nullptr);
bool gn_bwd_rows_supported(int dtype, int HW, int C1, int C2, int groups, bool has_mod);
bool grad_acc_fast_supported(int dtype, int C1, int C2, int pool);
int launch_grad_acc_fast(int dtype, const void* src, void* d1, void* d2, int C1, int C2, int acc1, int acc2, int64_t npix, hipStream_t s);
int launch_grad_acc(int dtype, const void* src, void* d1, void* d2, int C1, int C2, int acc1, int acc2, int pool, int N, int H, int W,
                    hipStream_t s);
int launch_attn_bwd_generic(int dtype, const void* qkv, const void* dO, int N, int S, int C, float* P, float* dS, void* dqkv, hipStream_t s);
int launch_lin_dinput(int dtype, const float* dY, const void* W, int R, int O, int K, float* dX, hipStream_t s);
int launch_lin_dweight(const float* dY, const float* X, int R, int O, int K, float* dW, float* dB, hipStream_t s);
int launch_silu_bwd(float* dy, const float* z, int n, hipStream_t s);
// small fp32 GEMMs (small_gemm.hip): mode 0 NT (forward linear), 1 NN (input gradient), 2 TN (weight gradient, accumulating)
int launch_transpose(int dtype, const void* src, int R, int C, void* dst, hipStream_t s);  // dst[c][r] = src[r][c], elements of the compute dtype
int launch_small_gemm(int dtype, int mode, const float* A, int lda, const void* B, int ldb, int M, int N, int K, const float* bias, int out_silu,
                      float* C, int ldc, hipStream_t s, float* Cpre = nullptr);
int launch_nsum(const float* Mx, int N, int C, int64_t stride, int estride, float* out, hipStream_t s);
// one launch for a set of Linear layers: per-64-row-tile (gemm) / per-32-column (sum) output bases, in floats
int launch_small_gemm_tn_tiled(const float* A, int lda, const float* B, int ldb, int M, int N, int K, float* C, int ldc,
                               const int64_t* mtile_off, hipStream_t s);
int launch_nsum_tiled(const float* Mx, int N, int C, int64_t stride, int estride, float* out, const int64_t* ctile_off, hipStream_t s);
int launch_grad_norm(const float* g, int64_t n, float* norm_out, float* scratch, hipStream_t s);
int launch_adam(float* p, const float* g, float* m, float* v, float* ema, int64_t n, float lr, float b1, float b2, float eps, int step,
                const float* norm, float max_norm, float ema_decay, float grad_scale, hipStream_t s);
int launch_amp_init(float* amp, float init_scale, hipStream_t s);
int launch_amp_scale(float* d, int64_t n, const float* amp, hipStream_t s);
int launch_adam_amp(float* p, const float* g, float* m, float* v, float* ema, int64_t n, float lr, float b1, float b2, float eps, const float* norm,
                    float max_norm, float ema_decay, float grad_scale, float* amp, float growth, float backoff, int interval, hipStream_t s);
int launch_grad_pack_bf16(const float* g, int64_t n, void* dst, int64_t n_pad, hipStream_t s);
int launch_shard_reduce_bf16(const void* recv, int world, int64_t per, float scale, void* out, hipStream_t s);
int launch_grad_unpack_bf16(const void* src, int64_t n, float* g, hipStream_t s);

}  // namespace dmme
