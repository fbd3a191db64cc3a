// Weight gradient of the 3x3 / 1x1 convolutions on the matrix cores.
//
//   dW[co][tap][ci] = sum over pixels p of  dY[p][co] * v[p*stride + tap][ci],   v = act(x)
//
// i.e. a GEMM whose reduction dimension is the PIXEL index: M = Cout, N = Cin, K = N*H*W.
// Both operands are NHWC, so K is the strided dimension of both; the fragments are therefore
// read TRANSPOSED out of row-major [pixel][channel] LDS tiles:
//   bf16: ds_read_b64_tr_b16 (4 pixels x 16 channels per 16-lane group, two reads per
//         32x32x16 fragment) for dY^T (A operand) and for the shifted activation tile (B);
//   fp32: v_mfma_f32_32x32x2_f32 takes one element per lane with the channel on the lane,
//         so plain ds_read_b32 of the same tiles already is the transposed view.
// One workgroup = 128 couts x 64 cins x one kernel ROW (3 taps: the three kw shifts reuse the
// dY fragments) x a strided subset of the 64-pixel tiles; the activation halo rows of that
// kernel row are staged with the forward prologue (GroupNorm affine + SiLU + Dropout2d mask)
// applied on the way, exactly like the forward kernel.  Partial sums go out with fp32
// atomics into a packed [co][tap][ci] image (ci contiguous across lanes: full-rate
// atomics); a small kernel folds it into the reference-layout gradient buffer.
#include <stdio.h>
#include <stdlib.h>

#include "conv_common.h"

namespace dmme {

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

constexpr int WG_CO = 128, WG_CI = 64, WG_PX = 64;

template <typename T>
struct WgGeom {
    // LDS row pitches chosen so the 4 rows of a transposed read land on disjoint bank ranges
    static constexpr int DY_PITCH = sizeof(T) == 2 ? 320 : 512;  // 128 couts
    static constexpr int V_PITCH = sizeof(T) == 2 ? 192 : 256;   // 64 cins
};

template <typename T, int KW>
__global__ void __launch_bounds__(256) wgrad_mfma_kernel(ConvArgs a, ConvTile g, const T* __restrict__ dY, float* __restrict__ dWp,
                                                         int nsplit, int shTW, int shTH) {
    constexpr int EPV = Frag<T>::EPV;
    constexpr int DYP = WgGeom<T>::DY_PITCH, VP = WgGeom<T>::V_PITCH;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    char* ldsY = lds;                 // [64 px][128 co]
    char* ldsV = lds + WG_PX * DYP;   // [TN*TH rows x HWd px][64 ci]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wco = (wave >> 1) * 64, wci = (wave & 1) * 32;
    const int r = lane & 31, h = lane >> 5;
    const int Cin = a.C1 + a.C2;
    const int n_ci = Cin / WG_CI, n_co = (a.Cout + WG_CO - 1) / WG_CO, n_kh = KW == 3 ? 3 : 1;
    int b = blockIdx.x;
    const int split = b % nsplit; b /= nsplit;
    const int kh = b % n_kh; b /= n_kh;
    const int cit = b % n_ci; b /= n_ci;
    const int cot = b;
    (void)n_co;
    const int co0 = cot * WG_CO, ci0 = cit * WG_CI;
    const int PAD = KW == 3 ? 1 : 0;
    const int Hv = a.up ? 2 * a.Hin : a.Hin, Wv = a.up ? 2 * a.Win : a.Win;
    const int mTW = (1 << shTW) - 1, mTH = (1 << shTH) - 1;
    const int v_rows = g.TN * g.TH * g.HWd;  // halo pixels of ONE kernel row
    const bool second = ci0 >= a.C1;
    const T* sbase = second ? (const T*)a.src2 : (const T*)a.src1;
    const int Cs = second ? a.C2 : a.C1, cs0 = second ? ci0 - a.C1 : ci0;

    f32x16 acc[KW][2];
#pragma unroll
    for (int k = 0; k < KW; ++k)
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int j = 0; j < 16; ++j) acc[k][mi][j] = 0.f;

    // transposed-read lane roles (bf16): 16-lane group gq, lane i = 4*q + p inside it
    const int tr_q = (lane & 15) >> 2, tr_p = lane & 3, tr_g1 = (lane >> 4) & 1;

    for (int tile = split; tile < g.tiles_m; tile += nsplit) {
        const int tx_blk = tile % g.tiles_x, ty_blk = (tile / g.tiles_x) % g.tiles_y;
        const int n0 = (tile / (g.tiles_x * g.tiles_y)) * g.TN;
        const int oy0 = ty_blk << shTH, ox0 = tx_blk << shTW;
        __syncthreads();
        // ---- stage dY tile: 64 pixels x 128 couts ----
        for (int u = tid; u < WG_PX * (WG_CO / EPV); u += 256) {
            const int m = u / (WG_CO / EPV), cu = u % (WG_CO / EPV);
            const int tx = m & mTW, ty = (m >> shTW) & mTH, tn = m >> (shTW + shTH);
            const int n = n0 + tn, co = co0 + cu * EPV;
            uint4 val = make_uint4(0u, 0u, 0u, 0u);
            if (n < a.N && co < a.Cout)
                val = *reinterpret_cast<const uint4*>(dY + (((int64_t)n * a.Hout + oy0 + ty) * a.Wout + ox0 + tx) * a.Cout + co);
            *reinterpret_cast<uint4*>(ldsY + m * DYP + cu * 16) = val;
        }
        // ---- stage the activation rows of this kernel row: (tn, ty) x HWd pixels x 64 cins, prologue applied ----
        for (int u = tid; u < v_rows * (WG_CI / EPV); u += 256) {
            const int row = u / (WG_CI / EPV), cu = u % (WG_CI / EPV);
            const int hx = row % g.HWd, ty = (row / g.HWd) & mTH, tn = row / (g.HWd << shTH);
            const int n = n0 + tn, iy = (oy0 + ty) * a.stride - PAD + kh, ix = ox0 * a.stride - PAD + hx;
            uint4 val = make_uint4(0u, 0u, 0u, 0u);
            if (n < a.N && iy >= 0 && iy < Hv && ix >= 0 && ix < Wv) {
                const int sy = a.up ? (iy >> 1) : iy, sx = a.up ? (ix >> 1) : ix;
                val = *reinterpret_cast<const uint4*>(sbase + (((int64_t)n * a.Hin + sy) * a.Win + sx) * Cs + cs0 + cu * EPV);
                const int64_t so = (int64_t)n * Cin + ci0 + cu * EPV;
                val = prologue_vec<T>(val, a.scale ? a.scale + so : nullptr, a.scale ? a.shift + so : nullptr,
                                      a.dmask ? a.dmask + so : nullptr, a.pro_silu);
            }
            *reinterpret_cast<uint4*>(ldsV + row * VP + cu * 16) = val;
        }
        __syncthreads();
        if constexpr (sizeof(T) == 2) {
            // ---- bf16: k-steps of 16 pixels; lane half h owns pixels 8h..8h+7 of the step (two 4-pixel blocks) ----
#pragma unroll
            for (int ks = 0; ks < WG_PX / 16; ++ks) {
                s16x8 af[2];
#pragma unroll
                for (int mi = 0; mi < 2; ++mi) {
                    const int colb = (wco + mi * 32 + 16 * tr_g1 + 4 * tr_p) * 2;
                    const char* p0 = ldsY + (16 * ks + 8 * h + tr_q) * DYP + colb;
                    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p0);
                    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p0 + 4 * DYP));
                    af[mi][0] = lo[0]; af[mi][1] = lo[1]; af[mi][2] = lo[2]; af[mi][3] = lo[3];
                    af[mi][4] = hi[0]; af[mi][5] = hi[1]; af[mi][6] = hi[2]; af[mi][7] = hi[3];
                }
                // halo row of the first pixel of each 4-pixel block (blocks never straddle a tile row: 4 | TW)
                int vrow[2];
#pragma unroll
                for (int half = 0; half < 2; ++half) {
                    const int m = 16 * ks + 8 * h + 4 * half + tr_q;
                    const int tx = m & mTW, tyn = m >> shTW;  // tyn = tn*TH + ty
                    vrow[half] = tyn * g.HWd + tx * a.stride;
                }
#pragma unroll
                for (int kw = 0; kw < KW; ++kw) {
                    const int colb = (wci + 16 * tr_g1 + 4 * tr_p) * 2;
                    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(ldsV + (vrow[0] + kw) * VP + colb));
                    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(ldsV + (vrow[1] + kw) * VP + colb));
                    s16x8 bfr;
                    bfr[0] = lo[0]; bfr[1] = lo[1]; bfr[2] = lo[2]; bfr[3] = lo[3];
                    bfr[4] = hi[0]; bfr[5] = hi[1]; bfr[6] = hi[2]; bfr[7] = hi[3];
#pragma unroll
                    for (int mi = 0; mi < 2; ++mi)
                        acc[kw][mi] = mma16v<T>(af[mi], bfr, acc[kw][mi]);
                }
            }
        } else {
            // ---- fp32: k-steps of 2 pixels; lane half h owns pixel 2*ks + h; channel on the lane ----
#pragma unroll 4
            for (int ks = 0; ks < WG_PX / 2; ++ks) {
                const int m = 2 * ks + h;
                const int tx = m & mTW, tyn = m >> shTW;
                const int vrow = tyn * g.HWd + tx * a.stride;
                float af[2];
#pragma unroll
                for (int mi = 0; mi < 2; ++mi) af[mi] = *reinterpret_cast<const float*>(ldsY + m * DYP + (wco + mi * 32 + r) * 4);
#pragma unroll
                for (int kw = 0; kw < KW; ++kw) {
                    const float bv = *reinterpret_cast<const float*>(ldsV + (vrow + kw) * VP + (wci + r) * 4);
#pragma unroll
                    for (int mi = 0; mi < 2; ++mi) acc[kw][mi] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[mi], bv, acc[kw][mi], 0, 0, 0);
                }
            }
        }
    }
    // ---- partial sums out: D[row = co][col = ci]; lane = ci column, registers = co rows ----
    const int ci = ci0 + wci + r;
#pragma unroll
    for (int kw = 0; kw < KW; ++kw) {
        const int tap = kh * KW + kw;
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const int co = co0 + wco + mi * 32 + (j & 3) + 8 * (j >> 2) + 4 * h;
                if (co < a.Cout) atomicAdd(dWp + ((int64_t)co * a.taps + tap) * Cin + ci, acc[kw][mi][j]);
            }
        }
    }
}


// ---- grouped weight gradients (3x3 and 1x1, stride 1, bf16) --------------------------------------------------
// Split-K weight gradients pay for their parallelism in atomics: every workgroup ends with one atomic per output it
// owns, and a layer launched alone needs hundreds of pixel splits to fill 256 CUs (measured: ~65 us of atomics per
// layer, more than the MFMA time of most layers).  The backward therefore DEFERS these weight gradients - every dY
// and every forward activation stays in its workspace - and runs them in ONE launch per kernel size over a plan-time
// job table: ~50 layers supply the parallelism, each job owns a long contiguous run of pixel tiles, and the atomic
// count drops by more than an order of magnitude.
// One job = (64*WM) couts x (64*WN) cins x all TAPS taps (4 waves 2x2, wave tile 32*WM co x 32*WN ci) over `ntiles`
// 64-pixel tiles: <9,1,1> for 3x3 (144 accumulator registers), <1,2,2> for 1x1.  The raw dY / input vectors of the
// next tile are prefetched into registers while the MFMAs of the current one run; the GN-affine/SiLU/dropout
// prologue is applied when they are written to LDS.
typedef __attribute__((address_space(3))) char lds_char;
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

template <int TAPS, int WM, int WN>
struct WgGroupGeom {
    static constexpr int CO = 64 * WM, CI = 64 * WN;
    static constexpr int DYP = CO * 2 + 64, VP = CI * 2 + 64;  // pitches: the 4 rows of a transposed read on disjoint banks
    static constexpr int MAX_ROWS = TAPS == 9 ? 160 : 64;     // halo rows of one 64-pixel tile
    static constexpr int UY = WG_PX * (CO / 8) / 256;          // dY vectors per thread
    static constexpr int UV = (MAX_ROWS * (CI / 8) + 255) / 256;  // input vectors per thread
    static constexpr size_t LDS = (size_t)WG_PX * DYP + (size_t)MAX_ROWS * VP;
};

template <typename T, int TAPS, int WM, int WN>
__global__ void __launch_bounds__(256, 2) wgrad_group_kernel(const WgLayer* __restrict__ layers, const WgJob* __restrict__ jobs,
                                                             const char* __restrict__ ws, const char* __restrict__ bws,
                                                             const float* __restrict__ drop_masks, float* __restrict__ wimage) {
    static_assert(sizeof(T) == 2, "grouped weight gradient: 16-bit tensors only");
    using GG = WgGroupGeom<TAPS, WM, WN>;
    constexpr int EPV = 8, CO = GG::CO, CI = GG::CI, DYP = GG::DYP, VP = GG::VP, UY = GG::UY, UV = GG::UV;
    constexpr int PAD = TAPS == 9 ? 1 : 0;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    lds_char* ldsY = (lds_char*)lds;      // [64 px][CO]
    lds_char* ldsV = ldsY + WG_PX * DYP;  // [halo px][CI]
    const WgJob job = jobs[blockIdx.x];
    if (job.ntiles <= 0) return;  // padding of a short XCD slice
    const WgLayer& L = layers[job.layer];
    const ConvTile g = L.g;
    const int shTW = L.shTW, shTH = L.shTH;
    const int N = L.N, Hin = L.Hin, Win = L.Win, C1 = L.C1, C2 = L.C2, up = L.up, Hout = L.Hout, Wout = L.Wout, Cout = L.Cout;
    const int pro_silu = L.act_off >= 0 ? 0 : L.pro_silu;
    const T* dY = (const T*)(bws + L.dy_off);
    float* dWp = wimage + L.dw_off;
    const float* scale = L.scale_off >= 0 && L.act_off < 0 ? (const float*)(ws + L.scale_off) : nullptr;
    const float* shift = scale ? (const float*)(ws + L.shift_off) : nullptr;
    const float* dmask = (drop_masks && L.dmask_off >= 0 && L.act_off < 0) ? drop_masks + L.dmask_off : nullptr;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wco = (wave >> 1) * 32 * WM, wci = (wave & 1) * 32 * WN;
    const int r = lane & 31, h = lane >> 5;
    const int Cin = C1 + C2;
    const int co0 = job.cot * CO, ci0 = job.cit * CI;
    const int Hv = up ? 2 * Hin : Hin, Wv = up ? 2 * Win : Win;
    const int mTW = (1 << shTW) - 1, mTH = (1 << shTH) - 1;
    const int halo_px = g.HH * g.HWd;
    const bool has_act = L.act_off >= 0;  // pre-activated input: one tensor over all C1 + C2 channels, nothing to apply
    const bool second = !has_act && ci0 >= C1;
    const T* sbase = has_act ? (const T*)((L.act_bws ? bws : ws) + L.act_off) : (const T*)(ws + (second ? L.src2_off : L.src1_off));
    const int Cs = has_act ? C1 + C2 : second ? C2 : C1, cs0 = second ? ci0 - C1 : ci0;
    f32x16 acc[TAPS][WM][WN];
#pragma unroll
    for (int k = 0; k < TAPS; ++k)
#pragma unroll
        for (int a = 0; a < WM; ++a)
#pragma unroll
            for (int b = 0; b < WN; ++b)
#pragma unroll
                for (int j = 0; j < 16; ++j) acc[k][a][b][j] = 0.f;
    const int tr_q = (lane & 15) >> 2, tr_p = lane & 3, tr_g1 = (lane >> 4) & 1;
    // LDS byte offsets of this lane's fragment rows for k-step 0 (pixel 8h + tr_q); pixel bits 2 (second half of the
    // fragment) and 4-5 (k-step) add the uniform offsets L.half_off / L.ks_off (bit fields of tx, ty, tn do not carry)
    const int a_base = (8 * h + tr_q) * DYP + (wco + 16 * tr_g1 + 4 * tr_p) * 2;
    int b_base;
    {
        const int m = 8 * h + tr_q;
        const int tx = m & mTW, ty = (m >> shTW) & mTH, tn = m >> (shTW + shTH);
        b_base = ((tn * g.HH + ty) * g.HWd + tx) * VP + (wci + 16 * tr_g1 + 4 * tr_p) * 2;
    }
    const int row_b = g.HWd * VP;

    uint4 ry[UY], rv[UV];
    unsigned vmask = 0;                            // bit k: halo vector k is inside the image (else zero padding)
    int n0_cur = 0;
    auto issue = [&](int tile) {
        const int tx_blk = tile % g.tiles_x, ty_blk = (tile / g.tiles_x) % g.tiles_y;
        const int n0 = (tile / (g.tiles_x * g.tiles_y)) * g.TN;
        const int oy0 = ty_blk << shTH, ox0 = tx_blk << shTW;
        n0_cur = n0;
        vmask = 0;
#pragma unroll
        for (int k = 0; k < UY; ++k) {
            const int u = tid + 256 * k;
            const int m = u / (CO / EPV), cu = u % (CO / EPV);
            const int tx = m & mTW, ty = (m >> shTW) & mTH, tn = m >> (shTW + shTH);
            const int n = n0 + tn, co = co0 + cu * EPV;
            ry[k] = make_uint4(0u, 0u, 0u, 0u);
            if (n < N && co < Cout)
                ry[k] = *reinterpret_cast<const uint4*>(dY + (((int64_t)n * Hout + oy0 + ty) * Wout + ox0 + tx) * Cout + co);
        }
#pragma unroll
        for (int k = 0; k < UV; ++k) {
            const int u = tid + 256 * k;
            const int row = u / (CI / EPV), cu = u % (CI / EPV);
            const int tn = (int)__umulhi((unsigned)row, g.magic_px), rem = row - tn * halo_px;
            const int hy = (int)__umulhi((unsigned)rem, g.magic_w), hx = rem - hy * g.HWd;
            const int n = n0 + tn, iy = oy0 - PAD + hy, ix = ox0 - PAD + hx;
            rv[k] = make_uint4(0u, 0u, 0u, 0u);
            if (row < g.a_rows && n < N && iy >= 0 && iy < Hv && ix >= 0 && ix < Wv) {
                const int sy = up ? (iy >> 1) : iy, sx = up ? (ix >> 1) : ix;
                rv[k] = *reinterpret_cast<const uint4*>(sbase + (((int64_t)n * Hin + sy) * Win + sx) * Cs + cs0 + cu * EPV);
                vmask |= 1u << k;
            }
        }
    };
    const int tile_end = job.tile0 + job.ntiles;
    issue(job.tile0);
    for (int tile = job.tile0; tile < tile_end; ++tile) {
        __syncthreads();
#pragma unroll
        for (int k = 0; k < UY; ++k) {
            const int u = tid + 256 * k;
            *reinterpret_cast<uint4*>(lds + (u / (CO / EPV)) * DYP + (u % (CO / EPV)) * 16) = ry[k];
        }
#pragma unroll
        for (int k = 0; k < UV; ++k) {
            const int u = tid + 256 * k;
            const int row = u / (CI / EPV), cu = u % (CI / EPV);
            uint4 val = rv[k];
            if (vmask & (1u << k)) {
                const int n = n0_cur + (int)__umulhi((unsigned)row, g.magic_px);
                const int64_t so = (int64_t)n * Cin + ci0 + cu * EPV;
                val = prologue_vec<T>(val, scale ? scale + so : nullptr, scale ? shift + so : nullptr, dmask ? dmask + so : nullptr, pro_silu);
            }
            if (row < g.a_rows) *reinterpret_cast<uint4*>(lds + WG_PX * DYP + row * VP + cu * 16) = val;
        }
        __syncthreads();
        if (tile + 1 < tile_end) issue(tile + 1);
#pragma unroll TAPS == 9 ? 1 : 4
        for (int ks = 0; ks < WG_PX / 16; ++ks) {
            s16x8 af[WM];
#pragma unroll
            for (int a = 0; a < WM; ++a) {
                lds_char* p0 = ldsY + a_base + 16 * ks * DYP + a * 64;
                const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)p0);
                const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p0 + 4 * DYP));
                af[a][0] = lo[0]; af[a][1] = lo[1]; af[a][2] = lo[2]; af[a][3] = lo[3];
                af[a][4] = hi[0]; af[a][5] = hi[1]; af[a][6] = hi[2]; af[a][7] = hi[3];
            }
            const int ks_b = b_base + L.ks_off[ks];
#pragma unroll
            for (int tap = 0; tap < TAPS; ++tap) {
                const int t_b = ks_b + (tap / 3) * row_b;
#pragma unroll
                for (int b = 0; b < WN; ++b) {
                    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(ldsV + t_b + (tap % 3) * VP + b * 64));
                    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(ldsV + t_b + L.half_off + (tap % 3) * VP + b * 64));
                    s16x8 bfr;
                    bfr[0] = lo[0]; bfr[1] = lo[1]; bfr[2] = lo[2]; bfr[3] = lo[3];
                    bfr[4] = hi[0]; bfr[5] = hi[1]; bfr[6] = hi[2]; bfr[7] = hi[3];
#pragma unroll
                    for (int a = 0; a < WM; ++a)
                        acc[tap][a][b] = mma16v<T>(af[a], bfr, acc[tap][a][b]);
                }
            }
        }
    }
#pragma unroll
    for (int tap = 0; tap < TAPS; ++tap)
#pragma unroll
        for (int a = 0; a < WM; ++a)
#pragma unroll
            for (int b = 0; b < WN; ++b) {
                const int ci = ci0 + wci + 32 * b + r;
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    const int co = co0 + wco + 32 * a + (j & 3) + 8 * (j >> 2) + 4 * h;
                    if (co < Cout) atomicAdd(dWp + ((int64_t)co * TAPS + tap) * Cin + ci, acc[tap][a][b][j]);
                }
            }
}

// ---- the same 3x3 job with BOTH operand tiles by LDS-DMA -------------------------------------------------------------------------
// Once the activated input exists as a tensor (WgLayer::act_off), neither operand needs a register: the dY tile (64 pixels x 64 couts)
// and the input halo tile (<= 160 rows x 64 cins) of tile t + 1 go global -> LDS into the other half of a double buffer while the MFMAs
// of tile t run; padding rows read a page of zeros.  One workgroup barrier per tile, no VGPR staging, no ds_write, no prologue.
// LDS layout [channel half][row][64 B]: a wave only ever reads the 32 channels of its own quadrant, and a transposed read touches 4
// consecutive rows x 32 B - at a 64-byte pitch those cover disjoint banks with no padding or swizzle, and every fragment address of a
// tile is (per-lane base) + (wave-uniform offset) + (immediate): under one VALU instruction per MFMA (the register-staged kernel spent
// eight, which made it VALU-issue bound: 4 cycles each against the MFMA's 32).
// STRIDE 2 (the three DownSample convs; no upsampling there): the 64-pixel tile's halo is 17 x 17 / 9 x 33 rows (four 4x4 images: 4 x 9 x 9) - V_ROWS 336, one
// workgroup per CU - instead of the per-layer kernel's ~65 us of atomics per layer.
template <typename T, int STRIDE, int V_ROWS>
__global__ void __launch_bounds__(256, STRIDE == 1 ? 2 : 1) wgrad_dma_kernel(const WgLayer* __restrict__ layers, const WgJob* __restrict__ jobs,
                                                                            const char* __restrict__ ws, const char* __restrict__ bws,
                                                                            const char* __restrict__ zero_page, float* __restrict__ wimage) {
    static_assert(sizeof(T) == 2, "bf16 only");
    constexpr int TAPS = 9, HB = 64;                       // bytes of one row of one channel half
    constexpr int Y_HALF = WG_PX * HB, Y_BYTES = 2 * Y_HALF;  // 8 KB
    constexpr int V_HALF = V_ROWS * HB, BUF = Y_BYTES + 2 * V_HALF, VU = (V_ROWS / 16 * 2 + 3) / 4;  // halo DMA instructions per wave
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const WgJob job = jobs[blockIdx.x];
    if (job.ntiles <= 0) return;  // padding of a short XCD slice
    const WgLayer& L = layers[job.layer];
    const ConvTile g = L.g;
    const int shTW = L.shTW, shTH = L.shTH;
    const int Hin = L.Hin, Win = L.Win, up = L.up, Hout = L.Hout, Wout = L.Wout, Cout = L.Cout;
    const int Cin = L.C1 + L.C2;
    const char* dY = bws + L.dy_off;
    const char* vbase = L.act_off >= 0 ? (L.act_bws ? bws : ws) + L.act_off : ws + L.src1_off;
    float* dWp = wimage + L.dw_off;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wco = (wave >> 1) * 32, wci = (wave & 1) * 32;
    const int r = lane & 31, h = lane >> 5;
    const int co0 = job.cot * 64, ci0 = job.cit * 64;
    const int Hv = up ? 2 * Hin : Hin, Wv = up ? 2 * Win : Win;
    const int mTW = (1 << shTW) - 1, mTH = (1 << shTH) - 1;
    const int halo_px = g.HH * g.HWd, HWo = Hout * Wout;

    f32x16 acc[TAPS];
#pragma unroll
    for (int k = 0; k < TAPS; ++k)
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[k][j] = 0.f;

    // ---- DMA descriptors (tile-invariant).  Instruction j = wave + 4 k of a tile fills 16 rows of channel half (j & 1):
    // lane -> row 16 (j >> 1) + (lane >> 2), 16-byte piece (lane & 3) of the half
    const int prow = lane >> 2, piece = lane & 3;
    unsigned y_off[2];  // byte offset of this lane's dY vector relative to the tile's first pixel
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int j = wave + 4 * k, m = 16 * (j >> 1) + prow;
        const int tx = m & mTW, ty = (m >> shTW) & mTH, tn = m >> (shTW + shTH);
        y_off[k] = (unsigned)(((tn * HWo + ty * Wout + tx) * Cout + co0 + (j & 1) * 32 + piece * 8) * 2);
    }
    int v_hy[VU], v_hx[VU], v_tn[VU];
#pragma unroll
    for (int k = 0; k < VU; ++k) {
        const int j = wave + 4 * k, row = 16 * (j >> 1) + prow;
        const int tn = (int)__umulhi((unsigned)row, g.magic_px), rem = row - tn * halo_px;
        const int hy = (int)__umulhi((unsigned)rem, g.magic_w);
        v_tn[k] = row < g.a_rows ? tn : -1;
        v_hy[k] = hy;
        v_hx[k] = rem - hy * g.HWd;
    }
    const int v_cb = (ci0 + (wave & 1) * 32 + piece * 8) * 2;  // (j & 1) == (wave & 1) for every k
    const int n_vj = 2 * ((g.a_rows + 15) >> 4);              // halo DMA instructions of a tile
    const unsigned lds0 = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(size_t)(lds_c*)lds);
    auto issue_tile = [&](int tile, int buf) __attribute__((always_inline)) {
        const int tx_blk = tile % g.tiles_x, ty_blk = (tile / g.tiles_x) % g.tiles_y;
        const int n0 = (tile / (g.tiles_x * g.tiles_y)) * g.TN;
        const int oy0 = ty_blk << shTH, ox0 = tx_blk << shTW;
        const char* ybase = dY + (int64_t)((n0 * Hout + oy0) * Wout + ox0) * Cout * 2;  // wave-uniform
        const unsigned lb = lds0 + (unsigned)(buf * BUF);
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int j = wave + 4 * k;
            glds16_hidden_s(ybase, y_off[k], lb + (unsigned)((j & 1) * Y_HALF + (j >> 1) * 1024));
        }
#pragma unroll
        for (int k = 0; k < VU; ++k) {
            const int j = wave + 4 * k;
            if (j >= n_vj) break;  // wave-uniform
            const int iy = oy0 * STRIDE - 1 + v_hy[k], ix = ox0 * STRIDE - 1 + v_hx[k];
            const bool ok = v_tn[k] >= 0 && (unsigned)iy < (unsigned)Hv && (unsigned)ix < (unsigned)Wv;
            const int sy = up ? (iy >> 1) : iy, sx = up ? (ix >> 1) : ix;
            const int64_t off = (int64_t)(((n0 + v_tn[k]) * Hin + sy) * Win + sx) * Cin * 2 + v_cb;
            const char* src = ok ? vbase + off : zero_page;
            glds16_hidden(src, lb + (unsigned)(Y_BYTES + (j & 1) * V_HALF + (j >> 1) * 1024));
        }
    };

    // ---- fragment addresses: lane (tr_q, tr_p, tr_g1) of a transposed read takes 8 bytes of row 8 h + tr_q (+ 4 for the second read) ----
    const int tr_q = (lane & 15) >> 2, tr_p = lane & 3, tr_g1 = (lane >> 4) & 1;
    const unsigned colb = (unsigned)((16 * tr_g1 + 4 * tr_p) * 2);
    const unsigned a_lane = (unsigned)((wco >> 5) * Y_HALF + (8 * h + tr_q) * HB) + colb;
    unsigned b_lane;
    {
        const int m = 8 * h + tr_q;
        const int tx = m & mTW, ty = (m >> shTW) & mTH, tn = m >> (shTW + shTH);
        b_lane = (unsigned)(Y_BYTES + (wci >> 5) * V_HALF + ((tn * g.HH + ty * STRIDE) * g.HWd + tx * STRIDE) * HB) + colb;
    }
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4_d;
#define WGD_TR(ADDR, IMM) __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_d*)(size_t)((ADDR) + (IMM)))
    const int row_b = g.HWd * HB, half_b = L.half_row * HB;
    int ks_b[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) ks_b[ks] = L.ks_row[ks] * HB;

    const int tile_end = job.tile0 + job.ntiles;
    issue_tile(job.tile0, 0);
    int buf = 0;
#pragma unroll 1
    for (int tile = job.tile0; tile < tile_end; ++tile) {
        wait_vm_all();     // this wave's share of tile `tile` has landed (nothing younger is in flight)
        __syncthreads();   // everyone's has - and everyone is done reading the other buffer
        if (tile + 1 < tile_end) issue_tile(tile + 1, buf ^ 1);
        const unsigned base = lds0 + (unsigned)(buf * BUF);
        const unsigned ya = base + a_lane, va = base + b_lane;
#pragma unroll
        for (int ks = 0; ks < WG_PX / 16; ++ks) {
            const s16x4 alo = WGD_TR(ya, ks * 16 * HB), ahi = WGD_TR(ya, ks * 16 * HB + 4 * HB);
            s16x8 af;
            af[0] = alo[0]; af[1] = alo[1]; af[2] = alo[2]; af[3] = alo[3];
            af[4] = ahi[0]; af[5] = ahi[1]; af[6] = ahi[2]; af[7] = ahi[3];
#pragma unroll
            for (int dy = 0; dy < 3; ++dy) {
                const unsigned vlo = va + (unsigned)(ks_b[ks] + dy * row_b), vhi = vlo + (unsigned)half_b;
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) {
                    const s16x4 lo = WGD_TR(vlo, dx * HB), hi = WGD_TR(vhi, dx * HB);
                    s16x8 bfr;
                    bfr[0] = lo[0]; bfr[1] = lo[1]; bfr[2] = lo[2]; bfr[3] = lo[3];
                    bfr[4] = hi[0]; bfr[5] = hi[1]; bfr[6] = hi[2]; bfr[7] = hi[3];
                    acc[dy * 3 + dx] = mma16v<T>(af, bfr, acc[dy * 3 + dx]);
                }
            }
        }
        buf ^= 1;
    }
#undef WGD_TR
    const int ci = ci0 + wci + r;
#pragma unroll
    for (int tap = 0; tap < TAPS; ++tap)
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int co = co0 + wco + (j & 3) + 8 * (j >> 2) + 4 * h;
            if (co < Cout) atomicAdd(dWp + ((int64_t)co * TAPS + tap) * Cin + ci, acc[tap][j]);
        }
}

// ---- 1x1 jobs (128 couts x 128 cins) the same way: no halo, both tiles are 64 pixels x 128 channels in [channel quarter][pixel][64 B] ----
template <typename T>
__global__ void __launch_bounds__(256, 2) wgrad_dma1_kernel(const WgLayer* __restrict__ layers, const WgJob* __restrict__ jobs,
                                                           const char* __restrict__ ws, const char* __restrict__ bws, float* __restrict__ wimage) {
    static_assert(sizeof(T) == 2, "bf16 only");
    constexpr int HB = 64, QB = WG_PX * HB, TILE = 4 * QB, BUF = 2 * TILE;  // 4 KB per quarter, 16 KB per operand tile, 32 KB per buffer
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const WgJob job = jobs[blockIdx.x];
    if (job.ntiles <= 0) return;
    const WgLayer& L = layers[job.layer];
    const ConvTile g = L.g;
    const int shTW = L.shTW, shTH = L.shTH;
    const int Hout = L.Hout, Wout = L.Wout, Cout = L.Cout, Cin = L.C1 + L.C2;
    const char* dY = bws + L.dy_off;
    const int co0 = job.cot * 128, ci0 = job.cit * 128;
    // the job's 128 input channels: the activated tensor, or (no norm in front of the conv) one of the two concatenated sources
    const bool has_act = L.act_off >= 0, second = !has_act && ci0 >= L.C1;
    const char* vbase = has_act ? (L.act_bws ? bws : ws) + L.act_off : ws + (second ? L.src2_off : L.src1_off);
    const int Cs = has_act ? Cin : second ? L.C2 : L.C1, cs0 = second ? ci0 - L.C1 : ci0;
    float* dWp = wimage + L.dw_off;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int mTW = (1 << shTW) - 1, mTH = (1 << shTH) - 1;
    const int HWo = Hout * Wout;
    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int j = 0; j < 16; ++j) acc[a][b][j] = 0.f;
    // DMA: instruction j = wave + 4 k (k < 4) of an operand tile fills 16 pixels of channel quarter (j & 3) = wave: rows 16 k + (lane >> 2)
    const int prow = lane >> 2, piece = lane & 3;
    unsigned y_off[4], v_off[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int m = 16 * k + prow;
        const int tx = m & mTW, ty = (m >> shTW) & mTH, tn = m >> (shTW + shTH);
        const int pix = tn * HWo + ty * Wout + tx;
        y_off[k] = (unsigned)((pix * Cout + co0 + wave * 32 + piece * 8) * 2);
        v_off[k] = (unsigned)((pix * Cs + cs0 + wave * 32 + piece * 8) * 2);
    }
    const unsigned lds0 = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(size_t)(lds_c*)lds);
    auto issue_tile = [&](int tile, int buf) __attribute__((always_inline)) {
        const int tx_blk = tile % g.tiles_x, ty_blk = (tile / g.tiles_x) % g.tiles_y;
        const int n0 = (tile / (g.tiles_x * g.tiles_y)) * g.TN;
        const int64_t p0 = (int64_t)((n0 * Hout + (ty_blk << shTH)) * Wout + (tx_blk << shTW));
        const char* ybase = dY + p0 * Cout * 2;
        const char* xbase = vbase + p0 * Cs * 2;
        const unsigned lb = lds0 + (unsigned)(buf * BUF + wave * QB);
#pragma unroll
        for (int k = 0; k < 4; ++k) glds16_hidden_s(ybase, y_off[k], lb + (unsigned)(k * 1024));
#pragma unroll
        for (int k = 0; k < 4; ++k) glds16_hidden_s(xbase, v_off[k], lb + (unsigned)(TILE + k * 1024));
    };
    const int tr_q = (lane & 15) >> 2, tr_p = lane & 3, tr_g1 = (lane >> 4) & 1;
    const unsigned lane_b = (unsigned)((8 * h + tr_q) * HB + (16 * tr_g1 + 4 * tr_p) * 2);
    const unsigned a_lane = (unsigned)((wave >> 1) * 2 * QB) + lane_b, b_lane = (unsigned)(TILE + (wave & 1) * 2 * QB) + lane_b;
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4_d;
#define WGD_TR(ADDR, IMM) __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_d*)(size_t)((ADDR) + (IMM)))
    const int tile_end = job.tile0 + job.ntiles;
    issue_tile(job.tile0, 0);
    int buf = 0;
#pragma unroll 1
    for (int tile = job.tile0; tile < tile_end; ++tile) {
        wait_vm_all();
        __syncthreads();
        if (tile + 1 < tile_end) issue_tile(tile + 1, buf ^ 1);
        const unsigned base = lds0 + (unsigned)(buf * BUF);
        const unsigned ya = base + a_lane, va = base + b_lane;
#pragma unroll
        for (int ks = 0; ks < WG_PX / 16; ++ks) {
            s16x8 af[2], bfr[2];
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                const s16x4 lo = WGD_TR(ya, a * QB + ks * 16 * HB), hi = WGD_TR(ya, a * QB + ks * 16 * HB + 4 * HB);
                af[a][0] = lo[0]; af[a][1] = lo[1]; af[a][2] = lo[2]; af[a][3] = lo[3];
                af[a][4] = hi[0]; af[a][5] = hi[1]; af[a][6] = hi[2]; af[a][7] = hi[3];
            }
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const s16x4 lo = WGD_TR(va, b * QB + ks * 16 * HB), hi = WGD_TR(va, b * QB + ks * 16 * HB + 4 * HB);
                bfr[b][0] = lo[0]; bfr[b][1] = lo[1]; bfr[b][2] = lo[2]; bfr[b][3] = lo[3];
                bfr[b][4] = hi[0]; bfr[b][5] = hi[1]; bfr[b][6] = hi[2]; bfr[b][7] = hi[3];
            }
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b)
                    acc[a][b] = mma16v<T>(af[a], bfr[b], acc[a][b]);
        }
        buf ^= 1;
    }
#undef WGD_TR
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int ci = ci0 + (wave & 1) * 64 + 32 * b + r;
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const int co = co0 + (wave >> 1) * 64 + 32 * a + (j & 3) + 8 * (j >> 2) + 4 * h;
                if (co < Cout) atomicAdd(dWp + (int64_t)co * Cin + ci, acc[a][b][j]);
            }
        }
}

static bool wg_tile(const ConvArgs& a, ConvTile& g) {
    if (!make_tile(a, WG_PX, 64, g)) return false;
    if (g.TW < 4) return false;  // 4-pixel transposed-read blocks must stay inside one tile row
    return true;
}

bool wgrad_mfma_supported(int dtype, const ConvArgs& a) {
    const int KC = is16(dtype) ? 64 : 32;
    (void)KC;
    if (a.in_nchw) return false;
    if (a.taps != 9 && a.taps != 1) return false;
    if (a.taps == 1 && (a.stride != 1 || a.up)) return false;
    if (a.up == 2) return false;
    if (a.C1 % WG_CI || a.C2 % WG_CI || a.C1 == 0) return false;
    if (a.Cout % 32) return false;
    ConvTile g;
    if (!wg_tile(a, g)) return false;
    const size_t lds = (size_t)WG_PX * (is16(dtype) ? 320 : 512) + (size_t)g.TN * g.TH * g.HWd * (is16(dtype) ? 192 : 256);
    return lds <= 64 * 1024;
}

template <typename T>
static int launch_wgrad_t(const ConvArgs& a, const void* dY, float* dWp, hipStream_t s) {
    ConvTile g{};
    DMME_REQUIRE(wg_tile(a, g), DMME_ERR_UNSUPPORTED, "wgrad_mfma: no tile");
    const int Cin = a.C1 + a.C2;
    const int n_ci = Cin / WG_CI, n_co = (a.Cout + WG_CO - 1) / WG_CO, n_kh = a.taps == 9 ? 3 : 1;
    const int base = n_ci * n_co * n_kh;
    int nsplit = 768 / base;
    if (nsplit < 1) nsplit = 1;
    if (nsplit > g.tiles_m) nsplit = g.tiles_m;
    int shTW = 0, shTH = 0;
    while ((1 << shTW) < g.TW) ++shTW;
    while ((1 << shTH) < g.TH) ++shTH;
    const size_t lds = (size_t)WG_PX * WgGeom<T>::DY_PITCH + (size_t)g.TN * g.TH * g.HWd * WgGeom<T>::V_PITCH;
    const dim3 grid((unsigned)(base * nsplit));
    if (a.taps == 9)
        hipLaunchKernelGGL((wgrad_mfma_kernel<T, 3>), grid, dim3(256), lds, s, a, g, (const T*)dY, dWp, nsplit, shTW, shTH);
    else
        hipLaunchKernelGGL((wgrad_mfma_kernel<T, 1>), grid, dim3(256), lds, s, a, g, (const T*)dY, dWp, nsplit, shTW, shTH);
    DMME_CHECK_LAUNCH();
    return DMME_OK;
}

// dWp: this conv's slice of the zero-initialised packed-layout gradient image ([co][tap][ci] fp32); partial
// sums are added atomically.  launch_wgrad_unpack folds the whole image into the reference-layout gradients.
int launch_wgrad_mfma(int dtype, const ConvArgs& a, const void* dY, float* dWp, hipStream_t s) {
    DMME_REQUIRE(wgrad_mfma_supported(dtype, a), DMME_ERR_UNSUPPORTED, "wgrad_mfma: unsupported shape");
    return dtype == DMME_BF16 ? launch_wgrad_t<bf16>(a, dY, dWp, s) : dtype == DMME_F16 ? launch_wgrad_t<f16>(a, dY, dWp, s) : launch_wgrad_t<float>(a, dY, dWp, s);
}

// table-driven: one workgroup per item (a run of cout rows of one conv weight)
__global__ void __launch_bounds__(256) wgrad_unpack_table_kernel(const PackItem* __restrict__ items, const float* __restrict__ image,
                                                                 float* __restrict__ grad_flat) {
    const PackItem it = items[blockIdx.x];
    const int64_t row = (int64_t)it.cin * it.taps;
    const int64_t total = (int64_t)it.rows * row, e0 = (int64_t)it.row0 * row;
    const float* src = image + it.dst_off;   // float offset of this weight's packed image
    float* dst = grad_flat + it.src_off;     // float offset of the reference-layout gradient
    for (int64_t e = threadIdx.x; e < total; e += blockDim.x) {
        const int64_t g = e0 + e;            // reference index (co*cin + ci)*taps + tap
        const int tap = (int)(g % it.taps);
        const int64_t q = g / it.taps;
        const int ci = (int)(q % it.cin);
        const int64_t co = q / it.cin;
        dst[g] += src[(co * it.taps + tap) * it.cin + ci];
    }
}
int launch_wgrad_unpack(const PackItem* items_dev, int n_items, const float* image, float* grad_flat, hipStream_t s) {
    if (n_items == 0) return DMME_OK;
    hipLaunchKernelGGL(wgrad_unpack_table_kernel, dim3(n_items), dim3(256), 0, s, items_dev, image, grad_flat);
    DMME_CHECK_LAUNCH();
    return DMME_OK;
}

bool wgrad_group_layer(int dtype, const ConvArgs& a, WgLayer& L, int* co_tile, int* ci_tile) {
    if (!is16(dtype) || (a.taps != 9 && a.taps != 1) || (a.stride != 1 && !(a.stride == 2 && a.taps == 9 && !a.up)) || a.in_nchw || a.up == 2) return false;
    if (a.taps == 1 && a.up) return false;
    const int CO = a.taps == 9 ? 64 : 128, CI = a.taps == 9 ? 64 : 128;
    const int Cin = a.C1 + a.C2;
    if (Cin % CI || a.C1 % CI || a.Cout % 8) return false;
    ConvTile g{};
    if (!make_tile(a, WG_PX, 64, g) || g.TW < 4 || g.a_rows > (a.taps == 1 ? 64 : a.stride == 2 ? 336 : 160) || a.N % g.TN) return false;
    L.g = g;
    L.shTW = L.shTH = 0;
    while ((1 << L.shTW) < g.TW) ++L.shTW;
    while ((1 << L.shTH) < g.TH) ++L.shTH;
    if ((1 << L.shTW) != g.TW || (1 << L.shTH) != g.TH) return false;
    const int VP = CI * 2 + 64;
    auto row_of = [&](int m) {  // halo row of tile pixel m (tap 0)
        const int tx = m & (g.TW - 1), ty = (m >> L.shTW) & (g.TH - 1), tn = m >> (L.shTW + L.shTH);
        return (tn * g.HH + ty * a.stride) * g.HWd + tx * a.stride;
    };
    for (int ks = 0; ks < 4; ++ks) {
        L.ks_off[ks] = row_of(16 * ks) * VP;
        L.ks_row[ks] = row_of(16 * ks);
    }
    L.half_off = row_of(4) * VP;
    L.half_row = row_of(4);
    L.N = a.N; L.Hin = a.Hin; L.Win = a.Win; L.C1 = a.C1; L.C2 = a.C2; L.up = a.up;
    L.Hout = a.Hout; L.Wout = a.Wout; L.Cout = a.Cout; L.pro_silu = a.pro_silu;
    *co_tile = CO;
    *ci_tile = CI;
    return true;
}

template <typename T>
static int launch_wgrad_group_t(int taps, const WgLayer* layers_dev, const WgJob* jobs_dev, int njobs, const void* ws, const void* bws,
                       const float* drop_masks, float* wimage, hipStream_t s, int dma, const void* zero_page) {
    if (njobs <= 0) return DMME_OK;
    if (taps == 9 && dma == 2 && zero_page) {  // the stride-2 table
        constexpr size_t lds = 2 * (WG_PX * 128 + 336 * 128);
        static bool attr = false;  // (one per instantiation, i.e. per T)
        if (!attr) {
            DMME_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_dma_kernel<T, 2, 336>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            attr = true;
        }
        hipLaunchKernelGGL((wgrad_dma_kernel<T, 2, 336>), dim3((unsigned)njobs), dim3(256), lds, s, layers_dev, jobs_dev, (const char*)ws, (const char*)bws,
                           (const char*)zero_page, wimage);
        DMME_CHECK_LAUNCH();
        return DMME_OK;
    }
    if (taps == 9 && dma && zero_page) {
        constexpr size_t lds = 2 * (WG_PX * 128 + 160 * 128);
        static_assert(lds <= 64 * 1024, "two workgroups per CU");
        hipLaunchKernelGGL((wgrad_dma_kernel<T, 1, 160>), dim3((unsigned)njobs), dim3(256), lds, s, layers_dev, jobs_dev, (const char*)ws, (const char*)bws,
                           (const char*)zero_page, wimage);
        DMME_CHECK_LAUNCH();
        return DMME_OK;
    }
    if (taps == 1 && dma) {
        constexpr size_t lds = 2 * 2 * 4 * WG_PX * 64;
        static_assert(lds <= 64 * 1024, "two workgroups per CU");
        hipLaunchKernelGGL((wgrad_dma1_kernel<T>), dim3((unsigned)njobs), dim3(256), lds, s, layers_dev, jobs_dev, (const char*)ws, (const char*)bws, wimage);
        DMME_CHECK_LAUNCH();
        return DMME_OK;
    }
    if (taps == 9) {
        constexpr size_t lds = WgGroupGeom<9, 1, 1>::LDS;
        static_assert(lds <= 64 * 1024, "LDS tile");
        hipLaunchKernelGGL((wgrad_group_kernel<T, 9, 1, 1>), dim3((unsigned)njobs), dim3(256), lds, s, layers_dev, jobs_dev, (const char*)ws,
                           (const char*)bws, drop_masks, wimage);
    } else {
        constexpr size_t lds = WgGroupGeom<1, 2, 2>::LDS;
        static_assert(lds <= 64 * 1024, "LDS tile");
        hipLaunchKernelGGL((wgrad_group_kernel<T, 1, 2, 2>), dim3((unsigned)njobs), dim3(256), lds, s, layers_dev, jobs_dev, (const char*)ws,
                           (const char*)bws, drop_masks, wimage);
    }
    DMME_CHECK_LAUNCH();
    return DMME_OK;
}

int launch_wgrad_group(int dtype, int taps, const WgLayer* layers_dev, const WgJob* jobs_dev, int njobs, const void* ws, const void* bws,
                       const float* drop_masks, float* wimage, hipStream_t s, int dma, const void* zero_page) {
    DMME_REQUIRE(is16(dtype) && (taps == 9 || taps == 1), DMME_ERR_UNSUPPORTED, "grouped weight gradient: 16-bit tensors, 3x3 or 1x1 only");
    if (dtype == DMME_F16) return launch_wgrad_group_t<f16>(taps, layers_dev, jobs_dev, njobs, ws, bws, drop_masks, wimage, s, dma, zero_page);
    return launch_wgrad_group_t<bf16>(taps, layers_dev, jobs_dev, njobs, ws, bws, drop_masks, wimage, s, dma, zero_page);
}

}  // namespace dmme
