// Implicit-GEMM convolution on the CDNA4 matrix cores (3x3 pad 1 / 1x1, stride 1|2,
// optional fused nearest-2x upsample, optional two-source channel concat).
//
//   D[pixel][cout] = sum_{tap, cin} act(A)[pixel + tap][cin] * W[cout][tap][cin]
//
// * activations NHWC, weights [Cout][tap][Cin]: both operands are K-contiguous, so an
//   MFMA fragment is one 16-byte load per lane;
// * one workgroup = BM output pixels (TN images x TH x TW spatial rectangle) x BN couts,
//   4 wavefronts in a 2x2 arrangement, 32x32 MFMA tiles;
// * per Cin chunk (128 bytes per pixel) the input HALO tile is staged ONCE into LDS with
//   the GroupNorm affine + SiLU + Dropout2d multiplier applied on the way (the
//   normalised tensor never exists in HBM) and is then reused by all 9 taps through
//   shifted LDS row addresses; the filter tile of each tap is staged per tap;
// * LDS rows are 128 data bytes + 16 pad bytes: consecutive rows land on different
//   16-byte bank slots, so the ds_read_b128 fragment reads are conflict-free;
// * epilogue: + bias + time-embedding projection + residual, optional SiLU, store.
//
// dtype float  : v_mfma_f32_32x32x2_f32   (exact fp32 fma chain, 157 TFLOP/s peak)
// dtype bf16   : v_mfma_f32_32x32x16_bf16 (fp32 accumulate, 2.5 PFLOP/s peak)
#include <stdio.h>

#include "conv_common.h"

namespace dmme {

template <typename T, int TAPS, int BM, int BN, bool ACC3 = false>
__global__ void __launch_bounds__(256) conv_mfma_kernel(ConvArgs a, ConvTile g) {
    constexpr int KC = Frag<T>::KC, EPV = Frag<T>::EPV;
    constexpr int MI = BM / 64, NI = BN / 64;  // 32x32 tiles per wave (2x2 waves)
    constexpr int KS = (TAPS == 9) ? 3 : 1, PAD = (TAPS == 9) ? 1 : 0;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    char* ldsA = lds;
    char* ldsB = lds + (size_t)g.a_rows * ROW_PITCH;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm0 = (wave >> 1) * (BM / 2), wn0 = (wave & 1) * (BN / 2);
    const int r = lane & 31, h = lane >> 5;

    const int tile_n = blockIdx.x % g.tiles_n, tile_m = blockIdx.x / g.tiles_n;
    const int tx_blk = tile_m % g.tiles_x, ty_blk = (tile_m / g.tiles_x) % g.tiles_y;
    const int n0 = (tile_m / (g.tiles_x * g.tiles_y)) * g.TN;
    const int oy0 = ty_blk * g.TH, ox0 = tx_blk * g.TW;
    const int co0 = tile_n * BN;

    const int Cin = a.C1 + a.C2;
    const int Hv = a.up ? 2 * a.Hin : a.Hin, Wv = a.up ? 2 * a.Win : a.Win;
    const int iy0 = oy0 * a.stride - PAD, ix0 = ox0 * a.stride - PAD;
    const int shTW = __builtin_ctz(g.TW), shTH = __builtin_ctz(g.TH);  // tile extents are powers of two
    const int mTW = g.TW - 1, mTH = g.TH - 1;

    // LDS row of this lane's output pixel (tap 0) for each M sub-tile
    int a_row[MI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
        const int m = wm0 + mi * 32 + r;
        const int tx = m & mTW, ty = (m >> shTW) & mTH, tn = m >> (shTW + shTH);
        a_row[mi] = (tn * g.HH + ty * a.stride) * g.HWd + tx * a.stride;
    }

    f32x16 acc[MI][NI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int j = 0; j < 16; ++j) acc[mi][ni][j] = 0.f;

    const int a_units = g.a_rows * 8;
    const int halo_px = g.HH * g.HWd;
    const T* wbase = (const T*)a.w;

    for (int c0 = 0; c0 < Cin; c0 += KC) {
        __syncthreads();  // every wave finished reading A and B of the previous chunk
        // ---- stage the activation halo tile for channels [c0, c0+KC) ----
        const bool second = c0 >= a.C1;
        const T* sbase = second ? (const T*)a.src2 : (const T*)a.src1;
        const int Cs = second ? a.C2 : a.C1;
        const int cs0 = second ? c0 - a.C1 : c0;
        for (int u = tid; u < a_units; u += 256) {
            const int row = u >> 3, cu = u & 7;
            const int tn = row / halo_px, rem = row % halo_px;
            const int hy = rem / g.HWd, hx = rem % g.HWd;
            const int n = n0 + tn, iy = iy0 + hy, ix = ix0 + hx;
            uint4 val = make_uint4(0u, 0u, 0u, 0u);
            if (n < a.N && iy >= 0 && iy < Hv && ix >= 0 && ix < Wv && !(a.up == 2 && ((iy | ix) & 1))) {
                const int sy = a.up ? (iy >> 1) : iy, sx = a.up ? (ix >> 1) : ix;
                const int64_t pix = ((int64_t)n * a.Hin + sy) * a.Win + sx;
                val = *reinterpret_cast<const uint4*>(sbase + pix * Cs + cs0 + cu * EPV);
                const int64_t so = (int64_t)n * Cin + c0 + cu * EPV;
                val = prologue_vec<T>(val, a.scale ? a.scale + so : nullptr, a.scale ? a.shift + so : nullptr,
                                      a.dmask ? a.dmask + so : nullptr, a.pro_silu);
            }
            *reinterpret_cast<uint4*>(ldsA + row * ROW_PITCH + cu * 16) = val;
        }
#pragma unroll 1
        for (int tap = 0; tap < TAPS; ++tap) {
            if (tap > 0) __syncthreads();  // previous tap's filter tile no longer read
            // ---- stage the filter tile [BN couts][KC cin] of this tap ----
            for (int u = tid; u < BN * 8; u += 256) {
                const int row = u >> 3, cu = u & 7;
                const int co = co0 + row;
                uint4 val = make_uint4(0u, 0u, 0u, 0u);
                if (co < a.Cout)
                    val = *reinterpret_cast<const uint4*>(wbase + ((int64_t)co * TAPS + tap) * Cin + c0 + cu * EPV);
                *reinterpret_cast<uint4*>(ldsB + row * ROW_PITCH + cu * 16) = val;
            }
            __syncthreads();
            const int tap_off = (tap / KS) * g.HWd + (tap % KS);
#pragma unroll
            for (int kg = 0; kg < 4; ++kg) {
                uint4 af[MI], bfr[NI];
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
                    af[mi] = *reinterpret_cast<const uint4*>(ldsA + (a_row[mi] + tap_off) * ROW_PITCH + kg * 32 + h * 16);
#pragma unroll
                for (int ni = 0; ni < NI; ++ni)
                    bfr[ni] = *reinterpret_cast<const uint4*>(ldsB + (wn0 + ni * 32 + r) * ROW_PITCH + kg * 32 + h * 16);
                mma_tile<typename MmaTag<T, ACC3>::type, MI, NI>(af, bfr, acc);
            }
        }
    }

    __syncthreads();  // operand tiles are dead: LDS becomes the output staging image
    // ---- epilogue: + bias + time embedding + residual, store ----
    auto pix_of = [&](int m) -> int {
        const int tx = m & mTW, ty = (m >> shTW) & mTH, tn = m >> (shTW + shTH);
        const int n = n0 + tn;
        return n < a.N ? (n * a.Hout + oy0 + ty) * a.Wout + ox0 + tx : -1;
    };
    conv_epilogue<T, BM, BN, MI, NI>(a, acc, co0, wn0, r, h, wm0, n0, g.TN, pix_of, reinterpret_cast<float*>(lds), ty_blk * g.tiles_x + tx_blk);
}

bool conv_mfma_supported(int dtype, const ConvArgs& a) {
    if (a.mix) return false;  // (mixed-precision convs have their own kernels: conv_pipe_supported / conv_out_thin_supported)
    const int KC = is16(dtype) ? 64 : 32;
    if (a.in_nchw) return false;
    if ((int64_t)a.N * a.Hout * a.Wout * a.Cout >= (1ll << 31)) return false;  // 32-bit offsets in the epilogue
    if (a.taps != 9 && a.taps != 1) return false;
    if (a.taps == 1 && (a.stride != 1 || a.up)) return false;
    if (a.stride != 1 && a.stride != 2) return false;
    if (a.C1 % KC || a.C2 % KC || a.C1 == 0) return false;
    if (a.res2) return false;  // two-source residual only in the generic kernel
    if (a.res1 && a.R1 != a.Cout) return false;
    if (a.Cout < 32 && !a.out_nchw) return false;  // the 3-channel output conv runs on a zero-padded cout tile
    ConvTile g;
    if (!make_tile(a, 64, 64, g)) return false;
    if (tile_lds(g, 64) > 64 * 1024) return false;
    return true;
}

static const int kCand[3][2] = {{128, 128}, {128, 64}, {64, 64}};

// pick the largest tile that still yields enough workgroups to fill 256 CUs twice
static int pick_tile(const ConvArgs& a, ConvTile& g) {
    int pick = -1;
    for (int i = 0; i < 3; ++i) {
        ConvTile t;
        if (!make_tile(a, kCand[i][0], kCand[i][1], t)) continue;
        if (tile_lds(t, kCand[i][1]) > 64 * 1024) continue;
        if (a.Cout <= 64 && kCand[i][1] > 64) continue;
        pick = i;
        g = t;
        if ((int64_t)t.tiles_m * t.tiles_n >= min_wgs()) break;
    }
    return pick;
}

template <typename T, int TAPS, bool ACC3 = false>
static int launch_sized(const ConvArgs& a, hipStream_t s) {
    ConvTile g{};
    const int pick = pick_tile(a, g);
    DMME_REQUIRE(pick >= 0, DMME_ERR_UNSUPPORTED, "conv_mfma: no tile fits (H=%d W=%d)", a.Hout, a.Wout);
    const dim3 grid((unsigned)(g.tiles_m * g.tiles_n));
    size_t lds = tile_lds(g, kCand[pick][1]);
    const size_t stage = (size_t)kCand[pick][0] * kCand[pick][1] * sizeof(float);  // epilogue staging image
    if (lds < stage) lds = stage;
    switch (pick) {
        case 0: hipLaunchKernelGGL((conv_mfma_kernel<T, TAPS, 128, 128, ACC3>), grid, dim3(256), lds, s, a, g); break;
        case 1: hipLaunchKernelGGL((conv_mfma_kernel<T, TAPS, 128, 64, ACC3>), grid, dim3(256), lds, s, a, g); break;
        default: hipLaunchKernelGGL((conv_mfma_kernel<T, TAPS, 64, 64, ACC3>), grid, dim3(256), lds, s, a, g); break;
    }
    DMME_CHECK_LAUNCH();
    return DMME_OK;
}

bool conv_pipe_stats_query(int dtype, const ConvArgs& a, int cg, int* tiles, int* px);

bool conv_stats_query(int dtype, const ConvArgs& a, int cg, int* tiles, int* px) {
    if (conv1x1_pipe_supported(dtype, a)) return conv1x1_stats_query(dtype, a, cg, tiles, px);
    if (conv_pipe_supported(dtype, a)) return conv_pipe_stats_query(dtype, a, cg, tiles, px);
    if (!conv_mfma_supported(dtype, a)) return conv_in_stats_query(dtype, a, cg, tiles, px);
    ConvTile g{};
    const int pick = pick_tile(a, g);
    if (pick < 0) return false;
    if (!stats_tile_ok(a, g, kCand[pick][1], cg, is16(dtype) ? 8 : 4)) return false;
    *tiles = g.tiles_x * g.tiles_y;
    *px = kCand[pick][0];
    return true;
}

void conv_mfma_label(int dtype, const ConvArgs& a, char* buf, int cap) {
    ConvTile g{};
    const int pick = pick_tile(a, g);
    snprintf(buf, (size_t)cap, "conv_mfma_kernel<%s,%d,%d,%d>", dtype == DMME_BF16 ? "bf16" : dtype == DMME_F16 ? "f16" : a.x3 ? "float:bf16x3" : "float", a.taps,
             pick >= 0 ? kCand[pick][0] : 0, pick >= 0 ? kCand[pick][1] : 0);
}

int launch_conv_mfma(int dtype, const ConvArgs& a, hipStream_t s) {
    DMME_REQUIRE(conv_mfma_supported(dtype, a), DMME_ERR_UNSUPPORTED, "conv_mfma: unsupported shape");
    if (dtype == DMME_BF16) return a.taps == 9 ? launch_sized<bf16, 9>(a, s) : launch_sized<bf16, 1>(a, s);
    if (dtype == DMME_F16) return a.taps == 9 ? launch_sized<f16, 9>(a, s) : launch_sized<f16, 1>(a, s);
    if (a.x3) return a.taps == 9 ? launch_sized<float, 9, true>(a, s) : launch_sized<float, 1, true>(a, s);
    return a.taps == 9 ? launch_sized<float, 9>(a, s) : launch_sized<float, 1>(a, s);
}

}  // namespace dmme
