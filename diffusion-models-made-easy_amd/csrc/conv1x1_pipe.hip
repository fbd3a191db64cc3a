// Software-pipelined 1x1 convolution (= GEMM over flattened pixels) for the matrix cores:
//   D[pixel][cout] = sum_cin act(A)[pixel][cin] * W[cout][cin]   (+ bias, + residual)
// replaces nn.Conv2d(kernel_size=1) of Attention.qkv_proj / proj and ResBlock.residual
// (models/ddpm.py:51-52,109) and the GroupNorm apply in front of qkv_proj.
//
// Same operand layout, swizzled 128-byte LDS rows, fused prologue and LDS-staged vector epilogue
// (incl. fused GroupNorm partials) as conv_pipe.hip.  Schedule: one barrier interval covers TWO
// 128-byte Cin chunks of both operands (32 MFMAs per wave for a 128x128 tile); the next interval's
// activation and weight tiles are loaded into registers before the MFMAs and written to LDS after.
#include <stdio.h>

#include "conv_common.h"

namespace dmme {

__device__ __forceinline__ int swz1(int row, int chunk) { return row * ROW_DATA + ((chunk ^ ((row >> 1) & 7)) << 4); }

template <typename T, int BM, int BN, bool ACC3 = false>
__global__ void __launch_bounds__(256, 2) conv1x1_pipe_kernel(ConvArgs a, int HW, int tiles_n, int xcd_order) {
    constexpr int KC = Frag<T>::KC, EPV = Frag<T>::EPV;
    constexpr int MI = BM / 64, NI = BN / 64;
    constexpr int UA = BM / 32, UB = BN / 32;  // 16-byte units per thread per chunk
    extern __shared__ __attribute__((aligned(16))) char lds[];
    char* ldsA = lds;                          // [2 chunks][BM rows][128 B]
    char* ldsB = lds + 2 * BM * ROW_DATA;      // [2 chunks][BN rows][128 B]
    // [2][Cin] scale / shift rows (ConvArgs::gni only), behind whichever is larger: the operand buffers or the epilogue's staging image
    float* gni_par = reinterpret_cast<float*>(lds + (2 * (BM + BN) * ROW_DATA > BM * BN * 4 ? 2 * (BM + BN) * ROW_DATA : BM * BN * 4));

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm0 = (wave >> 1) * (BM / 2), wn0 = (wave & 1) * (BN / 2);
    const int r = lane & 31, h = lane >> 5;
    const int cu = tid & 7, urow = tid >> 3;
    // The tiles_n cout tiles of one pixel tile read the same activations.  Workgroup L runs on XCD L % 8 and every XCD has its own
    // L2, so in plain order those re-reads go to HBM once per XCD; xcd_order (pixel tiles a multiple of 8) gives the cout tiles of a
    // pixel tile consecutive slots on ONE XCD (qkv: 6 tiles, 16.8 MB of activations fetched once instead of up to six times).
    int tile_n = blockIdx.x % tiles_n, tile_m = blockIdx.x / tiles_n;
    if (xcd_order) {
        const int x = blockIdx.x & 7, j = blockIdx.x >> 3;
        tile_n = j % tiles_n;
        tile_m = (j / tiles_n) * 8 + x;
    }
    const int p0 = tile_m * BM, co0 = tile_n * BN;
    const int Cin = a.C1 + a.C2;
    const int Mtot = a.N * HW;

    int a_n[UA];  // image index of the unit's pixel (scale/shift row), -1: past the last pixel
#pragma unroll
    for (int i = 0; i < UA; ++i) {
        const int p = p0 + urow + 32 * i;
        a_n[i] = p < Mtot ? p / HW : -1;
    }
    int b_off[UB];
#pragma unroll
    for (int k = 0; k < UB; ++k) {
        const int co = co0 + urow + 32 * k;
        b_off[k] = co < a.Cout ? co * Cin + cu * EPV : -1;
    }
    int a_base[MI], a_swz[MI], b_base[NI], b_swz[NI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
        const int row = wm0 + mi * 32 + r;
        a_base[mi] = row * ROW_DATA;
        a_swz[mi] = (row >> 1) & 7;
    }
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
        const int row = wn0 + ni * 32 + r;
        b_base[ni] = row * ROW_DATA;
        b_swz[ni] = (row >> 1) & 7;
    }
    f32x16 acc[MI][NI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int j = 0; j < 16; ++j) acc[mi][ni][j] = 0.f;

    uint4 areg[2][UA], breg[2][UB];
    const T* wbase = (const T*)a.w;
    const int nchunks = Cin / KC;

    auto load_step = [&](int ch0) {  // chunks ch0, ch0+1
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int c0 = (ch0 + c) * KC;
            if (c0 >= Cin) continue;
            const bool second = c0 >= a.C1;
            const T* sbase = second ? (const T*)a.src2 : (const T*)a.src1;
            const int Cs = second ? a.C2 : a.C1;
            const int cs = (second ? c0 - a.C1 : c0) + cu * EPV;
#pragma unroll
            for (int i = 0; i < UA; ++i)
                if (a_n[i] >= 0) areg[c][i] = *reinterpret_cast<const uint4*>(sbase + (int64_t)(p0 + urow + 32 * i) * Cs + cs);
#pragma unroll
            for (int k = 0; k < UB; ++k) {
                breg[c][k] = make_uint4(0u, 0u, 0u, 0u);
                if (b_off[k] >= 0) breg[c][k] = *reinterpret_cast<const uint4*>(wbase + b_off[k] + c0);
            }
        }
    };
    auto store_step = [&](int ch0) {
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int c0 = (ch0 + c) * KC;
            if (c0 >= Cin) continue;
            const int cc = c0 + cu * EPV;
#pragma unroll
            for (int i = 0; i < UA; ++i) {
                uint4 val = make_uint4(0u, 0u, 0u, 0u);
                if (a_n[i] >= 0) {
                    const int so = a_n[i] * Cin + cc;
                    if constexpr (sizeof(T) == 2) {
                        if (a.has_gni)
                            val = prologue_vec_ldsrows<T>(areg[c][i], gni_par + cc, gni_par + Cin + cc, a.dmask ? a.dmask + so : nullptr, a.pro_silu);
                        else
                            val = prologue_vec<T>(areg[c][i], a.scale ? a.scale + so : nullptr, a.scale ? a.shift + so : nullptr,
                                                  a.dmask ? a.dmask + so : nullptr, a.pro_silu);
                    } else {
                        val = prologue_vec<T>(areg[c][i], a.scale ? a.scale + so : nullptr, a.scale ? a.shift + so : nullptr,
                                              a.dmask ? a.dmask + so : nullptr, a.pro_silu);
                    }
                }
                *reinterpret_cast<uint4*>(ldsA + c * BM * ROW_DATA + swz1(urow + 32 * i, cu)) = val;
            }
#pragma unroll
            for (int k = 0; k < UB; ++k) *reinterpret_cast<uint4*>(ldsB + c * BN * ROW_DATA + swz1(urow + 32 * k, cu)) = breg[c][k];
        }
    };

    load_step(0);
    // the norm in front of this conv finished HERE (ConvArgs::gni; host-checked: the tile lies inside one image): scale / shift rows of
    // that image into LDS behind the operand buffers, from the producers' partials
    if constexpr (sizeof(T) == 2) {
        if (a.has_gni) {
            const int n = p0 / HW;
            for (int c = tid; c < Cin; c += 256) {
                float sc, sh;
                gn_in_scale_shift<8>(a, n, c, Cin, tile_n == 0 && p0 == n * HW, sc, sh);
                gni_par[c] = sc;
                gni_par[Cin + c] = sh;
            }
            __syncthreads();
        }
    }
    store_step(0);
    __syncthreads();
#pragma unroll 1
    for (int ch0 = 0; ch0 < nchunks; ch0 += 2) {
        const bool more = ch0 + 2 < nchunks;
        if (more) load_step(ch0 + 2);
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            if (ch0 + c >= nchunks) break;
#pragma unroll
            for (int kg = 0; kg < 4; ++kg) {
                const int cidx = kg * 2 + h;
                uint4 af[MI], bfr[NI];
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
                    af[mi] = *reinterpret_cast<const uint4*>(ldsA + c * BM * ROW_DATA + a_base[mi] + ((cidx ^ a_swz[mi]) << 4));
#pragma unroll
                for (int ni = 0; ni < NI; ++ni)
                    bfr[ni] = *reinterpret_cast<const uint4*>(ldsB + c * BN * ROW_DATA + b_base[ni] + ((cidx ^ b_swz[ni]) << 4));
                mma_tile<typename MmaTag<T, ACC3>::type, MI, NI>(af, bfr, acc);
            }
        }
        __syncthreads();
        if (more) store_step(ch0 + 2);
        __syncthreads();
    }

    auto pix_of = [&](int m) -> int { return p0 + m < Mtot ? p0 + m : -1; };
    // fused-statistics tiles never straddle an image (HW % BM == 0 is required for gn_part)
    conv_epilogue<T, BM, BN, MI, NI>(a, acc, co0, wn0, r, h, wm0, p0 / HW, HW >= BM ? 1 : 2, pix_of, reinterpret_cast<float*>(lds),
                                     (p0 % HW) / BM);
}

// ---- split-pass form (precision="fp16r32", ConvArgs::mix == 4): the 1x1 convs of the mode's fp32 level (the ResBlocks' residual convs,
// models/ddpm.py:109) - fp32 tensors in and out, every product as three fp16 MFMA passes.  Same tiles, LDS images and interval
// structure as the kernel above; a 128-byte LDS row is 32 input channels as [hi: 32 halves | lo: 32 halves] (as conv_pipe.hip's
// split form): the activation rows are split when they are staged (after the optional prologue, computed in fp32), the filter rows
// arrive packed that way (pack code 4).  Per chunk the six MFMA groups hi.hi + hi.lo + lo.hi of both 16-channel halves.
template <int BM, int BN>
__global__ void __launch_bounds__(256, 2) conv1x1_split_kernel(ConvArgs a, int HW, int tiles_n, int xcd_order) {
    typedef f16 T;
    constexpr int KCR = 32;  // input channels per 128-byte row
    constexpr int MI = BM / 64, NI = BN / 64;
    constexpr int UA = BM / 32, UB = BN / 32;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    char* ldsA = lds;
    char* ldsB = lds + 2 * BM * ROW_DATA;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm0 = (wave >> 1) * (BM / 2), wn0 = (wave & 1) * (BN / 2);
    const int r = lane & 31, h = lane >> 5;
    const int cu = tid & 7, urow = tid >> 3;
    int tile_n = blockIdx.x % tiles_n, tile_m = blockIdx.x / tiles_n;
    if (xcd_order) {
        const int x = blockIdx.x & 7, j = blockIdx.x >> 3;
        tile_n = j % tiles_n;
        tile_m = (j / tiles_n) * 8 + x;
    }
    const int p0 = tile_m * BM, co0 = tile_n * BN;
    const int Cin = a.C1 + a.C2;
    const int Mtot = a.N * HW;
    int a_n[UA];
#pragma unroll
    for (int i = 0; i < UA; ++i) {
        const int p = p0 + urow + 32 * i;
        a_n[i] = p < Mtot ? p / HW : -1;
    }
    int64_t b_off[UB];
#pragma unroll
    for (int k = 0; k < UB; ++k) {
        const int co = co0 + urow + 32 * k;
        b_off[k] = co < a.Cout ? (int64_t)co * 2 * Cin + cu * 8 : -1;
    }
    int a_base[MI], a_swz[MI], b_base[NI], b_swz[NI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
        const int row = wm0 + mi * 32 + r;
        a_base[mi] = row * ROW_DATA;
        a_swz[mi] = (row >> 1) & 7;
    }
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
        const int row = wn0 + ni * 32 + r;
        b_base[ni] = row * ROW_DATA;
        b_swz[ni] = (row >> 1) & 7;
    }
    f32x16 acc[MI][NI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int j = 0; j < 16; ++j) acc[mi][ni][j] = 0.f;
    uint4 areg[2][UA], breg[2][UB];
    const T* wbase = (const T*)a.w;
    const int nchunks = Cin / KCR;  // (even: Cin % 64 == 0, host-checked)
    auto load_step = [&](int ch0) {
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int c0 = (ch0 + c) * KCR;
            const bool second = c0 >= a.C1;
            const float* sbase = second ? (const float*)a.src2 : (const float*)a.src1;
            const int Cs = second ? a.C2 : a.C1;
            const int cs = (second ? c0 - a.C1 : c0) + cu * 4;
#pragma unroll
            for (int i = 0; i < UA; ++i)
                if (a_n[i] >= 0) areg[c][i] = *reinterpret_cast<const uint4*>(sbase + (int64_t)(p0 + urow + 32 * i) * Cs + cs);
#pragma unroll
            for (int k = 0; k < UB; ++k) {
                breg[c][k] = make_uint4(0u, 0u, 0u, 0u);
                if (b_off[k] >= 0) breg[c][k] = *reinterpret_cast<const uint4*>(wbase + b_off[k] + (ch0 + c) * 64);
            }
        }
    };
    auto store_step = [&](int ch0) {
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int cc = (ch0 + c) * KCR + cu * 4;
#pragma unroll
            for (int i = 0; i < UA; ++i) {
                uint2 hw = make_uint2(0u, 0u), lw = make_uint2(0u, 0u);
                if (a_n[i] >= 0) {
                    f32x4 y = __builtin_bit_cast(f32x4, areg[c][i]);
                    const int so = a_n[i] * Cin + cc;
                    if (a.scale) {
                        const f32x4 sc4 = *reinterpret_cast<const f32x4*>(a.scale + so), sh4 = *reinterpret_cast<const f32x4*>(a.shift + so);
#pragma unroll
                        for (int e = 0; e < 4; ++e) y[e] = fmaf(y[e], sc4[e], sh4[e]);
                    }
                    if (a.pro_silu) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) y[e] = silu_fast(y[e]);
                    }
                    if (a.dmask) {
                        const f32x4 m4 = *reinterpret_cast<const f32x4*>(a.dmask + so);
#pragma unroll
                        for (int e = 0; e < 4; ++e) y[e] *= m4[e];
                    }
                    typedef f16 hx4 __attribute__((ext_vector_type(4)));
                    hx4 hi, lo;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        hi[e] = (f16)y[e];
                        lo[e] = (f16)(y[e] - (float)hi[e]);
                    }
                    hw = __builtin_bit_cast(uint2, hi);
                    lw = __builtin_bit_cast(uint2, lo);
                }
                // channels 4 cu .. 4 cu + 3: 8 bytes of the hi half (16-byte piece cu >> 1) and of the lo half (piece 4 + (cu >> 1))
                char* rowp = ldsA + c * BM * ROW_DATA;
                *reinterpret_cast<uint2*>(rowp + swz1(urow + 32 * i, cu >> 1) + (cu & 1) * 8) = hw;
                *reinterpret_cast<uint2*>(rowp + swz1(urow + 32 * i, 4 + (cu >> 1)) + (cu & 1) * 8) = lw;
            }
#pragma unroll
            for (int k = 0; k < UB; ++k) *reinterpret_cast<uint4*>(ldsB + c * BN * ROW_DATA + swz1(urow + 32 * k, cu)) = breg[c][k];
        }
    };
    load_step(0);
    store_step(0);
    __syncthreads();
#pragma unroll 1
    for (int ch0 = 0; ch0 < nchunks; ch0 += 2) {
        const bool more = ch0 + 2 < nchunks;
        if (more) load_step(ch0 + 2);
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            uint4 af[4][MI], bfr[4][NI];  // k-groups 0, 1: hi halves of the chunk's channels 0-15 / 16-31; 2, 3: their lo halves
#pragma unroll
            for (int kg = 0; kg < 4; ++kg) {
                const int cidx = kg * 2 + h;
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
                    af[kg][mi] = *reinterpret_cast<const uint4*>(ldsA + c * BM * ROW_DATA + a_base[mi] + ((cidx ^ a_swz[mi]) << 4));
#pragma unroll
                for (int ni = 0; ni < NI; ++ni)
                    bfr[kg][ni] = *reinterpret_cast<const uint4*>(ldsB + c * BN * ROW_DATA + b_base[ni] + ((cidx ^ b_swz[ni]) << 4));
            }
#pragma unroll
            for (int half = 0; half < 2; ++half)
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni) {  // the small terms first
                        mma16<T>(af[2 + half][mi], bfr[half][ni], acc[mi][ni]);
                        mma16<T>(af[half][mi], bfr[2 + half][ni], acc[mi][ni]);
                        mma16<T>(af[half][mi], bfr[half][ni], acc[mi][ni]);
                    }
        }
        __syncthreads();
        if (more) store_step(ch0 + 2);
        __syncthreads();
    }
    auto pix_of = [&](int m) -> int { return p0 + m < Mtot ? p0 + m : -1; };
    conv_epilogue<float, BM, BN, MI, NI>(a, acc, co0, wn0, r, h, wm0, p0 / HW, HW >= BM ? 1 : 2, pix_of, reinterpret_cast<float*>(lds), (p0 % HW) / BM);
}

static bool split1_ok(const ConvArgs& a) {
    const int Cin = a.C1 + a.C2;
    return a.mix == 4 && a.taps == 1 && a.stride == 1 && !a.up && !a.in_nchw && !a.out_nchw && !a.out_silu && !a.res2 && !a.has_gni && !a.n_gno && Cin % 64 == 0 &&
           a.C1 % 32 == 0 && a.Cout % 4 == 0 && (!a.res1 || a.R1 == a.Cout) && (!a.tproj || a.nt == 1) && (int64_t)a.Cout * 2 * Cin < (1ll << 31) &&
           (int64_t)a.N * a.Hout * a.Wout * (Cin > a.Cout ? Cin : a.Cout) < (1ll << 31);
}
static int launch1_split(const ConvArgs& a, hipStream_t s) {
    constexpr int BM = 128, BN = 128;
    const int HW = a.Hout * a.Wout;
    const int64_t M = (int64_t)a.N * HW;
    const int tiles_n = (a.Cout + BN - 1) / BN;
    const int64_t tiles_m = (M + BM - 1) / BM;
    const int xcd_order = (tiles_n > 1 && tiles_m % 8 == 0) ? 1 : 0;
    const size_t lds = (size_t)BM * BN * 4;  // the fp32 epilogue image (>= the operand buffers' 64 KB)
    static bool attr = false;
    if (!attr) {
        DMME_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(conv1x1_split_kernel<BM, BN>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
        attr = true;
    }
    hipLaunchKernelGGL((conv1x1_split_kernel<BM, BN>), dim3((unsigned)(tiles_m * tiles_n)), dim3(256), lds, s, a, HW, tiles_n, xcd_order);
    DMME_CHECK_LAUNCH();
    return DMME_OK;
}

static const int k1Cand[3][2] = {{128, 128}, {128, 64}, {64, 64}};

static int pick1(const ConvArgs& a) {
    const int64_t M = (int64_t)a.N * a.Hout * a.Wout;
    int pick = -1;
    for (int i = 0; i < 3; ++i) {
        if (a.Cout <= 64 && k1Cand[i][1] > 64) continue;
        pick = i;
        const int64_t wgs = ((M + k1Cand[i][0] - 1) / k1Cand[i][0]) * ((a.Cout + k1Cand[i][1] - 1) / k1Cand[i][1]);
        if (wgs >= min_wgs()) break;
    }
    return pick;
}

bool conv1x1_pipe_supported(int dtype, const ConvArgs& a) {
    if (a.mix) return dtype == DMME_F16 && split1_ok(a);
    if (!conv_mfma_supported(dtype, a)) return false;
    if (a.taps != 1 || a.stride != 1 || a.up) return false;
    if (a.tproj && a.nt != 1) return false;  // per-image time rows need the image-aligned tiles of the 3x3 kernels
    if ((int64_t)a.Cout * (a.C1 + a.C2) >= (1ll << 31)) return false;
    return pick1(a) >= 0;
}

// can the tiled kernel finish the norm in front of this conv itself (its tile inside one image, LDS budget)?
bool conv1x1_pipe_gn_in_ok(int dtype, const ConvArgs& a) {
    if (!is16(dtype) || debug_route("no_gn_in_pipe") || debug_route("no_gn_in_pipe1") || !conv1x1_pipe_supported(dtype, a)) return false;
    const int pick = pick1(a);
    if (pick < 0) return false;
    const int BM = k1Cand[pick][0], BN = k1Cand[pick][1], HW = a.Hout * a.Wout;
    size_t lds = (size_t)2 * (BM + BN) * ROW_DATA;
    if (lds < (size_t)BM * BN * 4) lds = (size_t)BM * BN * 4;
    return HW % BM == 0 && lds + (size_t)2 * (a.C1 + a.C2) * 4 <= 80 * 1024;
}

bool conv1x1_stats_query(int dtype, const ConvArgs& a, int cg, int* tiles, int* px) {
    {
        ConvArgs b = a;  // the activation-stationary kernel takes this conv (with or without statistics): its tiles
        b.gn_part = nullptr;
        if (conv1x1_as_supported(dtype, b)) return conv1x1_as_stats_query(dtype, a, cg, tiles, px);
    }
    if (a.mix) {  // 128 x 128 tiles, fp32 output vectors of 4 channels
        const int HWm = a.Hout * a.Wout;
        if (!split1_ok(a) || HWm % 128 || a.Cout % 128 || cg % 4 || 128 % cg) return false;
        *tiles = HWm / 128;
        *px = 128;
        return true;
    }
    const int pick = pick1(a);
    if (pick < 0) return false;
    const int BM = k1Cand[pick][0], BN = k1Cand[pick][1], HW = a.Hout * a.Wout, vec = is16(dtype) ? 8 : 4;
    if (HW % BM) return false;
    if (a.out_silu || a.out_nchw || a.Cout % BN || a.Cout % vec || !(cg % vec == 0 || (vec == 8 && cg == 4)) || BN % cg) return false;
    *tiles = HW / BM;
    *px = BM;
    return true;
}

template <typename T, bool ACC3 = false>
static int launch1_t(const ConvArgs& a, hipStream_t s) {
    const int pick = pick1(a);
    DMME_REQUIRE(pick >= 0, DMME_ERR_UNSUPPORTED, "conv1x1_pipe: no tile");
    const int BM = k1Cand[pick][0], BN = k1Cand[pick][1], HW = a.Hout * a.Wout;
    const int64_t M = (int64_t)a.N * HW;
    const int tiles_n = (a.Cout + BN - 1) / BN;
    const int64_t tiles_m = (M + BM - 1) / BM;
    const dim3 grid((unsigned)(tiles_m * tiles_n));
    const bool xcd_off = (debug_route("no_xcd_order") != 0);
    const int xcd_order = (!xcd_off && tiles_n > 1 && tiles_m % 8 == 0) ? 1 : 0;
    size_t lds = (size_t)2 * (BM + BN) * ROW_DATA;
    if (lds < (size_t)BM * BN * 4) lds = (size_t)BM * BN * 4;
    if (a.has_gni) lds += (size_t)2 * (a.C1 + a.C2) * 4;
    static bool attr_done[3] = {false, false, false};
    int rc = DMME_OK;
#define DMME_C1_CASE(IDX, BM_, BN_)                                                                                                   \
    case IDX:                                                                                                                         \
        if (!attr_done[IDX]) {                                                                                                        \
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv1x1_pipe_kernel<T, BM_, BN_, ACC3>),                 \
                                               hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);                                \
            if (e != hipSuccess) rc = DMME_ERR_HIP;                                                                                   \
            attr_done[IDX] = rc == DMME_OK;                                                                                           \
        }                                                                                                                             \
        if (rc == DMME_OK) hipLaunchKernelGGL((conv1x1_pipe_kernel<T, BM_, BN_, ACC3>), grid, dim3(256), lds, s, a, HW, tiles_n, xcd_order); \
        break;
    switch (pick) {
        DMME_C1_CASE(0, 128, 128)
        DMME_C1_CASE(1, 128, 64)
        DMME_C1_CASE(2, 64, 64)
    }
#undef DMME_C1_CASE
    DMME_REQUIRE(rc == DMME_OK, DMME_ERR_HIP, "conv1x1_pipe: hipFuncSetAttribute failed");
    DMME_CHECK_LAUNCH();
    return DMME_OK;
}

int launch_conv1x1_pipe(int dtype, const ConvArgs& a, hipStream_t s) {
    DMME_REQUIRE(conv1x1_pipe_supported(dtype, a), DMME_ERR_UNSUPPORTED, "conv1x1_pipe: unsupported shape");
    if (a.mix) return launch1_split(a, s);
    if (conv1x1_as_supported(dtype, a)) return launch_conv1x1_as(a, s);
    if (dtype == DMME_BF16) return launch1_t<bf16>(a, s);
    if (dtype == DMME_F16) return launch1_t<f16>(a, s);
    return a.x3 ? launch1_t<float, true>(a, s) : launch1_t<float>(a, s);
}

void conv1x1_pipe_label(int dtype, const ConvArgs& a, char* buf, int cap) {
    if (a.mix) {
        snprintf(buf, (size_t)cap, "conv1x1_split_kernel<128,128>");
        return;
    }
    if (conv1x1_as_supported(dtype, a)) {
        snprintf(buf, (size_t)cap, "conv1x1_as_kernel<%d>", (a.C1 + a.C2) / 64);
        return;
    }
    const int pick = pick1(a);
    snprintf(buf, (size_t)cap, "conv1x1_pipe_kernel<%s,%d,%d>", dtype == DMME_BF16 ? "bf16" : dtype == DMME_F16 ? "f16" : a.x3 ? "float:bf16x3" : "float", pick >= 0 ? k1Cand[pick][0] : 0,
             pick >= 0 ? k1Cand[pick][1] : 0);
}

}  // namespace dmme
