// The network's LAST convolution: Cin = 128 (64..512) channels -> 3 (IDDPM: 6) outputs, 3x3, GroupNorm + SiLU in front, NCHW fp32 out
// (models/ddpm.py:287-291 output_conv).  On the tiled implicit-GEMM kernels it pads its 3 couts to a 64-wide tile: 21 x the matrix
// work and an input tile staged nine-tap-wise for it - 60 us for a layer whose whole input is 33.5 MB.
//
// Here the nine taps become COLUMNS of one GEMM: z[pixel][(cout, tap)] = sum_c act(x)[pixel][c] * w[cout][tap][c] has Cout * 9 = 27
// (54) columns - one (two) 32-wide MFMA tiles - and every input element is normalised, activated and multiplied exactly once;
// then out[cout][y][x] = bias + sum_taps z[(y + dy, x + dx)][(cout, tap)] is a 27-term gather out of LDS.  A workgroup takes a band
// of R image rows: z for its R + 2 rows (320 pixels at 32 x 32), the A fragments straight from global memory (each element is used
// once: nothing to stage), the weights in registers.  HBM-bound: the input once (x 1.25 for the band halo), 1.5 MB out.
#include <stdio.h>

#include <type_traits>

#include "conv_common.h"

namespace dmme {

constexpr int kThinMaxC = 512;

// NT: 32-column tiles of (cout, tap) pairs (1: Cout <= 3, 2: Cout <= 7)
// SPLIT (precision="fp16r32", ConvArgs::mix == 3): the input tensor is fp32; GroupNorm + SiLU run in fp32 on it and every product is
// three fp16 MFMA passes over hi / lo halves of activation and weight (filter rows packed [Cin / 32][hi 32 | lo 32], pack code 4) -
// the layer is HBM-bound (its matrix work is 27 columns wide), so the two extra passes are free and only the input bytes double.
template <int NT, typename T = bf16, bool SPLIT = false>
__global__ void __launch_bounds__(256) conv_out_thin_kernel(ConvArgs a, int R) {
    typedef typename Vec8<T>::type bf16x8;  // (8 operands of the kernel's 16-bit type; the name predates precision="fp16")
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int C = a.C1, W = a.Wout, H = a.Hout, NCOL = a.Cout * 9;
    float* par = reinterpret_cast<float*>(lds);  // [2][C] scale, shift of the band's image
    float* z = par + 2 * C;                      // [(R + 2) * W][ZP]
    constexpr int ZP = 32 * NT + 1;              // odd pitch: the gather below walks pixels with the column fixed
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int bands = H / R, n = blockIdx.x / bands, y0 = (blockIdx.x % bands) * R;
    const int npx = (R + 2) * W, ntile = npx / 32;  // W % 32 == 0: a 32-pixel MFMA tile is part of one image row

    // weights as the B operand: column nc = (cout, tap) -> w[cout][tap][c], c contiguous; columns past NCOL are zero.  Eight k-steps
    // (128 channels) of them live in registers; wider inputs reload per 128-channel slab (L2 hits)
    const int ksteps = C / 16;
    const T* wb = (const T*)a.w;
    bf16x8 wfrag[NT][8];
    bf16x8 wlo[SPLIT ? NT : 1][SPLIT ? 8 : 1];  // (SPLIT: the lo halves of the same channels)
    auto load_w = [&](int kc) __attribute__((always_inline)) {
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int nc = 32 * t + r;
#pragma unroll
            for (int kg = 0; kg < 8; ++kg) {
                bf16x8 v, vl;
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = vl[e] = (T)0.f;
                if constexpr (SPLIT) {
                    const int c0 = (kc + kg) * 16 + 8 * h;  // eight channels inside one 32-channel block
                    if (kc + kg < ksteps && nc < NCOL) {
                        const T* wr = wb + (int64_t)nc * 2 * C + (c0 >> 5) * 64 + (c0 & 31);
                        v = *reinterpret_cast<const bf16x8*>(wr);
                        vl = *reinterpret_cast<const bf16x8*>(wr + 32);
                    }
                    wlo[t][kg] = vl;
                } else {
                    if (kc + kg < ksteps && nc < NCOL) v = *reinterpret_cast<const bf16x8*>(wb + (int64_t)nc * C + (kc + kg) * 16 + 8 * h);
                }
                wfrag[t][kg] = v;
            }
        }
    };
    const bool w_once = ksteps <= 8;
    if (w_once) load_w(0);
    for (int c = tid; c < C; c += 256) {
        float sc = 1.f, sh = 0.f;
        if (a.has_gni) {  // the norm in front of this conv is finished here, from its producers' partials (gn_in_scale_shift)
            gn_in_scale_shift(a, n, c, C, y0 == 0, sc, sh);
        } else if (a.scale) {
            sc = a.scale[(int64_t)n * C + c];
            sh = a.shift[(int64_t)n * C + c];
        }
        par[c] = sc;
        par[C + c] = sh;
    }
    __syncthreads();

    typedef typename std::conditional<SPLIT, float, T>::type TX;  // element type of the input tensor
    const TX* xb = (const TX*)a.src1 + (int64_t)n * H * W * C;
    for (int tile = wave; tile < ntile; tile += 4) {
        const int p = tile * 32 + r;               // pixel of the band incl. its halo rows
        const int yy = y0 - 1 + p / W, xx = p % W;  // image coordinates (yy may be -1 or H: zero padding)
        const bool in = yy >= 0 && yy < H;
        const TX* px = xb + (int64_t)((in ? yy : 0) * W + xx) * C + 8 * h;
        f32x16 acc[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int j = 0; j < 16; ++j) acc[t][j] = 0.f;
        for (int kc = 0; kc < ksteps; kc += 8) {
            if (!w_once) load_w(kc);
            if constexpr (SPLIT) {
                uint4 raw[8][2];  // eight fp32 channels per k-group
#pragma unroll
                for (int kg = 0; kg < 8; ++kg)
                    if (kc + kg < ksteps) {
                        raw[kg][0] = *reinterpret_cast<const uint4*>(px + (kc + kg) * 16);
                        raw[kg][1] = *reinterpret_cast<const uint4*>(px + (kc + kg) * 16 + 4);
                    }
#pragma unroll
                for (int kg = 0; kg < 8; ++kg) {
                    if (kc + kg >= ksteps) break;
                    const int c0 = (kc + kg) * 16 + 8 * h;
                    bf16x8 ahi, alo;
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        // (the prologue in fp32 with the hardware exp2 / rcp, as the 16-bit kernels' prologue_vec16: ~1 ulp each, four orders
                        // under the three-pass product's own error; the libm expf of prologue_vec<float> made this HBM-bound layer VALU-bound)
                        f32x4 y = __builtin_bit_cast(f32x4, raw[kg][q]);
                        if (a.scale) {
                            const f32x4 sc4 = *reinterpret_cast<const f32x4*>(par + c0 + 4 * q), sh4 = *reinterpret_cast<const f32x4*>(par + C + c0 + 4 * q);
#pragma unroll
                            for (int e = 0; e < 4; ++e) y[e] = fmaf(y[e], sc4[e], sh4[e]);
                        }
                        if (a.pro_silu) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) y[e] = silu_fast(y[e]);
                        }
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const T hv = (T)(in ? y[e] : 0.f);
                            ahi[4 * q + e] = hv;
                            alo[4 * q + e] = (T)((in ? y[e] : 0.f) - (float)hv);
                        }
                    }
#pragma unroll
                    for (int t = 0; t < NT; ++t) {  // the small terms first
                        mma16<T>(__builtin_bit_cast(uint4, alo), __builtin_bit_cast(uint4, wfrag[t][kg]), acc[t]);
                        mma16<T>(__builtin_bit_cast(uint4, ahi), __builtin_bit_cast(uint4, wlo[t][kg]), acc[t]);
                        mma16<T>(__builtin_bit_cast(uint4, ahi), __builtin_bit_cast(uint4, wfrag[t][kg]), acc[t]);
                    }
                }
            } else {
            uint4 raw[8];
#pragma unroll
            for (int kg = 0; kg < 8; ++kg)
                if (kc + kg < ksteps) raw[kg] = *reinterpret_cast<const uint4*>(px + (kc + kg) * 16);
#pragma unroll
            for (int kg = 0; kg < 8; ++kg) {
                if (kc + kg >= ksteps) break;
                const int c0 = (kc + kg) * 16 + 8 * h;
                uint4 av = prologue_vec<T>(raw[kg], a.scale ? par + c0 : nullptr, a.scale ? par + C + c0 : nullptr, nullptr, a.pro_silu);
                if (!in) av = make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
                for (int t = 0; t < NT; ++t) mma16<T>(av, __builtin_bit_cast(uint4, wfrag[t][kg]), acc[t]);
            }
            }
        }
        // D[row = pixel][col = (cout, tap)]: lane = column r, registers = pixel rows (j & 3) + 8 (j >> 2) + 4 h
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int j = 0; j < 16; ++j) z[(tile * 32 + (j & 3) + 8 * (j >> 2) + 4 * h) * ZP + 32 * t + r] = acc[t][j];
    }
    __syncthreads();
    // gather: one output pixel per thread and pass
    const int hw = H * W;
    float* out = (float*)a.dst + (int64_t)n * a.Cout * hw;
    for (int q = tid; q < R * W; q += 256) {
        const int oy = q / W, ox = q % W;
        for (int k = 0; k < a.Cout; ++k) {
            float s = a.bias ? a.bias[k] : 0.f;
#pragma unroll
            for (int t9 = 0; t9 < 9; ++t9) {
                const int xs = ox + t9 % 3 - 1;
                if (xs < 0 || xs >= W) continue;  // (rows outside the image were zeroed at the source)
                s += z[((oy + t9 / 3) * W + xs) * ZP + k * 9 + t9];
            }
            out[(int64_t)k * hw + (y0 + oy) * W + ox] = s;
        }
    }
}

static int thin_rows(const ConvArgs& a) {  // band height: the z image of R + 2 rows must fit in LDS beside the parameters
    const int NT = a.Cout * 9 <= 32 ? 1 : 2;
    // (the split-pass form has no other kernel to fall back on - precision="fp16r32" refuses a plan whose fp32 level has a conv without
    //  one - so it may take up to 80 KB, two workgroups per compute unit still: the six-cout output conv of the Improved-DDPM UNet on
    //  64-wide maps needs 66.6 KB for its smallest band)
    const size_t cap = (a.mix == 3 ? 80 : 60) * 1024;
    for (int R = 16; R >= 2; R >>= 1)
        if (a.Hout % R == 0 && ((R + 2) * a.Wout) % 32 == 0 && (size_t)(R + 2) * a.Wout * (32 * NT + 1) * 4 + (size_t)2 * a.C1 * 4 <= cap &&
            (int64_t)a.N * (a.Hout / R) >= 512)
            return R;
    for (int R = 2; R <= 16; R <<= 1)  // small batches: the smallest band that fits
        if (a.Hout % R == 0 && ((R + 2) * a.Wout) % 32 == 0 && (size_t)(R + 2) * a.Wout * (32 * NT + 1) * 4 + (size_t)2 * a.C1 * 4 <= cap) return R;
    return 0;
}

bool conv_out_thin_supported(int dtype, const ConvArgs& a) {
    const bool off = getenv("DMME_NO_CONV_THIN") != nullptr;
    if (a.mix && a.mix != 3) return false;
    if (off || !is16(dtype) || a.x3) return false;
    if (a.mix == 3 && (dtype != DMME_F16 || a.C1 % 32)) return false;  // (whole 32-channel blocks of hi / lo halves)
    if (a.taps != 9 || a.stride != 1 || a.up || a.C2 || a.in_nchw || !a.out_nchw || a.out_silu || a.tproj || a.res1 || a.dmask || a.gn_part || a.n_gno)
        return false;
    if (a.Cout * 9 > 64 || a.C1 % 16 || a.C1 > kThinMaxC || a.Wout % 32 || a.Hin != a.Hout || a.Win != a.Wout) return false;
    return thin_rows(a) > 0;
}

int launch_conv_out_thin(const ConvArgs& a, hipStream_t s) {
    const int R = thin_rows(a);
    DMME_REQUIRE(R > 0, DMME_ERR_UNSUPPORTED, "conv_out_thin: unsupported shape");
    const int NT = a.Cout * 9 <= 32 ? 1 : 2;
    const size_t lds = (size_t)(R + 2) * a.Wout * (32 * NT + 1) * 4 + (size_t)2 * a.C1 * 4;
    const dim3 grid((unsigned)(a.N * (a.Hout / R)));
    if (a.mix == 3) {
        static bool attr = false;
        if (!attr) {  // more than the default 64 KB of dynamic LDS
            DMME_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_out_thin_kernel<1, f16, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
            DMME_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_out_thin_kernel<2, f16, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
            attr = true;
        }
        if (NT == 1)
            hipLaunchKernelGGL((conv_out_thin_kernel<1, f16, true>), grid, dim3(256), lds, s, a, R);
        else
            hipLaunchKernelGGL((conv_out_thin_kernel<2, f16, true>), grid, dim3(256), lds, s, a, R);
    } else if (a.f16) {
        if (NT == 1)
            hipLaunchKernelGGL((conv_out_thin_kernel<1, f16>), grid, dim3(256), lds, s, a, R);
        else
            hipLaunchKernelGGL((conv_out_thin_kernel<2, f16>), grid, dim3(256), lds, s, a, R);
    } else if (NT == 1)
        hipLaunchKernelGGL((conv_out_thin_kernel<1, bf16>), grid, dim3(256), lds, s, a, R);
    else
        hipLaunchKernelGGL((conv_out_thin_kernel<2, bf16>), grid, dim3(256), lds, s, a, R);
    DMME_CHECK_LAUNCH();
    return DMME_OK;
}

}  // namespace dmme
