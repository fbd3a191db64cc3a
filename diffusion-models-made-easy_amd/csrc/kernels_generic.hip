// Shape-generic HIP kernels: any channel count / resolution, used for the layers the
// MFMA kernels do not cover (3-channel input/output convs, odd test configurations)
// and as a second, structurally independent implementation for parity tests.
// Also: time embedding, small linears, layout converters, the parameter re-packer.
#include <stdlib.h>

#include <algorithm>

#include "common.h"

namespace dmme {

// ------------------------------------------------------------------ generic convolution
// One thread per output element; replaces nn.Conv2d 3x3/1x1 (+ the fused prologue /
// epilogue described in DESIGN.md) for arbitrary shapes.
template <typename T>
__global__ void __launch_bounds__(256) conv_generic_kernel(ConvArgs a) {
    const int64_t total = (int64_t)a.N * a.Hout * a.Wout * a.Cout;
    const int Cin = a.C1 + a.C2;
    const int Hv = a.up ? 2 * a.Hin : a.Hin, Wv = a.up ? 2 * a.Win : a.Win;
    const int k = a.taps == 9 ? 3 : 1, pad = a.taps == 9 ? 1 : 0;
    for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < total;
         idx += (int64_t)gridDim.x * blockDim.x) {
        const int co = (int)(idx % a.Cout);
        int64_t p = idx / a.Cout;
        const int ow = (int)(p % a.Wout);
        p /= a.Wout;
        const int oh = (int)(p % a.Hout);
        const int n = (int)(p / a.Hout);
        const T* w = (const T*)a.w + (int64_t)co * a.taps * Cin;
        const float* sc = a.scale ? a.scale + (int64_t)n * Cin : nullptr;
        const float* sh = a.scale ? a.shift + (int64_t)n * Cin : nullptr;
        const float* dm = a.dmask ? a.dmask + (int64_t)n * Cin : nullptr;
        float acc = 0.f;
        for (int kh = 0; kh < k; ++kh) {
            const int iy = oh * a.stride - pad + kh;
            if (iy < 0 || iy >= Hv) continue;
            if (a.up == 2 && (iy & 1)) continue;  // zero-insertion: only even rows / columns carry data
            const int sy = a.up ? (iy >> 1) : iy;
            for (int kw = 0; kw < k; ++kw) {
                const int ix = ow * a.stride - pad + kw;
                if (ix < 0 || ix >= Wv) continue;
                if (a.up == 2 && (ix & 1)) continue;
                const int sx = a.up ? (ix >> 1) : ix;
                const T* wt = w + (kh * k + kw) * Cin;
                const int64_t pix = ((int64_t)n * a.Hin + sy) * a.Win + sx;
                for (int ci = 0; ci < Cin; ++ci) {
                    float v;
                    if (a.in_nchw)
                        v = ((const float*)a.src1)[(((int64_t)n * a.C1 + ci) * a.Hin + sy) * a.Win + sx];
                    else if (ci < a.C1)
                        v = to_f(((const T*)a.src1)[pix * a.C1 + ci]);
                    else
                        v = to_f(((const T*)a.src2)[pix * a.C2 + (ci - a.C1)]);
                    if (sc) v = fmaf(v, sc[ci], sh[ci]);
                    if (a.pro_silu) v = silu_f(v);
                    if (dm) v *= dm[ci];
                    if (!a.in_nchw) v = to_f(from_f<T>(v));  // the MFMA path feeds operands in T; the fp32 network input stays fp32
                    acc = fmaf(v, to_f(wt[ci]), acc);
                }
            }
        }
        if (a.bias) acc += a.bias[co];
        if (a.tproj) acc += a.tproj[(int64_t)(a.nt == 1 ? 0 : n) * a.tproj_ld + co];
        const int64_t opix = ((int64_t)n * a.Hout + oh) * a.Wout + ow;
        if (a.res1) {
            if (co < a.R1)
                acc += to_f(((const T*)a.res1)[opix * a.R1 + co]);
            else
                acc += to_f(((const T*)a.res2)[opix * (a.Cout - a.R1) + (co - a.R1)]);
        }
        if (a.out_silu) acc = silu_f(acc);
        if (a.out_nchw)
            ((float*)a.dst)[(((int64_t)n * a.Cout + co) * a.Hout + oh) * a.Wout + ow] = acc;
        else
            ((T*)a.dst)[opix * a.Cout + co] = from_f<T>(acc);
    }
}

// ------------------------------------------------------------------ first-layer convolution
// 3x3 pad 1 stride 1 on the NCHW fp32 network input (or a thin NHWC tensor; Cin <= 8), NHWC T output: replaces
// UNet.input_conv (models/ddpm.py:219).  K = 9*Cin is far too small for MFMA; the op is
// bound by the output write.  Each thread computes 8 consecutive couts of one pixel from
// its 9*Cin inputs in registers; weights sit in LDS as [tap*Cin + ci][Cout] fp32.
template <typename T>
__global__ void __launch_bounds__(256) conv_in_kernel(ConvArgs a, int px_per_block) {
    extern __shared__ __attribute__((aligned(16))) float wl[];  // [9*Cin][Cout]
    const int Cin = a.C1, K = 9 * Cin, Cout = a.Cout;
    for (int e = threadIdx.x; e < K * Cout; e += blockDim.x) {
        const int co = e / K, k = e % K;  // packed layout [Cout][tap][Cin] == [Cout][k]
        wl[k * Cout + co] = to_f(((const T*)a.w)[e]);
    }
    __syncthreads();
    const int tpp = Cout / 8;              // threads per pixel
    const int ppb = 256 / tpp;             // pixels per pass
    const int cg = (threadIdx.x % tpp) * 8;
    const int64_t npix = (int64_t)a.N * a.Hout * a.Wout;
    const int64_t p0 = (int64_t)blockIdx.x * px_per_block;
    for (int pp = threadIdx.x / tpp; pp < px_per_block; pp += ppb) {
        const int64_t p = p0 + pp;
        if (p >= npix) break;
        const int ox = (int)(p % a.Wout), oy = (int)((p / a.Wout) % a.Hout), n = (int)(p / ((int64_t)a.Wout * a.Hout));
        float acc[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = a.bias ? a.bias[cg + j] : 0.f;
        for (int kh = 0; kh < 3; ++kh) {
            const int iy = oy - 1 + kh;
            if (iy < 0 || iy >= a.Hin) continue;
            for (int kw = 0; kw < 3; ++kw) {
                const int ix = ox - 1 + kw;
                if (ix < 0 || ix >= a.Win) continue;
                for (int ci = 0; ci < Cin; ++ci) {
                    const float v = a.in_nchw ? ((const float*)a.src1)[(((int64_t)n * Cin + ci) * a.Hin + iy) * a.Win + ix]
                                              : to_f(((const T*)a.src1)[(((int64_t)n * a.Hin + iy) * a.Win + ix) * Cin + ci]);
                    const float* wr = wl + ((kh * 3 + kw) * Cin + ci) * Cout + cg;
                    const float4 w0 = *reinterpret_cast<const float4*>(wr), w1 = *reinterpret_cast<const float4*>(wr + 4);
                    acc[0] = fmaf(v, w0.x, acc[0]); acc[1] = fmaf(v, w0.y, acc[1]);
                    acc[2] = fmaf(v, w0.z, acc[2]); acc[3] = fmaf(v, w0.w, acc[3]);
                    acc[4] = fmaf(v, w1.x, acc[4]); acc[5] = fmaf(v, w1.y, acc[5]);
                    acc[6] = fmaf(v, w1.z, acc[6]); acc[7] = fmaf(v, w1.w, acc[7]);
                }
            }
        }
        T* o = (T*)a.dst + p * Cout + cg;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = from_f<T>(acc[j]);
    }
}

// First conv on the matrix cores: Cin <= 3 NCHW fp32 input, K = 9 * Cin <= 28 as 14 steps of v_mfma_f32_32x32x2f32
// (the input stays fp32: same numerics as the scalar kernel above).  D[cout][pixel] = W[cout][k] . X[k][pixel]: the
// weights are the row operand (held in registers for the wave's lifetime), 32 consecutive pixels the columns, so a
// lane ends up with 4 consecutive couts of ITS pixel per register group: 8-byte (bf16) NHWC stores.
typedef float f32x16_g __attribute__((ext_vector_type(16)));
typedef float f32x4_g __attribute__((ext_vector_type(4)));
template <typename T, int CT>  // CT: 32-cout tiles per wave (Cout = 32 * CT)
__global__ void __launch_bounds__(256) conv_in_mfma_kernel(ConvArgs a, int nblocks) {
    const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
    const int Cin = a.C1, K = 9 * Cin;
    float wreg[CT][14];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int j = 0; j < 14; ++j) {
            const int k = 2 * j + h;
            wreg[ct][j] = k < K ? to_f(((const T*)a.w)[(int64_t)(ct * 32 + r) * K + k]) : 0.f;
        }
    const int HW = a.Hout * a.Wout;
    // per-lane tap table, once per wave: k-step j of this lane is input element (ci, dy, dx) - the divisions by the run-time Cin and
    // the bias loads used to be redone for every 32-pixel block (~1500 non-MFMA instructions per block, most of them this arithmetic)
    int t_off[14], t_dy[14], t_dx[14];
#pragma unroll
    for (int j = 0; j < 14; ++j) {
        const int k = 2 * j + h, tap = k / Cin, ci = k - tap * Cin;
        t_dy[j] = k < K ? tap / 3 - 1 : 1 << 20;  // (past K: never inside the image)
        t_dx[j] = tap % 3 - 1;
        t_off[j] = (ci * a.Hin + (tap / 3 - 1)) * a.Win + tap % 3 - 1;
    }
    __shared__ __attribute__((aligned(16))) float biasL[128];
    if (threadIdx.x < 32 * CT) biasL[threadIdx.x] = a.bias ? a.bias[threadIdx.x] : 0.f;
    __syncthreads();
    for (int blk = blockIdx.x * 4 + (threadIdx.x >> 6); blk < nblocks; blk += gridDim.x * 4) {
        const int p = blk * 32 + r;  // this lane's pixel (column); N*H*W is a multiple of 32
        const int n = p / HW, rem = p - n * HW, oy = rem / a.Wout, ox = rem - oy * a.Wout;
        const float* xb = (const float*)a.src1 + ((int64_t)n * Cin * a.Hin + oy) * a.Win + ox;
        float xv[14];
#pragma unroll
        for (int j = 0; j < 14; ++j) {
            const bool in = (unsigned)(oy + t_dy[j]) < (unsigned)a.Hin && (unsigned)(ox + t_dx[j]) < (unsigned)a.Win;
            xv[j] = in ? xb[t_off[j]] : 0.f;
        }
        constexpr int PITCH = CT * 32 + 4;  // bf16 elements per staged pixel row (+8 bytes: rows on different banks)
        __shared__ __attribute__((aligned(16))) unsigned short stage_in[4][32 * PITCH];
        unsigned short* st = stage_in[threadIdx.x >> 6];
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            f32x16_g acc;
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const f32x4_g b4 = *reinterpret_cast<const f32x4_g*>(biasL + ct * 32 + 8 * g4 + 4 * h);
                acc[4 * g4] = b4[0];
                acc[4 * g4 + 1] = b4[1];
                acc[4 * g4 + 2] = b4[2];
                acc[4 * g4 + 3] = b4[3];
            }
#pragma unroll
            for (int j = 0; j < 14; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wreg[ct][j], xv[j], acc, 0, 0, 0);
            if constexpr (sizeof(T) == 2) {
                // bf16: the wave's 32 pixels x (32 * CT) couts go through LDS so every global store writes whole pixel rows
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) {
                    typedef T bf16x4_g __attribute__((ext_vector_type(4)));  // (bf16 or IEEE half)
                    bf16x4_g v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = (T)acc[4 * gq + e];
                    *reinterpret_cast<bf16x4_g*>(st + r * PITCH + ct * 32 + 8 * gq + 4 * h) = v;
                    if (a.gn_part) {
                        // fused GroupNorm partial (groups of 4 channels = this register group): {mean, M2} of the 32 pixels x
                        // 4 stored (bf16-rounded) values, reduced over the 32 lanes of this half-wave
                        float s = 0.f, ss = 0.f;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float x = (float)v[e];
                            s += x;
                            ss = fmaf(x, x, ss);
                        }
                        s = half_sum(s);  // over this half-wave's 32 pixels
                        ss = half_sum(ss);
                        if (r == 0) {
                            const float mean = s * (1.f / 128.f);
                            float* po = a.gn_part + (((int64_t)n * a.gn_tiles + (rem >> 5)) * (a.Cout / 4) + ct * 8 + 2 * gq + h) * 2;
                            po[0] = mean;
                            po[1] = ss - s * mean;
                        }
                    }
                }
            } else {
                T* o = (T*)a.dst + (int64_t)p * a.Cout + ct * 32 + 4 * h;
#pragma unroll
                for (int gq = 0; gq < 4; ++gq)
                    *reinterpret_cast<float4*>(o + 8 * gq) = make_float4(acc[4 * gq], acc[4 * gq + 1], acc[4 * gq + 2], acc[4 * gq + 3]);
                if (a.gn_part) {  // fp32 tensors (a mixed plan's full-resolution level): the same partials, of the stored fp32 values
#pragma unroll
                    for (int gq = 0; gq < 4; ++gq) {
                        float s = 0.f, ss = 0.f;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float x = acc[4 * gq + e];
                            s += x;
                            ss = fmaf(x, x, ss);
                        }
                        s = half_sum(s);
                        ss = half_sum(ss);
                        if (r == 0) {
                            const float mean = s * (1.f / 128.f);
                            float* po = a.gn_part + (((int64_t)n * a.gn_tiles + (rem >> 5)) * (a.Cout / 4) + ct * 8 + 2 * gq + h) * 2;
                            po[0] = mean;
                            po[1] = ss - s * mean;
                        }
                    }
                }
            }
        }
        if constexpr (sizeof(T) == 2) {
            __builtin_amdgcn_wave_barrier();  // same wave: LDS operations complete in order
            constexpr int VPR = CT * 4;       // 16-byte vectors per pixel row
#pragma unroll
            for (int q = 0; q < 32 * VPR / 64; ++q) {
                const int it = lane + 64 * q, px = it / VPR, sg = it % VPR;
                const uint2 lo = *reinterpret_cast<const uint2*>(st + px * PITCH + sg * 8), hi = *reinterpret_cast<const uint2*>(st + px * PITCH + sg * 8 + 4);
                *reinterpret_cast<uint4*>((T*)a.dst + (int64_t)(blk * 32 + px) * a.Cout + sg * 8) = make_uint4(lo.x, lo.y, hi.x, hi.y);
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
}

static bool conv_in_mfma_supported(const ConvArgs& a) {
    return a.in_nchw && !a.out_nchw && a.taps == 9 && a.stride == 1 && !a.up && a.C2 == 0 && a.C1 <= 3 && !a.scale && !a.pro_silu && !a.dmask &&
           !a.tproj && !a.res1 && !a.out_silu && (a.Cout == 128 || a.Cout == 64 || a.Cout == 32) &&
           (!a.gn_part || (a.gn_cg == 4 && (a.Hout * a.Wout) % 32 == 0)) &&
           ((int64_t)a.N * a.Hout * a.Wout) % 32 == 0 && (int64_t)a.N * a.Hout * a.Wout * a.Cout < (1ll << 31) && !debug_route("no_conv_in_mfma");
}

// the first-conv kernel can emit GroupNorm partials of its output: one per 32-pixel block, groups of exactly 4 channels
bool conv_in_stats_query(int dtype, const ConvArgs& a, int cg, int* tiles, int* px) {
    if (cg != 4 || !conv_in_mfma_supported(a) || (a.Hout * a.Wout) % 32) return false;  // (16-bit and fp32 instances alike)
    *tiles = a.Hout * a.Wout / 32;
    *px = 32;
    return true;
}

static bool conv_in_supported(const ConvArgs& a) {
    return !a.out_nchw && a.taps == 9 && a.stride == 1 && !a.up && a.C2 == 0 && a.C1 <= 8 && !a.scale &&  // 6: data gradient of the IDDPM (eps, v) conv
           !a.pro_silu && !a.dmask && !a.tproj && !a.res1 && !a.out_silu && a.Cout % 8 == 0 && a.Cout <= 2048 &&
           256 % (a.Cout / 8) == 0 && (size_t)9 * a.C1 * a.Cout * 4 <= 48 * 1024;
}

const char* conv_generic_kernel_name(const ConvArgs& a) {
    return conv_in_mfma_supported(a) ? "conv_in_mfma_kernel" : conv_in_supported(a) ? "conv_in_kernel" : "conv_generic_kernel";
}

int launch_conv_generic(int dtype, const ConvArgs& a, hipStream_t s) {
    if (conv_in_mfma_supported(a)) {
        const int nblocks = (int)((int64_t)a.N * a.Hout * a.Wout / 32);
        unsigned grid = (unsigned)((nblocks + 3) / 4);  // persistent waves: the filter registers are loaded once per wave
        if (grid > 512u) grid = 512u;                    // (1024 / 512 / 256 workgroups: 38.6 / 32.2 / 33.4 us)
#define DMME_CIN(TT, CT) hipLaunchKernelGGL((conv_in_mfma_kernel<TT, CT>), dim3(grid), dim3(256), 0, s, a, nblocks)
        if (dtype == DMME_BF16) {
            if (a.Cout == 128) DMME_CIN(bf16, 4); else if (a.Cout == 64) DMME_CIN(bf16, 2); else DMME_CIN(bf16, 1);
        } else if (dtype == DMME_F16) {
            if (a.Cout == 128) DMME_CIN(f16, 4); else if (a.Cout == 64) DMME_CIN(f16, 2); else DMME_CIN(f16, 1);
        } else {
            if (a.Cout == 128) DMME_CIN(float, 4); else if (a.Cout == 64) DMME_CIN(float, 2); else DMME_CIN(float, 1);
        }
#undef DMME_CIN
        DMME_CHECK_LAUNCH();
        return DMME_OK;
    }
    if (conv_in_supported(a) && !a.gn_part) {
        const int64_t npix = (int64_t)a.N * a.Hout * a.Wout;
        const int ppb = 256 / (a.Cout / 8) * 8;  // 8 passes per block
        const unsigned blocks = (unsigned)((npix + ppb - 1) / ppb);
        const size_t lds = (size_t)9 * a.C1 * a.Cout * 4;
        if (dtype == DMME_BF16)
            hipLaunchKernelGGL(conv_in_kernel<bf16>, dim3(blocks), dim3(256), lds, s, a, ppb);
        else if (dtype == DMME_F16)
            hipLaunchKernelGGL(conv_in_kernel<f16>, dim3(blocks), dim3(256), lds, s, a, ppb);
        else
            hipLaunchKernelGGL(conv_in_kernel<float>, dim3(blocks), dim3(256), lds, s, a, ppb);
        DMME_CHECK_LAUNCH();
        return DMME_OK;
    }
    const int64_t total = (int64_t)a.N * a.Hout * a.Wout * a.Cout;
    if (total == 0) return DMME_OK;
    int64_t blocks = (total + 255) / 256;
    if (blocks > 65536) blocks = 65536;
    if (dtype == DMME_BF16)
        hipLaunchKernelGGL(conv_generic_kernel<bf16>, dim3((unsigned)blocks), dim3(256), 0, s, a);
    else if (dtype == DMME_F16)
        hipLaunchKernelGGL(conv_generic_kernel<f16>, dim3((unsigned)blocks), dim3(256), 0, s, a);
    else
        hipLaunchKernelGGL(conv_generic_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, s, a);
    DMME_CHECK_LAUNCH();
    return DMME_OK;
}

// ------------------------------------------------------------------ generic GroupNorm stats
// One workgroup per (n, group); two passes (mean, then centred second moment) in fp32.
// Handles groups that straddle the two concatenated sources.
template <typename T>
__global__ void __launch_bounds__(256) gn_generic_kernel(const T* __restrict__ s1, const T* __restrict__ s2, int HW,
                                                         int C1, int C2, int groups, const float* gamma,
                                                         const float* beta, float eps, float* scale, float* shift,
                                                         float* mean_rstd) {
    __shared__ float red[16];
    const int n = blockIdx.y, g = blockIdx.x;
    const int C = C1 + C2, cg = C / groups;
    const int64_t cnt = (int64_t)cg * HW;
    auto at = [&](int64_t e) -> float {
        const int c = g * cg + (int)(e % cg);
        const int64_t p = (int64_t)n * HW + e / cg;
        return c < C1 ? to_f(s1[p * C1 + c]) : to_f(s2[p * C2 + (c - C1)]);
    };
    float s = 0.f;
    for (int64_t e = threadIdx.x; e < cnt; e += blockDim.x) s += at(e);
    const float mean = block_sum(s, red) / (float)cnt;
    float q = 0.f;
    for (int64_t e = threadIdx.x; e < cnt; e += blockDim.x) {
        const float d = at(e) - mean;
        q = fmaf(d, d, q);
    }
    const float var = block_sum(q, red) / (float)cnt;
    const float rstd = 1.0f / sqrtf(var + eps);
    if (mean_rstd && threadIdx.x == 0) {
        mean_rstd[((int64_t)n * groups + g) * 2] = mean;
        mean_rstd[((int64_t)n * groups + g) * 2 + 1] = rstd;
    }
    for (int j = threadIdx.x; j < cg; j += blockDim.x) {
        const int c = g * cg + j;
        const float a = rstd * gamma[c];
        scale[(int64_t)n * C + c] = a;
        shift[(int64_t)n * C + c] = beta[c] - mean * a;
    }
}

int launch_gn_generic(int dtype, const void* src1, const void* src2, int N, int HW, int C1, int C2, int groups,
                      const float* gamma, const float* beta, float eps, float* scale, float* shift, float* mean_rstd,
                      hipStream_t s) {
    dim3 grid(groups, N);
    if (dtype == DMME_BF16)
        hipLaunchKernelGGL(gn_generic_kernel<bf16>, grid, dim3(256), 0, s, (const bf16*)src1, (const bf16*)src2, HW,
                           C1, C2, groups, gamma, beta, eps, scale, shift, mean_rstd);
    else if (dtype == DMME_F16)
        hipLaunchKernelGGL(gn_generic_kernel<f16>, grid, dim3(256), 0, s, (const f16*)src1, (const f16*)src2, HW,
                           C1, C2, groups, gamma, beta, eps, scale, shift, mean_rstd);
    else
        hipLaunchKernelGGL(gn_generic_kernel<float>, grid, dim3(256), 0, s, (const float*)src1, (const float*)src2,
                           HW, C1, C2, groups, gamma, beta, eps, scale, shift, mean_rstd);
    DMME_CHECK_LAUNCH();
    return DMME_OK;
}

// ------------------------------------------------------------------ generic attention
// One workgroup per (query token, image x head).  scores in LDS (S floats) + q (d floats).
// Head view (heads > 1, models/iddpm.py:35-47): head h of image b reads the qkv channels [h*3d, (h+1)*3d) as (q | k | v),
// d = C / heads; K is scaled by C^-0.5 (the full width); result row i = b*heads + h is stored as image i % N, head i / N
// (the reference's "(b head)" split read back as "(head b)").  heads == 1 is the DDPM single-head block (models/ddpm.py:54-63).
template <typename T>
__global__ void __launch_bounds__(256) attn_generic_kernel(const T* __restrict__ qkv, int S, int C, int heads, int N, T* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int d = C / heads;
    float* qs = sm;       // d
    float* ps = sm + d;   // S
    float* red = ps + S;  // 16
    const int bh = blockIdx.y, n = bh / heads, hd = bh % heads, i = blockIdx.x;
    const T* base = qkv + (int64_t)n * S * 3 * C + (int64_t)hd * 3 * d;
    const float kscale = powf((float)C, -0.5f);
    for (int c = threadIdx.x; c < d; c += blockDim.x) qs[c] = to_f(base[(int64_t)i * 3 * C + c]);
    __syncthreads();
    float lmax = -INFINITY;
    for (int j = threadIdx.x; j < S; j += blockDim.x) {
        const T* kr = base + (int64_t)j * 3 * C + d;
        float acc = 0.f;
        for (int c = 0; c < d; ++c) acc = fmaf(qs[c], to_f(from_f<T>(to_f(kr[c]) * kscale)), acc);
        ps[j] = acc;
        lmax = fmaxf(lmax, acc);
    }
    const float m = block_max(lmax, red);
    float lsum = 0.f;
    for (int j = threadIdx.x; j < S; j += blockDim.x) {
        const float e = expf(ps[j] - m);
        ps[j] = e;
        lsum += e;
    }
    const float tot = block_sum(lsum, red);
    __syncthreads();
    const float inv = 1.0f / tot;
    const int on = bh % N, oh = bh / N;
    for (int c = threadIdx.x; c < d; c += blockDim.x) {
        float acc = 0.f;
        for (int j = 0; j < S; ++j) acc = fmaf(to_f(from_f<T>(ps[j] * inv)), to_f(base[(int64_t)j * 3 * C + 2 * d + c]), acc);
        out[((int64_t)on * S + i) * C + oh * d + c] = from_f<T>(acc);
    }
}

// S = 16 (the 4x4 maps): ONE workgroup per (image, head) instead of one per query - q, k, v of the row block (16 x d each) in LDS as
// fp32, thread (i, j) computes score (i, j), the 16 lanes of a row reduce max / sum with DPP row operations, thread (i, c-block) the
// output.  Same roundings as attn_generic_kernel (k * C^-0.5 and the normalised probabilities rounded to T), which ran 16 active
// threads per workgroup through 256-long dependent dot products: 20 us for 0.03 GFLOP at B = 128.
template <typename T>
__global__ void __launch_bounds__(256) attn_s16_kernel(const T* __restrict__ qkv, int C, int heads, int N, T* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    constexpr int S = 16;
    const int d = C / heads, dp = d + 4;  // (+4 floats: rows of one quantity on different banks)
    float* qs = sm;                        // [16][dp]
    float* ks = qs + S * dp;
    float* vs = ks + S * dp;
    float* ps = vs + S * dp;               // [16][17]
    const int bh = blockIdx.x, n = bh / heads, hd = bh % heads, tid = threadIdx.x;
    const T* base = qkv + (int64_t)n * S * 3 * C + (int64_t)hd * 3 * d;
    const float kscale = powf((float)C, -0.5f);
    for (int e = tid; e < S * d; e += 256) {
        const int row = e / d, c = e - row * d;
        const T* p = base + (int64_t)row * 3 * C + c;
        qs[row * dp + c] = to_f(p[0]);
        ks[row * dp + c] = to_f(from_f<T>(to_f(p[d]) * kscale));
        vs[row * dp + c] = to_f(p[2 * d]);
    }
    __syncthreads();
    const int i = tid >> 4, j = tid & 15;
    float acc = 0.f;
    for (int c = 0; c < d; c += 4) {
        const f32x4_g a = *reinterpret_cast<const f32x4_g*>(qs + i * dp + c), b = *reinterpret_cast<const f32x4_g*>(ks + j * dp + c);
        acc = fmaf(a[0], b[0], acc);
        acc = fmaf(a[1], b[1], acc);
        acc = fmaf(a[2], b[2], acc);
        acc = fmaf(a[3], b[3], acc);
    }
    float m = acc;  // the 16 lanes of a DPP row are this query's 16 keys
    m = fmaxf(m, DMME_DPP_F(m, 0xB1));
    m = fmaxf(m, DMME_DPP_F(m, 0x4E));
    m = fmaxf(m, DMME_DPP_F(m, 0x141));
    m = fmaxf(m, DMME_DPP_F(m, 0x140));
    const float e = expf(acc - m);
    const float tot = row16_sum(e);
    ps[i * 17 + j] = to_f(from_f<T>(e * (1.0f / tot)));
    __syncthreads();
    const int on = bh % N, oh = bh / N;
    for (int c = j; c < d; c += 16) {  // lanes of a row: consecutive channels
        float o = 0.f;
#pragma unroll
        for (int jj = 0; jj < S; ++jj) o = fmaf(ps[i * 17 + jj], vs[jj * dp + c], o);
        out[((int64_t)on * S + i) * C + oh * d + c] = from_f<T>(o);
    }
}

int launch_attn_heads(int dtype, const void* qkv, int N, int S, int C, int heads, void* out, hipStream_t s) {
    DMME_REQUIRE(heads >= 1 && C % heads == 0, DMME_ERR_INVALID, "attention: width %d not divisible by %d heads", C, heads);
    if (S == 16 && (C / heads) % 4 == 0 && C / heads <= 1024 && !debug_route("no_attn_s16")) {
        const size_t lds16 = (size_t)(3 * 16 * (C / heads + 4) + 16 * 17) * sizeof(float);
        if (lds16 <= 64 * 1024) {
            if (dtype == DMME_BF16)
                hipLaunchKernelGGL(attn_s16_kernel<bf16>, dim3((unsigned)(N * heads)), dim3(256), lds16, s, (const bf16*)qkv, C, heads, N, (bf16*)out);
            else if (dtype == DMME_F16)
                hipLaunchKernelGGL(attn_s16_kernel<f16>, dim3((unsigned)(N * heads)), dim3(256), lds16, s, (const f16*)qkv, C, heads, N, (f16*)out);
            else
                hipLaunchKernelGGL(attn_s16_kernel<float>, dim3((unsigned)(N * heads)), dim3(256), lds16, s, (const float*)qkv, C, heads, N, (float*)out);
            DMME_CHECK_LAUNCH();
            return DMME_OK;
        }
    }
    const size_t lds = (size_t)(C / heads + S + 16) * sizeof(float);
    DMME_REQUIRE(lds <= 64 * 1024, DMME_ERR_UNSUPPORTED, "attention generic: d+S too large (%d+%d)", C / heads, S);
    dim3 grid(S, N * heads);
    if (dtype == DMME_BF16)
        hipLaunchKernelGGL(attn_generic_kernel<bf16>, grid, dim3(256), lds, s, (const bf16*)qkv, S, C, heads, N, (bf16*)out);
    else if (dtype == DMME_F16)
        hipLaunchKernelGGL(attn_generic_kernel<f16>, grid, dim3(256), lds, s, (const f16*)qkv, S, C, heads, N, (f16*)out);
    else
        hipLaunchKernelGGL(attn_generic_kernel<float>, grid, dim3(256), lds, s, (const float*)qkv, S, C, heads, N, (float*)out);
    DMME_CHECK_LAUNCH();
    return DMME_OK;
}
int launch_attn_generic(int dtype, const void* qkv, int N, int S, int C, void* out, hipStream_t s) {
    return launch_attn_heads(dtype, qkv, N, S, C, 1, out, s);
}

// scale-shift conditioning of iddpm.ResBlock (models/iddpm.py:117-118) folded into the GroupNorm's per-(n, c) affine:
// GN(h) * (1 + t_scale) + t_shift = h * (sc * (1 + t_scale)) + (sh * (1 + t_scale) + t_shift).  t rows: nt == 1 broadcasts.
__global__ void gn_modulate_kernel(float* __restrict__ scale, float* __restrict__ shift, const float* __restrict__ t_shift,
                                   const float* __restrict__ t_scale, int ld, int nt, int N, int C) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N * C) return;
    const int n = i / C, c = i % C, r = nt == 1 ? 0 : n;
    const float m = 1.0f + t_scale[(int64_t)r * ld + c];
    scale[i] = scale[i] * m;
    shift[i] = fmaf(shift[i], m, t_shift[(int64_t)r * ld + c]);
}
int launch_gn_modulate(float* scale, float* shift, const float* t_shift, const float* t_scale, int ld, int nt, int N, int C, hipStream_t s) {
    hipLaunchKernelGGL(gn_modulate_kernel, dim3((N * C + 255) / 256), dim3(256), 0, s, scale, shift, t_shift, t_scale, ld, nt, N, C);
    DMME_CHECK_LAUNCH();
    return DMME_OK;
}

// ------------------------------------------------------------------ time embedding pieces
// SinusoidalPositionEmbeddings.forward (models/ddpm.py:347-348): [sin(t f), cos(t f)]
__global__ void time_sinusoid_kernel(const int64_t* t, int nt, const float* freqs, int half, float* out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nt * half) return;
    const int r = i / half, k = i % half;
    const float arg = (float)t[r] * freqs[k];
    out[(int64_t)r * 2 * half + k] = sinf(arg);
    out[(int64_t)r * 2 * half + half + k] = cosf(arg);
}

int launch_time_sinusoid(const int64_t* t, int nt, const float* freqs, int half, float* out, hipStream_t s) {
    const int total = nt * half;
    hipLaunchKernelGGL(time_sinusoid_kernel, dim3((total + 255) / 256), dim3(256), 0, s, t, nt, freqs, half, out);
    DMME_CHECK_LAUNCH();
    return DMME_OK;
}

// nn.Linear on tiny batches: one wavefront per output feature, coalesced weight row,
// shuffle reduction; loops over the nt input rows (1 when sampling, B when training).
template <typename T>
__global__ void __launch_bounds__(256) linear_wave_kernel(const float* __restrict__ in, int nt, int K,
                                                          const T* __restrict__ W, const float* __restrict__ bias,
                                                          int Nout, int out_silu, float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int o = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (o >= Nout) return;
    const T* wr = W + (int64_t)o * K;
    for (int r = 0; r < nt; ++r) {
        const float* xr = in + (int64_t)r * K;
        float acc = 0.f;
        for (int k = lane; k < K; k += 64) acc = fmaf(xr[k], to_f(wr[k]), acc);
        acc = wave_sum(acc);
        if (lane == 0) {
            float v = acc + bias[o];
            if (out_silu) v = silu_f(v);
            out[(int64_t)r * Nout + o] = v;
        }
    }
}

int launch_linear_wave(int dtype, const float* in, int nt, int K, const void* W, const float* bias, int Nout,
                       int out_silu, float* out, hipStream_t s) {
    dim3 grid((Nout + 3) / 4);
    if (dtype == DMME_BF16)
        hipLaunchKernelGGL(linear_wave_kernel<bf16>, grid, dim3(256), 0, s, in, nt, K, (const bf16*)W, bias, Nout,
                           out_silu, out);
    else if (dtype == DMME_F16)
        hipLaunchKernelGGL(linear_wave_kernel<f16>, grid, dim3(256), 0, s, in, nt, K, (const f16*)W, bias, Nout,
                           out_silu, out);
    else
        hipLaunchKernelGGL(linear_wave_kernel<float>, grid, dim3(256), 0, s, in, nt, K, (const float*)W, bias, Nout,
                           out_silu, out);
    DMME_CHECK_LAUNCH();
    return DMME_OK;
}

// ------------------------------------------------------------------ layout converters
template <typename T>
__global__ void nchw_to_nhwc_kernel(const float* src, int C, int HW, T* dst, int64_t total) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const int64_t p = i / C;  // n*HW + hw
        const int64_t n = p / HW, hw = p % HW;
        dst[i] = from_f<T>(src[(n * C + c) * HW + hw]);
    }
}
template <typename T>
__global__ void nhwc_to_nchw_kernel(const T* src, int C, int HW, float* dst, int64_t total) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t hw = i % HW;
        const int64_t q = i / HW;  // n*C + c
        const int64_t n = q / C, c = q % C;
        dst[i] = to_f(src[(n * HW + hw) * C + c]);
    }
}
static inline unsigned grid_for(int64_t total) {
    int64_t b = (total + 255) / 256;
    return (unsigned)(b > 16384 ? 16384 : (b < 1 ? 1 : b));
}
int launch_nchw_to_nhwc(int dtype, const float* src, int N, int C, int HW, void* dst, hipStream_t s) {
    const int64_t total = (int64_t)N * C * HW;
    if (dtype == DMME_BF16)
        hipLaunchKernelGGL(nchw_to_nhwc_kernel<bf16>, dim3(grid_for(total)), dim3(256), 0, s, src, C, HW, (bf16*)dst, total);
    else if (dtype == DMME_F16)
        hipLaunchKernelGGL(nchw_to_nhwc_kernel<f16>, dim3(grid_for(total)), dim3(256), 0, s, src, C, HW, (f16*)dst, total);
    else
        hipLaunchKernelGGL(nchw_to_nhwc_kernel<float>, dim3(grid_for(total)), dim3(256), 0, s, src, C, HW, (float*)dst, total);
    DMME_CHECK_LAUNCH();
    return DMME_OK;
}
int launch_nhwc_to_nchw(int dtype, const void* src, int N, int C, int HW, float* dst, hipStream_t s) {
    const int64_t total = (int64_t)N * C * HW;
    if (dtype == DMME_BF16)
        hipLaunchKernelGGL(nhwc_to_nchw_kernel<bf16>, dim3(grid_for(total)), dim3(256), 0, s, (const bf16*)src, C, HW, dst, total);
    else if (dtype == DMME_F16)
        hipLaunchKernelGGL(nhwc_to_nchw_kernel<f16>, dim3(grid_for(total)), dim3(256), 0, s, (const f16*)src, C, HW, dst, total);
    else
        hipLaunchKernelGGL(nhwc_to_nchw_kernel<float>, dim3(grid_for(total)), dim3(256), 0, s, (const float*)src, C, HW, dst, total);
    DMME_CHECK_LAUNCH();
    return DMME_OK;
}

// ------------------------------------------------------------------ weight re-pack
// (Cout, Cin, k, k) fp32 -> [Cout][k*k][Cin] T.  Linear / 1x1 weights have taps = 1.
template <typename T>
__global__ void pack_weight_kernel(const float* src, int Cin, int taps, T* dst, int64_t total) {
    const int64_t row = (int64_t)Cin * taps;
    for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = e / row, rem = e % row;
        const int tap = (int)(rem / Cin), ci = (int)(rem % Cin);
        dst[e] = from_f<T>(src[(r * Cin + ci) * taps + tap]);
    }
}
// ... split halves for the three-pass fp16 kernels: [Cout][taps][Cin / 32][hi 32 | lo 32] (as pack_table_kernel's code 4)
__global__ void pack_weight_split_kernel(const float* src, int Cin, int taps, f16* dst, int64_t total) {
    const int64_t row = (int64_t)Cin * taps;
    for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = e / row, rem = e % row;
        const int tap = (int)(rem / Cin), ci = (int)(rem % Cin);
        const float w = src[(r * Cin + ci) * taps + tap];
        const f16 hi = (f16)w;
        f16* o = dst + (r * taps + tap) * (2 * (int64_t)Cin) + (ci >> 5) * 64 + (ci & 31);
        o[0] = hi;
        o[32] = (f16)(w - (float)hi);
    }
}
int launch_pack_weight(int dtype, const float* src, int Cout, int Cin, int taps, void* dst, hipStream_t s) {
    const int64_t total = (int64_t)Cout * Cin * taps;
    if (dtype == DMME_F16R32) {
        DMME_REQUIRE(Cin % 32 == 0, DMME_ERR_INVALID, "pack_weight(fp16r32): whole 32-channel blocks");
        hipLaunchKernelGGL(pack_weight_split_kernel, dim3(grid_for(total)), dim3(256), 0, s, src, Cin, taps, (f16*)dst, total);
        DMME_CHECK_LAUNCH();
        return DMME_OK;
    }
    if (dtype == DMME_BF16)
        hipLaunchKernelGGL(pack_weight_kernel<bf16>, dim3(grid_for(total)), dim3(256), 0, s, src, Cin, taps, (bf16*)dst, total);
    else if (dtype == DMME_F16)
        hipLaunchKernelGGL(pack_weight_kernel<f16>, dim3(grid_for(total)), dim3(256), 0, s, src, Cin, taps, (f16*)dst, total);
    else
        hipLaunchKernelGGL(pack_weight_kernel<float>, dim3(grid_for(total)), dim3(256), 0, s, src, Cin, taps, (float*)dst, total);
    DMME_CHECK_LAUNCH();
    return DMME_OK;
}

// table-driven: one workgroup per PackItem = a (cout range) x (cin range) sub-block of one tensor, at most 8192
// elements.  The sub-block is read with contiguous runs of nci*taps floats per cout into LDS and written back in the
// packed order ([cout][tap][cin], or [cin][taps-1-tap][cout] for the data-gradient copy) with contiguous runs too.
template <typename T>
__global__ void __launch_bounds__(256) pack_table_kernel(const PackItem* items, const float* ref, char* packed) {
    __shared__ __attribute__((aligned(16))) float tile[8192 + 256];
    const PackItem it = items[blockIdx.x];
    const float* src = ref + it.src_off;
    if (it.as_f32 == 1) {
        float* dst = (float*)(packed + it.dst_off);
        const int row = it.cin * it.taps, total = it.rows * row, e0 = it.row0 * row;
        for (int e = threadIdx.x; e < total; e += blockDim.x) dst[e0 + e] = src[e0 + e];
        return;
    }
    const int taps = it.taps, nci = it.nci, seg = nci * taps;  // contiguous source run per cout
    const int pitch = seg | 1;                                  // odd pitch: conflict-free column reads
    const int total = it.rows * seg;
    // source runs as 16-byte loads where the geometry allows (every run starts on a 16-byte boundary and is whole vectors long):
    // a quarter of the load instructions and of the index arithmetic of the scalar loop
    const float* run0 = src + ((int64_t)it.row0 * it.cin + it.ci0) * taps;
    const bool vec_in = (seg & 3) == 0 && ((it.cin * taps) & 3) == 0 && (((size_t)run0) & 15) == 0;
    // all of a thread's loads are requested before the first is stored (a load-then-store loop is one HBM round trip per iteration:
    // 8-32 dependent ones per workgroup, which is what this kernel's 125 us were)
    if (vec_in) {
        const int seg4 = seg >> 2, total4 = it.rows * seg4;  // <= 2048: at most 8 vectors per thread
        float4 v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int e = threadIdx.x + 256 * k;
            if (e < total4) {
                const int co = e / seg4, q = e - co * seg4;
                v[k] = *reinterpret_cast<const float4*>(run0 + (int64_t)co * it.cin * taps + 4 * q);
            }
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int e = threadIdx.x + 256 * k;
            if (e < total4) {
                const int co = e / seg4, q = e - co * seg4;
                float* t = tile + co * pitch + 4 * q;
                t[0] = v[k].x;
                t[1] = v[k].y;
                t[2] = v[k].z;
                t[3] = v[k].w;
            }
        }
    } else {
        for (int base = 0; base < total; base += 256 * 8) {
            float v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int e = base + threadIdx.x + 256 * k;
                if (e < total) {
                    const int co = e / seg, rem = e - co * seg;
                    v[k] = run0[(int64_t)co * it.cin * taps + rem];
                }
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int e = base + threadIdx.x + 256 * k;
                if (e < total) {
                    const int co = e / seg, rem = e - co * seg;
                    tile[co * pitch + rem] = v[k];
                }
            }
        }
    }
    __syncthreads();
    if (it.as_f32 == 3) {  // [co][tap][ci] in fp32 (a conv of a 16-bit plan that is routed to the fp32 / three-pass kernels)
        float* d32 = (float*)(packed + it.dst_off);
        for (int e = threadIdx.x; e < total; e += blockDim.x) {
            const int cil = e % nci, q = e / nci, tap = q % taps, co = q / taps;
            d32[((int64_t)(it.row0 + co) * taps + tap) * it.cin + it.ci0 + cil] = tile[co * pitch + cil * taps + tap];
        }
        return;
    }
    if (it.as_f32 == 4) {  // split halves: k-slot of channel ci = (ci / 32) * 64 + ci % 32 (hi), + 32 (lo); row length 2 * cin
        f16* dh = (f16*)(packed + it.dst_off);
        for (int e = threadIdx.x; e < total; e += blockDim.x) {
            const int cil = e % nci, q = e / nci, tap = q % taps, co = q / taps;
            const int ci = it.ci0 + cil;
            const float w = tile[co * pitch + cil * taps + tap];
            const f16 hi = (f16)w;
            f16* row = dh + ((int64_t)(it.row0 + co) * taps + tap) * (2 * it.cin) + (ci >> 5) * 64 + (ci & 31);
            row[0] = hi;
            row[32] = (f16)(w - (float)hi);
        }
        return;
    }
    T* dst = (T*)(packed + it.dst_off);
    constexpr int V = 16 / (int)sizeof(T);  // elements per 16-byte store
    if (it.as_f32 == 2) {  // dst[ci][taps-1-tap][co]: runs of `rows` couts
        if (it.rows % V == 0 && it.cout % V == 0 && it.row0 % V == 0) {
            const int rv = it.rows / V, totalv = rv * seg;
            for (int e = threadIdx.x; e < totalv; e += blockDim.x) {
                const int cv = e % rv, q = e / rv, tap = q % taps, cil = q / taps;
                T o[V];
#pragma unroll
                for (int j = 0; j < V; ++j) o[j] = from_f<T>(tile[(cv * V + j) * pitch + cil * taps + tap]);
                *reinterpret_cast<uint4*>(dst + ((int64_t)(it.ci0 + cil) * taps + (taps - 1 - tap)) * it.cout + it.row0 + cv * V) = *reinterpret_cast<const uint4*>(o);
            }
        } else {
            for (int e = threadIdx.x; e < total; e += blockDim.x) {
                const int co = e % it.rows, q = e / it.rows, tap = q % taps, cil = q / taps;
                dst[((int64_t)(it.ci0 + cil) * taps + (taps - 1 - tap)) * it.cout + it.row0 + co] = from_f<T>(tile[co * pitch + cil * taps + tap]);
            }
        }
    } else {               // dst[co][tap][ci]: runs of nci cins
        if (nci % V == 0 && it.cin % V == 0 && it.ci0 % V == 0) {
            const int nv = nci / V, totalv = it.rows * taps * nv;
            for (int e = threadIdx.x; e < totalv; e += blockDim.x) {
                const int cv = e % nv, q = e / nv, tap = q % taps, co = q / taps;
                T o[V];
#pragma unroll
                for (int j = 0; j < V; ++j) o[j] = from_f<T>(tile[co * pitch + (cv * V + j) * taps + tap]);
                *reinterpret_cast<uint4*>(dst + ((int64_t)(it.row0 + co) * taps + tap) * it.cin + it.ci0 + cv * V) = *reinterpret_cast<const uint4*>(o);
            }
        } else {
            for (int e = threadIdx.x; e < total; e += blockDim.x) {
                const int cil = e % nci, q = e / nci, tap = q % taps, co = q / taps;
                dst[((int64_t)(it.row0 + co) * taps + tap) * it.cin + it.ci0 + cil] = from_f<T>(tile[co * pitch + cil * taps + tap]);
            }
        }
    }
}
template <typename T>
__global__ void __launch_bounds__(256) cast_f32_to_16_kernel(const float* __restrict__ src, int64_t nvec, T* __restrict__ dst) {
    // eight values per thread: two 16-byte loads, one 16-byte store; grid-stride
    for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v < nvec; v += (int64_t)gridDim.x * blockDim.x) {
        typedef float f4 __attribute__((ext_vector_type(4)));
        typedef T t8 __attribute__((ext_vector_type(8)));
        const f4 a = *reinterpret_cast<const f4*>(src + 8 * v), b = *reinterpret_cast<const f4*>(src + 8 * v + 4);
        t8 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            o[e] = (T)a[e];
            o[4 + e] = (T)b[e];
        }
        *reinterpret_cast<t8*>(dst + 8 * v) = o;
    }
}
int launch_cast_f32_to_16(int dtype, const float* src, int64_t numel, void* dst, hipStream_t s) {
    DMME_REQUIRE(is16(dtype) && numel % 8 == 0, DMME_ERR_INVALID, "cast: 16-bit destination types, whole 16-byte vectors");
    const int64_t nvec = numel / 8;
    const unsigned blocks = (unsigned)std::min<int64_t>((nvec + 255) / 256, 256 * 16);
    if (dtype == DMME_F16)
        hipLaunchKernelGGL(cast_f32_to_16_kernel<f16>, dim3(blocks), dim3(256), 0, s, src, nvec, (f16*)dst);
    else
        hipLaunchKernelGGL(cast_f32_to_16_kernel<bf16>, dim3(blocks), dim3(256), 0, s, src, nvec, (bf16*)dst);
    DMME_CHECK_LAUNCH();
    return DMME_OK;
}

int launch_pack_table(int dtype, const PackItem* items_dev, int n_items, const float* ref_flat, void* packed,
                      hipStream_t s) {
    if (n_items == 0) return DMME_OK;
    if (dtype == DMME_BF16)
        hipLaunchKernelGGL(pack_table_kernel<bf16>, dim3(n_items), dim3(256), 0, s, items_dev, ref_flat, (char*)packed);
    else if (dtype == DMME_F16)
        hipLaunchKernelGGL(pack_table_kernel<f16>, dim3(n_items), dim3(256), 0, s, items_dev, ref_flat, (char*)packed);
    else
        hipLaunchKernelGGL(pack_table_kernel<float>, dim3(n_items), dim3(256), 0, s, items_dev, ref_flat, (char*)packed);
    DMME_CHECK_LAUNCH();
    return DMME_OK;
}

// ------------------------------------------------------------------ diagnostic: what one CU can pull from L2
// Every workgroup (one per CU) streams the same `bytes` of an L2-resident buffer `iters` times, DEPTH 16-byte loads in flight per
// thread (mode 0: into registers; mode 1: global -> LDS DMA, no registers; mode m >= 2: DMA gathering eight 128-byte rows per wave instruction, m 16-byte vectors apart).  tools/l2_stream.py turns the elapsed time into bytes
// per clock per CU - the number every tile-size decision in DESIGN.md section 4 leans on.
template <int DEPTH>
__global__ void __launch_bounds__(256) l2_stream_kernel(const uint4* __restrict__ buf, int nvec, int iters, unsigned* __restrict__ sink) {
    unsigned acc = 0;
    const int tid = threadIdx.x;
    for (int it = 0; it < iters; ++it) {
        for (int base = 0; base + DEPTH * 256 <= nvec; base += DEPTH * 256) {
            uint4 v[DEPTH];
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) v[d] = buf[base + d * 256 + tid];
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) acc ^= v[d].x ^ v[d].y ^ v[d].z ^ v[d].w;
        }
    }
    if (acc == 0x12345u) sink[blockIdx.x] = acc;  // keeps the loads alive
}
typedef __attribute__((address_space(3))) char l2s_lds_c;
template <int DEPTH>
__global__ void __launch_bounds__(256) l2_stream_dma_kernel(const uint4* __restrict__ buf, int nvec, int iters, unsigned* __restrict__ sink, int row_stride) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, wave = tid >> 6;
    (void)wave;
    unsigned acc = 0;
    for (int it = 0; it < iters; ++it) {
        int gbase = 0;  // gather mode: first row of this batch, in vectors
        for (int base = 0; base + DEPTH * 256 <= nvec; base += DEPTH * 256) {
            if (row_stride) {
                gbase += DEPTH * 32 * row_stride;
                if (gbase + (DEPTH * 32 + 32) * row_stride >= nvec) gbase = 0;
            }
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) {
#if defined(__HIP_DEVICE_COMPILE__)
                const unsigned l = (unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned)(size_t)(l2s_lds_c*)lds + (unsigned)((d * 4 + wave) * 1024)));
                // row_stride > 0: the convolutions' access shape - a wave instruction gathers eight 128-byte rows `row_stride` vectors apart
                // (cheap addressing: a per-lane offset plus a uniform base that wraps by compare - no division in the loop)
                const int idx = row_stride ? gbase + (d * 32 + (tid >> 3)) * row_stride + (tid & 7) : base + d * 256 + tid;
                __builtin_amdgcn_global_load_lds(buf + idx, (l2s_lds_c*)(size_t)l, 16, 0, 0);
#endif
            }
#if defined(__HIP_DEVICE_COMPILE__)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
        }
        acc ^= reinterpret_cast<unsigned*>(lds)[tid];
    }
    if (acc == 0x12345u) sink[blockIdx.x] = acc;
}
// ------------------------------------------------------------------ diagnostic: do VALU work and MFMAs of two waves on one SIMD overlap?
// 512 threads per workgroup, one workgroup per CU: waves 0-3 issue `iters` rounds of eight independent 32x32x16 bf16 MFMAs (the
// wave-specialised conv kernel's consumer), waves 4-7 `iters` rounds of the GroupNorm/SiLU prologue arithmetic on eight values (its
// producer).  mode bit 0: run the MFMA waves, bit 1: run the VALU waves, bit 2: accumulators in AccVGPRs (inline asm) instead of
// wherever the compiler puts them (arch VGPRs at this occupancy).
typedef float f32x16_d __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8_d __attribute__((ext_vector_type(8)));
__global__ void __launch_bounds__(512, 1) mfma_valu_kernel(int mode, int iters, float* __restrict__ sink) {
    const int wave = threadIdx.x >> 6;
    float out = 0.f;
    if (wave < 4) {
        if (!(mode & 1)) return;
        bf16x8_d a, b;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            a[e] = (__bf16)(float)(threadIdx.x + e);
            b[e] = (__bf16)(float)(threadIdx.x ^ e);
        }
        f32x16_d acc[8];
#pragma unroll
        for (int k = 0; k < 8; ++k)
#pragma unroll
            for (int j = 0; j < 16; ++j) acc[k][j] = 0.f;
        if (mode & 4) {
            for (int it = 0; it < iters; ++it) {
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
                for (int k = 0; k < 8; ++k) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[k]) : "v"(a), "v"(b));
#endif
            }
        } else {
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int k = 0; k < 8; ++k) acc[k] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[k], 0, 0, 0);
                asm volatile("" ::: "memory");
            }
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) out += acc[k][0] + acc[k][15];
    } else {
        if (!(mode & 2)) return;
        float v[8], sc = 1.0001f + threadIdx.x * 1e-6f, sh = 0.01f;
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = 0.1f * e + threadIdx.x * 1e-3f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int rep = 0; rep < 4; ++rep)  // ~ the prologue of four 16-byte halo units per MFMA round
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float u = fmaf(v[e], sc, sh);
                    v[e] = u * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * u)) * 0.999f;
                }
            asm volatile("" ::: "memory");
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) out += v[e];
    }
    if (out == 12345.678f) sink[blockIdx.x] = out;
}
int launch_mfma_valu(int mode, int iters, int blocks, float* sink, hipStream_t s) {
    hipLaunchKernelGGL(mfma_valu_kernel, dim3((unsigned)blocks), dim3(512), 0, s, mode, iters, sink);
    DMME_CHECK_LAUNCH();
    return DMME_OK;
}

// ------------------------------------------------------------------ diagnostic: what ONE kind of vector instruction costs beside MFMAs
// The probe above runs compiler-scheduled GroupNorm / SiLU arithmetic (which hipcc SLP-packs into v_pk_* instructions) and cannot
// tell instruction kinds apart.  Here every instruction is inline asm: waves 0-3 issue rounds of eight independent 32x32x16 (or
// sixteen 16x16x32) bf16 MFMAs, waves 4-7 `n_inner` x 8 instructions of ONE kind per round (eight independent registers), i.e.
// `n_inner` instructions per 32x32x16 MFMA slot of the partner wave on the same SIMD.  Both wave kinds record their own cycle count
// (s_memtime) so that the clock the chip holds does not enter: sink[2 b] = MFMA wave 0's cycles, sink[2 b + 1] = VALU wave 4's.
// kind: 0 v_fma_f32, 1 v_pk_fma_f32, 2 v_exp_f32, 3 v_cvt_pk_bf16_f32, 4 v_pk_mul_f32, 5 v_pk_add_f32, 6 v_add_f32, 7 v_rcp_f32,
// 8 the prologue of one halo dword in plain instructions (4 fma, 2 exp, 2 add, 2 rcp, 2 mul, 1 cvt_pk = 13), 9 the same packed
// (2 pk_fma, 2 exp, 1 pk_add, 2 rcp, 1 pk_mul, 1 cvt_pk = 9), 10 ds_write_b128, 11 ds_read_b128, 12 v_mul_f32 + v_and/v_lshl (unpack).
// flags: bit 0 MFMA waves run, bit 1 VALU waves run, bit 2 s_setprio 3 on the MFMA waves, bit 3 s_setprio 3 on the VALU waves,
// bit 4 16x16x32 MFMAs.
typedef float f32x2_d __attribute__((ext_vector_type(2)));
typedef float f32x4_d __attribute__((ext_vector_type(4)));
template <int KIND>
__global__ void __launch_bounds__(512, 1) issue_probe_kernel(int n_inner, int iters, int flags, long long* __restrict__ sink, const char* __restrict__ src) {
    __shared__ __attribute__((aligned(16))) char plds[KIND >= 13 ? 64 * 1024 : 512 * 16 * 2];
    const int wave = threadIdx.x >> 6;
#if defined(__HIP_DEVICE_COMPILE__)
    if (wave < 4) {
        if (!(flags & 1)) return;
        if (flags & 4) __builtin_amdgcn_s_setprio(3);
        bf16x8_d a, b;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            a[e] = (__bf16)(float)((threadIdx.x * 7 + e) % 13 - 6);
            b[e] = (__bf16)(float)(((threadIdx.x ^ e) * 5) % 11 - 5);
        }
        float out = 0.f;
        long long t0, t1;
        if (flags & 16) {
            f32x4_d acc[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) acc[k] = f32x4_d{0.f, 0.f, 0.f, 0.f};
            t0 = (long long)__builtin_amdgcn_s_memtime();
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int k = 0; k < 16; ++k) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[k]) : "v"(a), "v"(b));
            }
            t1 = (long long)__builtin_amdgcn_s_memtime();
#pragma unroll
            for (int k = 0; k < 16; ++k) out += acc[k][0] + acc[k][3];
        } else {
            f32x16_d acc[8];
#pragma unroll
            for (int k = 0; k < 8; ++k)
#pragma unroll
                for (int j = 0; j < 16; ++j) acc[k][j] = 0.f;
            t0 = (long long)__builtin_amdgcn_s_memtime();
            if (flags & 32) {  // with the conv consumer's LDS traffic: six 16-byte fragment reads per eight MFMAs
                const unsigned la = (unsigned)(size_t)(__attribute__((address_space(3))) char*)plds + (threadIdx.x & 255) * 16;
                f32x4_d fr[6];
                for (int it = 0; it < iters; ++it) {
#pragma unroll
                    for (int q = 0; q < 6; ++q) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fr[q]) : "v"(la), "n"(q * 4096) : "memory");
#pragma unroll
                    for (int k = 0; k < 8; ++k) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[k]) : "v"(a), "v"(b));
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                    for (int q = 0; q < 6; ++q) asm volatile("" ::"v"(fr[q]));
                }
            } else
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int k = 0; k < 8; ++k) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[k]) : "v"(a), "v"(b));
            }
            t1 = (long long)__builtin_amdgcn_s_memtime();
#pragma unroll
            for (int k = 0; k < 8; ++k) out += acc[k][0] + acc[k][15];
        }
        if (threadIdx.x == 0) sink[2 * blockIdx.x] = t1 - t0;
        if (out == 12345.678f) sink[2 * blockIdx.x] = 0;
        return;
    }
    if (!(flags & 2)) return;
    if (flags & 8) __builtin_amdgcn_s_setprio(3);
    float v[8], w[8];
    f32x2_d pv[8];
    unsigned u[8];
    const float sc = 1.0000001f, sh = 1e-9f;
    const f32x2_d psc = {1.0000001f, 0.9999999f}, psh = {1e-9f, -1e-9f};
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        v[e] = 0.1f * e + threadIdx.x * 1e-3f;
        w[e] = 0.f;
        pv[e] = f32x2_d{v[e], -v[e]};
        u[e] = 0u;
    }
    const unsigned laddr = (unsigned)(size_t)(__attribute__((address_space(3))) char*)plds + (threadIdx.x & 255) * 16;
    f32x4_d lv = {1.f, 2.f, 3.f, 4.f};
    const long long t0 = (long long)__builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        for (int j = 0; j < n_inner; ++j) {
            if constexpr (KIND == 8) {  // one halo dword (two channels), plain instructions
                asm volatile("v_fma_f32 %0, %4, %5, %6\n\tv_fma_f32 %1, %4, %5, %6\n\tv_fma_f32 %2, %4, %6, %5\n\tv_fma_f32 %3, %4, %6, %5"
                             : "=&v"(w[0]), "=&v"(w[1]), "=&v"(w[2]), "=&v"(w[3]) : "v"(v[0]), "v"(sc), "v"(sh));
                asm volatile("v_exp_f32 %0, %0\n\tv_exp_f32 %1, %1\n\tv_add_f32 %0, 1.0, %0\n\tv_add_f32 %1, 1.0, %1\n\tv_rcp_f32 %0, %0\n\tv_rcp_f32 %1, %1"
                             : "+v"(w[2]), "+v"(w[3]));
                asm volatile("v_mul_f32 %0, %0, %2\n\tv_mul_f32 %1, %1, %3\n\tv_cvt_pk_bf16_f32 %4, %0, %1"
                             : "+v"(w[0]), "+v"(w[1]), "+v"(w[2]), "+v"(w[3]), "=v"(u[0]));
            } else if constexpr (KIND == 9) {  // the same, packed
                f32x2_d y, e2;
                asm volatile("v_pk_fma_f32 %0, %2, %3, %4\n\tv_pk_fma_f32 %1, %2, %4, %3" : "=&v"(y), "=&v"(e2) : "v"(pv[0]), "v"(psc), "v"(psh));
                e2 = f32x2_d{__builtin_amdgcn_exp2f(e2[0]), __builtin_amdgcn_exp2f(e2[1])};
                asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(e2) : "v"(psc));
                e2 = f32x2_d{__builtin_amdgcn_rcpf(e2[0]), __builtin_amdgcn_rcpf(e2[1])};
                asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(y) : "v"(e2));
                asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(u[0]) : "v"(y[0]), "v"(y[1]));
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    if constexpr (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[e]) : "v"(sc), "v"(sh));
                    if constexpr (KIND == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(pv[e]) : "v"(psc), "v"(psh));
                    if constexpr (KIND == 2) asm volatile("v_exp_f32 %0, %1" : "=v"(w[e]) : "v"(v[e]));
                    if constexpr (KIND == 3) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(u[e]) : "v"(v[e]), "v"(sc));
                    if constexpr (KIND == 4) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(pv[e]) : "v"(psc));
                    if constexpr (KIND == 5) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(pv[e]) : "v"(psh));
                    if constexpr (KIND == 6) asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[e]) : "v"(sh));
                    if constexpr (KIND == 7) asm volatile("v_rcp_f32 %0, %1" : "=v"(w[e]) : "v"(v[e]));
                    if constexpr (KIND == 10) asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(laddr), "v"(lv), "n"((e & 1) * 4096) : "memory");
                    if constexpr (KIND == 11) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(lv) : "v"(laddr), "n"((e & 1) * 4096) : "memory");
                    if (KIND >= 13 && e != 0) continue;  // ONE request per inner iteration: n_inner / 8 per MFMA slot (the conv's producers: 1 / 8)
                    if constexpr (KIND == 13) {  // LDS-DMA of 1 KB: eight 128-byte rows 2304 bytes apart (a filter tap's access shape), 16 KB ring per wave
                        const unsigned l = (unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned)(size_t)(__attribute__((address_space(3))) char*)plds + (unsigned)((wave - 4) * 16384 + (j & 15) * 1024)));
                        const char* g = src + (size_t)(((threadIdx.x & 63) >> 3) * 2304 + (threadIdx.x & 7) * 16) + (size_t)(((it * 5 + j) & 63) * 8 * 2304);
                        asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(g), "s"(l) : "m0", "memory");
                    }
                    if constexpr (KIND == 14) {  // the same 1 KB into registers
                        const char* g = src + (size_t)(((threadIdx.x & 63) >> 3) * 2304 + (threadIdx.x & 7) * 16) + (size_t)(((it * 5 + j) & 63) * 8 * 2304);
                        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(lv) : "v"(g) : "memory");
                    }
                    if constexpr (KIND == 12) {
                        if (e & 1) asm volatile("v_and_b32 %0, 0xffff0000, %1" : "=v"(u[e]) : "v"(u[e ^ 1]));
                        else asm volatile("v_lshlrev_b32 %0, 16, %1" : "=v"(u[e]) : "v"(u[e ^ 1]));
                    }
                }
                if constexpr (KIND == 10 || KIND == 11) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if constexpr (KIND == 13 || KIND == 14) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");  // eight requests stay in flight per wave
            }
        }
    }
    const long long t1 = (long long)__builtin_amdgcn_s_memtime();
    float out = lv[0] + lv[3];
#pragma unroll
    for (int e = 0; e < 8; ++e) out += v[e] + w[e] + pv[e][0] + pv[e][1] + (float)u[e];
    if (threadIdx.x == 256) sink[2 * blockIdx.x + 1] = t1 - t0;
    if (out == 12345.678f) sink[2 * blockIdx.x + 1] = 0;
#endif
}
int launch_issue_probe(int kind, int n_inner, int iters, int flags, int blocks, long long* sink, const void* src, hipStream_t s) {
    DMME_REQUIRE((kind < 13 || src) && kind >= 0 && kind <= 14 && n_inner >= 0 && iters > 0 && blocks > 0 && sink, DMME_ERR_INVALID, "issue_probe: bad argument");
#define IP_CASE(K) if (kind == K) hipLaunchKernelGGL(issue_probe_kernel<K>, dim3((unsigned)blocks), dim3(512), 0, s, n_inner, iters, flags, sink, (const char*)src);
    IP_CASE(13) IP_CASE(14) IP_CASE(0) IP_CASE(1) IP_CASE(2) IP_CASE(3) IP_CASE(4) IP_CASE(5) IP_CASE(6) IP_CASE(7) IP_CASE(8) IP_CASE(9) IP_CASE(10) IP_CASE(11) IP_CASE(12)
#undef IP_CASE
    DMME_CHECK_LAUNCH();
    return DMME_OK;
}

int launch_l2_stream(const void* buf, int64_t bytes, int iters, int mode, int depth, int blocks, unsigned* sink, hipStream_t s) {
    const int nvec = (int)(bytes / 16);
    DMME_REQUIRE(buf && sink && nvec >= 16 * 256 && iters > 0 && blocks > 0, DMME_ERR_INVALID, "l2_stream: bad argument");
    const dim3 g((unsigned)blocks), b(256);
#define L2S_CASE(D)                                                                                                             \
    if (depth == D) {                                                                                                           \
        if (mode == 0) hipLaunchKernelGGL(l2_stream_kernel<D>, g, b, 0, s, (const uint4*)buf, nvec, iters, sink);               \
        else hipLaunchKernelGGL(l2_stream_dma_kernel<D>, g, b, D * 4096, s, (const uint4*)buf, nvec, iters, sink, mode >= 2 ? mode : 0); \
    }
    L2S_CASE(1) L2S_CASE(2) L2S_CASE(4) L2S_CASE(8) L2S_CASE(16)
#undef L2S_CASE
    DMME_CHECK_LAUNCH();
    return DMME_OK;
}

}  // namespace dmme
