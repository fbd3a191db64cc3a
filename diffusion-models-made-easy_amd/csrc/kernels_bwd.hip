// Backward kernels of the training step (shape-generic versions).
//
// Data-gradient of every convolution is NOT here: it is the forward implicit-GEMM kernel
// run on dY with transposed + tap-flipped weights (plan.hip).  This file holds what has no
// forward twin: weight gradients, bias / time-embedding column sums, GroupNorm(+SiLU
// +Dropout2d) backward, attention backward, gradient accumulation (channel split of
// torch.cat, 2x2 sum-pool of nn.Upsample), the small linears of the time MLP.
// Reference math: torch autograd of models/ddpm.py:118-133 (ResBlock), :54-75 (Attention).
#include <stdlib.h>

#include "common.h"

namespace dmme {

// four 16-bit values of type T held in two dwords -> fp32 (bf16: shifts / masks; IEEE half: conversions)
template <typename T>
__device__ __forceinline__ void unpack4_16(const uint2& raw, float (&v)[4]) {
    if constexpr (dtype_of<T>::value == DMME_BF16) {
        v[0] = __uint_as_float(raw.x << 16); v[1] = __uint_as_float(raw.x & 0xffff0000u);
        v[2] = __uint_as_float(raw.y << 16); v[3] = __uint_as_float(raw.y & 0xffff0000u);
    } else {
        typedef T t4 __attribute__((ext_vector_type(4)));
        const t4 x = __builtin_bit_cast(t4, raw);
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = (float)x[j];
    }
}

__device__ __forceinline__ float silu_grad(float u) {
    const float s = 1.0f / (1.0f + expf(-u));
    return s * (1.0f + u * (1.0f - s));
}

// ------------------------------------------------------------------ weight gradient (generic)
// dW[co][ci][kh][kw] += sum_{n,oy,ox} dY[n,oy,ox,co] * v[n, oy*s-pad+kh, ox*s-pad+kw, ci]
// with v = act(x) recomputed through the forward prologue.  One thread per weight element.
template <typename T>
__global__ void __launch_bounds__(256) wgrad_generic_kernel(ConvArgs a, const T* __restrict__ dY, float* __restrict__ dW, int rows_per_chunk) {
    const int Cin = a.C1 + a.C2;
    const int64_t total = (int64_t)a.Cout * a.taps * Cin;
    const int Hv = a.up ? 2 * a.Hin : a.Hin, Wv = a.up ? 2 * a.Win : a.Win;
    const int k = a.taps == 9 ? 3 : 1, pad = a.taps == 9 ? 1 : 0;
    // blockIdx.y selects a chunk of (n, oy) output rows; partial sums are merged with atomics
    const int row_begin = blockIdx.y * rows_per_chunk;
    const int row_end = min(row_begin + rows_per_chunk, a.N * a.Hout);
    for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int ci = (int)(idx % Cin);
        const int tap = (int)((idx / Cin) % a.taps);
        const int co = (int)(idx / ((int64_t)Cin * a.taps));
        const int kh = tap / k, kw = tap % k;
        float acc = 0.f;
        for (int row = row_begin; row < row_end; ++row) {
            const int n = row / a.Hout, oy = row % a.Hout;
            const int iy = oy * a.stride - pad + kh;
            if (iy < 0 || iy >= Hv) continue;
            const int sy = a.up ? (iy >> 1) : iy;
            float sc = 1.f, sh = 0.f, dm = 1.f;
            if (a.scale) {
                sc = a.scale[(int64_t)n * Cin + ci];
                sh = a.shift[(int64_t)n * Cin + ci];
            }
            if (a.dmask) dm = a.dmask[(int64_t)n * Cin + ci];
            for (int ox = 0; ox < a.Wout; ++ox) {
                const int ix = ox * a.stride - pad + kw;
                if (ix < 0 || ix >= Wv) continue;
                const int sx = a.up ? (ix >> 1) : ix;
                float v;
                if (a.in_nchw)
                    v = ((const float*)a.src1)[(((int64_t)n * a.C1 + ci) * a.Hin + sy) * a.Win + sx];
                else {
                    const int64_t pix = ((int64_t)n * a.Hin + sy) * a.Win + sx;
                    v = ci < a.C1 ? to_f(((const T*)a.src1)[pix * a.C1 + ci]) : to_f(((const T*)a.src2)[pix * a.C2 + (ci - a.C1)]);
                }
                if (a.scale) v = fmaf(v, sc, sh);
                if (a.pro_silu) v = silu_f(v);
                if (a.dmask) v *= dm;
                if (!a.in_nchw) v = to_f(from_f<T>(v));
                acc = fmaf(to_f(dY[(((int64_t)n * a.Hout + oy) * a.Wout + ox) * a.Cout + co]), v, acc);
            }
        }
        atomicAdd(&dW[((int64_t)co * Cin + ci) * a.taps + tap], acc);
    }
}

int launch_wgrad_generic(int dtype, const ConvArgs& a, const void* dY, float* dW, hipStream_t s) {
    const int64_t total = (int64_t)a.Cout * a.taps * (a.C1 + a.C2);
    int64_t blocks = (total + 255) / 256;
    if (blocks > 65536) blocks = 65536;
    // split the pixel loop so that small-weight convs (first / last layer) still fill the GPU
    const int rows = a.N * a.Hout;
    int chunks = (int)((4096 + blocks - 1) / blocks);
    if (chunks > rows) chunks = rows;
    if (chunks < 1) chunks = 1;
    const int rpc = (rows + chunks - 1) / chunks;
    chunks = (rows + rpc - 1) / rpc;
    dim3 grid((unsigned)blocks, (unsigned)chunks);
    if (dtype == DMME_BF16)
        hipLaunchKernelGGL(wgrad_generic_kernel<bf16>, grid, dim3(256), 0, s, a, (const bf16*)dY, dW, rpc);
    else if (dtype == DMME_F16)
        hipLaunchKernelGGL(wgrad_generic_kernel<f16>, grid, dim3(256), 0, s, a, (const f16*)dY, dW, rpc);
    else
        hipLaunchKernelGGL(wgrad_generic_kernel<float>, grid, dim3(256), 0, s, a, (const float*)dY, dW, rpc);
    DMME_CHECK_LAUNCH();
    return DMME_OK;
}

// ------------------------------------------------------------------ weight gradient, few output channels
// Cout <= 4 (the network's last conv): one workgroup per chunk of output rows, thread <-> (tap, 16-byte
// cin vector) holding Cout*EPV accumulators; the activation vector is recomputed through the prologue.
template <typename T>
__global__ void __launch_bounds__(256) wgrad_cout_small_kernel(ConvArgs a, const T* __restrict__ dY, float* __restrict__ dW, int rows_per_chunk) {
    constexpr int EPV = 16 / sizeof(T);
    const int Cin = a.C1;  // single source, Cin % EPV == 0
    const int vpt = Cin / EPV;                 // vectors per tap
    const int slot = threadIdx.x;              // (tap, vector)
    if (slot >= 9 * vpt) return;
    const int tap = slot / vpt, cv = slot % vpt, kh = tap / 3, kw = tap % 3, c0 = cv * EPV;
    float acc[4][EPV];
#pragma unroll
    for (int co = 0; co < 4; ++co)
#pragma unroll
        for (int j = 0; j < EPV; ++j) acc[co][j] = 0.f;
    const int row_begin = blockIdx.x * rows_per_chunk, row_end = min(row_begin + rows_per_chunk, a.N * a.Hout);
    for (int row = row_begin; row < row_end; ++row) {
        const int n = row / a.Hout, oy = row % a.Hout, iy = oy - 1 + kh;
        if (iy < 0 || iy >= a.Hin) continue;
        float sc[EPV], sh[EPV];
#pragma unroll
        for (int j = 0; j < EPV; ++j) {
            sc[j] = a.scale ? a.scale[(int64_t)n * Cin + c0 + j] : 1.f;
            sh[j] = a.scale ? a.shift[(int64_t)n * Cin + c0 + j] : 0.f;
        }
        for (int ox = 0; ox < a.Wout; ++ox) {
            const int ix = ox - 1 + kw;
            if (ix < 0 || ix >= a.Win) continue;
            const T* xp = (const T*)a.src1 + (((int64_t)n * a.Hin + iy) * a.Win + ix) * Cin + c0;
            float v[EPV];
#pragma unroll
            for (int j = 0; j < EPV; ++j) {
                float t = fmaf(to_f(xp[j]), sc[j], sh[j]);
                if (a.pro_silu) t = sizeof(T) == 2 ? silu_fast(t) : silu_f(t);  // same SiLU flavour as the forward prologue
                v[j] = to_f(from_f<T>(t));
            }
            const T* dp = dY + (((int64_t)n * a.Hout + oy) * a.Wout + ox) * a.Cout;
#pragma unroll
            for (int co = 0; co < 4; ++co) {
                if (co >= a.Cout) break;
                const float d = to_f(dp[co]);
#pragma unroll
                for (int j = 0; j < EPV; ++j) acc[co][j] = fmaf(d, v[j], acc[co][j]);
            }
        }
    }
    for (int co = 0; co < a.Cout && co < 4; ++co)
#pragma unroll
        for (int j = 0; j < EPV; ++j) atomicAdd(&dW[((int64_t)co * Cin + c0 + j) * 9 + tap], acc[co][j]);
}

// Cin <= 4 on the NCHW fp32 network input (the first conv): thread <-> cout with 9*Cin accumulators;
// the 9*Cin input values of a pixel are wave-uniform loads.
template <typename T, int CIN>
__global__ void __launch_bounds__(256) wgrad_cin_small_kernel(ConvArgs a, const T* __restrict__ dY, float* __restrict__ dW, int rows_per_chunk) {
    constexpr int Cin = CIN;
    const int co = threadIdx.x;
    if (co >= a.Cout) return;
    float acc[CIN * 9];
#pragma unroll
    for (int k = 0; k < CIN * 9; ++k) acc[k] = 0.f;
    const int row_begin = blockIdx.x * rows_per_chunk, row_end = min(row_begin + rows_per_chunk, a.N * a.Hout);
    const float* x = (const float*)a.src1;
    for (int row = row_begin; row < row_end; ++row) {
        const int n = row / a.Hout, oy = row % a.Hout;
        for (int ox = 0; ox < a.Wout; ++ox) {
            const float d = to_f(dY[(((int64_t)n * a.Hout + oy) * a.Wout + ox) * a.Cout + co]);
#pragma unroll
            for (int ci = 0; ci < CIN; ++ci) {
#pragma unroll
                for (int tap = 0; tap < 9; ++tap) {
                    const int iy = oy - 1 + tap / 3, ix = ox - 1 + tap % 3;
                    float v = 0.f;
                    if (iy >= 0 && iy < a.Hin && ix >= 0 && ix < a.Win) v = x[(((int64_t)n * Cin + ci) * a.Hin + iy) * a.Win + ix];
                    acc[ci * 9 + tap] = fmaf(d, v, acc[ci * 9 + tap]);
                }
            }
        }
    }
#pragma unroll
    for (int ci = 0; ci < CIN; ++ci)
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) atomicAdd(&dW[((int64_t)co * Cin + ci) * 9 + tap], acc[ci * 9 + tap]);
}

// Thin 3x3 weight gradients (first conv: Cin <= 3 on the NCHW fp32 input; last conv: Cout <= 3), stride 1:
//   out[k][c] = sum_p wide[p][c] * coef[p][k],   k = g*9 + tap,  g < G <= 3
// first conv: wide = dY [p][Cout],            coef[p][ci*9+tap] = x[n][ci][y-1+kh][x-1+kw]
// last conv:  wide = prologue(input) [q][Cin], coef[q][co*9+tap] = dY[n][y+1-kh][x+1-kw][co]   (substituting q = p + tap)
// so the wide NHWC tensor is read exactly once, without a halo.  A workgroup stages the 27 coefficients of 256
// pixels in LDS, thread <-> (4 channels, pixel lane) keeps 4 x 27 accumulators, the pixel lanes are merged through
// LDS and each workgroup ends with one atomic per weight element.
template <typename T, bool FIRST>
__global__ void __launch_bounds__(256) wgrad_thin_kernel(ConvArgs a, const T* __restrict__ dY, float* __restrict__ dW, int tiles_total, int g0) {
    constexpr int PX = 256, KP = 28;
    __shared__ __attribute__((aligned(16))) float lds[256 * 9 * 4];  // coef [PX][KP] (28 KB), then the merge buffer [lanes][9][Cw] (36 KB)
    const int Cw = FIRST ? a.Cout : a.C1, G = FIRST ? a.C1 : a.Cout;
    const int H = a.Hout, W = a.Wout, HW = H * W;
    const int vecs = Cw / 4, lanes_p = 256 / vecs;
    const int tid = threadIdx.x, vec = tid % vecs, pl = tid / vecs;
    const T* wide = FIRST ? dY : (const T*)a.src1;
    const float* xin = (const float*)a.src1;
    float acc[4][27];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int k = 0; k < 27; ++k) acc[j][k] = 0.f;
    const int total = a.N * HW;  // < 2^31 (checked by the launcher)
    for (int tile = blockIdx.x; tile < tiles_total; tile += gridDim.x) {
        const int p0 = tile * PX;
        // 16-bit tensors at >= 8 pixel lanes (every shape the networks have): the thread's <= 32 vectors of the wide tensor are
        // requested HERE, in front of the coefficient gathers - one round trip for both instead of one for the gathers and then one
        // per four pixels of the loop below (that was this kernel: ~20 dependent round trips per tile, 63 us for 33 MB)
        constexpr int NPRE = 32;
        const bool pre = sizeof(T) == 2 && lanes_p * NPRE >= PX;
        uint2 rawp[NPRE];
        if (pre) {
#pragma unroll
            for (int i = 0; i < NPRE; ++i) {
                const int pi = pl + i * lanes_p, p = p0 + pi;
                rawp[i] = make_uint2(0u, 0u);
                if (pi < PX && p < total) rawp[i] = *reinterpret_cast<const uint2*>(wide + (int64_t)p * Cw + 4 * vec);
            }
        }
        __syncthreads();
        {   // thread <-> pixel: its 27 coefficients are 27 independent gathers
            const int p = p0 + tid;
            const int n = p / HW, rem = p - n * HW, y = rem / W, x = rem - y * W;
            float cv[27];
#pragma unroll
            for (int k = 0; k < 27; ++k) {
                const int g = k / 9, tap = k % 9, kh = tap / 3, kw = tap % 3;
                const int sy = FIRST ? y - 1 + kh : y + 1 - kh, sx = FIRST ? x - 1 + kw : x + 1 - kw;
                float v = 0.f;
                if (p < total && g0 + g < G && sy >= 0 && sy < H && sx >= 0 && sx < W)
                    v = FIRST ? xin[(((int64_t)n * G + g0 + g) * H + sy) * W + sx] : to_f(dY[(((int64_t)n * H + sy) * W + sx) * G + g0 + g]);
                cv[k] = v;
            }
#pragma unroll
            for (int k = 0; k < 27; ++k) lds[tid * KP + k] = cv[k];
        }
        __syncthreads();
        // GroupNorm scale/shift of this thread's 4 channels: per tile when the tile lies in one image, else per pixel
        const int n_first = p0 / HW;
        const bool one_image = (min(p0 + PX, total) - 1) / HW == n_first;
        float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = make_float4(0.f, 0.f, 0.f, 0.f);
        if constexpr (!FIRST) {
            if (a.scale && one_image) {
                const int64_t so = (int64_t)n_first * Cw + 4 * vec;
                sc = *reinterpret_cast<const float4*>(a.scale + so);
                sh = *reinterpret_cast<const float4*>(a.shift + so);
            }
        }
#define WTHIN_PIXEL(PI, RAWP)                                                                                                   \
        {                                                                                                                      \
            const int pi = (PI);                                                                                               \
            const int p = p0 + pi;                                                                                             \
            const bool in = p < total;                                                                                         \
            float v[4] = {0.f, 0.f, 0.f, 0.f};                                                                                 \
            const uint2 raw = (RAWP);                                                                                          \
            if constexpr (sizeof(T) == 2) unpack4_16<T>(raw, v);                                                               \
            if constexpr (!FIRST) {                                                                                            \
                if (a.scale && !one_image && in) {                                                                             \
                    const int64_t so = (int64_t)(p / HW) * Cw + 4 * vec;                                                       \
                    sc = *reinterpret_cast<const float4*>(a.scale + so);                                                       \
                    sh = *reinterpret_cast<const float4*>(a.shift + so);                                                       \
                }                                                                                                              \
                v[0] = fmaf(v[0], sc.x, sh.x); v[1] = fmaf(v[1], sc.y, sh.y); v[2] = fmaf(v[2], sc.z, sh.z); v[3] = fmaf(v[3], sc.w, sh.w); \
                _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                                \
                    if (a.pro_silu) v[j] = silu_fast(v[j]);                                                                    \
                    v[j] = in ? to_f(from_f<T>(v[j])) : 0.f;                                                                   \
                }                                                                                                              \
            }                                                                                                                  \
            const float4* cp = reinterpret_cast<const float4*>(lds + pi * KP);                                                 \
            _Pragma("unroll") for (int k4 = 0; k4 < 7; ++k4) {                                                                 \
                const float4 c = cp[k4];                                                                                       \
                _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                                \
                    acc[j][4 * k4] = fmaf(v[j], c.x, acc[j][4 * k4]);                                                          \
                    if (4 * k4 + 1 < 27) acc[j][4 * k4 + 1] = fmaf(v[j], c.y, acc[j][4 * k4 + 1]);                             \
                    if (4 * k4 + 2 < 27) acc[j][4 * k4 + 2] = fmaf(v[j], c.z, acc[j][4 * k4 + 2]);                             \
                    if (4 * k4 + 3 < 27) acc[j][4 * k4 + 3] = fmaf(v[j], c.w, acc[j][4 * k4 + 3]);                             \
                }                                                                                                              \
            }                                                                                                                  \
        }
        if (pre) {
            if constexpr (sizeof(T) == 2) {
#pragma unroll
                for (int i = 0; i < NPRE; ++i)
                    if (pl + i * lanes_p < PX) WTHIN_PIXEL(pl + i * lanes_p, rawp[i])
            }
        } else
#pragma unroll 4
        for (int pi = pl; pi < PX; pi += lanes_p) {
            const int p = p0 + pi;
            const bool in = p < total;
            float v[4] = {0.f, 0.f, 0.f, 0.f};
            if constexpr (sizeof(T) == 2) {
                uint2 raw = make_uint2(0u, 0u);
                if (in) raw = *reinterpret_cast<const uint2*>(wide + (int64_t)p * Cw + 4 * vec);
                unpack4_16<T>(raw, v);
            } else {
                float4 raw = make_float4(0.f, 0.f, 0.f, 0.f);
                if (in) raw = *reinterpret_cast<const float4*>(wide + (int64_t)p * Cw + 4 * vec);
                v[0] = raw.x; v[1] = raw.y; v[2] = raw.z; v[3] = raw.w;
            }
            if constexpr (!FIRST) {
                if (a.scale && !one_image && in) {
                    const int64_t so = (int64_t)(p / HW) * Cw + 4 * vec;
                    sc = *reinterpret_cast<const float4*>(a.scale + so);
                    sh = *reinterpret_cast<const float4*>(a.shift + so);
                }
                v[0] = fmaf(v[0], sc.x, sh.x); v[1] = fmaf(v[1], sc.y, sh.y); v[2] = fmaf(v[2], sc.z, sh.z); v[3] = fmaf(v[3], sc.w, sh.w);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (a.pro_silu) v[j] = sizeof(T) == 2 ? silu_fast(v[j]) : silu_f(v[j]);  // same SiLU flavour as the forward prologue
                    v[j] = in ? to_f(from_f<T>(v[j])) : 0.f;
                }
            }
            const float4* cp = reinterpret_cast<const float4*>(lds + pi * KP);
#pragma unroll
            for (int k4 = 0; k4 < 7; ++k4) {
                const float4 c = cp[k4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    acc[j][4 * k4] = fmaf(v[j], c.x, acc[j][4 * k4]);
                    if (4 * k4 + 1 < 27) acc[j][4 * k4 + 1] = fmaf(v[j], c.y, acc[j][4 * k4 + 1]);
                    if (4 * k4 + 2 < 27) acc[j][4 * k4 + 2] = fmaf(v[j], c.z, acc[j][4 * k4 + 2]);
                    if (4 * k4 + 3 < 27) acc[j][4 * k4 + 3] = fmaf(v[j], c.w, acc[j][4 * k4 + 3]);
                }
            }
        }
#undef WTHIN_PIXEL
    }
#pragma unroll
    for (int g = 0; g < 3; ++g) {
        if (g0 + g >= G) break;
        __syncthreads();
#pragma unroll
        for (int t = 0; t < 9; ++t)
            *reinterpret_cast<float4*>(lds + ((int64_t)pl * 9 + t) * Cw + 4 * vec) =
                make_float4(acc[0][g * 9 + t], acc[1][g * 9 + t], acc[2][g * 9 + t], acc[3][g * 9 + t]);
        __syncthreads();
        // every workgroup adds into the same few KB: lanes walk the destination in address order (c-major, tap-minor),
        // so one wave instruction touches 2 (last conv) or ~7 (first conv) cache lines instead of one per lane
        for (int e = tid; e < 9 * Cw; e += 256) {
            const int c = e / 9, t = e - c * 9;
            float sum = 0.f;
            for (int l = 0; l < lanes_p; ++l) sum += lds[(l * 9 + t) * Cw + c];
            const int64_t idx = FIRST ? ((int64_t)c * G + g0 + g) * 9 + t : ((int64_t)(g0 + g) * Cw + c) * 9 + t;
            atomicAdd(dW + idx, sum);
        }
    }
}

static bool wgrad_thin_supported(const ConvArgs& a) {
    if (a.taps != 9 || a.stride != 1 || a.up || a.C2 || a.dmask || a.Hout != a.Hin || a.Wout != a.Win) return false;
    const int Cw = a.in_nchw ? a.Cout : a.C1, G = a.in_nchw ? a.C1 : a.Cout;
    if (G > 6 || Cw % 4 || Cw / 4 > 256 || 256 % (Cw / 4)) return false;  // thin side: 3 channels per pass (6 = the IDDPM (eps, v) output)
    if ((int64_t)a.N * a.Hout * a.Wout * (Cw > 27 ? Cw : 27) >= (1ll << 31)) return false;
    if (a.in_nchw) return !a.scale && !a.pro_silu;
    return true;
}

bool wgrad_small_supported(int dtype, const ConvArgs& a) {
    if (wgrad_thin_supported(a)) return true;
    const int EPV = is16(dtype) ? 8 : 4;
    if (a.taps != 9 || a.stride != 1 || a.up || a.C2 || a.dmask) return false;
    if (a.in_nchw) return a.C1 <= 4 && a.Cout <= 256 && !a.scale && !a.pro_silu;
    return a.Cout <= 4 && a.C1 % EPV == 0 && 9 * (a.C1 / EPV) <= 256;
}
int launch_wgrad_small(int dtype, const ConvArgs& a, const void* dY, float* dW, hipStream_t s) {
    if (wgrad_thin_supported(a) && !debug_route("no_wgrad_thin")) {
        const int64_t total = (int64_t)a.N * a.Hout * a.Wout;
        const int tiles = (int)((total + 255) / 256), grid = tiles < 256 ? tiles : 256;  // few workgroups: they all end in atomics on the same addresses
#define DMME_WTHIN(TT, FF) hipLaunchKernelGGL((wgrad_thin_kernel<TT, FF>), dim3(grid), dim3(256), 0, s, a, (const TT*)dY, dW, tiles, g0)
        const int G = a.in_nchw ? a.C1 : a.Cout;
        for (int g0 = 0; g0 < G; g0 += 3) {
            if (dtype == DMME_BF16) {
                if (a.in_nchw) DMME_WTHIN(bf16, true); else DMME_WTHIN(bf16, false);
            } else if (dtype == DMME_F16) {
                if (a.in_nchw) DMME_WTHIN(f16, true); else DMME_WTHIN(f16, false);
            } else {
                if (a.in_nchw) DMME_WTHIN(float, true); else DMME_WTHIN(float, false);
            }
            DMME_CHECK_LAUNCH();
        }
#undef DMME_WTHIN
        return DMME_OK;
    }
    const int rows = a.N * a.Hout;
    int chunks = rows < 256 ? rows : 256;  // few, long chunks: every chunk ends in one atomic per weight element (contended addresses)
    const int rpc = (rows + chunks - 1) / chunks;
    chunks = (rows + rpc - 1) / rpc;
    if (a.in_nchw) {
#define DMME_WCIN(TT, CC) hipLaunchKernelGGL((wgrad_cin_small_kernel<TT, CC>), dim3(chunks), dim3(256), 0, s, a, (const TT*)dY, dW, rpc)
        if (dtype == DMME_BF16) {
            switch (a.C1) { case 1: DMME_WCIN(bf16, 1); break; case 2: DMME_WCIN(bf16, 2); break; case 3: DMME_WCIN(bf16, 3); break; default: DMME_WCIN(bf16, 4); }
        } else if (dtype == DMME_F16) {
            switch (a.C1) { case 1: DMME_WCIN(f16, 1); break; case 2: DMME_WCIN(f16, 2); break; case 3: DMME_WCIN(f16, 3); break; default: DMME_WCIN(f16, 4); }
        } else {
            switch (a.C1) { case 1: DMME_WCIN(float, 1); break; case 2: DMME_WCIN(float, 2); break; case 3: DMME_WCIN(float, 3); break; default: DMME_WCIN(float, 4); }
        }
#undef DMME_WCIN
    } else {
        if (dtype == DMME_BF16)
            hipLaunchKernelGGL(wgrad_cout_small_kernel<bf16>, dim3(chunks), dim3(256), 0, s, a, (const bf16*)dY, dW, rpc);
        else if (dtype == DMME_F16)
            hipLaunchKernelGGL(wgrad_cout_small_kernel<f16>, dim3(chunks), dim3(256), 0, s, a, (const f16*)dY, dW, rpc);
        else
            hipLaunchKernelGGL(wgrad_cout_small_kernel<float>, dim3(chunks), dim3(256), 0, s, a, (const float*)dY, dW, rpc);
    }
    DMME_CHECK_LAUNCH();
    return DMME_OK;
}

// ------------------------------------------------------------------ column sums of dY
// rowsum[n][c] = sum over the image's pixels of dY[n, :, c]  (coalesced over c)
template <typename T>
__global__ void __launch_bounds__(256) colsum_kernel(const T* __restrict__ dY, int HW, int C, float* __restrict__ rowsum) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    const int n = blockIdx.y;
    if (c >= C) return;
    const T* p = dY + (int64_t)n * HW * C + c;
    float acc = 0.f;
    for (int i = 0; i < HW; ++i) acc += to_f(p[(int64_t)i * C]);
    rowsum[(int64_t)n * C + c] = acc;
}
// dbias[c] += sum_n rowsum[n][c];  d_tproj rows: per image (nt == N) or the single broadcast row
__global__ void __launch_bounds__(256) bias_tproj_kernel(const float* __restrict__ rowsum, int N, int C, float* __restrict__ dbias,
                                                         float* __restrict__ dtproj, int ld, int nt) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    float tot = 0.f;
    for (int n = 0; n < N; ++n) {
        const float v = rowsum[(int64_t)n * C + c];
        tot += v;
        if (dtproj && nt > 1) dtproj[(int64_t)n * ld + c] = v;
    }
    if (dbias) dbias[c] += tot;
    if (dtproj && nt == 1) dtproj[c] = tot;
}
// few channels (the 3-channel output conv): one workgroup per image, threads over the pixels, LDS reduction per channel
template <typename T>
__global__ void __launch_bounds__(256) colsum_thin_kernel(const T* __restrict__ dY, int HW, int C, float* __restrict__ rowsum) {
    __shared__ float red[256];
    const int n = blockIdx.x, tid = threadIdx.x;
    for (int c = 0; c < C; ++c) {
        float acc = 0.f;
        for (int p = tid; p < HW; p += 256) acc += to_f(dY[((int64_t)n * HW + p) * C + c]);
        red[tid] = acc;
        __syncthreads();
        for (int o = 128; o > 0; o >>= 1) {
            if (tid < o) red[tid] += red[tid + o];
            __syncthreads();
        }
        if (tid == 0) rowsum[(int64_t)n * C + c] = red[0];
        __syncthreads();
    }
}

int launch_colsum(int dtype, const void* dY, int N, int HW, int C, float* rowsum, float* dbias, float* dtproj, int ld, int nt,
                  hipStream_t s) {
    dim3 grid((C + 255) / 256, N);
    if (C <= 8 && HW >= 256) {
        if (dtype == DMME_BF16)
            hipLaunchKernelGGL(colsum_thin_kernel<bf16>, dim3(N), dim3(256), 0, s, (const bf16*)dY, HW, C, rowsum);
        else if (dtype == DMME_F16)
            hipLaunchKernelGGL(colsum_thin_kernel<f16>, dim3(N), dim3(256), 0, s, (const f16*)dY, HW, C, rowsum);
        else
            hipLaunchKernelGGL(colsum_thin_kernel<float>, dim3(N), dim3(256), 0, s, (const float*)dY, HW, C, rowsum);
    } else if (dtype == DMME_BF16)
        hipLaunchKernelGGL(colsum_kernel<bf16>, grid, dim3(256), 0, s, (const bf16*)dY, HW, C, rowsum);
 else if (dtype == DMME_F16)
        hipLaunchKernelGGL(colsum_kernel<f16>, grid, dim3(256), 0, s, (const f16*)dY, HW, C, rowsum);
    else
        hipLaunchKernelGGL(colsum_kernel<float>, grid, dim3(256), 0, s, (const float*)dY, HW, C, rowsum);
    DMME_CHECK_LAUNCH();
    hipLaunchKernelGGL(bias_tproj_kernel, dim3((C + 255) / 256), dim3(256), 0, s, rowsum, N, C, dbias, dtproj, ld, nt);
    DMME_CHECK_LAUNCH();
    return DMME_OK;
}

// ------------------------------------------------------------------ GroupNorm (+SiLU+mask) backward
// forward: u = gamma*xhat + beta (= x*scale + shift), v = silu(u)*mask (or v = u)
// given dv:  du = dv*mask*silu'(u);  dgamma += sum du*xhat;  dbeta += sum du
//            dx = rstd * (du*gamma - (S1 + xhat*S2)/cnt),  S1 = sum_g du*gamma, S2 = sum_g du*gamma*xhat
// One workgroup per (n, group); handles groups straddling the two concatenated sources.
template <typename T>
__global__ void __launch_bounds__(256) gn_bwd_generic_kernel(const T* __restrict__ dv, const T* __restrict__ x1, const T* __restrict__ x2,
                                                             int HW, int C1, int C2, int groups, const float* __restrict__ gamma,
                                                             const float* __restrict__ mean_rstd, const float* __restrict__ scale,
                                                             const float* __restrict__ shift, const float* __restrict__ dmask, int pro_silu,
                                                             T* __restrict__ dx1, T* __restrict__ dx2, int acc1, int acc2,
                                                             float* __restrict__ dgamma, float* __restrict__ dbeta, GnMod mod) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int n = blockIdx.y, g = blockIdx.x;
    const int C = C1 + C2, cg = C / groups;
    float* dg_l = sm;            // cg
    float* db_l = sm + cg;       // cg
    float* red = sm + 2 * cg;    // 16
    for (int j = threadIdx.x; j < 2 * cg; j += blockDim.x) sm[j] = 0.f;
    __syncthreads();
    const float mean = mean_rstd[((int64_t)n * groups + g) * 2], rstd = mean_rstd[((int64_t)n * groups + g) * 2 + 1];
    const int64_t cnt = (int64_t)cg * HW;
    auto elem = [&](int64_t e, float& du, float& xhat, int& c, int64_t& p) {
        c = g * cg + (int)(e % cg);
        p = (int64_t)n * HW + e / cg;
        const float x = c < C1 ? to_f(x1[p * C1 + c]) : to_f(x2[p * C2 + (c - C1)]);
        float d = to_f(dv[p * C + c]);
        if (dmask) d *= dmask[(int64_t)n * C + c];
        if (pro_silu) d *= silu_grad(fmaf(x, scale[(int64_t)n * C + c], shift[(int64_t)n * C + c]));
        du = d;
        xhat = (x - mean) * rstd;
    };
    float s1 = 0.f, s2 = 0.f;
    for (int64_t e = threadIdx.x; e < cnt; e += blockDim.x) {
        float du, xhat;
        int c;
        int64_t p;
        elem(e, du, xhat, c, p);
        const float gm = gamma[c] * mod.mul(n, c);
        s1 = fmaf(du, gm, s1);
        s2 = fmaf(du * gm, xhat, s2);
        atomicAdd(&dg_l[c - g * cg], du * xhat);
        atomicAdd(&db_l[c - g * cg], du);
    }
    const float S1 = block_sum(s1, red);
    const float S2 = block_sum(s2, red);
    const float inv = 1.0f / (float)cnt;
    for (int64_t e = threadIdx.x; e < cnt; e += blockDim.x) {
        float du, xhat;
        int c;
        int64_t p;
        elem(e, du, xhat, c, p);
        const float dx = rstd * (du * gamma[c] * mod.mul(n, c) - (S1 + xhat * S2) * inv);
        if (c < C1) {
            T* d = dx1 + p * C1 + c;
            *d = from_f<T>(acc1 ? to_f(*d) + dx : dx);
        } else {
            T* d = dx2 + p * C2 + (c - C1);
            *d = from_f<T>(acc2 ? to_f(*d) + dx : dx);
        }
    }
    __syncthreads();
    for (int j = threadIdx.x; j < cg; j += blockDim.x) {
        const int c = g * cg + j;
        const float m = mod.mul(n, c);
        atomicAdd(&dgamma[c], dg_l[j] * m);
        atomicAdd(&dbeta[c], db_l[j] * m);
        mod.emit(n, c, db_l[j], dg_l[j], gamma[c]);
    }
}

int launch_gn_bwd_generic(int dtype, const void* dv, const void* x1, const void* x2, int N, int HW, int C1, int C2, int groups,
                          const float* gamma, const float* mean_rstd, const float* scale, const float* shift, const float* dmask,
                          int pro_silu, void* dx1, void* dx2, int acc1, int acc2, float* dgamma, float* dbeta, GnMod mod, hipStream_t s) {
    const int cg = (C1 + C2) / groups;
    const size_t lds = (size_t)(2 * cg + 16) * sizeof(float);
    dim3 grid(groups, N);
    if (dtype == DMME_BF16)
        hipLaunchKernelGGL(gn_bwd_generic_kernel<bf16>, grid, dim3(256), lds, s, (const bf16*)dv, (const bf16*)x1, (const bf16*)x2, HW, C1, C2,
                           groups, gamma, mean_rstd, scale, shift, dmask, pro_silu, (bf16*)dx1, (bf16*)dx2, acc1, acc2, dgamma, dbeta, mod);
    else if (dtype == DMME_F16)
        hipLaunchKernelGGL(gn_bwd_generic_kernel<f16>, grid, dim3(256), lds, s, (const f16*)dv, (const f16*)x1, (const f16*)x2, HW, C1, C2,
                           groups, gamma, mean_rstd, scale, shift, dmask, pro_silu, (f16*)dx1, (f16*)dx2, acc1, acc2, dgamma, dbeta, mod);
    else
        hipLaunchKernelGGL(gn_bwd_generic_kernel<float>, grid, dim3(256), lds, s, (const float*)dv, (const float*)x1, (const float*)x2, HW, C1,
                           C2, groups, gamma, mean_rstd, scale, shift, dmask, pro_silu, (float*)dx1, (float*)dx2, acc1, acc2, dgamma, dbeta, mod);
    DMME_CHECK_LAUNCH();
    return DMME_OK;
}

// ------------------------------------------------------------------ gradient accumulation
// dst1[.., :C1] (+)= src[.., :C1], dst2[.., :C2] (+)= src[.., C1:]; with pool = 1 the source is
// 2H x 2W and each destination pixel receives the sum of its 2x2 block (nn.Upsample backward).
template <typename T>
__global__ void __launch_bounds__(256) grad_acc_kernel(const T* __restrict__ src, T* __restrict__ d1, T* __restrict__ d2, int C1, int C2,
                                                       int acc1, int acc2, int pool, int H, int W, int64_t total) {
    const int C = C1 + C2;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const int64_t p = i / C;  // n*H*W + y*W + x  (destination pixel)
        float v;
        if (pool) {
            const int x = (int)(p % W), y = (int)((p / W) % H);
            const int64_t n = p / ((int64_t)W * H);
            const int64_t b = ((n * 2 * H + 2 * y) * 2 * W + 2 * x) * C + c;
            v = to_f(src[b]) + to_f(src[b + C]) + to_f(src[b + (int64_t)2 * W * C]) + to_f(src[b + (int64_t)2 * W * C + C]);
        } else {
            v = to_f(src[i]);
        }
        if (c < C1) {
            T* d = d1 + p * C1 + c;
            *d = from_f<T>(acc1 ? to_f(*d) + v : v);
        } else {
            T* d = d2 + p * C2 + (c - C1);
            *d = from_f<T>(acc2 ? to_f(*d) + v : v);
        }
    }
}
int launch_grad_acc(int dtype, const void* src, void* d1, void* d2, int C1, int C2, int acc1, int acc2, int pool, int N, int H, int W,
                    hipStream_t s) {
    if (grad_acc_fast_supported(dtype, C1, C2, pool)) return launch_grad_acc_fast(dtype, src, d1, d2, C1, C2, acc1, acc2, (int64_t)N * H * W, s);
    const int64_t total = (int64_t)N * H * W * (C1 + C2);
    if (total == 0) return DMME_OK;
    int64_t b = (total + 255) / 256;
    if (b > 16384) b = 16384;
    if (dtype == DMME_BF16)
        hipLaunchKernelGGL(grad_acc_kernel<bf16>, dim3((unsigned)b), dim3(256), 0, s, (const bf16*)src, (bf16*)d1, (bf16*)d2, C1, C2, acc1, acc2, pool,
                           H, W, total);
    else if (dtype == DMME_F16)
        hipLaunchKernelGGL(grad_acc_kernel<f16>, dim3((unsigned)b), dim3(256), 0, s, (const f16*)src, (f16*)d1, (f16*)d2, C1, C2, acc1, acc2, pool,
                           H, W, total);
    else
        hipLaunchKernelGGL(grad_acc_kernel<float>, dim3((unsigned)b), dim3(256), 0, s, (const float*)src, (float*)d1, (float*)d2, C1, C2, acc1, acc2,
                           pool, H, W, total);
    DMME_CHECK_LAUNCH();
    return DMME_OK;
}

// ------------------------------------------------------------------ attention backward (generic)
// pass A (one workgroup per query row): recompute p_i, dP_i = dO_i V^T, dS_i = p_i (dP_i - <p_i, dP_i>),
//   keep P and dS (fp32, [N*heads][S][S]) and write dQ_i = scale * dS_i K.
// pass B (one workgroup per key row): dK_j = scale * dS[:, j]^T Q,  dV_j = P[:, j]^T dO.
// Head view as in attn_generic_kernel: (image n, head hd) reads qkv channels [hd*3d, (hd+1)*3d); its output row bh = n*heads + hd
// was stored at image bh % N, head bh / N, so that is where its dO comes from (models/iddpm.py:38-46).
template <typename T>
__global__ void __launch_bounds__(256) attn_bwd_rows_kernel(const T* __restrict__ qkv, const T* __restrict__ dO, int S, int C, int heads, int N,
                                                            float* __restrict__ P, float* __restrict__ dS, T* __restrict__ dqkv) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int d = C / heads;
    float* qs = sm;            // d
    float* dos = sm + d;       // d
    float* ps = sm + 2 * d;    // S
    float* ds = ps + S;        // S
    float* red = ds + S;       // 16
    const int bh = blockIdx.y, n = bh / heads, hd = bh % heads, i = blockIdx.x;
    const int64_t hoff = (int64_t)n * S * 3 * C + (int64_t)hd * 3 * d;
    const T* base = qkv + hoff;
    const T* dob = dO + (int64_t)(bh % N) * S * C + (int64_t)(bh / N) * d;
    const float kscale = powf((float)C, -0.5f);
    for (int c = threadIdx.x; c < d; c += blockDim.x) {
        qs[c] = to_f(base[(int64_t)i * 3 * C + c]);
        dos[c] = to_f(dob[(int64_t)i * C + c]);
    }
    __syncthreads();
    float lmax = -INFINITY;
    for (int j = threadIdx.x; j < S; j += blockDim.x) {
        const T* kr = base + (int64_t)j * 3 * C + d;
        const T* vr = base + (int64_t)j * 3 * C + 2 * d;
        float sacc = 0.f, dacc = 0.f;
        for (int c = 0; c < d; ++c) {
            sacc = fmaf(qs[c], to_f(kr[c]) * kscale, sacc);
            dacc = fmaf(dos[c], to_f(vr[c]), dacc);
        }
        ps[j] = sacc;
        ds[j] = dacc;
        lmax = fmaxf(lmax, sacc);
    }
    const float m = block_max(lmax, red);
    float lsum = 0.f;
    for (int j = threadIdx.x; j < S; j += blockDim.x) {
        const float e = expf(ps[j] - m);
        ps[j] = e;
        lsum += e;
    }
    const float inv = 1.0f / block_sum(lsum, red);
    float ldel = 0.f;
    for (int j = threadIdx.x; j < S; j += blockDim.x) {
        const float p = ps[j] * inv;
        ps[j] = p;
        ldel = fmaf(p, ds[j], ldel);
    }
    const float delta = block_sum(ldel, red);
    float* Prow = P + ((int64_t)bh * S + i) * S;
    float* dSrow = dS + ((int64_t)bh * S + i) * S;
    for (int j = threadIdx.x; j < S; j += blockDim.x) {
        const float v = ps[j] * (ds[j] - delta);
        ds[j] = v;
        Prow[j] = ps[j];
        dSrow[j] = v;
    }
    __syncthreads();
    for (int c = threadIdx.x; c < d; c += blockDim.x) {
        float acc = 0.f;
        for (int j = 0; j < S; ++j) acc = fmaf(ds[j], to_f(base[(int64_t)j * 3 * C + d + c]), acc);
        dqkv[hoff + (int64_t)i * 3 * C + c] = from_f<T>(acc * kscale);
    }
}
template <typename T>
__global__ void __launch_bounds__(256) attn_bwd_cols_kernel(const T* __restrict__ qkv, const T* __restrict__ dO, int S, int C, int heads, int N,
                                                            const float* __restrict__ P, const float* __restrict__ dS, T* __restrict__ dqkv) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int d = C / heads;
    float* pc = sm;       // S : P[:, j]
    float* dc = sm + S;   // S : dS[:, j]
    const int bh = blockIdx.y, n = bh / heads, hd = bh % heads, j = blockIdx.x;
    const int64_t hoff = (int64_t)n * S * 3 * C + (int64_t)hd * 3 * d;
    const T* base = qkv + hoff;
    const T* dob = dO + (int64_t)(bh % N) * S * C + (int64_t)(bh / N) * d;
    const float kscale = powf((float)C, -0.5f);
    for (int i = threadIdx.x; i < S; i += blockDim.x) {
        pc[i] = P[((int64_t)bh * S + i) * S + j];
        dc[i] = dS[((int64_t)bh * S + i) * S + j];
    }
    __syncthreads();
    for (int c = threadIdx.x; c < d; c += blockDim.x) {
        float dk = 0.f, dvv = 0.f;
        for (int i = 0; i < S; ++i) {
            dk = fmaf(dc[i], to_f(base[(int64_t)i * 3 * C + c]), dk);
            dvv = fmaf(pc[i], to_f(dob[(int64_t)i * C + c]), dvv);
        }
        dqkv[hoff + (int64_t)j * 3 * C + d + c] = from_f<T>(dk * kscale);
        dqkv[hoff + (int64_t)j * 3 * C + 2 * d + c] = from_f<T>(dvv);
    }
}
int launch_attn_heads_bwd(int dtype, const void* qkv, const void* dO, int N, int S, int C, int heads, float* P, float* dS, void* dqkv, hipStream_t s) {
    DMME_REQUIRE(heads >= 1 && C % heads == 0, DMME_ERR_INVALID, "attention backward: width %d not divisible by %d heads", C, heads);
    const int d = C / heads;
    const size_t ldsA = (size_t)(2 * d + 2 * S + 16) * sizeof(float), ldsB = (size_t)(2 * S) * sizeof(float);
    DMME_REQUIRE(ldsA <= 64 * 1024, DMME_ERR_UNSUPPORTED, "attention backward: d+S too large (%d+%d)", d, S);
    dim3 grid(S, N * heads);
    if (dtype == DMME_BF16) {
        hipLaunchKernelGGL(attn_bwd_rows_kernel<bf16>, grid, dim3(256), ldsA, s, (const bf16*)qkv, (const bf16*)dO, S, C, heads, N, P, dS, (bf16*)dqkv);
        DMME_CHECK_LAUNCH();
        hipLaunchKernelGGL(attn_bwd_cols_kernel<bf16>, grid, dim3(256), ldsB, s, (const bf16*)qkv, (const bf16*)dO, S, C, heads, N, P, dS, (bf16*)dqkv);
    } else if (dtype == DMME_F16) {
        hipLaunchKernelGGL(attn_bwd_rows_kernel<f16>, grid, dim3(256), ldsA, s, (const f16*)qkv, (const f16*)dO, S, C, heads, N, P, dS, (f16*)dqkv);
        DMME_CHECK_LAUNCH();
        hipLaunchKernelGGL(attn_bwd_cols_kernel<f16>, grid, dim3(256), ldsB, s, (const f16*)qkv, (const f16*)dO, S, C, heads, N, P, dS, (f16*)dqkv);
    } else {
        hipLaunchKernelGGL(attn_bwd_rows_kernel<float>, grid, dim3(256), ldsA, s, (const float*)qkv, (const float*)dO, S, C, heads, N, P, dS, (float*)dqkv);
        DMME_CHECK_LAUNCH();
        hipLaunchKernelGGL(attn_bwd_cols_kernel<float>, grid, dim3(256), ldsB, s, (const float*)qkv, (const float*)dO, S, C, heads, N, P, dS, (float*)dqkv);
    }
    DMME_CHECK_LAUNCH();
    return DMME_OK;
}
int launch_attn_bwd_generic(int dtype, const void* qkv, const void* dO, int N, int S, int C, float* P, float* dS, void* dqkv, hipStream_t s) {
    return launch_attn_heads_bwd(dtype, qkv, dO, N, S, C, 1, P, dS, dqkv, s);
}

// ------------------------------------------------------------------ small linears (time MLP) backward
// dX[r][k] = sum_o dY[r][o] W[o][k]      (fp32 activations, weights in T)
template <typename T>
__global__ void __launch_bounds__(256) lin_dinput_kernel(const float* __restrict__ dY, const T* __restrict__ W, int R, int O, int K, float* __restrict__ dX) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= R * K) return;
    const int r = idx / K, k = idx % K;
    float acc = 0.f;
    for (int o = 0; o < O; ++o) acc = fmaf(dY[(int64_t)r * O + o], to_f(W[(int64_t)o * K + k]), acc);
    dX[idx] = acc;
}
// dW[o][k] += sum_r dY[r][o] X[r][k];  dB[o] += sum_r dY[r][o]
__global__ void __launch_bounds__(256) lin_dweight_kernel(const float* __restrict__ dY, const float* __restrict__ X, int R, int O, int K,
                                                          float* __restrict__ dW, float* __restrict__ dB) {
    const int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (idx >= (int64_t)O * K) return;
    const int o = (int)(idx / K), k = (int)(idx % K);
    float acc = 0.f, b = 0.f;
    for (int r = 0; r < R; ++r) {
        const float d = dY[(int64_t)r * O + o];
        acc = fmaf(d, X[(int64_t)r * K + k], acc);
        b += d;
    }
    dW[idx] += acc;
    if (k == 0 && dB) dB[o] += b;
}
// dz = dy * silu'(z), in place on dy
__global__ void __launch_bounds__(256) silu_bwd_kernel(float* __restrict__ dy, const float* __restrict__ z, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dy[i] *= silu_grad(z[i]);
}
int launch_lin_dinput(int dtype, const float* dY, const void* W, int R, int O, int K, float* dX, hipStream_t s) {
    const int total = R * K;
    if (dtype == DMME_BF16)
        hipLaunchKernelGGL(lin_dinput_kernel<bf16>, dim3((total + 255) / 256), dim3(256), 0, s, dY, (const bf16*)W, R, O, K, dX);
    else if (dtype == DMME_F16)
        hipLaunchKernelGGL(lin_dinput_kernel<f16>, dim3((total + 255) / 256), dim3(256), 0, s, dY, (const f16*)W, R, O, K, dX);
    else
        hipLaunchKernelGGL(lin_dinput_kernel<float>, dim3((total + 255) / 256), dim3(256), 0, s, dY, (const float*)W, R, O, K, dX);
    DMME_CHECK_LAUNCH();
    return DMME_OK;
}
int launch_lin_dweight(const float* dY, const float* X, int R, int O, int K, float* dW, float* dB, hipStream_t s) {
    const int64_t total = (int64_t)O * K;
    hipLaunchKernelGGL(lin_dweight_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, dY, X, R, O, K, dW, dB);
    DMME_CHECK_LAUNCH();
    return DMME_OK;
}
int launch_silu_bwd(float* dy, const float* z, int n, hipStream_t s) {
    hipLaunchKernelGGL(silu_bwd_kernel, dim3((n + 255) / 256), dim3(256), 0, s, dy, z, n);
    DMME_CHECK_LAUNCH();
    return DMME_OK;
}

// ------------------------------------------------------------------ optimiser tail
// sum of squares of a flat fp32 buffer -> partial[blockIdx] (finalised by the consumer)
__global__ void __launch_bounds__(256) sumsq_partial_kernel(const float* __restrict__ g, int64_t n, float* __restrict__ partial) {
    __shared__ float red[16];
    float acc = 0.f;
    for (int64_t i = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) * 4; i < n; i += (int64_t)gridDim.x * blockDim.x * 4) {
        if (i + 3 < n) {
            const float4 v = *reinterpret_cast<const float4*>(g + i);
            acc += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
        } else {
            for (int64_t j = i; j < n; ++j) acc += g[j] * g[j];
        }
    }
    const float tot = block_sum(acc, red);
    if (threadIdx.x == 0) partial[blockIdx.x] = tot;
}
__global__ void __launch_bounds__(256) sumsq_final_kernel(const float* __restrict__ partial, int n, float* __restrict__ out) {
    __shared__ float red[16];
    float acc = 0.f;
    for (int i = threadIdx.x; i < n; i += blockDim.x) acc += partial[i];
    const float tot = block_sum(acc, red);
    if (threadIdx.x == 0) out[0] = sqrtf(tot);
}
int launch_grad_norm(const float* g, int64_t n, float* norm_out, float* scratch, hipStream_t s) {
    hipLaunchKernelGGL(sumsq_partial_kernel, dim3(1024), dim3(256), 0, s, g, n, scratch);
    DMME_CHECK_LAUNCH();
    hipLaunchKernelGGL(sumsq_final_kernel, dim3(1), dim3(256), 0, s, scratch, 1024, norm_out);
    DMME_CHECK_LAUNCH();
    return DMME_OK;
}

// clip-by-global-norm + Adam (+ EMA) over flat fp32 buffers, one pass:
//   g *= grad_scale (1 / world when the exchange left SUMS in the buffer: the mean's divide rides on this pass)
//   g *= min(1, max_norm / (norm + 1e-6))                       (torch.nn.utils.clip_grad_norm_; norm = grad_scale * ||buffer||)
//   m = b1 m + (1-b1) g ; v = b2 v + (1-b2) g^2
//   p -= lr/bc1 * m / (sqrt(v)/sqrt(bc2) + eps)                 (torch.optim.Adam, no weight decay)
//   ema = d ema + (1-d) p                                        (callbacks/ema.py:169-176)
__global__ void __launch_bounds__(256) adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, float* __restrict__ ema, int64_t n, float lr, float b1, float b2,
                                                   float eps, float bc1, float bc2_sqrt, const float* __restrict__ norm, float max_norm,
                                                   float ema_decay, float grad_scale) {
    float clip = grad_scale;
    if (norm && max_norm > 0.f) {
        const float c = max_norm / (norm[0] * grad_scale + 1e-6f);
        clip = c < 1.0f ? c * grad_scale : grad_scale;
    }
    const float step = lr / bc1;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float gi = g[i] * clip;
        const float mi = b1 * m[i] + (1.0f - b1) * gi;
        const float vi = b2 * v[i] + (1.0f - b2) * gi * gi;
        m[i] = mi;
        v[i] = vi;
        const float pi = p[i] - step * mi / (sqrtf(vi) / bc2_sqrt + eps);
        p[i] = pi;
        if (ema) ema[i] = ema_decay * ema[i] + (1.0f - ema_decay) * pi;
    }
}
int launch_adam(float* p, const float* g, float* m, float* v, float* ema, int64_t n, float lr, float b1, float b2, float eps, int step,
                const float* norm, float max_norm, float ema_decay, float grad_scale, hipStream_t s) {
    const float bc1 = 1.0f - powf(b1, (float)step), bc2s = sqrtf(1.0f - powf(b2, (float)step));
    int64_t blocks = (n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(adam_kernel, dim3((unsigned)blocks), dim3(256), 0, s, p, g, m, v, ema, n, lr, b1, b2, eps, bc1, bc2s, norm, max_norm, ema_decay, grad_scale);
    DMME_CHECK_LAUNCH();
    return DMME_OK;
}

// ---- dynamic loss scaling of a half-precision training step, device-resident (torch.cuda.amp.GradScaler semantics) ----
// amp[8] floats: [0] scale S, [1] growth tracker, [2] optimiser steps taken, [3] found_inf of the last step, [4] steps skipped.
// The loss gradient is multiplied by S before backward (amp_scale_kernel); the fused optimiser pass divides it out again, clips the
// UNSCALED norm, and leaves parameters and moments untouched when the (scaled) gradient norm is not finite; amp_update_kernel then
// halves S / resets the tracker, or counts the step and doubles S after `interval` finite steps in a row.  Nothing is read back.
__global__ void amp_init_kernel(float* __restrict__ amp, float init_scale) {
    if (threadIdx.x < 8) amp[threadIdx.x] = threadIdx.x == 0 ? init_scale : 0.f;
}
__global__ void __launch_bounds__(256) amp_scale_kernel(float* __restrict__ d, int64_t n, const float* __restrict__ amp) {
    const float S = amp[0];
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) d[i] *= S;
}
__global__ void __launch_bounds__(256) adam_amp_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                       float* __restrict__ v, float* __restrict__ ema, int64_t n, float lr, float b1, float b2,
                                                       float eps, const float* __restrict__ norm, float max_norm, float ema_decay,
                                                       float grad_scale, const float* __restrict__ amp) {
    const float nrm = norm[0];
    if (!(fabsf(nrm) <= 3.0e38f)) return;  // inf or NaN somewhere in the scaled gradient: this step is skipped (GradScaler.step)
    const float gs = grad_scale / amp[0];   // unscale (and the data-parallel mean's 1 / world)
    if (!(fabsf(gs) <= 3.0e38f)) return;    // a scale driven to 0 / a denormal by repeated back-offs: 0 x inf would write NaN - skipped too
    float clip = gs;
    if (max_norm > 0.f) {
        const float c = max_norm / (nrm * gs + 1e-6f);
        clip = c < 1.0f ? c * gs : gs;
    }
    const float t = amp[2] + 1.0f;  // optimiser steps incl. this one (skipped steps do not count: torch skips optimizer.step())
    const float bc1 = 1.0f - powf(b1, t), bc2_sqrt = sqrtf(1.0f - powf(b2, t));
    const float step = lr / bc1;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float gi = g[i] * clip;
        const float mi = b1 * m[i] + (1.0f - b1) * gi;
        const float vi = b2 * v[i] + (1.0f - b2) * gi * gi;
        m[i] = mi;
        v[i] = vi;
        const float pi = p[i] - step * mi / (sqrtf(vi) / bc2_sqrt + eps);
        p[i] = pi;
        if (ema) ema[i] = ema_decay * ema[i] + (1.0f - ema_decay) * pi;
    }
}
__global__ void amp_update_kernel(float* __restrict__ amp, const float* __restrict__ norm, float growth, float backoff, float interval) {
    if (threadIdx.x != 0) return;
    // (the same two tests as adam_amp_kernel: the step it skipped is the step counted as skipped; grad_scale > 0 is finite, so the
    //  unscale factor is finite exactly when 1 / S is)
    const bool bad = !(fabsf(norm[0]) <= 3.0e38f) || !(fabsf(1.0f / amp[0]) <= 3.0e38f);
    if (bad) {
        amp[0] = fmaxf(amp[0] * backoff, 6.103515625e-05f);  // floor 2^-14: S never reaches a denormal or 0
        if (!(amp[0] >= 6.103515625e-05f)) amp[0] = 6.103515625e-05f;  // (a NaN scale recovers as well)
        amp[1] = 0.f;
        amp[3] = 1.f;
        amp[4] += 1.f;
    } else {
        amp[2] += 1.f;
        amp[3] = 0.f;
        amp[1] += 1.f;
        if (amp[1] >= interval) {
            amp[0] *= growth;
            amp[1] = 0.f;
        }
    }
}
int launch_amp_init(float* amp, float init_scale, hipStream_t s) {
    hipLaunchKernelGGL(amp_init_kernel, dim3(1), dim3(64), 0, s, amp, init_scale);
    DMME_CHECK_LAUNCH();
    return DMME_OK;
}
int launch_amp_scale(float* d, int64_t n, const float* amp, hipStream_t s) {
    int64_t blocks = (n + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(amp_scale_kernel, dim3((unsigned)(blocks < 1 ? 1 : blocks)), dim3(256), 0, s, d, n, amp);
    DMME_CHECK_LAUNCH();
    return DMME_OK;
}
int launch_adam_amp(float* p, const float* g, float* m, float* v, float* ema, int64_t n, float lr, float b1, float b2, float eps, const float* norm,
                    float max_norm, float ema_decay, float grad_scale, float* amp, float growth, float backoff, int interval, hipStream_t s) {
    int64_t blocks = (n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(adam_amp_kernel, dim3((unsigned)blocks), dim3(256), 0, s, p, g, m, v, ema, n, lr, b1, b2, eps, norm, max_norm, ema_decay, grad_scale, amp);
    DMME_CHECK_LAUNCH();
    hipLaunchKernelGGL(amp_update_kernel, dim3(1), dim3(64), 0, s, amp, norm, growth, backoff, (float)interval);
    DMME_CHECK_LAUNCH();
    return DMME_OK;
}

// ---- gradient exchange with bf16 on the wire and fp32 accumulation (distributed.Bf16ShardExchange) ----
// pack: fp32 gradient slice -> bf16 send buffer (round-to-nearest-even), zero padded to a multiple of the world size
__global__ void __launch_bounds__(256) grad_pack_bf16_kernel(const float* __restrict__ g, int64_t n, bf16* __restrict__ dst, int64_t n_pad) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n_pad; i += (int64_t)gridDim.x * blockDim.x) dst[i] = (bf16)(i < n ? g[i] : 0.f);
}
// reduce: this rank's shard as received from every rank, [world][per] bf16 -> scale * sum in fp32 (fixed order: rank 0 first), rounded once
__global__ void __launch_bounds__(256) shard_reduce_bf16_kernel(const bf16* __restrict__ recv, int world, int64_t per, float scale, bf16* __restrict__ out) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < per; i += (int64_t)gridDim.x * blockDim.x) {
        float acc = 0.f;
        for (int j = 0; j < world; ++j) acc += (float)recv[(int64_t)j * per + i];
        out[i] = (bf16)(acc * scale);
    }
}
// unpack: gathered bf16 means -> the fp32 gradient slice every rank's optimiser reads (identical bits on every rank)
__global__ void __launch_bounds__(256) grad_unpack_bf16_kernel(const bf16* __restrict__ src, int64_t n, float* __restrict__ g) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) g[i] = (float)src[i];
}
static unsigned exch_blocks(int64_t n) {
    int64_t b = (n + 255) / 256;
    return (unsigned)(b > 4096 ? 4096 : b < 1 ? 1 : b);
}
int launch_grad_pack_bf16(const float* g, int64_t n, void* dst, int64_t n_pad, hipStream_t s) {
    hipLaunchKernelGGL(grad_pack_bf16_kernel, dim3(exch_blocks(n_pad)), dim3(256), 0, s, g, n, (bf16*)dst, n_pad);
    DMME_CHECK_LAUNCH();
    return DMME_OK;
}
int launch_shard_reduce_bf16(const void* recv, int world, int64_t per, float scale, void* out, hipStream_t s) {
    hipLaunchKernelGGL(shard_reduce_bf16_kernel, dim3(exch_blocks(per)), dim3(256), 0, s, (const bf16*)recv, world, per, scale, (bf16*)out);
    DMME_CHECK_LAUNCH();
    return DMME_OK;
}
int launch_grad_unpack_bf16(const void* src, int64_t n, float* g, hipStream_t s) {
    hipLaunchKernelGGL(grad_unpack_bf16_kernel, dim3(exch_blocks(n)), dim3(256), 0, s, (const bf16*)src, n, g);
    DMME_CHECK_LAUNCH();
    return DMME_OK;
}

}  // namespace dmme
