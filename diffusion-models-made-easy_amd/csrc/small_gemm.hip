// Small fp32 GEMMs of the time MLP at training batch (nt = B rows): LDS-tiled 64x64 output tile (double-buffered),
// K stage 32, 4x4 outputs per thread.  Three operand modes cover forward, input-gradient and
// weight-gradient of nn.Linear (models/ddpm.py:101-104, :211-217):
//   NT: C[m][n]  = act(sum_k A[m][k] * W[n][k] + bias[n])      A fp32, W in T   (forward)
//   NN: C[m][n]  =     sum_k A[m][k] * W[k][n]                  A fp32, W in T   (dX = dY W)
//   TN: C[m][n] +=     sum_k A[k][m] * B[k][n]                  A, B fp32        (dW += dY^T X)
#include "common.h"

namespace dmme {

enum { GEMM_NT = 0, GEMM_NN = 1, GEMM_TN = 2 };

// A stage = KS = 32 K values of a 64 x 64 output tile.  The next stage's operands are loaded into registers BEFORE the current
// stage's 512 FMAs per thread and written to the other LDS buffer after them (one barrier per stage): the first version loaded,
// synchronised, computed 16 K values and synchronised again - 32 dependent global round trips for K = 512, 57 us for 0.7 GFLOP.
template <typename T, int MODE>
__global__ void __launch_bounds__(256) small_gemm_kernel(const float* __restrict__ A, int lda, const void* __restrict__ Bv, int ldb, int M, int N,
                                                         int K, const float* __restrict__ bias, int out_silu, float* __restrict__ Cm, int ldc, int ksplit,
                                                         const int64_t* __restrict__ mtile_off, float* __restrict__ Cpre) {
    constexpr int KS = 32, UP = KS * 64 / 256;  // elements of each operand tile per thread and stage
    __shared__ float As[2][KS][64 + 4];
    __shared__ float Bs[2][KS][64 + 4];
    const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
    const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
    float acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
    float ra[UP], rb[UP];
    // element u of a stage's tiles: A as As[k][m], B as Bs[k][n]
    auto gload = [&](int k0) __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < UP; ++q) {
            const int u = tid + 256 * q;
            int kk, mm;
            float av = 0.f, bv = 0.f;
            if (MODE == GEMM_TN) {  // A[k][m]: consecutive threads -> consecutive m
                kk = u >> 6; mm = u & 63;
                if (k0 + kk < K && m0 + mm < M) av = A[(int64_t)(k0 + kk) * lda + m0 + mm];
            } else {                // A[m][k]: consecutive threads -> consecutive k
                mm = u / KS; kk = u % KS;
                if (k0 + kk < K && m0 + mm < M) av = A[(int64_t)(m0 + mm) * lda + k0 + kk];
            }
            ra[q] = av;
            int kb, nn;
            if (MODE == GEMM_NT) {  // W[n][k]
                nn = u / KS; kb = u % KS;
                if (k0 + kb < K && n0 + nn < N) bv = to_f(((const T*)Bv)[(int64_t)(n0 + nn) * ldb + k0 + kb]);
            } else if (MODE == GEMM_NN) {  // W[k][n]
                kb = u >> 6; nn = u & 63;
                if (k0 + kb < K && n0 + nn < N) bv = to_f(((const T*)Bv)[(int64_t)(k0 + kb) * ldb + n0 + nn]);
            } else {                // B[k][n] fp32
                kb = u >> 6; nn = u & 63;
                if (k0 + kb < K && n0 + nn < N) bv = ((const float*)Bv)[(int64_t)(k0 + kb) * ldb + n0 + nn];
            }
            rb[q] = bv;
        }
    };
    auto sstore = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < UP; ++q) {
            const int u = tid + 256 * q;
            if (MODE == GEMM_TN) As[buf][u >> 6][u & 63] = ra[q]; else As[buf][u % KS][u / KS] = ra[q];
            if (MODE == GEMM_NT) Bs[buf][u % KS][u / KS] = rb[q]; else Bs[buf][u >> 6][u & 63] = rb[q];
        }
    };
    // split-K: blockIdx.z owns every ksplit-th K stage and adds its partial product atomically (C zeroed by the caller)
    const int kstep = KS * ksplit;
    int k0 = blockIdx.z * KS, buf = 0;
    if (k0 < K) {
        gload(k0);
        sstore(0);
    }
    __syncthreads();
    for (; k0 < K; k0 += kstep) {
        const bool more = k0 + kstep < K;
        if (more) gload(k0 + kstep);
#pragma unroll
        for (int kk = 0; kk < KS; ++kk) {
            float a[4], b[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) a[i] = As[buf][kk][ty * 4 + i];
#pragma unroll
            for (int j = 0; j < 4; ++j) b[j] = Bs[buf][kk][tx * 4 + j];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(a[i], b[j], acc[i][j]);
        }
        if (more) sstore(buf ^ 1);
        __syncthreads();
        buf ^= 1;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = m0 + ty * 4 + i;
        if (m >= M) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = n0 + tx * 4 + j;
            if (n >= N) continue;
            float v = acc[i][j];
            if (MODE == GEMM_TN) {
                // scattered output: each 64-row tile of C has its own base (several Linear layers in one launch)
                float* crow = mtile_off ? Cm + mtile_off[blockIdx.y] + (int64_t)(ty * 4 + i) * ldc : Cm + (int64_t)m * ldc;
                crow[n] += v;
            } else if (ksplit > 1) {
                atomicAdd(&Cm[(int64_t)m * ldc + n], v);
            } else {
                if (bias) v += bias[n];
                if (Cpre) Cpre[(int64_t)m * ldc + n] = v;  // the pre-activation, kept for the backward pass (training)
                if (out_silu) v = silu_f(v);
                Cm[(int64_t)m * ldc + n] = v;
            }
        }
    }
}

// The same three GEMMs on the fp32 matrix cores, with NO LDS: a wave owns one 32 x 32 output tile and feeds v_mfma_f32_32x32x2_f32
// straight from global memory.  That MFMA takes K = 2 per instruction (lane half h supplies k = h), and the order of a sum's terms is
// free: per 8 K values a lane loads ITS four (k = 8 i + 4 h + e, e = 0..3: one 16-byte load where K is the contiguous dimension) and
// four MFMAs consume element e of both operands.  Same fp32 FMA arithmetic as the LDS-tiled VALU kernel above, one instruction where
// that kernel issues 32 FMAs and 8 LDS reads per lane: its 64 x 64 tiles left 16 workgroups with 12 k instructions per thread for the
// 512 x 512 Linear (74 us); here 64 waves run 256 MFMAs each.
typedef float f32x16_sg __attribute__((ext_vector_type(16)));
template <typename T, int MODE>
__global__ void __launch_bounds__(256) small_gemm_mfma_kernel(const float* __restrict__ A, int lda, const void* __restrict__ Bv, int ldb, int M, int N,
                                                              int K, const float* __restrict__ bias, int out_silu, float* __restrict__ Cm, int ldc,
                                                              int ksplit, const int64_t* __restrict__ mtile_off, float* __restrict__ Cpre, int wsplit) {
    // wsplit: the workgroup's four waves share ONE output tile and split its K range (summed through LDS in wave order) instead of
    // owning four tiles: these GEMMs are chains of round trips (a block of 64 K values per trip, one wave per SIMD, most CUs idle) -
    // a quarter of the chain per wave on four times the workgroups.  Same fp32 products; the order of a sum's terms is fixed.
    __shared__ float wred[3][16][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 31, h = lane >> 5;
    const int n0 = wsplit ? blockIdx.x * 32 : (blockIdx.x * 4 + wave) * 32, m0 = blockIdx.y * 32;
    if (n0 >= N) return;  // (wsplit: uniform over the workgroup)
    const int m = m0 + r, n = n0 + r;  // this lane's row of the A operand / column of the B operand
    const bool mok = m < M, nok = n < N;
    f32x16_sg acc;
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[j] = 0.f;
    // K slice of this workgroup (split-K over blockIdx.z in units of 8)
    const int k8 = (K + 7) / 8, per = (k8 + ksplit - 1) / ksplit;
    int kb = (int)blockIdx.z * per * 8, ke = min(K, kb + per * 8);
    if (wsplit) {
        const int per4 = ((ke - kb + 7) / 8 + 3) / 4 * 8;
        kb += wave * per4;
        ke = min(ke, kb + per4);
    }
    auto load = [&](int k0, float (&a)[4], float (&b)[4]) __attribute__((always_inline)) {
        const int k = k0 + 4 * h;
#pragma unroll
        for (int e = 0; e < 4; ++e) a[e] = b[e] = 0.f;
        if (MODE == GEMM_TN) {  // A[k][m], B[k][n] fp32
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (k + e < ke) {
                    if (mok) a[e] = A[(int64_t)(k + e) * lda + m];
                    if (nok) b[e] = ((const float*)Bv)[(int64_t)(k + e) * ldb + n];
                }
        } else {
            if (mok) {  // A[m][k]
                if (k + 3 < ke && (lda & 3) == 0) {
                    const float4 v = *reinterpret_cast<const float4*>(A + (int64_t)m * lda + k);
                    a[0] = v.x; a[1] = v.y; a[2] = v.z; a[3] = v.w;
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (k + e < ke) a[e] = A[(int64_t)m * lda + k + e];
                }
            }
            if (nok) {
                if (MODE == GEMM_NT) {  // W[n][k]
                    const T* wp = (const T*)Bv + (int64_t)n * ldb + k;
                    if (k + 3 < ke && (ldb & 3) == 0) {
                        if constexpr (__is_same(T, bf16)) {
                            const uint2 v = *reinterpret_cast<const uint2*>(wp);  // four bf16
                            b[0] = __uint_as_float(v.x << 16); b[1] = __uint_as_float(v.x & 0xffff0000u);
                            b[2] = __uint_as_float(v.y << 16); b[3] = __uint_as_float(v.y & 0xffff0000u);
                        } else if constexpr (sizeof(T) == 2) {
                            typedef T tx4 __attribute__((ext_vector_type(4)));
                            const tx4 v = *reinterpret_cast<const tx4*>(wp);  // four IEEE halves
                            b[0] = (float)v[0]; b[1] = (float)v[1]; b[2] = (float)v[2]; b[3] = (float)v[3];
                        } else {
                            const float4 v = *reinterpret_cast<const float4*>(wp);
                            b[0] = v.x; b[1] = v.y; b[2] = v.z; b[3] = v.w;
                        }
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if (k + e < ke) b[e] = to_f(wp[e]);
                    }
                } else {  // W[k][n]
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (k + e < ke) b[e] = to_f(((const T*)Bv)[(int64_t)(k + e) * ldb + n]);
                }
            }
        }
    };
    // blocks of 8 steps (64 K values): the next block's 16+ loads are in flight under the current block's 32 MFMAs (2048 cycles - about
    // one L2 round trip; one step ahead was not enough: 50 us for K = 512, i.e. a round trip per step)
    float a0[8][4], b0[8][4], a1[8][4], b1[8][4];
#define SG_LOAD_BLOCK(K0, AA, BB)                                              \
    {                                                                          \
        _Pragma("unroll") for (int st = 0; st < 8; ++st) load((K0) + 8 * st, AA[st], BB[st]); \
    }
#define SG_MMA_BLOCK(AA, BB)                                                   \
    {                                                                          \
        _Pragma("unroll") for (int st = 0; st < 8; ++st)                       \
            _Pragma("unroll") for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(AA[st][e], BB[st][e], acc, 0, 0, 0); \
    }
    int k0 = kb;
    if (k0 < ke) SG_LOAD_BLOCK(k0, a0, b0)
    for (; k0 < ke; k0 += 128) {
        if (k0 + 64 < ke) SG_LOAD_BLOCK(k0 + 64, a1, b1)
        SG_MMA_BLOCK(a0, b0)
        if (k0 + 128 < ke) SG_LOAD_BLOCK(k0 + 128, a0, b0)
        if (k0 + 64 < ke) SG_MMA_BLOCK(a1, b1)
    }
#undef SG_LOAD_BLOCK
#undef SG_MMA_BLOCK
    if (wsplit) {
        if (wave > 0) {
#pragma unroll
            for (int j = 0; j < 16; ++j) wred[wave - 1][j][lane] = acc[j];
        }
        __syncthreads();
        if (wave > 0) return;
#pragma unroll
        for (int w = 0; w < 3; ++w)
#pragma unroll
            for (int j = 0; j < 16; ++j) acc[j] += wred[w][j][lane];
    }
    if (!nok) return;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const int mr = (j & 3) + 8 * (j >> 2) + 4 * h, mm = m0 + mr;  // D: lane = column n, registers = rows
        if (mm >= M) continue;
        float v = acc[j];
        if (MODE == GEMM_TN) {
            float* crow = mtile_off ? Cm + mtile_off[mm >> 6] + (int64_t)(mm & 63) * ldc : Cm + (int64_t)mm * ldc;
            if (ksplit > 1) atomicAdd(&crow[n], v); else crow[n] += v;
        } else if (ksplit > 1) {
            atomicAdd(&Cm[(int64_t)mm * ldc + n], v);
        } else {
            if (bias) v += bias[n];
            if (Cpre) Cpre[(int64_t)mm * ldc + n] = v;
            if (out_silu) v = silu_f(v);
            Cm[(int64_t)mm * ldc + n] = v;
        }
    }
}

static bool sg_mfma() {
    const bool off = (debug_route("no_small_gemm_mfma") != 0);
    return !off;
}

int launch_small_gemm(int dtype, int mode, const float* A, int lda, const void* B, int ldb, int M, int N, int K, const float* bias, int out_silu,
                      float* C, int ldc, hipStream_t s, float* Cpre) {
    if (sg_mfma()) {
        // few waves and a long K with a linear epilogue: split the reduction (atomics into a zeroed C)
        int ksplit = 1;
        const int waves = ((N + 31) / 32) * ((M + 31) / 32);
        if (mode != GEMM_TN && !bias && !out_silu && waves < 512 && K >= 1024) {
            constexpr int cap = 8;
            ksplit = 1024 / waves;
            if (ksplit > cap) ksplit = cap;  // every split is one float atomic per output element
            if (ksplit > K / 64) ksplit = K / 64;
            if (ksplit < 1) ksplit = 1;
        }
        if (ksplit > 1) DMME_CHECK_HIP(hipMemsetAsync(C, 0, (size_t)M * ldc * sizeof(float), s));
        // four waves per tile where the tiles alone leave SIMDs idle and every wave still gets a block of K
        const int wsplit = (waves * ksplit <= 1024 && K / ksplit >= 256) ? 1 : 0;
        dim3 grid(wsplit ? (N + 31) / 32 : (N + 127) / 128, (M + 31) / 32, ksplit);
#define DMME_SGM(TT, MM) hipLaunchKernelGGL((small_gemm_mfma_kernel<TT, MM>), grid, dim3(256), 0, s, A, lda, B, ldb, M, N, K, bias, out_silu, C, ldc, ksplit, (const int64_t*)nullptr, Cpre, wsplit)
        if (mode == GEMM_TN) {
            DMME_SGM(float, GEMM_TN);
        } else if (dtype == DMME_BF16) {
            if (mode == GEMM_NT) DMME_SGM(bf16, GEMM_NT); else DMME_SGM(bf16, GEMM_NN);
        } else if (dtype == DMME_F16) {
            if (mode == GEMM_NT) DMME_SGM(f16, GEMM_NT); else DMME_SGM(f16, GEMM_NN);
        } else {
            if (mode == GEMM_NT) DMME_SGM(float, GEMM_NT); else DMME_SGM(float, GEMM_NN);
        }
#undef DMME_SGM
        DMME_CHECK_LAUNCH();
        return DMME_OK;
    }
    int ksplit = 1;
    const int wgs = ((N + 63) / 64) * ((M + 63) / 64);
    if (mode == GEMM_NN && !bias && !out_silu && wgs < 128 && K >= 1024) {  // few output tiles, long K: split the reduction
        ksplit = 256 / wgs;
        if (ksplit > K / 64) ksplit = K / 64;  // (>= two 32-wide stages per slice)
        if (ksplit < 1) ksplit = 1;
    }
    if (ksplit > 1) DMME_CHECK_HIP(hipMemsetAsync(C, 0, (size_t)M * ldc * sizeof(float), s));
    dim3 grid((N + 63) / 64, (M + 63) / 64, ksplit);
#define DMME_SG(TT, MM) hipLaunchKernelGGL((small_gemm_kernel<TT, MM>), grid, dim3(256), 0, s, A, lda, B, ldb, M, N, K, bias, out_silu, C, ldc, ksplit, (const int64_t*)nullptr, Cpre)
    if (mode == GEMM_TN) {
        DMME_SG(float, GEMM_TN);
    } else if (dtype == DMME_BF16) {
        if (mode == GEMM_NT) DMME_SG(bf16, GEMM_NT); else DMME_SG(bf16, GEMM_NN);
    } else if (dtype == DMME_F16) {
        if (mode == GEMM_NT) DMME_SG(f16, GEMM_NT); else DMME_SG(f16, GEMM_NN);
    } else {
        if (mode == GEMM_NT) DMME_SG(float, GEMM_NT); else DMME_SG(float, GEMM_NN);
    }
#undef DMME_SG
    DMME_CHECK_LAUNCH();
    return DMME_OK;
}

// dst[c][r] = src[r][c] (weights of a Linear, for the input-gradient GEMM: its K then runs along the contiguous dimension of both
// operands - the NN form reads W[k][n] with 2-byte loads a row apart and is 2.5 x slower than NT on the same shape)
template <typename T>
__global__ void __launch_bounds__(256) transpose_kernel(const T* __restrict__ src, int R, int Ccols, T* __restrict__ dst) {
    __shared__ T tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
    for (int i = ty; i < 32; i += 8)
        if (r0 + i < R && c0 + tx < Ccols) tile[i][tx] = src[(int64_t)(r0 + i) * Ccols + c0 + tx];
    __syncthreads();
    for (int i = ty; i < 32; i += 8)
        if (c0 + i < Ccols && r0 + tx < R) dst[(int64_t)(c0 + i) * R + r0 + tx] = tile[tx][i];
}
int launch_transpose(int dtype, const void* src, int R, int Ccols, void* dst, hipStream_t s) {
    const dim3 grid((Ccols + 31) / 32, (R + 31) / 32);
    if (is16(dtype))  // (a 16-bit transpose moves bits: one instantiation serves bf16 and IEEE half)
        hipLaunchKernelGGL(transpose_kernel<bf16>, grid, dim3(256), 0, s, (const bf16*)src, R, Ccols, (bf16*)dst);
    else
        hipLaunchKernelGGL(transpose_kernel<float>, grid, dim3(256), 0, s, (const float*)src, R, Ccols, (float*)dst);
    DMME_CHECK_LAUNCH();
    return DMME_OK;
}

// C[tile(m)][n] += sum_k A[k][m] * B[k][n] with a per-64-row-tile output base: the weight gradients of all the
// per-ResBlock time projections (models/ddpm.py:101-104) in one launch
int launch_small_gemm_tn_tiled(const float* A, int lda, const float* B, int ldb, int M, int N, int K, float* C, int ldc,
                               const int64_t* mtile_off, hipStream_t s) {
    if (sg_mfma()) {
        dim3 gridm((N + 127) / 128, (M + 31) / 32, 1);
        hipLaunchKernelGGL((small_gemm_mfma_kernel<float, GEMM_TN>), gridm, dim3(256), 0, s, A, lda, (const void*)B, ldb, M, N, K, (const float*)nullptr, 0, C,
                           ldc, 1, mtile_off, (float*)nullptr, 0);
        DMME_CHECK_LAUNCH();
        return DMME_OK;
    }
    dim3 grid((N + 63) / 64, (M + 63) / 64, 1);
    hipLaunchKernelGGL((small_gemm_kernel<float, GEMM_TN>), grid, dim3(256), 0, s, A, lda, (const void*)B, ldb, M, N, K, (const float*)nullptr, 0, C, ldc,
                       1, mtile_off, (float*)nullptr);
    DMME_CHECK_LAUNCH();
    return DMME_OK;
}

// out[c] += sum_n M[n*stride + c*estride]   (bias-type reductions over the batch), one workgroup per 32 columns
__global__ void __launch_bounds__(256) nsum_kernel(const float* __restrict__ Mx, int N, int C, int64_t stride, int estride, float* __restrict__ out,
                                                   const int64_t* __restrict__ ctile_off) {
    __shared__ float red[8][33];
    const int cl = threadIdx.x & 31, seg = threadIdx.x >> 5;
    const int c = blockIdx.x * 32 + cl;
    float acc = 0.f;
    if (c < C)
        for (int n = seg; n < N; n += 8) acc += Mx[(int64_t)n * stride + (int64_t)c * estride];
    red[seg][cl] = acc;
    __syncthreads();
    if (seg == 0 && c < C) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) t += red[k][cl];
        if (ctile_off) out[ctile_off[blockIdx.x] + cl] += t;  // per-32-column destination (several bias vectors in one launch)
        else out[c] += t;
    }
}
int launch_nsum(const float* Mx, int N, int C, int64_t stride, int estride, float* out, hipStream_t s) {
    hipLaunchKernelGGL(nsum_kernel, dim3((C + 31) / 32), dim3(256), 0, s, Mx, N, C, stride, estride, out, (const int64_t*)nullptr);
    DMME_CHECK_LAUNCH();
    return DMME_OK;
}

int launch_nsum_tiled(const float* Mx, int N, int C, int64_t stride, int estride, float* out, const int64_t* ctile_off, hipStream_t s) {
    hipLaunchKernelGGL(nsum_kernel, dim3((C + 31) / 32), dim3(256), 0, s, Mx, N, C, stride, estride, out, ctile_off);
    DMME_CHECK_LAUNCH();
    return DMME_OK;
}

}  // namespace dmme
