// Small fp32 GEMMs of the time MLP at training batch (nt = B rows): LDS-tiled 64x64 output tile,
// K step 16, 4x4 outputs per thread.  Three operand modes cover forward, input-gradient and
// weight-gradient of nn.Linear (models/ddpm.py:101-104, :211-217):
//   NT: C[m][n]  = act(sum_k A[m][k] * W[n][k] + bias[n])      A fp32, W in T   (forward)
//   NN: C[m][n]  =     sum_k A[m][k] * W[k][n]                  A fp32, W in T   (dX = dY W)
//   TN: C[m][n] +=     sum_k A[k][m] * B[k][n]                  A, B fp32        (dW += dY^T X)
#include "common.h"

namespace dmme {

enum { GEMM_NT = 0, GEMM_NN = 1, GEMM_TN = 2 };

template <typename T, int MODE>
__global__ void __launch_bounds__(256) small_gemm_kernel(const float* __restrict__ A, int lda, const void* __restrict__ Bv, int ldb, int M, int N,
                                                         int K, const float* __restrict__ bias, int out_silu, float* __restrict__ Cm, int ldc, int ksplit,
                                                         const int64_t* __restrict__ mtile_off) {
    __shared__ float As[16][64 + 4];
    __shared__ float Bs[16][64 + 4];
    const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
    const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
    float acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
    // split-K: blockIdx.z owns every ksplit-th K step and adds its partial product atomically (C zeroed by the caller)
    for (int k0 = blockIdx.z * 16; k0 < K; k0 += 16 * ksplit) {
        // stage A tile as As[k][m], B tile as Bs[k][n]
        for (int u = tid; u < 16 * 64; u += 256) {
            int kk, mm;
            float av = 0.f, bv = 0.f;
            if (MODE == GEMM_TN) {  // A[k][m]: consecutive threads -> consecutive m
                kk = u >> 6; mm = u & 63;
                if (k0 + kk < K && m0 + mm < M) av = A[(int64_t)(k0 + kk) * lda + m0 + mm];
            } else {                // A[m][k]: consecutive threads -> consecutive k
                mm = u >> 4; kk = u & 15;
                if (k0 + kk < K && m0 + mm < M) av = A[(int64_t)(m0 + mm) * lda + k0 + kk];
            }
            As[kk][mm] = av;
            int kb, nn;
            if (MODE == GEMM_NT) {  // W[n][k]
                nn = u >> 4; kb = u & 15;
                if (k0 + kb < K && n0 + nn < N) bv = to_f(((const T*)Bv)[(int64_t)(n0 + nn) * ldb + k0 + kb]);
            } else if (MODE == GEMM_NN) {  // W[k][n]
                kb = u >> 6; nn = u & 63;
                if (k0 + kb < K && n0 + nn < N) bv = to_f(((const T*)Bv)[(int64_t)(k0 + kb) * ldb + n0 + nn]);
            } else {                // B[k][n] fp32
                kb = u >> 6; nn = u & 63;
                if (k0 + kb < K && n0 + nn < N) bv = ((const float*)Bv)[(int64_t)(k0 + kb) * ldb + n0 + nn];
            }
            Bs[kb][nn] = bv;
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) {
            float a[4], b[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) a[i] = As[kk][ty * 4 + i];
#pragma unroll
            for (int j = 0; j < 4; ++j) b[j] = Bs[kk][tx * 4 + j];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(a[i], b[j], acc[i][j]);
        }
        __syncthreads();
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
                if (out_silu) v = silu_f(v);
                Cm[(int64_t)m * ldc + n] = v;
            }
        }
    }
}

int launch_small_gemm(int dtype, int mode, const float* A, int lda, const void* B, int ldb, int M, int N, int K, const float* bias, int out_silu,
                      float* C, int ldc, hipStream_t s) {
    int ksplit = 1;
    const int wgs = ((N + 63) / 64) * ((M + 63) / 64);
    if (mode == GEMM_NN && !bias && !out_silu && wgs < 128 && K >= 1024) {  // few output tiles, long K: split the reduction
        ksplit = 256 / wgs;
        if (ksplit > K / 64) ksplit = K / 64;
        if (ksplit < 1) ksplit = 1;
    }
    if (ksplit > 1) DMME_CHECK_HIP(hipMemsetAsync(C, 0, (size_t)M * ldc * sizeof(float), s));
    dim3 grid((N + 63) / 64, (M + 63) / 64, ksplit);
#define DMME_SG(TT, MM) hipLaunchKernelGGL((small_gemm_kernel<TT, MM>), grid, dim3(256), 0, s, A, lda, B, ldb, M, N, K, bias, out_silu, C, ldc, ksplit, (const int64_t*)nullptr)
    if (mode == GEMM_TN) {
        DMME_SG(float, GEMM_TN);
    } else if (dtype == DMME_BF16) {
        if (mode == GEMM_NT) DMME_SG(bf16, GEMM_NT); else DMME_SG(bf16, GEMM_NN);
    } else {
        if (mode == GEMM_NT) DMME_SG(float, GEMM_NT); else DMME_SG(float, GEMM_NN);
    }
#undef DMME_SG
    DMME_CHECK_LAUNCH();
    return DMME_OK;
}

// C[tile(m)][n] += sum_k A[k][m] * B[k][n] with a per-64-row-tile output base: the weight gradients of all the
// per-ResBlock time projections (models/ddpm.py:101-104) in one launch
int launch_small_gemm_tn_tiled(const float* A, int lda, const float* B, int ldb, int M, int N, int K, float* C, int ldc,
                               const int64_t* mtile_off, hipStream_t s) {
    dim3 grid((N + 63) / 64, (M + 63) / 64, 1);
    hipLaunchKernelGGL((small_gemm_kernel<float, GEMM_TN>), grid, dim3(256), 0, s, A, lda, (const void*)B, ldb, M, N, K, (const float*)nullptr, 0, C, ldc,
                       1, mtile_off);
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
