// Coalesced, HBM-bound backward helpers for NHWC tensors (16-byte loads, every element read
// once per pass): per-(image, channel) column sums of dY (bias / time-embedding gradients) and
// the GroupNorm(+SiLU+Dropout2d) backward split into
//   pass A  per-(n, c) sums  A = sum_p du,  B = sum_p du*xhat            (du = dv*mask*silu'(u))
//   finalize S1[n,g] = sum_{c in g} gamma_c A,  S2 = sum gamma_c B;  dgamma_c += sum_n B, dbeta_c += sum_n A
//   pass B  dx = rstd * (du*gamma - (S1 + xhat*S2)/cnt)                  (elementwise)
// The group sums follow from the channel sums, so no per-group reduction over pixels is needed.
#include <stdlib.h>

#include "conv_common.h"

namespace dmme {


template <typename T>
__device__ __forceinline__ void load_vec(const T* p, float (&v)[16 / sizeof(T)]) {
    const uint4 raw = *reinterpret_cast<const uint4*>(p);
    if constexpr (sizeof(T) == 4) {
        const float4 f = __builtin_bit_cast(float4, raw);
        v[0] = f.x; v[1] = f.y; v[2] = f.z; v[3] = f.w;
    } else {
        const typename Vec8<T>::type b = __builtin_bit_cast(typename Vec8<T>::type, raw);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = (float)b[j];
    }
}
// The pixel loops below issue the 16-byte loads of FOUR pixels before touching any of them: at one or two loads in flight per
// thread these kernels ran at ~2.3 TB/s (bandwidth = bytes in flight / memory latency), far under what HBM delivers.  The
// accumulation order per thread is unchanged (pixels in ascending order), so the results are the same bit for bit.
template <typename T>
__device__ __forceinline__ uint4 load_raw(const T* p) {
    return *reinterpret_cast<const uint4*>(p);
}
template <typename T>
__device__ __forceinline__ void unpack_vec(const uint4& raw, float (&v)[16 / sizeof(T)]) {
    if constexpr (sizeof(T) == 4) {
        const float4 f = __builtin_bit_cast(float4, raw);
        v[0] = f.x; v[1] = f.y; v[2] = f.z; v[3] = f.w;
    } else {
        const typename Vec8<T>::type b = __builtin_bit_cast(typename Vec8<T>::type, raw);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = (float)b[j];
    }
}
template <typename T>
__device__ __forceinline__ void store_vec(T* p, const float (&v)[16 / sizeof(T)]) {
    if constexpr (sizeof(T) == 4) {
        *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
    } else {
        typename Vec8<T>::type b;
#pragma unroll
        for (int j = 0; j < 8; ++j) b[j] = (T)v[j];
        *reinterpret_cast<typename Vec8<T>::type*>(p) = b;
    }
}

template <typename T>
__device__ __forceinline__ float silu_grad_f(float u) {
    float s;
    if constexpr (sizeof(T) == 2)
        s = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * u));  // bf16 tensors: hardware exp / rcp
    else
        s = 1.0f / (1.0f + expf(-u));
    return s * (1.0f + u * (1.0f - s));
}

// geometry shared by the kernels: a workgroup owns `chunk_px` pixels of one image, thread -> fixed
// 16-byte channel slot (tid % VPP) and pixel row phase (tid / VPP)
static bool vec_geometry(int dtype, int HW, int C, int& chunk_px, int& nchunks, int& ppw_out) {
    const int EPV = is16(dtype) ? 8 : 4;
    if (C % EPV) return false;
    const int VPP = C / EPV;
    if (VPP > 256) return false;
    int ppw = 256 / VPP;  // threads beyond ppw*VPP idle when VPP does not divide 256 (e.g. 768 channels)
    while (ppw > 1 && HW % ppw) --ppw;
    // 16 pixel rows per thread (four batches of four loads in flight).  Measured (training step, batch 128): 16 rows 11.96 ms, 8 rows
    // 12.27, 4 rows 13.3 (4x the workgroups, but 4x the per-chunk partial rows for the finalize kernels to sum), 32 rows 11.99, 64 rows 12.22
    int sweeps = HW / ppw;
    constexpr int max_sweeps = 16;
    if (sweeps > max_sweeps) sweeps = max_sweeps;
    while (sweeps > 1 && (HW / ppw) % sweeps) --sweeps;
    chunk_px = sweeps * ppw;
    nchunks = HW / chunk_px;
    ppw_out = ppw;
    return true;
}

// ------------------------------------------------------------------ column sums
template <typename T>
__global__ void __launch_bounds__(256) colsum_vec_kernel(const T* __restrict__ dY, int HW, int C, int chunk_px, int ppw, float* __restrict__ rowsum) {
    constexpr int EPV = 16 / sizeof(T);
    __shared__ float red[256 * EPV];
    const int tid = threadIdx.x, VPP = C / EPV, slot = tid % VPP, prow = tid / VPP;
    const int n = blockIdx.y;
    const int64_t p0 = (int64_t)n * HW + (int64_t)blockIdx.x * chunk_px;
    float acc[EPV];
#pragma unroll
    for (int j = 0; j < EPV; ++j) acc[j] = 0.f;
#define COL_ONE(RAW)                                                   \
    {                                                                  \
        float v[EPV];                                                  \
        unpack_vec<T>(RAW, v);                                         \
        _Pragma("unroll") for (int j = 0; j < EPV; ++j) acc[j] += v[j]; \
    }
    if (prow < ppw) {
        int p = prow;
        for (; p + 3 * ppw < chunk_px; p += 4 * ppw) {
            const T* q = dY + (p0 + p) * C + slot * EPV;
            const uint4 r0 = load_raw<T>(q), r1 = load_raw<T>(q + (int64_t)ppw * C), r2 = load_raw<T>(q + (int64_t)2 * ppw * C),
                        r3 = load_raw<T>(q + (int64_t)3 * ppw * C);
            COL_ONE(r0) COL_ONE(r1) COL_ONE(r2) COL_ONE(r3)
        }
        for (; p < chunk_px; p += ppw) {
            const uint4 r0 = load_raw<T>(dY + (p0 + p) * C + slot * EPV);
            COL_ONE(r0)
        }
    }
#undef COL_ONE
#pragma unroll
    for (int j = 0; j < EPV; ++j) red[tid * EPV + j] = acc[j];
    __syncthreads();
    if (tid < VPP) {
#pragma unroll
        for (int j = 0; j < EPV; ++j) {
            float s = 0.f;
            for (int r = 0; r < ppw; ++r) s += red[(r * VPP + tid) * EPV + j];
            atomicAdd(&rowsum[(int64_t)n * C + tid * EPV + j], s);
        }
    }
}

template <typename T>
__global__ void __launch_bounds__(256) colsum_group_kernel(const ColJob* __restrict__ jobs, char* __restrict__ bws) {
    constexpr int EPV = 16 / sizeof(T);
    __shared__ float red[256 * EPV];
    const ColJob jb = jobs[blockIdx.x];
    const T* dY = reinterpret_cast<const T*>(bws + jb.dy_off);
    float* rowsum = reinterpret_cast<float*>(bws + jb.rowsum_off);
    const int C = jb.C, ppw = jb.ppw, chunk_px = jb.chunk_px;
    const int tid = threadIdx.x, VPP = C / EPV, slot = tid % VPP, prow = tid / VPP;
    const int n = blockIdx.y;
    const int64_t p0 = (int64_t)n * jb.HW + (int64_t)jb.chunk * chunk_px;
    float acc[EPV];
#pragma unroll
    for (int j = 0; j < EPV; ++j) acc[j] = 0.f;
#define COL_ONE(RAW)                                                   \
    {                                                                  \
        float v[EPV];                                                  \
        unpack_vec<T>(RAW, v);                                         \
        _Pragma("unroll") for (int j = 0; j < EPV; ++j) acc[j] += v[j]; \
    }
    if (prow < ppw) {
        int p = prow;
        for (; p + 3 * ppw < chunk_px; p += 4 * ppw) {
            const T* q = dY + (p0 + p) * C + slot * EPV;
            const uint4 r0 = load_raw<T>(q), r1 = load_raw<T>(q + (int64_t)ppw * C), r2 = load_raw<T>(q + (int64_t)2 * ppw * C),
                        r3 = load_raw<T>(q + (int64_t)3 * ppw * C);
            COL_ONE(r0) COL_ONE(r1) COL_ONE(r2) COL_ONE(r3)
        }
        for (; p < chunk_px; p += ppw) {
            const uint4 r0 = load_raw<T>(dY + (p0 + p) * C + slot * EPV);
            COL_ONE(r0)
        }
    }
#undef COL_ONE
#pragma unroll
    for (int j = 0; j < EPV; ++j) red[tid * EPV + j] = acc[j];
    __syncthreads();
    if (tid < VPP) {
#pragma unroll
        for (int j = 0; j < EPV; ++j) {
            float sm = 0.f;
            for (int r = 0; r < ppw; ++r) sm += red[(r * VPP + tid) * EPV + j];
            atomicAdd(&rowsum[(int64_t)n * C + tid * EPV + j], sm);
        }
    }
}
int colsum_group_chunks(int dtype, int HW, int C, int* chunk_px, int* ppw) {
    int cp, nc, pw;
    if (!vec_geometry(dtype, HW, C, cp, nc, pw)) return 0;
    *chunk_px = cp;
    *ppw = pw;
    return nc;
}
int launch_colsum_group(int dtype, const ColJob* jobs_dev, int njobs, void* bws, int N, hipStream_t s) {
    if (njobs <= 0) return DMME_OK;
    dim3 grid((unsigned)njobs, (unsigned)N);
    if (dtype == DMME_BF16)
        hipLaunchKernelGGL(colsum_group_kernel<bf16>, grid, dim3(256), 0, s, jobs_dev, (char*)bws);
    else if (dtype == DMME_F16)
        hipLaunchKernelGGL(colsum_group_kernel<f16>, grid, dim3(256), 0, s, jobs_dev, (char*)bws);
    else
        hipLaunchKernelGGL(colsum_group_kernel<float>, grid, dim3(256), 0, s, jobs_dev, (char*)bws);
    DMME_CHECK_LAUNCH();
    return DMME_OK;
}

// dbias[c] += sum_n rowsum[n][c];  d_tproj rows (per image, or the single broadcast row); 32 channels per
// workgroup, the batch split 8 ways and combined through LDS
__global__ void __launch_bounds__(256) bias_tproj_fast_kernel(const float* __restrict__ rowsum, int N, int C, float* __restrict__ dbias,
                                                              float* __restrict__ dtproj, int ld, int nt) {
    __shared__ float red[8][33];
    const int cl = threadIdx.x & 31, seg = threadIdx.x >> 5;
    const int c = blockIdx.x * 32 + cl;
    float acc = 0.f;
    if (c < C)
        for (int n = seg; n < N; n += 8) {
            const float v = rowsum[(int64_t)n * C + c];
            acc += v;
            if (dtproj && nt > 1) dtproj[(int64_t)n * ld + c] = v;
        }
    red[seg][cl] = acc;
    __syncthreads();
    if (seg == 0 && c < C) {
        float tot = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) tot += red[k][cl];
        if (dbias) dbias[c] += tot;
        if (dtproj && nt == 1) dtproj[c] = tot;
    }
}

bool colsum_fast_supported(int dtype, int HW, int C) {
    int a, b, c;
    return vec_geometry(dtype, HW, C, a, b, c);
}

int launch_colsum_fast(int dtype, const void* dY, int N, int HW, int C, float* rowsum, float* dbias, float* dtproj, int ld, int nt,
                       hipStream_t s) {
    int chunk_px, nchunks, ppw;
    DMME_REQUIRE(vec_geometry(dtype, HW, C, chunk_px, nchunks, ppw), DMME_ERR_UNSUPPORTED, "colsum_fast: unsupported geometry");
    dim3 grid(nchunks, N);
    if (dtype == DMME_BF16)
        hipLaunchKernelGGL(colsum_vec_kernel<bf16>, grid, dim3(256), 0, s, (const bf16*)dY, HW, C, chunk_px, ppw, rowsum);
    else if (dtype == DMME_F16)
        hipLaunchKernelGGL(colsum_vec_kernel<f16>, grid, dim3(256), 0, s, (const f16*)dY, HW, C, chunk_px, ppw, rowsum);
    else
        hipLaunchKernelGGL(colsum_vec_kernel<float>, grid, dim3(256), 0, s, (const float*)dY, HW, C, chunk_px, ppw, rowsum);
    DMME_CHECK_LAUNCH();
    if (!dbias && !dtproj) return DMME_OK;  // the caller reduces rowsum later (launch_bias_tproj_group)
    hipLaunchKernelGGL(bias_tproj_fast_kernel, dim3((C + 31) / 32), dim3(256), 0, s, rowsum, N, C, dbias, dtproj, ld, nt);
    DMME_CHECK_LAUNCH();
    return DMME_OK;
}

// the same reduction for MANY convs in one launch: one workgroup per (conv, 32-channel block) job
__global__ void __launch_bounds__(256) bias_tproj_group_kernel(const BiasJob* __restrict__ jobs, const char* __restrict__ bws, float* __restrict__ grad_flat,
                                                               float* __restrict__ dtproj, int N, int ld, int nt) {
    __shared__ float red[8][33];
    const BiasJob jb = jobs[blockIdx.x];
    const float* rowsum = reinterpret_cast<const float*>(bws + jb.rowsum_off);
    const int cl = threadIdx.x & 31, seg = threadIdx.x >> 5;
    const int c = jb.cblock * 32 + cl, C = jb.C;
    float* dt = jb.tcol >= 0 ? dtproj + jb.tcol : nullptr;
    float acc = 0.f;
    if (c < C)
        for (int n = seg; n < N; n += 8) {
            const float v = rowsum[(int64_t)n * C + c];
            acc += v;
            if (dt && nt > 1) dt[(int64_t)n * ld + c] = v;
        }
    red[seg][cl] = acc;
    __syncthreads();
    if (seg == 0 && c < C) {
        float tot = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) tot += red[k][cl];
        grad_flat[jb.dbias_off + c] += tot;
        if (dt && nt == 1) dt[c] = tot;
    }
}
int launch_bias_tproj_group(const BiasJob* jobs_dev, int njobs, const void* bws, float* grad_flat, float* dtproj, int N, int ld, int nt, hipStream_t s) {
    if (njobs <= 0) return DMME_OK;
    hipLaunchKernelGGL(bias_tproj_group_kernel, dim3((unsigned)njobs), dim3(256), 0, s, jobs_dev, (const char*)bws, grad_flat, dtproj, N, ld, nt);
    DMME_CHECK_LAUNCH();
    return DMME_OK;
}

// ------------------------------------------------------------------ GroupNorm backward, pass A
template <typename T>
__global__ void __launch_bounds__(256) gn_bwd_sums_kernel(const T* __restrict__ dv, const T* __restrict__ x1, const T* __restrict__ x2, int HW,
                                                          int C1, int C2, int groups, const float* __restrict__ mean_rstd,
                                                          const float* __restrict__ scale, const float* __restrict__ shift,
                                                          const float* __restrict__ dmask, int pro_silu, int chunk_px, int ppw,
                                                          float* __restrict__ AB /* [pixel chunks][N][C][2] */) {
    constexpr int EPV = 16 / sizeof(T);
    __shared__ float red[256 * EPV * 2];
    const int C = C1 + C2, tid = threadIdx.x, VPP = C / EPV, slot = tid % VPP, prow = tid / VPP;
    const int n = blockIdx.y, c0 = slot * EPV, cg = C / groups;
    const bool second = c0 >= C1;
    const T* xs = second ? x2 : x1;
    const int Cs = second ? C2 : C1, cs0 = second ? c0 - C1 : c0;
    const int64_t p0 = (int64_t)n * HW + (int64_t)blockIdx.x * chunk_px;
    float sc[EPV], sh[EPV], dm[EPV], mu[EPV], rs[EPV];
#pragma unroll
    for (int j = 0; j < EPV; ++j) {
        const int c = c0 + j;
        sc[j] = scale[(int64_t)n * C + c];
        sh[j] = shift[(int64_t)n * C + c];
        dm[j] = dmask ? dmask[(int64_t)n * C + c] : 1.0f;
        mu[j] = mean_rstd[((int64_t)n * groups + c / cg) * 2];
        rs[j] = mean_rstd[((int64_t)n * groups + c / cg) * 2 + 1];
    }
    float a[EPV], b[EPV];
#pragma unroll
    for (int j = 0; j < EPV; ++j) a[j] = b[j] = 0.f;
#define SUMS_ONE(RD, RX)                                                                  \
    {                                                                                     \
        float d[EPV], xv[EPV];                                                            \
        unpack_vec<T>(RD, d);                                                             \
        unpack_vec<T>(RX, xv);                                                            \
        _Pragma("unroll") for (int j = 0; j < EPV; ++j) {                                 \
            float du = d[j] * dm[j];                                                      \
            if (pro_silu) du *= silu_grad_f<T>(fmaf(xv[j], sc[j], sh[j]));                \
            a[j] += du;                                                                   \
            b[j] = fmaf(du, (xv[j] - mu[j]) * rs[j], b[j]);                               \
        }                                                                                 \
    }
    if (prow < ppw) {
        int p = prow;
        for (; p + 3 * ppw < chunk_px; p += 4 * ppw) {
            const T* qd = dv + (p0 + p) * C + c0;
            const T* qx = xs + (p0 + p) * Cs + cs0;
            const int64_t sd = (int64_t)ppw * C, sx = (int64_t)ppw * Cs;
            const uint4 d0 = load_raw<T>(qd), d1 = load_raw<T>(qd + sd), d2 = load_raw<T>(qd + 2 * sd), d3 = load_raw<T>(qd + 3 * sd);
            const uint4 x0 = load_raw<T>(qx), x1 = load_raw<T>(qx + sx), x2 = load_raw<T>(qx + 2 * sx), x3 = load_raw<T>(qx + 3 * sx);
            SUMS_ONE(d0, x0) SUMS_ONE(d1, x1) SUMS_ONE(d2, x2) SUMS_ONE(d3, x3)
        }
        for (; p < chunk_px; p += ppw) {
            const uint4 d0 = load_raw<T>(dv + (p0 + p) * C + c0), x0 = load_raw<T>(xs + (p0 + p) * Cs + cs0);
            SUMS_ONE(d0, x0)
        }
    }
#undef SUMS_ONE
#pragma unroll
    for (int j = 0; j < EPV; ++j) {
        red[(tid * EPV + j) * 2] = a[j];
        red[(tid * EPV + j) * 2 + 1] = b[j];
    }
    __syncthreads();
    if (tid < VPP) {
#pragma unroll
        for (int j = 0; j < EPV; ++j) {
            float sa = 0.f, sb = 0.f;
            for (int r = 0; r < ppw; ++r) {
                sa += red[((r * VPP + tid) * EPV + j) * 2];
                sb += red[((r * VPP + tid) * EPV + j) * 2 + 1];
            }
            // one partial per (pixel chunk, image, channel), summed by the finalize kernel: four atomics per (image, channel) pair
            // cost more than the pass over the tensor (131 k float atomics per launch at 128 channels: ~20 of its 24 us)
            float* ab = AB + ((((int64_t)blockIdx.x * gridDim.y + n) * C) + tid * EPV + j) * 2;
            ab[0] = sa;
            ab[1] = sb;
        }
    }
}

// finalize: blocks [0, nb_s): S[n][g] = {sum gamma A, sum gamma B};
//           remaining blocks: 32 channels each, dgamma_c += sum_n B, dbeta_c += sum_n A (8-way split over n + LDS)
// {A, B} of one (image, channel) summed over the pixel chunks, fixed order (reproducible); the 8-byte loads of four chunks go out
// together - this kernel is nothing but dependent memory round trips
__device__ __forceinline__ float2 ab_sum(const float* __restrict__ AB, int nchunks, int64_t stride, int64_t pair) {
    const float2* q = reinterpret_cast<const float2*>(AB) + pair;
    const int64_t st = stride / 2;
    float2 s = q[0];
    int k = 1;
    for (; k + 3 < nchunks; k += 4) {
        const float2 v0 = q[k * st], v1 = q[(k + 1) * st], v2 = q[(k + 2) * st], v3 = q[(k + 3) * st];
        s.x += v0.x; s.y += v0.y;
        s.x += v1.x; s.y += v1.y;
        s.x += v2.x; s.y += v2.y;
        s.x += v3.x; s.y += v3.y;
    }
    for (; k < nchunks; ++k) {
        const float2 v = q[k * st];
        s.x += v.x;
        s.y += v.y;
    }
    return s;
}
__global__ void __launch_bounds__(256) gn_bwd_finalize_kernel(const float* __restrict__ AB, int nchunks, int N, int C, int groups, int nb_s,
                                                              const float* __restrict__ gamma, float* __restrict__ S,
                                                              float* __restrict__ dgamma, float* __restrict__ dbeta, GnMod mod) {
    const int cg = C / groups;
    const int64_t cstride = (int64_t)N * C * 2;
    if ((int)blockIdx.x < nb_s) {
        const int i = blockIdx.x * blockDim.x + threadIdx.x;
        if (i >= N * groups) return;
        const int n = i / groups, g = i % groups;
        float s1 = 0.f, s2 = 0.f;
        for (int j = 0; j < cg; ++j) {
            const int c = g * cg + j;
            const float gm = gamma[c] * mod.mul(n, c);
            const float2 ab = ab_sum(AB, nchunks, cstride, (int64_t)n * C + c);
            s1 = fmaf(gm, ab.x, s1);
            s2 = fmaf(gm, ab.y, s2);
        }
        S[(int64_t)i * 2] = s1;
        S[(int64_t)i * 2 + 1] = s2;
        return;
    }
    __shared__ float ra[8][33], rb[8][33];
    const int cl = threadIdx.x & 31, seg = threadIdx.x >> 5;
    const int c = ((int)blockIdx.x - nb_s) * 32 + cl;
    float a = 0.f, b = 0.f;
    if (c < C)
        for (int n = seg; n < N; n += 8) {
            const float2 ab = ab_sum(AB, nchunks, cstride, (int64_t)n * C + c);
            const float an = ab.x, bn = ab.y, m = mod.mul(n, c);
            a = fmaf(an, m, a);
            b = fmaf(bn, m, b);
            mod.emit(n, c, an, bn, gamma[c]);
        }
    ra[seg][cl] = a;
    rb[seg][cl] = b;
    __syncthreads();
    if (seg == 0 && c < C) {
        float ta = 0.f, tb = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            ta += ra[k][cl];
            tb += rb[k][cl];
        }
        dbeta[c] += ta;
        dgamma[c] += tb;
    }
}

// The same finalize, one workgroup per IMAGE (C <= 1024): a thread sums the pixel chunks of its channels with every load in flight at
// once (coalesced over channels: the per-chunk rows are [N][C] pairs), the group sums follow through LDS, and the cross-image sums
// dgamma / dbeta take one float atomic per (image, channel) - like the small-map kernel always did.  The (image, group)-per-thread
// form above walked cg x nchunks dependent loads per thread, twice over the buffer: 13.3 us for 8 MB; this one 4-5 us.
__global__ void __launch_bounds__(256) gn_bwd_finalize_image_kernel(const float* __restrict__ AB, int nchunks, int N, int C, int groups,
                                                                    const float* __restrict__ gamma, float* __restrict__ S,
                                                                    float* __restrict__ dgamma, float* __restrict__ dbeta, GnMod mod) {
    __shared__ float ga[1024], gb[1024];
    const int n = blockIdx.x, tid = threadIdx.x, cg = C / groups;
    const int64_t cstride = (int64_t)N * C;  // pairs per chunk row
    const float2* q0 = reinterpret_cast<const float2*>(AB) + (int64_t)n * C;
    for (int c = tid; c < C; c += 256) {
        const float2* q = q0 + c;
        float sa = 0.f, sb = 0.f;
        int k = 0;
        for (; k + 7 < nchunks; k += 8) {
            float2 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = q[(int64_t)(k + u) * cstride];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                sa += v[u].x;
                sb += v[u].y;
            }
        }
        for (; k < nchunks; ++k) {
            const float2 v = q[(int64_t)k * cstride];
            sa += v.x;
            sb += v.y;
        }
        const float gmm = gamma[c], m = mod.mul(n, c);
        ga[c] = gmm * m * sa;
        gb[c] = gmm * m * sb;
        atomicAdd(&dbeta[c], sa * m);
        atomicAdd(&dgamma[c], sb * m);
        mod.emit(n, c, sa, sb, gmm);
    }
    __syncthreads();
    if (tid < groups) {
        float s1 = 0.f, s2 = 0.f;
        for (int j = 0; j < cg; ++j) {
            s1 += ga[tid * cg + j];
            s2 += gb[tid * cg + j];
        }
        S[((int64_t)n * groups + tid) * 2] = s1;
        S[((int64_t)n * groups + tid) * 2 + 1] = s2;
    }
}

// pass B: dx, split over the two concatenated destinations, write or accumulate
template <typename T>
__global__ void __launch_bounds__(256) gn_bwd_apply_kernel(const T* __restrict__ dv, const T* __restrict__ x1, const T* __restrict__ x2, int HW,
                                                           int C1, int C2, int groups, const float* __restrict__ gamma,
                                                           const float* __restrict__ mean_rstd, const float* __restrict__ scale,
                                                           const float* __restrict__ shift, const float* __restrict__ dmask, int pro_silu,
                                                           const float* __restrict__ S, int chunk_px, int ppw, T* __restrict__ dx1, T* __restrict__ dx2,
                                                           int acc1, int acc2, GnMod mod, T* __restrict__ act, const float* __restrict__ AB,
                                                           float* __restrict__ dgamma, float* __restrict__ dbeta, float* __restrict__ rows,
                                                           const T* __restrict__ extra) {
    constexpr int EPV = 16 / sizeof(T);
    const int C = C1 + C2, tid = threadIdx.x, VPP = C / EPV, slot = tid % VPP, prow = tid / VPP;
    const int n = blockIdx.y, c0 = slot * EPV, cg = C / groups;
    // AB != null: the group sums come from the per-chunk channel sums HERE instead of from a finalize launch between the two passes
    // (gn_bwd_finalize_image_kernel's arithmetic, same order: same bits).  Every workgroup of an image repeats the merge - the partial
    // rows of one norm are a few MB, L2-resident, and the reads hide behind the other workgroups' streaming; the workgroup of the
    // image's first chunk also adds the batch sums (dgamma, dbeta, the IDDPM conditioning rows).  26 launches fewer per training step.
    __shared__ float ga[1024], gb[1024], gS1[256], gS2[256];
    if (AB) {
        const int nchunks = (int)gridDim.x, N = (int)gridDim.y;
        const int64_t cstride = (int64_t)N * C;  // pairs per chunk row
        const float2* q0 = reinterpret_cast<const float2*>(AB) + (int64_t)n * C;
        for (int c = tid; c < C; c += 256) {
            const float2* q = q0 + c;
            float sa = 0.f, sb = 0.f;
            int k = 0;
            for (; k + 7 < nchunks; k += 8) {
                float2 v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = q[(int64_t)(k + u) * cstride];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    sa += v[u].x;
                    sb += v[u].y;
                }
            }
            for (; k < nchunks; ++k) {
                const float2 v = q[(int64_t)k * cstride];
                sa += v.x;
                sb += v.y;
            }
            const float gmm = gamma[c], m = mod.mul(n, c);
            ga[c] = gmm * m * sa;
            gb[c] = gmm * m * sb;
            if (blockIdx.x == 0) {
                if (rows) {  // (see gn_bwd_small_kernel)
                    rows[(int64_t)n * C + c] = sa * m;
                    rows[((int64_t)N + n) * C + c] = sb * m;
                } else {
                    atomicAdd(&dbeta[c], sa * m);
                    atomicAdd(&dgamma[c], sb * m);
                }
                mod.emit(n, c, sa, sb, gmm);
            }
        }
        __syncthreads();
        if (tid < groups) {
            float s1 = 0.f, s2 = 0.f;
            for (int j = 0; j < cg; ++j) {
                s1 += ga[tid * cg + j];
                s2 += gb[tid * cg + j];
            }
            gS1[tid] = s1;
            gS2[tid] = s2;
        }
        __syncthreads();
    }
    if (prow >= ppw) return;
    const bool second = c0 >= C1;
    const T* xs = second ? x2 : x1;
    T* dst = second ? dx2 : dx1;
    const int acc = second ? acc2 : acc1;
    const int Cs = second ? C2 : C1, cs0 = second ? c0 - C1 : c0;
    const int64_t p0 = (int64_t)n * HW + (int64_t)blockIdx.x * chunk_px;
    const float inv = 1.0f / (float)((int64_t)cg * HW);
    // per-channel constants, folded (fewer live registers: 157 -> occupancy 3 of 8 was the kernel's limit):
    //   dx = rs (du gm - k1 - xhat k2),  du = d dm silu'(y),  xhat = (x - mu) rs
    //      = d silu'(y) Gd + x Cx + C0   with  Gd = rs gm dm,  Cx = -rs^2 k2,  C0 = rs (mu rs k2 - k1)
    float sc[EPV], sh[EPV], dm[EPV], Gd[EPV], Cx[EPV], C0[EPV];
#pragma unroll
    for (int j = 0; j < EPV; ++j) {
        const int c = c0 + j, g = c / cg;
        sc[j] = scale[(int64_t)n * C + c];
        sh[j] = shift[(int64_t)n * C + c];
        dm[j] = dmask ? dmask[(int64_t)n * C + c] : 1.0f;
        const float mu = mean_rstd[((int64_t)n * groups + g) * 2], rs = mean_rstd[((int64_t)n * groups + g) * 2 + 1];
        const float gm = gamma[c] * mod.mul(n, c);
        const float k1 = (AB ? gS1[g] : S[((int64_t)n * groups + g) * 2]) * inv;
        const float k2 = (AB ? gS2[g] : S[((int64_t)n * groups + g) * 2 + 1]) * inv;
        Gd[j] = rs * gm * dm[j];
        Cx[j] = -rs * rs * k2;
        C0[j] = rs * (mu * rs * k2 - k1);
    }
    // `extra` (single-source norms): one more addend of the source's gradient - the identity-residual branch of the block (d x += d out),
    // which used to be its own read-modify-write launch over the same tensor (27 per training step)
#define APPLY_ONE(RD, RX, RO, PP)                                                         \
    {                                                                                     \
        float d[EPV], xv[EPV], o[EPV], ev[EPV];                                           \
        unpack_vec<T>(RD, d);                                                             \
        unpack_vec<T>(RX, xv);                                                            \
        unpack_vec<T>(RO, o);                                                             \
        unpack_vec<T>(extra ? load_raw<T>(extra + (p0 + (PP)) * Cs + cs0) : zero4, ev);   \
        float yv[EPV];                                                                    \
        _Pragma("unroll") for (int j = 0; j < EPV; ++j) {                                 \
            yv[j] = fmaf(xv[j], sc[j], sh[j]);                                            \
            float du = d[j] * Gd[j];                                                      \
            if (pro_silu) du *= silu_grad_f<T>(yv[j]);                                    \
            const float dx = fmaf(xv[j], Cx[j], du + C0[j]) + ev[j];                      \
            o[j] = acc ? o[j] + dx : dx;                                                  \
        }                                                                                 \
        store_vec<T>(dst + (p0 + (PP)) * Cs + cs0, o);                                    \
        if (act) { /* the conv's pre-activated input, for the deferred weight gradient: the forward's own prologue (prologue_vec:    \
                      fma, SiLU, mask, in that order) on the same bits - written out on the registers: handing sc / sh / dm to a     \
                      function by pointer put the three arrays in scratch memory (48 bytes per lane, re-read per vector) */          \
            float av[EPV];                                                                \
            _Pragma("unroll") for (int j = 0; j < EPV; ++j) {                             \
                float a_ = yv[j];                                                         \
                if (pro_silu) a_ = sizeof(T) == 2 ? silu_fast(a_) : silu_f(a_);          \
                av[j] = dmask ? a_ * dm[j] : a_;                                          \
            }                                                                             \
            store_vec<T>(act + (p0 + (PP)) * C + c0, av);                                 \
        }                                                                                 \
    }
    const uint4 zero4 = make_uint4(0u, 0u, 0u, 0u);
    int p = prow;
    for (; p + 3 * ppw < chunk_px; p += 4 * ppw) {
        const T* qd = dv + (p0 + p) * C + c0;
        const T* qx = xs + (p0 + p) * Cs + cs0;
        const T* qo = dst + (p0 + p) * Cs + cs0;
        const int64_t sd = (int64_t)ppw * C, sx = (int64_t)ppw * Cs;
        const uint4 d0 = load_raw<T>(qd), d1 = load_raw<T>(qd + sd), d2 = load_raw<T>(qd + 2 * sd), d3 = load_raw<T>(qd + 3 * sd);
        const uint4 x0 = load_raw<T>(qx), x1 = load_raw<T>(qx + sx), x2 = load_raw<T>(qx + 2 * sx), x3 = load_raw<T>(qx + 3 * sx);
        uint4 o0 = zero4, o1 = zero4, o2 = zero4, o3 = zero4;
        if (acc) {
            o0 = load_raw<T>(qo);
            o1 = load_raw<T>(qo + sx);
            o2 = load_raw<T>(qo + 2 * sx);
            o3 = load_raw<T>(qo + 3 * sx);
        }
        APPLY_ONE(d0, x0, o0, p) APPLY_ONE(d1, x1, o1, p + ppw) APPLY_ONE(d2, x2, o2, p + 2 * ppw) APPLY_ONE(d3, x3, o3, p + 3 * ppw)
    }
    for (; p < chunk_px; p += ppw) {
        const uint4 d0 = load_raw<T>(dv + (p0 + p) * C + c0), x0 = load_raw<T>(xs + (p0 + p) * Cs + cs0);
        const uint4 o0 = acc ? load_raw<T>(dst + (p0 + p) * Cs + cs0) : zero4;
        APPLY_ONE(d0, x0, o0, p)
    }
#undef APPLY_ONE
}

// dst1[.., :C1] (+)= src[.., :C1], dst2[.., :C2] (+)= src[.., C1:]  with 16-byte accesses
template <typename T>
__global__ void __launch_bounds__(256) grad_acc_vec_kernel(const T* __restrict__ src, T* __restrict__ d1, T* __restrict__ d2, int C1, int C2,
                                                           int acc1, int acc2, int64_t nvec) {
    constexpr int EPV = 16 / sizeof(T);
    const int C = C1 + C2, VPP = C / EPV;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t p = i / VPP;
        const int c0 = (int)(i % VPP) * EPV;
        float v[EPV], o[EPV];
        load_vec<T>(src + p * C + c0, v);
        const bool second = c0 >= C1;
        T* dst = second ? d2 + p * C2 + (c0 - C1) : d1 + p * C1 + c0;
        if (second ? acc2 : acc1) {
            load_vec<T>(dst, o);
#pragma unroll
            for (int j = 0; j < EPV; ++j) v[j] += o[j];
        }
        store_vec<T>(dst, v);
    }
}
bool grad_acc_fast_supported(int dtype, int C1, int C2, int pool) {
    const int EPV = is16(dtype) ? 8 : 4;
    return !pool && C1 % EPV == 0 && C2 % EPV == 0;
}
int launch_grad_acc_fast(int dtype, const void* src, void* d1, void* d2, int C1, int C2, int acc1, int acc2, int64_t npix, hipStream_t s) {
    const int EPV = is16(dtype) ? 8 : 4;
    const int64_t nvec = npix * ((C1 + C2) / EPV);
    if (nvec == 0) return DMME_OK;
    int64_t blocks = (nvec + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    if (dtype == DMME_BF16)
        hipLaunchKernelGGL(grad_acc_vec_kernel<bf16>, dim3((unsigned)blocks), dim3(256), 0, s, (const bf16*)src, (bf16*)d1, (bf16*)d2, C1, C2, acc1, acc2, nvec);
    else if (dtype == DMME_F16)
        hipLaunchKernelGGL(grad_acc_vec_kernel<f16>, dim3((unsigned)blocks), dim3(256), 0, s, (const f16*)src, (f16*)d1, (f16*)d2, C1, C2, acc1, acc2, nvec);
    else
        hipLaunchKernelGGL(grad_acc_vec_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, s, (const float*)src, (float*)d1, (float*)d2, C1, C2, acc1, acc2, nvec);
    DMME_CHECK_LAUNCH();
    return DMME_OK;
}

// Small feature maps (<= 64 pixels): ONE workgroup per image does the whole GroupNorm(+SiLU+Dropout2d) backward - channel
// sums A, B in LDS (fixed-order reduction), the group sums, then a second pass over the (L2-resident) image for dx; the
// batch sums of dgamma / dbeta go out as one atomic per channel per image.  One launch instead of three.
template <typename T>
__global__ void __launch_bounds__(256) gn_bwd_small_kernel(const T* __restrict__ dv, const T* __restrict__ x1, const T* __restrict__ x2, int HW, int C1,
                                                           int C2, int groups, const float* __restrict__ gamma, const float* __restrict__ mean_rstd,
                                                           const float* __restrict__ scale, const float* __restrict__ shift,
                                                           const float* __restrict__ dmask, int pro_silu, T* __restrict__ dx1, T* __restrict__ dx2,
                                                           int acc1, int acc2, float* __restrict__ dgamma, float* __restrict__ dbeta, T* __restrict__ act,
                                                           float* __restrict__ rows, const T* __restrict__ extra) {
    constexpr int EPV = 16 / sizeof(T);
    __shared__ float red[256 * EPV * 2];
    __shared__ float chA[512], chB[512], gS1[64], gS2[64];
    // gridDim.y channel slices of whole groups per image (groups are independent): a quarter of the serial work per workgroup, four
    // times the workgroups - the kernel is three dependent phases over an image that already sits in L2
    const int C = C1 + C2, Cw = C / (int)gridDim.y, cb = (int)blockIdx.y * Cw;
    const int tid = threadIdx.x, VPP = Cw / EPV, ppw = 256 / VPP, slot = tid % VPP, prow = tid / VPP;
    const int n = blockIdx.x, c0 = cb + slot * EPV, cg = C / groups;
    const bool active = prow < ppw;
    const bool second = c0 >= C1;
    const T* xs = second ? x2 : x1;
    T* dst = second ? dx2 : dx1;
    const int acc = second ? acc2 : acc1;
    const int Cs = second ? C2 : C1, cs0 = second ? c0 - C1 : c0;
    const int64_t p0 = (int64_t)n * HW;
    float sc[EPV], sh[EPV], dm[EPV], mu[EPV], rs[EPV], gm[EPV];
#pragma unroll
    for (int j = 0; j < EPV; ++j) {
        const int c = active ? c0 + j : 0, g = c / cg;
        sc[j] = scale[(int64_t)n * C + c];
        sh[j] = shift[(int64_t)n * C + c];
        dm[j] = dmask ? dmask[(int64_t)n * C + c] : 1.0f;
        mu[j] = mean_rstd[((int64_t)n * groups + g) * 2];
        rs[j] = mean_rstd[((int64_t)n * groups + g) * 2 + 1];
        gm[j] = gamma[c];
    }
    float a[EPV], bq[EPV];
#pragma unroll
    for (int j = 0; j < EPV; ++j) a[j] = bq[j] = 0.f;
#define SMALL_SUMS_ONE(RD, RX)                                                            \
    {                                                                                     \
        float d[EPV], xv[EPV];                                                            \
        unpack_vec<T>(RD, d);                                                             \
        unpack_vec<T>(RX, xv);                                                            \
        _Pragma("unroll") for (int j = 0; j < EPV; ++j) {                                 \
            float du = d[j] * dm[j];                                                      \
            if (pro_silu) du *= silu_grad_f<T>(fmaf(xv[j], sc[j], sh[j]));                \
            a[j] += du;                                                                   \
            bq[j] = fmaf(du, (xv[j] - mu[j]) * rs[j], bq[j]);                             \
        }                                                                                 \
    }
    if (active) {
        int p = prow;
        for (; p + 3 * ppw < HW; p += 4 * ppw) {
            const T* qd = dv + (p0 + p) * C + c0;
            const T* qx = xs + (p0 + p) * Cs + cs0;
            const int64_t sd = (int64_t)ppw * C, sx = (int64_t)ppw * Cs;
            const uint4 d0 = load_raw<T>(qd), d1 = load_raw<T>(qd + sd), d2 = load_raw<T>(qd + 2 * sd), d3 = load_raw<T>(qd + 3 * sd);
            const uint4 x0 = load_raw<T>(qx), x1 = load_raw<T>(qx + sx), x2 = load_raw<T>(qx + 2 * sx), x3 = load_raw<T>(qx + 3 * sx);
            SMALL_SUMS_ONE(d0, x0) SMALL_SUMS_ONE(d1, x1) SMALL_SUMS_ONE(d2, x2) SMALL_SUMS_ONE(d3, x3)
        }
        for (; p < HW; p += ppw) {
            const uint4 d0 = load_raw<T>(dv + (p0 + p) * C + c0), x0 = load_raw<T>(xs + (p0 + p) * Cs + cs0);
            SMALL_SUMS_ONE(d0, x0)
        }
    }
#undef SMALL_SUMS_ONE
    if (active)
#pragma unroll
        for (int j = 0; j < EPV; ++j) {
            red[(tid * EPV + j) * 2] = a[j];
            red[(tid * EPV + j) * 2 + 1] = bq[j];
        }
    __syncthreads();
    for (int lc = tid; lc < Cw; lc += 256) {  // lc: channel within this slice
        const int v = lc / EPV, j = lc % EPV;
        float sa = 0.f, sb = 0.f;
        for (int r = 0; r < ppw; ++r) {
            sa += red[((r * VPP + v) * EPV + j) * 2];
            sb += red[((r * VPP + v) * EPV + j) * 2 + 1];
        }
        chA[lc] = sa;
        chB[lc] = sb;
        if (rows) {  // per-image rows [2][N][C], summed over the batch by the grouped bias reduction at the end of backward: 128 images
            rows[(int64_t)n * C + cb + lc] = sa;                              // adding into the same C addresses were 128-way atomics
            rows[((int64_t)gridDim.x + n) * C + cb + lc] = sb;               // (2.6 us of this kernel's 15)
        } else {
            atomicAdd(&dbeta[cb + lc], sa);
            atomicAdd(&dgamma[cb + lc], sb);
        }
    }
    __syncthreads();
    if (tid < Cw / cg) {  // the slice's groups
        float s1 = 0.f, s2 = 0.f;
        for (int j = 0; j < cg; ++j) {
            const int lc = tid * cg + j;
            s1 = fmaf(gamma[cb + lc], chA[lc], s1);
            s2 = fmaf(gamma[cb + lc], chB[lc], s2);
        }
        gS1[tid] = s1;
        gS2[tid] = s2;
    }
    __syncthreads();
    if (!active) return;
    const float inv = 1.0f / (float)((int64_t)cg * HW);
    float k1[EPV], k2[EPV];
#pragma unroll
    for (int j = 0; j < EPV; ++j) {
        const int g = (c0 - cb + j) / cg;
        k1[j] = gS1[g] * inv;
        k2[j] = gS2[g] * inv;
    }
#define SMALL_APPLY_ONE(RD, RX, RO, PP)                                                   \
    {                                                                                     \
        float d[EPV], xv[EPV], o[EPV], ev[EPV];                                           \
        unpack_vec<T>(RD, d);                                                             \
        unpack_vec<T>(RX, xv);                                                            \
        unpack_vec<T>(RO, o);                                                             \
        unpack_vec<T>(extra ? load_raw<T>(extra + (p0 + (PP)) * Cs + cs0) : zero4, ev);   \
        _Pragma("unroll") for (int j = 0; j < EPV; ++j) {                                 \
            float du = d[j] * dm[j];                                                      \
            if (pro_silu) du *= silu_grad_f<T>(fmaf(xv[j], sc[j], sh[j]));                \
            const float xhat = (xv[j] - mu[j]) * rs[j];                                   \
            const float dx = rs[j] * (du * gm[j] - (k1[j] + xhat * k2[j])) + ev[j];       \
            o[j] = acc ? o[j] + dx : dx;                                                  \
        }                                                                                 \
        store_vec<T>(dst + (p0 + (PP)) * Cs + cs0, o);                                    \
        if (act) *reinterpret_cast<uint4*>(act + (p0 + (PP)) * C + c0) = prologue_vec<T>(RX, sc, sh, dmask ? dm : nullptr, pro_silu); \
    }
    const uint4 zero4 = make_uint4(0u, 0u, 0u, 0u);
    int p = prow;
    for (; p + 3 * ppw < HW; p += 4 * ppw) {
        const T* qd = dv + (p0 + p) * C + c0;
        const T* qx = xs + (p0 + p) * Cs + cs0;
        const T* qo = dst + (p0 + p) * Cs + cs0;
        const int64_t sd = (int64_t)ppw * C, sx = (int64_t)ppw * Cs;
        const uint4 d0 = load_raw<T>(qd), d1 = load_raw<T>(qd + sd), d2 = load_raw<T>(qd + 2 * sd), d3 = load_raw<T>(qd + 3 * sd);
        const uint4 x0 = load_raw<T>(qx), x1 = load_raw<T>(qx + sx), x2 = load_raw<T>(qx + 2 * sx), x3 = load_raw<T>(qx + 3 * sx);
        uint4 o0 = zero4, o1 = zero4, o2 = zero4, o3 = zero4;
        if (acc) {
            o0 = load_raw<T>(qo);
            o1 = load_raw<T>(qo + sx);
            o2 = load_raw<T>(qo + 2 * sx);
            o3 = load_raw<T>(qo + 3 * sx);
        }
        SMALL_APPLY_ONE(d0, x0, o0, p) SMALL_APPLY_ONE(d1, x1, o1, p + ppw) SMALL_APPLY_ONE(d2, x2, o2, p + 2 * ppw)
        SMALL_APPLY_ONE(d3, x3, o3, p + 3 * ppw)
    }
    for (; p < HW; p += ppw) {
        const uint4 d0 = load_raw<T>(dv + (p0 + p) * C + c0), x0 = load_raw<T>(xs + (p0 + p) * Cs + cs0);
        const uint4 o0 = acc ? load_raw<T>(dst + (p0 + p) * Cs + cs0) : zero4;
        SMALL_APPLY_ONE(d0, x0, o0, p)
    }
#undef SMALL_APPLY_ONE
}

// 16x16 / 32x32 maps: the same one-launch form with the workgroup's slice of the image - (image, Cw channels of whole groups) - held in
// REGISTERS between the two phases: every thread keeps the raw 16-byte vectors of its ITERS pixels of d(act) and x (its pixel /
// channel-vector assignment is the same in both phases), so each tensor is read from memory once and all of a thread's loads are in
// flight together.  Configurations (ITERS, NT): gn_bwd_regs_pick; the uniform switches SILU / ACT (store the activated input) / ACC
// (a destination accumulates) / EXTRA (one more addend) are template arguments.  Replaces gn_bwd_sums + gn_bwd_apply (which read both tensors twice: 201 MB instead of 134 MB on a 128-channel
// 32x32 map at batch 128).  Channel sums: per-thread partials -> LDS -> NT / Cw partial rows -> one thread per channel, fixed order.
template <typename T, int ITERS, int NT, bool SILU, bool ACT, bool ACC, bool EXTRA>
__global__ void __launch_bounds__(NT) gn_bwd_regs_kernel(const T* __restrict__ dv, const T* __restrict__ x1, const T* __restrict__ x2, int HW, int C1,
                                                          int C2, int groups, const float* __restrict__ gamma, const float* __restrict__ mean_rstd,
                                                          const float* __restrict__ scale, const float* __restrict__ shift,
                                                          const float* __restrict__ dmask, int pro_silu, T* __restrict__ dx1, T* __restrict__ dx2,
                                                          int acc1, int acc2, float* __restrict__ dgamma, float* __restrict__ dbeta, T* __restrict__ act,
                                                          float* __restrict__ rows, const T* __restrict__ extra) {
    constexpr int EPV = 16 / sizeof(T);
    __shared__ float red[NT * EPV * 2];
    __shared__ float red2[2 * NT];
    __shared__ float chA[128], chB[128], gS1[32], gS2[32];
    const int C = C1 + C2, Cw = C / (int)gridDim.y, cb = (int)blockIdx.y * Cw;
    const int tid = threadIdx.x, VPP = Cw / EPV, ppw = NT / VPP, slot = tid % VPP, prow = tid / VPP;  // (VPP divides NT: host-checked)
    const int n = blockIdx.x, c0 = cb + slot * EPV, cg = C / groups;
    const bool second = cb >= C1;  // (a slice lies in one of the two concatenated sources: host-checked) - uniform
    const T* xs = second ? x2 : x1;
    T* dst = second ? dx2 : dx1;
    const int acc = second ? acc2 : acc1;
    const int Cs = second ? C2 : C1, cs0 = second ? c0 - C1 : c0;
    // Addresses = a uniform base (the image's first pixel: SGPRs) + a 32-bit per-thread byte offset + k uniform steps: a vector's
    // address is one add away from one register, so nothing address-like stays live between the phases (as 64-bit pointers per
    // vector the sixteen loads, eight stores and eight activated-input stores spilled).  An image is < 2 GB (host-checked).
    const int64_t p0 = (int64_t)n * HW;
    const char* bd = reinterpret_cast<const char*>(dv + p0 * C);
    const char* bx = reinterpret_cast<const char*>(xs + p0 * Cs);
    char* bo = reinterpret_cast<char*>(dst + p0 * Cs);
    char* ba = reinterpret_cast<char*>(act ? act + p0 * C : nullptr);
    const char* be = reinterpret_cast<const char*>(extra ? extra + p0 * Cs : nullptr);
    const uint32_t od = (uint32_t)(prow * C + c0) * 2u, ox = (uint32_t)(prow * Cs + cs0) * 2u;
    const uint32_t sd = (uint32_t)(ppw * C) * 2u, sx = (uint32_t)(ppw * Cs) * 2u;
    uint4 rd[ITERS], rx[ITERS];  // HW = ITERS * ppw exactly (host-checked): no predicates, no control flow around the vectors
#pragma unroll
    for (int k = 0; k < ITERS; ++k) {
        rd[k] = *reinterpret_cast<const uint4*>(bd + (od + k * sd));
        rx[k] = *reinterpret_cast<const uint4*>(bx + (ox + k * sx));
    }
    static_assert(sizeof(T) == 2, "pairs of 16-bit channels");
    // Per-channel constants as the register pairs a dword's two channels need: every operation below is a packed fp32 instruction.
    f32x2 sc[4], sh[4], dm[4];
#pragma unroll
    for (int d = 0; d < 4; ++d) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int c = c0 + 2 * d + h;
            sc[d][h] = scale[(int64_t)n * C + c];
            sh[d][h] = shift[(int64_t)n * C + c];
            dm[d][h] = dmask ? dmask[(int64_t)n * C + c] : 1.0f;
        }
    }
    // Phase 1, per vector as it lands: du = d(act) * mask * silu'(y) - summed in fp32 (sum du, sum du * x per channel; the channel's
    // thread below turns the second into sum du * xhat = rstd (sum du x - mean sum du): mean / rstd stay out of this loop's registers), then kept
    // as the 16-bit vector IN PLACE of the d(act) registers: phase 2 needs nothing else of d(act), y or the sigmoid, so the
    // transcendentals are evaluated once per element, not twice.  (du is rounded like every stored gradient; the sums take it unrounded.)
    // The conv's pre-activated input for the deferred weight gradient (fma, SiLU, mask: the forward's prologue on the same bits) does
    // not depend on the group sums either and leaves from here, its stores draining under the remaining loads.
    using V8 = typename Vec8<T>::type;
    const f32x2 one2 = f32x2{1.0f, 1.0f}, nl2e = f32x2{-1.4426950408889634f, -1.4426950408889634f};
    f32x2 a[4], bq[4];
#pragma unroll
    for (int d = 0; d < 4; ++d) a[d] = bq[d] = f32x2{0.f, 0.f};
#define REGS_PHASE1(SILU, ACT)                                                                     \
    _Pragma("unroll") for (int k = 0; k < ITERS; ++k) { /* (pixels in ascending order per thread, as the two-pass kernels) */ \
        const V8 vd = __builtin_bit_cast(V8, rd[k]), vx = __builtin_bit_cast(V8, rx[k]);            \
        V8 wd, wa;                                                                                  \
        _Pragma("unroll") for (int d = 0; d < 4; ++d) {                                             \
            const f32x2 xx = f32x2{(float)vx[2 * d], (float)vx[2 * d + 1]};                         \
            f32x2 du = f32x2{(float)vd[2 * d], (float)vd[2 * d + 1]} * dm[d];                       \
            const f32x2 y = __builtin_elementwise_fma(xx, sc[d], sh[d]);                            \
            f32x2 av = y;                                                                           \
            if (SILU) {                                                                             \
                f32x2 e = y * nl2e;                                                                 \
                e = one2 + f32x2{__builtin_amdgcn_exp2f(e[0]), __builtin_amdgcn_exp2f(e[1])};       \
                const f32x2 sg = f32x2{__builtin_amdgcn_rcpf(e[0]), __builtin_amdgcn_rcpf(e[1])};   \
                av = y * sg;                                                                        \
                du = du * (sg * __builtin_elementwise_fma(y, one2 - sg, one2));                     \
            }                                                                                       \
            a[d] = a[d] + du;                                                                       \
            bq[d] = __builtin_elementwise_fma(du, xx, bq[d]);                                       \
            wd[2 * d] = (T)du[0];                                                                   \
            wd[2 * d + 1] = (T)du[1];                                                               \
            if (ACT) {                                                                              \
                av = av * dm[d];                                                                    \
                wa[2 * d] = (T)av[0];                                                               \
                wa[2 * d + 1] = (T)av[1];                                                           \
            }                                                                                       \
        }                                                                                           \
        rd[k] = __builtin_bit_cast(uint4, wd);                                                      \
        if (ACT) *reinterpret_cast<uint4*>(ba + (od + k * sd)) = __builtin_bit_cast(uint4, wa);     \
        __builtin_amdgcn_sched_barrier(0); /* one vector at a time: interleaved by the scheduler they do not fit 128 registers */ \
    }
    REGS_PHASE1(SILU, ACT)
#undef REGS_PHASE1
    // phase 2's per-channel constants are requested here, to land under the reduction
    f32x2 mu[4], rs[4], gm[4];
#pragma unroll
    for (int d = 0; d < 4; ++d) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int c = c0 + 2 * d + h, g = c / cg;
            mu[d][h] = mean_rstd[((int64_t)n * groups + g) * 2];
            rs[d][h] = mean_rstd[((int64_t)n * groups + g) * 2 + 1];
            gm[d][h] = gamma[c];
        }
    }
#pragma unroll
    for (int d = 0; d < 4; ++d) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            red[(tid * EPV + 2 * d + h) * 2] = a[d][h];
            red[(tid * EPV + 2 * d + h) * 2 + 1] = bq[d][h];
        }
    }
    __syncthreads();
    {   // channel lc of the slice: its ppw per-thread partials in NT / Cw parts of ppw * Cw / NT rows each, then the parts in order
        const int lc = tid % Cw, part = tid / Cw, nparts = NT / Cw, rpp = ppw / nparts;
        const int v = lc / EPV, j = lc % EPV;
        float sa = 0.f, sb = 0.f;
        for (int r = part * rpp; r < (part + 1) * rpp; ++r) {
            sa += red[((r * VPP + v) * EPV + j) * 2];
            sb += red[((r * VPP + v) * EPV + j) * 2 + 1];
        }
        red2[(part * Cw + lc) * 2] = sa;
        red2[(part * Cw + lc) * 2 + 1] = sb;
        __syncthreads();
        if (tid < Cw) {
            sa = 0.f;
            sb = 0.f;
            for (int q = 0; q < nparts; ++q) {
                sa += red2[(q * Cw + tid) * 2];
                sb += red2[(q * Cw + tid) * 2 + 1];
            }
            const float* mr = mean_rstd + ((int64_t)n * groups + (cb + tid) / cg) * 2;
            sb = mr[1] * fmaf(-mr[0], sa, sb);  // sum du * xhat
            chA[tid] = sa;
            chB[tid] = sb;
            if (rows) {
                rows[(int64_t)n * C + cb + tid] = sa;
                rows[((int64_t)gridDim.x + n) * C + cb + tid] = sb;
            } else {
                atomicAdd(&dbeta[cb + tid], sa);
                atomicAdd(&dgamma[cb + tid], sb);
            }
        }
    }
    __syncthreads();
    if (tid < Cw / cg) {  // the slice's groups
        float s1 = 0.f, s2 = 0.f;
        for (int j = 0; j < cg; ++j) {
            const int lc = tid * cg + j;
            s1 = fmaf(gamma[cb + lc], chA[lc], s1);
            s2 = fmaf(gamma[cb + lc], chB[lc], s2);
        }
        gS1[tid] = s1;
        gS2[tid] = s2;
    }
    __syncthreads();
    const float inv = 1.0f / (float)((int64_t)cg * HW);
    // dx = rs (du gm - (k1 + xhat k2)) = du G + x Cx + C0 with the du of phase 1 (mask and silu' inside)
    f32x2 G[4], Cx[4], C0[4];
#pragma unroll
    for (int d = 0; d < 4; ++d) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int g = (c0 - cb + 2 * d + h) / cg;
            const float k1 = gS1[g] * inv, k2 = gS2[g] * inv, r = rs[d][h];
            G[d][h] = r * gm[d][h];
            Cx[d][h] = -r * r * k2;
            C0[d][h] = r * (mu[d][h] * r * k2 - k1);
        }
    }
#define REGS_PHASE2(ACC, EXTRA)                                                                    \
    _Pragma("unroll") for (int k0 = 0; k0 < ITERS; k0 += PB) {                                      \
        uint4 ro[PB], re[PB];                                                                       \
        _Pragma("unroll") for (int u = 0; u < PB; ++u) {                                            \
            if (ACC) ro[u] = acc ? *reinterpret_cast<const uint4*>(bo + (ox + (k0 + u) * sx)) : make_uint4(0u, 0u, 0u, 0u); /* (per source) */ \
            if (EXTRA) re[u] = *reinterpret_cast<const uint4*>(be + (ox + (k0 + u) * sx));          \
        }                                                                                           \
        _Pragma("unroll") for (int u = 0; u < PB; ++u) {                                            \
            const int k = k0 + u;                                                                   \
            const V8 vd = __builtin_bit_cast(V8, rd[k]), vx = __builtin_bit_cast(V8, rx[k]);        \
            V8 vo, ve, w;                                                                           \
            if (ACC) vo = __builtin_bit_cast(V8, ro[u]);                                            \
            if (EXTRA) ve = __builtin_bit_cast(V8, re[u]);                                          \
            _Pragma("unroll") for (int d = 0; d < 4; ++d) {                                         \
                const f32x2 du = f32x2{(float)vd[2 * d], (float)vd[2 * d + 1]}, xx = f32x2{(float)vx[2 * d], (float)vx[2 * d + 1]}; \
                f32x2 dx = __builtin_elementwise_fma(xx, Cx[d], __builtin_elementwise_fma(du, G[d], C0[d])); \
                if (EXTRA) dx = dx + f32x2{(float)ve[2 * d], (float)ve[2 * d + 1]};                 \
                if (ACC) dx = dx + f32x2{(float)vo[2 * d], (float)vo[2 * d + 1]};                   \
                w[2 * d] = (T)dx[0];                                                                \
                w[2 * d + 1] = (T)dx[1];                                                            \
            }                                                                                       \
            *reinterpret_cast<uint4*>(bo + (ox + k * sx)) = __builtin_bit_cast(uint4, w);           \
            __builtin_amdgcn_sched_barrier(0);                                                      \
        }                                                                                           \
    }
    constexpr int PB = ITERS >= 2 ? 2 : 1;  // phase 2 takes the vectors in pairs (the addends' loads of a pair in flight together)
    static_assert(ITERS % PB == 0, "whole pairs");
    REGS_PHASE2(ACC, EXTRA)
#undef REGS_PHASE2
}

// channel slices for gn_bwd_regs_kernel (0: the shape does not fit it): whole groups, whole 16-byte vectors, not straddling the two
// concatenated sources, a power-of-two vector count per pixel, at most 8 pixels per thread of the 512
// Configurations of gn_bwd_regs_kernel, in order of preference: (vectors per thread, threads).  1024 x 4 is the 32x32 / 16x16 form
// (one workgroup of 16 waves per CU); 1024 x 2 the same on tensors whose 1024 x 4 slicing leaves CUs without a workgroup (16x16 maps
// of 128 channels: 128 workgroups); 256 x 2 / 256 x 1 the 8x8 and 4x4 maps.
struct GnRegsCfg {
    int iters, nt;
};
static const GnRegsCfg GN_REGS_CFGS[] = {{4, 1024}, {2, 1024}, {2, 256}, {1, 256}};
// channel slices of configuration `cf` (0: the shape does not fit it): whole groups, whole 16-byte vectors, not straddling the two
// concatenated sources, a power-of-two vector count per pixel, exactly cf.iters pixels per thread
static int gn_bwd_regs_slices_cfg(GnRegsCfg cf, int N, int HW, int C1, int C2, int groups) {
    const int C = C1 + C2, cgs = C / groups, epv = 8;
    if (C % groups || C1 % epv || C2 % epv) return 0;
    for (int cand = 1; cand <= 16; cand <<= 1) {
        const int w = C / cand;
        if (C % cand || w % cgs || w % epv || C1 % w || w > 128 || w / cgs > 32) continue;
        const int vpp = w / epv;
        if ((vpp & (vpp - 1)) || cf.nt % vpp) continue;
        const int ppw = cf.nt / vpp;
        if (HW != cf.iters * ppw || (int64_t)N * cand > 65535 || (int64_t)HW * C * 2 >= (1ll << 31)) continue;
        return cand;
    }
    return 0;
}
// the configuration for a shape (-1: none) and its slices: the first that gives every CU a workgroup, else the first that fits
static int gn_bwd_regs_pick(int dtype, int N, int HW, int C1, int C2, int groups, int* slices) {
    if (getenv("DMME_NO_GN_BWD_REGS") || !is16(dtype)) return -1;
    int first = -1, first_slices = 0;
    for (int i = 0; i < (int)(sizeof(GN_REGS_CFGS) / sizeof(GN_REGS_CFGS[0])); ++i) {
        const int sl = gn_bwd_regs_slices_cfg(GN_REGS_CFGS[i], N, HW, C1, C2, groups);
        if (!sl) continue;
        if ((int64_t)N * sl >= 256) {
            *slices = sl;
            return i;
        }
        if (first < 0) {
            first = i;
            first_slices = sl;
        }
    }
    *slices = first_slices;
    return first;
}
static int gn_bwd_regs_slices(int dtype, int N, int HW, int C1, int C2, int groups) {
    int sl = 0;
    return gn_bwd_regs_pick(dtype, N, HW, C1, C2, groups, &sl) < 0 ? 0 : sl;
}

static bool gn_bwd_small_supported(int dtype, int HW, int C1, int C2, int groups) {
    const int EPV = is16(dtype) ? 8 : 4, C = C1 + C2;
    return HW <= 64 && C <= 512 && groups <= 64 && C % groups == 0 && C1 % EPV == 0 && C2 % EPV == 0 && C / EPV <= 256 && !debug_route("no_gn_small");
}

// pixel chunks of the two-pass GroupNorm backward = partial rows of its channel-sum scratch ([chunks][N][C][2] floats); 1 when the
// one-workgroup-per-image kernel or the generic path serves the shape
int gn_bwd_fast_chunks(int dtype, int HW, int C) {
    int cp, nc, pw;
    if (!vec_geometry(dtype, HW, C, cp, nc, pw)) return 1;
    return nc;
}

bool gn_bwd_fast_supported(int dtype, int HW, int C1, int C2) {
    const int EPV = is16(dtype) ? 8 : 4;
    int a, b, c;
    return (C1 % EPV) == 0 && vec_geometry(dtype, HW, C1 + C2, a, b, c);
}

// will launch_gn_bwd_fast leave the batch sums of dgamma / dbeta as per-image rows (instead of same-address atomics)?  Mirrors its dispatch.
bool gn_bwd_rows_supported(int dtype, int HW, int C1, int C2, int groups, bool has_mod) {
    if (debug_route("no_gn_bwd_rows")) return false;
    if (!has_mod && gn_bwd_small_supported(dtype, HW, C1, C2, groups)) return true;
    if (!has_mod && gn_bwd_regs_slices(dtype, 1, HW, C1, C2, groups)) return true;
    const int C = C1 + C2;
    return !debug_route("no_gn_bwd_image") && !debug_route("no_gn_bwd_fused_fin") && C <= 1024 && groups <= 256;
}

// AB: gn_bwd_fast_chunks * N*C*2 floats (every entry is written: no zeroing needed);  S: N*groups*2 floats of scratch
int launch_gn_bwd_fast(int dtype, const void* dv, const void* x1, const void* x2, int N, int HW, int C1, int C2, int groups,
                       const float* gamma, const float* mean_rstd, const float* scale, const float* shift, const float* dmask,
                       int pro_silu, void* dx1, void* dx2, int acc1, int acc2, float* dgamma, float* dbeta, float* AB, float* S, GnMod mod,
                       hipStream_t s, void* act, float* rows, const void* extra) {
    if (rows && !gn_bwd_rows_supported(dtype, HW, C1, C2, groups, mod.t_scale != nullptr)) rows = nullptr;  // (the plan asked the same question)
    int rslices = 0;
    if (const int rcfg = mod.t_scale ? -1 : gn_bwd_regs_pick(dtype, N, HW, C1, C2, groups, &rslices); rcfg >= 0) {
        // the four uniform switches of the two phases are template arguments: as branches inside one kernel the variants of a phase
        // shared one register allocation and spilled (240 bytes per lane against none)
#define REGS_LAUNCH_T(TT, I_, NT_, S_, A_, C_, E_)                                                                                              \
    hipLaunchKernelGGL((gn_bwd_regs_kernel<TT, I_, NT_, S_, A_, C_, E_>), dim3(N, rslices), dim3(NT_), 0, s, (const TT*)dv, (const TT*)x1,          \
                       (const TT*)x2, HW, C1, C2, groups, gamma, mean_rstd, scale, shift, dmask, pro_silu, (TT*)dx1, (TT*)dx2, acc1, acc2,          \
                       dgamma, dbeta, (TT*)act, rows, (const TT*)extra)
#define REGS_LAUNCH(I_, NT_, S_, A_, C_, E_)                              \
    do {                                                                  \
        if (dtype == DMME_F16) REGS_LAUNCH_T(f16, I_, NT_, S_, A_, C_, E_); \
        else REGS_LAUNCH_T(bf16, I_, NT_, S_, A_, C_, E_);                \
    } while (0)
#define REGS_LAUNCH_CE(I_, NT_, S_, A_)                              \
    do {                                                             \
        if (acc1 || acc2) {                                          \
            if (extra) REGS_LAUNCH(I_, NT_, S_, A_, true, true);     \
            else REGS_LAUNCH(I_, NT_, S_, A_, true, false);          \
        } else {                                                     \
            if (extra) REGS_LAUNCH(I_, NT_, S_, A_, false, true);    \
            else REGS_LAUNCH(I_, NT_, S_, A_, false, false);         \
        }                                                            \
    } while (0)
#define REGS_LAUNCH_SA(I_, NT_)                                \
    do {                                                       \
        if (pro_silu) {                                        \
            if (act) REGS_LAUNCH_CE(I_, NT_, true, true);      \
            else REGS_LAUNCH_CE(I_, NT_, true, false);         \
        } else {                                               \
            if (act) REGS_LAUNCH_CE(I_, NT_, false, true);     \
            else REGS_LAUNCH_CE(I_, NT_, false, false);        \
        }                                                      \
    } while (0)
        switch (rcfg) {
            case 0: REGS_LAUNCH_SA(4, 1024); break;
            case 1: REGS_LAUNCH_SA(2, 1024); break;
            case 2: REGS_LAUNCH_SA(2, 256); break;
            default: REGS_LAUNCH_SA(1, 256); break;
        }
#undef REGS_LAUNCH_SA
#undef REGS_LAUNCH_CE
#undef REGS_LAUNCH
        DMME_CHECK_LAUNCH();
        return DMME_OK;
    }
    if (!mod.t_scale && gn_bwd_small_supported(dtype, HW, C1, C2, groups)) {
        // channel slices: whole groups, whole 16-byte vectors, not straddling the two concatenated sources, <= 256 threads per pixel row
        const int Call = C1 + C2, cgs = Call / groups, epv = is16(dtype) ? 8 : 4;
        int slices = 1;
        const bool slice_off = (debug_route("no_gn_bwd_slices") != 0);
        for (int cand = 4; cand >= 2 && !slice_off; cand >>= 1) {
            const int w = Call / cand;
            if (Call % cand == 0 && w % cgs == 0 && w % epv == 0 && C1 % w == 0 && (int64_t)N * cand <= 1024) {
                slices = cand;
                break;
            }
        }
        if (dtype == DMME_BF16)
            hipLaunchKernelGGL(gn_bwd_small_kernel<bf16>, dim3(N, slices), dim3(256), 0, s, (const bf16*)dv, (const bf16*)x1, (const bf16*)x2, HW, C1, C2, groups,
                               gamma, mean_rstd, scale, shift, dmask, pro_silu, (bf16*)dx1, (bf16*)dx2, acc1, acc2, dgamma, dbeta, (bf16*)act, rows, (const bf16*)extra);
        else if (dtype == DMME_F16)
            hipLaunchKernelGGL(gn_bwd_small_kernel<f16>, dim3(N, slices), dim3(256), 0, s, (const f16*)dv, (const f16*)x1, (const f16*)x2, HW, C1, C2, groups,
                               gamma, mean_rstd, scale, shift, dmask, pro_silu, (f16*)dx1, (f16*)dx2, acc1, acc2, dgamma, dbeta, (f16*)act, rows, (const f16*)extra);
        else
            hipLaunchKernelGGL(gn_bwd_small_kernel<float>, dim3(N, slices), dim3(256), 0, s, (const float*)dv, (const float*)x1, (const float*)x2, HW, C1, C2,
                               groups, gamma, mean_rstd, scale, shift, dmask, pro_silu, (float*)dx1, (float*)dx2, acc1, acc2, dgamma, dbeta, (float*)act, rows, (const float*)extra);
        DMME_CHECK_LAUNCH();
        return DMME_OK;
    }
    int chunk_px, nchunks, ppw;
    const int C = C1 + C2;
    DMME_REQUIRE(vec_geometry(dtype, HW, C, chunk_px, nchunks, ppw), DMME_ERR_UNSUPPORTED, "gn_bwd_fast: unsupported geometry");
    dim3 grid(nchunks, N);
    if (dtype == DMME_BF16)
        hipLaunchKernelGGL(gn_bwd_sums_kernel<bf16>, grid, dim3(256), 0, s, (const bf16*)dv, (const bf16*)x1, (const bf16*)x2, HW, C1, C2, groups,
                           mean_rstd, scale, shift, dmask, pro_silu, chunk_px, ppw, AB);
    else if (dtype == DMME_F16)
        hipLaunchKernelGGL(gn_bwd_sums_kernel<f16>, grid, dim3(256), 0, s, (const f16*)dv, (const f16*)x1, (const f16*)x2, HW, C1, C2, groups,
                           mean_rstd, scale, shift, dmask, pro_silu, chunk_px, ppw, AB);
    else
        hipLaunchKernelGGL(gn_bwd_sums_kernel<float>, grid, dim3(256), 0, s, (const float*)dv, (const float*)x1, (const float*)x2, HW, C1, C2,
                           groups, mean_rstd, scale, shift, dmask, pro_silu, chunk_px, ppw, AB);
    DMME_CHECK_LAUNCH();
    const bool per_image_off = (debug_route("no_gn_bwd_image") != 0);
    const bool fused_off = (debug_route("no_gn_bwd_fused_fin") != 0);
    const bool fused_fin = !per_image_off && !fused_off && C <= 1024 && groups <= 256;  // the apply kernel merges the chunk sums itself
    if (fused_fin) {
    } else if (!per_image_off && C <= 1024 && groups <= 256) {
        hipLaunchKernelGGL(gn_bwd_finalize_image_kernel, dim3(N), dim3(256), 0, s, AB, nchunks, N, C, groups, gamma, S, dgamma, dbeta, mod);
    } else {
        const int nb_s = (N * groups + 255) / 256;
        hipLaunchKernelGGL(gn_bwd_finalize_kernel, dim3(nb_s + (C + 31) / 32), dim3(256), 0, s, AB, nchunks, N, C, groups, nb_s, gamma, S, dgamma, dbeta, mod);
    }
    DMME_CHECK_LAUNCH();
    if (dtype == DMME_BF16)
        hipLaunchKernelGGL(gn_bwd_apply_kernel<bf16>, grid, dim3(256), 0, s, (const bf16*)dv, (const bf16*)x1, (const bf16*)x2, HW, C1, C2, groups,
                           gamma, mean_rstd, scale, shift, dmask, pro_silu, S, chunk_px, ppw, (bf16*)dx1, (bf16*)dx2, acc1, acc2, mod, (bf16*)act,
                           fused_fin ? AB : nullptr, dgamma, dbeta, fused_fin ? rows : nullptr, (const bf16*)extra);
    else if (dtype == DMME_F16)
        hipLaunchKernelGGL(gn_bwd_apply_kernel<f16>, grid, dim3(256), 0, s, (const f16*)dv, (const f16*)x1, (const f16*)x2, HW, C1, C2, groups,
                           gamma, mean_rstd, scale, shift, dmask, pro_silu, S, chunk_px, ppw, (f16*)dx1, (f16*)dx2, acc1, acc2, mod, (f16*)act,
                           fused_fin ? AB : nullptr, dgamma, dbeta, fused_fin ? rows : nullptr, (const f16*)extra);
    else
        hipLaunchKernelGGL(gn_bwd_apply_kernel<float>, grid, dim3(256), 0, s, (const float*)dv, (const float*)x1, (const float*)x2, HW, C1, C2,
                           groups, gamma, mean_rstd, scale, shift, dmask, pro_silu, S, chunk_px, ppw, (float*)dx1, (float*)dx2, acc1, acc2, mod, (float*)act,
                           fused_fin ? AB : nullptr, dgamma, dbeta, fused_fin ? rows : nullptr, (const float*)extra);
    DMME_CHECK_LAUNCH();
    return DMME_OK;
}

}  // namespace dmme
