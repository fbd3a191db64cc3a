// Elementwise kernels of the diffusion process: forward noising, DDPM / DDIM updates,
// MSE loss, Philox normals / Dropout2d multipliers.  All HBM-bound, fp32, NCHW.
// The translation unit is compiled with -ffp-contract=off (csrc/Makefile) so each
// expression rounds exactly like the reference's sequence of separate torch ops
// (mul then add, never fma); square roots come from host tables.
#include "common.h"

namespace dmme {

// ------------------------------------------------------------------ Philox4x32-10
struct Philox {
    uint32_t c[4];
    uint32_t k[2];
};
__device__ __forceinline__ void philox_round(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
    const uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0;
    const uint32_t n1 = (uint32_t)p1;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
    const uint32_t n3 = (uint32_t)p0;
    c[0] = n0;
    c[1] = n1;
    c[2] = n2;
    c[3] = n3;
}
__device__ __forceinline__ void philox4x32_10(uint64_t seed, uint64_t ctr, uint32_t (&out)[4]) {
    uint32_t c[4] = {(uint32_t)ctr, (uint32_t)(ctr >> 32), 0u, 0u};
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        philox_round(c, k0, k1);
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c[0];
    out[1] = c[1];
    out[2] = c[2];
    out[3] = c[3];
}
__device__ __forceinline__ float u01(uint32_t x) {  // (0, 1]
    return ((float)(x >> 8) + 1.0f) * (1.0f / 16777216.0f);
}
__device__ __forceinline__ void normal4(uint64_t seed, uint64_t ctr, float (&z)[4]) {
    uint32_t r[4];
    philox4x32_10(seed, ctr, r);
    const float r0 = sqrtf(-2.0f * logf(u01(r[0]))), a0 = 6.283185307179586f * u01(r[1]);
    const float r1 = sqrtf(-2.0f * logf(u01(r[2]))), a1 = 6.283185307179586f * u01(r[3]);
    z[0] = r0 * cosf(a0);
    z[1] = r0 * sinf(a0);
    z[2] = r1 * cosf(a1);
    z[3] = r1 * sinf(a1);
}

__global__ void __launch_bounds__(256) randn_kernel(float* out, int64_t numel, uint64_t seed, uint64_t offset) {
    const int64_t quads = (numel + 3) / 4;
    for (int64_t q = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; q < quads; q += (int64_t)gridDim.x * blockDim.x) {
        float z[4];
        normal4(seed, offset + (uint64_t)q, z);
        const int64_t b = q * 4;
        if (b + 3 < numel) {
            *reinterpret_cast<float4*>(out + b) = make_float4(z[0], z[1], z[2], z[3]);
        } else {
            for (int j = 0; j < 4 && b + j < numel; ++j) out[b + j] = z[j];
        }
    }
}

static inline unsigned grid_for(int64_t work) {
    int64_t b = (work + 255) / 256;
    return (unsigned)(b > 2048 ? 2048 : (b < 1 ? 1 : b));
}

int launch_randn(float* out, int64_t numel, uint64_t seed, uint64_t offset, hipStream_t s) {
    if (numel <= 0) return DMME_OK;
    hipLaunchKernelGGL(randn_kernel, dim3(grid_for((numel + 3) / 4)), dim3(256), 0, s, out, numel, seed, offset);
    DMME_CHECK_LAUNCH();
    return DMME_OK;
}

// Dropout2d multipliers: 0 with probability p, else 1/(1-p)  (nn.Dropout2d, models/ddpm.py:29)
__global__ void __launch_bounds__(256) dropmask_kernel(float* out, int64_t numel, float p, uint64_t seed, uint64_t offset) {
    const int64_t quads = (numel + 3) / 4;
    const float keep = 1.0f / (1.0f - p);
    for (int64_t q = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; q < quads; q += (int64_t)gridDim.x * blockDim.x) {
        uint32_t r[4];
        philox4x32_10(seed, offset + (uint64_t)q, r);
        for (int j = 0; j < 4 && q * 4 + j < numel; ++j) out[q * 4 + j] = (u01(r[j]) <= p) ? 0.0f : keep;
    }
}
int launch_dropmask(float* out, int64_t numel, float p, uint64_t seed, uint64_t offset, hipStream_t s) {
    if (numel <= 0) return DMME_OK;
    hipLaunchKernelGGL(dropmask_kernel, dim3(grid_for((numel + 3) / 4)), dim3(256), 0, s, out, numel, p, seed, offset);
    DMME_CHECK_LAUNCH();
    return DMME_OK;
}

// ------------------------------------------------------------------ forward noising
// forward_process + Normal.sample + target re-derivation (see dmme_hip.h)
__global__ void __launch_bounds__(256) q_sample_kernel(const float* __restrict__ x0, const float* __restrict__ z,
                                                       const float* __restrict__ sqrt_abar,
                                                       const float* __restrict__ sqrt_1m_abar,
                                                       const int64_t* __restrict__ t, int64_t chw, int64_t total,
                                                       float* __restrict__ x_t, float* __restrict__ target) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t tn = t[i / chw];
        const float sa = sqrt_abar[tn], sd = sqrt_1m_abar[tn];
        const float mean = sa * x0[i];
        const float xt = mean + sd * z[i];
        x_t[i] = xt;
        if (target) target[i] = (xt - mean) / sd;
    }
}
int launch_q_sample(const float* x0, const float* z, const float* sqrt_abar, const float* sqrt_1m_abar,
                    const int64_t* t, int B, int64_t chw, float* x_t, float* target, hipStream_t s) {
    const int64_t total = (int64_t)B * chw;
    if (total <= 0) return DMME_OK;
    hipLaunchKernelGGL(q_sample_kernel, dim3(grid_for(total)), dim3(256), 0, s, x0, z, sqrt_abar, sqrt_1m_abar, t, chw, total,
                       x_t, target);
    DMME_CHECK_LAUNCH();
    return DMME_OK;
}

// ------------------------------------------------------------------ reverse updates
// explicitly rounded ops (never contracted into fma): same rounding sequence as the
// reference's separate torch ops
__device__ __forceinline__ float ddpm_update(float x, float e, float z, float c1, float c2, float sigma, int add_noise) {
    const float m = __fmul_rn(c1, __fsub_rn(x, __fmul_rn(c2, e)));
    return add_noise ? __fadd_rn(m, __fmul_rn(sigma, z)) : m;
}
__global__ void __launch_bounds__(256) ddpm_step_kernel(float* __restrict__ x, const float* __restrict__ eps,
                                                        const float* __restrict__ z, float c1, float c2, float sigma,
                                                        int add_noise, int64_t n4, int64_t numel) {
    for (int64_t q = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; q < n4; q += (int64_t)gridDim.x * blockDim.x) {
        const int64_t b = q * 4;
        if (b + 3 < numel) {
            float4 xv = *reinterpret_cast<const float4*>(x + b);
            const float4 ev = *reinterpret_cast<const float4*>(eps + b);
            float4 zv = make_float4(0.f, 0.f, 0.f, 0.f);
            if (add_noise) zv = *reinterpret_cast<const float4*>(z + b);
            xv.x = ddpm_update(xv.x, ev.x, zv.x, c1, c2, sigma, add_noise);
            xv.y = ddpm_update(xv.y, ev.y, zv.y, c1, c2, sigma, add_noise);
            xv.z = ddpm_update(xv.z, ev.z, zv.z, c1, c2, sigma, add_noise);
            xv.w = ddpm_update(xv.w, ev.w, zv.w, c1, c2, sigma, add_noise);
            *reinterpret_cast<float4*>(x + b) = xv;
        } else {
            for (int64_t i = b; i < numel; ++i) x[i] = ddpm_update(x[i], eps[i], add_noise ? z[i] : 0.f, c1, c2, sigma, add_noise);
        }
    }
}
int launch_ddpm_step(float* x, const float* eps, const float* z, float c1, float c2, float sigma, int add_noise,
                     int64_t numel, hipStream_t s) {
    if (numel <= 0) return DMME_OK;
    const int64_t n4 = (numel + 3) / 4;
    hipLaunchKernelGGL(ddpm_step_kernel, dim3(grid_for(n4)), dim3(256), 0, s, x, eps, z, c1, c2, sigma, add_noise, n4, numel);
    DMME_CHECK_LAUNCH();
    return DMME_OK;
}

__global__ void __launch_bounds__(256) ddim_step_kernel(float* __restrict__ x, const float* __restrict__ eps, float s1,
                                                        float s2, int64_t numel) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < numel; i += (int64_t)gridDim.x * blockDim.x) {
        const float x0_hat = __fdiv_rn(__fsub_rn(x[i], __fmul_rn(s1, eps[i])), s2);
        x[i] = __fmul_rn(s2, x0_hat);
    }
}
int launch_ddim_step(float* x, const float* eps, float s1, float s2, int64_t numel, hipStream_t s) {
    if (numel <= 0) return DMME_OK;
    hipLaunchKernelGGL(ddim_step_kernel, dim3(grid_for(numel)), dim3(256), 0, s, x, eps, s1, s2, numel);
    DMME_CHECK_LAUNCH();
    return DMME_OK;
}

// ------------------------------------------------------------------ MSE loss (+ gradient)
__global__ void __launch_bounds__(256) mse_partial_kernel(const float* __restrict__ eps, const float* __restrict__ target,
                                                          int64_t numel, float* __restrict__ d_eps, float gscale,
                                                          float* __restrict__ partial) {
    __shared__ float red[16];
    float acc = 0.f;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < numel; i += (int64_t)gridDim.x * blockDim.x) {
        const float d = target[i] - eps[i];
        acc += d * d;
        if (d_eps) d_eps[i] = (-2.0f * d) * gscale;
    }
    const float tot = block_sum(acc, red);
    if (threadIdx.x == 0) partial[blockIdx.x] = tot;
}
__global__ void __launch_bounds__(256) mse_final_kernel(const float* partial, int n, float inv_numel, float* loss) {
    __shared__ float red[16];
    float acc = 0.f;
    for (int i = threadIdx.x; i < n; i += blockDim.x) acc += partial[i];
    const float tot = block_sum(acc, red);
    if (threadIdx.x == 0) loss[0] = tot * inv_numel;
}
int launch_mse(const float* eps, const float* target, int64_t numel, float* loss, float* d_eps, float gscale,
               float* scratch, hipStream_t s) {
    DMME_REQUIRE(numel > 0, DMME_ERR_INVALID, "mse: numel must be positive");
    unsigned g = grid_for(numel);
    if (g > 1024) g = 1024;
    hipLaunchKernelGGL(mse_partial_kernel, dim3(g), dim3(256), 0, s, eps, target, numel, d_eps,
                       gscale / (float)numel, scratch);
    DMME_CHECK_LAUNCH();
    hipLaunchKernelGGL(mse_final_kernel, dim3(1), dim3(256), 0, s, scratch, (int)g, 1.0f / (float)numel, loss);
    DMME_CHECK_LAUNCH();
    return DMME_OK;
}

}  // namespace dmme
