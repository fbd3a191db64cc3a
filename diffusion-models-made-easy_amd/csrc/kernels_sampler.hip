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

// ------------------------------------------------------------------ Improved DDPM (learned variance)
// network output (B, 2C, H, W): channels [0, C) = eps, [C, 2C) = v (diffusion_models/iddpm.py:159).
// Sigma = exp(v log beta_t + (1 - v) log max(beta~_t, 1e-12)) (equations/iddpm/losses.py:34-37).
__device__ __forceinline__ float iddpm_std(float v, float log_beta, float log_beta_tilde) {
    return sqrtf(expf(__fadd_rn(__fmul_rn(v, log_beta), __fmul_rn(__fsub_rn(1.0f, v), log_beta_tilde))));
}
__global__ void __launch_bounds__(256) iddpm_step_kernel(float* __restrict__ x, const float* __restrict__ out, const float* __restrict__ z,
                                                         float c1, float c2, float log_beta, float log_beta_tilde, int add_noise,
                                                         int64_t chw, int64_t total) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t b = i / chw, r = i - b * chw;
        const float e = out[b * 2 * chw + r];
        const float m = __fmul_rn(c1, __fsub_rn(x[i], __fmul_rn(c2, e)));
        if (add_noise) {
            const float sd = iddpm_std(out[b * 2 * chw + chw + r], log_beta, log_beta_tilde);
            x[i] = __fadd_rn(m, __fmul_rn(sd, z[i]));
        } else {
            x[i] = m;
        }
    }
}
int launch_iddpm_step(float* x, const float* out, const float* z, float c1, float c2, float log_beta, float log_beta_tilde,
                      int add_noise, int B, int64_t chw, hipStream_t s) {
    const int64_t total = (int64_t)B * chw;
    if (total <= 0) return DMME_OK;
    hipLaunchKernelGGL(iddpm_step_kernel, dim3(grid_for(total)), dim3(256), 0, s, x, out, z, c1, c2, log_beta, log_beta_tilde, add_noise, chw,
                       total);
    DMME_CHECK_LAUNCH();
    return DMME_OK;
}

// Hybrid / VLB loss and its gradient w.r.t. the raw network output, one pass (diffusion_models/iddpm.py:62-116,
// equations/iddpm/losses.py:9-98).  coef: per-timestep fp32 table, 8 floats per t:
//   [0] 1/sqrt(alpha_t)  [1] beta_t/sqrt(1-abar_t)  [2] log beta_t  [3] log max(beta~_t, 1e-12)
//   [4] sqrt(abar_{t-1}) beta_t/(1-abar_t)  [5] sqrt(alpha_t)(1-abar_{t-1})/(1-abar_t)  [6] sqrt(beta~_t)  [7] unused
// L_vlb uses the predicted noise with a stop-gradient, so d/d(eps) comes from L_simple only and d/d(v) from L_vlb only.
__global__ void __launch_bounds__(256) iddpm_loss_kernel(const float* __restrict__ out, const float* __restrict__ x_t, const float* __restrict__ x_0,
                                                         const float* __restrict__ target, const int64_t* __restrict__ t,
                                                         const float* __restrict__ coef, int64_t chw, int64_t total, float w_simple, float w_vlb,
                                                         float gscale, float* __restrict__ d_out, float* __restrict__ partial) {
    __shared__ float red[16];
    float acc_s = 0.f, acc_v = 0.f;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t b = i / chw, r = i - b * chw;
        const int64_t tt = t[b];
        const float* c = coef + tt * 8;
        const float e = out[b * 2 * chw + r], v = out[b * 2 * chw + chw + r];
        const float xt = x_t[i], x0 = x_0[i];
        const float d = target[i] - e;
        acc_s += d * d;
        const float mean = c[0] * (xt - c[1] * e);
        const float sd = iddpm_std(v, c[2], c[3]);
        const float hl = 0.5f * (c[2] - c[3]);  // d(sd)/dv = sd * hl
        float vlb, dv;
        if (tt == 1) {  // discrete NLL of x_0 in bins of +-1/255 (losses.py:9-20)
            const float ap = (x0 + 1.0f / 255.0f - mean) / sd, am = (x0 - 1.0f / 255.0f - mean) / sd;
            const bool up = x0 < 1.0f, lo = x0 > -1.0f;
            const float Fp = up ? 0.5f * (1.0f + erff(ap * 0.70710678118654752f)) : 1.0f;
            const float Fm = lo ? 0.5f * (1.0f + erff(am * 0.70710678118654752f)) : 0.0f;
            const float prob = Fp - Fm;
            vlb = -logf(fmaxf(prob, 1e-12f));
            const float pp = up ? 0.3989422804014327f * expf(-0.5f * ap * ap) * ap : 0.f;
            const float pm = lo ? 0.3989422804014327f * expf(-0.5f * am * am) * am : 0.f;
            dv = prob >= 1e-12f ? hl * (pp - pm) / prob : 0.f;  // -(1/prob) d(prob)/d(sd) * sd * hl, d Phi(a)/d sd = -phi(a) a / sd
        } else {  // KL(q(x_{t-1} | x_t, x_0) || p_theta) (losses.py:23-31, torch kl_normal_normal)
            const float qm = c[4] * x0 + c[5] * xt;
            const float q = c[6] / sd, ratio = q * q;
            const float u = (qm - mean) / sd, t1 = u * u;
            vlb = 0.5f * (ratio + t1 - 1.0f - logf(ratio));
            dv = hl * (1.0f - ratio - t1);
        }
        acc_v += vlb;
        if (d_out) {
            d_out[b * 2 * chw + r] = (-2.0f * d) * (w_simple * gscale);
            d_out[b * 2 * chw + chw + r] = dv * (w_vlb * gscale);
        }
    }
    const float ts = block_sum(acc_s, red);
    __syncthreads();
    const float tv = block_sum(acc_v, red);
    if (threadIdx.x == 0) {
        partial[2 * blockIdx.x] = ts;
        partial[2 * blockIdx.x + 1] = tv;
    }
}
__global__ void __launch_bounds__(256) iddpm_loss_final_kernel(const float* partial, int n, float inv_numel, float w_simple, float w_vlb,
                                                               float* loss) {
    __shared__ float red[16];
    float a = 0.f, b = 0.f;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        a += partial[2 * i];
        b += partial[2 * i + 1];
    }
    const float ta = block_sum(a, red);
    __syncthreads();
    const float tb = block_sum(b, red);
    if (threadIdx.x == 0) {
        loss[1] = ta * inv_numel;  // L_simple
        loss[2] = tb * inv_numel;  // L_vlb
        loss[0] = w_simple * loss[1] + w_vlb * loss[2];
    }
}
int launch_iddpm_loss(const float* out, const float* x_t, const float* x_0, const float* target, const int64_t* t, const float* coef, int B,
                      int64_t chw, float w_simple, float w_vlb, float* loss, float* d_out, float gscale, float* scratch, hipStream_t s) {
    const int64_t total = (int64_t)B * chw;
    DMME_REQUIRE(total > 0, DMME_ERR_INVALID, "iddpm_loss: empty batch");
    unsigned g = grid_for(total);
    if (g > 512) g = 512;
    hipLaunchKernelGGL(iddpm_loss_kernel, dim3(g), dim3(256), 0, s, out, x_t, x_0, target, t, coef, chw, total, w_simple, w_vlb,
                       gscale / (float)total, d_out, scratch);
    DMME_CHECK_LAUNCH();
    hipLaunchKernelGGL(iddpm_loss_final_kernel, dim3(1), dim3(256), 0, s, scratch, (int)g, 1.0f / (float)total, w_simple, w_vlb, loss);
    DMME_CHECK_LAUNCH();
    return DMME_OK;
}

// ------------------------------------------------------------------ replayable chain step (hipGraph-friendly sampling loops)
// Everything that changes from one denoising step to the next lives in DEVICE memory, so the launch sequence of a step is the
// same every time and can be replayed from one captured graph: the loop state {i, t = t_table[i], Philox offset and seed, ticket}, the
// per-index scalars of the update (`coef[i][0..3]`) and the noise itself (drawn here from the same Philox stream / offsets
// dmme_randn would use: the chain is bit-identical to the eager loop under the same seed).  The block that takes the last
// ticket has, by construction, run after every block read the state, and advances it: i -= 1, t = t_table[i], offset += quads.
struct ChainState {
    long long i;
    long long t;
    unsigned long long offset;
    unsigned long long seed;
    unsigned int ticket, pad;
    unsigned long long reserved[3];
};
static_assert(sizeof(ChainState) == 64, "ChainState is eight 64-bit words (dmme_hip.h: dmme_chain_*)");

__global__ void chain_set_kernel(ChainState* st, long long i, const long long* __restrict__ t_table, unsigned long long seed, unsigned long long offset) {
    st->i = i;
    st->t = t_table[i];
    st->offset = offset;
    st->seed = seed;
    st->ticket = 0u;
    st->pad = 0u;
}

// KIND 0: DDPM (coef = 1/sqrt(alpha), beta/sqrt(1-abar), sqrt(beta)), 1: DDIM (sqrt(1-abar_tau_i), sqrt(abar_tau_{i-1})),
//      2: IDDPM (1/sqrt(alpha), beta/sqrt(1-abar), log beta, log max(beta~, 1e-12)); `out` has 2*chw values per image then
template <int KIND>
__global__ void __launch_bounds__(256) chain_update_kernel(float* __restrict__ x, const float* __restrict__ out, const float* __restrict__ coef,
                                                           const long long* __restrict__ t_table, ChainState* st, int64_t chw, int64_t n4) {
    const long long i = st->i, t = st->t;
    const unsigned long long off = st->offset, seed = st->seed;
    const float c0 = coef[4 * i], c1 = coef[4 * i + 1], c2 = coef[4 * i + 2], c3 = coef[4 * i + 3];
    const int add_noise = t != 1;
    for (int64_t q = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; q < n4; q += (int64_t)gridDim.x * blockDim.x) {
        const int64_t b = q * 4;
        float4 xv = *reinterpret_cast<const float4*>(x + b);
        float* xs = reinterpret_cast<float*>(&xv);
        if (KIND == 1) {
            const float4 ev = *reinterpret_cast<const float4*>(out + b);
            const float* es = reinterpret_cast<const float*>(&ev);
#pragma unroll
            for (int j = 0; j < 4; ++j) xs[j] = __fmul_rn(c1, __fdiv_rn(__fsub_rn(xs[j], __fmul_rn(c0, es[j])), c1));
        } else {
            float z[4] = {0.f, 0.f, 0.f, 0.f};
            if (add_noise) normal4(seed, off + (uint64_t)q, z);  // the reference draws and discards at t == 1: the offset still advances
            if (KIND == 0) {
                const float4 ev = *reinterpret_cast<const float4*>(out + b);
                const float* es = reinterpret_cast<const float*>(&ev);
#pragma unroll
                for (int j = 0; j < 4; ++j) xs[j] = ddpm_update(xs[j], es[j], z[j], c0, c1, c2, add_noise);
            } else {
                const int64_t img = b / chw, r = b - img * chw;  // chw % 4 == 0: a quad never straddles two images
                const float4 ev = *reinterpret_cast<const float4*>(out + img * 2 * chw + r);
                const float4 vv = *reinterpret_cast<const float4*>(out + img * 2 * chw + chw + r);
                const float* es = reinterpret_cast<const float*>(&ev);
                const float* vs = reinterpret_cast<const float*>(&vv);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float m = __fmul_rn(c0, __fsub_rn(xs[j], __fmul_rn(c1, es[j])));
                    xs[j] = add_noise ? __fadd_rn(m, __fmul_rn(iddpm_std(vs[j], c2, c3), z[j])) : m;
                }
            }
        }
        *reinterpret_cast<float4*>(x + b) = xv;
    }
    __syncthreads();  // every thread of this block has read the state (and is past its loads of it)
    if (threadIdx.x == 0) {
        const unsigned tk = atomicAdd(&st->ticket, 1u);
        if (tk == gridDim.x - 1) {  // last block: all others took their ticket after reading the state
            const long long ni = i > 0 ? i - 1 : 0;
            st->i = ni;
            st->t = t_table[ni];
            st->offset = off + (unsigned long long)n4;
            atomicExch(&st->ticket, 0u);
        }
    }
}

int launch_chain_set(void* state, int64_t i, const int64_t* t_table, uint64_t seed, uint64_t offset, hipStream_t s) {
    hipLaunchKernelGGL(chain_set_kernel, dim3(1), dim3(1), 0, s, (ChainState*)state, (long long)i, (const long long*)t_table, (unsigned long long)seed,
                       (unsigned long long)offset);
    DMME_CHECK_LAUNCH();
    return DMME_OK;
}

int launch_chain_update(int kind, float* x, const float* out, const float* coef, const int64_t* t_table, void* state, int B, int64_t chw,
                        hipStream_t s) {
    DMME_REQUIRE(kind >= 0 && kind <= 2, DMME_ERR_INVALID, "chain_update: unknown sampler kind %d", kind);
    DMME_REQUIRE(chw % 4 == 0, DMME_ERR_UNSUPPORTED, "chain_update: image size %lld is not a multiple of 4", (long long)chw);
    const int64_t n4 = (int64_t)B * chw / 4;
    if (n4 <= 0) return DMME_OK;
    const dim3 g(grid_for(n4)), b(256);
    ChainState* st = (ChainState*)state;
    const long long* tt = (const long long*)t_table;
    if (kind == 0)
        hipLaunchKernelGGL(chain_update_kernel<0>, g, b, 0, s, x, out, coef, tt, st, chw, n4);
    else if (kind == 1)
        hipLaunchKernelGGL(chain_update_kernel<1>, g, b, 0, s, x, out, coef, tt, st, chw, n4);
    else
        hipLaunchKernelGGL(chain_update_kernel<2>, g, b, 0, s, x, out, coef, tt, st, chw, n4);
    DMME_CHECK_LAUNCH();
    return DMME_OK;
}

// ------------------------------------------------------------------ input pipeline: HBM-resident uint8 dataset -> training batch
// out[b][c][y][x] = norm(ToTensor(flip_b(data[idx[b]])))[c][y][x]: torchvision's ToTensor (uint8 -> float / 255) followed by the
// reference's norm, (x - 0.5) * 2 (src/dmme/common/norm.py:4-6), and RandomHorizontalFlip as a per-image bit
// (data_modules/cifar10.py:33-44).  data: [n_images][C][H][W] uint8 (the CIFAR10 pickle's own layout).  One thread per 4 pixels.
__global__ void __launch_bounds__(256) image_batch_kernel(const uint8_t* __restrict__ data, const int64_t* __restrict__ idx,
                                                          const uint8_t* __restrict__ flip, int C, int H, int W, int64_t total4,
                                                          float* __restrict__ out) {
    const int W4 = W / 4;
    for (int64_t q = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; q < total4; q += (int64_t)gridDim.x * blockDim.x) {
        const int x4 = (int)(q % W4);
        int64_t r = q / W4;
        const int y = (int)(r % H);
        r /= H;
        const int c = (int)(r % C), b = (int)(r / C);
        const uint8_t* row = data + ((idx[b] * C + c) * H + y) * (int64_t)W;
        const bool fl = flip && flip[b];
        const uint32_t raw = *reinterpret_cast<const uint32_t*>(row + (fl ? W - 4 - 4 * x4 : 4 * x4));
        float4 o;
        float* ov = reinterpret_cast<float*>(&o);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint32_t u = (raw >> (8 * (fl ? 3 - j : j))) & 0xffu;
            ov[j] = __fmul_rn(__fsub_rn(__fdiv_rn((float)u, 255.0f), 0.5f), 2.0f);
        }
        *reinterpret_cast<float4*>(out + (((int64_t)b * C + c) * H + y) * W + 4 * x4) = o;
    }
}
int launch_image_batch(const uint8_t* data, const int64_t* idx, const uint8_t* flip, int B, int C, int H, int W, float* out, hipStream_t s) {
    DMME_REQUIRE(W % 4 == 0, DMME_ERR_UNSUPPORTED, "image_batch: width %d is not a multiple of 4", W);
    const int64_t total4 = (int64_t)B * C * H * (W / 4);
    if (total4 <= 0) return DMME_OK;
    hipLaunchKernelGGL(image_batch_kernel, dim3(grid_for(total4)), dim3(256), 0, s, data, idx, flip, C, H, W, total4, out);
    DMME_CHECK_LAUNCH();
    return DMME_OK;
}

}  // namespace dmme
