// GroupNorm statistics for NHWC tensors, HBM-bound: every element is read exactly once
// with 16-byte coalesced loads and kept in registers for a numerically robust local
// two-pass (mean, then centred second moment).  Per-workgroup partials (mean, M2) are
// merged with Chan's parallel-variance formula in a tiny finalize kernel that also
// folds gamma/beta into per-(n, channel) scale/shift vectors for the consumer conv.
//
// Replaces the statistics half of nn.GroupNorm (models/ddpm.py:17-18); the apply half
// is fused into the prologue of the convolution that consumes it (conv_mfma.hip).
#include <stdlib.h>

#include "conv_common.h"

namespace dmme {

constexpr int GN_MAX_SWEEPS = 8;

template <typename T>
__device__ __forceinline__ void load_vec_gn(const T* p, float (&v)[16 / sizeof(T)]) {
    const uint4 raw = *reinterpret_cast<const uint4*>(p);
    if constexpr (sizeof(T) == 4) {
        v[0] = __uint_as_float(raw.x); v[1] = __uint_as_float(raw.y); v[2] = __uint_as_float(raw.z); v[3] = __uint_as_float(raw.w);
    } else if constexpr (__is_same(T, f16)) {
        unpack8<f16>(raw, v);
    } else {
        v[0] = __uint_as_float(raw.x << 16); v[1] = __uint_as_float(raw.x & 0xffff0000u);
        v[2] = __uint_as_float(raw.y << 16); v[3] = __uint_as_float(raw.y & 0xffff0000u);
        v[4] = __uint_as_float(raw.z << 16); v[5] = __uint_as_float(raw.z & 0xffff0000u);
        v[6] = __uint_as_float(raw.w << 16); v[7] = __uint_as_float(raw.w & 0xffff0000u);
    }
}

template <typename T>
__global__ void __launch_bounds__(256) gn_partial_kernel(const T* __restrict__ s1, const T* __restrict__ s2, int HW,
                                                         int C1, int C2, int groups, int chunk_px, int nsweeps,
                                                         float* __restrict__ partial) {
    constexpr int EPV = 16 / sizeof(T);
    __shared__ float red[256 * EPV];
    __shared__ float gstat[256];
    const int tid = threadIdx.x;
    const int C = C1 + C2, VPP = C / EPV, cg = C / groups;
    const int slot = tid % VPP, prow = tid / VPP, ppw = 256 / VPP;
    const int c0 = slot * EPV;
    const bool second = c0 >= C1;
    const T* src = second ? s2 : s1;
    const int Cs = second ? C2 : C1, cs0 = second ? c0 - C1 : c0;
    const int n = blockIdx.y, chunk = blockIdx.x, nchunks = gridDim.x;
    const int64_t p0 = (int64_t)n * HW + (int64_t)chunk * chunk_px;

    float v[GN_MAX_SWEEPS][EPV];
#pragma unroll
    for (int sw = 0; sw < GN_MAX_SWEEPS; ++sw) {
        if (sw < nsweeps) {
            const uint4 raw = *reinterpret_cast<const uint4*>(src + (p0 + sw * ppw + prow) * Cs + cs0);
            if constexpr (sizeof(T) == 4) {
                const float4 f = __builtin_bit_cast(float4, raw);
                v[sw][0] = f.x; v[sw][1] = f.y; v[sw][2] = f.z; v[sw][3] = f.w;
            } else {
                typedef T tx8 __attribute__((ext_vector_type(8)));  // bf16 or IEEE half
                const tx8 b = __builtin_bit_cast(tx8, raw);
#pragma unroll
                for (int j = 0; j < EPV; ++j) v[sw][j] = (float)b[j];
            }
        } else {
#pragma unroll
            for (int j = 0; j < EPV; ++j) v[sw][j] = 0.f;
        }
    }
    const float inv_cnt = 1.0f / (float)(chunk_px * cg);

    // ---- pass 1: local mean per group ----
    float e[EPV];
#pragma unroll
    for (int j = 0; j < EPV; ++j) {
        float s = 0.f;
#pragma unroll
        for (int sw = 0; sw < GN_MAX_SWEEPS; ++sw) s += v[sw][j];
        e[j] = s;
    }
#pragma unroll
    for (int j = 0; j < EPV; ++j) red[tid * EPV + j] = e[j];
    __syncthreads();
    if (tid < groups) {
        float acc = 0.f;
        for (int c = tid * cg; c < (tid + 1) * cg; ++c) {
            const int sl = c / EPV, j = c % EPV;
            for (int pr = 0; pr < ppw; ++pr) acc += red[(pr * VPP + sl) * EPV + j];
        }
        gstat[tid] = acc * inv_cnt;
    }
    __syncthreads();
    // ---- pass 2: centred second moment ----
#pragma unroll
    for (int j = 0; j < EPV; ++j) {
        const float m = gstat[(c0 + j) / cg];
        float s = 0.f;
#pragma unroll
        for (int sw = 0; sw < GN_MAX_SWEEPS; ++sw) {
            if (sw < nsweeps) {
                const float d = v[sw][j] - m;
                s = fmaf(d, d, s);
            }
        }
        e[j] = s;
    }
#pragma unroll
    for (int j = 0; j < EPV; ++j) red[tid * EPV + j] = e[j];
    __syncthreads();
    if (tid < groups) {
        float acc = 0.f;
        for (int c = tid * cg; c < (tid + 1) * cg; ++c) {
            const int sl = c / EPV, j = c % EPV;
            for (int pr = 0; pr < ppw; ++pr) acc += red[(pr * VPP + sl) * EPV + j];
        }
        float* o = partial + (((int64_t)n * nchunks + chunk) * groups + tid) * 2;
        o[0] = gstat[tid];
        o[1] = acc;
    }
}

__global__ void __launch_bounds__(256) gn_finalize_kernel(const float* __restrict__ partial, int N, int nchunks, int groups,
                                                          int C, int chunk_cnt, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, float eps,
                                                          float* __restrict__ scale, float* __restrict__ shift,
                                                          float* __restrict__ mean_rstd) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N * groups) return;
    const int n = i / groups, g = i % groups, cg = C / groups;
    float na = 0.f, mean = 0.f, m2 = 0.f;
    const float m = (float)chunk_cnt;
    for (int k = 0; k < nchunks; ++k) {
        const float* p = partial + (((int64_t)n * nchunks + k) * groups + g) * 2;
        const float delta = p[0] - mean;
        const float tot = na + m;
        mean += delta * (m / tot);
        m2 += p[1] + delta * delta * (na * m / tot);
        na = tot;
    }
    const float rstd = 1.0f / sqrtf(m2 / na + eps);
    if (mean_rstd) {
        mean_rstd[(int64_t)i * 2] = mean;
        mean_rstd[(int64_t)i * 2 + 1] = rstd;
    }
    for (int j = 0; j < cg; ++j) {
        const int c = g * cg + j;
        const float a = rstd * gamma[c];
        scale[(int64_t)n * C + c] = a;
        shift[(int64_t)n * C + c] = beta[c] - mean * a;
    }
}

// Merge the per-tile {mean, M2} partials emitted by the producing convolutions (conv_common.h, fused
// statistics) of one tensor, or of the two tensors of a channel concat, into scale/shift (+ mean/rstd).
// A consumer group of cg = (C1+C2)/groups channels is a whole number of producer groups (C_i/groups each)
// of exactly one source; every partial of source i covers cnt_i elements.
template <bool BATCH, bool MOD>  // MOD: scale-shift conditioning folded in (IDDPM blocks).  BATCH: many partials per group (64x64 maps): break the load-latency chain; else the compact loop (a
                       // kernel this small is dominated by its cold instruction fetch: the unrolled form costs 0.8 us more)
__global__ void __launch_bounds__(256) gn_finalize_parts_kernel(const float* __restrict__ p1, int t1, int cnt1, int C1,
                                                                const float* __restrict__ p2, int t2, int cnt2, int C2, int N, int groups,
                                                                const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                                                float* __restrict__ scale, float* __restrict__ shift,
                                                                float* __restrict__ mean_rstd, const float* __restrict__ t_shift,
                                                                const float* __restrict__ t_scale, int t_ld, int nt) {
    // BATCH: eight lanes share one (image, group): each merges every eighth partial, then three butterfly steps of Chan's formula
    constexpr int LANES = BATCH ? 8 : 1;
    const int gi = blockIdx.x * blockDim.x + threadIdx.x;
    const int i = gi / LANES, sub = gi % LANES;
    if (i >= N * groups) return;  // whole 8-lane teams leave together
    const int n = i / groups, g = i % groups, C = C1 + C2, cg = C / groups;
    const int c_first = g * cg;
    const bool second = c_first >= C1;
    const float* p = second ? p2 : p1;
    const int tiles = second ? t2 : t1, cs = second ? C2 : C1;
    const float m = (float)(second ? cnt2 : cnt1);
    const int fg = cs / groups;                          // producer (fine) group size
    const int f0 = (second ? c_first - C1 : c_first) / fg, nf = cg / fg;
    float na = 0.f, mean = 0.f, m2 = 0.f;
    // gamma / beta of this thread's first channel go out FIRST: they do not depend on the statistics, and a kernel this small is
    // one chain of dependent memory round trips - fetched after the merge they were a second, exposed one.  A group's gamma / beta
    // are contiguous (<= 128 B), so the later channels of the loop below hit the lines this load brought in.
    const int jf = sub < cg ? sub : 0;
    const float gam0 = gamma[c_first + jf], bet0 = beta[c_first + jf];
    // partials in (tile, fine group) order
    const int npart = tiles * nf;
    if constexpr (!BATCH) {
        if (npart <= 16) {
            // every partial requested before the first is merged: one round trip (the loop below is npart dependent ones - most of
            // this kernel's 6 us); merged in the same order, so the same bits
            float2 v[16];
            int t = 0, f = 0;
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                v[k] = make_float2(0.f, 0.f);
                if (k < npart) {
                    v[k] = *reinterpret_cast<const float2*>(p + (((int64_t)n * tiles + t) * groups + f0 + f) * 2);
                    if (++f == nf) {
                        f = 0;
                        ++t;
                    }
                }
            }
#pragma unroll
            for (int k = 0; k < 16; ++k)
                if (k < npart) {
                    const float delta = v[k].x - mean, tot = na + m;
                    mean += delta * (m / tot);
                    m2 += v[k].y + delta * delta * (na * m / tot);
                    na = tot;
                }
        } else {
            for (int t = 0; t < tiles; ++t)
                for (int f = 0; f < nf; ++f) {
                    const float* q = p + (((int64_t)n * tiles + t) * groups + f0 + f) * 2;
                    const float delta = q[0] - mean, tot = na + m;
                    mean += delta * (m / tot);
                    m2 += q[1] + delta * delta * (na * m / tot);
                    na = tot;
                }
        }
    } else {
        for (int k = sub; k < npart; k += LANES) {
            const int t = nf == 1 ? k : k / nf, f = nf == 1 ? 0 : k - t * nf;
            const float2 v = *reinterpret_cast<const float2*>(p + (((int64_t)n * tiles + t) * groups + f0 + f) * 2);
            const float delta = v.x - mean, tot = na + m;
            mean += delta * (m / tot);
            m2 += v.y + delta * delta * (na * m / tot);
            na = tot;
        }
#pragma unroll
        for (int off = 1; off < LANES; off <<= 1) {
            const float nb = __shfl_xor(na, off, 64), mb = __shfl_xor(mean, off, 64), m2b = __shfl_xor(m2, off, 64);
            const float tot = na + nb;
            if (tot > 0.f) {
                const float delta = mb - mean, w = nb / tot;
                mean += delta * w;
                m2 += m2b + delta * delta * (na * w);
                na = tot;
            }
        }
    }
    const float rstd = 1.0f / sqrtf(m2 / na + eps);
    if (mean_rstd && sub == 0) {
        mean_rstd[(int64_t)i * 2] = mean;
        mean_rstd[(int64_t)i * 2 + 1] = rstd;
    }
    for (int j = sub; j < cg; j += LANES) {
        const int c = c_first + j;
        const float gm = j == jf ? gam0 : gamma[c], bt = j == jf ? bet0 : beta[c];
        float a = rstd * gm, b = bt - mean * a;
        if constexpr (MOD) {  // scale-shift conditioning folded in (same arithmetic as gn_modulate_kernel)
            const int64_t r = (int64_t)(nt == 1 ? 0 : n) * t_ld + c;
            const float mm = 1.0f + t_scale[r];
            a = a * mm;
            b = fmaf(b, mm, t_shift[r]);
        }
        scale[(int64_t)n * C + c] = a;
        shift[(int64_t)n * C + c] = b;
    }
}

int launch_gn_finalize_parts(const float* part1, int tiles1, int cnt1, int C1, const float* part2, int tiles2, int cnt2, int C2, int N, int groups,
                             const float* gamma, const float* beta, float eps, float* scale, float* shift, float* mean_rstd, const float* t_shift,
                             const float* t_scale, int t_ld, int nt, hipStream_t s) {
    const int tot = N * groups;
    const bool batch = (tiles1 > tiles2 ? tiles1 : tiles2) >= 16;
#define DMME_GNF(BB, MM)                                                                                                                        \
    hipLaunchKernelGGL((gn_finalize_parts_kernel<BB, MM>), dim3((tot * ((BB) ? 8 : 1) + 255) / 256), dim3(256), 0, s, part1, tiles1, cnt1, C1, part2, tiles2, cnt2, C2, N, \
                       groups, gamma, beta, eps, scale, shift, mean_rstd, t_shift, t_scale, t_ld, nt)
    if (t_scale) {
        if (batch) DMME_GNF(true, true); else DMME_GNF(false, true);
    } else {
        if (batch) DMME_GNF(true, false); else DMME_GNF(false, false);
    }
#undef DMME_GNF
    DMME_CHECK_LAUNCH();
    return DMME_OK;
}

// Small feature maps (<= 64 pixels, the 8x8 and 4x4 levels: their convs tile several images together and cannot fuse
// the statistics): ONE workgroup per image does the whole norm - channel sums through LDS atomics, group means, a second
// pass for the centred squares (the image is <= 64 KB: L2 hits), then scale / shift for every channel.  One launch
// instead of two, and no partial buffer.
// act (nullable): the consumer's whole prologue applied here as well - act[n][p][c] = T(silu?(x * scale + shift) * mask), the
// concatenated tensor the conv then reads with no prologue of its own.  On these maps a conv workgroup holds 64 pixels x 64 couts,
// so every element was normalised and passed through SiLU once per cout tile (4x) times the halo overlap (1.56x at 8x8, 2.25x at 4x4):
// cycle stamps put that arithmetic at 23 % of an 8x8 layer and 36 % of a 4x4 layer.  Same prologue_vec as the conv kernels, same
// inputs: the operands the matrix cores see are bit-identical.
template <typename T>
__global__ void __launch_bounds__(256) gn_small_kernel(const T* __restrict__ s1, const T* __restrict__ s2, int HW, int C1, int C2, int groups,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                                       float* __restrict__ scale, float* __restrict__ shift, float* __restrict__ mean_rstd,
                                                       T* __restrict__ act, int act_silu, const float* __restrict__ dmask) {
    constexpr int EPV = 16 / sizeof(T);
    __shared__ float csum[512], gmean[64], grstd[64];
    __shared__ __attribute__((aligned(16))) float asc[512], ash[512], adm[512];
    __shared__ float part[256 * EPV];  // [pixel phase][channel]: fixed-order (bit-reproducible) reduction over the phases
    const int C = C1 + C2, cg = C / groups, n = blockIdx.x, tid = threadIdx.x;
    const int VPP = C / EPV, ppw = 256 / VPP, slot = tid % VPP, prow = tid / VPP;
    const int c0 = slot * EPV;
    const bool second = c0 >= C1;
    const T* src = second ? s2 + (int64_t)n * HW * C2 + (c0 - C1) : s1 + (int64_t)n * HW * C1 + c0;
    const int Cs = second ? C2 : C1;
    const bool active = prow < ppw;
    float acc[EPV];
#pragma unroll
    for (int j = 0; j < EPV; ++j) acc[j] = 0.f;
    if (active)
        for (int p = prow; p < HW; p += ppw) {
            float v[EPV];
            load_vec_gn<T>(src + (int64_t)p * Cs, v);
#pragma unroll
            for (int j = 0; j < EPV; ++j) acc[j] += v[j];
        }
    if (active)
#pragma unroll
        for (int j = 0; j < EPV; ++j) part[prow * C + c0 + j] = acc[j];
    __syncthreads();
    for (int c = tid; c < C; c += 256) {
        float s = 0.f;
        for (int q = 0; q < ppw; ++q) s += part[q * C + c];
        csum[c] = s;
    }
    __syncthreads();
    const float inv_cnt = 1.f / (float)(HW * cg);
    if (tid < groups) {
        float s = 0.f;
        for (int j = 0; j < cg; ++j) s += csum[tid * cg + j];
        gmean[tid] = s * inv_cnt;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < EPV; ++j) acc[j] = 0.f;
    if (active) {
        float mu[EPV];
#pragma unroll
        for (int j = 0; j < EPV; ++j) mu[j] = gmean[(c0 + j) / cg];
        for (int p = prow; p < HW; p += ppw) {
            float v[EPV];
            load_vec_gn<T>(src + (int64_t)p * Cs, v);
#pragma unroll
            for (int j = 0; j < EPV; ++j) {
                const float d = v[j] - mu[j];
                acc[j] = fmaf(d, d, acc[j]);
            }
        }
#pragma unroll
        for (int j = 0; j < EPV; ++j) part[prow * C + c0 + j] = acc[j];
    }
    __syncthreads();
    for (int c = tid; c < C; c += 256) {
        float s = 0.f;
        for (int q = 0; q < ppw; ++q) s += part[q * C + c];
        csum[c] = s;
    }
    __syncthreads();
    if (tid < groups) {
        float s = 0.f;
        for (int j = 0; j < cg; ++j) s += csum[tid * cg + j];
        const float rstd = 1.0f / sqrtf(s * inv_cnt + eps);
        grstd[tid] = rstd;
        if (mean_rstd) {
            mean_rstd[((int64_t)n * groups + tid) * 2] = gmean[tid];
            mean_rstd[((int64_t)n * groups + tid) * 2 + 1] = rstd;
        }
    }
    __syncthreads();
    for (int c = tid; c < C; c += 256) {
        const int gq = c / cg;
        const float a = grstd[gq] * gamma[c], b = beta[c] - gmean[gq] * a;
        scale[(int64_t)n * C + c] = a;
        shift[(int64_t)n * C + c] = b;
        asc[c] = a;
        ash[c] = b;
        adm[c] = dmask ? dmask[(int64_t)n * C + c] : 1.0f;
    }
    if (!act) return;
    __syncthreads();
    if (active) {
        T* dst = act + (int64_t)n * HW * C + c0;
        for (int p = prow; p < HW; p += ppw) {
            const uint4 raw = *reinterpret_cast<const uint4*>(src + (int64_t)p * Cs);
            *reinterpret_cast<uint4*>(dst + (int64_t)p * C) = prologue_vec<T>(raw, asc + c0, ash + c0, dmask ? adm + c0 : nullptr, act_silu);
        }
    }
}

static bool gn_small_supported(int dtype, int HW, int C1, int C2, int groups) {
    const int EPV = is16(dtype) ? 8 : 4, C = C1 + C2;
    return HW <= 64 && C <= 512 && groups <= 64 && C % groups == 0 && C1 % EPV == 0 && C2 % EPV == 0 && C / EPV <= 256 && !debug_route("no_gn_small");
}

static bool gn_geometry(int dtype, int HW, int C1, int C2, int groups, int& chunk_px, int& nsweeps, int& nchunks) {
    const int EPV = is16(dtype) ? 8 : 4;
    const int C = C1 + C2;
    if (C % EPV || C1 % EPV || groups > 256 || C % groups) return false;
    const int VPP = C / EPV;
    if (VPP > 256 || 256 % VPP) return false;
    const int ppw = 256 / VPP;
    if (HW % ppw) return false;
    int sweeps = HW / ppw;
    if (sweeps > GN_MAX_SWEEPS) sweeps = GN_MAX_SWEEPS;
    while (sweeps > 1 && (HW / ppw) % sweeps) --sweeps;
    chunk_px = sweeps * ppw;
    nsweeps = sweeps;
    nchunks = HW / chunk_px;
    return true;
}

bool gn_fast_supported(int dtype, int N, int HW, int C1, int C2, int groups) {
    if (gn_small_supported(dtype, HW, C1, C2, groups)) return true;
    int a, b, c;
    (void)N;
    return gn_geometry(dtype, HW, C1, C2, groups, a, b, c);
}

size_t gn_fast_scratch_floats(int N, int HW, int C, int groups) {
    // worst case over dtypes: chunk of one sweep of the fp32 geometry
    int chunk_px, nsweeps, nchunks = 0;
    size_t best = 0;
    for (int dt = 0; dt < 2; ++dt)
        if (gn_geometry(dt, HW, C, 0, groups, chunk_px, nsweeps, nchunks)) {
            const size_t v = (size_t)N * nchunks * groups * 2;
            if (v > best) best = v;
        }
    return best;
}

bool gn_small_act_supported(int dtype, int HW, int C1, int C2, int groups) { return gn_small_supported(dtype, HW, C1, C2, groups); }

int launch_gn_fast(int dtype, const void* src1, const void* src2, int N, int HW, int C1, int C2, int groups,
                   const float* gamma, const float* beta, float eps, float* scale, float* shift, float* mean_rstd,
                   float* partial, hipStream_t s, void* act, int act_silu, const float* dmask) {
    if (gn_small_supported(dtype, HW, C1, C2, groups)) {
        if (dtype == DMME_BF16)
            hipLaunchKernelGGL(gn_small_kernel<bf16>, dim3(N), dim3(256), 0, s, (const bf16*)src1, (const bf16*)src2, HW, C1, C2, groups, gamma, beta,
                               eps, scale, shift, mean_rstd, (bf16*)act, act_silu, dmask);
        else if (dtype == DMME_F16)
            hipLaunchKernelGGL(gn_small_kernel<f16>, dim3(N), dim3(256), 0, s, (const f16*)src1, (const f16*)src2, HW, C1, C2, groups, gamma, beta,
                               eps, scale, shift, mean_rstd, (f16*)act, act_silu, dmask);
        else
            hipLaunchKernelGGL(gn_small_kernel<float>, dim3(N), dim3(256), 0, s, (const float*)src1, (const float*)src2, HW, C1, C2, groups, gamma,
                               beta, eps, scale, shift, mean_rstd, (float*)act, act_silu, dmask);
        DMME_CHECK_LAUNCH();
        return DMME_OK;
    }
    DMME_REQUIRE(!act, DMME_ERR_UNSUPPORTED, "gn_fast: pre-activated output needs the one-workgroup-per-image kernel");
    int chunk_px, nsweeps, nchunks;
    DMME_REQUIRE(gn_geometry(dtype, HW, C1, C2, groups, chunk_px, nsweeps, nchunks), DMME_ERR_UNSUPPORTED,
                 "gn_fast: unsupported geometry");
    const int C = C1 + C2;
    dim3 grid(nchunks, N);
    if (dtype == DMME_BF16)
        hipLaunchKernelGGL(gn_partial_kernel<bf16>, grid, dim3(256), 0, s, (const bf16*)src1, (const bf16*)src2, HW, C1, C2,
                           groups, chunk_px, nsweeps, partial);
    else if (dtype == DMME_F16)
        hipLaunchKernelGGL(gn_partial_kernel<f16>, grid, dim3(256), 0, s, (const f16*)src1, (const f16*)src2, HW, C1, C2,
                           groups, chunk_px, nsweeps, partial);
    else
        hipLaunchKernelGGL(gn_partial_kernel<float>, grid, dim3(256), 0, s, (const float*)src1, (const float*)src2, HW, C1,
                           C2, groups, chunk_px, nsweeps, partial);
    DMME_CHECK_LAUNCH();
    const int tot = N * groups;
    hipLaunchKernelGGL(gn_finalize_kernel, dim3((tot + 255) / 256), dim3(256), 0, s, partial, N, nchunks, groups, C,
                       chunk_px * (C / groups), gamma, beta, eps, scale, shift, mean_rstd);
    DMME_CHECK_LAUNCH();
    return DMME_OK;
}

}  // namespace dmme
