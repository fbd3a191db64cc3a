// Fused single-head self-attention on the matrix cores (bf16 in, fp32 accumulate):
//   out[n][i][:] = softmax_j( q_i . k_j * C^-0.5 ) v_j      (Attention.forward_attention,
//   models/ddpm.py:54-63; the S x S score matrix never exists in HBM)
//
// qkv is [N][S][3C] (NHWC output of the 1x1 qkv conv): q | k | v thirds along channels.
// One workgroup = one image x 128 queries, 4 wavefronts x 32 queries; keys stream through
// LDS in tiles of 32.
//   * scores are computed TRANSPOSED, S^T = K Q^T (A = K rows from LDS, B = Q^T held in
//     registers), so each lane owns one query column: the online-softmax row max / sum
//     are in-register reductions plus one lane<->lane+32 exchange;
//   * the S^T accumulator (key rows in registers, query on the lane) is converted to bf16
//     and used directly as the B operand of  O^T += V^T P^T  (no LDS round trip); the
//     V^T fragments come from the row-major V tile through ds_read_b64_tr_b16, whose
//     key order matches the accumulator's register order;
//   * K tile rows are padded by 16 B (conflict-free ds_read_b128), V tile rows by 64 B
//     (row pitch = 16 banks mod 64: the four key rows of a transposed read hit disjoint banks).
#include "common.h"

namespace dmme {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

constexpr int AT_KT = 32;  // keys per tile
constexpr int AT_QB = 128; // queries per workgroup

template <int C>
__global__ void __launch_bounds__(256) attn_mfma_kernel(const bf16* __restrict__ qkv, int S, bf16* __restrict__ out) {
    constexpr int KSTEPS = C / 16;   // k-steps of the QK^T product
    constexpr int CT = C / 32;       // 32-channel tiles of the output
    constexpr int KP = C * 2 + 16;   // K tile row pitch (bytes)
    constexpr int VP = C * 2 + 64;   // V tile row pitch (bytes)
    __shared__ __attribute__((aligned(16))) char lds[AT_KT * KP + AT_KT * VP];
    char* ldsK = lds;
    char* ldsV = lds + AT_KT * KP;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int qblocks = S / AT_QB;
    const int n = blockIdx.x / qblocks, qb = blockIdx.x % qblocks;
    const bf16* base = qkv + (int64_t)n * S * 3 * C;
    const int q_row = qb * AT_QB + wave * 32 + r;  // this lane's query

    // Q^T fragments (B operand): element j of k-step ks = Q[q_row][16 ks + 8 h + j]
    uint4 qf[KSTEPS];
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks)
        qf[ks] = *reinterpret_cast<const uint4*>(base + (int64_t)q_row * 3 * C + ks * 16 + h * 8);

    f32x16 o[CT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int j = 0; j < 16; ++j) o[ct][j] = 0.f;
    float m = -1e30f, l = 0.f;
    const float c1 = 1.4426950408889634f / sqrtf((float)C);  // C^-0.5 * log2(e)

    // transposed-read lane geometry: 16-lane group g, lane i = 4 q + p inside it
    const int tr_q = (lane & 15) >> 2, tr_p = lane & 3, tr_g1 = (lane >> 4) & 1;

    for (int k0 = 0; k0 < S; k0 += AT_KT) {
        __syncthreads();
        // ---- stage K and V tiles (32 keys x C) ----
        for (int u = tid; u < AT_KT * (C / 8); u += 256) {
            const int row = u / (C / 8), cu = u % (C / 8);
            const bf16* src = base + (int64_t)(k0 + row) * 3 * C + cu * 8;
            *reinterpret_cast<uint4*>(ldsK + row * KP + cu * 16) = *reinterpret_cast<const uint4*>(src + C);
            *reinterpret_cast<uint4*>(ldsV + row * VP + cu * 16) = *reinterpret_cast<const uint4*>(src + 2 * C);
        }
        __syncthreads();
        // ---- S^T tile (32 keys x 32 queries) = K Q^T ----
        f32x16 st;
#pragma unroll
        for (int j = 0; j < 16; ++j) st[j] = 0.f;
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks) {
            const uint4 kf = *reinterpret_cast<const uint4*>(ldsK + r * KP + ks * 32 + h * 16);
            st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, kf), __builtin_bit_cast(bf16x8, qf[ks]), st, 0, 0, 0);
        }
        // ---- online softmax for this lane's query (keys of this lane: 16 of the 32) ----
        float tmax = st[0];
#pragma unroll
        for (int j = 1; j < 16; ++j) tmax = fmaxf(tmax, st[j]);
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
        const float m_new = fmaxf(m, tmax * c1);
        const float alpha = exp2f(m - m_new);
        float psum = 0.f;
        float p[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            p[j] = exp2f(fmaf(st[j], c1, -m_new));
            psum += p[j];
        }
        l = fmaf(l, alpha, psum);
        m = m_new;
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
            for (int j = 0; j < 16; ++j) o[ct][j] *= alpha;
        // P^T as B operand: k-step s uses registers 8s..8s+7 (key = 16 s + 8 (j>>2) + 4 h + (j&3))
        bf16x8 pf[2];
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
            for (int j = 0; j < 8; ++j) pf[s2][j] = (bf16)p[8 * s2 + j];
        // ---- O^T += V^T P^T ----
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const int colb = (ct * 32 + 16 * tr_g1 + 4 * tr_p) * 2;
                const char* a0 = ldsV + (16 * s2 + 4 * h + tr_q) * VP + colb;
                const char* a1 = ldsV + (16 * s2 + 8 + 4 * h + tr_q) * VP + colb;
                const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)a0);
                const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)a1);
                s16x8 vf;
                vf[0] = lo[0]; vf[1] = lo[1]; vf[2] = lo[2]; vf[3] = lo[3];
                vf[4] = hi[0]; vf[5] = hi[1]; vf[6] = hi[2]; vf[7] = hi[3];
                o[ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, vf), pf[s2], o[ct], 0, 0, 0);
            }
        }
    }
    // ---- normalise and store: lane = query, registers = channels (j&3) + 8 (j>>2) + 4 h ----
    const float inv = 1.0f / (l + __shfl_xor(l, 32, 64));
    bf16* orow = out + ((int64_t)n * S + q_row) * C;
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
#pragma unroll
        for (int jg = 0; jg < 4; ++jg) {
            typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
            bf16x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = (bf16)(o[ct][jg * 4 + e] * inv);
            *reinterpret_cast<bf16x4*>(orow + ct * 32 + 8 * jg + 4 * h) = v;
        }
    }
}

bool attn_mfma_supported(int dtype, int N, int S, int C) {
    (void)N;
    return dtype == DMME_BF16 && (C == 128 || C == 256) && S >= AT_QB && S % AT_QB == 0;
}

int launch_attn_mfma(int dtype, const void* qkv, int N, int S, int C, void* out, hipStream_t s) {
    DMME_REQUIRE(attn_mfma_supported(dtype, N, S, C), DMME_ERR_UNSUPPORTED, "attn_mfma: unsupported shape S=%d C=%d", S, C);
    const dim3 grid((unsigned)(N * (S / AT_QB)));
    if (C == 256)
        hipLaunchKernelGGL(attn_mfma_kernel<256>, grid, dim3(256), 0, s, (const bf16*)qkv, S, (bf16*)out);
    else
        hipLaunchKernelGGL(attn_mfma_kernel<128>, grid, dim3(256), 0, s, (const bf16*)qkv, S, (bf16*)out);
    DMME_CHECK_LAUNCH();
    return DMME_OK;
}

}  // namespace dmme
