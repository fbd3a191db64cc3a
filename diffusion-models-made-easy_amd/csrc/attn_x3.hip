// Fused self-attention for the accurate mode (DMME_BF16X3): fp32 q | k | v in, fp32 out, both matrix products as three bf16 MFMA
// passes over hi/lo splits of their fp32 operands (conv_common.h: split4 / mma_x3), softmax in fp32.
//   out[n][i][:] = softmax_j( q_i . k_j * C^-0.5 ) v_j      (Attention.forward_attention, models/ddpm.py:54-63; through the head view
//   also MultiHeadAttention of models/iddpm.py:35-47)
// Same formulation as attn_mfma.hip: one workgroup = one (image, head) row x 32*NW queries, keys stream through LDS in tiles of 32,
// scores are computed TRANSPOSED (S^T = K Q^T) so each lane owns one query and the online-softmax statistics are per-lane scalars;
// the S^T accumulator (keys on the registers) is, four registers at a time, directly the B operand of O^T += V^T P^T.  The fp32
// path ran on the scalar generic kernel before (1.0 ms per launch at batch 128, 42 % of a bf16x3 step).
// Operands stay fp32 in LDS and are split when a fragment is read: 32x32x8 MFMAs take 4 k-values per lane = one 16-byte fp32 read.
#include <stdlib.h>

#include "conv_common.h"

namespace dmme {

constexpr int AX_KT = 32;  // keys per tile

struct AttnGeomX {  // same head view as attn_mfma.hip's AttnGeom
    int S, ld, Cfull, heads, N;
    float scale;
};

// C = head width (64 / 128 / 256); NW wavefronts of 32 queries each
template <int C, int NW>
__global__ void __launch_bounds__(64 * NW) attn_x3_kernel(const float* __restrict__ qkv, AttnGeomX g, float* __restrict__ out) {
    constexpr int QB = 32 * NW, NT = 64 * NW;
    constexpr int GS = C / 8;        // 8-channel k-groups of the score product
    constexpr int CT = C / 32;       // 32-channel tiles of the output
    constexpr int RP = C * 4 + 16;   // fp32 row pitch in LDS: +16 B walks consecutive rows over the banks (conflict-free 16-byte reads)
    extern __shared__ __attribute__((aligned(16))) char lds[];
    char* ldsQ = lds;
    char* ldsK = lds + QB * RP;
    char* ldsV = ldsK + AX_KT * RP;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int qblocks = g.S / QB;
    const int bh = blockIdx.x / qblocks, qb = blockIdx.x % qblocks;
    const float* base = qkv + (int64_t)(bh / g.heads) * g.S * g.ld + (int64_t)(bh % g.heads) * 3 * C;
    const int ld = g.ld;

    for (int u = tid; u < QB * (C / 4); u += NT) {
        const int row = u / (C / 4), cu = u % (C / 4);
        *reinterpret_cast<uint4*>(ldsQ + row * RP + cu * 16) = *reinterpret_cast<const uint4*>(base + (int64_t)(qb * QB + row) * ld + cu * 4);
    }
    const char* q_lds = ldsQ + (wave * 32 + r) * RP + h * 16;  // this lane's query row, its half of every 8-channel group

    f32x16 o[CT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int j = 0; j < 16; ++j) o[ct][j] = 0.f;
    float m = -1e30f, l = 0.f;
    const float c1 = 1.4426950408889634f * g.scale;  // Cfull^-0.5 * log2(e)

    for (int k0 = 0; k0 < g.S; k0 += AX_KT) {
        __syncthreads();  // the previous tile's fragments are read (first pass: Q is being written by other threads, fenced below)
        for (int u = tid; u < AX_KT * (C / 4); u += NT) {
            const int row = u / (C / 4), cu = u % (C / 4);
            const float* src = base + (int64_t)(k0 + row) * ld + cu * 4;
            *reinterpret_cast<uint4*>(ldsK + row * RP + cu * 16) = *reinterpret_cast<const uint4*>(src + C);
            *reinterpret_cast<uint4*>(ldsV + row * RP + cu * 16) = *reinterpret_cast<const uint4*>(src + 2 * C);
        }
        __syncthreads();
        // ---- S^T tile (32 keys x 32 queries) = K Q^T, three passes per 8-channel group ----
        f32x16 st;
#pragma unroll
        for (int j = 0; j < 16; ++j) st[j] = 0.f;
#pragma unroll 4
        for (int gs = 0; gs < GS; ++gs) {
            const uint4 kf = *reinterpret_cast<const uint4*>(ldsK + r * RP + gs * 32 + h * 16);
            const uint4 qf = *reinterpret_cast<const uint4*>(q_lds + gs * 32);
            mma_x3(split4(kf), split4(qf), st);
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
        if (__any(alpha != 1.0f)) {
#pragma unroll
            for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                for (int j = 0; j < 16; ++j) o[ct][j] *= alpha;
        }
        // ---- O^T += V^T P^T: key block b (8 keys) uses registers 4b .. 4b+3 of P^T (key = 8 b + 4 h + (j & 3)) ----
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const f32x4 pv = {p[4 * b], p[4 * b + 1], p[4 * b + 2], p[4 * b + 3]};
            const Split4 ps = split4(__builtin_bit_cast(uint4, pv));
            const char* vrow = ldsV + (8 * b + 4 * h) * RP + r * 4;
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
                f32x4 vv;
#pragma unroll
                for (int j = 0; j < 4; ++j) vv[j] = *reinterpret_cast<const float*>(vrow + j * RP + ct * 128);
                mma_x3(split4(__builtin_bit_cast(uint4, vv)), ps, o[ct]);
            }
        }
    }
    // ---- normalise and store: lane = query, registers = channels (j & 3) + 8 (j >> 2) + 4 h ----
    const float ltot = l + __shfl_xor(l, 32, 64);
    const float inv = 1.0f / ltot;
    const int q_row = qb * QB + wave * 32 + r;
    float* orow = out + (int64_t)(bh % g.N) * g.S * g.Cfull + (int64_t)(bh / g.N) * C + (int64_t)q_row * g.Cfull;
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int jg = 0; jg < 4; ++jg) {
            f32x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = o[ct][jg * 4 + e] * inv;
            *reinterpret_cast<f32x4*>(orow + ct * 32 + 8 * jg + 4 * h) = v;
        }
}

// queries per workgroup: as many 32-query waves as the fp32 Q tile leaves room for beside the K / V tiles in 160 KB of LDS
static int ax_waves(int D) { return D <= 128 ? 4 : 2; }

bool attn_x3_supported(int N, int S, int C, int heads) {
    (void)N;
    if (heads < 1 || C % heads) return false;
    const int D = C / heads;
    if (!(D == 64 || D == 128 || D == 256)) return false;
    const int QB = 32 * ax_waves(D);
    return S >= QB && S % QB == 0 && S % AX_KT == 0;
}

template <int D, int NW>
static int launch_ax(const float* qkv, const AttnGeomX& g, float* out, hipStream_t s) {
    const size_t lds = (size_t)(32 * NW + 2 * AX_KT) * (D * 4 + 16);
    static bool attr_done = false;
    if (!attr_done) {
        DMME_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(attn_x3_kernel<D, NW>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_done = true;
    }
    const int rows = g.N * g.heads, qblocks = g.S / (32 * NW);
    hipLaunchKernelGGL((attn_x3_kernel<D, NW>), dim3((unsigned)(rows * qblocks)), dim3(64 * NW), lds, s, qkv, g, out);
    DMME_CHECK_LAUNCH();
    return DMME_OK;
}

int launch_attn_x3(const void* qkv, int N, int S, int C, int heads, void* out, hipStream_t s) {
    DMME_REQUIRE(attn_x3_supported(N, S, C, heads), DMME_ERR_UNSUPPORTED, "attn_x3: unsupported shape S=%d C=%d heads=%d", S, C, heads);
    const AttnGeomX g{S, 3 * C, C, heads, N, 1.0f / sqrtf((float)C)};
    switch (C / heads) {
        case 256: return launch_ax<256, 2>((const float*)qkv, g, (float*)out, s);
        case 128: return launch_ax<128, 4>((const float*)qkv, g, (float*)out, s);
        default: return launch_ax<64, 4>((const float*)qkv, g, (float*)out, s);
    }
}

}  // namespace dmme
