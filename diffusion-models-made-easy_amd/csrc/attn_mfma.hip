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
#include <stdlib.h>

#include "conv_common.h"

namespace dmme {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

constexpr int AT_KT = 32;  // keys per tile
#ifndef AT_PAIR_STEPS
#define AT_PAIR_STEPS 1
#endif

// Head view (multi-head blocks of the IDDPM UNet, models/iddpm.py:35-47; heads == 1 is the DDPM block): a "row" bh = n*heads + hd
// reads the qkv channels [hd*3D, (hd+1)*3D) of image n as (q | k | v) and owns the output row block (image bh % N, head bh / N) --
// the reference splits "(b head)" and merges "(head b)".  scale = Cfull^-0.5 (the full width, not the head width).
struct AttnGeom {
    int S, ld, Cfull, heads, N;
    float scale;
};
template <int D>
__device__ __forceinline__ int64_t at_qkv_off(const AttnGeom& g, int bh) {
    return (int64_t)(bh / g.heads) * g.S * g.ld + (int64_t)(bh % g.heads) * 3 * D;
}
template <int D>
__device__ __forceinline__ int64_t at_o_off(const AttnGeom& g, int bh) {
    return (int64_t)(bh % g.N) * g.S * g.Cfull + (int64_t)(bh / g.N) * D;
}

// C = head width; NW wavefronts of 32 queries each per workgroup.
// xcd_order: workgroup L runs on XCD L % 8 and every XCD has its own L2; in plain order the query blocks of one (image, head) row
// land on different XCDs and each pulls that row's K and V from HBM (round-1 counters: 1.88x the algorithmic bytes).  With
// xcd_order the blocks of a row take consecutive slots of ONE XCD's round-robin share of the grid.
// one 32x32x16 MFMA on 16-bit operands (bf16 or IEEE half: same rate, same fragment layout)
template <typename T, typename V>
__device__ __forceinline__ void at_mma(const V& a, const V& b, f32x16& acc);
template <>
__device__ __forceinline__ void at_mma<bf16, bf16x8>(const bf16x8& a, const bf16x8& b, f32x16& acc) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
}
typedef f16 f16x8_at __attribute__((ext_vector_type(8)));
template <>
__device__ __forceinline__ void at_mma<f16, f16x8_at>(const f16x8_at& a, const f16x8_at& b, f32x16& acc) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
}

template <int C, int NW, typename T = bf16>
__global__ void __launch_bounds__(64 * NW) attn_mfma_kernel(const T* __restrict__ qkv, AttnGeom g, T* __restrict__ out, float* __restrict__ lse,
                                                            int xcd_order) {
    typedef T tx8 __attribute__((ext_vector_type(8)));  // 8 operands of the 16-bit type (bf16 or IEEE half)
    constexpr int AT_QB = 32 * NW, NT = 64 * NW;
    const int S = g.S;
    constexpr int KSTEPS = C / 16;   // k-steps of the QK^T product
    constexpr int CT = C / 32;       // 32-channel tiles of the output
    constexpr int KP = C * 2 + 16;   // K tile row pitch (bytes)
    constexpr int VP = C * 2 + 64;   // V tile row pitch (bytes)
    // Wide heads keep Q^T in LDS instead of registers (C = 256: 64 registers): with 128 accumulators the register file then has
    // room for the K / V prefetch below, whose absence left eight exposed global round trips per workgroup (37 us for 3.4 us of
    // matrix work at one workgroup per CU).
    constexpr bool QLDS = C > 128;
    constexpr int QP = C * 2 + 16;   // Q tile row pitch (bytes)
    extern __shared__ __attribute__((aligned(16))) char lds[];
    char* ldsK = lds;
    char* ldsV = lds + AT_KT * KP;
    char* ldsQ = lds + AT_KT * KP + AT_KT * VP;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int qblocks = S / AT_QB;
    int n = blockIdx.x / qblocks, qb = blockIdx.x % qblocks;  // n: the (image, head) row
    if (xcd_order) {
        const int x = blockIdx.x & 7, j = blockIdx.x >> 3;
        n = (j / qblocks) * 8 + x;
        qb = j % qblocks;
    }
    const T* base = qkv + at_qkv_off<C>(g, n);
    const int ld = g.ld;
    const int q_row = qb * AT_QB + wave * 32 + r;  // this lane's query

    // Q^T fragments (B operand): element j of k-step ks = Q[q_row][16 ks + 8 h + j]
    uint4 qf[QLDS ? 1 : KSTEPS];
    if constexpr (QLDS) {
        for (int u = tid; u < AT_QB * (C / 8); u += NT) {
            const int row = u / (C / 8), cu = u % (C / 8);
            *reinterpret_cast<uint4*>(ldsQ + row * QP + cu * 16) = *reinterpret_cast<const uint4*>(base + (int64_t)(qb * AT_QB + row) * ld + cu * 8);
        }
    } else {
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks)
            qf[ks] = *reinterpret_cast<const uint4*>(base + (int64_t)q_row * ld + ks * 16 + h * 8);
    }
    const char* q_lds = ldsQ + (wave * 32 + r) * QP + h * 16;

    f32x16 o[CT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int j = 0; j < 16; ++j) o[ct][j] = 0.f;
    float m = -1e30f, l = 0.f;
    const float c1 = 1.4426950408889634f * g.scale;  // Cfull^-0.5 * log2(e)

    // transposed-read lane geometry: 16-lane group g, lane i = 4 q + p inside it
    const int tr_q = (lane & 15) >> 2, tr_p = lane & 3, tr_g1 = (lane >> 4) & 1;

    // K / V tiles go global -> registers -> LDS and the loads of tile k+1 are issued before the matrix work of tile k, so their
    // round trip hides under it.
    constexpr int UNITS = AT_KT * (C / 8) / NT;
    static_assert(UNITS >= 1 && UNITS * NT == AT_KT * (C / 8), "attn_mfma: tile does not split evenly over the threads");
    // named registers, not arrays: hipcc leaves `uint4 kreg[UNITS]` in scratch memory here (every prefetched vector then makes a
    // round trip through the scratch buffer), whatever the unrolling
    uint4 kr0, kr1, kr2, kr3, kr4, kr5, kr6, kr7, vr0, vr1, vr2, vr3, vr4, vr5, vr6, vr7;
    static_assert(UNITS <= 8, "attn_mfma: more prefetch units than named registers");
#define AT_LD1(I, K0)                                                                                 \
    if constexpr (UNITS > I) {                                                                        \
        const int u = tid + I * NT, row = u / (C / 8), cu = u % (C / 8);                              \
        const T* src = base + (int64_t)((K0) + row) * ld + cu * 8;                                 \
        kr##I = *reinterpret_cast<const uint4*>(src + C);                                             \
        vr##I = *reinterpret_cast<const uint4*>(src + 2 * C);                                         \
    }
#define AT_ST1(I)                                                                                     \
    if constexpr (UNITS > I) {                                                                        \
        const int u = tid + I * NT, row = u / (C / 8), cu = u % (C / 8);                              \
        *reinterpret_cast<uint4*>(ldsK + row * KP + cu * 16) = kr##I;                                 \
        *reinterpret_cast<uint4*>(ldsV + row * VP + cu * 16) = vr##I;                                 \
    }
#define AT_FETCH(K0) AT_LD1(0, K0) AT_LD1(1, K0) AT_LD1(2, K0) AT_LD1(3, K0) AT_LD1(4, K0) AT_LD1(5, K0) AT_LD1(6, K0) AT_LD1(7, K0)
    AT_FETCH(0)
    for (int k0 = 0; k0 < S; k0 += AT_KT) {
        __syncthreads();
        // ---- stage K and V tiles (32 keys x C) ----
        AT_ST1(0) AT_ST1(1) AT_ST1(2) AT_ST1(3) AT_ST1(4) AT_ST1(5) AT_ST1(6) AT_ST1(7)
        __syncthreads();
        if (k0 + AT_KT < S) {
            AT_FETCH(k0 + AT_KT)
        }
#undef AT_FETCH
#undef AT_LD1
#undef AT_ST1
        // ---- S^T tile (32 keys x 32 queries) = K Q^T ----
        // One wave per SIMD: nothing hides an LDS read's latency but this wave's own instruction order.  Fragments are therefore
        // read a group of FG k-steps AHEAD of the MFMAs that use them (sched_barrier pins the order the compiler would otherwise
        // re-serialise into read -> MFMA -> read): the exposed latencies per tile drop from 32 to ~6.
        f32x16 st;
#pragma unroll
        for (int j = 0; j < 16; ++j) st[j] = 0.f;
        {
            constexpr int FG = 4;
            uint4 kf[2][FG], qv[2][FG];  // two fragment sets, indexed by the (compile-time) group parity: no register copies
#pragma unroll
            for (int u = 0; u < FG; ++u) {
                kf[0][u] = *reinterpret_cast<const uint4*>(ldsK + r * KP + u * 32 + h * 16);
                if constexpr (QLDS) qv[0][u] = *reinterpret_cast<const uint4*>(q_lds + u * 32);
                else qv[0][u] = qf[u];
            }
#pragma unroll
            for (int g0 = 0; g0 < KSTEPS; g0 += FG) {
                constexpr int dummy = 0;
                (void)dummy;
                const int cur = (g0 / FG) & 1, nxt = cur ^ 1;
                if (g0 + FG < KSTEPS) {
#pragma unroll
                    for (int u = 0; u < FG; ++u) {
                        kf[nxt][u] = *reinterpret_cast<const uint4*>(ldsK + r * KP + (g0 + FG + u) * 32 + h * 16);
                        if constexpr (QLDS) qv[nxt][u] = *reinterpret_cast<const uint4*>(q_lds + (g0 + FG + u) * 32);
                        else qv[nxt][u] = qf[(g0 + FG + u) < KSTEPS ? (g0 + FG + u) : 0];
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int u = 0; u < FG; ++u)
                    at_mma<T>(__builtin_bit_cast(tx8, kf[cur][u]), __builtin_bit_cast(tx8, qv[cur][u]), st);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        // V^T fragments of the first output tiles go out now: their latency hides under the softmax arithmetic
        constexpr int VG = CT >= 2 ? 2 : 1;  // output tiles per fragment group
        s16x4 vlo[2][VG][2], vhi[2][VG][2];  // two sets by group parity, as above
#define AT_VREAD(CT0, DST_LO, DST_HI)                                                                                         \
    _Pragma("unroll") for (int cu = 0; cu < VG; ++cu) _Pragma("unroll") for (int s2 = 0; s2 < 2; ++s2) {                       \
        const int colb = (((CT0) + cu) * 32 + 16 * tr_g1 + 4 * tr_p) * 2;                                                     \
        const char* a0 = ldsV + (16 * s2 + 4 * h + tr_q) * VP + colb;                                                         \
        const char* a1 = ldsV + (16 * s2 + 8 + 4 * h + tr_q) * VP + colb;                                                     \
        DST_LO[cu][s2] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)a0);               \
        DST_HI[cu][s2] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)a1);               \
    }
        AT_VREAD(0, vlo[0], vhi[0])
        // ---- online softmax for this lane's query (keys of this lane: 16 of the 32) ----
        float tmax = st[0];
#pragma unroll
        for (int j = 1; j < 16; ++j) tmax = fmaxf(tmax, st[j]);
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
        const float m_new = fmaxf(m, tmax * c1);
        const float alpha = __builtin_amdgcn_exp2f(m - m_new);  // (v_exp_f32: arguments <= 0, a flushed denormal is a zero weight)
        float psum = 0.f;
        float p[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            p[j] = __builtin_amdgcn_exp2f(fmaf(st[j], c1, -m_new));
            psum += p[j];
        }
        l = fmaf(l, alpha, psum);
        m = m_new;
        // the running maximum settles after the first tiles: skip the 16 CT multiplies (and the accumulator round trips
        // behind them) when no lane of the wave moved it.  alpha == 1 exactly then, so the result is the same bit for bit.
        if (__any(alpha != 1.0f)) {
#pragma unroll
            for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                for (int j = 0; j < 16; ++j) o[ct][j] *= alpha;
        }
        // P^T as B operand: k-step s uses registers 8s..8s+7 (key = 16 s + 8 (j>>2) + 4 h + (j&3))
        tx8 pf[2];
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
            for (int j = 0; j < 8; ++j) pf[s2][j] = (T)p[8 * s2 + j];
        // ---- O^T += V^T P^T ----
#pragma unroll
        for (int ct = 0; ct < CT; ct += VG) {
            const int cur = (ct / VG) & 1, nxt = cur ^ 1;
            if (ct + VG < CT) {
                AT_VREAD(ct + VG, vlo[nxt], vhi[nxt])
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int cu = 0; cu < VG; ++cu)
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    s16x8 vf;
                    vf[0] = vlo[cur][cu][s2][0]; vf[1] = vlo[cur][cu][s2][1]; vf[2] = vlo[cur][cu][s2][2]; vf[3] = vlo[cur][cu][s2][3];
                    vf[4] = vhi[cur][cu][s2][0]; vf[5] = vhi[cur][cu][s2][1]; vf[6] = vhi[cur][cu][s2][2]; vf[7] = vhi[cur][cu][s2][3];
                    at_mma<T>(__builtin_bit_cast(tx8, vf), pf[s2], o[ct + cu]);
                }
            __builtin_amdgcn_sched_barrier(0);
        }
#undef AT_VREAD
    }
    // ---- normalise and store: lane = query, registers = channels (j&3) + 8 (j>>2) + 4 h ----
    const float ltot = l + __shfl_xor(l, 32, 64);
    const float inv = 1.0f / ltot;
    // log2-domain log-sum-exp of the scaled scores, kept for the backward pass: p = exp2(s*c1 - lse)
    if (lse && h == 0) lse[(int64_t)n * S + q_row] = m + log2f(ltot);
    T* orow = out + at_o_off<C>(g, n) + (int64_t)q_row * g.Cfull;
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
#pragma unroll
        for (int jg = 0; jg < 4; ++jg) {
            typedef T bf16x4 __attribute__((ext_vector_type(4)));
            bf16x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = (T)(o[ct][jg * 4 + e] * inv);
            *reinterpret_cast<bf16x4*>(orow + ct * 32 + 8 * jg + 4 * h) = v;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// S = 256 keys (the 16x16 maps of the DDPM UNet, models/ddpm.py:54-63): the WHOLE score row of a query fits in registers (8 tiles of
// 32 keys x 16 accumulators), so the softmax is done once per row instead of once per key tile:
//   phase 1  S^T = K Q^T for all eight key tiles (Q^T fragments in registers, K tiles streamed);
//   phase 2  row max / exp2 / row sum over the lane's 128 scores + one lane <-> lane + 32 exchange; P^T -> 16-bit B operands (64 registers);
//   phase 3  O^T = sum over the eight V tiles of V^T P^T; scaled by 1 / row sum on the way out.
// The online form above spends ~630 VALU instructions per key tile (running maximum, rescaling 128 accumulators, per-tile address
// arithmetic) against 32 MFMAs; here the two streaming phases contain NO VALU instruction: K / V tiles arrive by LDS-DMA
// (global_load_lds_dwordx4: no registers, issued behind the compiler's back with counted waits) into an eight-slot ring, seven tiles
// ahead, ONE workgroup barrier per tile; rows are unpadded and XOR-swizzled on the DMA's source side (K: 16-byte piece ^ (row & 7),
// conflict-free ds_read_b128; V: piece ^ 2 (row & 3), the four key rows of a transposed read land in four different 32-byte slots);
// every fragment address is one per-lane base + an instruction offset (the sixteen tile steps are unrolled).
// PACING (`xcd_order` bits 1..: s_sleep 2 / 4 / 8 / 16 after every tile step, default 4; DMME_DEBUG_ROUTE=attn_sleep=0 turns it off): un-paced this
// kernel is 22 us against the online kernel's 30 - and on most boxes of the pool the WHOLE denoising step got 4 % slower with it
// (every other kernel of the step 4-5 % slower, in one process, alternating the two kernels per 100 forwards: tools/attn_flip.py),
// on a few boxes 1 % faster.  MFMA, LDS and the DMA path saturated on all 256 CUs at once trips the board's power / current
// management, which takes the clock down for milliseconds; a few idle cycles per tile step (kernel 25 us) avoid that and make the
// step 0.4 % faster than the online kernel on the limited boxes, 0.5 % on the others.
// PROJ (round 5; VERDICT round 4 item 5, the part that fits): the block's `proj` 1x1 conv, its bias and the residual add
// (models/ddpm.py:66-75: `x + proj(attention(norm(x)))`) run in the same launch - a FOURTH streaming phase with the structure of phase 1:
//   phase 4  out^T = Wp O^T over the C / 32 row tiles of the proj matrix (32 couts x C, a K tile's size and swizzle: same ring, same
//            fragment addresses).  O^T's accumulators, scaled by 1 / row sum and rounded to 16 bits (the rounding of the context
//            tensor the separate launch read back), become the B operand after ONE half-wave register swap per two registers
//            (v_permlane32_swap: lane (q, h) then holds channels 16 ks + 8 h .. + 7 of its query - the layout of the Q^T fragments);
//   epilogue bias + residual, one rounding, the output rows, and the GroupNorm partials of the next norm in the 1x1 kernel's format
//            ((mean, M2) per 32 pixels = one wave, per group; two wave reductions per group, no LDS).
// The context tensor is written only where a backward pass will read it (AttnProj::ctx; the proj conv's weight gradient).  All
// ordinary loads / stores of the epilogue are issued after the last LDS-DMA has been waited for: the counted vmcnt waits of the
// stream never see them.  Removes per block: one launch (15 us at batch 128), the context tensor's write and read-back.
struct AttnProj {
    const void* w;      // [C][C] 16-bit proj matrix, row = cout (the packed 1x1 layout)
    const float* bias;  // [C]
    const void* res;    // [N][S][C] the block's input (residual)
    void* dst;          // [N][S][C]
    float* gn_part;     // nullable: (mean, M2) per (image, 32-pixel tile, group) - ConvArgs::gn_part
    int gn_tiles, gn_cg;
    long long* stamps;  // diagnostic (null: off): cycle stamps of wave 0 of workgroup 0 in -DAT_STAMPS builds (dmme_debug_set_stamps; tools/stamp_attn.py)
    int dbg;            // DMME_DEBUG_ROUTE=attn_proj_dbg=<mask> (timing experiments, wrong results): 1 no statistics, 2 no epilogue loads / stores, 4 no phase-4 MFMAs, 8 image 0's K / V for every workgroup
};
// the fp32 product as a value of its own: hipcc otherwise folds `(half)(a * b)` into v_fma_mix*_f16 in one form of the kernel and not in
// the other - one rounding instead of two, a last-bit difference in ~3e-5 of the context tensor's elements between them
__device__ __forceinline__ float at_keep_f32(float x) {
    asm volatile("" : "+v"(x));
    return x;
}
__device__ __forceinline__ float at_sum64(float v) {
    v = half_sum(v);
    float a = v, b = v;
    permlane32_swap(a, b);
    return a + b;
}
template <int C, typename T = bf16, bool PROJ = false>
__global__ void __launch_bounds__(256) attn_full_kernel(const T* __restrict__ qkv, AttnGeom g, T* __restrict__ out, float* __restrict__ lse, int xcd_order, AttnProj pj) {
    typedef T tx8 __attribute__((ext_vector_type(8)));
    constexpr int S = 256, NKT = S / AT_KT, KSTEPS = C / 16, CT = C / 32;
    constexpr int NTOT = 2 * NKT + (PROJ ? CT : 0);  // tiles of the stream: K, V, then the proj matrix
    constexpr int ROWB = C * 2, TILEB = AT_KT * ROWB;      // bytes per key row / per tile (unpadded)
    constexpr int DPT = TILEB / (256 * 16);                // DMA wave-instructions per tile and wave
    // PAIR (-DAT_PAIR_STEPS=0: one barrier per tile, seven tiles ahead): the stream advances in steps of TWO tiles - one counted wait, one
    // workgroup barrier, two tile requests, then 2 x 16 MFMAs back to back.  With one wave per SIMD nothing covers the barrier and the
    // LDS round trip of a step's first fragments (~0.5 us of a ~0.85 us step for 0.28 us of MFMA; tools/time_attn_proj.py: the same with
    // every workgroup on one L2-resident image - the K / V stream is not what the kernel waits for); a pair pays that once.
    constexpr bool PAIR = AT_PAIR_STEPS != 0;
    constexpr int NSLOT_BYTES = 8 * TILEB;
    constexpr int NSLOT = 8, AHEAD = PAIR ? 6 : 7;  // (a four-slot ring, 48 KB in flight per CU, measured 24 us at C = 256: bytes in flight / latency)
    static_assert(!PAIR || (NTOT % 2 == 0 && NKT % 2 == 0), "pair steps: even tile counts per phase");
    static_assert(DPT >= 1 && DPT * 256 * 16 == TILEB, "attn_full: tile does not split over the DMA lanes");
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    constexpr int qblocks = S / 128;
    int n = blockIdx.x / qblocks, qb = blockIdx.x % qblocks;
    if (xcd_order & 1) {
        const int x = blockIdx.x & 7, j = blockIdx.x >> 3;
        n = (j / qblocks) * 8 + x;
        qb = j % qblocks;
    }
    const T* base = qkv + at_qkv_off<C>(g, n);
    const int ld = g.ld;
    const int q_row = qb * 128 + wave * 32 + r;
    // ---- the tile stream: tile t < NKT is K tile t, tile NKT + t is V tile t; slot = t % NSLOT ----
    // DMA instruction I of a tile (wave-instruction index wave + 4 i) fills LDS bytes [I * 1024, +1024) lane-linearly:
    // row = (I * 1024 + lane * 16) / ROWB, 16-byte piece pc of that row; it must hold SOURCE piece pc ^ key(row) (per 128-byte segment)
    const unsigned lds0 = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(size_t)(__attribute__((address_space(3))) char*)lds);
    // stamps go to LDS (behind the ring / the staging tiles) and to memory at the very end: a global store per stamp would join the
    // vmcnt queue of the counted waits
    // (-DAT_STAMPS builds only: compiled in, even unused, they cost the sampling step 0.3 %)
#ifdef AT_STAMPS
    long long* st_lds = reinterpret_cast<long long*>(lds + (PROJ ? (4 * 32 * (C * 4 + 16) > NSLOT_BYTES ? 4 * 32 * (C * 4 + 16) : NSLOT_BYTES) : NSLOT_BYTES));
    int st_i = 0;
    const bool stamping = PROJ && pj.stamps && blockIdx.x == 0 && tid == 0;
#define AF_STAMP() { if (PROJ && stamping && st_i < 120) st_lds[st_i++] = (long long)clock64(); }
#else
#define AF_STAMP()
#endif
    AF_STAMP()
    unsigned koff[DPT], voff[DPT];  // per-lane source byte offsets inside a K / V tile (row * ld * 2 + swizzled piece * 16)
#pragma unroll
    for (int i = 0; i < DPT; ++i) {
        const int I = wave + 4 * i, byte = I * 1024 + lane * 16, row = byte / ROWB, pc = (byte % ROWB) >> 4;
        koff[i] = (unsigned)(row * ld * 2 + (((pc & ~7) | ((pc & 7) ^ (row & 7))) << 4));
        voff[i] = (unsigned)(row * ld * 2 + (((pc & ~7) | ((pc & 7) ^ ((row & 3) << 1))) << 4));
    }
    unsigned woff[PROJ ? DPT : 1];  // ... inside a tile of the proj matrix (rows of C elements, a K tile's swizzle)
    if constexpr (PROJ) {
#pragma unroll
        for (int i = 0; i < DPT; ++i) {
            const int I = wave + 4 * i, byte = I * 1024 + lane * 16, row = byte / ROWB, pc = (byte % ROWB) >> 4;
            woff[i] = (unsigned)(row * ROWB + (((pc & ~7) | ((pc & 7) ^ (row & 7))) << 4));
        }
    }
    auto dma_tile = [&](int t) __attribute__((always_inline)) {
        const unsigned dst = lds0 + (unsigned)((t % NSLOT) * TILEB);
        if (PROJ && t >= 2 * NKT) {
            const char* src = (const char*)pj.w + (size_t)(t - 2 * NKT) * TILEB;
#pragma unroll
            for (int i = 0; i < DPT; ++i) glds16_hidden_s(src, woff[i], dst + (unsigned)((wave + 4 * i) * 1024));
            return;
        }
        const bool isv = t >= NKT;
        const T* kvb = (PROJ && (pj.dbg & 8)) ? qkv : base;  // (timing experiment: every workgroup streams image 0's K / V)
        const char* src = (const char*)(kvb + (int64_t)((isv ? t - NKT : t) * AT_KT) * ld + (isv ? 2 * C : C));
#pragma unroll
        for (int i = 0; i < DPT; ++i) glds16_hidden_s(src, isv ? voff[i] : koff[i], dst + (unsigned)((wave + 4 * i) * 1024));
    };
#pragma unroll
    for (int t = 0; t < AHEAD; ++t) dma_tile(t);
    // Q^T fragments (B operand of S^T = K Q^T): element j of k-step ks = Q[q_row][16 ks + 8 h + j]; ordinary loads, older than nothing
    // the counted waits below care about (they are waited for by the compiler before the first MFMA).  (Requested BEFORE the tiles -
    // so that the first counted wait asks for two tiles instead of all six - the launch's first 9.5 k cycles stay what they are:
    // 24 MB requested by 256 CUs at once, the HBM's rate.)
    uint4 qf[KSTEPS];
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks) qf[ks] = *reinterpret_cast<const uint4*>(base + (int64_t)q_row * ld + ks * 16 + h * 8);
    // per-lane fragment bases.  K: row r, piece (2 u + h) of k-step u -> ((2 u & 7) ^ x) with x = h ^ (r & 7): four variants
    unsigned kb[4];
    {
        const int x = h ^ (r & 7);
#pragma unroll
        for (int i = 0; i < 4; ++i) kb[i] = lds0 + (unsigned)(r * ROWB + (((2 * i) ^ x) << 4));
    }
    // V (transposed reads, lane i = 4 q + p of a 16-lane group g1): key row 4 h + q (+ 16 s2, + 8), channels 32 ct + 16 g1 + 4 p ..+3:
    // piece = 4 ct + 2 g1 + (p >> 1), swizzled piece = ((ct & 1) ^ q1) 4 + (g1 ^ q0) 2 + (p >> 1), 8 (p & 1) bytes into it
    const int tr_q = (lane & 15) >> 2, tr_p = lane & 3, tr_g1 = (lane >> 4) & 1;
    unsigned vb[2];
#pragma unroll
    for (int par = 0; par < 2; ++par)
        vb[par] = lds0 + (unsigned)((4 * h + tr_q) * ROWB + ((((par ^ (tr_q >> 1)) << 2) | ((tr_g1 ^ (tr_q & 1)) << 1) | (tr_p >> 1)) << 4) + ((tr_p & 1) << 3));
    typedef __attribute__((address_space(3))) char lc;
    typedef unsigned u32x4_af __attribute__((ext_vector_type(4)));
    typedef __attribute__((address_space(3))) u32x4_af lu4;

    f32x16 st[NKT];
#pragma unroll
    for (int t = 0; t < NKT; ++t)
#pragma unroll
        for (int j = 0; j < 16; ++j) st[t][j] = 0.f;
/* tile T is about to be read: it has landed in this wave's share when at most the tiles requested after it are outstanding; the    \
   barrier makes that true of every wave's share AND says everyone is done with tile T - 1, whose slot takes tile T + AHEAD */        \
#define AF_STEP_SYNC(TT)                                                                                                           \
    do {                                                                                                                           \
        if constexpr (PAIR) {                                                                                                      \
            if constexpr (((TT) & 1) != 0) break; /* the second tile of a pair: landed and free to read since the pair's barrier */  \
            /* requested so far: tiles up to TT + 5; tiles TT and TT + 1 have landed when only the younger ones are outstanding */   \
            constexpr int last_ = (TT) + 5 < NTOT - 1 ? (TT) + 5 : NTOT - 1;                                                       \
            wait_vm_keep<(last_ - ((TT) + 1)) * DPT>();                                                                            \
            asm volatile("s_barrier" ::: "memory");                                                                                \
            /* everyone is done with tiles TT - 2 and TT - 1: their slots take tiles TT + 6 and TT + 7 */                           \
            if ((TT) + 6 < NTOT) dma_tile((TT) + 6);                                                                               \
            if ((TT) + 7 < NTOT) dma_tile((TT) + 7);                                                                               \
        } else {                                                                                                                   \
        constexpr int younger_ = (TT) + AHEAD - 1 < NTOT ? AHEAD - 1 : NTOT - 1 - (TT);                                            \
        wait_vm_keep<younger_ * DPT>();                                                                                            \
        asm volatile("s_barrier" ::: "memory");                                                                                    \
        if ((TT) + AHEAD < NTOT) dma_tile((TT) + AHEAD);                                                                           \
        }                                                                                                                          \
        AF_STAMP()                                                                                                                 \
        /* (the requests moved behind the step's first fragment reads, their LDS round trip under the ~350 cycles of issuing them:  \
            +-0.0 % on the step, not kept) */                                                                                       \
        if ((xcd_order >> 1) == 1) __builtin_amdgcn_s_sleep(2);                                                                    \
        else if ((xcd_order >> 1) == 2) __builtin_amdgcn_s_sleep(4);                                                               \
        else if ((xcd_order >> 1) == 3) __builtin_amdgcn_s_sleep(8);                                                               \
        else if ((xcd_order >> 1) == 4) __builtin_amdgcn_s_sleep(16);                                                              \
    } while (0)
    // ---- phase 1: scores (tile index a literal: every address below is a per-lane base + an instruction offset) ----
#define AF_A_TILE(t, BF, ACC)                                                                                                      \
    {                                                                                                                              \
        AF_STEP_SYNC(t);                                                                                                           \
        constexpr unsigned so = (unsigned)(((t) % NSLOT) * TILEB);                                                                 \
        constexpr int FG = 4;                                                                                                      \
        uint4 kf[2][FG];                                                                                                           \
        _Pragma("unroll") for (int u = 0; u < FG; ++u) kf[0][u] = __builtin_bit_cast(uint4, *(const lu4*)((const lc*)(size_t)kb[u & 3] + (so + (unsigned)((2 * u) >> 3) * 128u))); \
        _Pragma("unroll") for (int g0 = 0; g0 < KSTEPS; g0 += FG) {                                                                \
            const int cur = (g0 / FG) & 1, nxt = cur ^ 1;                                                                          \
            if (g0 + FG < KSTEPS) {                                                                                                \
                _Pragma("unroll") for (int u = 0; u < FG; ++u) {                                                                   \
                    const int ks = g0 + FG + u;                                                                                    \
                    kf[nxt][u] = __builtin_bit_cast(uint4, *(const lu4*)((const lc*)(size_t)kb[ks & 3] + (so + (unsigned)((2 * ks) >> 3) * 128u)));           \
                }                                                                                                                  \
            }                                                                                                                      \
            __builtin_amdgcn_sched_barrier(0);                                                                                     \
            if (!(PROJ && (t) >= 2 * NKT && (pj.dbg & 4)))                                                                         \
            _Pragma("unroll") for (int u = 0; u < FG; ++u) at_mma<T>(__builtin_bit_cast(tx8, kf[cur][u]), __builtin_bit_cast(tx8, BF[g0 + u]), ACC); \
            __builtin_amdgcn_sched_barrier(0);                                                                                     \
        }                                                                                                                          \
        AF_STAMP()                                                                                                                 \
    }
#define AF_K_TILE(t) AF_A_TILE(t, qf, st[t])
    AF_K_TILE(0) AF_K_TILE(1) AF_K_TILE(2) AF_K_TILE(3) AF_K_TILE(4) AF_K_TILE(5) AF_K_TILE(6) AF_K_TILE(7)
#undef AF_K_TILE
    // ---- phase 2: softmax of the lane's query over its 128 keys (the other 128 are on lane ^ 32) ----
    const float c1 = 1.4426950408889634f * g.scale;  // Cfull^-0.5 * log2(e)
    float mx = st[0][0];
#pragma unroll
    for (int t = 0; t < NKT; ++t)
#pragma unroll
        for (int j = 0; j < 16; ++j) mx = fmaxf(mx, st[t][j]);
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float m = mx * c1;
    float l = 0.f;
    tx8 pf[NKT][2];  // P^T as B operand: k-step s2 of tile t uses registers 8 s2 .. 8 s2 + 7 (key = 16 s2 + 8 (j >> 2) + 4 h + (j & 3))
#pragma unroll
    for (int t = 0; t < NKT; ++t)
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const float p = __builtin_amdgcn_exp2f(fmaf(st[t][j], c1, -m));
            l += p;
            pf[t][j >> 3][j & 7] = (T)p;
        }
    const float ltot = l + __shfl_xor(l, 32, 64);
    AF_STAMP()
    // ---- phase 3: O^T = V^T P^T ----
    f32x16 o[CT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int j = 0; j < 16; ++j) o[ct][j] = 0.f;
#define AF_VREAD(CT0, DST_LO, DST_HI)                                                                                              \
    _Pragma("unroll") for (int cu = 0; cu < VG; ++cu) _Pragma("unroll") for (int s2 = 0; s2 < 2; ++s2) {                            \
        const int ct_ = (CT0) + cu;                                                                                                \
        const lc* a0 = (const lc*)(size_t)vb[ct_ & 1] + (so + (unsigned)((16 * s2) * ROWB + (ct_ >> 1) * 128));                     \
        DST_LO[cu][s2] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)a0);                    \
        DST_HI[cu][s2] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(a0 + 8 * ROWB));       \
    }
#define AF_V_TILE(t)                                                                                                               \
    {                                                                                                                              \
        AF_STEP_SYNC(NKT + (t));                                                                                                   \
        constexpr unsigned so = (unsigned)(((NKT + (t)) % NSLOT) * TILEB);                                                         \
        constexpr int VG = 2;                                                                                                      \
        s16x4 vlo[2][VG][2], vhi[2][VG][2];                                                                                        \
        AF_VREAD(0, vlo[0], vhi[0])                                                                                                \
        _Pragma("unroll") for (int ct = 0; ct < CT; ct += VG) {                                                                    \
            const int cur = (ct / VG) & 1, nxt = cur ^ 1;                                                                          \
            if (ct + VG < CT) {                                                                                                    \
                AF_VREAD(ct + VG, vlo[nxt], vhi[nxt])                                                                              \
            }                                                                                                                      \
            __builtin_amdgcn_sched_barrier(0);                                                                                     \
            _Pragma("unroll") for (int cu = 0; cu < VG; ++cu) _Pragma("unroll") for (int s2 = 0; s2 < 2; ++s2) {                    \
                s16x8 vf;                                                                                                          \
                vf[0] = vlo[cur][cu][s2][0]; vf[1] = vlo[cur][cu][s2][1]; vf[2] = vlo[cur][cu][s2][2]; vf[3] = vlo[cur][cu][s2][3]; \
                vf[4] = vhi[cur][cu][s2][0]; vf[5] = vhi[cur][cu][s2][1]; vf[6] = vhi[cur][cu][s2][2]; vf[7] = vhi[cur][cu][s2][3]; \
                at_mma<T>(__builtin_bit_cast(tx8, vf), pf[t][s2], o[ct + cu]);                                                     \
            }                                                                                                                      \
            __builtin_amdgcn_sched_barrier(0);                                                                                     \
        }                                                                                                                          \
        AF_STAMP()                                                                                                                 \
    }
    AF_V_TILE(0) AF_V_TILE(1) AF_V_TILE(2) AF_V_TILE(3) AF_V_TILE(4) AF_V_TILE(5) AF_V_TILE(6) AF_V_TILE(7)
#undef AF_V_TILE
#undef AF_VREAD
    // ---- normalise: lane = query, registers = channels (j & 3) + 8 (j >> 2) + 4 h ----
    const float inv = 1.0f / ltot;
    if constexpr (!PROJ) {
        if (lse && h == 0) lse[(int64_t)n * S + q_row] = m + log2f(ltot);  // log2-domain log-sum-exp of the scaled scores, for the backward pass
        T* orow = out + at_o_off<C>(g, n) + (int64_t)q_row * g.Cfull;
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
#pragma unroll
            for (int jg = 0; jg < 4; ++jg) {
                typedef T tx4 __attribute__((ext_vector_type(4)));
                tx4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = (T)at_keep_f32(o[ct][jg * 4 + e] * inv);
                *reinterpret_cast<tx4*>(orow + ct * 32 + 8 * jg + 4 * h) = v;
            }
        }
    } else {
        // O^T -> 16-bit B operands of phase 4.  Per 16 channels the lane holds {0-3} + 4 h (jg even) and {8-11} + 4 h (jg odd) as two
        // register pairs X, Y; swapping X's upper half-wave with Y's lower one leaves h = 0 with channels 0-7 and h = 1 with 8-15
        uint4 of[KSTEPS];
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                typedef T tx2 __attribute__((ext_vector_type(2)));
                float w4[4];
#pragma unroll
                for (int p2 = 0; p2 < 4; ++p2) {  // p2 = 0, 1: X (jg = 2 s2); 2, 3: Y (jg = 2 s2 + 1)
                    const int j = (2 * s2 + (p2 >> 1)) * 4 + (p2 & 1) * 2;
                    tx2 v;
                    v[0] = (T)at_keep_f32(o[ct][j] * inv);
                    v[1] = (T)at_keep_f32(o[ct][j + 1] * inv);
                    w4[p2] = __builtin_bit_cast(float, v);
                }
                permlane32_swap(w4[0], w4[2]);
                permlane32_swap(w4[1], w4[3]);
                of[2 * ct + s2] = make_uint4(__builtin_bit_cast(unsigned, w4[0]), __builtin_bit_cast(unsigned, w4[1]), __builtin_bit_cast(unsigned, w4[2]),
                                             __builtin_bit_cast(unsigned, w4[3]));
            }
        AF_STAMP()
        // ---- phase 4: out^T = Wp O^T, the proj matrix streamed in row tiles of 32 couts ----
        f32x16 ot[CT];
#pragma unroll
        for (int t = 0; t < CT; ++t)
#pragma unroll
            for (int j = 0; j < 16; ++j) ot[t][j] = 0.f;
#define AF_W_TILE(t) AF_A_TILE(2 * NKT + (t), of, ot[t])
        AF_W_TILE(0) AF_W_TILE(1) AF_W_TILE(2) AF_W_TILE(3)
        if constexpr (CT > 4) { AF_W_TILE(4) AF_W_TILE(5) AF_W_TILE(6) AF_W_TILE(7) }
#undef AF_W_TILE
        // ---- epilogue (every LDS-DMA has been waited for: ordinary loads / stores from here on) ----
        if (lse && h == 0) lse[(int64_t)n * S + q_row] = m + log2f(ltot);
        // The accumulators hold one pixel per lane and scattered couts: straight from there, every load / store instruction touches 32
        // rows of the tensor with 8 bytes each (first version: 16 us of a 40 us launch, the L1 re-fetching every line eight times).
        // So each wave stages its 32 x C fp32 tile in LDS (the ring is free; rows padded by 16 B: conflict-free both ways) and walks
        // it ROW-wise: lane = 8 consecutive couts of a row, a wave instruction = 64 / (C / 8) whole rows of the residual / output.
        asm volatile("s_barrier" ::: "memory");  // every wave has read the last tile: the ring is the staging area now
        AF_STAMP()
        constexpr int SP = C * 4 + 16;            // staging row pitch (bytes)
        constexpr int LPR = C / 8, RPI = 64 / LPR, NIT = 32 / RPI;  // lanes per row, rows per instruction, instructions per tile
        char* stg = lds + wave * (32 * SP);
        const int col = lane % LPR, rsub = lane / LPR;
        const int64_t tile_off = at_o_off<C>(g, n) + (int64_t)(q_row - r) * g.Cfull + col * 8;
        if (out) {  // the context tensor (the proj conv's weight gradient reads it): the 16-bit operands of phase 4, through the same staging
            constexpr int CP = C * 2 + 16;
#pragma unroll
            for (int ks = 0; ks < KSTEPS; ++ks) *reinterpret_cast<uint4*>(stg + r * CP + (ks * 16 + h * 8) * 2) = of[ks];
            wait_lgkm_all();
            asm volatile("" ::: "memory");
            T* obase = out + tile_off;
#pragma unroll
            for (int i = 0; i < NIT; ++i)
                *reinterpret_cast<uint4*>(obase + (int64_t)(rsub + RPI * i) * g.Cfull) = *reinterpret_cast<const uint4*>(stg + (rsub + RPI * i) * CP + col * 16);
            wait_lgkm_all();  // (read before the fp32 tile overwrites the region)
            asm volatile("" ::: "memory");
        }
#pragma unroll
        for (int t = 0; t < CT; ++t)
#pragma unroll
            for (int jg = 0; jg < 4; ++jg)
                *reinterpret_cast<f32x4*>(stg + r * SP + (t * 32 + 8 * jg + 4 * h) * 4) = f32x4{ot[t][jg * 4], ot[t][jg * 4 + 1], ot[t][jg * 4 + 2], ot[t][jg * 4 + 3]};
        const T* rbase = (const T*)pj.res + tile_off;
        T* dbase = (T*)pj.dst + tile_off;
        if (pj.dbg & 2) {
            if (ot[0][0] == 12345.f) dbase[0] = (T)ot[CT - 1][3];
            return;
        }
        uint4 rv[NIT];
#pragma unroll
        for (int i = 0; i < NIT; ++i) rv[i] = *reinterpret_cast<const uint4*>(rbase + (int64_t)(rsub + RPI * i) * g.Cfull);
        const f32x4 b0 = *reinterpret_cast<const f32x4*>(pj.bias + col * 8), b1 = *reinterpret_cast<const f32x4*>(pj.bias + col * 8 + 4);
        wait_lgkm_all();  // the wave's own staging writes (LDS serves a wave's instructions in order; this also pins the compiler)
        asm volatile("" ::: "memory");
        AF_STAMP()
        float xs[NIT][8];
#pragma unroll
        for (int i = 0; i < NIT; ++i) {
            const char* sp = stg + (rsub + RPI * i) * SP + col * 32;
            const f32x4 a0 = *reinterpret_cast<const f32x4*>(sp), a1 = *reinterpret_cast<const f32x4*>(sp + 16);
            const tx8 y = __builtin_bit_cast(tx8, rv[i]);
            tx8 o8;
#pragma unroll
            for (int e = 0; e < 4; ++e) {  // fp32 conv + bias, + residual, one rounding
                o8[e] = (T)(a0[e] + b0[e] + (float)y[e]);
                o8[4 + e] = (T)(a1[e] + b1[e] + (float)y[4 + e]);
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) xs[i][e] = (float)o8[e];  // (the statistics are those of the values the consumer reads back)
            *reinterpret_cast<uint4*>(dbase + (int64_t)(rsub + RPI * i) * g.Cfull) = __builtin_bit_cast(uint4, o8);
        }
        AF_STAMP()
        if (pj.gn_part && !(pj.dbg & 1)) {
            // this wave's 32 pixels are one statistics tile; a lane's 8 couts are one group of 8 or two groups of 4, the tile's other rows
            // of the same couts sit on the lanes LPR apart: mean, then M2 = sum (x - mean)^2, each by one or two register swaps
            auto col_sum = [&](float v) __attribute__((always_inline)) -> float {
                if constexpr (LPR == 16) {
                    float a = v, b = v;
                    permlane16_swap(a, b);
                    v = a + b;
                }
                float a = v, b = v;
                permlane32_swap(a, b);
                return a + b;
            };
            const bool cg8 = pj.gn_cg == 8;
            float sA = 0.f, sB = 0.f;
#pragma unroll
            for (int i = 0; i < NIT; ++i)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    sA += xs[i][e];
                    sB += xs[i][4 + e];
                }
            if (cg8) sA = sB = sA + sB;
            const float inv_n = cg8 ? 1.f / 256.f : 1.f / 128.f;
            const float mA = col_sum(sA) * inv_n, mB = col_sum(sB) * inv_n;
            float qA = 0.f, qB = 0.f;
#pragma unroll
            for (int i = 0; i < NIT; ++i)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float da = xs[i][e] - mA, db = xs[i][4 + e] - mB;
                    qA = fmaf(da, da, qA);
                    qB = fmaf(db, db, qB);
                }
            if (cg8) qA = qB = qA + qB;
            const float m2A = col_sum(qA), m2B = col_sum(qB);
            const int tile_s = (q_row - r) >> 5, G = C / pj.gn_cg;
            float* po = pj.gn_part + ((int64_t)n * pj.gn_tiles + tile_s) * G * 2;
            if (lane < LPR) {
                if (cg8) *reinterpret_cast<f32x2*>(po + col * 2) = f32x2{mA, m2A};
                else *reinterpret_cast<f32x4*>(po + col * 4) = f32x4{mA, m2A, mB, m2B};
            }
        }
        AF_STAMP()
#ifdef AT_STAMPS
        if (stamping)
            for (int i = 0; i < st_i; ++i) pj.stamps[i] = st_lds[i];
#endif
    }
#undef AF_STAMP
#undef AF_A_TILE
#undef AF_STEP_SYNC
}

// The same whole-row form for launches that do NOT fill the chip (batch < 128): a wave's chain above is 256 dependent MFMAs whatever
// the batch, so here the 256 keys are split over the four waves of a workgroup that owns only 32 queries (8 workgroups per image):
// each wave streams its own two K tiles and two V tiles through a private two-slot LDS area (no workgroup barrier while streaming),
// row maximum and row sum are exchanged through LDS, the four partial O^T tiles are summed through LDS (wave w finishes channels
// [64 w, 64 w + 64) of C = 256).  64 MFMAs per wave instead of 256.
template <int C, typename T = bf16>
__global__ void __launch_bounds__(256) attn_split_kernel(const T* __restrict__ qkv, AttnGeom g, T* __restrict__ out, float* __restrict__ lse) {
    typedef T tx8 __attribute__((ext_vector_type(8)));
    constexpr int S = 256, KSTEPS = C / 16, CT = C / 32;
    constexpr int ROWB = C * 2, TILEB = AT_KT * ROWB;
    constexpr int DPT = TILEB / 1024;  // DMA wave-instructions per tile (this wave alone)
    constexpr int QB = S / 32;         // query blocks per (image, head) row
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int n = blockIdx.x / QB, qb = blockIdx.x % QB;
    const T* base = qkv + at_qkv_off<C>(g, n);
    const int ld = g.ld;
    const int q_row = qb * 32 + r;
    char* ldsW = lds + wave * 2 * TILEB;
    float* xch = reinterpret_cast<float*>(lds + 4 * 2 * TILEB);  // [2][4 waves][32 queries]
    const unsigned lw0 = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(size_t)(__attribute__((address_space(3))) char*)ldsW);
    unsigned koff[DPT], voff[DPT];
#pragma unroll
    for (int i = 0; i < DPT; ++i) {
        const int byte = i * 1024 + lane * 16, row = byte / ROWB, pc = (byte % ROWB) >> 4;
        koff[i] = (unsigned)(row * ld * 2 + (((pc & ~7) | ((pc & 7) ^ (row & 7))) << 4));
        voff[i] = (unsigned)(row * ld * 2 + (((pc & ~7) | ((pc & 7) ^ ((row & 3) << 1))) << 4));
    }
    auto dma_k = [&](int kt, int slot) __attribute__((always_inline)) {
        const char* src = (const char*)(base + (int64_t)(kt * AT_KT) * ld + C);
#pragma unroll
        for (int i = 0; i < DPT; ++i) glds16_hidden_s(src, koff[i], lw0 + (unsigned)(slot * TILEB + i * 1024));
    };
    auto dma_v = [&](int kt, int slot) __attribute__((always_inline)) {
        const char* src = (const char*)(base + (int64_t)(kt * AT_KT) * ld + 2 * C);
#pragma unroll
        for (int i = 0; i < DPT; ++i) glds16_hidden_s(src, voff[i], lw0 + (unsigned)(slot * TILEB + i * 1024));
    };
    dma_k(2 * wave, 0);
    dma_k(2 * wave + 1, 1);
    uint4 qf[KSTEPS];
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks) qf[ks] = *reinterpret_cast<const uint4*>(base + (int64_t)q_row * ld + ks * 16 + h * 8);
    unsigned kb[4];
    {
        const int x = h ^ (r & 7);
#pragma unroll
        for (int i = 0; i < 4; ++i) kb[i] = lw0 + (unsigned)(r * ROWB + (((2 * i) ^ x) << 4));
    }
    const int tr_q = (lane & 15) >> 2, tr_p = lane & 3, tr_g1 = (lane >> 4) & 1;
    unsigned vb[2];
#pragma unroll
    for (int par = 0; par < 2; ++par)
        vb[par] = lw0 + (unsigned)((4 * h + tr_q) * ROWB + ((((par ^ (tr_q >> 1)) << 2) | ((tr_g1 ^ (tr_q & 1)) << 1) | (tr_p >> 1)) << 4) + ((tr_p & 1) << 3));
    typedef __attribute__((address_space(3))) char lc;
    typedef unsigned u32x4_as2 __attribute__((ext_vector_type(4)));
    typedef __attribute__((address_space(3))) u32x4_as2 lu4;
    wait_vm_all();  // both K tiles (and the Q fragments behind them) are in
    // ---- phase 1: this wave's 64 keys ----
    f32x16 st[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
#pragma unroll
        for (int j = 0; j < 16; ++j) st[t][j] = 0.f;
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks) {
            const uint4 kf = __builtin_bit_cast(uint4, *(const lu4*)((const lc*)(size_t)kb[ks & 3] + (unsigned)(t * TILEB + ((2 * ks) >> 3) * 128)));
            at_mma<T>(__builtin_bit_cast(tx8, kf), __builtin_bit_cast(tx8, qf[ks]), st[t]);
        }
    }
    wait_lgkm_all();  // every read of the K tiles has returned: the slots take the V tiles
    dma_v(2 * wave, 0);
    dma_v(2 * wave + 1, 1);
    // ---- phase 2: row maximum and row sum over the four waves ----
    const float c1 = 1.4426950408889634f * g.scale;
    float mx = st[0][0];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int j = 0; j < 16; ++j) mx = fmaxf(mx, st[t][j]);
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    if (h == 0) xch[wave * 32 + r] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(xch[r], xch[32 + r]), fmaxf(xch[64 + r], xch[96 + r]));
    const float m = mx * c1;
    float l = 0.f;
    tx8 pf[2][2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const float p = __builtin_amdgcn_exp2f(fmaf(st[t][j], c1, -m));
            l += p;
            pf[t][j >> 3][j & 7] = (T)p;
        }
    l += __shfl_xor(l, 32, 64);
    if (h == 0) xch[128 + wave * 32 + r] = l;
    __syncthreads();
    const float ltot = (xch[128 + r] + xch[160 + r]) + (xch[192 + r] + xch[224 + r]);  // (the same order in every wave)
    // ---- phase 3: partial O^T over this wave's keys ----
    f32x16 o[CT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int j = 0; j < 16; ++j) o[ct][j] = 0.f;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        if (t == 0) wait_vm_keep<DPT>();
        else wait_vm_keep<0>();
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const lc* a0 = (const lc*)(size_t)vb[ct & 1] + (unsigned)(t * TILEB + (16 * s2) * ROWB + (ct >> 1) * 128);
                const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)a0);
                const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(a0 + 8 * ROWB));
                s16x8 vf;
                vf[0] = lo[0]; vf[1] = lo[1]; vf[2] = lo[2]; vf[3] = lo[3];
                vf[4] = hi[0]; vf[5] = hi[1]; vf[6] = hi[2]; vf[7] = hi[3];
                at_mma<T>(__builtin_bit_cast(tx8, vf), pf[t][s2], o[ct]);
            }
    }
    // ---- sum of the four partial tiles: [wave][ct][j][lane] fp32 over the tile area; wave w finishes ct = w CT / 4 .. ----
    __syncthreads();  // every wave is done with its V tiles
    float* red = reinterpret_cast<float*>(lds);
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int j = 0; j < 16; ++j) red[((wave * CT + ct) * 16 + j) * 64 + lane] = o[ct][j];
    __syncthreads();
    const float inv = 1.0f / ltot;
    if (lse && wave == 0 && h == 0) lse[(int64_t)n * S + q_row] = m + log2f(ltot);
    T* orow = out + at_o_off<C>(g, n) + (int64_t)q_row * g.Cfull;
    constexpr int CPW = CT / 4;  // channel tiles per wave
#pragma unroll
    for (int c = 0; c < CPW; ++c) {
        const int ct = wave * CPW + c;
#pragma unroll
        for (int jg = 0; jg < 4; ++jg) {
            typedef T tx4 __attribute__((ext_vector_type(4)));
            tx4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int j = jg * 4 + e;
                float sum = 0.f;
#pragma unroll
                for (int w = 0; w < 4; ++w) sum += red[((w * CT + ct) * 16 + j) * 64 + lane];
                v[e] = (T)(sum * inv);
            }
            *reinterpret_cast<tx4*>(orow + ct * 32 + 8 * jg + 4 * h) = v;
        }
    }
}

template <int D, typename T>
static int launch_attn_split_t(const T* qkv, const AttnGeom& g, T* out, float* lse, hipStream_t s) {
    const int rows = g.N * g.heads;
    const size_t lds = (size_t)8 * AT_KT * D * 2 + 1024;
    static bool attr_done = false;
    if (!attr_done) {
        DMME_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(attn_split_kernel<D, T>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_done = true;
    }
    hipLaunchKernelGGL((attn_split_kernel<D, T>), dim3((unsigned)(rows * 8)), dim3(256), lds, s, qkv, g, out, lse);
    DMME_CHECK_LAUNCH();
    return DMME_OK;
}

template <int D, typename T, bool PROJ = false>
static int launch_attn_full_t(const T* qkv, const AttnGeom& g, T* out, float* lse, hipStream_t s, const AttnProj& pj = AttnProj{}) {
    const int rows = g.N * g.heads;
    const size_t ring = (size_t)8 * AT_KT * D * 2, stage = (size_t)4 * 32 * (D * 4 + 16);  // (PROJ: the epilogue's fp32 staging tiles lie over the ring)
#ifdef AT_STAMPS
    const size_t lds = (PROJ && stage > ring ? stage : ring) + 1024;  // (+ the diagnostic stamps' buffer)
#else
    const size_t lds = PROJ && stage > ring ? stage : ring;
#endif
    static bool attr_done = false;
    if (!attr_done) {
        DMME_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(attn_full_kernel<D, T, PROJ>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_done = true;
    }
    const int xcd_order = ((!debug_route("no_xcd_order") && rows % 8 == 0) ? 1 : 0) | ((debug_route("attn_sleep", -1) >= 0 ? debug_route("attn_sleep", -1) : rows * 2 >= 256 ? 2 : 0) << 1);  // pacing level (kernel comment): s_sleep 4 per tile where the launch fills the chip
    hipLaunchKernelGGL((attn_full_kernel<D, T, PROJ>), dim3((unsigned)(rows * 2)), dim3(256), lds, s, qkv, g, out, lse, xcd_order, pj);
    DMME_CHECK_LAUNCH();
    return DMME_OK;
}
// whole-row softmax kernel: 256 keys, head width 128 / 256
bool attn_full_takes(int S, int D) {
    return S == 256 && (D == 128 || D == 256) && !getenv("DMME_NO_ATTN_FULL");
}

bool attn_mfma_supported(int dtype, int N, int S, int C) { return attn_heads_mfma_supported(dtype, N, S, C, 1); }
// head width 64 / 128 / 256; 128-query workgroups, or 64-query ones for the 8x8 maps
bool attn_heads_mfma_supported(int dtype, int N, int S, int C, int heads) {
    (void)N;
    if (!is16(dtype) || heads < 1 || C % heads) return false;
    const int D = C / heads;
    return (D == 64 || D == 128 || D == 256) && S >= 64 && S % 64 == 0 && (S % 128 == 0 || S == 64);
}
static AttnGeom attn_geom(int N, int S, int C, int heads) { return AttnGeom{S, 3 * C, C, heads, N, 1.0f / sqrtf((float)C)}; }

template <int D, int NW, typename T>
static int launch_attn_fwd_nw(const T* qkv, const AttnGeom& g, T* out, float* lse, hipStream_t s) {
    const int rows = g.N * g.heads, qblocks = g.S / (32 * NW);
    const size_t lds = (size_t)AT_KT * (D * 2 + 16) + (size_t)AT_KT * (D * 2 + 64) + (D > 128 ? (size_t)32 * NW * (D * 2 + 16) : 0);
    static bool attr_done = false;
    if (!attr_done) {
        DMME_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(attn_mfma_kernel<D, NW, T>), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
        attr_done = true;
    }
    const bool xcd_off = (debug_route("no_xcd_order") != 0);
    const int xcd_order = (!xcd_off && qblocks > 1 && rows % 8 == 0) ? 1 : 0;
    hipLaunchKernelGGL((attn_mfma_kernel<D, NW, T>), dim3((unsigned)(rows * qblocks)), dim3(64 * NW), lds, s, qkv, g, out, lse, xcd_order);
    DMME_CHECK_LAUNCH();
    return DMME_OK;
}
template <int D, typename T>
static int launch_attn_fwd_t(const T* qkv, const AttnGeom& g, T* out, float* lse, hipStream_t s) {
    if (g.S % 128 == 0) return launch_attn_fwd_nw<D, 4, T>(qkv, g, out, lse, s);
    return launch_attn_fwd_nw<D, 2, T>(qkv, g, out, lse, s);
}
int launch_attn_heads_mfma(int dtype, const void* qkv, int N, int S, int C, int heads, void* out, float* lse, hipStream_t s) {
    DMME_REQUIRE(attn_heads_mfma_supported(dtype, N, S, C, heads), DMME_ERR_UNSUPPORTED, "attn_mfma: unsupported shape S=%d C=%d heads=%d", S, C, heads);
    const AttnGeom g = attn_geom(N, S, C, heads);
    if (attn_full_takes(S, C / heads) && N * heads * 2 < 256 && !debug_route("no_attn_split")) {  // the launch would leave CUs idle: keys split over the waves
        if (dtype == DMME_F16)
            return C / heads == 256 ? launch_attn_split_t<256, f16>((const f16*)qkv, g, (f16*)out, lse, s) : launch_attn_split_t<128, f16>((const f16*)qkv, g, (f16*)out, lse, s);
        return C / heads == 256 ? launch_attn_split_t<256, bf16>((const bf16*)qkv, g, (bf16*)out, lse, s) : launch_attn_split_t<128, bf16>((const bf16*)qkv, g, (bf16*)out, lse, s);
    }
    if (attn_full_takes(S, C / heads)) {
        if (dtype == DMME_F16)
            return C / heads == 256 ? launch_attn_full_t<256, f16>((const f16*)qkv, g, (f16*)out, lse, s) : launch_attn_full_t<128, f16>((const f16*)qkv, g, (f16*)out, lse, s);
        return C / heads == 256 ? launch_attn_full_t<256, bf16>((const bf16*)qkv, g, (bf16*)out, lse, s) : launch_attn_full_t<128, bf16>((const bf16*)qkv, g, (bf16*)out, lse, s);
    }
    if (dtype == DMME_F16) {
        switch (C / heads) {
            case 256: return launch_attn_fwd_t<256, f16>((const f16*)qkv, g, (f16*)out, lse, s);
            case 128: return launch_attn_fwd_t<128, f16>((const f16*)qkv, g, (f16*)out, lse, s);
            default: return launch_attn_fwd_t<64, f16>((const f16*)qkv, g, (f16*)out, lse, s);
        }
    }
    switch (C / heads) {
        case 256: return launch_attn_fwd_t<256, bf16>((const bf16*)qkv, g, (bf16*)out, lse, s);
        case 128: return launch_attn_fwd_t<128, bf16>((const bf16*)qkv, g, (bf16*)out, lse, s);
        default: return launch_attn_fwd_t<64, bf16>((const bf16*)qkv, g, (bf16*)out, lse, s);
    }
}
int launch_attn_mfma(int dtype, const void* qkv, int N, int S, int C, void* out, float* lse, hipStream_t s) {
    return launch_attn_heads_mfma(dtype, qkv, N, S, C, 1, out, lse, s);
}
// the single-head block with its proj conv + residual in the same launch (attn_full_kernel<.., PROJ>): wherever launch_attn_heads_mfma
// would pick the whole-row kernel
static bool attn_proj_shape_ok(int dtype, int N, int S, int C, int gn_cg, int gn_tiles) {
    if (!is16(dtype) || S != 256 || (C != 128 && C != 256) || N * 2 < 256) return false;  // the whole-row kernel's shapes, a launch that fills the chip
    return gn_tiles == 0 || ((gn_cg == 4 || gn_cg == 8) && C % gn_cg == 0 && gn_tiles == S / 32);
}
// (the plan's decision: shapes + the route switches, read when the plan is built - the launch itself checks shapes only)
bool attn_proj_fusable(int dtype, int N, int S, int C, int gn_cg, int gn_tiles) {
    if (!attn_full_takes(S, C) || debug_route("no_attn_proj")) return false;
    return attn_proj_shape_ok(dtype, N, S, C, gn_cg, gn_tiles);
}
int launch_attn_proj(int dtype, const void* qkv, int N, int S, int C, void* ctx, float* lse, const void* w, const float* bias, const void* res, void* dst,
                     float* gn_part, int gn_tiles, int gn_cg, hipStream_t s, long long* stamps) {
    DMME_REQUIRE(attn_proj_shape_ok(dtype, N, S, C, gn_part ? gn_cg : 0, gn_part ? gn_tiles : 0) && w && bias && res && dst, DMME_ERR_UNSUPPORTED,
                 "attn_proj: unsupported shape N=%d S=%d C=%d (statistics tiles %d, %d channels per group)", N, S, C, gn_tiles, gn_cg);
    const AttnGeom g = attn_geom(N, S, C, 1);
    const AttnProj pj{w, bias, res, dst, gn_part, gn_tiles, gn_cg, stamps, debug_route("attn_proj_dbg")};
    if (dtype == DMME_F16)
        return C == 256 ? launch_attn_full_t<256, f16, true>((const f16*)qkv, g, (f16*)ctx, lse, s, pj) : launch_attn_full_t<128, f16, true>((const f16*)qkv, g, (f16*)ctx, lse, s, pj);
    return C == 256 ? launch_attn_full_t<256, bf16, true>((const bf16*)qkv, g, (bf16*)ctx, lse, s, pj) : launch_attn_full_t<128, bf16, true>((const bf16*)qkv, g, (bf16*)ctx, lse, s, pj);
}

// =====================================================================================
// Backward.  With P = softmax(Q K^T s) (s = C^-0.5), O = P V and dO given:
//   dV = P^T dO,  dP = dO V^T,  dS = P o (dP - delta),  delta_i = <dO_i, O_i>,
//   dQ = s dS K,  dK = s dS^T Q.
// Kernel 1 (attn_bwd_scores) recomputes the transposed score tiles S^T = K Q^T and dP^T = V dO^T
// on the matrix cores with the query on the lane (as in the forward kernel), so the softmax
// row terms (saved log-sum-exp, delta) are per-lane scalars, and writes P and dS (bf16,
// [N][S][S], row = query).  The three remaining products are plain per-image GEMMs
// (attn_bgemm): dV = P^T dO and dK = s dS^T Q contract over the ROW index of two row-major
// matrices (both fragments via transposed LDS reads), dQ = s dS K is a row-major A times a
// transposed-read B.
template <int C, int NW, typename T>
__global__ void __launch_bounds__(64 * NW, 1) attn_bwd_scores_kernel(const T* __restrict__ qkv, const T* __restrict__ O,
                                                              const T* __restrict__ dO, const float* __restrict__ lse, AttnGeom g,
                                                              T* __restrict__ P, T* __restrict__ dS) {
    constexpr int AT_QB = 32 * NW, NT = 64 * NW;
    const int S = g.S, ld = g.ld;
    constexpr int KSTEPS = C / 16;
    constexpr int KP = C * 2 + 16;
    __shared__ __attribute__((aligned(16))) char lds[2 * AT_KT * KP];
    char* ldsK = lds;
    char* ldsV = lds + AT_KT * KP;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int qblocks = S / AT_QB;
    const int n = blockIdx.x / qblocks, qb = blockIdx.x % qblocks;  // n: the (image, head) row
    const T* base = qkv + at_qkv_off<C>(g, n);
    const int64_t orow = at_o_off<C>(g, n);
    const int q_row = qb * AT_QB + wave * 32 + r;
    uint4 qf[KSTEPS], dof[KSTEPS];
    float dpart = 0.f;
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks) {
        qf[ks] = *reinterpret_cast<const uint4*>(base + (int64_t)q_row * ld + ks * 16 + h * 8);
        dof[ks] = *reinterpret_cast<const uint4*>(dO + orow + (int64_t)q_row * g.Cfull + ks * 16 + h * 8);
        const uint4 ov = *reinterpret_cast<const uint4*>(O + orow + (int64_t)q_row * g.Cfull + ks * 16 + h * 8);
        const typename Vec8<T>::type a = __builtin_bit_cast(typename Vec8<T>::type, dof[ks]), b = __builtin_bit_cast(typename Vec8<T>::type, ov);
#pragma unroll
        for (int j = 0; j < 8; ++j) dpart = fmaf((float)a[j], (float)b[j], dpart);
    }
    const float delta = dpart + __shfl_xor(dpart, 32, 64);
    const float L = lse[(int64_t)n * S + q_row];
    const float c1 = 1.4426950408889634f * g.scale;
    T* Prow = P + ((int64_t)n * S + q_row) * S;
    T* dSrow = dS + ((int64_t)n * S + q_row) * S;

    // K / V tiles a tile ahead, through registers: one wave per SIMD has nothing else to put under a tile's global round trip (the
    // load-store-barrier form spent 8 exposed round trips per workgroup: 50 us per launch for 8.6 GFLOP)
    constexpr int NV = AT_KT * (C / 8) / NT;
    static_assert(NV * NT == AT_KT * (C / 8), "whole vectors per thread");
    typedef unsigned atb_u32x4 __attribute__((ext_vector_type(4)));  // (plain vectors: arrays of HIP's uint4 struct stayed in scratch memory)
    atb_u32x4 kreg[NV], vreg[NV];
#define ATB_FETCH(K0)                                                                     \
    _Pragma("unroll") for (int i = 0; i < NV; ++i) {                                       \
        const int u = tid + i * NT, row = u / (C / 8), cu = u % (C / 8);                   \
        const T* src = base + (int64_t)((K0) + row) * ld + cu * 8;                      \
        kreg[i] = *reinterpret_cast<const atb_u32x4*>(src + C);                            \
        vreg[i] = *reinterpret_cast<const atb_u32x4*>(src + 2 * C);                        \
    }
    ATB_FETCH(0)
    for (int k0 = 0; k0 < S; k0 += AT_KT) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int u = tid + i * NT, row = u / (C / 8), cu = u % (C / 8);
            *reinterpret_cast<atb_u32x4*>(ldsK + row * KP + cu * 16) = kreg[i];
            *reinterpret_cast<atb_u32x4*>(ldsV + row * KP + cu * 16) = vreg[i];
        }
        __syncthreads();
        {   // (unconditional: the last trip re-requests its own tile - as a branch the two arrays went to scratch memory)
            const int kn = k0 + AT_KT < S ? k0 + AT_KT : k0;
            ATB_FETCH(kn)
        }
        f32x16 st, dp;
#pragma unroll
        for (int j = 0; j < 16; ++j) st[j] = dp[j] = 0.f;
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks) {
            const uint4 kf = *reinterpret_cast<const uint4*>(ldsK + r * KP + ks * 32 + h * 16);
            const uint4 vf = *reinterpret_cast<const uint4*>(ldsV + r * KP + ks * 32 + h * 16);
            st = mma16v<T>(kf, qf[ks], st);
            dp = mma16v<T>(vf, dof[ks], dp);
        }
        // registers j <-> key k0 + (j&3) + 8*(j>>2) + 4*h : four consecutive keys per register quad
        typedef T bf16x4 __attribute__((ext_vector_type(4)));  // (four operands of the kernel's 16-bit type)
#pragma unroll
        for (int jg = 0; jg < 4; ++jg) {
            bf16x4 pv, dv;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float p = exp2f(fmaf(st[jg * 4 + e], c1, -L));
                pv[e] = (T)p;
                dv[e] = (T)(p * (dp[jg * 4 + e] - delta));
            }
            *reinterpret_cast<bf16x4*>(Prow + k0 + 8 * jg + 4 * h) = pv;
            *reinterpret_cast<bf16x4*>(dSrow + k0 + 8 * jg + 4 * h) = dv;
        }
    }
#undef ATB_FETCH
}

// per-image GEMM  out[m][n] = alpha * sum_k A(m,k) B(k,n),  M = K = S, N = C (T in, fp32 accumulate)
//   TRANS_A = 1: A(m,k) = X[k][m]  (X row-major [S][S], ldx = S)      -- dV, dK
//   TRANS_A = 0: A(m,k) = X[m][k]                                      -- dQ
//   B(k,n) = Y[k][n] (row-major, ldy), fragments by transposed LDS reads.
// One workgroup = one (image, head) row x 64 rows of the output; the 4 waves split the C columns (C >= 128), or 2 x 2 over
// (rows, columns) for C = 64.  Y and out are addressed through the head view: Y_O = 1 reads an output-layout tensor (dO),
// else a qkv-layout one at column offset ycol; out is always qkv-layout (dqkv) at column offset ocol.
template <int C, int TRANS_A, int Y_O, typename T>
__global__ void __launch_bounds__(256) attn_bgemm_kernel(const T* __restrict__ X, const T* __restrict__ Y, int ycol, AttnGeom g, float alpha,
                                                         T* __restrict__ out, int ocol) {
    constexpr int WM = C >= 128 ? 1 : 2;   // waves along the 64 output rows
    constexpr int MI = 2 / WM;             // 32-row tiles per wave
    constexpr int WN = C / (4 / WM);       // columns per wave
    constexpr int NI = WN / 32;            // 32-column tiles per wave
    const int S = g.S, ldx = S, ldy = Y_O ? g.Cfull : g.ld, ldo = g.ld;
    constexpr int YP = C * 2 + 64;   // Y tile pitch: 4 rows of a transposed read on disjoint banks
    constexpr int XP_T = 64 * 2 + 64;  // X tile [32 k][64 m] for transposed reads
    constexpr int XP_N = 32 * 2 + 16;  // X tile [64 m][32 k] for row reads
    __shared__ __attribute__((aligned(16))) char lds[32 * YP + (TRANS_A ? 32 * XP_T : 64 * XP_N)];
    char* ldsY = lds;
    char* ldsX = lds + 32 * YP;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int mblocks = S / 64;
    const int n = blockIdx.x / mblocks, m0 = (blockIdx.x % mblocks) * 64;
    const int wm = WM == 1 ? 0 : wave >> 1, wn = WM == 1 ? wave : wave & 1;
    const T* Xi = X + (int64_t)n * S * S;
    const T* Yi = Y + (Y_O ? at_o_off<C>(g, n) : at_qkv_off<C>(g, n) + ycol);
    const int tr_q = (lane & 15) >> 2, tr_p = lane & 3, tr_g1 = (lane >> 4) & 1;
    f32x16 acc[MI][NI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int j = 0; j < 16; ++j) acc[mi][ni][j] = 0.f;

    // both operand tiles a k-step ahead, through registers (as attn_bwd_scores_kernel: nothing else hides the round trip here)
    typedef unsigned bg_u32x4 __attribute__((ext_vector_type(4)));
    constexpr int NVY = 32 * (C / 8) / 256;
    static_assert(NVY * 256 == 32 * (C / 8), "whole vectors per thread");
    bg_u32x4 yreg[NVY], xreg;
#define BG_FETCH(K0)                                                                                             \
    {                                                                                                            \
        _Pragma("unroll") for (int i = 0; i < NVY; ++i) {                                                        \
            const int u = tid + i * 256, row = u / (C / 8), cu = u % (C / 8);                                    \
            yreg[i] = *reinterpret_cast<const bg_u32x4*>(Yi + (int64_t)((K0) + row) * ldy + cu * 8);             \
        }                                                                                                        \
        if (TRANS_A) { /* X[k0 + row][m0 .. m0+63] */                                                            \
            const int row = tid >> 3, cu = tid & 7;                                                              \
            xreg = *reinterpret_cast<const bg_u32x4*>(Xi + (int64_t)((K0) + row) * ldx + m0 + cu * 8);           \
        } else { /* X[m0 + row][k0 .. k0+31] */                                                                  \
            const int row = tid >> 2, cu = tid & 3;                                                              \
            xreg = *reinterpret_cast<const bg_u32x4*>(Xi + (int64_t)(m0 + row) * ldx + (K0) + cu * 8);           \
        }                                                                                                        \
    }
    BG_FETCH(0)
    for (int k0 = 0; k0 < S; k0 += 32) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < NVY; ++i) {
            const int u = tid + i * 256, row = u / (C / 8), cu = u % (C / 8);
            *reinterpret_cast<bg_u32x4*>(ldsY + row * YP + cu * 16) = yreg[i];
        }
        if (TRANS_A) {
            const int row = tid >> 3, cu = tid & 7;
            *reinterpret_cast<bg_u32x4*>(ldsX + row * XP_T + cu * 16) = xreg;
        } else {
            const int row = tid >> 2, cu = tid & 3;
            *reinterpret_cast<bg_u32x4*>(ldsX + row * XP_N + cu * 16) = xreg;
        }
        __syncthreads();
        {
            const int kn = k0 + 32 < S ? k0 + 32 : k0;  // (unconditional: the last trip re-requests its own tiles)
            BG_FETCH(kn)
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            s16x8 af[MI];
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) {
                const int mt = wm * MI + mi;  // 32-row tile inside the 64-row block
                if (TRANS_A) {
                    const char* p0 = ldsX + (16 * ks + 8 * h + tr_q) * XP_T + (mt * 32 + 16 * tr_g1 + 4 * tr_p) * 2;
                    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p0);
                    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p0 + 4 * XP_T));
                    af[mi][0] = lo[0]; af[mi][1] = lo[1]; af[mi][2] = lo[2]; af[mi][3] = lo[3];
                    af[mi][4] = hi[0]; af[mi][5] = hi[1]; af[mi][6] = hi[2]; af[mi][7] = hi[3];
                } else {
                    af[mi] = __builtin_bit_cast(s16x8, *reinterpret_cast<const uint4*>(ldsX + (mt * 32 + r) * XP_N + ks * 32 + h * 16));
                }
            }
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) {
                const char* p0 = ldsY + (16 * ks + 8 * h + tr_q) * YP + (wn * WN + ni * 32 + 16 * tr_g1 + 4 * tr_p) * 2;
                const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p0);
                const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p0 + 4 * YP));
                s16x8 bfr;
                bfr[0] = lo[0]; bfr[1] = lo[1]; bfr[2] = lo[2]; bfr[3] = lo[3];
                bfr[4] = hi[0]; bfr[5] = hi[1]; bfr[6] = hi[2]; bfr[7] = hi[3];
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
                    acc[mi][ni] = mma16v<T>(af[mi], bfr, acc[mi][ni]);
            }
        }
    }
#undef BG_FETCH
    T* Oi = out + at_qkv_off<C>(g, n) + ocol;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const int m = m0 + (wm * MI + mi) * 32 + (j & 3) + 8 * (j >> 2) + 4 * h;
                Oi[(int64_t)m * ldo + wn * WN + ni * 32 + r] = (T)(acc[mi][ni][j] * alpha);
            }
}

bool attn_bwd_mfma_supported(int dtype, int N, int S, int C) { return is16(dtype) && attn_heads_mfma_supported(dtype, N, S, C, 1); }

template <int D, typename T>
static int launch_attn_bwd_t(const T* qkv, const T* O, const T* dO, const float* lse, const AttnGeom& g, T* P, T* dS, T* dqkv,
                             hipStream_t s) {
    const int rows = g.N * g.heads, S = g.S;
    if (S % 128 == 0)
        hipLaunchKernelGGL((attn_bwd_scores_kernel<D, 4, T>), dim3((unsigned)(rows * (S / 128))), dim3(256), 0, s, qkv, O, dO, lse, g, P, dS);
    else
        hipLaunchKernelGGL((attn_bwd_scores_kernel<D, 2, T>), dim3((unsigned)(rows * (S / 64))), dim3(128), 0, s, qkv, O, dO, lse, g, P, dS);
    DMME_CHECK_LAUNCH();
    const dim3 grid((unsigned)(rows * (S / 64)));
    // dV = P^T dO ; dK = s dS^T Q ; dQ = s dS K
    hipLaunchKernelGGL((attn_bgemm_kernel<D, 1, 1, T>), grid, dim3(256), 0, s, P, dO, 0, g, 1.0f, dqkv, 2 * D);
    DMME_CHECK_LAUNCH();
    hipLaunchKernelGGL((attn_bgemm_kernel<D, 1, 0, T>), grid, dim3(256), 0, s, dS, qkv, 0, g, g.scale, dqkv, D);
    DMME_CHECK_LAUNCH();
    hipLaunchKernelGGL((attn_bgemm_kernel<D, 0, 0, T>), grid, dim3(256), 0, s, dS, qkv, D, g, g.scale, dqkv, 0);
    DMME_CHECK_LAUNCH();
    return DMME_OK;
}

// P, dS: N*heads*S*S bf16 each
int launch_attn_heads_bwd_mfma(int dtype, const void* qkv, const void* O, const void* dO, const float* lse, int N, int S, int C, int heads, void* P,
                               void* dS, void* dqkv, hipStream_t s) {
    DMME_REQUIRE(attn_heads_mfma_supported(dtype, N, S, C, heads), DMME_ERR_UNSUPPORTED, "attn_bwd_mfma: unsupported shape");
    const AttnGeom g = attn_geom(N, S, C, heads);
    if (dtype == DMME_F16) {
        switch (C / heads) {
            case 256: return launch_attn_bwd_t<256, f16>((const f16*)qkv, (const f16*)O, (const f16*)dO, lse, g, (f16*)P, (f16*)dS, (f16*)dqkv, s);
            case 128: return launch_attn_bwd_t<128, f16>((const f16*)qkv, (const f16*)O, (const f16*)dO, lse, g, (f16*)P, (f16*)dS, (f16*)dqkv, s);
            default: return launch_attn_bwd_t<64, f16>((const f16*)qkv, (const f16*)O, (const f16*)dO, lse, g, (f16*)P, (f16*)dS, (f16*)dqkv, s);
        }
    }
    switch (C / heads) {
        case 256: return launch_attn_bwd_t<256, bf16>((const bf16*)qkv, (const bf16*)O, (const bf16*)dO, lse, g, (bf16*)P, (bf16*)dS, (bf16*)dqkv, s);
        case 128: return launch_attn_bwd_t<128, bf16>((const bf16*)qkv, (const bf16*)O, (const bf16*)dO, lse, g, (bf16*)P, (bf16*)dS, (bf16*)dqkv, s);
        default: return launch_attn_bwd_t<64, bf16>((const bf16*)qkv, (const bf16*)O, (const bf16*)dO, lse, g, (bf16*)P, (bf16*)dS, (bf16*)dqkv, s);
    }
}
int launch_attn_bwd_mfma(int dtype, const void* qkv, const void* O, const void* dO, const float* lse, int N, int S, int C, void* P, void* dS,
                         void* dqkv, hipStream_t s) {
    return launch_attn_heads_bwd_mfma(dtype, qkv, O, dO, lse, N, S, C, 1, P, dS, dqkv, s);
}

}  // namespace dmme
