// 1x1 convolution with the ACTIVATIONS STATIONARY IN REGISTERS: nn.Conv2d(kernel_size=1) of Attention.qkv_proj / proj and
// ResBlock.residual (models/ddpm.py:51-52,109) at K = Cin = 128 / 256, the GroupNorm apply in front of qkv_proj included.
//
// These GEMMs have a SHORT K: a 128 x 128 output tile is 64 MFMAs per wave behind 128 KB of operands, so a tiled kernel that loads,
// stages and synchronises per tile (conv1x1_pipe.hip: 1536 workgroups for qkv, each re-reading and re-normalising its activation
// tile for one of six cout tiles) spends its life in exposed round trips: 46 us for 12.9 GFLOP / 67 MB.  Here a workgroup of eight
// waves owns 128 pixels for ALL couts:
//   * its activation tile (128 px x K) comes in once by LDS-DMA, is normalised in place by all 512 threads (rolled loop), is read into
//     MFMA fragments (64 px x K per MFMA wave: 16 x 2 x 4 registers at K = 256) and never touched again;
//   * the weights stream through an LDS ring in units of 64 couts x K (32 KB at K = 256) by LDS-DMA, up to two units in flight; the four
//     MFMA waves (pixel half, cout half of the unit) compute 64 px x 32 couts per unit with one weight fragment read per two MFMAs.
//     The MFMAs run transposed (weights as the A operand), so a half-wave register swap (v_permlane32_swap) leaves every lane with 8
//     consecutive couts of one pixel: the unit goes to an LDS stage as 16-byte vectors, bias being the accumulators' initial value;
//   * the other four waves are the STORE TEAM: they take each staged unit to memory in whole 128-byte rows, add the residual and keep
//     the GroupNorm partials (32-pixel tiles, mean then M2 by two wave reductions).  Measured on the first version (one team doing
//     both): with every CU storing at the same moment the stores are accepted at the chip's HBM write rate and a wave that issues them
//     stalls for the whole drain (0.9 us per unit, MFMAs idle); a second wave per SIMD takes that stall instead.
// One workgroup barrier per unit.  The MFMA waves' vector-memory queue holds only their weight DMA, so every wait is a counted
// `s_waitcnt vmcnt(N)` (retired in order); the store team uses ordinary loads / stores.  qkv (B = 128): 46 -> 27 us, proj 20.6 -> 17.
#include <stdio.h>

#include "conv_common.h"

namespace dmme {

typedef unsigned u32x4_as __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) u32x4_as lds_u32x4_as;
typedef __attribute__((address_space(3))) f32x4 lds_f32x4_as;

// One staged unit: [128 px][64 couts], bf16 (16-byte slots XOR-swizzled by the pixel row) - or fp32 where a residual is added before
// the one rounding (32-byte slots swizzled the same way)
__host__ __device__ constexpr int as_stage_bytes(bool res) { return res ? 128 * 256 : 128 * 128; }
__host__ __device__ constexpr int as_ring(bool res) { return res ? 2 : 3; }  // weight units in LDS (the fp32 stage takes the third slot's room)

__host__ __device__ inline int as_fold_bytes(int Cout) { return (Cout * 4 + 1023) & ~1023; }

// v_permlane32_swap: lanes 32-63 of x <-> lanes 0-31 of y.  As inline asm: this hipcc folds SEVERAL calls of the builtin
// (__builtin_amdgcn_permlane32_swap) with different operands into one instruction and broadcasts its result (seen in the ISA and in the
// output); the compiler's hazard recogniser does not see inside asm, so the caller puts as_mfma_drain() between MFMAs and a swap
// of their results.
__device__ __forceinline__ void as_swap32(float& x, float& y) {
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(x), "+v"(y));  // (a VALU result needs 2 wait states before a swap reads it)
#endif
}
__device__ __forceinline__ void as_mfma_drain() {  // >= 19 wait states: the longest MFMA-result-to-VALU-read distance of the ISA
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");
#endif
}

// sums over the lanes that share (lane & 7) - the 8 pixel rows of a wave-instruction that hold the same 16-byte channel slot
__device__ __forceinline__ float as_sum_rows(float s) {
    s += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, s), 0x128, 0xf, 0xf, true));  // row_ror:8 = lane ^ 8
    s += __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, s), 0x401F));                    // lane ^ 16
    float lo = s, hi = s;
    as_swap32(lo, hi);  // lo = lanes 0-31's value in every lane, hi = lanes 32-63's
    return lo + hi;     // lane ^ 32
}
__device__ __forceinline__ float as_sum_slots(float s, int nv) {  // + the group's other 16-byte slots (nv = 1, 2, 4 adjacent lanes)
    if (nv >= 2) s += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, s), 0xB1, 0xf, 0xf, true));
    if (nv == 4) s += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, s), 0x4E, 0xf, 0xf, true));
    return s;
}

// workgroup barrier that is also a compiler barrier (the builtin is IntrNoMem: LDS accesses may be moved across it)
__device__ __forceinline__ void as_barrier() {
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("s_barrier" ::: "memory");
#endif
}

// GroupNorm affine (+ SiLU) (+ dropout multiplier) of the landed activation tile, in place, by all 512 threads: a ROLLED loop (unrolled
// over the fragments it was 50 KB of code per launch and spilled); the same operations in the same order as prologue_vec<bf16>.
// par: [3][K] scale, shift, dropout multiplier of the tile's image (1, 0, 1 where absent: x * 1 + 0 and * 1 are exact)
template <int K, typename T>
__device__ __forceinline__ void as_prologue_tile(char* ldsA, const float* par, int tid, bool silu, bool dm) {
    typedef typename Vec8<T>::type bf16x8;  // (8 operands of the 16-bit type: bf16 or IEEE half)
    typedef __attribute__((address_space(3))) u32x4_as lv4;
    typedef __attribute__((address_space(3))) f32x4 lf4;
    const lds_c* P3 = (const lds_c*)par;
#pragma unroll 1
    for (int it = tid; it < 128 * (K / 8); it += 512) {  // 16-byte vector `it`: chunk-major, then row, then LDS piece
        const int piece = it & 7, row = (it >> 3) & 127, c = it >> 10;
        const int ch = c * 64 + ((piece ^ ((row >> 1) & 7)) << 3);  // the source piece this LDS piece holds
        lv4* p = (lv4*)(lds_c*)(ldsA + it * 16);
        bf16x8 x = __builtin_bit_cast(bf16x8, *p);
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; e += 4) {
            const f32x4 sc = *(const lf4*)(P3 + (ch + e) * 4), sh = *(const lf4*)(P3 + (K + ch + e) * 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) v[e + j] = fmaf((float)x[e + j], sc[j], sh[j]);
        }
        if (silu) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = silu_fast(v[e]);
        }
        if (dm) {
#pragma unroll
            for (int e = 0; e < 8; e += 4) {
                const f32x4 m = *(const lf4*)(P3 + (2 * K + ch + e) * 4);
#pragma unroll
                for (int j = 0; j < 4; ++j) v[e + j] *= m[j];
            }
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) x[e] = (T)v[e];
        *p = __builtin_bit_cast(u32x4_as, x);
    }
}

// 8 waves: 0-3 stream the weights and run the MFMAs, 4-7 take each finished unit out of LDS to memory (see the file comment)
template <int KCH, bool RES, typename T = bf16>
__global__ void __launch_bounds__(512) conv1x1_as_kernel(ConvArgs a, int HW, int NU) {
    typedef typename Vec8<T>::type bf16x8;  // (8 operands of the 16-bit type: bf16 or IEEE half)
    constexpr int K = 64 * KCH, NKS = 4 * KCH;
    constexpr int RING = as_ring(RES), STAGE = as_stage_bytes(RES);
    constexpr int U_BYTES = 64 * KCH * ROW_DATA;  // one unit of weights: [chunk][64 cout rows][128 B]
    constexpr int DPU = 2 * KCH;                  // weight DMA wave-instructions per unit and MFMA wave (8 rows of 128 B each)
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#define AS_STAMP(K) do { if (a.stamps && blockIdx.x < 2 && (threadIdx.x & 255) == 0) a.stamps[blockIdx.x * 32 + (threadIdx.x >> 8) * 16 + (K)] = (long long)wall_clock64(); } while (0)
    AS_STAMP(0);
    // (the tiles of image n dealt to XCD n % 8, where the whole-row attention kernel puts the image's query blocks - so that the K / V
    //  rows written here would be read back from that XCD's L2: +-0.1 % on the step, not kept; that kernel does not wait for K / V bytes)
    const int p0 = blockIdx.x * 128;
    float* foldL = reinterpret_cast<float*>(lds + RING * U_BYTES);
    char* stageL = lds + RING * U_BYTES + as_fold_bytes(a.Cout);  // [2][STAGE]
    // until the fragments are in registers: the activation tile [chunk][128 pixel rows][128 B] (2 units' worth) lies over ring slots
    // 1 and 2 (over the stage with a two-slot ring), the [3][K] GroupNorm affine / dropout rows over the other of the two
    char* ldsA = RES ? stageL : lds + U_BYTES;
    const bool has_pro = a.scale || a.dmask || a.pro_silu;
    float* parL = reinterpret_cast<float*>(RES ? lds + U_BYTES : stageL);

    if (wave >= 4) {
        // =================================== the store team ===================================
        const int t = tid - 256, sw = wave - 4;
        // bias (+ the uniform time row) and the prologue rows of the tile's image (host-checked: one image per tile) into LDS
        for (int c = t; c < a.Cout; c += 256) foldL[c] = (a.bias ? a.bias[c] : 0.f) + (a.tproj ? a.tproj[c] : 0.f);
        if (has_pro && t < K) {
            const int n = p0 / HW, so = n * K + t;
            float sc = 1.f, sh = 0.f;
            if (a.has_gni) {  // the norm in front of this conv is finished here, from its producers' partials
                gn_in_scale_shift(a, n, t, K, p0 == n * HW, sc, sh);
            } else if (a.scale) {
                sc = a.scale[so];
                sh = a.shift[so];
            }
            parL[t] = sc;
            parL[K + t] = sh;
            parL[2 * K + t] = a.dmask ? a.dmask[so] : 1.f;
        }
        wait_lgkm_all();
        as_barrier();  // (1) tile + unit 0 landed, bias / rows written
        if (has_pro) {
            as_prologue_tile<K, T>(ldsA, parL, tid, a.pro_silu != 0, a.dmask != nullptr);
            wait_lgkm_all();
            as_barrier();  // (1b) tile normalised
        }
        as_barrier();  // (2) fragments in registers
        // unit u: this wave takes pixel rows 32 sw .. 32 sw + 31; lane = (row & 7 = lane >> 3, 16-byte output slot = lane & 7), four row blocks
        const int slot8 = lane & 7;
        const T* res = (const T*)a.res1;
        T* dst = (T*)a.dst;
        const int cgs = a.gn_cg, nv = cgs >> 3;
        int64_t goff[4];
        int soff[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int px = 32 * sw + 8 * q + (lane >> 3);
            goff[q] = (int64_t)(p0 + px) * a.Cout + slot8 * 8;
            soff[q] = RES ? px * 256 + ((slot8 ^ (px & 7)) << 5) : px * 128 + ((slot8 ^ (px & 7)) << 4);
        }
#pragma unroll 1
        for (int u = 0; u < NU; ++u) {
            uint4 rv[4];
            if constexpr (RES) {
#pragma unroll
                for (int q = 0; q < 4; ++q) rv[q] = *reinterpret_cast<const uint4*>(res + goff[q] + u * 64);
            }
            as_barrier();  // unit u is staged (its own LDS reads of unit u - 1 fed the stores issued above: they have returned)
            if (u < 3) AS_STAMP(4 + 3 * u);
            const char* st = stageL + (u & 1) * STAGE;
            float xs[4][8];
            float s1 = 0.f, s1b = 0.f;  // (s1b: the vector's upper half where a group is 4 channels)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                uint4 raw;
                if constexpr (RES) {  // fp32 conv + bias, + residual, one rounding
                    const f32x4 v0 = *reinterpret_cast<const f32x4*>(st + soff[q]), v1 = *reinterpret_cast<const f32x4*>(st + soff[q] + 16);
                    const bf16x8 y = __builtin_bit_cast(bf16x8, rv[q]);
                    bf16x8 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        o[e] = (T)(v0[e] + (float)y[e]);
                        o[4 + e] = (T)(v1[e] + (float)y[4 + e]);
                    }
                    raw = __builtin_bit_cast(uint4, o);
                } else {
                    raw = *reinterpret_cast<const uint4*>(st + soff[q]);
                }
                *reinterpret_cast<uint4*>(dst + goff[q] + u * 64) = raw;
                if (a.gn_part) {
                    const bf16x8 o = __builtin_bit_cast(bf16x8, raw);
#pragma unroll
                    for (int e = 0; e < 8; ++e) {  // statistics of the values the consumer reads back
                        xs[q][e] = (float)o[e];
                        if (e < 4) s1 += xs[q][e];
                        else s1b += xs[q][e];
                    }
                }
            }
            if (a.gn_part && cgs == 4) {
                // 4-channel groups (a 128-channel tensor of a 32-group norm): the two halves of a 16-byte vector are two groups
                const float meanA = as_sum_rows(s1) * (1.f / 128.f), meanB = as_sum_rows(s1b) * (1.f / 128.f);
                float qa = 0.f, qb = 0.f;
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float da = xs[q][e] - meanA, db = xs[q][4 + e] - meanB;
                        qa = fmaf(da, da, qa);
                        qb = fmaf(db, db, qb);
                    }
                const float m2A = as_sum_rows(qa), m2B = as_sum_rows(qb);
                if (lane < 8) {
                    const int pw = p0 + 32 * sw, n = pw / HW, tile_s = (pw - n * HW) >> 5;
                    const int G = a.Cout / 4, g_first = (u * 64 + lane * 8) / 4;
                    float* o = a.gn_part + (((int64_t)n * a.gn_tiles + tile_s) * G + g_first) * 2;
                    *reinterpret_cast<f32x4*>(o) = f32x4{meanA, m2A, meanB, m2B};
                }
            } else if (a.gn_part) {
                // per (32 pixels, group of cgs = 8 nv couts): the mean by one reduction over the wave's rows (+ the group's slots), then
                // M2 = sum (x - mean)^2 by a second one
                s1 += s1b;
                const float mean = as_sum_slots(as_sum_rows(s1), nv) * (1.f / (float)(32 * cgs));
                float q2 = 0.f;
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float d = xs[q][e] - mean;
                        q2 = fmaf(d, d, q2);
                    }
                const float m2 = as_sum_slots(as_sum_rows(q2), nv);
                if (lane < 8 && (lane & (nv - 1)) == 0) {
                    const int pw = p0 + 32 * sw, n = pw / HW, tile_s = (pw - n * HW) >> 5;
                    const int G = a.Cout / cgs, c_first = u * 64 + lane * 8;
                    float* o = a.gn_part + (((int64_t)n * a.gn_tiles + tile_s) * G + c_first / cgs) * 2;
                    *reinterpret_cast<f32x2*>(o) = f32x2{mean, m2};
                }
            }
            if (u < 3) AS_STAMP(5 + 3 * u);
        }
        AS_STAMP(13);
        return;
    }

    // =================================== the MFMA team ===================================
    const int r = lane & 31, h = lane >> 5;
    const int pm = wave >> 1, cn = wave & 1;
    const unsigned ring_base = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(size_t)(lds_c*)lds);
    const unsigned a_base = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(size_t)(lds_c*)ldsA);
    // ---- weight units: lane (row & 7 = lane >> 3, piece = lane & 7) of wave-instruction I fills LDS row 8 (I & 7) + (lane >> 3) of
    // chunk I >> 3 lane-linearly, so the XOR swizzle goes on the SOURCE piece
    unsigned boff[DPU];
#pragma unroll
    for (int i = 0; i < DPU; ++i) {
        const int I = wave + 4 * i, c = I >> 3, row = 8 * (I & 7) + (lane >> 3);
        boff[i] = (unsigned)(row * K + c * 64 + ((lane & 7) ^ ((row >> 1) & 7)) * 8) * 2u;
    }
    auto dma_unit = [&](int u, int slot) __attribute__((always_inline)) {
        const char* sb = (const char*)a.w + (int64_t)u * 64 * K * 2;
#pragma unroll
        for (int i = 0; i < DPU; ++i) glds16_hidden_s(sb, boff[i], ring_base + (unsigned)(slot * U_BYTES + (wave + 4 * i) * 1024));
    };
    // ---- phase 0: the activation tile and unit 0 by DMA
#pragma unroll
    for (int i = 0; i < 4 * KCH; ++i) {
        const int I = wave + 4 * i, c = I >> 4, row = 8 * (I & 15) + (lane >> 3);
        const int c0 = c * 64;
        const bool second = c0 >= a.C1;
        const T* sbase = second ? (const T*)a.src2 : (const T*)a.src1;
        const int Cs = second ? a.C2 : a.C1, cb = second ? c0 - a.C1 : c0;
        glds16_hidden(sbase + (int64_t)(p0 + row) * Cs + cb + ((lane & 7) ^ ((row >> 1) & 7)) * 8, a_base + (unsigned)(I * 1024));
    }
    dma_unit(0, 0);
    AS_STAMP(1);
    wait_vm_keep<0>();
    as_barrier();  // (1)
    AS_STAMP(2);

    if (has_pro) {
        as_prologue_tile<K, T>(ldsA, parL, tid, a.pro_silu != 0, a.dmask != nullptr);
        wait_lgkm_all();
        as_barrier();  // (1b)
    }
    uint4 af[2][NKS];
    {
        const lds_c* A3 = (const lds_c*)ldsA;
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
            const int row = pm * 64 + mi * 32 + r;
            const int t = row * ROW_DATA + ((h ^ ((row >> 1) & 7)) << 4);
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks)
                af[mi][ks] = __builtin_bit_cast(uint4, *reinterpret_cast<const lds_u32x4_as*>(A3 + (ks >> 2) * (128 * ROW_DATA) + (t ^ ((ks & 3) << 5))));
        }
    }
    AS_STAMP(3);
    wait_vm_keep<0>();
    as_barrier();  // (2) every wave has its fragments: the tile's and the rows' LDS is free
    if (NU > 1) dma_unit(1, 1);
    if (RING > 2 && NU > 2) dma_unit(2, 2);

    // ---- unit loop.  The MFMAs run transposed - weights as the A operand, activations as B: D[cout][pixel], lane = pixel r, register
    // j = cout (j & 3) + 8 (j >> 2) + 4 h - so that one half-wave swap per register pair (v_permlane32_swap: lanes 32-63 of one
    // register <-> lanes 0-31 of the other) leaves every lane with 8 consecutive couts of ONE pixel: a 16-byte vector of the staged unit.
    const int tB = (cn * 32 + r) * ROW_DATA + ((h ^ (((cn * 32 + r) >> 1) & 7)) << 4);
    uint4 bfr[2][4];
#define AS_READ(SET, SLOT, KG)                                                                                                     \
    do {                                                                                                                           \
        const lds_c* rb_ = (const lds_c*)(size_t)(ring_base + (unsigned)((SLOT) * U_BYTES + (KG) * (64 * ROW_DATA)));              \
        _Pragma("unroll") for (int kk = 0; kk < 4; ++kk)                                                                           \
            bfr[SET][kk] = __builtin_bit_cast(uint4, *reinterpret_cast<const lds_u32x4_as*>(rb_ + (tB ^ (kk << 5))));               \
    } while (0)
    const lds_c* F3 = (const lds_c*)foldL;
    f32x16 fold;  // bias of the lane's 16 cout rows 4 h + 8 g + (0..3), register order: the first MFMA's C operand (no copies)
#define AS_FOLD(U)                                                                                                                 \
    _Pragma("unroll") for (int g = 0; g < 4; ++g) {                                                                                \
        const f32x4 t_ = *reinterpret_cast<const lds_f32x4_as*>(F3 + ((U) * 64 + cn * 32 + 8 * g + 4 * h) * 4);                    \
        fold[4 * g] = t_[0];                                                                                                       \
        fold[4 * g + 1] = t_[1];                                                                                                   \
        fold[4 * g + 2] = t_[2];                                                                                                   \
        fold[4 * g + 3] = t_[3];                                                                                                   \
    }
    AS_READ(0, 0, 0);
    AS_FOLD(0);
    int slot = 0;
    const int px = pm * 64 + lane;  // the pixel this lane owns after the swap
    char* stw = stageL + px * (RES ? 256 : 128);
    int stoff[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) stoff[g] = ((cn * 4 + g) ^ (px & 7)) << (RES ? 5 : 4);
#pragma unroll 1
    for (int u = 0; u < NU; ++u) {
        const int nslot = slot == RING - 1 ? 0 : slot + 1;
        f32x16 acc[2];
#pragma unroll
        for (int kg = 0; kg < KCH; ++kg) {  // 4 k-steps per group, the next group's fragments read under this group's MFMAs
            if (kg + 1 < KCH) {
                if (kg & 1)
                    AS_READ(0, slot, kg + 1);
                else
                    AS_READ(1, slot, kg + 1);
            }
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                for (int mi = 0; mi < 2; ++mi)
                {
                    if (kg == 0 && kk == 0) acc[mi] = fold;
                    mma16<T>(__builtin_bit_cast(uint4, bfr[kg & 1][kk]), __builtin_bit_cast(uint4, af[mi][4 * kg + kk]), acc[mi]);
                }
        }
        if (u < 3) AS_STAMP(4 + 3 * u);
        // ---- this wave's 64 px x 32 couts of unit u -> bf16, staged for the store team ----
        char* st = stw + (u & 1) * STAGE;
        as_mfma_drain();
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            float v[8];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float x = acc[0][4 * g + i], y = acc[1][4 * g + i];
                as_swap32(x, y);
                v[i] = x;      // lanes 0-31: pixel r of tile 0, cout 8 g + i; lanes 32-63: pixel r of tile 1
                v[4 + i] = y;  // ... cout 8 g + 4 + i
            }
            if constexpr (RES) {
                *reinterpret_cast<f32x4*>(st + stoff[g]) = f32x4{v[0], v[1], v[2], v[3]};
                *reinterpret_cast<f32x4*>(st + stoff[g] + 16) = f32x4{v[4], v[5], v[6], v[7]};
            } else {
                bf16x8 o;
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = (T)v[e];
                *reinterpret_cast<bf16x8*>(st + stoff[g]) = o;
            }
        }
        // unit u + 1 has landed in this wave's share (only the weights of units u + 2 .. u + RING - 1 are younger in its queue); the
        // staged unit is written
        if (RING > 2 && u + RING - 1 < NU)
            wait_vm_keep<(RING - 2) * DPU>();
        else
            wait_vm_keep<0>();
        as_barrier();  // unit u is staged, unit u + 1 landed everywhere, every wave is done with unit u's slot
        if (u < 3) AS_STAMP(5 + 3 * u);
        if (u + 1 < NU) {
            AS_READ(0, nslot, 0);
            AS_FOLD(u + 1);
            if (u + RING < NU) dma_unit(u + RING, slot);
        }
        slot = nslot;
    }
    AS_STAMP(13);
#undef AS_READ
#undef AS_FOLD
#undef AS_STAMP
}

static bool as_stats_cg_ok(int cg) { return cg == 4 || cg == 8 || cg == 16 || cg == 32; }

// shape rules (statistics aside)
static bool as_shape_ok(int dtype, const ConvArgs& a) {
    const bool off = getenv("DMME_NO_CONV1X1_AS") != nullptr;
    constexpr int min_units = 2;
    if (off || !is16(dtype) || a.x3 || a.mix) return false;
    if (a.taps != 1 || a.stride != 1 || a.up || a.in_nchw || a.out_nchw || a.out_silu || a.res2 || a.n_gno) return false;
    if (a.tproj && a.nt != 1) return false;
    const int K = a.C1 + a.C2;
    if ((K != 128 && K != 256) || a.C1 % 64 || a.Cout % 64 || a.Cout / 64 < min_units || a.Cout > 2048) return false;
    if (a.res1 && a.R1 != a.Cout) return false;
    constexpr int min_wgs = 128, max_wgs = 1 << 30;
    const int64_t M = (int64_t)a.N * a.Hout * a.Wout;
    if (M % 128 || M * a.Cout >= (1ll << 30)) return false;
    if (M / 128 < min_wgs || M / 128 > max_wgs) return false;  // one workgroup per 128 pixels and no split over couts: small maps keep the tiled kernel
    if ((a.scale || a.dmask || a.pro_silu) && (a.Hout * a.Wout) % 128) return false;  // the prologue rows of ONE image per tile
    return true;
}

bool conv1x1_as_supported(int dtype, const ConvArgs& a) {
    if (!as_shape_ok(dtype, a)) return false;
    if (a.gn_part && (!as_stats_cg_ok(a.gn_cg) || (a.Hout * a.Wout) % 32)) return false;
    return true;
}

// fused statistics of this kernel: one partial per 32 pixels
bool conv1x1_as_stats_query(int dtype, const ConvArgs& a, int cg, int* tiles, int* px) {
    if (!as_shape_ok(dtype, a) || !as_stats_cg_ok(cg) || (a.Hout * a.Wout) % 32) return false;
    *tiles = a.Hout * a.Wout / 32;
    *px = 32;
    return true;
}

template <int KCH, bool RES, typename T>
static int launch_as_inst_t(const ConvArgs& a, size_t lds, hipStream_t s) {
    static bool attr_done = false;
    if (!attr_done) {
        DMME_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(conv1x1_as_kernel<KCH, RES, T>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_done = true;
    }
    const int64_t M = (int64_t)a.N * a.Hout * a.Wout;
    hipLaunchKernelGGL((conv1x1_as_kernel<KCH, RES, T>), dim3((unsigned)(M / 128)), dim3(512), lds, s, a, a.Hout * a.Wout, a.Cout / 64);
    DMME_CHECK_LAUNCH();
    return DMME_OK;
}
template <int KCH, bool RES>
static int launch_as_inst(const ConvArgs& a, size_t lds, hipStream_t s) {
    return a.f16 ? launch_as_inst_t<KCH, RES, f16>(a, lds, s) : launch_as_inst_t<KCH, RES, bf16>(a, lds, s);
}

int launch_conv1x1_as(const ConvArgs& a, hipStream_t s) {
    DMME_REQUIRE(conv1x1_as_supported(a.f16 ? DMME_F16 : DMME_BF16, a), DMME_ERR_UNSUPPORTED, "conv1x1_as: unsupported shape");
    const int KCH = (a.C1 + a.C2) / 64;
    const bool res = a.res1 != nullptr;
    const size_t lds = (size_t)as_ring(res) * 64 * KCH * ROW_DATA + as_fold_bytes(a.Cout) + 2 * as_stage_bytes(res);
    DMME_REQUIRE(lds <= 160 * 1024, DMME_ERR_UNSUPPORTED, "conv1x1_as: LDS");
    if (KCH == 4) return res ? launch_as_inst<4, true>(a, lds, s) : launch_as_inst<4, false>(a, lds, s);
    return res ? launch_as_inst<2, true>(a, lds, s) : launch_as_inst<2, false>(a, lds, s);
}

}  // namespace dmme
