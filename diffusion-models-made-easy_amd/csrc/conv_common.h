// Pieces shared by the implicit-GEMM convolution kernels (conv_mfma.hip, conv_pipe.hip).
#pragma once
#include <stdlib.h>

#include "common.h"

namespace dmme {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef f16 f16x8 __attribute__((ext_vector_type(8)));
// the two 16-bit operand types (bf16, and IEEE half: precision="fp16") share every kernel: 8 elements per 16-byte vector, the same LDS
// images, the same MFMA shape (v_mfma_f32_32x32x16_{bf16,f16}: equal rate); only the conversions and the MFMA opcode differ
template <typename T>
struct Vec8 {
    typedef T type __attribute__((ext_vector_type(8)));
};
template <typename T>
__device__ __forceinline__ void unpack8(const uint4& raw, float (&v)[8]) {
    const typename Vec8<T>::type x = __builtin_bit_cast(typename Vec8<T>::type, raw);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = (float)x[e];
}
template <typename T>
__device__ __forceinline__ uint4 pack8(const float (&v)[8]) {
    typename Vec8<T>::type x;
#pragma unroll
    for (int e = 0; e < 8; ++e) x[e] = (T)v[e];
    return __builtin_bit_cast(uint4, x);
}

constexpr int ROW_DATA = 128;          // data bytes per LDS row (one Cin chunk of one pixel / cout)
constexpr int ROW_PITCH = ROW_DATA + 16;  // padded pitch

// LDS rows are 128 B (one 64-channel bf16 chunk); 16-byte piece `chunk` of row `row` lives at piece chunk ^ (row>>1)&7
__device__ __forceinline__ int swz_off(int row, int chunk) { return row * ROW_DATA + ((chunk ^ ((row >> 1) & 7)) << 4); }

// Device-pass-only instructions behind small helpers: the host pass of hipcc parses kernel bodies too, and it knows neither
// the gfx950 LDS-DMA builtin nor these s_waitcnt forms (an error there silently drops the kernel's host stub).
typedef __attribute__((address_space(3))) char lds_c;
__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_byte_addr) {
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_amdgcn_global_load_lds(gsrc, (lds_c*)(size_t)lds_byte_addr, 16, 0, 0);
#else
    (void)gsrc;
    (void)lds_byte_addr;
#endif
}
// The same DMA issued behind the compiler's back, for a wave that READS LDS itself while its DMA into another buffer is in flight: hipcc
// cannot tell the buffers apart and puts `s_waitcnt vmcnt(0)` in front of the first LDS read after a builtin DMA (the whole round trip
// lands before the matrix work instead of under it).  The asm form has no memory clobber; the caller must (i) keep every ordinary access
// to the destination buffer behind a workgroup barrier and (ii) retire the DMA itself with wait_vm_all() before that barrier - the
// compiler's own counted waits do not know these operations and can only over-wait (they retire in order).
__device__ __forceinline__ void glds16_hidden(const void* gsrc, unsigned lds_byte_addr) {
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(gsrc), "s"(lds_byte_addr) : "m0");
#else
    (void)gsrc;
    (void)lds_byte_addr;
#endif
}
// the same with a wave-uniform base pointer and a per-lane byte offset (no 64-bit vector add per instruction); M0 written by the add
__device__ __forceinline__ void glds16_hidden_s(const void* sbase, unsigned lane_off, unsigned lds_byte_addr) {
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(lane_off), "s"(sbase), "s"(lds_byte_addr) : "m0");
#else
    (void)sbase;
    (void)lane_off;
    (void)lds_byte_addr;
#endif
}
__device__ __forceinline__ void wait_vm_all() {
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
}
__device__ __forceinline__ void wait_lgkm_all() {
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
}
// workgroup barrier that orders LDS accesses only.  __syncthreads() is a workgroup-scope release + acquire: behind a store loop hipcc puts
// `s_waitcnt vmcnt(0)` in front of the s_barrier, i.e. every wave waits until its OUTPUT stores are acknowledged (1-2 k cycles under load)
// although nobody in the workgroup reads them.  Epilogues that only reuse an LDS staging area use this one.
__device__ __forceinline__ void lds_barrier() {
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
#endif
}
template <int N>
__device__ __forceinline__ void wait_vm_keep() {  // retire all but the N youngest vector-memory operations; all LDS operations
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
}

template <typename T>
struct Frag;
template <>
struct Frag<float> {
    static constexpr int KC = 32;  // channels per 128-byte chunk
    static constexpr int EPV = 4;  // elements per 16-byte vector
};
template <>
struct Frag<bf16> {
    static constexpr int KC = 64;
    static constexpr int EPV = 8;
};
template <>
struct Frag<f16> {
    static constexpr int KC = 64;
    static constexpr int EPV = 8;
};

__device__ __forceinline__ void mma_group(const uint4& a, const uint4& b, f32x16& acc, float*) {
    const f32x4 av = __builtin_bit_cast(f32x4, a), bv = __builtin_bit_cast(f32x4, b);
#pragma unroll
    for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[j], bv[j], acc, 0, 0, 0);
}
__device__ __forceinline__ void mma_group(const uint4& a, const uint4& b, f32x16& acc, bf16*) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc, 0, 0, 0);
}
__device__ __forceinline__ void mma_group(const uint4& a, const uint4& b, f32x16& acc, f16*) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), acc, 0, 0, 0);
}
// one 32x32x16 MFMA on 16-bit operands of type T
template <typename T>
__device__ __forceinline__ void mma16(const uint4& a, const uint4& b, f32x16& acc) {
    mma_group(a, b, acc, (T*)nullptr);
}
// ... as an expression (acc' = mma16v<T>(a, b, acc)), for kernels written around the builtin's value form
template <typename T, typename A, typename B>
__device__ __forceinline__ f32x16 mma16v(const A& a, const B& b, f32x16 acc) {
    mma_group(__builtin_bit_cast(uint4, a), __builtin_bit_cast(uint4, b), acc, (T*)nullptr);
    return acc;
}

// ---- accurate mode ("bf16x3"): fp32 tensors, every product as three bf16 MFMA passes ------------------------------------------
// x = hi + lo with hi = bf16(x), lo = bf16(x - hi): a*b ~= a_hi*b_hi + a_hi*b_lo + a_lo*b_hi, accumulated in fp32 by the matrix
// cores.  What is dropped (a_lo*b_lo and the rounding of lo) is ~2^-16 of a product, against 2^-9 for single-pass bf16, so a
// 44-convolution network stays within north_star's 1e-3 of the fp32 reference - at three bf16 MFMAs (32x32x8: 3 x 32 cycles per
// 8 channels) instead of four fp32 ones (32x32x2: 4 x 64 cycles).  The operands stay fp32 in HBM and LDS; the split happens
// on the fragment registers (10 VALU operations per 4 values: cvt_pk, shift / mask, packed subtract, cvt_pk).
struct X3 {};
template <typename T, bool ACC3>
struct MmaTag {
    typedef T type;
};
template <>
struct MmaTag<float, true> {
    typedef X3 type;
};
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
// PLAIN fp32 vector instructions, as inline asm so that neither the source nor hipcc's SLP vectoriser pairs them into v_pk_*_f32.
// Beside the MFMAs of the partner wave on the same SIMD a packed fp32 instruction costs its wave 10-15 cycles that overlap nothing
// (it waits for the matrix pipe), a plain one nothing for up to ~2 per MFMA slot and ~5 cycles beyond that (tools/issue_probe.py,
// DESIGN.md section 4, round 5): code that runs NEXT TO another wave's MFMAs (the wave-specialised kernel's producers) uses these;
// code that has the SIMD to itself (store loops between tiles) keeps the packed forms, which halve its instruction count.
__device__ __forceinline__ float fma_plain(float a, float b, float c) {
    float d;
    asm("v_fma_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}
__device__ __forceinline__ float mul_plain(float a, float b) {
    float d;
    asm("v_mul_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
__device__ __forceinline__ float add_plain(float a, float b) {
    float d;
    asm("v_add_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
__device__ __forceinline__ float sub_plain(float a, float b) {
    float d;
    asm("v_sub_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
__device__ __forceinline__ float add1_plain(float a) {
    float d;
    asm("v_add_f32 %0, 1.0, %1" : "=v"(d) : "v"(a));
    return d;
}
// y_k = (x_k S_k + H_k) / (1 + 2^(x_k S2_k + H2_k)), k = 0, 1 - the GroupNorm-affine + SiLU (+ mask) prologue of the two channels of a
// halo dword in twelve plain instructions.  ONE asm block, the two channels interleaved by hand: inline asm is invisible to hipcc's
// hazard recogniser, and on gfx950 a transcendental's result may not be read by the very next non-transcendental vector instruction
// (one wait state) - here the other channel's instruction always sits in between.  Bit-identical to the packed form (fma = fma).
__device__ __forceinline__ void pro_pair_plain(float x0, float x1, const f32x2& S, const f32x2& H, const f32x2& S2, const f32x2& H2, float& y0, float& y1) {
    float e0, e1;
    asm("v_fma_f32 %0, %4, %6, %8\n\t"
        "v_fma_f32 %2, %4, %10, %12\n\t"
        "v_fma_f32 %1, %5, %7, %9\n\t"
        "v_fma_f32 %3, %5, %11, %13\n\t"
        "v_exp_f32 %2, %2\n\t"
        "v_exp_f32 %3, %3\n\t"
        "v_add_f32 %2, 1.0, %2\n\t"
        "v_add_f32 %3, 1.0, %3\n\t"
        "v_rcp_f32 %2, %2\n\t"
        "v_rcp_f32 %3, %3\n\t"
        "v_mul_f32 %0, %0, %2\n\t"
        "v_mul_f32 %1, %1, %3"
        : "=&v"(y0), "=&v"(y1), "=&v"(e0), "=&v"(e1)
        : "v"(x0), "v"(x1), "v"(S[0]), "v"(S[1]), "v"(H[0]), "v"(H[1]), "v"(S2[0]), "v"(S2[1]), "v"(H2[0]), "v"(H2[1]));
}
typedef short s16x4 __attribute__((ext_vector_type(4)));
struct Split4 {
    uint2 hi, lo;
};
__device__ __forceinline__ unsigned pack_bf16(float a, float b) {
    const f32x2 v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}
__device__ __forceinline__ Split4 split4(const uint4& v) {
    const f32x4 x = __builtin_bit_cast(f32x4, v);
    Split4 s;
    const unsigned h01 = pack_bf16(x[0], x[1]), h23 = pack_bf16(x[2], x[3]);
    s.hi = make_uint2(h01, h23);
    s.lo = make_uint2(pack_bf16(x[0] - __uint_as_float(h01 << 16), x[1] - __uint_as_float(h01 & 0xffff0000u)),
                      pack_bf16(x[2] - __uint_as_float(h23 << 16), x[3] - __uint_as_float(h23 & 0xffff0000u)));
    return s;
}
__device__ __forceinline__ void mma_x3(const Split4& a, const Split4& b, f32x16& acc) {
    const s16x4 ah = __builtin_bit_cast(s16x4, a.hi), al = __builtin_bit_cast(s16x4, a.lo);
    const s16x4 bh = __builtin_bit_cast(s16x4, b.hi), bl = __builtin_bit_cast(s16x4, b.lo);
    acc = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(al, bh, acc, 0, 0, 0);  // the small terms first
    acc = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(ah, bl, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(ah, bh, acc, 0, 0, 0);
}
// all MI x NI products of one fragment group (16 bytes per lane of each operand row tile)
template <typename TAG, int MI, int NI>
__device__ __forceinline__ void mma_tile(const uint4 (&af)[MI], const uint4 (&bfr)[NI], f32x16 (&acc)[MI][NI]) {
    if constexpr (__is_same(TAG, X3)) {
        Split4 as[MI], bs[NI];
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) as[mi] = split4(af[mi]);
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) bs[ni] = split4(bfr[ni]);
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) mma_x3(as[mi], bs[ni], acc[mi][ni]);
    } else {
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) mma_group(af[mi], bfr[ni], acc[mi][ni], (TAG*)nullptr);
    }
}

// apply the fused prologue to one 16-byte vector of activations
template <typename T>
__device__ __forceinline__ uint4 prologue_vec(uint4 raw, const float* sc, const float* sh, const float* dm, int pro_silu);

template <>
__device__ __forceinline__ uint4 prologue_vec<float>(uint4 raw, const float* sc, const float* sh, const float* dm, int pro_silu) {
    f32x4 v = __builtin_bit_cast(f32x4, raw);
    if (sc) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(sc), b = *reinterpret_cast<const f32x4*>(sh);
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = fmaf(v[j], a[j], b[j]);
    }
    if (pro_silu) {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = silu_f(v[j]);
    }
    if (dm) {
        const f32x4 m = *reinterpret_cast<const f32x4*>(dm);
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] *= m[j];
    }
    return __builtin_bit_cast(uint4, v);
}
// 16-bit tensors: the eight values as four pairs (the two halves of a dword = adjacent channels), every operation a packed fp32
// instruction (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32: the same IEEE operations as the scalar forms, two per issue slot) - these
// prologues run on SIMDs that also feed the matrix cores, and VALU issue cycles are what they cost
template <typename T>
__device__ __forceinline__ uint4 prologue_vec16(uint4 raw, const float* sc, const float* sh, const float* dm, int pro_silu) {
    if (!sc && !pro_silu && !dm) return raw;
    typename Vec8<T>::type x = __builtin_bit_cast(typename Vec8<T>::type, raw);
    f32x2 v[4];
#pragma unroll
    for (int d = 0; d < 4; ++d) v[d] = f32x2{(float)x[2 * d], (float)x[2 * d + 1]};
    if (sc) {
        const f32x4 a0 = *reinterpret_cast<const f32x4*>(sc), a1 = *reinterpret_cast<const f32x4*>(sc + 4);
        const f32x4 b0 = *reinterpret_cast<const f32x4*>(sh), b1 = *reinterpret_cast<const f32x4*>(sh + 4);
        v[0] = __builtin_elementwise_fma(v[0], f32x2{a0[0], a0[1]}, f32x2{b0[0], b0[1]});
        v[1] = __builtin_elementwise_fma(v[1], f32x2{a0[2], a0[3]}, f32x2{b0[2], b0[3]});
        v[2] = __builtin_elementwise_fma(v[2], f32x2{a1[0], a1[1]}, f32x2{b1[0], b1[1]});
        v[3] = __builtin_elementwise_fma(v[3], f32x2{a1[2], a1[3]}, f32x2{b1[2], b1[3]});
    }
    if (pro_silu) {  // silu_fast, pairwise: x * rcp(1 + exp2(-log2(e) x))
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            f32x2 e = v[d] * f32x2{-1.4426950408889634f, -1.4426950408889634f};
            e = f32x2{__builtin_amdgcn_exp2f(e[0]), __builtin_amdgcn_exp2f(e[1])};
            e = f32x2{1.0f, 1.0f} + e;
            v[d] = v[d] * f32x2{__builtin_amdgcn_rcpf(e[0]), __builtin_amdgcn_rcpf(e[1])};
        }
    }
    if (dm) {
        const f32x4 m0 = *reinterpret_cast<const f32x4*>(dm), m1 = *reinterpret_cast<const f32x4*>(dm + 4);
        v[0] = v[0] * f32x2{m0[0], m0[1]};
        v[1] = v[1] * f32x2{m0[2], m0[3]};
        v[2] = v[2] * f32x2{m1[0], m1[1]};
        v[3] = v[3] * f32x2{m1[2], m1[3]};
    }
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        x[2 * d] = (T)v[d][0];
        x[2 * d + 1] = (T)v[d][1];
    }
    return __builtin_bit_cast(uint4, x);
}
template <>
__device__ __forceinline__ uint4 prologue_vec<bf16>(uint4 raw, const float* sc, const float* sh, const float* dm, int pro_silu) {
    return prologue_vec16<bf16>(raw, sc, sh, dm, pro_silu);
}
template <>
__device__ __forceinline__ uint4 prologue_vec<f16>(uint4 raw, const float* sc, const float* sh, const float* dm, int pro_silu) {
    return prologue_vec16<f16>(raw, sc, sh, dm, pro_silu);
}



// A GroupNorm finished by its CONSUMER (ConvArgs::gni): channel c's scale / shift of image n from the producers' per-tile partials.
// All partials of a consumer group come from one source with equal counts, so the mean is the average of the means and
// M2 = sum M2_i + cnt * sum (mean_i - mean)^2.  Every partial is requested before the first is used - ONE round trip (a
// load-then-add loop is npart dependent ones: that version made the consumer slower than the launch it saved).  `writer` (one
// workgroup per image) also leaves scale / shift / {mean, rstd} in memory for later readers (the backward pass).  A finalize kernel
// this small costs its dispatch, a cold instruction fetch and two dependent round trips: ~6 us between two convolutions.
// NB: partials requested per batch.  32 = all at once (one round trip; 64 registers of the preamble - fine where the kernel's own peak
// is higher anyway); 8 = a rolled loop of batches, twice (sums, then the centred squares out of L1 / L2): the four-wave kernels, whose
// main-loop register allocation the 32-wide form inflated (SGPR spills 28 -> 93 in the 128 x 64 instance).
// The loads of gn_in_scale_shift_g's one-batch form, issued early by a caller that has a round trip's worth of other work to put
// between them and the arithmetic (conv_kw.hip: the halo DMA loop): the <= 32 partials of channel c's group; `pre` goes to
// gn_in_scale_shift_g, which then reads no partial itself (more than 32 partials: it loads them as always, `pre` is ignored).
__device__ __forceinline__ void gn_in_prefetch(const GnIn& G, int n, int c, int C, float2 (&pre)[32]) {
    const int cg = C / G.groups, g = c / cg, c_first = g * cg;
    const bool second = c_first >= G.C1;
    const float* p = second ? G.p2 : G.p1;
    const int tiles = second ? G.t2 : G.t1, cs = second ? G.C2 : G.C1;
    const int fg = cs / G.groups, f0 = (second ? c_first - G.C1 : c_first) / fg, nf = cg / fg;
    const int lnf = nf == 1 ? 0 : nf == 2 ? 1 : 2, npart = tiles << lnf;
    const float* q0 = p + ((int64_t)n * tiles * G.groups + f0) * 2;
#pragma unroll
    for (int k = 0; k < 32; ++k) {
        pre[k] = make_float2(0.f, 0.f);
        if (k < npart && npart <= 32) pre[k] = *reinterpret_cast<const float2*>(q0 + ((k >> lnf) * G.groups + (k & (nf - 1))) * 2);
    }
}
template <int NB = 32>
__device__ __forceinline__ void gn_in_scale_shift_g(const GnIn& G, float* scale_w, float* shift_w, int n, int c, int C, bool writer, float& sc, float& sh,
                                                    const float2* pre = nullptr) {
    const int cg = C / G.groups, g = c / cg, c_first = g * cg;
    const bool second = c_first >= G.C1;
    const float* p = second ? G.p2 : G.p1;
    const int tiles = second ? G.t2 : G.t1, cs = second ? G.C2 : G.C1, cnt = second ? G.cnt2 : G.cnt1;
    const int fg = cs / G.groups, f0 = (second ? c_first - G.C1 : c_first) / fg, nf = cg / fg;  // nf = 1, 2, 4
    const int lnf = nf == 1 ? 0 : nf == 2 ? 1 : 2, npart = tiles << lnf;                        // <= 64 (host-checked)
    const float* q0 = p + ((int64_t)n * tiles * G.groups + f0) * 2;  // partial (t, f) at q0 + (t * groups + f) * 2
    const float gam = G.gamma[c], bet = G.beta[c];
    float sm = 0.f, s2 = 0.f, dd = 0.f, mean = 0.f;
    const float inv = 1.f / (float)npart;
    if (NB >= 32 && npart > 32) {
        // 33 .. 64 partials (32-pixel statistics tiles of a small batch under a concatenated norm): two batches of 32, the second pass
        // re-reads them (L1 / L2) - still no launch
#pragma unroll 1
        for (int base = 0; base < npart; base += 32) {
            float2 v[32];
#pragma unroll
            for (int k = 0; k < 32; ++k) {
                const int kk = base + k;
                v[k] = make_float2(0.f, 0.f);
                if (kk < npart) v[k] = *reinterpret_cast<const float2*>(q0 + ((kk >> lnf) * G.groups + (kk & (nf - 1))) * 2);
            }
#pragma unroll
            for (int k = 0; k < 32; ++k) {
                sm += v[k].x;
                s2 += v[k].y;
            }
        }
        mean = sm * inv;
#pragma unroll 1
        for (int base = 0; base < npart; base += 32) {
            float v[32];
#pragma unroll
            for (int k = 0; k < 32; ++k) {
                const int kk = base + k;
                v[k] = mean;
                if (kk < npart) v[k] = q0[((kk >> lnf) * G.groups + (kk & (nf - 1))) * 2];
            }
#pragma unroll
            for (int k = 0; k < 32; ++k) {
                const float d = v[k] - mean;
                dd = fmaf(d, d, dd);
            }
        }
    } else if constexpr (NB >= 32) {
        float2 v[32];
#pragma unroll
        for (int k = 0; k < 32; ++k) {
            v[k] = make_float2(0.f, 0.f);
            if (pre) v[k] = pre[k];
            else if (k < npart) v[k] = *reinterpret_cast<const float2*>(q0 + ((k >> lnf) * G.groups + (k & (nf - 1))) * 2);
        }
#pragma unroll
        for (int k = 0; k < 32; ++k) {
            sm += v[k].x;
            s2 += v[k].y;
        }
        mean = sm * inv;
#pragma unroll
        for (int k = 0; k < 32; ++k) {
            const float d = k < npart ? v[k].x - mean : 0.f;
            dd = fmaf(d, d, dd);
        }
    } else {
#pragma unroll 1
        for (int base = 0; base < npart; base += NB) {
            float2 v[NB];
#pragma unroll
            for (int k = 0; k < NB; ++k) {
                const int kk = base + k;
                v[k] = make_float2(0.f, 0.f);
                if (kk < npart) v[k] = *reinterpret_cast<const float2*>(q0 + ((kk >> lnf) * G.groups + (kk & (nf - 1))) * 2);
            }
#pragma unroll
            for (int k = 0; k < NB; ++k) {
                sm += v[k].x;
                s2 += v[k].y;
            }
        }
        mean = sm * inv;
#pragma unroll 1
        for (int base = 0; base < npart; base += NB) {
            float v[NB];
#pragma unroll
            for (int k = 0; k < NB; ++k) {
                const int kk = base + k;
                v[k] = mean;
                if (kk < npart) v[k] = q0[((kk >> lnf) * G.groups + (kk & (nf - 1))) * 2];
            }
#pragma unroll
            for (int k = 0; k < NB; ++k) {
                const float d = v[k] - mean;
                dd = fmaf(d, d, dd);
            }
        }
    }
    const float var = (s2 + (float)cnt * dd) * inv / (float)cnt;
    const float rstd = 1.0f / sqrtf(var + G.eps);
    sc = rstd * gam;
    sh = bet - mean * sc;
    if (G.t_scale) {  // same arithmetic as gn_finalize_parts_kernel<.., MOD>
        const int64_t r = (int64_t)(G.nt == 1 ? 0 : n) * G.t_ld + c;
        const float mm = 1.0f + G.t_scale[r];
        sc = sc * mm;
        sh = fmaf(sh, mm, G.t_shift[r]);
    }
    if (writer) {
        scale_w[(int64_t)n * C + c] = sc;
        shift_w[(int64_t)n * C + c] = sh;
        if (G.mean_rstd && c == c_first) {
            G.mean_rstd[((int64_t)n * G.groups + g) * 2] = mean;
            G.mean_rstd[((int64_t)n * G.groups + g) * 2 + 1] = rstd;
        }
    }
}

template <int NB = 32>
__device__ __forceinline__ void gn_in_scale_shift(const ConvArgs& a, int n, int c, int C, bool writer, float& sc, float& sh, const float2* pre = nullptr) {
    gn_in_scale_shift_g<NB>(a.gni, const_cast<float*>(a.scale), const_cast<float*>(a.shift), n, c, C, writer, sc, sh, pre);
}

// prologue_vec<bf16> with the scale / shift rows in LDS (a norm finished by this conv: ConvArgs::gni): typed LDS reads - through
// generic pointers they would be flat loads, which count on vmcnt and would drain the kernels' LDS-DMA queues
template <typename T = bf16>
__device__ __forceinline__ uint4 prologue_vec_ldsrows(uint4 raw, const float* sc_lds, const float* sh_lds, const float* dm, int pro_silu) {
    typedef __attribute__((address_space(3))) f32x4 lf4;
    typedef __attribute__((address_space(3))) char lc;
    const lc* s3 = (const lc*)sc_lds;
    const lc* h3 = (const lc*)sh_lds;
    typename Vec8<T>::type x = __builtin_bit_cast(typename Vec8<T>::type, raw);
    f32x2 v[4];  // (pairs and packed fp32 operations, as prologue_vec16)
#pragma unroll
    for (int e = 0; e < 8; e += 4) {
        const f32x4 s4 = *(const lf4*)(s3 + e * 4), h4 = *(const lf4*)(h3 + e * 4);
        v[e / 2] = __builtin_elementwise_fma(f32x2{(float)x[e], (float)x[e + 1]}, f32x2{s4[0], s4[1]}, f32x2{h4[0], h4[1]});
        v[e / 2 + 1] = __builtin_elementwise_fma(f32x2{(float)x[e + 2], (float)x[e + 3]}, f32x2{s4[2], s4[3]}, f32x2{h4[2], h4[3]});
    }
    if (pro_silu) {
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            f32x2 e2 = v[d] * f32x2{-1.4426950408889634f, -1.4426950408889634f};
            e2 = f32x2{__builtin_amdgcn_exp2f(e2[0]), __builtin_amdgcn_exp2f(e2[1])};
            e2 = f32x2{1.0f, 1.0f} + e2;
            v[d] = v[d] * f32x2{__builtin_amdgcn_rcpf(e2[0]), __builtin_amdgcn_rcpf(e2[1])};
        }
    }
    if (dm) {
        const f32x4 m0 = *reinterpret_cast<const f32x4*>(dm), m1 = *reinterpret_cast<const f32x4*>(dm + 4);
        v[0] = v[0] * f32x2{m0[0], m0[1]};
        v[1] = v[1] * f32x2{m0[2], m0[3]};
        v[2] = v[2] * f32x2{m1[0], m1[1]};
        v[3] = v[3] * f32x2{m1[2], m1[3]};
    }
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        x[2 * d] = (T)v[d][0];
        x[2 * d + 1] = (T)v[d][1];
    }
    return __builtin_bit_cast(uint4, x);
}

// Shared epilogue of the implicit-GEMM kernels.  acc layout (32x32 MFMA tile): lane = cout
// column r, registers j = pixel rows (j&3) + 8*(j>>2) + 4*h -- a lane's values are 2 bytes wide
// and 2*Cout bytes apart in the NHWC output, so storing them directly costs 64 narrow store
// (and residual load) instructions per lane.  The common case (NHWC output in T, time-embedding
// row uniform over the tile, no output SiLU) therefore transposes the tile through LDS:
// accumulators (+ bias + time embedding) go to an fp32 [BM][BN] image, then every thread
// handles whole 16-byte output vectors: one vector residual load, one vector store.
// Everything else takes the general path straight from the registers.
// `stage` must hold BM*BN floats and no wave may still be reading operand tiles from it.
// The staged fast epilogue in two parts, so a wave without accumulators (wave-specialised kernels) can join the store loop:
// conv_epilogue_stage: accumulators (+ bias + time embedding) -> fp32 [BM][BN] image in LDS      (no barrier)
// conv_epilogue_store: after a barrier - every thread handles whole 16-byte output vectors; NT = threads taking part
template <typename T>
__device__ __forceinline__ bool conv_epilogue_is_staged(const ConvArgs& a, int TN) {
    constexpr int VEC = 16 / sizeof(T);
    const bool uniform_t = !a.tproj || a.nt == 1 || TN == 1;
    return !a.out_silu && !a.out_nchw && uniform_t && (a.Cout % VEC) == 0;
}
// MSTRIDE: staged rows between a wave's consecutive 32-row accumulator tiles (32: contiguous rows)
// S16 (conv_epilogue_store; 16-bit tensors, no residual input): the staged image holds the ROUNDED outputs, [BM][BN + 8] T (a 272-byte
// pitch), written by the wave-specialised kernel's consumers from TRANSPOSED accumulators (conv_pipe.hip, E16): half the bytes, one
// rounding as before
constexpr int kStage16Pad = 8;
template <typename T, int BN, int MI, int NI, int MSTRIDE = 32>
__device__ __forceinline__ void conv_epilogue_stage(const ConvArgs& a, f32x16 (&acc)[MI][NI], int co0, int wn0, int r, int h, int wm0, int n0,
                                                    float* stage, bool with_trow = true) {
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
        const int c = wn0 + ni * 32 + r, co = co0 + c;
        float fold = 0.f;
        if (co < a.Cout) {
            fold = a.bias ? a.bias[co] : 0.f;
            if (a.r_bias) fold += a.r_bias[co];  // (the residual segment's 1x1 conv)
            if (a.tproj && with_trow) fold += a.tproj[(a.nt == 1 ? 0 : n0) * a.tproj_ld + co];
        }
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const int m = wm0 + mi * MSTRIDE + (j & 3) + 8 * (j >> 2) + 4 * h;
                stage[m * BN + c] = acc[mi][ni][j] + fold;
            }
    }
}
// residual vectors of this thread's items, loaded BEFORE the staging barrier (whole cout tiles, every pixel valid)
template <typename T, int BM, int BN, int NT, typename PixFn>
__device__ __forceinline__ void conv_epilogue_res_prefetch(const ConvArgs& a, int co0, PixFn pix_of, uint4 (&pre)[BM * (BN / (16 / (int)sizeof(T))) / NT]) {
    constexpr int VEC = 16 / sizeof(T), VPR = BN / VEC, ITEMS = BM * VPR / NT;
    const T* __restrict__ res = (const T*)a.res1;
#pragma unroll
    for (int q = 0; q < ITEMS; ++q) {
        const int it = threadIdx.x + q * NT, m = it / VPR, cg = it % VPR;
        pre[q] = make_uint4(0u, 0u, 0u, 0u);
        if (res) pre[q] = *reinterpret_cast<const uint4*>(res + pix_of(m) * a.Cout + co0 + cg * VEC);
    }
}
// direct_pass (wave-specialised kernel, whole-image tiles stored in two passes): -1 - statistics leave as partials; 0 - first pass, the
// group threads keep their (mean, M2) in `carry`; 1 - second pass: merged with the carry, and the consuming norms (ConvArgs::gno) get
// their scale / shift / {mean, rstd} rows from here - no finalize launch.
template <typename T, int BM, int BN, int NT, bool S16 = false, typename PixFn>
__device__ __forceinline__ void conv_epilogue_store(const ConvArgs& a, int co0, int n0, PixFn pix_of, float* stage, int tile_s,
                                                    const uint4* pre = nullptr, int direct_pass = -1, float* carry = nullptr) {
    constexpr int VEC = 16 / sizeof(T);
    constexpr int VPR = BN / VEC;  // vectors per pixel row of the tile
    T* __restrict__ dst = (T*)a.dst;
    const T* __restrict__ res = (const T*)a.res1;
    // fused GroupNorm partials: every thread owns ONE 16-byte channel vector (NT % VPR == 0), so it keeps
    // running sum / sum of squares of the values it stores (two halves of the vector separately)
    float s1a = 0.f, s2a = 0.f, s1b = 0.f, s2b = 0.f;
    f32x2 q1a = {0.f, 0.f}, q2a = {0.f, 0.f}, q1b = {0.f, 0.f}, q2b = {0.f, 0.f};  // (bf16: the same sums as pairs of adjacent channels)
    int cnt_items = 0;
    auto item = [&](int it, const uint4* pv) __attribute__((always_inline)) {
        const int m = it / VPR, cg = it % VPR;
        const int co = co0 + cg * VEC;
        const int opix = pix_of(m);  // -1: pixel belongs to an image past the batch
        if (opix < 0 || co >= a.Cout) return;
        const int off = opix * a.Cout + co;
        const float* sp = stage + m * BN + cg * VEC;
        ++cnt_items;
        if constexpr (sizeof(T) == 4) {
            f32x4 v = *reinterpret_cast<const f32x4*>(sp);
            if (res) {
                const f32x4 rv = pv ? __builtin_bit_cast(f32x4, *pv) : *reinterpret_cast<const f32x4*>(res + off);
                v += rv;
            }
            *reinterpret_cast<f32x4*>(dst + off) = v;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                s1a += v[e];
                s2a = fmaf(v[e], v[e], s2a);
            }
        } else {
            // 16-bit tensors: pairs (the two halves of an output dword = adjacent channels) through packed fp32 instructions - residual
            // add, ONE packed conversion per dword (v_cvt_pk_bf16_f32 / v_cvt_pk_f16_f32), and the statistics of the ROUNDED values
            // unpacked from the packed dwords (bf16: two shifts / masks per dword; half: two conversions) instead of eight single
            // conversions there and back: ~50 instead of ~90 instructions per 16-byte vector, on SIMDs that should be starting the next
            // tile's MFMAs
            typedef T tx2 __attribute__((ext_vector_type(2)));
            auto unpack2 = [](unsigned w) __attribute__((always_inline)) -> f32x2 {
                if constexpr (dtype_of<T>::value == DMME_BF16) return f32x2{__uint_as_float(w << 16), __uint_as_float(w & 0xffff0000u)};
                else return __builtin_convertvector(__builtin_bit_cast(tx2, w), f32x2);
            };
            unsigned ow[4];
            if constexpr (S16) {  // the staged values are the outputs
                const uint4 sv = *reinterpret_cast<const uint4*>(reinterpret_cast<const T*>(stage) + m * (BN + kStage16Pad) + cg * VEC);
                ow[0] = sv.x; ow[1] = sv.y; ow[2] = sv.z; ow[3] = sv.w;
            } else {
            const f32x4 v0 = *reinterpret_cast<const f32x4*>(sp), v1 = *reinterpret_cast<const f32x4*>(sp + 4);
            f32x2 vp[4] = {f32x2{v0[0], v0[1]}, f32x2{v0[2], v0[3]}, f32x2{v1[0], v1[1]}, f32x2{v1[2], v1[3]}};
            if (res) {
                const uint4 rw = pv ? *pv : *reinterpret_cast<const uint4*>(res + off);
                const unsigned rd[4] = {rw.x, rw.y, rw.z, rw.w};
#pragma unroll
                for (int d = 0; d < 4; ++d) vp[d] = vp[d] + unpack2(rd[d]);
            }
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                const tx2 pk = {(T)vp[d][0], (T)vp[d][1]};
                ow[d] = __builtin_bit_cast(unsigned, pk);
                // (half: keeps the compiler from re-deriving the rounded values with single conversions from the fp32 inputs)
                if constexpr (dtype_of<T>::value == DMME_F16) asm volatile("" : "+v"(ow[d]));
            }
            }
            *reinterpret_cast<uint4*>(dst + off) = make_uint4(ow[0], ow[1], ow[2], ow[3]);
#pragma unroll
            for (int d = 0; d < 4; ++d) {  // statistics of the values the consumer reads back: pair accumulators, folded after the loop
                const f32x2 x = unpack2(ow[d]);
                if (d < 2) {
                    q1a = q1a + x;
                    q2a = __builtin_elementwise_fma(x, x, q2a);
                } else {
                    q1b = q1b + x;
                    q2b = __builtin_elementwise_fma(x, x, q2b);
                }
            }
        }
    };
    if (pre) {
        constexpr int ITEMS = BM * VPR / NT;
#pragma unroll
        for (int q = 0; q < ITEMS; ++q) item(threadIdx.x + q * NT, pre + q);
    } else {
        for (int it = threadIdx.x; it < BM * VPR; it += NT) item(it, nullptr);
    }
    s1a += q1a[0] + q1a[1];
    s2a += q2a[0] + q2a[1];
    s1b += q1b[0] + q1b[1];
    s2b += q2b[0] + q2b[1];
    if (a.gn_part || direct_pass >= 0) {
        // per-thread (mean, M2) from <= 64 values (negligible cancellation), exchanged through LDS and merged
        // with Chan's formula by one thread per group.  TN == 1: the tile is one image.
        const int cgs = a.gn_cg;
        const bool split = cgs < VEC;            // bf16, 4-channel groups: the two vector halves are two groups
        constexpr int HALF = VEC > 4 ? 4 : VEC;
        float cntA = (float)(cnt_items * HALF), cntB = cntA;
        if (!split && VEC > 4) {                 // whole vector inside one group: merge the halves first
            s1a += s1b;
            s2a += s2b;
            cntA *= 2.f;
        }
        float meanA = s1a / cntA, m2A = s2a - s1a * meanA;
        float meanB = s1b / cntB, m2B = s2b - s1b * meanB;
        // Lanes VPR apart in a wave own the same channel vector (every thread stored the same number of items): merge them
        // pairwise with shuffles - equal counts make Chan's formula mean = (m1 + m2) / 2, M2 = M2a + M2b + (m1 - m2)^2 n / 2 -
        // then one entry per (wave, channel vector) goes through LDS and one thread per group merges those few.
        constexpr int LPW = VPR < 64 ? VPR : 64;  // distinct channel vectors among a wave's lanes
        float cnt = cntA;                         // (cntB == cntA whenever the B half is used)
#pragma unroll
        for (int off = LPW; off < 64; off <<= 1) {
            const float oA = __shfl_xor(meanA, off, 64), oA2 = __shfl_xor(m2A, off, 64);
            const float oB = __shfl_xor(meanB, off, 64), oB2 = __shfl_xor(m2B, off, 64);
            const float dA = oA - meanA, dB = oB - meanB;
            m2A += oA2 + dA * dA * (0.5f * cnt);
            m2B += oB2 + dB * dB * (0.5f * cnt);
            meanA = 0.5f * (meanA + oA);
            meanB = 0.5f * (meanB + oB);
            cnt *= 2.f;
        }
        lds_barrier();  // all reads of the staged tile are done: reuse it for the exchange
        const int wv = threadIdx.x >> 6, ln = threadIdx.x & 63;
        if (ln < LPW) {
            float* q = stage + (wv * LPW + ln) * 4;
            q[0] = meanA;
            q[1] = m2A;
            q[2] = meanB;
            q[3] = m2B;
        }
        lds_barrier();
        const int GT = BN / cgs;  // groups in this cout tile
        if ((int)threadIdx.x < GT) {
            const int g = threadIdx.x;
            // entries of channel vector v: one per wave when VPR <= 64 (wave w, lane v), else the waves w with
            // (w * 64) % VPR == v - v % 64 ... only VPR <= 64 occurs (BN <= 128 couts, 16-byte vectors of >= 4 elements)
            static_assert(VPR <= 64, "conv_epilogue_store: more channel vectors than lanes");
            constexpr int NW = NT / 64;
            int v_first, v_count, sub;
            if (split) {
                v_first = g >> 1; v_count = 1; sub = g & 1;
            } else {
                v_count = cgs / VEC; v_first = g * v_count; sub = 0;
            }
            float na = 0.f, mean = 0.f, M2 = 0.f;
            for (int vv = 0; vv < v_count; ++vv)
                for (int w = 0; w < NW; ++w) {
                    const float* q = stage + (w * LPW + v_first + vv) * 4 + 2 * sub;
                    const float delta = q[0] - mean, tot = na + cnt;
                    const float rt = __builtin_amdgcn_rcpf(tot);
                    mean += delta * (cnt * rt);
                    M2 += q[1] + delta * delta * (na * cnt * rt);
                    na = tot;
                }
            if (direct_pass < 0) {
                const int G = a.Cout / cgs;
                float* o = a.gn_part + (((int64_t)n0 * a.gn_tiles + tile_s) * G + (co0 / cgs + g)) * 2;
                o[0] = mean;
                o[1] = M2;
            } else if (direct_pass == 0) {
                carry[0] = mean;
                carry[1] = M2;
            } else {
                // the image's two halves (equal counts), then for each consuming norm its group = f adjacent groups of this tensor
                // (neighbouring threads of wave 0: equal counts again)
                const float d0 = carry[0] - mean;
                float m2f = M2 + carry[1] + d0 * d0 * (0.5f * na), meanf = 0.5f * (mean + carry[0]), nf = 2.f * na;
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    if (k >= a.n_gno) break;
                    const GnOut& Gk = a.gno[k];
                    const int f = Gk.cg / cgs;
                    float mk = meanf, m2k = m2f, nk = nf;
                    for (int o = 1; o < f; o <<= 1) {
                        const float om = __shfl_xor(mk, o, 64), o2 = __shfl_xor(m2k, o, 64);
                        const float d = om - mk;
                        m2k += o2 + d * d * (0.5f * nk);
                        mk = 0.5f * (mk + om);
                        nk *= 2.f;
                    }
                    const float rstd = 1.0f / sqrtf(m2k / nk + a.gn_eps);
                    const int c_first = Gk.c_off + co0 + g * cgs;
                    if (g % f == 0) {
                        float* o = Gk.mean_rstd + ((int64_t)n0 * (Gk.C / Gk.cg) + c_first / Gk.cg) * 2;
                        o[0] = mk;
                        o[1] = rstd;
                    }
                    for (int j = 0; j < cgs; ++j) {
                        const float sc = rstd * Gk.gamma[c_first + j];
                        Gk.scale[(int64_t)n0 * Gk.C + c_first + j] = sc;
                        Gk.shift[(int64_t)n0 * Gk.C + c_first + j] = Gk.beta[c_first + j] - mk * sc;
                    }
                }
            }
        }
    }
}

// The store loop for 64- / 128-pixel tiles of WHOLE images (TN = BM / HW images of HW = 2^k pixels; 256 threads), finishing the GroupNorms
// that consume this conv's output (ConvArgs::gno): complete per-(image, group) statistics -> scale / shift / {mean, rstd} rows, and the
// consumer's pre-activated input.  Statistics of the stored (rounded) values, merged with Chan's formula in a fixed order:
//   thread: one 16-byte vector per item (VEC values) -> wave: lanes VPR apart hold the same channel vector of consecutive pixels
//   (PB = 64 / VPR pixels, never straddling an image: HW >= PB) -> LDS -> one thread per (image, vector) over the image's blocks ->
//   one thread per (image, group) over the group's vectors.
// add_trow: the time-embedding row differs per image inside the tile (training: nt = N) and was left out of the staged image.
// `stage`: the fp32 [64][BN] image, followed by 16 KB of scratch this function uses (kDirectLds bytes in all).
constexpr int kDirectLds = 64 * 64 * 4 + 16 * 1024;  // (BM = 64; the 128-pixel kw tiles own 130 KB anyway)
template <typename T, int BN, int BM = 64, typename PixFn>
__device__ __forceinline__ void conv_epilogue_store_direct(const ConvArgs& a, int co0, int n0, int HW, PixFn pix_of, float* stage, bool add_trow) {
    constexpr int NT = 256, VEC = 16 / sizeof(T), VPR = BN / VEC, ITEMS = BM * VPR / NT, PB = 64 / VPR, NPB = BM / PB;
    static_assert(ITEMS >= 1 && NPB == 4 * ITEMS, "conv_epilogue_store_direct: tile shape");
    const int tid = threadIdx.x, wave = tid >> 6, vec = tid % VPR;
    const int TN = BM / HW, sh_hw = 31 - __builtin_clz(HW);
    float* blk = stage + BM * BN;          // [NPB][VPR][2]
    float* img = blk + NPB * VPR * 2;      // [TN][VPR][2]   (TN * VPR <= 256)
    float* mr = img + 512;                 // [2][TN * VPR][2] {mean, rstd} per (image, vector), per norm
    float* ssc = mr + 1024;                // [TN * BN] scale, then [TN * BN] shift of norm act_k
    float* ssh = ssc + 1024;
    T* __restrict__ dst = (T*)a.dst;
    const T* __restrict__ res = (const T*)a.res1;
    // gamma / beta of this thread's first (image, channel) go out now: they do not depend on the statistics, and fetched after them
    // they were an exposed round trip at the end of every tile
    float gm0[2] = {0.f, 0.f}, bt0[2] = {0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 2; ++k)
        if (k < a.n_gno) {
            const int cc = a.gno[k].c_off + co0 + tid % BN;
            gm0[k] = a.gno[k].gamma[cc];
            bt0[k] = a.gno[k].beta[cc];
        }
    uint4 kept[ITEMS];  // the stored vectors (for act)
#pragma unroll
    for (int q = 0; q < ITEMS; ++q) {
        const int it = tid + q * NT, m = it / VPR;
        const int co = co0 + vec * VEC;
        const int opix = pix_of(m);  // -1: image past the batch (its statistics are computed and never written)
        const int off = (opix < 0 ? 0 : opix) * a.Cout + co;
        const float* sp = stage + m * BN + vec * VEC;
        float v[VEC];
#pragma unroll
        for (int e = 0; e < VEC; e += 4) {
            const f32x4 t = *reinterpret_cast<const f32x4*>(sp + e);
            v[e] = t[0]; v[e + 1] = t[1]; v[e + 2] = t[2]; v[e + 3] = t[3];
        }
        if (add_trow) {
            const float* tr = a.tproj + (int64_t)(n0 + (m >> sh_hw) < a.N ? n0 + (m >> sh_hw) : 0) * a.tproj_ld + co;
#pragma unroll
            for (int e = 0; e < VEC; ++e) v[e] += tr[e];
        }
        float x[VEC];
        if constexpr (sizeof(T) == 4) {
            if (res) {
                const f32x4 rv = *reinterpret_cast<const f32x4*>(res + off);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] += rv[e];
            }
            const f32x4 o = {v[0], v[1], v[2], v[3]};
            if (opix >= 0) *reinterpret_cast<f32x4*>(dst + off) = o;
            kept[q] = __builtin_bit_cast(uint4, o);
#pragma unroll
            for (int e = 0; e < 4; ++e) x[e] = v[e];
        } else {
            if (res) {
                const typename Vec8<T>::type rv = *reinterpret_cast<const typename Vec8<T>::type*>(res + off);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] += (float)rv[e];
            }
            typename Vec8<T>::type o;
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = (T)v[e];
            if (opix >= 0) *reinterpret_cast<typename Vec8<T>::type*>(dst + off) = o;
            kept[q] = __builtin_bit_cast(uint4, o);
#pragma unroll
            for (int e = 0; e < 8; ++e) x[e] = (float)o[e];
        }
        // this vector: mean, M2 of VEC values; then the PB pixels of this wave with the same channel vector (equal counts)
        float s = 0.f;
#pragma unroll
        for (int e = 0; e < VEC; ++e) s += x[e];
        float mean = s * (1.f / VEC), m2 = 0.f;
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            const float d = x[e] - mean;
            m2 = fmaf(d, d, m2);
        }
        float cnt = (float)VEC;
#pragma unroll
        for (int o = VPR; o < 64; o <<= 1) {
            const float om = __shfl_xor(mean, o, 64), o2 = __shfl_xor(m2, o, 64);
            const float d = om - mean;
            m2 += o2 + d * d * (0.5f * cnt);
            mean = 0.5f * (mean + om);
            cnt *= 2.f;
        }
        if ((tid & 63) < VPR) {
            float* b = blk + ((4 * q + wave) * VPR + vec) * 2;
            b[0] = mean;
            b[1] = m2;
        }
    }
    __syncthreads();
    // (image, vector): the image's HW / PB blocks, in pixel order
    if (tid < TN * VPR) {
        const int tn = tid / VPR, v = tid % VPR, nb = HW / PB;
        const float cb = (float)(PB * VEC);
        float na = 0.f, mean = 0.f, m2 = 0.f;
        for (int b = 0; b < nb; ++b) {
            const float* q = blk + ((tn * nb + b) * VPR + v) * 2;
            const float delta = q[0] - mean, tot = na + cb;
            const float rt = __builtin_amdgcn_rcpf(tot);
            mean += delta * (cb * rt);
            m2 += q[1] + delta * delta * (na * cb * rt);
            na = tot;
        }
        img[tid * 2] = mean;
        img[tid * 2 + 1] = m2;
    }
    __syncthreads();
    // (image, group) of each norm: the group's cg / VEC vectors, in channel order
    const float cv = (float)(HW * VEC);
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        if (k >= a.n_gno) break;
        const GnOut& G = a.gno[k];
        const int f = G.cg / VEC, gpt = VPR / f;  // vectors per group, groups per image in this tile
        if (tid < TN * gpt) {
            const int tn = tid / gpt, gi = tid % gpt;
            float na = 0.f, mean = 0.f, m2 = 0.f;
            for (int j = 0; j < f; ++j) {
                const float* q = img + (tn * VPR + gi * f + j) * 2;
                const float delta = q[0] - mean, tot = na + cv;
                const float rt = __builtin_amdgcn_rcpf(tot);
                mean += delta * (cv * rt);
                m2 += q[1] + delta * delta * (na * cv * rt);
                na = tot;
            }
            const float rstd = 1.0f / sqrtf(m2 / na + a.gn_eps);
            for (int j = 0; j < f; ++j) {
                float* o = mr + (k * 256 + tn * VPR + gi * f + j) * 2;
                o[0] = mean;
                o[1] = rstd;
            }
            if (n0 + tn < a.N && G.mean_rstd) {
                float* o = G.mean_rstd + ((int64_t)(n0 + tn) * (G.C / G.cg) + (G.c_off + co0) / G.cg + gi) * 2;
                o[0] = mean;
                o[1] = rstd;
            }
        }
    }
    __syncthreads();
    // scale / shift rows: one (image, channel) per thread and pass
    for (int idx = tid; idx < TN * BN; idx += NT) {
        const int tn = idx / BN, c = idx % BN;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            if (k >= a.n_gno) break;
            const GnOut& G = a.gno[k];
            const float* q = mr + (k * 256 + tn * VPR + c / VEC) * 2;
            const int cc = G.c_off + co0 + c;
            const float gm = idx == tid ? gm0[k] : G.gamma[cc], bt = idx == tid ? bt0[k] : G.beta[cc];
            const float sc = q[1] * gm, sh = bt - q[0] * sc;
            if (n0 + tn < a.N) {
                G.scale[(int64_t)(n0 + tn) * G.C + cc] = sc;
                G.shift[(int64_t)(n0 + tn) * G.C + cc] = sh;
            }
            if (a.act && k == a.act_k) {
                ssc[idx] = sc;
                ssh[idx] = sh;
            }
        }
    }
    if (!a.act) return;
    __syncthreads();
    const GnOut& GA = a.gno[a.act_k];
    T* __restrict__ act = (T*)a.act;
#pragma unroll
    for (int q = 0; q < ITEMS; ++q) {
        const int it = tid + q * NT, m = it / VPR;
        const int opix = pix_of(m);
        if (opix < 0) continue;
        const int tn = m >> sh_hw;
        const float* dm = a.act_dmask ? a.act_dmask + (int64_t)(n0 + tn) * GA.C + GA.c_off + co0 + vec * VEC : nullptr;
        const uint4 o = prologue_vec<T>(kept[q], ssc + tn * BN + vec * VEC, ssh + tn * BN + vec * VEC, dm, a.act_silu);
        *reinterpret_cast<uint4*>(act + (int64_t)opix * GA.C + GA.c_off + co0 + vec * VEC) = o;
    }
}

// NT: threads of the workgroup that run the store loop (and the barriers)
template <typename T, int BM, int BN, int MI, int NI, int NT = 256, typename PixFn>
__device__ __forceinline__ void conv_epilogue(const ConvArgs& a, f32x16 (&acc)[MI][NI], int co0, int wn0, int r, int h,
                                              int wm0, int n0, int TN, PixFn pix_of, float* stage, int tile_s) {
    if (conv_epilogue_is_staged<T>(a, TN)) {
        conv_epilogue_stage<T, BN, MI, NI>(a, acc, co0, wn0, r, h, wm0, n0, stage);
        __syncthreads();
        conv_epilogue_store<T, BM, BN, NT>(a, co0, n0, pix_of, stage, tile_s);
        return;
    }
    // general path (final NCHW conv, SiLU outputs, per-image time rows inside one tile)
    // (fully unrolled: a runtime index into acc[][] would push the accumulators to scratch)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
        const int co = co0 + wn0 + ni * 32 + r;
        if (co >= a.Cout) continue;
        const float bias = a.bias ? a.bias[co] : 0.f;
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const int m = wm0 + mi * 32 + (j & 3) + 8 * (j >> 2) + 4 * h;
                const int opix = pix_of(m);
                if (opix < 0) continue;
                const int hw = a.Hout * a.Wout;
                const int n = opix / hw;
                float v = acc[mi][ni][j] + bias;
                if (a.tproj) v += a.tproj[(int64_t)(a.nt == 1 ? 0 : n) * a.tproj_ld + co];
                if (a.res1) v += to_f(((const T*)a.res1)[(int64_t)opix * a.Cout + co]);
                if (a.out_silu) v = silu_f(v);
                if (a.out_nchw)
                    ((float*)a.dst)[((int64_t)n * a.Cout + co) * hw + (opix - n * hw)] = v;
                else
                    ((T*)a.dst)[(int64_t)opix * a.Cout + co] = from_f<T>(v);
            }
        }
    }
}

// number of __syncthreads() the epilogue above executes (for the waves of a workgroup that do not take part in it)
template <typename T>
__device__ __forceinline__ int conv_epilogue_syncs(const ConvArgs& a, int TN) {
    constexpr int VEC = 16 / sizeof(T);
    const bool uniform_t = !a.tproj || a.nt == 1 || TN == 1;
    if (!a.out_silu && !a.out_nchw && uniform_t && (a.Cout % VEC) == 0) return a.gn_part ? 3 : 1;
    return 0;
}

// tile-selection threshold: smallest workgroup count a tile shape must still produce (tunable for experiments)
inline int min_wgs() {
    return 512;
}

// fused statistics need: the staged fast epilogue, one image per tile, whole cout tiles, 16-byte group slices
inline bool stats_tile_ok(const ConvArgs& a, const ConvTile& g, int BN, int cg, int vec) {
    return g.TN == 1 && !a.out_silu && !a.out_nchw && a.Cout % BN == 0 && a.Cout % vec == 0 && (cg % vec == 0 || (vec == 8 && cg == 4)) && BN % cg == 0 &&
           (!a.tproj || a.nt == 1 || g.TN == 1);
}

inline bool make_tile(const ConvArgs& a, int BM, int BN, ConvTile& g) {
    auto pow2 = [](int v) { return v > 0 && (v & (v - 1)) == 0; };
    // 32-wide tiles when the image is at least 32 wide: the 32 pixels of an MFMA row tile are then consecutive halo
    // rows, which the XOR swizzle keeps conflict-free (a 16-wide tile wraps to the next pixel row mid-fragment)
    int TW = a.Wout < 16 ? a.Wout : ((a.Wout % 32 == 0 && BM >= 128 && a.stride == 1) ? 32 : 16);
    if (!pow2(TW) || a.Wout % TW) return false;
    if (BM % TW) return false;
    int TH = BM / TW;
    if (TH > a.Hout) TH = a.Hout;
    if (!pow2(TH) || a.Hout % TH) return false;
    if (BM % (TW * TH)) return false;
    g.TW = TW;
    g.TH = TH;
    g.TN = BM / (TW * TH);
    const int k = a.taps == 9 ? 3 : 1;
    g.HH = (TH - 1) * a.stride + k;
    g.HWd = (TW - 1) * a.stride + k;
    g.tiles_x = a.Wout / TW;
    g.tiles_y = a.Hout / TH;
    g.tiles_m = g.tiles_x * g.tiles_y * ((a.N + g.TN - 1) / g.TN);
    g.tiles_n = (a.Cout + BN - 1) / BN;
    g.a_rows = g.TN * g.HH * g.HWd;
    g.magic_px = (unsigned)((0x100000000ull + (unsigned)(g.HH * g.HWd) - 1) / (unsigned)(g.HH * g.HWd));
    g.magic_w = (unsigned)((0x100000000ull + (unsigned)g.HWd - 1) / (unsigned)g.HWd);
    return true;
}

inline size_t tile_lds(const ConvTile& g, int BN) { return (size_t)(g.a_rows + BN) * ROW_PITCH; }


}  // namespace dmme
