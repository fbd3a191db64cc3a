// Pieces shared by the implicit-GEMM convolution kernels (conv_mfma.hip, conv_pipe.hip).
#pragma once
#include "common.h"

namespace dmme {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int ROW_DATA = 128;          // data bytes per LDS row (one Cin chunk of one pixel / cout)
constexpr int ROW_PITCH = ROW_DATA + 16;  // padded pitch

struct ConvTile {  // host-computed geometry, passed by value
    int TW, TH, TN;       // output tile: TN images x TH x TW pixels (product = BM)
    int HH, HWd;          // halo extent in (virtual) input space
    int tiles_x, tiles_y; // tiles per image
    int tiles_m, tiles_n;
    int a_rows;           // TN*HH*HWd
};

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

__device__ __forceinline__ void mma_group(const uint4& a, const uint4& b, f32x16& acc, float*) {
    const f32x4 av = __builtin_bit_cast(f32x4, a), bv = __builtin_bit_cast(f32x4, b);
#pragma unroll
    for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[j], bv[j], acc, 0, 0, 0);
}
__device__ __forceinline__ void mma_group(const uint4& a, const uint4& b, f32x16& acc, bf16*) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc, 0, 0, 0);
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
template <>
__device__ __forceinline__ uint4 prologue_vec<bf16>(uint4 raw, const float* sc, const float* sh, const float* dm, int pro_silu) {
    if (!sc && !pro_silu && !dm) return raw;
    bf16x8 x = __builtin_bit_cast(bf16x8, raw);
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = (float)x[j];
    if (sc) {
        const f32x4 a0 = *reinterpret_cast<const f32x4*>(sc), a1 = *reinterpret_cast<const f32x4*>(sc + 4);
        const f32x4 b0 = *reinterpret_cast<const f32x4*>(sh), b1 = *reinterpret_cast<const f32x4*>(sh + 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            v[j] = fmaf(v[j], a0[j], b0[j]);
            v[4 + j] = fmaf(v[4 + j], a1[j], b1[j]);
        }
    }
    if (pro_silu) {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = silu_fast(v[j]);
    }
    if (dm) {
        const f32x4 m0 = *reinterpret_cast<const f32x4*>(dm), m1 = *reinterpret_cast<const f32x4*>(dm + 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            v[j] *= m0[j];
            v[4 + j] *= m1[j];
        }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) x[j] = (bf16)v[j];
    return __builtin_bit_cast(uint4, x);
}


inline bool make_tile(const ConvArgs& a, int BM, int BN, ConvTile& g) {
    auto pow2 = [](int v) { return v > 0 && (v & (v - 1)) == 0; };
    int TW = a.Wout < 16 ? a.Wout : 16;
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
    return true;
}

inline size_t tile_lds(const ConvTile& g, int BN) { return (size_t)(g.a_rows + BN) * ROW_PITCH; }


}  // namespace dmme
