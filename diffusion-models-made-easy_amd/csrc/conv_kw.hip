// 3x3 stride-1 implicit-GEMM convolution for layers with FEW output pixels (8x8 and 4x4 maps at the benchmark batch, every layer at
// small batches): the K loop split over the waves of a workgroup ("kw").
//
// What the stamps of the four-wave pipelined kernel said about those layers: a 64 x 64 tile with 2 x 2 waves
// reads 2 KB of LDS fragments per MFMA (exactly the LDS read rate at full matrix rate), stages its filter tiles through registers and
// VGPR-sourced LDS stores, and serialises all of it between workgroup barriers - 9 k cycles per 64-channel chunk for 1.15 k cycles of
// matrix work.  Here
//   * every wave computes the WHOLE 64-pixel x (32 NI)-cout tile, over its own contiguous quarter of the (chunk, tap) units: a 64 x 64
//     wave tile needs 1 KB of fragments per MFMA, and the four partial tiles are summed through LDS once, at the end;
//   * a wave owns everything it reads: its units' filter taps arrive by LDS-DMA into a private ring (RING units of 32 NI rows x 128 B),
//     its chunk's halo tile (GroupNorm affine / SiLU / dropout applied on the way) sits in a private LDS region - so the main loop has
//     NO workgroup barrier: a wave waits on its own `s_waitcnt vmcnt(N)` and nothing else;
//   * the DMA is issued behind the compiler's back (glds16_hidden: hipcc would otherwise drain the whole ring before the first LDS
//     read after a DMA); the ring discipline below is what makes that safe.
// One workgroup per CU (4 waves x (halo + ring) <= 160 KB).  Epilogue: the shared LDS-staged one (bias, time row, residual, fused
// GroupNorm partials), or raw split-K partial sums for grids that would leave most CUs idle.
#include <stdio.h>

#include "conv_common.h"

namespace dmme {

// DENSE: the tile is TN WHOLE images (8x8, 4x4, 2x2 maps): the halo tile is just the tile's 64 pixels (one contiguous run of the NHWC
// tensor) plus one row of zeros, and a fragment row whose tap falls outside its image reads that row - 8 KB of halo per wave instead of
// up to 18 (4x4 maps: 144 halo rows for 64 pixels), which is what makes room for a ring deep enough to cover the L2 round trip.
// BM: 64 pixels, or 128 where that still gives every CU a workgroup (8x8 maps at the benchmark batch: two images per tile) - per
// workgroup the fixed ~5 us (arguments, first round trip, cross-wave sum, epilogue) is then paid for twice the matrix work.
constexpr int KW_PAR_BYTES = 512;

template <int NI, int RING, bool DENSE, int BM, typename T = bf16>
__global__ void __launch_bounds__(256, 1) conv3x3_kw_kernel(ConvArgs a, ConvTile g, int shTW, int shTH, int ksplit_dbg) {
    const int ksplit = ksplit_dbg;
    constexpr int KC = 64, EPV = 8, MI = BM / 32, BN = 32 * NI;
    constexpr int U_BYTES = BN * ROW_DATA;  // one unit of filters: BN cout rows of one tap of one 64-channel chunk
    constexpr int NPI = BN / 8;             // DMA wave-instructions per unit (8 rows of 128 B each)
    constexpr int D = RING - 1;             // units in flight ahead of the one being consumed
    constexpr int NQ = MI * NI;             // 32 x 32 sub-tiles of the output tile
    constexpr int OWN = NQ > 4 ? NQ / 4 : 1;  // sub-tiles a wave finishes: q = OWN * wave .. (consecutive couts of one 32-pixel row block)
    static_assert(OWN <= NI && NI % OWN == 0, "a wave's sub-tiles share their pixel rows");
    static_assert(D >= 1 && D * NPI < 64, "vmcnt is a 6-bit counter");
    extern __shared__ __attribute__((aligned(16))) char lds[];
#define KW_STAMP(K) do { if (a.stamps && blockIdx.x < 2 && blockIdx.y == 0 && threadIdx.x == 0) a.stamps[blockIdx.x * 8 + (K)] = (long long)wall_clock64(); } while (0)
    KW_STAMP(0);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int a_bytes = DENSE ? (BM + 8) * ROW_DATA : ((g.a_rows + 7) & ~7) * ROW_DATA;  // whole 1 KB DMA blocks; DENSE: row 64 = zeros
    const int wave_bytes = a_bytes + RING * U_BYTES + KW_PAR_BYTES;
    char* ldsA = lds + wave * wave_bytes;
    char* ldsR = ldsA + a_bytes;
    float* parW = reinterpret_cast<float*>(ldsR + RING * U_BYTES);  // [2][64]: scale / shift of the wave's current chunk (ConvArgs::gni only)

    const int tile_n = blockIdx.x % g.tiles_n, tile_m = blockIdx.x / g.tiles_n;
    const int tx_blk = tile_m % g.tiles_x, ty_blk = (tile_m / g.tiles_x) % g.tiles_y;
    const int n0 = (tile_m / (g.tiles_x * g.tiles_y)) * g.TN;
    const int oy0 = ty_blk << shTH, ox0 = tx_blk << shTW;
    const int co0 = tile_n * BN;
    const int Cin = a.C1 + a.C2;
    const int Hv = a.up ? 2 * a.Hin : a.Hin, Wv = a.up ? 2 * a.Win : a.Win;
    const int mTW = (1 << shTW) - 1, mTH = (1 << shTH) - 1;

    // this workgroup's chunks (split-K over blockIdx.y), this wave's units: unit = chunk * 9 + tap
    const int nchunks_all = Cin / KC;
    const int ch_begin = ksplit > 1 ? (int)blockIdx.y * nchunks_all / ksplit : 0;
    const int ch_end = ksplit > 1 ? ((int)blockIdx.y + 1) * nchunks_all / ksplit : nchunks_all;
    const int U = (ch_end - ch_begin) * 9;
    const int u0 = ch_begin * 9 + U * wave / 4, nu = ch_begin * 9 + U * (wave + 1) / 4 - u0;

    // ---- filter DMA: lane (row & 7 = lane >> 3, piece = lane & 7) of wave-instruction i fills LDS row 8 i + (lane >> 3) lane-linearly,
    // so the XOR swizzle goes on the SOURCE piece; rows past Cout re-read the last filter row (their columns are never stored)
    // One wave per SIMD: every instruction of the loop costs ~5 issue cycles, so the stream is strength-reduced - per-lane byte
    // offsets are fixed, the (tap, chunk) position is a scalar pointer that advances by counters, M0 is one s_add per instruction.
    unsigned boff[NPI];
#pragma unroll
    for (int i = 0; i < NPI; ++i) {
        const int row = 8 * i + (lane >> 3);
        const int co = co0 + row < a.Cout ? co0 + row : a.Cout - 1;
        boff[i] = (unsigned)(co * 9 * Cin + ((lane & 7) ^ ((row >> 1) & 7)) * EPV) * 2u;
    }
    const unsigned ring_base = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(size_t)(lds_c*)ldsR);
    int dtap = u0 % 9;                                                                // position of the next unit to request
    const char* dptr = (const char*)a.w + ((int64_t)dtap * Cin + (u0 / 9) * KC) * 2;  // wave-uniform
    unsigned dslot = ring_base;                                                       // LDS byte address of its ring slot
    auto dma_next = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < NPI; ++i) glds16_hidden_s(dptr, boff[i], dslot + (unsigned)(i * 8 * ROW_DATA));
        dptr += Cin * 2;
        if (++dtap == 9) {
            dtap = 0;
            dptr += (KC - 9 * Cin) * 2;
        }
        dslot = dslot + U_BYTES == ring_base + RING * U_BYTES ? ring_base : dslot + U_BYTES;
    };
    // Ring discipline: unit k of this wave lives in slot k % RING.  Its DMA is issued in step k - RING, after every fragment read of
    // unit k - RING has returned (they fed the MFMAs of that step), and is retired by `vmcnt(D * NPI)` in step k - 1 (vector memory
    // operations retire in order; only the D younger units may remain).
    KW_STAMP(5);
#pragma unroll
    for (int d = 0; d < D; ++d)
        if (d < nu) dma_next();
    KW_STAMP(6);

    // ---- halo tile of one chunk: this wave alone brings it in (once or twice per kernel) ----
    // Phase 1: raw rows by LDS-DMA (clamped addresses, swizzle on the source piece), asynchronous like the filter units and with no
    // register staging.  Phase 2, after the DMA has landed: a ROLLED loop over the wave's vectors zeroes the padding and applies the
    // GroupNorm affine / SiLU / dropout in place.  (The first version loaded 20 vectors per lane through registers, fully unrolled:
    // 30 KB of straight-line code that every launch fetched cold - 6.6 us from kernel entry to the first MFMA on a 4x4 layer.)
    const int halo_px = g.HH * g.HWd;
    const int n_it = (g.a_rows + 7) >> 3;
    const unsigned a_base = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(size_t)(lds_c*)ldsA);
    unsigned okmask = 0;  // bit i: this lane's vector of iteration i is inside an image of the batch
    const int HWo = a.Hout * a.Wout;
    // The norm in front of this conv finished HERE (ConvArgs::gni; one image per tile, host-checked): lane l derives scale / shift of
    // channel c0 + l of the wave's chunk from the producers' partials (gn_in_scale_shift: one batch of loads, in flight together with
    // the chunk's halo DMA) and leaves them in the wave's own LDS rows for the in-place transform.  At small batches every norm of the
    // 32x32 / 16x16 levels was a finalize launch between two latency-bound convs.
    const bool gni_writer = tile_n == 0 && tx_blk == 0 && ty_blk == 0;
    // (tiles of one image keep the chunk's rows in LDS in any case: read per vector from global memory inside the rolled transform loop
    // they were a dependent round trip per iteration - 4.8 us of a 15.6 us batch-1 launch)
    const bool rows_in_lds = a.has_gni || (a.scale && g.TN == 1);
    // (the partials' loads go out BEFORE the halo DMA loop - gni_pre - and are used behind it: their round trip and the loop's ~1 us
    // of address arithmetic overlap instead of following each other)
    // Only the tiles the small batches run: the 128-pixel ones have no registers to hold 32 partials across the loop (they spill).
    constexpr bool PRE = BM <= 64;
    float2 gpre[PRE ? 32 : 1];
    auto gni_pre = [&](int c0) __attribute__((always_inline)) {
        if constexpr (PRE)
            if (rows_in_lds && a.has_gni) gn_in_prefetch(a.gni, n0, c0 + lane, Cin, gpre);
    };
    auto gni_rows = [&](int c0) __attribute__((always_inline)) {
        if (!rows_in_lds) return;
        float sc, sh;
        if (a.has_gni) {
            gn_in_scale_shift(a, n0, c0 + lane, Cin, gni_writer, sc, sh, PRE ? gpre : nullptr);
        } else {
            sc = a.scale[n0 * Cin + c0 + lane];
            sh = a.shift[n0 * Cin + c0 + lane];
        }
        parW[lane] = sc;
        parW[64 + lane] = sh;
    };
    auto gni_vec = [&](uint4 raw, int piece_src, const float* dm) __attribute__((always_inline)) -> uint4 {
        typedef __attribute__((address_space(3))) f32x4 lf4;
        const lds_c* P3 = (const lds_c*)parW;
        typename Vec8<T>::type x = __builtin_bit_cast(typename Vec8<T>::type, raw);
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; e += 4) {
            const f32x4 s4 = *(const lf4*)(P3 + (piece_src * 8 + e) * 4), h4 = *(const lf4*)(P3 + (64 + piece_src * 8 + e) * 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) v[e + j] = fmaf((float)x[e + j], s4[j], h4[j]);
        }
        if (a.pro_silu) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = silu_fast(v[e]);
        }
        if (dm) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] *= dm[e];
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) x[e] = (T)v[e];
        return __builtin_bit_cast(uint4, x);
    };
    auto halo_issue = [&](int c0) __attribute__((always_inline)) {
        const bool second = c0 >= a.C1;
        const T* sbase = second ? (const T*)a.src2 : (const T*)a.src1;
        const int Cs = second ? a.C2 : a.C1;
        const int cb = second ? c0 - a.C1 : c0;
        okmask = 0;
        gni_pre(c0);
        if constexpr (DENSE) {
            const int gp0 = n0 * HWo + (lane >> 3), gp_end = a.N * HWo;  // the tile's pixels are consecutive in the tensor
#pragma unroll
            for (int i = 0; i < BM / 8; ++i) {
                const int row = 8 * i + (lane >> 3), gp = gp0 + 8 * i;
                const bool ok = gp < gp_end;
                okmask |= ok ? 1u << i : 0u;
                glds16_hidden(sbase + (int64_t)(ok ? gp : 0) * Cs + cb + ((lane & 7) ^ ((row >> 1) & 7)) * EPV, a_base + (unsigned)(i * 8 * ROW_DATA));
            }
            gni_rows(c0);
            return;
        }
#pragma unroll 1
        for (int i = 0; i < n_it; ++i) {
            const int row = 8 * i + (lane >> 3);
            const int tn = (int)__umulhi((unsigned)row, g.magic_px), rem = row - tn * halo_px;
            const int hy = (int)__umulhi((unsigned)rem, g.magic_w), hx = rem - hy * g.HWd;
            const int n = n0 + tn, iy = oy0 - 1 + hy, ix = ox0 - 1 + hx;
            const bool ok = row < g.a_rows && n < a.N && iy >= 0 && iy < Hv && ix >= 0 && ix < Wv;
            const int sy = a.up ? (iy >> 1) : iy, sx = a.up ? (ix >> 1) : ix;
            const int pix = ok ? (n * a.Hin + sy) * a.Win + sx : 0;
            okmask |= ok ? 1u << i : 0u;
            glds16_hidden(sbase + (int64_t)pix * Cs + cb + ((lane & 7) ^ ((row >> 1) & 7)) * EPV, a_base + (unsigned)(i * 8 * ROW_DATA));
        }
        gni_rows(c0);
    };
    const bool has_pro = a.scale || a.dmask || a.pro_silu;
    auto halo_finish = [&](int c0) __attribute__((always_inline)) {
        if constexpr (DENSE) {
            if (lane < 8) *reinterpret_cast<uint4*>(ldsA + BM * ROW_DATA + lane * 16) = make_uint4(0u, 0u, 0u, 0u);
            if (!has_pro) return;  // rows past the batch hold pixel 0's values: their output rows are never stored
#pragma unroll 1
            for (int i = 0; i < BM / 8; ++i) {
                const int row = 8 * i + (lane >> 3);
                uint4* p = reinterpret_cast<uint4*>(ldsA + row * ROW_DATA + (lane & 7) * 16);
                if ((okmask >> i) & 1u) {
                    const int n = n0 + ((row >> shTW) >> shTH);
                    const int so = n * Cin + c0 + ((lane & 7) ^ ((row >> 1) & 7)) * EPV;
                    if (rows_in_lds)
                        *p = gni_vec(*p, (lane & 7) ^ ((row >> 1) & 7), a.dmask ? a.dmask + so : nullptr);
                    else
                        *p = prologue_vec<T>(*p, a.scale ? a.scale + so : nullptr, a.scale ? a.shift + so : nullptr, a.dmask ? a.dmask + so : nullptr,
                                             a.pro_silu);
                }
            }
            return;
        }
#pragma unroll 1
        for (int i = 0; i < n_it; ++i) {
            const int row = 8 * i + (lane >> 3);
            uint4* p = reinterpret_cast<uint4*>(ldsA + row * ROW_DATA + (lane & 7) * 16);
            if (!((okmask >> i) & 1u)) {
                *p = make_uint4(0u, 0u, 0u, 0u);
            } else if (has_pro) {
                const int n = n0 + (int)__umulhi((unsigned)row, g.magic_px);
                const int so = n * Cin + c0 + ((lane & 7) ^ ((row >> 1) & 7)) * EPV;  // the source piece this LDS piece holds
                if (rows_in_lds)
                    *p = gni_vec(*p, (lane & 7) ^ ((row >> 1) & 7), a.dmask ? a.dmask + so : nullptr);
                else
                    *p = prologue_vec<T>(*p, a.scale ? a.scale + so : nullptr, a.scale ? a.shift + so : nullptr, a.dmask ? a.dmask + so : nullptr,
                                         a.pro_silu);
            }
        }
    };

    // fragment read bases: every wave reads all 64 pixels and all BN couts
    int a_row[MI];
    unsigned a_valid[MI];  // DENSE: bit t - tap t of this lane's pixel lies inside its image
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
        const int m = mi * 32 + r;
        const int tx = m & mTW, ty = (m >> shTW) & mTH, tn = m >> (shTW + shTH);
        a_row[mi] = DENSE ? m : (tn * g.HH + ty) * g.HWd + tx;
        a_valid[mi] = 0;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int yy = ty + t / 3 - 1, xx = tx + t % 3 - 1;
            a_valid[mi] |= (yy >= 0 && yy <= mTH && xx >= 0 && xx <= mTW) ? 1u << t : 0u;
        }
    }
    int b_base[NI], b_swz[NI];
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
        const int row = ni * 32 + r;
        b_base[ni] = row * ROW_DATA;
        b_swz[ni] = (row >> 1) & 7;
    }
    f32x16 acc[MI][NI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int j = 0; j < 16; ++j) acc[mi][ni][j] = 0.f;

    halo_issue((u0 / 9) * KC);
    KW_STAMP(1);
    // Main loop, software-pipelined over units: the fragments of unit k + 1 are read (12 / 16 ds_read_b128) before the MFMAs of unit k
    // issue, so the LDS latency and the address arithmetic sit under matrix work.  Unit k + 1 must have landed by then: the ring keeps
    // D - 1 units in flight behind it.  (tap, chunk, slot) advance by counters - no division in the loop.
    uint4 af[2][4][MI], bfr[2][4][NI];  // two fragment sets, ping-pong (indices are compile-time constants: registers, not scratch)
    // Fragment addresses: piece index (2 kg + h) ^ swz lives in byte-offset bits 4-6, so with t = row * 128 + ((h ^ swz) << 4) the
    // address of k-group kg is t ^ (kg << 5): one XOR per read.  The filter rows are fixed per lane: their four offsets are constants.
    int tb[4][NI];
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
        const int row = ni * 32 + r;
        const int t = row * ROW_DATA + ((h ^ ((row >> 1) & 7)) << 4);
#pragma unroll
        for (int kg = 0; kg < 4; ++kg) tb[kg][ni] = t ^ (kg << 5);
    }
    typedef unsigned u32x4_kw __attribute__((ext_vector_type(4)));
    typedef __attribute__((address_space(3))) u32x4_kw lds_u32x4;
    const lds_c* ldsA3 = (const lds_c*)ldsA;
#define KW_READ_FRAGS(SET, TAP, SLOT_ADDR)                                                                                         \
    do {                                                                                                                           \
        const int ty3_ = (TAP) >= 6 ? 2 : (TAP) >= 3 ? 1 : 0, tx3_ = (TAP) - 3 * ty3_;                                            \
        const int tap_off_ = DENSE ? ((ty3_ - 1) << shTW) + (tx3_ - 1) : ty3_ * g.HWd + tx3_;                                     \
        int ta_[MI];                                                                                                               \
        _Pragma("unroll") for (int mi = 0; mi < MI; ++mi) {                                                                        \
            int row_ = a_row[mi] + tap_off_;                                                                                       \
            if (DENSE && !((a_valid[mi] >> (TAP)) & 1u)) row_ = BM; /* the row of zeros */                                         \
            ta_[mi] = row_ * ROW_DATA + ((h ^ ((row_ >> 1) & 7)) << 4);                                                            \
        }                                                                                                                          \
        const lds_c* rb_ = (const lds_c*)(size_t)(SLOT_ADDR);                                                                      \
        _Pragma("unroll") for (int kg = 0; kg < 4; ++kg) {                                                                         \
            _Pragma("unroll") for (int mi = 0; mi < MI; ++mi)                                                                      \
                af[SET][kg][mi] = __builtin_bit_cast(uint4, *reinterpret_cast<const lds_u32x4*>(ldsA3 + (ta_[mi] ^ (kg << 5))));        \
            _Pragma("unroll") for (int ni = 0; ni < NI; ++ni)                                                                      \
                bfr[SET][kg][ni] = __builtin_bit_cast(uint4, *reinterpret_cast<const lds_u32x4*>(rb_ + tb[kg][ni]));                   \
        }                                                                                                                          \
    } while (0)
    // unit 0: its halo, then its fragments
    int chunk = u0 / 9, tap = u0 - chunk * 9;
    wait_vm_keep<0>();  // the first halo rows and the D units ahead of them
    KW_STAMP(7);
    halo_finish(chunk * KC);
    KW_STAMP(2);
    if (D < nu) dma_next();
    KW_READ_FRAGS(0, tap, ring_base);
    unsigned rslot = RING > 1 ? ring_base + U_BYTES : ring_base;  // LDS address of unit k + 1's slot
    int k = 0;
    // One step = the MFMAs of unit k (fragment set CUR), with everything that prepares later units issued BETWEEN its k-groups: the
    // request for unit k + RING goes out under k-group 0, the fragments of unit k + 1 (set NXT) are read under k-groups 1-3.
#define KW_MMA(CUR, KG) mma_tile<T, MI, NI>(af[CUR][KG], bfr[CUR][KG], acc)
#define KW_STEP(CUR, NXT)                                                                                                          \
    do {                                                                                                                           \
        const bool more_ = k + 1 < nu;                                                                                             \
        int ntap_ = tap + 1, nchunk_ = chunk;                                                                                      \
        if (ntap_ == 9) {                                                                                                          \
            ntap_ = 0;                                                                                                             \
            ++nchunk_;                                                                                                             \
        }                                                                                                                          \
        const bool sw_ = more_ && nchunk_ != chunk; /* wave-uniform: unit k + 1 starts a new 64-channel chunk */                   \
        const bool pre_ = more_ && !sw_;                                                                                           \
        wait_lgkm_all(); /* set CUR is complete - and unit k's slot is free for unit k + RING */                                   \
        __builtin_amdgcn_sched_barrier(0);                                                                                         \
        KW_MMA(CUR, 0);                                                                                                            \
        __builtin_amdgcn_sched_barrier(0);                                                                                         \
        if (pre_ && k + 1 + D < nu) dma_next();                                                                                    \
        __builtin_amdgcn_sched_barrier(0);                                                                                         \
        KW_MMA(CUR, 1);                                                                                                            \
        __builtin_amdgcn_sched_barrier(0);                                                                                         \
        if (pre_) {                                                                                                                \
            if (k + 1 + D < nu)                                                                                                    \
                wait_vm_keep<D * NPI>(); /* unit k + 1 has landed; units k + 2 .. k + 1 + D may be in flight */                    \
            else                                                                                                                   \
                wait_vm_keep<0>();                                                                                                 \
            KW_READ_FRAGS(NXT, ntap_, rslot);                                                                                      \
        }                                                                                                                          \
        __builtin_amdgcn_sched_barrier(0);                                                                                         \
        KW_MMA(CUR, 2);                                                                                                            \
        KW_MMA(CUR, 3);                                                                                                            \
        __builtin_amdgcn_sched_barrier(0);                                                                                         \
        if (sw_) {                                                                                                                 \
            halo_issue(nchunk_ * KC); /* unit k's fragments are in registers: the halo region is free */                            \
            if (k + 1 + D < nu) dma_next();                                                                                        \
            wait_vm_keep<0>();                                                                                                     \
            halo_finish(nchunk_ * KC);                                                                                             \
            KW_READ_FRAGS(NXT, ntap_, rslot);                                                                                      \
        }                                                                                                                          \
        tap = ntap_;                                                                                                               \
        chunk = nchunk_;                                                                                                           \
        rslot = rslot + U_BYTES == ring_base + RING * U_BYTES ? ring_base : rslot + U_BYTES;                                       \
        ++k;                                                                                                                       \
    } while (0)
#pragma unroll 1
    while (k + 1 < nu) {
        KW_STEP(0, 1);
        KW_STEP(1, 0);
    }
    if (k < nu) KW_STEP(0, 1);
#undef KW_STEP
#undef KW_MMA
#undef KW_READ_FRAGS
    KW_STAMP(3);

    // ---- sum the four partial tiles: wave w ends up with sub-tiles OWN w .. OWN w + OWN - 1 (q = mi * NI + ni), added in wave order ----
    float* red = reinterpret_cast<float*>(lds);  // [wave][q][j][lane]
    __syncthreads();                              // every wave is done with its halo / ring (all DMA retired by its last wait)
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
            const int q = mi * NI + ni;
            if (q / OWN != wave) {
#pragma unroll
                for (int j = 0; j < 16; ++j) red[((wave * NQ + q) * 16 + j) * 64 + lane] = acc[mi][ni][j];
            }
        }
    __syncthreads();
    f32x16 tot[1][OWN];
    const bool owner = wave * OWN < NQ;
    if (owner) {
#pragma unroll
        for (int o = 0; o < OWN; ++o) {
            f32x16 own;
#pragma unroll
            for (int j = 0; j < 16; ++j) own[j] = 0.f;
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int ni = 0; ni < NI; ++ni)
                    if (mi * NI + ni == wave * OWN + o) own = acc[mi][ni];
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                f32x16 p;
                if (w == wave) {
                    p = own;
                } else {
#pragma unroll
                    for (int j = 0; j < 16; ++j) p[j] = red[((w * NQ + wave * OWN + o) * 16 + j) * 64 + lane];
                }
                if (w == 0)
                    tot[0][o] = p;
                else
                    tot[0][o] += p;
            }
        }
    }
    const int wm0 = ((wave * OWN) / NI) * 32, wn0 = ((wave * OWN) % NI) * 32;  // of the owner waves

    auto pix_of = [&](int m) -> int {
        const int tx = m & mTW, ty = (m >> shTW) & mTH, tn = m >> (shTW + shTH);
        const int n = n0 + tn;
        return n < a.N ? (n * a.Hout + oy0 + ty) * a.Wout + ox0 + tx : -1;
    };
    if (ksplit > 1) {  // raw partial sums: lane = cout (coalesced 128-byte rows), register = pixel
        if (owner) {
            float* part = a.splitk + (int64_t)blockIdx.y * a.N * a.Hout * a.Wout * a.Cout;
#pragma unroll
            for (int o = 0; o < OWN; ++o) {
                const int co = co0 + wn0 + 32 * o + r;
                if (co < a.Cout) {
#pragma unroll
                    for (int j = 0; j < 16; ++j) {
                        const int opix = pix_of(wm0 + (j & 3) + 8 * (j >> 2) + 4 * h);
                        if (opix >= 0) part[(int64_t)opix * a.Cout + co] = tot[0][o][j];
                    }
                }
            }
        }
        KW_STAMP(4);
        return;
    }
    const int tile_s = ty_blk * g.tiles_x + tx_blk;
    if (BM >= 64 && a.n_gno > 0) {  // whole-image tile (host-checked): finish the consuming GroupNorms here
        if constexpr (BM >= 64) {
            const bool add_trow = a.tproj && a.nt != 1 && g.TN > 1;
            __syncthreads();
            if (owner) conv_epilogue_stage<T, BN, 1, OWN>(a, tot, co0, wn0, r, h, wm0, n0, reinterpret_cast<float*>(lds), !add_trow);
            __syncthreads();
            conv_epilogue_store_direct<T, BN, BM>(a, co0, n0, a.Hout * a.Wout, pix_of, reinterpret_cast<float*>(lds), add_trow);
        }
    } else if (conv_epilogue_is_staged<T>(a, g.TN)) {
        __syncthreads();  // the partial sums have been read: the staging image may overwrite them
        if (owner) conv_epilogue_stage<T, BN, 1, OWN>(a, tot, co0, wn0, r, h, wm0, n0, reinterpret_cast<float*>(lds));
        __syncthreads();
        // (a 32 x 32 tile is 128 output vectors: the store loop and its statistics run on the first two waves, the others only meet
        // its barriers)
        conv_epilogue_store<T, BM, BN, (BM * BN / 8 < 256 ? BM * BN / 8 : 256)>(a, co0, n0, pix_of, reinterpret_cast<float*>(lds), tile_s);
    } else if (owner) {
        conv_epilogue<T, BM, BN, 1, OWN>(a, tot, co0, wn0, r, h, wm0, n0, g.TN, pix_of, reinterpret_cast<float*>(lds), tile_s);  // general path: no barrier
    }
    KW_STAMP(4);
#undef KW_STAMP
}

// ---- host side ------------------------------------------------------------------------------------------------------------------
static bool kw_dense(const ConvArgs& a, const ConvTile& g, int BM) { return !a.up && g.TH == a.Hout && g.TW == a.Wout && g.TN * g.TH * g.TW == BM; }
static size_t kw_lds(const ConvArgs& a, const ConvTile& g, int BM, int NI, int ring) {
    const size_t halo = kw_dense(a, g, BM) ? (size_t)(BM + 8) * ROW_DATA : (size_t)((g.a_rows + 7) & ~7) * ROW_DATA;
    const size_t per_wave = halo + (size_t)ring * 32 * NI * ROW_DATA + KW_PAR_BYTES;
    const size_t red = (size_t)4 * BM * 32 * NI * 4;
    return 4 * per_wave > red ? 4 * per_wave : red;
}

// which instance (BM, NI, RING) runs this conv; false: not this kernel's shape
bool conv_kw_pick(int dtype, const ConvArgs& a, ConvTile& g, int* ni_out, int* ring_out, int* bm_out) {
    const bool off = getenv("DMME_NO_KW") != nullptr;
    constexpr int force_ni = 0, max_ring = 6;
    const int force_bm = debug_route("kw_bm64") ? 64 : 0;  // (the 8x8 level on 64-pixel tiles instead of two whole images per workgroup)
    if (off || !is16(dtype) || a.x3) return false;
    const int Cin = a.C1 + a.C2;
    if (a.taps != 9 || a.stride != 1 || a.up == 2 || a.in_nchw || Cin % 64 || a.C1 % 64 || a.Cout < 32) return false;
    if ((int64_t)a.Cout * 9 * Cin >= (1ll << 31) || (int64_t)a.N * a.Hin * a.Win * (a.C1 > a.C2 ? a.C1 : a.C2) >= (1ll << 31)) return false;
    // the largest tile that still gives every CU a workgroup (the filter stream per MFMA halves with BM, the fixed cost per workgroup
    // is paid for more work); below that, 64 x 32: the most workgroups
    // Fewer than 256 workgroups even at 64 x 32 (batch 1-2 on the 32x32 / 16x16 maps): 32-pixel tiles - twice the workgroups on a chip
    // that is mostly idle, and a third less halo to normalise per workgroup (the in-place transform is 4.5 us of a 14 us launch
    // there: VALU-bound on one wave per SIMD, stamps in tools/stamp_pipe.py).  DMME_KW_NO_BM32 keeps 64-pixel tiles.
    static const int kCand[5][2] = {{128, 2}, {128, 1}, {64, 2}, {64, 1}, {32, 1}};
    static const int kRings[4] = {6, 4, 3, 2};
    const bool bm32 = !debug_route("kw_no_bm32") && a.Cout % 32 == 0 && !a.n_gno;
    for (int c = 0; c < 5; ++c) {
        const int BM = kCand[c][0], NI = kCand[c][1];
        if ((force_ni && NI != force_ni) || (force_bm && BM != force_bm)) continue;
        if (a.Cout % (32 * NI) && c < 3) continue;  // (partial cout tiles only on the 64 x 32 tile)
        if (c == 4 && !bm32) continue;
        ConvTile t{};
        if (!make_tile(a, BM, 32 * NI, t) || t.a_rows > 256) continue;  // okmask: <= 32 halo vectors per lane
        if (BM == 128 && !kw_dense(a, t, BM)) continue;                  // the 128-pixel form is for whole-image tiles
        // the 32-pixel form: part of one image, and at most 32 statistics tiles per image (a consumer merges <= 64 partials per group,
        // two source groups per group where the norm runs over a concatenation)
        if (BM == 32 && (kw_dense(a, t, BM) || t.TN != 1 || a.Hout * a.Wout / 32 > 32)) continue;
        if (c < 3 && !force_ni && !force_bm && (int64_t)t.tiles_m * t.tiles_n < 256) continue;
        if (c == 3 && bm32 && !force_ni && !force_bm && (int64_t)t.tiles_m * t.tiles_n < 256) {  // would the 32-pixel tile apply?  then it does
            ConvTile t32{};
            if (make_tile(a, 32, 32, t32) && t32.a_rows <= 256 && !kw_dense(a, t32, 32) && t32.TN == 1 && a.Hout * a.Wout / 32 <= 32) continue;
        }
        int ring = 0;
        for (int cand : kRings)
            if (cand <= max_ring && kw_lds(a, t, BM, NI, cand) <= 160 * 1024) {
                ring = cand;
                break;
            }
        if (!ring) continue;
        g = t;
        *ni_out = NI;
        *ring_out = ring;
        *bm_out = BM;
        return true;
    }
    return false;
}

static int ilog2_kw(int v) {
    int s = 0;
    while ((1 << s) < v) ++s;
    return s;
}

template <int NI, int RING, bool DENSE, int BM, typename T>
static int launch_kw_inst_t(const ConvArgs& a, const ConvTile& g, int ksplit, size_t lds, hipStream_t s) {
    static bool attr_done = false;
    if (!attr_done) {
        DMME_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_kw_kernel<NI, RING, DENSE, BM, T>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           160 * 1024));
        attr_done = true;
    }
    const dim3 grid((unsigned)(g.tiles_m * g.tiles_n), (unsigned)ksplit);
    hipLaunchKernelGGL((conv3x3_kw_kernel<NI, RING, DENSE, BM, T>), grid, dim3(256), lds, s, a, g, ilog2_kw(g.TW), ilog2_kw(g.TH), ksplit);
    DMME_CHECK_LAUNCH();
    return DMME_OK;
}
static thread_local int g_kw_dtype = DMME_BF16;  // operand type of the launch being dispatched (launch_conv_kw)
template <int NI, int RING, bool DENSE, int BM>
static int launch_kw_inst(const ConvArgs& a, const ConvTile& g, int ksplit, size_t lds, hipStream_t s) {
    return g_kw_dtype == DMME_F16 ? launch_kw_inst_t<NI, RING, DENSE, BM, f16>(a, g, ksplit, lds, s) : launch_kw_inst_t<NI, RING, DENSE, BM, bf16>(a, g, ksplit, lds, s);
}

int launch_conv_kw(int dtype, const ConvArgs& a, const ConvTile& g, int NI, int ring, int BM, int ksplit, hipStream_t s) {
    g_kw_dtype = dtype;
    const size_t lds = kw_lds(a, g, BM, NI, ring);
    const bool dense = kw_dense(a, g, BM);
#define DMME_KW_CASE(NI_, RING_)                                                                                       \
    if (BM == 64 && NI == NI_ && ring == RING_)                                                                        \
        return dense ? launch_kw_inst<NI_, RING_, true, 64>(a, g, ksplit, lds, s) : launch_kw_inst<NI_, RING_, false, 64>(a, g, ksplit, lds, s);
    DMME_KW_CASE(1, 2)
    DMME_KW_CASE(1, 3)
    DMME_KW_CASE(1, 4)
    DMME_KW_CASE(1, 6)
    DMME_KW_CASE(2, 2)
    DMME_KW_CASE(2, 3)
    DMME_KW_CASE(2, 4)
    DMME_KW_CASE(2, 6)
#undef DMME_KW_CASE
#define DMME_KW_CASE32(RING_) \
    if (BM == 32 && !dense && NI == 1 && ring == RING_) return launch_kw_inst<1, RING_, false, 32>(a, g, ksplit, lds, s);
    DMME_KW_CASE32(2)
    DMME_KW_CASE32(3)
    DMME_KW_CASE32(4)
    DMME_KW_CASE32(6)
#undef DMME_KW_CASE32
#define DMME_KW_CASE128(NI_, RING_) \
    if (BM == 128 && dense && NI == NI_ && ring == RING_) return launch_kw_inst<NI_, RING_, true, 128>(a, g, ksplit, lds, s);
    DMME_KW_CASE128(1, 2)
    DMME_KW_CASE128(1, 3)
    DMME_KW_CASE128(1, 4)
    DMME_KW_CASE128(1, 6)
    DMME_KW_CASE128(2, 2)
    DMME_KW_CASE128(2, 3)
#undef DMME_KW_CASE128
    DMME_REQUIRE(false, DMME_ERR_UNSUPPORTED, "conv_kw: no such instance");
    return DMME_OK;
}

}  // namespace dmme
