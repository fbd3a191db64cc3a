// Software-pipelined 3x3 (stride 1) implicit-GEMM convolution for the matrix cores.
//
// Same math, operand layouts and fused prologue / epilogue as conv_mfma.hip; what differs
// is the schedule, driven by rocprofv3 counters of the first kernel (waves parked 64 % of
// their cycles at barriers, ~25 VALU instructions per MFMA from index arithmetic):
//   * every per-thread staging address (halo pixel -> source pixel, bounds, LDS slot) is
//     computed ONCE before the channel loop; per step only the channel offset changes;
//   * the filter tiles of one kernel ROW (3 taps) are staged together, so a Cin chunk
//     needs 3 barrier intervals instead of 9 and each interval carries 3x the MFMAs;
//   * the global loads of the next group (filters, and the next chunk's halo) are issued
//     into registers BEFORE the MFMAs of the current group and written to LDS after
//     them: HBM/L2 latency hides under the matrix work of the same workgroup;
//   * LDS rows are 128 B with an XOR swizzle (16-byte chunk index ^ (row>>1)&7) instead of
//     padding: ds_read_b128 stays conflict-free for 16 consecutive rows and the
//     128x128 tile needs 70.5 KB, so two workgroups (8 waves) share a CU.
#include <stdio.h>

#include "conv_common.h"

namespace dmme {



// GT: taps staged per barrier interval (3 = one kernel row, 9 = the whole 3x3 filter: fewer, longer intervals for
//     layers whose per-interval matrix work is too short to hide a global-load round trip);
// UA: halo 16-byte units a thread may own (a_rows * 8 <= 256 * UA).
template <typename T, int BM, int BN, int GT, int PIPE_UA, bool ACC3 = false>
__global__ void __launch_bounds__(256, GT == 9 ? 1 : 2) conv3x3_pipe_kernel(ConvArgs a, ConvTile g, int shTW, int shTH, int ksplit) {
    constexpr int KC = Frag<T>::KC, EPV = Frag<T>::EPV;
    constexpr int MI = BM / 64, NI = BN / 64;
    constexpr int UB = BN / 32;  // filter units per thread per tap
    extern __shared__ __attribute__((aligned(16))) char lds[];
    // diagnostic stamps (tools/stamp_pipe.py): wave 0 of workgroups 0 and 1 - entry, first tiles staged, main loop done, exit
#define PIPE_STAMP(K) do { if (a.stamps && blockIdx.x < 2 && blockIdx.y == 0 && threadIdx.x == 0) a.stamps[blockIdx.x * 8 + (K)] = (long long)wall_clock64(); } while (0)
    PIPE_STAMP(0);
    char* ldsA = lds;
    char* ldsB = lds + (size_t)g.a_rows * ROW_DATA;
    // Filter tiles by LDS-DMA (bf16, one kernel row per interval, 64-cout tiles: two filter buffers fit beside the halo): the next
    // interval's taps go global -> LDS into the OTHER buffer with no register or ds_write in between.  Interval stamps of the
    // register path: 1.6 k of a chunk's 11.6 k cycles were VGPR-sourced LDS stores of filter tiles, and one of its two barriers per
    // interval only protected the single buffer.
    constexpr bool DMAB_OK = sizeof(T) == 2 && GT == 3 && BN == 64;
    constexpr int B_BYTES = GT * BN * ROW_DATA;
    const bool dmab = DMAB_OK && a.dma_b;
    int cur = 0;  // filter buffer the matrix work reads (DMA path)
    float* gni_par = reinterpret_cast<float*>(ldsB + (size_t)B_BYTES * (dmab ? 2 : 1));  // [2][Cin] scale / shift rows (ConvArgs::gni only)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm0 = (wave >> 1) * (BM / 2), wn0 = (wave & 1) * (BN / 2);
    const int r = lane & 31, h = lane >> 5;
    const int cu = tid & 7, urow = tid >> 3;  // staging role: 16-byte chunk, row phase (32 rows per pass)

    const int tile_n = blockIdx.x % g.tiles_n, tile_m = blockIdx.x / g.tiles_n;
    const int tx_blk = tile_m % g.tiles_x, ty_blk = (tile_m / g.tiles_x) % g.tiles_y;
    const int n0 = (tile_m / (g.tiles_x * g.tiles_y)) * g.TN;
    const int oy0 = ty_blk << shTH, ox0 = tx_blk << shTW;
    const int co0 = tile_n * BN;
    const int Cin = a.C1 + a.C2;
    const int Hv = a.up ? 2 * a.Hin : a.Hin, Wv = a.up ? 2 * a.Win : a.Win;
    const int mTW = (1 << shTW) - 1, mTH = (1 << shTH) - 1;

    // split-K (ksplit > 1): blockIdx.y owns a contiguous range of the Cin chunks and writes raw fp32 partial sums
    const int nchunks_all = Cin / KC;
    const int ch_begin = ksplit > 1 ? (int)blockIdx.y * nchunks_all / ksplit : 0;
    const int nchunks = ksplit > 1 ? ((int)blockIdx.y + 1) * nchunks_all / ksplit : nchunks_all;

    // the first filter group goes out before anything else: its round trip overlaps the halo descriptor arithmetic below
    // (stamps: 3.1 us from kernel entry to the first load otherwise - instruction fetch of a cold kernel included)
    int b_off[UB];  // element offset of (cout row, tap 0, cin 0) + this thread's 16-byte chunk; -1: past Cout
#pragma unroll
    for (int k = 0; k < UB; ++k) {
        const int co = co0 + urow + 32 * k;
        b_off[k] = co < a.Cout ? co * 9 * Cin + cu * EPV : -1;
    }
    uint4 breg[GT][UB];
    const T* wbase = (const T*)a.w;
    auto load_B = [&](int c0, int grp) {
#pragma unroll
        for (int j = 0; j < GT; ++j)
#pragma unroll
            for (int k = 0; k < UB; ++k) {
                breg[j][k] = make_uint4(0u, 0u, 0u, 0u);
                if (b_off[k] >= 0) breg[j][k] = *reinterpret_cast<const uint4*>(wbase + b_off[k] + (grp * GT + j) * Cin + c0);
            }
    };
    // DMA form: a wave instruction fills 8 consecutive 128-byte rows lane-linearly, so the XOR swizzle goes on the SOURCE chunk; rows
    // past Cout re-read the last filter row (their output columns are never stored)
    int bd_off[UB];
#pragma unroll
    for (int k = 0; k < UB; ++k) {
        const int row = urow + 32 * k;
        const int co = co0 + row < a.Cout ? co0 + row : a.Cout - 1;
        bd_off[k] = co * 9 * Cin + (cu ^ ((row >> 1) & 7)) * EPV;
    }
    auto dma_B = [&](int c0, int grp, int buf) __attribute__((always_inline)) {
        const unsigned lbase = (unsigned)(size_t)(lds_c*)(ldsB + buf * B_BYTES) + (unsigned)(wave * 8 * ROW_DATA);
#pragma unroll
        for (int j = 0; j < GT; ++j)
#pragma unroll
            for (int k = 0; k < UB; ++k) {
                const unsigned l = (unsigned)__builtin_amdgcn_readfirstlane((int)(lbase + (unsigned)((j * BN + 32 * k) * ROW_DATA)));
                glds16_hidden(wbase + bd_off[k] + (grp * GT + j) * Cin + c0, l);
            }
    };
    if (dmab)
        dma_B(ch_begin * KC, 0, 0);
    else
        load_B(ch_begin * KC, 0);

    // ---- chunk-invariant staging descriptors ----
    int a_pix[PIPE_UA];  // source pixel index, -1: zero (padding / past the batch), -2: no such unit
    int a_ss[PIPE_UA];   // n * Cin (row of the scale/shift/mask tables)
    {
        const int halo_px = g.HH * g.HWd;
#pragma unroll
        for (int i = 0; i < PIPE_UA; ++i) {
            const int row = urow + 32 * i;
            a_pix[i] = -2;
            a_ss[i] = 0;
            if (row < g.a_rows) {
                const int tn = (int)__umulhi((unsigned)row, g.magic_px), rem = row - tn * halo_px;
                const int hy = (int)__umulhi((unsigned)rem, g.magic_w), hx = rem - hy * g.HWd;
                const int n = n0 + tn, iy = oy0 * a.stride - 1 + hy, ix = ox0 * a.stride - 1 + hx;
                a_pix[i] = -1;
                if (n < a.N && iy >= 0 && iy < Hv && ix >= 0 && ix < Wv && !(a.up == 2 && ((iy | ix) & 1))) {
                    const int sy = a.up ? (iy >> 1) : iy, sx = a.up ? (ix >> 1) : ix;
                    a_pix[i] = (n * a.Hin + sy) * a.Win + sx;
                    a_ss[i] = n * Cin;
                }
            }
        }
    }
    // fragment read bases
    int a_row[MI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
        const int m = wm0 + mi * 32 + r;
        const int tx = m & mTW, ty = (m >> shTW) & mTH, tn = m >> (shTW + shTH);
        a_row[mi] = (tn * g.HH + ty * a.stride) * g.HWd + tx * a.stride;
    }
    int b_base[NI], b_swz[NI];
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
        const int row = wn0 + ni * 32 + r;
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

    uint4 areg[PIPE_UA];

    auto load_A = [&](int c0) {
        const bool second = c0 >= a.C1;
        const T* sbase = second ? (const T*)a.src2 : (const T*)a.src1;
        const int Cs = second ? a.C2 : a.C1;
        const int cs = (second ? c0 - a.C1 : c0) + cu * EPV;
#pragma unroll
        for (int i = 0; i < PIPE_UA; ++i)
            if (a_pix[i] >= 0) areg[i] = *reinterpret_cast<const uint4*>(sbase + (int64_t)a_pix[i] * Cs + cs);
    };
    auto store_A = [&](int c0) {
        const int cc = c0 + cu * EPV;
#pragma unroll
        for (int i = 0; i < PIPE_UA; ++i) {
            if (a_pix[i] == -2) continue;
            uint4 val = make_uint4(0u, 0u, 0u, 0u);
            if (a_pix[i] >= 0) {
                const int so = a_ss[i] + cc;
                if constexpr (sizeof(T) == 2) {
                    if (a.has_gni)
                        val = prologue_vec_ldsrows<T>(areg[i], gni_par + cc, gni_par + Cin + cc, a.dmask ? a.dmask + so : nullptr, a.pro_silu);
                    else
                        val = prologue_vec<T>(areg[i], a.scale ? a.scale + so : nullptr, a.scale ? a.shift + so : nullptr,
                                              a.dmask ? a.dmask + so : nullptr, a.pro_silu);
                } else {
                    val = prologue_vec<T>(areg[i], a.scale ? a.scale + so : nullptr, a.scale ? a.shift + so : nullptr,
                                          a.dmask ? a.dmask + so : nullptr, a.pro_silu);
                }
            }
            *reinterpret_cast<uint4*>(ldsA + swz_off(urow + 32 * i, cu)) = val;
        }
    };
    auto store_B = [&]() {
#pragma unroll
        for (int j = 0; j < GT; ++j)
#pragma unroll
            for (int k = 0; k < UB; ++k)
                *reinterpret_cast<uint4*>(ldsB + j * BN * ROW_DATA + swz_off(urow + 32 * k, cu)) = breg[j][k];
    };

    // ---- prologue: first halo chunk (the first filter group is already in flight) ----
    load_A(ch_begin * KC);
    // the norm in front of this conv finished HERE (ConvArgs::gni; host-checked: one image per tile, no split-K): every thread derives
    // the scale / shift of a channel of the tile's image from the producers' partials into LDS rows behind the operand buffers
    if constexpr (sizeof(T) == 2) {
        if (a.has_gni) {
            for (int c = tid; c < Cin; c += 256) {
                float sc, sh;
                gn_in_scale_shift<8>(a, n0, c, Cin, tile_n == 0 && tx_blk == 0 && ty_blk == 0, sc, sh);
                gni_par[c] = sc;
                gni_par[Cin + c] = sh;
            }
            __syncthreads();
        }
    }
    PIPE_STAMP(1);
    store_A(ch_begin * KC);
    if (!dmab) store_B();
    if (dmab) wait_vm_all();
    __syncthreads();
    PIPE_STAMP(2);

    constexpr int NG = 9 / GT;  // groups per Cin chunk
#pragma unroll 1
    for (int ch = ch_begin; ch < nchunks; ++ch) {
#pragma unroll 1
        for (int grp = 0; grp < NG; ++grp) {
            // issue the next group's loads before the matrix work
            const bool last_grp = grp == NG - 1;
            const bool more = !(last_grp && ch == nchunks - 1);
            const int nc0 = (last_grp ? ch + 1 : ch) * KC;
            if (more) {
                if (last_grp) load_A(nc0);  // ahead of the DMA: hipcc guards the prefetch registers with a vmcnt(0) the DMA must not sit behind
                if (dmab)
                    dma_B(nc0, last_grp ? 0 : grp + 1, cur ^ 1);
                else
                    load_B(nc0, last_grp ? 0 : grp + 1);
            }
            const char* ldsBc = ldsB + cur * B_BYTES;
            // ---- GT taps x KC of matrix work out of LDS ----
#pragma unroll
            for (int j = 0; j < GT; ++j) {
                const int tap = grp * GT + j;
                const int tap_off = (tap / 3) * g.HWd + (tap % 3);
                int abase[MI], aswz[MI];
#pragma unroll
                for (int mi = 0; mi < MI; ++mi) {
                    const int row = a_row[mi] + tap_off;
                    abase[mi] = row * ROW_DATA;
                    aswz[mi] = (row >> 1) & 7;
                }
#pragma unroll
                for (int kg = 0; kg < 4; ++kg) {
                    const int cidx = kg * 2 + h;
                    uint4 af[MI], bfr[NI];
#pragma unroll
                    for (int mi = 0; mi < MI; ++mi)
                        af[mi] = *reinterpret_cast<const uint4*>(ldsA + abase[mi] + ((cidx ^ aswz[mi]) << 4));
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni)
                        bfr[ni] = *reinterpret_cast<const uint4*>(ldsBc + j * BN * ROW_DATA + b_base[ni] + ((cidx ^ b_swz[ni]) << 4));
                    mma_tile<typename MmaTag<T, ACC3>::type, MI, NI>(af, bfr, acc);
                }
            }
            if (dmab) {
                if (more && last_grp) {
                    __syncthreads();  // every wave is done reading this chunk's halo
                    store_A(nc0);
                }
                wait_vm_all();    // this wave's share of the next taps has landed ...
                __syncthreads();  // ... so has everyone's; and every wave is done with this interval's buffer
                cur ^= 1;
            } else {
                __syncthreads();  // every wave is done reading this group's tiles
                if (more) {
                    store_B();
                    if (last_grp) store_A(nc0);
                }
                __syncthreads();
            }
        }
    }

    // ---- epilogue: + bias + time embedding + residual, store ----
    auto pix_of = [&](int m) -> int {
        const int tx = m & mTW, ty = (m >> shTW) & mTH, tn = m >> (shTW + shTH);
        const int n = n0 + tn;
        return n < a.N ? (n * a.Hout + oy0 + ty) * a.Wout + ox0 + tx : -1;
    };
    PIPE_STAMP(3);
    if (ksplit > 1) {  // raw partial sums: lane = cout (coalesced 128-byte rows), register = pixel
        float* part = a.splitk + (int64_t)blockIdx.y * a.N * a.Hout * a.Wout * a.Cout;
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
            const int co = co0 + wn0 + ni * 32 + r;
            if (co >= a.Cout) continue;
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    const int opix = pix_of(wm0 + mi * 32 + (j & 3) + 8 * (j >> 2) + 4 * h);
                    if (opix >= 0) part[(int64_t)opix * a.Cout + co] = acc[mi][ni][j];
                }
        }
        PIPE_STAMP(4);
        return;
    }
    if constexpr (BM == 64) {
        if (a.n_gno > 0) {  // whole-image tile (host-checked): finish the consuming GroupNorms here
            const bool add_trow = a.tproj && a.nt != 1 && g.TN > 1;
            conv_epilogue_stage<T, BN, MI, NI>(a, acc, co0, wn0, r, h, wm0, n0, reinterpret_cast<float*>(lds), !add_trow);
            __syncthreads();
            conv_epilogue_store_direct<T, BN>(a, co0, n0, a.Hout * a.Wout, pix_of, reinterpret_cast<float*>(lds), add_trow);
            PIPE_STAMP(4);
            return;
        }
    }
    conv_epilogue<T, BM, BN, MI, NI>(a, acc, co0, wn0, r, h, wm0, n0, g.TN, pix_of, reinterpret_cast<float*>(lds), ty_blk * g.tiles_x + tx_blk);
    PIPE_STAMP(4);
#undef PIPE_STAMP
}

// second half of a split-K convolution: sum the partial images, then the usual epilogue (bias, time-embedding row, residual)
template <typename T>
__global__ void __launch_bounds__(256) conv_splitk_finish_kernel(ConvArgs a, int ksplit, int64_t total4) {
    const int C4 = a.Cout / 4, hw = a.Hout * a.Wout;
    const int64_t slice = (int64_t)a.N * hw * a.Cout;
    for (int64_t q = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; q < total4; q += (int64_t)gridDim.x * blockDim.x) {
        const int64_t p = q / C4;
        const int c = (int)(q - p * C4) * 4;
        float4 v = *reinterpret_cast<const float4*>(a.splitk + p * a.Cout + c);
        for (int z = 1; z < ksplit; ++z) {
            const float4 u = *reinterpret_cast<const float4*>(a.splitk + z * slice + p * a.Cout + c);
            v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
        }
        float f[4] = {v.x, v.y, v.z, v.w};
        const int n = (int)(p / hw);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float fold = a.bias ? a.bias[c + j] : 0.f;
            if (a.tproj) fold += a.tproj[(int64_t)(a.nt == 1 ? 0 : n) * a.tproj_ld + c + j];
            f[j] += fold;
            if (a.res1) f[j] += to_f(((const T*)a.res1)[p * a.Cout + c + j]);
            ((T*)a.dst)[p * a.Cout + c + j] = from_f<T>(f[j]);
        }
    }
}
static int pipe_ksplit(const ConvArgs& a, const ConvTile& g, int pick, int KC) {
    if (a.has_gni) return 1;  // (the rows are per workgroup: every K slice would derive them again)
    const bool off = (debug_route("no_splitk") != 0);
    if (off || !a.splitk || pick != 3 || a.stride != 1 || a.gn_part || a.n_gno || a.out_silu || a.out_nchw || a.res2 || a.Cout % 4) return 1;
    const int64_t wgs = (int64_t)g.tiles_m * g.tiles_n;
    if (wgs > 128) return 1;  // a full wave of workgroups already
    // stamps (tools/stamp_pipe.py): these layers stream 92 KB per 64-channel chunk per workgroup through ONE CU's L2 port
    // (~30 GB/s: 4.2 us per chunk against 0.55 us of MFMA) on half of the CUs, and a workgroup costs ~8 us before and after its
    // chunk loop, so the split only pays while every workgroup still gets its own CU: fill the chip once, no further
    int ks = (a.C1 + a.C2) / KC;
    const int room = (int)(256 / wgs);
    if (ks > room) ks = room;
    if (ks > 4) ks = 4;
    const int64_t out = (int64_t)a.N * a.Hout * a.Wout * a.Cout;
    while (ks > 1 && ks * out > a.splitk_cap) --ks;
    return ks < 1 ? 1 : ks;
}

// (Measured and removed: the filter DMA issued TWO stages ahead, which needs the halo loads as inline asm because hipcc waits
// vmcnt(0) at the first use of an ordinary load's result while an LDS-DMA is pending.  Consumer wait on a plain layer 450 -> 150
// cycles per stage, nothing with the GroupNorm/SiLU prologue, 65.4 -> 64.1 us over the network's launches: asm loads bypass the
// compiler's hazard tracking - too sharp a tool for 2 % of one kernel.  DESIGN.md section 4 keeps the numbers.)
#ifndef WS_CDMA
#define WS_CDMA 1
#endif
// WS_HYB=1: the producers request the taps of stages 1-5 (where they have slack), the consumers the rest.  Measured +0.4 % on the
// sampling step (stages 1-5 become producer-bound by ~0.25 k cycles while the consumers' own stage only shrinks from 1.50 to 1.47 k):
// not worth a producer that sits on the critical path again in the heavier prologues (dropout, split passes).  Off.
#ifndef WS_HYB
#define WS_HYB 0
#endif
#ifndef WS_SCHED2
#define WS_SCHED2 1
#endif
// the consumers' tap loop.  9: every per-tap condition of a stage is a compile-time constant (3 - a third of the code - costs 1.1 % of
// the sampling step: the scalar bookkeeping of a stage).  Measured on top of it and NOT kept (same-box A/B, all correct):
//  - the stage barrier rotated in front of the stage's last MFMA group, the next stage's first fragments requested behind it (the eight
//    MFMAs then run under that latency): consumer work per stage 1.46 -> 1.38 k cycles in the stamps, the step 0.7 % SLOWER (254-256
//    registers, spills in the fp32-staged instances: the fragments live across the barrier);
//  - the next tap's pixel fragments requested ~250 cycles before the barrier: +-0.1 %;
//  - the filter's base address in vector registers + v_readfirstlane instead of hipcc's per-stage reload of the argument block
//    (s_load_dwordx16 + lgkmcnt(0)): 0.7 % slower.
#ifndef WS_TP_UNROLL
#define WS_TP_UNROLL 9
#endif
// E16 staging of the wave-specialised kernel: transposed accumulators (lane = pixel, a register quad = four consecutive couts) -> the
// 16-bit [pixel][BN + 8] image; `row` = this lane's pixel row of the wave's first 32-pixel block, at the wave's first cout (+ 4 h);
// the wave's blocks are 64 staged rows apart
template <typename T, int BN, int MI, int NI>
__device__ __forceinline__ void ws_stage16(const f32x16 (&acc)[MI][NI], T* row) {
    typedef T tx2 __attribute__((ext_vector_type(2)));
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const tx2 lo = {(T)acc[mi][ni][4 * q], (T)acc[mi][ni][4 * q + 1]}, hi = {(T)acc[mi][ni][4 * q + 2], (T)acc[mi][ni][4 * q + 3]};
                *reinterpret_cast<uint2*>(row + mi * 64 * (BN + kStage16Pad) + 32 * ni + 8 * q) = make_uint2(__builtin_bit_cast(unsigned, lo), __builtin_bit_cast(unsigned, hi));
            }
}
struct WsTile { int n0, oy0, ox0, co0, ts; };
__device__ __forceinline__ WsTile ws_tile_of(const ConvTile& g, int shTW, int shTH, int kt, int bn) {
    const int t = (int)blockIdx.x + kt * (int)gridDim.x;
    const int tile_n = t % g.tiles_n, tile_m = t / g.tiles_n;
    const int tiles_img = g.tiles_x * g.tiles_y;
    WsTile r;
    const int sp = tile_m % tiles_img;
    r.n0 = tile_m / tiles_img;
    r.oy0 = (sp / g.tiles_x) << shTH;
    r.ox0 = (sp % g.tiles_x) << shTW;
    r.co0 = tile_n * bn;
    r.ts = sp;
    return r;
}

// bytes behind A1 that make R1 | R2 | A1 the 64 KB the epilogue's staging needs (0 for the 256-pixel tile's halo buffers)
__host__ __device__ inline int ws2_stage_pad(int a_bytes) { return a_bytes + 2 * 128 * ROW_DATA >= 64 * 1024 ? 0 : 64 * 1024 - 2 * 128 * ROW_DATA - a_bytes; }

// ---------------------------------------------------------------------------------------------------------------
// Wave-specialised, persistent 256-pixel x 128-cout kernel (bf16, one image per tile, an even number of 64-channel
// chunks).  In-kernel cycle stamps of the kernel above showed a workgroup spending under half of its life in MFMA
// phases: the GroupNorm/SiLU prologue, the filter staging, the first loads (an all-CU HBM burst, ~10k cycles) and the
// barriers all sit BETWEEN the matrix phases of the same four waves, and per MFMA it moves too many bytes: every
// 128x128 tile re-reads all filters (L2 -> CU sustains ~15 B/clk/CU) and VGPR-sourced LDS stores run at ~79 B/clk.
// Here a workgroup is 8 waves, one per CU, looping over its tiles:
//   waves 0-3  consumers: nothing but LDS fragment reads and MFMAs - 128 pixels x 64 couts each (8 MFMAs per 6
//              fragment reads, explicitly double-buffered: one MFMA wave per SIMD must hide the LDS latency itself);
//   waves 4-7  producers: ONE continuous stream over all the workgroup's tiles - the filter tap of the next stage by
//              LDS-DMA (global_load_lds_dwordx4, swizzle on the source address), the next chunk's halo a chunk ahead
//              through registers with the GN-affine / SiLU / dropout math, spread over the chunk's nine stages; the
//              first chunk and tap of the next tile are therefore in LDS before the current tile's epilogue starts.
// One barrier per stage (a stage = one filter tap of one 64-channel chunk).  Scale / shift / mask of a tile's image sit
// in LDS (double-buffered by tile), so the producers' math never waits on a global load behind the prefetch.  Every
// producer load is unconditional (clamped addresses, padding zeroed by a select), so the loads in flight can be counted:
// the stage barrier is a raw s_barrier behind `s_waitcnt vmcnt(N)` that retires the DMA but not the halo prefetch.
// LDS only fits with single-tap stages:
//   [A0 | R0 | R1 | R2 | A1] + parameters:  A = halo of one 64-channel chunk (double-buffered by chunk), R = ring of
//   filter taps (128 couts x 64 channels = 16 KB, filled by LDS-DMA one stage ahead; slot = stage % 3).
// A tile ends on slot R2 and buffer A1 (9 stages per chunk, even chunk count) while the next tile's first tap / chunk
// are already in R0 / A0, so R1|R2|A1 (>= 64 KB) stages the epilogue: two passes of 128 pixels.
// Parameter rows of image n for the wave-specialised kernel when the norm in front of it is finished HERE (ConvArgs::gni,
// gn_in_scale_shift in conv_common.h): one thread per channel.
__device__ __forceinline__ void ws_fill_par_gni(const ConvArgs& a, int n, bool writer, float* par, int Cin, int tid, int nthr) {
    for (int c = tid; c < Cin; c += nthr) {
        float sc, sh;
        const float dmk = a.dmask ? a.dmask[n * Cin + c] : 1.f;
        gn_in_scale_shift(a, n, c, Cin, writer, sc, sh);
        par[c] = sc * dmk;
        par[Cin + c] = sh * dmk;
        par[2 * Cin + c] = a.pro_silu ? -1.4426950408889634f * sc : 0.f;
        par[3 * Cin + c] = a.pro_silu ? -1.4426950408889634f * sh : -126.f;
    }
}

// BM = 128 (PIPE_UA = 7): the same kernel on 128-pixel tiles (4 x 32 or 8 x 16 pixels, halo <= 204 rows = 7 units, consumer waves 64 pixels x 64
// couts, ONE epilogue pass) for the layers whose 256-pixel tiling leaves CUs without a tile: the 32x32 maps at batch 32, the
// 128-cout layers of the 16x16 level at batch 128.  Per MFMA it moves twice the filter bytes and is behind the 256-pixel form
// wherever that fills the chip; against the four-wave kernel that ran these layers it keeps the producer / consumer split.
// SPLIT (precision="fp16r32": ConvArgs::mix): the tensors are fp32 and every product runs as THREE fp16 MFMA passes.  A 128-byte LDS
// row holds 32 input channels as [hi: 32 halves | lo: 32 halves], hi = f16(v), lo = f16(v - hi): the activation rows are split by the
// producers behind the prologue (which computes in fp32 anyway), the filter rows are packed that way (pack_table_kernel, code 4).  LDS
// images, DMA stream, ring and barriers are those of the 64-channel 16-bit chunk; of a stage's four k-groups 0, 1 are the hi halves
// and 2, 3 the lo halves, and the consumers run hi.hi + hi.lo + lo.hi (lo.lo, ~2^-22 of a product, is dropped): 48 MFMAs per stage
// instead of 32 over the same 24 fragment reads.  SPLIT = 2: the source tensor is 16-bit (the up-sampling conv that enters the fp32
// level from the 16-bit level below: lo = 0 exactly), residual and output are still fp32.
template <typename T, int SPLIT>
struct WsSplit {
    typedef T ts;   // residual / output tensor type
};
template <typename T>
struct WsSplit<T, 1> {
    typedef float ts;
};
template <typename T>
struct WsSplit<T, 2> {
    typedef float ts;
};
// RSEG (ConvArgs::r_w): a second K segment behind the nine taps of every tile - the ResBlock's 1x1 residual conv (models/ddpm.py:108-111,131)
// over the block's RAW input (one or two source tensors), accumulated into the same tile: `h + residual(x)` without the residual tensor,
// its launch, its write and its read-back.  Raw input needs no prologue and no halo, so the segment costs the producers no register and
// no VALU instruction: it runs in HALF-stages of 32 input channels whose operands both arrive by LDS-DMA - the tile's BM pixels x 32
// channels (16 / 8 KB, rows of 64 bytes, XOR-swizzled on the source address) and the filter block 128 couts x 32 channels (8 KB) - three
// slots deep at first, five once the main loop is through (a half-stage is ~600 cycles, a request under load ~2.4 k from issue to landed).
// While a tile's last main chunk runs, A0 (free then: the next tile's first chunk normally trickles into it) and R0 (free after tap 6)
// take the first two slots; R2 / R1 / A1 the others once the last tap has been read:
//   pixel slots:  A0[0:16K], A0[16K:32K], R2, A1[0:16K], A1[16K:32K] (half of that on 128-pixel tiles)
//   filter slots: R0[0:8K], R0[8K:16K], R1[0:8K], R1[8K:16K], A1 behind its two pixel slots
// The next tile's first chunk, which the main loop would have stored into A0 during those nine stages, is activated IN its registers
// instead and written in the segment's last stage (a stage that reads neither slot 0 nor 1, or nothing: one empty stage is appended
// where the count does not work out), together with the request for the next tile's tap 0 and second chunk.
// E16 (16-bit tensors, no residual INPUT tensor, 256-pixel tiles): the epilogue's staging phase was 2 x 1.9 k cycles of a ~46 k-cycle
// tile - 64 `ds_write_b32` per consumer wave and pass, the matrix pipe idle - before a store loop that all CUs run at the same moment
// (bound by the rate HBM takes the burst).  Here the consumers' MFMAs run TRANSPOSED (filter rows as the A operand): a lane then holds one
// pixel and, per register quad, four consecutive couts - two packed conversions and ONE 8-byte LDS store per quad, 32 stores per wave
// for the WHOLE tile, the rounded outputs as a [256][136] 16-bit image in R1 | R2 | A1.  Bias + time row (+ the residual segment's
// bias) are the accumulators' INITIAL value (a 128-float row per tile in LDS, written during the previous epilogue), so nothing is
// added at the end; one staging phase and one barrier instead of two, then the two 128-pixel store passes (statistics per pass as
// before).  One rounding, as in the fp32-staged form (the fp32 sum starts from the bias instead of ending with it).
template <int PIPE_UA, typename T = bf16, int BM = 256, int SPLIT = 0, bool RSEG = false, bool E16 = false>
__global__ void __launch_bounds__(512, 1) conv3x3_ws2_kernel(ConvArgs a, ConvTile g, int shTW, int shTH, int ntiles) {
    constexpr int KC = 64, EPV = 8, BN = 128, MI = BM / 64, NI = 2, UB = BN / 32;
    // CDMA (round 5): the filter taps are requested by the CONSUMER waves, behind the barrier that opens a stage, where their matrix
    // pipe waits for the stage's first fragments anyway.  Producer stamps (tools/stamp_ws.py, -DWS_PSTAMPS) had put the four LDS-DMA
    // instructions of a stage at ~1.3 k cycles of the producer wave - an LDS-DMA beside a SATURATED matrix pipe costs its wave ~250
    // cycles (tools/issue_probe.py kind 13), beside one that waits for LDS ~16 - which made the producer the longer half of every stage
    // and, in the stages that carry two halo units, the whole stage 0.6-2 k cycles longer.  -DWS_CDMA=0: the producers request them.
    constexpr bool CDMA = WS_CDMA != 0;
    // SCHED2: the producers' halo-unit schedule without two-unit stages (needs CDMA: no counted vmcnt waits depend on the order of the
    // halo loads any more; the split-pass form keeps the old schedule - its store_A has no separate arithmetic / write modes)
    constexpr bool SCHED2 = WS_SCHED2 != 0 && CDMA && SPLIT == 0;
    static_assert(!(SCHED2 && WS_HYB != 0), "the hybrid tap ownership counts the halo loads of the two-unit schedule");
    static_assert(!E16 || (SPLIT == 0 && BM == 256 && sizeof(T) == 2), "16-bit staging: 16-bit tensors, 256-pixel tiles");
    static_assert(!RSEG || (SPLIT == 0 && sizeof(T) == 2), "residual segment: 16-bit tensors");
    constexpr int KCR = SPLIT ? 32 : KC;  // input channels per chunk (KC = 16-bit k-slots per 128-byte row)
    constexpr int EPR = SPLIT ? 4 : EPV;  // input channels per producer lane and halo unit
    typedef typename WsSplit<T, SPLIT>::ts TS;
    static_assert(SPLIT == 0 || (BM == 256 && dtype_of<T>::value == DMME_F16), "split passes: fp16 hi / lo on the 256-pixel tile");
    constexpr int R_BYTES = BN * ROW_DATA;  // one tap
    constexpr int A_PITCH = ROW_DATA + 16;  // halo rows are padded, not swizzled: fragment reads use immediate offsets from one base
    static_assert((BM == 256 && PIPE_UA == 11) || (BM == 128 && PIPE_UA == 7), "unit schedules below: 11 units over 9 stages, or 7 units over the first 7");
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int a_bytes = g.a_rows * A_PITCH;
    const int Cin = a.C1 + a.C2;
    const int offA1 = a_bytes + 3 * R_BYTES, offR = a_bytes;
    // [2][4][Cin]: per channel of the tile's image, as the producers' packed prologue wants them: S = scale * mask, H = shift * mask,
    // S2 = -log2(e) * scale, H2 = -log2(e) * shift:  act(x) = (x S + H) / (1 + 2^(x S2 + H2))  [= silu(x scale + shift) * mask]
    // (the epilogue stages 128 x 128 floats through R1 | R2 | A1: a 128-pixel tile's halo buffer is padded up to that)
    float* par_base = reinterpret_cast<float*>(lds + 2 * a_bytes + 3 * R_BYTES + ws2_stage_pad(a_bytes));
    // E16: [2][BN] floats behind the parameter rows - bias + time row (+ residual-segment bias) of tile kt's couts in buffer kt & 1
    float* fold_base = par_base + 2 * 4 * Cin;
#define WS_FILL_FOLD(KT)                                                                              \
    if constexpr (E16) {                                                                              \
        if (tid < BN) {                                                                               \
            const TileXY ff_ = WS_TILE(KT);                                                           \
            const int co_ = ff_.co0 + tid;                                                            \
            float f_ = 0.f;                                                                           \
            if (co_ < a.Cout) {                                                                       \
                f_ = a.bias ? a.bias[co_] : 0.f;                                                      \
                if (a.r_bias) f_ += a.r_bias[co_];                                                    \
                if (a.tproj) f_ += a.tproj[(a.nt == 1 ? 0 : ff_.n0) * a.tproj_ld + co_];              \
            }                                                                                         \
            fold_base[((KT) & 1) * BN + tid] = f_;                                                    \
        }                                                                                             \
    }
#define WS_BUFA(p) (lds + (((p) & 1) ? offA1 : 0))
#define WS_RING(slot) (lds + offR + (slot) * R_BYTES)
#define WS_PAR(kt) (par_base + ((kt) & 1) * 4 * Cin)
#define WS_TILE(kt) ws_tile_of(g, shTW, shTH, (kt), BN)
    typedef WsTile TileXY;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool producer = wave >= 4;
    const int nchunks = Cin / KCR;
    const int CinW = SPLIT ? 2 * Cin : Cin;  // k-slots per (cout, tap) row of the packed filter
    const int Cres = RSEG ? a.r_C1 + a.r_C2 : 0;
    const int HS = Cres / 32;                                   // half-stages of the residual segment
    constexpr int RSL = 5;                                      // slots (half-stage j uses slot j % RSL)
    const int RT = HS + ((RSEG && (HS - 1) % RSL < 2) ? 1 : 0);  // ... and the empty stage that lets the last one leave A0 / R0 alone
    /* pixel slot k (BM rows x 64 B): 0, 1 in A0, 2 = R2, 3, 4 in A1;  filter slot k (128 rows x 64 B): 0, 1 = R0, 2, 3 = R1, 4 behind A1's pixel slots */
#define WS_RSA(k) ((k) < 2 ? lds + (k) * (BM * 64) : (k) == 2 ? lds + offR + 2 * R_BYTES : lds + offA1 + ((k) - 3) * (BM * 64))
#define WS_RSW(k) ((k) < 4 ? lds + offR + (k) * 8192 : lds + offA1 + 2 * (BM * 64))
    const int K = (ntiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;  // tiles of this workgroup
    const bool has_pro = a.scale || a.pro_silu || a.dmask;
    // scale / shift / mask of tile kt's image -> parameter buffer kt & 1 (all 512 threads).  Tiles 0 and 1 here; tile kt + 2
    // at the end of tile kt's epilogue (the producers' last use of that buffer - tile kt's last chunk - is behind them,
    // their first use for tile kt + 2 is nine stages before tile kt + 1 ends)
#define WS_FILL_PAR_N(KT, NTHR)                                           \
    {                                                                     \
        float* par = WS_PAR(KT);                                          \
        const TileXY ft_ = WS_TILE(KT);                                   \
        if (a.has_gni) {                                                  \
            ws_fill_par_gni(a, ft_.n0, ft_.ts == 0 && ft_.co0 == 0, par, Cin, tid, NTHR); \
        } else {                                                          \
            const int pn0 = ft_.n0;                                       \
            for (int c = tid; c < Cin; c += (NTHR)) {                     \
                const int so = pn0 * Cin + c;                             \
                const float sc_ = a.scale ? a.scale[so] : 1.f, sh_ = a.scale ? a.shift[so] : 0.f, dm_ = a.dmask ? a.dmask[so] : 1.f; \
                par[c] = sc_ * dm_;                                       \
                par[Cin + c] = sh_ * dm_;                                 \
                par[2 * Cin + c] = a.pro_silu ? -1.4426950408889634f * sc_ : 0.f;    \
                par[3 * Cin + c] = a.pro_silu ? -1.4426950408889634f * sh_ : -126.f; \
            }                                                             \
        }                                                                 \
    }
#define WS_FILL_PAR(KT) WS_FILL_PAR_N(KT, 512)
    // tile 0's rows: all 512 threads, the producers AFTER they have requested the first chunk's halo and the first two filter taps
    // (neither depends on the rows): the two round trips of a launch's start overlap instead of following each other.
    // tile 1's rows: by the consumer waves, below, while the producers bring in the first stage (the producers first read them nine
    // stages before tile 0 ends, behind dozens of workgroup barriers) - a second round trip off the preamble's critical path
    const int mTW = (1 << shTW) - 1, mTH = (1 << shTH) - 1;
    float gn_carry[2] = {0.f, 0.f};  // whole-image tiles finishing their norms: the group threads' first-pass (mean, M2)
    const bool estamp = tid == 0 && blockIdx.x == 0;  // diagnostic stamps of the epilogue passes (consumer wave 0, workgroup 0)
    int e_i = 0;
    // tile epilogue, 128 pixels per pass through R1|R2|A1: the consumer waves owning the pass's rows stage their
    // accumulators, then ALL 512 threads (the producers are between tiles) run the store loop
#define WS_ESTAMP() { if (estamp && a.stamps && e_i < 24) a.stamps[64 + e_i++] = (long long)clock64(); }
#define WS2_PASS(TT, P, STAGE_STMT)                                                                                                \
    {                                                                                                                              \
        constexpr int p = (P);                                                                                                     \
        auto pix_of = [&](int m) -> int {                                                                                          \
            const int mm = m + 128 * p;                                                                                            \
            const int tx = mm & mTW, ty = (mm >> shTW) & mTH;                                                                      \
            return ((TT).n0 * a.Hout + (TT).oy0 + ty) * a.Wout + (TT).ox0 + tx;                                                    \
        };                                                                                                                         \
        uint4 rpre[128 * (BN / (16 / (int)sizeof(TS))) / 512];                                                                     \
        if constexpr (!E16) conv_epilogue_res_prefetch<TS, 128, BN, 512>(a, (TT).co0, pix_of, rpre); /* in flight across the staging barrier */ \
        WS_ESTAMP()                                                                                                                \
        STAGE_STMT                                                                                                                 \
        WS_ESTAMP()                                                                                                                \
        lds_barrier();                                                                                                             \
        WS_ESTAMP()                                                                                                                \
        conv_epilogue_store<TS, 128, BN, 512, E16>(a, (TT).co0, (TT).n0, pix_of, stage, (BM / 128) * (TT).ts + p, E16 ? nullptr : rpre, a.n_gno ? p : -1, gn_carry); \
        WS_ESTAMP()                                                                                                                \
        lds_barrier(); /* everyone is done with the staging area (the output stores need no acknowledgement here) */                \
    }
    // consumer wave-row wr owns the 32-pixel blocks {wr, wr + 2, wr + 4, wr + 6} of the tile, so both wave-rows hold two
    // blocks of either epilogue pass and all four consumer waves stage at once
#define WS2_EPILOGUE(TT, KT, STAGE0, STAGE1)                                                                                       \
    {                                                                                                                              \
        float* stage = reinterpret_cast<float*>(WS_RING(1));                                                                       \
        WS2_PASS(TT, 0, STAGE0)                                                                                                    \
        if constexpr (BM == 256) { WS2_PASS(TT, 1, STAGE1) }                                                                       \
        if ((KT) + 2 < K) WS_FILL_PAR((KT) + 2)                                                                                    \
    }
    // E16: the whole tile staged at once ([256][BN + 8] T in R1 | R2 | A1), then the two store passes (their statistics per 128 pixels)
#define WS2_EPILOGUE16(TT, KT, STAGE_ALL)                                                                                          \
    {                                                                                                                              \
        T* st16 = reinterpret_cast<T*>(WS_RING(1));                                                                                \
        if ((KT) + 1 < K) WS_FILL_FOLD((KT) + 1)                                                                                   \
        WS_ESTAMP()                                                                                                                \
        STAGE_ALL                                                                                                                  \
        WS_ESTAMP()                                                                                                                \
        lds_barrier();                                                                                                             \
        WS_ESTAMP()                                                                                                                \
        _Pragma("unroll") for (int p = 0; p < 2; ++p) {                                                                            \
            auto pix_of = [&](int m) -> int {                                                                                      \
                const int mm = m + 128 * p;                                                                                        \
                const int tx = mm & mTW, ty = (mm >> shTW) & mTH;                                                                  \
                return ((TT).n0 * a.Hout + (TT).oy0 + ty) * a.Wout + (TT).ox0 + tx;                                                \
            };                                                                                                                     \
            conv_epilogue_store<TS, 128, BN, 512, true>(a, (TT).co0, (TT).n0, pix_of, reinterpret_cast<float*>(st16 + p * 128 * (BN + kStage16Pad)), \
                                                        2 * (TT).ts + p, nullptr, a.n_gno ? p : -1, gn_carry);                     \
            WS_ESTAMP()                                                                                                            \
        }                                                                                                                          \
        lds_barrier(); /* everyone is done with the staging area */                                                                \
        if ((KT) + 2 < K) WS_FILL_PAR((KT) + 2)                                                                                    \
    }
    if (producer) {
        const int ptid = tid & 255, cu = ptid & 7, urow = ptid >> 3, pw = ptid >> 6;
        const int Hv = a.up ? 2 * a.Hin : a.Hin, Wv = a.up ? 2 * a.Win : a.Win;
        int a_pix[PIPE_UA], a_sw[PIPE_UA];
        unsigned b_vo[UB];
#pragma unroll
        for (int i = 0; i < PIPE_UA; ++i) a_sw[i] = (urow + 32 * i) * A_PITCH + cu * (SPLIT ? 8 : 16);  // (SPLIT: the hi half; lo 64 bytes on)
#pragma unroll
        for (int k4 = 0; k4 < UB; ++k4) b_vo[k4] = (unsigned)(((urow + 32 * k4) * 9 * CinW + (cu ^ ((urow >> 1) & 7)) * EPV) * 2);
        auto pix_unit = [&](int i, const TileXY& t) __attribute__((always_inline)) -> int {
            const int row = urow + 32 * i;
            const int hy = (int)__umulhi((unsigned)row, g.magic_w), hx = row - hy * g.HWd;
            const int iy = t.oy0 - 1 + hy, ix = t.ox0 - 1 + hx;
            const bool in = iy >= 0 && iy < Hv && ix >= 0 && ix < Wv && !(a.up == 2 && ((iy | ix) & 1));
            const int sy = a.up ? (iy >> 1) : iy, sx = a.up ? (ix >> 1) : ix;
#ifdef WS_SAME_IMAGE  /* timing experiment (wrong results): every tile reads one of eight images, so the halo loads hit L2 */
            return row >= g.a_rows ? -2 : in ? ((t.n0 & 7) * a.Hin + sy) * a.Win + sx : -1;
#else
            return row >= g.a_rows ? -2 : in ? (t.n0 * a.Hin + sy) * a.Win + sx : -1;
#endif
        };
        auto set_pix = [&](int i, const TileXY& t) __attribute__((always_inline)) { a_pix[i] = pix_unit(i, t); };
        const char* wbase = (const char*)a.w;
        u32x4 areg[PIPE_UA];
        // this lane's 8 channels of the four parameter rows, as the pairs (2 d, 2 d + 1) the dwords of a halo vector hold: every
        // operand of the packed fp32 instructions below is a consecutive register pair as loaded
        f32x2 pS[4], pH[4], pS2[4], pH2[4];
        auto load_par = [&](const float* par, int c0) __attribute__((always_inline)) {
            const float* p = par + c0 + cu * EPR;
#pragma unroll
            for (int d = 0; d < EPR / 2; ++d) {
                pS[d] = *reinterpret_cast<const f32x2*>(p + 2 * d);
                pH[d] = *reinterpret_cast<const f32x2*>(p + Cin + 2 * d);
                pS2[d] = *reinterpret_cast<const f32x2*>(p + 2 * Cin + 2 * d);
                pH2[d] = *reinterpret_cast<const f32x2*>(p + 3 * Cin + 2 * d);
            }
        };
        auto load_A = [&](int i, int ch) __attribute__((always_inline)) {  // halo vector of 64-channel chunk `ch`
            constexpr int SB = SPLIT == 1 ? 4 : 2;  // bytes per source element
            const int c0 = ch * KCR;
            const bool second = c0 >= a.C1;
            const char* sbase = (const char*)(second ? a.src2 : a.src1) + (size_t)((second ? c0 - a.C1 : c0) * SB);
            const int Cs = second ? a.C2 : a.C1;
            const int px = a_pix[i] < 0 ? 0 : a_pix[i];
            const char* gp = sbase + (size_t)((unsigned)(px * Cs + cu * EPR) * (unsigned)SB);
            if constexpr (SPLIT == 2) {  // four 16-bit channels: 8 bytes
                const uint2 v2 = *reinterpret_cast<const uint2*>(gp);
                areg[i] = u32x4{v2.x, v2.y, 0u, 0u};
            } else {
                areg[i] = *reinterpret_cast<const u32x4*>(gp);
            }
        };
        // mode 0: prologue + write to LDS; 1 (RSEG, a tile's last chunk): prologue only, the activated vector stays in its register;
        // 2: write only (the segment's last stage)
        auto store_A = [&](int i, char* dstA, int mode = 0) __attribute__((always_inline)) {
            const int pix = a_pix[i], sw = a_sw[i];
            u32x4 val = areg[i];
            if constexpr (SPLIT != 0) {
                // four channels: prologue in packed fp32 (as below), then hi = f16(y), lo = f16(y - hi) -> 8 bytes each into the row's halves
                f32x2 y[2];
                if constexpr (SPLIT == 1) {
                    y[0] = f32x2{__uint_as_float(val[0]), __uint_as_float(val[1])};
                    y[1] = f32x2{__uint_as_float(val[2]), __uint_as_float(val[3])};
                } else {
                    typedef T tx4 __attribute__((ext_vector_type(4)));
                    const tx4 x = __builtin_bit_cast(tx4, make_uint2(val[0], val[1]));
                    y[0] = f32x2{(float)x[0], (float)x[1]};
                    y[1] = f32x2{(float)x[2], (float)x[3]};
                }
                if (has_pro) {  // (plain instructions: WS_PRO2 below)
#pragma unroll
                    for (int d = 0; d < 2; ++d) {
                        float y0, y1;
                        pro_pair_plain(y[d][0], y[d][1], pS[d], pH[d], pS2[d], pH2[d], y0, y1);
                        y[d] = f32x2{y0, y1};
                    }
                }
                typedef f16 hx4 __attribute__((ext_vector_type(4)));
                hx4 hi, lo;
#pragma unroll
                for (int d = 0; d < 2; ++d)
#pragma unroll
                    for (int k = 0; k < 2; ++k) {
                        const f16 hv = (f16)y[d][k];
                        hi[2 * d + k] = hv;
                        lo[2 * d + k] = (f16)sub_plain(y[d][k], (float)hv);
                    }
                uint2 hw = __builtin_bit_cast(uint2, hi), lw = __builtin_bit_cast(uint2, lo);
                if (pix < 0) hw = lw = make_uint2(0u, 0u);
                if (pix != -2) {
                    *reinterpret_cast<uint2*>(dstA + sw) = hw;
                    *reinterpret_cast<uint2*>(dstA + sw + 64) = lw;
                }
                return;
            }
            if (has_pro && mode != 2) {
                // per dword (two adjacent channels) in PLAIN fp32 instructions: four fmas, two exp2, two adds, two rcp, two multiplies,
                // one pack.  Round 3 wrote this in packed pairs (9 instructions per dword instead of 13) and round 2 had concluded that
                // "VALU work of this wave does not overlap the MFMAs of the consumer wave on its SIMD"; round 5's probe
                // (tools/issue_probe.py) shows both were the packed instructions: a v_pk_*_f32 beside a saturated matrix pipe costs its
                // wave 10-15 cycles that overlap nothing, the plain forms issue in the MFMAs' shadow.  Same values bit for bit
                // (an fma is an fma); consumer wait per GroupNorm stage 750-870 -> 120-140 cycles (tools/stamp_ws.py)
                typedef typename Vec8<T>::type tx8;
                const tx8 x = __builtin_bit_cast(tx8, val);
                tx8 o;
#pragma unroll
                for (int d = 0; d < 4; ++d) {
                    float xv[2], yv[2];
                    if constexpr (sizeof(T) == 2 && dtype_of<T>::value == DMME_BF16) {
                        xv[0] = __uint_as_float(val[d] << 16);
                        xv[1] = __uint_as_float(val[d] & 0xffff0000u);
                    } else {
                        xv[0] = (float)x[2 * d];
                        xv[1] = (float)x[2 * d + 1];
                    }
                    pro_pair_plain(xv[0], xv[1], pS[d], pH[d], pS2[d], pH2[d], yv[0], yv[1]);  // (no SiLU: S2 = 0, H2 = -126: the factor is exactly 1)
                    o[2 * d] = (T)yv[0];
                    o[2 * d + 1] = (T)yv[1];
                }
                val = __builtin_bit_cast(u32x4, o);
            }
            if (mode == 1) {
                areg[i] = val;
                return;
            }
            const u32x4 zero = {0u, 0u, 0u, 0u};
            if (pix < 0) val = zero;
            if (pix != -2) *reinterpret_cast<u32x4*>(dstA + sw) = val;
        };
        // one filter tap (cout tile co0, chunk c, tap t) -> ring slot: UB DMA instructions per wave.  Issued behind the compiler's back
        // (conv_common.h, glds16_hidden): hipcc counts only the halo loads, so its own waits for a halo register (ten younger loads:
        // vmcnt(10)) leave the youngest ten queue entries alone - the last two stages' DMAs among them - and the DMAs are retired by
        // the counted waits written out below.  This is what lets a tap be requested TWO stages ahead of its use: a DMA takes ~2.4 k
        // cycles from issue to landed under load, a stage ~2 k (stamps, tools/stamp_ws.py)
        auto dma_tap = [&](char* dstR, int co0, int c, int t) __attribute__((always_inline)) {
#if defined(WS_X_NO_DMA)  /* timing experiment (wrong results): the filter stream becomes 16-byte loads of ONE line per instruction */
            {
                const char* ub1 = wbase + ((size_t)co0 * 9 * CinW + (size_t)t * CinW + (size_t)c * KC) * 2;
                _Pragma("unroll") for (int k4 = 0; k4 < UB; ++k4) glds16_hidden(ub1 + (lane & 7) * 16, (unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned)(size_t)(lds_c*)dstR + (unsigned)(pw * 8 * ROW_DATA) + (unsigned)(32 * k4 * ROW_DATA))));
                return;
            }
#endif
            const unsigned lbase = (unsigned)(size_t)(lds_c*)dstR + (unsigned)(pw * 8 * ROW_DATA);
            const char* ub = wbase + ((size_t)co0 * 9 * CinW + (size_t)t * CinW + (size_t)c * KC) * 2;
#pragma unroll
            for (int k4 = 0; k4 < UB; ++k4) {
                const unsigned l = (unsigned)__builtin_amdgcn_readfirstlane((int)(lbase + (unsigned)(32 * k4 * ROW_DATA)));
                glds16_hidden(ub + b_vo[k4], l);
            }
        };
        // residual segment, half-stage j (32 raw channels) -> slot k: the tile's 256 pixels (4 DMA instructions per wave) and the
        // 128 x 32 filter block (2 per wave)
        // Pixel slot: wave pw sends rows (BM / 4) pw + 16 q + (lane >> 2) (q < BM / 64); filter slot: rows 32 pw + 16 q + (lane >> 2) (q < 2); LDS piece
        // lane & 3 of a 64-byte row holds source piece (lane & 3) ^ ((row >> 2) & 3) = (lane & 3) ^ ((lane >> 4) & 3).  The per-lane offsets
        // are recomputed per request from a laundered lane id - a dozen instructions per half-stage - instead of living in seven more
        // registers for the whole kernel (the allocator is at its 256-register limit here, and a spilled halo coordinate is reloaded
        // behind a full `vmcnt(0)` drain in every stage).
        constexpr int RA = BM / 64, RD = RA + 2;  // requests per wave: the pixels; pixels + filter block
        auto dma_rseg_A = [&](int k, const TileXY& t, int j) __attribute__((always_inline)) {
            int ln = lane;
            asm volatile("" : "+v"(ln));
            const int c0 = j * 32;
            const bool second = c0 >= a.r_C1;
            const int Cs = second ? a.r_C2 : a.r_C1;
            const int tile_pix0 = (t.n0 * a.Hin + t.oy0) * a.Win + t.ox0;
            const char* sb = (const char*)(second ? a.r_src2 : a.r_src1) + ((size_t)tile_pix0 * Cs + (second ? c0 - a.r_C1 : c0)) * 2;
            const unsigned la = (unsigned)(size_t)(lds_c*)WS_RSA(k) + (unsigned)(pw * (BM / 4) * 64);
            const int piece = ((ln & 3) ^ ((ln >> 4) & 3)) * 16;
#pragma unroll
            for (int q = 0; q < RA; ++q) {
                const int m = (BM / 4) * pw + 16 * q + (ln >> 2);
                const int rel = (m >> shTW) * a.Win + (m & mTW);
                const unsigned l = (unsigned)__builtin_amdgcn_readfirstlane((int)(la + (unsigned)(q * 1024)));
                glds16_hidden_s(sb, (unsigned)(rel * Cs * 2 + piece), l);
            }
        };
        auto dma_rseg_W = [&](int k, const TileXY& t, int j) __attribute__((always_inline)) {
            int ln = lane;
            asm volatile("" : "+v"(ln));
            const char* wb = (const char*)a.r_w + ((size_t)t.co0 * Cres + j * 32) * 2;
            const unsigned lw = (unsigned)(size_t)(lds_c*)WS_RSW(k) + (unsigned)(pw * 2048);
            const int piece = ((ln & 3) ^ ((ln >> 4) & 3)) * 16;
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const unsigned l = (unsigned)__builtin_amdgcn_readfirstlane((int)(lw + (unsigned)(q * 1024)));
                glds16_hidden_s(wb, (unsigned)((32 * pw + 16 * q + (ln >> 2)) * Cres * 2 + piece), l);
            }
        };
        TileXY tcur = WS_TILE(0), tnext = WS_TILE(K > 1 ? 1 : 0);
        {   // preamble: chunk 0 halo + tap 0 of the first tile; chunk 1 halo in flight
#pragma unroll
            for (int i = 0; i < PIPE_UA; ++i) {
                set_pix(i, tcur);
                load_A(i, 0);
            }
            dma_tap(WS_RING(0), tcur.co0, 0, 0);
            dma_tap(WS_RING(1), tcur.co0, 0, 1);
            WS_FILL_PAR(0)
            __syncthreads();
            load_par(WS_PAR(0), 0);
#pragma unroll
            for (int i = 0; i < PIPE_UA; ++i) store_A(i, WS_BUFA(0));
#pragma unroll
            for (int i = 0; i < PIPE_UA; ++i) load_A(i, 1);  // nchunks >= 2
            wait_vm_keep<PIPE_UA>();  // both taps have landed (they are older than the eleven loads just issued)
        }
        __syncthreads();
        int kt = 0, cc = 0, cg = 0;
        int p_i = 0;
        (void)p_i;
#define WS_A_UNIT(i)                                                        \
    {                                                                       \
        if (RSEG && last_c) { /* the next tile's first chunk stays in its registers until the residual segment is through */ \
            if (do_store) store_A((i), dstA, 1);                            \
        } else {                                                            \
            if (do_store) store_A((i), dstA);                               \
            WS_PSTT(1)                                                      \
            if (new_tile) set_pix((i), tnn);                                \
            WS_PSTT(2)                                                      \
            load_A((i), lc);                                                \
            WS_PSTT(3)                                                      \
        }                                                                   \
    }
#define WS_A_MATH(i) { if (do_store) store_A((i), dstA, 1); }  /* the prologue on the unit's registers, nothing written */
#define WS_A_WRITE(i)                                                        \
    {                                                                       \
        if (RSEG && last_c) { /* written in the residual segment's last stage */ } \
        else {                                                              \
            if (do_store) store_A((i), dstA, 2);                            \
            WS_PSTT(1)                                                      \
            if (new_tile) set_pix((i), tnn);                                \
            load_A((i), lc);                                                \
        }                                                                   \
    }
#define WS_A_ARRIVED(i) { asm volatile("" : "+v"(areg[i][0]), "+v"(areg[i][1]), "+v"(areg[i][2]), "+v"(areg[i][3])); }
        // stage TP = tap TP of chunk (kt, cc): DMA of the tap two stages on, 1-2 halo units of the next chunk, barrier.
        // Ring: stage s reads slot s % 3 (9 % 3 == 0: the same across chunks); the tap of stage s + 2 goes to slot (s + 2) % 3, which
        // stage s - 1 read last.  A tile's epilogue stages through R1 | R2 | A1, so across a tile boundary only tap 0 (R0) is sent
        // ahead; the new tile's first stage requests taps 1 and 2 together.
#ifdef WS_PSTAMPS
        // producer stamps go to LDS (4 KB behind the parameter rows) and to memory when the wave leaves: a global store per stamp would
        // join the vmcnt queue and shift every counted wait below by one
        long long* pst_lds = reinterpret_cast<long long*>(reinterpret_cast<char*>(par_base) + 2 * 4 * Cin * 4 + 1024);
#define WS_PSTT(TAG) { if (a.stamps && ptid == 0 && blockIdx.x == 0 && p_i < 500) pst_lds[p_i++] = (long long)clock64() | ((long long)(TAG) << 56); }
#else
#define WS_PSTT(TAG)
#endif
#define WS_PST() WS_PSTT(0)
#define WS_L(TP) (PIPE_UA == 11 ? (((TP) == 0 || (TP) == 8) ? 2 : 1) : ((TP) < 7 ? 1 : 0)) /* halo loads a stage issues (behind its DMA) */
        // RSEG, the LAST main chunk of a tile: no halo loads; behind their tap stages 5 / 6 request the pixels of the segment's half-stages
        // 0 / 1 (4 instructions each), stage 7 both filter blocks (2 + 2).  Requests a stage issues / of those, the ones behind its tap
#define WS_RD(TP) ((TP) <= 4 ? UB : (TP) <= 6 ? UB + RA : 0)
#define WS_RX(TP) (((TP) == 5 || (TP) == 6) ? RA : 0)
#define WS_PSTAGE(TP)                                                                                                   \
    {                                                                                                                   \
        WS_PST()                                                                                                        \
        const bool last_c = cc + 1 == nchunks;                                                                          \
        const bool have_n = !(last_c && kt + 1 == K);                                                                   \
        const int nk = have_n ? (last_c ? kt + 1 : kt) : kt, nc = have_n ? (last_c ? 0 : cc + 1) : cc;                  \
        const bool last_n = nc + 1 == nchunks;                                                                          \
        const bool have_l = have_n && !(last_n && nk + 1 == K);                                                         \
        const int lk = have_l ? (last_n ? nk + 1 : nk) : nk, lc = have_l ? (last_n ? 0 : nc + 1) : nc;                  \
        /* this stage's halo registers are taken as arrived before its DMA goes out (the compiler's counted wait sits    \
           here, in front of the stage's work) */                                                                       \
        if constexpr (PIPE_UA == 11 && SCHED2) {                                                                        \
            WS_A_ARRIVED(TP)                                                                                            \
            if ((TP) == 3) { WS_A_ARRIVED(9) }                                                                          \
            if ((TP) == 4) { WS_A_ARRIVED(10) }                                                                         \
        } else if constexpr (PIPE_UA == 11) {                                                                           \
            if ((TP) == 0) { WS_A_ARRIVED(0) WS_A_ARRIVED(1) }                                                          \
            else if ((TP) == 8) { WS_A_ARRIVED(9) WS_A_ARRIVED(10) }                                                    \
            else { WS_A_ARRIVED((TP) + 1) }                                                                             \
        } else if constexpr ((TP) < 7) { WS_A_ARRIVED(TP) }                                                             \
        __builtin_amdgcn_sched_barrier(0);                                                                              \
        WS_PST()                                                                                                        \
        bool sent = true;          /* this stage sent a tap two stages ahead */                                         \
        bool sent_next = false;    /* ... and, in front of it, the NEXT stage's tap (first stage of a later tile) */    \
        if constexpr (CDMA) {                                                                                           \
            /* the consumers request the taps of the stages in which this wave is the longer half (two halo units: 8, 0; the chunk's \
               last taps: 6, 7) - in stages 1-5, where it has slack, this wave does (WS_HYB): a tap request costs a consumer ~130 \
               cycles of matrix-pipe time per stage, a producer ~1.2 k cycles it has to spare there */                          \
            if constexpr (WS_HYB != 0 && (TP) >= 1 && (TP) <= 5) dma_tap(WS_RING(((TP) + 2) % 3), tcur.co0, cc, (TP) + 2);  \
        }                                                                                                               \
        else {                                                                                                          \
        if ((TP) == 0 && cc == 0 && kt > 0) {                                                                           \
            dma_tap(WS_RING(1), tcur.co0, 0, 1);                                                                        \
            sent_next = true;                                                                                           \
        }                                                                                                               \
        if ((TP) <= 6) dma_tap(WS_RING(((TP) + 2) % 3), tcur.co0, cc, (TP) + 2);                                        \
        else if (RSEG && last_c) { /* (the next tile's tap 0 goes out in the segment's last stage) */ }                 \
        else if ((TP) == 7) {                                                                                           \
            if (have_n) dma_tap(WS_RING(0), nk == kt ? tcur.co0 : tnext.co0, nc, 0);                                    \
            else sent = false;                                                                                          \
        } else {                                                                                                        \
            if (have_n && nk == kt) dma_tap(WS_RING(1), tcur.co0, nc, 1);                                               \
            else sent = false;                                                                                          \
        }                                                                                                               \
        }                                                                                                               \
        if constexpr (RSEG) {  /* pixel slots 0 / 1 lie in A0 (free since the previous chunk ended), filter slots 0 / 1 in R0 (tap 6 */ \
            if (last_c) {      /* was its last reader: free from stage 7 on - so the filter blocks of BOTH go out at stage 5 / 6   */ \
                if ((TP) == 5) dma_rseg_A(0, tcur, 0);                                                                  \
                if ((TP) == 6) dma_rseg_A(1, tcur, 1);                                                                  \
                if ((TP) == 7) { dma_rseg_W(0, tcur, 0); dma_rseg_W(1, tcur, 1); }                                      \
            }                                                                                                           \
        }                                                                                                               \
        __builtin_amdgcn_sched_barrier(0);                                                                              \
        WS_PST()                                                                                                        \
        {                                                                                                               \
            const bool do_store = have_n;                                                                               \
            char* dstA = WS_BUFA(cg + 1);                                                                               \
            const bool new_tile = have_l && lc == 0;                                                                    \
            const TileXY tnn = lk == kt ? tcur : tnext;                                                                 \
            if ((TP) == 0) load_par(WS_PAR(nk), nc* KCR);                                                               \
            if constexpr (PIPE_UA == 11 && SCHED2) {                                                                    \
                /* eleven units over nine stages WITHOUT a stage that carries two whole units: stage k takes unit k; the prologue \
                   arithmetic of units 9 / 10 (same chunk, same parameter rows) runs in stages 3 / 4 on their registers, their    \
                   stores + reloads in stages 6 / 7 - every stage <= 1.5 units of producer work */                               \
                WS_A_UNIT(TP)                                                                                           \
                if ((TP) == 3) { WS_A_MATH(9) }                                                                         \
                if ((TP) == 4) { WS_A_MATH(10) }                                                                        \
                if ((TP) == 6) { WS_A_WRITE(9) }                                                                        \
                if ((TP) == 7) { WS_A_WRITE(10) }                                                                       \
            } else if constexpr (PIPE_UA == 11) {                                                                       \
                if ((TP) == 0) { WS_A_UNIT(0) WS_A_UNIT(1) }                                                            \
                else if ((TP) == 8) { WS_A_UNIT(9) WS_A_UNIT(10) }                                                      \
                else { WS_A_UNIT((TP) + 1) }                                                                            \
            } else if constexpr ((TP) < 7) { WS_A_UNIT(TP) }                                                            \
        }                                                                                                               \
        __builtin_amdgcn_sched_barrier(0);                                                                              \
        WS_PST()                                                                                                        \
        /* the NEXT stage's tap must have landed; whatever was issued after it may stay in flight: the previous stage's  \
           halo loads, this stage's DMA (UB instructions) and halo loads - or, when that tap went out in this very stage, \
           only the second DMA and the loads */                                                                         \
        if constexpr (CDMA) {                                                                                           \
            /* no tap to retire: the halo prefetch is waited for where it is used (WS_A_ARRIVED), the halo stores by lgkmcnt; \
               the residual segment's first slots (requested in stages 5 - 7) must have landed when stage 8 hands over */ \
            if (RSEG && last_c) {                                                                                       \
                /* no halo loads in this chunk; requests per stage: the tap (stages 1-5), the segment's pixels behind it (5, 6), \
                   its first filter blocks (7).  Stage TP retires the tap stage TP - 1 sent; stage 8 hands over to half-stage 0 */ \
                if (WS_HYB != 0 && (TP) >= 1 && (TP) <= 6) wait_vm_keep<WS_RX(((TP) + 8) % 9) + ((TP) <= 5 ? UB : 0) + WS_RX(TP)>(); \
                else if ((TP) == 8) wait_vm_keep<2>();                                                                  \
                else wait_lgkm_all();                                                                                   \
            }                                                                                                           \
            else if (WS_HYB != 0 && (TP) >= 1 && (TP) <= 6) wait_vm_keep<WS_L(((TP) + 8) % 9) + ((TP) <= 5 ? UB : 0) + WS_L(TP)>(); \
            else wait_lgkm_all();                                                                                       \
        }                                                                                                               \
        else if (RSEG && last_c) {                                                                                      \
            /* stage 8 hands over to half-stage 0, whose filter block went out first in stage 7: only the second stays */ \
            if ((TP) == 0) wait_vm_keep<WS_L(8) + UB>();                                                                \
            else if ((TP) <= 6) wait_vm_keep<WS_RX(((TP) + 8) % 9) + WS_RD(TP)>();                                      \
            else if ((TP) == 7) wait_vm_keep<WS_RX(6) + 4>();  /* (tap 8, then stage 6's pixels and this stage's two blocks) */ \
            else wait_vm_keep<2>();                                                                                     \
        }                                                                                                               \
        else if (sent_next) wait_vm_keep<UB + WS_L(TP)>();                                                              \
        else if (sent) wait_vm_keep<WS_L(((TP) + 8) % 9) + UB + WS_L(TP)>();                                            \
        else wait_vm_keep<WS_L(((TP) + 8) % 9) + WS_L(TP)>();                                                           \
        WS_PST()                                                                                                        \
        __builtin_amdgcn_s_barrier();                                                                                   \
        if ((TP) == 8) {                                                                                                \
            ++cg;                                                                                                       \
            if (++cc == nchunks) { /* tile done: its epilogue (store loop shared with the consumers), then on */        \
                if constexpr (RSEG) {                                                                                   \
                    /* the residual segment: half-stage j reads slot j % 5.  Stage 0 requests 2, 3, 4 (their slots were the main loop's \
                       until now), stage j >= 1 requests j + 4 into the slot stage j - 1 read; requests retire in order, so the wait \
                       leaves the ones behind half-stage j + 1 in flight */ \
                    int sl = 2;                                                                                         \
                    for (int j = 0; j < RT; ++j) {                                                                      \
                        const bool fin = j + 1 == RT && kt + 1 < K;                                                     \
                        for (int h_ = (j == 0 ? 2 : j + 4); h_ <= j + 4 && h_ < HS; ++h_) {                             \
                            dma_rseg_A(sl, tcur, h_);                                                                   \
                            dma_rseg_W(sl, tcur, h_);                                                                   \
                            sl = sl == RSL - 1 ? 0 : sl + 1;                                                            \
                        }                                                                                               \
                        if (fin) { /* A0 and R0 have had their last reader: the next tile's chunk 0, tap 0; chunk 1 requested */ \
                            _Pragma("unroll") for (int i = 0; i < PIPE_UA; ++i) store_A(i, WS_BUFA(cg), 2);             \
                            dma_tap(WS_RING(0), tnext.co0, 0, 0);                                                       \
                            _Pragma("unroll") for (int i = 0; i < PIPE_UA; ++i) load_A(i, 1);                           \
                        }                                                                                               \
                        __builtin_amdgcn_sched_barrier(0);                                                              \
                        const int hmax = j + 4 < HS - 1 ? j + 4 : HS - 1;  /* youngest half-stage requested so far */   \
                        const int ahead = hmax - (j + 1);                   /* ... those behind the one stage j + 1 reads */ \
                        if (fin) wait_vm_keep<UB + PIPE_UA>();                                                          \
                        else if (ahead >= 3) wait_vm_keep<3 * RD>();                                                    \
                        else if (ahead == 2) wait_vm_keep<2 * RD>();                                                    \
                        else if (ahead == 1) wait_vm_keep<RD>();                                                        \
                        else wait_vm_keep<0>();                                                                         \
                        __builtin_amdgcn_s_barrier();                                                                   \
                    }                                                                                                   \
                }                                                                                                       \
                if constexpr (E16) { WS2_EPILOGUE16(tcur, kt, ;) } else { WS2_EPILOGUE(tcur, kt, ;, ;) }                         \
                /* a wait the COMPILER sees: the epilogue's conditional residual loads and their conditional uses are correlated \
                   branches it cannot prove, so it carried "a load into these registers may still be pending" around the loop \
                   and put `s_waitcnt vmcnt(0)` in front of stage 0's DMA - draining the halo prefetch at EVERY chunk start \
                   (the 1.6-2.2 k-cycle consumer wait at tap 0, tools/stamp_ws.py).  Here everything in flight (the next    \
                   tile's second chunk, its tap 0) was requested before the epilogue: it has landed */                       \
                /* (measured: that wait cost nothing - the producers were waiting for the barrier anyway - but a visible    \
                   vmcnt(0) HERE also waits for the store loop's output stores, ~1.5 k cycles per tile: not placed) */        \
                cc = 0;                                                                                                 \
                ++kt;                                                                                                   \
                tcur = tnext;                                                                                           \
                tnext = WS_TILE(kt + 1 < K ? kt + 1 : kt);                                                              \
            }                                                                                                           \
        }                                                                                                               \
    }
        const int GCH = K * nchunks;
        for (int ch = 0; ch < GCH; ++ch) {
            WS_PSTAGE(0) WS_PSTAGE(1) WS_PSTAGE(2) WS_PSTAGE(3) WS_PSTAGE(4) WS_PSTAGE(5) WS_PSTAGE(6) WS_PSTAGE(7) WS_PSTAGE(8)
        }
#ifdef WS_PSTAMPS
        if (a.stamps && ptid == 0 && blockIdx.x == 0)
            for (int i = 0; i < p_i; ++i) a.stamps[128 + i] = pst_lds[i];
#endif
#undef WS_PSTAGE
#undef WS_RD
#undef WS_RX
#undef WS_L
#undef WS_PST
#undef WS_PSTT
#undef WS_A_UNIT
#undef WS_A_MATH
#undef WS_A_WRITE
#undef WS_A_ARRIVED
        return;
    }

    // ---- consumers: wave tile 128 pixels x 64 couts ----
    WS_FILL_PAR(0)
    WS_FILL_FOLD(0)
    __syncthreads();
    if (K > 1) WS_FILL_PAR_N(1, 256)
    __builtin_amdgcn_s_setprio(3);  // the MFMA stream wins issue arbitration against the producer wave of its SIMD (~1 %; giving the
                                    // priority to the producers instead changes nothing: their GN-mode stage is not an issue-slot problem)
    const int wrow = wave >> 1, wn0 = (wave & 1) * 64;
    const int r = lane & 31, h = lane >> 5;
    int a_row[MI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
        const int m = (2 * mi + wrow) * 32 + r;  // interleaved 32-pixel blocks (see the epilogue)
        const int tx = m & mTW, ty = (m >> shTW) & mTH;
        a_row[mi] = (ty * g.HWd + tx) * A_PITCH + h * 16;  // byte offset of this lane's fragment row (tap 0, k-group 0)
    }
    int b_off[NI][4];  // byte offset of this lane's filter fragment per k-group (the ring stays XOR-swizzled: the DMA writes lane-linear)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
        const int row = wn0 + ni * 32 + r;
#pragma unroll
        for (int kg = 0; kg < 4; ++kg) b_off[ni][kg] = row * ROW_DATA + (((kg * 2 + h) ^ ((row >> 1) & 7)) << 4);
    }
    // CDMA: wave w requests rows 8 w + 32 k4 + (lane >> 3) of a tap (k4 < UB), LDS piece lane & 7 <- source piece (lane & 7) ^ ((row >> 1) & 7)
    unsigned cb_vo[UB];
#pragma unroll
    for (int k4 = 0; k4 < UB; ++k4) {
        const int urow_ = wave * 8 + (lane >> 3);
        cb_vo[k4] = (unsigned)(((urow_ + 32 * k4) * 9 * CinW + ((lane & 7) ^ ((urow_ >> 1) & 7)) * EPV) * 2);
    }
    auto cdma = [&](int slot, int co0, int c, int t) __attribute__((always_inline)) {
        const char* ub = (const char*)a.w + ((size_t)co0 * 9 * CinW + (size_t)t * CinW + (size_t)c * KC) * 2;
        const unsigned lbase = (unsigned)(size_t)(lds_c*)WS_RING(slot) + (unsigned)(wave * 8 * ROW_DATA);
#pragma unroll
        for (int k4 = 0; k4 < UB; ++k4)
            glds16_hidden_s(ub, cb_vo[k4], (unsigned)__builtin_amdgcn_readfirstlane((int)(lbase + (unsigned)(32 * k4 * ROW_DATA))));
    };
    // diagnostic cycle stamps (a.stamps null: off): consumer wave 0 of workgroup 0, [arrive, leave] of every barrier
    int stamp_i = 0;
    const bool stamping = a.stamps && tid == 0 && blockIdx.x == 0;
#define WS_STAMP() { if (stamping && stamp_i < 64) a.stamps[stamp_i++] = (long long)clock64(); }
    WS_STAMP()
    __syncthreads();  // preamble tiles are in LDS
    WS_STAMP()
    int cg = 0;
    for (int kt = 0; kt < K; ++kt) {
        const TileXY t = WS_TILE(kt);
        f32x16 acc[MI][NI];
        if constexpr (E16) {
            // transposed accumulators: lane = pixel r, register j of tile (mi, ni) = cout wn0 + 32 ni + 8 (j >> 2) + 4 h + (j & 3); they
            // start from the tile's bias + time row
            const float* fb = fold_base + (kt & 1) * BN + wn0 + 4 * h;
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4 f4 = *reinterpret_cast<const f32x4*>(fb + 32 * ni + 8 * q);
#pragma unroll
                    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                        for (int k = 0; k < 4; ++k) acc[mi][ni][4 * q + k] = f4[k];
                }
        } else {
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                for (int j = 0; j < 16; ++j) acc[mi][ni][j] = 0.f;
        }
#if defined(WS_X_NO_FRAGS)  /* timing experiments (wrong results): no fragment reads / no MFMAs */
#define WS_FRAGS(SET, KG)                                                                                                         \
    {                                                                                                                             \
        _Pragma("unroll") for (int mi = 0; mi < MI; ++mi) { u32x4 t_; asm volatile("" : "=v"(t_)); af[SET][mi] = __builtin_bit_cast(uint4, t_); }   \
        _Pragma("unroll") for (int ni = 0; ni < NI; ++ni) { u32x4 t_; asm volatile("" : "=v"(t_)); bfr[SET][ni] = __builtin_bit_cast(uint4, t_); }  \
        __builtin_amdgcn_sched_barrier(0);                                                                                        \
    }
#else
#define WS_FRAGS(SET, KG)                                                                                                         \
    {                                                                                                                             \
        _Pragma("unroll") for (int mi = 0; mi < MI; ++mi) af[SET][mi] = *reinterpret_cast<const uint4*>(pa[mi] + (KG) * 32);      \
        _Pragma("unroll") for (int ni = 0; ni < NI; ++ni) bfr[SET][ni] = *reinterpret_cast<const uint4*>(ldsR + b_off[ni][KG]);   \
        __builtin_amdgcn_sched_barrier(0);                                                                                        \
    }
#endif
#if defined(WS_X_NO_MFMA)
#define WS_MMAS(SET)                                                                                  \
    {                                                                                                 \
        _Pragma("unroll") for (int mi = 0; mi < MI; ++mi) asm volatile("" ::"v"(__builtin_bit_cast(u32x4, af[SET][mi])));   \
        _Pragma("unroll") for (int ni = 0; ni < NI; ++ni) asm volatile("" ::"v"(__builtin_bit_cast(u32x4, bfr[SET][ni])));  \
        __builtin_amdgcn_sched_barrier(0);                                                            \
    }
#else
#define WS_MMAS(SET)                                                                                  \
    {                                                                                                 \
        _Pragma("unroll") for (int mi = 0; mi < MI; ++mi)                                             \
            _Pragma("unroll") for (int ni = 0; ni < NI; ++ni) {                                       \
                if constexpr (E16) mma_group(bfr[SET][ni], af[SET][mi], acc[mi][ni], (T*)nullptr);    \
                else mma_group(af[SET][mi], bfr[SET][ni], acc[mi][ni], (T*)nullptr);                  \
            }                                                                                         \
        __builtin_amdgcn_sched_barrier(0);                                                            \
    }
#endif
#define WS_FA(SET, KG) { _Pragma("unroll") for (int mi = 0; mi < MI; ++mi) af[SET][mi] = *reinterpret_cast<const uint4*>(pa[mi] + (KG) * 32); }
#define WS_FB(SET, KG) { _Pragma("unroll") for (int ni = 0; ni < NI; ++ni) bfr[SET][ni] = *reinterpret_cast<const uint4*>(ldsR + b_off[ni][KG]); }
#define WS_SB() __builtin_amdgcn_sched_barrier(0);
#define WS_MM(SA, SB_)                                                                                \
    {                                                                                                 \
        _Pragma("unroll") for (int mi = 0; mi < MI; ++mi)                                             \
            _Pragma("unroll") for (int ni = 0; ni < NI; ++ni) mma_group(af[SA][mi], bfr[SB_][ni], acc[mi][ni], (T*)nullptr); \
        __builtin_amdgcn_sched_barrier(0);                                                            \
    }
        const TileXY tn = WS_TILE(kt + 1 < K ? kt + 1 : kt);
        const bool three = !a.mix2;
        for (int c = 0; c < nchunks; ++c, ++cg) {
            const char* ldsA = WS_BUFA(cg);
            const bool last_c = c + 1 == nchunks;
#pragma unroll WS_TP_UNROLL
            for (int tp = 0; tp < 9; ++tp) {
                const char* ldsR = WS_RING(tp % 3);
                const int tap_b = ((tp / 3) * g.HWd + (tp % 3)) * A_PITCH;
                const char* pa[MI];
#pragma unroll
                for (int mi = 0; mi < MI; ++mi) pa[mi] = ldsA + a_row[mi] + tap_b;
                uint4 af[2][MI], bfr[2][NI];
                // CDMA: the tap of stage s + 2 into ring slot (s + 2) % 3, which stage s - 1 read last (every wave is behind that stage's
                // barrier); a tile's epilogue stages through R1 | R2, so across a tile boundary only tap 0 (R0) goes ahead and the new
                // tile's first stage requests taps 1 and 2 together; with the residual segment the producers send the next tile's
                // tap 0 in the segment's last stage.  Behind the stage's first fragment reads: the matrix pipe waits for those anyway
                bool sent = false;
#define WS_CDMA_ISSUE()                                                                                               \
    if constexpr (CDMA) {                                                                                             \
        if (tp == 0 && c == 0 && kt > 0) cdma(1, t.co0, 0, 1);                                                        \
        if (tp <= 6) { if (WS_HYB == 0 || tp == 0 || tp == 6) { cdma((tp + 2) % 3, t.co0, c, tp + 2); sent = true; } } \
        else if (RSEG && last_c) { }                                                                                  \
        else if (tp == 7) { if (!(last_c && kt + 1 == K)) { cdma(0, last_c ? tn.co0 : t.co0, last_c ? 0 : c + 1, 0); sent = true; } } \
        else if (!last_c) { cdma(1, t.co0, c + 1, 1); sent = true; }                                                  \
        __builtin_amdgcn_sched_barrier(0);                                                                            \
    }
                if constexpr (SPLIT != 0) {
                    // k-groups 0, 1 = hi halves of the chunk's 32 channels, 2, 3 = lo halves; two A slots and two B slots as in the plain
                    // schedule, every fragment set requested one MFMA group (8 MFMAs) ahead of its first use
                    // (ConvArgs::mix2: the two hi.lo groups - the filter's lo half - are skipped; the fragment registers pass through
                    //  the same states either way)
                    WS_FA(0, 0) WS_FB(0, 0) WS_FB(1, 2) WS_SB()   // a0 = A.hi[0], b0 = W.hi[0], b1 = W.lo[0]
                    WS_CDMA_ISSUE()
                    WS_MM(0, 0)                                   // hi.hi, channels 0-15
                    WS_FA(1, 2) WS_SB()                           // a1 = A.lo[0]
                    if (three) { WS_MM(0, 1) }                    // hi.lo
                    WS_FA(0, 1) WS_FB(1, 1) WS_SB()               // a0 = A.hi[1], b1 = W.hi[1]
                    WS_MM(1, 0)                                   // lo.hi
                    WS_FB(0, 3) WS_FA(1, 3) WS_SB()               // b0 = W.lo[1], a1 = A.lo[1]
                    WS_MM(0, 1)                                   // hi.hi, channels 16-31
                    if (three) { WS_MM(0, 0) }                    // hi.lo
                    WS_MM(1, 1)                                   // lo.hi
                } else {
                    WS_FRAGS(0, 0) WS_FRAGS(1, 1) WS_CDMA_ISSUE() WS_MMAS(0) WS_FRAGS(0, 2) WS_MMAS(1) WS_FRAGS(1, 3) WS_MMAS(0) WS_MMAS(1)
                }
#undef WS_CDMA_ISSUE
                WS_STAMP()
                if constexpr (CDMA) {  // the NEXT stage's tap has landed (this stage's own request may stay in flight)
                    if (sent) wait_vm_keep<UB>();
                    else wait_vm_keep<0>();
                }
                __syncthreads();
                WS_STAMP()
            }
        }
        if constexpr (RSEG) {
            // the residual segment: half-stage j = 32 raw input channels, pixels and filter block in slot j % 5 (rows of 64 bytes: piece
            // (k-group * 2 + h) ^ ((row >> 2) & 3), and (row >> 2) & 3 == (r >> 2) & 3 for every row this lane reads); then the empty stage
            const int rsw = (r >> 2) & 3;
            const int rA = (wrow * 32 + r) * 64, rB = (wn0 + r) * 64;
            const int pk0 = (h ^ rsw) << 4, pk1 = ((2 + h) ^ rsw) << 4;
            int slot = 0;
            for (int j = 0; j < HS; ++j) {
                const char* sa = WS_RSA(slot) + rA;
                const char* sw = WS_RSW(slot) + rB;
                slot = slot == RSL - 1 ? 0 : slot + 1;
                uint4 af[2][MI], bfr[2][NI];
#pragma unroll
                for (int mi = 0; mi < MI; ++mi) af[0][mi] = *reinterpret_cast<const uint4*>(sa + mi * 4096 + pk0);
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) bfr[0][ni] = *reinterpret_cast<const uint4*>(sw + ni * 2048 + pk0);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int mi = 0; mi < MI; ++mi) af[1][mi] = *reinterpret_cast<const uint4*>(sa + mi * 4096 + pk1);
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) bfr[1][ni] = *reinterpret_cast<const uint4*>(sw + ni * 2048 + pk1);
                __builtin_amdgcn_sched_barrier(0);
                WS_MMAS(0) WS_MMAS(1)
                WS_STAMP()
                __syncthreads();
                WS_STAMP()
            }
            for (int j = HS; j < RT; ++j) __syncthreads();
        }
#undef WS_FA
#undef WS_FB
#undef WS_SB
#undef WS_MM
#undef WS_FRAGS
#undef WS_MMAS
        // the last stage read R2 and A1: R1|R2|A1 stages the epilogue
        if constexpr (E16) {
            // a register quad = four consecutive couts of pixel m: two packed conversions, one 8-byte store
            WS2_EPILOGUE16(t, kt, (ws_stage16<T, BN, MI, NI>(acc, st16 + (wrow * 32 + r) * (BN + kStage16Pad) + wn0 + 4 * h));)
        } else {
        WS2_EPILOGUE(t, kt, (conv_epilogue_stage<T, BN, 2, NI, 64>(a, reinterpret_cast<f32x16(&)[2][NI]>(acc[0]), t.co0, wn0, r, h, wrow * 32, t.n0, stage));,
                     (conv_epilogue_stage<T, BN, 2, NI, 64>(a, reinterpret_cast<f32x16(&)[2][NI]>(acc[2]), t.co0, wn0, r, h, wrow * 32, t.n0, stage));)
        }
        WS_STAMP()
    }
#undef WS_STAMP
#undef WS2_EPILOGUE
#undef WS2_EPILOGUE16
#undef WS_FILL_FOLD
#undef WS2_PASS
#undef WS_FILL_PAR
#undef WS_FILL_PAR_N
#undef WS_ESTAMP
#undef WS_BUFA
#undef WS_RING
#undef WS_RSA
#undef WS_RSW
#undef WS_PAR
#undef WS_TILE
}

static size_t ws2_lds(const ConvArgs& a, const ConvTile& g) {
    const size_t a_bytes = (size_t)g.a_rows * (ROW_DATA + 16);
    size_t extra = 0;
#ifdef WS_PSTAMPS
    extra = 4096;
#endif
    return 2 * a_bytes + 3 * (size_t)128 * ROW_DATA + (size_t)ws2_stage_pad((int)a_bytes) + (size_t)2 * 4 * (a.C1 + a.C2) * 4 + 1024 /* E16: fold rows */ + extra;
}

// the wave-specialised kernel applies (else 0): its tile goes to g
static int ws_pick(const ConvArgs& a, ConvTile& g) {
    constexpr int ws_min_tiles = 256;
    const int Cin = a.C1 + a.C2, nch = Cin / 64;
    if (a.taps != 9 || a.stride != 1 || Cin % 64 || a.C1 % 64 || nch < 2 || nch % 2 || Cin > 512 || a.out_silu || a.out_nchw || a.in_nchw || a.Cout % 128) return 0;
    ConvTile t;
    // staging area R1|R2|A1 must hold 128 x 128 floats; the halo must fit 11 units per producer lane
    if (make_tile(a, 256, 128, t) && t.TN == 1 && t.a_rows <= 352 && ws2_lds(a, t) <= 160 * 1024 && (size_t)t.a_rows * ROW_DATA + 2 * 128 * ROW_DATA >= 64 * 1024 &&
        t.tiles_m * t.tiles_n >= ws_min_tiles) {
        g = t;
        return 3;
    }
    // 128-pixel tiles where those fill the chip and the 256-pixel ones do not (return value 4)
    ConvTile u;
    if (!debug_route("no_ws128") && make_tile(a, 128, 128, u) && u.TN == 1 && u.a_rows <= 224 && ws2_lds(a, u) <= 160 * 1024 && u.tiles_m * u.tiles_n >= ws_min_tiles) {
        g = u;
        return 4;
    }
    return 0;
}

// the split-pass form of the wave-specialised kernel (ConvArgs::mix 1 / 2: fp32 tensors, three fp16 passes): 256-pixel tiles of one
// image, whole 32-channel blocks per source, whole 128-cout tiles; any number of tiles (a small batch just leaves workgroups idle)
static bool ws2s_pick(const ConvArgs& a, ConvTile& g) {
    const int Cin = a.C1 + a.C2;
    if ((a.mix != 1 && a.mix != 2) || a.taps != 9 || a.stride != 1 || a.up == 2 || Cin % 64 || a.C1 % 32 || Cin < 64 || Cin > 512 || a.out_silu || a.out_nchw || a.in_nchw ||
        a.Cout % 128 || a.res2 || a.n_gno || (a.tproj && a.nt != 1 && a.nt != a.N))
        return false;
    if ((int64_t)a.Cout * 9 * 2 * Cin >= (1ll << 30) || (int64_t)a.N * a.Hin * a.Win * Cin >= (1ll << 29)) return false;  // 32-bit byte offsets
    ConvTile t;
    if (!make_tile(a, 256, 128, t) || t.TN != 1 || t.a_rows > 352 || ws2_lds(a, t) > 160 * 1024 || (size_t)t.a_rows * ROW_DATA + 2 * 128 * ROW_DATA < 64 * 1024) return false;
    g = t;
    return true;
}
// candidate kernels: {BM, BN, GT}; the 9-tap variant serves layers with too little work per interval
// (few workgroups or stride 2) and owns a larger LDS footprint
static const int kPipeCand[5][3] = {{128, 128, 3}, {128, 64, 3}, {64, 64, 3}, {64, 64, 9}, {64, 64, 3}};
static const int kPipeUA[5] = {8, 8, 8, 11, 10};  // [4]: stride 2 (a 64-pixel tile's halo is 17 x 17 rows), two workgroups per CU
static size_t pipe_lds(const ConvTile& g, int BN, int GT) { return (size_t)g.a_rows * ROW_DATA + (size_t)GT * BN * ROW_DATA; }  // >= BM*BN*4 always
// the DMA filter path (bf16, GT = 3, 64-cout tiles) needs a second filter buffer inside the 2-workgroups-per-CU budget
static bool pipe_dma_ok(int dtype, const ConvTile& g, int BN, int GT) {
    const bool off = (debug_route("no_pipe_dma") != 0);
    return !off && is16(dtype) && GT == 3 && BN == 64 && pipe_lds(g, BN, GT) + (size_t)GT * BN * ROW_DATA <= 80 * 1024;
}

static int ilog2(int v) {
    int s = 0;
    while ((1 << s) < v) ++s;
    return s;
}

static bool pipe_fits(const ConvArgs& a, int i, ConvTile& t) {
    if (!make_tile(a, kPipeCand[i][0], kPipeCand[i][1], t)) return false;
    if (t.a_rows * 8 > 256 * kPipeUA[i]) return false;
    if (pipe_lds(t, kPipeCand[i][1], kPipeCand[i][2]) > (kPipeCand[i][2] == 9 ? 128 : 80) * 1024) return false;
    if (a.Cout <= 64 && kPipeCand[i][1] > 64) return false;
    return true;
}

static int pipe_pick(const ConvArgs& a, ConvTile& g) {
    int pick = -1;
    if (a.stride == 1)
        for (int i = 0; i < 3; ++i) {
            ConvTile t;
            if (!pipe_fits(a, i, t)) continue;
            pick = i;
            g = t;
            if ((int64_t)t.tiles_m * t.tiles_n >= min_wgs()) break;
        }
    // fewer workgroups than two per CU (or stride 2): nothing overlaps a load round trip but this workgroup's own
    // matrix work, so stage the whole 3x3 filter per interval
    // (forcing the 9-tap intervals on the 512-workgroup 8x8 layers is 45 % slower: 33 vs 22.5 us)
    if (pick < 0 || (pick == 2 && (int64_t)g.tiles_m * g.tiles_n < 2 * 256)) {
        ConvTile t;
        if (pipe_fits(a, 3, t)) {
            pick = 3;
            g = t;
        }
    }
    // stride 2: a 64-pixel tile's 17 x 17 halo and a whole 3x3 filter leave one workgroup per CU (110 KB); with 3-tap intervals two fit
    // (DownSample 32 -> 16 at B = 128: 65.8 -> 43.6 us, 16 -> 8: 59.8 -> 38.5 us)
    constexpr bool s2_gt9 = false;  // (stride-2 convs on whole-filter intervals, one workgroup per CU: measured slower)
    if (!s2_gt9 && a.stride == 2) {
        ConvTile t;
        if (pipe_fits(a, 4, t) && (int64_t)t.tiles_m * t.tiles_n >= 2 * 256) {
            pick = 4;
            g = t;
        }
    }
    return pick;
}

static bool rseg_shape_ok(const ConvArgs& a) {
    const int Cres = a.r_C1 + a.r_C2;
    return a.r_src1 && a.r_bias && a.r_C1 > 0 && a.r_C1 % 64 == 0 && a.r_C2 >= 0 && (a.r_C2 == 0 || a.r_src2) && Cres % 128 == 0 && Cres <= 512 && !a.up && !a.res1 &&
           !a.res2 && (int64_t)a.Cout * Cres < (1ll << 30) && (int64_t)a.N * a.Hin * a.Win * Cres < (1ll << 30);
}
bool conv_pipe_rseg_supported(int dtype, const ConvArgs& a) {
    if (!is16(dtype) || a.mix || !a.r_w || getenv("DMME_NO_WS") || !rseg_shape_ok(a) || !conv_pipe_supported(dtype, a)) return false;
    ConvTile gw{};
    return ws_pick(a, gw) != 0;
}

bool conv_pipe_supported(int dtype, const ConvArgs& a) {
    if (a.mix) {
        ConvTile gs{};
        return dtype == DMME_F16 && a.taps == 9 && ws2s_pick(a, gs);
    }
    if (!conv_mfma_supported(dtype, a)) return false;
    if (a.taps != 9 || (a.stride != 1 && a.stride != 2)) return false;
    if ((int64_t)a.Cout * 9 * (a.C1 + a.C2) >= (1ll << 31)) return false;
    if ((int64_t)a.N * a.Hin * a.Win >= (1ll << 31)) return false;
    ConvTile g;
    return pipe_pick(a, g) >= 0;
}

template <typename K>
static int set_lds_limit(K kernel, size_t bytes) {
    DMME_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    return DMME_OK;
}

static int launch_ws2_split(const ConvArgs& a, hipStream_t s) {
    ConvTile gw{};
    DMME_REQUIRE(ws2s_pick(a, gw), DMME_ERR_UNSUPPORTED, "conv3x3 (split fp16 passes): unsupported shape");
    static bool attr = false;
    if (!attr) {
        int rc0 = set_lds_limit(conv3x3_ws2_kernel<11, f16, 256, 1>, 160 * 1024);
        if (rc0 == DMME_OK) rc0 = set_lds_limit(conv3x3_ws2_kernel<11, f16, 256, 2>, 160 * 1024);
        if (rc0 != DMME_OK) return rc0;
        attr = true;
    }
    const int ntiles = gw.tiles_m * gw.tiles_n;
    const dim3 wgrid((unsigned)(ntiles < 256 ? ntiles : 256));
    if (a.mix == 2)
        hipLaunchKernelGGL((conv3x3_ws2_kernel<11, f16, 256, 2>), wgrid, dim3(512), ws2_lds(a, gw), s, a, gw, ilog2(gw.TW), ilog2(gw.TH), ntiles);
    else
        hipLaunchKernelGGL((conv3x3_ws2_kernel<11, f16, 256, 1>), wgrid, dim3(512), ws2_lds(a, gw), s, a, gw, ilog2(gw.TW), ilog2(gw.TH), ntiles);
    DMME_CHECK_LAUNCH();
    return DMME_OK;
}

// the 64 x 64 four-wave instances (8x8 / 4x4 maps, small batches) give way to the K-split-over-waves kernel (conv_kw.hip): pick 3
// (layers with fewer than 512 workgroups) always; the 512-workgroup 8x8 layers only where its 128-pixel tile applies (with 64-pixel
// tiles the two-per-CU four-wave kernel is as fast: 20.8 vs 21.4 us)
static bool kw_takes(int dtype, const ConvArgs& a, int pick, ConvTile& gk, int* ni, int* ring, int* bm) {
    constexpr int all = 0;
    if (pick != 3 && pick != 2) return false;
    if (!conv_kw_pick(dtype, a, gk, ni, ring, bm)) return false;
    return pick == 3 || all || *bm == 128;
}
// (a split of K over workgroups for this kernel was measured and removed: B = 1: 775 -> 846 steps/s without it, B = 8: 728 -> 792,
// B = 32: 586 -> 611 - a workgroup's fixed cost is ~5 us whatever its share of K, the finish kernel is one more dependent launch, and
// an unsplit conv finishes its norms itself)
static int kw_ksplit(const ConvArgs&, const ConvTile&) { return 1; }
template <typename T>
static int launch_kw_t(const ConvArgs& a, const ConvTile& gk, int ni, int ring, int bm, hipStream_t s) {
    const int ksplit = kw_ksplit(a, gk);
    const int rc = launch_conv_kw(dtype_of<T>::value, a, gk, ni, ring, bm, ksplit, s);
    if (rc != DMME_OK) return rc;
    if (ksplit > 1) {
        const int64_t total4 = (int64_t)a.N * a.Hout * a.Wout * (a.Cout / 4);
        hipLaunchKernelGGL(conv_splitk_finish_kernel<T>, dim3((unsigned)((total4 + 255) / 256)), dim3(256), 0, s, a, ksplit, total4);
        DMME_CHECK_LAUNCH();
    }
    return DMME_OK;
}

template <typename T, bool ACC3 = false>
static int launch_pipe_t(const ConvArgs& a, hipStream_t s) {
    ConvTile g{};
    const int pick = pipe_pick(a, g);
    DMME_REQUIRE(pick >= 0, DMME_ERR_UNSUPPORTED, "conv_pipe: no tile fits");
    const int ksplit = pipe_ksplit(a, g, pick, Frag<T>::KC);
    const dim3 grid((unsigned)(g.tiles_m * g.tiles_n), (unsigned)ksplit);
    if constexpr (sizeof(T) == 2) {
        const bool ws_off = getenv("DMME_NO_WS") != nullptr;
        ConvTile gw{};
        const int ws = ws_off ? 0 : ws_pick(a, gw);
        if (ws) {
            static bool ws_attr = false;  // (one flag per instantiation of this launcher, i.e. per T)
            if (!ws_attr) {
                int rc0 = set_lds_limit(conv3x3_ws2_kernel<11, T, 256>, 160 * 1024);
                if (rc0 == DMME_OK) rc0 = set_lds_limit(conv3x3_ws2_kernel<7, T, 128>, 160 * 1024);
                if (rc0 == DMME_OK) rc0 = set_lds_limit(conv3x3_ws2_kernel<11, T, 256, 0, true>, 160 * 1024);
                if (rc0 == DMME_OK) rc0 = set_lds_limit(conv3x3_ws2_kernel<7, T, 128, 0, true>, 160 * 1024);
                if (rc0 == DMME_OK) rc0 = set_lds_limit(conv3x3_ws2_kernel<11, T, 256, 0, false, true>, 160 * 1024);
                if (rc0 == DMME_OK) rc0 = set_lds_limit(conv3x3_ws2_kernel<11, T, 256, 0, true, true>, 160 * 1024);
                if (rc0 != DMME_OK) return rc0;
                ws_attr = true;
            }
            const int ntiles = gw.tiles_m * gw.tiles_n;
            const dim3 wgrid((unsigned)(ntiles < 256 ? ntiles : 256));
            // no residual INPUT tensor (the blocks' first convs, convs with a residual segment, data gradients that do not accumulate):
            // the 16-bit staged epilogue on transposed accumulators (DMME_DEBUG_ROUTE=no_ws_e16: the fp32-staged two-pass form)
            const bool e16 = !a.res1 && !a.res2 && !debug_route("no_ws_e16");
            if (a.r_w) {
                DMME_REQUIRE(rseg_shape_ok(a), DMME_ERR_UNSUPPORTED, "conv3x3 with a residual segment: raw channel counts outside the wave-specialised kernel's domain");
                if (ws == 4)
                    hipLaunchKernelGGL((conv3x3_ws2_kernel<7, T, 128, 0, true>), wgrid, dim3(512), ws2_lds(a, gw), s, a, gw, ilog2(gw.TW), ilog2(gw.TH), ntiles);
                else if (e16)
                    hipLaunchKernelGGL((conv3x3_ws2_kernel<11, T, 256, 0, true, true>), wgrid, dim3(512), ws2_lds(a, gw), s, a, gw, ilog2(gw.TW), ilog2(gw.TH), ntiles);
                else
                    hipLaunchKernelGGL((conv3x3_ws2_kernel<11, T, 256, 0, true>), wgrid, dim3(512), ws2_lds(a, gw), s, a, gw, ilog2(gw.TW), ilog2(gw.TH), ntiles);
            } else if (ws == 4)
                hipLaunchKernelGGL((conv3x3_ws2_kernel<7, T, 128>), wgrid, dim3(512), ws2_lds(a, gw), s, a, gw, ilog2(gw.TW), ilog2(gw.TH), ntiles);
            else if (e16)
                hipLaunchKernelGGL((conv3x3_ws2_kernel<11, T, 256, 0, false, true>), wgrid, dim3(512), ws2_lds(a, gw), s, a, gw, ilog2(gw.TW), ilog2(gw.TH), ntiles);
            else
                hipLaunchKernelGGL((conv3x3_ws2_kernel<11, T, 256>), wgrid, dim3(512), ws2_lds(a, gw), s, a, gw, ilog2(gw.TW), ilog2(gw.TH), ntiles);
            DMME_CHECK_LAUNCH();
            return DMME_OK;
        }
    }
    DMME_REQUIRE(!a.r_w, DMME_ERR_UNSUPPORTED, "conv3x3 with a residual segment: only the wave-specialised kernel takes it");
    if constexpr (sizeof(T) == 2) {
        ConvTile gk{};
        int kni = 0, kring = 0, kbm = 0;
        if (kw_takes(dtype_of<T>::value, a, pick, gk, &kni, &kring, &kbm)) return launch_kw_t<T>(a, gk, kni, kring, kbm, s);
    }
    size_t lds = pipe_lds(g, kPipeCand[pick][1], kPipeCand[pick][2]);
    ConvArgs ad = a;
    ad.dma_b = (sizeof(T) == 2 && !ACC3 && pipe_dma_ok(dtype_of<T>::value, g, kPipeCand[pick][1], kPipeCand[pick][2])) ? 1 : 0;
    if (ad.dma_b) lds += (size_t)kPipeCand[pick][2] * kPipeCand[pick][1] * ROW_DATA;
    if (a.has_gni) lds += (size_t)2 * (a.C1 + a.C2) * 4;  // scale / shift rows behind the operand buffers
    if (a.n_gno && lds < (size_t)kDirectLds) lds = kDirectLds;
    const int shTW = ilog2(g.TW), shTH = ilog2(g.TH);
    static bool attr_done[5] = {false, false, false, false, false};
    int rc = DMME_OK;
#define DMME_PIPE_CASE(IDX, BM_, BN_, GT_, UA_, LIM)                                                                          \
    case IDX:                                                                                                                 \
        if (!attr_done[IDX]) {                                                                                                \
            rc = set_lds_limit(conv3x3_pipe_kernel<T, BM_, BN_, GT_, UA_, ACC3>, (LIM) * 1024);                               \
            attr_done[IDX] = rc == DMME_OK;                                                                                   \
        }                                                                                                                     \
        if (rc == DMME_OK) hipLaunchKernelGGL((conv3x3_pipe_kernel<T, BM_, BN_, GT_, UA_, ACC3>), grid, dim3(256), lds, s, ad, g, shTW, shTH, ksplit); \
        break;
    switch (pick) {
        DMME_PIPE_CASE(0, 128, 128, 3, 8, 80)
        DMME_PIPE_CASE(1, 128, 64, 3, 8, 80)
        DMME_PIPE_CASE(2, 64, 64, 3, 8, 80)
        DMME_PIPE_CASE(3, 64, 64, 9, 11, 128)
        DMME_PIPE_CASE(4, 64, 64, 3, 10, 80)
    }
#undef DMME_PIPE_CASE
    if (rc != DMME_OK) return rc;
    DMME_CHECK_LAUNCH();
    if (ksplit > 1) {
        const int64_t total4 = (int64_t)a.N * a.Hout * a.Wout * (a.Cout / 4);
        hipLaunchKernelGGL(conv_splitk_finish_kernel<T>, dim3((unsigned)((total4 + 255) / 256)), dim3(256), 0, s, a, ksplit, total4);
        DMME_CHECK_LAUNCH();
    }
    return DMME_OK;
}

int launch_conv_pipe(int dtype, const ConvArgs& a, hipStream_t s) {
    DMME_REQUIRE(conv_pipe_supported(dtype, a), DMME_ERR_UNSUPPORTED, "conv_pipe: unsupported shape");
    if (a.mix) return launch_ws2_split(a, s);
    if (dtype == DMME_BF16) return launch_pipe_t<bf16>(a, s);
    if (dtype == DMME_F16) return launch_pipe_t<f16>(a, s);
    return a.x3 ? launch_pipe_t<float, true>(a, s) : launch_pipe_t<float>(a, s);
}

bool conv_gn_in_query(int dtype, const ConvArgs& a) {
    if (getenv("DMME_NO_GN_IN") || !is16(dtype)) return false;  // (read per plan build, like DMME_NO_GN_DIRECT: the tests toggle it)
    if (conv_out_thin_supported(dtype, a)) return true;       // the thin output conv keeps its image's rows in LDS anyway
    if (a.mix) return conv_pipe_supported(dtype, a);          // the split-pass kernel fills its rows like the wave-specialised kernel it is
    if (a.taps == 1) return conv1x1_as_supported(dtype, a) || conv1x1_pipe_gn_in_ok(dtype, a);  // the store team / the tiled kernel's preamble
    if (!conv_pipe_supported(dtype, a)) return false;
    ConvTile gw{};
    if (!getenv("DMME_NO_WS") && ws_pick(a, gw)) return true;
    // the K-split kernel (small batches): one image per tile, the wave's own chunk rows in its LDS
    ConvTile g{}, gk{};
    const int pick = pipe_pick(a, g);
    int kni = 0, kring = 0, kbm = 0;
    if (pick < 0) return false;
    if (!debug_route("no_gn_in_kw") && kw_takes(dtype, a, pick, gk, &kni, &kring, &kbm)) return gk.TN == 1 && kw_ksplit(a, gk) == 1;
    if (kw_takes(dtype, a, pick, gk, &kni, &kring, &kbm)) return false;
    // the four-wave kernel: rows in LDS behind its operand buffers (two workgroups per CU: the 80 KB budget must still hold)
    if (debug_route("no_gn_in_pipe") || g.TN != 1) return false;
    ConvArgs b = a;
    b.has_gni = 1;
    b.splitk = nullptr;
    size_t lds = pipe_lds(g, kPipeCand[pick][1], kPipeCand[pick][2]);
    if (pipe_dma_ok(dtype, g, kPipeCand[pick][1], kPipeCand[pick][2])) lds += (size_t)kPipeCand[pick][2] * kPipeCand[pick][1] * ROW_DATA;
    lds += (size_t)2 * (a.C1 + a.C2) * 4;
    return lds <= (size_t)(kPipeCand[pick][2] == 9 ? 128 : 80) * 1024;
}

bool conv_gn_direct_query(int dtype, const ConvArgs& a, const int* cg, int n) {
    const bool off = getenv("DMME_NO_GN_DIRECT") != nullptr;
    if (off || a.mix || n < 1 || n > 2 || a.taps != 9 || !conv_pipe_supported(dtype, a)) return false;
    const int VEC = is16(dtype) ? 8 : 4;
    const int HW = a.Hout * a.Wout;
    if (a.out_silu || a.out_nchw || a.res2 || a.Cout % VEC || HW > 64 || (HW & (HW - 1))) return false;
    if (is16(dtype) && !getenv("DMME_NO_WS")) {
        ConvTile gw{};
        if (ws_pick(a, gw)) return false;  // (conv_gn_direct_ws_query answers for that kernel)
    }
    ConvArgs b = a;  // would the split-K heuristic (few workgroups) take this conv?  then it keeps that path
    b.n_gno = 0;
    b.gn_part = nullptr;
    b.splitk = (float*)4096;
    b.splitk_cap = (int64_t)1 << 40;
    ConvTile g{};
    const int pick = pipe_pick(a, g);
    if (pick < 0) return false;
    int BN = 0, BM = 64;
    ConvTile gk{};
    int kni = 0, kring = 0, kbm = 0;
    if (kw_takes(dtype, a, pick, gk, &kni, &kring, &kbm)) {
        if (a.up || kw_ksplit(b, gk) != 1) return false;
        g = gk;
        BN = 32 * kni;
        BM = kbm;
    } else {
        if (kPipeCand[pick][0] != 64 || pipe_ksplit(b, g, pick, is16(dtype) ? 64 : 32) != 1) return false;
        BN = kPipeCand[pick][1];
    }
    if (g.TH != a.Hout || g.TW != a.Wout || g.TN * g.TH * g.TW != BM || a.Cout % BN) return false;  // whole images, whole cout tiles
    if (HW < 64 / (BN / VEC)) return false;  // a wave's pixels per channel vector must not straddle images
    for (int k = 0; k < n; ++k)
        if (cg[k] % VEC || BN % cg[k]) return false;
    return true;
}

// the wave-specialised kernel: its 256-pixel tile is a whole 16x16 image, stored in two passes whose statistics it merges itself
// (scale / shift / {mean, rstd} only - the first pass is in memory before the statistics exist, so no pre-activated output)
bool conv_gn_direct_ws_query(int dtype, const ConvArgs& a, const int* cg, int n) {
    const bool off = getenv("DMME_NO_GN_DIRECT") != nullptr || (debug_route("no_gn_direct_ws") != 0);
    if (off || a.mix || !is16(dtype) || getenv("DMME_NO_WS") || n < 1 || n > 2 || !conv_pipe_supported(dtype, a)) return false;
    ConvTile gw{};
    if (!ws_pick(a, gw) || gw.TH != a.Hout || gw.TW != a.Wout || gw.TH * gw.TW != 256) return false;
    const int cgs = a.gn_cg;  // this tensor's own group size
    if (cgs < 8 || cgs % 8 || 128 % cgs || !stats_tile_ok(a, gw, 128, cgs, 8)) return false;
    for (int k = 0; k < n; ++k) {
        const int f = cg[k] / cgs;
        if (cg[k] % cgs || (f != 1 && f != 2 && f != 4) || 128 % cg[k]) return false;
    }
    return true;
}

bool conv_pipe_stats_query(int dtype, const ConvArgs& a, int cg, int* tiles, int* px) {
    if (a.mix) {  // fp32 output vectors of 4 channels, one partial per 128-pixel epilogue pass
        ConvTile gs{};
        if (!ws2s_pick(a, gs) || !stats_tile_ok(a, gs, 128, cg, 4)) return false;
        *tiles = gs.tiles_x * gs.tiles_y * 2;
        *px = 128;
        return true;
    }
    if (is16(dtype) && !getenv("DMME_NO_WS")) {  // the wave-specialised kernel's tile, when it will run this conv
        ConvTile gw{};
        const int ws = ws_pick(a, gw);
        if (ws) {
            if (!stats_tile_ok(a, gw, 128, cg, 8)) return false;
            *tiles = gw.tiles_x * gw.tiles_y * (ws == 4 ? 1 : 2);  // one partial per 128-pixel epilogue pass
            *px = 128;
            return true;
        }
    }
    ConvTile g{};
    const int pick = pipe_pick(a, g);
    if (pick < 0) return false;
    {
        ConvTile gk{};
        int kni = 0, kring = 0, kbm = 0;
        if (kw_takes(dtype, a, pick, gk, &kni, &kring, &kbm)) {
            if (!stats_tile_ok(a, gk, 32 * kni, cg, 8)) return false;
            *tiles = gk.tiles_x * gk.tiles_y;
            *px = kbm;
            return true;
        }
    }
    if (!stats_tile_ok(a, g, kPipeCand[pick][1], cg, is16(dtype) ? 8 : 4)) return false;
    *tiles = g.tiles_x * g.tiles_y;
    *px = kPipeCand[pick][0];
    return true;
}

void conv_pipe_label(int dtype, const ConvArgs& a, char* buf, int cap) {
    if (a.mix) {
        snprintf(buf, (size_t)cap, a.mix == 2 ? "conv3x3_ws2_kernel<11,f16x3,src16>" : "conv3x3_ws2_kernel<11,f16x3>");
        return;
    }
    if (is16(dtype) && !getenv("DMME_NO_WS")) {
        ConvTile gw{};
        const int ws = ws_pick(a, gw);
        if (ws) {
            snprintf(buf, (size_t)cap, ws == 4 ? (a.r_w ? "conv3x3_ws2_kernel<7,128,res>" : "conv3x3_ws2_kernel<7,128>") : a.r_w ? "conv3x3_ws2_kernel<11,res>" : "conv3x3_ws2_kernel<11>");
            return;
        }
    }
    ConvTile g{};
    const int pick = pipe_pick(a, g);
    {
        ConvTile gk{};
        int kni = 0, kring = 0, kbm = 0;
        if (kw_takes(dtype, a, pick, gk, &kni, &kring, &kbm)) {
            snprintf(buf, (size_t)cap, "conv3x3_kw_kernel<%d,%d,%d>", kni, kring, kbm);
            return;
        }
    }
    snprintf(buf, (size_t)cap, "conv3x3_pipe_kernel<%s,%d,%d,%d,%d>", dtype == DMME_BF16 ? "bf16" : dtype == DMME_F16 ? "f16" : a.x3 ? "float:bf16x3" : "float",
             pick >= 0 ? kPipeCand[pick][0] : 0, pick >= 0 ? kPipeCand[pick][1] : 0, pick >= 0 ? kPipeCand[pick][2] : 0,
             pick >= 0 ? kPipeUA[pick] : 0);
}

}  // namespace dmme
