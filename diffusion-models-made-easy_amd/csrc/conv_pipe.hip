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


__device__ __forceinline__ int swz_off(int row, int chunk) { return row * ROW_DATA + ((chunk ^ ((row >> 1) & 7)) << 4); }

// GT: taps staged per barrier interval (3 = one kernel row, 9 = the whole 3x3 filter: fewer, longer intervals for
//     layers whose per-interval matrix work is too short to hide a global-load round trip);
// UA: halo 16-byte units a thread may own (a_rows * 8 <= 256 * UA).
template <typename T, int BM, int BN, int GT, int PIPE_UA>
__global__ void __launch_bounds__(256, GT == 9 ? 1 : 2) conv3x3_pipe_kernel(ConvArgs a, ConvTile g, int shTW, int shTH) {
    constexpr int KC = Frag<T>::KC, EPV = Frag<T>::EPV;
    constexpr int MI = BM / 64, NI = BN / 64;
    constexpr int UB = BN / 32;  // filter units per thread per tap
    extern __shared__ __attribute__((aligned(16))) char lds[];
    char* ldsA = lds;
    char* ldsB = lds + (size_t)g.a_rows * ROW_DATA;

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
    int b_off[UB];  // element offset of (cout row, tap 0, cin 0) + this thread's 16-byte chunk; -1: past Cout
#pragma unroll
    for (int k = 0; k < UB; ++k) {
        const int co = co0 + urow + 32 * k;
        b_off[k] = co < a.Cout ? co * 9 * Cin + cu * EPV : -1;
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
    uint4 breg[GT][UB];
    const T* wbase = (const T*)a.w;

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
                val = prologue_vec<T>(areg[i], a.scale ? a.scale + so : nullptr, a.scale ? a.shift + so : nullptr,
                                      a.dmask ? a.dmask + so : nullptr, a.pro_silu);
            }
            *reinterpret_cast<uint4*>(ldsA + swz_off(urow + 32 * i, cu)) = val;
        }
    };
    auto load_B = [&](int c0, int grp) {
#pragma unroll
        for (int j = 0; j < GT; ++j)
#pragma unroll
            for (int k = 0; k < UB; ++k) {
                breg[j][k] = make_uint4(0u, 0u, 0u, 0u);
                if (b_off[k] >= 0) breg[j][k] = *reinterpret_cast<const uint4*>(wbase + b_off[k] + (grp * GT + j) * Cin + c0);
            }
    };
    auto store_B = [&]() {
#pragma unroll
        for (int j = 0; j < GT; ++j)
#pragma unroll
            for (int k = 0; k < UB; ++k)
                *reinterpret_cast<uint4*>(ldsB + j * BN * ROW_DATA + swz_off(urow + 32 * k, cu)) = breg[j][k];
    };

    // ---- prologue: first halo chunk + first filter row ----
    load_A(0);
    load_B(0, 0);
    store_A(0);
    store_B();
    __syncthreads();

    const int nchunks = Cin / KC;
    constexpr int NG = 9 / GT;  // groups per Cin chunk
#pragma unroll 1
    for (int ch = 0; ch < nchunks; ++ch) {
#pragma unroll 1
        for (int grp = 0; grp < NG; ++grp) {
            // issue the next group's loads before the matrix work
            const bool last_grp = grp == NG - 1;
            const bool more = !(last_grp && ch == nchunks - 1);
            const int nc0 = (last_grp ? ch + 1 : ch) * KC;
            if (more) {
                load_B(nc0, last_grp ? 0 : grp + 1);
                if (last_grp) load_A(nc0);
            }
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
                        bfr[ni] = *reinterpret_cast<const uint4*>(ldsB + j * BN * ROW_DATA + b_base[ni] + ((cidx ^ b_swz[ni]) << 4));
#pragma unroll
                    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                        for (int ni = 0; ni < NI; ++ni) mma_group(af[mi], bfr[ni], acc[mi][ni], (T*)nullptr);
                }
            }
            __syncthreads();  // every wave is done reading this group's tiles
            if (more) {
                store_B();
                if (last_grp) store_A(nc0);
            }
            __syncthreads();
        }
    }

    // ---- epilogue: + bias + time embedding + residual, store ----
    auto pix_of = [&](int m) -> int {
        const int tx = m & mTW, ty = (m >> shTW) & mTH, tn = m >> (shTW + shTH);
        const int n = n0 + tn;
        return n < a.N ? (n * a.Hout + oy0 + ty) * a.Wout + ox0 + tx : -1;
    };
    conv_epilogue<T, BM, BN, MI, NI>(a, acc, co0, wn0, r, h, wm0, n0, g.TN, pix_of, reinterpret_cast<float*>(lds), ty_blk * g.tiles_x + tx_blk);
}

// candidate kernels: {BM, BN, GT}; the 9-tap variant serves layers with too little work per interval
// (few workgroups or stride 2) and owns a larger LDS footprint
static const int kPipeCand[4][3] = {{128, 128, 3}, {128, 64, 3}, {64, 64, 3}, {64, 64, 9}};
static const int kPipeUA[4] = {8, 8, 8, 10};
static size_t pipe_lds(const ConvTile& g, int BN, int GT) { return (size_t)g.a_rows * ROW_DATA + (size_t)GT * BN * ROW_DATA; }  // >= BM*BN*4 always

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
    if (pick < 0 || (pick == 2 && (int64_t)g.tiles_m * g.tiles_n < 2 * 256)) {
        ConvTile t;
        if (pipe_fits(a, 3, t)) {
            pick = 3;
            g = t;
        }
    }
    return pick;
}

bool conv_pipe_supported(int dtype, const ConvArgs& a) {
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

template <typename T>
static int launch_pipe_t(const ConvArgs& a, hipStream_t s) {
    ConvTile g{};
    const int pick = pipe_pick(a, g);
    DMME_REQUIRE(pick >= 0, DMME_ERR_UNSUPPORTED, "conv_pipe: no tile fits");
    const dim3 grid((unsigned)(g.tiles_m * g.tiles_n));
    const size_t lds = pipe_lds(g, kPipeCand[pick][1], kPipeCand[pick][2]);
    const int shTW = ilog2(g.TW), shTH = ilog2(g.TH);
    static bool attr_done[4] = {false, false, false, false};
    int rc = DMME_OK;
#define DMME_PIPE_CASE(IDX, BM_, BN_, GT_, UA_, LIM)                                                                          \
    case IDX:                                                                                                                 \
        if (!attr_done[IDX]) {                                                                                                \
            rc = set_lds_limit(conv3x3_pipe_kernel<T, BM_, BN_, GT_, UA_>, (LIM) * 1024);                                     \
            attr_done[IDX] = rc == DMME_OK;                                                                                   \
        }                                                                                                                     \
        if (rc == DMME_OK) hipLaunchKernelGGL((conv3x3_pipe_kernel<T, BM_, BN_, GT_, UA_>), grid, dim3(256), lds, s, a, g, shTW, shTH); \
        break;
    switch (pick) {
        DMME_PIPE_CASE(0, 128, 128, 3, 8, 80)
        DMME_PIPE_CASE(1, 128, 64, 3, 8, 80)
        DMME_PIPE_CASE(2, 64, 64, 3, 8, 80)
        DMME_PIPE_CASE(3, 64, 64, 9, 10, 128)
    }
#undef DMME_PIPE_CASE
    if (rc != DMME_OK) return rc;
    DMME_CHECK_LAUNCH();
    return DMME_OK;
}

int launch_conv_pipe(int dtype, const ConvArgs& a, hipStream_t s) {
    DMME_REQUIRE(conv_pipe_supported(dtype, a), DMME_ERR_UNSUPPORTED, "conv_pipe: unsupported shape");
    return dtype == DMME_BF16 ? launch_pipe_t<bf16>(a, s) : launch_pipe_t<float>(a, s);
}

bool conv_pipe_stats_query(int dtype, const ConvArgs& a, int cg, int* tiles, int* px) {
    ConvTile g{};
    const int pick = pipe_pick(a, g);
    if (pick < 0) return false;
    if (!stats_tile_ok(a, g, kPipeCand[pick][1], cg, dtype == DMME_BF16 ? 8 : 4)) return false;
    *tiles = g.tiles_x * g.tiles_y;
    *px = kPipeCand[pick][0];
    return true;
}

void conv_pipe_label(int dtype, const ConvArgs& a, char* buf, int cap) {
    ConvTile g{};
    const int pick = pipe_pick(a, g);
    snprintf(buf, (size_t)cap, "conv3x3_pipe_kernel<%s,%d,%d,%d,%d>", dtype == DMME_BF16 ? "bf16" : "float",
             pick >= 0 ? kPipeCand[pick][0] : 0, pick >= 0 ? kPipeCand[pick][1] : 0, pick >= 0 ? kPipeCand[pick][2] : 0,
             pick >= 0 ? kPipeUA[pick] : 0);
}

}  // namespace dmme
