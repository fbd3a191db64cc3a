// Level engine: ONE persistent launch runs every layer of a small-map resolution level (4x4 / 8x8 maps): the ResBlocks
// (GroupNorm + SiLU + Dropout2d + 3x3 conv + time row + residual, models/ddpm.py:118-133), their residual 1x1 convs (:108-111,131)
// and the single-head attention block of the 4x4 maps (:38-75), in the order UNet.forward walks them (:297-313).
//
// Why: at these resolutions a layer is ~1 us of matrix work (64 pixels x 32 couts x K = 2304 per workgroup: 72 MFMAs per wave) behind
// 10-30 us of fixed cost per launch - dispatch, cold instruction fetch, argument loads, the first round trip of the weight stream
// from beyond L2, a dependent epilogue (14 launches of 19 us on the 4x4 level, 10 of 33 us on the 8x8 level at
// batch 128; ALL of a batch-1 step).  Here the launch boundary between two layers becomes a hand-off between the 8 workgroups that
// share a pixel group, and the next layer's weight stream starts BEFORE the wait for that hand-off.
//
// Decomposition (lvl.h): workgroup (g, s) owns pixel group g (64 consecutive NHWC pixels = whole images) x cout slice s (32 of the
// 256 couts) of EVERY op.  Inside a workgroup the K loop is split over the four waves exactly as in conv_kw.hip (each wave computes the
// whole 64 x 32 tile over its own quarter of the (64-channel chunk, tap) units; private LDS-DMA filter ring, private copy of its
// chunks of the input, no workgroup barrier in the main loop; the four partial tiles are summed through LDS in wave order).
//
// Epilogue = the producer side of every GroupNorm that reads the output (LvlNorm): the slice holds whole groups of whole images, so
// the statistics are complete here; the workgroup stores the raw slice, the norms' scale / shift / {mean, rstd} rows and, per norm,
// the consumer's pre-activated input.  Every element is normalised ONCE, by its producer; a consumer only gathers.
//
// Hand-off (MI355X_MICROARCH.md, "Valid forms", first row of the sc1 table): the producer's stores of handed-off bytes are all
// 16-byte sc1 (write-through) stores; every storing wave waits vmcnt(0); workgroup barrier; ONE lane stores the flag word (sc1).  A
// consumer WAVE polls the 8 flag words of its group with sc1 loads (lanes 0-7, s_sleep between polls, bounded), and only after its own
// poll matched issues its loads of the handed-off bytes - all of them 16-byte sc1 buffer loads to registers.  No fence, no
// dependence on placement: a different workgroup -> XCD mapping changes speed only.  Flags carry the launch's epoch (device-resident
// counter bumped by the last workgroup to finish), so nothing is re-initialised between launches and a captured graph can replay it.
// Every wait is bounded: on a timeout the error word is set and the launch runs to its end (wrong numbers, no hang).
// All workgroups must be co-resident: grid = (groups resident at once) x 8 <= 256 = one per CU (LDS > 80 KB guarantees one per CU).
#include <stdio.h>

#include "conv_common.h"
#include "lvl.h"

namespace dmme {

// A workgroup's cout slice is NJ 32-cout blocks wide (template parameter; lvl.h: 32 couts x 8 slices, or 64 x 4 for large batches).
// Filter unit = (64-channel chunk, tap) of the slice: NJ x 4 KB, NJ x 4 DMA wave-instructions.
// One workgroup handles GB pixel groups per op iteration (GB = 2 when a workgroup owns several groups: the filter stream is then shared
// by twice the matrix work and every fixed cost of an iteration is paid half as often).  Per wave: the A image - GB groups x ONE
// 64-channel chunk (the K loop runs in passes of 256 channels, wave w takes chunk 4 p + w of pass p) - and the filter ring.
// A image: rows of 128 B on a 144-byte pitch (conflict-free ds_read_b128 WITHOUT a swizzle: a fragment address is one per-lane base
// per (tap, pixel block) computed once per launch, the k-group and the pixel group are instruction offsets), 64 pixel rows + one row
// of zeros per group (out-of-image taps read it).  GB = 1: 9.3 KB + 6 ring slots, GB = 2: 18.4 KB + 4 slots.
constexpr int LVL_PITCH = ROW_DATA + 16;
constexpr int LVL_GS = (LVL_BM + 1) * LVL_PITCH;                       // bytes per pixel group of the A image (9360)
constexpr int lvl_a_bytes(int gb) { return (gb * LVL_GS + 127) / 128 * 128; }
constexpr int LVL_WAVE_BYTES = lvl_a_bytes(2) + 16 * 1024;            // 35200: 16 KB of ring (GB = 1: + the 9 KB its A image leaves free)
constexpr int LVL_KEEP_OFF = 4 * LVL_WAVE_BYTES;                       // q / k / v slices of the attention block: [3][64 px][32 ch] T
constexpr int LVL_KEEP_BYTES = 3 * LVL_BM * LVL_BN * 2;
constexpr int LVL_BLK_OFF = LVL_KEEP_OFF + LVL_KEEP_BYTES;             // statistics exchange: [items = GB x NJ][2 norms][4 vectors][4 pixel blocks][2]
constexpr int LVL_LDS = LVL_BLK_OFF + 1024;
constexpr int LVL_SPIN_LIMIT = 1 << 19;                                // polls before a wait gives up (~a second)
static_assert(LVL_LDS <= 160 * 1024, "level engine: LDS budget");
static_assert(lvl_a_bytes(1) + 6 * 4096 <= LVL_WAVE_BYTES, "level engine: GB = 1 layout");

size_t lvl_engine_lds_bytes() { return LVL_LDS; }

typedef unsigned u32x4_lv __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) u32x4_lv lds_u32x4_lv;

// 16-byte sc1 (system-coherent level 1: L1-bypassing load, write-through store) accesses through a buffer resource whose base is
// wave-uniform; the byte offset is per lane.  Device pass only (the host pass of hipcc parses kernel bodies too).
#if defined(__HIP_DEVICE_COMPILE__)
typedef __amdgpu_buffer_rsrc_t lvl_rsrc;
__device__ __forceinline__ lvl_rsrc lvl_make_rsrc(const void* base, unsigned bytes) {
    // the descriptor's inputs made PROVABLY wave-uniform (they are: kernel arguments and op fields): otherwise hipcc wraps every buffer
    // access in a waterfall loop (cdna_hip_programming.md T20)
    const uint64_t b = (uint64_t)base;
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)b), hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(b >> 32));
    return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((uint64_t)hi << 32) | lo), 0, __builtin_amdgcn_readfirstlane((int)bytes), 0x00020000);
}
__device__ __forceinline__ uint4 lvl_ld(lvl_rsrc r, unsigned off) {
    return __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)off, 0, 16));
}
__device__ __forceinline__ void lvl_st(lvl_rsrc r, unsigned off, const uint4& v) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_lv, v), r, (int)off, 0, 16);
}
#else
typedef int lvl_rsrc;
__device__ __forceinline__ lvl_rsrc lvl_make_rsrc(const void*, unsigned) { return 0; }
__device__ __forceinline__ uint4 lvl_ld(lvl_rsrc, unsigned) { return make_uint4(0u, 0u, 0u, 0u); }
__device__ __forceinline__ void lvl_st(lvl_rsrc, unsigned, const uint4&) {}
#endif

__device__ __forceinline__ unsigned lvl_flag_load(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void lvl_flag_store(unsigned* p, unsigned v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// This WAVE waits until the `ns` flag words of row `f` (one per cout slice) carry this launch's epoch.  Lanes 0-7 poll (sc1 loads), the vote is
// wave-wide; bounded: after `limit` polls (LvlArgs::spin_limit, ~a second by default) the wave gives up, sets the launch's error word
// (ctl[2]: every other wait of the launch ends within 1024 polls) AND the plan's host-visible status word (LvlArgs::err_sys, a
// system-scope store into pinned host memory): the launch drains with wrong numbers instead of hanging, and the host side refuses
// to hand those numbers on (plan.hip: lvl_check - every entry point and dmme_unet_plan_check read that word).
struct LvlWaitCtx {
    unsigned* err;      // ctl[2]
    unsigned* err_sys;  // host-visible (nullable)
    int limit;
    unsigned tag;       // what the host reads: 1 + index of the run
};
__device__ __forceinline__ void lvl_wait_row(const unsigned* f, int ns, unsigned epoch, int lane, const LvlWaitCtx& c) {
    for (int spin = 0;; ++spin) {
        const unsigned v = lane < ns ? lvl_flag_load(f + lane) : epoch;
        if (__all(v == epoch)) return;
        if ((spin & 1023) == 1023) {  // a wait that timed out anywhere ends every other wait too: the launch drains in milliseconds
            const unsigned e = lvl_flag_load(c.err);
            if (e != 0u) return;
            if (spin >= c.limit) {
                if (lane == 0) {
                    lvl_flag_store(c.err, 1u);
                    if (c.err_sys) __hip_atomic_store(c.err_sys, c.tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                }
                return;
            }
        }
        __builtin_amdgcn_s_sleep(2);
    }
}
// (callers put a compiler barrier behind the wait: nothing orders later loads behind a relaxed atomic load for the compiler)
__device__ __forceinline__ void lvl_compiler_fence() { asm volatile("" ::: "memory"); }

// (mean, M2) of equal-count sets merged over the 16 pixels a wave holds of one channel vector (lanes 4 apart): DPP row rotations
// by 4 and by 8 (each lane meets the lanes that share lane & 3 in its 16-lane row, in disjoint pairs), then the row and half-wave
// swaps.  Chan's formula for equal counts: mean = (a + b) / 2, M2 = M2a + M2b + (a - b)^2 n / 2.  (ds_bpermute shuffles cost an LDS
// round trip per step: eight dependent ones per epilogue.)
__device__ __forceinline__ void lvl_merge_pair(float& mean, float& m2, float om, float o2, float cnt) {
    const float d = om - mean;
    m2 += o2 + d * d * (0.5f * cnt);
    mean = 0.5f * (mean + om);
}
__device__ __forceinline__ void lvl_merge16(float& mean, float& m2) {
    lvl_merge_pair(mean, m2, DMME_DPP_F(mean, 0x124), DMME_DPP_F(m2, 0x124), 8.f);   // row_ror:4
    lvl_merge_pair(mean, m2, DMME_DPP_F(mean, 0x128), DMME_DPP_F(m2, 0x128), 16.f);  // row_ror:8
    {
        float a = mean, b = mean, a2 = m2, b2 = m2;
        permlane16_swap(a, b);  // a: the even row's value in both rows of a pair, b: the odd row's
        permlane16_swap(a2, b2);
        mean = a;
        m2 = a2;
        lvl_merge_pair(mean, m2, b, b2, 32.f);
    }
    {
        float a = mean, b = mean, a2 = m2, b2 = m2;
        permlane32_swap(a, b);  // a: the lower half-wave's value everywhere, b: the upper's
        permlane32_swap(a2, b2);
        mean = a;
        m2 = a2;
        lvl_merge_pair(mean, m2, b, b2, 64.f);
    }
}

template <typename T, int GB, int NJ>
__global__ void __launch_bounds__(256, 1) lvl_engine_kernel(LvlArgs A, const LvlOp* __restrict__ ops) {
    constexpr int BN = 32 * NJ, NS = LVL_NS / NJ;  // couts per slice, slices per pixel group
    constexpr int NPI = BN / 8;                    // DMA wave-instructions per filter unit
    constexpr int U_BYTES = BN * ROW_DATA;
    constexpr int RING = GB == 1 ? 6 : NJ == 2 ? 2 : 4;   // filter units per wave ring (GB = 1: 24 KB, GB = 2: 16 KB)
    constexpr int D = RING - 1;                    // units requested ahead of the one being consumed
    constexpr int MI = 2 * GB;                     // 32-pixel row blocks per iteration
    constexpr int GQ = GB * NJ;                    // 64-pixel x 32-cout items per thread and iteration
    constexpr int A_BYTES = lvl_a_bytes(GB);
    static_assert(D * NPI < 64 && D >= 1, "vmcnt is a 6-bit counter");
    static_assert(NJ == 1 || GB == 2, "64-cout slices exist for two-group iterations only (the cross-wave sum uses the whole wave region)");
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    // Workgroups are dealt round-robin over the 8 XCDs, so the NS slices of a pixel group - neighbours in index - sat on NS different
    // XCDs and every one of them pulled the group's whole input across the fabric (the producers' write-through stores leave nothing in
    // L2): 3.2x / 4.3x the algorithmic bytes in the FETCH / WRITE counters (round 4).  Inside windows of 64 workgroups the index is
    // permuted so that a group's slices are the workgroups of ONE XCD (blockIdx % 8 equal): the first slice's gather fills that XCD's L2,
    // the others read it there.  Placement is a speed matter only - the hand-off protocol assumes nothing about it - and a group
    // still lies inside one window of the dispatch order, so a partially resident launch still consists of whole windows plus one.
    int vb = (int)blockIdx.x;
    if (A.xcd_group && (gridDim.x & 63u) == 0u) vb = (vb & ~63) + (vb & 7) * 8 + ((vb & 63) >> 3);
    const int s = vb % NS, b0 = vb / NS;
    char* ldsA = lds + wave * LVL_WAVE_BYTES;
    char* ldsR = ldsA + A_BYTES;
    T* keep = reinterpret_cast<T*>(lds + LVL_KEEP_OFF);
    float* blk = reinterpret_cast<float*>(lds + LVL_BLK_OFF);
    const unsigned epoch = lvl_flag_load(&A.ctl[0]) + 1u;  // (the counter moves only after EVERY workgroup of a launch has finished)
    const LvlWaitCtx err{&A.ctl[2], A.err_sys, A.spin_limit > 0 ? A.spin_limit : LVL_SPIN_LIMIT, (unsigned)A.run_tag};
    // debug knob (tests/test_gpu_level.py): workgroup 0 stops signalling - its consumers time out and the error must reach the caller
    const bool withhold = A.withhold > 0 && epoch >= (unsigned)A.withhold && blockIdx.x == 0;  // (from the launch with this epoch on)
    {   // the op table is read through the scalar cache, one op at a time, and every first touch of a line is a round trip to L2 or
        // beyond on the critical path of its op: touch every line now (independent scalar loads, one latency for all of them)
        const int* w = reinterpret_cast<const int*>(ops);
        const int n_lines = (A.n_ops * (int)sizeof(LvlOp) + 63) / 64;
        int sink = 0;
        for (int i = 0; i < n_lines; ++i) sink += w[i * 16];
        asm volatile("" ::"s"(sink));
    }
    const int sh = A.sh, sh2 = 2 * sh, HW = 1 << sh2, mW = (1 << sh) - 1;
    const int npix = A.N * HW;
    if (lane < GB * 9) *reinterpret_cast<uint4*>(ldsA + (lane / 9) * LVL_GS + LVL_BM * LVL_PITCH + (lane % 9) * 16) = make_uint4(0u, 0u, 0u, 0u);  // the rows of zeros

    // ---- per-lane fragment geometry (fixed for the launch): MFMA row r of pixel block mi is pixel mi * 32 + r of the iteration ----
    // ab[t][mi]: LDS address of this lane's 16 bytes of (tap t, pixel block mi of group 0, k-group 0); the same for every op and every
    // group (groups are whole images): pixel m + tap offset inside the image, else the group's row of zeros
    unsigned ab[9][2];
    {
        const unsigned a0 = (unsigned)(size_t)(lds_c*)ldsA + (unsigned)h * 16u;
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
            const int m = mi * 32 + r, tx = m & mW, ty = (m >> sh) & mW;
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int yy = ty + t / 3 - 1, xx = tx + t % 3 - 1;
                const bool in = yy >= 0 && yy <= mW && xx >= 0 && xx <= mW;
                ab[t][mi] = a0 + (unsigned)((in ? m + ((t / 3 - 1) << sh) + (t % 3 - 1) : LVL_BM) * LVL_PITCH);
            }
        }
    }
    int tb[4];  // filter fragment offsets inside a ring slot (row r; the XOR swizzle of conv_common.h)
#pragma unroll
    for (int kg = 0; kg < 4; ++kg) tb[kg] = (r * ROW_DATA + ((h ^ ((r >> 1) & 7)) << 4)) ^ (kg << 5);
    const unsigned ring_base = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(size_t)(lds_c*)ldsR);

    // ---- filter stream state of this wave (conv_kw.hip's ring discipline; the stream of an op starts before the op does) ----
    // (Loading an op's first units straight into registers when its stream starts - so that a 256-channel 3x3 conv has all nine units
    // requested before it begins - was measured and removed: hipcc copies loop-carried registers at the back edge and waits for the
    // loads where they are issued; B = 1, 4x4 run: 152 -> 198 us.)
    unsigned boff[NPI], boff2[NPI];  // (boff2, dptr2, d_cin2b: the op's second segment, see LvlOp::C3)
    const char* dptr2 = nullptr;
    int d_cin2b = 0, d_seg = 0;      // d_seg: units of the current segment not yet requested
    const char* dptr = nullptr;
    unsigned dslot = ring_base;
    int dtap = 0, d_taps = 9, d_cin2 = 0, d_left = 0;  // d_left: ring units of the current stream not yet requested
    int d_req = 0;                                       // ring units requested so far
    bool primed = false;
    auto dma_next = [&]() __attribute__((always_inline)) {
        // (the asm's "s" operands must be SGPRs: under register pressure hipcc has handed it VGPR copies of these uniform values -
        // "invalid operand" at assembly time - so they are pinned here; a readfirstlane of a value already in an SGPR folds away)
        const uint64_t dp_ = (uint64_t)dptr;
        const char* dps = reinterpret_cast<const char*>(((uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(dp_ >> 32)) << 32) |
                                                        (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)dp_));
        const unsigned dss = (unsigned)__builtin_amdgcn_readfirstlane((int)dslot);
#pragma unroll
        for (int i = 0; i < NPI; ++i) glds16_hidden_s(dps, boff[i], dss + (unsigned)(i * 8 * ROW_DATA));
        dptr += d_cin2;
        if (++dtap == d_taps) {  // next pass: the wave's chunk moves on by 4 chunks (256 channels), tap 0
            dtap = 0;
            dptr += 512 - d_taps * d_cin2;
        }
        dslot = dslot + U_BYTES == ring_base + RING * U_BYTES ? ring_base : dslot + U_BYTES;
        --d_left;
        ++d_req;
        if (--d_seg == 0 && d_left > 0) {  // the first segment is fully requested: the 1x1 segment follows in the same ring
#pragma unroll
            for (int i = 0; i < NPI; ++i) boff[i] = boff2[i];
            dptr = dptr2;
            d_taps = 1;
            d_cin2 = d_cin2b;
            dtap = 0;
        }
    };
    // start the filter stream of conv op `o` (pass-major, tap-minor; wave w owns chunk 4 p + w): the first D units into the ring
    auto prime = [&](const LvlOp& o) __attribute__((always_inline)) {
        const int Cin = o.C1 + o.C2, Cinb = o.C3 + o.C4, nu1 = (Cin >> 8) * o.taps, nu = nu1 + (Cinb >> 8);
#pragma unroll
        for (int i = 0; i < NPI; ++i) {
            const int row = 8 * i + (lane >> 3);
            boff[i] = (unsigned)((o.w_row0 + BN * s + row) * o.taps * Cin + ((lane & 7) ^ ((row >> 1) & 7)) * 8) * 2u;
            boff2[i] = (unsigned)((BN * s + row) * Cinb + ((lane & 7) ^ ((row >> 1) & 7)) * 8) * 2u;
        }
        dptr2 = A.packed + o.w2_off + wave * 128;
        d_cin2b = Cinb * 2;
        d_seg = nu1;
        d_taps = o.taps;
        d_cin2 = Cin * 2;
        dtap = 0;
        dptr = A.packed + o.w_off + wave * 128;
        dslot = ring_base;
        d_left = nu;
        d_req = 0;
#pragma unroll
        for (int d = 0; d < D; ++d)
            if (d_left > 0) dma_next();
        primed = true;
    };

    int stamp_it = 0;
#define LV_STAMP(K) do { if (A.stamps && (int)blockIdx.x == A.stamp_wg && tid == 0 && stamp_it < 120) A.stamps[stamp_it * 8 + (K)] = (long long)wall_clock64(); } while (0)
    // ------------------------------------------------------------------------------------------------------------------------
    const int NB = (A.NG + GB - 1) / GB;  // iterations per op over all workgroups: batches of GB groups
    for (int oi = 0; oi < A.n_ops; ++oi) {
        const LvlOp& op = ops[oi];
        for (int bt = b0; bt < NB; bt += A.NGS) {
            // this thread's items: pixel m of each of the GB groups x 8 channels (vector vec) of each of the slice's NJ 32-cout blocks;
            // item q = j * NJ + nj
            const int m = tid >> 2, vec = tid & 3;
            const int co0 = BN * s + vec * 8;       // first of the thread's 8 channels among the op's 256 (block nj: + 32 nj)
            int gpv[GB];                            // pixel index in the NHWC tensors per group (-1: past the batch)
#pragma unroll
            for (int j = 0; j < GB; ++j) {
                const int g = bt * GB + j, gp = g * LVL_BM + m;
                gpv[j] = (g < A.NG && gp < npix) ? gp : -1;
            }
            uint4 ovec[GQ];  // the items' values as stored (T)
            bool have_out = false;
            LV_STAMP(0);
            // gamma / beta of the norms this op finishes depend on nothing: requested early, used last (32-cout slices: before the main
            // loop; 64-cout slices: behind it, under the cross-wave sum - the main loop has no registers to spare)
            f32x4 gm[2][NJ][2], bt4[2][NJ][2];
#define LV_LOAD_GAMMA_BETA()                                                                                                        \
    do {                                                                                                                           \
        _Pragma("unroll") for (int kx = 0; kx < 2; ++kx) {                                                                         \
            if (kx >= op.n_norm) break;                                                                                            \
            const LvlNorm& G = op.norm[kx];                                                                                        \
            _Pragma("unroll") for (int nj = 0; nj < NJ; ++nj) {                                                                    \
                const float* gp_ = reinterpret_cast<const float*>(A.packed + G.gamma_off) + G.c_off + co0 + 32 * nj;               \
                const float* bp_ = reinterpret_cast<const float*>(A.packed + G.beta_off) + G.c_off + co0 + 32 * nj;                \
                gm[kx][nj][0] = *reinterpret_cast<const f32x4*>(gp_);                                                              \
                gm[kx][nj][1] = *reinterpret_cast<const f32x4*>(gp_ + 4);                                                          \
                bt4[kx][nj][0] = *reinterpret_cast<const f32x4*>(bp_);                                                             \
                bt4[kx][nj][1] = *reinterpret_cast<const f32x4*>(bp_ + 4);                                                         \
            }                                                                                                                      \
        }                                                                                                                          \
    } while (0)
            if (op.kind == LVL_NORM || (op.kind == LVL_CONV && NJ == 1)) LV_LOAD_GAMMA_BETA();

            if (op.kind == LVL_CONV) {
                const int Cin = op.C1 + op.C2, taps = op.taps, npass = Cin >> 8;
                if (!primed) prime(op);
                // what the items need besides the sums is requested NOW (residual, bias, time row: a round trip of 1-2 us that the main loop hides; 64-cout slices: behind the main loop,
                // like gamma / beta)
                uint4 resv[GQ];
                float fold[GQ][8];
#define LV_LOAD_FOLD()                                                                                                             \
    do {                                                                                                                           \
_Pragma("unroll") \
                for (int q = 0; q < GQ; ++q) { \
                    const int j = q / NJ, co = co0 + 32 * (q % NJ); \
                    const int gpc = gpv[j] < 0 ? 0 : gpv[j]; \
                    const float* bp = reinterpret_cast<const float*>(A.packed + op.b_off) + op.w_row0 + co; \
                    f32x4 b0v = *reinterpret_cast<const f32x4*>(bp), b1v = *reinterpret_cast<const f32x4*>(bp + 4); \
                    if (op.C3 > 0) { /* + the residual conv's bias */ \
                        const float* bq = reinterpret_cast<const float*>(A.packed + op.b2_off) + co; \
                        b0v += *reinterpret_cast<const f32x4*>(bq); \
                        b1v += *reinterpret_cast<const f32x4*>(bq + 4); \
                    } \
                    resv[q] = make_uint4(0u, 0u, 0u, 0u); \
                    if (op.res_off >= 0) { \
                        const lvl_rsrc rr = lvl_make_rsrc(A.ws + op.res_off, (unsigned)npix * (unsigned)op.res_C * 2u); \
                        resv[q] = lvl_ld(rr, (unsigned)(gpc * op.res_C + op.res_c0 + co) * 2u); \
                    } \
_Pragma("unroll") \
                    for (int e = 0; e < 4; ++e) { \
                        fold[q][e] = b0v[e]; \
                        fold[q][4 + e] = b1v[e]; \
                    } \
                    if (op.tproj_col >= 0) { \
                        const float* tp = A.tproj + (int64_t)(A.nt == 1 ? 0 : gpc >> sh2) * A.tproj_ld + op.tproj_col + co; \
                        const f32x4 t0 = *reinterpret_cast<const f32x4*>(tp), t1 = *reinterpret_cast<const f32x4*>(tp + 4); \
_Pragma("unroll") \
                        for (int e = 0; e < 4; ++e) { \
                            fold[q][e] += t0[e]; \
                            fold[q][4 + e] += t1[e]; \
                        } \
                    } \
                } \
    } while (0)
                if (NJ == 1) LV_LOAD_FOLD();
                f32x16 acc[MI][NJ];
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                    for (int nj = 0; nj < NJ; ++nj)
#pragma unroll
                        for (int j = 0; j < 16; ++j) acc[mi][nj][j] = 0.f;
                // fragment registers: 32-cout slices double-buffer whole units (16 MFMAs); 64-cout slices double-buffer k-groups (8 MFMAs
                // cover the next k-group's six reads) - whole units would be 192 registers beside 128 accumulators
                uint4 af[NJ == 1 ? 2 : 1][NJ == 1 ? 4 : 1][MI], bfr[NJ == 1 ? 2 : 1][NJ == 1 ? 4 : 1][NJ];
                uint4 ag[2][MI], bg[2][NJ];
/* fragments of the unit at tap TAP (a compile-time constant) of the current pass into set SET: the input rows from the A image (one   \
   base per (tap, pixel block); k-group and pixel group are instruction offsets), the filter rows from the ring slot at `rslot`,       \
   which then moves on */                                                                                                            \
#define LV_READ_FRAGS(SET, TAP)                                                                                                    \
    do {                                                                                                                           \
        _Pragma("unroll") for (int kg = 0; kg < 4; ++kg)                                                                           \
            _Pragma("unroll") for (int mi = 0; mi < MI; ++mi)                                                                      \
                af[SET][kg][mi] = __builtin_bit_cast(uint4, *reinterpret_cast<const lds_u32x4_lv*>((const lds_c*)(size_t)ab[TAP][mi & 1] + (kg * 32 + (mi >> 1) * LVL_GS))); \
        const lds_c* rb_ = (const lds_c*)(size_t)rslot;                                                                            \
        _Pragma("unroll") for (int kg = 0; kg < 4; ++kg)                                                                           \
            _Pragma("unroll") for (int nj = 0; nj < NJ; ++nj)                                                                      \
                bfr[SET][kg][nj] = __builtin_bit_cast(uint4, *reinterpret_cast<const lds_u32x4_lv*>(rb_ + tb[kg] + nj * 32 * ROW_DATA)); \
        rslot = rslot + U_BYTES == ring_base + RING * U_BYTES ? ring_base : rslot + U_BYTES;                                       \
        ++q_read;                                                                                                                  \
    } while (0)
/* before the fragments of ring unit q_read are read: it has landed when at most the units requested after it are outstanding */  \
#define LV_WAIT_UNIT()                                                                                                            \
    do {                                                                                                                           \
        if (d_req - (q_read + 1) >= D)                                                                                             \
            wait_vm_keep<D * NPI>();                                                                                           \
        else                                                                                                                       \
            wait_vm_keep<0>();                                                                                                     \
    } while (0)
#define LV_MMA(CUR, KG)                                                                            \
    do {                                                                                           \
        _Pragma("unroll") for (int mi = 0; mi < MI; ++mi)                                          \
            _Pragma("unroll") for (int nj = 0; nj < NJ; ++nj) mma16<T>(bfr[CUR][KG][nj], af[CUR][KG][mi], acc[mi][nj]); \
    } while (0)
// one step = the MFMAs of one unit (fragment set CUR); the request for a later unit goes out under k-group 0, the fragments of the
// pass's next unit (tap NTAP, set CUR ^ 1; NTAP < 0: none) are read under k-groups 1-3
#define LV_STEP(CUR, NTAP)                                                                                                         \
    do {                                                                                                                           \
        wait_lgkm_all(); /* set CUR is complete - and its ring slot is free for the unit RING further on */                        \
        __builtin_amdgcn_sched_barrier(0);                                                                                         \
        LV_MMA(CUR, 0);                                                                                                            \
        __builtin_amdgcn_sched_barrier(0);                                                                                         \
        if (d_left > 0) dma_next();                                                                                                \
        __builtin_amdgcn_sched_barrier(0);                                                                                         \
        LV_MMA(CUR, 1);                                                                                                            \
        __builtin_amdgcn_sched_barrier(0);                                                                                         \
        if ((NTAP) >= 0) {                                                                                                         \
            LV_WAIT_UNIT();                                                                                                        \
            LV_READ_FRAGS((CUR) ^ 1, (NTAP) < 0 ? 0 : (NTAP));                                                                     \
        }                                                                                                                          \
        __builtin_amdgcn_sched_barrier(0);                                                                                         \
        LV_MMA(CUR, 2);                                                                                                            \
        LV_MMA(CUR, 3);                                                                                                            \
        __builtin_amdgcn_sched_barrier(0);                                                                                         \
    } while (0)
                unsigned rslot = ring_base;  // LDS address of the slot of the next ring unit whose fragments will be read
                int q_read = 0;              // ring units read so far
                const int npass2 = (op.C3 + op.C4) >> 8;  // passes of the second segment (the block's 1x1 residual conv over its raw input)
                for (int pt = 0; pt < npass + npass2; ++pt) {
                    const bool seg2 = pt >= npass;
                    const int p = seg2 ? pt - npass : pt;
                    const int taps_p = seg2 ? 1 : taps;
                    // ---- A operand of this pass: chunk 4 p + wave of the GB groups' 64 pixels each, gathered after the hand-off ----
                    if (seg2 || !op.reuse_a) {
                        if (p == 0) {
                            const int w0 = seg2 ? op.wait2 : op.wait0, w1 = seg2 ? op.wait3 : op.wait1;
#pragma unroll
                            for (int j = 0; j < GB; ++j) {
                                const int g = bt * GB + j;
                                if (g >= A.NG) break;
                                if (w0 >= 0) lvl_wait_row(A.flags + ((int64_t)w0 * A.NG + g) * LVL_NS, NS, epoch, lane, err);
                                if (w1 >= 0) lvl_wait_row(A.flags + ((int64_t)w1 * A.NG + g) * LVL_NS, NS, epoch, lane, err);
                            }
                            lvl_compiler_fence();
                            if (!seg2) LV_STAMP(1);
                        }
                        const int c = p * 4 + wave;
                        const int Ca = seg2 ? op.C3 : op.C1, Cb = seg2 ? op.C4 : op.C2;
                        const bool second = c * 64 >= Ca;
                        const int Cs = second ? Cb : Ca, cb = second ? c * 64 - Ca : c * 64;
                        const int64_t aoff = seg2 ? (second ? op.a4_off : op.a3_off) : (second ? op.a2_off : op.a1_off);
                        const lvl_rsrc rs = lvl_make_rsrc(A.ws + aoff, (unsigned)npix * (unsigned)Cs * 2u);
                        uint4 v[GB][8];
#pragma unroll
                        for (int j = 0; j < GB; ++j)
#pragma unroll
                            for (int i = 0; i < 8; ++i) {
                                const int row = 8 * i + (lane >> 3), px = (bt * GB + j) * LVL_BM + row;
                                v[j][i] = lvl_ld(rs, (unsigned)((px < npix ? px : 0) * Cs + cb + (lane & 7) * 8) * 2u);
                            }
#pragma unroll
                        for (int j = 0; j < GB; ++j)
#pragma unroll
                            for (int i = 0; i < 8; ++i)
                                *reinterpret_cast<uint4*>(ldsA + j * LVL_GS + (8 * i + (lane >> 3)) * LVL_PITCH + (lane & 7) * 16) = v[j][i];
                        if (NJ == 2 && lane < GB * 9)  // (the cross-wave sum of the previous iteration ran over the rows of zeros)
                            *reinterpret_cast<uint4*>(ldsA + (lane / 9) * LVL_GS + LVL_BM * LVL_PITCH + (lane % 9) * 16) = make_uint4(0u, 0u, 0u, 0u);
                    }
                    if (pt == 0 && d_left > 0) dma_next();  // the ring is full now: RING units requested
                    LV_WAIT_UNIT();
                    if (pt == 0) LV_STAMP(2);
/* 64-cout slices.  k-group KG of the unit at tap TAP into set SET (the unit's slot is `rslot`) */                                \
#define LV2_READ(SET, TAP, KG)                                                                                                     \
    do {                                                                                                                           \
        _Pragma("unroll") for (int mi = 0; mi < MI; ++mi)                                                                          \
            ag[SET][mi] = __builtin_bit_cast(uint4, *reinterpret_cast<const lds_u32x4_lv*>((const lds_c*)(size_t)ab[TAP][mi & 1] + ((KG) * 32 + (mi >> 1) * LVL_GS))); \
        const lds_c* rb_ = (const lds_c*)(size_t)rslot;                                                                            \
        _Pragma("unroll") for (int nj = 0; nj < NJ; ++nj)                                                                          \
            bg[SET][nj] = __builtin_bit_cast(uint4, *reinterpret_cast<const lds_u32x4_lv*>(rb_ + tb[KG] + nj * 32 * ROW_DATA));    \
        __builtin_amdgcn_sched_barrier(0);                                                                                         \
    } while (0)
#define LV2_MMA(SET)                                                                               \
    do {                                                                                           \
        _Pragma("unroll") for (int mi = 0; mi < MI; ++mi)                                          \
            _Pragma("unroll") for (int nj = 0; nj < NJ; ++nj) mma16<T>(bg[SET][nj], ag[SET][mi], acc[mi][nj]); \
        __builtin_amdgcn_sched_barrier(0);                                                         \
    } while (0)
/* one unit (its k-group 0 is already on its way into set 0): the reads run one k-group ahead of the MFMAs; once the unit's last     \
   read has landed its ring slot takes the request for the unit two further on, and k-group 0 of the pass's next unit (tap NTAP)    \
   is read under the last MFMAs */                                                                                                 \
#define LV2_UNIT(TAP, NTAP)                                                                                                        \
    do {                                                                                                                           \
        LV2_READ(1, TAP, 1);                                                                                                       \
        LV2_MMA(0);                                                                                                                \
        LV2_READ(0, TAP, 2);                                                                                                       \
        LV2_MMA(1);                                                                                                                \
        LV2_READ(1, TAP, 3);                                                                                                       \
        LV2_MMA(0);                                                                                                                \
        wait_lgkm_all();                                                                                                           \
        __builtin_amdgcn_sched_barrier(0);                                                                                         \
        rslot = rslot + U_BYTES == ring_base + RING * U_BYTES ? ring_base : rslot + U_BYTES;                                       \
        ++q_read;                                                                                                                  \
        if (d_left > 0) dma_next();                                                                                                \
        __builtin_amdgcn_sched_barrier(0);                                                                                         \
        if ((NTAP) >= 0) {                                                                                                         \
            LV_WAIT_UNIT();                                                                                                        \
            LV2_READ(0, (NTAP) < 0 ? 0 : (NTAP), 0);                                                                               \
        }                                                                                                                          \
        LV2_MMA(1);                                                                                                                \
    } while (0)
                    if (NJ == 2) {
                        if (taps_p == 9) {
                            LV2_READ(0, 0, 0);
                            LV2_UNIT(0, 1);
                            LV2_UNIT(1, 2);
                            LV2_UNIT(2, 3);
                            LV2_UNIT(3, 4);
                            LV2_UNIT(4, 5);
                            LV2_UNIT(5, 6);
                            LV2_UNIT(6, 7);
                            LV2_UNIT(7, 8);
                            LV2_UNIT(8, -1);
                        } else {
                            LV2_READ(0, 4, 0);
                            LV2_UNIT(4, -1);
                        }
                    } else if (taps_p == 9) {
                        LV_READ_FRAGS(0, 0);
                        LV_STEP(0, 1);
                        LV_STEP(1, 2);
                        LV_STEP(0, 3);
                        LV_STEP(1, 4);
                        LV_STEP(0, 5);
                        LV_STEP(1, 6);
                        LV_STEP(0, 7);
                        LV_STEP(1, 8);
                        LV_STEP(0, -1);
                    } else {
                        LV_READ_FRAGS(0, 4);
                        LV_STEP(0, -1);
                    }
                }
#undef LV2_UNIT
#undef LV2_MMA
#undef LV2_READ
#undef LV_STEP
#undef LV_MMA
#undef LV_WAIT_UNIT
#undef LV_READ_FRAGS
                primed = false;
                LV_STAMP(3);
                // ---- partial tiles of this wave -> its own LDS (every unit has been consumed; the next stream starts after the epilogue).
                // 32-cout slices: 16 KB, the ring - the A image stays intact for an op that shares it; 64-cout slices: 32 KB from the
                // start of the wave's region (the next gather rewrites the A image and its rows of zeros) ----
                if (NJ == 2) {
                    LV_LOAD_FOLD();
                    LV_LOAD_GAMMA_BETA();
                }
#undef LV_LOAD_FOLD
                constexpr int RED_OFF = NJ == 1 ? A_BYTES : 0;
                // The MFMAs run transposed (filter rows as the A operand): a lane holds ONE pixel (column lane & 31) and, per group of four
                // registers g, the four consecutive couts 8 g + 4 (lane >> 5) ..: a partial tile is [32 pixels][32 couts] fp32 in 128-byte
                // rows, written as 16-byte pieces (chunk 2 g + h, XOR-swizzled by the pixel) - 4 LDS writes per tile instead of 16
                char* redw = ldsA + RED_OFF;
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                    for (int nj = 0; nj < NJ; ++nj)
#pragma unroll
                        for (int g4 = 0; g4 < 4; ++g4)
                            *reinterpret_cast<f32x4*>(redw + (mi * NJ + nj) * 4096 + r * 128 + (((2 * g4 + h) ^ (r & 7)) << 4)) =
                                f32x4{acc[mi][nj][4 * g4], acc[mi][nj][4 * g4 + 1], acc[mi][nj][4 * g4 + 2], acc[mi][nj][4 * g4 + 3]};
                __syncthreads();
                const int pr = m & 31;
#pragma unroll
                for (int q = 0; q < GQ; ++q) {
                    const int j = q / NJ, nj = q % NJ, co = co0 + 32 * nj;
                    float v[8];
#pragma unroll
                    for (int w = 0; w < 4; ++w) {
                        const char* pq = lds + w * LVL_WAVE_BYTES + RED_OFF + ((2 * j + (m >> 5)) * NJ + nj) * 4096 + pr * 128;
                        const f32x4 x0 = *reinterpret_cast<const f32x4*>(pq + (((2 * vec) ^ (pr & 7)) << 4)), x1 = *reinterpret_cast<const f32x4*>(pq + (((2 * vec + 1) ^ (pr & 7)) << 4));
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            v[e] = w == 0 ? x0[e] : v[e] + x0[e];
                            v[4 + e] = w == 0 ? x1[e] : v[4 + e] + x1[e];
                        }
                    }
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] += fold[q][e];
                    if (op.res_off >= 0) {
                        float rv[8];
                        unpack8<T>(resv[q], rv);
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] += rv[e];
                    }
                    ovec[q] = pack8<T>(v);
                    if (op.dst_off >= 0 && gpv[j] >= 0) {
                        const lvl_rsrc rd = lvl_make_rsrc(A.ws + op.dst_off, (unsigned)npix * (unsigned)op.dst_C * 2u);
                        lvl_st(rd, (unsigned)(gpv[j] * op.dst_C + op.dst_c0 + co) * 2u, ovec[q]);
                    }
                    if (GB == 1 && op.keep >= 0) *reinterpret_cast<uint4*>(keep + (op.keep * LVL_BM + m) * LVL_BN + vec * 8) = ovec[q];
                }
                have_out = true;
            } else if (op.kind == LVL_NORM) {
                // a tensor written before this launch (stride-2 / upsampling conv output): its slice is only normalised here
                const T* src = reinterpret_cast<const T*>(A.ws + op.dst_off);
#pragma unroll
                for (int q = 0; q < GQ; ++q)
                    ovec[q] = *reinterpret_cast<const uint4*>(src + (int64_t)(gpv[q / NJ] < 0 ? 0 : gpv[q / NJ]) * op.dst_C + op.dst_c0 + co0 + 32 * (q % NJ));
                have_out = true;
            } else if (GB == 1 && NJ == 1) {
                const int co = co0;
                // ---- single-head attention over the 16 pixels of each 4x4 image (models/ddpm.py:54-63): this workgroup holds channels
                // [32 s, 32 s + 32) of q, k, v of its 4 images (keep slots 0 / 1 / 2).  Partial scores over those channels -> exchange ->
                // full scores, softmax, P (rounded to T as attn_s16_kernel does) x this slice of v.
                const int g = bt;
                const int i_img = tid >> 6, qa = (tid >> 2) & 15, bq = tid & 3;
                float ps[4] = {0.f, 0.f, 0.f, 0.f};
                {
                    const T* qrow = keep + (0 * LVL_BM + i_img * 16 + qa) * LVL_BN;
                    float qv[32];
#pragma unroll
                    for (int c8 = 0; c8 < 4; ++c8) {
                        float t8[8];
                        unpack8<T>(*reinterpret_cast<const uint4*>(qrow + c8 * 8), t8);
#pragma unroll
                        for (int e = 0; e < 8; ++e) qv[c8 * 8 + e] = t8[e];
                    }
#pragma unroll
                    for (int bb = 0; bb < 4; ++bb) {
                        const T* krow = keep + (1 * LVL_BM + i_img * 16 + bq * 4 + bb) * LVL_BN;
#pragma unroll
                        for (int c8 = 0; c8 < 4; ++c8) {
                            float t8[8];
                            unpack8<T>(*reinterpret_cast<const uint4*>(krow + c8 * 8), t8);
#pragma unroll
                            for (int e = 0; e < 8; ++e) ps[bb] = fmaf(qv[c8 * 8 + e], to_f(from_f<T>(t8[e] * op.kscale)), ps[bb]);
                        }
                    }
                }
                const lvl_rsrc rsc = lvl_make_rsrc(A.ws + op.sc_off, (unsigned)A.NG * (unsigned)(LVL_NS * 1024 * 4));
                lvl_st(rsc, (unsigned)((g * LVL_NS + s) * 1024 + tid * 4) * 4u,
                       make_uint4(__float_as_uint(ps[0]), __float_as_uint(ps[1]), __float_as_uint(ps[2]), __float_as_uint(ps[3])));
                wait_vm_all();
                __syncthreads();
                if (tid == 0 && !withhold) lvl_flag_store(A.flags + ((int64_t)(oi * 2 + 1) * A.NG + g) * LVL_NS + s, epoch);
                lvl_wait_row(A.flags + ((int64_t)(oi * 2 + 1) * A.NG + g) * LVL_NS, NS, epoch, lane, err);
                lvl_compiler_fence();
                uint4 part[LVL_NS];
#pragma unroll
                for (int w = 0; w < LVL_NS; ++w) part[w] = lvl_ld(rsc, (unsigned)((g * LVL_NS + w) * 1024 + tid * 4) * 4u);
                float sc4[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int w = 0; w < LVL_NS; ++w) {  // fixed order: every workgroup of the group forms the same sums
                    sc4[0] += __uint_as_float(part[w].x);
                    sc4[1] += __uint_as_float(part[w].y);
                    sc4[2] += __uint_as_float(part[w].z);
                    sc4[3] += __uint_as_float(part[w].w);
                }
                float mx = fmaxf(fmaxf(sc4[0], sc4[1]), fmaxf(sc4[2], sc4[3]));  // a query's 16 keys: this lane's 4 and its quad's
                mx = fmaxf(mx, DMME_DPP_F(mx, 0xB1));  // quad_perm [1,0,3,2]
                mx = fmaxf(mx, DMME_DPP_F(mx, 0x4E));  // quad_perm [2,3,0,1]
                float ex[4], tot = 0.f;
#pragma unroll
                for (int bb = 0; bb < 4; ++bb) {
                    ex[bb] = expf(sc4[bb] - mx);
                    tot += ex[bb];
                }
                tot += DMME_DPP_F(tot, 0xB1);
                tot += DMME_DPP_F(tot, 0x4E);
                const float inv = 1.0f / tot;
                float* Pm = reinterpret_cast<float*>(lds);  // [4 images][16][16]; the A regions are idle during this op
                *reinterpret_cast<f32x4*>(Pm + tid * 4) = f32x4{to_f(from_f<T>(ex[0] * inv)), to_f(from_f<T>(ex[1] * inv)), to_f(from_f<T>(ex[2] * inv)),
                                                                 to_f(from_f<T>(ex[3] * inv))};
                __syncthreads();
                // out[pixel m][channels vec * 8 ..] = sum_b P[a][b] v[b][..]
                float ov[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                const float* prow = Pm + (m >> 4) * 256 + (m & 15) * 16;
#pragma unroll
                for (int b4 = 0; b4 < 4; ++b4) {
                    const f32x4 p4 = *reinterpret_cast<const f32x4*>(prow + b4 * 4);
#pragma unroll
                    for (int bb = 0; bb < 4; ++bb) {
                        float vv[8];
                        unpack8<T>(*reinterpret_cast<const uint4*>(keep + (2 * LVL_BM + (m & ~15) + b4 * 4 + bb) * LVL_BN + vec * 8), vv);
#pragma unroll
                        for (int e = 0; e < 8; ++e) ov[e] = fmaf(p4[bb], vv[e], ov[e]);
                    }
                }
                if (gpv[0] >= 0) {
                    const lvl_rsrc rd = lvl_make_rsrc(A.ws + op.dst_off, (unsigned)npix * (unsigned)op.dst_C * 2u);
                    lvl_st(rd, (unsigned)(gpv[0] * op.dst_C + op.dst_c0 + co) * 2u, pack8<T>(ov));
                }
            }

            // ---- the GroupNorms that read these slices: statistics, rows, the consumers' pre-activated inputs ----
            if (have_out && op.n_norm > 0) {
                float x[GQ][8];
#pragma unroll
                for (int q = 0; q < GQ; ++q) {
                    unpack8<T>(ovec[q], x[q]);  // statistics of the values the consumers read back (rounded to T)
                    float sm = 0.f;
#pragma unroll
                    for (int e = 0; e < 8; ++e) sm += x[q][e];
                    float mean = sm * 0.125f, m2 = 0.f;
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float d = x[q][e] - mean;
                        m2 = fmaf(d, d, m2);
                    }
                    lvl_merge16(mean, m2);  // the wave's 16 consecutive pixels of this channel vector: 128 values
                    // per norm: the group's f adjacent vectors merged in the wave (quad exchanges), one pair per (norm, vector, pixel block)
                    // in LDS - what a thread reads back is the finished statistic of its pixel block
#pragma unroll
                    for (int kx = 0; kx < 2; ++kx) {
                        if (kx >= op.n_norm) break;
                        const int f = op.norm[kx].cg >> 3;
                        float gmn = mean, gm2 = m2;
                        if (f >= 2) lvl_merge_pair(gmn, gm2, DMME_DPP_F(gmn, 0xB1), DMME_DPP_F(gm2, 0xB1), 128.f);  // quad_perm [1,0,3,2]
                        if (f >= 4) lvl_merge_pair(gmn, gm2, DMME_DPP_F(gmn, 0x4E), DMME_DPP_F(gm2, 0x4E), 256.f);  // quad_perm [2,3,0,1]
                        if (lane < 4) *reinterpret_cast<float2*>(blk + ((((q * 2 + kx) * 4 + lane) * 4 + wave) * 2)) = make_float2(gmn, gm2);
                    }
                }
                __syncthreads();
                const int nb = HW >> 4;  // pixel blocks (16 px) per image: 1 (4x4 maps) or 4 (8x8 maps: the image is the whole group)
#pragma unroll
                for (int q = 0; q < GQ; ++q) {
                    const int j = q / NJ, nj = q % NJ, co = co0 + 32 * nj;
                    const int n_img = (gpv[j] < 0 ? 0 : gpv[j]) >> sh2;
#pragma unroll
                    for (int kx = 0; kx < 2; ++kx) {
                        if (kx >= op.n_norm) break;
                        const LvlNorm& G = op.norm[kx];
                        const int f = G.cg >> 3, cgsh = 31 - __builtin_clz((unsigned)G.cg);
                        const float* pq = blk + ((q * 2 + kx) * 4 + vec) * 8;
                        float gmean, gm2;
                        if (nb == 1) {
                            const float2 p1 = *reinterpret_cast<const float2*>(pq + (m >> 4) * 2);
                            gmean = p1.x;
                            gm2 = p1.y;
                        } else {  // four equal-count sets: the mean of the means, M2 = sum M2_i + n sum (mean_i - mean)^2
                            const f32x4 p0 = *reinterpret_cast<const f32x4*>(pq), p1 = *reinterpret_cast<const f32x4*>(pq + 4);
                            gmean = 0.25f * ((p0[0] + p0[2]) + (p1[0] + p1[2]));
                            const float d0 = p0[0] - gmean, d1 = p0[2] - gmean, d2 = p1[0] - gmean, d3 = p1[2] - gmean;
                            gm2 = ((p0[1] + p0[3]) + (p1[1] + p1[3])) + (float)(128 * f) * ((d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3));
                        }
                        const float rstd = __builtin_amdgcn_rsqf(gm2 * __builtin_amdgcn_rcpf((float)(128 * f * nb)) + 1e-5f);
                        const int cn = G.c_off + co;
                        float sc[8], shf[8];
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            sc[e] = rstd * gm[kx][nj][e >> 2][e & 3];
                            shf[e] = bt4[kx][nj][e >> 2][e & 3] - gmean * sc[e];
                        }
                        if (gpv[j] >= 0 && (m & (HW - 1)) == 0) {  // first pixel of an image: the rows the backward pass reads
                            float* so = reinterpret_cast<float*>(A.ws + G.scale_off) + (int64_t)n_img * G.Cn + cn;
                            float* ho = reinterpret_cast<float*>(A.ws + G.shift_off) + (int64_t)n_img * G.Cn + cn;
                            *reinterpret_cast<f32x4*>(so) = f32x4{sc[0], sc[1], sc[2], sc[3]};
                            *reinterpret_cast<f32x4*>(so + 4) = f32x4{sc[4], sc[5], sc[6], sc[7]};
                            *reinterpret_cast<f32x4*>(ho) = f32x4{shf[0], shf[1], shf[2], shf[3]};
                            *reinterpret_cast<f32x4*>(ho + 4) = f32x4{shf[4], shf[5], shf[6], shf[7]};
                            if ((vec & (f - 1)) == 0) {
                                float* mo = reinterpret_cast<float*>(A.ws + G.mr_off) + ((int64_t)n_img * (G.Cn >> cgsh) + (cn >> cgsh)) * 2;
                                mo[0] = gmean;
                                mo[1] = rstd;
                            }
                        }
                        if (G.act_off >= 0 && gpv[j] >= 0) {
                            float y[8];
#pragma unroll
                            for (int e = 0; e < 8; ++e) y[e] = fmaf(x[q][e], sc[e], shf[e]);
                            if (G.act_silu) {
#pragma unroll
                                for (int e = 0; e < 8; ++e) y[e] = silu_fast(y[e]);
                            }
                            if (G.dmask_off >= 0 && A.drop_masks) {
                                const float* dm = A.drop_masks + G.dmask_off + (int64_t)n_img * G.Cn + cn;
                                const f32x4 d0 = *reinterpret_cast<const f32x4*>(dm), d1 = *reinterpret_cast<const f32x4*>(dm + 4);
#pragma unroll
                                for (int e = 0; e < 4; ++e) {
                                    y[e] *= d0[e];
                                    y[4 + e] *= d1[e];
                                }
                            }
                            const lvl_rsrc ra = lvl_make_rsrc(A.ws + G.act_off, (unsigned)npix * (unsigned)G.Cn * 2u);
                            lvl_st(ra, (unsigned)(gpv[j] * G.Cn + cn) * 2u, pack8<T>(y));
                        }
                    }
                }
            }

            // ---- publish: every store of every wave acknowledged, then one flag word per group ----
            // The filter stream of whatever conv comes next starts before the wait for its input - and, where the statistics barrier above
            // already separates this op's last read of the ring (the cross-wave sum) from new writes into it, even before the wait for
            // this op's own stores: the DMAs are issued behind the stores, the counted wait retires the stores and leaves the DMAs in
            // flight (they retire in order), so the ~0.5 us of issuing them hides under the stores' round trip.
            LV_STAMP(4);
            int oj = oi;
            if (bt + A.NGS >= NB) ++oj;
            const bool next_conv = oj < A.n_ops && ops[oj].kind == LVL_CONV;
            const bool early = next_conv && have_out && op.n_norm > 0;
            if (early) {
                prime(ops[oj]);
                if (d_req >= D) wait_vm_keep<D * NPI>();
                else wait_vm_all();  // (a stream shorter than the ring: rare, 1x1 convs of 256 channels)
            } else {
                wait_vm_all();
            }
            __syncthreads();
            LV_STAMP(5);
            if (op.signal && !withhold && tid < GB && bt * GB + tid < A.NG) lvl_flag_store(A.flags + ((int64_t)(oi * 2) * A.NG + bt * GB + tid) * LVL_NS + s, epoch);
            if (next_conv && !early) prime(ops[oj]);
            LV_STAMP(6);
            ++stamp_it;
        }
    }
#undef LV_LOAD_GAMMA_BETA
#undef LV_STAMP
    if (tid == 0) {  // the last workgroup to finish closes the epoch
        const unsigned old = __hip_atomic_fetch_add(&A.ctl[1], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (old == gridDim.x - 1) {
            lvl_flag_store(&A.ctl[1], 0u);
            lvl_flag_store(&A.ctl[0], epoch);
        }
    }
}

template <typename T, int GB, int NJ>
static int lvl_inst_setup() {
    static bool attr_done = false;
    if (!attr_done) {
        DMME_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(lvl_engine_kernel<T, GB, NJ>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_done = true;
    }
    return DMME_OK;
}

template <typename T, int GB, int NJ>
static int launch_lvl_inst(const LvlArgs& a, hipStream_t s) {
    const int rc = lvl_inst_setup<T, GB, NJ>();
    if (rc != DMME_OK) return rc;
    hipLaunchKernelGGL((lvl_engine_kernel<T, GB, NJ>), dim3((unsigned)(a.NGS * (LVL_NS / NJ))), dim3(256), LVL_LDS, s, a, a.ops);
    DMME_CHECK_LAUNCH();
    return DMME_OK;
}

// The hand-offs spin: every workgroup of a launch must be resident at the same time.  The plan sizes its grids by this figure
// (assign_levels) and falls back to per-op launches when a level does not fit.
template <typename T>
static int lvl_max_resident_t(int device) {
    int cus = 0, per = 0, worst = 1 << 30;
    DMME_CHECK_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device));
    int rc;
    if ((rc = lvl_inst_setup<T, 1, 1>()) != DMME_OK || (rc = lvl_inst_setup<T, 2, 1>()) != DMME_OK || (rc = lvl_inst_setup<T, 2, 2>()) != DMME_OK) return -1;
    DMME_CHECK_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per, lvl_engine_kernel<T, 1, 1>, 256, LVL_LDS));
    worst = per < worst ? per : worst;
    DMME_CHECK_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per, lvl_engine_kernel<T, 2, 1>, 256, LVL_LDS));
    worst = per < worst ? per : worst;
    DMME_CHECK_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per, lvl_engine_kernel<T, 2, 2>, 256, LVL_LDS));
    worst = per < worst ? per : worst;
    const int64_t n = (int64_t)cus * worst;
    return (int)(n < LVL_MAX_WG ? n : LVL_MAX_WG);
}

int lvl_engine_max_resident(int dtype, int device) {
    int cur = -1;
    if (hipGetDevice(&cur) != hipSuccess || cur != device) {
        if (hipSetDevice(device) != hipSuccess) {
            set_error("level engine: cannot select device %d", device);
            return -1;
        }
    }
    const int n = dtype == DMME_F16 ? lvl_max_resident_t<f16>(device) : lvl_max_resident_t<bf16>(device);
    if (cur >= 0 && cur != device) (void)hipSetDevice(cur);
    return n < 0 ? -1 : n;
}

int launch_lvl_engine(int dtype, const LvlArgs& a, hipStream_t s) {
    DMME_REQUIRE(dtype == DMME_BF16 || dtype == DMME_F16, DMME_ERR_UNSUPPORTED, "level engine: 16-bit operand types only");
    const int max_wg = a.max_wg > 0 && a.max_wg < LVL_MAX_WG ? a.max_wg : LVL_MAX_WG;
    DMME_REQUIRE(a.NGS >= 1 && (a.NJ == 1 || (a.NJ == 2 && a.GB == 2)) && a.NGS * (LVL_NS / a.NJ) <= max_wg && a.NG >= 1 && (a.sh == 2 || a.sh == 3) &&
                     (a.GB == 1 || a.GB == 2),
                 DMME_ERR_INVALID, "level engine: bad geometry (grid %d, device holds %d workgroups)", a.NGS * (LVL_NS / a.NJ), max_wg);
    if (dtype == DMME_F16) return a.NJ == 2 ? launch_lvl_inst<f16, 2, 2>(a, s) : a.GB == 2 ? launch_lvl_inst<f16, 2, 1>(a, s) : launch_lvl_inst<f16, 1, 1>(a, s);
    return a.NJ == 2 ? launch_lvl_inst<bf16, 2, 2>(a, s) : a.GB == 2 ? launch_lvl_inst<bf16, 2, 1>(a, s) : launch_lvl_inst<bf16, 1, 1>(a, s);
}

}  // namespace dmme
