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

constexpr int LVL_RING = 4;                                            // filter units (32 couts x 128 B = 4 KB) per wave ring
constexpr int LVL_D = LVL_RING - 1;                                    // units requested ahead of the one being consumed
constexpr int LVL_NPI = LVL_BN / 8;                                    // DMA wave-instructions per unit
constexpr int LVL_ZROW = 2 * LVL_BM;                                   // LDS row of zeros (out-of-image taps read it)
constexpr int LVL_A_BYTES = (2 * LVL_BM + 8) * ROW_DATA;               // two 64-channel chunks of the group's pixels + zeros
constexpr int LVL_U_BYTES = LVL_BN * ROW_DATA;
constexpr int LVL_WAVE_BYTES = LVL_A_BYTES + LVL_RING * LVL_U_BYTES;
constexpr int LVL_KEEP_OFF = 4 * LVL_WAVE_BYTES;                       // q / k / v slices of the attention block: [3][64 px][32 ch] T
constexpr int LVL_KEEP_BYTES = 3 * LVL_BM * LVL_BN * 2;
constexpr int LVL_BLK_OFF = LVL_KEEP_OFF + LVL_KEEP_BYTES;             // statistics exchange: [4 pixel blocks][4 vectors][2]
constexpr int LVL_LDS = LVL_BLK_OFF + 512;
constexpr int LVL_SPIN_LIMIT = 1 << 19;                                // polls before a wait gives up (~a second)
static_assert(LVL_LDS <= 160 * 1024, "level engine: LDS budget");
static_assert(LVL_D * LVL_NPI < 64, "vmcnt is a 6-bit counter");

size_t lvl_engine_lds_bytes() { return LVL_LDS; }

typedef unsigned u32x4_lv __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) u32x4_lv lds_u32x4_lv;

template <typename T>
struct Vec8 {
    typedef T type __attribute__((ext_vector_type(8)));
};

template <typename T>
__device__ __forceinline__ void mma16(const uint4& a, const uint4& b, f32x16& acc);
template <>
__device__ __forceinline__ void mma16<bf16>(const uint4& a, const uint4& b, f32x16& acc) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc, 0, 0, 0);
}
template <>
__device__ __forceinline__ void mma16<f16>(const uint4& a, const uint4& b, f32x16& acc) {
    typedef f16 f16x8_lv __attribute__((ext_vector_type(8)));
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_lv, a), __builtin_bit_cast(f16x8_lv, b), acc, 0, 0, 0);
}

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

// This WAVE waits until the LVL_NS flag words of row `f` carry this launch's epoch.  Lanes 0-7 poll (sc1 loads), the vote is
// wave-wide; bounded: after LVL_SPIN_LIMIT polls the error word is set and the wave goes on.
__device__ __forceinline__ void lvl_wait_row(const unsigned* f, unsigned epoch, int lane, unsigned* err) {
    for (int spin = 0;; ++spin) {
        const unsigned v = lane < LVL_NS ? lvl_flag_load(f + lane) : epoch;
        if (__all(v == epoch)) return;
        if ((spin & 1023) == 1023) {  // a wait that timed out anywhere ends every other wait too: the launch drains in milliseconds
            const unsigned e = lvl_flag_load(err);
            if (e != 0u) return;
            if (spin >= LVL_SPIN_LIMIT) {
                if (lane == 0) lvl_flag_store(err, 1u);
                return;
            }
        }
        __builtin_amdgcn_s_sleep(2);
    }
}
// (callers put a compiler barrier behind the wait: nothing orders later loads behind a relaxed atomic load for the compiler)
__device__ __forceinline__ void lvl_compiler_fence() { asm volatile("" ::: "memory"); }

template <typename T>
__global__ void __launch_bounds__(256, 1) lvl_engine_kernel(LvlArgs A, const LvlOp* __restrict__ ops) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int s = (int)blockIdx.x % LVL_NS, g0 = (int)blockIdx.x / LVL_NS;
    char* ldsA = lds + wave * LVL_WAVE_BYTES;
    char* ldsR = ldsA + LVL_A_BYTES;
    T* keep = reinterpret_cast<T*>(lds + LVL_KEEP_OFF);
    float* blk = reinterpret_cast<float*>(lds + LVL_BLK_OFF);
    const unsigned epoch = lvl_flag_load(&A.ctl[0]) + 1u;  // (the counter moves only after EVERY workgroup of a launch has finished)
    unsigned* const err = &A.ctl[2];

    const int sh = A.sh, sh2 = 2 * sh, HW = 1 << sh2, mW = (1 << sh) - 1;
    const int npix = A.N * HW;
    if (lane < 64) *reinterpret_cast<uint4*>(ldsA + LVL_ZROW * ROW_DATA + lane * 16) = make_uint4(0u, 0u, 0u, 0u);  // 8 rows of zeros

    // ---- per-lane fragment geometry (fixed for the launch): MFMA row r of pixel block mi is pixel mi * 32 + r of the group ----
    unsigned a_valid[2];  // bit t: tap t of this lane's pixel lies inside its image
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
        const int m = mi * 32 + r, tx = m & mW, ty = (m >> sh) & mW;
        a_valid[mi] = 0;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int yy = ty + t / 3 - 1, xx = tx + t % 3 - 1;
            a_valid[mi] |= (yy >= 0 && yy <= mW && xx >= 0 && xx <= mW) ? 1u << t : 0u;
        }
    }
    int tb[4];  // filter fragment offsets inside a ring slot (row r; the XOR swizzle of conv_common.h)
#pragma unroll
    for (int kg = 0; kg < 4; ++kg) tb[kg] = (r * ROW_DATA + ((h ^ ((r >> 1) & 7)) << 4)) ^ (kg << 5);
    const lds_c* ldsA3 = (const lds_c*)ldsA;
    const unsigned ring_base = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(size_t)(lds_c*)ldsR);

    // ---- filter stream state of this wave (conv_kw.hip's ring discipline; the stream of an op may start before the op does) ----
    unsigned boff[LVL_NPI];
    const char* dptr = nullptr;
    unsigned dslot = ring_base;
    int dtap = 0, d_taps = 9, d_cin2 = 0, d_left = 0;  // d_left: units of the current stream not yet requested
    bool primed = false;
    auto dma_next = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < LVL_NPI; ++i) glds16_hidden_s(dptr, boff[i], dslot + (unsigned)(i * 8 * ROW_DATA));
        dptr += d_cin2;
        if (++dtap == d_taps) {
            dtap = 0;
            dptr += 128 - d_taps * d_cin2;
        }
        dslot = dslot + LVL_U_BYTES == ring_base + LVL_RING * LVL_U_BYTES ? ring_base : dslot + LVL_U_BYTES;
        --d_left;
    };
    // start the filter stream of conv op `o`: the first LVL_D units of this wave
    auto prime = [&](const LvlOp& o) __attribute__((always_inline)) {
        const int Cin = o.C1 + o.C2, U = (Cin >> 6) * o.taps;
        const int u0 = U * wave / 4, nu = U * (wave + 1) / 4 - u0;
#pragma unroll
        for (int i = 0; i < LVL_NPI; ++i) {
            const int row = 8 * i + (lane >> 3);
            boff[i] = (unsigned)((o.w_row0 + LVL_BN * s + row) * o.taps * Cin + ((lane & 7) ^ ((row >> 1) & 7)) * 8) * 2u;
        }
        d_taps = o.taps;
        d_cin2 = Cin * 2;
        dtap = u0 % o.taps;
        dptr = A.packed + o.w_off + ((int64_t)dtap * Cin + (u0 / o.taps) * 64) * 2;
        dslot = ring_base;
        d_left = nu;
#pragma unroll
        for (int d = 0; d < LVL_D; ++d)
            if (d_left > 0) dma_next();
        primed = true;
    };

    // ------------------------------------------------------------------------------------------------------------------------
    for (int oi = 0; oi < A.n_ops; ++oi) {
        const LvlOp& op = ops[oi];
        for (int g = g0; g < A.NG; g += A.NGS) {
            const int m = tid >> 2, vec = tid & 3;  // this thread's item of the 64 x 32 slice: pixel m, channels vec * 8 ..
            const int gp = g * LVL_BM + m;          // pixel index in the NHWC tensors
            const bool okp = gp < npix;
            const int gpc = okp ? gp : 0;
            const int n_img = gpc >> sh2;
            const int co = LVL_BN * s + vec * 8;    // first of the thread's 8 channels among the op's 256
            uint4 ovec = make_uint4(0u, 0u, 0u, 0u);  // the slice's values as stored (T)
            bool have_out = false;

            if (op.kind == LVL_CONV) {
                const int Cin = op.C1 + op.C2, taps = op.taps, U = (Cin >> 6) * taps;
                const int u0 = U * wave / 4, nu = U * (wave + 1) / 4 - u0;
                const int c_lo = u0 / taps;
                if (!primed) prime(op);
                // ---- A operand: this wave's one or two 64-channel chunks of the group's 64 pixels, gathered after the hand-off ----
                if (!op.reuse_a) {
                    if (op.wait0 >= 0) lvl_wait_row(A.flags + ((int64_t)op.wait0 * A.NG + g) * LVL_NS, epoch, lane, err);
                    if (op.wait1 >= 0) lvl_wait_row(A.flags + ((int64_t)op.wait1 * A.NG + g) * LVL_NS, epoch, lane, err);
                    lvl_compiler_fence();
                    const int c_hi = (u0 + nu - 1) / taps;
                    for (int c = c_lo; c <= c_hi; ++c) {
                        const bool second = c * 64 >= op.C1;
                        const int Cs = second ? op.C2 : op.C1, cb = second ? c * 64 - op.C1 : c * 64;
                        const lvl_rsrc rs = lvl_make_rsrc(A.ws + (second ? op.a2_off : op.a1_off), (unsigned)npix * (unsigned)Cs * 2u);
                        uint4 v[8];
#pragma unroll
                        for (int i = 0; i < 8; ++i) {
                            const int row = 8 * i + (lane >> 3), p = g * LVL_BM + row;
                            v[i] = lvl_ld(rs, (unsigned)((p < npix ? p : 0) * Cs + cb + ((lane & 7) ^ ((row >> 1) & 7)) * 8) * 2u);
                        }
#pragma unroll
                        for (int i = 0; i < 8; ++i)
                            *reinterpret_cast<uint4*>(ldsA + ((c - c_lo) * LVL_BM + 8 * i + (lane >> 3)) * ROW_DATA + (lane & 7) * 16) = v[i];
                    }
                }
                // ---- main loop: units u0 .. u0 + nu - 1 of this wave, software-pipelined as in conv_kw.hip ----
                f32x16 acc[2];
#pragma unroll
                for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                    for (int j = 0; j < 16; ++j) acc[mi][j] = 0.f;
                uint4 af[2][4][2], bfr[2][4];
#define LV_READ_FRAGS(SET, TAP, SLOT_ADDR, AOFF)                                                                                  \
    do {                                                                                                                           \
        const int t9_ = taps == 9 ? (TAP) : 4;                                                                                     \
        const int ty3_ = t9_ >= 6 ? 2 : t9_ >= 3 ? 1 : 0, tx3_ = t9_ - 3 * ty3_;                                                   \
        const int tap_off_ = ((ty3_ - 1) << sh) + (tx3_ - 1) + (AOFF);                                                             \
        int ta_[2];                                                                                                                \
        _Pragma("unroll") for (int mi = 0; mi < 2; ++mi) {                                                                         \
            int row_ = mi * 32 + r + tap_off_;                                                                                     \
            if (!((a_valid[mi] >> t9_) & 1u)) row_ = LVL_ZROW;                                                                     \
            ta_[mi] = row_ * ROW_DATA + ((h ^ ((row_ >> 1) & 7)) << 4);                                                            \
        }                                                                                                                          \
        const lds_c* rb_ = (const lds_c*)(size_t)(SLOT_ADDR);                                                                      \
        _Pragma("unroll") for (int kg = 0; kg < 4; ++kg) {                                                                         \
            _Pragma("unroll") for (int mi = 0; mi < 2; ++mi)                                                                       \
                af[SET][kg][mi] = __builtin_bit_cast(uint4, *reinterpret_cast<const lds_u32x4_lv*>(ldsA3 + (ta_[mi] ^ (kg << 5))));    \
            bfr[SET][kg] = __builtin_bit_cast(uint4, *reinterpret_cast<const lds_u32x4_lv*>(rb_ + tb[kg]));                            \
        }                                                                                                                          \
    } while (0)
#define LV_MMA(CUR, KG)                                  \
    do {                                                 \
        mma16<T>(af[CUR][KG][0], bfr[CUR][KG], acc[0]);  \
        mma16<T>(af[CUR][KG][1], bfr[CUR][KG], acc[1]);  \
    } while (0)
                int chunk = c_lo, tap = u0 - c_lo * taps;
                if (d_left > 0) {
                    dma_next();
                    wait_vm_keep<LVL_NPI>();  // unit 0 (and everything older: the gather) has landed; one unit may be in flight
                } else {
                    wait_vm_keep<0>();
                }
                LV_READ_FRAGS(0, tap, ring_base, 0);
                unsigned rslot = ring_base + LVL_U_BYTES;
                int k = 0;
#define LV_STEP(CUR, NXT)                                                                                                          \
    do {                                                                                                                           \
        const bool more_ = k + 1 < nu;                                                                                             \
        int ntap_ = tap + 1, nchunk_ = chunk;                                                                                      \
        if (ntap_ == taps) {                                                                                                       \
            ntap_ = 0;                                                                                                             \
            ++nchunk_;                                                                                                             \
        }                                                                                                                          \
        wait_lgkm_all(); /* set CUR is complete - and unit k's slot is free for unit k + RING */                                   \
        __builtin_amdgcn_sched_barrier(0);                                                                                         \
        LV_MMA(CUR, 0);                                                                                                            \
        __builtin_amdgcn_sched_barrier(0);                                                                                         \
        const bool req_ = more_ && d_left > 0;                                                                                     \
        if (req_) dma_next();                                                                                                      \
        __builtin_amdgcn_sched_barrier(0);                                                                                         \
        LV_MMA(CUR, 1);                                                                                                            \
        __builtin_amdgcn_sched_barrier(0);                                                                                         \
        if (more_) {                                                                                                               \
            if (req_)                                                                                                              \
                wait_vm_keep<LVL_D * LVL_NPI>(); /* unit k + 1 has landed; units k + 2 .. k + 1 + D may be in flight */            \
            else                                                                                                                   \
                wait_vm_keep<0>();                                                                                                 \
            LV_READ_FRAGS(NXT, ntap_, rslot, (nchunk_ - c_lo) * LVL_BM);                                                           \
        }                                                                                                                          \
        __builtin_amdgcn_sched_barrier(0);                                                                                         \
        LV_MMA(CUR, 2);                                                                                                            \
        LV_MMA(CUR, 3);                                                                                                            \
        __builtin_amdgcn_sched_barrier(0);                                                                                         \
        tap = ntap_;                                                                                                               \
        chunk = nchunk_;                                                                                                           \
        rslot = rslot + LVL_U_BYTES == ring_base + LVL_RING * LVL_U_BYTES ? ring_base : rslot + LVL_U_BYTES;                       \
        ++k;                                                                                                                       \
    } while (0)
#pragma unroll 1
                while (k + 1 < nu) {
                    LV_STEP(0, 1);
                    LV_STEP(1, 0);
                }
                if (k < nu) LV_STEP(0, 1);
#undef LV_STEP
#undef LV_MMA
#undef LV_READ_FRAGS
                primed = false;
                // ---- partial tile of this wave -> its own ring (every unit has been consumed; the next stream starts after the epilogue;
                // the A image stays intact for an op that shares it) ----
                float* redw = reinterpret_cast<float*>(ldsR);
#pragma unroll
                for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                    for (int j = 0; j < 16; ++j) redw[(mi * 16 + j) * 64 + lane] = acc[mi][j];
                // what the item needs besides the sums goes out before the barrier: residual, bias, time row
                uint4 resv = make_uint4(0u, 0u, 0u, 0u);
                if (op.res_off >= 0) {
                    const lvl_rsrc rr = lvl_make_rsrc(A.ws + op.res_off, (unsigned)npix * (unsigned)op.res_C * 2u);
                    resv = lvl_ld(rr, (unsigned)(gpc * op.res_C + op.res_c0 + co) * 2u);
                }
                const float* bp = reinterpret_cast<const float*>(A.packed + op.b_off) + op.w_row0 + co;
                const f32x4 b0 = *reinterpret_cast<const f32x4*>(bp), b1 = *reinterpret_cast<const f32x4*>(bp + 4);
                float fold[8] = {b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};
                if (op.tproj_col >= 0) {
                    const float* tp = A.tproj + (int64_t)(A.nt == 1 ? 0 : n_img) * A.tproj_ld + op.tproj_col + co;
                    const f32x4 t0 = *reinterpret_cast<const f32x4*>(tp), t1 = *reinterpret_cast<const f32x4*>(tp + 4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        fold[e] += t0[e];
                        fold[4 + e] += t1[e];
                    }
                }
                __syncthreads();
                // acc layout of a 32 x 32 tile: lane = cout column + 32 * (pixel row bit 2), register j = pixel rows (j & 3) + 8 * (j >> 2)
                const int pr = m & 31, jj = (pr & 3) + 4 * (pr >> 3), hh = (pr >> 2) & 1;
                float v[8];
#pragma unroll
                for (int w = 0; w < 4; ++w) {
                    const float* q = reinterpret_cast<const float*>(lds + w * LVL_WAVE_BYTES + LVL_A_BYTES) + ((m >> 5) * 16 + jj) * 64 + hh * 32 + vec * 8;
                    const f32x4 x0 = *reinterpret_cast<const f32x4*>(q), x1 = *reinterpret_cast<const f32x4*>(q + 4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        v[e] = w == 0 ? x0[e] : v[e] + x0[e];
                        v[4 + e] = w == 0 ? x1[e] : v[4 + e] + x1[e];
                    }
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] += fold[e];
                if (op.res_off >= 0) {
                    float rv[8];
                    unpack8<T>(resv, rv);
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] += rv[e];
                }
                ovec = pack8<T>(v);
                have_out = true;
                if (op.dst_off >= 0 && okp) {
                    const lvl_rsrc rd = lvl_make_rsrc(A.ws + op.dst_off, (unsigned)npix * (unsigned)op.dst_C * 2u);
                    lvl_st(rd, (unsigned)(gp * op.dst_C + op.dst_c0 + co) * 2u, ovec);
                }
                if (op.keep >= 0) *reinterpret_cast<uint4*>(keep + (op.keep * LVL_BM + m) * LVL_BN + vec * 8) = ovec;
            } else if (op.kind == LVL_NORM) {
                // a tensor written before this launch (stride-2 / upsampling conv output): its slice is only normalised here
                const T* src = reinterpret_cast<const T*>(A.ws + op.dst_off);
                ovec = *reinterpret_cast<const uint4*>(src + (int64_t)gpc * op.dst_C + op.dst_c0 + co);
                have_out = true;
            } else {
                // ---- single-head attention over the 16 pixels of each 4x4 image (models/ddpm.py:54-63): this workgroup holds channels
                // [32 s, 32 s + 32) of q, k, v of its 4 images (keep slots 0 / 1 / 2).  Partial scores over those channels -> exchange ->
                // full scores, softmax, P (rounded to T as attn_s16_kernel does) x this slice of v.
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
                if (tid == 0) lvl_flag_store(A.flags + ((int64_t)(oi * 2 + 1) * A.NG + g) * LVL_NS + s, epoch);
                lvl_wait_row(A.flags + ((int64_t)(oi * 2 + 1) * A.NG + g) * LVL_NS, epoch, lane, err);
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
                mx = fmaxf(mx, __shfl_xor(mx, 1, 64));
                mx = fmaxf(mx, __shfl_xor(mx, 2, 64));
                float ex[4], tot = 0.f;
#pragma unroll
                for (int bb = 0; bb < 4; ++bb) {
                    ex[bb] = expf(sc4[bb] - mx);
                    tot += ex[bb];
                }
                tot += __shfl_xor(tot, 1, 64);
                tot += __shfl_xor(tot, 2, 64);
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
                if (okp) {
                    const lvl_rsrc rd = lvl_make_rsrc(A.ws + op.dst_off, (unsigned)npix * (unsigned)op.dst_C * 2u);
                    lvl_st(rd, (unsigned)(gp * op.dst_C + op.dst_c0 + co) * 2u, pack8<T>(ov));
                }
            }

            // ---- the GroupNorms that read this slice: statistics, rows, the consumers' pre-activated inputs ----
            if (have_out && op.n_norm > 0) {
                float x[8];
                unpack8<T>(ovec, x);  // statistics of the values the consumers read back (rounded to T)
                // gamma / beta do not depend on the statistics: requested now
                f32x4 gm[2][2], bt[2][2];
#pragma unroll
                for (int kx = 0; kx < 2; ++kx) {
                    if (kx >= op.n_norm) break;
                    const LvlNorm& G = op.norm[kx];
                    const float* gp_ = reinterpret_cast<const float*>(A.packed + G.gamma_off) + G.c_off + co;
                    const float* bp_ = reinterpret_cast<const float*>(A.packed + G.beta_off) + G.c_off + co;
                    gm[kx][0] = *reinterpret_cast<const f32x4*>(gp_);
                    gm[kx][1] = *reinterpret_cast<const f32x4*>(gp_ + 4);
                    bt[kx][0] = *reinterpret_cast<const f32x4*>(bp_);
                    bt[kx][1] = *reinterpret_cast<const f32x4*>(bp_ + 4);
                }
                float sm = 0.f;
#pragma unroll
                for (int e = 0; e < 8; ++e) sm += x[e];
                float mean = sm * 0.125f, m2 = 0.f;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float d = x[e] - mean;
                    m2 = fmaf(d, d, m2);
                }
                float cnt = 8.f;
#pragma unroll
                for (int o = 4; o < 64; o <<= 1) {  // lanes 4 apart: the same channel vector of the wave's 16 consecutive pixels
                    const float om = __shfl_xor(mean, o, 64), o2 = __shfl_xor(m2, o, 64);
                    const float d = om - mean;
                    m2 += o2 + d * d * (0.5f * cnt);
                    mean = 0.5f * (mean + om);
                    cnt *= 2.f;
                }
                if (lane < 4) {
                    blk[(wave * 4 + lane) * 2] = mean;
                    blk[(wave * 4 + lane) * 2 + 1] = m2;
                }
                __syncthreads();
                const int nb = HW >> 4, b_first = ((m >> 4) / nb) * nb;  // pixel blocks (16 px) of this thread's image
#pragma unroll
                for (int kx = 0; kx < 2; ++kx) {
                    if (kx >= op.n_norm) break;
                    const LvlNorm& G = op.norm[kx];
                    const int f = G.cg >> 3, v_first = (vec / f) * f;  // the group = f adjacent vectors of this slice
                    float gna = 0.f, gmean = 0.f, gm2 = 0.f;
                    for (int vv = 0; vv < f; ++vv) {
                        float na = 0.f, vmean = 0.f, vm2 = 0.f;  // (image, vector): the image's blocks in pixel order
                        for (int b = 0; b < nb; ++b) {
                            const float* q = blk + ((b_first + b) * 4 + v_first + vv) * 2;
                            const float delta = q[0] - vmean, totn = na + 128.f;
                            const float rt = __builtin_amdgcn_rcpf(totn);
                            vmean += delta * (128.f * rt);
                            vm2 += q[1] + delta * delta * (na * 128.f * rt);
                            na = totn;
                        }
                        const float delta = vmean - gmean, totn = gna + na;  // (image, group): the group's vectors in channel order
                        const float rt = __builtin_amdgcn_rcpf(totn);
                        gmean += delta * (na * rt);
                        gm2 += vm2 + delta * delta * (gna * na * rt);
                        gna = totn;
                    }
                    const float rstd = 1.0f / sqrtf(gm2 / gna + 1e-5f);
                    const int cn = G.c_off + co;
                    float sc[8], shf[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        sc[e] = rstd * gm[kx][e >> 2][e & 3];
                        shf[e] = bt[kx][e >> 2][e & 3] - gmean * sc[e];
                    }
                    if (okp && (m & (HW - 1)) == 0) {  // first pixel of an image: the rows the backward pass reads
                        float* so = reinterpret_cast<float*>(A.ws + G.scale_off) + (int64_t)n_img * G.Cn + cn;
                        float* ho = reinterpret_cast<float*>(A.ws + G.shift_off) + (int64_t)n_img * G.Cn + cn;
                        *reinterpret_cast<f32x4*>(so) = f32x4{sc[0], sc[1], sc[2], sc[3]};
                        *reinterpret_cast<f32x4*>(so + 4) = f32x4{sc[4], sc[5], sc[6], sc[7]};
                        *reinterpret_cast<f32x4*>(ho) = f32x4{shf[0], shf[1], shf[2], shf[3]};
                        *reinterpret_cast<f32x4*>(ho + 4) = f32x4{shf[4], shf[5], shf[6], shf[7]};
                        if (vec % f == 0) {
                            float* mo = reinterpret_cast<float*>(A.ws + G.mr_off) + ((int64_t)n_img * (G.Cn / G.cg) + cn / G.cg) * 2;
                            mo[0] = gmean;
                            mo[1] = rstd;
                        }
                    }
                    if (G.act_off >= 0 && okp) {
                        float y[8];
#pragma unroll
                        for (int e = 0; e < 8; ++e) y[e] = fmaf(x[e], sc[e], shf[e]);
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
                        lvl_st(ra, (unsigned)(gp * G.Cn + cn) * 2u, pack8<T>(y));
                    }
                }
            }

            // ---- publish: every store of every wave acknowledged, then one flag word ----
            wait_vm_all();
            __syncthreads();
            if (op.signal && tid == 0) lvl_flag_store(A.flags + ((int64_t)(oi * 2) * A.NG + g) * LVL_NS + s, epoch);
            // the filter stream of whatever conv comes next starts now, before the wait for its input
            {
                int oj = oi, gj = g + A.NGS;
                if (gj >= A.NG) {
                    ++oj;
                    gj = g0;
                }
                if (oj < A.n_ops && ops[oj].kind == LVL_CONV) prime(ops[oj]);
            }
        }
    }
    if (tid == 0) {  // the last workgroup to finish closes the epoch
        const unsigned old = __hip_atomic_fetch_add(&A.ctl[1], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (old == gridDim.x - 1) {
            lvl_flag_store(&A.ctl[1], 0u);
            lvl_flag_store(&A.ctl[0], epoch);
        }
    }
}

int launch_lvl_engine(int dtype, const LvlArgs& a, hipStream_t s) {
    DMME_REQUIRE(dtype == DMME_BF16 || dtype == DMME_F16, DMME_ERR_UNSUPPORTED, "level engine: 16-bit operand types only");
    DMME_REQUIRE(a.NGS >= 1 && a.NGS * LVL_NS <= LVL_MAX_WG && a.NG >= 1 && (a.sh == 2 || a.sh == 3), DMME_ERR_INVALID, "level engine: bad geometry");
    static bool attr_done[2] = {false, false};
    const int ti = dtype == DMME_F16 ? 1 : 0;
    const void* fn = ti ? reinterpret_cast<const void*>(lvl_engine_kernel<f16>) : reinterpret_cast<const void*>(lvl_engine_kernel<bf16>);
    if (!attr_done[ti]) {
        DMME_CHECK_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_done[ti] = true;
    }
    const dim3 grid((unsigned)(a.NGS * LVL_NS));
    if (ti)
        hipLaunchKernelGGL(lvl_engine_kernel<f16>, grid, dim3(256), LVL_LDS, s, a, a.ops);
    else
        hipLaunchKernelGGL(lvl_engine_kernel<bf16>, grid, dim3(256), LVL_LDS, s, a, a.ops);
    DMME_CHECK_LAUNCH();
    return DMME_OK;
}

}  // namespace dmme
