// Level-engine side of the plan (csrc/lvl_engine.hip): which stretches of the op list become ONE persistent launch, their op tables,
// the launch itself, and the status word of the engine's bounded hand-off waits.
#include "plan.h"

using namespace dmme;

namespace dmme {

// ---- level engine: which stretches of the op list become ONE persistent launch (lvl.h, lvl_engine.hip) ---------------------------
// A stretch qualifies when every op in it lives on one 4x4 / 8x8 map and has the shape the engine is built for: DDPM ResBlocks whose
// convs have 256 couts (8 slices of 32), inputs of whole 64-channel chunks, norms with group sizes 8 / 16 / 32, single-head attention on
// 4x4 maps.  Anything else (other widths, the IDDPM blocks, fp32) keeps its per-op launches.
static int op_level(const dmme_plan* P, const Op& o) {  // the map width all tensors of the op share (4 / 8), 0: not a level op
    auto hw = [&](int id, int& H, int& W) {
        if (id < 0) return true;
        const Tensor& t = P->tensors[id];
        if (H == 0) {
            H = t.H;
            W = t.W;
            return true;
        }
        return t.H == H && t.W == W;
    };
    int H = 0, W = 0;
    bool ok = true;
    if (o.kind == OP_CONV) {
        if (o.src1 < 0 || o.dst < 0 || o.stride != 1 || o.up || (o.taps != 1 && o.taps != 9)) return 0;
        for (int id : {o.src1, o.src2, o.dst, o.res1, o.res2}) ok = ok && hw(id, H, W);
    } else if (o.kind == OP_GN) {
        ok = hw(o.gn_src1, H, W) && hw(o.gn_src2, H, W);
    } else if (o.kind == OP_ATTN) {
        ok = hw(o.at_qkv, H, W) && hw(o.at_out, H, W);
    } else {
        return 0;
    }
    if (!ok || H != W || (H != 4 && H != 8)) return 0;
    return H;
}

struct LvlXAttach {  // a norm of this run finished by an op of an EARLIER run (the producer of a skip tensor)
    int run, op;
    LvlNorm norm;
};
static bool build_lvl_run(dmme_plan* P, int i0, int i1, int lvl_w, int64_t& ws, LvlRun& R, std::vector<std::pair<int, int64_t>>& gn_acts,
                          const std::unordered_map<int, std::pair<int, int>>& made, std::vector<LvlXAttach>& xattach) {
    const int B = P->B, G = P->cfg.num_groups, HW = lvl_w * lvl_w;
    const int64_t es = (int64_t)dtype_size(P->dtype);
    std::vector<LvlOp> pre, body;           // LVL_NORM ops of tensors written before the launch; the ops proper
    std::unordered_map<int, int> prod;      // tensor id -> index into `body` of the op that produces it in this run
    std::unordered_map<int, int> pre_of;    // tensor id -> index into `pre`
    std::vector<int> xsrc;                  // tensors whose norms an earlier run's op finishes (complete before this launch)
    const bool merge_res_off = (debug_route("lvl_no_res_merge") != 0);
    auto is_x = [&](int t) { return std::find(xsrc.begin(), xsrc.end(), t) != xsrc.end(); };
    // flag rows are (final op index) * 2 + which; body indices are shifted by pre.size() at the end: encode body rows as 1000000 + ...
    auto row_of = [&](int tensor) -> int {
        auto it = prod.find(tensor);
        if (it != prod.end()) return 1000000 + it->second * 2;
        auto jt = pre_of.find(tensor);
        if (jt != pre_of.end()) return jt->second * 2;  // the LVL_NORM op that pre-activates it (its act is what the consumer reads)
        return -1;
    };
    auto blank = [&]() {
        LvlOp o{};
        o.kind = LVL_CONV;
        o.taps = 1;
        o.wait0 = o.wait1 = o.wait2 = o.wait3 = -1;
        o.a1_off = o.a2_off = o.a3_off = o.a4_off = o.dst_off = o.res_off = o.sc_off = -1;
        o.w2_off = o.b2_off = -1;
        o.tproj_col = -1;
        o.keep = -1;
        o.signal = 1;
        for (auto& n : o.norm) n.act_off = n.dmask_off = -1;
        return o;
    };
    for (int oi = i0; oi < i1; ++oi) {
        const Op& o = P->ops[oi];
        if (o.kind == OP_GN) {
            if (o.gn_mod_col >= 0) return false;
            const Tensor& t1 = P->tensors[o.gn_src1];
            const int C2 = o.gn_src2 >= 0 ? P->tensors[o.gn_src2].C : 0, Cn = t1.C + C2;
            if (Cn % G) return false;
            const int cg = Cn / G;
            if (cg % 8 || 32 % cg || t1.C != 256 || (C2 != 0 && C2 != 256)) return false;
            int consumer = -1;  // the conv this norm feeds (exactly one: conv1 / conv2 / qkv_proj)
            for (int ci = oi + 1; ci < i1; ++ci)
                if (P->ops[ci].kind == OP_CONV && P->ops[ci].gn == oi) {
                    if (consumer >= 0) return false;
                    consumer = ci;
                }
            if (consumer < 0) return false;
            const Op& cv = P->ops[consumer];
            if (cv.src1 != o.gn_src1 || cv.src2 != o.gn_src2) return false;
            const int64_t act = ws;
            ws = align_up(ws + (int64_t)B * HW * Cn * es, 256);
            gn_acts.push_back({oi, act});
            int coff = 0;
            for (int src : {o.gn_src1, o.gn_src2}) {
                if (src < 0) continue;
                LvlNorm n{};
                n.gamma_off = P->params[o.gn_gamma].packed_off;
                n.beta_off = P->params[o.gn_beta].packed_off;
                n.scale_off = o.gn_scale;
                n.shift_off = o.gn_shift;
                n.mr_off = o.gn_mr;
                n.act_off = act;
                n.dmask_off = cv.dmask_off;
                n.Cn = Cn;
                n.cg = cg;
                n.c_off = coff;
                n.act_silu = cv.pro_silu;
                LvlOp* host = nullptr;
                auto it = prod.find(src);
                auto mt = made.find(src);
                if (it != prod.end()) {
                    host = &body[it->second];
                } else if (mt != made.end() && pre_of.find(src) == pre_of.end()) {
                    // produced by the engine in an earlier launch (a skip tensor of the down path): that op gets the norm, if it has room
                    int used = P->lvl_runs[mt->second.first].ops[mt->second.second].n_norm;
                    for (const LvlXAttach& xa : xattach) used += xa.run == mt->second.first && xa.op == mt->second.second;
                    if (used >= 2) return false;
                    xattach.push_back({mt->second.first, mt->second.second, n});
                    xsrc.push_back(src);
                    coff += P->tensors[src].C;
                    continue;
                } else {
                    auto jt = pre_of.find(src);
                    if (jt == pre_of.end()) {
                        LvlOp q = blank();
                        q.kind = LVL_NORM;
                        q.dst_off = P->tensors[src].off;
                        q.dst_C = P->tensors[src].C;
                        q.dst_c0 = 0;
                        pre_of[src] = (int)pre.size();
                        pre.push_back(q);
                        jt = pre_of.find(src);
                    }
                    host = &pre[jt->second];
                }
                if (host->n_norm >= 2) return false;
                // a 768-wide qkv tensor never feeds a norm; every other engine tensor is 256 wide: one LvlOp per tensor
                host->norm[host->n_norm++] = n;
                coff += P->tensors[src].C;
            }
            continue;
        }
        if (o.kind == OP_ATTN) {
            const Tensor& q = P->tensors[o.at_qkv];
            if (o.at_heads != 1 || HW != 16 || q.C != 768 || body.size() < 3 || body[body.size() - 1].keep != 2) return false;
            LvlOp a = blank();
            a.kind = LVL_ATTN;
            a.dst_off = P->tensors[o.at_out].off;
            a.dst_C = P->tensors[o.at_out].C;
            a.dst_c0 = 0;
            a.kscale = 1.0f / sqrtf((float)(q.C / 3));
            a.sc_off = ws;
            ws = align_up(ws + (int64_t)R.NG * LVL_NS * 1024 * 4, 256);
            prod[o.at_out] = (int)body.size();
            body.push_back(a);
            R.flops += 4.0 * B * HW * HW * (q.C / 3);
            continue;
        }
        // OP_CONV
        const Param& w = P->params[o.w];
        const Tensor& t1 = P->tensors[o.src1];
        const int C1 = t1.C, C2 = o.src2 >= 0 ? P->tensors[o.src2].C : 0, Cin = C1 + C2;
        const bool is_qkv = w.cout == 768 && oi + 1 < i1 && P->ops[oi + 1].kind == OP_ATTN && P->ops[oi + 1].at_qkv == o.dst;
        if ((w.cout != 256 && !is_qkv) || Cin % 64 || C1 % 64 || o.out_silu || o.res2 >= 0) return false;
        if (o.gn < 0 && (o.pro_silu || o.dmask_off >= 0)) return false;
        if (Cin % 256) return false;  // the K loop runs in passes of 256 channels: one 64-channel chunk per wave and pass
        LvlOp c = blank();
        c.taps = o.taps;
        if (o.gn >= 0) {  // the pre-activated input its norm's producers wrote
            int64_t act = -1;
            for (auto& ga : gn_acts)
                if (ga.first == o.gn) act = ga.second;
            if (act < 0) return false;
            c.a1_off = act;
            c.C1 = Cin;
            c.C2 = 0;
        } else {
            c.a1_off = t1.off;
            c.C1 = C1;
            c.a2_off = o.src2 >= 0 ? P->tensors[o.src2].off : -1;
            c.C2 = C2;
        }
        c.wait0 = row_of(o.src1);
        c.wait1 = o.src2 >= 0 ? row_of(o.src2) : -1;
        if (o.gn >= 0 && ((c.wait0 < 0 && !is_x(o.src1)) || (o.src2 >= 0 && c.wait1 < 0 && !is_x(o.src2)))) return false;  // (an act tensor has a producer)
        c.w_off = w.packed_off;
        c.b_off = P->params[o.b].packed_off;
        c.dst_off = P->tensors[o.dst].off;
        c.dst_C = P->tensors[o.dst].C;
        c.tproj_col = o.tproj_col;
        if (o.res1 >= 0) {
            if (P->tensors[o.res1].C != 256) return false;
            c.res_off = P->tensors[o.res1].off;
            c.res_C = 256;
            c.res_c0 = 0;
            // The residual is the output of the block's 1x1 residual conv, the op pushed just before this one, and nothing else reads
            // it: that conv becomes this op's second K segment (LvlOp::C3) - no residual tensor, one op and one hand-off less per block.
            auto rt = prod.find(o.res1);
            if (!merge_res_off && o.taps == 9 && rt != prod.end() && rt->second == (int)body.size() - 1) {
                const LvlOp& r = body.back();
                bool only_here = true;  // (forward readers of the residual tensor: this conv alone)
                for (int ci = i0; ci < (int)P->ops.size() && only_here; ++ci) {
                    const Op& q = P->ops[ci];
                    if (ci == oi) continue;
                    if (q.kind == OP_CONV && (q.src1 == o.res1 || q.src2 == o.res1 || q.res1 == o.res1 || q.res2 == o.res1)) only_here = false;
                    if (q.kind == OP_GN && (q.gn_src1 == o.res1 || q.gn_src2 == o.res1)) only_here = false;
                    if (q.kind == OP_ATTN && q.at_qkv == o.res1) only_here = false;
                }
                if (only_here && r.kind == LVL_CONV && r.taps == 1 && r.n_norm == 0 && r.keep < 0 && r.res_off < 0 && r.tproj_col < 0 && !r.reuse_a && r.w_row0 == 0 &&
                    r.dst_c0 == 0 && r.C3 == 0) {
                    c.C3 = r.C1;
                    c.C4 = r.C2;
                    c.a3_off = r.a1_off;
                    c.a4_off = r.a2_off;
                    c.wait2 = r.wait0;
                    c.wait3 = r.wait1;
                    c.w2_off = r.w_off;
                    c.b2_off = r.b_off;
                    c.res_off = -1;
                    prod.erase(rt);
                    body.pop_back();
                }
            }
        }
        R.flops += 2.0 * B * HW * (double)w.cout * Cin * o.taps;
        R.bytes += (double)w.cout * Cin * o.taps * es + (double)B * HW * (Cin + w.cout) * es;
        if (is_qkv) {
            for (int j = 0; j < 3; ++j) {
                LvlOp q = c;
                q.w_row0 = 256 * j;
                q.dst_c0 = 256 * j;
                q.keep = j;
                q.reuse_a = j > 0;
                if (j > 0) q.wait0 = q.wait1 = -1;
                body.push_back(q);
            }
            prod[o.dst] = (int)body.size() - 1;
        } else {
            prod[o.dst] = (int)body.size();
            body.push_back(c);
        }
    }
    const int shift = (int)pre.size();
    R.ops = pre;
    for (LvlOp c : body) {
        for (int* wr : {&c.wait0, &c.wait1, &c.wait2, &c.wait3})
            if (*wr >= 1000000) *wr = (*wr - 1000000) + shift * 2;
        R.ops.push_back(c);
    }
    R.made.clear();
    for (auto& kv : pre_of) R.made.push_back({kv.first, kv.second});
    for (auto& kv : prod) R.made.push_back({kv.first, kv.second + shift});
    return !body.empty();
}

void assign_levels(dmme_plan* P) {
    if (getenv("DMME_NO_LVL") || P->cfg.arch != DMME_ARCH_DDPM || P->x3 || (P->dtype != DMME_BF16 && P->dtype != DMME_F16)) return;
    const int mask = debug_route("lvl_mask", 12);  // bit 2: 4x4 maps, bit 3: 8x8 maps
    // The engine's hand-offs spin, so a launch only works if ALL its workgroups are resident together: size the grids by what THIS
    // device holds (compute units x workgroups per unit at the kernel's 150 KB of LDS), not by a constant; a level that needs more
    // than two iterations per workgroup at that size keeps its per-op launches (below).  (lvl_max_wg=: test knob, a smaller device.)
    if (P->device >= 0) {
        const int n = lvl_engine_max_resident(P->dtype, P->device);
        if (n < LVL_NS) return;  // (also a HIP error: no engine, the per-op kernels run)
        P->lvl_max_wg = n;
    }
    if (debug_route("lvl_max_wg", 0) > 0) P->lvl_max_wg = std::min(P->lvl_max_wg, debug_route("lvl_max_wg", 0));
    if (P->lvl_max_wg < LVL_NS) return;
    const int nO = (int)P->ops.size();
    std::unordered_map<int, std::pair<int, int>> made;  // tensor id -> (run, op) of the engine op that holds its slices
    int i = 0;
    while (i < nO) {
        const int L = op_level(P, P->ops[i]);
        if (!L) {
            ++i;
            continue;
        }
        int j = i;
        while (j < nO && op_level(P, P->ops[j]) == L) ++j;
        if (mask & L) {
            LvlRun R;
            R.op_first = i;
            R.op_last = j - 1;
            R.sh = L == 4 ? 2 : 3;
            R.NG = (P->B * L * L + LVL_BM - 1) / LVL_BM;
            int64_t ws = P->ws_bytes;
            std::vector<std::pair<int, int64_t>> gn_acts;
            std::vector<LvlXAttach> xattach;
            if (build_lvl_run(P, i, j, L, ws, R, gn_acts, debug_route("lvl_no_xrun") ? std::unordered_map<int, std::pair<int, int>>() : made, xattach)) {
                // two groups per op iteration where a workgroup owns several (the filter stream is shared by twice the matrix work);
                // the attention block keeps q / k / v of ONE group in LDS
                bool has_attn = false;
                for (const LvlOp& lo : R.ops) has_attn = has_attn || lo.kind == LVL_ATTN;
                int slots = P->lvl_max_wg / LVL_NS;
                R.GB = (R.NG > slots && !has_attn && !debug_route("lvl_gb1")) ? 2 : 1;
                const int nb = (R.NG + R.GB - 1) / R.GB;
                // still more than one iteration per workgroup: 64-cout slices (4 per group) - half the iterations, the input gathered
                // by half as many workgroups, 128 x 64 per filter unit instead of 128 x 32 (B = 128, 8x8 maps: 2 iterations -> 1)
                R.NJ = (R.GB == 2 && nb > slots && !debug_route("lvl_nj1")) ? 2 : 1;
                slots *= R.NJ;
                R.NGS = nb < slots ? nb : slots;
                // more than two iterations per op and workgroup: the per-layer kernels (tiles over the whole batch) are the better
                // route again - measured at batch 512 (DDIM): 8.2 ms per step with them, 8.9 with the engine
                const int max_iter = debug_route("lvl_max_iter", 2);
                if ((nb + slots - 1) / slots > max_iter) {
                    i = j;
                    continue;
                }
                P->ws_bytes = ws;
                for (const LvlXAttach& xa : xattach) {
                    LvlOp& host = P->lvl_runs[xa.run].ops[xa.op];
                    host.norm[host.n_norm++] = xa.norm;
                }
                for (auto& mk : R.made) made[mk.first] = {(int)P->lvl_runs.size(), mk.second};
                const int ri = (int)P->lvl_runs.size();
                for (int oi = i; oi < j; ++oi) {
                    Op& o = P->ops[oi];
                    o.lvl = ri;
                    o.lvl_first = oi == i;
                    if (o.kind == OP_GN) o.gn_direct = 1;  // (no launch of its own; its rows and act come from the engine)
                }
                for (auto& ga : gn_acts) {
                    Op& g = P->ops[ga.first];
                    g.gn_act = ga.second;
                    for (int ci = ga.first + 1; ci < j; ++ci)
                        if (P->ops[ci].kind == OP_CONV && P->ops[ci].gn == ga.first) {
                            g.gn_consumer = ci;
                            P->ops[ci].use_act = 1;
                        }
                }
                P->lvl_runs.push_back(R);
            }
        }
        i = j;
    }
}

// Forwards that no backward pass follows (sampling: dmme_unet_forward_nograd, dmme_chain_step): an engine conv whose RAW output no
// forward op reads need not store it - the per-op epilogues of a run are write-through stores at the chip's rate on the critical
// chain of hand-offs (tools/stamp_lvl.py: 6-10 us of a ~20 us op on the 8x8 maps).  A ResBlock's conv1 is read through its norm's
// pre-activated copy only, the qkv slices of the 4x4 attention from LDS.  Readers of a raw tensor: convs without a norm or outside the
// pre-activated route, residual inputs, norms and attention outside the engine, an LVL_NORM op of a later run, named tensors
// (dmme_unet_debug_read).  DMME_DEBUG_ROUTE=lvl_keep_raw: the full table everywhere.
void assign_lvl_nograd(dmme_plan* P) {
    if (P->lvl_runs.empty()) return;
    std::vector<char> need(P->tensors.size(), 0);
    auto mark = [&](int id) { if (id >= 0) need[id] = 1; };
    for (const Op& q : P->ops) {
        if (q.kind == OP_CONV) {
            if (!(q.gn >= 0 && q.use_act)) { mark(q.src1); mark(q.src2); }
            mark(q.res1);
            mark(q.res2);
        } else if (q.kind == OP_GN) {
            for (int id : {q.gn_src1, q.gn_src2}) {
                if (id < 0) continue;
                if (q.lvl < 0) { need[id] = 1; continue; }
                for (const LvlOp& lo : P->lvl_runs[q.lvl].ops)
                    if (lo.kind == LVL_NORM && lo.dst_off == P->tensors[id].off) need[id] = 1;
            }
        } else if (q.kind == OP_ATTN) {
            if (q.lvl < 0) mark(q.at_qkv);
        } else if (q.kind == OP_CAST) {
            mark(q.cast_src);
        }
    }
    for (const auto& kv : P->named) mark(kv.second);
    std::unordered_map<int64_t, char> need_off;
    for (size_t t = 0; t < P->tensors.size(); ++t) {
        auto it = need_off.find(P->tensors[t].off);
        if (it == need_off.end()) need_off[P->tensors[t].off] = need[t];
        else it->second = it->second || need[t];
    }
    const bool keep_all = debug_route("lvl_keep_raw") != 0;
    for (LvlRun& R : P->lvl_runs) {
        R.ops_nograd = R.ops;
        R.raw_skipped = 0;
        if (keep_all) continue;
        for (LvlOp& lo : R.ops_nograd) {
            if (lo.kind != LVL_CONV || lo.dst_off < 0) continue;
            auto it = need_off.find(lo.dst_off);
            if (it != need_off.end() && !it->second && (lo.n_norm > 0 || lo.keep >= 0)) {  // (something of the op is still handed on)
                lo.dst_off = -1;
                ++R.raw_skipped;
            }
        }
    }
}

// diagnostic: in-kernel stamps of one workgroup of one level run (dmme_debug_level_stamps)
static long long* g_lvl_stamps = nullptr;
static int g_lvl_stamp_run = -1, g_lvl_stamp_wg = 0;

int run_level(const dmme_plan* P, const LvlRun& R, const char* pk, char* ws, int nt, const float* drop_masks, hipStream_t s, bool keep_ctx) {
    DMME_REQUIRE(R.ops_dev && R.sync_dev, DMME_ERR_INVALID, "level engine: the plan was created without a device");
    LvlArgs a{};
    a.ops = (!keep_ctx && R.ops_nograd_dev) ? R.ops_nograd_dev : R.ops_dev;
    a.n_ops = (int)R.ops.size();
    a.ws = ws;
    a.packed = pk;
    a.drop_masks = drop_masks;
    a.tproj = (const float*)(ws + P->ws_tproj);
    a.tproj_ld = P->tproj_cols;
    a.nt = nt;
    a.N = P->B;
    a.sh = R.sh;
    a.NG = R.NG;
    a.NGS = R.NGS;
    a.GB = R.GB;
    a.NJ = R.NJ;
    a.ctl = R.sync_dev;
    a.flags = R.sync_dev + 16;
    a.err_sys = P->err_host;
    a.run_tag = 1 + (int)(&R - P->lvl_runs.data());
    a.spin_limit = debug_route("lvl_spin", 0);
    a.withhold = debug_route("lvl_withhold", 0);
    a.xcd_group = debug_route("lvl_no_xcd") ? 0 : 1;
    a.max_wg = P->lvl_max_wg;
    if (g_lvl_stamps && g_lvl_stamp_run == (int)(&R - P->lvl_runs.data())) {
        a.stamps = g_lvl_stamps;
        a.stamp_wg = g_lvl_stamp_wg;
    }
    return launch_lvl_engine(P->dtype, a, s);
}

// The level engine's hand-off waits are bounded: a wait that gives up (a workgroup that was never scheduled - fewer free compute
// units than the launch needs - or a fault) lets the launch drain with WRONG numbers and raises the plan's host-visible status word.
// Every entry point that enqueues work on the plan looks at that word first, and dmme_unet_plan_check is the call for hosts that
// replay a captured graph (no entry point runs then): no path hands results on with rc 0 once the word is set.
int lvl_check(const dmme_plan* P, const char* where, hipStream_t stream, bool have_stream) {
    if (!P->err_host) return DMME_OK;
    const unsigned v = __atomic_load_n(P->err_host, __ATOMIC_ACQUIRE);
    if (!v) return DMME_OK;
    // clear: the device-side sticky words (a set word makes every later wait of that run give up after 1024 polls) and the host word
    // - unless the caller's stream is being captured (synchronising calls would invalidate the capture; the word stays set and the
    // next check outside a capture clears it)
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (have_stream && hipStreamIsCapturing(stream, &cs) != hipSuccess) cs = hipStreamCaptureStatusNone;
    if (cs == hipStreamCaptureStatusNone) {
        int cur = -1;  // synchronise the PLAN's device, whatever device the calling thread has current
        const bool sw = hipGetDevice(&cur) == hipSuccess && cur != P->device && hipSetDevice(P->device) == hipSuccess;
        (void)hipDeviceSynchronize();
        for (const LvlRun& R : P->lvl_runs)
            if (R.sync_dev) (void)hipMemset(R.sync_dev + 2, 0, 4);
        __atomic_store_n(P->err_host, 0u, __ATOMIC_RELEASE);
        if (sw) (void)hipSetDevice(cur);
    }
    const int ri = (int)v - 1;
    const LvlRun* R = ri >= 0 && ri < (int)P->lvl_runs.size() ? &P->lvl_runs[ri] : nullptr;
    set_error("%s: a hand-off wait of the level engine timed out (run %d, %dx%d maps, %d workgroups that must all be resident at once; the device "
              "holds %d): every output of this plan since the last check is invalid.  Typical cause: compute units held by another stream / "
              "process / CU mask.  DMME_NO_LVL=1 selects the per-op kernels.",
              where, ri, R ? 1 << R->sh : 0, R ? 1 << R->sh : 0, R ? R->NGS * (LVL_NS / R->NJ) : 0, P->lvl_max_wg);
    return DMME_ERR_HIP;
}

}  // namespace dmme

extern "C" {

DMME_API int dmme_unet_plan_check(const dmme_plan* plan) {
    DMME_REQUIRE(plan, DMME_ERR_INVALID, "plan_check: null plan");
    return lvl_check(plan, "plan_check");
}

DMME_API int dmme_unet_plan_level_info(const dmme_plan* plan, char* buf, int cap) {
    DMME_REQUIRE(plan && buf && cap > 0, DMME_ERR_INVALID, "level_info: bad argument");
    std::string out;
    char tmp[192];
    snprintf(tmp, sizeof(tmp), "runs=%d", (int)plan->lvl_runs.size());
    out = tmp;
    for (const LvlRun& R : plan->lvl_runs) {
        unsigned ctl[3] = {0, 0, 0};
        if (R.sync_dev) DMME_CHECK_HIP(hipMemcpy(ctl, R.sync_dev, sizeof(ctl), hipMemcpyDeviceToHost));  // (synchronises with the device)
        snprintf(tmp, sizeof(tmp), " [map=%dx%d plan_ops=%d-%d engine_ops=%d groups=%d per_iteration=%d slice=%d workgroups=%d nograd_raw_skipped=%d epoch=%u err=%u]", 1 << R.sh, 1 << R.sh,
                 R.op_first, R.op_last, (int)R.ops.size(), R.NG, R.GB, 32 * R.NJ, R.NGS * (LVL_NS / R.NJ), R.raw_skipped, ctl[0], ctl[2]);
        out += tmp;
    }
    strncpy(buf, out.c_str(), (size_t)cap - 1);
    buf[cap - 1] = 0;
    return DMME_OK;
}

DMME_API int dmme_debug_level_stamps(void* buf, int run, int workgroup) {
    g_lvl_stamps = (long long*)buf;
    g_lvl_stamp_run = run;
    g_lvl_stamp_wg = workgroup;
    return DMME_OK;
}
}  // extern "C"
